// Time attention core of the divided space-time attention (model/LaviLa.py:246-283 with '(b n) f d').
// One problem per (clip, head, patch location): T queries x (T frame keys + CLS key), head dim 64.
// Pure HBM-bound op (~(T+1)/2 flop/B): algorithmic bytes = every q,k,v row read once + every o row written once.
//
// Structure: a workgroup owns P = 128/T neighbouring patch locations of one (clip, head); K and V of its P*T tokens
// are staged once into LDS as [frame][patch][64] (each global read is a full 128-B row segment), thread (patch, frame)
// (2 lanes per query, 32 dims each)
// keeps its query row and fp32 output row in registers and walks the T+1 keys on the VALU; LDS reads of a key row are
// broadcast across the T threads that share the patch.
#include "common.h"

template <int T>
#define CLS_REC 68

__global__ __launch_bounds__(256, 4) void time_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                        float* __restrict__ cls_partial, int B, int n, int heads) {
    // thread = (patch pi, query frame fq, half hf): each query row is split over two adjacent lanes (32 dims each)
    constexpr int P = 128 / T;
    __shared__ __attribute__((aligned(16))) unsigned int Ks[(T * P + 1) * 32];   // 128 B rows; last row = CLS
    __shared__ __attribute__((aligned(16))) unsigned int Vs[(T * P + 1) * 32];
    __shared__ float Ss[(T + 1) * 128];                                             // scores [key][query]
    const int tid = threadIdx.x;
    const int D = heads * 64;
    const int64_t ld = 3 * (int64_t)D;
    const int N = 1 + T * n;
    const int groups = (n + P - 1) / P;
    int bid = blockIdx.x;
    const int pg = bid % groups; bid /= groups;
    const int head = bid % heads;
    const int b = bid / heads;
    const bf16_t* base = qkv + (int64_t)b * N * ld + head * 64;
    const int p0 = pg * P;

    for (int idx = tid; idx < (T * P + 1) * 8; idx += 256) {
        const int row = idx >> 3, c = idx & 7;
        int64_t tok;
        bool valid = true;
        if (row == T * P) tok = 0;
        else {
            const int fr = row / P, pi = row % P;
            valid = (p0 + pi) < n;
            tok = 1 + (int64_t)fr * n + p0 + pi;
        }
        u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
        if (valid) {
            kv = *(const u32x4*)(base + tok * ld + D + c * 8);
            vv = *(const u32x4*)(base + tok * ld + 2 * D + c * 8);
        }
        *(u32x4*)(Ks + idx * 4) = kv;
        *(u32x4*)(Vs + idx * 4) = vv;
    }
    const int hf = tid & 1, fq = (tid >> 1) % T, pi = (tid >> 1) / T;
    const bool active = (p0 + pi) < n;
    const int64_t qtok = 1 + (int64_t)fq * n + p0 + (active ? pi : 0);
    float q[32];
    {
        const bf16_t* qp = base + qtok * ld + hf * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x4 u = *(const u32x4*)(qp + c * 8);
#pragma unroll
            for (int w = 0; w < 4; ++w) { q[c * 8 + 2 * w] = bf16_lo_to_f32(u[w]); q[c * 8 + 2 * w + 1] = bf16_hi_to_f32(u[w]); }
        }
    }
    __syncthreads();
    // pass 1: scores -> LDS (one float per (key, query)), running max.  Rolled loops keep the register footprint at
    // q[32] + o[32] + temporaries so that 3-4 waves per SIMD hide the LDS / HBM latency.
    const int qi = tid >> 1;
    float mx = -INFINITY;
#pragma unroll 2
    for (int j = 0; j <= T; ++j) {
        const int row = (j == T) ? T * P : j * P + pi;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x4 u = *(const u32x4*)(Ks + row * 32 + hf * 16 + c * 4);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a0 = fmaf(q[c * 8 + 2 * w], bf16_lo_to_f32(u[w]), a0);
                a1 = fmaf(q[c * 8 + 2 * w + 1], bf16_hi_to_f32(u[w]), a1);
            }
        }
        float sj = a0 + a1;
        sj += __shfl_xor(sj, 1, 64);
        if (hf == 0) Ss[j * 128 + qi] = sj;
        mx = fmaxf(mx, sj);
    }
    // pass 2: p = exp(s - max), l, o += p * v   (the pair's partner lane wrote Ss: same wave, in-order LDS)
    float l = 0.f;
    float o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
    const float mb = mx * 1.4426950408889634f;
#pragma unroll 2
    for (int j = 0; j <= T; ++j) {
        const int row = (j == T) ? T * P : j * P + pi;
        const float pj = __builtin_amdgcn_exp2f(Ss[j * 128 + qi] * 1.4426950408889634f - mb);
        l += pj;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x4 u = *(const u32x4*)(Vs + row * 32 + hf * 16 + c * 4);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                o[c * 8 + 2 * w] = fmaf(pj, bf16_lo_to_f32(u[w]), o[c * 8 + 2 * w]);
                o[c * 8 + 2 * w + 1] = fmaf(pj, bf16_hi_to_f32(u[w]), o[c * 8 + 2 * w + 1]);
            }
        }
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] *= inv;
    if (active) {
        bf16_t* op = out + ((int64_t)b * N + qtok) * D + head * 64 + hf * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x4 w = {pack_bf16(o[c * 8], o[c * 8 + 1]), pack_bf16(o[c * 8 + 2], o[c * 8 + 3]),
                       pack_bf16(o[c * 8 + 4], o[c * 8 + 5]), pack_bf16(o[c * 8 + 6], o[c * 8 + 7])};
            *(u32x4*)(op + c * 8) = w;
        }
    }
    // ---- CLS query folded in: partial over this workgroup's T*P keys (already in LDS); group 0 also counts the CLS key
    if (cls_partial == nullptr) return;
    __shared__ float cp[T * P + 1];
    __shared__ float cred[8];
    {
        const int nrows = T * P + (pg == 0 ? 1 : 0);                 // LDS row T*P is the CLS key/value
        float sc = -INFINITY;
        if (tid < nrows) {
            const int fr = tid / P, pi2 = tid % P;
            const bool valid = tid == T * P || (p0 + pi2) < n;
            if (valid) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    u32x4 qv = *(const u32x4*)(base + c * 8);            // q row of token 0 (uniform address)
                    u32x4 u = *(const u32x4*)(Ks + tid * 32 + c * 4);
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        a0 = fmaf(bf16_lo_to_f32(qv[w]), bf16_lo_to_f32(u[w]), a0);
                        a1 = fmaf(bf16_hi_to_f32(qv[w]), bf16_hi_to_f32(u[w]), a1);
                    }
                }
                sc = a0 + a1;
            }
            (void)fr;
        }
        float mx = wave_max(sc);
        if ((tid & 63) == 0) cred[tid >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(cred[0], cred[1]), fmaxf(cred[2], cred[3]));
        const float pj = (sc == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((sc - mx) * 1.4426950408889634f);
        if (tid <= T * P) cp[tid] = pj;
        float l = wave_sum(pj);
        if ((tid & 63) == 0) cred[4 + (tid >> 6)] = l;
        __syncthreads();
        if (tid < 64) {
            float o = 0.f;
            const unsigned short* vs16 = (const unsigned short*)Vs;
            for (int j = 0; j < nrows; ++j) o = fmaf(cp[j], __uint_as_float((unsigned)vs16[j * 64 + tid] << 16), o);
            float* rec = cls_partial + (((int64_t)b * heads + head) * groups + pg) * CLS_REC;
            rec[4 + tid] = o;
            if (tid == 0) { rec[0] = mx; rec[1] = (cred[4] + cred[5]) + (cred[6] + cred[7]); }
        }
    }
}

extern "C" int hh_time_attn_fwd(const void* qkv, void* out, float* cls_partial, int B, int T, int n, int heads, hh_stream_t stream) {
    HH_REQUIRE(B >= 0 && n > 0 && heads > 0, HH_ERR_SHAPE, "hh_time_attn_fwd: bad shape");
    HH_REQUIRE(T == 1 || T == 2 || T == 4 || T == 8 || T == 16 || T == 32, HH_ERR_UNSUPPORTED,
               "hh_time_attn_fwd: num_frames=%d unsupported (1,2,4,8,16,32)", T);
    HH_REQUIRE(HH_ALIGNED16(qkv) && HH_ALIGNED16(out), HH_ERR_ALIGN, "hh_time_attn_fwd: pointers must be 16-byte aligned");
    if (B == 0) return HH_OK;
    const int P = 128 / T;
    const int64_t blocks = (int64_t)B * heads * ((n + P - 1) / P);
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* in = (const bf16_t*)qkv;
    bf16_t* o = (bf16_t*)out;
#define LAUNCH(TT) hipLaunchKernelGGL(time_attn_kernel<TT>, dim3((unsigned)blocks), dim3(256), 0, s, in, o, cls_partial, B, n, heads)
    switch (T) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 4: LAUNCH(4); break;
        case 8: LAUNCH(8); break;
        case 16: LAUNCH(16); break;
        default: LAUNCH(32); break;
    }
#undef LAUNCH
    return hh_check_launch("hh_time_attn_fwd");
}

// Space attention core of the divided space-time attention (model/LaviLa.py:246-283 with '(b f) n d', attn() :194-198).
// One workgroup per (clip, frame, head): n queries x (n frame keys + CLS key), head dim 64, no mask, no dropout.
// HBM-bound in principle: algorithmic bytes per workgroup = (3*n + 2) * 128 B read + n * 128 B written.
//
// Structure: K tile [KP][64] (XOR-swizzled 16-B chunks) and V^T tile [64][KP+4] staged once in LDS (V is transposed
// while staging: 4 keys x 8 d register blocks -> ds_write_b64), then each wave walks 32-query blocks:
//   S^T = K . Q^T      mfma_f32_32x32x16_bf16, A = K rows (LDS), B = Q rows (registers, straight from HBM)
//   softmax over keys  in-lane over the 16 accumulator registers x tiles + one xor-32 shuffle (query = lane & 31)
//   O^T = V^T . P^T    the S^T accumulator tile is re-used as the B operand with no lane movement
//                      (k order inside a step: row 16s + 8(j>>2) + 4h + (j&3)); A = V^T rows (2 x ds_read_b64)
// Keys are processed in chunks of CH=5 tiles (160 keys) with an online-softmax merge between chunks, so n=576
// (336^2) runs through the same code.  Frame keys sit at rows 0..n-1, the CLS key at row n, rows > n are zero/masked.
#include "common.h"

#define CH 5

__device__ __forceinline__ u32x2 pack_keys(unsigned a, unsigned b, unsigned c, unsigned d, bool hi) {
    // pick the low or high bf16 of each of 4 dwords (4 keys, same d) -> 4 bf16 in key order
    u32x2 r;
    if (!hi) {
        r[0] = (a & 0xffffu) | (b << 16);
        r[1] = (c & 0xffffu) | (d << 16);
    } else {
        r[0] = (a >> 16) | (b & 0xffff0000u);
        r[1] = (c >> 16) | (d & 0xffff0000u);
    }
    return r;
}

// layout of one CLS partial record: [m, l, 0, 0, o[64]] fp32
#define CLS_REC 68

__global__ __launch_bounds__(256, 2) void space_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ cls_partial,
                                                            int B, int T, int n, int heads, int KP, int VS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                              // [KP][128 B]
    bf16_t* Vt = (bf16_t*)(smem + (size_t)KP * 128);   // [64][VS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = heads * 64;
    const int64_t ld = 3 * (int64_t)D;
    const int N = 1 + T * n;
    int bid = blockIdx.x;
    const int head = bid % heads; bid /= heads;
    const int f = bid % T;
    const int b = bid / T;
    const bf16_t* base = qkv + (int64_t)b * N * ld + head * 64;
    const bf16_t* q_ptr = base + (int64_t)(1 + f * n) * ld;
    const bf16_t* k_ptr = q_ptr + D;
    const bf16_t* v_ptr = q_ptr + 2 * D;
    const bf16_t* kc_ptr = base + D;       // CLS key / value (token 0)
    const bf16_t* vc_ptr = base + 2 * D;

    // ---- stage K (swizzled), rows >= n+1 zero
    for (int idx = tid; idx < KP * 8; idx += 256) {
        const int row = idx >> 3, pc = idx & 7, c = pc ^ (row & 7);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < n) v = *(const u32x4*)(k_ptr + (int64_t)row * ld + c * 8);
        else if (row == n) v = *(const u32x4*)(kc_ptr + c * 8);
        *(u32x4*)(Ks + idx * 16) = v;
    }
    // ---- stage V transposed: 4 keys x 8 d per task
    for (int idx = tid; idx < (n >> 2) * 8; idx += 256) {
        const int kg = idx >> 3, c = idx & 7;
        const bf16_t* src = v_ptr + (int64_t)(kg * 4) * ld + c * 8;
        u32x4 r0 = *(const u32x4*)(src), r1 = *(const u32x4*)(src + ld), r2 = *(const u32x4*)(src + 2 * ld),
              r3 = *(const u32x4*)(src + 3 * ld);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            *(u32x2*)(Vt + (size_t)(c * 8 + 2 * w) * VS + kg * 4) = pack_keys(r0[w], r1[w], r2[w], r3[w], false);
            *(u32x2*)(Vt + (size_t)(c * 8 + 2 * w + 1) * VS + kg * 4) = pack_keys(r0[w], r1[w], r2[w], r3[w], true);
        }
    }
    // CLS value at key n, zeros for keys n+1 .. VS-1
    for (int idx = tid; idx < 64 * (VS - n); idx += 256) {
        const int d = idx / (VS - n), kk = n + idx % (VS - n);
        Vt[(size_t)d * VS + kk] = (kk == n) ? vc_ptr[d] : (bf16_t)0.f;
    }
    __syncthreads();

    const int ql = lane & 31, h = lane >> 5;
    const int ntiles = KP >> 5;
    const float LOG2E = 1.4426950408889634f;
    for (int qb = wave; qb < (n >> 5); qb += 4) {
        // Q fragments: B operand, lane (col q = ql, half h) holds Q[q][16*ks + 8*h .. +8]
        bf16x8 qf[4];
        const bf16_t* qrow = q_ptr + (int64_t)(qb * 32 + ql) * ld + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qrow + 16 * ks);
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        float m_run = -INFINITY, l_run = 0.f;
        for (int t0 = 0; t0 < ntiles; t0 += CH) {
            f32x16 s[CH];
#pragma unroll
            for (int ti = 0; ti < CH; ++ti) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[ti][r] = 0.f;
                if (t0 + ti < ntiles) {
                    const int krow = (t0 + ti) * 32 + ql;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int c = 2 * ks + h;
                        bf16x8 kf = *(const bf16x8*)(Ks + krow * 128 + ((c ^ (krow & 7)) << 4));
                        s[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[ti], 0, 0, 0);
                    }
                }
            }
            // mask + running max   (key index of register r in tile ti: 32*(t0+ti) + (r&3) + 8*(r>>2) + 4*h)
            float mx = -INFINITY;
#pragma unroll
            for (int ti = 0; ti < CH; ++ti) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = (t0 + ti) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key > n) s[ti][r] = -INFINITY;
                    mx = fmaxf(mx, s[ti][r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);     // __builtin_amdgcn_exp2f(-inf) = 0 on the first chunk
            const float mb = m_new * LOG2E;
            float lsum = 0.f;
#pragma unroll
            for (int ti = 0; ti < CH; ++ti)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(s[ti][r] * LOG2E - mb);
                    s[ti][r] = pv;
                    lsum += pv;
                }
            lsum += __shfl_xor(lsum, 32, 64);
            l_run = l_run * alpha + lsum;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            // O^T += V^T . P^T
#pragma unroll
            for (int ti = 0; ti < CH; ++ti) {
                if (t0 + ti < ntiles) {
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        bf16x8 pf;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) pf[jj] = (bf16_t)s[ti][8 * st + jj];
                        const int key0 = (t0 + ti) * 32 + 16 * st + 4 * h;
                        const bf16_t* v0 = Vt + (size_t)ql * VS + key0;
                        const bf16_t* v1 = Vt + (size_t)(32 + ql) * VS + key0;
                        bf16x4 a0 = *(const bf16x4*)(v0), a1 = *(const bf16x4*)(v0 + 8);
                        bf16x4 c0 = *(const bf16x4*)(v1), c1 = *(const bf16x4*)(v1 + 8);
                        bf16x8 vf0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                        bf16x8 vf1 = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pf, o0, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, pf, o1, 0, 0, 0);
                    }
                }
            }
        }
        // normalise and store: lane owns query ql; register r of o{0,1} is d = 32*dt + (r&3) + 8*(r>>2) + 4*h
        const float inv = 1.f / l_run;
        bf16_t* orow = out + ((int64_t)b * N + 1 + f * n + qb * 32 + ql) * D + head * 64 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 w0 = {pack_bf16(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack_bf16(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv)};
            u32x2 w1 = {pack_bf16(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack_bf16(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv)};
            *(u32x2*)(orow + 8 * g) = w0;
            *(u32x2*)(orow + 32 + 8 * g) = w1;
        }
    }
    // ---- CLS query (model/LaviLa.py:255-258) folded in: partial softmax(q_cls . K_f^T) V_f over THIS frame's keys, which
    // are already in LDS (the CLS key itself is counted by frame 0 only); hh_cls_combine merges the T partials.
    if (cls_partial == nullptr) return;
    __syncthreads();                                   // every wave is done with its reads; reuse the front of Ks as scratch
    {
        float* scratch = (float*)(smem + (size_t)KP * 128 + (size_t)64 * VS * 2);     // [4 waves][CLS_REC]
        const int nkeys = n + (f == 0 ? 1 : 0);
        float q[64];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            u32x4 u = *(const u32x4*)(base + c * 8);                                 // q row of token 0 (pre-scaled)
#pragma unroll
            for (int w = 0; w < 4; ++w) { q[c * 8 + 2 * w] = bf16_lo_to_f32(u[w]); q[c * 8 + 2 * w + 1] = bf16_hi_to_f32(u[w]); }
        }
        // wave w owns keys [w*KQ, (w+1)*KQ), one or more keys per lane
        const int KQ = (nkeys + 3) / 4;
        const int k_lo = wave * KQ, k_hi = min(nkeys, k_lo + KQ);
        float mx = -INFINITY;
        for (int j = k_lo + lane; j < k_hi; j += 64) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                u32x4 u = *(const u32x4*)(Ks + j * 128 + ((c ^ (j & 7)) << 4));
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    a0 = fmaf(q[c * 8 + 2 * w], bf16_lo_to_f32(u[w]), a0);
                    a1 = fmaf(q[c * 8 + 2 * w + 1], bf16_hi_to_f32(u[w]), a1);
                }
            }
            mx = fmaxf(mx, a0 + a1);
        }
        mx = wave_max(mx);
        // second pass: p_j and the weighted V sum; lane d accumulates o[d] with p_j broadcast through readlane
        float l = 0.f, o = 0.f;
        for (int j0 = k_lo; j0 < k_hi; j0 += 64) {
            const int j = j0 + lane;
            float pj = 0.f;
            if (j < k_hi) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    u32x4 u = *(const u32x4*)(Ks + j * 128 + ((c ^ (j & 7)) << 4));
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        a0 = fmaf(q[c * 8 + 2 * w], bf16_lo_to_f32(u[w]), a0);
                        a1 = fmaf(q[c * 8 + 2 * w + 1], bf16_hi_to_f32(u[w]), a1);
                    }
                }
                pj = __builtin_amdgcn_exp2f((a0 + a1 - mx) * LOG2E);
            }
            l += pj;
            const int cnt = min(64, k_hi - j0);
            const bf16_t* vrow = Vt + (size_t)lane * VS + j0;                      // V^T row d = lane
            for (int jj = 0; jj < cnt; ++jj) o = fmaf(__shfl(pj, jj, 64), (float)vrow[jj], o);
        }
        l = wave_sum(l);
        if (k_lo >= k_hi) { mx = -INFINITY; l = 0.f; o = 0.f; }
        scratch[wave * CLS_REC + 4 + lane] = o;
        if (lane == 0) { scratch[wave * CLS_REC] = mx; scratch[wave * CLS_REC + 1] = l; }
        __syncthreads();
        if (tid < 64) {
            float m = fmaxf(fmaxf(scratch[0], scratch[CLS_REC]), fmaxf(scratch[2 * CLS_REC], scratch[3 * CLS_REC]));
            float lt = 0.f, ot = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float e = __builtin_amdgcn_exp2f((scratch[w * CLS_REC] - m) * LOG2E);     // exp2(-inf) = 0 for empty waves
                lt += scratch[w * CLS_REC + 1] * e;
                ot += scratch[w * CLS_REC + 4 + tid] * e;
            }
            float* rec = cls_partial + (((int64_t)b * heads + head) * T + f) * CLS_REC;
            rec[4 + tid] = ot;
            if (tid == 0) { rec[0] = m; rec[1] = lt; }
        }
    }
}

// merge G partial records per (clip, head) into out row 0:  o = sum_g o_g e^{m_g - m} / sum_g l_g e^{m_g - m}
__global__ __launch_bounds__(64) void cls_combine_kernel(const float* __restrict__ partial, int G, bf16_t* __restrict__ out,
                                                         int N, int heads) {
    const int head = blockIdx.x % heads, b = blockIdx.x / heads, d = threadIdx.x;
    const float* rec = partial + ((int64_t)b * heads + head) * G * CLS_REC;
    float m = -INFINITY;
    for (int g = 0; g < G; ++g) m = fmaxf(m, rec[g * CLS_REC]);
    float l = 0.f, o = 0.f;
    for (int g = 0; g < G; ++g) {
        const float e = __builtin_amdgcn_exp2f((rec[g * CLS_REC] - m) * 1.4426950408889634f);
        l += rec[g * CLS_REC + 1] * e;
        o += rec[g * CLS_REC + 4 + d] * e;
    }
    out[(int64_t)b * N * heads * 64 + head * 64 + d] = (bf16_t)(o / l);
}

extern "C" int hh_cls_combine(const float* partial, int G, void* out, int B, int N, int heads, hh_stream_t stream) {
    HH_REQUIRE(B >= 0 && G > 0 && N > 0 && heads > 0 && partial != nullptr, HH_ERR_SHAPE, "hh_cls_combine: bad arguments");
    if (B == 0) return HH_OK;
    hipLaunchKernelGGL(cls_combine_kernel, dim3((unsigned)(B * heads)), dim3(64), 0, (hipStream_t)stream, partial, G, (bf16_t*)out, N, heads);
    return hh_check_launch("hh_cls_combine");
}

extern "C" int hh_space_attn_fwd(const void* qkv, void* out, float* cls_partial, int B, int T, int n, int heads, hh_stream_t stream) {
    HH_REQUIRE(B >= 0 && T > 0 && heads > 0 && n > 0 && n % 32 == 0, HH_ERR_SHAPE, "hh_space_attn_fwd: n=%d must be a multiple of 32", n);
    HH_REQUIRE(HH_ALIGNED16(qkv) && HH_ALIGNED16(out), HH_ERR_ALIGN, "hh_space_attn_fwd: pointers must be 16-byte aligned");
    if (B == 0) return HH_OK;
    const int KP = ((n + 1 + 31) / 32) * 32;
    const int VS = KP + 4;
    const size_t lds = (size_t)KP * 128 + (size_t)64 * VS * 2 + 4 * CLS_REC * 4;
    HH_REQUIRE(lds <= 160 * 1024, HH_ERR_UNSUPPORTED, "hh_space_attn_fwd: n=%d needs %zu B of LDS (> 160 KiB)", n, lds);
    static size_t attr_set = 0;
    if (lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)space_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_space_attn_fwd: cannot reserve %zu B of LDS", lds);
        attr_set = lds;
    }
    const int64_t blocks = (int64_t)B * T * heads;
    hipLaunchKernelGGL(space_attn_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream,
                       (const bf16_t*)qkv, (bf16_t*)out, cls_partial, B, T, n, heads, KP, VS);
    return hh_check_launch("hh_space_attn_fwd");
}

// CLS-query attention: the CLS query of each (clip, head) attends ALL N keys (model/LaviLa.py:255-258), in both
// the time and the space attention of every block.  1 x N x 64 per problem; HBM-bound (reads K and V once).
// One workgroup per (clip, head): phase 1 scores (thread per key, fp32 dot with the query), block softmax,
// phase 2 weighted V sum with 8 threads per key row (full 128-B row reads).
#include "common.h"

__global__ __launch_bounds__(256) void cls_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                       int B, int N, int heads, float exp2_scale, int layout) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [N] scores + 64 q + 32*64 partial o + 8 red
    float* sc = sm;
    float* qs = sm + ((N + 3) & ~3);
    float* part = qs + 64;
    float* red = part + 32 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = heads * 64;
    // qkv layout (include/hh.h, hh_qkv_layout): element (row, which, head, d) at row * ld + which * ws + head * hs + d
    const int64_t ld = layout ? 64 : 3 * (int64_t)D;
    const int64_t hs = layout ? (int64_t)B * N * 64 : 64, ws = (int64_t)heads * hs;
    const int head = blockIdx.x % heads, b = blockIdx.x / heads;
    const bf16_t* base = qkv + (int64_t)b * N * ld + head * hs;
    if (tid < 64) qs[tid] = (float)base[tid];
    __syncthreads();
    float q[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) q[d] = qs[d];
    float mx = -INFINITY;
    for (int j = tid; j < N; j += 256) {
        const bf16_t* kp = base + (int64_t)j * ld + ws;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            u32x4 u = *(const u32x4*)(kp + c * 8);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a0 = fmaf(q[c * 8 + 2 * w], bf16_lo_to_f32(u[w]), a0);
                a1 = fmaf(q[c * 8 + 2 * w + 1], bf16_hi_to_f32(u[w]), a1);
            }
        }
        const float s = a0 + a1;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float l = 0.f;
    for (int j = tid; j < N; j += 256) {
        const float p = __builtin_amdgcn_exp2f((sc[j] - mx) * exp2_scale);      // log2(e), or 1 when q already carries it (space attention)
        sc[j] = p;
        l += p;
    }
    l = wave_sum(l);
    if (lane == 0) red[4 + wave] = l;
    __syncthreads();
    l = (red[4] + red[5]) + (red[6] + red[7]);
    // phase 2: thread (kl = tid>>3, c = tid&7) accumulates d = 8c..8c+7 over keys kl, kl+32, ...
    const int kl = tid >> 3, c = tid & 7;
    float o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = 0.f;
    for (int j = kl; j < N; j += 32) {
        const float p = sc[j];
        u32x4 u = *(const u32x4*)(base + (int64_t)j * ld + 2 * ws + c * 8);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            o[2 * w] = fmaf(p, bf16_lo_to_f32(u[w]), o[2 * w]);
            o[2 * w + 1] = fmaf(p, bf16_hi_to_f32(u[w]), o[2 * w + 1]);
        }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) part[kl * 64 + c * 8 + d] = o[d];
    __syncthreads();
    if (tid < 64) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) a += part[k * 64 + tid];
        out[(int64_t)b * N * D + head * 64 + tid] = (bf16_t)(a / l);
    }
}

extern "C" int hh_cls_attn_fwd(const void* qkv, int qkv_layout, void* out, int B, int N, int heads, int q_log2, hh_stream_t stream) {
    HH_REQUIRE(qkv_layout == HH_QKV_TOKEN_MAJOR || qkv_layout == HH_QKV_HEAD_MAJOR, HH_ERR_SHAPE, "hh_cls_attn_fwd: bad qkv_layout");
    HH_REQUIRE(B >= 0 && N > 0 && heads > 0, HH_ERR_SHAPE, "hh_cls_attn_fwd: bad shape");
    HH_REQUIRE(HH_ALIGNED16(qkv) && HH_ALIGNED16(out), HH_ERR_ALIGN, "hh_cls_attn_fwd: pointers must be 16-byte aligned");
    if (B == 0) return HH_OK;
    const size_t lds = (size_t)(((N + 3) & ~3) + 64 + 32 * 64 + 8) * 4;
    HH_REQUIRE(lds <= 160 * 1024, HH_ERR_UNSUPPORTED, "hh_cls_attn_fwd: N=%d needs %zu B of LDS", N, lds);
    static size_t attr_set = 65536;
    if (lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)cls_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_cls_attn_fwd: cannot reserve %zu B of LDS", lds);
        attr_set = lds;
    }
    hipLaunchKernelGGL(cls_attn_kernel, dim3((unsigned)(B * heads)), dim3(256), lds, (hipStream_t)stream,
                       (const bf16_t*)qkv, (bf16_t*)out, B, N, heads, q_log2 ? 1.f : 1.4426950408889634f, qkv_layout);
    return hh_check_launch("hh_cls_attn_fwd");
}

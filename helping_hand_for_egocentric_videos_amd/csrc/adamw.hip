// Fused AdamW over a flat fp32 arena (torch.optim.AdamW defaults semantics; run/train.py:199-203,520,
// utils/train_utils.py:28-48 param groups = two calls with different weight_decay).  HBM-bound: 28 B/param.
#include "common.h"

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * (1.f - lr * wd);
        const float mi = m[i] * b1 + (1.f - b1) * gi;
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

extern "C" int hh_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, hh_stream_t stream) {
    HH_REQUIRE(n >= 0 && step >= 1, HH_ERR_SHAPE, "hh_adamw_step: need n >= 0 and step >= 1");
    if (n == 0) return HH_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2));
    return hh_check_launch("hh_adamw_step");
}

// ---- device-driven AdamW over the whole arena (one call per step, no host-side launch plan) --------------------------------------
// The arena is a sequence of segments (one per parameter, 4-element aligned).  Which segments are updated, and with which bias
// correction, is decided ON THE DEVICE from seg_flag (> 0: the parameter received a gradient on some rank -- under data parallelism
// the flags are all-reduced, so every rank takes the same decision) and the per-segment step counters seg_step.
__global__ void adamw_seg_prepare_kernel(const float* __restrict__ seg_flag, int* __restrict__ seg_step, float* __restrict__ seg_coef,
                                         int n_seg, float lr, float b1, float b2) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    if (seg_flag[s] > 0.f) {
        const int t = seg_step[s] + 1;
        seg_step[s] = t;
        const double bc1 = 1.0 - pow((double)b1, (double)t);
        const double bc2 = 1.0 - pow((double)b2, (double)t);
        seg_coef[2 * s] = (float)bc1;
        seg_coef[2 * s + 1] = (float)sqrt(bc2);
    } else {
        seg_coef[2 * s] = 0.f;                       // 0 = leave p, m, v alone (torch.optim.AdamW skips grad-less parameters entirely)
        seg_coef[2 * s + 1] = 1.f;
    }
}

#define ADAMW_CHUNK 8192                             // elements per workgroup: 256 lanes x float4 x 8 iterations
__global__ __launch_bounds__(256) void adamw_arena_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, int64_t n, const int64_t* __restrict__ seg_off,
                                                          const int* __restrict__ seg_decay, const float* __restrict__ seg_coef,
                                                          int n_seg, float lr, float b1, float b2, float eps, float wd, int zero_grads) {
    const int64_t base = (int64_t)blockIdx.x * ADAMW_CHUNK + (int64_t)threadIdx.x * 4;
    if (base >= n) return;
    int lo = 0, hi = n_seg - 1;                      // segment of `base`: last s with seg_off[s] <= base
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg_off[mid] <= base) lo = mid; else hi = mid - 1;
    }
    int s = lo;
    int64_t s_end = seg_off[s + 1];
    float bc1 = seg_coef[2 * s], bc2s = seg_coef[2 * s + 1];
    float wds = seg_decay[s] ? wd : 0.f;
#pragma unroll 1
    for (int it = 0; it < ADAMW_CHUNK / 1024; ++it) {
        const int64_t i = base + (int64_t)it * 1024;
        if (i >= n) break;
        while (i >= s_end) {                         // segments are 4-element aligned: a float4 never straddles two
            ++s;
            s_end = seg_off[s + 1];
            bc1 = seg_coef[2 * s]; bc2s = seg_coef[2 * s + 1];
            wds = seg_decay[s] ? wd : 0.f;
        }
        const f32x4 gi = *reinterpret_cast<const f32x4*>(g + i);
        if (bc1 > 0.f) {
            f32x4 pi = *reinterpret_cast<const f32x4*>(p + i);
            f32x4 mi = *reinterpret_cast<const f32x4*>(m + i);
            f32x4 vi = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {            // same formulas as adamw_kernel
                float pe = pi[e] * (1.f - lr * wds);
                const float me = mi[e] * b1 + (1.f - b1) * gi[e];
                const float ve = vi[e] * b2 + (1.f - b2) * gi[e] * gi[e];
                const float denom = sqrtf(ve) / bc2s + eps;
                pe -= (lr / bc1) * (me / denom);
                pi[e] = pe; mi[e] = me; vi[e] = ve;
            }
            *reinterpret_cast<f32x4*>(p + i) = pi;
            *reinterpret_cast<f32x4*>(m + i) = mi;
            *reinterpret_cast<f32x4*>(v + i) = vi;
        }
        if (zero_grads) *reinterpret_cast<f32x4*>(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

extern "C" int hh_adamw_arena_step(float* p, float* g, float* m, float* v, int64_t n, const int64_t* seg_off, const int* seg_decay,
                                   int* seg_step, const float* seg_flag, float* seg_coef, int n_seg, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, int zero_grads, hh_stream_t stream) {
    HH_REQUIRE(n >= 0 && n_seg >= 1 && n % 4 == 0, HH_ERR_SHAPE, "hh_adamw_arena_step: need n >= 0, n % 4 == 0, n_seg >= 1 (got n=%lld, n_seg=%d)",
               (long long)n, n_seg);
    HH_REQUIRE(HH_ALIGNED16(p) && HH_ALIGNED16(g) && HH_ALIGNED16(m) && HH_ALIGNED16(v), HH_ERR_ALIGN,
               "hh_adamw_arena_step: p, g, m, v must be 16-byte aligned");
    if (n == 0) return HH_OK;
    hipLaunchKernelGGL(adamw_seg_prepare_kernel, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seg_flag,
                       seg_step, seg_coef, n_seg, lr, beta1, beta2);
    const int64_t blocks = (n + ADAMW_CHUNK - 1) / ADAMW_CHUNK;
    hipLaunchKernelGGL(adamw_arena_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, seg_off, seg_decay,
                       seg_coef, n_seg, lr, beta1, beta2, eps, weight_decay, zero_grads);
    return hh_check_launch("hh_adamw_arena_step");
}

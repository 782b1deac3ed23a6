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

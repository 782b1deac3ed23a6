// Can one SIMD overlap MFMA and transcendental VALU work?  Per loop iteration: 36 v_mfma_f32_16x16x32_bf16 and/or 32 v_exp_f32 + 16
// v_cvt_pk_bf16_f32 (one main-loop chunk of the joint-block space-attention kernel), register-only, W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 scripts/issue_probe.hip -o scripts/_bin/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)
template <int MODE>   // 0 MFMA only, 1 VALU only, 2 sequential (MFMAs then VALU), 3 interleaved 1 MFMA : 1 exp (+ cvt)
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(seed + i + j + threadIdx.x * 0.001f); b[i][j] = (__bf16)(seed * 0.5f + i - j); }
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float e[32];
    for (int i = 0; i < 32; ++i) e[i] = -seed * (i + 1) * 0.01f - threadIdx.x * 1e-4f;
    unsigned pk = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int m = 0; m < 36; ++m) acc[m % 12] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 3], b[(m >> 2) & 3], acc[m % 12], 0, 0, 0);
            FENCE();
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]) - 1.5f;
#pragma unroll
            for (int i = 0; i < 32; i += 2) { typedef __bf16 b2 __attribute__((ext_vector_type(2))); b2 v = {(__bf16)e[i], (__bf16)e[i + 1]}; pk ^= __builtin_bit_cast(unsigned, v); }
            FENCE();
        }
        if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < 36; ++m) {
                acc[m % 12] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 3], b[(m >> 2) & 3], acc[m % 12], 0, 0, 0);
                if (m < 32) e[m] = __builtin_amdgcn_exp2f(e[m]) - 1.5f;
                if (m >= 16 && m < 32) { typedef __bf16 b2 __attribute__((ext_vector_type(2))); const int i = 2 * (m - 16); b2 v = {(__bf16)e[i], (__bf16)e[i + 1]}; pk ^= __builtin_bit_cast(unsigned, v); }
                FENCE();
            }
        }
    }
    float sum = __uint_as_float(pk & 0xff);
    for (int i = 0; i < 12; ++i) sum += acc[i][0];
    for (int i = 0; i < 32; ++i) sum += e[i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}
template <int MODE> float run(float* out, int wgs, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD = workgroups (4 waves each) per CU
        const int wgs = 256 * wps;
        const float t0 = run<0>(out, wgs, iters), t1 = run<1>(out, wgs, iters), t2 = run<2>(out, wgs, iters), t3 = run<3>(out, wgs, iters);
        const double per = 1e6 / iters / wps;        // ns per iteration and wave on its SIMD
        printf("%d wave(s)/SIMD: ns per chunk-equivalent  MFMA only %6.1f | VALU only %6.1f | sequential %6.1f | interleaved %6.1f\n", wps, t0 * per, t1 * per, t2 * per, t3 * per);
    }
    return 0;
}

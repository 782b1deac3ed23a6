// Pure-MFMA power probe: sustained TFLOP/s and package power of register-only MFMA loops (no LDS, no memory) for the two bf16 shapes.
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_power.hip -o scripts/_bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512) void burn(float* out, int iters, unsigned seed) {
    bf16x8 a[4], b[4];
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u; a[i][j] = (__bf16)(((int)(s >> 16) & 0xff) / 128.0f - 1.0f);
            s = s * 1664525u + 1013904223u; b[i][j] = (__bf16)(((int)(s >> 16) & 0xff) / 128.0f - 1.0f);
        }
    float sum = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][15];
    }
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 16;
    const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;          // per launch: 256 blocks x 8 waves x iters x 16 (or 8) MFMAs
    const double flop_per_launch = 256.0 * 8 * iters * 16 * 16384.0;     // both shapes: 16 x 16384 = 8 x 32768 flop per iteration
    auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    double el = 0;
    while (el < seconds) {
        for (int i = 0; i < 10; ++i) {
            if (shape == 16) hipLaunchKernelGGL(burn<16>, dim3(256), dim3(512), 0, 0, out, iters, 1u);
            else hipLaunchKernelGGL(burn<32>, dim3(256), dim3(512), 0, 0, out, iters, 1u);
        }
        hipDeviceSynchronize(); launches += 10;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("shape %dx: %.1f TFLOP/s sustained over %.1f s\n", shape, launches * flop_per_launch / el / 1e12, el);
    return 0;
}

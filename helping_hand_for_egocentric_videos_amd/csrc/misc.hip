// Casts and transposes (host-side plumbing: bf16 weight copies, wgrad operand transposes).  HBM-bound.
#include "common.h"

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    const int64_t stride = (int64_t)gridDim.x * 256 * 8;
    for (; i + 8 <= n; i += stride) {
        f32x4 a = *(const f32x4*)(x + i), b = *(const f32x4*)(x + i + 4);
        u32x4 o = {pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(b[0], b[1]), pack_bf16(b[2], b[3])};
        *(u32x4*)(y + i) = o;
    }
    if (i < n && i + 8 > n)
        for (int64_t k = i; k < n; ++k) y[k] = (bf16_t)x[k];
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    const int64_t stride = (int64_t)gridDim.x * 256 * 8;
    for (; i + 8 <= n; i += stride) {
        u32x4 u = *(const u32x4*)(x + i);
        f32x4 a = {bf16_lo_to_f32(u[0]), bf16_hi_to_f32(u[0]), bf16_lo_to_f32(u[1]), bf16_hi_to_f32(u[1])};
        f32x4 b = {bf16_lo_to_f32(u[2]), bf16_hi_to_f32(u[2]), bf16_lo_to_f32(u[3]), bf16_hi_to_f32(u[3])};
        *(f32x4*)(y + i) = a;
        *(f32x4*)(y + i + 4) = b;
    }
    if (i < n && i + 8 > n)
        for (int64_t k = i; k < n; ++k) y[k] = (float)x[k];
}

// 64x64 tile transpose through LDS (padded), output bf16.
template <typename TIN>
__global__ __launch_bounds__(256) void transpose_kernel(const TIN* __restrict__ x, int64_t ldx, bf16_t* __restrict__ y,
                                                        int64_t ldy, int64_t rows, int64_t cols) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? (float)x[r * ldx + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int64_t c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) y[c * ldy + r] = (bf16_t)tile[tx][i];
    }
}

extern "C" int hh_cast_f32_to_bf16(const float* x, void* y, int64_t n, hh_stream_t stream) {
    HH_REQUIRE(n >= 0, HH_ERR_SHAPE, "hh_cast_f32_to_bf16: n < 0");
    HH_REQUIRE(HH_ALIGNED16(x) && HH_ALIGNED16(y), HH_ERR_ALIGN, "hh_cast_f32_to_bf16: pointers must be 16-byte aligned");
    if (n == 0) return HH_OK;
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
    return hh_check_launch("hh_cast_f32_to_bf16");
}

extern "C" int hh_cast_bf16_to_f32(const void* x, float* y, int64_t n, hh_stream_t stream) {
    HH_REQUIRE(n >= 0, HH_ERR_SHAPE, "hh_cast_bf16_to_f32: n < 0");
    HH_REQUIRE(HH_ALIGNED16(x) && HH_ALIGNED16(y), HH_ERR_ALIGN, "hh_cast_bf16_to_f32: pointers must be 16-byte aligned");
    if (n == 0) return HH_OK;
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, n);
    return hh_check_launch("hh_cast_bf16_to_f32");
}

extern "C" int hh_transpose_to_bf16(const void* x, int x_dtype, int64_t ldx, void* y, int64_t ldy, int64_t rows,
                                    int64_t cols, hh_stream_t stream) {
    HH_REQUIRE(rows >= 0 && cols >= 0 && ldx >= cols && ldy >= rows, HH_ERR_SHAPE, "hh_transpose_to_bf16: bad shape");
    HH_REQUIRE(x_dtype == HH_F32 || x_dtype == HH_BF16, HH_ERR_DTYPE, "hh_transpose_to_bf16: bad dtype");
    if (rows == 0 || cols == 0) return HH_OK;
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
    HH_REQUIRE(grid.y <= 65535u * 32u, HH_ERR_SHAPE, "hh_transpose_to_bf16: too many rows");
    if (x_dtype == HH_F32)
        hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (bf16_t*)y, ldy, rows, cols);
    else
        hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, cols);
    return hh_check_launch("hh_transpose_to_bf16");
}

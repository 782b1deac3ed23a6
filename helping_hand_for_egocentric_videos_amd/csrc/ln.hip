// LayerNorm forward/backward, one wave per row, row held in registers (two-pass mean / variance in fp32).
// Replaces nn.LayerNorm at model/LaviLa.py:439,456 (norm1/2/3 eps 1e-6, ln_pre 1e-5) and tfm_decoder.py:57.
// HBM-bound: algorithmic bytes per row = cols*(sizeof(in)+sizeof(out)).
#include "common.h"

static int ln_bwd_launch(const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd, const float* dy, float* dx,
                         float* dgamma, float* dbeta, int64_t rows, int cols, const float* dx_add, hipStream_t s);

template <int NV, typename TIN>
__device__ __forceinline__ void load_row(const TIN* x, int cols, int lane, float (&v)[NV][4]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < cols) {
            if constexpr (sizeof(TIN) == 4) {
                f32x4 t = *(const f32x4*)(x + c);
                v[i][0] = t[0]; v[i][1] = t[1]; v[i][2] = t[2]; v[i][3] = t[3];
            } else {
                u32x2 t = *(const u32x2*)(x + c);
                v[i][0] = bf16_lo_to_f32(t[0]); v[i][1] = bf16_hi_to_f32(t[0]);
                v[i][2] = bf16_lo_to_f32(t[1]); v[i][3] = bf16_hi_to_f32(t[1]);
            }
        } else {
            v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
        }
    }
}

template <int NV, typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TIN* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TOUT* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int64_t rows, int cols, float eps,
                                                     TOUT* __restrict__ y2, const float* __restrict__ pos, int pos_rows,
                                                     TOUT* __restrict__ y_cls = nullptr, int split_n = 0, const bf16_t* __restrict__ x_lo = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    // split_n > 0 (hh_layernorm_split_cls_fwd): rows are clips of split_n tokens whose first is the CLS token -- it goes to y_cls[clip],
    // the other split_n - 1 rows of a clip follow each other in y (the decoder's [B, T*n, D] grid, model/tfm_decoder.py:200-205)
    if (split_n > 0) {
        const int64_t clip = row / split_n, t = row % split_n;
        y = t == 0 ? y_cls + clip * cols - row * cols : y - (clip + 1) * (int64_t)cols;      // (both then indexed with row * cols below)
    }
    float v[NV][4];
    load_row<NV, TIN>(x + row * cols, cols, lane, v);
    if (x_lo != nullptr) {                         // bf16 pair stream (hh_layernorm_split_cls_fwd): the row is x + x_lo
        float w[NV][4];
        load_row<NV, bf16_t>(x_lo + row * cols, cols, lane, w);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] += w[i][j];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < cols) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (mean_out && lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < cols) {
            f32x4 g = *(const f32x4*)(gamma + c);
            f32x4 b = *(const f32x4*)(beta + c);
            float o0 = (v[i][0] - mean) * rstd * g[0] + b[0];
            float o1 = (v[i][1] - mean) * rstd * g[1] + b[1];
            float o2 = (v[i][2] - mean) * rstd * g[2] + b[2];
            float o3 = (v[i][3] - mean) * rstd * g[3] + b[3];
            if constexpr (sizeof(TOUT) == 4) {
                f32x4 o = {o0, o1, o2, o3};
                *(f32x4*)(y + row * cols + c) = o;
            } else {
                u32x2 o = {pack_bf16(o0, o1), pack_bf16(o2, o3)};
                *(u32x2*)(y + row * cols + c) = o;
            }
            if (y2 != nullptr) {                  // second output: LN(x) + pos[row % pos_rows]  (the key / query operand of an attention)
                const f32x4 pp = *(const f32x4*)(pos + (row % pos_rows) * cols + c);
                if constexpr (sizeof(TOUT) == 4) {
                    f32x4 o = {o0 + pp[0], o1 + pp[1], o2 + pp[2], o3 + pp[3]};
                    *(f32x4*)(y2 + row * cols + c) = o;
                } else {
                    u32x2 o = {pack_bf16(o0 + pp[0], o1 + pp[1]), pack_bf16(o2 + pp[2], o3 + pp[3])};
                    *(u32x2*)(y2 + row * cols + c) = o;
                }
            }
        }
    }
}

// Row statistics of a bf16 matrix for the LayerNorm fold of the GEMMs (include/hh.h, hh_gemm_epilogue.ln_stats): stats[row] = (rstd,
// -rstd * mean), two-pass in registers.  Rows [row0, rows).
template <int NV>
__global__ __launch_bounds__(256) void ln_rowstats_kernel(const bf16_t* __restrict__ z, int64_t ldz, float* __restrict__ stats, int64_t row0,
                                                          int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = row0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NV][4];
    load_row<NV, bf16_t>(z + row * ldz, cols, lane, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (lane == 0) { f32x2 o = {rstd, -rstd * mean}; *(f32x2*)(stats + 2 * row) = o; }
}

// rows [row0, rows) of z -> stats (library-internal: the producer side of the fold calls it for the rows whose statistics did not come
// out of the GEMM's own epilogue)
int hh_ln_rowstats_launch(const void* z, int64_t ldz, float* stats, int64_t row0, int64_t rows, int cols, float eps, hipStream_t s) {
    if (rows <= row0) return HH_OK;
    dim3 grid((unsigned)((rows - row0 + 3) / 4)), block(256);
    const int nv = (cols + 255) / 256;
#define LR(NV) hipLaunchKernelGGL((ln_rowstats_kernel<NV>), grid, block, 0, s, (const bf16_t*)z, ldz, stats, row0, rows, cols, eps)
    if (nv <= 2) LR(2); else if (nv <= 4) LR(4); else LR(8);
#undef LR
    return hh_check_launch("hh_ln_rowstats");
}

// Statistics of the producer side of the LayerNorm fold in ONE launch: rows [0, rows_part) have per-slice (sum, sum of squares) written by
// the persistent GEMM's epilogue (gemm256w4.hip, EPI 4) -- one thread per row, slices added in index order (deterministic); rows
// [rows_part, rows) (the GEMM's row tail) are read back from z itself, one wave per row, two-pass (the last blocks of the grid).
template <int NV>
__global__ __launch_bounds__(256) void ln_fold_stats_kernel(const float* __restrict__ partials, int slices, float* __restrict__ stats, int64_t rows_part,
                                                            const bf16_t* __restrict__ z, int64_t ldz, int64_t rows, int cols, float eps, int part_blocks) {
    if ((int)blockIdx.x < part_blocks) {
        const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (row >= rows_part) return;
        const f32x2* p = (const f32x2*)partials + row * slices;
        float s = 0.f, q = 0.f;
        if (slices == 8) {                          // (N = 1024: the tower's producers) all eight 8-byte slots requested at once -- as four 16-byte loads --
            f32x4 v[4];                             // and added in the same order as the loop below: one memory round trip instead of eight
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ((const f32x4*)p)[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) { s += v[i][0]; q += v[i][1]; s += v[i][2]; q += v[i][3]; }
        } else {
            for (int i = 0; i < slices; ++i) { const f32x2 v = p[i]; s += v[0]; q += v[1]; }
        }
        const float inv_cols = 1.f / (float)cols;
        const float mean = s * inv_cols;
        const float rstd = rsqrtf(fmaxf(q * inv_cols - mean * mean, 0.f) + eps);
        const f32x2 o = {rstd, -rstd * mean};
        *(f32x2*)(stats + 2 * row) = o;
        return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t row = rows_part + (int64_t)((int)blockIdx.x - part_blocks) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NV][4];
    load_row<NV, bf16_t>(z + row * ldz, cols, lane, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (lane == 0) { f32x2 o = {rstd, -rstd * mean}; *(f32x2*)(stats + 2 * row) = o; }
}
int hh_ln_fold_stats_launch(const float* partials, int slices, float* stats, int64_t rows_part, const void* z, int64_t ldz, int64_t rows, int cols, float eps,
                            hipStream_t s) {
    if (rows <= 0) return HH_OK;
    const int part_blocks = (int)((rows_part + 255) / 256);
    const unsigned grid = (unsigned)(part_blocks + (rows - rows_part + 3) / 4);
    const int nv = (cols + 255) / 256;
#define LF(NV) hipLaunchKernelGGL((ln_fold_stats_kernel<NV>), dim3(grid), dim3(256), 0, s, partials, slices, stats, rows_part, (const bf16_t*)z, ldz, rows, cols, eps, part_blocks)
    if (nv <= 2) LF(2); else if (nv <= 4) LF(4); else LF(8);
#undef LF
    return hh_check_launch("hh_gemm_bf16(LayerNorm statistics)");
}

extern "C" int hh_ln_rowstats(const void* z, int64_t ldz, float* stats, int64_t rows, int cols, float eps, hh_stream_t stream) {
    HH_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 2048 && ldz >= cols && ldz % 4 == 0, HH_ERR_SHAPE, "hh_ln_rowstats: cols=%d must be a multiple of 8 and <= 2048", cols);
    if (rows == 0) return HH_OK;                      // (an empty tensor's data pointer is null)
    HH_REQUIRE(z != nullptr && stats != nullptr && HH_ALIGNED16(z) && (((uintptr_t)stats) & 7) == 0, HH_ERR_ALIGN, "hh_ln_rowstats: bad pointers");
    return hh_ln_rowstats_launch(z, ldz, stats, 0, rows, cols, eps, (hipStream_t)stream);
}

// fused residual add + LayerNorm:  x (fp32, in place) += delta (bf16)  ;  y = LN(x)
// Keeps the fp32 residual read-modify-write out of the GEMM epilogues (where it is limited by per-CU memory throughput and
// cannot overlap the MFMA main loop) and does it here at streaming HBM rate; x is written back only when WRITE_X.
template <int NV, typename TOUT, bool WRITE_X>
__global__ __launch_bounds__(256) void add_ln_kernel(float* __restrict__ x, const bf16_t* __restrict__ delta,
                                                     const bf16_t* __restrict__ delta2,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     TOUT* __restrict__ y, int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NV][4], d[NV][4];
    load_row<NV, float>(x + row * cols, cols, lane, v);
    load_row<NV, bf16_t>(delta + row * cols, cols, lane, d);
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] += d[i][j];
    if (delta2 != nullptr) {                      // second pending branch, added AFTER the first: (x + d1) + d2
        load_row<NV, bf16_t>(delta2 + row * cols, cols, lane, d);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] += d[i][j];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        if (WRITE_X) {
            const int c = (i * 64 + lane) * 4;
            if (c < cols) { f32x4 o = {v[i][0], v[i][1], v[i][2], v[i][3]}; *(f32x4*)(x + row * cols + c) = o; }
        }
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float dd = v[i][j] - mean; q += dd * dd; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
            float o0 = (v[i][0] - mean) * rstd * g[0] + b[0], o1 = (v[i][1] - mean) * rstd * g[1] + b[1];
            float o2 = (v[i][2] - mean) * rstd * g[2] + b[2], o3 = (v[i][3] - mean) * rstd * g[3] + b[3];
            if constexpr (sizeof(TOUT) == 4) { f32x4 o = {o0, o1, o2, o3}; *(f32x4*)(y + row * cols + c) = o; }
            else { u32x2 o = {pack_bf16(o0, o1), pack_bf16(o2, o3)}; *(u32x2*)(y + row * cols + c) = o; }
        }
    }
}

// backward: dx = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma; dgamma += dy*xhat, dbeta += dy (atomics per block)
template <int NV, typename TIN>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TIN* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ dy, float* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int64_t rows, int cols, int rows_per_block, const float* dx_add) {
    __shared__ float red[2][4][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[NV][4], ab[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    for (int rr = wave; rr < rows_per_block; rr += 4) {
        const int64_t row = r0 + rr;
        if (row >= rows) break;
        float v[NV][4], d[NV][4];
        load_row<NV, TIN>(x + row * cols, cols, lane, v);
        load_row<NV, float>(dy + row * cols, cols, lane, d);
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = (i * 64 + lane) * 4;
            if (c < cols) {
                f32x4 g = *(const f32x4*)(gamma + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float xh = (v[i][j] - mu) * rs;
                    float gg = d[i][j] * g[j];
                    s1 += gg;
                    s2 += gg * xh;
                    ag[i][j] += d[i][j] * xh;
                    ab[i][j] += d[i][j];
                    v[i][j] = xh;
                    d[i][j] = gg;
                }
            }
        }
        s1 = wave_sum(s1) / (float)cols;
        s2 = wave_sum(s2) / (float)cols;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = (i * 64 + lane) * 4;
            if (c < cols) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rs * (d[i][j] - s1 - v[i][j] * s2);
                if (dx_add != nullptr) o += *(const f32x4*)(dx_add + row * cols + c);      // gradient of the residual path (may alias dx)
                *(f32x4*)(dx + row * cols + c) = o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[0][wave][(i * 64 + lane) * 4 + j] = ag[i][j];
            red[1][wave][(i * 64 + lane) * 4 + j] = ab[i][j];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        float a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        atomicAdd(dgamma + c, a);
        atomicAdd(dbeta + c, b);
    }
}

template <int NV>
static int ln_fwd_dispatch(const void* x, int xd, const float* g, const float* b, void* y, int yd, float* mo, float* ro,
                           int64_t rows, int cols, float eps, hipStream_t s, void* y2 = nullptr, const float* pos = nullptr, int pos_rows = 1,
                           void* y_cls = nullptr, int split_n = 0, const void* x_lo = nullptr) {
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (xd == HH_F32 && yd == HH_BF16)
        hipLaunchKernelGGL((ln_fwd_kernel<NV, float, bf16_t>), grid, block, 0, s, (const float*)x, g, b, (bf16_t*)y, mo, ro, rows, cols, eps, (bf16_t*)y2, pos, pos_rows, (bf16_t*)y_cls, split_n);
    else if (xd == HH_F32 && yd == HH_F32)
        hipLaunchKernelGGL((ln_fwd_kernel<NV, float, float>), grid, block, 0, s, (const float*)x, g, b, (float*)y, mo, ro, rows, cols, eps, (float*)y2, pos, pos_rows, (float*)y_cls, split_n);
    else if (xd == HH_BF16 && yd == HH_BF16)
        hipLaunchKernelGGL((ln_fwd_kernel<NV, bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)x, g, b, (bf16_t*)y, mo, ro, rows, cols, eps, (bf16_t*)y2, pos, pos_rows, (bf16_t*)y_cls, split_n, (const bf16_t*)x_lo);
    else
        hipLaunchKernelGGL((ln_fwd_kernel<NV, bf16_t, float>), grid, block, 0, s, (const bf16_t*)x, g, b, (float*)y, mo, ro, rows, cols, eps, (float*)y2, pos, pos_rows, (float*)y_cls, split_n, (const bf16_t*)x_lo);
    return hh_check_launch("hh_layernorm_fwd");
}

extern "C" int hh_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                                float* mean_out, float* rstd_out, int64_t rows, int cols, float eps, hh_stream_t stream) {
    HH_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 2048, HH_ERR_SHAPE, "hh_layernorm_fwd: cols=%d must be a multiple of 8 and <= 2048", cols);
    HH_REQUIRE((x_dtype == HH_F32 || x_dtype == HH_BF16) && (y_dtype == HH_F32 || y_dtype == HH_BF16), HH_ERR_DTYPE, "hh_layernorm_fwd: bad dtype");
    HH_REQUIRE(HH_ALIGNED16(x) && HH_ALIGNED16(y) && HH_ALIGNED16(gamma) && HH_ALIGNED16(beta), HH_ERR_ALIGN, "hh_layernorm_fwd: pointers must be 16-byte aligned");
    HH_REQUIRE((mean_out == nullptr) == (rstd_out == nullptr), HH_ERR_SHAPE, "hh_layernorm_fwd: mean_out/rstd_out must both be set or both NULL");
    if (rows == 0) return HH_OK;
    hipStream_t s = (hipStream_t)stream;
    int nv = (cols + 255) / 256;
    if (nv <= 2) return ln_fwd_dispatch<2>(x, x_dtype, gamma, beta, y, y_dtype, mean_out, rstd_out, rows, cols, eps, s);
    if (nv <= 4) return ln_fwd_dispatch<4>(x, x_dtype, gamma, beta, y, y_dtype, mean_out, rstd_out, rows, cols, eps, s);
    return ln_fwd_dispatch<8>(x, x_dtype, gamma, beta, y, y_dtype, mean_out, rstd_out, rows, cols, eps, s);
}

extern "C" int hh_layernorm_split_cls_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y_patches, void* y_cls, int y_dtype,
                                          int64_t clips, int tokens_per_clip, int cols, float eps, const void* x_lo, hh_stream_t stream) {
    HH_REQUIRE(x_lo == nullptr || (x_dtype == HH_BF16 && HH_ALIGNED16(x_lo)), HH_ERR_DTYPE, "hh_layernorm_split_cls_fwd: x_lo (bf16 pair stream) needs x_dtype = HH_BF16");
    HH_REQUIRE(clips >= 0 && tokens_per_clip >= 2 && cols > 0 && cols % 8 == 0 && cols <= 2048, HH_ERR_SHAPE,
               "hh_layernorm_split_cls_fwd: cols=%d must be a multiple of 8 and <= 2048, tokens_per_clip >= 2", cols);
    HH_REQUIRE((x_dtype == HH_F32 || x_dtype == HH_BF16) && (y_dtype == HH_F32 || y_dtype == HH_BF16), HH_ERR_DTYPE, "hh_layernorm_split_cls_fwd: bad dtype");
    if (clips == 0) return HH_OK;
    HH_REQUIRE(y_patches != nullptr && y_cls != nullptr && HH_ALIGNED16(x) && HH_ALIGNED16(y_patches) && HH_ALIGNED16(y_cls) && HH_ALIGNED16(gamma) && HH_ALIGNED16(beta), HH_ERR_ALIGN,
               "hh_layernorm_split_cls_fwd: pointers must be non-NULL and 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = clips * tokens_per_clip;
    const int nv = (cols + 255) / 256;
    if (nv <= 2) return ln_fwd_dispatch<2>(x, x_dtype, gamma, beta, y_patches, y_dtype, nullptr, nullptr, rows, cols, eps, s, nullptr, nullptr, 1, y_cls, tokens_per_clip, x_lo);
    if (nv <= 4) return ln_fwd_dispatch<4>(x, x_dtype, gamma, beta, y_patches, y_dtype, nullptr, nullptr, rows, cols, eps, s, nullptr, nullptr, 1, y_cls, tokens_per_clip, x_lo);
    return ln_fwd_dispatch<8>(x, x_dtype, gamma, beta, y_patches, y_dtype, nullptr, nullptr, rows, cols, eps, s, nullptr, nullptr, 1, y_cls, tokens_per_clip, x_lo);
}

extern "C" int hh_layernorm_pos_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, void* y_plus_pos, int y_dtype,
                                    const float* pos, int pos_rows, float* mean_out, float* rstd_out, int64_t rows, int cols, float eps,
                                    hh_stream_t stream) {
    HH_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 2048 && pos_rows > 0, HH_ERR_SHAPE, "hh_layernorm_pos_fwd: cols=%d must be a multiple of 8 and <= 2048, pos_rows > 0", cols);
    HH_REQUIRE((x_dtype == HH_F32 || x_dtype == HH_BF16) && (y_dtype == HH_F32 || y_dtype == HH_BF16), HH_ERR_DTYPE, "hh_layernorm_pos_fwd: bad dtype");
    if (rows == 0) return HH_OK;                      // (empty tensors have null data pointers)
    HH_REQUIRE(y_plus_pos != nullptr && pos != nullptr, HH_ERR_SHAPE, "hh_layernorm_pos_fwd: y_plus_pos / pos are NULL (use hh_layernorm_fwd)");
    HH_REQUIRE(HH_ALIGNED16(x) && HH_ALIGNED16(y) && HH_ALIGNED16(y_plus_pos) && HH_ALIGNED16(pos) && HH_ALIGNED16(gamma) && HH_ALIGNED16(beta), HH_ERR_ALIGN,
               "hh_layernorm_pos_fwd: pointers must be 16-byte aligned");
    HH_REQUIRE((mean_out == nullptr) == (rstd_out == nullptr), HH_ERR_SHAPE, "hh_layernorm_pos_fwd: mean_out/rstd_out must both be set or both NULL");
    if (rows == 0) return HH_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nv = (cols + 255) / 256;
    if (nv <= 2) return ln_fwd_dispatch<2>(x, x_dtype, gamma, beta, y, y_dtype, mean_out, rstd_out, rows, cols, eps, s, y_plus_pos, pos, pos_rows);
    if (nv <= 4) return ln_fwd_dispatch<4>(x, x_dtype, gamma, beta, y, y_dtype, mean_out, rstd_out, rows, cols, eps, s, y_plus_pos, pos, pos_rows);
    return ln_fwd_dispatch<8>(x, x_dtype, gamma, beta, y, y_dtype, mean_out, rstd_out, rows, cols, eps, s, y_plus_pos, pos, pos_rows);
}

extern "C" int hh_layernorm_bwd_add(const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd,
                                    const float* dy, const float* dx_add, float* dx, float* dgamma, float* dbeta, int64_t rows, int cols,
                                    hh_stream_t stream) {
    HH_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 1024, HH_ERR_SHAPE, "hh_layernorm_bwd_add: cols=%d must be a multiple of 8 and <= 1024", cols);
    HH_REQUIRE(x_dtype == HH_F32 || x_dtype == HH_BF16, HH_ERR_DTYPE, "hh_layernorm_bwd_add: bad dtype");
    HH_REQUIRE(HH_ALIGNED16(x) && HH_ALIGNED16(dy) && HH_ALIGNED16(dx) && HH_ALIGNED16(gamma) && HH_ALIGNED16(dx_add), HH_ERR_ALIGN, "hh_layernorm_bwd_add: pointers must be 16-byte aligned");
    if (rows == 0) return HH_OK;
    return ln_bwd_launch(x, x_dtype, gamma, mean, rstd, dy, dx, dgamma, dbeta, rows, cols, dx_add, (hipStream_t)stream);
}

extern "C" int hh_layernorm_bwd(const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd,
                                const float* dy, float* dx, float* dgamma, float* dbeta, int64_t rows, int cols,
                                hh_stream_t stream) {
    HH_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 1024, HH_ERR_SHAPE, "hh_layernorm_bwd: cols=%d must be a multiple of 8 and <= 1024", cols);
    HH_REQUIRE(x_dtype == HH_F32 || x_dtype == HH_BF16, HH_ERR_DTYPE, "hh_layernorm_bwd: bad dtype");
    HH_REQUIRE(HH_ALIGNED16(x) && HH_ALIGNED16(dy) && HH_ALIGNED16(dx) && HH_ALIGNED16(gamma), HH_ERR_ALIGN, "hh_layernorm_bwd: pointers must be 16-byte aligned");
    if (rows == 0) return HH_OK;
    return ln_bwd_launch(x, x_dtype, gamma, mean, rstd, dy, dx, dgamma, dbeta, rows, cols, nullptr, (hipStream_t)stream);
}

static int ln_bwd_launch(const void* x, int x_dtype, const float* gamma, const float* mean, const float* rstd, const float* dy, float* dx,
                         float* dgamma, float* dbeta, int64_t rows, int cols, const float* dx_add, hipStream_t s) {
    const int rpb = rows >= 16384 ? 64 : 8;              // few rows (decoder query side): more workgroups, fewer rows each
    dim3 grid((unsigned)((rows + rpb - 1) / rpb)), block(256);
    int nv = (cols + 255) / 256;
#define LAUNCH(NV, T) hipLaunchKernelGGL((ln_bwd_kernel<NV, T>), grid, block, 0, s, (const T*)x, gamma, mean, rstd, dy, dx, dgamma, dbeta, rows, cols, rpb, dx_add)
    if (x_dtype == HH_F32) { if (nv <= 2) LAUNCH(2, float); else LAUNCH(4, float); }
    else { if (nv <= 2) LAUNCH(2, bf16_t); else LAUNCH(4, bf16_t); }
#undef LAUNCH
    return hh_check_launch("hh_layernorm_bwd");
}

extern "C" int hh_add_layernorm_fwd(float* x, const void* delta, const void* delta2, int write_x, const float* gamma, const float* beta, void* y,
                                    int y_dtype, int64_t rows, int cols, float eps, hh_stream_t stream) {
    HH_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 2048, HH_ERR_SHAPE, "hh_add_layernorm_fwd: cols=%d must be a multiple of 8 and <= 2048", cols);
    HH_REQUIRE(y_dtype == HH_F32 || y_dtype == HH_BF16, HH_ERR_DTYPE, "hh_add_layernorm_fwd: bad dtype");
    if (rows == 0) return HH_OK;                      // (empty tensors have null data pointers)
    HH_REQUIRE(delta != nullptr, HH_ERR_SHAPE, "hh_add_layernorm_fwd: delta is NULL (use hh_layernorm_fwd)");
    HH_REQUIRE(HH_ALIGNED16(x) && HH_ALIGNED16(delta) && HH_ALIGNED16(y) && HH_ALIGNED16(gamma) && HH_ALIGNED16(beta), HH_ERR_ALIGN,
               "hh_add_layernorm_fwd: pointers must be 16-byte aligned");
    if (rows == 0) return HH_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const bf16_t* d = (const bf16_t*)delta;
    const int nv = (cols + 255) / 256;
    // algorithmic bytes per element: x read (4) [+ written (4)] + delta (2) [+ delta2 (2)] + y (2 or 4)
    HHProfScope prof(HH_PROF_ADD_LN, (double)rows * cols * (4 + (write_x ? 4 : 0) + 2 + (delta2 ? 2 : 0) + (y_dtype == HH_BF16 ? 2 : 4)), s);
#define LA(NV, T, W) do { hh_prof_note_kernel(HH_PROF_ADD_LN, "add_ln_kernel<" #NV ", " #T ", " #W ">"); \
                          hipLaunchKernelGGL((add_ln_kernel<NV, T, W>), grid, block, 0, s, x, d, (const bf16_t*)delta2, gamma, beta, (T*)y, rows, cols, eps); } while (0)
#define LB(NV) do { if (y_dtype == HH_BF16) { if (write_x) LA(NV, bf16_t, true); else LA(NV, bf16_t, false); } \
                    else { if (write_x) LA(NV, float, true); else LA(NV, float, false); } } while (0)
    if (nv <= 2) LB(2); else if (nv <= 4) LB(4); else LB(8);
#undef LB
#undef LA
    return hh_check_launch("hh_add_layernorm_fwd");
}

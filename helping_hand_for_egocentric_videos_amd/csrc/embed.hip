// TimeSformer front end: im2col for the 14x14/14 patch convolution and the fused
// "cls concat + positional/temporal embedding add + ln_pre" pass (model/LaviLa.py:218-223,540-559).
// HBM-bound; the patch GEMM itself runs on hh_gemm_bf16 (K padded to a multiple of 64).
#include "common.h"

// patches[(frame*n + py*G + px), k] = video[frame, c, py*P+i, px*P+j], k = c*P*P + i*P + j ; zero for k >= 3*P*P
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ video, bf16_t* __restrict__ patches,
                                                     int64_t frames, int H, int W, int P, int Kpad) {
    const int G = W / P, n = (H / P) * G, K = 3 * P * P;
    const int chunks = Kpad / 8;
    const int64_t total = frames * n * chunks;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ch = (int)(idx % chunks);
        const int64_t row = idx / chunks;
        const int64_t frame = row / n;
        const int pr = (int)(row % n), py = pr / G, px = pr % G;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = ch * 8 + q;
            if (k < K) {
                const int c = k / (P * P), rem = k % (P * P), i = rem / P, jj = rem % P;
                v[q] = video[((frame * 3 + c) * H + (py * P + i)) * (int64_t)W + (px * P + jj)];
            } else v[q] = 0.f;
        }
        u32x4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
        *(u32x4*)(patches + row * Kpad + ch * 8) = o;
    }
}

// The same gather through LDS (round 5): one workgroup per (frame, patch row py) loads the 3 x P image rows that row of patches is cut
// from -- whole W-float rows, coalesced 16-byte loads -- into LDS as bf16 and writes its W / P patch rows as full 16-byte chunks.  The
// direct kernel above reads P-float runs (56 B at P = 14) at a stride of one image row: 1.4 TB/s of 462 MB at the headline shape.
__global__ __launch_bounds__(256) void im2col_rows_kernel(const float* __restrict__ video, bf16_t* __restrict__ patches,
                                                          int H, int W, int P, int Kpad) {
    extern __shared__ __attribute__((aligned(16))) char im_smem[];
    bf16_t* tile = (bf16_t*)im_smem;                     // [3 * P][W]
    const int G = W / P, gy = H / P, K = 3 * P * P, chunks = Kpad / 8;
    const int64_t frame = blockIdx.x / gy;
    const int py = blockIdx.x % gy;
    const int w4 = W / 4;
    for (int idx = threadIdx.x; idx < 3 * P * w4; idx += 256) {
        const int r = idx / w4, x4 = idx % w4, c = r / P, i = r % P;
        const f32x4 v = *(const f32x4*)(video + ((frame * 3 + c) * H + (py * P + i)) * (int64_t)W + x4 * 4);
        const u32x2 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
        *(u32x2*)(tile + r * W + x4 * 4) = o;
    }
    __syncthreads();
    bf16_t* out = patches + (frame * (int64_t)(gy * G) + (int64_t)py * G) * Kpad;
    for (int idx = threadIdx.x; idx < G * chunks; idx += 256) {
        const int px = idx / chunks, ch = idx % chunks;
        bf16_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = ch * 8 + q;
            if (k < K) {
                const int r = k / P, jj = k % P;          // r = c * P + i
                v[q] = tile[r * W + px * P + jj];
            } else v[q] = (bf16_t)0.f;
        }
        *(bf16x8*)(out + (int64_t)px * Kpad + ch * 8) = (bf16x8){v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    }
}

// uint8 frames -> (u/255 - mean[c]) / std[c] -> bf16 patch rows (the host-side ToTensor + Normalize of
// data_loader/transforms.py:38-75 / run/train.py:442-445 fused into the im2col).  channels_last: frames are [F,H,W,3]
// (decoder output order), else [F,3,H,W].
__global__ __launch_bounds__(256) void im2col_u8_kernel(const unsigned char* __restrict__ video, bf16_t* __restrict__ patches,
                                                        int64_t frames, int H, int W, int P, int Kpad, int channels_last,
                                                        float m0, float m1, float m2, float s0, float s1, float s2) {
    const int G = W / P, n = (H / P) * G, K = 3 * P * P;
    const int chunks = Kpad / 8;
    const int64_t total = frames * n * chunks;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ch = (int)(idx % chunks);
        const int64_t row = idx / chunks;
        const int64_t frame = row / n;
        const int pr = (int)(row % n), py = pr / G, px = pr % G;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = ch * 8 + q;
            if (k < K) {
                const int c = k / (P * P), rem = k % (P * P), i = rem / P, jj = rem % P;
                const int64_t y = py * P + i, x = px * P + jj;
                const unsigned char u = channels_last ? video[((frame * H + y) * W + x) * 3 + c]
                                                      : video[((frame * 3 + c) * H + y) * (int64_t)W + x];
                v[q] = ((float)u / 255.f - mean[c]) / sd[c];
            } else v[q] = 0.f;
        }
        u32x4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
        *(u32x4*)(patches + row * Kpad + ch * 8) = o;
    }
}

// one wave per output token row; D <= NV*256 (guarded), D % 4 == 0
template <int NV>
__global__ __launch_bounds__(256) void embed_ln_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, const float* __restrict__ temporal,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ x, int B, int T, int n, int D, float eps,
                                                       bf16_t* __restrict__ z, float* __restrict__ zstats, float z_eps, bf16_t* __restrict__ z_lo) {
    const int lane = threadIdx.x & 63;
    const int64_t N = 1 + (int64_t)T * n;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * N) return;
    const int64_t b = row / N, t = row % N;
    float v[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (c >= D) {
        } else if (t == 0) {
            a = *(const f32x4*)(cls + c) + *(const f32x4*)(pos + c);
        } else {
            const int64_t f = (t - 1) / n, pp = (t - 1) % n;
            a = *(const f32x4*)(tok + ((b * T + f) * n + pp) * D + c) + *(const f32x4*)(pos + (1 + pp) * D + c) +
                *(const f32x4*)(temporal + f * D + c);
        }
        v[i][0] = a[0]; v[i][1] = a[1]; v[i][2] = a[2]; v[i][3] = a[3];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if ((i * 64 + lane) * 4 >= D) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) { float d = v[i][j] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c >= D) continue;
        f32x4 g = *(const f32x4*)(gamma + c), bb = *(const f32x4*)(beta + c), o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + bb[j];
        if (x != nullptr) *(f32x4*)(x + row * D + c) = o;
        if (z != nullptr) {                       // (kept for the second pass below: the bf16 rounding of the row the first block's fold multiplies)
            const u32x2 zz = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
            *(u32x2*)(z + row * D + c) = zz;
            if (z_lo != nullptr) {                // bf16 pair stream: x = z + z_lo
                const u32x2 ll = {pack_bf16(o[0] - bf16_lo_to_f32(zz[0]), o[1] - bf16_hi_to_f32(zz[0])), pack_bf16(o[2] - bf16_lo_to_f32(zz[1]), o[3] - bf16_hi_to_f32(zz[1]))};
                *(u32x2*)(z_lo + row * D + c) = ll;
            }
            v[i][0] = bf16_lo_to_f32(zz[0]); v[i][1] = bf16_hi_to_f32(zz[0]); v[i][2] = bf16_lo_to_f32(zz[1]); v[i][3] = bf16_hi_to_f32(zz[1]);
        }
    }
    if (z == nullptr) return;
    // z = bf16(x) and its row statistics (rstd, -rstd * mean) with the NEXT LayerNorm's eps: what block 0's folded norm3 consumes
    // (include/hh.h: hh_gemm_epilogue.ln_stats) -- two launches (cast + hh_ln_rowstats) and one read of x saved per tower pass
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if ((i * 64 + lane) * 4 < D) s2 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float m2 = wave_sum(s2) / (float)D;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if ((i * 64 + lane) * 4 >= D) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - m2; q2 += d * d; }
    }
    const float r2 = rsqrtf(wave_sum(q2) / (float)D + z_eps);
    if (lane == 0) { const f32x2 st = {r2, -r2 * m2}; *(f32x2*)(zstats + 2 * row) = st; }
}

extern "C" int hh_patch_im2col(const float* video, void* patches, int64_t frames, int H, int W, int P, int Kpad,
                               hh_stream_t stream) {
    HH_REQUIRE(frames >= 0 && H > 0 && W > 0 && P > 0 && H % P == 0 && W % P == 0, HH_ERR_SHAPE, "hh_patch_im2col: bad image/patch size");
    HH_REQUIRE(Kpad % 8 == 0 && Kpad >= 3 * P * P, HH_ERR_SHAPE, "hh_patch_im2col: Kpad=%d must be a multiple of 8 and >= 3*P*P", Kpad);
    HH_REQUIRE(HH_ALIGNED16(patches), HH_ERR_ALIGN, "hh_patch_im2col: output must be 16-byte aligned");
    if (frames == 0) return HH_OK;
    const size_t lds = (size_t)3 * P * W * 2;
    if (W % 4 == 0 && lds <= 64 * 1024 && HH_ALIGNED16(video) && frames * (H / P) < (1ll << 31)) {
        // whole image rows through LDS (coalesced both ways)
        hipLaunchKernelGGL(im2col_rows_kernel, dim3((unsigned)(frames * (H / P))), dim3(256), lds, (hipStream_t)stream, video, (bf16_t*)patches, H, W, P, Kpad);
        return hh_check_launch("hh_patch_im2col");
    }
    const int64_t total = frames * (H / P) * (W / P) * (Kpad / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, video, (bf16_t*)patches, frames, H, W, P, Kpad);
    return hh_check_launch("hh_patch_im2col");
}

extern "C" int hh_patch_im2col_u8(const uint8_t* video, void* patches, int64_t frames, int H, int W, int P, int Kpad,
                                  int channels_last, const float* mean3, const float* std3, hh_stream_t stream) {
    HH_REQUIRE(frames >= 0 && H > 0 && W > 0 && P > 0 && H % P == 0 && W % P == 0, HH_ERR_SHAPE, "hh_patch_im2col_u8: bad image/patch size");
    HH_REQUIRE(Kpad % 8 == 0 && Kpad >= 3 * P * P, HH_ERR_SHAPE, "hh_patch_im2col_u8: Kpad=%d must be a multiple of 8 and >= 3*P*P", Kpad);
    HH_REQUIRE(mean3 != nullptr && std3 != nullptr && std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, HH_ERR_SHAPE,
               "hh_patch_im2col_u8: mean/std (3 host floats each) required, std != 0");
    HH_REQUIRE(HH_ALIGNED16(patches), HH_ERR_ALIGN, "hh_patch_im2col_u8: output must be 16-byte aligned");
    if (frames == 0) return HH_OK;
    const int64_t total = frames * (H / P) * (W / P) * (Kpad / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(im2col_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, video, (bf16_t*)patches, frames, H, W, P,
                       Kpad, channels_last, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    return hh_check_launch("hh_patch_im2col_u8");
}

extern "C" int hh_embed_ln_pre(const float* tok, const float* cls, const float* pos, const float* temporal,
                               const float* gamma, const float* beta, float* x, int B, int T, int n, int D, float eps,
                               void* z_out, float* z_stats, float z_eps, void* z_lo, hh_stream_t stream) {
    HH_REQUIRE(z_lo == nullptr || (z_out != nullptr && HH_ALIGNED16(z_lo)), HH_ERR_SHAPE, "hh_embed_ln_pre: z_lo (bf16 pair stream) comes with z_out");
    HH_REQUIRE(x != nullptr || z_lo != nullptr, HH_ERR_SHAPE, "hh_embed_ln_pre: x may be NULL only with the bf16 pair outputs (z_out, z_lo)");
    HH_REQUIRE((z_out == nullptr) == (z_stats == nullptr) && (z_out == nullptr || (HH_ALIGNED16(z_out) && (((uintptr_t)z_stats) & 7) == 0 && z_eps > 0.f)), HH_ERR_SHAPE,
               "hh_embed_ln_pre: z_out and z_stats come together (16- / 8-byte aligned, z_eps > 0)");
    HH_REQUIRE(B >= 0 && T > 0 && n > 0 && D > 0 && D % 8 == 0 && D <= 2048, HH_ERR_SHAPE, "hh_embed_ln_pre: D=%d must be a multiple of 8, <= 2048", D);
    HH_REQUIRE(HH_ALIGNED16(tok) && HH_ALIGNED16(cls) && HH_ALIGNED16(pos) && HH_ALIGNED16(temporal) && HH_ALIGNED16(gamma) &&
               HH_ALIGNED16(beta) && (x == nullptr || HH_ALIGNED16(x)), HH_ERR_ALIGN, "hh_embed_ln_pre: pointers must be 16-byte aligned");
    if (B == 0) return HH_OK;
    const int64_t rows = (int64_t)B * (1 + (int64_t)T * n);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int nv = (D + 255) / 256;
    if (nv <= 1) hipLaunchKernelGGL(embed_ln_kernel<1>, grid, block, 0, s, tok, cls, pos, temporal, gamma, beta, x, B, T, n, D, eps, (bf16_t*)z_out, z_stats, z_eps, (bf16_t*)z_lo);
    else if (nv <= 2) hipLaunchKernelGGL(embed_ln_kernel<2>, grid, block, 0, s, tok, cls, pos, temporal, gamma, beta, x, B, T, n, D, eps, (bf16_t*)z_out, z_stats, z_eps, (bf16_t*)z_lo);
    else if (nv <= 4) hipLaunchKernelGGL(embed_ln_kernel<4>, grid, block, 0, s, tok, cls, pos, temporal, gamma, beta, x, B, T, n, D, eps, (bf16_t*)z_out, z_stats, z_eps, (bf16_t*)z_lo);
    else hipLaunchKernelGGL(embed_ln_kernel<8>, grid, block, 0, s, tok, cls, pos, temporal, gamma, beta, x, B, T, n, D, eps, (bf16_t*)z_out, z_stats, z_eps, (bf16_t*)z_lo);
    return hh_check_launch("hh_embed_ln_pre");
}

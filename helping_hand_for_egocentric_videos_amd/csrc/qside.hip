// Query side of the object-query decoder (model/tfm_decoder.py:430-461 forward_pre, :208-233 heads; run/train.py:124-125,187-189
// projections): 13 query rows per clip -- self-attention 13 x 13, q/out projections, FFN 512 -> 2048 -> 512, box / projection heads.
// < 1 % of the step's FLOPs, but TRAINABLE fp32 weights and gradients that the 1e-3 loss bound leaves no room to round to bf16
// (measured on the oracle: bf16 operands on these GEMMs double the error of hs, 2.6e-3 -> 5.7e-3, and push single loss terms to
// 6-7.5e-4 before the encoder's own bf16 noise is added).  So the GEMMs run on the bf16 matrix cores at fp32-grade accuracy:
//
//   hh_qgemm_f32x3   C = epilogue(prologue(A) . B) with fp32 operands split on the fly into bf16 hi + lo halves
//                    (x = hi + lo + O(2^-17 x)) and three MFMAs per product:  A.B ~= Ahi.Bhi + Alo.Bhi + Ahi.Blo   (the dropped
//                    lo.lo term is 2^-16 relative).  v_mfma_f32_16x16x32_bf16 x 3 is 5.3x the rate of the fp32 MFMA
//                    (v_mfma_f32_16x16x4_f32) at the same LDS operand bytes.  Three layouts = the three GEMMs of an nn.Linear:
//                      NT  C[m,n] = sum_k A[m,k] B[n,k]   forward      (A activations [M,K], B weight [N,K])
//                      NN  C[m,n] = sum_k A[m,k] B[k,n]   dgrad        (A = dY [M,K], B = weight [K,N])
//                      TN  C[m,n] = sum_k A[k,m] B[k,n]   wgrad        (A = dY [K,M], B = X [K,N]; + column sums of A = bias gradient)
//                    Prologue on A: scale, dropout mask (regenerated from the forward's counter-based hash).  Epilogue: bias, column
//                    scale, ReLU, dropout, ReLU-mask of a stored activation, fp32 residual.  64 x 64 x 64 tiles, 4 waves; the
//                    shapes are tiny (M = B*13 rows), so the kernel is built for launch economy, not for the MFMA roofline.
//   hh_qself_attn_fwd / _bwd   the 13 x 13 self-attention of nn.MultiheadAttention (tfm_decoder.py:433-436) per (clip, head), one
//                    wave each, fp32 VALU, attention-dropout by the same hash.
#include "common.h"

struct QGemm {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int M, N, K, mode;
    float a_scale;
    unsigned a_drop_thresh; float a_drop_scale; unsigned a_drop_seed; int a_drop_ld;
    const float* bias; float scale; int scale_ncols; int relu;
    unsigned drop_thresh; float drop_scale; unsigned drop_seed;
    const float* mask; int64_t ldmask; float mask_scale;
    const float* resid; int64_t ldr;
    float* colsum;
    int k_per_split;           // k-steps (of 64) per blockIdx.z slice; gridDim.z > 1: partial products are ADDED atomically into a zeroed C / colsum
    // head-batched form (round 5: the per-head maps into / out of memory space of the K/V-projection-free cross-attention): blockIdx.z
    // is the batch index (no split-K then) and offsets every operand by its stride
    int batch; int64_t sA, sB, sC, sBias, sColsum;
    const float* rowscale; int64_t ld_rs, sRs;     // NT / NN: bias[n] * rowscale[m * ld_rs];  TN: colsum[m] = sum_k A[k, m] * rowscale[k * ld_rs]
};

__device__ __forceinline__ bool q_keep(unsigned seed, unsigned idx, unsigned thresh) {
    unsigned h = seed ^ (idx * 0x9E3779B9u);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h >= thresh;
}

__device__ __forceinline__ void q_split(const f32x4& v, u32x2& hi, u32x2& lo) {
    bf16_t h[4];
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { h[e] = (bf16_t)v[e]; r[e] = v[e] - (float)h[e]; }
    hi = (u32x2){pack_bf16((float)h[0], (float)h[1]), pack_bf16((float)h[2], (float)h[3])};
    lo = (u32x2){pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3])};
}

#define QBK 64           // contraction elements per k-step
#define QROW 72          // bf16 per LDS tile row: 64 k + 8 pad (144 B: 16-B aligned fragments, rows spread over the banks)

// One operand tile (64 rows x 64 k) per k-step and thread = 16 fp32 values:
//   direct     (contraction index contiguous in memory): row tid >> 2, k = 16 (tid & 3) .. + 15        -> 4 x float4 along k
//   transposed (contraction index is the memory ROW):     k = 4 (tid >> 4) .. + 3, rows 4 (tid & 15) .. + 3 -> 4 x float4, one per k,
//              transposed in registers so that both flavours store 4 consecutive k per row (8-byte LDS writes, no 2-byte scatter)
// one 64 x 64 output tile: (bx, by, bz) = the block's coordinates inside ITS product's grid (tile column, tile row, split-K slice or batch
// index), gz = that grid's z extent -- blockIdx / gridDim for a plain launch, decoded from the flat block index by qgemm_group_kernel
template <int MODE>
__device__ __forceinline__ void qgemm_tile(QGemm& p, const int bx, const int by, const int bz, const int gz) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[2][64][QROW];        // [hi | lo][tile row][k]
    __shared__ __attribute__((aligned(16))) bf16_t sB[2][64][QROW];
    __shared__ float scs[64];
    constexpr bool A_T = MODE == 2, B_T = MODE != 0;                         // operand stored with the contraction index as its ROW
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = by * 64, n0 = bx * 64;
    const int nk_all = (p.K + QBK - 1) / QBK;
    if (p.batch > 1) {
        const int64_t z = bz;
        p.A += z * p.sA; p.B += z * p.sB; p.C += z * p.sC;
        if (p.bias) p.bias += z * p.sBias;
        if (p.colsum) p.colsum += z * p.sColsum;
        if (p.rowscale) p.rowscale += z * p.sRs;
    }
    const int kt0 = p.batch > 1 ? 0 : bz * p.k_per_split;
    const int nk = min(nk_all - kt0, p.k_per_split);                          // this slice's k-steps (split-K over bz)
    const bool split = p.batch <= 1 && gz > 1;
    const bool want_cs = A_T && p.colsum != nullptr && bx == 0;
    if (tid < 64) scs[tid] = 0.f;

    f32x4 ra0[4], rb0[4], ra1[4], rb1[4];           // two register stages: the global loads run TWO k-steps ahead of the MFMAs
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    auto prologue = [&](f32x4& v, int r, int c) {                            // (r, c) = row / first column of v in A's memory layout
        if (p.a_scale != 1.f) v *= p.a_scale;
        if (p.a_drop_thresh) {
            const unsigned idx = (unsigned)r * (unsigned)p.a_drop_ld + (unsigned)c;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = q_keep(p.a_drop_seed, idx + e, p.a_drop_thresh) ? v[e] * p.a_drop_scale : 0.f;
        }
    };
    auto load_tiles = [&](int kt, f32x4 (&ra)[4], f32x4 (&rb)[4]) {
        const int k0 = (kt0 + kt) * QBK;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (!A_T) {
                const int r = m0 + (tid >> 2), kk = k0 + 16 * (tid & 3) + 4 * u;
                const bool ok = r < p.M && kk < p.K;
                ra[u] = ok ? *(const f32x4*)(p.A + (int64_t)r * p.lda + kk) : z4;
                if (ok) prologue(ra[u], r, kk);
            } else {
                const int r = k0 + 4 * (tid >> 4) + u, c = m0 + 4 * (tid & 15);
                const bool ok = r < p.K && c < p.M;
                ra[u] = ok ? *(const f32x4*)(p.A + (int64_t)r * p.lda + c) : z4;
                if (ok) prologue(ra[u], r, c);
                if (want_cs) cs += (p.rowscale && ok) ? ra[u] * p.rowscale[(int64_t)r * p.ld_rs] : ra[u];
            }
            if constexpr (!B_T) {
                const int r = n0 + (tid >> 2), kk = k0 + 16 * (tid & 3) + 4 * u;
                rb[u] = (r < p.N && kk < p.K) ? *(const f32x4*)(p.B + (int64_t)r * p.ldb + kk) : z4;
            } else {
                const int r = k0 + 4 * (tid >> 4) + u, c = n0 + 4 * (tid & 15);
                rb[u] = (r < p.K && c < p.N) ? *(const f32x4*)(p.B + (int64_t)r * p.ldb + c) : z4;
            }
        }
    };
    auto store_op = [&](bf16_t (*dst)[64][QROW], const f32x4 (&v)[4], bool transposed) {
        if (!transposed) {
            const int i = tid >> 2, k = 16 * (tid & 3);
            u32x2 h[4], l[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q_split(v[u], h[u], l[u]);
            *(u32x4*)&dst[0][i][k] = (u32x4){h[0][0], h[0][1], h[1][0], h[1][1]};
            *(u32x4*)&dst[0][i][k + 8] = (u32x4){h[2][0], h[2][1], h[3][0], h[3][1]};
            *(u32x4*)&dst[1][i][k] = (u32x4){l[0][0], l[0][1], l[1][0], l[1][1]};
            *(u32x4*)&dst[1][i][k + 8] = (u32x4){l[2][0], l[2][1], l[3][0], l[3][1]};
        } else {
            const int k = 4 * (tid >> 4), i = 4 * (tid & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                    // row i + e of the tile: the 4 consecutive k this thread loaded
                const f32x4 t = {v[0][e], v[1][e], v[2][e], v[3][e]};
                u32x2 h, l;
                q_split(t, h, l);
                *(u32x2*)&dst[0][i + e][k] = h;
                *(u32x2*)&dst[1][i + e][k] = l;
            }
        }
    };

    f32x4 acc[4] = {z4, z4, z4, z4};
    const int fr = lane & 15, fk = 8 * (lane >> 4);
    auto mma = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 ah = *(const bf16x8*)&sA[0][16 * wave + fr][32 * ks + fk];
            const bf16x8 al = *(const bf16x8*)&sA[1][16 * wave + fr][32 * ks + fk];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 bh = *(const bf16x8*)&sB[0][16 * t + fr][32 * ks + fk];
                const bf16x8 bl = *(const bf16x8*)&sB[1][16 * t + fr][32 * ks + fk];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc[t], 0, 0, 0);
            }
        }
    };
    load_tiles(0, ra0, rb0);
    if (nk > 1) load_tiles(1, ra1, rb1);
    for (int kt = 0; kt < nk; kt += 2) {
        __syncthreads();                                                    // everyone is done reading the previous k-step's tiles
        store_op(sA, ra0, A_T);
        store_op(sB, rb0, B_T);
        __syncthreads();
        if (kt + 2 < nk) load_tiles(kt + 2, ra0, rb0);                      // lands two k-steps from now
        mma();
        if (kt + 1 >= nk) break;
        __syncthreads();
        store_op(sA, ra1, A_T);
        store_op(sB, rb1, B_T);
        __syncthreads();
        if (kt + 3 < nk) load_tiles(kt + 3, ra1, rb1);
        mma();
    }
    // epilogue: lane owns C[m][n .. n+3], m = tile row (lane & 15), n = 16 t + 4 (lane >> 4)
    const int m = m0 + 16 * wave + fr;
    if (m < p.M) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = n0 + 16 * t + 4 * (lane >> 4);
            if (n >= p.N) continue;
            f32x4 v = acc[t];
            if (p.bias) v += p.rowscale ? *(const f32x4*)(p.bias + n) * p.rowscale[(int64_t)m * p.ld_rs] : *(const f32x4*)(p.bias + n);
            if (n < p.scale_ncols) v *= p.scale;
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (p.drop_thresh) {
                const unsigned idx = (unsigned)m * (unsigned)p.N + (unsigned)n;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = q_keep(p.drop_seed, idx + e, p.drop_thresh) ? v[e] * p.drop_scale : 0.f;
            }
            if (p.mask) {
                const f32x4 mk = *(const f32x4*)(p.mask + (int64_t)m * p.ldmask + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] * p.mask_scale : 0.f;
            }
            if (p.resid) v += *(const f32x4*)(p.resid + (int64_t)m * p.ldr + n);
            if (split) {
#pragma unroll
                for (int e = 0; e < 4; ++e) unsafeAtomicAdd(p.C + (int64_t)m * p.ldc + n + e, v[e]);
            } else {
                *(f32x4*)(p.C + (int64_t)m * p.ldc + n) = v;
            }
        }
    }
    if (want_cs) {                                                          // bias gradient: colsum[m] = sum_k A_eff[k, m]
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&scs[4 * (tid & 15) + e], cs[e]);
        __syncthreads();
        if (tid < 64 && m0 + tid < p.M) {
            if (split) unsafeAtomicAdd(p.colsum + m0 + tid, scs[tid]); else p.colsum[m0 + tid] = scs[tid];
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void qgemm_kernel(QGemm p) {
    qgemm_tile<MODE>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z);
}

// Several independent products of ONE mode in one launch (hh_qgemm_f32x3_group): the nine weight gradients of a decoder layer, the two
// in-projection halves of its self-attention.  Flat grid; a block finds its product by the prefix sums of the products' grids.  Every tile
// is computed exactly as by the single launch (bit-identical results).
struct QGroup {
    QGemm p[HH_QGEMM_GROUP_MAX];
    int start[HH_QGEMM_GROUP_MAX + 1];       // first flat block index of product i; start[n] = grid size
    int gx[HH_QGEMM_GROUP_MAX], gy[HH_QGEMM_GROUP_MAX], gz[HH_QGEMM_GROUP_MAX];
    int n;
};

template <int MODE>
__global__ __launch_bounds__(256) void qgemm_group_kernel(QGroup g) {
    const int bid = blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && bid >= g.start[i + 1]) ++i;
    const int local = bid - g.start[i], gx = g.gx[i], gy = g.gy[i];
    QGemm p = g.p[i];
    qgemm_tile<MODE>(p, local % gx, (local / gx) % gy, local / (gx * gy), g.gz[i]);
}

static void q_drop_params(float p, unsigned* thresh, float* scale) {
    *thresh = 0u; *scale = 1.f;
    if (p <= 0.f) return;
    double t = (double)p * 4294967296.0;
    *thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    if (*thresh == 0u) *thresh = 1u;
    *scale = 1.f / (1.f - p);
}

// argument checks + kernel parameters + grid of one product (shared by the single and the grouped launch)
static int q_prepare(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, int mode,
                     const hh_qgemm_opts* o, QGemm& p, dim3& grid) {
    HH_REQUIRE(o != nullptr && M >= 0 && N > 0 && K > 0 && mode >= 0 && mode <= 2, HH_ERR_SHAPE, "hh_qgemm_f32x3: bad shape / mode (M=%d N=%d K=%d mode=%d)", M, N, K, mode);
    // contiguous dimension of every operand is read / written as float4
    const int a_contig = mode == 2 ? M : K, b_contig = mode == 0 ? K : N;
    HH_REQUIRE(a_contig % 4 == 0 && b_contig % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && lda >= a_contig && ldb >= b_contig && ldc >= N,
               HH_ERR_SHAPE, "hh_qgemm_f32x3: contiguous dimensions and leading dimensions must be multiples of 4 (M=%d N=%d K=%d mode=%d)", M, N, K, mode);
    HH_REQUIRE(HH_ALIGNED16(A) && HH_ALIGNED16(B) && HH_ALIGNED16(C) && HH_ALIGNED16(o->bias) && HH_ALIGNED16(o->resid) && HH_ALIGNED16(o->relu_mask),
               HH_ERR_ALIGN, "hh_qgemm_f32x3: pointers must be 16-byte aligned");
    HH_REQUIRE(o->colsum == nullptr || mode == 2, HH_ERR_UNSUPPORTED, "hh_qgemm_f32x3: colsum (bias gradient) is a by-product of the TN mode only");
    HH_REQUIRE(o->a_drop_p >= 0.f && o->a_drop_p < 1.f && o->drop_p >= 0.f && o->drop_p < 1.f, HH_ERR_SHAPE, "hh_qgemm_f32x3: dropout p must be in [0,1)");
    HH_REQUIRE((o->resid == nullptr || o->ldr % 4 == 0) && (o->relu_mask == nullptr || o->ldmask % 4 == 0), HH_ERR_SHAPE, "hh_qgemm_f32x3: ldr / ldmask must be multiples of 4");
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.mode = mode;
    p.a_scale = o->a_scale == 0.f ? 1.f : o->a_scale;
    q_drop_params(o->a_drop_p, &p.a_drop_thresh, &p.a_drop_scale);
    p.a_drop_seed = o->a_drop_seed; p.a_drop_ld = o->a_drop_ld;
    p.bias = o->bias; p.scale = o->scale == 0.f ? 1.f : o->scale; p.scale_ncols = o->scale == 0.f ? 0 : (o->scale_ncols > 0 ? o->scale_ncols : N);
    p.relu = o->relu;
    q_drop_params(o->drop_p, &p.drop_thresh, &p.drop_scale);
    p.drop_seed = o->drop_seed;
    p.mask = o->relu_mask; p.ldmask = o->ldmask; p.mask_scale = o->mask_scale == 0.f ? 1.f : o->mask_scale;
    p.resid = o->resid; p.ldr = o->ldr; p.colsum = o->colsum;
    p.batch = o->batch > 1 ? o->batch : 1;
    p.sA = o->stride_a; p.sB = o->stride_b; p.sC = o->stride_c; p.sBias = o->stride_bias; p.sColsum = o->stride_colsum;
    p.rowscale = o->rowscale; p.ld_rs = o->ld_rowscale; p.sRs = o->stride_rowscale;
    HH_REQUIRE(p.batch == 1 || (o->splitk <= 1 && p.batch <= 65535 && p.sA % 4 == 0 && p.sB % 4 == 0 && p.sC % 4 == 0 && p.sBias % 4 == 0 && !o->relu_mask && !o->resid),
               HH_ERR_UNSUPPORTED, "hh_qgemm_f32x3: the batched form takes no split-K / relu_mask / residual; strides must be multiples of 4");
    HH_REQUIRE(o->rowscale == nullptr || (mode == 2 ? o->colsum != nullptr : o->bias != nullptr), HH_ERR_UNSUPPORTED,
               "hh_qgemm_f32x3: rowscale weights the bias (NT / NN) or the column sums (TN): pass the one it applies to");
    const int nk_all = (K + QBK - 1) / QBK;
    int splits = o->splitk > 1 ? o->splitk : 1;
    if (splits > nk_all) splits = nk_all;
    HH_REQUIRE(splits == 1 || (!o->bias && !o->relu && o->drop_p == 0.f && !o->relu_mask && !o->resid && p.scale_ncols == 0), HH_ERR_UNSUPPORTED,
               "hh_qgemm_f32x3: split-K adds partial products atomically into a zeroed C: no epilogue options");
    p.k_per_split = (nk_all + splits - 1) / splits;
    splits = (nk_all + p.k_per_split - 1) / p.k_per_split;
    grid = dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)(p.batch > 1 ? p.batch : splits));
    return HH_OK;
}

extern "C" int hh_qgemm_f32x3(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
                              int mode, const hh_qgemm_opts* o, hh_stream_t stream) {
    QGemm p;
    dim3 grid;
    const int rc = q_prepare(A, lda, B, ldb, C, ldc, M, N, K, mode, o, p, grid);
    if (rc != HH_OK) return rc;
    if (M == 0) return HH_OK;
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(qgemm_kernel<0>, grid, dim3(256), 0, s, p);
    else if (mode == 1) hipLaunchKernelGGL(qgemm_kernel<1>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(qgemm_kernel<2>, grid, dim3(256), 0, s, p);
    return hh_check_launch("hh_qgemm_f32x3");
}

extern "C" int hh_qgemm_f32x3_group(const hh_qgemm_item* items, int n, hh_stream_t stream) {
    HH_REQUIRE(items != nullptr && n >= 1 && n <= HH_QGEMM_GROUP_MAX, HH_ERR_SHAPE, "hh_qgemm_f32x3_group: 1 .. %d products per launch (n=%d)", HH_QGEMM_GROUP_MAX, n);
    QGroup g;
    int64_t blocks = 0;
    g.n = 0;
    const int mode = items[0].mode;
    for (int i = 0; i < n; ++i) {
        const hh_qgemm_item& it = items[i];
        HH_REQUIRE(it.mode == mode, HH_ERR_UNSUPPORTED, "hh_qgemm_f32x3_group: all products of a group share one mode (product %d: %d, product 0: %d)", i, it.mode, mode);
        dim3 grid;
        const int rc = q_prepare(it.A, it.lda, it.B, it.ldb, it.C, it.ldc, it.M, it.N, it.K, it.mode, &it.opts, g.p[g.n], grid);
        if (rc != HH_OK) return rc;
        if (it.M == 0) continue;
        g.start[g.n] = (int)blocks;
        g.gx[g.n] = (int)grid.x; g.gy[g.n] = (int)grid.y; g.gz[g.n] = (int)grid.z;
        blocks += (int64_t)grid.x * grid.y * grid.z;
        HH_REQUIRE(blocks <= 0x7fffffff, HH_ERR_SHAPE, "hh_qgemm_f32x3_group: grid too large");
        ++g.n;
    }
    if (g.n == 0) return HH_OK;
    for (int i = g.n; i <= HH_QGEMM_GROUP_MAX; ++i) g.start[i] = (int)blocks;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks);
    if (mode == 0) hipLaunchKernelGGL(qgemm_group_kernel<0>, grid, dim3(256), 0, s, g);
    else if (mode == 1) hipLaunchKernelGGL(qgemm_group_kernel<1>, grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(qgemm_group_kernel<2>, grid, dim3(256), 0, s, g);
    return hh_check_launch("hh_qgemm_f32x3_group");
}

// ---------------------------------------------------------------------------------------------------------------------
// Self-attention over the Q <= 16 object queries of one clip (nn.MultiheadAttention inside forward_pre, tfm_decoder.py:433-436:
// q = k = norm1(tgt) + query_pos, v = norm1(tgt)).  qkv fp32 [B*Q, 3C] (q | k | v, head h at columns 64 h .. 64 h + 63 of each
// third, NOT pre-scaled), out fp32 [B*Q, C].  One wave per (clip, head): lane = (query i = lane & 15, group g = lane >> 4).
#define QS 65            // LDS row stride (floats) of a 16 x 64 head tile

__device__ __forceinline__ void qs_load_head(const float* src, int64_t ld, int Q, float (*dst)[QS], int lane) {
    // 16 rows x 64 floats: lane loads row (lane >> 2) columns 16 (lane & 3) .. +15, rows >= Q are zero
    const int r = lane >> 2, c = 16 * (lane & 3);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const f32x4 v = r < Q ? *(const f32x4*)(src + (int64_t)r * ld + c + 4 * e) : (f32x4){0.f, 0.f, 0.f, 0.f};
        dst[r][c + 4 * e] = v[0]; dst[r][c + 4 * e + 1] = v[1]; dst[r][c + 4 * e + 2] = v[2]; dst[r][c + 4 * e + 3] = v[3];
    }
}

// scores of query i against keys 4 g .. 4 g + 3, softmax over all 16 key slots (keys >= Q masked); returns p[4] and the row's
// statistics; with dropout, pd[4] = mask * p / (1 - p_drop)
__device__ __forceinline__ void qs_probs(const float (*q)[QS], const float (*k)[QS], int Q, int i, int g, float scale, unsigned bh,
                                         unsigned thresh, float dscale, unsigned seed, float (&p)[4], float (&pd)[4], bool (&keep)[4]) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < 64; ++d) {
        const float qv = q[i][d];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += qv * k[4 * g + e][d];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[e] = (4 * g + e < Q) ? s[e] * scale : -INFINITY; mx = fmaxf(mx, s[e]); }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { p[e] = (4 * g + e < Q) ? __expf(s[e] - mx) : 0.f; sum += p[e]; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        p[e] *= inv;
        keep[e] = thresh == 0u || q_keep(seed, (bh * 16u + (unsigned)i) * 16u + (unsigned)(4 * g + e), thresh);
        pd[e] = thresh == 0u ? p[e] : (keep[e] ? p[e] * dscale : 0.f);
    }
}

__global__ __launch_bounds__(256) void qself_attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int B, int Q, int heads,
                                                             unsigned thresh, float dscale, unsigned seed) {
    __shared__ float sq[4][16][QS], sk[4][16][QS], sv[4][16][QS], sp[4][16][17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + w;
    if (bh >= B * heads) return;
    const int b = bh / heads, h = bh % heads, C = heads * 64;
    const float* base = qkv + (int64_t)b * Q * 3 * C + h * 64;
    qs_load_head(base, 3 * C, Q, sq[w], lane);
    qs_load_head(base + C, 3 * C, Q, sk[w], lane);
    qs_load_head(base + 2 * C, 3 * C, Q, sv[w], lane);
    __builtin_amdgcn_wave_barrier();
    const int i = lane & 15, g = lane >> 4;
    float p[4], pd[4];
    bool keep[4];
    qs_probs(sq[w], sk[w], Q, i, g, 0.125f, (unsigned)bh, thresh, dscale, seed, p, pd, keep);
#pragma unroll
    for (int e = 0; e < 4; ++e) sp[w][i][4 * g + e] = pd[e];
    __builtin_amdgcn_wave_barrier();
    // O[i][16 g .. 16 g + 15] = sum_j P[i][j] V[j][:]
    float o[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) o[d] = 0.f;
    for (int j = 0; j < Q; ++j) {
        const float pj = sp[w][i][j];
#pragma unroll
        for (int d = 0; d < 16; ++d) o[d] += pj * sv[w][j][16 * g + d];
    }
    if (i < Q) {
        float* op = out + ((int64_t)b * Q + i) * C + h * 64 + 16 * g;
#pragma unroll
        for (int d = 0; d < 16; d += 4) *(f32x4*)(op + d) = (f32x4){o[d], o[d + 1], o[d + 2], o[d + 3]};
    }
}

// backward: dqkv [B*Q, 3C] from dout [B*Q, C]; probabilities (and the dropout mask) are recomputed from qkv
__global__ __launch_bounds__(128) void qself_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, float* __restrict__ dqkv,
                                                             int B, int Q, int heads, unsigned thresh, float dscale, unsigned seed) {
    __shared__ float sq[2][16][QS], sk[2][16][QS], sv[2][16][QS], sdo[2][16][QS], sp[2][16][17], sds[2][16][17];      // two waves per workgroup (LDS)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int bh = blockIdx.x * 2 + w;
    if (bh >= B * heads) return;
    const int b = bh / heads, h = bh % heads, C = heads * 64;
    const float* base = qkv + (int64_t)b * Q * 3 * C + h * 64;
    qs_load_head(base, 3 * C, Q, sq[w], lane);
    qs_load_head(base + C, 3 * C, Q, sk[w], lane);
    qs_load_head(base + 2 * C, 3 * C, Q, sv[w], lane);
    qs_load_head(dout + (int64_t)b * Q * C + h * 64, C, Q, sdo[w], lane);
    __builtin_amdgcn_wave_barrier();
    const int i = lane & 15, g = lane >> 4;
    float p[4], pd[4];
    bool keep[4];
    qs_probs(sq[w], sk[w], Q, i, g, 0.125f, (unsigned)bh, thresh, dscale, seed, p, pd, keep);
    // dPd[i][j] = sum_d dO[i][d] V[j][d];  dP = dPd * mask / (1 - p_drop);  dS = P (dP - sum_j dP P)
    float dp[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < 64; ++d) {
        const float dv = sdo[w][i][d];
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[e] += dv * sv[w][4 * g + e][d];
    }
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (thresh) dp[e] = keep[e] ? dp[e] * dscale : 0.f;                              // same mask as the forward
        dot += dp[e] * p[e];
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sp[w][i][4 * g + e] = pd[e];
        sds[w][i][4 * g + e] = (4 * g + e < Q && i < Q) ? p[e] * (dp[e] - dot) * 0.125f : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    // lane (row i, d-slice 16 g ..): dQ[i] = sum_j dS[i][j] K[j];  dK[i] = sum_q dS[q][i] Q[q];  dV[i] = sum_q Pd[q][i] dO[q]
    float dq[16], dk[16], dvv[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) { dq[d] = 0.f; dk[d] = 0.f; dvv[d] = 0.f; }
    for (int j = 0; j < Q; ++j) {
        const float ds_ij = sds[w][i][j], ds_ji = sds[w][j][i], p_ji = sp[w][j][i];
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            dq[d] += ds_ij * sk[w][j][16 * g + d];
            dk[d] += ds_ji * sq[w][j][16 * g + d];
            dvv[d] += p_ji * sdo[w][j][16 * g + d];
        }
    }
    if (i < Q) {
        float* o = dqkv + ((int64_t)b * Q + i) * 3 * C + h * 64 + 16 * g;
#pragma unroll
        for (int d = 0; d < 16; d += 4) {
            *(f32x4*)(o + d) = (f32x4){dq[d], dq[d + 1], dq[d + 2], dq[d + 3]};
            *(f32x4*)(o + C + d) = (f32x4){dk[d], dk[d + 1], dk[d + 2], dk[d + 3]};
            *(f32x4*)(o + 2 * C + d) = (f32x4){dvv[d], dvv[d + 1], dvv[d + 2], dvv[d + 3]};
        }
    }
}

extern "C" int hh_qself_attn_fwd(const float* qkv, float* out, int B, int Q, int heads, float dropout_p, uint32_t seed, hh_stream_t stream) {
    HH_REQUIRE(B >= 0 && Q > 0 && Q <= 16 && heads > 0, HH_ERR_SHAPE, "hh_qself_attn_fwd: need 0 < Q <= 16 (Q=%d)", Q);
    HH_REQUIRE(HH_ALIGNED16(qkv) && HH_ALIGNED16(out), HH_ERR_ALIGN, "hh_qself_attn_fwd: pointers must be 16-byte aligned");
    HH_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, HH_ERR_SHAPE, "hh_qself_attn_fwd: dropout_p must be in [0,1)");
    if (B == 0) return HH_OK;
    unsigned thr; float sc;
    q_drop_params(dropout_p, &thr, &sc);
    hipLaunchKernelGGL(qself_attn_fwd_kernel, dim3((unsigned)((B * heads + 3) / 4)), dim3(256), 0, (hipStream_t)stream, qkv, out, B, Q, heads, thr, sc, seed);
    return hh_check_launch("hh_qself_attn_fwd");
}

extern "C" int hh_qself_attn_bwd(const float* qkv, const float* dout, float* dqkv, int B, int Q, int heads, float dropout_p, uint32_t seed,
                                 hh_stream_t stream) {
    HH_REQUIRE(B >= 0 && Q > 0 && Q <= 16 && heads > 0, HH_ERR_SHAPE, "hh_qself_attn_bwd: need 0 < Q <= 16 (Q=%d)", Q);
    HH_REQUIRE(HH_ALIGNED16(qkv) && HH_ALIGNED16(dout) && HH_ALIGNED16(dqkv), HH_ERR_ALIGN, "hh_qself_attn_bwd: pointers must be 16-byte aligned");
    HH_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, HH_ERR_SHAPE, "hh_qself_attn_bwd: dropout_p must be in [0,1)");
    if (B == 0) return HH_OK;
    unsigned thr; float sc;
    q_drop_params(dropout_p, &thr, &sc);
    hipLaunchKernelGGL(qself_attn_bwd_kernel, dim3((unsigned)((B * heads + 1) / 2)), dim3(128), 0, (hipStream_t)stream, qkv, dout, dqkv, B, Q, heads, thr, sc, seed);
    return hh_check_launch("hh_qself_attn_bwd");
}

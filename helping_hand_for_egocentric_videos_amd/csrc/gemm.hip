// bf16 MFMA GEMM with fused epilogue:  C[m,n] = epi( sum_k A[m,k] * W[n,k] ).
// Replaces nn.Linear on the hot path (model/LaviLa.py:249 qkv, :281 proj, :186-189 fc1/fc2; tfm_decoder.py:156,
// nn.MultiheadAttention in-proj :438-441).  MFMA-bound: 2*M*N*K flops per call.
//
// Structure (v1): 128x128 block tile, BK=64, 256 threads = 4 waves (2x2), each wave 64x64 as 4x4 MFMA 16x16x32
// tiles.  Operands are swapped (MFMA A-operand = W rows, B-operand = A rows) so the accumulator holds C^T tiles:
// a lane owns 4 consecutive n for one m, which makes the epilogue's bias/residual/stores 8- or 16-byte vectors.
// Both operands are K-contiguous, staged HBM->LDS by LDS-DMA (global_load_lds_dwordx4), double buffered.
// LDS image: [row][8 chunks of 16 B]; chunk c of row r lives at position c ^ (r & 7) (XOR swizzle applied on the
// per-lane SOURCE address, LDS destination stays lane-linear) -> ds_read_b128 fragment reads are conflict-free.
// Block order is XCD-aware: blockIdx % 8 selects the XCD-local stream; each XCD walks groups of 8 m-tiles x all
// n-tiles so an A tile is re-used from that XCD's L2 across its n-tiles.
#include "gemm_common.h"
#include <stdlib.h>

#define BM 128
#define BN 128
#define BK 64
#define GROUP_M 8

// STAGES = 2: double buffer, 2 blocks/CU (large grids).  STAGES = 4: 4-deep LDS-DMA ring with counted vmcnt, three k-tiles in
// flight -- for small grids (row remainders, split-K slices, tiny shapes) whose k-loop is otherwise latency-bound.
template <bool OUT_BF16, int STAGES>
__global__ __launch_bounds__(256, STAGES == 2 ? 2 : 1) void gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [STAGES buffers][A 16 KB | W 16 KB]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- tile assignment: XCD-aware groups for tall problems, direct mapping (n-tiles spread over all XCDs) when there
    // are fewer m-tiles than XCDs (row remainders, skinny / split-K GEMMs)
    const int b = blockIdx.x;
    int mt, nt;
    if (p.Mt < 8) {
        mt = b % p.Mt;
        nt = b / p.Mt;
        if (nt >= p.Nt) return;
    } else {
        const int xcd = b & 7, j = b >> 3;
        const int per = GROUP_M * p.Nt;
        const int kg = j / per, r = j % per;
        nt = r / GROUP_M;
        const int mi = r % GROUP_M;
        mt = xcd + 8 * (kg * GROUP_M + mi);
        if (mt >= p.Mt) return;
    }
    const int64_t m0 = p.m_start + (int64_t)mt * BM;
    const int n0 = nt * BN;

    // ---- staging: wave w issues 4 LDS-DMA pieces per operand per k-tile; piece i covers tile rows (w*4+i)*8..+8
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        int64_t gm = m0 + row;
        if (gm > p.M - 1) gm = p.M - 1;
        a_src[i] = p.A + gm * p.lda + c * 8;
        w_src[i] = p.W + (int64_t)(n0 + row) * p.ldw + c * 8;
    }
    const int wm = wave >> 1, wn = wave & 1;
    // fragment read offsets (bytes inside an operand tile): row = base + (lane&15), chunk = ks*4 + (lane>>4)
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[2], w_off[2];     // per ks, for tile 0; tile t adds t*16 rows = t*2048 bytes
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + fq;
        a_off[ks] = (wm * 64 + frow) * 128 + ((c ^ (frow & 7)) << 4);
        w_off[ks] = (wn * 64 + frow) * 128 + ((c ^ (frow & 7)) << 4);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk_all = p.K / BK;
    int kt0 = 0, nk = nk_all;
    char* Cbase = (char*)p.C;
    if (p.e.splitk > 1) {
        const int per_split = (nk_all + p.e.splitk - 1) / p.e.splitk;
        kt0 = blockIdx.y * per_split;
        nk = nk_all - kt0 < per_split ? nk_all - kt0 : per_split;
        if (nk < 0) nk = 0;
        Cbase += (int64_t)blockIdx.y * p.e.split_stride * (OUT_BF16 ? 2 : 4);
    }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * 32768 + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(a_src[i] + (int64_t)(kt0 + kt) * BK, base + i * 1024);
            glds16(w_src[i] + (int64_t)(kt0 + kt) * BK, base + 16384 + i * 1024);
        }
    };
    if constexpr (STAGES == 2) {
        if (nk > 0) stage(0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < STAGES - 1; ++i)
            if (i < nk) stage(i, i);
    }
    for (int kt = 0; kt < nk; ++kt) {
        if constexpr (STAGES == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // tiles kt+1 .. min(kt+STAGES-2, nk-1) may stay in flight (8 LDS-DMA instructions per tile and wave)
            const int ahead = (nk - 1 - kt) < (STAGES - 2) ? (nk - 1 - kt) : (STAGES - 2);
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (STAGES == 2) {
            __syncthreads();
            if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        } else {
            asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
            if (kt + STAGES - 1 < nk) stage(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
        }
        const char* As = smem + (kt % STAGES) * 32768;
        const char* Ws = As + 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *(const bf16x8*)(As + a_off[ks] + t * 2048);
                wf[t] = *(const bf16x8*)(Ws + w_off[ks] + t * 2048);
            }
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tn], af[tm], acc[tn][tm], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns C[m][n..n+3], m = m0 + wm*64 + tm*16 + (lane&15), n = n0 + wn*64 + tn*16 + 4*(lane>>4)
    const hh_gemm_epilogue& e = p.e;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const int64_t m = m0 + wm * 64 + tm * 16 + frow;
        if (m >= p.M) continue;
        int64_t orow = m;
        if (e.remap_group > 0) orow = m + (m / e.remap_group) * e.remap_skip + e.remap_offset;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int n = n0 + wn * 64 + tn * 16 + 4 * fq;
            gemm_store4<OUT_BF16>(e, Cbase, p.ldc, orow, n, acc[tn][tm]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-tail kernel: the <= 64 rows that do not fill a 256-row tile of the persistent kernel (M = B*4097 always leaves B mod 256
// CLS-ish rows; 168 such launches per step).  The 128x128 kernel walks K serially (16-64 barrier-separated k-tiles: 16-64 us of
// a nearly idle chip per launch); here a workgroup owns 32 or 64 rows x 32 columns and its 4 or 8 waves SPLIT K, each streaming
// its operands from global memory directly in MFMA layout (v_mfma_f32_32x32x16_bf16: lane = row / column, 8 consecutive k per
// lane = one 16-B load; no LDS staging, no barrier in the k loop), then the partial accumulators meet in LDS and wave w
// finishes output tile w with the common epilogue.
template <bool OUT_BF16, int NWT, int MT>
__global__ __launch_bounds__(64 * NWT) void gemm_tail_kernel(GemmParams p) {
    // workgroup = 32 MT rows x 32 columns; its NWT waves split K.  [wave][m-tile][reg][lane] partials meet in LDS (<= 64 KiB)
    __shared__ float red[NWT * MT * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int kslice = p.K / NWT;                                 // multiple of 64 (checked by the launcher)
    const int k_begin = wave * kslice;
    const bf16_t* ap[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int64_t gm = p.m_start + i * 32 + l31;
        if (gm > p.M - 1) gm = p.M - 1;
        ap[i] = p.A + gm * p.lda + k_begin + 8 * h;
    }
    const bf16_t* wp = p.W + (int64_t)(n0 + l31) * p.ldw + k_begin + 8 * h;
    f32x16 acc[MT];                                               // C^T tiles (rows = n, columns = m)
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int k = 0; k < kslice; k += 64) {                        // 4 k-steps of 16 per iteration: 4 (MT + 1) independent 16-B loads in flight
        bf16x8 af[4][MT], wf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            wf[s] = *(const bf16x8*)(wp + k + 16 * s);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[s][i] = *(const bf16x8*)(ap[i] + k + 16 * s);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], af[s][mt], acc[mt], 0, 0, 0);
    }
    // ---- cross-wave reduction: wave w < MT finishes m-tile w
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * MT + mt) * 16 + r) * 64 + lane] = acc[mt][r];
    __syncthreads();
    if (wave >= MT) return;
    const int mt = wave;
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NWT; ++w) v += red[((w * MT + mt) * 16 + r) * 64 + lane];
        t[r] = v;
    }
    const int64_t m = p.m_start + mt * 32 + l31;
    if (m >= p.M) return;
    int64_t orow = m;
    const hh_gemm_epilogue& e = p.e;
    if (e.remap_group > 0) orow = m + (m / e.remap_group) * e.remap_skip + e.remap_offset;
    char* Cbase = (char*)p.C;
#pragma unroll
    for (int g = 0; g < 4; ++g) {                                 // accumulator register r: n = 8 (r >> 2) + 4 h + (r & 3)
        const f32x4 v = {t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
        gemm_store4<OUT_BF16>(e, Cbase, p.ldc, orow, n0 + 8 * g + 4 * h, v);
    }
}

int hh_tuning_gemm_tail();                  // gemm256.hip: hh_set_tuning("gemm_tail", 0) routes row tails back to the 128x128 kernel (A/B measurements)

int hh_ln_fold_stats_launch(const float* partials, int slices, float* stats, int64_t rows_part, const void* z, int64_t ldz, int64_t rows, int cols, float eps,
                            hipStream_t s);        // ln.hip
static int gemm_bf16_impl(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                          int64_t M, int N, int K, const hh_gemm_epilogue* epi, hh_stream_t stream, int64_t* stats_done_rows);

extern "C" int hh_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                            int64_t M, int N, int K, const hh_gemm_epilogue* epi, hh_stream_t stream) {
    int64_t stats_done = 0;                 // rows [0, stats_done) got per-slice row sums of z inside the GEMM kernel
    int rc = gemm_bf16_impl(A, lda, W, ldw, C, ldc, M, N, K, epi, stream, &stats_done);
    if (rc != HH_OK || epi->z_out == nullptr || M == 0) return rc;
    // producer side of the LayerNorm fold: rows [0, stats_done) have per-slice (sum, sum of squares) in z_partials (persistent kernel) ->
    // (rstd, -rstd * mean); the other rows (row tails, GEMMs on the generic kernels) get theirs from z itself -- one launch for both
    HH_REQUIRE(stats_done == 0 || epi->z_partials != nullptr, HH_ERR_SHAPE, "hh_gemm_bf16: z_partials workspace is NULL (hh_workspace_bytes_gemm_zstats)");
    return hh_ln_fold_stats_launch(epi->z_partials, N / 128, epi->z_stats, stats_done, epi->z_out, epi->z_ldc, M, N, epi->z_eps, (hipStream_t)stream);
}

static int gemm_bf16_impl(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                          int64_t M, int N, int K, const hh_gemm_epilogue* epi, hh_stream_t stream, int64_t* stats_done_rows) {
    HH_REQUIRE(epi != nullptr, HH_ERR_SHAPE, "hh_gemm_bf16: epilogue descriptor is NULL");
    HH_REQUIRE(M >= 0 && N > 0 && K > 0 && N % BN == 0 && K % BK == 0, HH_ERR_SHAPE,
               "hh_gemm_bf16: need N %% 128 == 0 and K %% 64 == 0 (M=%lld N=%d K=%d)", (long long)M, N, K);
    const bool cblk = epi->c_block_stride != 0;      // column-blocked C: N / 64 planes [rows, 64] (ldc is 64 by definition)
    HH_REQUIRE(lda >= K && ldw >= K && (cblk || ldc >= N) && lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, HH_ERR_SHAPE,
               "hh_gemm_bf16: bad leading dimensions lda=%lld ldw=%lld ldc=%lld", (long long)lda, (long long)ldw, (long long)ldc);
    HH_REQUIRE(!cblk || (epi->c_block_stride >= 64 * M && epi->c_block_stride % 8 == 0 && epi->splitk <= 1 && epi->remap_group == 0),
               HH_ERR_SHAPE, "hh_gemm_bf16: c_block_stride must be >= 64 * M and a multiple of 8 (no split-K / row remap with it)");
    HH_REQUIRE(HH_ALIGNED16(A) && HH_ALIGNED16(W) && HH_ALIGNED16(C) && HH_ALIGNED16(epi->bias) && HH_ALIGNED16(epi->resid),
               HH_ERR_ALIGN, "hh_gemm_bf16: pointers must be 16-byte aligned");
    HH_REQUIRE(epi->c_dtype == HH_F32 || epi->c_dtype == HH_BF16, HH_ERR_DTYPE, "hh_gemm_bf16: bad output dtype");
    HH_REQUIRE(epi->resid == nullptr || epi->ldr % 4 == 0, HH_ERR_SHAPE, "hh_gemm_bf16: ldr must be a multiple of 4");
    if (gemm_ln_ext(*epi)) {                 // LayerNorm fold (include/hh.h)
        HH_REQUIRE(epi->resid == nullptr && epi->remap_group == 0 && epi->splitk <= 1, HH_ERR_UNSUPPORTED,
                   "hh_gemm_bf16: the LayerNorm fold takes no fp32 residual / row remap / split-K");
        if (epi->ln_stats)
            HH_REQUIRE(epi->ln_colsum != nullptr && epi->bias != nullptr && HH_ALIGNED16(epi->ln_colsum) && (((uintptr_t)epi->ln_stats) & 7) == 0, HH_ERR_SHAPE,
                       "hh_gemm_bf16: ln_stats needs ln_colsum and bias (16-byte aligned; ln_stats 8-byte aligned)");
        HH_REQUIRE(epi->z_out != nullptr || (epi->z_update == 0 && epi->skip_c == 0), HH_ERR_UNSUPPORTED, "hh_gemm_bf16: z_update / skip_c need z_out");
        if (epi->z_out) {
            HH_REQUIRE(epi->z_resid != nullptr && epi->z_stats != nullptr && epi->act == HH_ACT_NONE && epi->colscale_cols == 0 && !cblk &&
                       epi->ln_stats == nullptr && epi->c_dtype == HH_BF16, HH_ERR_UNSUPPORTED,
                       "hh_gemm_bf16: z_out needs z_resid and z_stats, a bf16 row-major C, and no activation / column scale / ln_stats");
            if (epi->z_resid_lo != nullptr)
                HH_REQUIRE(epi->z_resid_dtype == HH_BF16 && epi->z_update != 0 && epi->z_out == epi->z_resid && epi->z_ldr == epi->z_ldc && epi->skip_c != 0 &&
                           HH_ALIGNED16(epi->z_resid_lo), HH_ERR_UNSUPPORTED,
                           "hh_gemm_bf16: z_resid_lo (bf16 pair stream) needs z_resid_dtype = HH_BF16, z_update, skip_c, z_out == z_resid and z_ldr == z_ldc");
            else
            HH_REQUIRE((epi->z_resid_dtype == HH_F32 || (epi->z_resid_dtype == HH_BF16 && epi->z_update == 0 && epi->z_ldr % 8 == 0)), HH_ERR_UNSUPPORTED,
                       "hh_gemm_bf16: z_resid_dtype must be HH_F32, or HH_BF16 without z_update (z_ldr %% 8 == 0)");
            HH_REQUIRE(epi->z_ldr >= N && epi->z_ldc >= N && epi->z_ldr % 4 == 0 && epi->z_ldc % 8 == 0 && N <= 2048 && HH_ALIGNED16(epi->z_resid) &&
                       HH_ALIGNED16(epi->z_out) && (((uintptr_t)epi->z_stats) & 7) == 0 && epi->z_eps > 0.f, HH_ERR_SHAPE,
                       "hh_gemm_bf16: bad z_resid / z_out / z_stats (N <= 2048, z_ldr %% 4 == 0, z_ldc %% 8 == 0, eps > 0)");
        }
    }
    if (M == 0) return HH_OK;
    GemmParams p;
    p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.C = C; p.ldc = cblk ? 64 : ldc;
    p.M = M; p.N = N; p.K = K; p.e = *epi;
    p.m_start = 0;
    p.skew_iters = 0;
    p.skew_phases = 0;
    p.tile_rows = 256;
    p.dynamic = 0;
    p.tile_slot = 0;
    p.group_m = 4;
    p.debug_nostore = 0;
    p.debug_ts = 0;
    p.tail_m = 0;
    p.tail_rows = 0;
    if (hh_gemm256_eligible(p)) {
        // full 256-row tiles on the 8-phase kernel; the (< 256)-row remainder on the 128x128 kernel so that it does not
        // cost a whole extra round of 256x256 blocks (M = B*4097 is never a multiple of 256)
        GemmParams pm = p;
        pm.tile_rows = hh_gemm256_tile_rows(p, (hipStream_t)stream);
        pm.M = (M / pm.tile_rows) * pm.tile_rows;
        // the <= 64 rows behind the last full tile (M = B * 4097: the B mod 256 CLS-ish rows) ride inside the persistent kernel: its
        // first N / 32 (x 2 beyond 32 rows) workgroups each finish one 32 x 32 piece (8 waves split K) before their tile walk -- the separate row-tail launch cost
        // ~11 us per GEMM, 144 times per step (skipping the tails altogether, a timing experiment, gave +1.3 % step throughput)
        const bool fold = hh_tuning_gemm_tail() == 1 && pm.M != M && M - pm.M <= 64 && K % 512 == 0 && epi->splitk <= 1;
        pm.tail_m = pm.M;
        pm.tail_rows = fold ? (int)(M - pm.M) : 0;
        bool folded = false;
        int rc = hh_gemm256_launch(pm, (hipStream_t)stream, &folded);
        if (rc == HH_OK && epi->z_out != nullptr) *stats_done_rows = pm.M;      // (the persistent kernel left per-slice row sums in z_partials)
        if (rc != HH_OK || pm.M == M || folded) return rc;
        p.m_start = pm.M;
    }
    if (M - p.m_start <= 64 && K % 256 == 0 && epi->splitk <= 1 && hh_tuning_gemm_tail()) {
        // (< 64)-row tail of a tall GEMM, or a GEMM that is this short altogether: split-K-in-workgroup kernel, 32-column workgroups,
        // 8 waves when K splits 8 ways into multiples of 64, one or two 32-row tiles
        const bool bf = epi->c_dtype == HH_BF16, w8 = K % 512 == 0, two = M - p.m_start > 32;
        const dim3 grid((unsigned)(N / 32));
        hipStream_t ts = (hipStream_t)stream;
        HHProfScope prof(HH_PROF_GEMM_OTHER, 2.0 * (double)(M - p.m_start) * N * K, ts);
#define TAIL(BF, NWT, MT) hipLaunchKernelGGL((gemm_tail_kernel<BF, NWT, MT>), grid, dim3(64 * NWT), 0, ts, p)
        if (bf) { if (w8) { if (two) TAIL(true, 8, 2); else TAIL(true, 8, 1); } else { if (two) TAIL(true, 4, 2); else TAIL(true, 4, 1); } }
        else    { if (w8) { if (two) TAIL(false, 8, 2); else TAIL(false, 8, 1); } else { if (two) TAIL(false, 4, 2); else TAIL(false, 4, 1); } }
#undef TAIL
        return hh_check_launch("hh_gemm_bf16(tail)");
    }
    p.Mt = (int)((M - p.m_start + BM - 1) / BM);
    p.Nt = N / BN;
    const int per_xcd_mt = (p.Mt + 7) / 8;
    const int groups = (per_xcd_mt + GROUP_M - 1) / GROUP_M;
    const unsigned grid = p.Mt < 8 ? (unsigned)(p.Mt * p.Nt) : 8u * (unsigned)groups * GROUP_M * (unsigned)p.Nt;
    hipStream_t s = (hipStream_t)stream;
    static std::atomic<uint64_t> attr_mask{0};
    if (hh_attr_needed(attr_mask)) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_gemm_bf16: cannot reserve 128 KB of LDS for the 128x128 kernel: %s", hipGetErrorString(e));
        hh_attr_done(attr_mask);
    }
    const unsigned splits = epi->splitk > 1 ? (unsigned)epi->splitk : 1u;
    HH_REQUIRE(splits <= 1024, HH_ERR_SHAPE, "hh_gemm_bf16: splitk too large");
    HH_REQUIRE(splits == 1 || (epi->resid == nullptr && epi->split_stride >= 0), HH_ERR_SHAPE, "hh_gemm_bf16: split-K partials take no residual");
    // deep ring when the launch cannot fill the chip twice anyway (real tiles, not the padded grid)
    const bool deep = (int64_t)p.Mt * p.Nt * splits <= 512 && K >= 256;
    HHProfScope prof(HH_PROF_GEMM_OTHER, 2.0 * (double)(M - p.m_start) * N * K, s);
    if (deep) {
        if (epi->c_dtype == HH_BF16) hipLaunchKernelGGL((gemm_bf16_kernel<true, 4>), dim3(grid, splits), dim3(256), 131072, s, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<false, 4>), dim3(grid, splits), dim3(256), 131072, s, p);
    } else {
        if (epi->c_dtype == HH_BF16) hipLaunchKernelGGL((gemm_bf16_kernel<true, 2>), dim3(grid, splits), dim3(256), 65536, s, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<false, 2>), dim3(grid, splits), dim3(256), 65536, s, p);
    }
    return hh_check_launch("hh_gemm_bf16");
}

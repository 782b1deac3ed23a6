// bf16 MFMA GEMM, 256x256 block tile, 8-phase software pipeline (the large-shape path of hh_gemm_bf16).
//
// Geometry: BM = BN = 256, BK = 64, 512 threads = 8 waves as 2 (M) x 4 (N).  LDS = 128 KiB ring of HALF-TILES
// (128 rows x 64 k bf16 = 16 KiB): [2 K-tile buffers] x {B-lo, A-lo, B-hi, A-hi}.  A wave's 128 x 64 output is made of
// 64 rows from A-lo + 64 rows from A-hi and 32 columns from B-lo + 32 from B-hi, so the four quadrant products of one
// K-tile consume the half-tiles in the order (A-lo, B-lo) -> B-hi -> A-hi and free them one by one:
//   phase 1: read B0 (B-lo, 4 ds_read_b128), A0 (A-lo, 8)   MFMA A0 x B0
//   phase 2: read B1 (B-hi, 4)                              MFMA A0 x B1
//   phase 3: read A1 (A-hi, 8; overwrites A0)               MFMA A1 x B1
//   phase 4: -                                              MFMA A1 x B0
// Each phase = {ds_reads ; ONE half-tile LDS-DMA prefetch (2 global_load_lds_dwordx4 per wave) ; s_barrier ;
// 16 x mfma_f32_16x16x32_bf16 ; s_barrier}.  Prefetch stream (tile t computing from buffer t&1):
//   phase 1 -> A-hi(t+1) into buffer (t+1)&1 ; phases 2,3,4 -> B-lo, A-lo, B-hi of tile t+2 into buffer t&1
// (each slot is re-staged only after its reads retired: B-lo's 4 reads are retired by the lgkmcnt(8) ahead of phase 1's
// first barrier, the others are >= 2 phases old).  ONE counted s_waitcnt vmcnt(6) per K-tile (phase 4) retires tile t+1
// while 3 half-tiles stay in flight across the barriers; never vmcnt(0) in the steady state.
// The wave rows (wr = 1) run one barrier behind (stagger): while one group issues MFMAs the other group on the same
// SIMDs reads LDS / issues DMA, which keeps the matrix pipe fed.
// LDS image of a half-tile: rows of 8 x 16-B chunks, chunk c of row r at position c ^ (r & 7) (swizzle on the DMA
// source address; ds_read_b128 fragment reads conflict-free).  Operands swapped (A-operand = W) so a lane owns 4
// consecutive n for one m: vector epilogue shared with gemm.hip.
#include "gemm_common.h"
#include <type_traits>
#include <string.h>
#include <stdlib.h>

#define HT_BYTES 16384
#define BUF_BYTES 65536
// half-tile slots inside a K-tile buffer
#define SLOT_BLO 0
#define SLOT_ALO 1
#define SLOT_BHI 2
#define SLOT_AHI 3

#define BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

template <bool OUT_BF16, bool STAGGER>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- XCD-aware tile assignment (same scheme as gemm.hip, 4 m-tiles x all n-tiles per XCD-local group)
    const int GROUP = p.group_m;
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int per = GROUP * p.Nt;
    const int kg = j / per, r = j % per;
    const int nt_i = r / GROUP, mi = r % GROUP;
    const int mt = xcd + 8 * (kg * GROUP + mi);
    if (mt >= p.Mt) return;
    const int64_t m0 = (int64_t)mt * 256;
    const int n0 = nt_i * 256;
    // Start-time skew of the first round of blocks (one per CU): without it all CUs reach their epilogues together and the
    // chip alternates between an HBM-idle main loop and an HBM-saturated store/residual burst.
    if (p.skew_iters > 0 && b < 256) {
        const int it = (int)(((int64_t)p.skew_iters * b) >> 8);
        for (int i = 0; i < it; ++i) __builtin_amdgcn_s_sleep(16);
    }

    // ---- staging sources: wave w stages pieces 2w, 2w+1 (8 rows each) of every half-tile
    const bf16_t* src[4][2];       // [slot][piece]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + (lane >> 3);           // row inside the half-tile, 0..127
        const int c = (lane & 7) ^ (row & 7);
        int64_t ma = m0 + row, mb = m0 + 128 + row;
        if (ma > p.M - 1) ma = p.M - 1;
        if (mb > p.M - 1) mb = p.M - 1;
        src[SLOT_ALO][i] = p.A + ma * p.lda + c * 8;
        src[SLOT_AHI][i] = p.A + mb * p.lda + c * 8;
        // W rows are permuted inside each 32-row group while staging: LDS row 16*tn + i holds n = 8*(i>>2) + 4*tn + (i&3),
        // so that an output lane (which owns MFMA rows 4g..4g+3 of both tn tiles) ends up with 8 CONSECUTIVE n
        // (one 16-byte store per tile pair instead of two 8-byte ones; the epilogue was store-issue bound).
        const int rl = row & 31, nperm = (row & ~31) + 8 * ((rl & 15) >> 2) + 4 * (rl >> 4) + (rl & 3);
        src[SLOT_BLO][i] = p.W + (int64_t)(n0 + nperm) * p.ldw + c * 8;
        src[SLOT_BHI][i] = p.W + (int64_t)(n0 + 128 + nperm) * p.ldw + c * 8;
    }
    const int nk = p.K / 64;
    auto stage = [&](int slot, int kt, int buf) {
        char* dst = smem + buf * BUF_BYTES + slot * HT_BYTES + wave * 2048;
        glds16(src[slot][0] + (int64_t)kt * 64, dst);
        glds16(src[slot][1] + (int64_t)kt * 64, dst + 1024);
    };

    // ---- fragment read offsets inside a half-tile: row = base + (lane & 15), chunk = ks*4 + (lane >> 4)
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + fq;
        a_off[ks] = (wr * 64 + frow) * 128 + ((c ^ (frow & 7)) << 4);     // + tm*2048
        b_off[ks] = (wc * 32 + frow) * 128 + ((c ^ (frow & 7)) << 4);     // + tn*2048
    }

    f32x4 acc[2][4][2][2];      // [mh][tm][nh][tn]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int e = 0; e < 2; ++e) acc[a][c][d][e] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: tile 0 (4 half-tiles) + first three half-tiles of tile 1
    stage(SLOT_BLO, 0, 0); stage(SLOT_ALO, 0, 0); stage(SLOT_BHI, 0, 0); stage(SLOT_AHI, 0, 0);
    if (nk > 1) {
        stage(SLOT_BLO, 1, 1); stage(SLOT_ALO, 1, 1); stage(SLOT_BHI, 1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    BARRIER();
    if (STAGGER && wr == 1) BARRIER();

    bf16x8 af[4][2], bf0[2][2], bf1[2][2];      // A sub-tile [tm][ks]; B0 / B1 sub-tiles [tn][ks]

#define MFMA_QUAD(MH, NH, BF)                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                   \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                               \
            _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                           \
                acc[MH][tm][NH][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[tn][ks], af[tm][ks], acc[MH][tm][NH][tn], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const char* base = smem + cur * BUF_BYTES;
        // ================= phase 1: A0 x B0
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bf0[tn][ks] = *(const bf16x8*)(base + SLOT_BLO * HT_BYTES + b_off[ks] + tn * 2048);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[tm][ks] = *(const bf16x8*)(base + SLOT_ALO * HT_BYTES + a_off[ks] + tm * 2048);
        if (t + 1 < nk) stage(SLOT_AHI, t + 1, cur ^ 1);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");          // the 4 B-lo reads (issued first) have retired
        BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        MFMA_QUAD(0, 0, bf0)
        BARRIER();
        // ================= phase 2: A0 x B1
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bf1[tn][ks] = *(const bf16x8*)(base + SLOT_BHI * HT_BYTES + b_off[ks] + tn * 2048);
        if (t + 2 < nk) stage(SLOT_BLO, t + 2, cur);
        BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        MFMA_QUAD(0, 1, bf1)
        BARRIER();
        // ================= phase 3: A1 x B1
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[tm][ks] = *(const bf16x8*)(base + SLOT_AHI * HT_BYTES + a_off[ks] + tm * 2048);
        if (t + 2 < nk) stage(SLOT_ALO, t + 2, cur);
        BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        MFMA_QUAD(1, 1, bf1)
        BARRIER();
        // ================= phase 4: A1 x B0 ; retire tile t+1
        if (t + 2 < nk) {
            stage(SLOT_BHI, t + 2, cur);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        MFMA_QUAD(1, 0, bf0)
        BARRIER();
    }
    if (STAGGER && wr == 0) BARRIER();
#undef MFMA_QUAD

    // ---- epilogue: lane owns C[m][n .. n+7]; residual rows are fetched in batches of 16 vectors (one M half) BEFORE any
    // use, so the read-modify-write is not serialised on memory latency
    const hh_gemm_epilogue& e = p.e;
    char* Cbase = (char*)p.C;
    const int ncol = n0 + wc * 32 + 8 * fq;                  // + nh*128 ; lane owns n .. n+7 (tile tn at +4*tn)
    constexpr int TN_OFF = 4;
    f32x4 bias_v[2][2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
        bias_v[nh][0] = e.bias ? *(const f32x4*)(e.bias + ncol + nh * 128) : (f32x4){0.f, 0.f, 0.f, 0.f};
        bias_v[nh][1] = e.bias ? *(const f32x4*)(e.bias + ncol + nh * 128 + TN_OFF) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        int64_t orow[4];
        bool live[4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const int64_t m = m0 + mh * 128 + wr * 64 + tm * 16 + frow;
            live[tm] = m < p.M;
            orow[tm] = (e.remap_group > 0) ? m + (m / e.remap_group) * e.remap_skip + e.remap_offset : m;
        }
        f32x4 rv[4][2][2];
        if (e.resid) {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const float* rp = e.resid + (live[tm] ? orow[tm] : 0) * e.ldr + ncol + nh * 128;
                    rv[tm][nh][0] = *(const f32x4*)rp;
                    rv[tm][nh][1] = *(const f32x4*)(rp + TN_OFF);
                }
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                const int n = ncol + nh * 128;
                f32x4 v0 = acc[mh][tm][nh][0] + bias_v[nh][0], v1 = acc[mh][tm][nh][1] + bias_v[nh][1];
                if (n < e.colscale_cols) { v0 *= e.colscale; v1 *= e.colscale; }
                if (e.act == HH_ACT_QUICKGELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v0[q] = quick_gelu(v0[q]); v1[q] = quick_gelu(v1[q]); }
                } else if (e.act == HH_ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
                }
                if (e.resid) { v0 += rv[tm][nh][0]; v1 += rv[tm][nh][1]; }
                if (!live[tm] || p.debug_nostore) continue;
                if constexpr (OUT_BF16) {
                    u32x4 o = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
                    *(u32x4*)((bf16_t*)Cbase + orow[tm] * p.ldc + gemm_ccol(e, n)) = o;
                } else {
                    *(f32x4*)((float*)Cbase + orow[tm] * p.ldc + gemm_ccol(e, n)) = v0;
                    *(f32x4*)((float*)Cbase + orow[tm] * p.ldc + gemm_ccol(e, n) + TN_OFF) = v1;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT, CONTINUOUS, FOUR-BARRIER kernel (the default for full 256-row tiles without residual / row remap, K >= 128).
//   * persistent: one workgroup per CU walks its tiles (virtual block id v = blockIdx.x + i * gridDim.x, same XCD-aware decode).
//     Per-block fixed costs measured on the one-tile-per-block kernel above: ~5.5 us of epilogue store drain + 3.5-6 us of prologue
//     latency / launch on a ~23 us main loop at K = 1024.
//   * continuous: the k-tile stream does not stop at tile boundaries.  During the last two k-tiles of a tile the staging slots that
//     have nothing left to fetch for it fetch k-tiles 0 and 1 of the workgroup's NEXT tile, so the next main loop starts right after
//     the epilogue: no 16-instruction prologue (1.1 us) and no wait for its first k-tile (2-3 us).  Since vmcnt retires in order
//     and the epilogue stores are younger than those loads, the waits of the next tile's first k-tile are counted as
//     vmcnt(N + STORES) and pass without waiting for the store drain (no other VMEM instruction may sit between them: hence no
//     residual / row remap, and the bias vector lives in LDS).  Source addresses are scalar tile bases + 4 per-lane 32-bit offsets.
//   * four barriers per k-tile: the two A-lo phases and the two A-hi phases of a k-tile are merged (32 MFMAs per wave between two
//     barriers instead of 16; one counted vmcnt wait per k-tile as before).  The in-kernel timeline put the eight-barrier main loop
//     at 2450-2680 cycles per k-tile against 2048 MFMA cycles: ~56 cycles are lost at each barrier hand-over between the two waves
//     of a SIMD; halving the hand-overs gave qkv 1170 -> 1225, fc1 1133 -> 1223, fc2 1305 -> 1350 TFLOP/s.  (One phase per k-tile --
//     64 MFMAs per hand-over -- was tried: slower, 1040 TFLOP/s, the prefetch lead shrinks to one k-tile, and it raced in one test.)
//   * EPI: the epilogue flavour is a template parameter (0 bias only, 1 bias + colscale on whole 128-column halves, 2 bias +
//     QuickGELU, 3 bias + ReLU).  A generic epilogue spends ~30 VALU instructions per 16-B store (~500 per tile and wave, 2.8 us of
//     matrix-core idle time per K = 1024 tile); bias-only needs 8.
// Earlier generations (persistent with a per-tile prologue: 1127 TFLOP/s in-step; continuous with eight barriers: qkv 1162 / fc1 1200)
// were measured against this kernel in round 1 ("GEMM kernel generations") and removed in round 2.
// debug timeline (hh_set_tuning("gemm256_debug_ts", 1)): s_memrealtime (100 MHz) of wave 0 at 5 points of the first 8 tiles of
// every workgroup; read back with hh_debug_gemm_timeline
#define TS_TILES 8
__device__ unsigned long long g_gemm_ts[512 * TS_TILES * 7];       // 5 x s_memrealtime + s_memtime at stamps 1, 2

// <= 32 rows x 32 columns of the (<= 64-row) tail inside the persistent kernel (see gemm.hip: gemm_tail_kernel, same arithmetic): the 8 waves
// split K, each streaming its operands from global memory directly in MFMA layout, the partial accumulators meet in `red` (32 KiB
// of the still unused staging ring) and wave 0 finishes the piece with the common epilogue.
template <bool OUT_BF16>
__device__ __forceinline__ void gemm256_tail_piece(const GemmParams& p, int piece, float* red, int tid) {
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int ncb = p.N / 32;                                     // pieces [0, ncb): tail rows 0..31, [ncb, 2 ncb): rows 32..63
    const int n0 = (piece % ncb) * 32, r0 = (piece / ncb) * 32;
    const int kslice = p.K / 8;                                   // multiple of 64 (K % 512 == 0, checked by the launcher)
    const int k_begin = wave * kslice;
    const bf16_t* ap = p.A + (p.tail_m + min(r0 + l31, p.tail_rows - 1)) * p.lda + k_begin + 8 * h;
    const bf16_t* wp = p.W + (int64_t)(n0 + l31) * p.ldw + k_begin + 8 * h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < kslice; k += 64) {
        bf16x8 af[4], wf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { wf[s] = *(const bf16x8*)(wp + k + 16 * s); af[s] = *(const bf16x8*)(ap + k + 16 * s); }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], af[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0 && r0 + l31 < p.tail_rows) {
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[(w * 16 + r) * 64 + lane];
            t[r] = v;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {                             // accumulator register r: n = 8 (r >> 2) + 4 h + (r & 3)
            const f32x4 v = {t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
            gemm_store4<OUT_BF16>(p.e, (char*)p.C, p.ldc, p.tail_m + r0 + l31, n0 + 8 * g + 4 * h, v);
        }
    }
    __syncthreads();                                              // `red` is staging-ring memory
}

template <bool OUT_BF16, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256d_kernel(GemmParams p) {
    constexpr bool STAGGER = true;
    constexpr int STORES = OUT_BF16 ? 16 : 32;        // global_store_dwordx4 per wave and tile in the epilogue (checked in the ISA)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int GROUP = p.group_m;
    const int per = GROUP * p.Nt;
    const int nk = p.K / 64;
    const int vmax = 8 * ((((p.Mt + 7) / 8) + GROUP - 1) / GROUP) * GROUP * p.Nt;      // size of the virtual grid
    auto decode = [&](int v, int64_t& m0, int& n0) -> bool {
        const int xcd = v & 7, j = v >> 3;
        const int kg = j / per, r = j % per;
        const int nt_i = r / GROUP, mi = r % GROUP;
        const int mt = xcd + 8 * (kg * GROUP + mi);
        m0 = (int64_t)mt * 256;
        n0 = nt_i * 256;
        return mt < p.Mt;
    };
    auto next_valid = [&](int v, int64_t& m0, int& n0) -> int {       // first valid virtual id >= v on this block's stride, or -1
        for (; v < vmax; v += gridDim.x)
            if (decode(v, m0, n0)) return v;
        return -1;
    };
    int64_t m0, nm0 = 0;
    int n0, nn0 = 0;
    int v = next_valid(blockIdx.x, m0, n0);
    if (p.tail_rows > 0)
        for (int piece = blockIdx.x; piece < (p.N / 32) * ((p.tail_rows + 31) / 32); piece += gridDim.x) gemm256_tail_piece<OUT_BF16>(p, piece, (float*)smem, tid);
    if (v < 0) return;
    // ---- staging sources: wave w stages pieces 2w, 2w+1 (8 rows each) of every half-tile.  The per-lane part of a source address
    // (row inside the half-tile, swizzled 16-B chunk) does not depend on the tile: 4 x 32-bit byte offsets; the tile part is a
    // scalar base, kept for the CURRENT and for the NEXT tile of this workgroup's walk.
    unsigned aoff[2], woff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        aoff[i] = (unsigned)((row * p.lda + c * 8) * 2);
        const int rl = row & 31, nperm = (row & ~31) + 8 * ((rl & 15) >> 2) + 4 * (rl >> 4) + (rl & 3);
        woff[i] = (unsigned)((nperm * p.ldw + c * 8) * 2);
    }
    const int64_t hiA = 128 * p.lda * 2, hiW = 128 * p.ldw * 2;          // bytes from the low to the high half-tile
    const char* cA = (const char*)(p.A + m0 * p.lda);
    const char* cW = (const char*)(p.W + (int64_t)n0 * p.ldw);
    const char* nA = cA;
    const char* nW = cW;
    auto stage_from = [&](const char* bA, const char* bW, int slot, int kt, int buf) {
        char* dst = smem + buf * BUF_BYTES + slot * HT_BYTES + wave * 2048;
        const bool isA = slot == SLOT_ALO || slot == SLOT_AHI;
        const char* bp = (isA ? bA : bW) + ((slot == SLOT_AHI) ? hiA : (slot == SLOT_BHI) ? hiW : 0) + (int64_t)kt * 128;
        asm volatile("" : "+s"(bp));            // keep the scalar base scalar: without this LLVM hoists 64-bit per-lane sums out of the k loop
        glds16(bp + (isA ? aoff[0] : woff[0]), dst);
        glds16(bp + (isA ? aoff[1] : woff[1]), dst + 1024);
    };
    // the whole bias vector lives in the LDS left over by the two staging buffers (N <= 8192): the epilogue used to fetch its
    // 16 bias values from global memory and stall on vmcnt(0) (~1.5 us per tile) before it could issue the next prologue
    float* bias_s = (float*)(smem + 2 * BUF_BYTES);
    for (int i = tid * 4; i < p.N; i += 512 * 4)
        *(f32x4*)(bias_s + i) = p.e.bias ? *(const f32x4*)(p.e.bias + i) : (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    BARRIER();
    // start-time skew: with equal tile times all 256 workgroups reach their epilogues together and 33 MB of stores hit the memory
    // system at once (the ~5 us between two main loops are that drain); p.skew_iters spreads the workgroups of an XCD over one
    // drain time, and the spread persists from tile to tile
    if (p.skew_iters > 0) {
        const int it = p.skew_iters * (int)((blockIdx.x >> 3) & 31);
        for (int i = 0; i < it; ++i) __builtin_amdgcn_s_sleep(8);
    }
    {   // first tile only: k-tiles 0 and 1 complete (16 LDS-DMA instructions per wave)
        stage_from(cA, cW, SLOT_BLO, 0, 0); stage_from(cA, cW, SLOT_ALO, 0, 0); stage_from(cA, cW, SLOT_BHI, 0, 0); stage_from(cA, cW, SLOT_AHI, 0, 0);
        stage_from(cA, cW, SLOT_BLO, 1, 1); stage_from(cA, cW, SLOT_ALO, 1, 1); stage_from(cA, cW, SLOT_BHI, 1, 1); stage_from(cA, cW, SLOT_AHI, 1, 1);
    }
    bool first = true;
    int kpar = 0;                   // LDS buffer of this tile's k-tile 0 (the k-tile stream runs on across tiles)
    int tile_i = 0;
    auto stamp = [&](int k) {
        if (p.debug_ts && tid == 0 && tile_i < TS_TILES && blockIdx.x < 512)
        {
            unsigned long long* r = g_gemm_ts + ((int)blockIdx.x * TS_TILES + tile_i) * 7;
            r[k] = __builtin_amdgcn_s_memrealtime();
            if (k == 1 || k == 2) r[4 + k] = __builtin_amdgcn_s_memtime();
        }
    };

    // ---- fragment read offsets inside a half-tile: row = base + (lane & 15), chunk = ks*4 + (lane >> 4)
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + fq;
        a_off[ks] = (wr * 64 + frow) * 128 + ((c ^ (frow & 7)) << 4);     // + tm*2048
        b_off[ks] = (wc * 32 + frow) * 128 + ((c ^ (frow & 7)) << 4);     // + tn*2048
    }

    f32x4 acc[2][4][2][2];      // [mh][tm][nh][tn]
    bf16x8 af[4][2], bf0[2][2], bf1[2][2];      // A sub-tile [tm][ks]; B0 / B1 sub-tiles [tn][ks]

#define MFMA_QUAD(MH, NH, BF)                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                   \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                               \
            _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                           \
                acc[MH][tm][NH][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[tn][ks], af[tm][ks], acc[MH][tm][NH][tn], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);

    for (;;) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int e = 0; e < 2; ++e) acc[a][c][d][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        stamp(0);
        // the walk's next tile, known up front: its k-tiles 0 and 1 are staged during THIS tile's last two k-tiles
        const int nv = next_valid(v + gridDim.x, nm0, nn0);
        const bool has_next = nv >= 0;
        if (has_next) {
            nA = (const char*)(p.A + nm0 * p.lda);
            nW = (const char*)(p.W + (int64_t)nn0 * p.ldw);
        }
        // k-tile 0: first tile -> from the prologue; later tiles -> retired by the previous tile's last phase-4 wait
        if (first) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        BARRIER();
        if (STAGGER && wr == 1) BARRIER();
        stamp(1);

        for (int t = 0; t < nk; ++t) {
            const int cur = (t + kpar) & 1;
            const char* base = smem + cur * BUF_BYTES;
            // ================= phase I: A0 x (B0, B1) -- 32 MFMAs per wave between two barriers
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bf0[tn][ks] = *(const bf16x8*)(base + SLOT_BLO * HT_BYTES + b_off[ks] + tn * 2048);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bf1[tn][ks] = *(const bf16x8*)(base + SLOT_BHI * HT_BYTES + b_off[ks] + tn * 2048);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[tm][ks] = *(const bf16x8*)(base + SLOT_ALO * HT_BYTES + a_off[ks] + tm * 2048);
            if (t >= 1) {                                                     // A-hi(1) came with the prologue / the previous epilogue
                if (t + 1 < nk) stage_from(cA, cW, SLOT_AHI, t + 1, cur ^ 1);
                else if (has_next) stage_from(nA, nW, SLOT_AHI, 0, cur ^ 1);
            }
            BARRIER();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            MFMA_QUAD(0, 0, bf0)
            MFMA_QUAD(0, 1, bf1)
            BARRIER();
            // ================= phase II: A1 x (B1, B0) ; stage k-tile t+2 (B-lo, A-lo, B-hi) ; retire k-tile t+1
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[tm][ks] = *(const bf16x8*)(base + SLOT_AHI * HT_BYTES + a_off[ks] + tm * 2048);
            if (t + 2 < nk || has_next) {
                if (t + 2 < nk) {
                    stage_from(cA, cW, SLOT_BLO, t + 2, cur); stage_from(cA, cW, SLOT_ALO, t + 2, cur); stage_from(cA, cW, SLOT_BHI, t + 2, cur);
                } else {
                    stage_from(nA, nW, SLOT_BLO, t + 2 - nk, cur); stage_from(nA, nW, SLOT_ALO, t + 2 - nk, cur); stage_from(nA, nW, SLOT_BHI, t + 2 - nk, cur);
                }
                if (t == 0 && !first) {                    // in order: [.. A-hi(1)][STORES][B-lo, A-lo, B-hi of k-tile 2] -> keep stores + 6
                    if (OUT_BF16) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                }
            } else {                                       // last tile of the walk, last two k-tiles: nothing staged
                if (t == 0 && !first) {                    // nk == 2: only the stores are younger than A-hi(1)
                    if (OUT_BF16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            BARRIER();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            MFMA_QUAD(1, 1, bf1)
            MFMA_QUAD(1, 0, bf0)
            BARRIER();
        }
        if (STAGGER && wr == 0) BARRIER();
        stamp(2);

        // ---- epilogue of tile (m0, n0).  Bias is fetched and waited for FIRST; then the next tile's prologue is issued;
        // then exactly STORES store instructions per wave and nothing else.
        const hh_gemm_epilogue& e = p.e;
        const int ncol = n0 + wc * 32 + 8 * fq;
        f32x4 bias_v[2][2];                    // from the LDS copy: no VMEM instruction, no vmcnt stall in front of the prologue
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            bias_v[nh][0] = *(const f32x4*)(bias_s + ncol + nh * 128);
            bias_v[nh][1] = *(const f32x4*)(bias_s + ncol + nh * 128 + 4);
        }
        // A-hi of the next tile's k-tile 1 goes out BEFORE the stores (it must not queue behind the store drain)
        if (has_next) stage_from(nA, nW, SLOT_AHI, 1, (kpar + nk + 1) & 1);
        stamp(3);
        char* Cbase = (char*)p.C;
        const int64_t ccol[2] = {gemm_ccol(e, ncol), gemm_ccol(e, ncol + 128)};
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                const int64_t orow = m0 + mh * 128 + wr * 64 + tm * 16 + frow;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const int n = ncol + nh * 128;
                    f32x4 v0 = acc[mh][tm][nh][0] + bias_v[nh][0], v1 = acc[mh][tm][nh][1] + bias_v[nh][1];
                    if constexpr (EPI == 1) {
                        if (n0 + nh * 128 < e.colscale_cols) { v0 *= e.colscale; v1 *= e.colscale; }      // uniform: colscale_cols % 128 == 0
                    } else if constexpr (EPI == 2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v0[q] = quick_gelu(v0[q]); v1[q] = quick_gelu(v1[q]); }
                    } else if constexpr (EPI == 3) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
                    }
                    if constexpr (OUT_BF16) {
                        u32x4 o = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
                        *(u32x4*)((bf16_t*)Cbase + orow * p.ldc + ccol[nh]) = o;
                    } else {
                        *(f32x4*)((float*)Cbase + orow * p.ldc + ccol[nh]) = v0;
                        *(f32x4*)((float*)Cbase + orow * p.ldc + ccol[nh] + 4) = v1;
                    }
                }
            }
        stamp(4);
        ++tile_i;
        if (!has_next) break;
        v = nv; m0 = nm0; n0 = nn0;
        cA = nA; cW = nW;
        kpar = (kpar + nk) & 1;
        first = false;
    }
#undef MFMA_QUAD
}


// ---- tuning state.  hh_set_tuning() is the library's ONLY process-global mutable state besides the per-stream CU budget table
// (runtime.cpp); no entry point reads the environment.  The knobs select between kernels that compute the same result.
static int g_debug_ts = 0;         // debug: record the per-tile timeline of the persistent kernel (1: every launch, last one wins; n > 1: only the n-th launch after the knob was set)
static int g_ts_count = 0;
static int g_nostore = 0;          // debug: skip the epilogue stores (timing experiments only)
static int g_group = 0;            // m-tiles per XCD-local group (weight-panel reuse factor); 0 = per-shape default
static int g_pskew = 0;            // persistent kernel: start skew quantum (s_sleep(8) units per workgroup index in its XCD)
static int g_skew = -1;            // start skew of the one-tile-per-block kernel: -1 auto (on with an fp32 residual), 0 off, 1 on
bool hh_gemm256w4p_ln_ext_ok(const GemmParams& p, bool w4);      // gemm256w4.hip: the 4-wave persistent kernel implements this LayerNorm-fold epilogue
int hh_gemm256w4p_ln_epi(const hh_gemm_epilogue& e);
void hh_gemm256w4p_set_ln_ext(int v);
// "gemm_tile224": 224-row tiles where they remove a partial round (hh_gemm256_tile_rows).  OFF by default: at M = 100384 (196-token
// frames), N = 1024 it gains 3-6 % per launch alone (profiles/r4_tile224.md), but the benchmarked row counts (n = 256 tokens per frame:
// M = B * 4097) are whole rounds of 256-row tiles already -- the rule below never selects it there
static int g_tile224 = 0;
static int g_ln_pskew = 0, g_ln_phases = 4;      // "gemm_ln_pskew" / "gemm_ln_phases": start skew of the LayerNorm-fold producer GEMMs (EPI 4)
static int g_mode = 5;             // "gemm256": 0 = 128x128 kernel only, 1 = one tile per block, 2 = + wave-row stagger, 3 = persistent 8-wave kernel, 4 = 4-wave kernel of gemm256w4.hip, one tile per block, 5 = persistent 4-wave kernel where K allows, else 3 (default)
static int g_dynamic = 1;           // "gemm256_dynamic": 4-wave persistent kernel takes its tiles from per-XCD atomic counters (1, default) or by static stride (0)
static int g_min_tiles = 192;       // "gemm256_min_tiles": fewest 256x256 tiles for which the 256x256 kernels are used
static int g_tail = 1;             // "gemm_tail": 1 = row tails of <= 64 rows inside the persistent kernel (in <= 32-row pieces), else the split-K-in-workgroup tail kernel (gemm.hip); 2 = always the tail kernel; 0 = the 128x128 kernel

int hh_tuning_gemm_tail() { return g_tail; }
static int g_space_dbg = 0;
int hh_tuning_space_debug() { return g_space_dbg; }
static int g_space_joint = 1;
int hh_tuning_space_joint() { return g_space_joint; }
static int g_space_waves = 0;      // waves per workgroup of the joint space kernel: 0 = automatic, 4 / 8 / 12 = force where the shape divides
int hh_tuning_space_waves() { return g_space_waves; }
static int g_space_prog = 1;       // 1 = progressive K / V staging where a specialised kernel exists (n = 576)
int hh_tuning_space_prog() { return g_space_prog; }
static int g_space_mfma32 = 1;     // "space_mfma32": 1 = space attention on the 32x32x16 software-pipelined kernel where n <= 256 and n % 64 == 0 (round 6); 0 = the joint-block kernel
int hh_tuning_space_mfma32() { return g_space_mfma32; }
static int g_mattn_no_ticket = 0;  // "mattn_no_ticket": tests only -- hh_mattn_* behave as if the stream-slot table were full (one key slice per unit)
int hh_tuning_mattn_no_ticket() { return g_mattn_no_ticket; }



extern "C" int hh_set_tuning(const char* name, int value) {
    if (name && !strcmp(name, "gemm256") && value >= 0 && value <= 5) { g_mode = value; return HH_OK; }
    if (name && !strcmp(name, "gemm_tail") && value >= 0 && value <= 2) { g_tail = value; return HH_OK; }
    if (name && !strcmp(name, "space_debug")) { g_space_dbg = value; return HH_OK; }
    if (name && !strcmp(name, "space_joint")) { g_space_joint = value; return HH_OK; }
    if (name && !strcmp(name, "space_prog") && value >= 0 && value <= 2) { g_space_prog = value; return HH_OK; }
    if (name && !strcmp(name, "space_mfma32") && value >= 0 && value <= 3) { g_space_mfma32 = value; return HH_OK; }
    if (name && !strcmp(name, "mattn_no_ticket") && (value == 0 || value == 1)) { g_mattn_no_ticket = value; return HH_OK; }
    if (name && !strcmp(name, "space_waves") && (value == 0 || value == 4 || value == 12)) { g_space_waves = value; return HH_OK; }
    if (name && !strcmp(name, "gemm256_skew")) { g_skew = value; return HH_OK; }
    if (name && !strcmp(name, "gemm256_pskew") && value >= 0 && value <= 64) { g_pskew = value; return HH_OK; }
    if (name && !strcmp(name, "gemm256_debug_nostore")) { g_nostore = value; return HH_OK; }
    if (name && !strcmp(name, "gemm256_debug_ts")) { g_debug_ts = value; g_ts_count = 0; return HH_OK; }
    if (name && !strcmp(name, "gemm256_dynamic") && (value == 0 || value == 1)) { g_dynamic = value; return HH_OK; }
    if (name && !strcmp(name, "gemm256_min_tiles") && value >= 1 && value <= 4096) { g_min_tiles = value; return HH_OK; }
    if (name && !strcmp(name, "gemm_tile224") && value >= 0 && value <= 1) { g_tile224 = value; return HH_OK; }
    if (name && !strcmp(name, "gemm_ln_pskew") && value >= 0 && value <= 256) { g_ln_pskew = value; return HH_OK; }
    if (name && !strcmp(name, "gemm_ln_phases") && value >= 0 && value <= 32) { g_ln_phases = value; return HH_OK; }
    if (name && !strcmp(name, "gemm_ln_w4") && (value == 0 || value == 1)) { hh_gemm256w4p_set_ln_ext(value); return HH_OK; }
    if (name && !strcmp(name, "gemm256_group") && value >= 0 && value <= 64) { g_group = value; return HH_OK; }
    hh_set_error("hh_set_tuning: unknown knob '%s' or value %d out of range", name ? name : "(null)", value);
    return HH_ERR_UNSUPPORTED;
}

#define P_LDS(N) (2 * BUF_BYTES + (size_t)(N) * 4)       // persistent kernel: two staging buffers + the bias vector


bool hh_gemm256_eligible(const GemmParams& p) {
    // at least ~3/4 of the CUs must get a 256x256 tile for the 8-wave kernels: below that the 128x128 kernel (4x the tiles, two workgroups
    // per CU) wins -- measured on the text tower's N = 768 shapes at M = 12320 (147 tiles): 43 vs 65 us (K = 768), 97 vs 152 us (K = 3072).
    // The persistent 4-wave kernel wins from half the CUs on (same shapes, 144 tiles: 30.4 vs 34.4 us and 75 vs 91 us alone, and it
    // holds 144 CUs instead of all of them beside the vision tower): "gemm256_min_tiles" / 128 for the shapes it takes
    const bool w4 = g_mode == 5 && p.K >= 384 && p.K % 128 == 0 && p.N <= 4096 && p.e.resid == nullptr && p.e.remap_group == 0 &&
                    (p.e.colscale_cols == 0 || (p.e.act == HH_ACT_NONE && p.e.colscale_cols % 128 == 0));
    const int min_tiles = (w4 && g_min_tiles == 192) ? 128 : g_min_tiles;
    if (gemm_ln_ext(p.e) && !hh_gemm256w4p_ln_ext_ok(p, w4)) return false;      // the LayerNorm fold: persistent 4-wave kernel or the generic 128x128 one
    return g_mode > 0 && p.N % 256 == 0 && p.M >= 2048 && p.e.splitk <= 1 && (p.M / 256) * (p.N / 256) >= min_tiles;
}

int hh_gemm256w4_launch(const GemmParams& p, unsigned grid, hipStream_t s);      // gemm256w4.hip (4 waves x 128x128, one tile per workgroup)
int hh_gemm256w4p_launch(const GemmParams& p, int epi, unsigned pg, hipStream_t s);   // gemm256w4.hip (4 waves x 128x128, persistent)
int hh_gemm256w4_timeline(unsigned long long* out, int blocks);
bool hh_gemm256w4_timeline_is_last();
void hh_gemm256w4_timeline_mark(bool w4);

// Tile height for a GEMM of p.M rows (all of them) that hh_gemm256_eligible admitted: 224 where the persistent 4-wave kernel has the
// instantiation (bf16, bias only or the LayerNorm-fold producer) and the 224-row tiling needs fewer round-equivalents -- a 224-row tile
// takes ~0.9 of a 256-row tile's time (7/8 of the MFMAs and epilogue bytes, the same W staging) -- and >= 16 rows follow the last tile
// (its A-hi staging over-reads that far).  M = 100384, N = 1024: 7 x 0.9 = 6.3 against 7 rounds; M = 131104 (config 2): 10 x 0.9 against 8 -> 256.
int hh_gemm256_tile_rows(const GemmParams& p, hipStream_t s) {
    if (!g_tile224 || g_mode != 5 || g_nostore || p.e.c_dtype != HH_BF16 || p.e.resid != nullptr || p.e.remap_group != 0) return 256;
    if (!(p.K >= 384 && p.K % 128 == 0 && p.N <= 4096)) return 256;
    const int epi = gemm_ln_ext(p.e) ? hh_gemm256w4p_ln_epi(p.e) : ((p.e.act == HH_ACT_NONE && p.e.colscale_cols == 0) ? 0 : -1);
    if (epi != 0 && epi != 4) return 256;
    const int64_t ncu = hh_stream_cu_count(s) & ~7, nt = p.N / 256;
    const int64_t mt224 = p.M / 224, mt256 = p.M / 256;
    if (ncu <= 0 || p.M - mt224 * 224 < 16) return 256;
    const int64_t r224 = (mt224 * nt + ncu - 1) / ncu, r256 = (mt256 * nt + ncu - 1) / ncu;
    return r224 * 90 < r256 * 98 ? 224 : 256;
}

int hh_gemm256_launch(const GemmParams& pin, hipStream_t s, bool* tail_done) {
    if (tail_done) *tail_done = false;
    static std::atomic<uint64_t> attr_mask{0};
    if (hh_attr_needed(attr_mask)) {
        hipError_t e = hipSuccess;
#define ATTR1(...) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm256_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES)
#define ATTRD(...) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm256d_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_LDS(8192))
        ATTR1(true, true); ATTR1(false, true); ATTR1(true, false); ATTR1(false, false);
        ATTRD(true, 0); ATTRD(false, 0); ATTRD(true, 1); ATTRD(false, 1); ATTRD(true, 2); ATTRD(false, 2); ATTRD(true, 3); ATTRD(false, 3);
#undef ATTR1
#undef ATTRD
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_gemm_bf16: cannot reserve %d B of LDS for the 256x256 kernels: %s", (int)P_LDS(8192), hipGetErrorString(e));
        hh_attr_done(attr_mask);
    }
    const int stagger = g_mode != 1;
    GemmParams p = pin;
    {
        // one round ~ nk k-tiles x ~3300 cycles + epilogue; s_sleep(16) ~ 1024 cycles per iteration
        const bool heavy = p.e.resid != nullptr || p.e.c_dtype == HH_F32;
        const bool on = g_skew < 0 ? heavy : g_skew > 0;
        p.skew_iters = on ? (int)(((int64_t)(p.K / 64) * 3300 + (heavy ? 30000 : 12000)) / 1024) : 0;
    }
    p.Mt = (int)((p.M + p.tile_rows - 1) / p.tile_rows);
    p.Nt = p.N / 256;
    // m-tiles per XCD-local group: 8 (8 m-tiles x 4 n-tiles per round and XCD).  Rounds 1-4 walked the short-K GEMMs up to 12 n-tiles (qkv,
    // proj) 16 x 2 (measured on the plain four-barrier kernel: proj 1084 -> 1122, qkv 1181 -> 1201 TFLOP/s); with the LayerNorm fold in their
    // epilogues 8 x 4 fetches 1289 instead of 2342 MiB per qkv launch and runs 1127 vs 1106 TFLOP/s alone (profiles/r5_gemm_group_fetch.md),
    // and the step gains +1.1 ... +1.7 % in three same-session A/Bs (profiles/r5_ab_gemm_group.txt); "gemm256_group" overrides
    const int GROUP = g_group > 0 ? g_group : 8;
    p.group_m = GROUP;
    p.rev_m = p.e.walk_reverse != 0;
    p.debug_nostore = g_nostore;
    p.debug_ts = g_debug_ts == 1;
    const int per_xcd_mt = (p.Mt + 7) / 8;
    const int groups = (per_xcd_mt + GROUP - 1) / GROUP;
    const unsigned grid = 8u * (unsigned)groups * GROUP * (unsigned)p.Nt;
    const bool bf = p.e.c_dtype == HH_BF16;
    if ((g_mode == 3 || g_mode == 5) && p.e.resid == nullptr && p.e.remap_group == 0 && p.K >= 128 && p.M % p.tile_rows == 0 && p.N <= 8192 && !g_nostore) {
        // epilogue flavour; a column scale together with an activation, or a scale boundary inside a 128-column half, take the
        // one-tile-per-block kernel below (generic epilogue)
        const bool scaled = p.e.colscale_cols > 0;
        int epi = -1;
        if (gemm_ln_ext(p.e)) epi = hh_gemm256w4p_ln_epi(p.e);      // (eligible only on the 4-wave kernel: hh_gemm256w4p_ln_ext_ok)
        else if (p.e.act == HH_ACT_NONE) epi = scaled ? (p.e.colscale_cols % 128 == 0 ? 1 : -1) : 0;
        else if (!scaled) epi = p.e.act == HH_ACT_QUICKGELU ? 2 : p.e.act == HH_ACT_RELU ? 3 : -1;
        if (epi >= 0) {
            const int ncu = hh_stream_cu_count(s) & ~7;      // CU budget of this stream; the stride of the tile walk must keep blockIdx & 7 == XCD
            const unsigned pg = grid < (unsigned)ncu ? grid : (unsigned)ncu;
            p.skew_iters = g_pskew;
            p.skew_phases = 0;
            if ((epi == 4 || epi == 7) && g_ln_pskew > 0) { p.skew_iters = g_ln_pskew; p.skew_phases = g_ln_phases; }
            HHProfScope prof(HH_PROF_GEMM256, 2.0 * (double)p.M * p.N * p.K, s);
            if (g_debug_ts > 1) p.debug_ts = (++g_ts_count == g_debug_ts);
            if (g_mode == 5 && p.K >= 384 && p.K % 128 == 0 && p.N <= 4096) {
                const int slot = g_dynamic ? hh_stream_slot(s) : -1;
                p.dynamic = slot >= 0;
                p.tile_slot = slot >= 0 ? slot : 0;      // (N: bias vector + epilogue scratch share the 32 KB of LDS the ring leaves)
                static const char* const w4p_names[18] = {"gemm256w4p_kernel<false, 0, 256>", "gemm256w4p_kernel<true, 0, 256>", "gemm256w4p_kernel<false, 1, 256>", "gemm256w4p_kernel<true, 1, 256>",
                                                          "gemm256w4p_kernel<false, 2, 256>", "gemm256w4p_kernel<true, 2, 256>", "gemm256w4p_kernel<false, 3, 256>", "gemm256w4p_kernel<true, 3, 256>",
                                                          "", "gemm256w4p_kernel<true, 4, 256>", "", "gemm256w4p_kernel<true, 5, 256>", "", "gemm256w4p_kernel<true, 6, 256>", "", "gemm256w4p_kernel<true, 7, 256>", "", "gemm256w4p_kernel<true, 8, 256>"};
                hh_prof_note_kernel(HH_PROF_GEMM256, p.tile_rows == 224 ? (epi == 4 ? "gemm256w4p_kernel<true, 4, 224>" : "gemm256w4p_kernel<true, 0, 224>") : w4p_names[epi * 2 + (bf ? 1 : 0)]);
                int rc = hh_gemm256w4p_launch(p, epi, pg, s);
                if (tail_done) *tail_done = p.tail_rows > 0;
                return rc;
            }
            if (epi >= 4 || p.tile_rows != 256) { hh_set_error("hh_gemm_bf16: internal: LayerNorm-fold epilogue / 224-row tiles on the 8-wave kernel"); return HH_ERR_UNSUPPORTED; }
            hh_gemm256w4_timeline_mark(false);
            hh_prof_note_kernel(HH_PROF_GEMM256, "gemm256d_kernel (8-wave persistent)");
#define LAUNCHD(BF, E) hipLaunchKernelGGL((gemm256d_kernel<BF, E>), dim3(pg), dim3(512), P_LDS(p.N), s, p)
            switch (epi * 2 + (bf ? 1 : 0)) {
                case 0: LAUNCHD(false, 0); break;
                case 1: LAUNCHD(true, 0); break;
                case 2: LAUNCHD(false, 1); break;
                case 3: LAUNCHD(true, 1); break;
                case 4: LAUNCHD(false, 2); break;
                case 5: LAUNCHD(true, 2); break;
                case 6: LAUNCHD(false, 3); break;
                default: LAUNCHD(true, 3); break;
            }
#undef LAUNCHD
            if (tail_done) *tail_done = p.tail_rows > 0;
            return hh_check_launch("hh_gemm_bf16(256x256 persistent)");
        }
    }
    if (p.tile_rows != 256) { hh_set_error("hh_gemm_bf16: internal: 224-row tiles outside the persistent 4-wave kernel"); return HH_ERR_UNSUPPORTED; }
    if (gemm_ln_ext(p.e)) {                  // (hh_gemm256_eligible admits the fold only where the persistent 4-wave kernel takes it)
        hh_set_error("hh_gemm_bf16: internal: LayerNorm-fold epilogue reached a 256x256 kernel that does not implement it");
        return HH_ERR_UNSUPPORTED;
    }
    HHProfScope prof(HH_PROF_GEMM_OTHER, 2.0 * (double)p.M * p.N * p.K, s);
    if (g_mode == 4 && p.K >= 256 && p.K % 128 == 0 && !g_nostore) return hh_gemm256w4_launch(p, grid, s);
    if (stagger) {
        if (bf) hipLaunchKernelGGL((gemm256_kernel<true, true>), dim3(grid), dim3(512), 2 * BUF_BYTES, s, p);
        else hipLaunchKernelGGL((gemm256_kernel<false, true>), dim3(grid), dim3(512), 2 * BUF_BYTES, s, p);
    } else {
        if (bf) hipLaunchKernelGGL((gemm256_kernel<true, false>), dim3(grid), dim3(512), 2 * BUF_BYTES, s, p);
        else hipLaunchKernelGGL((gemm256_kernel<false, false>), dim3(grid), dim3(512), 2 * BUF_BYTES, s, p);
    }
    return hh_check_launch("hh_gemm_bf16(256x256)");
}

// debug: copy the timeline of the last persistent launch (hh_set_tuning("gemm256_debug_ts", 1)); out[blocks][8 tiles][7]: 5 stamps + shader-clock counter at stamps 1 and 2
extern "C" int hh_debug_gemm_timeline(unsigned long long* out, int blocks) {
    HH_REQUIRE(out != nullptr && blocks > 0 && blocks <= 512, HH_ERR_SHAPE, "hh_debug_gemm_timeline: bad arguments");
    if (hh_gemm256w4_timeline_is_last()) return hh_gemm256w4_timeline(out, blocks);
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_ts), sizeof(unsigned long long) * (size_t)blocks * TS_TILES * 7);
    HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_debug_gemm_timeline: %s", hipGetErrorString(e));
    return HH_OK;
}

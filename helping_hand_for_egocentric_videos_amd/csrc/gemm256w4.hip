// The 256x256x64 bf16 GEMM tile on FOUR waves of 128x128 instead of eight of 128x64 -- one wave per SIMD with all 512 registers
// (256 accumulators in AGPRs + four fragment sets).  Replaces nn.Linear on the hot path like gemm256.hip (model/LaviLa.py:249 qkv,
// :281 proj, :186-189 fc1 / fc2).  Motivation: the step sits at the package power cap, where throughput = power / energy per flop; with 8
// waves every k-tile costs 192 KB of LDS fragment reads per workgroup (each A fragment is read by 4 waves, each W fragment by 2), with
// 4 waves 128 KB, and half the waits and barriers.  Same LDS image, staging and epilogue conventions as gemm256.hip, bit-identical
// results.  Two kernels: gemm256w4_kernel (hh_set_tuning("gemm256", 4)): one tile per workgroup, compiler-scheduled -- the simple
// form, kept as the readable statement of the pipeline and measured at 770-1090 TFLOP/s; gemm256w4p_kernel (5, the default):
// persistent, hand-placed instruction stream, 1215-1469 TFLOP/s.
//
// Software pipeline of the single wave per SIMD: the four quadrant products of a k-tile run back to back, 32 MFMAs each, and the 8
// ds_read_b128 of the NEXT quadrant's fresh operand are interleaved with them (two k-tiles per loop iteration, because the roles of
// the two W fragment sets swap):
//   Q0(t) = A-lo x W-lo   reads W-hi(t)          stages W-lo(t+2)
//   Q1(t) = A-lo x W-hi   reads A-hi(t)          stages W-hi(t+2)
//   Q2(t) = A-hi x W-hi   reads A-lo(t+1)        stages A-hi(t+2)
//   Q3(t) = A-hi x W-lo   reads W-lo(t+1)        stages A-lo(t+3)
// A half-tile is re-staged right after the barrier that follows its last read and is consumed seven quadrants later: seven
// half-tile DMAs (4 instructions per wave each) are in flight after each staging call, one counted s_waitcnt vmcnt(24) + one s_barrier per quadrant.
#include "gemm_common.h"
// epilogue traffic (C / z / residual rows: streamed once per launch) with the non-temporal cache policy when HH_EPI_NT is defined: it should
// not evict the A / W panels the tile walk keeps in L2 (experiment, round 5)
#ifdef HH_EPI_NT
#define W4_ST(P, V) __builtin_nontemporal_store((V), (P))
#define W4_LD(P) __builtin_nontemporal_load(P)
#else
#define W4_ST(P, V) (*(P) = (V))
#define W4_LD(P) (*(P))
#endif
#if defined(HH_EPI_NT) || defined(HH_EPI_NTX)      // NTX: only the fp32 residual rows (read and written once per producer launch)
#define W4_STX(P, V) __builtin_nontemporal_store((V), (P))
#define W4_LDX(P) __builtin_nontemporal_load(P)
#else
#define W4_STX(P, V) (*(P) = (V))
#define W4_LDX(P) (*(P))
#endif

#define W4_HT 16384
#define W4_BUF 65536
#define W4_BLO 0
#define W4_ALO 1
#define W4_BHI 2
#define W4_AHI 3

#define W4_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// The 256 accumulator registers are asm-owned AGPRs a[0:255] (an MFMA takes C / D there; the 128 fragment registers + addresses
// fill the VGPR half of the unified file).  As C++ variables hipcc kept them in VGPRs and spilled 111-146 registers -- every reload
// is a VMEM load followed by s_waitcnt vmcnt(0), which also drains the LDS-DMA pipeline.  Accumulator (mh, tm, nh, tn) lives in
// a[16 * ((mh*4 + tm)*2 + nh) + 4*tn .. +3]; 16 independent MFMAs separate two on the same accumulator.
template <int BASE>
__device__ __forceinline__ void w4_mfma(const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(w), "v"(a), "i"(BASE), "i"(BASE + 3));
}
template <int MH, int NH, int I>
__device__ __forceinline__ void w4_quad(const bf16x8 (&AF)[4][2], const bf16x8 (&WF)[4][2]) {
    constexpr int ks = I / 16, tn = (I / 4) % 4, tm = I % 4;
    w4_mfma<4 * (((MH * 4 + tm) * 2 + NH) * 4 + tn)>(WF[tn][ks], AF[tm][ks]);
    if constexpr (I + 1 < 32) w4_quad<MH, NH, I + 1>(AF, WF);
}
template <int I>
__device__ __forceinline__ void w4_zero() {
    asm volatile("v_accvgpr_write_b32 a[%c0], 0" :: "i"(I));
    if constexpr (I + 1 < 256) w4_zero<I + 1>();
}
template <int BASE>
__device__ __forceinline__ f32x4 w4_acc_read() {
    f32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n v_accvgpr_read_b32 %1, a[%c5]\n v_accvgpr_read_b32 %2, a[%c6]\n v_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "i"(BASE), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3));
    return v;
}
#define A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define W4_CLOBBER_AGPRS() asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", A8(1), A8(2), A8(3), A8(4), A8(5), A8(6), A8(7), A8(8), \
    A8(9), A8(10), A8(11), A8(12), A8(13), A8(14), A8(15), A8(16), A8(17), A8(18), A8(19), A8(20), A8(21), A8(22), A8(23), A8(24), "a250", "a251", "a252", "a253", \
    "a254", "a255")

template <bool OUT_BF16, int MH, int TM>
__device__ __forceinline__ void w4_store_rows(const GemmParams& p, int64_t m0, int n0, int wr, int wc, int frow, int fq) {
    const hh_gemm_epilogue& e = p.e;
    const int64_t m = m0 + MH * 128 + wr * 64 + TM * 16 + frow;
    if (m < p.M) {
        const int64_t orow = (e.remap_group > 0) ? m + (m / e.remap_group) * e.remap_skip + e.remap_offset : m;
        constexpr int B0 = 16 * ((MH * 4 + TM) * 2 + 0), B1 = 16 * ((MH * 4 + TM) * 2 + 1);
        const int nb = n0 + wc * 64 + 8 * fq;
        gemm_store8<OUT_BF16>(e, (char*)p.C, p.ldc, orow, nb, w4_acc_read<B0>(), w4_acc_read<B0 + 4>());
        gemm_store8<OUT_BF16>(e, (char*)p.C, p.ldc, orow, nb + 32, w4_acc_read<B0 + 8>(), w4_acc_read<B0 + 12>());
        gemm_store8<OUT_BF16>(e, (char*)p.C, p.ldc, orow, nb + 128, w4_acc_read<B1>(), w4_acc_read<B1 + 4>());
        gemm_store8<OUT_BF16>(e, (char*)p.C, p.ldc, orow, nb + 160, w4_acc_read<B1 + 8>(), w4_acc_read<B1 + 12>());
    }
    if constexpr (TM + 1 < 4) w4_store_rows<OUT_BF16, MH, TM + 1>(p, m0, n0, wr, wc, frow, fq);
    else if constexpr (MH == 0) w4_store_rows<OUT_BF16, 1, 0>(p, m0, n0, wr, wc, frow, fq);
}

template <bool OUT_BF16>
__global__ __launch_bounds__(256, 1) void gemm256w4_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

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

    // ---- staging: wave w stages pieces 4w .. 4w+3 (8 rows each) of every half-tile
    unsigned aoff[4], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        aoff[i] = (unsigned)((row * p.lda + c * 8) * 2);
        const int rl = row & 31, nperm = (row & ~31) + 8 * ((rl & 15) >> 2) + 4 * (rl >> 4) + (rl & 3);
        woff[i] = (unsigned)((nperm * p.ldw + c * 8) * 2);
    }
    const int64_t hiA = 128 * p.lda * 2, hiW = 128 * p.ldw * 2;
    const char* cA = (const char*)(p.A + m0 * p.lda);
    const char* cW = (const char*)(p.W + (int64_t)n0 * p.ldw);
    const int nk = p.K / 64;
    auto stage = [&](int slot, int kt) {
        if (kt >= nk) return;                                          // (uniform; the counted waits below assume nk >= 4 and handle the tail)
        char* dst = smem + (kt & 1) * W4_BUF + slot * W4_HT + wave * 4096;
        const bool isA = slot == W4_ALO || slot == W4_AHI;
        const char* bp = (isA ? cA : cW) + ((slot == W4_AHI) ? hiA : (slot == W4_BHI) ? hiW : 0) + (int64_t)kt * 128;
        asm volatile("" : "+s"(bp));
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(bp + (isA ? aoff[i] : woff[i]), dst + i * 1024);
    };

    // ---- fragment read offsets inside a half-tile: row = base + (lane & 15), chunk = ks*4 + (lane >> 4)
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + fq;
        a_off[ks] = (wr * 64 + frow) * 128 + ((c ^ (frow & 7)) << 4);     // + tm*2048
        b_off[ks] = (wc * 64 + frow) * 128 + ((c ^ (frow & 7)) << 4);     // + tn*2048
    }

    W4_CLOBBER_AGPRS();          // a[0:255] belong to the asm statements below
    w4_zero<0>();

    bf16x8 fa[4][2], fa2[4][2], fw0[4][2], fw1[4][2];      // A (current / next) and the two W fragment sets, [tile][ks]

#define W4_READ(DST, SLOT, KT, OFF) _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)      \
        DST[t_][ks] = *(const bf16x8*)(smem + ((KT) & 1) * W4_BUF + (SLOT) * W4_HT + OFF[ks] + t_ * 2048);
#define W4_QUAD(MH, NH, AF, WF) w4_quad<MH, NH, 0>(AF, WF);
#define W4_INTERLEAVE()
#define W4_EDGE(VM) do { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM) : "memory"); W4_BARRIER(); } while (0)

    // ---- prologue: k-tiles 0 and 1 (8 half-tiles fill the ring); A-lo(0) / W-lo(0) go to registers, then A-lo(2) takes A-lo(0)'s slot.
    // Steady-state queue at the top of Q0(t), oldest first: [W-hi(t) landed] A-hi(t), A-lo(t+1), W-lo(t+1), W-hi(t+1), A-hi(t+1), A-lo(t+2)
    stage(W4_ALO, 0); stage(W4_BLO, 0); stage(W4_BHI, 0); stage(W4_AHI, 0);
    stage(W4_ALO, 1); stage(W4_BLO, 1); stage(W4_BHI, 1); stage(W4_AHI, 1);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                   // A-lo(0), W-lo(0) landed (6 half-tiles younger)
    W4_BARRIER();
    W4_READ(fa, W4_ALO, 0, a_off)
    W4_READ(fw0, W4_BLO, 0, b_off)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_BARRIER();                                                       // every wave has A-lo(0) / W-lo(0) in registers
    stage(W4_ALO, 2);
    if (nk > 5) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // W-hi(0)
    W4_BARRIER();

    // a quadrant edge: the half-tile read in the coming quadrant has landed (6 younger half-tile DMAs = 24 instructions stay in
    // flight; near the end of K fewer were issued -> drain), and every wave is done with the half-tile read in the last quadrant
#define W4_EDGE_T(T) do { if ((T) + 3 < nk) W4_EDGE(24); else W4_EDGE(0); } while (0)
    for (int t = 0; t < nk; t += 2) {
        // ================= k-tile t (W sets: fw0 = W-lo, fw1 = W-hi)
        W4_READ(fw1, W4_BHI, t, b_off)
        stage(W4_BLO, t + 2);                                           // W-lo(t) was read in Q3(t-1) (or the prologue)
        W4_QUAD(0, 0, fa, fw0)
        W4_INTERLEAVE()
        W4_EDGE_T(t);                                                   // A-hi(t) landed; all waves read W-hi(t)
        W4_READ(fa2, W4_AHI, t, a_off)
        stage(W4_BHI, t + 2);
        W4_QUAD(0, 1, fa, fw1)
        W4_INTERLEAVE()
        W4_EDGE_T(t);                                                   // A-lo(t+1) landed; all waves read A-hi(t)
        if (t + 1 < nk) { W4_READ(fa, W4_ALO, t + 1, a_off) }
        stage(W4_AHI, t + 2);
        W4_QUAD(1, 1, fa2, fw1)
        W4_INTERLEAVE()
        W4_EDGE_T(t);                                                   // W-lo(t+1) landed; all waves read A-lo(t+1)
        if (t + 1 < nk) { W4_READ(fw1, W4_BLO, t + 1, b_off) }          // (fw1 is free after Q2; it becomes W-lo of k-tile t+1)
        stage(W4_ALO, t + 3);
        W4_QUAD(1, 0, fa2, fw0)
        W4_INTERLEAVE()
        W4_EDGE_T(t);                                                   // W-hi(t+1) landed; all waves read W-lo(t+1)
        if (t + 1 >= nk) break;
        // ================= k-tile t+1 (W sets swapped: fw1 = W-lo, fw0 = W-hi)
        W4_READ(fw0, W4_BHI, t + 1, b_off)
        stage(W4_BLO, t + 3);
        W4_QUAD(0, 0, fa, fw1)
        W4_INTERLEAVE()
        W4_EDGE_T(t + 1);
        W4_READ(fa2, W4_AHI, t + 1, a_off)
        stage(W4_BHI, t + 3);
        W4_QUAD(0, 1, fa, fw0)
        W4_INTERLEAVE()
        W4_EDGE_T(t + 1);
        if (t + 2 < nk) { W4_READ(fa, W4_ALO, t + 2, a_off) }
        stage(W4_AHI, t + 3);
        W4_QUAD(1, 1, fa2, fw0)
        W4_INTERLEAVE()
        W4_EDGE_T(t + 1);
        if (t + 2 < nk) { W4_READ(fw0, W4_BLO, t + 2, b_off) }
        stage(W4_ALO, t + 4);
        W4_QUAD(1, 0, fa2, fw1)
        W4_INTERLEAVE()
        W4_EDGE_T(t + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");      // (the last MFMAs have written their AGPRs before the epilogue reads them)

    // ---- epilogue: lane owns C[m][n .. n+7] for each tile pair
    w4_store_rows<OUT_BF16, 0, 0>(p, m0, n0, wr, wc, frow, fq);
#undef W4_READ
#undef W4_QUAD
#undef W4_INTERLEAVE
#undef W4_EDGE
#undef W4_EDGE_T
}


// =====================================================================================================================
// Persistent form (hh_set_tuning("gemm256", 5)): each workgroup walks its tiles (same XCD-aware virtual grid as gemm256d_kernel) and
// the half-tile DMA stream runs on ACROSS tile boundaries -- "k-tile nk" of a tile is k-tile 0 of the walk's next tile (K % 128 == 0
// keeps the ring parity), so there is no per-tile prologue and no drained tail except on the walk's last tile.  The accumulators are
// not zeroed: the first k-tile's ks = 0 MFMAs take C = 0.  Bias lives in LDS; the epilogue issues exactly STORES store instructions
// per wave, and the first six quadrant edges of the following tile (whose half-tiles were issued before those stores) wait with
// vmcnt(24 + STORES) (capped at the 6-bit maximum).
//
// ONE wave per SIMD issues strictly in order, ~4 cycles per instruction, and an MFMA leaves 8 of its 16 cycles to other instructions
// (MI355X_MICROARCH.md): everything that is not an MFMA must fit ~2 instructions per MFMA gap or the matrix core waits for the
// instruction stream itself (the first version of this kernel, with compiler-scheduled reads / address arithmetic / run-time
// branches on the tile position: 880 cycles per quadrant against 512 of MFMA work -- even with barriers, DMA and reads disabled).
// So the loop body is specialised at compile time for its place in the tile (first / middle / last iterations, with or without a
// following tile), every fragment read and LDS-DMA is an asm statement at a fixed place between the MFMAs (8 reads under MFMAs
// 1..15, 4 DMA pieces under MFMAs 17..23), LDS-DMA sources are scalar running pointers + four loop-invariant lane offsets (saddr
// form, no vector address arithmetic), and one s_waitcnt + s_barrier closes a quadrant.
#define W4_TS_TILES 8
__device__ unsigned long long g_gemm4_ts[512 * W4_TS_TILES * 7];

template <int BASE>
__device__ __forceinline__ void w4_mfma0(const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" :: "v"(w), "v"(a), "i"(BASE), "i"(BASE + 3));
}
__device__ __forceinline__ unsigned w4_lds_u32(const char* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
template <int IMM>
__device__ __forceinline__ void w4_ldsread(bf16x8& r, unsigned a) {
    static_assert(IMM >= 0 && IMM < 65536, "ds_read offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r) : "v"(a), "n"(IMM));
}
// one 16-B-per-lane LDS-DMA piece: global address = scalar base + lane offset, LDS destination = wave base + IMM (through M0)
template <int IMM>
__device__ __forceinline__ void w4_dma(unsigned voff, const char* sbase, unsigned wave_lds) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(wave_lds), "n"(IMM) : "memory", "scc", "m0");
}
// closes a quadrant: the eight fragment registers read under it are valid, the half-tile read in the next quadrant has landed
// (VM younger DMA / store instructions may stay in flight), every wave is done with the half-tile read in this one
template <int VM>
__device__ __forceinline__ void w4_edge(bf16x8 (&F)[4][2]) {
    asm volatile("s_waitcnt vmcnt(%8) lgkmcnt(0)\n\ts_barrier"
                 : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[2][0]), "+v"(F[2][1]), "+v"(F[3][0]), "+v"(F[3][1]) : "n"(VM) : "memory");
}
// 32 MFMAs AF x WF -> accumulator quadrant (MH, NH); under them the 8 reads of the half-tile at raddr + RIMM into RD and, if DMA_ON,
// the 4 pieces of the half-tile staged from `src` to LDS offset wave base + DIMM
// TR = 224 (tile of 224 rows, see gemm256w4p_kernel): the fourth 16-row block of the A-hi half does not exist -- its MFMAs are left out,
// the reads / DMA pieces keep their places
template <int TR, int MH, int NH, bool ZERO, int RIMM, int DIMM, bool DMA_ON, int I>
__device__ __forceinline__ void w4q(const bf16x8 (&AF)[4][2], const bf16x8 (&WF)[4][2], bf16x8 (&RD)[4][2], const unsigned (&raddr)[2],
                                    const unsigned (&voff)[4], const char* src, unsigned wave_lds) {
    constexpr int ks = I / 16, tn = (I / 4) % 4, tm = I % 4;
    constexpr int BASE = 4 * (((MH * 4 + tm) * 2 + NH) * 4 + tn);
    if constexpr (!(TR == 224 && MH == 1 && tm == 3)) {
        if constexpr (ZERO && ks == 0) w4_mfma0<BASE>(WF[tn][ks], AF[tm][ks]);
        else w4_mfma<BASE>(WF[tn][ks], AF[tm][ks]);
    }
    if constexpr (I % 2 == 1 && I < 16) {
        constexpr int j = I / 2;                                   // read j: fragment tile j >> 1, k-step j & 1
        w4_ldsread<RIMM + (j >> 1) * 2048>(RD[j >> 1][j & 1], raddr[j & 1]);
    }
    if constexpr (DMA_ON && I >= 16 && I < 24) {                   // M0 in one MFMA gap, the load in the next (the MFMA between them covers the M0 hazard)
        constexpr int i = (I - 16) / 2;
        if constexpr (I % 2 == 0) asm volatile("s_add_u32 m0, %0, %1" :: "s"(wave_lds), "n"(DIMM + i * 1024) : "scc", "m0");
        else asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(voff[i]), "s"(src) : "memory");
    }
    if constexpr (I + 1 < 32) w4q<TR, MH, NH, ZERO, RIMM, DIMM, DMA_ON, I + 1>(AF, WF, RD, raddr, voff, src, wave_lds);
}

// register-resident state of the k loop
struct W4State {
    bf16x8 fa[4][2], fa2[4][2], fw0[4][2], fw1[4][2];     // A (current / next) and the two W fragment sets, [tile][k-step]
    unsigned ra0[2], ra1[2], rb0[2], rb1[2];              // per-lane LDS addresses of the A / W fragment reads, LDS buffer 0 / 1
    unsigned aoff[4], woff[4];                            // per-lane global byte offsets of the four DMA pieces of an A / W half-tile
    const char *pAL, *pAH, *pWL, *pWH;                    // running DMA sources: the next k-tile each half-tile slot stages
    const char *nAL, *nAH, *nWL, *nWH;                    // k-tile 0 of the walk's next tile
    unsigned wave_lds;                                    // LDS address of this wave's 4 KB share of half-tile slot 0, buffer 0
};
// places of a two-k-tile loop iteration in a tile
enum { W4V_FIRST0 = 0,        // k-tiles 0, 1 of the walk's first tile
       W4V_NEXT0,             // k-tiles 0, 1 of a later tile: the previous epilogue's stores are younger than the half-tiles of the first six edges
       W4V_MID,               // steady state
       W4V_REBASE,            // k-tiles nk-4, nk-3, another tile follows: the staging pointers cross into it
       W4V_REBASE_LAST,       // k-tiles nk-4, nk-3 of the walk's last tile: the stream ends, the pipeline starts to drain
       W4V_TAIL_LAST };       // k-tiles nk-2, nk-1 of the walk's last tile: nothing staged, full waits
template <int V, int VM_ST>
__device__ __forceinline__ constexpr int w4_vm(int edge) {
    return V == W4V_NEXT0 ? (edge < 6 ? VM_ST : 24) : V == W4V_REBASE_LAST ? (edge < 4 ? 24 : 0) : V == W4V_TAIL_LAST ? 0 : 24;
}
#define W4_IMM(BUF, SLOT) ((BUF) * W4_BUF + (SLOT) * W4_HT)
template <int V, int VM_ST, int TR = 256>
__device__ __forceinline__ void w4_iter(W4State& s) {
    constexpr bool Z = V == W4V_FIRST0 || V == W4V_NEXT0;
    constexpr bool RB = V == W4V_REBASE || V == W4V_REBASE_LAST;
    constexpr bool ON = V != W4V_TAIL_LAST;                         // DMA of quadrants 0..6
    constexpr bool ON7 = V != W4V_TAIL_LAST && V != W4V_REBASE_LAST;
    // ================= k-tile t (even: LDS buffer 0; W sets: fw0 = W-lo, fw1 = W-hi)
    // Q0 = A-lo x W-lo   reads W-hi(t)      stages W-lo(t+2)
    w4q<TR, 0, 0, Z, W4_IMM(0, W4_BHI), W4_IMM(0, W4_BLO), ON, 0>(s.fa, s.fw0, s.fw1, s.rb0, s.woff, s.pWL, s.wave_lds);
    s.pWL += 128;
    w4_edge<w4_vm<V, VM_ST>(0)>(s.fw1);
    // Q1 = A-lo x W-hi   reads A-hi(t)      stages W-hi(t+2)
    w4q<TR, 0, 1, Z, W4_IMM(0, W4_AHI), W4_IMM(0, W4_BHI), ON, 0>(s.fa, s.fw1, s.fa2, s.ra0, s.woff, s.pWH, s.wave_lds);
    s.pWH += 128;
    w4_edge<w4_vm<V, VM_ST>(1)>(s.fa2);
    // Q2 = A-hi x W-hi   reads A-lo(t+1)    stages A-hi(t+2)
    w4q<TR, 1, 1, Z, W4_IMM(0, W4_ALO), W4_IMM(0, W4_AHI), ON, 0>(s.fa2, s.fw1, s.fa, s.ra1, s.aoff, s.pAH, s.wave_lds);
    s.pAH += 128;
    w4_edge<w4_vm<V, VM_ST>(2)>(s.fa);
    // Q3 = A-hi x W-lo   reads W-lo(t+1)    stages A-lo(t+3)
    w4q<TR, 1, 0, Z, W4_IMM(0, W4_BLO), W4_IMM(1, W4_ALO), ON, 0>(s.fa2, s.fw0, s.fw1, s.rb1, s.aoff, s.pAL, s.wave_lds);
    s.pAL = RB ? s.nAL : s.pAL + 128;
    w4_edge<w4_vm<V, VM_ST>(3)>(s.fw1);
    // ================= k-tile t+1 (odd: LDS buffer 1; W sets swapped: fw1 = W-lo, fw0 = W-hi)
    w4q<TR, 0, 0, false, W4_IMM(0, W4_BHI), W4_IMM(1, W4_BLO), ON, 0>(s.fa, s.fw1, s.fw0, s.rb1, s.woff, s.pWL, s.wave_lds);
    s.pWL = RB ? s.nWL : s.pWL + 128;
    w4_edge<w4_vm<V, VM_ST>(4)>(s.fw0);
    w4q<TR, 0, 1, false, W4_IMM(0, W4_AHI), W4_IMM(1, W4_BHI), ON, 0>(s.fa, s.fw0, s.fa2, s.ra1, s.woff, s.pWH, s.wave_lds);
    s.pWH = RB ? s.nWH : s.pWH + 128;
    w4_edge<w4_vm<V, VM_ST>(5)>(s.fa2);
    // (the reads of k-tile t+2 past the walk's end fetch stale LDS into registers nobody uses)
    w4q<TR, 1, 1, false, W4_IMM(0, W4_ALO), W4_IMM(1, W4_AHI), ON, 0>(s.fa2, s.fw0, s.fa, s.ra0, s.aoff, s.pAH, s.wave_lds);
    s.pAH = RB ? s.nAH : s.pAH + 128;
    w4_edge<w4_vm<V, VM_ST>(6)>(s.fa);
    w4q<TR, 1, 0, false, W4_IMM(0, W4_BLO), W4_IMM(0, W4_ALO), ON7, 0>(s.fa2, s.fw1, s.fw0, s.rb0, s.aoff, s.pAL, s.wave_lds);
    s.pAL += 128;
    w4_edge<w4_vm<V, VM_ST>(7)>(s.fw0);
}

// <= 32 rows x 32 columns of the row tail (see gemm256_tail_piece in gemm256.hip: the same 8 K-slices, summed in the same order; here
// each of the 4 waves runs slices w and w + 4)
template <bool OUT_BF16>
__device__ __forceinline__ void w4_tail_piece(const GemmParams& p, int piece, float* red, int tid) {
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int ncb = p.N / 32;
    const int n0 = (piece % ncb) * 32, r0 = (piece / ncb) * 32;
    const int kslice = p.K / 8;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int sl = wave + 4 * half;
        const int k_begin = sl * kslice;
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
        for (int r = 0; r < 16; ++r) red[(sl * 16 + r) * 64 + lane] = acc[r];
    }
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
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = {t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
            gemm_store4<OUT_BF16>(p.e, (char*)p.C, p.ldc, p.tail_m + r0 + l31, n0 + 8 * g + 4 * h, v);
        }
    }
    __syncthreads();
}

// quick_gelu (gemm_common.h) on four values with the multiplies / the add as packed fp32 operations (same roundings: bit-identical)
__device__ __forceinline__ f32x4 w4_quick_gelu4(f32x4 v) {
    f32x4 t = v * -2.4554669595930157f;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = __builtin_amdgcn_exp2f(t[q]);
    t = t + 1.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = __builtin_amdgcn_rcpf(t[q]);
    return v * t;
}

// ---- epilogue.  Lane (frow, fq) of wave (wr, wc) owns, for each of its 8 row groups IDX = mh*4 + tm (row m0 + mh*128 + wr*64 + tm*16 +
// frow), the columns n0 + wc*128 + 64 nh + 32 jj + 8 fq + [0, 8) for j = 2 nh + jj = 0..3.
// EPI: 0 bias, 1 bias + column scale, 2 bias + QuickGELU, 3 bias + ReLU, 4 bias + the producer side of the LayerNorm fold (include/hh.h:
// z = z_resid + value, bf16, + row statistics), 5 / 6 = 1 / 2 with the consumer side of the fold in front:
// value = rstd[row] * acc - rstd[row] * mean[row] * colsum[n] + bias[n].
template <int BASE16>
__device__ __forceinline__ void w4p_acc_group(f32x4 (&a)[4], f32x4 (&b)[4]) {
    a[0] = w4_acc_read<BASE16 + 0>();       b[0] = w4_acc_read<BASE16 + 4>();
    a[1] = w4_acc_read<BASE16 + 8>();       b[1] = w4_acc_read<BASE16 + 12>();
    a[2] = w4_acc_read<BASE16 + 16 + 0>();  b[2] = w4_acc_read<BASE16 + 16 + 4>();
    a[3] = w4_acc_read<BASE16 + 16 + 8>();  b[3] = w4_acc_read<BASE16 + 16 + 12>();
}
template <int EPI, int IDX>
__device__ __forceinline__ void w4p_finish_group(float sc, const f32x4 (&bias_v)[4][2], f32x4 (&a)[4], f32x4 (&b)[4]) {
    w4p_acc_group<16 * (IDX * 2)>(a, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] += bias_v[j][0]; b[j] += bias_v[j][1];
        if constexpr (EPI == 1) {
            a[j] *= sc; b[j] *= sc;                                    // (1 outside the scaled columns: exact)
        } else if constexpr (EPI == 2) {
            a[j] = w4_quick_gelu4(a[j]); b[j] = w4_quick_gelu4(b[j]);
        } else if constexpr (EPI == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[j][q] = fmaxf(a[j][q], 0.f); b[j][q] = fmaxf(b[j][q], 0.f); }
        }
    }
}
// consumer side of the LayerNorm fold: st = (rstd, -rstd * mean) of this lane's row; the column sums stay in registers, the bias is re-read
// from LDS per group (the two vectors together would not fit the register file beside the next tile's first fragments)
template <int EPI, int IDX>
__device__ __forceinline__ void w4p_finish_group_ln(float sc, const float* bias_l, const f32x4 (&cs_v)[4][2], f32x2 st, f32x4 (&a)[4], f32x4 (&b)[4]) {
    w4p_acc_group<16 * (IDX * 2)>(a, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 b0 = *(const f32x4*)(bias_l + (j & 1) * 32 + (j >> 1) * 64), b1 = *(const f32x4*)(bias_l + (j & 1) * 32 + (j >> 1) * 64 + 4);
        a[j] = a[j] * st[0] + (cs_v[j][0] * st[1] + b0);
        b[j] = b[j] * st[0] + (cs_v[j][1] * st[1] + b1);
        if constexpr (EPI == 5) {
            a[j] *= sc; b[j] *= sc;
        } else {
            a[j] = w4_quick_gelu4(a[j]); b[j] = w4_quick_gelu4(b[j]);
        }
    }
}
// per-tile constants of the LayerNorm-fold epilogues
struct W4Ln {
    const float* bias_l;          // LDS: bias + this lane's first column
    const float* stats;           // consumer: LDS, (rstd, -rstd * mean) of this lane's first row; row group IDX adds 2 * rowoff(IDX)
    f32x4 cs_v[4][2];             // consumer: column sums of this lane's 32 columns
};
// Consumer side: what a tile's epilogue needs from memory -- bias and column sums of its 256 columns, the statistics of its 256 rows: 4 KB --
// is fetched by LDS-DMA one tile AHEAD (one 1-KB piece per wave, issued when the walk's next tile is known, i.e. a whole main loop
// before its epilogue) into a double-buffered record in the LDS the full-length bias vector occupies in the other instantiations.
// With global loads at the start of the epilogue instead, every tile waited a memory latency for them (fc1: +92 us per launch).
#define W4_LNREC 4096          // [bias 256 f32 | colsum 256 f32 | stats 256 x (rstd, -rstd * mean)]
// first row of row group IDX = mh*4 + tm relative to the wave's first row: the wave's A-hi rows start 128 rows behind its A-lo rows (256-row
// tile: waves interleaved in 64-row blocks) or 64 (224-row tile: 112 contiguous rows per wave row, 7 groups)
template <int TR = 256>
__device__ __forceinline__ constexpr int w4_rowoff(int IDX) { return (IDX >> 2) * (TR == 224 ? 64 : 128) + (IDX & 3) * 16; }
template <int TR> __device__ __forceinline__ constexpr int w4_groups() { return TR == 224 ? 7 : 8; }

// bf16 output: group IDX goes through the wave's LDS scratch (16 rows x 256 B, 16-B chunks XOR-swizzled with the row: conflict-free both
// ways) and leaves as 4 stores of 4 rows x 256 B; the stores of group IDX - 1 are issued behind the arithmetic of group IDX.
template <int EPI, int IDX, int TR = 256>
__device__ __forceinline__ void w4p_store_tile(const GemmParams& p, bf16_t* cptr, int64_t rowbase, char* scr, int frow, int fq, int l15, int l4,
                                               float sc, const f32x4 (&bias_v)[4][2], const W4Ln& ln, f32x2 st, u32x4 (*prev)[4] = nullptr) {
    f32x4 a[4], b[4];
    f32x2 st_next = st;
    if constexpr (EPI == 5) {
        if constexpr (IDX + 1 < 8) st_next = *(const f32x2*)(ln.stats + 2 * w4_rowoff(IDX + 1));      // one group ahead (TR = 256 only)
        w4p_finish_group_ln<EPI, IDX>(sc, ln.bias_l, ln.cs_v, st, a, b);
    } else w4p_finish_group<EPI, IDX>(sc, bias_v, a, b);
    u32x4 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (u32x4){pack_bf16(a[j][0], a[j][1]), pack_bf16(a[j][2], a[j][3]), pack_bf16(b[j][0], b[j][1]), pack_bf16(b[j][2], b[j][3])};
    if constexpr (IDX > 0) {                                           // the previous group's rows (its LDS reads were issued before this group's arithmetic)
        const int64_t r0 = rowbase + w4_rowoff<TR>(IDX - 1) + l4;
#pragma unroll
        for (int q = 0; q < 4; ++q) W4_ST((u32x4*)(cptr + (r0 + 4 * q) * p.ldc), (*prev)[q]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *(u32x4*)(scr + frow * 256 + (((j * 4 + fq) ^ frow) << 4)) = o[j];      // chunk = 8 nh + 4 jj + fq = 4 j + fq
    u32x4 rd[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int r = 4 * q + l4; rd[q] = *(const u32x4*)(scr + r * 256 + ((l15 ^ r) << 4)); }
    if constexpr (IDX + 1 < w4_groups<TR>()) w4p_store_tile<EPI, IDX + 1, TR>(p, cptr, rowbase, scr, frow, fq, l15, l4, sc, bias_v, ln, st_next, &rd);
    else {
        const int64_t r0 = rowbase + w4_rowoff<TR>(IDX) + l4;
#pragma unroll
        for (int q = 0; q < 4; ++q) W4_ST((u32x4*)(cptr + (r0 + 4 * q) * p.ldc), rd[q]);
    }
}
// straight from the MFMA layout (a store instruction covers 16 rows x 64 B of bf16): fp32 output, and the QuickGELU epilogue, whose
// arithmetic (two transcendentals per value) is longer than even these slow stores -- the LDS round trip only adds to it
template <bool OUT_BF16, int EPI, int IDX>
__device__ __forceinline__ void w4p_store_rows_direct(const GemmParams& p, int64_t rowbase, float sc, const f32x4 (&bias_v)[4][2], const int64_t (&ccol)[4],
                                                      const W4Ln& ln, f32x2 st) {
    const int64_t orow = rowbase + w4_rowoff(IDX);
    f32x4 a[4], b[4];
    f32x2 st_next = st;
    if constexpr (EPI == 6) {
        if constexpr (IDX + 1 < 8) st_next = *(const f32x2*)(ln.stats + 2 * w4_rowoff(IDX + 1));
        w4p_finish_group_ln<EPI, IDX>(sc, ln.bias_l, ln.cs_v, st, a, b);
    } else w4p_finish_group<EPI, IDX>(sc, bias_v, a, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (OUT_BF16) {
            u32x4 o = {pack_bf16(a[j][0], a[j][1]), pack_bf16(a[j][2], a[j][3]), pack_bf16(b[j][0], b[j][1]), pack_bf16(b[j][2], b[j][3])};
            W4_ST((u32x4*)((bf16_t*)p.C + orow * p.ldc + ccol[j]), o);
        } else {
            W4_ST((f32x4*)((float*)p.C + orow * p.ldc + ccol[j]), a[j]);
            W4_ST((f32x4*)((float*)p.C + orow * p.ldc + ccol[j] + 4), b[j]);
        }
    }
    if constexpr (IDX + 1 < 8) w4p_store_rows_direct<OUT_BF16, EPI, IDX + 1>(p, rowbase, sc, bias_v, ccol, ln, st_next);
}

// ---- producer side of the LayerNorm fold (EPI 4; the attention output projections, model/LaviLa.py:281 -> :372,388).  For row group IDX the
// lane adds, in the MFMA layout, the fp32 residual row (z_resid: 8 loads per group, issued ONE GROUP AHEAD) to value = acc + bias, sums
// z and z^2 over its 32 columns (the four lanes of a row are 16 apart: two cross-lane adds), and sends z -- and C = bf16(value), unless
// skip_c -- through the wave's LDS scratch to 4-rows-x-256-B stores.  The (sum, sum of squares) of the wave's 128 columns go to
// z_partials[row][n0 / 128 + wc]; gemm.hip's finalize pass adds the N / 128 slices in a fixed order.
// RB (round 5, EPI 7): the residual rows are bf16 (the z = bf16(x) this block's norm3 consumed; the time branch's z1 only feeds norm1): ONE
// 16-byte load per 8 columns instead of two -- kept raw in xa[j] (bit pattern), widened where the group is consumed; xb is unused.
// RM: residual mode -- 0 = fp32 rows, 1 = bf16 rows (EPI 7), 2 = bf16 PAIR hi + lo (EPI 8: xa = 16 raw bytes of hi, xb = 16 raw bytes of lo)
template <int IDX, int TR = 256, int RM = 0>
__device__ __forceinline__ void w4p_zload(const float* xw0, const bf16_t* xl0, unsigned xlo, f32x4 (&xa)[4], f32x4 (&xb)[4], int64_t ldx) {
    if constexpr (RM != 0) {
        const bf16_t* r = (const bf16_t*)xw0 + (int64_t)w4_rowoff<TR>(IDX) * ldx + xlo;
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[j] = W4_LDX((const f32x4*)(r + (j & 1) * 32 + (j >> 1) * 64));      // 8 bf16 = 16 bytes, not yet widened
        if constexpr (RM == 2) {
            const bf16_t* q = xl0 + (int64_t)w4_rowoff<TR>(IDX) * ldx + xlo;
#pragma unroll
            for (int j = 0; j < 4; ++j) xb[j] = W4_LDX((const f32x4*)(q + (j & 1) * 32 + (j >> 1) * 64));
        }
    } else {
        const float* r = xw0 + (int64_t)w4_rowoff<TR>(IDX) * ldx + xlo;       // (wave-uniform base + the lane's 32-bit offset: see the write-back below)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xa[j] = W4_LDX((const f32x4*)(r + (j & 1) * 32 + (j >> 1) * 64));
            xb[j] = W4_LDX((const f32x4*)(r + (j & 1) * 32 + (j >> 1) * 64 + 4));
        }
    }
}
__device__ __forceinline__ void w4p_lds_rows(char* scr, int frow, int fq, int l15, int l4, const f32x4 (&a)[4], const f32x4 (&b)[4], u32x4 (&rd)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32x4 o = {pack_bf16(a[j][0], a[j][1]), pack_bf16(a[j][2], a[j][3]), pack_bf16(b[j][0], b[j][1]), pack_bf16(b[j][2], b[j][3])};
        *(u32x4*)(scr + frow * 256 + (((j * 4 + fq) ^ frow) << 4)) = o;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int r = 4 * q + l4; rd[q] = *(const u32x4*)(scr + r * 256 + ((l15 ^ r) << 4)); }
}
template <int IDX, int TR = 256, int RM = 0>
__device__ __forceinline__ void w4p_store_tile_z(const GemmParams& p, bf16_t* cw0, bf16_t* zw0, float* part0, char* scr,
                                                 int frow, int fq, int l15, int l4, const float* bias_l, float* xw0, bf16_t* xl0, unsigned xoff, unsigned xlo, unsigned coff,
                                                 unsigned zoff, f32x4 (&xa)[4], f32x4 (&xb)[4], f32x4 (&xna)[4], f32x4 (&xnb)[4]) {
    constexpr bool RB = RM != 0;
    // cw0 / zw0 / xw0 / part0: wave-uniform pointers to (first row of the wave's 64, its first column) of C / z / the residual / the
    // statistics slot; coff / zoff / xoff (store phase: row l4, 16-B piece l15) and xlo (MFMA layout: row frow, columns 8 fq) are the
    // lane's 32-bit element offsets -- four registers instead of four 64-bit per-lane pointers (the kernel sits at the 256-register limit)
    // (xa, xb): residual rows of group IDX; (xna, xnb): group IDX + 1, already requested.  Group IDX + 2 is requested as soon as group
    // IDX's registers are free (below): TWO groups (16 loads, 16 KB per wave) in flight on the same 64 registers that one-ahead
    // prefetching held anyway (measured: no faster than one group ahead -- all workgroups run this epilogue at once and together
    // sit at the HBM ceiling)
    f32x4 a[4], b[4];
    w4p_acc_group<16 * (IDX * 2)>(a, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] += *(const f32x4*)(bias_l + (j & 1) * 32 + (j >> 1) * 64);
        b[j] += *(const f32x4*)(bias_l + (j & 1) * 32 + (j >> 1) * 64 + 4);
    }
    u32x4 rd[4];
    if (!p.e.skip_c) {
        w4p_lds_rows(scr, frow, fq, l15, l4, a, b, rd);
        bf16_t* cw = cw0 + (int64_t)w4_rowoff<TR>(IDX) * p.ldc;
#pragma unroll
        for (int q = 0; q < 4; ++q) W4_ST((u32x4*)(cw + (int64_t)(4 * q) * p.ldc + coff), rd[q]);
    }
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (RB) {
            const u32x4 r = __builtin_bit_cast(u32x4, xa[j]);
            a[j] += (f32x4){bf16_lo_to_f32(r[0]), bf16_hi_to_f32(r[0]), bf16_lo_to_f32(r[1]), bf16_hi_to_f32(r[1])};
            b[j] += (f32x4){bf16_lo_to_f32(r[2]), bf16_hi_to_f32(r[2]), bf16_lo_to_f32(r[3]), bf16_hi_to_f32(r[3])};
            if constexpr (RM == 2) {                       // the low halves of the pair stream
                const u32x4 q = __builtin_bit_cast(u32x4, xb[j]);
                a[j] += (f32x4){bf16_lo_to_f32(q[0]), bf16_hi_to_f32(q[0]), bf16_lo_to_f32(q[1]), bf16_hi_to_f32(q[1])};
                b[j] += (f32x4){bf16_lo_to_f32(q[2]), bf16_hi_to_f32(q[2]), bf16_lo_to_f32(q[3]), bf16_hi_to_f32(q[3])};
            }
        } else { a[j] += xa[j]; b[j] += xb[j]; }
        s4 += a[j]; s4 += b[j];
        q4 += a[j] * a[j]; q4 += b[j] * b[j];
    }
    if constexpr (IDX + 2 < w4_groups<TR>()) w4p_zload<IDX + 2, TR, RM>(xw0, xl0, xlo, xa, xb, p.e.z_ldr);      // (xa / xb are consumed: re-targeted right away)
    if (!RB && p.e.z_update) {
        // x <- x + branch in place (the rows this lane loaded xa / xb from).  Straight from the MFMA layout a store instruction would
        // cover 16 rows x 4 pieces of 16 B at a 32-B stride (measured: fc2 + 228 us per launch, most of it these 64 stores per tile);
        // through the wave's LDS scratch -- one 64-column half (16 rows x 256 B, the bf16 image's geometry and swizzle) at a time -- the
        // rows leave as 4-rows-x-256-B stores of whole lines.
        // (wave-uniform base + ONE lane-invariant 32-bit offset: per-lane 64-bit row pointers would be hoisted out of the tile loop and,
        // live across the main loop, push the register allocator into the asm-owned AGPRs -- scripts/check_isa_hazards.py)
        float* xw = xw0 + (int64_t)w4_rowoff<TR>(IDX) * p.e.z_ldr;          // row 0 of the group, the wave's column 0 (uniform)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                *(f32x4*)(scr + frow * 256 + (((jj * 8 + 2 * fq) ^ frow) << 4)) = a[2 * h + jj];
                *(f32x4*)(scr + frow * 256 + (((jj * 8 + 2 * fq + 1) ^ frow) << 4)) = b[2 * h + jj];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * q + l4;
                const f32x4 v = *(const f32x4*)(scr + r * 256 + ((l15 ^ r) << 4));
                W4_STX((f32x4*)(xw + (int64_t)(4 * q) * p.e.z_ldr + h * 64 + xoff), v);
            }
        }
    }
    float sm = (s4[0] + s4[1]) + (s4[2] + s4[3]), sq = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);
    sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
    if (fq == 0) { const f32x2 o = {sm, sq}; *(f32x2*)(part0 + (int64_t)(w4_rowoff<TR>(IDX) + frow) * 2 * (p.N >> 7)) = o; }
    w4p_lds_rows(scr, frow, fq, l15, l4, a, b, rd);
    bf16_t* zw = zw0 + (int64_t)w4_rowoff<TR>(IDX) * p.e.z_ldc;
#pragma unroll
    for (int q = 0; q < 4; ++q) W4_ST((u32x4*)(zw + (int64_t)(4 * q) * p.e.z_ldc + zoff), rd[q]);
    if constexpr (RM == 2) {
        // pair stream: lo' = bf16(x' - hi'), same route (z_ldr == z_ldc: the two halves have one geometry); a / b become the remainders in place
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned h0 = pack_bf16(a[j][0], a[j][1]), h1 = pack_bf16(a[j][2], a[j][3]), h2 = pack_bf16(b[j][0], b[j][1]), h3 = pack_bf16(b[j][2], b[j][3]);
            a[j] -= (f32x4){bf16_lo_to_f32(h0), bf16_hi_to_f32(h0), bf16_lo_to_f32(h1), bf16_hi_to_f32(h1)};
            b[j] -= (f32x4){bf16_lo_to_f32(h2), bf16_hi_to_f32(h2), bf16_lo_to_f32(h3), bf16_hi_to_f32(h3)};
        }
        w4p_lds_rows(scr, frow, fq, l15, l4, a, b, rd);
        bf16_t* lw = xl0 + (int64_t)w4_rowoff<TR>(IDX) * p.e.z_ldc;
#pragma unroll
        for (int q = 0; q < 4; ++q) W4_ST((u32x4*)(lw + (int64_t)(4 * q) * p.e.z_ldc + zoff), rd[q]);
    }
    if constexpr (IDX + 1 < w4_groups<TR>()) w4p_store_tile_z<IDX + 1, TR, RM>(p, cw0, zw0, part0, scr, frow, fq, l15, l4, bias_l, xw0, xl0, xoff, xlo, coff, zoff, xna, xnb, xa, xb);
}

// Dynamic tile walk: per stream slot (runtime.cpp: hh_stream_slot) 8 per-XCD tile counters + the count of finished workgroups; the last
// workgroup of a launch zeroes them again (kernels of one stream never overlap).  Zero-initialised device memory, no allocation.
__device__ unsigned g_w4_tile_cnt[32][16];

//
// TR = 224: tiles of 224 rows x 256 columns.  A launch whose 256-row tiles do not fill whole rounds ends with most CUs idle -- M = 100384
// (196-token frames), N = 1024: 392 x 4 = 1568 tiles = 6.125 rounds on 256 CUs: 224 workgroups run 6 tiles, 32 run 7, and the seventh round
// (12 % of the launch) keeps one CU in eight busy.  The same rows in 224-row tiles are 448 x 4 = 1792 tiles = exactly 7 rounds of 7/8 the
// work.  A wave row owns 112 CONTIGUOUS rows (A-lo: its first 64, A-hi: the next 48): the lane offsets of the two A half-tiles stay
// equal, A-hi starts 64 rows in, the 16 A-hi rows per wave row that do not exist are staged from whatever follows (the next wave row's /
// the next tile's rows -- the caller guarantees >= 16 rows behind the last tile) and never multiplied: the MFMAs of row group 7 are
// left out of the instruction stream (w4q), the epilogue stops after group 6.  Per-element arithmetic is unchanged: bit-identical results.
template <bool OUT_BF16, int EPI, int TR = 256>
__global__ __launch_bounds__(256, 1) void gemm256w4p_kernel(GemmParams p) {
    static_assert(TR == 256 || (TR == 224 && OUT_BF16 && (EPI == 0 || EPI == 4)), "224-row tiles: bf16 bias-only and LayerNorm-fold producer epilogues");
    constexpr bool PRODUCER = EPI == 4 || EPI == 7 || EPI == 8;          // LayerNorm fold, producer side; 7 = bf16 residual rows (z_resid_dtype HH_BF16), 8 = bf16 pair stream (z_resid_lo)
    constexpr bool CONSUMER = EPI == 5 || EPI == 6;
    constexpr bool RB = EPI == 7 || EPI == 8;
    constexpr int RM = EPI == 8 ? 2 : EPI == 7 ? 1 : 0;
    constexpr int WROWS = TR == 224 ? 112 : 64;              // tile row of wave row 1's first A-lo row
    // global_store_dwordx4 per wave and tile in the epilogue that are younger than every load of it (checked in the ISA); the producer
    // side of the LayerNorm fold (EPI 4) waits for its residual loads group by group: only the last group's z stores are certain to trail
    constexpr int STORES = PRODUCER ? 4 : OUT_BF16 ? 4 * w4_groups<TR>() : 64;
    constexpr int VM_ST = 24 + STORES > 63 ? 63 : 24 + STORES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int GROUP = p.group_m;
    const int per = GROUP * p.Nt;
    const int nk = p.K / 64;                                  // even (K % 128 == 0) and >= 6, checked by the launcher
    // ---- tile walk.  XCD x (= blockIdx & 7) owns the m-tiles x, x + 8, ... and walks them in groups of GROUP m-tiles x all n-tiles
    // (n-major inside a group: GROUP workgroups share a W panel, consecutive ones an A panel); its tiles have the dense indices
    // 0 .. tiles_x - 1.  A workgroup's first tile is static (index blockIdx >> 3); every further one comes from the XCD's atomic counter
    // (p.dynamic), fetched one tile ahead: a workgroup that starts late or runs beside another stream's kernels simply takes fewer
    // tiles -- with equal static shares every delayed workgroup delayed the whole launch (the step's first GEMMs, which run beside the
    // text tower and the previous step's decoder, spanned up to 5x their own work).
    const int xcd = blockIdx.x & 7, nwx = gridDim.x >> 3;
    const int mtx = p.Mt > xcd ? (p.Mt - xcd + 7) / 8 : 0;    // m-tiles of this XCD
    const int tiles_x = mtx * p.Nt;
    const int gfull = mtx / GROUP, grem = mtx % GROUP;
    auto decode = [&](int d, int64_t& m0, int& n0) {
        int kg, nt_i, mi;
        if (d < gfull * per) { kg = d / per; const int r = d % per; nt_i = r / GROUP; mi = r % GROUP; }
        else { const int r = d - gfull * per; kg = gfull; nt_i = r / grem; mi = r % grem; }
        const int mt = xcd + 8 * (kg * GROUP + mi);
        m0 = (int64_t)(p.rev_m ? p.Mt - 1 - mt : mt) * TR;
        n0 = nt_i * 256;
    };
    unsigned* tcnt = g_w4_tile_cnt[p.tile_slot];
    // a workgroup is done with the counters once its last fetch has returned; the last one to say so resets them for the next launch
    auto finish = [&]() {
        if (p.dynamic && tid == 0) {
            const unsigned done = atomicAdd(tcnt + 8, 1u);
            if (done == gridDim.x - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) atomicExch(tcnt + i, 0u);
            }
        }
    };
    int64_t m0 = 0, nm0 = 0;
    int n0 = 0, nn0 = 0;
    int d = blockIdx.x >> 3;                                  // dense index of the current tile
    const bool any = d < tiles_x;
    if (any) decode(d, m0, n0);
    if (p.tail_rows > 0)
        for (int piece = blockIdx.x; piece < (p.N / 32) * ((p.tail_rows + 31) / 32); piece += gridDim.x) w4_tail_piece<OUT_BF16>(p, piece, (float*)smem, tid);
    if (!any) { finish(); return; }

    W4State s;
    // ---- staging: wave w stages pieces 4w .. 4w+3 (8 rows each) of every half-tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        const int arow = TR == 224 ? (row >> 6) * 112 + (row & 63) : row;      // half-tile row -> row of the tile (A-lo; A-hi: + hiA)
        s.aoff[i] = (unsigned)((arow * p.lda + c * 8) * 2);
        // W half-tile row R (its 64-row halves belong to wave columns wc = 0 / 1) holds column 128 wc + 64 NH + perm(R & 63) of the
        // tile: a wave owns 128 CONTIGUOUS columns, so that the epilogue can write 256-byte row segments
        const int rl = row & 31, nperm = (row >> 6) * 128 + (row & 32) + 8 * ((rl & 15) >> 2) + 4 * (rl >> 4) + (rl & 3);
        s.woff[i] = (unsigned)((nperm * p.ldw + c * 8) * 2);
    }
    const int64_t hiA = (TR == 224 ? 64 : 128) * p.lda * 2, hiW = 64 * p.ldw * 2;

    float* bias_s = (float*)(smem + 2 * W4_BUF);
    if constexpr (!CONSUMER) {
        for (int i = tid * 4; i < p.N; i += 256 * 4)
            *(f32x4*)(bias_s + i) = p.e.bias ? *(const f32x4*)(p.e.bias + i) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    W4_BARRIER();
    // consumer side of the LayerNorm fold: the epilogue record of tile (tm0, tn0) -> LDS record `par` (W4_LNREC), one 1-KB LDS-DMA piece per
    // wave: bias / column sums of the tile's 256 columns, statistics of its rows 0..127 / 128..255.  (s_nop 4: see the tile-counter atomic.)
    const unsigned lane16 = (unsigned)lane * 16u;
    auto ln_prefetch = [&](int par, int64_t tm0, int tn0) {
        const char* src = wave == 0 ? (const char*)(p.e.bias + tn0) : wave == 1 ? (const char*)(p.e.ln_colsum + tn0)
                                    : (const char*)(p.e.ln_stats + 2 * tm0) + (wave - 2) * 1024;
        unsigned dst = w4_lds_u32(smem + 2 * W4_BUF) + (unsigned)par * W4_LNREC + (unsigned)wave * 1024u;
        asm volatile("" : "+s"(src), "+s"(dst));
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane16), "s"(src), "s"(dst) : "memory", "m0");
    };
    if (p.skew_iters > 0) {
        // start skew: workgroups of an XCD begin `skew_iters` sleeps apart (P phases, or a 32-step ramp).  The tiles of a launch take equal
        // time, so without it every CU reaches its epilogue together; for the producer side of the LayerNorm fold (EPI 4: 256 KB of fp32
        // residual rows per tile) that is a 64 MB read burst per round during which no matrix core works
        const int it = p.skew_iters * (int)(p.skew_phases > 0 ? (blockIdx.x >> 3) % p.skew_phases : (blockIdx.x >> 3) & 31);
        for (int i = 0; i < it; ++i) __builtin_amdgcn_s_sleep(8);
    }

    int tile_i = 0;
    auto stamp = [&](int k) {
        if (p.debug_ts && tid == 0 && tile_i < W4_TS_TILES && blockIdx.x < 512) {
            unsigned long long* r = g_gemm4_ts + ((int)blockIdx.x * W4_TS_TILES + tile_i) * 7;
            r[k] = __builtin_amdgcn_s_memrealtime();
            if (k == 1 || k == 2) r[4 + k] = __builtin_amdgcn_s_memtime();
        }
    };

    // ---- fragment read addresses inside a half-tile: row = base + (lane & 15), chunk = ks*4 + (lane >> 4)
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + fq;
        s.ra0[ks] = w4_lds_u32(smem) + (wr * 64 + frow) * 128 + ((c ^ (frow & 7)) << 4);
        s.rb0[ks] = w4_lds_u32(smem) + (wc * 64 + frow) * 128 + ((c ^ (frow & 7)) << 4);
        s.ra1[ks] = s.ra0[ks] + W4_BUF; s.rb1[ks] = s.rb0[ks] + W4_BUF;
    }
    {
        unsigned wl = w4_lds_u32(smem) + (unsigned)wave * 4096u;
        asm volatile("" : "+s"(wl));
        s.wave_lds = wl;
    }

    W4_CLOBBER_AGPRS();          // a[0:255] belong to the asm statements

    // ---- prologue of the walk's first tile: k-tiles 0 and 1 (8 half-tiles fill the ring); A-lo(0) / W-lo(0) go to registers, then
    // A-lo(2) takes A-lo(0)'s place.  Queue at the top of Q0(t), oldest first:
    // [W-hi(t) landed] A-hi(t), A-lo(t+1), W-lo(t+1), W-hi(t+1), A-hi(t+1), A-lo(t+2)
    {
        const char* cA = (const char*)(p.A + m0 * p.lda);
        const char* cW = (const char*)(p.W + (int64_t)n0 * p.ldw);
        s.pAL = cA; s.pAH = cA + hiA; s.pWL = cW; s.pWH = cW + hiW;
        s.nAL = s.nAH = s.nWL = s.nWH = cA;
    }
#define W4_STAGE(BUF, SLOT, VOFF, PTR) do { w4_dma<W4_IMM(BUF, SLOT)>(s.VOFF[0], s.PTR, s.wave_lds); w4_dma<W4_IMM(BUF, SLOT) + 1024>(s.VOFF[1], s.PTR, s.wave_lds);   \
        w4_dma<W4_IMM(BUF, SLOT) + 2048>(s.VOFF[2], s.PTR, s.wave_lds); w4_dma<W4_IMM(BUF, SLOT) + 3072>(s.VOFF[3], s.PTR, s.wave_lds); s.PTR += 128; } while (0)
    if constexpr (CONSUMER) ln_prefetch(0, m0, n0);                      // (older than every DMA of the ring: the prologue's waits cover it)
    W4_STAGE(0, W4_ALO, aoff, pAL); W4_STAGE(0, W4_BLO, woff, pWL); W4_STAGE(0, W4_BHI, woff, pWH); W4_STAGE(0, W4_AHI, aoff, pAH);
    W4_STAGE(1, W4_ALO, aoff, pAL); W4_STAGE(1, W4_BLO, woff, pWL); W4_STAGE(1, W4_BHI, woff, pWH); W4_STAGE(1, W4_AHI, aoff, pAH);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                   // A-lo(0), W-lo(0) landed (6 half-tiles younger)
    W4_BARRIER();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j == 0) { w4_ldsread<W4_IMM(0, W4_ALO)>(s.fa[0][0], s.ra0[0]); w4_ldsread<W4_IMM(0, W4_BLO)>(s.fw0[0][0], s.rb0[0]); }
        if (j == 1) { w4_ldsread<W4_IMM(0, W4_ALO)>(s.fa[0][1], s.ra0[1]); w4_ldsread<W4_IMM(0, W4_BLO)>(s.fw0[0][1], s.rb0[1]); }
        if (j == 2) { w4_ldsread<W4_IMM(0, W4_ALO) + 2048>(s.fa[1][0], s.ra0[0]); w4_ldsread<W4_IMM(0, W4_BLO) + 2048>(s.fw0[1][0], s.rb0[0]); }
        if (j == 3) { w4_ldsread<W4_IMM(0, W4_ALO) + 2048>(s.fa[1][1], s.ra0[1]); w4_ldsread<W4_IMM(0, W4_BLO) + 2048>(s.fw0[1][1], s.rb0[1]); }
        if (j == 4) { w4_ldsread<W4_IMM(0, W4_ALO) + 4096>(s.fa[2][0], s.ra0[0]); w4_ldsread<W4_IMM(0, W4_BLO) + 4096>(s.fw0[2][0], s.rb0[0]); }
        if (j == 5) { w4_ldsread<W4_IMM(0, W4_ALO) + 4096>(s.fa[2][1], s.ra0[1]); w4_ldsread<W4_IMM(0, W4_BLO) + 4096>(s.fw0[2][1], s.rb0[1]); }
        if (j == 6) { w4_ldsread<W4_IMM(0, W4_ALO) + 6144>(s.fa[3][0], s.ra0[0]); w4_ldsread<W4_IMM(0, W4_BLO) + 6144>(s.fw0[3][0], s.rb0[0]); }
        if (j == 7) { w4_ldsread<W4_IMM(0, W4_ALO) + 6144>(s.fa[3][1], s.ra0[1]); w4_ldsread<W4_IMM(0, W4_BLO) + 6144>(s.fw0[3][1], s.rb0[1]); }
    }
    w4_edge<63>(s.fa);                                                   // (lgkmcnt(0): both sets are in registers; every wave has read A-lo(0) / W-lo(0))
    asm volatile("" : "+v"(s.fw0[0][0]), "+v"(s.fw0[0][1]), "+v"(s.fw0[1][0]), "+v"(s.fw0[1][1]), "+v"(s.fw0[2][0]), "+v"(s.fw0[2][1]), "+v"(s.fw0[3][0]), "+v"(s.fw0[3][1]));
    W4_STAGE(0, W4_ALO, aoff, pAL);                                      // A-lo(2)
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                   // W-hi(0)
    W4_BARRIER();

    bool first = true;
    for (;;) {
        stamp(0);
        // the walk's next tile: static stride, or (dynamic) one atomic fetch by lane 0 of wave 0 -- issued here, older than every DMA of
        // this tile, so the counted waits of the first iteration's last edges (whose half-tiles are younger) have retired it
        unsigned tk = 0;
        if (p.dynamic && tid == 0) {
            const unsigned zero = 0, one = 1;
            // (s_nop 4: the address SGPRs may have just come back from a VGPR spill lane -- v_readlane -> VMEM SGPR read needs 5 wait states, and
            // the hazard recogniser does not look inside inline asm: without it the LayerNorm-fold instantiation faulted on a stale pointer)
            asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(tk) : "v"(zero), "v"(one), "s"(tcnt + xcd) : "memory");
        }
        stamp(1);
        if (first) w4_iter<W4V_FIRST0, VM_ST, TR>(s); else w4_iter<W4V_NEXT0, VM_ST, TR>(s);
        for (int t = 2; t < nk - 4; t += 2) w4_iter<W4V_MID, VM_ST, TR>(s);
        int nd = d + nwx;
        if (p.dynamic) {                                       // publish the fetched index to the four waves (one barrier per tile)
            volatile __attribute__((address_space(3))) unsigned* slot = (volatile __attribute__((address_space(3))) unsigned*)(smem + 2 * W4_BUF + p.N * 4 + 4 * 4096 - 16);      // (the end of the epilogue scratch, idle between epilogues)
            if (tid == 0) { asm volatile("" : "+v"(tk)); *slot = tk; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_BARRIER();
            nd = __builtin_amdgcn_readfirstlane((int)*slot) + nwx;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const bool has_next = nd < tiles_x;
        if (has_next) {
            decode(nd, nm0, nn0);
            const char* nA = (const char*)(p.A + nm0 * p.lda);
            const char* nW = (const char*)(p.W + (int64_t)nn0 * p.ldw);
            s.nAL = nA; s.nAH = nA + hiA; s.nWL = nW; s.nWH = nW + hiW;
            // its epilogue record, into the other LDS record (this tile's is read in the epilogue below); it lands -- and every wave's
            // counted wait + barrier of the next edges publishes it -- a whole main loop before it is read
            if constexpr (CONSUMER) ln_prefetch((tile_i + 1) & 1, nm0, nn0);
        }
        asm volatile("" : "+s"(s.nAL), "+s"(s.nAH), "+s"(s.nWL), "+s"(s.nWH));
        if (has_next) { w4_iter<W4V_REBASE, VM_ST, TR>(s); w4_iter<W4V_MID, VM_ST, TR>(s); }
        else { w4_iter<W4V_REBASE_LAST, VM_ST, TR>(s); w4_iter<W4V_TAIL_LAST, VM_ST, TR>(s); }
        stamp(2);
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");      // the last MFMAs have written their AGPRs; the next tile's first fragments are in registers

        // ---- epilogue of tile (m0, n0): bias from LDS, then exactly STORES store instructions per wave
        {
            const hh_gemm_epilogue& e = p.e;
            const int nb = n0 + wc * 128 + 8 * fq;                     // lane's columns: nb + 64 nh + 32 jj + [0, 8)
            f32x4 bias_v[4][2];
            if constexpr (EPI < 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cj = nb + (j & 1) * 32 + (j >> 1) * 64;
                    bias_v[j][0] = *(const f32x4*)(bias_s + cj);
                    bias_v[j][1] = *(const f32x4*)(bias_s + cj + 4);
                }
            }
            float sc = 1.f;
            if constexpr (EPI == 1 || EPI == 5) {
                if (n0 + wc * 128 < e.colscale_cols) sc = e.colscale;  // colscale_cols % 128 == 0: the wave's 128 columns are all in or all out
            }
            stamp(3);
            W4Ln ln;
            f32x2 st0 = {1.f, 0.f};
            if constexpr (CONSUMER) {
                const float* rec = (const float*)(smem + 2 * W4_BUF + (tile_i & 1) * W4_LNREC);     // this tile's record (LDS)
                const int cl = wc * 128 + 8 * fq;                                   // the lane's first column inside the tile
                ln.bias_l = rec + cl;
                ln.stats = rec + 512 + 2 * (wr * 64 + frow);                         // (consumer epilogues: 256-row tiles only)
                st0 = *(const f32x2*)(ln.stats);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cj = cl + (j & 1) * 32 + (j >> 1) * 64;
                    ln.cs_v[j][0] = *(const f32x4*)(rec + 256 + cj);
                    ln.cs_v[j][1] = *(const f32x4*)(rec + 256 + cj + 4);
                }
            }
            if constexpr (PRODUCER) {
                char* scr = smem + 2 * W4_BUF + p.N * 4 + wave * 4096;
                const int64_t row0 = m0 + wr * WROWS;                     // first of the wave's rows; n0 + wc * 128: its first column (all uniform)
                const int col0 = n0 + wc * 128;
                float* xw0 = RB ? (float*)((bf16_t*)e.z_resid + row0 * e.z_ldr + col0) : (float*)e.z_resid + row0 * e.z_ldr + col0;      // (RB: really a bf16 pointer)
                bf16_t* cw0 = (bf16_t*)p.C + row0 * p.ldc + col0;
                bf16_t* zw0 = (bf16_t*)e.z_out + row0 * e.z_ldc + col0;
                float* part0 = e.z_partials + (row0 * (int64_t)(p.N >> 7) + ((n0 >> 7) + wc)) * 2;
                int lo = lane;
                asm volatile("" : "+v"(lo));                            // (recomputed per tile: not values to keep live across the main loop)
                const unsigned l4u = (unsigned)(lo >> 4), l15u = (unsigned)(lo & 15);
                const int l15 = (int)l15u, l4 = (int)l4u, frow_ = l15, fq_ = l4;        // (same bit fields of the lane in both layouts)
                const float* bias_z = bias_s + col0 + 8 * fq_;
                const unsigned xoff = l4u * (unsigned)e.z_ldr + 4u * l15u;                    // store phase, fp32 residual rows
                const unsigned coff = l4u * (unsigned)p.ldc + 8u * l15u, zoff = l4u * (unsigned)e.z_ldc + 8u * l15u;      // store phase, bf16 rows
                const unsigned xlo = l15u * (unsigned)e.z_ldr + 8u * l4u;                     // MFMA layout: row frow = lane & 15, columns 8 fq
                f32x4 xa[4], xb[4], xna[4], xnb[4];
                bf16_t* xl0 = RM == 2 ? (bf16_t*)e.z_resid_lo + row0 * e.z_ldr + col0 : nullptr;
                w4p_zload<0, TR, RM>(xw0, xl0, xlo, xa, xb, e.z_ldr);
                w4p_zload<1, TR, RM>(xw0, xl0, xlo, xna, xnb, e.z_ldr);
                w4p_store_tile_z<0, TR, RM>(p, cw0, zw0, part0, scr, frow_, fq_, l15, l4, bias_z, xw0, xl0, xoff, xlo, coff, zoff, xa, xb, xna, xnb);
            } else if constexpr (OUT_BF16 && EPI != 2 && EPI != 6) {
                // through this wave's 4 KB of LDS: a lane finishes 4 x 16 B of one row (MFMA layout), the wave then stores 4 rows x 256 B
                // per instruction.  A store instruction covering 16 rows x 64 B takes ~270 cycles on the CU's store path, 4 rows x 256 B
                // 66 (scripts/store_probe.hip) -- the epilogue was bound by exactly that.
                char* scr = smem + 2 * W4_BUF + p.N * 4 + wave * 4096;
                const int l15 = lane & 15, l4 = lane >> 4;
                const int ncol = n0 + wc * 128 + l15 * 8;              // this lane's 8 columns in the store phase
                bf16_t* cptr = (bf16_t*)p.C + gemm_ccol(e, ncol);
                w4p_store_tile<EPI, 0, TR>(p, cptr, m0 + wr * WROWS, scr, frow, fq, l15, l4, sc, bias_v, ln, st0);
            } else {
                int64_t ccol[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) ccol[j] = gemm_ccol(e, nb + (j & 1) * 32 + (j >> 1) * 64);
                w4p_store_rows_direct<OUT_BF16, EPI, 0>(p, m0 + wr * 64 + frow, sc, bias_v, ccol, ln, st0);
            }
        }
        stamp(4);
        ++tile_i;
        if (!has_next) break;
        d = nd; m0 = nm0; n0 = nn0;
        first = false;
    }
    finish();
#undef W4_STAGE
}

// LayerNorm fold (include/hh.h) on this kernel: bf16 outputs; producer = bias only (EPI 4), consumer = with the q column scale on whole
// 128-column halves (EPI 5) or QuickGELU (EPI 6).  Everything else with the fold runs on the generic 128x128 kernel.
static int g_w4_ln_ext = 1;         // hh_set_tuning("gemm_ln_w4", 0): keep LayerNorm-fold GEMMs off this kernel (A/B, debugging)
void hh_gemm256w4p_set_ln_ext(int v) { g_w4_ln_ext = v; }
int hh_gemm256w4p_ln_epi(const hh_gemm_epilogue& e) {
    if (e.c_dtype != HH_BF16) return -1;
    if (e.z_out) return (e.ln_stats == nullptr && e.act == HH_ACT_NONE && e.colscale_cols == 0 && e.c_block_stride == 0) ? (e.z_resid_lo ? 8 : e.z_resid_dtype == HH_BF16 ? 7 : 4) : -1;
    if (e.ln_stats) {                        // (callers check N >= 2048: the two epilogue records live where the N-float bias vector is)
        if (e.act == HH_ACT_NONE && e.colscale_cols > 0 && e.colscale_cols % 128 == 0) return 5;
        if (e.act == HH_ACT_QUICKGELU && e.colscale_cols == 0) return 6;
    }
    return -1;
}
bool hh_gemm256w4p_ln_ext_ok(const GemmParams& p, bool w4) {
    return g_w4_ln_ext && w4 && p.M >= 256 && hh_gemm256w4p_ln_epi(p.e) >= 0 && (p.e.ln_stats == nullptr || p.N * 4 >= 2 * W4_LNREC);
}

static bool g_w4_ts_last = false;
bool hh_gemm256w4_timeline_is_last() { return g_w4_ts_last; }
void hh_gemm256w4_timeline_mark(bool w4) { g_w4_ts_last = w4; }
int hh_gemm256w4_timeline(unsigned long long* out, int blocks) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm4_ts), sizeof(unsigned long long) * (size_t)blocks * W4_TS_TILES * 7);
    if (e != hipSuccess) { hh_set_error("hh_debug_gemm_timeline: %s", hipGetErrorString(e)); return HH_ERR_LAUNCH; }
    return HH_OK;
}

#define W4P_LDS(N) (2 * W4_BUF + (size_t)(N) * 4 + 4 * 4096)      // staging ring + bias vector + 4 KB of epilogue scratch per wave
int hh_gemm256w4p_launch(const GemmParams& p, int epi, unsigned pg, hipStream_t s) {
    static std::atomic<uint64_t> attr_mask{0};
    if (hh_attr_needed(attr_mask)) {
        hipError_t e = hipSuccess;
#define ATTRP(...) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm256w4p_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4P_LDS(4096))
        ATTRP(true, 0); ATTRP(false, 0); ATTRP(true, 1); ATTRP(false, 1); ATTRP(true, 2); ATTRP(false, 2); ATTRP(true, 3); ATTRP(false, 3);
        ATTRP(true, 4); ATTRP(true, 5); ATTRP(true, 6); ATTRP(true, 7); ATTRP(true, 8);
        ATTRP(true, 0, 224); ATTRP(true, 4, 224);
#undef ATTRP
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_gemm_bf16: cannot reserve %d B of LDS for the persistent 4-wave kernel: %s", (int)W4P_LDS(4096), hipGetErrorString(e));
        hh_attr_done(attr_mask);
    }
    const bool bf = p.e.c_dtype == HH_BF16;
#define LAUNCHP(BF, E) hipLaunchKernelGGL((gemm256w4p_kernel<BF, E>), dim3(pg), dim3(256), W4P_LDS(p.N), s, p)
    if (p.tile_rows == 224) {
        if (!bf || (epi != 0 && epi != 4)) { hh_set_error("hh_gemm_bf16: internal: 224-row tiles with an epilogue that has no such instantiation"); return HH_ERR_UNSUPPORTED; }
        if (epi == 0) hipLaunchKernelGGL((gemm256w4p_kernel<true, 0, 224>), dim3(pg), dim3(256), W4P_LDS(p.N), s, p);
        else hipLaunchKernelGGL((gemm256w4p_kernel<true, 4, 224>), dim3(pg), dim3(256), W4P_LDS(p.N), s, p);
        g_w4_ts_last = true;
        return hh_check_launch("hh_gemm_bf16(224x256 persistent, 4 waves)");
    }
    if (epi >= 4) {
        if (!bf) { hh_set_error("hh_gemm_bf16: the LayerNorm-fold epilogues of the persistent kernel write bf16"); return HH_ERR_UNSUPPORTED; }
        if (epi == 4) LAUNCHP(true, 4); else if (epi == 5) LAUNCHP(true, 5); else if (epi == 6) LAUNCHP(true, 6); else if (epi == 7) LAUNCHP(true, 7); else LAUNCHP(true, 8);
        g_w4_ts_last = true;
        return hh_check_launch("hh_gemm_bf16(256x256 persistent, 4 waves, LayerNorm fold)");
    }
    switch (epi * 2 + (bf ? 1 : 0)) {
        case 0: LAUNCHP(false, 0); break;
        case 1: LAUNCHP(true, 0); break;
        case 2: LAUNCHP(false, 1); break;
        case 3: LAUNCHP(true, 1); break;
        case 4: LAUNCHP(false, 2); break;
        case 5: LAUNCHP(true, 2); break;
        case 6: LAUNCHP(false, 3); break;
        default: LAUNCHP(true, 3); break;
    }
#undef LAUNCHP
    g_w4_ts_last = true;
    return hh_check_launch("hh_gemm_bf16(256x256 persistent, 4 waves)");
}

int hh_gemm256w4_launch(const GemmParams& p, unsigned grid, hipStream_t s) {
    static std::atomic<uint64_t> attr_mask{0};
    if (hh_attr_needed(attr_mask)) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm256w4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W4_BUF);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm256w4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W4_BUF);
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_gemm_bf16: cannot reserve %d B of LDS for the 4-wave kernel: %s", (int)(2 * W4_BUF), hipGetErrorString(e));
        hh_attr_done(attr_mask);
    }
    if (p.e.c_dtype == HH_BF16) hipLaunchKernelGGL((gemm256w4_kernel<true>), dim3(grid), dim3(256), 2 * W4_BUF, s, p);
    else hipLaunchKernelGGL((gemm256w4_kernel<false>), dim3(grid), dim3(256), 2 * W4_BUF, s, p);
    return hh_check_launch("hh_gemm_bf16(256x256, 4 waves)");
}

// EXPERIMENT (round 3, hh_set_tuning("gemm256", 4)): the 256x256x64 bf16 GEMM tile on FOUR waves of 128x128 instead of eight of
// 128x64 -- one wave per SIMD with all 512 registers (256 accumulators + three fragment sets).  Motivation: the step sits at the
// package power cap and a register-only MFMA loop holds twice the rate of the GEMM at that power, so the energy goes into moving
// operands; with 8 waves every k-tile costs 192 KB of LDS fragment reads per workgroup (each A fragment is read by 4 waves, each W
// fragment by 2), with 4 waves 128 KB.  Same LDS image, staging and epilogue conventions as gemm256.hip; one tile per workgroup (no
// persistence / continuity): it is measured against gemm256_kernel (tuning value 2), the 8-wave kernel of the same structure.
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

int hh_gemm256w4_launch(const GemmParams& p, unsigned grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)gemm256w4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W4_BUF);
        hipFuncSetAttribute((const void*)gemm256w4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W4_BUF);
        attr_done = true;
    }
    if (p.e.c_dtype == HH_BF16) hipLaunchKernelGGL((gemm256w4_kernel<true>), dim3(grid), dim3(256), 2 * W4_BUF, s, p);
    else hipLaunchKernelGGL((gemm256w4_kernel<false>), dim3(grid), dim3(256), 2 * W4_BUF, s, p);
    return hh_check_launch("hh_gemm_bf16(256x256, 4 waves)");
}

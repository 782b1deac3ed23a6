// Space attention on v_mfma_f32_32x32x16_bf16 (round 6): see the block comments below.  Both kernels stay within 256 registers per wave (two
// waves per SIMD), so hipcc selects the VGPR form of every MFMA: the score tiles must sit in arch VGPRs -- v_exp_f32 cannot read an AGPR.  (A
// 512-register one-wave-per-SIMD variant got the AGPR form: 16 v_accvgpr_read per half-step and a bunched schedule; -mllvm
// -amdgpu-mfma-vgpr-form=1 cured that, the variant lost for another reason -- see the persistent kernel's comment.)
#include "attn_space_dev.h"

HH_SPACE_REDO_COUNTER(g_space32_redo);

// ---- 32-query blocks on v_mfma_f32_32x32x16_bf16, software-pipelined INSIDE the wave (round 6) ----------------------
// Why: the joint-block kernel above runs a chunk as three serial phases per in-order wave (score MFMAs -> exponentials -> PV MFMAs) and a
// 16x16x32 MFMA holds the SIMD's vector issue for 8 of its 16 cycles (MI355X_MICROARCH.md, cycle constants): 36 MFMAs + 32 v_exp_f32 + 16
// v_cvt_pk per 64-query x 32-key chunk are 288 + 256 + 72 issue cycles against 576 cycles of matrix-core time -- issue-bound even when
// perfectly overlapped, and measured at ~2200 cycles per chunk and wave (MFMA busy 34 %).  Here
//   * every product is a 32x32x16 MFMA (8 of 32 cycles of issue): S^T[32 keys][32 queries] = K . Q^T (4 MFMAs over d = 64), the
//     probabilities stay on the lane of their query (16 registers = 16 keys), and P^T IS the B operand of O^T[32 d][32 queries] += V^T . P^T
//     with the key order of the contraction chosen to match the accumulator layout (registers 8m..8m+7 of lane-half hh = keys 16m + 4hh + 0..3
//     and 16m + 8 + 4hh + 0..3: two ds_read_b64_tr_b16 per V^T fragment); row sums = a third product against an all-ones operand;
//   * a wave owns TWO 32-query blocks a, b whose chains QK -> exp -> PV run half a chunk apart, so that each half-step holds 10 independent
//     MFMAs (QK of one block + PV of the same block's previous chunk: 320 cycles) and the 16 exponentials + 8 conversions of the OTHER
//     block (128 + 36 issue cycles + 80 of the MFMAs' own): the exponentials hide under the matrix core inside ONE wave, no partner needed;
//   * no reference maximum: scores are base-2 logits and fp32 / bf16 carry them from 2^-126 to 2^127, the softmax is shift-invariant, so
//     P = exp2(s) is used as it is and a block whose row sum leaves [2^-100, 2^100] (a row maximum outside about [-100, +92], i.e. natural
//     logits outside [-69, +64]) is redone on the running-maximum path (space16_chunk) -- the accumulator initialiser of a 32x32 product
//     would cost 16 registers per block that the pipeline needs for the second block;
//   * same LDS images and swizzles as above (both 32x32 fragment reads are conflict-free on them: DESIGN.md 4.2), whole chunk loop unrolled so
//     that every fragment address is a lane constant + immediate;
//   * the output leaves through LDS (the K tile is dead after the last chunk): O^T has the query on the lane and 4 consecutive d per register
//     group, rows are rebuilt with ds_write_b64 / ds_read_b128 and stored as 8 full 128-B lines per instruction.
#define NW32 4
__device__ __forceinline__ void sp32_qk(const char* const (&kb)[4], int ci, const bf16x8 (&q)[4], f32x16& s) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(kb[ks] + ci * 4096);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, q[ks], ks == 0 ? z : s, 0, 0, 0);
    }
}
__device__ __forceinline__ void sp32_exp(const f32x16& s, bf16x8 (&p)[2]) {
    float e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(s[i]);
#pragma unroll
    for (int m = 0; m < 2; ++m)
        p[m] = (bf16x8){(bf16_t)e[8 * m], (bf16_t)e[8 * m + 1], (bf16_t)e[8 * m + 2], (bf16_t)e[8 * m + 3],
                        (bf16_t)e[8 * m + 4], (bf16_t)e[8 * m + 5], (bf16_t)e[8 * m + 6], (bf16_t)e[8 * m + 7]};
}
// the CLS chunk: only its row 0 (accumulator register 0 of lanes 0..31) is a key
__device__ __forceinline__ void sp32_exp_cls(const f32x16& s, bf16x8 (&p)[2], int lane) {
    const float e = lane < 32 ? __builtin_amdgcn_exp2f(s[0]) : 0.f;
    const bf16_t zb = (bf16_t)0.f;
    p[0] = (bf16x8){(bf16_t)e, zb, zb, zb, zb, zb, zb, zb};
    p[1] = (bf16x8){zb, zb, zb, zb, zb, zb, zb, zb};
}
template <bool CLS>
__device__ __forceinline__ void sp32_pv(const char* const (&vb)[2], int ci, const bf16x8 (&p)[2], f32x16 (&o)[2], f32x16& ol) {
    const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
    // the row-sum products first: their operands are in registers when the half-step starts, so they cover the LDS latency of the fragments
#pragma unroll
    for (int m = 0; m < (CLS ? 1 : 2); ++m) ol = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, p[m], ol, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < (CLS ? 1 : 2); ++m) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const bf16x4 a0 = lds_tr4(vb[dt] + ci * 4096 + m * 2048), a1 = lds_tr4(vb[dt] + ci * 4096 + m * 2048 + 1024);
            const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, p[m], o[dt], 0, 0, 0);
        }
    }
}
// One steady-state half-step: row sums + PV of block Z (chunk cz, probabilities from the previous half-step), scores of block Z's NEXT chunk
// (10 MFMAs = 320 cycles of the matrix core), and under them the 16 exponentials + 8 conversions of block Y.  The group barriers pin the
// interleave the scheduler would otherwise bunch (all exponentials after the first MFMA: 7 per gap): per MFMA gap 2 v_exp_f32 + 1
// v_cvt_pk_bf16_f32 (= 8 + 16 + 4 issue cycles of the gap's 32), the fragment reads in the first gaps.
// K fragments: chunk c feeds the scores of block a (half-step B of chunk c - 1) and of block b (half-step A of chunk c), so they are read
// ONCE, a whole half-step before their first use (LOADK: half-step A of chunk c - 1 reads chunk c + ... see the callers) and kept for both
// -- 4 KB less LDS traffic per chunk and wave and no LDS latency in front of the score chain.
__device__ __forceinline__ void sp32_load_k(const char* const (&kb)[4], int ci, bf16x8 (&kf)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const bf16x8*)(kb[ks] + ci * 4096);
}
__device__ __forceinline__ void sp32_qk_regs(const bf16x8 (&kf)[4], const bf16x8 (&q)[4], f32x16& s) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], q[ks], ks == 0 ? z : s, 0, 0, 0);
}
template <bool LOADK>
__device__ __forceinline__ void sp32_half_step(const char* const (&kb)[4], const char* const (&vb)[2], int cz, const bf16x8 (&kf)[4], int cn, bf16x8 (&kfn)[4],
                                               const bf16x8 (&qz)[4], f32x16& sz, const bf16x8 (&pz)[2], f32x16 (&oz)[2], f32x16& olz, const f32x16& sy,
                                               bf16x8 (&py)[2], bool has_pv) {
    // MFMA order: the two row-sum products, the score chain (done four MFMAs = 128 cycles before the half-step ends, so the next half-step's
    // first exponentials do not wait for it), the four PV products (their V^T fragments are read under the first six MFMAs)
    const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
    if (has_pv) {
        olz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pz[0], olz, 0, 0, 0);
        olz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pz[1], olz, 0, 0, 0);
    }
    sp32_qk_regs(kf, qz, sz);
    if (has_pv) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x4 a0 = lds_tr4(vb[dt] + cz * 4096 + m * 2048), a1 = lds_tr4(vb[dt] + cz * 4096 + m * 2048 + 1024);
                const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                oz[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pz[m], oz[dt], 0, 0, 0);
            }
    }
    if (LOADK) sp32_load_k(kb, cn, kfn);
    sp32_exp(sy, py);
    asm volatile("" : "+v"(py[0]), "+v"(py[1]));     // the probabilities are made HERE (LLVM otherwise sinks them to their use in the next half-step)
    if (LOADK) asm volatile("" : "+v"(kfn[0]), "+v"(kfn[1]), "+v"(kfn[2]), "+v"(kfn[3]));       // ... and so are the next chunk's K fragments
#ifndef HH_SP32_NO_GROUPS
    if (has_pv) {
#pragma unroll
        for (int g = 0; g < 10; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // MFMA
            if (g < 2 || (LOADK && g == 6)) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);         // DS reads: V^T under MFMAs 0-1, the next K under MFMA 6
            if (g >= 1 && g <= 8) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);     // VALU: the conversion of the previous gap's pair
            if (g < 8) __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);         // TRANS
        }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
}
// O^T of one 32-query block -> the wave's LDS rows (row q: 128 B, 16-B chunk c at position c ^ (q & 7)), normalised, bf16
__device__ __forceinline__ void sp32_o_to_lds(char* rows, const f32x16 (&o)[2], float l, int r, int hh) {
    const float inv = 1.f / l;
    char* row = rows + r * 128 + 8 * hh;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const u32x2 w = {pack_bf16(o[dt][4 * k4] * inv, o[dt][4 * k4 + 1] * inv), pack_bf16(o[dt][4 * k4 + 2] * inv, o[dt][4 * k4 + 3] * inv)};
            *(u32x2*)(row + (((4 * dt + k4) ^ (r & 7)) << 4)) = w;
        }
}

template <int NCH>
__global__ __launch_bounds__(64 * NW32, 2) void space_attn32_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                   float* __restrict__ cls_partial, int B, int T, int heads, int dbg, int layout) {
    constexpr int n = NCH * 32, KP = n + 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + (size_t)KP * 128;
    float* scratch = (float*)(smem + (size_t)KP * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = heads * 64;
    const int N = 1 + T * n;
    const int rev = layout >> 1;                       // (bit 1, HH_QKV_WALK_REVERSE: the problems last to first)
    layout &= 1;
    const int64_t ld = layout ? 64 : 3 * (int64_t)D;
    const int64_t hs = layout ? (int64_t)B * N * 64 : 64, ws = (int64_t)heads * hs;
    int bid = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int head = bid % heads; bid /= heads;
    const int f = bid % T;
    const int b = bid / T;
    const bf16_t* base = qkv + (int64_t)b * N * ld + head * hs;
    const bf16_t* q_ptr = base + (int64_t)(1 + f * n) * ld;
    const int r = lane & 31, hh = lane >> 5;
    bf16x8 qa[4], qb[4];
    auto load_q = [&](int q0) {
        const bf16_t* ra = q_ptr + (int64_t)(q0 + r) * ld + 8 * hh;
        const bf16_t* rb = ra + 32 * ld;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qa[ks] = *(const bf16x8*)(ra + 16 * ks); qb[ks] = *(const bf16x8*)(rb + 16 * ks); }
    };
    const int q0 = wave * 64;                          // this wave's 64 queries (n <= 256: at most one group per wave)
    const bool active = q0 < n;
    // debug mode 3: cls_partial is a [workgroups, 8] uint64 buffer of s_memtime stamps of wave 0 (start / staged / chunks done / K, V free / done)
    unsigned long long* stamps = dbg == 3 ? (unsigned long long*)cls_partial + (int64_t)blockIdx.x * 8 : nullptr;
    if (dbg == 3) { cls_partial = nullptr; if (tid == 0) stamps[0] = __builtin_readcyclecounter(); }
    if (active) load_q(q0);                            // in flight together with the K / V staging
    if (dbg != 2) space_stage<NW32>(Ks, Vs, base, q_ptr, ld, ws, n, KP, lane, wave);
    else for (int i = tid * 16; i < KP * 256; i += 64 * NW32 * 16) *(u32x4*)(smem + i) = (u32x4){0u, 0u, 0u, 0u};     // (compute only: finite operands)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (dbg == 3 && tid == 0) stamps[1] = __builtin_readcyclecounter();
    if (dbg == 1) {                                    // debug: memory traffic only
        if (active) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                *(bf16x8*)(out + ((int64_t)b * N + 1 + f * n + q0 + r) * D + head * 64 + 16 * ks + 8 * hh) = qa[ks];
                *(bf16x8*)(out + ((int64_t)b * N + 1 + f * n + q0 + 32 + r) * D + head * 64 + 16 * ks + 8 * hh) = qb[ks];
            }
        }
        return;
    }
    // lane-constant fragment addresses of chunk 0 (the swizzle keys use row bits 0-3 / 1 only: the same for every chunk)
    const int kz = (r & 7) ^ ((r >> 3) & 1);
    const char* kb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kb[ks] = Ks + r * 128 + (((2 * ks + hh) ^ kz) << 4);
    const int i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, e16 = (lane >> 4) & 1, vz = ((q4 >> 1) & 1) << 2;
    const char* vb[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) vb[dt] = Vs + (4 * hh + q4) * 128 + (((4 * dt + 2 * e16 + (pp >> 1)) ^ vz) << 4) + 8 * (pp & 1);
    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 oa[2] = {z16, z16}, ob[2] = {z16, z16};
    float la = 1.f, lb = 1.f;
    bool bad_a = false, bad_b = false;
    if (active) {
        f32x16 sa, sb, ola = z16, olb = z16;
        bf16x8 pa[2], pb[2];
        bf16x8 kf[2][4];                               // K fragments of chunk c in kf[c & 1]
        sp32_load_k(kb, 0, kf[0]);
        sp32_qk_regs(kf[0], qa, sa);
#pragma unroll
        for (int ci = 0; ci < NCH; ++ci) {
            // half-step A: PV of b (chunk ci - 1) + scores of b (chunk ci) | probabilities of a (chunk ci) | reads K of chunk ci + 1
            sp32_half_step<true>(kb, vb, ci - 1, kf[ci & 1], ci + 1, kf[(ci + 1) & 1], qb, sb, pb, ob, olb, sa, pa, ci > 0);
            // half-step B: PV of a (chunk ci) + scores of a (chunk ci + 1; the last one is the CLS chunk) | probabilities of b (chunk ci)
            sp32_half_step<false>(kb, vb, ci, kf[(ci + 1) & 1], 0, kf[ci & 1], qa, sa, pa, oa, ola, sb, pb, true);
        }
        sp32_pv<false>(vb, NCH - 1, pb, ob, olb);
        sp32_qk_regs(kf[NCH & 1], qb, sb);
        sp32_exp_cls(sa, pa, lane);
        __builtin_amdgcn_sched_barrier(0);
        sp32_exp_cls(sb, pb, lane);
        sp32_pv<true>(vb, NCH, pa, oa, ola);
        sp32_pv<true>(vb, NCH, pb, ob, olb);
        if (dbg == 3 && tid == 0) stamps[2] = __builtin_readcyclecounter();
        // ---- a row sum outside [2^-100, 2^100]: that block is redone on the running-maximum path (16-query routines, K / V still resident)
        la = ola[0]; lb = olb[0];
        bad_a = __builtin_amdgcn_ballot_w64(!(la >= 7.9e-31f && la <= 1.2e30f)) != 0;
        bad_b = __builtin_amdgcn_ballot_w64(!(lb >= 7.9e-31f && lb <= 1.2e30f)) != 0;
        if (bad_a || bad_b) {
            const int c = lane & 15, g = lane >> 4, nt = (n >> 4) + 1;
#pragma unroll 1
            for (int sub = 0; sub < 4; ++sub) {
                if (!(sub < 2 ? bad_a : bad_b)) continue;
                if ((sub & 1) == 0) HH_SPACE_REDO_NOTE(g_space32_redo, lane);          // (one count per 32-query block)
                const int row0 = q0 + 16 * sub;
                const bf16_t* qrow = q_ptr + (int64_t)(row0 + c) * ld + 8 * g;
                bf16x8 q16[2];
                q16[0] = *(const bf16x8*)(qrow);
                q16[1] = *(const bf16x8*)(qrow + 32);
                f32x4 o2[4] = {z4, z4, z4, z4};
                float m_run = -INFINITY, l_run = 0.f;
                for (int t = 0; t + 1 < nt; ++t) space16_chunk<1, false>(Ks, Vs, q16, t, lane, o2, m_run, l_run);
                space16_chunk<1, true>(Ks, Vs, q16, nt - 1, lane, o2, m_run, l_run);
                space_store_block(o2, l_run, out + ((int64_t)b * N + 1 + f * n + row0 + c) * D + head * 64 + 16 * g, c, D);
            }
        }
    }
    // ---- the CLS query's partial over this frame's keys needs K / V: it runs BEFORE the rows go through the K tile
    if (cls_partial != nullptr) {
        bf16x8 qc[2];
        qc[0] = *(const bf16x8*)(base + 8 * (lane >> 4));
        qc[1] = *(const bf16x8*)(base + 8 * (lane >> 4) + 32);
        space16_cls_wave<NW32>(Ks, Vs, scratch, qc, n, f == 0, lane, wave);
    }
    __syncthreads();                                   // every wave is done with K / V
    if (dbg == 3 && tid == 0) stamps[3] = __builtin_readcyclecounter();
    if (cls_partial != nullptr && tid < 64)
        space16_cls_merge<NW32>(scratch, cls_partial + (((int64_t)b * heads + head) * T + f) * CLS_REC, tid);
    if (!active) return;
    char* rows = Ks + wave * 8192;                     // the wave's 64 output rows, in the dead K tile (KP * 128 >= 4 * 8192 needs n >= 224;
                                                       // smaller n: fewer active waves, same bound per wave: q0 + 64 <= n <= KP)
    if (!bad_a) sp32_o_to_lds(rows, oa, la, r, hh);
    if (!bad_b) sp32_o_to_lds(rows + 32 * 128, ob, lb, r, hh);
    // 8 rows x 128 B per instruction: lane l takes chunk l & 7 of row 8 t + (l >> 3)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if ((t < 4 ? bad_a : bad_b)) continue;         // (wave-uniform)
        const int row = 8 * t + (lane >> 3);
        const u32x4 v = *(const u32x4*)(rows + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
        *(u32x4*)(out + ((int64_t)b * N + 1 + f * n + q0 + row) * D + head * 64 + 8 * (lane & 7)) = v;
    }
    if (dbg == 3 && tid == 0) { stamps[4] = __builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamps[5] = __builtin_readcyclecounter(); }
}


// ---- the same chunk pipeline in a PERSISTENT, WAVE-SPECIALISED workgroup, one per CU (round 6, second step).  Timeline of the kernel above
// (debug mode 3, B = 32): a workgroup lives 19 us -- 9.4 us waiting for its K / V / Q, 4.8-6.3 us in the chunk loop, 3.3-4.4 us until its stores
// are taken -- i.e. with two workgroups per CU nothing is in flight for a CU while its workgroups compute, and the chip's memory system (the
// kernel's real bound: 1074 MB per call) idles a quarter of the time.  A first persistent form (4 waves, one per SIMD, 512 registers, the next
// problem's K / V / Q requested into parked registers before the chunk loop) was built and measured at 315-367 us: a wave that issues 26
// loads + 8 stores into a saturated memory system is BLOCKED in the issue of those instructions for about the memory time of a problem,
// and with one wave per SIMD nothing else runs -- memory time and compute time add up (10.8 us per problem).  Hence two roles:
//   * waves 0-3 (one per SIMD) only compute: Q fragments from LDS, the chunk loop above, the running-maximum redo, the CLS partial, their 64
//     output rows into LDS -- no vector-memory instruction on the normal path;
//   * waves 4-7 (the second wave of each SIMD) only move data: while problem i is computed they store the rows of problem i - 1 from LDS,
//     request K / V / Q of problem i + 1 into their registers (26 x 16 B per lane, 104 KB in flight per CU), and between the two workgroup
//     barriers of a problem boundary write them to LDS (the image the LDS-DMA of the kernels above writes: lane l of piece pc at pc * 1024 +
//     16 l) -- plain loads and ds_write, so hipcc's own counted waits apply (an LDS-DMA in flight would turn them into vmcnt(0)).
// Two barriers per problem: A = "K / V / Q of problem i are dead" (loaders then overwrite them, compute waves write their rows), B = "K / V / Q
// of problem i + 1 are visible, rows of problem i complete".  (A third barrier that frees the Q tile right after the fragment reads, so that
// the next Q rows -- 32 of the 104 KB -- go to LDS under the chunk loop, was measured SLOWER: 11 300 instead of 9 500 cycles per problem.)
#define NW32P 8
template <int NCH>
__global__ __launch_bounds__(64 * NW32P, 1) void space_attn32p_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                     float* __restrict__ cls_partial, int B, int T, int heads, int dbg, int layout) {
    constexpr int n = NCH * 32, KP = n + 32, PIECES = KP / 8, PPW = PIECES / 4, QPW = n / 8 / 4;
    static_assert(n == 256, "four compute waves x 64 queries");
    static_assert(PIECES % 4 == 0 && (n / 8) % 4 == 0, "every loader wave moves the same number of pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + (size_t)KP * 128;
    char* Qs = smem + (size_t)KP * 256;
    char* rows_all = Qs + (size_t)n * 128;
    float* scratch = (float*)(rows_all + 4 * 8192);                 // two sets of 4 CLS records (the merge of problem i - 1 runs while problem i is computed)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = heads * 64;
    const int N = 1 + T * n;
    const int rev = layout >> 1;                       // (bit 1, HH_QKV_WALK_REVERSE: the problems last to first)
    layout &= 1;
    const int64_t ld = layout ? 64 : 3 * (int64_t)D;
    const int64_t hs = layout ? (int64_t)B * N * 64 : 64, ws = (int64_t)heads * hs;
    const int P = B * T * heads;
    const int first = blockIdx.x, step = gridDim.x;
    if (first >= P) return;
    // (clip, frame, head) of problem i, advanced by `step` problems with carries: the three integer divisions of a direct decode cost ~1500
    // cycles per problem on the scalar unit (measured: the gap between two problems' stamps), serial to everything
    struct Walk { int b, f, head; };
    auto decode = [&](int i) { Walk w; int bid = rev ? P - 1 - i : i; w.head = bid % heads; bid /= heads; w.f = bid % T; w.b = bid / T; return w; };
    const int dh = step % heads, df = (step / heads) % T, db = step / heads / T;
    auto advance = [&](Walk& w) {
        if (!rev) {
            w.head += dh; int c = w.head >= heads; w.head -= c ? heads : 0;
            w.f += df + c; c = w.f >= T; w.f -= c ? T : 0;
            w.b += db + c;
        } else {
            w.head -= dh; int c = w.head < 0; w.head += c ? heads : 0;
            w.f -= df + c; c = w.f < 0; w.f += c ? T : 0;
            w.b -= db + c;
        }
    };
    unsigned long long* stamps = dbg == 3 ? (unsigned long long*)cls_partial + (int64_t)blockIdx.x * 8 : nullptr;
    if (dbg == 3) cls_partial = nullptr;
    const bool stamp_wave = dbg == 3 && tid == 0;

    if (wave >= 4) {
        // ================================================================ loader waves
        const int lw = wave - 4;
        // piece j of this wave = rows 8 (lw + 4 j) + (lane >> 3): row & 7 = lane >> 3 and (row >> 3) & 1 = lw & 1 for every j, so a stream's
        // swizzled chunk is one lane constant; K / V pieces 0 .. PPW-2 hold frame keys, the last one the CLS row (rows n .. n + 31)
        const unsigned ck = (unsigned)((lane & 7) ^ (lane >> 3) ^ (lw & 1)), cv = (unsigned)((lane & 7) ^ (((lane >> 4) & 1) << 2));
        const unsigned ld_b = (unsigned)ld * 2u;
        const unsigned rowoff = (unsigned)(lw * 8 + (lane >> 3)) * ld_b;
        const unsigned voff_k = rowoff + ck * 16u, voff_v = rowoff + cv * 16u;
        const unsigned voff_o = ((unsigned)(lane >> 3) * (unsigned)D + 8u * (unsigned)(lane & 7)) * 2u;
        // TWO register sets: while problem i is computed, set `cur` holds problem i + 1 (requested one problem ago, written to LDS at the next
        // boundary) and set `oth` receives problem i + 2 -- two problems' worth of requests (208 KB per CU) stay in flight, also across the
        // boundary itself.  (One set: the requests of problem i + 1 were issued only after barrier B and took 12 200 cycles against a 7 600-cycle
        // chunk loop -- every compute wave waited 4 500 cycles at barrier A, the CU had nothing in flight during the 1 900-cycle boundary.)
        struct Regs { u32x4 k[PPW], v[PPW], q[QPW]; };
        Regs X, Y;
        // The requests are inline asm and the waits manual: with compiler-visible loads hipcc allocated address temporaries inside the other
        // set's destination registers and made the ds_writes of `cur` wait for the requests of `oth` issued a moment earlier (vmcnt(25) .. (0) at
        // the boundary: a full memory latency per problem).  Nothing reads a set between its request and the landed() that follows a whole
        // problem later (scripts/check_isa_hazards.py scans for it).
        auto request = [&](const Walk& w, Regs& R) {
            const int b = w.b, f = w.f, head = w.head;
            const char* base = (const char*)(qkv + (int64_t)b * N * ld + head * hs);       // the clip's row 0 (CLS) of this head's q plane
            const char* q_ptr = base + (int64_t)(1 + f * n) * ld * 2;
#define SP32_LD(DST, VOFF, SBASE) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(DST) : "v"(VOFF), "s"(SBASE))
#pragma unroll
            for (int j = 0; j < QPW; ++j) { const char* sb = q_ptr + (int64_t)(32 * j) * ld_b; SP32_LD(R.q[j], voff_k, sb); }     // Q rows: the K swizzle
#pragma unroll
            for (int j = 0; j < PPW - 1; ++j) {
                const char* sk = q_ptr + (int64_t)(32 * j) * ld_b + ws * 2;
                const char* sv = q_ptr + (int64_t)(32 * j) * ld_b + ws * 4;
                SP32_LD(R.k[j], voff_k, sk);
                SP32_LD(R.v[j], voff_v, sv);
            }
            const char* sk = base + ws * 2;
            const char* sv = base + ws * 4;
            const unsigned ok = ck * 16u, ov = cv * 16u;
            SP32_LD(R.k[PPW - 1], ok, sk);
            SP32_LD(R.v[PPW - 1], ov, sv);
#undef SP32_LD
        };
        // `younger` = vector-memory instructions this wave issued AFTER the set's request (0, or the 2 PPW + QPW loads of the other set)
        auto landed = [&](Regs& R, bool other_requested) {
            if (other_requested) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPW + QPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < QPW; ++j) asm volatile("" : "+v"(R.q[j]));
#pragma unroll
            for (int j = 0; j < PPW; ++j) asm volatile("" : "+v"(R.k[j]), "+v"(R.v[j]));
        };
        auto q_to_lds = [&](const Regs& R) {
#pragma unroll
            for (int j = 0; j < QPW; ++j) *(u32x4*)(Qs + (lw + 4 * j) * 1024 + lane * 16) = R.q[j];
        };
        auto kv_to_lds = [&](const Regs& R) {
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                *(u32x4*)(Ks + (lw + 4 * j) * 1024 + lane * 16) = R.k[j];
                *(u32x4*)(Vs + (lw + 4 * j) * 1024 + lane * 16) = R.v[j];
            }
        };
        // [2 sets][4 compute waves]: bit 0 / 1 = block a / b was redone.  (A plain LDS pointer: as `volatile unsigned*` it became a generic pointer,
        // i.e. flat_load + s_waitcnt vmcnt(0) -- which waits for every request in flight.  The barriers order the accesses.)
        const unsigned* bad_flags = (const unsigned*)(scratch + 2 * 4 * CLS_REC);
        // rows of problem ip (compute wave w's 64 rows at rows_all + 8192 w; this loader wave stores those of compute wave lw) + its CLS record
        auto flush = [&](const Walk& w, int par_prev) {
            unsigned bm = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) bm |= (bad_flags[par_prev * 4 + w] & 3u) << (2 * w);
            const unsigned bad_mask = (unsigned)__builtin_amdgcn_readfirstlane((int)bm);
            const int b = w.b, f = w.f, head = w.head;
            const char* rows = rows_all + lw * 8192;
            if (cls_partial != nullptr && lw == 0)
                space16_cls_merge<4>(scratch + par_prev * (4 * CLS_REC), cls_partial + (((int64_t)b * heads + head) * T + f) * CLS_REC, lane);
            char* orow = (char*)(out + ((int64_t)b * N + 1 + f * n + lw * 64) * D + head * 64);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if ((bad_mask >> (2 * lw + half)) & 1u) continue;       // (wave-uniform: a redone block was stored by its compute wave)
                u32x4 v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = 8 * (4 * half + t) + (lane >> 3);
                    v[t] = *(const u32x4*)(rows + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) *(u32x4*)(orow + (int64_t)(8 * (4 * half + t)) * D * 2 + (size_t)voff_o) = v[t];
            }
        };
        int par = 0;
        bool have_prev = false;
        Walk wcur = decode(first), wprev = wcur, wreq = wcur;       // problems i, i - step (once have_prev), i + 2 step
        auto body = [&](int i, Regs& cur, Regs& oth) {
            if (have_prev) flush(wprev, par ^ 1);
            const bool req = i + 2 * step < P;
            if (req) request(wreq, oth);
            if (i + step < P) landed(cur, req);        // (requested a whole problem ago; the stores of flush() are older than the new requests)
            __syncthreads();                           // A(i): the compute waves are done with K / V / Q of problem i
            if (i + step < P) { q_to_lds(cur); kv_to_lds(cur); }
            __syncthreads();                           // B(next): K / V / Q of the next problem visible, rows of problem i complete
            wprev = wcur; have_prev = true; par ^= 1;
            advance(wcur); advance(wreq);
        };
        request(wreq, X);
        landed(X, false);
        q_to_lds(X);
        kv_to_lds(X);
        __syncthreads();                               // B(first)
        advance(wreq);
        if (first + step < P) request(wreq, X);
        advance(wreq);
        for (int i = first; i < P;) {
            body(i, X, Y);
            i += step;
            if (i >= P) break;
            body(i, Y, X);
            i += step;
        }
        flush(wprev, par ^ 1);
        return;
    }

    // ==================================================================== compute waves
    const int r = lane & 31, hh = lane >> 5;
    const int q0 = wave * 64;
    const int kz = (r & 7) ^ ((r >> 3) & 1);
    const char* kb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kb[ks] = Ks + r * 128 + (((2 * ks + hh) ^ kz) << 4);
    const int i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, e16 = (lane >> 4) & 1, vz = ((q4 >> 1) & 1) << 2;
    const char* vb[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) vb[dt] = Vs + (4 * hh + q4) * 128 + (((4 * dt + 2 * e16 + (pp >> 1)) ^ vz) << 4) + 8 * (pp & 1);
    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    char* rows = rows_all + wave * 8192;
    unsigned* bad_flags = (unsigned*)(scratch + 2 * 4 * CLS_REC);
    if (stamp_wave) stamps[6] = __builtin_amdgcn_s_memrealtime();          // (100 MHz: absolute time of the workgroup's start)
    __syncthreads();                                   // B(first)
    int par = 0;
    Walk wc = decode(first);
    for (int i = first; i < P; i += step) {
        const int b = wc.b, f = wc.f, head = wc.head;
        advance(wc);
        const bool st = stamp_wave;                     // debug mode 3: per-phase cycle totals over all problems of this workgroup (compute wave 0)
        unsigned long long t0 = 0, t1 = 0, t2 = 0;
        if (st) t0 = __builtin_readcyclecounter();
        // (an opaque copy of the lane index per problem: everything addressed from it -- Q fragments, the rows, the redo path -- is recomputed
        // here instead of being hoisted out of the problem loop, where ~35 loop-invariant address registers took the chunk loop's arch VGPRs to
        // the 256 cap and the scheduler gave up the pinned interleave)
        int lane_i = lane;
        asm volatile("" : "+v"(lane_i));
        const int r_i = lane_i & 31, hh_i = lane_i >> 5, kz_i = (r_i & 7) ^ ((r_i >> 3) & 1);
        bf16x8 qa[4], qb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qa[ks] = *(const bf16x8*)(Qs + (q0 + r_i) * 128 + (((2 * ks + hh_i) ^ kz_i) << 4));
            qb[ks] = *(const bf16x8*)(Qs + (q0 + 32 + r_i) * 128 + (((2 * ks + hh_i) ^ kz_i) << 4));
        }
        f32x16 oa[2] = {z16, z16}, ob[2] = {z16, z16}, sa, sb, ola = z16, olb = z16;
        bf16x8 pa[2], pb[2];
        bf16x8 kf[2][4];                               // K fragments of chunk c in kf[c & 1]
        sp32_load_k(kb, 0, kf[0]);
        sp32_qk_regs(kf[0], qa, sa);
#pragma unroll
        for (int ci = 0; ci < NCH; ++ci) {
            sp32_half_step<true>(kb, vb, ci - 1, kf[ci & 1], ci + 1, kf[(ci + 1) & 1], qb, sb, pb, ob, olb, sa, pa, ci > 0);
            sp32_half_step<false>(kb, vb, ci, kf[(ci + 1) & 1], 0, kf[ci & 1], qa, sa, pa, oa, ola, sb, pb, true);
        }
        sp32_pv<false>(vb, NCH - 1, pb, ob, olb);
        sp32_qk_regs(kf[NCH & 1], qb, sb);
        sp32_exp_cls(sa, pa, lane);
        __builtin_amdgcn_sched_barrier(0);
        sp32_exp_cls(sb, pb, lane);
        sp32_pv<true>(vb, NCH, pa, oa, ola);
        sp32_pv<true>(vb, NCH, pb, ob, olb);
        if (st) t1 = __builtin_readcyclecounter();
        const float la = ola[0], lb = olb[0];
        const bool bad_a = __builtin_amdgcn_ballot_w64(!(la >= 7.9e-31f && la <= 1.2e30f)) != 0;
        const bool bad_b = __builtin_amdgcn_ballot_w64(!(lb >= 7.9e-31f && lb <= 1.2e30f)) != 0;
        const bf16_t* base = qkv + (int64_t)b * N * ld + head * hs;
        if (bad_a || bad_b) {
            const bf16_t* q_ptr = base + (int64_t)(1 + f * n) * ld;
            bf16_t* orow = out + ((int64_t)b * N + 1 + f * n + q0) * D + head * 64;
            const int c = lane_i & 15, g = lane_i >> 4, nt = (n >> 4) + 1;
#pragma unroll 1
            for (int sub = 0; sub < 4; ++sub) {
                if (!(sub < 2 ? bad_a : bad_b)) continue;
                if ((sub & 1) == 0) HH_SPACE_REDO_NOTE(g_space32_redo, lane);          // (one count per 32-query block)
                const bf16_t* qrow = q_ptr + (int64_t)(q0 + 16 * sub + c) * ld + 8 * g;
                bf16x8 q16[2];
                q16[0] = *(const bf16x8*)(qrow);
                q16[1] = *(const bf16x8*)(qrow + 32);
                f32x4 o2[4] = {z4, z4, z4, z4};
                float m_run = -INFINITY, l_run = 0.f;
                for (int t = 0; t + 1 < nt; ++t) space16_chunk<1, false>(Ks, Vs, q16, t, lane_i, o2, m_run, l_run);
                space16_chunk<1, true>(Ks, Vs, q16, nt - 1, lane_i, o2, m_run, l_run);
                space_store_block(o2, l_run, orow + (int64_t)(16 * sub + c) * D + 16 * g, c, D);
            }
        }
        if (lane == 0) bad_flags[par * 4 + wave] = (bad_a ? 1u : 0u) | (bad_b ? 2u : 0u);
        if (cls_partial != nullptr) {
            bf16x8 qc[2];                             // the CLS query = row 0 of the clip: one 128-B row, read through the scalar / vector cache
            qc[0] = *(const bf16x8*)(base + 8 * (lane_i >> 4));
            qc[1] = *(const bf16x8*)(base + 8 * (lane_i >> 4) + 32);
            space16_cls_wave<4>(Ks, Vs, scratch + par * (4 * CLS_REC), qc, n, f == 0, lane_i, wave);
        }
        __syncthreads();                               // A(i)
        if (st) t2 = __builtin_readcyclecounter();
        if (!bad_a) sp32_o_to_lds(rows, oa, la, r_i, hh_i);
        if (!bad_b) sp32_o_to_lds(rows + 32 * 128, ob, lb, r_i, hh_i);
        __syncthreads();                               // B(next)
        if (st) {
            const unsigned long long t3 = __builtin_readcyclecounter();
            stamps[0] += t1 - t0; stamps[1] += t2 - t1; stamps[2] += t3 - t2; stamps[3] += 1;
            if (i == first) stamps[4] = t0;
            stamps[5] = t3;
            stamps[7] = __builtin_amdgcn_s_memrealtime();
        }
        par ^= 1;
    }
}


// ---- the same idea on 16x16x32 MFMAs for frames whose 32-query blocks do not split evenly over the SIMDs (round 6: n = 576, config 4).
// The progressive kernel of attn_space.hip runs a chunk of a wave's three 16-query blocks as three serial phases (12 score MFMAs -> 24
// exponentials -> 15 PV MFMAs): ~4000 cycles per chunk for a SIMD's three waves against 1296 of matrix-core time.  Here the three blocks of
// a wave run their chains QK -> exp -> PV a THIRD of a chunk apart: third-step u holds the scores of block u % 3 (chunk u / 3: 4 MFMAs),
// the probabilities of block (u + 2) % 3 (8 v_exp_f32 + 4 v_cvt_pk) and the PV + row-sum products of block (u + 1) % 3 (5 MFMAs) -- nine
// independent MFMAs and the exponentials of ANOTHER block in every third-step, no second score buffer (a block's scores are overwritten two
// third-steps after their exponentials), K / V^T fragments of a chunk read once and used by all three blocks.  No reference maximum (see
// above); plain K / V staging (the fragment reads stay compiler-visible so that the group barriers can place them).
#define NW16P 12
template <int U, int NC>       // third-step U of a wave's walk over NC chunks (the last chunk holds only the CLS key, in row 0 of its first tile)
__device__ __forceinline__ void sp16_third_step(const char* kb0, const char* kb1, const char* vb0, const char* vb2, const bf16x8 (&q)[3][2], f32x4 (&s)[3][2],
                                                bf16x8 (&pf)[3], f32x4 (&o)[3][4], f32x4 (&ol)[3], bf16x8 (&kf)[2][2], bf16x8 (&vf)[4], int lane) {
    constexpr int X = U % 3, CQ = U / 3;                     // scores: block X, chunk CQ
    constexpr int Y = (U + 2) % 3, CE = (U - 1) / 3;         // probabilities: block Y, chunk CE (U >= 1)
    constexpr int Z = (U + 1) % 3, CP = (U - 2) / 3;         // PV: block Z, chunk CP (U >= 2)
    constexpr bool QK = CQ < NC, EX = U >= 1 && CE < NC, PV = U >= 2 && CP < NC;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
    if constexpr (PV) {
        if constexpr (Z == 2 - 2 || (U - 2) % 3 == 0) {      // first PV third-step of chunk CP: its V^T fragments (two key tiles per operand)
            constexpr int OB = CP == NC - 1 ? 0 : 2048;      // (the CLS chunk has one tile: the second half re-reads it, its probabilities are 0)
#define SP16_V(DT, VC, SUB) do { const bf16x4 a = lds_tr4(VC + CP * 4096 + SUB), b = lds_tr4(VC + CP * 4096 + OB + SUB); \
                                 vf[DT] = (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; } while (0)
            SP16_V(0, vb0, 0); SP16_V(1, vb0, 8); SP16_V(2, vb2, 0); SP16_V(3, vb2, 8);
#undef SP16_V
        }
        ol[Z] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[Z], ol[Z], 0, 0, 0);
    }
    if constexpr (QK) {
        if constexpr (X == 0) {                              // first score third-step of chunk CQ: its K fragments
            kf[0][0] = *(const bf16x8*)(kb0 + CQ * 4096); kf[0][1] = *(const bf16x8*)(kb1 + CQ * 4096);
            if constexpr (CQ < NC - 1) { kf[1][0] = *(const bf16x8*)(kb0 + CQ * 4096 + 2048); kf[1][1] = *(const bf16x8*)(kb1 + CQ * 4096 + 2048); }
        }
        s[X][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0][0], q[X][0], z4, 0, 0, 0);
        s[X][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0][1], q[X][1], s[X][0], 0, 0, 0);
        if constexpr (CQ < NC - 1) {
            s[X][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1][0], q[X][0], z4, 0, 0, 0);
            s[X][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1][1], q[X][1], s[X][1], 0, 0, 0);
        }
    }
    if constexpr (PV) {
        o[Z][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0], pf[Z], o[Z][0], 0, 0, 0);
        o[Z][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[1], pf[Z], o[Z][1], 0, 0, 0);
        o[Z][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[2], pf[Z], o[Z][2], 0, 0, 0);
        o[Z][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[3], pf[Z], o[Z][3], 0, 0, 0);
    }
    if constexpr (EX) {
        const bf16_t zb = (bf16_t)0.f;
        if constexpr (CE < NC - 1) {
            float e[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(s[Y][0][r]); e[4 + r] = __builtin_amdgcn_exp2f(s[Y][1][r]); }
            pf[Y] = (bf16x8){(bf16_t)e[0], (bf16_t)e[1], (bf16_t)e[2], (bf16_t)e[3], (bf16_t)e[4], (bf16_t)e[5], (bf16_t)e[6], (bf16_t)e[7]};
        } else {
            // the CLS key: row 0 of the tile = register 0 of lanes 0..15 (the lane index is recomputed here -- v_mbcnt -- instead of living in a
            // register across the walk: the kernel sits at the 168-register cap of three waves per SIMD)
            (void)lane;
            const unsigned lid = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            const float e0 = lid < 16u ? __builtin_amdgcn_exp2f(s[Y][0][0]) : 0.f;
            pf[Y] = (bf16x8){(bf16_t)e0, zb, zb, zb, zb, zb, zb, zb};
        }
        asm volatile("" : "+v"(pf[Y]));                      // made HERE (LLVM otherwise sinks the exponentials to their use, a third-step later)
    }
#ifndef HH_SP16_NO_GROUPS
    if constexpr (QK && EX && PV && CQ < NC - 1 && CE < NC - 1) {        // the steady state: 9 MFMAs, 8 exponentials, 4 conversions
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // MFMA
            if (g == 0) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);              // this third-step's fragment reads (0, 4 or 8)
            if (g >= 1 && (g & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);      // VALU: a conversion of finished exponentials
            if (g < 8) __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);               // TRANS: one exponential per 16-cycle gap
        }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
}

// progressive staging (PROG): the K / V rows are requested in four segments in key order (S0 + S1 + S2 + S3 pieces of 8 rows; E0 / E1 / E2 =
// the chunks at which segments 1 / 2 / 3 begin) and the walk starts on segment 0 while the rest is in flight: a counted s_waitcnt vmcnt +
// raw s_barrier in front of the third-step that first touches a segment's K rows (its V rows are touched two third-steps later).  Every
// vector-memory instruction of the kernel is inline asm (the Q loads, the LDS-DMA), so hipcc's own wait-count pass inserts nothing -- with a
// compiler-visible LDS-DMA it would put vmcnt(0) in front of every LDS read (DESIGN_HISTORY 4.2) -- and the fragment reads stay plain loads
// that the group barriers can place; the memory clobber of the wait keeps them below it.
template <int BASE>
__device__ __forceinline__ void sp16_wait(int c3) {      // vmcnt(BASE + 2 c3), c3 in {1, 2} wave-uniform: this wave's LDS-DMAs of the ragged last segment
    if (c3 == 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(BASE + 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(BASE + 2) : "memory");
}
template <int NC, int U0, int U1, bool PROG, int E0, int E1, int E2>
struct Sp16Walk {
    template <typename... A> __device__ __forceinline__ static void run(int c3, A&... a) {
        if constexpr (U0 < U1) {
            if constexpr (PROG && U0 == 3 * E0) sp16_wait<4>(c3);            // segment 1 landed everywhere (segments 2, 3 may be in flight: 2 + 2 + ... per tile pair)
            if constexpr (PROG && U0 == 3 * E1) sp16_wait<0>(c3);            // segment 2
            if constexpr (PROG && U0 == 3 * E2) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // segment 3
            sp16_third_step<U0, NC>(a...);
            Sp16Walk<NC, U0 + 1, U1, PROG, E0, E1, E2>::run(c3, a...);
        }
    }
};
__device__ __forceinline__ void sp16_dma(const char* sbase, unsigned voff, unsigned dst_lds) {       // one 1-KB piece: lane l's 16 B at sbase + voff -> LDS dst + 16 l
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(dst_lds) : "memory", "m0");
}
__device__ __forceinline__ bf16x8 sp16_gld128(const bf16_t* ptr) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr));
    return __builtin_bit_cast(bf16x8, r);
}

template <int NQB, bool PROG>            // n = 16 * NQB keys per frame = NW16P waves x 3 blocks
__global__ __launch_bounds__(64 * NW16P, 1) void space_attn16p_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                      float* __restrict__ cls_partial, int B, int T, int heads, int layout, int dbg) {
    constexpr int n = NQB * 16, nt = NQB + 1, KP = ((n + 1 + 31) / 32) * 32, NC = NQB / 2 + 1;
    static_assert(NQB == NW16P * 3 && NQB % 2 == 0, "one group of three blocks per wave, chunks of two key tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + (size_t)KP * 128;
    float* scratch = (float*)(smem + (size_t)KP * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = heads * 64;
    const int N = 1 + T * n;
    const int rev = layout >> 1;
    layout &= 1;
    const int64_t ld = layout ? 64 : 3 * (int64_t)D;
    const int64_t hs = layout ? (int64_t)B * N * 64 : 64, ws = (int64_t)heads * hs;
    int bid = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int head = bid % heads; bid /= heads;
    const int f = bid % T;
    const int b = bid / T;
    const bf16_t* base = qkv + (int64_t)b * N * ld + head * hs;
    const bf16_t* q_ptr = base + (int64_t)(1 + f * n) * ld;
    const int c = lane & 15, g = lane >> 4;
    const int gb = wave * 3;
    bf16x8 q[3][2];
    // segments of 12 / 24 / 24 / 16 pieces (the progressive kernel's split): every wave stages 1 / 2 / 2 pieces per tile of segments 0-2 and 2
    // (waves 0-3) or 1 of the last
    constexpr int S0 = 12, S1 = 24, S2 = 24, S3 = KP / 8 - 60, E0 = S0 / 4, E1 = E0 + S1 / 4, E2 = E1 + S2 / 4;
    static_assert(!PROG || (NW16P == 12 && S3 > 12 && S3 <= 24), "segment plan of n = 576");
    const int c3 = (wave < S3 - NW16P) ? 2 : 1;
    if constexpr (PROG) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bf16_t* qrow = q_ptr + (int64_t)((gb + j) * 16 + c) * ld + 8 * g;
            q[j][0] = sp16_gld128(qrow);
            q[j][1] = sp16_gld128(qrow + 32);
        }
        const char* kbase = (const char*)(base + ws);
        const char* vbase = (const char*)(base + 2 * ws);
        const unsigned ld_b = (unsigned)ld * 2u;
        auto seg = [&](int p0, int cnt) {
            for (int which = 0; which < 2; ++which)
                for (int i = wave; i < cnt; i += NW16P) {          // (wave-uniform trip counts)
                    const int pc = p0 + i, row = pc * 8 + (lane >> 3);
                    const unsigned rowidx = row < n ? (unsigned)(1 + f * n + row) : 0u;      // rows >= n take the CLS row (key n; the rest is masked)
                    const unsigned ch = (unsigned)((lane & 7) ^ (which == 0 ? kswz(row) : vswz(row)));
                    sp16_dma(which == 0 ? kbase : vbase, rowidx * ld_b + ch * 16u, (unsigned)(size_t)(__attribute__((address_space(3))) char*)((which == 0 ? Ks : Vs) + pc * 1024));
                }
        };
        seg(0, S0); seg(S0, S1); seg(S0 + S1, S2); seg(S0 + S1 + S2, S3);
        sp16_wait<8>(c3);                                  // the Q rows (older than every LDS-DMA) and segment 0 have landed everywhere
        asm volatile("" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[2][0]), "+v"(q[2][1]));
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bf16_t* qrow = q_ptr + (int64_t)((gb + j) * 16 + c) * ld + 8 * g;
            q[j][0] = *(const bf16x8*)(qrow);
            q[j][1] = *(const bf16x8*)(qrow + 32);
        }
        space_stage<NW16P>(Ks, Vs, base, q_ptr, ld, ws, n, KP, lane, wave);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const int kz = (c & 7) ^ (c >> 3), trq = c >> 2, trp = c & 3, vz = ((trq >> 1) & 1) << 2;
    const char* kb0 = Ks + c * 128 + ((g ^ kz) << 4);
    const char* kb1 = Ks + c * 128 + (((g + 4) ^ kz) << 4);
    const char* vb0 = Vs + (4 * g + trq) * 128 + (((2 * trp) ^ vz) << 4);
    const char* vb2 = Vs + (4 * g + trq) * 128 + (((2 * trp + 1) ^ vz) << 4);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 s[3][2], o[3][4], ol[3];
    bf16x8 pf[3], kf[2][2], vf[4];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ol[j] = z4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[j][dt] = z4;
    }
    if (dbg == 1) {                                    // debug: memory traffic only (stage K / V, read Q, write the rows) -- hh_set_tuning("space_debug", 1)
        if constexpr (PROG) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            bf16_t* op = out + ((int64_t)b * N + 1 + f * n + (gb + j) * 16 + c) * D + head * 64 + 16 * g;
            *(bf16x8*)(op) = q[j][0]; *(bf16x8*)(op + 8) = q[j][1];
        }
        return;
    }
    Sp16Walk<NC, 0, 3 * NC + 2, PROG, E0, E1, E2>::run(c3, kb0, kb1, vb0, vb2, q, s, pf, o, ol, kf, vf, lane);
    if (dbg == 2) return;                              // debug: no output, no CLS partial (staging + walk only)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        bf16_t* op = out + ((int64_t)b * N + 1 + f * n + (gb + j) * 16 + c) * D + head * 64 + 16 * g;
        float l_run = ol[j][0];
        // a row sum outside [2^-100, 2^100] (no reference maximum here): the block is redone on the running-maximum path
        if (__builtin_amdgcn_ballot_w64(!(l_run >= 7.9e-31f && l_run <= 1.2e30f)) != 0) {
            HH_SPACE_REDO_NOTE(g_space32_redo, lane);
            f32x4 o2[4] = {z4, z4, z4, z4};
            float m_run = -INFINITY;
            l_run = 0.f;
            for (int t = 0; t + 1 < nt; ++t) space16_chunk<1, false>(Ks, Vs, q[j], t, lane, o2, m_run, l_run);
            space16_chunk<1, true>(Ks, Vs, q[j], nt - 1, lane, o2, m_run, l_run);
            space_store_block(o2, l_run, op, c, D);
        } else {
            space_store_block(o[j], l_run, op, c, D);
        }
    }
    if (cls_partial == nullptr) return;
    space16_cls_partial<NW16P>(Ks, Vs, scratch, base, cls_partial + (((int64_t)b * heads + head) * T + f) * CLS_REC, n, f == 0, tid, lane, wave);
}

int hh_space_attn16p_launch(const void* qkv, int layout_rev, void* out, float* cls_partial, int B, int T, int n, int heads, int progressive, int dbg, hipStream_t stream) {
    typedef void (*k_t)(const bf16_t*, bf16_t*, float*, int, int, int, int, int);
    HH_REQUIRE(n == 576, HH_ERR_UNSUPPORTED, "hh_space_attn_fwd: the third-step pipelined 16x16x32 kernel is built for n = 576 (got %d)", n);
    const k_t k = progressive ? (k_t)space_attn16p_kernel<36, true> : (k_t)space_attn16p_kernel<36, false>;
    const int KP = ((n + 1 + 31) / 32) * 32;
    const size_t lds = (size_t)KP * 256 + ((size_t)KP + 12 * CLS_REC + 24) * 4;
    static size_t attr[2] = {0, 0};
    if (lds > attr[progressive ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_space_attn_fwd: cannot reserve %zu B of LDS", lds);
        attr[progressive ? 1 : 0] = lds;
    }
    hh_prof_note_kernel(HH_PROF_SPACE_ATTN, progressive ? "space_attn16p_kernel<36, true>" : "space_attn16p_kernel<36, false>");
    hipLaunchKernelGGL(k, dim3((unsigned)((int64_t)B * T * heads)), dim3(64 * NW16P), lds, stream, (const bf16_t*)qkv, (bf16_t*)out, cls_partial, B, T, heads, layout_rev, dbg);
    return hh_check_launch("hh_space_attn_fwd");
}

// mode: hh_set_tuning("space_mfma32"): 1 (default) = one problem per workgroup, 2 = the persistent wave-specialised kernel where n == 256.  The
// persistent kernel is NOT the default although it is the fastest of the three inside the step (286-300 us against 294-312): it holds every
// CU for its whole duration, and the decoder stream's span inside the pipelined step grows from 0.74 to 0.92 of the step (79 -> 97-100 ms,
// profiles/r6_space_instep.txt) at unchanged clips/s -- the margin that eight ranks with RCCL kernels will need
int hh_space_attn32_launch(const void* qkv, int layout_rev, void* out, float* cls_partial, int B, int T, int n, int heads, int mode, int dbg, hipStream_t stream) {
    typedef void (*k32_t)(const bf16_t*, bf16_t*, float*, int, int, int, int, int);
    const int nch = n / 32, KP = n + 32;
    HH_REQUIRE(n % 64 == 0 && n >= 64 && n <= 256, HH_ERR_UNSUPPORTED, "hh_space_attn_fwd: the 32x32x16 kernels take n in {64, 128, 192, 256}, got %d", n);
    if (n == 256 && mode == 2 && dbg != 1 && dbg != 2) {
        // persistent form: one workgroup per CU walks the problems, the next problem's K / V / Q are requested before the current chunk loop
        const k32_t kp = (k32_t)space_attn32p_kernel<8>;
        const size_t ldsp = (size_t)KP * 256 + (size_t)n * 128 + 4 * 8192 + ((size_t)2 * 4 * CLS_REC + 64) * 4;
        static size_t attrp32 = 0;
        if (ldsp > attrp32) {
            hipError_t e = hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
            HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_space_attn_fwd: cannot reserve %zu B of LDS", ldsp);
            attrp32 = ldsp;
        }
        const int64_t P = (int64_t)B * T * heads;
        const int cus = hh_stream_cu_count(stream);
        hh_prof_note_kernel(HH_PROF_SPACE_ATTN, "space_attn32p_kernel<8>");
        hipLaunchKernelGGL(kp, dim3((unsigned)(P < cus ? P : cus)), dim3(64 * NW32P), ldsp, stream,
                           (const bf16_t*)qkv, (bf16_t*)out, cls_partial, B, T, heads, dbg, layout_rev);
        return hh_check_launch("hh_space_attn_fwd");
    }
    const size_t lds = (size_t)KP * 256 + ((size_t)KP + 12 * CLS_REC + 24) * 4;          // (the same size as the 16x16x32 kernels reserve)
    const k32_t k32 = nch == 8 ? (k32_t)space_attn32_kernel<8> : nch == 6 ? (k32_t)space_attn32_kernel<6> : nch == 4 ? (k32_t)space_attn32_kernel<4> : (k32_t)space_attn32_kernel<2>;
    static size_t attr32[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (lds > attr32[nch]) {
        hipError_t e = hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_space_attn_fwd: cannot reserve %zu B of LDS", lds);
        attr32[nch] = lds;
    }
    hh_prof_note_kernel(HH_PROF_SPACE_ATTN, nch == 8 ? "space_attn32_kernel<8>" : nch == 6 ? "space_attn32_kernel<6>" : nch == 4 ? "space_attn32_kernel<4>" : "space_attn32_kernel<2>");
    hipLaunchKernelGGL(k32, dim3((unsigned)((int64_t)B * T * heads)), dim3(64 * NW32), lds, stream,
                       (const bf16_t*)qkv, (bf16_t*)out, cls_partial, B, T, heads, dbg, layout_rev);
    return hh_check_launch("hh_space_attn_fwd");
}

unsigned long long hh_space32_redo_read(int reset) {
    unsigned long long v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_space32_redo), sizeof(v)) != hipSuccess) return 0;
    if (reset) { const unsigned long long z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_space32_redo), &z, sizeof(z)); }
    return v;
}

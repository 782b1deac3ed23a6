// Space attention core of the divided space-time attention (model/LaviLa.py:246-283 with '(b f) n d', attn() :194-198).
// One workgroup per (clip, frame, head): n queries x (n frame keys + CLS key), head dim 64, no mask, no dropout.
// HBM-bound in principle: algorithmic bytes per workgroup = (3*n + 2) * 128 B read + n * 128 B written.
//
// Structure: K tile and V tile, both row-major [KP][64] with XOR-swizzled 16-B chunks, staged once in LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no VALU), then each wave walks 16-query blocks on v_mfma_f32_16x16x32_bf16
// (see "16-query blocks" below).  Frame keys sit at rows 0..n-1, the CLS key at row n, rows > n are zero/masked.
// The q columns arrive pre-scaled by d^-1/2 * log2(e) (QKV GEMM epilogue), so scores are base-2 logits and a probability is ONE
// v_exp_f32 -- see "fast path" below for how the softmax's VALU work was moved onto the matrix core.
// The first generation (32-query blocks on 32x32x16 MFMAs, 4 waves per workgroup: 418 us per call at B = 32, VALU busy 49 %, MFMA
// 18 %) was measured against this kernel in round 1 (365 -> 352 us) and removed in round 2.
#include "common.h"

#include "attn_space_dev.h"

HH_SPACE_REDO_COUNTER(g_space_redo);

// One 16-query block of one (clip, frame, head) problem against the nt key tiles staged in LDS: fast path, running-maximum redo, store.
__device__ __forceinline__ void space16_block(const char* Ks, const char* Vs, const bf16x8 (&q)[2], int nt, int lane, bf16_t* op) {
    const int c = lane & 15, g = lane >> 4;
    // lane-constant fragment addresses of key tile 0 (see kswz / vswz: the swizzle keys do not depend on the tile)
    const int kz = (c & 7) ^ (c >> 3), trq = c >> 2, trp = c & 3, vz = ((trq >> 1) & 1) << 2;
    const char* kb0 = Ks + c * 128 + ((g ^ kz) << 4);
    const char* kb1 = Ks + c * 128 + (((g + 4) ^ kz) << 4);
    const char* vb0 = Vs + (4 * g + trq) * 128 + (((2 * trp) ^ vz) << 4);
    const char* vb1 = vb0 + 8;
    const char* vb2 = Vs + (4 * g + trq) * 128 + (((2 * trp + 1) ^ vz) << 4);
    const char* vb3 = vb2 + 8;
    const bool fast = nt > CH16;                       // short frames (n <= 128) take the running-maximum path (first chunk would hold the CLS tile)
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 o[4] = {z4, z4, z4, z4};
    float l_run = 0.f;
    bool redo = !fast;
    if (fast) {
        f32x4 ol = z4;
        float m_ref = 0.f;
        space16_fast<CH16, false, true>(kb0, kb1, vb0, vb1, vb2, vb3, q, lane, o, ol, m_ref);
        int t0 = CH16;
        for (; t0 + CH16 < nt; t0 += CH16) {
            const int off = t0 * 2048;
            space16_fast<CH16, false, false>(kb0 + off, kb1 + off, vb0 + off, vb1 + off, vb2 + off, vb3 + off, q, lane, o, ol, m_ref);
        }
        const int off = t0 * 2048;
#define S16_FT(R) space16_fast<R, true, false>(kb0 + off, kb1 + off, vb0 + off, vb1 + off, vb2 + off, vb3 + off, q, lane, o, ol, m_ref)
        switch (nt - t0) {                             // 1 .. CH16 tiles left, the last one is the CLS tile
            case 1: S16_FT(1); break;
            case 2: S16_FT(2); break;
            case 3: S16_FT(3); break;
            case 4: S16_FT(4); break;
            case 5: S16_FT(5); break;
            case 6: S16_FT(6); break;
            case 7: S16_FT(7); break;
            case 8: S16_FT(8); break;
            default: S16_FT(CH16); break;
        }
#undef S16_FT
        l_run = ol[0];
        // a score more than 2^127 above the block's reference maximum: l is not finite -> redo this block with the running maximum
        redo = __builtin_amdgcn_ballot_w64(!(l_run <= 3.0e38f)) != 0;
    }
    if (redo) {
        if (fast) HH_SPACE_REDO_NOTE(g_space_redo, lane);      // (short frames always run this path: not a redo)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = z4;
        float m_run = -INFINITY;
        l_run = 0.f;
        for (int t0 = 0; t0 + 1 < nt; ++t0) space16_chunk<1, false>(Ks, Vs, q, t0, lane, o, m_run, l_run);
        space16_chunk<1, true>(Ks, Vs, q, nt - 1, lane, o, m_run, l_run);
    }
    const float inv = 1.f / l_run;
    const u32x4 w0 = {pack_bf16(o[0][0] * inv, o[0][1] * inv), pack_bf16(o[0][2] * inv, o[0][3] * inv),
                      pack_bf16(o[1][0] * inv, o[1][1] * inv), pack_bf16(o[1][2] * inv, o[1][3] * inv)};
    const u32x4 w1 = {pack_bf16(o[2][0] * inv, o[2][1] * inv), pack_bf16(o[2][2] * inv, o[2][3] * inv),
                      pack_bf16(o[3][0] * inv, o[3][1] * inv), pack_bf16(o[3][2] * inv, o[3][3] * inv)};
    *(u32x4*)(op) = w0;
    *(u32x4*)(op + 8) = w1;
}

__global__ __launch_bounds__(64 * NW16, 4) void space_attn16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                  float* __restrict__ cls_partial, int B, int T, int n, int heads, int KP, int dbg, int layout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + (size_t)KP * 128;
    float* scratch = (float*)(smem + (size_t)KP * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = heads * 64;
    const int N = 1 + T * n;
    // qkv layout (include/hh.h, hh_qkv_layout): element (row, which, head, d) at row * ld + which * ws + head * hs + d
    const int64_t ld = layout ? 64 : 3 * (int64_t)D;
    const int64_t hs = layout ? (int64_t)B * N * 64 : 64, ws = (int64_t)heads * hs;
    int bid = blockIdx.x;
    const int head = bid % heads; bid /= heads;
    const int f = bid % T;
    const int b = bid / T;
    const bf16_t* base = qkv + (int64_t)b * N * ld + head * hs;
    const bf16_t* q_ptr = base + (int64_t)(1 + f * n) * ld;
    if (dbg != 2) space_stage<NW16>(Ks, Vs, base, q_ptr, ld, ws, n, KP, lane, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int c = lane & 15, g = lane >> 4;
    const int nt = (n >> 4) + 1;                       // 16-key tiles incl. the CLS tile
    if (dbg == 1) {                                    // debug: memory traffic only
        for (int qb = wave; qb < (n >> 4); qb += NW16) {
            const bf16_t* qrow = q_ptr + (int64_t)(qb * 16 + c) * ld + 8 * g;
            const u32x4 a = *(const u32x4*)(qrow), bq = *(const u32x4*)(qrow + 32);
            bf16_t* op = out + ((int64_t)b * N + 1 + f * n + qb * 16 + c) * D + head * 64 + 16 * g;
            *(u32x4*)(op) = a; *(u32x4*)(op + 8) = bq;
        }
        return;
    }
    for (int qb = wave; qb < (n >> 4); qb += NW16) {
        bf16x8 q[2];
        const bf16_t* qrow = q_ptr + (int64_t)(qb * 16 + c) * ld + 8 * g;
        q[0] = *(const bf16x8*)(qrow);
        q[1] = *(const bf16x8*)(qrow + 32);
        space16_block(Ks, Vs, q, nt, lane, out + ((int64_t)b * N + 1 + f * n + qb * 16 + c) * D + head * 64 + 16 * g);
    }
    if (cls_partial == nullptr) return;
    space16_cls_partial<NW16>(Ks, Vs, scratch, base, cls_partial + (((int64_t)b * heads + head) * T + f) * CLS_REC, n, f == 0, tid, lane, wave);
}


// ---- joint query blocks (round 2, second pass).  In the kernel above every MFMA consumes a fresh 1 KB LDS fragment: per workgroup
// 16 blocks x 70 KB of fragment reads against 1264 MFMAs, and the measured compute-only time (230-250 us) is about the SUM of the
// LDS, MFMA and VALU streams (~100 + 80 + 80 us) rather than their maximum.  Here a wave owns JB query blocks AT ONCE: a K or V^T
// fragment is read once and feeds JB MFMAs (LDS fragment traffic / JB), and the JB blocks are independent dependency chains inside
// one in-order wave.  4 waves per workgroup and two workgroups per CU = 2 waves per SIMD, so a wave may hold 256 registers:
// JB x (NTJ score tiles + 4 output tiles + row sums + Q + the -m_ref accumulator initialiser).  Every wave has exactly
// n / 16 / (4 JB) groups (the host picks JB so that this divides), and the Q rows of a wave's first group are requested BEFORE the
// K / V staging wait, so staging and Q latency overlap.  Same LDS layout, swizzles, fragment addressing and softmax scheme
// (fixed per-block reference maximum inside the MFMA accumulator, all-ones row-sum MFMA, running-maximum redo) as above.
// Round 3: the number of waves per workgroup is a template parameter.  At n = 576 (config 4: 336 px) K and V fill the CU's LDS
// (155 KB), so there is ONE workgroup per CU; with 4 waves that is one wave per SIMD walking three groups one after the other
// (346 us per call at B = 4 = 0.22 of the HBM roofline).  12 waves x 3 joint blocks cover the 36 query blocks of a frame at once:
// 3 waves per SIMD (168 registers each), every Q row requested before the staging wait.
#define NWJ 4
template <int JB, int NT, bool CLS, bool FIRST>
__device__ __forceinline__ void spacej_chunk(const char* kc0, const char* kc1, const char* vc0, const char* vc1, const char* vc2,
                                             const char* vc3, const bf16x8 (&q)[JB][2], int lane, f32x4 (&o)[JB][4], f32x4 (&ol)[JB],
                                             float (&m_ref)[JB]) {
    static_assert(!(FIRST && CLS) && (!FIRST || NT >= 2), "the first chunk holds two plain key tiles");
    const int g = lane >> 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
    f32x4 s[JB][NT];
    constexpr int NR = FIRST ? 2 : 0;
    if (FIRST) {
#pragma unroll
        for (int ti = 0; ti < NR; ++ti) {
            const bf16x8 k0 = *(const bf16x8*)(kc0 + ti * 2048), k1 = *(const bf16x8*)(kc1 + ti * 2048);
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                s[j][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, q[j][0], z4, 0, 0, 0);
                s[j][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, q[j][1], s[j][ti], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            float mx = fmaxf(fmaxf(fmaxf(s[j][0][0], s[j][0][1]), fmaxf(s[j][0][2], s[j][0][3])),
                             fmaxf(fmaxf(s[j][1][0], s[j][1][1]), fmaxf(s[j][1][2], s[j][1][3])));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            m_ref[j] = mx;
            s[j][0] -= mx;
            s[j][1] -= mx;
        }
    }
    f32x4 minit[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) minit[j] = (f32x4){-m_ref[j], -m_ref[j], -m_ref[j], -m_ref[j]};
#pragma unroll
    for (int ti = NR; ti < NT; ++ti) {
        const bf16x8 k0 = *(const bf16x8*)(kc0 + ti * 2048), k1 = *(const bf16x8*)(kc1 + ti * 2048);
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            s[j][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, q[j][0], minit[j], 0, 0, 0);
            s[j][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, q[j][1], s[j][ti], 0, 0, 0);
        }
    }
    if (CLS) {                                        // the chunk's last tile holds nothing but the CLS key (its row 0)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!(g == 0 && r == 0)) s[j][NT - 1][r] = -INFINITY;
    }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[j][ti][r] = __builtin_amdgcn_exp2f(s[j][ti][r]);
#pragma unroll
    for (int pr = 0; pr < (NT + 1) / 2; ++pr) {
        const bool has_b = 2 * pr + 1 < NT;
        bf16x8 pf[JB];
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const f32x4 pa = s[j][2 * pr], pb = s[j][has_b ? 2 * pr + 1 : 2 * pr];
            pf[j] = (bf16x8){(bf16_t)pa[0], (bf16_t)pa[1], (bf16_t)pa[2], (bf16_t)pa[3],
                             (bf16_t)(has_b ? pb[0] : 0.f), (bf16_t)(has_b ? pb[1] : 0.f), (bf16_t)(has_b ? pb[2] : 0.f), (bf16_t)(has_b ? pb[3] : 0.f)};
        }
        const int oa = 2 * pr * 2048, ob = has_b ? oa + 2048 : oa;
#define SJ_PV(DT, VC) do { const bf16x4 a0 = lds_tr4(VC + oa), a1 = lds_tr4(VC + ob);                                     \
                           const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};                       \
                           _Pragma("unroll") for (int j = 0; j < JB; ++j)                                                    \
                               o[j][DT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pf[j], o[j][DT], 0, 0, 0); } while (0)
        SJ_PV(0, vc0); SJ_PV(1, vc1); SJ_PV(2, vc2); SJ_PV(3, vc3);
#undef SJ_PV
#pragma unroll
        for (int j = 0; j < JB; ++j) ol[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[j], ol[j], 0, 0, 0);
    }
}

template <int JB, int NTJ, bool DBG, int NWV = NWJ, int WPS = 2>
__global__ __launch_bounds__(64 * NWV, WPS) void space_attnj_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                 float* __restrict__ cls_partial, int B, int T, int n, int heads, int KP,
                                                                 int dbg, int layout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + (size_t)KP * 128;
    float* scratch = (float*)(smem + (size_t)KP * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = heads * 64;
    const int N = 1 + T * n;
    // qkv layout (include/hh.h, hh_qkv_layout): element (row, which, head, d) at row * ld + which * ws + head * hs + d
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
    const int c = lane & 15, g = lane >> 4;
    const int nqb = n >> 4, nt = nqb + 1;              // 16-key tiles incl. the CLS tile
    bf16x8 q[JB][2];
    auto load_q = [&](int gb) {
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const bf16_t* qrow = q_ptr + (int64_t)((gb + j) * 16 + c) * ld + 8 * g;
            q[j][0] = *(const bf16x8*)(qrow);
            q[j][1] = *(const bf16x8*)(qrow + 32);
        }
    };
    // debug mode 3: cls_partial is a [workgroups, 4] uint64 buffer of s_memtime stamps (start / staged / computed / done) of wave 0
    unsigned long long* stamps = (DBG && dbg == 3) ? (unsigned long long*)cls_partial + (int64_t)blockIdx.x * 4 : nullptr;
    if (DBG && dbg == 3) { cls_partial = nullptr; if (tid == 0) stamps[0] = __builtin_readcyclecounter(); }
    load_q(wave * JB);                                 // in flight together with the K / V staging
    if (!DBG || dbg != 2) space_stage<NWV>(Ks, Vs, base, q_ptr, ld, ws, n, KP, lane, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (DBG && dbg == 3 && tid == 0) stamps[1] = __builtin_readcyclecounter();
    if (DBG && dbg == 1) {                             // debug: memory traffic only
        for (int gb = wave * JB; gb < nqb; gb += NWV * JB) {
            if (gb != wave * JB) load_q(gb);
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                bf16_t* op = out + ((int64_t)b * N + 1 + f * n + (gb + j) * 16 + c) * D + head * 64 + 16 * g;
                *(bf16x8*)(op) = q[j][0]; *(bf16x8*)(op + 8) = q[j][1];
            }
        }
        return;
    }
    const int kz = (c & 7) ^ (c >> 3), trq = c >> 2, trp = c & 3, vz = ((trq >> 1) & 1) << 2;
    const char* kb0 = Ks + c * 128 + ((g ^ kz) << 4);
    const char* kb1 = Ks + c * 128 + (((g + 4) ^ kz) << 4);
    const char* vb0 = Vs + (4 * g + trq) * 128 + (((2 * trp) ^ vz) << 4);
    const char* vb1 = vb0 + 8;
    const char* vb2 = Vs + (4 * g + trq) * 128 + (((2 * trp + 1) ^ vz) << 4);
    const char* vb3 = vb2 + 8;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int gb = wave * JB; gb < nqb; gb += NWV * JB) {
        if (gb != wave * JB) load_q(gb);
        f32x4 o[JB][4], ol[JB];
        float m_ref[JB];
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            ol[j] = z4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[j][dt] = z4;
        }
        spacej_chunk<JB, NTJ, false, true>(kb0, kb1, vb0, vb1, vb2, vb3, q, lane, o, ol, m_ref);
        int t0 = NTJ;
        for (; t0 + NTJ < nt; t0 += NTJ) {
            const int off = t0 * 2048;
            spacej_chunk<JB, NTJ, false, false>(kb0 + off, kb1 + off, vb0 + off, vb1 + off, vb2 + off, vb3 + off, q, lane, o, ol, m_ref);
        }
        {
            const int off = t0 * 2048;
#define SJ_FT(R) spacej_chunk<JB, (R), true, false>(kb0 + off, kb1 + off, vb0 + off, vb1 + off, vb2 + off, vb3 + off, q, lane, o, ol, m_ref)
            static_assert(NTJ % 2 == 0 && NTJ >= 2 && NTJ <= 8, "chunks of an even number of key tiles");
            switch (nt - t0) {                         // an ODD number (n / 16 is even) of 1 .. NTJ - 1 tiles left, the last one is the CLS tile
                case 1: SJ_FT(1); break;
                case 3: if (NTJ > 3) SJ_FT(NTJ > 3 ? 3 : 1); break;
                case 5: if (NTJ > 5) SJ_FT(NTJ > 5 ? 5 : 1); break;
                default: if (NTJ > 7) SJ_FT(NTJ > 7 ? 7 : 1); break;
            }
#undef SJ_FT
        }
        if (DBG && dbg == 3 && tid == 0) stamps[2] = __builtin_readcyclecounter();
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            bf16_t* op = out + ((int64_t)b * N + 1 + f * n + (gb + j) * 16 + c) * D + head * 64 + 16 * g;
            float l_run = ol[j][0];
            // a score more than 2^127 above the block's reference maximum: l is not finite -> redo this block with the running maximum
            if (__builtin_amdgcn_ballot_w64(!(l_run <= 3.0e38f)) != 0) {
                HH_SPACE_REDO_NOTE(g_space_redo, lane);
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
    }
    if (DBG && dbg == 3 && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamps[3] = __builtin_readcyclecounter(); }
    if (cls_partial == nullptr) return;
    space16_cls_partial<NWV>(Ks, Vs, scratch, base, cls_partial + (((int64_t)b * heads + head) * T + f) * CLS_REC, n, f == 0, tid, lane, wave);
}


// ---- progressive staging (round 3): one workgroup per CU at n = 576 means that NOTHING overlaps a workgroup's K / V staging --
// 148 KB per (frame, head), a third of the kernel's life.  Here the key rows are staged in four segments (issue order = key order)
// and the chunk loop starts on segment 0 while segments 1-3 are still in flight: counted s_waitcnt vmcnt(N) + one workgroup barrier
// at each segment boundary.  Two things make the counted waits hold: (1) every wave issues a compile-time number of LDS-DMA
// instructions per segment (segments 0-2 are multiples of the wave count; the ragged last one gives the first waves one piece more,
// a wave-uniform choice between two immediates -- padding it with DUPLICATE pieces instead, i.e. two LDS-DMA writes of the same
// bytes to the same LDS address in flight, corrupted the tile: measured, 5-10 % of the outputs wrong); (2) hipcc must not see a single vector-memory dependency while the DMA is in flight: its
// wait-count pass marks global_load_lds as a FLAT access to both memory and LDS ("pending flat") and from then on turns EVERY
// vmcnt wait it inserts or is given (the builtin included) into vmcnt(0) -- measured in the ISA: the first counted wait came out as
// s_waitcnt vmcnt(0) lgkmcnt(0).  So the Q rows, the counted waits and the K / V fragment reads (ds_read_b128 / ds_read_b64_tr_b16 +
// counted lgkmcnt; hipcc also puts vmcnt(0) in front of every LDS read that follows an LDS-DMA it cannot prove disjoint, see
// attn_time.hip) are opaque inline asm, each wait naming the registers it makes valid so that no consumer is scheduled above it.
#define HH_SP_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory")
// s_waitcnt vmcnt(BASE + 2 * extra), extra in {0 .. 3} wave-uniform (the immediates are compile-time, the choice a scalar branch)
template <int BASE>
__device__ __forceinline__ void sp_wait_vmcnt(int extra) {
    if (extra == 0) HH_SP_WAIT_VMCNT(BASE);
    else if (extra == 1) HH_SP_WAIT_VMCNT(BASE + 2);
    else if (extra == 2) HH_SP_WAIT_VMCNT(BASE + 4);
    else HH_SP_WAIT_VMCNT(BASE + 6);
}
__device__ __forceinline__ bf16x8 sp_gld128_raw(const bf16_t* p) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p));
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ unsigned sp_lds_u32(const char* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
template <int OFF>
__device__ __forceinline__ bf16x8 sp_rd128_raw(unsigned a) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
    return __builtin_bit_cast(bf16x8, r);
}
template <int OFF>
__device__ __forceinline__ bf16x4 sp_tr4_raw(unsigned a) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
    return __builtin_bit_cast(bf16x4, r);
}
// counted LDS wait that also "touches" the fragments it makes valid, so that no consumer is scheduled above it
#define SP_LGKM2(N, A, B) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(A), "+v"(B) : "n"(N))
#define SP_LGKM8(N, A, B, C, D, E, F, G, H) asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(A), "+v"(B), "+v"(C), "+v"(D), "+v"(E), "+v"(F), "+v"(G), "+v"(H) : "n"(N))

template <int JB, int NT, bool CLS, bool FIRST>
__device__ __forceinline__ void spacep_chunk(unsigned k0a, unsigned k1a, unsigned v0a, unsigned v1a, unsigned v2a, unsigned v3a,
                                             const bf16x8 (&q)[JB][2], int lane, f32x4 (&o)[JB][4], f32x4 (&ol)[JB], float (&m_ref)[JB]) {
    static_assert(NT == 1 || NT == 2, "chunks of one (CLS) or two key tiles");
    static_assert(!(FIRST && CLS) && (!FIRST || NT == 2), "the first chunk holds two plain key tiles");
    const int g = lane >> 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
    f32x4 s[JB][NT];
    bf16x8 ka0 = sp_rd128_raw<0>(k0a), ka1 = sp_rd128_raw<0>(k1a);
    constexpr int KB = NT == 2 ? 2048 : 0;             // (a single-tile chunk never uses kb0 / kb1)
    bf16x8 kb0 = sp_rd128_raw<KB>(k0a), kb1 = sp_rd128_raw<KB>(k1a);
    f32x4 minit[JB];
    if (!FIRST) {
#pragma unroll
        for (int j = 0; j < JB; ++j) minit[j] = (f32x4){-m_ref[j], -m_ref[j], -m_ref[j], -m_ref[j]};
    }
    SP_LGKM2(2, ka0, ka1);
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        s[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka0, q[j][0], FIRST ? z4 : minit[j], 0, 0, 0);
        s[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka1, q[j][1], s[j][0], 0, 0, 0);
    }
    if (NT == 2) {
        SP_LGKM2(0, kb0, kb1);
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            s[j][NT - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb0, q[j][0], FIRST ? z4 : minit[j], 0, 0, 0);
            s[j][NT - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb1, q[j][1], s[j][NT - 1], 0, 0, 0);
        }
    }
    // the V^T fragments of this chunk's key pair: requested now, consumed after the exponentials
    // (v1a = v0a + 8 and v3a = v2a + 8 at every call site: the second 8-byte halves are read through the immediate offset)
    (void)v1a; (void)v3a;
    bf16x4 a0 = sp_tr4_raw<0>(v0a), a1 = sp_tr4_raw<8>(v0a), a2 = sp_tr4_raw<0>(v2a), a3 = sp_tr4_raw<8>(v2a);
    // second key tile of the pair; a single-tile chunk re-reads the first (its probabilities are 0, the operand only has to be finite
    // -- a register COPY of a0 here would be taken before the read has landed)
    constexpr int OB = NT == 2 ? 2048 : 0;
    bf16x4 b0 = sp_tr4_raw<OB>(v0a), b1 = sp_tr4_raw<OB + 8>(v0a), b2 = sp_tr4_raw<OB>(v2a), b3 = sp_tr4_raw<OB + 8>(v2a);
    if (FIRST) {
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            float mx = fmaxf(fmaxf(fmaxf(s[j][0][0], s[j][0][1]), fmaxf(s[j][0][2], s[j][0][3])),
                             fmaxf(fmaxf(s[j][1][0], s[j][1][1]), fmaxf(s[j][1][2], s[j][1][3])));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            m_ref[j] = mx;
            s[j][0] -= mx;
            s[j][1] -= mx;
        }
    }
    if (CLS) {                                        // the chunk's last tile holds nothing but the CLS key (its row 0)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!(g == 0 && r == 0)) s[j][NT - 1][r] = -INFINITY;
    }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[j][ti][r] = __builtin_amdgcn_exp2f(s[j][ti][r]);
    bf16x8 pf[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const f32x4 pa = s[j][0], pb = s[j][NT - 1];
        pf[j] = (bf16x8){(bf16_t)pa[0], (bf16_t)pa[1], (bf16_t)pa[2], (bf16_t)pa[3],
                         (bf16_t)(NT == 2 ? pb[0] : 0.f), (bf16_t)(NT == 2 ? pb[1] : 0.f), (bf16_t)(NT == 2 ? pb[2] : 0.f), (bf16_t)(NT == 2 ? pb[3] : 0.f)};
    }
    SP_LGKM8(0, a0, a1, a2, a3, b0, b1, b2, b3);
    const bf16x8 af0 = {a0[0], a0[1], a0[2], a0[3], b0[0], b0[1], b0[2], b0[3]};
    const bf16x8 af1 = {a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3]};
    const bf16x8 af2 = {a2[0], a2[1], a2[2], a2[3], b2[0], b2[1], b2[2], b2[3]};
    const bf16x8 af3 = {a3[0], a3[1], a3[2], a3[3], b3[0], b3[1], b3[2], b3[3]};
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        o[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, pf[j], o[j][0], 0, 0, 0);
        o[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, pf[j], o[j][1], 0, 0, 0);
        o[j][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af2, pf[j], o[j][2], 0, 0, 0);
        o[j][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af3, pf[j], o[j][3], 0, 0, 0);
        ol[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[j], ol[j], 0, 0, 0);
    }
}

// pieces [P0, P0 + CNT) of the K tile and of the V tile (8 rows each), CPS = ceil(CNT / NWV) LDS-DMA instructions per tile and wave
template <int NWV, int P0, int CNT>
__device__ __forceinline__ void spacep_stage_seg(char* Ks, char* Vs, const bf16_t* base, const bf16_t* q_ptr, int64_t ld, int64_t ws, int n,
                                                 int lane, int wave) {
    constexpr int CPS = (CNT + NWV - 1) / NWV;
#pragma unroll
    for (int r = 0; r < CPS; ++r) {
        const int i = wave + r * NWV;
        if (i >= CNT) continue;                      // (wave-uniform; only the last segment may be ragged)
        const int pc = P0 + i, row = pc * 8 + (lane >> 3);
        const bf16_t* src = (row < n) ? q_ptr + (int64_t)row * ld : base;
        glds16(src + ws + ((lane & 7) ^ kswz(row)) * 8, Ks + pc * 1024);
    }
#pragma unroll
    for (int r = 0; r < CPS; ++r) {
        const int i = wave + r * NWV;
        if (i >= CNT) continue;
        const int pc = P0 + i, row = pc * 8 + (lane >> 3);
        const bf16_t* src = (row < n) ? q_ptr + (int64_t)row * ld : base;
        glds16(src + 2 * ws + ((lane & 7) ^ vswz(row)) * 8, Vs + pc * 1024);
    }
}

// n = 16 * NQB keys per frame, compile-time: NWV waves x JB joint blocks = all NQB query blocks at once, chunks of two key tiles,
// K / V staged in four segments of S0..S3 pieces (multiples of 4 = whole chunks; S0 + S1 + S2 + S3 = KP / 8)
template <int JB, int NWV, int WPS, int NQB, int S0, int S1, int S2, int S3>
__global__ __launch_bounds__(64 * NWV, WPS) void space_attnp_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                    float* __restrict__ cls_partial, int B, int T, int heads, int layout) {
    constexpr int n = NQB * 16, nt = NQB + 1, KP = ((n + 1 + 31) / 32) * 32, PIECES = KP / 8;
    static_assert(NWV * JB == NQB && NQB % 2 == 0, "one group of JB blocks per wave");
    static_assert(S0 + S1 + S2 + S3 == PIECES && S0 % 4 == 0 && S1 % 4 == 0 && S2 % 4 == 0 && S0 >= 4 && S1 >= 4 && S2 >= 4, "segments are whole chunks");
    static_assert(S0 % NWV == 0 && S1 % NWV == 0 && S2 % NWV == 0, "every wave stages the same number of pieces of segments 0-2 (counted waits)");
    constexpr int C0 = (S0 + NWV - 1) / NWV, C1 = (S1 + NWV - 1) / NWV, C2 = (S2 + NWV - 1) / NWV, C3 = (S3 + NWV - 1) / NWV;
    static_assert(2 * (C1 + C2 + C3) < 64, "vmcnt immediates");
    static_assert(S3 == 0 || S3 > (C3 - 1) * NWV, "ragged last segment");
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
    const int c = lane & 15, g = lane >> 4;
    const int gb = wave * JB;
    bf16x8 q[JB][2];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const bf16_t* qrow = q_ptr + (int64_t)((gb + j) * 16 + c) * ld + 8 * g;
        q[j][0] = sp_gld128_raw(qrow);
        q[j][1] = sp_gld128_raw(qrow + 32);
    }
    spacep_stage_seg<NWV, 0, S0>(Ks, Vs, base, q_ptr, ld, ws, n, lane, wave);
    spacep_stage_seg<NWV, S0, S1>(Ks, Vs, base, q_ptr, ld, ws, n, lane, wave);
    spacep_stage_seg<NWV, S0 + S1, S2>(Ks, Vs, base, q_ptr, ld, ws, n, lane, wave);
    if (S3 > 0) spacep_stage_seg<NWV, S0 + S1 + S2, S3>(Ks, Vs, base, q_ptr, ld, ws, n, lane, wave);
    // LDS-DMA instructions of THIS wave in the last (possibly ragged) segment, per tile: C3 for the first waves, C3 - 1 for the rest
    const int c3 = (S3 > 0 && wave < S3 - (C3 - 1) * NWV) ? C3 : (C3 > 0 ? C3 - 1 : 0);
    const int kz = (c & 7) ^ (c >> 3), trq = c >> 2, trp = c & 3, vz = ((trq >> 1) & 1) << 2;
    const unsigned kb0 = sp_lds_u32(Ks + c * 128 + ((g ^ kz) << 4));
    const unsigned kb1 = sp_lds_u32(Ks + c * 128 + (((g + 4) ^ kz) << 4));
    const unsigned vb0 = sp_lds_u32(Vs + (4 * g + trq) * 128 + (((2 * trp) ^ vz) << 4));
    const unsigned vb1 = vb0 + 8;
    const unsigned vb2 = sp_lds_u32(Vs + (4 * g + trq) * 128 + (((2 * trp + 1) ^ vz) << 4));
    const unsigned vb3 = vb2 + 8;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 o[JB][4], ol[JB];
    float m_ref[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        ol[j] = z4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[j][dt] = z4;
    }
    constexpr int E0 = S0 / 4, E1 = E0 + S1 / 4, E2 = E1 + S2 / 4, NCH = NQB / 2;      // chunk indices at which segments 1, 2, 3 begin; plain chunks
    static_assert(JB == 3 || JB == 4, "the first wait names the Q registers of 3 or 4 joint blocks");
    // Q rows (older than every LDS-DMA) and segment 0 have landed
    // (raw s_barrier: __syncthreads() carries a workgroup-scope release fence, for which hipcc drains every LDS-DMA -- vmcnt(0))
    static_assert(C3 <= 4, "sp_wait_vmcnt covers up to 3 extra pairs");
    sp_wait_vmcnt<2 * (C1 + C2)>(c3);
    asm volatile("" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[2][0]), "+v"(q[2][1]));      // consumers of Q stay below the wait
    if (JB == 4) asm volatile("" : "+v"(q[JB - 1][0]), "+v"(q[JB - 1][1]));
    __builtin_amdgcn_s_barrier();
    spacep_chunk<JB, 2, false, true>(kb0, kb1, vb0, vb1, vb2, vb3, q, lane, o, ol, m_ref);
#define SP_RUN(FROM, TO) for (int ci = (FROM); ci < (TO); ++ci) { const unsigned off = (unsigned)ci * 4096u;                        \
        spacep_chunk<JB, 2, false, false>(kb0 + off, kb1 + off, vb0 + off, vb1 + off, vb2 + off, vb3 + off, q, lane, o, ol, m_ref); }
    SP_RUN(1, E0 < NCH ? E0 : NCH)
    sp_wait_vmcnt<2 * C2>(c3);
    __builtin_amdgcn_s_barrier();
    SP_RUN(E0, E1 < NCH ? E1 : NCH)
    sp_wait_vmcnt<0>(c3);
    __builtin_amdgcn_s_barrier();
    SP_RUN(E1, E2 < NCH ? E2 : NCH)
    HH_SP_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    SP_RUN(E2, NCH)
#undef SP_RUN
    {
        const unsigned off = (unsigned)NCH * 4096u;     // the CLS tile
        spacep_chunk<JB, 1, true, false>(kb0 + off, kb1 + off, vb0 + off, vb1 + off, vb2 + off, vb3 + off, q, lane, o, ol, m_ref);
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        bf16_t* op = out + ((int64_t)b * N + 1 + f * n + (gb + j) * 16 + c) * D + head * 64 + 16 * g;
        float l_run = ol[j][0];
        // a score more than 2^127 above the block's reference maximum: l is not finite -> redo this block with the running maximum
        if (__builtin_amdgcn_ballot_w64(!(l_run <= 3.0e38f)) != 0) {
            HH_SPACE_REDO_NOTE(g_space_redo, lane);
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
    space16_cls_partial<NWV>(Ks, Vs, scratch, base, cls_partial + (((int64_t)b * heads + head) * T + f) * CLS_REC, n, f == 0, tid, lane, wave);
}


#define NWMAX 12        // most waves per workgroup of any variant (CLS scratch records)

// merge G partial records per (clip, head) into out row 0:  o = sum_g o_g e^{m_g - m} / sum_g l_g e^{m_g - m}
__global__ __launch_bounds__(64) void cls_combine_kernel(const float* __restrict__ partial, int G, bf16_t* __restrict__ out,
                                                         int N, int heads) {
    const int head = blockIdx.x % heads, b = blockIdx.x / heads, d = threadIdx.x;
    const float* rec = partial + ((int64_t)b * heads + head) * G * CLS_REC;
    const float LOG2E = 1.4426950408889634f;
    // statistics: lane g owns record g (G <= 64 per pass) -> the G dependent scalar loads of a serial loop become one wave max / sum.
    // The first 32 records' o columns are requested together with the statistics (they do not depend on them; only their weights do):
    // one memory round trip instead of two (round 5: 14.1 -> 8.2 us per launch alone, 18.9 -> 16.6 us in the step, where the records come
    // from HBM rather than L2)
    constexpr int PRE = 32;                       // (space: G = T = 16, time: G = n T / 128 = 32 at the headline shape)
    float opre[PRE];
#pragma unroll
    for (int g = 0; g < PRE; ++g) opre[g] = g < G ? rec[g * CLS_REC + 4 + d] : 0.f;
    float m = -INFINITY;
    for (int g0 = 0; g0 < G; g0 += 64) m = fmaxf(m, wave_max(g0 + d < G ? rec[(g0 + d) * CLS_REC] : -INFINITY));
    float l = 0.f, o = 0.f;
    for (int g0 = 0; g0 < G; g0 += 64) {
        const bool ok = g0 + d < G;
        const float e = ok ? __builtin_amdgcn_exp2f((rec[(g0 + d) * CLS_REC] - m) * LOG2E) : 0.f;
        l += wave_sum(ok ? rec[(g0 + d) * CLS_REC + 1] * e : 0.f);
        const int cnt = min(64, G - g0);
        int g = 0;
        if (g0 == 0) {
#pragma unroll
            for (; g < PRE; ++g)
                if (g < cnt) o += opre[g] * __shfl(e, g, 64);
        }
#pragma unroll 8
        for (; g < cnt; ++g) o += rec[(g0 + g) * CLS_REC + 4 + d] * __shfl(e, g, 64);     // independent coalesced loads
    }
    out[(int64_t)b * N * heads * 64 + head * 64 + d] = (bf16_t)(o / l);
}

extern "C" int hh_cls_combine(const float* partial, int G, void* out, int B, int N, int heads, hh_stream_t stream) {
    HH_REQUIRE(B >= 0 && G > 0 && N > 0 && heads > 0 && partial != nullptr, HH_ERR_SHAPE, "hh_cls_combine: bad arguments");
    if (B == 0) return HH_OK;
    hipLaunchKernelGGL(cls_combine_kernel, dim3((unsigned)(B * heads)), dim3(64), 0, (hipStream_t)stream, partial, G, (bf16_t*)out, N, heads);
    return hh_check_launch("hh_cls_combine");
}

unsigned long long hh_space32_redo_read(int reset);      // attn_space32.hip
// blocks redone on the running-maximum path since the last reset (all space-attention kernels); synchronises the device
extern "C" int64_t hh_debug_space_redo_count(int reset) {
    unsigned long long v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_space_redo), sizeof(v)) != hipSuccess) return -1;
    if (reset) { const unsigned long long z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_space_redo), &z, sizeof(z)) != hipSuccess) return -1; }
    return (int64_t)(v + hh_space32_redo_read(reset));
}

int hh_tuning_space_debug();
int hh_tuning_space_joint();
int hh_tuning_space_waves();
int hh_tuning_space_prog();
int hh_tuning_space_mfma32();
int hh_space_attn16p_launch(const void* qkv, int layout_rev, void* out, float* cls_partial, int B, int T, int n, int heads, int progressive, int dbg, hipStream_t stream);   // attn_space32.hip
int hh_space_attn32_launch(const void* qkv, int layout_rev, void* out, float* cls_partial, int B, int T, int n, int heads, int mode, int dbg, hipStream_t stream);   // attn_space32.hip
#ifndef JNT4
#define JNT4 2          // key tiles per chunk of the joint kernel at 4 / 3 / 2 query blocks per wave
#endif
#ifndef JNT3
#define JNT3 2
#endif
#ifndef JNT2
#define JNT2 6
#endif


extern "C" int hh_space_attn_fwd(const void* qkv, int qkv_layout, void* out, float* cls_partial, int B, int T, int n, int heads,
                                 hh_stream_t stream) {
    const int walk_rev = qkv_layout & HH_QKV_WALK_REVERSE;          // (bit 1: the joint-block kernel takes it in the same argument)
    qkv_layout &= ~HH_QKV_WALK_REVERSE;
    HH_REQUIRE(qkv_layout == HH_QKV_TOKEN_MAJOR || qkv_layout == HH_QKV_HEAD_MAJOR, HH_ERR_SHAPE, "hh_space_attn_fwd: bad qkv_layout");
    HH_REQUIRE(B >= 0 && T > 0 && heads > 0 && n > 0 && n % 32 == 0, HH_ERR_SHAPE, "hh_space_attn_fwd: n=%d must be a multiple of 32", n);
    HH_REQUIRE(HH_ALIGNED16(qkv) && HH_ALIGNED16(out), HH_ERR_ALIGN, "hh_space_attn_fwd: pointers must be 16-byte aligned");
    if (B == 0) return HH_OK;
    const int KP = ((n + 1 + 31) / 32) * 32;
    const size_t lds16 = (size_t)KP * 256 + ((size_t)KP + NWMAX * CLS_REC + 2 * NWMAX) * 4;
    HH_REQUIRE(lds16 <= 160 * 1024, HH_ERR_UNSUPPORTED, "hh_space_attn_fwd: n=%d needs %zu B of LDS (> 160 KiB)", n, lds16);
    static size_t attr16 = 0;
    if (lds16 > attr16) {
        hipError_t e = hipFuncSetAttribute((const void*)space_attn16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_space_attn_fwd: cannot reserve %zu B of LDS", lds16);
        attr16 = lds16;
    }
    HHProfScope prof(HH_PROF_SPACE_ATTN, 8.0 * B * (1.0 + (double)T * n) * heads * 64, (hipStream_t)stream);
    const int nqb = n >> 4, joint = hh_tuning_space_joint();
    if (joint && hh_tuning_space_mfma32() >= 2 && n == 576 && hh_tuning_space_debug() <= 2)     // round 6, opt-in: third-step pipelined 16x16x32 kernel (attn_space32.hip);
        return hh_space_attn16p_launch(qkv, qkv_layout | walk_rev, out, cls_partial, B, T, n, heads,       // 2 = with progressive staging, 3 = plain staging
                                       hh_tuning_space_mfma32() == 2, hh_tuning_space_debug(), (hipStream_t)stream);
    if (joint && hh_tuning_space_mfma32() && n % 64 == 0 && n <= 256)          // round 6: 32x32x16 MFMAs (attn_space32.hip)
        return hh_space_attn32_launch(qkv, qkv_layout | walk_rev, out, cls_partial, B, T, n, heads, hh_tuning_space_mfma32(), hh_tuning_space_debug(), (hipStream_t)stream);
    int jb = !joint ? 0 : nqb % (NWJ * 4) == 0 ? 4 : nqb % (NWJ * 3) == 0 ? 3 : nqb % (NWJ * 2) == 0 ? 2 : 0;
    if (jb) {
        typedef void (*kern_t)(const bf16_t*, bf16_t*, float*, int, int, int, int, int, int, int);
        const int dbg = hh_tuning_space_debug();
        // waves per workgroup: "space_waves" 0 = automatic -- 12 waves x 3 blocks when the K / V tiles leave room for only one
        // workgroup per CU (n >= 320) and n / 16 is a multiple of 36, else 4 waves; 4 / 12 force a variant where it divides
        // (8 waves x 2 blocks at n = 256 -- four waves per SIMD, 128 registers, 26 spilled -- was measured at 348 vs 240 us and removed)
        const int want = hh_tuning_space_waves();
        int nw = NWJ, slot = jb;
        kern_t kern = jb == 4 ? (dbg ? (kern_t)space_attnj_kernel<4, JNT4, true> : (kern_t)space_attnj_kernel<4, JNT4, false>)
                    : jb == 3 ? (kern_t)space_attnj_kernel<3, JNT3, false> : (kern_t)space_attnj_kernel<2, JNT2, false>;
        if (!dbg && nqb == 36 && (want == 12 || want == 0) && hh_tuning_space_prog()) {
            // n = 576 (config 4): 12 waves x 3 blocks, K / V staged progressively in segments of 12 / 24 / 24 / 16 pieces
            typedef void (*kernp_t)(const bf16_t*, bf16_t*, float*, int, int, int, int);
            const kernp_t kp = (kernp_t)space_attnp_kernel<3, 12, 3, 36, 12, 24, 24, 16>;
            static size_t attrp = 0;
            if (lds16 > attrp) {
                hipError_t e = hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
                HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_space_attn_fwd: cannot reserve %zu B of LDS", lds16);
                attrp = lds16;
            }
            hh_prof_note_kernel(HH_PROF_SPACE_ATTN, "space_attnp_kernel<3, 12, 3, 36, 12, 24, 24, 16>");
            hipLaunchKernelGGL(kp, dim3((unsigned)((int64_t)B * T * heads)), dim3(64 * 12), lds16, (hipStream_t)stream,
                               (const bf16_t*)qkv, (bf16_t*)out, cls_partial, B, T, heads, qkv_layout | walk_rev);
            return hh_check_launch("hh_space_attn_fwd");
        }
        if (!dbg && nqb == 16 && (want == 4 || want == 0) && hh_tuning_space_prog() == 2) {
            // experiment (space_prog = 2): n = 256, 4 waves x 4 blocks, K / V in three segments of 12 pieces
            typedef void (*kernp_t)(const bf16_t*, bf16_t*, float*, int, int, int, int);
            const kernp_t kp = (kernp_t)space_attnp_kernel<4, 4, 2, 16, 12, 12, 12, 0>;
            static size_t attrp2 = 0;
            if (lds16 > attrp2) {
                hipError_t e = hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
                HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_space_attn_fwd: cannot reserve %zu B of LDS", lds16);
                attrp2 = lds16;
            }
            hh_prof_note_kernel(HH_PROF_SPACE_ATTN, "space_attnp_kernel<4, 4, 2, 16, 12, 12, 12, 0>");
            hipLaunchKernelGGL(kp, dim3((unsigned)((int64_t)B * T * heads)), dim3(64 * 4), lds16, (hipStream_t)stream,
                               (const bf16_t*)qkv, (bf16_t*)out, cls_partial, B, T, heads, qkv_layout);
            return hh_check_launch("hh_space_attn_fwd");
        }
        if (!dbg && nqb % 36 == 0 && (want == 12 || (want == 0 && 2 * lds16 > 160 * 1024))) {
            kern = (kern_t)space_attnj_kernel<3, JNT3, false, 12, 3>; nw = 12; slot = 5;
        }
        if (dbg) slot = 1;                              // (slot of the LDS-size attribute cache)
        static size_t attrj[7] = {0, 0, 0, 0, 0, 0, 0};
        if (lds16 > attrj[slot]) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
            HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_space_attn_fwd: cannot reserve %zu B of LDS", lds16);
            attrj[slot] = lds16;
        }
        hh_prof_note_kernel(HH_PROF_SPACE_ATTN, nw == 12 ? "space_attnj_kernel<3, 2, false, 12, 3>" : jb == 4 ? (dbg ? "space_attnj_kernel<4, 2, true, 4, 2> (debug)" : "space_attnj_kernel<4, 2, false, 4, 2>")
                                                : jb == 3 ? "space_attnj_kernel<3, 2, false, 4, 2>" : "space_attnj_kernel<2, 6, false, 4, 2>");
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)B * T * heads)), dim3(64 * nw), lds16, (hipStream_t)stream,
                           (const bf16_t*)qkv, (bf16_t*)out, cls_partial, B, T, n, heads, KP, dbg, qkv_layout | walk_rev);
        return hh_check_launch("hh_space_attn_fwd");
    }
    hh_prof_note_kernel(HH_PROF_SPACE_ATTN, "space_attn16_kernel");
    hipLaunchKernelGGL(space_attn16_kernel, dim3((unsigned)((int64_t)B * T * heads)), dim3(64 * NW16), lds16, (hipStream_t)stream,
                       (const bf16_t*)qkv, (bf16_t*)out, cls_partial, B, T, n, heads, KP, hh_tuning_space_debug(), qkv_layout);
    return hh_check_launch("hh_space_attn_fwd");
}

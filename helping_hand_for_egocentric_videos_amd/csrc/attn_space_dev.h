// Device-side pieces shared by the space-attention translation units (attn_space.hip: 16x16x32 kernels + the C entry point;
// attn_space32.hip: the 32x32x16 kernels of round 6, compiled with VGPR-form MFMAs): LDS images and swizzles, staging, the running-maximum
// chunk, the CLS query's partial, the full-line block store.
#pragma once
#include "common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// 4 keys x 1 d (column-major delivery) of a row-major [key][64 d] bf16 tile: hardware-transposing LDS read.
// Lane i = 4q + p of each 16-lane group supplies the address of row q, columns 4p..4p+3; it receives column i, rows 0..3.
__device__ __forceinline__ bf16x4 lds_tr4(const char* addr) {
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
    return __builtin_bit_cast(bf16x4, r);
}

// debug / test counter: query blocks that took the running-maximum redo path (one atomic per redone block: the path is rare).  Each translation
// unit owns one counter (no relocatable device code); hh_debug_space_redo_count() sums them.
#define HH_SPACE_REDO_COUNTER(NAME) __device__ unsigned long long NAME = 0ull
#define HH_SPACE_REDO_NOTE(NAME, LANE) do { if ((LANE) == 0) atomicAdd(&NAME, 1ull); } while (0)

// layout of one CLS partial record: [m, l, 0, 0, o[64]] fp32
#define CLS_REC 68

// XOR-swizzle keys of the two LDS tiles (16-B chunk c of row r lives at position c ^ key(r)); chosen with the bank model of
// MI355X_MICROARCH.md: K rows are read as ds_read_b128 fragments by 32 consecutive rows x 2 chunks -> (r ^ (r >> 3)) & 7 is
// conflict-free (plain r & 7 is 2-way: rows 8/16/24 apart alias); V rows are read by ds_read_b64_tr_b16 as 4 rows x 64 B per
// half-wave -> flipping the 64-B half with row bit 1 is conflict-free.
// Round 2: the K key uses row bits 0-3 only ((r & 7) ^ bit 3) -- the same bank picture for the 16 consecutive rows a fragment read
// touches, but constant for a lane across key tiles, so a fragment address is lane base + tile * 2048 (an immediate offset).
__device__ __forceinline__ int kswz(int r) { return (r & 7) ^ ((r >> 3) & 1); }
__device__ __forceinline__ int vswz(int r) { return ((r >> 1) & 1) << 2; }

// stage K and V of one (clip, frame, head) problem by LDS-DMA (1 KiB = 8 rows per wave instruction); rows >= n take the CLS
// token's row (key n is the CLS key; rows > n are masked in S and multiplied by P = 0, they only have to be finite)
template <int NWV>
__device__ __forceinline__ void space_stage(char* Ks, char* Vs, const bf16_t* base, const bf16_t* q_ptr, int64_t ld, int64_t ws, int n,
                                            int KP, int lane, int wave) {
    const int pieces = KP >> 3;
    for (int pc = wave; pc < pieces; pc += NWV) {
        const int row = pc * 8 + (lane >> 3);
        const int c = (lane & 7) ^ kswz(row);
        const bf16_t* src = (row < n) ? q_ptr + (int64_t)row * ld : base;
        glds16(src + ws + c * 8, Ks + pc * 1024);
    }
    for (int pc = wave; pc < pieces; pc += NWV) {
        const int row = pc * 8 + (lane >> 3);
        const int c = (lane & 7) ^ vswz(row);
        const bf16_t* src = (row < n) ? q_ptr + (int64_t)row * ld : base;
        glds16(src + 2 * ws + c * 8, Vs + pc * 1024);
    }
}

// ---- 16-query blocks on v_mfma_f32_16x16x32_bf16, 8 waves per workgroup (4 waves per SIMD with two workgroups per CU).
// The 32-query predecessor was VALU-issue-bound at 2 waves per SIMD (SQ counters: VALU busy 49 %, MFMA 18 %).  Here a wave owns 16
// queries: a 16-key score tile is 4 accumulator registers, a chunk of 9 tiles 36 registers, the whole kernel < 128 VGPRs; Q comes
// from HBM directly in MFMA layout (lane = query, 8 d); PV contracts two 16-key tiles per MFMA (k-slots jj < 4 -> first tile row
// 4g+jj, jj >= 4 -> second tile) with V^T fetched by the transposing LDS read, d permuted so that a lane ends with 16 consecutive d
// of its query; the 1/l normalisation is applied to the 16 output registers instead of the probabilities.
#define NW16 8
#ifndef CH16
#define CH16 9
#endif
#ifndef SB_S
#define SB_S 4          // key tiles between scheduling barriers in the score loop
#endif
#ifndef SB_P
#define SB_P 2          // tile pairs between scheduling barriers in the PV loop
#endif
template <int NTC, bool CLS>
__device__ __forceinline__ void space16_chunk(const char* Ks, const char* Vs, const bf16x8 (&q)[2], int t0, int lane,
                                              f32x4 (&o)[4], float& m_run, float& l_run) {
    const int c = lane & 15, g = lane >> 4;
    const int trq = c >> 2, trp = c & 3;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 s[NTC];
#pragma unroll
    for (int ti = 0; ti < NTC; ++ti) {
        const int krow = (t0 + ti) * 16 + c;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(Ks + krow * 128 + (((g + 4 * ks) ^ kswz(krow)) << 4));
            s[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, q[ks], ks == 0 ? z4 : s[ti], 0, 0, 0);
        }
    }
    if (CLS) {                                        // the chunk's last tile holds nothing but the CLS key (its row 0)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!(g == 0 && j == 0)) s[NTC - 1][j] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int ti = 0; ti < NTC; ++ti)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[ti][j]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);              // scores are base-2 logits
    float lsum = 0.f;
#pragma unroll
    for (int ti = 0; ti < NTC; ++ti)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pv = __builtin_amdgcn_exp2f(s[ti][j] - m_new);
            s[ti][j] = pv;
            lsum += pv;
        }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    l_run = l_run * alpha + lsum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
    // O^T += V^T . P^T, two key tiles per MFMA
#pragma unroll
    for (int pr = 0; pr < (NTC + 1) / 2; ++pr) {
        const bool has_b = 2 * pr + 1 < NTC;
        const f32x4 pa = s[2 * pr], pb = s[has_b ? 2 * pr + 1 : 2 * pr];
        const bf16x8 pf = {(bf16_t)pa[0], (bf16_t)pa[1], (bf16_t)pa[2], (bf16_t)pa[3],
                           (bf16_t)(has_b ? pb[0] : 0.f), (bf16_t)(has_b ? pb[1] : 0.f), (bf16_t)(has_b ? pb[2] : 0.f), (bf16_t)(has_b ? pb[3] : 0.f)};
        const int ra = (t0 + 2 * pr) * 16 + 4 * g + trq, rb = has_b ? ra + 16 : ra;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int ch = 2 * trp + (dt >> 1), sub = (dt & 1) * 8;
            const bf16x4 a0 = lds_tr4(Vs + ra * 128 + ((ch ^ vswz(ra)) << 4) + sub);
            const bf16x4 a1 = lds_tr4(Vs + rb * 128 + ((ch ^ vswz(rb)) << 4) + sub);
            const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pf, o[dt], 0, 0, 0);
        }
    }
}


// ---- fast path (round 2).  The chunk above issues 261 vector instructions per 9 key tiles against 38 MFMAs (VALU busy 71 %, MFMA
// 24 % in profiles/r1_final_sq_summary.md): 87 of them LDS address arithmetic, 36 fma + 36 exp for the probabilities, 36 adds for
// the row sums, 19 max3.  Here
//   * fragment addresses are lane constants + tile * 2048: one v_add per fragment stream and chunk, the rest immediates;
//   * ONE reference maximum per 16-query block, taken from the first two key tiles, enters the score MFMA as its accumulator
//     initialiser (C = -m_ref), so S - m_ref comes out of the matrix core and a probability is a bare v_exp_f32; exact maths (the
//     final o / l does not depend on the reference), and safe in fp32 unless a later score exceeds that reference by more than
//     127 in base 2 -- then l is not finite and the block is redone on the running-maximum path above;
//   * the row sum l = sum_j P_j is a fifth PV MFMA against an all-ones A operand (it sums the same bf16 probabilities the PV
//     product uses).
// Left per chunk of 9 tiles: 36 exp + 18 cvt_pk + ~20 moves / address instructions against 43 MFMAs (first chunk of a block: + 13 max,
// 8 sub, 28 accumulator-initialiser moves): 182 vector instructions per 17-tile block instead of 483.
// Measured (scripts/space_probe.py, B = 32): 352 -> 331 us per call, 316 us with scheduling barriers every 4 score tiles / 2 PV pairs
// (they stop the scheduler from hoisting every LDS fragment read to the top of a chunk: 5 -> 2 spilled VGPRs at the 128-register cap).  The VALU count fell 2.65x but the kernel is no longer VALU-bound:
// staging + Q loads + O stores alone take 252 us, compute alone 251 us -- every MFMA consumes a fresh 1 KB LDS fragment (4 SIMDs x
// 1 KB / 16 clk = the 256 B/clk LDS peak), so LDS, MFMA and VALU issue are three comparable ~80-100 us streams of in-order waves.
// Tried on top of this and measured slower: streaming key tiles in pairs (16 live score registers, less ILP: 359 us) and a
// persistent 16-wave workgroup with double-buffered K/V and prefetched Q (375 us; 8 spilled VGPRs at the 128-register cap).
template <int NTC, bool CLS, bool FIRST>
__device__ __forceinline__ void space16_fast(const char* kc0, const char* kc1, const char* vc0, const char* vc1, const char* vc2, const char* vc3,
                                             const bf16x8 (&q)[2], int lane, f32x4 (&o)[4], f32x4& ol, float& m_ref) {
    static_assert(!(FIRST && CLS), "the first chunk of a block never holds the CLS tile (n >= 144 on this path)");
    const int g = lane >> 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
    f32x4 s[NTC];
    constexpr int NR = FIRST ? 2 : 0;
    if (FIRST) {
#pragma unroll
        for (int ti = 0; ti < NR; ++ti) {
            s[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kc0 + ti * 2048), q[0], z4, 0, 0, 0);
            s[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kc1 + ti * 2048), q[1], s[ti], 0, 0, 0);
        }
        float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        m_ref = mx;
#pragma unroll
        for (int ti = 0; ti < NR; ++ti) s[ti] -= mx;
    }
    const f32x4 minit = {-m_ref, -m_ref, -m_ref, -m_ref};
#pragma unroll
    for (int ti = NR; ti < NTC; ++ti) {
        s[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kc0 + ti * 2048), q[0], minit, 0, 0, 0);
        s[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kc1 + ti * 2048), q[1], s[ti], 0, 0, 0);
        if (ti % SB_S == SB_S - 1) __builtin_amdgcn_sched_barrier(0);     // keep the scheduler from hoisting every K fragment read (128-VGPR cap)
    }
    if (CLS) {                                        // the chunk's last tile holds nothing but the CLS key (its row 0)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!(g == 0 && j == 0)) s[NTC - 1][j] = -INFINITY;
    }
#pragma unroll
    for (int ti = 0; ti < NTC; ++ti)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[ti][j] = __builtin_amdgcn_exp2f(s[ti][j]);
#pragma unroll
    for (int pr = 0; pr < (NTC + 1) / 2; ++pr) {
        const bool has_b = 2 * pr + 1 < NTC;
        const f32x4 pa = s[2 * pr], pb = s[has_b ? 2 * pr + 1 : 2 * pr];
        const bf16x8 pf = {(bf16_t)pa[0], (bf16_t)pa[1], (bf16_t)pa[2], (bf16_t)pa[3],
                           (bf16_t)(has_b ? pb[0] : 0.f), (bf16_t)(has_b ? pb[1] : 0.f), (bf16_t)(has_b ? pb[2] : 0.f), (bf16_t)(has_b ? pb[3] : 0.f)};
        const int oa = 2 * pr * 2048, ob = has_b ? oa + 2048 : oa;
#define S16_PV(DT, VC) do { const bf16x4 a0 = lds_tr4(VC + oa), a1 = lds_tr4(VC + ob);                                   \
                            const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};                     \
                            o[DT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pf, o[DT], 0, 0, 0); } while (0)
        S16_PV(0, vc0); S16_PV(1, vc1); S16_PV(2, vc2); S16_PV(3, vc3);
#undef S16_PV
        ol = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, ol, 0, 0, 0);
        if (pr % SB_P == SB_P - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// CLS query (model/LaviLa.py:255-258) over THIS frame's keys on the matrix core: wave w takes the 16-key tiles w, w + NW16, ...
// (the CLS key's own tile is counted by frame 0 only) with B = q_cls replicated in all 16 columns, keeps an online-softmax partial
// (m, l, o[64]) and the workgroup merges its NW16 partials through LDS.  The VALU version (space_cls_partial) issued ~350 vector
// instructions per wave -- 36 % of this kernel's VALU work.
template <int NWV>
__device__ __forceinline__ void space16_cls_wave(const char* Ks, const char* Vs, float* scratch, const bf16x8 (&qc)[2],
                                                 int n, bool first_frame, int lane, int wave) {
    const int c = lane & 15, g = lane >> 4;
    const int trq = c >> 2, trp = c & 3;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int nfull = n >> 4, ntiles = nfull + (first_frame ? 1 : 0);
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[4] = {z4, z4, z4, z4};
    // batches of up to CB of this wave's tiles: all scores first (independent MFMAs), ONE maximum / row-sum reduction per batch,
    // then the PV products two tiles per MFMA -- at n = 256 a wave's 3-5 tiles are one batch (the tile-by-tile online softmax this
    // replaces was a ~1500-cycle dependent chain at the tail of every workgroup)
    constexpr int CB = 6;
    for (int t0 = wave; t0 < ntiles; t0 += NWV * CB) {
        f32x4 s[CB];
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int t = t0 + i * NWV;
            s[i] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (t < ntiles) {                          // wave-uniform
                const int krow = t * 16 + c;
                f32x4 a = z4;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(Ks + krow * 128 + (((g + 4 * ks) ^ kswz(krow)) << 4));
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qc[ks], a, 0, 0, 0);
                }
                if (t == nfull) {                      // CLS tile: only its row 0 is a key
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!(g == 0 && j == 0)) a[j] = -INFINITY;
                }
                s[i] = a;
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < CB; ++i) mx = fmaxf(mx, fmaxf(fmaxf(s[i][0], s[i][1]), fmaxf(s[i][2], s[i][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // base-2 logits (q carries log2 e)
        float ls = 0.f;
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[i][j] = __builtin_amdgcn_exp2f(s[i][j] - m_new); ls += s[i][j]; }
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
        for (int i = 0; i < CB; i += 2) {
            const int ta = t0 + i * NWV, tb = ta + NWV;
            if (ta >= ntiles) break;                   // wave-uniform
            const bf16x8 pf = {(bf16_t)s[i][0], (bf16_t)s[i][1], (bf16_t)s[i][2], (bf16_t)s[i][3],
                               (bf16_t)s[i + 1][0], (bf16_t)s[i + 1][1], (bf16_t)s[i + 1][2], (bf16_t)s[i + 1][3]};   // tile b missing: zeros
            const int ra = ta * 16 + 4 * g + trq, rb = (tb < ntiles ? tb : ta) * 16 + 4 * g + trq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int ch = 2 * trp + (dt >> 1), sub = (dt & 1) * 8;
                const bf16x4 a0 = lds_tr4(Vs + ra * 128 + ((ch ^ vswz(ra)) << 4) + sub);
                const bf16x4 a1 = lds_tr4(Vs + rb * 128 + ((ch ^ vswz(rb)) << 4) + sub);
                const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pf, o[dt], 0, 0, 0);
            }
        }
    }
    // per-wave partial -> LDS: [wave][m, l, -, -, o[64]]; accumulator lane (c, g) register (dt, j) = d 16 g + 4 dt + j, same for every c
    float* wrec = scratch + wave * CLS_REC;
    if (c == 0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(f32x4*)(wrec + 4 + 16 * g + 4 * dt) = o[dt];
        if (g == 0) { wrec[0] = m_run; wrec[1] = l_run; }
    }
}

// merge the NWV per-wave partials of a workgroup (one wave calls this, after a barrier) into the frame's record
template <int NWV>
__device__ __forceinline__ void space16_cls_merge(const float* scratch, float* rec, int tid) {
    {
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < NWV; ++w) m = fmaxf(m, scratch[w * CLS_REC]);
        float l = 0.f, ot = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            const float e = __builtin_amdgcn_exp2f(scratch[w * CLS_REC] - m);
            l += scratch[w * CLS_REC + 1] * e;
            ot += scratch[w * CLS_REC + 4 + tid] * e;
        }
        rec[4 + tid] = ot;
        if (tid == 0) { rec[0] = m * 0.6931471805599453f; rec[1] = l; }      // the record's maximum is in natural-log units (hh_cls_combine)
    }
}

template <int NWV>
__device__ __forceinline__ void space16_cls_partial(const char* Ks, const char* Vs, float* scratch, const bf16_t* base, float* rec,
                                                    int n, bool first_frame, int tid, int lane, int wave) {
    bf16x8 qc[2];                                     // the CLS query of this clip and head (row 0 of the clip)
    qc[0] = *(const bf16x8*)(base + 8 * (lane >> 4));
    qc[1] = *(const bf16x8*)(base + 8 * (lane >> 4) + 32);
    space16_cls_wave<NWV>(Ks, Vs, scratch, qc, n, first_frame, lane, wave);
    __syncthreads();
    if (tid < 64) space16_cls_merge<NWV>(scratch, rec, tid);
}

// op: this lane's 16 columns of query row c (out + row * D + head * 64 + 16 g); the block leaves as two stores of 8 full 128-B lines
// (common.h: hh_fullline_swap) -- called with all 64 lanes active
__device__ __forceinline__ void space_store_block(const f32x4 (&o)[4], float l, bf16_t* op, int c, int64_t D) {
    const float inv = 1.f / l;
    const u32x4 w0 = {pack_bf16(o[0][0] * inv, o[0][1] * inv), pack_bf16(o[0][2] * inv, o[0][3] * inv),
                      pack_bf16(o[1][0] * inv, o[1][1] * inv), pack_bf16(o[1][2] * inv, o[1][3] * inv)};
    const u32x4 w1 = {pack_bf16(o[2][0] * inv, o[2][1] * inv), pack_bf16(o[2][2] * inv, o[2][3] * inv),
                      pack_bf16(o[3][0] * inv, o[3][1] * inv), pack_bf16(o[3][2] * inv, o[3][3] * inv)};
    u32x4 x, y;
    hh_fullline_swap(w0, w1, x, y);
    bf16_t* p0 = op - (int64_t)(c >> 3) * 8 * D + 8 * (c >> 3);       // row c & 7, piece 2 g + (c >> 3)
    *(u32x4*)(p0) = x;
    *(u32x4*)(p0 + 8 * D) = y;
}


// Space attention core of the divided space-time attention (model/LaviLa.py:246-283 with '(b f) n d', attn() :194-198).
// One workgroup per (clip, frame, head): n queries x (n frame keys + CLS key), head dim 64, no mask, no dropout.
// HBM-bound in principle: algorithmic bytes per workgroup = (3*n + 2) * 128 B read + n * 128 B written.
//
// Structure: K tile and V tile, both row-major [KP][64] with XOR-swizzled 16-B chunks, staged once in LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no VALU), then each wave walks 32-query blocks:
//   S^T = K . Q^T      mfma_f32_32x32x16_bf16, A = K rows (LDS), B = Q rows (registers, straight from HBM)
//   softmax over keys  in-lane over the 16 accumulator registers x tiles + one xor-32 shuffle (query = lane & 31)
//   O^T = V^T . P^T    the S^T accumulator tile is re-used as the B operand with no lane movement
//                      (k order inside a step: row 16s + 8(j>>2) + 4h + (j&3)); A = V^T fragments fetched from the
//                      row-major V tile by the hardware-transposing ds_read_b64_tr_b16
// Keys are processed in chunks of CH=5 tiles (160 keys; CH = 4 / 6 / 8 measured slower: 433 / 441 / 487 us vs 418 us, 8 spills)
// with an online-softmax merge between chunks, so n=576
// (336^2) runs through the same code.  Frame keys sit at rows 0..n-1, the CLS key at row n, rows > n are zero/masked.
#include "common.h"
#include <stdlib.h>

#ifndef CH
#define CH 5
#endif
#define NW 4            // waves per workgroup

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

// layout of one CLS partial record: [m, l, 0, 0, o[64]] fp32
#define CLS_REC 68

// XOR-swizzle keys of the two LDS tiles (16-B chunk c of row r lives at position c ^ key(r)); chosen with the bank model of
// MI355X_MICROARCH.md: K rows are read as ds_read_b128 fragments by 32 consecutive rows x 2 chunks -> (r ^ (r >> 3)) & 7 is
// conflict-free (plain r & 7 is 2-way: rows 8/16/24 apart alias); V rows are read by ds_read_b64_tr_b16 as 4 rows x 64 B per
// half-wave -> flipping the 64-B half with row bit 1 is conflict-free.
__device__ __forceinline__ int kswz(int r) { return (r ^ (r >> 3)) & 7; }
__device__ __forceinline__ int vswz(int r) { return ((r >> 1) & 1) << 2; }

// One chunk of key tiles for one 32-query block: NTC full 32-key tiles starting at tile t0 and, if CLS, the tile that holds
// nothing but the CLS key (n % 32 == 0: key n is row 0 of tile n/32).  Everything is unrolled and unguarded: no zero-filled
// accumulators (the first MFMA of a tile takes a constant-zero C), no per-register key masks (the CLS tile contributes ONE
// score, register 0 of the h = 0 half-wave, and ONE k-slot of one PV step), no empty tile slots.  The guarded predecessor
// spent 1420 VALU instructions per query block (SQ_INSTS_VALU), ~60 % of the SIMD issue time of the whole kernel.
template <int NTC, bool CLS>
__device__ __forceinline__ void space_chunk(const char* Ks, const char* Vs, const bf16x8 (&qf)[4], int t0, int lane,
                                            f32x16& o0, f32x16& o1, float& m_run, float& l_run) {
    const int ql = lane & 31, h = lane >> 5;
    const float LOG2E = 1.4426950408889634f;
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    f32x16 z16;
#pragma unroll
    for (int r = 0; r < 16; ++r) z16[r] = 0.f;
    f32x16 s[NTC > 0 ? NTC : 1];
#pragma unroll
    for (int ti = 0; ti < NTC; ++ti) {
        const int krow = (t0 + ti) * 32 + ql;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = 2 * ks + h;
            bf16x8 kf = *(const bf16x8*)(Ks + krow * 128 + ((c ^ kswz(krow)) << 4));
            s[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? z16 : s[ti], 0, 0, 0);
        }
    }
    float xc = -INFINITY;
    if (CLS) {
        const int krow = (t0 + NTC) * 32 + ql;
        f32x16 sc = z16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = 2 * ks + h;
            bf16x8 kf = *(const bf16x8*)(Ks + krow * 128 + ((c ^ kswz(krow)) << 4));
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sc, 0, 0, 0);
        }
        xc = h == 0 ? sc[0] : -INFINITY;                  // key n = tile row 0 = register 0 of the lower half-wave
    }
    float mx = xc;
#pragma unroll
    for (int ti = 0; ti < NTC; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[ti][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
    const float mb = m_new * LOG2E;
    float lsum = 0.f;
#pragma unroll
    for (int ti = 0; ti < NTC; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(s[ti][r] * LOG2E - mb);
            s[ti][r] = pv;
            lsum += pv;
        }
    float pc = 0.f;
    if (CLS) {
        pc = __builtin_amdgcn_exp2f(xc * LOG2E - mb);     // 0 in the upper half-wave
        lsum += pc;
    }
    lsum += __shfl_xor(lsum, 32, 64);
    l_run = l_run * alpha + lsum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    // O^T += V^T . P^T ; A operand element j of lane (d, h) = V[key0 + 8(j>>2) + (j&3)][d], key0 = 32 tile + 16 st + 4 h
    auto vfrag = [&](int kb, int dt) -> bf16x8 {
        const int col = 32 * dt + 16 * tg + 4 * tp;               // d column of this lane's address
        const int ch = col >> 3, sub = (col & 7) * 2;
        bf16x4 a0 = lds_tr4(Vs + kb * 128 + ((ch ^ vswz(kb)) << 4) + sub);
        bf16x4 a1 = lds_tr4(Vs + (kb + 8) * 128 + ((ch ^ vswz(kb + 8)) << 4) + sub);
        return (bf16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    };
#pragma unroll
    for (int ti = 0; ti < NTC; ++ti) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 pf;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pf[jj] = (bf16_t)s[ti][8 * st + jj];
            const int kb = (t0 + ti) * 32 + 16 * st + 4 * h + tq;            // this lane's address row (first 4-key group)
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(kb, 0), pf, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(kb, 1), pf, o1, 0, 0, 0);
        }
    }
    if (CLS) {
        const bf16x8 pf = {(bf16_t)pc, 0, 0, 0, 0, 0, 0, 0};      // k-slot 0 of h = 0 is key 32 tile + 0 = n
        const int kb = (t0 + NTC) * 32 + 4 * h + tq;
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(kb, 0), pf, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(kb, 1), pf, o1, 0, 0, 0);
    }
}

// One 32-query block of one (clip, frame, head) problem: S^T = K.Q^T, online softmax over CH-tile chunks, O^T = V^T.P^T.
// nf = n / 32 full key tiles, followed by the CLS tile (which rides on the last chunk).
__device__ __forceinline__ void space_query_block(const char* Ks, const char* Vs, const bf16x8 (&qf)[4], bf16_t* orow, int nf, int lane) {
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    int t0 = 0;
    for (; t0 + CH < nf; t0 += CH) space_chunk<CH, false>(Ks, Vs, qf, t0, lane, o0, o1, m_run, l_run);
    switch (nf - t0) {                                        // 1 .. CH tiles left (nf >= 1), plus the CLS tile
        case 1: space_chunk<1, true>(Ks, Vs, qf, t0, lane, o0, o1, m_run, l_run); break;
        case 2: space_chunk<2, true>(Ks, Vs, qf, t0, lane, o0, o1, m_run, l_run); break;
        case 3: space_chunk<3, true>(Ks, Vs, qf, t0, lane, o0, o1, m_run, l_run); break;
#if CH >= 5
        case 4: space_chunk<4, true>(Ks, Vs, qf, t0, lane, o0, o1, m_run, l_run); break;
#endif
#if CH >= 6
        case 5: space_chunk<5, true>(Ks, Vs, qf, t0, lane, o0, o1, m_run, l_run); break;
#endif
#if CH >= 7
        case 6: space_chunk<6, true>(Ks, Vs, qf, t0, lane, o0, o1, m_run, l_run); break;
#endif
#if CH >= 8
        case 7: space_chunk<7, true>(Ks, Vs, qf, t0, lane, o0, o1, m_run, l_run); break;
#endif
        default: space_chunk<CH, true>(Ks, Vs, qf, t0, lane, o0, o1, m_run, l_run); break;
    }
    const float inv = 1.f / l_run;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        u32x2 w0 = {pack_bf16(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack_bf16(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv)};
        u32x2 w1 = {pack_bf16(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack_bf16(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv)};
        *(u32x2*)(orow + 8 * g) = w0;
        *(u32x2*)(orow + 32 + 8 * g) = w1;
    }
}

// CLS query (model/LaviLa.py:255-258) folded in: partial softmax(q_cls . K_f^T) V_f over THIS frame's keys (already in LDS;
// the CLS key itself is counted by frame 0 only); hh_cls_combine merges the T partials.  All NWV waves participate.
template <int NWV>
__device__ __forceinline__ void space_cls_partial(const char* Ks, const char* Vs, float* scratch, int KP, const bf16_t* base,
                                                  float* rec, int n, bool first_frame, int tid, int lane, int wave) {
    const float LOG2E = 1.4426950408889634f;
    float* cs = scratch;                      // [KP] scores / probabilities
    float* wrec = scratch + KP;               // [NWV][CLS_REC] per-wave partial o, + 2*NWV reduction slots
    float* red = wrec + NWV * CLS_REC;
    const int nkeys = n + (first_frame ? 1 : 0);
    float mx = -INFINITY;
    for (int j = tid; j < nkeys; j += 64 * NWV) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            u32x4 qv = *(const u32x4*)(base + c * 8);                               // q row of token 0 (uniform, pre-scaled)
            u32x4 u = *(const u32x4*)(Ks + j * 128 + ((c ^ kswz(j)) << 4));
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a0 = fmaf(bf16_lo_to_f32(qv[w]), bf16_lo_to_f32(u[w]), a0);
                a1 = fmaf(bf16_hi_to_f32(qv[w]), bf16_hi_to_f32(u[w]), a1);
            }
        }
        cs[j] = a0 + a1;
        mx = fmaxf(mx, a0 + a1);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, red[w]);
    float l = 0.f;
    for (int j = tid; j < nkeys; j += 64 * NWV) {
        const float pj = __builtin_amdgcn_exp2f((cs[j] - mx) * LOG2E);
        cs[j] = pj;
        l += pj;
    }
    l = wave_sum(l);
    if (lane == 0) red[NWV + wave] = l;
    __syncthreads();
    float o = 0.f;
    const int dch = lane >> 3, dsub = (lane & 7) * 2;
    for (int j = wave; j < nkeys; j += NWV) {
        const unsigned short vv = *(const unsigned short*)(Vs + j * 128 + ((dch ^ vswz(j)) << 4) + dsub);
        o = fmaf(cs[j], __uint_as_float((unsigned)vv << 16), o);
    }
    wrec[wave * CLS_REC + lane] = o;
    __syncthreads();
    if (tid < 64) {
        float ot = 0.f, lt = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { ot += wrec[w * CLS_REC + tid]; lt += red[NWV + w]; }
        rec[4 + tid] = ot;
        if (tid == 0) { rec[0] = mx; rec[1] = lt; }
    }
}

// stage K and V of one (clip, frame, head) problem by LDS-DMA (1 KiB = 8 rows per wave instruction); rows >= n take the CLS
// token's row (key n is the CLS key; rows > n are masked in S and multiplied by P = 0, they only have to be finite)
template <int NWV>
__device__ __forceinline__ void space_stage(char* Ks, char* Vs, const bf16_t* base, const bf16_t* q_ptr, int64_t ld, int D, int n,
                                            int KP, int lane, int wave) {
    const int pieces = KP >> 3;
    for (int pc = wave; pc < pieces; pc += NWV) {
        const int row = pc * 8 + (lane >> 3);
        const int c = (lane & 7) ^ kswz(row);
        const bf16_t* src = (row < n) ? q_ptr + (int64_t)row * ld : base;
        glds16(src + D + c * 8, Ks + pc * 1024);
    }
    for (int pc = wave; pc < pieces; pc += NWV) {
        const int row = pc * 8 + (lane >> 3);
        const int c = (lane & 7) ^ vswz(row);
        const bf16_t* src = (row < n) ? q_ptr + (int64_t)row * ld : base;
        glds16(src + 2 * D + c * 8, Vs + pc * 1024);
    }
}

// ---- variant A: one workgroup per problem (any n with (n+1) keys fitting LDS once), 4 waves
__global__ __launch_bounds__(64 * NW, 2) void space_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ cls_partial,
                                                            int B, int T, int n, int heads, int KP) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                                   // [KP][128 B], 16-B chunk c of row r at position c ^ (r & 7)
    char* Vs = smem + (size_t)KP * 128;                // same layout, row-major V
    float* scratch = (float*)(smem + (size_t)KP * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = heads * 64;
    const int64_t ld = 3 * (int64_t)D;
    const int N = 1 + T * n;
    int bid = blockIdx.x;
    const int head = bid % heads; bid /= heads;
    const int f = bid % T;
    const int b = bid / T;
    const bf16_t* base = qkv + (int64_t)b * N * ld + head * 64;      // token 0 (CLS) of this clip / head
    const bf16_t* q_ptr = base + (int64_t)(1 + f * n) * ld;
    space_stage<NW>(Ks, Vs, base, q_ptr, ld, D, n, KP, lane, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int ql = lane & 31, h = lane >> 5;
    for (int qb = wave; qb < (n >> 5); qb += NW) {
        bf16x8 qf[4];
        const bf16_t* qrow = q_ptr + (int64_t)(qb * 32 + ql) * ld + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qrow + 16 * ks);
        space_query_block(Ks, Vs, qf, out + ((int64_t)b * N + 1 + f * n + qb * 32 + ql) * D + head * 64 + 4 * h, n >> 5, lane);
    }
    if (cls_partial == nullptr) return;
    space_cls_partial<NW>(Ks, Vs, scratch, KP, base, cls_partial + (((int64_t)b * heads + head) * T + f) * CLS_REC, n, f == 0, tid, lane, wave);
}

// ---- variant C: 16-query blocks on v_mfma_f32_16x16x32_bf16, 8 waves per workgroup (4 waves per SIMD with two workgroups per CU).
// The 32-query kernel above is VALU-issue-bound at 2 waves per SIMD (SQ counters: VALU busy 49 %, MFMA 18 %).  Here a wave owns 16
// queries: a 16-key score tile is 4 accumulator registers, a chunk of 9 tiles 36 registers, the whole kernel < 128 VGPRs; Q comes
// from HBM directly in MFMA layout (lane = query, 8 d); PV contracts two 16-key tiles per MFMA (k-slots jj < 4 -> first tile row
// 4g+jj, jj >= 4 -> second tile) with V^T fetched by the transposing LDS read, d permuted so that a lane ends with 16 consecutive d
// of its query; the 1/l normalisation is applied to the 16 output registers instead of the probabilities.
#define NW16 8
#define CH16 9
template <int NTC, bool CLS>
__device__ __forceinline__ void space16_chunk(const char* Ks, const char* Vs, const bf16x8 (&q)[2], int t0, int lane,
                                              f32x4 (&o)[4], float& m_run, float& l_run) {
    const int c = lane & 15, g = lane >> 4;
    const int trq = c >> 2, trp = c & 3;
    const float LOG2E = 1.4426950408889634f;
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
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
    const float mb = m_new * LOG2E;
    float lsum = 0.f;
#pragma unroll
    for (int ti = 0; ti < NTC; ++ti)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pv = __builtin_amdgcn_exp2f(s[ti][j] * LOG2E - mb);
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

// CLS query (model/LaviLa.py:255-258) over THIS frame's keys on the matrix core: wave w takes the 16-key tiles w, w + NW16, ...
// (the CLS key's own tile is counted by frame 0 only) with B = q_cls replicated in all 16 columns, keeps an online-softmax partial
// (m, l, o[64]) and the workgroup merges its NW16 partials through LDS.  The VALU version (space_cls_partial) issued ~350 vector
// instructions per wave -- 36 % of this kernel's VALU work.
__device__ __forceinline__ void space16_cls_partial(const char* Ks, const char* Vs, float* scratch, const bf16_t* base, float* rec,
                                                    int n, bool first_frame, int tid, int lane, int wave) {
    const int c = lane & 15, g = lane >> 4;
    const int trq = c >> 2, trp = c & 3;
    const float LOG2E = 1.4426950408889634f;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    bf16x8 qc[2];
    qc[0] = *(const bf16x8*)(base + 8 * g);
    qc[1] = *(const bf16x8*)(base + 8 * g + 32);
    const int nfull = n >> 4, ntiles = nfull + (first_frame ? 1 : 0);
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[4] = {z4, z4, z4, z4};
    for (int t = wave; t < ntiles; t += NW16) {
        const int krow = t * 16 + c;
        f32x4 s = z4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(Ks + krow * 128 + (((g + 4 * ks) ^ kswz(krow)) << 4));
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qc[ks], s, 0, 0, 0);
        }
        if (t == nfull) {                              // CLS tile: only its row 0 is a key
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!(g == 0 && j == 0)) s[j] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        const float mb = m_new * LOG2E;
        float p[4], ls = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { p[j] = __builtin_amdgcn_exp2f(s[j] * LOG2E - mb); ls += p[j]; }
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        l_run = l_run * alpha + ls;
        m_run = m_new;
        const bf16x8 pf = {(bf16_t)p[0], (bf16_t)p[1], (bf16_t)p[2], (bf16_t)p[3], 0, 0, 0, 0};
        const int ra = t * 16 + 4 * g + trq;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int ch = 2 * trp + (dt >> 1), sub = (dt & 1) * 8;
            const bf16x4 a0 = lds_tr4(Vs + ra * 128 + ((ch ^ vswz(ra)) << 4) + sub);
            const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a0[0], a0[1], a0[2], a0[3]};      // second half meets zero probabilities
            o[dt] *= alpha;
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pf, o[dt], 0, 0, 0);
        }
    }
    // per-wave partial -> LDS: [wave][m, l, -, -, o[64]]; accumulator lane (c, g) register (dt, j) = d 16 g + 4 dt + j, same for every c
    float* wrec = scratch + wave * CLS_REC;
    if (c == 0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(f32x4*)(wrec + 4 + 16 * g + 4 * dt) = o[dt];
        if (g == 0) { wrec[0] = m_run; wrec[1] = l_run; }
    }
    __syncthreads();
    if (tid < 64) {
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW16; ++w) m = fmaxf(m, scratch[w * CLS_REC]);
        float l = 0.f, ot = 0.f;
#pragma unroll
        for (int w = 0; w < NW16; ++w) {
            const float e = __builtin_amdgcn_exp2f((scratch[w * CLS_REC] - m) * LOG2E);
            l += scratch[w * CLS_REC + 1] * e;
            ot += scratch[w * CLS_REC + 4 + tid] * e;
        }
        rec[4 + tid] = ot;
        if (tid == 0) { rec[0] = m; rec[1] = l; }
    }
}

__global__ __launch_bounds__(64 * NW16, 4) void space_attn16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                  float* __restrict__ cls_partial, int B, int T, int n, int heads, int KP) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + (size_t)KP * 128;
    float* scratch = (float*)(smem + (size_t)KP * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = heads * 64;
    const int64_t ld = 3 * (int64_t)D;
    const int N = 1 + T * n;
    int bid = blockIdx.x;
    const int head = bid % heads; bid /= heads;
    const int f = bid % T;
    const int b = bid / T;
    const bf16_t* base = qkv + (int64_t)b * N * ld + head * 64;
    const bf16_t* q_ptr = base + (int64_t)(1 + f * n) * ld;
    space_stage<NW16>(Ks, Vs, base, q_ptr, ld, D, n, KP, lane, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int c = lane & 15, g = lane >> 4;
    const int nt = (n >> 4) + 1;                       // 16-key tiles incl. the CLS tile
    for (int qb = wave; qb < (n >> 4); qb += NW16) {
        bf16x8 q[2];
        const bf16_t* qrow = q_ptr + (int64_t)(qb * 16 + c) * ld + 8 * g;
        q[0] = *(const bf16x8*)(qrow);
        q[1] = *(const bf16x8*)(qrow + 32);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 o[4] = {z4, z4, z4, z4};
        float m_run = -INFINITY, l_run = 0.f;
        int t0 = 0;
        for (; t0 + CH16 < nt; t0 += CH16) space16_chunk<CH16, false>(Ks, Vs, q, t0, lane, o, m_run, l_run);
#define S16_TAIL(R) space16_chunk<R, true>(Ks, Vs, q, t0, lane, o, m_run, l_run)
        switch (nt - t0) {                             // 1 .. CH16 tiles left, the last one is the CLS tile
            case 1: S16_TAIL(1); break;
            case 2: S16_TAIL(2); break;
            case 3: S16_TAIL(3); break;
            case 4: S16_TAIL(4); break;
            case 5: S16_TAIL(5); break;
            case 6: S16_TAIL(6); break;
            case 7: S16_TAIL(7); break;
            case 8: S16_TAIL(8); break;
            default: S16_TAIL(9); break;
        }
#undef S16_TAIL
        const float inv = 1.f / l_run;
        bf16_t* op = out + ((int64_t)b * N + 1 + f * n + qb * 16 + c) * D + head * 64 + 16 * g;
        const u32x4 w0 = {pack_bf16(o[0][0] * inv, o[0][1] * inv), pack_bf16(o[0][2] * inv, o[0][3] * inv),
                          pack_bf16(o[1][0] * inv, o[1][1] * inv), pack_bf16(o[1][2] * inv, o[1][3] * inv)};
        const u32x4 w1 = {pack_bf16(o[2][0] * inv, o[2][1] * inv), pack_bf16(o[2][2] * inv, o[2][3] * inv),
                          pack_bf16(o[3][0] * inv, o[3][1] * inv), pack_bf16(o[3][2] * inv, o[3][3] * inv)};
        *(u32x4*)(op) = w0;
        *(u32x4*)(op + 8) = w1;
    }
    if (cls_partial == nullptr) return;
    space16_cls_partial(Ks, Vs, scratch, base, cls_partial + (((int64_t)b * heads + head) * T + f) * CLS_REC, n, f == 0, tid, lane, wave);
}

// merge G partial records per (clip, head) into out row 0:  o = sum_g o_g e^{m_g - m} / sum_g l_g e^{m_g - m}
__global__ __launch_bounds__(64) void cls_combine_kernel(const float* __restrict__ partial, int G, bf16_t* __restrict__ out,
                                                         int N, int heads) {
    const int head = blockIdx.x % heads, b = blockIdx.x / heads, d = threadIdx.x;
    const float* rec = partial + ((int64_t)b * heads + head) * G * CLS_REC;
    const float LOG2E = 1.4426950408889634f;
    // statistics: lane g owns record g (G <= 64 per pass) -> the G dependent scalar loads of a serial loop become one wave max / sum
    float m = -INFINITY;
    for (int g0 = 0; g0 < G; g0 += 64) m = fmaxf(m, wave_max(g0 + d < G ? rec[(g0 + d) * CLS_REC] : -INFINITY));
    float l = 0.f, o = 0.f;
    for (int g0 = 0; g0 < G; g0 += 64) {
        const bool ok = g0 + d < G;
        const float e = ok ? __builtin_amdgcn_exp2f((rec[(g0 + d) * CLS_REC] - m) * LOG2E) : 0.f;
        l += wave_sum(ok ? rec[(g0 + d) * CLS_REC + 1] * e : 0.f);
        const int cnt = min(64, G - g0);
#pragma unroll 8
        for (int g = 0; g < cnt; ++g) o += rec[(g0 + g) * CLS_REC + 4 + d] * __shfl(e, g, 64);     // independent coalesced loads
    }
    out[(int64_t)b * N * heads * 64 + head * 64 + d] = (bf16_t)(o / l);
}

extern "C" int hh_cls_combine(const float* partial, int G, void* out, int B, int N, int heads, hh_stream_t stream) {
    HH_REQUIRE(B >= 0 && G > 0 && N > 0 && heads > 0 && partial != nullptr, HH_ERR_SHAPE, "hh_cls_combine: bad arguments");
    if (B == 0) return HH_OK;
    hipLaunchKernelGGL(cls_combine_kernel, dim3((unsigned)(B * heads)), dim3(64), 0, (hipStream_t)stream, partial, G, (bf16_t*)out, N, heads);
    return hh_check_launch("hh_cls_combine");
}

extern "C" int hh_space_attn_fwd(const void* qkv, void* out, float* cls_partial, int B, int T, int n, int heads, hh_stream_t stream) {
    HH_REQUIRE(B >= 0 && T > 0 && heads > 0 && n > 0 && n % 32 == 0, HH_ERR_SHAPE, "hh_space_attn_fwd: n=%d must be a multiple of 32", n);
    HH_REQUIRE(HH_ALIGNED16(qkv) && HH_ALIGNED16(out), HH_ERR_ALIGN, "hh_space_attn_fwd: pointers must be 16-byte aligned");
    if (B == 0) return HH_OK;
    const int KP = ((n + 1 + 31) / 32) * 32;
    // HH_SPACE_ATTN: 2 (default) = 16-query blocks, 8 waves per workgroup (365 us/call at B = 32); 0 = 32-query blocks on
    // 32x32x16 MFMAs, 4 waves per workgroup (420 us).  A persistent double-buffered variant of the latter was measured at 480 us
    // and removed.
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("HH_SPACE_ATTN"); mode = e ? atoi(e) : 2; }
    const size_t lds16 = (size_t)KP * 256 + ((size_t)KP + NW16 * CLS_REC + 2 * NW16) * 4;
    if (mode == 2 && lds16 <= 160 * 1024) {
        static size_t attr16 = 0;
        if (lds16 > attr16) {
            hipError_t e = hipFuncSetAttribute((const void*)space_attn16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
            HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_space_attn_fwd: cannot reserve %zu B of LDS", lds16);
            attr16 = lds16;
        }
        hipLaunchKernelGGL(space_attn16_kernel, dim3((unsigned)((int64_t)B * T * heads)), dim3(64 * NW16), lds16, (hipStream_t)stream,
                           (const bf16_t*)qkv, (bf16_t*)out, cls_partial, B, T, n, heads, KP);
        return hh_check_launch("hh_space_attn_fwd(16-query blocks)");
    }
    const size_t lds = (size_t)KP * 256 + ((size_t)KP + NW * CLS_REC + 2 * NW) * 4;
    HH_REQUIRE(lds <= 160 * 1024, HH_ERR_UNSUPPORTED, "hh_space_attn_fwd: n=%d needs %zu B of LDS (> 160 KiB)", n, lds);
    static size_t attr_set = 0;
    if (lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)space_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_space_attn_fwd: cannot reserve %zu B of LDS", lds);
        attr_set = lds;
    }
    const int64_t blocks = (int64_t)B * T * heads;
    hipLaunchKernelGGL(space_attn_kernel, dim3((unsigned)blocks), dim3(64 * NW), lds, (hipStream_t)stream,
                       (const bf16_t*)qkv, (bf16_t*)out, cls_partial, B, T, n, heads, KP);
    return hh_check_launch("hh_space_attn_fwd");
}

// Decoder cross-attention WITHOUT the memory-side K/V projections ("memory-space attention"; nn.MultiheadAttention inside
// TransformerDecoderLayer.forward_pre, /root/reference/model/tfm_decoder.py:433-441, round 5).
//
// The reference projects all M = T*n memory tokens to K_l = (memory + pos) Wk_l^T + bk_l and V_l = memory Wv_l^T + bv_l for each of
// the six layers (825 GFLOP forward at B = 32, twice that backward, a 1.6 GB bf16 buffer written once and read three times).  But
//     q_h . K_l[i]          = (q_h Wk_l[h]) . (memory + pos)[i]  +  q_h . bk_l[h]        (the bias is constant over the keys of a query:
//                                                                                        it drops out of the softmax)
//     sum_i p_i V_l[i]      = (sum_i p_i memory[i]) Wv_l[h]^T  +  (sum_i p_i) bv_l[h]
// so the 13 query rows of a head are mapped INTO memory space (qt = q_h Wk_l[h], 512 wide), attend over the un-projected rows
// mp = memory + pos / mem = memory that all layers and heads share, and the value projection is applied to the 13 pooled rows
// afterwards (both by head-batched hh_qgemm_f32x3 launches on the query side).  This file is the attention core at d = 512:
//
//   hh_mattn_fwd   pooled[r, h, :] = sum_i Pd[r, h, i] mem[i, :],  lse2,  rsum = sum_i Pd   (Pd = dropout(softmax(qt . mp^T)))
//   hh_mattn_bwd   dqt[r, h, :]    = sum_i dS[r, h, i] mp[i, :]    and the transposed probabilities / score gradients Pd^T, dS^T
//                  (bf16 [B, rows, M], rows = layer*128 + head*16 + query) from which ONE batched TN GEMM after the last layer makes
//                  d memory = sum_layers Pd^T dpooled + dS^T qt  (gemm_tn.hip: hh_gemm_tn_bf16_batched2) -- the layers' contributions
//                  to d memory are never accumulated through HBM.
//
// Work decomposition (both kernels): one workgroup of 8 waves per (clip, group of 4 heads, key slice); wave w owns head 4 hg + (w >> 1)
// -- 16 query slots, Q <= 16 real -- and HALF of the 512 contraction / output dims (w & 1).  Keys are streamed in chunks of 32 rows:
// the chunk's mp and mem rows (64 KB) are staged once per workgroup by LDS-DMA (one 1-KB row per wave instruction, XOR-swizzled
// through the per-lane source address) into a double buffer and feed all 4 heads;
//   S^T partial = mp[32 keys, my 256 dims] . qt^T      16 row reads (ds_read_b128) x 2 MFMAs (qt enters as a bf16 hi + lo pair)
//   the two dim-halves of a head swap their partials through LDS (2 KB per wave) and both hold S^T = the B operand layout of
//   O^T += mem^T[my 256 dims, 32 keys] . Pd^T          32 transposing reads (ds_read_b64_tr_b16) x 2 MFMAs (Pd as hi + lo)
// Every product is exact with respect to the stored bf16 memory rows (query-side values as hi + lo pairs: x = hi + lo + O(2^-17 x)),
// as in xattn.hip.  Per chunk and wave 64 MFMAs (forward) / 96 (backward) against 16 KB / 24 KB of LDS fragment reads; the same
// 64 KB of staging serve 4 heads, so the kernels sit near the balance point of HBM (2 KB per key and layer) and the matrix cores.
// Bank model of MI355X_MICROARCH.md: 16-B chunk c of row r lives at position c ^ (2 (r & 7)); both the row reads (16 keys x 4
// chunks per instruction) and the transposed reads (8 rows x 32 B per half wave) are conflict-free with it.
//
// hipcc and LDS-DMA (see attn_space.hip): only the DMA and its waits are inline asm (below: ma_dma_row); the LDS reads are ordinary loads the
// compiler schedules -- it never sees the global_load_lds builtin, so it has no reason to turn its own waits into vmcnt(0).
#include "common.h"

typedef short ma_s16x4 __attribute__((ext_vector_type(4)));

#define MA_C 512                       // decoder width = contraction length of the scores = width of the pooled rows
#define MA_H 8                         // heads
#define MA_KC 32                       // keys per chunk
#define MA_ROWB (MA_C * 2)             // bytes per staged row
#define MA_HALF (MA_KC * MA_ROWB)      // 32 KB: the chunk's mp rows; the mem rows follow
#define MA_STAGE (2 * MA_HALF)         // 64 KB per stage, two stages
#define MA_XOFF (2 * MA_STAGE)         // exchange area behind the stages
#define MA_LOG2E 1.4426950408889634f

#define ma_split_hl(X, HI, LO) do { const float x_ = (X); const bf16_t h_ = (bf16_t)x_; (HI) = h_; (LO) = (bf16_t)(x_ - (float)h_); } while (0)

// same counter-based mask as xattn.hip: element (clip * heads + head, query, key) kept iff hash >= thresh (= p * 2^32)
__device__ __forceinline__ bool ma_keep(unsigned seed, unsigned bh, unsigned qq, unsigned key, unsigned thresh) {
    unsigned h = seed ^ (bh * 0x9E3779B9u) ^ (key * 0x85EBCA6Bu) ^ (qq * 0xC2B2AE35u);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h >= thresh;
}

__device__ __forceinline__ unsigned ma_lds_u32(const char* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }

// LDS reads are ordinary (compiler-visible) loads: the compiler places its own lgkmcnt waits and interleaves reads and MFMAs.  Only the
// LDS-DMA is inline asm -- hipcc turns every wait into vmcnt(0) once it has seen the global_load_lds BUILTIN (attn_space.hip), but it does
// not look inside asm; the DMA's completion is ordered by this file's explicit vmcnt(0) + s_barrier (memory clobbers).
typedef __attribute__((address_space(3))) const u32x4* ma_lds_v4;
typedef __attribute__((address_space(3))) const f32x4* ma_lds_f4;
typedef __attribute__((address_space(3))) ma_s16x4* ma_lds_s4;
template <int OFF>
__device__ __forceinline__ bf16x8 ma_rd128(unsigned a) { return __builtin_bit_cast(bf16x8, *(ma_lds_v4)(size_t)(a + OFF)); }
template <int OFF>
__device__ __forceinline__ f32x4 ma_rd128f(unsigned a) { return *(ma_lds_f4)(size_t)(a + OFF); }
template <int OFF>
__device__ __forceinline__ void ma_wr128f(unsigned a, f32x4 v) { *(__attribute__((address_space(3))) f32x4*)(size_t)(a + OFF) = v; }
template <int OFF>
__device__ __forceinline__ bf16x4 ma_tr4(unsigned a) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ma_lds_s4)(size_t)(a + OFF)));
}
#define MA_BARRIER() asm volatile("s_barrier" ::: "memory")
#define MA_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MA_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// one 1-KB row of the stage: lane l fetches the 16 bytes at src + voff (voff = 16 (l ^ key(row)): the swizzle) to LDS dst + 16 l
__device__ __forceinline__ void ma_dma_row(const bf16_t* src, unsigned voff, unsigned dst) {
    // (s_nop 4: the scalar operands may come straight out of a v_readfirstlane -- VALU writes SGPR -> VMEM reads it needs 5 wait states, and
    // the compiler's hazard recogniser does not look inside inline asm; scripts/check_isa_hazards.py)
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(src), "s"(dst) : "memory", "m0");
}

// reductions over the four 16-lane groups of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48 hold the same query): two register swaps on the
// vector unit (v_permlane16_swap / v_permlane32_swap) instead of two ds_bpermute round trips through the LDS crossbar
__device__ __forceinline__ float ma_max_groups(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float ma_sum_groups(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

struct MaCommon {
    const bf16_t* mp; const bf16_t* mem; int64_t ld;     // [B, M, ld] bf16 (ld >= 512)
    int B, Q, M, slices, keys_per_slice;
    int Mv;                                              // keys [Mv, M) are padding (rows of zeros): masked out of every softmax (round 6: any memory length)
    unsigned drop_thresh; float drop_scale; unsigned seed;
};

// lane-constant LDS byte offsets (inside a stage half): row reads rb[b] for k-step ks = 4 a + b (a adds 256 B), transposed reads
// tb[dl] for d-tile dt = 8 a + dl (a adds 256 B); key tile 1 adds 16 rows = 16 KB in both
struct MaAddr { unsigned rb[4], tb[8]; };       // (they carry the stage: ma_addr_flip after every chunk)

__device__ __forceinline__ MaAddr ma_addr(int lane, int dh, unsigned lds0) {
    MaAddr A;
    const int kq = lane & 15, g = lane >> 4;
    const int f = 2 * (kq & 7);
#pragma unroll
    for (int b = 0; b < 4; ++b) A.rb[b] = lds0 + (unsigned)(kq * MA_ROWB + 16 * (32 * dh + ((4 * b + g) ^ f)));
    const int i = lane & 15;
    const int kr = 4 * g + (i >> 2), f2 = kr & 7, b1 = (i >> 1) & 1;
#pragma unroll
    for (int dl = 0; dl < 8; ++dl) A.tb[dl] = lds0 + (unsigned)(kr * MA_ROWB + 16 * (32 * dh + 2 * (dl ^ f2) + b1) + 8 * (i & 1));
    return A;
}

__device__ __forceinline__ void ma_addr_flip(MaAddr& A) {
#pragma unroll
    for (int b = 0; b < 4; ++b) A.rb[b] ^= MA_STAGE;
#pragma unroll
    for (int dl = 0; dl < 8; ++dl) A.tb[dl] ^= MA_STAGE;
}

// wave w stages rows 4 w .. 4 w + 3 of the chunk's mp rows and of its mem rows (8 DMAs of 1 KB)
__device__ __forceinline__ void ma_stage(const MaCommon& p, int b, int k0, int wave, unsigned lane16, unsigned stage_lds) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kr = 4 * wave + i;
        const int64_t row = ((int64_t)b * p.M + k0 + kr) * p.ld;
        const unsigned voff = lane16 ^ (unsigned)(32 * (kr & 7));              // 16 (lane ^ 2 (kr & 7)): the swizzle key is wave-uniform
        ma_dma_row(p.mp + row, voff, stage_lds + kr * MA_ROWB);
        ma_dma_row(p.mem + row, voff, stage_lds + MA_HALF + kr * MA_ROWB);
    }
}

// fp32 rows -> MFMA B operand fragments (lane (q = lane & 15, g): dims 256 dh + 32 ks + 8 g .. + 7) as bf16 hi + lo
__device__ __forceinline__ void ma_load_q(const float* row, bool valid, int dh, int g, bf16x8 (&hi)[8], bf16x8 (&lo)[8]) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (valid) {
            const float* s = row + 256 * dh + 32 * ks + 8 * g;
            const f32x4 a = *(const f32x4*)s, b = *(const f32x4*)(s + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ma_split_hl(a[j], hi[ks][j], lo[ks][j]); ma_split_hl(b[j], hi[ks][4 + j], lo[ks][4 + j]); }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { hi[ks][j] = (bf16_t)0.f; lo[ks][j] = (bf16_t)0.f; }
        }
    }
}

// S^T partial of one 32-key chunk over this wave's 256 dims: acc[kt] (lane (q, keys 16 kt + 4 g + r)) += rows[HALF_OFF] . (hi + lo)^T
template <int HALF_OFF>
__device__ __forceinline__ void ma_scores(const MaAddr& A, const bf16x8 (&hi)[8], const bf16x8 (&lo)[8], f32x4 (&acc)[2]) {
    const unsigned a0 = A.rb[0], a1 = A.rb[1], a2 = A.rb[2], a3 = A.rb[3];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        // k-steps 0..3 then 4..7 of this key tile: eight reads in flight, consumed four at a time
        bf16x8 f0 = kt == 0 ? ma_rd128<HALF_OFF>(a0) : ma_rd128<HALF_OFF + 16384>(a0);
        bf16x8 f1 = kt == 0 ? ma_rd128<HALF_OFF>(a1) : ma_rd128<HALF_OFF + 16384>(a1);
        bf16x8 f2 = kt == 0 ? ma_rd128<HALF_OFF>(a2) : ma_rd128<HALF_OFF + 16384>(a2);
        bf16x8 f3 = kt == 0 ? ma_rd128<HALF_OFF>(a3) : ma_rd128<HALF_OFF + 16384>(a3);
        bf16x8 f4 = kt == 0 ? ma_rd128<HALF_OFF + 256>(a0) : ma_rd128<HALF_OFF + 16384 + 256>(a0);
        bf16x8 f5 = kt == 0 ? ma_rd128<HALF_OFF + 256>(a1) : ma_rd128<HALF_OFF + 16384 + 256>(a1);
        bf16x8 f6 = kt == 0 ? ma_rd128<HALF_OFF + 256>(a2) : ma_rd128<HALF_OFF + 16384 + 256>(a2);
        bf16x8 f7 = kt == 0 ? ma_rd128<HALF_OFF + 256>(a3) : ma_rd128<HALF_OFF + 16384 + 256>(a3);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0, hi[0], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0, lo[0], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1, hi[1], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1, lo[1], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2, hi[2], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2, lo[2], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f3, hi[3], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f3, lo[3], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f4, hi[4], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f4, lo[4], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f5, hi[5], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f5, lo[5], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f6, hi[6], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f6, lo[6], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f7, hi[7], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f7, lo[7], acc[kt], 0, 0, 0);
    }
}

// X^T[my 256 dims, q] += rows[HALF_OFF]^T[dims, 32 keys] . (ph + pl)[keys, q]: 16 d-tiles, the rows fetched by transposing reads in the
// key order of the S^T accumulator layout (lane (q, g): keys 4 g .. 4 g + 3 of key tile 0, then of key tile 1)
template <int HALF_OFF>
__device__ __forceinline__ void ma_pool(const MaAddr& A, const bf16x8& ph, const bf16x8& pl, f32x4 (&o)[16]) {
    const unsigned (&t)[8] = A.tb;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int quad = 0; quad < 2; ++quad) {
            // four d-tiles: eight transposing reads, consumed as they land
            bf16x4 a0, b0, a1, b1, a2, b2, a3, b3;
            if (half == 0) {
                a0 = ma_tr4<HALF_OFF>(t[4 * quad + 0]); b0 = ma_tr4<HALF_OFF + 16384>(t[4 * quad + 0]);
                a1 = ma_tr4<HALF_OFF>(t[4 * quad + 1]); b1 = ma_tr4<HALF_OFF + 16384>(t[4 * quad + 1]);
                a2 = ma_tr4<HALF_OFF>(t[4 * quad + 2]); b2 = ma_tr4<HALF_OFF + 16384>(t[4 * quad + 2]);
                a3 = ma_tr4<HALF_OFF>(t[4 * quad + 3]); b3 = ma_tr4<HALF_OFF + 16384>(t[4 * quad + 3]);
            } else {
                a0 = ma_tr4<HALF_OFF + 256>(t[4 * quad + 0]); b0 = ma_tr4<HALF_OFF + 16384 + 256>(t[4 * quad + 0]);
                a1 = ma_tr4<HALF_OFF + 256>(t[4 * quad + 1]); b1 = ma_tr4<HALF_OFF + 16384 + 256>(t[4 * quad + 1]);
                a2 = ma_tr4<HALF_OFF + 256>(t[4 * quad + 2]); b2 = ma_tr4<HALF_OFF + 16384 + 256>(t[4 * quad + 2]);
                a3 = ma_tr4<HALF_OFF + 256>(t[4 * quad + 3]); b3 = ma_tr4<HALF_OFF + 16384 + 256>(t[4 * quad + 3]);
            }
            const int dt = 8 * half + 4 * quad;
            const bf16x8 v0 = {a0[0], a0[1], a0[2], a0[3], b0[0], b0[1], b0[2], b0[3]};
            const bf16x8 v1 = {a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3]};
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, ph, o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, pl, o[dt], 0, 0, 0);
            o[dt + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, ph, o[dt + 1], 0, 0, 0);
            o[dt + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, pl, o[dt + 1], 0, 0, 0);
            const bf16x8 v2 = {a2[0], a2[1], a2[2], a2[3], b2[0], b2[1], b2[2], b2[3]};
            const bf16x8 v3 = {a3[0], a3[1], a3[2], a3[3], b3[0], b3[1], b3[2], b3[3]};
            o[dt + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v2, ph, o[dt + 2], 0, 0, 0);
            o[dt + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v2, pl, o[dt + 2], 0, 0, 0);
            o[dt + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v3, ph, o[dt + 3], 0, 0, 0);
            o[dt + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v3, pl, o[dt + 3], 0, 0, 0);
        }
    }
}

// the same in four parts (row 4 w + I of both halves: 2 DMAs each), so that the issue of a stage's 8 DMAs -- 600 cycles per wave when they go
// out back to back, 1600 for the waves that lose the arbitration (scripts/mattn_timeline.hip) -- is spread over the phases of a chunk
template <int I>
__device__ __forceinline__ void ma_stage_part(const MaCommon& p, int b, int k0, int wave, unsigned lane16, unsigned stage_lds) {
    const int kr = 4 * wave + I;
    const int64_t row = ((int64_t)b * p.M + k0 + kr) * p.ld;
    const unsigned voff = lane16 ^ (unsigned)(32 * (kr & 7));
    ma_dma_row(p.mp + row, voff, stage_lds + kr * MA_ROWB);
    ma_dma_row(p.mem + row, voff, stage_lds + MA_HALF + kr * MA_ROWB);
}

// workgroup index -> (clip, head group, key slice); the two head groups of a (clip, slice) unit are blockIdx b and b + 8: the same
// XCD under round-robin placement (speed only), so the second one finds the unit's rows in that XCD's L2
__device__ __forceinline__ bool ma_unit(const MaCommon& p, int& b, int& hg, int& slice) {
    const int idx = blockIdx.x, xcd = idx & 7, j = idx >> 3;
    hg = j & 1;
    const int u = (j >> 1) * 8 + xcd;
    if (u >= p.B * p.slices) return false;
    b = u / p.slices;
    slice = u % p.slices;
    return true;
}

// Slice hand-off without a merge launch: every workgroup of a (clip, head group) stores its partial rows, then takes a ticket; the one whose
// ticket is the last folds all slices.  Producer / consumer forms of MI355X_MICROARCH.md (inter-workgroup visibility): every storing wave
// waits vmcnt(0), workgroup barrier, ONE lane releases at agent scope + waits + takes the ticket (agent-scope atomic); the last arriver's
// lane acquires at agent scope + waits, a barrier publishes that to its other waves, plain loads follow.  The counter is reset by the last
// arriver (kernels of one stream never overlap; one table row per launch stream: runtime.cpp hh_stream_slot).
#define MA_TICKET_UNITS 4096
__device__ unsigned g_ma_ticket[32][MA_TICKET_UNITS];

__device__ __forceinline__ bool ma_last_arriver(unsigned* ticket, unsigned total, int* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev + 1u == total;
        if (last) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *lds_flag = last;
    }
    __syncthreads();
    return *lds_flag != 0;
}

// scripts/mattn_timeline.hip compiles this file with -DMA_TIMELINE: workgroup 0's waves then add up the shader cycles of the phases of a
// chunk (the product build contains none of this)
#ifdef MA_TIMELINE
__device__ unsigned long long g_ma_tl[8][8];
#define MA_TL_DECL unsigned long long tl_t = __builtin_readcyclecounter(); unsigned long long tl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define MA_TL(I) do { const unsigned long long n_ = __builtin_readcyclecounter(); tl_acc[I] += n_ - tl_t; tl_t = n_; } while (0)
#define MA_TL_FLUSH() do { if (blockIdx.x == 0 && lane == 0) for (int i_ = 0; i_ < 8; ++i_) g_ma_tl[wave][i_] = tl_acc[i_]; } while (0)
#else
#define MA_TL_DECL
#define MA_TL(I)
#define MA_TL_FLUSH()
#endif

struct MaFwd {
    MaCommon c;
    const float* qt;          // [B*Q, H*C] fp32: row (clip, query), column head * 512 + k
    float* o_part;            // [slices][B*Q, H*C]: un-normalised partial pooled rows (slices == 1: unused)
    float* st_part;           // [slices][B*Q, H][4]: running maximum (base-2 logits), sum of p, sum of dropped p, -
    float* pooled;            // [B*Q, H*C] final, normalised rows
    float* lse2; float* rsum; // [B*Q, H] final statistics
    int slot;                 // ticket table row of the launch stream
};

__global__ __launch_bounds__(512) void mattn_fwd_kernel(MaFwd p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int b, hg, slice;
    if (!ma_unit(p.c, b, hg, slice)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave >> 1, dh = wave & 1, head = 4 * hg + rt;
    const int ql = lane & 15, g = lane >> 4;
    const int Q = p.c.Q;
    const unsigned lds0 = ma_lds_u32(smem);
    MaAddr A = ma_addr(lane, dh, lds0);
    const unsigned voff = 16u * (unsigned)lane;
    const int k_begin = slice * p.c.keys_per_slice;
    const int k_end = min(p.c.M, k_begin + p.c.keys_per_slice);
    const int nchunks = (k_end - k_begin) / MA_KC;
    if (nchunks > 0) ma_stage(p.c, b, k_begin, wave, voff, lds0);

    bf16x8 qh[8], qlo[8];
    const int64_t qrow = (int64_t)b * Q + (ql < Q ? ql : 0);
    ma_load_q(p.qt + qrow * (MA_H * MA_C) + head * MA_C, ql < Q, dh, g, qh, qlo);
    f32x4 o[16];
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f, rs_run = 0.f;
    const unsigned xme = lds0 + MA_XOFF + wave * 2048 + lane * 16, xpartner = lds0 + MA_XOFF + (wave ^ 1) * 2048 + lane * 16;
    const unsigned bh = (unsigned)(b * MA_H + head);

    MA_TL_DECL;
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);        // the second-dispatched half loses every issue arbitration otherwise (MI355X_MICROARCH.md, two waves per SIMD)
    for (int c = 0; c < nchunks; ++c) {
        MA_WAIT_VM0();                                   // this wave's rows of chunk c have landed
        MA_TL(0);
        MA_BARRIER();                                    // everyone's have; everyone has left chunk c - 1 (its stage is free)
        MA_TL(1);
        const bool more = c + 1 < nchunks;                 // (wave-uniform)
        const int kn = k_begin + (c + 1) * MA_KC;
        const unsigned sn = lds0 + ((c + 1) & 1) * MA_STAGE;
        if (more) ma_stage_part<0>(p.c, b, kn, wave, voff, sn);
        MA_TL(2);
        f32x4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        ma_scores<0>(A, qh, qlo, s);
        ma_wr128f<0>(xme, s[0]);
        ma_wr128f<1024>(xme, s[1]);
        if (more) ma_stage_part<1>(p.c, b, kn, wave, voff, sn);
        MA_WAIT_LGKM0();
        MA_TL(3);
        MA_BARRIER();
        MA_TL(4);
        f32x4 x0 = ma_rd128f<0>(xpartner), x1 = ma_rd128f<1024>(xpartner);
        s[0] += x0;
        s[1] += x1;
        // online softmax in base 2 (this lane: query ql, keys 4 g + r of both key tiles)
        float t[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[r] = s[0][r] * MA_LOG2E; t[4 + r] = s[1][r] * MA_LOG2E; }
        const bool ragged = k_begin + (c + 1) * MA_KC > p.c.Mv;          // (wave-uniform: the chunk holds padding keys)
        if (ragged) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k_begin + c * MA_KC + 4 * g + 16 * (j >> 2) + (j & 3) >= p.c.Mv) t[j] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])), fmaxf(fmaxf(t[4], t[5]), fmaxf(t[6], t[7])));
        mx = ma_max_groups(mx);
        if (more) ma_stage_part<2>(p.c, b, kn, wave, voff, sn);
        const float m_new = fmaxf(m_run, mx);
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull) {              // rescale only when some row's maximum moved
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) o[dt] *= alpha;
            l_run *= alpha;
            rs_run *= alpha;
            m_run = m_new;
        }
        bf16x8 ph, pl;
        const int kbase = k_begin + c * MA_KC + 4 * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float pv = __builtin_amdgcn_exp2f(t[j] - m_run);
            if (ragged && t[j] == -INFINITY) pv = 0.f;                      // (a slice of padding only: m_run is still -inf, exp2(-inf + inf) is NaN)
            l_run += pv;
            if (p.c.drop_thresh) pv = ma_keep(p.c.seed, bh, (unsigned)ql, (unsigned)(kbase + 16 * (j >> 2) + (j & 3)), p.c.drop_thresh) ? pv * p.c.drop_scale : 0.f;
            rs_run += pv;
            ma_split_hl(pv, ph[j], pl[j]);
        }
        if (more) ma_stage_part<3>(p.c, b, kn, wave, voff, sn);
        MA_TL(5);
        ma_pool<MA_HALF>(A, ph, pl, o);
        ma_addr_flip(A);
        MA_TL(6);
    }
    MA_TL_FLUSH();
    l_run = ma_sum_groups(l_run);
    rs_run = ma_sum_groups(rs_run);
    const bool live = ql < Q;
    const int64_t r = (int64_t)b * Q + (live ? ql : 0);
    const bool final_ = p.c.slices == 1;
    const float sc = final_ ? 1.f / l_run : 1.f;
    if (live) {
        float* orow = (final_ ? p.pooled : p.o_part + (int64_t)slice * p.c.B * Q * (MA_H * MA_C)) + r * (MA_H * MA_C) + head * MA_C + 256 * dh + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) *(f32x4*)(orow + 16 * dt) = o[dt] * sc;
        if (dh == 0 && g == 0) {
            if (final_) {
                p.lse2[r * MA_H + head] = m_run + __builtin_amdgcn_logf(l_run);      // v_log_f32 is base 2
                p.rsum[r * MA_H + head] = rs_run * sc;
            } else {
                *(f32x4*)(p.st_part + (((int64_t)slice * p.c.B * Q + r) * MA_H + head) * 4) = (f32x4){m_run, l_run, rs_run, 0.f};
            }
        }
    }
    if (final_) return;
    // ---- the last of this (clip, head group)'s `slices` workgroups folds them:  pooled = sum_s 2^(m_s - m) O_s / L,  L = sum_s 2^(m_s - m) l_s
    int* flag = (int*)(smem + MA_XOFF);
    if (!ma_last_arriver(&g_ma_ticket[p.slot][b * 2 + hg], (unsigned)p.c.slices, flag)) return;
    // (all 512 threads work on independent loads: first the slices' statistics -> per-(query, head) weights in LDS, then every thread folds
    // 13 float4 of the 52 x 512 output values over the slices; a serial loop over (query, head) x slice cost ~100 us of load latency)
    const int64_t plane = (int64_t)p.c.B * Q * MA_H;                // (row, head) pairs per slice
    const int S = p.c.slices, NP = Q * 4;
    float* wtab = (float*)smem;                                      // [NP <= 64][S <= 64] weights 2^(m_s - m) / L   (the stages are dead by now)
    float* stat = wtab + 64 * 64;                                    // [NP][S][4] raw statistics (64 KB at most)
    for (int e = tid; e < NP * S; e += 512) {
        const int pr = e / S, s_ = e % S;
        const int64_t rh = ((int64_t)b * Q + pr / 4) * MA_H + 4 * hg + (pr & 3);
        *(f32x4*)(stat + 4 * e) = *(const f32x4*)(p.st_part + (s_ * plane + rh) * 4);
    }
    __syncthreads();
    if (tid < NP) {
        float m = -INFINITY;
        for (int s_ = 0; s_ < S; ++s_) m = fmaxf(m, stat[4 * (tid * S + s_)]);
        float L = 0.f, RS = 0.f;
        for (int s_ = 0; s_ < S; ++s_) {
            const float w = __builtin_amdgcn_exp2f(stat[4 * (tid * S + s_)] - m);
            L += w * stat[4 * (tid * S + s_) + 1];
            RS += w * stat[4 * (tid * S + s_) + 2];
            wtab[tid * S + s_] = w;
        }
        const float inv = 1.f / L;
        for (int s_ = 0; s_ < S; ++s_) wtab[tid * S + s_] *= inv;
        const int64_t rh = ((int64_t)b * Q + tid / 4) * MA_H + 4 * hg + (tid & 3);
        p.lse2[rh] = m + __builtin_amdgcn_logf(L);
        p.rsum[rh] = RS * inv;
    }
    __syncthreads();
    for (int v = tid; v < NP * 128; v += 512) {                     // float4 v of the [NP][512] block
        const int pr = v >> 7;
        const int64_t at = (((int64_t)b * Q + pr / 4) * MA_H + 4 * hg + (pr & 3)) * MA_C + 4 * (v & 127);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int s_ = 0;
        for (; s_ + 4 <= S; s_ += 4) {
            const f32x4 a0 = *(const f32x4*)(p.o_part + (s_ + 0) * plane * MA_C + at), a1 = *(const f32x4*)(p.o_part + (s_ + 1) * plane * MA_C + at);
            const f32x4 a2 = *(const f32x4*)(p.o_part + (s_ + 2) * plane * MA_C + at), a3 = *(const f32x4*)(p.o_part + (s_ + 3) * plane * MA_C + at);
            acc += a0 * wtab[pr * S + s_]; acc += a1 * wtab[pr * S + s_ + 1]; acc += a2 * wtab[pr * S + s_ + 2]; acc += a3 * wtab[pr * S + s_ + 3];
        }
        for (; s_ < S; ++s_) acc += *(const f32x4*)(p.o_part + s_ * plane * MA_C + at) * wtab[pr * S + s_];
        *(f32x4*)(p.pooled + at) = acc;
    }
}

static void ma_drop_params(float p, unsigned* thresh, float* scale) {
    if (p <= 0.f) { *thresh = 0u; *scale = 1.f; return; }
    double t = (double)p * 4294967296.0;
    *thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    if (*thresh == 0u) *thresh = 1u;
    *scale = 1.f / (1.f - p);
}

static int ma_check(const char* what, int B, int Q, int M, int heads, int C, int64_t ld, int slices) {
    HH_REQUIRE(heads == MA_H && C == MA_C, HH_ERR_UNSUPPORTED, "%s: built for the reference decoder's d_model 512 / 8 heads (tfm_decoder.py:51) (heads=%d C=%d)", what, heads, C);
    HH_REQUIRE(B >= 0 && Q > 0 && Q <= 16 && M > 0 && M % MA_KC == 0, HH_ERR_SHAPE, "%s: need 0 < Q <= 16 and M %% 32 == 0 (Q=%d M=%d)", what, Q, M);
    HH_REQUIRE(ld >= MA_C && ld % 8 == 0, HH_ERR_SHAPE, "%s: memory leading dimension %lld too small / unaligned", what, (long long)ld);
    HH_REQUIRE(slices >= 1 && slices <= 64, HH_ERR_SHAPE, "%s: slices must be in [1, 64] (the fold keeps their statistics in LDS)", what);
    return HH_OK;
}

// keys per slice: whole chunks, no empty slice
static void ma_slicing(int M, int* slices, int* per) {
    int chunks = M / MA_KC;
    int s = *slices < chunks ? *slices : chunks;
    int cps = (chunks + s - 1) / s;
    *per = cps * MA_KC;
    *slices = (chunks + cps - 1) / cps;
}

int hh_tuning_mattn_no_ticket();          // gemm256.hip (the tuning table)

extern "C" int hh_mattn_slices(int M, int slices) {
    if (M <= 0 || M % MA_KC != 0 || slices < 1) return -1;
    int per;
    ma_slicing(M, &slices, &per);
    return slices;
}

extern "C" int64_t hh_workspace_bytes_mattn_fwd(int B, int Q, int slices) {
    if (B < 0 || Q <= 0 || Q > 16 || slices < 1) return -1;
    if (slices <= 1) return 16;
    return (int64_t)slices * B * Q * MA_H * (MA_C + 4) * 4;
}

extern "C" int hh_mattn_fwd(const float* qt, const void* mp, const void* mem, int64_t ld, float* pooled, float* lse2, float* rsum, float* workspace,
                            int slices, int B, int Q, int M, int heads, int C, float dropout_p, uint32_t seed, int keys_valid, hh_stream_t stream) {
    int rc = ma_check("hh_mattn_fwd", B, Q, M, heads, C, ld, slices);
    if (rc) return rc;
    HH_REQUIRE(keys_valid <= M, HH_ERR_SHAPE, "hh_mattn_fwd: keys_valid = %d > M = %d", keys_valid, M);
    HH_REQUIRE(HH_ALIGNED16(qt) && HH_ALIGNED16(mp) && HH_ALIGNED16(mem) && HH_ALIGNED16(pooled) && HH_ALIGNED16(workspace), HH_ERR_ALIGN,
               "hh_mattn_fwd: pointers must be 16-byte aligned");
    HH_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, HH_ERR_SHAPE, "hh_mattn_fwd: dropout_p must be in [0,1)");
    if (B == 0) return HH_OK;
    MaFwd p;
    p.c.mp = (const bf16_t*)mp; p.c.mem = (const bf16_t*)mem; p.c.ld = ld; p.c.B = B; p.c.Q = Q; p.c.M = M;
    p.c.Mv = keys_valid > 0 ? keys_valid : M;
    p.c.slices = slices;
    ma_slicing(M, &p.c.slices, &p.c.keys_per_slice);
    ma_drop_params(dropout_p, &p.c.drop_thresh, &p.c.drop_scale);
    p.c.seed = seed;
    HH_REQUIRE(p.c.slices == 1 || workspace != nullptr, HH_ERR_SHAPE, "hh_mattn_fwd: slices > 1 needs hh_workspace_bytes_mattn_fwd() bytes of workspace");
    p.qt = qt; p.pooled = pooled; p.lse2 = lse2; p.rsum = rsum;
    p.slot = hh_tuning_mattn_no_ticket() ? -1 : hh_stream_slot((hipStream_t)stream);
    if (p.c.slices > 1 && (p.slot < 0 || 2 * B > MA_TICKET_UNITS)) {
        // no ticket row for the last-arriver fold (more than 32 launch streams seen in this process, or B too large): one slice per (clip, head
        // group) -- slower at small B, same results up to the fp32 re-association of the fold; the workspace stays unused
        p.c.slices = 1;
        ma_slicing(M, &p.c.slices, &p.c.keys_per_slice);
    }
    const int64_t rows_h = (int64_t)B * Q * MA_H;
    p.o_part = workspace;
    p.st_part = p.c.slices == 1 ? nullptr : workspace + (int64_t)p.c.slices * rows_h * MA_C;
    static std::atomic<uint64_t> attr_mask{0};
    if (hh_attr_needed(attr_mask)) {
        hipError_t e = hipFuncSetAttribute((const void*)mattn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MA_XOFF + 8 * 2048);
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_mattn_fwd: cannot reserve %d B of LDS: %s", (int)(MA_XOFF + 8 * 2048), hipGetErrorString(e));
        hh_attr_done(attr_mask);
    }
    const int units = B * p.c.slices;
    const unsigned grid = 16u * (unsigned)((units + 7) / 8);
    hipStream_t s = (hipStream_t)stream;
    {
        HHProfScope prof(HH_PROF_XATTN_FWD, 4.0 * (double)B * M * MA_C, s);                  // mp and mem rows, bf16, once
        hipLaunchKernelGGL(mattn_fwd_kernel, dim3(grid), dim3(512), MA_XOFF + 8 * 2048, s, p);
    }
    return hh_check_launch("hh_mattn_fwd");
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of one layer's cross-attention in memory space.  With Pd = mask o P / (1 - p) (P = softmax of the scores), pooled = Pd mem,
// rsum = Pd 1 and the head output O = pooled Wv^T + rsum bv:
//     d Pd[q, i] = dpooled[q] . mem[i] + cb[q]          (cb = dO . bv, the rsum term)
//     d P        = mask / (1 - p) o d Pd,      dS = P o (dP - delta),      delta[q] = sum_i P dP = dO[q] . O[q]
//     d qt[q]    = sum_i dS[q, i] mp[i]                 (this kernel, per key slice)
//     d mem     += Pd^T dpooled,   d mp += dS^T qt      (all layers at once: hh_gemm_tn_bf16_batched2 on the Pd^T / dS^T this kernel
//                                                        leaves in bf16, rows layer * 128 + head * 16 + query, keys contiguous)
// Same decomposition as the forward; per chunk and wave: S^T and dP^T partials (32 MFMAs each, exchanged together: 4 KB per wave),
// P / dS on the vector unit, d qt^T += mp^T . dS^T (32 MFMAs, dS as hi + lo).  The wave with dim-half 0 stores Pd^T, the other dS^T.
struct MaBwd {
    MaCommon c;
    const float* qt;          // [B*Q, H*C]
    const float* dpooled;     // [B*Q, H*C]
    const float* lse2;        // [B*Q, H]
    const float* dca;         // [B*Q, C'] gradient of the attention output (C' = H * 64 columns, head-major)
    const float* ca;          // [B*Q, C'] the attention output itself (delta = sum_n dca * ca over a head's 64 columns)
    const float* bv;          // [C'] value bias of the layer (cb = sum_n dca * bv)
    float* dqt_part;          // [slices][B*Q, H*C] (slices > 1)
    float* dqt;               // [B*Q, H*C] final
    int slot;
    bf16_t* pdT; bf16_t* dsT; // [B, rows_total, M] bf16; this layer's rows start at row_off
    bf16_t* qt16; bf16_t* dp16; // [B, rows_total, C] bf16 copies of qt / dpooled (rows as above; slice 0 writes them), or NULL
    int rows_total, row_off;
};

__global__ __launch_bounds__(512) void mattn_bwd_kernel(MaBwd p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int b, hg, slice;
    if (!ma_unit(p.c, b, hg, slice)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave >> 1, dh = wave & 1, head = 4 * hg + rt;
    const int ql = lane & 15, g = lane >> 4;
    const int Q = p.c.Q;
    const unsigned lds0 = ma_lds_u32(smem);
    MaAddr A = ma_addr(lane, dh, lds0);
    const unsigned voff = 16u * (unsigned)lane;
    const int k_begin = slice * p.c.keys_per_slice;
    const int k_end = min(p.c.M, k_begin + p.c.keys_per_slice);
    const int nchunks = (k_end - k_begin) / MA_KC;
    if (nchunks > 0) ma_stage(p.c, b, k_begin, wave, voff, lds0);

    const bool live = ql < Q;
    const int64_t r = (int64_t)b * Q + (live ? ql : 0);
    bf16x8 qh[8], qlo[8], dh_[8], dlo[8];
    ma_load_q(p.qt + r * (MA_H * MA_C) + head * MA_C, live, dh, g, qh, qlo);
    ma_load_q(p.dpooled + r * (MA_H * MA_C) + head * MA_C, live, dh, g, dh_, dlo);
    // row constants: lse2, delta = dO . O and cb = dO . bv over the head's 64 columns (16 per lane, summed over the 4 lane groups)
    float delta = 0.f, cb = 0.f;
    {
        const float* d = p.dca + r * (MA_H * 64) + head * 64 + 16 * g;
        const float* o_ = p.ca + r * (MA_H * 64) + head * 64 + 16 * g;
        const float* bvp = p.bv + head * 64 + 16 * g;
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            const f32x4 dv = *(const f32x4*)(d + e), ov = *(const f32x4*)(o_ + e);
            delta += dv[0] * ov[0] + dv[1] * ov[1] + dv[2] * ov[2] + dv[3] * ov[3];
            const f32x4 bb = *(const f32x4*)(bvp + e);
            cb += dv[0] * bb[0] + dv[1] * bb[1] + dv[2] * bb[2] + dv[3] * bb[3];
        }
        delta = ma_sum_groups(delta);
        cb = ma_sum_groups(cb);
    }
    const float lse = live ? p.lse2[r * MA_H + head] : 0.f;
    // bf16 copies of this head's qt / dpooled rows for the batched d-memory GEMM (hi halves = the rounded values; zero rows beyond Q)
    if (slice == 0 && p.qt16 != nullptr) {
        const int64_t row16 = ((int64_t)b * p.rows_total + p.row_off + head * 16 + ql) * MA_C + 256 * dh + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            *(bf16x8*)(p.qt16 + row16 + 32 * ks) = qh[ks];
            *(bf16x8*)(p.dp16 + row16 + 32 * ks) = dh_[ks];
        }
    }
    f32x4 acc[16];
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned xme = lds0 + MA_XOFF + wave * 4096 + lane * 16, xpartner = lds0 + MA_XOFF + (wave ^ 1) * 4096 + lane * 16;
    const unsigned bh = (unsigned)(b * MA_H + head);
    bf16_t* outT = (dh == 0 ? p.pdT : p.dsT) + ((int64_t)b * p.rows_total + p.row_off + head * 16 + ql) * p.c.M;

    if (wave >= 4) __builtin_amdgcn_s_setprio(1);        // (as in the forward)
    for (int c = 0; c < nchunks; ++c) {
        MA_WAIT_VM0();
        MA_BARRIER();
        const bool more = c + 1 < nchunks;                 // the next stage's 8 DMAs go out two at a time between the phases (see ma_stage_part)
        const int kn = k_begin + (c + 1) * MA_KC;
        const unsigned sn = lds0 + ((c + 1) & 1) * MA_STAGE;
        if (more) ma_stage_part<0>(p.c, b, kn, wave, voff, sn);
        f32x4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dp[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        ma_scores<0>(A, qh, qlo, s);
        ma_wr128f<0>(xme, s[0]);
        ma_wr128f<1024>(xme, s[1]);
        if (more) ma_stage_part<1>(p.c, b, kn, wave, voff, sn);
        ma_scores<MA_HALF>(A, dh_, dlo, dp);
        ma_wr128f<2048>(xme, dp[0]);
        ma_wr128f<3072>(xme, dp[1]);
        if (more) ma_stage_part<2>(p.c, b, kn, wave, voff, sn);
        MA_WAIT_LGKM0();
        MA_BARRIER();
        f32x4 x0 = ma_rd128f<0>(xpartner), x1 = ma_rd128f<1024>(xpartner), x2 = ma_rd128f<2048>(xpartner), x3 = ma_rd128f<3072>(xpartner);
        s[0] += x0; s[1] += x1; dp[0] += x2; dp[1] += x3;
        bf16x8 sh, sl;
        bf16x4 st0, st1;                                  // what this wave stores: Pd (dim-half 0) or dS (dim-half 1), keys 4 g .. of each key tile
        const int kbase = k_begin + c * MA_KC + 4 * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sv = j < 4 ? s[0][j] : s[1][j - 4], dv = j < 4 ? dp[0][j] : dp[1][j - 4];
            const float pv = (live && kbase + 16 * (j >> 2) + (j & 3) < p.c.Mv) ? __builtin_amdgcn_exp2f(sv * MA_LOG2E - lse) : 0.f;
            // (delta = dO . O contains the value-bias term rsum * cb, so cb belongs to dP with or without dropout)
            float pd = pv, dpm = dv + cb;
            if (p.c.drop_thresh) {
                const bool keep = ma_keep(p.c.seed, bh, (unsigned)ql, (unsigned)(kbase + 16 * (j >> 2) + (j & 3)), p.c.drop_thresh);
                pd = keep ? pv * p.c.drop_scale : 0.f;
                dpm = keep ? (dv + cb) * p.c.drop_scale : 0.f;
            }
            const float ds = pv * (dpm - delta);
            ma_split_hl(ds, sh[j], sl[j]);
            const bf16_t ov = (bf16_t)(dh == 0 ? pd : ds);
            if (j < 4) st0[j] = ov; else st1[j - 4] = ov;
        }
        *(bf16x4*)(outT + kbase) = st0;
        *(bf16x4*)(outT + kbase + 16) = st1;
        if (more) ma_stage_part<3>(p.c, b, kn, wave, voff, sn);
        ma_pool<0>(A, sh, sl, acc);
        ma_addr_flip(A);
    }
    const bool final_ = p.c.slices == 1;
    if (live) {
        float* orow = (final_ ? p.dqt : p.dqt_part + (int64_t)slice * p.c.B * Q * (MA_H * MA_C)) + r * (MA_H * MA_C) + head * MA_C + 256 * dh + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) *(f32x4*)(orow + 16 * dt) = acc[dt];
    }
    if (final_) return;
    // the last of this (clip, head group)'s workgroups adds the slices' planes in a fixed order (see ma_last_arriver)
    int* flag = (int*)(smem + MA_XOFF);
    if (!ma_last_arriver(&g_ma_ticket[p.slot][b * 2 + hg], (unsigned)p.c.slices, flag)) return;
    const int64_t plane = (int64_t)p.c.B * Q * (MA_H * MA_C);
    const int S = p.c.slices;
    for (int v = tid; v < Q * 4 * 128; v += 512) {                  // float4 v of this (clip, head group)'s [Q * 4][512] block: independent loads
        const int pr = v >> 7;
        const int64_t at = ((int64_t)b * Q + pr / 4) * (MA_H * MA_C) + (4 * hg + (pr & 3)) * MA_C + 4 * (v & 127);
        f32x4 a_ = {0.f, 0.f, 0.f, 0.f};
        int s_ = 0;
        for (; s_ + 4 <= S; s_ += 4) {
            const f32x4 a0 = *(const f32x4*)(p.dqt_part + (s_ + 0) * plane + at), a1 = *(const f32x4*)(p.dqt_part + (s_ + 1) * plane + at);
            const f32x4 a2 = *(const f32x4*)(p.dqt_part + (s_ + 2) * plane + at), a3 = *(const f32x4*)(p.dqt_part + (s_ + 3) * plane + at);
            a_ += a0; a_ += a1; a_ += a2; a_ += a3;
        }
        for (; s_ < S; ++s_) a_ += *(const f32x4*)(p.dqt_part + s_ * plane + at);
        *(f32x4*)(p.dqt + at) = a_;
    }
}

extern "C" int64_t hh_workspace_bytes_mattn_bwd(int B, int Q, int slices) {
    if (B < 0 || Q <= 0 || Q > 16 || slices < 1) return -1;
    if (slices == 1) return 16;
    return (int64_t)slices * B * Q * MA_H * MA_C * 4;
}

extern "C" int hh_mattn_bwd(const float* qt, const float* dpooled, const float* lse2, const float* dca, const float* ca, const float* bv,
                            const void* mp, const void* mem, int64_t ld, float* dqt, float* workspace, int slices, void* pdT, void* dsT, void* qt16, void* dp16,
                            int rows_total, int row_off, int B, int Q, int M, int heads, int C, float dropout_p, uint32_t seed, int keys_valid, hh_stream_t stream) {
    int rc = ma_check("hh_mattn_bwd", B, Q, M, heads, C, ld, slices);
    if (rc) return rc;
    HH_REQUIRE(keys_valid <= M, HH_ERR_SHAPE, "hh_mattn_bwd: keys_valid = %d > M = %d", keys_valid, M);
    HH_REQUIRE(HH_ALIGNED16(qt) && HH_ALIGNED16(dpooled) && HH_ALIGNED16(dca) && HH_ALIGNED16(ca) && HH_ALIGNED16(bv) && HH_ALIGNED16(mp) && HH_ALIGNED16(mem) &&
               HH_ALIGNED16(dqt) && HH_ALIGNED16(workspace) && HH_ALIGNED16(pdT) && HH_ALIGNED16(dsT) && HH_ALIGNED16(qt16) && HH_ALIGNED16(dp16), HH_ERR_ALIGN,
               "hh_mattn_bwd: pointers must be 16-byte aligned");
    HH_REQUIRE(pdT != nullptr && dsT != nullptr && (qt16 == nullptr) == (dp16 == nullptr), HH_ERR_SHAPE, "hh_mattn_bwd: pdT / dsT are required; qt16 and dp16 come together");
    HH_REQUIRE(rows_total >= row_off + MA_H * 16 && row_off >= 0 && row_off % 16 == 0, HH_ERR_SHAPE, "hh_mattn_bwd: the layer's 128 rows [row_off, row_off + 128) must lie inside rows_total");
    HH_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, HH_ERR_SHAPE, "hh_mattn_bwd: dropout_p must be in [0,1)");
    if (B == 0) return HH_OK;
    MaBwd p;
    p.c.mp = (const bf16_t*)mp; p.c.mem = (const bf16_t*)mem; p.c.ld = ld; p.c.B = B; p.c.Q = Q; p.c.M = M;
    p.c.Mv = keys_valid > 0 ? keys_valid : M;
    p.c.slices = slices;
    ma_slicing(M, &p.c.slices, &p.c.keys_per_slice);
    HH_REQUIRE(p.c.slices == 1 || workspace != nullptr, HH_ERR_SHAPE, "hh_mattn_bwd: slices > 1 needs hh_workspace_bytes_mattn_bwd() bytes of workspace");
    p.slot = hh_tuning_mattn_no_ticket() ? -1 : hh_stream_slot((hipStream_t)stream);
    if (p.c.slices > 1 && (p.slot < 0 || 2 * B > MA_TICKET_UNITS)) {
        // no ticket row for the last-arriver fold (more than 32 launch streams seen in this process, or B too large): one slice per (clip, head
        // group) -- slower at small B, same results up to the fp32 re-association of the fold; the workspace stays unused
        p.c.slices = 1;
        ma_slicing(M, &p.c.slices, &p.c.keys_per_slice);
    }
    ma_drop_params(dropout_p, &p.c.drop_thresh, &p.c.drop_scale);
    p.c.seed = seed;
    p.qt = qt; p.dpooled = dpooled; p.lse2 = lse2; p.dca = dca; p.ca = ca; p.bv = bv; p.dqt_part = workspace; p.dqt = dqt;
    p.pdT = (bf16_t*)pdT; p.dsT = (bf16_t*)dsT; p.qt16 = (bf16_t*)qt16; p.dp16 = (bf16_t*)dp16; p.rows_total = rows_total; p.row_off = row_off;
    static std::atomic<uint64_t> attr_mask{0};
    if (hh_attr_needed(attr_mask)) {
        hipError_t e = hipFuncSetAttribute((const void*)mattn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MA_XOFF + 8 * 4096);
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_mattn_bwd: cannot reserve %d B of LDS: %s", (int)(MA_XOFF + 8 * 4096), hipGetErrorString(e));
        hh_attr_done(attr_mask);
    }
    const int units = B * p.c.slices;
    const unsigned grid = 16u * (unsigned)((units + 7) / 8);
    hipStream_t s = (hipStream_t)stream;
    HHProfScope prof(HH_PROF_XATTN_BWD, 4.0 * (double)B * M * MA_C + 4.0 * (double)B * M * MA_H * 16 * 2, s);      // rows read once + Pd^T / dS^T written
    hipLaunchKernelGGL(mattn_bwd_kernel, dim3(grid), dim3(512), MA_XOFF + 8 * 4096, s, p);
    return hh_check_launch("hh_mattn_bwd");
}

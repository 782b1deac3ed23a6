// Decoder cross-attention core, forward and backward (nn.MultiheadAttention inside
// TransformerDecoderLayer.forward_pre, model/tfm_decoder.py:438-441): per (clip, head) Q<=16 queries attend
// M = T*n memory keys, head dim 64, no mask (key_padding_mask is all-False, :203), dropout handled by the caller.
// HBM-bound: forward reads K and V once (2*M*128 B per (clip, head)); backward reads K,V and writes dK,dV.
//
// One workgroup per (clip, head), 4 waves; wave w owns keys [32*(4*i+w), +32) for i = 0,1,...
//  forward:  S^T = K.Q^T (mfma 16x16x32, K fragments straight from HBM, Q^T in registers), online softmax per wave
//            (query = lane & 15, keys in the 4 accumulator registers x lane>>4), O^T += V^T.P^T with the S^T accumulator
//            re-used as B operand (k order 16*(j>>2) + 4g + (j&3)); V^T comes from a wave-private LDS transpose.
//            The four waves' (m, l, O) partials are merged through LDS; lse = m + ln(l) is saved for the backward.
//  backward: per 16-key tile, non-swapped S, dP give P and dS with key on the lane -> they are the B operands of
//            dV^T = dO^T.P and dK^T = Q^T.dS (mfma 16x16x16, contraction over the 16 query slots); swapped S^T, dP^T
//            give dS^T for dQ^T += K^T.dS^T (contraction over keys, K^T via the wave-private LDS transpose).
// Round 2: every operand that comes from the 13-row query side (q, dO, the probabilities P and the score gradients dS) enters the
// MFMAs as a bf16 hi + lo pair (x = hi + lo + O(2^-17 x)), so the products are exact with respect to the bf16 K / V the memory side
// stores: 2 MFMAs where one side is K or V, 3 (hi.hi + lo.hi + hi.lo) where both sides are query-side values.  The kernels stay
// HBM-bound (K, V streamed once); the point is parity -- with K/V shared, decoder gradients now agree with the fp32 oracle at the
// level of its own ReLU-kink noise instead of one bf16 rounding per operand.
#include "common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

#define VSTRIDE 36

// hi = bf16(x), lo = bf16(x - hi); a macro because vector elements cannot bind to references
#define split_hl(X, HI, LO) do { const float x_ = (X); const bf16_t h_ = (bf16_t)x_; (HI) = h_; (LO) = (bf16_t)(x_ - (float)h_); } while (0)

// counter-based dropout mask for element (clip*heads+head, query, key): keep iff hash >= thresh (thresh = p * 2^32)
__device__ __forceinline__ bool drop_keep(unsigned seed, unsigned bh, unsigned qq, unsigned key, unsigned thresh) {
    unsigned h = seed ^ (bh * 0x9E3779B9u) ^ (key * 0x85EBCA6Bu) ^ (qq * 0xC2B2AE35u);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h >= thresh;
}   // bf16 elements per d-row of the wave-private transposed tile (32 keys + 4 pad)

__device__ __forceinline__ u32x2 pick4(unsigned a, unsigned b, unsigned c, unsigned d, bool hi) {
    u32x2 r;
    if (!hi) { r[0] = (a & 0xffffu) | (b << 16); r[1] = (c & 0xffffu) | (d << 16); }
    else { r[0] = (a >> 16) | (b & 0xffff0000u); r[1] = (c >> 16) | (d & 0xffff0000u); }
    return r;
}

// transpose 32 rows x 64 d (bf16, row stride ld elements) into tile[64][VSTRIDE]; one 4-row x 8-d block per lane
__device__ __forceinline__ void stage_transposed(const bf16_t* src, int64_t ld, bf16_t* tile, int lane) {
    const int kg = lane >> 3, c = lane & 7;
    const bf16_t* s = src + (int64_t)(kg * 4) * ld + c * 8;
    u32x4 r0 = *(const u32x4*)(s), r1 = *(const u32x4*)(s + ld), r2 = *(const u32x4*)(s + 2 * ld), r3 = *(const u32x4*)(s + 3 * ld);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        *(u32x2*)(tile + (c * 8 + 2 * w) * VSTRIDE + kg * 4) = pick4(r0[w], r1[w], r2[w], r3[w], false);
        *(u32x2*)(tile + (c * 8 + 2 * w + 1) * VSTRIDE + kg * 4) = pick4(r0[w], r1[w], r2[w], r3[w], true);
    }
}

// A-operand fragment (16x16x32) of the transposed tile for d-tile dt: lane (d = 16dt + (lane&15), g) holds keys
// {4g..4g+3} and {16+4g..16+4g+3}
__device__ __forceinline__ bf16x8 tile_frag(const bf16_t* tile, int dt, int lane) {
    const bf16_t* p = tile + (dt * 16 + (lane & 15)) * VSTRIDE + 4 * (lane >> 4);
    bf16x4 a = *(const bf16x4*)p, b = *(const bf16x4*)(p + 16);
    bf16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return r;
}

__device__ __forceinline__ void load_row_frag_f32_hl(const float* row, bool valid, bf16x8& hi, bf16x8& lo) {
    if (valid) {
        const f32x4 a = *(const f32x4*)row, b = *(const f32x4*)(row + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { split_hl(a[j], hi[j], lo[j]); split_hl(b[j], hi[4 + j], lo[4 + j]); }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (bf16_t)0.f; lo[j] = (bf16_t)0.f; }
    }
}

__device__ __forceinline__ bf16x8 load_row_frag_f32(const float* row, bool valid) {
    bf16x8 r;
    if (valid) {
        f32x4 a = *(const f32x4*)row, b = *(const f32x4*)(row + 4);
        r[0] = (bf16_t)a[0]; r[1] = (bf16_t)a[1]; r[2] = (bf16_t)a[2]; r[3] = (bf16_t)a[3];
        r[4] = (bf16_t)b[0]; r[5] = (bf16_t)b[1]; r[6] = (bf16_t)b[2]; r[7] = (bf16_t)b[3];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (bf16_t)0.f;
    }
    return r;
}

__global__ __launch_bounds__(256) void xattn_fwd_kernel(const float* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, int64_t ldkv, float* __restrict__ out,
                                                        float* __restrict__ lse, int B, int Q, int M, int heads, unsigned drop_thresh,
                                                        float drop_scale, unsigned seed, int chunk) {
    // gridDim.y > 1: blockIdx.y owns the key slice [y*chunk, (y+1)*chunk) and writes a slice-normalised (out, lse) pair into the
    // y-th plane of out/lse (the caller passes the workspace there); xattn_fwd_merge_kernel folds the planes
    __shared__ __attribute__((aligned(16))) bf16_t tiles[4][64 * VSTRIDE];
    __shared__ float ml[2][4][16];
    __shared__ __attribute__((aligned(16))) float obuf[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 15, g = lane >> 4;
    const int head = blockIdx.x % heads, b = blockIdx.x / heads;
    const int C = heads * 64;
    const float LOG2E = 1.4426950408889634f;
    bf16x8 qf[2], qfl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        load_row_frag_f32_hl(q + ((int64_t)b * Q + (ql < Q ? ql : 0)) * C + head * 64 + 32 * ks + 8 * g, ql < Q, qf[ks], qfl[ks]);
    const bf16_t* kb = k + (int64_t)b * M * ldkv + head * 64;
    const bf16_t* vb = v + (int64_t)b * M * ldkv + head * 64;
    bf16_t* tile = tiles[wave];
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int k_begin = blockIdx.y * chunk, k_end = min(M, k_begin + chunk);
    out += (int64_t)blockIdx.y * B * Q * C;
    lse += (int64_t)blockIdx.y * B * heads * Q;
    for (int k0 = k_begin + wave * 32; k0 < k_end; k0 += 128) {
        f32x4 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bf16_t* kr = kb + (int64_t)(k0 + 16 * t + ql) * ldkv + 8 * g;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 kf = *(const bf16x8*)(kr + 32 * ks);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qfl[ks], s[t], 0, 0, 0);
            }
        }
        stage_transposed(vb + (int64_t)k0 * ldkv, ldkv, tile, lane);
        float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        const float mb = m_new * LOG2E;
        float lsum = 0.f;
        bf16x8 pf, pfl;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(s[t][r] * LOG2E - mb);
                lsum += p;
                if (drop_thresh) p = drop_keep(seed, blockIdx.x, ql, k0 + 16 * t + 4 * g + r, drop_thresh) ? p * drop_scale : 0.f;
                split_hl(p, pf[4 * t + r], pfl[4 * t + r]);
            }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            o[dt] *= alpha;
            const bf16x8 vt = tile_frag(tile, dt, lane);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pf, o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pfl, o[dt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- merge the four waves' partials
    if (g == 0) { ml[0][wave][ql] = m_run; ml[1][wave][ql] = l_run; }
    __syncthreads();
    float m = fmaxf(fmaxf(ml[0][0][ql], ml[0][1][ql]), fmaxf(ml[0][2][ql], ml[0][3][ql]));
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) l += ml[1][w][ql] * __builtin_amdgcn_exp2f((ml[0][w][ql] - m) * LOG2E);
    const float sc = __builtin_amdgcn_exp2f((m_run - m) * LOG2E) / l;     // m_run = -inf (wave saw no keys) -> 0
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *(f32x4*)&obuf[wave][ql][dt * 16 + 4 * g] = o[dt] * sc;
    __syncthreads();
    for (int e = tid; e < Q * 64; e += 256) {
        const int qq = e >> 6, d = e & 63;
        out[((int64_t)b * Q + qq) * C + head * 64 + d] = (obuf[0][qq][d] + obuf[1][qq][d]) + (obuf[2][qq][d] + obuf[3][qq][d]);
    }
    if (tid < Q) {
        float mm = fmaxf(fmaxf(ml[0][0][tid], ml[0][1][tid]), fmaxf(ml[0][2][tid], ml[0][3][tid]));
        float ll = 0.f;
        for (int w = 0; w < 4; ++w) ll += ml[1][w][tid] * __builtin_amdgcn_exp2f((ml[0][w][tid] - mm) * LOG2E);
        lse[((int64_t)b * heads + head) * Q + tid] = mm + logf(ll);
    }
}

// out[b,q,c] = sum_s part_out[s,b,q,c] * exp(part_lse[s,b,h,q] - lse[b,h,q]),  lse = logsumexp_s part_lse: one block per (clip, head)
__global__ __launch_bounds__(256) void xattn_fwd_merge_kernel(const float* __restrict__ part_out, const float* __restrict__ part_lse,
                                                              float* __restrict__ out, float* __restrict__ lse, int B, int Q, int heads,
                                                              int splits) {
    __shared__ float w[64][16];
    const int tid = threadIdx.x, head = blockIdx.x % heads, b = blockIdx.x / heads, C = heads * 64;
    if (tid < Q) {
        const int64_t base = ((int64_t)b * heads + head) * Q + tid, plane = (int64_t)B * heads * Q;
        float m = -INFINITY;
        for (int s = 0; s < splits; ++s) m = fmaxf(m, part_lse[s * plane + base]);
        float l = 0.f;
        for (int s = 0; s < splits; ++s) l += expf(part_lse[s * plane + base] - m);
        const float tot = m + logf(l);
        lse[base] = tot;
        for (int s = 0; s < splits; ++s) w[s][tid] = expf(part_lse[s * plane + base] - tot);
    }
    __syncthreads();
    const int64_t plane = (int64_t)B * Q * C;
    for (int e = tid; e < Q * 64; e += 256) {
        const int qq = e >> 6, d = e & 63;
        const int64_t at = ((int64_t)b * Q + qq) * C + head * 64 + d;
        float a = 0.f;
        for (int s = 0; s < splits; ++s) a += part_out[s * plane + at] * w[s][qq];
        out[at] = a;
    }
}

__global__ __launch_bounds__(256) void xattn_bwd_kernel(const float* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, int64_t ldkv, const float* __restrict__ out,
                                                        const float* __restrict__ lse, const float* __restrict__ dout,
                                                        float* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv,
                                                        int64_t lddkv, int B, int Q, int M, int heads, unsigned drop_thresh,
                                                        float drop_scale, unsigned seed) {
    __shared__ __attribute__((aligned(16))) bf16_t tiles[4][64 * VSTRIDE];
    __shared__ __attribute__((aligned(16))) float qbuf[4][16][64];
    __shared__ float stat[2][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 15, g = lane >> 4;
    const int head = blockIdx.x % heads, b = blockIdx.x / heads;
    const int C = heads * 64;
    const float LOG2E = 1.4426950408889634f;
    const float* qrow = q + ((int64_t)b * Q) * C + head * 64;
    const float* dorow = dout + ((int64_t)b * Q) * C + head * 64;
    const float* orow = out + ((int64_t)b * Q) * C + head * 64;
    // delta[q] = sum_d dO*O, lse[q]
    if (tid < 64) {
        const int qq = tid & 15, part = tid >> 4;
        float acc = 0.f;
        if (qq < Q)
            for (int d = part * 16; d < part * 16 + 16; ++d) acc += dorow[(int64_t)qq * C + d] * orow[(int64_t)qq * C + d];
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        if (part == 0) {
            stat[0][qq] = acc;
            stat[1][qq] = (qq < Q) ? lse[((int64_t)b * heads + head) * Q + qq] : 0.f;
        }
    }
    __syncthreads();
    // row fragments (lane row/col = ql, d = 32ks + 8g + j) and transposed fragments (lane d = 16dt + ql, q = 4g + jj)
    bf16x8 qf[2], dof[2], qfl[2], dofl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        load_row_frag_f32_hl(qrow + (int64_t)(ql < Q ? ql : 0) * C + 32 * ks + 8 * g, ql < Q, qf[ks], qfl[ks]);
        load_row_frag_f32_hl(dorow + (int64_t)(ql < Q ? ql : 0) * C + 32 * ks + 8 * g, ql < Q, dof[ks], dofl[ks]);
    }
    s16x4 qT[4], doT[4], qTl[4], doTl[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        // row m = ql of transposed tile dt is d = 16 (m >> 2) + 4 dt + (m & 3): the dV^T / dK^T accumulators of the four tiles then give a
        // lane (key, g) the 16 CONSECUTIVE d 16 g .. 16 g + 15 -> two 16-B stores per key row and matrix (a full 128-B line per 4 lanes)
        // instead of eight 8-B stores scattered 32 B apart
        const int dperm = 16 * (ql >> 2) + 4 * dt + (ql & 3);
        bf16x4 a, c, al, cl;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int qq = 4 * g + jj;
            split_hl((qq < Q) ? qrow[(int64_t)qq * C + dperm] : 0.f, a[jj], al[jj]);
            split_hl((qq < Q) ? dorow[(int64_t)qq * C + dperm] : 0.f, c[jj], cl[jj]);
        }
        qT[dt] = __builtin_bit_cast(s16x4, a);
        doT[dt] = __builtin_bit_cast(s16x4, c);
        qTl[dt] = __builtin_bit_cast(s16x4, al);
        doTl[dt] = __builtin_bit_cast(s16x4, cl);
    }
    // per-lane row constants: non-swapped form has q = 4g + r in the registers, swapped form has q = ql on the lane
    float lse_r[4], del_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { lse_r[r] = stat[1][4 * g + r] * LOG2E; del_r[r] = stat[0][4 * g + r]; }
    const float lse_l = stat[1][ql] * LOG2E, del_l = stat[0][ql];

    const bf16_t* kb = k + (int64_t)b * M * ldkv + head * 64;
    const bf16_t* vb = v + (int64_t)b * M * ldkv + head * 64;
    bf16_t* dkb = dk + (int64_t)b * M * lddkv + head * 64;
    bf16_t* dvb = dv + (int64_t)b * M * lddkv + head * 64;
    bf16_t* tile = tiles[wave];
    f32x4 dqa[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dqa[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // blockIdx.y owns a contiguous slice of the keys (one workgroup per (clip, head) = 256 workgroups left three quarters of the
    // memory pipeline idle: 2 MB per CU at ~20 GB/s per CU); dq partials of the slices are summed by the caller in a fixed order
    const int m_per = ((M / 32 + gridDim.y - 1) / gridDim.y) * 32;
    const int m_lo = blockIdx.y * m_per, m_hi = min(M, m_lo + m_per);
    // Software pipeline over the wave's 32-key blocks: the K / V rows of block i + 1 (MFMA-layout fragments of both 16-key tiles and
    // the four rows per lane of the K^T staging) are requested before block i is computed.  Without it every block exposed the
    // memory latency of three dependent loads: 172 us per call at 3.1 TB/s of algorithmic bytes, MFMA busy 16 %, VALU 38 %.
    const int skg = lane >> 3, sc8 = lane & 7;
    bf16x8 nkf[2][2], nvf[2][2];
    u32x4 nst[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                nkf[t][ks] = *(const bf16x8*)(kb + (int64_t)(k0 + 16 * t + ql) * ldkv + 32 * ks + 8 * g);
                nvf[t][ks] = *(const bf16x8*)(vb + (int64_t)(k0 + 16 * t + ql) * ldkv + 32 * ks + 8 * g);
            }
        const bf16_t* sp = kb + (int64_t)(k0 + skg * 4) * ldkv + sc8 * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) nst[i] = *(const u32x4*)(sp + i * ldkv);
    };
    if (m_lo + wave * 32 < m_hi) fetch(m_lo + wave * 32);
    for (int k0 = m_lo + wave * 32; k0 < m_hi; k0 += 128) {
        bf16x8 ckf[2][2], cvf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { ckf[t][ks] = nkf[t][ks]; cvf[t][ks] = nvf[t][ks]; }
#pragma unroll
        for (int w = 0; w < 4; ++w) {                                        // K^T tile of this block (stage_transposed, from the prefetched rows)
            *(u32x2*)(tile + (sc8 * 8 + 2 * w) * VSTRIDE + skg * 4) = pick4(nst[0][w], nst[1][w], nst[2][w], nst[3][w], false);
            *(u32x2*)(tile + (sc8 * 8 + 2 * w + 1) * VSTRIDE + skg * 4) = pick4(nst[0][w], nst[1][w], nst[2][w], nst[3][w], true);
        }
        if (k0 + 128 < m_hi) fetch(k0 + 128);
        bf16x8 dsT, dsTl;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t key = k0 + 16 * t + ql;
            const bf16x8 (&kf)[2] = ckf[t];
            const bf16x8 (&vf)[2] = cvf[t];
            // non-swapped: lane = key, registers = q (4g + r)
            f32x4 s = zero, dp = zero;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfl[ks], kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[ks], vf[ks], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dofl[ks], vf[ks], dp, 0, 0, 0);
            }
            bf16x4 pb, dsb, pbl, dsbl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = (4 * g + r) < Q;
                const float p = live ? __builtin_amdgcn_exp2f(s[r] * LOG2E - lse_r[r]) : 0.f;
                float pd = p, dpm = dp[r];
                if (drop_thresh) {
                    const bool keep = drop_keep(seed, blockIdx.x, 4 * g + r, (unsigned)key, drop_thresh);
                    pd = keep ? p * drop_scale : 0.f;
                    dpm = keep ? dp[r] * drop_scale : 0.f;
                }
                split_hl(pd, pb[r], pbl[r]);
                split_hl(p * (dpm - del_r[r]), dsb[r], dsbl[r]);
            }
            const s16x4 pbs = __builtin_bit_cast(s16x4, pb), dsbs = __builtin_bit_cast(s16x4, dsb);
            const s16x4 pbls = __builtin_bit_cast(s16x4, pbl), dsbls = __builtin_bit_cast(s16x4, dsbl);
            unsigned wv[8], wk[8];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 gv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(doT[dt], pbs, zero, 0, 0, 0);   // dV^T[d = 16 g + 4 dt + j][key]
                gv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(doTl[dt], pbs, gv, 0, 0, 0);
                gv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(doT[dt], pbls, gv, 0, 0, 0);
                f32x4 gk = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qT[dt], dsbs, zero, 0, 0, 0);   // dK^T
                gk = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qTl[dt], dsbs, gk, 0, 0, 0);
                gk = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qT[dt], dsbls, gk, 0, 0, 0);
                wv[2 * dt] = pack_bf16(gv[0], gv[1]); wv[2 * dt + 1] = pack_bf16(gv[2], gv[3]);
                wk[2 * dt] = pack_bf16(gk[0], gk[1]); wk[2 * dt + 1] = pack_bf16(gk[2], gk[3]);
            }
            *(u32x4*)(dvb + key * lddkv + 16 * g) = (u32x4){wv[0], wv[1], wv[2], wv[3]};
            *(u32x4*)(dvb + key * lddkv + 16 * g + 8) = (u32x4){wv[4], wv[5], wv[6], wv[7]};
            *(u32x4*)(dkb + key * lddkv + 16 * g) = (u32x4){wk[0], wk[1], wk[2], wk[3]};
            *(u32x4*)(dkb + key * lddkv + 16 * g + 8) = (u32x4){wk[4], wk[5], wk[6], wk[7]};
            // swapped: lane = q (ql), registers = key (4g + r)
            f32x4 st = zero, dpt = zero;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[ks], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qfl[ks], st, 0, 0, 0);
                dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ks], dof[ks], dpt, 0, 0, 0);
                dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ks], dofl[ks], dpt, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (ql < Q) ? __builtin_amdgcn_exp2f(st[r] * LOG2E - lse_l) : 0.f;
                float dpm = dpt[r];
                if (drop_thresh) dpm = drop_keep(seed, blockIdx.x, ql, (unsigned)(k0 + 16 * t + 4 * g + r), drop_thresh) ? dpt[r] * drop_scale : 0.f;
                split_hl(p * (dpm - del_l), dsT[4 * t + r], dsTl[4 * t + r]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 kt = tile_frag(tile, dt, lane);
            dqa[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsT, dqa[dt], 0, 0, 0);
            dqa[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsTl, dqa[dt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *(f32x4*)&qbuf[wave][ql][dt * 16 + 4 * g] = dqa[dt];
    __syncthreads();
    for (int e = tid; e < Q * 64; e += 256) {
        const int qq = e >> 6, d = e & 63;
        dq[(((int64_t)blockIdx.y * B + b) * Q + qq) * C + head * 64 + d] = (qbuf[0][qq][d] + qbuf[1][qq][d]) + (qbuf[2][qq][d] + qbuf[3][qq][d]);
    }
}

static int xattn_check(const char* what, int B, int Q, int M, int heads, int64_t ld) {
    HH_REQUIRE(B >= 0 && Q > 0 && Q <= 16 && heads > 0 && M > 0 && M % 32 == 0, HH_ERR_SHAPE,
               "%s: need 0 < Q <= 16 and M %% 32 == 0 (Q=%d M=%d)", what, Q, M);
    HH_REQUIRE(ld >= heads * 64 && ld % 8 == 0, HH_ERR_SHAPE, "%s: k/v leading dimension %lld too small / unaligned", what, (long long)ld);
    return HH_OK;
}

extern "C" int hh_xattn_fwd(const float* q, const void* k, const void* v, int64_t ldkv, float* out, float* lse,
                            int B, int Q, int M, int heads, float dropout_p, uint32_t seed, hh_stream_t stream);

static void drop_params(float p, unsigned* thresh, float* scale) {
    if (p <= 0.f) { *thresh = 0u; *scale = 1.f; return; }
    double t = (double)p * 4294967296.0;
    *thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    if (*thresh == 0u) *thresh = 1u;
    *scale = 1.f / (1.f - p);
}

extern "C" int hh_xattn_fwd_split(const float* q, const void* k, const void* v, int64_t ldkv, float* out, float* lse, float* workspace,
                                  int splits, int B, int Q, int M, int heads, float dropout_p, uint32_t seed, hh_stream_t stream) {
    int rc = xattn_check("hh_xattn_fwd_split", B, Q, M, heads, ldkv);
    if (rc) return rc;
    HH_REQUIRE(splits >= 1 && splits <= 64, HH_ERR_SHAPE, "hh_xattn_fwd_split: splits must be in [1, 64]");
    HH_REQUIRE(splits == 1 || workspace, HH_ERR_SHAPE, "hh_xattn_fwd_split: splits > 1 needs hh_workspace_bytes_xattn_fwd() bytes of workspace");
    HH_REQUIRE(HH_ALIGNED16(q) && HH_ALIGNED16(k) && HH_ALIGNED16(v) && HH_ALIGNED16(out) && HH_ALIGNED16(workspace), HH_ERR_ALIGN,
               "hh_xattn_fwd_split: pointers must be 16-byte aligned");
    HH_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, HH_ERR_SHAPE, "hh_xattn_fwd_split: dropout_p must be in [0,1)");
    if (B == 0) return HH_OK;
    int chunk = (((M + splits - 1) / splits) + 127) / 128 * 128;           // whole 128-key rounds of the four waves
    splits = (M + chunk - 1) / chunk;                                       // no empty slice
    if (splits == 1) return hh_xattn_fwd(q, k, v, ldkv, out, lse, B, Q, M, heads, dropout_p, seed, stream);
    unsigned thr; float sc;
    drop_params(dropout_p, &thr, &sc);
    float* part_out = workspace;
    float* part_lse = workspace + (int64_t)splits * B * Q * heads * 64;
    {
        HHProfScope prof(HH_PROF_XATTN_FWD, 4.0 * (double)B * M * heads * 64, (hipStream_t)stream);
        hipLaunchKernelGGL(xattn_fwd_kernel, dim3((unsigned)(B * heads), (unsigned)splits), dim3(256), 0, (hipStream_t)stream, q, (const bf16_t*)k,
                           (const bf16_t*)v, ldkv, part_out, part_lse, B, Q, M, heads, thr, sc, seed, chunk);
    }
    hipLaunchKernelGGL(xattn_fwd_merge_kernel, dim3((unsigned)(B * heads)), dim3(256), 0, (hipStream_t)stream, part_out, part_lse, out, lse,
                       B, Q, heads, splits);
    return hh_check_launch("hh_xattn_fwd_split");
}

extern "C" int hh_xattn_fwd(const float* q, const void* k, const void* v, int64_t ldkv, float* out, float* lse,
                            int B, int Q, int M, int heads, float dropout_p, uint32_t seed, hh_stream_t stream) {
    int rc = xattn_check("hh_xattn_fwd", B, Q, M, heads, ldkv);
    if (rc) return rc;
    HH_REQUIRE(HH_ALIGNED16(q) && HH_ALIGNED16(k) && HH_ALIGNED16(v) && HH_ALIGNED16(out), HH_ERR_ALIGN, "hh_xattn_fwd: pointers must be 16-byte aligned");
    HH_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, HH_ERR_SHAPE, "hh_xattn_fwd: dropout_p must be in [0,1)");
    if (B == 0) return HH_OK;
    unsigned thr; float sc;
    drop_params(dropout_p, &thr, &sc);
    HHProfScope prof(HH_PROF_XATTN_FWD, 4.0 * (double)B * M * heads * 64, (hipStream_t)stream);         // K and V rows, bf16
    hipLaunchKernelGGL(xattn_fwd_kernel, dim3((unsigned)(B * heads)), dim3(256), 0, (hipStream_t)stream, q, (const bf16_t*)k,
                       (const bf16_t*)v, ldkv, out, lse, B, Q, M, heads, thr, sc, seed, M);
    return hh_check_launch("hh_xattn_fwd");
}

extern "C" int hh_xattn_bwd(const float* q, const void* k, const void* v, int64_t ldkv, const float* out, const float* lse,
                            const float* dout, float* dq, int dq_splits, void* dk, void* dv, int64_t lddkv, int B, int Q, int M, int heads,
                            float dropout_p, uint32_t seed, hh_stream_t stream) {
    int rc = xattn_check("hh_xattn_bwd", B, Q, M, heads, ldkv);
    if (rc) return rc;
    HH_REQUIRE(lddkv >= heads * 64 && lddkv % 8 == 0, HH_ERR_SHAPE, "hh_xattn_bwd: bad lddkv");
    HH_REQUIRE(HH_ALIGNED16(q) && HH_ALIGNED16(k) && HH_ALIGNED16(v) && HH_ALIGNED16(dout) && HH_ALIGNED16(dk) && HH_ALIGNED16(dv) &&
               HH_ALIGNED16(out) && HH_ALIGNED16(dq), HH_ERR_ALIGN, "hh_xattn_bwd: pointers must be 16-byte aligned");
    HH_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, HH_ERR_SHAPE, "hh_xattn_bwd: dropout_p must be in [0,1)");
    if (B == 0) return HH_OK;
    unsigned thr; float sc;
    drop_params(dropout_p, &thr, &sc);
    HH_REQUIRE(dq_splits >= 1 && dq_splits <= 64, HH_ERR_SHAPE, "hh_xattn_bwd: dq_splits must be in [1, 64]");
    HHProfScope prof(HH_PROF_XATTN_BWD, 8.0 * (double)B * M * heads * 64, (hipStream_t)stream);         // K, V read + dK, dV written
    hipLaunchKernelGGL(xattn_bwd_kernel, dim3((unsigned)(B * heads), (unsigned)dq_splits), dim3(256), 0, (hipStream_t)stream, q, (const bf16_t*)k,
                       (const bf16_t*)v, ldkv, out, lse, dout, dq, (bf16_t*)dk, (bf16_t*)dv, lddkv, B, Q, M, heads, thr, sc, seed);
    return hh_check_launch("hh_xattn_bwd");
}

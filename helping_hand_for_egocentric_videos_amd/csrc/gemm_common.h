// Shared pieces of the bf16 GEMM kernels (gemm.hip: 128x128 tile, gemm256.hip: 256x256 8-phase).
#pragma once
#include "common.h"

struct GemmParams {
    const bf16_t* A; int64_t lda;
    const bf16_t* W; int64_t ldw;
    void* C; int64_t ldc;
    int64_t M; int N; int K;
    int Mt, Nt;
    int debug_nostore;        // timing experiments only
    int debug_ts;             // persistent 256x256 kernel: record the per-tile timeline (debug)
    int group_m;              // 256x256 kernel: m-tiles per XCD-local group
    int rev_m;                // persistent 4-wave kernel: walk the m-tiles last to first (hh_gemm_epilogue.walk_reverse)
    int skew_iters;           // 256x256 kernel: start-time skew quantum (0 = off)
    int tile_rows;            // persistent 4-wave kernel: 256, or 224 (gemm256w4.hip: whole rounds where 256-row tiles leave a partial one; opt-in)
    int skew_phases;          // persistent 4-wave kernel: 0 = the quantum times (workgroup index in its XCD) & 31, P > 0 = times (index % P)
    int dynamic, tile_slot;   // 4-wave persistent kernel: tiles beyond a workgroup's first come from per-XCD atomic counters (slot of the launch stream)
    int64_t m_start;          // first row handled by this launch (rows [m_start, M) are tiled)
    int64_t tail_m;           // persistent 256x256 kernel: rows [tail_m, tail_m + tail_rows) (<= 64 rows behind the last full tile)
    int tail_rows;            //   are computed by the first N / 32 workgroups before their tile walk (0 = none)
    hh_gemm_epilogue e;
};

__device__ __forceinline__ float quick_gelu(float v) {      // x * sigmoid(1.702 x), openai_model.py:177-179
    return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * v));
}

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// Column term of a C address: row-major C -> n; column-blocked C (hh_gemm_epilogue.c_block_stride, ldc = 64) -> plane n / 64, column
// n % 64.  The 4 / 8 consecutive columns a lane stores never straddle a 64-column block.
__device__ __forceinline__ int64_t gemm_ccol(const hh_gemm_epilogue& e, int n) {
    return e.c_block_stride ? (int64_t)(n >> 6) * e.c_block_stride + (n & 63) : (int64_t)n;
}

// Epilogue for one accumulator tile in the swapped (C^T) layout: lane owns C[orow][n .. n+3].
// LayerNorm fold (include/hh.h): consumer side -- v <- rstd[m] * v - rstd[m] * mean[m] * colsum[n] + bias[n] (orow == m: no remap with it);
// producer side -- z = z_resid + (acc + bias) stored as bf16 beside C (its row statistics come from hh_ln_rowstats-style passes
// of the caller: gemm.hip).
template <bool OUT_BF16>
__device__ __forceinline__ void gemm_store4(const hh_gemm_epilogue& e, char* Cbase, int64_t ldc, int64_t orow, int n, f32x4 v) {
    if (e.ln_stats) {
        const f32x2 st = *(const f32x2*)(e.ln_stats + 2 * orow);
        v = v * st[0] + (*(const f32x4*)(e.ln_colsum + n) * st[1] + *(const f32x4*)(e.bias + n));
    } else if (e.bias) v += *(const f32x4*)(e.bias + n);
    if (n < e.colscale_cols) v *= e.colscale;
    if (e.act == HH_ACT_QUICKGELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = quick_gelu(v[q]);
    } else if (e.act == HH_ACT_RELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
    }
    if (e.resid) v += *(const f32x4*)(e.resid + orow * e.ldr + n);
    if (e.z_out) {
        f32x4 z;
        if (e.z_resid_lo) {                           // bf16 pair stream: x = hi + lo, both halves updated in place; hi' is the z of the next LayerNorm
            const u32x2 r = *(const u32x2*)((const bf16_t*)e.z_resid + orow * e.z_ldr + n);
            const u32x2 q = *(const u32x2*)((const bf16_t*)e.z_resid_lo + orow * e.z_ldr + n);
            z = (f32x4){bf16_lo_to_f32(r[0]) + bf16_lo_to_f32(q[0]), bf16_hi_to_f32(r[0]) + bf16_hi_to_f32(q[0]),
                        bf16_lo_to_f32(r[1]) + bf16_lo_to_f32(q[1]), bf16_hi_to_f32(r[1]) + bf16_hi_to_f32(q[1])} + v;
            const u32x2 h = {pack_bf16(z[0], z[1]), pack_bf16(z[2], z[3])};
            const u32x2 l = {pack_bf16(z[0] - bf16_lo_to_f32(h[0]), z[1] - bf16_hi_to_f32(h[0])), pack_bf16(z[2] - bf16_lo_to_f32(h[1]), z[3] - bf16_hi_to_f32(h[1]))};
            *(u32x2*)((bf16_t*)e.z_resid + orow * e.z_ldr + n) = h;
            *(u32x2*)((bf16_t*)e.z_resid_lo + orow * e.z_ldr + n) = l;
            return;
        }
        if (e.z_resid_dtype == HH_BF16) {
            const u32x2 r = *(const u32x2*)((const bf16_t*)e.z_resid + orow * e.z_ldr + n);
            z = (f32x4){bf16_lo_to_f32(r[0]), bf16_hi_to_f32(r[0]), bf16_lo_to_f32(r[1]), bf16_hi_to_f32(r[1])} + v;
        } else z = *(const f32x4*)((const float*)e.z_resid + orow * e.z_ldr + n) + v;
        if (e.z_update) *(f32x4*)((float*)e.z_resid + orow * e.z_ldr + n) = z;      // x <- x + branch, in place (this lane read the same 16 bytes)
        u32x2 o = {pack_bf16(z[0], z[1]), pack_bf16(z[2], z[3])};
        *(u32x2*)((bf16_t*)e.z_out + orow * e.z_ldc + n) = o;
        if (e.skip_c) return;
    }
    if constexpr (OUT_BF16) {
        u32x2 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
        *(u32x2*)((bf16_t*)Cbase + orow * ldc + gemm_ccol(e, n)) = o;
    } else {
        *(f32x4*)((float*)Cbase + orow * ldc + gemm_ccol(e, n)) = v;
    }
}

// 8 consecutive columns per lane (256x256 kernel): one 16-byte store for bf16, two for fp32.
template <bool OUT_BF16>
__device__ __forceinline__ void gemm_store8(const hh_gemm_epilogue& e, char* Cbase, int64_t ldc, int64_t orow, int n, f32x4 v0, f32x4 v1) {
    if (e.bias) { v0 += *(const f32x4*)(e.bias + n); v1 += *(const f32x4*)(e.bias + n + 4); }
    if (n < e.colscale_cols) { v0 *= e.colscale; v1 *= e.colscale; }
    if (e.act == HH_ACT_QUICKGELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0[q] = quick_gelu(v0[q]); v1[q] = quick_gelu(v1[q]); }
    } else if (e.act == HH_ACT_RELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
    }
    if (e.resid) { v0 += *(const f32x4*)(e.resid + orow * e.ldr + n); v1 += *(const f32x4*)(e.resid + orow * e.ldr + n + 4); }
    if constexpr (OUT_BF16) {
        u32x4 o = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
        *(u32x4*)((bf16_t*)Cbase + orow * ldc + gemm_ccol(e, n)) = o;
    } else {
        *(f32x4*)((float*)Cbase + orow * ldc + gemm_ccol(e, n)) = v0;
        *(f32x4*)((float*)Cbase + orow * ldc + gemm_ccol(e, n) + 4) = v1;
    }
}

// include/hh.h: LayerNorm-fold fields of the epilogue in use (only gemm_store4 and the persistent 4-wave kernel implement them)
__host__ __device__ __forceinline__ bool gemm_ln_ext(const hh_gemm_epilogue& e) { return e.ln_stats != nullptr || e.z_out != nullptr; }

int hh_gemm256_tile_rows(const GemmParams& p, hipStream_t s);                  // gemm256.hip: 256, or 224 where that removes a partial round
int hh_gemm256_launch(const GemmParams& p, hipStream_t s, bool* tail_done);   // gemm256.hip; returns HH_OK or an error; *tail_done: p.tail_rows were computed
bool hh_gemm256_eligible(const GemmParams& p);

// Causal self-attention of the CLIP text tower (model/openai_model.py:182-232 resblocks: nn.MultiheadAttention with the causal
// mask of model/LaviLa.py:636-642; context length 77, width 768 = 12 heads x 64).  One WAVE per (caption, head): the whole
// problem (77 x 77 x 64) lives in its registers and 12 KB of LDS.  Same operand tricks as the time-attention kernel:
//   S^T = K . Q^T      v_mfma_f32_16x16x32_bf16, A = K rows, B = Q rows, both loaded from HBM directly in MFMA layout (16-B loads)
//   softmax            lane = query, 4 registers per key tile + xor-16 / xor-32 exchanges; causal mask key <= query
//   O^T = V^T . P^T    two key tiles per MFMA (k-slots jj < 4 -> tile 2u row 4g+jj, jj >= 4 -> tile 2u+1 row 4g+jj-4); V^T from the
//                      row-major V tile in LDS (LDS-DMA) through ds_read_b64_tr_b16, rows permuted so a lane ends with 16 consecutive d
// q is pre-scaled by d^-0.5 in the QKV GEMM epilogue.  Replaces F.scaled_dot_product_attention + two permute copies per layer
// (174 us -> see DESIGN.md).
#include "common.h"

typedef short s16x4_x __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void xglds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ bf16x4 xlds_tr4(const char* addr) {
    s16x4_x r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_x*)addr);
    return __builtin_bit_cast(bf16x4, r);
}

#define TX_TILES 5                     // 16-row tiles: context length <= 80
#define TX_VROWS 96                    // V rows staged per wave (3 MFMA k-groups of 32 keys)

__global__ __launch_bounds__(256) void text_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int S, int L, int heads) {
    __shared__ __attribute__((aligned(16))) char Vsm[4 * TX_VROWS * 128];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = heads * 64;
    const int64_t ld = 3 * (int64_t)W;
    const int wid = blockIdx.x * 4 + wave;
    if (wid >= S * heads) return;                                  // waves are independent: no workgroup barrier below
    const int s = wid / heads, head = wid % heads;
    const bf16_t* base = qkv + (int64_t)s * L * ld + head * 64;
    const int c = lane & 15, g = lane >> 4;
    const float LOG2E = 1.4426950408889634f;
    char* vbuf = Vsm + wave * (TX_VROWS * 128);
    // rows beyond the caption repeat its last row: finite values that only ever meet zero probabilities
#pragma unroll
    for (int i = 0; i < TX_VROWS / 8; ++i) {
        const int r = min(8 * i + (lane >> 3), L - 1);
        xglds16(base + r * ld + 2 * W + (lane & 7) * 8, vbuf + i * 1024);
    }
    bf16x8 kf[TX_TILES][2], qf[TX_TILES][2];
#pragma unroll
    for (int t = 0; t < TX_TILES; ++t) {
        const bf16_t* rp = base + min(16 * t + c, L - 1) * ld + 8 * g;
        qf[t][0] = *(const bf16x8*)(rp);
        qf[t][1] = *(const bf16x8*)(rp + 32);
        kf[t][0] = *(const bf16x8*)(rp + W);
        kf[t][1] = *(const bf16x8*)(rp + W + 32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int trq = c >> 2, trp = c & 3;
#pragma unroll
    for (int qt = 0; qt < TX_TILES; ++qt) {
        if (16 * qt >= L) break;
        const int query = 16 * qt + c;
        f32x4 sc[TX_TILES];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt <= qt; ++kt) {
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], qf[qt][0], z4, 0, 0, 0);
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][1], qf[qt][1], sc[kt], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (16 * kt + 4 * g + j > query) sc[kt][j] = -INFINITY;            // causal (also hides keys >= L from stored queries)
                m = fmaxf(m, sc[kt][j]);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float mb = m * LOG2E;
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt <= qt; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sc[kt][j] = __builtin_amdgcn_exp2f(sc[kt][j] * LOG2E - mb);
                l += sc[kt][j];
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(l);
        f32x4 o[4] = {z4, z4, z4, z4};
#pragma unroll
        for (int u = 0; u <= qt / 2; ++u) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pf[j] = (bf16_t)(sc[2 * u][j] * inv);
                pf[4 + j] = (2 * u + 1 <= qt) ? (bf16_t)(sc[2 * u + 1 <= qt ? 2 * u + 1 : 0][j] * inv) : (bf16_t)0.f;
            }
            const char* vb = vbuf + (32 * u + 4 * g + trq) * 128 + 32 * trp;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x4 t0 = xlds_tr4(vb + 8 * dt), t1 = xlds_tr4(vb + 16 * 128 + 8 * dt);
                const bf16x8 af = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pf, o[dt], 0, 0, 0);
            }
        }
        if (query < L) {
            bf16_t* op = out + ((int64_t)s * L + query) * W + head * 64 + 16 * g;
            const u32x4 w0 = {pack_bf16(o[0][0], o[0][1]), pack_bf16(o[0][2], o[0][3]), pack_bf16(o[1][0], o[1][1]), pack_bf16(o[1][2], o[1][3])};
            const u32x4 w1 = {pack_bf16(o[2][0], o[2][1]), pack_bf16(o[2][2], o[2][3]), pack_bf16(o[3][0], o[3][1]), pack_bf16(o[3][2], o[3][3])};
            *(u32x4*)(op) = w0;
            *(u32x4*)(op + 8) = w1;
        }
    }
}

extern "C" int hh_text_attn_fwd(const void* qkv, void* out, int S, int L, int heads, hh_stream_t stream) {
    HH_REQUIRE(S >= 0 && L > 0 && L <= 16 * TX_TILES && heads > 0, HH_ERR_SHAPE,
               "hh_text_attn_fwd: context length %d unsupported (1..%d), head dim must be 64", L, 16 * TX_TILES);
    HH_REQUIRE(HH_ALIGNED16(qkv) && HH_ALIGNED16(out), HH_ERR_ALIGN, "hh_text_attn_fwd: pointers must be 16-byte aligned");
    if (S == 0) return HH_OK;
    const int64_t waves = (int64_t)S * heads;
    hipLaunchKernelGGL(text_attn_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qkv, (bf16_t*)out, S, L, heads);
    return hh_check_launch("hh_text_attn_fwd");
}

// Weight-gradient GEMM in its natural layout:  C[m, n] = sum_k At[k, m] * Bt[k, n]  with BOTH operands k-major (row = token).
// dW = dY^T . X of an nn.Linear contracts over the tokens (K = B*M = 131 104 here), and dY [tokens, out] / X [tokens, in] are
// exactly At / Bt as they sit in memory -- the NT kernel (gemm.hip) needed both transposed first: 6 transposes, 3.7 GB of HBM
// traffic and 1.8 ms per training step (tfm_decoder.py K/V in-projection and memory projection of all 6 layers).
//
// 128 x 128 tile per workgroup (4 waves as 2 x 2, wave tile 64 x 64 = 4 x 4 v_mfma_f32_16x16x32_bf16), BK = 64 tokens.  Tiles are
// staged k-major in LDS by LDS-DMA (row = 128 columns = 256 B = 16 lanes x 16 B); an MFMA operand wants 8 consecutive k per lane,
// which the hardware-transposing ds_read_b64_tr_b16 delivers from the k-major tile (two reads of 4 k each).  16-B chunk c of row
// r lives at position c ^ tn_key(r), tn_key(r) = 2 (r & 3) + 8 ((r >> 3) & 1): the transposing read is banked per 32-lane half = two 16-lane
// groups 8 rows apart, each touching 4 rows x 32 B -- row bits 0-1 spread a group's rows, row bit 3 keeps the two groups apart (round 6;
// without it the halves collided 2-way: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 2.0, profiles/r5_mattn_pmc.txt).
// Split-K over the token dimension: blockIdx.y owns a 64-aligned slice and writes its fp32 partial tile; rows beyond K read a
// zero line.  Operands are swapped as in gemm.hip (accumulator = C^T tile) so that a lane owns 4 consecutive n of one m.
#include "common.h"

typedef short s16x4_tn __attribute__((ext_vector_type(4)));
__device__ __attribute__((aligned(16))) char g_tn_zero[16];

__device__ __forceinline__ void tn_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ int tn_key(int r) { return (2 * (r & 3)) | (((r >> 3) & 1) << 3); }
__device__ __forceinline__ bf16x4 tn_tr4(const char* addr) {
    s16x4_tn r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_tn*)addr);
    return __builtin_bit_cast(bf16x4, r);
}

struct TnParams {
    const bf16_t* At; int64_t lda;
    const bf16_t* Bt; int64_t ldb;
    float* C; int64_t split_stride;        // partials [splits][M][N]
    float* colsum;                         // optional [splits][M]: sum_k At[k, m] of the slice (the bias gradient of dY = At), or NULL
    int M, N; int64_t K;
    int k_per_split;                       // multiple of 64
    // batched two-pair form (hh_gemm_tn_bf16_batched2): C_z = At_z^T Bt_z + At2_z^T Bt2_z, z = blockIdx.z, no split-K
    const bf16_t* At2; const bf16_t* Bt2;  // second operand pair (same leading dimensions and K), or NULL
    int64_t sA, sB, sC;                    // batch strides in elements
};

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 16384];       // [stage][A | B][64 rows x 256 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int Nt = p.N / 128;
    const int m0 = (blockIdx.x / Nt) * 128, n0 = (blockIdx.x % Nt) * 128;
    const int64_t k_begin = (int64_t)blockIdx.y * p.k_per_split;
    int64_t k_end = k_begin + p.k_per_split;
    if (k_end > p.K) k_end = p.K;
    float* Cp = p.C + (int64_t)blockIdx.y * p.split_stride + (int64_t)blockIdx.z * p.sC;
    const int nk1 = (int)((k_end - k_begin + 63) / 64);
    const int nk = p.At2 ? 2 * nk1 : nk1;                                  // the second pair's k-tiles follow the first's (batched form: one slice)
    const bf16_t* A1 = p.At + (int64_t)blockIdx.z * p.sA;
    const bf16_t* B1 = p.Bt + (int64_t)blockIdx.z * p.sB;
    const bf16_t* A2 = p.At2 ? p.At2 + (int64_t)blockIdx.z * p.sA : nullptr;
    const bf16_t* B2 = p.Bt2 ? p.Bt2 + (int64_t)blockIdx.z * p.sB : nullptr;

    // staging: wave w stages rows 16 w .. 16 w + 15 of both operands: 4 instructions x 4 rows each
    const int srow = lane >> 4, schunk = lane & 15;
    auto stage = [&](int kt, int buf) {
        char* dA = smem + buf * 32768 + wave * 4096;
        char* dB = dA + 16384;
        const bool second = kt >= nk1;
        const bf16_t* Ap = second ? A2 : A1;
        const bf16_t* Bp = second ? B2 : B1;
        if (second) kt -= nk1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 16 + 4 * i + srow;                      // row inside the k-tile
            const int c = schunk ^ tn_key(r);                             // logical chunk that lives at this lane's position
            const int64_t k = k_begin + (int64_t)kt * 64 + r;
            const bool ok = k < k_end;
            tn_glds16(ok ? (const void*)(Ap + k * p.lda + m0 + c * 8) : (const void*)g_tn_zero, dA + i * 1024);
            tn_glds16(ok ? (const void*)(Bp + k * p.ldb + n0 + c * 8) : (const void*)g_tn_zero, dB + i * 1024);
        }
    };
    // fragment reads: 16-lane group g of the wave reads k rows 8g .. 8g+7 (+ 32 per k-step); lane i = 4 q + pc inside the group
    // supplies the address of row q, columns 4 pc .. 4 pc + 3 of the 16-column MFMA tile
    const int g = lane >> 4, q = (lane & 15) >> 2, pc = lane & 3;
    auto frag = [&](const char* tile, int col0, int ks) -> bf16x8 {     // col0: first column of the 16-wide MFMA tile
        const int r0 = 32 * ks + 8 * g + q, r1 = r0 + 4;
        const int ch = (col0 >> 3) + (pc >> 1);                           // 16-B chunk of this lane's 8-B piece
        const bf16x4 a = tn_tr4(tile + r0 * 256 + ((ch ^ tn_key(r0)) << 4) + (pc & 1) * 8);
        const bf16x4 b = tn_tr4(tile + r1 * 256 + ((ch ^ tn_key(r1)) << 4) + (pc & 1) * 8);
        return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    };

    // column sums of At as a by-product (bias gradient of the nn.Linear whose dY is At): the workgroups of n-tile 0, waves wn == 0,
    // run one more MFMA per A fragment against an all-ones operand -- every row of that C^T tile is sum_k At[k, m]
    const bool want_cs = p.colsum != nullptr && n0 == 0 && wn == 0;
    const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
    f32x4 cs[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 acc[4][4];                                                      // [tn][tm]: C^T tiles (rows n, columns m)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nk > 0) stage(0, 0);
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                  // tile t landed everywhere; everyone left buffer cur ^ 1
        if (t + 1 < nk) stage(t + 1, cur ^ 1);
        const char* tA = smem + cur * 32768;
        const char* tB = tA + 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = frag(tA, wm * 64 + i * 16, ks);
                bf[i] = frag(tB, wn * 64 + i * 16, ks);
            }
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tn], af[tm], acc[tn][tm], 0, 0, 0);
            if (want_cs) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) cs[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[tm], cs[tm], 0, 0, 0);
            }
        }
    }
    // accumulator tile [tn][tm]: lane (c = lane & 15 -> m, g -> n rows 4g .. 4g+3)
    const int mrow = lane & 15;
    if (want_cs && g == 0) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) p.colsum[(int64_t)blockIdx.y * p.M + m0 + wm * 64 + tm * 16 + mrow] = cs[tm][0];
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const int m = m0 + wm * 64 + tm * 16 + mrow;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int n = n0 + wn * 64 + tn * 16 + 4 * g;
            *(f32x4*)(Cp + (int64_t)m * p.N + n) = acc[tn][tm];
        }
    }
}

extern "C" int hh_gemm_tn_bf16(const void* At, int64_t lda, const void* Bt, int64_t ldb, float* partials, float* colsum_partials, int M,
                               int N, int64_t K, int splits, hh_stream_t stream) {
    HH_REQUIRE(M > 0 && N > 0 && K > 0 && M % 128 == 0 && N % 128 == 0, HH_ERR_SHAPE,
               "hh_gemm_tn_bf16: need M %% 128 == 0 and N %% 128 == 0 (M=%d N=%d K=%lld)", M, N, (long long)K);
    HH_REQUIRE(lda >= M && ldb >= N && lda % 8 == 0 && ldb % 8 == 0, HH_ERR_SHAPE, "hh_gemm_tn_bf16: bad leading dimensions");
    HH_REQUIRE(HH_ALIGNED16(At) && HH_ALIGNED16(Bt) && HH_ALIGNED16(partials), HH_ERR_ALIGN, "hh_gemm_tn_bf16: pointers must be 16-byte aligned");
    HH_REQUIRE(splits >= 1 && splits <= 4096, HH_ERR_SHAPE, "hh_gemm_tn_bf16: splits out of range");
    TnParams p;
    p.At = (const bf16_t*)At; p.lda = lda; p.Bt = (const bf16_t*)Bt; p.ldb = ldb; p.C = partials; p.split_stride = (int64_t)M * N; p.colsum = colsum_partials;
    p.M = M; p.N = N; p.K = K;
    p.At2 = p.Bt2 = nullptr; p.sA = p.sB = p.sC = 0;
    const int64_t ktiles = (K + 63) / 64;
    p.k_per_split = (int)(((ktiles + splits - 1) / splits) * 64);
    HHProfScope prof(HH_PROF_GEMM_TN, 2.0 * (double)M * N * (double)K, (hipStream_t)stream);
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)((M / 128) * (N / 128)), (unsigned)splits), dim3(256), 0, (hipStream_t)stream, p);
    return hh_check_launch("hh_gemm_tn_bf16");
}

// out[i] = sum_s partials[s, i]: the split-K planes of hh_gemm_tn_bf16 added up in plane order (deterministic), 16 B per lane and
// four planes in flight; replaces the stock reduction over dim 0 (about 200 us for 64 MB of planes; this one reads them at HBM speed).
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int splits, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4* p = (const f32x4*)part + i;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= splits; s += 4) {
        const f32x4 a = __builtin_nontemporal_load(p + (int64_t)s * n4), b = __builtin_nontemporal_load(p + (int64_t)(s + 1) * n4);
        const f32x4 c = __builtin_nontemporal_load(p + (int64_t)(s + 2) * n4), d = __builtin_nontemporal_load(p + (int64_t)(s + 3) * n4);
        acc += a; acc += b; acc += c; acc += d;
    }
    for (; s < splits; ++s) acc += __builtin_nontemporal_load(p + (int64_t)s * n4);
    ((f32x4*)out)[i] = acc;
}

extern "C" int hh_sum_partials(const float* partials, float* out, int splits, int64_t n, hh_stream_t stream) {
    HH_REQUIRE(splits >= 1 && splits <= 4096 && n >= 0 && n % 4 == 0, HH_ERR_SHAPE, "hh_sum_partials: need 1 <= splits <= 4096 and n %% 4 == 0 (splits=%d n=%lld)",
               splits, (long long)n);
    HH_REQUIRE(HH_ALIGNED16(partials) && HH_ALIGNED16(out), HH_ERR_ALIGN, "hh_sum_partials: pointers must be 16-byte aligned");
    if (n == 0) return HH_OK;
    const int64_t n4 = n / 4, blocks = (n4 + 255) / 256;
    HH_REQUIRE(blocks <= 0x7fffffff, HH_ERR_SHAPE, "hh_sum_partials: n too large");
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partials, out, splits, n4);
    return hh_check_launch("hh_sum_partials");
}

// C_z [M, N] fp32 = At_z^T Bt_z (+ At2_z^T Bt2_z), z = 0 .. batch - 1: the d-memory GEMM of the K/V-projection-free decoder cross-attention
// (mattn.hip): per clip, At = the transposed probabilities Pd^T [rows, M keys], Bt = the pooled-row gradients [rows, 512]; second pair =
// the score gradients dS^T and the mapped queries; rows = 6 layers x 128.  K (rows) is short, M (keys) long: one workgroup per 128 x 128
// output tile walks all of K, no split-K, no partial planes.
extern "C" int hh_gemm_tn_bf16_batched2(const void* At, const void* Bt, const void* At2, const void* Bt2, int64_t lda, int64_t ldb, int64_t stride_a,
                                        int64_t stride_b, float* C, int64_t stride_c, int M, int N, int64_t K, int batch, hh_stream_t stream) {
    HH_REQUIRE(M > 0 && N > 0 && K > 0 && M % 128 == 0 && N % 128 == 0 && batch >= 0 && batch <= 65535, HH_ERR_SHAPE,
               "hh_gemm_tn_bf16_batched2: need M %% 128 == 0, N %% 128 == 0, batch <= 65535 (M=%d N=%d K=%lld batch=%d)", M, N, (long long)K, batch);
    HH_REQUIRE((At2 == nullptr) == (Bt2 == nullptr) && (At2 == nullptr || K % 64 == 0), HH_ERR_SHAPE, "hh_gemm_tn_bf16_batched2: the second pair comes whole and needs K %% 64 == 0");
    HH_REQUIRE(lda >= M && ldb >= N && lda % 8 == 0 && ldb % 8 == 0 && stride_a % 8 == 0 && stride_b % 8 == 0 && stride_c % 4 == 0, HH_ERR_SHAPE,
               "hh_gemm_tn_bf16_batched2: bad leading dimensions / strides");
    HH_REQUIRE(HH_ALIGNED16(At) && HH_ALIGNED16(Bt) && HH_ALIGNED16(At2) && HH_ALIGNED16(Bt2) && HH_ALIGNED16(C), HH_ERR_ALIGN, "hh_gemm_tn_bf16_batched2: pointers must be 16-byte aligned");
    if (batch == 0) return HH_OK;
    TnParams p;
    p.At = (const bf16_t*)At; p.lda = lda; p.Bt = (const bf16_t*)Bt; p.ldb = ldb; p.C = C; p.split_stride = 0; p.colsum = nullptr;
    p.M = M; p.N = N; p.K = K;
    p.k_per_split = (int)(((K + 63) / 64) * 64);
    p.At2 = (const bf16_t*)At2; p.Bt2 = (const bf16_t*)Bt2; p.sA = stride_a; p.sB = stride_b; p.sC = stride_c;
    HHProfScope prof(HH_PROF_GEMM_TN, 2.0 * (double)M * N * (double)K * (At2 ? 2 : 1) * batch, (hipStream_t)stream);
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)((M / 128) * (N / 128)), 1u, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, p);
    return hh_check_launch("hh_gemm_tn_bf16_batched2");
}

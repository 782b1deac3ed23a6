// Time attention core of the divided space-time attention (model/LaviLa.py:246-283 with '(b n) f d').
// One problem per (clip, head, patch location): T queries x (T frame keys + CLS key), head dim 64.
// Pure HBM-bound op (~(T+1)/2 flop/B): algorithmic bytes = every q,k,v row read once + every o row written once.
//
// Two kernels:
//   time_attn_mfma_kernel<T>   T <= 16 (default): one wave per 128 tokens, everything on the matrix core (see its header below)
//   time_attn_mfma32_kernel    T = 32 (BASELINE config 4): a patch location spans two 16-row tiles
// The first, VALU-only implementation (a workgroup owns 128/T neighbouring patch locations, K and V rows staged once into LDS, 2
// lanes per query walk the T+1 keys with broadcast LDS reads: 392 us per call at B = 32 against 244 us) was removed in round 2.
#include "common.h"

#define CLS_REC 68

// ---------------------------------------------------------------------------------------------------------------------
// MFMA variant (T <= 16, T a power of two): the VALU kernel above issues ~1200 vector instructions per (clip, head, patch)
// problem (SQ_INSTS_VALU = 159 M per launch at B = 32: 260 us of pure VALU issue), which -- not HBM -- bounded it at ~400 us.
// Here ONE WAVE owns 128 consecutive tokens of a (clip, head) = 128/T patch locations = 8 tiles of 16 rows (row r of a tile =
// patch r / T, frame r % T) and runs every product on the matrix core with operands that come from HBM already in MFMA layout:
//   S^T = K . Q^T          v_mfma_f32_16x16x32_bf16 x2, A = K rows, B = Q rows, both plain 16-B global loads (lane = row, 8 d)
//   CLS key column         x2, A = K_cls replicated in rows 0/4/8/12 -> register 0 of every lane = q . k_cls
//   softmax over keys      4 registers in-lane + xor-16 / xor-32 exchanges (lane = query); for T < 16 the tile is block-diagonal
//   O^T = V^T . P^T        x4 (one per 16 d), B = normalised P^T straight from the S^T registers (k-slot 8g+jj: jj < 4 -> key
//                          row 4g+jj, jj = 4 of g = 0 -> CLS key), A = V^T from the row-major V tile in LDS (staged by LDS-DMA,
//                          2 KB per tile, double-buffered per wave) through the transposing ds_read_b64_tr_b16; row m of tile
//                          dt is d = 16 (m >> 2) + 4 dt + (m & 3), so that a lane ends up with 16 CONSECUTIVE d of its query
//                          (two 16-B stores)
//   CLS query (folded)     the same MFMAs with B = q_cls in all 16 columns, online-softmax accumulated over the wave's 8 tiles
//                          into one partial record per wave (same record grid as the VALU kernel: ceil(n / (128/T)) per head).
// No workgroup barrier: the four waves of a workgroup are independent.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tglds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ bf16x4 tlds_tr4(const char* addr) {
    s16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)addr);
    return __builtin_bit_cast(bf16x4, r);
}
// The same read as opaque inline asm (callers follow a group of them with tlds_wait()).  hipcc puts s_waitcnt vmcnt(0) in front of
// every LDS read that follows an LDS-DMA it cannot prove disjoint -- which also waits for the write acknowledgements of the
// previous tile's output stores; with the read hidden, the counted wait the kernel issues itself is the only one.
template <int OFF>
__device__ __forceinline__ bf16x4 tlds_tr4_raw(unsigned lds_addr) {
    s16x4_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
    return __builtin_bit_cast(bf16x4, r);
}
__device__ __forceinline__ void tlds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_u32(const char* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }

// s_waitcnt vmcnt(N) as the BUILTIN (not inline asm): hipcc's wait-count pass sees it and knows the older loads have landed.  With
// an opaque asm wait it re-waits with vmcnt(0) at the first use of the loop-carried Q / K registers -- AFTER the next tile's
// loads have been issued, i.e. it waits for the prefetch it is supposed to overlap with.
#define HH_WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (15 << 8) | (((N) >> 4) << 14))

// Round 6: any frame count.  T is the tile's frame SLOT count (a power of two); PAD = the clip has Tr < T real frames (Tr = 3, 5, 12 ...): rows
// r with r % T >= Tr are padding -- their loads are clamped to the last real frame (valid memory), their keys masked, their queries never
// stored, the CLS query does not see them.  PAD = false compiles to exactly the round-5 kernel.
template <int T, bool PAD>
__global__ __launch_bounds__(256) void time_attn_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                             float* __restrict__ cls_partial, int B, int n, int heads, int layout, int Tr) {
    constexpr int P = 128 / T;         // patch locations per wave (= per CLS partial record)
    constexpr int PT = 16 / T;         // patch locations per 16-row tile
    __shared__ __attribute__((aligned(16))) char Vsm[4 * 2 * 2048];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = heads * 64;
    const int N = 1 + (PAD ? Tr : T) * n;
    // qkv layout (include/hh.h, hh_qkv_layout): element (row, which, head, d) at row * ld + which * ws + head * hs + d
    const int rev = layout >> 1;                       // (bit 1 of the layout argument, HH_QKV_WALK_REVERSE: the problems last to first)
    layout &= 1;
    const int64_t ld = layout ? 64 : 3 * (int64_t)D;
    const int64_t hs = layout ? (int64_t)B * N * 64 : 64, ws = (int64_t)heads * hs;
    const int groups = (n + P - 1) / P;
    int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    if (wid >= (int64_t)B * heads * groups) return;
    if (rev) wid = (int64_t)B * heads * groups - 1 - wid;
    const int pg = (int)(wid % groups); wid /= groups;
    const int head = (int)(wid % heads);
    const int b = (int)(wid / heads);
    const bf16_t* base = qkv + (int64_t)b * N * ld + head * hs;
    const int p0 = pg * P;
    const int ntiles = min(8, (n - p0 + PT - 1) / PT);
    const int c = lane & 15, g = lane >> 4;
    const float LOG2E = 1.4426950408889634f;
    char* vbuf = Vsm + wave * 4096;

    // token row of tile t, tile row r; patches beyond n are clamped (their keys only meet their own, never stored, queries)
    auto tokrow = [&](int t, int r) -> int64_t {
        const int patch = min(p0 + t * PT + r / T, n - 1);
        const int fr = PAD ? min(r % T, Tr - 1) : r % T;
        return 1 + (int64_t)fr * n + patch;
    };
    auto issue = [&](int t, bf16x8 (&qq)[2], bf16x8 (&kk)[2]) {
        char* dst = vbuf + (t & 1) * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            tglds16(base + tokrow(t, (lane >> 3) + 8 * i) * ld + 2 * ws + (lane & 7) * 8, dst + i * 1024);
        const bf16_t* rp = base + tokrow(t, c) * ld + 8 * g;
        qq[0] = *(const bf16x8*)(rp);
        qq[1] = *(const bf16x8*)(rp + 32);
        kk[0] = *(const bf16x8*)(rp + ws);
        kk[1] = *(const bf16x8*)(rp + ws + 32);
    };

    // per-(clip, head) constants: CLS key as an A tile (rows 0, 4, 8, 12), CLS query as a B tile (every column), CLS value slot
    bf16x8 a2[2], qc[2];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kc = *(const bf16x8*)(base + ws + 8 * g + 32 * ks);
        a2[ks] = (c & 3) == 0 ? kc : zero8;
        qc[ks] = *(const bf16x8*)(base + 8 * g + 32 * ks);
    }
    bf16_t vcls[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const bf16_t v = base[2 * ws + 16 * (c >> 2) + 4 * dt + (c & 3)];
        vcls[dt] = g == 0 ? v : (bf16_t)0.f;
    }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    float mc = -INFINITY, lc = 0.f;
    f32x4 accc[4] = {z4, z4, z4, z4};
    const int trq = c >> 2, trp = c & 3;                 // ds_read_tr roles inside a 16-lane group: row, 4-column piece

    // V^T fragments of tile t.  Read BEFORE the next tile's LDS-DMA is issued: hipcc drains every outstanding LDS-DMA (vmcnt(0)) in
    // front of an LDS read it cannot prove disjoint, which would serialise the prefetch behind the PV step
    auto vfrags = [&](int t, bf16x4 (&vf)[4]) {
        const unsigned vb = lds_u32(vbuf + (t & 1) * 2048 + (4 * g + trq) * 128 + 32 * trp);
        vf[0] = tlds_tr4_raw<0>(vb); vf[1] = tlds_tr4_raw<8>(vb); vf[2] = tlds_tr4_raw<16>(vb); vf[3] = tlds_tr4_raw<24>(vb);
        tlds_wait();
    };
    // Every MFMA that reads the (loop-carried) Q / K registers is issued BEFORE the next tile's loads: hipcc waits for those
    // registers with vmcnt(0) at their first use inside the loop, which must not include the prefetch.
    auto scores = [&](const bf16x8 (&q)[2], const bf16x8 (&k)[2], f32x4& s, f32x4& sc, f32x4& s3) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[0], q[0], z4, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[1], q[1], s, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[0], q[0], z4, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[1], q[1], sc, 0, 0, 0);
        if (cls_partial != nullptr) {
            s3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[0], qc[0], z4, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[1], qc[1], s3, 0, 0, 0);
        }
    };
    auto compute = [&](int t, const f32x4& s, const f32x4& sc, const f32x4& s3, const bf16x4 (&vf)[4]) {
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = ((T == 16 || ((4 * g + j) / T) == (c / T)) && (!PAD || (4 * g + j) % T < Tr)) ? s[j] : -INFINITY;
        float m = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        m = fmaxf(m, sc[0]);
        const float mb = m * LOG2E;
        float p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = __builtin_amdgcn_exp2f(x[j] * LOG2E - mb);
        const float pc = __builtin_amdgcn_exp2f(sc[0] * LOG2E - mb);
        float l = (p[0] + p[1]) + (p[2] + p[3]);
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        l += pc;
        const float inv = __builtin_amdgcn_rcpf(l);
        const bf16x8 pf = {(bf16_t)(p[0] * inv), (bf16_t)(p[1] * inv), (bf16_t)(p[2] * inv), (bf16_t)(p[3] * inv),
                           (bf16_t)(g == 0 ? pc * inv : 0.f), 0, 0, 0};
        bf16x8 af[4];
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x4 tr = vf[dt];
            af[dt] = (bf16x8){tr[0], tr[1], tr[2], tr[3], vcls[dt], 0, 0, 0};
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], pf, z4, 0, 0, 0);
        }
        const u32x4 w0 = {pack_bf16(o[0][0], o[0][1]), pack_bf16(o[0][2], o[0][3]), pack_bf16(o[1][0], o[1][1]), pack_bf16(o[1][2], o[1][3])};
        const u32x4 w1 = {pack_bf16(o[2][0], o[2][1]), pack_bf16(o[2][2], o[2][3]), pack_bf16(o[3][0], o[3][1]), pack_bf16(o[3][2], o[3][3])};
        if constexpr (T == 16) {
            // a tile is one patch x 16 frames: the guard is wave-uniform and the 16 rows are 16 separate 128-B lines; lanes c and c ^ 8
            // swap one piece so that each store writes 8 FULL lines (common.h: hh_fullline_swap)
            u32x4 x, y;
            hh_fullline_swap(w0, w1, x, y);
            if (p0 + t * PT < n) {
                const int64_t col = head * 64 + 16 * g + 8 * (c >> 3);
                if (!PAD || (c & 7) < Tr) *(u32x4*)(out + ((int64_t)b * N + tokrow(t, c & 7)) * D + col) = x;
                if (!PAD || 8 + (c & 7) < Tr) *(u32x4*)(out + ((int64_t)b * N + tokrow(t, 8 + (c & 7))) * D + col) = y;
            }
        } else if (p0 + t * PT + c / T < n && (!PAD || c % T < Tr)) {
            bf16_t* op = out + ((int64_t)b * N + tokrow(t, c)) * D + head * 64 + 16 * g;
            *(u32x4*)(op) = w0;
            *(u32x4*)(op + 8) = w1;
        }
        if (cls_partial == nullptr) return;
        // ---- CLS query over this tile's keys (lane (c, g) register j = key row 4g+j, identical for every c)
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = (p0 + t * PT + (4 * g + j) / T < n && (!PAD || (4 * g + j) % T < Tr)) ? s3[j] : -INFINITY;
        float gm = fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3]));
        gm = fmaxf(gm, __shfl_xor(gm, 16, 64));
        gm = fmaxf(gm, __shfl_xor(gm, 32, 64));
        const bool first = pg == 0 && t == 0;            // the CLS key itself is counted once per (clip, head)
        float yc = -INFINITY;
        if (first) {
            f32x4 s33 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[0], qc[0], z4, 0, 0, 0);
            s33 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[1], qc[1], s33, 0, 0, 0);
            yc = s33[0];
            gm = fmaxf(gm, yc);
        }
        const float m_new = fmaxf(mc, gm);
        const float alpha = __builtin_amdgcn_exp2f((mc - m_new) * LOG2E);
        const float mb3 = m_new * LOG2E;
        float p3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) p3[j] = __builtin_amdgcn_exp2f(y[j] * LOG2E - mb3);
        const float p3c = first ? __builtin_amdgcn_exp2f(yc * LOG2E - mb3) : 0.f;
        float ls = (p3[0] + p3[1]) + (p3[2] + p3[3]);
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        lc = lc * alpha + ls + p3c;
        mc = m_new;
        const bf16x8 pf3 = {(bf16_t)p3[0], (bf16_t)p3[1], (bf16_t)p3[2], (bf16_t)p3[3], (bf16_t)(g == 0 ? p3c : 0.f), 0, 0, 0};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) accc[dt][j] *= alpha;
            accc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], pf3, accc[dt], 0, 0, 0);
        }
    };

    bf16x8 qa[2], ka[2], qb[2], kb[2];
    issue(0, qa, ka);
    for (int t = 0; t < ntiles; t += 2) {
        // tile t: V rows in LDS, Q / K rows in registers.  VMEM retires in order and the only younger instructions are the two
        // output stores of tile t-1 (every processed tile has a stored query): do not wait for their write acknowledgements
        bf16x4 vf[4];
        f32x4 s, sc, s3 = z4;
        if (t == 0) HH_WAIT_VMCNT(0);
        else HH_WAIT_VMCNT(2);
        vfrags(t, vf);
        scores(qa, ka, s, sc, s3);
        if (t + 1 < ntiles) issue(t + 1, qb, kb);
        compute(t, s, sc, s3, vf);
        if (t + 1 >= ntiles) break;
        HH_WAIT_VMCNT(2);
        vfrags(t + 1, vf);
        scores(qb, kb, s, sc, s3);
        if (t + 2 < ntiles) issue(t + 2, qa, ka);
        compute(t + 1, s, sc, s3, vf);
    }
    if (cls_partial != nullptr && c == 0) {
        float* rec = cls_partial + (((int64_t)b * heads + head) * groups + pg) * CLS_REC;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(f32x4*)(rec + 4 + 16 * g + 4 * dt) = accc[dt];
        if (g == 0) { rec[0] = mc; rec[1] = lc; }
    }
}

// T = 32 (BASELINE config 4): a patch location spans TWO 16-row tiles (frames 0-15 and 16-31); one wave owns 4 patch locations
// (= 128 tokens = one CLS record, as above).  Scores are four 16x16 blocks per patch (2 key tiles x 2 query tiles); the PV
// product of a query tile contracts over all 32 frame keys in ONE 16x16x32 MFMA (k-slots jj < 4 -> key row 4g+jj of tile 0,
// jj >= 4 -> key row 16+4g+jj-4 of tile 1) and takes the CLS key in a second MFMA whose only non-zero k-slot is slot 0 of g = 0.
// Round 6: PAD = 16 < Tr < 32 real frames: the second key / query tile is ragged (same rules as above).
template <bool PAD>
__global__ __launch_bounds__(256) void time_attn_mfma32_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                               float* __restrict__ cls_partial, int B, int n, int heads, int layout, int Tr) {
    constexpr int T = 32, P = 4;
    __shared__ __attribute__((aligned(16))) char Vsm[4 * 2 * 4096];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = heads * 64;
    const int N = 1 + (PAD ? Tr : T) * n;
    // qkv layout (include/hh.h, hh_qkv_layout): element (row, which, head, d) at row * ld + which * ws + head * hs + d
    const int rev = layout >> 1;                       // (bit 1 of the layout argument, HH_QKV_WALK_REVERSE: the problems last to first)
    layout &= 1;
    const int64_t ld = layout ? 64 : 3 * (int64_t)D;
    const int64_t hs = layout ? (int64_t)B * N * 64 : 64, ws = (int64_t)heads * hs;
    const int groups = (n + P - 1) / P;
    int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    if (wid >= (int64_t)B * heads * groups) return;
    if (rev) wid = (int64_t)B * heads * groups - 1 - wid;
    const int pg = (int)(wid % groups); wid /= groups;
    const int head = (int)(wid % heads);
    const int b = (int)(wid / heads);
    const bf16_t* base = qkv + (int64_t)b * N * ld + head * hs;
    const int p0 = pg * P;
    const int npatch = min(P, n - p0);
    const int c = lane & 15, g = lane >> 4;
    const float LOG2E = 1.4426950408889634f;
    char* vbuf = Vsm + wave * 8192;
    auto tok = [&](int u, int fr) -> int64_t { return 1 + (int64_t)(PAD ? min(fr, Tr - 1) : fr) * n + p0 + u; };
    auto issue = [&](int u, bf16x8 (&qq)[2][2], bf16x8 (&kk)[2][2]) {
        char* dst = vbuf + (u & 1) * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            tglds16(base + tok(u, (lane >> 3) + 8 * i) * ld + 2 * ws + (lane & 7) * 8, dst + i * 1024);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const bf16_t* rp = base + tok(u, 16 * kt + c) * ld + 8 * g;
            qq[kt][0] = *(const bf16x8*)(rp);
            qq[kt][1] = *(const bf16x8*)(rp + 32);
            kk[kt][0] = *(const bf16x8*)(rp + ws);
            kk[kt][1] = *(const bf16x8*)(rp + ws + 32);
        }
    };
    bf16x8 a2[2], qc[2];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kc = *(const bf16x8*)(base + ws + 8 * g + 32 * ks);
        a2[ks] = (c & 3) == 0 ? kc : zero8;
        qc[ks] = *(const bf16x8*)(base + 8 * g + 32 * ks);
    }
    bf16_t vcls[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const bf16_t v = base[2 * ws + 16 * (c >> 2) + 4 * dt + (c & 3)];
        vcls[dt] = g == 0 ? v : (bf16_t)0.f;
    }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    float mc = -INFINITY, lc = 0.f;
    f32x4 accc[4] = {z4, z4, z4, z4};
    const int trq = c >> 2, trp = c & 3;

    auto vfrags = [&](int u, bf16x8 (&vf)[4]) {                      // before the next LDS-DMA issue (see time_attn_mfma_kernel)
        const unsigned vb = lds_u32(vbuf + (u & 1) * 4096 + (4 * g + trq) * 128 + 32 * trp);
        const bf16x4 a0 = tlds_tr4_raw<0>(vb), a1 = tlds_tr4_raw<8>(vb), a2_ = tlds_tr4_raw<16>(vb), a3 = tlds_tr4_raw<24>(vb);
        const bf16x4 b0 = tlds_tr4_raw<2048>(vb), b1 = tlds_tr4_raw<2056>(vb), b2 = tlds_tr4_raw<2064>(vb), b3 = tlds_tr4_raw<2072>(vb);
        tlds_wait();
        vf[0] = (bf16x8){a0[0], a0[1], a0[2], a0[3], b0[0], b0[1], b0[2], b0[3]};
        vf[1] = (bf16x8){a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3]};
        vf[2] = (bf16x8){a2_[0], a2_[1], a2_[2], a2_[3], b2[0], b2[1], b2[2], b2[3]};
        vf[3] = (bf16x8){a3[0], a3[1], a3[2], a3[3], b3[0], b3[1], b3[2], b3[3]};
    };
    // all MFMAs that read the loop-carried Q / K registers, issued before the next patch's loads (see time_attn_mfma_kernel)
    struct Scores { f32x4 s0[2], s1[2], sc[2], y0, y1; };
    auto scores = [&](const bf16x8 (&q)[2][2], const bf16x8 (&k)[2][2], Scores& r) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            r.s0[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[0][0], q[qt][0], z4, 0, 0, 0);
            r.s0[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[0][1], q[qt][1], r.s0[qt], 0, 0, 0);
            r.s1[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[1][0], q[qt][0], z4, 0, 0, 0);
            r.s1[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[1][1], q[qt][1], r.s1[qt], 0, 0, 0);
            r.sc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[0], q[qt][0], z4, 0, 0, 0);
            r.sc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[1], q[qt][1], r.sc[qt], 0, 0, 0);
        }
        r.y0 = z4; r.y1 = z4;
        if (cls_partial != nullptr) {
            r.y0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[0][0], qc[0], z4, 0, 0, 0);
            r.y0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[0][1], qc[1], r.y0, 0, 0, 0);
            r.y1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[1][0], qc[0], z4, 0, 0, 0);
            r.y1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[1][1], qc[1], r.y1, 0, 0, 0);
        }
    };
    auto compute = [&](int u, const Scores& r, const bf16x8 (&vf)[4]) {
        bf16x8 afk[4], afc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            afk[dt] = vf[dt];
            afc[dt] = (bf16x8){vcls[dt], 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const f32x4 s0 = r.s0[qt], sc = r.sc[qt];
            f32x4 s1 = r.s1[qt];
            if (PAD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) s1[j] = 16 + 4 * g + j < Tr ? s1[j] : -INFINITY;     // padded frames of the second key tile
            }
            float m = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            m = fmaxf(m, sc[0]);
            const float mb = m * LOG2E;
            float p[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[j] = __builtin_amdgcn_exp2f(s0[j] * LOG2E - mb);
                p[4 + j] = __builtin_amdgcn_exp2f(s1[j] * LOG2E - mb);
            }
            const float pc = __builtin_amdgcn_exp2f(sc[0] * LOG2E - mb);
            float l = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            l += pc;
            const float inv = __builtin_amdgcn_rcpf(l);
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)(p[j] * inv);
            const bf16x8 pfc = {(bf16_t)(g == 0 ? pc * inv : 0.f), 0, 0, 0, 0, 0, 0, 0};
            f32x4 o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afk[dt], pf, z4, 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afc[dt], pfc, o[dt], 0, 0, 0);
            }
            const u32x4 w0 = {pack_bf16(o[0][0], o[0][1]), pack_bf16(o[0][2], o[0][3]), pack_bf16(o[1][0], o[1][1]), pack_bf16(o[1][2], o[1][3])};
            const u32x4 w1 = {pack_bf16(o[2][0], o[2][1]), pack_bf16(o[2][2], o[2][3]), pack_bf16(o[3][0], o[3][1]), pack_bf16(o[3][2], o[3][3])};
            u32x4 x, y;                                                  // 8 full 128-B lines per store (common.h: hh_fullline_swap)
            hh_fullline_swap(w0, w1, x, y);
            const int64_t col = head * 64 + 16 * g + 8 * (c >> 3);
            if (!PAD || 16 * qt + (c & 7) < Tr) *(u32x4*)(out + ((int64_t)b * N + tok(u, 16 * qt + (c & 7))) * D + col) = x;
            if (!PAD || 16 * qt + 8 + (c & 7) < Tr) *(u32x4*)(out + ((int64_t)b * N + tok(u, 16 * qt + 8 + (c & 7))) * D + col) = y;
        }
        if (cls_partial == nullptr) return;
        const f32x4 y0 = r.y0;
        f32x4 y1 = r.y1;
        if (PAD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y1[j] = 16 + 4 * g + j < Tr ? y1[j] : -INFINITY;
        }
        float gm = fmaxf(fmaxf(fmaxf(y0[0], y0[1]), fmaxf(y0[2], y0[3])), fmaxf(fmaxf(y1[0], y1[1]), fmaxf(y1[2], y1[3])));
        gm = fmaxf(gm, __shfl_xor(gm, 16, 64));
        gm = fmaxf(gm, __shfl_xor(gm, 32, 64));
        const bool first = pg == 0 && u == 0;
        float yc = -INFINITY;
        if (first) {
            f32x4 s33 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[0], qc[0], z4, 0, 0, 0);
            s33 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[1], qc[1], s33, 0, 0, 0);
            yc = s33[0];
            gm = fmaxf(gm, yc);
        }
        const float m_new = fmaxf(mc, gm);
        const float alpha = __builtin_amdgcn_exp2f((mc - m_new) * LOG2E);
        const float mb3 = m_new * LOG2E;
        float p3[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p3[j] = __builtin_amdgcn_exp2f(y0[j] * LOG2E - mb3);
            p3[4 + j] = __builtin_amdgcn_exp2f(y1[j] * LOG2E - mb3);
        }
        const float p3c = first ? __builtin_amdgcn_exp2f(yc * LOG2E - mb3) : 0.f;
        float ls = ((p3[0] + p3[1]) + (p3[2] + p3[3])) + ((p3[4] + p3[5]) + (p3[6] + p3[7]));
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        lc = lc * alpha + ls + p3c;
        mc = m_new;
        bf16x8 pf3;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf3[j] = (bf16_t)p3[j];
        const bf16x8 pf3c = {(bf16_t)(g == 0 ? p3c : 0.f), 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) accc[dt][j] *= alpha;
            accc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afk[dt], pf3, accc[dt], 0, 0, 0);
            if (first) accc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afc[dt], pf3c, accc[dt], 0, 0, 0);
        }
    };

    bf16x8 qa[2][2], ka[2][2], qb[2][2], kb[2][2];
    issue(0, qa, ka);
    for (int u = 0; u < npatch; u += 2) {
        bf16x8 vf[4];
        Scores r;
        if (u == 0) HH_WAIT_VMCNT(0);
        else HH_WAIT_VMCNT(4);           // the 4 output stores of the previous patch may stay in flight
        vfrags(u, vf);
        scores(qa, ka, r);
        if (u + 1 < npatch) issue(u + 1, qb, kb);
        compute(u, r, vf);
        if (u + 1 >= npatch) break;
        HH_WAIT_VMCNT(4);
        vfrags(u + 1, vf);
        scores(qb, kb, r);
        if (u + 2 < npatch) issue(u + 2, qa, ka);
        compute(u + 1, r, vf);
    }
    if (cls_partial != nullptr && c == 0) {
        float* rec = cls_partial + (((int64_t)b * heads + head) * groups + pg) * CLS_REC;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(f32x4*)(rec + 4 + 16 * g + 4 * dt) = accc[dt];
        if (g == 0) { rec[0] = mc; rec[1] = lc; }
    }
}

extern "C" int hh_time_attn_fwd(const void* qkv, int qkv_layout, void* out, float* cls_partial, int B, int T, int n, int heads,
                                hh_stream_t stream) {
    const int walk_rev = qkv_layout & HH_QKV_WALK_REVERSE;          // (bit 1: the kernels take it in the same argument)
    qkv_layout &= ~HH_QKV_WALK_REVERSE;
    HH_REQUIRE(qkv_layout == HH_QKV_TOKEN_MAJOR || qkv_layout == HH_QKV_HEAD_MAJOR, HH_ERR_SHAPE, "hh_time_attn_fwd: bad qkv_layout");
    HH_REQUIRE(B >= 0 && n > 0 && heads > 0, HH_ERR_SHAPE, "hh_time_attn_fwd: bad shape");
    HH_REQUIRE(T >= 1 && T <= 32, HH_ERR_UNSUPPORTED, "hh_time_attn_fwd: num_frames=%d unsupported (1 .. 32)", T);
    HH_REQUIRE(HH_ALIGNED16(qkv) && HH_ALIGNED16(out), HH_ERR_ALIGN, "hh_time_attn_fwd: pointers must be 16-byte aligned");
    if (B == 0) return HH_OK;
    int TP = 1;                                                  // frame slots of a tile: the next power of two (padding rows are masked)
    while (TP < T) TP *= 2;
    const bool pad = TP != T;
    const int P = 128 / TP;
    const int64_t blocks = (int64_t)B * heads * ((n + P - 1) / P);
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* in = (const bf16_t*)qkv;
    bf16_t* o = (bf16_t*)out;
    HHProfScope prof(HH_PROF_TIME_ATTN, 8.0 * B * (1.0 + (double)T * n) * heads * 64, s);
    const unsigned wg = (unsigned)((blocks + 3) / 4);           // one wave per record, four waves per workgroup
    if (TP == 32) {
        hh_prof_note_kernel(HH_PROF_TIME_ATTN, pad ? "time_attn_mfma32_kernel<true>" : "time_attn_mfma32_kernel<false>");
        if (pad) hipLaunchKernelGGL(time_attn_mfma32_kernel<true>, dim3(wg), dim3(256), 0, s, in, o, cls_partial, B, n, heads, qkv_layout | walk_rev, T);
        else hipLaunchKernelGGL(time_attn_mfma32_kernel<false>, dim3(wg), dim3(256), 0, s, in, o, cls_partial, B, n, heads, qkv_layout | walk_rev, T);
        return hh_check_launch("hh_time_attn_fwd(T=32)");
    }
#define LAUNCHM(TT, PD) do { hh_prof_note_kernel(HH_PROF_TIME_ATTN, "time_attn_mfma_kernel<" #TT ", " #PD ">"); \
                         hipLaunchKernelGGL((time_attn_mfma_kernel<TT, PD>), dim3(wg), dim3(256), 0, s, in, o, cls_partial, B, n, heads, qkv_layout | walk_rev, T); } while (0)
    switch (TP) {
        case 1: LAUNCHM(1, false); break;
        case 2: LAUNCHM(2, false); break;
        case 4: if (pad) LAUNCHM(4, true); else LAUNCHM(4, false); break;
        case 8: if (pad) LAUNCHM(8, true); else LAUNCHM(8, false); break;
        default: if (pad) LAUNCHM(16, true); else LAUNCHM(16, false); break;
    }
#undef LAUNCHM
    return hh_check_launch("hh_time_attn_fwd");
}

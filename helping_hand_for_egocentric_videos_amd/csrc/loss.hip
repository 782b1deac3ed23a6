// Loss tail of the training step (run/train.py:126-149,183-192) as a handful of fused kernels instead of ~250 stock elementwise /
// reduction launches: the eps-clamped row normalisation of sim_matrix (model/metric.py:363-375) forward / backward, EgoNCE
// (model/loss.py:8-70: mask build + row / column log-softmax + masked means, loss AND d loss / d sim in one launch), the 582-way
// masked cross-entropy of WordContrastiveLoss (model/loss.py:95-104) and compute_tv_accuracy (model/metric.py:378-392).
// All fp32, latency-bound (<= 1280 x 256 logits): one workgroup or one wave per row; accumulation order differs from torch's, the
// results agree with the oracle to ~1e-6 relative (tests/test_step_gpu.py, tests/test_kernels_gpu.py).
#include "common.h"
#include <math.h>

// ---- y = x / max(||x||_2, eps) per row (metric.py:370-373: a / clamp(a.norm(dim=-1), min=eps)); norm is kept for the backward
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                          float* __restrict__ norm, int rows, int cols, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * ldx;
    float ss = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = xr[c]; ss += v * v; }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss), d = fmaxf(nrm, eps);
    float* yr = y + (int64_t)row * cols;
    for (int c = lane; c < cols; c += 64) yr[c] = xr[c] / d;
    if (lane == 0) norm[row] = nrm;
}

// dx = (dy - y (y . dy)) / ||x||  (||x|| >= eps),  dy / eps otherwise (the clamp is then the constant divisor)
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ norm,
                                                          const float* __restrict__ dy, int64_t lddy, float* __restrict__ dx, int rows,
                                                          int cols, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* yr = y + (int64_t)row * cols;
    const float* gr = dy + (int64_t)row * lddy;
    float* dr = dx + (int64_t)row * cols;
    const float nrm = norm[row];
    if (nrm < eps) {
        for (int c = lane; c < cols; c += 64) dr[c] = gr[c] / eps;
        return;
    }
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) dot += yr[c] * gr[c];
    dot = wave_sum(dot);
    for (int c = lane; c < cols; c += 64) dr[c] = (gr[c] - yr[c] * dot) / nrm;
}

extern "C" int hh_rownorm_fwd(const float* x, int64_t ldx, float* y, float* norm, int rows, int cols, float eps, hh_stream_t stream) {
    HH_REQUIRE(rows >= 0 && cols > 0 && ldx >= cols, HH_ERR_SHAPE, "hh_rownorm_fwd: bad shape");
    if (rows == 0) return HH_OK;
    hipLaunchKernelGGL(rownorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, norm, rows, cols, eps);
    return hh_check_launch("hh_rownorm_fwd");
}

extern "C" int hh_rownorm_bwd(const float* y, const float* norm, const float* dy, int64_t lddy, float* dx, int rows, int cols, float eps,
                              hh_stream_t stream) {
    HH_REQUIRE(rows >= 0 && cols > 0 && lddy >= cols, HH_ERR_SHAPE, "hh_rownorm_bwd: bad shape");
    if (rows == 0) return HH_OK;
    hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, norm, dy, lddy, dx, rows, cols, eps);
    return hh_check_launch("hh_rownorm_bwd");
}

// ---- EgoNCE with a per-row pad flag (model/loss.py:26-70 as run/train.py:144-149 calls it: multi_pad_mask = pad[:, None].repeat).
// x [Rn = R * Bg, Bg] similarities (row i = rephrase i % R of clip c = i / R), sim_v / sim_n [Bg, Bg] (either may be null), pad [Rn]
// (0 = caption absent: the row is dropped, loss.py:42-56).  positives: mask_ij = ((sim_v[c, j] * sim_n[c, j]) + [c == j]) * pad_i > thr.
//   loss = - mean_{kept i} ( sum_j mask_ij logsoftmax_j(x_i. / T)_j / sum_j mask_ij ) - mean_j ( sum_i mask_ij logsoftmax_i(x_.j / T)_i / sum_i mask_ij )
// (the column softmax runs over the kept rows).  Three launches: row statistics (one wave per row), column statistics (16 row groups x
// 64 columns per workgroup, online softmax, merged through LDS), then the loss and d loss / d x element by element -- the backward
// pass of the autograd node is a scalar multiply.
struct EgoMask {
    const float* sim_v; const float* sim_n; int Bg; float thr;
    __device__ __forceinline__ bool operator()(int c, int j, float p) const {
        float e = 0.f;
        if (sim_v && sim_n) e = sim_v[(int64_t)c * Bg + j] * sim_n[(int64_t)c * Bg + j];
        else if (sim_n) e = sim_n[(int64_t)c * Bg + j];
        else if (sim_v) e = sim_v[(int64_t)c * Bg + j];
        return (e + (c == j ? 1.f : 0.f)) * p > thr;
    }
};

__global__ __launch_bounds__(256) void egonce_rows_kernel(const float* __restrict__ x, int64_t ldx, EgoMask positive, const float* __restrict__ pad,
                                                          int R, int Bg, float inv_t, float* __restrict__ row_lse, float* __restrict__ row_cnt,
                                                          float* __restrict__ row_li) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= R * Bg) return;
    const float p = pad[i];
    if (p == 0.f) { if (lane == 0) { row_lse[i] = 0.f; row_cnt[i] = 0.f; row_li[i] = 0.f; } return; }
    const float* xr = x + (int64_t)i * ldx;
    const int c = i / R;
    float m = -INFINITY;
    for (int j = lane; j < Bg; j += 64) m = fmaxf(m, xr[j] * inv_t);
    m = wave_max(m);
    float s = 0.f, cnt = 0.f, ms = 0.f;
    for (int j = lane; j < Bg; j += 64) {
        const float z = xr[j] * inv_t;
        s += expf(z - m);
        if (positive(c, j, p)) { cnt += 1.f; ms += z; }
    }
    s = wave_sum(s); cnt = wave_sum(cnt); ms = wave_sum(ms);
    const float lse = m + logf(s);
    // (cnt >= 1: the diagonal is a positive of every kept row; loss.py clamps likewise)
    if (lane == 0) { row_lse[i] = lse; row_cnt[i] = cnt; row_li[i] = ms / fmaxf(cnt, 1.f) - lse; }
}

#define EGO_RG 16                                   // row groups per workgroup of the column pass
__global__ __launch_bounds__(64 * EGO_RG) void egonce_cols_kernel(const float* __restrict__ x, int64_t ldx, EgoMask positive,
                                                                  const float* __restrict__ pad, int R, int Bg, float inv_t,
                                                                  float* __restrict__ col_lse, float* __restrict__ col_cnt,
                                                                  float* __restrict__ col_lj) {
    __shared__ float sm[4][EGO_RG][64];
    const int jl = threadIdx.x & 63, rg = threadIdx.x >> 6, j = blockIdx.x * 64 + jl, Rn = R * Bg;
    float m = -INFINITY, s = 0.f, cnt = 0.f, ms = 0.f;
    if (j < Bg) {
#pragma unroll 4
        for (int i = rg; i < Rn; i += EGO_RG) {
            const float p = pad[i];
            if (p == 0.f) continue;                 // (wave-uniform: a row index is shared by the 64 lanes of a wave)
            const float z = x[(int64_t)i * ldx + j] * inv_t;
            const float mn = fmaxf(m, z);
            s = s * expf(m - mn) + expf(z - mn);
            m = mn;
            if (positive(i / R, j, p)) { cnt += 1.f; ms += z; }
        }
    }
    sm[0][rg][jl] = m; sm[1][rg][jl] = s; sm[2][rg][jl] = cnt; sm[3][rg][jl] = ms;
    __syncthreads();
    if (rg == 0 && j < Bg) {
        float M = -INFINITY;
        for (int g = 0; g < EGO_RG; ++g) M = fmaxf(M, sm[0][g][jl]);
        float S = 0.f, C = 0.f, MS = 0.f;
        for (int g = 0; g < EGO_RG; ++g) {
            if (sm[1][g][jl] > 0.f) S += sm[1][g][jl] * expf(sm[0][g][jl] - M);
            C += sm[2][g][jl]; MS += sm[3][g][jl];
        }
        const float lse = M + logf(S);
        col_lse[j] = lse;
        col_cnt[j] = C;
        col_lj[j] = MS / C - lse;                   // (no clamp: a clip whose captions are all absent gives NaN, as in loss.py:66)
    }
}

__global__ __launch_bounds__(256) void egonce_grad_kernel(const float* __restrict__ x, int64_t ldx, EgoMask positive, const float* __restrict__ pad,
                                                          int R, int Bg, float inv_t, const float* __restrict__ row_lse,
                                                          const float* __restrict__ row_cnt, const float* __restrict__ row_li,
                                                          const float* __restrict__ col_lse, const float* __restrict__ col_cnt,
                                                          const float* __restrict__ col_lj, float* __restrict__ loss, float* __restrict__ grad) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, Rn = R * Bg;
    float kept = 0.f;
    for (int i = tid; i < Rn; i += 256) kept += pad[i] != 0.f ? 1.f : 0.f;
    kept = wave_sum(kept);
    if (lane == 0) red[wave] = kept;
    __syncthreads();
    const float n_kept = red[0] + red[1] + red[2] + red[3];
    if (blockIdx.x == 0) {                          // the loss itself: - mean_kept li - mean_j lj (fixed summation order)
        float li = 0.f, lj = 0.f;
        for (int i = tid; i < Rn; i += 256) li += row_li[i];
        for (int j = tid; j < Bg; j += 256) lj += col_lj[j];
        li = wave_sum(li); lj = wave_sum(lj);
        if (lane == 0) { red[4 + wave] = li; }
        __syncthreads();
        const float li_tot = red[4] + red[5] + red[6] + red[7];
        __syncthreads();
        if (lane == 0) red[4 + wave] = lj;
        __syncthreads();
        if (tid == 0) loss[0] = -li_tot / n_kept - (red[4] + red[5] + red[6] + red[7]) / (float)Bg;
    }
    const float gi = inv_t / n_kept, gj = inv_t / (float)Bg;
    const int64_t e = (int64_t)blockIdx.x * 256 + tid;
    if (e >= (int64_t)Rn * Bg) return;
    const int i = (int)(e / Bg), j = (int)(e % Bg);
    const float p = pad[i];
    float g = 0.f;
    if (p != 0.f) {
        const float z = x[(int64_t)i * ldx + j] * inv_t;
        const float pos = positive(i / R, j, p) ? 1.f : 0.f;
        g = -(gi * (pos / fmaxf(row_cnt[i], 1.f) - expf(z - row_lse[i])) + gj * (pos / col_cnt[j] - expf(z - col_lse[j])));
    }
    grad[e] = g;
}

extern "C" int64_t hh_workspace_bytes_egonce(int R, int Bg) { return (R <= 0 || Bg <= 0) ? -1 : ((int64_t)3 * R * Bg + 3 * (int64_t)Bg) * 4; }

extern "C" int hh_egonce_fwd(const float* x, int64_t ldx, const float* sim_v, const float* sim_n, const float* pad, int R, int Bg,
                             float temperature, float vn_threshold, float* loss, float* grad, float* scratch, hh_stream_t stream) {
    HH_REQUIRE(R > 0 && Bg > 0 && ldx >= Bg && temperature > 0.f, HH_ERR_SHAPE, "hh_egonce_fwd: bad shape (R=%d, Bg=%d)", R, Bg);
    HH_REQUIRE(x && pad && loss && grad && scratch, HH_ERR_SHAPE, "hh_egonce_fwd: null pointer");
    const int Rn = R * Bg;
    float *row_lse = scratch, *row_cnt = scratch + Rn, *row_li = scratch + 2 * (int64_t)Rn;
    float *col_lse = scratch + 3 * (int64_t)Rn, *col_cnt = col_lse + Bg, *col_lj = col_cnt + Bg;
    const EgoMask mask{sim_v, sim_n, Bg, vn_threshold};
    const float inv_t = 1.f / temperature;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(egonce_rows_kernel, dim3((unsigned)((Rn + 3) / 4)), dim3(256), 0, s, x, ldx, mask, pad, R, Bg, inv_t, row_lse, row_cnt, row_li);
    hipLaunchKernelGGL(egonce_cols_kernel, dim3((unsigned)((Bg + 63) / 64)), dim3(64 * EGO_RG), 0, s, x, ldx, mask, pad, R, Bg, inv_t, col_lse, col_cnt, col_lj);
    hipLaunchKernelGGL(egonce_grad_kernel, dim3((unsigned)(((int64_t)Rn * Bg + 255) / 256)), dim3(256), 0, s, x, ldx, mask, pad, R, Bg, inv_t, row_lse,
                       row_cnt, row_li, col_lse, col_cnt, col_lj, loss, grad);
    return hh_check_launch("hh_egonce_fwd");
}

// ---- masked cross-entropy of the word loss (loss.py:95-104): row r has ground-truth noun gt[r]; logits z_k = sim[r, k] / T, except
// nouns k whose similarity to the ground truth exceeds the threshold (noun_sim[gt[r], k] > thr, k != gt[r]) -- their logit is the
// constant -1 / T (masked_fill(noun_mask, -1)).  ce[r] = logsumexp(z) - z_gt for valid rows, 0 otherwise; grad[r, k] = d ce[r] / d sim[r, k]
// (0 on the masked columns and on invalid rows).  One wave per row.
__global__ __launch_bounds__(256) void masked_ce_kernel(const float* __restrict__ sim, int64_t lds_, const float* __restrict__ noun_sim,
                                                        const int64_t* __restrict__ gt, const unsigned char* __restrict__ valid, int rows,
                                                        int V, float inv_t, float thr, float* __restrict__ ce, float* __restrict__ grad) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* gr = grad + (int64_t)row * V;
    if (!valid[row]) {
        for (int k = lane; k < V; k += 64) gr[k] = 0.f;
        if (lane == 0) ce[row] = 0.f;
        return;
    }
    const int64_t g = gt[row];
    if (g < 0 || g >= V) {                                  // F.cross_entropy raises on such a target; no host round trip here: the row turns NaN
        for (int k = lane; k < V; k += 64) gr[k] = __builtin_nanf("");
        if (lane == 0) ce[row] = __builtin_nanf("");
        return;
    }
    const float* sr = sim + (int64_t)row * lds_;
    const float* nr = noun_sim + g * V;
    float m = -INFINITY;
    for (int k = lane; k < V; k += 64) {
        const bool masked = k != g && nr[k] > thr;
        m = fmaxf(m, (masked ? -1.f : sr[k]) * inv_t);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int k = lane; k < V; k += 64) {
        const bool masked = k != g && nr[k] > thr;
        s += expf((masked ? -1.f : sr[k]) * inv_t - m);
    }
    s = wave_sum(s);
    const float lse = m + logf(s);
    for (int k = lane; k < V; k += 64) {
        const bool masked = k != g && nr[k] > thr;
        const float pk = expf((masked ? -1.f : sr[k]) * inv_t - lse);
        gr[k] = masked ? 0.f : (pk - (k == g ? 1.f : 0.f)) * inv_t;
    }
    if (lane == 0) ce[row] = lse - sr[g] * inv_t;
}

extern "C" int hh_masked_ce_fwd(const float* sim, int64_t ld, const float* noun_sim, const int64_t* gt, const unsigned char* valid, int rows,
                                int V, float temperature, float threshold, float* ce, float* grad, hh_stream_t stream) {
    HH_REQUIRE(rows >= 0 && V > 0 && ld >= V && temperature > 0.f, HH_ERR_SHAPE, "hh_masked_ce_fwd: bad shape");
    if (rows == 0) return HH_OK;
    hipLaunchKernelGGL(masked_ce_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, sim, ld, noun_sim, gt, valid, rows,
                       V, 1.f / temperature, threshold, ce, grad);
    return hh_check_launch("hh_masked_ce_fwd");
}

// ---- per-caption token statistics of the step (run/train.py:124,144): eot[r] = argmax_l text[r, l] (the EOT token has the largest id of the
// vocabulary; first index on ties, as torch.argmax) and pad[r] = (#(text[r, :] != 0) != 2) as fp32 (0 = an empty rephrase slot [SOT, EOT]).
// One wave per row; replaces ne / sum / ne / cast + float / argmax (7 stock launches, two of them int64 reductions).
__global__ __launch_bounds__(256) void text_flags_kernel(const int64_t* __restrict__ text, int rows, int L, int64_t* __restrict__ eot, float* __restrict__ pad) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int64_t* t = text + (int64_t)row * L;
    int64_t best = INT64_MIN;
    int bi = 0x7fffffff, nz = 0;
    for (int l = lane; l < L; l += 64) {
        const int64_t v = t[l];
        nz += v != 0;
        if (v > best) { best = v; bi = l; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int64_t ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        nz += __shfl_xor(nz, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { eot[row] = bi; pad[row] = nz != 2 ? 1.f : 0.f; }
}

extern "C" int hh_text_flags(const int64_t* text, int rows, int L, int64_t* eot, float* pad, hh_stream_t stream) {
    HH_REQUIRE(rows >= 0 && L > 0, HH_ERR_SHAPE, "hh_text_flags: bad arguments");
    if (rows == 0) return HH_OK;                      // (an empty tensor's data pointer is null)
    HH_REQUIRE(text && eot && pad, HH_ERR_SHAPE, "hh_text_flags: null pointer");
    hipLaunchKernelGGL(text_flags_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, text, rows, L, eot, pad);
    return hh_check_launch("hh_text_flags");
}

// ---- compute_tv_accuracy (metric.py:378-392): similarity [Bg, Bg] (row stride lds_: the first rephrase of every clip), positives
// pos_ij = (sim_v_ij * sim_n_ij + [i == j] + [i != j and text_cos_ij > 0.99]) > 0; out[0] = mean_j pos[argmax_i sim_ij, j] (video ->
// text), out[1] = mean_i pos[i, argmax_j sim_ij] (text -> video); first index on ties, as torch.argmax.  One workgroup.
__global__ __launch_bounds__(256) void tv_accuracy_kernel(const float* __restrict__ sim, int64_t lds_, const float* __restrict__ text_cos,
                                                          const float* __restrict__ sim_v, const float* __restrict__ sim_n, int Bg,
                                                          float* __restrict__ out) {
    __shared__ float red[2 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto pos = [&](int i, int j) -> float {
        const float same = (i != j && text_cos[(int64_t)i * Bg + j] > 0.99f) ? 1.f : 0.f;
        return ((sim_v[(int64_t)i * Bg + j] * sim_n[(int64_t)i * Bg + j] + (i == j ? 1.f : 0.f)) + same) > 0.f ? 1.f : 0.f;
    };
    float vt = 0.f, tv = 0.f;
    for (int k = tid; k < Bg; k += 256) {
        int bi = 0, bj = 0;
        float mi = -INFINITY, mj = -INFINITY;
        for (int t = 0; t < Bg; ++t) {
            const float a = sim[(int64_t)t * lds_ + k];          // column k: argmax over rows
            if (a > mi) { mi = a; bi = t; }
            const float b = sim[(int64_t)k * lds_ + t];          // row k: argmax over columns
            if (b > mj) { mj = b; bj = t; }
        }
        vt += pos(bi, k);
        tv += pos(k, bj);
    }
    vt = wave_sum(vt); tv = wave_sum(tv);
    if (lane == 0) { red[wave] = vt; red[4 + wave] = tv; }
    __syncthreads();
    if (tid == 0) {
        out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)Bg;
        out[1] = (red[4] + red[5] + red[6] + red[7]) / (float)Bg;
    }
}

extern "C" int hh_tv_accuracy(const float* sim, int64_t ld, const float* text_cos, const float* sim_v, const float* sim_n, int Bg, float* out,
                              hh_stream_t stream) {
    HH_REQUIRE(Bg > 0 && ld >= Bg && sim && text_cos && sim_v && sim_n && out, HH_ERR_SHAPE, "hh_tv_accuracy: bad arguments");
    hipLaunchKernelGGL(tv_accuracy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sim, ld, text_cos, sim_v, sim_n, Bg, out);
    return hh_check_launch("hh_tv_accuracy");
}

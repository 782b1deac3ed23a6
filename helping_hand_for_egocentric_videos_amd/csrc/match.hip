// Hungarian matching and matched-pair box losses on device (model/box_utils.py:43-92,156-173,249-279;
// utils/box_ops.py:9-61; scipy.optimize.linear_sum_assignment restated as in oracle/lsap.c).
// Removes the reference's three device->host syncs per step.  Integer/index work must be BIT-EXACT with the
// reference given identical fp32 boxes, so the fp32 cost is evaluated in the reference's operation order with
// FMA contraction disabled, and the assignment runs the same shortest-augmenting-path algorithm in fp64.
// One thread per frame (problems are <= 16 x 8); latency-bound, negligible bytes.
#include "common.h"
#pragma clang fp contract(off)

#define MAXQ 16
#define MAXK 16

struct Box { float x0, y0, x1, y1; };

__device__ __forceinline__ Box to_xyxy(const float* c) {
    Box b;
    b.x0 = c[0] - 0.5f * c[2];
    b.y0 = c[1] - 0.5f * c[3];
    b.x1 = c[0] + 0.5f * c[2];
    b.y1 = c[1] + 0.5f * c[3];
    return b;
}

// utils/box_ops.py:24-61 (IoU with union+1e-4; GIoU hull term without eps)
__device__ __forceinline__ float giou_xyxy(const Box& a, const Box& b) {
    const float area_a = (a.x1 - a.x0) * (a.y1 - a.y0);
    const float area_b = (b.x1 - b.x0) * (b.y1 - b.y0);
    const float w = fmaxf(fminf(a.x1, b.x1) - fmaxf(a.x0, b.x0), 0.f);
    const float h = fmaxf(fminf(a.y1, b.y1) - fmaxf(a.y0, b.y0), 0.f);
    const float inter = w * h;
    const float uni = (area_a + area_b) - inter;
    const float iou = inter / (uni + 0.0001f);
    const float hw = fmaxf(fmaxf(a.x1, b.x1) - fminf(a.x0, b.x0), 0.f);
    const float hh = fmaxf(fmaxf(a.y1, b.y1) - fminf(a.y0, b.y0), 0.f);
    const float hull = hw * hh;
    return iou - (hull - uni) / hull;
}

// Rectangular LSAP, nr <= nc, shortest augmenting path (Crouse 2016) as scipy implements it.
// cost row-major [nr][ldc] (double).  col4row out.  Returns false if infeasible.
__device__ bool lsap_wide(int nr, int nc, const double* cost, int ldc, int* col4row) {
    double u[MAXK], v[MAXQ], spc[MAXQ];
    int path[MAXQ], row4col[MAXQ], remaining[MAXQ];
    bool SR[MAXK], SC[MAXQ];
    for (int i = 0; i < nr; ++i) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = 0; j < nc; ++j) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
    for (int cur = 0; cur < nr; ++cur) {
        double min_val = 0.0;
        int i = cur, num_remaining = nc, sink = -1;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        for (int t = 0; t < nr; ++t) SR[t] = false;
        for (int t = 0; t < nc; ++t) { SC[t] = false; spc[t] = INFINITY; }
        while (sink == -1) {
            int index = -1;
            double lowest = INFINITY;
            SR[i] = true;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = min_val + cost[i * ldc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            min_val = lowest;
            if (min_val == INFINITY) return false;
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = true;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += min_val;
        for (int t = 0; t < nr; ++t) if (SR[t] && t != cur) u[t] += min_val - spc[col4row[t]];
        for (int t = 0; t < nc; ++t) if (SC[t]) v[t] -= min_val - spc[t];
        int j = sink;
        for (;;) {
            const int ii = path[j];
            row4col[j] = ii;
            const int t = col4row[ii]; col4row[ii] = j; j = t;
            if (ii == cur) break;
        }
    }
    return true;
}

__global__ __launch_bounds__(64) void match_boxes_kernel(const float* __restrict__ pred, int Qtot, int q0, int q,
                                                         const float* __restrict__ raw, const int* __restrict__ given_count,
                                                         int k, float img, float w_l1,
                                                         float w_giou, const float* __restrict__ class_cost, float w_class,
                                                         float* __restrict__ tgt, int* __restrict__ tgt_count,
                                                         int64_t* __restrict__ mp, int64_t* __restrict__ mt,
                                                         int* __restrict__ mn, int64_t F) {
    const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (f >= F) return;
    // ---- prepare_targets (box_utils.py:249-279, center_crop=False): clip to [0,img]/img, keep x1>x0 && y1>y0, -> cxcywh
    float t[MAXK][4];
    int cnt = 0;
    if (given_count) {
        cnt = given_count[f];
        if (cnt > k) cnt = k;
        for (int i = 0; i < cnt; ++i)
            for (int c = 0; c < 4; ++c) t[i][c] = raw[(f * k + i) * 4 + c];
    } else
    for (int i = 0; i < k; ++i) {
        const float* r = raw + (f * k + i) * 4;
        const float x0 = fminf(fmaxf(r[0], 0.f), img) / img, y0 = fminf(fmaxf(r[1], 0.f), img) / img;
        const float x1 = fminf(fmaxf(r[2], 0.f), img) / img, y1 = fminf(fmaxf(r[3], 0.f), img) / img;
        if (x1 > x0 && y1 > y0) {
            t[cnt][0] = (x0 + x1) / 2.f; t[cnt][1] = (y0 + y1) / 2.f; t[cnt][2] = x1 - x0; t[cnt][3] = y1 - y0;
            ++cnt;
        }
    }
    for (int i = 0; i < k; ++i)
        for (int c = 0; c < 4; ++c) tgt[(f * k + i) * 4 + c] = (i < cnt) ? t[i][c] : 0.f;
    tgt_count[f] = cnt;
    const int nm = cnt < q ? cnt : q;
    mn[f] = nm;
    for (int i = 0; i < k; ++i) { mp[f * k + i] = -1; mt[f * k + i] = -1; }
    if (nm == 0) return;
    // ---- cost C = w_l1 * cdist_L1 + w_giou * (-GIoU)   (box_utils.py:74-81), fp32, reference op order
    double cost[MAXQ * MAXK];   // stored [pred][tgt] or transposed [tgt][pred] so that rows <= cols
    const bool transpose = cnt < q;            // scipy: transpose when nc < nr (rows = preds)
    for (int i = 0; i < q; ++i) {
        const float* p = pred + (f * Qtot + q0 + i) * 4;
        const Box pb = to_xyxy(p);
        for (int j = 0; j < cnt; ++j) {
            float l1 = 0.f;
            for (int c = 0; c < 4; ++c) l1 = l1 + fabsf(p[c] - t[j][c]);
            const Box tb = to_xyxy(t[j]);
            const float g = giou_xyxy(pb, tb);
            float cst = w_l1 * l1 + w_giou * (-g);
            // exclude_class=False (box_utils.py:83-85): C += cost_class * (-softmax(logits)[query, label of target j]); the
            // caller passes that gathered term, indexed by the target's slot in the frame's kept-target list
            if (class_cost) cst = cst + w_class * class_cost[((f * q) + i) * k + j];
            if (transpose) cost[j * q + i] = (double)cst; else cost[i * cnt + j] = (double)cst;
        }
    }
    int col4row[MAXK];
    if (transpose) {
        // rows = targets (cnt), cols = preds (q); result sorted by pred index
        lsap_wide(cnt, q, cost, q, col4row);
        int order[MAXK];
        for (int i = 0; i < cnt; ++i) order[i] = i;
        for (int i = 1; i < cnt; ++i) {           // insertion sort by pred index (col4row)
            const int key = order[i];
            int j = i - 1;
            while (j >= 0 && col4row[order[j]] > col4row[key]) { order[j + 1] = order[j]; --j; }
            order[j + 1] = key;
        }
        for (int i = 0; i < cnt; ++i) { mp[f * k + i] = col4row[order[i]]; mt[f * k + i] = order[i]; }
    } else {
        lsap_wide(q, cnt, cost, cnt, col4row);
        for (int i = 0; i < q; ++i) { mp[f * k + i] = i; mt[f * k + i] = col4row[i]; }
    }
}

// generic batched LSAP (word loss, loss.py:83-93): rows with row_valid != 0 (in order) x all nc columns
__global__ __launch_bounds__(64) void lsap_rows_kernel(const float* __restrict__ cost, const unsigned char* __restrict__ row_valid,
                                                       int64_t* __restrict__ col_of_row, int64_t P, int nr, int nc) {
    const int64_t p = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= P) return;
    int rows[MAXK], cnt = 0;
    for (int r = 0; r < nr; ++r) { col_of_row[p * nr + r] = -1; if (row_valid[p * nr + r]) rows[cnt++] = r; }
    if (cnt == 0) return;
    double c[MAXQ * MAXK];
    int col4row[MAXQ];
    if (cnt <= nc) {
        for (int i = 0; i < cnt; ++i) for (int j = 0; j < nc; ++j) c[i * nc + j] = (double)cost[(p * nr + rows[i]) * nc + j];
        lsap_wide(cnt, nc, c, nc, col4row);
        for (int i = 0; i < cnt; ++i) col_of_row[p * nr + rows[i]] = col4row[i];
    } else {
        for (int i = 0; i < cnt; ++i) for (int j = 0; j < nc; ++j) c[j * cnt + i] = (double)cost[(p * nr + rows[i]) * nc + j];
        lsap_wide(nc, cnt, c, cnt, col4row);        // col4row[j] = row index assigned to column j
        for (int j = 0; j < nc; ++j) col_of_row[p * nr + rows[col4row[j]]] = j;
    }
}

// matched-pair losses (box_utils.py:156-173): sums[0] += sum |p - t|, sums[1] += sum (1 - GIoU(p,t))
__global__ __launch_bounds__(64) void box_loss_fwd_kernel(const float* __restrict__ pred, int Qtot, int q0,
                                                          const float* __restrict__ tgt, int k,
                                                          const int64_t* __restrict__ mp, const int64_t* __restrict__ mt,
                                                          const int* __restrict__ mn, float* __restrict__ sums, int64_t F) {
    const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
    float a = 0.f, g = 0.f;
    if (f < F) {
        const int nm = mn[f];
        for (int i = 0; i < nm; ++i) {
            const float* p = pred + (f * Qtot + q0 + mp[f * k + i]) * 4;
            const float* t = tgt + (f * k + mt[f * k + i]) * 4;
            for (int c = 0; c < 4; ++c) a += fabsf(p[c] - t[c]);
            g += 1.f - giou_xyxy(to_xyxy(p), to_xyxy(t));
        }
    }
    a = wave_sum(a);
    g = wave_sum(g);
    if (threadIdx.x == 0) { atomicAdd(sums, a); atomicAdd(sums + 1, g); }
}

// d(pred) for L = g_l1 * sum|p-t| + g_giou * sum(1-GIoU); dpred [F,Qtot,4] must be zero-initialised by the caller
__global__ __launch_bounds__(64) void box_loss_bwd_kernel(const float* __restrict__ pred, int Qtot, int q0,
                                                          const float* __restrict__ tgt, int k,
                                                          const int64_t* __restrict__ mp, const int64_t* __restrict__ mt,
                                                          const int* __restrict__ mn, const float* __restrict__ g_l1,
                                                          const float* __restrict__ g_giou, float* __restrict__ dpred, int64_t F,
                                                          const float* __restrict__ c_l1, const float* __restrict__ c_giou) {
    const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (f >= F) return;
    const float gl = *g_l1 * (c_l1 ? *c_l1 : 1.f), gg = *g_giou * (c_giou ? *c_giou : 1.f);
    const int nm = mn[f];
    for (int i = 0; i < nm; ++i) {
        const int64_t pi = (f * Qtot + q0 + mp[f * k + i]) * 4;
        const float* p = pred + pi;
        const float* t = tgt + (f * k + mt[f * k + i]) * 4;
        const Box a = to_xyxy(p), b = to_xyxy(t);
        const float Ap = (a.x1 - a.x0) * (a.y1 - a.y0), At = (b.x1 - b.x0) * (b.y1 - b.y0);
        const float dxw = fminf(a.x1, b.x1) - fmaxf(a.x0, b.x0), dyh = fminf(a.y1, b.y1) - fmaxf(a.y0, b.y0);
        const float iw = fmaxf(dxw, 0.f), ih = fmaxf(dyh, 0.f), inter = iw * ih;
        const float U = (Ap + At) - inter, Ue = U + 0.0001f;
        const float hw = fmaxf(fmaxf(a.x1, b.x1) - fminf(a.x0, b.x0), 0.f), hh = fmaxf(fmaxf(a.y1, b.y1) - fminf(a.y0, b.y0), 0.f);
        const float hull = hw * hh;
        const float dg_dinter = 1.f / Ue + inter / (Ue * Ue) - 1.f / hull;
        const float dg_dAp = -inter / (Ue * Ue) + 1.f / hull;
        const float dg_dhull = -U / (hull * hull);
        const float s = -gg;
        float dx0 = s * dg_dAp * (-(a.y1 - a.y0)), dx1 = s * dg_dAp * (a.y1 - a.y0);
        float dy0 = s * dg_dAp * (-(a.x1 - a.x0)), dy1 = s * dg_dAp * (a.x1 - a.x0);
        const float d_iw = dxw > 0.f ? s * dg_dinter * ih : 0.f;
        const float d_ih = dyh > 0.f ? s * dg_dinter * iw : 0.f;
        if (a.x0 > b.x0) dx0 -= d_iw; else if (a.x0 == b.x0) dx0 -= 0.5f * d_iw;
        if (a.x1 < b.x1) dx1 += d_iw; else if (a.x1 == b.x1) dx1 += 0.5f * d_iw;
        if (a.y0 > b.y0) dy0 -= d_ih; else if (a.y0 == b.y0) dy0 -= 0.5f * d_ih;
        if (a.y1 < b.y1) dy1 += d_ih; else if (a.y1 == b.y1) dy1 += 0.5f * d_ih;
        const float d_hw = s * dg_dhull * hh, d_hh = s * dg_dhull * hw;
        if (a.x0 < b.x0) dx0 -= d_hw; else if (a.x0 == b.x0) dx0 -= 0.5f * d_hw;
        if (a.x1 > b.x1) dx1 += d_hw; else if (a.x1 == b.x1) dx1 += 0.5f * d_hw;
        if (a.y0 < b.y0) dy0 -= d_hh; else if (a.y0 == b.y0) dy0 -= 0.5f * d_hh;
        if (a.y1 > b.y1) dy1 += d_hh; else if (a.y1 == b.y1) dy1 += 0.5f * d_hh;
        float sg[4];
        for (int c = 0; c < 4; ++c) { const float d = p[c] - t[c]; sg[c] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
        dpred[pi + 0] += dx0 + dx1 + gl * sg[0];
        dpred[pi + 1] += dy0 + dy1 + gl * sg[1];
        dpred[pi + 2] += 0.5f * (dx1 - dx0) + gl * sg[2];
        dpred[pi + 3] += 0.5f * (dy1 - dy0) + gl * sg[3];
    }
}

extern "C" int hh_match_boxes(const float* pred, int Qtot, int q0, int q, const float* raw_boxes, const int32_t* given_count, int k, float img,
                              float w_l1, float w_giou, const float* class_cost, float w_class, float* tgt_cxcywh, int32_t* tgt_count, int64_t* match_pred,
                              int64_t* match_tgt, int32_t* match_n, int64_t F, hh_stream_t stream) {
    HH_REQUIRE(F >= 0 && q > 0 && q <= MAXQ && k > 0 && k <= MAXK && q0 >= 0 && q0 + q <= Qtot, HH_ERR_SHAPE,
               "hh_match_boxes: need 0 < q <= %d, 0 < k <= %d, q0+q <= Qtot (q=%d k=%d q0=%d Qtot=%d)", MAXQ, MAXK, q, k, q0, Qtot);
    if (F == 0) return HH_OK;
    hipLaunchKernelGGL(match_boxes_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, (hipStream_t)stream, pred, Qtot, q0, q,
                       raw_boxes, given_count, k, img, w_l1, w_giou, class_cost, w_class, tgt_cxcywh, tgt_count, match_pred, match_tgt, match_n, F);
    return hh_check_launch("hh_match_boxes");
}

extern "C" int hh_lsap_rows(const float* cost, const uint8_t* row_valid, int64_t* col_of_row, int64_t P, int nr, int nc,
                            hh_stream_t stream) {
    HH_REQUIRE(P >= 0 && nr > 0 && nr <= MAXK && nc > 0 && nc <= MAXQ, HH_ERR_SHAPE, "hh_lsap_rows: need nr <= %d, nc <= %d", MAXK, MAXQ);
    if (P == 0) return HH_OK;
    hipLaunchKernelGGL(lsap_rows_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, (hipStream_t)stream, cost, row_valid, col_of_row, P, nr, nc);
    return hh_check_launch("hh_lsap_rows");
}

extern "C" int hh_box_loss_fwd(const float* pred, int Qtot, int q0, const float* tgt_cxcywh, int k, const int64_t* match_pred,
                               const int64_t* match_tgt, const int32_t* match_n, float* sums, int64_t F, hh_stream_t stream) {
    HH_REQUIRE(F >= 0 && k > 0 && k <= MAXK, HH_ERR_SHAPE, "hh_box_loss_fwd: bad shape");
    if (F == 0) return HH_OK;
    hipLaunchKernelGGL(box_loss_fwd_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, (hipStream_t)stream, pred, Qtot, q0,
                       tgt_cxcywh, k, match_pred, match_tgt, match_n, sums, F);
    return hh_check_launch("hh_box_loss_fwd");
}

extern "C" int hh_box_loss_bwd(const float* pred, int Qtot, int q0, const float* tgt_cxcywh, int k, const int64_t* match_pred,
                               const int64_t* match_tgt, const int32_t* match_n, const float* g_l1, const float* g_giou,
                               float* dpred, int64_t F, hh_stream_t stream) {
    HH_REQUIRE(F >= 0 && k > 0 && k <= MAXK, HH_ERR_SHAPE, "hh_box_loss_bwd: bad shape");
    if (F == 0) return HH_OK;
    hipLaunchKernelGGL(box_loss_bwd_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, (hipStream_t)stream, pred, Qtot, q0,
                       tgt_cxcywh, k, match_pred, match_tgt, match_n, g_l1, g_giou, dpred, F, nullptr, nullptr);
    return hh_check_launch("hh_box_loss_bwd");
}

// ---- the scalar tail of the step's two box losses (box_utils.py:156-173,142-154,445-461; run/train.py:161-183) in ONE launch instead of
// ~35 scalar / [F]-sized stock ops: per box type t in {hand, object}
//   loss_bbox_t = sums_t[0] / nb_t, loss_giou_t = sums_t[1] / nb_t, total_t = (w_l1 * loss_bbox_t + w_giou * loss_giou_t) / denom,
//   cardinality_error_t = mean_f | #(argmax[f, q0_t .. q0_t + q_t) != no_object) - count_t[f] |      (argmax may be NULL: skipped, 0)
// out[8] = total_h, total_o, loss_bbox_h, loss_giou_h, loss_bbox_o, loss_giou_o, card_h, card_o; coef[4] = d total_h / d sums_h[0], d total_h
// / d sums_h[1], d total_o / d sums_o[0], d total_o / d sums_o[1] (the backward's constants).  One workgroup.
__global__ __launch_bounds__(256) void box_tail_kernel(const float* __restrict__ sums_h, const float* __restrict__ sums_o, const float* __restrict__ nb,
                                                       const int* __restrict__ count_h, const int* __restrict__ count_o,
                                                       const int64_t* __restrict__ argmax, int Q, int q0_h, int q_h, int q0_o, int q_o, int64_t no_object,
                                                       int64_t F, float w_l1, float w_giou, float denom, float* __restrict__ out, float* __restrict__ coef) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float ch = 0.f, co = 0.f;
    if (argmax != nullptr) {
        for (int64_t f = tid; f < F; f += 256) {
            int nh = 0, no = 0;
            for (int j = 0; j < q_h; ++j) nh += argmax[f * Q + q0_h + j] != no_object;
            for (int j = 0; j < q_o; ++j) no += argmax[f * Q + q0_o + j] != no_object;
            ch += fabsf((float)nh - (float)count_h[f]);
            co += fabsf((float)no - (float)count_o[f]);
        }
    }
    ch = wave_sum(ch); co = wave_sum(co);
    if (lane == 0) { red[0][wave] = ch; red[1][wave] = co; }
    __syncthreads();
    if (tid == 0) {
        const float nh = nb[0], no = nb[1];
        const float lbh = sums_h[0] / nh, lgh = sums_h[1] / nh, lbo = sums_o[0] / no, lgo = sums_o[1] / no;
        out[0] = (w_l1 * lbh + w_giou * lgh) / denom;
        out[1] = (w_l1 * lbo + w_giou * lgo) / denom;
        out[2] = lbh; out[3] = lgh; out[4] = lbo; out[5] = lgo;
        out[6] = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)F;
        out[7] = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)F;
        coef[0] = w_l1 / nh / denom; coef[1] = w_giou / nh / denom; coef[2] = w_l1 / no / denom; coef[3] = w_giou / no / denom;
    }
}

extern "C" int hh_box_tail_fwd(const float* sums_h, const float* sums_o, const float* num_boxes, const int32_t* count_h, const int32_t* count_o,
                               const int64_t* argmax, int Q, int q0_h, int q_h, int q0_o, int q_o, int64_t no_object, int64_t F, float w_l1, float w_giou,
                               float denom, float* out, float* coef, hh_stream_t stream) {
    HH_REQUIRE(sums_h && sums_o && num_boxes && count_h && count_o && out && coef && F > 0 && denom > 0.f, HH_ERR_SHAPE, "hh_box_tail_fwd: bad arguments");
    HH_REQUIRE(argmax == nullptr || (Q > 0 && q0_h >= 0 && q0_h + q_h <= Q && q0_o >= 0 && q0_o + q_o <= Q), HH_ERR_SHAPE, "hh_box_tail_fwd: query slices outside Q");
    hipLaunchKernelGGL(box_tail_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums_h, sums_o, num_boxes, count_h, count_o, argmax, Q, q0_h, q_h, q0_o,
                       q_o, no_object, F, w_l1, w_giou, denom, out, coef);
    return hh_check_launch("hh_box_tail_fwd");
}

// hh_box_loss_bwd with the upstream gradient given as g[0] * coef_l1[0] / g[0] * coef_giou[0] (the constants hh_box_tail_fwd left on the
// device): dpred is ADDED to, so the two box types -- disjoint query slices -- share one zero-initialised buffer
extern "C" int hh_box_loss_bwd_scaled(const float* pred, int Qtot, int q0, const float* tgt_cxcywh, int k, const int64_t* match_pred,
                                      const int64_t* match_tgt, const int32_t* match_n, const float* g, const float* coef_l1, const float* coef_giou,
                                      float* dpred, int64_t F, hh_stream_t stream) {
    HH_REQUIRE(F >= 0 && k > 0 && k <= MAXK && g && coef_l1 && coef_giou, HH_ERR_SHAPE, "hh_box_loss_bwd_scaled: bad arguments");
    if (F == 0) return HH_OK;
    hipLaunchKernelGGL(box_loss_bwd_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, (hipStream_t)stream, pred, Qtot, q0,
                       tgt_cxcywh, k, match_pred, match_tgt, match_n, g, g, dpred, F, coef_l1, coef_giou);
    return hh_check_launch("hh_box_loss_bwd_scaled");
}

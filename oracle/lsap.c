/* Oracle (TEST INFRASTRUCTURE): rectangular linear sum assignment, plain C restatement of the
 * algorithm behind scipy.optimize.linear_sum_assignment (scipy 1.15.3 in this image; the reference
 * calls it at /root/reference/model/box_utils.py:91 and /root/reference/model/loss.py:92).
 *
 * Published algorithm: D. F. Crouse, "On implementing 2D rectangular assignment algorithms",
 * IEEE T-AES 52(4), 2016 -- shortest augmenting path with dual variables u, v; rows are augmented in
 * order; among equal reduced costs an unassigned column is preferred; the column scan order is the
 * `remaining` list, initialised descending and compacted by swap-with-last.  Costs are double.
 * Wide-or-square problems (nr <= nc) are solved directly; tall ones are solved transposed and the
 * result is re-sorted by row.  Output: rows ascending, len = min(nr, nc).  Checked against scipy's
 * own answers on 400 problems (random, integer-tie-heavy, wide, empty and degenerate matrices) stored in
 * tests/golden/lsap_scipy.npz: tests/test_oracle_golden.py::test_lsap_known_answers.
 *
 * Returns the number of assigned pairs, or -1 when infeasible / on bad input.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int augment_from(int nr, int nc, const double *cost, const double *u, const double *v,
                        int *path, const int *row4col, double *spc, int i, unsigned char *SR,
                        unsigned char *SC, int *remaining, double *p_min)
{
    double min_val = 0.0;
    int num_remaining = nc;
    for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    memset(SR, 0, (size_t)nr);
    memset(SC, 0, (size_t)nc);
    for (int j = 0; j < nc; ++j) spc[j] = INFINITY;
    int sink = -1;
    while (sink == -1) {
        int index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (int it = 0; it < num_remaining; ++it) {
            int j = remaining[it];
            double r = min_val + cost[(size_t)i * nc + j] - u[i] - v[j];
            if (r < spc[j]) { path[j] = i; spc[j] = r; }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1;
        int j = remaining[index];
        if (row4col[j] == -1) sink = j; else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

static int cmp_pair(const void *a, const void *b)
{
    const long long *x = (const long long *)a, *y = (const long long *)b;
    return (x[0] > y[0]) - (x[0] < y[0]);
}

int hh_oracle_lsap(int nr, int nc, const double *cost_in, long long *rows, long long *cols)
{
    if (nr < 0 || nc < 0) return -1;
    if (nr == 0 || nc == 0) return 0;
    for (size_t t = 0; t < (size_t)nr * nc; ++t)
        if (isnan(cost_in[t]) || cost_in[t] == -INFINITY) return -1;
    int transpose = nc < nr;
    double *cost = (double *)malloc(sizeof(double) * (size_t)nr * nc);
    if (transpose) {
        for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) cost[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
        int t = nr; nr = nc; nc = t;
    } else memcpy(cost, cost_in, sizeof(double) * (size_t)nr * nc);
    double *u = (double *)calloc((size_t)nr, sizeof(double)), *v = (double *)calloc((size_t)nc, sizeof(double));
    double *spc = (double *)malloc(sizeof(double) * (size_t)nc);
    int *path = (int *)malloc(sizeof(int) * (size_t)nc), *col4row = (int *)malloc(sizeof(int) * (size_t)nr);
    int *row4col = (int *)malloc(sizeof(int) * (size_t)nc), *remaining = (int *)malloc(sizeof(int) * (size_t)nc);
    unsigned char *SR = (unsigned char *)malloc((size_t)nr), *SC = (unsigned char *)malloc((size_t)nc);
    for (int j = 0; j < nc; ++j) { path[j] = -1; row4col[j] = -1; }
    for (int i = 0; i < nr; ++i) col4row[i] = -1;
    int ok = 1;
    for (int cur = 0; cur < nr && ok; ++cur) {
        double min_val;
        int sink = augment_from(nr, nc, cost, u, v, path, row4col, spc, cur, SR, SC, remaining, &min_val);
        if (sink < 0) { ok = 0; break; }
        u[cur] += min_val;
        for (int i = 0; i < nr; ++i) if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
        for (int j = 0; j < nc; ++j) if (SC[j]) v[j] -= min_val - spc[j];
        int j = sink;
        for (;;) {
            int i = path[j];
            row4col[j] = i;
            int t = col4row[i]; col4row[i] = j; j = t;
            if (i == cur) break;
        }
    }
    int n = -1;
    if (ok) {
        n = nr;
        if (transpose) {
            long long *pairs = (long long *)malloc(sizeof(long long) * 2 * (size_t)nr);
            for (int i = 0; i < nr; ++i) { pairs[2 * i] = col4row[i]; pairs[2 * i + 1] = i; }
            qsort(pairs, (size_t)nr, 2 * sizeof(long long), cmp_pair);
            for (int i = 0; i < nr; ++i) { rows[i] = pairs[2 * i]; cols[i] = pairs[2 * i + 1]; }
            free(pairs);
        } else {
            for (int i = 0; i < nr; ++i) { rows[i] = i; cols[i] = col4row[i]; }
        }
    }
    free(cost); free(u); free(v); free(spc); free(path); free(col4row); free(row4col); free(remaining); free(SR); free(SC);
    return n;
}

/*
 * ORACLE (test infrastructure, never shipped or measured as the product).
 *
 * CPU restatement of the linear-assignment solver the reference calls:
 *   pleas/core/solvers.py:29-31  ->  scipy.optimize.linear_sum_assignment(A, maximize)
 * scipy's solver is a third-party dependency that is not under /root/reference
 * (requirements.txt:160 pins scipy==1.11.4; the image has 1.15.3).  Its published
 * algorithm is D. F. Crouse, "On implementing 2D rectangular assignment
 * algorithms", IEEE T-AES 52(4), 2016: shortest augmenting paths with dual
 * variables, rows inserted in order 0..n-1, no initial reduction.  The tie
 * behaviour restated here is pinned against scipy itself
 * by tests/test_oracle_lap.py and against tests/golden/lap_small.npz.
 *
 * Square problems only (the hot path only produces C x C costs).
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/liblsap_oracle.so oracle/lsap.c
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* cost: n*n row-major (double). maximize != 0 negates. col4row out: n int64. Returns 0, or -1 if infeasible. */
int oracle_lsap_f64(const double *cost_in, int64_t n, int maximize, int64_t *col4row)
{
    if (n <= 0) return 0;
    double *cost = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
    double *u = (double *)calloc((size_t)n, sizeof(double));
    double *v = (double *)calloc((size_t)n, sizeof(double));
    double *shortest = (double *)malloc(sizeof(double) * (size_t)n);
    int64_t *path = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t *row4col = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t *remaining = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    char *seen_row = (char *)malloc((size_t)n);
    char *seen_col = (char *)malloc((size_t)n);
    int rc = 0;

    for (int64_t k = 0; k < n * n; ++k) cost[k] = maximize ? -cost_in[k] : cost_in[k];
    for (int64_t k = 0; k < n; ++k) { path[k] = -1; row4col[k] = -1; col4row[k] = -1; }

    for (int64_t cur = 0; cur < n && rc == 0; ++cur) {
        /* --- search: grow a shortest-path tree from row `cur` until an unassigned column is reached */
        int64_t live = n;
        for (int64_t t = 0; t < n; ++t) {
            remaining[t] = n - 1 - t;      /* reversed fill: constant matrices give the identity */
            seen_row[t] = 0; seen_col[t] = 0;
            shortest[t] = INFINITY;
        }
        double dist = 0.0;
        int64_t i = cur, sink = -1;
        while (sink < 0) {
            int64_t best_at = -1;
            double best = INFINITY;
            seen_row[i] = 1;
            for (int64_t t = 0; t < live; ++t) {
                int64_t j = remaining[t];
                double r = dist + cost[i * n + j] - u[i] - v[j];
                if (r < shortest[j]) { shortest[j] = r; path[j] = i; }
                /* on an exact tie prefer a column that ends the search */
                if (shortest[j] < best || (shortest[j] == best && row4col[j] < 0)) {
                    best = shortest[j];
                    best_at = t;
                }
            }
            dist = best;
            if (dist == INFINITY) { rc = -1; break; }
            int64_t j = remaining[best_at];
            if (row4col[j] < 0) sink = j; else i = row4col[j];
            seen_col[j] = 1;
            remaining[best_at] = remaining[--live];   /* swap-remove */
        }
        if (rc) break;
        /* --- dual update */
        u[cur] += dist;
        for (int64_t r = 0; r < n; ++r)
            if (seen_row[r] && r != cur) u[r] += dist - shortest[col4row[r]];
        for (int64_t c = 0; c < n; ++c)
            if (seen_col[c]) v[c] -= dist - shortest[c];
        /* --- augment along the tree path back to `cur` */
        int64_t j = sink;
        for (;;) {
            int64_t r = path[j];
            row4col[j] = r;
            int64_t prev = col4row[r];
            col4row[r] = j;
            j = prev;
            if (r == cur) break;
        }
    }
    free(cost); free(u); free(v); free(shortest); free(path); free(row4col); free(remaining);
    free(seen_row); free(seen_col);
    return rc;
}

/* float32 input convenience: upcast exactly as numpy does when scipy receives float32. */
int oracle_lsap_f32(const float *cost_in, int64_t n, int maximize, int64_t *col4row)
{
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
    for (int64_t k = 0; k < n * n; ++k) tmp[k] = (double)cost_in[k];
    int rc = oracle_lsap_f64(tmp, n, maximize, col4row);
    free(tmp);
    return rc;
}

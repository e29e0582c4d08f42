// Standalone replay of the closed-form PLeaS kernels on the exact ResNet-101 layer list (tools/hipbench/rn101_layers.txt),
// batch 16: the grouped normal-equations launch (neq_batch_kernel + neq_reduce_kernel: A += U^T U of all 104 layers)
// `reps` times, then ONE batched Cholesky solve of all systems (potrf_diag / trsm_rows / trail_update), so that
// rocprofv3 --kernel-trace / --pmc can be pointed at a plain binary.  The stem (3 input channels) takes the vendor path in
// the product and is skipped here.  Usage: neq_batch_rn101 <layers.txt> <reps> [solve=1] [only: 0 all, 1 3x3 stride 1, 2 1x1, 3 rest]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pleas_hip.h"
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
static float* dev_rand(size_t n) { std::vector<float> h(n); for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f; float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d; }
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "rn101_layers.txt"; int reps = argc > 2 ? atoi(argv[2]) : 5, N = 16;
    const int solve = argc > 3 ? atoi(argv[3]) : 1, only = argc > 4 ? atoi(argv[4]) : 0;
    FILE* f = fopen(path, "r"); if (!f) { printf("cannot open %s\n", path); return 1; }
    int n; fscanf(f, "%d", &n);
    std::vector<pleas_neq_layer> L; std::vector<int> Ks, Ns; double flops = 0, bytes = 0, sumk2 = 0, chol = 0;
    for (int i = 0; i < n; ++i) { int co, ci, h, w, k, s, p; fscanf(f, "%d %d %d %d %d %d %d", &co, &ci, &h, &w, &k, &s, &p);
        if (ci < 16) continue;
        const int kind = (k == 3 && s == 1) ? 1 : (k == 1 && s == 1) ? 2 : 3;
        if (only && kind != only) continue;
        int ho = (h + 2 * p - k) / s + 1, wo = (w + 2 * p - k) / s + 1; size_t P = (size_t)N * ho * wo; size_t K = (size_t)ci * k * k;
        pleas_neq_layer l{}; l.N = N; l.Cin = ci; l.Hin = h; l.Win = w; l.KH = l.KW = k; l.stride = s; l.pad = p;
        l.ip = dev_rand((size_t)N * ci * h * w); float* A; hipMalloc(&A, K * K * 4); hipMemset(A, 0, K * K * 4); l.A = A;
        L.push_back(l); Ks.push_back((int)K); Ns.push_back(co);
        flops += (double)K * K * (double)P; bytes += (double)N * ci * h * w * 4; sumk2 += (double)K * K;
        chol += (double)K * K * K / 3 + 2.0 * co * (double)K * K; }
    n = (int)L.size();
    size_t wsb = pleas_normal_eq_ws_bytes(L.data(), n); void* ws; hipMalloc(&ws, wsb);
    int rc = pleas_normal_eq_accum(L.data(), n, ws, wsb, 1, 0); if (rc) { printf("error %d %s\n", rc, pleas_last_error()); return 1; }
    hipDeviceSynchronize(); hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) pleas_normal_eq_accum(L.data(), n, ws, wsb, 0, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    double info[4] = {0, 0, 0, 0}; pleas_normal_eq_plan_info(L.data(), n, info);
    printf("layers=%d sum K^2 = %.3e; triangle %.1f GFLOP, %.1f MB of inputs per batch; %.3f ms per batch (incl. slab reduce) -> %.1f TF/s (ws %.1f MB)\n",
           n, sumk2, flops / 1e9, bytes / 1e6, ms / reps, flops / (ms / reps * 1e-3) / 1e12, wsb / 1e6);
    printf("   executed by the grid: %.1f GFLOP in %d items (%d blocks left to finalize) -> %.1f TF/s = %.3f of the 157.3 TF/s fp32 matrix peak\n",
           info[1] / 1e9, (int)info[2], (int)info[3], info[1] / (ms / reps * 1e-3) / 1e12, info[1] / (ms / reps * 1e-3) / 1e12 / 157.3);
    if (solve) {
        std::vector<float*> Ap, Bp;
        for (int i = 0; i < n; ++i) { Ap.push_back(L[i].A); Bp.push_back(dev_rand((size_t)Ns[i] * Ks[i])); }
        int* info; hipMalloc(&info, n * sizeof(int));
        pleas_normal_eq_finalize(L.data(), n, 0);       // the lag-class copies, once (as NormalEqFitter.solve does)
        hipEventRecord(a, 0);
        rc = pleas_cholesky_solve_batched(Ap.data(), Bp.data(), Ks.data(), Ns.data(), n, 1e-2f, info, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        if (rc) { printf("solve error %d %s\n", rc, pleas_last_error()); return 1; }
        std::vector<int> hinfo(n); hipMemcpy(hinfo.data(), info, n * sizeof(int), hipMemcpyDeviceToHost);
        int bad = 0; for (int v : hinfo) bad += v != 0;
        printf("batched Cholesky + substitution of %d systems: %.1f ms, %.1f GFLOP -> %.1f TF/s; %d flagged\n", n, ms, chol / 1e9, chol / (ms * 1e-3) / 1e12, bad);
    }
    return 0;
}

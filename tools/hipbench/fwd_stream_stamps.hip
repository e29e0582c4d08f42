// Per-role cycle accounting of the streamed forward (build the library with -DPLEAS_FWDS_STAMPS=1): for every workgroup the
// wall cycles, the cycles its MFMA waves / its two producer groups spent OUTSIDE the tick barrier, chunks and ticks.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pleas_hip.h"
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
extern "C" int pleas_fwds_stamps_read(long long* out, int n_wg);
static float* dev_rand(size_t n) { std::vector<float> h(n); for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f; float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d; }
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "rn101_layers.txt"; int reps = argc > 2 ? atoi(argv[2]) : 5, N = 16;
    FILE* f = fopen(path, "r"); if (!f) { printf("cannot open %s\n", path); return 1; }
    int n; fscanf(f, "%d", &n);
    std::vector<pleas_fwd_layer> L(n); double flops = 0;
    for (auto& l : L) { int co, ci, h, w, k, s, p; fscanf(f, "%d %d %d %d %d %d %d", &co, &ci, &h, &w, &k, &s, &p);
        int ho = (h + 2 * p - k) / s + 1, wo = (w + 2 * p - k) / s + 1; size_t P = (size_t)N * ho * wo;
        l.N = N; l.Cout = co; l.Cin = ci; l.Hin = h; l.Win = w; l.KH = l.KW = k; l.stride = s; l.pad = p; l.Csrc = co; l.n_merged = co; l.flags = (k > 1 && ci % 32 == 0) ? PLEAS_FWD_KPOS_MAJOR : 0;
        l.dscale = 2.0f / (co * P); l.loss_scale = 1.0f / (co * P);
        l.ip = dev_rand((size_t)N * ci * h * w); l.w = dev_rand((size_t)co * ci * k * k); l.bias = nullptr;
        l.o1 = dev_rand(co * P); l.o2 = dev_rand(co * P); float* r; hipMalloc(&r, co * P * 4); l.resid = r;
        std::vector<int32_t> id(co); for (int i = 0; i < co; ++i) id[i] = i; int32_t* m; hipMalloc(&m, co * 4); hipMemcpy(m, id.data(), co * 4, hipMemcpyHostToDevice); l.row1 = m; l.row2 = m;
        flops += 2.0 * co * ci * k * k * (double)P; }
    float* loss; hipMalloc(&loss, n * 4);
    size_t wsb = pleas_fwd_batch_ws_bytes(L.data(), n); void* ws; hipMalloc(&ws, wsb);
    int rc = pleas_fwd_batch(L.data(), n, loss, ws, wsb, 1, 0); if (rc) { printf("error %d %s\n", rc, pleas_last_error()); return 1; }
    hipDeviceSynchronize(); hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) pleas_fwd_batch(L.data(), n, loss, ws, wsb, 0, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    printf("layers=%d %.1f GFLOP per update; %.3f ms per update -> %.1f TF/s\n", n, flops / 1e9, ms / reps, flops / (ms / reps * 1e-3) / 1e12);
    std::vector<long long> st(256 * 16);
    if (pleas_fwds_stamps_read(st.data(), 256)) { printf("no stamps\n"); return 1; }
    double sum[16] = {0}; long long mx[16] = {0}, mn[16]; for (int k = 0; k < 16; ++k) mn[k] = 1ll << 62;
    for (int w = 0; w < 256; ++w) for (int k = 0; k < 16; ++k) { sum[k] += st[w * 16 + k]; mx[k] = std::max(mx[k], st[w * 16 + k]); mn[k] = std::min(mn[k], st[w * 16 + k]); }
    const char* names[8] = {"wall cycles", "consumer busy", "chunks", "group0 busy", "group0 active ticks", "group1 busy", "group1 active ticks", "ticks"};
    for (int k = 0; k < 8; ++k) printf("%-22s mean %12.0f  min %12lld  max %12lld\n", names[k], sum[k] / 256, mn[k], mx[k]);
    printf("per tick: wall %.0f cycles, consumer busy %.0f; per active tick: group0 busy %.0f, group1 busy %.0f (clock64 ticks)\n",
           sum[0] / sum[7], sum[1] / sum[7], sum[3] / sum[4], sum[5] / sum[6]);
    printf("group 0 per active tick: (1) wait + LDS writes %.0f, (2) epilogue consume %.0f, (3) enter + requests %.0f, (4) gathers %.0f; of (1): waiting for vmcnt(0) %.0f\n",
           sum[8] / sum[4], sum[9] / sum[4], sum[10] / sum[4], sum[11] / sum[4], sum[12] / sum[4]);
    return 0;
}

// Standalone replay of the grouped weight-gradient launch on the exact ResNet-101 layer list
// (tools/hipbench/rn101_layers.txt), batch 16: same kernels, same grid as one bench.py PLeaS update
// (argv[4] = 1 skips the 3-channel stem, as the launches before round 4 did: it went to the vendor's kernel then).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <dlfcn.h>
#include "pleas_hip.h"
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
static float* dev_rand(size_t n) { std::vector<float> h(n); for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f; float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d; }
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "rn101_layers.txt"; int reps = argc > 2 ? atoi(argv[2]) : 5, N = 16;
    FILE* f = fopen(path, "r"); if (!f) { printf("cannot open %s\n", path); return 1; }
    int n; fscanf(f, "%d", &n);
    std::vector<pleas_wgrad_layer> L; double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) { int co, ci, h, w, k, s, p; fscanf(f, "%d %d %d %d %d %d %d", &co, &ci, &h, &w, &k, &s, &p);
        if (ci < 16 && argc > 4 && atoi(argv[4]) == 1) continue;
        int ho = (h + 2 * p - k) / s + 1, wo = (w + 2 * p - k) / s + 1; size_t P = (size_t)N * ho * wo;
        pleas_wgrad_layer l{}; l.N = N; l.Cout = co; l.Cin = ci; l.Hin = h; l.Win = w; l.KH = l.KW = k; l.stride = s; l.pad = p;
        l.flags = (k > 1 && ci % 32 == 0) ? PLEAS_WGRAD_KPOS_MAJOR : 0;
        l.resid = dev_rand(co * P); l.ip = dev_rand((size_t)N * ci * h * w); float* g; hipMalloc(&g, (size_t)co * ci * k * k * 4); l.grad = g;
        L.push_back(l);
        flops += 2.0 * co * ci * k * k * (double)P; bytes += ((double)N * ci * h * w + (double)co * P + (double)co * ci * k * k) * 4; }
    n = (int)L.size();
    size_t wsb = pleas_wgrad_batch_ws_bytes(L.data(), n); void* ws; hipMalloc(&ws, wsb);
    int rc = pleas_wgrad_batch(L.data(), n, ws, wsb, 1, 0); if (rc) { printf("error %d %s\n", rc, pleas_last_error()); return 1; }
    hipDeviceSynchronize(); hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) pleas_wgrad_batch(L.data(), n, ws, wsb, 0, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    printf("layers=%d algorithmic %.1f GFLOP %.1f MB per update; %.3f ms per update (incl. slab reduce) -> %.1f TF/s\n", n, flops / 1e9, bytes / 1e6, ms / reps, flops / (ms / reps * 1e-3) / 1e12);
    // PLEAS_TIMELINE_OUT=file (study builds of the library only): one more launch, its per-item record (start, end in 10 ns
    // ticks, hardware id, work) written as int64 quadruples
    if (const char* out = getenv("PLEAS_TIMELINE_OUT")) {
        typedef int (*read_fn)(long long*, int);
        read_fn rd = (read_fn)dlsym(RTLD_DEFAULT, "pleas_wgrad_timeline_read");
        if (!rd) { printf("no timeline in this build\n"); return 1; }
        std::vector<long long> rec((size_t)32768 * 4);
        rd(rec.data(), 32768);                 // reset
        hipDeviceSynchronize();
        pleas_wgrad_batch(L.data(), n, ws, wsb, 0, 0); hipDeviceSynchronize();
        const int got = rd(rec.data(), 32768);
        FILE* o = fopen(out, "wb"); fwrite(rec.data(), sizeof(long long) * 4, got > 0 ? got : 0, o); fclose(o);
        printf("timeline: %d items -> %s\n", got, out);
    }
    return 0;
}

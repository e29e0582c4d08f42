// Standalone replay of the grouped merged-layer forward (+target+residual+loss) on the exact ResNet-101 layer
// list (tools/hipbench/rn101_layers.txt), batch 16, full merge: same kernel, same grid as one bench.py PLeaS update.
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
    const int kpos = argc > 3 ? atoi(argv[3]) : 1;   // 1: k x k layers with Cin % 32 == 0 use kernel-position-major weights
    // PLEAS_TAP_GROUP=g, PLEAS_TAP_WINDOW=w: the source outputs o1 / o2 are the w-th 16-sample window of tensors that hold
    // g times as many samples (the taps of ONE source forward over g updates' batches, as PleasFitter.steps groups them)
    const int tap_group = getenv("PLEAS_TAP_GROUP") ? atoi(getenv("PLEAS_TAP_GROUP")) : 1;
    const int tap_window = getenv("PLEAS_TAP_WINDOW") ? atoi(getenv("PLEAS_TAP_WINDOW")) : 0;
    FILE* f = fopen(path, "r"); if (!f) { printf("cannot open %s\n", path); return 1; }
    int n; fscanf(f, "%d", &n);
    std::vector<pleas_fwd_layer> L(n); double flops = 0, bytes = 0;
    for (auto& l : L) { int co, ci, h, w, k, s, p; fscanf(f, "%d %d %d %d %d %d %d", &co, &ci, &h, &w, &k, &s, &p);
        int ho = (h + 2 * p - k) / s + 1, wo = (w + 2 * p - k) / s + 1; size_t P = (size_t)N * ho * wo;
        l.N = N; l.Cout = co; l.Cin = ci; l.Hin = h; l.Win = w; l.KH = l.KW = k; l.stride = s; l.pad = p; l.Csrc = co; l.n_merged = co; l.flags = (kpos && k > 1 && ci % 32 == 0) ? PLEAS_FWD_KPOS_MAJOR : 0;
        l.dscale = 2.0f / (co * P); l.loss_scale = 1.0f / (co * P);
        l.ip = dev_rand((size_t)N * ci * h * w); l.w = dev_rand((size_t)co * ci * k * k); l.bias = nullptr;
        l.o1 = dev_rand(co * P * tap_group) + (size_t)co * P * tap_window; l.o2 = dev_rand(co * P * tap_group) + (size_t)co * P * tap_window; float* r; hipMalloc(&r, co * P * 4); l.resid = r;
        std::vector<int32_t> id(co); for (int i = 0; i < co; ++i) id[i] = i; int32_t* m; hipMalloc(&m, co * 4); hipMemcpy(m, id.data(), co * 4, hipMemcpyHostToDevice); l.row1 = m; l.row2 = m;
        flops += 2.0 * co * ci * k * k * (double)P; bytes += ((double)N * ci * h * w + 3.0 * co * P + (double)co * ci * k * k) * 4; }
    float* loss; hipMalloc(&loss, n * 4);
    size_t wsb = pleas_fwd_batch_ws_bytes(L.data(), n); void* ws; hipMalloc(&ws, wsb);
    int rc = pleas_fwd_batch(L.data(), n, loss, ws, wsb, 1, 0); if (rc) { printf("error %d %s\n", rc, pleas_last_error()); return 1; }
    hipDeviceSynchronize(); hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) pleas_fwd_batch(L.data(), n, loss, ws, wsb, 0, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    printf("layers=%d algorithmic %.1f GFLOP %.1f MB per update; %.3f ms per update -> %.1f TF/s\n", n, flops / 1e9, bytes / 1e6, ms / reps, flops / (ms / reps * 1e-3) / 1e12);
    // PLEAS_TIMELINE_OUT=file (study builds of the library only): one more launch, its per-item record (start, end in 10 ns
    // ticks, hardware id, work) written as int64 quadruples
    if (const char* out = getenv("PLEAS_TIMELINE_OUT")) {
        typedef int (*read_fn)(long long*, int);
        read_fn rd = (read_fn)dlsym(RTLD_DEFAULT, "pleas_fwd_timeline_read");
        if (!rd) { printf("no timeline in this build\n"); return 1; }
        std::vector<long long> rec((size_t)32768 * 4);
        rd(rec.data(), 32768);                 // reset
        hipDeviceSynchronize();
        pleas_fwd_batch(L.data(), n, loss, ws, wsb, 0, 0); hipDeviceSynchronize();
        const int got = rd(rec.data(), 32768);
        FILE* o = fopen(out, "wb"); fwrite(rec.data(), sizeof(long long) * 4, got > 0 ? got : 0, o); fclose(o);
        printf("timeline: %d items -> %s\n", got, out);
    }
    return 0;
}

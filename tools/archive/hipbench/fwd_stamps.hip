// In-kernel cycle stamps of fwd_batch_kernel (library built with -DPLEAS_FWD_ABLATE=16): per work item, cycles spent in
// the prologue (decode + first loads + first barrier), the K loop and the epilogue, aggregated by (chunks, TM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include "pleas_hip.h"
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
extern "C" int pleas_fwd_debug_read(long long* out, int n_items);
static float* dev_rand(size_t n) { std::vector<float> h(n); for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f; float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d; }
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "rn101_layers.txt"; int N = 16;
    FILE* f = fopen(path, "r"); if (!f) { printf("cannot open %s\n", path); return 1; }
    int n; fscanf(f, "%d", &n);
    std::vector<pleas_fwd_layer> L(n);
    for (auto& l : L) { int co, ci, h, w, k, s, p; fscanf(f, "%d %d %d %d %d %d %d", &co, &ci, &h, &w, &k, &s, &p);
        int ho = (h + 2 * p - k) / s + 1, wo = (w + 2 * p - k) / s + 1; size_t P = (size_t)N * ho * wo;
        l.N = N; l.Cout = co; l.Cin = ci; l.Hin = h; l.Win = w; l.KH = l.KW = k; l.stride = s; l.pad = p; l.Csrc = co; l.n_merged = co; l.flags = (k > 1 && ci % 32 == 0) ? PLEAS_FWD_KPOS_MAJOR : 0;
        l.dscale = 2.0f / (co * P); l.loss_scale = 1.0f / (co * P);
        l.ip = dev_rand((size_t)N * ci * h * w); l.w = dev_rand((size_t)co * ci * k * k); l.bias = nullptr;
        l.o1 = dev_rand(co * P); l.o2 = dev_rand(co * P); float* r; hipMalloc(&r, co * P * 4); l.resid = r;
        std::vector<int32_t> id(co); for (int i = 0; i < co; ++i) id[i] = i; int32_t* m; hipMalloc(&m, co * 4); hipMemcpy(m, id.data(), co * 4, hipMemcpyHostToDevice); l.row1 = m; l.row2 = m; }
    float* loss; hipMalloc(&loss, n * 4);
    size_t wsb = pleas_fwd_batch_ws_bytes(L.data(), n); void* ws; hipMalloc(&ws, wsb);
    pleas_fwd_batch(L.data(), n, loss, ws, wsb, 1, 0); hipDeviceSynchronize();
    pleas_fwd_batch(L.data(), n, loss, ws, wsb, 0, 0); hipDeviceSynchronize();
    std::vector<long long> st(32768 * 4, 0);
    if (pleas_fwd_debug_read(st.data(), 32768)) { printf("debug read failed\n"); return 1; }
    struct Agg { double pro = 0, loop = 0, epi = 0; long cnt = 0; };
    std::map<long long, Agg> agg; Agg all;
    for (int i = 0; i < 32768; ++i) { long long key = st[i * 4 + 3]; if (!key) continue; Agg& a = agg[key];
        a.pro += st[i * 4]; a.loop += st[i * 4 + 1]; a.epi += st[i * 4 + 2]; a.cnt++; all.pro += st[i * 4]; all.loop += st[i * 4 + 1]; all.epi += st[i * 4 + 2]; all.cnt++; }
    printf("%8s %4s %6s %10s %12s %12s %10s %8s\n", "chunks", "TM", "items", "prologue", "loop", "loop/chunk", "epilogue", "share%");
    double tot = all.pro + all.loop + all.epi;
    for (auto& kv : agg) { const Agg& a = kv.second; long ch = kv.first / 1000, tm = kv.first % 1000;
        printf("%8ld %4ld %6ld %10.0f %12.0f %12.0f %10.0f %8.1f\n", ch, tm, a.cnt, a.pro / a.cnt, a.loop / a.cnt, a.loop / a.cnt / ch, a.epi / a.cnt, 100 * (a.pro + a.loop + a.epi) / tot); }
    printf("all items %ld: prologue %.1f%%  loop %.1f%%  epilogue %.1f%% of stamped cycles (clock64 ticks)\n", all.cnt, 100 * all.pro / tot, 100 * all.loop / tot, 100 * all.epi / tot);
    return 0;
}

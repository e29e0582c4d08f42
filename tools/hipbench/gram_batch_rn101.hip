// Standalone replay of the grouped matching contraction on the exact ResNet-101 node list
// (tools/hipbench/rn101_nodes.txt), batch 16: same kernel, same grid as one bench.py matching batch.
// Used for rocprofv3 --pmc passes (a bare C++ process is robust under the counter collector).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pleas_hip.h"
#pragma clang diagnostic ignored "-Wunused-value"
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "rn101_nodes.txt";
    int reps = argc > 2 ? atoi(argv[2]) : 5, B = 16;
    if (argc > 4) pleas_gram_batch_tune(atoi(argv[3]), atoi(argv[4]));
    FILE* f = fopen(path, "r");
    if (!f) { printf("cannot open %s\n", path); return 1; }
    int n, ng; fscanf(f, "%d %d", &n, &ng);
    std::vector<int> gC(ng); for (auto& c : gC) fscanf(f, "%d", &c);
    std::vector<pleas_gram_node> nodes(n);
    double flops = 0, bytes = 0; size_t maxel = 0;
    const bool with_sources = strstr(path, "derived") != nullptr;   // 4th column: index of the node this one is derived from, or -1
    int n_derived = 0;
    for (auto& nd : nodes) { int C, HW, g, src = -1; fscanf(f, "%d %d %d", &C, &HW, &g); if (with_sources) fscanf(f, "%d", &src);
        nd.B = B; nd.C = C; nd.HW = HW; nd.group = g;
        if (src >= 0) { nd.derived = 1; nd.source = src; ++n_derived; continue; }
        flops += 2.0 * C * C * (double)B * HW; bytes += 2.0 * C * (double)B * HW * 4; maxel = std::max(maxel, (size_t)B * C * HW); }
    // distinct operand buffers per node (as in the real forward): total = bytes
    std::vector<float> h(maxel); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& nd : nodes) {
        if (nd.derived) { float* v; hipMalloc(&v, (size_t)nd.C * 4); hipMemcpy(v, h.data(), (size_t)nd.C * 4, hipMemcpyHostToDevice);
            nd.scale_x = nd.shift_x = nd.scale_y = nd.shift_y = v; continue; }
        size_t el = (size_t)nd.B * nd.C * nd.HW; float *x, *y; hipMalloc(&x, el * 4); hipMalloc(&y, el * 4);
        hipMemcpy(x, h.data(), el * 4, hipMemcpyHostToDevice); hipMemcpy(y, h.data() + 1, (el - 1) * 4, hipMemcpyHostToDevice); nd.x = x; nd.y = y; }
    std::vector<float*> acc(ng); for (int g = 0; g < ng; ++g) { hipMalloc(&acc[g], (size_t)gC[g] * gC[g] * 4); hipMemset(acc[g], 0, (size_t)gC[g] * gC[g] * 4); }
    size_t wsb = pleas_gram_batch_ws_bytes(nodes.data(), n, gC.data(), ng); void* ws; hipMalloc(&ws, wsb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int rc = pleas_gram_batch(nodes.data(), n, acc.data(), gC.data(), ng, 1, 1, ws, wsb, 1, 0);
    if (rc) { printf("error %d %s\n", rc, pleas_last_error()); return 1; }
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) pleas_gram_batch(nodes.data(), n, acc.data(), gC.data(), ng, 1, 1, ws, wsb, 0, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (n_derived) printf("(%d of the nodes derived, not contracted) ", n_derived);
    printf("nodes=%d groups=%d algorithmic %.1f GFLOP %.1f MB per batch; ws %.1f MB; %.3f ms per batch (contract+reduce) -> %.1f TF/s\n",
           n, ng, flops / 1e9, bytes / 1e6, wsb / 1e6, ms / reps, flops / (ms / reps * 1e-3) / 1e12);
    return 0;
}

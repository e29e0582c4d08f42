// Standalone ablation harness for gram_partial_kernel (build on the GPU box with hipcc).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../pleas_merging_amd/csrc/gram.hip"
namespace pleas { thread_local char g_last_error[256]; bool g_prof_on = false;
void prof_begin(int, double, double, hipStream_t) {} void prof_end(hipStream_t) {} }

int main(int argc, char** argv) {
    int B = 16, C = argc > 1 ? atoi(argv[1]) : 1024, HW = argc > 2 ? atoi(argv[2]) : 196;
    int tb = argc > 3 ? atoi(argv[3]) : 512, mc = argc > 4 ? atoi(argv[4]) : 2;
    pleas_gram_tune(tb, mc);
    size_t n = (size_t)B * C * HW;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    float *x, *y, *acc; void* ws;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&acc, (size_t)C * C * 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(y, h.data(), n * 4, hipMemcpyHostToDevice);
    size_t wsb = pleas_gram_ws_bytes(B, C, HW); hipMalloc(&ws, wsb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) pleas_gram_accum(x, y, B, C, HW, 1, 1, acc, ws, wsb, 0);
    hipDeviceSynchronize();
    int reps = 50;
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) pleas_gram_accum(x, y, B, C, HW, 1, 1, acc, ws, wsb, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double fl = 2.0 * C * C * (double)B * HW;
    printf("C=%d HW=%d tb=%d mc=%d: %.1f us per call (partial+finalize) -> %.1f TF/s\n", C, HW, tb, mc, ms / reps * 1e3, fl / (ms / reps * 1e-3) / 1e12);
    return 0;
}

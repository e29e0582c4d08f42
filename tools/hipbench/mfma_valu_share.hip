// Do fp32-input MFMAs and ordinary VALU instructions of OTHER waves on the same SIMD overlap, or do they take turns?
// One workgroup per CU, 8 waves (two per SIMD): waves 0-3 issue back-to-back v_mfma_f32_32x32x2_f32 (4 independent
// accumulators), waves 4-7 run a chain of `valu_per_iter` dependent-free v_fma_f32 per loop iteration (or idle).
// Reported: MFMA rate with the VALU waves idle / busy, and the VALU waves' own instruction rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>   // 0: MFMA waves only, 1: + VALU waves (fp32 fma), 2: VALU waves only, 3: MFMA + SALU-only waves, 4: bf16 MFMA + VALU
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    const long long t0 = clock64();
    if (wave < 4) {
        if (MODE == 2) { out[blockIdx.x * 512 + threadIdx.x] = 0; return; }
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        float x = threadIdx.x * 0.001f, y = 1.0f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
        }
        float s = 0; for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (threadIdx.x == 0) cyc[blockIdx.x * 2] = clock64() - t0;
    } else {
        if (MODE == 0) { out[blockIdx.x * 512 + threadIdx.x] = 0; return; }
        float v[8]; for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.01f + j;
        const float c = 1.0001f, d = 0.5f;
        if (MODE == 3) {
            int sacc = 0;
            for (int i = 0; i < iters * 32; ++i) { asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 3\n s_add_i32 %0, %0, 5\n s_add_i32 %0, %0, 7" : "+s"(sacc)); }
            out[blockIdx.x * 512 + threadIdx.x] = (float)sacc;
        } else {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], c, d);     // 128 independent-ish VALU per iteration
            }
            float s = 0; for (int j = 0; j < 8; ++j) s += v[j];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
        if (threadIdx.x == 256) cyc[blockIdx.x * 2 + 1] = clock64() - t0;
    }
}
template <int MODE> static void run(const char* tag, float* out, long long* cyc, int iters) {
    hipMemset(cyc, 0, 256 * 2 * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, cyc); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
    hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    long long h[512]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, v = 0; for (int i = 0; i < 256; ++i) { m += h[2 * i]; v += h[2 * i + 1]; } m /= 256; v /= 256;
    const double mfmas = (double)iters * 32, valus = (double)iters * 128;
    printf("%-44s %.3f ms | MFMA waves: %.0f cycles = %.1f per MFMA | VALU waves: %.0f cycles = %.2f per VALU instr\n", tag, ms,
           m, m / mfmas, v, v / valus);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4); long long* cyc; hipMalloc(&cyc, 256 * 2 * 8);
    const int iters = 2000;
    run<0>("fp32 MFMA waves alone", out, cyc, iters);
    run<2>("VALU waves alone", out, cyc, iters);
    run<1>("fp32 MFMA + VALU waves on every SIMD", out, cyc, iters);
    run<3>("fp32 MFMA + SALU-only waves", out, cyc, iters);
    run<0>("fp32 MFMA waves alone (again)", out, cyc, iters);
    return 0;
}

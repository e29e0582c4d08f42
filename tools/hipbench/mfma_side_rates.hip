// What does a wave that shares its SIMD with a wave issuing back-to-back fp32 MFMAs still get done?  Per instruction class:
// cycles per instruction of the SIDE wave alone and beside the MFMA wave (and the MFMA wave's own rate beside it).
//   classes: 0 v_fma_f32 (vector ALU)   1 v_readlane + s_add (VALU -> SGPR)   2 s_add_i32 (scalar ALU)
//            3 ds_write_b128            4 ds_read_b128                        5 global_load_dwordx4 (L2-resident, 1 KB per wave)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CLS, bool MFMA_ON, bool SIDE_ON, int PRIO>
__global__ __launch_bounds__(512) void k(float* out, const float* __restrict__ src, int iters, long long* cyc) {
    __shared__ f32x4 lds[2048];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t0 = clock64();
    if (wave < 4) {
        if (!MFMA_ON) return;
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        const float x = threadIdx.x * 0.001f, y = 1.0f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
        float s = 0; for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (threadIdx.x == 0) cyc[blockIdx.x * 2] = clock64() - t0;
        return;
    }
    if (!SIDE_ON) return;
    if (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);      // the side wave outranks the (older) MFMA wave at the issue arbiter
    float v[8]; for (int j = 0; j < 8; ++j) v[j] = lane * 0.01f + j;
    f32x4 w = {v[0], v[1], v[2], v[3]}, racc = {0, 0, 0, 0};
    int sacc = 0;
    const f32x4* g = reinterpret_cast<const f32x4*>(src) + (size_t)(blockIdx.x % 64) * 4096 + lane;
    const int n = iters / 4;      // 32 side instructions per iteration
    for (int i = 0; i < n; ++i) {
        if (CLS == 0) {
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], 1.0001f, 0.5f);
        } else if (CLS == 1) {
#pragma unroll
            for (int u = 0; u < 32; ++u) sacc += __builtin_amdgcn_readlane(__float_as_int(v[u & 7]), u & 63);
        } else if (CLS == 2) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("s_add_i32 %0, %0, 7" : "+s"(sacc));
        } else if (CLS == 3) {
#pragma unroll
            for (int u = 0; u < 32; ++u) lds[(wave - 4) * 512 + (u & 7) * 64 + lane] = w;
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        } else if (CLS == 4) {
#pragma unroll
            for (int u = 0; u < 32; ++u) racc += lds[(wave - 4) * 512 + (u & 7) * 64 + lane];
        } else {
#pragma unroll
            for (int u = 0; u < 32; ++u) racc += g[(size_t)((i * 32 + u) & 63) * 64];
        }
    }
    float s = racc[0] + racc[1] + racc[2] + racc[3] + (float)sacc;
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 256) cyc[blockIdx.x * 2 + 1] = clock64() - t0;
}
template <int CLS, int PRIO> static void run(const char* tag, float* out, const float* src, long long* cyc, int iters) {
    long long h[512]; double m[3], v[3];
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(cyc, 0, sizeof(h));
        if (mode == 0) hipLaunchKernelGGL((k<CLS, true, false, PRIO>), dim3(256), dim3(512), 0, 0, out, src, iters, cyc);
        if (mode == 1) hipLaunchKernelGGL((k<CLS, false, true, PRIO>), dim3(256), dim3(512), 0, 0, out, src, iters, cyc);
        if (mode == 2) hipLaunchKernelGGL((k<CLS, true, true, PRIO>), dim3(256), dim3(512), 0, 0, out, src, iters, cyc);
        hipDeviceSynchronize(); hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        m[mode] = v[mode] = 0; for (int i = 0; i < 256; ++i) { m[mode] += h[2 * i]; v[mode] += h[2 * i + 1]; } m[mode] /= 256; v[mode] /= 256;
    }
    const double mf = (double)iters * 32, si = (double)(iters / 4) * 32;
    printf("%-28s side alone %6.1f cycles/instr | beside MFMA %6.1f cycles/instr (x%.1f) | MFMA alone %.1f, beside %.1f cycles each\n", tag,
           v[1] / si, v[2] / si, v[2] / v[1], m[0] / mf, m[2] / mf);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4); long long* cyc; hipMalloc(&cyc, 256 * 2 * 8);
    float* src; hipMalloc(&src, 64 * 4096 * 16 + 65536); hipMemset(src, 0, 64 * 4096 * 16 + 65536);
    const int iters = 2000;
    printf("-- side waves at the default priority (they are the YOUNGER waves of the workgroup)\n");
    run<0, 0>("v_fma_f32", out, src, cyc, iters);
    run<1, 0>("v_readlane_b32", out, src, cyc, iters);
    run<2, 0>("s_add_i32", out, src, cyc, iters);
    run<3, 0>("ds_write_b128", out, src, cyc, iters);
    run<4, 0>("ds_read_b128", out, src, cyc, iters);
    run<5, 0>("global_load_dwordx4 (L2)", out, src, cyc, iters);
    printf("-- side waves at s_setprio 3\n");
    run<0, 3>("v_fma_f32", out, src, cyc, iters);
    run<1, 3>("v_readlane_b32", out, src, cyc, iters);
    run<2, 3>("s_add_i32", out, src, cyc, iters);
    run<3, 3>("ds_write_b128", out, src, cyc, iters);
    run<4, 3>("ds_read_b128", out, src, cyc, iters);
    run<5, 3>("global_load_dwordx4 (L2)", out, src, cyc, iters);
    return 0;
}

// Step-by-step approach from the bare LDS+MFMA loop (0.97 of peak) to the structure of the grouped kernels
// (128x128 tile, BK = 32, register-staged double buffering): which ingredient costs the ~20 %?
//   mode 0: ds_read + MFMA + barrier
//   mode 1: + 8 ds_write_b128 per thread per chunk (register data) into the other buffer
//   mode 2: + 8 global_load_dwordx4 per thread per chunk (issued before the MFMAs, waited before the writes),
//           operand rows K-contiguous like in gram.hip; footprint chosen by argv (L2-resident or HBM-sized)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kLds = 36, TILE = 128;

template <int MODE>
__global__ __launch_bounds__(256) void loop_kernel(const float* __restrict__ src, float* out, int chunks, int rows_total, int krow) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 4 * TILE * kLds; i += 256) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // staging: thread owns k-columns (tid % 8) * 4 of rows tid / 8 + 32 q of both operands
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    f32x4 ra[4], rb[4];
    for (int q = 0; q < 4; ++q) { ra[q] = f32x4{1.f, 2.f, 3.f, 4.f}; rb[q] = f32x4{.5f, .25f, .125f, 1.f}; }
    const int row0 = (int)((blockIdx.x * 2u * TILE) % (unsigned)(rows_total - 2 * TILE));
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        if (MODE >= 2) {
            const int k0 = (c * 32) % (krow - 32) / 4 * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ra[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + srow + 32 * q) * krow + k0 + scol);
                rb[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + TILE + srow + 32 * q) * krow + k0 + scol);
            }
        }
        const float* a = smem + buf * TILE * kLds + (wm * 64 + (lane & 31)) * kLds + 4 * (lane >> 5);
        const float* b = smem + (2 + buf) * TILE * kLds + (wn * 64 + (lane & 31)) * kLds + 4 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * kLds + kk * 8);
                fb[s] = *reinterpret_cast<const f32x4*>(b + s * 32 * kLds + kk * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
        }
        if (MODE >= 1) {
            float* wa = smem + (buf ^ 1) * TILE * kLds;
            float* wb = smem + (2 + (buf ^ 1)) * TILE * kLds;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<f32x4*>(wa + (srow + 32 * q) * kLds + scol) = ra[q];
                *reinterpret_cast<f32x4*>(wb + (srow + 32 * q) * kLds + scol) = rb[q];
            }
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}


// mode 3: same work as mode 2, but in the "write after the barrier" order: barrier -> ds_write chunk c+1 (loaded during the
// previous iteration) -> issue loads of chunk c+2 -> MFMAs of chunk c.  The LDS writes and the load issue sit behind
// the barrier instead of in front of it, and the writes' latency is covered by the MFMAs (they target the other buffer).
__global__ __launch_bounds__(256) void loop_kernel_wab(const float* __restrict__ src, float* out, int chunks, int rows_total, int krow) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 4 * TILE * kLds; i += 256) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    f32x4 ra[4], rb[4];
    const int row0 = (int)((blockIdx.x * 2u * TILE) % (unsigned)(rows_total - 2 * TILE));
    auto load = [&](int c) {
        const int k0 = (c * 32) % (krow - 32) / 4 * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + srow + 32 * q) * krow + k0 + scol);
            rb[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + TILE + srow + 32 * q) * krow + k0 + scol);
        }
    };
    load(1);
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        __syncthreads();
        {
            float* wa = smem + (buf ^ 1) * TILE * kLds;
            float* wb = smem + (2 + (buf ^ 1)) * TILE * kLds;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<f32x4*>(wa + (srow + 32 * q) * kLds + scol) = ra[q];
                *reinterpret_cast<f32x4*>(wb + (srow + 32 * q) * kLds + scol) = rb[q];
            }
        }
        load(c + 2);
        const float* a = smem + buf * TILE * kLds + (wm * 64 + (lane & 31)) * kLds + 4 * (lane >> 5);
        const float* b = smem + (2 + buf) * TILE * kLds + (wn * 64 + (lane & 31)) * kLds + 4 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * kLds + kk * 8);
                fb[s] = *reinterpret_cast<const f32x4*>(b + s * 32 * kLds + kk * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    s += ra[0][0] + rb[0][0];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

static void run_wab(const char* tag, const float* src, int rows_total, int krow) {
    const int chunks = 512, grid = 256 * 2 * 8;
    const size_t lds = 76 * 1024;
    hipFuncSetAttribute((const void*)loop_kernel_wab, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float* out; hipMalloc(&out, (size_t)grid * 256 * 4);
    hipLaunchKernelGGL(loop_kernel_wab, dim3(grid), dim3(256), lds, 0, src, out, chunks, rows_total, krow);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(loop_kernel_wab, dim3(grid), dim3(256), lds, 0, src, out, chunks, rows_total, krow);
    hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double flop = (double)grid * 4 * chunks * 64 * 4096.0;
    printf("%-62s %.2f ms -> %.1f TF/s (%.3f of 157.3)\n", tag, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
    hipFree(out);
}

// mode 4 (round 3): PRODUCER / CONSUMER waves.  512 threads: waves 0-3 only read LDS and issue MFMAs (the 2 x 2 wave grid of
// the 128 x 128 tile, as before), waves 4-7 only stage: ds_write chunk c+1 (loaded during the previous iteration) into the
// other buffer, then issue the global loads of chunk c+2.  One barrier per chunk for all eight waves.  Two workgroups per CU
// = four waves per SIMD, i.e. <= 128 VGPRs per wave (the consumers' 64 accumulators + fragments fit; the producers need ~40).
// Question: does taking every VMEM / ds_write / address instruction out of the MFMA waves' streams lift the loop?
__global__ __launch_bounds__(512, 4) void loop_kernel_pc(const float* __restrict__ src, float* out, int chunks, int rows_total, int krow) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, t = tid & 255, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const bool producer = tid >= 256;
    for (int i = tid; i < 4 * TILE * kLds; i += 512) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
    const int row0 = (int)((blockIdx.x * 2u * TILE) % (unsigned)(rows_total - 2 * TILE));
    float s = 0.f;
    if (producer) {
        const int srow = t >> 3, scol = (t & 7) * 4;
        f32x4 ra[4], rb[4];
        auto load = [&](int c) {
            const int k0 = (c * 32) % (krow - 32) / 4 * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ra[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + srow + 32 * q) * krow + k0 + scol);
                rb[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + TILE + srow + 32 * q) * krow + k0 + scol);
            }
        };
        load(1);
        for (int c = 0; c < chunks; ++c) {
            const int buf = c & 1;
            __syncthreads();
            float* wa = smem + (buf ^ 1) * TILE * kLds;
            float* wb = smem + (2 + (buf ^ 1)) * TILE * kLds;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<f32x4*>(wa + (srow + 32 * q) * kLds + scol) = ra[q];
                *reinterpret_cast<f32x4*>(wb + (srow + 32 * q) * kLds + scol) = rb[q];
            }
            load(c + 2);
        }
        s = ra[0][0] + rb[0][0];
    } else {
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int c = 0; c < chunks; ++c) {
            const int buf = c & 1;
            __syncthreads();
            const float* a = smem + buf * TILE * kLds + (wm * 64 + (lane & 31)) * kLds + 4 * (lane >> 5);
            const float* b = smem + (2 + buf) * TILE * kLds + (wn * 64 + (lane & 31)) * kLds + 4 * (lane >> 5);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f32x4 fa[2], fb[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    fa[q] = *reinterpret_cast<const f32x4*>(a + q * 32 * kLds + kk * 8);
                    fb[q] = *reinterpret_cast<const f32x4*>(b + q * 32 * kLds + kk * 8);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                        for (int sn = 0; sn < 2; ++sn)
                            acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
            }
        }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    }
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

static void run_pc(const char* tag, const float* src, int rows_total, int krow) {
    const int chunks = 512, grid = 256 * 2 * 8;
    const size_t lds = 76 * 1024;
    hipFuncSetAttribute((const void*)loop_kernel_pc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float* out; hipMalloc(&out, (size_t)grid * 512 * 4);
    hipLaunchKernelGGL(loop_kernel_pc, dim3(grid), dim3(512), lds, 0, src, out, chunks, rows_total, krow);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(loop_kernel_pc, dim3(grid), dim3(512), lds, 0, src, out, chunks, rows_total, krow);
    hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double flop = (double)grid * 4 * chunks * 64 * 4096.0;
    printf("%-62s %.2f ms -> %.1f TF/s (%.3f of 157.3)\n", tag, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
    hipFree(out);
}

template <int MODE>
static void run(const char* tag, const float* src, int rows_total, int krow) {
    const int chunks = 512, grid = 256 * 2 * 8;
    const size_t lds = 76 * 1024;
    hipFuncSetAttribute((const void*)loop_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float* out; hipMalloc(&out, (size_t)grid * 256 * 4);
    hipLaunchKernelGGL((loop_kernel<MODE>), dim3(grid), dim3(256), lds, 0, src, out, chunks, rows_total, krow);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((loop_kernel<MODE>), dim3(grid), dim3(256), lds, 0, src, out, chunks, rows_total, krow);
    hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double flop = (double)grid * 4 * chunks * 64 * 4096.0;
    printf("%-62s %.2f ms -> %.1f TF/s (%.3f of 157.3)\n", tag, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
    hipFree(out);
}

int main() {
    const int krow = 3136;
    float* small; hipMalloc(&small, (size_t)1024 * krow * 4); hipMemset(small, 0, (size_t)1024 * krow * 4);       // 12.8 MB
    float* big; hipMalloc(&big, (size_t)65536 * krow * 4); hipMemset(big, 0, (size_t)65536 * krow * 4);            // 822 MB
    run<0>("mode 0: ds_read + MFMA + barrier", small, 1024, krow);
    run<1>("mode 1: + ds_write of the next chunk", small, 1024, krow);
    run<2>("mode 2: + global loads (12.8 MB footprint: L2 / MALL)", small, 1024, krow);
    run<2>("mode 2: + global loads (822 MB footprint: HBM)", big, 65536, krow);
    run_wab("mode 3: write-after-barrier order (12.8 MB footprint)", small, 1024, krow);
    run_wab("mode 3: write-after-barrier order (822 MB footprint: HBM)", big, 65536, krow);
    run_pc("mode 4: producer / consumer waves, 512 threads (12.8 MB footprint)", small, 1024, krow);
    run_pc("mode 4: producer / consumer waves, 512 threads (822 MB footprint: HBM)", big, 65536, krow);
    run<2>("mode 2 again (12.8 MB)", small, 1024, krow);
    run_wab("mode 3 again (12.8 MB)", small, 1024, krow);
    return 0;
}

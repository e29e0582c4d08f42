// Round 4: does loading the next chunk STRAIGHT INTO LDS (global_load_lds_dwordx4: no staging registers, no ds_write, no
// wait-then-write in the MFMA waves' instruction streams) lift the 128x128 / BK 32 tile loop of the grouped kernels?
//   ref : register-staged double buffering as in gram.hip (mode 2 of mfma_loop2.hip), row stride 36 floats
//   dma : 8 global_load_lds_dwordx4 per thread per chunk into the other buffer, rows of 32 floats (no padding: a wave's 64
//         16-byte granules land back to back) with the granule index XOR-swizzled by the row so that ds_read_b128 stays
//         conflict free; s_waitcnt vmcnt(0) + one barrier per chunk
//   dma4: the same with 16-deep chunks and FOUR buffers (loads run three chunks ahead)
// The layout is verified first (one chunk, LDS dumped and compared on the host).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TILE = 128;
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// granule (16 bytes) of (row r, k slot s) in a [128][32]-float operand buffer
__device__ __forceinline__ int gran(int r, int s) { return r * 8 + (s ^ ((r >> 1) & 7)); }

// one operand buffer = 128 rows x 32 floats = 1024 granules = 16 wave-instructions; wave w of 4 issues instructions 4 j + w
__device__ __forceinline__ void dma_chunk(const float* src, int krow, int row0, int k0, float* lds_a, float* lds_b, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = (4 * j + wave) * 64 + lane;          // granule this lane fills
        const int r = g >> 3, s = (g & 7) ^ ((r >> 1) & 7);
        const float* ga = src + (size_t)(row0 + r) * krow + k0 + 4 * s;
        const float* gb = src + (size_t)(row0 + TILE + r) * krow + k0 + 4 * s;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ga), LDS_PTR(lds_a + (4 * j + wave) * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gb), LDS_PTR(lds_b + (4 * j + wave) * 256), 16, 0, 0);
    }
}

__global__ __launch_bounds__(256) void dma_check(const float* __restrict__ src, float* out, int krow) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dma_chunk(src, krow, 0, 32, smem, smem + TILE * 32, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 2 * TILE * 32; i += 256) out[i] = smem[i];
}

template <bool DMA>
__global__ __launch_bounds__(256, 2) void loop_kernel(const float* __restrict__ src, float* out, int chunks, int rows_total, int krow) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDW = DMA ? 32 : 36;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 4 * TILE * LDW; i += 256) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    f32x4 ra[4], rb[4];
    const int row0 = (int)((blockIdx.x * 2u * TILE) % (unsigned)(rows_total - 2 * TILE));
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        const int k0 = (c * 32) % (krow - 32) / 4 * 4;
        if constexpr (DMA) {
            dma_chunk(src, krow, row0, k0, smem + (buf ^ 1) * TILE * LDW, smem + (2 + (buf ^ 1)) * TILE * LDW, wave, lane);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ra[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + srow + 32 * q) * krow + k0 + scol);
                rb[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + TILE + srow + 32 * q) * krow + k0 + scol);
            }
        }
        const float* a = smem + buf * TILE * LDW;
        const float* b = smem + (2 + buf) * TILE * LDW;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ra_ = wm * 64 + s * 32 + (lane & 31), rb_ = wn * 64 + s * 32 + (lane & 31), q = 2 * kk + (lane >> 5);
                if constexpr (DMA) {
                    fa[s] = *reinterpret_cast<const f32x4*>(a + 4 * gran(ra_, q));
                    fb[s] = *reinterpret_cast<const f32x4*>(b + 4 * gran(rb_, q));
                } else {
                    fa[s] = *reinterpret_cast<const f32x4*>(a + ra_ * LDW + 4 * q);
                    fb[s] = *reinterpret_cast<const f32x4*>(b + rb_ * LDW + 4 * q);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
        }
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            float* wa = smem + (buf ^ 1) * TILE * LDW;
            float* wb = smem + (2 + (buf ^ 1)) * TILE * LDW;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<f32x4*>(wa + (srow + 32 * q) * LDW + scol) = ra[q];
                *reinterpret_cast<f32x4*>(wb + (srow + 32 * q) * LDW + scol) = rb[q];
            }
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

// four 16-deep stages: a stage = two operands x 128 rows x 16 floats (4 granules per row, swizzled by (r >> 2) & 3 ... rows of
// 64 bytes: 16 consecutive rows x one k slot must hit 16 different 4-bank groups: granule % 16 = (r & 3) * 4 + (s ^ ((r >> 2) & 3)))
__device__ __forceinline__ int gran16(int r, int s) { return r * 4 + (s ^ ((r >> 2) & 3)); }
__global__ __launch_bounds__(256, 2) void loop_kernel_dma4(const float* __restrict__ src, float* out, int chunks, int rows_total, int krow) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STG = 2 * TILE * 16;    // floats per stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 4 * STG; i += 256) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int row0 = (int)((blockIdx.x * 2u * TILE) % (unsigned)(rows_total - 2 * TILE));
    auto issue = [&](int c) {      // stage c & 3 <- chunk c: 512 granules per operand = 8 instructions, 2 per wave per operand
        float* st = smem + (c & 3) * STG;
        const int k0 = (c * 16) % (krow - 16) / 4 * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int g = (2 * wave + j) * 64 + lane;
            const int r = g >> 2, s = (g & 3) ^ ((r >> 2) & 3);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + (size_t)(row0 + r) * krow + k0 + 4 * s), LDS_PTR(st + (2 * wave + j) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + (size_t)(row0 + TILE + r) * krow + k0 + 4 * s), LDS_PTR(st + TILE * 16 + (2 * wave + j) * 256), 16, 0, 0);
        }
    };
    issue(1); issue(2);
    for (int c = 0; c < 2 * chunks; ++c) {
        issue(c + 3);
        const float* a = smem + (c & 3) * STG;
        const float* b = a + TILE * 16;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ra_ = wm * 64 + s * 32 + (lane & 31), rb_ = wn * 64 + s * 32 + (lane & 31), q = 2 * kk + (lane >> 5);
                fa[s] = *reinterpret_cast<const f32x4*>(a + 4 * gran16(ra_, q));
                fb[s] = *reinterpret_cast<const f32x4*>(b + 4 * gran16(rb_, q));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // chunk c+1 has landed (c+2, c+3: 4 loads each still in flight)
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <class K>
static void run(const char* tag, K kern, size_t lds, const float* src, int rows_total, int krow) {
    const int chunks = 512, grid = 256 * 2 * 8;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float* out; hipMalloc(&out, (size_t)grid * 256 * 4);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, out, chunks, rows_total, krow);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", tag); exit(1); }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, out, chunks, rows_total, krow);
    hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double flop = (double)grid * 4 * chunks * 64 * 4096.0;
    printf("%-72s %.2f ms -> %.1f TF/s (%.3f of 157.3)\n", tag, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
    hipFree(out);
}

int main() {
    const int krow = 3136;
    {   // layout check
        std::vector<float> h((size_t)256 * krow);
        for (int r = 0; r < 256; ++r) for (int k = 0; k < krow; ++k) h[(size_t)r * krow + k] = (float)(r * 4096 + k);
        float *d, *o; hipMalloc(&d, h.size() * 4); hipMalloc(&o, 2 * TILE * 32 * 4);
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipFuncSetAttribute((const void*)dma_check, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        hipLaunchKernelGGL(dma_check, dim3(1), dim3(256), 64 * 1024, 0, d, o, krow);
        std::vector<float> got(2 * TILE * 32);
        hipMemcpy(got.data(), o, got.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int op = 0; op < 2; ++op) for (int r = 0; r < TILE; ++r) for (int s = 0; s < 8; ++s) for (int e = 0; e < 4; ++e) {
            const int g = r * 8 + (s ^ ((r >> 1) & 7));
            const float want = (float)((op * TILE + r) * 4096 + 32 + 4 * s + e);
            if (got[(size_t)op * TILE * 32 + 4 * g + e] != want && bad++ < 5) printf("  layout mismatch op %d row %d slot %d e %d: %.0f != %.0f\n", op, r, s, e, got[(size_t)op * TILE * 32 + 4 * g + e], want);
        }
        printf("global_load_lds_dwordx4 layout check: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
        hipFree(d); hipFree(o);
        if (bad) return 1;
    }
    float* small; hipMalloc(&small, (size_t)1024 * krow * 4); hipMemset(small, 0, (size_t)1024 * krow * 4);       // 12.8 MB
    float* big; hipMalloc(&big, (size_t)65536 * krow * 4); hipMemset(big, 0, (size_t)65536 * krow * 4);            // 822 MB
    for (int rep = 0; rep < 2; ++rep) {
        run("ref : register-staged double buffer (12.8 MB footprint: L2 / MALL)", loop_kernel<false>, 76 * 1024, small, 1024, krow);
        run("ref : register-staged double buffer (822 MB footprint: HBM)", loop_kernel<false>, 76 * 1024, big, 65536, krow);
        run("dma : global_load_lds x4, two 32-deep buffers (12.8 MB)", loop_kernel<true>, 64 * 1024, small, 1024, krow);
        run("dma : global_load_lds x4, two 32-deep buffers (822 MB: HBM)", loop_kernel<true>, 64 * 1024, big, 65536, krow);
        run("dma4: global_load_lds x4, four 16-deep stages (12.8 MB)", loop_kernel_dma4, 64 * 1024, small, 1024, krow);
        run("dma4: global_load_lds x4, four 16-deep stages (822 MB: HBM)", loop_kernel_dma4, 64 * 1024, big, 65536, krow);
    }
    return 0;
}

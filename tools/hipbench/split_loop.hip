// Round 5: what can the split-bf16 arithmetic reach on the NT tile loop (128 x 128 tile, BK = 32, 4 waves), before it goes
// into the product kernels?  fp32-EQUIVALENT flop (2 * 128 * 128 * 32 per chunk and workgroup) per second, operands streamed
// from an HBM-sized buffer exactly as gram.hip / conv.hip read them (rows K-contiguous, 16-byte loads).
//   mode 0  fp32 MFMA 32x32x2, register-staged double buffering (today's exact tile)
//   mode 1  fp32 in HBM, split into three bf16 planes when the chunk goes to LDS; ONE LDS image, two barriers per chunk
//           (the round-3 study structure of gram.hip)
//   mode 2  as 1 with TWO LDS images (one barrier per chunk; the conversions of chunk c+1 may interleave with the MFMAs of c)
//   mode 3  planes already in HBM (bf16 [plane][row][k], what a producer would emit): no conversions; ONE LDS image
//   mode 4  as 3 with TWO LDS images
//   mode 5  planes in HBM CHUNK-MAJOR and interleaved, [k / 32][row][plane][32]: a tile's chunk is 24 KB contiguous (what a producer that
//           knows its consumer would write); ONE LDS image;  mode 6  as 5 with TWO images
// usage: split_loop <mode> <chunks per split> <splits> <C> <krow>   (grid = (C / 128)^2 tiles x splits)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int kLds = 36, TILE = 128, kSplitLd = 104;   // bf16 per row of the split image: 3 planes x 32 + 8 pad (208 B)

__device__ __forceinline__ uint32_t bf16_pair(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t (&p)[3]) {
    p[0] = bf16_pair(a, b);
    const float ra = a - __builtin_bit_cast(float, p[0] << 16), rb = b - __builtin_bit_cast(float, p[0] & 0xffff0000u);
    p[1] = bf16_pair(ra, rb);
    p[2] = bf16_pair(ra - __builtin_bit_cast(float, p[1] << 16), rb - __builtin_bit_cast(float, p[1] & 0xffff0000u));
}

template <int MODE>
__global__ __launch_bounds__(256, MODE == 0 || MODE == 1 || MODE == 3 || MODE == 5 ? 2 : 1)
void loop_kernel(const float* __restrict__ src, float* out, int chunks, int rows_total, int krow) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // workgroup = (tile of a C x C product, K split): tiles of one split share their operand rows through L2, as in the grouped
    // launches (rows_total = 2 C: operand A rows 0 .. C-1, operand B rows C .. 2C-1; every split walks `chunks` chunks of its own)
    const int C = rows_total / 2, T = C / TILE;
    const int tile = blockIdx.x % (T * T), split = blockIdx.x / (T * T);
    const int rowA = (tile / T) * TILE, rowB = C + (tile % T) * TILE;
    const int cbase0 = split * chunks;
    if constexpr (MODE == 0) {
        const int srow = tid >> 3, scol = (tid & 7) * 4;
        f32x4 ra[4], rb[4];
        auto load = [&](int c) {
            const int k0 = ((cbase0 + c) * 32) % (krow - 32) / 4 * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ra[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(rowA + srow + 32 * q) * krow + k0 + scol);
                rb[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(rowB + srow + 32 * q) * krow + k0 + scol);
            }
        };
        auto store = [&](int buf) {
            float* wa = smem + buf * TILE * kLds;
            float* wb = smem + (2 + buf) * TILE * kLds;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<f32x4*>(wa + (srow + 32 * q) * kLds + scol) = ra[q];
                *reinterpret_cast<f32x4*>(wb + (srow + 32 * q) * kLds + scol) = rb[q];
            }
        };
        load(0); store(0); __syncthreads();
        for (int c = 0; c < chunks; ++c) {
            const int buf = c & 1;
            load(c + 1);
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
            store(buf ^ 1);
            __syncthreads();
        }
    } else {
        constexpr bool PLANES = MODE >= 3, DOUBLE = MODE == 2 || MODE == 4 || MODE == 6, CHUNKED = MODE >= 5;
        __bf16* img = reinterpret_cast<__bf16*>(smem);            // [buffers][2 operands][TILE][kSplitLd]
        constexpr int kImg = 2 * TILE * kSplitLd;
        // fp32 source: thread owns k-columns (tid % 8) * 4 of rows tid / 8 + 32 q; planes source: 8 k of plane p of row r per
        // 16-byte load: a chunk of one operand is 128 rows x 3 planes x 4 loads = 1536 loads = 6 per thread
        const int srow = tid >> 3, scol = (tid & 7) * 4;
        f32x4 ra[4], rb[4];
        u32x4 pa[6], pb[6];
        const __bf16* psrc = reinterpret_cast<const __bf16*>(src);
        const size_t plane_elems = (size_t)rows_total * krow;      // bf16 elements per plane (the buffer holds 3 planes in 1.5x the bytes)
        auto load = [&](int c) {
            const int k0 = ((cbase0 + c) * 32) % (krow - 32) / 8 * 8;
            if constexpr (!PLANES) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ra[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(rowA + srow + 32 * q) * krow + k0 + scol);
                    rb[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(rowB + srow + 32 * q) * krow + k0 + scol);
                }
            } else if constexpr (CHUNKED) {
                const size_t cbase = (size_t)(((cbase0 + c) % (krow / 32)) * rows_total) * 96;      // bf16 elements
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    pa[i] = *reinterpret_cast<const u32x4*>(psrc + cbase + (size_t)rowA * 96 + (size_t)(tid + 256 * i) * 8);
                    pb[i] = *reinterpret_cast<const u32x4*>(psrc + cbase + (size_t)rowB * 96 + (size_t)(tid + 256 * i) * 8);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int id = tid + 256 * i;                  // 0 .. 1535: (plane, row, k-octet)
                    const int oct = id & 3, row = (id >> 2) & 127, pl = id >> 9;
                    pa[i] = *reinterpret_cast<const u32x4*>(psrc + pl * plane_elems + (size_t)(rowA + row) * krow + k0 + 8 * oct);
                    pb[i] = *reinterpret_cast<const u32x4*>(psrc + pl * plane_elems + (size_t)(rowB + row) * krow + k0 + 8 * oct);
                }
            }
        };
        auto store = [&](int buf) {
            __bf16* A = img + buf * kImg;
            __bf16* B = A + TILE * kSplitLd;
            if constexpr (!PLANES) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t lo[3], hi[3];
                    split3_pair(ra[q][0], ra[q][1], lo);
                    split3_pair(ra[q][2], ra[q][3], hi);
#pragma unroll
                    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(A + (srow + 32 * q) * kSplitLd + p * 32 + scol) = u32x2{lo[p], hi[p]};
                    split3_pair(rb[q][0], rb[q][1], lo);
                    split3_pair(rb[q][2], rb[q][3], hi);
#pragma unroll
                    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(B + (srow + 32 * q) * kSplitLd + p * 32 + scol) = u32x2{lo[p], hi[p]};
                }
            } else if constexpr (CHUNKED) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int e = (tid + 256 * i) * 8, row = e / 96, col = e - row * 96;
                    *reinterpret_cast<u32x4*>(A + row * kSplitLd + col) = pa[i];
                    *reinterpret_cast<u32x4*>(B + row * kSplitLd + col) = pb[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int id = tid + 256 * i;
                    const int oct = id & 3, row = (id >> 2) & 127, pl = id >> 9;
                    *reinterpret_cast<u32x4*>(A + row * kSplitLd + pl * 32 + 8 * oct) = pa[i];
                    *reinterpret_cast<u32x4*>(B + row * kSplitLd + pl * 32 + 8 * oct) = pb[i];
                }
            }
        };
        auto compute = [&](int buf) {
            const __bf16* a16 = img + buf * kImg + (wm * 64 + (lane & 31)) * kSplitLd + 8 * (lane >> 5);
            const __bf16* b16 = img + buf * kImg + TILE * kSplitLd + (wn * 64 + (lane & 31)) * kSplitLd + 8 * (lane >> 5);
#pragma unroll
            for (int g16 = 0; g16 < 2; ++g16) {
                bf16x8 sa[2][3], sb[2][3];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        sa[s][p] = *reinterpret_cast<const bf16x8*>(a16 + s * 32 * kSplitLd + p * 32 + g16 * 16);
                        sb[s][p] = *reinterpret_cast<const bf16x8*>(b16 + s * 32 * kSplitLd + p * 32 + g16 * 16);
                    }
#pragma unroll
                for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn) {
                        f32x16 c = acc[sm][sn];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][0], sb[sn][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][2], sb[sn][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][1], sb[sn][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][0], sb[sn][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][1], sb[sn][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][0], sb[sn][0], c, 0, 0, 0);
                        acc[sm][sn] = c;
                    }
            }
        };
        load(0); store(0); __syncthreads();
        for (int c = 0; c < chunks; ++c) {
            load(c + 1);
            if constexpr (DOUBLE) {
                compute(c & 1);
                store((c & 1) ^ 1);
                __syncthreads();
            } else {
                compute(0);
                __syncthreads();
                store(0);
                __syncthreads();
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE>
static int run(const float* src, float* out, int chunks, int wgs, int rows_total, int krow) {
    const size_t lds = MODE == 0 ? (size_t)4 * TILE * kLds * 4 : (size_t)((MODE == 2 || MODE == 4 || MODE == 6) ? 2 : 1) * 2 * TILE * kSplitLd * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(loop_kernel<MODE>, dim3(wgs), dim3(256), lds, 0, src, out, chunks, rows_total, krow);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, a, b));
        const double flop = 2.0 * TILE * TILE * 32.0 * chunks * wgs;
        printf("mode %d lds %zu B wgs %d chunks %d: %.3f ms, %.1f fp32-equivalent TFLOP/s (%.2f of the 157.3 fp32 matrix peak)\n",
               MODE, lds, wgs, chunks, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
    }
    return 0;
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, chunks = argc > 2 ? atoi(argv[2]) : 98, splits = argc > 3 ? atoi(argv[3]) : 16;
    const int Cside = argc > 4 ? atoi(argv[4]) : 1024, krow = argc > 5 ? atoi(argv[5]) : 50176;
    const int rows_total = 2 * Cside, wgs = (Cside / TILE) * (Cside / TILE) * splits;
    float *src, *out;
    const size_t n = (size_t)rows_total * krow;
    CK(hipMalloc(&src, n * 4 * 3 / 2 + 1024));      // fp32 rows, or three bf16 planes of the same shape
    CK(hipMalloc(&out, (size_t)wgs * 256 * 4));
    CK(hipMemset(src, 0x3c, n * 4 * 3 / 2 + 1024));    // 0x3c3c3c3c: a small finite float / bf16 pattern
    switch (mode) {
        case 0: return run<0>(src, out, chunks, wgs, rows_total, krow);
        case 1: return run<1>(src, out, chunks, wgs, rows_total, krow);
        case 2: return run<2>(src, out, chunks, wgs, rows_total, krow);
        case 3: return run<3>(src, out, chunks, wgs, rows_total, krow);
        case 4: return run<4>(src, out, chunks, wgs, rows_total, krow);
        case 5: return run<5>(src, out, chunks, wgs, rows_total, krow);
        case 6: return run<6>(src, out, chunks, wgs, rows_total, krow);
    }
    return 2;
}

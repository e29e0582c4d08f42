// Ceiling of the tile's inner loop on gfx950: LDS fragment reads (ds_read_b128, row stride 36) + v_mfma_f32_32x32x2_f32
// + one barrier per K chunk, no global memory at all.  Sweeps workgroups per CU (via the dynamic LDS size) and the
// number of MFMAs between two barriers, to tell what limits the grouped kernels' ~0.73 of the fp32 matrix peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kLds = 36;

template <int MT, int KK, bool BARRIER>   // wave tile (32 MT)^2, KK steps of 8 k per chunk
__global__ __launch_bounds__(256) void loop_kernel(float* out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    constexpr int TILE = 64 * MT;
    for (int i = tid; i < 4 * TILE * kLds; i += 256) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
    __syncthreads();
    f32x16 acc[MT][MT];
    for (int a = 0; a < MT; ++a) for (int b = 0; b < MT; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        const float* a = smem + buf * TILE * kLds + (wm * (TILE / 2) + (lane & 31)) * kLds + 4 * (lane >> 5);
        const float* b = smem + (2 + buf) * TILE * kLds + (wn * (TILE / 2) + (lane & 31)) * kLds + 4 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            f32x4 fa[MT], fb[MT];
#pragma unroll
            for (int s = 0; s < MT; ++s) {
                fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * kLds + (kk & 3) * 8);
                fb[s] = *reinterpret_cast<const f32x4*>(b + s * 32 * kLds + (kk & 3) * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sm = 0; sm < MT; ++sm)
#pragma unroll
                    for (int sn = 0; sn < MT; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
        }
        if (BARRIER) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < MT; ++a) for (int b = 0; b < MT; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MT, int KK, bool BARRIER>
static void run(int wg_per_cu, const char* tag) {
    const int chunks = 4096 / KK, grid = 256 * wg_per_cu * 4;
    size_t lds = (size_t)(160 * 1024 / wg_per_cu) / 1024 * 1024;
    if (lds > 64 * 1024) lds = 64 * 1024 + (wg_per_cu == 1 ? 60 * 1024 : (wg_per_cu == 2 ? 12 * 1024 : 0));
    const size_t need = (size_t)4 * 64 * MT * kLds * sizeof(float);
    if (lds < need) { printf("%-28s wg/CU=%d: tile does not fit\n", tag, wg_per_cu); return; }
    hipFuncSetAttribute((const void*)loop_kernel<MT, KK, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float* out; hipMalloc(&out, (size_t)grid * 256 * 4);
    hipLaunchKernelGGL((loop_kernel<MT, KK, BARRIER>), dim3(grid), dim3(256), lds, 0, out, chunks);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((loop_kernel<MT, KK, BARRIER>), dim3(grid), dim3(256), lds, 0, out, chunks);
    hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double flop = (double)grid * 4 /*waves*/ * chunks * KK * 4 * MT * MT * 4096.0;
    printf("%-28s wg/CU=%d (LDS %3zu KB): %.2f ms -> %.1f TF/s (%.3f of 157.3)\n", tag, wg_per_cu, lds / 1024, ms, flop / (ms * 1e-3) / 1e12,
           flop / (ms * 1e-3) / 1e12 / 157.3);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 3, 4}) run<2, 4, true>(w, "128x128 tile, BK=32, barrier");
    for (int w : {1, 2, 3, 4}) run<2, 4, false>(w, "128x128 tile, BK=32, no barrier");
    for (int w : {2, 3, 4}) run<2, 2, true>(w, "128x128 tile, BK=16, barrier");
    for (int w : {2, 3, 4}) run<2, 8, true>(w, "128x128 tile, BK=64, barrier");
    for (int w : {2, 4}) run<1, 4, true>(w, "64x64 tile, BK=32, barrier");
    return 0;
}

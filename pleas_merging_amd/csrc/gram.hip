// Cross-feature contraction for activation / weight matching on gfx950 (MI355X).
//
//   G[i][j] = sum_{b,p} x[b][i][p] * y[b][j][p]          (fp32 MFMA, exact-fp32 products)
//   nx[i]   = sum_{b,p} x[b][i][p]^2 ,  ny[j] likewise     (fused into the staging pass)
//   acc[i][j] (+)= G                      (inner product)
//              or -sqrt(max(0, nx[i] + ny[j] - 2 G))       (negative Euclidean distance)
//
// Replaces the movedim/reshape copies + torch.cdist mm-path of the reference
// (pleas/methods/activation_matching.py:14-46) and the per-batch accumulation (:123-134).
//
// Layout: the [B][C][HW] operands are read in place (NCHW): for a fixed sample the
// C x HW slab has the contraction index contiguous for BOTH operands, so this is an
// NT GEMM whose K axis is the flattened (b, p) index.  A workgroup owns a TILE x TILE
// output tile and a contiguous range of K chunks (split-K); partial tiles go to the
// workspace and a second kernel reduces the slabs in a fixed order (deterministic),
// applies the epilogue and accumulates.
//
// Roofline: per node 2*C^2*K flop over 2*C*K*4 bytes = C/4 flop/B:
// fp32-MFMA-bound for C >= 128 (157 TFLOP/s), HBM-bound for C <= 64.
#include <algorithm>

#include "common.hpp"

namespace pleas {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 32;      // K chunk (floats) staged per step
constexpr int kLds = 36;     // padded LDS row stride: 16-B aligned rows, conflict-free ds_read_b128
constexpr int kThreads = 256;

struct GramGeom {
    const float* x;
    const float* y;
    float* gpart;  // [S][C][C]
    float* npart;  // [S][2][C]
    int C;
    uint32_t HW;
    uint32_t Ktot;  // B * HW
    int nchunks;
    int chunks_per_split;
    int tiles;  // tiles per matrix side
};

// One operand tile in flight between global memory and LDS.
template <int PASSES, int VEC>
struct Stage {
    float v[PASSES][VEC];
};

template <int TILE, int VEC>
__global__ __launch_bounds__(kThreads) void gram_partial_kernel(const GramGeom g) {
    constexpr int MT = TILE / 64;                    // 32x32 MFMA tiles per wave per side
    constexpr int LANES_PER_ROW = kBK / VEC;         // 8 (16-B loads) or 32 (4-B loads)
    constexpr int ROWS_PER_PASS = kThreads / LANES_PER_ROW;
    constexpr int PASSES = TILE / ROWS_PER_PASS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [2][TILE][kLds]
    float* Bs = smem + 2 * TILE * kLds;       // [2][TILE][kLds]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = blockIdx.x;
    const int tn = bid % g.tiles;
    bid /= g.tiles;
    const int tm = bid % g.tiles;
    const int split = bid / g.tiles;
    const int i0 = tm * TILE, j0 = tn * TILE;
    const int c_begin = split * g.chunks_per_split;
    const int c_end = min(c_begin + g.chunks_per_split, g.nchunks);

    const int srow = tid / LANES_PER_ROW;
    const int scol = (tid % LANES_PER_ROW) * VEC;

    Stage<PASSES, VEC> ra, rb;
    float sqa[PASSES], sqb[PASSES];
#pragma unroll
    for (int q = 0; q < PASSES; ++q) sqa[q] = sqb[q] = 0.f;
    f32x16 acc[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Row validity is fixed per thread; loads are UNCONDITIONAL from clamped (always valid) addresses
    // and masked when written to LDS, so all loads of a chunk stay in flight behind the MFMAs
    // (a predicated load makes hipcc branch and wait vmcnt(0) per load).
    unsigned rows_ok_a = 0, rows_ok_b = 0;
    size_t row_off_a[PASSES], row_off_b[PASSES];
#pragma unroll
    for (int q = 0; q < PASSES; ++q) {
        const int row = srow + q * ROWS_PER_PASS;
        const int gi = i0 + row, gj = j0 + row;
        if (gi < g.C) rows_ok_a |= 1u << q;
        if (gj < g.C) rows_ok_b |= 1u << q;
        row_off_a[q] = (size_t)min(gi, g.C - 1) * g.HW;
        row_off_b[q] = (size_t)min(gj, g.C - 1) * g.HW;
    }
    bool staged_kin = false;
    auto load_chunk = [&](int c) {
        const uint32_t k = (uint32_t)c * kBK + scol;
        const bool kin = k < g.Ktot;
        const uint32_t n = kin ? k / g.HW : 0u;
        const uint32_t p = kin ? k - n * g.HW : 0u;
        const size_t base = (size_t)n * g.C * g.HW + p;
        staged_kin = kin;
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            if constexpr (VEC == 4) {
                const f32x4 va = *reinterpret_cast<const f32x4*>(g.x + base + row_off_a[q]);
                const f32x4 vb = *reinterpret_cast<const f32x4*>(g.y + base + row_off_b[q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ra.v[q][e] = va[e];
                    rb.v[q][e] = vb[e];
                }
            } else {
                ra.v[q][0] = g.x[base + row_off_a[q]];
                rb.v[q][0] = g.y[base + row_off_b[q]];
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float* a = As + buf * TILE * kLds;
        float* b = Bs + buf * TILE * kLds;
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            const int row = srow + q * ROWS_PER_PASS;
            const bool oka = staged_kin && ((rows_ok_a >> q) & 1u);
            const bool okb = staged_kin && ((rows_ok_b >> q) & 1u);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                ra.v[q][e] = oka ? ra.v[q][e] : 0.f;
                rb.v[q][e] = okb ? rb.v[q][e] : 0.f;
            }
            if constexpr (VEC == 4) {
                f32x4 va = {ra.v[q][0], ra.v[q][1], ra.v[q][2], ra.v[q][3]};
                f32x4 vb = {rb.v[q][0], rb.v[q][1], rb.v[q][2], rb.v[q][3]};
                *reinterpret_cast<f32x4*>(a + row * kLds + scol) = va;
                *reinterpret_cast<f32x4*>(b + row * kLds + scol) = vb;
            } else {
                a[row * kLds + scol] = ra.v[q][0];
                b[row * kLds + scol] = rb.v[q][0];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                sqa[q] = fmaf(ra.v[q][e], ra.v[q][e], sqa[q]);
                sqb[q] = fmaf(rb.v[q][e], rb.v[q][e], sqb[q]);
            }
        }
    };
    auto compute = [&](int buf) {
        const float* a = As + buf * TILE * kLds + (wm * (TILE / 2) + (lane & 31)) * kLds + 4 * (lane >> 5);
        const float* b = Bs + buf * TILE * kLds + (wn * (TILE / 2) + (lane & 31)) * kLds + 4 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < kBK / 8; ++kk) {
            f32x4 fa[MT], fb[MT];
#pragma unroll
            for (int s = 0; s < MT; ++s) {
                fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * kLds + kk * 8);
                fb[s] = *reinterpret_cast<const f32x4*>(b + s * 32 * kLds + kk * 8);
            }
            // Lanes 0-31 feed k = 8kk+e, lanes 32-63 feed k = 8kk+4+e: any pairing of k is valid
            // as long as A and B agree, and it lets one 16-B LDS read serve four MFMA steps.
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sm = 0; sm < MT; ++sm)
#pragma unroll
                    for (int sn = 0; sn < MT; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
        }
    };

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
    }
    __syncthreads();
#ifndef PLEAS_GRAM_ABLATE
#define PLEAS_GRAM_ABLATE 0
#endif
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        const bool more = c + 1 < c_end;
        // ablation builds (tools/hipbench): 1 = no global loads, 2 = no MFMA, 3 = no LDS restage/barrier
        if (more && PLEAS_GRAM_ABLATE != 1 && PLEAS_GRAM_ABLATE != 3) load_chunk(c + 1);
        if (PLEAS_GRAM_ABLATE != 2) compute(PLEAS_GRAM_ABLATE == 3 ? 0 : buf);
        if (PLEAS_GRAM_ABLATE != 3) {
            if (more) store_chunk(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- partial tile -> workspace slab `split`
    float* gp = g.gpart + (size_t)split * g.C * g.C;
#pragma unroll
    for (int sm = 0; sm < MT; ++sm)
#pragma unroll
        for (int sn = 0; sn < MT; ++sn) {
            const int j = j0 + wn * (TILE / 2) + sn * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wm * (TILE / 2) + sm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (i < g.C && j < g.C) gp[(size_t)i * g.C + j] = acc[sm][sn][r];
            }
        }
    // ---- squared row norms of this K range (x rows from the tn == 0 column of blocks, y rows from tm == 0)
#pragma unroll
    for (int q = 0; q < PASSES; ++q) {
        float sa = sqa[q], sb = sqb[q];
#pragma unroll
        for (int off = 1; off < LANES_PER_ROW; off <<= 1) {
            sa += __shfl_xor(sa, off);
            sb += __shfl_xor(sb, off);
        }
        if ((tid % LANES_PER_ROW) == 0) {
            const int row = srow + q * ROWS_PER_PASS;
            if (tn == 0 && i0 + row < g.C) g.npart[((size_t)split * 2 + 0) * g.C + i0 + row] = sa;
            if (tm == 0 && j0 + row < g.C) g.npart[((size_t)split * 2 + 1) * g.C + j0 + row] = sb;
        }
    }
}

// Ordered reduction of the split-K slabs + epilogue + accumulation into the group matrix.
__global__ __launch_bounds__(256) void gram_finalize_kernel(const float* __restrict__ gpart,
                                                            const float* __restrict__ npart, float* __restrict__ acc,
                                                            int C, int S, int epilogue, int accumulate) {
    const size_t total = (size_t)C * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx / C), j = (int)(idx - (size_t)i * C);
        float gsum = 0.f, nx = 0.f, ny = 0.f;
        for (int s = 0; s < S; ++s) {
            gsum += gpart[(size_t)s * total + idx];
            nx += npart[((size_t)s * 2 + 0) * C + i];
            ny += npart[((size_t)s * 2 + 1) * C + j];
        }
        float v = gsum;
        if (epilogue == PLEAS_EPI_NEG_CDIST) v = -sqrtf(fmaxf(nx + ny - 2.f * gsum, 0.f));
        acc[idx] = accumulate ? acc[idx] + v : v;
    }
}

struct GramPlan {
    int tile, vec, tiles, nchunks, cps, S;
    size_t ws_bytes;
};

static int g_target_blocks = 512;
static int g_min_chunks = 2;

static GramPlan make_plan(int B, int C, int64_t HW, bool aligned) {
    GramPlan p;
    p.tile = C > 64 ? 128 : 64;
    p.vec = (HW % 4 == 0 && aligned) ? 4 : 1;
    p.tiles = (int)ceil_div(C, p.tile);
    const int64_t K = (int64_t)B * HW;
    p.nchunks = (int)ceil_div(K, kBK);
    int S = g_target_blocks / (p.tiles * p.tiles);
    if (S < 1) S = 1;
    int cap = p.nchunks / g_min_chunks;
    if (cap < 1) cap = 1;
    if (S > cap) S = cap;
    p.cps = (int)ceil_div(p.nchunks, S);
    p.S = (int)ceil_div(p.nchunks, p.cps);
    p.ws_bytes = (size_t)p.S * ((size_t)C * C + 2 * (size_t)C) * sizeof(float);
    return p;
}

}  // namespace pleas

using namespace pleas;

extern "C" void pleas_gram_tune(int target_blocks, int min_chunks_per_split) {
    if (target_blocks > 0) g_target_blocks = target_blocks;
    if (min_chunks_per_split > 0) g_min_chunks = min_chunks_per_split;
}

extern "C" size_t pleas_gram_ws_bytes(int B, int C, int64_t HW) {
    if (B <= 0 || C <= 0 || HW <= 0) return 0;
    return make_plan(B, C, HW, true).ws_bytes;  // alignment does not change the size
}

extern "C" int pleas_gram_accum(const float* x, const float* y, int B, int C, int64_t HW, int epilogue, int accumulate,
                                float* acc, void* ws, size_t ws_bytes, void* stream_) {
    if (!x || !y || !acc) return bad_arg("null tensor pointer");
    if (B <= 0 || C <= 0 || HW <= 0) return bad_arg("B, C, HW must be positive");
    if ((int64_t)B * HW >= (1ll << 31) || HW >= (1ll << 31)) return bad_arg("B*HW must be < 2^31");
    if (epilogue != PLEAS_EPI_INNER && epilogue != PLEAS_EPI_NEG_CDIST) return bad_arg("epilogue");
    const bool aligned = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    const GramPlan p = make_plan(B, C, HW, aligned);
    if (!ws || ws_bytes < p.ws_bytes) {
        std::snprintf(g_last_error, sizeof(g_last_error), "gram workspace too small: need %zu bytes", p.ws_bytes);
        return PLEAS_ENOMEM;
    }
    hipStream_t stream = (hipStream_t)stream_;
    GramGeom g;
    g.x = x;
    g.y = y;
    g.gpart = (float*)ws;
    g.npart = g.gpart + (size_t)p.S * C * C;
    g.C = C;
    g.HW = (uint32_t)HW;
    g.Ktot = (uint32_t)((int64_t)B * HW);
    g.nchunks = p.nchunks;
    g.chunks_per_split = p.cps;
    g.tiles = p.tiles;
    const dim3 grid((unsigned)(p.tiles * p.tiles * p.S));
    const size_t lds = (size_t)4 * p.tile * kLds * sizeof(float);
    const double kk = (double)B * (double)HW;
    {
    ProfScope prof(kProfGramPartial, 2.0 * C * (double)C * kk, 2.0 * C * kk * sizeof(float), stream);
    if (p.tile == 128 && p.vec == 4)
        hipLaunchKernelGGL((gram_partial_kernel<128, 4>), grid, dim3(kThreads), lds, stream, g);
    else if (p.tile == 128)
        hipLaunchKernelGGL((gram_partial_kernel<128, 1>), grid, dim3(kThreads), lds, stream, g);
    else if (p.vec == 4)
        hipLaunchKernelGGL((gram_partial_kernel<64, 4>), grid, dim3(kThreads), lds, stream, g);
    else
        hipLaunchKernelGGL((gram_partial_kernel<64, 1>), grid, dim3(kThreads), lds, stream, g);
    }
    PLEAS_LAUNCH_CHECK("gram_partial_kernel");
    ProfScope prof2(kProfGramFinalize, 0.0, ((double)p.S + 2.0) * C * (double)C * sizeof(float), stream);
    const size_t total = (size_t)C * C;
    const unsigned fgrid = (unsigned)std::min<size_t>(ceil_div((int64_t)total, 256), 2048);
    hipLaunchKernelGGL(gram_finalize_kernel, dim3(fgrid), dim3(256), 0, stream, g.gpart, g.npart, acc, C, p.S, epilogue,
                       accumulate);
    PLEAS_LAUNCH_CHECK("gram_finalize_kernel");
    return PLEAS_OK;
}

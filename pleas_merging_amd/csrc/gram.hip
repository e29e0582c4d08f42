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
#include <mutex>
#include <vector>

#include "common.hpp"

namespace pleas {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef f32x4 f32x4u __attribute__((aligned(4)));   // 16 bytes at a 4-byte-aligned address: still ONE global_load_dwordx4

constexpr int kBK = 32;      // K chunk (floats) staged per step
constexpr int kLds = 36;     // padded LDS row stride: 16-B aligned rows, conflict-free ds_read_b128
constexpr int kThreads = 256;

struct GramGeom {
    const float* x;
    const float* y;
    float* gpart;  // [S][C][C]
    float* npart;  // [S][2][C]
    float* spart;  // [S][2][C] row sums (only when `sums`): lets the reduce pass derive affine images of this node
    int sums;
    int C;
    uint32_t HW;
    uint32_t Ktot;  // B * HW
    int nchunks;
    int chunks_per_split;
    int tiles;  // tiles per matrix side
    // PAD form (images with HW % 4 != 0, e.g. the 7 x 7 maps of a ResNet's last stage): the K axis is (sample, pixel padded to
    // a multiple of four) -- a thread's four k never straddle two samples, both operands come through under-aligned 16-byte
    // loads instead of four 4-byte loads, and the padding pixels are stored as zeros
    uint32_t HWp, Kk;   // padded pixels per sample, B * HWp  (== HW, Ktot outside the PAD form)
    uint32_t bytes;     // 4 * B * C * HW: clamps the PAD form's last run
};

// One operand tile in flight between global memory and LDS.
template <int PASSES, int VEC>
struct Stage {
    float v[PASSES][VEC];
};

// STUDY arithmetic (SPLIT = 1, off unless pleas_gram_split_bf16(1) / PLEAS_GRAM_SPLIT_BF16=1): an fp32 value as the exact
// sum of three bf16 values, v = h1 + h2 + h3 (8 + 8 + 8 significant bits, each the round-to-nearest bf16 of what the
// previous ones left; the exponent range of bf16 is fp32's), so that an fp32 product becomes bf16-MFMA products:
//   x * y = x1 y1 + (x1 y2 + x2 y1) + (x2 y2 + x1 y3 + x3 y1)  + terms <= 2^-26 |x y| (dropped)
// Six v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each, K = 2).
// The split is done ONCE per element, when a K chunk goes from registers to LDS (three bf16 planes per row, the bytes
// of an fp32 row and a half); splitting at fragment-read time instead -- every element once per wave that uses it -- was
// bound by the conversions' issue slots (measured: 0.88 of the fp32 peak).  LDS is single-buffered here (two barriers per
// chunk, the next chunk waits in registers); two workgroups per CU alternate between converting and multiplying.
// Not for inf / NaN operands (inf - inf in the residual).
constexpr int kSplitLd = kSplitRow;      // common.hpp: three bf16 planes per LDS row
__device__ __forceinline__ void split3(const float (&v)[4], u32x2_t (&h)[3]) {      // h[plane] = four bf16, k order kept
    uint32_t lo[3], hi[3];
    split3_pair(v[0], v[1], lo);
    split3_pair(v[2], v[3], hi);
#pragma unroll
    for (int p = 0; p < 3; ++p) h[p] = u32x2_t{lo[p], hi[p]};
}

// One workgroup's share: output tile (tm, tn) over K chunks [c_begin, c_end) -> slab `split`.
template <int TILE, int VEC, int SPLIT = 0, int PAD = 0>
__device__ __forceinline__ void gram_tile(const GramGeom& g, float* smem, const int tm, const int tn, const int split,
                                          const int c_begin, const int c_end) {
    static_assert(!PAD || (VEC == 4 && !SPLIT), "the padded form: 16-byte loads, exact arithmetic");
    constexpr int MT = TILE / 64;                    // 32x32 MFMA tiles per wave per side
    constexpr int LANES_PER_ROW = kBK / VEC;         // 8 (16-B loads) or 32 (4-B loads)
    constexpr int ROWS_PER_PASS = kThreads / LANES_PER_ROW;
    constexpr int PASSES = TILE / ROWS_PER_PASS;
    float* As = smem;                         // [2][TILE][kLds]
    float* Bs = smem + 2 * TILE * kLds;       // [2][TILE][kLds]
    __bf16* As16 = reinterpret_cast<__bf16*>(smem);      // SPLIT: [TILE][kSplitLd], one buffer per operand
    __bf16* Bs16 = As16 + TILE * kSplitLd;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = tm * TILE, j0 = tn * TILE;

    const int srow = SPLIT ? split_stage_row(tid / LANES_PER_ROW) : tid / LANES_PER_ROW;      // SPLIT: VEC == 4, 32 rows per pass
    const int scol = (tid % LANES_PER_ROW) * VEC;

    Stage<PASSES, VEC> ra, rb;
    float sqa[PASSES], sqb[PASSES], sma[PASSES], smb[PASSES];
#pragma unroll
    for (int q = 0; q < PASSES; ++q) sqa[q] = sqb[q] = sma[q] = smb[q] = 0.f;
    const bool want_sums = g.sums != 0;   // block-uniform
    // EVERY tile does the row-norm / row-sum arithmetic, although only the first block column / row stores it: the tiles of
    // one (node, K range) stream the same operand rows through one XCD's L2 and stay in step only if they do the same work
    // per chunk.  Round 3 let the other tiles skip it (commit 4ffa37f): they ran ahead, and the launch fetched 15.2 GB
    // instead of 9.9 GB per ResNet-101 batch for no gain in time (profiles/r04_gram_traffic_bisect.txt; build with
    // -DPLEAS_GRAM_NORMS_ALWAYS=0 to reproduce).
#ifndef PLEAS_GRAM_NORMS_ALWAYS
#define PLEAS_GRAM_NORMS_ALWAYS 1
#endif
    const bool norm_a = tn == 0 || PLEAS_GRAM_NORMS_ALWAYS, norm_b = tm == 0 || PLEAS_GRAM_NORMS_ALWAYS;
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
    // BYTE offsets in 32 bits (the plan builders refuse nodes of 2^30 elements or more): a load is then the uniform base
    // pointer + one 32-bit vector offset, i.e. ONE vector add per load instead of a 64-bit add chain
    uint32_t row_off_a[PASSES], row_off_b[PASSES];
#pragma unroll
    for (int q = 0; q < PASSES; ++q) {
        const int row = srow + q * ROWS_PER_PASS;
        const int gi = i0 + row, gj = j0 + row;
        if (gi < g.C) rows_ok_a |= 1u << q;
        if (gj < g.C) rows_ok_b |= 1u << q;
        row_off_a[q] = 4u * ((uint32_t)min(gi, g.C - 1) * g.HW);
        row_off_b[q] = 4u * ((uint32_t)min(gj, g.C - 1) * g.HW);
    }
    bool staged_kin = false;
    bool staged_full = false;                                         // the staged chunk lies inside K (block-uniform)
    const bool interior = i0 + TILE <= g.C && j0 + TILE <= g.C;       // every row of both operand tiles exists
    // (sample, pixel) of this thread's k in the chunk being loaded: ONE division per work item, then steps of kBK
    // (chunks are loaded in order); images smaller than a chunk keep the division
    const uint32_t HWk = PAD ? g.HWp : g.HW, Kk = PAD ? g.Kk : g.Ktot;      // pixels per sample / length of the K axis
    uint32_t lk = (uint32_t)c_begin * kBK + scol;
    uint32_t ln = lk / HWk, lp = lk - ln * HWk;
    unsigned pmask = 0xFu;                                                    // PAD: which of the thread's four pixels exist
    const char* xb = reinterpret_cast<const char*>(g.x);
    const char* yb = reinterpret_cast<const char*>(g.y);
    auto load_chunk = [&](int c) {
        const uint32_t k = lk;
        const bool kin = k < Kk;
        staged_full = !PAD && (uint32_t)(c + 1) * kBK <= Kk;
        const uint32_t n = kin ? ln : 0u;
        const uint32_t p = kin ? lp : 0u;
        const uint32_t base = 4u * (n * (uint32_t)g.C * g.HW + p);
        staged_kin = kin;
        if constexpr (PAD) {
            pmask = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pmask |= (p + e < g.HW ? 1u : 0u) << e;
        }
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            if constexpr (PAD) {
                // rows start at 4-byte-aligned addresses only: one under-aligned 16-byte load per operand; the run that would
                // read past the tensor's last float goes element by element, clamped (its extra elements are masked)
                const uint32_t oa = base + row_off_a[q], ob = base + row_off_b[q];
                if (oa + 16u <= g.bytes) {
                    const f32x4 va = *(const __attribute__((address_space(1))) f32x4u*)(xb + oa);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ra.v[q][e] = va[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ra.v[q][e] = *(const __attribute__((address_space(1))) float*)(xb + min(oa + 4u * e, g.bytes - 4u));
                }
                if (ob + 16u <= g.bytes) {
                    const f32x4 vb = *(const __attribute__((address_space(1))) f32x4u*)(yb + ob);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb.v[q][e] = vb[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb.v[q][e] = *(const __attribute__((address_space(1))) float*)(yb + min(ob + 4u * e, g.bytes - 4u));
                }
            } else if constexpr (VEC == 4) {
                const f32x4 va = *(const __attribute__((address_space(1))) f32x4*)(xb + (base + row_off_a[q]));
                const f32x4 vb = *(const __attribute__((address_space(1))) f32x4*)(yb + (base + row_off_b[q]));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ra.v[q][e] = va[e];
                    rb.v[q][e] = vb[e];
                }
            } else {
                ra.v[q][0] = *(const __attribute__((address_space(1))) float*)(xb + (base + row_off_a[q]));
                rb.v[q][0] = *(const __attribute__((address_space(1))) float*)(yb + (base + row_off_b[q]));
            }
        }
        // next chunk
        lk += kBK;
        if (HWk >= (uint32_t)kBK) {
            lp += kBK;
            if (lp >= HWk) {
                lp -= HWk;
                ++ln;
            }
        } else {
            ln = lk / HWk;
            lp = lk - ln * HWk;
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
            // a tile inside the matrix on a chunk inside K needs no masking (block-uniform test, no selects)
            if constexpr (PAD) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const bool pe = (pmask >> e) & 1u;
                    ra.v[q][e] = (oka && pe) ? ra.v[q][e] : 0.f;
                    rb.v[q][e] = (okb && pe) ? rb.v[q][e] : 0.f;
                }
            } else if (!(interior && staged_full)) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    ra.v[q][e] = oka ? ra.v[q][e] : 0.f;
                    rb.v[q][e] = okb ? rb.v[q][e] : 0.f;
                }
            }
            if constexpr (SPLIT) {
                static_assert(!SPLIT || VEC == 4, "the split image is written four k at a time");
                u32x2_t ha[3], hb[3];
                split3(ra.v[q], ha);
                split3(rb.v[q], hb);
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    *reinterpret_cast<u32x2_t*>(As16 + row * kSplitLd + p * kBK + scol) = ha[p];
                    *reinterpret_cast<u32x2_t*>(Bs16 + row * kSplitLd + p * kBK + scol) = hb[p];
                }
            } else if constexpr (VEC == 4) {
                f32x4 va = {ra.v[q][0], ra.v[q][1], ra.v[q][2], ra.v[q][3]};
                f32x4 vb = {rb.v[q][0], rb.v[q][1], rb.v[q][2], rb.v[q][3]};
                *reinterpret_cast<f32x4*>(a + row * kLds + scol) = va;
                *reinterpret_cast<f32x4*>(b + row * kLds + scol) = vb;
            } else {
                a[row * kLds + scol] = ra.v[q][0];
                b[row * kLds + scol] = rb.v[q][0];
            }
            // row norms / sums are stored by the first block column (x rows) and the first block row (y rows) only:
            // the other tiles skip the arithmetic (block-uniform conditions)
            if (norm_a) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) sqa[q] = fmaf(ra.v[q][e], ra.v[q][e], sqa[q]);
                if (want_sums) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) sma[q] += ra.v[q][e];
                }
            }
            if (norm_b) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) sqb[q] = fmaf(rb.v[q][e], rb.v[q][e], sqb[q]);
                if (want_sums) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) smb[q] += rb.v[q][e];
                }
            }
        }
    };
    auto compute = [&](int buf) {
        const float* a = As + buf * TILE * kLds + (wm * (TILE / 2) + (lane & 31)) * kLds + 4 * (lane >> 5);
        const float* b = Bs + buf * TILE * kLds + (wn * (TILE / 2) + (lane & 31)) * kLds + 4 * (lane >> 5);
        if constexpr (SPLIT) {
            // lane (r, h) of k group g reads k = 16 g + 8 h .. + 7 of its row from each plane: the operand map of the MFMA
            const __bf16* a16 = As16 + (wm * (TILE / 2) + (lane & 31)) * kSplitLd + 8 * (lane >> 5);
            const __bf16* b16 = Bs16 + (wn * (TILE / 2) + (lane & 31)) * kSplitLd + 8 * (lane >> 5);
#pragma unroll
            for (int g16 = 0; g16 < kBK / 16; ++g16) {
                bf16x8 sa[MT][3], sb[MT][3];
#pragma unroll
                for (int s = 0; s < MT; ++s)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        sa[s][p] = *reinterpret_cast<const bf16x8*>(a16 + s * 32 * kSplitLd + p * kBK + g16 * 16);
                        sb[s][p] = *reinterpret_cast<const bf16x8*>(b16 + s * 32 * kSplitLd + p * kBK + g16 * 16);
                    }
#pragma unroll
                for (int sm = 0; sm < MT; ++sm)
#pragma unroll
                    for (int sn = 0; sn < MT; ++sn) {
                        f32x16 c = acc[sm][sn];      // smallest terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][0], sb[sn][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][2], sb[sn][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][1], sb[sn][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][0], sb[sn][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][1], sb[sn][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[sm][0], sb[sn][0], c, 0, 0, 0);
                        acc[sm][sn] = c;
                    }
            }
            return;
        }
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
    if constexpr (SPLIT) {
        for (int c = c_begin; c < c_end; ++c) {
            const bool more = c + 1 < c_end;
            if (more) load_chunk(c + 1);      // stays in registers while this chunk is multiplied
            compute(0);
            __syncthreads();                  // every wave is done reading the image
            if (more) store_chunk(0);
            __syncthreads();
        }
    } else
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
    gfloat* gp = PLEAS_GLOBAL_W(g.gpart) + (size_t)split * g.C * g.C;
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
            if (tn == 0 && i0 + row < g.C) PLEAS_GLOBAL_W(g.npart)[((size_t)split * 2 + 0) * g.C + i0 + row] = sa;
            if (tm == 0 && j0 + row < g.C) PLEAS_GLOBAL_W(g.npart)[((size_t)split * 2 + 1) * g.C + j0 + row] = sb;
        }
    }
    if (want_sums) {   // plain row sums of the same K range, same layout
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            float sa = sma[q], sb = smb[q];
#pragma unroll
            for (int off = 1; off < LANES_PER_ROW; off <<= 1) {
                sa += __shfl_xor(sa, off);
                sb += __shfl_xor(sb, off);
            }
            if ((tid % LANES_PER_ROW) == 0) {
                const int row = srow + q * ROWS_PER_PASS;
                if (tn == 0 && i0 + row < g.C) PLEAS_GLOBAL_W(g.spart)[((size_t)split * 2 + 0) * g.C + i0 + row] = sa;
                if (tm == 0 && j0 + row < g.C) PLEAS_GLOBAL_W(g.spart)[((size_t)split * 2 + 1) * g.C + j0 + row] = sb;
            }
        }
    }
}

// Single-node launch: grid = tiles x tiles x S.
template <int TILE, int VEC>
__global__ __launch_bounds__(kThreads) void gram_partial_kernel(const GramGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    const int tn = bid % g.tiles;
    bid /= g.tiles;
    const int tm = bid % g.tiles;
    const int split = bid / g.tiles;
    const int c_begin = split * g.chunks_per_split;
    gram_tile<TILE, VEC>(g, smem, tm, tn, split, c_begin, min(c_begin + g.chunks_per_split, g.nchunks));
}

template <int TILE, int VEC>
__global__ __launch_bounds__(kThreads) void gram_partial_split_kernel(const GramGeom g) {      // study arithmetic
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    const int tn = bid % g.tiles;
    bid /= g.tiles;
    const int tm = bid % g.tiles;
    const int split = bid / g.tiles;
    const int c_begin = split * g.chunks_per_split;
    gram_tile<TILE, VEC, 1>(g, smem, tm, tn, split, c_begin, min(c_begin + g.chunks_per_split, g.nchunks));
}

// ---- grouped launch: every tracked node of a batch in ONE grid --------------------------------
struct GramNodeDev {      // device node table entry
    const float* x;
    const float* y;
    float* gpart;         // [S][C][C] slabs of this node
    float* npart;         // [S][2][C]
    float* spart;         // [S][2][C] row sums, sources of derived nodes only
    const float* ax;      // derived node: value = ax[c] * source_x + bx[c] (per channel), likewise ay / by for y
    const float* bx;
    const float* ay;
    const float* by;
    int C;
    uint32_t HW;
    uint32_t Ktot;
    uint32_t HWp, Kk;     // PAD forms: padded pixels per sample, B * HWp (else HW, Ktot)
    int variant;          // 0: <128,4>  1: <128,1>  2: <64,4>  3: <64,1>  4: <128,4,PAD>  5: <64,4,PAD>
    int S;
    int group;
    int source;           // >= 0: derived from that node's slabs (no contraction of its own); -1: contracted
    int sums;             // this node also writes row sums
};
struct GramItemDev {      // one workgroup of the grouped launch
    int node, tm, tn, split, c_begin, c_end, pad0, pad1;
};

__global__ __launch_bounds__(kThreads, 2) void gram_batch_kernel(const GramNodeDev* __restrict__ nodes,
                                                              const GramItemDev* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GramItemDev it = items[blockIdx.x];
    if (it.node < 0) return;   // padding of the XCD-aware item order
    const GramNodeDev nd = nodes[it.node];
    GramGeom g;
    g.x = nd.x;
    g.y = nd.y;
    g.gpart = nd.gpart;
    g.npart = nd.npart;
    g.spart = nd.spart;
    g.sums = nd.sums;
    g.C = nd.C;
    g.HW = nd.HW;
    g.Ktot = nd.Ktot;
    g.HWp = nd.HWp;
    g.Kk = nd.Kk;
    g.bytes = 4u * nd.C * nd.Ktot;
    g.nchunks = 0;
    g.chunks_per_split = 0;
    g.tiles = 0;
    switch (nd.variant) {  // block-uniform
        case 0: gram_tile<128, 4>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;
        case 1: gram_tile<128, 1>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;
        case 2: gram_tile<64, 4>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;
        case 3: gram_tile<64, 1>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;
        case 4: gram_tile<128, 4, 0, 1>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;
        default: gram_tile<64, 4, 0, 1>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;
    }
}

// The same grid with the study arithmetic on the 128-wide 16-byte variant (every MFMA-bound node of a ResNet); a kernel of
// its own so that the exact kernel's register allocation does not depend on it.
__global__ __launch_bounds__(kThreads, 2) void gram_batch_split_kernel(const GramNodeDev* __restrict__ nodes,
                                                                    const GramItemDev* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GramItemDev it = items[blockIdx.x];
    if (it.node < 0) return;
    const GramNodeDev nd = nodes[it.node];
    GramGeom g;
    g.x = nd.x;
    g.y = nd.y;
    g.gpart = nd.gpart;
    g.npart = nd.npart;
    g.spart = nd.spart;
    g.sums = nd.sums;
    g.C = nd.C;
    g.HW = nd.HW;
    g.Ktot = nd.Ktot;
    g.HWp = nd.HWp;
    g.Kk = nd.Kk;
    g.bytes = 4u * nd.C * nd.Ktot;
    g.nchunks = 0;
    g.chunks_per_split = 0;
    g.tiles = 0;
    switch (nd.variant) {
        case 0: gram_tile<128, 4, 1>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;
        case 1: gram_tile<128, 1>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;
        case 2: gram_tile<64, 4, 1>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;
        case 3: gram_tile<64, 1>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;
        case 4: gram_tile<128, 4, 0, 1>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;      // exact: no split variant
        default: gram_tile<64, 4, 0, 1>(g, smem, it.tm, it.tn, it.split, it.c_begin, it.c_end); break;
    }
}

// Writes the per-batch operand pointers into the device node table (kernel arguments carry them,
// so no host staging buffer has to outlive the call).
constexpr int kPtrBatch = 224;
struct GramPtrBatch {
    int base, count;
    const float* x[kPtrBatch];
    const float* y[kPtrBatch];
};
__global__ void gram_set_ptrs_kernel(GramNodeDev* __restrict__ nodes, const GramPtrBatch b) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < b.count) {
        nodes[b.base + t].x = b.x[t];
        nodes[b.base + t].y = b.y[t];
    }
}

// Per group: acc (+)= sum over its nodes (fixed order) of epilogue(sum over slabs of G, norms).
// One thread per element; blocks are mapped to (group, local block) through a table.
struct GramGroupDev {
    float* acc;
    int C;
    int node_begin, node_end;  // range in the group-ordered node index list
    int blk_begin;             // first block of this group in the reduce grid
    int pad;
};
__global__ __launch_bounds__(256) void gram_group_reduce_kernel(const GramNodeDev* __restrict__ nodes,
                                                                const GramGroupDev* __restrict__ groups,
                                                                const int* __restrict__ group_nodes,
                                                                const int* __restrict__ blk_group, int epilogue,
                                                                int accumulate) {
    const int gidx = blk_group[blockIdx.x];
    const GramGroupDev gr = groups[gidx];
    const size_t total = (size_t)gr.C * gr.C;
    const size_t idx = (size_t)(blockIdx.x - gr.blk_begin) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx / gr.C), j = (int)(idx - (size_t)i * gr.C);
    float out = accumulate ? gr.acc[idx] : 0.f;
    for (int t = gr.node_begin; t < gr.node_end; ++t) {
        const GramNodeDev nd = nodes[group_nodes[t]];
        if (nd.source >= 0) {
            // Derived node: x' = ax[i] x + bx[i], y' = ay[j] y + by[j] (an eval-mode BatchNorm of a contracted node).
            // Its inner products and norms follow from the source's G, squared norms and row sums -- no contraction:
            //   <x'_i, y'_j> = ax ay G + ax by Sx_i + bx ay Sy_j + K bx by,   |x'_i|^2 = ax^2 Nx_i + 2 ax bx Sx_i + K bx^2.
            // Combined in fp64 (this pass is HBM-bound); the fp32 sums are the same ones a contraction would produce.
            const GramNodeDev src = nodes[nd.source];
            float gsum = 0.f, nx = 0.f, ny = 0.f, sx = 0.f, sy = 0.f;
            for (int s = 0; s < src.S; ++s) {
                gsum += src.gpart[(size_t)s * total + idx];
                nx += src.npart[((size_t)s * 2 + 0) * gr.C + i];
                ny += src.npart[((size_t)s * 2 + 1) * gr.C + j];
                sx += src.spart[((size_t)s * 2 + 0) * gr.C + i];
                sy += src.spart[((size_t)s * 2 + 1) * gr.C + j];
            }
            const double ax = nd.ax[i], bx = nd.bx[i], ay = nd.ay[j], by = nd.by[j], K = (double)src.Ktot;
            const double inner = ax * ay * (double)gsum + ax * by * (double)sx + bx * ay * (double)sy + K * bx * by;
            if (epilogue == PLEAS_EPI_NEG_CDIST) {
                const double n1 = ax * ax * (double)nx + 2.0 * ax * bx * (double)sx + K * bx * bx;
                const double n2 = ay * ay * (double)ny + 2.0 * ay * by * (double)sy + K * by * by;
                out += -sqrtf(fmaxf((float)(n1 + n2 - 2.0 * inner), 0.f));
            } else {
                out += (float)inner;
            }
            continue;
        }
        float gsum = 0.f, nx = 0.f, ny = 0.f;
        for (int s = 0; s < nd.S; ++s) {
            gsum += nd.gpart[(size_t)s * total + idx];
            nx += nd.npart[((size_t)s * 2 + 0) * gr.C + i];
            ny += nd.npart[((size_t)s * 2 + 1) * gr.C + j];
        }
        out += epilogue == PLEAS_EPI_NEG_CDIST ? -sqrtf(fmaxf(nx + ny - 2.f * gsum, 0.f)) : gsum;
    }
    gr.acc[idx] = out;
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

// the matching contraction under pleas_arith(PLEAS_ARITH_SPLIT_BF16): gram_tile<TILE, 4, SPLIT = 1> on 16-byte-loadable nodes
static bool split_bf16() { return arith_mode() == 1; }
extern "C" void pleas_gram_split_bf16(int on) { pleas_arith(on ? PLEAS_ARITH_SPLIT_BF16 : PLEAS_ARITH_FP32); }   // round-3 name

extern "C" size_t pleas_gram_ws_bytes(int B, int C, int64_t HW) {
    if (B <= 0 || C <= 0 || HW <= 0) return 0;
    return make_plan(B, C, HW, true).ws_bytes;  // alignment does not change the size
}

extern "C" int pleas_gram_accum(const float* x, const float* y, int B, int C, int64_t HW, int epilogue, int accumulate,
                                float* acc, void* ws, size_t ws_bytes, void* stream_) {
    if (!x || !y || !acc) return bad_arg("null tensor pointer");
    if (B <= 0 || C <= 0 || HW <= 0) return bad_arg("B, C, HW must be positive");
    if ((int64_t)B * HW >= (1ll << 31) || HW >= (1ll << 31)) return bad_arg("B*HW must be < 2^31");
    if ((int64_t)B * C * HW >= (1ll << 30)) return bad_arg("B*C*HW must be < 2^30 (the tile addresses an operand with 32-bit byte offsets)");
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
    g.spart = nullptr;
    g.sums = 0;
    g.C = C;
    g.HW = (uint32_t)HW;
    g.Ktot = (uint32_t)((int64_t)B * HW);
    g.HWp = g.HW;
    g.Kk = g.Ktot;
    g.bytes = 4u * (uint32_t)C * g.Ktot;
    g.nchunks = p.nchunks;
    g.chunks_per_split = p.cps;
    g.tiles = p.tiles;
    const dim3 grid((unsigned)(p.tiles * p.tiles * p.S));
    const size_t lds = (size_t)4 * p.tile * kLds * sizeof(float);
    const double kk = (double)B * (double)HW;
    {
    ProfScope prof(kProfGramPartial, 2.0 * C * (double)C * kk, 2.0 * C * kk * sizeof(float), stream);
    if (p.vec == 4 && split_bf16()) {
        const size_t lds16 = (size_t)2 * p.tile * kSplitLd * sizeof(__bf16);      // 52 KB at tile 128
        if (p.tile == 128)
            hipLaunchKernelGGL((gram_partial_split_kernel<128, 4>), grid, dim3(kThreads), lds16, stream, g);
        else
            hipLaunchKernelGGL((gram_partial_split_kernel<64, 4>), grid, dim3(kThreads), lds16, stream, g);
    } else if (p.tile == 128 && p.vec == 4)
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

// =========================================================================================
// Grouped launch: all tracked nodes of one batch -> one contraction grid + one reduce grid.
// =========================================================================================
namespace pleas {

static int g_item_chunks = 112;
static int g_xcd_order = 2;  // 0: longest first; 1: compact tile block per XCD inside a (node, split); 2: whole (node, split) per XCD

struct BatchPlan {
    std::vector<int64_t> key;
    std::vector<GramNodeDev> nodes;
    std::vector<GramItemDev> items;
    std::vector<GramGroupDev> groups;
    std::vector<int> group_nodes, blk_group;
    size_t off_nodes = 0, off_items = 0, off_groups = 0, off_gn = 0, off_bg = 0, off_slabs = 0, total = 0;
    int reduce_blocks = 0;
    size_t lds = 0;
    double flops = 0, bytes = 0, slab_bytes = 0;
    bool uploaded = false;
};
static PlanCache<BatchPlan, 2> g_bplans;
static std::mutex g_bplan_mu;

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Layout + work list for a node sequence; device pointers are filled relative to `ws` later.
static int build_batch_plan(BatchPlan& P, const pleas_gram_node* nd, int n, float* const* group_acc, const int* group_C,
                            int n_groups) {
    P.nodes.assign(n, GramNodeDev());
    P.items.clear();
    P.groups.assign(n_groups, GramGroupDev());
    P.group_nodes.clear();
    P.blk_group.clear();
    P.flops = P.bytes = P.slab_bytes = 0;
    P.lds = 0;
    std::vector<size_t> slab_off(n), spart_off(n);
    size_t slabs = 0;
    struct Work { double w; GramItemDev it; };
    std::vector<Work> work;
    std::vector<XcdWork<GramItemDev>> xwork;
    std::vector<char> is_source(n, 0);
    for (int i = 0; i < n; ++i) {
        if (!nd[i].derived) continue;
        const int src = nd[i].source;
        if (src < 0 || src >= n || src == i || nd[src].derived) return bad_arg("gram_batch: derived node needs a contracted source node");
        if (nd[src].C != nd[i].C || nd[src].B != nd[i].B || nd[src].HW != nd[i].HW) return bad_arg("gram_batch: derived node shape differs from its source");
        if (!nd[i].scale_x || !nd[i].shift_x || !nd[i].scale_y || !nd[i].shift_y) return bad_arg("gram_batch: derived node without scale / shift");
        is_source[src] = 1;
    }
    for (int i = 0; i < n; ++i) {
        const int B = nd[i].B, C = nd[i].C;
        const int64_t HW = nd[i].HW;
        if (B <= 0 || C <= 0 || HW <= 0 || (int64_t)B * HW >= (1ll << 31)) return bad_arg("gram_batch: node shape");
        if ((int64_t)B * C * HW >= (1ll << 30)) return bad_arg("gram_batch: a node must hold fewer than 2^30 elements per batch");
        if (nd[i].group < 0 || nd[i].group >= n_groups || group_C[nd[i].group] != C) return bad_arg("gram_batch: node group");
        const int tile = C > 64 ? 128 : 64;
        const int vec = (HW % 4 == 0) ? 4 : 1;  // operand alignment is checked per call
        const int tiles = (int)ceil_div(C, tile);
        // images with HW % 4 != 0 and at least four pixels: the padded K axis with under-aligned 16-byte loads instead of one
        // pixel per load (PLEAS_GRAM_PADK=0 keeps the scalar form for A/B)
        static const bool padk = !(std::getenv("PLEAS_GRAM_PADK") && std::atoi(std::getenv("PLEAS_GRAM_PADK")) == 0);
        const bool pad = padk && vec == 1 && HW >= 4;
        const int64_t HWp = pad ? (HW + 3) / 4 * 4 : HW;
        const int nchunks = (int)ceil_div((int64_t)B * HWp, kBK);
        const int S = (int)ceil_div(nchunks, g_item_chunks);
        const int cps = (int)ceil_div(nchunks, S);
        GramNodeDev& d = P.nodes[i];
        d.C = C;
        d.HW = (uint32_t)HW;
        d.Ktot = (uint32_t)((int64_t)B * HW);
        d.HWp = (uint32_t)HWp;
        d.Kk = (uint32_t)((int64_t)B * HWp);
        d.variant = pad ? (tile == 128 ? 4 : 5) : (tile == 128 ? 0 : 2) + (vec == 4 ? 0 : 1);
        d.group = nd[i].group;
        d.source = nd[i].derived ? nd[i].source : -1;
        d.sums = is_source[i];
        d.ax = nd[i].scale_x; d.bx = nd[i].shift_x; d.ay = nd[i].scale_y; d.by = nd[i].shift_y;
        if (nd[i].derived) {   // no contraction, no slabs: the reduce pass reads the source's
            d.S = 0;
            slab_off[i] = 0;
            spart_off[i] = 0;
            continue;
        }
        d.S = S;
        slab_off[i] = slabs;
        slabs += (size_t)S * ((size_t)C * C + 2 * (size_t)C);
        spart_off[i] = slabs;
        if (is_source[i]) slabs += (size_t)S * 2 * (size_t)C;
        P.lds = std::max(P.lds, (size_t)4 * tile * kLds * sizeof(float));
        const double kk = (double)B * (double)HW;
        P.flops += 2.0 * C * (double)C * kk;
        P.bytes += 2.0 * C * kk * sizeof(float);
        for (int s = 0; s < S; ++s)
            for (int tm = 0; tm < tiles; ++tm)
                for (int tn = 0; tn < tiles; ++tn) {
                    Work w;
                    w.it = GramItemDev{i, tm, tn, s, s * cps, std::min((s + 1) * cps, nchunks), 0, 0};
                    w.w = (double)(w.it.c_end - w.it.c_begin) * tile * tile;
                    work.push_back(w);
                    // every tile of one (node, K range) streams the same operand rows, chunk by chunk and roughly in step
                    xwork.push_back(XcdWork<GramItemDev>{w.w, (int64_t)i * 65536 + s, w.it});
                }
    }
    if (g_xcd_order == 2) {
        P.items = xcd_order_items(xwork, GramItemDev{-1, 0, 0, 0, 0, 0, 0, 0});
    } else {
        std::stable_sort(work.begin(), work.end(), [](const Work& a, const Work& b) { return a.w > b.w; });
        P.items.reserve(work.size());
        for (auto& w : work) P.items.push_back(w.it);
    }
    // XCD-aware tile order: workgroups b and b+8 share an XCD (private L2).  Inside every run of
    // items that belongs to one (node, split), give each XCD a compact sr x sc block of output
    // tiles, so the operand rows a tile row / column needs are fetched by one L2 instead of eight.
    if (g_xcd_order == 1) {
        size_t pos = 0;
        while (pos < P.items.size()) {
            size_t end = pos;
            while (end < P.items.size() && P.items[end].node == P.items[pos].node && P.items[end].split == P.items[pos].split)
                ++end;
            const int var_ = P.nodes[P.items[pos].node].variant;
            const int t = (int)ceil_div(P.nodes[P.items[pos].node].C, (var_ < 2 || var_ == 4) ? 128 : 64);
            if ((int)(end - pos) == t * t && t >= 4 && (t & (t - 1)) == 0) {
                int sr = 1, sc = 1;  // sr * sc = t*t/8, as square as possible, sc >= sr
                for (int area = t * t / 8; sr * sc < area;) (sc <= sr ? sc : sr) *= 2;
                if (sr > sc) std::swap(sr, sc);
                const int blocks_c = t / sc;
                int seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (size_t g = pos; g < end; ++g) {
                    const int xcd = (int)(g % 8), k = seen[xcd]++;
                    P.items[g].tm = (xcd / blocks_c) * sr + k / sc;
                    P.items[g].tn = (xcd % blocks_c) * sc + k % sc;
                }
            }
            pos = end;
        }
    }
    // groups: node lists in first-appearance order, reduce blocks of 256 elements
    int blk = 0;
    for (int g = 0; g < n_groups; ++g) {
        GramGroupDev& G = P.groups[g];
        G.acc = group_acc[g];
        G.C = group_C[g];
        G.node_begin = (int)P.group_nodes.size();
        for (int i = 0; i < n; ++i)
            if (nd[i].group == g) P.group_nodes.push_back(i);
        G.node_end = (int)P.group_nodes.size();
        G.blk_begin = blk;
        const int nb = G.node_end > G.node_begin ? (int)ceil_div((int64_t)G.C * G.C, 256) : 0;
        for (int b = 0; b < nb; ++b) P.blk_group.push_back(g);
        blk += nb;
    }
    P.reduce_blocks = blk;
    size_t off = 0;
    P.off_nodes = off;
    off = align_up(off + P.nodes.size() * sizeof(GramNodeDev), 256);
    P.off_items = off;
    off = align_up(off + P.items.size() * sizeof(GramItemDev), 256);
    P.off_groups = off;
    off = align_up(off + P.groups.size() * sizeof(GramGroupDev), 256);
    P.off_gn = off;
    off = align_up(off + P.group_nodes.size() * sizeof(int), 256);
    P.off_bg = off;
    off = align_up(off + P.blk_group.size() * sizeof(int), 256);
    P.off_slabs = off;
    P.slab_bytes = (double)slabs * sizeof(float);
    P.total = off + slabs * sizeof(float);
    // slab offsets (in floats, relative to the slab region); absolute pointers are set at upload
    for (int i = 0; i < n; ++i) {
        P.nodes[i].gpart = reinterpret_cast<float*>(slab_off[i]);
        P.nodes[i].npart = reinterpret_cast<float*>(slab_off[i] + (size_t)P.nodes[i].S * P.nodes[i].C * P.nodes[i].C);
        P.nodes[i].spart = reinterpret_cast<float*>(spart_off[i]);
    }
    P.uploaded = false;
    return PLEAS_OK;
}

static std::vector<int64_t> batch_key(const pleas_gram_node* nd, int n, float* const* group_acc, const int* group_C,
                                      int n_groups, const void* ws) {
    std::vector<int64_t> k;
    k.reserve(4 * n + 2 * n_groups + 3);
    k.push_back(n);
    k.push_back(n_groups);
    k.push_back((int64_t)(uintptr_t)ws);
    k.push_back(g_item_chunks * 4 + g_xcd_order);
    for (int i = 0; i < n; ++i) {
        k.push_back(nd[i].B);
        k.push_back(nd[i].C);
        k.push_back(nd[i].HW);
        k.push_back(nd[i].group);
        k.push_back(nd[i].derived ? nd[i].source : -1);
        if (nd[i].derived)
            for (const float* ptr : {nd[i].scale_x, nd[i].shift_x, nd[i].scale_y, nd[i].shift_y}) k.push_back((int64_t)(uintptr_t)ptr);
    }
    for (int g = 0; g < n_groups; ++g) {
        k.push_back(group_C[g]);
        k.push_back((int64_t)(uintptr_t)group_acc[g]);
    }
    return k;
}

}  // namespace pleas

extern "C" void pleas_gram_batch_tune(int item_chunks, int xcd_order) {
    if (item_chunks > 0) g_item_chunks = item_chunks;
    if (xcd_order >= 0) g_xcd_order = xcd_order;
}

extern "C" size_t pleas_gram_batch_ws_bytes(const pleas_gram_node* nodes, int n_nodes, const int* group_C, int n_groups) {
    if (!nodes || n_nodes <= 0 || !group_C || n_groups <= 0) return 0;
    BatchPlan tmp;
    std::vector<float*> acc(n_groups, nullptr);
    if (build_batch_plan(tmp, nodes, n_nodes, acc.data(), group_C, n_groups) != PLEAS_OK) return 0;
    return tmp.total;
}

extern "C" int pleas_gram_batch(const pleas_gram_node* nodes, int n_nodes, float* const* group_acc, const int* group_C,
                                int n_groups, int epilogue, int accumulate, void* ws, size_t ws_bytes, int ws_fresh,
                                void* stream_) {
    if (!nodes || n_nodes <= 0 || !group_acc || !group_C || n_groups <= 0) return bad_arg("gram_batch: empty input");
    if (epilogue != PLEAS_EPI_INNER && epilogue != PLEAS_EPI_NEG_CDIST) return bad_arg("epilogue");
    for (int i = 0; i < n_nodes; ++i) {
        if (nodes[i].derived) continue;
        if (!nodes[i].x || !nodes[i].y) return bad_arg("gram_batch: null operand");
        if (nodes[i].HW % 4 == 0 && ((((uintptr_t)nodes[i].x | (uintptr_t)nodes[i].y) & 15) != 0))
            return bad_arg("gram_batch: operands must be 16-byte aligned");
    }
    for (int g = 0; g < n_groups; ++g)
        if (!group_acc[g]) return bad_arg("gram_batch: null group matrix");
    hipStream_t stream = (hipStream_t)stream_;
    std::lock_guard<std::mutex> lk(g_bplan_mu);
    std::vector<int64_t> key = batch_key(nodes, n_nodes, group_acc, group_C, n_groups, ws);
    BatchPlan* hit = g_bplans.find(key);
    if (!hit) {
        hit = &g_bplans.take();
        const int rc = build_batch_plan(*hit, nodes, n_nodes, group_acc, group_C, n_groups);
        if (rc != PLEAS_OK) return rc;
        hit->key.swap(key);
    }
    BatchPlan& P = *hit;
    if (ws_fresh) P.uploaded = false;  // caller says the tables inside ws are not (or no longer) there
    if (!ws || ws_bytes < P.total) {
        std::snprintf(g_last_error, sizeof(g_last_error), "gram_batch workspace too small: need %zu bytes", P.total);
        P.key.clear();
        return PLEAS_ENOMEM;
    }
    char* base = (char*)ws;
    if (!P.uploaded) {  // static tables: once per (shape sequence, workspace, group matrices)
        g_bplans.claims_workspace(P);
        float* slab0 = reinterpret_cast<float*>(base + P.off_slabs);
        std::vector<GramNodeDev> abs_nodes = P.nodes;
        for (auto& d : abs_nodes) {
            d.gpart = slab0 + reinterpret_cast<size_t>(d.gpart);
            d.npart = slab0 + reinterpret_cast<size_t>(d.npart);
            d.spart = slab0 + reinterpret_cast<size_t>(d.spart);
        }
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_nodes, abs_nodes.data(), abs_nodes.size() * sizeof(GramNodeDev),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_items, P.items.data(), P.items.size() * sizeof(GramItemDev),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_groups, P.groups.data(), P.groups.size() * sizeof(GramGroupDev),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_gn, P.group_nodes.data(), P.group_nodes.size() * sizeof(int),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_bg, P.blk_group.data(), P.blk_group.size() * sizeof(int),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipStreamSynchronize(stream));  // host vectors may now change; happens once per plan
        P.uploaded = true;
    }
    GramNodeDev* dnodes = reinterpret_cast<GramNodeDev*>(base + P.off_nodes);
    for (int b0 = 0; b0 < n_nodes; b0 += kPtrBatch) {
        GramPtrBatch pb;
        pb.base = b0;
        pb.count = std::min(kPtrBatch, n_nodes - b0);
        for (int t = 0; t < pb.count; ++t) {
            pb.x[t] = nodes[b0 + t].x;
            pb.y[t] = nodes[b0 + t].y;
        }
        hipLaunchKernelGGL(gram_set_ptrs_kernel, dim3(1), dim3(256), 0, stream, dnodes, pb);
        PLEAS_LAUNCH_CHECK("gram_set_ptrs_kernel");
    }
    {
        ProfScope prof(kProfGramPartial, P.flops, P.bytes, stream);
        if (split_bf16())
            hipLaunchKernelGGL(gram_batch_split_kernel, dim3((unsigned)P.items.size()), dim3(kThreads), P.lds, stream, dnodes,
                               reinterpret_cast<const GramItemDev*>(base + P.off_items));
        else
            hipLaunchKernelGGL(gram_batch_kernel, dim3((unsigned)P.items.size()), dim3(kThreads), P.lds, stream, dnodes,
                               reinterpret_cast<const GramItemDev*>(base + P.off_items));
    }
    PLEAS_LAUNCH_CHECK("gram_batch_kernel");
    {
        ProfScope prof(kProfGramFinalize, 0.0, P.slab_bytes, stream);
        hipLaunchKernelGGL(gram_group_reduce_kernel, dim3((unsigned)P.reduce_blocks), dim3(256), 0, stream, dnodes,
                           reinterpret_cast<const GramGroupDev*>(base + P.off_groups),
                           reinterpret_cast<const int*>(base + P.off_gn), reinterpret_cast<const int*>(base + P.off_bg),
                           epilogue, accumulate);
    }
    PLEAS_LAUNCH_CHECK("gram_group_reduce_kernel");
    return PLEAS_OK;
}

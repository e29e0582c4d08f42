// Normal equations of the PLeaS layer objective on gfx950: A += U^T U for ALL merged layers of one
// batch in one grouped fp32-MFMA launch (the right-hand side B^T += op . U is pleas_wgrad_batch with
// PLEAS_WGRAD_ACCUMULATE | PLEAS_WGRAD_KPOS_MAJOR).
//
// The reference minimises  sum_batches mean((L(ip) - op)^2)  per layer with 401 Adam steps
// (pleas/methods/pleas_merging.py:281-291, :357-358); its closed form is  W^T = A^-1 B  with
// U = im2col(ip) (rows = samples x output pixels, columns = kernel position x input channel).
//
// U is never materialised.  Column k = (r, ci) of U is the r-shifted (strided) view of input
// channel ci, so the block of A between kernel positions (rx, ry) is an NT contraction over the
// flattened pixel index P = (n, oh, ow) of two shifted views of the SAME tensor, read in place
// from NCHW.  Only block tiles of the LOWER triangle are computed (A is symmetric).  Work items
// (layer, tile row, tile column, P range) of all layers form one grid sorted longest-first; P ranges
// longer than one item go through slabs and an ordered reduce (deterministic, no atomics).
//
// Algorithmic work: K^2 * N*HWo flop per layer and batch for the triangle (K = KH*KW*Cin);
// ResNet-101, batch 16: 6.7e11 flop per batch (4.18e10 per sample), MFMA bound.
//
// Stride-1 "same" k x k layers (ResNet: every 3x3 but three; 80 % of the flops above) -- LAG CLASSES.  Substituting
// q = o + d_x (the position operand x reads) turns the block between taps (x, y) into
//     C(delta, Q)[ci][cj] = sum_{q in Q} X[ci][q] * X[cj][q + delta],   delta = d_y - d_x,
// Q = the rectangle of q for which o, q and q + delta are all inside the image.  Per axis there are only 7 distinct
// (delta, Q) among the 9 tap pairs (delta = 0: three windows; delta = +-1, +-2: the full lag range each), so the 81
// blocks of a 3x3 layer take 49 distinct values, 29 up to transposition -- against the 45 blocks of the lower triangle.
// Only those 29 are contracted (tile form neq_lag_tile: operand x is the UNSHIFTED image, 16-byte loads, zeroed
// outside Q when staged; operand y is read delta floats further on -- one 16-byte load at a 4-byte-aligned address --
// unmasked, because x's zeros already void every product outside Q); pleas_normal_eq_finalize copies /
// transposes them into the other 16 blocks ONCE, after the last batch (and after the all-reduce of a multi-GPU run).
#include <algorithm>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace pleas {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef f32x4 f32x4u __attribute__((aligned(4)));   // 16 bytes at a 4-byte-aligned address: still ONE global_load_dwordx4

constexpr int nBK = 32;
constexpr int nLds = 36;
constexpr int nThreads = 256;

struct NeqLayerDev {
    const float* ip;  // [N][Cin][Hin][Win]
    float* A;         // [K][K], K = R*Cin, kernel-position-major index k = r*Cin + ci
    float* slab;      // [S][items of this layer][T*T] when S > 1 (indexed through item.slot)
    int Cin, Hin, Win, Hout, Wout, KH, KW, stride, pad;
    uint32_t HWo, Ktot;
    int S, K;
    int variant;      // bit0: 64-wide tiles, bit1: scalar loads, bits 2-3: 0 direct, 1 shifted loader, 2 lag classes
    int total;        // floats in ip (lag form: clamps the shifted operand's loads)
};
struct NeqItemDev {
    int layer, tm, tn, rx, ry, split, c_begin, c_end, slot, pad0;
};

// ---- epilogue shared by both tile forms: S == 1 -> A += tile; else tile -> this (slot, split) slab.
// The accumulators hold 16 scattered elements per MFMA tile and lane; they go through LDS once ([row][T + 4]; the staging
// buffers are free after the K loop's last barrier) so that every thread then owns 16-byte runs of a row: all of a
// thread's read-modify-writes of A are 16-byte accesses, requested together (the accumulator registers are free by then).
template <int T>
__device__ __forceinline__ void neq_store_tile(const NeqLayerDev& L, const NeqItemDev& it, f32x16 (&acc)[T / 64][T / 64], float* smem) {
    constexpr int MT = T / 64, EL = T + 4, VPT = T * T / 4 / nThreads;   // 16-byte vectors per thread: 16 (T = 128) / 4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = it.tm * T, j0 = it.tn * T;
    float* Ct = smem;
#pragma unroll
    for (int sm = 0; sm < MT; ++sm)
#pragma unroll
        for (int sn = 0; sn < MT; ++sn) {
            const int lj = wn * (T / 2) + sn * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int li = wm * (T / 2) + sm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Ct[li * EL + lj] = acc[sm][sn][r];
            }
        }
    __syncthreads();
    if (L.S > 1) {
        gfloat* slab = PLEAS_GLOBAL_W(L.slab) + ((size_t)it.slot * L.S + it.split) * (T * T);
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int v = tid + q * nThreads, li = v / (T / 4), lj = (v % (T / 4)) * 4;
            *(__attribute__((address_space(1))) f32x4*)(slab + li * T + lj) = *reinterpret_cast<const f32x4*>(Ct + li * EL + lj);
        }
        return;
    }
    gfloat* Ab = PLEAS_GLOBAL_W(L.A) + ((size_t)it.rx * L.Cin + i0) * L.K + (size_t)it.ry * L.Cin + j0;
    if ((L.Cin & 3) == 0 && (((size_t)L.A) & 15) == 0) {      // 16-byte runs never straddle the block's edge
        f32x4 old[VPT];
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int v = tid + q * nThreads, li = v / (T / 4), lj = (v % (T / 4)) * 4;
            const bool in = i0 + li < L.Cin && j0 + lj < L.Cin;
            old[q] = *(const __attribute__((address_space(1))) f32x4*)(Ab + (in ? (size_t)li * L.K + lj : 0));
        }
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int v = tid + q * nThreads, li = v / (T / 4), lj = (v % (T / 4)) * 4;
            if (i0 + li < L.Cin && j0 + lj < L.Cin) {
                const f32x4 add = *reinterpret_cast<const f32x4*>(Ct + li * EL + lj);
                *(__attribute__((address_space(1))) f32x4*)(Ab + (size_t)li * L.K + lj) = old[q] + add;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int v = tid + q * nThreads, li = v / (T / 4), lj = (v % (T / 4)) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i0 + li < L.Cin && j0 + lj + e < L.Cin) Ab[(size_t)li * L.K + lj + e] += Ct[li * EL + lj + e];
        }
    }
}

template <int T, int VEC, bool SHIFT>
__device__ __forceinline__ void neq_tile(const NeqLayerDev& L, const NeqItemDev& it, float* smem) {
    constexpr int MT = T / 64;
    constexpr int LPR = nBK / VEC, RPP = nThreads / LPR, PASS = T / RPP;
    float* As = smem;
    float* Bs = smem + 2 * T * nLds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = it.tm * T, j0 = it.tn * T;
    const uint32_t HWi = (uint32_t)L.Hin * L.Win;
    const int khx = it.rx / L.KW, kwx = it.rx - khx * L.KW, khy = it.ry / L.KW, kwy = it.ry - khy * L.KW;
    const int dhx = khx - L.pad, dwx = kwx - L.pad, dhy = khy - L.pad, dwy = kwy - L.pad;
    const int srow = tid / LPR, scol = (tid % LPR) * VEC;
    float ra[PASS][VEC], rb[PASS][VEC];
    unsigned oka = 0, okb = 0;
    uint32_t offa[PASS], offb[PASS];
#pragma unroll
    for (int q = 0; q < PASS; ++q) {
        const int gi = i0 + srow + q * RPP, gj = j0 + srow + q * RPP;
        if (gi < L.Cin) oka |= 1u << q;
        if (gj < L.Cin) okb |= 1u << q;
        offa[q] = (uint32_t)min(gi, L.Cin - 1) * HWi;
        offb[q] = (uint32_t)min(gj, L.Cin - 1) * HWi;
    }
    f32x16 acc[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    bool ina = false, inb = false;
    auto load_chunk = [&](int c) {
        const uint32_t P = (uint32_t)c * nBK + scol;
        const bool in = P < L.Ktot;
        const uint32_t n = in ? P / L.HWo : 0u;
        const uint32_t p = in ? P - n * L.HWo : 0u;
        size_t basea, baseb;
        if constexpr (SHIFT) {
            const int oh = (int)(p / (uint32_t)L.Wout), ow = (int)(p - (uint32_t)oh * L.Wout);
            const int iha = oh * L.stride + dhx, iwa = ow * L.stride + dwx;
            const int ihb = oh * L.stride + dhy, iwb = ow * L.stride + dwy;
            ina = in && iha >= 0 && iha < L.Hin && iwa >= 0 && iwa < L.Win;
            inb = in && ihb >= 0 && ihb < L.Hin && iwb >= 0 && iwb < L.Win;
            basea = (size_t)n * L.Cin * HWi + (ina ? (size_t)iha * L.Win + iwa : 0);
            baseb = (size_t)n * L.Cin * HWi + (inb ? (size_t)ihb * L.Win + iwb : 0);
        } else {
            ina = inb = in;
            basea = baseb = (size_t)n * L.Cin * HWi + p;
        }
#pragma unroll
        for (int q = 0; q < PASS; ++q) {
            if constexpr (VEC == 4) {
                const f32x4 va = *(const __attribute__((address_space(1))) f32x4*)(PLEAS_GLOBAL(L.ip) + basea + offa[q]);
                const f32x4 vb = *(const __attribute__((address_space(1))) f32x4*)(PLEAS_GLOBAL(L.ip) + baseb + offb[q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ra[q][e] = va[e];
                    rb[q][e] = vb[e];
                }
            } else {
                ra[q][0] = PLEAS_GLOBAL(L.ip)[basea + offa[q]];
                rb[q][0] = PLEAS_GLOBAL(L.ip)[baseb + offb[q]];
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float* a = As + buf * T * nLds;
        float* b = Bs + buf * T * nLds;
#pragma unroll
        for (int q = 0; q < PASS; ++q) {
            const bool fa = ina && ((oka >> q) & 1u), fb = inb && ((okb >> q) & 1u);
            const int row = srow + q * RPP;
            if constexpr (VEC == 4) {
                f32x4 va = {fa ? ra[q][0] : 0.f, fa ? ra[q][1] : 0.f, fa ? ra[q][2] : 0.f, fa ? ra[q][3] : 0.f};
                f32x4 vb = {fb ? rb[q][0] : 0.f, fb ? rb[q][1] : 0.f, fb ? rb[q][2] : 0.f, fb ? rb[q][3] : 0.f};
                *reinterpret_cast<f32x4*>(a + row * nLds + scol) = va;
                *reinterpret_cast<f32x4*>(b + row * nLds + scol) = vb;
            } else {
                a[row * nLds + scol] = fa ? ra[q][0] : 0.f;
                b[row * nLds + scol] = fb ? rb[q][0] : 0.f;
            }
        }
    };
    auto compute = [&](int buf) {
        const float* a = As + buf * T * nLds + (wm * (T / 2) + (lane & 31)) * nLds + 4 * (lane >> 5);
        const float* b = Bs + buf * T * nLds + (wn * (T / 2) + (lane & 31)) * nLds + 4 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < nBK / 8; ++kk) {
            f32x4 fa[MT], fb[MT];
#pragma unroll
            for (int s = 0; s < MT; ++s) {
                fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * nLds + kk * 8);
                fb[s] = *reinterpret_cast<const f32x4*>(b + s * 32 * nLds + kk * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sm = 0; sm < MT; ++sm)
#pragma unroll
                    for (int sn = 0; sn < MT; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
        }
    };
    if (it.c_begin < it.c_end) {
        load_chunk(it.c_begin);
        store_chunk(0);
    }
    __syncthreads();
    for (int c = it.c_begin; c < it.c_end; ++c) {
        const int buf = (c - it.c_begin) & 1;
        const bool more = c + 1 < it.c_end;
        if (more) load_chunk(c + 1);
        compute(buf);
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }
    neq_store_tile<T>(L, it, acc, smem);
}

// ---- lag-class tile (stride 1, "same" padding): item (rx, ry) is the CANONICAL block of its class -----------------
template <int T, int VEC>
__device__ __forceinline__ void neq_lag_tile(const NeqLayerDev& L, const NeqItemDev& it, float* smem) {
    constexpr int MT = T / 64;
    constexpr int LPR = nBK / VEC, RPP = nThreads / LPR, PASS = T / RPP;
    float* As = smem;
    float* Bs = smem + 2 * T * nLds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = it.tm * T, j0 = it.tn * T;
    const int HW = L.Hin * L.Win;
    const int khx = it.rx / L.KW, kwx = it.rx - khx * L.KW, khy = it.ry / L.KW, kwy = it.ry - khy * L.KW;
    const int dhx = khx - L.pad, dwx = kwx - L.pad, dhy = khy - L.pad, dwy = kwy - L.pad;
    const int delta = (dhy - dhx) * L.Win + (dwy - dwx);     // operand y reads delta floats further on
    // the window Q of q = o + d_x: o, q and q + delta inside the image
    const int qh0 = max(0, max(-dhx, -dhy)) + dhx, qh1 = L.Hin - 1 - max(0, max(dhx, dhy)) + dhx;
    const int qw0 = max(0, max(-dwx, -dwy)) + dwx, qw1 = L.Win - 1 - max(0, max(dwx, dwy)) + dwx;
    const int srow = tid / LPR, scol = (tid % LPR) * VEC;
    unsigned oka = 0, okb = 0;
    int offa[PASS], offb[PASS];
#pragma unroll
    for (int q = 0; q < PASS; ++q) {
        const int gi = i0 + srow + q * RPP, gj = j0 + srow + q * RPP;
        if (gi < L.Cin) oka |= 1u << q;
        if (gj < L.Cin) okb |= 1u << q;
        offa[q] = min(gi, L.Cin - 1) * HW;
        offb[q] = min(gj, L.Cin - 1) * HW;
    }
    f32x16 acc[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 ra[PASS], rb0[PASS];                 // VEC == 1: element 0 only
    unsigned win = 0;                          // bit e: pixel e of this thread's run is inside Q
    auto load_chunk = [&](int c) {
        const uint32_t P = (uint32_t)c * nBK + scol;
        const bool in = P < L.Ktot;
        const int n = in ? (int)(P / (uint32_t)HW) : 0;
        const int p = in ? (int)(P - (uint32_t)n * HW) : 0;
        int qh = p / L.Win, qw = p - qh * L.Win;
        win = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (in && qh >= qh0 && qh <= qh1 && qw >= qw0 && qw <= qw1) win |= 1u << e;
            if (++qw == L.Win) { qw = 0; ++qh; }
        }
        const int base = n * L.Cin * HW + p;
        if constexpr (VEC == 4) {
#pragma unroll
            for (int q = 0; q < PASS; ++q) {
                ra[q] = *(const __attribute__((address_space(1))) f32x4*)(PLEAS_GLOBAL(L.ip) + base + offa[q]);
                // ONE 16-byte load at a 4-byte-aligned address.  Whatever lies outside the tensor lies outside its image,
                // hence outside Q: only the run that straddles the tensor's first / last float goes element by element
                const int g = base + delta + offb[q];
                if (g >= 0 && g <= L.total - 4) {
                    rb0[q] = *(const __attribute__((address_space(1))) f32x4u*)(PLEAS_GLOBAL(L.ip) + g);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb0[q][e] = PLEAS_GLOBAL(L.ip)[min(max(g + e, 0), L.total - 1)];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < PASS; ++q) {
                ra[q][0] = PLEAS_GLOBAL(L.ip)[base + offa[q]];
                rb0[q][0] = PLEAS_GLOBAL(L.ip)[min(max(base + delta + offb[q], 0), L.total - 1)];
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float* a = As + buf * T * nLds;
        float* b = Bs + buf * T * nLds;
#pragma unroll
        for (int q = 0; q < PASS; ++q) {
            const bool fa = (oka >> q) & 1u, fb = (okb >> q) & 1u;
            const int row = srow + q * RPP;
            if constexpr (VEC == 4) {
                f32x4 va = {(fa && (win & 1u)) ? ra[q][0] : 0.f, (fa && (win & 2u)) ? ra[q][1] : 0.f,
                            (fa && (win & 4u)) ? ra[q][2] : 0.f, (fa && (win & 8u)) ? ra[q][3] : 0.f};
                f32x4 vb = rb0[q];
                if (!fb) vb = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(a + row * nLds + scol) = va;
                *reinterpret_cast<f32x4*>(b + row * nLds + scol) = vb;
            } else {
                a[row * nLds + scol] = (fa && (win & 1u)) ? ra[q][0] : 0.f;
                b[row * nLds + scol] = fb ? rb0[q][0] : 0.f;
            }
        }
    };
    auto compute = [&](int buf) {
        const float* a = As + buf * T * nLds + (wm * (T / 2) + (lane & 31)) * nLds + 4 * (lane >> 5);
        const float* b = Bs + buf * T * nLds + (wn * (T / 2) + (lane & 31)) * nLds + 4 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < nBK / 8; ++kk) {
            f32x4 fa[MT], fb[MT];
#pragma unroll
            for (int s = 0; s < MT; ++s) {
                fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * nLds + kk * 8);
                fb[s] = *reinterpret_cast<const f32x4*>(b + s * 32 * nLds + kk * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sm = 0; sm < MT; ++sm)
#pragma unroll
                    for (int sn = 0; sn < MT; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
        }
    };
    if (it.c_begin < it.c_end) {
        load_chunk(it.c_begin);
        store_chunk(0);
    }
    __syncthreads();
    for (int c = it.c_begin; c < it.c_end; ++c) {
        const int buf = (c - it.c_begin) & 1;
        const bool more = c + 1 < it.c_end;
        if (more) load_chunk(c + 1);
        compute(buf);
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }
    neq_store_tile<T>(L, it, acc, smem);
}

__global__ __launch_bounds__(nThreads) void neq_batch_kernel(const NeqLayerDev* __restrict__ layers,
                                                             const NeqItemDev* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const NeqItemDev it = items[blockIdx.x];
    if (it.layer < 0) return;   // padding of the XCD-aware item order
    const NeqLayerDev L = layers[it.layer];
    switch (L.variant) {
        case 0: neq_tile<128, 4, false>(L, it, smem); break;
        case 1: neq_tile<64, 4, false>(L, it, smem); break;
        case 2: neq_tile<128, 1, false>(L, it, smem); break;
        case 3: neq_tile<64, 1, false>(L, it, smem); break;
        case 6: neq_tile<128, 1, true>(L, it, smem); break;
        case 7: neq_tile<64, 1, true>(L, it, smem); break;
        case 8: neq_lag_tile<128, 4>(L, it, smem); break;
        case 9: neq_lag_tile<64, 4>(L, it, smem); break;
        case 10: neq_lag_tile<128, 1>(L, it, smem); break;
        default: neq_lag_tile<64, 1>(L, it, smem); break;
    }
}

// A += sum over splits of the slabs (one block per (slot) tile, threads over the T*T tile)
struct NeqReduceDev {
    int layer, tm, tn, rx, ry, slot, T, pad;
};
constexpr int nRedParts = 16;   // workgroups per tile: a tile's T*T elements are independent
__global__ __launch_bounds__(256) void neq_reduce_kernel(const NeqLayerDev* __restrict__ layers,
                                                         const NeqReduceDev* __restrict__ red) {
    const NeqReduceDev rd = red[blockIdx.x / nRedParts];
    const NeqLayerDev L = layers[rd.layer];
    const int TT = rd.T * rd.T, part = blockIdx.x % nRedParts, per = TT / nRedParts;
    for (int e = part * per + threadIdx.x; e < (part + 1) * per; e += blockDim.x) {
        const int li = e / rd.T, lj = e - li * rd.T;
        const int gi = rd.tm * rd.T + li, gj = rd.tn * rd.T + lj;
        if (gi >= L.Cin || gj >= L.Cin) continue;
        float s = 0.f;
        for (int k = 0; k < L.S; ++k) s += PLEAS_GLOBAL(L.slab)[((size_t)rd.slot * L.S + k) * TT + e];   // fixed order
        L.A[((size_t)rd.rx * L.Cin + gi) * L.K + (size_t)rd.ry * L.Cin + gj] += s;
    }
}

// ---- lag classes: which blocks of the lower triangle are contracted, which are copies (host side) -------------------
// 1-D class of the tap pair (dx, dy) along an axis of extent n: (delta, first q, last q) with q = o + dx.
struct LagAxis { int delta, lo, hi; };
static inline LagAxis lag_axis(int dx, int dy, int n) {
    return LagAxis{dy - dx, std::max(0, std::max(-dx, -dy)) + dx, n - 1 - std::max(0, std::max(dx, dy)) + dx};
}
struct LagKey {
    int v[6];
    bool operator==(const LagKey& o) const { return std::equal(v, v + 6, o.v); }
};
static inline LagKey lag_key(int rx, int ry, int KW, int pad, int H, int W) {
    const LagAxis h = lag_axis(rx / KW - pad, ry / KW - pad, H), w = lag_axis(rx % KW - pad, ry % KW - pad, W);
    return LagKey{{h.delta, h.lo, h.hi, w.delta, w.lo, w.hi}};
}
// C(delta, Q)^T = C(-delta, Q + delta)
static inline LagKey lag_transposed(const LagKey& k) {
    return LagKey{{-k.v[0], k.v[1] + k.v[0], k.v[2] + k.v[0], -k.v[3], k.v[4] + k.v[3], k.v[5] + k.v[3]}};
}
struct NeqCopy { int rx, ry, sx, sy, transposed; };   // block (rx, ry) := block (sx, sy) [transposed]
// Lower-triangle blocks (rx >= ry) of a lag-form layer: `canon` are contracted, `copies` are filled by finalize.
static void lag_blocks(int R, int KW, int pad, int H, int W, std::vector<std::pair<int, int>>& canon, std::vector<NeqCopy>& copies) {
    std::vector<LagKey> keys;
    for (int rx = 0; rx < R; ++rx)
        for (int ry = 0; ry <= rx; ++ry) {
            const LagKey k = lag_key(rx, ry, KW, pad, H, W), kt = lag_transposed(k);
            int hit = -1, tr = 0;
            for (size_t c = 0; c < keys.size() && hit < 0; ++c) {
                if (keys[c] == k) hit = (int)c, tr = 0;
                else if (keys[c] == kt) hit = (int)c, tr = 1;
            }
            // a diagonal block (delta = 0: symmetric, lower tiles only) is never a source; with rx == ry it never is a copy
            if (hit >= 0 && canon[hit].first != canon[hit].second && rx != ry) {
                copies.push_back(NeqCopy{rx, ry, canon[hit].first, canon[hit].second, tr});
            } else {
                keys.push_back(k);
                canon.emplace_back(rx, ry);
            }
        }
}

constexpr int nCopyBatch = 64;
struct NeqCopyBatch {
    int count;
    NeqCopy c[nCopyBatch];
};
// A[(rx, i), (ry, j)] = A[(sx, i), (sy, j)]  or, transposed,  A[(sx, j), (sy, i)];  32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void neq_finalize_kernel(float* __restrict__ A, int Cin, int K, const NeqCopyBatch b) {
    __shared__ float tile[32][33];
    const NeqCopy c = b.c[blockIdx.z];
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        // source element that lands on (i0 + r, j0 + tx) -- read row-wise in the source either way
        const int si = c.transposed ? j0 + r : i0 + r, sj = c.transposed ? i0 + tx : j0 + tx;
        tile[r][tx] = (si < Cin && sj < Cin) ? A[((size_t)c.sx * Cin + si) * K + (size_t)c.sy * Cin + sj] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int di = i0 + r, dj = j0 + tx;
        if (di < Cin && dj < Cin)
            A[((size_t)c.rx * Cin + di) * K + (size_t)c.ry * Cin + dj] = c.transposed ? tile[tx][r] : tile[r][tx];
    }
}

static bool lag_form(const pleas_neq_layer& l) {
    const char* env = std::getenv("PLEAS_NEQ_LAG");       // PLEAS_NEQ_LAG=0: every block contracted (A/B experiments)
    if (env && env[0] == '0') return false;
    // geometry only -- NOT the batch size: accumulate (any batch) and finalize (no batch at hand) must classify a layer alike
    return l.KH == l.KW && l.KH > 1 && l.stride == 1 && 2 * l.pad == l.KH - 1;
}
// the lag tile addresses its input with 32-bit element offsets
static bool lag_fits(const pleas_neq_layer& l) { return (int64_t)l.N * l.Cin * l.Hin * l.Win < (1ll << 31); }

constexpr int nPtrBatch = 224;
struct NeqPtrBatch {
    int base, count;
    const float* ip[nPtrBatch];
    float* A[nPtrBatch];
};
__global__ void neq_set_ptrs_kernel(NeqLayerDev* __restrict__ layers, const NeqPtrBatch b) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < b.count) {
        layers[b.base + t].ip = b.ip[t];
        layers[b.base + t].A = b.A[t];
    }
}

static int g_neq_item_chunks = 112;

struct NeqPlan {
    std::vector<int64_t> key;
    std::vector<NeqLayerDev> layers;
    std::vector<NeqItemDev> items;
    std::vector<NeqReduceDev> red;
    size_t off_layers = 0, off_items = 0, off_red = 0, off_slabs = 0, total = 0, lds = 0;
    double flops = 0, flops_exec = 0, bytes = 0;
    int64_t n_copies = 0;
    bool uploaded = false;
};
static PlanCache<NeqPlan, 1> g_nplans;
static std::mutex g_nplan_mu;
static size_t nalign(size_t v) { return (v + 255) / 256 * 256; }

static int build_neq_plan(NeqPlan& P, const pleas_neq_layer* ly, int n) {
    P.layers.assign(n, NeqLayerDev());
    P.items.clear();
    P.red.clear();
    P.flops = P.flops_exec = P.bytes = 0;
    P.n_copies = 0;
    P.lds = 0;
    std::vector<size_t> slab_off(n, 0);
    size_t slabs = 0;
    // XCD-aware order (common.hpp), OFF by default for this kernel (PLEAS_XCD_ORDER=1 turns it on): every item of one
    // (layer, K range) reads the same pixel range of the SAME input tensor (ResNet-101: 0.9 - 3.2 MB, fits the 4 MB L2 of
    // one XCD), so they can all go to one XCD.  Measured (round 3, tools/r03/r03_run2.sh): FETCH_SIZE x 2 falls from 8.8 GB to
    // 3.1 GB per batch (1.0 GB of inputs), the launch takes 5.89 instead of 5.72 ms -- and the six strided layers alone 1.79
    // instead of 0.83 ms, because a layer's ~200 items no longer spread over the eight XCDs.  MFMA-issue bound, not L2 bound.
    std::vector<XcdWork<NeqItemDev>> work;
    for (int i = 0; i < n; ++i) {
        const pleas_neq_layer& l = ly[i];
        if (l.N <= 0 || l.Cin <= 0 || l.Hin <= 0 || l.Win <= 0 || l.KH <= 0 || l.KW <= 0 || l.stride <= 0 || l.pad < 0)
            return bad_arg("normal_eq: layer geometry");
        const int Hout = (l.Hin + 2 * l.pad - l.KH) / l.stride + 1, Wout = (l.Win + 2 * l.pad - l.KW) / l.stride + 1;
        if (Hout <= 0 || Wout <= 0) return bad_arg("normal_eq: empty output");
        const int64_t HWo = (int64_t)Hout * Wout, K = (int64_t)l.N * HWo;
        if (K >= (1ll << 31)) return bad_arg("normal_eq: N*Hout*Wout must be < 2^31");
        NeqLayerDev& d = P.layers[i];
        d.Cin = l.Cin; d.Hin = l.Hin; d.Win = l.Win; d.Hout = Hout; d.Wout = Wout;
        d.KH = l.KH; d.KW = l.KW; d.stride = l.stride; d.pad = l.pad;
        d.HWo = (uint32_t)HWo;
        d.Ktot = (uint32_t)K;
        const int R = l.KH * l.KW;
        d.K = R * l.Cin;
        const int T = l.Cin > 64 ? 128 : 64;
        const bool direct = R == 1 && l.stride == 1 && l.pad == 0;
        const bool lag = !direct && lag_form(l);
        if (lag && !lag_fits(l))      // an error, not another form: pleas_normal_eq_finalize classifies by geometry alone
            return bad_arg("normal_eq: a stride-1 k x k layer's input must hold fewer than 2^31 elements per call");
        const bool vec = (direct || lag) && HWo % 4 == 0;
        d.variant = (T == 64 ? 1 : 0) | (vec ? 0 : 2) | (direct ? 0 : lag ? 8 : 4);
        d.total = lag ? (int)((int64_t)l.N * l.Cin * l.Hin * l.Win) : 0;
        const int nchunks = (int)ceil_div(K, nBK);
        const int S = (int)ceil_div(nchunks, g_neq_item_chunks);
        const int cps = (int)ceil_div(nchunks, S);
        d.S = S;
        P.lds = std::max(P.lds, (size_t)4 * T * nLds * sizeof(float));
        const int tiles = (int)ceil_div(l.Cin, T);
        int slot = 0;
        slab_off[i] = slabs;
        // lag form: only one block per class (up to transposition) is contracted; pleas_normal_eq_finalize fills the rest
        std::vector<std::pair<int, int>> canon;
        std::vector<NeqCopy> copies;
        if (lag) lag_blocks(R, l.KW, l.pad, l.Hin, l.Win, canon, copies);
        P.n_copies += (int64_t)copies.size();
        auto contracted = [&](int rx, int ry) {
            return !lag || std::find(canon.begin(), canon.end(), std::make_pair(rx, ry)) != canon.end();
        };
        int64_t tiles_done = 0;
        for (int rx = 0; rx < R; ++rx)
            for (int tm = 0; tm < tiles; ++tm)
                for (int ry = 0; ry < R; ++ry)
                    for (int tn = 0; tn < tiles; ++tn) {
                        if (ry * tiles + tn > rx * tiles + tm) continue;  // lower triangle of block tiles only
                        if (!contracted(rx, ry)) continue;
                        for (int s = 0; s < S; ++s) {
                            XcdWork<NeqItemDev> w;
                            w.it = NeqItemDev{i, tm, tn, rx, ry, s, s * cps, std::min((s + 1) * cps, nchunks), slot, 0};
                            w.w = (double)(w.it.c_end - w.it.c_begin) * T * T;
                            w.key = (int64_t)i * 4096 + s;
                            work.push_back(w);
                        }
                        if (S > 1) P.red.push_back(NeqReduceDev{i, tm, tn, rx, ry, slot, T, 0});
                        ++slot;
                        ++tiles_done;
                    }
        if (S > 1) slabs += (size_t)slot * S * T * T;
        P.flops += (double)d.K * d.K * (double)K;  // the path's work -- lower triangle: half of 2 K^2 P
        // what the grid executes: whole tiles (ragged ones and the diagonal's upper halves included), lag copies left out
        P.flops_exec += 2.0 * (double)tiles_done * T * T * (double)nchunks * nBK;
        P.bytes += (double)l.Cin * l.N * l.Hin * l.Win * sizeof(float);
    }
    P.items = xcd_order_items(work, NeqItemDev{-1, 0, 0, 0, 0, 0, 0, 0, 0, 0}, /*by_default=*/false);
    size_t off = 0;
    P.off_layers = off;
    off = nalign(off + P.layers.size() * sizeof(NeqLayerDev));
    P.off_items = off;
    off = nalign(off + P.items.size() * sizeof(NeqItemDev));
    P.off_red = off;
    off = nalign(off + P.red.size() * sizeof(NeqReduceDev));
    P.off_slabs = off;
    P.total = off + slabs * sizeof(float);
    for (int i = 0; i < n; ++i) P.layers[i].slab = reinterpret_cast<float*>(slab_off[i]);
    P.uploaded = false;
    return PLEAS_OK;
}

}  // namespace pleas

using namespace pleas;

extern "C" size_t pleas_normal_eq_ws_bytes(const pleas_neq_layer* layers, int n_layers) {
    if (!layers || n_layers <= 0) return 0;
    NeqPlan tmp;
    if (build_neq_plan(tmp, layers, n_layers) != PLEAS_OK) return 0;
    return tmp.total;
}

extern "C" int pleas_normal_eq_plan_info(const pleas_neq_layer* layers, int n_layers, double* info) {
    if (!layers || n_layers <= 0 || !info) return bad_arg("normal_eq_plan_info: empty layer list");
    NeqPlan tmp;
    const int rc = build_neq_plan(tmp, layers, n_layers);
    if (rc != PLEAS_OK) return rc;
    info[0] = tmp.flops;
    info[1] = tmp.flops_exec;
    info[2] = (double)std::count_if(tmp.items.begin(), tmp.items.end(), [](const NeqItemDev& it) { return it.layer >= 0; });
    info[3] = (double)tmp.n_copies;
    return PLEAS_OK;
}

extern "C" int pleas_normal_eq_finalize(const pleas_neq_layer* layers, int n_layers, void* stream_) {
    if (!layers || n_layers <= 0) return bad_arg("normal_eq_finalize: empty layer list");
    hipStream_t stream = (hipStream_t)stream_;
    for (int i = 0; i < n_layers; ++i) {
        const pleas_neq_layer& l = layers[i];
        if (!l.A) return bad_arg("normal_eq_finalize: null pointer");
        if (l.N <= 0 || l.Cin <= 0 || l.Hin <= 0 || l.Win <= 0 || l.KH <= 0 || l.KW <= 0 || l.stride <= 0 || l.pad < 0)
            return bad_arg("normal_eq_finalize: layer geometry");
        const bool direct = l.KH * l.KW == 1 && l.stride == 1 && l.pad == 0;
        if (direct || !lag_form(l)) continue;          // every block of its lower triangle was contracted
        std::vector<std::pair<int, int>> canon;
        std::vector<NeqCopy> copies;
        lag_blocks(l.KH * l.KW, l.KW, l.pad, l.Hin, l.Win, canon, copies);
        const unsigned t32 = (unsigned)ceil_div(l.Cin, 32);
        for (size_t c0 = 0; c0 < copies.size(); c0 += nCopyBatch) {
            NeqCopyBatch b;
            b.count = (int)std::min<size_t>(nCopyBatch, copies.size() - c0);
            std::copy(copies.begin() + c0, copies.begin() + c0 + b.count, b.c);
            hipLaunchKernelGGL(neq_finalize_kernel, dim3(t32, t32, (unsigned)b.count), dim3(256), 0, stream, l.A, l.Cin,
                               l.KH * l.KW * l.Cin, b);
            PLEAS_LAUNCH_CHECK("neq_finalize_kernel");
        }
    }
    return PLEAS_OK;
}

extern "C" int pleas_normal_eq_accum(const pleas_neq_layer* layers, int n_layers, void* ws, size_t ws_bytes, int ws_fresh,
                                     void* stream_) {
    if (!layers || n_layers <= 0) return bad_arg("normal_eq: empty layer list");
    for (int i = 0; i < n_layers; ++i) {
        if (!layers[i].ip || !layers[i].A) return bad_arg("normal_eq: null pointer");
        if (((uintptr_t)layers[i].ip & 15) != 0) return bad_arg("normal_eq: 16-byte alignment");
    }
    hipStream_t stream = (hipStream_t)stream_;
    std::lock_guard<std::mutex> lk(g_nplan_mu);
    std::vector<int64_t> key;
    key.push_back(n_layers);
    key.push_back((int64_t)(uintptr_t)ws);
    key.push_back(g_neq_item_chunks);
    for (int i = 0; i < n_layers; ++i)
        for (int v : {layers[i].N, layers[i].Cin, layers[i].Hin, layers[i].Win, layers[i].KH, layers[i].KW, layers[i].stride,
                      layers[i].pad})
            key.push_back(v);
    NeqPlan* hit = g_nplans.find(key);
    if (!hit) {
        hit = &g_nplans.take();
        const int rc = build_neq_plan(*hit, layers, n_layers);
        if (rc != PLEAS_OK) return rc;
        hit->key.swap(key);
    }
    NeqPlan& P = *hit;
    if (ws_fresh) P.uploaded = false;
    if (!ws || ws_bytes < P.total) {
        std::snprintf(g_last_error, sizeof(g_last_error), "normal_eq workspace too small: need %zu bytes", P.total);
        P.key.clear();
        return PLEAS_ENOMEM;
    }
    char* base = (char*)ws;
    if (!P.uploaded) {
        g_nplans.claims_workspace(P);
        float* slab0 = reinterpret_cast<float*>(base + P.off_slabs);
        std::vector<NeqLayerDev> abs_layers = P.layers;
        for (auto& d : abs_layers) d.slab = slab0 + reinterpret_cast<size_t>(d.slab);
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_layers, abs_layers.data(), abs_layers.size() * sizeof(NeqLayerDev),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_items, P.items.data(), P.items.size() * sizeof(NeqItemDev),
                                       hipMemcpyHostToDevice, stream));
        if (!P.red.empty())
            PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_red, P.red.data(), P.red.size() * sizeof(NeqReduceDev),
                                           hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipStreamSynchronize(stream));
        P.uploaded = true;
    }
    NeqLayerDev* dl = reinterpret_cast<NeqLayerDev*>(base + P.off_layers);
    for (int b0 = 0; b0 < n_layers; b0 += nPtrBatch) {
        NeqPtrBatch pb;
        pb.base = b0;
        pb.count = std::min(nPtrBatch, n_layers - b0);
        for (int t = 0; t < pb.count; ++t) {
            pb.ip[t] = layers[b0 + t].ip;
            pb.A[t] = layers[b0 + t].A;
        }
        hipLaunchKernelGGL(neq_set_ptrs_kernel, dim3(1), dim3(256), 0, stream, dl, pb);
        PLEAS_LAUNCH_CHECK("neq_set_ptrs_kernel");
    }
    ProfScope prof(kProfNormalEq, P.flops, P.bytes, stream);
    hipLaunchKernelGGL(neq_batch_kernel, dim3((unsigned)P.items.size()), dim3(nThreads), P.lds, stream, dl,
                       reinterpret_cast<const NeqItemDev*>(base + P.off_items));
    PLEAS_LAUNCH_CHECK("neq_batch_kernel");
    if (!P.red.empty()) {
        hipLaunchKernelGGL(neq_reduce_kernel, dim3((unsigned)P.red.size() * nRedParts), dim3(256), 0, stream, dl,
                           reinterpret_cast<const NeqReduceDev*>(base + P.off_red));
        PLEAS_LAUNCH_CHECK("neq_reduce_kernel");
    }
    return PLEAS_OK;
}

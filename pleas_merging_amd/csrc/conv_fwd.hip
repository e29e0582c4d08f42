// PLeaS layer fitting on gfx950: forward of ALL merged layers of one update + regression target +
// residual + loss in one grouped fp32-MFMA launch.
//
// Replaces, per layer and per update (pleas/methods/pleas_merging.py):
//   :281  out   = layer(ip)                              (vendor conv / linear)
//   :116-123, :147  op = block-merge of the two source layers' outputs   (index_select x4 + cat)
//   :282  loss  = mean((out - op)^2)
//   first node of :287   resid = 2 (out - op) / numel
// `out` and `op` are never written: the epilogue gathers/averages the source outputs, forms the
// residual in registers, stores ONLY the residual (input of pleas_wgrad_batch) and one partial
// sum of squares per workgroup.
//
// Formulation: implicit GEMM  out[co][P] = sum_k W[co][k] * U[k][P],  P = (n, oh, ow) flattened,
// k = (ci, kh, kw) in the standard weight order.  A workgroup owns TM output channels x 128 pixels;
// W rows are K-contiguous (A operand, 16-B LDS reads feeding four MFMA steps, as in gram.hip);
// U is read in place from the NCHW input: a thread owns ONE pixel for the whole K loop (its (n, oh, ow)
// is decoded once) and walks (ci, kh, kw) incrementally, so consecutive lanes read consecutive
// addresses for stride-1 layers; its 16 values of a chunk are one contiguous run of the [pixel][k] LDS
// tile (four 16-B stores), so BOTH operands are read back with the 16-B / four-MFMA-step pattern of
// gram.hip.  The epilogue goes through LDS once to turn the accumulator layout (one pixel per lane)
// into 16-B runs along the pixel axis for the target gathers and the residual stores.
// Items of all layers are sorted longest-first into one grid.
//
// Work: 2*Cout*Cin*KH*KW*N*HWo flop per layer (same as the weight gradient), fp32-MFMA bound.
#include <algorithm>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace pleas {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef PLEAS_FWD_ABLATE
#define PLEAS_FWD_ABLATE 0   // experiments only (tools/hipbench/run_fwd_ablate.sh): 1 no target gathers, 2 no epilogue,
#endif                       // 4 no input-operand loads, 8 no weight loads, 16 per-item cycle stamps (pleas_fwd_debug_read)
constexpr int fBK = 32;
constexpr int fLdsA = 36;    // W tile rows: [TM][36]   (k contiguous)
constexpr int fLdsB = 36;    // U tile rows: [128 pixels][36] (k contiguous: a thread's 16 k values of its pixel are one run)
constexpr int fTN = 128;
constexpr int fThreads = 256;

struct FwdLayerDev {
    const float* ip;      // merged input  [N][Cin][Hin][Win]
    const float* w;       // [Cout][Kd], Kd = Cin*KH*KW
    const float* bias;    // [Cout] or null
    const float* o1;      // source outputs [N][Csrc][HWo]
    const float* o2;
    const int32_t* row1;  // [Cout] block maps of the output axis (-1 = absent)
    const int32_t* row2;
    float* resid;         // [N][Cout][HWo]
    int Cout, Cin, Hin, Win, Hout, Wout, KH, KW, stride, pad, Csrc, n_merged;
    uint32_t HWo, Ptot, Kd;
    float dscale;         // 2 / (numel * world)
    int variant;          // bit0: TM == 64, bit1: scalar W loads, bit2: W is kernel-position-major [Cout][KH*KW][Cin]
    int part_base;        // first loss-partial slot of this layer
    int pad0;
};
struct FwdItemDev {
    int layer, tm, tp, slot;  // slot: loss-partial index
};

#if (PLEAS_FWD_ABLATE & 16)
__device__ long long g_fwd_stamps[32768][4];   // per work item: prologue, K loop, epilogue cycles, chunks
#define PLEAS_FWD_STAMP(var) const long long var = clock64()
#else
#define PLEAS_FWD_STAMP(var)
#endif

template <int TM, int VECA>
__device__ __forceinline__ void fwd_tile(const FwdLayerDev& L, const FwdItemDev& it, float* smem, float* __restrict__ partials) {
    constexpr int MTM = TM / 64;
    constexpr int LPR = fBK / VECA, RPP = fThreads / LPR, PASS = TM / RPP;
    float* As = smem;                      // [2][TM][fLdsA]
    float* Bs = smem + 2 * TM * fLdsA;     // [2][fTN][fLdsB]
    PLEAS_FWD_STAMP(st0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = it.tm * TM;
    const uint32_t p0 = (uint32_t)it.tp * fTN;
    const uint32_t HWi = (uint32_t)L.Hin * L.Win;
    const int R = L.KH * L.KW;
    const bool kpos = (L.variant & 4) != 0;
    const int nchunks = (int)((L.Kd + fBK - 1) / fBK);

    // ---- A (weights) staging: rows co, k contiguous
    const int arow = tid / LPR, acol = (tid % LPR) * VECA;
    float ra[PASS][VECA];
    unsigned oka = 0;
    uint32_t offa[PASS];
#pragma unroll
    for (int q = 0; q < PASS; ++q) {
        const int gi = i0 + arow + q * RPP;
        if (gi < L.Cout) oka |= 1u << q;
        offa[q] = (uint32_t)min(gi, L.Cout - 1) * L.Kd;
    }
    // ---- B (input) staging: this thread's pixel and half of the chunk's k rows
    const int bpix = tid & 127;
    const int bhalf = __builtin_amdgcn_readfirstlane(tid >> 7);  // wave-uniform: the k walk below stays on the scalar unit
    const uint32_t P = p0 + bpix;
    const bool pin = P < L.Ptot;
    const uint32_t pn = pin ? P / L.HWo : 0u, pp = pin ? P - pn * L.HWo : 0u;
    const int oh = (int)(pp / (uint32_t)L.Wout), ow = (int)(pp - (uint32_t)oh * L.Wout);
    const int ih0 = oh * L.stride - L.pad, iw0 = ow * L.stride - L.pad;
    const size_t pbase = (size_t)pn * L.Cin * HWi;
    float rb[16];
    unsigned okb = 0;
    bool kina = false;
    // per-pixel tap table for kernels larger than 1x1 (KH*KW <= 64, checked on the host)
    unsigned long long tapmask = 0;
    const long long pixoff = (long long)ih0 * L.Win + iw0;
    for (int r = 0; r < R; ++r) {
        const int kh = r / L.KW, kw = r - kh * L.KW;
        const bool ok = pin && ih0 + kh >= 0 && ih0 + kh < L.Hin && iw0 + kw >= 0 && iw0 + kw < L.Win;
        tapmask |= (unsigned long long)ok << r;
    }

    f32x16 acc[MTM][2];
#pragma unroll
    for (int a = 0; a < MTM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto load_chunk = [&](int c) {
        {
            // kernel-position-major weights: chunk c = (channel block c / R, tap c % R); its 32 k values sit at
            // [tap][block * 32 ..] of the weight row
            const uint32_t k = kpos ? (uint32_t)(c % R) * L.Cin + (uint32_t)(c / R) * fBK + acol : (uint32_t)c * fBK + acol;
            kina = k < L.Kd;
            const uint32_t kc = kina ? k : 0u;
#pragma unroll
            for (int q = 0; q < PASS; ++q) {
                if constexpr ((PLEAS_FWD_ABLATE & 8) != 0) {
#pragma unroll
                    for (int e = 0; e < VECA; ++e) ra[q][e] = (float)(kc + e);
                } else if constexpr (VECA == 4) {
                    const f32x4 v = *(const __attribute__((address_space(1))) f32x4*)(PLEAS_GLOBAL(L.w) + offa[q] + kc);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ra[q][e] = v[e];
                } else {
                    ra[q][0] = PLEAS_GLOBAL(L.w)[offa[q] + kc];
                }
            }
        }
        {
            // Every lane of a wave walks the SAME k values (wave-uniform ci, kh, kw); only the pixel differs.
            // Loads are unconditional: an out-of-range tap reads element 0 (offset masked to zero, no branch) and is
            // zeroed when the tile is written to LDS.
            const uint32_t k = (uint32_t)c * fBK + bhalf * 16;
            okb = 0;
            if (R == 1 || kpos) {
                // ONE tap per chunk (1x1 / Linear, or kernel-position-major weights: channel block c / R, tap c % R; the
                // taps of a channel block run back to back, so their shifted re-reads of the same input rows hit L1/L2).
                // Per thread: one masked base offset per chunk; per load: + a wave-uniform channel offset.
                const int cb = kpos ? c / R : c;
                const int r = kpos ? c - cb * R : 0;
                const int kh = r / L.KW, kw = r - kh * L.KW;
                const bool ok_tap = (tapmask >> r) & 1ull;
                const uint32_t ch0 = (uint32_t)cb * fBK + bhalf * 16;
                const long long voff = ok_tap ? (long long)pbase + pixoff + (kh * L.Win + kw) : 0ll;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint32_t ch = ch0 + q;
                    const bool ok = ok_tap && ch < (uint32_t)L.Cin;
                    okb |= (ok ? 1u : 0u) << q;
                    const long long lin = (long long)min(ch, (uint32_t)L.Cin - 1u) * HWi;   // scalar; masked lanes read ip[lin]
                    if constexpr ((PLEAS_FWD_ABLATE & 4) != 0) rb[q] = (float)((voff + lin) & 7); else
                    rb[q] = PLEAS_GLOBAL(L.ip)[voff + lin];
                }
            } else {
                // general kernel: the tap validity of this thread's pixel is a precomputed bit mask and its pixel
                // offset a constant; (ci, r, kh, kw) are wave-uniform and walk on the scalar unit
                int ci = (int)(k / (uint32_t)R);
                int r = (int)(k - (uint32_t)ci * R);
                int kh = r / L.KW, kw = r - kh * L.KW;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const bool ok = ((tapmask >> r) & 1ull) && (k + q) < L.Kd;
                    okb |= (ok ? 1u : 0u) << q;
                    const long long lin = (long long)ci * HWi + kh * L.Win + kw;   // scalar
                    const size_t off = (size_t)((long long)pbase + pixoff + lin) & (size_t)(-(long long)ok);
                    if constexpr ((PLEAS_FWD_ABLATE & 4) != 0) rb[q] = (float)(off & 7); else
                    rb[q] = PLEAS_GLOBAL(L.ip)[off];
                    ++r;
                    ++kw;
                    const int cw = kw == L.KW;
                    kw = cw ? 0 : kw;
                    kh += cw;
                    const int cr = r == R;
                    r = cr ? 0 : r;
                    kh = cr ? 0 : kh;
                    ci += cr;
                }
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float* a = As + buf * TM * fLdsA;
        float* b = Bs + buf * fTN * fLdsB;
#pragma unroll
        for (int q = 0; q < PASS; ++q) {
            const bool ok = kina && ((oka >> q) & 1u);
            const int row = arow + q * RPP;
            if constexpr (VECA == 4) {
                f32x4 v = {ok ? ra[q][0] : 0.f, ok ? ra[q][1] : 0.f, ok ? ra[q][2] : 0.f, ok ? ra[q][3] : 0.f};
                *reinterpret_cast<f32x4*>(a + row * fLdsA + acol) = v;
            } else {
                a[row * fLdsA + acol] = ok ? ra[q][0] : 0.f;
            }
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((okb >> (q4 * 4 + e)) & 1u) ? rb[q4 * 4 + e] : 0.f;
            *reinterpret_cast<f32x4*>(b + bpix * fLdsB + bhalf * 16 + q4 * 4) = v;
        }
    };
    auto compute = [&](int buf) {
        const float* a = As + buf * TM * fLdsA + (wm * (TM / 2) + (lane & 31)) * fLdsA + 4 * (lane >> 5);
        const float* b = Bs + buf * fTN * fLdsB + (wn * 64 + (lane & 31)) * fLdsB + 4 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < fBK / 8; ++kk) {
            f32x4 fa[MTM], fb[2];
#pragma unroll
            for (int s = 0; s < MTM; ++s) fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * fLdsA + kk * 8);
#pragma unroll
            for (int s = 0; s < 2; ++s) fb[s] = *reinterpret_cast<const f32x4*>(b + s * 32 * fLdsB + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sm = 0; sm < MTM; ++sm)
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
        }
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    PLEAS_FWD_STAMP(st1);
    // The block maps and biases of the epilogue are requested before the LAST chunk's MFMAs (no staging loads are in
    // flight then): their round trip is covered by that chunk instead of opening the epilogue.
    constexpr int ROWS = TM / 8, GB = 8, NB = ROWS / GB;
    int m1[ROWS], m2[ROWS];
    float bias_v[ROWS];
    auto load_maps = [&]() {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int co = min(i0 + (tid >> 5) + 8 * j, L.Cout - 1);
            m1[j] = PLEAS_GLOBAL_I(L.row1)[co];
            m2[j] = PLEAS_GLOBAL_I(L.row2)[co];
            bias_v[j] = L.bias ? PLEAS_GLOBAL(L.bias)[co] : 0.f;
        }
    };
    for (int c = 0; c + 1 < nchunks; ++c) {
        const int buf = c & 1;
        load_chunk(c + 1);
        compute(buf);
        store_chunk(buf ^ 1);
        __syncthreads();
    }
    load_maps();
    compute((nchunks - 1) & 1);
    __syncthreads();
    PLEAS_FWD_STAMP(st2);

    // ---- epilogue.  Accumulators hold one pixel per lane; go through LDS once ([co][pixel], stride 132) so that
    //      each thread then owns 4 consecutive pixels of one output channel: 16-B target gathers, 16-B residual stores.
    if constexpr ((PLEAS_FWD_ABLATE & 2) != 0) {   // keep the accumulators alive with one store per wave
        float s = 0.f;
#pragma unroll
        for (int sm = 0; sm < MTM; ++sm)
#pragma unroll
            for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[sm][sn][r];
        if (s == 12345.678f) partials[L.part_base + it.slot] = s;
        return;
    }
    constexpr int EL = 132;
    float* Ct = smem;  // [TM][EL] floats <= the staging buffers just released by the last barrier of the K loop
    auto spill_acc = [&]() {
#pragma unroll
        for (int sm = 0; sm < MTM; ++sm)
#pragma unroll
            for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lco = wm * (TM / 2) + sm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    Ct[lco * EL + wn * 64 + sn * 32 + (lane & 31)] = acc[sm][sn][r];
                }
        __syncthreads();
    };
    float sq = 0.f;
    const bool vec_ok = (L.HWo % 4 == 0);   // then a 4-pixel group never straddles samples and is 16-B aligned
    const int pg = (tid & 31) * 4;           // pixel group of this thread
    const uint32_t Pg = p0 + pg;
    const bool gin = Pg < L.Ptot;
    const uint32_t gn = gin ? Pg / L.HWo : 0u, gp = gin ? Pg - gn * L.HWo : 0u;
    // Each thread owns TM/8 output channels (lco = tid/32 + 8 j) x 4 pixels.  Order, chosen so that global-memory latency
    // is exposed once instead of once per step: block maps + bias -> first batch of target gathers (8 channels, branch
    // free: absent / out-of-range rows read element 0 and are masked) -> accumulators through LDS (the gathers are in
    // flight meanwhile) -> second batch issued -> first consumed -> second consumed.
    if (vec_ok) {
        f32x4 ta[NB][GB], tb[NB][GB];
        auto gather = [&](const int bt) {
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int j = bt * GB + u;
                const size_t oa = (((size_t)gn * L.Csrc + max(m1[j], 0)) * L.HWo + gp) & (size_t)(-(long long)(gin && m1[j] >= 0));
                const size_t ob = (((size_t)gn * L.Csrc + max(m2[j], 0)) * L.HWo + gp) & (size_t)(-(long long)(gin && m2[j] >= 0));
                if constexpr ((PLEAS_FWD_ABLATE & 1) != 0) {
                    ta[bt][u] = f32x4{(float)(oa & 3), 0.f, 0.f, 0.f};
                    tb[bt][u] = f32x4{(float)(ob & 3), 0.f, 0.f, 0.f};
                } else {
                    ta[bt][u] = *(const __attribute__((address_space(1))) f32x4*)(PLEAS_GLOBAL(L.o1) + oa);
                    tb[bt][u] = *(const __attribute__((address_space(1))) f32x4*)(PLEAS_GLOBAL(L.o2) + ob);
                }
            }
        };
        auto consume = [&](const int bt) {
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int j = bt * GB + u;
                const int lco = (tid >> 5) + 8 * j, co = i0 + lco;
                const bool live = gin && co < L.Cout;
                const float coef = co < L.n_merged ? 0.5f : 1.0f;
                const float fa = m1[j] >= 0 ? 1.f : 0.f, fb = m2[j] >= 0 ? 1.f : 0.f;
                const f32x4 o = *reinterpret_cast<const f32x4*>(Ct + lco * EL + pg);
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dd = (o[e] + bias_v[j]) - (ta[bt][u][e] * fa + tb[bt][u][e] * fb) * coef;
                    sq = live ? fmaf(dd, dd, sq) : sq;
                    d[e] = L.dscale * dd;
                }
                if (live) *(__attribute__((address_space(1))) f32x4*)(PLEAS_GLOBAL_W(L.resid) + ((size_t)gn * L.Cout + co) * L.HWo + gp) = d;
            }
        };
        gather(0);
        spill_acc();
        if constexpr (NB > 1) gather(1);
        consume(0);
        if constexpr (NB > 1) consume(1);
    } else {
        spill_acc();
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int lco = (tid >> 5) + 8 * j, co = i0 + lco;
            if (co >= L.Cout || !gin) continue;
            const float coef = co < L.n_merged ? 0.5f : 1.0f;
            const f32x4 o = *reinterpret_cast<const f32x4*>(Ct + lco * EL + pg);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t Pe = Pg + e;
                if (Pe >= L.Ptot) break;
                const uint32_t n = Pe / L.HWo, p = Pe - n * L.HWo;
                float a = 0.f, b = 0.f;
                if (m1[j] >= 0) a = PLEAS_GLOBAL(L.o1)[((size_t)n * L.Csrc + m1[j]) * L.HWo + p];
                if (m2[j] >= 0) b = PLEAS_GLOBAL(L.o2)[((size_t)n * L.Csrc + m2[j]) * L.HWo + p];
                const float dd = (o[e] + bias_v[j]) - (a + b) * coef;
                sq = fmaf(dd, dd, sq);
                PLEAS_GLOBAL_W(L.resid)[((size_t)n * L.Cout + co) * L.HWo + p] = L.dscale * dd;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
    __syncthreads();  // everyone is done with Ct: reuse its first floats for the block sum
    if (lane == 0) smem[wave] = sq;
    __syncthreads();
    if (tid == 0) partials[L.part_base + it.slot] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
#if (PLEAS_FWD_ABLATE & 16)
    if (tid == 0 && blockIdx.x < 32768) {
        const long long st3 = clock64();
        g_fwd_stamps[blockIdx.x][0] = st1 - st0;
        g_fwd_stamps[blockIdx.x][1] = st2 - st1;
        g_fwd_stamps[blockIdx.x][2] = st3 - st2;
        g_fwd_stamps[blockIdx.x][3] = nchunks * 1000 + TM;
    }
#endif
}

__global__ __launch_bounds__(fThreads, 2) void fwd_batch_kernel(const FwdLayerDev* __restrict__ layers,
                                                             const FwdItemDev* __restrict__ items,
                                                             float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdItemDev it = items[blockIdx.x];
    if (it.layer < 0) return;   // padding of the XCD-aware item order
    const FwdLayerDev L = layers[it.layer];
    switch (L.variant & 3) {
        case 0: fwd_tile<128, 4>(L, it, smem, partials); break;
        case 1: fwd_tile<64, 4>(L, it, smem, partials); break;
        case 2: fwd_tile<128, 1>(L, it, smem, partials); break;
        default: fwd_tile<64, 1>(L, it, smem, partials); break;
    }
}

// loss[l] = scale[l] * sum of this layer's partials (fixed order, fp64 combine)
struct FwdLossDev {
    int begin, count;
    float scale;
    int pad;
};
__global__ __launch_bounds__(64) void fwd_loss_kernel(const float* __restrict__ partials, const FwdLossDev* __restrict__ ld,
                                                      float* __restrict__ loss) {
    const FwdLossDev d = ld[blockIdx.x];
    double s = 0.0;
    for (int i = threadIdx.x; i < d.count; i += 64) s += (double)partials[d.begin + i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) loss[blockIdx.x] = (float)(s * (double)d.scale);
}

constexpr int fPtrBatch = 56;
struct FwdPtrBatch {
    int base, count;
    const float* ip[fPtrBatch];
    const float* w[fPtrBatch];
    const float* bias[fPtrBatch];
    const float* o1[fPtrBatch];
    const float* o2[fPtrBatch];
    const int32_t* row1[fPtrBatch];
    const int32_t* row2[fPtrBatch];
    float* resid[fPtrBatch];
};
__global__ void fwd_set_ptrs_kernel(FwdLayerDev* __restrict__ layers, const FwdPtrBatch b) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < b.count) {
        FwdLayerDev& L = layers[b.base + t];
        L.ip = b.ip[t];
        L.w = b.w[t];
        L.bias = b.bias[t];
        L.o1 = b.o1[t];
        L.o2 = b.o2[t];
        L.row1 = b.row1[t];
        L.row2 = b.row2[t];
        L.resid = b.resid[t];
    }
}

struct FwdPlan {
    std::vector<int64_t> key;
    std::vector<FwdLayerDev> layers;
    std::vector<FwdItemDev> items;
    std::vector<FwdLossDev> loss;
    size_t off_layers = 0, off_items = 0, off_loss = 0, off_parts = 0, total = 0, lds = 0;
    double flops = 0, bytes = 0;
    int n_parts = 0;
    bool uploaded = false;
};
static FwdPlan g_fplan;
static std::mutex g_fplan_mu;
static size_t falign(size_t v) { return (v + 255) / 256 * 256; }

static int build_fwd_plan(FwdPlan& P, const pleas_fwd_layer* ly, int n) {
    P.layers.assign(n, FwdLayerDev());
    P.items.clear();
    P.loss.assign(n, FwdLossDev());
    P.flops = P.bytes = 0;
    P.lds = 0;
    std::vector<XcdWork<FwdItemDev>> work;
    int parts = 0;
    for (int i = 0; i < n; ++i) {
        const pleas_fwd_layer& l = ly[i];
        if (l.N <= 0 || l.Cout <= 0 || l.Cin <= 0 || l.Hin <= 0 || l.Win <= 0 || l.KH <= 0 || l.KW <= 0 || l.stride <= 0 ||
            l.pad < 0 || l.Csrc <= 0 || l.n_merged < 0)
            return bad_arg("conv_fwd: layer geometry");
        const int Hout = (l.Hin + 2 * l.pad - l.KH) / l.stride + 1, Wout = (l.Win + 2 * l.pad - l.KW) / l.stride + 1;
        if (Hout <= 0 || Wout <= 0) return bad_arg("conv_fwd: empty output");
        if (l.KH * l.KW > 64) return bad_arg("conv_fwd: kernels larger than 64 taps are not supported");
        const int64_t HWo = (int64_t)Hout * Wout, Ptot = (int64_t)l.N * HWo, Kd = (int64_t)l.Cin * l.KH * l.KW;
        if (Ptot >= (1ll << 31) || (int64_t)l.Cout * Kd >= (1ll << 32)) return bad_arg("conv_fwd: tensor too large");
        FwdLayerDev& d = P.layers[i];
        d.Cout = l.Cout; d.Cin = l.Cin; d.Hin = l.Hin; d.Win = l.Win; d.Hout = Hout; d.Wout = Wout;
        d.KH = l.KH; d.KW = l.KW; d.stride = l.stride; d.pad = l.pad; d.Csrc = l.Csrc; d.n_merged = l.n_merged;
        d.HWo = (uint32_t)HWo; d.Ptot = (uint32_t)Ptot; d.Kd = (uint32_t)Kd;
        d.dscale = l.dscale;
        const int TM = l.Cout > 64 ? 128 : 64;
        d.variant = (TM == 64 ? 1 : 0) | (Kd % 4 == 0 ? 0 : 2);
        if (l.flags & PLEAS_FWD_KPOS_MAJOR) {
            if (l.Cin % fBK != 0) return bad_arg("conv_fwd: kernel-position-major weights need Cin % 32 == 0");
            d.variant |= 4;
        }
        d.part_base = parts;
        const int tms = (int)ceil_div(l.Cout, TM), tps = (int)ceil_div(Ptot, fTN);
        int slot = 0;
        // Within a layer the output-channel tile runs fastest (PLEAS_FWD_ORDER=3 restores pixel-tile-fastest): with
        // workgroup b on XCD b % 8, an XCD then keeps meeting the same few weight tiles, which stay in its L2, while
        // every input tile is streamed once per XCD that needs it.
        static const bool tp_major = !(std::getenv("PLEAS_FWD_ORDER") && std::atoi(std::getenv("PLEAS_FWD_ORDER")) == 3);
        for (int outer = 0; outer < (tp_major ? tps : tms); ++outer)
            for (int inner = 0; inner < (tp_major ? tms : tps); ++inner) {
                const int tm = tp_major ? inner : outer, tp = tp_major ? outer : inner;
                XcdWork<FwdItemDev> w;
                w.it = FwdItemDev{i, tm, tp, slot++};
                w.w = (double)ceil_div(Kd, fBK) * TM;
                // all items of a layer re-read its weights (and, across tm, its input): keep them on one XCD; layers
                // with many pixel tiles are cut into runs of 32 tiles so that the 8 queues still balance
                w.key = (int64_t)i * 65536 + tp / 32;
                work.push_back(w);
            }
        P.loss[i] = FwdLossDev{parts, slot, l.loss_scale, 0};
        parts += slot;
        P.lds = std::max(P.lds, (size_t)(2 * TM * fLdsA + 2 * fTN * fLdsB) * sizeof(float));  // >= TM*132 floats (epilogue)
        P.flops += 2.0 * l.Cout * (double)Kd * (double)Ptot;
        P.bytes += ((double)l.Cin * l.N * l.Hin * l.Win + 3.0 * l.Cout * (double)Ptot) * sizeof(float);
    }
    // Off by default for this kernel (PLEAS_XCD_ORDER=1 turns it on): it cuts FETCH_SIZE by 32 % (8.3 -> 6.1 GB per launch)
    // but costs 1-4 % of time -- co-resident workgroups of one layer reach their latency-bound epilogues together,
    // while the plain longest-first order mixes layers on a CU.
    P.items = xcd_order_items(work, FwdItemDev{-1, 0, 0, 0}, /*by_default=*/false);
    if (const char* env = std::getenv("PLEAS_FWD_ORDER")) {   // experiments: 1 = pseudo-random order, 2 = long / short interleaved
        const int mode = std::atoi(env);
        if (mode == 1) {
            uint64_t st = 0x9E3779B97F4A7C15ull;
            for (size_t i = P.items.size(); i > 1; --i) {
                st = st * 6364136223846793005ull + 1442695040888963407ull;
                std::swap(P.items[i - 1], P.items[(size_t)((st >> 33) % i)]);
            }
        } else if (mode == 2) {   // longest-first list folded: item k from the front, then item k from the back
            std::vector<FwdItemDev> folded;
            folded.reserve(P.items.size());
            size_t lo = 0, hi = P.items.size();
            while (lo < hi) {
                folded.push_back(P.items[lo++]);
                if (lo < hi) folded.push_back(P.items[--hi]);
            }
            P.items.swap(folded);
        }
    }
    P.n_parts = parts;
    size_t off = 0;
    P.off_layers = off;
    off = falign(off + P.layers.size() * sizeof(FwdLayerDev));
    P.off_items = off;
    off = falign(off + P.items.size() * sizeof(FwdItemDev));
    P.off_loss = off;
    off = falign(off + P.loss.size() * sizeof(FwdLossDev));
    P.off_parts = off;
    P.total = off + (size_t)parts * sizeof(float);
    P.uploaded = false;
    return PLEAS_OK;
}

}  // namespace pleas

using namespace pleas;

#if (PLEAS_FWD_ABLATE & 16)
extern "C" int pleas_fwd_debug_read(long long* out, int n_items) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_stamps), sizeof(long long) * 4 * (size_t)std::min(n_items, 32768)) == hipSuccess
               ? 0 : 1;
}
#endif

extern "C" size_t pleas_fwd_batch_ws_bytes(const pleas_fwd_layer* layers, int n_layers) {
    if (!layers || n_layers <= 0) return 0;
    FwdPlan tmp;
    if (build_fwd_plan(tmp, layers, n_layers) != PLEAS_OK) return 0;
    return tmp.total;
}

extern "C" int pleas_fwd_batch(const pleas_fwd_layer* layers, int n_layers, float* loss, void* ws, size_t ws_bytes,
                               int ws_fresh, void* stream_) {
    if (!layers || n_layers <= 0 || !loss) return bad_arg("conv_fwd: empty layer list");
    for (int i = 0; i < n_layers; ++i) {
        const pleas_fwd_layer& l = layers[i];
        if (!l.ip || !l.w || !l.o1 || !l.o2 || !l.row1 || !l.row2 || !l.resid) return bad_arg("conv_fwd: null pointer");
        if (((uintptr_t)l.w & 15) != 0) return bad_arg("conv_fwd: weights must be 16-byte aligned");
    }
    hipStream_t stream = (hipStream_t)stream_;
    std::lock_guard<std::mutex> lk(g_fplan_mu);
    FwdPlan& P = g_fplan;
    std::vector<int64_t> key;
    key.push_back(n_layers);
    key.push_back((int64_t)(uintptr_t)ws);
    for (int i = 0; i < n_layers; ++i) {
        const pleas_fwd_layer& l = layers[i];
        for (int v : {l.N, l.Cout, l.Cin, l.Hin, l.Win, l.KH, l.KW, l.stride, l.pad, l.Csrc, l.n_merged, l.flags})
            key.push_back(v);
        int32_t bits[2];
        std::memcpy(&bits[0], &l.dscale, 4);
        std::memcpy(&bits[1], &l.loss_scale, 4);
        key.push_back(bits[0]);
        key.push_back(bits[1]);
    }
    if (key != P.key) {
        const int rc = build_fwd_plan(P, layers, n_layers);
        if (rc != PLEAS_OK) return rc;
        P.key.swap(key);
    }
    if (ws_fresh) P.uploaded = false;
    if (!ws || ws_bytes < P.total) {
        std::snprintf(g_last_error, sizeof(g_last_error), "conv_fwd workspace too small: need %zu bytes", P.total);
        P.key.clear();
        return PLEAS_ENOMEM;
    }
    char* base = (char*)ws;
    if (!P.uploaded) {
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_layers, P.layers.data(), P.layers.size() * sizeof(FwdLayerDev),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_items, P.items.data(), P.items.size() * sizeof(FwdItemDev),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_loss, P.loss.data(), P.loss.size() * sizeof(FwdLossDev),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipStreamSynchronize(stream));
        P.uploaded = true;
    }
    FwdLayerDev* dl = reinterpret_cast<FwdLayerDev*>(base + P.off_layers);
    for (int b0 = 0; b0 < n_layers; b0 += fPtrBatch) {
        FwdPtrBatch pb;
        pb.base = b0;
        pb.count = std::min(fPtrBatch, n_layers - b0);
        for (int t = 0; t < pb.count; ++t) {
            const pleas_fwd_layer& l = layers[b0 + t];
            pb.ip[t] = l.ip; pb.w[t] = l.w; pb.bias[t] = l.bias; pb.o1[t] = l.o1; pb.o2[t] = l.o2;
            pb.row1[t] = l.row1; pb.row2[t] = l.row2; pb.resid[t] = l.resid;
        }
        hipLaunchKernelGGL(fwd_set_ptrs_kernel, dim3(1), dim3(64), 0, stream, dl, pb);
        PLEAS_LAUNCH_CHECK("fwd_set_ptrs_kernel");
    }
    float* parts = reinterpret_cast<float*>(base + P.off_parts);
    {
        ProfScope prof(kProfConvFwd, P.flops, P.bytes, stream);
        hipLaunchKernelGGL(fwd_batch_kernel, dim3((unsigned)P.items.size()), dim3(fThreads), P.lds, stream, dl,
                           reinterpret_cast<const FwdItemDev*>(base + P.off_items), parts);
    }
    PLEAS_LAUNCH_CHECK("fwd_batch_kernel");
    hipLaunchKernelGGL(fwd_loss_kernel, dim3(n_layers), dim3(64), 0, stream, parts,
                       reinterpret_cast<const FwdLossDev*>(base + P.off_loss), loss);
    PLEAS_LAUNCH_CHECK("fwd_loss_kernel");
    return PLEAS_OK;
}

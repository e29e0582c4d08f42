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
    int variant;          // bit0: TM == 64, bit1: scalar W loads, bit2: W is kernel-position-major [Cout][KH*KW][Cin],
                          // bit3: flat-shift tile (fwd_flat_tile), bits 4-5: its KIND
    int part_base;        // first loss-partial slot of this layer
    int bn_relu;          // PLAIN == 2: ReLU after the affine map (+ identity)
    // PLAIN == 2 (pleas_conv2d_bn_act_fwd): the convolution's output goes to `resid` AND its activated image
    // z = act(y * bn_scale[co] + bn_shift[co] (+ bn_res)) to `bn_z` -- the BatchNorm / add / ReLU pass that follows a frozen
    // source's convolution, without reading y back from memory
    const float* bn_scale;   // [Cout]
    const float* bn_shift;   // [Cout]
    const float* bn_res;     // [N][Cout][HWo] or null
    float* bn_z;             // [N][Cout][HWo]
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

// ---- shared by both tile forms: the block maps / biases of a tile's output channels, and the epilogue
template <int TM, int PLAIN = 0>
__device__ __forceinline__ void fwd_load_maps(const FwdLayerDev& L, const int i0, int (&m1)[TM / 8], int (&m2)[TM / 8],
                                              float (&bias_v)[TM / 8]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < TM / 8; ++j) {
        const int co = min(i0 + (tid >> 5) + 8 * j, L.Cout - 1);
        if constexpr (PLAIN == 2) {      // no block maps: their registers carry the channel's affine map (as bit patterns)
            m1[j] = __float_as_int(PLEAS_GLOBAL(L.bn_scale)[co]);
            m2[j] = __float_as_int(PLEAS_GLOBAL(L.bn_shift)[co]);
            bias_v[j] = L.bias ? PLEAS_GLOBAL(L.bias)[co] : 0.f;
            continue;
        }
        // a plain convolution (pleas_conv2d_fwd) has no block maps: every source row is absent, the target is zero
        m1[j] = L.row1 ? PLEAS_GLOBAL_I(L.row1)[co] : -1;
        m2[j] = L.row2 ? PLEAS_GLOBAL_I(L.row2)[co] : -1;
        bias_v[j] = L.bias ? PLEAS_GLOBAL(L.bias)[co] : 0.f;
    }
}

// PLAIN = 1 (pleas_conv2d_fwd): no target -- nothing is gathered, no loss partial; the epilogue is bias + store
// PLAIN = 2 (pleas_conv2d_bn_act_fwd): the same, and the activated image  act(fma(y, scale, shift) + identity)  is stored
// beside y -- the arithmetic of bn_act_kernel (elementwise.hip) on the value that is still in registers; the identity takes
// the gather slots of the target
template <int TM, int PLAIN = 0>
__device__ __forceinline__ void fwd_epilogue(const FwdLayerDev& L, const FwdItemDev& it, f32x16 (&acc)[TM / 64][2],
                                             const int (&m1)[TM / 8], const int (&m2)[TM / 8],
                                             const float (&bias_v)[TM / 8], float* smem, float* __restrict__ partials) {
    constexpr int MTM = TM / 64;
    constexpr int ROWS = TM / 8, GB = 8, NB = ROWS / GB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = it.tm * TM;
    const uint32_t p0 = (uint32_t)it.tp * fTN;
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
        // per-thread bases: the (sample, pixel group) part of every address is the same for all of the thread's channels, so a
        // gather / store address is the tensor's uniform base pointer + ONE 32-bit multiply-add in bytes (the plan builder
        // refuses tensors of 4 GB and more)
        const uint32_t hw4 = 4u * L.HWo;
        const uint32_t gbase = 4u * (gn * (uint32_t)L.Csrc * L.HWo + gp);
        const uint32_t rbase = 4u * ((gn * (uint32_t)L.Cout + (uint32_t)i0 + (uint32_t)(tid >> 5)) * L.HWo + gp);
        // PLAIN == 2: the identity (shaped like the output) stands where the first source's outputs are gathered from
        const bool has_res = PLAIN == 2 && L.bn_res != nullptr;
        const char* o1b = reinterpret_cast<const char*>(PLAIN == 2 ? (has_res ? L.bn_res : L.ip) : L.o1);
        const char* o2b = reinterpret_cast<const char*>(L.o2);
        char* rb_ = reinterpret_cast<char*>(L.resid);
        char* zb_ = reinterpret_cast<char*>(L.bn_z);
        auto gather = [&](const int bt) {
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int j = bt * GB + u;
                const uint32_t oa = (gin && m1[j] >= 0) ? gbase + (uint32_t)m1[j] * hw4 : 0u;
                const uint32_t ob = (gin && m2[j] >= 0) ? gbase + (uint32_t)m2[j] * hw4 : 0u;
                if constexpr (PLAIN == 2) {
                    // unconditional load from a valid address (the output's own offset into the identity; the first floats
                    // of the input for rows / pixels outside the layer and when there is no identity -- then voided by a
                    // select): no branch around a load
                    const bool live = has_res && gin && i0 + (tid >> 5) + 8 * j < L.Cout;
                    const uint32_t oz = live ? rbase + (uint32_t)(8 * j) * hw4 : 0u;
                    const f32x4 idn = *(const __attribute__((address_space(1))) f32x4*)(o1b + oz);
                    ta[bt][u] = has_res ? idn : f32x4{0.f, 0.f, 0.f, 0.f};
                } else if constexpr (PLAIN) {
                    ta[bt][u] = tb[bt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else if constexpr ((PLEAS_FWD_ABLATE & 1) != 0) {
                    ta[bt][u] = f32x4{(float)(oa & 3), 0.f, 0.f, 0.f};
                    tb[bt][u] = f32x4{(float)(ob & 3), 0.f, 0.f, 0.f};
                } else {
                    ta[bt][u] = *(const __attribute__((address_space(1))) f32x4*)(o1b + oa);
                    tb[bt][u] = *(const __attribute__((address_space(1))) f32x4*)(o2b + ob);
                }
            }
        };
        auto consume = [&](const int bt) {
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int j = bt * GB + u;
                const int lco = (tid >> 5) + 8 * j, co = i0 + lco;
                const bool live = gin && co < L.Cout;
                if constexpr (PLAIN == 2) {
                    const float a = __int_as_float(m1[j]), b = __int_as_float(m2[j]);
                    const f32x4 o = *reinterpret_cast<const f32x4*>(Ct + lco * EL + pg);
                    f32x4 y, z;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        y[e] = o[e] + bias_v[j];
                        const float sm = fmaf(y[e], a, b) + ta[bt][u][e];
                        z[e] = L.bn_relu ? fmaxf(sm, 0.f) : sm;
                    }
                    if (live) {
                        *(__attribute__((address_space(1))) f32x4*)(rb_ + (rbase + (uint32_t)(8 * j) * hw4)) = y;
                        *(__attribute__((address_space(1))) f32x4*)(zb_ + (rbase + (uint32_t)(8 * j) * hw4)) = z;
                    }
                    continue;
                }
                // target = (o1 * [present] + o2 * [present]) * coef, coef in {0.5, 1}: folding coef into the two factors is
                // exact (power of two), so  fma(o2, cb, o1 * ca)  rounds once, like the sum it replaces
                const float coef = co < L.n_merged ? 0.5f : 1.0f;
                const float ca = m1[j] >= 0 ? coef : 0.f, cb = m2[j] >= 0 ? coef : 0.f;
                const f32x4 o = *reinterpret_cast<const f32x4*>(Ct + lco * EL + pg);
                f32x4 d;
                float s4 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dd = (o[e] + bias_v[j]) - fmaf(tb[bt][u][e], cb, ta[bt][u][e] * ca);
                    s4 = fmaf(dd, dd, s4);
                    d[e] = L.dscale * dd;
                }
                sq += live ? s4 : 0.f;
                if (live) *(__attribute__((address_space(1))) f32x4*)(rb_ + (rbase + (uint32_t)(8 * j) * hw4)) = d;
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
                if constexpr (PLAIN == 2) {
                    const size_t at = ((size_t)n * L.Cout + co) * L.HWo + p;
                    const float y = o[e] + bias_v[j];
                    const float sm = fmaf(y, __int_as_float(m1[j]), __int_as_float(m2[j])) + (L.bn_res ? PLEAS_GLOBAL(L.bn_res)[at] : 0.f);
                    PLEAS_GLOBAL_W(L.resid)[at] = y;
                    PLEAS_GLOBAL_W(L.bn_z)[at] = L.bn_relu ? fmaxf(sm, 0.f) : sm;
                    continue;
                }
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
    if (tid == 0 && partials) partials[L.part_base + it.slot] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
}

template <int TM, int VECA, int PLAIN = 0>
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
    constexpr int ROWS = TM / 8;
    int m1[ROWS], m2[ROWS];
    float bias_v[ROWS];
    auto load_maps = [&]() { fwd_load_maps<TM, PLAIN>(L, i0, m1, m2, bias_v); };
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

    fwd_epilogue<TM, PLAIN>(L, it, acc, m1, m2, bias_v, smem, partials);
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


// ------------------------------------------------------------------------------------------------------------------
// "Flat-shift" tile: stride-1, same-size convolutions (1x1, 3x3 pad 1, 5x5 pad 2 ...) with Cin % 32 == 0 -- 97 % of a
// ResNet's merged-layer flops.  For a fixed channel the HW pixels of a sample are contiguous, and tap (kh, kw) of a
// stride-1 same-size convolution reads the SAME flat pixel run shifted by delta = (kh - pad) * W + (kw - pad).  So the
// input operand of a whole channel block (32 channels x all KH*KW taps) is ONE LDS image
//     Bs[k][j],  j = 0 .. 128 + 2 * halo,   halo = pad * (W + 1),   holding flat pixels p0 - halo .. p0 + 127 + halo,
// loaded once per channel block (16-B loads along the pixel axis for 1x1 layers, 9x fewer loads than one gather per tap
// for 3x3), and the MFMA B fragment of tap r is a plain ds_read_b32 at  Bs[k][halo + pixel + delta_r].  Border taps
// (left / right column, top / bottom row, neighbouring sample) are not masked value by value: a lane whose pixel does not
// have tap r reads the row's ZERO COLUMN instead (one select on the address per tap and 32-pixel fragment).
// Weights: kernel-position-major chunks (tap r, channel block) exactly as in fwd_tile; accumulators and epilogue shared.
constexpr int fFlatRow1 = 132;     // Bs row stride of 1x1 layers: 128 pixels + zero column, 16-B aligned rows
constexpr int fFlatRowK = 260;     // (round 2-3 layout of the k x k image, [k][column]; kept for the 1x1 forms' sizing only)
constexpr int fFlatPix = 36;       // k x k image since round 4: [column][36] -- a pixel's 32 channels contiguous (+ 4 pad: 16-byte
                                   // aligned rows, conflict-free ds_read_b128 as for the weight tile), so that ONE 16-byte LDS read
                                   // feeds four MFMA steps instead of four 4-byte reads (an LDS read costs the SIMD ~10 cycles
                                   // whatever its width)
constexpr int fFlatColsK = 248;    // columns of that image: 128 + 2 * halo data columns + one zero row
#ifndef PLEAS_FWD_TIMELINE
#define PLEAS_FWD_TIMELINE 0   // study builds (tools/r04/timeline.sh): per work item, when and where it ran
#endif
#if PLEAS_FWD_TIMELINE
__shared__ long long g_tl_phase[2];   // wall clock at the end of the prologue / of the K loop (flat forms)
#define PLEAS_TL_PHASE(k) do { if (threadIdx.x == 0) g_tl_phase[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PLEAS_TL_PHASE(k)
#endif
// KIND 0: 1x1, 16-B loads along the pixel axis (HW % 4 == 0), two images (double buffer)
// KIND 1: 1x1, scalar loads (HW % 4 != 0, e.g. 7 x 7), two images
// KIND 2: k x k, scalar loads, ONE image per channel block shared by all taps
// SPLIT = 1 (pleas_arith(PLEAS_ARITH_SPLIT_BF16); KIND 0 and 2): every chunk goes to LDS as three bf16 planes (common.hpp), ONE
// image per operand (two barriers per chunk), six v_mfma_f32_32x32x16_bf16 per 16-deep k step.
//   weights:      [TM][kSplitRow] rows (3 planes x 32 k), fragments by ds_read_b128 -- the operand map of the MFMA
//   k x k input:  [column][kSplitRow], a pixel's 32 channels contiguous per plane: the same 16-byte fragment reads
//   1 x 1 input:  [plane][k][fSplitPixRow] (pixels contiguous, as they come from memory: 8-byte writes), fragments by
//                 ds_read_b64_tr_b16 -- the hardware transposes 4 k x 16 pixels per 16-lane group
constexpr int fSplitPixRow = 160;  // bf16 per k row of the 1 x 1 split image: 128 pixels + 32 pad = 320 B (rows 16 banks apart:
                                   // the four rows x eight 8-byte column chunks of a half-wave's transposed read hit 32 distinct bank pairs)
typedef short s16x4_t __attribute__((ext_vector_type(4)));
template <int TM, int KIND, int SPLIT = 0, int PLAIN = 0>
__device__ __forceinline__ void fwd_flat_tile(const FwdLayerDev& L, const FwdItemDev& it, float* smem, float* __restrict__ partials) {
    static_assert(!SPLIT || KIND != 1, "the scalar 1 x 1 form (7 x 7 images) has no split variant");
    constexpr int MTM = TM / 64;
    constexpr int LPR = fBK / 4, RPP = fThreads / LPR, PASS = TM / RPP;   // weight staging: 16-B loads
    constexpr int Lr = KIND == 2 ? fFlatPix : fFlatRow1;                // compile-time: LDS offsets are immediates
    constexpr int jz = KIND == 2 ? fFlatColsK - 1 : Lr - 1;              // the zero column (k x k: the zero ROW of the image)
    constexpr int NBUF = KIND == 2 ? 1 : 2;
    constexpr int MCOL = KIND == 2 ? 4 : 2;                              // scalar form: column passes of 64 lanes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = it.tm * TM;
    const uint32_t p0 = (uint32_t)it.tp * fTN;
    const int R = KIND == 2 ? L.KH * L.KW : 1;
    const int W = L.Win;
    const uint32_t HW = L.HWo;                       // == Hin * Win
    const int halo = KIND == 2 ? L.pad * (W + 1) : 0;
    const int span = fTN + 2 * halo;                 // data columns of a Bs row
    const int CB = L.Cin / fBK;
    const int nchunks = CB * R;
    float* As = smem;                                // [2][TM][fLdsA]
    float* Bs = smem + 2 * TM * fLdsA;               // 1x1: [NBUF][32][Lr]; k x k: [fFlatColsK][fFlatPix]
    __bf16* As16 = reinterpret_cast<__bf16*>(smem);  // SPLIT: [TM][kSplitRow]
    __bf16* Bs16 = As16 + TM * kSplitRow;            // SPLIT: k x k [fFlatColsK][kSplitRow]; 1 x 1 [3][32][fSplitPixRow]

    // ---- A (weights) staging, as in fwd_tile with VECA = 4
    const int arow = SPLIT ? split_stage_row(tid / LPR) : tid / LPR, acol = (tid % LPR) * 4;      // SPLIT: conflict-free plane writes
    f32x4 ra0[PASS], ra1[PASS];    // two register sets: the weights of chunk c + 2 are requested while chunk c computes
    unsigned oka = 0;
    uint32_t offa[PASS];
#pragma unroll
    for (int q = 0; q < PASS; ++q) {
        const int gi = i0 + arow + q * RPP;
        if (gi < L.Cout) oka |= 1u << q;
        offa[q] = (uint32_t)min(gi, L.Cout - 1) * L.Kd;
    }
    // ---- B (input) staging.
    //   KIND 0: thread = (row tid / 32 + 8 i, pixel group tid % 32), i < 4: four 16-B loads per chunk
    //   else  : wave w owns rows 8 w .. 8 w + 7, lane covers columns lane + 64 m (columns >= span unused)
    uint32_t voff[MCOL];
    unsigned vok = 0;
    constexpr int NRB = KIND == 0 ? 16 : 8 * MCOL;
    float rb0[NRB], rb1[KIND == 2 ? 1 : NRB];   // 1x1 forms: two sets as well (a new image every chunk)
    if constexpr (KIND == 0) {
        const uint32_t P4 = p0 + 4u * (tid & 31);
        const bool ok = P4 < L.Ptot;
        const uint32_t n = ok ? P4 / HW : 0u, p = ok ? P4 - n * HW : 0u;
        voff[0] = ok ? n * (uint32_t)L.Cin * HW + p + (uint32_t)(tid >> 5) * HW : 0u;
        voff[1] = 0;
        vok = ok ? 1u : 0u;
    } else {
#pragma unroll
        for (int m = 0; m < MCOL; ++m) {
            const int j = lane + 64 * m;
            const long long Pv = (long long)p0 - halo + j;
            const bool ok = j < span && Pv >= 0 && Pv < (long long)L.Ptot;
            const uint32_t n = ok ? (uint32_t)Pv / HW : 0u, p = ok ? (uint32_t)Pv - n * HW : 0u;
            voff[m] = ok ? n * (uint32_t)L.Cin * HW + p : 0u;
            vok |= (ok ? 1u : 0u) << m;
        }
    }
    const int M = (span + 63) / 64;                    // scalar form: column passes that hold data (wave-uniform)

    // ---- MFMA-side view of this lane's two pixels (fragments sn = 0, 1): LDS column and tap validity
    int jb[2];
    unsigned tapok[2];
#pragma unroll
    for (int sn = 0; sn < 2; ++sn) {
        const int q = wn * 64 + sn * 32 + (lane & 31);
        const uint32_t P = p0 + q;
        jb[sn] = halo + q;
        unsigned mask = 0;
        if (P < L.Ptot) {
            if constexpr (KIND == 2) {
                const uint32_t n = P / HW, pp = P - n * HW;
                const int oh = (int)(pp / (uint32_t)W), ow = (int)(pp - (uint32_t)oh * W);
                for (int r = 0; r < R; ++r) {
                    const int kh = r / L.KW, kw = r - kh * L.KW;
                    const int ih = oh + kh - L.pad, iw = ow + kw - L.pad;
                    mask |= (unsigned)(ih >= 0 && ih < L.Hin && iw >= 0 && iw < W) << r;
                }
            } else {
                mask = 1u;
            }
        }
        tapok[sn] = mask;
    }

    f32x16 acc[MTM][2];
#pragma unroll
    for (int a = 0; a < MTM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto load_a = [&](int cb, int r, f32x4 (&ra)[PASS]) {
        const uint32_t k = (uint32_t)r * L.Cin + (uint32_t)cb * fBK + acol;    // kernel-position-major (== plain for 1x1)
#pragma unroll
        for (int q = 0; q < PASS; ++q) {
            if constexpr ((PLEAS_FWD_ABLATE & 8) != 0) ra[q] = f32x4{(float)(k & 3), 0.f, 1.f, 0.f}; else
            ra[q] = *(const __attribute__((address_space(1))) f32x4*)(PLEAS_GLOBAL(L.w) + offa[q] + k);
        }
    };
    // Tiles whose rows / pixels all exist store their staged values as they are (block-uniform tests): every select that is
    // not executed is vector-pipe time the fp32 MFMAs get back (a vector instruction costs ~4.5 cycles of it).
    const bool full_a = i0 + TM <= L.Cout;
    const bool full_b = p0 + fTN <= L.Ptot;
    auto store_a = [&](int buf, const f32x4 (&ra)[PASS]) {
        if constexpr (SPLIT) {
#pragma unroll
            for (int q = 0; q < PASS; ++q) {
                const bool ok = full_a || ((oka >> q) & 1u);
                split3_store4(As16 + (arow + q * RPP) * kSplitRow, acol, ok ? ra[q][0] : 0.f, ok ? ra[q][1] : 0.f,
                              ok ? ra[q][2] : 0.f, ok ? ra[q][3] : 0.f);
            }
            return;
        }
        float* a = As + buf * TM * fLdsA;
        if (full_a) {
#pragma unroll
            for (int q = 0; q < PASS; ++q) *reinterpret_cast<f32x4*>(a + (arow + q * RPP) * fLdsA + acol) = ra[q];
            return;
        }
#pragma unroll
        for (int q = 0; q < PASS; ++q) {
            const bool ok = (oka >> q) & 1u;
            const f32x4 v = {ok ? ra[q][0] : 0.f, ok ? ra[q][1] : 0.f, ok ? ra[q][2] : 0.f, ok ? ra[q][3] : 0.f};
            *reinterpret_cast<f32x4*>(a + (arow + q * RPP) * fLdsA + acol) = v;
        }
    };
    auto load_b = [&](int cb, auto& rb) {
        const cgfloat* base = PLEAS_GLOBAL(L.ip) + (size_t)cb * fBK * HW;
        if constexpr (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v;
                if constexpr ((PLEAS_FWD_ABLATE & 4) != 0) v = f32x4{(float)(cb & 3), 0.f, 1.f, 0.f}; else
                v = *(const __attribute__((address_space(1))) f32x4*)(base + voff[0] + (uint32_t)(8 * i) * HW);
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[4 * i + e] = v[e];
            }
        } else {
            const cgfloat* wbase = base + (size_t)(8 * wave) * HW;     // wave-uniform row block
#pragma unroll
            for (int kr = 0; kr < 8; ++kr)
#pragma unroll
                for (int m = 0; m < MCOL; ++m)
                    if (m < M) {
                        if constexpr ((PLEAS_FWD_ABLATE & 4) != 0) rb[kr * MCOL + m] = (float)((cb + kr) & 3); else
                        rb[kr * MCOL + m] = wbase[(size_t)kr * HW + voff[m]];
                    }
        }
    };
    auto store_b = [&](int buf, const auto& rb) {
        if constexpr (SPLIT && KIND == 0) {
            // this thread holds k rows tid / 32 + 8 i of pixels 4 (tid % 32) .. + 3: per row and plane one 8-byte write
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t lo[3], hi[3];
                split3_pair(vok ? rb[4 * i] : 0.f, vok ? rb[4 * i + 1] : 0.f, lo);
                split3_pair(vok ? rb[4 * i + 2] : 0.f, vok ? rb[4 * i + 3] : 0.f, hi);
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<u32x2_t*>(Bs16 + (p * fBK + (tid >> 5) + 8 * i) * fSplitPixRow + 4 * (tid & 31)) = u32x2_t{lo[p], hi[p]};
            }
            return;
        } else if constexpr (SPLIT && KIND == 2) {
            // channels 8 wave .. + 7 of columns lane + 64 m: per column and plane ONE 16-byte write
#pragma unroll
            for (int m = 0; m < MCOL; ++m)
                if (m < M && lane + 64 * m < span) {
                    const bool ok = (vok >> m) & 1u;
                    uint32_t w[4][3];
#pragma unroll
                    for (int h = 0; h < 4; ++h)
                        split3_pair(ok ? rb[(2 * h) * MCOL + m] : 0.f, ok ? rb[(2 * h + 1) * MCOL + m] : 0.f, w[h]);
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        *reinterpret_cast<u32x4_t*>(Bs16 + (lane + 64 * m) * kSplitRow + p * 32 + 8 * wave) =
                            u32x4_t{w[0][p], w[1][p], w[2][p], w[3][p]};
                }
            return;
        }
        float* b = Bs + buf * fBK * Lr;
        if constexpr (KIND == 0) {
            if (full_b) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = {rb[4 * i], rb[4 * i + 1], rb[4 * i + 2], rb[4 * i + 3]};
                    *reinterpret_cast<f32x4*>(b + ((tid >> 5) + 8 * i) * Lr + 4 * (tid & 31)) = v;
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = {vok ? rb[4 * i] : 0.f, vok ? rb[4 * i + 1] : 0.f, vok ? rb[4 * i + 2] : 0.f, vok ? rb[4 * i + 3] : 0.f};
                *reinterpret_cast<f32x4*>(b + ((tid >> 5) + 8 * i) * Lr + 4 * (tid & 31)) = v;
            }
        } else if constexpr (KIND == 2) {
            // this thread holds channels 8 wave .. + 7 of columns lane + 64 m: two 16-byte runs of the column's row
#pragma unroll
            for (int m = 0; m < MCOL; ++m)
                if (m < M && lane + 64 * m < span) {
                    const bool ok = (vok >> m) & 1u;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 v = {ok ? rb[(4 * h + 0) * MCOL + m] : 0.f, ok ? rb[(4 * h + 1) * MCOL + m] : 0.f,
                                         ok ? rb[(4 * h + 2) * MCOL + m] : 0.f, ok ? rb[(4 * h + 3) * MCOL + m] : 0.f};
                        *reinterpret_cast<f32x4*>(b + (lane + 64 * m) * fFlatPix + 8 * wave + 4 * h) = v;
                    }
                }
        } else {
#pragma unroll
            for (int kr = 0; kr < 8; ++kr)
#pragma unroll
                for (int m = 0; m < MCOL; ++m)
                    if (m < M && lane + 64 * m < span)
                        b[(8 * wave + kr) * Lr + lane + 64 * m] = ((vok >> m) & 1u) ? rb[kr * MCOL + m] : 0.f;
        }
    };
    // tap offsets of a k x k layer without a division per chunk: tap r -> r + 1 is one pixel to the right, or (at the end of a
    // kernel row) W - KW + 1 pixels on; taps are visited in order 0 .. R - 1 per channel block (all scalar-unit arithmetic)
    const int delta0 = -(L.pad * W + L.pad);
    auto tap_delta = [&](int r) {
        if constexpr (KIND != 2) return 0;
        int kh = 0, rr = r;
        while (rr >= L.KW) { rr -= L.KW; ++kh; }      // r < 32, KW >= 3: at most a few scalar steps
        return delta0 + kh * W + rr;
    };
    auto compute = [&](int abuf, int bbuf, int r) {
        const int delta = tap_delta(r);
        const float* a = As + abuf * TM * fLdsA + (wm * (TM / 2) + (lane & 31)) * fLdsA + 4 * (lane >> 5);
        const float* bb = Bs + bbuf * fBK * Lr + 4 * (lane >> 5) * Lr;     // half-wave h takes k = 8 kk + e + 4 h
        // A lane whose tap falls outside the image takes 0: by a select on the value it read at its own (always valid:
        // the halo is part of the image) address.  Reading the row's zero column instead put that lane on the bank of
        // some other lane of its half-wave: 22 % of the k x k forms' LDS cycles were such 2-way conflicts
        // (profiles/r03_lds_fwd_batch_rn101_before.txt).  PLEAS_FWD_ZEROCOL=1 at build time restores the zero-column read.
#ifndef PLEAS_FWD_ZEROCOL
#define PLEAS_FWD_ZEROCOL 0
#endif
        const bool ok0 = (tapok[0] >> r) & 1u, ok1 = (tapok[1] >> r) & 1u;
        if constexpr (SPLIT) {
            const __bf16* a16 = As16 + (wm * (TM / 2) + (lane & 31)) * kSplitRow + 8 * (lane >> 5);
#pragma unroll
            for (int g16 = 0; g16 < fBK / 16; ++g16) {
                bf16x8_t sa[MTM][3], sb[2][3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int s_ = 0; s_ < MTM; ++s_) sa[s_][p] = *reinterpret_cast<const bf16x8_t*>(a16 + s_ * 32 * kSplitRow + p * 32 + g16 * 16);
                if constexpr (KIND == 2) {
                    // a lane whose tap is off the image reads the zero row (one select on the address per tap and fragment)
                    const __bf16* c0 = Bs16 + (ok0 ? jb[0] + delta : jz) * kSplitRow + 8 * (lane >> 5) + g16 * 16;
                    const __bf16* c1 = Bs16 + (ok1 ? jb[1] + delta : jz) * kSplitRow + 8 * (lane >> 5) + g16 * 16;
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        sb[0][p] = *reinterpret_cast<const bf16x8_t*>(c0 + p * 32);
                        sb[1][p] = *reinterpret_cast<const bf16x8_t*>(c1 + p * 32);
                    }
                } else {
                    // transposed reads: lane 4 q + pp of a 16-lane group supplies row q, columns 4 pp .. + 3 of the group's
                    // 4 k x 16 pixel block and receives its own pixel's four k; two reads give the lane its 8 k
                    const int u = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
                    const int krow = g16 * 16 + 8 * (lane >> 5) + q;
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            const __bf16* base = Bs16 + (p * fBK + krow) * fSplitPixRow + wn * 64 + sn * 32 + 16 * u + 4 * pp;
                            const s16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base));
                            const s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + 4 * fSplitPixRow));
                            typedef short s16x8_t __attribute__((ext_vector_type(8)));
                            const s16x8_t both = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                            sb[sn][p] = __builtin_bit_cast(bf16x8_t, both);
                        }
                }
#pragma unroll
                for (int sm = 0; sm < MTM; ++sm) {
                    acc[sm][0] = split3_mfma(sa[sm], sb[0], acc[sm][0]);
                    acc[sm][1] = split3_mfma(sa[sm], sb[1], acc[sm][1]);
                }
            }
            return;
        }
        if constexpr (KIND == 2) {
            // [column][k] image: a lane's four k of an MFMA group are ONE 16-byte read; a lane whose tap is off the image
            // reads the zero row (one select on the address per tap and fragment instead of one per value)
            const float* c0 = Bs + (ok0 ? jb[0] + delta : jz) * fFlatPix + 4 * (lane >> 5);
            const float* c1 = Bs + (ok1 ? jb[1] + delta : jz) * fFlatPix + 4 * (lane >> 5);
#pragma unroll
            for (int kk = 0; kk < fBK / 8; ++kk) {
                f32x4 fa[MTM];
#pragma unroll
                for (int s = 0; s < MTM; ++s) fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * fLdsA + kk * 8);
                const f32x4 f0 = *reinterpret_cast<const f32x4*>(c0 + kk * 8), f1 = *reinterpret_cast<const f32x4*>(c1 + kk * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int sm = 0; sm < MTM; ++sm) {
                        acc[sm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], f0[e], acc[sm][0], 0, 0, 0);
                        acc[sm][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], f1[e], acc[sm][1], 0, 0, 0);
                    }
            }
            return;
        }
        constexpr bool zero_col = PLEAS_FWD_ZEROCOL || KIND != 2;      // 1 x 1 forms: only pixels past the tensor's end
        const float* b0 = bb + ((zero_col && !ok0) ? jz : jb[0] + delta);
        const float* b1 = bb + ((zero_col && !ok1) ? jz : jb[1] + delta);
#pragma unroll
        for (int kk = 0; kk < fBK / 8; ++kk) {
            f32x4 fa[MTM];
#pragma unroll
            for (int s = 0; s < MTM; ++s) fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * fLdsA + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float f0 = b0[(kk * 8 + e) * Lr], f1 = b1[(kk * 8 + e) * Lr];
                if constexpr (!zero_col) {
                    f0 = ok0 ? f0 : 0.f;
                    f1 = ok1 ? f1 : 0.f;
                }
#pragma unroll
                for (int sm = 0; sm < MTM; ++sm) {
                    acc[sm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], f0, acc[sm][0], 0, 0, 0);
                    acc[sm][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], f1, acc[sm][1], 0, 0, 0);
                }
            }
        }
    };

    // zero column of every row of every image (never overwritten: data columns end at span - 1 < jz); k x k: the zero row
    if constexpr (SPLIT) {
        if constexpr (KIND == 2) {       // the zero row of the split image: 3 planes x 32 k
            if (tid < 12) *reinterpret_cast<u32x4_t*>(Bs16 + jz * kSplitRow + 8 * tid) = u32x4_t{0u, 0u, 0u, 0u};
        }                                // 1 x 1: pixels past the tensor's end are stored as zeros (no zero column)
    } else if constexpr (KIND == 2) {
        if (tid < fFlatPix / 4) *reinterpret_cast<f32x4*>(Bs + jz * fFlatPix + 4 * tid) = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if (tid < fBK * NBUF) Bs[tid * Lr + jz] = 0.f;
    load_a(0, 0, ra0);
    load_b(0, rb0);
    store_a(0, ra0);
    store_b(0, rb0);
    __syncthreads();
    PLEAS_TL_PHASE(0);
    constexpr int ROWS = TM / 8;
    int m1[ROWS], m2[ROWS];
    float bias_v[ROWS];
    // Software pipeline, two chunks deep for the global loads: while chunk c computes, the operands of chunk c + 1 are in
    // flight or in registers (requested one iteration earlier) and those of chunk c + 2 are requested.  k x k forms load
    // the input image of the next channel block two taps before its first use.
    int bsel = 0;
    int r1 = R > 1 ? 1 : 0, cb1 = R > 1 ? 0 : 1;       // (channel block, tap) of chunk c + 1
    if (nchunks > 1) {
        load_a(cb1, r1, ra1);
        if constexpr (KIND != 2) load_b(cb1, rb1);
    }
    auto step = [&](int c, int r, f32x4 (&ra_next)[PASS], f32x4 (&ra_far)[PASS], auto& rb_next, auto& rb_far) {
        // chunk c computes; chunk c + 1 (operands in *_next) is stored afterwards; chunk c + 2 is requested into *_far
        int r2 = r1 + 1, cb2 = cb1;
        if (r2 == R) { r2 = 0; ++cb2; }
        const bool far = c + 2 < nchunks;
        const bool newb = r1 == 0;                   // chunk c + 1 starts a new channel block
        if (far) load_a(cb2, r2, ra_far);
        if constexpr (KIND != 2) {
            if (far) load_b(cb2, rb_far);
        } else {
            // the image of block cb1 + 1 (first used by chunk c + 1 + (R - r1)) is requested when chunk c + 1 is its
            // block's second-to-last tap, i.e. two chunks before the store below needs it
            if (r1 == R - 2 && cb1 + 1 < CB) load_b(cb1 + 1, rb0);     // ONE register set: an image per R chunks
        }
        compute(c & 1, bsel, r);
        if constexpr (SPLIT) {
            __syncthreads();                         // every wave is done reading the (single) images
            store_a(0, ra_next);
            if (newb) {
                if constexpr (KIND == 2) store_b(0, rb0);
                else store_b(0, rb_next);
            }
            __syncthreads();
            r1 = r2;
            cb1 = cb2;
            return;
        }
        store_a((c + 1) & 1, ra_next);
        if (newb) {
            if constexpr (NBUF == 2) {
                bsel ^= 1;
                store_b(bsel, rb_next);
            } else {
                __syncthreads();                     // every wave is done reading the single image
                store_b(0, rb0);
            }
        }
        __syncthreads();
        r1 = r2;
        cb1 = cb2;
    };
    int r = 0;
    int c = 0;
    for (; c + 2 < nchunks; c += 2) {
        const int ra_ = r1;                          // tap of chunk c + 1, read before step() advances it
        step(c, r, ra1, ra0, rb1, rb0);
        const int rb_ = r1;
        step(c + 1, ra_, ra0, ra1, rb0, rb1);
        r = rb_;
    }
    if (c + 1 < nchunks) {
        const int ra_ = r1;
        step(c, r, ra1, ra0, rb1, rb0);
        r = ra_;
    }
    fwd_load_maps<TM, PLAIN>(L, i0, m1, m2, bias_v);        // epilogue operands, requested under the last chunk's MFMAs
    compute((nchunks - 1) & 1, bsel, r);
    __syncthreads();
    PLEAS_TL_PHASE(1);
    fwd_epilogue<TM, PLAIN>(L, it, acc, m1, m2, bias_v, smem, partials);
}

// One kernel per tile form (register allocation and LDS are then per form, not the maximum over all of them); the host
// launches each form's slice of the item list.  Form ids: 0-3 = fwd_tile<128,4>, <64,4>, <128,1>, <64,1>;
// 4-6 = fwd_flat_tile<128, KIND 0-2>;  7-9 = fwd_flat_tile<64, KIND 0-2>.
constexpr int fForms = 10;
#if PLEAS_FWD_TIMELINE
__device__ long long g_fwd_timeline[32768][4];   // start, end (100 MHz wall clock), flat forms: prologue | K loop << 32 [ticks], form | chunks * TM << 8
__device__ int g_fwd_timeline_n;
#endif
// SPLIT = 1: the split-bf16 variant of a flat form (4, 6, 7, 9: KIND 0 / 2), launched instead of the exact one when the plan
// was built under pleas_arith(PLEAS_ARITH_SPLIT_BF16); every other form runs the exact arithmetic in the same launch group.
constexpr bool fwd_form_splits(int form) { return form == 4 || form == 6 || form == 7 || form == 9; }
template <int FORM, int SPLIT = 0>
__global__ __launch_bounds__(fThreads, (SPLIT && FORM >= 7) ? 3 : 2) void fwd_batch_kernel(const FwdLayerDev* __restrict__ layers,
                                                             const FwdItemDev* __restrict__ items,
                                                             float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdItemDev it = items[blockIdx.x];
    if (it.layer < 0) return;   // padding of the XCD-aware item order
    const FwdLayerDev L = layers[it.layer];
#if PLEAS_FWD_TIMELINE
    struct Stamp {
        long long t0; int form; long long work;
        __device__ ~Stamp() {
            __syncthreads();
            if (threadIdx.x == 0) {
                const int slot = atomicAdd(&g_fwd_timeline_n, 1);
                if (slot < 32768) {
                    g_fwd_timeline[slot][0] = t0;
                    g_fwd_timeline[slot][1] = __builtin_amdgcn_s_memrealtime();
                    g_fwd_timeline[slot][2] = form >= 4 ? ((g_tl_phase[0] - t0) | ((g_tl_phase[1] - t0) << 32)) : 0;
                    g_fwd_timeline[slot][3] = form | (work << 8);
                }
            }
        }
    } stamp{(long long)__builtin_amdgcn_s_memrealtime(), FORM,
            (long long)((L.Kd + fBK - 1) / fBK) * ((L.variant & 1) ? 64 : 128)};
#endif
    if constexpr (FORM == 0) fwd_tile<128, 4>(L, it, smem, partials);
    else if constexpr (FORM == 1) fwd_tile<64, 4>(L, it, smem, partials);
    else if constexpr (FORM == 2) fwd_tile<128, 1>(L, it, smem, partials);
    else if constexpr (FORM == 3) fwd_tile<64, 1>(L, it, smem, partials);
    else if constexpr (FORM < 7) fwd_flat_tile<128, FORM - 4, SPLIT>(L, it, smem, partials);
    else fwd_flat_tile<64, FORM - 7, SPLIT>(L, it, smem, partials);
}
// A plain convolution (pleas_conv2d_fwd): ONE layer, described in the kernel arguments; the work item is the block index
// (output-channel tile fastest, as in the grouped plan), no tables, no target, no loss.
template <int FORM, int SPLIT = 0, int BN = 0>
__global__ __launch_bounds__(fThreads, (SPLIT && FORM >= 7) ? 3 : 2) void conv2d_fwd_kernel(const FwdLayerDev L, const int tms) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const FwdItemDev it{0, (int)(blockIdx.x % (unsigned)tms), (int)(blockIdx.x / (unsigned)tms), 0};
    constexpr int PLAIN = 1 + BN;      // 2: the BatchNorm / add / ReLU image is written beside the output
    if constexpr (FORM == 0) fwd_tile<128, 4, PLAIN>(L, it, smem, nullptr);
    else if constexpr (FORM == 1) fwd_tile<64, 4, PLAIN>(L, it, smem, nullptr);
    else if constexpr (FORM == 2) fwd_tile<128, 1, PLAIN>(L, it, smem, nullptr);
    else if constexpr (FORM == 3) fwd_tile<64, 1, PLAIN>(L, it, smem, nullptr);
    else if constexpr (FORM < 7) fwd_flat_tile<128, FORM - 4, SPLIT, PLAIN>(L, it, smem, nullptr);
    else fwd_flat_tile<64, FORM - 7, SPLIT, PLAIN>(L, it, smem, nullptr);
}
// side streams + events of the library for the concurrent forms (created once per process; no device memory)
constexpr int fLanes = 3;          // side streams (+ the caller's stream = four hardware queues)
struct FwdSideStreams {
    hipStream_t streams[fLanes];
    hipEvent_t forked, joined[fLanes];
    bool ok = false;
};
static FwdSideStreams& fwd_side_streams() {
    // one set per device (streams belong to the device that was current when they were created)
    constexpr int kMaxDev = 16;
    static FwdSideStreams sets[kMaxDev];
    static bool tried[kMaxDev] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
        static FwdSideStreams none;      // ok == false: the forms go back to back on the caller's stream
        return none;
    }
    FwdSideStreams& s = sets[dev];
    if (!tried[dev]) {
        tried[dev] = true;
        bool ok = hipEventCreateWithFlags(&s.forked, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < fLanes; ++i)
            ok = hipStreamCreateWithFlags(&s.streams[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&s.joined[i], hipEventDisableTiming) == hipSuccess;
        s.ok = ok;
    }
    return s;
}
static inline int fwd_form_of(int variant) {
    if (variant & 8) return ((variant & 1) ? 7 : 4) + ((variant >> 4) & 3);
    return variant & 3;
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
    size_t off_layers = 0, off_items = 0, off_loss = 0, off_parts = 0, total = 0;
    int form_begin[fForms] = {0}, form_count[fForms] = {0};
    size_t form_lds[fForms] = {0};
    std::vector<float> item_work;      // per item (aligned with `items`): relative duration, 0 for padding
    // A launch UNIT is a form's kernel over a contiguous slice of the form's items.  At first every form is one unit; after
    // the calibration launch a form that alone would outlast a lane's fair share is cut into slices of equal work.
    struct Unit { int form, begin, count, lane; double ms; };
    std::vector<Unit> units;           // in launch order (longest first)
    double flops = 0, bytes = 0;
    int n_parts = 0;
    bool uploaded = false;
    bool split = false;                // built under pleas_arith(PLEAS_ARITH_SPLIT_BF16): the flat forms launch their split kernels
    // lanes from MEASURED durations: launch kCalibAt of a plan brackets every form's kernel with events on its lane; a
    // later launch that finds them complete deals the forms again, longest measured duration first
    int launches = 0, calib = 0;       // calib: 0 not measured yet, 1 events recorded, 2 lanes dealt from measurements
    // the calibration launch's events, around every unit's kernel on its lane.  They belong to the PLAN (a slot of the plan
    // cache, alive for the process): two plans that calibrate at the same time must not read each other's timings.
    hipEvent_t t0[24], t1[24];
    int timed = 0;                     // 0 events not created yet, 1 created, -1 creation failed (no calibration)
    bool timing_events() {
        if (timed == 0) {
            timed = 1;
            for (int u = 0; u < 24 && timed == 1; ++u)
                if (hipEventCreate(&t0[u]) != hipSuccess || hipEventCreate(&t1[u]) != hipSuccess) timed = -1;
        }
        return timed == 1;
    }
};
constexpr int fMaxUnits = 24;
static PlanCache<FwdPlan, 1> g_fplans;
constexpr int kCalibAt = 3;
// measured durations per layer-list geometry (the plan key without its workspace address): a new fitter on the same
// layers -- every job of a bench run -- starts from the lanes the previous one measured
static std::vector<std::pair<std::vector<int64_t>, std::vector<FwdPlan::Unit>>> g_fcalib;
static std::vector<int64_t> fwd_geometry_key(const std::vector<int64_t>& key) {
    std::vector<int64_t> g(key);
    if (g.size() > 1) g[1] = 0;      // key[1] is the workspace address
    return g;
}

// longest-processing-time first over the units' expected durations: the longest unit keeps the caller's stream (lane 0) and
// is launched first, every other unit goes to the lane that is least loaded so far
static void fwd_deal_lanes(FwdPlan& P) {
    std::stable_sort(P.units.begin(), P.units.end(), [](const FwdPlan::Unit& a, const FwdPlan::Unit& b) { return a.ms > b.ms; });
    double load[fLanes + 1] = {0};
    for (size_t o = 0; o < P.units.size(); ++o) {
        int best = 0;
        for (int l = 1; l <= fLanes; ++l)
            if (load[l] < load[best]) best = l;
        if (o == 0) best = 0;
        P.units[o].lane = best;
        load[best] += P.units[o].ms;
    }
}
// After the calibration launch (measured duration per form).  What the in-job timelines showed: a form with few long items
// (stride-2 layers, the 7 x 7 layers: 400 - 900 items for 512 workgroup slots) needs WALL time whatever runs beside it,
// while the form with the most items (1 x 1 layers with K <= 256: 20 000 short ones) soaks up whatever the chip has left.
// So: (1) every form but that FILLER is cut into slices of equal work if it outlasts 0.7 of a lane's fair share, and the
// units are list-scheduled onto the least-loaded lane in ascending order of their form's item count -- low parallelism
// first; (2) the filler is cut into one slice per lane, sized so that all lanes end together, launched last on each.
static void fwd_slice_by_work(const FwdPlan& P, const FwdPlan::Unit& u, const std::vector<double>& share, std::vector<FwdPlan::Unit>& out) {
    double work = 0, want = 0;
    for (int i = 0; i < u.count; ++i) work += P.item_work[u.begin + i];
    for (double v : share) want += v;
    int begin = u.begin;
    double acc = 0, upto = 0;
    size_t final_slice = 0;          // the last slice with a share takes what rounding left over
    for (size_t k = 0; k < share.size(); ++k)
        if (share[k] > 0) final_slice = k;
    if (!(want > 0) || !(work > 0)) {      // nothing to divide by: the unit stays whole (on the first lane that wanted a share)
        out.push_back(FwdPlan::Unit{u.form, u.begin, u.count, (int)final_slice, u.ms});
        return;
    }
    for (size_t k = 0; k < share.size(); ++k) {
        upto += work * share[k] / want;
        int end = begin;
        while (end < u.begin + u.count && share[k] > 0 && (k == final_slice || acc + P.item_work[end] <= upto)) acc += P.item_work[end++];
        if (end == begin && share[k] > 0 && end < u.begin + u.count) acc += P.item_work[end++];
        if (end > begin)                   // never a slice without items (a 0-block grid is an invalid launch)
            out.push_back(FwdPlan::Unit{u.form, begin, end - begin, (int)k, u.ms * share[k] / want});   // lane field: slice index
        begin = end;
    }
}
// every form's slices must tile [form_begin, form_begin + form_count) exactly, each with at least one item
static bool fwd_units_cover(const FwdPlan& P) {
    for (int f = 0; f < fForms; ++f) {
        std::vector<std::pair<int, int>> sl;
        for (const auto& u : P.units)
            if (u.form == f) {
                if (u.count <= 0 || u.lane < 0 || u.lane > fLanes) return false;
                sl.emplace_back(u.begin, u.count);
            }
        std::sort(sl.begin(), sl.end());
        int at = P.form_begin[f];
        for (const auto& s2 : sl) {
            if (s2.first != at) return false;
            at += s2.second;
        }
        if (at != P.form_begin[f] + P.form_count[f]) return false;
    }
    return true;
}
// Returns false -- and leaves the plan's units as they were -- when the measurements are unusable (an event that failed to
// time: ms <= 0) or the result would not launch every item exactly once; the caller then keeps the static lanes.
static bool fwd_schedule_measured(FwdPlan& P) {
    double total = 0;
    for (const auto& u : P.units) {
        if (!(u.ms > 0)) return false;
        total += u.ms;
    }
    if (!(total > 0)) return false;
    const std::vector<FwdPlan::Unit> before = P.units;
    const double fair = total / (fLanes + 1);
    int filler = -1;
    for (size_t i = 0; i < P.units.size(); ++i)
        if (filler < 0 || P.units[i].count > P.units[filler].count) filler = (int)i;
    if (filler >= 0 && (P.units[filler].ms < 0.15 * total || P.units[filler].count < 8 * (fLanes + 1))) filler = -1;
    // (1) the other forms, long ones in equal slices
    std::vector<FwdPlan::Unit> rest;
    for (size_t ui = 0; ui < P.units.size(); ++ui) {
        if ((int)ui == filler) continue;
        const FwdPlan::Unit u = P.units[ui];
        int parts = (u.ms > 0.7 * fair && u.count >= 64) ? (int)std::min<double>(4.0, std::ceil(u.ms / (0.45 * fair))) : 1;
        const int room = fMaxUnits - (fLanes + 1) - (int)rest.size() - (int)(P.units.size() - ui - 1);
        parts = std::max(1, std::min(parts, room));
        if (parts <= 1) rest.push_back(u);
        else fwd_slice_by_work(P, u, std::vector<double>(parts, 1.0), rest);
    }
    std::stable_sort(rest.begin(), rest.end(), [&](const FwdPlan::Unit& a, const FwdPlan::Unit& b) {
        return P.form_count[a.form] != P.form_count[b.form] ? P.form_count[a.form] < P.form_count[b.form] : a.ms > b.ms;
    });
    double load[fLanes + 1] = {0};
    for (size_t o = 0; o < rest.size(); ++o) {
        int best = 0;
        for (int l = 1; l <= fLanes; ++l)
            if (load[l] < load[best]) best = l;
        rest[o].lane = best;
        load[best] += rest[o].ms;
    }
    // (2) the filler levels the lanes
    if (filler >= 0) {
        const FwdPlan::Unit f = P.units[filler];
        double sum = f.ms;
        for (double v : load) sum += v;
        const double level = sum / (fLanes + 1);
        std::vector<double> share(fLanes + 1);
        for (int l = 0; l <= fLanes; ++l) share[l] = std::max(0.0, level - load[l]);
        std::vector<FwdPlan::Unit> slices;
        fwd_slice_by_work(P, f, share, slices);
        for (auto& sl : slices)
            if (sl.count > 0) rest.push_back(sl);      // .lane = slice index = the lane it levels
    }
    P.units.swap(rest);
    if (!fwd_units_cover(P)) {
        P.units = before;
        return false;
    }
    return true;
}

// Geometry checks, tile height, tile form and LDS bytes of ONE layer (shared by the grouped plan and the plain convolution).
static int fwd_describe(const pleas_fwd_layer& l, FwdLayerDev& d, size_t& lds_bytes, int& TM_out) {
    if (l.N <= 0 || l.Cout <= 0 || l.Cin <= 0 || l.Hin <= 0 || l.Win <= 0 || l.KH <= 0 || l.KW <= 0 || l.stride <= 0 ||
        l.pad < 0 || l.Csrc <= 0 || l.n_merged < 0)
        return bad_arg("conv_fwd: layer geometry");
    const int Hout = (l.Hin + 2 * l.pad - l.KH) / l.stride + 1, Wout = (l.Win + 2 * l.pad - l.KW) / l.stride + 1;
    if (Hout <= 0 || Wout <= 0) return bad_arg("conv_fwd: empty output");
    if (l.KH * l.KW > 64) return bad_arg("conv_fwd: kernels larger than 64 taps are not supported");
    const int64_t HWo = (int64_t)Hout * Wout, Ptot = (int64_t)l.N * HWo, Kd = (int64_t)l.Cin * l.KH * l.KW;
    if (Ptot >= (1ll << 31) || (int64_t)l.Cout * Kd >= (1ll << 32)) return bad_arg("conv_fwd: tensor too large");
    if (Ptot * std::max(l.Cout, l.Csrc) >= (1ll << 30)) return bad_arg("conv_fwd: outputs of 4 GB and more per call are not supported");
    d.Cout = l.Cout; d.Cin = l.Cin; d.Hin = l.Hin; d.Win = l.Win; d.Hout = Hout; d.Wout = Wout;
    d.KH = l.KH; d.KW = l.KW; d.stride = l.stride; d.pad = l.pad; d.Csrc = l.Csrc; d.n_merged = l.n_merged;
    d.HWo = (uint32_t)HWo; d.Ptot = (uint32_t)Ptot; d.Kd = (uint32_t)Kd;
    d.dscale = l.dscale;
    // short-K layers (1x1 with few input channels) are bound by their epilogue's memory traffic, not by the MFMAs:
    // 64-row tiles (52 KB of LDS, <= 132 registers) let THREE workgroups share a CU and overlap more of it
    // (round 5: 128-row tiles for them too are 1.6 % faster per launch group in both arithmetics -- half the tiles re-read and, under
    // the split arithmetic, re-convert each input chunk -- so the threshold is 0 unless PLEAS_FWD_TM64_K says otherwise:
    // 2.570 -> 2.529 ms exact, 1.884 -> 1.837 ms split, profiles/r05_exp_source_conv.txt; 64-row tiles remain for Cout <= 64)
    static const int tm64_k = std::getenv("PLEAS_FWD_TM64_K") ? std::atoi(std::getenv("PLEAS_FWD_TM64_K")) : 0;
    const int TM = (l.Cout > 64 && !(l.KH * l.KW == 1 && l.stride == 1 && Kd <= tm64_k && l.Cin % fBK == 0)) ? 128 : 64;
    d.variant = (TM == 64 ? 1 : 0) | (Kd % 4 == 0 ? 0 : 2);
    if (l.flags & PLEAS_FWD_KPOS_MAJOR) {
        if (l.Cin % fBK != 0) return bad_arg("conv_fwd: kernel-position-major weights need Cin % 32 == 0");
        d.variant |= 4;
    }
    lds_bytes = (size_t)(2 * TM * fLdsA + 2 * fTN * fLdsB) * sizeof(float);  // >= TM*132 floats (epilogue)
    {
        // flat-shift form: stride 1, square odd kernel with "same" padding, whole 32-channel blocks, and (for k > 1)
        // kernel-position-major weights; its LDS must not exceed the general form's (two workgroups per CU)
        static const bool flat_on = !(std::getenv("PLEAS_FWD_FLAT") && std::atoi(std::getenv("PLEAS_FWD_FLAT")) == 0);
        const int R = l.KH * l.KW;
        const bool same = l.stride == 1 && l.KH == l.KW && (l.KH & 1) && l.pad == (l.KH - 1) / 2;
        const int halo = l.pad * (l.Win + 1);
        const int kind = R > 1 ? 2 : (HWo % 4 == 0 ? 0 : 1);
        const int Lr = kind == 2 ? fFlatColsK : fFlatRow1;      // k x k: columns of the [column][36] image, zero row included
        const size_t flat_lds = std::max((size_t)(2 * TM * fLdsA + (kind == 2 ? fFlatColsK * fFlatPix : 2 * fBK * Lr)) * sizeof(float),
                                         (size_t)TM * 132 * sizeof(float));
        if (flat_on && same && l.Cin % fBK == 0 && R <= 32 && (R == 1 || (l.flags & PLEAS_FWD_KPOS_MAJOR)) &&
            fTN + 2 * halo < Lr && flat_lds <= lds_bytes && (int64_t)l.N * l.Cin * HWo < (1ll << 32)) {
            d.variant |= 8 | (kind << 4);
            lds_bytes = flat_lds;
            if (arith_mode() == 1 && kind != 1) {
                // split-bf16 images (fwd_flat_tile<.., SPLIT = 1>): weights [TM][kSplitRow] + input k x k [columns][kSplitRow] /
                // 1 x 1 [3][32][fSplitPixRow] bf16, ONE image each; the epilogue stages [TM][132] floats in the same memory
                const size_t img = (size_t)(TM * kSplitRow + (kind == 2 ? fFlatColsK * kSplitRow : 3 * fBK * fSplitPixRow)) * sizeof(__bf16);
                lds_bytes = std::max(img, (size_t)TM * 132 * sizeof(float));
            }
        }
    }
    TM_out = TM;
    return PLEAS_OK;
}

static std::mutex g_fplan_mu;
static size_t falign(size_t v) { return (v + 255) / 256 * 256; }

static int build_fwd_plan(FwdPlan& P, const pleas_fwd_layer* ly, int n) {
    P.split = arith_mode() == 1;
    P.layers.assign(n, FwdLayerDev());
    P.items.clear();
    for (int f = 0; f < fForms; ++f) P.form_begin[f] = P.form_count[f] = 0;
    P.loss.assign(n, FwdLossDev());
    P.flops = P.bytes = 0;
    std::vector<XcdWork<FwdItemDev>> work[fForms];
    double form_work[fForms] = {0};
    for (int f = 0; f < fForms; ++f) P.form_lds[f] = 0;
    int parts = 0;
    for (int i = 0; i < n; ++i) {
        const pleas_fwd_layer& l = ly[i];
        FwdLayerDev& d = P.layers[i];
        size_t lds_bytes = 0;
        int TM = 128;
        if (const int rc = fwd_describe(l, d, lds_bytes, TM); rc != PLEAS_OK) return rc;
        const int64_t Ptot = d.Ptot, Kd = d.Kd;
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
                work[fwd_form_of(d.variant)].push_back(w);
                form_work[fwd_form_of(d.variant)] += w.w;
            }
        P.loss[i] = FwdLossDev{parts, slot, l.loss_scale, 0};
        parts += slot;
        P.form_lds[fwd_form_of(d.variant)] = std::max(P.form_lds[fwd_form_of(d.variant)], lds_bytes);
        P.flops += 2.0 * l.Cout * (double)Kd * (double)Ptot;
        P.bytes += ((double)l.Cin * l.N * l.Hin * l.Win + 3.0 * l.Cout * (double)Ptot) * sizeof(float);
    }
    // Off by default for this kernel (PLEAS_XCD_ORDER=1 turns it on): it cuts FETCH_SIZE by 32 % (8.3 -> 6.1 GB per launch)
    // but costs 1-4 % of time -- co-resident workgroups of one layer reach their latency-bound epilogues together,
    // while the plain longest-first order mixes layers on a CU.
    for (int f = 0; f < fForms; ++f) {
        P.form_begin[f] = (int)P.items.size();
        std::vector<FwdItemDev> part = xcd_order_items(work[f], FwdItemDev{-1, 0, 0, 0}, /*by_default=*/false);
        if (const char* env = std::getenv("PLEAS_FWD_ORDER")) {   // experiments: 1 = pseudo-random order, 2 = long / short interleaved
            const int mode = std::atoi(env);
            if (mode == 1) {
                uint64_t st = 0x9E3779B97F4A7C15ull;
                for (size_t i = part.size(); i > 1; --i) {
                    st = st * 6364136223846793005ull + 1442695040888963407ull;
                    std::swap(part[i - 1], part[(size_t)((st >> 33) % i)]);
                }
            } else if (mode == 2) {   // longest-first list folded: item k from the front, then item k from the back
                std::vector<FwdItemDev> folded;
                folded.reserve(part.size());
                size_t lo = 0, hi = part.size();
                while (lo < hi) {
                    folded.push_back(part[lo++]);
                    if (lo < hi) folded.push_back(part[--hi]);
                }
                part.swap(folded);
            }
        }
        P.items.insert(P.items.end(), part.begin(), part.end());
        P.form_count[f] = (int)part.size();
    }
    P.item_work.assign(P.items.size(), 0.f);
    for (size_t k = 0; k < P.items.size(); ++k)
        if (P.items[k].layer >= 0) {
            const FwdLayerDev& d = P.layers[P.items[k].layer];
            P.item_work[k] = (float)(ceil_div(d.Kd, fBK) * ((d.variant & 1) ? 64 : 128));
        }
    // launch order: the form with the most work first (its tail is then covered by nothing, the small ones' tails are short)
    // Expected duration of a form relative to its MFMA work (measured on the ResNet-101 list, each form alone): the
    // general tile (stride-2 layers: few, long items; stem: scalar weight loads) runs at ~0.4x the flat forms' rate, the
    // scalar-pixel 1x1 form (7x7 images) at ~0.5x.
    // These static weights deal the FIRST launches of a plan only; launch kCalibAt measures every form on its lane and the
    // lanes are dealt again from those durations (pleas_fwd_batch).
    P.units.clear();
    for (int f = 0; f < fForms; ++f)
        if (P.form_count[f] > 0)
            P.units.push_back(FwdPlan::Unit{f, P.form_begin[f], P.form_count[f], 0,
                                            form_work[f] * (f < 4 ? 2.5 : ((f == 5 || f == 8) ? 2.0 : 1.0))});
    fwd_deal_lanes(P);
    P.launches = P.calib = 0;
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

#if PLEAS_FWD_TIMELINE
extern "C" int pleas_fwd_timeline_read(long long* out, int max_items) {      // study builds only; resets the record
    int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_fwd_timeline_n), sizeof(int)) != hipSuccess) return -1;
    n = std::min(std::min(n, max_items), 32768);
    if (n > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_timeline), (size_t)n * 4 * sizeof(long long)) != hipSuccess) return -1;
    const int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_timeline_n), &zero, sizeof(int));
    return n;
}
#endif

}  // namespace pleas

using namespace pleas;

#if (PLEAS_FWD_ABLATE & 16)
extern "C" int pleas_fwd_debug_read(long long* out, int n_items) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_stamps), sizeof(long long) * 4 * (size_t)std::min(n_items, 32768)) == hipSuccess
               ? 0 : 1;
}
#endif

extern "C" int pleas_fwd_plan_lanes(double* form_ms, int* form_lane, int* form_items) {
    if (!form_ms || !form_lane || !form_items) return bad_arg("fwd_plan_lanes: null pointer");
    std::lock_guard<std::mutex> lk(g_fplan_mu);
    if (!g_fplans.last) return bad_arg("fwd_plan_lanes: no grouped forward has run yet");
    const FwdPlan& P = *g_fplans.last;
    for (int f = 0; f < fForms; ++f) {
        form_ms[f] = 0;
        form_lane[f] = 0;
        form_items[f] = P.form_count[f];
    }
    for (const auto& u : P.units) {      // a form cut into slices: summed duration, lanes as decimal digits (lane + 1 each)
        form_ms[u.form] += u.ms;
        form_lane[u.form] = form_lane[u.form] * 10 + u.lane + 1;
    }
    return P.calib;
}

extern "C" int pleas_fwd_plan_units(const pleas_fwd_layer* layers, int n_layers, const double* form_ms, int* units, int max_units) {
    if (!layers || n_layers <= 0 || !units || max_units <= 0) return bad_arg("fwd_plan_units: empty layer list / null output");
    FwdPlan tmp;
    const int rc = build_fwd_plan(tmp, layers, n_layers);
    if (rc != PLEAS_OK) return rc;
    if (form_ms) {       // as if the calibration launch had measured these per-form durations
        for (auto& u : tmp.units) u.ms = form_ms[u.form];
        (void)fwd_schedule_measured(tmp);      // unusable measurements: the static units stay
    }
    const int n = (int)std::min<size_t>(tmp.units.size(), (size_t)max_units);
    for (int i = 0; i < n; ++i) {
        units[4 * i + 0] = tmp.units[i].form;
        units[4 * i + 1] = tmp.units[i].begin;
        units[4 * i + 2] = tmp.units[i].count;
        units[4 * i + 3] = tmp.units[i].lane;
    }
    return (int)tmp.units.size();
}

extern "C" size_t pleas_fwd_batch_ws_bytes(const pleas_fwd_layer* layers, int n_layers) {
    if (!layers || n_layers <= 0) return 0;
    FwdPlan tmp;
    if (build_fwd_plan(tmp, layers, n_layers) != PLEAS_OK) return 0;
    return tmp.total;
}

// y = conv(x, w) (+ bias); with `scale`: also z = act(y * scale[c] + shift[c] (+ res)) in the same epilogue
static int conv2d_launch(const float* x, const float* w, const float* bias, float* y, const float* scale, const float* shift,
                         const float* res, float* z, int relu, int N, int Cin, int Hin, int Win, int Cout, int KH, int KW,
                         int stride, int pad, int flags, void* stream_) {
    if (!x || !w || !y) return bad_arg("conv2d_fwd: null pointer");
    if (((uintptr_t)w & 15) != 0 || ((uintptr_t)x & 15) != 0) return bad_arg("conv2d_fwd: x and w must be 16-byte aligned");
    const bool bn = scale != nullptr;
    if (bn && (!shift || !z)) return bad_arg("conv2d_bn_act_fwd: scale needs shift and z");
    if (bn && ((((uintptr_t)y | (uintptr_t)z | (uintptr_t)res) & 15) != 0))
        return bad_arg("conv2d_bn_act_fwd: y, z and res must be 16-byte aligned");
    pleas_fwd_layer l{};
    l.ip = x; l.w = w; l.bias = bias; l.resid = y;
    l.N = N; l.Cout = Cout; l.Cin = Cin; l.Hin = Hin; l.Win = Win; l.KH = KH; l.KW = KW; l.stride = stride; l.pad = pad;
    l.Csrc = 1; l.n_merged = 0; l.dscale = 1.f; l.loss_scale = 0.f; l.flags = flags;
    FwdLayerDev d{};
    size_t lds = 0;
    int TM = 128;
    if (const int rc = fwd_describe(l, d, lds, TM); rc != PLEAS_OK) return rc;
    // no block maps: every source row is absent (fwd_load_maps), the masked gathers read the first floats of x
    d.ip = x; d.w = w; d.bias = bias; d.o1 = x; d.o2 = x; d.row1 = nullptr; d.row2 = nullptr; d.resid = y;
    d.part_base = 0;
    d.bn_scale = scale; d.bn_shift = shift; d.bn_res = res; d.bn_z = z; d.bn_relu = relu ? 1 : 0;
    const int tms = (int)ceil_div(Cout, TM), tps = (int)ceil_div((int64_t)d.Ptot, fTN);
    const dim3 grid((unsigned)((int64_t)tms * tps));
    hipStream_t st = (hipStream_t)stream_;
    const double out_floats = (double)Cout * d.Ptot;
    ProfScope prof(kProfConv2d, 2.0 * Cout * (double)d.Kd * (double)d.Ptot,
                   ((double)Cin * N * Hin * Win + out_floats * (bn ? (res ? 3.0 : 2.0) : 1.0)) * sizeof(float), st);
    const int form = fwd_form_of(d.variant);
    const bool split = arith_mode() == 1 && fwd_form_splits(form);
#define PLEAS_CONV_LAUNCH(F, S, B) case F: hipLaunchKernelGGL((conv2d_fwd_kernel<F, S, B>), grid, dim3(fThreads), lds, st, d, tms); break
#define PLEAS_CONV_FORMS(B)                                                                                              \
    if (split) {                                                                                                         \
        switch (form) {                                                                                                  \
            PLEAS_CONV_LAUNCH(4, 1, B); PLEAS_CONV_LAUNCH(6, 1, B); PLEAS_CONV_LAUNCH(7, 1, B); PLEAS_CONV_LAUNCH(9, 1, B); \
        }                                                                                                                \
    } else {                                                                                                             \
        switch (form) {                                                                                                  \
            PLEAS_CONV_LAUNCH(0, 0, B); PLEAS_CONV_LAUNCH(1, 0, B); PLEAS_CONV_LAUNCH(2, 0, B); PLEAS_CONV_LAUNCH(3, 0, B); \
            PLEAS_CONV_LAUNCH(4, 0, B); PLEAS_CONV_LAUNCH(5, 0, B); PLEAS_CONV_LAUNCH(6, 0, B); PLEAS_CONV_LAUNCH(7, 0, B); \
            PLEAS_CONV_LAUNCH(8, 0, B); PLEAS_CONV_LAUNCH(9, 0, B);                                                      \
        }                                                                                                                \
    }
    if (bn) {
        PLEAS_CONV_FORMS(1)
    } else {
        PLEAS_CONV_FORMS(0)
    }
#undef PLEAS_CONV_FORMS
#undef PLEAS_CONV_LAUNCH
    PLEAS_LAUNCH_CHECK("conv2d_fwd_kernel");
    return PLEAS_OK;
}

extern "C" int pleas_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Hin, int Win,
                                int Cout, int KH, int KW, int stride, int pad, int flags, void* stream_) {
    return conv2d_launch(x, w, bias, y, nullptr, nullptr, nullptr, nullptr, 0, N, Cin, Hin, Win, Cout, KH, KW, stride, pad, flags,
                         stream_);
}

extern "C" int pleas_conv2d_bn_act_fwd(const float* x, const float* w, const float* bias, float* y, const float* scale,
                                       const float* shift, const float* res, float* z, int relu, int N, int Cin, int Hin,
                                       int Win, int Cout, int KH, int KW, int stride, int pad, int flags, void* stream_) {
    if (!scale || !shift || !z) return bad_arg("conv2d_bn_act_fwd: null pointer");
    return conv2d_launch(x, w, bias, y, scale, shift, res, z, relu, N, Cin, Hin, Win, Cout, KH, KW, stride, pad, flags, stream_);
}

extern "C" int pleas_fwd_batch(const pleas_fwd_layer* layers, int n_layers, float* loss, void* ws, size_t ws_bytes,
                               int ws_fresh, void* stream_) {
    if (!layers || n_layers <= 0 || !loss) return bad_arg("conv_fwd: empty layer list");
    for (int i = 0; i < n_layers; ++i) {
        const pleas_fwd_layer& l = layers[i];
        if (!l.ip || !l.w || !l.o1 || !l.o2 || !l.row1 || !l.row2 || !l.resid) return bad_arg("conv_fwd: null pointer");
        if (((uintptr_t)l.w & 15) != 0) return bad_arg("conv_fwd: weights must be 16-byte aligned");
        if (((uintptr_t)l.ip & 15) != 0) return bad_arg("conv_fwd: the merged input must be 16-byte aligned");
    }
    hipStream_t stream = (hipStream_t)stream_;
    std::lock_guard<std::mutex> lk(g_fplan_mu);
    std::vector<int64_t> key;
    key.push_back(n_layers * 2 + arith_mode());      // plans (LDS sizes, kernels) differ between the arithmetics
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
    FwdPlan* hit = g_fplans.find(key);
    if (!hit) {
        hit = &g_fplans.take();
        const int rc = build_fwd_plan(*hit, layers, n_layers);
        if (rc != PLEAS_OK) return rc;
        hit->key.swap(key);
    }
    FwdPlan& P = *hit;
    if (P.calib == 0 && P.launches == 0) {      // a new plan: start from what another fitter measured on this geometry
        const std::vector<int64_t> geo = fwd_geometry_key(P.key);
        for (const auto& kv : g_fcalib)
            if (kv.first == geo) {
                P.units = kv.second;
                P.calib = 2;
            }
    }
    if (ws_fresh) P.uploaded = false;
    if (!ws || ws_bytes < P.total) {
        std::snprintf(g_last_error, sizeof(g_last_error), "conv_fwd workspace too small: need %zu bytes", P.total);
        P.key.clear();
        return PLEAS_ENOMEM;
    }
    char* base = (char*)ws;
    if (!P.uploaded) {
        g_fplans.claims_workspace(P);
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
        // The forms are independent grids.  Back to back on one stream each would wait for the previous one's LAST
        // workgroup (a stride-2 3x3 layer has 28 items of 144 chunks: half a millisecond of tail on an empty chip), so every
        // form but the largest goes to one of THREE side streams of the library (fork / join with events around the group;
        // the runtime multiplexes a process's streams onto four hardware queues, so more lanes would only queue up behind
        // each other): forms are dealt to the least-loaded lane, largest first, and the long items of the small forms run
        // beside the large forms' thousands of short ones, as in one grid.
        ProfScope prof(kProfConvFwd, P.flops, P.bytes, stream);
        const FwdItemDev* items = reinterpret_cast<const FwdItemDev*>(base + P.off_items);
        FwdSideStreams& side = fwd_side_streams();
        static const bool serial = std::getenv("PLEAS_FWD_SERIAL") && std::atoi(std::getenv("PLEAS_FWD_SERIAL")) != 0;
        const int active = (int)P.units.size();
        const bool fork = !serial && active > 1 && side.ok;
        static const bool calibrate = !(std::getenv("PLEAS_FWD_CALIBRATE") && std::atoi(std::getenv("PLEAS_FWD_CALIBRATE")) == 0);
        ++P.launches;
        if (P.calib == 1) {          // the calibration launch's events: all complete?  then cut long units and deal the lanes
            bool ready = true;
            for (int u = 0; ready && u < active; ++u) ready = hipEventQuery(P.t1[u]) == hipSuccess;
            if (ready) {
                for (int u = 0; u < active; ++u) {
                    float ms = 0.f;
                    P.units[u].ms = hipEventElapsedTime(&ms, P.t0[u], P.t1[u]) == hipSuccess ? ms : 0.0;
                }
                P.calib = 2;
                if (fwd_schedule_measured(P)) {      // else: static lanes kept, nothing published to other fitters
                    if (g_fcalib.size() >= 16) g_fcalib.erase(g_fcalib.begin());
                    g_fcalib.emplace_back(fwd_geometry_key(P.key), P.units);
                }
            }
            (void)hipGetLastError();   // hipEventQuery's "not ready" is not an error of this call
        }
        const bool measure = fork && calibrate && P.calib == 0 && P.launches == kCalibAt && active <= fMaxUnits && P.timing_events();
        if (fork) PLEAS_HIP_CHECK(hipEventRecord(side.forked, stream));
        bool lane_used[fLanes + 1] = {false};
        for (size_t o = 0; o < P.units.size(); ++o) {
            const FwdPlan::Unit& un = P.units[o];
            const int f = un.form;
            const int lane = fork ? un.lane : 0;          // lane 0 = the caller's stream
            hipStream_t st = lane == 0 ? stream : side.streams[lane - 1];
            if (lane > 0 && !lane_used[lane]) PLEAS_HIP_CHECK(hipStreamWaitEvent(st, side.forked, 0));
            lane_used[lane] = true;
            const dim3 grid((unsigned)un.count);
            const FwdItemDev* its = items + un.begin;
            const size_t lds = P.form_lds[f];
            if (measure) PLEAS_HIP_CHECK(hipEventRecord(P.t0[o], st));
            if (P.split && fwd_form_splits(f)) {
                switch (f) {
#define PLEAS_FWD_LAUNCH(F) case F: hipLaunchKernelGGL((fwd_batch_kernel<F, 1>), grid, dim3(fThreads), lds, st, dl, its, parts); break
                    PLEAS_FWD_LAUNCH(4); PLEAS_FWD_LAUNCH(6); PLEAS_FWD_LAUNCH(7); PLEAS_FWD_LAUNCH(9);
#undef PLEAS_FWD_LAUNCH
                }
            } else
            switch (f) {
#define PLEAS_FWD_LAUNCH(F) case F: hipLaunchKernelGGL((fwd_batch_kernel<F, 0>), grid, dim3(fThreads), lds, st, dl, its, parts); break
                PLEAS_FWD_LAUNCH(0); PLEAS_FWD_LAUNCH(1); PLEAS_FWD_LAUNCH(2); PLEAS_FWD_LAUNCH(3); PLEAS_FWD_LAUNCH(4);
                PLEAS_FWD_LAUNCH(5); PLEAS_FWD_LAUNCH(6); PLEAS_FWD_LAUNCH(7); PLEAS_FWD_LAUNCH(8); PLEAS_FWD_LAUNCH(9);
#undef PLEAS_FWD_LAUNCH
            }
            if (measure) PLEAS_HIP_CHECK(hipEventRecord(P.t1[o], st));
        }
        if (measure) P.calib = 1;
        for (int lane = 1; lane <= fLanes; ++lane)
            if (lane_used[lane]) {
                PLEAS_HIP_CHECK(hipEventRecord(side.joined[lane - 1], side.streams[lane - 1]));
                PLEAS_HIP_CHECK(hipStreamWaitEvent(stream, side.joined[lane - 1], 0));
            }
    }
    PLEAS_LAUNCH_CHECK("fwd_batch_kernel");
    hipLaunchKernelGGL(fwd_loss_kernel, dim3(n_layers), dim3(64), 0, stream, parts,
                       reinterpret_cast<const FwdLossDev*>(base + P.off_loss), loss);
    PLEAS_LAUNCH_CHECK("fwd_loss_kernel");
    return PLEAS_OK;
}

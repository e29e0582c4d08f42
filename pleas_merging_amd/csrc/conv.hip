// PLeaS layer fitting on gfx950: weight gradients of ALL merged layers of one update in one
// grouped fp32-MFMA launch, plus the fused target / residual / loss pass.
//
// Replaces, per layer and per update, the autograd backward of
//     loss_l = mean((layer(ip) - op)^2)          pleas/methods/pleas_merging.py:281-287
// i.e.  gW[co][ci][kh][kw] = sum_{n,oh,ow} resid[n][co][oh][ow] * ip[n][ci][oh*s+kh-p][ow*s+kw-p]
// with  resid = 2 (out - op) / numel,  op = block-merged source outputs (:116-147).
//
// Formulation: for a FIXED kernel position r = (kh, kw) the gradient slice gW[:, :, r] is an NT
// contraction over the flattened pixel index P = (n, oh, ow) between
//     X = resid viewed [N][Cout][HWo]        (rows = co, P contiguous)  and
//     Y_r = the r-shifted, strided view of ip (rows = ci, P contiguous for stride 1),
// exactly the structure of the matching contraction (gram.hip): both operands are read in place
// from NCHW, no im2col buffer, no NHWC transposes.  A work item is (layer, co-tile, ci-tile, r,
// P-range); items of all layers are sorted longest-first into one grid.  Layers whose P range fits
// one item write their gradient directly; longer ones write slabs that a second grid sums in a
// fixed order (deterministic, no atomics).
//
// Roofline: 2*Cout*Cin*KH*KW*N*HWo flop per layer and update (ResNet-101, batch 16: 2.5e11 flop per
// update over all layers), fp32 MFMA bound (157.3 TFLOP/s).
#include <algorithm>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace pleas {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef f32x4 f32x4u __attribute__((aligned(4)));   // 16 bytes at a 4-byte-aligned address: still ONE global_load_dwordx4

constexpr int cBK = 32;
constexpr int cLds = 36;
constexpr int cThreads = 256;

struct WgradLayerDev {
    const float* resid;  // [N][Cout][HWo]
    const float* ip;     // [N][Cin][Hin][Win]
    float* out;          // gradient, standard [Cout][Cin][KH][KW]
    float* slab;         // [S][Cout][R*Cin] when S > 1
    int Cout, Cin, Hin, Win, Hout, Wout, KH, KW, stride, pad;
    uint32_t HWo, Ktot;  // Ktot = N * HWo
    int S;
    int variant;         // bit0: TM==64, bit1: TN==64, bit2: X scalar loads, bit3: Y shifted (scalar) loader,
                         // bit4: Y shifted through aligned 16-byte loads (stride-1 "same" layers; X carries the border mask)
    int flags;           // PLEAS_WGRAD_ACCUMULATE | PLEAS_WGRAD_KPOS_MAJOR
    int total;           // floats in ip (bit4 form: clamps the shifted operand's loads)
    // bit5 of variant, "virtual channels" (layers with fewer than 16 input channels, e.g. the 3-channel 7x7 stem): the
    // N axis of the contraction is j = (ci, kh, kw) flattened -- Cin / KH / KW above then describe that 1x1 VIEW
    // (Cin = real Cin * R, KH = KW = 1: epilogue and slab reduce see an ordinary 1x1 layer whose standard layout
    // [Cout][Cin*R] IS the real layer's [Cout][Cin][KH][KW]) and these three the real layer for the loader
    int rCin, rKW, rR;
};
struct WgradItemDev {
    int layer, tm, tn, r, split, c_begin, c_end, pad;
};

// YMODE: 0 = Y read in place (1x1, stride 1), 1 = shifted / strided view, one pixel per load, 2 = stride-1 "same" layer
// with HW % 4 == 0: tap r reads the SAME flat pixel run `delta = dh * W + dw` further on, so a thread's four pixels are
// ONE 16-byte load at a 4-byte-aligned address (the hardware takes it; two aligned loads + a register shift move twice the
// bytes through the 64 B/clk vector L1 and measured 5 % SLOWER than one pixel per load); nothing of Y is masked -- the
// taps that fall off the image are voided by zeroing the RESIDUAL at those output pixels (a rectangle of (oh, ow)).
// SPLIT = 1 (pleas_arith(PLEAS_ARITH_SPLIT_BF16); 16-byte-loadable operands only: YMODE 0 / 2): the chunk goes to LDS as three
// bf16 planes per row (common.hpp), ONE image per operand, two barriers per chunk, six v_mfma_f32_32x32x16_bf16 per k step.
template <int TM, int TN, int VECX, int YMODE, int SPLIT = 0>
__device__ __forceinline__ void wgrad_tile(const WgradLayerDev& L, const WgradItemDev& it, float* smem) {
    static_assert(!SPLIT || (VECX == 4 && (YMODE == 0 || YMODE == 2)), "the split image is written four k at a time");
    constexpr int MTM = TM / 64, MTN = TN / 64;
    constexpr bool YSHIFT = YMODE == 1;
    constexpr bool YROWS = YMODE == 3;     // every row of the Y tile has its own (channel, tap): virtual channels
    constexpr int VECY = (YMODE == 1 || YMODE == 3) ? 1 : (YMODE == 2 ? 4 : VECX);
    static_assert(YMODE != 2 || VECX == 4, "the 16-byte shifted form needs HW % 4 == 0, hence vector loads of X too");
    constexpr int LPR_X = cBK / VECX, RPP_X = cThreads / LPR_X, PASS_X = TM / RPP_X;
    constexpr int LPR_Y = cBK / VECY, RPP_Y = cThreads / LPR_Y, PASS_Y = TN / RPP_Y;
    float* As = smem;                    // [2][TM][cLds]
    float* Bs = smem + 2 * TM * cLds;    // [2][TN][cLds]
    __bf16* As16 = reinterpret_cast<__bf16*>(smem);      // SPLIT: [TM][kSplitRow], then [TN][kSplitRow]
    __bf16* Bs16 = As16 + TM * kSplitRow;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = it.tm * TM, j0 = it.tn * TN;
    const int kh = it.r / L.KW, kw = it.r - kh * L.KW;
    const int dh = kh - L.pad, dw = kw - L.pad;
    const uint32_t HWi = (uint32_t)L.Hin * L.Win;

    // SPLIT (16-byte loads of both operands: 8 lanes per row, 32 rows per pass): bank-conflict-free plane writes
    const int xrow = SPLIT ? split_stage_row(tid / LPR_X) : tid / LPR_X, xcol = (tid % LPR_X) * VECX;
    const int yrow = SPLIT ? split_stage_row(tid / LPR_Y) : tid / LPR_Y, ycol = (tid % LPR_Y) * VECY;
    float rx[PASS_X][VECX], ry[PASS_Y][VECY];
    unsigned okx = 0, oky = 0;
    unsigned win = 0xFu;                  // YMODE 2: bit e = output pixel e of this thread's run has tap r inside the image
    const int delta = dh * L.Win + dw;    // YMODE 2
    const int oh0 = max(0, -dh), oh1 = L.Hout - 1 - max(0, dh), ow0 = max(0, -dw), ow1 = L.Wout - 1 - max(0, dw);
    uint32_t offx[PASS_X], offy[PASS_Y];  // < 2^32: one sample's C*HW slab
#pragma unroll
    for (int q = 0; q < PASS_X; ++q) {
        const int gi = i0 + xrow + q * RPP_X;
        if (gi < L.Cout) okx |= 1u << q;
        offx[q] = (uint32_t)min(gi, L.Cout - 1) * L.HWo;
    }
    int tapq[YROWS ? PASS_Y : 1];         // YROWS: (kh - pad) | (kw - pad) << 16 of this thread's rows
    unsigned okyc = 0;                    // YROWS: rows whose tap is inside the image at this thread's pixel of the chunk
#pragma unroll
    for (int q = 0; q < PASS_Y; ++q) {
        const int gj = j0 + yrow + q * RPP_Y;
        if (gj < L.Cin) oky |= 1u << q;
        if constexpr (YROWS) {
            const int jj = min(gj, L.Cin - 1), cj = jj / L.rR, rr = jj - cj * L.rR;
            const int khq = rr / L.rKW, kwq = rr - khq * L.rKW;
            tapq[q] = ((khq - L.pad) & 0xffff) | ((kwq - L.pad) << 16);
            offy[q] = (uint32_t)cj * HWi;
        } else {
            offy[q] = (uint32_t)min(gj, L.Cin - 1) * HWi;
        }
    }
    f32x16 acc[MTM][MTN];
#pragma unroll
    for (int a = 0; a < MTM; ++a)
#pragma unroll
        for (int b = 0; b < MTN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    bool kinx = false, kiny = false;
    // (sample, pixel, row, column) of this thread's first pixel of the chunk being loaded, for the X and the Y operand:
    // ONE set of divisions per work item, then steps of cBK pixels (chunks are loaded in order).  The divisions per chunk --
    // two to four, ~25 vector instructions each -- were a third of the loop's vector work, and vector instructions are what
    // the fp32 MFMAs wait for.  Images smaller than a chunk keep the divisions.
    struct Cursor { uint32_t P, n, p; int oh, ow; };
    constexpr bool XHW = YMODE == 2, YHW = YSHIFT || YROWS;      // who needs (row, column)
    const bool step_ok = L.HWo >= (uint32_t)cBK;
    const int q32 = cBK / L.Wout, r32 = cBK - q32 * L.Wout;
    auto cur_at = [&](Cursor& k, uint32_t P, bool hw) {
        k.P = P;
        k.n = P / L.HWo;
        k.p = P - k.n * L.HWo;
        k.oh = k.ow = 0;
        if (hw) {
            k.oh = (int)(k.p / (uint32_t)L.Wout);
            k.ow = (int)(k.p - (uint32_t)k.oh * L.Wout);
        }
    };
    auto cur_step = [&](Cursor& k, bool hw) {
        if (!step_ok) {
            cur_at(k, k.P + cBK, hw);
            return;
        }
        k.P += cBK;
        k.p += cBK;
        if (hw) {
            k.oh += q32;
            k.ow += r32;
            if (k.ow >= L.Wout) { k.ow -= L.Wout; ++k.oh; }
        }
        if (k.p >= L.HWo) {
            k.p -= L.HWo;
            ++k.n;
            if (hw) k.oh -= L.Hout;
        }
    };
    Cursor cx, cy;
    cur_at(cx, (uint32_t)it.c_begin * cBK + xcol, XHW);
    cur_at(cy, (uint32_t)it.c_begin * cBK + ycol, YHW);
    auto load_chunk = [&](int c) {
        {   // X: residual rows, direct
            (void)c;
            kinx = cx.P < L.Ktot;
            const uint32_t n = kinx ? cx.n : 0u;
            const uint32_t p = kinx ? cx.p : 0u;
            const size_t base = (size_t)n * L.Cout * L.HWo + p;
            if constexpr (YMODE == 2) {
                int oh = kinx ? cx.oh : 0, ow = kinx ? cx.ow : 0;
                win = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (oh >= oh0 && oh <= oh1 && ow >= ow0 && ow <= ow1) win |= 1u << e;
                    if (++ow == L.Wout) { ow = 0; ++oh; }
                }
            }
#pragma unroll
            for (int q = 0; q < PASS_X; ++q) {
                if constexpr (VECX == 4) {
                    const f32x4 v = *(const __attribute__((address_space(1))) f32x4*)(PLEAS_GLOBAL(L.resid) + base + offx[q]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rx[q][e] = v[e];
                } else {
                    rx[q][0] = PLEAS_GLOBAL(L.resid)[base + offx[q]];
                }
            }
        }
        {   // Y: input rows, shifted by the kernel position (or direct for 1x1 stride 1)
            bool in = cy.P < L.Ktot;
            const uint32_t n = in ? cy.n : 0u;
            const uint32_t p = in ? cy.p : 0u;
            size_t base;
            if constexpr (YROWS) {
                const int oh = in ? cy.oh : 0, ow = in ? cy.ow : 0;
                const int ihb = oh * L.stride, iwb = ow * L.stride;
                base = (size_t)n * L.rCin * HWi;
                okyc = 0;
#pragma unroll
                for (int q = 0; q < PASS_Y; ++q) {
                    const int ih = ihb + (int)(short)(tapq[q] & 0xffff), iw = iwb + (tapq[q] >> 16);
                    const bool inq = in && ih >= 0 && ih < L.Hin && iw >= 0 && iw < L.Win;
                    okyc |= (inq ? 1u : 0u) << q;
                    ry[q][0] = PLEAS_GLOBAL(L.ip)[base + offy[q] + (inq ? (size_t)ih * L.Win + iw : 0)];
                }
            } else if constexpr (YSHIFT) {
                const int oh = in ? cy.oh : 0, ow = in ? cy.ow : 0;
                const int ih = oh * L.stride + dh, iw = ow * L.stride + dw;
                in = in && ih >= 0 && ih < L.Hin && iw >= 0 && iw < L.Win;
                base = (size_t)n * L.Cin * HWi + (in ? (size_t)ih * L.Win + iw : 0);
            } else {
                base = (size_t)n * L.Cin * HWi + p;
            }
            kiny = in;
            if constexpr (YROWS) {
                // loaded above, row by row
            } else if constexpr (YMODE == 2) {
                // whatever lies outside the tensor lies outside its image, i.e. at an output pixel whose residual is zeroed:
                // only the run that straddles the tensor's first / last float goes element by element, clamped
                const int b0 = (int)base + delta;
#pragma unroll
                for (int q = 0; q < PASS_Y; ++q) {
                    const int g = b0 + (int)offy[q];
                    if (g >= 0 && g <= L.total - 4) {
                        const f32x4 v = *(const __attribute__((address_space(1))) f32x4u*)(PLEAS_GLOBAL(L.ip) + g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) ry[q][e] = v[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ry[q][e] = PLEAS_GLOBAL(L.ip)[min(max(g + e, 0), L.total - 1)];
                    }
                }
            } else
#pragma unroll
            for (int q = 0; q < PASS_Y; ++q) {
                if constexpr (VECY == 4) {
                    const f32x4 v = *(const __attribute__((address_space(1))) f32x4*)(PLEAS_GLOBAL(L.ip) + base + offy[q]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ry[q][e] = v[e];
                } else {
                    ry[q][0] = PLEAS_GLOBAL(L.ip)[base + offy[q]];
                }
            }
        }
        cur_step(cx, XHW);
        cur_step(cy, YHW);
    };
    auto store_chunk = [&](int buf) {
        float* a = As + buf * TM * cLds;
        float* b = Bs + buf * TN * cLds;
#pragma unroll
        for (int q = 0; q < PASS_X; ++q) {
            const bool ok = kinx && ((okx >> q) & 1u);
            const int row = xrow + q * RPP_X;
            if constexpr (SPLIT) {
                split3_store4(As16 + row * kSplitRow, xcol, (ok && (win & 1u)) ? rx[q][0] : 0.f, (ok && (win & 2u)) ? rx[q][1] : 0.f,
                              (ok && (win & 4u)) ? rx[q][2] : 0.f, (ok && (win & 8u)) ? rx[q][3] : 0.f);
            } else if constexpr (VECX == 4) {
                f32x4 v = {(ok && (win & 1u)) ? rx[q][0] : 0.f, (ok && (win & 2u)) ? rx[q][1] : 0.f,
                           (ok && (win & 4u)) ? rx[q][2] : 0.f, (ok && (win & 8u)) ? rx[q][3] : 0.f};
                *reinterpret_cast<f32x4*>(a + row * cLds + xcol) = v;
            } else {
                a[row * cLds + xcol] = ok ? rx[q][0] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < PASS_Y; ++q) {
            const bool ok = (YROWS ? ((okyc >> q) & 1u) != 0 : kiny) && ((oky >> q) & 1u);
            const int row = yrow + q * RPP_Y;
            if constexpr (SPLIT) {
                const bool keep = YMODE == 2 ? ((oky >> q) & 1u) != 0 : ok;
                split3_store4(Bs16 + row * kSplitRow, ycol, keep ? ry[q][0] : 0.f, keep ? ry[q][1] : 0.f, keep ? ry[q][2] : 0.f,
                              keep ? ry[q][3] : 0.f);
            } else if constexpr (YMODE == 2) {
                const bool rowok = (oky >> q) & 1u;
                f32x4 v = {rowok ? ry[q][0] : 0.f, rowok ? ry[q][1] : 0.f, rowok ? ry[q][2] : 0.f, rowok ? ry[q][3] : 0.f};
                *reinterpret_cast<f32x4*>(b + row * cLds + ycol) = v;
            } else if constexpr (VECY == 4) {
                f32x4 v = {ok ? ry[q][0] : 0.f, ok ? ry[q][1] : 0.f, ok ? ry[q][2] : 0.f, ok ? ry[q][3] : 0.f};
                *reinterpret_cast<f32x4*>(b + row * cLds + ycol) = v;
            } else {
                b[row * cLds + ycol] = ok ? ry[q][0] : 0.f;
            }
        }
    };
    auto compute = [&](int buf) {
        if constexpr (SPLIT) {
            // lane (r, h) of k group g reads k = 16 g + 8 h .. + 7 of its row from each plane: the operand map of the MFMA
            const __bf16* a16 = As16 + (wm * (TM / 2) + (lane & 31)) * kSplitRow + 8 * (lane >> 5);
            const __bf16* b16 = Bs16 + (wn * (TN / 2) + (lane & 31)) * kSplitRow + 8 * (lane >> 5);
#pragma unroll
            for (int g16 = 0; g16 < cBK / 16; ++g16) {
                bf16x8_t sa[MTM][3], sb[MTN][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int s_ = 0; s_ < MTM; ++s_) sa[s_][p] = *reinterpret_cast<const bf16x8_t*>(a16 + s_ * 32 * kSplitRow + p * 32 + g16 * 16);
#pragma unroll
                    for (int s_ = 0; s_ < MTN; ++s_) sb[s_][p] = *reinterpret_cast<const bf16x8_t*>(b16 + s_ * 32 * kSplitRow + p * 32 + g16 * 16);
                }
#pragma unroll
                for (int sm = 0; sm < MTM; ++sm)
#pragma unroll
                    for (int sn = 0; sn < MTN; ++sn) acc[sm][sn] = split3_mfma(sa[sm], sb[sn], acc[sm][sn]);
            }
            return;
        }
        const float* a = As + buf * TM * cLds + (wm * (TM / 2) + (lane & 31)) * cLds + 4 * (lane >> 5);
        const float* b = Bs + buf * TN * cLds + (wn * (TN / 2) + (lane & 31)) * cLds + 4 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < cBK / 8; ++kk) {
            f32x4 fa[MTM], fb[MTN];
#pragma unroll
            for (int s = 0; s < MTM; ++s) fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * cLds + kk * 8);
#pragma unroll
            for (int s = 0; s < MTN; ++s) fb[s] = *reinterpret_cast<const f32x4*>(b + s * 32 * cLds + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sm = 0; sm < MTM; ++sm)
#pragma unroll
                    for (int sn = 0; sn < MTN; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
        }
    };

    if (it.c_begin < it.c_end) {
        load_chunk(it.c_begin);
        store_chunk(0);
    }
    __syncthreads();
    if constexpr (SPLIT) {
        for (int c = it.c_begin; c < it.c_end; ++c) {
            const bool more = c + 1 < it.c_end;
            if (more) load_chunk(c + 1);      // stays in registers while this chunk is multiplied
            compute(0);
            __syncthreads();                  // every wave is done reading the image
            if (more) store_chunk(0);
            __syncthreads();
        }
    } else
    for (int c = it.c_begin; c < it.c_end; ++c) {
        const int buf = (c - it.c_begin) & 1;
        const bool more = c + 1 < it.c_end;
        if (more) load_chunk(c + 1);
        compute(buf);
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // epilogue: direct (S == 1) into the gradient ([Cout][Cin][R], or kernel-position-major [Cout][R][Cin]), else into this
    // split's slab ([S][Cout][R][Cin]).  Wherever the Cin axis is contiguous in the destination (slabs, kernel-position-major
    // gradients, 1x1 layers) the accumulators go through LDS once ([co][TN + 4]; the staging buffers are free after the K
    // loop's last barrier) and every thread writes 16-byte runs along ci; otherwise one element per lane as before.
    const int R = L.KH * L.KW;
    const bool kpos = (L.flags & PLEAS_WGRAD_KPOS_MAJOR) != 0;
    const bool rows = (L.S > 1 || kpos || R == 1) && (L.Cin & 3) == 0 && (((size_t)(L.S > 1 ? L.slab : L.out)) & 15) == 0;
    if (rows) {
        // SPLIT tiles of 128 rows stage one 64-row half at a time (the waves of row half h): the staging tile then fits the
        // split images' 52 KB and three workgroups share a CU
        constexpr int HALVES = (SPLIT && TM == 128) ? 2 : 1, HM = TM / HALVES;
        constexpr int EL = TN + 4, VPT = HM * TN / 4 / cThreads;
        float* Ct = smem;
        const size_t row_len = (size_t)R * L.Cin;          // floats per output channel in the row-contiguous layouts
        gfloat* dst = L.S > 1 ? PLEAS_GLOBAL_W(L.slab) + (size_t)it.split * L.Cout * row_len : PLEAS_GLOBAL_W(L.out);
        const bool add = L.S == 1 && (L.flags & PLEAS_WGRAD_ACCUMULATE);
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            if (HALVES == 1 || wm == h) {
#pragma unroll
                for (int sm = 0; sm < MTM; ++sm)
#pragma unroll
                    for (int sn = 0; sn < MTN; ++sn) {
                        const int lci = wn * (TN / 2) + sn * 32 + (lane & 31);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int lco = (HALVES == 1 ? wm * (TM / 2) : 0) + sm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            Ct[lco * EL + lci] = acc[sm][sn][r];
                        }
                    }
            }
            __syncthreads();
            const int r0 = i0 + h * HM;
            f32x4 old[VPT];
            if (add) {
#pragma unroll
                for (int q = 0; q < VPT; ++q) {
                    const int v = tid + q * cThreads, lco = v / (TN / 4), lci = (v % (TN / 4)) * 4;
                    const bool in = r0 + lco < L.Cout && j0 + lci < L.Cin;
                    old[q] = *(const __attribute__((address_space(1))) f32x4*)(dst + (in ? (size_t)(r0 + lco) * row_len + (size_t)it.r * L.Cin + j0 + lci : 0));
                }
            }
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
                const int v = tid + q * cThreads, lco = v / (TN / 4), lci = (v % (TN / 4)) * 4;
                if (r0 + lco < L.Cout && j0 + lci < L.Cin) {
                    f32x4 val = *reinterpret_cast<const f32x4*>(Ct + lco * EL + lci);
                    if (add) val += old[q];
                    *(__attribute__((address_space(1))) f32x4*)(dst + (size_t)(r0 + lco) * row_len + (size_t)it.r * L.Cin + j0 + lci) = val;
                }
            }
            if (h + 1 < HALVES) __syncthreads();      // the half is written out: its staging rows are free again
        }
        return;
    }
#pragma unroll
    for (int sm = 0; sm < MTM; ++sm)
#pragma unroll
        for (int sn = 0; sn < MTN; ++sn) {
            const int ci = j0 + wn * (TN / 2) + sn * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = i0 + wm * (TM / 2) + sm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < L.Cout && ci < L.Cin) {
                    if (L.S == 1) {
                        const size_t o = kpos ? (size_t)co * ((size_t)R * L.Cin) + (size_t)it.r * L.Cin + ci
                                              : ((size_t)co * L.Cin + ci) * R + it.r;
                        gfloat* outp = PLEAS_GLOBAL_W(L.out);
                        outp[o] = (L.flags & PLEAS_WGRAD_ACCUMULATE) ? outp[o] + acc[sm][sn][r] : acc[sm][sn][r];
                    } else
                        PLEAS_GLOBAL_W(L.slab)[((size_t)it.split * L.Cout + co) * ((size_t)R * L.Cin) + (size_t)it.r * L.Cin + ci] =
                            acc[sm][sn][r];
                }
            }
        }
}

// the split-bf16 forms: 16-byte-loadable operands only (1x1 stride-1 layers and stride-1 "same" k x k layers on images with
// HW % 4 == 0); the plan sends every other layer's items to the exact kernel
template <int TM, int TN>
__device__ __forceinline__ void wgrad_dispatch_split(const WgradLayerDev& L, const WgradItemDev& it, float* smem) {
    if (L.variant & 16) wgrad_tile<TM, TN, 4, 2, 1>(L, it, smem);
    else wgrad_tile<TM, TN, 4, 0, 1>(L, it, smem);
}

template <int TM, int TN>
__device__ __forceinline__ void wgrad_dispatch(const WgradLayerDev& L, const WgradItemDev& it, float* smem) {
    const bool xs = L.variant & 4, ys = L.variant & 8;
    if (L.variant & 32) {
        if (xs) wgrad_tile<TM, TN, 1, 3>(L, it, smem);
        else wgrad_tile<TM, TN, 4, 3>(L, it, smem);
    } else if (L.variant & 16) wgrad_tile<TM, TN, 4, 2>(L, it, smem);
    else if (!xs && !ys) wgrad_tile<TM, TN, 4, 0>(L, it, smem);
    else if (!xs && ys) wgrad_tile<TM, TN, 4, 1>(L, it, smem);
    else if (xs && !ys) wgrad_tile<TM, TN, 1, 0>(L, it, smem);
    else wgrad_tile<TM, TN, 1, 1>(L, it, smem);
}

#ifndef PLEAS_WGRAD_TIMELINE
#define PLEAS_WGRAD_TIMELINE 0   // study builds (tools/r04/timeline.sh): per work item, when and where it ran
#endif
#if PLEAS_WGRAD_TIMELINE
__device__ long long g_wgrad_timeline[16384][4];   // start, end (100 MHz wall clock), HW_ID | XCC_ID << 32, chunks * TM * TN
#endif

__global__ __launch_bounds__(cThreads, 2) void wgrad_batch_kernel(const WgradLayerDev* __restrict__ layers,
                                                               const WgradItemDev* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WgradItemDev it = items[blockIdx.x];
#if PLEAS_WGRAD_TIMELINE
    const long long t_start = __builtin_amdgcn_s_memrealtime();
    if (it.layer < 0) {
        if (threadIdx.x == 0 && blockIdx.x < 16384) {
            g_wgrad_timeline[blockIdx.x][0] = t_start; g_wgrad_timeline[blockIdx.x][1] = t_start;
            g_wgrad_timeline[blockIdx.x][2] = -1; g_wgrad_timeline[blockIdx.x][3] = 0;
        }
        return;
    }
#else
    if (it.layer < 0) return;   // padding of the XCD-aware item order
#endif
    const WgradLayerDev L = layers[it.layer];
    switch (L.variant & 3) {  // block-uniform
        case 0: wgrad_dispatch<128, 128>(L, it, smem); break;
        case 1: wgrad_dispatch<64, 128>(L, it, smem); break;
        case 2: wgrad_dispatch<128, 64>(L, it, smem); break;
        default: wgrad_dispatch<64, 64>(L, it, smem); break;
    }
#if PLEAS_WGRAD_TIMELINE
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        const long long hw = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(63508) << 32);
        g_wgrad_timeline[blockIdx.x][0] = t_start;
        g_wgrad_timeline[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
        g_wgrad_timeline[blockIdx.x][2] = hw;
        g_wgrad_timeline[blockIdx.x][3] = (long long)(it.c_end - it.c_begin) * ((L.variant & 1) ? 64 : 128) * ((L.variant & 2) ? 64 : 128);
    }
#endif
}

__global__ __launch_bounds__(cThreads, 3) void wgrad_batch_split_kernel(const WgradLayerDev* __restrict__ layers,
                                                                     const WgradItemDev* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WgradItemDev it = items[blockIdx.x];
    if (it.layer < 0) return;   // padding of the XCD-aware item order
    const WgradLayerDev L = layers[it.layer];
    switch (L.variant & 3) {  // block-uniform
        case 0: wgrad_dispatch_split<128, 128>(L, it, smem); break;
        case 1: wgrad_dispatch_split<64, 128>(L, it, smem); break;
        case 2: wgrad_dispatch_split<128, 64>(L, it, smem); break;
        default: wgrad_dispatch_split<64, 64>(L, it, smem); break;
    }
}

// per-update operand pointers -> device layer table (carried in kernel arguments)
constexpr int cPtrBatch = 150;
struct WgradPtrBatch {
    int base, count;
    const float* resid[cPtrBatch];
    const float* ip[cPtrBatch];
    float* out[cPtrBatch];
};
__global__ void wgrad_set_ptrs_kernel(WgradLayerDev* __restrict__ layers, const WgradPtrBatch b) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < b.count) {
        layers[b.base + t].resid = b.resid[t];
        layers[b.base + t].ip = b.ip[t];
        layers[b.base + t].out = b.out[t];
    }
}

// Slab reduce for layers with S > 1: out[(co*Cin+ci)*R + r] = sum_s slab[s][co][r*Cin+ci]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradLayerDev* __restrict__ layers,
                                                           const int* __restrict__ blk_layer,
                                                           const int* __restrict__ blk_begin) {
    const int li = blk_layer[blockIdx.x];
    const WgradLayerDev L = layers[li];
    const int R = L.KH * L.KW;
    const size_t per = (size_t)L.Cout * L.Cin * R;
    const size_t idx = (size_t)(blockIdx.x - blk_begin[blockIdx.x]) * blockDim.x + threadIdx.x;  // internal (co, r, ci)
    if (idx >= per) return;
    const size_t rowlen = (size_t)R * L.Cin;
    const int co = (int)(idx / rowlen);
    const int rem = (int)(idx - (size_t)co * rowlen);
    const int r = rem / L.Cin, ci = rem - r * L.Cin;
    float s = 0.f;
    for (int k = 0; k < L.S; ++k) s += L.slab[(size_t)k * per + idx];
    const size_t o = (L.flags & PLEAS_WGRAD_KPOS_MAJOR) ? idx : ((size_t)co * L.Cin + ci) * R + r;
    L.out[o] = (L.flags & PLEAS_WGRAD_ACCUMULATE) ? L.out[o] + s : s;
}

// ------------------------------------------------------------------------------------------------
// Fused regression target + residual + loss partials for one layer:
//   t      = coef(c) * ( [row1[c] >= 0] o1[n][row1[c]][p] + [row2[c] >= 0] o2[n][row2[c]][p] )
//   resid  = dscale * (out - t)            (written over `out` when resid == out)
//   part[blockIdx] = sum over this block of (out - t)^2
// Replaces pleas_merging.py:116-123 + :147 (index_select x4 + cat for the outputs), :282 and the first
// node of its backward.
constexpr int kTrBlocks = 2048;
template <int VEC>
__global__ __launch_bounds__(256) void target_residual_kernel(const float* __restrict__ out, const float* __restrict__ o1,
                                                              const float* __restrict__ o2, const int32_t* __restrict__ row1,
                                                              const int32_t* __restrict__ row2, int n_merged, int C, int Csrc,
                                                              uint32_t HW, uint32_t total_v, float dscale,
                                                              float* __restrict__ resid, float* __restrict__ part) {
    // flattened [N*C][HW] view, VEC elements per thread per step; a VEC-wide piece never crosses a row
    // (HW % VEC == 0), so one 32-bit division per piece locates its (n, c) row
    const uint32_t hw_v = HW / VEC;
    float s = 0.f;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < total_v; v += gridDim.x * blockDim.x) {
        const uint32_t row = v / hw_v, pv = v - row * hw_v;
        const uint32_t n = row / (uint32_t)C, c = row - n * (uint32_t)C;
        const int r1 = row1[c], r2 = row2[c];
        const float coef = (int)c < n_merged ? 0.5f : 1.0f;
        float a[VEC], b[VEC], o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = b[e] = 0.f;
        const size_t po = (size_t)pv * VEC;
        if constexpr (VEC == 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(out + (size_t)row * HW + po);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = q[e];
            if (r1 >= 0) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(o1 + ((size_t)n * Csrc + r1) * HW + po);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = t[e];
            }
            if (r2 >= 0) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(o2 + ((size_t)n * Csrc + r2) * HW + po);
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = t[e];
            }
        } else {
            o[0] = out[(size_t)row * HW + po];
            if (r1 >= 0) a[0] = o1[((size_t)n * Csrc + r1) * HW + po];
            if (r2 >= 0) b[0] = o2[((size_t)n * Csrc + r2) * HW + po];
        }
        float d[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            d[e] = o[e] - (a[e] + b[e]) * coef;
            s = fmaf(d[e], d[e], s);
            d[e] *= dscale;
        }
        if constexpr (VEC == 4) {
            f32x4 q = {d[0], d[1], d[2], d[3]};
            *reinterpret_cast<f32x4*>(resid + (size_t)row * HW + po) = q;
        } else {
            resid[(size_t)row * HW + po] = d[0];
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// loss[l] = scale[l] * sum(part[l][0..nparts[l]))  for all layers of an update in one launch
__global__ __launch_bounds__(64) void loss_final_kernel(const float* __restrict__ part, const int* __restrict__ nparts,
                                                        const float* __restrict__ scale, int stride,
                                                        float* __restrict__ loss) {
    const int l = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts[l]; i += 64) s += (double)part[(size_t)l * stride + i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) loss[l] = (float)(s * (double)scale[l]);
}

// ------------------------------------------------------------------------------------------------ host plan
static int g_wgrad_item_chunks = 112;

struct WgradPlan {
    std::vector<int64_t> key;
    std::vector<WgradLayerDev> layers;
    std::vector<WgradItemDev> items;
    std::vector<int> blk_layer, blk_begin;
    size_t off_layers = 0, off_items = 0, off_bl = 0, off_bb = 0, off_slabs = 0, total = 0, lds = 0;
    double flops = 0, bytes = 0;
    bool uploaded = false;
    int n_split = 0;         // items [0, n_split) run the split-bf16 kernel, the rest the exact one (pleas_arith)
    size_t lds_split = 0;
    double flops_split = 0;
};
// A few plans are kept (keyed by layer geometry + workspace address): a caller may alternate between grouped launches,
// e.g. the two gradient buckets of a data-parallel update, without rebuilding and re-uploading the tables each time.
constexpr int kWgradPlans = 4;
static WgradPlan g_wplans[kWgradPlans];
static unsigned g_wplan_turn = 0;
static std::mutex g_wplan_mu;

static size_t walign(size_t v) { return (v + 255) / 256 * 256; }

static int build_wgrad_plan(WgradPlan& P, const pleas_wgrad_layer* ly, int n) {
    P.layers.assign(n, WgradLayerDev());
    P.items.clear();
    P.blk_layer.clear();
    P.blk_begin.clear();
    P.flops = P.bytes = 0;
    P.lds = 0;
    std::vector<size_t> slab_off(n, 0);
    size_t slabs = 0;
    std::vector<XcdWork<WgradItemDev>> work, work_split;
    const bool split_on = arith_mode() == 1;
    P.n_split = 0;
    P.lds_split = 0;
    P.flops_split = 0;
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        const pleas_wgrad_layer& l = ly[i];
        if (l.N <= 0 || l.Cout <= 0 || l.Cin <= 0 || l.Hin <= 0 || l.Win <= 0 || l.KH <= 0 || l.KW <= 0 || l.stride <= 0 ||
            l.pad < 0)
            return bad_arg("wgrad: layer geometry");
        const int Hout = (l.Hin + 2 * l.pad - l.KH) / l.stride + 1, Wout = (l.Win + 2 * l.pad - l.KW) / l.stride + 1;
        if (Hout <= 0 || Wout <= 0) return bad_arg("wgrad: empty output");
        const int64_t HWo = (int64_t)Hout * Wout, HWi = (int64_t)l.Hin * l.Win, K = (int64_t)l.N * HWo;
        if (K >= (1ll << 31)) return bad_arg("wgrad: N*Hout*Wout must be < 2^31");
        WgradLayerDev& d = P.layers[i];
        d.Cout = l.Cout; d.Cin = l.Cin; d.Hin = l.Hin; d.Win = l.Win; d.Hout = Hout; d.Wout = Wout;
        d.KH = l.KH; d.KW = l.KW; d.stride = l.stride; d.pad = l.pad;
        d.HWo = (uint32_t)HWo;
        d.Ktot = (uint32_t)K;
        d.flags = l.flags;
        int R = l.KH * l.KW;
        // fewer than 16 input channels (the 3-channel stem): one tile row per (channel, tap) instead of one tap per item with
        // 3 of 64 rows alive -- the layer is planned as the 1x1 view [Cout][Cin * R] of its own gradient
        const bool virt = l.Cin < 16 && R > 1 && !(l.flags & PLEAS_WGRAD_KPOS_MAJOR) && (int64_t)l.Cin * R < (1 << 20);
        int Cin = l.Cin;
        d.rCin = l.Cin; d.rKW = l.KW; d.rR = R;
        if (virt) {
            Cin = l.Cin * R;
            R = 1;
            d.Cin = Cin; d.KH = d.KW = 1;
        }
        const int TM = l.Cout > 64 ? 128 : 64, TN = Cin > 64 ? 128 : 64;
        const bool ydirect = !virt && R == 1 && l.stride == 1 && l.pad == 0;
        const bool xvec = HWo % 4 == 0;
        // direct Y shares X's vector width, so it also needs HWi == HWo (true for 1x1 stride 1)
        d.variant = (TM == 64 ? 1 : 0) | (TN == 64 ? 2 : 0) | (xvec ? 0 : 4) | (ydirect ? 0 : 8) | (virt ? 32 : 0);
        // stride-1 "same" k x k layers on images with HW % 4 == 0: the shifted operand comes through aligned 16-byte loads
        // (PLEAS_WGRAD_VECSHIFT=0 keeps the one-pixel-per-load form for A/B)
        static const bool vecshift = !(std::getenv("PLEAS_WGRAD_VECSHIFT") && std::atoi(std::getenv("PLEAS_WGRAD_VECSHIFT")) == 0);
        const int64_t total = (int64_t)l.N * l.Cin * HWi;
        d.total = 0;
        if (vecshift && !virt && !ydirect && xvec && l.stride == 1 && l.KH == l.KW && 2 * l.pad == l.KH - 1 && HWi == HWo &&
            total < (1ll << 31)) {
            d.variant |= 16;
            d.total = (int)total;
        }
        const int nchunks = (int)ceil_div(K, cBK);
        const int S = (int)ceil_div(nchunks, g_wgrad_item_chunks);
        const int cps = (int)ceil_div(nchunks, S);
        d.S = S;
        if (S > 1) {
            slab_off[i] = slabs;
            slabs += (size_t)S * l.Cout * Cin * R;
            const int nb = (int)ceil_div((int64_t)l.Cout * Cin * R, 256);
            for (int b = 0; b < nb; ++b) {
                P.blk_layer.push_back(i);
                P.blk_begin.push_back(blk);
            }
            blk += nb;
        }
        // split-bf16 arithmetic: the 16-byte-loadable forms (X vector loads; Y direct or the aligned shifted form)
        const bool split = split_on && xvec && !virt && (ydirect || (d.variant & 16));
        if (split) {
            d.variant |= 64;
            P.lds_split = std::max(P.lds_split, (size_t)(TM + TN) * kSplitRow * sizeof(__bf16));
            // the epilogue stages [64][TN + 4] floats (one row half at a time for 128-row tiles) in the same memory
            P.lds_split = std::max(P.lds_split, (size_t)64 * (TN + 4) * sizeof(float));
            P.flops_split += 2.0 * l.Cout * (double)Cin * R * (double)K;
        } else {
            P.lds = std::max(P.lds, (size_t)2 * (TM + TN) * cLds * sizeof(float));
        }
        P.flops += 2.0 * l.Cout * (double)Cin * R * (double)K;
        P.bytes += ((double)l.Cout * K + (double)l.Cin * l.N * HWi) * sizeof(float);
        const int tms = (int)ceil_div(l.Cout, TM), tns = (int)ceil_div(Cin, TN);
        for (int s = 0; s < S; ++s)
            for (int r = 0; r < R; ++r)
                for (int tm = 0; tm < tms; ++tm)
                    for (int tn = 0; tn < tns; ++tn) {
                        XcdWork<WgradItemDev> w;
                        w.it = WgradItemDev{i, tm, tn, r, s, s * cps, std::min((s + 1) * cps, nchunks), 0};
                        w.w = (double)(w.it.c_end - w.it.c_begin) * TM * TN;
                        w.key = (int64_t)i * 65536 + s;   // every (tile, tap) of one K range reads the same residual / input rows
                        (split ? work_split : work).push_back(w);
                    }
    }
    P.items = xcd_order_items(work_split, WgradItemDev{-1, 0, 0, 0, 0, 0, 0, 0});
    P.n_split = (int)P.items.size();
    {
        std::vector<WgradItemDev> rest = xcd_order_items(work, WgradItemDev{-1, 0, 0, 0, 0, 0, 0, 0});
        P.items.insert(P.items.end(), rest.begin(), rest.end());
    }
    size_t off = 0;
    P.off_layers = off;
    off = walign(off + P.layers.size() * sizeof(WgradLayerDev));
    P.off_items = off;
    off = walign(off + P.items.size() * sizeof(WgradItemDev));
    P.off_bl = off;
    off = walign(off + P.blk_layer.size() * sizeof(int));
    P.off_bb = off;
    off = walign(off + P.blk_begin.size() * sizeof(int));
    P.off_slabs = off;
    P.total = off + slabs * sizeof(float);
    for (int i = 0; i < n; ++i) P.layers[i].slab = reinterpret_cast<float*>(slab_off[i]);
    P.uploaded = false;
    return PLEAS_OK;
}

// one side stream + two events per device for the exact grid of a launch under the split arithmetic (created once per process)
struct WgradSideStream {
    hipStream_t stream;
    hipEvent_t forked, joined;
    bool ok = false;
};
static WgradSideStream& wgrad_side_stream() {
    constexpr int kMaxDev = 16;
    static WgradSideStream sets[kMaxDev];
    static bool tried[kMaxDev] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
        static WgradSideStream none;
        return none;
    }
    WgradSideStream& s = sets[dev];
    if (!tried[dev]) {
        tried[dev] = true;
        s.ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&s.forked, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.joined, hipEventDisableTiming) == hipSuccess;
    }
    return s;
}

static std::vector<int64_t> wgrad_key(const pleas_wgrad_layer* ly, int n, const void* ws) {
    std::vector<int64_t> k;
    k.push_back(n);
    k.push_back((int64_t)(uintptr_t)ws);
    k.push_back(g_wgrad_item_chunks * 2 + arith_mode());      // plans differ between the arithmetics (pleas_arith)
    for (int i = 0; i < n; ++i) {
        const pleas_wgrad_layer& l = ly[i];
        for (int v : {l.N, l.Cout, l.Cin, l.Hin, l.Win, l.KH, l.KW, l.stride, l.pad, l.flags}) k.push_back(v);
    }
    return k;
}

}  // namespace pleas

using namespace pleas;

#if PLEAS_WGRAD_TIMELINE
extern "C" int pleas_wgrad_timeline_read(long long* out, int max_items) {      // study builds only
    const int n = std::min(max_items, 16384);
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgrad_timeline), (size_t)n * 4 * sizeof(long long)) == hipSuccess ? n : -1;
}
#endif

extern "C" void pleas_wgrad_tune(int item_chunks) {
    if (item_chunks > 0) g_wgrad_item_chunks = item_chunks;
}

extern "C" size_t pleas_wgrad_batch_ws_bytes(const pleas_wgrad_layer* layers, int n_layers) {
    if (!layers || n_layers <= 0) return 0;
    WgradPlan tmp;
    if (build_wgrad_plan(tmp, layers, n_layers) != PLEAS_OK) return 0;
    return tmp.total;
}

extern "C" int pleas_wgrad_batch(const pleas_wgrad_layer* layers, int n_layers, void* ws, size_t ws_bytes, int ws_fresh,
                                 void* stream_) {
    if (!layers || n_layers <= 0) return bad_arg("wgrad: empty layer list");
    for (int i = 0; i < n_layers; ++i) {
        if (!layers[i].resid || !layers[i].ip || !layers[i].grad) return bad_arg("wgrad: null pointer");
        if ((((uintptr_t)layers[i].resid | (uintptr_t)layers[i].ip) & 15) != 0) return bad_arg("wgrad: 16-byte alignment");
    }
    hipStream_t stream = (hipStream_t)stream_;
    std::lock_guard<std::mutex> lk(g_wplan_mu);
    std::vector<int64_t> key = wgrad_key(layers, n_layers, ws);
    WgradPlan* hit = nullptr;
    for (auto& cand : g_wplans)
        if (cand.key == key) hit = &cand;
    if (!hit) {
        hit = &g_wplans[g_wplan_turn++ % kWgradPlans];
        hit->key.clear();
        const int rc = build_wgrad_plan(*hit, layers, n_layers);
        if (rc != PLEAS_OK) return rc;
        hit->key.swap(key);
    }
    WgradPlan& P = *hit;
    if (ws_fresh) P.uploaded = false;  // caller says the tables inside ws are not (or no longer) there
    if (!ws || ws_bytes < P.total) {
        std::snprintf(g_last_error, sizeof(g_last_error), "wgrad workspace too small: need %zu bytes", P.total);
        P.key.clear();
        return PLEAS_ENOMEM;
    }
    char* base = (char*)ws;
    if (!P.uploaded) {
        for (auto& other : g_wplans)   // its tables go into `ws`: whatever another plan had there is gone
            if (&other != &P && other.key.size() > 1 && other.key[1] == (int64_t)(uintptr_t)ws) other.uploaded = false;
        float* slab0 = reinterpret_cast<float*>(base + P.off_slabs);
        std::vector<WgradLayerDev> abs_layers = P.layers;
        for (auto& d : abs_layers) d.slab = slab0 + reinterpret_cast<size_t>(d.slab);
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_layers, abs_layers.data(), abs_layers.size() * sizeof(WgradLayerDev),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_items, P.items.data(), P.items.size() * sizeof(WgradItemDev),
                                       hipMemcpyHostToDevice, stream));
        if (!P.blk_layer.empty()) {
            PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_bl, P.blk_layer.data(), P.blk_layer.size() * sizeof(int),
                                           hipMemcpyHostToDevice, stream));
            PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_bb, P.blk_begin.data(), P.blk_begin.size() * sizeof(int),
                                           hipMemcpyHostToDevice, stream));
        }
        PLEAS_HIP_CHECK(hipStreamSynchronize(stream));
        P.uploaded = true;
    }
    WgradLayerDev* dl = reinterpret_cast<WgradLayerDev*>(base + P.off_layers);
    for (int b0 = 0; b0 < n_layers; b0 += cPtrBatch) {
        WgradPtrBatch pb;
        pb.base = b0;
        pb.count = std::min(cPtrBatch, n_layers - b0);
        for (int t = 0; t < pb.count; ++t) {
            pb.resid[t] = layers[b0 + t].resid;
            pb.ip[t] = layers[b0 + t].ip;
            pb.out[t] = layers[b0 + t].grad;
        }
        hipLaunchKernelGGL(wgrad_set_ptrs_kernel, dim3(1), dim3(256), 0, stream, dl, pb);
        PLEAS_LAUNCH_CHECK("wgrad_set_ptrs_kernel");
    }
    {
        ProfScope prof(kProfConvWgrad, P.flops, P.bytes, stream);
        const WgradItemDev* its = reinterpret_cast<const WgradItemDev*>(base + P.off_items);
        const int n_exact = (int)P.items.size() - P.n_split;
        // Under the split arithmetic the launch is TWO grids (split-bf16 tiles; exact tiles of the forms without a split variant:
        // 7 x 7 images, strided layers, the stem -- few, long items).  Back to back on one stream the second would start when the
        // first one's last workgroup has ended (0.39 ms for two layers alone, profiles/r05_pmc_split_3x3s1_14.txt): the exact grid
        // goes to a side stream of the library instead (fork / join with events, as the forward's forms do).
        WgradSideStream& side = wgrad_side_stream();
        const bool fork = P.n_split > 0 && n_exact > 0 && side.ok;
        hipStream_t st_exact = stream;
        if (fork) {
            PLEAS_HIP_CHECK(hipEventRecord(side.forked, stream));
            PLEAS_HIP_CHECK(hipStreamWaitEvent(side.stream, side.forked, 0));
            st_exact = side.stream;
        }
        if (n_exact > 0) {
            hipLaunchKernelGGL(wgrad_batch_kernel, dim3((unsigned)n_exact), dim3(cThreads), P.lds, st_exact, dl, its + P.n_split);
            PLEAS_LAUNCH_CHECK("wgrad_batch_kernel");
        }
        if (P.n_split > 0) {
            hipLaunchKernelGGL(wgrad_batch_split_kernel, dim3((unsigned)P.n_split), dim3(cThreads), P.lds_split, stream, dl, its);
            PLEAS_LAUNCH_CHECK("wgrad_batch_split_kernel");
        }
        if (fork) {
            PLEAS_HIP_CHECK(hipEventRecord(side.joined, side.stream));
            PLEAS_HIP_CHECK(hipStreamWaitEvent(stream, side.joined, 0));
        }
        if (!P.blk_layer.empty()) {
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)P.blk_layer.size()), dim3(256), 0, stream, dl,
                               reinterpret_cast<const int*>(base + P.off_bl),
                               reinterpret_cast<const int*>(base + P.off_bb));
            PLEAS_LAUNCH_CHECK("wgrad_reduce_kernel");
        }
    }
    return PLEAS_OK;
}

extern "C" int pleas_target_residual(const float* out, const float* o1, const float* o2, const int32_t* row1,
                                     const int32_t* row2, int n_merged, int N, int C, int Csrc, int64_t HW, float dscale,
                                     float* resid, float* partials, int* n_partials, void* stream_) {
    if (!out || !o1 || !o2 || !row1 || !row2 || !resid || !partials || !n_partials) return bad_arg("target_residual: null");
    if (N <= 0 || C <= 0 || Csrc <= 0 || HW <= 0) return bad_arg("target_residual: shape");
    const int64_t total = (int64_t)N * C * HW;
    if (total >= (1ll << 32)) return bad_arg("target_residual: more than 2^32 elements");
    const bool vec = HW % 4 == 0 &&
                     ((((uintptr_t)out | (uintptr_t)o1 | (uintptr_t)o2 | (uintptr_t)resid) & 15) == 0);
    const int64_t pieces = vec ? total / 4 : total;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(pieces, 256), kTrBlocks));
    *n_partials = blocks;
    hipStream_t stream = (hipStream_t)stream_;
    ProfScope prof(kProfSqerr, 0.0, 4.0 * total * sizeof(float), stream);
    if (vec)
        hipLaunchKernelGGL((target_residual_kernel<4>), dim3(blocks), dim3(256), 0, stream, out, o1, o2, row1, row2, n_merged,
                           C, Csrc, (uint32_t)HW, (uint32_t)pieces, dscale, resid, partials);
    else
        hipLaunchKernelGGL((target_residual_kernel<1>), dim3(blocks), dim3(256), 0, stream, out, o1, o2, row1, row2, n_merged,
                           C, Csrc, (uint32_t)HW, (uint32_t)pieces, dscale, resid, partials);
    PLEAS_LAUNCH_CHECK("target_residual_kernel");
    return PLEAS_OK;
}

extern "C" int pleas_target_residual_max_partials(void) { return kTrBlocks; }

extern "C" int pleas_loss_final(const float* partials, const int* n_partials, const float* scale, int stride, int n_layers,
                                float* loss, void* stream_) {
    if (!partials || !n_partials || !scale || !loss || n_layers <= 0 || stride <= 0) return bad_arg("loss_final");
    hipLaunchKernelGGL(loss_final_kernel, dim3(n_layers), dim3(64), 0, (hipStream_t)stream_, partials, n_partials, scale,
                       stride, loss);
    PLEAS_LAUNCH_CHECK("loss_final_kernel");
    return PLEAS_OK;
}

// PLeaS layer fitting on gfx950: forward of ALL merged layers of one update + regression target + residual + loss as
// ONE PERSISTENT grid with producer / consumer waves ("streamed" form of pleas_fwd_batch, round 4).
//
// Replaces, per layer and per update (pleas/methods/pleas_merging.py): :281 out = layer(ip); :116-123, :147 op = block-merge
// of the two source layers' outputs; :282 loss = mean((out - op)^2); first node of :287 resid = 2 (out - op) / numel.
//
// Why this shape: the one-workgroup-per-item forms of conv_fwd.hip sat at 0.58 of the fp32 matrix
// peak because a work item is short (2-144 K chunks, median 8) and every item paid its own prologue (first loads) and
// epilogue (target gathers, residual stores) with the MFMA pipe idle.  Here
//   * ONE workgroup per CU (768 threads, all of the CU's LDS) lives for the whole launch and walks a host-made list of items;
//   * waves 0-7 (two per SIMD) ONLY read LDS and issue MFMAs: each owns 64 (32) output channels x 32 pixels of the item's
//     128 (64) x 128 tile; at an item's last chunk they drop their accumulators into an LDS tile and go on with the next item;
//   * waves 8-11 (one per SIMD) do ALL memory work: global loads of the chunk two steps ahead (across item boundaries), the
//     LDS writes of the chunk one step ahead, and the previous item's epilogue in slices (target gathers issued one step,
//     consumed the next) while the consumers are already multiplying the next item;
//   * one s_barrier per step ("tick") for all twelve waves; every LDS buffer is written in a tick in which nobody reads it.
// Formulation as before: implicit GEMM out[co][P] = sum_k W[co][k] U[k][P]; W tile rows k-contiguous in LDS (16-byte reads
// feeding four MFMA steps), U as an image [k][pixel] read by 4-byte LDS loads (one value per lane and MFMA).  Input forms:
//   VEC   1x1 stride 1, HW % 4 == 0, Cin % 32 == 0: 16-byte loads along the pixel axis;
//   FLAT  k x k stride-1 "same" convolutions with kernel-position-major weights: ONE LDS image per 32-channel block shared
//         by all taps (tap r reads it shifted by (kh - pad) W + (kw - pad)); the image travels in two halves in the load
//         slots of the two steps before its first use;
//   GEN   everything else (strided layers, the 3-channel stem, 7 x 7 images, Linear): one gathered element per load.
// Deterministic: every item owns four loss-partial slots (one per producer wave), summed in a fixed order.
#include <algorithm>
#include <mutex>
#include <queue>
#include <vector>

#include "common.hpp"

namespace pleas {
namespace fwds {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kBK = 32, kTN = 128;
constexpr int kCW = 8, kPW = 8, kThreads = 64 * (kCW + kPW);      // 8 consumer waves, 2 producer groups of 4 waves
constexpr int kAhead = 3;      // an element is requested three ticks before it is multiplied
constexpr int kLdsA = 36;      // W tile rows [TM][36] (k contiguous; 16-byte aligned, conflict-free ds_read_b128)
constexpr int kCt = 132;       // accumulator tile rows [TM][132]
constexpr int kLr1 = 132;      // image rows of the VEC / GEN forms: 128 pixels
constexpr int kLrMax = 208;    // widest image row (FLAT: 128 + 2 * halo data columns <= 207)
constexpr int oAs = 0;                                  // [2][128][36]
constexpr int oCt = oAs + 2 * 128 * kLdsA;              // [128][132]
constexpr int oB = oCt + 128 * kCt;                     // [2][32][kLrMax]
constexpr int oMaps = oB + 2 * 32 * kLrMax;             // [2][128][4]: row1, row2, bias, coef (0 = row past Cout)
constexpr int oRing = oMaps + 2 * 128 * 4;              // [4][64]: the records of the items in flight
constexpr int kLdsFloats = oRing + 4 * 64;
static_assert(kLdsFloats * 4 <= 163840, "one workgroup takes the CU's LDS");

// record of one work item (64 dwords; lane l of a loading wave holds dword l)
enum { rIP = 0, rW = 2, rBIAS = 4, rO1 = 6, rO2 = 8, rROW1 = 10, rROW2 = 12, rRESID = 14,      // pointers: patched per launch
       rLAYER = 16, rI0, rP0, rPSLOT, rNB, rNCH, rFLAGS, rLR, rCOUT, rCIN, rHIN, rWIN, rWOUT, rKH, rKW, rSTRIDE, rPAD, rCSRC,
       rHWO, rPTOT, rKD, rDSCALE, rHALO, rR, rCB, rHWI, rS, rNMERGED, rIMG0, rUsed };
static_assert(rUsed <= 64, "record is 64 dwords");
constexpr int fEND = 1, fTM64 = 2, fFormShift = 2, fSCALARA = 16, fKPOS = 32, fVECEPI = 64;
constexpr int FORM_VEC = 0, FORM_FLAT = 1, FORM_GEN = 2;
constexpr int kSliceCh = 2;    // output channels per thread and epilogue slice (x 4 pixels): kSliceCh * 8 channels of the tile per slice
constexpr int kParts = 8;      // loss partials per item: one per producer wave

#ifndef PLEAS_FWDS_STAMPS
#define PLEAS_FWDS_STAMPS 0      // experiments only (tools/hipbench): per-workgroup cycle accounting of the three roles
#endif
#if PLEAS_FWDS_STAMPS
__device__ long long g_fwds_stamps[1024][16];  // + 8..11: group 0's cycles in its phases (1) write, (2) consume, (3) request, (4) gather   // per workgroup: total, consumer busy, chunks, group 0 busy, its active ticks, group 1 busy, its active ticks, ticks
#define FWDS_T(var) const long long var = clock64()
#else
#define FWDS_T(var)
#endif

struct Meta {       // what the write step of a tick needs to know about the loads the previous tick issued
    int code;       // bits 0-1: weights (0 none, 1 16-byte rows, 2 scalar rows); 2-4: input (0 none, 1 VEC image, 2 GEN image,
                    // 3 / 4 first / second half of a FLAT image); 5: image buffer; 8..: weight rows per thread
    int lrspan;     // image row stride | FLAT: data columns << 16
    unsigned oka, okb;   // per thread: which of its weight rows / input values are real (the rest is written as zero)
};

// An item's record in ONE register per wave (lane l holds dword l): fields come out by v_readlane, no memory access.
__device__ __forceinline__ int rl(const int rec, const int k) { return __builtin_amdgcn_readlane(rec, k); }
__device__ __forceinline__ const char* rlp(const int rec, const int k) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane(rec, k), hi = (unsigned)__builtin_amdgcn_readlane(rec, k + 1);
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ const float* rec_ptr(const int* ring, int slot, int k) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(ring[slot * 64 + k]);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane(ring[slot * 64 + k + 1]);
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int rec_int(const int* ring, int slot, int k) { return __builtin_amdgcn_readfirstlane(ring[slot * 64 + k]); }

template <int MTM>
__device__ __forceinline__ void mfma_chunk(const float* a, const float* b, const int Lr, const bool ok, f32x16 (&acc)[2]) {
#pragma unroll
    for (int kk = 0; kk < kBK / 8; ++kk) {
        f32x4 fa[MTM];
#pragma unroll
        for (int s = 0; s < MTM; ++s) fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * kLdsA + kk * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float f = b[(kk * 8 + e) * Lr];
            f = ok ? f : 0.f;
#pragma unroll
            for (int s = 0; s < MTM; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][e], f, acc[s], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(kThreads, 1) void fwd_stream_kernel(const int* __restrict__ recs, const int2* __restrict__ wg,
                                                                 float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* ring = reinterpret_cast<int*>(smem + oRing);
    int* maps = reinterpret_cast<int*>(smem + oMaps);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int first = wg[blockIdx.x].x, count = wg[blockIdx.x].y;      // records of this workgroup, the END record included
    if (wave == kCW) ring[lane] = recs[(size_t)first * 64 + lane];
    __syncthreads();
    // Element x of the workgroup's stream (a K chunk of an item, or a bubble) is LOADED at tick x - 3, WRITTEN to LDS at tick
    // x - 1 and MULTIPLIED at tick x.  Ticks start at -3.

    if (wave < kCW) {
        // ============================================================ consumers: LDS reads + MFMAs only
        const int wm = wave >> 2, wn = wave & 3;
        f32x16 acc[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
        int jC = 0, tC = 0, ntC = 0, nbC = 0, nchC = 0, flagsC = 0;
        int Lr = kLr1, Wimg = 1, KW = 1, pad = 0, R = 1, form = 0, TM = 128, img0 = 0, jb = 0;
        unsigned tapok = ~0u;
        bool enter = true, finished = false;
#if PLEAS_FWDS_STAMPS
        long long c_busy = 0, c_chunks = 0, c_ticks = 0;
        const long long c_t0 = clock64();
#endif
        for (int g = -kAhead;; ++g) {
            FWDS_T(cs0);
            if (g >= 0) {
                if (enter) {
                    const int rc = ring[(jC & 3) * 64 + lane];      // the item's record: one LDS read per wave
                    flagsC = rl(rc, rFLAGS);
                    nbC = rl(rc, rNB);
                    nchC = rl(rc, rNCH);
                    ntC = nbC + nchC;
                    tC = 0;
                    enter = false;
                    if (!(flagsC & fEND)) {
                        form = (flagsC >> fFormShift) & 3;
                        TM = (flagsC & fTM64) ? 64 : 128;
                        Lr = rl(rc, rLR);
                        R = rl(rc, rR);
                        img0 = rl(rc, rIMG0);
                        const int q = wn * 32 + (lane & 31);
                        jb = rl(rc, rHALO) + q;
                        tapok = ~0u;
                        if (form == FORM_FLAT) {
                            Wimg = rl(rc, rWIN);
                            KW = rl(rc, rKW);
                            pad = rl(rc, rPAD);
                            const int Hin = rl(rc, rHIN);
                            const unsigned HW = (unsigned)rl(rc, rHWO), Ptot = (unsigned)rl(rc, rPTOT);
                            const unsigned P = (unsigned)rl(rc, rP0) + (unsigned)q;
                            unsigned mask = 0;
                            if (P < Ptot) {
                                const unsigned n = P / HW, pp = P - n * HW;
                                const int oh = (int)(pp / (unsigned)Wimg), ow = (int)(pp - (unsigned)oh * Wimg);
                                for (int r = 0; r < R; ++r) {
                                    const int kh = r / KW, kw = r - kh * KW;
                                    const int ih = oh + kh - pad, iw = ow + kw - pad;
                                    mask |= (unsigned)(ih >= 0 && ih < Hin && iw >= 0 && iw < Wimg) << r;
                                }
                            }
                            tapok = mask;
                        }
                    }
                }
                if (tC >= nbC && tC < ntC) {
                    const int c = tC - nbC;
                    int r = 0, delta = 0, img = img0 + c;
                    if (form == FORM_FLAT) {
                        const int cb = c / R;
                        r = c - cb * R;
                        const int kh = r / KW, kw = r - kh * KW;
                        delta = (kh - pad) * Wimg + (kw - pad);
                        img = img0 + cb;
                    }
                    const bool ok = (tapok >> r) & 1u;
                    const float* a = smem + oAs + (g & 1) * (128 * kLdsA) + (wm * (TM / 2) + (lane & 31)) * kLdsA + 4 * (lane >> 5);
                    const float* b = smem + oB + (img & 1) * (32 * kLrMax) + 4 * (lane >> 5) * Lr + jb + delta;
                    if (TM == 128) mfma_chunk<2>(a, b, Lr, ok, acc);
                    else mfma_chunk<1>(a, b, Lr, ok, acc);
                    if (c == nchC - 1) {      // the item is complete: accumulators -> LDS tile [co][pixel], start over
                        float* Ct = smem + oCt;
                        const int nS = TM / 64;
#pragma unroll
                        for (int s = 0; s < 2; ++s)
                            if (s < nS) {
#pragma unroll
                                for (int rr = 0; rr < 16; ++rr) {
                                    const int lco = wm * (TM / 2) + s * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
                                    Ct[lco * kCt + wn * 32 + (lane & 31)] = acc[s][rr];
                                    acc[s][rr] = 0.f;
                                }
                            }
                    }
                }
#if PLEAS_FWDS_STAMPS
                if (tC >= nbC && tC < ntC) ++c_chunks;
#endif
                if (++tC >= ntC) {
                    if (flagsC & fEND) finished = true;
                    else {
                        ++jC;
                        enter = true;
                    }
                }
            }
#if PLEAS_FWDS_STAMPS
            c_busy += clock64() - cs0;
            ++c_ticks;
#endif
            __syncthreads();
            if (finished) break;
        }
#if PLEAS_FWDS_STAMPS
        if (tid == 0 && blockIdx.x < 1024) {
            g_fwds_stamps[blockIdx.x][0] = clock64() - c_t0;
            g_fwds_stamps[blockIdx.x][1] = c_busy;
            g_fwds_stamps[blockIdx.x][2] = c_chunks;
            g_fwds_stamps[blockIdx.x][7] = c_ticks;
        }
#endif
        return;
    }

    // ================================================================ producers: every global access of the launch
    // TWO groups of four waves.  Group q handles the elements of parity q: at its active ticks (every other tick) it
    //   (1) waits for everything it requested two ticks ago and writes that element's tiles to LDS (+ the next record / the
    //       block maps it fetched), (2) finishes the epilogue slice whose targets it gathered two ticks ago,
    //   (3) requests the tiles of its next element, (4) gathers the targets of its next epilogue slice.
    // Waiting happens BEFORE anything new is requested, so "wait for all outstanding loads" (what the compiler emits when
    // paths request different numbers of loads) costs nothing, and every request has two ticks to complete.
    // Uniform per-item values are re-read from the record ring where they are used (a broadcast LDS read each).
    const int pt = tid - 64 * kCW, grp = pt >> 8, pq = pt & 255, pw = pq >> 6;
    // ---- the load cursor: element eL (parity grp) = tick tL of item jL
    int eL = grp, jL = 0, tL = grp, ntL = 0, nbL = 0, flagsL = 0;
    int recL = 0, recE = ring[lane];      // the records of the cursor's item and of the item in the epilogue: one register each
    bool need_enter = true, at_end = false;
    int g_stop = 0x7fffffff;
    uint32_t voff[4] = {0, 0, 0, 0};      // this thread's image columns [bytes] (GEN: [0] pixel offset, [1] [2] tap mask)
    unsigned vokL = 0;
    f32x4 ra[4], rb[4];
    Meta me = {0, 0, 0, 0};
    int fdn = 0, fd_slot = -1, mp_item = -1;      // mp_item: the item whose maps (mp1, mp2, mpb) are on their way
    int mp1 = -1, mp2 = -1;
    float mpb = 0.f;
    // ---- epilogue: item ep_cur's accumulators are in the LDS tile from tick ep_elast on
    int ep_cur = 0, ep_elast = rl(recE, rNB) + rl(recE, rNCH) - 1;
    int ep_s = -1;            // slice whose targets are in (ta, tb), -1: none
    bool ep_setup = false;
    float sq = 0.f;
    bool gin = false;
    uint32_t gbase = 0, rbase = 0;        // [bytes]
    f32x4 ta[kSliceCh], tb[kSliceCh];

    auto enter_item = [&]() __attribute__((always_inline)) {
        recL = ring[(jL & 3) * 64 + lane];
        flagsL = rl(recL, rFLAGS);
        nbL = rl(recL, rNB);
        ntL = nbL + rl(recL, rNCH);
        need_enter = false;
        const bool owner = tL == 0;      // the group that handles the item's first element also fetches the next record / the maps
        if (owner && jL + 1 < count) {
            if (pw == 0) fdn = recs[(size_t)(first + jL + 1) * 64 + lane];
            fd_slot = (jL + 1) & 3;
        }
        if (flagsL & fEND) {
            at_end = true;
            g_stop = eL - tL + ntL - 1;      // the END record's last element
            return;
        }
        const int form = (flagsL >> fFormShift) & 3;
        const int Cin = rl(recL, rCIN);
        const unsigned p0 = (unsigned)rl(recL, rP0), Ptot = (unsigned)rl(recL, rPTOT);
        const unsigned HWo = (unsigned)rl(recL, rHWO);
        if (form == FORM_VEC) {
            const unsigned P4 = p0 + 4u * (pq & 31);
            const bool ok = P4 < Ptot;
            const unsigned n = ok ? P4 / HWo : 0u, p = ok ? P4 - n * HWo : 0u;
            voff[0] = ok ? 4u * (n * (unsigned)Cin * HWo + p + (unsigned)(pq >> 5) * HWo) : 0u;
            vokL = ok ? 1u : 0u;
        } else if (form == FORM_FLAT) {
            const int halo = rl(recL, rHALO);
            const int span = kTN + 2 * halo;
            vokL = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int j = lane + 64 * m;
                const long long Pv = (long long)p0 - halo + j;
                const bool ok = j < span && Pv >= 0 && Pv < (long long)Ptot;
                const unsigned n = ok ? (unsigned)Pv / HWo : 0u, p = ok ? (unsigned)Pv - n * HWo : 0u;
                voff[m] = ok ? 4u * (n * (unsigned)Cin * HWo + p) : 0u;
                vokL |= (ok ? 1u : 0u) << m;
            }
        } else {
            const int stride = rl(recL, rSTRIDE), Wout = rl(recL, rWOUT), Hin = rl(recL, rHIN);
            const int Win = rl(recL, rWIN), pad = rl(recL, rPAD), KW = rl(recL, rKW);
            const int R = rl(recL, rR);
            const unsigned HWi = (unsigned)rl(recL, rHWI);
            const unsigned P = p0 + (unsigned)(pq & 127);
            const bool pin = P < Ptot;
            const unsigned pn = pin ? P / HWo : 0u, pp = pin ? P - pn * HWo : 0u;
            const int oh = (int)(pp / (unsigned)Wout), ow = (int)(pp - (unsigned)oh * Wout);
            const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
            voff[0] = (uint32_t)((int)(pn * (unsigned)Cin * HWi) + ih0 * Win + iw0);     // elements; may be "negative" (masked taps)
            unsigned long long tapmask = 0;
            for (int r = 0; r < R; ++r) {
                const int kh = r / KW, kw = r - kh * KW;
                const bool ok = pin && ih0 + kh >= 0 && ih0 + kh < Hin && iw0 + kw >= 0 && iw0 + kw < Win;
                tapmask |= (unsigned long long)ok << r;
            }
            voff[1] = (uint32_t)tapmask;
            voff[2] = (uint32_t)(tapmask >> 32);
        }
        if (owner) {      // block maps / bias of the tile's output channels: into LDS at this group's next active tick
            mp_item = jL;
            if (pq < ((flagsL & fTM64) ? 64 : 128)) {
                const unsigned Cout = (unsigned)rl(recL, rCOUT);
                const unsigned co = (unsigned)rl(recL, rI0) + (unsigned)pq;
                const unsigned cc = min(co, Cout - 1u);
                const int* row1 = reinterpret_cast<const int*>(rlp(recL, rROW1));
                const int* row2 = reinterpret_cast<const int*>(rlp(recL, rROW2));
                const float* bias = reinterpret_cast<const float*>(rlp(recL, rBIAS));
                mp1 = PLEAS_GLOBAL_I(row1)[cc];
                mp2 = PLEAS_GLOBAL_I(row2)[cc];
                mpb = bias ? PLEAS_GLOBAL(bias)[cc] : 0.f;
            }
        }
    };

    // loads of chunk c of the cursor's item into (ra, rb); me describes them for the write step
    auto issue_chunk = [&](const int c) __attribute__((always_inline)) {
        const int form = (flagsL >> fFormShift) & 3;
        const bool kpos = (flagsL & fKPOS) != 0;
        const int TM = (flagsL & fTM64) ? 64 : 128;
        const int R = rl(recL, rR), Cin = rl(recL, rCIN);
        const unsigned Kd = (unsigned)rl(recL, rKD), Cout = (unsigned)rl(recL, rCOUT);
        const unsigned i0 = (unsigned)rl(recL, rI0), HWi = (unsigned)rl(recL, rHWI);
        const int img0 = rl(recL, rIMG0);
        const char* wB = rlp(recL, rW);
        const char* ipB = rlp(recL, rIP);
        int cb = c, r = 0;
        if (kpos || form == FORM_FLAT) {
            cb = c / R;
            r = c - cb * R;
        }
        int code = 0;
        // ---- weights
        if (!(flagsL & fSCALARA)) {
            const unsigned k = (kpos || form == FORM_FLAT ? (unsigned)r * Cin + (unsigned)cb * kBK : (unsigned)c * kBK) + 4u * (pq & 7);
            const bool kina = k < Kd;
            const unsigned kc = kina ? k : 0u;
            unsigned oka = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < TM / 32) {
                    const unsigned gi = i0 + (unsigned)(pq >> 3) + 32u * q;
                    if (gi < Cout && kina) oka |= 1u << q;
                    ra[q] = *(const __attribute__((address_space(1))) f32x4*)(wB + 4u * (min(gi, Cout - 1u) * Kd + kc));
                }
            code = 1 | ((TM / 32) << 8);
            me.oka = oka;
        } else {
            const unsigned k = (unsigned)c * kBK + (unsigned)(pq & 31);
            const bool kina = k < Kd;
            const unsigned kc = kina ? k : 0u;
            unsigned oka = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < TM / 8) {
                    const unsigned gi = i0 + (unsigned)(pq >> 5) + 8u * q;
                    if (gi < Cout && kina) oka |= 1u << q;
                    ra[q >> 2][q & 3] = *(const __attribute__((address_space(1))) float*)(wB + 4u * (min(gi, Cout - 1u) * Kd + kc));
                }
            code = 2 | ((TM / 8) << 8);
            me.oka = oka;
        }
        // ---- input
        me.lrspan = rl(recL, rLR);
        if (form == FORM_VEC) {
            const char* base = ipB + (size_t)c * kBK * HWi * 4u;
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = *(const __attribute__((address_space(1))) f32x4*)(base + (voff[0] + 4u * (unsigned)(8 * i) * HWi));
            code |= (1 << 2) | (((img0 + c) & 1) << 5);
            me.okb = vokL;
        } else if (form == FORM_GEN) {
            const int khalf = __builtin_amdgcn_readfirstlane(pq >> 7);
            const int Win = rl(recL, rWIN), KW = rl(recL, rKW);
            const unsigned long long tapmask = ((unsigned long long)voff[2] << 32) | voff[1];
            unsigned okb = 0;
            if (R == 1 || kpos) {
                const int kh = r / KW, kw = r - kh * KW;
                const bool ok_tap = (tapmask >> r) & 1ull;
                const unsigned ch0 = (unsigned)cb * kBK + (unsigned)khalf * 16u;
                const int voffg = ok_tap ? (int)voff[0] + kh * Win + kw : 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const unsigned ch = ch0 + q;
                    const bool ok = ok_tap && ch < (unsigned)Cin;
                    okb |= (ok ? 1u : 0u) << q;
                    const unsigned lin = min(ch, (unsigned)Cin - 1u) * HWi;
                    rb[q >> 2][q & 3] = *(const __attribute__((address_space(1))) float*)(ipB + 4u * (unsigned)(voffg + (int)lin));
                }
            } else {
                const unsigned k = (unsigned)c * kBK + (unsigned)khalf * 16u;
                int ci = (int)(k / (unsigned)R);
                int rr = (int)(k - (unsigned)ci * R);
                int kh = rr / KW, kw = rr - kh * KW;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const bool ok = ((tapmask >> rr) & 1ull) && (k + q) < Kd;
                    okb |= (ok ? 1u : 0u) << q;
                    const int off = ok ? (int)voff[0] + (int)((unsigned)ci * HWi) + kh * Win + kw : 0;
                    rb[q >> 2][q & 3] = *(const __attribute__((address_space(1))) float*)(ipB + 4u * (unsigned)off);
                    ++rr;
                    ++kw;
                    const int cw = kw == KW;
                    kw = cw ? 0 : kw;
                    kh += cw;
                    const int cr = rr == R;
                    rr = cr ? 0 : rr;
                    kh = cr ? 0 : kh;
                    ci += cr;
                }
            }
            code |= (2 << 2) | (((img0 + c) & 1) << 5);
            me.okb = okb;
        } else {
            // FLAT: tap 0 carries the second half of its block's image, the last tap the first half of the next block's
            const int CB = rl(recL, rCB);
            const int half = r == 0 ? 1 : ((r == R - 1 && cb + 1 < CB) ? 0 : -1);
            if (half >= 0) {
                const int blk = half ? cb : cb + 1;
                const int span = kTN + 2 * rl(recL, rHALO);
                const int M = (span + 63) / 64;
                const char* base = ipB + (size_t)(blk * kBK + 8 * pw + 4 * half) * HWi * 4u;
#pragma unroll
                for (int kr = 0; kr < 4; ++kr)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        if (m < M) rb[kr][m] = *(const __attribute__((address_space(1))) float*)(base + (voff[m] + 4u * (unsigned)kr * HWi));
                code |= ((3 + half) << 2) | (((img0 + blk) & 1) << 5);
                me.lrspan |= span << 16;
                me.okb = vokL;
            }
        }
        me.code = code;
    };
    // the bubble right before a FLAT item's first chunk carries the first half of its first image
    auto issue_first_half = [&]() __attribute__((always_inline)) {
        const unsigned HWi = (unsigned)rl(recL, rHWI);
        const int span = kTN + 2 * rl(recL, rHALO);
        const int M = (span + 63) / 64;
        const char* base = rlp(recL, rIP) + (size_t)(8 * pw) * HWi * 4u;
#pragma unroll
        for (int kr = 0; kr < 4; ++kr)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (m < M) rb[kr][m] = *(const __attribute__((address_space(1))) float*)(base + (voff[m] + 4u * (unsigned)kr * HWi));
        me.code = (3 << 2) | ((rl(recL, rIMG0) & 1) << 5);
        me.lrspan = rl(recL, rLR) | (span << 16);
        me.okb = vokL;
    };

    auto write_set = [&](const int stage) __attribute__((always_inline)) {
        float* a = smem + oAs + stage * (128 * kLdsA);
        const int am = me.code & 3, bm = (me.code >> 2) & 7, buf = (me.code >> 5) & 1, rowsA = me.code >> 8;
        if (am == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < rowsA) {
                    const bool ok = (me.oka >> q) & 1u;
                    const f32x4 v = {ok ? ra[q][0] : 0.f, ok ? ra[q][1] : 0.f, ok ? ra[q][2] : 0.f, ok ? ra[q][3] : 0.f};
                    *reinterpret_cast<f32x4*>(a + ((pq >> 3) + 32 * q) * kLdsA + 4 * (pq & 7)) = v;
                }
        } else if (am == 2) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < rowsA) a[((pq >> 5) + 8 * q) * kLdsA + (pq & 31)] = ((me.oka >> q) & 1u) ? ra[q >> 2][q & 3] : 0.f;
        }
        float* b = smem + oB + buf * (32 * kLrMax);
        if (bm == 1) {
            const bool ok = me.okb & 1u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = {ok ? rb[i][0] : 0.f, ok ? rb[i][1] : 0.f, ok ? rb[i][2] : 0.f, ok ? rb[i][3] : 0.f};
                *reinterpret_cast<f32x4*>(b + ((pq >> 5) + 8 * i) * kLr1 + 4 * (pq & 31)) = v;
            }
        } else if (bm == 2) {
            const int khalf = pq >> 7;
#pragma unroll
            for (int q = 0; q < 16; ++q) b[(khalf * 16 + q) * kLr1 + (pq & 127)] = ((me.okb >> q) & 1u) ? rb[q >> 2][q & 3] : 0.f;
        } else if (bm >= 3) {
            const int half = bm - 3, Lr = me.lrspan & 0xffff, span = me.lrspan >> 16;
#pragma unroll
            for (int kr = 0; kr < 4; ++kr)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (lane + 64 * m < span) b[(8 * pw + 4 * half + kr) * Lr + lane + 64 * m] = ((me.okb >> m) & 1u) ? rb[kr][m] : 0.f;
        }
    };

    // ---- epilogue slices: this thread owns channels (pq / 32) + 8 j of the tile and 4 consecutive pixels
    auto ep_gather = [&](const int s) __attribute__((always_inline)) {
        if (!(rl(recE, rFLAGS) & fVECEPI)) return;
        const char* o1 = rlp(recE, rO1);
        const char* o2 = rlp(recE, rO2);
        const unsigned hw4 = 4u * (unsigned)rl(recE, rHWO);
        const int* mp = maps + (ep_cur & 1) * (128 * 4);
#pragma unroll
        for (int u = 0; u < kSliceCh; ++u) {
            const int lco = (pq >> 5) + 8 * (kSliceCh * s + u);
            const i32x4 mm = *reinterpret_cast<const i32x4*>(mp + lco * 4);
            const uint32_t oa = (gin && mm[0] >= 0) ? gbase + (unsigned)mm[0] * hw4 : 0u;
            const uint32_t ob = (gin && mm[1] >= 0) ? gbase + (unsigned)mm[1] * hw4 : 0u;
            ta[u] = *(const __attribute__((address_space(1))) f32x4*)(o1 + oa);
            tb[u] = *(const __attribute__((address_space(1))) f32x4*)(o2 + ob);
        }
    };
    auto ep_consume = [&](const int s) __attribute__((always_inline)) {
        const int flags = rl(recE, rFLAGS);
        const float dscale = __int_as_float(rl(recE, rDSCALE));
        const unsigned HWo = (unsigned)rl(recE, rHWO);
        char* resid = const_cast<char*>(rlp(recE, rRESID));
        const int* mp = maps + (ep_cur & 1) * (128 * 4);
        const float* Ct = smem + oCt;
        const int pg = (pq & 31) * 4;
#pragma unroll
        for (int u = 0; u < kSliceCh; ++u) {
            const int j = kSliceCh * s + u;
            const int lco = (pq >> 5) + 8 * j;
            const i32x4 mm = *reinterpret_cast<const i32x4*>(mp + lco * 4);
            const float bias = __int_as_float(mm[2]), coef = __int_as_float(mm[3]);
            const bool live = gin && coef != 0.f;
            const f32x4 o = *reinterpret_cast<const f32x4*>(Ct + lco * kCt + pg);
            if (flags & fVECEPI) {
                // target = (o1 * [present] + o2 * [present]) * coef, coef in {0.5, 1}: folding coef into the two factors is exact
                const float ca = mm[0] >= 0 ? coef : 0.f, cb = mm[1] >= 0 ? coef : 0.f;
                f32x4 d;
                float s4 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dd = (o[e] + bias) - fmaf(tb[u][e], cb, ta[u][e] * ca);
                    s4 = fmaf(dd, dd, s4);
                    d[e] = dscale * dd;
                }
                sq += live ? s4 : 0.f;
                if (live) *(__attribute__((address_space(1))) f32x4*)(resid + (rbase + (unsigned)(8 * j) * (4u * HWo))) = d;
            } else if (live) {
                const char* o1 = rlp(recE, rO1);
                const char* o2 = rlp(recE, rO2);
                const unsigned Ptot = (unsigned)rl(recE, rPTOT), Csrc = (unsigned)rl(recE, rCSRC);
                const unsigned Cout = (unsigned)rl(recE, rCOUT);
                const unsigned co = (unsigned)rl(recE, rI0) + (unsigned)lco;
                const unsigned Pg = (unsigned)rl(recE, rP0) + 4u * (pq & 31);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned Pe = Pg + e;
                    if (Pe < Ptot) {
                        const unsigned n = Pe / HWo, p = Pe - n * HWo;
                        float av = 0.f, bv = 0.f;
                        if (mm[0] >= 0) av = *(const __attribute__((address_space(1))) float*)(o1 + 4u * ((n * Csrc + (unsigned)mm[0]) * HWo + p));
                        if (mm[1] >= 0) bv = *(const __attribute__((address_space(1))) float*)(o2 + 4u * ((n * Csrc + (unsigned)mm[1]) * HWo + p));
                        const float dd = (o[e] + bias) - (av + bv) * coef;
                        sq = fmaf(dd, dd, sq);
                        *(__attribute__((address_space(1))) float*)(resid + 4u * ((n * Cout + co) * HWo + p)) = dscale * dd;
                    }
                }
            }
        }
    };

#if PLEAS_FWDS_STAMPS
    long long p_busy = 0, p_active = 0, p_ph[5] = {0, 0, 0, 0, 0};
#endif
    for (int g = -kAhead;; ++g) {
        if (((g + kAhead) & 1) == grp) {
            FWDS_T(ps0);
#if PLEAS_FWDS_STAMPS
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): how long do the requests of two ticks ago still take?
            const long long psw = clock64();
#endif
            // ---- (1) LDS writes: the element requested two ticks ago (element g + 1), the next record, the block maps
            if (fd_slot >= 0) {
                if (pw == 0) ring[fd_slot * 64 + lane] = fdn;
                fd_slot = -1;
            }
            if (mp_item >= 0) {
                const int rm = ring[(mp_item & 3) * 64 + lane];
                if (pq < ((rl(rm, rFLAGS) & fTM64) ? 64 : 128)) {
                    const unsigned co = (unsigned)rl(rm, rI0) + (unsigned)pq;
                    // coef: 0.5 merged unit, 1 separate unit, 0 = row past Cout (never live)
                    const float coef = co < (unsigned)rl(rm, rCOUT) ? (co < (unsigned)rl(rm, rNMERGED) ? 0.5f : 1.0f) : 0.f;
                    const i32x4 v = {coef != 0.f ? mp1 : -1, coef != 0.f ? mp2 : -1, __float_as_int(mpb), __float_as_int(coef)};
                    *reinterpret_cast<i32x4*>(maps + (mp_item & 1) * (128 * 4) + pq * 4) = v;
                }
                mp_item = -1;
            }
            write_set((g + 1) & 1);
            FWDS_T(ps1);
            // ---- (2) epilogue: finish the slice gathered two ticks ago
            if (ep_s >= 0) {
                ep_consume(ep_s);
                if (ep_s + 2 >= rl(recE, rS)) {      // this group's last slice of the item
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
                    if (lane == 0) partials[rl(recE, rPSLOT) + 4 * grp + pw] = sq;
                    ++ep_cur;      // its record is in the ring: the consumers are multiplying it (or it is the END record)
                    recE = ring[(ep_cur & 3) * 64 + lane];
                    ep_elast += rl(recE, rNB) + rl(recE, rNCH);
                    ep_setup = false;
                }
                ep_s = -1;
            }
            FWDS_T(ps2);
            // ---- (3) requests for element g + 3
            me.code = 0;
            if (!at_end) {
                if (need_enter) enter_item();
                if (!at_end) {
                    if (tL >= nbL) issue_chunk(tL - nbL);
                    else if (((flagsL >> fFormShift) & 3) == FORM_FLAT && tL == nbL - 1) issue_first_half();
                    tL += 2;
                    eL += 2;
                    if (tL >= ntL) {      // ticks of an item >= 3: at most one boundary per step
                        tL -= ntL;
                        ++jL;
                        need_enter = true;
                    }
                }
            }
            FWDS_T(ps3);
            // ---- (4) epilogue: gather the targets of this group's next slice
            {
                const int s = g - ep_elast;
                if (s >= 0 && !(rl(recE, rFLAGS) & fEND) && s < rl(recE, rS)) {
                    if (!ep_setup) {
                        ep_setup = true;
                        sq = 0.f;
                        const unsigned HWo = (unsigned)rl(recE, rHWO), Ptot = (unsigned)rl(recE, rPTOT);
                        const unsigned Pg = (unsigned)rl(recE, rP0) + 4u * (pq & 31);
                        gin = Pg < Ptot;
                        const unsigned gn = gin ? Pg / HWo : 0u, gp = gin ? Pg - gn * HWo : 0u;
                        gbase = 4u * (gn * (unsigned)rl(recE, rCSRC) * HWo + gp);
                        rbase = 4u * ((gn * (unsigned)rl(recE, rCOUT) + (unsigned)rl(recE, rI0) + (unsigned)(pq >> 5)) * HWo + gp);
                    }
                    ep_gather(s);
                    ep_s = s;
                }
            }
#if PLEAS_FWDS_STAMPS
            {
                const long long pe = clock64();
                p_busy += pe - ps0;
                p_ph[4] += psw - ps0; p_ph[0] += ps1 - ps0; p_ph[1] += ps2 - ps1; p_ph[2] += ps3 - ps2; p_ph[3] += pe - ps3;
                ++p_active;
            }
#endif
        }
        __syncthreads();
        if (g >= g_stop) break;
    }
#if PLEAS_FWDS_STAMPS
    if (pq == 0 && blockIdx.x < 1024) {
        g_fwds_stamps[blockIdx.x][3 + 2 * grp] = p_busy;
        g_fwds_stamps[blockIdx.x][4 + 2 * grp] = p_active;
        if (grp == 0)
            for (int k = 0; k < 5; ++k) g_fwds_stamps[blockIdx.x][8 + k] = p_ph[k];
    }
#endif
}

// every record's eight pointers from its layer's row of the pointer table (which the host rewrites per launch)
__global__ void fwd_stream_patch_kernel(int* __restrict__ recs, const int* __restrict__ layer_ptrs, int n_recs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_recs * 16) return;
    const int rec = t >> 4, d = t & 15;
    recs[(size_t)rec * 64 + d] = layer_ptrs[recs[(size_t)rec * 64 + rLAYER] * 16 + d];
}

struct LossDev {
    int begin, count;
    float scale;
    int pad;
};
__global__ __launch_bounds__(64) void fwd_stream_loss_kernel(const float* __restrict__ partials, const LossDev* __restrict__ ld,
                                                             float* __restrict__ loss) {
    const LossDev d = ld[blockIdx.x];
    double s = 0.0;
    for (int i = threadIdx.x; i < d.count; i += 64) s += (double)partials[d.begin + i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) loss[blockIdx.x] = (float)(s * (double)d.scale);
}

constexpr int kPtrBatch = 56;
struct PtrBatch {
    int base, count;
    const void* p[kPtrBatch][8];
};
__global__ void fwd_stream_set_ptrs_kernel(int* __restrict__ layer_ptrs, const PtrBatch b) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < b.count * 8) {
        const unsigned long long v = (unsigned long long)b.p[t >> 3][t & 7];
        layer_ptrs[(b.base + (t >> 3)) * 16 + 2 * (t & 7)] = (int)(unsigned)(v & 0xffffffffull);
        layer_ptrs[(b.base + (t >> 3)) * 16 + 2 * (t & 7) + 1] = (int)(unsigned)(v >> 32);
    }
}

struct Plan {
    std::vector<int64_t> key;
    std::vector<int> recs;            // 64 dwords per record, workgroup by workgroup
    std::vector<int2> wg;             // (first record, records) per workgroup
    std::vector<LossDev> loss;
    size_t off_recs = 0, off_wg = 0, off_ptrs = 0, off_loss = 0, off_parts = 0, total = 0;
    int n_wg = 0, n_parts = 0, n_items = 0;
    double flops = 0, bytes = 0;
    bool uploaded = false;
    int forms[3] = {0, 0, 0}, bubbles = 0, ticks_max = 0, ticks_min = 0;
};
static PlanCache<Plan, 1> g_plans;
static std::mutex g_mu;
static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

static int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;      // no device (host-only plan queries): the MI355X's count
        (void)hipGetLastError();
    }
    return cus;
}

struct HostItem {
    int layer, i0, p0, slot, nch, flags, S;
    double cost;
};

// Items of all layers dealt to the workgroups: longest-processing-time first (the sort is stable, so equal items of a layer
// go to neighbouring workgroups at the same position of their lists: at any time the chip works on few layers, whose
// weights stay in the L2s), then per workgroup the bubbles that keep an item's epilogue clear of the next item's.
static int build_plan(Plan& P, const pleas_fwd_layer* ly, int n, int n_wg) {
    P.loss.assign(n, LossDev());
    P.flops = P.bytes = 0;
    P.forms[0] = P.forms[1] = P.forms[2] = 0;
    std::vector<HostItem> items;
    std::vector<std::vector<int>> lay(n);     // per layer: the record's constant dwords
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
        const int64_t in_elems = (int64_t)l.N * l.Cin * l.Hin * l.Win;
        if (Ptot >= (1ll << 31) || (int64_t)l.Cout * Kd >= (1ll << 32) || in_elems >= (1ll << 31))
            return bad_arg("conv_fwd: tensor too large");
        const int R = l.KH * l.KW;
        const bool kpos = (l.flags & PLEAS_FWD_KPOS_MAJOR) != 0;
        if (kpos && l.Cin % kBK != 0) return bad_arg("conv_fwd: kernel-position-major weights need Cin % 32 == 0");
        const int TM = (l.Cout > 64 && !(R == 1 && l.stride == 1 && Kd < 256 && l.Cin % kBK == 0)) ? 128 : 64;
        const bool same = l.stride == 1 && l.KH == l.KW && (l.KH & 1) && l.pad == (l.KH - 1) / 2;
        const int halo = l.pad * (l.Win + 1);
        int form = FORM_GEN, Lr = kLr1;
        if (R == 1 && l.stride == 1 && l.pad == 0 && HWo % 4 == 0 && l.Cin % kBK == 0) form = FORM_VEC;
        else if (same && R > 1 && R <= 32 && l.Cin % kBK == 0 && kpos && kTN + 2 * halo + 1 <= kLrMax) {
            form = FORM_FLAT;
            Lr = (kTN + 2 * halo + 1 + 3) / 4 * 4;
        }
        int flags = (TM == 64 ? fTM64 : 0) | (form << fFormShift) | (Kd % 4 == 0 ? 0 : fSCALARA) | (kpos ? fKPOS : 0) |
                    (HWo % 4 == 0 ? fVECEPI : 0);
        const int nch = (int)ceil_div(Kd, kBK);
        std::vector<int>& c = lay[i];
        c.assign(64, 0);
        c[rLAYER] = i; c[rFLAGS] = flags; c[rLR] = Lr; c[rCOUT] = l.Cout; c[rCIN] = l.Cin; c[rHIN] = l.Hin; c[rWIN] = l.Win;
        c[rWOUT] = Wout; c[rKH] = l.KH; c[rKW] = l.KW; c[rSTRIDE] = l.stride; c[rPAD] = l.pad; c[rCSRC] = l.Csrc;
        c[rHWO] = (int)HWo; c[rPTOT] = (int)Ptot; c[rKD] = (int)Kd; std::memcpy(&c[rDSCALE], &l.dscale, 4);
        c[rHALO] = form == FORM_FLAT ? halo : 0; c[rR] = R; c[rCB] = l.Cin / kBK; c[rHWI] = l.Hin * l.Win; c[rS] = TM / (8 * kSliceCh);
        c[rNMERGED] = l.n_merged; c[rNCH] = nch;
        const int tms = (int)ceil_div(l.Cout, TM), tps = (int)ceil_div(Ptot, kTN);
        int slot = 0;
        for (int tp = 0; tp < tps; ++tp)
            for (int tm = 0; tm < tms; ++tm) {
                HostItem it;
                it.layer = i; it.i0 = tm * TM; it.p0 = tp * kTN; it.slot = parts + kParts * slot; it.nch = nch; it.flags = flags;
                it.S = TM / (8 * kSliceCh);
                // relative duration: the chunks' MFMA time, or -- short K -- the epilogue's memory time (3 x tile bytes)
                const double mfma = (double)nch * (TM / 64) * 0.5, mem = (TM / 64) * 1.6 + 0.5;
                it.cost = std::max(mfma, mem) + 0.3;
                items.push_back(it);
                ++slot;
            }
        P.loss[i] = LossDev{parts, kParts * slot, l.loss_scale, 0};
        parts += kParts * slot;
        P.forms[form] += slot;
        P.flops += 2.0 * l.Cout * (double)Kd * (double)Ptot;
        P.bytes += ((double)l.Cin * l.N * l.Hin * l.Win + 3.0 * l.Cout * (double)Ptot) * sizeof(float);
    }
    P.n_items = (int)items.size();
    P.n_parts = parts;
    // ---- longest-processing-time first onto n_wg lists
    std::vector<int> order(items.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return items[a].cost > items[b].cost; });
    typedef std::pair<double, int> Load;
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int w = 0; w < n_wg; ++w) heap.push(Load(0.0, w));
    std::vector<std::vector<int>> lists(n_wg);
    for (int idx : order) {
        Load top = heap.top();
        heap.pop();
        lists[top.second].push_back(idx);
        top.first += items[idx].cost;
        heap.push(top);
    }
    // ---- records
    P.recs.clear();
    P.wg.assign(n_wg, int2{0, 0});
    P.bubbles = 0;
    P.ticks_max = 0;
    P.ticks_min = 1 << 30;
    for (int w = 0; w < n_wg; ++w) {
        P.wg[w].x = (int)(P.recs.size() / 64);
        int prevS = 0, ticks = 0, img = 0;
        for (int idx : lists[w]) {
            const HostItem& it = items[idx];
            std::vector<int> rec = lay[it.layer];
            const int form = (it.flags >> fFormShift) & 3;
            // ticks of an item >= 3 (record pipeline) and >= the epilogue slices of its predecessor + 2: slice s of an item
            // whose accumulators reach the LDS tile at tick T is gathered at tick T + s and finished at T + s + 2, and the
            // next item's accumulators must not arrive before that
            const int nb = std::max(std::max(form == FORM_FLAT ? 1 : 0, 3 - it.nch), prevS + 2 - it.nch);
            rec[rI0] = it.i0; rec[rP0] = it.p0; rec[rPSLOT] = it.slot; rec[rNB] = nb; rec[rIMG0] = img;
            img += form == FORM_FLAT ? rec[rCB] : it.nch;      // LDS images the item reads (one per chunk / per channel block)
            P.recs.insert(P.recs.end(), rec.begin(), rec.end());
            P.bubbles += nb;
            ticks += nb + it.nch;
            prevS = it.S;
        }
        std::vector<int> end(64, 0);
        end[rFLAGS] = fEND;
        end[rNB] = std::max(3, prevS + 2);
        P.recs.insert(P.recs.end(), end.begin(), end.end());
        ticks += end[rNB];
        P.wg[w].y = (int)lists[w].size() + 1;
        P.ticks_max = std::max(P.ticks_max, ticks);
        P.ticks_min = std::min(P.ticks_min, ticks);
    }
    P.n_wg = n_wg;
    size_t off = 0;
    P.off_recs = off;
    off = align256(off + P.recs.size() * sizeof(int));
    P.off_wg = off;
    off = align256(off + P.wg.size() * sizeof(int2));
    P.off_ptrs = off;
    off = align256(off + (size_t)n * 16 * sizeof(int));
    P.off_loss = off;
    off = align256(off + P.loss.size() * sizeof(LossDev));
    P.off_parts = off;
    P.total = off + (size_t)parts * sizeof(float);
    P.uploaded = false;
    return PLEAS_OK;
}

#if PLEAS_FWDS_STAMPS
extern "C" int pleas_fwds_stamps_read(long long* out, int n_wg) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwds_stamps), sizeof(long long) * 16 * (size_t)std::min(n_wg, 1024)) == hipSuccess ? 0 : 1;
}
#endif

size_t stream_ws_bytes(const pleas_fwd_layer* layers, int n_layers) {
    Plan tmp;
    if (build_plan(tmp, layers, n_layers, device_cus()) != PLEAS_OK) return 0;
    return tmp.total;
}

int stream_plan_info(const pleas_fwd_layer* layers, int n_layers, int n_wg, int* info) {
    Plan tmp;
    const int rc = build_plan(tmp, layers, n_layers, n_wg > 0 ? n_wg : device_cus());
    if (rc != PLEAS_OK) return rc;
    info[0] = tmp.n_items; info[1] = tmp.n_wg; info[2] = tmp.forms[0]; info[3] = tmp.forms[1]; info[4] = tmp.forms[2];
    info[5] = tmp.bubbles; info[6] = tmp.ticks_min; info[7] = tmp.ticks_max;
    return PLEAS_OK;
}

int stream_launch(const pleas_fwd_layer* layers, int n_layers, float* loss, void* ws, size_t ws_bytes, int ws_fresh,
                  hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<int64_t> key;
    key.push_back(n_layers);
    key.push_back((int64_t)(uintptr_t)ws);
    for (int i = 0; i < n_layers; ++i) {
        const pleas_fwd_layer& l = layers[i];
        for (int v : {l.N, l.Cout, l.Cin, l.Hin, l.Win, l.KH, l.KW, l.stride, l.pad, l.Csrc, l.n_merged, l.flags}) key.push_back(v);
        int32_t bits[2];
        std::memcpy(&bits[0], &l.dscale, 4);
        std::memcpy(&bits[1], &l.loss_scale, 4);
        key.push_back(bits[0]);
        key.push_back(bits[1]);
    }
    Plan* hit = g_plans.find(key);
    if (!hit) {
        hit = &g_plans.take();
        const int rc = build_plan(*hit, layers, n_layers, device_cus());
        if (rc != PLEAS_OK) return rc;
        hit->key.swap(key);
    }
    Plan& P = *hit;
    if (ws_fresh) P.uploaded = false;
    if (!ws || ws_bytes < P.total) {
        std::snprintf(g_last_error, sizeof(g_last_error), "conv_fwd workspace too small: need %zu bytes", P.total);
        P.key.clear();
        return PLEAS_ENOMEM;
    }
    char* base = (char*)ws;
    if (!P.uploaded) {
        g_plans.claims_workspace(P);
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_recs, P.recs.data(), P.recs.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_wg, P.wg.data(), P.wg.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_loss, P.loss.data(), P.loss.size() * sizeof(LossDev), hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipStreamSynchronize(stream));
        static bool lds_set = false;
        if (!lds_set) {
            PLEAS_HIP_CHECK(hipFuncSetAttribute((const void*)fwd_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsFloats * 4));
            lds_set = true;
        }
        P.uploaded = true;
    }
    int* recs = reinterpret_cast<int*>(base + P.off_recs);
    int* ptrs = reinterpret_cast<int*>(base + P.off_ptrs);
    for (int b0 = 0; b0 < n_layers; b0 += kPtrBatch) {
        PtrBatch pb;
        pb.base = b0;
        pb.count = std::min(kPtrBatch, n_layers - b0);
        for (int t = 0; t < pb.count; ++t) {
            const pleas_fwd_layer& l = layers[b0 + t];
            const void* v[8] = {l.ip, l.w, l.bias, l.o1, l.o2, l.row1, l.row2, l.resid};
            for (int k = 0; k < 8; ++k) pb.p[t][k] = v[k];
        }
        hipLaunchKernelGGL(fwd_stream_set_ptrs_kernel, dim3((pb.count * 8 + 63) / 64), dim3(64), 0, stream, ptrs, pb);
        PLEAS_LAUNCH_CHECK("fwd_stream_set_ptrs_kernel");
    }
    const int n_recs = (int)(P.recs.size() / 64);
    hipLaunchKernelGGL(fwd_stream_patch_kernel, dim3((n_recs * 16 + 255) / 256), dim3(256), 0, stream, recs, ptrs, n_recs);
    PLEAS_LAUNCH_CHECK("fwd_stream_patch_kernel");
    float* parts = reinterpret_cast<float*>(base + P.off_parts);
    {
        ProfScope prof(kProfConvFwd, P.flops, P.bytes, stream);
        hipLaunchKernelGGL(fwd_stream_kernel, dim3(P.n_wg), dim3(kThreads), kLdsFloats * 4, stream, recs,
                           reinterpret_cast<const int2*>(base + P.off_wg), parts);
    }
    PLEAS_LAUNCH_CHECK("fwd_stream_kernel");
    hipLaunchKernelGGL(fwd_stream_loss_kernel, dim3(n_layers), dim3(64), 0, stream, parts,
                       reinterpret_cast<const LossDev*>(base + P.off_loss), loss);
    PLEAS_LAUNCH_CHECK("fwd_stream_loss_kernel");
    return PLEAS_OK;
}

}  // namespace fwds
}  // namespace pleas

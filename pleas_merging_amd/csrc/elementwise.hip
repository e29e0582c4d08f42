// HBM-bound streaming kernels of the merging path (gfx950): block gather/average, fused
// masked Adam, squared-error reduction.  All are coalesced 16-B-per-lane streams where the
// layout allows, grid-strided over <= 2048 workgroups (8 per CU).
#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace pleas {

thread_local char g_last_error[256] = "";

// ---- profiling records: one (start, stop) event pair per profiled launch, resolved at collect time
bool g_prof_on = false;
unsigned g_prof_mask = ~0u;
struct ProfRec {
    hipEvent_t a, b;
    int kernel;
    double flops, bytes;
};
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_free;
static std::mutex g_prof_mu;

static hipEvent_t prof_event() {
    if (!g_prof_free.empty()) {
        hipEvent_t e = g_prof_free.back();
        g_prof_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);   // a failed creation surfaces as an invalid-handle error at collect time
    return e;
}
void prof_begin(int kernel, double flops, double bytes, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.a = prof_event();
    r.b = prof_event();
    r.kernel = kernel;
    r.flops = flops;
    r.bytes = bytes;
    (void)hipEventRecord(r.a, stream);
    g_prof_recs.push_back(r);
}
void prof_end(hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof_recs.back().b, stream);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kEwThreads = 256;
constexpr int kEwMaxBlocks = 2048;

static inline unsigned ew_grid(int64_t work_items) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(work_items, kEwThreads), kEwMaxBlocks));
}

// ------------------------------------------------------------------------------------------
// out[o][r][c][i] = coef(r) * ([row1[r],col1[c] present] w1[o][row1[r]][col1[c]][i] + same for w2)
// One thread per (o, r, c, i4) where i4 indexes VEC-wide pieces of the contiguous `inner` run.
template <int VEC>
__global__ __launch_bounds__(kEwThreads) void merge_blocks_kernel(
    const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ out, int64_t outer, int rows_out,
    int cols_out, int64_t inner, int rows_src, int cols_src, const int32_t* __restrict__ row1,
    const int32_t* __restrict__ row2, const int32_t* __restrict__ col1, const int32_t* __restrict__ col2,
    int n_merged_rows) {
    const int64_t inner_v = inner / VEC;
    const int64_t total = outer * rows_out * (int64_t)cols_out * inner_v;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t iv = idx % inner_v;
        int64_t t = idx / inner_v;
        const int c = (int)(t % cols_out);
        t /= cols_out;
        const int r = (int)(t % rows_out);
        const int64_t o = t / rows_out;
        const int r1 = row1[r], r2 = row2[r];
        const int c1 = col1 ? col1[c] : c, c2 = col2 ? col2[c] : c;
        const float coef = r < n_merged_rows ? 0.5f : 1.0f;
        float a[VEC], b[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = b[e] = 0.f;
        if (r1 >= 0 && c1 >= 0) {
            const float* src = w1 + ((o * rows_src + r1) * cols_src + c1) * inner + iv * VEC;
            if constexpr (VEC == 4) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = q[e];
            } else {
                a[0] = src[0];
            }
        }
        if (r2 >= 0 && c2 >= 0) {
            const float* src = w2 + ((o * rows_src + r2) * cols_src + c2) * inner + iv * VEC;
            if constexpr (VEC == 4) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = q[e];
            } else {
                b[0] = src[0];
            }
        }
        float* dst = out + ((o * rows_out + r) * cols_out + c) * inner + iv * VEC;
        if constexpr (VEC == 4) {
            f32x4 q;
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = (a[e] + b[e]) * coef;
            *reinterpret_cast<f32x4*>(dst) = q;
        } else {
            dst[0] = (a[0] + b[0]) * coef;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Inference BatchNorm folded to one affine map per channel, optional residual add, optional ReLU:
//   bn = x[n][c][i] * scale[c] + shift[c];   sum = bn (+ res[n][c][i]);   act = relu ? max(sum, 0) : sum
// One pass over x (and res): replaces the vendor BN + in-place add + in-place ReLU chain (3 kernels, 7 tensor passes)
// of a frozen source model.  `y_act` is always written; `y_bn` / `y_sum` (nullable) are the intermediate values for
// callers that track them (activation matching measures every node).  `inner_v` = HW / VEC.
// `rows_per_map` != 0: the tensor is several batches back to back along n, each with its own affine map
// (scale / shift [batch][channel]; rows_per_map = samples per batch * channels) -- train-mode BatchNorm of a forward that
// carries several matching batches.
template <int VEC>
__global__ __launch_bounds__(kEwThreads) void bn_act_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ res, float* __restrict__ y_bn,
                                                            float* __restrict__ y_sum, float* __restrict__ y,
                                                            int64_t total_v, unsigned inner_v, unsigned channels,
                                                            int relu, unsigned rows_per_map) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_v;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const unsigned row = (unsigned)(idx / inner_v);   // n * channels + c  (< 2^31, checked by the host)
        unsigned c = row % channels;
        if (rows_per_map) c += (row / rows_per_map) * channels;      // kernel-uniform test
        const float a = scale[c], b = shift[c];
        if constexpr (VEC == 4) {
            const f32x4 q = reinterpret_cast<const f32x4*>(x)[idx];
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (res) r = reinterpret_cast<const f32x4*>(res)[idx];
            f32x4 bn, sm, o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bn[e] = fmaf(q[e], a, b);
                sm[e] = bn[e] + r[e];
                o[e] = relu ? fmaxf(sm[e], 0.f) : sm[e];
            }
            if (y_bn) reinterpret_cast<f32x4*>(y_bn)[idx] = bn;
            if (y_sum) reinterpret_cast<f32x4*>(y_sum)[idx] = sm;
            reinterpret_cast<f32x4*>(y)[idx] = o;
        } else {
            const float bn = fmaf(x[idx], a, b);
            const float sm = bn + (res ? res[idx] : 0.f);
            if (y_bn) y_bn[idx] = bn;
            if (y_sum) y_sum[idx] = sm;
            y[idx] = relu ? fmaxf(sm, 0.f) : sm;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same affine map + ReLU followed by a max pooling window (the stem of a ResNet: bn1 -> relu -> maxpool), one pass:
//   y[n][c][oh][ow] = max over the window of act(x[n][c][ih][iw] * scale[c] + shift[c]),  padding = -inf (never wins)
// The full-resolution activation between ReLU and the pooling is never written (nothing between two hooked layers is
// consumed by the PLeaS loop): x is read once, y is a quarter of it.  scale == NULL: plain max pooling of x.
// A NaN in the window wins, as in the vendor's pooling kernel.  One thread per output element; the rows a window shares
// with the next output row are re-read from L1/L2 by the same workgroup.
__global__ __launch_bounds__(kEwThreads) void bn_act_pool_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, float* __restrict__ y,
                                                                 int64_t total, unsigned channels, int H, int W, int Ho,
                                                                 int Wo, int KH, int KW, int stride, int pad, int relu) {
    const unsigned plane_o = (unsigned)Ho * (unsigned)Wo;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const unsigned row = (unsigned)(idx / plane_o);      // n * channels + c
        const unsigned at = (unsigned)(idx - (int64_t)row * plane_o);
        const int oh = (int)(at / (unsigned)Wo), ow = (int)(at % (unsigned)Wo);
        const unsigned c = row % channels;
        const float a = scale ? scale[c] : 1.f, b = scale ? shift[c] : 0.f;
        const float* src = x + (int64_t)row * H * W;
        const int h0 = oh * stride - pad, w0 = ow * stride - pad;
        float best = -__builtin_inff();
        for (int kh = 0; kh < KH; ++kh) {
            const int ih = h0 + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < KW; ++kw) {
                const int iw = w0 + kw;
                if (iw < 0 || iw >= W) continue;
                float v = src[ih * W + iw];
                if (scale) v = fmaf(v, a, b);
                if (relu) v = fmaxf(v, 0.f);
                if (v > best || v != v) best = v;
            }
        }
        y[idx] = best;
    }
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kEwThreads) void masked_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                 const float* __restrict__ mask, float* __restrict__ m,
                                                                 float* __restrict__ v, int64_t n, float one_minus_b1,
                                                                 float b2, float one_minus_b2, float step_size,
                                                                 float bc2_sqrt, float eps) {
    // Same operation order as torch's single-tensor Adam: lerp for m, mul+addcmul for v,
    // (sqrt(v) / sqrt(bc2) + eps) denominator, addcdiv update.
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (mask) gi *= mask[i];
        float mi = m[i];
        mi = mi + one_minus_b1 * (gi - mi);
        float vi = v[i] * b2;
        vi = vi + one_minus_b2 * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - (step_size * mi) / denom;
        m[i] = mi;
        v[i] = vi;
    }
}

// ------------------------------------------------------------------------------------------
constexpr int kSqBlocks = 1024;

__global__ __launch_bounds__(kEwThreads) void sqerr_partial_kernel(const float* __restrict__ a,
                                                                   const float* __restrict__ b, int64_t n, float dscale,
                                                                   float* __restrict__ diff, float* __restrict__ part) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        s = fmaf(d, d, s);
        if (diff) diff[i] = dscale * d;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    __shared__ float ws[kEwThreads / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < kEwThreads / 64; ++w) t += ws[w];
        part[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(64) void sqerr_final_kernel(const float* __restrict__ part, int nparts, float scale,
                                                         int accumulate, float* __restrict__ out) {
    // fp64 combine of <= 1024 partials in a fixed order (one wave)
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += (double)part[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) {
        const float r = (float)(s * (double)scale);
        out[0] = accumulate ? out[0] + r : r;
    }
}

}  // namespace pleas

using namespace pleas;

extern "C" const char* pleas_version(void) { return "pleas_hip 0.5.0 gfx950"; }
extern "C" const char* pleas_last_error(void) { return g_last_error; }

// ---- arithmetic of the contraction kernels: -1 = not decided yet (PLEAS_ARITH read once), 0 = exact fp32 MFMA, 1 = split bf16
static int g_arith = -1;
namespace pleas {
int arith_mode() {
    if (g_arith < 0) {
        const char* e = std::getenv("PLEAS_ARITH");
        const char* legacy = std::getenv("PLEAS_GRAM_SPLIT_BF16");
        g_arith = ((e && (std::strcmp(e, "split_bf16") == 0 || std::strcmp(e, "1") == 0)) || (legacy && legacy[0] == '1')) ? 1 : 0;
    }
    return g_arith;
}
}  // namespace pleas
extern "C" void pleas_arith(int mode) { g_arith = mode == PLEAS_ARITH_SPLIT_BF16 ? 1 : 0; }
extern "C" int pleas_arith_get(void) { return pleas::arith_mode(); }

extern "C" void pleas_prof_enable(int on) { g_prof_on = on != 0; }

extern "C" void pleas_prof_select(unsigned kernel_mask) { g_prof_mask = kernel_mask; }

extern "C" void pleas_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_recs) {
        g_prof_free.push_back(r.a);
        g_prof_free.push_back(r.b);
    }
    g_prof_recs.clear();
}

extern "C" int pleas_prof_collect(int kernel, int64_t* launches, double* total_ms, double* flops, double* bytes) {
    if (kernel < 0 || kernel >= kProfCount || !launches || !total_ms || !flops || !bytes) return bad_arg("prof_collect");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    *launches = 0;
    *total_ms = *flops = *bytes = 0.0;
    for (auto& r : g_prof_recs) {
        if (r.kernel != kernel) continue;
        PLEAS_HIP_CHECK(hipEventSynchronize(r.b));
        float ms = 0.f;
        PLEAS_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        *launches += 1;
        *total_ms += ms;
        *flops += r.flops;
        *bytes += r.bytes;
    }
    return PLEAS_OK;
}

extern "C" int pleas_merge_blocks(const float* w1, const float* w2, float* out, int64_t outer, int rows_out,
                                  int cols_out, int64_t inner, int rows_src, int cols_src, const int32_t* row1,
                                  const int32_t* row2, const int32_t* col1, const int32_t* col2, int n_merged_rows,
                                  void* stream_) {
    if (!w1 || !w2 || !out || !row1 || !row2) return bad_arg("null pointer");
    if (outer < 0 || rows_out < 0 || cols_out < 0 || inner <= 0 || rows_src <= 0 || cols_src <= 0)
        return bad_arg("negative size");
    if ((col1 == nullptr) != (col2 == nullptr)) return bad_arg("col1/col2 must both be given or both NULL");
    if (!col1 && cols_out != cols_src) return bad_arg("cols_out != cols_src without column maps");
    const int64_t total = outer * rows_out * (int64_t)cols_out * inner;
    if (total == 0) return PLEAS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const bool vec = inner % 4 == 0 && ((((uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)out) & 15) == 0);
    ProfScope prof(kProfMergeBlocks, 0.0, 3.0 * total * sizeof(float), stream);
    if (vec)
        hipLaunchKernelGGL((merge_blocks_kernel<4>), dim3(ew_grid(total / 4)), dim3(kEwThreads), 0, stream, w1, w2, out,
                           outer, rows_out, cols_out, inner, rows_src, cols_src, row1, row2, col1, col2, n_merged_rows);
    else
        hipLaunchKernelGGL((merge_blocks_kernel<1>), dim3(ew_grid(total)), dim3(kEwThreads), 0, stream, w1, w2, out,
                           outer, rows_out, cols_out, inner, rows_src, cols_src, row1, row2, col1, col2, n_merged_rows);
    PLEAS_LAUNCH_CHECK("merge_blocks_kernel");
    return PLEAS_OK;
}

static int bn_act_launch(const float* x, const float* scale, const float* shift, const float* res, float* y_bn,
                         float* y_sum, float* y, int64_t n, int channels, int64_t inner, int relu, void* stream_,
                         int64_t n_per_map = 0) {
    if (!x || !scale || !shift || !y) return bad_arg("null pointer");
    if (n < 0 || channels <= 0 || inner <= 0) return bad_arg("negative size");
    if (n_per_map < 0 || (n_per_map > 0 && n % n_per_map != 0)) return bad_arg("samples do not split into whole batches");
    const unsigned rows_per_map = (n_per_map > 0 && n_per_map < n) ? (unsigned)(n_per_map * channels) : 0u;
    const int64_t total = n * channels * inner;
    if (total == 0) return PLEAS_OK;
    if (n * channels >= ((int64_t)1 << 31) || inner >= ((int64_t)1 << 31)) return bad_arg("tensor too large");
    hipStream_t stream = (hipStream_t)stream_;
    const uintptr_t all = (uintptr_t)x | (uintptr_t)y | (uintptr_t)res | (uintptr_t)y_bn | (uintptr_t)y_sum;
    const bool vec = inner % 4 == 0 && (all & 15) == 0;
    const double passes = 2.0 + (res ? 1.0 : 0.0) + (y_bn ? 1.0 : 0.0) + (y_sum ? 1.0 : 0.0);
    ProfScope prof(kProfBnAct, 0.0, passes * total * sizeof(float), stream);
    if (vec)
        hipLaunchKernelGGL((bn_act_kernel<4>), dim3(ew_grid(total / 4)), dim3(kEwThreads), 0, stream, x, scale, shift, res,
                           y_bn, y_sum, y, total / 4, (unsigned)(inner / 4), (unsigned)channels, relu, rows_per_map);
    else
        hipLaunchKernelGGL((bn_act_kernel<1>), dim3(ew_grid(total)), dim3(kEwThreads), 0, stream, x, scale, shift, res,
                           y_bn, y_sum, y, total, (unsigned)inner, (unsigned)channels, relu, rows_per_map);
    PLEAS_LAUNCH_CHECK("bn_act_kernel");
    return PLEAS_OK;
}

extern "C" int pleas_bn_act(const float* x, const float* scale, const float* shift, const float* res, float* y,
                            int64_t n, int channels, int64_t inner, int relu, void* stream_) {
    return bn_act_launch(x, scale, shift, res, nullptr, nullptr, y, n, channels, inner, relu, stream_);
}

extern "C" int pleas_bn_act_tracked(const float* x, const float* scale, const float* shift, const float* res, float* y_bn,
                                    float* y_sum, float* y, int64_t n, int channels, int64_t inner, int relu,
                                    void* stream_) {
    return bn_act_launch(x, scale, shift, res, y_bn, y_sum, y, n, channels, inner, relu, stream_);
}

// The ResNet stem's window (3 x 3, stride 2, padding 1) on rows of a multiple of 8 columns: one thread = FOUR adjacent
// outputs of one row = columns 8q-1 .. 8q+7 of three input rows: two 16-byte loads and one scalar per row (9 loads for 4
// outputs, where the general kernel issues 36), one 16-byte store.  32-bit index arithmetic (`units` < 2^31, host-checked).
__global__ __launch_bounds__(kEwThreads) void bn_act_pool3s2_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift,
                                                                    float* __restrict__ y, unsigned units,
                                                                    unsigned channels, int H, int W, int Ho, int relu) {
    const unsigned Wq = (unsigned)W / 8, per_plane = (unsigned)Ho * Wq;
    const float ninf = -__builtin_inff();
    for (unsigned u = blockIdx.x * blockDim.x + threadIdx.x; u < units; u += gridDim.x * blockDim.x) {
        const unsigned row = u / per_plane, at = u - row * per_plane;
        const int oh = (int)(at / Wq), q = (int)(at % Wq);
        const unsigned c = row % channels;
        const float a = scale ? scale[c] : 1.f, b = scale ? shift[c] : 0.f;
        const float* src = x + (int64_t)row * H * W + 8 * q;
        float best[4] = {ninf, ninf, ninf, ninf};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = 2 * oh - 1 + kh;
            if (ih < 0 || ih >= H) continue;
            const float* p = src + ih * W;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
            float v[9];
            v[0] = q > 0 ? p[-1] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[1 + e] = lo[e], v[5 + e] = hi[e];
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                if (scale) v[e] = fmaf(v[e], a, b);
                if (relu) v[e] = fmaxf(v[e], 0.f);
            }
            if (q == 0) v[0] = ninf;       // the padding column never wins
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float w = v[2 * j + t];
                    if (w > best[j] || w != w) best[j] = w;
                }
        }
        f32x4 o = {best[0], best[1], best[2], best[3]};
        reinterpret_cast<f32x4*>(y)[u] = o;       // y[row][oh][4q .. 4q+3]: unit order IS the output order
    }
}

extern "C" int pleas_bn_act_maxpool(const float* x, const float* scale, const float* shift, float* y, int64_t n,
                                    int channels, int H, int W, int KH, int KW, int stride, int pad, int relu,
                                    void* stream_) {
    if (!x || !y || (scale && !shift)) return bad_arg("null pointer");
    if (n < 0 || channels <= 0 || H <= 0 || W <= 0) return bad_arg("negative size");
    if (KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || 2 * pad > KH || 2 * pad > KW)
        return bad_arg("pooling window: need kernel > 0, stride > 0, 0 <= pad <= kernel / 2");
    if (H + 2 * pad < KH || W + 2 * pad < KW) return bad_arg("pooling window larger than the padded input");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;     // floor mode
    const int64_t total = n * channels * Ho * Wo;
    if (total == 0) return PLEAS_OK;
    if (n * channels >= ((int64_t)1 << 31) || (int64_t)H * W >= ((int64_t)1 << 31)) return bad_arg("tensor too large");
    hipStream_t stream = (hipStream_t)stream_;
    ProfScope prof(kProfBnAct, 0.0, ((double)n * channels * H * W + (double)total) * sizeof(float), stream);
    const bool stem = KH == 3 && KW == 3 && stride == 2 && pad == 1 && W % 8 == 0 && total / 4 < ((int64_t)1 << 31) &&
                      (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    if (stem)
        hipLaunchKernelGGL(bn_act_pool3s2_kernel, dim3(ew_grid(total / 4)), dim3(kEwThreads), 0, stream, x, scale, shift, y,
                           (unsigned)(total / 4), (unsigned)channels, H, W, Ho, relu);
    else
        hipLaunchKernelGGL(bn_act_pool_kernel, dim3(ew_grid(total)), dim3(kEwThreads), 0, stream, x, scale, shift, y, total,
                           (unsigned)channels, H, W, Ho, Wo, KH, KW, stride, pad, relu);
    PLEAS_LAUNCH_CHECK("bn_act_pool_kernel");
    return PLEAS_OK;
}

extern "C" int pleas_bn_act_tracked_batches(const float* x, const float* scale, const float* shift, const float* res,
                                            float* y_bn, float* y_sum, float* y, int64_t n_per_batch, int batches,
                                            int channels, int64_t inner, int relu, void* stream_) {
    if (batches <= 0 || n_per_batch < 0) return bad_arg("batches");
    return bn_act_launch(x, scale, shift, res, y_bn, y_sum, y, n_per_batch * batches, channels, inner, relu, stream_,
                         n_per_batch);
}

extern "C" int pleas_masked_adam(float* p, const float* g, const float* mask, float* m, float* v, int64_t n, float lr,
                                 float b1, float b2, float eps, int step, void* stream_) {
    if (!p || !g || !m || !v) return bad_arg("null pointer");
    if (n < 0 || step < 1) return bad_arg("n < 0 or step < 1");
    if (n == 0) return PLEAS_OK;
    // scalars prepared in double exactly like torch.optim.Adam (_single_tensor_adam)
    const double bc1 = 1.0 - std::pow((double)b1, step);
    const double bc2 = 1.0 - std::pow((double)b2, step);
    const double step_size = (double)lr / bc1;
    const double bc2_sqrt = std::sqrt(bc2);
    ProfScope prof(kProfMaskedAdam, 0.0, 8.0 * n * sizeof(float), (hipStream_t)stream_);
    hipLaunchKernelGGL(masked_adam_kernel, dim3(ew_grid(n)), dim3(kEwThreads), 0, (hipStream_t)stream_, p, g, mask, m, v,
                       n, (float)(1.0 - (double)b1), b2, (float)(1.0 - (double)b2), (float)step_size,
                       (float)bc2_sqrt, eps);
    PLEAS_LAUNCH_CHECK("masked_adam_kernel");
    return PLEAS_OK;
}

extern "C" size_t pleas_sqerr_ws_bytes(int64_t n) {
    (void)n;
    return kSqBlocks * sizeof(float);
}

extern "C" int pleas_sqerr(const float* a, const float* b, int64_t n, float scale, int accumulate, float* out,
                           float dscale, float* diff, void* ws, size_t ws_bytes, void* stream_) {
    if (!a || !b || !out) return bad_arg("null pointer");
    if (n < 0) return bad_arg("n < 0");
    if (!ws || ws_bytes < pleas_sqerr_ws_bytes(n)) return PLEAS_ENOMEM;
    hipStream_t stream = (hipStream_t)stream_;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, kEwThreads), kSqBlocks));
    ProfScope prof(kProfSqerr, 0.0, (diff ? 3.0 : 2.0) * n * sizeof(float), stream);
    hipLaunchKernelGGL(sqerr_partial_kernel, dim3(blocks), dim3(kEwThreads), 0, stream, a, b, n, dscale, diff,
                       (float*)ws);
    PLEAS_LAUNCH_CHECK("sqerr_partial_kernel");
    hipLaunchKernelGGL(sqerr_final_kernel, dim3(1), dim3(64), 0, stream, (const float*)ws, blocks, scale, accumulate,
                       out);
    PLEAS_LAUNCH_CHECK("sqerr_final_kernel");
    return PLEAS_OK;
}

// ------------------------------------------------------------------------------------------
// Bias gradient of a merged layer: gb[c] = sum over samples and pixels of resid[n][c][p] (pleas_merging.py:287, the bias
// node of autograd's backward).  One workgroup per channel; fixed summation order -- thread t takes pixels t, t + 256, ... of
// every sample in turn (nested loops, no division; consecutive lanes read consecutive addresses; 16-byte loads when HW % 4 == 0),
// then a tree over the threads: deterministic, no atomics.
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, int N, int C, long long HW,
                                                          float* __restrict__ out) {
    __shared__ float red[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    if ((HW & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const long long q = HW >> 2;
        for (int n = 0; n < N; ++n) {
            const f32x4* row = reinterpret_cast<const f32x4*>(x + ((size_t)n * C + c) * HW);
            float a = 0.f;
            for (long long i = tid; i < q; i += 256) {
                const f32x4 v = row[i];
                a += (v[0] + v[1]) + (v[2] + v[3]);
            }
            s += a;
        }
    } else {
        for (int n = 0; n < N; ++n) {
            const float* row = x + ((size_t)n * C + c) * HW;
            float a = 0.f;
            for (long long p = tid; p < HW; p += 256) a += row[p];
            s += a;
        }
    }
    red[tid] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) out[c] = red[0];
}

extern "C" int pleas_channel_sum(const float* x, int N, int C, int64_t HW, float* out, void* stream_) {
    if (!x || !out) return bad_arg("channel_sum: null pointer");
    if (N <= 0 || C <= 0 || HW <= 0) return bad_arg("channel_sum: empty tensor");
    hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream_, x, N, C, (long long)HW, out);
    PLEAS_LAUNCH_CHECK("channel_sum_kernel");
    return PLEAS_OK;
}

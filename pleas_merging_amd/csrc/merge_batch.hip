// Grouped block gather / average: the merged inputs `ip` of ALL layers of one PLeaS update in one launch.
//
// Replaces, per layer and per update, the input half of get_model_orig_activations
// (pleas/methods/pleas_merging.py:125-147: index_select x4 + mean + cat), i.e. what pleas_merge_blocks does for one
// tensor, for the whole layer list: 105 launches of 5-10 us each become one HBM-bound grid.
//
// Tensors are viewed [outer][rows][inner]; out[o][r][i] = coef(r) * ([row1[r] >= 0] w1[o][row1[r]][i] + [row2[r] >= 0]
// w2[o][row2[r]][i]), coef = 0.5 for r < n_merged else 1.  A workgroup owns kSpan consecutive 16-B (or 4-B) pieces of
// one tensor; the block -> tensor table is built once per shape sequence, only pointers are sent per update.
#include <algorithm>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace pleas {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int mThreads = 256;
constexpr int mUnroll = 4;
constexpr int mSpan = mThreads * mUnroll;   // pieces per workgroup

struct MergeItemDev {
    const float* w1;
    const float* w2;
    float* out;
    const int32_t* row1;
    const int32_t* row2;
    uint32_t total_v;      // pieces of this tensor (outer * rows_out * inner_v)
    uint32_t inner_v;      // pieces per row
    uint32_t rows_out, rows_src;
    int n_merged, vec;     // vec: 4 (16-B pieces) or 1
    int first_block;
    // subsampled form (vec == 1): out pixel p = (oh, ow) of `w_out` columns reads source pixel (oh * stride, ow * stride) of an
    // image with `w_src` columns and `inner_src` pixels; stride == 0: the plain form (source rows are inner_v long)
    int stride;
    uint32_t w_out, w_src, inner_src, pad;
};

__global__ __launch_bounds__(mThreads) void merge_batch_kernel(const MergeItemDev* __restrict__ items,
                                                               const int* __restrict__ block_item) {
    const MergeItemDev it = items[block_item[blockIdx.x]];
    const uint32_t base = (uint32_t)(blockIdx.x - it.first_block) * mSpan + threadIdx.x;
    // all loads of the unrolled pieces are issued before the first store
    uint32_t idx[mUnroll];
    int r1[mUnroll], r2[mUnroll];
    uint32_t o_[mUnroll], iv_[mUnroll], r_[mUnroll];
#pragma unroll
    for (int u = 0; u < mUnroll; ++u) {
        idx[u] = base + u * mThreads;
        const uint32_t id = min(idx[u], it.total_v - 1);
        const uint32_t t = id / it.inner_v;
        iv_[u] = id - t * it.inner_v;
        o_[u] = t / it.rows_out;
        r_[u] = t - o_[u] * it.rows_out;
        r1[u] = PLEAS_GLOBAL_I(it.row1)[r_[u]];
        r2[u] = PLEAS_GLOBAL_I(it.row2)[r_[u]];
    }
    if (it.vec == 4) {
        f32x4 a[mUnroll], b[mUnroll];
#pragma unroll
        for (int u = 0; u < mUnroll; ++u) {
            const size_t oa = ((size_t)o_[u] * it.rows_src + max(r1[u], 0)) * it.inner_v + iv_[u];
            const size_t ob = ((size_t)o_[u] * it.rows_src + max(r2[u], 0)) * it.inner_v + iv_[u];
            a[u] = reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(PLEAS_GLOBAL(it.w1))[oa];
            b[u] = reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(PLEAS_GLOBAL(it.w2))[ob];
        }
#pragma unroll
        for (int u = 0; u < mUnroll; ++u) {
            if (idx[u] >= it.total_v) continue;
            const float coef = (int)r_[u] < it.n_merged ? 0.5f : 1.0f;
            f32x4 q;   // absent sources were read from row 0 (always valid) and are dropped here, never multiplied
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = ((r1[u] >= 0 ? a[u][e] : 0.f) + (r2[u] >= 0 ? b[u][e] : 0.f)) * coef;
            reinterpret_cast<__attribute__((address_space(1))) f32x4*>(PLEAS_GLOBAL_W(it.out))[idx[u]] = q;
        }
    } else {
        float a[mUnroll], b[mUnroll];
#pragma unroll
        for (int u = 0; u < mUnroll; ++u) {
            uint32_t src = iv_[u], row_len = it.inner_v;
            if (it.stride > 0) {       // block-uniform
                const uint32_t oh = iv_[u] / it.w_out, ow = iv_[u] - oh * it.w_out;
                src = (oh * it.w_src + ow) * (uint32_t)it.stride;
                row_len = it.inner_src;
            }
            a[u] = PLEAS_GLOBAL(it.w1)[((size_t)o_[u] * it.rows_src + max(r1[u], 0)) * row_len + src];
            b[u] = PLEAS_GLOBAL(it.w2)[((size_t)o_[u] * it.rows_src + max(r2[u], 0)) * row_len + src];
        }
#pragma unroll
        for (int u = 0; u < mUnroll; ++u) {
            if (idx[u] >= it.total_v) continue;
            const float coef = (int)r_[u] < it.n_merged ? 0.5f : 1.0f;
            PLEAS_GLOBAL_W(it.out)[idx[u]] = ((r1[u] >= 0 ? a[u] : 0.f) + (r2[u] >= 0 ? b[u] : 0.f)) * coef;
        }
    }
}

constexpr int mPtrBatch = 90;
struct MergePtrBatch {
    int base, count;
    const float* w1[mPtrBatch];
    const float* w2[mPtrBatch];
    float* out[mPtrBatch];
    const int32_t* row1[mPtrBatch];
    const int32_t* row2[mPtrBatch];
};
__global__ void merge_set_ptrs_kernel(MergeItemDev* __restrict__ items, const MergePtrBatch b) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < b.count) {
        MergeItemDev& d = items[b.base + t];
        d.w1 = b.w1[t];
        d.w2 = b.w2[t];
        d.out = b.out[t];
        d.row1 = b.row1[t];
        d.row2 = b.row2[t];
    }
}

struct MergePlan {
    std::vector<int64_t> key;
    std::vector<MergeItemDev> items;
    std::vector<int> block_item;
    size_t off_items = 0, off_blocks = 0, total = 0;
    double bytes = 0;
    bool uploaded = false;
};
static PlanCache<MergePlan, 1> g_mplans;
static std::mutex g_mplan_mu;
static size_t malign(size_t v) { return (v + 255) / 256 * 256; }

static int build_merge_plan(MergePlan& P, const pleas_merge_item* it, int n) {
    P.items.assign(n, MergeItemDev());
    P.block_item.clear();
    P.bytes = 0;
    for (int i = 0; i < n; ++i) {
        const pleas_merge_item& m = it[i];
        if (m.outer < 0 || m.rows_out < 0 || m.inner <= 0 || m.rows_src <= 0 || m.n_merged < 0)
            return bad_arg("merge_batch: tensor geometry");
        const bool sub = m.sub_stride > 1;
        int64_t sub_wo = 0;
        if (sub) {
            if (m.sub_h <= 0 || m.sub_w <= 0) return bad_arg("merge_batch: subsampled tensor without an image size");
            sub_wo = ceil_div((int64_t)m.sub_w, m.sub_stride);
            if (m.inner != ceil_div((int64_t)m.sub_h, m.sub_stride) * sub_wo || (int64_t)m.sub_h * m.sub_w >= (1ll << 31))
                return bad_arg("merge_batch: inner must be ceil(sub_h / stride) * ceil(sub_w / stride)");
        }
        const bool vec = !sub && m.inner % 4 == 0;   // pointer alignment is checked per call
        const int64_t inner_v = vec ? m.inner / 4 : m.inner;
        const int64_t total_v = m.outer * m.rows_out * inner_v;
        if (total_v >= (1ll << 32) || inner_v >= (1ll << 32)) return bad_arg("merge_batch: tensor too large");
        MergeItemDev& d = P.items[i];
        d.total_v = (uint32_t)total_v;
        d.inner_v = (uint32_t)inner_v;
        d.rows_out = (uint32_t)m.rows_out;
        d.rows_src = (uint32_t)m.rows_src;
        d.n_merged = m.n_merged;
        d.vec = vec ? 4 : 1;
        d.stride = sub ? m.sub_stride : 0;
        d.w_out = (uint32_t)sub_wo;
        d.w_src = sub ? (uint32_t)m.sub_w : 0u;
        d.inner_src = sub ? (uint32_t)(m.sub_h * m.sub_w) : 0u;
        d.first_block = (int)P.block_item.size();
        const int64_t nb = ceil_div(total_v, mSpan);
        for (int64_t b = 0; b < nb; ++b) P.block_item.push_back(i);
        P.bytes += 3.0 * (double)(m.outer * m.rows_out * m.inner) * sizeof(float);
    }
    P.off_items = 0;
    P.off_blocks = malign(P.items.size() * sizeof(MergeItemDev));
    P.total = P.off_blocks + malign(P.block_item.size() * sizeof(int));
    P.uploaded = false;
    return PLEAS_OK;
}

}  // namespace pleas

using namespace pleas;

extern "C" size_t pleas_merge_batch_ws_bytes(const pleas_merge_item* items, int n_items) {
    if (!items || n_items <= 0) return 0;
    MergePlan tmp;
    if (build_merge_plan(tmp, items, n_items) != PLEAS_OK) return 0;
    return std::max<size_t>(tmp.total, 256);
}

extern "C" int pleas_merge_batch(const pleas_merge_item* items, int n_items, void* ws, size_t ws_bytes, int ws_fresh,
                                 void* stream_) {
    if (!items || n_items <= 0) return bad_arg("merge_batch: empty tensor list");
    for (int i = 0; i < n_items; ++i) {
        const pleas_merge_item& m = items[i];
        if (!m.w1 || !m.w2 || !m.out || !m.row1 || !m.row2) return bad_arg("merge_batch: null pointer");
        if (m.sub_stride <= 1 && m.inner % 4 == 0 && ((((uintptr_t)m.w1 | (uintptr_t)m.w2 | (uintptr_t)m.out) & 15) != 0))
            return bad_arg("merge_batch: tensors with inner % 4 == 0 must be 16-byte aligned");
    }
    hipStream_t stream = (hipStream_t)stream_;
    std::lock_guard<std::mutex> lk(g_mplan_mu);
    std::vector<int64_t> key;
    key.push_back(n_items);
    key.push_back((int64_t)(uintptr_t)ws);
    for (int i = 0; i < n_items; ++i) {
        const pleas_merge_item& m = items[i];
        for (int64_t v : {m.outer, m.inner, (int64_t)m.rows_out, (int64_t)m.rows_src, (int64_t)m.n_merged, (int64_t)m.sub_stride,
                          (int64_t)m.sub_h, (int64_t)m.sub_w})
            key.push_back(v);
    }
    MergePlan* hit = g_mplans.find(key);
    if (!hit) {
        hit = &g_mplans.take();
        const int rc = build_merge_plan(*hit, items, n_items);
        if (rc != PLEAS_OK) return rc;
        hit->key.swap(key);
    }
    MergePlan& P = *hit;
    if (ws_fresh) P.uploaded = false;
    if (P.block_item.empty()) return PLEAS_OK;
    if (!ws || ws_bytes < P.total) {
        std::snprintf(g_last_error, sizeof(g_last_error), "merge_batch workspace too small: need %zu bytes", P.total);
        P.key.clear();
        return PLEAS_ENOMEM;
    }
    char* base = (char*)ws;
    if (!P.uploaded) {
        g_mplans.claims_workspace(P);
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_items, P.items.data(), P.items.size() * sizeof(MergeItemDev),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipMemcpyAsync(base + P.off_blocks, P.block_item.data(), P.block_item.size() * sizeof(int),
                                       hipMemcpyHostToDevice, stream));
        PLEAS_HIP_CHECK(hipStreamSynchronize(stream));
        P.uploaded = true;
    }
    MergeItemDev* di = reinterpret_cast<MergeItemDev*>(base + P.off_items);
    for (int b0 = 0; b0 < n_items; b0 += mPtrBatch) {
        MergePtrBatch pb;
        pb.base = b0;
        pb.count = std::min(mPtrBatch, n_items - b0);
        for (int t = 0; t < pb.count; ++t) {
            const pleas_merge_item& m = items[b0 + t];
            pb.w1[t] = m.w1; pb.w2[t] = m.w2; pb.out[t] = m.out; pb.row1[t] = m.row1; pb.row2[t] = m.row2;
        }
        hipLaunchKernelGGL(merge_set_ptrs_kernel, dim3(1), dim3(128), 0, stream, di, pb);
        PLEAS_LAUNCH_CHECK("merge_set_ptrs_kernel");
    }
    {
        ProfScope prof(kProfMergeBlocks, 0.0, P.bytes, stream);
        hipLaunchKernelGGL(merge_batch_kernel, dim3((unsigned)P.block_item.size()), dim3(mThreads), 0, stream, di,
                           reinterpret_cast<const int*>(base + P.off_blocks));
    }
    PLEAS_LAUNCH_CHECK("merge_batch_kernel");
    return PLEAS_OK;
}

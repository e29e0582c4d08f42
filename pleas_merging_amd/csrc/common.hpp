// Shared host-side helpers of libpleas_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pleas_hip.h"

namespace pleas {

extern thread_local char g_last_error[256];

inline int hip_fail(hipError_t e, const char* what) {
    std::snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
    return PLEAS_EHIP;
}

inline int bad_arg(const char* what) {
    std::snprintf(g_last_error, sizeof(g_last_error), "invalid argument: %s", what);
    return PLEAS_EINVAL;
}

#define PLEAS_HIP_CHECK(expr)                                  \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return ::pleas::hip_fail(e_, #expr); \
    } while (0)

#define PLEAS_LAUNCH_CHECK(name)                                   \
    do {                                                           \
        hipError_t e_ = hipGetLastError();                         \
        if (e_ != hipSuccess) return ::pleas::hip_fail(e_, name);  \
    } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Pointers that reach a kernel through a device-side table are "generic" to hipcc, which then emits
// flat_load / flat_store: those count on lgkmcnt as well as vmcnt, so every LDS wait in the MFMA loop
// also waits for the global prefetch.  Casting to the global address space gives global_load / global_store.
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) const float cgfloat;
typedef __attribute__((address_space(1))) const int gcint;
#define PLEAS_GLOBAL(p) ((::pleas::cgfloat*)(p))
#define PLEAS_GLOBAL_W(p) ((::pleas::gfloat*)(p))
#define PLEAS_GLOBAL_I(p) ((::pleas::gcint*)(p))

// ---- alternative arithmetic of the contraction kernels (pleas_arith): an fp32 value as the EXACT sum of three bf16 values,
// v = h1 + h2 + h3 (8 + 8 + 8 significant bits, each the round-to-nearest bf16 of what the previous ones left; bf16 has
// fp32's exponent range), so that an fp32 product becomes six bf16-MFMA products with fp32 accumulation:
//   x * y = x1 y1 + (x1 y2 + x2 y1) + (x2 y2 + x1 y3 + x3 y1)  + terms <= 2^-26 |x y| (dropped: below half an fp32 ulp)
// Six v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each, K = 2): 2.67x
// less matrix-pipe time at fp32 accuracy.  The split is done once per element when a K chunk goes from registers to LDS.
// Not for inf / NaN operands (inf - inf in the residual).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kSplitRow = 104;   // bf16 per LDS row of a split image: 3 planes x 32 k + 8 pad = 208 B (13 16-byte slots: odd,
                                 // so the 16 rows of a ds_read_b128 lane group fall on 16 different slots)
__device__ __forceinline__ uint32_t split_bf16_pair(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));      // a in the low half
}
// two values at a time: one v_cvt_pk_bf16_f32 per pair and plane, the bf16 values back as floats by a shift / a mask
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t (&p)[3]) {
    p[0] = split_bf16_pair(a, b);
    const float ra = a - __builtin_bit_cast(float, p[0] << 16), rb = b - __builtin_bit_cast(float, p[0] & 0xffff0000u);
    p[1] = split_bf16_pair(ra, rb);
    p[2] = split_bf16_pair(ra - __builtin_bit_cast(float, p[1] << 16), rb - __builtin_bit_cast(float, p[1] & 0xffff0000u));
}
// Which row of a 32-row staging pass the 8-lane group g = tid / 8 (0 .. 31) writes under the split arithmetic.  Rows of a split
// image are kSplitRow = 208 bytes apart (the stride that keeps the 16-byte fragment READS conflict-free) and a lane writes 8 bytes
// per plane, so the four CONSECUTIVE rows of a half-wave overlap in 4 of 64 banks pairwise (a two-way conflict on every plane
// write: profiles/r05_pmc_split_*.txt); rows 4 apart start 16 banks apart and tile the 64 banks exactly.  Which lane group
// stages which row is free -- global rows are whole cache lines either way.
__device__ __forceinline__ int split_stage_row(int g) { return ((g & 3) << 2) | ((g >> 2) & 3) | (g & 16); }
// four consecutive k of one row -> the row's three planes of an LDS split image (`row` points at the row's plane 0, `col` = k)
__device__ __forceinline__ void split3_store4(__bf16* row, int col, float v0, float v1, float v2, float v3) {
    uint32_t lo[3], hi[3];
    split3_pair(v0, v1, lo);
    split3_pair(v2, v3, hi);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2_t*>(row + p * 32 + col) = u32x2_t{lo[p], hi[p]};
}
// the six products of one 16-deep k step, smallest terms first
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16_t split3_mfma(const bf16x8_t (&a)[3], const bf16x8_t (&b)[3], f32x16_t c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    return c;
}
// process-wide arithmetic switch (pleas_arith): 0 = exact fp32 MFMA (default), 1 = split bf16 where a kernel has the form
int arith_mode();


// ---- opt-in per-kernel timing with HIP events on the launch stream (pleas_prof_* in the C-ABI)
enum ProfKernel { kProfGramPartial = 0, kProfGramFinalize, kProfLsap, kProfMergeBlocks, kProfMaskedAdam, kProfSqerr,
                  kProfConvFwd, kProfConvWgrad, kProfNormalEq, kProfSolve, kProfBnAct, kProfConv2d, kProfCount };
extern bool g_prof_on;
extern unsigned g_prof_mask;   // bit k set: kernel id k is recorded while profiling is on
void prof_begin(int kernel, double flops, double bytes, hipStream_t stream);
void prof_end(hipStream_t stream);

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(int kernel, double flops, double bytes, hipStream_t stream) : s(stream), on(g_prof_on && ((g_prof_mask >> kernel) & 1u)) {
        if (on) prof_begin(kernel, flops, bytes, s);
    }
    ~ProfScope() {
        if (on) prof_end(s);
    }
};


// ---- a few plans per grouped launch (host side).
// A plan (work-item tables of one layer / node list) is keyed by its geometry AND the address of the caller's workspace
// the tables are uploaded to (element WS of the key).  Keeping several lets callers that alternate inside one process --
// two fitters, the per-group contractions of weight matching, matching next to PLeaS -- find their plan again instead
// of rebuilding and re-uploading it at every call.  Slots are recycled round robin.  Guarded by the caller's mutex.
template <class Plan, int WS, int N = 8>
struct PlanCache {
    Plan slots[N];
    unsigned turn = 0;
    Plan* last = nullptr;       // most recently used (diagnostics)
    Plan* find(const std::vector<int64_t>& key) {
        for (auto& p : slots)
            if (!p.key.empty() && p.key == key) return last = &p;
        return nullptr;
    }
    Plan& take() {              // slot for a new plan: the caller builds it, then sets its key
        Plan& p = slots[turn++ % N];
        p.key.clear();
        p.uploaded = false;
        return *(last = &p);
    }
    // p's tables are about to be written into its workspace: other plans that had theirs at that address lost them
    void claims_workspace(const Plan& p) {
        for (auto& o : slots)
            if (&o != &p && (int)o.key.size() > WS && (int)p.key.size() > WS && o.key[WS] == p.key[WS]) o.uploaded = false;
    }
};

// ---- XCD-aware work-item order for grouped launches (host side).
// Workgroup b of a grid runs on XCD b % 8, and every XCD has a private L2.  Items that read the same operand rows
// (same `key`) are therefore given grid positions of ONE residue class, so that their re-reads hit that L2 instead
// of crossing to the MALL / HBM once per XCD.  Groups go to the least-loaded of 8 queues (longest items first), the
// queues are interleaved (position 8 j + x = j-th item of queue x) and short queues are padded with `noop`.
// PLEAS_XCD_ORDER=0 in the environment keeps the plain longest-first order (A/B experiments).
template <class Item>
struct XcdWork {
    double w;       // relative duration of the item
    int64_t key;    // sharing key
    Item it;
};
template <class Item>
inline std::vector<Item> xcd_order_items(std::vector<XcdWork<Item>>& work, const Item& noop, bool by_default = true) {
    std::vector<Item> out;
    const char* env = std::getenv("PLEAS_XCD_ORDER");
    if (env ? env[0] == '0' : !by_default) {
        std::stable_sort(work.begin(), work.end(), [](const XcdWork<Item>& a, const XcdWork<Item>& b) { return a.w > b.w; });
        for (auto& x : work) out.push_back(x.it);
        return out;
    }
    struct Group { double total = 0, longest = 0; std::vector<size_t> idx; };
    std::vector<Group> groups;
    {
        std::vector<size_t> order(work.size());
        for (size_t i = 0; i < work.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return work[a].key < work[b].key; });
        int64_t last = 0;
        for (size_t n = 0; n < order.size(); ++n) {
            const size_t i = order[n];
            if (n == 0 || work[i].key != last) groups.emplace_back();
            last = work[i].key;
            Group& g = groups.back();
            g.total += work[i].w;
            g.longest = std::max(g.longest, work[i].w);
            g.idx.push_back(i);
        }
    }
    // balance first (largest groups placed first, each on the least-loaded queue), then run every queue longest items first
    std::stable_sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) { return a.total > b.total; });
    constexpr int kXcds = 8;
    std::vector<size_t> qgroups[kXcds];
    double load[kXcds] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        int best = 0;
        for (int x = 1; x < kXcds; ++x)
            if (load[x] < load[best]) best = x;
        load[best] += groups[gi].total;
        qgroups[best].push_back(gi);
    }
    std::vector<size_t> queue[kXcds];
    for (int x = 0; x < kXcds; ++x) {
        std::stable_sort(qgroups[x].begin(), qgroups[x].end(),
                         [&](size_t a, size_t b) { return groups[a].longest > groups[b].longest; });
        for (size_t gi : qgroups[x])
            for (size_t i : groups[gi].idx) queue[x].push_back(i);
    }
    size_t rows = 0;
    for (int x = 0; x < kXcds; ++x) rows = std::max(rows, queue[x].size());
    out.assign(rows * kXcds, noop);
    for (int x = 0; x < kXcds; ++x)
        for (size_t j = 0; j < queue[x].size(); ++j) out[j * kXcds + x] = work[queue[x][j]].it;
    return out;
}

}  // namespace pleas

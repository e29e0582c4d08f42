// Batched dense linear assignment on gfx950: one workgroup per problem, all problems of a
// model pair concurrently, cost matrices read in place (no D2H copy).
//
// Replaces pleas/core/solvers.py:18-33 (scipy.optimize.linear_sum_assignment on the host).
// Algorithm: shortest augmenting paths with fp64 duals (Crouse 2016), rows inserted in
// order.  To return exactly the assignment scipy returns -- also on tied inputs -- the
// column scan is expressed as a parallel arg-min over the key
//     (path length asc, assigned? asc, position in the `remaining` list: desc if unassigned, asc if assigned)
// which is the winner of scipy's sequential scan, and the
// `remaining` list is maintained with the same reversed fill and swap-remove.
//
// Latency-bound by construction (one dependent row scan per tree-growth step), so the design
// goal is a short critical path per step:
//   * each thread owns COLS columns whose state (column dual, tentative length, list position,
//     assigned row) lives in REGISTERS; all its cost loads of a step are issued back to back;
//   * the arg-min is a DPP row-scan inside each wave (no LDS round trips), one LDS slot per wave
//     and ONE workgroup barrier per step; every thread then replays the same scalar bookkeeping,
//     with a single writer per LDS word;
//   * all problems run concurrently (largest first), one CU each.
#include <algorithm>
#include <vector>

#include <limits>

#include "common.hpp"

namespace pleas {

#ifndef PLEAS_LSAP_THREADS
#define PLEAS_LSAP_THREADS 256
#endif
constexpr int kLsapThreads = PLEAS_LSAP_THREADS;
constexpr int kLsapWaves = kLsapThreads / 64;
constexpr int kMaxBatch = 128;

struct LsapBatch {
    const float* cost[kMaxBatch];
    int64_t* out[kMaxBatch];
    int n[kMaxBatch];
    int maximize;
};

struct Cand {
    double val;
    int key;    // (tie key << kColBits) | column ; smaller wins
    int owner;  // row assigned to that column (-1: free), filled in for the per-wave candidates only
};

// DPP move: lanes without a source keep their own value (harmless for a min reduction).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int x) {
    return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xF, false);
}

// Wave-wide lexicographic min of (val, key), returned in every lane: first the minimum of val (six DPP steps on the two
// halves of the double + one v_min_f64 each; lane 63 ends up with the wave's minimum), then the minimum of the keys of
// the lanes that attain it (six DPP steps on one int).  Comparing (val, key) pairs in every step instead cost three
// compares and three selects per step.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_min_f64(double x) {
    const int lo = dpp_mov<CTRL, ROW_MASK>(__double2loint(x));
    const int hi = dpp_mov<CTRL, ROW_MASK>(__double2hiint(x));
    return __builtin_fmin(x, __hiloint2double(hi, lo));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min_i32(int x) {
    const int o = dpp_mov<CTRL, ROW_MASK>(x);
    return o < x ? o : x;
}

__device__ __forceinline__ Cand wave_min(Cand c) {
    double m = c.val;
    m = dpp_min_f64<0x111, 0xF>(m);  // row_shr:1
    m = dpp_min_f64<0x112, 0xF>(m);  // row_shr:2
    m = dpp_min_f64<0x114, 0xF>(m);  // row_shr:4
    m = dpp_min_f64<0x118, 0xF>(m);  // row_shr:8  -> lane 15 of every row holds the row's min
    m = dpp_min_f64<0x142, 0xA>(m);  // row_bcast:15 into rows 1 and 3
    m = dpp_min_f64<0x143, 0xC>(m);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's min
    Cand r;
    r.val = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m), 63), __builtin_amdgcn_readlane(__double2loint(m), 63));
    int k = c.val == r.val ? c.key : 0x7fffffff;
    k = dpp_min_i32<0x111, 0xF>(k);
    k = dpp_min_i32<0x112, 0xF>(k);
    k = dpp_min_i32<0x114, 0xF>(k);
    k = dpp_min_i32<0x118, 0xF>(k);
    k = dpp_min_i32<0x142, 0xA>(k);
    k = dpp_min_i32<0x143, 0xC>(k);
    r.key = __builtin_amdgcn_readlane(k, 63);
    r.owner = -1;
    return r;
}

struct LsapShared {
    double* u;         // row duals
    double* shortest;  // tentative lengths, published at the end of a search
    int* path;         // predecessor row per column
    int* row4col;
    int* col4row;
    int* remaining;    // unscanned columns in scipy's order
    int* rowlist;      // rows of the current tree, in visit order
};

// Column owned by (thread, slot k): contiguous VEC-wide runs so that cost loads are 16-B wide.
template <int COLS>
__device__ __forceinline__ int col_of(int tid, int k) {
    constexpr int VEC = COLS < 4 ? COLS : 4;
    return ((k / VEC) * kLsapThreads + tid) * VEC + (k % VEC);
}

constexpr int kDeadKey = 0x7fffffff;
constexpr int kColBits = 12;      // column field of the tie key: n <= 4096; the rank field above it is < 2 n (13 bits)
constexpr int kColMask = (1 << kColBits) - 1;

// Tie key of a live column: (rank in scipy's scan order << kColBits) | column; smaller wins among equal path lengths.
__device__ __forceinline__ int tie_key(int n, int pos, int r4c, int j) {
    const int tie = r4c < 0 ? (n - 1 - pos) : (n + pos);
    return (tie << kColBits) | j;
}

template <int COLS>
__device__ void lsap_solve(const LsapShared sh, const float* __restrict__ cost, const int n, const float sign,
                           int64_t* __restrict__ out, Cand (*wave_best)[kLsapWaves]) {
    constexpr int VEC = COLS < 4 ? COLS : 4;
    const int tid = threadIdx.x;
    const bool vec_ok = (n % VEC == 0) && ((reinterpret_cast<uintptr_t>(cost) & 15) == 0);
    // Per owned column, in registers: dual v, tentative length sj, predecessor row, position in `remaining`, assigned
    // row at the start of the search, and the tie key (kDeadKey once the column is scanned, or beyond n).  The step
    // loop below is written with selects only: as branches (one exec-mask region per column and comparison) it ran to
    // ~850 instructions per step for 8 columns and WAS the step time (1.7 us).
    double v[COLS], sj[COLS];
    int pos[COLS], r4c[COLS], pathk[COLS], keyk[COLS];
#pragma unroll
    for (int k = 0; k < COLS; ++k) v[k] = 0.0;
    for (int t = tid; t < n; t += kLsapThreads) {
        sh.u[t] = 0.0;
        sh.row4col[t] = -1;
        sh.col4row[t] = -1;
        sh.path[t] = -1;
    }
    __syncthreads();

    for (int cur = 0; cur < n; ++cur) {
#pragma unroll
        for (int k = 0; k < COLS; ++k) {
            const int j = col_of<COLS>(tid, k);
            pos[k] = j < n ? n - 1 - j : -1;  // reversed fill: column j sits at position n-1-j
            sj[k] = INFINITY;
            pathk[k] = -1;
            r4c[k] = j < n ? sh.row4col[j] : 0;
            keyk[k] = j < n ? tie_key(n, pos[k], r4c[k], j) : kDeadKey;
        }
        for (int t = tid; t < n; t += kLsapThreads) sh.remaining[t] = n - 1 - t;
        __syncthreads();

        double dist = 0.0;
        int i = cur, sink = -1, live = n, nrows = 0, parity = 0;
        float c[COLS];
        auto load_row = [&](int row) {   // this thread's columns of one cost row, all loads issued back to back
            const float* __restrict__ crow = cost + (size_t)row * n;
            if (vec_ok) {
#pragma unroll
                for (int g = 0; g < COLS / VEC; ++g) {
                    const int j0 = col_of<COLS>(tid, g * VEC);
                    if constexpr (VEC == 4) {
                        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (j0 < n) q = *reinterpret_cast<const float4*>(crow + j0);
                        c[g * 4 + 0] = q.x, c[g * 4 + 1] = q.y, c[g * 4 + 2] = q.z, c[g * 4 + 3] = q.w;
                    } else if constexpr (VEC == 2) {
                        float2 q = make_float2(0.f, 0.f);
                        if (j0 < n) q = *reinterpret_cast<const float2*>(crow + j0);
                        c[g * 2 + 0] = q.x, c[g * 2 + 1] = q.y;
                    } else {
                        c[g] = j0 < n ? crow[j0] : 0.f;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < COLS; ++k) {
                    const int j = col_of<COLS>(tid, k);
                    c[k] = j < n ? crow[j] : 0.f;
                }
            }
        };
        load_row(i);
        while (sink < 0) {
            if (tid == 0) sh.rowlist[nrows] = i;
            ++nrows;
            const double ui = sh.u[i];
            // pass 1: relax the live columns, thread-local minimum of their tentative lengths
            double cv[COLS];
            Cand best;
            best.val = INFINITY;
            best.owner = -1;
#pragma unroll
            for (int k = 0; k < COLS; ++k) {
                const bool livecol = keyk[k] != kDeadKey;
                const double r = ((dist + (double)(sign * c[k])) - ui) - v[k];
                const bool upd = livecol & (r < sj[k]);
                sj[k] = upd ? r : sj[k];
                pathk[k] = upd ? i : pathk[k];
                cv[k] = livecol ? sj[k] : (double)INFINITY;
                best.val = __builtin_fmin(best.val, cv[k]);   // no NaNs here; the sign of a zero does not matter below
            }
            // pass 2: smallest tie key among the columns that attain it (dead columns carry kDeadKey)
            best.key = kDeadKey;
#pragma unroll
            for (int k = 0; k < COLS; ++k) {
                const int key = cv[k] == best.val ? keyk[k] : kDeadKey;
                best.key = key < best.key ? key : best.key;
            }
            best = wave_min(best);
            {   // the row this wave's candidate leads to: read now, beside the barrier wait, not after it
                const int jw = best.key & kColMask;
                best.owner = (best.key != kDeadKey && jw < n) ? sh.row4col[jw] : -1;
            }
            if ((tid & 63) == 0) wave_best[parity][tid >> 6] = best;
            __syncthreads();  // the only barrier of a step
            {
                Cand wb[kLsapWaves];
#pragma unroll
                for (int w = 0; w < kLsapWaves; ++w) wb[w] = wave_best[parity][w];
                best.val = wb[0].val;
#pragma unroll
                for (int w = 1; w < kLsapWaves; ++w) best.val = __builtin_fmin(best.val, wb[w].val);
                best.key = kDeadKey;
                best.owner = -1;
#pragma unroll
                for (int w = 0; w < kLsapWaves; ++w) {
                    const int key = wb[w].val == best.val ? wb[w].key : kDeadKey;
                    const bool take = key < best.key;
                    best.key = take ? key : best.key;
                    best.owner = take ? wb[w].owner : best.owner;
                }
            }
            parity ^= 1;
            // Every thread holds the same winner and replays the same bookkeeping; each LDS word
            // has one writer, and `remaining` slots written here are not read before the next barrier.
            dist = best.val;
            const int j = best.key & kColMask;
            const int tie = best.key >> kColBits;
            const int pj = tie < n ? (n - 1 - tie) : (tie - n);
            const int owner = best.owner;
            if (owner >= 0) load_row(owner);   // the next row's loads fly while the bookkeeping below runs
            --live;
            const int moved = sh.remaining[live];
            const bool swap = pj != live;
            // only the threads that own column j (now scanned) or the column moved into its slot have anything to do
            const int my_run = col_of<COLS>(tid, 0) / VEC;   // == tid: a thread's columns are runs tid, tid + T, ...
            const bool mine = (my_run == (j / VEC) % kLsapThreads) | (swap & (my_run == (moved / VEC) % kLsapThreads));
            if (mine) {
#pragma unroll
                for (int k = 0; k < COLS; ++k) {
                    const int jj = col_of<COLS>(tid, k);
                    const bool hit = jj == j;
                    const bool mv = swap & (jj == moved);
                    pos[k] = hit ? -1 : (mv ? pj : pos[k]);
                    keyk[k] = hit ? kDeadKey : (mv ? tie_key(n, pj, r4c[k], jj) : keyk[k]);
                }
            }
            if (tid == 0 && swap) sh.remaining[pj] = moved;
            if (owner < 0)
                sink = j;
            else
                i = owner;
        }

        // publish tentative lengths and predecessors, then dual update (rows in LDS, columns in registers)
#pragma unroll
        for (int k = 0; k < COLS; ++k) {
            const int j = col_of<COLS>(tid, k);
            if (j < n) {
                sh.shortest[j] = sj[k];
                sh.path[j] = pathk[k];
            }
            if (j < n && pos[k] < 0) v[k] -= dist - sj[k];
        }
        __syncthreads();
        for (int t = tid; t < nrows; t += kLsapThreads) {
            const int r = sh.rowlist[t];
            sh.u[r] += (r == cur) ? dist : dist - sh.shortest[sh.col4row[r]];
        }
        __syncthreads();
        if (tid == 0) {  // augment along the tree path back to `cur`
            int j = sink;
            for (;;) {
                const int r = sh.path[j];
                sh.row4col[j] = r;
                const int prev = sh.col4row[r];
                sh.col4row[r] = j;
                j = prev;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    for (int t = tid; t < n; t += kLsapThreads) out[t] = sh.col4row[t];
}

__global__ __launch_bounds__(kLsapThreads) void lsap_kernel(const LsapBatch batch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsap_smem[];
    __shared__ Cand wave_best[2][kLsapWaves];
    // The kernel is one long dependent chain on a single CU; other streams may fill the rest of the GPU meanwhile
    // (source forwards enqueued during the solve), and their waves then share this CU's SIMDs: issue ours first.
    __builtin_amdgcn_s_setprio(3);
    const int prob = blockIdx.x;
    const int n = batch.n[prob];
    LsapShared sh;
    sh.u = reinterpret_cast<double*>(lsap_smem);
    sh.shortest = sh.u + n;
    sh.path = reinterpret_cast<int*>(sh.shortest + n);
    sh.row4col = sh.path + n;
    sh.col4row = sh.row4col + n;
    sh.remaining = sh.col4row + n;
    sh.rowlist = sh.remaining + n;
    const float sign = batch.maximize ? -1.f : 1.f;
    const float* cost = batch.cost[prob];
    int64_t* out = batch.out[prob];
    if (n <= kLsapThreads)
        lsap_solve<1>(sh, cost, n, sign, out, wave_best);
    else if (n <= 2 * kLsapThreads)
        lsap_solve<2>(sh, cost, n, sign, out, wave_best);
    else if (n <= 4 * kLsapThreads)
        lsap_solve<4>(sh, cost, n, sign, out, wave_best);
    else if (n <= 8 * kLsapThreads)
        lsap_solve<8>(sh, cost, n, sign, out, wave_best);
    else
        lsap_solve<16>(sh, cost, n, sign, out, wave_best);   // n <= 4096: sixteen columns per thread, 144 KB of LDS state
}

}  // namespace pleas

using namespace pleas;

// Host solve of ONE problem with the same scan order and tie rule (the explicit host entry point of the
// C-ABI, for callers whose cost matrix lives on the host -- weight matching of CPU state dicts, BASELINE.json's
// configs[0]).  Never selected implicitly: device tensors always go to the kernel above.
template <class T>
static void lsap_host_solve(const T* cost, int n, double sign, int64_t* col4row_out) {
    std::vector<double> u(n, 0.0), v(n, 0.0), shortest(n);
    std::vector<int> path(n, -1), col4row(n, -1), row4col(n, -1), remaining(n);
    std::vector<char> in_rows(n), in_cols(n);
    for (int cur = 0; cur < n; ++cur) {
        std::fill(shortest.begin(), shortest.end(), std::numeric_limits<double>::infinity());
        std::fill(in_rows.begin(), in_rows.end(), 0);
        std::fill(in_cols.begin(), in_cols.end(), 0);
        int count = n;
        for (int j = 0; j < n; ++j) remaining[j] = n - 1 - j;          // unscanned columns, reversed
        double min_val = 0.0;
        int i = cur, sink = -1;
        while (sink < 0) {
            in_rows[i] = 1;
            double lowest = std::numeric_limits<double>::infinity();
            int index = -1;
            const T* row = cost + (size_t)i * n;
            for (int it = 0; it < count; ++it) {                       // list order; strict update; ties prefer a free column
                const int j = remaining[it];
                const double r = min_val + sign * (double)row[j] - u[i] - v[j];
                if (r < shortest[j]) {
                    shortest[j] = r;
                    path[j] = i;
                }
                if (shortest[j] < lowest || (shortest[j] == lowest && row4col[j] == -1)) {
                    lowest = shortest[j];
                    index = it;
                }
            }
            min_val = lowest;
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j;
            else i = row4col[j];
            in_cols[j] = 1;
            remaining[index] = remaining[--count];                     // swap-remove
        }
        u[cur] += min_val;
        for (int r = 0; r < n; ++r)
            if (in_rows[r] && r != cur) u[r] += min_val - shortest[col4row[r]];
        for (int j = 0; j < n; ++j)
            if (in_cols[j]) v[j] -= min_val - shortest[j];
        int j = sink;
        for (;;) {                                                     // augment back to `cur`
            const int r = path[j];
            row4col[j] = r;
            std::swap(col4row[r], j);
            if (r == cur) break;
        }
    }
    for (int r = 0; r < n; ++r) col4row_out[r] = col4row[r];
}

extern "C" int pleas_lsap_host(const void* cost, int is_double, int n, int maximize, int64_t* col_ind) {
    if (!cost || !col_ind) return bad_arg("null pointer");
    if (n < 1) return bad_arg("n < 1");
    const double sign = maximize ? -1.0 : 1.0;
    if (is_double) lsap_host_solve((const double*)cost, n, sign, col_ind);
    else lsap_host_solve((const float*)cost, n, sign, col_ind);
    return PLEAS_OK;
}

extern "C" int pleas_lsap_batched(const float* const* cost, const int* n, int nprob, int maximize,
                                  int64_t* const* col_ind, void* stream_) {
    if (nprob < 0) return bad_arg("nprob");
    if (nprob == 0) return PLEAS_OK;
    if (!cost || !n || !col_ind) return bad_arg("null array");
    for (int p = 0; p < nprob; ++p) {
        if (n[p] < 1 || n[p] > PLEAS_LSAP_MAX_N) return bad_arg("n out of range [1, PLEAS_LSAP_MAX_N]");
        if (!cost[p] || !col_ind[p]) return bad_arg("null problem pointer");
    }
    hipStream_t stream = (hipStream_t)stream_;
    // largest problems first: they are the tail of the launch
    std::vector<int> order(nprob);
    for (int p = 0; p < nprob; ++p) order[p] = p;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return n[a] > n[b]; });
    for (int start = 0; start < nprob; start += kMaxBatch) {
        const int cnt = std::min(kMaxBatch, nprob - start);
        LsapBatch batch;
        int nmax = 0;
        for (int q = 0; q < cnt; ++q) {
            const int p = order[start + q];
            batch.cost[q] = cost[p];
            batch.out[q] = col_ind[p];
            batch.n[q] = n[p];
            nmax = std::max(nmax, n[p]);
        }
        batch.maximize = maximize;
        const size_t lds = (size_t)nmax * (2 * sizeof(double) + 5 * sizeof(int));
        ProfScope prof(kProfLsap, 0.0, 0.0, stream);
        hipLaunchKernelGGL(lsap_kernel, dim3(cnt), dim3(kLsapThreads), lds, stream, batch);
        PLEAS_LAUNCH_CHECK("lsap_kernel");
    }
    return PLEAS_OK;
}

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
// Latency-bound by construction (one dependent row scan per tree-growth step): the design
// goal is a short critical path per step -- one coalesced row read, one wave-shuffle
// reduction, ONE workgroup barrier -- and full concurrency across the problems.
#include <algorithm>
#include <vector>

#include "common.hpp"

namespace pleas {

constexpr int kLsapThreads = 256;
constexpr int kLsapWaves = kLsapThreads / 64;
constexpr int kMaxBatch = 64;

struct LsapBatch {
    const float* cost[kMaxBatch];
    int64_t* out[kMaxBatch];
    int n[kMaxBatch];
    int maximize;
};

struct Cand {
    double val;
    int key;  // tie key, smaller wins
    int col;
};

__device__ __forceinline__ bool better(const Cand& a, const Cand& b) {  // a strictly better than b
    return a.val < b.val || (a.val == b.val && a.key < b.key);
}

__device__ __forceinline__ Cand shfl_xor_cand(const Cand& c, int off) {
    Cand o;
    o.val = __shfl_xor(c.val, off);
    o.key = __shfl_xor(c.key, off);
    o.col = __shfl_xor(c.col, off);
    return o;
}

__global__ __launch_bounds__(kLsapThreads) void lsap_kernel(const LsapBatch batch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsap_smem[];
    const int prob = blockIdx.x;
    const int n = batch.n[prob];
    const float* __restrict__ cost = batch.cost[prob];
    const float sign = batch.maximize ? -1.f : 1.f;
    const int tid = threadIdx.x;

    double* u = reinterpret_cast<double*>(lsap_smem);  // row duals
    double* v = u + n;                                  // column duals
    double* shortest = v + n;                           // tentative path length per column
    int* path = reinterpret_cast<int*>(shortest + n);   // predecessor row per column
    int* row4col = path + n;
    int* col4row = row4col + n;
    int* remaining = col4row + n;                       // unscanned columns, scipy's order
    int* pos = remaining + n;                           // position of a column in `remaining`, -1 once scanned
    int* rowseen = pos + n;                             // rows in the current tree
    __shared__ Cand wave_best[2][kLsapWaves];

    for (int t = tid; t < n; t += kLsapThreads) {
        u[t] = 0.0;
        v[t] = 0.0;
        row4col[t] = -1;
        col4row[t] = -1;
        path[t] = -1;
    }
    __syncthreads();

    for (int cur = 0; cur < n; ++cur) {
        for (int t = tid; t < n; t += kLsapThreads) {
            remaining[t] = n - 1 - t;
            pos[t] = n - 1 - t;  // column t sits at position n-1-t
            shortest[t] = INFINITY;
            rowseen[t] = 0;
        }
        __syncthreads();

        double dist = 0.0;
        int i = cur, sink = -1, live = n;
        int parity = 0;
        while (sink < 0) {
            if (tid == 0) rowseen[i] = 1;
            const double ui = u[i];
            const float* __restrict__ crow = cost + (size_t)i * n;
            Cand best;
            best.val = INFINITY;
            best.key = 0x7fffffff;
            best.col = -1;
            for (int j = tid; j < n; j += kLsapThreads) {
                const int pj = pos[j];
                if (pj < 0) continue;
                const double c = (double)(sign * crow[j]);
                const double r = ((dist + c) - ui) - v[j];
                double sj = shortest[j];
                if (r < sj) {
                    sj = r;
                    shortest[j] = r;
                    path[j] = i;
                }
                Cand cand;
                cand.val = sj;
                cand.key = row4col[j] < 0 ? (n - 1 - pj) : (n + pj);
                cand.col = j;
                if (better(cand, best)) best = cand;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const Cand o = shfl_xor_cand(best, off);
                if (better(o, best)) best = o;
            }
            if ((tid & 63) == 0) wave_best[parity][tid >> 6] = best;
            __syncthreads();  // the only barrier of a step
            best = wave_best[parity][0];
#pragma unroll
            for (int w = 1; w < kLsapWaves; ++w) {
                const Cand o = wave_best[parity][w];
                if (better(o, best)) best = o;
            }
            parity ^= 1;
            // every thread now holds the same winner; bookkeeping is replicated, each LDS word has one writer
            dist = best.val;
            const int j = best.col;
            const int pj = best.key < n ? (n - 1 - best.key) : (best.key - n);
            const int owner = row4col[j];
            --live;
            const int moved = remaining[live];
            // Single-writer rule: pos[c] belongs to thread c % T (the only reader of pos[c] in the scan),
            // `remaining` is written by thread 0 at a slot nobody reads before the next barrier.
            if (tid == (j % kLsapThreads)) pos[j] = -1;
            if (pj != live) {
                if (tid == (moved % kLsapThreads)) pos[moved] = pj;
                if (tid == 0) remaining[pj] = moved;
            }
            if (owner < 0)
                sink = j;
            else
                i = owner;
        }

        // dual update (parallel), then augmentation (serial walk along the tree path)
        for (int t = tid; t < n; t += kLsapThreads) {
            if (t == cur)
                u[t] += dist;
            else if (rowseen[t])
                u[t] += dist - shortest[col4row[t]];
        }
        for (int t = tid; t < n; t += kLsapThreads)
            if (pos[t] < 0) v[t] -= dist - shortest[t];
        __syncthreads();
        if (tid == 0) {
            int j = sink;
            for (;;) {
                const int r = path[j];
                row4col[j] = r;
                const int prev = col4row[r];
                col4row[r] = j;
                j = prev;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    int64_t* out = batch.out[prob];
    for (int t = tid; t < n; t += kLsapThreads) out[t] = col4row[t];
}

}  // namespace pleas

using namespace pleas;

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
        const size_t lds = (size_t)nmax * (3 * sizeof(double) + 6 * sizeof(int));
        ProfScope prof(kProfLsap, 0.0, 0.0, stream);
        hipLaunchKernelGGL(lsap_kernel, dim3(cnt), dim3(kLsapThreads), lds, stream, batch);
        PLEAS_LAUNCH_CHECK("lsap_kernel");
    }
    return PLEAS_OK;
}

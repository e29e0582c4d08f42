// Batched SPD solve on gfx950 for the PLeaS normal equations:  X (A + lambda*mean(diag A) I) = B^T
// for many independent (A, B^T) of different sizes in one call.
//
// Closed-form counterpart of the reference's Adam loop (pleas/methods/pleas_merging.py:357-375):
// after pleas_normal_eq_accum / pleas_wgrad_batch have built A (K x K) and B^T (N x K) per layer,
// every row of B^T is one right-hand side.
//
// Algorithm: right-looking blocked Cholesky, panel width 64, on the LOWER triangle, with the
// right-hand-side rows appended below A so that forward substitution is part of the same sweep:
//   step j:  (1) factor the 64x64 diagonal block in LDS                         (one workgroup / problem)
//            (2) panel rows (rest of A and all RHS rows) <- rows * L11^-T        (one thread / row)
//                and mirror the A part of the panel into the UPPER triangle (U = L^T)
//            (3) trailing block -= panel . panel^T                               (fp32 MFMA tiles)
// then back substitution sweeps the panels in reverse:  X_j = (Y_j) L11^-1 ;  Y_{<j} -= X_j . U_{<j,j}^T,
// which is the SAME NT tile kernel as (3) because U was stored row-wise.  All problems that are
// still active at a step share that step's launches (grid.y = problem), so a call costs
// 5 * max(K)/64 launches regardless of the number of problems.
#include <algorithm>
#include <vector>

#include "common.hpp"

namespace pleas {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int sNB = 64;       // panel width
constexpr int sMaxProb = 96;  // problems per launch group (kernel-argument table)
constexpr int sLds = 36;

struct SolveProb {
    float* A;   // K x K row-major
    float* Bt;  // N x K row-major (rows = right-hand sides); overwritten with the solution
    int K, N;
};
struct SolveBatch {
    SolveProb p[sMaxProb];
    int count;
};

__device__ __forceinline__ float* row_ptr(const SolveProb& P, int i) {  // row i of the stacked [A; Bt]
    return i < P.K ? P.A + (size_t)i * P.K : P.Bt + (size_t)(i - P.K) * P.K;
}

// ---- ridge: A[d][d] += lambda * mean(diag A) ------------------------------------------------------------
__global__ __launch_bounds__(256) void ridge_kernel(const SolveBatch b, float lambda) {
    const SolveProb P = b.p[blockIdx.x];
    __shared__ float red[256];
    float s = 0.f;
    for (int d = threadIdx.x; d < P.K; d += 256) s += P.A[(size_t)d * P.K + d];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float add = lambda * red[0] / (float)P.K;
    for (int d = threadIdx.x; d < P.K; d += 256) P.A[(size_t)d * P.K + d] += add;
}

// ---- (1) diagonal block: in-LDS Cholesky of A[j0:j0+nb, j0:j0+nb] ----------------------------------------
__global__ __launch_bounds__(256) void potrf_diag_kernel(const SolveBatch b, int j0, int* __restrict__ info, int info_base) {
    const SolveProb P = b.p[blockIdx.x];
    if (j0 >= P.K) return;
    const int nb = min(sNB, P.K - j0);
    __shared__ float D[sNB][sNB + 1];
    for (int e = threadIdx.x; e < nb * nb; e += 256) {
        const int r = e / nb, c = e - r * nb;
        D[r][c] = c <= r ? P.A[(size_t)(j0 + r) * P.K + j0 + c] : 0.f;
    }
    __syncthreads();
    for (int k = 0; k < nb; ++k) {
        float piv = D[k][k];
        if (!(piv > 0.f)) {  // not positive definite in fp32: flag, keep going with a tiny pivot
            if (threadIdx.x == 0 && info[info_base + blockIdx.x] == 0) info[info_base + blockIdx.x] = j0 + k + 1;
            piv = 1e-30f;
        }
        const float d = sqrtf(piv);
        __syncthreads();
        for (int r = k + threadIdx.x; r < nb; r += 256) D[r][k] = (r == k) ? d : D[r][k] / d;
        __syncthreads();
        for (int e = threadIdx.x; e < (nb - k - 1) * (nb - k - 1); e += 256) {
            const int r = k + 1 + e / (nb - k - 1), c = k + 1 + e % (nb - k - 1);
            if (c <= r) D[r][c] -= D[r][k] * D[c][k];
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < nb * nb; e += 256) {
        const int r = e / nb, c = e - r * nb;
        if (c <= r) P.A[(size_t)(j0 + r) * P.K + j0 + c] = D[r][c];
    }
}

// ---- (2) panel rows: x L11^T = a  (forward)  |  x L11 = y (backward), one thread per row --------------------
template <bool BACKWARD>
__global__ __launch_bounds__(64) void trsm_rows_kernel(const SolveBatch b, int j0) {
    const SolveProb P = b.p[blockIdx.y];
    if (j0 >= P.K) return;
    const int nb = min(sNB, P.K - j0);
    __shared__ float L[sNB][sNB + 1];
    for (int e = threadIdx.x; e < nb * nb; e += 64) {
        const int r = e / nb, c = e - r * nb;
        L[r][c] = c <= r ? P.A[(size_t)(j0 + r) * P.K + j0 + c] : 0.f;
    }
    __syncthreads();
    // forward: rows below the diagonal block of A plus all RHS rows; backward: RHS rows only
    const int first = BACKWARD ? P.K : j0 + nb;
    const int i = first + blockIdx.x * 64 + threadIdx.x;
    if (i >= P.K + P.N) return;
    float* row = row_ptr(P, i) + j0;
    float x[sNB];
#pragma unroll
    for (int t = 0; t < sNB; ++t) x[t] = t < nb ? row[t] : 0.f;
    if (!BACKWARD) {
#pragma unroll
        for (int t = 0; t < sNB; ++t) {
            if (t < nb) {
                float s = x[t];
#pragma unroll
                for (int q = 0; q < sNB; ++q)
                    if (q < t) s -= x[q] * L[t][q];
                x[t] = s / L[t][t];
            }
        }
    } else {
#pragma unroll
        for (int tt = 0; tt < sNB; ++tt) {
            const int t = sNB - 1 - tt;
            if (t < nb) {
                float s = x[t];
#pragma unroll
                for (int q = 0; q < sNB; ++q)
                    if (q > t && q < nb) s -= x[q] * L[q][t];
                x[t] = s / L[t][t];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < sNB; ++t)
        if (t < nb) row[t] = x[t];
    if (!BACKWARD && i < P.K) {  // mirror L21 into the upper triangle: U[j0+t][i] = L[i][j0+t]
#pragma unroll
        for (int t = 0; t < sNB; ++t)
            if (t < nb) P.A[(size_t)(j0 + t) * P.K + i] = x[t];
    }
}

// ---- (3) trailing update: out[i][c] -= sum_t Prow_i[j0+t] * Qrow_c[j0+t], 128x128 MFMA tiles -------------------
//   forward : i in [j0+nb, K+N), c in [j0+nb, K), tiles strictly above the diagonal of A skipped
//   backward: i in [K, K+N) (RHS rows), c in [0, j0); Q rows are rows c of A read in the UPPER triangle
template <bool BACKWARD>
__global__ __launch_bounds__(256) void trail_update_kernel(const SolveBatch b, int j0) {
    const SolveProb P = b.p[blockIdx.z];
    if (j0 >= P.K) return;
    const int nb = min(sNB, P.K - j0);
    const int i_first = BACKWARD ? P.K : j0 + nb, i_end = P.K + P.N;
    const int c_first = BACKWARD ? 0 : j0 + nb, c_end = BACKWARD ? j0 : P.K;
    const int i0 = i_first + blockIdx.y * 128, c0 = c_first + blockIdx.x * 128;
    if (i0 >= i_end || c0 >= c_end) return;
    if (!BACKWARD && i0 < P.K && c0 > min(i0 + 127, P.K - 1)) return;  // whole tile above the diagonal of A
    __shared__ __attribute__((aligned(16))) float As[128 * sLds * 2];   // P tile then Q tile, one 32-wide chunk
    float* Bs = As + 128 * sLds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    const int srow = tid >> 3, scol = (tid & 7) * 4;  // 32 rows x 8 lanes x 4 floats per pass
    for (int ch = 0; ch < sNB / 32; ++ch) {
        const int t0 = ch * 32;
        if (t0 >= nb) break;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = srow + q * 32;
            float va[4], vb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = t0 + scol + e;
                const int gi = i0 + r, gc = c0 + r;
                va[e] = (gi < i_end && t < nb) ? row_ptr(P, gi)[j0 + t] : 0.f;
                vb[e] = (gc < c_end && t < nb) ? P.A[(size_t)gc * P.K + j0 + t] : 0.f;
            }
            *reinterpret_cast<f32x4*>(As + r * sLds + scol) = f32x4{va[0], va[1], va[2], va[3]};
            *reinterpret_cast<f32x4*>(Bs + r * sLds + scol) = f32x4{vb[0], vb[1], vb[2], vb[3]};
        }
        __syncthreads();
        const float* a = As + (wm * 64 + (lane & 31)) * sLds + 4 * (lane >> 5);
        const float* bq = Bs + (wn * 64 + (lane & 31)) * sLds + 4 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                fa[s] = *reinterpret_cast<const f32x4*>(a + s * 32 * sLds + kk * 8);
                fb[s] = *reinterpret_cast<const f32x4*>(bq + s * 32 * sLds + kk * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sm = 0; sm < 2; ++sm)
#pragma unroll
                    for (int sn = 0; sn < 2; ++sn)
                        acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][e], fb[sn][e], acc[sm][sn], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int sm = 0; sm < 2; ++sm)
#pragma unroll
        for (int sn = 0; sn < 2; ++sn) {
            const int c = c0 + wn * 64 + sn * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wm * 64 + sm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                // forward: only the lower triangle of A (c <= i) and the RHS rows are live
                if (i < i_end && c < c_end && (BACKWARD || i >= P.K || c <= i)) row_ptr(P, i)[c] -= acc[sm][sn][r];
            }
        }
}

}  // namespace pleas

using namespace pleas;

extern "C" int pleas_cholesky_solve_batched(float* const* A, float* const* Bt, const int* K, const int* N, int nprob,
                                            float lambda, int* info, void* stream_) {
    if (nprob < 0) return bad_arg("cholesky_solve: nprob");
    if (nprob == 0) return PLEAS_OK;
    if (!A || !Bt || !K || !N || !info) return bad_arg("cholesky_solve: null array");
    for (int p = 0; p < nprob; ++p)
        if (!A[p] || !Bt[p] || K[p] <= 0 || N[p] < 0) return bad_arg("cholesky_solve: bad problem");
    hipStream_t stream = (hipStream_t)stream_;
    PLEAS_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int) * (size_t)nprob, stream));
    double flops = 0;
    for (int p = 0; p < nprob; ++p) flops += (double)K[p] * K[p] * K[p] / 3.0 + 2.0 * (double)N[p] * K[p] * K[p];
    ProfScope prof(kProfSolve, flops, 0.0, stream);
    for (int g0 = 0; g0 < nprob; g0 += sMaxProb) {
        SolveBatch b;
        b.count = std::min(sMaxProb, nprob - g0);
        int Kmax = 0, rows_max = 0;
        for (int q = 0; q < b.count; ++q) {
            b.p[q] = SolveProb{A[g0 + q], Bt[g0 + q], K[g0 + q], N[g0 + q]};
            Kmax = std::max(Kmax, K[g0 + q]);
            rows_max = std::max(rows_max, K[g0 + q] + N[g0 + q]);
        }
        if (lambda > 0.f) {
            hipLaunchKernelGGL(ridge_kernel, dim3(b.count), dim3(256), 0, stream, b, lambda);
            PLEAS_LAUNCH_CHECK("ridge_kernel");
        }
        for (int j0 = 0; j0 < Kmax; j0 += sNB) {  // factorisation + forward substitution
            hipLaunchKernelGGL(potrf_diag_kernel, dim3(b.count), dim3(256), 0, stream, b, j0, info, g0);
            const int nbmax = std::min(sNB, Kmax - j0);
            const int rows = rows_max - j0 - nbmax;  // rows below the diagonal block (A rows + RHS rows)
            if (rows > 0) {
                hipLaunchKernelGGL((trsm_rows_kernel<false>), dim3((unsigned)ceil_div(rows, 64), b.count), dim3(64), 0, stream,
                                   b, j0);
                const int cols = Kmax - j0 - nbmax;
                if (cols > 0)
                    hipLaunchKernelGGL((trail_update_kernel<false>),
                                       dim3((unsigned)ceil_div(cols, 128), (unsigned)ceil_div(rows, 128), b.count), dim3(256),
                                       0, stream, b, j0);
            }
        }
        int Nmax = 0;
        for (int q = 0; q < b.count; ++q) Nmax = std::max(Nmax, N[g0 + q]);
        if (Nmax > 0) {
            const int last = (Kmax - 1) / sNB * sNB;
            for (int j0 = last; j0 >= 0; j0 -= sNB) {  // back substitution
                hipLaunchKernelGGL((trsm_rows_kernel<true>), dim3((unsigned)ceil_div(Nmax, 64), b.count), dim3(64), 0, stream, b,
                                   j0);
                if (j0 > 0)
                    hipLaunchKernelGGL((trail_update_kernel<true>),
                                       dim3((unsigned)ceil_div(j0, 128), (unsigned)ceil_div(Nmax, 128), b.count), dim3(256), 0,
                                       stream, b, j0);
            }
        }
        PLEAS_LAUNCH_CHECK("cholesky_solve kernels");
    }
    return PLEAS_OK;
}

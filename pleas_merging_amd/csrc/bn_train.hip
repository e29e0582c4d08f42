// Train-mode BatchNorm of the matching forward (gfx950).
//
// Neither reference driver calls .eval() before activation_matching (run_domainnet.py:172-186, :257-264), so the twin
// forward normalises every BatchNorm2d with the BATCH's statistics and updates the running statistics as a side effect.
// A train-mode BatchNorm is still one affine map per channel,
//     y = x * scale[c] + shift[c],   scale = gamma / sqrt(var_b + eps),   shift = beta - mean * scale,
// with mean / var_b (biased) over (n, h, w).  This file computes that fold on the device -- one streaming pass over x --
// so that the rest of the chain is the same pleas_bn_act_tracked launch as in eval mode and tracked BatchNorm nodes can
// be DERIVED from their convolution node in the matching reduce pass (gram.hip) instead of being contracted.
//
//   bn_stats_partial_kernel   grid (C, S): workgroup (c, s) sums x and x^2 of channel c over samples n = s, s+S, ...
//                             (every (n, c) slab is HW contiguous floats: 16-B loads), fp64 accumulation
//   bn_train_finalize_kernel  one workgroup: per channel combines the S partials in a fixed order (deterministic), writes
//                             scale / shift, updates running_mean / running_var (unbiased) with the momentum or the
//                             cumulative average, then bumps num_batches_tracked
// HBM-bound: 4 B read per element once (the vendor train-mode BatchNorm reads x twice and writes once; with
// pleas_bn_act_tracked the chain BatchNorm -> (+identity) -> ReLU costs two reads of x + one of the residual + the writes).
#include "common.hpp"

namespace pleas {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBnThreads = 256;

__global__ __launch_bounds__(kBnThreads) void bn_stats_partial_kernel(const float* __restrict__ x, int n_samples,
                                                                      int channels, int64_t inner, int vec,
                                                                      double* __restrict__ part) {
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    // blockIdx.z: which of the batches that lie back to back along the sample axis (each has its own statistics)
    x += (int64_t)blockIdx.z * n_samples * channels * inner;
    part += (int64_t)blockIdx.z * S * channels * 2;
    double a = 0.0, b = 0.0;
    for (int n = s; n < n_samples; n += S) {
        const float* row = x + ((int64_t)n * channels + c) * inner;
        if (vec) {
            const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
            const int64_t inner4 = inner >> 2;
            for (int64_t i = threadIdx.x; i < inner4; i += kBnThreads) {
                const f32x4 q = row4[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double v = (double)q[e];
                    a += v;
                    b = fma(v, v, b);
                }
            }
        } else {
            for (int64_t i = threadIdx.x; i < inner; i += kBnThreads) {
                const double v = (double)row[i];
                a += v;
                b = fma(v, v, b);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    __shared__ double wa[kBnThreads / 64], wb[kBnThreads / 64];
    if ((threadIdx.x & 63) == 0) {
        wa[threadIdx.x >> 6] = a;
        wb[threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int w = 0; w < kBnThreads / 64; ++w) {
            ta += wa[w];
            tb += wb[w];
        }
        part[((int64_t)s * channels + c) * 2 + 0] = ta;
        part[((int64_t)s * channels + c) * 2 + 1] = tb;
    }
}

__global__ __launch_bounds__(1024) void bn_train_finalize_kernel(double* __restrict__ part, int S, int channels,
                                                                 int batches, double count,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, double eps,
                                                                 double momentum, float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var,
                                                                 int64_t* __restrict__ num_batches_tracked,
                                                                 float* __restrict__ scale, float* __restrict__ shift) {
    // (1) every (batch, channel) pair on its own thread: combine the S partials in a fixed order, write scale / shift
    //     [batch][channel], leave (mean, unbiased variance) in the workspace's tail [batch][channel][2];
    // (2) per channel, the batches IN ORDER: the running statistics move on as after that many forwards of the module --
    //     exponential factor as torch.nn.BatchNorm2d.forward: `momentum`, or 1 / (batches seen including this one).
    double* moments = part + (int64_t)batches * S * channels * 2;
    const int64_t seen = (momentum < 0.0 && num_batches_tracked) ? *num_batches_tracked : 0;
    for (int idx = threadIdx.x; idx < batches * channels; idx += blockDim.x) {
        const int bt = idx / channels, c = idx - bt * channels;
        const double* pb = part + (int64_t)bt * S * channels * 2;
        double a = 0.0, b = 0.0;
        for (int s = 0; s < S; ++s) {
            a += pb[((int64_t)s * channels + c) * 2 + 0];
            b += pb[((int64_t)s * channels + c) * 2 + 1];
        }
        const double mean = a / count;
        double var = b / count - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double g = gamma ? (double)gamma[c] : 1.0;
        const double sc = g / sqrt(var + eps);
        scale[idx] = (float)sc;
        shift[idx] = (float)((beta ? (double)beta[c] : 0.0) - mean * sc);
        moments[(int64_t)idx * 2 + 0] = mean;
        moments[(int64_t)idx * 2 + 1] = count > 1.0 ? var * count / (count - 1.0) : var;
    }
    __syncthreads();   // one workgroup: the moments are visible, and every thread has read the old count
    if (running_mean && running_var) {
        for (int c = threadIdx.x; c < channels; c += blockDim.x) {
            double rm = (double)running_mean[c], rv = (double)running_var[c];
            for (int bt = 0; bt < batches; ++bt) {
                double factor = momentum;
                if (momentum < 0.0) factor = num_batches_tracked ? 1.0 / (double)(seen + bt + 1) : 0.0;
                const double* m = moments + ((int64_t)bt * channels + c) * 2;
                // the module stores fp32 after every forward: round like it does, batch by batch
                rm = (double)(float)((1.0 - factor) * rm + factor * m[0]);
                rv = (double)(float)((1.0 - factor) * rv + factor * m[1]);
            }
            running_mean[c] = (float)rm;
            running_var[c] = (float)rv;
        }
    }
    if (threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += batches;
}

static inline int bn_splits(int64_t n, int channels) {
    // enough workgroups to fill 256 CUs a few times over without splitting a sample's slab
    int64_t s = ceil_div(1024, channels);
    return (int)std::max<int64_t>(1, std::min<int64_t>(s, n));
}

}  // namespace pleas

using namespace pleas;

extern "C" size_t pleas_bn_train_ws_bytes(int64_t n, int channels) {
    if (n <= 0 || channels <= 0) return 0;
    return (size_t)(bn_splits(n, channels) + 1) * channels * 2 * sizeof(double);      // S partial slabs + the moments
}

extern "C" int pleas_bn_train_fold_batches(const float* x, int64_t n, int batches, int channels, int64_t inner,
                                           const float* gamma, const float* beta, double eps, double momentum,
                                           float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                           float* scale, float* shift, void* ws, size_t ws_bytes, void* stream_) {
    if (!x || !scale || !shift) return bad_arg("null pointer");
    if (n <= 0 || channels <= 0 || inner <= 0 || batches <= 0) return bad_arg("empty batch");
    if ((running_mean == nullptr) != (running_var == nullptr)) return bad_arg("running_mean / running_var: both or neither");
    if (n >= ((int64_t)1 << 31) || channels > 65535 || batches > 65535) return bad_arg("tensor too large");
    if (!ws || ws_bytes < (size_t)batches * pleas_bn_train_ws_bytes(n, channels)) return PLEAS_ENOMEM;
    if ((uintptr_t)ws & 7) return bad_arg("workspace must be 8-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const int S = bn_splits(n, channels);
    const int vec = (inner % 4 == 0 && ((uintptr_t)x & 15) == 0) ? 1 : 0;
    ProfScope prof(kProfBnAct, 0.0, (double)batches * n * channels * inner * sizeof(float), stream);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(channels, S, batches), dim3(kBnThreads), 0, stream, x, (int)n, channels,
                       inner, vec, (double*)ws);
    PLEAS_LAUNCH_CHECK("bn_stats_partial_kernel");
    hipLaunchKernelGGL(bn_train_finalize_kernel, dim3(1), dim3(1024), 0, stream, (double*)ws, S, channels, batches,
                       (double)n * (double)inner, gamma, beta, eps, momentum, running_mean, running_var,
                       num_batches_tracked, scale, shift);
    PLEAS_LAUNCH_CHECK("bn_train_finalize_kernel");
    return PLEAS_OK;
}

extern "C" int pleas_bn_train_fold(const float* x, int64_t n, int channels, int64_t inner, const float* gamma,
                                   const float* beta, double eps, double momentum, float* running_mean,
                                   float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                                   void* ws, size_t ws_bytes, void* stream_) {
    return pleas_bn_train_fold_batches(x, n, 1, channels, inner, gamma, beta, eps, momentum, running_mean, running_var,
                                       num_batches_tracked, scale, shift, ws, ws_bytes, stream_);
}

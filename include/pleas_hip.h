/*
 * pleas_hip.h -- C ABI of libpleas_hip.so: the MI355X (gfx950) kernels behind the
 * PLeaS activation-matching + least-squares merging hot path.
 *
 * Conventions (all entry points):
 *   - plain C types only; every data pointer is a DEVICE pointer borrowed for the
 *     duration of the call unless marked HOST; nothing is retained after return;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and the call returns without synchronising;
 *   - no hidden device allocation: scratch comes from a caller-sized workspace
 *     (`*_ws_bytes` tells how much);
 *   - return value: 0 on success, negative PLEAS_E* on error; never throws.
 *
 * The reference has no native code: each entry point replaces the
 * vendor-library kernels that the cited reference Python line dispatches to.
 */
#ifndef PLEAS_HIP_H
#define PLEAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLEAS_OK 0
#define PLEAS_EINVAL (-22)   /* bad argument (shape, null pointer, unsupported size) */
#define PLEAS_ENOMEM (-12)   /* workspace too small */
#define PLEAS_EHIP (-5)      /* a HIP runtime call failed (see pleas_last_error) */

/* Library / build identification: "pleas_hip <version> gfx950". */
const char* pleas_version(void);
/* Text of the last HIP error seen by this thread's calls ("" if none). */
const char* pleas_last_error(void);

/* ------------------------------------------------------------------------------------
 * Cross features: Gram / negative Euclidean distance between the channels of two
 * activation (or weight) tensors, accumulated into a C x C matrix.
 *
 * Replaces: pleas/methods/activation_matching.py:31-46 (cross_features_cdist:
 *   movedim+reshape copies, torch.cdist mm-path) and :14-28 (cross_features_inner_product),
 *   plus the per-batch / per-node accumulation at :123-127 and :129-134.
 *
 * x, y: contiguous fp32 tensors viewed as [B][C][HW] (for NCHW activations: B = batch,
 *   HW = H*W; for axis `a` of any contiguous tensor: B = prod(shape[:a]), C = shape[a],
 *   HW = prod(shape[a+1:])).  The contraction runs over all (b, hw).
 * epilogue: PLEAS_EPI_INNER  -> v = sum_k x_ik y_jk
 *           PLEAS_EPI_NEG_CDIST -> v = -sqrt(max(0, |x_i|^2 + |y_j|^2 - 2 sum_k x_ik y_jk))
 * accumulate: 0 -> acc = v ; 1 -> acc += v           (acc: C x C fp32, row-major)
 * ws: workspace of at least pleas_gram_ws_bytes(B, C, HW) bytes.
 * Deterministic: a fixed (B, C, HW) always reduces in the same order.
 */
#define PLEAS_EPI_INNER 0
#define PLEAS_EPI_NEG_CDIST 1
size_t pleas_gram_ws_bytes(int B, int C, int64_t HW);
int pleas_gram_accum(const float* x, const float* y, int B, int C, int64_t HW, int epilogue, int accumulate,
                     float* acc, void* ws, size_t ws_bytes, void* stream);

/* Grouped form: ALL tracked nodes of one batch in one contraction grid plus one reduce grid.
 *
 * Replaces one whole iteration of the hot loop at activation_matching.py:120-127 plus the
 * per-group sum at :129-134: node i contributes epilogue(x_i, y_i) to the matrix of its group.
 * nodes[i]      : HOST array; x, y as in pleas_gram_accum ([B][C][HW] views), group index
 * group_acc[g]  : DEVICE C_g x C_g fp32 matrices (HOST array of pointers); group_C[g] HOST sizes
 * accumulate    : 0 -> group matrices are overwritten by this batch's sum, 1 -> added to
 * Work is cut into (node, tile, K-range) items of near-equal length, sorted longest first; each
 * node's partial products go to its own slab in `ws`, and the reduce grid sums slabs and nodes in a
 * FIXED order (deterministic, no atomics).  ws >= pleas_gram_batch_ws_bytes(...).
 * The work list is cached per (shape sequence, ws, group matrices) and its device tables live at the
 * start of `ws`: pass ws_fresh = 1 on the first call with a workspace (or whenever its content may
 * have been overwritten since the previous call), 0 otherwise; then only operand pointers are sent.
 */
typedef struct pleas_gram_node {
    const float* x;
    const float* y;
    int B;
    int C;
    int64_t HW;
    int group;
    /* Derived node (derived != 0): its operands are per-channel affine images of node `source`'s operands,
     *   x' = scale_x[c] * x + shift_x[c],  y' = scale_y[c] * y + shift_y[c]   (DEVICE arrays of length C)
     * -- an eval-mode BatchNorm of a tracked convolution output.  Its inner products and norms follow from the
     * source's products, squared norms and row sums, so it is NOT contracted: x / y are ignored, `source` must be a
     * contracted node of the same shape in this list.  Zero-initialised fields = an ordinary node. */
    int derived;
    int source;
    const float* scale_x;
    const float* shift_x;
    const float* scale_y;
    const float* shift_y;
} pleas_gram_node;
size_t pleas_gram_batch_ws_bytes(const pleas_gram_node* nodes, int n_nodes, const int* group_C, int n_groups);
int pleas_gram_batch(const pleas_gram_node* nodes, int n_nodes, float* const* group_acc, const int* group_C,
                     int n_groups, int epilogue, int accumulate, void* ws, size_t ws_bytes, int ws_fresh, void* stream);

/* ------------------------------------------------------------------------------------
 * Batched linear assignment (square, dense), one workgroup per problem.
 *
 * Replaces: pleas/core/solvers.py:18-33 (scipy_solve_lsa -> D2H copy +
 *   scipy.optimize.linear_sum_assignment), called at activation_matching.py:173 and
 *   weight_matching.py:78.
 *
 * cost[p]   : DEVICE pointer to an n[p] x n[p] row-major fp32 matrix (HOST array of pointers)
 * n[p]      : HOST array, 1 <= n[p] <= PLEAS_LSAP_MAX_N
 * col_ind[p]: DEVICE pointer to n[p] int64 (HOST array of pointers); on completion row i is
 *             assigned to column col_ind[p][i]
 * Arithmetic: costs are widened to fp64; duals and path lengths are fp64; the scan order and
 * tie rule are those of scipy's shortest-augmenting-path solver, so the output is the same
 * assignment scipy returns, including on degenerate (tied) inputs.
 */
#define PLEAS_LSAP_MAX_N 4096
int pleas_lsap_batched(const float* const* cost, const int* n, int nprob, int maximize, int64_t* const* col_ind,
                       void* stream);
/* The same solver for ONE problem whose cost matrix is in HOST memory (fp32 when is_double == 0, else fp64; n >= 1, no
 * upper limit), synchronous, no GPU involved: the explicit host entry point for callers that hold host data
 * (weight matching of CPU state dicts: reference weight_matching.py:78 with CPU tensors, BASELINE.json configs[0]).
 * Same scan order and tie rule, hence the same col_ind as scipy.optimize.linear_sum_assignment. */
int pleas_lsap_host(const void* cost, int is_double, int n, int maximize, int64_t* col_ind);

/* ------------------------------------------------------------------------------------
 * Block gather / average used by partial merging and by the PLeaS regression targets.
 *
 * Replaces: pleas/methods/partial_matching.py:122-129 (1-axis tensors) and :157-172
 *   (2-axis block matrix), and pleas/methods/pleas_merging.py:116-147 (index_select x8 + cat).
 *
 * Tensors are viewed as [outer][rows][cols][inner] (contiguous).  For every output
 * (o, r, c, i):   out = coef(r) * ( [row1[r]>=0 && col1[c]>=0] w1[o, row1[r], col1[c], i]
 *                                 + [row2[r]>=0 && col2[c]>=0] w2[o, row2[r], col2[c], i] )
 * with coef(r) = 0.5 for r < n_merged_rows and 1 otherwise.  row1/row2 (length rows_out) and
 * col1/col2 (length cols_out) are DEVICE int32 maps with -1 = "absent"; col1 == NULL means
 * cols are passed through unchanged (cols_out must equal cols_src).
 */
int pleas_merge_blocks(const float* w1, const float* w2, float* out, int64_t outer, int rows_out, int cols_out,
                       int64_t inner, int rows_src, int cols_src, const int32_t* row1, const int32_t* row2,
                       const int32_t* col1, const int32_t* col2, int n_merged_rows, void* stream);

/* Grouped form: the 1-axis block merge of MANY tensors (the merged layer inputs of one PLeaS update,
 * pleas_merging.py:125-147) in ONE launch.  items: HOST array; tensor pointers and maps are DEVICE pointers.
 * Tensors are viewed [outer][rows][inner]; semantics per tensor as pleas_merge_blocks without column maps.
 * Tensors with inner % 4 == 0 must be 16-byte aligned.  ws / ws_fresh as in pleas_gram_batch.
 * sub_stride > 1 (the input of a 1x1 convolution with that stride, which reads every sub_stride-th pixel of every
 * sub_stride-th line): the sources are images of sub_h x sub_w pixels (inner_src = sub_h * sub_w) and `out` holds only the
 * pixels the layer reads, inner = ceil(sub_h / sub_stride) * ceil(sub_w / sub_stride) -- the layer then is a dense 1x1
 * stride-1 convolution for pleas_fwd_batch / pleas_wgrad_batch.  0 or 1: inner is the sources' inner size. */
typedef struct pleas_merge_item {
    const float* w1;      /* [outer][rows_src][inner] */
    const float* w2;
    float* out;           /* [outer][rows_out][inner] */
    const int32_t* row1;  /* [rows_out], -1 = absent */
    const int32_t* row2;
    int64_t outer, inner;
    int rows_out, rows_src, n_merged;
    int sub_stride, sub_h, sub_w;
} pleas_merge_item;
size_t pleas_merge_batch_ws_bytes(const pleas_merge_item* items, int n_items);
int pleas_merge_batch(const pleas_merge_item* items, int n_items, void* ws, size_t ws_bytes, int ws_fresh, void* stream);

/* ------------------------------------------------------------------------------------
 * Frozen-source forward: inference BatchNorm (+ residual add) (+ ReLU) in one pass.
 *
 * Replaces: the BatchNorm2d -> (out += identity) -> ReLU module chain inside `model1(x); model2(x)`
 *   (pleas/methods/pleas_merging.py:267-268, models put in eval() at :352-353): three vendor
 *   kernels and seven tensor passes per chain become one kernel, one read (two with a residual)
 *   and one write.
 *
 * x, res (nullable), y: [n][channels][inner] contiguous fp32;  scale[c] = weight[c] / sqrt(var[c] + eps),
 * shift[c] = bias[c] - mean[c] * scale[c]  (DEVICE, prepared once by the caller).
 *   y = x * scale[c] + shift[c] (+ res);   relu != 0 applies max(., 0).   y must not alias x.
 */
int pleas_bn_act(const float* x, const float* scale, const float* shift, const float* res, float* y, int64_t n,
                 int channels, int64_t inner, int relu, void* stream);
/* Same pass with the intermediate values kept, for activation matching (activation_matching.py:49-100 measures EVERY
 * node of the chain): y_bn = x * scale + shift and y_sum = y_bn + res are written when non-NULL, y is the final value. */
int pleas_bn_act_tracked(const float* x, const float* scale, const float* shift, const float* res, float* y_bn,
                         float* y_sum, float* y, int64_t n, int channels, int64_t inner, int relu, void* stream);

/* The tracked pass on a tensor that holds `batches` batches back to back along n, each with its OWN affine map: scale /
 * shift are [batches][channels] (pleas_bn_train_fold_batches).  Replaces the same module chains when the twin forward of
 * activation_matching.py:49-100 carries several matching batches and the models are in train mode (batch statistics are
 * per batch: run_domainnet.py:172-186 never calls .eval()). */
int pleas_bn_act_tracked_batches(const float* x, const float* scale, const float* shift, const float* res, float* y_bn,
                                 float* y_sum, float* y, int64_t n_per_batch, int batches, int channels, int64_t inner,
                                 int relu, void* stream);

/* The same map followed by a max pooling window, one pass (a ResNet's stem: bn1 -> relu -> maxpool).
 *
 * Replaces: BatchNorm2d -> ReLU -> nn.MaxPool2d(kernel, stride, padding) of a frozen source forward
 *   (pleas/methods/pleas_merging.py:267-268 runs the whole model; only Conv2d / Linear inputs and outputs are hooked,
 *   :197-243, so the full-resolution activation between ReLU and the pooling is consumed by nothing and is not written).
 *   y[n][c][oh][ow] = max over the KH x KW window at (oh * stride - pad, ow * stride - pad) of act(x * scale[c] + shift[c]);
 *   out-of-range taps never win (torch pads with -inf), a NaN tap does; floor-mode output size, dilation 1.
 *   scale == NULL: plain max pooling of x.  y: [n][channels][Ho][Wo], must not alias x. */
int pleas_bn_act_maxpool(const float* x, const float* scale, const float* shift, float* y, int64_t n, int channels, int H,
                         int W, int KH, int KW, int stride, int pad, int relu, void* stream);

/* Train-mode BatchNorm of the matching forward, folded to the same per-channel affine map.
 *
 * Replaces: the BatchNorm2d modules inside the cross module's forward when the caller's models are in train mode --
 *   which is what both reference drivers do: no .eval() before activation_matching
 *   (experiments/shared_label_space/run_domainnet.py:172-186, :257-264; pleas/methods/activation_matching.py:119-122
 *   never touches the mode) -- i.e. torch.nn.functional.batch_norm(training=True): batch statistics for the
 *   normalisation, running statistics updated as a side effect.
 *
 * One streaming pass over x [n][channels][inner] (fp64 accumulation, deterministic two-stage reduce) gives, per channel,
 *   mean, var_b (biased);  scale = gamma / sqrt(var_b + eps);  shift = beta - mean * scale     (gamma / beta NULL = 1 / 0)
 * and, when running_mean / running_var are non-NULL,
 *   running <- (1 - f) running + f {mean, var_b * count / (count - 1)},   f = momentum, or 1 / (batches so far + 1) when
 *   momentum < 0 (torch's momentum=None);   *num_batches_tracked += 1 when non-NULL.
 * The chain's values then come from pleas_bn_act_tracked(x, scale, shift, ...) exactly as in eval mode.
 * ws: >= pleas_bn_train_ws_bytes(n, channels) bytes, 8-byte aligned, caller-owned.
 */
size_t pleas_bn_train_ws_bytes(int64_t n, int channels);
int pleas_bn_train_fold(const float* x, int64_t n, int channels, int64_t inner, const float* gamma, const float* beta,
                        double eps, double momentum, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* scale, float* shift, void* ws, size_t ws_bytes,
                        void* stream);
/* `batches` batches of n samples each, back to back along the sample axis: every batch is folded on its own samples, IN
 * ORDER -- scale / shift are [batches][channels]; running statistics and counter end where `batches` successive forwards
 * of the module leave them.  ws: batches * pleas_bn_train_ws_bytes(n, channels) bytes. */
int pleas_bn_train_fold_batches(const float* x, int64_t n, int batches, int channels, int64_t inner, const float* gamma,
                                const float* beta, double eps, double momentum, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, float* scale, float* shift, void* ws, size_t ws_bytes,
                                void* stream);

/* ------------------------------------------------------------------------------------
 * Fused masked Adam step over a flat parameter arena.
 *
 * Replaces: pleas/methods/pleas_merging.py:288-291 (`param.grad *= mask` for every
 *   parameter, then torch.optim.Adam.step with default betas/eps, no weight decay).
 *
 * g <- g * mask (mask may be NULL = all ones); m <- m + (1-b1)(g - m); v <- b2 v + (1-b2) g g;
 * p <- p - (lr / (1 - b1^step)) * m / (sqrt(v) / sqrt(1 - b2^step) + eps).   step counts from 1.
 */
int pleas_masked_adam(float* p, const float* g, const float* mask, float* m, float* v, int64_t n, float lr,
                      float b1, float b2, float eps, int step, void* stream);

/* ------------------------------------------------------------------------------------
 * Squared-error reduction: out[0] (+)= scale * sum_k (a[k] - b[k])^2 ; also writes
 * diff[k] = dscale * (a[k] - b[k]) if diff != NULL.
 *
 * Replaces: pleas/methods/pleas_merging.py:282 (`((out - target)**2).mean()`) and the
 *   first node of its autograd backward (2 (out - target) / numel).
 * ws: at least pleas_sqerr_ws_bytes(n) bytes.  Deterministic two-stage reduction.
 */
size_t pleas_sqerr_ws_bytes(int64_t n);
int pleas_sqerr(const float* a, const float* b, int64_t n, float scale, int accumulate, float* out, float dscale,
                float* diff, void* ws, size_t ws_bytes, void* stream);

/* Bias gradient of a merged layer: out[c] = sum_{n, p} x[n][c][p] for x = the layer's residual [N][C][HW] (Linear: HW = 1).
 * Replaces the bias node of autograd's backward (pleas_merging.py:287).  Deterministic (fixed order, no atomics). */
int pleas_channel_sum(const float* x, int N, int C, int64_t HW, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * PLeaS layer fitting: fused target/residual/loss pass and grouped weight-gradient launch.
 *
 * Replaces, for merging='perm_gradmask': the output half of get_model_orig_activations
 *   (pleas_merging.py:116-123, :147), the per-layer MSE (:282) and the whole autograd backward
 *   of the merged layers (:287).
 *
 * pleas_target_residual: out = merged layer output [N][C][HW]; o1, o2 = source layer outputs
 *   [N][Csrc][HW]; row1/row2/n_merged = block maps as in pleas_merge_blocks.
 *   resid = dscale * (out - target)   (resid may alias out);   partials[0..*n_partials) receive this
 *   call's per-workgroup sums of (out - target)^2 (*n_partials <= pleas_target_residual_max_partials()).
 * pleas_loss_final: loss[l] = scale[l] * sum(partials[l*stride .. +n_partials[l]))  (all DEVICE arrays).
 * pleas_wgrad_batch: for every layer  grad[co][ci][kh][kw] = sum_{n,oh,ow} resid[n][co][oh][ow] *
 *   ip[n][ci][oh*stride+kh-pad][ow*stride+kw-pad]  (a Linear layer is Hin=Win=KH=KW=1).
 *   layers: HOST array; resid/ip/grad DEVICE pointers (16-byte aligned); ws >= pleas_wgrad_batch_ws_bytes.
 *   Deterministic; the work list is cached per (geometry sequence, ws); ws_fresh as in pleas_gram_batch.
 */
/* pleas_fwd_batch: forward of every merged layer of one update with the target, residual and loss fused:
 *   out   = conv(ip, w) (+ bias)                      (never stored)
 *   tgt   = coef(co) * ([row1[co]>=0] o1[n][row1[co]] + [row2[co]>=0] o2[n][row2[co]]),  coef = 0.5 for co < n_merged
 *   resid = dscale * (out - tgt)      -> layers[i].resid  (the input of pleas_wgrad_batch)
 *   loss[i] = loss_scale * sum (out - tgt)^2           (DEVICE float[n_layers], fixed summation order)
 * A Linear layer is Hin = Win = KH = KW = 1.  w and ip must be 16-byte aligned.  ws / ws_fresh as in pleas_gram_batch.
 * Stride-1 "same" convolutions with Cin % 32 == 0 (k x k: with PLEAS_FWD_KPOS_MAJOR) take the flat-shift tile forms; the
 * forms of a call run as separate grids on the caller's stream and up to three side streams of the library, joined
 * before the call's last kernel (fwd_loss) is enqueued on `stream`.
 */
typedef struct pleas_fwd_layer {
    const float* ip;     /* [N][Cin][Hin][Win] merged input */
    const float* w;      /* [Cout][Cin][KH][KW], or [Cout][KH][KW][Cin] with PLEAS_FWD_KPOS_MAJOR */
    const float* bias;   /* [Cout] or NULL */
    const float* o1;     /* [N][Csrc][Hout*Wout] source-layer outputs */
    const float* o2;
    const int32_t* row1; /* [Cout] block maps (DEVICE) */
    const int32_t* row2;
    float* resid;        /* [N][Cout][Hout*Wout] */
    int N, Cout, Cin, Hin, Win, KH, KW, stride, pad, Csrc, n_merged;
    float dscale, loss_scale;
    int flags;           /* PLEAS_FWD_* */
} pleas_fwd_layer;
/* Weights stored kernel-position-major (the layout pleas_wgrad_batch writes with PLEAS_WGRAD_KPOS_MAJOR); needs
 * Cin % 32 == 0.  Every 32-deep K chunk then has ONE tap: per-thread addressing is as cheap as for a 1x1 layer and
 * the taps of a channel block re-read the same input rows back to back. */
#define PLEAS_FWD_KPOS_MAJOR 1
size_t pleas_fwd_batch_ws_bytes(const pleas_fwd_layer* layers, int n_layers);
/* Diagnostics of the latest pleas_fwd_batch plan: per tile form (10 entries each) the duration measured on its lane by
 * the plan's calibration launch [ms] (0 before it), the lane it runs on (0 = the caller's stream) and its work items.
 * Returns 0 = lanes still dealt from static weights, 1 = calibration launch in flight, 2 = lanes dealt from measurements. */
int pleas_fwd_plan_lanes(double* form_ms, int* form_lane, int* form_items);
/* Host only (no GPU): the launch units of the grouped forward of `layers` -- one kernel launch each: a tile form over a
 * contiguous slice of that form's work items, on a lane -- as 4 ints per unit (form, first item, items, lane) in launch
 * order.  form_ms == NULL: the units of a fresh plan (one per form, lanes from static weights); else the units after a
 * calibration launch that measured form_ms[10] milliseconds per form (long forms cut into slices of equal work).
 * Returns the number of units (> max_units: only the first max_units were written) or a negative error code. */
int pleas_fwd_plan_units(const pleas_fwd_layer* layers, int n_layers, const double* form_ms, int* units, int max_units);
int pleas_fwd_batch(const pleas_fwd_layer* layers, int n_layers, float* loss, void* ws, size_t ws_bytes, int ws_fresh,
                    void* stream);
/* pleas_conv2d_fwd: a plain convolution  y = conv(x, w) (+ bias)  on the tile forms of pleas_fwd_batch (fp32 MFMA 32x32x2,
 * fixed summation order: bit-for-bit repeatable, unlike MIOpen's split-K 3 x 3 kernels).  Replaces the vendor convolution under
 * the frozen sources' k x k layers -- the two `model(x)` calls of pleas/methods/pleas_merging.py:267-268 and of the twin
 * graph, activation_matching.py:49-100 -- so that the same job returns the same assignment and weights run after run.
 *   x [N][Cin][Hin][Win], w [Cout][Cin][KH][KW] (or [Cout][KH][KW][Cin] with PLEAS_FWD_KPOS_MAJOR), y [N][Cout][Hout][Wout];
 *   x and w 16-byte aligned; square kernels up to 64 taps; no workspace, no tables: the layer rides in the kernel arguments. */
int pleas_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Hin, int Win, int Cout,
                     int KH, int KW, int stride, int pad, int flags, void* stream);
/* pleas_conv2d_bn_act_fwd: the same convolution with the eval-mode BatchNorm / residual add / ReLU that follows it in a frozen
 * source (torchvision Bottleneck: conv -> bn -> relu, conv3 -> bn3 -> (+ identity) -> relu, downsample conv -> bn) in its
 * epilogue:  y = conv(x, w) (+ bias)  is stored for the hooks of pleas_merging.py:197-231 (the layer's OUTPUT is a regression
 * target), and  z = act(y * scale[c] + shift[c] (+ res))  -- bit for bit what pleas_bn_act makes of y -- is stored beside it
 * from the registers, so y is not read back.  res (or NULL) and z are shaped like y; y, z, res 16-byte aligned. */
int pleas_conv2d_bn_act_fwd(const float* x, const float* w, const float* bias, float* y, const float* scale,
                            const float* shift, const float* res, float* z, int relu, int N, int Cin, int Hin, int Win,
                            int Cout, int KH, int KW, int stride, int pad, int flags, void* stream);
typedef struct pleas_wgrad_layer {
    const float* resid; /* [N][Cout][Hout*Wout] */
    const float* ip;    /* [N][Cin][Hin][Win]   */
    float* grad;        /* [Cout][Cin][KH][KW]  */
    int N, Cout, Cin, Hin, Win, KH, KW, stride, pad;
    int flags;          /* PLEAS_WGRAD_* */
} pleas_wgrad_layer;
#define PLEAS_WGRAD_ACCUMULATE 1  /* grad += instead of grad = (normal equations: B^T over many batches) */
#define PLEAS_WGRAD_KPOS_MAJOR 2  /* write grad as [Cout][KH*KW][Cin] (kernel-position-major columns) */
int pleas_target_residual(const float* out, const float* o1, const float* o2, const int32_t* row1, const int32_t* row2,
                          int n_merged, int N, int C, int Csrc, int64_t HW, float dscale, float* resid, float* partials,
                          int* n_partials, void* stream);
int pleas_target_residual_max_partials(void);
int pleas_loss_final(const float* partials, const int* n_partials, const float* scale, int stride, int n_layers,
                     float* loss, void* stream);
size_t pleas_wgrad_batch_ws_bytes(const pleas_wgrad_layer* layers, int n_layers);
int pleas_wgrad_batch(const pleas_wgrad_layer* layers, int n_layers, void* ws, size_t ws_bytes, int ws_fresh,
                      void* stream);

/* ------------------------------------------------------------------------------------
 * The path's exchange step under data parallelism (no reference counterpart -- the reference is
 * single-GPU): buf[0..n) := sum over the ranks of `comm`, in place, ONE RCCL all-reduce (fp32, sum) enqueued on `stream`.
 * What is exchanged: the flat cost arena of activation matching (all group matrices back to back, ResNet-101: 41 MB) before
 * the LAPs; the A / B arenas of the closed form before the solve; the gradient arena of an Adam update.
 * `comm` is the CALLER's ncclComm_t (RCCL is resolved at run time from the instance already loaded in the process, else
 * $PLEAS_RCCL_LIB, else librccl.so.1 -- a communicator is only valid in the library instance that created it).
 * The Python host side uses torch.distributed (backend "nccl" = RCCL) for the same step. */
int pleas_allreduce_sum(float* buf, int64_t n, void* comm, void* stream);

/* ------------------------------------------------------------------------------------
 * Normal equations of the per-layer PLeaS objective (closed form of what the reference's Adam
 * loop, pleas_merging.py:281-291 + :357-358, approximates):  W^T = A^-1 B  with
 *   A = sum_batches U^T U   (K x K, K = KH*KW*Cin, index k = (kh*KW + kw)*Cin + ci)
 *   B^T = sum_batches op . U  -> pleas_wgrad_batch(resid := op, flags = ACCUMULATE | KPOS_MAJOR).
 * pleas_normal_eq_accum adds one batch's U^T U of EVERY listed layer into its A (LOWER triangle of
 * 64/128-wide block tiles only -- see pleas_normal_eq_finalize for the blocks it leaves to that call; the strict
 * upper part of A is left untouched), one grouped
 * fp32-MFMA launch, U = im2col(ip) never materialised.  ws / ws_fresh as in pleas_gram_batch.
 */
typedef struct pleas_neq_layer {
    const float* ip; /* [N][Cin][Hin][Win] merged layer input */
    float* A;        /* [K][K] fp32, accumulated in place */
    int N, Cin, Hin, Win, KH, KW, stride, pad;
} pleas_neq_layer;
size_t pleas_normal_eq_ws_bytes(const pleas_neq_layer* layers, int n_layers);
int pleas_normal_eq_accum(const pleas_neq_layer* layers, int n_layers, void* ws, size_t ws_bytes, int ws_fresh,
                          void* stream);
/* Stride-1 "same" k x k layers: the block of A between two kernel positions depends only on their LAG and on which
 * border rows / columns the pair can reach, so the 45 lower-triangle blocks of a 3x3 layer hold 29 distinct matrices
 * (up to transposition).  pleas_normal_eq_accum contracts only those; pleas_normal_eq_finalize copies / transposes
 * them into the remaining blocks of the lower triangle.  Call it ONCE after the last batch (after the all-reduce of A
 * in a multi-GPU run) and before the solve; it overwrites (idempotent).  Layers of other geometries are left alone.
 * pleas_normal_eq_plan_info (host only, no GPU): info[0] = the path's flops per call (K^2 * N*HWo per layer),
 * info[1] = flops the grid executes, info[2] = work items, info[3] = blocks left to finalize. */
int pleas_normal_eq_finalize(const pleas_neq_layer* layers, int n_layers, void* stream);
int pleas_normal_eq_plan_info(const pleas_neq_layer* layers, int n_layers, double* info);

/* Batched SPD solve of the normal equations (blocked Cholesky + forward/back substitution, panel 64,
 * trailing updates on fp32 MFMA tiles).  For every problem p:  X (A_p + lambda*mean(diag A_p) I) = Bt_p,
 * i.e. each of the N_p rows of Bt_p (length K_p) is one right-hand side and is overwritten by its
 * solution.  A_p: K_p x K_p row-major, LOWER triangle read; overwritten (L below the diagonal, L^T
 * above).  A, Bt, K, N: HOST arrays (of DEVICE pointers / sizes); info: DEVICE int[nprob], 0 = ok,
 * j > 0 = pivot j not positive in fp32 (the caller should redo that problem in higher precision).
 * Problems of different sizes share the launches: cost = 5 * max(K)/64 launches per 96 problems.
 */
int pleas_cholesky_solve_batched(float* const* A, float* const* Bt, const int* K, const int* N, int nprob, float lambda,
                                 int* info, void* stream);

/* ------------------------------------------------------------------------------------
 * Opt-in live timing of the library's kernels with HIP events recorded on the launch stream
 * (used by bench.py for the roofline figure; off by default, no cost when off).
 * kernel ids: 0 gram_partial, 1 gram_finalize, 2 lsap, 3 merge_blocks, 4 masked_adam, 5 sqerr,
 *             6 conv_fwd, 7 conv_wgrad, 8 normal_eq, 9 solve, 10 bn_act.
 * pleas_prof_collect waits for the recorded events of that kernel and returns the number of
 * launches, their summed duration, and the summed ALGORITHMIC flops / bytes of those launches.
 */
void pleas_prof_enable(int on);
void pleas_prof_select(unsigned kernel_mask); /* bit k: record kernel id k (default all); many-launch kernels can be left out */
void pleas_prof_reset(void);
int pleas_prof_collect(int kernel, int64_t* launches, double* total_ms, double* flops, double* bytes);

/* Tuning hook for experiments: split-K target workgroup count and minimum K chunks per split. */
void pleas_gram_tune(int target_blocks, int min_chunks_per_split);
void pleas_gram_batch_tune(int item_chunks, int xcd_order);
/* Arithmetic of the contraction kernels (matching contraction, grouped forward, weight gradient, plain convolution).
 *   PLEAS_ARITH_FP32       (default) exact fp32 MFMA, v_mfma_f32_32x32x2_f32: bitwise an fmaf chain;
 *   PLEAS_ARITH_SPLIT_BF16 every fp32 operand as the exact sum of three bf16 values, six v_mfma_f32_32x32x16_bf16 products per
 *                          k step with fp32 accumulation (the three dropped terms are below half an fp32 ulp of the product):
 *                          fp32 accuracy at 2.67x less matrix-pipe time.  Tile forms without a split variant (scalar-load
 *                          forms: 7 x 7 images, strided layers, the stem) keep the exact arithmetic inside the same launch.
 * Process-wide; plans are keyed by it.  PLEAS_ARITH=split_bf16 in the environment sets the initial value.  The headline
 * numbers and every default-path test run with PLEAS_ARITH_FP32; bench.py reports the other as `alt_arith`. */
#define PLEAS_ARITH_FP32 0
#define PLEAS_ARITH_SPLIT_BF16 1
void pleas_arith(int mode);
int pleas_arith_get(void);
/* round-3 name of pleas_arith(on ? PLEAS_ARITH_SPLIT_BF16 : PLEAS_ARITH_FP32) */
void pleas_gram_split_bf16(int on);
void pleas_wgrad_tune(int item_chunks);

#ifdef __cplusplus
}
#endif
#endif /* PLEAS_HIP_H */

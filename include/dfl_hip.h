/*
 * dfl_hip.h -- C ABI of libdfl_hip.so: the MI355X (gfx950) implementation of the U-Net hot path of
 * rg2/DeepFluoroLabeling-IPCAI2020 (train_test_code/unet.py, dice.py, ncc.py, util.py ensemble).
 *
 * The reference has no FFI layer: its hot path is torch.nn calls made from Python.  Each entry
 * point below replaces one group of those calls and cites it (paths relative to the reference repo root).
 *
 * Conventions
 *   - every function returns 0 on success or a negative dfl_status; it never throws and never allocates or
 *     frees device memory: the caller owns every buffer and passes raw device pointers;
 *   - work is enqueued asynchronously on the given hipStream_t (pass the host framework's current stream);
 *   - re-entrant; no global mutable state except the thread-local last-error string;
 *   - arithmetic type: fp32 (exact-f32 MFMA v_mfma_f32_32x32x2_f32 for the contractions, fp64 for the
 *     cross-block part of the statistics/loss reductions); dfl_set_math_mode selects bf16-product variants, and
 *     math mode 4 ("bf16 storage") keeps the network's internal activations, their gradients and the GEMM
 *     copies of the weights as bf16 in HBM: argument blocks then carry *_bf16 flags, the flagged pointers address
 *     bf16 elements (declared `float*` / `const float*` here for the common case) and their ld* count elements;
 *   - ACTIVATION LAYOUT inside the network is NHWC ("pixel-major"): element (n,y,x,c) of a tensor with pixel
 *     stride ld (in floats, ld >= C) lives at ((n*H + y)*W + x)*ld + c.  A channel slice of a wider buffer is
 *     expressed by offsetting the pointer and keeping ld (this is how torch.cat in unet.py:256-257 is made
 *     free).  The network input (C = 1) and the two outputs (seg / heat maps) are plain NCHW as in the
 *     reference, so the layout is invisible at the boundary.
 */
#ifndef DFL_HIP_H
#define DFL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dfl_stream_t; /* hipStream_t */

typedef enum {
  DFL_OK = 0,
  DFL_ERR_INVALID_ARG = -1,
  DFL_ERR_UNSUPPORTED = -2,
  DFL_ERR_LAUNCH = -3,
  DFL_ERR_WORKSPACE = -4
} dfl_status;

/* Library version (major*10000 + minor*100 + patch) and the last error message of the calling thread. */
int dfl_version(void);
const char* dfl_last_error(void);
/* sizeof() of the argument structs, in declaration order (conv, wgrad, pack_job, bn_finalize, colstats,
 * bn_bwd_finalize, bn_relu_bwd, affine_copy, pool, head_fwd, head_bwd, loss, ensemble, op): lets a binding written
 * in another language verify its struct mirrors at load time.  Returns -1 past the end. */
int dfl_sizeof(int which);

/* ------------------------------------------------------------------------------------------------------------
 * Convolution as a gather-GEMM on the fp32 matrix cores.
 *   y[m, n] = epilogue( sum_{t < KH*KW} sum_{c < Cin} X(m, t, c) * W(k = t*Cin + c, n) )
 * m runs over the N*Hout*Wout output pixels; X(m,t,c) is the input at pixel (oy*stride - pad + t/KW,
 * ox*stride - pad + t%KW), channel c, after the optional per-channel affine in_scale/in_shift (BatchNorm
 * applied on load), and 0 outside the image (zero padding is applied AFTER the affine, as the reference pads
 * the BatchNorm output).
 * Replaces nn.Conv2d 3x3 (unet.py:211,218), 1x1 residual (unet.py:207,229-231), 2x2/stride-2 down-sampling
 * (unet.py:93,171), nn.ConvTranspose2d(k=2,s=2) (unet.py:240,255; scatter2x2 = 1) and -- with the packed
 * weights of dfl_pack_weights -- every data-gradient of those layers (torch autograd, train.py:422).
 * Epilogue order: + bias[n]; ReLU (unet.py:213,220); + add[m,n]*add_scale[n] + add_shift[n] (the BatchNorm
 * of the block's last conv, summed with the residual: unet.py:229-231); + old y (accumulate); store;
 * per-channel partial sums of v and v*u for BatchNorm (u = v, or u = stat_other[m,n]).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* x;        /* input activations, NHWC, pixel stride ldx */
  const float* w;        /* quad-packed weights [ceil(KH*KW*Cin/4)][Ntot][4] (dfl_pack_weights), 16-byte aligned */
  const float* bias;     /* [Ntot / (scatter2x2 ? 4 : 1)] or NULL */
  const float* in_scale; /* [Cin] or NULL */
  const float* in_shift; /* [Cin] or NULL (required when in_scale is given) */
  const float* add;      /* [M][ldadd] or NULL */
  const float* add_scale;/* [Ntot] or NULL (=> scale 1, shift 0) */
  const float* add_shift;
  const float* stat_other; /* [M][ldso] or NULL (=> u = v) */
  float* y;              /* output, NHWC, pixel stride ldy */
  float* stat_partials;  /* [gridM][2][Ntot] or NULL; gridM = dfl_conv_grid_m(args).  With scatter2x2 (bf16 tensors only):
                            [4*gridM][2][Cout], the sums of y's Cout channels (stat_other is then addressed like y) */
  float* partial;        /* split-K scratch [splits][M][Ntot], required when splits > 1 */
  int32_t N, Hin, Win, Cin, ldx;
  int32_t KH, KW, stride, pad;
  int32_t Hout, Wout, Ntot, ldy;
  int32_t ldadd, ldso;
  int32_t relu;
  int32_t accumulate;
  int32_t scatter2x2;    /* ConvTranspose2d(k2,s2): Ntot = 4*Cout, column ab*Cout+co of input pixel (i,j) is stored
                            at output pixel (2i + ab/2, 2j + ab%2) of an [N, Hout, Wout] image, channel co */
  int32_t splits;        /* 0/1: one pass.  > 1: K is cut into `splits` slices (small-M, long-K layers of the deep
                            U-Net levels), raw sums go to `partial` and a finish kernel applies the epilogue */
  /* "Split" operands (math mode 1, bf16x3, fast path only): the 16-byte slot of 4 consecutive fp32 values holds their
   * 4 hi bf16 followed by their 4 lo bf16 (hi = bf16(v), lo = bf16(v - hi)) -- same addressing, and the kernel copies
   * the slot to LDS instead of splitting it itself (a value is otherwise split once per tap and column block). */
  int32_t w_split;       /* w was packed with dfl_pack_job.split = 1 (2: bf16 chunk layout, required when x_bf16) */
  int32_t x_split;       /* x is a split tensor (written by dfl_bn_relu_bwd_apply with split_out); needs in_scale == NULL */
  /* bf16 tensors (math mode 4).  x_bf16: x is bf16 NHWC (Cin % 16 == 0, ldx % 8 == 0), w is the bf16 chunk layout
   * [ceil(K/16)][Ntot][16] of dfl_pack_job.split = 2; the patch-resident kernels of csrc/convp_bf16.hip run.
   * y_bf16: y, add and stat_other are bf16 (values are rounded once, statistics are taken from the rounded values).
   * The direct small-K kernels (Cin <= 4: the network's first layer) take x_bf16 = 0, y_bf16 = 1. */
  int32_t x_bf16, y_bf16;
  /* BatchNorm + ReLU backward fused into the operand (bf16 patch kernels, x_mode = 1): x is dy, x2 the saved ReLU output r
   * (same shape, pixel stride ldx2) and the convolution's operand is  [r > 0] * (A*dy + B*r + C)  with A, B, C = in_scale[0..C),
   * [C..2C), [2C..3C) -- the coefficients dfl_bn_bwd_finalize leaves (in_scale NULL: plain ReLU backward [r > 0] * dy; in_shift
   * must be NULL).  What dfl_bn_relu_bwd_apply would materialise is never written: one tensor pass and one launch per layer less. */
  const float* x2;
  int32_t ldx2, x_mode;
  /* "Live" BatchNorm statistics (bf16 patch kernels, training; round 4).  Between a convolution and the next one stands
   * nn.BatchNorm2d in training mode (unet.py:214-222): the consumer needs scale / shift made of the producer's batch statistics.
   * dfl_bn_finalize between the two is a 5 us launch in the dependency chain, 44 times per step.  Instead:
   *   stat_totals  the producer ADDS its workgroups' column sums (sum v, sum v*v of the stored values) into [DFL_BN_R][2][Ntot]
   *                doubles with hardware fp64 atomics (row = workgroup % DFL_BN_R spreads the contention; the addends are fp32
   *                values of similar magnitude, their fp64 sum is exact -- independent of the order -- unless one of them is below
   *                2^-29 of the total).  The caller zeroes the rows before the producer runs.  stat_partials may be NULL then.
   *   in_tot       (instead of in_scale / in_shift) the consumer derives scale / shift of its input channels itself, with the
   *                arithmetic of dfl_bn_finalize: mean = s1 / count, var = s2 / count - mean^2 (biased, >= 0),
   *                scale = gamma / sqrt(var + eps), shift = beta - mean * scale, from in_tot [DFL_BN_R][2][Cin], in_gamma, in_beta.
   *   add_tot      the same for the epilogue's "+ add * add_scale + add_shift" (add_gamma, add_beta, [DFL_BN_R][2][Ntot]).
   * dfl_bn_finalize_live (one batched launch per forward pass) turns the totals into the vectors the backward pass and the
   * module state need (scale, shift, mean, invstd, running statistics). */
  double* stat_totals;
  const double* in_tot;
  const float* in_gamma;
  const float* in_beta;
  const double* add_tot;
  const float* add_gamma;
  const float* add_beta;
  double in_count, add_count;
  float bn_eps;
  int32_t reserved3;
  /* ... and for the BatchNorm + ReLU backward operand (x_mode = 1): with in_tot the three coefficients are not read from
   * in_scale but derived from the totals (sum dy, sum dy*r) [DFL_BN_R][2][Cin] left by the kernel that produced dy
   * (stat_totals together with stat_other), in_gamma, the layer's saved mean / 1/std (in_mean, in_invstd) and in_count -- the
   * arithmetic of dfl_bn_bwd_finalize. */
  const float* in_mean;
  const float* in_invstd;
  /* x_mode = 1, 3x3 stride 1 (round 4): the operand the kernel forms while it stages its patches is ALSO written to x_out
   * (bf16 [N][Hin][Win] pixels of ldxo elements, Cin channels; every element exactly once: a workgroup stores the interior of
   * its patch, its K slice's channels, column tile 0 only) -- the layer's weight gradient then reads one plain tensor
   * (dfl_wgrad_args.d_mode = 0 with bias_partial) instead of forming the same values again from (dy, r) in every (cm, cg) tile.
   * NULL: nothing is written. */
  void* x_out;
  int32_t ldxo;
  /* Latency form (round 5; bf16 tensors: csrc/convs_bf16.hip, fp32 tensors in math modes 0 / 1: csrc/convs_f32.hip): 1 asks for the
   * kernel whose dependency chain is shortest instead of the one with the highest throughput -- for the small problems of a batch-1
   * inference forward (the per-image loops of util.py:116-165 and :318-356: 44 dependent convolutions of 0.01 - 1.4 GFLOP each).  A
   * hint: honoured for at most 65536 output pixels and 1.5 GFLOP, Cin / 16 a power of two (<= 1024 channels), no statistics / live
   * totals / x_mode / x_out; otherwise the throughput kernels run.  Same contract, same roundings (another fp32 summation order);
   * dfl_conv_suggest_splits and dfl_conv_config (16 + 39) answer for the form the hint selects. */
  int32_t latency_form;
  /* Output affine (latency form only, inference): the stored value becomes bf16(out_scale[n] * bf16(v) + out_shift[n]) with v what
   * the epilogue would have stored (fp32 tensors: fma(v, out_scale[n], out_shift[n]), nothing is rounded) -- the eval-mode BatchNorm
   * between this convolution and the next one (unet.py:214-218) applied by the PRODUCER with the consumer's two roundings, so that
   * the consumer reads its operand plain (no affine on load; zero padding after BatchNorm stays zero).  Rejected by every other kernel: ask dfl_conv_config (16 + 39 = latency form) before relying on it. */
  const float* out_scale;
  const float* out_shift;
} dfl_conv_args;
#define DFL_BN_R 8

int dfl_conv2d(const dfl_conv_args* a, dfl_stream_t stream);
/* The last 3x3 convolution of a residual block and the block's 1x1 convolution (unet.py:218-231) as ONE launch.  Defined as
 * dfl_conv2d(a) followed by dfl_conv2d(b) -- both outputs are written, same roundings (y1 is rounded to bf16 before b's epilogue
 * adds its BatchNorm) -- and performed as one kernel when dfl_conv_pair_ok(a, b) says 1: a in latency form without K slices, add
 * or scatter; b a 1x1 / stride-1 convolution over the same pixels and columns with add == a->y, add_scale / add_shift, no ReLU,
 * no statistics, its input bf16 (Cin % 16 == 0) or the 1-channel fp32 image of the network's first block.  Otherwise the two
 * launches run one after the other.  A batch-1 forward has 11 such pairs among its 44 dependent convolutions.
 * dfl_conv_pair_ok: 0 = two launches; 1 = one kernel; 2 = a is K-sliced (a->splits > 1): one kernel + ONE finish launch for both
 * outputs, and a->partial must then hold 2 * splits * M * Ntot floats (the second half takes the 1x1 product's slices) -- with a
 * buffer of the single-convolution size the caller must not ask for the pair. */
int dfl_conv2d_pair(const dfl_conv_args* a, const dfl_conv_args* b, dfl_stream_t stream);
int dfl_conv_pair_ok(const dfl_conv_args* a, const dfl_conv_args* b);
/* Number of row blocks whose statistics dfl_conv2d will write for these args (= first dim of stat_partials);
 * depends on a->splits, so set that first. */
int dfl_conv_grid_m(const dfl_conv_args* a);
/* Suggested split-K factor for these args (>= 1); the caller sizes `partial` as splits*M*Ntot floats. */
int dfl_conv_suggest_splits(const dfl_conv_args* a);

/* Geometry search of the bf16 convolution (csrc/convp_bf16.hip).  A geometry is 5 integers: tile configuration (the
 * value dfl_conv_config reports - 16), images per patch, patch height, patch width, K slices.  By default a cost model
 * picks one per layer; a tuning table (measured on the device: tools/tune_convp.py -> dfl_amd/tune/gfx950_convp.txt,
 * loaded when the library is opened) overrides it for the layers it lists.
 *   dfl_conv_candidates     the valid geometries of a layer: writes up to max_rows rows of 5 integers, returns their number
 *   dfl_conv_force_geometry every following dfl_conv_* call of this thread uses `geom` (error if invalid for the layer);
 *                           NULL returns to the table / model.  For tuners and tests.
 *   dfl_conv_tune_add       table entry: key = {N, Hin, Win, Cin, Ntot, KH, KW, stride, pad, scatter2x2} -> geom; a key
 *                           that is already present is replaced; key = NULL empties the table.  An entry is used when it
 *                           is valid and agrees with the caller's `splits`. */
int dfl_conv_candidates(const dfl_conv_args* a, int32_t* out, int32_t max_rows);
int dfl_conv_force_geometry(const int32_t* geom);
int dfl_conv_tune_add(const int32_t* key, const int32_t* geom);

/* ------------------------------------------------------------------------------------------------------------
 * Weight gradient of the same family of layers (torch autograd of unet.py:93,207,211,218,240):
 *   dw[(cm*Cg + cg)*T + t] = sum_m d[m, cm] * G(m, t, cg)
 * G is gathered exactly like X above (taps, stride, pad, affine-on-load, zero padding); d is dense over the
 * same M pixels.  Conv2d: G = layer input, d = output gradient, (cm,cg) = (Cout,Cin) => dw has the torch
 * layout [Cout][Cin][KH][KW].  ConvTranspose2d(k2,s2): G = output gradient gathered with stride 2, d = layer
 * input => [Cin][Cout][2][2].  The pixel range is split over `splits` blocks; with splits > 1 the kernel writes
 * partial[split][T][Cm][Cg] (tap-major: the accumulator tiles store as full 128-byte rows instead of 4-byte
 * scatters) and dfl_sum_partials / dfl_reduce_batch finish the sum into dw, transposing to [Cm][Cg][T].
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* g;        /* gathered tensor, NHWC, pixel stride ldg */
  const float* d;        /* dense tensor [M][ldd] */
  const float* in_scale; /* [Cg] or NULL: affine on load of g */
  const float* in_shift;
  float* dw;             /* [Cm][Cg][T] when splits == 1 */
  float* partial;        /* [splits][T][Cm][Cg] when splits > 1 */
  int32_t N, Hin, Win, Cg, ldg;
  int32_t KH, KW, stride, pad;
  int32_t Hout, Wout, Cm, ldd;
  int32_t splits;
  int32_t d_split;       /* d is a split tensor (see dfl_conv_args.x_split); math mode 1 fast path only */
  int32_t g_bf16, d_bf16;/* g / d are bf16 tensors (math mode 4; csrc/wgradp_bf16.hip when both are; the direct small-K
                            kernels take g_bf16 = 0, d_bf16 = 1) */
  /* BatchNorm + ReLU backward fused into the dense operand (bf16 patch kernels, d_mode = 1): d is dy, d2 the saved ReLU output
   * r (pixel stride ldd2), coef the [3][Cm] coefficients of dfl_bn_bwd_finalize (NULL: plain ReLU backward) and the operand is
   * [r > 0] * (A*dy + B*r + C), as dfl_conv_args.x_mode.  bias_partial (optional; 3x3 bf16 patch kernel: with d_mode 0 as well --
   * d is then the operand itself, materialised by dfl_conv_args.x_out): [splits][Cm] fp32 -- the column sums of the
   * operand over each pixel slice, i.e. the partial sums of the layer's BIAS gradient that dfl_bn_relu_bwd_apply used to leave. */
  int32_t d_mode;
  const float* d2;
  const float* coef;
  float* bias_partial;
  int32_t ldd2, reserved2;
  /* Live statistics (dfl_conv_args.stat_totals; round 4): instead of `coef` the kernel derives A, B, C itself from coef_tot
   * [DFL_BN_R][2][Cm] = totals of (sum dy, sum dy*r), bn_gamma, the saved mean / 1/std and bn_count (dfl_bn_bwd_finalize's arithmetic). */
  const double* coef_tot;
  const float* bn_gamma;
  const float* bn_mean;
  const float* bn_invstd;
  double bn_count;
} dfl_wgrad_args;

int dfl_conv2d_wgrad(const dfl_wgrad_args* a, dfl_stream_t stream);
/* Suggested number of splits for a problem (>= 1); the caller sizes `partial` from it. */
int dfl_wgrad_suggest_splits(const dfl_wgrad_args* a);

/* dst[r*T + t] = sum_{s < splits} src[s*n + t*(n/T) + r],  r < n/T, t < T  (T = 1: plain sum of slices). */
int dfl_sum_partials(const float* src, float* dst, int64_t n, int32_t splits, int32_t T, dfl_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Batched weight re-layout (one launch for the whole network).  dfl_conv2d consumes its weights "quad-packed":
 *   w[(k/4)*Ntot*4 + n*4 + k%4] = W(k, n),  k < K = taps*Cin rounded up to a multiple of 4 with zeros,
 * i.e. one float4 holds four consecutive k of one output column.  Job j builds that matrix from a contiguous
 * parameter src[A][B][C] (C = KH*KW) with one of three index maps:
 *   kind 1: k = c*B + b,  n = a        forward operand of Conv2d        (src [Cout][Cin][T])
 *                                      data gradient of ConvTranspose2d (src [Cin][Cout][T]: k = (tap,co), n = ci)
 *   kind 2: k = c'*A + a, n = b        data gradient of a stride-1 Conv2d (src [Cout][Cin][T]); flip: c = C-1-c'
 *   kind 3: k = a,  n = c*B + b        scatter forms: ConvTranspose2d forward (src [Cin][Cout][4]) and the data
 *                                      gradient of Conv2d(k2,s2) (src [Cout][Cin][4])
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* src;
  float* dst;            /* ceil(K/4) * Ntot * 4 floats */
  int32_t A, B, C;
  int32_t kind, flip;
  int32_t split;         /* 1: write split quads (4 hi bf16 | 4 lo bf16) instead of 4 floats (dfl_conv_args.w_split);
                            2: bf16 chunk layout w[(k/16)*Ntot*16 + n*16 + k%16] (K rounded up to 16 with zeros): one MFMA B
                            fragment of 32 columns is 1 KiB contiguous; dst holds ceil(K/16)*Ntot*16 bf16 */
  /* dfl_pack_weights_tiled only (round 4): a SECOND layout of the same parameter written from the same read of it (the
   * forward operand and the data-gradient operand of a layer: the fp32 master is read once instead of twice), and the job's
   * first tile in the launch's flat tile list (tiles of 32 x 32 x C; a job has (A/32)*(B/32) of them). */
  float* dst2;           /* NULL: one layout */
  int32_t kind2, flip2;
  int32_t first_tile;
  int32_t split2;        /* format of dst2 (as split; round 5: the tiled launches write every format, not only the bf16 chunks) */
} dfl_pack_job;

/* jobs: DEVICE pointer to njobs dfl_pack_job records; max_elems = max over jobs of A*B*C. */
int dfl_pack_weights(const dfl_pack_job* jobs_dev, int32_t njobs, int64_t max_elems, dfl_stream_t stream);
/* The layouts of parameters with A % 32 == 0, B % 32 == 0, C <= 9 (any format: fp32 quads, split quads, bf16 chunks): one workgroup
 * per 32 x 32 x C tile of all jobs (first_tile: ascending prefix sums, total_tiles their sum), 16-byte loads, both layouts of a job
 * from one LDS tile. */
int dfl_pack_weights_tiled(const dfl_pack_job* jobs_dev, int32_t njobs, int32_t total_tiles, dfl_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * BatchNorm2d (unet.py:215,222; torch defaults eps 1e-5, momentum 0.1).
 * Training forward: the producing dfl_conv2d leaves per-block partial sums; dfl_bn_finalize reduces them in
 * fp64, emits scale = gamma*invstd, shift = beta - mean*scale (consumers apply them on load), saves mean and
 * invstd for backward, and updates running_mean / running_var (unbiased) / num_batches_tracked.
 * Eval forward (util.py:120,173,261,313): dfl_bn_eval_prepare derives scale/shift from the running statistics.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* partials; /* [nblocks][2][C]: sum x, sum x^2 */
  const float* gamma;    /* [C] */
  const float* beta;     /* [C] */
  float* running_mean;   /* [C] or NULL */
  float* running_var;    /* [C] or NULL */
  int64_t* num_batches_tracked; /* scalar or NULL */
  float* scale;          /* [C] out */
  float* shift;          /* [C] out */
  float* save_mean;      /* [C] out */
  float* save_invstd;    /* [C] out */
  int64_t count;         /* N*H*W */
  int32_t nblocks, C;
  float eps, momentum;
} dfl_bn_finalize_args;

int dfl_bn_finalize(const dfl_bn_finalize_args* a, dfl_stream_t stream);

/* The same for "live" statistics (dfl_conv_args.stat_totals): ONE launch for all BatchNorm layers of a forward pass, after the
 * last of them -- the convolutions in between derived their scale / shift from the totals themselves.  Each job turns
 * totals [DFL_BN_R][2][C] into scale / shift / save_mean / save_invstd (what the backward pass and tests read) and updates the
 * running statistics (momentum, unbiased variance, num_batches_tracked += 1: nn.BatchNorm2d in training mode, unet.py:215,222).
 * `jobs_dev` lives in device memory. */
typedef struct {
  const double* totals;
  const float* gamma; const float* beta;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;   /* all three NULL: no module state to update */
  float* scale; float* shift; float* save_mean; float* save_invstd;
  int64_t count;
  int32_t C;
  float eps, momentum;
  int32_t reserved;
} dfl_bn_live_job;
int dfl_bn_finalize_live(const dfl_bn_live_job* jobs_dev, int32_t njobs, int32_t max_C, dfl_stream_t stream);
/* Backward counterpart: what dfl_bn_bwd_finalize writes besides the coefficients (which the consumers derive themselves) --
 * dgamma = invstd * (sum dy*r - mean * sum dy), dbeta = sum dy, and optionally `sum_out` = sum dy (the bias gradient of the
 * residual 1x1 convolution that shares the block's output gradient) -- for a batch of layers in one launch. */
typedef struct {
  const double* totals;
  const float* save_mean; const float* save_invstd;
  float* dgamma; float* dbeta; float* sum_out;     /* sum_out may be NULL */
  int32_t C, reserved;
} dfl_bn_bwd_live_job;
int dfl_bn_bwd_finalize_live(const dfl_bn_bwd_live_job* jobs_dev, int32_t njobs, int32_t max_C, dfl_stream_t stream);
int dfl_bn_eval_prepare(const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float* scale, float* shift, int32_t C, float eps,
                        dfl_stream_t stream);

/* Row-block count the streaming reductions below use for an [M][C] tensor (pure function; the caller sizes the
 * partial buffers and sets `nblocks` from it). */
int dfl_rowblock_count(int64_t M, int32_t C);

/* Per-channel partial sums over pixels: partials[blk][0][c] = sum a, partials[blk][1][c] = sum a*b (b NULL => a*a);
 * `partials` holds nblocks*2*C floats, nblocks = dfl_rowblock_count(M, C). */
typedef struct {
  const float* a; const float* b;
  float* partials;
  int64_t M;
  int32_t C, lda, ldb, nblocks;
  int32_t bf16, reserved;   /* bf16 = 1: a and b are bf16 tensors */
} dfl_colstats_args;
int dfl_colstats(const dfl_colstats_args* a, dfl_stream_t stream);

/* BatchNorm + ReLU backward (torch autograd of unet.py:213-215,220-222), two steps:
 * dfl_bn_bwd_finalize: from partial sums of (dy, dy*r) (r = the saved ReLU output that was normalised) computes
 *   dgamma, dbeta and the three coefficients A,B,C such that  d(pre-activation) = [r > 0] * (A*dy + B*r + C).
 * dfl_bn_relu_bwd_apply: materialises that tensor and leaves partial sums of it (the conv bias gradient).
 * With coef == NULL the apply step is the plain ReLU backward [r > 0] * dy (batch_norm=False). */
typedef struct {
  const float* partials; /* [nblocks][2][C]: sum dy, sum dy*r */
  const float* gamma; const float* save_mean; const float* save_invstd;
  float* dgamma; float* dbeta;   /* [C] */
  float* coef;                   /* [3][C] */
  int64_t count;                 /* elements per channel; 0: save_mean / save_invstd are constants (eval mode, running
                                    statistics): A = gamma * invstd, B = C = 0 */
  int32_t nblocks, C;
} dfl_bn_bwd_finalize_args;
int dfl_bn_bwd_finalize(const dfl_bn_bwd_finalize_args* a, dfl_stream_t stream);

typedef struct {
  const float* dy; const float* r; const float* coef; /* coef [3][C] or NULL */
  float* dpre;            /* [M][ldo] */
  float* partials;        /* [nblocks][C] sums of dpre (nblocks = dfl_rowblock_count(M, C)), or NULL */
  int64_t M;
  int32_t C, lddy, ldr, ldo, nblocks;
  int32_t split_out;      /* 1: dpre is written as a split tensor (needs C % 4 == 0, ldo % 4 == 0, 16-byte alignment) */
  int32_t bf16, reserved; /* bf16 = 1: dy, r and dpre are bf16 tensors (C % 8 == 0, ld % 8 == 0); the sums are of the rounded dpre */
} dfl_bn_relu_bwd_args;
int dfl_bn_relu_bwd_apply(const dfl_bn_relu_bwd_args* a, dfl_stream_t stream);

/* out[c] = sum_{b < nblocks} partials[b*stride + c]  (fp64 accumulation). */
int dfl_reduce_partials(const float* partials, float* out, int32_t nblocks, int32_t stride, int32_t C,
                        dfl_stream_t stream);

/* Batched form of dfl_reduce_partials / dfl_sum_partials: ONE launch finishes many small sums (the per-layer bias
 * gradients and the pixel-slice partials of dfl_conv2d_wgrad of a whole backward pass; the recorded program defers
 * them to a few flush points instead of one launch per layer).  Job j:  dst[i'] = sum_{k < count} src[k*stride + i],
 * i < n, accumulated in fp64 in a fixed order (bit-reproducible); i' = i for T <= 1, else the tap-major ->
 * [..][T] transposition i' = (i % (n/T))*T + i/(n/T) of dfl_conv2d_wgrad's slices.  first_block is the prefix sum of
 * dfl_reduce_job_blocks(n, count) over the preceding jobs; total_blocks the sum over all jobs. */
typedef struct {
  const float* src;
  float* dst;
  int64_t n, stride;
  int32_t count, first_block;
  int32_t T, reserved;
} dfl_reduce_job;
int dfl_reduce_job_blocks(int64_t n, int32_t count);
/* jobs: DEVICE pointer to njobs records. */
int dfl_reduce_batch(const dfl_reduce_job* jobs_dev, int32_t njobs, int32_t total_blocks, dfl_stream_t stream);

/* y[m,c] = x[m,c]*scale[c] + shift[c] (+ y_old when accumulate).  scale NULL => copy / add.  Used for the
 * BatchNorm output when a block has no residual branch (do_res=False, unet.py:229) and for the centre-crop copy
 * of the bridge in unpadded mode (unet.py:248-257): src/dst are [N,H,W] windows given by offsets. */
typedef struct {
  const float* x; float* y; const float* scale; const float* shift;
  int32_t N, H, W, C;           /* window size */
  int32_t ldx, xH, xW, xoy, xox;/* source image dims and window origin */
  int32_t ldy, yH, yW, yoy, yox;
  int32_t accumulate;
  int32_t bf16;                 /* 1: x and y are bf16 tensors (C % 8 == 0, ld % 8 == 0) */
} dfl_affine_copy_args;
int dfl_affine_copy(const dfl_affine_copy_args* a, dfl_stream_t stream);

/* Bilinear x2 up-sampling and its adjoint (up_mode='upsample': nn.Upsample(mode='bilinear', scale_factor=2) in front of a 1x1
 * convolution, train_test_code/unet.py:242-244; align_corners=False as torch's default).  fwd: y[N,2H,2W,C] = up(x[N,H,W,C]).
 * bwd: x (+)= up^T(y): x receives the gradient of the small tensor from y, the gradient of the large one. */
typedef struct {
  const void* x; void* y;       /* NHWC with pixel strides ldx / ldy (elements); fp32 or bf16 (both the same) */
  int32_t N, H, W, C;           /* SMALL grid: x is H x W, y is 2H x 2W */
  int32_t ldx, ldy;
  int32_t bf16;                 /* 1: bf16 tensors (C % 8 == 0, ld % 8 == 0) */
  int32_t accumulate;           /* bwd only: add to x instead of overwriting it */
} dfl_upsample_args;
int dfl_upsample2x_fwd(const dfl_upsample_args* a, dfl_stream_t stream);
int dfl_upsample2x_bwd(const dfl_upsample_args* a, dfl_stream_t stream);

/* F.max_pool2d(x, 2) (unet.py:169) and its backward (first maximum in scan order wins; gradient is ADDED to dx). */
typedef struct {
  const float* x; float* y;     /* fwd: x -> y.  bwd: x = saved input, y = dy (read), dx accumulated */
  float* dx;
  int32_t N, H, W, C, ldx, ldy, lddx;   /* H, W: input size; output is floor(H/2) x floor(W/2) */
  int32_t bf16;                          /* 1: x, y and dx are bf16 tensors (C % 8 == 0, ld % 8 == 0) */
} dfl_pool_args;
int dfl_maxpool2x2_fwd(const dfl_pool_args* a, dfl_stream_t stream);
int dfl_maxpool2x2_bwd(const dfl_pool_args* a, dfl_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Output heads (unet.py:176-191): logits = seg_conv(x) (1x1, no bias); seg = Softmax2d(logits) (or logits);
 * heat = lands_1x1[1](lands_1x1[0](cat(x, logits))) (two bias-free 1x1 convs; the second is optional).
 * x is NHWC; seg and heat are written NCHW (the reference's output layout).
 * Limits: n_classes <= 16, num_lands <= 32, n_mid <= 48 (the specialised small-array kernels serve up to 8 / 16 / 24).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* x;         /* [M][ldx], F features */
  const float* w_seg;     /* [NC][F] */
  const float* w_l1;      /* [NM][F+NC] or NULL when L == 0 */
  const float* w_l2;      /* [L][NM] or NULL (single 1x1: NM == L) */
  float* seg;             /* [N][NC][H][W] */
  float* heat;            /* [N][L][H][W] or NULL */
  int32_t N, H, W, F, ldx;
  int32_t NC, NM, L;
  int32_t softmax;
  int32_t x_bf16;         /* 1: x is a bf16 tensor (F % 8 == 0, ldx % 8 == 0); seg / heat stay fp32 */
} dfl_head_fwd_args;
int dfl_head_fwd(const dfl_head_fwd_args* a, dfl_stream_t stream);

/* Head backward: reads dseg/dheat (NCHW, dheat may be NULL), the saved features and seg, writes dx (NHWC) and a
 * per-pixel scratch [M][DFL_HEAD_SCRATCH_LD] holding cat(x,logits) | dlogits | dmid | mid | dheat (padded,
 * offsets below) from which the three weight gradients are taken with dfl_conv2d_wgrad (1x1). */
#define DFL_HEAD_MAX_NC 8
#define DFL_HEAD_MAX_L 16
#define DFL_HEAD_MAX_NM 24
/* Heads beyond those counts (train.py --num-classes is free) run the same kernels compiled with larger per-thread arrays
 * (generic bounds, no fused weight gradients): up to */
#define DFL_HEAD_LARGE_NC 16
#define DFL_HEAD_LARGE_L 32
#define DFL_HEAD_LARGE_NM 48
typedef struct {
  const float* x; const float* seg; const float* dseg; const float* dheat;
  const float* w_seg; const float* w_l1; const float* w_l2;
  float* dx;              /* [M][lddx] */
  float* scratch;         /* [M][scratch_ld] */
  int32_t N, H, W, F, ldx, lddx;
  int32_t NC, NM, L;
  int32_t softmax;
  int32_t scratch_ld;     /* >= dfl_head_scratch_ld(F) */
  int32_t x_bf16;         /* 1: x and dx are bf16 tensors; the scratch rows stay fp32 */
  /* Fused weight gradients (bf16 features with F == 32 only): when dw_seg is set the kernel takes the three weight
   * gradients itself -- per pixel tile the rows [dlogits | dmid | dheat] and [x | logits | mid], rounded to bf16, are
   * multiplied on the matrix cores (fp32 accumulation), every workgroup leaves one 64 x 64 partial in wg_partial
   * (dfl_head_wgrad_blocks(M) * 4096 floats) and a finish kernel of the same call sums them, in a fixed order, into
   * dw_seg [NC][F], dw_l1 [NM][F+NC] and dw_l2 [L][NM] (NULL where the head has no such layer).  `scratch` is then
   * neither read nor written and may be NULL. */
  float* dw_seg; float* dw_l1; float* dw_l2; float* wg_partial;
  /* Live statistics (matrix-core kernels only; round 4): dx is the gradient entering the BatchNorm of the decoder's last
   * convolution.  With stat_totals the kernel adds (sum dx, sum dx * r) of the values it stores -- r = stat_other, the saved ReLU
   * output of that convolution, [M][ldso] bf16 -- to the layer's totals [DFL_BN_R][2][F] (dfl_conv_args.stat_totals): the
   * dfl_colstats pass over dx and r and its finalize launch are not needed. */
  const float* stat_other;
  double* stat_totals;
  int32_t ldso, reserved4;
} dfl_head_bwd_args;
int dfl_head_bwd(const dfl_head_bwd_args* a, dfl_stream_t stream);
int dfl_head_wgrad_blocks(int64_t M);  /* workgroups (= partial slots) of the fused form for M pixels */
/* scratch row layout for F features: [0, Fc) cat(x, logits) with Fc = roundup4(F+NC_MAX) ; then dlogits (8),
 * dmid (24), mid (24), dheat (16). */
int dfl_head_scratch_ld(int32_t F);
int dfl_head_scratch_off(int32_t F, int32_t which); /* which: 0 cat, 1 dlogits, 2 dmid, 3 mid, 4 dheat */
/* The same for a given head shape: the scratch row of a large head (above) is wider. */
int dfl_head_scratch_ld_for(int32_t F, int32_t NC, int32_t NM, int32_t L);
int dfl_head_scratch_off_for(int32_t F, int32_t NC, int32_t NM, int32_t L, int32_t which);

/* ------------------------------------------------------------------------------------------------------------
 * Losses (dice.py:14-86, ncc.py:12-38) with their closed-form gradients.
 * seg/heat are the network outputs seen through util.center_crop (util.py:92-114): 4-D strided views given by
 * element strides (sN, sC, sH; unit stride along W) over an [B, C, h, w] window; targets likewise.
 *   loss = dice_wgt * mean_n( sum_c d_nc / Ceff ) + heat_wgt * mean_{n,l}( -(ncc_nl + 1)/2 )
 * with d_nc = (-2*sum(t*s) + 1e-4) / (sum t^2 + sum s^2 + 1e-4), classes c >= skip_bg.
 * Outputs: loss (1 float), and when grads are requested dseg/dheat as dense [B,C,h,w] tensors.
 * `sums` is a caller-provided scratch of dfl_loss_scratch_doubles(B,C,L) doubles.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* seg; const float* tseg;   /* may be NULL when C == 0 (NCC only) */
  const float* heat; const float* theat; /* NULL when L == 0 */
  float* loss;
  float* dseg; float* dheat;             /* NULL => no gradient */
  float* ncc_vals;                       /* [B][L] raw NCC values (ncc.py:38) or NULL */
  double* sums;
  int64_t seg_sN, seg_sC, seg_sH, tseg_sN, tseg_sC, tseg_sH;
  int64_t heat_sN, heat_sC, heat_sH, theat_sN, theat_sC, theat_sH;
  int32_t B, C, L, h, w;
  int32_t skip_bg;
  float dice_wgt, heat_wgt;
  /* Two-stage use (value in forward, gradient in backward -- what autograd asks for): stage 0 = everything in one call;
   * stage 1 = sums + value only, the gradient coefficients stay in `sums`; stage 2 = gradient only, from the coefficients a
   * stage-1 call left in the same `sums` (seg / heat / targets and sizes as then).  In stage 2 the gradients are
   * multiplied by *grad_scale (a DEVICE scalar: the incoming gradient of the loss; NULL = 1) and may be written as
   * strided [B,C,h,w] windows (element strides; all 0 = dense) -- e.g. the interior of a zero-bordered full-size tensor,
   * which is the gradient of the uncropped network output without a padding pass. */
  const float* grad_scale;
  int64_t dseg_sN, dseg_sC, dseg_sH, dheat_sN, dheat_sC, dheat_sH;
  int32_t stage, reserved;
} dfl_loss_args;
int dfl_dice_ncc_loss(const dfl_loss_args* a, dfl_stream_t stream);
int64_t dfl_loss_scratch_doubles(int32_t B, int32_t C, int32_t L);

/* ------------------------------------------------------------------------------------------------------------
 * Ensemble reduction for one image (util.py:326-373): labels = argmax_c( mean_nets crop(seg) ) with torch.max's
 * first-maximum rule, uint8; heat = mean_nets( (crop(h) - min) / (max - min) ), min/max over the whole cropped
 * [L,h,w] tensor of each net.  seg_ptrs/heat_ptrs: DEVICE arrays of nnets pointers to [C,Hp,Wp]/[L,Hp,Wp] NCHW
 * outputs (batch 1); the crop window starts at (oy, ox).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* const* seg_ptrs; const float* const* heat_ptrs; /* heat_ptrs NULL when L == 0 */
  uint8_t* labels;        /* [h][w] */
  float* avg_seg;         /* [C][h][w] or NULL */
  float* heat_out;        /* [L][h][w] or NULL */
  float* minmax;          /* scratch of 2*nnets*65 floats (final [nnets][2] table + partials) */
  int32_t nnets, C, L, Hp, Wp, h, w, oy, ox;
  int32_t raw_heat;       /* 1: plain mean of the heat maps, no min-max normalisation (util.py:217-229) */
} dfl_ensemble_args;
int dfl_ensemble_reduce(const dfl_ensemble_args* a, dfl_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused multi-tensor SGD step (torch.optim.SGD as configured at train.py:333-334):
 *   g = grad*grad_scale + wd*p ; buf = first ? g : mom*buf + g ; g = nesterov ? g + mom*buf : buf ; p -= lr*g
 * over a flat parameter arena (all tensors laid out back to back; n = total element count).
 * ------------------------------------------------------------------------------------------------------------ */
int dfl_sgd_step(float* p, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                 float weight_decay, float grad_scale, int32_t nesterov, int32_t first_step, dfl_stream_t stream);

/* The same update AND the bf16 weight re-layout of the next step in one pass over the weights (round 4; optimizer.step() followed
 * by the re-layout the next forward needs, train.py:333-334,424): the jobs of dfl_pack_weights_tiled, whose workgroups update
 * their 32 x 32 x C tile of the fp32 master (p, momentum buffer written back) before they emit its layouts from LDS, followed by
 * "plain" jobs (kind = DFL_PACK_PLAIN: src, A = element count, no layouts; DFL_SGD_PLAIN_TILE elements per workgroup) for what
 * has no tiled layout -- biases, BatchNorm parameters, the first layer, the heads.  Gradient and momentum buffer of an element
 * live at src + grad_delta and src + buf_delta (ELEMENTS; the three arenas share one layout; multiples of 4 for the tiled
 * jobs' 16-byte accesses).  No job may share elements with another one. */
#define DFL_PACK_PLAIN 100
#define DFL_SGD_PLAIN_TILE 4096
typedef struct {
  const dfl_pack_job* jobs_dev;
  int64_t grad_delta, buf_delta;
  int32_t njobs, total_tiles;
  float lr, momentum, weight_decay, grad_scale;
  int32_t nesterov, reserved;
} dfl_sgd_pack_args;
int dfl_sgd_pack_tiled(const dfl_sgd_pack_args* a, dfl_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * GPU-side input pipeline: the deterministic part of the reference loader (train_test_code/dataset.py) from raw
 * device arrays, for a whole batch in two launches.
 *   x[b]     = standardise(reflect_pad(proj[b], pad))    (dataset.py:287-293; mean / unbiased std of the PADDED image)
 *   masks[b] = one-hot(labels[b]) as float [C][H][W]     (dataset.py:448-452)
 *   heats[b] = L Gaussian maps exp(-((x-mx)^2+(y-my)^2)/(2 sigma^2)) / (2 pi sigma^2), zero for landmarks that are
 *              inf or outside [0,W-1]x[0,H-1]             (dataset.py:302-325, 421-429)
 * Any of x / masks / heats may be NULL (skipped).  lands is [B][2][L]: row 0 = column (x), row 1 = row (y).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* proj;            /* [B][H][W] raw intensities */
  const unsigned char* labels;  /* [B][H][W] */
  const float* lands;           /* [B][2][L] */
  float* x;                     /* [B][1][H+2*pad][W+2*pad] */
  float* masks;                 /* [B][C][H][W] */
  float* heats;                 /* [B][L][H][W] */
  double* scratch;              /* dfl_prep_scratch_doubles(B) doubles (needed when x != NULL and standardize) */
  int32_t B, H, W, pad, C, L;
  float sigma;
  int32_t standardize;
} dfl_prep_args;
int64_t dfl_prep_scratch_doubles(int32_t B);
int dfl_prep_batch(const dfl_prep_args* a, dfl_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Landmark extraction from predicted heat maps (est_lands_csv.py:96-124, the step after the network in the paper's
 * pipeline).  For image b and landmark l:  (row, col) = arg-max of heats[b][l] over the pixels whose segmentation
 * label equals label_for_land[l] (all pixels when segs / label_for_land is NULL or the entry is negative; ties -> lowest
 * flat index), kept only if ncc_2d(Gaussian 25x25 template (sigma), 25x25 window of the heat map reflect-padded by 12
 * around it) >= min_ncc (0.9 in the reference); otherwise (-1, -1).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* heats;             /* [B][L][H][W] */
  const unsigned char* segs;      /* [B][H][W] labels or NULL */
  const int32_t* label_for_land;  /* [L] or NULL */
  int32_t* rowcol;                /* [B][L][2], 8-byte aligned: a map's slot is also where the workgroups of its arg-max pass meet (64-bit atomic max) before the answer is written there */
  float* ncc;                     /* [B][L] correlation at the arg-max (0 where none), or NULL */
  int32_t B, L, H, W;
  float sigma, min_ncc;
} dfl_est_lands_args;
int dfl_est_lands(const dfl_est_lands_args* a, dfl_stream_t stream);

/* Hard Dice of predicted label maps against ground truth (compute_actual_dice_on_test.py:63-93): for image b and
 * label l = 1..C-1 (background excluded)  dice[b][l-1] = 2*|est==l & gt==l| / (|est==l| + |gt==l|), 1.0 when the label
 * occurs in neither.  counts[b][l][3] = (|est==l|, |gt==l|, |both|) for l = 0..C-1 (exact integers). */
int dfl_hard_dice(const unsigned char* est, const unsigned char* gt, int64_t pixels_per_image, int32_t B, int32_t C,
                  int64_t* counts, double* dice, dfl_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Product arithmetic of the convolution / weight-gradient GEMMs (fast paths; odd channel counts always use fp32):
 *   0 "fp32"    v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulation.
 *   1 "bf16x3"  every fp32 operand value is split into hi + lo bf16 parts when it is staged in LDS and the product is
 *               hi*hi + hi*lo + lo*hi on the bf16 matrix pipe (fp32 accumulation): relative error 2^-16 per product
 *               -- 32x tighter than the TF32 products PyTorch uses for the reference on NVIDIA GPUs by default -- and
 *               ~1.8x the throughput of mode 0.  Forward outputs stay within the 1e-4 parity bar (measured 2e-5).
 *   2 "bf16x6"  three parts, six products: as exact as fp32 multiplication (convolutions only).
 *   3 "bf16"    plain bf16 products (operands rounded to bf16, fp32 accumulation, fp32 tensors): 2^-9 per product; the
 *               arithmetic BASELINE configs[1] names.  NOT inside the 1e-4 forward bar (measured ~1e-2); validated by
 *               training quality (tests/test_gpu_unet.py).
 *   4 "bf16s"   bf16 STORAGE (BASELINE configs[1] as named): activations, activation gradients and the GEMM copies of the
 *               weights live in HBM as bf16; products bf16, accumulation / BatchNorm statistics / losses / master weights /
 *               weight gradients fp32.  The mode is a property of the recorded program (the caller sets the *_bf16 flags
 *               of the argument blocks and allocates bf16 tensors); entry points look at the flags, not at the mode.
 *               Same validation as mode 3 (training quality, label agreement).
 * Process-wide; initial value from the environment variable DFL_MATH, else DFL_MATH_DEFAULT.
 * ------------------------------------------------------------------------------------------------------------ */
#define DFL_MATH_DEFAULT 0
int dfl_get_math_mode(void);
int dfl_set_math_mode(int32_t mode);

/* ------------------------------------------------------------------------------------------------------------
 * Program execution: run a recorded list of the calls above with ONE host->library transition.  The host builds
 * the array once per (network, input shape) and replays it every step (forward, backward); this is the launch
 * path bench.py times.  `args` points to the struct the matching function takes.
 * ------------------------------------------------------------------------------------------------------------ */
typedef enum {
  DFL_OP_CONV = 1, DFL_OP_WGRAD = 2, DFL_OP_SUM_PARTIALS = 3, DFL_OP_PACK = 4, DFL_OP_BN_FINALIZE = 5,
  DFL_OP_BN_EVAL = 6, DFL_OP_COLSTATS = 7, DFL_OP_BN_BWD_FINALIZE = 8, DFL_OP_BN_RELU_BWD = 9,
  DFL_OP_REDUCE_PARTIALS = 10, DFL_OP_AFFINE_COPY = 11, DFL_OP_POOL_FWD = 12, DFL_OP_POOL_BWD = 13,
  DFL_OP_HEAD_FWD = 14, DFL_OP_HEAD_BWD = 15, DFL_OP_MEMSET = 16, DFL_OP_REDUCE_BATCH = 17,
  DFL_OP_RECORD = 18, DFL_OP_WAIT = 19, DFL_OP_UPSAMPLE_FWD = 20, DFL_OP_UPSAMPLE_BWD = 21,
  DFL_OP_BN_FINALIZE_LIVE = 22, DFL_OP_BN_BWD_FINALIZE_LIVE = 23, DFL_OP_CONV_PAIR = 24
} dfl_op_kind;

typedef struct { const float* src; float* dst; int64_t n; int32_t splits; int32_t T; } dfl_sum_partials_args;
typedef struct { const dfl_pack_job* jobs_dev; int64_t max_elems; int32_t njobs; int32_t tiled; } dfl_pack_args;   /* tiled: max_elems = total tiles, dfl_pack_weights_tiled */
typedef struct { const float* gamma; const float* beta; const float* running_mean; const float* running_var;
                 float* scale; float* shift; int32_t C; float eps; } dfl_bn_eval_args;
typedef struct { const float* partials; float* out; int32_t nblocks, stride, C, reserved; } dfl_reduce_partials_args;
typedef struct { void* ptr; int64_t bytes; } dfl_memset_args; /* zero fill */
typedef struct { const dfl_reduce_job* jobs_dev; int32_t njobs, total_blocks; } dfl_reduce_batch_args;
typedef struct { const dfl_bn_live_job* jobs_dev; int32_t njobs, max_C; } dfl_bn_live_args;
typedef struct { const dfl_bn_bwd_live_job* jobs_dev; int32_t njobs, max_C; } dfl_bn_bwd_live_args;
typedef struct { const dfl_conv_args* a; const dfl_conv_args* b; } dfl_conv_pair_args;   /* dfl_conv2d_pair */

/* DFL_OP_RECORD: record library event `event` on the op's stream; DFL_OP_WAIT: make the op's stream wait for it.
 * Events are library-owned, identified by small integers the program chooses (0..65535). */
typedef struct { int32_t event; int32_t reserved; } dfl_sync_args;
#define DFL_MAX_SIDE_STREAMS 2

typedef struct {
  int32_t kind;          /* dfl_op_kind */
  int32_t stream;        /* 0 = the stream passed to dfl_exec; 1..DFL_MAX_SIDE_STREAMS = library-owned side streams.
                            Independent kernels of small layers (a layer's weight gradient next to its data
                            gradient) fill the GPU better side by side; ordering is the program's job (RECORD/WAIT). */
  const void* args;
} dfl_op;

int dfl_exec(const dfl_op* ops, int32_t n_ops, dfl_stream_t stream);
/* Same, with a hipEvent pair recorded on `stream` around every op; blocks until done and returns the per-op
 * milliseconds in ms_out[n_ops].  Measurement aid for bench.py (roofline.achieved); not used on the timed path. */
int dfl_exec_timed(const dfl_op* ops, int32_t n_ops, dfl_stream_t stream, float* ms_out);

/* hipGraph form of a program (BASELINE configs[4]: "test_ensemble.py 5-model ensemble, hipGraph-captured"; the reference
 * enqueues every torch op of UNet.forward one by one, util.py:318-373).  dfl_graph_capture records the ops -- in program
 * order on one stream, side-stream annotations ignored, which is a valid schedule of any program -- into a hipGraph by
 * stream capture on a private stream (nothing executes; `stream` is accepted for symmetry and unused: callers usually
 * sit on the null stream, which cannot capture) and instantiates it; dfl_graph_launch replays it on any stream.  The argument structs are
 * consumed at capture time: pointers and sizes are frozen into the graph, memory CONTENTS are read at replay.  The
 * graph stays valid as long as the buffers it names do. */
/* Tuning knob (process-wide, like the math mode): dfl_conv2d takes its row-tiled 3x3 kernels (conv_rows.hip: wide
 * images, tiles of whole row segments, each kernel row staged once for its three taps) only when they yield at least
 * this many workgroups -- they have no split-K form; default 512.  Returns the previous value.  Tests lower it to
 * exercise those kernels on small problems. */
int dfl_set_conv_rows_min_tiles(int32_t n);

typedef struct dfl_graph_s* dfl_graph_t;
int dfl_graph_capture(const dfl_op* ops, int32_t n_ops, dfl_stream_t stream, dfl_graph_t* graph_out);
int dfl_graph_launch(dfl_graph_t graph, dfl_stream_t stream);
int dfl_graph_nodes(dfl_graph_t graph);      /* kernel / memset nodes captured (diagnostics), negative = error */
int dfl_graph_destroy(dfl_graph_t graph);
/* Tile configuration the launcher picks for these arguments (index into the instantiation tables documented in
 * csrc/conv_gemm.hip / csrc/wgrad_gemm.hip); lets a profile be matched to kernel template names. */
int dfl_conv_config(const dfl_conv_args* a);
int dfl_wgrad_config(const dfl_wgrad_args* a);

#ifdef __cplusplus
}
#endif
#endif /* DFL_HIP_H */

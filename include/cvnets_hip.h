/*
 * cvnets_hip.h — C ABI of libcvnets_hip.so: the MI355X (gfx950) kernels behind the CVNets backbone
 * forward/backward hot path (MobileViT block, transformer encoder, conv-BN-act stacks).
 *
 * The reference (apple/ml-cvnets) is pure Python and has no FFI; its "operator API" for this path is
 * the set of torch.nn / torch.nn.functional calls made by cvnets/layers/* and cvnets/modules/*.  Each
 * entry point below replaces one such call (cited as file:line relative to the reference root).  A
 * reference maintainer binds them with ctypes; the in-tree binding is
 * ml-cvnets_amd/cvnets_amd/_lib.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless noted; no torch / C++ types cross this boundary;
 *   - `dtype` selects the storage type of activation tensors: CVH_DT_F32 (0) or CVH_DT_BF16 (1);
 *     parameters, statistics and parameter gradients are always float32; accumulation is float32;
 *   - activations are NHWC: a [B,C,H,W] feature map is a row-major [B*H*W][C] matrix, C % 8 == 0;
 *     token matrices are [rows][C];
 *   - `stream` is a hipStream_t; kernels are enqueued asynchronously and are hipGraph-capturable
 *     (no allocation, no synchronisation inside);
 *   - return value: 0 on success, a positive hipError_t, or a negative code for rejected arguments
 *     (-1 unknown dtype, -2 unsupported shape/alignment).
 */
#ifndef CVNETS_HIP_H_
#define CVNETS_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define CVH_DT_F32 0
#define CVH_DT_BF16 1
#define CVH_ACT_NONE 0
#define CVH_ACT_SILU 1 /* nn.SiLU  — cvnets/layers/activation/swish.py */
#define CVH_ACT_GELU 2 /* nn.GELU (erf) — cvnets/layers/activation/gelu.py:11-18 */
#define CVH_ACT_RELU 3 /* nn.ReLU — cvnets/layers/activation/relu.py (segmentation heads: model.activation.name = relu) */
/* FFN pairing with a STORED derivative (transformer-sized hidden widths; cvnets/modules/transformer.py:140-155): a forward GEMM called with
 * act = CVH_ACT_GELU_D applies GELU and writes GELU'(pre-activation) - not the pre-activation - to save_pre (cvh_conv_gemm on the
 * large-tile kernels only: -2 elsewhere); a backward GEMM called with actgrad_act = CVH_ACT_DERIV multiplies by actgrad_aux as it is.
 * The erf / exp of the derivative are then evaluated once, in the forward epilogue that needs them anyway, and the separate
 * activation-backward pass over the 4x-wide hidden gradient becomes one multiply in the dX GEMM's epilogue. */
#define CVH_ACT_DERIV 17
#define CVH_ACT_GELU_D 18

/* ---- layout / dtype plumbing ------------------------------------------------------------------ */
/* NCHW float32 -> NHWC `dtype`, channels zero-padded to Cp (Cp % 8 == 0).  Replaces the implicit
 * memory-format conversion of model.to(memory_format=channels_last), main_train.py:72-79. */
int cvh_nchw_to_nhwc(int dtype, const float* in, void* out, int B, int C, int H, int W, int Cp, void* stream);
int cvh_nhwc_to_nchw(int dtype, const void* in, float* out, int B, int C, int H, int W, int Cs, void* stream);
/* Device-side input stage: RandomMixup / RandomCutmix (data/transforms/image_torch.py:100-140, 290-336, applied at
 * engine/training_engine.py:238) fused with the layout / dtype conversion above.  in = NCHW float32 batch; outside the box
 * out[b] = lam*in[b] + (1-lam)*in[(b-1) mod B], inside [y1,y2) x [x1,x2) out[b] = in[(b-1) mod B] (mixup: empty box; cutmix: lam = 1).
 * Cp > 0: out is NHWC `dtype` with channels zero-padded to Cp;  Cp == 0: out is NCHW float32 (the reference's own output format). */
int cvh_mix_batch(int dtype, const float* in, void* out, int B, int C, int H, int W, int Cp, float lam, int x1, int y1, int x2, int y2,
                  void* stream);
/* Parameter packing (float32 torch layout [Cout][Cin][KH][KW] -> `dtype`).  mode 0: forward pack
 * [Cout][KH*KW][pad8(Cin)];  mode 1: dX pack (transposed, taps flipped) [Cin][KH*KW][pad8(Cout)];
 * mode 2: depthwise [KH*KW][C];  mode 3: patch-dX pack [KH*KW][pad8(Cin)][pad8(Cout)] (kernel == stride convs).  Replaces autocast's per-call weight cast (engine/utils.py:19-36). */
int cvh_weight_pack(int dtype, const float* w, void* out, int Cout, int Cin, int KHW, int mode, void* stream);
/* every conv / linear weight of a model packed by ONE launch: table = n_entries x {src ptr, dst element offset, Cout, Cin, KHW,
 * mode, first global element} (int64, device memory), out = flat `dtype` buffer. */
int cvh_weight_pack_multi(int dtype, const long long* table, int n_entries, long long total, void* out, void* stream);
int cvh_cast_from_f32(int dtype, const float* in, void* out, long long n, void* stream);
int cvh_cast_to_f32(int dtype, const void* in, float* out, long long n, void* stream);

/* ---- dense conv / linear as implicit GEMM on MFMA ----------------------------------------------- */
/* out[M=B*Ho*Wo][N] = epilogue( im2col(cat(src1,src2))[M][KH*KW*(C1+C2)] x wgt[N][KH*KW*(C1+C2)]^T )
 * epilogue order: +bias[N] -> (save_pre) -> act -> (* act'(actgrad_aux)) -> dropout -> +residual -> store
 *                 -> optional BatchNorm column statistics (sum, sumsq of the stored values) written to
 *                    stats_part[cvh_conv_gemm_grid_rows(M,N)][2][N].
 * Replaces nn.Conv2d.forward (cvnets/layers/conv_layer.py:18-66, called at :254-255), F.linear
 * (cvnets/layers/linear_layer.py:90), torch.cat before the fusion conv (cvnets/modules/mobilevit_block.py:287),
 * the activation / dropout / residual that follow (cvnets/modules/transformer.py:140-155), and — with the
 * mode-1 weight pack — their dX backward. */
/* 1: cvh_conv_gemm(bf16, plain linear [M x K] x [N x K]^T) accepts act = CVH_ACT_GELU_D (a kernel with the stored-derivative epilogue runs) */
int cvh_conv_gemm_takes_gelu_d(long long M, int K, int N);
int cvh_conv_gemm(int dtype, const void* src1, const void* src2, int C1, int C2, const void* wgt, void* out,
                  int B, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int dil, int N,
                  const float* bias, int act, void* save_pre, const void* actgrad_aux, int actgrad_act,
                  const void* residual, float drop_p, const unsigned long long* seed, unsigned int stream_id,
                  float* stats_part, void* stream);
int cvh_conv_gemm_grid_rows(int M, int N);
/* ---- the stem: 3x3 stride-2 pad-1 conv of the 3-channel NCHW image batch (fp32 or bf16) into Cout = 16 / 32 NHWC bf16 channels ----
 * Replaces, for MobileViT's conv_1 (cvnets/models/classification/mobilevit.py:62-72; ConvLayer2d.forward, cvnets/layers/conv_layer.py:254-255),
 * the NCHW -> NHWC(8) repack + the K = 72 implicit GEMM of the generic path: the image planes are read once, straight from NCHW.
 * y is the raw conv output (BatchNorm follows through stats_part [cvh_stem_rows][2][Cout] = per-workgroup (sum, sum of squares) of the
 * stored values, or NULL).  weight = torch layout [Cout][3][3][3] in w_dtype.  Needs W % 4 == 0; returns -2 for shapes it does not cover. */
int cvh_stem_rows(int B, int H, int W, int Cout);
int cvh_stem_conv_fwd(int in_dtype, const void* x_nchw, int w_dtype, const void* weight, void* y_bf16, float* stats_part, int B, int H, int W,
                      int Cout, void* stream);
/* weight-gradient partials of the same conv: part[cvh_stem_rows][Cout][9][8] float32 (the [N][KH*KW][pad8(Cin)] layout of cvh_gemm_dw's
 * scratch; channels 3..7 are zero), summed by cvh_reduce_multi (kind 1) into the torch-layout gradient.  The image has no gradient. */
int cvh_stem_conv_dw(int in_dtype, const void* x_nchw, const void* dy_bf16, float* part, int B, int H, int W, int Cout, void* stream);

/* measurement aid (bench.py): launches since the last reset and the sum of their algorithmic bytes (input tensor(s) + output + every [M][N]
 * epilogue operand) of the two GEMM kernel families: out[0..1] = gemm_stream_kernel, out[2..3] = conv_gemm_kernel; out = long long[4];
 * reset != 0 clears the tallies */
int cvh_stream_counters(int reset, long long* out);
/* the same for the kernel families of the fused InvertedResidual block: family 0 = dwx_fwd_kernel (x + y2), 1 = dwx_bwd_kernel (x + g2 + y2 +
 * g1), 2 = ir_pb_kernel (dy (+ y3) + y2 + g2), 3 = ir_exp_bwd_kernel (g1 + x + dx), 4 = ir_red_fwd_kernel (y2 + y3); out = long long[2]
 * (launches, algorithmic bytes: every operand and result tensor once, no halo); -2 for an unknown family */
int cvh_family_counters(int family, int reset, long long* out);
/* scratch floats cvh_gemm_dw / cvh_gemm_dw_bias want for a convolution geometry (a multiple of N * KH*KW*(C1+C2): the number of partial rows
 * the chosen kernel writes).  Equals cvh_gemm_dw_scratch_elems(M, N, K) except for the 3x3 stride-1 convs of the MobileViT blocks. */
long long cvh_gemm_dw_scratch_elems_conv(int dtype, int B, int H, int W, int Ho, int Wo, int C1, int C2, int KH, int KW, int stride, int pad,
                                         int dil, int N, int with_bias);
/* dW[N][Cin_real][KH][KW] (float32, torch layout) = (accumulate ? dW : 0) + dY[M][N]^T x im2col(src)[M][K].  The M range is split
 * over workgroups; with `scratch` (>= cvh_gemm_dw_scratch_elems(M, N, K) floats) every split stores its partial tile and a second
 * kernel sums them (no atomics); with scratch == NULL the splits add into dW with fp32 atomics (accumulate must be 1 and dW zeroed /
 * holding the running gradient).  Replaces the weight-gradient half of Conv2d / Linear backward. */
int cvh_gemm_dw(int dtype, const void* dy, const void* src1, const void* src2, int C1, int C2, float* dw,
                int B, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int dil, int N,
                int Cin_real, float* scratch, long long scratch_elems, int accumulate, void* stream);
long long cvh_gemm_dw_scratch_elems(int M, int N, int Ktot);
/* cvh_gemm_dw that also emits the bias gradient of the same layer: bias_part[rows][N] (rows = cvh_gemm_dw_scratch_elems / (N*Ktot),
 * requires scratch) receives the column sums of dY over each partial row's share of M; their sum over rows is db (cvh_reduce_multi,
 * kind 0).  dY is read once for dW and db.  cvh_gemm_dw_folds_bias() == 0: the kernel this shape runs on cannot (use cvh_colsum). */
int cvh_gemm_dw_bias(int dtype, const void* dy, const void* src1, const void* src2, int C1, int C2, float* dw, float* bias_part,
                     int B, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int dil, int N,
                     int Cin_real, float* scratch, long long scratch_elems, int accumulate, void* stream);
int cvh_gemm_dw_folds_bias(int dtype, int M, int N, int Ktot);
/* dX of a non-overlapping strided conv (kernel == stride, pad 0: ViT patch-embedding convs, cvnets/models/classification/vit.py:89-123):
 * dx[B][H][W][Cin] = scatter(dy[B*Ho*Wo][Cout] x wgt^T), wgt = mode-3 pack. */
int cvh_conv_dx_patch(int dtype, const void* dy, const void* wgt, void* dx, int B, int Ho, int Wo, int Cout, int KH, int KW,
                      int stride, int Cin, int H, int W, void* stream);

/* ---- depthwise 3x3 conv (groups == C) ----------------------------------------------------------- */
/* Replaces nn.Conv2d(groups=C) in InvertedResidual (cvnets/modules/mobilenetv2.py:194-207). */
int cvh_dwconv_fwd(int dtype, const void* x, const void* wp, void* y, int B, int H, int W, int Ho, int Wo, int C, int K,
                   int stride, int pad, int dil, float* stats_part, void* stream);
int cvh_dwconv_rows(int B, int Ho, int Wo, int C, int K, int stride, int pad, int dil); /* rows of stats_part written by cvh_dwconv_fwd */
int cvh_dwconv_bwd_x(int dtype, const void* dy, const void* wp, void* dx, int B, int H, int W, int Ho, int Wo, int C, int K,
                     int stride, int pad, int dil, void* stream);
int cvh_dwconv_bwd_w(int dtype, const void* x, const void* dy, float* part, int B, int H, int W, int Ho, int Wo, int C, int K,
                     int stride, int pad, int dil, void* stream);
int cvh_dwconv_bwd_w_rows(int B, int Ho, int Wo, int C, int K, int stride, int pad, int dil); /* rows of part[rows][C*K*K] */

/* ---- BatchNorm2d (train-mode batch statistics) + activation ------------------------------------- */
/* Replaces nn.BatchNorm2d (cvnets/layers/normalization/batch_norm.py:14-49) followed by nn.SiLU, and the
 * residual add of InvertedResidual.forward (cvnets/modules/mobilenetv2.py:231-235). */
int cvh_colreduce_rows(long long rows, int C); /* rows of part[rows][2][C] for the three reducers below */
int cvh_bn_stats(int dtype, const void* x, long long rows, int C, float* part, void* stream);
int cvh_bn_finalize(const float* part, int R, int C, double count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                    void* stream);
int cvh_bn_eval_coeff(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C, float* mean,
                      float* invstd, float* scale, float* shift, void* stream);
int cvh_bn_apply(int dtype, const void* x, const float* scale, const float* shift, int act, const void* residual, void* y,
                 long long rows, int C, void* stream);
/* cvh_bn_apply that ALSO emits what the next InvertedResidual block needs of its input (csrc/bngram.hip; bf16, C in {16, 32, 64}):
 * gram_s[C*C + C] = G = y^T y (full symmetric [C][C]) followed by s = 1^T y, of the values as stored, from the same pass — instead of a
 * second read of y by cvh_gemm_dw(y, y) + cvh_colsum (cvnets/modules/mobilenetv2.py:231-235 feeding the next block's :180-207).
 * part = float[R * (C*C + C)] scratch, R = cvh_bn_apply_gram_rows(rows, C) (0: not covered, use the separate entry points). */
int cvh_bn_apply_gram_rows(long long rows, int C);
int cvh_bn_apply_gram(int dtype, const void* x, const float* scale, const float* shift, int act, const void* residual, void* y,
                      long long rows, int C, float* part, int R, float* gram_s, void* stream);
int cvh_bn_bwd_reduce(int dtype, const void* x, const void* dout, const float* scale, const float* shift, const float* mean,
                      const float* invstd, int act, long long rows, int C, float* part, void* stream);
int cvh_bn_bwd_finalize(const float* part, int R, int C, double count, const float* gamma, const float* mean, const float* invstd,
                        int training, int accumulate, float* dgamma, float* dbeta, float* ca, float* cb, float* cc, void* stream);
/* the same for partial rows whose second sum was taken against the BatchNorm's OUTPUT out = gamma * xhat + beta instead of xhat
 * (cvh_ir_exp_bwd_s forms them where `out` is the next block's input): sum dz * xhat = (sum dz * out - beta * sum dz) / gamma */
int cvh_bn_bwd_finalize_out(const float* part, int R, int C, double count, const float* gamma, const float* beta, const float* mean,
                            const float* invstd, int training, int accumulate, float* dgamma, float* dbeta, float* ca, float* cb, float* cc,
                            void* stream);
int cvh_bn_bwd_apply(int dtype, const void* x, const void* dout, const float* scale, const float* shift, int act, const float* ca,
                     const float* cb, const float* cc, void* dx, long long rows, int C, void* stream);

/* ---- BatchNorm links: train-mode BatchNorm folded into the kernels on either side of it -------------- */
/* ConvLayer2d stacks (conv -> BN -> SiLU -> conv ..., cvnets/layers/conv_layer.py:254-255; InvertedResidual.forward,
 * cvnets/modules/mobilenetv2.py:231-235) with NO standalone BatchNorm pass over the activations: a producer kernel emits the column
 * statistics of the tensor it writes as per-workgroup partial rows (finalised by cvh_bn_finalize / cvh_bn_bwd_finalize, which only
 * touch [rows][2][C] floats); the consumer applies the resulting per-channel coefficients while loading its operand
 * (cvh_operand_xf).  Only raw conv outputs (forward) and the activation-gradient products g = dz * act'(bn(y)) (backward) ever reach
 * HBM.  The struct lives in HOST memory; the pointers inside are device pointers. */
typedef struct cvh_operand_xf {
  int mode;           /* 0: v = a;  1: v = act(c0*a + c1);  2: v = c0*a + c1*src2 + c2   (c* indexed by the operand's channel) */
  const void* src2;   /* second source tensor, same shape / dtype as the operand (mode 2) */
  const float* c0;
  const float* c1;
  const float* c2;
  int act;            /* CVH_ACT_* (mode 1) */
} cvh_operand_xf;
/* Pointwise conv / linear GEMM with operand transform:
 *   out[M][N] = epilogue( xf(a)[M][K] x wgt[N][K]^T )   (wgt = mode-0 pack for forward, mode-1 pack for dX)
 *   e_mode 0: store (+residual); stats_part[cvh_conv_gemm_grid_rows(M,N)][2][N] receives (sum, sumsq) of the stored values;
 *   e_mode 1: BatchNorm-backward epilogue: g = acc * act'(e_aux*scale + shift) is stored, e_aux[M][N] = raw output of the BatchNorm'd
 *             conv, e_stats[4][N] = its forward statistics (mean, invstd, scale, shift); stats_part receives (sum g, sum g*xhat)
 *             — the input of cvh_bn_bwd_finalize.
 * stats_part may be NULL.  Replaces Conv2d(1x1) forward / dX + the BatchNorm passes around it. */
int cvh_pw_gemm_bn(int dtype, const void* a, const cvh_operand_xf* a_xf, int K, const void* wgt, void* out, long long M, int N,
                   const void* residual, int e_mode, const void* e_aux, const float* e_stats, int e_act, float* stats_part,
                   void* stream);
/* dW[N][Cin_real] = (accumulate ? dW : 0) + xf(dy)[M][N]^T x xf(x)[M][K]  (scratch as in cvh_gemm_dw) */
int cvh_pw_gemm_dw_bn(int dtype, const void* dy, const cvh_operand_xf* dy_xf, const void* x, const cvh_operand_xf* x_xf, float* dw,
                      long long M, int N, int K, int Cin_real, float* scratch, long long scratch_elems, int accumulate, void* stream);
/* Depthwise 3x3 (pad 1, stride 1 / 2) on LDS-staged tiles: y = dwconv(xf(x)); stats_part[cvh_dwconv_bn_rows()][2][C] receives
 * (sum, sumsq) of y (NULL: skip). */
int cvh_dwconv_bn_rows(int B, int Ho, int Wo, int C, int stride);
int cvh_dwconv_bn_fwd(int dtype, const void* x, const cvh_operand_xf* x_xf, const void* wp, void* y, int B, int H, int W, int Ho, int Wo,
                      int C, int stride, float* stats_part, void* stream);
/* Backward of the same conv in ONE pass over its operands: dy = xf(g_out) (mode 2: ca*g_out + cb*y_out + cc; mode 0: g_out is dy),
 * z = act(scale*x_raw + shift) from in_stats[4][C] (the BatchNorm in FRONT of the conv, required):
 *   g_in = dwconv^T(dy) * act'(scale*x_raw + shift) is stored;
 *   stats_part[rows][2][C] receives (sum g_in, sum g_in*xhat) (input of cvh_bn_bwd_finalize);
 *   dw_part[rows][C*9] receives each workgroup's share of dW[c][tap] = sum_pixels dy * z(shifted) (sum the rows with cvh_sum_partials;
 *   torch layout [C][1][3][3]);  rows = cvh_dwconv_bn_rows(). */
int cvh_dwconv_bn_bwd(int dtype, const void* g_out, const cvh_operand_xf* dy_xf, const void* x_raw, const float* in_stats, int in_act,
                      const void* wp, void* g_in, float* stats_part, float* dw_part, int B, int H, int W, int Ho, int Wo, int C, int stride,
                      void* stream);

/* The wide half of InvertedResidual.forward (cvnets/modules/mobilenetv2.py:180-207,231-235: exp 1x1 -> BatchNorm -> act -> depthwise 3x3 ->
 * BatchNorm) with the 4x-wide expansion output y1 = x W1^T never in HBM (bf16; Cin in {16, 32, 64, 96, 128}, stride 1 only up to 64;
 * hid % 8 == 0), everything GEMM- or stencil-shaped on the matrix pipe (csrc/dwx.hip):
 *   cvh_gram_bn_stats  one partial-statistics row part[2][hid] = (sum y1, sum y1^2) for cvh_bn_finalize from the Gram matrix G = x^T x
 *                      ([K] rows of pitch Gp: cvh_gemm_dw on (x, x)) and s = 1^T x (cvh_colsum) of the NARROW input — y1 is linear in x;
 *   cvh_dwx_fwd        y2 = dwconv(act(scale1 * (x W1^T) + shift1)), statistics of y2 -> stats_part[cvh_dwx_rows()][2][hid] (NULL: skip);
 *   cvh_dwx_bwd        g_in = dwconv^T(dy) * act'(bn1(y1)) with y1 recomputed from x at the tile's own pixels, dy = ca * g_out + cb * y_out
 *                      + cc (y_out == NULL: dy = g_out); dw_part[rows][hid * 9] = per-workgroup depthwise weight gradients,
 *                      stats_part[rows][2][hid] = (sum g_in, sum g_in * xhat1).  in_stats = [4][hid] (mean, invstd, scale, shift).
 * w1 = [hid][Cin], wd = [9][hid] packed weights (cvh_weight_pack).  Replaces Conv2d(1x1) + BatchNorm2d + SiLU + Conv2d(groups = C) and
 * their autograd backward; returns -2 for geometries it does not cover (callers fall back to cvh_pw_gemm_bn + cvh_dwconv_bn_*). */
int cvh_dwx_rows(int B, int Ho, int Wo, int hid, int stride);
/* rows of cvh_dwx_fwd's stats_part.  Full strips (H % 8 == 0, W % 16 == 0, Cin <= 64, hid % 64 == 0) run the strip-streaming kernel of
 * csrc/dwxs.hip: a loader wave per workgroup streams the block input top to bottom through LDS, expansion and stencil are one software
 * pipeline, the rows two chunks share are computed once. */
int cvh_dwx_fwd_rows(int B, int H, int W, int Cin, int hid, int stride);
/* out[i] = alpha * a[i] + (b ? b[i] : 0): the column sums s = 1^T x of an InvertedResidual OUTPUT without reading it — the block ends in a
 * train-mode BatchNorm (cvnets/modules/mobilenetv2.py:208-219, batch_norm.py:14-49), so sum_rows bn(y)[c] = rows * beta[c] (+ the input's
 * column sums on the residual path); feeds cvh_gram_bn_stats of the next block. */
int cvh_axpb(const float* a, float alpha, const float* b, float* out, int n, void* stream);
int cvh_gram_bn_stats(const float* G, const float* s, const void* w1, float* part, int hid, int K, int Gp, void* stream);
int cvh_dwx_fwd(int dtype, const void* x, const void* w1, const float* scale1, const float* shift1, int act1, const void* wd, void* y2,
                float* stats_part, int B, int H, int W, int Ho, int Wo, int Cin, int hid, int stride, void* stream);
int cvh_dwx_bwd(int dtype, const void* x, const void* w1, const float* in_stats, int act1, const void* g_out, const void* y_out,
                const float* ca, const float* cb, const float* cc, const void* wd, void* g_in, float* stats_part, float* dw_part,
                int B, int H, int W, int Ho, int Wo, int Cin, int hid, int stride, void* stream);

/* Conv2d(1x1).backward of the block's PROJECTION conv (cvnets/modules/mobilenetv2.py:208-219; it sits behind depthwise -> BatchNorm -> SiLU)
 * from ONE pass over the 4x-wide y2 [M][hid] (bf16, SiLU; csrc/ir_pb.hip):
 *   g2 = (dy3 W3) * act'(bn2(y2)) stored [M][hid]; stats_part[rows][2][hid] = (sum g2, sum g2 * xhat2) for cvh_bn_bwd_finalize;
 *   dw_part[rows][Cout][hid] = partial dW3 = dy3^T act(bn2(y2)) (sum the rows with cvh_sum_partials / cvh_reduce_multi).
 * dy = dy3 [M][Cout], or — y3 != NULL — the block's output gradient with dy3 = c3[0] * dy + c3[1] * y3 + c3[2] formed on load (c3 = the
 * [3][Cout] coefficients of cvh_bn_bwd_finalize for the projection's BatchNorm: no cvh_bn_bwd_apply pass).  st2 = [4][hid] mean, invstd,
 * scale, shift; w3t = [hid][Cout] transposed pack (cvh_weight_pack mode 1).  Covers hid % 64 == 0, Cout in {32, 64, 96, 128, 160},
 * M >= 4096: rows = cvh_ir_pb_rows() is 0 otherwise (callers fall back to cvh_pw_gemm_dw_bn + cvh_pw_gemm_bn). */
int cvh_ir_pb_rows(int M, int hid, int Cout);
int cvh_ir_pb(int dtype, const void* dy, const void* y3, const float* c3, const void* y2, const float* st2, int act, const void* w3t, void* g2,
              float* stats_part, float* dw_part, int M, int hid, int Cout, void* stream);

/* Linear side of a BatchNorm link (y = x W^T in front of the BatchNorm, e.g. the 1x1 expansion conv): with coef[3][N] = (ca, cb, cc) of
 * cvh_bn_bwd_finalize and g = dz * act'(bn(y)),
 *   dX = g (diag(ca) W) + x (W^T diag(cb) W) + 1 (cc^T W): cvh_bn_dx_weights writes wcat[pad8(K)][N + pad8(K)] (`dtype`) and
 *        bias[pad8(K)] so that ONE cvh_conv_gemm over the channel-concat (src1 = g, src2 = x) with that weight / bias is dX;
 *   dW = diag(ca) (g^T x) + diag(cb) W (x^T x) + cc (1^T x): cvh_bn_dw_combine folds P = g^T x [N][K] (cvh_gemm_dw on g), the Gram
 *        matrix G = x^T x [pad8(K)][pad8(K)] (cvh_gemm_dw on x, x) and the column sums s[pad8(K)] (cvh_colsum) into dw[N][K].
 * w = float32 [N][K] (torch [N][K][1][1]).  y itself is never read.  Replaces Conv2d(1x1).backward behind a BatchNorm. */
int cvh_bn_dx_weights(int dtype, const float* w, const float* coef, void* wcat, float* bias, int N, int K, void* stream);
/* dX (as above: [g | x] wcat^T + bias (+ residual)) AND the raw product P = g^T x of dW from ONE pass over g [M][hid] (bf16; the 4x-wide
 * gradient is streamed once instead of once per product): p_part[cvh_ir_exp_bwd_rows][hid][Cin] float32 partial products, summed with
 * cvh_sum_partials and finished by cvh_bn_dw_combine.  Covers (hid, Cin) = (64, 16), (128, 32), (256, 64) with M >= 65536 — the expansion
 * convs of MobileViT's layer_1 .. layer_3 (cvnets/modules/mobilenetv2.py:180-193); cvh_ir_exp_bwd_rows returns 0 for anything else. */
/* The projection conv of the same block in the forward pass as a read-dominated stream (csrc/ir_fwd.hip):
 *   out[M][N] = act(scale * y2 + shift)[M][hid] x wgt[N][hid]^T, stats_part[cvh_ir_red_fwd_rows][2][N] = (sum, sumsq) of the stored values
 * (NULL: skip) — cvh_pw_gemm_bn with a mode-1 operand transform and e_mode 0, for (hid, N) = (64, 32), (128, 64), (256, 64), (256, 96), bf16,
 * M >= 65536 (cvnets/modules/mobilenetv2.py:208-219); cvh_ir_red_fwd_rows returns 0 for anything else. */
int cvh_ir_red_fwd_rows(long long M, int hid, int N);
int cvh_ir_red_fwd(int dtype, const void* y2, const float* scale, const float* shift, int act, const void* wgt, void* out, float* stats_part,
                   long long M, int hid, int N, void* stream);
int cvh_ir_exp_bwd_rows(long long M, int hid, int Cin);
int cvh_ir_exp_bwd(int dtype, const void* g, const void* x, const void* wcat, const float* bias, const void* residual, void* dx,
                   float* p_part, long long M, int hid, int Cin, void* stream);
/* cvh_ir_exp_bwd that also leaves s_part[cvh_ir_exp_bwd_rows][2][Cin]: per workgroup, the column sums of dX (as stored) and of dX * x —
 * the statistics the BatchNorm backward of the PREVIOUS block's projection (whose output x is) needs of its incoming gradient dX
 * (cvnets/modules/mobilenetv2.py:231-235 chained: block k's output is block k+1's input); s_part == NULL: cvh_ir_exp_bwd */
int cvh_ir_exp_bwd_s(int dtype, const void* g, const void* x, const void* wcat, const float* bias, const void* residual, void* dx,
                     float* p_part, float* s_part, long long M, int hid, int Cin, void* stream);
int cvh_bn_dw_combine(const float* P, const float* w, const float* G, const float* s, const float* coef, float* dw, int N, int K,
                      int accumulate, void* stream);

/* ---- reductions / small ops --------------------------------------------------------------------- */
/* `accumulate` != 0: results are ADDED to the destination (parameter gradients written straight into .grad buffers) */
int cvh_colsum(int dtype, const void* x, long long rows, int C, float* part, float* out, float scale, int accumulate, void* stream); /* bias grads */
/* Deferred multi-tensor reduction: out = (accumulate ? out : 0) + scale * sum over `rows` partial rows, for up to thousands of small
 * tensors in ceil(n / CVH_REDUCE_MAX) launches.  kind 0: out[j] = sum_r part[r*row_stride + j], j < n_out.  kind 1: `part` rows are
 * cvh_gemm_dw split partials [N][KH*KW*Cin] and out is the torch weight gradient [N][Cin_real][KH*KW] (n_out = N*Ktot,
 * row_stride = N*Ktot).  Producers that can leave their partial rows un-summed: cvh_gemm_dw / cvh_pw_gemm_dw_bn with dw == NULL
 * (scratch then holds cvh_gemm_dw_scratch_elems / (N*Ktot) rows), cvh_colsum with out == NULL, cvh_layernorm_bwd, cvh_dwconv_bn_bwd.
 * `descs` is HOST memory (copied into the kernel arguments: hipGraph-capturable). */
#define CVH_REDUCE_MAX 48
typedef struct cvh_reduce_desc {
  const float* part;
  float* out;
  long long row_stride;
  long long n_out;
  int rows;
  int kind;
  int N, Ktot, Cin, Cin_real, khw;
  float scale;
  int accumulate;
  int pad_;
} cvh_reduce_desc;
int cvh_reduce_multi(const cvh_reduce_desc* descs, int n, void* stream);
int cvh_sum_partials(const float* part, int R, int stride, int Wd, float* out, float scale, int accumulate, void* stream);
/* GlobalPool(mean) cvnets/layers/global_pool.py:60-71 */
int cvh_pool_fwd(int dtype, const void* x, void* y, int B, int HW, int C, void* stream);
int cvh_pool_bwd(int dtype, const void* dy, void* dx, int B, int HW, int C, void* stream);
/* nn.AdaptiveAvgPool2d(OS) on x[B][H][W][C] -> y[B][OS][OS][C] with torch's (overlapping) windows; PSPNet pyramid bins
 * (cvnets/modules/pspnet_module.py:73-88), OS = 1 = global pool.  bwd = exact adjoint. */
int cvh_adaptive_pool_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, int OS, void* stream);
int cvh_adaptive_pool_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int C, int OS, void* stream);
/* Dropout cvnets/layers/dropout.py:11-29: counter-based mask = f(*seed, stream_id, element index) */
int cvh_dropout(int dtype, const void* x, void* y, long long n, float p, const unsigned long long* seed, unsigned int stream_id,
                void* stream);
/* StochasticDepth, torchvision "row" mode (cvnets/layers/stochastic_depth.py:10-18; cvnets/modules/transformer.py:140-155):
 * y[r] = res[r] + x[r] * keep(sample(r)) / (1-p), one Bernoulli(1-p) draw per sample; sample(r) follows the unfold map of cvh_attn_*
 * (ph, pw, H, W; ph = pw = H = 1, W = S: contiguous sequences of S tokens).  res == NULL: y = x * keep (also the backward). */
int cvh_drop_path(int dtype, const void* x, const void* res, void* y, long long rows, int C, int ph, int pw, int H, int W, float p,
                  const unsigned long long* seed, unsigned int stream_id, void* stream);
int cvh_seed_advance(unsigned long long* seed, void* stream);
/* nn.Dropout2d (cvnets/layers/dropout.py:32-50): drops whole channels per sample, NHWC x[B][HW][C]; keep(b, c) is regenerated from
 * (*seed, stream_id), so the backward pass is the same call on dY. */
int cvh_dropout2d(int dtype, const void* x, void* y, int B, int HW, int C, float p, const unsigned long long* seed, unsigned int stream_id,
                  void* stream);
/* torch.cat(dim=1) of n <= 8 NHWC tensors parts[i][rows][channels[i]] into whole[rows][sum channels] (split == 0), or the inverse copy
 * (split != 0: gradient of the concat).  ASPP branch concat, cvnets/modules/aspp_block.py:118-121.  `parts` / `channels` are HOST arrays. */
int cvh_cat_channels(int dtype, void* const* parts, const int* channels, int n, void* whole, long long rows, int split, void* stream);
int cvh_add(int dtype, const void* a, const void* b, void* y, long long n, void* stream);

/* F.interpolate(mode="bilinear") for feature maps that are not a multiple of the patch: align_corners=False in
 * MobileViTBlock.unfolding/folding (cvnets/modules/mobilevit_block.py:191-200, 260-266), align_corners=True in
 * MobileViTBlockv2.resize_input_if_needed (:595-603); bwd = exact adjoint (gather form). */
int cvh_resize_bilinear_fwd(int dtype, const void* x, void* y, int B, int H, int W, int Ho, int Wo, int C, int align_corners, void* stream);
int cvh_resize_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int Ho, int Wo, int C, int align_corners, void* stream);

/* The reference LayerNorm's channel-first branch as it fires on a [B', S, C] token tensor with S == C
 * (cvnets/layers/normalization/layer_norm.py:53-66): statistics per (sequence, channel) over the S tokens, (std + eps) denominator,
 * gamma / beta indexed by the token position.  Sequences are addressed through the (ph, pw, n_w, H, W) row map of cvh_attn_*.
 * stats[nseq][2][C] = (mean, std) is written by fwd and read by bwd; part[nseq][2][S] receives each sequence's contribution to
 * (dgamma, dbeta) — sum over its nseq rows with cvh_sum_partials. */
int cvh_ln_seq_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats, int nseq, int S, int C, int ph,
                   int pw, int n_w, int H, int W, float eps, void* stream);
int cvh_ln_seq_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* stats, void* dx, float* part, int nseq, int S,
                   int C, int ph, int pw, int n_w, int H, int W, float eps, void* stream);

/* ---- CLIP heads and loss ------------------------------------------------------------------------- */
/* dst[r] = src[idx[r]] (scatter == 0) / dst[idx[r]] = src[r] (scatter == 1, dst pre-zeroed): the EOT-token gather
 * token_emb[arange(B), text_tokens.argmax(-1)] (cvnets/text_encoders/transformer.py:413-421) and its adjoint. */
int cvh_rows_gather_idx(int dtype, const void* src, const long long* idx, void* dst, int R, int C, int scatter, void* stream);
/* F.normalize(x, dim=-1) (simple_projection_head.py:83-84, text_encoders/transformer.py:424-425); inv_norm[R] is saved for bwd */
int cvh_l2norm_fwd(int dtype, const void* x, void* y, float* inv_norm, int R, int C, float eps, void* stream);
int cvh_l2norm_bwd(int dtype, const void* y, const void* dy, const float* inv_norm, void* dx, int R, int C, float eps, void* stream);
/* ContrastiveLossClip._forward_clip (loss_fn/multi_modal_img_text/contrastive_loss_clip.py:77-94): per-row cross-entropy of
 * (*scale) * logits[N][M] against labels i + label_offset (= rank * N).  fwd: loss_rows[N], lse[N]; bwd: dlogits and per-row
 * contributions to d(*scale), both already multiplied by the upstream scalar *gout. */
int cvh_scaled_ce_fwd(int dtype, const void* logits, const float* scale, float* loss_rows, float* lse, int N, int M, int label_offset,
                      void* stream);
int cvh_scaled_ce_bwd(int dtype, const void* logits, const float* scale, const float* lse, const float* gout, void* dlogits,
                      float* dscale_rows, int N, int M, int label_offset, void* stream);

/* ---- optimizer + EMA step (SURVEY 8f "next" row 1) ------------------------------------------------- */
/* torch.optim.AdamW as configured by optim/adamw.py:16-46 (decoupled weight decay, bias correction, no amsgrad) for ALL parameter
 * tensors in one launch, optionally followed in the same pass by EMA.update_parameters (cvnets/misc/averaging_utils.py:43-55).
 * table[n+1][8] int64 = {param ptr, grad ptr, ema ptr or 0, offset into the flat moment buffers, numel, group index, prefix start, 0};
 * group_hp[g][2] = {lr, weight_decay} and *step (number of steps taken so far, advanced by the call) live on the device so the
 * launch is hipGraph-capturable; inv_grad_scale (nullable) multiplies the gradients (GradScaler unscale). */
int cvh_adamw_multi(const long long* table, int n_tensors, long long total, float* m_flat, float* v_flat, const float* group_hp, float beta1,
                    float beta2, float eps, float* step, const float* inv_grad_scale, float ema_momentum, void* stream);
/* dst = dst*(1-momentum) + momentum*src over a table[n+1][4] = {dst ptr, src ptr, numel, prefix start}: EMA of the float buffers */
int cvh_lerp_multi(const long long* table, int n_tensors, long long total, float momentum, void* stream);

/* ---- classification loss (SURVEY 8f "next" row 2) -------------------------------------------------- */
/* CrossEntropy._compute_loss (loss_fn/classification/cross_entropy.py:65-92) = F.cross_entropy(logits[N][M], labels, weight=None,
 * ignore_index, label_smoothing), per-row part: loss_rows[N] (0 for ignored rows) and lse[N]; the mean over valid rows is the
 * caller's.  bwd: dlogits = *gout * (softmax - (1-eps) * onehot - eps/M), *gout = upstream gradient / number of valid rows. */
int cvh_ce_fwd(int dtype, const void* logits, const long long* labels, float label_smoothing, long long ignore_index, float* loss_rows,
               float* lse, int N, int M, void* stream);
/* the 'mean' reduction of the same call in one launch: out2[0] = sum(loss_rows) / max(1, #labels != ignore_index), out2[1] = 1 / that count */
int cvh_ce_mean(const float* loss_rows, const long long* labels, long long ignore_index, float* out2, int N, void* stream);
int cvh_ce_bwd(int dtype, const void* logits, const long long* labels, const float* lse, const float* gout, float label_smoothing,
               long long ignore_index, void* dlogits, int N, int M, void* stream);

/* the same loss for PROBABILITY targets target[N][M] (float32; the mixtures RandomMixup / RandomCutmix produce): per row
 * sum_j t'_ij (lse_i - z_ij) with t' = (1-eps) t + eps/M; tsum[N] = sum_j t'_ij is saved for bwd; mean over all rows is the caller's. */
int cvh_ce_soft_fwd(int dtype, const void* logits, const float* target, float label_smoothing, float* loss_rows, float* lse, float* tsum, int N,
                    int M, void* stream);
int cvh_ce_soft_bwd(int dtype, const void* logits, const float* target, const float* lse, const float* tsum, const float* gout,
                    float label_smoothing, void* dlogits, int N, int M, void* stream);

/* ---- MobileViTv2: GroupNorm(1) ("layer_norm_2d") and linear self-attention ------------------------------ */
/* LayerNorm2D_NCHW = nn.GroupNorm(num_groups=1) (cvnets/layers/normalization/layer_norm.py:75-108) on an NHWC map [B][HW][C]:
 * per-sample statistics over HW*C, per-channel affine.  stats[B][2] = (mean, rstd) is written by fwd and read by bwd;
 * part is scratch of B * cvh_gn_chunks() * 2 floats (fwd) / B * cvh_gn_chunks() * 2*C floats (bwd: per-chunk column sums of
 * dy and dy*xhat — reduce over its B*chunks rows (cvh_sum_partials) for dbeta [0,C) / dgamma [C,2C)); coeff is B*2 scratch. */
int cvh_gn_chunks(int B, int HW, int C);
int cvh_gn_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats, float* part, int B, int HW, int C,
               float eps, void* stream);
int cvh_gn_bwd(int dtype, const void* x, const void* dy, const float* stats, const float* gamma, void* dx, float* part, float* coeff, int B,
               int HW, int C, void* stream);
/* LinearSelfAttention._forward_self_attn (cvnets/layers/linear_attention.py:147-162) on the un-unfolded map: kvq is the
 * qkv_proj output [B*H*W][2C+8] = key | value | query | 7 zero columns; pixels (h, w) with equal (h % ph, w % pw) form the
 * reference's [b, :, p, :] slice of F.unfold (cvnets/modules/mobilevit_block.py:526-540).  out[B*H*W][C] = relu(value) *
 * sum_n softmax_n(query) * key; cv[B*ph*pw][C] (the context vectors) is saved for bwd.  bwd writes dkvq in the same layout. */
int cvh_linattn_fwd(int dtype, const void* kvq, void* out, float* cv, int B, int H, int W, int ph, int pw, int C, void* stream);
int cvh_linattn_bwd(int dtype, const void* kvq, const float* cv, const void* dout, void* dkvq, int B, int H, int W, int ph, int pw, int C,
                    void* stream);

/* ---- token plumbing for ViT / CLIP --------------------------------------------------------------- */
/* out[b][0] = cls, out[b][1+n] = patch[b][n] + pos[n]  (cls == NULL: no class token).  VisionTransformer.extract_patch_embeddings,
 * cvnets/models/classification/vit.py:480-509.  bwd: dpatch = rows of dout; dcls / dpos via cvh_batch_sum over the batch. */
int cvh_vit_embed_fwd(int dtype, const void* patch, const float* pos, const float* cls, void* out, int B, int N, int E, void* stream);
int cvh_vit_embed_bwd(int dtype, const void* dout, void* dpatch, int B, int N, int E, int has_cls, void* stream);
int cvh_batch_sum(int dtype, const void* x, float* out, int B, long long L, int accumulate, void* stream);
/* dst[r][0:C] = src[r][0:C] with independent row strides (class-token rows, vit.py:562-565) */
int cvh_rows_copy(int dtype, const void* src, void* dst, long long rows, int C, long long src_stride, long long dst_stride, void* stream);
/* text tower: out[r] = table[tok[r]] + pos[r % S] (cvnets/text_encoders/transformer.py:321-341); bwd scatter-adds into dtable */
int cvh_embed_lookup_fwd(int dtype, const long long* tok, const float* table, const float* pos, void* out, long long rows, int S, int E,
                         void* stream);
int cvh_embed_lookup_bwd(int dtype, const long long* tok, const void* dout, float* dtable, long long rows, int E, long long padding_idx,
                         void* stream);

/* ---- LayerNorm over channels -------------------------------------------------------------------- */
/* Replaces nn.LayerNorm, channel-last branch (cvnets/layers/normalization/layer_norm.py:67-68). */
int cvh_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      long long rows, int C, float eps, void* stream);
int cvh_layernorm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                      float* part, long long rows, int C, void* stream);
/* Channel-FIRST branch of the same class on a feature map (layer_norm.py:53-66: x.shape[1] == C, x.ndim > 2): per pixel, over its channels,
 * y = beta[c] + gamma[c] * (x - u) / (std + eps) with the biased std - on the NHWC storage a row-wise normalisation of [rows = pixels][C].
 * rstd[] = 1 / (std + eps); part = cvh_ln_bwd_rows(rows) partial rows [2][C] of dgamma / dbeta. */
int cvh_layernorm_cf_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                         long long rows, int C, float eps, void* stream);
int cvh_layernorm_cf_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                         float* part, long long rows, int C, float eps, void* stream);
/* the same with dx += dres: the gradient arriving over the residual branch that forks off in front of the LayerNorm (x -> LN(x) and
 * x -> ... + x, cvnets/modules/transformer.py:139-155) joins inside this kernel instead of in a separate elementwise add */
int cvh_layernorm_bwd_res(int dtype, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                          float* part, long long rows, int C, const void* dres, void* stream);
/* cvh_layernorm_bwd_res that ALSO stores dxd = dx with the keep mask of the Dropout in front of this LayerNorm applied (x = res +
 * Dropout(linear(h)), cvnets/modules/transformer.py:140-155): exactly cvh_dropout(dx) with (drop_p, seed, stream_id), from the same pass.
 * cvh_ln_bwd_drop_ok(C) == 0: not covered for this width (call cvh_dropout on dx). */
int cvh_ln_bwd_drop_ok(int C);
int cvh_layernorm_bwd_res_drop(int dtype, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                               float* part, long long rows, int C, const void* dres, void* dxd, float drop_p,
                               const unsigned long long* seed, unsigned int stream_id, void* stream);
int cvh_ln_bwd_rows(long long rows); /* rows of part[rows][2][C] (dgamma | dbeta) */

/* ---- fused multi-head self-attention ------------------------------------------------------------ */
/* qkv [rows][3*h*c] -> out [rows][h*c]; sequences addressed through the MobileViT unfold map
 * (ph,pw,n_w,H,W) or contiguously (ph=pw=1,H=1,W=n_w=S).  Replaces MultiHeadAttention.forward_default
 * lines 148-233 (cvnets/layers/multi_head_attention.py) and MobileViTBlock.unfolding/folding
 * (cvnets/modules/mobilevit_block.py:186-267).  lse [nseq][h][S] float32 is saved for backward. */
int cvh_attn_fwd(int dtype, const void* qkv, void* out, float* lse, const unsigned char* kpm, int nseq, int S, int h, int c,
                 int ph, int pw, int n_w, int H, int W, float scaling, int causal, void* stream);
int cvh_attn_bwd(int dtype, const void* qkv, const void* out, const void* dout, void* dqkv, const float* lse, float* dsum,
                 const unsigned char* kpm, int nseq, int S, int h, int c, int ph, int pw, int n_w, int H, int W, float scaling,
                 int causal, void* stream);
/* The same with dropout on the attention probabilities (MultiHeadAttention.attn_dropout, cvnets/layers/multi_head_attention.py:217-218):
 * O = dropout(softmax(S)) V.  The keep mask is a counter-based function of (*seed, stream_id, sequence, head, query, key) — the generator
 * of cvh_dropout — regenerated by the backward kernels, never stored; drop_p == 0 is cvh_attn_fwd / cvh_attn_bwd. */
int cvh_attn_fwd_drop(int dtype, const void* qkv, void* out, float* lse, const unsigned char* kpm, int nseq, int S, int h, int c,
                      int ph, int pw, int n_w, int H, int W, float scaling, int causal, float drop_p,
                      const unsigned long long* seed, unsigned int stream_id, void* stream);
int cvh_attn_bwd_drop(int dtype, const void* qkv, const void* out, const void* dout, void* dqkv, const float* lse, float* dsum,
                      const unsigned char* kpm, int nseq, int S, int h, int c, int ph, int pw, int n_w, int H, int W, float scaling,
                      int causal, float drop_p, const unsigned long long* seed, unsigned int stream_id, void* stream);

/* The same with a general ADDITIVE attention mask (cvnets/layers/multi_head_attention.py:197-208: `attn = attn + attn_mask`, mask
 * [N, S, T]; pinned by the reference's tests/modules/test_transformer.py): bias float32, natural units, -inf allowed, [S][S] shared by every
 * sequence (bias_stride 0) or [nseq][S][S] (bias_stride >= S*S elements); added to the scaled scores before the softmax in the forward and
 * in both recomputations of the backward.  bias == NULL is cvh_attn_fwd_drop / cvh_attn_bwd_drop; `causal` stays the generated fast path. */
int cvh_attn_fwd_mask(int dtype, const void* qkv, void* out, float* lse, const unsigned char* kpm, const float* bias, long long bias_stride,
                      int nseq, int S, int h, int c, int ph, int pw, int n_w, int H, int W, float scaling, int causal, float drop_p,
                      const unsigned long long* seed, unsigned int stream_id, void* stream);
int cvh_attn_bwd_mask(int dtype, const void* qkv, const void* out, const void* dout, void* dqkv, const float* lse, float* dsum,
                      const unsigned char* kpm, const float* bias, long long bias_stride, int nseq, int S, int h, int c, int ph, int pw,
                      int n_w, int H, int W, float scaling, int causal, float drop_p, const unsigned long long* seed, unsigned int stream_id,
                      void* stream);

/* ---- data-parallel exchange on a communicator of its own (RCCL over xGMI) --------------------------------------------------------
 * Replaces what the reference reaches through torch.distributed: the rendezvous + communicator creation of utils/ddp_utils.py:47-89
 * (init_process_group("nccl") and the dummy all_reduce at :84-85), the gradient averaging of DistributedDataParallel (main_train.py:91-96),
 * the parameter / buffer broadcast of its constructor and forward, and the feature gather of the contrastive loss
 * (loss_fn/multi_modal_img_text/contrastive_loss_clip.py:144-172 through utils/tensor_utils.py:121-122).
 * One process per GPU; the device that is current at cvh_comm_init is the communicator's device.  `id128` = 128 opaque bytes produced on
 * rank 0 and carried to the other ranks by the launcher's own key-value rendezvous (cvnets_amd/comm.py uses the TCP store of env://).
 * Collectives are enqueued on `stream` (stream-ordered, hipGraph-capturable); buffers are device pointers; dtype = CVH_DT_F32 / CVH_DT_BF16,
 * or CVH_COMM_BYTES for the two that only move data (broadcast, all-gather: counts are then in bytes).
 * librccl is opened at run time on first use: single-GPU users of this library never load it.
 * Returns 0, a negative code (-2 arguments, -3 librccl unavailable), a hipError_t, or 1000 + ncclResult_t. */
#define CVH_COMM_BYTES 100
int cvh_comm_available(void);                     /* 1 when librccl could be opened and has every entry point used here */
int cvh_comm_unique_id(void* id128);              /* rank 0: ncclGetUniqueId */
int cvh_comm_init(void** comm, int world, int rank, const void* id128);  /* every rank: ncclCommInitRank on the current device */
int cvh_comm_destroy(void* comm);
int cvh_comm_world(void* comm);
int cvh_comm_rank(void* comm);
/* in place: buf = sum over ranks (average != 0: mean over ranks, formed by the collective itself) */
int cvh_comm_allreduce(void* comm, void* buf, long long count, int dtype, int average, void* stream);
int cvh_comm_broadcast(void* comm, void* buf, long long count, int dtype, int root, void* stream);
/* recv[world * count_per_rank] = concatenation of every rank's send[count_per_rank], in rank order */
int cvh_comm_allgather(void* comm, const void* send, void* recv, long long count_per_rank, int dtype, void* stream);
/* recv[count_per_rank] = slice `rank` of the sum over ranks of send[world * count_per_rank] */
int cvh_comm_reducescatter(void* comm, const void* send, void* recv, long long count_per_rank, int dtype, void* stream);
/* launches so far: out[0..3] = all-reduce, broadcast, all-gather, reduce-scatter (long long[4]); reset != 0 clears */
int cvh_comm_counters(int reset, long long* out);

/* experiment knob for tools/kernel_bench.py (A/B of kernel variants in one process); never needed for correct results */
int cvh_set_tuning(int key, int value);

#ifdef __cplusplus
}
#endif
#endif /* CVNETS_HIP_H_ */

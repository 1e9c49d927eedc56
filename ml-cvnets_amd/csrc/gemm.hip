// Implicit-GEMM convolution / linear kernels on MFMA (gfx950), NHWC activations.
//
//   conv_gemm  (NT):  out[m, n] = epilogue( sum_k A(m, k) * Wp[n, k] )
//        m = (b, ho, wo) output pixel, k = (kh, kw, c) with c contiguous (NHWC), Wp = packed weights
//        [N][KH*KW*Cin].  Pointwise conv / nn.Linear are the KH = KW = 1 special case (A = x[m, :]).
//        A can be the channel-concat of two tensors (MobileViT fusion conv reads cat(res, fm) without
//        materialising it).  Used for: forward of every dense conv / linear, and for dX of stride-1
//        convs / linears with the flipped-transposed weight pack.
//   gemm_tn  (dW):    dW[n, k] += sum_m dY[m, n] * A(m, k)      (reduction over pixels, split over m)
//
// Replaces (reference, ATen calls): nn.Conv2d.forward  cvnets/layers/conv_layer.py:18-66,254-255;
// F.linear cvnets/layers/linear_layer.py:90; and their autograd backward.
#include "common.hpp"
#include "cvnets_hip.h"

#include "gemm_params.hpp"

#include "conv_gemm.hpp"
#include "gemm_tn.hpp"

// dw[n][c][tap] (torch layout) = (or +=) sum over splits of part[split][n][tap*Cin + c].
// Block = 16 consecutive output elements x 16 split lanes (the split loop is the long dimension for pointwise convs).
__global__ __launch_bounds__(256) void gemm_dw_reduce_kernel(const float* __restrict__ part, int splits, int N, int Ktot, int Cin, int Cin_real,
                                                             int khw, float* __restrict__ dw, int accumulate) {
  __shared__ float red[16][17];
  const size_t total = (size_t)N * Ktot;
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  for (size_t base = (size_t)blockIdx.x * 16; base < total; base += (size_t)gridDim.x * 16) {
    const size_t idx = base + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < total) {
      int sp = sl;
      for (; sp + 48 < splits; sp += 64) {
        s0 += part[(size_t)sp * total + idx];
        s1 += part[(size_t)(sp + 16) * total + idx];
        s2 += part[(size_t)(sp + 32) * total + idx];
        s3 += part[(size_t)(sp + 48) * total + idx];
      }
      for (; sp < splits; sp += 16) s0 += part[(size_t)sp * total + idx];
    }
    red[sl][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && idx < total) {
      float s = 0.f;
#pragma unroll
      for (int l = 0; l < 16; ++l) s += red[l][e];
      const int k = (int)(idx % Ktot);
      const int n = (int)(idx / Ktot);
      const int tap = k / Cin, c = k - tap * Cin;
      if (c < Cin_real) {
        float* d = dw + ((size_t)n * Cin_real + c) * khw + tap;
        *d = accumulate ? *d + s : s;
      }
    }
    __syncthreads();
  }
}

std::atomic<long long> g_cg_launches{0}, g_cg_bytes{0};  // updated from autograd worker threads
static std::atomic<long long> g_fam[5][2];
void cvh_family_tally(int family, long long bytes) {
  if (family < 0 || family >= 5) return;
  g_fam[family][0] += 1;
  g_fam[family][1] += bytes;
}
extern "C" int cvh_family_counters(int family, int reset, long long* out) {
  if (family < 0 || family >= 5) return -2;
  if (out != nullptr) { out[0] = g_fam[family][0]; out[1] = g_fam[family][1]; }
  if (reset) { g_fam[family][0] = 0; g_fam[family][1] = 0; }
  return 0;
}
void gemm_stream_counters(int reset, long long* out2);  // gemm_stream.hip
extern "C" int cvh_stream_counters(int reset, long long* out) {
  if (out != nullptr) {
    gemm_stream_counters(0, out);
    out[2] = g_cg_launches;
    out[3] = g_cg_bytes;
  }
  if (reset) {
    gemm_stream_counters(1, nullptr);
    g_cg_launches = 0;
    g_cg_bytes = 0;
  }
  return 0;
}

extern "C" int cvh_conv_gemm_grid_rows(int M, int N) {
  // number of stats-partial rows conv_gemm writes for an (M, N) problem (== gridDim.x)
  (void)N;
  int mt = (M + 127) / 128;
  const int cap = cvh_tune_get(CVH_TUNE_GEMM_GRID);
  return mt < cap ? mt : cap;
}

/* 1: a plain linear of these sizes, called with act = CVH_ACT_GELU_D, runs on a kernel that stores the derivative (bf16 only) */
extern "C" int cvh_conv_gemm_takes_gelu_d(long long M, int K, int N) {
  ConvGemmParams p;
  p.src1 = nullptr; p.src2 = nullptr; p.C1 = K; p.C2 = 0; p.wgt = nullptr; p.out = nullptr;
  p.B = 1; p.H = 1; p.W = (int)M; p.Ho = 1; p.Wo = (int)M; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.M = (int)M; p.N = N; p.Ktot = K;
  p.bias = nullptr; p.act = CVH_ACT_GELU_D; p.save_pre = nullptr; p.actgrad_aux = nullptr; p.actgrad_act = 0;
  p.residual = nullptr; p.drop_p = 0.f; p.seed = nullptr; p.stream_id = 0; p.stats_part = nullptr;
  p.m_tiles = 0;
  conv_gemm_params_no_fx(p);
  p.sc_s = 0; p.sc_KW = p.sc_C = p.sc_H = p.sc_W = p.sc_Ho = p.sc_Wo = 0;
  if (M <= 0 || M > 0x7fffffffLL || (K % 8) || (N % 8)) return 0;
  return (!gemm_stream_eligible(p) && gemm_big_eligible(p)) ? 1 : 0;
}

extern "C" int cvh_conv_gemm(int dtype, const void* src1, const void* src2, int C1, int C2, const void* wgt, void* out,
                             int B, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int dil, int N,
                             const float* bias, int act, void* save_pre, const void* actgrad_aux, int actgrad_act,
                             const void* residual, float drop_p, const unsigned long long* seed, unsigned int stream_id,
                             float* stats_part, void* stream) {
  if ((C1 % 8) != 0 || (C2 % 8) != 0 || C1 <= 0 || (N % 8) != 0) return -2;
  if (src2 == nullptr && C2 != 0) return -2;
  ConvGemmParams p;
  p.src1 = src1; p.src2 = src2; p.C1 = C1; p.C2 = C2; p.wgt = wgt; p.out = out;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.M = B * Ho * Wo; p.N = N; p.Ktot = KH * KW * (C1 + C2);
  p.bias = bias; p.act = act; p.save_pre = save_pre; p.actgrad_aux = actgrad_aux; p.actgrad_act = actgrad_act;
  p.residual = residual; p.drop_p = drop_p; p.seed = seed; p.stream_id = stream_id; p.stats_part = stats_part;
  p.m_tiles = 0;
  conv_gemm_params_no_fx(p);
  p.sc_s = 0; p.sc_KW = p.sc_C = p.sc_H = p.sc_W = p.sc_Ho = p.sc_Wo = 0;
  if (p.M <= 0 || N <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (act == CVH_ACT_GELU_D) {  // the stored-derivative epilogue exists in the large-tile kernels only (the caller checks cvh_conv_gemm_takes_gelu_d)
    if (dtype != CVH_DT_BF16 || gemm_stream_eligible(p) || !gemm_big_eligible(p)) return -2;
    return launch_gemm_big(p, st);
  }
  if (dtype == CVH_DT_BF16 && conv3x3_eligible(p)) return launch_conv3x3(p, cvh_conv_gemm_grid_rows(p.M, N), st);  // MobileViT-block 3x3 convs
  if (dtype == CVH_DT_BF16 && gemm_stream_eligible(p)) return launch_gemm_stream(p, st);  // short-K token linears / 1x1 convs (MobileViT blocks)
  if (dtype == CVH_DT_BF16 && gemm_big_eligible(p)) return launch_gemm_big(p, st);  // transformer-sized linears (ViT-B / CLIP)
  int nf = choose_nf(N);
  if (const int fill = cvh_tune_get(CVH_TUNE_GEMM_FILL); fill > 0) {
    // small-M problems (the 128-image shard of an 8-GPU run: layer_4 / layer_5 convolutions with 8 k - 32 k rows): the widest N tile leaves
    // most CUs without a workgroup — narrower tiles re-read the (L2-resident) A panel but fill the chip
    const int mt = (p.M + 127) / 128;
    while (nf > 1 && (long long)mt * ((N + 32 * nf - 1) / (32 * nf)) < fill) --nf;
  }
  const bool bk64 = p.Ktot >= 64;
  if (dtype == CVH_DT_BF16 && src2 == nullptr && KH == 1 && KW == 1 && stride == 1 && pad == 0 && p.Ktot <= 64 && p.M >= 4096 &&
      !cvh_tune_get(CVH_TUNE_NO_WAVE_PRIVATE)) {
    // Plain pointwise GEMMs whose whole K is ONE step (the 1x1 convs of the high-resolution stages): weights resident in LDS, every wave
    // streams its own rows, no workgroup barrier in the tile loop (conv_gemm.hpp, WP): 3.5 -> 4.4 TB/s on [4.2M x 64] x [64 x 256].
    // (Measured and not adopted: the same with K tiles of 128 / 160 for the d = 96 ... 160 transformer linears — 8-10 operand vectors
    // per lane in flight across the epilogue spill, the N tile has to shrink to 64-96 columns, and the step time does not move.)
    return p.Ktot <= 32 ? dispatch_conv_gemm_nf<bf16_t, 32, 0, 1>(p, nf, st) : dispatch_conv_gemm_nf<bf16_t, 64, 0, 1>(p, nf, st);
  }
  if (dtype == CVH_DT_BF16) {
    // narrow-output pointwise GEMMs whose K is a little over one 64-wide step (the channel-concat dX of a 1x1 expansion conv:
    // K = 5 * Cin = 80 / 160): a 128-wide K tile keeps the weights resident for K <= 128 and halves the barriers per byte
    if (nf <= 2 && p.Ktot > 64 && p.Ktot <= 256 && (p.Ktot % 64) != 0 && p.KH == 1 && p.KW == 1 && cvh_tune_get(8) == 0)
      return nf == 1 ? launch_conv_gemm<bf16_t, 1, 128, 0>(p, st) : launch_conv_gemm<bf16_t, 2, 128, 0>(p, st);
    return bk64 ? dispatch_conv_gemm_nf<bf16_t, 64, 0>(p, nf, st) : dispatch_conv_gemm_nf<bf16_t, 32, 0>(p, nf, st);
  } else if (dtype == CVH_DT_F32) {
    return dispatch_conv_gemm_nf<float, 32, 0>(p, nf, st);
  }
  return -1;
}

// dX of a non-overlapping strided conv (kernel == stride, pad 0; the ViT patch-embedding convs, cvnets/models/classification/vit.py:89-123):
// every input pixel belongs to exactly one window, so dX = scatter( dY[M][Cout] x Wp3[(kh,kw,c)][Cout]^T ) — one GEMM whose
// epilogue writes each 8-channel chunk to its pixel.  wgt = mode-3 pack.  Pixels outside every window (H > Ho*s) are zeroed first.
extern "C" int cvh_conv_dx_patch(int dtype, const void* dy, const void* wgt, void* dx, int B, int Ho, int Wo, int Cout, int KH, int KW,
                                 int stride, int Cin, int H, int W, void* stream) {
  if ((Cout % 8) != 0 || (Cin % 8) != 0 || KH != stride || KW != stride || Ho * stride > H || Wo * stride > W) return -2;
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = dtype == CVH_DT_BF16 ? 2 : 4;
  if (Ho * stride != H || Wo * stride != W) {
    hipError_t e = hipMemsetAsync(dx, 0, (size_t)B * H * W * Cin * esz, st);
    if (e != hipSuccess) return (int)e;
  }
  ConvGemmParams p;
  p.src1 = dy; p.src2 = nullptr; p.C1 = Cout; p.C2 = 0; p.wgt = wgt; p.out = dx;
  p.B = B * Ho * Wo; p.H = 1; p.W = 1; p.Ho = 1; p.Wo = 1; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.M = B * Ho * Wo; p.N = KH * KW * Cin; p.Ktot = Cout;
  p.bias = nullptr; p.act = 0; p.save_pre = nullptr; p.actgrad_aux = nullptr; p.actgrad_act = 0; p.residual = nullptr;
  p.drop_p = 0.f; p.seed = nullptr; p.stream_id = 0; p.stats_part = nullptr; p.m_tiles = 0;
  conv_gemm_params_no_fx(p);
  p.sc_s = stride; p.sc_KW = KW; p.sc_C = Cin; p.sc_H = H; p.sc_W = W; p.sc_Ho = Ho; p.sc_Wo = Wo;
  if (p.M <= 0) return 0;
  const int nf = choose_nf(p.N);
  const bool bk64 = p.Ktot >= 64;
  if (dtype == CVH_DT_BF16) return bk64 ? dispatch_conv_gemm_nf<bf16_t, 64, 0>(p, nf, st) : dispatch_conv_gemm_nf<bf16_t, 32, 0>(p, nf, st);
  if (dtype == CVH_DT_F32) return dispatch_conv_gemm_nf<float, 32, 0>(p, nf, st);
  return -1;
}

static bool tn_skinny_shape(int M, int N, int Ktot);
// shapes of the direct-to-LDS dW kernel (gemm_big.hip: gemm_tn128_kernel; its eligibility also needs a plain pointwise bf16 problem)
static bool tn_big_shape(int M, int N, int Ktot) {
  return M >= 2048 && N >= 64 && (N % 8) == 0 && Ktot >= 64 && (Ktot % 8) == 0 && !tn_skinny_shape(M, N, Ktot);
}

// one small output tile under millions of rows (gemm_tn.hpp: gemm_tn_skinny_kernel).  A property of the SHAPE alone, so that the scratch
// size query needs no more than (M, N, Ktot): such problems always produce 4 partial rows per split; launches the skinny kernel cannot
// take (fp32, im2col, two sources, operand transforms) run gemm_tn_kernel with four times the splits instead.
static bool tn_skinny_shape(int M, int N, int Ktot) {
  if (cvh_tune_get(CVH_TUNE_NO_SKINNY)) return false;
  const int nt = (N + 31) / 32, kt = (Ktot + 31) / 32;
  return M >= 32768 && N <= 128 && Ktot <= 64 && nt * kt <= 4;
}

static void tn_plan(int M, int N, int Ktot, int* out_tiles, int* k_tiles, int* splits, int* mps) {
  // the direct-to-LDS kernel's tile: 256 x 256 where that moves fewer operand columns per row (gemm_big.hip: gemm_tn256_shape), else 128 x 128
  const bool t256 = tn_big_shape(M, N, Ktot) && cvh_tune_get(CVH_TUNE_BIG_GEMM) && gemm_tn256_shape(N, Ktot);
  const int tw = t256 ? 256 : 128;
  const int n_tiles = (N + tw - 1) / tw;
  *k_tiles = (Ktot + tw - 1) / tw;
  *out_tiles = n_tiles * *k_tiles;
  int sp;
  if (tn_big_shape(M, N, Ktot)) {
    // transformer-sized dW (gemm_tn128_kernel, 2 workgroups per CU = 512 slots; gemm_tn256_kernel: one 8-wave workgroup per CU = 256 slots
    // whose 64-row step moves twice the bytes): pick the split count that minimises
    // (rounds of workgroups) x (64-row steps per split) + the cost of summing the partial tiles.  Plain "enough workgroups"
    // planning lands on e.g. 144 tiles x 8 splits = 2.25 rounds, i.e. a third of the machine idle in the last round.
    const int slots = t256 ? 256 : 512, total_steps = (M + 63) / 64;
    double best = 1e30;
    sp = 1;
    for (int s = 1; s <= 512 && s * 4 <= total_steps + 3; ++s) {  // at least ~4 steps (256 rows) per split
      const int rounds = (*out_tiles * s + slots - 1) / slots;
      const int steps = (total_steps + s - 1) / s;
      const double t = (double)rounds * steps * (t256 ? 2.2 : 1.8) + (double)(s + 1) * (double)N * Ktot * 4.0 / 3.0e6;  // microseconds
      if (t < best) { best = t; sp = s; }
    }
  } else {
    // enough splits over M to put ~2 deep-prefetching workgroups on every CU
    // (K-heavy 3x3 problems have many output tiles and long per-split MFMA chains: they like twice as many workgroups)
    const int target_wgs = cvh_tune_get(CVH_TUNE_TN_WGS) * (*out_tiles >= 8 ? 2 : 1);
    sp = (target_wgs + *out_tiles - 1) / *out_tiles;
  }
  TnRowsGeom rg;
  if (tn_big_shape(M, N, Ktot) && cvh_tune_get(CVH_TUNE_BIG_GEMM) && !t256 && gemm_tn_rows_plan(M, N, Ktot, &rg)) {
    // whole-row workgroups (gemm_rows.hip): one partial row per workgroup along M; a launch of this shape that kernel cannot take runs the
    // tiled kernels on the same splits
    *splits = rg.splits;
    *mps = rg.m_per_split;
    return;
  }
  int max_splits = (M + 255) / 256;
  if (tn_skinny_shape(M, N, Ktot)) {  // gemm_tn_skinny_kernel: 4 partial rows per workgroup, >= 4 stages per wave
    const int wgs = cvh_tune_get(CVH_TUNE_SKINNY_WGS) > 0 ? cvh_tune_get(CVH_TUNE_SKINNY_WGS) : 512;
    sp = wgs;
    max_splits = (M + 511) / 512;
  }
  if (sp > max_splits) sp = max_splits;
  if (sp < 1) sp = 1;
  int m = (M + sp - 1) / sp;
  if (tn_skinny_shape(M, N, Ktot)) m = ((m + 127) / 128) * 128;  // four interleaved 32-row stages
  m = ((m + 63) / 64) * 64;  // whole 64-row reduction steps (gemm_tn128_kernel); also a multiple of the 32-row step of gemm_tn_kernel
  *splits = (M + m - 1) / m;
  *mps = m;
}

// part[split][total] -> dw[total] (linear layers: torch layout == GEMM layout), 16 bytes per lane, splits summed in registers
__global__ __launch_bounds__(256) void gemm_dw_reduce_linear_kernel(const float* __restrict__ part, int splits, size_t total4, float* __restrict__ dw,
                                                                    int accumulate) {
  const float4* __restrict__ p4 = reinterpret_cast<const float4*>(part);
  float4* __restrict__ d4 = reinterpret_cast<float4*>(dw);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    float4 s = accumulate ? d4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = 0;
    for (; sp + 3 < splits; sp += 4) {  // four independent loads in flight
      const float4 a = p4[(size_t)sp * total4 + i], b = p4[(size_t)(sp + 1) * total4 + i];
      const float4 c = p4[(size_t)(sp + 2) * total4 + i], d = p4[(size_t)(sp + 3) * total4 + i];
      s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
      s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; sp < splits; ++sp) {
      const float4 a = p4[(size_t)sp * total4 + i];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    d4[i] = s;
  }
}

extern "C" long long cvh_gemm_dw_scratch_elems(int M, int N, int Ktot) {
  if (M <= 0) return 0;
  int ot, kt, sp, mps;
  tn_plan(M, N, Ktot, &ot, &kt, &sp, &mps);
  return (long long)sp * (tn_skinny_shape(M, N, Ktot) ? 4 : 1) * N * Ktot;
}

// 1 when cvh_gemm_dw_bias can emit the column sums of dY from the dW kernel it would pick (everything but the transformer-sized
// direct-to-LDS kernel of gemm_big.hip, whose operands never pass through registers)
extern "C" int cvh_gemm_dw_folds_bias(int dtype, int M, int N, int Ktot) {
  // the direct-to-LDS kernel folds it only through a padded column of ones: needs K % 128 != 0 (gemm_big.hip)
  if (dtype == CVH_DT_BF16 && cvh_tune_get(CVH_TUNE_BIG_GEMM) && tn_big_shape(M, N, Ktot))
    return (gemm_tn256_shape(N, Ktot) || gemm_tn_rows_plan(M, N, Ktot, nullptr)) ? 1 : ((Ktot % 128) != 0 ? 1 : 0);  // the 256 x 256 kernel sums the columns itself where K leaves no padded column
  return M > 0 ? 1 : 0;
}

static int gemm_dw_finish(const GemmTNParams& p, int splits, int N, int KH, int KW, int C1, int C2, int Cin_real, float* dw, int accumulate,
                          hipStream_t st);

// Scratch size of cvh_gemm_dw / cvh_gemm_dw_bias for a CONVOLUTION geometry: the 3x3 stride-1 convs of the MobileViT blocks go through
// conv3x3_dw_kernel (conv3x3_dw.hip), which writes its own number of partial rows; everything else is cvh_gemm_dw_scratch_elems.
extern "C" long long cvh_gemm_dw_scratch_elems_conv(int dtype, int B, int H, int W, int Ho, int Wo, int C1, int C2, int KH, int KW, int stride,
                                                    int pad, int dil, int N, int with_bias) {
  GemmTNParams p;
  p.dy = nullptr; p.src1 = nullptr; p.src2 = C2 ? reinterpret_cast<const void*>(1) : nullptr; p.C1 = C1; p.C2 = C2; p.dw = nullptr;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.M = B * Ho * Wo; p.N = N; p.Ktot = KH * KW * (C1 + C2); p.Cin_real = C1 + C2;
  p.dy_xf = make_xf(nullptr); p.x_xf = make_xf(nullptr);
  p.bias_part = with_bias ? reinterpret_cast<float*>(1) : nullptr; p.part = nullptr;
  if (dtype == CVH_DT_BF16 && conv3x3_dw_eligible(p)) return (long long)conv3x3_dw_rows(p) * N * p.Ktot;
  return cvh_gemm_dw_scratch_elems(p.M, N, p.Ktot);
}

static int gemm_dw_run(int dtype, GemmTNParams p, int N, int KH, int KW, int C1, int C2, int Cin_real, float* dw, float* scratch,
                       long long scratch_elems, int accumulate, hipStream_t st, bool fx) {
  if (p.M <= 0) return 0;
  if (!fx && dtype == CVH_DT_BF16 && scratch != nullptr && conv3x3_dw_eligible(p)) {  // 3x3 convs of the MobileViT blocks: halo tile in LDS, X read once
    const int rows3 = conv3x3_dw_rows(p);
    const long long row_elems = (long long)N * p.Ktot;
    if (scratch_elems >= rows3 * row_elems) {
      p.part = scratch;
      const int rc = launch_conv3x3_dw(p, st);
      if (rc) return rc;
      const long long have = scratch_elems / row_elems;  // a caller that sized the scratch with the generic query sums `have` rows
      if (dw == nullptr && have > rows3) {
        hipError_t e = hipMemsetAsync(scratch + rows3 * row_elems, 0, (size_t)(have - rows3) * row_elems * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
      }
      return gemm_dw_finish(p, rows3, N, KH, KW, C1, C2, Cin_real, dw, accumulate, st);
    }
  }
  int out_tiles, splits, mps;
  tn_plan(p.M, N, p.Ktot, &out_tiles, &p.k_tiles, &splits, &mps);
  p.m_per_split = mps;
  // scratch path: every split writes its partial tile, one reduce kernel sums them (assign or accumulate into dw).
  // Without scratch: fp32 atomics into dw, which the caller must have zeroed (accumulate semantics only).
  p.part = nullptr;
  const bool skinny_shape = tn_skinny_shape(p.M, N, p.Ktot);
  const int rows = splits * (skinny_shape ? 4 : 1);
  if (scratch != nullptr) {
    if (scratch_elems < (long long)rows * N * p.Ktot) return -2;
    p.part = scratch;
  } else if (!accumulate || skinny_shape || p.bias_part != nullptr) {
    return -2;
  }
  // tn_plan counted 256-wide tiles where the direct-to-LDS kernel would use them (a property of the shape); a launch of that shape which
  // that kernel cannot take (fp32, operand transforms, im2col, two sources, no scratch) runs gemm_tn_kernel on 128-wide tiles: same
  // splits, same scratch, the tile grid of ITS tiling
  if (!(dtype == CVH_DT_BF16 && !fx && p.part != nullptr && gemm_tn_big_eligible(p))) {
    p.k_tiles = (p.Ktot + 127) / 128;
    out_tiles = ((N + 127) / 128) * p.k_tiles;
  }
  // plain operands, or a plain dY with act(c0 * x + c1) on X (the projection dW of the fused InvertedResidual blocks)
  const bool fx_ok = !fx || (p.dy_xf.mode == 0 && p.x_xf.mode == 1 && p.bias_part == nullptr);
  const bool skinny = skinny_shape && fx_ok && dtype == CVH_DT_BF16 && KH * KW == 1 && p.stride == 1 && p.pad == 0 && C2 == 0;
  if (skinny && fx) {
    const int nt = (N + 31) / 32, kt = (p.Ktot + 31) / 32;
    const dim3 g(splits), b(256);
#define SKINNY_XF(NT_, KT_, PF_) \
  if (nt == NT_ && kt == KT_) hipLaunchKernelGGL((gemm_tn_skinny_kernel<NT_, KT_, PF_, 0, 1>), g, b, 0, st, p);
    SKINNY_XF(1, 1, 4) SKINNY_XF(2, 1, 4) SKINNY_XF(3, 1, 3) SKINNY_XF(4, 1, 3) SKINNY_XF(1, 2, 3) SKINNY_XF(2, 2, 3)
#undef SKINNY_XF
    CVH_CHECK_LAUNCH();
    return gemm_dw_finish(p, rows, N, KH, KW, C1, C2, Cin_real, dw, accumulate, st);
  }
  if (skinny) {
    const int nt = (N + 31) / 32, kt = (p.Ktot + 31) / 32;
    const dim3 g(splits), b(256);
#define SKINNY(NT_, KT_, PF_)                                                                                       \
  if (nt == NT_ && kt == KT_) {                                                                                      \
    if (p.bias_part) hipLaunchKernelGGL((gemm_tn_skinny_kernel<NT_, KT_, PF_, 1>), g, b, 0, st, p);                  \
    else hipLaunchKernelGGL((gemm_tn_skinny_kernel<NT_, KT_, PF_, 0>), g, b, 0, st, p);                             \
  }
    SKINNY(1, 1, 4) SKINNY(2, 1, 4) SKINNY(3, 1, 3) SKINNY(4, 1, 3) SKINNY(1, 2, 4) SKINNY(2, 2, 3)
#undef SKINNY
    CVH_CHECK_LAUNCH();
    return gemm_dw_finish(p, rows, N, KH, KW, C1, C2, Cin_real, dw, accumulate, st);
  }
  if (skinny_shape) p.m_per_split = mps / 4;  // same partial-row count from the general kernel
  dim3 grid(out_tiles, rows);
  if (fx && p.dy_xf.mode == 2) {
    return -2;  // not instantiated (see gemm_fx.hip): the linear case goes through cvh_bn_dw_combine
  } else if (fx) {
    if (p.dy_xf.mode != 0) return -2;
    if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, 0, 1>), grid, dim3(256), 0, st, p);
    else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((gemm_tn_kernel<float, 0, 1>), grid, dim3(256), 0, st, p);
    else return -1;
  } else if (dtype == CVH_DT_BF16 && p.part != nullptr && cvh_tune_get(CVH_TUNE_BIG_GEMM) && !gemm_tn256_shape(p.N, p.Ktot) && gemm_tn_rows_eligible(p)) {
    const int rc = launch_gemm_tn_rows(p, st);  // MobileViT-sized token linears under >= 128 k rows: whole rows per workgroup
    if (rc) return rc;
  } else if (dtype == CVH_DT_BF16 && p.part != nullptr && gemm_tn_big_eligible(p)) {  // transformer-sized linears (ViT-B / CLIP)
    if (p.bias_part != nullptr && !gemm_tn256_shape(p.N, p.Ktot) && (p.Ktot % 128) == 0) return -2;  // cvh_gemm_dw_folds_bias() says so
    const int rc = launch_gemm_tn_big(p, splits, st);
    if (rc) return rc;
  } else if (dtype == CVH_DT_BF16) {
    const bool pw = KH * KW == 1 && p.stride == 1 && p.pad == 0;
    if (pw) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, 1, 0>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, 0, 0>), grid, dim3(256), 0, st, p);
  } else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((gemm_tn_kernel<float, 0, 0>), grid, dim3(256), 0, st, p);
  else return -1;
  CVH_CHECK_LAUNCH();
  return gemm_dw_finish(p, rows, N, KH, KW, C1, C2, Cin_real, dw, accumulate, st);
}

static int gemm_dw_finish(const GemmTNParams& p, int splits, int N, int KH, int KW, int C1, int C2, int Cin_real, float* dw, int accumulate,
                          hipStream_t st) {
  if (dw == nullptr) return p.part ? 0 : -2;  // partial tiles only: the caller sums the splits later (cvh_reduce_multi)
  if (p.part && KH * KW == 1 && Cin_real == p.Ktot && ((size_t)N * p.Ktot) % 4 == 0 && splits <= 64 && (size_t)N * p.Ktot >= 65536) {
    const size_t total4 = (size_t)N * p.Ktot / 4;
    int g = (int)((total4 + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(gemm_dw_reduce_linear_kernel, dim3(g), dim3(256), 0, st, p.part, splits, total4, dw, accumulate);
    CVH_CHECK_LAUNCH();
  } else if (p.part) {
    const size_t total = (size_t)N * p.Ktot;
    int g = (int)((total + 15) / 16);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(gemm_dw_reduce_kernel, dim3(g), dim3(256), 0, st, p.part, splits, N, p.Ktot, C1 + C2, Cin_real, KH * KW, dw, accumulate);
    CVH_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int cvh_gemm_dw_bias(int dtype, const void* dy, const void* src1, const void* src2, int C1, int C2, float* dw, float* bias_part,
                                int B, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int dil, int N,
                                int Cin_real, float* scratch, long long scratch_elems, int accumulate, void* stream) {
  if ((C1 % 8) != 0 || (C2 % 8) != 0 || (N % 8) != 0) return -2;
  GemmTNParams p;
  p.dy = dy; p.src1 = src1; p.src2 = src2; p.C1 = C1; p.C2 = C2; p.dw = dw;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.M = B * Ho * Wo; p.N = N; p.Ktot = KH * KW * (C1 + C2); p.Cin_real = Cin_real;
  p.dy_xf = make_xf(nullptr); p.x_xf = make_xf(nullptr);
  p.bias_part = bias_part;
  return gemm_dw_run(dtype, p, N, KH, KW, C1, C2, Cin_real, dw, scratch, scratch_elems, accumulate, (hipStream_t)stream, false);
}

extern "C" int cvh_gemm_dw(int dtype, const void* dy, const void* src1, const void* src2, int C1, int C2, float* dw,
                           int B, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int dil, int N,
                           int Cin_real, float* scratch, long long scratch_elems, int accumulate, void* stream) {
  return cvh_gemm_dw_bias(dtype, dy, src1, src2, C1, C2, dw, nullptr, B, H, W, Ho, Wo, KH, KW, stride, pad, dil, N, Cin_real, scratch,
                          scratch_elems, accumulate, stream);
}

// Pointwise dW with both operands transformed on load: dy = ca*g + cb*y + cc (BatchNorm input gradient formed on the fly), x =
// act(scale*x_raw + shift) (the normalised activation recomputed from the raw producer output) — see bnlink.hpp.
extern "C" int cvh_pw_gemm_dw_bn(int dtype, const void* dy, const cvh_operand_xf* dy_xf, const void* x, const cvh_operand_xf* x_xf, float* dw,
                                 long long M, int N, int K, int Cin_real, float* scratch, long long scratch_elems, int accumulate,
                                 void* stream) {
  if ((K % 8) != 0 || (N % 8) != 0 || M > 0x7fffffffLL) return -2;
  GemmTNParams p;
  p.dy = dy; p.src1 = x; p.src2 = nullptr; p.C1 = K; p.C2 = 0; p.dw = dw;
  p.B = (int)M; p.H = 1; p.W = 1; p.Ho = 1; p.Wo = 1; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.M = (int)M; p.N = N; p.Ktot = K; p.Cin_real = Cin_real;
  p.dy_xf = make_xf(dy_xf); p.x_xf = make_xf(x_xf);
  p.bias_part = nullptr;
  if (p.dy_xf.mode == 2 && p.dy_xf.src2 == nullptr) return -2;
  return gemm_dw_run(dtype, p, N, 1, 1, K, 0, Cin_real, dw, scratch, scratch_elems, accumulate, (hipStream_t)stream, true);
}

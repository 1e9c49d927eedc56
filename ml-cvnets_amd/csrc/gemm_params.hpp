// Parameter block shared by the implicit-GEMM kernels (gemm.hip) and the large-tile linear kernel (gemm_big.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "bnlink.hpp"

struct ConvGemmParams {
  const void* src1;
  const void* src2;
  int C1, C2;
  const void* wgt;
  void* out;
  int B, H, W, Ho, Wo, KH, KW, stride, pad, dil;
  int M, N, Ktot;
  const float* bias;
  int act;
  void* save_pre;
  const void* actgrad_aux;
  int actgrad_act;
  const void* residual;
  float drop_p;
  const unsigned long long* seed;
  unsigned int stream_id;
  float* stats_part;
  int m_tiles;
  // output scatter (dX of a non-overlapping strided conv, kernel == stride, pad 0): GEMM row m = (b, ho, wo) of the sc_Ho x sc_Wo
  // map, column n = (kh, kw, c) -> out[b][ho*sc_s + kh][wo*sc_s + kw][c] of a sc_H x sc_W x sc_C map.  sc_s == 0: plain [M][N] output.
  int sc_s, sc_KW, sc_C, sc_H, sc_W, sc_Ho, sc_Wo;
  // BatchNorm links (conv_gemm_kernel<..., FX = 1>, pointwise GEMMs only; bnlink.hpp)
  OperandXf a_xf;        // transform of the A operand on load
  int e_mode;            // 0: plain epilogue (stats_part: sum, sumsq of the stored values); 1: BatchNorm-backward epilogue (sum g, sum g*xhat)
  const void* e_aux;     // [M][N] raw output of the BatchNorm'd conv (e_mode 1)
  const float* e_stats;  // [4][N] its forward statistics (mean, invstd, scale, shift)
  int e_act;
};
inline void conv_gemm_params_no_fx(ConvGemmParams& p) {
  p.a_xf = make_xf(nullptr);
  p.e_mode = 0; p.e_aux = nullptr; p.e_stats = nullptr; p.e_act = 0;
}

struct GemmTNParams {
  const void* dy;
  const void* src1;
  const void* src2;
  int C1, C2;
  float* dw;
  int B, H, W, Ho, Wo, KH, KW, stride, pad, dil;
  int M, N, Ktot;
  int Cin_real;
  int m_per_split;
  int k_tiles;
  float* part;  // [splits][N][Ktot] partial products (no atomics); nullptr -> fp32 atomics straight into dw
  float* bias_part;  // optional [rows][N] column sums of dY over each partial row's m-range (the bias gradient of the same layer)
  OperandXf dy_xf, x_xf;  // gemm_tn_kernel<..., FX = 1>: operand transforms on load (pointwise only)
};

// gemm_big.hip
bool gemm_tn_big_eligible(const GemmTNParams& p);
bool gemm_tn256_shape(int N, int Ktot);  // the direct-to-LDS dW product runs on 256 x 256 tiles for this (N, K)
int launch_gemm_tn_big(const GemmTNParams& p, int splits, hipStream_t st);
bool gemm_big_eligible(const ConvGemmParams& p);
int launch_gemm_big(const ConvGemmParams& p, hipStream_t st);

// gemm_rows.hip: dW of the MobileViT-sized token linears under >= 128 k rows, whole rows per workgroup
struct TnRowsGeom {
  int n_parts, k_parts;  // workgroup columns over N and K (1 x 1 where the [N x K] block fits one workgroup's accumulators)
  int NP, KP;            // columns of dY / X per part
  int WN, WK, PN, PK;    // 16 waves as WN x WK, PN x PK 16 x 16 tiles each
  int py, px;            // LDS row pitch of the dY / X images in 16-byte chunks
  int stage_bytes, nstage, ipw;
  int splits, m_per_split;
  int dbg;
};
bool gemm_tn_rows_plan(int M, int N, int K, TnRowsGeom* out);  // a property of the shape (bf16 pointwise problems)
bool gemm_tn_rows_eligible(const GemmTNParams& p);
int launch_gemm_tn_rows(const GemmTNParams& p, hipStream_t st);

// gemm_stream.hip
bool gemm_stream_eligible(const ConvGemmParams& p);
int launch_gemm_stream(const ConvGemmParams& p, hipStream_t st);
// conv3x3.hip
bool conv3x3_eligible(const ConvGemmParams& p);
int launch_conv3x3(const ConvGemmParams& p, int rows, hipStream_t st);
// conv3x3_dw.hip
bool conv3x3_dw_eligible(const GemmTNParams& p);
int conv3x3_dw_rows(const GemmTNParams& p);
int launch_conv3x3_dw(const GemmTNParams& p, hipStream_t st);
bool gemm_stream_fx_eligible(const ConvGemmParams& p);
int launch_gemm_stream_fx(const ConvGemmParams& p, int rows, hipStream_t st);

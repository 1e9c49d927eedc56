// Pointwise conv / linear GEMMs with operand transforms and BatchNorm links (bnlink.hpp): the conv_gemm_kernel<..., FX = 1>
// instantiations and their entry point.  Forward and dX of the 1x1 convolutions inside conv-BN-act stacks
// (InvertedResidual.forward, cvnets/modules/mobilenetv2.py:231-235) without standalone BatchNorm passes.
#include "conv_gemm.hpp"

extern "C" int cvh_pw_gemm_bn(int dtype, const void* a, const cvh_operand_xf* a_xf, int K, const void* wgt, void* out, long long M, int N,
                              const void* residual, int e_mode, const void* e_aux, const float* e_stats, int e_act,
                              float* stats_part, void* stream) {
  if ((K % 8) != 0 || K <= 0 || (N % 8) != 0 || M > 0x7fffffffLL) return -2;
  if (e_mode == 1 && (e_aux == nullptr || e_stats == nullptr)) return -2;
  ConvGemmParams p;
  p.src1 = a; p.src2 = nullptr; p.C1 = K; p.C2 = 0; p.wgt = wgt; p.out = out;
  p.B = (int)M; p.H = 1; p.W = 1; p.Ho = 1; p.Wo = 1; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.M = (int)M; p.N = N; p.Ktot = K;
  p.bias = nullptr; p.act = 0; p.save_pre = nullptr; p.actgrad_aux = nullptr; p.actgrad_act = 0;
  p.residual = residual; p.drop_p = 0.f; p.seed = nullptr; p.stream_id = 0; p.stats_part = stats_part;
  p.m_tiles = 0;
  p.sc_s = 0; p.sc_KW = p.sc_C = p.sc_H = p.sc_W = p.sc_Ho = p.sc_Wo = 0;
  p.a_xf = make_xf(a_xf);
  if (p.a_xf.mode == 2 && p.a_xf.src2 == nullptr) return -2;
  p.e_mode = e_mode; p.e_aux = e_aux; p.e_stats = e_stats; p.e_act = e_act;
  if (p.M <= 0 || N <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int nf = choose_nf(N);
  const bool bk64 = p.Ktot >= 64;
  // mode 2 (two-source operand) is not instantiated for GEMMs: the only BatchNorm-input-gradient consumer on the path is linear and
  // takes the algebraic route of bnlink.hip (measured: the two-source kernels ran at 1.3-2.2 TB/s, the plain ones at 3.5-4.8)
  if (p.a_xf.mode == 2) return -2;
  if (dtype == CVH_DT_BF16 && gemm_stream_fx_eligible(p))   // plain operand + statistics / BatchNorm-backward epilogue: the 16-wave streaming kernel
    return launch_gemm_stream_fx(p, cvh_conv_gemm_grid_rows((int)M, N), st);
  if (dtype == CVH_DT_BF16) {
    const bool wp = p.Ktot <= (bk64 ? 64 : 32) && !cvh_tune_get(CVH_TUNE_NO_WAVE_PRIVATE);  // single K step: barrier-free wave-private staging
    if (wp) return bk64 ? dispatch_conv_gemm_nf<bf16_t, 64, 1, 1>(p, nf, st) : dispatch_conv_gemm_nf<bf16_t, 32, 1, 1>(p, nf, st);
    return bk64 ? dispatch_conv_gemm_nf<bf16_t, 64, 1>(p, nf, st) : dispatch_conv_gemm_nf<bf16_t, 32, 1>(p, nf, st);
  }
  if (dtype == CVH_DT_F32) return dispatch_conv_gemm_nf<float, 32, 1>(p, nf, st);
  return -1;
}

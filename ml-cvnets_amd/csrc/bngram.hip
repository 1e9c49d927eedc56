// bn_apply_gram_kernel: the BatchNorm apply that closes a conv -> BatchNorm (-> act) (+ residual) chain (ConvLayer2d.forward,
// cvnets/layers/conv_layer.py:254-255; the tail of InvertedResidual.forward, cvnets/modules/mobilenetv2.py:231-235) and, from the SAME pass
// over the tensor, what the NEXT InvertedResidual block needs of its input before it can start: the Gram matrix G = y^T y and the column
// sums s = 1^T y of the narrow block input (csrc/dwx.hip: the statistics of the expansion's BatchNorm follow from them, y1 = x W1^T being
// linear in x; the backward pass reuses G for the weight gradient).  Without this kernel the Gram matrix is a second full read of the
// tensor the apply pass has just written (gemm_tn_skinny_kernel on (x, x): 1.0 ms per MobileViT-S step at 1024 images).
//
//   y = act(x * scale[c] + shift[c]) (+ residual)           exactly bn_apply_kernel's arithmetic, 16-byte pieces, 1 KB per store instruction
//   G[i][j] = sum_rows y[r][i] y[r][j],  s[i] = sum_rows y[r][i]      from the bf16 values AS STORED, on the matrix pipe
//
// Waves are independent: a wave streams "super-steps" of NL x 64 pieces (32 / 64 / 128 rows for 64 / 32 / 16 channels), keeps the finished
// rows in a private LDS tile and multiplies the tile with itself — v_mfma_f32_16x16x32_bf16 with BOTH operands read "down the rows" by the
// LDS transpose read (ds_read_b64_tr_b16), upper-triangular 16 x 16 output tiles only.  The K slot <-> row assignment of a 32-row K step is
// (lane group l4, j) <-> row 4 l4 + j (j < 4), 16 + 4 l4 + (j - 4): a 32-lane pass of the transpose read then touches 8 CONSECUTIVE rows, and
// with a row pitch of (C / 2 + 8) dwords (C >= 32) they land on 8 disjoint bank octets.  No workgroup barrier in the loop; the next
// super-step's global loads are issued as soon as the current one's registers are packed, and fly under the stores and the MFMAs.
// The workgroup's partial [C*C + C] row (G mirrored to a full matrix, then s) is summed over workgroups by cvh_sum_partials (double).
#include "common.hpp"
#include "cvnets_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float bg_f32x4;
typedef short bg_v4s __attribute__((ext_vector_type(4)));

// waves per workgroup (two workgroups per CU): 64 channels hold 40 accumulator registers and run three waves per SIMD (168 registers)
template <int C> constexpr int bg_waves() { return C >= 64 ? 6 : 8; }
constexpr int BG_NL = 4;

template <int C> struct BgGeom {
  static constexpr int CG = C / 8;                     // 16-byte pieces per row
  static constexpr int RPS = BG_NL * 64 / CG;          // rows per super-step
  static constexpr int PITCH = C * 2 + (C >= 32 ? 32 : 0);  // LDS row pitch in bytes
  static constexpr int NB = C / 16;                    // 16-channel blocks
  static constexpr int NT = NB * (NB + 1) / 2;         // upper-triangular output tiles
  static constexpr int KSTEPS = RPS / 32;
  static constexpr int TILE_BYTES = RPS * PITCH;
};

__device__ __forceinline__ bf16x8_t bg_tr_frag(const unsigned char* lo, int hi_off) {
  const bg_v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bg_v4s*)(lo));
  const bg_v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bg_v4s*)(lo + hi_off));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// ACT: CVH_ACT_NONE / CVH_ACT_SILU compiled in; -1 = the run-time activation code
template <int ACT> __device__ __forceinline__ float act_of(float v, int act) { return act_fwd(v, ACT < 0 ? act : ACT); }

template <int C, int ACT>
__global__ __launch_bounds__(64 * bg_waves<C>(), C >= 64 ? 3 : 4) void bn_apply_gram_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                                          const float* __restrict__ shift, int act,
                                                                          const bf16_t* __restrict__ residual, bf16_t* __restrict__ y,
                                                                          long long rows, float* __restrict__ part) {
  using GM = BgGeom<C>;
  constexpr int BG_WAVES = bg_waves<C>();
  constexpr int CG = GM::CG, PITCH = GM::PITCH, NB = GM::NB, NT = GM::NT;
  constexpr int RED_FLOATS = NT * 256 + C;
  constexpr int SMEM = BG_WAVES * GM::TILE_BYTES > RED_FLOATS * 4 ? BG_WAVES * GM::TILE_BYTES : RED_FLOATS * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  unsigned char* tile = smem + wave * GM::TILE_BYTES;
  const int cg = lane % CG;  // 64 % CG == 0: a lane owns the same 8 channels in every piece it touches
  // the lane's 16 coefficients: in registers, or (C = 64: 40 accumulator registers) re-read from LDS at every piece
  constexpr bool COEF_LDS = C >= 64;
  __shared__ __attribute__((aligned(16))) float coef[COEF_LDS ? 2 * C : 4];
  float sc[8], sh[8];
  if (COEF_LDS) {
    for (int i = tid; i < 2 * C; i += 64 * BG_WAVES) coef[i] = i < C ? scale[i] : shift[i - C];
    __syncthreads();
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j]; }
  }
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;
  bg_f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = bg_f32x4{0.f, 0.f, 0.f, 0.f};

  const long long total = rows * CG;                       // pieces
  const long long nss = (total + BG_NL * 64 - 1) / (BG_NL * 64);  // super-steps
  const long long wstride = (long long)gridDim.x * BG_WAVES;
  const bool has_res = residual != nullptr;
  // LDS addresses of this lane: piece writes (row (it * 64 + lane) / CG, piece cg) and transpose reads (row 4 l4 + (l15 >> 2), column group l15 & 3)
  const int wr_off = (lane / CG) * PITCH + cg * 16;
  const int rd_off = (4 * l4 + (l15 >> 2)) * PITCH + 8 * (l15 & 3);

  // rolling prefetch: piece slot `it` of the NEXT super-step is requested as soon as the current one's registers are consumed, so every
  // load has a whole super-step (stores, LDS traffic, MFMAs of the other slots) to arrive and no second register set is held
  V8<bf16_t> xv[BG_NL], rv[BG_NL];
  auto load = [&](long long ss, int it) __attribute__((always_inline)) {
    const bf16_t* xb = x + ss * (BG_NL * 64 * 8);  // wave-uniform base, 32-bit lane offsets
    const bf16_t* rb = has_res ? residual + ss * (BG_NL * 64 * 8) : nullptr;
    const long long left = total - ss * (BG_NL * 64);
    const int q = it * 64 + lane;
    const int o = (q < left ? q : 0) * 8;
    xv[it] = v8_load<bf16_t>(xb + o);
    if (has_res) rv[it] = v8_load<bf16_t>(rb + o);
  };

  long long ss = (long long)blockIdx.x * BG_WAVES + wave;
  if (ss < nss) {
#pragma unroll
    for (int it = 0; it < BG_NL; ++it) load(ss, it);
  }
  for (; ss < nss; ss += wstride) {
    bf16_t* yb = y + ss * (BG_NL * 64 * 8);
    const long long left = total - ss * (BG_NL * 64);
    const bool more = ss + wstride < nss;
#pragma unroll
    for (int it = 0; it < BG_NL; ++it) {
      float f[8];
      v8_unpack(xv[it], f);
      if (COEF_LDS) {
        *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(coef + cg * 8);
        *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(coef + cg * 8 + 4);
        *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(coef + C + cg * 8);
        *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(coef + C + cg * 8 + 4);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = act_of<ACT>(f[j] * sc[j] + sh[j], act);
      if (has_res) {
        float r[8];
        v8_unpack(rv[it], r);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += r[j];
      }
      if (more) load(ss + wstride, it);
      V8<bf16_t> ov;
      v8_pack(f, ov);
      const int q = it * 64 + lane;
      const bool ok = q < left;
      if (ok) st16(yb + q * 8, ov.d);
      ov = v8_mask(ov, ok);  // rows past the end contribute nothing
      *reinterpret_cast<uint4*>(tile + it * (64 / CG) * PITCH + wr_off) = ov.d;
      v8_unpack(ov, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) cs[j] += f[j];
    }
    wave_lds_sync();
#pragma unroll
    for (int kb = 0; kb < GM::KSTEPS; ++kb) {
      bf16x8_t fr[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) fr[b] = bg_tr_frag(tile + kb * 32 * PITCH + rd_off + b * 32, 16 * PITCH);
      int t = 0;
#pragma unroll
      for (int b0 = 0; b0 < NB; ++b0)
#pragma unroll
        for (int b1 = b0; b1 < NB; ++b1) {
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[b0], fr[b1], acc[t], 0, 0, 0);  // D[i][j] += sum_k y[k][16 b0 + i] y[k][16 b1 + j]
          ++t;
        }
    }
    wave_lds_sync();  // tile consumed before the next super-step overwrites it
  }

  // ---- workgroup partial: waves add their accumulators in a fixed order (bit-reproducible), then one [C*C + C] row leaves ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  for (int i = tid; i < RED_FLOATS; i += 64 * BG_WAVES) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = wave_strided_sum(cs[j], CG);
  lds_ordered_accumulate(wave, BG_WAVES, true, [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(t * 4 + e) * 64 + lane] += acc[t][e];
    if (lane < CG) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[NT * 256 + lane * 8 + j] += cs[j];
    }
  });
  float* prow = part + (size_t)blockIdx.x * (C * C + C);
  for (int i = tid; i < C * C; i += 64 * BG_WAVES) {
    int r = i / C, c = i - r * C;
    if (r > c) { const int q = r; r = c; c = q; }  // lower triangle = mirror of the upper one
    const int b0 = r >> 4, b1 = c >> 4;
    const int t = b0 * NB - b0 * (b0 - 1) / 2 + (b1 - b0);
    int ri = r & 15, ci = c & 15;
    // a diagonal tile holds both halves; accumulator element (row 4 l4 + e, column l15) sits at [(4 t + e) * 64 + 16 l4 + l15]
    prow[i] = red[(t * 4 + (ri & 3)) * 64 + 16 * (ri >> 2) + ci];
  }
  for (int i = tid; i < C; i += 64 * BG_WAVES) prow[C * C + i] = red[NT * 256 + i];
}

template <int C, int ACT> int bg_launch_act(const void* x, const float* scale, const float* shift, int act, const void* residual, void* y,
                                            long long rows, float* part, int R, hipStream_t st) {
  hipLaunchKernelGGL((bn_apply_gram_kernel<C, ACT>), dim3(R), dim3(64 * bg_waves<C>()), 0, st, (const bf16_t*)x, scale, shift, act,
                     (const bf16_t*)residual, (bf16_t*)y, rows, part);
  CVH_CHECK_LAUNCH();
  return 0;
}
template <int C> int bg_launch(const void* x, const float* scale, const float* shift, int act, const void* residual, void* y, long long rows,
                               float* part, int R, hipStream_t st) {
  if (act == CVH_ACT_NONE) return bg_launch_act<C, CVH_ACT_NONE>(x, scale, shift, act, residual, y, rows, part, R, st);
  if (act == CVH_ACT_SILU) return bg_launch_act<C, CVH_ACT_SILU>(x, scale, shift, act, residual, y, rows, part, R, st);
  return bg_launch_act<C, -1>(x, scale, shift, act, residual, y, rows, part, R, st);
}

}  // namespace

/* partial rows (= workgroups) of cvh_bn_apply_gram; 0: this (rows, C) is not covered (use cvh_bn_apply + cvh_gemm_dw + cvh_colsum) */
extern "C" int cvh_bn_apply_gram_rows(long long rows, int C) {
  if (!(C == 16 || C == 32 || C == 64) || rows <= 0) return 0;
  if (cvh_tune_get(21)) return 0;  // CVH_TUNE key 21: 1 = separate apply and Gram passes (A/B runs)
  const long long pieces = rows * (C / 8);
  const long long nss = (pieces + BG_NL * 64 - 1) / (BG_NL * 64);
  const int waves = C == 64 ? bg_waves<64>() : bg_waves<16>();
  long long wgs = (nss + waves - 1) / waves;
  if (wgs > 512) wgs = 512;  // two 8-wave workgroups per CU
  return (int)wgs;
}

extern "C" int cvh_bn_apply_gram(int dtype, const void* x, const float* scale, const float* shift, int act, const void* residual, void* y,
                                 long long rows, int C, float* part, int R, float* gram_s, void* stream) {
  if (dtype != CVH_DT_BF16) return -1;
  if (R <= 0 || R != cvh_bn_apply_gram_rows(rows, C)) return -2;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  switch (C) {
    case 16: rc = bg_launch<16>(x, scale, shift, act, residual, y, rows, part, R, st); break;
    case 32: rc = bg_launch<32>(x, scale, shift, act, residual, y, rows, part, R, st); break;
    case 64: rc = bg_launch<64>(x, scale, shift, act, residual, y, rows, part, R, st); break;
    default: return -2;
  }
  if (rc) return rc;
  return cvh_sum_partials(part, R, C * C + C, C * C + C, gram_s, 1.0f, 0, stream);
}

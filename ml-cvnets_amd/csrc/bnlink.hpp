// BatchNorm links: train-mode BatchNorm folded into the kernels on either side of it.
//
//   producer epilogue   column statistics of the tensor it just wrote (forward: sum, sumsq of the raw conv output; backward:
//                       sum g, sum g*xhat of the gradient it just formed) leave the kernel as per-workgroup partial rows
//                       [rows][2][C]; cvh_bn_finalize / cvh_bn_bwd_finalize turn them into per-channel coefficients (forward: mean,
//                       invstd, scale, shift + running statistics; backward: dgamma, dbeta, ca, cb, cc).  (Measured and rejected:
//                       device-scope atomics into a slot accumulator + "last workgroup finalises" — 0.25-1.4 M atomics per launch
//                       run at ~10 ns each per memory channel and cost 150-500 us.)
//   consumer prologue   applies the coefficients while loading its operand (OperandXf): the normalised / activated tensor and the
//                       BatchNorm input gradient are never written to HBM.
//
// Replaces (reference): nn.BatchNorm2d + nn.SiLU between the convolutions of ConvLayer2d stacks, cvnets/layers/conv_layer.py:254-255,
// cvnets/layers/normalization/batch_norm.py:14-49, cvnets/modules/mobilenetv2.py:231-235, and their autograd backward.
#pragma once
#include "common.hpp"
#include "cvnets_hip.h"

// operand transform applied on load (per-channel coefficient vectors, channel = fastest dimension of the operand)
//   mode 0: v = a
//   mode 1: v = act(c0 * a + c1)                 forward: BatchNorm apply + activation of the raw producer output
//   mode 2: v = c0 * a + c1 * a2 + c2            backward: dy = ca * g + cb * y + cc (BatchNorm input gradient)
struct OperandXf {
  int mode;
  const void* src2;
  const float* c0;
  const float* c1;
  const float* c2;
  int act;
};

__host__ inline OperandXf make_xf(const cvh_operand_xf* x) {
  OperandXf r;
  if (x == nullptr) {
    r.mode = 0; r.src2 = nullptr; r.c0 = r.c1 = r.c2 = nullptr; r.act = 0;
  } else {
    r.mode = x->mode; r.src2 = x->src2; r.c0 = x->c0; r.c1 = x->c1; r.c2 = x->c2; r.act = x->act;
  }
  return r;
}

// ---- 8-wide operand transforms ----------------------------------------------------------------------------------------------
struct Coef8 {
  float a[8], b[8], c[8];
};
__device__ __forceinline__ void coef8_vec(float* o, const float* v, int ch, bool ok) {  // 2 x 16-byte loads (vectors are 32-byte aligned: ch % 8 == 0)
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (ok && v != nullptr) {
    a = *reinterpret_cast<const float4*>(v + ch);
    b = *reinterpret_cast<const float4*>(v + ch + 4);
  }
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void coef8_load(Coef8& k, const OperandXf& x, int ch, bool ok) {
  coef8_vec(k.a, x.c0, ch, ok);
  coef8_vec(k.b, x.c1, ch, ok);
  coef8_vec(k.c, x.c2, ch, ok);
}
// mode 1 / mode 2 on one 8-element vector; `valid` == false yields zeros (padding rows / out-of-image taps must stay zero AFTER the
// transform)
template <typename T>
__device__ __forceinline__ V8<T> xf_apply(const V8<T>& a, const V8<T>& a2, const Coef8& k, int mode, int act, bool valid) {
  if (!valid) return v8_zero<T>();
  float f[8];
  v8_unpack(a, f);
  if (mode == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = k.a[j] * f[j] + k.b[j];
    act_fwd8(f, act);
  } else {
    float g[8];
    v8_unpack(a2, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = k.a[j] * f[j] + k.b[j] * g[j] + k.c[j];
  }
  V8<T> r;
  v8_pack(f, r);
  return r;
}

// Which VALU op classes overlap with MFMA on one SIMD (gfx950)?  Per iteration and wave: NM mfma 16x16x32 bf16 (4 independent accumulators) and NV VALU of ONE class
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
constexpr int NM = 8, NV = 32;
__device__ unsigned long long g_cycles;
// CLS 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_exp_f32, 3: v_cvt_pk_bf16_f32, 4: v_pk_mul_f32, 5: v_add_u32 (int), 6: v_rcp
template <int CLS, bool DOM, bool DOV> __global__ __launch_bounds__(256) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8_t a; for (int i = 0; i < 8; ++i) a[i] = (__bf16)(lane * 0.01f + i);
  f32x4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x2_t p[8]; float t[8]; unsigned u[8];
  for (int i = 0; i < 8; ++i) { p[i] = f32x2_t{lane * 0.001f + i, 0.5f}; t[i] = 0.1f * i + lane * 1e-3f; u[i] = lane + i; }
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (DOM) {
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc[i & 3], 0, 0, 0);
    }
    if (DOV) {
#pragma unroll
      for (int r = 0; r < NV / 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (CLS == 0) t[i] = __builtin_fmaf(t[i], 1.0001f, 0.5f);
          if (CLS == 1) p[i] = p[i] * f32x2_t{1.0001f, 0.9999f} + f32x2_t{0.5f, 0.25f};
          if (CLS == 2) t[i] = __builtin_amdgcn_exp2f(t[i]);
          if (CLS == 3) { unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(p[i], bf16x2_t)); p[i][0] = __uint_as_float(w); }
          if (CLS == 4) p[i] = p[i] * p[(i + 1) & 7];
          if (CLS == 5) u[i] = u[i] * 3u + (u[(i + 1) & 7] ^ 0x55u);
          if (CLS == 6) t[i] = __builtin_amdgcn_rcpf(t[i]);
        }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1] + t[i] + (float)u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) g_cycles = c1 - c0;
}
template <int CLS> __global__ __launch_bounds__(256) void ks(float* out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8_t a; for (int i = 0; i < 8; ++i) a[i] = (__bf16)(lane * 0.01f + i);
  f32x4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x2_t p[8]; float t[8];
  for (int i = 0; i < 8; ++i) { p[i] = f32x2_t{lane * 0.001f + i, 0.5f}; t[i] = 0.1f * i + lane * 1e-3f; }
  // waves of one SIMD: wave w of blocks b, b+1, ... land on SIMD w -> alternate by block parity so that every SIMD hosts both roles
  const bool mf = __builtin_amdgcn_readfirstlane((int)((blockIdx.x >> 8) & 1)) != 0;
  if (mf) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc[i & 3], 0, 0, 0);
  } else {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < NV / 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (CLS == 0) t[i] = __builtin_fmaf(t[i], 1.0001f, 0.5f);
          if (CLS == 1) p[i] = p[i] * f32x2_t{1.0001f, 0.9999f} + f32x2_t{0.5f, 0.25f};
          if (CLS == 2) t[i] = __builtin_amdgcn_exp2f(t[i]);
        }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1] + t[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CLS> void runs(const char* name, float* out) {
  for (int wps : {2, 4}) {
    const int iters = 3000; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    dim3 grid(256 * wps);
    ks<CLS><<<grid, 256>>>(out, 10); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); ks<CLS><<<grid, 256>>>(out, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-16s SPECIALISED waves/SIMD %d (half MFMA-only, half VALU-only waves): %6.1f cycles per iteration of the slower role; serial would be (V+M)*%d/2\n", name, wps, ms * 1e-3 * 2.4e9 / iters, wps);
  }
}
template <int CLS, bool DOM, bool DOV> float run1(float* out, int wps) {
  const int iters = 3000; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  dim3 grid(256 * wps);
  k<CLS, DOM, DOV><<<grid, 256>>>(out, 10); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<CLS, DOM, DOV><<<grid, 256>>>(out, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long cyc = 0; (void)hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));
  printf("      [CLS %d M %d V %d w%d: %.1f us, shader-clock cycles of wave 0: %.1f per iteration -> clock %.2f GHz]\n", CLS, (int)DOM, (int)DOV, wps, ms * 1e3, (double)cyc / iters, (double)cyc / (ms * 1e-3) * 1e-9);
  return ms * 1e-3 * 2.4e9 / ((double)iters * wps);
}
template <int CLS> void run(const char* name, float* out) {
  for (int wps : {2, 4}) {
    const float v = run1<CLS, false, true>(out, wps), m = run1<CLS, true, false>(out, wps), b = run1<CLS, true, true>(out, wps);
    printf("%-16s waves/SIMD %d: VALU alone %6.1f  MFMA alone %6.1f  both %6.1f   (sum %6.1f, max %6.1f)\n", name, wps, v, m, b, v + m, v > m ? v : m);
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 4 * 256 * 4 * sizeof(float));
  printf("cycles (2.4 GHz nominal) per iteration per resident wave; iteration = %d mfma16x16x32 + %d VALU of one class, same wave, independent\n", NM, NV);
  runs<0>("v_fma_f32", out); runs<1>("v_pk_fma_f32", out); runs<2>("v_exp_f32", out);
  run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out); run<2>("v_exp_f32", out); run<6>("v_rcp_f32", out); run<3>("v_cvt_pk_bf16", out); run<4>("v_pk_mul_f32", out); run<5>("int mad/xor", out);
  return 0;
}

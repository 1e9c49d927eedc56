// Do VALU, MFMA and LDS work of different (and of the same) waves overlap on one SIMD / CU?  (gfx950)
// per iteration and wave: NL ds_read_b128, NM mfma 16x16x32 bf16, NV VALU (1/4 transcendental, 3/4 v_pk_fma_f32)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int NL = 7, NM = 7, NV = 32;
// MODE bit0: VALU, bit1: MFMA, bit2: LDS; bit3: dependent chain LDS -> MFMA -> VALU (as in the real kernels); bit4: wave-specialised (even waves VALU, odd waves MFMA + LDS)
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
  __syncthreads();
  bf16x8_t a; for (int i = 0; i < 8; ++i) a[i] = (__bf16)(lane * 0.01f + i);
  f32x4_t acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x2_t p[4]; for (int i = 0; i < 4; ++i) p[i] = f32x2_t{lane * 0.001f + i, 0.5f};
  float t[4] = {0.1f, 0.2f, 0.3f, 0.4f};
  bool doV = MODE & 1, doM = MODE & 2, doL = MODE & 4;
  const unsigned blk = blockIdx.x;
  if (MODE & 16) { const bool even = ((wave + blk) & 1) == 0; doV = even; doM = !even; doL = !even; }
  const unsigned char* base = lds + (lane & 15) * 32 + (lane >> 4) * 16 + wave * 2048;
  for (int it = 0; it < iters; ++it) {
    bf16x8_t b[NL];
    if (doL) {
#pragma unroll
      for (int i = 0; i < NL; ++i) b[i] = *reinterpret_cast<const bf16x8_t*>(base + ((i * 576 + it * 16) & 1023));
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) b[i] = a;
    }
    if (doM) {
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[i % NL], acc[i & 1], 0, 0, 0);
    } else if (doL) {
#pragma unroll
      for (int i = 0; i < NL; ++i) t[i & 3] += (float)b[i][0];
    }
    if (doV) {
      if (MODE & 8) { p[0][0] += acc[0][0]; p[1][0] += acc[1][1]; }
#pragma unroll
      for (int r = 0; r < NV / 8; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = p[i] * f32x2_t{1.0001f, 0.9999f} + f32x2_t{0.5f, 0.25f};
#pragma unroll
        for (int i = 0; i < 2; ++i) t[i] = __builtin_amdgcn_exp2f(t[i] + p[i][0]);
#pragma unroll
        for (int i = 0; i < 2; ++i) p[i + 2][1] += t[i];
      }
      if (MODE & 8) { a[0] = (__bf16)p[0][0]; }
    }
  }
  float s = acc[0][0] + acc[1][1] + acc[0][2] + acc[1][3] + t[0] + t[1] + t[2] + t[3];
  for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* out) {
  printf("%-34s", name);
  for (int wps : {1, 2, 3, 4}) {
    const int iters = 3000; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    dim3 grid(256 * wps);
    k<MODE><<<grid, 256>>>(out, 10); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE><<<grid, 256>>>(out, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  w%d: %7.1f cyc/iter/wave-slot", wps, ms * 1e-3 * 2.4e9 / ((double)iters * wps));
  }
  printf("\n");
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 4 * 256 * 4 * sizeof(float));
  printf("cycles (2.4 GHz nominal) per iteration per resident wave of a SIMD; an iteration = %d ds_read_b128 + %d mfma16x16x32 + %d VALU\n", NL, NM, NV);
  run<1>("VALU only", out); run<2>("MFMA only", out); run<4>("LDS only", out);
  run<3>("VALU + MFMA independent", out); run<5>("VALU + LDS", out); run<6>("LDS -> MFMA", out);
  run<7>("all three, VALU independent", out); run<15>("all three, dependent chain", out);
  run<16 + 7>("wave-specialised (half VALU, half MFMA+LDS)", out);
  return 0;
}

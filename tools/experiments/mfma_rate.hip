// MFMA 16x16x32 bf16 issue rate: independent accumulators vs dependent chains, 1 / 2 / 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <int NACC> __global__ __launch_bounds__(256) void k(float* out, int iters) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i); }
  f32x4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(float* out) {
  for (int wps : {1, 2, 4}) {
    const int iters = 4000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * wps);
    k<NACC><<<grid, 256>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC><<<grid, 256>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mfma16x16x32 chains=%d waves/SIMD %d: %.1f us -> %.2f cycles per MFMA per SIMD (2.4 GHz)\n", NACC, wps, ms * 1e3, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
  }
}
int main() {
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4 * sizeof(float));
  run<1>(out); run<2>(out); run<4>(out); run<8>(out);
  return 0;
}

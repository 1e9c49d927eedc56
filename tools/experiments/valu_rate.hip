// VALU issue-rate microbenchmark (gfx950): cycles per wave64 instruction of the op classes the SiLU epilogues are made of, at 1 / 2 / 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a[8]; f32x2_t p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2_t{a[i], a[i] + 0.5f}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
        if (OP == 1) a[i] = __builtin_amdgcn_rcpf(a[i]);
        if (OP == 2) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
        if (OP == 3) p[i] = p[i] * f32x2_t{1.0001f, 1.0002f} + f32x2_t{0.5f, 0.25f};
        if (OP == 4) { uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(p[i], bf16x2_t)); p[i][0] = __uint_as_float(u); }
        if (OP == 5) p[i] = p[i] + 1.0f;
        if (OP == 6) a[i] = __uint_as_float(__float_as_uint(a[i]) & 0xffff0000u);
        if (OP == 7) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
        if (OP == 8) p[i] = p[i] * p[(i + 1) & 7];
      }
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, float* out) {
  for (int wps : {1, 2, 4}) {
    const int iters = 2000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * wps);  // 256-thread blocks: 4 waves, one per SIMD
    k<OP><<<grid, 256>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<grid, 256>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 64;
    // per SIMD: wps waves x instr_per_wave; assume 2.4 GHz
    printf("%-14s waves/SIMD %d: %.2f us  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wps, ms * 1e3, ms * 1e-3 * 2.4e9 / (instr_per_wave * wps));
  }
}
int main() {
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4 * sizeof(float));
  run<0>("v_exp_f32", out); run<1>("v_rcp_f32", out); run<2>("v_fma_f32", out); run<3>("v_pk_fma_f32", out); run<4>("v_cvt_pk_bf16", out);
  run<5>("v_pk_add_f32", out); run<6>("v_and_b32", out); run<7>("v_exp_f16", out); run<8>("v_pk_mul_f32", out);
  return 0;
}

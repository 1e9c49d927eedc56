// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element lands in which (lane, slot)?
// LDS holds elem[i] = i (u16).  Lane l supplies byte offset addr[l]; we print the 4 returned u16 per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const int* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + (uint32_t)addr[threadIdx.x];
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = r.x & 0xffff;
  out[threadIdx.x * 4 + 1] = r.x >> 16;
  out[threadIdx.x * 4 + 2] = r.y & 0xffff;
  out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
  // experiment 1: canonical — lane l points at element 4*l (contiguous 4 elems per lane)
  // experiment 2: row-strided — lane i in group g: row r = i/4 (stride 100 elems), quad q = i%4, group offset 1000*g
  for (int exp = 0; exp < 2; ++exp) {
    std::vector<int> addr(64);
    for (int l = 0; l < 64; ++l) {
      int g = l >> 4, i = l & 15;
      addr[l] = exp == 0 ? 2 * (4 * l) : 2 * (1000 * g + 100 * (i / 4) + 4 * (i % 4));
    }
    int* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
    hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out);
    std::vector<uint16_t> out(256);
    hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
    printf("experiment %d\n", exp);
    for (int l = 0; l < 64; ++l) printf("lane %2d (addr elem %4d): %4d %4d %4d %4d\n", l, addr[l] / 2, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
  }
  return 0;
}

// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of the CVNets hot path.
// Wave = 64 lanes everywhere.  Storage type T is float (parity mode) or bf16_t (bf16 mode);
// all arithmetic accumulates in fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <type_traits>

#define CVH_DT_F32 0
#define CVH_DT_BF16 1

#define CVH_ACT_NONE 0
#define CVH_ACT_SILU 1
#define CVH_ACT_GELU 2
#define CVH_ACT_RELU 3
#define CVH_ACT_DERIV 17   /* backward: multiply by a stored derivative (include/cvnets_hip.h) */
#define CVH_ACT_GELU_D 18  /* forward: GELU that stores its derivative */

struct bf16_t {
  uint16_t v;
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// f32 -> bf16, round-to-nearest-even: native __bf16 conversions lower to the gfx950 hardware v_cvt_pk_bf16_f32 (one VALU op per
// PAIR instead of ~6 integer ops per value — the GEMM epilogues were VALU-bound on the bit-twiddling version).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> __device__ __forceinline__ float to_f(T x);
template <> __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t x) { return bf2f(x.v); }
template <typename T> __device__ __forceinline__ T from_f(float f);
template <> __device__ __forceinline__ float from_f<float>(float f) { return f; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float f) {
  bf16_t r;
  r.v = f2bf(f);
  return r;
}
// value as it will read back after being stored as T
template <typename T> __device__ __forceinline__ float round_to(float f) { return to_f<T>(from_f<T>(f)); }

// ---------------------------------------------------------------------------------------------
// 8-element and 4-element vectors of T (16 B / 8 B for bf16, 32 B / 16 B for f32)
// ---------------------------------------------------------------------------------------------
template <typename T> struct V8;
template <> struct V8<bf16_t> {
  uint4 d;
};
template <> struct V8<float> {
  float4 a, b;
};
template <typename T> struct V4;
template <> struct V4<bf16_t> {
  uint2 d;
};
template <> struct V4<float> {
  float4 a;
};

template <typename T> __device__ __forceinline__ V8<T> v8_zero();
template <> __device__ __forceinline__ V8<bf16_t> v8_zero<bf16_t>() {
  V8<bf16_t> r;
  r.d = make_uint4(0, 0, 0, 0);
  return r;
}
template <> __device__ __forceinline__ V8<float> v8_zero<float>() {
  V8<float> r;
  r.a = make_float4(0, 0, 0, 0);
  r.b = r.a;
  return r;
}
// CVH_NT_LOADS / CVH_NT_STORES (compile-time, per translation unit: tools/build_variant.py): non-temporal 16-byte global accesses for the
// kernels that stream tensors far larger than the caches exactly once
typedef unsigned int cvh_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16(const void* p) {
#ifdef CVH_NT_LOADS
  const cvh_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const cvh_u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const uint4*>(p);
#endif
}
__device__ __forceinline__ void st16(void* p, uint4 u) {
#ifdef CVH_NT_STORES
  cvh_u32x4 v = {u.x, u.y, u.z, u.w};
  __builtin_nontemporal_store(v, reinterpret_cast<cvh_u32x4*>(p));
#else
  *reinterpret_cast<uint4*>(p) = u;
#endif
}
template <typename T> __device__ __forceinline__ V8<T> v8_load(const T* p);
template <> __device__ __forceinline__ V8<bf16_t> v8_load<bf16_t>(const bf16_t* p) {
  V8<bf16_t> r;
  r.d = ld16(p);
  return r;
}
template <> __device__ __forceinline__ V8<float> v8_load<float>(const float* p) {
  V8<float> r;
  const uint4 a = ld16(p), b = ld16(p + 4);
  r.a = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
  r.b = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
  return r;
}
template <typename T> __device__ __forceinline__ void v8_store(T* p, const V8<T>& v);
template <> __device__ __forceinline__ void v8_store<bf16_t>(bf16_t* p, const V8<bf16_t>& v) { st16(p, v.d); }
template <> __device__ __forceinline__ void v8_store<float>(float* p, const V8<float>& v) {
  st16(p, make_uint4(__float_as_uint(v.a.x), __float_as_uint(v.a.y), __float_as_uint(v.a.z), __float_as_uint(v.a.w)));
  st16(p + 4, make_uint4(__float_as_uint(v.b.x), __float_as_uint(v.b.y), __float_as_uint(v.b.z), __float_as_uint(v.b.w)));
}
// branch-free predicated load, in two halves so that the load stays a PREFETCH: v8_load_clamped reads base[ok ? off : 0 ...] (always
// a valid address) and returns whatever is there; v8_mask zeroes it where the value is CONSUMED.  (Conditional per-element loads of
// several register arrays in one loop make the optimizer sink them behind pointer phis, which pins the arrays in scratch memory;
// masking right at the load would make the wave wait for the data on the spot.)
template <typename T> __device__ __forceinline__ V8<T> v8_load_clamped(const T* base, size_t off, bool ok) {
  return *reinterpret_cast<const V8<T>*>(base + (ok ? off : (size_t)0));
}
__device__ __forceinline__ V8<bf16_t> v8_mask(V8<bf16_t> v, bool ok) {
  const uint32_t m = ok ? 0xffffffffu : 0u;
  v.d.x &= m; v.d.y &= m; v.d.z &= m; v.d.w &= m;
  return v;
}
__device__ __forceinline__ V8<float> v8_mask(V8<float> v, bool ok) {
  return ok ? v : v8_zero<float>();
}
template <typename T> __device__ __forceinline__ V8<T> v8_load_if(const T* base, size_t off, bool ok) {  // load + mask (no prefetch distance needed)
  return v8_mask(v8_load_clamped<T>(base, off, ok), ok);
}
template <typename T> __device__ __forceinline__ V4<T> v4_load(const T* p) { return *reinterpret_cast<const V4<T>*>(p); }
template <typename T> __device__ __forceinline__ void v4_store(T* p, const V4<T>& v) { *reinterpret_cast<V4<T>*>(p) = v; }

__device__ __forceinline__ void v8_unpack(const V8<bf16_t>& v, float* f) {
  const uint32_t w[4] = {v.d.x, v.d.y, v.d.z, v.d.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void v8_unpack(const V8<float>& v, float* f) {
  f[0] = v.a.x; f[1] = v.a.y; f[2] = v.a.z; f[3] = v.a.w;
  f[4] = v.b.x; f[5] = v.b.y; f[6] = v.b.z; f[7] = v.b.w;
}
__device__ __forceinline__ void v8_pack(const float* f, V8<bf16_t>& v) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = f2bf_pk(f[2 * i], f[2 * i + 1]);
  v.d = make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void v8_pack(const float* f, V8<float>& v) {
  v.a = make_float4(f[0], f[1], f[2], f[3]);
  v.b = make_float4(f[4], f[5], f[6], f[7]);
}
__device__ __forceinline__ void v4_unpack(const V4<bf16_t>& v, float* f) {
  f[0] = __uint_as_float(v.d.x << 16); f[1] = __uint_as_float(v.d.x & 0xffff0000u);
  f[2] = __uint_as_float(v.d.y << 16); f[3] = __uint_as_float(v.d.y & 0xffff0000u);
}
__device__ __forceinline__ void v4_unpack(const V4<float>& v, float* f) {
  f[0] = v.a.x; f[1] = v.a.y; f[2] = v.a.z; f[3] = v.a.w;
}
__device__ __forceinline__ void v4_pack(const float* f, V4<bf16_t>& v) {
  v.d.x = f2bf_pk(f[0], f[1]);
  v.d.y = f2bf_pk(f[2], f[3]);
}
__device__ __forceinline__ void v4_pack(const float* f, V4<float>& v) { v.a = make_float4(f[0], f[1], f[2], f[3]); }

// ---------------------------------------------------------------------------------------------
// activations (fp32 math)
// ---------------------------------------------------------------------------------------------
// sigmoid through the hardware reciprocal (v_rcp_f32, 1 ulp): the plain `1.0f / (...)` expands into the ~10-instruction IEEE
// division sequence (v_div_scale / v_div_fmas / v_div_fixup), which made every SiLU-bearing kernel VALU-bound on this chip
// (HBM : VALU is ~14 fp32 ops per byte).  exp(-x) overflows to +inf for x < -88 -> rcp(inf) = 0, the correct limit.
// e^x as ONE v_exp_f32 (2^(x * log2 e)).  `__expf` compiles to the range-reduced expansion (v_rndne / v_ldexp / compare + exec-mask branches,
// ~20 instructions and a branch per call): in the SiLU-heavy depthwise and BatchNorm-link kernels that was a third of all VALU work.
// Relative error <= 2^-23 * (1 + |x| log2 e): 1e-6 for the |x| < 10 of activations and softmax exponents.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float act_fwd(float x, int act) {
  if (act == CVH_ACT_SILU) return x * sigmoidf_(x);
  if (act == CVH_ACT_GELU) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));  // exact erf GELU
  if (act == CVH_ACT_RELU) return fmaxf(x, 0.0f);
  return x;
}
__device__ __forceinline__ float act_grad(float x, int act) {  // d act(x) / dx
  if (act == CVH_ACT_SILU) {
    float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
  }
  if (act == CVH_ACT_GELU) {
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    float pdf = 0.3989422804014327f * fast_exp(-0.5f * x * x);
    return cdf + x * pdf;
  }
  if (act == CVH_ACT_RELU) return x > 0.0f ? 1.0f : 0.0f;
  if (act == CVH_ACT_DERIV) return x;  // x IS the stored derivative
  return 1.0f;
}
// 8-wide forms: ONE (wave-uniform) dispatch on `act` per vector instead of one per element — the per-element scalar branches
// chop the unrolled element loops into basic blocks and leave the VALU without independent work to interleave
__device__ __forceinline__ void act_fwd8(float* v, int act) {
  if (act == CVH_ACT_SILU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * sigmoidf_(v[j]);
  } else if (act == CVH_ACT_GELU || act == CVH_ACT_GELU_D) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.5f * v[j] * (1.0f + erff(v[j] * 0.70710678118654752f));
  } else if (act == CVH_ACT_RELU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.0f);
  }
}
// v[j] = GELU(x[j]) and d[j] = GELU'(x[j]) from one erf (CVH_ACT_GELU_D: the derivative is stored for the backward GEMM's epilogue)
__device__ __forceinline__ void gelu_fwd_deriv8(float* v, float* d) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j];
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * fast_exp(-0.5f * x * x);
    v[j] = x * cdf;
    d[j] = cdf + x * pdf;
  }
}
// g[j] *= act'(x[j])
__device__ __forceinline__ void act_grad8_mul(float* g, const float* x, int act) {
  if (act == CVH_ACT_SILU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = sigmoidf_(x[j]);
      g[j] *= s * (1.0f + x[j] * (1.0f - s));
    }
  } else if (act == CVH_ACT_GELU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float cdf = 0.5f * (1.0f + erff(x[j] * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * fast_exp(-0.5f * x[j] * x[j]);
      g[j] *= cdf + x[j] * pdf;
    }
  } else if (act == CVH_ACT_RELU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = x[j] > 0.0f ? g[j] : 0.0f;
  } else if (act == CVH_ACT_DERIV) {  // x already holds act'(pre-activation) (stored by a CVH_ACT_GELU_D forward)
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] *= x[j];
  }
}

// ---------------------------------------------------------------------------------------------
// counter-based RNG for dropout: keep-mask is a pure function of (seed, stream id, element index)
// so backward regenerates the identical mask without storing it.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// One 32-bit hash word serves TWO consecutive elements (16 uniform bits each, keep <=> u16 >= round(p * 2^16)): the keep decision
// costs one mix32 per element pair instead of three per element (the integer multiplies are quarter rate — in the token GEMM epilogues
// the generator used to be the largest single VALU item).  The drop probability is p rounded to 2^-16 (0.1 -> 0.100006).
struct DropKey { uint32_t key, thr; };
__device__ __forceinline__ DropKey drop_key(uint64_t seed, uint32_t stream, float p) {
  DropKey k;
  k.key = mix32(stream * 0x9e3779b9u + (uint32_t)seed) ^ (uint32_t)(seed >> 32);
  k.thr = (uint32_t)(p * 65536.0f + 0.5f);
  return k;
}
// pair = element index >> 1; hi = (uint32_t)(pair >> 32) * 0x85ebca6bu (zero below 2^33 elements)
__device__ __forceinline__ uint32_t drop_word(const DropKey& k, uint32_t pair_lo, uint32_t hi) { return mix32(pair_lo ^ k.key ^ hi); }
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint32_t stream, uint64_t idx, float p, float inv_keep) {
  const DropKey k = drop_key(seed, stream, p);
  const uint64_t pair = idx >> 1;
  const uint32_t h = drop_word(k, (uint32_t)pair, (uint32_t)(pair >> 32) * 0x85ebca6bu);
  const uint32_t u = (idx & 1) ? (h >> 16) : (h & 0xffffu);
  return u >= k.thr ? inv_keep : 0.0f;
}
// v[0..7] *= keep-scale of elements idx8 .. idx8 + 7 (idx8 % 8 == 0): the same masks as eight dropout_scale calls
__device__ __forceinline__ void dropout_scale8(const DropKey& k, uint64_t idx8, float inv_keep, float* v) {
  const uint64_t pair = idx8 >> 1;
  const uint32_t lo = (uint32_t)pair, hi = (uint32_t)(pair >> 32) * 0x85ebca6bu;  // pair .. pair + 3 share the high word (idx8 % 8 == 0)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = drop_word(k, lo + q, hi);
    v[2 * q] *= (h & 0xffffu) >= k.thr ? inv_keep : 0.0f;
    v[2 * q + 1] *= (h >> 16) >= k.thr ? inv_keep : 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------
// LDS transpose reads next to direct-to-LDS loads.  With __builtin_amdgcn_ds_read_tr16_b64_* the compiler cannot tell which LDS bytes the
// read touches, so while ANY global_load_lds of the wave is outstanding it puts `s_waitcnt vmcnt(0)` in front of every such read: a kernel
// that requests the next reduction step's tiles and then reads the current ones waits for the NEXT step to land before its first MFMA (found
// in round 6 in every dW kernel: gemm_tn128 / gemm_tn256 / gemm_tn_rows ran load -> wait -> compute with no overlap at all).  These forms
// issue the same instruction from inline asm: the compiler places no wait, the kernel orders the data itself - tr_wait() before the first
// use of the fragments (it names them, so the MFMAs cannot be scheduled above it), the kernel's own vmcnt + barrier before a buffer is read.
// ---------------------------------------------------------------------------------------------
#ifndef CVH_TR_ASM
#define CVH_TR_ASM 1  // 0 (tools/build_variant.py): the builtin form, for A/B runs
#endif
typedef short cvh_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr32(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
template <int OFF> __device__ __forceinline__ cvh_v4s ds_read_tr16_b64_raw(unsigned addr) {
#if CVH_TR_ASM
  cvh_v4s r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
#else
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cvh_v4s*)(size_t)(addr + OFF));
#endif
}
// one MFMA operand (8 reduction elements per lane) from two transpose reads HI bytes apart
template <int HI> __device__ __forceinline__ bf16x8_t tr_frag_raw(unsigned addr) {
  const cvh_v4s lo = ds_read_tr16_b64_raw<0>(addr), hi = ds_read_tr16_b64_raw<HI>(addr);
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
template <int HI> __device__ __forceinline__ bf16x8_t tr_frag_raw2(unsigned addr, unsigned hi_off) {  // run-time distance between the two reads
  const cvh_v4s lo = ds_read_tr16_b64_raw<0>(addr), hi = ds_read_tr16_b64_raw<HI>(addr + hi_off);
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// 16 plain bytes of an LDS image that direct-to-LDS loads are filling elsewhere (same reason, same contract: tr_wait1 before use)
typedef unsigned cvh_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cvh_u32x4 lds_read_b128_raw(unsigned addr) {
#if CVH_TR_ASM
  cvh_u32x4 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
  return r;
#else
  return *(__attribute__((address_space(3))) const cvh_u32x4*)(size_t)addr;
#endif
}
__device__ __forceinline__ void tr_wait1(cvh_u32x4& a) {
#if CVH_TR_ASM
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a));
#endif
}
__device__ __forceinline__ void tr_wait1(bf16x8_t& a) {
#if CVH_TR_ASM
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a));
#endif
}
// every LDS read of the wave has returned; the named fragments are "produced" here
__device__ __forceinline__ void tr_wait(bf16x8_t& a, bf16x8_t& b, bf16x8_t& c, bf16x8_t& d) {
#if CVH_TR_ASM
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#endif
}
__device__ __forceinline__ void tr_wait(bf16x8_t& a, bf16x8_t& b, bf16x8_t& c, bf16x8_t& d, bf16x8_t& e, bf16x8_t& f) {
#if CVH_TR_ASM
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
#endif
}

// ---------------------------------------------------------------------------------------------
// wave-level reductions (64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------
// MFMA 32x32 tiles.  Operand fragment = 8 consecutive K elements per lane:
//   A: lane l holds A[row = l & 31][k = 8*(l >> 5) + 0..7]    (per 16-wide K step)
//   B: lane l holds B[col = l & 31][k = 8*(l >> 5) + 0..7]    (B stored [N][K], K contiguous)
//   C/D: acc[r] of lane l is D[row = (r & 3) + 8*(r >> 2) + 4*(l >> 5)][col = l & 31]
// bf16: one v_mfma_f32_32x32x16_bf16; f32: eight v_mfma_f32_32x32x2_f32 (exact f32 fma chain) that
// consume the same 8-per-lane fragment (k-pairs {j, 8+j}); A and B use the same k assignment so the
// contraction is identical.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  bf16x8_t v;
};
template <> struct Frag<float> {
  float v[8];
};

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// read the fragment for rows [row0, row0+32) and K offset k0 from an LDS tile with `pitch` elements per row
__device__ __forceinline__ Frag<bf16_t> lds_frag(const bf16_t* tile, int pitch, int row0, int k0, int lane) {
  const bf16_t* p = tile + (row0 + (lane & 31)) * pitch + k0 + 8 * (lane >> 5);
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8_t*>(p);
  return f;
}
__device__ __forceinline__ Frag<float> lds_frag(const float* tile, int pitch, int row0, int k0, int lane) {
  const float* p = tile + (row0 + (lane & 31)) * pitch + k0 + 8 * (lane >> 5);
  Frag<float> f;
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
  return f;
}
// same fragment from a tile whose rows are only 8-byte aligned (pitch % 4 == 0 elements): two ds_read_b64
__device__ __forceinline__ Frag<bf16_t> lds_frag_a8(const bf16_t* tile, int pitch, int row0, int k0, int lane) {
  const bf16_t* p = tile + (row0 + (lane & 31)) * pitch + k0 + 8 * (lane >> 5);
  const uint2 a = *reinterpret_cast<const uint2*>(p);
  const uint2 b = *reinterpret_cast<const uint2*>(p + 4);
  Frag<bf16_t> f;
  f.v = __builtin_bit_cast(bf16x8_t, make_uint4(a.x, a.y, b.x, b.y));
  return f;
}
__device__ __forceinline__ Frag<float> lds_frag_a8(const float* tile, int pitch, int row0, int k0, int lane) {
  return lds_frag(tile, pitch, row0, k0, lane);
}
// ordering of LDS traffic between the lanes of ONE wave (per-wave private LDS regions): LDS operations of a wave execute in
// program order, so only the compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Workgroup barrier for kernels that keep global loads / stores in flight ACROSS it.  __syncthreads() is a fence + s_barrier, and for the fence
// the compiler puts `s_waitcnt vmcnt(0)` in front of the barrier: every wave then drains ALL its outstanding global loads (a prefetch meant to
// fly for several more stages) and — vmcnt counts them too on this target — all its result STORES at every barrier; found with
// gemm_tn_rows_kernel, whose 3, 5 or 8 stages "in flight" all took the same time.  What such a barrier really has to order is LDS traffic between
// the waves: a wave's LDS writes and reads issued before it are complete (lgkmcnt(0)); data that arrives in LDS by global_load_lds needs the
// issuing wave's own (counted) s_waitcnt vmcnt in front of the call.  The "memory" clobber keeps the compiler from moving memory accesses across.
// -DCVH_FENCED_BARRIERS restores __syncthreads() everywhere (A/B builds).
__device__ __forceinline__ void wg_barrier_lds() {
#ifdef CVH_FENCED_BARRIERS
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
// XCD-aware block remap (MI355X: block b runs on XCD b % 8, each XCD has its own L2): give every XCD one CONTIGUOUS chunk of the
// logical work list so that neighbouring work items (which share halo rows / operand panels) hit the same L2.  Bijective for any n.
__device__ __forceinline__ int xcd_chunk_id(int b, int n) {
  const int q = n / 8, r = n % 8, x = b % 8, i = b / 8;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
__device__ __forceinline__ void mma32(f32x16_t& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16_t& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t acc_zero() {
  f32x16_t z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}

// LDS row pitch (elements) for a tile with `k` K-elements per row: +16 B pad makes the 16 rows of a
// ds_read_b128 lane group land on distinct 4-bank slots (pitch/4 dwords odd multiple of 4).
template <typename T> __host__ __device__ constexpr int lds_pitch(int k) { return k + 16 / (int)sizeof(T); }
// Row padding (bf16 elements) of LDS tiles whose ds_read_b128 fragments follow the 16x16x32 MFMA operand layout — lane (l15, l4) reads the
// 16 bytes at row l15, chunk l4 of a 64-byte K step.  The hardware services a b128 read in the lane groups {0-3, 12-15, 20-27}, {4-11,
// 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): a group mixes rows 0-3 / 12-15 at chunk 0 with rows 4-11 at chunk 1, so the row pitch
// in 16-byte units must be = 2 (mod 4) for its 16 lanes to land on 16 distinct bank quads.  The "+ 16 bytes" padding of rounds 2-4 made
// the pitch ODD (right for the 32-row fragments of conv_gemm, 2-way conflicts here: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.42 - 0.48
// on every kernel of this layout, profiles/r05y_pmc_lds.txt).  Tiles whose unpadded row is a multiple of 64 bytes take + 32 bytes.
#ifndef CVH_M16_PAD
#define CVH_M16_PAD 16
#endif

// Deterministic replacement for "every thread atomicAdd()s its partial sums into a small LDS array" at the end of a kernel: the threads
// that share a destination take turns in a fixed order (turn 0 .. nturns-1, one workgroup barrier per turn; threads of one turn must hit
// distinct addresses).  The order in which LDS float atomics retire differs from run to run; with the turns the kernel's statistics /
// partial rows — and with them the whole training step — are bit-reproducible.  Call from workgroup-uniform control flow.
template <typename F> __device__ __forceinline__ void lds_ordered_accumulate(int my_turn, int nturns, bool active, F&& add) {
  for (int t = 0; t < nturns; ++t) {
    if (active && my_turn == t) add();
    __syncthreads();
  }
}
// fixed-shape butterfly over the lanes l, l + stride, l + 2*stride, ... of a wave (stride a power of two): afterwards every lane holds the sum
__device__ __forceinline__ float wave_strided_sum(float v, int stride) {
  for (int m = stride; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// compile-time loop: body receives std::integral_constant<int, I> (keeps register arrays statically indexed)
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Row map of a token "sequence" on an NHWC feature map (MobileViTBlock.unfolding, cvnets/modules/mobilevit_block.py:186-231, as index
// arithmetic): row(s, n) = (b*H + nh*ph + i)*W + nw*pw + j with s = b*ph*pw + i*pw + j, n = nh*n_w + nw.  ph = pw = 1, H = 1,
// W = n_w = S is the plain contiguous [B][S][d] case.
struct SeqMap {
  int ph, pw, n_w, H, W;
};
__device__ __forceinline__ int seq_row(const SeqMap& m, int s, int n) {
  const int P = m.ph * m.pw;
  const int b = s / P, pi = s - b * P;
  const int i = pi / m.pw, j = pi - i * m.pw;
  const int nh = n / m.n_w, nw = n - nh * m.n_w;
  return (b * m.H + nh * m.ph + i) * m.W + nw * m.pw + j;
}

// run-time tuning knobs (A/B experiments from tools/kernel_bench.py; defaults = shipped configuration)
#define CVH_TUNE_TN_PITCH 1      /* 0: 80-byte rows + ds_read_b128, 1: 72-byte rows + 2 x ds_read_b64 */
#define CVH_TUNE_TN_WGS 2        /* target number of gemm_tn workgroups */
#define CVH_TUNE_GEMM_GRID 3     /* conv_gemm grid.x cap */
#define CVH_TUNE_DW_XCD 4        /* depthwise: XCD-contiguous block mapping on/off */
#define CVH_TUNE_BIG_GEMM 5    /* 1: transformer-sized linears use the 128x128 direct-to-LDS kernel (gemm_big.hip), 0: conv_gemm */
#define CVH_TUNE_COLRED_ROWS 6 /* cap on the number of partial rows (= workgroups) of the column-reduction kernels */
#define CVH_TUNE_LN_PER_ROW 12 /* 1: LayerNorm on the one-row-per-wave kernels instead of the grouped ones */
#define CVH_TUNE_NO_SKINNY 9   /* 1: pointwise dW of small tiles stays on gemm_tn_kernel */
#define CVH_TUNE_SKINNY_WGS 10 /* workgroups of gemm_tn_skinny_kernel (0: 512) */
#define CVH_TUNE_NO_WAVE_PRIVATE 11 /* 1: single-K-step BatchNorm-link GEMMs keep the cooperative (barrier) staging */
#define CVH_TUNE_NO_STREAM_GEMM 13 /* 1: short-K pointwise GEMMs stay on conv_gemm / gemm_nt128 instead of gemm_stream_kernel */
#define CVH_TUNE_NO_CONV3X3 14 /* 1: dense 3x3 convolutions stay on the im2col conv_gemm_kernel instead of conv3x3_kernel */
#define CVH_TUNE_NO_CONV3X3_DW 15 /* 1: the weight gradient of those convolutions stays on the im2col gemm_tn_kernel instead of conv3x3_dw_kernel */
#define CVH_TUNE_BIG_MIN_N 16 /* narrowest output the ragged direct-to-LDS GEMM takes (0: the default, 192) */
#define CVH_TUNE_NO_TN256 19 /* 1: the direct-to-LDS dW product stays on 128 x 128 tiles (gemm_big.hip: gemm_tn256_shape) */
#define CVH_TUNE_GEMM_FILL 18 /* conv_gemm: narrow the N tile until the launch has at least this many workgroups (0: off) */
#define CVH_TUNE_MAX 32
int cvh_tune_get(int key);

// host-side tally of a kernel family's launches and algorithmic bytes (cvh_family_counters; defined in gemm.hip)
void cvh_family_tally(int family, long long bytes);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: cache of the size already granted, one slot per device
// (a process that touches a second GPU must set it there too).  One static instance per kernel instantiation.
struct DynSmemAttr {
  std::atomic<size_t> granted[16];
  DynSmemAttr() { for (auto& g : granted) g.store(0); }
  hipError_t ensure(const void* fn, size_t smem) {
    if (smem <= 64 * 1024) return hipSuccess;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<size_t>& g = granted[dev & 15];
    if (smem <= g.load(std::memory_order_acquire)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) g.store(smem, std::memory_order_release);
    return e;
  }
};


#define CVH_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e_ = hipGetLastError();             \
    if (e_ != hipSuccess) return (int)e_;          \
  } while (0)

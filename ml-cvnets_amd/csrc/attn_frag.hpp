// Shared pieces of the fused attention kernels (csrc/attention.hip: tile-streaming kernels with every mask / dropout feature;
// csrc/attn_res.hip: operand-resident kernels for short unmasked sequences): the parameter block and the MFMA operand fragments.
#pragma once
#include "common.hpp"

struct AttnParams {
  const void* qkv;   // T [rows][3d]
  void* out;         // fwd: T [rows][d]
  const void* dout;  // bwd: T [rows][d]
  void* dqkv;        // bwd: T [rows][3d]
  float* lse;        // [nseq][h][S]
  float* dsum;       // [nseq][h][S]   D = rowsum(dO * O)
  const unsigned char* kpm;  // optional key padding mask [nseq][S] (nonzero = masked)
  const float* bias;         // optional additive mask (multi_head_attention.py:197-208): [S][S] (bias_stride 0) or [nseq][S][S], natural-log units, -inf allowed
  long long bias_stride;     // elements between the masks of consecutive sequences
  int nseq, S, h, c, d;
  SeqMap map;
  float scaling;
  int causal;
  // dropout on the attention probabilities (MultiHeadAttention.attn_dropout, cvnets/layers/multi_head_attention.py:217): the keep mask is
  // a function of (seed, stream, sequence, head, query, key) and is REGENERATED in the backward kernels, never stored
  float drop_p;
  const unsigned long long* seed;
  unsigned int stream_id;
};

// operand fragment read "down the rows" of a row-major [k][n] LDS tile (the transposed operand):
// element j = tile[k0 + 8*(lane>>5) + j][col0 + (lane&31)]
// bf16: two gfx950 LDS transpose reads (ds_read_b64_tr_b16).  Inside a 16-lane group lane i = 4r + q supplies the address of 4
// contiguous elements of block row r (columns 4q..4q+3) and receives COLUMN i of that 4 x 16 block, i.e. 4 consecutive k of one n —
// exactly half an MFMA operand.  Needs 8-byte aligned addresses: pitch % 4 == 0, col0 % 4 == 0 (measured on MI355X with
// tools/experiments/tr_probe.hip).  Replaces 8 ds_read_u16 + 4 pack VALU ops per fragment.
typedef short tr_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Frag<bf16_t> lds_frag_strided(const bf16_t* tile, int pitch, int k0, int col0, int lane) {
  const int i = lane & 15;
  const bf16_t* p = tile + (k0 + 8 * (lane >> 5) + (i >> 2)) * pitch + col0 + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
  const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(p));
  const tr_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(p + 4 * pitch));
  typedef short v8s __attribute__((ext_vector_type(8)));
  const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  Frag<bf16_t> f;
  f.v = __builtin_bit_cast(bf16x8_t, both);
  return f;
}
__device__ __forceinline__ Frag<float> lds_frag_strided(const float* tile, int pitch, int k0, int col0, int lane) {
  const float* p = tile + (k0 + 8 * (lane >> 5)) * pitch + col0 + (lane & 31);
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = p[j * pitch];
  return f;
}

// The same transposed-operand fragment with the contraction index PERMUTED into the order in which a 32x32 accumulator holds its rows:
// slot j of lane half h <-> tile row k0 + 8*(j>>2) + 4*h + (j&3).  A 32x32 accumulator whose ROWS are the next product's contraction
// index (P^T, dS^T, P, dS) is then fed to the next MFMA straight from registers (frag_from_acc) — the probabilities never touch LDS.
__device__ __forceinline__ Frag<bf16_t> lds_frag_strided_perm(const bf16_t* tile, int pitch, int k0, int col0, int lane) {
  const int i = lane & 15;
  const bf16_t* p = tile + (k0 + 4 * (lane >> 5) + (i >> 2)) * pitch + col0 + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
  const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(p));
  const tr_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(p + 8 * pitch));
  typedef short v8s __attribute__((ext_vector_type(8)));
  const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  Frag<bf16_t> f;
  f.v = __builtin_bit_cast(bf16x8_t, both);
  return f;
}
__device__ __forceinline__ Frag<float> lds_frag_strided_perm(const float* tile, int pitch, int k0, int col0, int lane) {
  const float* p = tile + (k0 + 4 * (lane >> 5)) * pitch + col0 + (lane & 31);
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = p[((j & 3) + 8 * (j >> 2)) * pitch];
  return f;
}
// registers 8t .. 8t+7 of a 32x32 accumulator as the operand fragment of k-step t (rows 16t .. 16t+15 in the permuted order above)
template <typename T> __device__ __forceinline__ Frag<T> frag_from_acc(const f32x16_t& acc, int t);
template <> __device__ __forceinline__ Frag<bf16_t> frag_from_acc<bf16_t>(const f32x16_t& acc, int t) {
  const uint4 u = make_uint4(f2bf_pk(acc[8 * t + 0], acc[8 * t + 1]), f2bf_pk(acc[8 * t + 2], acc[8 * t + 3]),
                             f2bf_pk(acc[8 * t + 4], acc[8 * t + 5]), f2bf_pk(acc[8 * t + 6], acc[8 * t + 7]));
  Frag<bf16_t> f;
  f.v = __builtin_bit_cast(bf16x8_t, u);
  return f;
}
template <> __device__ __forceinline__ Frag<float> frag_from_acc<float>(const f32x16_t& acc, int t) {
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = acc[8 * t + j];
  return f;
}
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;


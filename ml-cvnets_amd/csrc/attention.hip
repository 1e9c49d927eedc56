// Fused multi-head self-attention (flash style) on MFMA, forward + backward, for the token matrices of
// MobileViT / ViT / CLIP.  The S x S score matrix never reaches HBM (the reference materialises it in
// fp32 and keeps it for backward: cvnets/layers/multi_head_attention.py:187-233).
//
// Layout: qkv is the [rows][3*d] output of the qkv projection (d = heads * c; columns [q | k | v], each
// split head-major).  A "sequence" is a set of rows given by SeqMap:
//     row(s, n) = (b*H + nh*ph + i)*W + nw*pw + j,   s = b*ph*pw + i*pw + j,  n = nh*n_w + nw
// which is exactly MobileViTBlock.unfolding (cvnets/modules/mobilevit_block.py:186-231) applied to an
// NHWC feature map — so unfold/fold cost nothing.  ph = pw = 1, H = 1, W = n_w = S gives the plain
// contiguous [B][S][d] case (ViT / CLIP / standalone MultiHeadAttention).
//
// One 64-lane wave = one (sequence, head, 32-row block); a workgroup is NW (1/2/4) waves working on consecutive blocks of
// the SAME (sequence, head), which share the staged K/V (or Q/dO) tiles; per-wave tiles live in that wave's LDS slice.
// Every kernel forms the TRANSPOSED scores S^T = K Q^T, so the MFMA accumulator holds keys along rows and
// the query along lane&31: the softmax statistics (max / sum / lse / D) are per-lane scalars.
#include "common.hpp"
#include "cvnets_hip.h"

#ifndef ATTN_WPE_F
#define ATTN_WPE_F 2
#endif
#ifndef ATTN_WPE_Q
#define ATTN_WPE_Q 2
#endif
#ifndef ATTN_WPE_K
#define ATTN_WPE_K 2
#endif
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

// keep-scale (0 or 1/(1-p)) of one attention probability
__device__ __forceinline__ float attn_keep(const AttnParams& p, unsigned long long seed, float inv_keep, int s, int head, int q, int key) {
  const unsigned long long idx = (((unsigned long long)s * p.h + head) * p.S + q) * (unsigned long long)p.S + key;
  return dropout_scale(seed, p.stream_id, idx, p.drop_p, inv_keep);
}


template <typename T, int VEC> __device__ __forceinline__ void ld_vec(const T* p, float* f) {
  if (VEC == 4) {
    v4_unpack(v4_load<T>(p), f);
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) f[e] = to_f<T>(p[e]);
  }
}
template <typename T, int VEC> __device__ __forceinline__ void st_vec(T* p, const float* f) {
  if (VEC == 4) {
    V4<T> v;
    v4_pack(f, v);
    v4_store<T>(p, v);
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) p[e] = from_f<T>(f[e]);
  }
}

// stage `nrows` rows x CP columns (first c valid, rest zero) of a [rows][ld] global matrix into LDS.
// rowidx[r] < 0 marks an absent row (zero filled).
template <typename T, int CP, int VEC>
__device__ __forceinline__ void stage_rows(T* lds, int pitch, const T* g, int ld, int col0, const int* rowidx, int nrows, int c, float scale,
                                           int tid, int nthreads) {
  constexpr int CH = CP / VEC;
  for (int idx = tid; idx < nrows * CH; idx += nthreads) {
    const int r = idx / CH, cc = (idx - r * CH) * VEC;
    float f[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) f[e] = 0.f;
    const int ri = rowidx[r];
    if (ri >= 0 && cc < c) {
      ld_vec<T, VEC>(g + (size_t)ri * ld + col0 + cc, f);
      if (scale != 1.0f) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] *= scale;
      }
    }
    st_vec<T, VEC>(lds + r * pitch + cc, f);
  }
}

// Register-prefetched staging of an [NROWS x CP] tile (rows picked by an index list, first c columns valid, the rest zero) by NT
// threads.  `load` only ISSUES the global reads (into registers) so it can be placed a whole tile ahead of its use and overlap
// with the MFMA / softmax work of the current tile; `store` writes the registers (optionally scaled) into LDS.  The plain
// stage_rows above is load -> wait -> store per element group, which serialises one HBM round trip per loop iteration.
template <typename T, int CP, int VEC, int NROWS, int NT>
struct TileStager {
  static constexpr int CH = CP / VEC;
  static constexpr int IT = (NROWS * CH + NT - 1) / NT;
  V4<T> q[VEC == 4 ? IT : 1];
  T e[VEC == 4 ? 1 : IT][VEC == 4 ? 1 : VEC];
  unsigned okm;  // VEC == 4: validity of q[it] (bit it), applied in store()

  __device__ __forceinline__ void load(const T* __restrict__ g, int ld, int col0, const int* rowidx, int nrows, int c, int tid) {
    if constexpr (VEC == 4) {
      // two branch-free phases: all row indices (LDS) first, then all global loads from clamped addresses; validity is applied when the
      // registers are written to LDS.  (One `if (ok) { index read; load }` per element serialised an LDS round trip in front of every load
      // and put the loads behind exec-mask branches — s_waitcnt vmcnt(0) everywhere.)
      int ri[IT];
      okm = 0;
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * NT;
        const int r = idx / CH, cc = (idx - r * CH) * VEC;
        const bool in = idx < NROWS * CH && r < nrows && cc < c;
        ri[it] = rowidx[in ? r : 0];
        if (in && ri[it] >= 0) okm |= 1u << it;
      }
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * NT;
        const int r = idx / CH, cc = (idx - r * CH) * VEC;
        const bool ok = (okm >> it) & 1u;
        q[it] = v4_load<T>(g + (ok ? (size_t)ri[it] * ld + col0 + cc : (size_t)0));
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NT;
      const int r = idx / CH, cc = (idx - r * CH) * VEC;
      int ri = -1;
      if (idx < NROWS * CH && r < nrows && cc < c) ri = rowidx[r];
      if constexpr (VEC == 4) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        v4_pack(z, q[it]);
        if (ri >= 0) q[it] = v4_load<T>(g + (size_t)ri * ld + col0 + cc);
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) e[it][k] = (ri >= 0) ? g[(size_t)ri * ld + col0 + cc + k] : from_f<T>(0.f);
      }
    }
  }
  // sum over this thread's VEC elements of iteration `it` of the product with another stager loaded over the same rows / columns
  __device__ __forceinline__ float dot(const TileStager& o, int it) const {
    float s = 0.f;
    if constexpr (VEC == 4) {
      float a[4], b[4];
      v4_unpack(q[it], a);
      v4_unpack(o.q[it], b);
#pragma unroll
      for (int k = 0; k < 4; ++k) s += a[k] * b[k];
      return ((okm >> it) & 1u) ? s : 0.f;
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) s += to_f<T>(e[it][k]) * to_f<T>(o.e[it][k]);
      return s;
    }
  }
  __device__ __forceinline__ void store(T* lds, int pitch, float scale, int tid) const {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NT;
      if (idx >= NROWS * CH) continue;
      const int r = idx / CH, cc = (idx - r * CH) * VEC;
      if constexpr (VEC == 4) {
        V4<T> v = q[it];
        if (!((okm >> it) & 1u)) {
          float z[4] = {0.f, 0.f, 0.f, 0.f};
          v4_pack(z, v);
        }
        if (scale != 1.0f) {
          float f[4];
          v4_unpack(v, f);
#pragma unroll
          for (int k = 0; k < 4; ++k) f[k] *= scale;
          v4_pack(f, v);
        }
        v4_store<T>(lds + r * pitch + cc, v);
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) lds[r * pitch + cc + k] = (scale != 1.0f) ? from_f<T>(to_f<T>(e[it][k]) * scale) : e[it][k];
      }
    }
  }
};

// Two [NROWS x CP] tiles that share their row list (K and V of one key tile; Q and dO of one query tile), prefetched into registers by
// NT threads a whole tile ahead of their use.  The row list is PADDED to a multiple of the tile height with -1 (absent row), so the hot
// loop has no bounds tests and no branches: all row indices are read from LDS first, then all global loads are issued from clamped
// addresses; validity is applied when the registers are written to LDS.  One row index read and (for equal leading dimensions) one
// 64-bit address product serve both tiles.  (The per-tile TileStager pair above cost as many instructions per tile as the softmax.)
template <typename T, int CP, int VEC, int NROWS, int NT>
struct PairStager {
  static constexpr int CH = CP / VEC;
  static constexpr int IT = (NROWS * CH + NT - 1) / NT;
  static constexpr bool EXACT = (NROWS * CH) % NT == 0;
  V4<T> qa[VEC == 4 ? IT : 1], qb[VEC == 4 ? IT : 1];
  T ea[VEC == 4 ? 1 : IT], eb[VEC == 4 ? 1 : IT];
  unsigned okm;

  // ga / gb already point at column 0 of the wanted column block; rows = padded row list of this tile
  __device__ __forceinline__ void load(const T* __restrict__ ga, int lda, const T* __restrict__ gb, int ldb, const int* rows, int c, int tid) {
    int ri[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NT;
      ri[it] = rows[(EXACT || idx < NROWS * CH) ? idx / CH : 0];
    }
    okm = 0;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NT;
      const int cc = (idx % CH) * VEC;
      const bool ok = (EXACT || idx < NROWS * CH) && cc < c && ri[it] >= 0;
      okm |= (ok ? 1u : 0u) << it;
      const size_t oa = ok ? (size_t)ri[it] * lda + cc : (size_t)0;
      const size_t ob = ok ? (size_t)ri[it] * ldb + cc : (size_t)0;
      if constexpr (VEC == 4) {
        qa[it] = v4_load<T>(ga + oa);
        qb[it] = v4_load<T>(gb + ob);
      } else {
        ea[it] = ga[oa];
        eb[it] = gb[ob];
      }
    }
  }
  __device__ __forceinline__ void store(T* la, float scale_a, T* lb, int pitch, int tid) const {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NT;
      if (!EXACT && idx >= NROWS * CH) continue;
      const int r = idx / CH, cc = (idx % CH) * VEC;
      const bool ok = (okm >> it) & 1u;
      if constexpr (VEC == 4) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        V4<T> zero, va = qa[it], vb = qb[it];
        v4_pack(z, zero);
        if (scale_a != 1.0f) {
          float f[4];
          v4_unpack(va, f);
#pragma unroll
          for (int k = 0; k < 4; ++k) f[k] *= scale_a;
          v4_pack(f, va);
        }
        v4_store<T>(la + r * pitch + cc, ok ? va : zero);
        v4_store<T>(lb + r * pitch + cc, ok ? vb : zero);
      } else {
        la[r * pitch + cc] = ok ? from_f<T>(to_f<T>(ea[it]) * scale_a) : from_f<T>(0.f);
        lb[r * pitch + cc] = ok ? eb[it] : from_f<T>(0.f);
      }
    }
  }
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

// Key visibility without per-score branches: every kernel builds, once, a table (LDS) or a per-lane flag of the keys that no query may
// see (beyond the sequence end, or set in the key padding mask); the causal rule is a compare.  Masked scores become -inf, so exp2 -> 0.
template <int FEAT> __device__ __forceinline__ int key_dead_flag(const AttnParams& p, int s, int key) {
  if (key >= p.S) return 1;
  if (!FEAT) return 0;
  return (p.kpm && p.kpm[(size_t)s * p.S + key]) ? 1 : 0;
}

// =============================================================================================
// shared-memory carving (dynamic LDS; every offset is a multiple of 16 B)
// =============================================================================================
template <typename T> __device__ __forceinline__ T* carve(char*& p, int elems) {
  T* r = reinterpret_cast<T*>(p);
  p += ((size_t)elems * sizeof(T) + 15) / 16 * 16;
  return r;
}
static size_t carve_bytes(size_t elems, size_t esz) { return (elems * esz + 15) / 16 * 16; }

// =============================================================================================
// forward
// =============================================================================================
// CPK = head width rounded up to the 16-wide MFMA k-step (the contraction length of Q K^T); the LDS tiles and the output fragments are
// CP = CPK rounded up to 32 columns wide (columns >= c are zero).
template <typename T, int CPK, int VEC, int NW, int FEAT>
__global__ __launch_bounds__(64 * NW, FEAT ? 2 : ATTN_WPE_F) void attn_fwd_kernel(AttnParams p) {
  constexpr int CP = (CPK + 31) / 32 * 32;
  constexpr int KB = (NW == 1) ? 32 : 64;  // NW == 1 <=> S <= 32: one 32-key tile covers the sequence
  constexpr int PQ = lds_pitch<T>(CP);
  constexpr int NFC = CP / 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const bool causal = FEAT && p.causal;  // FEAT == 0: no causal rule, no key padding mask, no dropout (compiled out)
  char* sp = smem_raw;
  T* Ks = carve<T>(sp, KB * PQ);
  T* Vs = carve<T>(sp, KB * PQ);
  T* Qs_all = carve<T>(sp, NW * 32 * PQ);
  int* rk = carve<int>(sp, (p.S + 63) & ~63);  // row index of every key of this sequence, -1 padded to the tile height
  int* rq_all = carve<int>(sp, NW * 32);
  int* kd = carve<int>(sp, (p.S + 63) & ~63);  // 1 = key can never be seen

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T* Qs = Qs_all + wave * 32 * PQ;
  int* rq = rq_all + wave * 32;
  const int nqb = (p.S + 31) / 32;
  const int nqg = (nqb + NW - 1) / NW;
  // XCD-contiguous work order: the query groups and heads of one sequence (and the 4 pixel-interleaved sequences of one image) run on
  // the SAME XCD, so their shared K / V / Q rows are fetched into one L2 once instead of once per XCD (block b runs on XCD b % 8)
  const int bid = xcd_chunk_id(blockIdx.x, gridDim.x);
  const int qg = bid % nqg;
  const int head = (bid / nqg) % p.h;
  const int s = bid / (nqg * p.h);
  const int q0 = (qg * NW + wave) * 32;
  const int q0_last = (qg * NW + NW - 1) * 32;
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const int ld = 3 * p.d;

  for (int i = tid; i < ((p.S + 63) & ~63); i += 64 * NW) {
    rk[i] = i < p.S ? seq_row(p.map, s, i) : -1;
    kd[i] = key_dead_flag<FEAT>(p, s, i);
  }
  if (lane < 32) rq[lane] = (q0 + lane < p.S) ? seq_row(p.map, s, q0 + lane) : -1;
  __syncthreads();
  PairStager<T, CP, VEC, KB, 64 * NW> kvst;
  const T* kcol = qkv + p.d + head * p.c;
  auto load_kv = [&](int kv0) { kvst.load(kcol, ld, kcol + p.d, ld, rk + kv0, p.c, tid); };
  {
    TileStager<T, CP, VEC, 32, 64> qst;
    qst.load(qkv, ld, head * p.c, rq, 32, p.c, lane);
    load_kv(0);  // first K/V tile requested together with Q
    qst.store(Qs, PQ, p.scaling * kLog2e, lane);  // scores come out in log2 units: exp(s - m) is one v_exp_f32, no multiply
  }

  const int my_q = q0 + (lane & 31);
  const float* bias_row = (FEAT && p.bias != nullptr && my_q < p.S) ? p.bias + (size_t)s * p.bias_stride + (size_t)my_q * p.S : nullptr;
  const bool drop = FEAT && p.drop_p > 0.f;
  const unsigned long long seed = drop ? *p.seed : 0ull;
  const float inv_keep = drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  float m_run = -1e30f, l_run = 0.f;  // running max (log2 units) and normaliser
  f32x16_t oacc[NFC];
#pragma unroll
  for (int f = 0; f < NFC; ++f) oacc[f] = acc_zero();

  for (int kv0 = 0; kv0 < p.S; kv0 += KB) {
    if (causal && kv0 > q0_last + 31) break;  // tile entirely in the future of every query of the workgroup
    __syncthreads();                             // previous tile's K/V (and this wave's P) fully consumed
    kvst.store(Ks, 1.0f, Vs, PQ, tid);           // rows past the sequence end arrive as zeros (V rows feed the MFMA k-dimension)
    if (kv0 + KB < p.S) load_kv(kv0 + KB);       // next tile's HBM reads fly under this tile's MFMA + softmax
    __syncthreads();
    const bool full_tile = !causal && (!FEAT || (p.kpm == nullptr && p.bias == nullptr)) && kv0 + KB <= p.S;  // workgroup-uniform: no per-element visibility tests

    // S^T[key][q] = sum_c K[key][c] * Qs[q][c]
    f32x16_t sacc[KB / 32];
#pragma unroll
    for (int f = 0; f < KB / 32; ++f) sacc[f] = acc_zero();
#pragma unroll
    for (int kk = 0; kk < CPK; kk += 16) {
      Frag<T> bq = lds_frag(Qs, PQ, 0, kk, lane);
#pragma unroll
      for (int f = 0; f < KB / 32; ++f) {
        Frag<T> ak = lds_frag(Ks, PQ, f * 32, kk, lane);
        mma32(sacc[f], ak, bq);
      }
    }
    if (!full_tile) {  // one workgroup-uniform branch per tile, selects inside
#pragma unroll
      for (int f = 0; f < KB / 32; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int key0 = kv0 + f * 32 + 8 * i + 4 * (lane >> 5);
          const int4 dd = *reinterpret_cast<const int4*>(kd + key0);
          const int dead[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool hide = dead[e] != 0 || (causal && key0 + e > my_q);
            float sv = sacc[f][4 * i + e];
            if (FEAT && bias_row != nullptr && key0 + e < p.S) sv += bias_row[key0 + e] * kLog2e;  // scores are in log2 units
            sacc[f][4 * i + e] = hide ? -INFINITY : sv;
          }
        }
    }
    float mx = -1e30f;
#pragma unroll
    for (int f = 0; f < KB / 32; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[f][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2_fast(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int f = 0; f < KB / 32; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = exp2_fast(sacc[f][r] - m_new);
        rs += pv;  // the softmax normaliser runs over the undropped probabilities
        sacc[f][r] = pv;
      }
    if (drop) {
#pragma unroll
      for (int f = 0; f < KB / 32; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[f][r] *= attn_keep(p, seed, inv_keep, s, head, my_q, kv0 + f * 32 + acc_row(r, lane));
    }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int f = 0; f < NFC; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[f][r] *= alpha;
    // O^T[c][q] += sum_key V[key][c] * P^T[key][q]: the accumulator rows of P^T (keys) ARE this product's contraction index, so P^T goes
    // from the accumulator registers straight into the B operand and V is read in the matching key order
#pragma unroll
    for (int f = 0; f < KB / 32; ++f)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const Frag<T> bp = frag_from_acc<T>(sacc[f], t);
#pragma unroll
        for (int fc = 0; fc < NFC; ++fc) {
          Frag<T> av = lds_frag_strided_perm(Vs, PQ, f * 32 + 16 * t, fc * 32, lane);
          mma32(oacc[fc], av, bp);
        }
      }
  }

  if (my_q < p.S) {
    const float inv_l = 1.0f / l_run;
    T* out = reinterpret_cast<T*>(p.out);
    const size_t row = (size_t)rq[lane & 31];
#pragma unroll
    for (int f = 0; f < NFC; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cb = f * 32 + 8 * i + 4 * (lane >> 5);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = oacc[f][4 * i + e] * inv_l;
        if (VEC == 4) {
          if (cb < p.c) st_vec<T, 4>(out + row * p.d + head * p.c + cb, o);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (cb + e < p.c) out[row * p.d + head * p.c + cb + e] = from_f<T>(o[e]);
        }
      }
    if (lane < 32 && p.lse) p.lse[((size_t)s * p.h + head) * p.S + my_q] = (m_run + __log2f(l_run)) * kLn2;
  }
}

// =============================================================================================
// backward dQ: wave = (sequence, head, 32-query block); the NW waves of a workgroup share the K/V tiles
// =============================================================================================
template <typename T, int CPK, int VEC, int NW, int FEAT>
__global__ __launch_bounds__(64 * NW, FEAT ? 2 : ATTN_WPE_Q) void attn_bwd_dq_kernel(AttnParams p) {
  constexpr int CP = (CPK + 31) / 32 * 32;
  constexpr int KB = (NW == 1) ? 32 : 64;  // NW == 1 <=> S <= 32: one 32-key tile covers the sequence
  constexpr int PQ = lds_pitch<T>(CP);
  constexpr int NFC = CP / 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const bool causal = FEAT && p.causal;  // FEAT == 0: no causal rule, no key padding mask, no dropout (compiled out)
  char* sp = smem_raw;
  T* Ks = carve<T>(sp, KB * PQ);
  T* Vs = carve<T>(sp, KB * PQ);
  T* Qs_all = carve<T>(sp, NW * 32 * PQ);
  T* dOs_all = carve<T>(sp, NW * 32 * PQ);
  int* rk = carve<int>(sp, (p.S + 63) & ~63);
  int* rq_all = carve<int>(sp, NW * 32);
  float* dw_all = carve<float>(sp, NW * 32);
  int* kd = carve<int>(sp, (p.S + 63) & ~63);  // 1 = key can never be seen

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T* Qs = Qs_all + wave * 32 * PQ;
  T* dOs = dOs_all + wave * 32 * PQ;
  int* rq = rq_all + wave * 32;
  const int nqb = (p.S + 31) / 32;
  const int nqg = (nqb + NW - 1) / NW;
  // XCD-contiguous work order: the query groups and heads of one sequence (and the 4 pixel-interleaved sequences of one image) run on
  // the SAME XCD, so their shared K / V / Q rows are fetched into one L2 once instead of once per XCD (block b runs on XCD b % 8)
  const int bid = xcd_chunk_id(blockIdx.x, gridDim.x);
  const int qg = bid % nqg;
  const int head = (bid / nqg) % p.h;
  const int s = bid / (nqg * p.h);
  const int q0 = (qg * NW + wave) * 32;
  const int q0_last = (qg * NW + NW - 1) * 32;
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const T* dout = reinterpret_cast<const T*>(p.dout);
  const int ld = 3 * p.d;

  for (int i = tid; i < ((p.S + 63) & ~63); i += 64 * NW) {
    rk[i] = i < p.S ? seq_row(p.map, s, i) : -1;
    kd[i] = key_dead_flag<FEAT>(p, s, i);
  }
  if (lane < 32) rq[lane] = (q0 + lane < p.S) ? seq_row(p.map, s, q0 + lane) : -1;
  __syncthreads();
  PairStager<T, CP, VEC, KB, 64 * NW> kvst;
  const T* kcol = qkv + p.d + head * p.c;
  auto load_kv = [&](int kv0) { kvst.load(kcol, ld, kcol + p.d, ld, rk + kv0, p.c, tid); };
  const int my_q = q0 + (lane & 31);
  const float* bias_row = (FEAT && p.bias != nullptr && my_q < p.S) ? p.bias + (size_t)s * p.bias_stride + (size_t)my_q * p.S : nullptr;
  const bool q_ok = my_q < p.S;
  const size_t sidx = ((size_t)s * p.h + head) * p.S + (q_ok ? my_q : 0);
  float dsum;
  {
    // D[q] = sum_c dO[q][c] * O[q][c] of this wave's 32 queries, formed here from the dO tile on its way into LDS and the matching O
    // rows (no separate pass over dO and O), and published for the dK/dV kernel that runs next
    typedef TileStager<T, CP, VEC, 32, 64> QS;
    QS qst, dst, ost;
    qst.load(qkv, ld, head * p.c, rq, 32, p.c, lane);
    dst.load(dout, p.d, head * p.c, rq, 32, p.c, lane);
    ost.load(reinterpret_cast<const T*>(p.out), p.d, head * p.c, rq, 32, p.c, lane);
    load_kv(0);
    qst.store(Qs, PQ, p.scaling * kLog2e, lane);  // scores in log2 units
    dst.store(dOs, PQ, 1.0f, lane);
    float* dw = dw_all + wave * 32;
    constexpr int CH = QS::CH, RPI = 64 / CH;  // a row's chunks sit in CH consecutive lanes
    static_assert(64 % CH == 0 && QS::IT * RPI == 32, "stager layout");
#pragma unroll
    for (int it = 0; it < QS::IT; ++it) {
      float v = dst.dot(ost, it);
#pragma unroll
      for (int m = CH / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane % CH == 0) dw[it * RPI + lane / CH] = v;
    }
    wave_lds_sync();
    dsum = dw[lane & 31];
    if (lane < 32 && q_ok) p.dsum[sidx] = dsum;
  }
  const float lse = p.lse[sidx] * kLog2e;
  const bool drop = FEAT && p.drop_p > 0.f;
  const unsigned long long seed = drop ? *p.seed : 0ull;
  const float inv_keep = drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;

  f32x16_t dqacc[NFC];
#pragma unroll
  for (int f = 0; f < NFC; ++f) dqacc[f] = acc_zero();

  for (int kv0 = 0; kv0 < p.S; kv0 += KB) {
    if (causal && kv0 > q0_last + 31) break;
    __syncthreads();
    kvst.store(Ks, 1.0f, Vs, PQ, tid);
    if (kv0 + KB < p.S) load_kv(kv0 + KB);
    __syncthreads();
    const bool full_tile = !causal && (!FEAT || (p.kpm == nullptr && p.bias == nullptr)) && kv0 + KB <= p.S;

    f32x16_t sacc[KB / 32], dpacc[KB / 32];
#pragma unroll
    for (int f = 0; f < KB / 32; ++f) { sacc[f] = acc_zero(); dpacc[f] = acc_zero(); }
#pragma unroll
    for (int kk = 0; kk < CPK; kk += 16) {
      Frag<T> bq = lds_frag(Qs, PQ, 0, kk, lane);
      Frag<T> bd = lds_frag(dOs, PQ, 0, kk, lane);
#pragma unroll
      for (int f = 0; f < KB / 32; ++f) {
        Frag<T> ak = lds_frag(Ks, PQ, f * 32, kk, lane);
        Frag<T> av = lds_frag(Vs, PQ, f * 32, kk, lane);
        mma32(sacc[f], ak, bq);   // S^T  = K Q^T
        mma32(dpacc[f], av, bd);  // dP^T = V dO^T
      }
    }
    if (drop) {  // dP = keep * (dO V^T)
#pragma unroll
      for (int f = 0; f < KB / 32; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) dpacc[f][r] *= attn_keep(p, seed, inv_keep, s, head, my_q, kv0 + f * 32 + acc_row(r, lane));
    }
    if (!full_tile) {  // (rows of absent queries are computed but never stored)
#pragma unroll
      for (int f = 0; f < KB / 32; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int key0 = kv0 + f * 32 + 8 * i + 4 * (lane >> 5);
          const int4 dd = *reinterpret_cast<const int4*>(kd + key0);
          const int dead[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool hide = !q_ok || dead[e] != 0 || (causal && key0 + e > my_q);
            float sv = sacc[f][4 * i + e];
            if (FEAT && bias_row != nullptr && key0 + e < p.S) sv += bias_row[key0 + e] * kLog2e;
            sacc[f][4 * i + e] = hide ? -INFINITY : sv;  // exp2 -> 0
          }
        }
    }
#pragma unroll
    for (int f = 0; f < KB / 32; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) dpacc[f][r] = exp2_fast(sacc[f][r] - lse) * (dpacc[f][r] - dsum);  // dS^T
    // dQ^T[c][q] += sum_key K[key][c] * dS^T[key][q]: dS^T from the accumulator registers, K read in the matching key order
#pragma unroll
    for (int f = 0; f < KB / 32; ++f)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const Frag<T> bs = frag_from_acc<T>(dpacc[f], t);
#pragma unroll
        for (int fc = 0; fc < NFC; ++fc) {
          Frag<T> ak = lds_frag_strided_perm(Ks, PQ, f * 32 + 16 * t, fc * 32, lane);
          mma32(dqacc[fc], ak, bs);
        }
      }
  }

  if (q_ok) {
    T* dqkv = reinterpret_cast<T*>(p.dqkv);
    const size_t row = (size_t)rq[lane & 31];
#pragma unroll
    for (int f = 0; f < NFC; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cb = f * 32 + 8 * i + 4 * (lane >> 5);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = dqacc[f][4 * i + e] * p.scaling;
        if (VEC == 4) {
          if (cb < p.c) st_vec<T, 4>(dqkv + row * ld + head * p.c + cb, o);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (cb + e < p.c) dqkv[row * ld + head * p.c + cb + e] = from_f<T>(o[e]);
        }
      }
  }
}

// =============================================================================================
// backward dK, dV: wave = (sequence, head, 32-key block); the NW waves of a workgroup share the Q/dO tiles.
// Here the scores are formed UNtransposed, S = Q K^T (queries along the accumulator rows, the wave's keys along lane&31), so that the
// accumulator rows of P and dS are the contraction index of dV^T = dO^T P and dK^T = Q^T dS: both go from registers into the B operand
// (see lds_frag_strided_perm) and the results come out as [channel rows][key lanes] — 4 consecutive channels of one key per register
// group, stored with 8-byte writes.  The softmax statistics are per accumulator ROW here and are read from a small LDS table.
// =============================================================================================
template <typename T, int CPK, int VEC, int NW, int FEAT>
__global__ __launch_bounds__(64 * NW, FEAT ? 2 : ATTN_WPE_K) void attn_bwd_dkv_kernel(AttnParams p) {
  constexpr int CP = (CPK + 31) / 32 * 32;
  constexpr int QB = 32;
  constexpr int PQ = lds_pitch<T>(CP);
  constexpr int NFC = CP / 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const bool causal = FEAT && p.causal;  // FEAT == 0: no causal rule, no key padding mask, no dropout (compiled out)
  char* sp = smem_raw;
  T* Qs = carve<T>(sp, QB * PQ);
  T* dOs = carve<T>(sp, QB * PQ);
  T* Ks_all = carve<T>(sp, NW * 32 * PQ);
  T* Vs_all = carve<T>(sp, NW * 32 * PQ);
  float* stat = carve<float>(sp, 2 * QB);  // [lse * log2e | D] of the current query tile
  int* rq = carve<int>(sp, (p.S + 31) & ~31);  // row index of every query of this sequence, -1 padded to the tile height
  int* rk_all = carve<int>(sp, NW * 32);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  T* Ks = Ks_all + wave * 32 * PQ;
  T* Vs = Vs_all + wave * 32 * PQ;
  int* rk = rk_all + wave * 32;
  const int nkb = (p.S + 31) / 32;
  const int nkg = (nkb + NW - 1) / NW;
  const int bid = xcd_chunk_id(blockIdx.x, gridDim.x);  // XCD-contiguous work order (see attn_fwd_kernel)
  const int kg = bid % nkg;
  const int head = (bid / nkg) % p.h;
  const int s = bid / (nkg * p.h);
  const int k0 = (kg * NW + wave) * 32;
  const int k0_first = kg * NW * 32;
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const T* dout = reinterpret_cast<const T*>(p.dout);
  const int ld = 3 * p.d;

  for (int i = tid; i < ((p.S + 31) & ~31); i += 64 * NW) rq[i] = i < p.S ? seq_row(p.map, s, i) : -1;
  if (lane < 32) rk[lane] = (k0 + lane < p.S) ? seq_row(p.map, s, k0 + lane) : -1;
  __syncthreads();
  PairStager<T, CP, VEC, QB, 64 * NW> qdst;
  float lse_n = 0.f, dsum_n = 0.f;  // softmax statistics of the prefetched query tile (threads 0..31)
  auto load_q = [&](int qb0) {
    qdst.load(qkv + head * p.c, ld, dout + head * p.c, p.d, rq + qb0, p.c, tid);
    if (tid < QB) {
      const int q = qb0 + tid;
      const size_t si = ((size_t)s * p.h + head) * p.S + (q < p.S ? q : 0);
      lse_n = p.lse[si];
      dsum_n = p.dsum[si];
    }
  };
  auto q_block_skipped = [&](int qb0) { return causal && qb0 + QB - 1 < k0_first; };  // all its queries precede every key of the workgroup
  int qb_next = 0;
  while (qb_next < p.S && q_block_skipped(qb_next)) qb_next += QB;
  {
    TileStager<T, CP, VEC, 32, 64> kst, vst;
    kst.load(qkv, ld, p.d + head * p.c, rk, 32, p.c, lane);
    vst.load(qkv, ld, 2 * p.d + head * p.c, rk, 32, p.c, lane);
    if (qb_next < p.S) load_q(qb_next);
    kst.store(Ks, PQ, 1.0f, lane);
    vst.store(Vs, PQ, 1.0f, lane);
  }

  f32x16_t dkacc[NFC], dvacc[NFC];
#pragma unroll
  for (int f = 0; f < NFC; ++f) { dkacc[f] = acc_zero(); dvacc[f] = acc_zero(); }
  const bool drop = FEAT && p.drop_p > 0.f;
  const unsigned long long seed = drop ? *p.seed : 0ull;
  const float inv_keep = drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  const int my_key = k0 + (lane & 31);
  const bool key_dead = key_dead_flag<FEAT>(p, s, my_key) != 0;
  const float* bias_col = (FEAT && p.bias != nullptr && my_key < p.S) ? p.bias + (size_t)s * p.bias_stride + my_key : nullptr;

  while (qb_next < p.S) {
    const int qb0 = qb_next;
    __syncthreads();
    qdst.store(Qs, p.scaling * kLog2e, dOs, PQ, tid);  // scores in log2 units; dK is scaled back by ln 2 at the end
    if (tid < QB) {
      stat[tid] = lse_n * kLog2e;
      stat[QB + tid] = dsum_n;
    }
    qb_next = qb0 + QB;
    if (qb_next < p.S) load_q(qb_next);  // next query tile (+ its statistics) in flight during this tile's math
    __syncthreads();

    const bool full_tile = !causal && (!FEAT || (p.kpm == nullptr && p.bias == nullptr)) && qb0 + QB <= p.S && k0 + 32 <= p.S;

    f32x16_t sacc = acc_zero(), dpacc = acc_zero();
#pragma unroll
    for (int kk = 0; kk < CPK; kk += 16) {
      Frag<T> aq = lds_frag(Qs, PQ, 0, kk, lane);
      Frag<T> ad = lds_frag(dOs, PQ, 0, kk, lane);
      Frag<T> bk = lds_frag(Ks, PQ, 0, kk, lane);
      Frag<T> bv = lds_frag(Vs, PQ, 0, kk, lane);
      mma32(sacc, aq, bk);   // S  = Q K^T
      mma32(dpacc, ad, bv);  // dP = dO V^T
    }
    if (!full_tile) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qb0 + acc_row(r, lane);
        const bool hide = key_dead || q >= p.S || (causal && my_key > q);
        float sv = sacc[r];
        if (FEAT && bias_col != nullptr && q < p.S) sv += bias_col[(size_t)q * p.S] * kLog2e;
        sacc[r] = hide ? -INFINITY : sv;  // exp2 -> 0
      }
    }
    if (drop) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float keep = attn_keep(p, seed, inv_keep, s, head, qb0 + acc_row(r, lane), my_key);
        const int i = r >> 2, e = r & 3;
        const float pv = exp2_fast(sacc[r] - stat[8 * i + 4 * half + e]);
        sacc[r] = pv * keep;                                                   // dV = (keep * P)^T dO
        dpacc[r] = pv * (dpacc[r] * keep - stat[QB + 8 * i + 4 * half + e]);   // dS = P * (keep * dP - D)
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 l4 = *reinterpret_cast<const float4*>(stat + 8 * i + 4 * half);
        const float4 d4 = *reinterpret_cast<const float4*>(stat + QB + 8 * i + 4 * half);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pv = exp2_fast(sacc[4 * i + e] - lv[e]);
          sacc[4 * i + e] = pv;
          dpacc[4 * i + e] = pv * (dpacc[4 * i + e] - dv[e]);
        }
      }
    }
    // dV^T[c][key] += sum_q dO[q][c] P[q][key] ;  dK^T[c][key] += sum_q Qs[q][c] dS[q][key]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const Frag<T> bp = frag_from_acc<T>(sacc, t);
      const Frag<T> bs = frag_from_acc<T>(dpacc, t);
#pragma unroll
      for (int f = 0; f < NFC; ++f) {
        Frag<T> ado = lds_frag_strided_perm(dOs, PQ, 16 * t, f * 32, lane);
        Frag<T> aqq = lds_frag_strided_perm(Qs, PQ, 16 * t, f * 32, lane);
        mma32(dvacc[f], ado, bp);
        mma32(dkacc[f], aqq, bs);
      }
    }
  }

  if (my_key < p.S) {
    T* dqkv = reinterpret_cast<T*>(p.dqkv);
    const size_t row = (size_t)rk[lane & 31];
#pragma unroll
    for (int f = 0; f < NFC; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cb = f * 32 + 8 * i + 4 * half;
        float dk[4], dv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dk[e] = dkacc[f][4 * i + e] * kLn2;
          dv[e] = dvacc[f][4 * i + e];
        }
        T* pk = dqkv + row * ld + p.d + head * p.c + cb;
        T* pv = dqkv + row * ld + 2 * p.d + head * p.c + cb;
        if (VEC == 4) {
          if (cb < p.c) {
            st_vec<T, 4>(pk, dk);
            st_vec<T, 4>(pv, dv);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (cb + e < p.c) {
              pk[e] = from_f<T>(dk[e]);
              pv[e] = from_f<T>(dv[e]);
            }
        }
      }
  }
}

// =============================================================================================
// C ABI
// =============================================================================================
static AttnParams make_params(const void* qkv, void* out, const void* dout, void* dqkv, float* lse, float* dsum, const unsigned char* kpm,
                              int nseq, int S, int h, int c, int ph, int pw, int n_w, int H, int W, float scaling, int causal) {
  AttnParams p;
  p.qkv = qkv; p.out = out; p.dout = dout; p.dqkv = dqkv; p.lse = lse; p.dsum = dsum; p.kpm = kpm;
  p.nseq = nseq; p.S = S; p.h = h; p.c = c; p.d = h * c;
  p.map.ph = ph; p.map.pw = pw; p.map.n_w = n_w; p.map.H = H; p.map.W = W;
  p.scaling = scaling; p.causal = causal;
  p.drop_p = 0.f; p.seed = nullptr; p.stream_id = 0;
  p.bias = nullptr; p.bias_stride = 0;
  return p;
}

enum { K_FWD = 0, K_DQ = 1, K_DKV = 2 };

template <typename T, int CPK, int NW> static size_t attn_smem(int which, int S) {
  constexpr int CP = (CPK + 31) / 32 * 32;
  const size_t PQ = lds_pitch<T>(CP), e = sizeof(T);
  const size_t KB = NW == 1 ? 32 : 64;
  size_t b = carve_bytes(NW * 32, 4);
  b += which != K_DKV ? 2 * carve_bytes((S + 63) & ~63, 4) : carve_bytes((S + 31) & ~31, 4);
  if (which == K_FWD) b += 2 * carve_bytes(KB * PQ, e) + carve_bytes(NW * 32 * PQ, e);
  else if (which == K_DQ) b += 2 * carve_bytes(KB * PQ, e) + 2 * carve_bytes(NW * 32 * PQ, e) + carve_bytes(NW * 32, 4);
  else b += 2 * carve_bytes(32 * PQ, e) + 2 * carve_bytes(NW * 32 * PQ, e) + carve_bytes(64, 4);
  return b;
}

template <typename T, int CPK, int VEC, int NW, int FEAT>
static int launch_attn_feat(int which, const AttnParams& p, hipStream_t st) {
  const size_t smem = attn_smem<T, CPK, NW>(which, p.S);
  if (smem > 160 * 1024) return -2;
  const int nb = (p.S + 31) / 32;
  const int grid = p.nseq * p.h * ((nb + NW - 1) / NW);
  const void* fn = which == K_FWD ? reinterpret_cast<const void*>(attn_fwd_kernel<T, CPK, VEC, NW, FEAT>)
                   : which == K_DQ ? reinterpret_cast<const void*>(attn_bwd_dq_kernel<T, CPK, VEC, NW, FEAT>)
                                   : reinterpret_cast<const void*>(attn_bwd_dkv_kernel<T, CPK, VEC, NW, FEAT>);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  if (which == K_FWD) hipLaunchKernelGGL((attn_fwd_kernel<T, CPK, VEC, NW, FEAT>), dim3(grid), dim3(64 * NW), smem, st, p);
  else if (which == K_DQ) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, CPK, VEC, NW, FEAT>), dim3(grid), dim3(64 * NW), smem, st, p);
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, CPK, VEC, NW, FEAT>), dim3(grid), dim3(64 * NW), smem, st, p);
  CVH_CHECK_LAUNCH();
  return 0;
}

template <typename T, int CPK, int VEC, int NW>
static int launch_attn(int which, const AttnParams& p, hipStream_t st) {
  const bool feat = p.causal || p.kpm != nullptr || p.drop_p > 0.f || p.bias != nullptr;
  return feat ? launch_attn_feat<T, CPK, VEC, NW, 1>(which, p, st) : launch_attn_feat<T, CPK, VEC, NW, 0>(which, p, st);
}
template <typename T, int CPK, int VEC>
static int dispatch_nw(int which, const AttnParams& p, hipStream_t st) {
  const int nb = (p.S + 31) / 32;
  if (nb >= 4) return launch_attn<T, CPK, VEC, 4>(which, p, st);
  if (nb >= 2) return launch_attn<T, CPK, VEC, 2>(which, p, st);
  return launch_attn<T, CPK, VEC, 1>(which, p, st);
}
template <typename T, int CPK>
static int dispatch_vec(int which, const AttnParams& p, hipStream_t st) {
  if (p.c % 4 == 0) return dispatch_nw<T, CPK, 4>(which, p, st);
  return dispatch_nw<T, CPK, 1>(which, p, st);
}
template <typename T>
static int dispatch_cpk(int which, const AttnParams& p, hipStream_t st) {
  if (p.c <= 32) return dispatch_vec<T, 32>(which, p, st);
  if (p.c <= 48) return dispatch_vec<T, 48>(which, p, st);
  return dispatch_vec<T, 64>(which, p, st);
}
static int dispatch_attn(int dtype, int which, const AttnParams& p, hipStream_t st) {
  if (dtype == CVH_DT_BF16) return dispatch_cpk<bf16_t>(which, p, st);
  if (dtype == CVH_DT_F32) return dispatch_cpk<float>(which, p, st);
  return -1;
}

extern "C" int cvh_attn_fwd_drop(int dtype, const void* qkv, void* out, float* lse, const unsigned char* kpm, int nseq, int S, int h, int c,
                                 int ph, int pw, int n_w, int H, int W, float scaling, int causal, float drop_p,
                                 const unsigned long long* seed, unsigned int stream_id, void* stream) {
  if (c > 64 || c <= 0 || S <= 0 || S > 4096) return -2;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && seed == nullptr)) return -2;
  AttnParams p = make_params(qkv, out, nullptr, nullptr, lse, nullptr, kpm, nseq, S, h, c, ph, pw, n_w, H, W, scaling, causal);
  p.drop_p = drop_p; p.seed = seed; p.stream_id = stream_id;
  return dispatch_attn(dtype, K_FWD, p, (hipStream_t)stream);
}
extern "C" int cvh_attn_fwd(int dtype, const void* qkv, void* out, float* lse, const unsigned char* kpm, int nseq, int S, int h, int c,
                            int ph, int pw, int n_w, int H, int W, float scaling, int causal, void* stream) {
  return cvh_attn_fwd_drop(dtype, qkv, out, lse, kpm, nseq, S, h, c, ph, pw, n_w, H, W, scaling, causal, 0.f, nullptr, 0, stream);
}

extern "C" int cvh_attn_bwd(int dtype, const void* qkv, const void* out, const void* dout, void* dqkv, const float* lse, float* dsum,
                            const unsigned char* kpm, int nseq, int S, int h, int c, int ph, int pw, int n_w, int H, int W, float scaling,
                            int causal, void* stream) {
  return cvh_attn_bwd_drop(dtype, qkv, out, dout, dqkv, lse, dsum, kpm, nseq, S, h, c, ph, pw, n_w, H, W, scaling, causal, 0.f, nullptr, 0, stream);
}

extern "C" int cvh_attn_bwd_drop(int dtype, const void* qkv, const void* out, const void* dout, void* dqkv, const float* lse, float* dsum,
                                 const unsigned char* kpm, int nseq, int S, int h, int c, int ph, int pw, int n_w, int H, int W, float scaling,
                                 int causal, float drop_p, const unsigned long long* seed, unsigned int stream_id, void* stream) {
  if (c > 64 || c <= 0 || S <= 0 || S > 4096) return -2;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && seed == nullptr)) return -2;
  AttnParams p = make_params(qkv, nullptr, dout, dqkv, const_cast<float*>(lse), dsum, kpm, nseq, S, h, c, ph, pw, n_w, H, W, scaling, causal);
  p.drop_p = drop_p; p.seed = seed; p.stream_id = stream_id;
  hipStream_t st = (hipStream_t)stream;
  p.out = const_cast<void*>(out);  // the dQ kernel forms D = rowsum(dO * O) itself
  int rc = dispatch_attn(dtype, K_DQ, p, st);
  if (rc) return rc;
  return dispatch_attn(dtype, K_DKV, p, st);
}

/* general additive attention mask (multi_head_attention.py:197-208): bias [S][S] (bias_stride 0) or [nseq][S][S] float32 */
extern "C" int cvh_attn_fwd_mask(int dtype, const void* qkv, void* out, float* lse, const unsigned char* kpm, const float* bias,
                                 long long bias_stride, int nseq, int S, int h, int c, int ph, int pw, int n_w, int H, int W, float scaling,
                                 int causal, float drop_p, const unsigned long long* seed, unsigned int stream_id, void* stream) {
  if (c > 64 || c <= 0 || S <= 0 || S > 4096) return -2;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && seed == nullptr)) return -2;
  if (bias_stride != 0 && bias_stride < (long long)S * S) return -2;
  AttnParams p = make_params(qkv, out, nullptr, nullptr, lse, nullptr, kpm, nseq, S, h, c, ph, pw, n_w, H, W, scaling, causal);
  p.drop_p = drop_p; p.seed = seed; p.stream_id = stream_id;
  p.bias = bias; p.bias_stride = bias_stride;
  return dispatch_attn(dtype, K_FWD, p, (hipStream_t)stream);
}

extern "C" int cvh_attn_bwd_mask(int dtype, const void* qkv, const void* out, const void* dout, void* dqkv, const float* lse, float* dsum,
                                 const unsigned char* kpm, const float* bias, long long bias_stride, int nseq, int S, int h, int c, int ph,
                                 int pw, int n_w, int H, int W, float scaling, int causal, float drop_p, const unsigned long long* seed,
                                 unsigned int stream_id, void* stream) {
  if (c > 64 || c <= 0 || S <= 0 || S > 4096) return -2;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && seed == nullptr)) return -2;
  if (bias_stride != 0 && bias_stride < (long long)S * S) return -2;
  AttnParams p = make_params(qkv, nullptr, dout, dqkv, const_cast<float*>(lse), dsum, kpm, nseq, S, h, c, ph, pw, n_w, H, W, scaling, causal);
  p.drop_p = drop_p; p.seed = seed; p.stream_id = stream_id;
  p.bias = bias; p.bias_stride = bias_stride;
  hipStream_t st = (hipStream_t)stream;
  p.out = const_cast<void*>(out);
  int rc = dispatch_attn(dtype, K_DQ, p, st);
  if (rc) return rc;
  return dispatch_attn(dtype, K_DKV, p, st);
}

// Operand-resident attention kernels for the short, unmasked sequences of the MobileViT blocks (S <= 256 tokens per sequence, bf16, no causal
// rule / key padding / additive mask / dropout): cvnets/layers/multi_head_attention.py:135-239 on the token matrices of
// cvnets/modules/mobilevit_block.py:186-231.
//
// The tile-streaming kernels of attention.hip re-stage a 64-key K/V tile per loop step behind two workgroup barriers and carry the running
// softmax statistics from tile to tile: at S = 256 that is four dependent (scores -> max -> exp -> sum -> rescale -> P V) chains per wave
// with eight barriers, each chain too short to hide its own LDS / MFMA / transcendental latencies at two waves per SIMD (profiles/
// r06_ab_runs.txt: neither occupancy nor the staging instruction count nor the prefetch distance moved them).  Here a workgroup keeps ALL
// keys of its (sequence, head) group in LDS — 256 rows of K and V (forward, dQ) or of Q and dO (dK/dV), staged once, one barrier — and a
// wave then works through a whole 32-row block against every key with no synchronisation at all:
//   forward : S^T = K Q^T for all keys (8 independent accumulators at S = 256), exact row maximum and sum (no running rescale), P^T from
//             the accumulator registers into O^T += V^T P^T;
//   dQ      : per 64 keys S^T, dP^T -> dS^T -> dQ^T += K^T dS^T, the steps independent of each other apart from the dQ accumulator;
//   dK / dV : the wave's 32 keys against every 32-query block.
// The wave's own operand (its 32 query rows, or its 32 key rows) never touches LDS: the B fragments come straight from global memory
// (16 contiguous bytes per lane and k-step), which keeps the LDS footprint at ~60-75 KB, two workgroups per CU, so one workgroup's staging
// runs under the other's arithmetic.
//
// Work decomposition: 256 resident rows = HPW heads x SP keys (SP = 64 / 128 / 256 for S <= 64 / 128 / 256, HPW = 256 / SP heads of one
// sequence per workgroup); 4 waves x 2 tasks, task = (head, 32-row block).  Sequences with S < SP leave the absent keys masked.
#include "attn_frag.hpp"
#include "cvnets_hip.h"

#ifndef ARES_DBG
#define ARES_DBG 0  // developer builds (tools/build_variant.py): skip pieces of attn_res_fwd_kernel to time them; results are WRONG when non-zero
#endif              // 1 no K / V tile loads after the first item, 16 no output stores, 64 no Q fragment loads after the first task
namespace {

// B-operand fragment of one 16-wide k-step straight from global memory: lane (n = lane & 31, half = lane >> 5) receives the 8 elements
// [k0 + 8 half, k0 + 8 half + 8) of its row (rowp = the row's first column of this head; 8-byte aligned: c % 4 == 0).  Columns >= c are not
// touched in memory (the address is clamped) and must read as zero: gl_frag_raw only ISSUES the two loads, gl_frag_mask zeroes the pieces
// beyond c when the fragment is consumed - a select right behind the load would make the wave wait for it (and for every older load, the
// prefetched tiles included) at the point of issue.
__device__ __forceinline__ Frag<bf16_t> gl_frag_raw(const bf16_t* __restrict__ rowp, int k0, int c, int lane) {
  const int col = k0 + 8 * (lane >> 5);
  const uint2 a = *reinterpret_cast<const uint2*>(rowp + (col + 4 <= c ? col : 0));
  const uint2 b = *reinterpret_cast<const uint2*>(rowp + (col + 8 <= c ? col + 4 : 0));
  Frag<bf16_t> f;
  f.v = __builtin_bit_cast(bf16x8_t, make_uint4(a.x, a.y, b.x, b.y));
  return f;
}
__device__ __forceinline__ Frag<bf16_t> gl_frag_mask(const Frag<bf16_t>& raw, int k0, int c, int lane) {
  const int col = k0 + 8 * (lane >> 5);
  const bool ok0 = col + 4 <= c, ok1 = col + 8 <= c;
  const uint4 u = __builtin_bit_cast(uint4, raw.v);
  Frag<bf16_t> f;
  f.v = __builtin_bit_cast(bf16x8_t, make_uint4(ok0 ? u.x : 0u, ok0 ? u.y : 0u, ok1 ? u.z : 0u, ok1 ? u.w : 0u));
  return f;
}

// "The value is needed HERE": an empty asm that reads and writes the registers makes the compiler place the s_waitcnt of their loads at
// this point.  Used to collect the prefetched tiles / fragments at the end of a task's arithmetic, BEFORE its result stores are issued: the
// wave's memory counter counts loads and stores alike and the compiler cannot count stores that sit behind lane predicates, so a wait placed
// behind them (at the first real use, the top of the next task or item) is vmcnt(0) and drains the stores - one write latency per task.
__device__ __forceinline__ void touch(uint2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
__device__ __forceinline__ void touch(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void touch(Frag<bf16_t>& f) {
  uint4 u = __builtin_bit_cast(uint4, f.v);
  asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w));
  f.v = __builtin_bit_cast(bf16x8_t, u);
}

// sum over the 8 elements of the product of two fragments
__device__ __forceinline__ float frag_dot(const Frag<bf16_t>& x, const Frag<bf16_t>& y) {
  const uint4 a = __builtin_bit_cast(uint4, x.v), b = __builtin_bit_cast(uint4, y.v);
  const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s += __uint_as_float(aw[i] << 16) * __uint_as_float(bw[i] << 16);
    s += __uint_as_float(aw[i] & 0xffff0000u) * __uint_as_float(bw[i] & 0xffff0000u);
  }
  return s;
}

template <int CPK, int NKT> struct ResCfg {
  static constexpr int PK = CPK + 8;   // LDS row pitch (elements): odd multiple of 16 bytes
  static constexpr int SP = 32 * NKT;  // keys per head held by a workgroup
  static constexpr int HPW = 8 / NKT;  // heads per workgroup
  static constexpr int NCH = CPK / 4;  // 8-byte chunks per staged row
  static constexpr int NFC = (CPK + 31) / 32;
  static constexpr int ROWS = 256;
};

// Two [256][CPK] tiles (rows = HPW heads x SP sequence positions, columns = the head's channels, zero from column c on and for absent rows)
// on their way from global memory into LDS through registers, 256 threads: load() only ISSUES the reads (2 x NCH 8-byte pieces per thread)
// so that it can be placed a whole work item ahead of store().  Thread t owns the 8-byte column pieces (t & 3) + 4 k of the rows (t >> 2) + 64 m
// (k < NCH / 4, m < 4): every address is one per-thread base plus compile-time offsets (the obvious idx = t + 256 it, row = idx / NCH mapping
// costs ~40 registers of hoisted per-iteration offsets).
template <int CPK, int NKT> struct ResStager {
  using C = ResCfg<CPK, NKT>;
  static constexpr int NKC = C::NCH / 4;
  uint2 ra[4][NKC], rb[4][NKC];
  unsigned rowok;
  __device__ __forceinline__ void load(const bf16_t* __restrict__ ga, int lda, const bf16_t* __restrict__ gb, int ldb, const int* rows, int head0,
                                       int c, int tid) {
    rowok = 0;
    const int cq = 4 * (tid & 3);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      // tile row (tid >> 2) + 64 m: its head and sequence position are compile-time functions of m
      const int hl = (64 * m) / C::SP, pos = (tid >> 2) + (64 * m) % C::SP;
      const int gr = rows[pos];
      rowok |= (gr >= 0 ? 1u : 0u) << m;
      const size_t row = gr >= 0 ? (size_t)gr : (size_t)0;
      const bf16_t* pa = ga + row * lda + (size_t)(head0 + hl) * c;
      const bf16_t* pb = gb + row * ldb + (size_t)(head0 + hl) * c;
#pragma unroll
      for (int k = 0; k < NKC; ++k) {
        const int col = cq + 16 * k < c ? cq + 16 * k : 0;
        ra[m][k] = *reinterpret_cast<const uint2*>(pa + col);
        rb[m][k] = *reinterpret_cast<const uint2*>(pb + col);
      }
    }
  }
  __device__ __forceinline__ void collect() {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int k = 0; k < NKC; ++k) {
        touch(ra[m][k]);
        touch(rb[m][k]);
      }
  }
  __device__ __forceinline__ void store(bf16_t* la, bf16_t* lb, int c, int tid) const {
    const int cq = 4 * (tid & 3);
    bf16_t* pa = la + (tid >> 2) * C::PK + cq;
    bf16_t* pb = lb + (tid >> 2) * C::PK + cq;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int k = 0; k < NKC; ++k) {
        const bool ok = ((rowok >> m) & 1u) && cq + 16 * k < c;
        *reinterpret_cast<uint2*>(pa + 64 * m * C::PK + 16 * k) = ok ? ra[m][k] : make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(pb + 64 * m * C::PK + 16 * k) = ok ? rb[m][k] : make_uint2(0u, 0u);
      }
  }
};

// Work items (sequence, head group) of one persistent workgroup: item = i + k * (number of workgroups), i = the workgroup's place in the
// XCD-chunked order.  The workgroups of one XCD then work on the head groups of the SAME sequences at the same time, so the cache lines the
// heads of a pixel row share (a head's K or V piece is 72 - 128 bytes of it) are fetched into that XCD's L2 once; a workgroup that walks a
// contiguous range instead meets those lines again 20 us later, after 60 other workgroups have streamed 14 MB through the 4 MB L2.
struct ResRange {
  int first, last, step, ngrp;
};
template <int HPW> __device__ __forceinline__ ResRange res_range(const AttnParams& p) {
  ResRange r;
  r.ngrp = p.h / HPW;
  r.first = xcd_chunk_id(blockIdx.x, gridDim.x);
  r.last = p.nseq * r.ngrp;
  r.step = gridDim.x;
  return r;
}

// the (CPK / 16) raw B fragments (gl_frag_raw: to be masked at their use) of the 32 rows [32 blk, 32 blk + 32) of sequence positions, read from `base` (leading dimension ld, first
// column col0): lane & 31 = row
template <int CPK>
__device__ __forceinline__ void gl_frags(Frag<bf16_t>* f, const bf16_t* __restrict__ base, int ld, int col0, const int* rows, int blk, int S, int c, int lane) {
  const int pos = 32 * blk + (lane & 31);
  const size_t row = (size_t)rows[pos < S ? pos : 0];
  const bf16_t* rp = base + row * ld + col0;
#pragma unroll
  for (int kk = 0; kk < CPK / 16; ++kk) f[kk] = gl_frag_raw(rp, 16 * kk, c, lane);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------------------------
template <int CPK, int NKT>
__global__ __launch_bounds__(256, 2) void attn_res_fwd_kernel(AttnParams p) {
  using C = ResCfg<CPK, NKT>;
  constexpr int PK = C::PK, SP = C::SP, HPW = C::HPW, NFC = C::NFC, NK = CPK / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);  // [256][PK]
  bf16_t* Vs = Ks + 256 * PK;                         // [256][PK]; its transposed reads run up to 8 elements past a row's pitch: the tables follow
  int* rk2 = reinterpret_cast<int*>(Vs + 256 * PK);   // [2][SP] row of every sequence position (-1: absent) of the current and the next item

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ResRange wr = res_range<HPW>(p);
  if (wr.first >= wr.last) return;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(p.qkv);
  const int ld = 3 * p.d;
  const float sc2 = p.scaling * kLog2e;  // scores in log2 units: exp(x) = exp2(x log2 e)

  int s = wr.first / wr.ngrp, head0 = (wr.first - s * wr.ngrp) * HPW;
  for (int i = tid; i < SP; i += 256) rk2[i] = i < p.S ? seq_row(p.map, s, i) : -1;
  __syncthreads();
  // (the first task's fragments are requested BEFORE the first tiles, the order every later item has them in: the compiler merges the
  // outstanding-load counts of the loop's entry and its back edge, and fragments younger than the tiles on either path make the wait in
  // front of every item's first task drain the tile prefetch that has just been issued)
  Frag<bf16_t> bqn[NK];  // the Q fragments of the wave's NEXT task, requested one task ahead
  gl_frags<CPK>(bqn, qkv, ld, (head0 + wave / NKT) * p.c, rk2, wave % NKT, p.S, p.c, lane);
  ResStager<CPK, NKT> stg;
  stg.load(qkv + p.d, ld, qkv + 2 * p.d, ld, rk2, head0, p.c, tid);
  stg.collect();  // (nothing is outstanding when the loop is entered - as on its back edge)
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) touch(bqn[kk]);

#pragma unroll 1
  for (int item = wr.first, parity = 0; item < wr.last; item += wr.step, parity ^= 1) {
    const int buf = parity;
    const int* rk = rk2 + buf * SP;
    int* rkn = rk2 + (buf ^ 1) * SP;
    const bool more = item + wr.step < wr.last;
    // (the last item requests itself again - cache hits nobody consumes: a conditional request leaves the prefetch registers with two
    // definitions, and the copies the compiler then places right behind the loads wait for them at the point of issue)
    const int item_n = more ? item + wr.step : item;
    const int s_n = item_n / wr.ngrp, head0_n = (item_n - s_n * wr.ngrp) * HPW;
    for (int i = tid; i < SP; i += 256) rkn[i] = i < p.S ? seq_row(p.map, s_n, i) : -1;
    stg.store(Ks, Vs, p.c, tid);  // (the barrier that ended the previous item freed the tiles)
    __syncthreads();
    if (!(ARES_DBG & 1)) stg.load(qkv + p.d, ld, qkv + 2 * p.d, ld, rkn, head0_n, p.c, tid);  // the next item's K / V fly under this item's arithmetic

#pragma unroll  // (a rolled loop merges the two tasks' outstanding-load counts: every wait becomes vmcnt(0))
    for (int j = 0; j < 2; ++j) {
      const int tk = wave + 4 * j;
      const int hl = tk / NKT, qb = tk - hl * NKT, head = head0 + hl;
      const int my_q = 32 * qb + (lane & 31);
      const bool q_ok = my_q < p.S;
      const size_t qrow = (size_t)rk[q_ok ? my_q : 0];
      Frag<bf16_t> bq[NK];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) bq[kk] = gl_frag_mask(bqn[kk], 16 * kk, p.c, lane);
      if (ARES_DBG & 64) {
      } else if (j == 0) gl_frags<CPK>(bqn, qkv, ld, (head0 + (tk + 4) / NKT) * p.c, rk, (tk + 4) % NKT, p.S, p.c, lane);
      else gl_frags<CPK>(bqn, qkv, ld, (head0_n + wave / NKT) * p.c, rkn, wave % NKT, p.S, p.c, lane);
      const bf16_t* Kh = Ks + hl * SP * PK;
      const bf16_t* Vh = Vs + hl * SP * PK;

      // the keys in groups of up to 128 (4 accumulators): one group = exact softmax (S <= 128); two groups are merged by the running-maximum
      // rule (one rescale of the output accumulators) - all of S = 256 at once needs 128 accumulator registers on top of the prefetched tiles
      constexpr int GF = NKT < 4 ? NKT : 4, NG = NKT / GF;
      float m_run = -INFINITY, rs = 0.f;  // running maximum (scaled, log2 units) and normaliser
      f32x16_t oacc[NFC];
#pragma unroll
      for (int fc = 0; fc < NFC; ++fc) oacc[fc] = acc_zero();
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        f32x16_t sacc[GF];
#pragma unroll
        for (int f = 0; f < GF; ++f) sacc[f] = acc_zero();
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
#pragma unroll
          for (int f = 0; f < GF; ++f) mma32(sacc[f], lds_frag(Kh, PK, 32 * (GF * g + f), 16 * kk, lane), bq[kk]);  // S^T = K Q^T
          __builtin_amdgcn_sched_barrier(0);  // one k-step's operand reads in flight at a time (registers)
        }
        if (p.S < SP) {  // workgroup-uniform: absent keys can never be seen
          int lim = p.S - 4 * (lane >> 5);
          asm volatile("" : "+v"(lim));  // (keeps the compare results from being hoisted out of the task loop into SGPR pairs)
#pragma unroll
          for (int f = 0; f < GF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (32 * (GF * g + f) + (r & 3) + 8 * (r >> 2) >= lim) sacc[f][r] = -INFINITY;
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int f = 0; f < GF; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[f][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * sc2);  // sc2 > 0: the maximum of the scaled scores
        const f32x2_t sc2v = {sc2, sc2}, mbv = {m_new, m_new};
        f32x2_t rs2 = {0.f, 0.f};
#pragma unroll
        for (int f = 0; f < GF; ++f)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2_t t = f32x2_t{sacc[f][r], sacc[f][r + 1]} * sc2v - mbv;
            const f32x2_t e = {exp2_fast(t[0]), exp2_fast(t[1])};
            rs2 += e;
            sacc[f][r] = e[0];
            sacc[f][r + 1] = e[1];
          }
        float rg = rs2[0] + rs2[1];
        rg += __shfl_xor(rg, 32, 64);
        if (g > 0) {
          const float alpha = exp2_fast(m_run - m_new);
          rs = rs * alpha + rg;
          const f32x2_t av = {alpha, alpha};
#pragma unroll
          for (int fc = 0; fc < NFC; ++fc)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const f32x2_t o = f32x2_t{oacc[fc][r], oacc[fc][r + 1]} * av;
              oacc[fc][r] = o[0];
              oacc[fc][r + 1] = o[1];
            }
        } else {
          rs = rg;
        }
        m_run = m_new;
        // O^T[c][q] += sum_key V[key][c] P^T[key][q]: P^T from the accumulator registers, V read in the matching (permuted) key order
#pragma unroll
        for (int f = 0; f < GF; ++f)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const Frag<bf16_t> bp = frag_from_acc<bf16_t>(sacc[f], t);
#pragma unroll
            for (int fc = 0; fc < NFC; ++fc) mma32(oacc[fc], lds_frag_strided_perm(Vh, PK, 32 * (GF * g + f) + 16 * t, 32 * fc, lane), bp);
            __builtin_amdgcn_sched_barrier(0);
          }
      }
      const float mb = m_run;

      // collect what was requested a task ago (nothing younger is outstanding), then issue this task's stores
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) touch(bqn[kk]);
      if (j == 1) stg.collect();
      if (q_ok) {
        const float inv_l = 1.0f / rs;
        bf16_t* out = reinterpret_cast<bf16_t*>(p.out) + qrow * p.d + head * p.c;
#pragma unroll
        for (int fc = 0; fc < NFC; ++fc)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int cb = 32 * fc + 8 * i + 4 * (lane >> 5);
            if (cb < p.c && (!(ARES_DBG & 16) || oacc[fc][4 * i] == 1.2345f))
              *reinterpret_cast<uint2*>(out + cb) = make_uint2(f2bf_pk(oacc[fc][4 * i] * inv_l, oacc[fc][4 * i + 1] * inv_l),
                                                               f2bf_pk(oacc[fc][4 * i + 2] * inv_l, oacc[fc][4 * i + 3] * inv_l));
          }
        if (lane < 32 && p.lse) p.lse[((size_t)s * p.h + head) * p.S + my_q] = (mb + __log2f(rs)) * kLn2;
      }
    }
    __syncthreads();  // every wave is done with this item's tiles and row table
    s = s_n;
    head0 = head0_n;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// backward dQ (also forms D = rowsum(dO * O) and publishes it for the dK / dV kernel)
// ---------------------------------------------------------------------------------------------------------------------------------------
template <int CPK, int NKT>
__global__ __launch_bounds__(256, 2) void attn_res_dq_kernel(AttnParams p) {
  using C = ResCfg<CPK, NKT>;
  constexpr int PK = C::PK, SP = C::SP, HPW = C::HPW, NFC = C::NFC, NK = CPK / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* Vs = reinterpret_cast<bf16_t*>(smem_raw);  // [256][PK] read along the rows only
  bf16_t* Ks = Vs + 256 * PK;                         // [256][PK] also read transposed (up to 8 elements past a row's pitch: the tables follow)
  int* rk2 = reinterpret_cast<int*>(Ks + 256 * PK);   // [2][SP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ResRange wr = res_range<HPW>(p);
  if (wr.first >= wr.last) return;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(p.qkv);
  const bf16_t* dout = reinterpret_cast<const bf16_t*>(p.dout);
  const bf16_t* outp = reinterpret_cast<const bf16_t*>(p.out);
  const int ld = 3 * p.d;
  const float sc2 = p.scaling * kLog2e;

  int s = wr.first / wr.ngrp, head0 = (wr.first - s * wr.ngrp) * HPW;
  for (int i = tid; i < SP; i += 256) rk2[i] = i < p.S ? seq_row(p.map, s, i) : -1;
  __syncthreads();
  ResStager<CPK, NKT> stg;
  Frag<bf16_t> bqn[NK], bdn[NK], bon[NK];  // Q, dO, O fragments of the wave's NEXT task
  float lsen;
  auto request = [&](const int* rows, int sq, int h0, int tk) __attribute__((always_inline)) {
    const int head = h0 + tk / NKT, blk = tk % NKT;
    gl_frags<CPK>(bqn, qkv, ld, head * p.c, rows, blk, p.S, p.c, lane);
    gl_frags<CPK>(bdn, dout, p.d, head * p.c, rows, blk, p.S, p.c, lane);
    gl_frags<CPK>(bon, outp, p.d, head * p.c, rows, blk, p.S, p.c, lane);
    const int q = 32 * blk + (lane & 31);
    lsen = p.lse[((size_t)sq * p.h + head) * p.S + (q < p.S ? q : 0)];
  };
  request(rk2, s, head0, wave);  // (before the first tiles: see attn_res_fwd_kernel)
  stg.load(qkv + p.d, ld, qkv + 2 * p.d, ld, rk2, head0, p.c, tid);
  auto collect_frags = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      touch(bqn[kk]);
      touch(bdn[kk]);
      touch(bon[kk]);
    }
    touch(lsen);
  };
  stg.collect();  // (nothing is outstanding when the loop is entered - as on its back edge)
  collect_frags();

#pragma unroll 1
  for (int item = wr.first, parity = 0; item < wr.last; item += wr.step, parity ^= 1) {
    const int buf = parity;
    const int* rk = rk2 + buf * SP;
    int* rkn = rk2 + (buf ^ 1) * SP;
    const bool more = item + wr.step < wr.last;
    // (the last item requests itself again - cache hits nobody consumes: a conditional request leaves the prefetch registers with two
    // definitions, and the copies the compiler then places right behind the loads wait for them at the point of issue)
    const int item_n = more ? item + wr.step : item;
    const int s_n = item_n / wr.ngrp, head0_n = (item_n - s_n * wr.ngrp) * HPW;
    for (int i = tid; i < SP; i += 256) rkn[i] = i < p.S ? seq_row(p.map, s_n, i) : -1;
    stg.store(Ks, Vs, p.c, tid);
    __syncthreads();
    stg.load(qkv + p.d, ld, qkv + 2 * p.d, ld, rkn, head0_n, p.c, tid);

#pragma unroll  // (a rolled loop merges the two tasks' outstanding-load counts: every wait becomes vmcnt(0))
    for (int j = 0; j < 2; ++j) {
      const int tk = wave + 4 * j;
      const int hl = tk / NKT, qb = tk - hl * NKT, head = head0 + hl;
      const int my_q = 32 * qb + (lane & 31);
      const bool q_ok = my_q < p.S;
      const size_t qrow = (size_t)rk[q_ok ? my_q : 0];
      const size_t sidx = ((size_t)s * p.h + head) * p.S + (q_ok ? my_q : 0);
      Frag<bf16_t> bq[NK], bd[NK];
      float dpart = 0.f;
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        bq[kk] = gl_frag_mask(bqn[kk], 16 * kk, p.c, lane);
        bd[kk] = gl_frag_mask(bdn[kk], 16 * kk, p.c, lane);
        dpart += frag_dot(bd[kk], bon[kk]);  // (dO is zero beyond c)
      }
      // an absent query contributes nothing: lse = +inf makes every probability 0
      const float lse2 = q_ok ? lsen * kLog2e : INFINITY;
      if (j == 0) request(rk, s, head0, tk + 4);
      else request(rkn, s_n, head0_n, wave);
      const float dsum = dpart + __shfl_xor(dpart, 32, 64);
      if (lane < 32 && q_ok) p.dsum[sidx] = dsum;
      const f32x2_t sc2v = {sc2, sc2}, lsev = {lse2, lse2}, dsv = {dsum, dsum};
      const bf16_t* Kh = Ks + hl * SP * PK;
      const bf16_t* Vh = Vs + hl * SP * PK;

      f32x16_t dqacc[NFC];
#pragma unroll
      for (int fc = 0; fc < NFC; ++fc) dqacc[fc] = acc_zero();
#pragma unroll 2
      for (int f = 0; f < NKT; ++f) {
        f32x16_t sacc = acc_zero(), dpacc = acc_zero();
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          mma32(sacc, lds_frag(Kh, PK, 32 * f, 16 * kk, lane), bq[kk]);   // S^T  = K Q^T
          mma32(dpacc, lds_frag(Vh, PK, 32 * f, 16 * kk, lane), bd[kk]);  // dP^T = V dO^T
        }
        if (p.S < SP) {
          const int lim = p.S - 32 * f - 4 * (lane >> 5);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if ((r & 3) + 8 * (r >> 2) >= lim) sacc[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_t t = f32x2_t{sacc[r], sacc[r + 1]} * sc2v - lsev;
          const f32x2_t pv = {exp2_fast(t[0]), exp2_fast(t[1])};
          const f32x2_t ds = pv * (f32x2_t{dpacc[r], dpacc[r + 1]} - dsv);  // dS^T
          dpacc[r] = ds[0];
          dpacc[r + 1] = ds[1];
        }
        // dQ^T[c][q] += sum_key K[key][c] dS^T[key][q]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const Frag<bf16_t> bs = frag_from_acc<bf16_t>(dpacc, t);
#pragma unroll
          for (int fc = 0; fc < NFC; ++fc) mma32(dqacc[fc], lds_frag_strided_perm(Kh, PK, 32 * f + 16 * t, 32 * fc, lane), bs);
        }
      }

      collect_frags();  // what was requested a task ago, before this task's stores (see touch())
      if (j == 1) stg.collect();
      if (q_ok) {
        bf16_t* dq = reinterpret_cast<bf16_t*>(p.dqkv) + qrow * ld + head * p.c;
        const float sc = p.scaling;
#pragma unroll
        for (int fc = 0; fc < NFC; ++fc)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int cb = 32 * fc + 8 * i + 4 * (lane >> 5);
            if (cb < p.c)
              *reinterpret_cast<uint2*>(dq + cb) = make_uint2(f2bf_pk(dqacc[fc][4 * i] * sc, dqacc[fc][4 * i + 1] * sc),
                                                              f2bf_pk(dqacc[fc][4 * i + 2] * sc, dqacc[fc][4 * i + 3] * sc));
          }
      }
    }
    __syncthreads();
    s = s_n;
    head0 = head0_n;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// backward dK, dV: the scores are formed UNtransposed, S = Q K^T (queries along the accumulator rows, the wave's keys along lane & 31), so
// the rows of P and dS are the contraction index of dV^T = dO^T P and dK^T = Q^T dS; the softmax statistics are per accumulator row and
// come from an LDS table.
// ---------------------------------------------------------------------------------------------------------------------------------------
template <int CPK, int NKT>
__global__ __launch_bounds__(256, 2) void attn_res_dkv_kernel(AttnParams p) {
  using C = ResCfg<CPK, NKT>;
  constexpr int PK = C::PK, SP = C::SP, HPW = C::HPW, NFC = C::NFC, NK = CPK / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem_raw);     // [256][PK]
  bf16_t* dOs = Qs + 256 * PK;                           // [256][PK] (both also read transposed; the tables follow)
  float* st = reinterpret_cast<float*>(dOs + 256 * PK);  // [2][256] lse * log2 e (+inf: absent query), D
  int* rq2 = reinterpret_cast<int*>(st + 512);           // [2][SP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const ResRange wr = res_range<HPW>(p);
  if (wr.first >= wr.last) return;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(p.qkv);
  const bf16_t* dout = reinterpret_cast<const bf16_t*>(p.dout);
  const int ld = 3 * p.d;
  const float sc2 = p.scaling * kLog2e;
  const f32x2_t sc2v = {sc2, sc2};

  int s = wr.first / wr.ngrp, head0 = (wr.first - s * wr.ngrp) * HPW;
  for (int i = tid; i < SP; i += 256) rq2[i] = i < p.S ? seq_row(p.map, s, i) : -1;
  __syncthreads();
  ResStager<CPK, NKT> stg;
  float lse_n, dsum_n;  // this thread's entry of the next item's statistics table (256 threads = HPW x SP entries)
  auto request_item = [&](const int* rows, int sq, int h0) __attribute__((always_inline)) {
    stg.load(qkv, ld, dout, p.d, rows, h0, p.c, tid);
    const int hl = tid / SP, q = tid - hl * SP;
    const bool ok = q < p.S;
    const size_t si = ((size_t)sq * p.h + h0 + hl) * p.S + (ok ? q : 0);
    lse_n = p.lse[si];  // (raw: converted / masked when the table is written, not behind the load)
    dsum_n = p.dsum[si];
  };
  Frag<bf16_t> bkn[NK], bvn[NK];  // K, V fragments of the wave's NEXT task
  auto request = [&](const int* rows, int h0, int tk) __attribute__((always_inline)) {
    const int head = h0 + tk / NKT, blk = tk % NKT;
    gl_frags<CPK>(bkn, qkv, ld, p.d + head * p.c, rows, blk, p.S, p.c, lane);
    gl_frags<CPK>(bvn, qkv, ld, 2 * p.d + head * p.c, rows, blk, p.S, p.c, lane);
  };
  request(rq2, head0, wave);  // (before the first tiles: see attn_res_fwd_kernel)
  request_item(rq2, s, head0);
  auto collect_frags = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      touch(bkn[kk]);
      touch(bvn[kk]);
    }
  };
  auto collect_item = [&]() __attribute__((always_inline)) {
    stg.collect();
    touch(lse_n);
    touch(dsum_n);
  };
  collect_item();  // (nothing is outstanding when the loop is entered - as on its back edge)
  collect_frags();

#pragma unroll 1
  for (int item = wr.first, parity = 0; item < wr.last; item += wr.step, parity ^= 1) {
    const int buf = parity;
    const int* rq = rq2 + buf * SP;
    int* rqn = rq2 + (buf ^ 1) * SP;
    const bool more = item + wr.step < wr.last;
    // (the last item requests itself again - cache hits nobody consumes: a conditional request leaves the prefetch registers with two
    // definitions, and the copies the compiler then places right behind the loads wait for them at the point of issue)
    const int item_n = more ? item + wr.step : item;
    const int s_n = item_n / wr.ngrp, head0_n = (item_n - s_n * wr.ngrp) * HPW;
    for (int i = tid; i < SP; i += 256) rqn[i] = i < p.S ? seq_row(p.map, s_n, i) : -1;
    stg.store(Qs, dOs, p.c, tid);
    {
      const bool ok = tid % SP < p.S;
      st[tid] = ok ? lse_n * kLog2e : INFINITY;
      st[256 + tid] = ok ? dsum_n : 0.f;
    }
    __syncthreads();
    request_item(rqn, s_n, head0_n);

#pragma unroll  // (a rolled loop merges the two tasks' outstanding-load counts: every wait becomes vmcnt(0))
    for (int j = 0; j < 2; ++j) {
      const int tk = wave + 4 * j;
      const int hl = tk / NKT, kb = tk - hl * NKT, head = head0 + hl;
      const int my_key = 32 * kb + (lane & 31);
      const bool key_ok = my_key < p.S;
      const size_t krow = (size_t)rq[key_ok ? my_key : 0];
      Frag<bf16_t> bk[NK], bv[NK];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        bk[kk] = gl_frag_mask(bkn[kk], 16 * kk, p.c, lane);
        bv[kk] = gl_frag_mask(bvn[kk], 16 * kk, p.c, lane);
      }
      if (j == 0) request(rq, head0, tk + 4);
      else request(rqn, head0_n, wave);
      const bf16_t* Qh = Qs + hl * SP * PK;
      const bf16_t* dOh = dOs + hl * SP * PK;
      const float* sth = st + hl * SP;

      f32x16_t dkacc[NFC], dvacc[NFC];
#pragma unroll
      for (int fc = 0; fc < NFC; ++fc) { dkacc[fc] = acc_zero(); dvacc[fc] = acc_zero(); }
#pragma unroll 2
      for (int f = 0; f < NKT; ++f) {
        f32x16_t sacc = acc_zero(), dpacc = acc_zero();
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          mma32(sacc, lds_frag(Qh, PK, 32 * f, 16 * kk, lane), bk[kk]);    // S  = Q K^T
          mma32(dpacc, lds_frag(dOh, PK, 32 * f, 16 * kk, lane), bv[kk]);  // dP = dO V^T
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 l4 = *reinterpret_cast<const float4*>(sth + 32 * f + 8 * i + 4 * half);
          const float4 d4 = *reinterpret_cast<const float4*>(sth + 256 + 32 * f + 8 * i + 4 * half);
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const f32x2_t lv = e == 0 ? f32x2_t{l4.x, l4.y} : f32x2_t{l4.z, l4.w};
            const f32x2_t dv = e == 0 ? f32x2_t{d4.x, d4.y} : f32x2_t{d4.z, d4.w};
            const f32x2_t t = f32x2_t{sacc[4 * i + e], sacc[4 * i + e + 1]} * sc2v - lv;
            const f32x2_t pv = {exp2_fast(t[0]), exp2_fast(t[1])};
            const f32x2_t ds = pv * (f32x2_t{dpacc[4 * i + e], dpacc[4 * i + e + 1]} - dv);
            sacc[4 * i + e] = pv[0];
            sacc[4 * i + e + 1] = pv[1];
            dpacc[4 * i + e] = ds[0];
            dpacc[4 * i + e + 1] = ds[1];
          }
        }
        // dV^T[c][key] += sum_q dO[q][c] P[q][key] ;  dK^T[c][key] += sum_q Q[q][c] dS[q][key]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const Frag<bf16_t> bp = frag_from_acc<bf16_t>(sacc, t);
          const Frag<bf16_t> bs = frag_from_acc<bf16_t>(dpacc, t);
#pragma unroll
          for (int fc = 0; fc < NFC; ++fc) {
            mma32(dvacc[fc], lds_frag_strided_perm(dOh, PK, 32 * f + 16 * t, 32 * fc, lane), bp);
            mma32(dkacc[fc], lds_frag_strided_perm(Qh, PK, 32 * f + 16 * t, 32 * fc, lane), bs);
          }
        }
      }

      collect_frags();  // what was requested a task ago, before this task's stores (see touch())
      if (j == 1) collect_item();
      if (key_ok) {
        bf16_t* dk = reinterpret_cast<bf16_t*>(p.dqkv) + krow * ld + p.d + head * p.c;
        const float sc = p.scaling;
#pragma unroll
        for (int fc = 0; fc < NFC; ++fc)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int cb = 32 * fc + 8 * i + 4 * half;
            if (cb < p.c) {
              *reinterpret_cast<uint2*>(dk + cb) = make_uint2(f2bf_pk(dkacc[fc][4 * i] * sc, dkacc[fc][4 * i + 1] * sc),
                                                              f2bf_pk(dkacc[fc][4 * i + 2] * sc, dkacc[fc][4 * i + 3] * sc));
              *reinterpret_cast<uint2*>(dk + p.d + cb) = make_uint2(f2bf_pk(dvacc[fc][4 * i], dvacc[fc][4 * i + 1]),
                                                                    f2bf_pk(dvacc[fc][4 * i + 2], dvacc[fc][4 * i + 3]));
            }
          }
      }
    }
    __syncthreads();
    s = s_n;
    head0 = head0_n;
  }
}

template <int CPK, int NKT> size_t res_smem(int which) {
  using C = ResCfg<CPK, NKT>;
  return (size_t)2 * 256 * C::PK * 2 + (size_t)2 * C::SP * 4 + (which == 2 ? 512 * 4 : 0);
}

template <int CPK, int NKT> int res_launch(int which, const AttnParams& p, hipStream_t st) {
  using C = ResCfg<CPK, NKT>;
  if (p.h % C::HPW) return -100;
  const size_t smem = res_smem<CPK, NKT>(which);
  // persistent workgroups, two per CU: each works through a contiguous range of (sequence, head group) items
  const int nitems = p.nseq * (p.h / C::HPW);
  int wgs = cvh_tune_get(26);
  if (wgs <= 0) wgs = 512;
  const dim3 grid(nitems < wgs ? nitems : wgs);
  if (which == 0) {
    static DynSmemAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(attn_res_fwd_kernel<CPK, NKT>), smem); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((attn_res_fwd_kernel<CPK, NKT>), grid, dim3(256), smem, st, p);
  } else if (which == 1) {
    static DynSmemAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(attn_res_dq_kernel<CPK, NKT>), smem); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((attn_res_dq_kernel<CPK, NKT>), grid, dim3(256), smem, st, p);
  } else {
    static DynSmemAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(attn_res_dkv_kernel<CPK, NKT>), smem); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((attn_res_dkv_kernel<CPK, NKT>), grid, dim3(256), smem, st, p);
  }
  CVH_CHECK_LAUNCH();
  return 0;
}

template <int CPK> int res_dispatch_s(int which, const AttnParams& p, hipStream_t st) {
  if (p.S <= 64) return res_launch<CPK, 2>(which, p, st);
  if (p.S <= 128) return res_launch<CPK, 4>(which, p, st);
  return res_launch<CPK, 8>(which, p, st);
}

}  // namespace

// which: 0 forward, 1 dQ (+ D), 2 dK / dV.  Returns -100 when the call is outside the resident kernels' domain (the caller then runs the
// tile-streaming kernels of attention.hip): bf16 only, no mask / dropout feature, 32 < S <= 256, head width a multiple of 4 in 33 ... 64,
// a head count the workgroup's head group divides.  CVH_TUNE key 25: 1 = never, 2 = only S > 128, 3 = only S <= 128 (A/B runs).
int attn_res_launch(int dtype, int which, const AttnParams& p, hipStream_t st) {
  if (dtype != CVH_DT_BF16) return -100;
  if (p.causal || p.kpm != nullptr || p.bias != nullptr || p.drop_p > 0.f) return -100;
  if (p.S <= 32 || p.S > 256 || (p.c % 4) != 0 || p.c <= 32 || p.c > 64 || p.scaling <= 0.f) return -100;
  const int tune = cvh_tune_get(25);
  if (tune == 1 || (tune == 2 && p.S <= 128) || (tune == 3 && p.S > 128)) return -100;
  if (p.c <= 48) return res_dispatch_s<48>(which, p, st);
  return res_dispatch_s<64>(which, p, st);
}

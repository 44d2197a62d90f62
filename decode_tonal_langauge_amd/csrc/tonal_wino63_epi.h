// Epilogues of the V-form F(6,3) NT kernel (tonal_wino63.hip), round 4.  Same arithmetic contracts as tonal_wino43v_epi.h
// (reference: models/synthesis_models.py:87-97 and their backward); the unit of work is a HEX: six conv rows from eight
// Winograd products (points 0, +-1, +-2, +-1/2, inf - the set cleared by oracle/winograd_f63_gate.py).
//
// The kernel hands the matrix pipe the hexes of a wave in the permuted order of the F(4,3) kernel: a lane (lr, lh) owns ONE
// column and SIXTEEN CONSECUTIVE hexes, accumulator element e = hex 16 lh + e of the wave's 32: 96 consecutive conv rows,
// 48 consecutive pooled rows, 8 next-stage hexes.  Row logic is half-wave uniform and runs on the scalar ALU; 1-bit words
// are lane masks; stores are buffer stores with scalar row offsets (tonal_wino43v_epi.h explains each of these).
#pragma once
#include "tonal_common.h"
#include "tonal_wino43_epi.h"
#include "tonal_wino43v_epi.h"

namespace tl {

// diagnostic builds (-DV6_STAMP=1, tonal_wino63.hip): lane 0 of wave 0 writes s_memtime into its workgroup's slot; nullptr otherwise
__device__ __forceinline__ void v6_stamp(unsigned long long* st, int k) {
  if (st != nullptr) st[k] = __builtin_amdgcn_s_memtime();
}

#ifndef V6_ST_AUX
#define V6_ST_AUX 0
#endif
constexpr unsigned V6_DROP = 0x80000000u;     // added to any in-range byte offset (< 2^31) it stays past every resource
constexpr unsigned long long V6_HI = 0xffffffff00000000ull;

// A 4-byte buffer store at (per-lane offset vo) + (scalar offset so) + (compile-time constant k).  Interior tiles (IMM: no
// lane carries a dropped-store offset) put k into the instruction's immediate - 512 distinct scalar offsets per tile would
// spill; elsewhere k rides in the scalar offset: measured on gfx950, a store whose per-lane offset is the out-of-range
// marker is NOT reliably dropped once an immediate is added to it (a lane of a partial tile lost its stores, differently
// from run to run).
template <bool IMM>
__device__ __forceinline__ void v6_store_at(float x, __amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned so, unsigned k) {
  // V6_ST_AUX: cache policy of the bulk V / Y / Vd stores (0 default; 2 nt; 16 sc1 = write through and drop the line from the
  // XCD's L2 - the consumers are later launches: measured in round 5, see profiles/r05_kernel_notes.md)
  if constexpr (IMM) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rs, vo + k, so, V6_ST_AUX);
  else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rs, vo, so + k, V6_ST_AUX);
}

// [aL | aH], [bL | bH] -> [aL | bL], [aH | bH]   (L / H: lanes 0-31 / 32-63; v_permlane32_swap)
// The hexes a lane owns are consecutive, so the two hexes of a pair (the 2 x 32 B that are contiguous in the pair layout) sit
// in ONE lane: stored as they are, a store instruction writes 32-byte runs - and partial-line writes are read-modify-write at
// the memory side (PMC: the Y / Vd-writing epilogue fetched 30 GB it never reads).  After the swap, lanes 0-31 carry hex h of
// lane-half 0's pair and lanes 32-63 hex h + 1 of the SAME pair and columns: one instruction writes 64-byte runs.
__device__ __forceinline__ void v6_pair_swap(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// The eight transforms of a hex PAIR of both lane-halves (ha / hb: the hexes h and h + 1 every lane owns) to memory in whole
// 128-byte lines of the pair layout.  v6_pair_swap puts hex h of lane-half 0's pair into lanes 0-31 and hex h + 1 of the same
// pair into lanes 32-63 (ha: half 0's pair, hb: half 1's); v_permlane16_swap then pairs transforms t, t + 1: lanes 0-15 / 16-31
// of a 32-lane row = transform t / t + 1 of the wave's columns 0-15 (second register: columns 16-31).  A line of the layout is
// [transform t: hex 0 (8 channels), hex 1][transform t + 1: hex 0, hex 1]: one store instruction now writes two whole lines
// (stored lane by lane it wrote 32-byte runs: read-modify-write at the memory side - the PMC counters showed 30 GB of reads
// the Y / Vd-writing epilogue never asks for).  voP / voQ: the per-lane offsets of the two registers (v6_pair_offsets).
template <bool IMM>
__device__ __forceinline__ void v6_store_hex_pair(float (&ha)[8], float (&hb)[8], __amdgpu_buffer_rsrc_t rs, unsigned voAP, unsigned voAQ,
                                                  unsigned voBP, unsigned voBQ, unsigned so) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v6_pair_swap(ha[i], hb[i]);
#pragma unroll
  for (int t = 0; t < 8; t += 2) {
    {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ha[t]), __float_as_uint(ha[t + 1]), false, false);
      v6_store_at<IMM>(__uint_as_float(r[0]), rs, voAP, so, (unsigned)(t * 64));
      v6_store_at<IMM>(__uint_as_float(r[1]), rs, voAQ, so, (unsigned)(t * 64));
    }
    {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(hb[t]), __float_as_uint(hb[t + 1]), false, false);
      v6_store_at<IMM>(__uint_as_float(r[0]), rs, voBP, so, (unsigned)(t * 64));
      v6_store_at<IMM>(__uint_as_float(r[1]), rs, voBQ, so, (unsigned)(t * 64));
    }
  }
}
// per-lane byte offset of the first register of v6_store_hex_pair inside its hex pair (the second: + 1024): 8-channel chunk
// (colbase / 8 + bit 3 of lr), transform parity (bit 4 of lr), hex of the pair (lh), channel (lr & 7)
__device__ __forceinline__ unsigned v6_pair_offset(int colbase, int lr, int lh) {
  return (unsigned)((colbase >> 3) + ((lr >> 3) & 1)) * 512u + (unsigned)((lr >> 4) & 1) * 64u + (unsigned)lh * 32u + (unsigned)(lr & 7) * 4u;
}

// the six conv rows of a hex from its eight products: y = A^T m,
//   A^T = [1 1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 -1/2 0; 0 1 1 4 4 1/4 1/4 0; 0 1 -1 8 -8 1/8 -1/8 0; 0 1 1 16 16 1/16 1/16 0;
//          0 1 -1 32 -32 1/32 -1/32 1]
__device__ __forceinline__ void wino63_rows(const f32x16 (&acc)[8], int e, float (&y)[6]) {
  const float m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e], m5 = acc[5][e], m6 = acc[6][e];
  const float a12 = m1 + m2, s12 = m1 - m2, a34 = m3 + m4, s34 = m3 - m4, a56 = m5 + m6, s56 = m5 - m6;
  y[0] = ((acc[0][e] + a12) + a34) + a56;
  y[1] = fmaf(0.5f, s56, fmaf(2.f, s34, s12));
  y[2] = fmaf(0.25f, a56, fmaf(4.f, a34, a12));
  y[3] = fmaf(0.125f, s56, fmaf(8.f, s34, s12));
  y[4] = fmaf(0.0625f, a56, fmaf(16.f, a34, a12));
  y[5] = fmaf(0.03125f, s56, fmaf(32.f, s34, s12)) + acc[7][e];
}
// V = B^T d of eight rows (the input transform of a hex),
//   B^T = [-1 0 21/4 0 -21/4 0 1 0; 0 1 1 -17/4 -17/4 1 1 0; 0 -1 1 17/4 -17/4 -1 1 0; 0 1/2 1/4 -5/2 -5/4 2 1 0;
//          0 -1/2 1/4 5/2 -5/4 -2 1 0; 0 2 4 -5/2 -5 1/2 1 0; 0 -2 4 5/2 -5 -1/2 1 0; 0 -1 0 21/4 0 -21/4 0 1]
// every product fused (one rounding per fmaf): all writers of V / Vd agree bit for bit whatever vector width they use
__device__ __forceinline__ void wino63_bt(const float (&d)[8], float (&v)[8]) {
  v[0] = fmaf(5.25f, d[2] - d[4], d[6] - d[0]);
  const float e1 = fmaf(-4.25f, d[4], d[2] + d[6]), o1 = fmaf(-4.25f, d[3], d[1] + d[5]);
  v[1] = e1 + o1;
  v[2] = e1 - o1;
  const float e2 = fmaf(0.25f, d[2], fmaf(-1.25f, d[4], d[6])), o2 = fmaf(0.5f, d[1], fmaf(-2.5f, d[3], 2.f * d[5]));
  v[3] = e2 + o2;
  v[4] = e2 - o2;
  const float e3 = fmaf(4.f, d[2], fmaf(-5.f, d[4], d[6])), o3 = fmaf(2.f, d[1], fmaf(-2.5f, d[3], 0.5f * d[5]));
  v[5] = e3 + o3;
  v[6] = e3 - o3;
  v[7] = fmaf(5.25f, d[3] - d[5], d[7] - d[1]);
}

// lane's bit j of w set ? a : b without a lane mask in scalar registers: sign-extended bit field + bit-field insert.  (The
// epilogues below transpose the 1-bit words of their rows once per tile - bit_transpose32 - so that a lane holds the bits
// of ITS column for 32 rows; the F(4,3) epilogues broadcast a row's word with two v_readlane per use, and with 96 rows per
// lane those scalar masks spilled through v_writelane: ~400 of 2 100 instructions of the fused conv1 epilogue.)
__device__ __forceinline__ float selbit(uint32_t w, int j, float a, float b) {
  const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)w, j, 1);
  return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}

// Row bookkeeping of one wave tile.  Hw: first hex of the wave; rows R = 6 Hw + 96 lh + r, r < 96.
struct v6_rows {
  int tA, tB;                 // time index of the first conv row of half 0 / half 1
  long long seqA, seqB;       // their sequences
};
__device__ __forceinline__ v6_rows v6_rows_of(long long Hw, int Tp) {
  v6_rows r;
  const unsigned R = (unsigned)(6 * Hw);                    // host-checked: M < 2^31
  const unsigned s = R / (unsigned)Tp;
  r.seqA = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)s);
  r.tA = __builtin_amdgcn_readfirstlane((int)(R - s * (unsigned)Tp));
  r.tB = r.tA + 96;
  r.seqB = r.seqA;
  while (r.tB >= Tp) {
    r.tB -= Tp;
    ++r.seqB;
  }
  return r;
}
// 96 half-wave-uniform row flags
struct bits96 {
  unsigned long long lo;      // rows 0..63
  uint32_t hi;                // rows 64..95
};
// lane mask of row j (compile-time j after unrolling): bit j of a for lanes 0-31, of b for lanes 32-63
__device__ __forceinline__ unsigned long long mask96(const bits96& a, const bits96& b, int j) {
  return j < 64 ? mask2l(a.lo, b.lo, j) : mask2(a.hi, b.hi, j - 64);
}
__device__ __forceinline__ bits96 v6_in_bits96(long long R0h, long long M) {
  bits96 r;
  r.lo = v5_in_bits<64, 1>(R0h, M);
  r.hi = (uint32_t)v5_in_bits<32, 1>(R0h + 64, M);
  return r;
}
__device__ __forceinline__ bits96 v6_valid_bits96(int t0, int Tp, int tlim, long long R0h, long long M) {
  bits96 r;
  r.lo = v5_valid_bits<64, 1>(t0, Tp, tlim, R0h, M);
  int t1 = t0 + 64;
  while (t1 >= Tp) t1 -= Tp;
  r.hi = (uint32_t)v5_valid_bits<32, 1>(t1, Tp, tlim, R0h + 64, M);
  return r;
}

// ------------------------------------------------------------------------------------------
// Forward: bias + LeakyReLU + max-pool (2,1) + arg-max / sign bits.
//   VOUT false: pooled rows, arg-max and sign words go out in the layout [seq * out_tp + t'] (out_tp = p.out_tp, or Tp / 2
//               when 0; rows t' >= out_tp are dropped): the row stride of the stage's OUTPUT is decoupled from the hex
//               padding of its input (conv3 of the reference stack: Tp / 2 = 51 is odd, the next stage pools pairs).
//   VOUT true:  V = B^T d of the pooled output for the next stage's F(6,3) kernels (next-stage hex H' of a sequence = its
//               pooled rows 6 H' .. 6 H' + 7; Tp % 12 == 0), bits in the [seq * Tp / 2 + t'] layout; p.out optional (tests).
// xch: 8 x 64 x 2 floats of LDS that nothing else uses; the function holds ONE workgroup barrier when VOUT.
// ------------------------------------------------------------------------------------------
template <bool VOUT, bool FULL>
__device__ __forceinline__ void v6_epilogue_pool(const tl_nt_params& p, const f32x16 (&acc)[8], const v5_pre_pool& pre, float* xch,
                                                 long long R0, int n0, int wm, int wn, int lr_in, int lh, long long tm,
                                                 unsigned long long* st = nullptr) {
  int lr = lr_in;                                           // (opaque copy: see v6_epilogue_c1w)
  asm volatile("" : "+v"(lr));
  const int colbase = n0 + wn * 32;
  const bool colok = colbase < p.N;                         // N % 32 == 0 (host-checked)
  const int col = colbase + lr;
  const long long Hw = tm * 128 + wm * 32;
  const int Tp = p.Tp, Tq = Tp >> 1;
  const v6_rows rw = v6_rows_of(Hw, Tp);
  // pooled row j of a half (conv rows 2 j, 2 j + 1 of its 96) is an output row
  const unsigned long long vA = v5_valid_bits<48, 2>(rw.tA, Tp, p.Tvalid, 6 * Hw, p.M);
  const unsigned long long vB = v5_valid_bits<48, 2>(rw.tB, Tp, p.Tvalid, 6 * Hw + 96, p.M);
  const float bv = pre.bv;
  const long long P0 = 3 * Hw;                              // first pooled row of the wave (hex layout)
  const long long prows = p.M >> 1;
  const int out_tp = (!VOUT && p.out_tp > 0) ? p.out_tp : Tq;
  const int tqa0 = rw.tA >> 1, tqb0 = rw.tB >> 1;
  // rows that are stored: pooled time below out_tp, inside the matrix (pad rows below out_tp get zeros, as before)
  const unsigned long long kA = v5_valid_bits<48, 1>(tqa0, Tq, out_tp, P0, prows);
  const unsigned long long kB = v5_valid_bits<48, 1>(tqb0, Tq, out_tp, P0 + 48, prows);
  const long long nseq = p.M / Tp;
  const long long obase = rw.seqA * out_tp;                 // output row of (seqA, t' = 0): every offset below is >= 0
  const __amdgpu_buffer_rsrc_t rsO = rsrc_of(!VOUT ? p.out + obase * (long long)p.ldo : nullptr,
                                            !VOUT ? (nseq * out_tp - obase) * (long long)p.ldo * 4 : 0);
  const unsigned ldo4 = (unsigned)p.ldo * 4u;
  const unsigned col4 = colok ? (unsigned)col * 4u : V6_DROP;
  int ta = tqa0, tb = tqb0;
  unsigned ra = (unsigned)tqa0 * ldo4, rb = ((unsigned)(rw.seqB - rw.seqA) * (unsigned)out_tp + (unsigned)tqb0) * ldo4;
  const unsigned seq_step = (unsigned)(out_tp - Tq) * ldo4;  // (wraps: added when a half enters its next sequence)
  float pv[48];
  uint32_t wb0 = 0, wb1 = 0, ws0 = 0, ws1 = 0;
  v6_stamp(st, 8);
  static_for<0, 16>([&](auto E) {
    constexpr int e = decltype(E)::value;
    float y[6];
    wino63_rows(acc, e, y);
    unsigned vo_e = 0;
    if constexpr (!VOUT) vo_e = selmu(V6_HI, rb, ra) + col4;
    static_for<0, 3>([&](auto H) {
      constexpr int h = decltype(H)::value, j = 3 * e + h;
      const float y0 = lrelu01(y[2 * h] + bv, p.slope), y1 = lrelu01(y[2 * h + 1] + bv, p.slope);
      const bool valid = __builtin_amdgcn_inverse_ballot_w64(mask2l(vA, vB, j));
      const bool gt = (y1 > y0) && valid;
      const float o = valid ? (gt ? y1 : y0) : 0.f;
      pv[j] = o;
      if constexpr (j < 32) {
        wb0 = wb0 + wb0 + (uint32_t)gt;
        ws0 = ws0 + ws0 + (uint32_t)(o > 0.f);
      } else {
        wb1 = wb1 + wb1 + (uint32_t)gt;
        ws1 = ws1 + ws1 + (uint32_t)(o > 0.f);
      }
      if constexpr (!VOUT) {
        const unsigned vo = selmu(mask2l(kA, kB, j), vo_e, V5_OOB);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsO, vo, (unsigned)h * ldo4, 0);
      }
    });
    if constexpr (!VOUT) {
      // the next hex of each half: three pooled rows on, into the next sequence where the time index wraps (selects, no
      // branches: a branch per hex would cut the epilogue into basic blocks)
      ta += 3;
      ra += 3u * ldo4;
      const bool wa = ta >= Tq;
      ta -= wa ? Tq : 0;
      ra += wa ? seq_step : 0u;
      tb += 3;
      rb += 3u * ldo4;
      const bool wbb = tb >= Tq;
      tb -= wbb ? Tq : 0;
      rb += wbb ? seq_step : 0u;
    }
  });
  v6_stamp(st, 9);
  {
    // bit words: after the transposes lane lr of a half holds those of its pooled rows lr (block 0) and 32 + lr (block 1,
    // lr < 16).  Buffer stores with a constant count per wave and tile (v6_stores below).
    wb0 = bit_transpose32(__builtin_bitreverse32(wb0), lr);
    ws0 = bit_transpose32(__builtin_bitreverse32(ws0), lr);
    wb1 = bit_transpose32(__builtin_bitreverse32(wb1) >> 16, lr);
    ws1 = bit_transpose32(__builtin_bitreverse32(ws1) >> 16, lr);
    const long long wleft = (nseq * out_tp - obase) * (long long)p.ld_obits * 4;
    const long long wbase = obase * (long long)p.ld_obits + (colbase >> 5);
    const __amdgpu_buffer_rsrc_t rsB = rsrc_of(p.obits + wbase, wleft - (long long)(colbase >> 5) * 4);
    const __amdgpu_buffer_rsrc_t rsS = rsrc_of(p.osign ? p.osign + wbase : nullptr, p.osign ? wleft - (long long)(colbase >> 5) * 4 : 0);
    const unsigned t0h = (unsigned)(lh ? tqb0 : tqa0);
    const unsigned s0h = lh ? (unsigned)(rw.seqB - rw.seqA) : 0u;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const unsigned r = 32u * b + (unsigned)lr;
      unsigned t = t0h + r;
      const unsigned c = t / (unsigned)Tq;
      t -= c * (unsigned)Tq;
      const bool keep = colok && r < 48u && t < (unsigned)out_tp && (P0 + 48 * lh + r) < prows;
      const unsigned wo = keep ? ((s0h + c) * (unsigned)out_tp + t) * (unsigned)p.ld_obits * 4u : V5_OOB;
      __builtin_amdgcn_raw_buffer_store_b32(b ? wb1 : wb0, rsB, wo, 0u, 0);
      __builtin_amdgcn_raw_buffer_store_b32(b ? ws1 : ws0, rsS, wo, 0u, 0);
    }
  }
  v6_stamp(st, 10);
  if constexpr (VOUT) {
    if (p.out != nullptr) {                                 // (tests, TONAL_STORE_P1: the raw pooled rows as well, [seq * Tp / 2 + t'])
      const __amdgpu_buffer_rsrc_t rsP = rsrc_of(p.out + P0 * (long long)p.ldo, (prows - P0) * (long long)p.ldo * 4);
      const unsigned pvoff = colok ? ((unsigned)(48 * lh) * (unsigned)p.ldo + (unsigned)col) * 4u : V5_OOB;
#pragma unroll
      for (int j = 0; j < 48; ++j) {
        const unsigned vo = selmu(mask2l(kA, kB, j), pvoff, V5_OOB);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv[j]), rsP, vo, (unsigned)j * (unsigned)p.ldo * 4u, 0);
      }
    }
    // ---- V of the pooled output: next-stage hex H' of a half = its pooled rows 6 H' .. 6 H' + 7 ----
    const int slot = wm * 2 + lh;
    {
      float2 v2 = {pv[0], pv[1]};
      *reinterpret_cast<float2*>(xch + (slot * 64 + wn * 32 + lr) * 2) = v2;
    }
    // (not __syncthreads(): its fence would also wait for the vector-memory operations in flight - the next tile's LDS-DMA)
    __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    v6_stamp(st, 11);
    float nb0 = 0.f, nb1 = 0.f;
    if (slot < 7) {                                         // (the last half-wave of a tile: finished by the fix-up pass)
      const float2 v2 = *reinterpret_cast<const float2*>(xch + ((slot + 1) * 64 + wn * 32 + lr) * 2);
      nb0 = v2.x;
      nb1 = v2.y;
    }
    // hex H' ends its sequence (rows 6, 7 belong to the next one: zero): pooled time of its first row == Tq - 6
    uint32_t seA = 0, seB = 0, nvA = 0, nvB = 0;
    {
      int t_a = tqa0, t_b = tqb0;
      const long long qa = (Hw >> 1), qb = (Hw >> 1) + 8;   // next-stage hex index of H' = 0
      for (int k = 0; k < 8; ++k) {
        seA |= (uint32_t)(t_a == Tq - 6) << k;
        seB |= (uint32_t)(t_b == Tq - 6) << k;
        nvA |= (uint32_t)(FULL || qa + k < p.vout_quads) << k;
        nvB |= (uint32_t)(FULL || qb + k < p.vout_quads) << k;
        t_a += 6;
        if (t_a >= Tq) t_a -= Tq;
        t_b += 6;
        if (t_b >= Tq) t_b -= Tq;
      }
    }
    const long long Hn = Hw >> 1;                           // first next-stage hex of the wave (Hw % 32 == 0)
    // pair layout (tonal_wino63.hip): hex pair stride 16 ld_vout floats, 8-channel chunk 128, transform 16, hex % 2 8
    const __amdgpu_buffer_rsrc_t rsV = rsrc_of(p.vout + Hn * 8 * (long long)p.ld_vout,
                                              (p.vout_quads - Hn) * 8 * (long long)p.ld_vout * 4);
    const unsigned pair4 = (unsigned)p.ld_vout * 64u;        // bytes per hex pair
    // after v6_pair_swap: lanes 0-31 / 32-63 = hex q / q + 1 of a pair; first the pairs of lane-half 0's eight hexes, then half 1's
    const unsigned offP = v6_pair_offset(colbase, lr, lh);
    const unsigned vvA = colok ? offP : V6_DROP, vvB = colok ? 4u * pair4 + offP : V6_DROP;
    const unsigned long long mraw = wm == 3 ? V6_HI : 0ull;   // H' = 7 of the tile's last half-wave
#pragma unroll
    for (int qp = 0; qp < 4; ++qp) {
      float vv[2][8];
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        const int q = 2 * qp + hq;
        const unsigned long long mend = mask2(seA, seB, q);
        float d[8];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = pv[6 * q + k];
        d[6] = selm(mend, 0.f, q < 7 ? pv[(6 * q + 6) % 48] : nb0);
        d[7] = selm(mend, 0.f, q < 7 ? pv[(6 * q + 7) % 48] : nb1);
        wino63_bt(d, vv[hq]);
        if (q == 7) {                                       // raw rows for tl_wino63_v_fixup (which owns rows 6, 7 of this hex)
#pragma unroll
          for (int k = 0; k < 6; ++k) vv[hq][k] = selm(mraw, d[k], vv[hq][k]);
        }
      }
      const int q = 2 * qp;
      const unsigned voA = selmu(mask2(nvA, nvA >> 1, q), vvA, V6_DROP), voB = selmu(mask2(nvB, nvB >> 1, q), vvB, V6_DROP);
      v6_store_hex_pair<FULL>(vv[0], vv[1], rsV, voA, voA + 1024u, voB, voB + 1024u, (unsigned)qp * pair4);
    }
    v6_stamp(st, 12);
    // the tile's first two pooled rows: rows 6, 7 of the last hex of the tile in front (tl_wino63_v_fixup)
    {
      const bool hw = wm == 0 && p.vhalo != nullptr;
      const __amdgpu_buffer_rsrc_t rsH = rsrc_of(hw ? p.vhalo + tm * 2 * (long long)p.N : nullptr, hw ? 2LL * p.N * 4 : 0);
      const unsigned ho = (colok && lh == 0) ? (unsigned)col * 4u : V5_OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv[0]), rsH, ho, 0u, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv[1]), rsH, ho, (unsigned)p.N * 4u, 0);
    }
  }
}

// Vector-memory stores every wave issues per tile (lower bound where a branch adds some), see v5_stores
template <int EPI>
constexpr int v6_stores() {
  return EPI == W_EPI_POOL ? 52 : EPI == W_EPI_POOLV ? 70 : (EPI == W_EPI_MASK || EPI == W_EPI_LRELU) ? 96 : (EPI == W_EPI_MASKY || EPI == W_EPI_GY) ? 63 : 0;
}

// ------------------------------------------------------------------------------------------
// Input gradient: out[R][col] = y * LeakyReLU'(stage input), the sign of the input from its 1-bit array (auxbits).
// ------------------------------------------------------------------------------------------
struct v6_pre_mask {
  uint32_t s[3];               // sign words of the half's rows: lane lr holds those of rows lr, 32 + lr, 64 + lr
};
__device__ __forceinline__ v6_pre_mask v6_prefetch_mask(const tl_nt_params& p, long long R0, int n0, int wm, int wn, int lr, int lh) {
  const int colbase = n0 + wn * 32;
  const long long ra = R0 + wm * 192 + 96 * lh + lr;
  v6_pre_mask r = {{0u, 0u, 0u}};
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (colbase < p.N && ra + 32 * k < p.M) r.s[k] = p.auxbits[(ra + 32 * k) * (long long)p.ld_auxbits + (colbase >> 5)];
  return r;
}
template <bool FULL>
__device__ __forceinline__ void v6_epilogue_mask(const tl_nt_params& p, const f32x16 (&acc)[8], const v6_pre_mask& pre, long long R0,
                                                 int n0, int wm, int wn, int lr_in, int lh) {
  int lr = lr_in;                                           // (opaque copy: see v6_epilogue_c1w)
  asm volatile("" : "+v"(lr));
  const int colbase = n0 + wn * 32;
  const bool colok = colbase < p.N;
  const int col = colbase + lr;
  const long long Rw = R0 + wm * 192;                       // first row of the wave; a half covers 96 rows
  const __amdgpu_buffer_rsrc_t rsO = rsrc_of(p.out + Rw * (long long)p.ldo, (p.M - Rw) * (long long)p.ldo * 4);
  const unsigned ovoff = colok ? ((unsigned)(96 * lh) * (unsigned)p.ldo + (unsigned)col) * 4u : V5_OOB;
  const unsigned ldo4 = (unsigned)p.ldo * 4u;
  bits96 inA = {~0ull, ~0u}, inB = inA;
  if (!FULL) {
    inA = v6_in_bits96(Rw, p.M);
    inB = v6_in_bits96(Rw + 96, p.M);
  }
  uint32_t sT[3];                                           // bit j of sT[k]: sign of (row 32 k + j, this lane's column)
#pragma unroll
  for (int k = 0; k < 3; ++k) sT[k] = bit_transpose32(pre.s[k], lr);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float y[6];
    wino63_rows(acc, e, y);
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      const int r = 6 * e + h;
      const float o = y[h] * selbit(sT[r >> 5], r & 31, 1.f, p.slope);
      const unsigned vo = FULL ? ovoff : selmu(mask96(inA, inB, r), ovoff, V5_OOB);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsO, vo, (unsigned)r * ldo4, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Plain rows: out[R][col] = LeakyReLU(y + bias) - the 7-tap convolutions of the CNN-RNN classifier (tl_conv7_wino63v_nt,
// reference models/deep_classifiers.py:250-256), whose three 3-tap segments the K loop has summed in the transform domain.
// ------------------------------------------------------------------------------------------
template <bool FULL>
__device__ __forceinline__ void v6_epilogue_lrelu(const tl_nt_params& p, const f32x16 (&acc)[8], const v5_pre_pool& pre, long long R0,
                                                  int n0, int wm, int wn, int lr_in, int lh) {
  int lr = lr_in;                                           // (opaque copy: see v6_epilogue_c1w)
  asm volatile("" : "+v"(lr));
  const int colbase = n0 + wn * 32;
  const bool colok = colbase < p.N;
  const int col = colbase + lr;
  const long long Rw = R0 + wm * 192;                       // first row of the wave; a half covers 96 rows
  const __amdgpu_buffer_rsrc_t rsO = rsrc_of(p.out + Rw * (long long)p.ldo, (p.M - Rw) * (long long)p.ldo * 4);
  const unsigned ovoff = colok ? ((unsigned)(96 * lh) * (unsigned)p.ldo + (unsigned)col) * 4u : V5_OOB;
  const unsigned ldo4 = (unsigned)p.ldo * 4u;
  bits96 inA = {~0ull, ~0u}, inB = inA;
  if (!FULL) {
    inA = v6_in_bits96(Rw, p.M);
    inB = v6_in_bits96(Rw + 96, p.M);
  }
  const float bv = pre.bv;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float y[6];
    wino63_rows(acc, e, y);
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      const int r = 6 * e + h;
      const float o = lrelu01(y[h] + bv, p.slope);
      const unsigned vo = FULL ? ovoff : selmu(mask96(inA, inB, r), ovoff, V5_OOB);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsO, vo, (unsigned)r * ldo4, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Input gradient that hands the stage BELOW its weight-gradient and input-gradient operands (epilogue 6, MASKY).  The rows
// this GEMM produces are the pooled gradient rows G of the stage below (conv3's input gradient = conv2's pooled output
// gradient); a lane owns 96 consecutive ones of a column = 32 hexes of that stage (three pooled rows = six conv rows each).
// Instead of G it writes, per hex and in the pair layout,
//   Y  = A dz   (eight planes: the second operand of the stage's weight gradient, tl_conv3_wino63v_tn with loader 3)
//   Vd = B^T (dz rows 6 h - 2 .. 6 h + 5)  (the operand of its input gradient)
// where dz is G un-pooled with the stage's arg-max bits (abits) and zeroed past the valid time (Tvalid_in conv rows).
// The weight-gradient kernel of that stage then runs without a transform (the Y side costs it 8 of 45 ms at conv2 and
// every C_in-tile workgroup repeats it) and G itself is never stored.  The first hex of a lane takes its front row from
// the half-wave below (through xch, one raw barrier), the first hex of a tile from the tile in front: it is stored raw
// (rows in slots 2..7) and finished by tl_wino63_vd_fixup from vhalo[tile - 1] = that tile's last pooled row, un-pooled.
// ------------------------------------------------------------------------------------------
struct v6_pre_masky {
  uint32_t s[3], a[3];         // sign words (auxbits) and arg-max words (abits) of the half's rows lr, 32 + lr, 64 + lr
};
// GTP: the bit arrays keep p.out_tp rows per sequence, [seq * out_tp + t], where the rows of this GEMM count p.Tp (the one-tap
// stage behind a POOL epilogue with out_tp: epilogue 7); rows t >= out_tp have no words (and are never valid)
template <bool GTP = false>
__device__ __forceinline__ v6_pre_masky v6_prefetch_masky(const tl_nt_params& p, long long R0, int n0, int wm, int wn, int lr, int lh) {
  const int colbase = n0 + wn * 32;
  const long long ra = R0 + wm * 192 + 96 * lh + lr;
  v6_pre_masky r;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    r.s[k] = r.a[k] = 0u;
    if (colbase < p.N && ra + 32 * k < p.M) {
      long long row = ra + 32 * k;
      bool have = true;
      if constexpr (GTP) {
        const unsigned R = (unsigned)row, sq = R / (unsigned)p.Tp, t = R - sq * (unsigned)p.Tp;       // host-checked: M < 2^31
        have = t < (unsigned)p.out_tp;
        row = (long long)sq * p.out_tp + t;
      }
      if (have) {
        r.s[k] = p.auxbits[row * (long long)p.ld_auxbits + (colbase >> 5)];
        r.a[k] = p.abits[row * (long long)p.ld_abits + (colbase >> 5)];
      }
    }
  }
  return r;
}
// DIRECT (epilogue 7): the accumulators ARE the rows - batch i < 6 of the kernel's eight GEMMs is row 6 H + i of hex H (a
// one-tap stage: tl_conv1_wino63v_dgrad_nt) - where the 3-tap form takes them from the inverse transform.
template <bool FULL, bool DIRECT = false>
__device__ __forceinline__ void v6_epilogue_masky(const tl_nt_params& p, const f32x16 (&acc)[8], const v6_pre_masky& pre, float* xch,
                                                  long long R0, int n0, int wm, int wn, int lr_in, int lh, long long tm,
                                                  unsigned long long* st = nullptr) {
  int lr = lr_in;                                           // (opaque copy: see v6_epilogue_c1w)
  asm volatile("" : "+v"(lr));
  const int colbase = n0 + wn * 32;
  const bool colok = colbase < p.N;
  const int col = colbase + lr;
  const long long Rw = R0 + wm * 192;                       // first row of the wave; a half covers 96 rows = 32 hexes below
  const int Tp = p.Tp, hps = Tp / 3;                        // rows / hexes of the stage below per sequence
  const v6_rows rw = v6_rows_of(tm * 128 + wm * 32, Tp);   // (time index of row Rw / Rw + 96)
  // Everything per row / per hex is a bit of a word this LANE holds (its half's copy), tested with v_bfe_i32: no lane masks
  // in scalar registers (this epilogue has 96 rows x 3 flags and 32 hexes x 2 flags per lane; they spilled).
  //   okw: rows whose gradient counts (pooled time below Tvalid_in / 2, inside the matrix)
  //   we / wo: ok and the arg-max bit clear / set - the masks of the un-pool select
  //   firstw: hexes that start their sequence (the row in front belongs to the sequence before: zero)
  uint32_t we[3], wo[3], sT[3], firstw = 0;
  {
    const bits96 okA = v6_valid_bits96(rw.tA, Tp, p.Tvalid_in >> 1, Rw, p.M);
    const bits96 okB = v6_valid_bits96(rw.tB, Tp, p.Tvalid_in >> 1, Rw + 96, p.M);
    const uint32_t okw[3] = {lh ? (uint32_t)okB.lo : (uint32_t)okA.lo, lh ? (uint32_t)(okB.lo >> 32) : (uint32_t)(okA.lo >> 32),
                             lh ? okB.hi : okA.hi};
    uint32_t fA = 0, fB = 0;
    int ha = rw.tA / 3, hb = rw.tB / 3;
    for (int k = 0; k < 32; ++k) {
      fA |= (uint32_t)(ha == 0) << k;
      fB |= (uint32_t)(hb == 0) << k;
      ha = ha + 1 == hps ? 0 : ha + 1;
      hb = hb + 1 == hps ? 0 : hb + 1;
    }
    firstw = lh ? fB : fA;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      sT[k] = bit_transpose32(pre.s[k], lr);
      const uint32_t aT = bit_transpose32(pre.a[k], lr);
      we[k] = okw[k] & ~aT;
      wo[k] = okw[k] & aT;
    }
  }
  const long long Hb = Rw / 3;                              // first hex (of the stage below) of the wave: 64 per wave
  const unsigned pair4 = (unsigned)p.ld_vout * 64u;        // bytes per hex pair (pair layout, tonal_wino63.hip)
  const __amdgpu_buffer_rsrc_t rsY = rsrc_of(p.vout + Hb * 8 * (long long)p.ld_vout, (p.vout_quads - Hb) * 8 * (long long)p.ld_vout * 4);
  const __amdgpu_buffer_rsrc_t rsD = rsrc_of(p.vout2 + Hb * 8 * (long long)p.ld_vout, (p.vout_quads - Hb) * 8 * (long long)p.ld_vout * 4);
  // (a dropped store carries V6_DROP: the immediates added below must not wrap it back into the resource)
  // after v6_pair_swap: lanes 0-31 / 32-63 = hex j / j + 1 of a pair; first the pairs of lane-half 0's 32 hexes, then half 1's
  const unsigned offP = v6_pair_offset(colbase, lr, lh);
  const unsigned vvA = colok ? offP : V6_DROP, vvB = colok ? 16u * pair4 + offP : V6_DROP;
  const int nexA = (int)(p.M / 3 - Hb < 64 ? p.M / 3 - Hb : 64) - lh, nexB = nexA - 32;      // (hex j + lh of the pair exists: j < nex)
  float pe = 0.f, po = 0.f;                                 // the pooled row in front of the current hex, un-pooled
  float first_d[6], v1keep[8];                              // hex 0 of the lane waits for the exchange (and hex 1, its pair, with it)
#pragma unroll
  for (int k = 0; k < 6; ++k) first_d[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) v1keep[k] = 0.f;
  auto andf = [](float x, uint32_t m) { return __uint_as_float(__float_as_uint(x) & m); };
  v6_stamp(st, 8);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (e == 1) v6_stamp(st, 9);
    if (e == 4) v6_stamp(st, 10);
    if (e == 8) v6_stamp(st, 11);
    if (e == 12) v6_stamp(st, 12);
    float y[6];
    if constexpr (DIRECT) {
#pragma unroll
      for (int h = 0; h < 6; ++h) y[h] = acc[h][e];
    } else {
      wino63_rows(acc, e, y);
    }
    float dzr[12];                                          // the six pooled rows of accumulator element e, un-pooled
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      const int r = 6 * e + h;
      const float t = y[h] * selbit(sT[r >> 5], r & 31, 1.f, p.slope);
      dzr[2 * h] = andf(t, (uint32_t)__builtin_amdgcn_sbfe((int)we[r >> 5], r & 31, 1));
      dzr[2 * h + 1] = andf(t, (uint32_t)__builtin_amdgcn_sbfe((int)wo[r >> 5], r & 31, 1));
    }
    float Yp[2][8], Vp[2][8];                               // the pair of hexes (of the stage below) of accumulator element e
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int j = 2 * e + hh;
      const float* d6 = dzr + 6 * hh;
      // Y = A dz
      const float ev1 = (d6[0] + d6[2]) + d6[4], od1 = (d6[1] + d6[3]) + d6[5];
      const float ev2 = fmaf(16.f, d6[4], fmaf(4.f, d6[2], d6[0])), od2 = fmaf(32.f, d6[5], fmaf(8.f, d6[3], 2.f * d6[1]));
      const float ev3 = fmaf(0.0625f, d6[4], fmaf(0.25f, d6[2], d6[0])), od3 = fmaf(0.03125f, d6[5], fmaf(0.125f, d6[3], 0.5f * d6[1]));
      Yp[hh][0] = d6[0];
      Yp[hh][1] = ev1 + od1;
      Yp[hh][2] = ev1 - od1;
      Yp[hh][3] = ev2 + od2;
      Yp[hh][4] = ev2 - od2;
      Yp[hh][5] = ev3 + od3;
      Yp[hh][6] = ev3 - od3;
      Yp[hh][7] = d6[5];
      if (j == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) first_d[k] = d6[k];
      } else {
        const uint32_t keep = ~(uint32_t)__builtin_amdgcn_sbfe((int)firstw, j, 1);      // all ones unless the hex starts its sequence
        float d[8];
        d[0] = andf(pe, keep);
        d[1] = andf(po, keep);
#pragma unroll
        for (int k = 0; k < 6; ++k) d[2 + k] = d6[k];
        wino63_bt(d, Vp[hh]);
      }
      pe = d6[4];
      po = d6[5];
    }
    {
      const int j = 2 * e;
      // (scalar offset: the hex pair; the transform rides in the instruction's immediate where no lane is dropped)
      const unsigned so = (unsigned)e * pair4;
      const unsigned voA = FULL ? vvA : (j < nexA ? vvA : V6_DROP), voB = FULL ? vvB : (j < nexB ? vvB : V6_DROP);
      v6_store_hex_pair<FULL>(Yp[0], Yp[1], rsY, voA, voA + 1024u, voB, voB + 1024u, so);
      if (e == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v1keep[i] = Vp[1][i];
      } else {
        v6_store_hex_pair<FULL>(Vp[0], Vp[1], rsD, voA, voA + 1024u, voB, voB + 1024u, so);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  v6_stamp(st, 13);
  // ---- hex 0 of every lane: its front row is the last row of the half-wave below ----
  const int slot = wm * 2 + lh;
  {
    float2 v2 = {pe, po};
    *reinterpret_cast<float2*>(xch + (slot * 64 + wn * 32 + lr) * 2) = v2;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0)  (not __syncthreads(): see v6_epilogue_pool)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    float fe = 0.f, fo = 0.f;
    if (slot > 0) {
      const float2 v2 = *reinterpret_cast<const float2*>(xch + ((slot - 1) * 64 + wn * 32 + lr) * 2);
      fe = v2.x;
      fo = v2.y;
    }
    const uint32_t keep = ~(uint32_t)__builtin_amdgcn_sbfe((int)firstw, 0, 1);
    float d[8], v[8];
    d[0] = andf(fe, keep);
    d[1] = andf(fo, keep);
#pragma unroll
    for (int k = 0; k < 6; ++k) d[2 + k] = first_d[k];
    wino63_bt(d, v);
    // (the tile's very first hex: rows raw in slots 2..7 for tl_wino63_vd_fixup, which owns its front row)
    const bool raw = slot == 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = raw ? (k < 2 ? 0.f : d[k]) : v[k];
    const unsigned voA = FULL ? vvA : (0 < nexA ? vvA : V6_DROP), voB = FULL ? vvB : (0 < nexB ? vvB : V6_DROP);
    v6_store_hex_pair<FULL>(v, v1keep, rsD, voA, voA + 1024u, voB, voB + 1024u, 0u);
  }
  // the tile's last pooled row, un-pooled: the front row of the next tile's first hex
  {
    const bool hw = wm == 3 && p.vhalo != nullptr;
    const __amdgpu_buffer_rsrc_t rsH = rsrc_of(hw ? p.vhalo + tm * 2 * (long long)p.N : nullptr, hw ? 2LL * p.N * 4 : 0);
    const unsigned ho = (colok && lh == 1) ? (unsigned)col * 4u : V5_OOB;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pe), rsH, ho, 0u, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, po), rsH, ho, (unsigned)p.N * 4u, 0);
  }
}

// ------------------------------------------------------------------------------------------
// Input gradient of stage 2 with the fused first-stage weight gradient (tonal_wino43v_epi.h, v5_epilogue_c1w): G1 = y *
// LeakyReLU'(sign bit) is contracted on the spot with the raw signal, dW1[o][j] = sum_rows G1[row][o] x[seq][2 t + a + j].
// Prefetch: lane lr of a half requests the two bit words and the sample window x[seq][2 t .. 2 t + 3] of its rows lr,
// 32 + lr, 64 + lr.  Body: two passes of 48 rows; the windows of a pass sit in a wave-private LDS table [half][48][4] and
// come back per row as one ds_read_b128 at a half-wave-uniform address.  xw: 1.5 KB of LDS per wave that nothing else uses.
// ------------------------------------------------------------------------------------------
struct v6_pre_c1w {
  uint32_t s[3], c[3];
  f32x4 x[2];                  // samples x[seq][2 t .. 2 t + 7] of the lane's rows 3 lr, 3 lr + 1, 3 lr + 2 (windows of four, two apart)
};
__device__ __forceinline__ v6_pre_c1w v6_prefetch_c1w(const tl_nt_params& p, long long R0, int n0, int wm, int wn, int lr, int lh) {
  const int colbase = n0 + wn * 32;
  const long long rh = R0 + wm * 192 + 96 * lh;             // first row of the half
  v6_pre_c1w r;
  const int Tp = p.Tp;
  const long long nseq = p.M / Tp;
  const unsigned s0 = (unsigned)(R0 / Tp);                   // first sequence of the tile (the resource starts there)
  const __amdgpu_buffer_rsrc_t rsX = rsrc_of(p.c1x + (long long)s0 * p.c1T, (nseq - s0) * (long long)p.c1T * 4);
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    r.s[k] = r.c[k] = 0u;
    const long long rk = rh + lr + 32 * k;
    if (colbase < p.N && rk < p.M) {
      r.s[k] = p.auxbits[rk * (long long)p.ld_auxbits + (colbase >> 5)];
      r.c[k] = p.c1bits[rk * (long long)p.ld_auxbits + (colbase >> 5)];
    }
  }
  // three rows = half a hex: inside one sequence (Tp % 6 == 0).  Rows past the valid time / the matrix read whatever the
  // resource still covers or zeros: their dz is 0
  const unsigned Rx = (unsigned)(rh + 3 * lr), sx = Rx / (unsigned)Tp, tx = Rx - sx * (unsigned)Tp;     // M < 2^31 (host-checked)
  const unsigned xo = ((sx - s0) * (unsigned)p.c1T + 2u * tx) * 4u;
  r.x[0] = __builtin_bit_cast(f32x4, (v4u)__builtin_amdgcn_raw_buffer_load_b128(rsX, xo, 0u, 0));
  r.x[1] = __builtin_bit_cast(f32x4, (v4u)__builtin_amdgcn_raw_buffer_load_b128(rsX, xo + 16u, 0u, 0));
  return r;
}
__device__ __forceinline__ void v6_epilogue_c1w(const tl_nt_params& p, const f32x16 (&acc)[8], const v6_pre_c1w& pre, float* xw,
                                                float* red, long long R0, int n0, int wm, int wn, int lr_in, int lh, long long tm) {
  // (an opaque copy of the lane index: what the bit transposes derive from it is then computed here, per tile, instead of
  // being hoisted out of the tile loop into registers that do not exist - they came back as scratch reloads)
  int lr = lr_in;
  asm volatile("" : "+v"(lr));
  const int colbase = n0 + wn * 32;
  const bool colok = colbase < p.N;
  const int col = colbase + lr;
  const long long Hw = tm * 128 + wm * 32;
  const long long Rw = 6 * Hw;
  const int Tp = p.Tp;
  const v6_rows rw = v6_rows_of(Hw, Tp);
  // rows that count: time below Tvalid, inside the matrix
  const bits96 okA = v6_valid_bits96(rw.tA, Tp, p.Tvalid, Rw, p.M);
  const bits96 okB = v6_valid_bits96(rw.tB, Tp, p.Tvalid, Rw + 96, p.M);
  float* xh = xw + lh * 256;                                 // this half's sample table: rows 3 l .. 3 l + 2 at xh + 8 l
  *reinterpret_cast<f32x4*>(xh + 8 * lr) = pre.x[0];
  *reinterpret_cast<f32x4*>(xh + 8 * lr + 4) = pre.x[1];
  c1w_acc ca;
  ca.clear();
  uint32_t sT[3], cT[3];                                    // bit j: sign / conv1 arg-max of (row 32 k + j, this lane's column)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    sT[k] = bit_transpose32(pre.s[k], lr);
    cT[k] = bit_transpose32(pre.c[k], lr);
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float y[6];
    wino63_rows(acc, e, y);
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      const int r = 6 * e + h;
      // window of row r: samples 2 (r % 3) .. + 3 of the eight of its row triple (half-wave-uniform address: a broadcast)
      const float2 xa = *reinterpret_cast<const float2*>(xh + 8 * (r / 3) + 2 * (r % 3));
      const float2 xb = *reinterpret_cast<const float2*>(xh + 8 * (r / 3) + 2 * (r % 3) + 2);
      const float dz = selm0(mask96(okA, okB, r), y[h] * selbit(sT[r >> 5], r & 31, 1.f, p.slope));
      const uint32_t am = (uint32_t)__builtin_amdgcn_sbfe((int)cT[r >> 5], r & 31, 1);
      auto pick = [am](float a, float b) { return __uint_as_float((__float_as_uint(a) & am) | (__float_as_uint(b) & ~am)); };
      ca.s[0] = fmaf(dz, pick(xa.y, xa.x), ca.s[0]);
      ca.s[1] = fmaf(dz, pick(xb.x, xa.y), ca.s[1]);
      ca.s[2] = fmaf(dz, pick(xb.y, xb.x), ca.s[2]);
      ca.b += dz;
    }
    __builtin_amdgcn_sched_barrier(0);                       // (keeps the loads of later hexes from being hoisted: they spill)
  }
  __syncthreads();                                          // `red` is a K-loop stage: every wave past its last fragment read
  c1w_reduce_store<4, 64>(p, red, ca, wm, wn * 32 + lr, lh, tm, col, colok);
}

}  // namespace tl

// Epilogues of the V-form F(4,3) NT kernel (tonal_wino43v.hip), round 4: the same arithmetic as tonal_wino43_epi.h
// (reference: models/synthesis_models.py:87-97 and their backward) re-laid so that everything that depends on the ROW
// of an element is half-wave uniform and runs on the scalar ALU.
//
// The kernel hands the matrix pipe the quads of a wave in a permuted order (the LDS row of a quad is chosen by the
// LDS-DMA source address, so the permutation costs nothing): MFMA row 8 g + 4 lh + j holds quad 16 lh + 4 g + j of the
// wave's 32.  A lane (lr, lh) then owns ONE column and SIXTEEN CONSECUTIVE quads, accumulator element e = quad 16 lh + e:
// 64 consecutive conv rows, 32 consecutive pooled rows, 8 next-stage quads.  Consequences:
//   * time index, row validity, sequence ends are two scalars per wave (one per half), not per-lane cursors;
//   * a 1-bit-per-element word of a row (arg-max / sign arrays: one 32-bit word per row and 32 columns) IS the lane mask
//     of that row for one half-wave: two v_readlane build the 64-bit mask of a v_cndmask - no shuffles (the round-3
//     epilogues issued 64 - 388 ds_bpermute per tile), and a ballot goes back to memory through two v_writelane;
//   * stores are buffer stores on a per-wave resource: scalar row offset, one per-lane offset for the whole tile, rows /
//     columns outside the matrix dropped by an out-of-range offset instead of exec-masked branches;
//   * the pooled rows a next-stage quad needs (4 Q' .. 4 Q' + 5) sit in one lane for 7 quads of 8, so the forward pass
//     of a stage can write V = B^T d of its OUTPUT directly (POOLV): the raw pooled rows of conv2 are not stored and the
//     stand-alone transform kernel disappears.  The 8th quad takes two rows from the next half-wave (through LDS); the
//     last quad of a tile takes them from the next tile: it is stored raw and finished by tl_wino43_v_fixup.
#pragma once
#include "tonal_common.h"
#include "tonal_wino43_epi.h"
#include <type_traits>

namespace tl {

// compile-time loop: the body sees its index as a constant expression (lane numbers of v_writelane are immediates)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// Transpose of a 32 x 32 bit matrix held one word per lane in each half-wave: in, lane c holds bit j = M[j][c]; out, lane j
// holds bit c = M[j][c].  Five butterfly stages (lane ^ 16, 8, 4, 2, 1 through ds_swizzle; rotate + bit-field insert).
// This is how a lane's 32 arg-max / sign bits (one per pooled row of ITS column, shifted in as they are produced) become
// the row words of the bit arrays.  (First tried: every ballot -> v_writelane.  128 live scalar results per tile spill
// through v_readlane / v_writelane, and a v_writelane in inline asm whose scalar source was written by a VECTOR
// instruction reads stale values in some waves on gfx950 - the hazard recognizer does not look inside inline asm.)
__device__ __forceinline__ uint32_t bit_transpose32(uint32_t w, int lr) {
#define TL_BT_STAGE(K, M)                                                                        \
  {                                                                                              \
    const uint32_t pw = (uint32_t)__builtin_amdgcn_ds_swizzle((int)w, 0x1f | ((K) << 10));        \
    const bool hi = (lr & (K)) != 0;                                                              \
    const uint32_t r = __builtin_amdgcn_alignbit(pw, pw, hi ? (K) : 32 - (K));                    \
    const uint32_t mm = hi ? ~(uint32_t)(M) : (uint32_t)(M);                                      \
    w = (w & mm) | (r & ~mm);                                                                     \
  }
  TL_BT_STAGE(16, 0x0000ffffu)
  TL_BT_STAGE(8, 0x00ff00ffu)
  TL_BT_STAGE(4, 0x0f0f0f0fu)
  TL_BT_STAGE(2, 0x33333333u)
  TL_BT_STAGE(1, 0x55555555u)
#undef TL_BT_STAGE
  return w;
}
// lane's bit of m set ? a : b  /  ? a : 0   (v_cndmask on an SGPR pair, written so that the compiler sees it: it schedules
// around the masks' producers - v_readlane, v_cmp - and inserts the wait states those need)
__device__ __forceinline__ float selm(unsigned long long m, float a, float b) { return __builtin_amdgcn_inverse_ballot_w64(m) ? a : b; }
__device__ __forceinline__ float selm0(unsigned long long m, float a) { return __builtin_amdgcn_inverse_ballot_w64(m) ? a : 0.f; }
__device__ __forceinline__ unsigned selmu(unsigned long long m, unsigned a, unsigned b) {
  return __builtin_amdgcn_inverse_ballot_w64(m) ? a : b;
}
// 64-bit lane mask of a half-wave-uniform condition: bit j of wa for lanes 0-31, of wb for lanes 32-63
__device__ __forceinline__ unsigned long long mask2(uint32_t wa, uint32_t wb, int j) {
  const uint32_t lo = (uint32_t)(-(int)((wa >> j) & 1u)), hi = (uint32_t)(-(int)((wb >> j) & 1u));
  return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
__device__ __forceinline__ unsigned long long mask2l(unsigned long long wa, unsigned long long wb, int j) {
  const uint32_t lo = (uint32_t)(-(int)((wa >> j) & 1ull)), hi = (uint32_t)(-(int)((wb >> j) & 1ull));
  return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
// the word lane `l` of each half-wave holds, as the 64-bit lane mask {half 0: word of lane l, half 1: word of lane 32 + l}
__device__ __forceinline__ unsigned long long words_as_mask(uint32_t w, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)w, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)w, 32 + l);
  return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
__device__ __forceinline__ int clip31(long long v) { return (int)(v < 0 ? 0 : (v < 0x7fffffffLL ? v : 0x7fffffffLL)); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, clip31(bytes), 0x00020000);
}
// LeakyReLU for 0 <= slope <= 1 (host-checked): max(z, slope z) - the same value as z > 0 ? z : slope z, one instruction less
__device__ __forceinline__ float lrelu01(float z, float slope) { return fmaxf(z, z * slope); }
constexpr unsigned V5_OOB = 0xfffffff0u;      // a per-lane offset no resource covers: the access is dropped / reads 0

// the four conv rows of a quad from its six Winograd products (expressions of tonal_wino43_epi.h)
__device__ __forceinline__ void wino43_rows(const f32x16 (&acc)[6], int e, float (&y)[4]) {
  const float m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e];
  const float a12 = m1 + m2, s12 = m1 - m2, a34 = m3 + m4, s34 = m3 - m4;
  y[0] = (acc[0][e] + a12) + a34;
  y[1] = s12 + 2.f * s34;
  y[2] = a12 + 4.f * a34;
  y[3] = (s12 + 8.f * s34) + acc[5][e];
}

// Row bookkeeping of one wave tile.  Qw: first quad of the wave; rows R = 4 Qw + 64 lh + r, r < 64.
struct v5_rows {
  int tA, tB;                 // time index of the first conv row of half 0 / half 1
  long long seqA, seqB;       // their sequences
};
__device__ __forceinline__ v5_rows v5_rows_of(long long Qw, int Tp) {
  v5_rows r;
  const unsigned R = (unsigned)(4 * Qw);                    // host-checked: M < 2^31
  const unsigned s = R / (unsigned)Tp;                      // (integer division runs on the vector ALU: once per tile)
  r.seqA = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)s);
  r.tA = __builtin_amdgcn_readfirstlane((int)(R - s * (unsigned)Tp));
  r.tB = r.tA + 64;
  r.seqB = r.seqA;
  while (r.tB >= Tp) {
    r.tB -= Tp;
    ++r.seqB;
  }
  return r;
}
// bit k (k < N): row k * STEP of a half - time (t0 + k * STEP) mod Tp - lies below `tlim` and inside the matrix (row index
// R0h + k * STEP < M).  Closed form per sequence segment (a span of 64 rows crosses few sequence ends), scalar ALU only:
// the eight waves of a workgroup share one scalar unit, a loop over the rows would cost microseconds per tile.
template <int N, int STEP>
__device__ __forceinline__ unsigned long long v5_in_bits(long long R0h, long long M) {
  long long nin = (M - R0h + STEP - 1) / STEP;
  nin = nin < 0 ? 0 : (nin > N ? N : nin);
  return nin >= 64 ? ~0ull : ((1ull << nin) - 1ull);
}
template <int N, int STEP>
__device__ __forceinline__ unsigned long long v5_valid_bits(int t0, int Tp, int tlim, long long R0h, long long M) {
  unsigned long long w = 0;
  int k = 0, t = t0;
  while (k < N) {
    int nseg = (Tp - t) / STEP;                              // rows up to the end of this sequence (t0, Tp multiples of 4)
    nseg = nseg < N - k ? nseg : N - k;
    int nval = t < tlim ? (tlim - t + STEP - 1) / STEP : 0;
    nval = nval < nseg ? nval : nseg;
    w |= (nval >= 64 ? ~0ull : ((1ull << nval) - 1ull)) << k;
    k += nseg;
    t = 0;
  }
  return w & v5_in_bits<N, STEP>(R0h, M);
}

// ------------------------------------------------------------------------------------------
// Forward: bias + LeakyReLU + max-pool (2,1) + arg-max / sign bits.  VOUT: also (or only) V of the pooled output for the
// next stage.  xch: 8 x 64 x 2 floats of LDS that nothing else uses; the function holds ONE workgroup barrier when VOUT
// (every wave of the workgroup must call it).
// ------------------------------------------------------------------------------------------
// What an epilogue reads from global memory is requested by v5_prefetch_* BEFORE the kernel issues the next tile's first
// LDS-DMA pieces: vmcnt counts in issue order, so a load issued behind those pieces would make its consumer wait for
// them as well - one HBM round trip per tile with no MFMA in flight (the round-3 epilogues started with such a wait).
struct v5_pre_pool {
  float bv;
};
__device__ __forceinline__ v5_pre_pool v5_prefetch_pool(const tl_nt_params& p, int n0, int wn, int lr) {
  const int colbase = n0 + wn * 32;
  v5_pre_pool r;
  r.bv = (colbase < p.N && p.bias) ? p.bias[colbase + lr] : 0.f;
  return r;
}
template <bool VOUT, bool FULL>
__device__ __forceinline__ void v5_epilogue_pool(const tl_nt_params& p, const f32x16 (&acc)[6], const v5_pre_pool& pre, float* xch,
                                                 long long R0, int n0, int wm, int wn, int lr, int lh, long long tm) {
  const int colbase = n0 + wn * 32;
  const bool colok = colbase < p.N;                         // N % 32 == 0 (host-checked): the wave's columns are in or out together
  const int col = colbase + lr;
  const long long Qw = (R0 >> 2) + wm * 32;
  const int Tp = p.Tp;
  const v5_rows rw = v5_rows_of(Qw, Tp);
  // pooled row j of a half (conv rows 2 j, 2 j + 1 of its 64) is an output row
  const uint32_t vA = (uint32_t)v5_valid_bits<32, 2>(rw.tA, Tp, p.Tvalid, 4 * Qw, p.M);
  const uint32_t vB = (uint32_t)v5_valid_bits<32, 2>(rw.tB, Tp, p.Tvalid, 4 * Qw + 64, p.M);
  const float bv = pre.bv;
  const long long P0 = 2 * Qw;                              // first pooled row of the wave
  const long long prows = p.M >> 1;
  const unsigned lane_row = (unsigned)(32 * lh);
  // pooled output rows [P0, P0 + 64) x ldo; rows past the matrix are never stored (valid bit clear -> offset out of range)
  const bool want_p = !VOUT || p.out != nullptr;
  const __amdgpu_buffer_rsrc_t rsO = rsrc_of(want_p ? p.out + P0 * (long long)p.ldo : nullptr,
                                            want_p ? (prows - P0) * (long long)p.ldo * 4 : 0);
  const unsigned ovoff = colok ? (lane_row * (unsigned)p.ldo + (unsigned)col) * 4u : V5_OOB;
  const unsigned ldo4 = (unsigned)p.ldo * 4u;
  // rows of the matrix (not: valid time) - the old epilogue stores zeros into the pad rows of a sequence
  // (FULL: an interior tile - the masks are constants and the per-store selects fold away)
  const uint32_t inA = FULL ? ~0u : (uint32_t)v5_in_bits<32, 2>(4 * Qw, p.M), inB = FULL ? ~0u : (uint32_t)v5_in_bits<32, 2>(4 * Qw + 64, p.M);
  float pv[32];
  uint32_t wbits = 0, wsign = 0;
  static_for<0, 16>([&](auto E) {
    constexpr int e = decltype(E)::value;
    float y[4];
    wino43_rows(acc, e, y);
    static_for<0, 2>([&](auto H) {
      constexpr int h = decltype(H)::value, j = 2 * e + h;
      const float y0 = lrelu01(y[2 * h] + bv, p.slope), y1 = lrelu01(y[2 * h + 1] + bv, p.slope);
      const bool valid = __builtin_amdgcn_inverse_ballot_w64(mask2(vA, vB, j));
      const bool gt = (y1 > y0) && valid;
      const float o = valid ? (gt ? y1 : y0) : 0.f;
      pv[j] = o;
      // the lane's own column: one bit per pooled row, row j in bit 31 - j until the reversal below
      wbits = wbits + wbits + (uint32_t)gt;                 // (shift in: v_addc with the compare as carry)
      wsign = wsign + wsign + (uint32_t)(o > 0.f);
      if constexpr (!VOUT) {
        const unsigned vo = selmu(mask2(inA, inB, j), ovoff, V5_OOB);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsO, vo, (unsigned)j * ldo4, 0);
      }
    });
  });
  {
    // bit words: lane lr of a half holds those of pooled row P0 + 32 lh + lr.  Buffer stores like the rest (a lane with
    // nothing to write carries an out-of-range offset): the number of stores a wave issues per tile is a constant, which
    // is what lets the kernel wait for its next tile's operands WITHOUT waiting for these stores (V5_STORES below)
    wbits = bit_transpose32(__builtin_bitreverse32(wbits), lr);
    wsign = bit_transpose32(__builtin_bitreverse32(wsign), lr);
    const long long P = P0 + 32 * lh + lr;
    const unsigned wo = (colok && 2 * P < p.M) ? (unsigned)((32 * lh + lr) * p.ld_obits) * 4u : V5_OOB;
    const long long wbase = P0 * (long long)p.ld_obits + (colbase >> 5), wleft = (prows - P0) * (long long)p.ld_obits * 4;
    const __amdgpu_buffer_rsrc_t rsB = rsrc_of(p.obits + wbase, wleft);
    const __amdgpu_buffer_rsrc_t rsS = rsrc_of(p.osign ? p.osign + wbase : nullptr, p.osign ? wleft : 0);
    __builtin_amdgcn_raw_buffer_store_b32(wbits, rsB, wo, 0u, 0);
    __builtin_amdgcn_raw_buffer_store_b32(wsign, rsS, wo, 0u, 0);
  }
  if constexpr (VOUT) {
    if (p.out != nullptr) {                                 // (tests, TONAL_STORE_P2: the raw pooled rows as well)
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const unsigned vo = selmu(mask2(inA, inB, j), ovoff, V5_OOB);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv[j]), rsO, vo, (unsigned)j * ldo4, 0);
      }
    }
    // ---- V of the pooled output: next-stage quad Q' of a half = its pooled rows 4 Q' .. 4 Q' + 5 ----
    const int Tq = Tp >> 1;                                 // pooled rows per sequence (host-checked: Tp % 8 == 0)
    const int slot = wm * 2 + lh;
    {
      float2 v2 = {pv[0], pv[1]};
      *reinterpret_cast<float2*>(xch + (slot * 64 + wn * 32 + lr) * 2) = v2;
    }
    // (not __syncthreads(): its fence would also wait for the vector-memory operations in flight - the next tile's LDS-DMA)
    __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float nb0 = 0.f, nb1 = 0.f;
    if (slot < 7) {                                         // (the last half-wave of a tile: finished by the fix-up pass)
      const float2 v2 = *reinterpret_cast<const float2*>(xch + ((slot + 1) * 64 + wn * 32 + lr) * 2);
      nb0 = v2.x;
      nb1 = v2.y;
    }
    // quad Q' ends its sequence (rows 4, 5 belong to the next one: zero): pooled time of its first row == Tq - 4
    uint32_t seA = 0, seB = 0, nvA = 0, nvB = 0;
    {
      int ta = rw.tA >> 1, tb = rw.tB >> 1;
      const long long qa = (Qw >> 1), qb = (Qw >> 1) + 8;   // next-stage quad index of Q' = 0
      for (int k = 0; k < 8; ++k) {
        seA |= (uint32_t)(ta == Tq - 4) << k;
        seB |= (uint32_t)(tb == Tq - 4) << k;
        nvA |= (uint32_t)(FULL || qa + k < p.vout_quads) << k;
        nvB |= (uint32_t)(FULL || qb + k < p.vout_quads) << k;
        ta += 4;
        if (ta >= Tq) ta -= Tq;
        tb += 4;
        if (tb >= Tq) tb -= Tq;
      }
    }
    const long long Qn = Qw >> 1;                           // first next-stage quad of the wave (Qw % 32 == 0)
    const __amdgpu_buffer_rsrc_t rsV = rsrc_of(p.vout + Qn * 6 * (long long)p.ld_vout,
                                              (p.vout_quads - Qn) * 6 * (long long)p.ld_vout * 4);
    const unsigned vvoff = colok ? ((unsigned)(8 * lh * 6) * (unsigned)p.ld_vout + (unsigned)col) * 4u : V5_OOB;
    const unsigned ldv4 = (unsigned)p.ld_vout * 4u;
    const unsigned long long mraw = wm == 3 ? 0xffffffff00000000ull : 0ull;   // Q' = 7 of the tile's last half-wave
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned long long mend = mask2(seA, seB, q);
      const float d0 = pv[4 * q], d1 = pv[4 * q + 1], d2 = pv[4 * q + 2], d3 = pv[4 * q + 3];
      const float d4 = selm(mend, 0.f, q < 7 ? pv[(4 * q + 4) & 31] : nb0), d5 = selm(mend, 0.f, q < 7 ? pv[(4 * q + 5) & 31] : nb1);
      float v[6];
      // (the expressions of wino43_xform_kernel)
      const float s1 = d4 - 4.f * d2, s2 = d3 - 4.f * d1, s3 = d4 - d2, t = d3 - d1;
      v[0] = 4.f * d0 + (d4 - 5.f * d2);
      v[1] = s1 + s2;
      v[2] = s1 - s2;
      v[3] = s3 + 2.f * t;
      v[4] = s3 - 2.f * t;
      v[5] = (4.f * d1 - 5.f * d3) + d5;
      if (q == 7) {                                         // raw rows for tl_wino43_v_fixup (which owns rows 4, 5 of this quad)
        v[0] = selm(mraw, d0, v[0]);
        v[1] = selm(mraw, d1, v[1]);
        v[2] = selm(mraw, d2, v[2]);
        v[3] = selm(mraw, d3, v[3]);
      }
      const unsigned vo = selmu(mask2(nvA, nvB, q), vvoff, V5_OOB);
#pragma unroll
      for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i]), rsV, vo, (unsigned)(q * 6 + i) * ldv4, 0);
    }
    // the tile's first two pooled rows: rows 4, 5 of the last quad of the tile in front (tl_wino43_v_fixup)
    {
      const bool hw = wm == 0 && p.vhalo != nullptr;
      const __amdgpu_buffer_rsrc_t rsH = rsrc_of(hw ? p.vhalo + tm * 2 * (long long)p.N : nullptr, hw ? 2LL * p.N * 4 : 0);
      const unsigned ho = (colok && lh == 0) ? (unsigned)col * 4u : V5_OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv[0]), rsH, ho, 0u, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv[1]), rsH, ho, (unsigned)p.N * 4u, 0);
    }
  }
}

// Vector-memory stores every wave issues per tile in an epilogue, whatever the tile (lower bound where a branch adds
// some): the kernel's end-of-tile wait leaves that many operations in flight - the LDS-DMA pieces of the next tile's
// first K-step, issued BEFORE the epilogue, are older and have landed (vmcnt counts in issue order, 6 bits).
template <int EPI>
constexpr int v5_stores() {
  return EPI == W_EPI_POOL ? 34 : EPI == W_EPI_POOLV ? 52 : EPI == W_EPI_MASK ? 63 : 0;
}

// ------------------------------------------------------------------------------------------
// Input gradient: out[R][col] = y * LeakyReLU'(stage input), the sign of the input from its 1-bit array (auxbits).
// ------------------------------------------------------------------------------------------
struct v5_pre_mask {
  uint32_t sA, sB;             // sign words of the half's rows: lane lr holds those of rows lr and 32 + lr
};
__device__ __forceinline__ v5_pre_mask v5_prefetch_mask(const tl_nt_params& p, long long R0, int n0, int wm, int wn, int lr, int lh) {
  const int colbase = n0 + wn * 32;
  const long long ra = R0 + wm * 128 + 64 * lh + lr, rb = ra + 32;
  v5_pre_mask r = {0u, 0u};
  if (colbase < p.N && ra < p.M) r.sA = p.auxbits[ra * (long long)p.ld_auxbits + (colbase >> 5)];
  if (colbase < p.N && rb < p.M) r.sB = p.auxbits[rb * (long long)p.ld_auxbits + (colbase >> 5)];
  return r;
}
template <bool FULL>
__device__ __forceinline__ void v5_epilogue_mask(const tl_nt_params& p, const f32x16 (&acc)[6], const v5_pre_mask& pre, long long R0,
                                                 int n0, int wm, int wn, int lr, int lh) {
  const int colbase = n0 + wn * 32;
  const bool colok = colbase < p.N;
  const int col = colbase + lr;
  const long long Qw = (R0 >> 2) + wm * 32;
  const long long Rw = 4 * Qw;                              // first row of the wave; a half covers 64 rows
  const uint32_t sA = pre.sA, sB = pre.sB;
  const __amdgpu_buffer_rsrc_t rsO = rsrc_of(p.out + Rw * (long long)p.ldo, (p.M - Rw) * (long long)p.ldo * 4);
  const unsigned ovoff = colok ? ((unsigned)(64 * lh) * (unsigned)p.ldo + (unsigned)col) * 4u : V5_OOB;
  const unsigned ldo4 = (unsigned)p.ldo * 4u;
  const unsigned long long inA = FULL ? ~0ull : v5_in_bits<64, 1>(Rw, p.M), inB = FULL ? ~0ull : v5_in_bits<64, 1>(Rw + 64, p.M);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float y[4];
    wino43_rows(acc, e, y);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int r = 4 * e + h;
      const unsigned long long mpos = words_as_mask(r < 32 ? sA : sB, r & 31);
      const float o = selm(mpos, y[h], y[h] * p.slope);
      const unsigned vo = selmu(mask2l(inA, inB, r), ovoff, V5_OOB);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsO, vo, (unsigned)r * ldo4, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Input gradient of stage 2 with the fused first-stage weight gradient (tonal_wino43_epi.h, epilogue 4): G1 = y *
// LeakyReLU'(sign bit) is contracted on the spot with the raw signal, dW1[o][j] = sum_rows G1[row][o] x[seq][2 t + a + j].
// Prefetch: lane lr of a half requests the two bit words and the sample window x[seq][2 t .. 2 t + 3] of rows lr and
// 32 + lr of its half.  Body: the windows go to a wave-private LDS table [half][64 rows][4] and come back per row as one
// ds_read_b128 at a half-wave-uniform address (a broadcast; LDS reads do not queue behind the LDS-DMA of the next tile);
// the bit words become lane masks.  xw: 2 KB of LDS per wave that nothing else uses.
// red: LDS scratch of c1w_reduce_store (a K-loop stage, hence the barrier in front of it).
// ------------------------------------------------------------------------------------------
struct v5_pre_c1w {
  uint32_t sA, sB, cA, cB;
  f32x4 xA, xB;
};
__device__ __forceinline__ v5_pre_c1w v5_prefetch_c1w(const tl_nt_params& p, long long R0, int n0, int wm, int wn, int lr, int lh) {
  const int colbase = n0 + wn * 32;
  const long long ra = R0 + wm * 128 + 64 * lh + lr, rb = ra + 32;
  v5_pre_c1w r;
  r.sA = r.sB = r.cA = r.cB = 0u;
  if (colbase < p.N && ra < p.M) {
    r.sA = p.auxbits[ra * (long long)p.ld_auxbits + (colbase >> 5)];
    r.cA = p.c1bits[ra * (long long)p.ld_auxbits + (colbase >> 5)];
  }
  if (colbase < p.N && rb < p.M) {
    r.sB = p.auxbits[rb * (long long)p.ld_auxbits + (colbase >> 5)];
    r.cB = p.c1bits[rb * (long long)p.ld_auxbits + (colbase >> 5)];
  }
  // sample windows (rows past the valid time / the matrix read whatever the resource still covers or zeros: their dz is 0)
  const int Tp = p.Tp;
  const unsigned Ra = (unsigned)ra, sa = Ra / (unsigned)Tp, ta = Ra - sa * (unsigned)Tp;     // M < 2^31 (host-checked)
  unsigned sb = sa, tb = ta + 32;
  while (tb >= (unsigned)Tp) {
    tb -= (unsigned)Tp;
    ++sb;
  }
  const long long nseq = p.M / Tp;
  const unsigned s0 = (unsigned)(R0 / Tp);                   // first sequence of the tile (the resource starts there)
  const __amdgpu_buffer_rsrc_t rsX = rsrc_of(p.c1x + (long long)s0 * p.c1T, (nseq - s0) * (long long)p.c1T * 4);
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  r.xA = __builtin_bit_cast(f32x4, (v4u)__builtin_amdgcn_raw_buffer_load_b128(rsX, ((sa - s0) * (unsigned)p.c1T + 2u * ta) * 4u, 0u, 0));
  r.xB = __builtin_bit_cast(f32x4, (v4u)__builtin_amdgcn_raw_buffer_load_b128(rsX, ((sb - s0) * (unsigned)p.c1T + 2u * tb) * 4u, 0u, 0));
  return r;
}
__device__ __forceinline__ void v5_epilogue_c1w(const tl_nt_params& p, const f32x16 (&acc)[6], const v5_pre_c1w& pre, float* xw,
                                                float* red, long long R0, int n0, int wm, int wn, int lr, int lh, long long tm) {
  const int colbase = n0 + wn * 32;
  const bool colok = colbase < p.N;
  const int col = colbase + lr;
  const long long Qw = (R0 >> 2) + wm * 32;
  const long long Rw = 4 * Qw;
  const int Tp = p.Tp;
  const v5_rows rw = v5_rows_of(Qw, Tp);
  // rows that count: time below Tvalid, inside the matrix
  const unsigned long long okA = v5_valid_bits<64, 1>(rw.tA, Tp, p.Tvalid, Rw, p.M);
  const unsigned long long okB = v5_valid_bits<64, 1>(rw.tB, Tp, p.Tvalid, Rw + 64, p.M);
  float* xh = xw + lh * 256;                                 // this half's table: row r at xh + 4 r
  *reinterpret_cast<f32x4*>(xh + 4 * lr) = pre.xA;
  *reinterpret_cast<f32x4*>(xh + 4 * (32 + lr)) = pre.xB;
  c1w_acc ca;
  ca.clear();
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float y[4];
    wino43_rows(acc, e, y);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int r = 4 * e + h;
      const f32x4 xs = *reinterpret_cast<const f32x4*>(xh + 4 * r);
      const unsigned long long mpos = words_as_mask(r < 32 ? pre.sA : pre.sB, r & 31);
      const unsigned long long mam = words_as_mask(r < 32 ? pre.cA : pre.cB, r & 31);
      const float dz = selm0(mask2l(okA, okB, r), selm(mpos, y[h], y[h] * p.slope));
      ca.s[0] = fmaf(dz, selm(mam, xs[1], xs[0]), ca.s[0]);
      ca.s[1] = fmaf(dz, selm(mam, xs[2], xs[1]), ca.s[1]);
      ca.s[2] = fmaf(dz, selm(mam, xs[3], xs[2]), ca.s[2]);
      ca.b += dz;
    }
  }
  __syncthreads();                                          // `red` is a K-loop stage: every wave past its last fragment read
  c1w_reduce_store<4, 64>(p, red, ca, wm, wn * 32 + lr, lh, tm, col, colok);
}

}  // namespace tl

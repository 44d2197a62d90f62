// Epilogues shared by the F(4,3) NT kernels of tonal_wino.hip (operands transformed in the GEMM loop) and
// tonal_wino43v.hip (operands pre-transformed, fed by LDS-DMA): the four conv rows of a quad from its six Winograd
// products, then bias + LeakyReLU + max-pool + arg-max / sign bits (forward), bias + LeakyReLU (7-tap segments), the
// LeakyReLU' mask (input gradient) or the fused first-stage weight gradient.  One definition, so the two kernel families
// cannot drift apart.  Reference arithmetic: models/synthesis_models.py:87-97 and their backward.
#pragma once
#include "tonal_common.h"

namespace tl {

enum { W_LOAD_DIRECT = 0, W_LOAD_UNPOOL = 1, W_LOAD_V = 2 };
enum { W_EPI_LRELU = 1, W_EPI_POOL = 2, W_EPI_MASK = 3, W_EPI_C1W = 4, W_EPI_POOLV = 5, W_EPI_MASKY = 6, W_EPI_GY = 7 };   // numbering of tl_nt_params.epilogue

// ------------------------------------------------------------------------------------------
// Fused first-stage weight gradient (epilogue 4).  The input gradient of conv2 is G1 = dL/dZ of conv1
// at its arg-max; conv1 has one input channel, so its weight gradient is a contraction of G1 with the
// raw signal: dW1[o][j] = sum_rows G1[row][o] * x[seq][2t + a + j].  Doing it on the accumulators
// removes the 13.4 GB store of G1 and the kernel that re-read it.
// ------------------------------------------------------------------------------------------
struct c1w_acc {
  float s[3], b;
  __device__ __forceinline__ void clear() { s[0] = s[1] = s[2] = b = 0.f; }
};
// Walks the rows of one lane's accumulator elements without divisions: row R of the P1 row space is
// (sequence, t) with R = seq * Tp + t; `wofs` indexes the two bit arrays, `xo` the raw signal.
struct c1w_cursor {
  long long wofs, xo;
  int t;
  __device__ __forceinline__ void init(const tl_nt_params& p, long long R, int colbase) {
    const long long seq = R / p.Tp;
    t = (int)(R - seq * p.Tp);
    wofs = R * (long long)p.ld_auxbits + (colbase >> 5);
    xo = seq * (long long)p.c1T + 2 * t;
  }
  __device__ __forceinline__ void advance(const tl_nt_params& p, int rows) {
    t += rows;
    wofs += (long long)rows * p.ld_auxbits;
    xo += 2 * rows;
    while (t >= p.Tp) {
      t -= p.Tp;
      xo += p.c1T - 2 * p.Tp;
    }
  }
};
// one row at offset h from the cursor: G1 = y * LeakyReLU'(sign bit), contracted with x[2t + a + j]
__device__ __forceinline__ void c1w_row(c1w_acc& a, const tl_nt_params& p, const c1w_cursor& c, int h, float y, int lr) {
  const long long w = c.wofs + (long long)h * p.ld_auxbits;
  const bool pos = (p.auxbits[w] >> lr) & 1u;
  const bool am = (p.c1bits[w] >> lr) & 1u;
  const float* xp = p.c1x + c.xo + 2 * h;                 // wave-uniform address: broadcast loads
  const float x0 = xp[0], x1 = xp[1], x2 = xp[2], x3 = xp[3];
  const float dz = pos ? y : y * p.slope;
  a.s[0] = fmaf(dz, am ? x1 : x0, a.s[0]);
  if (p.c1kt > 1) a.s[1] = fmaf(dz, am ? x2 : x1, a.s[1]);
  if (p.c1kt > 2) a.s[2] = fmaf(dz, am ? x3 : x2, a.s[2]);
  a.b += dz;
}
// the same with the row's two bit words and four signal samples already in registers
__device__ __forceinline__ void c1w_row_vals(c1w_acc& a, const tl_nt_params& p, uint32_t sword, uint32_t cword, float x0,
                                             float x1, float x2, float x3, float y, int lr) {
  const bool pos = (sword >> lr) & 1u;
  const bool am = (cword >> lr) & 1u;
  const float dz = pos ? y : y * p.slope;
  a.s[0] = fmaf(dz, am ? x1 : x0, a.s[0]);
  if (p.c1kt > 1) a.s[1] = fmaf(dz, am ? x2 : x1, a.s[1]);
  if (p.c1kt > 2) a.s[2] = fmaf(dz, am ? x3 : x2, a.s[2]);
  a.b += dz;
}
// block reduction over the row dimension: lanes lr / lr + 32 and the NWM waves that share a column;
// red: LDS [NWM][NCOL][5].  Writes c1partial[tile][j][col] for the block's NCOL columns.
template <int NWM, int NCOL>
__device__ __forceinline__ void c1w_reduce_store(const tl_nt_params& p, float* red, const c1w_acc& a, int wm, int cl,
                                                 int lh, long long tile, int col, bool colok) {
  float v[4] = {a.s[0], a.s[1], a.s[2], a.b};
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] += __shfl_xor(v[j], 32);
  if (lh == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(wm * NCOL + cl) * 4 + j] = v[j];
  }
  __syncthreads();
  if (wm == 0 && lh == 0 && colok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NWM; ++w) t += red[(w * NCOL + cl) * 4 + j];
      v[j] = t;
    }
    float* dst = p.c1partial + tile * (long long)(p.c1kt + 1) * p.N;
    for (int j = 0; j < p.c1kt; ++j) dst[(long long)j * p.N + col] = v[j];
    dst[(long long)p.c1kt * p.N + col] = v[3];
  }
  __syncthreads();
}


// Accumulator layout: wave (wm, wn) of a workgroup tile that starts at conv row R0 / column n0 holds, for transform i,
// acc[i][e] = M_i[quad (R0 >> 2) + wm * 32 + 4 * lh + (e & 3) + 8 * (e >> 2)][column n0 + wn * 32 + lr].
// `lds`: the workgroup's LDS, free for scratch when this is called (every wave past its last fragment read).
// FULL: the tile lies wholly inside the output (rows R0 .. R0 + 511 < M, columns n0 .. n0 + 63 < N) - the per-store
// bounds tests are compile-time true and the 32 - 64 stores of a lane are straight-line code instead of exec-masked regions.
template <int EPI, bool FULL = false>
__device__ __forceinline__ void wino43_epilogue(const tl_nt_params& p, const f32x16 (&acc)[6], float* lds, long long R0,
                                                int n0, int wm, int wn, int lr, int lh, long long tm) {
  // ---- epilogue: the four conv rows of a quad from its six products ----
  const long long Q0 = (R0 >> 2) + wm * 32 + 4 * lh;       // quad of accumulator element e = 0
  const int col = n0 + wn * 32 + lr;
  const int colbase = n0 + wn * 32;
  const bool colok = FULL || col < p.N;
  float bv = 0.f;
  if constexpr (EPI == W_EPI_POOL || EPI == W_EPI_LRELU) bv = (colok && p.bias) ? p.bias[col] : 0.f;
  uint32_t wbits = 0, wsign = 0;
  (void)wbits; (void)wsign;
  // MASK: the 64 sign words this lane group needs (rows 4 (Q0 + qo) + h), one per lane, fetched up front
  uint32_t swordA = 0, swordB = 0;
  if constexpr (EPI == W_EPI_MASK) {
    if (p.auxbits != nullptr) {
      // lane lr holds the word of row index lr (e = lr >> 2, h = lr & 3) in A and of row index 32 + lr in B
      const int eA = lr >> 2, eB = 8 + (lr >> 2), hh = lr & 3;
      const long long RA = 4 * (Q0 + (eA & 3) + 8 * (eA >> 2)) + hh, RB = 4 * (Q0 + (eB & 3) + 8 * (eB >> 2)) + hh;
      if (FULL || (RA < p.M && colbase < p.N)) swordA = p.auxbits[RA * (long long)p.ld_auxbits + (colbase >> 5)];
      if (FULL || (RB < p.M && colbase < p.N)) swordB = p.auxbits[RB * (long long)p.ld_auxbits + (colbase >> 5)];
    }
  }
  // C1WGRAD: per row two bit words and four signal samples.  Lane lr of each half-wave fetches them for
  // row index lr (set A) and 32 + lr (set B) up front; the row loop reads them with lane shuffles
  // instead of 6 dependent global loads per row.
  uint32_t cwA = 0, cwB = 0;
  f32x4 xsA = {0.f, 0.f, 0.f, 0.f}, xsB = xsA;
  if constexpr (EPI == W_EPI_C1W) {
    const int eA = lr >> 2, eB = 8 + (lr >> 2), hh = lr & 3;
    const long long RA = 4 * (Q0 + (eA & 3) + 8 * (eA >> 2)) + hh, RB = 4 * (Q0 + (eB & 3) + 8 * (eB >> 2)) + hh;
    c1w_cursor c;
    if (FULL || (RA < p.M && colbase < p.N)) {
      c.init(p, RA, colbase);
      swordA = p.auxbits[c.wofs];
      cwA = p.c1bits[c.wofs];
      if (c.t < p.Tvalid) xsA = f32x4{p.c1x[c.xo], p.c1x[c.xo + 1], p.c1x[c.xo + 2], p.c1x[c.xo + 3]};
    }
    if (FULL || (RB < p.M && colbase < p.N)) {
      c.init(p, RB, colbase);
      swordB = p.auxbits[c.wofs];
      cwB = p.c1bits[c.wofs];
      if (c.t < p.Tvalid) xsB = f32x4{p.c1x[c.xo], p.c1x[c.xo + 1], p.c1x[c.xo + 2], p.c1x[c.xo + 3]};
    }
  }
  c1w_acc ca;
  ca.clear();
  c1w_cursor cur;                                          // time index of the quad without per-row divisions
  if constexpr (EPI == W_EPI_POOL || EPI == W_EPI_C1W) cur.init(p, 4 * Q0, colbase);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int qo = (e & 3) + 8 * (e >> 2);
    const long long Q = Q0 + qo;
    if constexpr (EPI == W_EPI_POOL || EPI == W_EPI_C1W)
      if (e > 0) cur.advance(p, (e & 3) ? 4 : 20);           // quad offsets 0,1,2,3, 8,.. -> row steps 4,4,4,20
    const float m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e];
    const float a12 = m1 + m2, s12 = m1 - m2, a34 = m3 + m4, s34 = m3 - m4;
    float y[4];
    y[0] = (acc[0][e] + a12) + a34;
    y[1] = s12 + 2.f * s34;
    y[2] = a12 + 4.f * a34;
    y[3] = (s12 + 8.f * s34) + acc[5][e];
    if constexpr (EPI == W_EPI_POOL) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const long long P = 2 * Q + h;                      // pooled row
        const float y0 = lrelu(y[2 * h] + bv, p.slope), y1 = lrelu(y[2 * h + 1] + bv, p.slope);
        const bool rowok = FULL || 2 * P < p.M;
        const bool valid = rowok && (cur.t + 2 * h) < p.Tvalid;     // Tp % 4 == 0: a quad never wraps
        const bool sel = valid && colok && (y1 > y0);
        const float o = valid ? (sel ? y1 : y0) : 0.f;
        if (rowok && colok) p.out[P * (long long)p.ldo + col] = o;
        // lane lr of each half-wave keeps the two bit words of pooled row number lr of this lane group
        const unsigned long long m = __ballot(sel);
        const unsigned long long ms = __ballot(o > 0.f);
        if (lr == 2 * e + h) {
          wbits = (uint32_t)(m >> (32 * lh));
          wsign = (uint32_t)(ms >> (32 * lh));
        }
      }
    } else if constexpr (EPI == W_EPI_LRELU) {
      const long long R = 4 * Q;
      if (FULL || (R < p.M && colok)) {                     // M % 4 == 0: the quad shares validity
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          // aux: taps the segments do not cover, accumulated by the caller (pre-activation, no bias)
          const float extra = p.aux != nullptr ? p.aux[(R + h) * (long long)p.ldaux + col] : 0.f;
          p.out[(R + h) * (long long)p.ldo + col] = lrelu((y[h] + extra) + bv, p.slope);
        }
      }
    } else if constexpr (EPI == W_EPI_C1W) {
      {
        const bool live = FULL || (4 * Q < p.M && colok);     // Tp % 4 == 0: the quad stays inside one sequence
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const int src = ((4 * e + h) & 31) + 32 * lh;
          const uint32_t sw = __shfl((e < 8) ? swordA : swordB, src), cw = __shfl((e < 8) ? cwA : cwB, src);
          const f32x4 xs = (e < 8) ? xsA : xsB;
          const float x0 = __shfl(xs[0], src), x1 = __shfl(xs[1], src), x2 = __shfl(xs[2], src), x3 = __shfl(xs[3], src);
          if (live && cur.t + h < p.Tvalid) c1w_row_vals(ca, p, sw, cw, x0, x1, x2, x3, y[h], lr);
        }
      }
    } else {
      const long long R = 4 * Q;
      if (FULL || (R < p.M && colok)) {                     // M % 4 == 0: the quad shares validity
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          bool pos;
          if (p.auxbits != nullptr) {
            // word of row index 4 e + h: lane (4 e + h) & 31 of this half-wave holds it
            const uint32_t mine = (e < 8) ? swordA : swordB;
            const uint32_t word = __shfl(mine, ((4 * e + h) & 31) + 32 * lh);
            pos = (word >> lr) & 1u;
          } else {
            pos = p.aux[(R + h) * (long long)p.ldaux + col] > 0.f;
          }
          p.out[(R + h) * (long long)p.ldo + col] = pos ? y[h] : y[h] * p.slope;
        }
      }
    }
  }
  if constexpr (EPI == W_EPI_POOL) {
    const int e = lr >> 1, h = lr & 1;
    const long long P = 2 * (Q0 + (e & 3) + 8 * (e >> 2)) + h;
    if (FULL || (2 * P < p.M && colbase < p.N)) {
      p.obits[P * (long long)p.ld_obits + (colbase >> 5)] = wbits;
      if (p.osign != nullptr) p.osign[P * (long long)p.ld_obits + (colbase >> 5)] = wsign;
    }
  }
  if constexpr (EPI == W_EPI_C1W) c1w_reduce_store<4, 64>(p, lds, ca, wm, wn * 32 + lr, lh, tm, col, colok);
}

}  // namespace tl

// Winograd F(4,3) weight gradient of the pooled 3-tap convolutions (conv2 / conv3 of the ECoG stack,
// models/synthesis_models.py:91-97; their dW in loss.backward(), models/synthesis_trainer.py:226) on the
// fp32 matrix pipe of gfx950.
//
// By the transposition principle the weight gradient of y = A^T[(G g) . (B^T d)] is
//     dg = G^T [ sum_quads (B^T d) (x) (A dy) ]
// with, per quad Q (conv rows 4Q..4Q+3, input rows d0..d5 = 4Q..4Q+5),
//     V = B^T d :  V0 = 4d0 - 5d2 + d4        V1 = -4d1 - 4d2 + d3 + d4    V2 = 4d1 - 4d2 - d3 + d4
//                  V3 = -2d1 - d2 + 2d3 + d4  V4 = 2d1 - d2 - 2d3 + d4     V5 = 4d1 - 5d3 + d5
//     Y = A dy  :  Y0 = dy0   Y1 = dy0+dy1+dy2+dy3   Y2 = dy0-dy1+dy2-dy3
//                  Y3 = dy0+2dy1+4dy2+8dy3   Y4 = dy0-2dy1+4dy2-8dy3   Y5 = dy3
// i.e. six accumulated outer products [C_in x C_out] per quad instead of the twelve of the direct form
// (F(2,3): eight): half the MFMA work.  dy is the un-pooled dZ: of each pool pair exactly one row is the
// pooled gradient (arg-max bit), so Y needs two loads and a handful of selects.
//
// Both transforms are applied ONCE, by the thread that stages the operand from global memory into LDS
// (transform-at-staging): the LDS tiles hold V and Y, and the MFMA loop is a pure GEMM loop - one
// ds_read_b32 per operand, no VALU between the MFMAs.  (The F(2,3) kernel transforms fragments in the
// consumer loop, every wave redoing it; for F(4,3) that costs 3.7 VALU ops per MFMA.)
//
// Workgroup: 4 waves as 2 (C_in) x 2 (C_out), tile 64 x 64, wave tile 32 x 32 x 6 transforms = 96
// accumulator registers; 48 KB LDS -> two or three independent workgroups per CU (their barriers and
// epilogues overlap).  A K-step is 8 quads = 32 conv rows; every thread stages one (quad, channel pair) of
// V (6 row loads -> 6 transformed float2) and one of Y; global loads run two K-steps ahead.  LDS layout
// [transform][quad][64 channels]; rows of odd quads are stored with the two 32-channel halves swapped so
// that the two lane halves of a fragment read (quad 2s / 2s + 1) hit disjoint banks without padding.
#include "tonal_common.h"
#include <type_traits>

namespace tl {

#ifndef T4_PIN
#define T4_PIN 1          // 1: scheduling fence after the global loads of a K-step (see kstep)
#endif
constexpr int T4_BM = 64, T4_BN = 64, T4_Q = 8;          // C_in tile, C_out tile, quads per K-step
constexpr int T4_PLANE = T4_Q * 64;                       // floats per transform plane
constexpr int T4_TILE = 6 * T4_PLANE;                     // floats per operand tile (12 KB)

__global__ __launch_bounds__(256, 2) void wino43_tn_kernel(const tl_tn_params p) {
  __shared__ __attribute__((aligned(16))) float lds[4 * T4_TILE];
  float* As = lds;                       // [2][6][8][64]  V
  float* Bs = lds + 2 * T4_TILE;         // [2][6][8][64]  Y
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int ntm = (p.Mdim + T4_BM - 1) / T4_BM, ntn = (p.Ndim + T4_BN - 1) / T4_BN;
  const long long tiles = (long long)ntm * ntn;
  const long long nwg = tiles * p.splitk;
  long long bid = (long long)blockIdx.y * gridDim.x + blockIdx.x;
  {   // XCD-aware order: the workgroups of one reduction split (same activation / gradient rows, all
      // C_in x C_out tiles) are resident on one XCD together, so its L2 serves the 8-fold panel re-reads
    const long long q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int z = (int)(bid / tiles);
  const int tt = (int)(bid % tiles);
  const int m0 = (tt / ntn) * T4_BM, n0 = (tt % ntn) * T4_BN;

  const long long quads_all = p.Krows >> 2;
  const long long ksteps_all = (quads_all + T4_Q - 1) / T4_Q;
  const long long per = (ksteps_all + p.splitk - 1) / p.splitk;
  const long long ks_begin = z * per;
  long long ks_end = ks_begin + per;
  if (ks_end > ksteps_all) ks_end = ksteps_all;
  const long long nsteps = ks_end > ks_begin ? ks_end - ks_begin : 0;

  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // ---- staging: every thread transforms one (quad, channel pair) of V and one of Y per K-step: the same
  // branch-free instruction stream in all four waves, which the scheduler interleaves with the MFMAs.
  // Measured alternatives at the conv2 shape (this form: 50.5 ms): roles split by wave with float4 tasks
  // inside a shared loop (exec-masked transform blocks the matrix pipe idles through) 53.0 ms; the same
  // split with the whole K loop instantiated per role behind a scalar branch 52.3 ms; a 128 x 128 tile
  // with one wave per SIMD and 384 accumulators does not fit the register file (850 spilled registers). ----
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int qi = tid >> 5, c2 = tid & 31;                   // quad of the K-step, channel pair
  const int sw = ((c2 * 2) ^ ((qi & 1) << 5));              // swizzled channel position inside the 64-wide row
  // A: rows 4 q .. 4 q + 5 of the activation matrix, clamped into it (a clamped row only ever meets dy = 0
  // or enters through the Winograd identity, which holds for any finite d)
  const long long a_last = (p.A_rows < p.Krows + 2 ? p.A_rows : p.Krows + 2) - 1;
  const int acol = m0 + ((m0 + c2 * 2) < p.Mdim ? c2 * 2 : 0);
  // B: pooled rows 2 q, 2 q + 1 and their arg-max bits
  const long long b_last = p.B_rows - 1;
  const int ncol = n0 + c2 * 2;
  const bool bnok = ncol < p.Ndim;
  const int ncolc = bnok ? ncol : n0;
  long long quad = ks_begin * T4_Q + qi;                    // quad the NEXT load fetches for this thread
  int tq = (int)((4 * quad) % p.Tp);                        // time index of its first conv row
  const int dstep = (4 * T4_Q) % p.Tp;
  long long ld_q0 = ks_begin * T4_Q;                        // first quad of the K-step the next load fetches (uniform)
  const unsigned a_toff = (unsigned)(qi * 4 * p.lda + (acol - m0));
  const unsigned b_toff = (unsigned)(qi * 2 * p.ldb + (ncolc - n0));
  const unsigned w_toff = (unsigned)(qi * 2 * p.ld_bbits + (ncolc >> 5));

  struct stage_regs {
    f32x2 d[6], g[2];
    uint32_t wa, wb;  // arg-max words of the two pooled rows (bits of this thread's channel pair at ncolc & 31)
    uint32_t ok;      // bit 0 / 1: pair a / b holds a valid gradient
  };
  stage_regs rP, rQ;
  f32x2 bsum = {0.f, 0.f};          // running column sums of dZ (bias gradient) of this thread's channel pair

  auto load_regs = [&](auto FAST, stage_regs& r) {
    constexpr bool fast = decltype(FAST)::value;
    const long long row0 = 4 * quad;
    long long pa = 2 * quad, pb = 2 * quad + 1;
    bool va = bnok && tq < p.Tvalid, vb = bnok && tq + 2 < p.Tvalid;
    if constexpr (fast) {
      // wave-uniform row base (scalar registers, advanced per step) + a per-thread 32-bit offset that is fixed
      // for the whole kernel: no 64-bit vector address arithmetic in the loop
      const float* au = p.A + (ld_q0 * 4) * (long long)p.lda + m0;
#pragma unroll
      for (int j = 0; j < 6; ++j) r.d[j] = *reinterpret_cast<const f32x2*>(au + (long long)j * p.lda + a_toff);
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        long long row = row0 + j;
        row = row < a_last ? row : a_last;
        r.d[j] = *reinterpret_cast<const f32x2*>(p.A + row * (long long)p.lda + acol);
      }
      va = va && 4 * quad < p.Krows && pa <= b_last;
      vb = vb && 4 * quad + 2 < p.Krows && pb <= b_last;
      pa = pa < b_last ? pa : b_last;
      pb = pb < b_last ? pb : b_last;
    }
    if constexpr (fast) {
      const float* bu = p.B + (ld_q0 * 2) * (long long)p.ldb + n0;
      const uint32_t* wu = p.bbits + (ld_q0 * 2) * (long long)p.ld_bbits;
      r.g[0] = *reinterpret_cast<const f32x2*>(bu + b_toff);
      r.g[1] = *reinterpret_cast<const f32x2*>(bu + p.ldb + b_toff);
      r.wa = wu[w_toff];
      r.wb = wu[p.ld_bbits + w_toff];
    } else {
      r.g[0] = *reinterpret_cast<const f32x2*>(p.B + pa * (long long)p.ldb + ncolc);
      r.g[1] = *reinterpret_cast<const f32x2*>(p.B + pb * (long long)p.ldb + ncolc);
      r.wa = p.bbits[pa * (long long)p.ld_bbits + (ncolc >> 5)];
      r.wb = p.bbits[pb * (long long)p.ld_bbits + (ncolc >> 5)];
    }
    r.ok = (va ? 1u : 0u) | (vb ? 2u : 0u);
    ld_q0 += T4_Q;
    quad += T4_Q;
    tq += dstep;
    if (tq >= p.Tp) tq -= p.Tp;
  };

  auto store_a = [&](const stage_regs& r, int buf) {
    const f32x2 d0 = r.d[0], d1 = r.d[1], d2 = r.d[2], d3 = r.d[3], d4 = r.d[4], d5 = r.d[5];
    const f32x2 s1 = d4 - 4.f * d2, s2 = d3 - 4.f * d1, s3 = d4 - d2, t = d3 - d1;
    f32x2 o[6];
    o[0] = 4.f * d0 + (d4 - 5.f * d2);
    o[1] = s1 + s2;
    o[2] = s1 - s2;
    o[3] = s3 + 2.f * t;
    o[4] = s3 - 2.f * t;
    o[5] = (4.f * d1 - 5.f * d3) + d5;
    float* dst = As + buf * T4_TILE + qi * 64 + sw;
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x2*>(dst + i * T4_PLANE) = o[i];
  };
  // Y from the two pooled gradients a, b of the quad and their arg-max bits: the un-pooled rows are
  // dy0 = a (bit clear) | dy1 = a (bit set), dy2 / dy3 likewise from b - selected with bit masks (v_bfe_i32
  // + v_and) instead of compare / select pairs
  auto store_b = [&](const stage_regs& r, int buf) {
    const int sh = ncolc & 31;
    f32x2 o[6];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint32_t ma = (uint32_t)__builtin_amdgcn_sbfe((int)r.wa, sh + c, 1);      // all ones if the odd row won
      const uint32_t mb = (uint32_t)__builtin_amdgcn_sbfe((int)r.wb, sh + c, 1);
      const uint32_t ua = (r.ok & 1u) ? __float_as_uint(r.g[0][c]) : 0u, ub = (r.ok & 2u) ? __float_as_uint(r.g[1][c]) : 0u;
      const float e_a = __uint_as_float(ua & ~ma), o_a = __uint_as_float(ua & ma);
      const float e_b = __uint_as_float(ub & ~mb), o_b = __uint_as_float(ub & mb);
      o[0][c] = e_a;
      o[1][c] = (e_a + o_a) + (e_b + o_b);
      o[2][c] = (e_a - o_a) + (e_b - o_b);
      o[3][c] = fmaf(4.f, fmaf(2.f, o_b, e_b), fmaf(2.f, o_a, e_a));
      o[4][c] = fmaf(4.f, fmaf(-2.f, o_b, e_b), fmaf(-2.f, o_a, e_a));
      o[5][c] = o_b;
    }
    bsum += o[1];                                        // Y1 = dy0 + dy1 + dy2 + dy3: the bias gradient of the quad
    float* dst = Bs + buf * T4_TILE + qi * 64 + sw;
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x2*>(dst + i * T4_PLANE) = o[i];
  };

  // ---- MFMA side: k-slice sl of a K-step = quads 2 sl (lanes 0-31) and 2 sl + 1 (lanes 32-63) ----
  const int a_off = lh * 64 + ((wm * 32 + lr) ^ (lh << 5));
  const int b_off = lh * 64 + ((wn * 32 + lr) ^ (lh << 5));
  float fa0[6], fb0[6], fa1[6], fb1[6];
  auto load_frag = [&](float (&fa)[6], float (&fb)[6], int buf, int sl) {
    const float* a_s = As + buf * T4_TILE + sl * 128 + a_off;
    const float* b_s = Bs + buf * T4_TILE + sl * 128 + b_off;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      fa[i] = a_s[i * T4_PLANE];
      fb[i] = b_s[i * T4_PLANE];
    }
  };
  auto mfma6 = [&](const float (&fa)[6], const float (&fb)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[i], acc[i], 0, 0, 0);
  };

  // One K-step.  The tiles of step s + 2 are requested at the top, the MFMAs of slice 3 of the previous
  // step (carried in registers across the barrier) run while the first fragments of this step arrive,
  // the operands of step s + 1 are transformed and written to the other LDS buffer mid-step.
  auto kstep = [&](auto TAIL, long long s, stage_regs& r_ld, const stage_regs& r_st) {
    constexpr bool tail = decltype(TAIL)::value;
    const int buf = (int)(s & 1);
    load_frag(fa0, fb0, buf, 0);
    if constexpr (!tail) load_regs(std::true_type{}, r_ld);
    else if (s + 3 < nsteps) load_regs(std::false_type{}, r_ld);
#if T4_PIN
    // keep the global loads HERE: left alone, the scheduler sinks them two K-steps down, next to the
    // transform that consumes them (shorter live ranges), and the prefetch becomes an exposed round trip
    __builtin_amdgcn_sched_barrier(0);
#endif
    mfma6(fa1, fb1);                                        // slice 3 of the previous step
    load_frag(fa1, fb1, buf, 1);
    mfma6(fa0, fb0);
    if (!tail || s + 1 < nsteps) store_a(r_st, buf ^ 1);
    load_frag(fa0, fb0, buf, 2);
    mfma6(fa1, fb1);
    if (!tail || s + 1 < nsteps) store_b(r_st, buf ^ 1);
    load_frag(fa1, fb1, buf, 3);
    mfma6(fa0, fb0);
    __syncthreads();
  };
  using Y = std::true_type;
  using N = std::false_type;

#pragma unroll
  for (int i = 0; i < 6; ++i) fa1[i] = fb1[i] = 0.f;         // carried slice of step -1: adds nothing

  // three register sets: the loads of step s + 3 are issued at the top of step s
  stage_regs rR;
  if (nsteps > 0) {
    load_regs(N{}, rP);
    store_a(rP, 0);
    store_b(rP, 0);
    if (nsteps > 1) load_regs(N{}, rQ);
    if (nsteps > 2) load_regs(N{}, rR);
  }
  __syncthreads();
  long long s = 0;
  const bool whole = p.A_rows >= p.Krows && 2 * p.B_rows >= p.Krows;
  // kstep(s, ld, st): loads step s + 3 into ld (the set that held step s, already in LDS), stores st = step s + 1
  if (whole)
    for (; s + 8 < nsteps; s += 3) {
      kstep(N{}, s, rP, rQ);
      kstep(N{}, s + 1, rQ, rR);
      kstep(N{}, s + 2, rR, rP);
    }
  for (; s < nsteps; s += 3) {
    kstep(Y{}, s, rP, rQ);
    if (s + 1 < nsteps) kstep(Y{}, s + 1, rQ, rR);
    if (s + 2 < nsteps) kstep(Y{}, s + 2, rR, rP);
  }
  mfma6(fa1, fb1);

  // bias-gradient partial sums: the workgroups of the first C_in tile write the column sums of their split
  if (p.colsum != nullptr && m0 == 0) {
    __syncthreads();                                         // all fragment reads of the last step are done
    float* red = lds;                                        // [8 quads][64 channels]
    *reinterpret_cast<f32x2*>(red + qi * 64 + c2 * 2) = bsum;
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < T4_Q; ++q) t += red[q * 64 + tid];
      if (n0 + tid < p.Ndim) p.colsum[(long long)z * p.Ndim + n0 + tid] = t;
    }
  }
  float* out = p.slab + (long long)z * p.slab_stride;
  const int col = n0 + wn * 32 + lr;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (m < p.Mdim && col < p.Ndim) out[((long long)i * p.Mdim + m) * (long long)p.ldc + col] = acc[i][e];
    }
}

// dW (O, I, 3) = G^T M from the reduced transforms red[6][I][ld]:
//   G^T = [ 1/4 -1/6 -1/6 1/24  1/24 0 ;  0 -1/6 1/6 1/12 -1/12 0 ;  0 -1/6 -1/6 1/6 1/6 1 ]
__global__ void wino43_wgrad_finalize_kernel(const float* __restrict__ red, float* __restrict__ gw, int O, int I, int ld) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)O * I) return;
  const int i = (int)(idx / O), o = (int)(idx % O);
  const long long plane = (long long)I * ld;
  const float* s = red + (long long)i * ld + o;
  const float m0 = s[0], m1 = s[plane], m2 = s[2 * plane], m3 = s[3 * plane], m4 = s[4 * plane], m5 = s[5 * plane];
  const float a12 = m1 + m2, s12 = m2 - m1, a34 = m3 + m4, s34 = m3 - m4;
  float* d = gw + ((long long)o * I + i) * 3;
  d[0] = 0.25f * m0 - (1.f / 6.f) * a12 + (1.f / 24.f) * a34;
  d[1] = (1.f / 6.f) * s12 + (1.f / 12.f) * s34;
  d[2] = (1.f / 6.f) * (a34 - a12) + m5;
}

}  // namespace tl

extern "C" int tl_conv3_wino43_tn(const tl_tn_params* pp, void* stream) {
  using namespace tl;
  TL_REQUIRE(pp != nullptr, "wino43_tn: null params");
  tl_tn_params p = *pp;
  if (p.splitk < 1) p.splitk = 1;
  TL_REQUIRE(p.A && p.B && p.slab && p.bbits, "wino43_tn: null A/B/bbits/slab");
  TL_REQUIRE(p.J == 3 && p.loader == 1, "wino43_tn: 3 taps, UNPOOL loader only");
  TL_REQUIRE(p.Krows > 0 && p.Krows % 4 == 0 && p.Mdim > 0 && p.Ndim > 0, "wino43_tn: bad sizes (Krows %% 4 must be 0)");
  TL_REQUIRE(p.Krows + 64 < (1LL << 31), "wino43_tn: more than 2^31 reduction rows");
  TL_REQUIRE(p.A_rows > 0 && p.B_rows > 0, "wino43_tn: empty operand");
  TL_REQUIRE(p.Mdim % 4 == 0 && p.Ndim % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0, "wino43_tn: dims/ld must be multiples of 4");
  TL_REQUIRE(p.lda >= p.Mdim && p.ldb >= p.Ndim && p.ldc >= p.Ndim, "wino43_tn: leading dimension too small");
  TL_REQUIRE(p.Tp > 0 && p.Tp % 4 == 0 && p.Tvalid % 2 == 0, "wino43_tn: Tp %% 4 == 0 and an even Tvalid needed");
  TL_REQUIRE(p.ld_bbits * 32 >= p.Ndim, "wino43_tn: bbits row too short");
  TL_REQUIRE(p.splitk <= 65535, "wino43_tn: splitk too large");
  TL_REQUIRE(p.splitk == 1 || p.slab_stride >= 6LL * p.Mdim * p.ldc, "wino43_tn: slab_stride smaller than 6*Mdim*ldc");
  const long long t = (long long)((p.Mdim + T4_BM - 1) / T4_BM) * ((p.Ndim + T4_BN - 1) / T4_BN);
  TL_REQUIRE(t < (1LL << 31), "wino43_tn: grid too large");
  hipLaunchKernelGGL(wino43_tn_kernel, dim3((unsigned)t, (unsigned)p.splitk, 1), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("wino43_tn");
}

extern "C" int tl_wino43_wgrad_finalize(const float* red, float* gw, int O, int I, int ld, void* stream) {
  using namespace tl;
  TL_REQUIRE(red && gw && O > 0 && I > 0 && ld >= O, "wino43_wgrad_finalize: bad arguments");
  const long long n = (long long)O * I;
  hipLaunchKernelGGL(wino43_wgrad_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     red, gw, O, I, ld);
  return check_launch("wino43_wgrad_finalize");
}

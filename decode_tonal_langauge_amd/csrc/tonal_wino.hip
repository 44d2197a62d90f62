// Winograd forms of the 3-tap (k,1) convolutions (conv2 / conv3 of the ECoG stack,
// models/synthesis_models.py:91-97) on the fp32 matrix pipe of gfx950: F(2,3) for all three passes
// (wino_nt_kernel, wino_tn_kernel) and F(4,3) for the forward / input-gradient passes
// (wino43_nt_kernel, the default; its own header is further down).  F(2,3):
//
// The max-pool (2,1) that follows each of these convolutions groups the conv rows in pairs
// (2P, 2P+1); a pair needs the four input rows d0..d3 = 2P .. 2P+3 and
//     y0 = m0 + m1 + m2,  y1 = m1 - m2 - m3,   m_i = (B^T d)_i . (G g)_i
//     B^T d = [d0 - d2, d1 + d2, d2 - d1, d1 - d3],  G g = [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2]
// i.e. 4 channel contractions per pair instead of 6: 2/3 of the MFMA work of the direct form for
// the forward pass, the input gradient (same algorithm on dZ with flipped, transposed taps) and -
// by the transposition principle - the weight gradient:
//     dg = G^T [ (A dy) (x) (B^T d) ],   A dy = [dy0, dy0 + dy1, dy0 - dy1, -dy1]
// All transform constants are 0, +-1, 1/2: exact in fp32; only the summation order differs from
// the direct convolution.
//
// Layout is that of tonal_gemm.hip (rows = (sequence, time), channels last).  LDS keeps the staged
// input rows in two planes (even rows E, odd rows O, 144-byte row stride), so the four rows of a
// pair are E[p], O[p], E[p+1], O[p+1] and a fragment of B^T d is two conflict-free ds_read_b128
// plus one vector add.
#include "tonal_common.h"
#include "tonal_wino43_epi.h"
#include <type_traits>

// The LLVM scheduler sinks the ds_reads of the next fragment set towards their first use; pinning the
// hand-written order with scheduling fences (-DWINO_FENCE) measured 3 % slower, so it is off.
#ifdef WINO_FENCE
#define W_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define W_SCHED_FENCE() ((void)0)
#endif

namespace tl {


constexpr int W_BP = 128;            // output pairs per workgroup (256 conv rows)
constexpr int W_BN = 128;            // output columns per workgroup
constexpr int W_BK = 32;             // K depth of one stage
constexpr int W_LD = W_BK + 4;       // 36 floats: conflict-free ds_read_b128
constexpr int W_PR = W_BP + 1;       // staged pairs per plane
#ifndef WINO_MI
#define WINO_MI 1
#endif
constexpr int W_MI = WINO_MI;       // 32-pair MFMA tiles per wave along M (1: 8 waves, 2: 4 waves per workgroup)

// ------------------------------------------------------------------------------------------
// weights: torch (O, I, 3, 1) -> forward taps [4][O][ld_f] and input-gradient taps [4][I][ld_d]
// (flipped and transposed: V_j' = W_{2-j'}^T)
// ------------------------------------------------------------------------------------------
__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ fwd, float* __restrict__ dgr,
                                    int O, int I, int ld_f, int ld_d) {
  const long long n_f = (long long)O * ld_f, n_d = (long long)I * ld_d;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (fwd != nullptr && idx < n_f) {
    const int o = (int)(idx / ld_f), i = (int)(idx % ld_f);
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (i < I) {
      const float* s = w + ((long long)o * I + i) * 3;
      g0 = s[0], g1 = s[1], g2 = s[2];
    }
    fwd[idx] = g0;
    fwd[n_f + idx] = 0.5f * ((g0 + g2) + g1);
    fwd[2 * n_f + idx] = 0.5f * ((g0 + g2) - g1);
    fwd[3 * n_f + idx] = g2;
  }
  if (dgr != nullptr && idx < n_d) {
    const int i = (int)(idx / ld_d), o = (int)(idx % ld_d);
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (o < O) {
      const float* s = w + ((long long)o * I + i) * 3;
      g0 = s[2], g1 = s[1], g2 = s[0];          // flipped taps
    }
    dgr[idx] = g0;
    dgr[n_d + idx] = 0.5f * ((g0 + g2) + g1);
    dgr[2 * n_d + idx] = 0.5f * ((g0 + g2) - g1);
    dgr[3 * n_d + idx] = g2;
  }
}


// ------------------------------------------------------------------------------------------
// NT form: forward (DIRECT loader, POOL epilogue) and input gradient (UNPOOL loader, MASK epilogue)
// A K-step is (32-deep channel chunk, transform index i); step i accumulates into m_i.
// Software pipeline as in nt_window_kernel: register-staged global loads (B two steps ahead, the
// A chunk one chunk ahead), LDS double buffering, fragment sets F0/F1 with the last k-group of a
// step carried across the barrier.
// ------------------------------------------------------------------------------------------
template <int LOADER, int EPI, int MI>
__global__ __launch_bounds__(512 / MI, 2 / MI) void wino_nt_kernel(const tl_nt_params p) {
  // MI = 1: 8 waves, 4 (pairs) x 2 (columns), wave tile 32 pairs x 64 columns, 2 waves per SIMD
  // MI = 2: 4 waves, 2 x 2, wave tile 64 pairs x 64 columns (256 accumulator registers), 1 wave per SIMD
  constexpr int NTHR = 512 / MI;
  constexpr int PLANE = W_PR * W_LD;
  constexpr int A_F4 = (LOADER == W_LOAD_DIRECT) ? ((2 * W_PR * 8 + NTHR - 1) / NTHR) : ((W_PR * 8 + NTHR - 1) / NTHR);
  constexpr int B_F4 = W_BN * 8 / NTHR;

  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * PLANE + 2 * W_BN * W_LD];
  float* As = lds;                               // [2 buffers][2 planes][W_PR][W_LD]
  float* Bs = lds + 2 * 2 * PLANE;               // [2][W_BN][W_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const int ntn = (p.N + W_BN - 1) / W_BN;
  const long long ntm = (p.M + 2 * W_BP - 1) / (2 * W_BP);
  const long long nwg = ntm * ntn;
  long long bid = blockIdx.x;
  {
    const long long q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const long long tm = bid / ntn;
  const int tn = (int)(bid % ntn);
  const long long R0 = tm * (2 * W_BP);          // first conv row of the tile (even)
  const int n0 = tn * W_BN;
  const int nchunks = p.K / W_BK;                // host-checked: K % 32 == 0
  const int nsteps = nchunks * 4;

  f32x16 acc[4][MI][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][m][j][e] = 0.f;

  f32x4 ra[A_F4];
  uint32_t rbits[A_F4];
  f32x4 rbP[B_F4], rbQ[B_F4];
  (void)rbits;

  const long long Abase = R0 + p.row_shift;      // first staged input row (even)
  const float* aptr[A_F4];
  const uint32_t* abptr[A_F4];
  bool aok[A_F4];
  (void)abptr;
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int idx = tid + i * NTHR;
    const int r = idx >> 3, c4 = idx & 7;
    if constexpr (LOADER == W_LOAD_DIRECT) {
      const long long row = Abase + r;
      aok[i] = r < 2 * W_PR && row >= 0 && row < p.A_rows;
      aptr[i] = p.A + (aok[i] ? row : 0) * (long long)p.lda + c4 * 4;
      abptr[i] = nullptr;
    } else {
      const long long prow = (Abase >> 1) + r;
      aok[i] = r < W_PR && prow >= 0 && prow < p.A_rows && (int)((2 * prow) % p.Tp) < p.Tvalid_in;
      aptr[i] = p.A + (aok[i] ? prow : 0) * (long long)p.lda + c4 * 4;
      abptr[i] = p.abits + (aok[i] ? prow : 0) * (long long)p.ld_abits;
    }
  }
  const float* bptr[B_F4];
#pragma unroll
  for (int i = 0; i < B_F4; ++i) {
    const int idx = tid + i * NTHR;
    const int r = idx >> 3, c4 = idx & 7;
    bptr[i] = p.Bw + (long long)((n0 + r) < p.N ? n0 + r : 0) * p.ldb + c4 * 4;
  }
  const long long tap_stride = (long long)p.N * p.ldb;

  auto load_a = [&](int chunk) {
    const int kc = chunk * W_BK;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      // unconditional (branch-free) loads from clamped addresses.  DIRECT: a row outside the matrix
      // only feeds pairs the epilogue zeroes; UNPOOL: masked at LDS-store time.
      ra[i] = *reinterpret_cast<const f32x4*>(aptr[i] + kc);
      if constexpr (LOADER == W_LOAD_UNPOOL) rbits[i] = abptr[i][kc >> 5];
    }
  };
  auto store_a = [&](int buf) {
    float* dst = As + buf * 2 * PLANE;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + i * NTHR;
      const int r = idx >> 3, c4 = idx & 7;
      if constexpr (LOADER == W_LOAD_DIRECT) {
        if (r < 2 * W_PR) *reinterpret_cast<f32x4*>(dst + (r & 1) * PLANE + (r >> 1) * W_LD + c4 * 4) = ra[i];
      } else {
        if (r < W_PR) {
          f32x4 e, o;
          const uint32_t nibv = rbits[i] >> ((c4 * 4) & 31);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool odd = (nibv >> q) & 1u;
            const float g = aok[i] ? ra[i][q] : 0.f;
            e[q] = odd ? 0.f : g;
            o[q] = odd ? g : 0.f;
          }
          *reinterpret_cast<f32x4*>(dst + r * W_LD + c4 * 4) = e;
          *reinterpret_cast<f32x4*>(dst + PLANE + r * W_LD + c4 * 4) = o;
        }
      }
    }
  };
  auto load_b = [&](f32x4 (&rb)[B_F4], int step) {
    const long long off = (long long)(step & 3) * tap_stride + (step >> 2) * W_BK;
#pragma unroll
    for (int i = 0; i < B_F4; ++i)          // rows past N are clamped: they only feed columns never stored
      rb[i] = *reinterpret_cast<const f32x4*>(bptr[i] + off);
  };
  auto store_b = [&](const f32x4 (&rb)[B_F4], int buf) {
    float* dst = Bs + buf * W_BN * W_LD;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int idx = tid + i * NTHR;
      const int r = idx >> 3, c4 = idx & 7;
      *reinterpret_cast<f32x4*>(dst + r * W_LD + c4 * 4) = rb[i];
    }
  };

  // fragment sets: x, y are the two staged rows whose sum / difference is (B^T d)_i
  f32x4 fx0[MI], fy0[MI], fb0[2], fx1[MI], fy1[MI], fb1[2];
  const int a_lane = (wm * (32 * MI) + lr) * W_LD + lh * 4;
  const int b_lane = (wn * 64 + lr) * W_LD + lh * 4;
  auto load_frag = [&](auto I, f32x4 (&fx)[MI], f32x4 (&fy)[MI], f32x4 (&fb)[2], int abuf, int bbuf, int kk) {
    constexpr int i = decltype(I)::value;
    // i = 0: E[p] - E[p+1]   i = 1: O[p] + E[p+1]   i = 2: E[p+1] - O[p]   i = 3: O[p] - O[p+1]
    constexpr int xo = (i == 0) ? 0 : ((i == 2) ? W_LD : PLANE);
    constexpr int yo = (i == 0 || i == 1) ? W_LD : ((i == 2) ? PLANE : PLANE + W_LD);
    const float* a_s = As + abuf * 2 * PLANE + a_lane + kk * 8;
    const float* b_s = Bs + bbuf * W_BN * W_LD + b_lane + kk * 8;
#pragma unroll
    for (int m = 0; m < MI; ++m) {
      fx[m] = *reinterpret_cast<const f32x4*>(a_s + m * 32 * W_LD + xo);
      fy[m] = *reinterpret_cast<const f32x4*>(a_s + m * 32 * W_LD + yo);
    }
    fb[0] = *reinterpret_cast<const f32x4*>(b_s);
    fb[1] = *reinterpret_cast<const f32x4*>(b_s + 32 * W_LD);
  };
  auto mfma_group = [&](auto I, const f32x4 (&fx)[MI], const f32x4 (&fy)[MI], const f32x4 (&fb)[2]) {
    constexpr int i = decltype(I)::value;
    f32x4 a[MI];
#pragma unroll
    for (int m = 0; m < MI; ++m) a[m] = (i == 1) ? (fx[m] + fy[m]) : (fx[m] - fy[m]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        acc[i][m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q], fb[0][q], acc[i][m][0], 0, 0, 0);
        acc[i][m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q], fb[1][q], acc[i][m][1], 0, 0, 0);
      }
  };

  // One K-step (transform i of one chunk).  LAST = the final chunk: no further A chunk, B loads /
  // stores only while steps remain - resolved at compile time, so the loop body has no branches.
  auto kstep = [&](auto I, auto LAST, int s, f32x4 (&rb_ld)[B_F4], const f32x4 (&rb_st)[B_F4]) {
    constexpr int i = decltype(I)::value;
    constexpr bool last = decltype(LAST)::value;
    using Prev = std::integral_constant<int, (i + 3) & 3>;
    const int chunk = s >> 2;
    const int abuf = chunk & 1, bbuf = s & 1;
    load_frag(I, fx0, fy0, fb0, abuf, bbuf, 0);
    if constexpr (!last || i < 2) load_b(rb_ld, s + 2);
    if constexpr (!last && i == 0) load_a(chunk + 1);
    W_SCHED_FENCE();
    mfma_group(Prev{}, fx1, fy1, fb1);                   // k-group 3 of the previous step (registers)
    load_frag(I, fx1, fy1, fb1, abuf, bbuf, 1);
    W_SCHED_FENCE();
    mfma_group(I, fx0, fy0, fb0);
    load_frag(I, fx0, fy0, fb0, abuf, bbuf, 2);
    W_SCHED_FENCE();
    mfma_group(I, fx1, fy1, fb1);
    if constexpr (!last || i < 3) store_b(rb_st, bbuf ^ 1);
    if constexpr (!last && i == 3) store_a(abuf ^ 1);
    load_frag(I, fx1, fy1, fb1, abuf, bbuf, 3);
    W_SCHED_FENCE();
    mfma_group(I, fx0, fy0, fb0);
    __syncthreads();
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using Mid = std::false_type;
  using Last = std::true_type;

  // the carried group of the (non-existent) step -1: zero operands, adds nothing
#pragma unroll
  for (int m = 0; m < MI; ++m) fx1[m] = fy1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  fb1[0] = fb1[1] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_a(0);
  load_b(rbP, 0);
  store_a(0);
  store_b(rbP, 0);
  load_b(rbQ, 1);
  __syncthreads();
  int s = 0;
  for (; s + 4 < nsteps; s += 4) {
    kstep(I0{}, Mid{}, s, rbP, rbQ);
    kstep(I1{}, Mid{}, s + 1, rbQ, rbP);
    kstep(I2{}, Mid{}, s + 2, rbP, rbQ);
    kstep(I3{}, Mid{}, s + 3, rbQ, rbP);
  }
  kstep(I0{}, Last{}, s, rbP, rbQ);
  kstep(I1{}, Last{}, s + 1, rbQ, rbP);
  kstep(I2{}, Last{}, s + 2, rbP, rbQ);
  kstep(I3{}, Last{}, s + 3, rbQ, rbP);
  mfma_group(I3{}, fx1, fy1, fb1);

  // ---- epilogue: y0 = m0 + m1 + m2, y1 = m1 - m2 - m3 per (pair, column) ----
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const long long P0 = (R0 >> 1) + wm * (32 * MI) + mi * 32 + 4 * lh;       // pair of accumulator element e = 0
    const int t0 = (int)((2 * P0) % p.Tp);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int colbase = n0 + wn * 64 + ni * 32;
      const int col = colbase + lr;
      const bool colok = col < p.N;
      if constexpr (EPI == W_EPI_POOL) {
        const float bv = (colok && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int po = (e & 3) + 8 * (e >> 2);
          const long long P = P0 + po;
          const float m1 = acc[1][mi][ni][e], m2 = acc[2][mi][ni][e];
          const float y0 = lrelu((acc[0][mi][ni][e] + m1) + m2 + bv, p.slope);
          const float y1 = lrelu((m1 - m2) - acc[3][mi][ni][e] + bv, p.slope);
          const bool rowok = 2 * P < p.M;
          const bool valid = rowok && ((t0 + 2 * po) % p.Tp) < p.Tvalid;
          const bool sel = valid && colok && (y1 > y0);
          const float o = valid ? (sel ? y1 : y0) : 0.f;
          if (rowok && colok) p.out[P * (long long)p.ldo + col] = o;
          const unsigned long long m = __ballot(sel);
          const unsigned long long ms = __ballot(o > 0.f);
          if (lr == 0 && rowok && colbase < p.N) {
            p.obits[P * (long long)p.ld_obits + (colbase >> 5)] = (uint32_t)(m >> (32 * lh));
            if (p.osign != nullptr) p.osign[P * (long long)p.ld_obits + (colbase >> 5)] = (uint32_t)(ms >> (32 * lh));
          }
        }
      } else if constexpr (EPI == W_EPI_C1W) {
        c1w_acc ca;
        ca.clear();
        c1w_cursor cur;
        cur.init(p, 2 * P0, colbase);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int po = (e & 3) + 8 * (e >> 2);
          if (e > 0) cur.advance(p, (e & 3) ? 2 : 10);       // pair offsets 0,1,2,3, 8,.. -> row steps 2,2,2,10
          const long long R = 2 * (P0 + po);
          const float m1 = acc[1][mi][ni][e], m2 = acc[2][mi][ni][e];
          const float v0 = (acc[0][mi][ni][e] + m1) + m2;
          const float v1 = (m1 - m2) - acc[3][mi][ni][e];
          if (R < p.M && colok) {                            // Tp is even: the pair stays inside one sequence
            if (cur.t < p.Tvalid) c1w_row(ca, p, cur, 0, v0, lr);
            if (cur.t + 1 < p.Tvalid) c1w_row(ca, p, cur, 1, v1, lr);
          }
        }
        static_assert(EPI != W_EPI_C1W || MI == 1, "the fused conv1 weight gradient assumes one row tile per wave");
        c1w_reduce_store<4, 128>(p, lds, ca, wm, wn * 64 + ni * 32 + lr, lh, tm, col, colok);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int po = (e & 3) + 8 * (e >> 2);
          const long long R = 2 * (P0 + po);
          const float m1 = acc[1][mi][ni][e], m2 = acc[2][mi][ni][e];
          float v0 = (acc[0][mi][ni][e] + m1) + m2;
          float v1 = (m1 - m2) - acc[3][mi][ni][e];
          if (R < p.M && colok) {                            // M is even: the pair shares validity
            bool pos0, pos1;
            if (p.auxbits != nullptr) {
              pos0 = (p.auxbits[R * (long long)p.ld_auxbits + (colbase >> 5)] >> lr) & 1u;
              pos1 = (p.auxbits[(R + 1) * (long long)p.ld_auxbits + (colbase >> 5)] >> lr) & 1u;
            } else {
              pos0 = p.aux[R * (long long)p.ldaux + col] > 0.f;
              pos1 = p.aux[(R + 1) * (long long)p.ldaux + col] > 0.f;
            }
            v0 = pos0 ? v0 : v0 * p.slope;
            v1 = pos1 ? v1 : v1 * p.slope;
            p.out[R * (long long)p.ldo + col] = v0;
            p.out[(R + 1) * (long long)p.ldo + col] = v1;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// F(4,3): 6 channel contractions per 4 conv rows (two pool pairs) - half the MFMA work of the
// direct form.  Rows 4Q .. 4Q+5 feed quad Q:
//   V0 = 4d0 - 5d2 + d4           V1 = -4d1 - 4d2 + d3 + d4      V2 = 4d1 - 4d2 - d3 + d4
//   V3 = -2d1 - d2 + 2d3 + d4     V4 = 2d1 - d2 - 2d3 + d4       V5 = 4d1 - 5d3 + d5
//   U  = G g,  G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   y0 = M0+M1+M2+M3+M4   y1 = (M1-M2) + 2(M3-M4)   y2 = (M1+M2) + 4(M3+M4)   y3 = (M1-M2) + 8(M3-M4) + M5
// (fp32 error of this form measured 1.5x that of the direct convolution on the conv2 shape).
// Workgroup: 8 waves, 4 (quads) x 2 (columns); wave tile 32 quads x 32 columns x 6 transforms = 96
// accumulator registers; block tile 128 quads (512 conv rows) x 64 columns; a K-step is a 16-deep
// channel chunk carrying all six transforms (48 MFMAs per wave between barriers).  LDS keeps the
// staged rows in four planes (row mod 4), 80-byte row stride: conflict-free ds_read_b128.
// ------------------------------------------------------------------------------------------
#ifndef W4_PIN
#define W4_PIN 1          // 1: pin the weight-tile loads at the top of a K-step (conv2 fwd 51.5 -> 47.9 ms);
                          // 2: the A loads too (spills); 0: leave both to the scheduler
#endif
constexpr int W4_BQ = 128, W4_BN = 64, W4_BK = 16, W4_LD = W4_BK + 4, W4_QR = W4_BQ + 1;

__global__ void wino43_weights_kernel(const float* __restrict__ w, float* __restrict__ fwd, float* __restrict__ dgr,
                                      int O, int I, int ld_f, int ld_d) {
  const long long n_f = (long long)O * ld_f, n_d = (long long)I * ld_d;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  auto emit = [](float* dst, long long n, long long at, float g0, float g1, float g2) {
    const float s = g0 + g2;
    dst[at] = 0.25f * g0;
    dst[n + at] = (-1.f / 6.f) * (s + g1);
    dst[2 * n + at] = (-1.f / 6.f) * (s - g1);
    dst[3 * n + at] = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
    dst[4 * n + at] = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
    dst[5 * n + at] = g2;
  };
  if (fwd != nullptr && idx < n_f) {
    const int o = (int)(idx / ld_f), i = (int)(idx % ld_f);
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (i < I) {
      const float* s = w + ((long long)o * I + i) * 3;
      g0 = s[0], g1 = s[1], g2 = s[2];
    }
    emit(fwd, n_f, idx, g0, g1, g2);
  }
  if (dgr != nullptr && idx < n_d) {
    const int i = (int)(idx / ld_d), o = (int)(idx % ld_d);
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (o < O) {
      const float* s = w + ((long long)o * I + i) * 3;
      g0 = s[2], g1 = s[1], g2 = s[0];          // flipped taps
    }
    emit(dgr, n_d, idx, g0, g1, g2);
  }
}

// taps-wide filter (O, I, taps) -> [6][O][nseg I]: segment s (columns s I ..) = F(4,3) transform of taps 3s..3s+2
__global__ void wino43_weights7_kernel(const float* __restrict__ w, float* __restrict__ fwd, int O, int I, int taps, int nseg) {
  const long long n = (long long)O * nseg * I;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % ((long long)nseg * I));
  const long long o = idx / ((long long)nseg * I);
  const int seg = c / I, i = c % I;
  const float* s = w + (o * I + i) * taps + 3 * seg;
  const float g0 = s[0], g1 = (3 * seg + 1 < taps) ? s[1] : 0.f, g2 = (3 * seg + 2 < taps) ? s[2] : 0.f;
  const float sm = g0 + g2;
  fwd[idx] = 0.25f * g0;
  fwd[n + idx] = (-1.f / 6.f) * (sm + g1);
  fwd[2 * n + idx] = (-1.f / 6.f) * (sm - g1);
  fwd[3 * n + idx] = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
  fwd[4 * n + idx] = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
  fwd[5 * n + idx] = g2;
}

// NSEG > 1: a (3 NSEG - 2 .. 3 NSEG)-tap convolution as NSEG three-tap segments accumulated in the same
// six products - segment s reads the input rows shifted by 3 s and the weight columns [s K, (s+1) K).
template <int LOADER, int EPI, int NSEG = 1>
__global__ __launch_bounds__(512, 2) void wino43_nt_kernel(const tl_nt_params p) {
  constexpr int NTHR = 512;
  constexpr int PLANE = W4_QR * W4_LD;
  constexpr int ABUF = 4 * PLANE;
  constexpr int BBUF = 6 * W4_BN * W4_LD;
  constexpr int AROWS = 4 * W4_BQ + 2;                   // staged input rows
  constexpr int A_F4 = (LOADER == W_LOAD_DIRECT) ? ((AROWS * 4 + NTHR - 1) / NTHR) : ((AROWS / 2 * 4 + NTHR - 1) / NTHR);
  constexpr int B_F4 = 6 * W4_BN * 4 / NTHR;             // 3

  __shared__ __attribute__((aligned(16))) float lds[2 * ABUF + 2 * BBUF];
  float* As = lds;                                        // [2][4 planes][W4_QR][W4_LD]
  float* Bs = lds + 2 * ABUF;                             // [2][6][W4_BN][W4_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const int ntn = (p.N + W4_BN - 1) / W4_BN;
  const long long ntm = (p.M + 4 * W4_BQ - 1) / (4 * W4_BQ);
  const long long nwg = ntm * ntn;
  long long bid = blockIdx.x;
  {
    const long long q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const long long tm = bid / ntn;
  const int tn = (int)(bid % ntn);
  const long long R0 = tm * (4 * W4_BQ);
  const int n0 = tn * W4_BN;
  const int cps = p.K / W4_BK;                            // host-checked: K % 16 == 0, K >= 16
  const int nsteps = NSEG * cps;

  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  f32x4 raP[A_F4], raQ[A_F4];   // the A chunk streams from HBM: prefetched two K-steps ahead
  uint32_t rbitsP[A_F4], rbitsQ[A_F4];
  f32x4 rb[B_F4];               // the weight tile comes from L2 / Infinity Cache: one K-step of prefetch
  (void)rbitsP; (void)rbitsQ;

  const long long Abase = R0 + p.row_shift;               // first staged input row (even)
  const float* aptr[A_F4];
  const uint32_t* abptr[A_F4];
  bool aok[A_F4];
  (void)abptr;
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int idx = tid + i * NTHR;
    const int r = idx >> 2, c4 = idx & 3;
    if constexpr (LOADER == W_LOAD_DIRECT) {
      const long long row = Abase + r;
      aok[i] = r < AROWS && row >= 0 && row < p.A_rows;
      aptr[i] = p.A + (aok[i] ? row : 0) * (long long)p.lda + c4 * 4;
      abptr[i] = nullptr;
    } else {
      const long long prow = (Abase >> 1) + r;
      aok[i] = r < AROWS / 2 && prow >= 0 && prow < p.A_rows && (int)((2 * prow) % p.Tp) < p.Tvalid_in;
      aptr[i] = p.A + (aok[i] ? prow : 0) * (long long)p.lda + c4 * 4;
      abptr[i] = p.abits + (aok[i] ? prow : 0) * (long long)p.ld_abits;
    }
  }
  const float* bptr[B_F4];
  const long long tap_stride = (long long)p.N * p.ldb;
#pragma unroll
  for (int i = 0; i < B_F4; ++i) {
    const int idx = tid + i * NTHR;
    const int it = idx >> 8, r = (idx >> 2) & 63, c4 = idx & 3;
    bptr[i] = p.Bw + it * tap_stride + (long long)((n0 + r) < p.N ? n0 + r : 0) * p.ldb + c4 * 4;
  }

  auto load_a = [&](f32x4 (&ra)[A_F4], uint32_t (&rbits)[A_F4], int step) {
    const auto kc = [&] {
      if constexpr (NSEG > 1) {
        static_assert(NSEG <= 3 && LOADER == W_LOAD_DIRECT, "segments: direct loader, at most three");
        const int seg = (step >= cps) + (step >= 2 * cps);
        return step * W4_BK + seg * (3LL * p.lda - p.K);  // next three input rows, channel 0
      } else {
        return step * W4_BK;
      }
    }();
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {                      // branch-free: see wino_nt_kernel
      ra[i] = *reinterpret_cast<const f32x4*>(aptr[i] + kc);
      if constexpr (LOADER == W_LOAD_UNPOOL) rbits[i] = abptr[i][kc >> 5];
    }
  };
  auto store_a = [&](const f32x4 (&ra)[A_F4], const uint32_t (&rbits)[A_F4], int buf, int step) {
    float* dst = As + buf * ABUF;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + i * NTHR;
      const int r = idx >> 2, c4 = idx & 3;
      if constexpr (LOADER == W_LOAD_DIRECT) {
        if (r < AROWS) *reinterpret_cast<f32x4*>(dst + (r & 3) * PLANE + (r >> 2) * W4_LD + c4 * 4) = ra[i];
      } else {
        if (r < AROWS / 2) {
          f32x4 e, o;
          const uint32_t nibv = rbits[i] >> ((step * W4_BK + c4 * 4) & 31);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool odd = (nibv >> q) & 1u;
            const float g = aok[i] ? ra[i][q] : 0.f;
            e[q] = odd ? 0.f : g;
            o[q] = odd ? g : 0.f;
          }
          float* d2 = dst + ((r & 1) * 2) * PLANE + (r >> 1) * W4_LD + c4 * 4;   // pair r = staged rows 2r, 2r+1
          *reinterpret_cast<f32x4*>(d2) = e;
          *reinterpret_cast<f32x4*>(d2 + PLANE) = o;
        }
      }
    }
  };
  auto load_b = [&](f32x4 (&rb)[B_F4], int step) {
    const int kc = step * W4_BK;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bptr[i] + kc);
  };
  auto store_b = [&](const f32x4 (&rb)[B_F4], int buf) {
    float* dst = Bs + buf * BBUF;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int idx = tid + i * NTHR;
      const int it = idx >> 8, r = (idx >> 2) & 63, c4 = idx & 3;
      *reinterpret_cast<f32x4*>(dst + (it * W4_BN + r) * W4_LD + c4 * 4) = rb[i];
    }
  };

  const int a_lane = (wm * 32 + lr) * W4_LD + lh * 4;
  const int b_lane = (wn * 32 + lr) * W4_LD + lh * 4;
  // Software pipeline of one K-step (two 8-deep k-groups g0, g1; transforms split in alpha = {1..4},
  // whose operands are rows d1..d4, and beta = {0, 5}): the beta MFMAs of g1 are carried in registers
  // across the barrier and run while the alpha fragments of the next step are read from LDS.
  f32x4 cv0 = {0.f, 0.f, 0.f, 0.f}, cv5 = cv0, cu0 = cv0, cu5 = cv0;
  auto rd = [&](const float* ptr) { return *reinterpret_cast<const f32x4*>(ptr); };
  auto alpha_mfma = [&](const f32x4 (&v)[4], const f32x4 (&u)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i][q], u[i][q], acc[i + 1], 0, 0, 0);
  };
  auto beta_mfma = [&](const f32x4& v0, const f32x4& u0, const f32x4& v5, const f32x4& u5) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[q], u0[q], acc[0], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(v5[q], u5[q], acc[5], 0, 0, 0);
    }
  };
  auto kstep = [&](auto TAIL, int s, f32x4 (&ra_ld)[A_F4], uint32_t (&rbits_ld)[A_F4], const f32x4 (&ra_st)[A_F4],
                   const uint32_t (&rbits_st)[A_F4]) {
    constexpr bool tail = decltype(TAIL)::value;
    const int buf = s & 1;
    const float* a_s = As + buf * ABUF + a_lane;
    const float* b_s = Bs + buf * BBUF + b_lane;
    constexpr int US = W4_BN * W4_LD;
    // alpha(g0) reads
    f32x4 d1 = rd(a_s + PLANE), d2 = rd(a_s + 2 * PLANE), d3 = rd(a_s + 3 * PLANE), d4 = rd(a_s + W4_LD);
    f32x4 u[4] = {rd(b_s + US), rd(b_s + 2 * US), rd(b_s + 3 * US), rd(b_s + 4 * US)};
    if (!tail || s + 1 < nsteps) load_b(rb, s + 1);
#if W4_PIN == 1
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (!tail || s + 2 < nsteps) load_a(ra_ld, rbits_ld, s + 2);
    // keep the global loads here: left alone, the scheduler sinks them next to the LDS stores that
    // consume them (shorter live ranges), which turns the prefetch into an exposed L2 round trip
#if W4_PIN == 2
    __builtin_amdgcn_sched_barrier(0);
#endif
    beta_mfma(cv0, cu0, cv5, cu5);                        // carried from the previous step
    // beta(g0) reads
    f32x4 d0 = rd(a_s), d5 = rd(a_s + PLANE + W4_LD), u0 = rd(b_s), u5 = rd(b_s + 5 * US);
    f32x4 v[4];
    {
      const f32x4 s1 = d4 - 4.f * d2, s2 = d3 - 4.f * d1, s3 = d4 - d2, t = d3 - d1;
      v[0] = s1 + s2;
      v[1] = s1 - s2;
      v[2] = s3 + 2.f * t;
      v[3] = s3 - 2.f * t;
    }
    f32x4 t0 = d4 - 5.f * d2, t5 = 4.f * d1 - 5.f * d3;
    alpha_mfma(v, u);
    f32x4 v0 = 4.f * d0 + t0, v5 = t5 + d5;
    // alpha(g1) reads
    d1 = rd(a_s + PLANE + 8), d2 = rd(a_s + 2 * PLANE + 8), d3 = rd(a_s + 3 * PLANE + 8), d4 = rd(a_s + W4_LD + 8);
    f32x4 w[4] = {rd(b_s + US + 8), rd(b_s + 2 * US + 8), rd(b_s + 3 * US + 8), rd(b_s + 4 * US + 8)};
    beta_mfma(v0, u0, v5, u5);
    if (!tail || s + 1 < nsteps) store_b(rb, buf ^ 1);
    // beta(g1) reads
    d0 = rd(a_s + 8), d5 = rd(a_s + PLANE + W4_LD + 8), cu0 = rd(b_s + 8), cu5 = rd(b_s + 5 * US + 8);
    {
      const f32x4 s1 = d4 - 4.f * d2, s2 = d3 - 4.f * d1, s3 = d4 - d2, t = d3 - d1;
      v[0] = s1 + s2;
      v[1] = s1 - s2;
      v[2] = s3 + 2.f * t;
      v[3] = s3 - 2.f * t;
    }
    t0 = d4 - 5.f * d2, t5 = 4.f * d1 - 5.f * d3;
    alpha_mfma(v, w);
    cv0 = 4.f * d0 + t0, cv5 = t5 + d5;
    if (!tail || s + 1 < nsteps) store_a(ra_st, rbits_st, buf ^ 1, s + 1);
    __syncthreads();
  };
  using Mid = std::false_type;
  using Tail = std::true_type;

  load_a(raP, rbitsP, 0);
  load_b(rb, 0);
  store_a(raP, rbitsP, 0, 0);
  store_b(rb, 0);
  if (nsteps > 1) load_a(raQ, rbitsQ, 1);
  __syncthreads();
  int s = 0;
  for (; s + 3 < nsteps; s += 2) {
    kstep(Mid{}, s, raP, rbitsP, raQ, rbitsQ);
    kstep(Mid{}, s + 1, raQ, rbitsQ, raP, rbitsP);
  }
  for (; s < nsteps; s += 2) {
    kstep(Tail{}, s, raP, rbitsP, raQ, rbitsQ);
    if (s + 1 < nsteps) kstep(Tail{}, s + 1, raQ, rbitsQ, raP, rbitsP);
  }
  beta_mfma(cv0, cu0, cv5, cu5);

  wino43_epilogue<EPI>(p, acc, lds, R0, n0, wm, wn, lr, lh, tm);
}

// ------------------------------------------------------------------------------------------
// TN form (weight gradient): slab[z][i][m][n] = sum_{pairs in split z} (B^T d)_i[m] * (A dy)_i[n]
// with d = the activation rows 2P..2P+3 (A operand, C_in) and dy = the un-pooled dZ rows 2P, 2P+1
// (B operand, C_out): dy0 = G (bit clear), dy1 = G (bit set).  Stored transforms: i = 3 uses +dy1
// (the finalize step flips its sign).  Tile 128 (C_in) x 64 (C_out), wave tile 64 x 32 x 4
// transforms = 128 accumulator registers, 2 workgroups per CU; a K-step is 32 rows = 16 pairs.
// ------------------------------------------------------------------------------------------
#ifndef WT_BATCH_XF
#define WT_BATCH_XF 1      // 1: transforms of a k-step batched ahead of its MFMAs (conv2 wgrad 57.3 -> 54.2 ms)
#endif
constexpr int WT_BN = 64, WT_LDA = 128 + 4, WT_LDB = WT_BN + 4, WT_AR = W_BK + 2;

__global__ __launch_bounds__(256, 2) void wino_tn_kernel(const tl_tn_params p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * WT_AR * WT_LDA + 2 * W_BK * WT_LDB];
  float* As = lds;                                 // [2][34 rows][132]
  float* Bs = lds + 2 * WT_AR * WT_LDA;            // [2][32 rows (pair r -> rows 2r: dy0, 2r+1: dy1)][68]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int ntm = (p.Mdim + 127) / 128, ntn = (p.Ndim + WT_BN - 1) / WT_BN;
  const long long tiles = (long long)ntm * ntn;
  const long long nwg = tiles * p.splitk;
  long long bid = (long long)blockIdx.y * gridDim.x + blockIdx.x;
  {
    const long long q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int z = (int)(bid / tiles);
  const int tt = (int)(bid % tiles);
  const int m0 = (tt / ntn) * 128, n0 = (tt % ntn) * WT_BN;

  const long long ksteps_all = (p.Krows + W_BK - 1) / W_BK;
  const long long per = (ksteps_all + p.splitk - 1) / p.splitk;
  const long long ks_begin = z * per;
  long long ks_end = ks_begin + per;
  if (ks_end > ksteps_all) ks_end = ksteps_all;
  const long long nsteps = ks_end > ks_begin ? ks_end - ks_begin : 0;

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  f32x4 raP[5], raQ[5], rbP, rbQ;
  uint32_t rnP, rnQ;

  // Global loads are unconditional (no divergent branches in the K loop): row / column indices are
  // clamped into the matrices.  A clamped activation row only ever meets a zero dZ pair (rows past
  // the end belong to invalid time steps); an invalid dZ pair is zeroed when it is written to LDS.
  const int a_last = (int)(p.A_rows < p.Krows + 2 ? p.A_rows : p.Krows + 2) - 1;
  const int b_lim = (int)(2 * p.B_rows < p.Krows ? 2 * p.B_rows : p.Krows);
  const int kbase = (int)(ks_begin * W_BK);
  const int acol = m0 + (((m0 + (tid & 31) * 4) < p.Mdim) ? (tid & 31) * 4 : 0);
  int a_row = kbase + (tid >> 5);                  // thread t stages rows (t >> 5) + 8 i, i < 5
  const int ncol = n0 + (tid & 15) * 4;
  const bool bnok = ncol < p.Ndim;
  const int ncolc = bnok ? ncol : n0;
  const int dstep = W_BK % p.Tp;
  int brow = kbase + 2 * (tid >> 4);               // conv row of this thread's pair (even)
  int bt = brow % p.Tp;
  const int b_last = (int)p.B_rows - 1;

  // FAST (compile time): every row of the chunk is inside both matrices, so the addresses are a
  // wave-uniform base (SGPRs, advanced per step) plus a per-thread 32-bit offset fixed for the whole
  // kernel - no per-load clamping or 64-bit VALU arithmetic.  The last three chunks of a split and any
  // launch whose matrices are shorter than the reduction take the clamped path.
  long long ld_row0 = kbase;                       // first row of the chunk the next load_tiles fetches
  const unsigned a_toff = (unsigned)((tid >> 5) * p.lda + (acol - m0));
  const unsigned b_toff = (unsigned)((tid >> 4) * p.ldb + (ncolc - n0));
  const unsigned bb_toff = (unsigned)((tid >> 4) * p.ld_bbits + (ncolc >> 5));
  auto load_tiles = [&](auto FAST, f32x4 (&ra)[5], f32x4& rb, uint32_t& rn) {
    if constexpr (decltype(FAST)::value) {
      const float* au = p.A + ld_row0 * (long long)p.lda + m0;
#pragma unroll
      for (int i = 0; i < 5; ++i) ra[i] = *reinterpret_cast<const f32x4*>(au + (long long)(8 * i) * p.lda + a_toff);
      const long long pr0 = ld_row0 >> 1;
      rb = *reinterpret_cast<const f32x4*>(p.B + pr0 * (long long)p.ldb + n0 + b_toff);
      rn = ((p.bbits[pr0 * (long long)p.ld_bbits + bb_toff] >> (ncolc & 31)) & 0xFu) |
           ((bnok && bt < p.Tvalid) ? 0x10u : 0u);
    } else {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int row = min(a_row + 8 * i, a_last);
        ra[i] = *reinterpret_cast<const f32x4*>(p.A + (long long)row * p.lda + acol);
      }
      const int pr = min(brow >> 1, b_last);
      rb = *reinterpret_cast<const f32x4*>(p.B + (long long)pr * p.ldb + ncolc);
      // arg-max nibble of the 4 columns in bits 0..3, "pair is valid" in bit 4
      rn = ((p.bbits[(long long)pr * p.ld_bbits + (ncolc >> 5)] >> (ncolc & 31)) & 0xFu) |
           ((bnok && brow < b_lim && bt < p.Tvalid) ? 0x10u : 0u);
    }
    ld_row0 += W_BK;
    a_row += W_BK;
    brow += W_BK;
    bt += dstep;
    if (bt >= p.Tp) bt -= p.Tp;
  };
  auto store_tiles = [&](const f32x4 (&ra)[5], const f32x4& rb, const uint32_t& rn, int buf) {
    float* da = As + buf * WT_AR * WT_LDA;
    float* db = Bs + buf * W_BK * WT_LDB;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = tid + i * 256;
      const int r = idx >> 5, c4 = idx & 31;
      if (r < WT_AR) *reinterpret_cast<f32x4*>(da + r * WT_LDA + c4 * 4) = ra[i];
    }
    const int r = tid >> 4, c4 = tid & 15;
    f32x4 e, o;
    const bool rv = (rn & 0x10u) != 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool odd = (rn >> q) & 1u;
      const float g = rv ? rb[q] : 0.f;
      e[q] = odd ? 0.f : g;
      o[q] = odd ? g : 0.f;
    }
    *reinterpret_cast<f32x4*>(db + (2 * r) * WT_LDB + c4 * 4) = e;
    *reinterpret_cast<f32x4*>(db + (2 * r + 1) * WT_LDB + c4 * 4) = o;
  };

  // fragments of one MFMA k-step (2 pairs): lane (lr, lh) of k-step q works on pair 2q + lh of the
  // chunk: activation rows 4q + 2lh .. + 3, dZ rows 4q + 2lh (dy0) and + 1 (dy1).
  float fa0[4][2], fb0[2], fa1[4][2], fb1[2];
  auto load_frag = [&](float (&fa)[4][2], float (&fb)[2], int buf, int q) {
    const float* a_s = As + buf * WT_AR * WT_LDA + (q * 4 + 2 * lh) * WT_LDA + wm * 64 + lr;
    const float* b_s = Bs + buf * W_BK * WT_LDB + (q * 4 + 2 * lh) * WT_LDB + wn * 32 + lr;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      fa[d][0] = a_s[d * WT_LDA];
      fa[d][1] = a_s[d * WT_LDA + 32];
    }
    fb[0] = b_s[0];
    fb[1] = b_s[WT_LDB];
  };
  auto mfma_group = [&](const float (&fa)[4][2], const float (&fb)[2]) {
    const float e = fb[0], o = fb[1];
    const float b1 = e + o, b2 = e - o;
#if WT_BATCH_XF
    // all eight transformed operands first, then the eight MFMAs: no VALU -> MFMA read-after-write
    // stall (s_nop) in front of every MFMA
    float t[4][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const float d0 = fa[0][mi], d1 = fa[1][mi], d2 = fa[2][mi], d3 = fa[3][mi];
      t[0][mi] = d0 - d2;
      t[1][mi] = d1 + d2;
      t[2][mi] = d2 - d1;
      t[3][mi] = d1 - d3;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      acc[0][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[0][mi], e, acc[0][mi], 0, 0, 0);
      acc[1][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[1][mi], b1, acc[1][mi], 0, 0, 0);
      acc[2][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[2][mi], b2, acc[2][mi], 0, 0, 0);
      acc[3][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[3][mi], o, acc[3][mi], 0, 0, 0);
    }
#else
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const float d0 = fa[0][mi], d1 = fa[1][mi], d2 = fa[2][mi], d3 = fa[3][mi];
      acc[0][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0 - d2, e, acc[0][mi], 0, 0, 0);
      acc[1][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1 + d2, b1, acc[1][mi], 0, 0, 0);
      acc[2][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(d2 - d1, b2, acc[2][mi], 0, 0, 0);
      acc[3][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1 - d3, o, acc[3][mi], 0, 0, 0);
    }
#endif
  };
  // TAIL = false: steady state, the step prefetches the tiles of step s + 2 and writes those of
  // s + 1 to LDS unconditionally (no branch in the loop body); TAIL = true: the last <= 3 steps.
  auto kstep = [&](auto TAIL, long long s, f32x4 (&ra_ld)[5], f32x4& rb_ld, uint32_t& rn_ld,
                   const f32x4 (&ra_st)[5], const f32x4& rb_st, const uint32_t& rn_st) {
    constexpr bool tail = decltype(TAIL)::value;
    const int buf = (int)(s & 1);
    load_frag(fa0, fb0, buf, 0);
    if constexpr (!tail) load_tiles(std::true_type{}, ra_ld, rb_ld, rn_ld);
    else if (s + 2 < nsteps) load_tiles(std::false_type{}, ra_ld, rb_ld, rn_ld);
    mfma_group(fa1, fb1);                          // last k-step of the previous step (registers)
    load_frag(fa1, fb1, buf, 1);
    mfma_group(fa0, fb0);
    load_frag(fa0, fb0, buf, 2);
    mfma_group(fa1, fb1);
    load_frag(fa1, fb1, buf, 3);
    mfma_group(fa0, fb0);
    load_frag(fa0, fb0, buf, 4);
    mfma_group(fa1, fb1);
    if (!tail || s + 1 < nsteps) store_tiles(ra_st, rb_st, rn_st, buf ^ 1);
    load_frag(fa1, fb1, buf, 5);
    mfma_group(fa0, fb0);
    load_frag(fa0, fb0, buf, 6);
    mfma_group(fa1, fb1);
    load_frag(fa1, fb1, buf, 7);
    mfma_group(fa0, fb0);
    __syncthreads();
  };
  using Y = std::true_type;
  using N = std::false_type;

#pragma unroll
  for (int d = 0; d < 4; ++d) fa1[d][0] = fa1[d][1] = 0.f;       // carried group of step -1: adds nothing
  fb1[0] = fb1[1] = 0.f;

  if (nsteps > 0) {
    load_tiles(N{}, raP, rbP, rnP);
    store_tiles(raP, rbP, rnP, 0);
    if (nsteps > 1) load_tiles(N{}, raQ, rbQ, rnQ);
  }
  __syncthreads();
  long long s = 0;
  // steady state: the chunk fetched at step s is s + 2 <= nsteps - 4, at least 96 rows before the end
  const bool whole = p.A_rows >= p.Krows && 2 * p.B_rows >= p.Krows;
  if (whole)
    for (; s + 5 < nsteps; s += 2) {
      kstep(N{}, s, raP, rbP, rnP, raQ, rbQ, rnQ);
      kstep(N{}, s + 1, raQ, rbQ, rnQ, raP, rbP, rnP);
    }
  for (; s < nsteps; s += 2) {
    kstep(Y{}, s, raP, rbP, rnP, raQ, rbQ, rnQ);
    if (s + 1 < nsteps) kstep(Y{}, s + 1, raQ, rbQ, rnQ, raP, rbP, rnP);
  }
  mfma_group(fa1, fb1);

  float* out = p.slab + (long long)z * p.slab_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int col = n0 + wn * 32 + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < p.Mdim && col < p.Ndim) out[((long long)i * p.Mdim + m) * (long long)p.ldc + col] = acc[i][mi][e];
      }
    }
}

// dW (O, I, 3) = G^T M from the reduced transforms red[4][I][ld] (M_3 stored with flipped sign)
__global__ void wino_wgrad_finalize_kernel(const float* __restrict__ red, float* __restrict__ gw, int O, int I, int ld) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)O * I) return;
  const int i = (int)(idx / O), o = (int)(idx % O);
  const long long plane = (long long)I * ld;
  const float* s = red + (long long)i * ld + o;
  const float m0 = s[0], m1 = s[plane], m2 = s[2 * plane], m3 = -s[3 * plane];
  float* d = gw + ((long long)o * I + i) * 3;
  const float h = 0.5f * (m1 + m2);
  d[0] = m0 + h;
  d[1] = 0.5f * (m1 - m2);
  d[2] = h + m3;
}

}  // namespace tl

extern "C" int tl_wino_weights(const float* w, float* fwd, float* dgr, int O, int I, int ld_f, int ld_d, void* stream) {
  using namespace tl;
  TL_REQUIRE(w != nullptr && (fwd != nullptr || dgr != nullptr), "wino_weights: null pointer");
  TL_REQUIRE(O > 0 && I > 0, "wino_weights: bad sizes");
  TL_REQUIRE((fwd == nullptr || ld_f >= I) && (dgr == nullptr || ld_d >= O), "wino_weights: leading dimension too small");
  const long long nf = fwd ? (long long)O * ld_f : 0, nd = dgr ? (long long)I * ld_d : 0;
  const long long n = nf > nd ? nf : nd;
  hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, fwd,
                     dgr, O, I, ld_f, ld_d);
  return check_launch("wino_weights");
}

extern "C" int tl_conv3_wino_nt(const tl_nt_params* pp, void* stream) {
  using namespace tl;
  TL_REQUIRE(pp != nullptr, "wino_nt: null params");
  const tl_nt_params& p = *pp;
  TL_REQUIRE(p.A && p.Bw && (p.out || p.epilogue == W_EPI_C1W), "wino_nt: null A/Bw/out");
  TL_REQUIRE(p.J == 3, "wino_nt: the Winograd form is for 3-tap convolutions");
  TL_REQUIRE(p.M >= 0 && p.M % 2 == 0 && p.N > 0 && p.K > 0, "wino_nt: bad M/N/K %lld/%d/%d", (long long)p.M, p.N, p.K);
  TL_REQUIRE(p.K % W_BK == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0, "wino_nt: K %% 32, lda %% 4, ldb %% 4 must be 0");
  TL_REQUIRE(p.lda >= p.K && p.ldb >= p.K, "wino_nt: lda/ldb smaller than K");
  TL_REQUIRE(p.Tp > 0 && p.Tp % 2 == 0, "wino_nt: Tp must be positive and even");
  TL_REQUIRE(p.splitk <= 1, "wino_nt: no split-K");
  const long long nwg = ((p.M + 2 * W_BP - 1) / (2 * W_BP)) * ((p.N + W_BN - 1) / W_BN);
  if (nwg <= 0) return TL_OK;
  TL_REQUIRE(nwg < (1LL << 31), "wino_nt: grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (p.loader == W_LOAD_DIRECT && p.epilogue == W_EPI_POOL) {
    TL_REQUIRE(p.row_shift == 0, "wino_nt: forward needs row_shift 0");
    TL_REQUIRE(p.obits != nullptr && p.Tvalid % 2 == 0, "wino_nt: POOL needs obits and an even Tvalid");
    TL_REQUIRE(p.N % 32 == 0 && p.ld_obits * 32 >= p.N, "wino_nt: POOL needs N %% 32 == 0");
    hipLaunchKernelGGL((wino_nt_kernel<W_LOAD_DIRECT, W_EPI_POOL, W_MI>), dim3((unsigned)nwg), dim3(512 / W_MI), 0, st, p);
  } else if (p.loader == W_LOAD_UNPOOL && p.epilogue == W_EPI_MASK) {
    TL_REQUIRE(p.row_shift == -2, "wino_nt: input gradient needs row_shift -2");
    TL_REQUIRE(p.abits != nullptr && (p.aux != nullptr || p.auxbits != nullptr), "wino_nt: UNPOOL/MASK need abits and aux or auxbits");
    TL_REQUIRE(p.Tvalid_in % 2 == 0, "wino_nt: UNPOOL needs an even Tvalid_in");
    hipLaunchKernelGGL((wino_nt_kernel<W_LOAD_UNPOOL, W_EPI_MASK, W_MI>), dim3((unsigned)nwg), dim3(512 / W_MI), 0, st, p);
  } else if (p.loader == W_LOAD_UNPOOL && p.epilogue == W_EPI_C1W) {
    TL_REQUIRE(p.row_shift == -2 && p.abits != nullptr && p.Tvalid_in % 2 == 0, "wino_nt: input gradient needs row_shift -2, abits, even Tvalid_in");
    TL_REQUIRE(p.auxbits && p.c1x && p.c1bits && p.c1partial, "wino_nt: epilogue 4 needs auxbits, c1x, c1bits, c1partial");
    TL_REQUIRE(p.c1kt >= 1 && p.c1kt <= 3 && p.c1T >= 2 * p.Tvalid + 2, "wino_nt: epilogue 4: 1..3 taps, c1T >= 2*Tvalid + 2");
    hipLaunchKernelGGL((wino_nt_kernel<W_LOAD_UNPOOL, W_EPI_C1W, 1>), dim3((unsigned)nwg), dim3(512), 0, st, p);
  } else {
    set_error("wino_nt: unsupported loader/epilogue combination %d/%d", p.loader, p.epilogue);
    return TL_EINVAL;
  }
  return check_launch("wino_nt");
}

extern "C" int tl_wino43_weights(const float* w, float* fwd, float* dgr, int O, int I, int ld_f, int ld_d, void* stream) {
  using namespace tl;
  TL_REQUIRE(w != nullptr && (fwd != nullptr || dgr != nullptr), "wino43_weights: null pointer");
  TL_REQUIRE(O > 0 && I > 0, "wino43_weights: bad sizes");
  TL_REQUIRE((fwd == nullptr || ld_f >= I) && (dgr == nullptr || ld_d >= O), "wino43_weights: leading dimension too small");
  const long long nf = fwd ? (long long)O * ld_f : 0, nd = dgr ? (long long)I * ld_d : 0;
  const long long n = nf > nd ? nf : nd;
  hipLaunchKernelGGL(wino43_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, fwd,
                     dgr, O, I, ld_f, ld_d);
  return check_launch("wino43_weights");
}

extern "C" int tl_conv3_wino43_nt(const tl_nt_params* pp, void* stream) {
  using namespace tl;
  TL_REQUIRE(pp != nullptr, "wino43_nt: null params");
  const tl_nt_params& p = *pp;
  TL_REQUIRE(p.A && p.Bw && (p.out || p.epilogue == W_EPI_C1W), "wino43_nt: null A/Bw/out");
  TL_REQUIRE(p.J == 3, "wino43_nt: the Winograd form is for 3-tap convolutions");
  TL_REQUIRE(p.M >= 0 && p.M % 4 == 0 && p.N > 0 && p.K > 0, "wino43_nt: bad M/N/K %lld/%d/%d", (long long)p.M, p.N, p.K);
  TL_REQUIRE(p.K % 32 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0, "wino43_nt: K %% 32, lda %% 4, ldb %% 4 must be 0");
  TL_REQUIRE(p.lda >= p.K && p.ldb >= p.K, "wino43_nt: lda/ldb smaller than K");
  TL_REQUIRE(p.Tp > 0 && p.Tp % 4 == 0, "wino43_nt: Tp must be a positive multiple of 4");
  TL_REQUIRE(p.splitk <= 1, "wino43_nt: no split-K");
  const long long nwg = ((p.M + 4 * W4_BQ - 1) / (4 * W4_BQ)) * ((p.N + W4_BN - 1) / W4_BN);
  if (nwg <= 0) return TL_OK;
  TL_REQUIRE(nwg < (1LL << 31), "wino43_nt: grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (p.loader == W_LOAD_DIRECT && p.epilogue == W_EPI_POOL) {
    TL_REQUIRE(p.row_shift == 0, "wino43_nt: forward needs row_shift 0");
    TL_REQUIRE(p.obits != nullptr && p.Tvalid % 2 == 0, "wino43_nt: POOL needs obits and an even Tvalid");
    TL_REQUIRE(p.N % 32 == 0 && p.ld_obits * 32 >= p.N, "wino43_nt: POOL needs N %% 32 == 0");
    hipLaunchKernelGGL((wino43_nt_kernel<W_LOAD_DIRECT, W_EPI_POOL>), dim3((unsigned)nwg), dim3(512), 0, st, p);
  } else if (p.loader == W_LOAD_UNPOOL && p.epilogue == W_EPI_MASK) {
    TL_REQUIRE(p.row_shift == -2, "wino43_nt: input gradient needs row_shift -2");
    TL_REQUIRE(p.abits != nullptr && (p.aux != nullptr || p.auxbits != nullptr), "wino43_nt: UNPOOL/MASK need abits and aux or auxbits");
    TL_REQUIRE(p.Tvalid_in % 2 == 0, "wino43_nt: UNPOOL needs an even Tvalid_in");
    hipLaunchKernelGGL((wino43_nt_kernel<W_LOAD_UNPOOL, W_EPI_MASK>), dim3((unsigned)nwg), dim3(512), 0, st, p);
  } else if (p.loader == W_LOAD_UNPOOL && p.epilogue == W_EPI_C1W) {
    TL_REQUIRE(p.row_shift == -2 && p.abits != nullptr && p.Tvalid_in % 2 == 0, "wino43_nt: input gradient needs row_shift -2, abits, even Tvalid_in");
    TL_REQUIRE(p.auxbits && p.c1x && p.c1bits && p.c1partial, "wino43_nt: epilogue 4 needs auxbits, c1x, c1bits, c1partial");
    TL_REQUIRE(p.c1kt >= 1 && p.c1kt <= 3 && p.c1T >= 2 * p.Tvalid + 2, "wino43_nt: epilogue 4: 1..3 taps, c1T >= 2*Tvalid + 2");
    hipLaunchKernelGGL((wino43_nt_kernel<W_LOAD_UNPOOL, W_EPI_C1W>), dim3((unsigned)nwg), dim3(512), 0, st, p);
  } else {
    set_error("wino43_nt: unsupported loader/epilogue combination %d/%d", p.loader, p.epilogue);
    return TL_EINVAL;
  }
  return check_launch("wino43_nt");
}

extern "C" int tl_wino43_weights7(const float* w, float* fwd, int O, int I, int taps, int nseg, void* stream) {
  using namespace tl;
  TL_REQUIRE(w != nullptr && fwd != nullptr, "wino43_weights7: null pointer");
  TL_REQUIRE(O > 0 && I > 0 && taps >= 4 && taps <= 9, "wino43_weights7: bad sizes (4..9 taps)");
  TL_REQUIRE((nseg == 2 || nseg == 3) && 3 * (nseg - 1) < taps, "wino43_weights7: 2 or 3 segments, each with a tap");
  const long long n = (long long)O * nseg * I;
  TL_REQUIRE(n < (1LL << 31), "wino43_weights7: too large");
  hipLaunchKernelGGL(wino43_weights7_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, fwd,
                     O, I, taps, nseg);
  return check_launch("wino43_weights7");
}

extern "C" int tl_conv7_wino43_nt(const tl_nt_params* pp, void* stream) {
  using namespace tl;
  TL_REQUIRE(pp != nullptr, "conv7_wino43: null params");
  const tl_nt_params& p = *pp;
  TL_REQUIRE(p.A && p.Bw && p.out, "conv7_wino43: null A/Bw/out");
  TL_REQUIRE(p.J >= 4 && p.J <= 9, "conv7_wino43: 4..6 taps (two segments) or 7..9 (three)");
  const int nseg = (p.J + 2) / 3;
  TL_REQUIRE(p.M >= 0 && p.M % 4 == 0 && p.N > 0 && p.K > 0, "conv7_wino43: bad M/N/K %lld/%d/%d", (long long)p.M, p.N, p.K);
  TL_REQUIRE(p.K % 32 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0, "conv7_wino43: K %% 32, lda %% 4, ldb %% 4 must be 0");
  TL_REQUIRE(p.lda >= p.K && p.ldb >= nseg * p.K, "conv7_wino43: lda < K or ldb < segments * K");
  TL_REQUIRE(p.aux == nullptr || p.ldaux >= p.N, "conv7_wino43: ldaux < N");
  TL_REQUIRE(p.loader == W_LOAD_DIRECT && p.epilogue == W_EPI_LRELU && p.row_shift == 0 && p.splitk <= 1,
             "conv7_wino43: DIRECT loader, LRELU epilogue, row_shift 0, no split-K");
  const long long nwg = ((p.M + 4 * W4_BQ - 1) / (4 * W4_BQ)) * ((p.N + W4_BN - 1) / W4_BN);
  if (nwg <= 0) return TL_OK;
  TL_REQUIRE(nwg < (1LL << 31), "conv7_wino43: grid too large");
  if (nseg == 3)
    hipLaunchKernelGGL((wino43_nt_kernel<W_LOAD_DIRECT, W_EPI_LRELU, 3>), dim3((unsigned)nwg), dim3(512), 0,
                       (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((wino43_nt_kernel<W_LOAD_DIRECT, W_EPI_LRELU, 2>), dim3((unsigned)nwg), dim3(512), 0,
                       (hipStream_t)stream, p);
  return check_launch("conv7_wino43");
}

extern "C" int tl_conv3_wino_tn(const tl_tn_params* pp, void* stream) {
  using namespace tl;
  TL_REQUIRE(pp != nullptr, "wino_tn: null params");
  tl_tn_params p = *pp;
  if (p.splitk < 1) p.splitk = 1;
  TL_REQUIRE(p.A && p.B && p.slab && p.bbits, "wino_tn: null A/B/bbits/slab");
  TL_REQUIRE(p.J == 3 && p.loader == W_LOAD_UNPOOL, "wino_tn: 3 taps, UNPOOL loader only");
  TL_REQUIRE(p.Krows > 0 && p.Krows % 2 == 0 && p.Mdim > 0 && p.Ndim > 0, "wino_tn: bad sizes");
  TL_REQUIRE(p.Krows + 64 < (1LL << 31), "wino_tn: more than 2^31 reduction rows");
  TL_REQUIRE(p.Mdim % 4 == 0 && p.Ndim % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0, "wino_tn: dims/ld must be multiples of 4");
  TL_REQUIRE(p.Tp > 0 && p.Tp % 2 == 0 && p.Tvalid % 2 == 0, "wino_tn: even Tp/Tvalid needed");
  TL_REQUIRE(p.ld_bbits * 32 >= p.Ndim, "wino_tn: bbits row too short");
  TL_REQUIRE(p.splitk <= 65535, "wino_tn: splitk too large");
  const long long t = (long long)((p.Mdim + 127) / 128) * ((p.Ndim + WT_BN - 1) / WT_BN);
  TL_REQUIRE(t < (1LL << 31), "wino_tn: grid too large");
  hipLaunchKernelGGL(wino_tn_kernel, dim3((unsigned)t, (unsigned)p.splitk, 1), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("wino_tn");
}

extern "C" int tl_wino_wgrad_finalize(const float* red, float* gw, int O, int I, int ld, void* stream) {
  using namespace tl;
  TL_REQUIRE(red && gw && O > 0 && I > 0 && ld >= O, "wino_wgrad_finalize: bad arguments");
  const long long n = (long long)O * I;
  hipLaunchKernelGGL(wino_wgrad_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, red,
                     gw, O, I, ld);
  return check_launch("wino_wgrad_finalize");
}

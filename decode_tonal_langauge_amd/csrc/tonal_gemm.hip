// fp32 MFMA implicit-GEMM kernels for gfx950 (MI355X): windowed NT (forward / input-gradient
// of the (k,1) convolutions, Linear, LSTM h.W_hh^T) and windowed TN (weight gradients).
//
// Both use v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  A single accumulator chain
// already issues back to back (64-cycle issue = 64-cycle dependent latency), so the design
// effort goes into keeping the matrix pipe fed: register-staged global loads issued one K-step
// ahead, LDS double buffering with one barrier per K-step, 2 workgroups (8 waves) per CU.
#include "tonal_common.h"
#include <stdlib.h>

namespace tl {

constexpr int BN = 128;   // column tile
constexpr int BK = 32;    // K depth of one LDS stage
constexpr int LDS_LD = BK + 4;   // 36 floats = 144 B rows: conflict-free ds_read_b128 (9 odd)

enum { LOAD_DIRECT = 0, LOAD_UNPOOL = 1 };
enum { EPI_STORE = 0, EPI_LRELU = 1, EPI_POOL = 2, EPI_MASK = 3 };

// ------------------------------------------------------------------------------------------
// NT window kernel
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// shared epilogue of the NT kernels.  C/D map of 32x32x2: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
// ------------------------------------------------------------------------------------------
template <int EPI, int MI, int NI>
__device__ __forceinline__ void nt_epilogue(const tl_nt_params& p, f32x16 (&acc)[MI][NI], long long R0, int n0, int wm,
                                            int wn, int lr, int lh, int z) {
  // C/D map of 32x32x2: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = n0 + wn * (NI * 32) + ni * 32 + lr;
      const bool colok = col < p.N;
      const long long rbase = R0 + wm * (MI * 32) + mi * 32;
      if constexpr (EPI == EPI_POOL) {
        const float bv = (colok && p.bias) ? p.bias[col] : 0.f;
        int tq = (int)((rbase + 4 * lh) % p.Tp);              // time index of row rbase + 4 lh; rows below step by 2 / 8
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long long Rq = rbase + 8 * q + 4 * lh;        // even conv row of this lane's 4 rows
          if (q > 0) {
            tq += 8;
            while (tq >= p.Tp) tq -= p.Tp;
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float y0 = lrelu(acc[mi][ni][4 * q + 2 * e] + bv, p.slope);
            const float y1 = lrelu(acc[mi][ni][4 * q + 2 * e + 1] + bv, p.slope);
            const bool rowok = (Rq + 2 * e) < p.M;              // Tp, Tvalid even: the pair shares validity
            int te = tq + 2 * e;
            if (te >= p.Tp) te -= p.Tp;
            const bool valid_e = rowok && te < p.Tvalid;
            const bool sel = valid_e && colok && (y1 > y0);
            const float o = valid_e ? (sel ? y1 : y0) : 0.f;
            const long long prow = (Rq >> 1) + e;
            if (rowok && colok) p.out[prow * (long long)p.ldo + col] = o;
            const unsigned long long m = __ballot(sel);
            const unsigned long long ms = __ballot(o > 0.f);
            if (lr == 0 && rowok && (n0 + wn * (NI * 32) + ni * 32) < p.N) {
              const long long at = prow * (long long)p.ld_obits + ((n0 + wn * (NI * 32) + ni * 32) >> 5);
              p.obits[at] = (uint32_t)(m >> (32 * lh));
              if (p.osign != nullptr) p.osign[at] = (uint32_t)(ms >> (32 * lh));
            }
          }
        }
      } else {
        float bv = 0.f;
        if constexpr (EPI == EPI_STORE || EPI == EPI_LRELU) bv = (colok && p.bias && p.splitk == 1) ? p.bias[col] : 0.f;
        float* outp = p.out + (long long)z * p.slab_stride;
        // MASK from sign bits: the 16 words this lane group needs (one per row) are fetched by lanes
        // 0..15 of each half-wave up front and read back with a shuffle - no dependent load per row
        uint32_t sword = 0;
        if constexpr (EPI == EPI_MASK) {
          if (p.auxbits != nullptr) {
            const int es = lr & 15;
            const long long Rs = rbase + (es & 3) + 8 * (es >> 2) + 4 * lh;
            const int cb = n0 + wn * (NI * 32) + ni * 32;
            if (Rs < p.M && cb < p.N) sword = p.auxbits[Rs * (long long)p.ld_auxbits + (cb >> 5)];
          }
        }
        // MASK from the stage input itself (no sign words: the 1x1 stack): all 16 values of the tile are requested before the
        // first store - behind a store the compiler may not move a load that could alias it, and one dependent load per row
        // (16 round trips per 32 x 32 tile) made this epilogue the longest part of a short-K launch
        float ax[16];
        if constexpr (EPI == EPI_MASK) {
          if (p.auxbits == nullptr) {
            // (through a buffer resource like the stores below: rows from M on and columns from N on read 0)
            const long long ab = (p.M - rbase) * (long long)p.ldaux * 4;
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(p.aux + rbase * (long long)p.ldaux), 0, (int)(ab <= 0 ? 0 : (ab < 0x7fffffffLL ? ab : 0x7fffffffLL)), 0x00020000);
            const unsigned lda4 = (unsigned)p.ldaux * 4u;
            const unsigned xo0 = colok ? (unsigned)(4 * lh) * lda4 + (unsigned)col * 4u : 0x80000000u;
#pragma unroll
            for (int e = 0; e < 16; ++e)
              ax[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, xo0 + (unsigned)((e & 3) + 8 * (e >> 2)) * lda4, 0u, 0));
          }
        }
        // stores through a buffer resource over the rows of this 32-row tile that exist (rows from M on are past its end and
        // dropped by the range check; a lane whose column is past N carries an offset no resource reaches): one 32-bit add per
        // store instead of a 64-bit address and a branch
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(outp + rbase * (long long)p.ldo), 0,
            (int)((p.M - rbase) <= 0 ? 0 : ((p.M - rbase) * (long long)p.ldo * 4 < 0x7fffffffLL ? (p.M - rbase) * (long long)p.ldo * 4 : 0x7fffffffLL)),
            0x00020000);
        const unsigned ldo4 = (unsigned)p.ldo * 4u;
        const unsigned vo0 = colok ? (unsigned)(4 * lh) * ldo4 + (unsigned)col * 4u : 0x80000000u;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          uint32_t word = 0;
          if constexpr (EPI == EPI_MASK) word = __shfl(sword, e + 32 * lh);
          float v = acc[mi][ni][e] + bv;
          if constexpr (EPI == EPI_LRELU) v = lrelu(v, p.slope);
          if constexpr (EPI == EPI_MASK) {
            bool pos;
            if (p.auxbits != nullptr)
              pos = (word >> lr) & 1u;
            else
              pos = ax[e] > 0.f;
            v = pos ? v : v * p.slope;
          }
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsO, vo0 + (unsigned)((e & 3) + 8 * (e >> 2)) * ldo4, 0u, 0);
        }
      }
    }
  }
}

// BM = 256 runs 8 waves (512 threads, 4 x 2): the per-thread staging work per MFMA halves, which is
// what limits the 128-row variant (scripts/mfma_ablate.hip); BM = 32 is the skinny-M streaming form.
template <int BM>
struct nt_cfg {
  static constexpr int NTHR = (BM == 256) ? 512 : 256;
  static constexpr int WM = (BM == 256) ? 4 : ((BM == 128) ? 2 : 1);   // waves along M
  static constexpr int WN = (NTHR / 64) / WM;                            // waves along N
};

template <int BM, int LOADER, int EPI>
__global__ __launch_bounds__(nt_cfg<BM>::NTHR, 2) void nt_window_kernel(const tl_nt_params p) {
  constexpr int NTHR = nt_cfg<BM>::NTHR;
  constexpr int WM = nt_cfg<BM>::WM;
  constexpr int WN = nt_cfg<BM>::WN;
  constexpr int MI = BM / (32 * WM);               // 32x32 tiles per wave along M
  constexpr int NI = BN / (32 * WN);
  constexpr int AROWS = BM + 6;                    // staged rows incl. the tap window (J <= 7)
  constexpr int A_F4 = (LOADER == LOAD_DIRECT) ? ((AROWS * 8 + NTHR - 1) / NTHR) : ((AROWS / 2 * 8 + NTHR - 1) / NTHR);
  constexpr int B_F4 = BN * 8 / NTHR;

  __shared__ __attribute__((aligned(16))) float lds[2 * AROWS * LDS_LD + 2 * BN * LDS_LD];
  float* As = lds;                                 // [2][AROWS][LDS_LD]
  float* Bs = lds + 2 * AROWS * LDS_LD;            // [2][BN][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;

  // XCD-aware tile order: blocks b and b+8 share an XCD/L2, so give each XCD a contiguous
  // run of logical tiles; consecutive logical tiles walk N first and share the A panel.
  const int ntn = (p.N + BN - 1) / BN;
  const long long ntm = (p.M + BM - 1) / BM;
  const long long nwg = ntm * ntn;
  long long bid = blockIdx.x;
  {
    const long long q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const long long tm = bid / ntn;
  const int tn = (int)(bid % ntn);
  const long long R0 = tm * BM;
  const int n0 = tn * BN;
  const int J = p.J;
  const int arows_used = BM + J - 1;

  // K range (split-K only meaningful for J == 1 plain GEMMs, but handled generally)
  const int nkc_all = (p.K + BK - 1) / BK;
  const int z = blockIdx.y;
  const int kc_per = (nkc_all + p.splitk - 1) / p.splitk;
  const int kc_begin = z * kc_per;
  const int kc_end = min(nkc_all, kc_begin + kc_per);
  const int nchunks = max(0, kc_end - kc_begin);
  const int nsteps = nchunks * J;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 ra[A_F4];          // staged A (DIRECT) or G (UNPOOL)
  uint32_t rbits[A_F4];    // UNPOOL: raw 32-bit arg word (bit set -> odd row of the pair); the
                           // nibble is extracted at LDS-store time so the load stays in flight
  f32x4 rbP[B_F4], rbQ[B_F4];    // two B staging sets: B is prefetched TWO K-steps ahead (HBM/L2 latency
                           // under full load exceeds one 64-MFMA step; measured in scripts/mfma_ablate.hip)

  const long long Abase = R0 + p.row_shift;        // first staged A row (even for UNPOOL)

  // Per-thread row pointers and validity are loop-invariant (a thread stages the same rows of
  // every K-chunk): hoist them, so a K-step issues its global loads with one 64-bit add each.
  const float* aptr[A_F4];
  const uint32_t* abptr[A_F4];
  bool aok[A_F4];
  (void)abptr;
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int idx = tid + i * NTHR;
    const int r = idx >> 3, c4 = idx & 7;
    if constexpr (LOADER == LOAD_DIRECT) {
      const long long row = Abase + r;
      aok[i] = r < arows_used && row >= 0 && row < p.A_rows;
      aptr[i] = p.A + (aok[i] ? row : 0) * (long long)p.lda + c4 * 4;
      abptr[i] = nullptr;
    } else {
      const long long prow = (Abase >> 1) + r;     // Abase is even
      aok[i] = (r < (arows_used + 1) / 2) && prow >= 0 && prow < p.A_rows &&
               (int)((2 * prow) % p.Tp) < p.Tvalid_in;
      aptr[i] = p.A + (aok[i] ? prow : 0) * (long long)p.lda + c4 * 4;
      abptr[i] = p.abits + (aok[i] ? prow : 0) * (long long)p.ld_abits;
    }
  }
  const float* bptr[B_F4];
  bool bok[B_F4];
#pragma unroll
  for (int i = 0; i < B_F4; ++i) {
    const int idx = tid + i * NTHR;
    const int r = idx >> 3, c4 = idx & 7;
    bok[i] = (n0 + r) < p.N;
    bptr[i] = p.Bw + (long long)(bok[i] ? n0 + r : 0) * p.ldb + c4 * 4;
  }
  const bool ktail = (p.K % BK) != 0;             // wave-uniform: the hot shapes have K % 32 == 0
  const int kq = (tid & 7) * 4;
  const long long tap_stride = (long long)p.N * p.ldb;

  auto load_a = [&](int chunk) {
    const int kc = (kc_begin + chunk) * BK;
    const bool kok = !ktail || (kc + kq) < p.K;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      uint32_t nib = 0;
      if (aok[i] && kok) {
        v = *reinterpret_cast<const f32x4*>(aptr[i] + kc);
        if constexpr (LOADER == LOAD_UNPOOL) nib = abptr[i][kc >> 5];
      }
      ra[i] = v;
      rbits[i] = nib;
    }
  };
  auto store_a = [&](int buf) {
    float* dst = As + buf * AROWS * LDS_LD;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + i * NTHR;
      const int r = idx >> 3, c4 = idx & 7;
      if constexpr (LOADER == LOAD_DIRECT) {
        if (r < AROWS) *reinterpret_cast<f32x4*>(dst + r * LDS_LD + c4 * 4) = ra[i];
      } else {
        if (r < AROWS / 2) {
          f32x4 e, o;
          const uint32_t nibv = rbits[i] >> ((c4 * 4) & 31);     // kc is a multiple of 32
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool odd = (nibv >> q) & 1u;
            e[q] = odd ? 0.f : ra[i][q];
            o[q] = odd ? ra[i][q] : 0.f;
          }
          *reinterpret_cast<f32x4*>(dst + (2 * r) * LDS_LD + c4 * 4) = e;
          *reinterpret_cast<f32x4*>(dst + (2 * r + 1) * LDS_LD + c4 * 4) = o;
        }
      }
    }
  };
  auto load_b = [&](f32x4 (&rb)[B_F4], int chunk, int j) {
    const int kc = (kc_begin + chunk) * BK;
    const bool kok = !ktail || (kc + kq) < p.K;
    const long long off = (long long)j * tap_stride + kc;     // wave-uniform
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (bok[i] && kok) v = *reinterpret_cast<const f32x4*>(bptr[i] + off);
      rb[i] = v;
    }
  };
  auto store_b = [&](const f32x4 (&rb)[B_F4], int buf) {
    float* dst = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int idx = tid + i * NTHR;
      const int r = idx >> 3, c4 = idx & 7;
      *reinterpret_cast<f32x4*>(dst + r * LDS_LD + c4 * 4) = rb[i];
    }
  };

  // Fragment sets F0 / F1 (statically named): while the MFMAs of one 8-deep k-group run, the
  // ds_read_b128 of the next group are in flight; the last group of a K-step is carried in
  // registers ACROSS the barrier, so the matrix pipe has work the moment the barrier opens, and
  // the LDS stores of the next stage are issued mid-step (their target buffer was released at
  // the previous barrier), leaving nothing but the barrier itself at the end of a step.
  f32x4 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
  auto load_frag = [&](f32x4 (&fa)[MI], f32x4 (&fb)[NI], int abuf, int bbuf, int j, int kk) {
    const float* a_s = As + abuf * AROWS * LDS_LD + (wm * (MI * 32) + lr + j) * LDS_LD + lh * 4 + kk * 8;
    const float* b_s = Bs + bbuf * BN * LDS_LD + (wn * (NI * 32) + lr) * LDS_LD + lh * 4 + kk * 8;
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a_s + i * 32 * LDS_LD);
#pragma unroll
    for (int i = 0; i < NI; ++i) fb[i] = *reinterpret_cast<const f32x4*>(b_s + i * 32 * LDS_LD);
  };
  auto mfma_group = [&](const f32x4 (&fa)[MI], const f32x4 (&fb)[NI]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][q], fb[ni][q], acc[mi][ni], 0, 0, 0);
  };

  // One K-step.  `rb_ld` receives B(s+2); `rb_st` holds B(s+1) (loaded one step ago) and is written
  // to LDS mid-step.  A(chunk+1) is loaded at the first tap of a chunk and stored at its last tap.
  // (chunk, tap) counters advanced without divisions
  int c_cur = 0, j_cur = 0;          // step s
  int c_ld = 0, j_ld = 0;            // step s + 2 (next B tile to load)
  auto advance = [&](int& c, int& j) {
    if (++j == J) {
      j = 0;
      ++c;
    }
  };
  auto kstep = [&](int s, f32x4 (&rb_ld)[B_F4], const f32x4 (&rb_st)[B_F4]) {
    const int chunk = c_cur, j = j_cur;
    const int abuf = chunk & 1, bbuf = s & 1;
    const bool more = s + 1 < nsteps;
    const bool has_next_chunk = chunk + 1 < nchunks;
    load_frag(fa0, fb0, abuf, bbuf, j, 0);
    if (s + 2 < nsteps) load_b(rb_ld, c_ld, j_ld);
    if (j == 0 && has_next_chunk) load_a(chunk + 1);
    if (s > 0) mfma_group(fa1, fb1);             // k-group 3 of the previous step (registers)
    load_frag(fa1, fb1, abuf, bbuf, j, 1);
    mfma_group(fa0, fb0);
    load_frag(fa0, fb0, abuf, bbuf, j, 2);
    mfma_group(fa1, fb1);
    if (more) store_b(rb_st, bbuf ^ 1);
    if (j == J - 1 && has_next_chunk) store_a(abuf ^ 1);
    load_frag(fa1, fb1, abuf, bbuf, j, 3);
    mfma_group(fa0, fb0);
    advance(c_cur, j_cur);
    advance(c_ld, j_ld);
    __syncthreads();
  };

  if (nsteps > 0) {
    load_a(0);
    load_b(rbP, 0, 0);
    store_a(0);
    store_b(rbP, 0);
    advance(c_ld, j_ld);
    if (nsteps > 1) load_b(rbQ, c_ld, j_ld);
    advance(c_ld, j_ld);
  }
  __syncthreads();
  int s = 0;
  for (; s + 1 < nsteps; s += 2) {
    kstep(s, rbP, rbQ);
    kstep(s + 1, rbQ, rbP);
  }
  if (s < nsteps) kstep(s, rbP, rbQ);
  if (nsteps > 0) mfma_group(fa1, fb1);

  nt_epilogue<EPI, MI, NI>(p, acc, R0, n0, wm, wn, lr, lh, z);
}


// ------------------------------------------------------------------------------------------
// NT window kernel, direct-to-LDS staging (global_load_lds_dwordx4): no staging VGPRs, no ds_write.
// In the register-staged kernel the B (weight) tile costs 6-7 % and the A tile 1-6 % of the matrix
// pipe (ablation with build variants); here a K-stage is 16 deep, B lives in a 4-slot LDS ring
// (three K-steps of latency tolerance), A in 2 slots, and one counted s_waitcnt + raw s_barrier
// closes a step.  LDS rows are 64 B (unpadded, as the DMA requires a lane-linear image); the 16-B
// chunk index is XOR-swizzled with (row >> 2) & 3 on the per-lane SOURCE address and on the
// fragment read, which makes the ds_read_b128 of 16 consecutive rows conflict-free.
// Preconditions (host-checked): DIRECT loader, row_shift == 0, K % 16 == 0.
// ------------------------------------------------------------------------------------------
#ifndef G_ABL
#define G_ABL 0                        // timing-only ablations of nt_glds_kernel (scripts/bench_nt1.py): 1 no epilogue stores, 2 no A pieces
#endif                                 // after the first, 4 no B pieces after the first three, 8 no wait / barrier in the loop
constexpr int GK = 16;                 // K depth of a stage
constexpr int G_AROWS = 144;           // 130 staged rows, 9 DMA pieces of 16 rows
constexpr int G_NB = 4;                // B ring slots

// NI = 32 x 32 tiles per wave along N: 2 = the 128-column tile, 1 = a 64-column tile for N <= 64 (the last layer of the 1x1 stack,
// conv5: with the 128-column tile half of their MFMAs were masked columns)
template <int EPI, int NI>
__global__ __launch_bounds__(256, 2) void nt_glds_kernel(const tl_nt_params p) {
  constexpr int BM = 128, MI = 2, WN = 2, BN = 64 * NI;          // (shadows the file's 128-column constant)
  __shared__ __attribute__((aligned(16))) float lds[2 * G_AROWS * GK + G_NB * BN * GK];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;

  const int ntn = (p.N + BN - 1) / BN;
  const long long ntm = (p.M + BM - 1) / BM;
  const long long nwg = ntm * ntn;
  long long bid = blockIdx.x;
  {
    const long long q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const long long tm = bid / ntn;
  const int tn = (int)(bid % ntn);
  const long long R0 = tm * BM;
  const int n0 = tn * BN;
  const int J = p.J;

  const int nkc_all = p.K / GK;
  const int z = blockIdx.y;
  const int kc_per = (nkc_all + p.splitk - 1) / p.splitk;
  const int kc_begin = z * kc_per;
  const int kc_end = min(nkc_all, kc_begin + kc_per);
  const int nchunks = max(0, kc_end - kc_begin);
  const int nsteps = nchunks * J;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // DMA piece = 16 rows x 64 B; lane -> (row = lane >> 2, physical chunk = lane & 3), and the source
  // chunk is the swizzled one: chunk ^ ((row >> 2) & 3) with row % 16 == lane >> 2
  const int prow = lane >> 2;
  const int src_chunk = (lane & 3) ^ ((lane >> 4) & 3);
  // A pieces: every wave issues 3 (pieces w, w+4 and 8); B pieces: 2 (w, w+4)
  const float* asrc[3];
  unsigned adst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int piece = (i < 2) ? wave + 4 * i : 8;
    long long row = R0 + piece * 16 + prow;
    if (row > p.A_rows - 1) row = p.A_rows - 1;          // clamped rows only feed masked outputs
    asrc[i] = p.A + row * (long long)p.lda + src_chunk * 4;
    adst[i] = (unsigned)(piece * 16 * GK * 4);
  }
  const float* bsrc[NI];
  unsigned bdst[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int piece = wave + 4 * i;
    int n = n0 + piece * 16 + prow;
    if (n > p.N - 1) n = p.N - 1;                         // clamped columns are masked by the epilogue
    bsrc[i] = p.Bw + (long long)n * p.ldb + src_chunk * 4;
    bdst[i] = (unsigned)(piece * 16 * GK * 4);
  }
  const long long tap_stride = (long long)p.N * p.ldb;
  char* const lds_a = reinterpret_cast<char*>(lds);
  char* const lds_b = reinterpret_cast<char*>(lds) + 2 * G_AROWS * GK * 4;

  auto dma = [&](const float* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  };
  auto issue_a = [&](int chunk) {
    const int kc = (kc_begin + chunk) * GK;
    char* base = lds_a + (chunk & 1) * (G_AROWS * GK * 4);
#pragma unroll
    for (int i = 0; i < 3; ++i) dma(asrc[i] + kc, base + adst[i]);
  };
  auto issue_b = [&](int step, int chunk, int j) {
    const long long off = (long long)j * tap_stride + (long long)(kc_begin + chunk) * GK;
    char* base = lds_b + (step & (G_NB - 1)) * (BN * GK * 4);
#pragma unroll
    for (int i = 0; i < NI; ++i) dma(bsrc[i] + off, base + bdst[i]);
  };

  // fragment read offsets (bytes): row r, logical chunk c -> r*64 + ((c ^ ((r >> 2) & 3)) * 16)
  f32x4 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
  auto load_frag = [&](f32x4 (&fa)[MI], f32x4 (&fb)[NI], int abuf, int bslot, int j, int kk) {
    const int c = 2 * kk + lh;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r = wm * 64 + i * 32 + lr + j;
      fa[i] = *reinterpret_cast<const f32x4*>(lds_a + abuf * (G_AROWS * GK * 4) + r * 64 + ((c ^ ((r >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = wn * (NI * 32) + i * 32 + lr;
      fb[i] = *reinterpret_cast<const f32x4*>(lds_b + bslot * (BN * GK * 4) + r * 64 + ((c ^ ((r >> 2) & 3)) << 4));
    }
  };
  auto mfma_group = [&](const f32x4 (&fa)[MI], const f32x4 (&fb)[NI]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][q], fb[ni][q], acc[mi][ni], 0, 0, 0);
  };

  // ---- prologue: A(0), B(0..2) in flight; wait for A(0) and B(0)
  int c_ld = 0, j_ld = 0;
  auto advance = [&](int& c, int& j) {
    if (++j == J) {
      j = 0;
      ++c;
    }
  };
  if (nsteps > 0) {
    issue_a(0);
    issue_b(0, 0, 0);
    advance(c_ld, j_ld);
    if (nsteps > 1) issue_b(1, c_ld, j_ld);
    advance(c_ld, j_ld);
    if (nsteps > 2) issue_b(2, c_ld, j_ld);
    advance(c_ld, j_ld);
    // everything but the last two B pieces-pairs must have landed (A(0), B(0))
    if (nsteps > 2) {
      if constexpr (NI == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else if (nsteps > 1) {
      if constexpr (NI == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // s_waitcnt needs an immediate: dispatch the (wave-uniform) count over the few values it takes
  auto wait_all_but = [&](int n) {
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };

  int c_cur = 0, j_cur = 0;
  for (int s = 0; s < nsteps; ++s) {
    const int chunk = c_cur, j = j_cur;
    const int abuf = chunk & 1, bslot = s & (G_NB - 1);
    load_frag(fa0, fb0, abuf, bslot, j, 0);
    // DMA issue order inside a step: A(chunk+1) (first tap of a chunk; its slot held chunk-1), then
    // B(s+3) (its ring slot held B(s-1), released at the last barrier)
    const bool has_next = chunk + 1 < nchunks;
    const bool lda_ = (j == 0) && has_next;
    const bool ldb = s + 3 < nsteps;
    if (lda_ && !(G_ABL & 2)) issue_a(chunk + 1);
    if (ldb && !(G_ABL & 4)) issue_b(s + 3, c_ld, j_ld);
    if (s > 0) mfma_group(fa1, fb1);             // k-group 1 of the previous step (registers)
    load_frag(fa1, fb1, abuf, bslot, j, 1);
    mfma_group(fa0, fb0);
    advance(c_cur, j_cur);
    advance(c_ld, j_ld);
    if (s + 1 < nsteps) {
      // The next step reads B(s+1) and, when it opens a chunk, A(chunk+1).  vmcnt retires this
      // wave's pieces in issue order, so allow exactly the pieces issued AFTER the youngest needed
      // one to stay in flight: this step's own, B(s+2) (issued one step ago) and - when the next
      // step stays in this chunk - an A(chunk+1) issued one step ago.
      int n = (ldb ? NI : 0);                              // (NI B pieces per wave and step)
      if (J == 1) {
        // A(chunk+1) was issued this step (before B(s+3)) and is needed next step
      } else {
        n += (lda_ ? 3 : 0) + ((s + 2 < nsteps) ? NI : 0);
        if (j == 1 && j != J - 1 && has_next) n += 3;
      }
      if (!(G_ABL & 8)) {
        wait_all_but((G_ABL & 6) ? 0 : n);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("" ::: "memory");
    }
  }
  if (nsteps > 0) mfma_group(fa1, fb1);
#if G_ABL & 1
  {                                                      // timing only: one store per lane instead of the epilogue
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    if (t == 12345.678f) p.out[0] = t;
  }
#else
  nt_epilogue<EPI, MI, NI>(p, acc, R0, n0, wm, wn, lr, lh, z);
#endif
}

template <int EPI>
static int launch_glds(const tl_nt_params& p, hipStream_t st) {
  const bool narrow = p.N <= 64 && p.J == 1;              // one 64-column tile (the wait counts of the narrow form cover J == 1)
  const int bn = narrow ? 64 : BN;
  const long long nwg = ((p.M + 127) / 128) * ((p.N + bn - 1) / bn);
  if (nwg <= 0) return TL_OK;
  TL_REQUIRE(nwg < (1LL << 31), "nt_glds: grid too large");
  dim3 grid((unsigned)nwg, (unsigned)p.splitk, 1);
  if (narrow)
    hipLaunchKernelGGL((nt_glds_kernel<EPI, 1>), grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((nt_glds_kernel<EPI, 2>), grid, dim3(256), 0, st, p);
  return check_launch("nt_glds");
}

template <int BM, int LOADER, int EPI>
static int launch_nt(const tl_nt_params& p, hipStream_t st) {
  const long long ntm = (p.M + BM - 1) / BM;
  const long long ntn = (p.N + BN - 1) / BN;
  const long long nwg = ntm * ntn;
  if (nwg <= 0) return TL_OK;
  TL_REQUIRE(nwg < (1LL << 31), "nt_window: grid too large");
  dim3 grid((unsigned)nwg, (unsigned)p.splitk, 1);
  hipLaunchKernelGGL((nt_window_kernel<BM, LOADER, EPI>), grid, dim3(nt_cfg<BM>::NTHR), 0, st, p);
  return check_launch("nt_window");
}

template <int BM>
static int dispatch_nt(const tl_nt_params& p, hipStream_t st) {
  if constexpr (BM == 32) {
    if (p.loader != LOAD_DIRECT) {
      set_error("nt_window: bm = 32 supports the DIRECT loader only");
      return TL_EINVAL;
    }
  }
  if (p.loader == LOAD_DIRECT) {
    switch (p.epilogue) {
      case EPI_STORE: return launch_nt<BM, LOAD_DIRECT, EPI_STORE>(p, st);
      case EPI_LRELU: return launch_nt<BM, LOAD_DIRECT, EPI_LRELU>(p, st);
      case EPI_POOL: return launch_nt<BM, LOAD_DIRECT, EPI_POOL>(p, st);
      case EPI_MASK: return launch_nt<BM, LOAD_DIRECT, EPI_MASK>(p, st);
    }
  } else if constexpr (BM != 32) {
    switch (p.epilogue) {
      case EPI_STORE: return launch_nt<BM, LOAD_UNPOOL, EPI_STORE>(p, st);
      case EPI_MASK: return launch_nt<BM, LOAD_UNPOOL, EPI_MASK>(p, st);
    }
  }
  set_error("nt_window: unsupported loader/epilogue combination %d/%d", p.loader, p.epilogue);
  return TL_EINVAL;
}

// ------------------------------------------------------------------------------------------
// TN window kernel: slab[z][j*Mdim + m][n] = sum_R A[R + j][m] * Bz[R][n]
// LDS tiles are k-major ([BK][128 + 4]); fragments are conflict-free ds_read_b32.
// ------------------------------------------------------------------------------------------
constexpr int TN_LD = 128 + 4;

template <int LOADER>
__global__ __launch_bounds__(256, 2) void tn_window_kernel(const tl_tn_params p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BK * TN_LD];
  float* As = lds;                        // [2][BK][TN_LD]
  float* Bs = lds + 2 * BK * TN_LD;       // [2][BK][TN_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const int ntm1 = (p.Mdim + 127) / 128;          // m tiles per tap
  const int ntn = (p.Ndim + 127) / 128;
  // Work item = (split z, tap j, m tile, n tile).  Every tile of one split reads the SAME rows of
  // A and B, so all tiles of a split must share an L2: blocks b and b+8 sit on one XCD, hence the
  // bijective remap that hands each XCD a contiguous run of work items (split-major).
  const long long tiles = (long long)ntm1 * ntn * p.J;
  const long long nwg = tiles * p.splitk;
  long long bid = (long long)blockIdx.y * gridDim.x + blockIdx.x;
  {
    const long long q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int z = (int)(bid / tiles);
  int tid_tile = (int)(bid % tiles);
  const int tn = tid_tile % ntn;
  tid_tile /= ntn;
  const int tm = tid_tile % ntm1;
  const int j = tid_tile / ntm1;
  const int m0 = tm * 128, n0 = tn * 128;

  const long long ksteps_all = (p.Krows + BK - 1) / BK;
  const long long per = (ksteps_all + p.splitk - 1) / p.splitk;
  const long long ks_begin = z * per;
  long long ks_end = ks_begin + per;
  if (ks_end > ksteps_all) ks_end = ksteps_all;
  const long long nsteps = ks_end > ks_begin ? ks_end - ks_begin : 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // column sums of B over this split (the bias gradient of a one-tap layer: B = the layer's output gradient): the first row
  // tile's workgroup of a (split, column tile) adds up the rows it stages anyway - no pass of its own over B
  const bool do_cs = LOADER == LOAD_DIRECT && p.colsum != nullptr && tm == 0 && j == 0;
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};

  // two staging sets (P/Q): tiles are prefetched two K-steps ahead
  f32x4 raP[4], rbP[4], raQ[4], rbQ[4];
  uint32_t rnP[2], rnQ[2];
  (void)rnP; (void)rnQ;

  // Hoisted per-thread state: row pointers advance by a wave-uniform stride per K-step, column
  // validity is loop-invariant, row validity ((row % Tp) < Tvalid) is tracked incrementally.
  constexpr int NB = (LOADER == LOAD_DIRECT) ? 4 : 2;
  const int a_lim = (int)(p.A_rows < p.Krows + j ? p.A_rows : p.Krows + j);
  const int b_lim = (LOADER == LOAD_DIRECT) ? (int)(p.B_rows < p.Krows ? p.B_rows : p.Krows)
                                            : (int)(2 * p.B_rows < p.Krows ? 2 * p.B_rows : p.Krows);
  const int kbase = (int)(ks_begin * BK);
  const float* aptr[4];
  int arow[4];
  const bool amok = (m0 + (tid & 31) * 4) < p.Mdim;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (tid + i * 256) >> 5;
    arow[i] = kbase + r + j;
    aptr[i] = p.A + (long long)arow[i] * p.lda + m0 + (tid & 31) * 4;
  }
  const float* bptr[NB];
  const uint32_t* bbptr[NB];
  int brow[NB], bt[NB];
  (void)bbptr;
  const int ncol = n0 + (tid & 31) * 4;
  const bool bnok = ncol < p.Ndim;
  const int dstep = BK % p.Tp;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int r = (tid + i * 256) >> 5;
    if constexpr (LOADER == LOAD_DIRECT) {
      brow[i] = kbase + r;                              // B row
      bptr[i] = p.B + (long long)brow[i] * p.ldb + ncol;
      bbptr[i] = nullptr;
    } else {
      brow[i] = kbase + 2 * r;                          // even conv row of the pooled pair
      bptr[i] = p.B + (long long)(brow[i] >> 1) * p.ldb + ncol;
      bbptr[i] = p.bbits + (long long)(brow[i] >> 1) * p.ld_bbits + (ncol >> 5);
    }
    bt[i] = brow[i] % p.Tp;
  }
  const long long a_step = (long long)BK * p.lda;
  const long long b_step = (LOADER == LOAD_DIRECT) ? (long long)BK * p.ldb : (long long)(BK / 2) * p.ldb;
  const long long bb_step = (long long)(BK / 2) * p.ld_bbits;

  // loads the tiles of the NEXT not-yet-loaded step and advances the per-thread state
  auto load_tiles = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], uint32_t (&rnib)[2], long long /*step*/) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (amok && arow[i] < a_lim) v = *reinterpret_cast<const f32x4*>(aptr[i]);
      ra[i] = v;
      aptr[i] += a_step;
      arow[i] += BK;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      uint32_t nib = 0;
      if (bnok && brow[i] < b_lim && bt[i] < p.Tvalid) {
        v = *reinterpret_cast<const f32x4*>(bptr[i]);
        if constexpr (LOADER == LOAD_UNPOOL) nib = *bbptr[i];
      }
      rb[i] = v;
      if constexpr (LOADER == LOAD_DIRECT) {
        if (do_cs) cs += v;
      }
      if constexpr (LOADER == LOAD_UNPOOL) {
        rnib[i] = nib;
        bbptr[i] += bb_step;
      }
      bptr[i] += b_step;
      brow[i] += BK;
      bt[i] += dstep;
      if (bt[i] >= p.Tp) bt[i] -= p.Tp;
    }
  };
  auto store_tiles = [&](const f32x4 (&ra)[4], const f32x4 (&rb)[4], const uint32_t (&rnib)[2], int buf) {
    float* da = As + buf * BK * TN_LD;
    float* db = Bs + buf * BK * TN_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int r = idx >> 5, c4 = idx & 31;
      *reinterpret_cast<f32x4*>(da + r * TN_LD + c4 * 4) = ra[i];
    }
    if constexpr (LOADER == LOAD_DIRECT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;
        const int r = idx >> 5, c4 = idx & 31;
        *reinterpret_cast<f32x4*>(db + r * TN_LD + c4 * 4) = rb[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        const int pr = idx >> 5, c4 = idx & 31;
        f32x4 e, o;
        const uint32_t nibv = rnib[i] >> (ncol & 31);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool odd = (nibv >> q) & 1u;
          e[q] = odd ? 0.f : rb[i][q];
          o[q] = odd ? rb[i][q] : 0.f;
        }
        *reinterpret_cast<f32x4*>(db + (2 * pr) * TN_LD + c4 * 4) = e;
        *reinterpret_cast<f32x4*>(db + (2 * pr + 1) * TN_LD + c4 * 4) = o;
      }
    }
  };

  // same software pipeline as the NT kernel: k-groups of 8 rows (4 MFMA k-steps), fragment
  // sets F0/F1, the last group carried across the barrier, LDS stores issued mid-step
  float ta0[4][2], tb0[4][2], ta1[4][2], tb1[4][2];
  auto load_frag = [&](float (&fa)[4][2], float (&fb)[4][2], int buf, int g) {
    const float* a_s = As + buf * BK * TN_LD + (g * 8 + lh) * TN_LD + wm * 64 + lr;
    const float* b_s = Bs + buf * BK * TN_LD + (g * 8 + lh) * TN_LD + wn * 64 + lr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fa[q][0] = a_s[q * 2 * TN_LD];
      fa[q][1] = a_s[q * 2 * TN_LD + 32];
      fb[q][0] = b_s[q * 2 * TN_LD];
      fb[q][1] = b_s[q * 2 * TN_LD + 32];
    }
  };
  auto mfma_group = [&](const float (&fa)[4][2], const float (&fb)[4][2]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][mi], fb[q][ni], acc[mi][ni], 0, 0, 0);
  };

  auto kstep = [&](long long s, f32x4 (&ra_ld)[4], f32x4 (&rb_ld)[4], uint32_t (&rn_ld)[2], const f32x4 (&ra_st)[4],
                   const f32x4 (&rb_st)[4], const uint32_t (&rn_st)[2]) {
    const int buf = (int)(s & 1);
    load_frag(ta0, tb0, buf, 0);
    if (s + 2 < nsteps) load_tiles(ra_ld, rb_ld, rn_ld, s + 2);
    if (s > 0) mfma_group(ta1, tb1);
    load_frag(ta1, tb1, buf, 1);
    mfma_group(ta0, tb0);
    load_frag(ta0, tb0, buf, 2);
    mfma_group(ta1, tb1);
    if (s + 1 < nsteps) store_tiles(ra_st, rb_st, rn_st, buf ^ 1);
    load_frag(ta1, tb1, buf, 3);
    mfma_group(ta0, tb0);
    __syncthreads();
  };

  if (nsteps > 0) {
    load_tiles(raP, rbP, rnP, 0);
    store_tiles(raP, rbP, rnP, 0);
    if (nsteps > 1) load_tiles(raQ, rbQ, rnQ, 1);
  }
  __syncthreads();
  long long s = 0;
  for (; s + 1 < nsteps; s += 2) {
    kstep(s, raP, rbP, rnP, raQ, rbQ, rnQ);
    kstep(s + 1, raQ, rbQ, rnQ, raP, rbP, rnP);
  }
  if (s < nsteps) kstep(s, raP, rbP, rnP, raQ, rbQ, rnQ);
  if (nsteps > 0) mfma_group(ta1, tb1);

  if (do_cs) {                                            // (workgroup-uniform; every LDS read of the loop is behind its last barrier)
    float* red = lds;                                     // [8 row groups][128 columns]
    *reinterpret_cast<f32x4*>(red + (tid >> 5) * 128 + (tid & 31) * 4) = cs;
    __syncthreads();
    if (tid < 128 && n0 + tid < p.Ndim) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += red[g * 128 + tid];
      p.colsum[(long long)z * p.Ndim + n0 + tid] = t;
    }
  }

  float* out = p.slab + (long long)z * p.slab_stride;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + ni * 32 + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < p.Mdim && col < p.Ndim)
          out[((long long)j * p.Mdim + m) * (long long)p.ldc + col] = acc[mi][ni][e];
      }
    }
}


// ------------------------------------------------------------------------------------------
// TN, all three taps per workgroup (the conv weight gradient): one staged activation tile
// (BK + 2 rows) and one staged dZ tile feed the accumulators of all 3 taps, so the staging work
// per MFMA is half that of the one-tap kernel.  Block tile 128 (C_in) x 64 (C_out), wave tile
// 64 x 32 x 3 taps = 96 accumulator registers, still 2 workgroups per CU.
// ------------------------------------------------------------------------------------------
constexpr int T3_BN = 64, T3_LDA = 128 + 4, T3_LDB = T3_BN + 4, T3_AR = BK + 2;

template <int LOADER>
__global__ __launch_bounds__(256, 2) void tn3_kernel(const tl_tn_params p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * T3_AR * T3_LDA + 2 * BK * T3_LDB];
  float* As = lds;
  float* Bs = lds + 2 * T3_AR * T3_LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int ntm = (p.Mdim + 127) / 128, ntn = (p.Ndim + T3_BN - 1) / T3_BN;
  const long long tiles = (long long)ntm * ntn;
  const long long nwg = tiles * p.splitk;
  long long bid = (long long)blockIdx.y * gridDim.x + blockIdx.x;
  {
    const long long q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int z = (int)(bid / tiles);
  const int tt = (int)(bid % tiles);
  const int m0 = (tt / ntn) * 128, n0 = (tt % ntn) * T3_BN;

  const long long ksteps_all = (p.Krows + BK - 1) / BK;
  const long long per = (ksteps_all + p.splitk - 1) / p.splitk;
  const long long ks_begin = z * per;
  long long ks_end = ks_begin + per;
  if (ks_end > ksteps_all) ks_end = ksteps_all;
  const long long nsteps = ks_end > ks_begin ? ks_end - ks_begin : 0;

  f32x16 acc[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  constexpr int NB = (LOADER == LOAD_DIRECT) ? 2 : 1;
  f32x4 raP[5], raQ[5], rbP[NB], rbQ[NB];
  uint32_t rnP[NB], rnQ[NB];
  (void)rnP; (void)rnQ;

  const int a_lim = (int)(p.A_rows < p.Krows + 2 ? p.A_rows : p.Krows + 2);
  const int b_lim = (LOADER == LOAD_DIRECT) ? (int)(p.B_rows < p.Krows ? p.B_rows : p.Krows)
                                            : (int)(2 * p.B_rows < p.Krows ? 2 * p.B_rows : p.Krows);
  const int kbase = (int)(ks_begin * BK);
  const float* aptr[5];
  int arow[5];
  bool aact[5];
  const bool amok = (m0 + (tid & 31) * 4) < p.Mdim;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int r = (tid + i * 256) >> 5;           // 0..39, rows >= 34 unused
    aact[i] = r < T3_AR;
    arow[i] = kbase + r;
    aptr[i] = p.A + (long long)arow[i] * p.lda + m0 + (tid & 31) * 4;
  }
  const float* bptr[NB];
  const uint32_t* bbptr[NB];
  int brow[NB], bt[NB];
  (void)bbptr;
  const int ncol = n0 + (tid & 15) * 4;
  const bool bnok = ncol < p.Ndim;
  const int dstep = BK % p.Tp;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int r = (tid + i * 256) >> 4;           // 16 float4 per 64-column row
    if constexpr (LOADER == LOAD_DIRECT) {
      brow[i] = kbase + r;
      bptr[i] = p.B + (long long)brow[i] * p.ldb + ncol;
      bbptr[i] = nullptr;
    } else {
      brow[i] = kbase + 2 * r;
      bptr[i] = p.B + (long long)(brow[i] >> 1) * p.ldb + ncol;
      bbptr[i] = p.bbits + (long long)(brow[i] >> 1) * p.ld_bbits + (ncol >> 5);
    }
    bt[i] = brow[i] % p.Tp;
  }
  const long long a_step = (long long)BK * p.lda;
  const long long b_step = (LOADER == LOAD_DIRECT) ? (long long)BK * p.ldb : (long long)(BK / 2) * p.ldb;
  const long long bb_step = (long long)(BK / 2) * p.ld_bbits;

  auto load_tiles = [&](f32x4 (&ra)[5], f32x4 (&rb)[NB], uint32_t (&rn)[NB]) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (aact[i] && amok && arow[i] < a_lim) v = *reinterpret_cast<const f32x4*>(aptr[i]);
      ra[i] = v;
      aptr[i] += a_step;
      arow[i] += BK;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      uint32_t nib = 0;
      if (bnok && brow[i] < b_lim && bt[i] < p.Tvalid) {
        v = *reinterpret_cast<const f32x4*>(bptr[i]);
        if constexpr (LOADER == LOAD_UNPOOL) nib = *bbptr[i];
      }
      rb[i] = v;
      if constexpr (LOADER == LOAD_UNPOOL) {
        rn[i] = nib;
        bbptr[i] += bb_step;
      }
      bptr[i] += b_step;
      brow[i] += BK;
      bt[i] += dstep;
      if (bt[i] >= p.Tp) bt[i] -= p.Tp;
    }
  };
  auto store_tiles = [&](const f32x4 (&ra)[5], const f32x4 (&rb)[NB], const uint32_t (&rn)[NB], int buf) {
    float* da = As + buf * T3_AR * T3_LDA;
    float* db = Bs + buf * BK * T3_LDB;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = tid + i * 256;
      const int r = idx >> 5, c4 = idx & 31;
      if (r < T3_AR) *reinterpret_cast<f32x4*>(da + r * T3_LDA + c4 * 4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + i * 256;
      const int r = idx >> 4, c4 = idx & 15;
      if constexpr (LOADER == LOAD_DIRECT) {
        *reinterpret_cast<f32x4*>(db + r * T3_LDB + c4 * 4) = rb[i];
      } else {
        f32x4 e, o;
        const uint32_t nibv = rn[i] >> (ncol & 31);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool odd = (nibv >> q) & 1u;
          e[q] = odd ? 0.f : rb[i][q];
          o[q] = odd ? rb[i][q] : 0.f;
        }
        *reinterpret_cast<f32x4*>(db + (2 * r) * T3_LDB + c4 * 4) = e;
        *reinterpret_cast<f32x4*>(db + (2 * r + 1) * T3_LDB + c4 * 4) = o;
      }
    }
  };

  // fragments of one k-group (8 rows = 4 MFMA k-steps): A needs rows lh .. lh+8 (taps 0..2)
  float fa0[10][2], fb0[4], fa1[10][2], fb1[4];
  auto load_frag = [&](float (&fa)[10][2], float (&fb)[4], int buf, int g) {
    const float* a_s = As + buf * T3_AR * T3_LDA + (g * 8 + lh) * T3_LDA + wm * 64 + lr;
    const float* b_s = Bs + buf * BK * T3_LDB + (g * 8 + lh) * T3_LDB + wn * 32 + lr;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      fa[r][0] = a_s[r * T3_LDA];
      fa[r][1] = a_s[r * T3_LDA + 32];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) fb[q] = b_s[q * 2 * T3_LDB];
  };
  auto mfma_group = [&](const float (&fa)[10][2], const float (&fb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[j][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * q + j][mi], fb[q], acc[j][mi], 0, 0, 0);
  };
  auto kstep = [&](long long s, f32x4 (&ra_ld)[5], f32x4 (&rb_ld)[NB], uint32_t (&rn_ld)[NB], const f32x4 (&ra_st)[5],
                   const f32x4 (&rb_st)[NB], const uint32_t (&rn_st)[NB]) {
    const int buf = (int)(s & 1);
    load_frag(fa0, fb0, buf, 0);
    if (s + 2 < nsteps) load_tiles(ra_ld, rb_ld, rn_ld);
    if (s > 0) mfma_group(fa1, fb1);
    load_frag(fa1, fb1, buf, 1);
    mfma_group(fa0, fb0);
    load_frag(fa0, fb0, buf, 2);
    mfma_group(fa1, fb1);
    if (s + 1 < nsteps) store_tiles(ra_st, rb_st, rn_st, buf ^ 1);
    load_frag(fa1, fb1, buf, 3);
    mfma_group(fa0, fb0);
    __syncthreads();
  };

  if (nsteps > 0) {
    load_tiles(raP, rbP, rnP);
    store_tiles(raP, rbP, rnP, 0);
    if (nsteps > 1) load_tiles(raQ, rbQ, rnQ);
  }
  __syncthreads();
  long long s = 0;
  for (; s + 1 < nsteps; s += 2) {
    kstep(s, raP, rbP, rnP, raQ, rbQ, rnQ);
    kstep(s + 1, raQ, rbQ, rnQ, raP, rbP, rnP);
  }
  if (s < nsteps) kstep(s, raP, rbP, rnP, raQ, rbQ, rnQ);
  if (nsteps > 0) mfma_group(fa1, fb1);

  float* out = p.slab + (long long)z * p.slab_stride;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int col = n0 + wn * 32 + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < p.Mdim && col < p.Ndim) out[((long long)j * p.Mdim + m) * (long long)p.ldc + col] = acc[j][mi][e];
      }
    }
}

// ------------------------------------------------------------------------------------------
// TN, skinny M (<= 32 rows of output): slab[z][m][n] = sum_k A[k][m] * B[k][n].  Used for
// dh = dgates . W_hh on the few distinct label rows: B (5.4 GB) is streamed exactly once, so the
// tile is 32 x 512 with a 16-deep K stage (33 KB in flight per workgroup, 2 workgroups per CU) and
// the kernel is HBM-bound, not MFMA-bound.
// ------------------------------------------------------------------------------------------
constexpr int SK_BK = 16, SK_BN = 512, SK_LDB = SK_BN + 4, SK_LDA = 32 + 4;

__global__ __launch_bounds__(256, 2) void tn_skinny_kernel(const tl_tn_params p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * SK_BK * SK_LDA + 2 * SK_BK * SK_LDB];
  float* As = lds;
  float* Bs = lds + 2 * SK_BK * SK_LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * SK_BN;
  const int z = blockIdx.y;
  const long long ksteps_all = (p.Krows + SK_BK - 1) / SK_BK;
  const long long per = (ksteps_all + p.splitk - 1) / p.splitk;
  const long long ks_begin = z * per;
  long long ks_end = ks_begin + per;
  if (ks_end > ksteps_all) ks_end = ksteps_all;
  const long long nsteps = ks_end > ks_begin ? ks_end - ks_begin : 0;

  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  f32x4 rb[8], ra;
  const bool mask_rows = p.Tp > 1;

  auto load_tiles = [&](long long step) {
    const long long k0 = (ks_begin + step) * SK_BK;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + i * 256;
      const int r = idx >> 7, c4 = idx & 127;
      const long long row = k0 + r;
      const int n = n0 + c4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      bool ok = row < p.Krows && row < p.B_rows && n < p.Ndim;
      if (mask_rows) ok = ok && (int)(row % p.Tp) < p.Tvalid;
      if (ok) v = *reinterpret_cast<const f32x4*>(p.B + row * (long long)p.ldb + n);
      rb[i] = v;
    }
    {
      const int r = (tid >> 3) & 15, c4 = tid & 7;
      const long long row = k0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (tid < 128 && row < p.Krows && row < p.A_rows && c4 * 4 < p.Mdim)
        v = *reinterpret_cast<const f32x4*>(p.A + row * (long long)p.lda + c4 * 4);
      ra = v;
    }
  };
  auto store_tiles = [&](int buf) {
    float* db = Bs + buf * SK_BK * SK_LDB;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + i * 256;
      const int r = idx >> 7, c4 = idx & 127;
      *reinterpret_cast<f32x4*>(db + r * SK_LDB + c4 * 4) = rb[i];
    }
    if (tid < 128) {
      const int r = tid >> 3, c4 = tid & 7;
      *reinterpret_cast<f32x4*>(As + buf * SK_BK * SK_LDA + r * SK_LDA + c4 * 4) = ra;
    }
  };

  if (nsteps > 0) {
    load_tiles(0);
    store_tiles(0);
  }
  __syncthreads();
  for (long long s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    if (more) load_tiles(s + 1);
    const float* a_s = As + (s & 1) * SK_BK * SK_LDA + lh * SK_LDA + lr;
    const float* b_s = Bs + (s & 1) * SK_BK * SK_LDB + lh * SK_LDB + wave * 128 + lr;
#pragma unroll
    for (int kk = 0; kk < SK_BK / 2; ++kk) {
      const float fa = a_s[kk * 2 * SK_LDA];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, b_s[kk * 2 * SK_LDB + ni * 32], acc[ni], 0, 0, 0);
    }
    if (more) store_tiles((s + 1) & 1);
    __syncthreads();
  }
  float* out = p.slab + (long long)z * p.slab_stride;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int col = n0 + wave * 128 + ni * 32 + lr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (m < p.Mdim && col < p.Ndim) out[(long long)m * p.ldc + col] = acc[ni][e];
    }
  }
}

}  // namespace tl

extern "C" int tl_gemm_nt_window(const tl_nt_params* pp, void* stream) {
  using namespace tl;
  TL_REQUIRE(pp != nullptr, "nt_window: null params");
  tl_nt_params p = *pp;
  if (p.splitk < 1) p.splitk = 1;
  if (p.bm == 0) p.bm = 128;
  TL_REQUIRE(p.A && p.Bw && p.out, "nt_window: null A/Bw/out");
  TL_REQUIRE(p.M >= 0 && p.N > 0 && p.K > 0, "nt_window: bad M/N/K %lld/%d/%d", (long long)p.M, p.N, p.K);
  TL_REQUIRE(p.K % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0, "nt_window: K, lda, ldb must be multiples of 4");
  TL_REQUIRE(p.lda >= p.K && p.ldb >= p.K, "nt_window: lda/ldb smaller than K");
  TL_REQUIRE(p.J >= 1 && p.J <= 7, "nt_window: J must be 1..7");
  TL_REQUIRE(p.row_shift == 0 || p.row_shift == -(p.J - 1), "nt_window: row_shift must be 0 or -(J-1)");
  TL_REQUIRE(p.bm == 256 || p.bm == 128 || p.bm == 32, "nt_window: bm must be 256, 128 or 32");
  TL_REQUIRE(p.Tp > 0, "nt_window: Tp must be positive");
  TL_REQUIRE(p.splitk == 1 || p.epilogue == EPI_STORE, "nt_window: split-K needs the STORE epilogue");
  TL_REQUIRE(p.splitk <= 65535, "nt_window: splitk too large");
  if (p.loader == LOAD_UNPOOL) {
    TL_REQUIRE(p.abits != nullptr, "nt_window: UNPOOL loader needs abits");
    TL_REQUIRE((p.row_shift % 2) == 0 && p.Tp % 2 == 0 && p.Tvalid_in % 2 == 0, "nt_window: UNPOOL needs even shift/Tp/Tvalid_in");
    TL_REQUIRE(p.K % 32 == 0 || p.ld_abits * 32 >= p.K, "nt_window: abits row too short");
    TL_REQUIRE(p.bm != 32, "nt_window: UNPOOL loader needs bm = 128 or 256");
  }
  if (p.epilogue == EPI_POOL) {
    TL_REQUIRE(p.obits != nullptr, "nt_window: POOL epilogue needs obits");
    TL_REQUIRE(p.Tp % 2 == 0 && p.Tvalid % 2 == 0, "nt_window: POOL needs even Tp/Tvalid");
    TL_REQUIRE(p.N % 32 == 0 && p.ld_obits * 32 >= p.N, "nt_window: POOL needs N %% 32 == 0");
  }
  if (p.epilogue == EPI_MASK) TL_REQUIRE(p.aux != nullptr || p.auxbits != nullptr, "nt_window: MASK epilogue needs aux or auxbits");
  hipStream_t st = (hipStream_t)stream;
  // Direct-to-LDS staging variant.  Round 1 measured it equal to the register-staged kernel on the conv2 forward (129.6 vs
  // 128.6 TFLOP/s) and kept the simpler one; on what is left for this entry point since the Winograd kernels took the 3-tap
  // stages - conv4 / conv5 forward, the 1x1 stack, the Linear layer - it is the faster one (train step 209.7 -> 208.9 ms,
  // same-call A/B, round 4): default on; TONAL_GLDS=0 selects the register-staged kernel (the A/B partner,
  // tests/test_gpu_parity.py holds the two against each other).
  // (an A/B switch of the test suite: honoured only under TONAL_AB=1, like the per-switch variables of _kernels.py)
  const char* ab = getenv("TONAL_AB");
  const char* genv = (ab != nullptr && ab[0] == '1') ? getenv("TONAL_GLDS") : nullptr;
  const bool glds_on = genv == nullptr || genv[0] != '0';
  if (glds_on && p.J <= 3 && p.bm == 128 && p.loader == LOAD_DIRECT && p.row_shift == 0 && (p.K % GK) == 0 && p.A_rows > 0) {
    switch (p.epilogue) {
      case EPI_STORE: return launch_glds<EPI_STORE>(p, st);
      case EPI_LRELU: return launch_glds<EPI_LRELU>(p, st);
      case EPI_POOL: return launch_glds<EPI_POOL>(p, st);
      case EPI_MASK: return launch_glds<EPI_MASK>(p, st);
    }
  }
  if (p.bm == 256) return dispatch_nt<256>(p, st);
  return p.bm == 128 ? dispatch_nt<128>(p, st) : dispatch_nt<32>(p, st);
}

namespace tl {
// Short reductions (a few hundred rows: the weight gradients of SynthesisLite's Linear / LSTM layers at batch 64): the MFMA
// kernels' 128 x 128 tiles leave a handful of workgroups with a short K loop of dependent latency (17 - 30 us).  Here a
// workgroup takes a 32 x 128 tile and walks the rows in chunks of 64: one round of loads per chunk (issued before the
// arithmetic of the chunk in front of it, 40 KB of LDS), plain FMAs on a 4 x 4 patch per thread, rows in order
// (deterministic); masked rows ((R % Tp) >= Tvalid) are zero-filled, not read.
constexpr int TS_K = 64, TS_M = 32, TS_N = 128;
__global__ __launch_bounds__(256) void tn_short_kernel(const tl_tn_params p) {
  __shared__ __attribute__((aligned(16))) float As[TS_K * TS_M];
  __shared__ __attribute__((aligned(16))) float Bs[TS_K * TS_N];
  const int tid = threadIdx.x, tn = tid & 31, tm = tid >> 5;
  const int ntn = (p.Ndim + TS_N - 1) / TS_N;
  const int mt = (int)(blockIdx.x / ntn) * TS_M, nt = (int)(blockIdx.x % ntn) * TS_N;
  const int kall = (int)(p.Krows < p.A_rows ? (p.Krows < p.B_rows ? p.Krows : p.B_rows) : (p.A_rows < p.B_rows ? p.A_rows : p.B_rows));
  // reduction split blockIdx.y: rows [kb, kr) = whole 64-row chunks (slab blockIdx.y; the caller sums the splitk slabs)
  const int per = ((kall + (int)gridDim.y - 1) / (int)gridDim.y + TS_K - 1) / TS_K * TS_K;
  const int kb = (int)blockIdx.y * per;
  const int kr = kb + per < kall ? kb + per : kall;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // a chunk: A 64 rows x 8 float4, B 64 rows x 32 float4 (Mdim, Ndim % 4 == 0: a float4 is whole or absent)
  f32x4 ra[2], rb[8];
  // (every load is issued, from a clamped address, and zeroed afterwards: predicated loads come out as one exec-masked
  // region each, with the latencies in series)
  int ta[2], tb[8];                                          // (row offset in the chunk) % Tp: the division is done once
#pragma unroll
  for (int i = 0; i < 2; ++i) ta[i] = ((tid + i * 256) >> 3) % p.Tp;
#pragma unroll
  for (int i = 0; i < 8; ++i) tb[i] = ((tid + i * 256) >> 5) % p.Tp;
  const int kstep_mod = TS_K % p.Tp;
  int r0m = kb % p.Tp;                                       // r0 % Tp of the chunk being fetched
  auto fetch = [&](int r0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256, R = r0 + (idx >> 3), c = mt + (idx & 7) * 4;
      int t = ta[i] + r0m;
      t = t >= p.Tp ? t - p.Tp : t;
      const bool ok = R < kr && t < p.Tvalid && c < p.Mdim;
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.A + (long long)(R < kall ? R : kall - 1) * p.lda + (c < p.Mdim ? c : p.Mdim - 4));
      ra[i] = ok ? v : zero;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + i * 256, R = r0 + (idx >> 5), c = nt + (idx & 31) * 4;
      int t = tb[i] + r0m;
      t = t >= p.Tp ? t - p.Tp : t;
      const bool ok = R < kr && t < p.Tvalid && c < p.Ndim;
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.B + (long long)(R < kall ? R : kall - 1) * p.ldb + (c < p.Ndim ? c : p.Ndim - 4));
      rb[i] = ok ? v : zero;
    }
    r0m += kstep_mod;
    r0m = r0m >= p.Tp ? r0m - p.Tp : r0m;
  };
  f32x4 acc[4] = {zero, zero, zero, zero};
  // optional column sums of A (the bias gradient of a Linear layer rides along): the workgroups of the first column tile,
  // threads 0..31, one column each, rows in order
  const bool do_cs = p.colsum != nullptr && nt == 0 && gridDim.y == 1 && tid < TS_M;
  float csum = 0.f;
  fetch(kb);
  for (int r0 = kb; r0 < kr; r0 += TS_K) {
    __syncthreads();                                       // the chunk in front has been consumed
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(As + ((tid + i * 256) >> 3) * TS_M + ((tid + i * 256) & 7) * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(Bs + ((tid + i * 256) >> 5) * TS_N + ((tid + i * 256) & 31) * 4) = rb[i];
    __syncthreads();
    if (r0 + TS_K < kr) fetch(r0 + TS_K);
    if (do_cs) {
#pragma unroll 8
      for (int R = 0; R < TS_K; ++R) csum += As[R * TS_M + tid];
    }
#pragma unroll 8
    for (int R = 0; R < TS_K; ++R) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(As + R * TS_M + tm * 4);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(Bs + R * TS_N + tn * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += av[i] * bv;
    }
  }
  const int m0 = mt + tm * 4, n0 = nt + tn * 4;
  if (m0 < p.Mdim && n0 < p.Ndim) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(p.slab + (long long)blockIdx.y * p.slab_stride + (long long)(m0 + i) * p.ldc + n0) = acc[i];
  }
  if (do_cs && mt + tid < p.Mdim) p.colsum[mt + tid] = csum;
}
}  // namespace tl

extern "C" int tl_gemm_tn_window(const tl_tn_params* pp, void* stream) {
  using namespace tl;
  TL_REQUIRE(pp != nullptr, "tn_window: null params");
  tl_tn_params p = *pp;
  if (p.splitk < 1) p.splitk = 1;
  TL_REQUIRE(p.A && p.B && p.slab, "tn_window: null A/B/slab");
  TL_REQUIRE(p.Krows > 0 && p.Mdim > 0 && p.Ndim > 0, "tn_window: bad sizes");
  TL_REQUIRE(p.Krows + 64 < (1LL << 31), "tn_window: more than 2^31 reduction rows");
  TL_REQUIRE(p.Mdim % 4 == 0 && p.Ndim % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0, "tn_window: dims/ld must be multiples of 4");
  TL_REQUIRE(p.J >= 1 && p.J <= 3, "tn_window: J must be 1..3");
  TL_REQUIRE(p.Tp > 0, "tn_window: Tp must be positive");
  TL_REQUIRE(p.splitk <= 65535, "tn_window: splitk too large");
  if (p.loader == LOAD_UNPOOL) {
    TL_REQUIRE(p.bbits != nullptr, "tn_window: UNPOOL loader needs bbits");
    TL_REQUIRE(p.Tp % 2 == 0 && p.Tvalid % 2 == 0, "tn_window: UNPOOL needs even Tp/Tvalid");
    TL_REQUIRE(p.ld_bbits * 32 >= p.Ndim, "tn_window: bbits row too short");
  }
  hipStream_t st = (hipStream_t)stream;
  if (p.J == 1 && p.loader == LOAD_DIRECT && p.Krows <= 8 * TS_K && p.splitk <= 8 && (long long)p.Mdim * p.Ndim <= (1 << 20) &&
      (p.splitk == 1 || p.slab_stride >= (long long)p.Mdim * p.ldc) &&
      p.ldc % 4 == 0) {                                             // short reduction, small output: latency-bound
    const long long nwg = (long long)((p.Mdim + TS_M - 1) / TS_M) * ((p.Ndim + TS_N - 1) / TS_N);
    hipLaunchKernelGGL(tn_short_kernel, dim3((unsigned)nwg, (unsigned)p.splitk), dim3(256), 0, st, p);
    return check_launch("tn_short");
  }
  TL_REQUIRE(p.colsum == nullptr || (p.J == 1 && p.loader == LOAD_DIRECT && p.Mdim > 32),
             "tn_window: colsum comes from the short-reduction kernel (Krows <= 512, splitk <= 8, Mdim * Ndim <= 2^20) or the one-tap direct kernel (Mdim > 32)");
  if (p.Mdim <= 32 && p.J == 1 && p.loader == LOAD_DIRECT) {       // skinny-M streaming variant
    dim3 grid((unsigned)((p.Ndim + SK_BN - 1) / SK_BN), (unsigned)p.splitk, 1);
    hipLaunchKernelGGL(tn_skinny_kernel, grid, dim3(256), 0, st, p);
    return check_launch("tn_skinny");
  }
  if (p.J == 3) {                                                    // all taps per workgroup
    const long long t3 = (long long)((p.Mdim + 127) / 128) * ((p.Ndim + T3_BN - 1) / T3_BN);
    TL_REQUIRE(t3 < (1LL << 31), "tn_window: grid too large");
    dim3 grid((unsigned)t3, (unsigned)p.splitk, 1);
    if (p.loader == LOAD_DIRECT)
      hipLaunchKernelGGL((tn3_kernel<LOAD_DIRECT>), grid, dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL((tn3_kernel<LOAD_UNPOOL>), grid, dim3(256), 0, st, p);
    return check_launch("tn3");
  }
  const long long ntm = (p.Mdim + 127) / 128, ntn = (p.Ndim + 127) / 128;
  const long long nwg = ntm * ntn * p.J;
  TL_REQUIRE(nwg < (1LL << 31), "tn_window: grid too large");
  dim3 grid((unsigned)nwg, (unsigned)p.splitk, 1);
  if (p.loader == LOAD_DIRECT)
    hipLaunchKernelGGL((tn_window_kernel<LOAD_DIRECT>), grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((tn_window_kernel<LOAD_UNPOOL>), grid, dim3(256), 0, st, p);
  return check_launch("tn_window");
}

extern "C" int tl_sizeof_nt_params(void) { return (int)sizeof(tl_nt_params); }
extern "C" int tl_sizeof_tn_params(void) { return (int)sizeof(tl_tn_params); }

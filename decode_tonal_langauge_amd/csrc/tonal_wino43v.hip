// Winograd F(4,3) with the INPUT TRANSFORM HOISTED OUT of the GEMM kernels (round 3).
//
// conv2 / conv3 of the ECoG stack (models/synthesis_models.py:91-97) are 79 % of the train step's FLOPs.
// The F(4,3) kernels of tonal_wino.hip / tonal_wino43_tn.hip stage raw activation rows and apply
// V = B^T d (six transforms of the rows 4Q..4Q+5 of a quad Q) next to the MFMAs - and on gfx950 the fp32
// MFMA runs at the fp32 VECTOR rate: every VALU instruction, every register-staged global load and every
// ds_write of the two waves of a SIMD displaces matrix-pipe time one for one
// (profiles/r02_kernel_notes.md section 3a: 2.9 cycles per VALU, 23 per global load, 13 per ds_write_b64).
// V depends on the activations only, and both the forward pass and the weight gradient consume exactly
// the same V.  So the producer of an activation writes V once (1.5 x the bytes of the raw rows - HBM is
// at 8 % of its roof in these kernels), and the two GEMM kernels below become transform-free:
//
//   tl_wino43_input_transform   P (rows, channels-last)  ->  V[quad][6][ldv]         (stand-alone form;
//                               tl_conv1_fwd writes V directly for the first stage)
//   wino43v_nt_kernel           forward: M_i[quad][n] = sum_k V_i[quad][k] U_i[n][k], i < 6, a batched
//                               NT GEMM whose two operands go global -> LDS by `buffer_load_dwordx4 ..
//                               offen lds` (LDS-DMA: no staging VGPRs, no ds_write, no vector address
//                               arithmetic: SGPR resource + a per-lane offset that is fixed for the whole
//                               kernel + a scalar K offset), followed by the pool epilogue of
//                               tonal_wino.hip (y = A^T M, bias, LeakyReLU, max-pool, arg-max bits)
//   wino43v_tn_kernel           weight gradient: slab_i[c_in][c_out] = sum_quads V_i (x) Y_i, V by
//                               LDS-DMA, Y = A dy built from the pooled gradient + arg-max bits at
//                               staging time as before
//
// LDS images written by LDS-DMA are lane-linear (1 KiB per wave-instruction), so rows cannot be padded;
// bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle of the 16-byte chunk
// index that is applied to the per-lane SOURCE address and to the fragment read.
#include "tonal_common.h"
#include "tonal_wino43_epi.h"
#include "tonal_wino43v_epi.h"
#include <type_traits>

namespace tl {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// P -> V.  Thread = (quad, 4 channels).  Rows of a quad that lie beyond its sequence (t >= Tp) are
// taken as zero: they only ever reach conv rows the epilogues mask (the Winograd identity holds for
// any finite value there; zero keeps the cancellation error of the last valid row smallest).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void wino43_xform_kernel(const float* __restrict__ P, float* __restrict__ V,
                                                            long long nq, int Tp, int C, int ldp, int ldv) {
  const int c4n = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nq * c4n) return;
  const long long q = idx / c4n;
  const int c = (int)(idx - q * c4n) * 4;
  const int tq = Tp >> 2;
  const long long seq = q / tq;
  const int t0 = (int)(q - seq * tq) * 4;
  const float* src = P + (seq * Tp + t0) * (long long)ldp + c;
  f32x4 d[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    if (t0 + j < Tp) d[j] = *reinterpret_cast<const f32x4*>(src + (long long)j * ldp);
    else d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const f32x4 s1 = d[4] - 4.f * d[2], s2 = d[3] - 4.f * d[1], s3 = d[4] - d[2], t = d[3] - d[1];
  float* dst = V + q * 6 * (long long)ldv + c;
  *reinterpret_cast<f32x4*>(dst) = 4.f * d[0] + (d[4] - 5.f * d[2]);
  *reinterpret_cast<f32x4*>(dst + ldv) = s1 + s2;
  *reinterpret_cast<f32x4*>(dst + 2LL * ldv) = s1 - s2;
  *reinterpret_cast<f32x4*>(dst + 3LL * ldv) = s3 + 2.f * t;
  *reinterpret_cast<f32x4*>(dst + 4LL * ldv) = s3 - 2.f * t;
  *reinterpret_cast<f32x4*>(dst + 5LL * ldv) = (4.f * d[1] - 5.f * d[3]) + d[5];
}

// B^T d of six un-pooled gradient rows with every product fused (one rounding per fmaf, nothing left for the compiler to
// contract): the writers of Vd - the stand-alone kernel below and both tilings of the weight-gradient kernel - agree bit
// for bit whatever vector width they are compiled at.
__device__ __forceinline__ void vd_transform(float d0, float d1, float d2, float d3, float d4, float d5, float (&v)[6]) {
  const float s1 = fmaf(-4.f, d2, d4), s2 = fmaf(-4.f, d1, d3), s3 = d4 - d2, t = d3 - d1;
  v[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
  v[1] = s1 + s2;
  v[2] = s1 - s2;
  v[3] = fmaf(2.f, t, s3);
  v[4] = fmaf(-2.f, t, s3);
  v[5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
}

// ------------------------------------------------------------------------------------------
// (G, arg-max bits) -> Vd: the input-gradient operand of a pooled 3-tap stage.  Vd[quad q][6][ldv] = B^T of the
// un-pooled dZ rows 4q-2 .. 4q+3 (row Rz of dZ = G[Rz / 2] where the arg-max bit equals Rz & 1 and (Rz % Tp) < Tvalid,
// else 0).  Thread = (quad, 4 channels): three pooled rows + three bit words in, six 16-byte stores out; no LDS and few
// registers on purpose - the kernel is HBM-bound and is launched on a side stream BESIDE the MFMA-bound weight-gradient
// kernel of the same stage, whose two workgroups per CU leave the register file room for one such wave per SIMD.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void wino43_unpool_xform_kernel(const float* __restrict__ G, const uint32_t* __restrict__ bits,
                                                                     float* __restrict__ V, long long nq, long long g_rows,
                                                                     int Tp, int Tvalid, int C, int ldg, int ld_bits, int ldv) {
  const int c4n = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nq * c4n) return;
  const long long q = idx / c4n;
  const int c = (int)(idx - q * c4n) * 4;
  const int tq = (int)((4 * q) % Tp);                       // time index of conv row 4q
  f32x4 d[6];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const long long pr = 2 * q - 1 + j;                     // pooled row; its conv rows have time tq - 2 + 2j, + 1
    const int t = tq - 2 + 2 * j;
    const bool ok = pr >= 0 && pr < g_rows && t >= 0 && t < Tvalid;      // t < 0: the previous sequence's pad rows
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    uint32_t w = 0;
    if (ok) {
      g = *reinterpret_cast<const f32x4*>(G + pr * (long long)ldg + c);
      w = bits[pr * (long long)ld_bits + (c >> 5)] >> (c & 31);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool odd = (w >> k) & 1u;
      d[2 * j][k] = odd ? 0.f : g[k];
      d[2 * j + 1][k] = odd ? g[k] : 0.f;
    }
  }
  f32x4 o[6];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v[6];
    vd_transform(d[0][k], d[1][k], d[2][k], d[3][k], d[4][k], d[5][k], v);
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i][k] = v[i];
  }
  float* dst = V + q * 6 * (long long)ldv + c;
#pragma unroll
  for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(dst + (long long)i * ldv) = o[i];
}

// ------------------------------------------------------------------------------------------
// Second half of the V-writing forward epilogue (tonal_wino43v_epi.h, POOLV): the last output quad of every 512-row tile
// needs pooled rows 4, 5 from the next tile.  The tile left rows 0..3 raw in the quad's transform slots 0..3 and every
// tile stored its first two pooled rows to vhalo; thread = (tile, 4 channels) transforms the quad in place with the
// expressions of wino43_xform_kernel.  Rows 4, 5 are zero where the quad ends its sequence or the matrix.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void wino43_v_fixup_kernel(float* __restrict__ V, const float* __restrict__ halo, long long quads,
                                                              long long tiles, int Tq, int C, int ldv) {
  const int c4n = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= tiles * c4n) return;
  const long long t = idx / c4n;
  const int c = (int)(idx - t * c4n) * 4;
  const long long q = t * 64 + 63;
  if (q >= quads) return;
  float* v = V + q * 6 * (long long)ldv + c;
  f32x4 d[6];
#pragma unroll
  for (int j = 0; j < 4; ++j) d[j] = *reinterpret_cast<const f32x4*>(v + (long long)j * ldv);
  const int tq = (int)((4 * q) % Tq);
  if (tq + 4 < Tq && t + 1 < tiles) {
    d[4] = *reinterpret_cast<const f32x4*>(halo + ((t + 1) * 2) * (long long)C + c);
    d[5] = *reinterpret_cast<const f32x4*>(halo + ((t + 1) * 2 + 1) * (long long)C + c);
  } else {
    d[4] = d[5] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const f32x4 s1 = d[4] - 4.f * d[2], s2 = d[3] - 4.f * d[1], s3 = d[4] - d[2], tt = d[3] - d[1];
  *reinterpret_cast<f32x4*>(v) = 4.f * d[0] + (d[4] - 5.f * d[2]);
  *reinterpret_cast<f32x4*>(v + ldv) = s1 + s2;
  *reinterpret_cast<f32x4*>(v + 2LL * ldv) = s1 - s2;
  *reinterpret_cast<f32x4*>(v + 3LL * ldv) = s3 + 2.f * tt;
  *reinterpret_cast<f32x4*>(v + 4LL * ldv) = s3 - 2.f * tt;
  *reinterpret_cast<f32x4*>(v + 5LL * ldv) = (4.f * d[1] - 5.f * d[3]) + d[5];
}

// ------------------------------------------------------------------------------------------
// Forward NT kernel on V.  Workgroup: 8 waves as 4 (quads) x 2 (columns); block tile 128 quads (512 conv
// rows) x 64 columns; wave tile 32 quads x 32 columns x 6 transforms = 96 accumulator registers.  A K-step
// is 16 channels: A stage [6][128 quads][64 B] = 48 KB, B stage [6][64 columns][64 B] = 24 KB, two stages
// = 144 KB (one workgroup per CU, two waves per SIMD).  A stage is filled by 72 LDS-DMA pieces (16 rows x
// 64 B each), 9 per wave, issued right after the barrier that opens the previous step; one counted wait +
// one barrier closes a step.  The MFMA loop is 24 ds_read_b128 + 48 MFMAs per wave and K-step and nothing
// else; the last k-group of a step is carried in registers across the barrier.
// ------------------------------------------------------------------------------------------
constexpr int V4_BQ = 128, V4_BN = 64, V4_BK = 16;
constexpr int V4_ROWB = V4_BK * 4;                       // bytes per LDS row (64)
constexpr int V4_A_BYTES = 6 * V4_BQ * V4_ROWB;          // 49152
constexpr int V4_B_BYTES = 6 * V4_BN * V4_ROWB;          // 24576
constexpr int V4_STAGE = V4_A_BYTES + V4_B_BYTES;        // 73728
constexpr int V4_APIECES = 6 * V4_BQ / 16;               // 48
constexpr int V4_BPIECES = 6 * V4_BN / 16;               // 24

typedef __attribute__((address_space(3))) void lds_void_t;
#ifndef V4_STAGGER
#define V4_STAGGER 0      // 1: waves 4-7 issue their LDS-DMA pieces in the second half of a K-step (see kstep)
#endif
#ifndef V4_PERSIST
#define V4_PERSIST 1      // workgroups per CU of the persistent launch (the next tile's first fetch overlaps the epilogue); 0:
#endif                    // one workgroup per tile, as before
#ifndef V4_SCHED
#define V4_SCHED 1        // 1: hand-specified issue order of a K-step (see kstep)
#endif
#ifndef V4_LEAN
#define V4_LEAN 1         // round 4: quads of a wave in the order of tonal_wino43v_epi.h (a lane owns 16 consecutive quads), scalar-side
#endif                    // epilogues, first K-step of a tile without the empty carried group; 0: the round-3 kernel (A/B partner)
#ifndef V4_ABL
#define V4_ABL 0          // timing-only build variants (scripts/build_v_variants.sh): 1 no steady-state DMA, 2 no epilogue,
#endif                    // 4 no barrier, 8 order pinned at the top of a K-step, 16 no fragment reads

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)lds_dst, 16, voff, soff, 0, 0);
}
// the same with the non-temporal cache policy (aux = 2): for a stream every CU reads once or eight times within a few
// microseconds (the V operand of the NT kernels) and that should not push a re-used operand (the taps) out of the L2
__device__ __forceinline__ void dma16_nt(__amdgpu_buffer_rsrc_t rs, char* lds_dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)lds_dst, 16, voff, soff, 0, 2);
}
#ifndef V4_NT
#define V4_NT 0           // cache policy experiment of the NT kernel: 1 = V (streaming operand) non-temporal, 2 = taps non-temporal
#endif

template <int EPI>
__global__ __launch_bounds__(512, 2) void wino43v_nt_kernel(const tl_nt_params p) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * V4_STAGE + (EPI == W_EPI_POOLV ? 4096 : (EPI == W_EPI_C1W && V4_LEAN) ? 16384 : 0)];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const int ntn = (p.N + V4_BN - 1) / V4_BN;
  const long long ntm = (p.M + 4 * V4_BQ - 1) / (4 * V4_BQ);
  const long long nwg = ntm * ntn;
  const int nsteps = p.K / V4_BK;                          // host-checked: K % 16 == 0, K >= 16
  f32x16 acc[6];

  // ---- LDS-DMA plan.  Piece = 16 rows x 64 B; lane -> (row = lane >> 2, physical chunk = lane & 3); the
  // source chunk is the swizzled one, chunk ^ ((row >> 2) & 3) with row % 16 == lane >> 2.
  const int prow = lane >> 2;
  const int src_chunk = (lane & 3) ^ ((lane >> 4) & 3);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.Bw, 0, (int)(6LL * p.N * p.ldb * 4), 0x00020000);
  unsigned adst[6], bdst[3];
#pragma unroll
  for (int t = 0; t < 6; ++t) adst[t] = (unsigned)((((wave * 6 + t) >> 3) * V4_BQ + ((wave * 6 + t) & 7) * 16) * V4_ROWB);
#pragma unroll
  for (int t = 0; t < 3; ++t) bdst[t] = (unsigned)(V4_A_BYTES + (((wave * 3 + t) >> 2) * V4_BN + ((wave * 3 + t) & 3) * 16) * V4_ROWB);
  // A workgroup walks the tiles blockIdx.x, + gridDim.x, ... (the host launches one workgroup per CU, a multiple of 8: a
  // tile sequence stays on one XCD) and fetches the first K-step of its next tile BEFORE the epilogue of the current one:
  // the epilogue (5 % of a tile) then runs beside that fetch instead of in front of it.
  struct tile_t {
    long long tm, R0;
    int n0;
    __amdgpu_buffer_rsrc_t rsA;
    unsigned avoff[6], bvoff[3];
  };
  auto setup = [&](long long bid) -> tile_t {
    {
      const long long q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
      bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    tile_t t;
    t.tm = bid / ntn;
    t.R0 = t.tm * (4 * V4_BQ);
    t.n0 = (int)(bid % ntn) * V4_BN;
    const long long Qt = t.tm * V4_BQ;                     // first quad of the tile
    // A pieces of this wave: pa = 6 wave + t -> (transform i = pa >> 3, 16-quad block j = pa & 7).  Quads past the
    // end of V are clamped (they only feed rows the epilogue masks).
    const long long q_left = p.A_rows - Qt;                // quads addressable from the tile start (> 0)
    const long long a_span = q_left * 6 * (long long)p.lda * 4;
    t.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + Qt * 6 * (long long)p.lda), 0,
                                              (int)(a_span < 0x7fffffffLL ? a_span : 0x7fffffffLL), 0x00020000);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int pa = wave * 6 + k;
      const int i = pa >> 3, j = pa & 7;
#if V4_LEAN
      // LDS row rho = 8 g + 4 lh + j' of a wave's 32 (the MFMA row whose results lane half lh holds in accumulator
      // elements 4 g + j') takes quad 16 lh + 4 g + j': a lane owns 16 consecutive quads (tonal_wino43v_epi.h)
      const int rho = (j & 1) * 16 + prow;
      long long ql = (j >> 1) * 32 + ((rho & 3) | ((rho >> 3) << 2) | (((rho >> 2) & 1) << 4));
#else
      long long ql = j * 16 + prow;
#endif
      if (ql > q_left - 1) ql = q_left - 1;
      t.avoff[k] = (unsigned)(((ql * 6 + i) * p.lda + src_chunk * 4) * 4);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int pb = wave * 3 + k;
      const int i = pb >> 2, j = pb & 3;
      int n = t.n0 + j * 16 + prow;
      if (n > p.N - 1) n = p.N - 1;                        // clamped columns are masked by the epilogue
      t.bvoff[k] = (unsigned)((((long long)i * p.N + n) * p.ldb + src_chunk * 4) * 4);
    }
    return t;
  };
  tile_t cur = setup(blockIdx.x);
#if V4_LEAN
  // Next tile of this workgroup WITHOUT the divisions of setup(): with a persistent launch of 8 k workgroups the XCD-aware
  // remap sends tile sequence blockIdx.x + j gridDim.x to b' + j gridDim.x / 8, so (tm, tn) advance by a fixed (dq, dr) with
  // a carry; the V quads of a tile always exist (host-checked: V holds whole 128-quad tiles), so the per-lane source
  // offsets of the A pieces never change and those of the B pieces move with the column tile.  (setup() costs ~250
  // scalar / vector instructions per tile; a wave issues at most one instruction of ANY kind per ~4 cycles, and with
  // the matrix pipe idle between two tiles every one of them is exposed.)
  const int walk = (int)(gridDim.x >> 3), walk_q = walk / ntn, walk_r = walk - walk_q * ntn;
  const long long a_tile_bytes = (long long)V4_BQ * 6 * p.lda * 4, a_total_bytes = p.A_rows * 6LL * p.lda * 4;
  unsigned bv_lane[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int pb = wave * 3 + k;
    bv_lane[k] = (unsigned)((((long long)(pb >> 2) * p.N + (pb & 3) * 16 + prow) * p.ldb + src_chunk * 4) * 4);
  }
  auto advance = [&](const tile_t& c) -> tile_t {
    tile_t t = c;
    int tn = c.n0 / V4_BN + walk_r;
    long long tm = c.tm + walk_q;
    if (tn >= ntn) {
      tn -= ntn;
      ++tm;
    }
    t.tm = tm;
    t.R0 = tm * (4 * V4_BQ);
    t.n0 = tn * V4_BN;
    const long long ab = tm * a_tile_bytes, left = a_total_bytes - ab;
    t.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.A) + ab), 0,
                                              (int)(left < 0x7fffffffLL ? left : 0x7fffffffLL), 0x00020000);
    if (t.n0 + V4_BN <= p.N) {
      const unsigned nb4 = (unsigned)t.n0 * (unsigned)p.ldb * 4u;
#pragma unroll
      for (int k = 0; k < 3; ++k) t.bvoff[k] = bv_lane[k] + nb4;
    } else {                                               // column tile past the edge: clamped columns (masked by the epilogue)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int pb = wave * 3 + k;
        int n = t.n0 + (pb & 3) * 16 + prow;
        if (n > p.N - 1) n = p.N - 1;
        t.bvoff[k] = (unsigned)((((long long)(pb >> 2) * p.N + n) * p.ldb + src_chunk * 4) * 4);
      }
    }
    return t;
  };
#endif
  auto issue = [&](const tile_t& tl_, int step) {
    char* base = lds + (step & 1) * V4_STAGE;
    const unsigned soff = (unsigned)step * (V4_BK * 4);
    if (!(V4_ABL & 64) || step < 2) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (V4_NT & 1) dma16_nt(tl_.rsA, base + adst[t], tl_.avoff[t], soff);
        else dma16(tl_.rsA, base + adst[t], tl_.avoff[t], soff);
      }
    }
    if (!(V4_ABL & 128) || step < 2) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (V4_NT & 2) dma16_nt(rsB, base + bdst[t], tl_.bvoff[t], soff);
        else dma16(rsB, base + bdst[t], tl_.bvoff[t], soff);
      }
    }
  };

  // ---- fragment reads: row r of a tile, logical chunk c = 2 g + lh -> r * 64 + ((c ^ ((r >> 2) & 3)) << 4)
  const int sw = (lr >> 2) & 3;
  const int a_row = (wm * 32 + lr) * V4_ROWB, b_row = V4_A_BYTES + (wn * 32 + lr) * V4_ROWB;
  const int c_g0 = ((lh) ^ sw) << 4, c_g1 = ((2 + lh) ^ sw) << 4;
  f32x4 fa0[6], fb0[6], fa1[6], fb1[6];
  auto load_frag = [&](f32x4 (&fa)[6], f32x4 (&fb)[6], int stage, int cg) {
    const char* a_s = lds + stage * V4_STAGE + a_row + cg;
    const char* b_s = lds + stage * V4_STAGE + b_row + cg;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      fa[i] = *reinterpret_cast<const f32x4*>(a_s + i * (V4_BQ * V4_ROWB));
      fb[i] = *reinterpret_cast<const f32x4*>(b_s + i * (V4_BN * V4_ROWB));
    }
  };
  auto mfma_group = [&](const f32x4 (&fa)[6], const f32x4 (&fb)[6]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q], fb[i][q], acc[i], 0, 0, 0);
  };
  issue(cur, 0);
  auto kstep = [&](auto LAST, auto LATE, int s) {
    const int stage = s & 1;
    if (!(V4_ABL & 16) || s == 0) load_frag(fa0, fb0, stage, c_g0);
    if constexpr (!decltype(LAST)::value)
      if (!(V4_ABL & 1) || s == 0) issue(cur, s + 1);       // the other stage was released at the last barrier
    mfma_group(fa1, fb1);                                   // k-group 1 of the previous step (registers)
    if (!(V4_ABL & 16) || s == 0) load_frag(fa1, fb1, stage, c_g1);
    mfma_group(fa0, fb0);
#if V4_SCHED
    // Issue order of the step (sched_group_barrier: 0x008 MFMA, 0x010 vector memory, 0x100 LDS read).  An LDS-DMA
    // piece costs the issuing wave ~60 cycles; issued as a clump after the barrier by both waves of a SIMD at once it
    // idles the matrix pipe (ablation: 6.4 of 45.4 ms), one piece per two MFMAs hides behind the partner's MFMAs.
    // The reads of k-group 1 are spread over the MFMAs of k-group 0 so none of their latency is left at the barrier.
    // V4_STAGGER: waves 4-7 (the SIMD partners of waves 0-3) issue their pieces in the SECOND half of the step, so the two
    // waves of a SIMD are never in their DMA-issue phase together (MI355X_MICROARCH.md, two waves per SIMD, item 9)
    constexpr bool late = decltype(LATE)::value;
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
    if constexpr (!decltype(LAST)::value && !late) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    } else {
      __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
    }
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if constexpr (!decltype(LAST)::value && late)
        if (t < 9) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    }
#endif
    if constexpr (!decltype(LAST)::value) {
      // the builtin (not inline asm) so that the compiler's own wait-count bookkeeping sees the drained counters: after
      // an asm wait it re-waits lgkmcnt(0) behind the first reads of the next step, in front of the carried MFMAs
#if V4_ABL & 32
      __builtin_amdgcn_s_waitcnt(0x3f7f & ~0x0f00);         // lgkmcnt(0)
#else
      __builtin_amdgcn_s_waitcnt(0x0070);                   // vmcnt(0) lgkmcnt(0)
#endif
#if !(V4_ABL & 4)
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
    }
  };
#if V4_LEAN
  // First K-step of a tile: the accumulators start from the zero constant of the first MFMA of each (no 96 moves) and there
  // is no carried k-group (round 3 ran 24 MFMAs on zero operands per tile here: 1.5 % of a tile's matrix time).
  auto mfma_group0 = [&](const f32x4 (&fa)[6], const f32x4 (&fb)[6]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][0], fb[i][0], zero, 0, 0, 0);
#pragma unroll
    for (int q = 1; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q], fb[i][q], acc[i], 0, 0, 0);
  };
  auto kstep_first = [&](auto LAST) {
    load_frag(fa0, fb0, 0, c_g0);
    if constexpr (!decltype(LAST)::value) issue(cur, 1);
    load_frag(fa1, fb1, 0, c_g1);
    mfma_group0(fa0, fb0);
#if V4_SCHED
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if constexpr (!decltype(LAST)::value)
        if (t < 9) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    }
#endif
    if constexpr (!decltype(LAST)::value) {
      __builtin_amdgcn_s_waitcnt(0x0070);                   // vmcnt(0) lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };
#endif
#if V4_LEAN
  __builtin_amdgcn_s_waitcnt(0x0f70);                       // vmcnt(0): the first tile's first stage
#endif
  for (long long vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
#if !V4_LEAN
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) fa1[i] = fb1[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // carried k-group of step -1: adds nothing
#endif
#if V4_LEAN
    // stage 0 of this tile has landed (waited for at the end of the tile in front), every wave is past that epilogue
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#else
    // stage 0 of this tile has landed (so have the stores of the epilogue in front of it), every wave is past that epilogue
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
#if V4_LEAN
    // What the epilogue reads from global memory is requested in front of the LAST K-step: the loads return under its 72
    // MFMAs and - vmcnt counts in issue order - in front of the next tile's LDS-DMA pieces, so their consumer does not
    // wait for those (tonal_wino43v_epi.h)
    auto prefetch = [&] {
      if constexpr (EPI == W_EPI_POOL || EPI == W_EPI_POOLV) return v5_prefetch_pool(p, cur.n0, wn, lr);
      else if constexpr (EPI == W_EPI_MASK) return v5_prefetch_mask(p, cur.R0, cur.n0, wm, wn, lr, lh);
      else return v5_prefetch_c1w(p, cur.R0, cur.n0, wm, wn, lr, lh);
    };
    decltype(prefetch()) pre;
    if (nsteps > 1) {
      kstep_first(std::false_type{});
      for (int s = 1; s + 1 < nsteps; ++s) kstep(std::false_type{}, std::false_type{}, s);
      pre = prefetch();
      __builtin_amdgcn_sched_barrier(0);
      kstep(std::true_type{}, std::false_type{}, nsteps - 1);
    } else {
      pre = prefetch();
      __builtin_amdgcn_sched_barrier(0);
      kstep_first(std::true_type{});
    }
#else
#if V4_STAGGER
    if (wave >= 4) {
      for (int s = 0; s + 1 < nsteps; ++s) kstep(std::false_type{}, std::true_type{}, s);
    } else
#endif
    {
      for (int s = 0; s + 1 < nsteps; ++s) kstep(std::false_type{}, std::false_type{}, s);
    }
    kstep(std::true_type{}, std::false_type{}, nsteps - 1);
#endif
    mfma_group(fa1, fb1);

#if V4_ABL & 2
    {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[i][e];
      if (t == 12345.678f) p.out[tid] = t;
    }
#endif
    // the first K-step of the next tile goes into stage 0 - last read in step nsteps - 2 when nsteps is even, and every
    // wave is past that step's closing barrier; the last step's stage is still being read by slower waves
    const tile_t done = cur;
    const long long nb = vb + gridDim.x;
#if V4_LEAN
    // the prefetched words have landed (they were requested a K-step ago; nothing else is in flight).  Stated here so that the
    // compiler does not place a vmcnt(0) of its own behind the conditional LDS-DMA issue below - in front of the epilogue
    __builtin_amdgcn_s_waitcnt(0x0f70);
#endif
    if (V4_PERSIST && nb < nwg) {
      if (nsteps & 1) __syncthreads();
#if V4_LEAN
      cur = advance(cur);
#else
      cur = setup(nb);
#endif
      issue(cur, 0);
    }
#if V4_LEAN
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue (tonal_wino43v_epi.h): the four conv rows of a quad from its six products, then pool (-> P and / or V
    // of the next stage) / mask / fused first-stage weight gradient; row logic on the scalar ALU ----
#if !(V4_ABL & 2)
    float* scratch = reinterpret_cast<float*>(lds + ((nsteps - 1) & 1) * V4_STAGE);
    (void)scratch;
    // (workgroup-uniform) interior tile: every row / column / output quad exists - the in-matrix masks of the stores fold away
    const bool full = done.R0 + 4 * V4_BQ <= p.M && done.n0 + V4_BN <= p.N;
    if constexpr (EPI == W_EPI_POOL) {
      if (full) v5_epilogue_pool<false, true>(p, acc, pre, nullptr, done.R0, done.n0, wm, wn, lr, lh, done.tm);
      else v5_epilogue_pool<false, false>(p, acc, pre, nullptr, done.R0, done.n0, wm, wn, lr, lh, done.tm);
    } else if constexpr (EPI == W_EPI_POOLV) {
      float* xch = reinterpret_cast<float*>(lds + 2 * V4_STAGE);
      if (full) v5_epilogue_pool<true, true>(p, acc, pre, xch, done.R0, done.n0, wm, wn, lr, lh, done.tm);
      else v5_epilogue_pool<true, false>(p, acc, pre, xch, done.R0, done.n0, wm, wn, lr, lh, done.tm);
    } else if constexpr (EPI == W_EPI_MASK) {
      if (full) v5_epilogue_mask<true>(p, acc, pre, done.R0, done.n0, wm, wn, lr, lh);
      else v5_epilogue_mask<false>(p, acc, pre, done.R0, done.n0, wm, wn, lr, lh);
    } else {
      // (the reduction takes the LAST step's stage as its scratch, behind a barrier: slower waves may still be reading it)
      v5_epilogue_c1w(p, acc, pre, reinterpret_cast<float*>(lds + 2 * V4_STAGE) + wave * 512, scratch, done.R0, done.n0, wm, wn,
                      lr, lh, done.tm);
    }
#endif
    {
      // The next tile's first stage (issued in front of the epilogue) has landed; the epilogue's own stores need not: a
      // wave issues v5_stores<EPI>() of them per tile and vmcnt counts in issue order.  (Round 3 waited for vmcnt(0) here:
      // every tile paid the write latency of its last store with no MFMA in flight.)
      constexpr int N = V4_PERSIST ? v5_stores<EPI>() : 0;
      __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    }
#else
    // ---- epilogue (shared with the in-loop-transform kernels, tonal_wino43_epi.h): the four conv rows of a quad from
    // its six products, then pool / mask / fused first-stage weight gradient.  The reduction of the fused weight gradient
    // takes the LAST step's stage as its scratch (behind a barrier: other waves may still be reading it) ----
    if constexpr (EPI == W_EPI_C1W) __syncthreads();
#if !(V4_ABL & 2)
    float* scratch = reinterpret_cast<float*>(lds + ((nsteps - 1) & 1) * V4_STAGE);
    if (done.R0 + 4 * V4_BQ <= p.M && done.n0 + V4_BN <= p.N)     // (workgroup-uniform) interior tile: no per-store bounds tests
      wino43_epilogue<EPI, true>(p, acc, scratch, done.R0, done.n0, wm, wn, lr, lh, done.tm);
    else
      wino43_epilogue<EPI, false>(p, acc, scratch, done.R0, done.n0, wm, wn, lr, lh, done.tm);
#endif
#endif
    if (!V4_PERSIST && nb < nwg) {                         // (no prefetch: plain sequence of tiles)
      __syncthreads();
      cur = setup(nb);
      issue(cur, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Weight gradient on V.  The kernel of tonal_wino43_tn.hip with its A side replaced: V tiles
// [6][8 quads][64 channels] arrive by LDS-DMA (3 pieces of 4 rows x 256 B per wave and K-step) into a
// 4-slot ring, three K-steps ahead of their use; the Y side (un-pooled dZ -> A dy, needs the arg-max bits)
// is staged through registers exactly as before.  Per thread and K-step this removes six 8-byte row
// loads, 26 transform VALU instructions and six ds_write_b64.  Rows of odd quads keep the two
// 32-channel halves swapped (conflict-free fragment reads without padding): applied on the SOURCE chunk.
// ------------------------------------------------------------------------------------------
constexpr int T4_BN = 64, T4_Q = 8;                       // C_out tile, quads per K-step (C_in tile: 32 MW)
constexpr int T4_PLANE = T4_Q * 64;                       // floats per transform plane
constexpr int T4_TILE = 6 * T4_PLANE;                     // floats per operand tile (12 KB)
constexpr int T4V_NA = 4;                                 // V ring slots
#ifndef T4V_NT_STORE
#define T4V_NT_STORE 1     // Vd is written once and read by a later kernel: stream it past the L2
#endif
#ifndef T4V_ABL
#define T4V_ABL 0          // timing-only ablations (results garbage): 1 no steady-state V pieces, 2 no steady-state Y loads,
#endif                     // 4 no Vd stores, 8 no Y transform / LDS store in the steady state
#ifndef T4V_SCHED
#define T4V_SCHED 0        // > 0: hand-specified issue order of a K-step with this many VALU per MFMA slot
#endif

// WVD: this launch covers the first C_in tile only and also writes Vd (see write_vd below); the other C_in tiles run in
// a second launch of the plain instantiation (a run-time branch around the Vd code splits the K-step into basic blocks
// the scheduler cannot interleave across: 43.6 -> 46.3 ms for every workgroup).  mt0 / mtn: first C_in tile, tile count.
//
// MW: waves along C_in.  MW = 2: 64 x 64 tile, 4 waves, two workgroups per CU, a channel PAIR of one quad per thread
// on the Y side.  MW = 4 (C_in a multiple of 128): 128 x 64 tile, 8 waves, one workgroup per CU; the Y tile is the same
// 8 quads x 64 channels but is now built by 512 threads (one channel each) for twice the MFMA work - the transform
// VALU per MFMA halves, which is what bounds this kernel (fp32 MFMA shares its issue rate with the vector ALU:
// r03_kernel_notes.md section 3).  The V tile is held as MW / 2 half-tiles of 64 channels in the layout of the 64-wide
// kernel, so pieces, swizzle and fragment reads are the same code.  Same k order per accumulator: bit-identical results.
template <bool WVD, int MW>
__global__ __launch_bounds__(128 * MW, MW == 2 ? 2 : 1) void wino43v_tn_kernel(const tl_tn_params p, int mt0, int mtn) {
  constexpr int NH = MW / 2;                     // 64-channel half-tiles of V per K-step
  constexpr int CPT = 4 / MW;                    // Y channels per thread
  constexpr int TPQ = 64 / CPT;                  // threads per quad row of the Y tile
  constexpr int BM = 32 * MW;
  __shared__ __attribute__((aligned(1024))) float lds[(T4V_NA * NH + 2) * T4_TILE];
  float* As = lds;                               // [4][NH][6][8][64]  V ring
  float* Bs = lds + T4V_NA * NH * T4_TILE;       // [2][6][8][64]      Y
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int ntn = (p.Ndim + T4_BN - 1) / T4_BN;
  const long long tiles = (long long)mtn * ntn;
  const long long nwg = tiles * p.splitk;
  long long bid = (long long)blockIdx.y * gridDim.x + blockIdx.x;
  {
    const long long q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int z = (int)(bid / tiles);
  const int tt = (int)(bid % tiles);
  const int m0 = (mt0 + tt / ntn) * BM, n0 = (tt % ntn) * T4_BN;

  const long long quads_all = p.Krows >> 2;
  const long long ksteps_all = (quads_all + T4_Q - 1) / T4_Q;
  const long long per = (ksteps_all + p.splitk - 1) / p.splitk;
  const long long ks_begin = z * per;
  long long ks_end = ks_begin + per;
  if (ks_end > ksteps_all) ks_end = ksteps_all;
  const int nsteps = ks_end > ks_begin ? (int)(ks_end - ks_begin) : 0;

  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // ---- V by LDS-DMA: piece 3 wave + t -> half-tile pc / 12, then pc % 12 -> (transform i = pc >> 1, quads
  // 4 (pc & 1) .. + 3); lane -> (quad offset lane >> 4, physical 16-byte chunk lane & 15); odd quads: halves swapped =
  // source chunk ^ 8
  const long long v_q0 = ks_begin * T4_Q;                           // first quad of this split
  const long long v_left = (p.A_rows - v_q0) * 6 * (long long)p.lda * 4 - (long long)m0 * 4;
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.A + v_q0 * 6 * (long long)p.lda + m0), 0, (int)(v_left < 0x7fffffffLL ? (v_left > 0 ? v_left : 0) : 0x7fffffffLL),
      0x00020000);
  unsigned vvoff[3], vdst[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int half = (wave * 3 + t) / 12, pc = (wave * 3 + t) % 12;
    const int i = pc >> 1, ql = (pc & 1) * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ ((ql & 1) << 3);
    vvoff[t] = (unsigned)((((long long)ql * 6 + i) * p.lda + half * 64 + chunk * 4) * 4);
    vdst[t] = (unsigned)((half * T4_TILE + i * T4_PLANE + (pc & 1) * 4 * 64) * 4);
  }
  const unsigned v_step = (unsigned)(T4_Q * 6 * p.lda * 4);        // bytes per K-step (host-checked to fit)
  auto issue_v = [&](int step) {
    char* base = reinterpret_cast<char*>(As) + (step & (T4V_NA - 1)) * (NH * T4_TILE * 4);
    const unsigned soff = (unsigned)step * v_step;
#pragma unroll
    for (int t = 0; t < 3; ++t) dma16(rsV, base + vdst[t], vvoff[t], soff);
  };

  // ---- Y staging (registers), as in wino43_tn_kernel ----
  using yv = std::conditional_t<CPT == 2, f32x2, float>;    // the CPT channels a thread transforms
  const int qi = tid / TPQ, c2 = tid % TPQ;                 // quad of the K-step, channel (pair)
  const int sw = ((c2 * CPT) ^ ((qi & 1) << 5));            // swizzled channel position inside the 64-wide row
  const long long b_last = p.B_rows - 1;
  const int ncol = n0 + c2 * CPT;
  const bool bnok = ncol < p.Ndim;
  const int ncolc = bnok ? ncol : n0;
  long long quad = ks_begin * T4_Q + qi;                    // quad the NEXT load fetches for this thread
  int tq = (int)((4 * quad) % p.Tp);                        // time index of its first conv row
  const int dstep = (4 * T4_Q) % p.Tp;
  long long ld_q0 = ks_begin * T4_Q;
  const unsigned b_toff = (unsigned)(qi * 2 * p.ldb + (ncolc - n0));
  const unsigned w_toff = (unsigned)(qi * 2 * p.ld_bbits + (ncolc >> 5));

  struct stage_regs {
    yv g[2];
    uint32_t wa, wb;  // arg-max words of the two pooled rows
    uint32_t ok;      // bit 0 / 1: pair a / b holds a valid gradient (bit 2: the pooled row in front of the quad)
    yv gp;            // write_vd only: pooled row 2 q - 1 (conv rows 4 q - 2, 4 q - 1), its arg-max word, the quad index
    uint32_t wp;
    int q;
  };
  stage_regs rP, rQ, rR;
  // The workgroups of the first C_in tile also WRITE the input-gradient operand of this stage: Vd[quad][6][C_out] = the
  // F(4,3) input transform of the un-pooled dZ rows 4 q - 2 .. 4 q + 3 (the rows the input gradient of conv rows
  // 4 q .. 4 q + 3 contracts with the flipped taps).  They already hold dZ rows 4 q .. 4 q + 3 for the Y transform; one more
  // pooled row and six 8-byte stores per thread and K-step turn the input-gradient pass into the transform-free V-form
  // kernel (wino43v_nt_kernel) without a pass of its own over G (19.7 GB written beside MFMA-bound work).
  constexpr bool write_vd = WVD;
  typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
  const unsigned vd_qstride = (unsigned)(6 * p.ld_vd * 4), vd_coff = (unsigned)((ncolc - n0) * 4);
  long long vd_left = write_vd ? (quads_all - v_q0) * 6 * (long long)p.ld_vd * 4 - (long long)n0 * 4 : 0;
  vd_left = vd_left < 0 ? 0 : (vd_left < 0x7fffffffLL ? vd_left : 0x7fffffffLL);
  const __amdgpu_buffer_rsrc_t rsVd = __builtin_amdgcn_make_buffer_rsrc(
      write_vd ? (void*)(p.vd + v_q0 * 6 * (long long)p.ld_vd + n0) : (void*)p.slab, 0, (int)vd_left, 0x00020000);
  yv bsum = {};

  auto load_regs = [&](auto FAST, stage_regs& r) {
    constexpr bool fast = decltype(FAST)::value;
    long long pa = 2 * quad, pb = 2 * quad + 1;
    bool va = bnok && tq < p.Tvalid, vb = bnok && tq + 2 < p.Tvalid;
    uint32_t okp = 0;
    if constexpr (write_vd) {
      long long pr = 2 * quad - 1;
      const bool vp = bnok && quad > 0 && tq >= 2 && tq - 2 < p.Tvalid && 4 * quad - 2 < p.Krows && pr <= b_last;
      pr = pr < 0 ? 0 : (pr < b_last ? pr : b_last);
      if constexpr (fast) {
        if constexpr (CPT == 2)
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r.gp) : "v"(p.B + pr * (long long)p.ldb + ncolc) : "memory");
        else
          asm volatile("global_load_dword %0, %1, off" : "=v"(r.gp) : "v"(p.B + pr * (long long)p.ldb + ncolc) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(r.wp) : "v"(p.bbits + pr * (long long)p.ld_bbits + (ncolc >> 5)) : "memory");
      } else {
        r.gp = *reinterpret_cast<const yv*>(p.B + pr * (long long)p.ldb + ncolc);
        r.wp = p.bbits[pr * (long long)p.ld_bbits + (ncolc >> 5)];
      }
      r.q = (int)quad;
      okp = vp ? 4u : 0u;
    }
    if constexpr (fast) {
      const float* bu = p.B + (ld_q0 * 2) * (long long)p.ldb + n0;
      const uint32_t* wu = p.bbits + (ld_q0 * 2) * (long long)p.ld_bbits;
      if constexpr (write_vd) {
        // Loads the compiler does not count (inline asm): with the Vd stores in the same vmcnt stream its own wait in
        // front of the transform comes out as vmcnt(9..10) - a drain of the three-step prefetch every K-step (the
        // launch ran at 45 % of the plain kernel's rate).  The counted wait is placed by hand in kstep.
        if constexpr (CPT == 2) {
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r.g[0]) : "v"(bu + b_toff) : "memory");
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r.g[1]) : "v"(bu + p.ldb + b_toff) : "memory");
        } else {
          asm volatile("global_load_dword %0, %1, off" : "=v"(r.g[0]) : "v"(bu + b_toff) : "memory");
          asm volatile("global_load_dword %0, %1, off" : "=v"(r.g[1]) : "v"(bu + p.ldb + b_toff) : "memory");
        }
        asm volatile("global_load_dword %0, %1, off" : "=v"(r.wa) : "v"(wu + w_toff) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(r.wb) : "v"(wu + p.ld_bbits + w_toff) : "memory");
      } else {
        r.g[0] = *reinterpret_cast<const yv*>(bu + b_toff);
        r.g[1] = *reinterpret_cast<const yv*>(bu + p.ldb + b_toff);
        r.wa = wu[w_toff];
        r.wb = wu[p.ld_bbits + w_toff];
      }
    } else {
      va = va && 4 * quad < p.Krows && pa <= b_last;
      vb = vb && 4 * quad + 2 < p.Krows && pb <= b_last;
      pa = pa < b_last ? pa : b_last;
      pb = pb < b_last ? pb : b_last;
      r.g[0] = *reinterpret_cast<const yv*>(p.B + pa * (long long)p.ldb + ncolc);
      r.g[1] = *reinterpret_cast<const yv*>(p.B + pb * (long long)p.ldb + ncolc);
      r.wa = p.bbits[pa * (long long)p.ld_bbits + (ncolc >> 5)];
      r.wb = p.bbits[pb * (long long)p.ld_bbits + (ncolc >> 5)];
    }
    r.ok = (va ? 1u : 0u) | (vb ? 2u : 0u) | okp;
    ld_q0 += T4_Q;
    quad += T4_Q;
    tq += dstep;
    if (tq >= p.Tp) tq -= p.Tp;
  };
  auto store_b = [&](const stage_regs& r, int buf) {
    const int sh = ncolc & 31;
    yv o[6], vd[6];
    auto at = [](auto& v, int c) -> float& {
      if constexpr (CPT == 2) return reinterpret_cast<float*>(&v)[c];
      else return v;
    };
    auto cat = [](const yv& v, int c) -> float {
      if constexpr (CPT == 2) return v[c];
      else return v;
    };
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const uint32_t ma = (uint32_t)__builtin_amdgcn_sbfe((int)r.wa, sh + c, 1);      // all ones if the odd row won
      const uint32_t mb = (uint32_t)__builtin_amdgcn_sbfe((int)r.wb, sh + c, 1);
      const uint32_t ua = (r.ok & 1u) ? __float_as_uint(cat(r.g[0], c)) : 0u, ub = (r.ok & 2u) ? __float_as_uint(cat(r.g[1], c)) : 0u;
      const float e_a = __uint_as_float(ua & ~ma), o_a = __uint_as_float(ua & ma);
      const float e_b = __uint_as_float(ub & ~mb), o_b = __uint_as_float(ub & mb);
      at(o[0], c) = e_a;
      at(o[1], c) = (e_a + o_a) + (e_b + o_b);
      at(o[2], c) = (e_a - o_a) + (e_b - o_b);
      at(o[3], c) = fmaf(4.f, fmaf(2.f, o_b, e_b), fmaf(2.f, o_a, e_a));
      at(o[4], c) = fmaf(4.f, fmaf(-2.f, o_b, e_b), fmaf(-2.f, o_a, e_a));
      at(o[5], c) = o_b;
      if constexpr (write_vd) {
        // d0..d5 = dZ rows 4 q - 2 .. 4 q + 3 = (e_p, o_p, e_a, o_a, e_b, o_b); B^T d as in wino43_xform_kernel
        const uint32_t mp = (uint32_t)__builtin_amdgcn_sbfe((int)r.wp, sh + c, 1);
        const uint32_t up = (r.ok & 4u) ? __float_as_uint(cat(r.gp, c)) : 0u;
        const float d0 = __uint_as_float(up & ~mp), d1 = __uint_as_float(up & mp);
        float v[6];
        vd_transform(d0, d1, e_a, o_a, e_b, o_b, v);
#pragma unroll
        for (int i = 0; i < 6; ++i) at(vd[i], c) = v[i];
      }
    }
    bsum += o[1];
    float* dst = Bs + buf * T4_TILE + qi * 64 + sw;
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<yv*>(dst + i * T4_PLANE) = o[i];
    if constexpr (write_vd && !(T4V_ABL & 4)) {
      // buffer stores relative to the first quad of this split: a lane with nothing to write (column past C_out, quad of
      // the padding) carries an offset past the resource and the hardware drops its store - no branch in the K-step (an
      // exec-mask branch around the stores splits it into basic blocks the scheduler cannot interleave across)
      const unsigned off = (bnok && r.q < (int)quads_all) ? (unsigned)(r.q - (int)v_q0) * vd_qstride + vd_coff : 0xfffffff0u;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if constexpr (CPT == 2)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, vd[i]), rsVd, off, (unsigned)(i * p.ld_vd * 4), T4V_NT_STORE ? 2 : 0);
        else
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vd[i]), rsVd, off, (unsigned)(i * p.ld_vd * 4), T4V_NT_STORE ? 2 : 0);
      }
    }
  };

  // ---- MFMA side: k-slice sl of a K-step = quads 2 sl (lanes 0-31) and 2 sl + 1 (lanes 32-63) ----
  const int a_off = (wm >> 1) * T4_TILE + lh * 64 + (((wm & 1) * 32 + lr) ^ (lh << 5));
  const int b_off = lh * 64 + ((wn * 32 + lr) ^ (lh << 5));
  float fa0[6], fb0[6], fa1[6], fb1[6];
  auto load_frag = [&](float (&fa)[6], float (&fb)[6], int abuf, int bbuf, int sl) {
    const float* a_s = As + abuf * (NH * T4_TILE) + sl * 128 + a_off;
    const float* b_s = Bs + bbuf * T4_TILE + sl * 128 + b_off;
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

  // One K-step: V(s + 3) and the Y registers of step s + 3 are requested at the top; Y(s + 1) is transformed and
  // written mid-step.  Closing wait: everything up to V(s + 1) must have landed - the younger operations of this
  // wave are the 4 Y loads + 3 pieces of steps s + 2 and s + 3 = 14 (tail steps drain completely).
  auto kstep = [&](auto TAIL, int s, stage_regs& r_ld, const stage_regs& r_st) {
    constexpr bool tail = decltype(TAIL)::value;
    const int abuf = s & (T4V_NA - 1), bbuf = s & 1;
    load_frag(fa0, fb0, abuf, bbuf, 0);
    if constexpr (!tail) {
      if (!(T4V_ABL & 2)) load_regs(std::true_type{}, r_ld);
      if (!(T4V_ABL & 1)) issue_v(s + 3);
    } else if (s + 3 < nsteps) {
      load_regs(std::false_type{}, r_ld);
      issue_v(s + 3);
    }
#if !T4V_SCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
    mfma6(fa1, fb1);                                        // slice 3 of the previous step
    load_frag(fa1, fb1, abuf, bbuf, 1);
    mfma6(fa0, fb0);
    load_frag(fa0, fb0, abuf, bbuf, 2);
    mfma6(fa1, fb1);
    if constexpr (write_vd) {
      // r_st was loaded (asm) two steps ago: 6 loads, then 3 pieces + 6 stores of that step, 15 operations of the last
      // step and the 9 loads / pieces issued at the top of this one = 33 younger operations may stay in flight.  Tail
      // steps drain completely (their own loads are compiler-counted, but r_st may still come from the asm loads).
      stage_regs& w = const_cast<stage_regs&>(r_st);
      if constexpr (!tail)
        asm volatile("s_waitcnt vmcnt(33)" : "+v"(w.g[0]), "+v"(w.g[1]), "+v"(w.gp), "+v"(w.wa), "+v"(w.wb), "+v"(w.wp));
      else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w.g[0]), "+v"(w.g[1]), "+v"(w.gp), "+v"(w.wa), "+v"(w.wb), "+v"(w.wp));
    }
    if ((!tail && !(T4V_ABL & 8)) || (tail && s + 1 < nsteps)) store_b(r_st, bbuf ^ 1);
    load_frag(fa1, fb1, abuf, bbuf, 3);
    mfma6(fa0, fb0);
#if T4V_SCHED
    // issue order of the step: the first fragment reads, then one vector-memory operation (4 Y loads, 3 V pieces),
    // two transform VALU and one fragment read per MFMA slot, the three Y stores late (0x008 MFMA, 0x002 VALU,
    // 0x010 vector memory, 0x100 / 0x200 LDS read / write)
    if constexpr (!tail) {
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int t = 0; t < 24; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (t < 7) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, T4V_SCHED, 0);
        if (t < 18) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (t >= 18 && t < 21) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
#endif
    // (a workgroup that also writes Vd issues 2 more loads and 6 stores per step - stores count in vmcnt too: the
    // younger operations are then the 6 stores of step s - 2 and 15 + 15 of steps s - 1 and s = 36)
    if constexpr (!tail) {
      if constexpr (write_vd) asm volatile("s_waitcnt vmcnt(36) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  using Y = std::true_type;
  using N = std::false_type;

#pragma unroll
  for (int i = 0; i < 6; ++i) fa1[i] = fb1[i] = 0.f;

  if (nsteps > 0) {
    issue_v(0);
    load_regs(N{}, rP);
    store_b(rP, 0);
    if (nsteps > 1) {
      issue_v(1);
      load_regs(N{}, rQ);
    }
    if (nsteps > 2) {
      issue_v(2);
      load_regs(N{}, rR);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  int s = 0;
  const bool whole = 2 * p.B_rows >= p.Krows;
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

  if (p.colsum != nullptr && m0 == 0) {
    __syncthreads();
    float* red = lds;
    *reinterpret_cast<yv*>(red + qi * 64 + c2 * CPT) = bsum;
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

// ------------------------------------------------------------------------------------------
// Weight gradient on V, 128 x 64 tile, 8 waves, one workgroup per CU: the product kernel where C_in % 128 == 0 and
// C_out % 64 == 0.  Same tile, same MFMA side and bit-identical results as wino43v_tn_kernel<., 4>; the Y side has no
// load that returns into registers:
//   * the pooled gradient rows of a K-step (16 rows x 64 channels, 4 KB) AND their arg-max words (16 rows x 8 bytes)
//     arrive raw by LDS-DMA, four steps ahead, in a 6-slot ring: waves 0-3 fetch one gradient piece each, waves 4-7 the
//     piece with the words (all four the same one: every wave issues exactly one piece, the step stays free of branches);
//   * a wave owns one quad, a lane one channel: one ds_read2st64_b32 fetches its pooled pair, a ds_read2_b64 at a
//     wave-uniform address the two 64-bit arg-max words, v_readfirstlane moves them to scalar registers - where the word
//     pair IS the lane mask of the un-pool select (v_cndmask on an SGPR pair; row validity folds in on the scalar ALU);
//   * the vector-memory operations that are left - 3 V pieces, 1 G / word piece and, for the workgroups that also write
//     Vd, the two 16-byte stores of the PREVIOUS step's transform (transposed over groups of four lanes and kept in
//     registers across the barrier) - are issued one per
//     MFMA in the first half of the step (sched_group_barrier), the transform runs in the second half.
// (Scalar loads for the words were tried first: they return out of order, so none may be pending at a counted LDS wait,
// which leaves them one half-step to complete - and they miss to HBM every step: 3.5 of 32.7 ms at conv2.)
// One K-step body serves every step: prefetches past the end of a split read memory the resource still covers (or
// zeros), the quads of a step past the end get all-zero masks, so the vmcnt arithmetic of the closing wait never
// changes: what may stay in flight are the operations of this step and the one before, 2 x (4 or 6).
// ------------------------------------------------------------------------------------------
#ifndef T8_SCHED
#define T8_SCHED 1
#endif
#ifndef T8_ABL
#define T8_ABL 0           // timing-only: 1 no V pieces, 2 no G piece, 4 no Vd stores, 8 no transform, 16 no barrier,
                           // 512 Vd stores issued with offsets past the resource (no memory traffic)
#endif
template <bool WVD>
__global__ __launch_bounds__(512, 1) void wino43v_tn8_kernel(const tl_tn_params p, int mt0, int mtn) {
  constexpr int NA = T4V_NA, NG = 6;             // V ring slots; raw ring slots (steps s - 1 .. s + 4 are live)
  constexpr int GW = 16 * 64, GT = GW + 64 * 4;  // floats per raw slot: 16 gradient rows, then 64 x 16 bytes of arg-max words
  __shared__ __attribute__((aligned(1024))) float lds[(NA * 2 + 2) * T4_TILE + NG * GT];
  float* As = lds;                               // [4][2][6][8][64]  V ring (two 64-channel half-tiles)
  float* Bs = lds + NA * 2 * T4_TILE;            // [2][6][8][64]     Y
  float* Gs = Bs + 2 * T4_TILE;                  // [6]{[16][64] raw pooled gradient rows, [64][4] words (row r: its two at 4 r)}
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int ntn = p.Ndim / T4_BN;
  const long long tiles = (long long)mtn * ntn;
  const long long nwg = tiles * p.splitk;
  long long bid = (long long)blockIdx.y * gridDim.x + blockIdx.x;
  {
    const long long q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  // (integer division runs on the vector ALU: what is derived from its result stays there unless it is moved back)
  const int z = __builtin_amdgcn_readfirstlane((int)(bid / tiles));
  const int tt = __builtin_amdgcn_readfirstlane((int)(bid % tiles));
  const int mi = __builtin_amdgcn_readfirstlane(tt / ntn);                          // C_in tile of this workgroup
  const int m0 = (mt0 + mi) * 128, n0 = __builtin_amdgcn_readfirstlane((tt % ntn) * T4_BN);

  const long long quads_all = p.Krows >> 2;
  const long long ksteps_all = (quads_all + T4_Q - 1) / T4_Q;
  const long long per = __builtin_amdgcn_readfirstlane((int)((ksteps_all + p.splitk - 1) / p.splitk));
  const long long ks_begin = z * per;
  long long ks_end = ks_begin + per;
  if (ks_end > ksteps_all) ks_end = ksteps_all;
  const int nsteps = ks_end > ks_begin ? (int)(ks_end - ks_begin) : 0;

  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  auto clip31 = [](long long v) { return (int)(v < 0 ? 0 : (v < 0x7fffffffLL ? v : 0x7fffffffLL)); };
  // ---- V by LDS-DMA (as wino43v_tn_kernel<., 4>): piece 3 wave + t -> half-tile, transform, four quads
  const long long v_q0 = ks_begin * T4_Q;
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.A + v_q0 * 6 * (long long)p.lda + m0), 0, clip31((p.A_rows - v_q0) * 6 * (long long)p.lda * 4 - (long long)m0 * 4),
      0x00020000);
  unsigned vvoff[3], vdst[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int half = (wave * 3 + t) / 12, pc = (wave * 3 + t) % 12;
    const int i = pc >> 1, ql = (pc & 1) * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ ((ql & 1) << 3);
    vvoff[t] = (unsigned)((((long long)ql * 6 + i) * p.lda + half * 64 + chunk * 4) * 4);
    vdst[t] = (unsigned)((half * T4_TILE + i * T4_PLANE + (pc & 1) * 4 * 64) * 4);
  }
  const unsigned v_step = (unsigned)(T4_Q * 6 * p.lda * 4);
  auto issue_v = [&](int step) {
    char* base = reinterpret_cast<char*>(As) + (step & (NA - 1)) * (2 * T4_TILE * 4);
    const unsigned soff = (unsigned)step * v_step;
#pragma unroll
    for (int t = 0; t < 3; ++t) dma16(rsV, base + vdst[t], vvoff[t], soff);
  };
  // ---- raw pooled gradient rows and arg-max words by LDS-DMA.  Waves 0-3: gradient piece wave = rows 4 t .. 4 t + 3 of
  // the step's 16, lane -> (row lane >> 4, 16-byte chunk lane & 15), LDS image row-major [16][64].  Waves 4-7: the word
  // piece, lane r < 16 -> 16 bytes from the first word of this column tile in row r (two words used), lanes >= 16 carry
  // an offset past the resource (zeros, no memory access)
  const long long g_r0 = ks_begin * 2 * T4_Q;
  const bool wpiece = wave >= 4;
  const int ldw4 = p.ld_bbits * 4;
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(
      wpiece ? (void*)(p.bbits + g_r0 * (long long)p.ld_bbits + (n0 >> 5)) : (void*)(p.B + g_r0 * (long long)p.ldb + n0), 0,
      wpiece ? clip31((p.B_rows - g_r0) * (long long)ldw4 - (long long)(n0 >> 5) * 4)
             : clip31((p.B_rows - g_r0) * (long long)p.ldb * 4 - (long long)n0 * 4),
      0x00020000);
  const unsigned gvoff = wpiece ? (lane < 16 ? (unsigned)(lane * ldw4) : 0xfffffff0u)
                                : (unsigned)((((wave & 3) * 4 + (lane >> 4)) * (long long)p.ldb + (lane & 15) * 4) * 4);
  const unsigned gdst = wpiece ? (unsigned)(GW * 4) : (unsigned)((wave & 3) * 256 * 4);
  const unsigned g_step = wpiece ? (unsigned)(2 * T4_Q * ldw4) : (unsigned)(2 * T4_Q * p.ldb * 4);
  auto issue_g = [&](int step, int slot) {
    dma16(rsG, reinterpret_cast<char*>(Gs) + slot * (GT * 4) + gdst, gvoff, (unsigned)step * g_step);
  };
  auto next6 = [](int v) { return v == NG - 1 ? 0 : v + 1; };

  // ---- Y side: wave = quad of the K-step, lane = channel ----
  constexpr bool write_vd = WVD;
  const int sw = lane ^ ((wave & 1) << 5);                  // swizzled channel position inside the 64-wide row
  // (integer division runs on the vector ALU: without the readfirstlane everything derived from tq stays there)
  int tq = __builtin_amdgcn_readfirstlane((int)((4 * (v_q0 + wave)) % p.Tp));   // first conv row of the NEXT quad to transform
  const int dstep = __builtin_amdgcn_readfirstlane((4 * T4_Q) % p.Tp);
  float bsum = 0.f;
  const unsigned vd_qstride = (unsigned)(6 * p.ld_vd * 4);
  const __amdgpu_buffer_rsrc_t rsVd = __builtin_amdgcn_make_buffer_rsrc(
      write_vd ? (void*)(p.vd + v_q0 * 6 * (long long)p.ld_vd + n0) : (void*)p.slab, 0,
      write_vd ? clip31((quads_all - v_q0) * 6 * (long long)p.ld_vd * 4 - (long long)n0 * 4) : 0, 0x00020000);
  // Vd: the mtn workgroups of one (split, C_out tile) build the same Y; they take turns at Vd - the workgroup of C_in
  // tile mi transforms and stores the quads of the steps sd with sd % mtn == mi, in a block of its own behind the Y
  // transform (a wave-uniform branch: ~40 vector instructions and two 16-byte stores every mtn-th step).  The quad is
  // transposed inside each group of four lanes first, so that a lane holds four consecutive channels of ONE transform
  // row: rows 0..3 by lane & 3, rows 4, 5 by the lanes with lane & 3 < 2.  (Measured before: one launch for C_in tile 0
  // that wrote all of Vd ran at 0.60 of the MFMA peak against 0.80 for the others - 19.7 GB of stores in 13.5 ms at
  // conv2; fewer, wider stores alone did not help, thinning the stream over the whole op does.)
  const unsigned vdA_lane = (unsigned)((lane & 3) * p.ld_vd * 4 + (lane & ~3) * 4);
  const unsigned vdB_lane = (unsigned)((4 + (lane & 3)) * p.ld_vd * 4 + (lane & ~3) * 4);
  auto sel = [](unsigned long long m, float v) -> float {   // lane's bit of m set ? v : 0
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
    return r;
  };
  // Y(sd) -> Bs[sd & 1]; with WVD also Vd of the step's quads -> pend.  Row logic against four loop-invariant quad
  // limits (a quad q of step sd < nsteps of this split is below q_end), 32-bit scalar arithmetic, no short-circuit
  // evaluation (a branch would split the K-step into basic blocks):
  //   pooled row 2 q     usable: q < min(quads, (B_rows + 1) / 2, q_end)      [4 q < Krows, 2 q <= B_rows - 1]
  //   pooled row 2 q + 1 usable: q < min(quads, B_rows / 2, q_end)
  //   pooled row 2 q - 1 usable: 0 < q < min(quads + 1, (B_rows + 2) / 2, q_end)
  const int blast = (int)(p.B_rows - 1), nquads = (int)quads_all, q_first = (int)v_q0;
  const int q_end = q_first + nsteps * T4_Q;
  auto min3 = [](int a, int b, int c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); };
  const int qa_lim = min3(nquads, (blast + 2) / 2, q_end), qb_lim = min3(nquads, (blast + 1) / 2, q_end);
  const int qp_lim = min3(nquads + 1, (blast + 3) / 2, q_end), qs_lim = nquads < q_end ? nquads : q_end;
  struct y_in {
    float ga, gb;
    unsigned long long m_oa, m_ea, m_ob, m_eb;
    int tq;               // time index of the quad's first conv row
  };
  struct y_out {
    float e_a, o_a, e_b, o_b;
  };
  auto uni = [](unsigned long long v) -> unsigned long long {   // a wave-uniform value out of vector registers
    return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
           ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
  };
  // slot / pslot: ring slots of steps sd and sd - 1
  auto fetch_y = [&](int sd, int slot, int pslot, y_in& y) {
    const int q = q_first + sd * T4_Q + wave;
    const int va = (tq < p.Tvalid) & (q < qa_lim), vb = (tq + 2 < p.Tvalid) & (q < qb_lim);
    const float* gs = Gs + slot * GT;
    y.ga = gs[wave * 128 + lane];
    y.gb = gs[wave * 128 + 64 + lane];
    const unsigned long long* ws = reinterpret_cast<const unsigned long long*>(gs + GW + wave * 8);   // rows 2 wave, 2 wave + 1
    const unsigned long long wa = uni(ws[0]), wb = uni(ws[2]);
    y.m_oa = va ? wa : 0ull;
    y.m_ea = va ? ~wa : 0ull;
    y.m_ob = vb ? wb : 0ull;
    y.m_eb = vb ? ~wb : 0ull;
    y.tq = tq;
    tq += dstep;
    if (tq >= p.Tp) tq -= p.Tp;
  };
  auto compute_y = [&](int sd, const y_in& y) -> y_out {
    const float o_a = sel(y.m_oa, y.ga), e_a = sel(y.m_ea, y.ga);
    const float o_b = sel(y.m_ob, y.gb), e_b = sel(y.m_eb, y.gb);
    float o[6];
    o[0] = e_a;
    o[1] = (e_a + o_a) + (e_b + o_b);
    o[2] = (e_a - o_a) + (e_b - o_b);
    o[3] = fmaf(4.f, fmaf(2.f, o_b, e_b), fmaf(2.f, o_a, e_a));
    o[4] = fmaf(4.f, fmaf(-2.f, o_b, e_b), fmaf(-2.f, o_a, e_a));
    o[5] = o_b;
    bsum += o[1];
    float* dst = Bs + (sd & 1) * T4_TILE + wave * 64 + sw;
#pragma unroll
    for (int i = 0; i < 6; ++i) dst[i * T4_PLANE] = o[i];
    return y_out{e_a, o_a, e_b, o_b};
  };
  typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
  // Vd of the quads of step sd (slot / pslot: ring slots of steps sd, sd - 1)
  auto vd_part = [&](int sd, int slot, int pslot, int tqc, const y_out& u) {
    const int q = q_first + sd * T4_Q + wave;
    // rows 4 q - 2, 4 q - 1 = pooled row 2 q - 1: the row in front of this wave's pair, for wave 0 the last row of the
    // previous step's tile (still in the ring; in front of the first step: the pieces the prologue fetched)
    const int vp = (q > 0) & (tqc >= 2) & (tqc - 2 < p.Tvalid) & (q < qp_lim);
    const int prow = wave > 0 ? slot * GT + (2 * wave - 1) * 64 : pslot * GT + 15 * 64;
    const int pwrd = wave > 0 ? slot * GT + GW + (2 * wave - 1) * 4 : pslot * GT + GW + 15 * 4;
    const float gp = Gs[prow + lane];
    const unsigned long long wp = uni(*reinterpret_cast<const unsigned long long*>(Gs + pwrd));
    const float d1 = sel(vp ? wp : 0ull, gp), d0 = sel(vp ? ~wp : 0ull, gp);
    float v[6];
    vd_transform(d0, d1, u.e_a, u.o_a, u.e_b, u.o_b, v);
    // 4 x 4 transpose over the lanes of a group (register n of lane r <- register r of lane n): exchange with lane ^ 1
    // on the register pairs (0,1) (2,3), then with lane ^ 2 on (0,2) (1,3)
    const bool odd = lane & 1, hi = lane & 2;
    auto xch = [](float x, auto CTRL) -> float {
      return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), decltype(CTRL)::value, 0xf, 0xf, true));
    };
    using X1 = std::integral_constant<int, 0xB1>;         // quad_perm [1,0,3,2]
    using X2 = std::integral_constant<int, 0x4E>;         // quad_perm [2,3,0,1]
    auto stage = [&](float& a, float& b, bool up, auto CTRL) {
      const float r = xch(up ? a : b, CTRL);
      a = up ? r : a;
      b = up ? b : r;
    };
    stage(v[0], v[1], odd, X1{});
    stage(v[2], v[3], odd, X1{});
    stage(v[4], v[5], odd, X1{});
    stage(v[0], v[2], hi, X2{});
    stage(v[1], v[3], hi, X2{});
    const f32x4 va4 = {v[0], v[1], v[2], v[3]};
    const f32x4 vb4 = {v[4], v[5], xch(v[4], X2{}), xch(v[5], X2{})};   // (rows 4, 5: lanes 0, 1 of a group)
    const bool ok = q < qs_lim && !(T8_ABL & 512);
    const unsigned qoff = (unsigned)(q - q_first) * vd_qstride;
    if (!(T8_ABL & 4)) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, va4), rsVd, ok ? qoff + vdA_lane : 0xfffffff0u, 0u,
                                             T4V_NT_STORE ? 2 : 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, vb4), rsVd,
                                             ok && (lane & 3) < 2 ? qoff + vdB_lane : 0xfffffff0u, 0u, T4V_NT_STORE ? 2 : 0);
    }
  };

  // ---- MFMA side (as wino43v_tn_kernel<., 4>) ----
  const int a_off = (wm >> 1) * T4_TILE + lh * 64 + (((wm & 1) * 32 + lr) ^ (lh << 5));
  const int b_off = lh * 64 + ((wn * 32 + lr) ^ (lh << 5));
  float fa0[6], fb0[6], fa1[6], fb1[6], fa2[6], fb2[6], fac[6], fbc[6];   // slices 0..2 of a step; slice 3, carried
  auto load_frag = [&](float (&fa)[6], float (&fb)[6], int abuf, int bbuf, int sl) {
    const float* a_s = As + abuf * (2 * T4_TILE) + sl * 128 + a_off;
    const float* b_s = Bs + bbuf * T4_TILE + sl * 128 + b_off;
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
#pragma unroll
  for (int i = 0; i < 6; ++i) fac[i] = fbc[i] = 0.f;

  {   // (also for an empty split: every access is clamped by its resource, the masks are all zero)
    issue_v(0);
    issue_v(1);
    issue_v(2);
    issue_g(0, 0);
    issue_g(1, 1);
    issue_g(2, 2);
    issue_g(3, 3);
    if (write_vd && g_r0 > 0) {
      // slot of step -1: pooled rows g_r0 - 4 .. g_r0 - 1 -> its rows 12 .. 15 (one gradient piece); the word piece
      // covers the slot's 16 rows, lanes 12 .. 15 fetch (g_r0 is a multiple of 16)
      const int back = wpiece ? 16 : 4;
      const long long left = p.B_rows - (g_r0 - back) < back ? p.B_rows - (g_r0 - back) : (long long)back;
      const __amdgpu_buffer_rsrc_t rsGm = __builtin_amdgcn_make_buffer_rsrc(
          wpiece ? (void*)(p.bbits + (g_r0 - 16) * (long long)p.ld_bbits + (n0 >> 5)) : (void*)(p.B + (g_r0 - 4) * (long long)p.ldb + n0), 0,
          wpiece ? clip31(left * ldw4 - (long long)(n0 >> 5) * 4) : clip31(left * p.ldb * 4 - (long long)n0 * 4), 0x00020000);
      dma16(rsGm, reinterpret_cast<char*>(Gs) + ((NG - 1) * GT + (wpiece ? GW : 12 * 64)) * 4,
            wpiece ? (lane >= 12 && lane < 16 ? (unsigned)(lane * ldw4) : 0xfffffff0u)
                   : (unsigned)(((lane >> 4) * (long long)p.ldb + (lane & 15) * 4) * 4), 0u);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      y_in y0;
      fetch_y(0, 0, NG - 1, y0);
      const y_out u0 = compute_y(0, y0);
      if (write_vd && mi == 0) vd_part(0, 0, NG - 1, y0.tq, u0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  int slot1 = 1, slot4 = 4;                                 // ring slots of steps s + 1 and s + 4
  int turn = mtn > 1 ? 1 : 0;                               // (s + 1) % mtn: whose turn it is to write Vd for step s + 1
  for (int s = 0; s < nsteps; ++s) {
    const int abuf = s & (NA - 1), bbuf = s & 1;
    const int slot0 = slot1 == 0 ? NG - 1 : slot1 - 1;
    __builtin_amdgcn_sched_barrier(0);
    if (!(T8_ABL & 1)) issue_v(s + 3);
    if (!(T8_ABL & 2)) issue_g(s + 4, slot4);
    load_frag(fa0, fb0, abuf, bbuf, 0);
    load_frag(fa1, fb1, abuf, bbuf, 1);
    mfma6(fac, fbc);                                        // slice 3 of the previous step
    mfma6(fa0, fb0);
#if T8_SCHED
    // first half of the step: one vector-memory operation behind each MFMA while there are any, the fragment reads of
    // slices 0 and 1 (the compiler pairs them across the two slices: 12 ds_read2st64_b32) behind the MFMAs of the carried
    // slice (0x008 MFMA, 0x010 vector memory, 0x100 LDS read)
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (t < 4) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      if (t < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    // second half: the transform of step s + 1 beside the MFMAs of slices 1 and 2
    y_in yn;
    y_out un = {};
    if (!(T8_ABL & 8)) fetch_y(s + 1, slot1, slot0, yn);
    load_frag(fa2, fb2, abuf, bbuf, 2);
    load_frag(fac, fbc, abuf, bbuf, 3);
    mfma6(fa1, fb1);
    if (!(T8_ABL & 8)) un = compute_y(s + 1, yn);
    if constexpr (write_vd) {
      if (turn == mi && s + 1 < nsteps && !(T8_ABL & 8)) vd_part(s + 1, slot1, slot0, yn.tq, un);
      turn = turn + 1 == mtn ? 0 : turn + 1;
    }
    mfma6(fa2, fb2);
    __builtin_amdgcn_sched_barrier(0);                      // (the closing wait would be hoisted over these MFMAs)
    // everything issued up to step s - 2 has landed: V(s + 1), G(s + 2).  In flight: the 4 pieces of this step and of the
    // one before (a Vd store among them only makes the wait reach further back)
    __builtin_amdgcn_s_waitcnt(0x0078);                           // vmcnt(8) lgkmcnt(0)
#if !(T8_ABL & 16)
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
    slot1 = next6(slot1);
    slot4 = next6(slot4);
  }
  mfma6(fac, fbc);

  if (p.colsum != nullptr && m0 == 0) {
    __syncthreads();
    float* red = lds;
    red[wave * 64 + lane] = bsum;
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < T4_Q; ++q) t += red[q * 64 + tid];
      p.colsum[(long long)z * p.Ndim + n0 + tid] = t;
    }
  }
  float* out = p.slab + (long long)z * p.slab_stride;
  const int col = n0 + wn * 32 + lr;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      out[((long long)i * p.Mdim + m) * (long long)p.ldc + col] = acc[i][e];
    }
}

// F(4,3) taps of a 3-tap filter, torch (O, I, 3) -> forward [6][O][ld_f] / input-gradient [6][I][ld_d] (taps flipped)
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

extern "C" int tl_wino43_wgrad_finalize(const float* red, float* gw, int O, int I, int ld, void* stream) {
  using namespace tl;
  TL_REQUIRE(red && gw && O > 0 && I > 0 && ld >= O, "wino43_wgrad_finalize: bad arguments");
  const long long n = (long long)O * I;
  hipLaunchKernelGGL(wino43_wgrad_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     red, gw, O, I, ld);
  return check_launch("wino43_wgrad_finalize");
}

extern "C" int tl_wino43_input_transform(const float* P, float* V, int64_t rows, int Tp, int C, int ldp, int ldv,
                                         void* stream) {
  using namespace tl;
  TL_REQUIRE(P && V, "wino43_input_transform: null pointer");
  TL_REQUIRE(rows > 0 && Tp > 0 && Tp % 4 == 0 && rows % Tp == 0, "wino43_input_transform: rows must be whole sequences of Tp %% 4 == 0 rows");
  TL_REQUIRE(C > 0 && C % 4 == 0 && ldp >= C && ldv >= C && ldp % 4 == 0 && ldv % 4 == 0, "wino43_input_transform: C/ldp/ldv must be multiples of 4");
  const long long nq = rows / 4;
  const long long n = nq * (C / 4);
  TL_REQUIRE((n + 255) / 256 < (1LL << 31), "wino43_input_transform: grid too large");
  hipLaunchKernelGGL(wino43_xform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, V, nq, Tp,
                     C, ldp, ldv);
  return check_launch("wino43_input_transform");
}

// NT passes on a pre-transformed operand (loader 2): A = V[quad][6][lda], A_rows = quads in V, M = output rows (4 per
// quad).  Forward: V of the stage input, POOL epilogue.  Input gradient: V of the un-pooled dZ rows 4q-2 .. 4q+3 (written
// by tl_conv3_wino43v_tn), taps = the flipped / transposed set, MASK or fused-conv1-weight-gradient epilogue.
extern "C" int tl_conv3_wino43v_nt(const tl_nt_params* pp, void* stream) {
  using namespace tl;
  TL_REQUIRE(pp != nullptr, "wino43v_nt: null params");
  const tl_nt_params& p = *pp;
  TL_REQUIRE(p.A && p.Bw && (p.out || p.epilogue == W_EPI_C1W || p.epilogue == W_EPI_POOLV), "wino43v_nt: null V/Bw/out");
  TL_REQUIRE(p.loader == W_LOAD_V, "wino43v_nt: loader 2 (pre-transformed operand) only");
  TL_REQUIRE(p.J == 3 && p.splitk <= 1, "wino43v_nt: 3 taps, no split-K");
  TL_REQUIRE(p.M > 0 && p.M % 4 == 0 && p.N > 0 && p.K >= 16 && p.K % 16 == 0, "wino43v_nt: M %% 4, K %% 16 needed");
  TL_REQUIRE(p.A_rows >= p.M / 4, "wino43v_nt: V holds fewer quads than M / 4");
  TL_REQUIRE(p.lda >= p.K && p.ldb >= p.K && p.lda % 4 == 0 && p.ldb % 4 == 0, "wino43v_nt: bad leading dimensions");
  TL_REQUIRE(p.Tp > 0 && p.Tp % 4 == 0, "wino43v_nt: Tp must be a positive multiple of 4");
  TL_REQUIRE(6LL * p.N * p.ldb * 4 < (1LL << 31), "wino43v_nt: tap set larger than a buffer resource");
  TL_REQUIRE(128LL * 6 * p.lda * 4 + 4LL * p.K < (1LL << 31), "wino43v_nt: tile span too large");
  const long long nwg = ((p.M + 4 * V4_BQ - 1) / (4 * V4_BQ)) * ((p.N + V4_BN - 1) / V4_BN);
  TL_REQUIRE(nwg < (1LL << 31), "wino43v_nt: grid too large");
#if V4_LEAN
  // (the epilogues of tonal_wino43v_epi.h: a wave's 32 columns are in or out of the matrix together; 32-bit row arithmetic)
  TL_REQUIRE(p.N % 32 == 0 && p.M + 4 * V4_BQ < (1LL << 31), "wino43v_nt: N %% 32 == 0 and M < 2^31 - 512 needed");
  TL_REQUIRE(p.slope >= 0.f && p.slope <= 1.f, "wino43v_nt: LeakyReLU slope must lie in [0, 1]");
  TL_REQUIRE(p.A_rows >= ((p.M + 4 * V4_BQ - 1) / (4 * V4_BQ)) * V4_BQ, "wino43v_nt: V must hold whole 128-quad tiles (pad it with zero quads)");
#endif
  hipStream_t st = (hipStream_t)stream;
#if V4_PERSIST
  const long long ngrid = nwg < 256 * V4_PERSIST ? nwg : 256 * V4_PERSIST;      // one workgroup per CU (144 KB of LDS each)
#else
  const long long ngrid = nwg;
#endif
  if (p.epilogue == W_EPI_POOL) {
    TL_REQUIRE(p.row_shift == 0 && p.out && p.ldo >= p.N, "wino43v_nt: forward needs row_shift 0 and an output");
    TL_REQUIRE(p.obits != nullptr && p.Tvalid % 2 == 0 && p.Tvalid <= p.Tp, "wino43v_nt: POOL needs obits and an even Tvalid");
    TL_REQUIRE(p.N % 32 == 0 && p.ld_obits * 32 >= p.N, "wino43v_nt: POOL needs N %% 32 == 0");
    hipLaunchKernelGGL((wino43v_nt_kernel<W_EPI_POOL>), dim3((unsigned)ngrid), dim3(512), 0, st, p);
#if V4_LEAN
  } else if (p.epilogue == W_EPI_POOLV) {
    TL_REQUIRE(p.row_shift == 0 && (p.out == nullptr || p.ldo >= p.N), "wino43v_nt: forward needs row_shift 0");
    TL_REQUIRE(p.obits != nullptr && p.Tvalid % 2 == 0 && p.Tvalid <= p.Tp, "wino43v_nt: POOLV needs obits and an even Tvalid");
    TL_REQUIRE(p.ld_obits * 32 >= p.N && p.Tp % 8 == 0, "wino43v_nt: POOLV needs Tp %% 8 == 0 (output quads inside one sequence)");
    TL_REQUIRE(p.vout && p.vhalo && p.ld_vout >= p.N && p.vout_quads >= p.M / 8, "wino43v_nt: POOLV needs vout (>= M / 8 quads) and vhalo");
    TL_REQUIRE(64LL * 6 * p.ld_vout * 4 < (1LL << 31), "wino43v_nt: ld_vout too large");
    hipLaunchKernelGGL((wino43v_nt_kernel<W_EPI_POOLV>), dim3((unsigned)ngrid), dim3(512), 0, st, p);
#endif
  } else if (p.epilogue == W_EPI_MASK) {
    TL_REQUIRE(p.row_shift == -2 && p.ldo >= p.N, "wino43v_nt: input gradient needs row_shift -2");
#if V4_LEAN
    TL_REQUIRE(p.auxbits != nullptr, "wino43v_nt: MASK needs auxbits (the sign bits of the stage input)");
#else
    TL_REQUIRE(p.aux != nullptr || p.auxbits != nullptr, "wino43v_nt: MASK needs aux or auxbits");
#endif
    hipLaunchKernelGGL((wino43v_nt_kernel<W_EPI_MASK>), dim3((unsigned)ngrid), dim3(512), 0, st, p);
  } else if (p.epilogue == W_EPI_C1W) {
    TL_REQUIRE(p.row_shift == -2, "wino43v_nt: input gradient needs row_shift -2");
    TL_REQUIRE(p.auxbits && p.c1x && p.c1bits && p.c1partial, "wino43v_nt: epilogue 4 needs auxbits, c1x, c1bits, c1partial");
    TL_REQUIRE(p.c1kt >= 1 && p.c1kt <= 3 && p.c1T >= 2 * p.Tvalid + 2, "wino43v_nt: epilogue 4: 1..3 taps, c1T >= 2*Tvalid + 2");
    hipLaunchKernelGGL((wino43v_nt_kernel<W_EPI_C1W>), dim3((unsigned)ngrid), dim3(512), 0, st, p);
  } else {
    set_error("wino43v_nt: unsupported epilogue %d", p.epilogue);
    return TL_EINVAL;
  }
  return check_launch("wino43v_nt");
}

extern "C" int tl_wino43_v_fixup(float* V, const float* vhalo, int64_t quads, int64_t tiles, int Tq, int C, int ldv, void* stream) {
  using namespace tl;
  TL_REQUIRE(V && vhalo, "wino43_v_fixup: null pointer");
  TL_REQUIRE(quads > 0 && tiles > 0 && Tq > 0 && Tq % 4 == 0, "wino43_v_fixup: quads, tiles > 0 and Tq %% 4 == 0 needed");
  TL_REQUIRE(C > 0 && C % 4 == 0 && ldv >= C && ldv % 4 == 0, "wino43_v_fixup: C / ldv must be multiples of 4");
  const long long n = (long long)tiles * (C / 4);
  TL_REQUIRE((n + 255) / 256 < (1LL << 31), "wino43_v_fixup: grid too large");
  hipLaunchKernelGGL(wino43_v_fixup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, V, vhalo,
                     (long long)quads, (long long)tiles, Tq, C, ldv);
  return check_launch("wino43_v_fixup");
}

// weight gradient on V (a_form 1): A = V[quad][6][lda], A_rows = quads held by V (a whole number of 8-quad K-steps)
extern "C" int tl_conv3_wino43v_tn(const tl_tn_params* pp, void* stream) {
  using namespace tl;
  TL_REQUIRE(pp != nullptr, "wino43v_tn: null params");
  tl_tn_params p = *pp;
  if (p.splitk < 1) p.splitk = 1;
  TL_REQUIRE(p.A && p.B && p.slab && p.bbits, "wino43v_tn: null V/B/bbits/slab");
  TL_REQUIRE(p.J == 3 && p.loader == 1, "wino43v_tn: 3 taps, UNPOOL loader only");
  TL_REQUIRE(p.Krows > 0 && p.Krows % 4 == 0 && p.Mdim > 0 && p.Ndim > 0, "wino43v_tn: bad sizes (Krows %% 4 must be 0)");
  TL_REQUIRE(p.Krows + 64 < (1LL << 31), "wino43v_tn: more than 2^31 reduction rows");
  TL_REQUIRE(p.B_rows > 0, "wino43v_tn: empty operand");
  TL_REQUIRE(p.Mdim % 64 == 0 && p.Ndim % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0, "wino43v_tn: Mdim %% 64, Ndim/ld %% 4 needed");
  TL_REQUIRE(p.lda >= p.Mdim && p.ldb >= p.Ndim && p.ldc >= p.Ndim, "wino43v_tn: leading dimension too small");
  TL_REQUIRE(p.Tp > 0 && p.Tp % 4 == 0 && p.Tvalid % 2 == 0, "wino43v_tn: Tp %% 4 == 0 and an even Tvalid needed");
  TL_REQUIRE(p.ld_bbits * 32 >= p.Ndim, "wino43v_tn: bbits row too short");
  TL_REQUIRE(p.splitk <= 65535, "wino43v_tn: splitk too large");
  TL_REQUIRE(p.splitk == 1 || p.slab_stride >= 6LL * p.Mdim * p.ldc, "wino43v_tn: slab_stride smaller than 6*Mdim*ldc");
  const long long ksteps_all = ((p.Krows >> 2) + T4_Q - 1) / T4_Q;
  TL_REQUIRE(p.A_rows >= ksteps_all * T4_Q, "wino43v_tn: V must hold whole 8-quad K-steps (pad it with zero quads)");
  const long long per = (ksteps_all + p.splitk - 1) / p.splitk;
  TL_REQUIRE((per + 4) * (long long)T4_Q * 6 * p.lda * 4 < (1LL << 31), "wino43v_tn: a reduction split spans more than 2 GB of V: raise splitk");
  TL_REQUIRE(p.vd == nullptr || (p.ld_vd >= p.Ndim && p.ld_vd % 2 == 0), "wino43v_tn: ld_vd must cover Ndim");
  TL_REQUIRE(p.vd == nullptr || (per + 4) * (long long)T4_Q * 6 * p.ld_vd * 4 < (1LL << 31), "wino43v_tn: a reduction split spans more than 2 GB of Vd: raise splitk");
  // C_in tile: 128 (8 waves) when C_in allows it, 64 (4 waves) otherwise or on request (p.bm; 127 = the 128-wide tile on
  // the kernel that stages Y through registers - the A/B partner of wino43v_tn8_kernel)
  TL_REQUIRE(p.bm == 0 || p.bm == 64 || ((p.bm == 128 || p.bm == 127) && p.Mdim % 128 == 0),
             "wino43v_tn: bm must be 0, 64, or 127 / 128 with Mdim %% 128 == 0");
  const bool dma8_ok = p.Mdim % 128 == 0 && p.Ndim % 64 == 0 && p.ld_bbits % 2 == 0 && p.B_rows < (1LL << 30) &&
                       (per + 5) * 2LL * T4_Q * p.ldb * 4 < (1LL << 31);
  TL_REQUIRE(p.bm != 128 || dma8_ok, "wino43v_tn: bm 128 needs Mdim %% 128 == 0, Ndim %% 64 == 0, an even ld_bbits");
  const bool dma8 = p.bm == 128 || (p.bm == 0 && dma8_ok);
  const int bm = p.bm == 64 ? 64 : (p.bm ? 128 : (p.Mdim % 128 == 0 ? 128 : 64));
  const int ntm = (p.Mdim + bm - 1) / bm, ntn = (p.Ndim + T4_BN - 1) / T4_BN;
  TL_REQUIRE((long long)ntm * ntn < (1LL << 31), "wino43v_tn: grid too large");
  hipStream_t st = (hipStream_t)stream;
  auto launch = [&](auto WVD, int mt0, int mtn) {
    constexpr bool wvd = decltype(WVD)::value;
    const dim3 grid((unsigned)(mtn * ntn), (unsigned)p.splitk, 1);
    if (dma8) hipLaunchKernelGGL((wino43v_tn8_kernel<wvd>), grid, dim3(512), 0, st, p, mt0, mtn);
    else if (bm == 128) hipLaunchKernelGGL((wino43v_tn_kernel<wvd, 4>), grid, dim3(512), 0, st, p, mt0, mtn);
    else hipLaunchKernelGGL((wino43v_tn_kernel<wvd, 2>), grid, dim3(256), 0, st, p, mt0, mtn);
  };
  if (dma8) {
    // one launch; with Vd its workgroups take turns at it (p.part 1 = "the Vd launch only" has nothing to do here)
    if (p.part == 1) return TL_OK;
    if (p.vd != nullptr) launch(std::true_type{}, 0, ntm);
    else launch(std::false_type{}, 0, ntm);
  } else if (p.vd != nullptr) {
    // first C_in tile: the instantiation that also writes Vd; the other tiles: the plain one.  p.part selects one of the
    // two launches (1: the Vd tile, 2: the rest) so a caller can put them on different streams; 0: both, in order
    if (p.part != 2) {
      launch(std::true_type{}, 0, 1);
      int rc = check_launch("wino43v_tn (Vd)");
      if (rc) return rc;
    }
    if (ntm > 1 && p.part != 1) launch(std::false_type{}, 1, ntm - 1);
  } else {
    launch(std::false_type{}, 0, ntm);
  }
  return check_launch("wino43v_tn");
}

// Vd of a stage from its pooled output gradient G (g_rows rows = conv_rows / 2) and arg-max bits: conv_rows / 4 quads
extern "C" int tl_wino43_unpool_transform(const float* G, const uint32_t* bits, float* V, int64_t conv_rows, int64_t g_rows,
                                          int Tp, int Tvalid, int C, int ldg, int ld_bits, int ldv, void* stream) {
  using namespace tl;
  TL_REQUIRE(G && bits && V, "wino43_unpool_transform: null pointer");
  TL_REQUIRE(conv_rows > 0 && conv_rows % 4 == 0 && g_rows > 0 && Tp > 0 && Tp % 4 == 0 && Tvalid % 2 == 0 && Tvalid <= Tp,
             "wino43_unpool_transform: conv_rows %% 4, Tp %% 4, even Tvalid <= Tp needed");
  TL_REQUIRE(C > 0 && C % 4 == 0 && ldg >= C && ldv >= C && ldg % 4 == 0 && ldv % 4 == 0 && ld_bits * 32 >= C,
             "wino43_unpool_transform: C/ldg/ldv must be multiples of 4, bits row must cover C");
  const long long nq = conv_rows / 4;
  const long long n = nq * (C / 4);
  TL_REQUIRE((n + 255) / 256 < (1LL << 31), "wino43_unpool_transform: grid too large");
  hipLaunchKernelGGL(wino43_unpool_xform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, G, bits, V,
                     nq, (long long)g_rows, Tp, Tvalid, C, ldg, ld_bits, ldv);
  return check_launch("wino43_unpool_transform");
}

// HBM-bound kernels of the synthesis train step (gfx950): first conv stage (C_in = 1), layout
// packs / split-K reductions, bias-gradient column sums, LSTM cell, concat + dropout glue,
// L1 + MCD, fused NAdam, tone-dynamics gather.  All are streaming kernels: coalesced loads
// along the channel axis, one pass over each tensor, no atomics (deterministic sums).
#include "tonal_common.h"
#include <string.h>
#include <math.h>

namespace tl {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// counter-based uniform in [0,1): splitmix64 finaliser over (seed, index)
__device__ __forceinline__ float u01(uint64_t seed, uint64_t idx) {
  uint64_t zz = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
  zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
  zz ^= zz >> 31;
  return (float)(zz >> 40) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------
// conv1 forward: one workgroup per sequence, thread = output channel(s)
// ------------------------------------------------------------------------------------------
constexpr int MAXKT = 8;

template <int KT>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ P,
                                                        uint32_t* __restrict__ bits, uint32_t* __restrict__ sign,
                                                        long long S, int T, int C1, int Tp, int Tout,
                                                        float slope) {
  constexpr int kt = KT;                         // template parameter: the tap loops unroll without scalar branches
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const long long seq = blockIdx.x;
  for (int i = threadIdx.x; i < T; i += blockDim.x) xs[i] = x[seq * T + i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int o = threadIdx.x; o < C1; o += blockDim.x) {   // C1 % 64 == 0 -> whole waves stay active
    float wv[MAXKT];
#pragma unroll
    for (int j = 0; j < MAXKT; ++j) wv[j] = j < kt ? w[o * kt + j] : 0.f;
    const float bv = b[o];
    for (int p = 0; p < Tp; ++p) {
      float out = 0.f;
      bool sel = false;
      if (p < Tout) {
        float z0 = 0.f, z1 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXKT; ++j)
          if (j < kt) {
            z0 = fmaf(wv[j], xs[2 * p + j], z0);
            z1 = fmaf(wv[j], xs[2 * p + 1 + j], z1);
          }
        const float y0 = lrelu(z0 + bv, slope), y1 = lrelu(z1 + bv, slope);
        sel = y1 > y0;
        out = sel ? y1 : y0;
      }
      const long long row = seq * Tp + p;
      P[row * C1 + o] = out;
      const unsigned long long m = __ballot(sel);
      const unsigned long long ms = __ballot(out > 0.f);
      if (lane == 0) {
        bits[row * (C1 >> 5) + (o >> 5)] = (uint32_t)m;
        bits[row * (C1 >> 5) + (o >> 5) + 1] = (uint32_t)(m >> 32);
        if (sign != nullptr) {
          sign[row * (C1 >> 5) + (o >> 5)] = (uint32_t)ms;
          sign[row * (C1 >> 5) + (o >> 5) + 1] = (uint32_t)(ms >> 32);
        }
      }
    }
  }
}

// Same stage with 16-byte stores: a thread owns FOUR consecutive output channels (C1 / 4 threads per row, several
// rows per pass), so a pooled row leaves as 2 KB of contiguous dwordx4 stores instead of 4-byte ones, and the
// arg-max / sign words are assembled from the 8 lanes x 4 bits that make up 32 channels with three xor-shuffles.
// Used when C1 is a multiple of 128 (the 512- and 1024-wide stages of the models); HBM-write bound.
template <int KT>
__global__ __launch_bounds__(256) void conv1_fwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ P,
                                                           uint32_t* __restrict__ bits, uint32_t* __restrict__ sign,
                                                           long long S, int T, int C1, int Tp, int Tout,
                                                           float slope) {
  constexpr int kt = KT;                         // template parameter: the tap loops unroll without scalar branches
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const long long seq = blockIdx.x;
  for (int i = threadIdx.x; i < T; i += blockDim.x) xs[i] = x[seq * T + i];
  __syncthreads();
  const int groups = C1 >> 2;                    // threads per row (<= 256, a multiple of 32)
  const int rpp = 256 / groups;                  // rows per pass
  const int g = threadIdx.x % groups, rsub = threadIdx.x / groups;
  const int o = 4 * g;
  float wv[4][MAXKT], bv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bv[c] = b[o + c];
#pragma unroll
    for (int j = 0; j < MAXKT; ++j) wv[c][j] = j < kt ? w[(o + c) * kt + j] : 0.f;
  }
  const int sh = 4 * (threadIdx.x & 7);
  for (int p0 = 0; p0 < Tp; p0 += rpp) {
    const int p = p0 + rsub;                     // Tp % rpp may be non-zero: guard the stores, keep the shuffles uniform
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    uint32_t nib = 0, nsg = 0;
    if (p < Tout) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float z0 = 0.f, z1 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXKT; ++j)
          if (j < kt) {
            z0 = fmaf(wv[c][j], xs[2 * p + j], z0);
            z1 = fmaf(wv[c][j], xs[2 * p + 1 + j], z1);
          }
        const float y0 = lrelu(z0 + bv[c], slope), y1 = lrelu(z1 + bv[c], slope);
        const bool sel = y1 > y0;
        const float v = sel ? y1 : y0;
        out[c] = v;
        nib |= (sel ? 1u : 0u) << c;
        nsg |= (v > 0.f ? 1u : 0u) << c;
      }
    }
    uint32_t wb = nib << sh, ws = nsg << sh;
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      wb |= __shfl_xor(wb, m);
      ws |= __shfl_xor(ws, m);
    }
    if (p < Tp) {
      const long long row = seq * Tp + p;
      *reinterpret_cast<f32x4*>(P + row * C1 + o) = out;
      if ((threadIdx.x & 7) == 0) {
        bits[row * (C1 >> 5) + (o >> 5)] = wb;
        if (sign != nullptr) sign[row * (C1 >> 5) + (o >> 5)] = ws;
      }
    }
  }
}

// Same stage, writing the F(4,3) INPUT TRANSFORM of its pooled output for the next stage (tonal_wino43v.hip):
// V[(seq * Tp/4 + q)][6][C1] from the pooled rows 4q..4q+5 (rows past Tout, and past the sequence, are zero), so the
// next stage's forward / weight-gradient GEMMs read V by LDS-DMA and the 13 GB of P1 need not exist at all (P is
// optional: tests and the direct-form kernels want it).  Thread = 4 channels x one quad; the two halo rows of a quad
// are recomputed (3 MACs per element from the LDS-resident signal) rather than exchanged.  HBM-write bound.
#ifndef CONV1_NT
#define CONV1_NT 1      // same-box A/B under rocprofv3: 4.79 -> 4.65 ms per launch (20 GB written once, read by the next kernel)
#endif
template <int KT>
__global__ __launch_bounds__(256) void conv1_fwd_vq_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ P,
                                                           float* __restrict__ V, uint32_t* __restrict__ bits,
                                                           uint32_t* __restrict__ sign, long long S, int T, int C1,
                                                           int Tp, int Tout, float slope) {
  constexpr int kt = KT;                         // template parameter: the tap loops unroll without scalar branches
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const long long seq = blockIdx.x;
  for (int i = threadIdx.x; i < T; i += blockDim.x) xs[i] = x[seq * T + i];
  __syncthreads();
  const int groups = C1 >> 2;                    // threads per quad (<= 256, a multiple of 32)
  const int qpp = 256 / groups;                  // quads per pass
  const int g = threadIdx.x % groups, qsub = threadIdx.x / groups;
  const int o = 4 * g;
  float wv[4][MAXKT], bv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bv[c] = b[o + c];
#pragma unroll
    for (int j = 0; j < MAXKT; ++j) wv[c][j] = j < kt ? w[(o + c) * kt + j] : 0.f;
  }
  const int sh = 4 * (threadIdx.x & 7);
  const int Tq = Tp >> 2;
  // one pooled row (4 channels of this thread) + its arg-max / sign bit words assembled over the 8 lanes of a 32-channel group
  auto row = [&](int p, bool live, f32x4& out, uint32_t& wb, uint32_t& ws) {
    out = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t nib = 0, nsg = 0;
    if (live && p < Tout) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float z0 = 0.f, z1 = 0.f;
#pragma unroll
        for (int jj = 0; jj < MAXKT; ++jj)
          if (jj < kt) {
            z0 = fmaf(wv[c][jj], xs[2 * p + jj], z0);
            z1 = fmaf(wv[c][jj], xs[2 * p + 1 + jj], z1);
          }
        const float y0 = lrelu(z0 + bv[c], slope), y1 = lrelu(z1 + bv[c], slope);
        const bool sel = y1 > y0;
        const float v = sel ? y1 : y0;
        out[c] = v;
        nib |= (sel ? 1u : 0u) << c;
        nsg |= (v > 0.f ? 1u : 0u) << c;
      }
    }
    uint32_t a = nib << sh, e = nsg << sh;
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      a |= __shfl_xor(a, m);
      e |= __shfl_xor(e, m);
    }
    wb = a;
    ws = e;
  };
  // a thread walks CONSECUTIVE quads (its row group takes quads [qsub * per, (qsub + 1) * per)): rows 4, 5 of a quad are
  // rows 0, 1 of the next, so four new pooled rows per quad instead of six (the shuffles stay uniform: every thread of a
  // wave runs the same trip count)
  const int per = (Tq + qpp - 1) / qpp;
  const int qb = qsub * per;
  f32x4 d[6];
  uint32_t wb[6], ws[6];
  row(4 * qb, qb < Tq, d[0], wb[0], ws[0]);
  row(4 * qb + 1, qb < Tq, d[1], wb[1], ws[1]);
  for (int i = 0; i < per; ++i) {
    const int q = qb + i;
    const bool live = q < Tq;
#pragma unroll
    for (int j = 2; j < 6; ++j) row(4 * q + j, live, d[j], wb[j], ws[j]);
    if (live) {
      const long long row0 = seq * Tp + 4 * q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (P != nullptr) *reinterpret_cast<f32x4*>(P + (row0 + j) * C1 + o) = d[j];
        if ((threadIdx.x & 7) == 0) {
          bits[(row0 + j) * (C1 >> 5) + (o >> 5)] = wb[j];
          if (sign != nullptr) sign[(row0 + j) * (C1 >> 5) + (o >> 5)] = ws[j];
        }
      }
      const f32x4 s1 = d[4] - 4.f * d[2], s2 = d[3] - 4.f * d[1], s3 = d[4] - d[2], t = d[3] - d[1];
      float* dst = V + (seq * Tq + q) * 6LL * C1 + o;
      // V1 (20 GB at the north-star shape) is written once and read by a later kernel: CONV1_NT streams it past the L2
      auto stv = [](float* p_, const f32x4 v) {
#if CONV1_NT
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p_));
#else
        *reinterpret_cast<f32x4*>(p_) = v;
#endif
      };
      stv(dst, 4.f * d[0] + (d[4] - 5.f * d[2]));
      stv(dst + C1, s1 + s2);
      stv(dst + 2LL * C1, s1 - s2);
      stv(dst + 3LL * C1, s3 + 2.f * t);
      stv(dst + 4LL * C1, s3 - 2.f * t);
      stv(dst + 5LL * C1, (4.f * d[1] - 5.f * d[3]) + d[5]);
    }
    d[0] = d[4]; d[1] = d[5];
    wb[0] = wb[4]; wb[1] = wb[5];
    ws[0] = ws[4]; ws[1] = ws[5];
  }
}

// Inverted dropout in place on a flat buffer (nn.Dropout of the deep classifiers, reference
// models/deep_classifiers.py:81,258, active when the synthesis trainer runs them in train mode): keep with
// probability 1 - p, scale by 1 / (1 - p); the counter-hash stream of the synthesis model's dropout (u01 above),
// indexed by the element's position in the buffer.
__global__ __launch_bounds__(256) void dropout_scale_kernel(float* __restrict__ x, long long n, float p, float inv_keep,
                                                            uint64_t seed) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    x[i] = u01(seed, (uint64_t)i) >= p ? x[i] * inv_keep : 0.f;
}

// The per-step scalars of a HIP-graph-replayed train step (tl_nadam_multi_dev, tl_lite_cat_dev / tl_lite_uncat_dev) are
// written by a one-thread launch whose ARGUMENTS carry the values: stream-ordered in front of the replay with no host
// buffer that a later step could overwrite before an asynchronous copy has read it.
__global__ void set_step_scalars_kernel(float* __restrict__ sc, uint64_t* __restrict__ seed, float cg, float cm, float bc2,
                                        uint64_t seedval) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (sc != nullptr) {
      sc[0] = cg;
      sc[1] = cm;
      sc[2] = bc2;
    }
    if (seed != nullptr) *seed = seedval;
  }
}

// The same plus the step's input tensors copied into the graph's static buffers: everything a replay needs, one launch
// instead of one small copy per tensor and the scalar launch.  16-byte units where source, destination and size allow.
struct stage_args {
  const char* src[4];
  char* dst[4];
  long long nbytes[4];
  int n;
};
__global__ __launch_bounds__(256) void stage_step_kernel(stage_args a, float* __restrict__ sc, uint64_t* __restrict__ seed, float cg,
                                                         float cm, float bc2, uint64_t seedval) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (sc != nullptr) {
      sc[0] = cg;
      sc[1] = cm;
      sc[2] = bc2;
    }
    if (seed != nullptr) *seed = seedval;
  }
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
  for (int i = 0; i < a.n; ++i) {
    const char* s = a.src[i];
    char* d = a.dst[i];
    const long long nb = a.nbytes[i];
    if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)nb) & 15) == 0) {
      for (long long k = t; k < (nb >> 4); k += nt) reinterpret_cast<float4*>(d)[k] = reinterpret_cast<const float4*>(s)[k];
    } else if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)nb) & 3) == 0) {
      for (long long k = t; k < (nb >> 2); k += nt) reinterpret_cast<float*>(d)[k] = reinterpret_cast<const float*>(s)[k];
    } else {
      for (long long k = t; k < nb; k += nt) d[k] = s[k];
    }
  }
}

// conv1 weight/bias gradient partials: block handles a contiguous range of sequences;
// thread owns channels tid and tid + 256 (C1 <= 512), barriers are outside every guard.
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ G,
                                                          const uint32_t* __restrict__ bits, float* __restrict__ partial,
                                                          long long S, int T, int kt, int C1, int Tp, int Tout) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const long long per = (S + gridDim.x - 1) / gridDim.x;
  const long long s0 = blockIdx.x * per;
  const long long s1 = s0 + per < S ? s0 + per : S;
  float acc[2][MAXKT + 1];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j <= MAXKT; ++j) acc[h][j] = 0.f;
  for (long long seq = s0; seq < s1; ++seq) {
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) xs[i] = x[seq * T + i];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int o = threadIdx.x + h * 256;
      if (o < C1) {
        int p = 0;
        for (; p + 4 <= Tout; p += 4) {          // 4 independent loads in flight per thread
          float g[4];
          uint32_t wb[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const long long row = seq * Tp + p + u;
            g[u] = G[row * C1 + o];
            wb[u] = bits[row * (C1 >> 5) + (o >> 5)];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int a = (wb[u] >> (o & 31)) & 1;
#pragma unroll
            for (int j = 0; j < MAXKT; ++j)
              if (j < kt) acc[h][j] = fmaf(g[u], xs[2 * (p + u) + a + j], acc[h][j]);
            acc[h][MAXKT] += g[u];
          }
        }
        for (; p < Tout; ++p) {
          const long long row = seq * Tp + p;
          const float g = G[row * C1 + o];
          const uint32_t wbit = bits[row * (C1 >> 5) + (o >> 5)];
          const int a = (wbit >> (o & 31)) & 1;
#pragma unroll
          for (int j = 0; j < MAXKT; ++j)
            if (j < kt) acc[h][j] = fmaf(g, xs[2 * p + a + j], acc[h][j]);
          acc[h][MAXKT] += g;
        }
      }
    }
  }
  float* dst = partial + (long long)blockIdx.x * (kt + 1) * C1;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int o = threadIdx.x + h * 256;
    if (o < C1) {
#pragma unroll
      for (int j = 0; j < MAXKT; ++j)
        if (j < kt) dst[j * C1 + o] = acc[h][j];
      dst[kt * C1 + o] = acc[h][MAXKT];
    }
  }
}

// ------------------------------------------------------------------------------------------
// generic 4-D permute + slab reduction
// ------------------------------------------------------------------------------------------
struct perm_args {
  long long d[4], s[4], lim[4];
  long long zs;
  int nz;
};
__global__ __launch_bounds__(256) void permute_reduce_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             const float* __restrict__ bias_last, perm_args a,
                                                             long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const long long i3 = r % a.d[3]; r /= a.d[3];
    const long long i2 = r % a.d[2]; r /= a.d[2];
    const long long i1 = r % a.d[1]; r /= a.d[1];
    const long long i0 = r;
    float acc = 0.f;
    if (i0 < a.lim[0] && i1 < a.lim[1] && i2 < a.lim[2] && i3 < a.lim[3]) {
      const long long off = i0 * a.s[0] + i1 * a.s[1] + i2 * a.s[2] + i3 * a.s[3];
      for (int z = 0; z < a.nz; ++z) acc += src[off + z * a.zs];
      if (bias_last) acc += bias_last[i3];
    }
    dst[i] = acc;
  }
}

// two slab sums in one launch: dstA[i] = sum_z srcA[z nA + i], dstB[j] = sum_z srcB[z nB + j] (z ascending) - the per-window
// partials of a convolution's weight and bias gradient (tl_lite_conv_bwd) reduced together
__global__ __launch_bounds__(256) void sum_slabs2_kernel(const float* __restrict__ srcA, float* __restrict__ dstA, long long nA,
                                                         const float* __restrict__ srcB, float* __restrict__ dstB, long long nB,
                                                         int nz) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nA + nB; i += (long long)gridDim.x * blockDim.x) {
    const bool second = i >= nA;
    const float* s = second ? srcB + (i - nA) : srcA + i;
    const long long zs = second ? nB : nA;
    float acc = 0.f;
#pragma unroll 8
    for (int z = 0; z < nz; ++z) acc += s[z * zs];
    if (second) dstB[i - nA] = acc; else dstA[i] = acc;
  }
}

// same, for few outputs and many slabs: one wave per output element, lanes stride over the slabs,
// fixed-shape shuffle tree (deterministic)
__global__ __launch_bounds__(256) void permute_reduce_zpar_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                  const float* __restrict__ bias_last, perm_args a,
                                                                  long long total) {
  const int lane = threadIdx.x & 63;
  const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long i = wave; i < total; i += nwaves) {
    long long r = i;
    const long long i3 = r % a.d[3]; r /= a.d[3];
    const long long i2 = r % a.d[2]; r /= a.d[2];
    const long long i1 = r % a.d[1]; r /= a.d[1];
    const long long i0 = r;
    float acc = 0.f;
    const bool ok = i0 < a.lim[0] && i1 < a.lim[1] && i2 < a.lim[2] && i3 < a.lim[3];
    if (ok) {
      const long long off = i0 * a.s[0] + i1 * a.s[1] + i2 * a.s[2] + i3 * a.s[3];
      for (int z = lane; z < a.nz; z += 64) acc += src[off + z * a.zs];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane == 0) dst[i] = ok ? acc + (bias_last ? bias_last[i3] : 0.f) : 0.f;
  }
}

// masked column sums (bias gradients): partial[blk][ncols].  float4 per thread along the
// columns, 256/(ncols/4) rows per pass, 4 independent row loads in flight per thread, LDS
// reduction over the row lanes (fixed order -> deterministic).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ G, float* __restrict__ partial,
                                                     long long rows, int ncols, int ld, int Tp, int Tvalid) {
  __shared__ f32x4 red[256];
  const int tpr = ncols >> 2;                 // threads per row (ncols % 4 == 0, tpr <= 256)
  const int rpb = 256 / tpr;                  // rows per pass
  const int c4 = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const bool active = rl < rpb;
  const long long per = (rows + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * per;
  const long long r1 = r0 + per < rows ? r0 + per : rows;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    long long r = r0 + rl;
    int t = (int)(r % Tp);
    const int dt = rpb % Tp;
    for (; r + 3LL * rpb < r1; r += 4LL * rpb) {
      f32x4 v[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ok[u] = t < Tvalid;
        v[u] = *reinterpret_cast<const f32x4*>(G + (r + (long long)u * rpb) * ld + c4 * 4);
        t += dt;
        if (t >= Tp) t -= Tp;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) acc += v[u];
    }
    for (; r < r1; r += rpb) {
      if (t < Tvalid) acc += *reinterpret_cast<const f32x4*>(G + r * ld + c4 * 4);
      t += dt;
      if (t >= Tp) t -= Tp;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < tpr) {
    f32x4 sum = red[threadIdx.x];
    for (int q = 1; q < rpb; ++q) sum += red[threadIdx.x + q * tpr];
    *reinterpret_cast<f32x4*>(partial + (long long)blockIdx.x * ncols + threadIdx.x * 4) = sum;
  }
}

// ------------------------------------------------------------------------------------------
// LSTM cell (torch gate order i, f, g, o)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* __restrict__ hh, const float* __restrict__ x_t,
                                                            const float* __restrict__ w_ih, const float* __restrict__ b_ih,
                                                            const float* __restrict__ b_hh, const float* __restrict__ c_prev,
                                                            float* __restrict__ act, float* __restrict__ c,
                                                            float* __restrict__ h, int U, int H, int in_dim, int ld_hh) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)U * H) return;
  const int u = (int)(i / H), k = (int)(i % H);
  float pre[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const long long row = (long long)q * H + k;
    float ih = 0.f;
    for (int d = 0; d < in_dim; ++d) ih = fmaf(x_t[u * in_dim + d], w_ih[row * in_dim + d], ih);
    ih += b_ih[row];
    float hv = b_hh[row];
    if (hh) hv += hh[(long long)u * ld_hh + row];
    pre[q] = ih + hv;
  }
  const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
  const float cp = c_prev ? c_prev[i] : 0.f;
  const float cn = fg * cp + ig * gg;
  const long long ab = (long long)u * 4 * H + k;
  act[ab] = ig;
  act[ab + H] = fg;
  act[ab + 2LL * H] = gg;
  act[ab + 3LL * H] = og;
  c[i] = cn;
  h[i] = og * tanhf(cn);
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dh_rec,
                                                            const float* __restrict__ dc_next, const float* __restrict__ act,
                                                            const float* __restrict__ c, const float* __restrict__ c_prev,
                                                            float* __restrict__ dgates, float* __restrict__ dgates_t,
                                                            float* __restrict__ dc_prev, int U, int H, int ldt) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)U * H) return;
  const int u = (int)(i / H), k = (int)(i % H);
  float dht = 0.f;
  if (dh) dht += dh[i];
  if (dh_rec) dht += dh_rec[i];
  const long long ab = (long long)u * 4 * H + k;
  const float ig = act[ab], fg = act[ab + H], gg = act[ab + 2LL * H], og = act[ab + 3LL * H];
  const float tc = tanhf(c[i]);
  const float dog = dht * tc;
  float dc = dht * og * (1.f - tc * tc);
  if (dc_next) dc += dc_next[i];
  const float cp = c_prev ? c_prev[i] : 0.f;
  const float d_i = dc * gg * ig * (1.f - ig);
  const float d_f = dc * cp * fg * (1.f - fg);
  const float d_g = dc * ig * (1.f - gg * gg);
  const float d_o = dog * og * (1.f - og);
  dgates[ab] = d_i;
  dgates[ab + H] = d_f;
  dgates[ab + 2LL * H] = d_g;
  dgates[ab + 3LL * H] = d_o;
  if (dgates_t) {
    dgates_t[((long long)k) * ldt + u] = d_i;
    dgates_t[((long long)H + k) * ldt + u] = d_f;
    dgates_t[(2LL * H + k) * ldt + u] = d_g;
    dgates_t[(3LL * H + k) * ldt + u] = d_o;
  }
  dc_prev[i] = dc * fg;
}

__global__ __launch_bounds__(256) void lstm_ih_grad_kernel(const float* __restrict__ dgates, const float* __restrict__ x,
                                                           float* __restrict__ dw_ih, float* __restrict__ db,
                                                           float* __restrict__ db2, int L, int U, int H, int in_dim) {
  const long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (r >= 4LL * H) return;
  float accw[MAXKT];
#pragma unroll
  for (int d = 0; d < MAXKT; ++d) accw[d] = 0.f;
  float accb = 0.f;
  for (int tu = 0; tu < L * U; ++tu) {
    const float g = dgates[(long long)tu * 4 * H + r];
    accb += g;
#pragma unroll
    for (int d = 0; d < MAXKT; ++d)
      if (d < in_dim) accw[d] = fmaf(g, x[tu * in_dim + d], accw[d]);
  }
#pragma unroll
  for (int d = 0; d < MAXKT; ++d)
    if (d < in_dim) dw_ih[r * in_dim + d] = accw[d];
  db[r] = accb;
  if (db2) db2[r] = accb;
}

// few gate rows (SynthesisLite: 4H = 256): one wave per row, lanes stride over the (t,u) pairs
__global__ __launch_bounds__(64) void lstm_ih_grad_wave_kernel(const float* __restrict__ dgates, const float* __restrict__ x,
                                                               float* __restrict__ dw_ih, float* __restrict__ db,
                                                               float* __restrict__ db2, int LU, int H, int in_dim) {
  const long long r = blockIdx.x;
  const int lane = threadIdx.x;
  float accw[MAXKT];
#pragma unroll
  for (int d = 0; d < MAXKT; ++d) accw[d] = 0.f;
  float accb = 0.f;
  for (int tu = lane; tu < LU; tu += 64) {
    const float g = dgates[(long long)tu * 4 * H + r];
    accb += g;
#pragma unroll
    for (int d = 0; d < MAXKT; ++d)
      if (d < in_dim) accw[d] = fmaf(g, x[tu * in_dim + d], accw[d]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    accb += __shfl_down(accb, o);
#pragma unroll
    for (int d = 0; d < MAXKT; ++d) accw[d] += __shfl_down(accw[d], o);
  }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < MAXKT; ++d)
      if (d < in_dim) dw_ih[r * in_dim + d] = accw[d];
    db[r] = accb;
    if (db2) db2[r] = accb;
  }
}

// ------------------------------------------------------------------------------------------
// dgates . W_hh of the label LSTM on its U <= 8 DISTINCT label rows (round 6): W_hh is 5.4 GB at the north-star shape and the
// product is a pure stream of it - what matters is bytes in flight, not the matrix pipe (the MFMA form it replaces, a skinny TN
// GEMM through LDS with split-K slabs, ran at 5.2 TB/s; this one at 5.8, a library GEMM on the same operands at 6.0).
//   lstm_gw_kernel    slab[rb][u][k] = sum_{n in block rb} g[u][n] W[n][k]   (a thread owns 4 consecutive k, a workgroup 1024
//                     columns x `rpb` rows with g of the row block in LDS as [row][8]; eight 16-byte loads in flight per lane;
//                     the caller sums the slabs).  Same arithmetic in a different summation order (fp32 FMA chains along n).
// (The forward product h W_hh^T stays on tl_gemm_nt_window: two row-dot streaming forms of it - h in registers, h staged in LDS -
// ran at 4.3 and 3.6 TB/s against the MFMA form's 5.3; measured and removed, profiles/r06_kernel_notes.md 7.)
// ------------------------------------------------------------------------------------------
constexpr int HW_MAXU = 8;
constexpr int GW_COLS = 1024, GW_MAXROWS = 1024;
__global__ __launch_bounds__(256, 2) void lstm_gw_kernel(const float* __restrict__ g, const float* __restrict__ W,
                                                         float* __restrict__ slab, int U, long long N, int K, long long ldg,
                                                         long long ldw, int rpb) {
  __shared__ __attribute__((aligned(16))) float sg[GW_MAXROWS * HW_MAXU];       // [row of the block][u] (rows u >= U: zero)
  const int k = blockIdx.x * GW_COLS + 4 * threadIdx.x;
  const long long r0 = (long long)blockIdx.y * rpb;
  const int nr = (int)(N - r0 < rpb ? N - r0 : rpb);
  for (int i = threadIdx.x; i < HW_MAXU * rpb; i += 256) {
    const int u = i / rpb, r = i - u * rpb;                             // (consecutive threads read consecutive n of one u)
    sg[r * HW_MAXU + u] = (u < U && r < nr) ? g[(long long)u * ldg + r0 + r] : 0.f;
  }
  __syncthreads();
  f32x4 acc[HW_MAXU];
#pragma unroll
  for (int u = 0; u < HW_MAXU; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (k < K) {                                                          // K % 4 == 0: the float4 is in or out
    const float* wp = W + r0 * ldw + k;
    auto row = [&](const f32x4 wv, int r) {
      const f32x4 ga = *reinterpret_cast<const f32x4*>(sg + r * HW_MAXU), gb = *reinterpret_cast<const f32x4*>(sg + r * HW_MAXU + 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[u] += ga[u] * wv;                                           // (u >= U: a zero factor)
        acc[u + 4] += gb[u] * wv;
      }
    };
    int r = 0;
#pragma unroll 1
    for (; r + 8 <= nr; r += 8) {
      f32x4 wv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) wv[q] = *reinterpret_cast<const f32x4*>(wp + (long long)(r + q) * ldw);
#pragma unroll
      for (int q = 0; q < 8; ++q) row(wv[q], r + q);
    }
    for (; r < nr; ++r) row(*reinterpret_cast<const f32x4*>(wp + (long long)r * ldw), r);
#pragma unroll
    for (int u = 0; u < HW_MAXU; ++u)
      if (u < U) *reinterpret_cast<f32x4*>(slab + ((long long)blockIdx.y * U + u) * K + k) = acc[u];
  }
}

// One inference step of the LSTM in ONE launch (instead of split-K GEMM + cell): a workgroup owns 32
// batch rows and 8 hidden units = 32 gate columns (the recurrent weight is packed unit-major, row 4 u + g,
// so a tile holds whole cells); gates = h_prev . Wp^T + xp[t], then the cell update.  8 interleaved K slices
// per row block, one wave each; operand fragments come straight from L2 (each value feeds one wave, nothing
// to share through LDS), the eight partial tiles are summed through LDS by the cell epilogue.
// h is double buffered by the caller: every workgroup reads all of h_in while others write h_out.
constexpr int LF_NKS = 8;
constexpr int NRB = 1;      // row blocks per workgroup (64-row tiles measured slower: 6.3 vs 4.2 ms on 400 steps)
__global__ __launch_bounds__(512, 2) void lstm_step_fused_kernel(const float* __restrict__ xp, long long xp_row_stride,
                                                                    const float* __restrict__ wp,
                                                                    const float* __restrict__ h_in,
                                                                    float* __restrict__ h_out, float* __restrict__ c,
                                                                    int B, int H, int first) {
  __shared__ __attribute__((aligned(16))) float red[LF_NKS * NRB * 32 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rb = wave % NRB, ks = wave / NRB;
  const int lr = lane & 31, kh = lane >> 5;
  const int row0 = blockIdx.y * (NRB * 32);
  const int u0 = blockIdx.x * 8;
  // the cell inputs of this thread's (row, unit): fetched up front, used after the product
  const int rloc = tid >> 3, ucell = tid & 7;
  const bool cell = tid < NRB * 256 && row0 + rloc < B;
  float xg[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
  if (cell) {
    const float* x = xp + (long long)(row0 + rloc) * xp_row_stride + (u0 + ucell);
#pragma unroll
    for (int g = 0; g < 4; ++g) xg[g] = x[(long long)g * H];
    if (!first) cprev = c[(long long)(row0 + rloc) * H + u0 + ucell];
  }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  if (!first) {
    int row = row0 + rb * 32 + lr;
    row = row < B ? row : B - 1;                         // clamped rows are never written
    const float* ap = h_in + (long long)row * H + 4 * kh;
    const float* bp = wp + (long long)(4 * u0 + lr) * H + 4 * kh;
    const int nchunk = H >> 3;                           // host-checked: H % 8 == 0
    // two chunks per trip: deeper prefetch measured SLOWER (400 steps, hidden 800, batch 64: 2 -> 4.2 ms, 4 -> 4.5,
    // 8 -> 5.1, 16 = the whole slice in flight -> 5.4): the step is bound by the 52 fp32 MFMAs per wave (two waves
    // per SIMD: 3 us) running behind the first loads, and a long burst only delays those
#pragma unroll 2
    for (int kc = ks; kc < nchunk; kc += LF_NKS) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ap + kc * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(bp + kc * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
    }
  }
  float* mine = red + (ks * NRB + rb) * 1024;            // [32 rows][32 gate columns]
#pragma unroll
  for (int e = 0; e < 16; ++e) mine[((e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + lr] = acc[e];
  __syncthreads();
  if (cell) {                                            // one (row, unit) cell per thread
    f32x4 pre = {0.f, 0.f, 0.f, 0.f};
    if (!first) {
#pragma unroll
      for (int z = 0; z < LF_NKS; ++z)
        pre += *reinterpret_cast<const f32x4*>(red + (z * NRB + (rloc >> 5)) * 1024 + (rloc & 31) * 32 + 4 * ucell);
    }
    const float ig = sigmoidf_(pre[0] + xg[0]), fg = sigmoidf_(pre[1] + xg[1]), gg = tanhf(pre[2] + xg[2]),
                og = sigmoidf_(pre[3] + xg[3]);
    const long long i = (long long)(row0 + rloc) * H + u0 + ucell;
    const float cn = fg * cprev + ig * gg;               // cprev = 0 on the first step
    c[i] = cn;
    h_out[i] = og * tanhf(cn);
  }
}

// ------------------------------------------------------------------------------------------
// concat + dropout glue
// ------------------------------------------------------------------------------------------
// (IDX = unsigned where the element count allows it: the index arithmetic - three divisions per element - is what this
// kernel spends its time on, and the 64-bit forms cost several times the 32-bit ones)
template <typename IDX>
__global__ __launch_bounds__(256) void concat_pack_kernel(const float* __restrict__ O5, const float* __restrict__ h,
                                                          const int32_t* __restrict__ uid, float* __restrict__ Xc, int B,
                                                          int C, int Tp, int lat, int Cc, int Lc, int ld5, int ldh, int ldx,
                                                          float p_drop, uint64_t seed, long long drop_row0) {
  const IDX total = (IDX)((long long)B * C * Tp * ldx);
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  for (IDX i = (IDX)blockIdx.x * (IDX)blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * (IDX)blockDim.x) {
    const IDX row = i / (IDX)ldx;
    const int col = (int)(i - row * (IDX)ldx);
    const IDX seq = row / (IDX)Tp;
    const int t = (int)(row - seq * (IDX)Tp);
    const int b = (int)(seq / (IDX)C);
    const int cch = (int)(seq - (IDX)b * (IDX)C);
    float v = 0.f;
    if (t < lat) {
      if (col < Cc) {
        v = O5[(long long)row * ld5 + col];
        if (p_drop > 0.f) v = u01(seed, (uint64_t)((drop_row0 + (long long)row) * Cc + col)) >= p_drop ? v * keep_scale : 0.f;
      } else if (col < Cc + Lc) {
        v = h[(long long)uid[b] * ldh + ((long long)(col - Cc) * lat + t) * C + cch];
      }
    }
    Xc[i] = v;
  }
}

template <typename IDX>
__global__ __launch_bounds__(256) void concat_g5_kernel(const float* __restrict__ dXc, const float* __restrict__ O5,
                                                        float* __restrict__ G5, long long rows, int Cc, int ld5, int ldx,
                                                        float slope, float p_drop, uint64_t seed, long long drop_row0) {
  const IDX total = (IDX)(rows * Cc);
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  for (IDX i = (IDX)blockIdx.x * (IDX)blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * (IDX)blockDim.x) {
    const IDX rowi = i / (IDX)Cc;
    const int col = (int)(i - rowi * (IDX)Cc);
    const long long row = (long long)rowi;
    float g = dXc[row * ldx + col];
    if (p_drop > 0.f) g = u01(seed, (uint64_t)((drop_row0 + row) * Cc + col)) >= p_drop ? g * keep_scale : 0.f;
    const float a = O5[row * ld5 + col];
    G5[row * ld5 + col] = a > 0.f ? g : g * slope;
  }
}

__global__ __launch_bounds__(256) void concat_dh_kernel(const float* __restrict__ dXc, const int32_t* __restrict__ members,
                                                        const int32_t* __restrict__ offsets, float* __restrict__ dh, int U,
                                                        int C, int Tp, int lat, int Cc, int Lc, int ldh, int ldx, int Bn) {
  const long long per_u = (long long)Lc * lat * C;
  const long long total = (long long)U * per_u;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int u = (int)(i / per_u);
    long long r = i % per_u;
    const int cch = (int)(r % C); r /= C;
    const int t = (int)(r % lat);
    const int lc = (int)(r / lat);
    float acc = 0.f;
    if (members != nullptr) {
      for (int q = offsets[u]; q < offsets[u + 1]; ++q) {
        const int b = members[q];
        acc += dXc[(((long long)b * C + cch) * Tp + t) * ldx + Cc + lc];
      }
    } else {
      // no member lists: `offsets` holds the batch's label ids (B entries) - scan them in batch order, which is the
      // order a stable sort of the ids would list the members in (same sum, no argsort / cumsum in front of this kernel)
      for (int b = 0; b < Bn; ++b)
        if (offsets[b] == u) acc += dXc[(((long long)b * C + cch) * Tp + t) * ldx + Cc + lc];
    }
    dh[(long long)u * ldh + ((long long)lc * lat + t) * C + cch] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// L1 loss gradient + L1 / MCD statistics: one workgroup, thread per row, fixed-order reduction
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void l1_mcd_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                     float* __restrict__ dout, float* __restrict__ stats, int B, int D,
                                                     int ldd, int trunc_targets, float grad_scale) {
  __shared__ float s1[1024], s2[1024];    // (16 waves: a batch of 64 rows is four dependent rounds of loads, not sixteen)
  float l1 = 0.f, mcd = 0.f;
  const float gs = grad_scale / ((float)B * (float)D);
  // a wave per row, lanes across the D outputs (coalesced), a shuffle tree per row: the row sums are formed in a
  // fixed order, lane 0 of each wave carries them (a thread-per-row loop took 22-54 us for 5-20 k elements)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll 8          // rows are independent: let the loads of several rows be in flight together
  for (int b = wave; b < B; b += nw) {
    float a1 = 0.f, a2 = 0.f;
    for (int d = lane; d < D; d += 64) {
      float t = tgt[(long long)b * D + d];
      if (trunc_targets) t = truncf(t);
      const float df = out[(long long)b * D + d] - t;
      a1 += fabsf(df);
      a2 = fmaf(df, df, a2);
      if (dout) dout[(long long)b * ldd + d] = df > 0.f ? gs : (df < 0.f ? -gs : 0.f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a1 += __shfl_down(a1, off);
      a2 += __shfl_down(a2, off);
    }
    if (lane == 0) {
      l1 += a1;
      mcd += 4.342944819032518f * sqrtf(2.f * a2);   // 10 / ln(10)
    }
  }
  s1[threadIdx.x] = l1;
  s2[threadIdx.x] = mcd;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      s1[threadIdx.x] += s1[threadIdx.x + off];
      s2[threadIdx.x] += s2[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    stats[0] += s1[0] / ((float)B * (float)D);
    stats[1] += s2[0] / (float)B;
    stats[2] = s1[0] / ((float)B * (float)D);    // last-step values
    stats[3] = s2[0] / (float)B;
  }
}

// ------------------------------------------------------------------------------------------
// fused NAdam: 4 reads + 3 writes of 4 B per element, float4 wide
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void nadam_one(float& p, float g, float& m, float& v, float cg, float cm, float b1,
                                          float b2, float bc2, float eps, float wd, float gscale) {
  // every multiply-add is an explicit fma: the three kernels that inline this (per tensor, tensor list,
  // low rank) then round identically whatever the compiler would have contracted in their loops
  g = fmaf(wd, p, g * gscale);
  m = fmaf(g - m, 1.f - b1, m);
  v = fmaf(v, b2, (1.f - b2) * g * g);
  const float denom = sqrtf(v / bc2) + eps;
  p = fmaf(-cg, g / denom, p);
  p = fmaf(-cm, m / denom, p);
}
__global__ __launch_bounds__(256) void nadam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n, float cg,
                                                    float cm, float b1, float b2, float bc2, float eps, float wd,
                                                    float gscale) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float pq = pv[q], mq = mv[q], vq = vv[q];
      nadam_one(pq, gv[q], mq, vq, cg, cm, b1, b2, bc2, eps, wd, gscale);
      pv[q] = pq;
      mv[q] = mq;
      vv[q] = vq;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  for (long long i = (n4 << 2) + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride)
    nadam_one(p[i], g[i], m[i], v[i], cg, cm, b1, b2, bc2, eps, wd, gscale);
}

// The same update for a whole list of tensors in one launch (a model's many small tensors cost a
// launch each otherwise: 16 of the 66 launches of a SynthesisLite step).  A block owns one
// NM_CHUNK-element chunk of one tensor; entries[e].block0 = first block of tensor e (ascending).
constexpr int NM_CHUNK = 256 * 16;
__global__ __launch_bounds__(256) void nadam_multi_kernel(const tl_nadam_entry* __restrict__ entries, int count, float cg,
                                                          float cm, float b1, float b2, float bc2, float eps, float wd,
                                                          float gscale, const float* __restrict__ sc) {
  if (sc != nullptr) {      // step scalars in device memory (a HIP graph replays the launch with new values): cg, cm, bc2
    cg = sc[0];
    cm = sc[1];
    bc2 = sc[2];
  }
  const long long blk = blockIdx.x;
  int lo = 0, hi = count - 1;                         // last entry with block0 <= blk
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (entries[mid].block0 <= blk) lo = mid; else hi = mid - 1;
  }
  const tl_nadam_entry e = entries[lo];
  const long long base = (blk - e.block0) * NM_CHUNK;
  const long long end = (base + NM_CHUNK < e.n) ? base + NM_CHUNK : e.n;
  float* __restrict__ p = e.p;
  const float* __restrict__ g = e.g;
  float* __restrict__ m = e.m;
  float* __restrict__ v = e.v;
  const long long end4 = base + ((end - base) & ~3LL);
  for (long long i = base + 4 * threadIdx.x; i < end4; i += 4 * 256) {
    f32x4 pv = *reinterpret_cast<f32x4*>(p + i);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
    f32x4 mv = *reinterpret_cast<f32x4*>(m + i);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float pq = pv[q], mq = mv[q], vq = vv[q];
      nadam_one(pq, gv[q], mq, vq, cg, cm, b1, b2, bc2, eps, wd, gscale);
      pv[q] = pq;
      mv[q] = mq;
      vv[q] = vq;
    }
    *reinterpret_cast<f32x4*>(p + i) = pv;
    *reinterpret_cast<f32x4*>(m + i) = mv;
    *reinterpret_cast<f32x4*>(v + i) = vv;
  }
  for (long long i = end4 + threadIdx.x; i < end; i += 256) nadam_one(p[i], g[i], m[i], v[i], cg, cm, b1, b2, bc2, eps, wd, gscale);
}

// NAdam on a parameter whose gradient is low rank: g = fa^T . fb (fa: kr x rows, fb: kr x cols),
// formed in registers and never written to HBM.  Used for label_lstm.weight_hh_l0 (98.7 % of the
// parameters): kr <= (L-1) * U = 32 distinct (step, label) rows, so the 5.4 GB gradient tensor and
// its write + read disappear from the step.  Block tile 32 rows x 256 columns; a thread owns 8 rows
// x 4 columns; the factor tiles sit in LDS ([k][32] and [k][256]).
constexpr int LR_TR = 32, LR_TC = 256, LR_MAXK = 64, LR_MAXU = 8;
// DH (round 6): the same pass also produces dh = fa[0:U] . p_OLD (U x cols) - the last step of the label LSTM's BPTT,
// dh_1 = dgates_2 . W_hh, which needs only the weight as it was before this update and whose result the W_hh gradient does
// not depend on (h_0 = 0: the gradient has no term for the first step; rows 0..U-1 of fa ARE dgates_2).  A workgroup walks
// `row_tiles` consecutive 32-row tiles with the column factor tile resident, accumulates its U x 4 partial sums per thread in
// registers over the rows, reduces its four row groups through LDS in a fixed order and writes one partial slab
// dh_slab[blockIdx.x][U][cols]; the caller sums the slabs (deterministic: no atomics).  One 5.4 GB stream of W_hh less per step.
template <bool DH>
__global__ __launch_bounds__(256) void nadam_lowrank_kernel(float* __restrict__ p, float* __restrict__ m,
                                                            float* __restrict__ v, const float* __restrict__ fa,
                                                            const float* __restrict__ fb, int kr, int rows, int cols,
                                                            int ldfa, int ldfb, float cg, float cm, float b1, float b2,
                                                            float bc2, float eps, float wd, float gscale,
                                                            float* __restrict__ dh_slab, int U, int row_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lr_lds[];
  float* sa = lr_lds;                        // [kr][LR_TR]
  float* sb = lr_lds + kr * LR_TR;           // [kr][LR_TC]
  const int c0 = blockIdx.y * LR_TC;
  for (int i = threadIdx.x; i < kr * (LR_TC / 4); i += 256) {
    const int k = i / (LR_TC / 4), c4 = (i % (LR_TC / 4)) * 4;
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    if (c0 + c4 < cols) val = *reinterpret_cast<const f32x4*>(fb + (long long)k * ldfb + c0 + c4);
    *reinterpret_cast<f32x4*>(sb + k * LR_TC + c4) = val;
  }
  const int cg4 = (threadIdx.x & 63) * 4, rg = (threadIdx.x >> 6) * 8;
  const int col = c0 + cg4;
  const bool colok = col < cols;
  f32x4 dh[DH ? LR_MAXU : 1];
  if constexpr (DH) {
#pragma unroll
    for (int u = 0; u < LR_MAXU; ++u) dh[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nt = DH ? row_tiles : 1;
#pragma unroll 1
  for (int it = 0; it < nt; ++it) {
    const int r0 = (blockIdx.x * nt + it) * LR_TR;
    if (r0 >= rows) break;                   // (uniform over the workgroup)
    if (it > 0) __syncthreads();             // the row factor tile of the tile before is still being read
    for (int i = threadIdx.x; i < kr * LR_TR; i += 256) {
      const int k = i / LR_TR, r = i % LR_TR;
      sa[i] = (r0 + r) < rows ? fa[(long long)k * ldfa + r0 + r] : 0.f;
    }
    // DH: the weight rows of this thread are fetched ahead of the gradient arithmetic (they are in flight while it runs) and
    // meet the first U factor rows - the dgates of the product - inside that loop, where those are in registers anyway
    f32x4 pold[DH ? 8 : 1];
    if constexpr (DH) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = r0 + rg + r;
        pold[r] = (colok && row < rows) ? reinterpret_cast<const f32x4*>(p)[((long long)row * cols + col) >> 2]
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();
    f32x4 g[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) g[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    int k = 0;
    if constexpr (DH) {
#pragma unroll
      for (int u = 0; u < LR_MAXU; ++u) {
        if (u < U) {                         // (U <= kr, uniform)
          const f32x4 bv = *reinterpret_cast<const f32x4*>(sb + u * LR_TC + cg4);
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(sa + u * LR_TR + rg);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(sa + u * LR_TR + rg + 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            g[r] += a0[r] * bv;
            g[r + 4] += a1[r] * bv;
            dh[u] += a0[r] * pold[r];        // the weight BEFORE the update
            dh[u] += a1[r] * pold[r + 4];
          }
        }
      }
      k = U;
    }
    for (; k < kr; ++k) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(sb + k * LR_TC + cg4);
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(sa + k * LR_TR + rg);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(sa + k * LR_TR + rg + 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        g[r] += a0[r] * bv;
        g[r + 4] += a1[r] * bv;
      }
    }
    if (colok) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = r0 + rg + r;
        if (row >= rows) break;
        const long long at = ((long long)row * cols + col) >> 2;
        f32x4 pv, mv = reinterpret_cast<f32x4*>(m)[at], vv = reinterpret_cast<f32x4*>(v)[at];
        if constexpr (DH) pv = pold[r];
        else pv = reinterpret_cast<f32x4*>(p)[at];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float pq = pv[q], mq = mv[q], vq = vv[q];
          nadam_one(pq, g[r][q], mq, vq, cg, cm, b1, b2, bc2, eps, wd, gscale);
          pv[q] = pq;
          mv[q] = mq;
          vv[q] = vq;
        }
        reinterpret_cast<f32x4*>(p)[at] = pv;
        reinterpret_cast<f32x4*>(m)[at] = mv;
        reinterpret_cast<f32x4*>(v)[at] = vv;
      }
    }
  }
  if constexpr (DH) {
    // the four row groups of the workgroup hold partial sums for the same 256 columns: summed in the order 0, 1, 2, 3
    // (the scratch overlays the factor tiles, which nobody reads any more behind the barrier)
    __syncthreads();
    float* red = lr_lds;                                 // [3][LR_MAXU][LR_TC]
    const int w = threadIdx.x >> 6;
    if (w > 0) {
#pragma unroll
      for (int u = 0; u < LR_MAXU; ++u) *reinterpret_cast<f32x4*>(red + ((w - 1) * LR_MAXU + u) * LR_TC + cg4) = dh[u];
    }
    __syncthreads();
    if (w == 0 && colok) {
#pragma unroll
      for (int u = 0; u < LR_MAXU; ++u) {
        if (u >= U) break;
        f32x4 s = dh[u];
#pragma unroll
        for (int ww = 0; ww < 3; ++ww) s += *reinterpret_cast<const f32x4*>(red + (ww * LR_MAXU + u) * LR_TC + cg4);
        *reinterpret_cast<f32x4*>(dh_slab + ((long long)blockIdx.x * U + u) * cols + col) = s;
      }
    }
  }
}

__global__ void tone_dynamics_kernel(const long long* __restrict__ tone, const long long* __restrict__ syl,
                                     const float* __restrict__ table, float* __restrict__ labels, int32_t* err, int B,
                                     int n_tones, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  const int b = i / L, l = i % L;
  const long long t = tone[b];
  labels[((long long)b * 2) * L + l] = (float)syl[b];
  if (t < 0 || t >= n_tones) {
    *err = 1;
    labels[((long long)b * 2 + 1) * L + l] = 0.f;
  } else {
    labels[((long long)b * 2 + 1) * L + l] = table[t * L + l];
  }
}

// out[i] = LeakyReLU(sum_z slab[z][i] + bias[i % ncols]): the split-K reduction of a Linear layer with its bias and activation
__global__ __launch_bounds__(256) void splitk_bias_lrelu_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                                float* __restrict__ out, int nz, long long n, int ncols,
                                                                float slope) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int z = 0; z < nz; ++z) acc += slab[(long long)z * n + i];
    if (bias) acc += bias[i % ncols];
    out[i] = acc > 0.f ? acc : acc * slope;
  }
}

// out[r][n] = sum_k x[r][k] w[n][k] + bias[n] for a handful of output columns (a logistic-regression head on the flattened
// window: models/simple_classifiers.py:34-60): one workgroup per row, wave w takes columns w, w + 4, ..; the row and the
// weight rows stream through 16-byte loads (the weights stay in L2 across rows), partial sums meet in a wave reduction.
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int K, int N,
                                                          long long ldx, int act) {
  const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xr = x + (long long)r * ldx;
  const int K4 = K >> 2;
  for (int n = wave; n < N; n += 4) {
    const float* wr = w + (long long)n * K;
    float acc = 0.f;
    for (int i = lane; i < K4; i += 64) {
      const float4 a = reinterpret_cast<const float4*>(xr)[i];
      const float4 b = reinterpret_cast<const float4*>(wr)[i];
      acc = fmaf(a.x, b.x, acc);
      acc = fmaf(a.y, b.y, acc);
      acc = fmaf(a.z, b.z, acc);
      acc = fmaf(a.w, b.w, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
      const float v = acc + (bias ? bias[n] : 0.f);
      out[(long long)r * N + n] = act == 1 ? 1.f / (1.f + expf(-v)) : v;
    }
  }
}

// arg-max of both classifiers' scores + the gather above + the (tone, syllable) pair id, one thread per window: the label pass
// of a train step in one launch (five ATen kernels + tone_dynamics_kernel otherwise).  First maximum wins and a NaN counts as the
// maximum (the first NaN wins), as torch.argmax - a diverged classifier labels a window the same way on both paths.
__global__ void labels_from_scores_kernel(const float* __restrict__ st, const float* __restrict__ ss, const float* __restrict__ table,
                                          float* __restrict__ labels, long long* __restrict__ tone, long long* __restrict__ syl,
                                          int32_t* __restrict__ pair, int32_t* err, int B, int nt, int ns, int n_rows, int n_syl,
                                          int L) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int t = 0, y = 0;
  float best = st[(long long)b * nt];
  for (int i = 1; i < nt; ++i) {
    const float v = st[(long long)b * nt + i];
    if (v > best || (v != v && best == best)) { best = v; t = i; }
  }
  best = ss[(long long)b * ns];
  for (int i = 1; i < ns; ++i) {
    const float v = ss[(long long)b * ns + i];
    if (v > best || (v != v && best == best)) { best = v; y = i; }
  }
  tone[b] = t;
  syl[b] = y;
  if (pair != nullptr) pair[b] = t * n_syl + y;
  const bool ok = t < n_rows;
  if (!ok) *err = 1;
  for (int l = 0; l < L; ++l) {
    labels[((long long)b * 2) * L + l] = (float)y;
    labels[((long long)b * 2 + 1) * L + l] = ok ? table[(long long)t * L + l] : 0.f;
  }
}

static inline unsigned grid_for(long long total, int block = 256, long long cap = 256LL * 32) {
  long long g = (total + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

}  // namespace tl

using namespace tl;

extern "C" const char* tl_last_error(void) { return tl::g_err; }
extern "C" int tl_version(void) { return 100; }
extern "C" int tl_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return TL_ENODEV;
  }
  return n;
}

extern "C" int tl_conv1_fwd(const float* x, const float* w, const float* b, float* P, uint32_t* bits, uint32_t* sign,
                            int64_t S, int T, int ktaps, int C1, int Tp, int Tout, float slope, void* stream) {
  TL_REQUIRE(x && w && b && P && bits, "conv1_fwd: null pointer");
  TL_REQUIRE(S > 0 && S < (1LL << 31), "conv1_fwd: bad S");
  TL_REQUIRE(ktaps >= 1 && ktaps <= MAXKT, "conv1_fwd: ktaps must be 1..%d", MAXKT);
  TL_REQUIRE(C1 % 64 == 0, "conv1_fwd: C1 must be a multiple of 64");
  TL_REQUIRE(Tout >= 0 && Tout <= Tp && 2 * Tout + ktaps - 1 <= T, "conv1_fwd: Tout/Tp/T inconsistent (%d,%d,%d)", Tout, Tp, T);
  TL_REQUIRE((size_t)T * 4 <= 64 * 1024, "conv1_fwd: T too large for the LDS window");
  const bool v4 = C1 % 128 == 0 && C1 <= 1024 && (C1 & (C1 - 1)) == 0;      // 256 % (C1 / 4) == 0: whole rows per pass
#define TL_C1_LAUNCH(KT_)                                                                                                 \
  case KT_:                                                                                                                \
    if (v4)                                                                                                                \
      hipLaunchKernelGGL(conv1_fwd_v4_kernel<KT_>, dim3((unsigned)S), dim3(256), (size_t)T * 4, (hipStream_t)stream, x, w, b, P, \
                         bits, sign, (long long)S, T, C1, Tp, Tout, slope);                                                \
    else                                                                                                                   \
      hipLaunchKernelGGL(conv1_fwd_kernel<KT_>, dim3((unsigned)S), dim3(256), (size_t)T * 4, (hipStream_t)stream, x, w, b, P,    \
                         bits, sign, (long long)S, T, C1, Tp, Tout, slope);                                                \
    break;
  switch (ktaps) {
    TL_C1_LAUNCH(1) TL_C1_LAUNCH(2) TL_C1_LAUNCH(3) TL_C1_LAUNCH(4) TL_C1_LAUNCH(5) TL_C1_LAUNCH(6) TL_C1_LAUNCH(7) TL_C1_LAUNCH(8)
  }
#undef TL_C1_LAUNCH
  return check_launch("conv1_fwd");
}

extern "C" int tl_conv1_fwd_v(const float* x, const float* w, const float* b, float* P, float* V, uint32_t* bits,
                              uint32_t* sign, int64_t S, int T, int ktaps, int C1, int Tp, int Tout, float slope,
                              void* stream) {
  TL_REQUIRE(x && w && b && V && bits, "conv1_fwd_v: null pointer");
  TL_REQUIRE(S > 0 && S < (1LL << 31), "conv1_fwd_v: bad S");
  TL_REQUIRE(ktaps >= 1 && ktaps <= MAXKT, "conv1_fwd_v: ktaps must be 1..%d", MAXKT);
  TL_REQUIRE(C1 % 128 == 0 && C1 <= 1024 && (C1 & (C1 - 1)) == 0, "conv1_fwd_v: C1 must be 128, 256, 512 or 1024");
  TL_REQUIRE(Tp > 0 && Tp % 4 == 0, "conv1_fwd_v: Tp must be a multiple of 4");
  TL_REQUIRE(Tout >= 0 && Tout <= Tp && 2 * Tout + ktaps - 1 <= T, "conv1_fwd_v: Tout/Tp/T inconsistent (%d,%d,%d)", Tout, Tp, T);
  TL_REQUIRE((size_t)T * 4 <= 64 * 1024, "conv1_fwd_v: T too large for the LDS window");
#define TL_C1_LAUNCH(KT_)                                                                                                 \
  case KT_:                                                                                                                \
    hipLaunchKernelGGL(conv1_fwd_vq_kernel<KT_>, dim3((unsigned)S), dim3(256), (size_t)T * 4, (hipStream_t)stream, x, w, b, P, V, \
                       bits, sign, (long long)S, T, C1, Tp, Tout, slope);                                                  \
    break;
  switch (ktaps) {
    TL_C1_LAUNCH(1) TL_C1_LAUNCH(2) TL_C1_LAUNCH(3) TL_C1_LAUNCH(4) TL_C1_LAUNCH(5) TL_C1_LAUNCH(6) TL_C1_LAUNCH(7) TL_C1_LAUNCH(8)
  }
#undef TL_C1_LAUNCH
  return check_launch("conv1_fwd_v");
}

extern "C" int tl_dropout_scale(float* x, int64_t n, float p, uint64_t seed, void* stream) {
  TL_REQUIRE(x && n > 0, "dropout_scale: bad arguments");
  TL_REQUIRE(p >= 0.f && p < 1.f, "dropout_scale: p must be in [0, 1)");
  if (p == 0.f) return TL_OK;
  long long g = (n + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(dropout_scale_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (long long)n, p,
                     1.0f / (1.0f - p), seed);
  return check_launch("dropout_scale");
}

extern "C" int tl_set_step_scalars(float* scalars_dev, uint64_t* seed_dev, float coef_grad, float coef_mom, float bias_corr2,
                                   uint64_t seed, void* stream) {
  TL_REQUIRE(scalars_dev || seed_dev, "set_step_scalars: nothing to set");
  hipLaunchKernelGGL(set_step_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scalars_dev, seed_dev, coef_grad, coef_mom,
                     bias_corr2, seed);
  return check_launch("set_step_scalars");
}

extern "C" int tl_stage_step(float* scalars_dev, uint64_t* seed_dev, float coef_grad, float coef_mom, float bias_corr2, uint64_t seed,
                             const void* const* src, void* const* dst, const int64_t* nbytes, int n, void* stream) {
  TL_REQUIRE(n >= 0 && n <= 4, "stage_step: at most 4 tensors");
  TL_REQUIRE(n == 0 || (src && dst && nbytes), "stage_step: null table");
  stage_args a;
  long long most = 0;
  a.n = n;
  for (int i = 0; i < 4; ++i) {
    a.src[i] = i < n ? (const char*)src[i] : nullptr;
    a.dst[i] = i < n ? (char*)dst[i] : nullptr;
    a.nbytes[i] = i < n ? (long long)nbytes[i] : 0;
    TL_REQUIRE(i >= n || (a.src[i] && a.dst[i] && a.nbytes[i] >= 0), "stage_step: null tensor %d", i);
    if (a.nbytes[i] > most) most = a.nbytes[i];
  }
  long long blocks = (most / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(stage_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, scalars_dev, seed_dev, coef_grad,
                     coef_mom, bias_corr2, seed);
  return check_launch("stage_step");
}

extern "C" int tl_conv1_wgrad(const float* x, const float* G, const uint32_t* bits, float* partial, int nblk,
                              int64_t S, int T, int ktaps, int C1, int Tp, int Tout, void* stream) {
  TL_REQUIRE(x && G && bits && partial, "conv1_wgrad: null pointer");
  TL_REQUIRE(nblk > 0 && S > 0, "conv1_wgrad: bad sizes");
  TL_REQUIRE(ktaps >= 1 && ktaps <= MAXKT, "conv1_wgrad: ktaps must be 1..%d", MAXKT);
  TL_REQUIRE(C1 % 32 == 0 && C1 <= 512, "conv1_wgrad: C1 must be a multiple of 32 and <= 512");
  TL_REQUIRE(Tout >= 0 && Tout <= Tp && 2 * Tout + ktaps - 1 <= T, "conv1_wgrad: Tout/Tp/T inconsistent");
  TL_REQUIRE((size_t)T * 4 <= 64 * 1024, "conv1_wgrad: T too large for the LDS window");
  hipLaunchKernelGGL(conv1_wgrad_kernel, dim3((unsigned)nblk), dim3(256), (size_t)T * 4, (hipStream_t)stream, x, G,
                     bits, partial, (long long)S, T, ktaps, C1, Tp, Tout);
  return check_launch("conv1_wgrad");
}

namespace tl {
// Permutation whose innermost destination dimension is strided in the source while another dimension K is contiguous there
// (the per-step packs of the Linear layer's 63 MB weight): 32 x 32 tiles through LDS, reads run along K, writes along the
// innermost dimension.  The element-per-thread kernel above reads such a source one cache line per element (0.6 TB/s).
template <int K>
__global__ __launch_bounds__(256) void permute_tiled_kernel(const float* __restrict__ src, float* __restrict__ dst, perm_args a) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long tk_n = (a.d[K] + 31) / 32, t3_n = (a.d[3] + 31) / 32;
  long long b = blockIdx.x;
  const long long t3 = b % t3_n; b /= t3_n;
  const long long tk = b % tk_n; b /= tk_n;
  // the two dimensions that are neither K nor 3, in order
  constexpr int P = (K == 0) ? 1 : 0, Q = (K == 2) ? 1 : 2;
  const long long iq = b % a.d[Q], ip = b / a.d[Q];
  long long idx[4];
  idx[P] = ip;
  idx[Q] = iq;
  const bool pq_ok = ip < a.lim[P] && iq < a.lim[Q];
  const long long base = ip * a.s[P] + iq * a.s[Q];
  {
    const long long ik = tk * 32 + tx;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long i3 = t3 * 32 + ty + 8 * r;
      float v = 0.f;
      if (pq_ok && ik < a.d[K] && ik < a.lim[K] && i3 < a.d[3] && i3 < a.lim[3]) v = src[base + ik * a.s[K] + i3 * a.s[3]];
      tile[ty + 8 * r][tx] = v;
    }
  }
  __syncthreads();
  const long long i3 = t3 * 32 + tx;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long ik = tk * 32 + ty + 8 * r;
    if (ik < a.d[K] && i3 < a.d[3]) {
      idx[K] = ik;
      dst[((idx[0] * a.d[1] + idx[1]) * a.d[2] + idx[2]) * a.d[3] + i3] = tile[tx][ty + 8 * r];
    }
  }
}
}  // namespace tl

extern "C" int tl_permute_reduce(const float* src, float* dst, const int64_t dims[4], const int64_t strides[4],
                                 const int64_t lims[4], int nz, int64_t zs, const float* bias_last, void* stream) {
  TL_REQUIRE(src && dst && dims && strides && lims, "permute_reduce: null pointer");
  TL_REQUIRE(nz >= 1, "permute_reduce: nz must be >= 1");
  perm_args a;
  long long total = 1;
  for (int i = 0; i < 4; ++i) {
    TL_REQUIRE(dims[i] >= 1, "permute_reduce: dims must be >= 1");
    a.d[i] = dims[i];
    a.s[i] = strides[i];
    a.lim[i] = lims[i];
    total *= dims[i];
  }
  a.zs = zs;
  a.nz = nz;
  if (nz == 1 && bias_last == nullptr && a.s[3] != 1 && a.d[3] >= 16) {
    // a dimension that is contiguous in the source and long enough for whole read lines: the tiled transpose
    int k = -1;
    for (int i = 0; i < 3; ++i)
      if (a.s[i] == 1 && a.d[i] >= 32) k = i;
    if (k >= 0) {
      const long long blocks = (total / a.d[k] / a.d[3]) * ((a.d[k] + 31) / 32) * ((a.d[3] + 31) / 32);
      if (blocks > 0 && blocks < (1LL << 31)) {
        const dim3 grid((unsigned)blocks);
        if (k == 0)
          hipLaunchKernelGGL(permute_tiled_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, a);
        else if (k == 1)
          hipLaunchKernelGGL(permute_tiled_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, a);
        else
          hipLaunchKernelGGL(permute_tiled_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, a);
        return check_launch("permute_tiled");
      }
    }
  }
  if (nz >= 64 && total <= 16384)
    hipLaunchKernelGGL(permute_reduce_zpar_kernel, dim3(grid_for(total * 64)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, bias_last, a, total);
  else
    hipLaunchKernelGGL(permute_reduce_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       bias_last, a, total);
  return check_launch("permute_reduce");
}

extern "C" int tl_sum_slabs2(const float* srcA, float* dstA, int64_t nA, const float* srcB, float* dstB, int64_t nB, int nz,
                             void* stream) {
  TL_REQUIRE(srcA && dstA && nA > 0 && nz > 0 && nB >= 0 && (nB == 0 || (srcB && dstB)), "sum_slabs2: bad arguments");
  hipLaunchKernelGGL(sum_slabs2_kernel, dim3(grid_for(nA + nB)), dim3(256), 0, (hipStream_t)stream, srcA, dstA, (long long)nA, srcB,
                     dstB, (long long)nB, nz);
  return check_launch("sum_slabs2");
}

extern "C" int tl_colsum(const float* G, float* partial, int nblk, int64_t rows, int ncols, int ld, int Tp,
                         int Tvalid, void* stream) {
  TL_REQUIRE(G && partial && nblk > 0 && rows > 0 && ncols > 0 && ld >= ncols && Tp > 0, "colsum: bad arguments");
  TL_REQUIRE(ncols % 4 == 0 && ncols <= 1024 && ld % 4 == 0, "colsum: ncols must be a multiple of 4 and <= 1024");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, G, partial,
                     (long long)rows, ncols, ld, Tp, Tvalid);
  return check_launch("colsum");
}

extern "C" int tl_lstm_cell_fwd(const float* hh, const float* x_t, const float* w_ih, const float* b_ih,
                                const float* b_hh, const float* c_prev, float* act, float* c, float* h, int U, int H,
                                int in_dim, int ld_hh, void* stream) {
  TL_REQUIRE(x_t && w_ih && b_ih && b_hh && act && c && h, "lstm_cell_fwd: null pointer");
  TL_REQUIRE(U > 0 && H > 0 && in_dim > 0, "lstm_cell_fwd: bad sizes");
  const long long total = (long long)U * H;
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     hh, x_t, w_ih, b_ih, b_hh, c_prev, act, c, h, U, H, in_dim, ld_hh);
  return check_launch("lstm_cell_fwd");
}

extern "C" int tl_lstm_gw(const float* g, const float* W, float* slab, int U, int64_t N, int K, int64_t ldg, int64_t ldw,
                          int rows_per_block, void* stream) {
  TL_REQUIRE(g && W && slab, "lstm_gw: null pointer");
  TL_REQUIRE(U >= 1 && U <= HW_MAXU && N > 0 && K > 0 && K % 4 == 0, "lstm_gw: 1 <= U <= %d, K %% 4 == 0 needed", HW_MAXU);
  TL_REQUIRE(rows_per_block >= 8 && rows_per_block <= GW_MAXROWS, "lstm_gw: rows_per_block must be 8..%d", GW_MAXROWS);
  TL_REQUIRE(ldg >= N && ldw >= K && ldw % 4 == 0, "lstm_gw: leading dimensions too small (ldw a multiple of 4)");
  TL_REQUIRE((((uintptr_t)W | (uintptr_t)slab) & 15) == 0, "lstm_gw: W and slab must be 16-byte aligned");
  const long long nb = (N + rows_per_block - 1) / rows_per_block;
  TL_REQUIRE(nb <= 65535, "lstm_gw: more than 65 535 row blocks");
  hipLaunchKernelGGL(lstm_gw_kernel, dim3((unsigned)((K + GW_COLS - 1) / GW_COLS), (unsigned)nb), dim3(256), 0, (hipStream_t)stream, g, W,
                     slab, U, (long long)N, K, (long long)ldg, (long long)ldw, rows_per_block);
  return check_launch("lstm_gw");
}

// All T steps of an inference LSTM, one fused launch per step (lstm_step_fused_kernel).  wp: the recurrent weight packed
// unit-major, row 4 u + g = W_hh row g H + u; xp keeps the torch gate-major columns.  h_a / h_b: ping-pong
// buffers; the final state is in h_a when T is odd, h_b when T is even (returned through *last_in_b).
extern "C" int tl_lstm_infer_seq_fused(const float* xp, int64_t xp_row_stride, const float* wp, float* h_a, float* h_b,
                                       float* c, int B, int H, int T, int* last_in_b, void* stream) {
  TL_REQUIRE(xp && wp && h_a && h_b && c && last_in_b && B > 0 && H > 0 && T > 0, "lstm_infer_seq_fused: bad arguments");
  TL_REQUIRE(H % 8 == 0, "lstm_infer_seq_fused: hidden width must be a multiple of 8 (pad the packed weights)");
  TL_REQUIRE(xp_row_stride >= (int64_t)T * 4 * H, "lstm_infer_seq_fused: xp rows are (b, t): row stride must cover T steps");
  const dim3 grid((unsigned)(H / 8), (unsigned)((B + 31) / 32));
  TL_REQUIRE(grid.y <= 65535u, "lstm_infer_seq_fused: batch too large");
  for (int t = 0; t < T; ++t) {
    const float* hin = (t & 1) ? h_a : h_b;
    float* hout = (t & 1) ? h_b : h_a;
    hipLaunchKernelGGL(lstm_step_fused_kernel, grid, dim3(512), 0, (hipStream_t)stream, xp + (long long)t * 4 * H,
                       (long long)xp_row_stride, wp, hin, hout, c, B, H, t == 0 ? 1 : 0);
  }
  *last_in_b = (T & 1) ? 0 : 1;
  return check_launch("lstm_infer_seq_fused");
}

extern "C" int tl_lstm_cell_bwd(const float* dh, const float* dh_rec, const float* dc_next, const float* act,
                                const float* c, const float* c_prev, float* dgates, float* dgates_t, float* dc_prev,
                                int U, int H, int ldt, void* stream) {
  TL_REQUIRE(act && c && dgates && dc_prev, "lstm_cell_bwd: null pointer");
  TL_REQUIRE(U > 0 && H > 0 && (!dgates_t || ldt >= U), "lstm_cell_bwd: bad sizes");
  const long long total = (long long)U * H;
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dh, dh_rec, dc_next, act, c, c_prev, dgates, dgates_t, dc_prev, U, H, ldt);
  return check_launch("lstm_cell_bwd");
}

extern "C" int tl_lstm_ih_grad(const float* dgates, const float* x, float* dw_ih, float* db, float* db2, int L, int U, int H,
                               int in_dim, void* stream) {
  TL_REQUIRE(dgates && x && dw_ih && db, "lstm_ih_grad: null pointer");
  TL_REQUIRE(in_dim >= 1 && in_dim <= MAXKT, "lstm_ih_grad: in_dim must be 1..%d", MAXKT);
  const long long total = 4LL * H;
  if (total <= 8192 && (long long)L * U >= 64)
    hipLaunchKernelGGL(lstm_ih_grad_wave_kernel, dim3((unsigned)total), dim3(64), 0, (hipStream_t)stream, dgates, x, dw_ih,
                       db, db2, L * U, H, in_dim);
  else
    hipLaunchKernelGGL(lstm_ih_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dgates, x, dw_ih, db, db2, L, U, H, in_dim);
  return check_launch("lstm_ih_grad");
}

extern "C" int tl_concat_pack(const float* O5, const float* h, const int32_t* uid, float* Xc, int B, int C, int Tp,
                              int lat, int Cc, int Lc, int ld5, int ldh, int ldx, float p_drop, uint64_t seed,
                              int64_t drop_row0, void* stream) {
  TL_REQUIRE(O5 && h && uid && Xc, "concat_pack: null pointer");
  TL_REQUIRE(ldx >= Cc + Lc && ld5 >= Cc && lat <= Tp && p_drop >= 0.f && p_drop < 1.f, "concat_pack: bad arguments");
  const long long total = (long long)B * C * Tp * ldx;
  // (the grid-stride loop adds up to 2^31 to the index before it tests it: 32-bit indices below 2^31 elements only)
  if (total < (1LL << 31))
    hipLaunchKernelGGL(concat_pack_kernel<unsigned>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, O5, h, uid, Xc, B,
                       C, Tp, lat, Cc, Lc, ld5, ldh, ldx, p_drop, seed, (long long)drop_row0);
  else
    hipLaunchKernelGGL(concat_pack_kernel<long long>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, O5, h, uid, Xc, B,
                       C, Tp, lat, Cc, Lc, ld5, ldh, ldx, p_drop, seed, (long long)drop_row0);
  return check_launch("concat_pack");
}

extern "C" int tl_concat_unpack_bwd(const float* dXc, const float* O5, const int32_t* members, const int32_t* offsets,
                                    float* G5, float* dh, int B, int U, int C, int Tp, int lat, int Cc, int Lc, int ld5,
                                    int ldh, int ldx, float slope, float p_drop, uint64_t seed, int64_t drop_row0,
                                    void* stream) {
  TL_REQUIRE(dXc && O5 && offsets && G5 && dh, "concat_unpack_bwd: null pointer");
  TL_REQUIRE(ldx >= Cc + Lc && ld5 >= Cc && lat <= Tp && p_drop >= 0.f && p_drop < 1.f, "concat_unpack_bwd: bad arguments");
  const long long rows = (long long)B * C * Tp;
  if (rows * Cc < (1LL << 31))
    hipLaunchKernelGGL(concat_g5_kernel<unsigned>, dim3(grid_for(rows * Cc)), dim3(256), 0, (hipStream_t)stream, dXc, O5, G5, rows,
                       Cc, ld5, ldx, slope, p_drop, seed, (long long)drop_row0);
  else
    hipLaunchKernelGGL(concat_g5_kernel<long long>, dim3(grid_for(rows * Cc)), dim3(256), 0, (hipStream_t)stream, dXc, O5, G5, rows,
                       Cc, ld5, ldx, slope, p_drop, seed, (long long)drop_row0);
  int rc = check_launch("concat_g5");
  if (rc) return rc;
  const long long total = (long long)U * Lc * lat * C;
  hipLaunchKernelGGL(concat_dh_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dXc, members, offsets,
                     dh, U, C, Tp, lat, Cc, Lc, ldh, ldx, B);
  return check_launch("concat_dh");
}

extern "C" int tl_l1_mcd(const float* out, const float* targets, float* dout, float* stats, int B, int D, int ldd,
                         int trunc_targets, float grad_scale, void* stream) {
  TL_REQUIRE(out && targets && stats && B > 0 && D > 0 && ldd >= D, "l1_mcd: bad arguments");
  hipLaunchKernelGGL(l1_mcd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, out, targets, dout, stats, B, D, ldd,
                     trunc_targets, grad_scale);
  return check_launch("l1_mcd");
}

extern "C" int tl_nadam(float* p, const float* g, float* m, float* v, int64_t n, float coef_grad, float coef_mom,
                        float beta1, float beta2, float bias_corr2, float eps, float weight_decay, float grad_scale,
                        void* stream) {
  TL_REQUIRE(p && g && m && v && n > 0, "nadam: bad arguments");
  TL_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "nadam: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(nadam_kernel, dim3(grid_for(n / 4 + 1, 256, 256LL * 8)), dim3(256), 0, (hipStream_t)stream, p, g,
                     m, v, (long long)n, coef_grad, coef_mom, beta1, beta2, bias_corr2, eps, weight_decay, grad_scale);
  return check_launch("nadam");
}

extern "C" int tl_nadam_multi(const tl_nadam_entry* entries_dev, int count, int64_t total_blocks, float coef_grad,
                              float coef_mom, float beta1, float beta2, float bias_corr2, float eps, float weight_decay,
                              float grad_scale, void* stream) {
  TL_REQUIRE(entries_dev != nullptr && count > 0, "nadam_multi: empty table");
  TL_REQUIRE(total_blocks > 0 && total_blocks < (1LL << 31), "nadam_multi: bad block count %lld", (long long)total_blocks);
  hipLaunchKernelGGL(nadam_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, entries_dev, count,
                     coef_grad, coef_mom, beta1, beta2, bias_corr2, eps, weight_decay, grad_scale, (const float*)nullptr);
  return check_launch("nadam_multi");
}

extern "C" int tl_nadam_multi_dev(const tl_nadam_entry* entries_dev, int count, int64_t total_blocks, const float* scalars_dev,
                                  float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  TL_REQUIRE(entries_dev != nullptr && count > 0 && scalars_dev != nullptr, "nadam_multi_dev: empty table / null scalars");
  TL_REQUIRE(total_blocks > 0 && total_blocks < (1LL << 31), "nadam_multi_dev: bad block count %lld", (long long)total_blocks);
  hipLaunchKernelGGL(nadam_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, entries_dev, count,
                     0.f, 0.f, beta1, beta2, 1.f, eps, weight_decay, grad_scale, scalars_dev);
  return check_launch("nadam_multi_dev");
}

extern "C" int tl_nadam_multi_chunk(void) { return NM_CHUNK; }

static int nadam_lowrank_launch(float* p, float* m, float* v, const float* fa, const float* fb, int kr, int rows, int cols,
                                int ldfa, int ldfb, float coef_grad, float coef_mom, float beta1, float beta2,
                                float bias_corr2, float eps, float weight_decay, float grad_scale, float* dh_slab, int U,
                                int row_tiles, void* stream) {
  TL_REQUIRE(p && m && v && rows > 0 && cols > 0, "nadam_lowrank: bad arguments");
  TL_REQUIRE(kr >= 0 && kr <= LR_MAXK && (kr == 0 || (fa && fb)), "nadam_lowrank: rank must be 0..%d with both factors", LR_MAXK);
  TL_REQUIRE(cols % 4 == 0 && ldfb % 4 == 0 && ldfa >= rows && ldfb >= cols, "nadam_lowrank: cols / ldfb must be multiples of 4, ld >= extent");
  TL_REQUIRE((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)fb) & 15) == 0, "nadam_lowrank: pointers must be 16-byte aligned");
  const bool dh = dh_slab != nullptr;
  if (dh) {
    TL_REQUIRE(U >= 1 && U <= LR_MAXU && U <= kr && row_tiles >= 1, "nadam_lowrank_dh: 1 <= U <= min(%d, rank) and row_tiles >= 1 needed", LR_MAXU);
    TL_REQUIRE(((uintptr_t)dh_slab & 15) == 0, "nadam_lowrank_dh: dh_slab must be 16-byte aligned");
  }
  const int tiles = (rows + LR_TR - 1) / LR_TR;
  const unsigned gx = (unsigned)(dh ? (tiles + row_tiles - 1) / row_tiles : tiles), gy = (unsigned)((cols + LR_TC - 1) / LR_TC);
  TL_REQUIRE(gy <= 65535u, "nadam_lowrank: more than 16.7 M columns");
  size_t lds = (size_t)kr * (LR_TR + LR_TC) * 4;
  if (dh && lds < (size_t)3 * LR_MAXU * LR_TC * 4) lds = (size_t)3 * LR_MAXU * LR_TC * 4;      // (the reduction scratch overlays the tiles)
  const void* fn = dh ? reinterpret_cast<const void*>(nadam_lowrank_kernel<true>) : reinterpret_cast<const void*>(nadam_lowrank_kernel<false>);
  if (lds > 64 * 1024) {      // ranks 57..64 need more than 64 KB of the CU's 160 KB: raise the per-block limit once
    static thread_local int raised_on[2] = {-1, -1};
    int devid = 0;
    (void)hipGetDevice(&devid);
    if (raised_on[dh] != devid) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LR_MAXK * (LR_TR + LR_TC) * 4);
      if (e != hipSuccess) {
        set_error("nadam_lowrank: cannot raise the dynamic LDS limit for rank %d: %s", kr, hipGetErrorString(e));
        return TL_ELAUNCH;
      }
      raised_on[dh] = devid;
    }
  }
  if (dh)
    hipLaunchKernelGGL(nadam_lowrank_kernel<true>, dim3(gx, gy), dim3(256), lds, (hipStream_t)stream, p, m, v, fa, fb, kr, rows,
                       cols, ldfa, ldfb, coef_grad, coef_mom, beta1, beta2, bias_corr2, eps, weight_decay, grad_scale, dh_slab, U,
                       row_tiles);
  else
    hipLaunchKernelGGL(nadam_lowrank_kernel<false>, dim3(gx, gy), dim3(256), lds, (hipStream_t)stream, p, m, v, fa, fb, kr, rows,
                       cols, ldfa, ldfb, coef_grad, coef_mom, beta1, beta2, bias_corr2, eps, weight_decay, grad_scale, nullptr, 0, 1);
  return check_launch("nadam_lowrank");
}

extern "C" int tl_nadam_lowrank(float* p, float* m, float* v, const float* fa, const float* fb, int kr, int rows, int cols,
                                int ldfa, int ldfb, float coef_grad, float coef_mom, float beta1, float beta2,
                                float bias_corr2, float eps, float weight_decay, float grad_scale, void* stream) {
  return nadam_lowrank_launch(p, m, v, fa, fb, kr, rows, cols, ldfa, ldfb, coef_grad, coef_mom, beta1, beta2, bias_corr2, eps,
                              weight_decay, grad_scale, nullptr, 0, 1, stream);
}

extern "C" int tl_nadam_lowrank_dh(float* p, float* m, float* v, const float* fa, const float* fb, int kr, int rows, int cols,
                                   int ldfa, int ldfb, float coef_grad, float coef_mom, float beta1, float beta2,
                                   float bias_corr2, float eps, float weight_decay, float grad_scale, float* dh_slab, int U,
                                   int row_tiles, void* stream) {
  TL_REQUIRE(dh_slab != nullptr, "nadam_lowrank_dh: dh_slab needed (ceil(rows / (32 row_tiles)) x U x cols floats)");
  return nadam_lowrank_launch(p, m, v, fa, fb, kr, rows, cols, ldfa, ldfb, coef_grad, coef_mom, beta1, beta2, bias_corr2, eps,
                              weight_decay, grad_scale, dh_slab, U, row_tiles, stream);
}

extern "C" int tl_splitk_bias_lrelu(const float* slab, const float* bias, float* out, int nz, int64_t n, int ncols, float slope,
                                    void* stream) {
  TL_REQUIRE(slab && out && nz > 0 && n > 0 && ncols > 0, "splitk_bias_lrelu: bad arguments");
  hipLaunchKernelGGL(splitk_bias_lrelu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, slab, bias, out, nz,
                     (long long)n, ncols, slope);
  return check_launch("splitk_bias_lrelu");
}

extern "C" int tl_linear_rows(const float* x, const float* w, const float* bias, float* out, int B, int K, int N, int64_t ldx,
                              int act, void* stream) {
  using namespace tl;
  TL_REQUIRE(x && w && out, "linear_rows: null pointer");
  TL_REQUIRE(B > 0 && K > 0 && N > 0 && N <= 64, "linear_rows: bad sizes (1 <= N <= 64 output columns)");
  TL_REQUIRE(act == 0 || act == 1, "linear_rows: act must be 0 (none) or 1 (sigmoid)");
  TL_REQUIRE(K % 4 == 0 && ldx >= K && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0,
             "linear_rows: K and ldx must be multiples of 4 and x, w 16-byte aligned (rows are read as float4)");
  hipLaunchKernelGGL(linear_rows_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, K, N, (long long)ldx, act);
  return check_launch("linear_rows");
}

extern "C" int tl_labels_from_scores(const float* tone_scores, const float* syl_scores, const float* table, float* labels,
                                     int64_t* tone, int64_t* syl, int32_t* pair, int32_t* err, int B, int n_tone_cls,
                                     int n_syl_cls, int n_rows, int n_syl, int L, void* stream) {
  TL_REQUIRE(tone_scores && syl_scores && table && labels && tone && syl && err, "labels_from_scores: null pointer");
  TL_REQUIRE(B > 0 && n_tone_cls > 0 && n_syl_cls > 0 && n_rows > 0 && L > 0, "labels_from_scores: bad sizes");
  hipLaunchKernelGGL(labels_from_scores_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, tone_scores, syl_scores,
                     table, labels, (long long*)tone, (long long*)syl, pair, err, B, n_tone_cls, n_syl_cls, n_rows, n_syl, L);
  return check_launch("labels_from_scores");
}

extern "C" int tl_tone_dynamics(const int64_t* tone, const int64_t* syl, const float* table, float* labels,
                                int32_t* err, int B, int n_tones, int L, void* stream) {
  TL_REQUIRE(tone && syl && table && labels && err && B > 0 && n_tones > 0 && L > 0, "tone_dynamics: bad arguments");
  const int total = B * L;
  hipLaunchKernelGGL(tone_dynamics_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)tone, (const long long*)syl, table, labels, err, B, n_tones, L);
  return check_launch("tone_dynamics");
}

// SynthesisLite kernels (reference models/synthesis_models.py:201-296): Conv1d + BatchNorm1d +
// LeakyReLU + MaxPool1d(2) blocks, the small label LSTM (whole sequence in one launch), and the
// concat + dropout glue.  The two Linear layers reuse the MFMA GEMM kernels.  At the reference's
// Lite sizes (32 ch x 200 samples, batch 64) every tensor is KBs-MBs: the step is launch-bound, so
// each kernel does a whole layer stage per launch and no kernel needs more than one pass.
#include "tonal_common.h"
#include <math.h>
#include <type_traits>

namespace tl {

constexpr int LT = 64;   // time tile of the direct conv kernels

// z[b][o][t] = bias[o] + sum_{i,j} w[o][i][j] * x[b][i][t + j - pad]   (zero padding)
// part[(b*ntile + tile)][o][0..1] = (sum_t z, sum_t z^2) over the tile
__global__ __launch_bounds__(512) void lite_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ z,
                                                            float* __restrict__ part, int Cin, int Cout, int T, int k,
                                                            int pad) {
  extern __shared__ __attribute__((aligned(16))) float xs[];      // [Cin][LT + k - 1]
  const int b = blockIdx.y, tile = blockIdx.x, t0 = tile * LT;
  const int W = LT + k - 1;
  for (int i = threadIdx.x; i < Cin * W; i += blockDim.x) {
    const int ci = i / W, tt = t0 + (i % W) - pad;
    xs[i] = (tt >= 0 && tt < T) ? x[((long long)b * Cin + ci) * T + tt] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: weight addresses become scalar
  const int t = t0 + lane;
  const bool ok = t < T;
  auto finish = [&](int o, float acc) {
    if (ok) z[((long long)b * Cout + o) * T + t] = acc;
    float s1 = ok ? acc : 0.f, s2 = ok ? acc * acc : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s1 += __shfl_down(s1, off);
      s2 += __shfl_down(s2, off);
    }
    if (lane == 0 && part) {
      float* pp = part + (((long long)b * gridDim.x + tile) * Cout + o) * 2;
      pp[0] = s1;
      pp[1] = s2;
    }
  };
  // output channels a lane accumulates at once (one LDS read feeds OB FMAs).  4 with eight waves, not 8 with four: the
  // weights of an input channel are then ONE batch of scalar loads behind one drained wait instead of two
  constexpr int OB = 4;
  const int nwaves = blockDim.x >> 6;
  auto blocked = [&](auto KC) {
    constexpr int K = decltype(KC)::value;            // compile-time tap count: the j loops unroll without branches
    for (int o0 = wave * OB; o0 < Cout; o0 += nwaves * OB) {
      float acc[OB];
#pragma unroll
      for (int u = 0; u < OB; ++u) acc[u] = bias[o0 + u];
      for (int ci = 0; ci < Cin; ++ci) {
        float xv[K];
#pragma unroll
        for (int j = 0; j < K; ++j) xv[j] = xs[ci * W + lane + j];
#pragma unroll
        for (int u = 0; u < OB; ++u) {
          const float* wo = w + ((long long)(o0 + u) * Cin + ci) * K;       // wave-uniform: scalar loads
#pragma unroll
          for (int j = 0; j < K; ++j) acc[u] = fmaf(wo[j], xv[j], acc[u]);  // same (ci, j) order as the scalar loop below
        }
      }
#pragma unroll
      for (int u = 0; u < OB; ++u) finish(o0 + u, acc[u]);
    }
  };
  if (Cout % (nwaves * OB) == 0 && k == 5) {
    blocked(std::integral_constant<int, 5>{});
  } else if (Cout % (nwaves * OB) == 0 && k == 3) {
    blocked(std::integral_constant<int, 3>{});
  } else {
    for (int o = wave; o < Cout; o += nwaves) {
      float acc = bias[o];
      const float* wo = w + (long long)o * Cin * k;
      for (int ci = 0; ci < Cin; ++ci)
        for (int j = 0; j < k; ++j) acc = fmaf(wo[ci * k + j], xs[ci * W + lane + j], acc);
      finish(o, acc);
    }
  }
}

// mean / rstd from the partial sums (fixed order), running-stat update (momentum, unbiased var)
__global__ __launch_bounds__(64) void lite_bn_finalize_kernel(const float* __restrict__ part, float* __restrict__ mean,
                                                              float* __restrict__ rstd, float* __restrict__ run_mean,
                                                              float* __restrict__ run_var, int nparts, int C, long long count,
                                                              float momentum, float eps, int training,
                                                              long long* __restrict__ tracked) {
  const int c = blockIdx.x;                      // one wave per channel, lanes stride over the partials
  const int lane = threadIdx.x;
  if (training && tracked != nullptr && c == 0 && lane == 0) tracked[0] += 1;     // BatchNorm's num_batches_tracked
  if (training) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < nparts; i += 64) {
      s1 += part[((long long)i * C + c) * 2];
      s2 += part[((long long)i * C + c) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1 += __shfl_down(s1, o);
      s2 += __shfl_down(s2, o);
    }
    if (lane == 0) {
      const double m = s1 / (double)count;
      double var = s2 / (double)count - m * m;
      if (var < 0.0) var = 0.0;
      mean[c] = (float)m;
      rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
      const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
  } else if (lane == 0) {
    mean[c] = run_mean[c];
    rstd[c] = 1.f / sqrtf(run_var[c] + eps);
  }
}

// y[b][c][p] = max_{a<2} lrelu(gamma*(z[2p+a]-mean)*rstd + beta)
__global__ __launch_bounds__(256) void lite_bn_act_pool_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ y,
                                                                   long long total, int C, int T, float slope) {
  const int Tp = T / 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % Tp);
    const long long bc = i / Tp;
    const int c = (int)(bc % C);
    const float g = gamma[c] * rstd[c], sh = beta[c] - mean[c] * g;
    const float a0 = lrelu(fmaf(z[bc * T + 2 * p], g, sh), slope);
    const float a1 = lrelu(fmaf(z[bc * T + 2 * p + 1], g, sh), slope);
    y[i] = a1 > a0 ? a1 : a0;
  }
}

// backward of pool + lrelu: dbn[b][c][t] (gradient w.r.t. the BN output), and per-(b) partial sums
// (sum dbn, sum dbn*xhat) per channel
__global__ __launch_bounds__(256) void lite_bn_act_pool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ dbn, float* __restrict__ part, int C,
                                                                   int T, float slope) {
  __shared__ float r1[256], r2[256];
  const int b = blockIdx.y, c = blockIdx.x;
  const int Tp = T / 2;
  const float g = gamma[c] * rstd[c], sh = beta[c] - mean[c] * g;
  const long long base = ((long long)b * C + c) * T;
  float s1 = 0.f, s2 = 0.f;
  for (int p = threadIdx.x; p < (T + 1) / 2; p += blockDim.x) {
    float d0 = 0.f, d1 = 0.f;
    if (p < Tp) {
      const float z0 = z[base + 2 * p], z1 = z[base + 2 * p + 1];
      const float n0 = fmaf(z0, g, sh), n1 = fmaf(z1, g, sh);
      const float a0 = lrelu(n0, slope), a1 = lrelu(n1, slope);
      const float gy = dy[((long long)b * C + c) * Tp + p];
      if (a1 > a0) d1 = gy * (n1 > 0.f ? 1.f : slope); else d0 = gy * (n0 > 0.f ? 1.f : slope);
      dbn[base + 2 * p] = d0;
      dbn[base + 2 * p + 1] = d1;
      s1 += d0 + d1;
      s2 += d0 * (z0 - mean[c]) * rstd[c] + d1 * (z1 - mean[c]) * rstd[c];
    } else if (2 * p < T) {
      dbn[base + 2 * p] = 0.f;          // odd T: the last sample is dropped by the pool
    }
  }
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      r1[threadIdx.x] += r1[threadIdx.x + off];
      r2[threadIdx.x] += r2[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[((long long)b * C + c) * 2] = r1[0];
    part[((long long)b * C + c) * 2 + 1] = r2[0];
  }
}

// dgamma, dbeta and, in place, dz = gamma*rstd*(dbn - mean(dbn) - xhat*mean(dbn*xhat))  (train)
//                                dz = gamma*rstd*dbn                                      (eval)
// one workgroup per (channel, window): the channel's two sums over the batch partials are re-formed by the first wave of
// every workgroup (B loads per lane-strided pass, the same fp64 tree in every workgroup: identical values everywhere), so
// no separate reduction launch stands between the partials and their use; window 0 also writes dgamma / dbeta
__global__ __launch_bounds__(256) void lite_bn_dz_kernel(float* __restrict__ dbn, const float* __restrict__ z,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ part,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C, int T,
                                                         long long count, int training) {
  __shared__ float sm[2];
  const int c = blockIdx.x, b = blockIdx.y;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < B; i += 64) {
      s1 += part[((long long)i * C + c) * 2];
      s2 += part[((long long)i * C + c) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1 += __shfl_down(s1, o);
      s2 += __shfl_down(s2, o);
    }
    if (lane == 0) {
      sm[0] = (float)s1;
      sm[1] = (float)s2;
      if (b == 0) {
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
      }
    }
  }
  __syncthreads();
  const float su1 = sm[0], su2 = sm[1];
  const float gr = gamma[c] * rstd[c], mu = mean[c], rs = rstd[c];
  const long long row = ((long long)b * C + c) * T;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    float d = dbn[row + t];
    if (training) {
      const float xh = (z[row + t] - mu) * rs;
      d = d - su1 / (float)count - xh * su2 / (float)count;
    }
    dbn[row + t] = gr * d;
  }
}

// conv backward: dx[b][i][t] = sum_{o,j} dz[b][o][t - j + pad] w[o][i][j];  per-b weight-gradient partials
__global__ __launch_bounds__(512) void lite_conv_dx_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                           float* __restrict__ dx, int Cin, int Cout, int T, int k, int pad) {
  extern __shared__ __attribute__((aligned(16))) float ds[];      // [Cout][LT + k - 1]
  const int b = blockIdx.y, t0 = blockIdx.x * LT;
  const int W = LT + k - 1;
  for (int i = threadIdx.x; i < Cout * W; i += blockDim.x) {
    const int o = i / W, tt = t0 + (i % W) - (k - 1 - pad);
    ds[i] = (tt >= 0 && tt < T) ? dz[((long long)b * Cout + o) * T + tt] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = t0 + lane;
  constexpr int CB = 4;        // input channels a lane accumulates at once (eight waves: see lite_conv_fwd_kernel)
  const int nwaves = blockDim.x >> 6;
  auto blocked = [&](auto KC) {
    constexpr int K = decltype(KC)::value;
    for (int c0 = wave * CB; c0 < Cin; c0 += nwaves * CB) {
      float acc[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) acc[u] = 0.f;
      for (int o = 0; o < Cout; ++o) {
        float dv[K];
#pragma unroll
        for (int j = 0; j < K; ++j) dv[j] = ds[o * W + lane + (K - 1 - j)];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
          const float* wr = w + ((long long)o * Cin + c0 + u) * K;
#pragma unroll
          for (int j = 0; j < K; ++j) acc[u] = fmaf(wr[j], dv[j], acc[u]);
        }
      }
      if (t < T) {
#pragma unroll
        for (int u = 0; u < CB; ++u) dx[((long long)b * Cin + c0 + u) * T + t] = acc[u];
      }
    }
  };
  if (Cin % (nwaves * CB) == 0 && k == 5) {
    blocked(std::integral_constant<int, 5>{});
  } else if (Cin % (nwaves * CB) == 0 && k == 3) {
    blocked(std::integral_constant<int, 3>{});
  } else {
    for (int ci = wave; ci < Cin; ci += nwaves) {
      float acc = 0.f;
      for (int o = 0; o < Cout; ++o)
        for (int j = 0; j < k; ++j)      // dz index t - j + pad = t0 + lane + (k-1-j) - (k-1-pad)
          acc = fmaf(w[((long long)o * Cin + ci) * k + j], ds[o * W + lane + (k - 1 - j)], acc);
      if (t < T) dx[((long long)b * Cin + ci) * T + t] = acc;
    }
  }
}
// dwpart[b][o][i][j] = sum_t dz[b][o][t] x[b][i][t + j - pad];  dbpart[b][o] = sum_t dz[b][o][t]
// One weight element per thread; the block's rows of x[b] (all input channels) and dz[b] (the output channels its 256
// elements touch) are staged in LDS once, so the T-long dot products read LDS instead of 2 T global loads per thread.
__global__ __launch_bounds__(256) void lite_conv_dw_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                           float* __restrict__ dwpart, float* __restrict__ dbpart, int Cin,
                                                           int Cout, int T, int k, int pad, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) float sm[];      // xs[Cin][T], dzs[no][T]
  const int b = blockIdx.x;
  const int n = Cout * Cin * k;
  const int e0 = blockIdx.y * blockDim.x;
  const int e = e0 + threadIdx.x;       // one weight element per thread
  const int per_o = Cin * k;
  const int o_lo = e0 / per_o;
  const int e_hi = min(n, e0 + (int)blockDim.x) - 1;
  const int no = e_hi / per_o - o_lo + 1;
  if (use_lds) {
    float* xs = sm;
    float* dzs = sm + (long long)Cin * T;
    for (int i = threadIdx.x; i < Cin * T; i += blockDim.x) xs[i] = x[(long long)b * Cin * T + i];
    for (int i = threadIdx.x; i < no * T; i += blockDim.x) dzs[i] = dz[((long long)b * Cout + o_lo) * T + i];
    __syncthreads();
    if (e < n) {
      const int j = e % k, ci = (e / k) % Cin, o = e / per_o;
      const float* dzr = dzs + (o - o_lo) * T;
      const float* xr = xs + ci * T;
      const int lo = max(0, pad - j), hi = min(T, T + pad - j);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int t = lo;
      for (; t + 4 <= hi; t += 4) {
        a0 = fmaf(dzr[t], xr[t + j - pad], a0);
        a1 = fmaf(dzr[t + 1], xr[t + 1 + j - pad], a1);
        a2 = fmaf(dzr[t + 2], xr[t + 2 + j - pad], a2);
        a3 = fmaf(dzr[t + 3], xr[t + 3 + j - pad], a3);
      }
      for (; t < hi; ++t) a0 = fmaf(dzr[t], xr[t + j - pad], a0);
      dwpart[(long long)b * n + e] = (a0 + a1) + (a2 + a3);
    }
  } else if (e < n) {
    const int j = e % k, ci = (e / k) % Cin, o = e / per_o;
    const float* dzr = dz + ((long long)b * Cout + o) * T;
    const float* xr = x + ((long long)b * Cin + ci) * T;
    const int lo = max(0, pad - j), hi = min(T, T + pad - j);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int t = lo;
    for (; t + 4 <= hi; t += 4) {
      a0 = fmaf(dzr[t], xr[t + j - pad], a0);
      a1 = fmaf(dzr[t + 1], xr[t + 1 + j - pad], a1);
      a2 = fmaf(dzr[t + 2], xr[t + 2 + j - pad], a2);
      a3 = fmaf(dzr[t + 3], xr[t + 3 + j - pad], a3);
    }
    for (; t < hi; ++t) a0 = fmaf(dzr[t], xr[t + j - pad], a0);
    dwpart[(long long)b * n + e] = (a0 + a1) + (a2 + a3);
  }
  if (blockIdx.y == 0) {
    // bias partials: a wave per output channel, lanes across time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = wave; o < Cout; o += 4) {
      const float* dzr = dz + ((long long)b * Cout + o) * T;
      float acc = 0.f;
      for (int t = lane; t < T; t += 64) acc += dzr[t];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
      if (lane == 0) dbpart[(long long)b * Cout + o] = acc;
    }
  }
}

// ---- small LSTM, whole sequence per launch, one workgroup per batch row ---------------------
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

__global__ __launch_bounds__(256) void lite_lstm_fwd_kernel(const float* __restrict__ xl, const float* __restrict__ w_ih,
                                                            const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                                            const float* __restrict__ b_hh, float* __restrict__ act,
                                                            float* __restrict__ cs, float* __restrict__ hs, int L, int H,
                                                            int in_dim) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // h[H], c[H], gates[4H]
  float* h = sm;
  float* c = sm + H;
  float* gt = sm + 2 * H;
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < H; i += blockDim.x) h[i] = c[i] = 0.f;
  __syncthreads();
  // H == 64 (the reference's default, models/synthesis_models.py:211): thread r keeps row r of W_hh in registers for the
  // whole sequence (4H = 256 threads x 64 floats), so a step is 64 FMAs on broadcast LDS reads of h with no memory
  // traffic - the generic loop below re-reads its 16 KB row set from L2 every step (43 -> ~10 us at batch 64)
  const bool regw = (H == 64) && (blockDim.x == 256);
  float wreg[64];
  if (regw) {
#pragma unroll
    for (int q = 0; q < 64; ++q) wreg[q] = w_hh[(long long)threadIdx.x * 64 + q];
  }
  for (int t = 0; t < L; ++t) {
    for (int r = threadIdx.x; r < 4 * H; r += blockDim.x) {
      float acc = b_ih[r];
      for (int d = 0; d < in_dim; ++d) acc = fmaf(xl[((long long)b * L + t) * in_dim + d], w_ih[r * in_dim + d], acc);
      float hh = b_hh[r];
      if (regw) {
#pragma unroll
        for (int q = 0; q < 64; ++q) hh = fmaf(h[q], wreg[q], hh);
      } else {
        const float* wr = w_hh + (long long)r * H;
        for (int q = 0; q < H; ++q) hh = fmaf(h[q], wr[q], hh);
      }
      const float pre = acc + hh;
      gt[r] = (r >= 2 * H && r < 3 * H) ? tanhf(pre) : sigm(pre);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < H; q += blockDim.x) {
      const float cn = gt[H + q] * c[q] + gt[q] * gt[2 * H + q];
      c[q] = cn;
      h[q] = gt[3 * H + q] * tanhf(cn);
      cs[((long long)b * L + t) * H + q] = cn;
      hs[((long long)b * L + t) * H + q] = h[q];
    }
    for (int r = threadIdx.x; r < 4 * H; r += blockDim.x) act[((long long)b * L + t) * 4 * H + r] = gt[r];
    __syncthreads();
  }
}

// BPTT: dh_last (B,H) -> dgates (B,L,4H) (pre-activation gradients)
__global__ __launch_bounds__(256) void lite_lstm_bwd_kernel(const float* __restrict__ dh_last, const float* __restrict__ w_hh,
                                                            const float* __restrict__ act, const float* __restrict__ cs,
                                                            float* __restrict__ dgates, int L, int H, int ld_dh) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // dh[H], dc[H], dg[4H]
  float* dh = sm;
  float* dc = sm + H;
  float* dg = sm + 2 * H;
  float* part4 = sm + 6 * H;          // [4][64] partial sums (H == 64 path)
  const int b = blockIdx.x;
  const bool regw = (H == 64) && (blockDim.x == 256);
  float wreg[64];
  if (regw) {
    const int pw = threadIdx.x >> 6, q = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 64; ++j) wreg[j] = w_hh[(long long)(pw * 64 + j) * 64 + q];      // coalesced over q
  }
  for (int q = threadIdx.x; q < H; q += blockDim.x) {
    dh[q] = dh_last[(long long)b * ld_dh + q];
    dc[q] = 0.f;
  }
  __syncthreads();
  for (int t = L - 1; t >= 0; --t) {
    const float* a = act + ((long long)b * L + t) * 4 * H;
    for (int q = threadIdx.x; q < H; q += blockDim.x) {
      const float ig = a[q], fg = a[H + q], gg = a[2 * H + q], og = a[3 * H + q];
      const float cn = cs[((long long)b * L + t) * H + q];
      const float cp = t > 0 ? cs[((long long)b * L + t - 1) * H + q] : 0.f;
      const float tc = tanhf(cn);
      const float dcur = dc[q] + dh[q] * og * (1.f - tc * tc);
      dg[q] = dcur * gg * ig * (1.f - ig);
      dg[H + q] = dcur * cp * fg * (1.f - fg);
      dg[2 * H + q] = dcur * ig * (1.f - gg * gg);
      dg[3 * H + q] = dh[q] * tc * og * (1.f - og);
      dc[q] = dcur * fg;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < 4 * H; r += blockDim.x) dgates[((long long)b * L + t) * 4 * H + r] = dg[r];
    if (regw) {
      // wave p contracts gate rows [64 p, 64 p + 64) with its register-resident slice of W_hh (broadcast reads of dg),
      // the four partial sums meet in LDS: all 256 threads work instead of 64 (152 -> ~40 us at batch 64)
      const int pw = threadIdx.x >> 6, q = threadIdx.x & 63;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 64; ++j) acc = fmaf(dg[pw * 64 + j], wreg[j], acc);
      part4[pw * 64 + q] = acc;
      __syncthreads();
      if (threadIdx.x < 64) dh[q] = (part4[q] + part4[64 + q]) + (part4[128 + q] + part4[192 + q]);
    } else {
      for (int q = threadIdx.x; q < H; q += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < 4 * H; ++r) acc = fmaf(dg[r], w_hh[(long long)r * H + q], acc);
        dh[q] = acc;
      }
    }
    __syncthreads();
  }
}

// feat[b][0:F] = y2[b][:] (flattened conv features), feat[b][F:F+H] = h_L[b]; dropout (fc.0) on all
__global__ __launch_bounds__(256) void lite_cat_kernel(const float* __restrict__ y2, const float* __restrict__ hs,
                                                       float* __restrict__ feat, int B, int F, int H, int L, int ldf,
                                                       float p_drop, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed = *seed_dev;      // dropout seed in device memory (HIP-graph replays draw new masks)
  const long long total = (long long)B * ldf;
  const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % ldf), b = (int)(i / ldf);
    float v = 0.f;
    if (col < F) v = y2[(long long)b * F + col];
    else if (col < F + H) v = hs[((long long)b * L + (L - 1)) * H + (col - F)];
    if (p_drop > 0.f && col < F + H) {
      uint64_t zz = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)(b * (long long)(F + H) + col) + 1);
      zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
      zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
      zz ^= zz >> 31;
      const float u = (float)(zz >> 40) * (1.0f / 16777216.0f);
      v = u >= p_drop ? v * ks : 0.f;
    }
    feat[i] = v;
  }
}
__global__ __launch_bounds__(256) void lite_uncat_kernel(const float* __restrict__ dfeat, float* __restrict__ dy2,
                                                         float* __restrict__ dh, int B, int F, int H, int ldf, float p_drop,
                                                         uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed = *seed_dev;
  const long long total = (long long)B * (F + H);
  const float ks = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % (F + H)), b = (int)(i / (F + H));
    float v = dfeat[(long long)b * ldf + col];
    if (p_drop > 0.f) {
      uint64_t zz = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)i + 1);
      zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
      zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
      zz ^= zz >> 31;
      const float u = (float)(zz >> 40) * (1.0f / 16777216.0f);
      v = u >= p_drop ? v * ks : 0.f;
    }
    if (col < F) dy2[(long long)b * F + col] = v; else dh[(long long)b * H + (col - F)] = v;
  }
}

static inline unsigned lgrid(long long total) {
  long long g = (total + 255) / 256;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return (unsigned)g;
}

}  // namespace tl
using namespace tl;

extern "C" int tl_lite_conv_fwd(const float* x, const float* w, const float* bias, float* z, float* part, int B, int Cin,
                                int Cout, int T, int k, int pad, void* stream) {
  TL_REQUIRE(x && w && bias && z, "lite_conv_fwd: null pointer");
  TL_REQUIRE(B > 0 && B <= 65535 && Cin > 0 && Cout > 0 && T > 0 && k >= 1 && 2 * pad == k - 1, "lite_conv_fwd: bad sizes (needs 'same' padding)");
  const size_t lds = (size_t)Cin * (LT + k - 1) * 4;
  TL_REQUIRE(lds <= 64 * 1024, "lite_conv_fwd: Cin too large for the LDS tile");
  dim3 grid((T + LT - 1) / LT, B);
  hipLaunchKernelGGL(lite_conv_fwd_kernel, grid, dim3(512), lds, (hipStream_t)stream, x, w, bias, z, part, Cin, Cout, T, k, pad);
  return check_launch("lite_conv_fwd");
}
extern "C" int tl_lite_bn_finalize(const float* part, float* mean, float* rstd, float* run_mean, float* run_var,
                                   int nparts, int C, int64_t count, float momentum, float eps, int training, int64_t* tracked, void* stream) {
  TL_REQUIRE(mean && rstd && run_mean && run_var && C > 0 && (part || !training), "lite_bn_finalize: bad arguments");
  hipLaunchKernelGGL(lite_bn_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, part, mean, rstd,
                     run_mean, run_var, nparts, C, (long long)count, momentum, eps, training, (long long*)tracked);
  return check_launch("lite_bn_finalize");
}
extern "C" int tl_lite_bn_act_pool_fwd(const float* z, const float* mean, const float* rstd, const float* gamma,
                                       const float* beta, float* y, int B, int C, int T, float slope, void* stream) {
  TL_REQUIRE(z && mean && rstd && gamma && beta && y && T >= 2, "lite_bn_act_pool_fwd: bad arguments");
  const long long total = (long long)B * C * (T / 2);
  hipLaunchKernelGGL(lite_bn_act_pool_fwd_kernel, dim3(lgrid(total)), dim3(256), 0, (hipStream_t)stream, z, mean, rstd,
                     gamma, beta, y, total, C, T, slope);
  return check_launch("lite_bn_act_pool_fwd");
}
extern "C" int tl_lite_bn_act_pool_bwd(const float* dy, const float* z, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, float* dz, float* dgamma, float* dbeta,
                                       float* work, int B, int C, int T, float slope, int training, void* stream) {
  TL_REQUIRE(dy && z && mean && rstd && gamma && beta && dz && dgamma && dbeta && work, "lite_bn_act_pool_bwd: null pointer");
  TL_REQUIRE(B > 0 && B <= 65535 && C > 0 && T >= 2, "lite_bn_act_pool_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  float* part = work;                         // [B][C][2]
  hipLaunchKernelGGL(lite_bn_act_pool_bwd_kernel, dim3(C, B), dim3(256), 0, st, dy, z, mean, rstd, gamma, beta, dz, part, C, T, slope);
  hipLaunchKernelGGL(lite_bn_dz_kernel, dim3(C, B), dim3(256), 0, st, dz, z, mean, rstd, gamma, part, dgamma, dbeta, B, C, T,
                     (long long)B * T, training);
  return check_launch("lite_bn_act_pool_bwd");
}
extern "C" int tl_lite_conv_bwd(const float* dz, const float* x, const float* w, float* dx, float* dwpart, float* dbpart,
                                int B, int Cin, int Cout, int T, int k, int pad, void* stream) {
  TL_REQUIRE(dz && x && w && dwpart && dbpart, "lite_conv_bwd: null pointer");
  TL_REQUIRE(B > 0 && B <= 65535 && 2 * pad == k - 1, "lite_conv_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    const size_t lds = (size_t)Cout * (LT + k - 1) * 4;
    TL_REQUIRE(lds <= 64 * 1024, "lite_conv_bwd: Cout too large for the LDS tile");
    hipLaunchKernelGGL(lite_conv_dx_kernel, dim3((T + LT - 1) / LT, B), dim3(512), lds, st, dz, w, dx, Cin, Cout, T, k, pad);
  }
  const int max_o = (255 + Cin * k - 1) / (Cin * k) + 1;              // output channels a block of 256 elements can touch
  const size_t lds_dw = ((size_t)Cin * T + (size_t)max_o * T) * 4;
  const int use_lds = lds_dw <= 64 * 1024;
  hipLaunchKernelGGL(lite_conv_dw_kernel, dim3(B, (Cout * Cin * k + 255) / 256), dim3(256), use_lds ? lds_dw : 0, st, dz, x,
                     dwpart, dbpart, Cin, Cout, T, k, pad, use_lds);
  return check_launch("lite_conv_bwd");
}
extern "C" int tl_lite_lstm_fwd(const float* xl, const float* w_ih, const float* w_hh, const float* b_ih,
                                const float* b_hh, float* act, float* cs, float* hs, int B, int L, int H, int in_dim,
                                void* stream) {
  TL_REQUIRE(xl && w_ih && w_hh && b_ih && b_hh && act && cs && hs && B > 0 && L > 0 && H > 0, "lite_lstm_fwd: bad arguments");
  const size_t lds = (size_t)6 * H * 4;
  TL_REQUIRE(lds <= 64 * 1024, "lite_lstm_fwd: hidden size too large");
  hipLaunchKernelGGL(lite_lstm_fwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, xl, w_ih, w_hh, b_ih, b_hh, act, cs,
                     hs, L, H, in_dim);
  return check_launch("lite_lstm_fwd");
}
extern "C" int tl_lite_lstm_bwd(const float* dh_last, const float* w_hh, const float* act, const float* cs,
                                float* dgates, int B, int L, int H, int ld_dh, void* stream) {
  TL_REQUIRE(dh_last && w_hh && act && cs && dgates && B > 0 && L > 0 && H > 0, "lite_lstm_bwd: bad arguments");
  const size_t lds = (size_t)(6 * H + 256) * 4;
  TL_REQUIRE(lds <= 64 * 1024, "lite_lstm_bwd: hidden size too large");
  hipLaunchKernelGGL(lite_lstm_bwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, dh_last, w_hh, act, cs, dgates, L,
                     H, ld_dh);
  return check_launch("lite_lstm_bwd");
}
extern "C" int tl_lite_cat(const float* y2, const float* hs, float* feat, int B, int F, int H, int L, int ldf,
                           float p_drop, uint64_t seed, void* stream) {
  TL_REQUIRE(y2 && hs && feat && ldf >= F + H && p_drop >= 0.f && p_drop < 1.f, "lite_cat: bad arguments");
  hipLaunchKernelGGL(lite_cat_kernel, dim3(lgrid((long long)B * ldf)), dim3(256), 0, (hipStream_t)stream, y2, hs, feat, B, F,
                     H, L, ldf, p_drop, seed, (const uint64_t*)nullptr);
  return check_launch("lite_cat");
}
extern "C" int tl_lite_uncat(const float* dfeat, float* dy2, float* dh, int B, int F, int H, int ldf, float p_drop,
                             uint64_t seed, void* stream) {
  TL_REQUIRE(dfeat && dy2 && dh && ldf >= F + H && p_drop >= 0.f && p_drop < 1.f, "lite_uncat: bad arguments");
  hipLaunchKernelGGL(lite_uncat_kernel, dim3(lgrid((long long)B * (F + H))), dim3(256), 0, (hipStream_t)stream, dfeat, dy2,
                     dh, B, F, H, ldf, p_drop, seed, (const uint64_t*)nullptr);
  return check_launch("lite_uncat");
}
// the same two with the dropout seed read from device memory (one uint64): the launches can sit in a HIP graph whose
// replays draw a fresh mask each, the caller updating *seed_dev between replays
extern "C" int tl_lite_cat_dev(const float* y2, const float* hs, float* feat, int B, int F, int H, int L, int ldf,
                               float p_drop, const uint64_t* seed_dev, void* stream) {
  TL_REQUIRE(y2 && hs && feat && seed_dev && ldf >= F + H && p_drop >= 0.f && p_drop < 1.f, "lite_cat_dev: bad arguments");
  hipLaunchKernelGGL(lite_cat_kernel, dim3(lgrid((long long)B * ldf)), dim3(256), 0, (hipStream_t)stream, y2, hs, feat, B, F,
                     H, L, ldf, p_drop, (uint64_t)0, seed_dev);
  return check_launch("lite_cat_dev");
}
extern "C" int tl_lite_uncat_dev(const float* dfeat, float* dy2, float* dh, int B, int F, int H, int ldf, float p_drop,
                                 const uint64_t* seed_dev, void* stream) {
  TL_REQUIRE(dfeat && dy2 && dh && seed_dev && ldf >= F + H && p_drop >= 0.f && p_drop < 1.f, "lite_uncat_dev: bad arguments");
  hipLaunchKernelGGL(lite_uncat_kernel, dim3(lgrid((long long)B * (F + H))), dim3(256), 0, (hipStream_t)stream, dfeat, dy2,
                     dh, B, F, H, ldf, p_drop, (uint64_t)0, seed_dev);
  return check_launch("lite_uncat_dev");
}

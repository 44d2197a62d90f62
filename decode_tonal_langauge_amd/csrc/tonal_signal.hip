// preprocess/signal band-extraction kernels (gfx950), all fp64 arithmetic like the reference's
// float64 path (preprocess/signal/frequency_filter.py): Gaussian-bank analytic envelope as an
// exact circular convolution with host-supplied complex taps, zero-phase IIR (filtfilt), causal
// biquad cascade (sosfilt) and the causal FIR bank.
#include "tonal_common.h"

namespace tl {

constexpr int SIG_TB = 1024;   // samples per workgroup (256 threads x 4)
constexpr int SIG_SPT = 4;     // samples per thread

template <typename T>
__device__ __forceinline__ double ld_as_f64(const void* p, long long i) {
  return (double)reinterpret_cast<const T*>(p)[i];
}

// y[c][t] = mean_b | sum_k taps[b][k] * x[c][(t - (k - half)) mod T] |   (or the real part)
// One workgroup = one channel x 1024 samples; the circular window is staged once in LDS as f64;
// the taps are wave-uniform (scalar loads), each thread keeps 4 samples x nb complex sums.
template <typename TIN, int NB>
__global__ __launch_bounds__(256) void gauss_envelope_kernel(const void* __restrict__ x, const double* __restrict__ taps,
                                                             double* __restrict__ y, long long T, int ntap, int half,
                                                             int envelope) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int c = blockIdx.y;
  const long long t0 = (long long)blockIdx.x * SIG_TB;
  const int win = SIG_TB + ntap - 1;
  // xs[i] = x[(t0 + half - (ntap-1) + i) mod T]
  long long base = (t0 + half - (ntap - 1)) % T;
  if (base < 0) base += T;
  for (int i = threadIdx.x; i < win; i += blockDim.x) {
    long long src = base + i;
    src %= T;
    xs[i] = ld_as_f64<TIN>(x, (long long)c * T + src);
  }
  __syncthreads();
  double re[SIG_SPT][NB], im[SIG_SPT][NB];
#pragma unroll
  for (int s = 0; s < SIG_SPT; ++s)
#pragma unroll
    for (int b = 0; b < NB; ++b) re[s][b] = im[s][b] = 0.0;
  // sample lt uses x[t - n] with n = k - half: xs index = lt + (ntap - 1) - k
  for (int k = 0; k < ntap; ++k) {
    double xv[SIG_SPT];
#pragma unroll
    for (int s = 0; s < SIG_SPT; ++s) xv[s] = xs[threadIdx.x + s * 256 + (ntap - 1) - k];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double hr = taps[((long long)b * ntap + k) * 2];
      const double hi = taps[((long long)b * ntap + k) * 2 + 1];
#pragma unroll
      for (int s = 0; s < SIG_SPT; ++s) {
        re[s][b] = fma(hr, xv[s], re[s][b]);
        im[s][b] = fma(hi, xv[s], im[s][b]);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < SIG_SPT; ++s) {
    const long long t = t0 + threadIdx.x + s * 256;
    if (t < T) {
      double acc = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b) acc += envelope ? sqrt(re[s][b] * re[s][b] + im[s][b] * im[s][b]) : re[s][b];
      y[(long long)c * T + t] = acc / NB;
    }
  }
}

// The same bank with the kernel's Hermitian symmetry used (the reference's per-band DFT multiplier is real, so
// h_b[-n] = conj(h_b[n])): y = h[0] x[t] + sum_{n>0} Re h[n] (x[t-n] + x[t+n]) + i Im h[n] (x[t-n] - x[t+n]).  The sum and
// the difference of a sample pair are formed once for all bands: 2 adds + 2 NB FMAs per pair and sample instead of 4 NB
// FMAs (0.56 x the fp64 operations at NB = 8).  taps: (half + 1, NB, 2) = Re / Im of h_b[n], n = 0..half, tap-major: the 16
// coefficients of a tap are one 128-byte run for the scalar loads.
template <typename TIN, int NB>
__global__ __launch_bounds__(256) void gauss_envelope_sym_kernel(const void* __restrict__ x, const double* __restrict__ taps,
                                                                 double* __restrict__ y, long long T, int half, int envelope) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int c = blockIdx.y;
  const long long t0 = (long long)blockIdx.x * SIG_TB;
  const int win = SIG_TB + 2 * half;
  // xs[i] = x[(t0 - half + i) mod T]
  long long base = (t0 - half) % T;
  if (base < 0) base += T;
  for (int i = threadIdx.x; i < win; i += blockDim.x) xs[i] = ld_as_f64<TIN>(x, (long long)c * T + (base + i) % T);
  __syncthreads();
  double re[SIG_SPT][NB], im[SIG_SPT][NB];
  {
    double xv[SIG_SPT];
#pragma unroll
    for (int s = 0; s < SIG_SPT; ++s) xv[s] = xs[threadIdx.x + s * 256 + half];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double hr = taps[b * 2], hi = taps[b * 2 + 1];
#pragma unroll
      for (int s = 0; s < SIG_SPT; ++s) {
        re[s][b] = hr * xv[s];
        im[s][b] = hi * xv[s];
      }
    }
  }
  for (int n = 1; n <= half; ++n) {
    double sm[SIG_SPT], df[SIG_SPT];
#pragma unroll
    for (int s = 0; s < SIG_SPT; ++s) {
      const double xm = xs[threadIdx.x + s * 256 + half - n], xp = xs[threadIdx.x + s * 256 + half + n];
      sm[s] = xm + xp;
      df[s] = xm - xp;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double hr = taps[((long long)n * NB + b) * 2];
      const double hi = taps[((long long)n * NB + b) * 2 + 1];
#pragma unroll
      for (int s = 0; s < SIG_SPT; ++s) {
        re[s][b] = fma(hr, sm[s], re[s][b]);
        im[s][b] = fma(hi, df[s], im[s][b]);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < SIG_SPT; ++s) {
    const long long t = t0 + threadIdx.x + s * 256;
    if (t < T) {
      double acc = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b) acc += envelope ? sqrt(re[s][b] * re[s][b] + im[s][b] * im[s][b]) : re[s][b];
      y[(long long)c * T + t] = acc / NB;
    }
  }
}

// ------------------------------------------------------------------------------------------
// The same bank by overlap-save on an LDS-resident FFT: a workgroup (256 threads) takes one channel and one segment of
// 1024 - 2 half output samples.  The transform is a 1024-point Stockham radix-4 (five stages); a thread owns the four
// positions tid + 256 r of the natural order, which are exactly the inputs of its first-stage butterfly and the outputs of
// its last-stage one - so a transform takes its input from registers and leaves its output in registers, with four LDS
// round trips (and barriers) between; the thread's twelve twiddles are loaded once.  Per segment: the window straight from
// memory -> forward transform -> the spectrum stays in registers; per band: x the band's kernel spectrum (host: FFT_1024
// of the same truncated taps the time-domain kernels convolve with, / 1024) -> inverse transform -> |.| accumulated.
// 2.4 x fewer fp64 operations per output than the symmetric time-domain kernel; four workgroups per CU (34 KB of LDS).
// Index padding d + (d >> 5) keeps the strided stores of the first stages off the same banks.
// ------------------------------------------------------------------------------------------
constexpr int OLS_N = 1024, OLS_Q = OLS_N / 4, OLS_PAD = OLS_N + (OLS_N >> 5);
__device__ __forceinline__ int ols_idx(int d) { return d + (d >> 5); }
typedef double ols_d2 __attribute__((ext_vector_type(2)));

// sqrt for the magnitudes: v_rsq_f64 (2^-23) and one Newton step on the residual x - y^2 (fused): ~2^-45 relative, far inside
// the 1e-9 the golden holds; the library sqrt spends three times the instructions on the last bits and on denormal scaling,
// and 32 magnitudes per thread were a third of the kernel's vector instructions
// (rsq(0) = inf is clamped to 2^500, so v = 0 gives exactly 0 without a compare and two selects; v below 1e-300 comes out
// wrong in relative terms and right to 1e-150 in absolute ones)
__device__ __forceinline__ double ols_sqrt(double v) {
  const double r = __builtin_fmin(__builtin_amdgcn_rsq(v), 0x1p500);
  double yv = v * r;
  const double e = fma(-yv, yv, v);
  yv = fma(e, 0.5 * r, yv);
  return yv;
}

template <bool INV, typename R = double>
__device__ __forceinline__ void ols_bfly(const R (&xr)[4], const R (&xi)[4], R (&yr)[4], R (&yi)[4]) {
  const R t0r = xr[0] + xr[2], t0i = xi[0] + xi[2], t1r = xr[0] - xr[2], t1i = xi[0] - xi[2];
  const R t2r = xr[1] + xr[3], t2i = xi[1] + xi[3];
  const R dr = xr[1] - xr[3], di = xi[1] - xi[3];
  const R t3r = INV ? -di : di, t3i = INV ? dr : -dr;    // (a1 - a3) * (-i) forward, (+i) inverse
  yr[0] = t0r + t2r; yi[0] = t0i + t2i;
  yr[1] = t1r + t3r; yi[1] = t1i + t3i;
  yr[2] = t0r - t2r; yi[2] = t0i - t2i;
  yr[3] = t1r - t3r; yi[3] = t1i - t3i;
}
// v: the thread's values at positions tid + 256 r, in and out.  wc / ws: its twiddles (cos, -sin) of stages 1..4.
// LDS layouts, one per exchange (a layout only has to agree between a stage's stores and the next stage's loads): the
// outputs of stages 0 and 1 are stored digit-major - element d at (digit of d the store's r runs over) x 264 + (the rest of
// d) = r x 264 + tid - so a ds_write_b64's 16-lane groups and the next stage's 32-lane read groups (which then start at
// (tid & 3) x 264 + ..: +8 banks per digit) each touch every bank once; stages 2 and 3 are conflict-free in natural order.
// (The padded natural order used before, d + (d >> 5), stored stage 0 two-way and stage 1 four-way conflicted.)
constexpr int OLS_DS = OLS_Q + 8;                            // digit stride; 4 OLS_DS = OLS_PAD
template <bool INV>
__device__ __forceinline__ void ols_fft(double (&vr)[4], double (&vi)[4], double* __restrict__ lds, const ols_d2* __restrict__ twl,
                                        const double (&wc)[2][3], const double (&ws)[2][3], int tid) {
  double* bre[2] = {lds, lds + 2 * OLS_PAD};
  double* bim[2] = {lds + OLS_PAD, lds + 3 * OLS_PAD};
  double yr[4], yi[4];
  ols_bfly<INV>(vr, vi, yr, yi);                            // stage 0 (Ns = 1, no twiddles): outputs 4 tid + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bre[0][tid + r * OLS_DS] = yr[r];
    bim[0][tid + r * OLS_DS] = yi[r];
  }
  __syncthreads();
#pragma unroll
  for (int st = 1; st < 5; ++st) {
    const int src = (st - 1) & 1, dst = st & 1, Ns = 1 << (2 * st);
    // element tid + 256 r of the previous stage's output
    const int rd = st == 1 ? (tid & 3) * OLS_DS + (tid >> 2) : st == 2 ? ((tid >> 2) & 3) * OLS_DS + 4 * (tid >> 4) + (tid & 3) : tid;
    const int rds = st < 3 ? OLS_Q / 4 : OLS_Q;
    double xr[4], xi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xr[r] = bre[src][rd + r * rds];
      xi[r] = bim[src][rd + r * rds];
    }
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      // stages 1, 2 have 4 / 16 distinct twiddle sets: a 60-entry LDS table; stages 3, 4: the thread's own, in registers
      double c, sn;
      if (st < 3) {
        const ols_d2 w = twl[(st == 1 ? 0 : 12) + (tid & (Ns - 1)) * 3 + r - 1];
        c = w[0];
        sn = INV ? -w[1] : w[1];
      } else {
        c = wc[st - 3][r - 1];
        sn = INV ? -ws[st - 3][r - 1] : ws[st - 3][r - 1];
      }
      const double tr = fma(-xi[r], sn, xr[r] * c), ti = fma(xr[r], sn, xi[r] * c);
      xr[r] = tr;
      xi[r] = ti;
    }
    if (st < 4) {
      ols_bfly<INV>(xr, xi, yr, yi);
      // outputs ((tid >> 2 st) << (2 st + 2)) + (tid & (Ns - 1)) + r Ns
      const int wr = st == 1 ? tid : ((tid >> (2 * st)) << (2 * st + 2)) + (tid & (Ns - 1));
      const int wrs = st == 1 ? OLS_DS : Ns;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bre[dst][wr + r * wrs] = yr[r];
        bim[dst][wr + r * wrs] = yi[r];
      }
      __syncthreads();
    } else {
      ols_bfly<INV>(xr, xi, vr, vi);                        // last stage (Ns = 256): outputs tid + 256 r = the thread's own
    }
  }
}

// lead: the window starts at t0 - lead; skip: its first skip outputs are the wrapped ones (dropped); circular: indices wrap
// over the recording (the Gaussian bank) or samples before the start are zero (the causal FIR bank).
// NBT: compile-time band count (0: the run-time argument nb)
template <typename TIN, typename TOUT, int NBT>
__global__ __launch_bounds__(OLS_Q) void ols_bank_kernel(const void* __restrict__ x, const ols_d2* __restrict__ G,
                                                         const ols_d2* __restrict__ tw, TOUT* __restrict__ y, long long T, int nb,
                                                         int lead, int skip, int circular, int envelope) {
  const int NB = NBT ? NBT : nb;
  __shared__ __attribute__((aligned(16))) double lds[4 * OLS_PAD];
  const int tid = threadIdx.x, c = blockIdx.y;
  const int Lv = OLS_N - skip;
  const long long t0 = (long long)blockIdx.x * Lv;
  long long base = t0 - lead;
  if (circular) {
    base %= T;
    if (base < 0) base += T;
  }
  // W_{4 Ns}^{r k} = W_N^{r k N / (4 Ns)}, k = tid % Ns
  __shared__ __attribute__((aligned(16))) ols_d2 twl[60];   // stages 1 (k < 4) and 2 (k < 16), [k][r - 1]
  if (tid < 60) {
    const int st = tid < 12 ? 1 : 2, e = tid < 12 ? tid : tid - 12;
    twl[tid] = tw[(e % 3 + 1) * (e / 3) * (OLS_Q >> (2 * st))];
  }
  double wc[2][3], ws[2][3];
#pragma unroll
  for (int st = 3; st < 5; ++st) {
    const int step = (tid & ((1 << (2 * st)) - 1)) * (OLS_Q >> (2 * st));
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      const ols_d2 w = tw[r * step];
      wc[st - 3][r - 1] = w[0];
      ws[st - 3][r - 1] = w[1];
    }
  }
  double vr[4], vi[4], Xr[4], Xi[4], acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long ti = base + tid + r * OLS_Q;
    // circular: base < T; one conditional subtraction wraps it when T >= 1024 (the 64-bit % was a third of the kernel's
    // instructions); shorter recordings keep the division
    const long long tw_ = T >= OLS_N ? (ti >= T ? ti - T : ti) : ti % T;
    vr[r] = circular ? ld_as_f64<TIN>(x, (long long)c * T + tw_)
                     : ((ti >= 0 && ti < T) ? ld_as_f64<TIN>(x, (long long)c * T + ti) : 0.0);
    vi[r] = 0.0;
    acc[r] = 0.0;
  }
  ols_fft<false>(vr, vi, lds, twl, wc, ws, tid);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    Xr[r] = vr[r];
    Xi[r] = vi[r];
  }
  ols_d2 g[4], gn[4];                                       // this band's kernel spectrum, the next band's (in flight)
#pragma unroll
  for (int r = 0; r < 4; ++r) g[r] = G[tid + r * OLS_Q];
  for (int b = 0; b < NB; ++b) {
    const int bn = b + 1 < NB ? b + 1 : b;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gn[r] = G[(long long)bn * OLS_N + tid + r * OLS_Q];
      vr[r] = fma(-Xi[r], g[r][1], Xr[r] * g[r][0]);
      vi[r] = fma(Xr[r], g[r][1], Xi[r] * g[r][0]);
    }
    ols_fft<true>(vr, vi, lds, twl, wc, ws, tid);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[r] += envelope ? ols_sqrt(fma(vr[r], vr[r], vi[r] * vi[r])) : vr[r];
      g[r] = gn[r];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = tid + r * OLS_Q;
    const long long t = t0 + i - skip;
    if (i >= skip && t < T) y[(long long)c * T + t] = (TOUT)(acc[r] / NB);
  }
}

// ------------------------------------------------------------------------------------------
// Band-limited form of the same overlap-save bank.  A Gaussian band's kernel spectrum is non-zero (above 1e-12 of its peak:
// the caller checks) on fewer than 256 of the 1024 bins, k0 <= k < k0 + 256, so its inverse transform is
//   z[4 m + r] = W^{-k0 n} . IDFT_256( Z[k0 + k'] W_1024^{-k' r} )[m]
// : four independent 256-point transforms, one per residue r - ONE WAVE EACH (lane l owns k' = l + 64 j going in and
// m = l + 64 j coming out), with the residue's twiddle folded into the kernel spectrum on the host (Gp[b][r][k']).  A
// 256-point radix-4 Stockham has four stages = three exchanges, and they are private to the wave: no workgroup barrier in
// the band loop, 4/5 of the butterflies, 3/4 of the LDS round trips of the 1024-point inverse.  The forward transform runs
// the same way round (radix-4 over the thread's four samples, one workgroup-wide exchange, then a wave-private 256-point
// transform per residue of the BIN index); its result goes to LDS once (16 KB) and every wave reads its window of it per
// band.  |z| does not see the modulation W^{-k0 n}; the real part (envelope = 0) multiplies it back from the twiddle table.
// ------------------------------------------------------------------------------------------
constexpr int OLS_WQ = 256;
__device__ __forceinline__ void ols_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// v: the lane's values at l + 64 j, in and out; buf: the wave's own OLS_WPL (re, im) pairs; tc / ts: (cos, -sin) of stages
// 1..3.  16-byte accesses (two 8-byte planes get merged by the compiler into ds_read2_b64, which moves half the bytes per
// clock of ds_read_b64 / ds_read_b128).  Layouts per exchange as in ols_fft: digit-major after stages 0 and 1 - digit stride
// 68 resp. 72 elements, which is what keeps the ds_read_b128 lane groups {0-3, 12-15, 20-27}, .. of the NEXT stage on
// sixteen different 16-byte slots - natural order after stage 2.
// The last exchange does not go through LDS at all: stage 3 wants, in register r of lane (h, lo) [h = l >> 4], what stage 2
// left in register h of lane (r, lo) - a 4 x 4 transpose between the register index and the lane's top two bits, which is
// v_permlane32_swap (register bit 1 <-> lane bit 5) followed by v_permlane16_swap (register bit 0 <-> lane bit 4): sixteen
// one-pass instructions for the four complex doubles instead of four 16-byte stores, a wait and four 16-byte loads.
constexpr int OLS_WS0 = 64 + 4, OLS_WS1 = 64 + 8, OLS_WPL = 4 * OLS_WS1, OLS_XS = OLS_Q + 4;
template <bool S32>
__device__ __forceinline__ void ols_lane_swap(double& a, double& b) {
  unsigned a0 = (unsigned)__double2loint(a), a1 = (unsigned)__double2hiint(a);
  unsigned b0 = (unsigned)__double2loint(b), b1 = (unsigned)__double2hiint(b);
  const auto r0 = S32 ? __builtin_amdgcn_permlane32_swap(a0, b0, false, false) : __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
  const auto r1 = S32 ? __builtin_amdgcn_permlane32_swap(a1, b1, false, false) : __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
  a = __hiloint2double((int)r1[0], (int)r0[0]);
  b = __hiloint2double((int)r1[1], (int)r0[1]);
}
template <bool S32>
__device__ __forceinline__ void ols_lane_swap(float& a, float& b) {
  const unsigned a0 = __float_as_uint(a), b0 = __float_as_uint(b);
  const auto r0 = S32 ? __builtin_amdgcn_permlane32_swap(a0, b0, false, false) : __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
  a = __uint_as_float(r0[0]);
  b = __uint_as_float(r0[1]);
}
template <typename R>
__device__ __forceinline__ void ols_lane_transpose(R (&v)[4]) {
  ols_lane_swap<true>(v[0], v[2]);
  ols_lane_swap<true>(v[1], v[3]);
  ols_lane_swap<false>(v[0], v[1]);
  ols_lane_swap<false>(v[2], v[3]);
}
template <bool INV, typename R>
__device__ __forceinline__ void ols_twiddle(R (&xr)[4], R (&xi)[4], const R (&tc)[3], const R (&ts)[3]) {
#pragma unroll
  for (int r = 1; r < 4; ++r) {
    const R c = tc[r - 1], sn = INV ? -ts[r - 1] : ts[r - 1];
    const R tr = fma(-xi[r], sn, xr[r] * c), ti = fma(xr[r], sn, xi[r] * c);
    xr[r] = tr;
    xi[r] = ti;
  }
}
// (re, im) pair of the transform's working precision: fp64 for float64 recordings; fp32 for float32 ones - the reference's
// own arithmetic there (scipy.fft keeps single precision: complex64, preprocess/signal/frequency_filter.py:167-181)
template <typename R>
using ols_r2 = R __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ols_sqrt(float v) { return __builtin_sqrtf(v); }
template <bool INV, typename R>
__device__ __forceinline__ void ols_fft256_wave(R (&vr)[4], R (&vi)[4], ols_r2<R>* __restrict__ buf, const R (&tc)[3][3],
                                                const R (&ts)[3][3], int l) {
  using ols_d2 = ols_r2<R>;
  R xr[4], xi[4], yr[4], yi[4];
  ols_bfly<INV>(vr, vi, yr, yi);                              // stage 0: outputs 4 l + r, stored digit-major (stride 68)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ols_d2 t;
    t[0] = yr[r];
    t[1] = yi[r];
    buf[l + r * OLS_WS0] = t;
  }
  ols_wave_sync();
  const int rd0 = (l & 3) * OLS_WS0 + (l >> 2);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const ols_d2 t = buf[rd0 + r * 16];
    xr[r] = t[0];
    xi[r] = t[1];
  }
  ols_twiddle<INV>(xr, xi, tc[0], ts[0]);
  ols_bfly<INV>(xr, xi, yr, yi);                              // stage 1: outputs 16 (l >> 2) + (l & 3) + 4 r, digit-major (72)
  ols_wave_sync();                                            // (the LDS queue of a wave is in order: reads above, then writes)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ols_d2 t;
    t[0] = yr[r];
    t[1] = yi[r];
    buf[l + r * OLS_WS1] = t;
  }
  ols_wave_sync();
  const int rd1 = ((l >> 2) & 3) * OLS_WS1 + 4 * (l >> 4) + (l & 3);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const ols_d2 t = buf[rd1 + r * 16];
    xr[r] = t[0];
    xi[r] = t[1];
  }
  ols_twiddle<INV>(xr, xi, tc[1], ts[1]);
  ols_bfly<INV>(xr, xi, yr, yi);                              // stage 2: outputs 64 (l >> 4) + (l & 15) + 16 r
  ols_lane_transpose(yr);                                     // -> element l + 64 r in register r
  ols_lane_transpose(yi);
  ols_twiddle<INV>(yr, yi, tc[2], ts[2]);
  ols_bfly<INV>(yr, yi, vr, vi);                              // stage 3 (Ns = 64): outputs l + 64 r = the lane's own
  ols_wave_sync();                                            // the next transform's stores stay behind this one's loads
}

// R: working precision (see ols_r2).  The kernel-spectrum and twiddle tables stay fp64 in memory and are rounded on load.
template <typename TIN, int NBT, typename R>
__global__ __launch_bounds__(OLS_Q) void ols_bank_bl_kernel(const void* __restrict__ x, const ols_d2* __restrict__ Gp,
                                                            const int* __restrict__ k0s, const ols_d2* __restrict__ tw,
                                                            double* __restrict__ y, long long T, int nb, int lead, int skip,
                                                            int envelope) {
  const int NB = NBT ? NBT : nb;
  // region A: the forward transform's exchange [q][t], then the spectrum, bin 4 k + q at q x OLS_XS + k; region B: the waves' planes
  using r2 = ols_r2<R>;
  __shared__ __attribute__((aligned(16))) r2 ldsA[4 * OLS_XS];
  __shared__ __attribute__((aligned(16))) r2 ldsB[4 * OLS_WPL];
  const int tid = threadIdx.x, c = blockIdx.y;
  const int w = tid >> 6, l = tid & 63;
  const int Lv = OLS_N - skip;
  const long long t0 = (long long)blockIdx.x * Lv;
  long long base = t0 - lead;                                 // in (-T, 2 T): T >= 1024 >= lead, t0 < T + 1024
  if (base < 0) base += T;
  if (base >= T) base -= T;
  R vr[4], vi[4], acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    long long ti = base + tid + r * OLS_Q;                    // < 2 T (the host requires T >= 1024): no 64-bit division
    if (ti >= T) ti -= T;
    vr[r] = (R)ld_as_f64<TIN>(x, (long long)c * T + ti);
    vi[r] = (R)0;
    acc[r] = (R)0;
  }
  r2* wbuf = ldsB + w * OLS_WPL;
  R tc[3][3], ts[3][3];                                  // W_{4 Ns}^{r k}, k = l % Ns, as W_1024^{r k 256 / Ns}
#pragma unroll
  for (int st = 1; st < 4; ++st) {
    const int step = (l & ((1 << (2 * st)) - 1)) * (OLS_WQ >> (2 * st));
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      const ols_d2 t = tw[r * step];
      tc[st - 1][r - 1] = (R)t[0];
      ts[st - 1][r - 1] = (R)t[1];
    }
  }
  // Forward transform, the same way round: X[4 k + q] = DFT_256_t( W_1024^{t q} . sum_r x[t + 256 r] W_4^{r q} )[k] - the
  // radix-4 over r and the twiddle in the thread's registers (t = tid), ONE workgroup-wide exchange that hands residue q to
  // wave q (lane l takes t = l + 64 j), then the wave-private 256-point transform.  Three barriers up to the band loop
  // instead of the five of the 1024-point Stockham (ols_fft).
  {
    R yr[4], yi[4], c3[3], s3[3];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      const ols_d2 t = tw[tid * q];
      c3[q - 1] = (R)t[0];
      s3[q - 1] = (R)t[1];
    }
    ols_bfly<false>(vr, vi, yr, yi);
    ols_twiddle<false>(yr, yi, c3, s3);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      r2 t;
      t[0] = yr[q];
      t[1] = yi[q];
      ldsA[q * OLS_Q + tid] = t;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const r2 t = ldsA[w * OLS_Q + l + 64 * j];
      vr[j] = t[0];
      vi[j] = t[1];
    }
    __syncthreads();                                          // region A is read: the spectrum may overwrite it
    ols_fft256_wave<false>(vr, vi, wbuf, tc, ts, l);
#pragma unroll
    for (int j = 0; j < 4; ++j) {                             // bin 4 (l + 64 j) + w
      r2 t;
      t[0] = vr[j];
      t[1] = vi[j];
      ldsA[w * OLS_XS + l + 64 * j] = t;
    }
  }
  const r2* Xs = ldsA;
  auto ldg = [&](long long i) -> r2 {
    const ols_d2 t = Gp[i];
    r2 o;
    o[0] = (R)t[0];
    o[1] = (R)t[1];
    return o;
  };
  r2 g[4], gn[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) g[r] = ldg(w * OLS_WQ + l + 64 * r);
  __syncthreads();
  for (int b = 0; b < NB; ++b) {
    const int bn = b + 1 < NB ? b + 1 : b;
    const int k0 = k0s[b];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gn[r] = ldg((long long)(bn * 4 + w) * OLS_WQ + l + 64 * r);
      const int bin = (k0 + l + 64 * r) & (OLS_N - 1);
      const r2 X = Xs[(bin & 3) * OLS_XS + (bin >> 2)];
      vr[r] = fma(-X[1], g[r][1], X[0] * g[r][0]);
      vi[r] = fma(X[0], g[r][1], X[1] * g[r][0]);
    }
    ols_fft256_wave<true>(vr, vi, wbuf, tc, ts, l);
    if (envelope) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += ols_sqrt(fma(vr[r], vr[r], vi[r] * vi[r]));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {                           // Re(z W^{-k0 n}), n = 4 (l + 64 r) + w
        const ols_d2 e = tw[(k0 * (4 * (l + 64 * r) + w)) & (OLS_N - 1)];
        acc[r] += fma(vi[r], (R)e[1], vr[r] * (R)e[0]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) g[r] = gn[r];
    ols_wave_sync();
  }
  __syncthreads();                                            // every wave is done with the spectrum: its region takes the outputs
  R* lds = reinterpret_cast<R*>(ldsA);
#pragma unroll
  for (int r = 0; r < 4; ++r) lds[l + 72 * w + 288 * r] = acc[r];   // sample 4 l + w + 256 r, residue-major
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = tid + r * OLS_Q;
    const long long t = t0 + i - skip;
    if (i >= skip && t < T) y[(long long)c * T + t] = (double)(lds[(tid >> 2) + 72 * (tid & 3) + 288 * r] / (R)NB);
  }
}

// generic band count (slow path): loops bands outermost, one band at a time
template <typename TIN>
__global__ __launch_bounds__(256) void gauss_envelope_generic_kernel(const void* __restrict__ x,
                                                                     const double* __restrict__ taps,
                                                                     double* __restrict__ y, long long T, int nb, int ntap,
                                                                     int half, int envelope) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int c = blockIdx.y;
  const long long t0 = (long long)blockIdx.x * SIG_TB;
  const int win = SIG_TB + ntap - 1;
  long long base = (t0 + half - (ntap - 1)) % T;
  if (base < 0) base += T;
  for (int i = threadIdx.x; i < win; i += blockDim.x) xs[i] = ld_as_f64<TIN>(x, (long long)c * T + (base + i) % T);
  __syncthreads();
  double acc[SIG_SPT];
#pragma unroll
  for (int s = 0; s < SIG_SPT; ++s) acc[s] = 0.0;
  for (int b = 0; b < nb; ++b) {
    double re[SIG_SPT], im[SIG_SPT];
#pragma unroll
    for (int s = 0; s < SIG_SPT; ++s) re[s] = im[s] = 0.0;
    for (int k = 0; k < ntap; ++k) {
      const double hr = taps[((long long)b * ntap + k) * 2];
      const double hi = taps[((long long)b * ntap + k) * 2 + 1];
#pragma unroll
      for (int s = 0; s < SIG_SPT; ++s) {
        const double xv = xs[threadIdx.x + s * 256 + (ntap - 1) - k];
        re[s] = fma(hr, xv, re[s]);
        im[s] = fma(hi, xv, im[s]);
      }
    }
#pragma unroll
    for (int s = 0; s < SIG_SPT; ++s) acc[s] += envelope ? sqrt(re[s] * re[s] + im[s] * im[s]) : re[s];
  }
#pragma unroll
  for (int s = 0; s < SIG_SPT; ++s) {
    const long long t = t0 + threadIdx.x + s * 256;
    if (t < T) y[(long long)c * T + t] = acc[s] / nb;
  }
}

// ------------------------------------------------------------------------------------------
// IIR: direct-form II transposed, one lane per channel (the recurrence is sequential in time;
// parallelism is across channels).  fp64 is mandatory: the order-4 band-pass has poles at
// |p| = 0.998 and an fp32 recurrence diverges.
// ------------------------------------------------------------------------------------------
constexpr int MAX_TAPS = 17;
#ifndef FF_LANES8
#define FF_LANES8 1       // filtfilt with ntaps <= 9: the state over 8 lanes per channel (0: one lane per channel)
#endif

template <typename TIN>
__device__ __forceinline__ double ext_sample(const void* x, long long cbase, long long T, int edge, long long i) {
  // odd extension of scipy.signal.filtfilt (padtype='odd'); same roundings (2*x0 - x[k])
  if (i < edge) return 2.0 * ld_as_f64<TIN>(x, cbase) - ld_as_f64<TIN>(x, cbase + (edge - i));
  if (i >= edge + T) return 2.0 * ld_as_f64<TIN>(x, cbase + T - 1) - ld_as_f64<TIN>(x, cbase + T - 2 - (i - edge - T));
  return ld_as_f64<TIN>(x, cbase + (i - edge));
}

// filtfilt in three launches: (1) build the odd-extended signal TIME-MAJOR in `work` ([next][C]) so
// that in the sequential pass the 64 lanes of a wave (= 64 channels) read one 512-B line per time
// step; (2) forward and backward recurrences in place, one lane per channel, 8 samples of loads in
// flight; (3) write the centre part back channel-major.
template <typename TIN>
__global__ __launch_bounds__(256) void filtfilt_build_kernel(const void* __restrict__ x, double* __restrict__ work, int C,
                                                             long long T, int edge) {
  const long long next = T + 2LL * edge;
  const long long total = next * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long ti = i / C;
    work[i] = ext_sample<TIN>(x, (long long)c * T, T, edge, ti);
  }
}

template <int NT>
__global__ __launch_bounds__(64) void filtfilt_iir_kernel(const double* __restrict__ b, const double* __restrict__ a,
                                                          const double* __restrict__ zi, double* __restrict__ work, int C,
                                                          long long next, int ntaps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double bb[NT], aa[NT], z[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    bb[k] = k < ntaps ? b[k] : 0.0;
    aa[k] = k < ntaps ? a[k] : 0.0;
  }
  constexpr int U = 8;
  // pass 0: work[0] (extended input) -> work[1]; pass 1: work[1] reversed -> work[0].
  // Separate in/out buffers let the next chunk's loads fly while this chunk's recurrence runs.
  for (int pass = 0; pass < 2; ++pass) {
    const double* __restrict__ in = work + (long long)pass * next * C + c;
    double* __restrict__ out = work + (long long)(1 - pass) * next * C + c;
    const long long first = pass == 0 ? 0 : next - 1;
    const long long dir = pass == 0 ? 1 : -1;
    const double x0 = in[first * C];
#pragma unroll
    for (int k = 0; k < NT; ++k) z[k] = (k < ntaps - 1) ? zi[k] * x0 : 0.0;
    double cur[U], nxt[U];
    long long n = 0;
    if (next >= U) {
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = in[(first + dir * u) * C];
    }
    for (; n + U <= next; n += U) {
      const bool more = n + 2 * U <= next;
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = in[(first + dir * (n + U + u)) * C];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // scipy lfilter order: y = z0 + b0*x;  z_k = (z_{k+1} + x*b_{k+1}) - y*a_{k+1}
        const double yv = z[0] + bb[0] * cur[u];
#pragma unroll
        for (int k = 0; k < NT - 1; ++k) z[k] = (z[k + 1] + cur[u] * bb[k + 1]) - yv * aa[k + 1];
        out[(first + dir * (n + u)) * C] = yv;
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
      }
    }
    for (; n < next; ++n) {
      const double xs = in[(first + dir * n) * C];
      const double yv = z[0] + bb[0] * xs;
#pragma unroll
      for (int k = 0; k < NT - 1; ++k) z[k] = (z[k + 1] + xs * bb[k + 1]) - yv * aa[k + 1];
      out[(first + dir * n) * C] = yv;
    }
  }
}

// The same recurrence with the STATE spread over lanes (ntaps <= 9): a channel owns a 16-lane DPP row, lane k < 8 of it
// holds z_k (lanes 8..15 carry zero coefficients and stay zero), a wave 4 channels.  One lane per channel leaves 4
// wavefronts on the chip for 256 channels, each issuing the ~34 fp64 operations of a sample back to back (214 cycles per
// sample; the stream is bound by instruction issue, ~7 cycles per dependent fp64 / DPP instruction: interleaving a second
// channel set per wave doubled the time).  Here a sample is 9 instructions: the output y = z_0 + b_0 x in lane 0, its
// broadcast over the row (one v_mov_b64_dpp row_newbcast), the shift z_{k+1} -> lane k (one DPP move per 32-bit half), and
// ONE state update per lane - the same operations in the same order on every element as scipy's loop (the file is
// compiled without FMA contraction), so the result stays bit-identical to it.  Parallel in time it cannot be: the
// reference's (b, a) form is a rounding trajectory, not a well-conditioned function.
__device__ __forceinline__ double dpp_bcast_row(double v) {                  // lane 0 of each 16-lane row to the row
  const long long x = __builtin_bit_cast(long long, v);
  const long long y = __builtin_amdgcn_update_dpp(x, x, 0x150, 0xf, 0xf, false);   // v_mov_b64_dpp row_newbcast:0
  return __builtin_bit_cast(double, y);
}
__device__ __forceinline__ double dpp_shl1(double v) {                       // lane i <- lane i + 1 (0 past the row)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x101, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x101, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(64) void filtfilt_iir8_kernel(const double* __restrict__ b, const double* __restrict__ a,
                                                           const double* __restrict__ zi, double* __restrict__ work, int C,
                                                           long long next, int ntaps) {
  const int lane = threadIdx.x, k = lane & 15;
  const int ch = blockIdx.x * 4 + (lane >> 4);
  const bool live = ch < C;
  const int chc = live ? ch : C - 1;
  const double b0 = b[0];
  const double bk = k + 1 < ntaps ? b[k + 1] : 0.0, ak = k + 1 < ntaps ? a[k + 1] : 0.0;
  const double zik = k < ntaps - 1 ? zi[k] : 0.0;
  constexpr int U = 8;                                      // = lanes per channel: lane k stores sample k of a chunk
  for (int pass = 0; pass < 2; ++pass) {
    const double* __restrict__ in = work + (long long)pass * next * C + chc;
    double* __restrict__ out = work + (long long)(1 - pass) * next * C + chc;
    const long long first = pass == 0 ? 0 : next - 1;
    const long long dir = pass == 0 ? 1 : -1;
    double z = zik * in[first * C];
    double cur[U], nxt[U];
    long long n = 0;
    if (next >= U) {
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = in[(first + dir * u) * C];
    }
    for (; n + U <= next; n += U) {
      const bool more = n + 2 * U <= next;
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = in[(first + dir * (n + U + u)) * C];
      }
      double mine = 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // scipy lfilter order: y = z0 + b0*x;  z_k = (z_{k+1} + x*b_{k+1}) - y*a_{k+1}
        const double yv = dpp_bcast_row(z + b0 * cur[u]);
        const double zs = dpp_shl1(z);                      // (lanes 8..15 of the row hold zeros: z_8 = 0 comes for free)
        z = (zs + cur[u] * bk) - yv * ak;
        mine = k == u ? yv : mine;
      }
      if (live && k < U) out[(first + dir * (n + k)) * C] = mine;
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
      }
    }
    for (; n < next; ++n) {
      const double xs = in[(first + dir * n) * C];
      const double yv = dpp_bcast_row(z + b0 * xs);
      const double zs = dpp_shl1(z);
      z = (zs + xs * bk) - yv * ak;
      if (live && k == 0) out[(first + dir * n) * C] = yv;
    }
  }
}

// centre part of the time-major work buffer back to channel-major: 64 x 64 tiles through LDS (both sides coalesced)
__global__ __launch_bounds__(256) void filtfilt_out_kernel(const double* __restrict__ work, double* __restrict__ y, int C,
                                                           long long T, int edge) {
  __shared__ double tile[64][65];
  const long long t0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;       // 64 x 4
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long long t = t0 + ly + 4 * r;
    const int c = c0 + lx;
    if (t < T && c < C) tile[ly + 4 * r][lx] = work[(t + edge) * C + c];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = c0 + ly + 4 * r;
    const long long t = t0 + lx;
    if (t < T && c < C) y[(long long)c * T + t] = tile[lx][ly + 4 * r];
  }
}

// ------------------------------------------------------------------------------------------
// filtfilt, TIME-PARALLEL (opt-in: TONAL_KERNELS=butter=scan).  The recurrence is linear, so a pass over the extended signal
// splits exactly into blocks of L samples: (1) every (channel, block) runs the recurrence from a ZERO state and keeps the
// state it ends in, s_j; (2) the states at the block starts follow z_{j+1} = A^L z_j + s_j - a prefix "sum" over the blocks,
// evaluated per channel by a Hillis-Steele scan with the host-supplied matrices A^(L 2^m) (one workgroup per channel, a
// thread per (block, state row), log2 steps); (3) every (channel, block) re-runs the recurrence from its true start state and writes the
// outputs.  Steps (1) and (3) are scipy's loop (same operations, same order); step (2) is where the arithmetic differs from
// the sequential kernel: the direct-form states of this filter are ~1e6 x the output with cancellation (eight poles
// clustered at z = 1), so the matrices come as double-double pairs (exact powers rounded once, from the host) and every
// matrix-vector product is a compensated dot product (two_prod / two_sum: the result is the correctly rounded exact value up
// to ~1e-31 of the terms) - in plain fp64 the same scheme lands 6e-3 from the reference (round 2).  What is left against
// the sequential kernel is 2e-8 - 5e-8 relative, the size of the reference's OWN rounding: scipy's loop differs from the
// same loop in 64-bit-mantissa arithmetic by 1e-8 - 3e-8 (tests/test_signal_scan_notes.py), so no re-association can
// promise 1e-9.  Hence opt-in; the default stays the bit-exact sequential kernel.  ntaps <= 9.
// ------------------------------------------------------------------------------------------
constexpr int FS_NS = 8;

template <bool APPLY>
__global__ __launch_bounds__(64) void filtfilt_scan_block_kernel(const double* __restrict__ b, const double* __restrict__ a,
                                                                 const double* __restrict__ in, double* __restrict__ out,
                                                                 const double* __restrict__ Z, double* __restrict__ S, int C,
                                                                 long long next, int ntaps, int L, long long first,
                                                                 long long dir) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const long long j = blockIdx.y;
  const long long n0 = j * L, n1 = (n0 + L < next) ? n0 + L : next;
  double bb[FS_NS + 1], aa[FS_NS + 1], z[FS_NS + 1];
#pragma unroll
  for (int k = 0; k <= FS_NS; ++k) {
    bb[k] = k < ntaps ? b[k] : 0.0;
    aa[k] = k < ntaps ? a[k] : 0.0;
    z[k] = 0.0;
  }
  if constexpr (APPLY) {
#pragma unroll
    for (int k = 0; k < FS_NS; ++k) z[k] = Z[(j * FS_NS + k) * C + c];
  }
  const double* __restrict__ ip = in + c;
  double* __restrict__ op = out + c;
  constexpr int U = 8;
  double cur[U], nxt[U];
  long long n = n0;
  if (n1 - n0 >= U) {
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = ip[(first + dir * (n0 + u)) * C];
  }
  for (; n + U <= n1; n += U) {
    const bool more = n + 2 * U <= n1;
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) nxt[u] = ip[(first + dir * (n + U + u)) * C];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // scipy lfilter order: y = z0 + b0*x;  z_k = (z_{k+1} + x*b_{k+1}) - y*a_{k+1}
      const double yv = z[0] + bb[0] * cur[u];
#pragma unroll
      for (int k = 0; k < FS_NS; ++k) z[k] = (z[k + 1] + cur[u] * bb[k + 1]) - yv * aa[k + 1];
      if constexpr (APPLY) op[(first + dir * (n + u)) * C] = yv;
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
  }
  for (; n < n1; ++n) {
    const double xs = ip[(first + dir * n) * C];
    const double yv = z[0] + bb[0] * xs;
#pragma unroll
    for (int k = 0; k < FS_NS; ++k) z[k] = (z[k + 1] + xs * bb[k + 1]) - yv * aa[k + 1];
    if constexpr (APPLY) op[(first + dir * n) * C] = yv;
  }
  if constexpr (!APPLY) {
#pragma unroll
    for (int k = 0; k < FS_NS; ++k) S[(j * FS_NS + k) * C + c] = z[k];
  }
}

// One workgroup per channel; a thread per (block, state row): FS_BLK blocks x 8 rows = 1024 threads per chunk of blocks, the
// state carried from chunk to chunk.  Row i of v <- M u + v is ONE compensated dot product of 8 terms (two_prod by fma,
// two_sum, the low words of M and all rounding errors in a second accumulator): correctly rounded up to ~1e-31 of the terms.
// The matrices of all levels sit in LDS (a lane reads its own row), the vectors are exchanged through LDS.
constexpr int FS_BLK = 128;
__device__ __forceinline__ double scan_row(const double* __restrict__ Mrow, const double* __restrict__ u, double v) {
  double s = v, e = 0.0;
#pragma unroll
  for (int k = 0; k < FS_NS; ++k) {
    const double mh = Mrow[2 * k], ml = Mrow[2 * k + 1], uk = u[k];
    const double pr = mh * uk;
    const double pe = fma(mh, uk, -pr);                   // two_prod: mh * uk = pr + pe exactly
    const double s2 = s + pr;                             // two_sum: s + pr = s2 + se exactly
    const double bv = s2 - s;
    const double se = (s - (s2 - bv)) + (pr - bv);
    s = s2;
    e += (pe + se) + ml * uk;
  }
  return s + e;
}

__global__ __launch_bounds__(FS_BLK * FS_NS) void filtfilt_scan_prefix_kernel(const double* __restrict__ S, double* __restrict__ Z,
                                                                              const double* __restrict__ M, int nlev,
                                                                              const double* __restrict__ zi,
                                                                              const double* __restrict__ in, int C, long long nb,
                                                                              int ntaps, long long first) {
  __shared__ __attribute__((aligned(16))) double sm[7 * FS_NS * FS_NS * 2];      // levels 0..6 of A^(L 2^m): 7 KB
  __shared__ __attribute__((aligned(16))) double sv[2][FS_BLK][FS_NS];           // 16 KB
  __shared__ double scar[FS_NS];
  const int c = blockIdx.x, tid = threadIdx.x, i = tid & 7, jl = tid >> 3;
  for (int q = tid; q < 7 * FS_NS * FS_NS * 2; q += FS_BLK * FS_NS) sm[q] = q < nlev * FS_NS * FS_NS * 2 ? M[q] : 0.0;
  if (tid < FS_NS) scar[tid] = tid < ntaps - 1 ? zi[tid] * in[first * C + c] : 0.0;     // scipy: zi * x_ext[0]
  __syncthreads();
  for (long long chunk0 = 0; chunk0 < nb; chunk0 += FS_BLK) {
    const long long j = chunk0 + jl;
    double v = j < nb ? S[(j * FS_NS + i) * C + c] : 0.0;
    if (jl == 0) {
      Z[(chunk0 * FS_NS + i) * C + c] = scar[i];
      v = scan_row(sm + i * FS_NS * 2, scar, v);          // state at the END of the chunk's first block
    }
    int buf = 0;
    const int nact = (nb - chunk0 < FS_BLK) ? (int)(nb - chunk0) : FS_BLK;      // blocks of this chunk (uniform)
#pragma unroll 1
    for (int lev = 0, d = 1; d < nact; ++lev, d <<= 1) {
      sv[buf][jl][i] = v;
      __syncthreads();
      if (jl >= d) v = scan_row(sm + (lev * FS_NS + i) * FS_NS * 2, sv[buf][jl - d], v);
      buf ^= 1;
    }
    // v = row i of the state at the end of block j = at the start of block j + 1
    if (j + 1 < nb) Z[((j + 1) * FS_NS + i) * C + c] = v;
    __syncthreads();                                       // (scar is still being read by the first block's lanes above)
    if (jl == nact - 1) scar[i] = v;
    __syncthreads();
  }
}

// the odd-extended recording TIME-MAJOR, like filtfilt_build_kernel, with both sides coalesced: 64 x 64 tiles through LDS
// for the centre part, the 2 x edge extension rows directly
template <typename TIN>
__global__ __launch_bounds__(256) void filtfilt_build_tiled_kernel(const void* __restrict__ x, double* __restrict__ work, int C,
                                                                   long long T, int edge) {
  __shared__ double tile[64][65];
  const long long t0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;       // 64 x 4
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = c0 + ly + 4 * r;
    const long long t = t0 + lx;
    if (t < T && c < C) tile[ly + 4 * r][lx] = ld_as_f64<TIN>(x, (long long)c * T + t);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long long t = t0 + ly + 4 * r;
    const int c = c0 + lx;
    if (t < T && c < C) work[(t + edge) * C + c] = tile[lx][ly + 4 * r];
  }
  if (blockIdx.x == 0) {                                         // the extension rows of this channel tile
    for (int q = threadIdx.x; q < 2 * edge * 64; q += 256) {
      const int c = c0 + (q & 63), e = q >> 6;
      if (c >= C) continue;
      const long long ti = e < edge ? e : T + edge + (e - edge);
      work[ti * C + c] = ext_sample<TIN>(x, (long long)c * T, T, edge, ti);
    }
  }
}

constexpr int MAX_SEC = 8;
template <typename TIN>
__global__ __launch_bounds__(64) void sosfilt_kernel(const void* __restrict__ x, const double* __restrict__ sos,
                                                     double* __restrict__ y, int C, long long T, int nsec) {
#pragma clang fp contract(off)
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s[MAX_SEC][6], z0[MAX_SEC], z1[MAX_SEC];
#pragma unroll
  for (int q = 0; q < MAX_SEC; ++q) {
#pragma unroll
    for (int k = 0; k < 6; ++k) s[q][k] = q < nsec ? sos[q * 6 + k] : (k == 0 || k == 3 ? 1.0 : 0.0);
    z0[q] = z1[q] = 0.0;
  }
  const long long cb = (long long)c * T;
  for (long long i = 0; i < T; ++i) {
    double v = ld_as_f64<TIN>(x, cb + i);
#pragma unroll
    for (int q = 0; q < MAX_SEC; ++q)
      if (q < nsec) {
        // scipy _sosfilt order: y = b0*x + z0; z0 = (b1*x - a1*y) + z1; z1 = b2*x - a2*y
        const double yv = __dadd_rn(__dmul_rn(s[q][0], v), z0[q]);
        z0[q] = __dadd_rn(__dsub_rn(__dmul_rn(s[q][1], v), __dmul_rn(s[q][4], yv)), z1[q]);
        z1[q] = __dsub_rn(__dmul_rn(s[q][2], v), __dmul_rn(s[q][5], yv));
        v = yv;
      }
    y[cb + i] = v;
  }
}

// causal FIR bank with zero initial state; mean over bands
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void fir_bank_kernel(const void* __restrict__ x, const double* __restrict__ taps,
                                                       TOUT* __restrict__ y, long long T, int nb, int ntap) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int c = blockIdx.y;
  const long long t0 = (long long)blockIdx.x * SIG_TB;
  const int win = SIG_TB + ntap - 1;
  for (int i = threadIdx.x; i < win; i += blockDim.x) {
    const long long src = t0 - (ntap - 1) + i;
    xs[i] = (src >= 0 && src < T) ? ld_as_f64<TIN>(x, (long long)c * T + src) : 0.0;
  }
  __syncthreads();
  double acc[SIG_SPT];
#pragma unroll
  for (int s = 0; s < SIG_SPT; ++s) acc[s] = 0.0;
  for (int b = 0; b < nb; ++b) {
    double part[SIG_SPT];
#pragma unroll
    for (int s = 0; s < SIG_SPT; ++s) part[s] = 0.0;
    for (int k = 0; k < ntap; ++k) {
      const double h = taps[(long long)b * ntap + k];
#pragma unroll
      for (int s = 0; s < SIG_SPT; ++s) part[s] = fma(h, xs[threadIdx.x + s * 256 + (ntap - 1) - k], part[s]);
    }
#pragma unroll
    for (int s = 0; s < SIG_SPT; ++s) acc[s] += part[s];
  }
#pragma unroll
  for (int s = 0; s < SIG_SPT; ++s) {
    const long long t = t0 + threadIdx.x + s * 256;
    if (t < T) y[(long long)c * T + t] = (TOUT)(acc[s] / nb);
  }
}

}  // namespace tl

using namespace tl;

extern "C" int tl_gauss_envelope(const void* x, int x_is_f64, const double* taps, double* y, int C, int64_t T,
                                 int nb, int ntap, int half, int envelope, void* stream) {
  TL_REQUIRE(x && taps && y, "gauss_envelope: null pointer");
  TL_REQUIRE(C > 0 && C <= 65535 && T > 0 && nb > 0, "gauss_envelope: bad sizes");
  TL_REQUIRE(ntap >= 1 && ntap <= T, "gauss_envelope: ntap must be in 1..T (fold the kernel on the host)");
  TL_REQUIRE(half >= 0 && half < ntap, "gauss_envelope: half out of range");
  const size_t lds = (size_t)(SIG_TB + ntap - 1) * sizeof(double);
  TL_REQUIRE(lds <= 64 * 1024, "gauss_envelope: %d taps exceed the LDS window", ntap);
  dim3 grid((unsigned)((T + SIG_TB - 1) / SIG_TB), (unsigned)C);
  hipStream_t st = (hipStream_t)stream;
#define GE_LAUNCH(TIN)                                                                                              \
  do {                                                                                                              \
    if (nb == 8)                                                                                                    \
      hipLaunchKernelGGL((gauss_envelope_kernel<TIN, 8>), grid, dim3(256), lds, st, x, taps, y, (long long)T, ntap, \
                         half, envelope);                                                                           \
    else                                                                                                            \
      hipLaunchKernelGGL((gauss_envelope_generic_kernel<TIN>), grid, dim3(256), lds, st, x, taps, y, (long long)T,  \
                         nb, ntap, half, envelope);                                                                 \
  } while (0)
  if (x_is_f64) GE_LAUNCH(double); else GE_LAUNCH(float);
#undef GE_LAUNCH
  return check_launch("gauss_envelope");
}

// Hermitian form of tl_gauss_envelope for 8 bands: taps (half + 1, 8, 2) = h_b[n], n = 0..half (tap-major); h_b[-n] = conj(h_b[n]) is implied
extern "C" int tl_gauss_envelope_sym(const void* x, int x_is_f64, const double* taps, double* y, int C, int64_t T, int nb,
                                     int half, int envelope, void* stream) {
  TL_REQUIRE(x && taps && y, "gauss_envelope_sym: null pointer");
  TL_REQUIRE(C > 0 && C <= 65535 && T > 0, "gauss_envelope_sym: bad sizes");
  TL_REQUIRE(nb == 8, "gauss_envelope_sym: 8 bands only (use tl_gauss_envelope)");
  TL_REQUIRE(half >= 0 && 2LL * half + 1 <= T, "gauss_envelope_sym: 2 half + 1 taps must fit the recording");
  const size_t lds = (size_t)(SIG_TB + 2 * half) * sizeof(double);
  TL_REQUIRE(lds <= 64 * 1024, "gauss_envelope_sym: %d taps exceed the LDS window", 2 * half + 1);
  dim3 grid((unsigned)((T + SIG_TB - 1) / SIG_TB), (unsigned)C);
  hipStream_t st = (hipStream_t)stream;
  if (x_is_f64)
    hipLaunchKernelGGL((gauss_envelope_sym_kernel<double, 8>), grid, dim3(256), lds, st, x, taps, y, (long long)T, half, envelope);
  else
    hipLaunchKernelGGL((gauss_envelope_sym_kernel<float, 8>), grid, dim3(256), lds, st, x, taps, y, (long long)T, half, envelope);
  return check_launch("gauss_envelope_sym");
}

// overlap-save form of tl_gauss_envelope for 8 bands on a 1024-point FFT: G (8, 1024, 2) = FFT_1024 of each band's truncated
// kernel h_b[n], n = -half..half placed at 0..2 half, divided by 1024; tw (1024, 2) = (cos, -sin)(2 pi m / 1024)
extern "C" int tl_hilbert_ols(const void* x, int x_is_f64, const double* G, const double* tw, double* y, int C, int64_t T,
                              int nb, int half, int nfft, int envelope, void* stream) {
  TL_REQUIRE(x && G && tw && y, "hilbert_ols: null pointer");
  TL_REQUIRE(C > 0 && C <= 65535 && T > 0, "hilbert_ols: bad sizes");
  TL_REQUIRE(nb >= 1 && nb <= 64, "hilbert_ols: 1..64 bands");
  TL_REQUIRE(nfft == OLS_N, "hilbert_ols: nfft must be %d", OLS_N);
  TL_REQUIRE(half >= 0 && 2 * half <= OLS_N / 2 && 2LL * half + 1 <= T, "hilbert_ols: the kernels must span at most %d taps", OLS_N / 2 + 1);
  const int Lv = OLS_N - 2 * half;
  dim3 grid((unsigned)((T + Lv - 1) / Lv), (unsigned)C);
  hipStream_t st = (hipStream_t)stream;
  const ols_d2* g2 = reinterpret_cast<const ols_d2*>(G);
  const ols_d2* t2 = reinterpret_cast<const ols_d2*>(tw);
  if (x_is_f64)
    if (nb == 8) hipLaunchKernelGGL((ols_bank_kernel<double, double, 8>), grid, dim3(OLS_Q), 0, st, x, g2, t2, y, (long long)T, nb, half, 2 * half, 1, envelope);
    else hipLaunchKernelGGL((ols_bank_kernel<double, double, 0>), grid, dim3(OLS_Q), 0, st, x, g2, t2, y, (long long)T, nb, half, 2 * half, 1, envelope);
  else
    if (nb == 8) hipLaunchKernelGGL((ols_bank_kernel<float, double, 8>), grid, dim3(OLS_Q), 0, st, x, g2, t2, y, (long long)T, nb, half, 2 * half, 1, envelope);
    else hipLaunchKernelGGL((ols_bank_kernel<float, double, 0>), grid, dim3(OLS_Q), 0, st, x, g2, t2, y, (long long)T, nb, half, 2 * half, 1, envelope);
  return check_launch("hilbert_ols");
}

// band-limited overlap-save form (ols_bank_bl_kernel): Gp (nb, 4, 256, 2) = G_b[(k0_b + k) % 1024] . exp(+2 pi i k r / 1024)
// for residue r < 4 and k < 256, where G_b is tl_hilbert_ols's spectrum and every |G_b| outside its window
// [k0_b, k0_b + 256) is negligible (the caller's check); k0 (nb) int32 on the device
extern "C" int tl_hilbert_ols_bl(const void* x, int x_is_f64, const double* Gp, const int* k0, const double* tw, double* y, int C,
                                 int64_t T, int nb, int half, int nfft, int envelope, void* stream) {
  TL_REQUIRE(x && Gp && k0 && tw && y, "hilbert_ols_bl: null pointer");
  TL_REQUIRE(C > 0 && C <= 65535 && T > 0, "hilbert_ols_bl: bad sizes");
  TL_REQUIRE(nb >= 1 && nb <= 64, "hilbert_ols_bl: 1..64 bands");
  TL_REQUIRE(nfft == OLS_N, "hilbert_ols_bl: nfft must be %d", OLS_N);
  TL_REQUIRE(half >= 0 && 2 * half <= OLS_N / 2 && 2LL * half + 1 <= T, "hilbert_ols_bl: the kernels must span at most %d taps", OLS_N / 2 + 1);
  TL_REQUIRE(T >= OLS_N, "hilbert_ols_bl: the recording must hold at least %d samples", OLS_N);
  const int Lv = OLS_N - 2 * half;
  dim3 grid((unsigned)((T + Lv - 1) / Lv), (unsigned)C);
  hipStream_t st = (hipStream_t)stream;
  const ols_d2* g2 = reinterpret_cast<const ols_d2*>(Gp);
  const ols_d2* t2 = reinterpret_cast<const ols_d2*>(tw);
  // x_is_f64: 1 float64 recording, fp64 transforms; 0 float32 recording, fp32 transforms (the reference's own precision for
  // that dtype: scipy.fft stays in complex64); 2 float32 recording, fp64 transforms (round 3's behaviour)
  const bool f32_math = x_is_f64 == 0;
  if (x_is_f64 == 1)
    if (nb == 8) hipLaunchKernelGGL((ols_bank_bl_kernel<double, 8, double>), grid, dim3(OLS_Q), 0, st, x, g2, k0, t2, y, (long long)T, nb, half, 2 * half, envelope);
    else hipLaunchKernelGGL((ols_bank_bl_kernel<double, 0, double>), grid, dim3(OLS_Q), 0, st, x, g2, k0, t2, y, (long long)T, nb, half, 2 * half, envelope);
  else
    if (f32_math && nb == 8) hipLaunchKernelGGL((ols_bank_bl_kernel<float, 8, float>), grid, dim3(OLS_Q), 0, st, x, g2, k0, t2, y, (long long)T, nb, half, 2 * half, envelope);
    else if (f32_math) hipLaunchKernelGGL((ols_bank_bl_kernel<float, 0, float>), grid, dim3(OLS_Q), 0, st, x, g2, k0, t2, y, (long long)T, nb, half, 2 * half, envelope);
    else if (nb == 8) hipLaunchKernelGGL((ols_bank_bl_kernel<float, 8, double>), grid, dim3(OLS_Q), 0, st, x, g2, k0, t2, y, (long long)T, nb, half, 2 * half, envelope);
    else hipLaunchKernelGGL((ols_bank_bl_kernel<float, 0, double>), grid, dim3(OLS_Q), 0, st, x, g2, k0, t2, y, (long long)T, nb, half, 2 * half, envelope);
  return check_launch("hilbert_ols_bl");
}

extern "C" int tl_filtfilt_f64(const void* x, int x_is_f64, const double* b, const double* a, const double* zi,
                               double* y, double* work, int C, int64_t T, int ntaps, void* stream) {
  TL_REQUIRE(x && b && a && zi && y && work, "filtfilt: null pointer");
  TL_REQUIRE(ntaps >= 2 && ntaps <= MAX_TAPS, "filtfilt: ntaps must be 2..%d", MAX_TAPS);
  TL_REQUIRE(C > 0 && T > 3LL * ntaps, "filtfilt: the input must be longer than padlen = %d", 3 * ntaps);
  hipStream_t st = (hipStream_t)stream;
  const int edge = 3 * ntaps;
  const long long next = T + 2LL * edge;
  long long g = (next * C + 255) / 256;
  if (g > 8192) g = 8192;
  if (x_is_f64)
    hipLaunchKernelGGL((filtfilt_build_kernel<double>), dim3((unsigned)g), dim3(256), 0, st, x, work, C, (long long)T, edge);
  else
    hipLaunchKernelGGL((filtfilt_build_kernel<float>), dim3((unsigned)g), dim3(256), 0, st, x, work, C, (long long)T, edge);
  dim3 grid((unsigned)((C + 63) / 64));
  if (ntaps <= 9 && FF_LANES8)                              // state over 8 lanes per channel: 8 channels per wave
    hipLaunchKernelGGL(filtfilt_iir8_kernel, dim3((unsigned)((C + 3) / 4)), dim3(64), 0, st, b, a, zi, work, C, next, ntaps);
  else if (ntaps <= 5)
    hipLaunchKernelGGL((filtfilt_iir_kernel<5>), grid, dim3(64), 0, st, b, a, zi, work, C, next, ntaps);
  else if (ntaps <= 9)
    hipLaunchKernelGGL((filtfilt_iir_kernel<9>), grid, dim3(64), 0, st, b, a, zi, work, C, next, ntaps);
  else
    hipLaunchKernelGGL((filtfilt_iir_kernel<MAX_TAPS>), grid, dim3(64), 0, st, b, a, zi, work, C, next, ntaps);
  hipLaunchKernelGGL(filtfilt_out_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)((C + 63) / 64)), dim3(256), 0, st, work, y, C,
                     (long long)T, edge);
  return check_launch("filtfilt");
}

extern "C" int tl_filtfilt_scan_f64(const void* x, int x_is_f64, const double* b, const double* a, const double* zi,
                                    const double* M, int nlev, double* y, double* work, double* swork, int C, int64_t T,
                                    int ntaps, int L, void* stream) {
  TL_REQUIRE(x && b && a && zi && M && y && work && swork, "filtfilt_scan: null pointer");
  TL_REQUIRE(ntaps >= 2 && ntaps <= FS_NS + 1, "filtfilt_scan: ntaps must be 2..%d", FS_NS + 1);
  TL_REQUIRE(C > 0 && C <= 65535 && T > 3LL * ntaps && L >= 8, "filtfilt_scan: bad sizes (the input must be longer than padlen = %d)", 3 * ntaps);
  hipStream_t st = (hipStream_t)stream;
  const int edge = 3 * ntaps;
  const long long next = T + 2LL * edge;
  const long long nb = (next + L - 1) / L;
  TL_REQUIRE(nb <= 65535, "filtfilt_scan: more than 65 535 blocks of %d samples (raise L)", L);
  TL_REQUIRE(nlev >= 7, "filtfilt_scan: 7 matrix levels A^(L 2^m), m < 7, needed (chunks of %d blocks)", FS_BLK);
  const dim3 tgrid((unsigned)((T + 63) / 64), (unsigned)((C + 63) / 64));
  if (x_is_f64)
    hipLaunchKernelGGL((filtfilt_build_tiled_kernel<double>), tgrid, dim3(256), 0, st, x, work, C, (long long)T, edge);
  else
    hipLaunchKernelGGL((filtfilt_build_tiled_kernel<float>), tgrid, dim3(256), 0, st, x, work, C, (long long)T, edge);
  double* S = swork;
  double* Z = swork + nb * FS_NS * C;
  const dim3 grid((unsigned)((C + 63) / 64), (unsigned)nb);
  for (int pass = 0; pass < 2; ++pass) {
    const double* in = work + (long long)pass * next * C;
    double* out = work + (long long)(1 - pass) * next * C;
    const long long first = pass == 0 ? 0 : next - 1, dir = pass == 0 ? 1 : -1;
    hipLaunchKernelGGL((filtfilt_scan_block_kernel<false>), grid, dim3(64), 0, st, b, a, in, out, nullptr, S, C, next, ntaps, L,
                       first, dir);
    hipLaunchKernelGGL(filtfilt_scan_prefix_kernel, dim3((unsigned)C), dim3(FS_BLK * FS_NS), 0, st, S, Z, M, nlev, zi, in, C, nb,
                       ntaps, first);
    hipLaunchKernelGGL((filtfilt_scan_block_kernel<true>), grid, dim3(64), 0, st, b, a, in, out, Z, nullptr, C, next, ntaps, L,
                       first, dir);
  }
  hipLaunchKernelGGL(filtfilt_out_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)((C + 63) / 64)), dim3(256), 0, st, work, y, C,
                     (long long)T, edge);
  return check_launch("filtfilt_scan");
}

extern "C" int tl_sosfilt_f64(const void* x, int x_is_f64, const double* sos, double* y, int C, int64_t T, int nsec,
                              void* stream) {
  TL_REQUIRE(x && sos && y, "sosfilt: null pointer");
  TL_REQUIRE(nsec >= 1 && nsec <= MAX_SEC && C > 0 && T > 0, "sosfilt: nsec must be 1..%d", MAX_SEC);
  dim3 grid((unsigned)((C + 63) / 64));
  hipStream_t st = (hipStream_t)stream;
  if (x_is_f64)
    hipLaunchKernelGGL((sosfilt_kernel<double>), grid, dim3(64), 0, st, x, sos, y, C, (long long)T, nsec);
  else
    hipLaunchKernelGGL((sosfilt_kernel<float>), grid, dim3(64), 0, st, x, sos, y, C, (long long)T, nsec);
  return check_launch("sosfilt");
}

// overlap-save form of tl_fir_bank (same causal convolution, zero initial state): G (nb, 1024, 2) = FFT_1024 of each band's taps
// / 1024, tw as for tl_hilbert_ols; ntap <= 513
extern "C" int tl_fir_bank_ols(const void* x, int x_is_f64, const double* G, const double* tw, void* y, int y_is_f64, int C,
                               int64_t T, int nb, int ntap, void* stream) {
  TL_REQUIRE(x && G && tw && y, "fir_bank_ols: null pointer");
  TL_REQUIRE(C > 0 && C <= 65535 && T > 0 && nb >= 1 && nb <= 64, "fir_bank_ols: bad sizes");
  TL_REQUIRE(ntap >= 1 && ntap - 1 <= OLS_N / 2, "fir_bank_ols: at most %d taps", OLS_N / 2 + 1);
  const int skip = ntap - 1, Lv = OLS_N - skip;
  dim3 grid((unsigned)((T + Lv - 1) / Lv), (unsigned)C);
  hipStream_t st = (hipStream_t)stream;
  const ols_d2* g2 = reinterpret_cast<const ols_d2*>(G);
  const ols_d2* t2 = reinterpret_cast<const ols_d2*>(tw);
  if (x_is_f64 && y_is_f64)
    hipLaunchKernelGGL((ols_bank_kernel<double, double, 0>), grid, dim3(OLS_Q), 0, st, x, g2, t2, (double*)y, (long long)T, nb, skip, skip, 0, 0);
  else if (!x_is_f64 && y_is_f64)
    hipLaunchKernelGGL((ols_bank_kernel<float, double, 0>), grid, dim3(OLS_Q), 0, st, x, g2, t2, (double*)y, (long long)T, nb, skip, skip, 0, 0);
  else if (!x_is_f64 && !y_is_f64)
    hipLaunchKernelGGL((ols_bank_kernel<float, float, 0>), grid, dim3(OLS_Q), 0, st, x, g2, t2, (float*)y, (long long)T, nb, skip, skip, 0, 0);
  else
    hipLaunchKernelGGL((ols_bank_kernel<double, float, 0>), grid, dim3(OLS_Q), 0, st, x, g2, t2, (float*)y, (long long)T, nb, skip, skip, 0, 0);
  return check_launch("fir_bank_ols");
}

extern "C" int tl_fir_bank(const void* x, int x_is_f64, const double* taps, void* y, int y_is_f64, int C, int64_t T,
                           int nb, int ntap, void* stream) {
  TL_REQUIRE(x && taps && y, "fir_bank: null pointer");
  TL_REQUIRE(C > 0 && C <= 65535 && T > 0 && nb > 0 && ntap > 0, "fir_bank: bad sizes");
  const size_t lds = (size_t)(SIG_TB + ntap - 1) * sizeof(double);
  TL_REQUIRE(lds <= 64 * 1024, "fir_bank: %d taps exceed the LDS window", ntap);
  dim3 grid((unsigned)((T + SIG_TB - 1) / SIG_TB), (unsigned)C);
  hipStream_t st = (hipStream_t)stream;
  if (x_is_f64 && y_is_f64)
    hipLaunchKernelGGL((fir_bank_kernel<double, double>), grid, dim3(256), lds, st, x, taps, (double*)y, (long long)T, nb, ntap);
  else if (!x_is_f64 && y_is_f64)
    hipLaunchKernelGGL((fir_bank_kernel<float, double>), grid, dim3(256), lds, st, x, taps, (double*)y, (long long)T, nb, ntap);
  else if (!x_is_f64 && !y_is_f64)
    hipLaunchKernelGGL((fir_bank_kernel<float, float>), grid, dim3(256), lds, st, x, taps, (float*)y, (long long)T, nb, ntap);
  else
    hipLaunchKernelGGL((fir_bank_kernel<double, float>), grid, dim3(256), lds, st, x, taps, (float*)y, (long long)T, nb, ntap);
  return check_launch("fir_bank");
}

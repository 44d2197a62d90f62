// Remaining preprocess/signal steps behind the same run(data, params) plugin ABI:
// per-channel z-score (whole recording or a baseline interval), common-average re-reference and
// rolling z-score.  HBM-bound reductions; statistics are accumulated in fp64.
#include "tonal_common.h"

namespace tl {

template <typename T>
__device__ __forceinline__ double ldd(const void* p, long long i) { return (double)reinterpret_cast<const T*>(p)[i]; }

__device__ __forceinline__ double block_sum(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int off = blockDim.x / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  const double s = red[0];
  __syncthreads();
  return s;
}

// stats[c] = (mean, population std) of x[c][t0:t1]  (numpy: mean, then mean of squared deviations)
template <typename T>
__global__ __launch_bounds__(256) void row_stats_kernel(const void* __restrict__ x, double* __restrict__ stats, long long Tn,
                                                        long long t0, long long t1) {
  __shared__ double red[256];
  const long long base = (long long)blockIdx.x * Tn;
  double s = 0.0;
  for (long long t = t0 + threadIdx.x; t < t1; t += blockDim.x) s += ldd<T>(x, base + t);
  const double mean = block_sum(s, red) / (double)(t1 - t0);
  double q = 0.0;
  for (long long t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
    const double d = ldd<T>(x, base + t) - mean;
    q += d * d;
  }
  const double var = block_sum(q, red) / (double)(t1 - t0);
  if (threadIdx.x == 0) {
    stats[2 * blockIdx.x] = mean;
    stats[2 * blockIdx.x + 1] = sqrt(var);
  }
}

// y[c][t] = (x[c][t] - mean[c]) / std[c]; NaN results -> 0 when zero_nans
template <typename T>
__global__ __launch_bounds__(256) void row_normalise_kernel(const void* __restrict__ x, const double* __restrict__ stats,
                                                            T* __restrict__ y, long long total, long long Tn, int zero_nans) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long c = i / Tn;
    double v = (ldd<T>(x, i) - stats[2 * c]) / stats[2 * c + 1];
    if (zero_nans && v != v) v = 0.0;
    y[i] = (T)v;
  }
}

// y[c][t] = x[c][t] - mean_{c in include} x[c][t]
template <typename T>
__global__ __launch_bounds__(256) void car_kernel(const void* __restrict__ x, const int* __restrict__ include, T* __restrict__ y,
                                                  int C, long long Tn, int n_inc) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < Tn; t += (long long)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int c = 0; c < C; ++c)
      if (include[c]) s += ldd<T>(x, (long long)c * Tn + t);
    const double m = s / (double)n_inc;
    for (int c = 0; c < C; ++c) y[(long long)c * Tn + t] = (T)(ldd<T>(x, (long long)c * Tn + t) - m);
  }
}

// pandas rolling(window=W, min_periods=1): mean and sample std (ddof=1) over the non-NaN values of
// x[c][max(0,t-W+1) .. t]; z = (x - mean)/std.  Two-pass per output on an LDS-staged window.
constexpr int RZ_TB = 256;
template <typename T>
__global__ __launch_bounds__(256) void rolling_zscore_kernel(const void* __restrict__ x, double* __restrict__ y, long long Tn,
                                                             int W, int zero_nans) {
  extern __shared__ __attribute__((aligned(16))) double xs[];     // [RZ_TB + W - 1]
  const int c = blockIdx.y;
  const long long t0 = (long long)blockIdx.x * RZ_TB;
  const int win = RZ_TB + W - 1;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  for (int i = threadIdx.x; i < win; i += blockDim.x) {
    const long long src = t0 - (W - 1) + i;
    xs[i] = (src >= 0 && src < Tn) ? ldd<T>(x, (long long)c * Tn + src) : nan;
  }
  __syncthreads();
  const long long t = t0 + threadIdx.x;
  if (t >= Tn) return;
  double s = 0.0;
  int n = 0;
  for (int k = 0; k < W; ++k) {
    const double v = xs[threadIdx.x + k];
    if (v == v) {
      s += v;
      ++n;
    }
  }
  const double xv = xs[threadIdx.x + W - 1];
  double out = nan;
  if (n >= 1) {
    const double mean = s / n;
    double q = 0.0;
    for (int k = 0; k < W; ++k) {
      const double v = xs[threadIdx.x + k];
      if (v == v) q += (v - mean) * (v - mean);
    }
    if (n >= 2) out = (xv - mean) / sqrt(q / (n - 1));
  }
  if (zero_nans && out != out) out = 0.0;
  y[(long long)c * Tn + t] = out;
}


// ------------------------------------------------------------------------------------------
// FFT resampling (scipy.signal.resample as used by preprocess/signal/downsample.py:21-27): the DFT of
// an arbitrary length n is evaluated with Bluestein's chirp-z identity on a power-of-two Stockham
// FFT (fp64, complex interleaved, batched over channels).  Chirps, their spectra and the twiddle
// table are coefficient data prepared by the host.
// ------------------------------------------------------------------------------------------
struct c64 { double re, im; };
__device__ __forceinline__ c64 cmul(c64 a, c64 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ c64 cconj(c64 a) { return {a.re, -a.im}; }

// one radix-2 Stockham pass: out[j0] = a + w b, out[j0 + ns] = a - w b
__global__ __launch_bounds__(256) void fft_pass_kernel(const c64* __restrict__ in, c64* __restrict__ out,
                                                       const c64* __restrict__ tw, long long total_half, int m2, int ns,
                                                       int inverse) {
  const int half = m2 >> 1;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total_half; i += (long long)gridDim.x * blockDim.x) {
    const long long c = i / half;
    const int j = (int)(i % half);
    const int k = j & (ns - 1);
    c64 w = tw[(long long)k * (half / ns)];                 // exp(-2 pi i k / (2 ns))
    if (inverse) w.im = -w.im;
    const c64 a = in[c * m2 + j];
    const c64 b = cmul(in[c * m2 + j + half], w);
    const int j0 = ((j - k) << 1) + k;
    out[c * m2 + j0] = {a.re + b.re, a.im + b.im};
    out[c * m2 + j0 + ns] = {a.re - b.re, a.im - b.im};
  }
}

// a[c][j] = x[c][j] * conj(w[j]) for j < n, 0 up to m2
template <typename T>
__global__ __launch_bounds__(256) void rs_prep_kernel(const void* __restrict__ x, const c64* __restrict__ w, c64* __restrict__ a,
                                                      long long total, int n, int m2) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % m2);
    const long long c = i / m2;
    c64 v = {0.0, 0.0};
    if (j < n) {
      const double xv = ldd<T>(x, c * n + j);
      v = {xv * w[j].re, -xv * w[j].im};
    }
    a[i] = v;
  }
}
// A[c][m] *= Bf[m]
__global__ __launch_bounds__(256) void rs_cmul_kernel(c64* __restrict__ A, const c64* __restrict__ Bf, long long total, int m2) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    A[i] = cmul(A[i], Bf[i % m2]);
}
// Spectrum hand-over.  c1 (c, m2a) holds the chirp-convolved forward transform (un-normalised inverse
// FFT, scale 1/m2a folded in here): X[k] = conj(w1[k]) c1[k] / m2a, k < nx.  Build the half-spectrum Y
// of scipy.signal.resample (copy up to the smaller Nyquist; double / halve the shared Nyquist bin),
// extend it Hermitian-ly as irfft does, and emit a2[k] = conj(Yc[k]) conj(w2[k]) padded to m2b.
__global__ __launch_bounds__(256) void rs_spec_kernel(const c64* __restrict__ c1, const c64* __restrict__ w1,
                                                      const c64* __restrict__ w2, c64* __restrict__ a2, long long total,
                                                      int nx, int num, int m2a, int m2b) {
  const int N = nx < num ? nx : num;
  const int nyq = N / 2 + 1;
  const double inv = 1.0 / (double)m2a;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % m2b);
    const long long c = i / m2b;
    c64 v = {0.0, 0.0};
    if (k < num) {
      const int kk = (k <= num / 2) ? k : num - k;          // index into the half spectrum Y
      c64 y = {0.0, 0.0};
      if (kk < nyq) {
        const c64 cx = c1[c * m2a + kk];
        y = cmul(cconj(w1[kk]), cx);
        y.re *= inv;
        y.im *= inv;
        if (N % 2 == 0 && kk == N / 2) {
          const double f = (num < nx) ? 2.0 : ((nx < num) ? 0.5 : 1.0);
          y.re *= f;
          y.im *= f;
        }
      }
      if (kk == 0 || (num % 2 == 0 && kk == num / 2)) y.im = 0.0;      // C2R ignores these imaginary parts
      const c64 yc = (k <= num / 2) ? y : cconj(y);
      v = cmul(cconj(yc), cconj(w2[k]));
    }
    a2[i] = v;
  }
}
// y[c][m] = Re(conj(w2[m]) c2[m]) / (m2b * nx)      (ifft = conj(DFT(conj .))/num, times num/nx)
template <typename T>
__global__ __launch_bounds__(256) void rs_out_kernel(const c64* __restrict__ c2, const c64* __restrict__ w2, T* __restrict__ y,
                                                     long long total, int num, int m2b, double scale) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i % num);
    const long long c = i / num;
    const c64 d = cmul(cconj(w2[m]), c2[c * m2b + m]);
    y[i] = (T)(d.re * scale);
  }
}

// ------------------------------------------------------------------------------------------
// DFT-domain Gaussian Hilbert bank (frequency_filter.py:155-184 taken literally): X = DFT(x); per band
// z_b = IDFT(X . K_b), K_b = H_b x analytic multiplier (real, host coefficient data); y = mean_b |z_b| or Re z_b.
// Arbitrary recording length through Bluestein on the Stockham passes above.  This is the path for bands whose
// time-domain kernels are too long for tl_gauss_envelope's LDS window (low bands at a raw recording rate).
// ------------------------------------------------------------------------------------------
// X[c][k] = conj(w[k]) c1[c][k] / m2, k < n   (c1 = chirp-convolved forward transform, un-normalised)
__global__ __launch_bounds__(256) void hb_spec_kernel(const c64* __restrict__ c1, const c64* __restrict__ w, c64* __restrict__ X,
                                                      long long total, int n, int m2) {
  const double inv = 1.0 / (double)m2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % n);
    const long long c = i / n;
    c64 v = cmul(cconj(w[k]), c1[c * m2 + k]);
    X[i] = {v.re * inv, v.im * inv};
  }
}
// a[c][k] = conj(X[c][k] K[k]) conj(w[k]) for k < n, 0 up to m2: Bluestein input of DFT(conj(Z)), Z = X K
__global__ __launch_bounds__(256) void hb_band_prep_kernel(const c64* __restrict__ X, const double* __restrict__ K,
                                                           const c64* __restrict__ w, c64* __restrict__ a, long long total,
                                                           int n, int m2) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % m2);
    const long long c = i / m2;
    c64 v = {0.0, 0.0};
    if (k < n) {
      const c64 x = X[c * n + k];
      const double kk = K[k];
      v = cmul(c64{x.re * kk, -x.im * kk}, cconj(w[k]));
    }
    a[i] = v;
  }
}
// d = conj(w[t]) c2[c][t] / m2 = DFT(conj Z)[t]; z[t] = conj(d) / n.  y (+)= (|d| or Re d) * scale
__global__ __launch_bounds__(256) void hb_accum_kernel(const c64* __restrict__ c2, const c64* __restrict__ w, double* __restrict__ y,
                                                       long long total, int n, int m2, double scale, int envelope, int first) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % n);
    const long long c = i / n;
    const c64 d = cmul(cconj(w[t]), c2[c * m2 + t]);
    const double v = (envelope ? sqrt(d.re * d.re + d.im * d.im) : d.re) * scale;
    y[i] = first ? v : y[i] + v;
  }
}

static inline unsigned sgrid(long long total) {
  long long g = (total + 255) / 256;
  if (g < 1) g = 1;
  if (g > 8192) g = 8192;
  return (unsigned)g;
}

}  // namespace tl
using namespace tl;

extern "C" int tl_row_zscore(const void* x, int is_f64, void* y, double* stats, int C, int64_t T, int64_t t0, int64_t t1,
                             int zero_nans, void* stream) {
  TL_REQUIRE(x && y && stats && C > 0 && T > 0, "row_zscore: bad arguments");
  TL_REQUIRE(t0 >= 0 && t1 <= T && t0 < t1, "row_zscore: statistics interval out of bounds");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)C * T;
  if (is_f64) {
    hipLaunchKernelGGL((row_stats_kernel<double>), dim3(C), dim3(256), 0, st, x, stats, (long long)T, (long long)t0, (long long)t1);
    hipLaunchKernelGGL((row_normalise_kernel<double>), dim3(sgrid(total)), dim3(256), 0, st, x, stats, (double*)y, total, (long long)T, zero_nans);
  } else {
    hipLaunchKernelGGL((row_stats_kernel<float>), dim3(C), dim3(256), 0, st, x, stats, (long long)T, (long long)t0, (long long)t1);
    hipLaunchKernelGGL((row_normalise_kernel<float>), dim3(sgrid(total)), dim3(256), 0, st, x, stats, (float*)y, total, (long long)T, zero_nans);
  }
  return check_launch("row_zscore");
}

extern "C" int tl_car(const void* x, int is_f64, const int32_t* include, void* y, int C, int64_t T, int n_inc, void* stream) {
  TL_REQUIRE(x && include && y && C > 0 && T > 0 && n_inc > 0, "car: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (is_f64)
    hipLaunchKernelGGL((car_kernel<double>), dim3(sgrid(T)), dim3(256), 0, st, x, include, (double*)y, C, (long long)T, n_inc);
  else
    hipLaunchKernelGGL((car_kernel<float>), dim3(sgrid(T)), dim3(256), 0, st, x, include, (float*)y, C, (long long)T, n_inc);
  return check_launch("car");
}

extern "C" int tl_rolling_zscore(const void* x, int is_f64, double* y, int C, int64_t T, int window, int zero_nans, void* stream) {
  TL_REQUIRE(x && y && C > 0 && C <= 65535 && T > 0, "rolling_zscore: bad arguments");
  TL_REQUIRE(window > 1, "rolling_zscore: window_size must be greater than 1.");
  const size_t lds = (size_t)(RZ_TB + window - 1) * sizeof(double);
  TL_REQUIRE(lds <= 64 * 1024, "rolling_zscore: window of %d samples exceeds the LDS tile", window);
  dim3 grid((unsigned)((T + RZ_TB - 1) / RZ_TB), (unsigned)C);
  hipStream_t st = (hipStream_t)stream;
  if (is_f64)
    hipLaunchKernelGGL((rolling_zscore_kernel<double>), grid, dim3(256), lds, st, x, y, (long long)T, window, zero_nans);
  else
    hipLaunchKernelGGL((rolling_zscore_kernel<float>), grid, dim3(256), lds, st, x, y, (long long)T, window, zero_nans);
  return check_launch("rolling_zscore");
}

// in-place power-of-two FFT of `buf` (C, m2) complex128 with scratch `tmp`; result ends in `buf`
static int fft_pow2(tl::c64* buf, tl::c64* tmp, const tl::c64* tw, int C, int m2, int inverse, hipStream_t st) {
  tl::c64* in = buf;
  tl::c64* out = tmp;
  const long long total_half = (long long)C * (m2 / 2);
  for (int ns = 1; ns < m2; ns <<= 1) {
    hipLaunchKernelGGL(tl::fft_pass_kernel, dim3(sgrid(total_half)), dim3(256), 0, st, in, out, tw, total_half, m2, ns, inverse);
    tl::c64* t = in; in = out; out = t;
  }
  if (in != buf && hipMemcpyAsync(buf, in, sizeof(tl::c64) * (size_t)C * m2, hipMemcpyDeviceToDevice, st) != hipSuccess)
  {
    tl::set_error("fft_pow2: device copy failed");
    return -1;
  }
  return check_launch("fft_pow2");
}

extern "C" int tl_fft_resample(const void* x, int is_f64, void* y, int C, int64_t nx, int64_t num, const double* w1,
                               const double* bf1, const double* tw1, int m2a, const double* w2, const double* bf2,
                               const double* tw2, int m2b, double* work, void* stream) {
  TL_REQUIRE(x && y && w1 && bf1 && tw1 && w2 && bf2 && tw2 && work, "fft_resample: null pointer");
  TL_REQUIRE(C > 0 && nx > 1 && num > 0, "fft_resample: bad sizes");
  TL_REQUIRE(m2a >= 2 * nx - 1 && (m2a & (m2a - 1)) == 0 && m2b >= 2 * num - 1 && (m2b & (m2b - 1)) == 0,
             "fft_resample: m2a/m2b must be powers of two >= 2n-1");
  hipStream_t st = (hipStream_t)stream;
  const int mmax = m2a > m2b ? m2a : m2b;
  tl::c64* A = reinterpret_cast<tl::c64*>(work);                 // (C, mmax)
  tl::c64* T = A + (size_t)C * mmax;                              // scratch (C, mmax)
  tl::c64* A2 = T + (size_t)C * mmax;                             // (C, m2b)
  const tl::c64* W1 = reinterpret_cast<const tl::c64*>(w1);
  const tl::c64* W2 = reinterpret_cast<const tl::c64*>(w2);
  const long long ta = (long long)C * m2a, tb = (long long)C * m2b;
  if (is_f64)
    hipLaunchKernelGGL((tl::rs_prep_kernel<double>), dim3(sgrid(ta)), dim3(256), 0, st, x, W1, A, ta, (int)nx, m2a);
  else
    hipLaunchKernelGGL((tl::rs_prep_kernel<float>), dim3(sgrid(ta)), dim3(256), 0, st, x, W1, A, ta, (int)nx, m2a);
  int rc = fft_pow2(A, T, reinterpret_cast<const tl::c64*>(tw1), C, m2a, 0, st);
  if (rc) return rc;
  hipLaunchKernelGGL(tl::rs_cmul_kernel, dim3(sgrid(ta)), dim3(256), 0, st, A, reinterpret_cast<const tl::c64*>(bf1), ta, m2a);
  rc = fft_pow2(A, T, reinterpret_cast<const tl::c64*>(tw1), C, m2a, 1, st);
  if (rc) return rc;
  hipLaunchKernelGGL(tl::rs_spec_kernel, dim3(sgrid(tb)), dim3(256), 0, st, A, W1, W2, A2, tb, (int)nx, (int)num, m2a, m2b);
  rc = fft_pow2(A2, T, reinterpret_cast<const tl::c64*>(tw2), C, m2b, 0, st);
  if (rc) return rc;
  hipLaunchKernelGGL(tl::rs_cmul_kernel, dim3(sgrid(tb)), dim3(256), 0, st, A2, reinterpret_cast<const tl::c64*>(bf2), tb, m2b);
  rc = fft_pow2(A2, T, reinterpret_cast<const tl::c64*>(tw2), C, m2b, 1, st);
  if (rc) return rc;
  const long long to = (long long)C * num;
  const double scale = 1.0 / ((double)m2b * (double)nx);
  if (is_f64)
    hipLaunchKernelGGL((tl::rs_out_kernel<double>), dim3(sgrid(to)), dim3(256), 0, st, A2, W2, (double*)y, to, (int)num, m2b, scale);
  else
    hipLaunchKernelGGL((tl::rs_out_kernel<float>), dim3(sgrid(to)), dim3(256), 0, st, A2, W2, (float*)y, to, (int)num, m2b, scale);
  return check_launch("fft_resample");
}

extern "C" int tl_hilbert_fft(const void* x, int is_f64, double* y, int C, int64_t T, const double* kernels, int nb,
                              const double* w, const double* bf, const double* tw, int m2, int envelope, double* work,
                              void* stream) {
  TL_REQUIRE(x && y && kernels && w && bf && tw && work, "hilbert_fft: null pointer");
  TL_REQUIRE(C > 0 && T > 1 && nb > 0, "hilbert_fft: bad sizes");
  TL_REQUIRE(m2 >= 2 * T - 1 && (m2 & (m2 - 1)) == 0, "hilbert_fft: m2 must be a power of two >= 2T-1");
  TL_REQUIRE(T < (1LL << 30), "hilbert_fft: recording too long");
  hipStream_t st = (hipStream_t)stream;
  tl::c64* A = reinterpret_cast<tl::c64*>(work);                 // (C, m2)
  tl::c64* Tm = A + (size_t)C * m2;                              // scratch (C, m2)
  tl::c64* X = Tm + (size_t)C * m2;                              // (C, T) spectrum
  const tl::c64* W = reinterpret_cast<const tl::c64*>(w);
  const tl::c64* BF = reinterpret_cast<const tl::c64*>(bf);
  const tl::c64* TW = reinterpret_cast<const tl::c64*>(tw);
  const long long ta = (long long)C * m2, tx = (long long)C * T;
  const int n = (int)T;
  if (is_f64)
    hipLaunchKernelGGL((tl::rs_prep_kernel<double>), dim3(sgrid(ta)), dim3(256), 0, st, x, W, A, ta, n, m2);
  else
    hipLaunchKernelGGL((tl::rs_prep_kernel<float>), dim3(sgrid(ta)), dim3(256), 0, st, x, W, A, ta, n, m2);
  int rc = fft_pow2(A, Tm, TW, C, m2, 0, st);
  if (rc) return rc;
  hipLaunchKernelGGL(tl::rs_cmul_kernel, dim3(sgrid(ta)), dim3(256), 0, st, A, BF, ta, m2);
  rc = fft_pow2(A, Tm, TW, C, m2, 1, st);
  if (rc) return rc;
  hipLaunchKernelGGL(tl::hb_spec_kernel, dim3(sgrid(tx)), dim3(256), 0, st, A, W, X, tx, n, m2);
  const double scale = 1.0 / ((double)m2 * (double)T * (double)nb);
  for (int b = 0; b < nb; ++b) {
    hipLaunchKernelGGL(tl::hb_band_prep_kernel, dim3(sgrid(ta)), dim3(256), 0, st, X, kernels + (size_t)b * T, W, A, ta, n, m2);
    rc = fft_pow2(A, Tm, TW, C, m2, 0, st);
    if (rc) return rc;
    hipLaunchKernelGGL(tl::rs_cmul_kernel, dim3(sgrid(ta)), dim3(256), 0, st, A, BF, ta, m2);
    rc = fft_pow2(A, Tm, TW, C, m2, 1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(tl::hb_accum_kernel, dim3(sgrid(tx)), dim3(256), 0, st, A, W, y, tx, n, m2, scale, envelope, b == 0 ? 1 : 0);
  }
  return check_launch("hilbert_fft");
}

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

template <typename TIN>
__device__ __forceinline__ double ext_sample(const void* x, long long cbase, long long T, int edge, long long i) {
#pragma clang fp contract(off)
  // odd extension of scipy.signal.filtfilt (padtype='odd')
  if (i < edge) return __dsub_rn(__dmul_rn(2.0, ld_as_f64<TIN>(x, cbase)), ld_as_f64<TIN>(x, cbase + (edge - i)));
  if (i >= edge + T)
    return __dsub_rn(__dmul_rn(2.0, ld_as_f64<TIN>(x, cbase + T - 1)), ld_as_f64<TIN>(x, cbase + T - 2 - (i - edge - T)));
  return ld_as_f64<TIN>(x, cbase + (i - edge));
}

template <typename TIN>
__global__ __launch_bounds__(64) void filtfilt_kernel(const void* __restrict__ x, const double* __restrict__ b,
                                                      const double* __restrict__ a, const double* __restrict__ zi,
                                                      double* __restrict__ y, double* __restrict__ work, int C,
                                                      long long T, int ntaps) {
#pragma clang fp contract(off)   // HIP's __dadd_rn/__dmul_rn are plain + and *: keep them un-fused
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int edge = 3 * ntaps;
  const long long next = T + 2LL * edge;
  double bb[MAX_TAPS], aa[MAX_TAPS], z[MAX_TAPS];
#pragma unroll
  for (int k = 0; k < MAX_TAPS; ++k) {
    bb[k] = k < ntaps ? b[k] : 0.0;
    aa[k] = k < ntaps ? a[k] : 0.0;
  }
  const long long cb = (long long)c * T;
  double* w = work + (long long)c * next;
  // forward pass over the odd-extended signal
  {
    const double x0 = ext_sample<TIN>(x, cb, T, edge, 0);
#pragma unroll
    for (int k = 0; k < MAX_TAPS; ++k) z[k] = (k < ntaps - 1) ? __dmul_rn(zi[k], x0) : 0.0;
    for (long long i = 0; i < next; ++i) {
      const double xv = ext_sample<TIN>(x, cb, T, edge, i);
      // same operation order and roundings as scipy's lfilter C loop (no FMA contraction):
      // y = z0 + b0*x;  z_k = (z_{k+1} + x*b_{k+1}) - y*a_{k+1}
      const double yv = __dadd_rn(z[0], __dmul_rn(bb[0], xv));
#pragma unroll
      for (int k = 0; k < MAX_TAPS - 1; ++k)
        z[k] = __dsub_rn(__dadd_rn(z[k + 1], __dmul_rn(xv, bb[k + 1])), __dmul_rn(yv, aa[k + 1]));
      w[i] = yv;
    }
  }
  // backward pass
  {
    const double x0 = w[next - 1];
#pragma unroll
    for (int k = 0; k < MAX_TAPS; ++k) z[k] = (k < ntaps - 1) ? __dmul_rn(zi[k], x0) : 0.0;
    for (long long i = next - 1; i >= 0; --i) {
      const double xv = w[i];
      // same operation order and roundings as scipy's lfilter C loop (no FMA contraction):
      // y = z0 + b0*x;  z_k = (z_{k+1} + x*b_{k+1}) - y*a_{k+1}
      const double yv = __dadd_rn(z[0], __dmul_rn(bb[0], xv));
#pragma unroll
      for (int k = 0; k < MAX_TAPS - 1; ++k)
        z[k] = __dsub_rn(__dadd_rn(z[k + 1], __dmul_rn(xv, bb[k + 1])), __dmul_rn(yv, aa[k + 1]));
      if (i >= edge && i < edge + T) y[cb + (i - edge)] = yv;
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

extern "C" int tl_filtfilt_f64(const void* x, int x_is_f64, const double* b, const double* a, const double* zi,
                               double* y, double* work, int C, int64_t T, int ntaps, void* stream) {
  TL_REQUIRE(x && b && a && zi && y && work, "filtfilt: null pointer");
  TL_REQUIRE(ntaps >= 2 && ntaps <= MAX_TAPS, "filtfilt: ntaps must be 2..%d", MAX_TAPS);
  TL_REQUIRE(C > 0 && T > 3LL * ntaps, "filtfilt: the input must be longer than padlen = %d", 3 * ntaps);
  dim3 grid((unsigned)((C + 63) / 64));
  hipStream_t st = (hipStream_t)stream;
  if (x_is_f64)
    hipLaunchKernelGGL((filtfilt_kernel<double>), grid, dim3(64), 0, st, x, b, a, zi, y, work, C, (long long)T, ntaps);
  else
    hipLaunchKernelGGL((filtfilt_kernel<float>), grid, dim3(64), 0, st, x, b, a, zi, y, work, C, (long long)T, ntaps);
  return check_launch("filtfilt");
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

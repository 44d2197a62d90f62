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

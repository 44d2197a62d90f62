// What limits a loop of independent fp32 MFMAs (32x32x2) below the 157.3 TFLOP/s peak on gfx950?  Register-resident
// operands ("fresh" A and B registers for every MFMA, the Winograd pattern), random / zero / sign-constant data,
// 4 / 6 / 8 accumulators, one or two waves per SIMD, two MFMA orders.  Round 2 measured 146 TFLOP/s for six accumulators
// and 155 for eight on random data; this separates accumulator count, residency, order and data.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_pattern.hip -o /tmp/mfma_pattern && /tmp/mfma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ORDER 0: q outer, accumulator inner (the product kernels); 1: accumulator outer, q inner (4 dependent MFMAs in a row)
template <int NACC, int ORDER, int UNROLL>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 256) lds[i] = src[(blockIdx.x * 8192 + i) & ((1 << 22) - 1)];
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f32x4 a[NACC], b[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    a[i] = *reinterpret_cast<const f32x4*>(lds + ((tid * 4 + i * 1024) & 8188));
    b[i] = *reinterpret_cast<const f32x4*>(lds + ((tid * 4 + i * 1024 + 512) & 8188));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (ORDER == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[i][q], acc[i], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[i][q], acc[i], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + tid] = s;
}

template <int NACC, int ORDER, int UNROLL>
void run(const char* name, const float* src, float* out, int wps) {
  // wps waves per SIMD: 2 -> two 256-thread workgroups per CU (34 KB of LDS each), 1 -> one (100 KB each)
  const int nwg = 256 * wps * 4, iters = 6000 / (NACC * UNROLL) * 4;
  const size_t lds = wps == 2 ? 34 * 1024 : 100 * 1024;
  auto kern = k<NACC, ORDER, UNROLL>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, src, out, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double fl = (double)nwg * 4 * iters * UNROLL * 4 * NACC * 4096.0;
  printf("  %-54s %d wave/SIMD %8.3f ms  %7.2f TFLOP/s\n", name, wps, ms, fl / ms / 1e9);
  fflush(stdout);
}

int main() {
  float *src, *out;
  hipMalloc(&src, (1 << 22) * sizeof(float));
  hipMalloc(&out, 2048 * 256 * sizeof(float));
  std::vector<float> h(1 << 22);
  for (int pat = 0; pat < 4; ++pat) {
    unsigned x = 12345u;
    for (auto& v : h) {
      x = x * 1664525u + 1013904223u;
      const float r = ((int)(x >> 9) - (1 << 22)) * (1.0f / (1 << 22));          // uniform [-1, 1)
      v = pat == 0 ? r : pat == 1 ? 0.f : pat == 2 ? fabsf(r) : (float)((int)(r * 8.f));   // random / zero / positive / small integers
    }
    hipMemcpy(src, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    printf("data: %s\n", pat == 0 ? "uniform [-1,1)" : pat == 1 ? "zeros" : pat == 2 ? "uniform [0,1)" : "integers -8..7");
    for (int wps = 2; wps >= 1; --wps) {
      run<4, 0, 1>("4 acc, q outer", src, out, wps);
      run<6, 0, 1>("6 acc, q outer", src, out, wps);
      run<6, 0, 2>("6 acc, q outer, body x 2", src, out, wps);
      run<6, 1, 1>("6 acc, accumulator outer (4 dependent in a row)", src, out, wps);
      run<8, 0, 1>("8 acc, q outer", src, out, wps);
      run<8, 1, 1>("8 acc, accumulator outer", src, out, wps);
      if (pat > 0) break;
    }
  }
  return 0;
}

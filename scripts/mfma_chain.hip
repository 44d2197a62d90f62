// How fast can one SIMD issue fp32 MFMAs (32x32x2) when every MFMA takes FRESH A and B operand registers
// (the Winograd F(4,3) pattern: six independent products V_i . U_i per k-slice, no operand shared between
// MFMAs) compared with the direct-GEMM pattern (2 x 2 accumulators, every operand feeds two MFMAs)?
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
// Operands live in registers (random values, re-randomised cheaply so nothing is hoisted); 256-thread
// workgroups, two per CU (2 waves per SIMD), whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int UNIQUE, int FROM_LDS>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
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
    if (FROM_LDS) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        a[i] = *reinterpret_cast<const f32x4*>(lds + ((tid * 4 + i * 1024 + it * 36) & 8188));
        b[i] = *reinterpret_cast<const f32x4*>(lds + ((tid * 4 + i * 1024 + 512 + it * 36) & 8188));
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (UNIQUE) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[i][q], acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i >> 1][q], b[i & 1][q], acc[i], 0, 0, 0);   // 2 x (NACC/2) reuse
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + tid] = s;
}

template <int NACC, int UNIQUE, int FROM_LDS>
void run(const char* name, const float* src, float* out) {
  const int nwg = 512 * 8, iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, UNIQUE, FROM_LDS>), dim3(nwg), dim3(256), 0, 0, src, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NACC, UNIQUE, FROM_LDS>), dim3(nwg), dim3(256), 0, 0, src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double fl = (double)nwg * 4 * iters * 4 * NACC * 4096.0;
  printf("%-58s %8.3f ms  %7.2f TFLOP/s\n", name, ms, fl / ms / 1e9);
}

int main() {
  float *src, *out;
  hipMalloc(&src, (1 << 22) * sizeof(float));
  hipMalloc(&out, 512 * 8 * 256 * sizeof(float));
  std::vector<float> h(1 << 22);
  unsigned x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
  hipMemcpy(src, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  run<4, 0, 0>("4 acc, shared operands (2x2), registers", src, out);
  run<4, 1, 0>("4 acc, unique operands, registers", src, out);
  run<6, 1, 0>("6 acc, unique operands, registers", src, out);
  run<8, 0, 0>("8 acc, shared operands (4x2), registers", src, out);
  run<8, 1, 0>("8 acc, unique operands, registers", src, out);
  run<4, 0, 1>("4 acc, shared operands, ds_read_b128 per k-group", src, out);
  run<6, 1, 1>("6 acc, unique operands, ds_read_b128 per k-group", src, out);
  run<8, 1, 1>("8 acc, unique operands, ds_read_b128 per k-group", src, out);
  return 0;
}

// Which fp32 MFMA shape sustains more FLOP/s on random data at the Winograd wave tile (32 quads x 32 columns x 6
// transforms = 96 accumulator registers, operands re-read from LDS by ds_read_b128, two waves per SIMD)?
//   v_mfma_f32_32x32x2_f32 : 64 cycles, 2 048 MACs           v_mfma_f32_16x16x4_f32 : 32 cycles, 1 024 MACs
// Same cycles per FLOP and - with the fragment layouts below - the same LDS read bytes per FLOP: per transform and
// 16-deep K chunk four ds_read_b128 feed 8 (32x32x2) or 16 (16x16x4) MFMAs.  MI355X_MICROARCH.md, DVFS give-back item 7
// reports that for bf16 the smaller shape holds a higher clock under load (+12-15 % FLOP/s at equal cycles); this asks
// the same question for fp32.  In-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LDS_FLOATS = 36864;   // 144 KB, as the product kernel declares

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* __restrict__ out,
                                            unsigned long long* __restrict__ stamps, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  const int tid = threadIdx.x;
  for (int i = tid; i < LDS_FLOATS; i += 512) lds[i] = src[(blockIdx.x * 4099 + i) & ((1 << 22) - 1)];
  __syncthreads();
  const int lane = tid & 63;
  // fragment addresses: rows of 64 B, one b128 per lane (the product layout; swizzle omitted - both arms read the same way)
  const int row32 = (lane & 31) * 16 + (lane >> 5) * 4;          // 32x32x2: row = lane % 32, k-quad = lane / 32 (+2 for g = 1)
  const int row16 = (lane & 15) * 16 + (lane >> 4) * 4;          // 16x16x4: row = lane % 16, k-quad = lane / 16
  unsigned long long t0 = 0, r0 = 0;
  if (SHAPE == 32) {
    f32x16 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
      const int base = (it * 1552) & 8191;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        f32x4 a[6], b[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          a[i] = *reinterpret_cast<const f32x4*>(lds + base + i * 2048 + row32 + g * 8);
          b[i] = *reinterpret_cast<const f32x4*>(lds + 18432 + base + i * 1024 + row32 + g * 8);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[i][q], acc[i], 0, 0, 0);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 512 + tid] = s;
  } else {
    f32x4 acc[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
      const int base = (it * 1552) & 8191;
#pragma unroll
      for (int h = 0; h < 2; ++h) {                 // three transforms at a time: 12 reads, 48 MFMAs, like a k-group of the 32-wide arm
        f32x4 al[3], ah[3], bl[3], bh[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int t = 3 * h + i;
          al[i] = *reinterpret_cast<const f32x4*>(lds + base + t * 2048 + row16);
          ah[i] = *reinterpret_cast<const f32x4*>(lds + base + t * 2048 + 256 + row16);
          bl[i] = *reinterpret_cast<const f32x4*>(lds + 18432 + base + t * 1024 + row16);
          bh[i] = *reinterpret_cast<const f32x4*>(lds + 18432 + base + t * 1024 + 256 + row16);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int t = 3 * h + i;
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[i][q], bl[i][q], acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[i][q], bh[i][q], acc[t][1], 0, 0, 0);
            acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i][q], bl[i][q], acc[t][2], 0, 0, 0);
            acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i][q], bh[i][q], acc[t][3], 0, 0, 0);
          }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += acc[i][e][0] + acc[i][e][1] + acc[i][e][2] + acc[i][e][3];
    out[blockIdx.x * 512 + tid] = s;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
double run(const char* name, const float* src, float* out, unsigned long long* stamps, int nwg, int iters, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<SHAPE>), dim3(nwg), dim3(512), 0, 0, src, out, stamps, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> h(2 * nwg);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk(nwg);
  for (int i = 0; i < nwg; ++i) clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;   // GHz
  std::sort(clk.begin(), clk.end());
  const double fl = (double)nwg * 8 * iters * 96.0 * 4096.0;   // per wave and iteration 96 MFMA-equivalents of 2 048 MACs
  printf("%-28s %8.3f ms  %7.2f TFLOP/s  clock %.3f GHz (median of %d workgroups)\n", name, ms, fl / ms / 1e9, clk[nwg / 2], nwg);
  return fl / ms / 1e9;
}

int main() {
  float *src, *out;
  unsigned long long* stamps;
  const int nwg = 256;
  hipMalloc(&src, (1 << 22) * sizeof(float));
  hipMalloc(&out, nwg * 512 * sizeof(float));
  hipMalloc(&stamps, 2 * nwg * 8);
  std::vector<float> h(1 << 22);
  unsigned x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
  hipMemcpy(src, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  const int iters = 4000;      // 4000 x 96 x 64 cycles x 2 waves / SIMD = 49 M cycles = ~21 ms per launch
  // sustained load first (the clock settles), then interleaved rounds
  for (int r = 0; r < 2; ++r) { run<32>("warm 32x32x2", src, out, stamps, nwg, iters, 20); run<16>("warm 16x16x4", src, out, stamps, nwg, iters, 20); }
  for (int r = 0; r < 4; ++r) {
    run<32>("32x32x2  (96 acc, LDS reads)", src, out, stamps, nwg, iters, 20);
    run<16>("16x16x4  (96 acc, LDS reads)", src, out, stamps, nwg, iters, 20);
  }
  return 0;
}

// Feasibility probe: the F(4,3) NT main loop with the six Winograd products on the bf16 matrix cores,
// fp32 operands split in three bf16 terms (6 cross products per fp32 product, fp32 accumulate).
//   hipcc -O3 --offload-arch=gfx950 scripts/bf16x6_probe.hip -o /tmp/bf16x6_probe && /tmp/bf16x6_probe
// One K-step (16 channels) of a wave tile of 32 quads x 32 columns x 6 transforms:
//   VAR 0  36 bf16 MFMAs only (operands in registers)
//   VAR 1  + LDS fragment reads (12 x b128 input rows fp32, 18 x b128 pre-split bf16 weights)
//   VAR 2  + input transform (B^T d) on the VALU
//   VAR 3  + three-way bf16 split of the 48 transformed values per lane
//   VAR 4  + barrier per K-step
// Reports algorithmic (direct-form) TFLOP/s = 2*128 rows*32 cols*16 ch*3 taps per wave K-step; the fp32
// F(4,3) kernel of the repo runs conv2 forward at 209 algorithmic TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int LD = 20, QR = 129, PLANE = QR * LD, ABUF = 4 * PLANE;   // fp32 input rows, 4 planes (row mod 4)
constexpr int BN = 64, UB = 6 * 3 * BN * 8;                            // weights: [6][3 splits][64 cols][8 dwords]

__device__ inline uint32_t pk_hi(float a, float b) {   // truncating split term of two floats -> packed bf16
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}

template <int VAR, int WGS>
__global__ __launch_bounds__(512, WGS) void k(const float* __restrict__ A, float* __restrict__ out, int nsteps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                                   // [2][ABUF]
  uint32_t* Us = reinterpret_cast<uint32_t*>(lds + 2 * ABUF);   // [2][UB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 2 * ABUF + 2 * UB; i += 512) lds[i] = A[(long long)(blockIdx.x & 1023) * 4096 + (i & 4095)];
  __syncthreads();
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  u32x4 vs[6][3], us[6][3];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) {
    vs[i][j] = u32x4{0x3f803f80u + i, 0x3f003f00u + j, 0x3e803e80u, 0x3f803f00u};
    us[i][j] = u32x4{0x3f803f80u + j, 0x3f003f00u + i, 0x3e803e80u, 0x3f803f00u};
  }
  const int a_lane = (wm * 32 + lr) * LD + lh * 4;
  const int u_lane = ((wn * 32 + lr) * 2 + lh) * 4;            // dwords: [col][lh][4]
  auto rd = [&](const float* p) { return *reinterpret_cast<const f32x4*>(p); };
  auto split = [&](const f32x4& x, const f32x4& y, u32x4& hi, u32x4& mid, u32x4& lo) {
    // values x[0..3] (k-group 0), y[0..3] (k-group 1): element j of the fragment = x[j], 4 + j = y[j]
    float r[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    uint32_t H[4], M[4], L[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float a0 = r[2 * q], a1 = r[2 * q + 1];
      const float h0 = __uint_as_float(__float_as_uint(a0) & 0xffff0000u), h1 = __uint_as_float(__float_as_uint(a1) & 0xffff0000u);
      const float r0 = a0 - h0, r1 = a1 - h1;
      const float m0 = __uint_as_float(__float_as_uint(r0) & 0xffff0000u), m1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
      const float s0 = r0 - m0, s1 = r1 - m1;
      H[q] = pk_hi(h0, h1); M[q] = pk_hi(m0, m1); L[q] = pk_hi(s0, s1);
    }
    hi = u32x4{H[0], H[1], H[2], H[3]}; mid = u32x4{M[0], M[1], M[2], M[3]}; lo = u32x4{L[0], L[1], L[2], L[3]};
  };
  auto mf = [&](const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    const float* a_s = As + buf * ABUF + a_lane;
    const uint32_t* u_s = Us + buf * UB + u_lane;
    if (VAR >= 1) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) us[i][j] = *reinterpret_cast<const u32x4*>(u_s + (i * 3 + j) * BN * 8);
      f32x4 d[6][2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        d[0][g] = rd(a_s + 8 * g); d[1][g] = rd(a_s + PLANE + 8 * g); d[2][g] = rd(a_s + 2 * PLANE + 8 * g);
        d[3][g] = rd(a_s + 3 * PLANE + 8 * g); d[4][g] = rd(a_s + LD + 8 * g); d[5][g] = rd(a_s + PLANE + LD + 8 * g);
      }
      if (VAR >= 2) {
        f32x4 v[6][2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const f32x4 d0 = d[0][g], d1 = d[1][g], d2 = d[2][g], d3 = d[3][g], d4 = d[4][g], d5 = d[5][g];
          const f32x4 s1 = d4 - 4.f * d2, s2 = d3 - 4.f * d1, s3 = d4 - d2, t = d3 - d1;
          v[1][g] = s1 + s2; v[2][g] = s1 - s2; v[3][g] = s3 + 2.f * t; v[4][g] = s3 - 2.f * t;
          v[0][g] = 4.f * d0 + (d4 - 5.f * d2); v[5][g] = (4.f * d1 - 5.f * d3) + d5;
        }
        if (VAR >= 3) {
#pragma unroll
          for (int i = 0; i < 6; ++i) split(v[i][0], v[i][1], vs[i][0], vs[i][1], vs[i][2]);
        } else {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            vs[i][0] = __builtin_bit_cast(u32x4, v[i][0]); vs[i][1] = __builtin_bit_cast(u32x4, v[i][1]);
            vs[i][2] = __builtin_bit_cast(u32x4, v[i][0] + v[i][1]);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          vs[i][0] = __builtin_bit_cast(u32x4, d[i][0]); vs[i][1] = __builtin_bit_cast(u32x4, d[i][1]);
          vs[i][2] = __builtin_bit_cast(u32x4, d[(i + 1) % 6][0]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {                         // small terms first
      mf(vs[i][2], us[i][0], acc[i]); mf(vs[i][0], us[i][2], acc[i]); mf(vs[i][1], us[i][1], acc[i]);
      mf(vs[i][1], us[i][0], acc[i]); mf(vs[i][0], us[i][1], acc[i]); mf(vs[i][0], us[i][0], acc[i]);
    }
    if (VAR >= 4) __syncthreads();
  }
  float sum = 0.f;
  for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) sum += acc[i][e];
  out[(long long)blockIdx.x * 512 + tid] = sum;
}

template <int VAR, int WGS>
void run(const char* name, const float* A, float* out, int nwg, int nsteps) {
  const size_t shm = (2 * ABUF + 2 * UB) * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<VAR, WGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VAR, WGS>), dim3(nwg), dim3(512), shm, 0, A, out, nsteps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<VAR, WGS>), dim3(nwg), dim3(512), shm, 0, A, out, nsteps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double alg = (double)nwg * 8 * nsteps * (2.0 * 128 * 32 * 16 * 3);
  const double issued = (double)nwg * 8 * nsteps * 36 * 32768.0;
  printf("%-52s %8.3f ms  algorithmic %7.1f TFLOP/s  bf16 issued %7.1f TFLOP/s (%s)\n", name, ms, alg / ms / 1e9,
         issued / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
  const int nwg = 256 * 24, nsteps = 64;
  float *A, *out;
  const size_t na = (size_t)1024 * 4096 + 4096;
  hipMalloc(&A, na * sizeof(float));
  hipMalloc(&out, (size_t)nwg * 512 * sizeof(float));
  std::vector<float> h(na);
  unsigned x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
  hipMemcpy(A, h.data(), na * sizeof(float), hipMemcpyHostToDevice);
  printf("LDS per workgroup: %.1f KB\n", (2 * ABUF + 2 * UB) * 4 / 1024.0);
  run<0, 1>("V0 36 bf16 MFMAs / K-step only", A, out, nwg, nsteps);
  run<1, 1>("V1 + 30 ds_read_b128", A, out, nwg, nsteps);
  run<2, 1>("V2 + input transform", A, out, nwg, nsteps);
  run<3, 1>("V3 + 3-way bf16 split (trunc)", A, out, nwg, nsteps);
  run<4, 1>("V4 + barrier per K-step (1 WG/CU)", A, out, nwg, nsteps);
  return 0;
}

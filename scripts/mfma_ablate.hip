// Ablation of the fp32-MFMA GEMM main loop on gfx950: which ingredient costs the matrix pipe?
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_ablate.hip -o /tmp/mfma_ablate && /tmp/mfma_ablate
// Each variant runs the same 64 MFMAs (32x32x2 f32, 4 accumulators) per "K-step" per wave, 256-thread
// workgroups, 2 workgroups per CU, and reports TFLOP/s of the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LDS_LD = 36;
constexpr int AROWS = 130;

template <int VAR, int LOADS = (VAR >= 3), int STORES = (VAR >= 3)>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out,
                                            int nsteps, int lda) {
  __shared__ __attribute__((aligned(16))) float lds[2 * AROWS * LDS_LD + 2 * 128 * LDS_LD];
  float* As = lds;
  float* Bs = lds + 2 * AROWS * LDS_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // fill LDS once
  for (int i = tid; i < 2 * AROWS * LDS_LD + 2 * 128 * LDS_LD; i += 256) lds[i] = A[(long long)blockIdx.x * 4096 + (i & 4095)];
  __syncthreads();
  f32x4 fa[2], fb[2];
  fa[0] = fa[1] = fb[0] = fb[1] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
  f32x4 ra[5], rb[4];
  const long long R0 = (long long)blockIdx.x * 128;
  for (int s = 0; s < nsteps; ++s) {
    if (LOADS) {   // global loads for the next step (register staged)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256, r = idx >> 3, c4 = idx & 7;
        rb[i] = *reinterpret_cast<const f32x4*>(B + (long long)r * 512 + ((s * 32) & 511) + c4 * 4);
      }
      if ((s % 3) == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          int idx = tid + i * 256; if (idx > 1039) idx = 1039;
          const int r = idx >> 3, c4 = idx & 7;
          ra[i] = *reinterpret_cast<const f32x4*>(A + (R0 + r) * (long long)lda + ((s * 32) & 511) + c4 * 4);
        }
      }
    }
    const float* a_s = As + (s & 1) * AROWS * LDS_LD + (wm * 64 + lr + (s % 3)) * LDS_LD + lh * 4;
    const float* b_s = Bs + (s & 1) * 128 * LDS_LD + (wn * 64 + lr) * LDS_LD + lh * 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (VAR >= 1) {
        fa[0] = *reinterpret_cast<const f32x4*>(a_s + kk * 8);
        fa[1] = *reinterpret_cast<const f32x4*>(a_s + 32 * LDS_LD + kk * 8);
        fb[0] = *reinterpret_cast<const f32x4*>(b_s + kk * 8);
        fb[1] = *reinterpret_cast<const f32x4*>(b_s + 32 * LDS_LD + kk * 8);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][q], fb[ni][q], acc[mi][ni], 0, 0, 0);
    }
    if (LOADS && !STORES) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(rb[i]));
      if ((s % 3) == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) asm volatile("" ::"v"(ra[i]));
      }
    }
    if (STORES) {
      if (!LOADS) { for (int i = 0; i < 4; ++i) rb[i] = fa[0]; for (int i = 0; i < 5; ++i) ra[i] = fb[0]; }
      float* db = Bs + ((s + 1) & 1) * 128 * LDS_LD;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256, r = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<f32x4*>(db + r * LDS_LD + c4 * 4) = rb[i];
      }
      if ((s % 3) == 0) {
        float* da = As + ((s + 1) & 1) * AROWS * LDS_LD;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          int idx = tid + i * 256; if (idx > 1039) idx = 1039;
          const int r = idx >> 3, c4 = idx & 7;
          *reinterpret_cast<f32x4*>(da + r * LDS_LD + c4 * 4) = ra[i];
        }
      }
    }
    if (VAR >= 2) __syncthreads();
  }
  float sum = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
  out[(long long)blockIdx.x * 256 + tid] = sum;
}


// prefetch distance 2: registers set P (even steps) / Q (odd steps); loads issued at step s are stored to LDS
// at the end of step s+1 (a full extra K-step of latency tolerance)
__global__ __launch_bounds__(256, 2) void kd2(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out,
                                              int nsteps, int lda) {
  __shared__ __attribute__((aligned(16))) float lds[2 * AROWS * LDS_LD + 2 * 128 * LDS_LD];
  float* As = lds;
  float* Bs = lds + 2 * AROWS * LDS_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  for (int i = tid; i < 2 * AROWS * LDS_LD + 2 * 128 * LDS_LD; i += 256) lds[i] = A[(long long)blockIdx.x * 4096 + (i & 4095)];
  __syncthreads();
  f32x4 fa[2], fb[2];
  f32x4 rbP[4], rbQ[4];
  for (int i = 0; i < 4; ++i) rbP[i] = rbQ[i] = f32x4{0, 0, 0, 0};
  auto loadB = [&](f32x4 (&rb)[4], int s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256, r = idx >> 3, c4 = idx & 7;
      rb[i] = *reinterpret_cast<const f32x4*>(B + (long long)r * 512 + ((s * 32) & 511) + c4 * 4);
    }
  };
  auto storeB = [&](const f32x4 (&rb)[4], int s) {
    float* db = Bs + ((s + 1) & 1) * 128 * LDS_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256, r = idx >> 3, c4 = idx & 7;
      *reinterpret_cast<f32x4*>(db + r * LDS_LD + c4 * 4) = rb[i];
    }
  };
  auto compute = [&](int s) {
    const float* a_s = As + (s & 1) * AROWS * LDS_LD + (wm * 64 + lr + (s % 3)) * LDS_LD + lh * 4;
    const float* b_s = Bs + (s & 1) * 128 * LDS_LD + (wn * 64 + lr) * LDS_LD + lh * 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      fa[0] = *reinterpret_cast<const f32x4*>(a_s + kk * 8);
      fa[1] = *reinterpret_cast<const f32x4*>(a_s + 32 * LDS_LD + kk * 8);
      fb[0] = *reinterpret_cast<const f32x4*>(b_s + kk * 8);
      fb[1] = *reinterpret_cast<const f32x4*>(b_s + 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][q], fb[ni][q], acc[mi][ni], 0, 0, 0);
    }
  };
  for (int s = 0; s < nsteps; s += 2) {
    loadB(rbP, s + 2);
    compute(s);
    storeB(rbQ, s);          // loaded one step ago
    __syncthreads();
    loadB(rbQ, s + 3);
    compute(s + 1);
    storeB(rbP, s + 1);
    __syncthreads();
  }
  float sum = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
  out[(long long)blockIdx.x * 256 + tid] = sum;
}

template <int VAR, int LOADS = (VAR >= 3), int STORES = (VAR >= 3)>
void run(const char* name, const float* A, const float* B, float* out, int nwg, int nsteps, int lda) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VAR, LOADS, STORES>), dim3(nwg), dim3(256), 0, 0, A, B, out, nsteps, lda);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<VAR, LOADS, STORES>), dim3(nwg), dim3(256), 0, 0, A, B, out, nsteps, lda);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double fl = (double)nwg * 4 * nsteps * 64 * 4096.0;
  printf("%-44s %8.3f ms  %7.2f TFLOP/s\n", name, ms, fl / ms / 1e9);
}

int main() {
  const int nwg = 512 * 40, nsteps = 48, lda = 512;
  const long long arows = (long long)nwg * 128 + 256;
  float *A, *B, *out;
  hipMalloc(&A, arows * lda * sizeof(float));
  hipMalloc(&B, 512 * 512 * 3 * sizeof(float));
  hipMalloc(&out, (size_t)nwg * 256 * sizeof(float));
  {  // random (not zero) operands: zero data inflates MFMA clocks (DVFS), cdna guide rule 25
    std::vector<float> h(1 << 24);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
    for (long long off = 0; off < arows * lda; off += (1 << 24)) {
      long long n = arows * lda - off; if (n > (1 << 24)) n = 1 << 24;
      hipMemcpy(A + off, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    }
    hipMemcpy(B, h.data(), 512 * 512 * 3 * sizeof(float), hipMemcpyHostToDevice);
  }
  run<0>("V0 MFMA only (operands in registers)", A, B, out, nwg, nsteps, lda);
  run<1>("V1 + ds_read_b128 fragments", A, B, out, nwg, nsteps, lda);
  run<2>("V2 + barrier per K-step", A, B, out, nwg, nsteps, lda);
  run<3>("V3 + global loads + LDS stores (full loop)", A, B, out, nwg, nsteps, lda);
  run<2, 1, 0>("V4 barrier + global loads only", A, B, out, nwg, nsteps, lda);
  run<2, 0, 1>("V5 barrier + LDS stores only", A, B, out, nwg, nsteps, lda);
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kd2, dim3(nwg), dim3(256), 0, 0, A, B, out, nsteps, lda); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kd2, dim3(nwg), dim3(256), 0, 0, A, B, out, nsteps, lda);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-44s %8.3f ms  %7.2f TFLOP/s\n", "V6 B only: loads dist-2 + stores", ms, (double)nwg * 4 * nsteps * 64 * 4096.0 / ms / 1e9);
  }
  return 0;
}

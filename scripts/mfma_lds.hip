// What do the fragment reads of the F(4,3) NT-V K-step cost the fp32 matrix pipe?  The product kernel's wave tile
// (32 quads x 32 columns x 6 transforms, 96 accumulators, 8 waves per workgroup, one workgroup per CU) with its LDS
// images and swizzle, no LDS-DMA, no barrier, no epilogue: per 16-deep K-step 24 ds_read_b128 + 48 MFMAs per wave.
// Register-resident operands run at 155 TFLOP/s on this part (scripts/mfma_pattern.hip); this measures what is lost
// to the reads alone, by how they are issued, and whether it scales with their number or their bytes.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BQ = 128, BN = 64, ROWB = 64;
constexpr int A_BYTES = 6 * BQ * ROWB, B_BYTES = 6 * BN * ROWB, STAGE = A_BYTES + B_BYTES;

// MODE 0: operands stay in registers (no reads)         1: 24 x b128 per step, compiler order
//      2: 24 x b128, one read per two MFMAs (product)   3: 12 x b128 per step (every fragment feeds two steps)
//      4: 48 x b64 per step, one read per MFMA          5: as 2 without the swizzle (bank conflicts)
//      6: as 2, reads of the NEXT k-group issued in the first half of the current group's MFMAs only
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* __restrict__ out, int nsteps) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  float* lf = reinterpret_cast<float*>(lds);
  for (int i = tid; i < 2 * STAGE / 4; i += 512) lf[i] = src[(blockIdx.x * 4099 + i) & ((1 << 22) - 1)];
  __syncthreads();
  const int sw = MODE == 5 ? 0 : (lr >> 2) & 3;
  const int a_row = (wm * 32 + lr) * ROWB, b_row = A_BYTES + (wn * 32 + lr) * ROWB;
  const int c_g0 = (lh ^ sw) << 4, c_g1 = ((2 + lh) ^ sw) << 4;
  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f32x4 fa0[6], fb0[6], fa1[6], fb1[6];
  auto load_frag = [&](f32x4 (&fa)[6], f32x4 (&fb)[6], int stage, int cg) {
    const char* a_s = lds + stage * STAGE + a_row + cg;
    const char* b_s = lds + stage * STAGE + b_row + cg;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (MODE == 4) {
        const f32x2 a0 = *reinterpret_cast<const f32x2*>(a_s + i * (BQ * ROWB)), a1 = *reinterpret_cast<const f32x2*>(a_s + i * (BQ * ROWB) + 8);
        const f32x2 b0 = *reinterpret_cast<const f32x2*>(b_s + i * (BN * ROWB)), b1 = *reinterpret_cast<const f32x2*>(b_s + i * (BN * ROWB) + 8);
        fa[i] = f32x4{a0[0], a0[1], a1[0], a1[1]};
        fb[i] = f32x4{b0[0], b0[1], b1[0], b1[1]};
      } else {
        fa[i] = *reinterpret_cast<const f32x4*>(a_s + i * (BQ * ROWB));
        fb[i] = *reinterpret_cast<const f32x4*>(b_s + i * (BN * ROWB));
      }
    }
  };
  auto mfma_group = [&](const f32x4 (&fa)[6], const f32x4 (&fb)[6]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q], fb[i][q], acc[i], 0, 0, 0);
  };
  load_frag(fa0, fb0, 0, c_g0);
  load_frag(fa1, fb1, 0, c_g1);
  for (int s = 0; s < nsteps; ++s) {
    const int stage = s & 1;
    if (MODE == 0) {
      mfma_group(fa1, fb1);
      mfma_group(fa0, fb0);
    } else if (MODE == 3) {
      if (s & 1) load_frag(fa0, fb0, stage, c_g0);
      mfma_group(fa1, fb1);
      if (!(s & 1)) load_frag(fa1, fb1, stage, c_g1);
      mfma_group(fa0, fb0);
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);
    } else {
      load_frag(fa0, fb0, stage, c_g0);
      mfma_group(fa1, fb1);                  // k-group 1 of the previous step
      load_frag(fa1, fb1, stage, c_g1);
      mfma_group(fa0, fb0);
      if (MODE == 2 || MODE == 5) {
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
#pragma unroll
        for (int t = 0; t < 12; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      } else if (MODE == 4) {
        __builtin_amdgcn_sched_group_barrier(0x100, 24, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
#pragma unroll
        for (int t = 0; t < 24; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      } else if (MODE == 6) {
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
#pragma unroll
        for (int t = 0; t < 12; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
      }
    }
  }
  mfma_group(fa1, fb1);
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) t += acc[i][e];
  out[blockIdx.x * 512 + tid] = t;
}

template <int MODE>
void run(const char* name, const float* src, float* out) {
  const int nwg = 256, nsteps = 2000, reps = 10;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE>), dim3(nwg), dim3(512), 0, 0, src, out, nsteps);
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE>), dim3(nwg), dim3(512), 0, 0, src, out, nsteps);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double fl = (double)nwg * 8 * (nsteps * 48.0 + 24.0) * 4096.0;
  printf("  %-62s %8.3f ms  %7.2f TFLOP/s\n", name, ms, fl / ms / 1e9);
  fflush(stdout);
}

int main() {
  float *src, *out;
  (void)hipMalloc(&src, (1 << 22) * sizeof(float));
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  std::vector<float> h(1 << 22);
  unsigned x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
  (void)hipMemcpy(src, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  for (int r = 0; r < 2; ++r) {
    run<0>("registers only", src, out);
    run<1>("24 ds_read_b128 per step, compiler order", src, out);
    run<2>("24 ds_read_b128, one per two MFMAs (product order)", src, out);
    run<6>("24 ds_read_b128, one per MFMA in the first half of a group", src, out);
    run<3>("12 ds_read_b128 per step", src, out);
    run<4>("48 ds_read_b64 per step, one per MFMA", src, out);
    run<5>("24 ds_read_b128, product order, no swizzle (conflicts)", src, out);
  }
  return 0;
}

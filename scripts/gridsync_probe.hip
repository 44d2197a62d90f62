// probe: latency of cooperative_groups grid.sync() on MI355X for several grid sizes
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(256) void sync_loop(float* buf, int iters) {
  cg::grid_group g = cg::this_grid();
  float v = 0.f;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    buf[gid] = v + 1.f;                   // a write every workgroup's successor phase reads
    g.sync();
    v = buf[(gid + 256) % (gridDim.x * blockDim.x)];
  }
  buf[gid] = v;
}

int main() {
  int dev = 0;
  hipSetDevice(dev);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, dev);
  printf("cooperativeLaunch=%d CUs=%d\n", p.cooperativeLaunch, p.multiProcessorCount);
  float* buf;
  hipMalloc(&buf, 1024 * 256 * sizeof(float));
  hipMemset(buf, 0, 1024 * 256 * sizeof(float));
  for (int nb : {32, 64, 128, 256, 512}) {
    for (int iters : {10, 210}) {
      void* args[] = {&buf, &iters};
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipError_t rc = hipLaunchCooperativeKernel((void*)sync_loop, dim3(nb), dim3(256), args, 0, 0);   // warm
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      rc = hipLaunchCooperativeKernel((void*)sync_loop, dim3(nb), dim3(256), args, 0, 0);
      hipEventRecord(e1, 0);
      hipError_t rc2 = hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("blocks %4d iters %4d: rc %d/%d  %.1f us total\n", nb, iters, (int)rc, (int)rc2, ms * 1e3);
    }
  }
  float h[4];
  hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
  printf("check %g (expect 210)\n", h[0]);
  return 0;
}

// Wave / workgroup dispatch rate of the device: duration of kernels that do (almost) nothing, as a function of the
// number of workgroups and the workgroup size.  hipcc --offload-arch=gfx950 -O2 -o wave_rate tools/wave_rate.hip
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void k_empty(int* p) {
  if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1;
}
__global__ void k_touch(float* p, int n) {  // one 16-byte load+store per thread: "tiny work"
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float4* q = reinterpret_cast<float4*>(p) + i % (size_t)n;
  float4 v = *q;
  v.x += 1.0f;
  *q = v;
}

int main() {
  float* buf;
  const int n = 1 << 22;
  hipMalloc(&buf, sizeof(float4) * n);
  hipMemset(buf, 0, sizeof(float4) * n);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks[] = {64, 256, 1024};
  const int grids[] = {256, 1024, 2048, 4096, 8192, 16384, 32768};
  for (int which = 0; which < 2; ++which)
    for (int bs : blocks)
      for (int g : grids) {
        for (int w = 0; w < 5; ++w) {
          if (which == 0) hipLaunchKernelGGL(k_empty, dim3(g), dim3(bs), 0, 0, (int*)nullptr);
          else hipLaunchKernelGGL(k_touch, dim3(g), dim3(bs), 0, 0, buf, n);
        }
        hipDeviceSynchronize();
        const int reps = 50;
        hipEventRecord(a, 0);
        for (int r = 0; r < reps; ++r) {
          if (which == 0) hipLaunchKernelGGL(k_empty, dim3(g), dim3(bs), 0, 0, (int*)nullptr);
          else hipLaunchKernelGGL(k_touch, dim3(g), dim3(bs), 0, 0, buf, n);
        }
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double us = ms * 1e3 / reps;
        const double waves = (double)g * ((bs + 63) / 64);
        printf("%s block=%4d grid=%6d waves=%8.0f  %7.2f us/launch  %6.2f waves/ns  %6.1f WG/us\n", which ? "touch" : "empty", bs, g, waves, us,
               waves / (us * 1e3), g / us);
      }
  return 0;
}

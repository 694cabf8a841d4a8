// cost of running straight-line code once: N instructions between two clock reads, first launch after other kernels vs repeated
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))
#define R1024(x) R4(R256(x))
template <int KB>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* t) {
  float a = threadIdx.x * 1.0f, b = 1.0001f;
  unsigned long long t0 = wall_clock64();
  // 8-byte VOP3 instructions: 128 per KB
  if (KB >= 1) { R64(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
  if (KB >= 4) { R64(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));) R256(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
  if (KB >= 16) { R1024(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));) R256(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = t1; }
  if (a == 1.234f) out[0] = a;
}
__global__ void flush(float* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
template <int KB> int run(float* out, unsigned long long* t, float* junk, int nwg) {
  std::vector<double> cold, warm;
  for (int rep = 0; rep < 9; ++rep) {
    flush<<<65536, 256>>>(junk, 16 << 20);
    k<KB><<<nwg, 256>>>(out, t); CK(hipDeviceSynchronize());
    unsigned long long h[2]; CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost)); cold.push_back((h[1] - h[0]) / 100.0);
    k<KB><<<nwg, 256>>>(out, t); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost)); warm.push_back((h[1] - h[0]) / 100.0);
  }
  std::sort(cold.begin(), cold.end()); std::sort(warm.begin(), warm.end());
  printf("~%2d KB of straight-line VALU code, %3d WGs: after other kernels %.2f us, launched again right away %.2f us\n", KB == 1 ? 1 : (KB == 4 ? 4 : 16), nwg, cold[4], warm[4]);
  return 0;
}
int main() {
  float* out; unsigned long long* t; float* junk;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&t, 16 * 1024)); CK(hipMalloc(&junk, (size_t)(16 << 20) * 4)); CK(hipMemset(junk, 0, (size_t)(16 << 20) * 4));
  for (int nwg : {1, 39}) { run<1>(out, t, junk, nwg); run<4>(out, t, junk, nwg); run<16>(out, t, junk, nwg); }
  return 0;
}

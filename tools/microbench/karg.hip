// latency of scalar loads from the kernel-argument segment (first touch, second touch), plain launch and graph replay
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Big { unsigned long long v[32]; };
__global__ __launch_bounds__(256) void k(Big a, unsigned long long* t, const float* g) {
  const unsigned long long* ka = (const unsigned long long*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned long long t0 = wall_clock64();
  unsigned long long x0, x1, x2;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(x0) : "s"(ka) : "memory");
  unsigned long long t1 = wall_clock64();
  asm volatile("s_load_dwordx2 %0, %1, 0x80\n s_waitcnt lgkmcnt(0)" : "=s"(x1) : "s"(ka) : "memory");   // another cache line of the segment
  unsigned long long t2 = wall_clock64();
  asm volatile("s_load_dwordx2 %0, %1, 0x8\n s_waitcnt lgkmcnt(0)" : "=s"(x2) : "s"(ka) : "memory");    // the first line again
  unsigned long long t3 = wall_clock64();
  float v = g[threadIdx.x];   // a cold global load for comparison
  asm volatile("" :: "v"(v));
  unsigned long long t4 = wall_clock64();
  if (threadIdx.x == 0) { unsigned long long* o = t + 8 * blockIdx.x; o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = x0 + x1 + x2; }
}
__global__ void flush(float* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
int main() {
  unsigned long long* t; float* g; float* junk; const size_t nj = 96 << 20;
  CK(hipMalloc(&t, 8 * 8 * 64)); CK(hipMalloc(&g, 1 << 20)); CK(hipMalloc(&junk, nj * 4)); CK(hipMemset(junk, 0, nj * 4));
  Big a{}; hipStream_t s; CK(hipStreamCreate(&s));
  hipGraph_t gr; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 4; ++i) { flush<<<4096, 256, 0, s>>>(junk, 1 << 20); k<<<8, 256, 0, s>>>(a, t, g); }
  CK(hipStreamEndCapture(s, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<double> d1, d2, d3, d4;
    for (int rep = 0; rep < 20; ++rep) {
      if (mode == 0) { flush<<<4096, 256, 0, s>>>(junk, 1 << 20); k<<<8, 256, 0, s>>>(a, t, g); } else CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      unsigned long long h[8]; CK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
      d1.push_back((h[1] - h[0]) / 100.0); d2.push_back((h[2] - h[1]) / 100.0); d3.push_back((h[3] - h[2]) / 100.0); d4.push_back((h[4] - h[3]) / 100.0);
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%s: first s_load of the kernarg segment %.2f us, another line %.2f us, same line again %.2f us; cold global load %.2f us\n",
           mode ? "graph replay" : "plain launch", med(d1), med(d2), med(d3), med(d4));
  }
  return 0;
}

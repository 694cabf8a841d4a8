// per-CU load-rate microbenchmark: NWG workgroups of 256 threads each fetch `bytes` of a shared buffer in different patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
// MODE 0: weight-tile pattern (lane (j,s): row j*512B + s*32B, 16 loads hi/lo x 4 chunks x 2 tiles)   MODE 1: contiguous 1 KB per wave-instruction
template <int MODE, int NLOADS, bool PASS2, int NT>
__global__ __launch_bounds__(NT) void k(const char* __restrict__ buf, size_t stride_wg, float* out, unsigned long long* t) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, s = lane >> 4;
  const char* base = buf + (size_t)blockIdx.x * stride_wg;
  float pre = 0;
  if (PASS2) {  // first pass: bring the data into the L2 (not timed)
#pragma unroll
    for (int i = 0; i < NLOADS; ++i) {
      size_t off;
      if (MODE == 0) { const int n = i >> 3, c = (i >> 1) & 3, pl = i & 1; off = (size_t)(32 * w + 16 * n + j) * 512 + c * 128 + s * 32 + pl * 16; }
      else off = ((size_t)(w * NLOADS + i) * 64 + lane) * 16;
      pre += reinterpret_cast<const float4*>(base + off)->y;
    }
    asm volatile("" ::"v"(pre));
    // evict the L1: read 64 KB of something else
    for (int i = 0; i < 16; ++i) pre += reinterpret_cast<const float4*>(buf + (8 << 20) + ((size_t)(w * 16 + i) * 64 + lane) * 16)->y;
    asm volatile("" ::"v"(pre));
    __syncthreads();
  }
  unsigned long long t0 = wall_clock64();
  float4 v[NLOADS];
#pragma unroll
  for (int i = 0; i < NLOADS; ++i) {
    size_t off;
    if (MODE == 0) { const int n = i >> 3, c = (i >> 1) & 3, pl = i & 1; off = (size_t)(32 * w + 16 * n + j) * 512 + c * 128 + s * 32 + pl * 16; }
    else off = ((size_t)(w * NLOADS + i) * 64 + lane) * 16;
    v[i] = *reinterpret_cast<const float4*>(base + off);
  }
  float acc = 0;
#pragma unroll
  for (int i = 0; i < NLOADS; ++i) acc += v[i].x + v[i].w;
  asm volatile("" ::"v"(acc));
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = t1; }
  if (acc + pre == 1.234e-30f) out[0] = acc;
}
__global__ void flush(float* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
template <int MODE, int NLOADS, bool PASS2 = false, int NT = 256> int run(const char* name, const char* buf, size_t stride, int nwg, float* out, unsigned long long* t, float* junk, size_t njunk) {
  std::vector<double> med;
  for (int rep = 0; rep < 9; ++rep) {
    flush<<<(unsigned)((njunk + 255) / 256), 256>>>(junk, njunk);   // push the buffer out of the L2s
    k<MODE, NLOADS, PASS2, NT><<<nwg, NT>>>(buf, stride, out, t);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(2 * nwg);
    CK(hipMemcpy(h.data(), t, sizeof(unsigned long long) * 2 * nwg, hipMemcpyDeviceToHost));
    std::vector<double> d;
    for (int i = 0; i < nwg; ++i) d.push_back((h[2 * i + 1] - h[2 * i]) / 100.0);
    std::sort(d.begin(), d.end());
    med.push_back(d[nwg / 2]);
  }
  std::sort(med.begin(), med.end());
  printf("%-44s nwg %3d  %3d thr %3d KB per WG: median WG %.2f us -> %.1f GB/s per CU\n", name, nwg, NT, NLOADS * NT / 64, med[4], NLOADS * NT / 64 * 1024 / med[4] / 1e3);
  return 0;
}
int main() {
  const size_t sz = 64 << 20;
  char* buf; float* out; unsigned long long* t; float* junk;
  CK(hipMalloc(&buf, sz)); CK(hipMemset(buf, 0, sz)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&t, 16 * 1024));
  const size_t njunk = 96 << 20; CK(hipMalloc(&junk, njunk * 4)); CK(hipMemset(junk, 0, njunk * 4));
  for (int nwg : {39, 256}) {
    run<1, 16, false, 256>("contiguous cold", buf, 0, nwg, out, t, junk, njunk);
    run<1, 8, false, 512>("contiguous cold", buf, 0, nwg, out, t, junk, njunk);
    run<1, 4, false, 1024>("contiguous cold", buf, 0, nwg, out, t, junk, njunk);
    run<1, 16, true, 256>("contiguous L2 warm", buf, 0, nwg, out, t, junk, njunk);
    run<1, 8, true, 512>("contiguous L2 warm", buf, 0, nwg, out, t, junk, njunk);
    run<1, 4, true, 1024>("contiguous L2 warm", buf, 0, nwg, out, t, junk, njunk);
    run<1, 12, false, 1024>("contiguous cold", buf, 0, nwg, out, t, junk, njunk);
    run<1, 12, true, 1024>("contiguous L2 warm", buf, 0, nwg, out, t, junk, njunk);
  }
  return 0;
}

// Which short sequences around v_rcp_f32 are CORRECTLY ROUNDED on gfx950?  (round 5, the k_alloc_tsdf instruction diet)
//
// The projective update divides three times per voxel (1/z of the projection, 1/d^2 of the measurement weight, the blend); the
// compiler's IEEE division is v_div_scale x2 + v_rcp + 7 fma/mul + v_div_fmas + v_div_fixup = 12 instructions because it must
// also be right when an operand or the quotient is subnormal or huge.  For operands in a guarded range a shorter sequence gives
// the same bits -- IF it does, which is what this program establishes for the hardware's v_rcp_f32:
//   * reciprocal: EXHAUSTIVELY, every float32 in [2^lo, 2^hi) -- enumeration is the proof;
//   * quotient a / b (Markstein: y = RN(1/b), q = RN(a y), r = a - b q exactly, q' = RN(q + r y)): a theorem given y, checked
//     here on 2^36 random pairs as a guard against a slip in the restatement.
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench/exact_division.hip -o /tmp/exdiv && /tmp/exdiv
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ inline float rcp_a(float x) {  // one Newton step
  const float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ inline float rcp_b(float x) {  // two
  const float r = rcp_a(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
template <int V>
__device__ inline float rcp_v(float x) {
  return V == 0 ? __builtin_amdgcn_rcpf(x) : V == 1 ? rcp_a(x) : rcp_b(x);
}
template <int V>
__device__ inline float div_v(float a, float b) {
  const float y = rcp_v<V>(b);
  const float q = a * y;
  const float r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}

template <int V>
__global__ void k_rcp(uint32_t first, uint32_t count, unsigned long long* bad, uint32_t* example) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t bits = first + (uint32_t)i;
    const float x = __uint_as_float(bits);
    const float want = 1.0f / x, got = rcp_v<V>(x);
    if (__float_as_uint(want) != __float_as_uint(got)) {
      if (atomicAdd(bad, 1ull) == 0) *example = bits;
    }
  }
}

__device__ inline uint32_t mix(uint64_t& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(s >> 32);
}
// a, b: random mantissas, exponents uniform in [elo, ehi] (biased), random sign of a
template <int V>
__global__ void k_div(uint64_t seed, int per_thread, int elo, int ehi, unsigned long long* bad, uint32_t* example) {
  uint64_t s = seed ^ ((blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull);
  mix(s);
  for (int k = 0; k < per_thread; ++k) {
    const uint32_t ra = mix(s), rb = mix(s), re = mix(s);
    const uint32_t ea = elo + (re & 0xffff) % (ehi - elo + 1), eb = elo + (re >> 16) % (ehi - elo + 1);
    const float a = __uint_as_float((ra & 0x807fffffu) | (ea << 23)), b = __uint_as_float((rb & 0x007fffffu) | (eb << 23));
    const float want = a / b, got = div_v<V>(a, b);
    if (__float_as_uint(want) != __float_as_uint(got)) {
      if (atomicAdd(bad, 1ull) == 0) {
        example[0] = __float_as_uint(a);
        example[1] = __float_as_uint(b);
      }
    }
  }
}

int main() {
  unsigned long long* bad;
  uint32_t* ex;
  hipMalloc(&bad, 8);
  hipMalloc(&ex, 8);
  auto report = [&](const char* what) {
    unsigned long long h;
    uint32_t e[2];
    hipDeviceSynchronize();
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(e, ex, 8, hipMemcpyDeviceToHost);
    float f0, f1;
    memcpy(&f0, &e[0], 4);
    memcpy(&f1, &e[1], 4);
    printf("%-72s mismatches %llu", what, h);
    if (h) printf("   e.g. 0x%08x (%g) 0x%08x (%g)", e[0], f0, e[1], f1);
    printf("\n");
    hipMemset(bad, 0, 8);
    hipMemset(ex, 0, 8);
  };
  hipMemset(bad, 0, 8);
  hipMemset(ex, 0, 8);
  // reciprocal, exhaustive over biased exponents [lo, hi): every mantissa
  const int ranges[][2] = {{1, 254}, {27, 227}, {64, 190}};  // all normals; [2^-100, 2^100); [2^-63, 2^63)
  for (auto& r : ranges) {
    const uint32_t first = (uint32_t)r[0] << 23, count = (uint32_t)(r[1] - r[0]) << 23;
    char buf[128];
    snprintf(buf, sizeof buf, "1/x, x in [2^%d, 2^%d): v_rcp_f32 alone", r[0] - 127, r[1] - 127);
    k_rcp<0><<<4096, 256>>>(first, count, bad, ex);
    report(buf);
    snprintf(buf, sizeof buf, "1/x, x in [2^%d, 2^%d): v_rcp_f32 + one Newton step (2 fma)", r[0] - 127, r[1] - 127);
    k_rcp<1><<<4096, 256>>>(first, count, bad, ex);
    report(buf);
    snprintf(buf, sizeof buf, "1/x, x in [2^%d, 2^%d): v_rcp_f32 + two Newton steps (4 fma)", r[0] - 127, r[1] - 127);
    k_rcp<2><<<4096, 256>>>(first, count, bad, ex);
    report(buf);
  }
  // quotient: 2^36 random pairs with exponents in [2^-40, 2^40]
  k_div<1><<<16384, 256>>>(12345, 16384, 127 - 40, 127 + 40, bad, ex);
  report("a/b (Markstein on the one-step reciprocal), 2^36 pairs, |exp| <= 40");
  k_div<2><<<16384, 256>>>(777, 16384, 127 - 40, 127 + 40, bad, ex);
  report("a/b (Markstein on the two-step reciprocal), 2^36 pairs, |exp| <= 40");
  k_div<1><<<16384, 256>>>(99, 16384, 127 - 3, 127 + 3, bad, ex);
  report("a/b (Markstein on the one-step reciprocal), 2^36 pairs, |exp| <= 3");
  return 0;
}

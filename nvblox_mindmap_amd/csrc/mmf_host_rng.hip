// mmf_host_rng.hip -- host-only: the draw behind the reference's vertex sampling,
//     indices = torch.randperm(num_vertices)[:desired_num_vertices]      (mindmap/data_loading/vertex_sampling.py:143-145)
// on torch's CPU default generator, restated so that it costs O(kept) swaps + the generator advance instead of a full
// permutation of every mesh vertex (and so that it never enters torch's parallel fill, which costs milliseconds above
// 32 768 elements on a box with a small CPU quota).
//
// What torch does (ATen/native/TensorFactories.cpp: randperm_cpu, the n < UINT32_MAX / 20 branch):
//     r = [0, 1, ..., n-1];  for i in 0 .. n-2:  z = generator->random() % (n - i);  swap(r[i], r[i + z])
// with generator->random() = the next 32-bit output of at::mt19937 (the standard MT19937, ATen/core/MT19937RNGEngine.h).
// After step i, r[i] is final, so the first k entries need the first k steps only; the remaining draws just advance the
// engine.  The engine state is torch's own serialised generator state (torch.get_rng_state(), CPUGeneratorImplState: legacy
// POD {u64 seed; i32 left; i32 seeded; u64 next; u64 state[624]; ...}), read and written back in place, so the generator
// ends up exactly where torch.randperm(n) would have left it (tests/test_cpu_host_rng.py compares draws AND states).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/mmfusion.h"

namespace {

constexpr int kN = 624, kM = 397;
constexpr size_t kOffLeft = 8, kOffSeeded = 12, kOffNext = 16, kOffState = 24;
constexpr size_t kStateBytes = 5056;  // sizeof(at::CPUGeneratorImplState)

struct Engine {
  uint32_t st[kN];
  int left;
  uint32_t next;

  static inline uint32_t twist(uint32_t u, uint32_t v) {
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
  }
  void next_state() {
    uint32_t* p = st;
    left = kN;
    next = 0;
    for (int j = kN - kM + 1; --j; p++) *p = p[kM] ^ twist(p[0], p[1]);
    for (int j = kM; --j; p++) *p = p[kM - kN] ^ twist(p[0], p[1]);
    *p = p[kM - kN] ^ twist(p[0], st[0]);
  }
  inline uint32_t draw() {
    if (--left == 0) next_state();
    uint32_t y = st[next++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  // advance by m draws without producing them
  void skip(int64_t m) {
    while (m > 0) {
      if (left <= 1) {  // the next draw refills
        next_state();   // left = 624, next = 0: that draw takes st[0]
        left = kN;      // (draw() decrements before it tests: after the refilling draw left stays 624, next = 1)
        next = 1;
        m--;
        continue;
      }
      const int64_t take = m < (int64_t)(left - 1) ? m : (int64_t)(left - 1);
      left -= (int)take;
      next += (uint32_t)take;
      m -= take;
    }
  }
};

}  // namespace

extern "C" int mmf_host_randperm_prefix(uint8_t* torch_cpu_rng_state, int64_t state_bytes, int64_t n, int64_t k, int64_t* out) {
  if (!torch_cpu_rng_state || state_bytes != (int64_t)kStateBytes || n < 0 || k < 0 || k > n || (k > 0 && !out)) return MMF_ERR_INVALID_ARG;
  if ((uint64_t)n >= (uint64_t)0xffffffffu / 20u) return MMF_ERR_INVALID_ARG;  // torch switches algorithm there
  Engine e;
  int32_t left32;
  uint64_t next64;
  memcpy(&left32, torch_cpu_rng_state + kOffLeft, 4);
  memcpy(&next64, torch_cpu_rng_state + kOffNext, 8);
  e.left = left32;
  e.next = (uint32_t)next64;
  const uint8_t* sp = torch_cpu_rng_state + kOffState;
  for (int i = 0; i < kN; ++i) {
    uint64_t w;
    memcpy(&w, sp + 8 * (size_t)i, 8);
    e.st[i] = (uint32_t)w;
  }
  // sparse Fisher-Yates: only displaced entries are stored (open addressing, position -> value)
  int64_t cap = 16;
  while (cap < 4 * k + 16) cap <<= 1;
  std::vector<int64_t> keys((size_t)cap, -1), vals((size_t)cap, 0);
  auto slot_of = [&](int64_t pos) -> int64_t {
    uint64_t h = (uint64_t)pos * 0x9e3779b97f4a7c15ull;
    int64_t s = (int64_t)(h >> 20) & (cap - 1);
    while (keys[(size_t)s] != -1 && keys[(size_t)s] != pos) s = (s + 1) & (cap - 1);
    return s;
  };
  auto get = [&](int64_t pos) -> int64_t {
    const int64_t s = slot_of(pos);
    return keys[(size_t)s] == pos ? vals[(size_t)s] : pos;
  };
  auto put = [&](int64_t pos, int64_t v) {
    const int64_t s = slot_of(pos);
    keys[(size_t)s] = pos;
    vals[(size_t)s] = v;
  };
  const int64_t steps = n > 0 ? n - 1 : 0;  // torch draws n - 1 numbers
  const int64_t front = k < steps ? k : steps;
  for (int64_t i = 0; i < front; ++i) {
    const int64_t z = (int64_t)((uint64_t)e.draw() % (uint64_t)(n - i));
    const int64_t a = get(i), b = get(i + z);
    out[i] = b;       // r[i] after the swap: final
    put(i + z, a);
  }
  if (k > front) out[front] = get(front);  // k == n: the last element is whatever is left (no draw)
  e.skip(steps - front);
  left32 = e.left;
  next64 = e.next;
  memcpy(torch_cpu_rng_state + kOffLeft, &left32, 4);
  memcpy(torch_cpu_rng_state + kOffNext, &next64, 8);
  uint8_t* wp = torch_cpu_rng_state + kOffState;
  for (int i = 0; i < kN; ++i) {
    const uint64_t w = e.st[i];
    memcpy(wp + 8 * (size_t)i, &w, 8);
  }
  (void)kOffSeeded;
  return MMF_OK;
}

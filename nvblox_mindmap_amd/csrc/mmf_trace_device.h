// mmf_trace_device.h -- optional per-workgroup timeline of the fused frame kernels (diagnostics; tools/wg_trace.py).
// One record per workgroup: {role id, start, end} in 100 MHz wall-clock ticks, at slot (id / 10 - 1) * 8192 + blockIdx.x.  Off (null buffer) = one scalar load per
// workgroup.  The buffer pointer lives in a translation-unit-local device variable; each .hip file that includes this header
// exports a setter (set_wg_trace_*), mmf_debug_wg_trace calls them all.
#pragma once
#include <hip/hip_runtime.h>

namespace mmf {

static __device__ unsigned long long* g_wg_trace = nullptr;  // records of 3 u64, slot = base(kernel) + blockIdx.x
static __device__ int g_wg_trace_cap = 0;

#ifdef MMF_WG_TRACE
__device__ inline long long wg_trace_begin() { return g_wg_trace ? (long long)wall_clock64() : 0; }

// call from every thread of the workgroup role (or at least thread 0); only thread 0 records.  No atomics: a shared record
// counter serialises thousands of workgroups and distorts what it measures.
__device__ inline void wg_trace_end(long long t0, int id) {
  if (g_wg_trace && threadIdx.x == 0) {
    const long long t1 = (long long)wall_clock64();
    const int slot = ((id & 0xff) / 10 - 1) * 8192 + (int)blockIdx.x;  // every fused kernel launches fewer than 8192 workgroups
    if (slot < g_wg_trace_cap && blockIdx.x < 8192) {
      g_wg_trace[3 * slot] = (unsigned long long)id;
      g_wg_trace[3 * slot + 1] = (unsigned long long)t0;
      g_wg_trace[3 * slot + 2] = (unsigned long long)t1;
    }
  }
}
__device__ inline bool wg_trace_on() { return g_wg_trace != nullptr; }
constexpr int kWgTraceBuilt = 1;
#else
// Default build: the hooks compile to nothing (even one scalar load at the top of every workgroup is measurable on kernels
// made of thousands of 3 us workgroups).  `make WG_TRACE=1` builds the instrumented library for tools/wg_trace.py.
__device__ inline long long wg_trace_begin() { return 0; }
__device__ inline void wg_trace_end(long long, int) {}
__device__ inline bool wg_trace_on() { return false; }
constexpr int kWgTraceBuilt = 0;
#endif

#define MMF_DEFINE_WG_TRACE_SETTER(name)                                                                 \
  int name(unsigned long long* buf, int cap) {                                                           \
    if (!kWgTraceBuilt) return buf ? 2 : 0;                                                               \
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace), &buf, sizeof(buf)) != hipSuccess) return 1;             \
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace_cap), &cap, sizeof(cap)) != hipSuccess) return 1;         \
    return 0;                                                                                             \
  }

// record ids
enum : int {
  kTrFrontRay = 10, kTrFrontMaskRows = 11, kTrFrontDecay = 12,
  kTrAllocJob = 20, kTrAllocMaskCols = 21,
  kTrTsdfPass = 30, kTrTsdfNew = 31,
  kTrSphereAlloc = 40, kTrSphereTrace = 41,
  kTrAppFrame = 50,
  kTrFeatureFlat = 60,
};

}  // namespace mmf

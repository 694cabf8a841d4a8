// mmf_alloc_device.h -- single-workgroup block allocation (flag compaction + hash lookup / insertion) as a device
// function, so that it can be one role of several horizontally fused launches (k_alloc_jobs, k_sphere_alloc).
#pragma once
#include "mmf_device.h"

namespace mmf {

__device__ inline u64 grid_cell_key(const KeySrc& ks, int cell) {
  int gz = cell % ks.nz;
  int t = cell / ks.nz;
  int gy = t % ks.ny;
  int gx = t / ks.ny;
  return pack_key(gx + ks.ox, gy + ks.oy, gz + ks.oz);
}

// mode 0: the cell is a cell of the dense view grid; mode 1: the cell is a position of another layer's live
// list and the producer of the flags (k_app_candidates) stored the block key next to the flag.
__device__ inline u64 cell_key(const KeySrc& ks, const Scratch& sc, int cell) {
  return ks.mode == 0 ? grid_cell_key(ks, cell) : sc.cell_key[cell];
}

// Deallocation of the blocks flagged by the decay pass: ONE workgroup of 64*NW threads compacts the live list in place
// (order preserving), pushes the freed slots in live order and accounts the tombstones (amortised rebuild).  IPT
// consecutive entries per thread, so a few thousand live blocks are one pass (the pass cost is the workgroup scan).
// ERASE: also drop the dead blocks from the hash / dense table / slot keys here; false when the decay workgroup that
// found the block already did (fused frame: decay_body<true>).  No-op when nothing was flagged.
// DENSE: the layer is known to be indexed by its dense table (bounded workspace): the hash paths are compiled out.  These
// bodies run in ONE workgroup, once, through a cold instruction cache -- their duration follows their code size.
// WMAX: the dead blocks are not read from kill[] but decided here, from the block's largest weight (LayerDev::wmax, kept
// current by k_tsdf_pass): dead <=> wmax * decay_f < decay_thr, i.e. every voxel of the block would fall under the
// threshold -- the whole deallocation of a Mapper.decay() in one workgroup that touches no voxel (kill / any_kill unused).
template <int NW, int IPT, bool ERASE, bool DENSE = false, bool WMAX = false>
__device__ inline void live_compact_body(const LayerDev& L, uint8_t* __restrict__ kill, int* any_kill, int* lds, int* carry,
                                         float decay_f = 0.0f, float decay_thr = 0.0f) {
  static_assert(IPT % 4 == 0, "entries are fetched four at a time");
  static_assert(!WMAX || ERASE, "the wmax rule is the only pass over the dead blocks: it must also drop them from the index");
  const bool dense = DENSE || L.dense != nullptr;
  constexpr int NT = 64 * NW;
  // One memory round trip instead of three: the flag, the two counters and the thread's entries of the first pass (the
  // only pass for up to NT * IPT live blocks) are requested together, the entries from clamped -- always valid --
  // positions, before any of them is looked at.
  const int pending = WMAX ? 1 : *any_kill;
  const int n = L.ctr[0];
  const int free0 = L.ctr[1];
  int4 pre_v[IPT / 4];
  uint32_t pre_k[IPT / 4];
  {
    const int last4 = (L.cap & ~3) - 4;
#pragma unroll
    for (int g = 0; g < IPT; g += 4) {
      int p = (int)threadIdx.x * IPT + g;
      p = p < last4 ? p : last4;
      pre_v[g / 4] = *reinterpret_cast<const int4*>(L.live + p);
      pre_k[g / 4] = WMAX ? 0u : *reinterpret_cast<const uint32_t*>(kill + p);
    }
  }
  if (!pending) return;
  if (threadIdx.x == 0) {
    carry[0] = 0;      // survivors written so far
    carry[1] = free0;  // free stack size
  }
  __syncthreads();
  for (int base = 0; base < n; base += NT * IPT) {
    const int i0 = base + (int)threadIdx.x * IPT;
    int slot[IPT];
    bool dead_q[IPT];
    int keep = 0, dead = 0;
#pragma unroll
    for (int g = 0; g < IPT; g += 4) {
      unsigned k4 = 0;
      int s4[4] = {-1, -1, -1, -1};
      if (i0 + g + 3 < n) {
        // (i0 + g + 3 < n <= cap: the clamp above did not move a prefetched position of the first pass)
        const int4 v = base == 0 ? pre_v[g / 4] : *reinterpret_cast<const int4*>(L.live + i0 + g);
        s4[0] = v.x, s4[1] = v.y, s4[2] = v.z, s4[3] = v.w;
        if (!WMAX) {
          k4 = base == 0 ? pre_k[g / 4] : *reinterpret_cast<const uint32_t*>(kill + i0 + g);
          if (k4) *reinterpret_cast<uint32_t*>(kill + i0 + g) = 0u;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (i0 + g + q < n) {
            s4[q] = L.live[i0 + g + q];
            if (!WMAX && kill[i0 + g + q]) {
              k4 |= 1u << (8 * q);
              kill[i0 + g + q] = 0;
            }
          }
      }
      if (WMAX) {  // four independent gathers (slot 0 stands in for the entries past the list: not counted below)
        float w4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w4[q] = L.wmax[s4[q] < 0 ? 0 : s4[q]];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (w4[q] * decay_f < decay_thr) k4 |= 1u << (8 * q);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        slot[g + q] = s4[q];
        dead_q[g + q] = ((k4 >> (8 * q)) & 0xffu) != 0u;
        if (i0 + g + q < n) {
          if (dead_q[g + q]) dead++;
          else keep++;
        }
      }
    }
    int ea, eb, ta, tb;
    block_excl_scan2<NW>(keep, dead, lds, ea, eb, ta, tb);
    const int c0 = carry[0], c1 = carry[1];
    __syncthreads();  // every read of live[base..] and carry happened before any write below
    int wk = c0 + ea, wd = c1 + eb;
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      if (i0 + q >= n) continue;
      if (dead_q[q]) {
        L.free_stack[wd++] = slot[q];
        if (ERASE) {
          if (!dense) hash_erase(L, L.slot_key[slot[q]]);  // tombstone; dropped at the next rebuild
          dense_set(L, L.slot_key[slot[q]], 0);
          L.slot_key[slot[q]] = kEmptyKey;
        }
      } else {
        L.live[wk++] = slot[q];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      carry[0] = c0 + ta;
      carry[1] = c1 + tb;
    }
    __syncthreads();
  }
  const int n_live = carry[0];
  const int n_tomb = L.ctr[4] + (carry[1] - free0);
  // amortised rebuild: only when tombstones fill more than a quarter of the table
  const bool rebuild = !dense && (unsigned)n_tomb * 4u > L.hmask + 1u;
  __syncthreads();
  if (rebuild) {
    for (unsigned h = threadIdx.x; h <= L.hmask; h += NT) L.htab[h].key = kEmptyKey;
    __syncthreads();
    for (int i = threadIdx.x; i < n_live; i += NT) {
      const int slot = L.live[i];
      hash_insert(L, L.slot_key[slot], slot);
    }
  }
  if (threadIdx.x == 0) {
    L.ctr[0] = n_live;
    L.ctr[1] = carry[1];
    L.ctr[4] = rebuild ? 0 : n_tomb;
    if (rebuild) L.ctr[5]++;  // diagnostics (mmf_debug_hash_state): table rebuilds since the layer was reset
    if (L.hint_live) *L.hint_live = n_live;
    if (!WMAX) *any_kill = 0;
  }
}

// Small cell counts (bounded workspaces: a few thousand cells): count + scan + emit fused into ONE launch
// of one 1024-thread workgroup, 4096 cells per pass with a running carry.  Same candidate order, same
// slot assignment as the three-kernel path.
// MODE: the job's KeySrc mode when the launcher knows it (0 grid cells, 1 list cells), -1 = read it from the job.
// NW: waves of the calling workgroup (16 = the 1024-thread form; 4 when the job shares a launch of 256-thread workgroups).
template <bool DENSE = false, int MODE = -1, int NW = 16>
__device__ inline void alloc_job_body(const AllocJob& J, long long* stats, int* lds, int* carry, int* ctx) {
  const LayerDev& L = J.L;
  const bool dense = DENSE || L.dense != nullptr;
  const int mode = MODE >= 0 ? MODE : J.ks.mode;
  const KeySrc& ks = J.ks;
  const Scratch& sc = J.sc;
  const int stat_upd = J.stat_upd, stat_new = J.stat_new;
  int ncells = J.ncells;
  if (J.timeline && threadIdx.x == 0) J.timeline[0] = wall_clock64();
  if (J.kill) {  // a decay pass ran in the previous launch (and already erased its dead blocks from the index): compact
                 // the live list / push the freed slots before allocating (slot reuse order is spec)
    live_compact_body<NW, 64 / NW, false, DENSE>(L, J.kill, J.any_kill, lds, carry);  // 4096 entries per pass
    __syncthreads();
  }
  if (J.timeline && threadIdx.x == 0) J.timeline[1] = wall_clock64();
  if (mode == 1) {  // list cells: only the producer's live positions carry meaningful flags
    const int nl = *ks.n_live;
    ncells = ncells < nl ? ncells : nl;
  }
  if (J.zero_me && (int)threadIdx.x < J.zero_n) J.zero_me[threadIdx.x * J.zero_stride] = 0;
  if (threadIdx.x == 0) {
    carry[0] = 0;
    carry[1] = 0;
    ctx[0] = L.ctr[0];
    ctx[1] = L.ctr[1];
    ctx[2] = L.ctr[2];
    ctx[3] = L.ctr[1] + (L.cap - L.ctr[2]);  // room
  }
  // no barrier here: ctx / carry are first read after the barriers of the workgroup scan below, so the counter
  // round trip of thread 0 overlaps with everyone's table loads
  for (int base = 0; base < ncells; base += 256 * NW) {  // 4 cells per thread
    const int cell0 = base + threadIdx.x * 4;
    uint32_t f4 = 0;
    if (cell0 < ncells) f4 = *reinterpret_cast<const uint32_t*>(sc.flags + cell0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (cell0 + k >= ncells) f4 &= ~(0xffu << (8 * k));  // stale flags beyond the live count
    int slot4[4] = {0, 0, 0, 0};
    u64 key4[4] = {0, 0, 0, 0};
    int nf = 0, nn = 0;
    if (cell0 < ncells) {
      // Every load of this pass is issued before the first one is consumed.  Grid cells have computable keys,
      // so their table entries are fetched without waiting for the flags; list cells read key + flag together
      // and look the flagged ones up in a second round.
      if (mode == 0) {
        unsigned h4[4] = {0, 0, 0, 0};
        uint4 e4[4];
        int d4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          e4[k] = make_uint4(0, 0, 0, 0);
          if (cell0 + k < ncells) {
            key4[k] = grid_cell_key(ks, cell0 + k);
            if (dense) {
              int x, y, z;
              unpack_key(key4[k], x, y, z);
              d4[k] = (int)L.dense[dense_cell(L, x, y, z)];
            } else {
              h4[k] = hash_key(key4[k]) & L.hmask;
              e4[k] = hash_load(L, h4[k]);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if ((f4 >> (8 * k)) & 0xffu) {
            slot4[k] = dense ? d4[k] - 1 : hash_resolve(L, key4[k], h4[k], e4[k]);
            nf++;
            nn += slot4[k] < 0;
          }
        }
      } else if (f4) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((f4 >> (8 * k)) & 0xffu) key4[k] = sc.cell_key[cell0 + k];
        unsigned h4[4] = {0, 0, 0, 0};
        uint4 e4[4];
        int d4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          e4[k] = make_uint4(0, 0, 0, 0);
          if ((f4 >> (8 * k)) & 0xffu) {
            if (dense) {
              int x, y, z;
              unpack_key(key4[k], x, y, z);
              d4[k] = (int)L.dense[dense_cell(L, x, y, z)];
            } else {
              h4[k] = hash_key(key4[k]) & L.hmask;
              e4[k] = hash_load(L, h4[k]);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if ((f4 >> (8 * k)) & 0xffu) {
            slot4[k] = dense ? d4[k] - 1 : hash_resolve(L, key4[k], h4[k], e4[k]);
            nf++;
            nn += slot4[k] < 0;
          }
        }
      }
    }
    if (J.timeline && threadIdx.x == 0) J.timeline[2] = wall_clock64();  // table loads consumed
    int ea, eb, ta, tb;
    block_excl_scan2<NW>(nf, nn, lds, ea, eb, ta, tb);
    if (J.timeline && threadIdx.x == 0) J.timeline[3] = wall_clock64();  // scan done
    const int old_live = ctx[0], old_free = ctx[1], old_bump = ctx[2], room = ctx[3];
    if (f4) {
      int pos = carry[0] + ea, rnk = carry[1] + eb, reused = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!((f4 >> (8 * k)) & 0xffu)) continue;
        int slot = slot4[k];
        const bool is_new = slot < 0;
        if (is_new) {
          if (rnk < room) {
            slot = rnk < old_free ? L.free_stack[old_free - 1 - rnk] : old_bump + (rnk - old_free);
            if (!dense) reused += hash_insert(L, key4[k], slot);
            dense_set(L, key4[k], slot + 1);
            L.slot_key[slot] = key4[k];
            L.live[old_live + rnk] = slot;
          }
          rnk++;
        }
        sc.cand_slot[pos] = slot;
        sc.cand_key[pos] = key4[k];
        sc.cand_new[pos] = is_new ? 1 : 0;
        if (J.stamp && slot >= 0) L.stamp[slot] = (J.stamp << 1) | (is_new ? 1 : 0);
        pos++;
      }
      if (mode == 0) *reinterpret_cast<uint32_t*>(sc.flags + cell0) = 0u;  // grid flags: all-zero for the next frame
      if (reused) atomicSub(&L.ctr[4], reused);  // tombstones that became keys again
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      carry[0] += ta;
      carry[1] += tb;
    }
    __syncthreads();
  }
  if (J.timeline && threadIdx.x == 0) J.timeline[4] = wall_clock64();  // inserts + candidate list written
  if (threadIdx.x == 0) {
    const int old_live = ctx[0], old_free = ctx[1], old_bump = ctx[2], room = ctx[3];
    const int n_cand = carry[0], n_new = carry[1];
    const int granted = n_new < room ? n_new : room;
    if (granted < n_new) atomicOr(&L.ctr[3], 1);
    const int from_free = granted < old_free ? granted : old_free;
    L.ctr[0] = old_live + granted;
    L.ctr[1] = old_free - from_free;
    L.ctr[2] = old_bump + (granted - from_free);
    *sc.cand_count = n_cand;
    if (sc.hint_cand) *sc.hint_cand = n_cand;
    if (L.hint_live) *L.hint_live = L.ctr[0];
    if (stats) {
      if (stat_upd >= 0) stats[stat_upd] += n_cand;
      if (stat_new >= 0) stats[stat_new] += granted;
    }
    if (J.timeline) J.timeline[5] = wall_clock64();
  }
}


// The allocation job of k_alloc_tsdf: dense-table layer, view-grid cells, shared by `nwg` workgroups of 64 * NW threads that
// each take ONE pass over 64 * NW * 4 * G consecutive cells (4 * G consecutive cells per thread; grid coordinates are stepped,
// not divided out; keys are packed only for the flagged cells).  Workgroup w publishes its candidate / new-block counts and
// sums those of the workgroups before it (decoupled prefix: w words to read, all published within a microsecond of each other;
// workgroups are dispatched in index order and never wait on a later one), so the whole job is one load round trip + one
// workgroup scan + the writes instead of a pass per 2 048 cells in sequence.  Same candidate order, slot assignment and
// outputs as alloc_job_body<true, 0>.  Every granted new block is also published to the launch as it is assigned: three
// self-validating 64-bit words {tag | slot}, {tag | key low}, {tag | key high} at J.pub[kPubRec + 3 * rank ..]; the last
// workgroup publishes the number of granted blocks {tag | n} at J.pub[0].  All with relaxed agent-scope atomics, no fences: a
// reader polls until the tag of the word it needs is this launch's.  The grid flags are left set: the launch's TSDF
// workgroups read them too; a later launch of the frame clears them.
// (kPubRec, kAllocMaxWgs: mmf_device.h)
template <int NW, int G>
__device__ inline void alloc_grid_multi_body(const AllocJob& J, long long* stats, int* lds, int* carry, int* ctx, int w, int nwg) {
  const LayerDev& L = J.L;
  const KeySrc& ks = J.ks;
  const Scratch& sc = J.sc;
  constexpr int NT = 64 * NW, CPT = 4 * G;
  const int ncells = J.ncells;
  long long* tl = (J.timeline && w == nwg - 1 && threadIdx.x == 0) ? J.timeline : nullptr;  // diagnostics: the last workgroup
  if (tl) tl[0] = tl[1] = wall_clock64();
  if (w == 0 && J.zero_me && (int)threadIdx.x < J.zero_n) J.zero_me[threadIdx.x * J.zero_stride] = 0;
  if (threadIdx.x == 0) {
    ctx[0] = L.ctr[0];
    ctx[1] = L.ctr[1];
    ctx[2] = L.ctr[2];
    ctx[3] = L.ctr[1] + (L.cap - L.ctr[2]);  // room
  }
  // (no barrier: ctx is first read after the barriers of the workgroup scan below)
  const u64 tg = (u64)J.pub_tag << 32;
  const int cell0 = w * NT * CPT + (int)threadIdx.x * CPT;
  const int gz0 = cell0 % ks.nz, t0 = cell0 / ks.nz, gy0 = t0 % ks.ny, gx0 = t0 / ks.ny;
  uint32_t f[G];
  int d[CPT];
  {  // every load before the first is consumed; cells past the grid read entry 0 and are masked
    int gx = gx0, gy = gy0, gz = gz0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int c4 = cell0 + 4 * g;
      f[g] = *reinterpret_cast<const uint32_t*>(sc.flags + (unsigned)(c4 < ncells ? c4 : 0));
      {  // byte k := 0xff where the cell's byte equals the frame's tag, else 0 (bytes of other frames are stale, never cleared)
        const uint32_t x = f[g] ^ (0x01010101u * (uint32_t)J.flag_value);
        uint32_t eq = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (((x >> (8 * k)) & 0xffu) == 0u) eq |= 0xffu << (8 * k);
        f[g] = eq;
      }
      if (c4 >= ncells) f[g] = 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool in = c4 + k < ncells;
        if (!in) f[g] &= ~(0xffu << (8 * k));  // stale flags beyond the grid
        const unsigned idx = in ? (unsigned)dense_cell(L, gx + ks.ox, gy + ks.oy, gz + ks.oz) : 0u;
        d[4 * g + k] = (int)L.dense[idx];
        if (++gz == ks.nz) {
          gz = 0;
          if (++gy == ks.ny) {
            gy = 0;
            ++gx;
          }
        }
      }
    }
  }
  int nf = 0, nn = 0;
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((f[g] >> (8 * k)) & 0xffu) {
        nf++;
        nn += d[4 * g + k] == 0;
      }
  if (tl) tl[2] = wall_clock64();  // loads consumed
  int ea, eb, ta, tb;
  block_excl_scan2<NW>(nf, nn, lds, ea, eb, ta, tb);
  if (tl) tl[3] = wall_clock64();  // scan done
  if (threadIdx.x == 0) {
    // (ctx above is complete: thread 0 passed the scan's barriers after writing it)
    __hip_atomic_store(J.pub + 1 + w, tg | ((u64)(unsigned)ta << 16) | (u64)(unsigned)tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int pa = 0, pb = 0;
    bool failed = false;
    const long long t_start = wall_clock64();  // 100 MHz: a predecessor that has not published within 50 ms never will
    for (int v = 0; v < w; ++v) {
      u64 x = 0;
      for (;;) {
        x = __hip_atomic_load(J.pub + 1 + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == J.pub_tag) break;
        if (wall_clock64() - t_start > 5000000ll) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if ((unsigned)(x >> 32) != J.pub_tag) {  // gave up waiting (never observed): the word is stale -- its counts are NOT used
        failed = true;
        continue;
      }
      pa += (int)((x >> 16) & 0xffffu);
      pb += (int)(x & 0xffffu);
    }
    if (failed) {  // this workgroup grants nothing and lists nothing (in-bounds by construction); the frame is reported as failed
      atomicOr(&L.ctr[3], 2);
      if (J.host_err) *J.host_err = 1;
      pa = pb = 0;
    }
    carry[0] = pa;
    carry[1] = pb;
    carry[2] = failed ? 1 : 0;
  }
  __syncthreads();
  if (tl) tl[4] = wall_clock64();  // counts of the earlier workgroups collected
  const int old_live = ctx[0], old_free = ctx[1], old_bump = ctx[2];
  const int room = carry[2] ? 0 : ctx[3];  // a failed prefix wait: no slot is granted by this workgroup
  if (nf && !carry[2]) {
    int pos = carry[0] + ea, rnk = carry[1] + eb;
    int gx = gx0, gy = gy0, gz = gz0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if ((f[g] >> (8 * k)) & 0xffu) {
          const u64 key = pack_key(gx + ks.ox, gy + ks.oy, gz + ks.oz);
          int slot = d[4 * g + k] - 1;
          const bool is_new = slot < 0;
          if (is_new) {
            if (rnk < room) {
              slot = rnk < old_free ? L.free_stack[old_free - 1 - rnk] : old_bump + (rnk - old_free);
              L.dense[dense_cell(L, gx + ks.ox, gy + ks.oy, gz + ks.oz)] = (unsigned short)(slot + 1);
              L.slot_key[slot] = key;
              L.live[old_live + rnk] = slot;
              u64* rec = J.pub + kPubRec + 3 * (size_t)rnk;
              __hip_atomic_store(rec, tg | (u64)(unsigned)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(rec + 1, tg | (key & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(rec + 2, tg | (key >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            rnk++;
          }
          sc.cand_slot[pos] = slot;
          sc.cand_key[pos] = key;
          sc.cand_new[pos] = is_new ? 1 : 0;
          if (J.stamp && slot >= 0) L.stamp[slot] = (J.stamp << 1) | (is_new ? 1 : 0);
          pos++;
        }
        if (++gz == ks.nz) {
          gz = 0;
          if (++gy == ks.ny) {
            gy = 0;
            ++gx;
          }
        }
      }
  }
  if (w == nwg - 1 && threadIdx.x == 0) {  // the last workgroup knows the totals
    const int n_cand = carry[0] + ta, n_new = carry[1] + tb;
    const int granted = n_new < room ? n_new : room;
    if (granted < n_new) atomicOr(&L.ctr[3], 1);
    const int from_free = granted < old_free ? granted : old_free;
    L.ctr[0] = old_live + granted;
    L.ctr[1] = old_free - from_free;
    L.ctr[2] = old_bump + (granted - from_free);
    *sc.cand_count = n_cand;
    if (sc.hint_cand) *sc.hint_cand = n_cand;
    if (L.hint_live) *L.hint_live = old_live + granted;
    __hip_atomic_store(J.pub, tg | (u64)(unsigned)granted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stats) {
      if (J.stat_upd >= 0) stats[J.stat_upd] += n_cand;
      if (J.stat_new >= 0) stats[J.stat_new] += granted;
    }
    if (tl) tl[5] = wall_clock64();
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Scalable, single-launch forms (any number of cells / live blocks): the view grid of an UNBOUNDED workspace holds ~10^6
// cells and its pools 10^5 blocks, where one workgroup (alloc_job_body, live_compact_body) would take dozens of serial
// passes and the three-kernel count / scan / emit path three launches per layer.  Every workgroup takes ONE chunk; what a
// chunk needs from the chunks before it -- how many candidates / new blocks (kept / dead list entries) they hold -- comes from
// an in-launch exclusive scan over self-validating 64-bit words
//     [63:42] launch tag   [41:40] status: 1 = the chunk's own counts, 2 = its exclusive prefix   [39:20] a   [19:0] b
// written and polled with relaxed agent-scope atomics (no fences; a stale word carries an old tag):
//   every chunk publishes its counts and arrives on one counter; the chunk that arrives LAST scans all counts (its whole
//   workgroup, 256 chunks per step), publishes every chunk's prefix and resets the counter; every chunk polls ITS prefix word.
// (First version: a decoupled look-back, every chunk's wave 0 polling its 64 predecessors.  With more chunks than one window
// -- 90 .. 256 here -- the launch took 15 to 190 us: thousands of lanes spinning on agent-scope loads starve the very stores
// they wait for.  One poller per chunk and one scanner: the cost is two memory round trips after the last chunk has counted.)
// Counts are 20 bits: pools below 2^20 blocks (the host falls back to the three-kernel path above that).
// lb: [0] arrival counter, [2 .. 2 + nwg) counts, [2 + nwg .. 2 + 2 nwg) prefixes.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kBigChunk = 1024;  // cells / list entries per workgroup and group: 256 threads x 4
constexpr unsigned kLbCountMax = (1u << 20) - 1u;

__device__ inline u64 lb_pack(unsigned tag, unsigned status, unsigned a, unsigned b) {
  return ((u64)(tag & 0x3fffffu) << 42) | ((u64)(status & 3u) << 40) | ((u64)(a & kLbCountMax) << 20) | (u64)(b & kLbCountMax);
}

__device__ inline u64 lb_poll(const u64* word, unsigned tag, unsigned status, int* err) {
  u64 x = 0;
  const long long t_start = wall_clock64();  // 100 MHz: 50 ms
  for (;;) {
    x = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(x >> 42) == (tag & 0x3fffffu) && (unsigned)((x >> 40) & 3ull) == status) return x;
    if (wall_clock64() - t_start > 5000000ll) break;
    __builtin_amdgcn_s_sleep(4);
  }
  atomicOr(err, 2);  // (never observed) the launch's hand-over failed: reported like k_alloc_tsdf's
  return lb_pack(tag, status, 0u, 0u);
}

// Exclusive prefix (pa, pb) of (ta, tb) over the chunks before w -> out2[0], out2[1] (LDS); called by every thread of the
// workgroup (256 threads) with workgroup-uniform arguments; scan: >= 10 ints of LDS, out2: >= 7 ints (out2[6] is scratch).
__device__ inline void lookback_exclusive(u64* lb, unsigned tag, int w, int nwg, int ta, int tb, int* scan, int* out2, int* err) {
  if (threadIdx.x == 0) {
    __hip_atomic_store(lb + 2 + w, lb_pack(tag, 1u, (unsigned)ta, (unsigned)tb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long old = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(lb), 1ull, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
    out2[6] = (old == (unsigned long long)(nwg - 1)) ? 1 : 0;
  }
  __syncthreads();
  if (out2[6]) {  // the last chunk to arrive: every count is published (a word not yet visible is polled for)
    int ca = 0, cb = 0;
    for (int base = 0; base < nwg; base += 256) {
      const int v = base + (int)threadIdx.x;
      int a = 0, b = 0;
      if (v < nwg) {
        const u64 x = lb_poll(lb + 2 + v, tag, 1u, err);
        a = (int)((x >> 20) & kLbCountMax);
        b = (int)(x & kLbCountMax);
      }
      int ea, eb, sa, sb;
      block_excl_scan2<4>(a, b, scan, ea, eb, sa, sb);
      if (v < nwg)
        __hip_atomic_store(lb + 2 + nwg + v, lb_pack(tag, 2u, (unsigned)(ca + ea), (unsigned)(cb + eb)), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      ca += sa;
      cb += sb;
    }
    if (threadIdx.x == 0) __hip_atomic_store(lb, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
  }
  if (threadIdx.x == 0) {
    const u64 x = lb_poll(lb + 2 + nwg + w, tag, 2u, err);
    out2[0] = (int)((x >> 20) & kLbCountMax);
    out2[1] = (int)(x & kLbCountMax);
  }
  __syncthreads();
}

// Allocation job J, chunk w of nwg: 1024 G cells (G <= 4; the launcher picks G so that a job is at most ~128 chunks where it can:
// every look-back hop is a memory round trip, ~1-2 us across XCDs, and the inclusive prefixes travel 64 chunks per hop).
//   1. every thread reads the flags of its 4 G consecutive cells; the FLAGGED cells of the chunk are compacted, in cell order, into
//      an LDS list (workgroup scan).  Flagged cells cluster (blocks along a surface are neighbours in the grid): left with their
//      threads, one thread would walk a dozen dependent lookups / insertions in sequence while its neighbours idle.
//   2. the list is walked 256 items per round, one item per thread: index lookup (hash probe or dense table), a workgroup scan of
//      "is new" per round -> the chunk's candidate and new-block counts and every item's rank among the chunk's new blocks.
//   3. look back for the counts of the chunks before this one.
//   4. every item is assigned: candidate position = prefix + list position, new blocks take their pool slot by rank (free stack
//      first, then the bump pointer), are inserted into the index and appended to the live list.
// Same candidate order, slot assignment and outputs as alloc_job_body / the three-kernel path.  Hash layers: lookups and the CAS
// insertions of other chunks run concurrently -- safe, because a chunk only ever looks up the keys of ITS cells, an inserted key
// is never one another chunk looks up, and a probe that passes a freshly claimed entry simply continues to the next.
constexpr int kBigMaxG = 4;
struct AllocBigLds {
  int scan[10];
  int sh[8];
  uint16_t cell[kBigChunk * kBigMaxG];  // flagged cells of the chunk (offset within the chunk), cell order
  int slot[kBigChunk * kBigMaxG];       // their pool slots (-1 - rank among the chunk's new blocks for a new block)
};

template <int MODE>
__device__ inline void alloc_big_body(const AllocJob& J, long long* stats, AllocBigLds& S, int w, int nwg, int G) {
  const LayerDev& L = J.L;
  const KeySrc& ks = J.ks;
  const Scratch& sc = J.sc;
  int* sh = S.sh;
  int ncells = J.ncells;
  if (MODE == 1) {
    const int nl = *ks.n_live;
    ncells = ncells < nl ? ncells : nl;
  }
  if (w == 0 && J.zero_me && (int)threadIdx.x < J.zero_n) J.zero_me[threadIdx.x * J.zero_stride] = 0;
  if (threadIdx.x == 0) {
    sh[2] = L.ctr[0];
    sh[3] = L.ctr[1];
    sh[4] = L.ctr[2];
    sh[5] = L.ctr[1] + (L.cap - L.ctr[2]);  // room
  }
  const int chunk0 = w * kBigChunk * G;
  const int t0 = (int)threadIdx.x * 4 * G;  // the thread's first cell, chunk-relative
  uint32_t fw[kBigMaxG];
#pragma unroll
  for (int g = 0; g < kBigMaxG; ++g) {
    fw[g] = 0u;
    const int cell = chunk0 + t0 + 4 * g;
    if (g < G && cell < ncells) fw[g] = *reinterpret_cast<const uint32_t*>(sc.flags + cell);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (cell + k >= ncells) fw[g] &= ~(0xffu << (8 * k));  // stale flags beyond the grid / the live count
  }
  int nf = 0;
#pragma unroll
  for (int g = 0; g < kBigMaxG; ++g)
#pragma unroll
    for (int k = 0; k < 4; ++k) nf += ((fw[g] >> (8 * k)) & 0xffu) ? 1 : 0;
  int ea, eb, n_items, tb;
  block_excl_scan2<4>(nf, 0, S.scan, ea, eb, n_items, tb);
  {
    int at = ea;
#pragma unroll
    for (int g = 0; g < kBigMaxG; ++g) {
      if (!fw[g]) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((fw[g] >> (8 * k)) & 0xffu) S.cell[at++] = (uint16_t)(t0 + 4 * g + k);
      if (MODE == 0) *reinterpret_cast<uint32_t*>(sc.flags + chunk0 + t0 + 4 * g) = 0u;  // grid flags: all-zero for the next frame
    }
  }
  __syncthreads();
  auto key_of = [&](int cell) -> u64 { return MODE == 0 ? grid_cell_key(ks, cell) : sc.cell_key[cell]; };
  int n_new = 0;
  for (int base = 0; base < n_items; base += 256) {  // (uniform trip count)
    const int j = base + (int)threadIdx.x;
    int slot = 0;
    if (j < n_items) slot = layer_lookup(L, key_of(chunk0 + (int)S.cell[j]));
    const int is_new = (j < n_items && slot < 0) ? 1 : 0;
    int ra, rb, ta, tb2;
    block_excl_scan2<4>(is_new, 0, S.scan, ra, rb, ta, tb2);
    if (j < n_items) S.slot[j] = is_new ? -1 - (n_new + ra) : slot;
    n_new += ta;
  }
  lookback_exclusive(sc.lb, J.lb_tag, w, nwg, n_items, n_new, S.scan, sh, &L.ctr[3]);
  const int old_live = sh[2], old_free = sh[3], old_bump = sh[4], room = sh[5];
  for (int j = (int)threadIdx.x; j < n_items; j += 256) {
    const int cell = chunk0 + (int)S.cell[j];
    const u64 key = key_of(cell);
    int slot = S.slot[j];
    const bool is_new = slot < 0;
    if (is_new) {
      const int rnk = sh[1] + (-1 - slot);
      slot = -1;
      if (rnk < room) {
        slot = rnk < old_free ? L.free_stack[old_free - 1 - rnk] : old_bump + (rnk - old_free);
        if (hash_insert(L, key, slot)) atomicSub(&L.ctr[4], 1);  // a tombstone became a key again (few per frame)
        dense_set(L, key, slot + 1);
        L.slot_key[slot] = key;
        L.live[old_live + rnk] = slot;
      }
    }
    const int pos = sh[0] + j;
    sc.cand_slot[pos] = slot;
    sc.cand_key[pos] = key;
    sc.cand_new[pos] = is_new ? 1 : 0;
    if (J.stamp && slot >= 0) L.stamp[slot] = (J.stamp << 1) | (is_new ? 1 : 0);
  }
  if (w == nwg - 1 && threadIdx.x == 0) {  // the last chunk knows the totals
    const int n_cand = sh[0] + n_items, n_newt = sh[1] + n_new;
    const int granted = n_newt < room ? n_newt : room;
    if (granted < n_newt) atomicOr(&L.ctr[3], 1);
    const int from_free = granted < old_free ? granted : old_free;
    L.ctr[0] = old_live + granted;
    L.ctr[1] = old_free - from_free;
    L.ctr[2] = old_bump + (granted - from_free);
    *sc.cand_count = n_cand;
    if (sc.hint_cand) *sc.hint_cand = n_cand;
    if (L.hint_live) *L.hint_live = old_live + granted;
    if (stats) {
      if (J.stat_upd >= 0) stats[J.stat_upd] += n_cand;
      if (J.stat_new >= 0) stats[J.stat_new] += granted;
    }
  }
}

// Deallocation of dead blocks, chunk w of nwg (256 threads, 4 list entries each): order-preserving, in place -- a chunk
// writes its survivors at or before its own first entry, into positions whose owners published their counts (hence loaded
// their entries) before this chunk could learn its offset.  WMAX: dead <=> wmax[slot] * decay_f < decay_thr (the light decay
// of a fused frame), else the kill flags of a voxel pass (cleared here).  The dead blocks leave the index here (hash
// tombstone / dense table / slot key); the last chunk books the tombstones and raises *rebuild when they exceed a quarter
// of the table -- the caller follows with k_hash_clear_if / k_hash_insert_live_if on that flag.
template <bool WMAX>
__device__ inline void live_compact_big_body(const LayerDev& L, uint8_t* __restrict__ kill, u64* lb, unsigned tag, int* rebuild,
                                             int* snap6, float decay_f, float decay_thr, int* lds, int* sh, int w, int nwg) {
  const bool dense = L.dense != nullptr;
  const int n = L.ctr[0];
  const int free0 = L.ctr[1];
  const int i0 = w * kBigChunk + (int)threadIdx.x * 4;
  int slot[4];
  bool dead[4];
  int keep = 0, nd = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    slot[q] = -1;
    dead[q] = false;
    if (i0 + q < n) {
      slot[q] = L.live[i0 + q];
      if (WMAX) {
        const float wm = L.wmax[slot[q]] * decay_f;
        dead[q] = wm < decay_thr;
        if (L.epoch && !dead[q]) {
          // lazy decay (LayerDev::epoch): the voxels of this block stay behind; its summaries are brought forward here, one
          // multiplication per decay like the voxels' own -- max / min commute with the monotone W -> W f exactly
          L.wmax[slot[q]] = wm;
          const float wn = L.wmin[slot[q]] * decay_f;
          L.wmin[slot[q]] = wn;
          if (!(wn > 1e-4f) && L.block_free[slot[q]]) L.block_free[slot[q]] = 0;  // all-free needs every W > 1e-4
        }
      } else if (kill[i0 + q]) {
        dead[q] = true;
        kill[i0 + q] = 0;
      }
      if (dead[q]) nd++;
      else keep++;
    }
  }
  int ea, eb, ta, tb;
  block_excl_scan2<4>(keep, nd, lds, ea, eb, ta, tb);
  lookback_exclusive(lb, tag, w, nwg, ta, tb, lds, sh, &L.ctr[3]);
  int wk = sh[0] + ea, wd = free0 + sh[1] + eb;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (slot[q] < 0) continue;
    if (dead[q]) {
      L.free_stack[wd++] = slot[q];
      const u64 key = L.slot_key[slot[q]];
      if (!dense) hash_erase(L, key);
      dense_set(L, key, 0);
      L.slot_key[slot[q]] = kEmptyKey;
    } else {
      L.live[wk++] = slot[q];
    }
  }
  if (w == nwg - 1 && threadIdx.x == 0) {
    const int n_live = sh[0] + ta, n_dead = sh[1] + tb;
    const int n_tomb = dense ? 0 : L.ctr[4] + n_dead;
    const bool rb = !dense && (unsigned)n_tomb * 4u > L.hmask + 1u;
    L.ctr[0] = n_live;
    L.ctr[1] = free0 + n_dead;
    // The count stays where it is until the rebuild has RUN (k_hash_insert_live_if zeroes it): the conditional rebuild launches
    // follow only every kRebuildEvery-th compaction, and a request that has not been served yet is raised again by the next
    // compaction from the same count.
    L.ctr[4] = n_tomb;
    if (rebuild) *rebuild = rb ? 1 : 0;
    if (snap6) *snap6 = n_live;
    if (L.hint_live) *L.hint_live = n_live;
  }
}

}  // namespace mmf

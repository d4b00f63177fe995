// bin_sort.h — the slot order and the class lists in two launches (small and medium registries).
//
// The radix sort of kernels.h needs eight dependent launches for a 17-bit key (servant scan,
// slot generation, then histogram / column scan / scatter per digit) and every one of them is
// launch-latency-bound below ~1M slots. This path gets the same order — (key, registry
// index), the reference's arg-min with first-wins ties, task_dispatcher.cc:440-447 — and the
// same per-class lists out of two, and it has no launch-wide prefix scan at all:
//
//   k_front_bins   three kinds of workgroups that need nothing from each other:
//     * bin boundaries: the key space is cut into B equal bins; "how many slots of class c
//       have a key below bin j" is a sum of per-servant closed forms
//       (first_slot_not_below_direct32, dispatch_core.h) — so the START of every bin in the
//       global order and in every class list is known without a histogram or a scan. The
//       workgroup of the last boundary (= all slots) also leaves the class list starts, the
//       slot count and the per-batch counter reset behind.
//     * slot tiles: a tile is a group of G consecutive servants; its workgroup adds up the slot
//       counts of the servants BEFORE the tile itself (a few thousand closed forms — cheaper
//       than waiting for a scan), generates the tile's slots (key, class, owner) from the
//       servants' columns in LDS, and writes them to the staging array, at the tile's own place
//       (the slots' generation indexes), grouped by bin and in slot order inside a group
//       (ballot-ranked like the radix scatter, no atomics). A run table says where each bin's
//       group starts and how long it is.
//     * request classification (task_classify_block): own servants are left as servant indexes
//       (slot_base is being written in this very launch), consuming counts go out per wave.
//   k_bin_sort     one workgroup per bin, everything in LDS: the bin's runs are read in tile
//                  (= slot) order through the run table, stable counting passes over the key
//                  bits below the bin make that (key, slot) order — the slot's global rank is
//                  bin start + position —, and a stable partition by class (wave ballots) gives
//                  its place in its class list (bin's start in that list + earlier slots of the
//                  class). One more workgroup computes the chunk prefix of the consuming counts.
//
// Real pools are full of ties (a cluster has a handful of machine types, and every machine of a
// type offers the same utilisation values: cfg2's 126k slots hold a dozen groups of ~1000 equal
// keys), so bins cannot be made small by making them many, and a comparison network over a
// 1300-record bin costs ~40 us; the counting passes do not care.
// A bin that does not fit the LDS buffers (kBinCap records) makes the batch report
// DeviceParams::window_miss; the host repeats it with the radix sort and stays with that until
// the registry changes structure. Exactness never depends on the bins: every slot lands in the
// bin its key names, bins are ordered completely, and the starts are exact counts.
#ifndef YADCC_AMD_BIN_SORT_H_
#define YADCC_AMD_BIN_SORT_H_

#include "kernels.h"

namespace ydc {

constexpr uint32_t kMaxBins = 2048;
constexpr uint32_t kBinCap = 4096;       // records one workgroup of k_bin_sort orders
constexpr uint32_t kBinMaxTiles = 2048;  // slot tiles (k_bin_sort keeps two words per tile in LDS)
constexpr uint32_t kBinMaxGroup = 32;    // servants per slot tile, at most
constexpr uint32_t kBinMaxServants = 4096;  // (every slot tile adds up the servants before it)

struct BinTable {
  uint32_t n_bins;  // B, a power of two <= kMaxBins
  uint32_t shift;   // bin of a slot = key >> shift (B << shift == 2^key_bits)
  // [(B + 1) * (C + 1)] row j: slots whose bin is below j — entries 0..C-1 per class, entry C
  // of all classes. Row 0 is zero, row B the class sizes.
  uint32_t* base;
  // [n_tiles * B] the tile's slots of the bin: (offset inside the tile's staging region << 16) |
  // count (a tile holds at most 2048 slots).
  uint32_t* runs;
  uint32_t group;   // G: servants per slot tile at most (kBinMaxGroup)
  uint32_t n_tiles;
  // Slot tile t = servants [tile_start[t], tile_start[t + 1]): cut by the host so that every
  // tile holds about the same number of slots (by the servants' static bound min(max_tasks,
  // nproc); at most `group` servants and 2048 slots) — the launch lasts as long as its fullest
  // tile. tile_base[t] = slots of the servants before the tile (written by the tile itself).
  const uint32_t* tile_start;
  uint32_t* tile_base;
};

// Workgroup of boundary j (1 .. B): row j of BinTable::base. With a handful of classes the
// per-class sums are reduced in registers and across the wave before they touch LDS.
__device__ __forceinline__ void front_bin_boundary(const ServantTable& sv, uint32_t n_classes,
                                                   const PartTable& parts, uint32_t cap_bits,
                                                   uint32_t comp_shift, const BinTable& bt, uint32_t j,
                                                   uint32_t max_slots, uint32_t* slot_base,
                                                   uint32_t* cls_begin, DeviceParams* prm) {
  extern __shared__ uint32_t fsm[];  // n_classes + 1: entry n_classes = all classes
  uint32_t* cls_cnt = fsm;
  const uint32_t row = n_classes + 1;
  for (uint32_t c = threadIdx.x; c < row; c += blockDim.x) cls_cnt[c] = 0;
  __syncthreads();
  const bool few = n_classes <= 4;
  const bool narrow = cap_bits <= 10;  // 32-bit arithmetic throughout (keys are below 2^32 here)
  const uint64_t K = (uint64_t)j << bt.shift;
  uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, all = 0;
#pragma unroll 1
  for (uint32_t s0 = threadIdx.x; s0 < sv.n; s0 += 4 * blockDim.x) {
    // Four servants per round, their column loads in flight together (clamped index).
    uint32_t cls[4], run[4], nproc[4], load[4], mt[4], fl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t s = min(s0 + u * blockDim.x, sv.n - 1);
      cls[u] = sv.class_of[s];
      run[u] = sv.running[s];
      nproc[u] = sv.nproc[s];
      load[u] = sv.load[s];
      mt[u] = sv.max_tasks[s];
      fl[u] = sv.flags[s];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (s0 + u * blockDim.x >= sv.n || cls[u] == kNone) continue;
      uint32_t cnt;
      if (j == bt.n_bins) {  // everything is below the end of the key space
        cnt = servant_slot_count(nproc[u], load[u], mt[u], run[u], fl[u]);
      } else if (narrow) {
        const uint32_t part_key = parts.n_parts > 1 ? parts.cls_comp[cls[u]] << comp_shift : 0u;
        cnt = first_slot_not_below_direct32(nproc[u], load[u], mt[u], run[u], fl[u], part_key, (uint32_t)K,
                                            cap_bits) - run[u];
      } else {
        const uint64_t part_key = parts.n_parts > 1 ? (uint64_t)parts.cls_comp[cls[u]] << comp_shift : 0ull;
        cnt = first_slot_not_below_direct(nproc[u], load[u], mt[u], run[u], fl[u], part_key, K, cap_bits) - run[u];
      }
      all += cnt;
      if (few) {
        a0 += cls[u] == 0 ? cnt : 0u;
        a1 += cls[u] == 1 ? cnt : 0u;
        a2 += cls[u] == 2 ? cnt : 0u;
        a3 += cls[u] == 3 ? cnt : 0u;
      } else if (cnt) {
        atomicAdd(&cls_cnt[cls[u]], cnt);
      }
    }
  }
  auto wave_add = [&](uint32_t v, uint32_t slot) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&cls_cnt[slot], v);
  };
  wave_add(all, n_classes);
  if (few) {
    wave_add(a0, 0);
    if (n_classes > 1) wave_add(a1, 1);
    if (n_classes > 2) wave_add(a2, 2);
    if (n_classes > 3) wave_add(a3, 3);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < row; i += blockDim.x) bt.base[(size_t)j * row + i] = cls_cnt[i];
  if (j != bt.n_bins) return;
  // The last boundary has the class sizes: what k_servant_scan leaves behind besides slot_base.
  for (uint32_t k = threadIdx.x; k <= n_classes; k += blockDim.x) bt.base[k] = 0;  // row 0
  if (threadIdx.x < 64) prm->n_changed[threadIdx.x] = prm->n_sampled[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    const uint32_t m = cls_cnt[n_classes];
    slot_base[sv.n] = m;
    prm->overflow = m > max_slots ? 1u : 0u;
    prm->n_slots = m > max_slots ? 0u : m;
    prm->reserved0 = 0;
    prm->zone_rows = 0;
    prm->chunk_sims = 0;
    prm->granted = 0;
    prm->consuming = 0;
    prm->rank_offset = 0;
    prm->window_miss = 0;
    prm->batch_seq += 1;
    uint32_t acc = 0;
    for (uint32_t c = 0; c < n_classes; ++c) {
      cls_begin[c] = acc;
      acc += cls_cnt[c];
    }
    cls_begin[n_classes] = acc;
    if (parts.n_parts > 1) {
      uint32_t cnt[16];  // kMaxComponents
      for (uint32_t g = 0; g < parts.n_parts; ++g) cnt[g] = 0;
      for (uint32_t c = 0; c < n_classes; ++c) cnt[parts.cls_comp[c]] += cls_cnt[c];
      uint32_t a2s = 0;
      for (uint32_t g = 0; g < parts.n_parts; ++g) {
        parts.rank_base[g] = a2s;
        a2s += cnt[g];
      }
      parts.rank_base[parts.n_parts] = a2s;
    }
  }
}

// Workgroup of slot tile `tile`: servants [tile_start[tile], tile_start[tile + 1]).
// Values are (class << gbits) | slot (gbits == 0: one class, the slot alone).
__device__ __forceinline__ void front_slot_tile(const ServantTable& sv, uint32_t tile, uint32_t cap_bits,
                                                uint32_t comp_shift, uint32_t gbits,
                                                const uint32_t* cls_comp, uint32_t n_parts,
                                                const BinTable& bt, uint32_t max_slots,
                                                uint32_t* slot_base, uint32_t* owner, uint2* stage) {
  extern __shared__ uint32_t fsm[];
  __shared__ uint32_t lds[17];
  const uint32_t B = bt.n_bins, G = bt.group;
  uint32_t* start = fsm;              // [B] the bin's group inside the tile's region
  uint32_t* wcnt_all = fsm + B;       // [4 waves][B]
  uint32_t* col = fsm + 5 * B;        // [6][G] the tile's servants: class, nproc, load, max_tasks, running, flags
  uint32_t* lbase = col + 6 * G;      // [G + 1] local prefix of their slot counts
  const uint32_t s0 = bt.tile_start[tile], ns = bt.tile_start[tile + 1] - s0;
  // The tile's own servants (loads issued ahead of the long loop below).
  if (threadIdx.x < ns) {
    const uint32_t s = s0 + threadIdx.x;
    const uint32_t cls = sv.class_of[s], nproc = sv.nproc[s], load = sv.load[s], mt = sv.max_tasks[s],
                   run = sv.running[s], fl = sv.flags[s];
    col[0 * G + threadIdx.x] = cls;
    col[1 * G + threadIdx.x] = nproc;
    col[2 * G + threadIdx.x] = load;
    col[3 * G + threadIdx.x] = mt;
    col[4 * G + threadIdx.x] = run;
    col[5 * G + threadIdx.x] = fl;
    lbase[threadIdx.x] = cls == kNone ? 0u : servant_slot_count(nproc, load, mt, run, fl);
  }
  // Slots of the servants before the tile (its first generation index).
  uint32_t mine = 0;
#pragma unroll 1
  for (uint32_t sa = threadIdx.x; sa < s0; sa += 8 * blockDim.x) {
    uint32_t cls[8], run[8], nproc[8], load[8], mt[8], fl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t s = min(sa + u * blockDim.x, sv.n - 1);
      cls[u] = sv.class_of[s];
      run[u] = sv.running[s];
      nproc[u] = sv.nproc[s];
      load[u] = sv.load[s];
      mt[u] = sv.max_tasks[s];
      fl[u] = sv.flags[s];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (sa + u * blockDim.x < s0 && cls[u] != kNone)
        mine += servant_slot_count(nproc[u], load[u], mt[u], run[u], fl[u]);
  }
  for (uint32_t d = threadIdx.x; d < 4 * B; d += blockDim.x) wcnt_all[d] = 0;
  uint32_t base;
  (void)block_exclusive_scan(mine, lds, &base);  // (its barriers also publish col / lbase)
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t i = 0; i < ns; ++i) {
      const uint32_t t = lbase[i];
      lbase[i] = acc;
      acc += t;
    }
    lbase[ns] = acc;
  }
  __syncthreads();
  // (a registry that overflows the workspace fails the batch: generate nothing)
  const uint32_t total = (uint64_t)base + lbase[ns] <= max_slots ? lbase[ns] : 0u;
  if (threadIdx.x < ns) slot_base[s0 + threadIdx.x] = base + lbase[threadIdx.x];
  if (threadIdx.x == 0) bt.tile_base[tile] = base;
  uint32_t bbits = 0;
  while ((1u << bbits) < B) ++bbits;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t items = (total + blockDim.x - 1) / blockDim.x;  // <= kSortItems: a tile holds <= 2048 slots
  const uint32_t wave_span = items * 64;
  const uint64_t lt_mask = (1ull << lane) - 1;
  uint32_t* wcnt = wcnt_all + wave * B;
  uint32_t key[kSortItems], val[kSortItems], rank[kSortItems];
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t x = wave * wave_span + j * 64 + lane;  // slot of the tile, generation order
    const bool valid = (uint32_t)j < items && x < total;
    key[j] = val[j] = rank[j] = 0;
    uint32_t d = 0;
    if (valid) {
      uint32_t lo = 0, hi = ns;  // lbase[lo] <= x < lbase[hi]
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (lbase[mid] <= x) lo = mid; else hi = mid;
      }
      const uint32_t cls = col[0 * G + lo], nproc = col[1 * G + lo], load = col[2 * G + lo],
                     mt = col[3 * G + lo], fl = col[5 * G + lo];
      const uint32_t r = col[4 * G + lo] + (x - lbase[lo]);
      const uint32_t g = base + x;
      owner[g] = s0 + lo;
      uint64_t k64 = slot_key_exact(slot_tier(nproc, fl, r), r, slot_capacity(nproc, load, mt, r), cap_bits);
      if (n_parts > 1) k64 |= (uint64_t)cls_comp[cls] << comp_shift;
      key[j] = (uint32_t)k64;
      val[j] = gbits ? (cls << gbits) | g : g;
      d = key[j] >> bt.shift;
    }
    if ((uint32_t)j < items) {  // (uniform)
      uint64_t peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 11; ++b) {  // kMaxBins == 2^11
        if ((uint32_t)b < bbits) {
          const uint64_t m = __ballot((d >> b) & 1u);
          peers &= ((d >> b) & 1u) ? m : ~m;
        }
      }
      uint32_t before = 0;
      if (valid) before = wcnt[d];
      rank[j] = before + (uint32_t)__popcll(peers & lt_mask);
      // (every lane of the wave has read wcnt[d] in the instruction above)
      if (valid && (peers & lt_mask) == 0) wcnt[d] = before + (uint32_t)__popcll(peers);
    }
  }
  __syncthreads();
  // Per bin: earlier waves' slots; the bins' groups one behind the other in the tile's region.
  {
    const uint32_t per = (B + blockDim.x - 1) / blockDim.x;
    const uint32_t d0 = threadIdx.x * per, d1 = min(B, d0 + per);
    uint32_t sum = 0;
    for (uint32_t d = d0; d < d1; ++d) {
      uint32_t off = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t t = wcnt_all[w * B + d];
        wcnt_all[w * B + d] = off;
        off += t;
      }
      start[d] = off;  // (the count, for a moment)
      sum += off;
    }
    uint32_t tot;
    uint32_t acc = block_exclusive_scan(sum, lds, &tot);
    for (uint32_t d = d0; d < d1; ++d) {
      const uint32_t cnt = start[d];
      start[d] = acc;
      bt.runs[(size_t)tile * B + d] = (acc << 16) | cnt;
      acc += cnt;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t x = wave * wave_span + j * 64 + lane;
    if ((uint32_t)j < items && x < total) {
      const uint32_t d = key[j] >> bt.shift;
      stage[base + start[d] + wcnt[d] + rank[j]] = make_uint2(key[j], val[j]);
    }
  }
}

// Workgroups [0, n_tiles): slot tiles; [n_tiles, n_tiles + B): bin boundaries 1 .. B; the rest: requests.
__global__ __launch_bounds__(256) void k_front_bins(ServantTable sv, uint32_t n_classes, uint32_t max_slots,
                                                    uint32_t* slot_base, uint32_t* cls_begin,
                                                    PartTable parts, DeviceParams* prm, uint32_t cap_bits,
                                                    uint32_t comp_shift, uint32_t gbits, BinTable bt,
                                                    uint32_t* owner, uint2* stage, ClassifyArgs ca) {
  const uint32_t blk = blockIdx.x;
#ifdef YDC_PHASE_PROBE
  if (threadIdx.x == 0 && blk < 2400) ydc_phase_probe[40000 + blk * 2] = wall_clock64();
  struct ProbeEnd {
    uint32_t b;
    __device__ ~ProbeEnd() {
      if (threadIdx.x == 0 && b < 2400) ydc_phase_probe[40000 + b * 2 + 1] = wall_clock64();
    }
  } probe_end{blk};
#endif
  // Longest first: a launch of ~1200 workgroups takes ~6 us to hand them all out (in index
  // order), and a slot tile works for ~8 us, a bin boundary for ~6, a request block for ~3 —
  // the tiles used to start behind the 512 boundaries and ended the launch at 14.5 us.
  if (blk < bt.n_tiles) {
    front_slot_tile(sv, blk, cap_bits, comp_shift, gbits, parts.cls_comp, parts.n_parts, bt, max_slots,
                    slot_base, owner, stage);
  } else if (blk < bt.n_bins + bt.n_tiles) {
    front_bin_boundary(sv, n_classes, parts, cap_bits, comp_shift, bt, blk - bt.n_tiles + 1, max_slots,
                       slot_base, cls_begin, prm);
  } else {
    task_classify_block(ca, blk - bt.n_bins - bt.n_tiles, prm);
  }
}

// ---------------------------------------------------------------------------
// k_bin_sort: workgroup j < B orders bin j and writes the slots' places in the global order
// (rank_to_g[rank] = slot) and in the class lists (list[position] = {rank, slot}); workgroup B
// (if launched) is the chunk prefix of the consuming counts.
//
// No comparison sort. (1) The bin's records are groups of the tiles' staging regions, each in
// slot order (k_front_bins); the run table says where they are, so reading them tile by tile IS
// reading them in slot order. (2) One stable counting pass per 8 bits of the key below the
// bin's own bits (ballot-ranked like the radix scatter, all in LDS) makes that (key, slot)
// order — the global order inside the bin. (3) A stable partition by class gives the places in
// the class lists. A record lives in LDS as one word: key bits below the bin | slot | class.
// ---------------------------------------------------------------------------
constexpr uint32_t kBinThreads = 1024;
constexpr uint32_t kBinWaves = kBinThreads / 64;
constexpr uint32_t kBinRounds = kBinCap / kBinThreads;  // records per thread
// LDS words: two record buffers, the counter table ([waves][256]; the tile tables before that),
// digit totals / starts, class-list cursors.
constexpr uint32_t kBinLdsWords = 2 * kBinCap + kBinWaves * 256 + 256 + 256;
static_assert(2 * kBinMaxTiles + 1 <= kBinWaves * 256 + 256, "tile tables alias the counter table");

struct BinSortArgs {
  const uint2* stage;
  BinTable bt;
  uint32_t n_classes, gbits;
  uint32_t slot_bits, cls_bits;  // of the LDS word (cls_bits == 0: one class)
  const uint32_t* slot_base;     // (a tile's staging region starts at its first servant's slot_base)
  const uint32_t* cls_begin;     // [C + 1]
  uint2* list;                   // class lists: {global rank, slot}
  uint32_t* rank_to_g;
  // [ceil(M / 64) * C] level table: entry (m, c) = position in class c's list of its first slot
  // whose global rank is >= 64 m. The matching kernel's level guesses (the class states after
  // the n lowest-ranked slots are gone) read it instead of searching the lists: one lookup and
  // one 64-entry window per class.
  uint32_t* level_tab;
};

__global__ __launch_bounds__(kBinThreads, 8) void k_bin_sort(BinSortArgs a, DeviceParams* prm, PrefixArgs pa) {
  extern __shared__ __attribute__((aligned(16))) uint32_t bsm[];
  __shared__ uint32_t lds[17];
#ifdef YDC_PHASE_PROBE
#define YDC_BPROBE(slot)                                                                              \
  do {                                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 2100) ydc_phase_probe[46000 + blockIdx.x * 6 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define YDC_BPROBE(slot) do { } while (0)
#endif
  YDC_BPROBE(0);
  if (blockIdx.x == a.bt.n_bins) {
    chunk_prefix_block(pa, prm);
    YDC_BPROBE(5);
    return;
  }
  uint32_t* buf0 = bsm;
  uint32_t* buf1 = bsm + kBinCap;
  uint32_t* tab = bsm + 2 * kBinCap;           // [kBinWaves][256]
  uint32_t* dtot = tab + kBinWaves * 256;      // [256]
  uint32_t* cbase = dtot + 256;                // [256] next free position of every class list
  uint32_t* rsrc = tab;                        // [n_tiles] (step 1 only) where the tile's run sits in the staging array
  uint32_t* rstart = tab + kBinMaxTiles;       // [n_tiles + 1] first record of the tile's run (runs into dtot)
  const uint32_t C = a.n_classes, row = C + 1, j = blockIdx.x, T = a.bt.n_tiles;
  const uint32_t lo = a.bt.base[(size_t)j * row + C], hi = a.bt.base[(size_t)(j + 1) * row + C];
  const uint32_t n = hi - lo;
  // The bin's runs: tiles 2 * threadIdx.x and 2 * threadIdx.x + 1 (T <= 2 * kBinThreads).
  uint32_t r0 = 0, r1 = 0, b0 = 0, b1 = 0;
  const uint32_t t0 = 2 * threadIdx.x;
  if (t0 < T) {
    r0 = a.bt.runs[(size_t)t0 * a.bt.n_bins + j];
    b0 = a.bt.tile_base[t0];
  }
  if (t0 + 1 < T) {
    r1 = a.bt.runs[(size_t)(t0 + 1) * a.bt.n_bins + j];
    b1 = a.bt.tile_base[t0 + 1];
  }
  if (n == 0 || prm->n_slots == 0) return;
  if (n > kBinCap) {
    if (threadIdx.x == 0) prm->window_miss = 1;  // the host repeats the batch with the radix sort
    return;
  }
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1;
  const uint32_t gmask = a.gbits ? (1u << a.gbits) - 1 : 0xFFFFFFFFu;
  const uint32_t sbits = a.slot_bits + a.cls_bits;  // word = key bits below the bin << sbits | slot << cls_bits | class
  const uint32_t kmask = a.bt.shift ? (1u << a.bt.shift) - 1 : 0u;
  for (uint32_t c = threadIdx.x; c < C; c += blockDim.x) cbase[c] = a.cls_begin[c] + a.bt.base[(size_t)j * row + c];
  // ---- (1) the runs, tile by tile
  {
    uint32_t total;
    const uint32_t ex = block_exclusive_scan((r0 & 0xFFFFu) + (r1 & 0xFFFFu), lds, &total);
    if (t0 < T) {
      rstart[t0] = ex;
      rsrc[t0] = b0 + (r0 >> 16);
    }
    if (t0 + 1 < T) {
      rstart[t0 + 1] = ex + (r0 & 0xFFFFu);
      rsrc[t0 + 1] = b1 + (r1 >> 16);
    }
    if (threadIdx.x == 0) rstart[T] = total;
    __syncthreads();
    if (total != n) {  // (cannot happen: the closed forms and the generated slots disagree)
      if (threadIdx.x == 0) prm->window_miss = 1;
      return;
    }
  }
  YDC_BPROBE(1);  // run table read, tile starts scanned
#pragma unroll
  for (int k = 0; k < (int)kBinRounds; ++k) {
    const uint32_t i = k * kBinThreads + threadIdx.x;
    if (i < n) {
      uint32_t tl = 0, th = T;  // rstart[tl] <= i < rstart[th]
      while (th - tl > 1) {
        const uint32_t mid = (tl + th) >> 1;
        if (rstart[mid] <= i) tl = mid; else th = mid;
      }
      const uint2 r = a.stage[rsrc[tl] + (i - rstart[tl])];
      const uint32_t slot = r.y & gmask, cls = a.gbits ? r.y >> a.gbits : 0u;
      buf0[i] = ((r.x & kmask) << sbits) | (slot << a.cls_bits) | cls;
    }
  }
  __syncthreads();
  YDC_BPROBE(2);  // records staged in LDS
  // ---- (2) stable counting passes over the key bits below the bin, 8 at a time
  uint32_t* src = buf0;
  uint32_t* dst = buf1;
  const uint32_t R = (n + kBinThreads - 1) / kBinThreads;  // rounds: wave w owns [w * R * 64, (w + 1) * R * 64)
  for (uint32_t sh = 0; sh < a.bt.shift; sh += 8) {
    const uint32_t dbits = min(8u, a.bt.shift - sh);
    for (uint32_t t = threadIdx.x; t < kBinWaves * 256; t += blockDim.x) tab[t] = 0;
    __syncthreads();
    uint32_t* wtab = tab + wave * 256;
    uint32_t e[kBinRounds], dig[kBinRounds], rk[kBinRounds];
#pragma unroll
    for (int r = 0; r < (int)kBinRounds; ++r) {
      e[r] = dig[r] = rk[r] = 0;
      if ((uint32_t)r < R) {  // (uniform)
        const uint32_t idx = (wave * R + r) * 64 + lane;
        const bool valid = idx < n;
        uint32_t d = 0;
        if (valid) {
          e[r] = src[idx];
          d = (e[r] >> (sbits + sh)) & ((1u << dbits) - 1);
        }
        dig[r] = d;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          if ((uint32_t)b < dbits) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
          }
        }
        uint32_t before = 0;
        if (valid) before = wtab[d];
        rk[r] = before + (uint32_t)__popcll(peers & lt_mask);
        if (valid && (peers & lt_mask) == 0) wtab[d] = before + (uint32_t)__popcll(peers);
      }
    }
    __syncthreads();
    uint32_t mine = 0;  // records with digit threadIdx.x (threads 0..255)
    if (threadIdx.x < 256) {
      uint32_t off = 0;
#pragma unroll
      for (int w = 0; w < (int)kBinWaves; ++w) {
        const uint32_t t = tab[w * 256 + threadIdx.x];
        tab[w * 256 + threadIdx.x] = off;
        off += t;
      }
      mine = off;
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(mine, lds, &total);
    if (threadIdx.x < 256) dtot[threadIdx.x] = ex;  // start of the digit
    __syncthreads();
#pragma unroll
    for (int r = 0; r < (int)kBinRounds; ++r) {
      if ((uint32_t)r < R) {
        const uint32_t idx = (wave * R + r) * 64 + lane;
        if (idx < n) dst[dtot[dig[r]] + wtab[dig[r]] + rk[r]] = e[r];
      }
    }
    __syncthreads();
    uint32_t* t = src;
    src = dst;
    dst = t;
  }
  YDC_BPROBE(3);  // counting passes done
  // ---- (3) places. Positions are walked 1024 at a time; within a round (wave, lane) order ==
  // position order, so "earlier slots of the same class" = earlier rounds (cbase) + earlier
  // waves of this round (tab) + lower lanes of this wave (ballot match on the class bits).
  for (uint32_t t = threadIdx.x; t < kBinWaves * 256; t += blockDim.x) tab[t] = 0;
  __syncthreads();
  const uint32_t cmask = a.cls_bits ? (1u << a.cls_bits) - 1 : 0u;
  // (The walk is aligned to the global rank: wave w of a round starts at a rank that is a
  // multiple of 64, so the level table's entries are the counts at wave starts.)
  const uint32_t skew = lo & 63u;
  for (uint32_t q0 = 0; q0 < n + skew; q0 += blockDim.x) {
    const uint32_t q = q0 + threadIdx.x;
    const bool valid = q >= skew && q - skew < n;
    const uint32_t p = q - skew;
    uint32_t slot = 0, cls = 0;
    if (valid) {
      const uint32_t w = src[p];
      slot = (w >> a.cls_bits) & ((a.slot_bits < 32 ? 1u << a.slot_bits : 0u) - 1u);
      cls = w & cmask;
    }
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if ((uint32_t)b < a.cls_bits) {
        const uint64_t m = __ballot((cls >> b) & 1u);
        peers &= ((cls >> b) & 1u) ? m : ~m;
      }
    }
    const uint32_t before_in_wave = (uint32_t)__popcll(peers & lt_mask);
    if (valid && before_in_wave == 0) tab[wave * 256 + cls] = (uint32_t)__popcll(peers);
    __syncthreads();
    {
      // Level table: this wave's first rank, if it belongs to this bin.
      const uint32_t r0 = lo - skew + q0 + wave * 64;
      if (a.level_tab && r0 >= lo && r0 < hi) {
        for (uint32_t c = lane; c < C; c += 64) {
          uint32_t pos = cbase[c];
          for (uint32_t w = 0; w < wave; ++w) pos += tab[w * 256 + c];
          a.level_tab[(size_t)(r0 >> 6) * C + c] = pos;
        }
      }
    }
    if (valid) {
      uint32_t pos = cbase[cls] + before_in_wave;
      for (uint32_t w = 0; w < wave; ++w) pos += tab[w * 256 + cls];
      const uint32_t rank = lo + p;
      a.list[pos] = make_uint2(rank, slot);
      a.rank_to_g[rank] = slot;
    }
    __syncthreads();
    if (threadIdx.x < C) {
      uint32_t sum = 0;
#pragma unroll
      for (int w = 0; w < (int)kBinWaves; ++w) {
        sum += tab[w * 256 + threadIdx.x];
        tab[w * 256 + threadIdx.x] = 0;
      }
      cbase[threadIdx.x] += sum;
    }
    __syncthreads();
  }
  YDC_BPROBE(4);  // places written
}

}  // namespace ydc
#endif  // YADCC_AMD_BIN_SORT_H_

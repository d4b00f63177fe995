// bin_sort.h — the slot order in three launches (small and medium registries).
//
// The radix sort of kernels.h needs seven dependent launches for a 17-bit key (slot
// generation, then histogram / column scan / scatter per digit) and every one of them is
// launch-latency-bound below ~1M slots. This path gets the same order — (key, registry
// index), the reference's arg-min with first-wins ties, task_dispatcher.cc:440-447 — and the
// same per-class lists out of three:
//
//   k_servant_scan_bins  workgroup 0: the servant scan of k_servant_scan. The others: the key
//                        space is cut into B equal bins; "how many slots of class c have a key
//                        below bin j" is a sum of per-servant closed forms
//                        (first_slot_not_below_direct, dispatch_core.h) — so the START of every
//                        bin in the global order and in every class list is known before a
//                        single slot has been generated. No histogram, no scan.
//   k_slot_bin           thread per slot (as k_slot_gen): key, class, owner; a tile's records of
//                        a bin go into the bin's region of the staging array as one run in slot
//                        order, at whatever position the bin's arrival counter hands out (one
//                        global atomic per tile and bin). Extra workgroups classify the
//                        requests, as in k_slot_gen.
//   k_bin_sort           one workgroup per bin, everything in LDS: the runs are put into tile
//                        (= slot) order by counting, stable counting passes over the key bits
//                        below the bin make that (key, slot) order — the slot's global rank is
//                        bin start + position —, and a stable partition by class (wave ballots,
//                        the ranking of the radix scatter) gives its place in its class list
//                        (bin's start in that list + earlier slots of the class). One more
//                        workgroup computes the chunk prefix of the consuming counts.
//
// Real pools are full of ties (a cluster has a handful of machine types, and every machine of a
// type offers the same utilisation values: cfg2's 126k slots hold a dozen groups of ~1000 equal
// keys), so bins cannot be made small by making them many, and a comparison network over a
// 1300-record bin costs ~40 us; the counting passes do not care.
// A bin that does not fit the LDS buffers (kBinCap records) makes the batch report
// DeviceParams::window_miss; the host repeats it with the radix sort and stays with that until
// the registry changes structure.
// Exactness never depends on the bins: every slot lands in the bin its key names, bins are
// sorted completely, and the starts are exact counts.
#ifndef YADCC_AMD_BIN_SORT_H_
#define YADCC_AMD_BIN_SORT_H_

#include "kernels.h"

namespace ydc {

constexpr uint32_t kMaxBins = 2048;
constexpr uint32_t kBinCap = 4096;  // records one workgroup of k_bin_sort sorts (8 B each in LDS)

struct BinTable {
  uint32_t n_bins;  // B, a power of two <= kMaxBins
  uint32_t shift;   // bin of a slot = key >> shift (B << shift == 2^key_bits)
  // [(B + 1) * (C + 1)] row j: slots whose bin is below j — entries 0..C-1 per class, entry C
  // of all classes. Row 0 is zero, row B the class sizes.
  uint32_t* base;
  uint32_t* fill;   // [B] records that have arrived in the bin (k_slot_bin)
};

// Workgroup w >= 1 of k_servant_scan_bins: boundaries j = (w - 1) * per + 1 .. (w - 1) * per + per
// (up to B), row j of BinTable::base each. The workgroup's 2^sub_shift-thread parts take the
// boundaries in turn: a boundary is one pass of such a part over the servants. With a handful
// of classes the per-class sums are reduced in registers and across the wave before they touch
// LDS (one LDS atomic per wave, class and boundary instead of one per servant and boundary on
// two or three addresses).
__device__ __forceinline__ void bin_count_block(const ServantTable& sv, uint32_t n_classes,
                                                const PartTable& parts, uint32_t cap_bits,
                                                uint32_t comp_shift, const BinTable& bt, uint32_t group,
                                                uint32_t per, uint32_t sub_shift) {
  extern __shared__ uint32_t cls_cnt[];  // per * (n_classes + 1): entry n_classes = all classes
  const uint32_t row = n_classes + 1;
  for (uint32_t c = threadIdx.x; c < per * row; c += blockDim.x) cls_cnt[c] = 0;
  __syncthreads();
  const uint32_t j0 = group * per + 1;
  const uint32_t nb = min(per, bt.n_bins + 1 - j0);
  const bool few = n_classes <= 4;
  const bool narrow = cap_bits <= 10;  // 32-bit arithmetic throughout (keys are below 2^32 here)
  const uint32_t sub = threadIdx.x >> sub_shift, st = threadIdx.x & ((1u << sub_shift) - 1);
  const uint32_t stride = 1u << sub_shift;
#pragma unroll 1
  for (uint32_t b = sub; b < nb; b += blockDim.x >> sub_shift) {
    const uint32_t j = j0 + b;
    const uint64_t K = (uint64_t)j << bt.shift;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, all = 0;
#pragma unroll 1
    for (uint32_t s0 = st; s0 < sv.n; s0 += 2 * stride) {
      // Two servants per round, their column loads in flight together (clamped index).
      uint32_t cls[2], run[2], nproc[2], load[2], mt[2], fl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t s = min(s0 + u * stride, sv.n - 1);
        cls[u] = sv.class_of[s];
        run[u] = sv.running[s];
        nproc[u] = sv.nproc[s];
        load[u] = sv.load[s];
        mt[u] = sv.max_tasks[s];
        fl[u] = sv.flags[s];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (s0 + u * stride >= sv.n || cls[u] == kNone) continue;
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
          atomicAdd(&cls_cnt[b * row + cls[u]], cnt);
        }
      }
    }
    auto wave_add = [&](uint32_t v, uint32_t slot) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
      if ((threadIdx.x & 63) == 0 && v) atomicAdd(&cls_cnt[b * row + slot], v);
    };
    wave_add(all, n_classes);
    if (few) {
      wave_add(a0, 0);
      if (n_classes > 1) wave_add(a1, 1);
      if (n_classes > 2) wave_add(a2, 2);
      if (n_classes > 3) wave_add(a3, 3);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < nb * row; i += blockDim.x) bt.base[(size_t)j0 * row + i] = cls_cnt[i];
}

__global__ __launch_bounds__(1024, 8) void k_servant_scan_bins(ServantTable sv, uint32_t n_classes,
                                                            uint32_t max_slots, uint32_t* slot_base,
                                                            uint32_t* cls_begin, uint32_t* chunk_consuming,
                                                            uint32_t n_chunks, PartTable parts,
                                                            uint32_t tile_size, uint32_t* tile_first,
                                                            DeviceParams* prm, uint32_t cap_bits,
                                                            uint32_t comp_shift, BinTable bt, uint32_t per,
                                                            uint32_t sub_shift) {
  if (blockIdx.x == 0) {
    for (uint32_t k = threadIdx.x; k <= n_classes; k += blockDim.x) bt.base[k] = 0;  // row 0
    for (uint32_t k = threadIdx.x; k < bt.n_bins; k += blockDim.x) bt.fill[k] = 0;
    servant_scan_block(sv, n_classes, max_slots, slot_base, cls_begin, chunk_consuming, n_chunks, parts,
                       tile_size, tile_first, prm);
    return;
  }
  bin_count_block(sv, n_classes, parts, cap_bits, comp_shift, bt, blockIdx.x - 1, per, sub_shift);
}

// ---------------------------------------------------------------------------
// k_slot_bin: slot generation into bins. Workgroups [0, gen_blocks): a tile of 256 x `items`
// consecutive slots of the generation order each (servant-major, running ascending) — wave w
// owns items * 64 consecutive slots and walks them 64 at a time, like the radix scatter;
// workgroups behind them classify requests (task_classify_block). Values are
// (class << gbits) | slot (gbits == 0: one class, the slot alone).
//
// A tile's records of one bin form ONE contiguous run of the bin's staging region, in slot
// order (ranked with wave ballots, no LDS atomics); where the run starts is whatever the bin's
// arrival counter hands out. k_bin_sort puts the runs of a bin into tile order — which is slot
// order — without comparing anything.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_slot_bin(ServantTable sv, const uint32_t* slot_base,
                                                  DeviceParams* prm, uint32_t cap_bits,
                                                  uint32_t* owner, uint32_t gen_blocks, uint32_t items,
                                                  uint32_t gbits, ClassifyArgs ca, uint32_t comp_shift,
                                                  const uint32_t* tile_first, BinTable bt,
                                                  uint32_t row, uint2* stage) {
  extern __shared__ uint32_t hb[];  // start[B] | cnt[4 waves][B]
  if (blockIdx.x >= gen_blocks) {
    task_classify_block(ca, blockIdx.x - gen_blocks, prm);
    return;
  }
  constexpr uint32_t kWindow = 2048;
  __shared__ uint32_t win[kWindow];
  __shared__ uint32_t run_ends[2];
  const uint32_t B = bt.n_bins;
  const uint32_t M = prm->n_slots;
  const uint32_t tile = blockIdx.x, base = tile * (blockDim.x * items);
  if (base >= M) return;  // (uniform: the whole workgroup)
  for (uint32_t d = threadIdx.x; d < 4 * B; d += blockDim.x) hb[B + d] = 0;
  const uint32_t g_end = min(M, base + blockDim.x * items);
  // Owners: a tile's owners are one short run of servants (k_slot_gen has the details).
  if (threadIdx.x == 0) run_ends[0] = tile_first[tile];
  if (threadIdx.x == 64) run_ends[1] = tile_first[tile + 1];  // >= the owner of slot g_end - 1
  __syncthreads();
  const uint32_t s_first = run_ends[0];
  const uint32_t n_run = run_ends[1] - s_first + 1;
  const bool windowed = n_run <= kWindow;
  if (windowed)
    for (uint32_t i = threadIdx.x; i < n_run; i += blockDim.x) win[i] = slot_base[s_first + i];
  __syncthreads();
  uint32_t bbits = 0;
  while ((1u << bbits) < B) ++bbits;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t wave_span = items * 64;
  const uint64_t lt_mask = (1ull << lane) - 1;
  uint32_t* wcnt = hb + B + wave * B;
  uint32_t key[kSortItems], val[kSortItems], rank[kSortItems];
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t g = base + wave * wave_span + j * 64 + lane;
    const bool valid = (uint32_t)j < items && g < g_end;
    key[j] = val[j] = rank[j] = 0;
    uint32_t d = 0;
    if (valid) {
      uint32_t s;
      if (windowed) {
        uint32_t lo = 0, hi = n_run;  // win[lo] <= g < win[hi] (hi == n_run: beyond the run)
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (win[mid] <= g) lo = mid; else hi = mid;
        }
        s = s_first + lo;
      } else {
        s = owner_of_slot(slot_base, sv.n, g);
      }
      const uint32_t r = sv.running[s] + (g - slot_base[s]);
      owner[g] = s;
      const uint32_t nproc = sv.nproc[s], flags = sv.flags[s];
      const uint32_t cap = slot_capacity(nproc, sv.load[s], sv.max_tasks[s], r);
      uint64_t k64 = slot_key_exact(slot_tier(nproc, flags, r), r, cap, cap_bits);
      const uint32_t cls = sv.class_of[s];
      if (ca.n_parts > 1) k64 |= (uint64_t)ca.cls_comp[cls] << comp_shift;
      key[j] = (uint32_t)k64;
      val[j] = gbits ? (cls << gbits) | g : g;
      d = key[j] >> bt.shift;
    }
    if ((uint32_t)j < items) {  // (wave-uniform)
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
  // Earlier waves' records of the bin; one arrival-counter update per tile and bin.
  for (uint32_t d = threadIdx.x; d < B; d += blockDim.x) {
    uint32_t off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t t = hb[B + w * B + d];
      hb[B + w * B + d] = off;
      off += t;
    }
    hb[d] = off ? bt.base[(size_t)d * row + (row - 1)] + atomicAdd(&bt.fill[d], off) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t g = base + wave * wave_span + j * 64 + lane;
    if ((uint32_t)j < items && g < g_end) {
      const uint32_t d = key[j] >> bt.shift;
      stage[hb[d] + wcnt[d] + rank[j]] = make_uint2(key[j], val[j]);
    }
  }
}

// ---------------------------------------------------------------------------
// k_bin_sort: workgroup j < B orders bin j and writes the slots' places in the global order
// (rank_to_g[rank] = slot) and in the class lists (list[position] = {rank, slot}); workgroup B
// (if launched) is the chunk prefix of the consuming counts.
//
// No comparison sort. (1) The bin's staging region is a set of runs, one per tile, each in slot
// order (k_slot_bin): a count of the records per tile, a scan over the tiles and the record's
// offset inside its run (index - first index of the run) put the records into slot order.
// (2) One stable counting pass per 8 bits of the key below the bin's own bits (ballot-ranked
// like the radix scatter, all in LDS) makes that (key, slot) order — the global order inside
// the bin. (3) A stable partition by class gives the places in the class lists.
// A record lives in LDS as one word: key bits below the bin | slot | class.
// ---------------------------------------------------------------------------
constexpr uint32_t kBinThreads = 1024;
constexpr uint32_t kBinWaves = kBinThreads / 64;
constexpr uint32_t kBinRounds = kBinCap / kBinThreads;  // records per thread
constexpr uint32_t kBinMaxTiles = 2048;                 // (both tile tables fit the counter table)
// LDS words: two record buffers, the counter table ([waves][256]; the tile tables before that),
// digit totals / starts, class-list cursors.
constexpr uint32_t kBinLdsWords = 2 * kBinCap + kBinWaves * 256 + 256 + 256;

struct BinSortArgs {
  const uint2* stage;
  BinTable bt;
  uint32_t n_classes, gbits;
  uint32_t slot_bits, cls_bits;  // of the LDS word (cls_bits == 0: one class)
  uint32_t tile_shift, n_tiles;  // tile of a slot = slot >> tile_shift (k_slot_bin's tiles)
  const uint32_t* cls_begin;     // [C + 1] (k_servant_scan)
  uint2* list;                   // class lists: {global rank, slot}
  uint32_t* rank_to_g;
};

__global__ __launch_bounds__(kBinThreads, 8) void k_bin_sort(BinSortArgs a, DeviceParams* prm, PrefixArgs pa) {
  extern __shared__ __attribute__((aligned(16))) uint32_t bsm[];
  __shared__ uint32_t lds[17];
  if (blockIdx.x == a.bt.n_bins) {
    chunk_prefix_block(pa, prm);
    return;
  }
  uint32_t* buf0 = bsm;
  uint32_t* buf1 = bsm + kBinCap;
  uint32_t* tab = bsm + 2 * kBinCap;           // [kBinWaves][256]
  uint32_t* dtot = tab + kBinWaves * 256;      // [256]
  uint32_t* cbase = dtot + 256;                // [256] next free position of every class list
  uint32_t* tcnt = tab;                        // [n_tiles] (step 1 only)
  uint32_t* tfirst = tab + kBinMaxTiles;       // [n_tiles]
  const uint32_t C = a.n_classes, row = C + 1, j = blockIdx.x;
  const uint32_t lo = a.bt.base[(size_t)j * row + C], hi = a.bt.base[(size_t)(j + 1) * row + C];
  const uint32_t n = hi - lo;
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
  for (uint32_t t = threadIdx.x; t < a.n_tiles; t += blockDim.x) {
    tcnt[t] = 0;
    tfirst[t] = 0xFFFFFFFFu;
  }
  for (uint32_t c = threadIdx.x; c < C; c += blockDim.x) cbase[c] = a.cls_begin[c] + a.bt.base[(size_t)j * row + c];
  __syncthreads();
  // ---- (1) runs -> slot order
  uint32_t word[kBinRounds], tile[kBinRounds];
#pragma unroll
  for (int k = 0; k < (int)kBinRounds; ++k) {
    const uint32_t i = k * kBinThreads + threadIdx.x;
    word[k] = tile[k] = 0;
    if (i < n) {
      const uint2 r = a.stage[lo + i];
      const uint32_t slot = r.y & gmask, cls = a.gbits ? r.y >> a.gbits : 0u;
      word[k] = ((r.x & kmask) << sbits) | (slot << a.cls_bits) | cls;
      tile[k] = slot >> a.tile_shift;
      atomicAdd(&tcnt[tile[k]], 1u);
      atomicMin(&tfirst[tile[k]], i);
    }
  }
  __syncthreads();
  {
    // Exclusive scan of the tile counts, two tiles per thread (n_tiles <= 2 * kBinThreads).
    const uint32_t t0 = 2 * threadIdx.x;
    const uint32_t v0 = t0 < a.n_tiles ? tcnt[t0] : 0u, v1 = t0 + 1 < a.n_tiles ? tcnt[t0 + 1] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(v0 + v1, lds, &total);
    if (t0 < a.n_tiles) tcnt[t0] = ex;
    if (t0 + 1 < a.n_tiles) tcnt[t0 + 1] = ex + v0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < (int)kBinRounds; ++k) {
    const uint32_t i = k * kBinThreads + threadIdx.x;
    if (i < n) buf0[tcnt[tile[k]] + (i - tfirst[tile[k]])] = word[k];
  }
  __syncthreads();
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
  // ---- (3) places. Positions are walked 1024 at a time; within a round (wave, lane) order ==
  // position order, so "earlier slots of the same class" = earlier rounds (cbase) + earlier
  // waves of this round (tab) + lower lanes of this wave (ballot match on the class bits).
  for (uint32_t t = threadIdx.x; t < kBinWaves * 256; t += blockDim.x) tab[t] = 0;
  __syncthreads();
  const uint32_t cmask = a.cls_bits ? (1u << a.cls_bits) - 1 : 0u;
  for (uint32_t p0 = 0; p0 < n; p0 += blockDim.x) {
    const uint32_t p = p0 + threadIdx.x;
    const bool valid = p < n;
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
}

}  // namespace ydc
#endif  // YADCC_AMD_BIN_SORT_H_

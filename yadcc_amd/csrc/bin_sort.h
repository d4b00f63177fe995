// bin_sort.h — the slot order in three launches (small and medium registries).
//
// The radix sort of kernels.h needs seven dependent launches for a 17-bit key (slot
// generation, then histogram / column scan / scatter per digit) and every one of them is
// launch-latency-bound below ~1M slots. This path gets the same order — (key, registry
// index), the reference's arg-min with first-wins ties, task_dispatcher.cc:440-447 — and the
// same per-class lists out of three:
//
//   k_servant_scan_bins  workgroup 0: the servant scan of k_servant_scan. Workgroups 1..B: the
//                        key space is cut into B equal bins; "how many slots of class c have a
//                        key below bin j" is a sum of per-servant closed forms
//                        (first_slot_not_below, dispatch_core.h) — so the START of every bin in
//                        the global order and in every class list is known before a single
//                        slot has been generated. No histogram, no scan.
//   k_slot_bin           thread per slot (as k_slot_gen): key, class, owner; the record goes
//                        into its bin's region of the staging array, at whatever position the
//                        bin's arrival counter hands out (one global atomic per tile and bin;
//                        the order inside a bin is settled next). Extra workgroups classify the
//                        requests, as in k_slot_gen.
//   k_bin_sort           one workgroup per bin: the bin's records (a few hundred) are sorted
//                        in LDS by (key, slot) — a bitonic network — which gives every slot
//                        its global rank (bin start + position); a stable partition by class
//                        (wave ballots, the ranking of the radix scatter) gives its place in
//                        its class list (bin's start in that list + earlier slots of the class).
//                        One more workgroup computes the chunk prefix of the consuming counts.
//
// A bin that does not fit the LDS buffer (kBinCap records — tens of thousands of slots with
// one and the same utilisation) makes the batch report DeviceParams::window_miss; the host
// repeats it with the radix sort and stays with that until the registry changes structure.
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

constexpr uint32_t kBinsPerGroup = 8;  // most bin boundaries one workgroup of the first launch evaluates

// Workgroup w >= 1 of k_servant_scan_bins: boundaries j = (w - 1) * per + 1 .. (w - 1) * per + per
// (per <= kBinsPerGroup), row j of BinTable::base each. A thread keeps its servants' columns in
// registers across the boundaries; with a handful of classes the per-class sums are reduced in
// registers and across the wave before they touch LDS (one LDS atomic per wave, class and
// boundary instead of one per servant and boundary on two or three addresses).
__device__ __forceinline__ void bin_count_block(const ServantTable& sv, uint32_t n_classes,
                                                const PartTable& parts, uint32_t cap_bits,
                                                uint32_t comp_shift, const BinTable& bt, uint32_t group,
                                                uint32_t per) {
  extern __shared__ uint32_t cls_cnt[];  // per * (n_classes + 1): entry n_classes = all classes
  const uint32_t row = n_classes + 1;
  for (uint32_t c = threadIdx.x; c < per * row; c += blockDim.x) cls_cnt[c] = 0;
  __syncthreads();
  const uint32_t j0 = group * per + 1;
  const bool few = n_classes <= 4;
  uint32_t acc[kBinsPerGroup][5];  // [boundary][class | all] (few classes: registers)
#pragma unroll
  for (int b = 0; b < (int)kBinsPerGroup; ++b)
#pragma unroll
    for (int c = 0; c < 5; ++c) acc[b][c] = 0;
  for (uint32_t s = threadIdx.x; s < sv.n; s += blockDim.x) {
    const uint32_t cls = sv.class_of[s];
    if (cls == kNone) continue;
    const uint32_t run = sv.running[s], nproc = sv.nproc[s], load = sv.load[s], mt = sv.max_tasks[s],
                   fl = sv.flags[s];
    const uint64_t part_key = parts.n_parts > 1 ? (uint64_t)parts.cls_comp[cls] << comp_shift : 0ull;
#pragma unroll
    for (int b = 0; b < (int)kBinsPerGroup; ++b) {
      if ((uint32_t)b < per) {
        const uint64_t K = (uint64_t)(j0 + b) << bt.shift;
        const uint32_t cnt = first_slot_not_below_direct(nproc, load, mt, run, fl, part_key, K, cap_bits) - run;
        acc[b][4] += cnt;
        if (few) {
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[b][c] += cls == (uint32_t)c ? cnt : 0u;
        } else if (cnt) {
          atomicAdd(&cls_cnt[b * row + cls], cnt);
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < (int)kBinsPerGroup; ++b) {
    if ((uint32_t)b < per) {
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        if (c == 4 || (few && (uint32_t)c < n_classes)) {
          uint32_t v = acc[b][c];
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
          if ((threadIdx.x & 63) == 0 && v) atomicAdd(&cls_cnt[b * row + (c == 4 ? n_classes : (uint32_t)c)], v);
        }
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < per * row; i += blockDim.x) bt.base[(size_t)j0 * row + i] = cls_cnt[i];
}

__global__ __launch_bounds__(1024) void k_servant_scan_bins(ServantTable sv, uint32_t n_classes,
                                                            uint32_t max_slots, uint32_t* slot_base,
                                                            uint32_t* cls_begin, uint32_t* chunk_consuming,
                                                            uint32_t n_chunks, PartTable parts,
                                                            uint32_t tile_size, uint32_t* tile_first,
                                                            DeviceParams* prm, uint32_t cap_bits,
                                                            uint32_t comp_shift, BinTable bt, uint32_t per) {
  if (blockIdx.x == 0) {
    for (uint32_t k = threadIdx.x; k <= n_classes; k += blockDim.x) bt.base[k] = 0;  // row 0
    for (uint32_t k = threadIdx.x; k < bt.n_bins; k += blockDim.x) bt.fill[k] = 0;
    servant_scan_block(sv, n_classes, max_slots, slot_base, cls_begin, chunk_consuming, n_chunks, parts,
                       tile_size, tile_first, prm);
    return;
  }
  bin_count_block(sv, n_classes, parts, cap_bits, comp_shift, bt, blockIdx.x - 1, per);
}

// ---------------------------------------------------------------------------
// k_slot_bin: slot generation into bins. Workgroups [0, gen_blocks): 256 x `items` consecutive
// slots of the generation order each (servant-major, running ascending); workgroups behind
// them classify requests (task_classify_block). Values are (class << gbits) | slot (gbits == 0:
// one class, the slot alone).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_slot_bin(ServantTable sv, const uint32_t* slot_base,
                                                  DeviceParams* prm, uint32_t cap_bits,
                                                  uint32_t* owner, uint32_t gen_blocks, uint32_t items,
                                                  uint32_t gbits, ClassifyArgs ca, uint32_t comp_shift,
                                                  const uint32_t* tile_first, BinTable bt,
                                                  uint32_t row, uint2* stage) {
  extern __shared__ uint32_t hb[];  // count[B] | start[B]
  if (blockIdx.x >= gen_blocks) {
    task_classify_block(ca, blockIdx.x - gen_blocks, prm);
    return;
  }
  constexpr uint32_t kWindow = 2048;
  __shared__ uint32_t win[kWindow];
  __shared__ uint32_t run_ends[2];
  const uint32_t B = bt.n_bins;
  for (uint32_t d = threadIdx.x; d < B; d += blockDim.x) hb[d] = 0;
  const uint32_t M = prm->n_slots;
  const uint32_t tile = blockIdx.x, base = tile * (blockDim.x * items);
  if (base >= M) return;  // (uniform: the whole workgroup)
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
  uint32_t key[kSortItems], val[kSortItems], loc[kSortItems];
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t g = base + j * blockDim.x + threadIdx.x;
    key[j] = val[j] = loc[j] = 0;
    if ((uint32_t)j < items && g < g_end) {
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
      loc[j] = atomicAdd(&hb[key[j] >> bt.shift], 1u);
    }
  }
  __syncthreads();
  // One arrival-counter update per tile and bin: where this tile's records of the bin go.
  for (uint32_t d = threadIdx.x; d < B; d += blockDim.x) {
    const uint32_t cnt = hb[d];
    hb[B + d] = cnt ? bt.base[(size_t)d * row + (row - 1)] + atomicAdd(&bt.fill[d], cnt) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t g = base + j * blockDim.x + threadIdx.x;
    if ((uint32_t)j < items && g < g_end)
      stage[hb[B + (key[j] >> bt.shift)] + loc[j]] = make_uint2(key[j], val[j]);
  }
}

// ---------------------------------------------------------------------------
// k_bin_sort: workgroup j < B sorts bin j and writes the slots' places in the global order
// (rank_to_g[rank] = slot) and in the class lists (list[position] = {rank, slot}); workgroup B
// (if launched) is the chunk prefix of the consuming counts.
// ---------------------------------------------------------------------------
struct BinSortArgs {
  const uint2* stage;
  BinTable bt;
  uint32_t n_classes, gbits;
  const uint32_t* cls_begin;  // [C + 1] (k_servant_scan)
  uint2* list;                // class lists: {global rank, slot}
  uint32_t* rank_to_g;
};

__global__ __launch_bounds__(256) void k_bin_sort(BinSortArgs a, DeviceParams* prm, PrefixArgs pa) {
  extern __shared__ __attribute__((aligned(16))) uint64_t srt[];  // kBinCap composite keys
  __shared__ uint32_t cbase[kMaxWaveClasses];      // next free position of every class list
  __shared__ uint32_t wcnt[4][kMaxWaveClasses];    // this round's slots per wave and class
  if (blockIdx.x == a.bt.n_bins) {
    chunk_prefix_block(pa, prm);
    return;
  }
  const uint32_t C = a.n_classes, row = C + 1, j = blockIdx.x;
  const uint32_t lo = a.bt.base[(size_t)j * row + C], hi = a.bt.base[(size_t)(j + 1) * row + C];
  const uint32_t n = hi - lo;
  if (n == 0 || prm->n_slots == 0) return;
  if (n > kBinCap) {
    if (threadIdx.x == 0) prm->window_miss = 1;  // the host repeats the batch with the radix sort
    return;
  }
  uint32_t npad = 64;
  while (npad < n) npad <<= 1;
  // Composite sort key: key | slot | class, the class in the low bits so that it never decides
  // (slots are unique): ascending order == (key, registry order of the slot).
  const uint32_t cbits = a.gbits ? 32 - a.gbits : 0;
  const uint32_t gmask = a.gbits ? (1u << a.gbits) - 1 : 0xFFFFFFFFu;
  for (uint32_t i = threadIdx.x; i < npad; i += blockDim.x) {
    uint64_t v = ~0ull;
    if (i < n) {
      const uint2 r = a.stage[lo + i];
      const uint32_t low = a.gbits ? ((r.y & gmask) << cbits) | (r.y >> a.gbits) : r.y;
      v = ((uint64_t)r.x << 32) | low;
    }
    srt[i] = v;
  }
  for (uint32_t c = threadIdx.x; c < C; c += blockDim.x) cbase[c] = a.cls_begin[c] + a.bt.base[(size_t)j * row + c];
  for (uint32_t c = threadIdx.x; c < 4 * kMaxWaveClasses; c += blockDim.x) (&wcnt[0][0])[c] = 0;
  __syncthreads();
  // Bitonic network over npad entries, npad / 2 compare-exchanges per step.
  for (uint32_t k = 2; k <= npad; k <<= 1) {
    for (uint32_t jj = k >> 1; jj > 0; jj >>= 1) {
      for (uint32_t i = threadIdx.x; i < (npad >> 1); i += blockDim.x) {
        const uint32_t ia = ((i & ~(jj - 1)) << 1) | (i & (jj - 1));
        const uint32_t ib = ia | jj;
        const uint64_t x = srt[ia], y = srt[ib];
        const bool up = (ia & k) == 0;
        if ((x > y) == up) {
          srt[ia] = y;
          srt[ib] = x;
        }
      }
      __syncthreads();
    }
  }
  // Places. Positions are walked 256 at a time; within a round (wave, lane) order == position
  // order, so "earlier slots of the same class" = earlier rounds (cbase) + earlier waves of this
  // round (wcnt) + lower lanes of this wave (ballot match on the class bits).
  uint32_t nb = 0;
  while ((1u << nb) < C) ++nb;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1;
  for (uint32_t p0 = 0; p0 < n; p0 += blockDim.x) {
    const uint32_t p = p0 + threadIdx.x;
    const bool valid = p < n;
    uint32_t slot = 0, cls = 0;
    if (valid) {
      const uint32_t low = (uint32_t)srt[p];
      slot = a.gbits ? low >> cbits : low;
      cls = a.gbits ? low & ((1u << cbits) - 1) : 0u;
    }
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if ((uint32_t)b < nb) {
        const uint64_t m = __ballot((cls >> b) & 1u);
        peers &= ((cls >> b) & 1u) ? m : ~m;
      }
    }
    const uint32_t before_in_wave = (uint32_t)__popcll(peers & lt_mask);
    if (valid && before_in_wave == 0) wcnt[wave][cls] = (uint32_t)__popcll(peers);
    __syncthreads();
    if (valid) {
      uint32_t pos = cbase[cls] + before_in_wave;
      for (uint32_t w = 0; w < wave; ++w) pos += wcnt[w][cls];
      const uint32_t rank = lo + p;
      a.list[pos] = make_uint2(rank, slot);
      a.rank_to_g[rank] = slot;
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < C; c += blockDim.x) {
      cbase[c] += wcnt[0][c] + wcnt[1][c] + wcnt[2][c] + wcnt[3][c];
      wcnt[0][c] = wcnt[1][c] = wcnt[2][c] = wcnt[3][c] = 0;
    }
    __syncthreads();
  }
}

}  // namespace ydc
#endif  // YADCC_AMD_BIN_SORT_H_

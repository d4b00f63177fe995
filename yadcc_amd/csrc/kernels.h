// kernels.h — HIP kernels of the MI355X task-dispatch path (gfx950, wave64).
//
// Pipeline of one batch (all on one stream, sizes that depend on device data
// are read from DeviceParams, so the host never waits for them):
//
//   k_servant_scan   per-servant free-slot counts -> slot_base[], class sizes, counter reset
//   k_slot_gen       one (key, generation index) pair per free slot + the first sort pass's
//                    tile histograms; extra workgroups classify the requests (eligible-class
//                    mask, own-servant slot range, consuming count per chunk)
//   k_radix_hist / _scan / _scatter   stable LSD radix sort of the slots by key (the chunk
//                    prefix of the consuming counts rides in a histogram launch)
//   k_radix_scatter_classed / class passes   per-class sorted lists (rank, slot)
//   (bin_sort.h: registries up to 600k slots get the same order and lists from two launches —
//    k_front_bins: bin starts in closed form | slot tiles | request classification, no scan;
//    k_bin_sort: counting passes per bin in LDS — instead of everything above this line)
//   k_match_pass (match_kernel.h)     chunk-parallel speculative replay of the greedy picks;
//                    pass 0 makes its own level guesses (k_guess_init: > 64 classes, sharded)
//   k_finalize                        rank -> slot -> servant index, utilisation; running_tasks in
//                    closed form from the final class states (same launch)
//
// The arithmetic (capacity, keys, class-state machine) lives in dispatch_core.h
// and is shared with the CPU model in tests/model.
#ifndef YADCC_AMD_KERNELS_H_
#define YADCC_AMD_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dispatch_core.h"

namespace ydc {

constexpr int kSortThreads = 256;
constexpr int kSortItems = 8;  // most; a pass may use fewer (SortIn::items) for more, smaller tiles
constexpr int kSortWaves = kSortThreads / 64;

// Sort tiles and the 8 XCDs. Workgroups go to the XCDs round-robin and every XCD has an L2 of
// its own. What tile t writes continues what tile t - 1 wrote: its run of every digit in the
// scatter's output, its column of the digit-major histogram table — word for word, so the two
// share 32-byte sectors and 128-byte lines. Handed to different XCDs the halves of a sector
// leave from two caches as masked writes (WRITE_SIZE 2.2 - 2.4x the bytes written, rounds 1-3);
// workgroup b therefore takes tile (b mod 8) * ceil(n / 8) + b / 8: every XCD works through a
// contiguous range of tiles in order, and neighbours in memory meet in one L2.
constexpr uint32_t kXcds = 8;
__host__ __device__ __forceinline__ uint32_t xcd_grid(uint32_t n_tiles) {
  return (n_tiles + kXcds - 1) / kXcds * kXcds;
}
__device__ __forceinline__ uint32_t xcd_tile(uint32_t block, uint32_t n_tiles, uint32_t on = 1) {
  if (!on) return block;
  return (block % kXcds) * ((n_tiles + kXcds - 1) / kXcds) + block / kXcds;  // >= n_tiles: nothing to do
}


// Device-resident scalars produced and consumed by the kernels.
struct DeviceParams {
  uint32_t n_slots;       // M: free slots of this batch
  uint32_t overflow;      // M exceeded the workspace
  uint32_t reserved0;     // (was: need_shared)
  uint32_t n_changed[64]; // pass r (index r & 63) changed some chunk's end state: not final yet
  // Sampled count of the end states pass r changed (chunks with index % 16 == 0 only: an
  // estimate at a sixteenth of the same-address atomics), same indexing as the flags.
  uint32_t n_sampled[64];
  uint32_t chunk_sims;    // chunk simulations executed (all rounds)
  uint32_t granted;       // requests that got a slot (k_running_out)
  uint32_t consuming;     // requests with at least one eligible class (k_chunk_prefix)
  uint32_t batch_seq;     // batches this context has started (never reset)
  // Multi-GPU with a sharded sort (k_window): this rank only generated the slots of a key
  // window. rank_offset = slots of the whole registry that sort below the window (global rank
  // of local position 0; 0 otherwise); window_miss = some rank's window did not cover what its
  // requests reached (every rank sets it alike; the batch is repeated with the full sort).
  uint32_t rank_offset;
  uint32_t window_miss;
  // Multi-GPU over the inter-process mailbox transport (k_mailbox_all_gather): a peer's data
  // did not arrive in time. Never reset by a batch: the group is broken.
  uint32_t exchange_timeout;
  // Pipelined batches (ydc_dispatch_device_async): an earlier batch of the pipeline did not
  // become final within its pre-launched passes — the batches enqueued behind it take no effect
  // (their finalise is gated on this) and the host replays from there. Cleared by the host.
  uint32_t pipeline_broken;
  // k_finalize's servant workgroups that have reported (the last one hands the outcome to the
  // host and clears it).
  uint32_t fin_reports;
  // zone_guess.h: chunks the walk of the dedicated tier's end left start cursors for (0: none).
  uint32_t zone_rows;
  // The 100 MHz wall clock when the batch's first kernel reset the counters and when its finalise
  // handed over the outcome: what the batch cost on the device, queue gaps included (the host
  // weighs optional steps against it: ydc_api.hip zone_decide).
  uint32_t t_begin_lo, t_begin_hi, t_end_lo, t_end_hi;
};

// Measurement builds only (`make probe`): wall-clock stamps the kernels leave behind
// (match_kernel.h: k_match_pass phases per chunk; bin_sort.h: the front's workgroups and
// k_bin_sort's phases; wide_kernel.h: the walk's accumulators). Compiled out of the product.
#ifdef YDC_PHASE_PROBE
constexpr uint32_t kProbeSlots = 12, kProbeChunks = 8192;
__device__ unsigned long long ydc_phase_probe[kProbeChunks * kProbeSlots];
#endif

struct ServantTable {
  const uint32_t* version;
  const uint32_t* nproc;
  const uint32_t* load;
  const uint32_t* max_tasks;
  const uint32_t* running;
  const uint32_t* flags;
  const uint32_t* class_of;
  uint32_t n;
};

// ---------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}

typedef __attribute__((address_space(3))) uint32_t lds_u32_t;  // (an LDS word named by its byte address)

// Minimum over the 64 lanes (every lane gets it). Six v_min_u32_dpp.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = min(v, dpp_u32<0x111>(0xFFFFFFFFu, v));       // row_shr:1
  v = min(v, dpp_u32<0x112>(0xFFFFFFFFu, v));       // row_shr:2
  v = min(v, dpp_u32<0x114>(0xFFFFFFFFu, v));       // row_shr:4
  v = min(v, dpp_u32<0x118>(0xFFFFFFFFu, v));       // row_shr:8
  v = min(v, dpp_u32<0x142, 0xa>(0xFFFFFFFFu, v));  // row_bcast:15
  v = min(v, dpp_u32<0x143, 0xc>(0xFFFFFFFFu, v));  // row_bcast:31
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Inclusive scan over the wave (ds_bpermute based; used off the hot loops).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d);
    if (lane >= (uint32_t)d) v += o;
  }
  return v;
}

// Exclusive scan of one value per thread over a workgroup of `blockDim.x`
// threads (multiple of 64, <= 1024). Returns the exclusive prefix; *total gets
// the workgroup sum. `lds` needs 17 words.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds,
                                                         uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t nw = blockDim.x >> 6;
  uint32_t inc = wave_inclusive_scan(v);
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    uint32_t w = lane < nw ? lds[lane] : 0;
    uint32_t winc = wave_inclusive_scan(w);
    if (lane < nw) lds[lane] = winc - w;
    if (lane == nw - 1) lds[16] = winc;
  }
  __syncthreads();
  uint32_t r = lds[wave] + inc - v;
  *total = lds[16];
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------------------
// k_servant_scan: ONE workgroup. slot_base[s] = number of free slots of the
// servants before s; cls_begin[c] = number of slots of the classes before c.
// ---------------------------------------------------------------------------
// Registry parts (host_tables.h: classes that no request can choose between are independent):
// cls_comp[c] = part of class c, n_parts of them (1: cls_comp is not read).
struct PartTable {
  const uint32_t* cls_comp;
  uint32_t n_parts;
  uint32_t* rank_base;  // [n_parts + 1] slots of the parts before g (k_servant_scan writes it)
};

// What follows the scan proper (shared by the two forms of the scan): per-batch counter
// reset, the run end behind the last tile, class list starts, first rank of every part.
// carry / last_with_slots / cls_cnt: the scan's LDS results; my_last: this thread's last
// servant with slots.
__device__ __forceinline__ void servant_scan_finish(const ServantTable& sv, uint32_t n_classes,
                                                    uint32_t max_slots, uint32_t* slot_base,
                                                    uint32_t* cls_begin, const PartTable& parts,
                                                    uint32_t tile_shift, uint32_t tile_mask,
                                                    uint32_t* tile_first, DeviceParams* prm,
                                                    uint32_t my_last, const uint32_t& carry,
                                                    uint32_t& last_with_slots, const uint32_t* cls_cnt) {
  if (threadIdx.x < 64) prm->n_changed[threadIdx.x] = prm->n_sampled[threadIdx.x] = 0;
  if (tile_first) {  // last servant that has slots: one LDS atomic per wave
    uint32_t v = my_last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(&last_with_slots, v);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    uint32_t m = carry;
    slot_base[sv.n] = m;
    // (behind the last tile: the last servant that has slots, as the end of that tile's run)
    if (tile_first && m <= max_slots) tile_first[(m + tile_mask) >> tile_shift] = last_with_slots;
    prm->overflow = m > max_slots ? 1u : 0u;
    prm->n_slots = m > max_slots ? 0u : m;
    prm->reserved0 = 0;
    prm->zone_rows = 0;
    {
      const unsigned long long t = wall_clock64();
      prm->t_begin_lo = (uint32_t)t;
      prm->t_begin_hi = (uint32_t)(t >> 32);
    }
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
      // Slots per part -> first global rank of every part (slots are ordered part-major).
      uint32_t cnt[16];  // kMaxComponents
      for (uint32_t g = 0; g < parts.n_parts; ++g) cnt[g] = 0;
      for (uint32_t c = 0; c < n_classes; ++c) cnt[parts.cls_comp[c]] += cls_cnt[c];
      uint32_t a2 = 0;
      for (uint32_t g = 0; g < parts.n_parts; ++g) {
        parts.rank_base[g] = a2;
        a2 += cnt[g];
      }
      parts.rank_base[parts.n_parts] = a2;
    }
  }
}

__device__ __forceinline__ void servant_scan_block(const ServantTable& sv, uint32_t n_classes,
                                                   uint32_t max_slots, uint32_t* slot_base,
                                                   uint32_t* cls_begin, uint32_t* chunk_consuming,
                                                   uint32_t n_chunks, const PartTable& parts,
                                                   uint32_t tile_size, uint32_t* tile_first,
                                                   DeviceParams* prm) {
  // Per-batch reset of the request-side counters (saves a memset launch).
  for (uint32_t k = threadIdx.x; k < n_chunks * parts.n_parts; k += blockDim.x) chunk_consuming[k] = 0;
  __shared__ uint32_t lds[17];
  __shared__ uint32_t carry, last_with_slots;
  extern __shared__ uint32_t cls_cnt[];  // n_classes + 1
  for (uint32_t c = threadIdx.x; c <= n_classes; c += blockDim.x) cls_cnt[c] = 0;
  if (threadIdx.x == 0) {
    carry = 0;
    last_with_slots = 0;
  }
  __syncthreads();
  // A servant's six columns are fetched together, one slab of servants ahead of the scan
  // that uses them (unconditional loads at a clamped index: one round trip per slab, and it
  // overlaps the previous slab's scan).
  struct Row {
    uint32_t cls, nproc, load, max_tasks, running, flags;
  };
  auto fetch = [&](uint32_t s) {
    const uint32_t i = min(s, sv.n ? sv.n - 1 : 0u);
    Row r{kNone, 0, 0, 0, 0, 0};
    if (sv.n) r = Row{sv.class_of[i], sv.nproc[i], sv.load[i], sv.max_tasks[i], sv.running[i], sv.flags[i]};
    if (s >= sv.n) r.cls = kNone;
    return r;
  };
  const uint32_t tile_shift = 31 - (uint32_t)__clz((int)tile_size), tile_mask = tile_size - 1;
  uint32_t my_last = 0;
  Row next = fetch(threadIdx.x);
  for (uint32_t s0 = 0; s0 < sv.n; s0 += blockDim.x) {
    const uint32_t s = s0 + threadIdx.x;
    const Row row = next;
    next = fetch(s + blockDim.x);
    const uint32_t cls = row.cls;
    const uint32_t k = cls == kNone ? 0u
                                    : servant_slot_count(row.nproc, row.load, row.max_tasks,
                                                         row.running, row.flags);
    uint32_t total;
    uint32_t ex = block_exclusive_scan(k, lds, &total);
    uint32_t base = carry + ex;
    if (s < sv.n) {
      slot_base[s] = base;
      if (k) {
        atomicAdd(&cls_cnt[cls], k);
        my_last = s;  // (ascending per thread)
        // tile_first[t] = owner of the first slot of sort tile t (k_slot_gen would otherwise
        // find it with a dependent binary search over slot_base): the tiles that start inside
        // this servant's slots. Tiles are 2^tile_shift slots.
        if (tile_first)
          for (uint32_t t = (base + tile_mask) >> tile_shift; ((uint64_t)t << tile_shift) < (uint64_t)base + k; ++t)
            tile_first[t] = s;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
  servant_scan_finish(sv, n_classes, max_slots, slot_base, cls_begin, parts, tile_shift, tile_mask,
                      tile_first, prm, my_last, carry, last_with_slots, cls_cnt);
}


// The same scan as SEVERAL workgroups, one per slab of 1024 servants (cfg3: 8, cfg4: 16), none of
// which waits for another: workgroup b adds up the slot counts of the servants BEFORE its slab
// itself (closed forms: b rounds of independent column loads, no barrier between them — the
// trick of k_front_bins' slot tiles) and scans its own slab once. The last workgroup has then
// seen every servant: it also counts per class and does what follows the scan. The slab loop of
// one workgroup paid a block scan and two barriers per slab with one round trip in flight
// (cfg4: 24 us); a run of consecutive servants per thread (one scan, strided loads) took 56.
__device__ __forceinline__ void servant_scan_multi(const ServantTable& sv, uint32_t n_classes,
                                                   uint32_t max_slots, uint32_t* slot_base,
                                                   uint32_t* cls_begin, uint32_t* chunk_consuming,
                                                   uint32_t n_chunks, const PartTable& parts,
                                                   uint32_t tile_size, uint32_t* tile_first,
                                                   DeviceParams* prm) {
  const bool last = blockIdx.x + 1 == gridDim.x;
  __shared__ uint32_t lds[17];
  __shared__ uint32_t carry, last_with_slots;
  extern __shared__ uint32_t cls_cnt[];  // n_classes + 1
  for (uint32_t c = threadIdx.x; c <= n_classes; c += blockDim.x) cls_cnt[c] = 0;
  if (threadIdx.x == 0) {
    carry = 0;
    last_with_slots = 0;
  }
  // Per-batch reset of the request-side counters (saves a memset launch): a stripe per workgroup.
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_chunks * parts.n_parts;
       k += gridDim.x * blockDim.x)
    chunk_consuming[k] = 0;
  __syncthreads();
  const uint32_t slab0 = blockIdx.x * blockDim.x;
  uint32_t before = 0, my_last = 0;
  for (uint32_t s0 = threadIdx.x; s0 < slab0; s0 += 4 * blockDim.x) {
    uint32_t cls[4], nproc[4], load[4], mt[4], run[4], fl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // (four rows per thread in flight; s < slab0 <= sv.n)
      const uint32_t s = min(s0 + u * blockDim.x, slab0 - 1);
      cls[u] = sv.class_of[s];
      nproc[u] = sv.nproc[s];
      load[u] = sv.load[s];
      mt[u] = sv.max_tasks[s];
      run[u] = sv.running[s];
      fl[u] = sv.flags[s];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t s = s0 + u * blockDim.x;
      if (s >= slab0 || cls[u] == kNone) continue;
      const uint32_t k = servant_slot_count(nproc[u], load[u], mt[u], run[u], fl[u]);
      before += k;
      if (last && k) {
        atomicAdd(&cls_cnt[cls[u]], k);
        my_last = max(my_last, s);
      }
    }
  }
  // the slab's own servants
  const uint32_t s = slab0 + threadIdx.x;
  uint32_t cls = kNone, k = 0;
  if (s < sv.n) {
    cls = sv.class_of[s];
    if (cls != kNone) k = servant_slot_count(sv.nproc[s], sv.load[s], sv.max_tasks[s], sv.running[s], sv.flags[s]);
  }
  {
    uint32_t v = before;  // one LDS atomic per wave
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&carry, v);
  }
  uint32_t total;
  const uint32_t ex = block_exclusive_scan(k, lds, &total);  // (barriers inside: carry is complete behind it)
  const uint32_t tile_shift = 31 - (uint32_t)__clz((int)tile_size), tile_mask = tile_size - 1;
  const uint32_t base = carry + ex;
  if (s < sv.n) {
    slot_base[s] = base;
    if (k) {
      if (last) atomicAdd(&cls_cnt[cls], k);
      my_last = s;
      // tile_first[t] = owner of the first slot of sort tile t: the tiles that start inside this
      // servant's slots (k_slot_gen would otherwise find it with a dependent binary search).
      if (tile_first)
        for (uint32_t t = (base + tile_mask) >> tile_shift; ((uint64_t)t << tile_shift) < (uint64_t)base + k; ++t)
          tile_first[t] = s;
    }
  }
  if (!last) return;
  __syncthreads();
  if (threadIdx.x == 0) carry += total;
  __syncthreads();
  servant_scan_finish(sv, n_classes, max_slots, slot_base, cls_begin, parts, tile_shift, tile_mask,
                      tile_first, prm, my_last, carry, last_with_slots, cls_cnt);
}

__global__ __launch_bounds__(1024) void k_servant_scan(ServantTable sv, uint32_t n_classes,
                                                       uint32_t max_slots, uint32_t* slot_base,
                                                       uint32_t* cls_begin, uint32_t* chunk_consuming,
                                                       uint32_t n_chunks, PartTable parts,
                                                       uint32_t tile_size, uint32_t* tile_first,
                                                       DeviceParams* prm) {
  if (gridDim.x > 1)
    servant_scan_multi(sv, n_classes, max_slots, slot_base, cls_begin, chunk_consuming, n_chunks, parts,
                       tile_size, tile_first, prm);
  else
    servant_scan_block(sv, n_classes, max_slots, slot_base, cls_begin, chunk_consuming, n_chunks, parts,
                       tile_size, tile_first, prm);
}

// ---------------------------------------------------------------------------
// Request classification: thread per request. Rides in k_slot_gen's launch (both only need
// k_servant_scan's output, and a launch costs about what either of them does).
// ---------------------------------------------------------------------------
struct TaskColumns {
  const uint32_t* env_id;
  const uint32_t* min_version;
  const uint32_t* requestor_ip;
};

struct ClassifyArgs {
  TaskColumns tk;
  uint32_t n_tasks;
  const uint64_t* cls_env;
  const uint32_t* cls_ver;
  uint32_t n_classes, words;
  uint32_t env_words;  // words of a class's environment mask (cls_env: env_words per class)
  // Lookup form of the same test (host_tables.h; NULL: loop over the classes instead).
  const uint32_t* ver_sorted;
  const uint64_t* env_ver_mask;
  uint32_t n_versions;
  const uint32_t* ip_sorted;
  const uint32_t* ip_servant;
  uint32_t n_servants;
  const uint32_t* slot_base;
  uint32_t chunk_size;
  uint64_t* mask;
  uint32_t* self_lo;
  uint32_t* self_hi;
  uint32_t* chunk_consuming;  // [chunk * n_parts + part]
  const uint32_t* cls_comp;   // part of every class (n_parts > 1 only)
  uint32_t n_parts;
  // Nullable, one part: consuming requests among the last `tail_len` requests of every chunk
  // that has a successor (the warm-ups of matching pass 0, match_kernel.h).
  uint32_t* chunk_tail;
  uint32_t tail_len;
  // The bin sort's front (bin_sort.h) classifies in the launch that computes slot_base, so:
  // by_servant != 0: an own servant is left as its index (self_hi = kSelfServant); per_wave != 0:
  // chunk_consuming is indexed by WAVE of 64 requests ([wave * n_parts + part], plain stores,
  // every entry written — nothing to reset) and the chunk prefix adds the waves of a chunk up.
  uint32_t by_servant, per_wave;
  uint32_t n_ip;  // entries of the ip table (>= n_servants: a servant may answer to several host ids)
  // The ip table as a hash (host_tables.h: ip_hash): {host id, value} per slot.
  const uint2* ip_hash;
  uint32_t ip_hash_shift;
  const uint32_t* ip_filter;  // one bit per host id in front of the table (host_tables.h)
  uint32_t ip_filter_shift;
  // Requests per thread (1, or 4: task_classify_block_multi — workgroups of 4 x 256 requests).
  uint32_t per_thread;
  uint32_t xcd_gen;  // (k_slot_gen's slot tiles in XCD-contiguous order)
  // Nullable (lookup form only): row of the (digest, version threshold) lookup the request
  // falls into (kNone: a digest nobody has) — k_sim_wide's eligible-class lists (host_tables.h).
  uint32_t* row_out;
};

__device__ __forceinline__ void task_classify_block(const ClassifyArgs& a, uint32_t block,
                                                    DeviceParams* prm) {
  const uint32_t t = block * blockDim.x + threadIdx.x;
  if (t >= a.n_tasks) return;
  // A wave is as slow as its slowest lane, and what it waits for here are dependent memory
  // round trips (~1 us each on a busy chip), not bytes: the three columns are fetched at once,
  // the requestor's host is ONE probe of the hashed ip table for almost every lane (the
  // binary search over 16k hosts was 14 dependent loads — and nine lanes in ten skipping it
  // did not help their wave), and that probe and the mask row are in flight together.
  const uint32_t env = a.tk.env_id[t], minv = a.tk.min_version[t], rip = a.tk.requestor_ip[t];
  const uint32_t hmask = (1u << (32 - a.ip_hash_shift)) - 1;
  uint32_t hslot = (rip * 0x9E3779B1u) >> a.ip_hash_shift;
  const uint32_t fbit = (rip * 0x85EBCA6Bu) >> a.ip_filter_shift;
  uint2 probe = make_uint2(0u, kNone);
  if (a.ip_filter[fbit >> 5] >> (fbit & 31) & 1u) probe = a.ip_hash[hslot];
  uint64_t any = 0;
  uint32_t first_cls = 0;  // an eligible class (names the request's part of the registry)
  {
    if (a.env_ver_mask) {
      uint32_t vi = 0;  // class versions below min_version
      for (uint32_t j = 0; j < a.n_versions; ++j) vi += a.ver_sorted[j] < minv;
      const uint32_t n_env = 64 * a.env_words;  // digests >= n_env: nobody has them
      if (a.row_out) a.row_out[t] = env < n_env ? env * (a.n_versions + 1) + vi : kNone;
      const uint64_t* row = a.env_ver_mask + ((size_t)min(env, n_env - 1) * (a.n_versions + 1) + vi) * a.words;
      for (uint32_t w = 0; w < a.words; ++w) {
        const uint64_t m = env < n_env ? row[w] : 0;
        a.mask[(size_t)t * a.words + w] = m;
        if (!any && m) first_cls = w * 64 + (uint32_t)__builtin_ctzll(m);
        any |= m;
      }
    } else
    for (uint32_t w = 0; w < a.words; ++w) {
      uint64_t m = 0;
      if (env < 64 * a.env_words) {
        const uint32_t c0 = w * 64, c1 = min(c0 + 64, a.n_classes);
        for (uint32_t c = c0; c < c1; ++c) {
          if (((a.cls_env[(size_t)c * a.env_words + (env >> 6)] >> (env & 63)) & 1u) &&
              a.cls_ver[c] >= minv)
            m |= 1ull << (c - c0);
        }
      }
      a.mask[(size_t)t * a.words + w] = m;
      if (!any && m) first_cls = w * 64 + (uint32_t)__builtin_ctzll(m);
      any |= m;
    }
  }
  uint32_t lo = kNone, hi = kNone;
  while (probe.y != kNone && probe.x != rip) {  // (a collision: the next slot)
    hslot = (hslot + 1) & hmask;
    probe = a.ip_hash[hslot];
  }
  if (probe.y != kNone) {
    if (probe.y & 0x80000000u) {
      lo = probe.y & 0x7FFFFFFFu;  // several servants on the host: `self` is resolved at replay time
      hi = kSelfShared;
    } else if (a.by_servant) {
      lo = probe.y;
      hi = kSelfServant;
    } else {
      const uint32_t s = probe.y;
      const uint32_t b = a.slot_base[s], e = a.slot_base[s + 1];
      if (e > b) {
        lo = b;
        hi = e;
      }
    }
  }
  a.self_lo[t] = lo;
  a.self_hi[t] = hi;
  // chunk_size is a multiple of 64 and waves start at multiples of 64, so all
  // lanes of a wave fall into the same chunk: one atomic per wave (and part).
  uint64_t consuming = __ballot(any != 0);
  const uint32_t leader = (uint32_t)__builtin_ctzll(__ballot(true));
  if (a.per_wave) {
    // One entry per wave and part, written whatever the count.
    const uint32_t G = a.n_parts ? a.n_parts : 1u;
    const uint32_t part = G > 1 && any ? a.cls_comp[first_cls] : 0u;
    for (uint32_t g = 0; g < G; ++g) {
      const uint64_t m = __ballot(any != 0 && part == g);
      if ((threadIdx.x & 63) == leader) a.chunk_consuming[(size_t)(t >> 6) * G + g] = (uint32_t)__popcll(m);
    }
    const uint32_t t_wave = t - (threadIdx.x & 63);
    if (G == 1 && a.chunk_tail && (t_wave + 64) % a.chunk_size == 0 && t_wave + 64 <= a.n_tasks &&
        (threadIdx.x & 63) == leader)
      a.chunk_tail[t / a.chunk_size] = (uint32_t)__popcll(consuming >> (64 - a.tail_len));
  } else if (a.n_parts <= 1) {
    if (consuming && (threadIdx.x & 63) == leader)
      atomicAdd(&a.chunk_consuming[t / a.chunk_size], (uint32_t)__popcll(consuming));
    // The last wave of a chunk (all of its 64 requests exist: another chunk follows or the
    // batch ends right here) leaves the count of its last tail_len requests behind.
    const uint32_t t_wave = t - (threadIdx.x & 63);
    if (a.chunk_tail && (t_wave + 64) % a.chunk_size == 0 && t_wave + 64 <= a.n_tasks &&
        (threadIdx.x & 63) == leader)
      a.chunk_tail[t / a.chunk_size] = (uint32_t)__popcll(consuming >> (64 - a.tail_len));
  } else {
    const uint32_t part = any ? a.cls_comp[first_cls] : 0xFFFFFFFFu;
    while (consuming) {
      const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)part, (int)__builtin_ctzll(consuming));
      const uint64_t same = __ballot(part == g);
      if ((threadIdx.x & 63) == leader)
        atomicAdd(&a.chunk_consuming[(size_t)(t / a.chunk_size) * a.n_parts + g], (uint32_t)__popcll(same));
      consuming &= ~same;
    }
  }
}

// The same classification with R requests per thread (thread j of workgroup b: requests
// b * 256 R + k * 256 + j, k < R — every wave still holds 64 consecutive requests per k), for
// the large batches of the radix path: the lookup form with one mask word, slot ranges of own
// servants, counts per chunk. Every step issues its loads for all R requests before anything
// waits, so a wave has R round trips in flight where the one-request form has one; what the
// wave count of a 4M-request batch (62.5k waves, 8 rounds of a full chip) made a chain of
// ~8 us per round becomes two rounds (cfg4: 62 -> see profiles/r04).
template <int R>
__device__ __forceinline__ void task_classify_block_multi(const ClassifyArgs& a, uint32_t block,
                                                          DeviceParams* prm) {
  const uint32_t t0 = block * (blockDim.x * R) + threadIdx.x;
  const uint32_t last = a.n_tasks - 1;  // (n_tasks > 0: no workgroup otherwise)
  uint32_t env[R], minv[R], rip[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const uint32_t t = min(t0 + k * blockDim.x, last);
    env[k] = a.tk.env_id[t];
    minv[k] = a.tk.min_version[t];
    rip[k] = a.tk.requestor_ip[t];
  }
  const uint32_t hmask = (1u << (32 - a.ip_hash_shift)) - 1;
  const uint32_t n_env = 64 * a.env_words;  // digests >= n_env: nobody has them
  uint32_t hslot[R], fbit[R], fword[R];
  uint2 probe[R];
  uint64_t m[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    hslot[k] = (rip[k] * 0x9E3779B1u) >> a.ip_hash_shift;
    fbit[k] = (rip[k] * 0x85EBCA6Bu) >> a.ip_filter_shift;
    fword[k] = a.ip_filter[fbit[k] >> 5];
    uint32_t vi = 0;  // class versions below min_version
    for (uint32_t j = 0; j < a.n_versions; ++j) vi += a.ver_sorted[j] < minv[k];
    m[k] = a.env_ver_mask[(size_t)min(env[k], n_env - 1) * (a.n_versions + 1) + vi];
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    probe[k] = make_uint2(0u, kNone);
    if (fword[k] >> (fbit[k] & 31) & 1u) probe[k] = a.ip_hash[hslot[k]];
  }
  uint32_t s[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    while (probe[k].y != kNone && probe[k].x != rip[k]) {  // (a collision: the next slot)
      hslot[k] = (hslot[k] + 1) & hmask;
      probe[k] = a.ip_hash[hslot[k]];
    }
    s[k] = probe[k].y < 0x80000000u ? probe[k].y : 0u;  // the one servant on the host (0: none, read anyway)
  }
  uint32_t b[R], e[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    b[k] = a.slot_base[s[k]];
    e[k] = a.slot_base[s[k] + 1];
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const uint32_t t = t0 + k * blockDim.x;
    const bool live = t < a.n_tasks;
    const uint64_t any = live && env[k] < n_env ? m[k] : 0;
    uint32_t lo = kNone, hi = kNone;
    if (probe[k].y != kNone) {
      if (probe[k].y & 0x80000000u) {
        lo = probe[k].y & 0x7FFFFFFFu;  // several servants on the host: resolved at replay time
        hi = kSelfShared;
      } else if (e[k] > b[k]) {
        lo = b[k];
        hi = e[k];
      }
    }
    if (live) {
      a.mask[t] = any;
      a.self_lo[t] = lo;
      a.self_hi[t] = hi;
    }
    // (chunk_size is a multiple of 64 and the waves of a k start at multiples of 64: all lanes
    // of a wave fall into the same chunk — one atomic per wave, k and part)
    uint64_t consuming = __ballot(any != 0);
    const bool leader = (threadIdx.x & 63) == 0;
    const uint32_t t_wave = t - (threadIdx.x & 63);
    if (a.n_parts <= 1) {
      if (consuming && leader) atomicAdd(&a.chunk_consuming[t / a.chunk_size], (uint32_t)__popcll(consuming));
      if (a.chunk_tail && (t_wave + 64) % a.chunk_size == 0 && t_wave + 64 <= a.n_tasks && leader)
        a.chunk_tail[t / a.chunk_size] = (uint32_t)__popcll(consuming >> (64 - a.tail_len));
    } else {
      const uint32_t part = any ? a.cls_comp[(uint32_t)__builtin_ctzll(any)] : 0xFFFFFFFFu;
      while (consuming) {
        const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)part, (int)__builtin_ctzll(consuming));
        const uint64_t same = __ballot(part == g);
        if (leader)
          atomicAdd(&a.chunk_consuming[(size_t)(t / a.chunk_size) * a.n_parts + g], (uint32_t)__popcll(same));
        consuming &= ~same;
      }
    }
  }
}

// before[k] = number of consuming requests in the chunks before k; before[n_chunks] = their
// total. ONE workgroup: an extra workgroup of the first histogram launch of the sort (which
// runs after the classification anyway), or k_chunk_prefix when nothing is sorted.
struct PrefixArgs {
  const uint32_t* chunk_consuming;  // NULL: nothing to do; [chunk * n_parts + part]
  uint32_t n_chunks;
  uint32_t* before;                 // [(n_chunks + 1) * n_parts], row n_chunks: the totals
  uint32_t n_parts;
  // != 0: chunk_consuming holds one entry per wave of 64 requests (ClassifyArgs::per_wave),
  // n_waves of them, waves_per_chunk to a chunk.
  uint32_t waves_per_chunk, n_waves;
};

__device__ __forceinline__ void chunk_prefix_block(const PrefixArgs& a, DeviceParams* prm) {
  // Each thread owns a run of consecutive chunks (K is a few thousand at most: the chunk
  // size grows with the batch): independent loads, one block scan, one barrier.
  __shared__ uint32_t lds[17];
  const uint32_t G = a.n_parts ? a.n_parts : 1;
  const uint32_t per = (a.n_chunks + blockDim.x - 1) / blockDim.x;
  const uint32_t b = min(a.n_chunks, threadIdx.x * per), e = min(a.n_chunks, b + per);
  uint32_t all = 0;
  // Consuming requests of chunk k, part g.
  auto count = [&](uint32_t k, uint32_t g) {
    if (!a.waves_per_chunk) return a.chunk_consuming[(size_t)k * G + g];
    uint32_t c = 0;
    for (uint32_t w = k * a.waves_per_chunk; w < min(a.n_waves, (k + 1) * a.waves_per_chunk); ++w)
      c += a.chunk_consuming[(size_t)w * G + g];
    return c;
  };
  for (uint32_t g = 0; g < G; ++g) {  // (one part almost always)
    uint32_t sum = 0;
    for (uint32_t k = b; k < e; ++k) sum += count(k, g);
    uint32_t total;
    uint32_t acc = block_exclusive_scan(sum, lds, &total);
    for (uint32_t k = b; k < e; ++k) {
      a.before[(size_t)k * G + g] = acc;
      acc += count(k, g);
    }
    if (threadIdx.x == 0) a.before[(size_t)a.n_chunks * G + g] = total;
    all += total;
  }
  if (threadIdx.x == 0) prm->consuming = all;
}

__global__ __launch_bounds__(1024) void k_chunk_prefix(PrefixArgs a, DeviceParams* prm) {
  chunk_prefix_block(a, prm);
}

// ---------------------------------------------------------------------------
// k_slot_gen: thread per slot, generation order (servant-major, running ascending).
// ---------------------------------------------------------------------------
// Workgroup t < gen_blocks generates the slots of sort tile t (256 x `items` slots, element
// j * 256 + thread like the sort kernels) and, while it has their keys, the tile's histogram
// of the FIRST sort pass (digit = the low `bits0` key bits, or with fused0 class bits above
// bits0 - fused0 key bits when the sort has a single pass): hist[d * gen_blocks + t].
// Workgroups [gen_blocks, gridDim.x) classify requests instead (task_classify_block).
template <typename KeyT>
__global__ __launch_bounds__(256) void k_slot_gen(ServantTable sv, const uint32_t* slot_base,
                                                  DeviceParams* prm, uint32_t exact,
                                                  uint32_t cap_bits, KeyT* keys, uint32_t* vals,
                                                  uint16_t* cls_by_g, uint32_t* owner,
                                                  uint32_t gen_blocks,
                                                  uint32_t items, uint32_t bits0, uint32_t fused0,
                                                  uint32_t gbits, uint32_t* hist, ClassifyArgs ca,
                                                  uint32_t comp_shift, const uint32_t* r_first,
                                                  const uint32_t* gslot_base, uint32_t packed,
                                                  const uint32_t* tile_first) {
  extern __shared__ uint32_t h0[];  // 1 << bits0
#ifdef YDC_PHASE_PROBE
  // Measurement builds (tools/gen_probe.py): start and end of every stride-th workgroup, where
  // k_front_bins leaves its own stamps on the bin-sort path (never both in one batch).
  const uint32_t probe_stride = (gridDim.x + 2399) / 2400;
  const bool probed = threadIdx.x == 0 && blockIdx.x % probe_stride == 0;
  if (probed) ydc_phase_probe[40000 + blockIdx.x / probe_stride * 2] = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ydc_phase_probe[39990] = gridDim.x;
    ydc_phase_probe[39991] = gen_blocks;
    ydc_phase_probe[39992] = probe_stride;
  }
#define YDC_GEN_PROBE_END()                                                                    \
  do {                                                                                         \
    if (probed) ydc_phase_probe[40000 + blockIdx.x / probe_stride * 2 + 1] = wall_clock64();   \
  } while (0)
#else
#define YDC_GEN_PROBE_END() do { } while (0)
#endif
  const uint32_t gen_grid = xcd_grid(gen_blocks);
  if (blockIdx.x >= gen_grid) {
    if (ca.per_thread == 4) task_classify_block_multi<4>(ca, blockIdx.x - gen_grid, prm);
    else task_classify_block(ca, blockIdx.x - gen_grid, prm);
    YDC_GEN_PROBE_END();
    return;
  }
  // Owners: slots are servant-major, so a tile's owners are one short run of servants. Two
  // threads find its ends (the only long chains of dependent loads), the run's slot_base
  // entries go to LDS, and every slot searches there.
  constexpr uint32_t kWindow = 2048;
  __shared__ uint32_t win[kWindow];
  __shared__ uint32_t run_ends[2];
  const uint32_t radix = 1u << bits0, kbits = bits0 - fused0;
  for (uint32_t d = threadIdx.x; d < radix; d += blockDim.x) h0[d] = 0;
  const uint32_t M = prm->n_slots;
  const uint32_t tile = xcd_tile(blockIdx.x, gen_blocks, ca.xcd_gen), base = tile * (blockDim.x * items);
  if (tile >= gen_blocks) {
    YDC_GEN_PROBE_END();
    return;
  }
  const uint32_t g_end = min(M, base + blockDim.x * items);  // (base >= M: nothing to generate)
  if (base < M) {
    if (tile_first) {  // (k_servant_scan left the owners of the tiles' first slots behind)
      if (threadIdx.x == 0) run_ends[0] = tile_first[tile];
      if (threadIdx.x == 64) run_ends[1] = tile_first[tile + 1];  // >= the owner of slot g_end - 1
    } else {
      if (threadIdx.x == 0) run_ends[0] = owner_of_slot(slot_base, sv.n, base);
      if (threadIdx.x == 64) run_ends[1] = owner_of_slot(slot_base, sv.n, g_end - 1);
    }
  }
  __syncthreads();
  const uint32_t s_first = base < M ? run_ends[0] : 0;
  const uint32_t n_run = base < M ? run_ends[1] - s_first + 1 : 0;
  const bool windowed = n_run <= kWindow;  // (zero-slot servants in between can make it long)
  if (windowed)
    for (uint32_t i = threadIdx.x; i < n_run; i += blockDim.x) win[i] = slot_base[s_first + i];
  __syncthreads();
  for (uint32_t j = 0; j < items; ++j) {
    const uint32_t g = base + j * blockDim.x + threadIdx.x;
    if (g >= M) break;
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
    // Key window (multi-GPU, k_window): slot_base is the LOCAL prefix, the servant's first local
    // slot is r_first[s], and slots are named by their index in the whole registry's
    // generation order (gslot_base) — the same name on every rank.
    const uint32_t run0 = sv.running[s];
    uint32_t r = (r_first ? r_first[s] : run0) + (g - slot_base[s]);
    const uint32_t gg = gslot_base ? gslot_base[s] + (r - run0) : g;
    owner[gg] = s;
    uint32_t nproc = sv.nproc[s], flags = sv.flags[s];
    uint32_t cap = slot_capacity(nproc, sv.load[s], sv.max_tasks[s], r);
    uint32_t tier = slot_tier(nproc, flags, r);
    uint64_t key = exact ? slot_key_exact(tier, r, cap, cap_bits) : slot_key_fp64(tier, r, cap);
    const uint32_t cls = sv.class_of[s];
    // Several independent parts: slots are ordered part-major (the id rides above the key).
    if (ca.n_parts > 1) key |= (uint64_t)ca.cls_comp[cls] << comp_shift;
    const uint32_t sort_val = gbits ? (cls << gbits) | gg : gg;
    if (sizeof(KeyT) == 4 && packed) {
      // 32-bit keys: (key, value) side by side, one 8-byte store — and one per sort pass
      // instead of two scattered 4-byte ones (each of those moves a 32-byte sector).
      reinterpret_cast<uint2*>(keys)[g] = make_uint2((uint32_t)key, sort_val);
    } else {
      keys[g] = (KeyT)key;
      vals[g] = sort_val;
    }
    // (The sort's value is the slot; with room above its bits (gbits != 0) the class rides
    // there, so that class digits need no gather, SortIn::gbits.)
    if (cls_by_g) cls_by_g[gg] = (uint16_t)cls;
    uint32_t d = (uint32_t)key & ((1u << kbits) - 1);
    if (fused0) d |= cls << kbits;
    atomicAdd(&h0[d], 1u);
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < radix; d += blockDim.x) hist[d * gen_blocks + tile] = h0[d];
  YDC_GEN_PROBE_END();
}
#undef YDC_GEN_PROBE_END

// ---------------------------------------------------------------------------
// Stable LSD radix sort, three kernels per pass, digit width chosen per pass (up
// to kMaxRadixBits: 17-bit keys sort in two 9-bit passes, 21-bit keys in 11 + 10).
// Digit of element i: bits [shift, shift + bits) of keys[i], or (class pass) of
// cls_by_g[vals[i]] with the element's key being its index (== global rank).
// ---------------------------------------------------------------------------
constexpr int kMaxRadixBits = 11;
constexpr int kMaxFusedClsBits = 5;   // class bits k_radix_scatter_classed can rank on
constexpr uint32_t kMaxFusedClasses = 8;  // ... and the class count up to which the planner asks it to

// 32-bit keys travel with their values in 8-byte records (packed: `keys` points to uint2
// {key, value}, `vals` is unused); 64-bit keys (fp64 format) keep two arrays.
template <typename KeyT>
struct SortIn {
  const KeyT* keys;         // NULL: key == index (+ rank_offset) — first class pass, unpacked
  const uint32_t* vals;
  const uint16_t* cls_by_g; // non-NULL in the class pass
  uint32_t shift;
  uint32_t bits;            // digit width of this pass
  uint32_t items;           // elements per thread (tile = 256 * items), <= kSortItems
  // Last key pass with the class folded in (few classes, room left in the digit): the digit
  // is (class << (bits - fused_cls_bits)) | key bits, see k_radix_scatter_classed. 0: off.
  uint32_t fused_cls_bits;
  // gbits != 0: vals carry the class above gbits slot bits (class digits come from there and
  // cls_by_g is only the "class digit needed" flag). out_mask: applied to the values this pass
  // writes — the last pass of the sort strips the class again.
  uint32_t gbits, out_mask;
  uint32_t packed;        // input is records
  uint32_t key_is_index;  // first class pass: the key is the element's index (its global rank)
  uint32_t xcd_hist, xcd_scatter;  // XCD-contiguous tile order in the histogram / scatter launch
  // Measurement only (tests/tools/scatter_probe.hip; 0 in the library): 1 = the scatter's stores go
  // to the element's own place (coalesced), 2 = no gather of the tile's histogram column, 4 = no
  // stores at all.
  uint32_t dbg;
};

template <typename KeyT>
__device__ __forceinline__ void sort_load(const SortIn<KeyT>& in, uint32_t i, uint32_t rank_off,
                                          bool want_val, KeyT& key, uint32_t& val) {
  if (in.packed) {
    const uint2 r = reinterpret_cast<const uint2*>(in.keys)[i];
    key = in.key_is_index ? (KeyT)(i + rank_off) : (KeyT)r.x;
    val = r.y;
  } else {
    key = in.keys && !in.key_is_index ? in.keys[i] : (KeyT)(i + rank_off);
    val = want_val ? in.vals[i] : 0u;
  }
}
template <typename OutKeyT>
__device__ __forceinline__ void sort_store(OutKeyT* out_keys, uint32_t* out_vals, uint32_t packed,
                                           uint32_t pos, OutKeyT key, uint32_t val) {
  if (packed) {
    reinterpret_cast<uint2*>(out_keys)[pos] = make_uint2((uint32_t)key, val);
  } else {
    out_keys[pos] = key;
    out_vals[pos] = val;
  }
}

template <typename KeyT>
__device__ __forceinline__ uint32_t sort_class(const SortIn<KeyT>& in, uint32_t val) {
  return in.gbits ? val >> in.gbits : (uint32_t)in.cls_by_g[val];
}

template <typename KeyT>
__device__ __forceinline__ uint32_t sort_digit(const SortIn<KeyT>& in, uint32_t i, KeyT key,
                                               uint32_t val) {
  const uint32_t mask = (1u << in.bits) - 1;
  if (in.fused_cls_bits) {
    const uint32_t kbits = in.bits - in.fused_cls_bits;
    return (sort_class(in, val) << kbits) | ((uint32_t)(key >> in.shift) & ((1u << kbits) - 1));
  }
  if (in.cls_by_g) return (sort_class(in, val) >> in.shift) & mask;
  return (uint32_t)(key >> in.shift) & mask;
}

// hist[d * n_tiles + tile] = number of elements of the tile with digit d.
// Workgroup n_tiles (if launched): the chunk prefix (chunk_prefix_block).
template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(SortIn<KeyT> in, DeviceParams* prm,
                                                             uint32_t n_tiles, uint32_t* hist,
                                                             PrefixArgs pa) {
  extern __shared__ uint32_t h[];  // radix
  if (blockIdx.x == xcd_grid(n_tiles)) {
    chunk_prefix_block(pa, prm);
    return;
  }
  const uint32_t radix = 1u << in.bits;
  const uint32_t M = prm->n_slots;
  const uint32_t tile = xcd_tile(blockIdx.x, n_tiles, in.xcd_hist);
  if (tile >= n_tiles) return;
  for (uint32_t d = threadIdx.x; d < radix; d += kSortThreads) h[d] = 0;
  __syncthreads();
  const uint32_t base = tile * (kSortThreads * in.items);
  if (base < M) {
    // All loads first (clamped, unconditional), so that they are in flight together.
    KeyT key[kSortItems];
    uint32_t val[kSortItems];
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
      const uint32_t i = min(base + j * kSortThreads + threadIdx.x, M - 1);
      sort_load(in, i, prm->rank_offset, in.cls_by_g != nullptr, key[j], val[j]);
    }
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
      const uint32_t i = base + j * kSortThreads + threadIdx.x;
      if ((uint32_t)j < in.items && i < M) atomicAdd(&h[sort_digit(in, i, key[j], val[j])], 1u);
    }
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < radix; d += kSortThreads) hist[d * n_tiles + tile] = h[d];
}

// One workgroup per digit: exclusive scan of its row of tile counts, row total.
__global__ __launch_bounds__(256) void k_radix_scan(uint32_t n_tiles, uint32_t* hist,
                                                    uint32_t* row_total) {
  __shared__ uint32_t lds[17];
  uint32_t* row = hist + (size_t)blockIdx.x * n_tiles;
  const uint32_t per = (n_tiles + blockDim.x - 1) / blockDim.x;
  const uint32_t b = threadIdx.x * per;
  const uint32_t e = min(n_tiles, b + per);
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; ++i) sum += row[i];
  uint32_t total;
  uint32_t acc = block_exclusive_scan(sum, lds, &total);
  for (uint32_t i = b; i < e; ++i) {
    uint32_t v = row[i];
    row[i] = acc;
    acc += v;
  }
  if (threadIdx.x == 0) row_total[blockIdx.x] = total;
}

// Scatter. Within a tile, wave w owns items*64 consecutive elements and walks them
// 64 at a time, so (tile, wave, round, lane) order == index order and the rank of
// an element among equal digits is: digit start + earlier tiles (scanned hist) +
// earlier waves + earlier rounds of this wave + lower lanes with the same digit
// (wave ballot match) — no atomics, deterministic, stable.
// (65 VGPRs = seven waves per SIMD; forced to 64 it spills 26 of them: cfg4's four passes 145 -> 185 us)
template <typename KeyT, typename OutKeyT>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(
    SortIn<KeyT> in, const DeviceParams* prm, uint32_t n_tiles, const uint32_t* hist,
    const uint32_t* row_total, OutKeyT* out_keys, uint32_t* out_vals) {
  extern __shared__ uint32_t sm[];  // cnt[kSortWaves][radix] | dstart[radix]
  __shared__ uint32_t lds[17];
  const uint32_t radix = 1u << in.bits;
  uint32_t* cnt = sm;
  uint32_t* dstart = sm + kSortWaves * radix;
  const uint32_t M = prm->n_slots;
  const uint32_t tile = xcd_tile(blockIdx.x, n_tiles, in.xcd_scatter);
  const uint32_t base = tile * (kSortThreads * in.items);
  if (tile >= n_tiles || base >= M) return;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t wave_span = in.items * 64;  // consecutive elements a wave owns
  {
    // Start of every digit: exclusive scan of the row totals (each thread owns
    // radix / 256 consecutive digits) + the elements of earlier tiles.
    const uint32_t per = (radix + kSortThreads - 1) / kSortThreads;
    const uint32_t d0 = threadIdx.x * per, d1 = min(radix, d0 + per);
    uint32_t sum = 0;
    for (uint32_t d = d0; d < d1; ++d) sum += row_total[d];
    uint32_t total;
    uint32_t acc = block_exclusive_scan(sum, lds, &total);
    for (uint32_t d = d0; d < d1; ++d) {
      dstart[d] = acc + ((in.dbg & 2) ? 0u : hist[d * n_tiles + tile]);
      acc += row_total[d];
    }
    for (uint32_t d = threadIdx.x; d < kSortWaves * radix; d += kSortThreads) cnt[d] = 0;
  }
  __syncthreads();
  KeyT key[kSortItems];
  uint32_t val[kSortItems], dig[kSortItems], rank[kSortItems];
  const uint64_t lt_mask = (1ull << lane) - 1;
  uint32_t* wcnt = cnt + wave * radix;
  // All loads first (clamped, unconditional): eight round trips in flight together instead of
  // one behind the other, each in front of its ranking step (tests/tools/scatter_probe.hip: the
  // pass without its stores took 25 us of 44 for 5M records — eight dependent memory latencies).
  const uint32_t rank_off = prm->rank_offset;
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t i = min(base + wave * wave_span + j * 64 + lane, M - 1);
    sort_load(in, i, rank_off, true, key[j], val[j]);
  }
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t i = base + wave * wave_span + j * 64 + lane;
    const bool valid = (uint32_t)j < in.items && i < M;
    uint32_t d = 0;
    if (valid) d = sort_digit(in, i, key[j], val[j]);
    dig[j] = d;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < kMaxRadixBits; ++b) {
      if ((uint32_t)b < in.bits) {
        const uint64_t m = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
      }
    }
    uint32_t before = 0;
    if (valid) before = wcnt[d];
    rank[j] = before + (uint32_t)__popcll(peers & lt_mask);
    // The lowest lane of each peer group publishes the group's size. All lanes
    // of this wave have read wcnt[d] in the instruction above.
    if (valid && (peers & lt_mask) == 0) wcnt[d] = before + (uint32_t)__popcll(peers);
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < radix; d += kSortThreads) {
    uint32_t off = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) {
      uint32_t t = cnt[w * radix + d];
      cnt[w * radix + d] = off;
      off += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t i = base + wave * wave_span + j * 64 + lane;
    if ((uint32_t)j < in.items && i < M) {
      const uint32_t pos = dstart[dig[j]] + wcnt[dig[j]] + rank[j];
      if (in.dbg & 4) { if (pos == 0xFFFFFFF0u) sort_store(out_keys, out_vals, in.packed, 0, (OutKeyT)key[j], val[j]); }
      else if (in.dbg & 1) sort_store(out_keys, out_vals, in.packed, base + wave * wave_span + j * 64 + lane, (OutKeyT)key[j], val[j] & in.out_mask);
      else
      sort_store(out_keys, out_vals, in.packed, pos, (OutKeyT)key[j], val[j] & in.out_mask);
    }
  }
}

// Last key pass and class partition in one (SortIn::fused_cls_bits != 0, a handful of classes).
// The digit is (class, top key bits), so the elements land class-major and key-sorted within
// the class: the per-class lists. What the separate class pass got for free — an element's
// GLOBAL rank, i.e. its position in the key-only order — is computed beside it: the same
// stable ranking on the key bits of the digit alone (digit start and earlier tiles: sums over
// the classes of the scanned histogram; within the tile: a second ballot match). Writes
// out_rank[pos] = global rank, out_vals[pos] = slot, rank_to_g[global rank] = slot.
template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter_classed(
    SortIn<KeyT> in, const DeviceParams* prm, uint32_t n_tiles, const uint32_t* hist,
    const uint32_t* row_total, uint32_t* out_rank, uint32_t* out_vals, uint32_t* rank_to_g) {
  // cnt[kSortWaves][radix] | dstart[radix] | kcnt[kSortWaves][kradix] | kstart[kradix]
  extern __shared__ uint32_t sm[];
  __shared__ uint32_t lds[17];
  const uint32_t radix = 1u << in.bits;
  const uint32_t kbits = in.bits - in.fused_cls_bits;
  const uint32_t kradix = 1u << kbits, n_cls = 1u << in.fused_cls_bits;
  uint32_t* cnt = sm;
  uint32_t* dstart = sm + kSortWaves * radix;
  uint32_t* kcnt = dstart + radix;
  uint32_t* kstart = kcnt + kSortWaves * kradix;
  const uint32_t M = prm->n_slots;
  const uint32_t tile = xcd_tile(blockIdx.x, n_tiles, in.xcd_scatter);
  const uint32_t base = tile * (kSortThreads * in.items);
  if (tile >= n_tiles || base >= M) return;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t wave_span = in.items * 64;
  {
    const uint32_t per = (radix + kSortThreads - 1) / kSortThreads;
    const uint32_t d0 = threadIdx.x * per, d1 = min(radix, d0 + per);
    uint32_t sum = 0;
    for (uint32_t d = d0; d < d1; ++d) sum += row_total[d];
    uint32_t total;
    uint32_t acc = block_exclusive_scan(sum, lds, &total);
    for (uint32_t d = d0; d < d1; ++d) {
      dstart[d] = acc + hist[d * n_tiles + tile];
      acc += row_total[d];
    }
    // (row totals and this tile's column of the scanned table, once more in LDS — cnt[] is
    // not in use yet —: the sums over the classes below read them 2 n_cls times per key digit)
    uint32_t* rt_l = cnt;
    uint32_t* h_l = cnt + radix;
    for (uint32_t d = threadIdx.x; d < radix; d += kSortThreads) {
      rt_l[d] = row_total[d];
      h_l[d] = hist[d * n_tiles + tile];
    }
    __syncthreads();  // (and lds[] is reused by the second scan)
    // Key-only digit starts: all classes' elements with a smaller key digit + the elements
    // of earlier tiles with the same key digit.
    const uint32_t kper = (kradix + kSortThreads - 1) / kSortThreads;
    const uint32_t k0 = min(kradix, threadIdx.x * kper), k1 = min(kradix, k0 + kper);
    sum = 0;
    for (uint32_t d = k0; d < k1; ++d)
      for (uint32_t c = 0; c < n_cls; ++c) sum += rt_l[(c << kbits) | d];
    acc = block_exclusive_scan(sum, lds, &total);
    for (uint32_t d = k0; d < k1; ++d) {
      uint32_t all = 0, earlier = 0;
      for (uint32_t c = 0; c < n_cls; ++c) {
        all += rt_l[(c << kbits) | d];
        earlier += h_l[(c << kbits) | d];
      }
      kstart[d] = acc + earlier;
      acc += all;
    }
    __syncthreads();  // rt_l / h_l are cnt[]: read before it is cleared
    for (uint32_t d = threadIdx.x; d < kSortWaves * radix; d += kSortThreads) cnt[d] = 0;
    for (uint32_t d = threadIdx.x; d < kSortWaves * kradix; d += kSortThreads) kcnt[d] = 0;
  }
  __syncthreads();
  uint32_t val[kSortItems], dig[kSortItems], rank[kSortItems], krank[kSortItems];
  const uint64_t lt_mask = (1ull << lane) - 1;
  uint32_t* wcnt = cnt + wave * radix;
  uint32_t* wkcnt = kcnt + wave * kradix;
  KeyT keyv[kSortItems];
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {  // (all loads first, in flight together: see k_radix_scatter)
    const uint32_t i = min(base + wave * wave_span + j * 64 + lane, M - 1);
    sort_load(in, i, prm->rank_offset, true, keyv[j], val[j]);
  }
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t i = base + wave * wave_span + j * 64 + lane;
    const bool valid = (uint32_t)j < in.items && i < M;
    uint32_t d = 0;
    if (valid) d = sort_digit(in, i, keyv[j], val[j]);
    dig[j] = d;
    uint64_t kpeers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < kMaxRadixBits; ++b) {
      if ((uint32_t)b < kbits) {
        const uint64_t m = __ballot((d >> b) & 1u);
        kpeers &= ((d >> b) & 1u) ? m : ~m;
      }
    }
    uint64_t peers = kpeers;  // same key digit; now the same class as well
#pragma unroll
    for (int b = 0; b < kMaxFusedClsBits; ++b) {
      if ((uint32_t)b < in.fused_cls_bits) {
        const uint64_t m = __ballot((d >> (kbits + b)) & 1u);
        peers &= ((d >> (kbits + b)) & 1u) ? m : ~m;
      }
    }
    const uint32_t kd = d & (kradix - 1);
    uint32_t before = 0, kbefore = 0;
    if (valid) {
      before = wcnt[d];
      kbefore = wkcnt[kd];
    }
    rank[j] = before + (uint32_t)__popcll(peers & lt_mask);
    krank[j] = kbefore + (uint32_t)__popcll(kpeers & lt_mask);
    if (valid && (peers & lt_mask) == 0) wcnt[d] = before + (uint32_t)__popcll(peers);
    if (valid && (kpeers & lt_mask) == 0) wkcnt[kd] = kbefore + (uint32_t)__popcll(kpeers);
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < radix + kradix; d += kSortThreads) {
    uint32_t* col = d < radix ? cnt + d : kcnt + (d - radix);
    const uint32_t stride = d < radix ? radix : kradix;
    uint32_t off = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) {
      uint32_t t = col[w * stride];
      col[w * stride] = off;
      off += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t i = base + wave * wave_span + j * 64 + lane;
    if ((uint32_t)j < in.items && i < M) {
      const uint32_t kd = dig[j] & (kradix - 1);
      const uint32_t pos = dstart[dig[j]] + wcnt[dig[j]] + rank[j];
      const uint32_t grank = kstart[kd] + wkcnt[kd] + krank[j];
      sort_store(out_rank, out_vals, in.packed, pos, grank + prm->rank_offset, val[j] & in.out_mask);
      rank_to_g[grank] = val[j] & in.out_mask;
    }
  }
}

// Thread per (chunk, class): level guess, everything dirty.
// base (nullable): consuming requests of the ranks before this one (multi-GPU).
// (per part of the registry: before[k * n_parts + g], base[g], parts.rank_base[g])
__global__ __launch_bounds__(256) void k_guess_init(ClassLists L, const uint32_t* before,
                                                    uint32_t n_chunks, const uint32_t* base,
                                                    PartTable parts, ClassState* guess,
                                                    uint8_t* dirty) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t C = L.n_classes;
  if (i >= n_chunks * C) return;
  uint32_t k = i / C, c = i - k * C;
  const uint32_t G = parts.n_parts, g = G > 1 ? parts.cls_comp[c] : 0u;
  guess[i] = level_guess(L, c, (G > 1 ? parts.rank_base[g] : 0u) + before[(size_t)k * G + g] +
                                   (base ? base[g] : 0u));
  if (c == 0) dirty[k] = 1;
}

// Thread per chunk, any number of classes (slow path: more than kMaxWaveClasses classes).
__global__ __launch_bounds__(64) void k_sim_generic(ClassLists L, TaskTable T, uint32_t n_tasks,
                                                    uint32_t chunk_size, uint32_t n_chunks,
                                                    const ClassState* guess, ClassState* endst,
                                                    uint8_t* dirty, uint32_t* slot_of,
                                                    ClassRun* runs, SharedIpTable shared,
                                                    uint32_t round, DeviceParams* prm) {
  if (blockIdx.x == 0 && threadIdx.x == 0) prm->n_changed[round & 63] = 0;
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_chunks) return;
  if (!dirty[k]) return;
  const uint32_t C = L.n_classes;
  const uint32_t t0 = k * chunk_size, t1 = min(n_tasks, t0 + chunk_size);
  sim_chunk(L, T, t0, t1, guess + (size_t)k * C, endst + (size_t)k * C, slot_of,
            runs + (size_t)k * C, shared.pos_last ? &shared : nullptr);
  dirty[k] = 0;
  atomicAdd(&prm->chunk_sims, 1u);
}

// pos_last[s] = position in the class lists of servant s's last slot (registries with hosts
// that run several servants only: SharedIpTable). Thread per list position.
__global__ __launch_bounds__(256) void k_pos_last(ClassLists L, const uint32_t* owner,
                                                  const uint32_t* slot_base, const DeviceParams* prm,
                                                  uint32_t* pos_last) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= prm->n_slots) return;
  const uint32_t g = list_slot(L, i), s = owner[g];
  if (g + 1 == slot_base[s + 1]) pos_last[s] = i;
}

// ---------------------------------------------------------------------------
// k_update: thread per (chunk, class). The start guess of chunk k+1 becomes the
// end state chunk k reached in its latest simulation; chunks whose guess changed
// are marked dirty. Two replays of a chunk from different start states fall
// into step after a few dozen requests (every request takes the smallest
// admissible slot, which pulls lagging classes level), so end states are right
// long before start states are and a few rounds suffice. A round in which no
// guess changes proves guess[k+1] == end[k] for every k, i.e. the chunk results
// are exactly the sequential ones (chunk 0 always starts from the true state).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update(uint32_t n_classes, uint32_t n_chunks,
                                                const ClassState* endst, ClassState* guess,
                                                uint8_t* dirty, uint32_t round,
                                                DeviceParams* prm) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool changed = false;
  if (n_chunks && i < (n_chunks - 1) * n_classes) {
    const ClassState en = endst[i];  // (k, c) -> guess of (k + 1, c)
    ClassState* g = guess + i + n_classes;
    if (!class_state_equal(*g, en)) {
      *g = en;
      dirty[i / n_classes + 1] = 1;
      changed = true;
    }
  }
  const uint64_t b = __ballot(changed);
  if (b && (threadIdx.x & 63) == 0) atomicAdd(&prm->n_changed[round & 63], (uint32_t)__popcll(b));
}

// ---------------------------------------------------------------------------
// k_finalize: workgroups [0, req_blocks): thread per request — slot_of[t] is the global rank
// of the request's slot (matching passes) or its generation index (thread-per-chunk path:
// slot_is_rank == 0); rank -> slot through the key-sorted order, slot -> servant through the
// owner table slot_gen left; utilisation. Workgroups behind them: thread per servant —
// running_tasks after the batch, in closed form from the final class states: a class list is
// consumed from the front (holes aside), so the slots the batch took on a servant are those
// that sort before the entry at its class's cursor (servant_slots_before) — no per-slot
// flags, no atomics on servant counters.
// ---------------------------------------------------------------------------
struct RunningArgs {
  const ClassState* end_state;    // [C] state after this context's last request (NULL: no request)
  const ClassState* start_state;  // [C] multi-GPU: state before its first request (NULL: batch start)
  ClassLists L;
  const uint32_t* gslot_base;     // registry-wide slot names (== slot_base unless the sort is sharded)
  const uint32_t* cls_comp;
  uint32_t n_parts, comp_shift, exact, cap_bits;
  uint32_t n_servants;
  uint32_t* running_out;  // running + taken
  uint32_t* out_a;        // nullable: the caller's copy
  uint32_t* out_b;        // nullable: a second copy (never the resident column: see k_finalize)
  uint32_t* taken_out;    // nullable: multi-GPU, this rank's slot delta
  // Pipelined batches: a batch that is not final latches DeviceParams::pipeline_broken, and no
  // batch takes effect while it is set.
  uint32_t pipelined;
  // Non-NULL: the device address of a page-locked DeviceParams — the launch's last servant
  // workgroup stores the batch's outcome there itself (the 560-byte read-back was a blit kernel
  // of its own: 3.9 us of a 60 us step).
  DeviceParams* host_outcome;
  uint32_t srv_blocks;
};

// Slots of servant s (class c) that sort before what `st` has not consumed yet.
__device__ __forceinline__ uint32_t taken_until(const ServantTable& sv, const RunningArgs& ra,
                                                const uint32_t* owner, uint32_t s, uint32_t c,
                                                uint32_t nproc, uint32_t load, uint32_t max_tasks,
                                                uint32_t running, uint32_t flags,
                                                const ClassState& st) {
  // Holes are slots of ONE servant that were stepped over: for that servant everything from
  // the first hole on is unconsumed; for everybody else everything below the cursor is taken.
  const bool owner_of_holes = st.lo < st.cursor && st.hown_lo == ra.gslot_base[s];
  const uint32_t x = owner_of_holes ? st.lo : st.cursor;
  const uint32_t end = ra.L.cls_begin[c + 1];
  if (x >= end) return servant_slot_count(nproc, load, max_tasks, running, flags);  // all of them
  const uint32_t g = list_slot(ra.L, x), hs = owner[g];
  const uint32_t hr = sv.running[hs] + (g - ra.gslot_base[hs]);
  const uint64_t part_key = ra.n_parts > 1 ? (uint64_t)ra.cls_comp[c] << ra.comp_shift : 0ull;
  const uint64_t hkey = slot_sort_key(sv.nproc[hs], sv.load[hs], sv.max_tasks[hs], sv.flags[hs], hr,
                                      part_key, ra.exact != 0, ra.cap_bits);
  return servant_slots_before(nproc, load, max_tasks, running, flags, s, part_key, hkey, hs,
                              ra.exact != 0, ra.cap_bits);
}

__global__ __launch_bounds__(256) void k_finalize(ServantTable sv, const uint32_t* slot_base,
                                                  const uint32_t* owner, const uint32_t* rank_to_g,
                                                  const uint32_t* slot_of, uint32_t n_tasks,
                                                  uint32_t slot_is_rank, uint32_t* out_idx,
                                                  double* out_util, uint32_t check_slot,
                                                  DeviceParams* prm, uint32_t g_mask,
                                                  uint32_t req_blocks, RunningArgs ra,
                                                  uint32_t rank_stride) {
  // Pre-launched behind the matching passes: only takes effect once they have converged (and,
  // with a sharded sort, only if every rank's key window covered what its requests reached).
  const bool final = (check_slot == kNone || prm->n_changed[check_slot] == 0) && !prm->window_miss &&
                     !(ra.pipelined && (prm->pipeline_broken || prm->overflow));
  if (ra.pipelined && !final && blockIdx.x == req_blocks && threadIdx.x == 0)
    __hip_atomic_store(&prm->pipeline_broken, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x < req_blocks) {
    if (!final) return;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tasks) return;
    uint32_t g = slot_of[t];
    if (g >= kIdxEnvNotFound) {
      if (out_idx) out_idx[t] = g;
      if (out_util) out_util[t] = -1.0;
    } else {
      // (rank_stride 2: the values of the key-sorted records; class bits above the slot)
      if (slot_is_rank) g = rank_to_g[(size_t)(g - prm->rank_offset) * rank_stride] & g_mask;
      const uint32_t s = owner[g];
      if (out_idx) out_idx[t] = s;
      if (out_util) {
        uint32_t r = sv.running[s] + (g - slot_base[s]);
        out_util[t] = slot_utilization(r, slot_capacity(sv.nproc[s], sv.load[s], sv.max_tasks[s], r));
      }
    }
    return;
  }
  __shared__ uint32_t lds[17];
  const uint32_t s = (blockIdx.x - req_blocks) * blockDim.x + threadIdx.x;
  uint32_t taken = 0, running = 0;
  if (s < ra.n_servants) {
    running = sv.running[s];
    const uint32_t c = sv.class_of[s];
    if (final && ra.end_state && c != kNone) {
      const uint32_t nproc = sv.nproc[s], load = sv.load[s], mt = sv.max_tasks[s], fl = sv.flags[s];
      taken = taken_until(sv, ra, owner, s, c, nproc, load, mt, running, fl, ra.end_state[c]);
      if (ra.start_state)
        taken -= taken_until(sv, ra, owner, s, c, nproc, load, mt, running, fl, ra.start_state[c]);
    }
    // When the passes have not converged yet this writes running unchanged everywhere and
    // the launch is repeated later. None of the outputs may alias sv.running: taken_until
    // reads the running value of OTHER servants (the head of the class list).
    const uint32_t v = running + taken;
    ra.running_out[s] = v;
    if (ra.out_a) ra.out_a[s] = v;
    if (ra.out_b) ra.out_b[s] = v;
    if (ra.taken_out) ra.taken_out[s] = taken;
  }
  // One counter update per workgroup (same-address atomics serialise at ~10 ns each).
  uint32_t total;
  (void)block_exclusive_scan(taken, lds, &total);
  if (threadIdx.x == 0 && total) atomicAdd(&prm->granted, total);
  if (ra.host_outcome) {
    // The servant workgroup that reports last has everybody's grants behind it: it hands the
    // counters to the host (plain stores to page-locked memory; the host waits for the launch).
    __shared__ uint32_t last;
    if (threadIdx.x == 0) {
      __threadfence();
      last = atomicAdd(&prm->fin_reports, 1u) == ra.srv_blocks - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (last) {
      if (threadIdx.x == 0) {
        const unsigned long long t = wall_clock64();
        prm->t_end_lo = (uint32_t)t;
        prm->t_end_hi = (uint32_t)(t >> 32);
      }
      __syncthreads();
      __threadfence();
      const uint32_t* src = (const uint32_t*)prm;
      uint32_t* dst = (uint32_t*)ra.host_outcome;
      for (uint32_t i = threadIdx.x; i < sizeof(DeviceParams) / 4; i += blockDim.x)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x == 0) __hip_atomic_store(&prm->fin_reports, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Registry maintenance.
// Servant expiry (task_dispatcher.cc:503-516): rows removed[0..n_removed) (ascending) vanish,
// the others move up in order. Thread per old row, all six columns.
struct CompactCols {
  uint32_t* col[6];
};
__global__ __launch_bounds__(256) void k_compact_rows(CompactCols in, CompactCols out,
                                                      const uint32_t* removed, uint32_t n_removed,
                                                      uint32_t n) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t before = lower_bound_u32(removed, n_removed, s);  // removed rows below s
  if (before < n_removed && removed[before] == s) return;         // s itself goes
#pragma unroll
  for (int k = 0; k < 6; ++k) out.col[k][s - before] = in.col[k][s];
}

__global__ __launch_bounds__(256) void k_release_slots(const uint32_t* servant_idx, uint32_t n,
                                                       uint32_t n_servants, uint32_t* running) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && servant_idx[i] < n_servants) atomicSub(&running[servant_idx[i]], 1u);
}
// The same for long lists on registries whose counters fit a workgroup's LDS (<= 16384 servants):
// 16384 released slots per workgroup are counted there first and every servant's total goes out
// as ONE atomic — 100k releases on 2000 servants are 50 decrements of every counter, and
// same-address atomics take ~12 ns each (MI355X_MICROARCH.md: fan-in).
constexpr uint32_t kReleaseTile = 16384;
__global__ __launch_bounds__(1024) void k_release_slots_counted(const uint32_t* servant_idx, uint32_t n,
                                                                uint32_t n_servants, uint32_t* running) {
  extern __shared__ uint32_t rel_cnt[];  // [n_servants]
  for (uint32_t s = threadIdx.x; s < n_servants; s += blockDim.x) rel_cnt[s] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kReleaseTile;
#pragma unroll 4
  for (uint32_t j = threadIdx.x; j < kReleaseTile; j += 1024) {
    const uint32_t i = base + j;
    if (i < n) {
      const uint32_t s = servant_idx[i];
      if (s < n_servants) atomicAdd(&rel_cnt[s], 1u);
    }
  }
  __syncthreads();
  for (uint32_t s = threadIdx.x; s < n_servants; s += blockDim.x) {
    const uint32_t k = rel_cnt[s];
    if (k) atomicSub(&running[s], k);
  }
}

// ---------------------------------------------------------------------------
// Multi-GPU helpers (rank-range sharding of one batch).
// ---------------------------------------------------------------------------
// What every rank tells the others before the batch (all-gathered, `stride` = n_parts + 1 words
// per rank): consuming requests per part of the registry, then its number of requests.
__global__ void k_rank_meta(const uint32_t* before_totals, uint32_t n_parts, uint32_t n_requests,
                            uint32_t* meta) {
  if (blockIdx.x == 0 && threadIdx.x <= n_parts)
    meta[threadIdx.x] = threadIdx.x < n_parts ? before_totals[threadIdx.x] : n_requests;
}
// The nearest rank below `rank` that has requests (-1: none). Ranks without requests are
// transparent: nobody reads what they publish, so they add no hop to the chain of end states.
__device__ __forceinline__ int rank_predecessor(const uint32_t* meta, uint32_t stride, uint32_t rank) {
  for (int q = (int)rank - 1; q >= 0; --q)
    if (meta[(size_t)q * stride + stride - 1] != 0) return q;
  return -1;
}
// base[g] = consuming requests of part g on the ranks before `rank` (meta[r * stride + g]).
__global__ void k_rank_base(const uint32_t* meta, uint32_t rank, uint32_t n_parts, uint32_t stride,
                            uint32_t* base) {
  if (threadIdx.x < n_parts && blockIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t r = 0; r < rank; ++r) acc += meta[(size_t)r * stride + threadIdx.x];
    base[threadIdx.x] = acc;
  }
}
// What a rank publishes after a matching pass: the end state of its last chunk (C entries; a
// rank without requests publishes its start state, which nobody reads — such ranks are
// transparent, see rank_predecessor) followed by one entry whose cursor is the number of
// chunks the pass found inconsistent.
// shift (nullable; sharded sort): list positions are local to the rank's key window; what is
// published is the position in the whole registry's lists, local + shift[c].
__global__ __launch_bounds__(256) void k_pack_boundary(ClassLists L, const ClassState* endst,
                                                       uint32_t n_chunks, const ClassState* boundary_in,
                                                       const DeviceParams* prm, uint32_t pass,
                                                       ClassState* out, const uint32_t* shift) {
  const uint32_t C = L.n_classes;
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    ClassState s;
    if (n_chunks) {
      s = endst[(size_t)(n_chunks - 1) * C + c];
    } else if (boundary_in) {
      s = boundary_in[c];
    } else {
      s.cursor = s.lo = L.cls_begin[c];
      s.hown_lo = s.hown_hi = kNone;
    }
    if (shift) {
      s.cursor += shift[c];
      s.lo += shift[c];
    }
    out[c] = s;
  } else if (c == C) {
    ClassState s;
    s.cursor = prm->n_changed[pass & 63];
    s.lo = prm->overflow;
    s.hown_lo = prm->window_miss;  // (sharded sort: this rank's window exceeded its workspace)
    s.hown_hi = 0;
    out[C] = s;
  }
}
// After the all-gather of a pass's records: the pass was final only if NO rank changed an end
// state. Overwrites this rank's flag of the pass with the global one, so that the gating of
// the pre-launched passes (and the host's single look) is the same on every rank.
// boundary_local (ranks > 0): the state this rank's first chunk continues from — the end state
// of the nearest rank below that has requests, or the state before the first request of the
// batch when there is none. Thread per class.
__global__ __launch_bounds__(256) void k_global_flag(const ClassState* bounds, uint32_t rec,
                                                     uint32_t n_classes, uint32_t n_ranks,
                                                     uint32_t rank, uint32_t pass,
                                                     const uint32_t* meta, uint32_t stride,
                                                     const uint32_t* cls_begin,
                                                     ClassState* boundary_local, DeviceParams* prm) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) {
    uint32_t any = 0, over = 0;
    for (uint32_t g = 0; g < n_ranks; ++g) {
      any |= bounds[(size_t)g * rec + n_classes].cursor;
      over |= bounds[(size_t)g * rec + n_classes].lo;
    }
    prm->n_changed[pass & 63] = any ? 1u : 0u;
    if (over) prm->overflow = 1;
  }
  if (c >= n_classes || rank == 0) return;
  const int q = rank_predecessor(meta, stride, rank);
  ClassState st;
  if (q >= 0) {
    st = bounds[(size_t)q * rec + c];
  } else {
    st.cursor = st.lo = cls_begin[c];
    st.hown_lo = st.hown_hi = kNone;
  }
  boundary_local[c] = st;
}
// running_out[s] = running[s] + sum over ranks of delta[g][s].
// out_a / out_b (nullable): the caller's copy and, when committing, the resident column.
__global__ __launch_bounds__(256) void k_sum_deltas(const uint32_t* running, const uint32_t* deltas,
                                                    uint32_t n, uint32_t n_ranks, uint32_t* running_out,
                                                    uint32_t* out_a, uint32_t* out_b,
                                                    const DeviceParams* prm) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  if (prm->window_miss) return;  // the batch is repeated with the full sort: nothing to commit
  uint32_t acc = running[s];
  for (uint32_t g = 0; g < n_ranks; ++g) acc += deltas[(size_t)g * n + s];
  running_out[s] = acc;
  if (out_a) out_a[s] = acc;
  if (out_b) out_b[s] = acc;
}

// ---------------------------------------------------------------------------
// Sharded sort. Slot keys are a closed form of (servant, running) and strictly
// increasing per servant, so "how many slots of the registry sort below key K" is a sum of
// per-servant closed-form counts (first_slot_not_below). Every rank evaluates the same
// kWindowThresholds - 1 evenly spaced keys (k_key_count), picks the two that bracket the global
// ranks its own requests can reach — consuming requests of the ranks before it, its own, and a
// margin on both sides for classes that run ahead of or behind the global level — and
// generates, sorts and partitions only the slots between them (k_window; k_slot_gen with
// r_first). Class-list positions are then local to the window; neighbours exchange them as
// positions in the whole registry's lists (k_pack_boundary / k_boundary_in). A window that did
// not cover what the requests reached is detected (k_boundary_in) and the batch repeated with
// the full sort on every rank.
// ---------------------------------------------------------------------------
constexpr uint32_t kWindowThresholds = 128;

// cum[j] = slots with key < (j << shift), j = 1 .. kWindowThresholds - 1. One workgroup per j.
__global__ __launch_bounds__(256) void k_key_count(ServantTable sv, uint32_t cap_bits, uint32_t shift,
                                                   uint32_t* cum) {
  __shared__ uint32_t lds[17];
  const uint32_t j = blockIdx.x + 1;
  const uint64_t K = (uint64_t)j << shift;
  uint32_t cnt = 0;
  for (uint32_t s = threadIdx.x; s < sv.n; s += blockDim.x) {
    if (sv.class_of[s] == kNone) continue;
    const uint32_t run = sv.running[s];
    cnt += first_slot_not_below(sv.nproc[s], sv.load[s], sv.max_tasks[s], run, sv.flags[s], 0, K,
                                cap_bits) - run;
  }
  uint32_t total;
  (void)block_exclusive_scan(cnt, lds, &total);
  if (threadIdx.x == 0) cum[j] = total;
}

struct WindowArgs {
  const uint32_t* cum;        // k_key_count
  uint32_t cap_bits, shift;
  const uint32_t* totals;     // k_rank_meta of every rank (all-gathered), `stride` words each
  uint32_t stride;
  uint32_t rank, n_ranks, margin;
  uint32_t max_local;         // workspace / launch bound on the slots of the window
  const uint32_t* gslot_base; // [S + 1] prefix of the slot counts of the whole registry
  const uint32_t* cls_begin_glob;  // [C + 1] class lists of the whole registry
  uint32_t n_classes;
  // out
  uint32_t* r_first;          // [S] first `running` value of the servant inside the window
  uint32_t* lbase;            // [S + 1] local prefix of the slot counts
  uint32_t* cls_begin;        // [C + 1] local class lists
  uint32_t* shift_out;        // [C] registry-wide list position = local position + shift
  uint32_t* winrec;           // [2 C] registry-wide list positions [start, end) the window covers
};

// ONE workgroup (like k_servant_scan): thresholds, per-servant windows, local prefix, class sizes.
__global__ __launch_bounds__(1024) void k_window(ServantTable sv, WindowArgs a, DeviceParams* prm) {
  __shared__ uint32_t lds[17];
  __shared__ uint32_t carry;
  __shared__ uint64_t k_lo_s, k_hi_s;
  __shared__ uint32_t hi_is_end;
  extern __shared__ uint32_t cls_acc[];  // below[C] | len[C]
  const uint32_t C = a.n_classes;
  for (uint32_t c = threadIdx.x; c < 2 * C; c += blockDim.x) cls_acc[c] = 0;
  if (threadIdx.x == 0) {
    carry = 0;
    const uint32_t M = prm->n_slots;  // slots of the whole registry (k_servant_scan)
    uint32_t base = 0;
    for (uint32_t r = 0; r < a.rank; ++r) base += a.totals[(size_t)r * a.stride];
    const uint32_t n = a.totals[(size_t)a.rank * a.stride];
    const uint32_t want_lo = base > a.margin ? base - a.margin : 0u;
    const uint64_t want_hi64 = (uint64_t)base + n + a.margin;
    const uint32_t want_hi = want_hi64 > M ? M : (uint32_t)want_hi64;
    // Largest threshold with at most want_lo slots below it; smallest with at least want_hi.
    uint32_t jl = 0, jh = kWindowThresholds;
    for (uint32_t j = 1; j < kWindowThresholds; ++j) {
      if (a.cum[j] <= want_lo) jl = j;
    }
    for (uint32_t j = kWindowThresholds - 1; j >= 1; --j) {
      if (a.cum[j] >= want_hi) jh = j;
    }
    k_lo_s = (uint64_t)jl << a.shift;
    k_hi_s = (uint64_t)jh << a.shift;
    hi_is_end = jh == kWindowThresholds;
  }
  __syncthreads();
  const uint64_t k_lo = k_lo_s, k_hi = k_hi_s;
  const bool to_end = hi_is_end != 0;
  uint32_t below_total = 0;
  for (uint32_t s0 = 0; s0 < sv.n; s0 += blockDim.x) {
    const uint32_t s = s0 + threadIdx.x;
    uint32_t len = 0, below = 0, cls = kNone, r0 = 0, run = 0;
    if (s < sv.n) {
      cls = sv.class_of[s];
      run = sv.running[s];
      r0 = run;
      if (cls != kNone) {
        const uint32_t nproc = sv.nproc[s], load = sv.load[s], mt = sv.max_tasks[s], fl = sv.flags[s];
        const uint32_t top = run + servant_slot_count(nproc, load, mt, run, fl);
        r0 = k_lo ? first_slot_not_below(nproc, load, mt, run, fl, 0, k_lo, a.cap_bits) : run;
        const uint32_t r1 = to_end ? top : first_slot_not_below(nproc, load, mt, run, fl, 0, k_hi, a.cap_bits);
        below = r0 - run;
        len = r1 - r0;
      }
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(len, lds, &total);
    if (s < sv.n) {
      a.lbase[s] = carry + ex;
      a.r_first[s] = r0;
      if (cls != kNone) {
        if (below) atomicAdd(&cls_acc[cls], below);
        if (len) atomicAdd(&cls_acc[C + cls], len);
      }
    }
    below_total += below;
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
  uint32_t all_below;
  (void)block_exclusive_scan(below_total, lds, &all_below);
  if (threadIdx.x == 0) {
    const uint32_t m = carry;
    a.lbase[sv.n] = m;
    if (m > a.max_local) {
      prm->window_miss = 1;  // (the workspace bound: treated like any other miss)
      prm->n_slots = 0;
    } else {
      prm->n_slots = m;
    }
    prm->rank_offset = all_below;
    uint32_t acc = 0;
    for (uint32_t c = 0; c < C; ++c) {
      a.cls_begin[c] = acc;
      const uint32_t start = a.cls_begin_glob[c] + cls_acc[c];
      a.shift_out[c] = start - acc;
      a.winrec[2 * c] = start;
      a.winrec[2 * c + 1] = start + cls_acc[C + c];
      acc += cls_acc[C + c];
    }
    a.cls_begin[C] = acc;
  }
}

// After the all-gather of a pass's records (k_pack_boundary), sharded sort: the global "some
// rank changed an end state" flag (like k_global_flag), the predecessor's end state translated
// into this rank's local list positions, and the check that every rank's window covered what
// its requests reached — evaluated for ALL ranks from the gathered windows, so every rank
// comes to the same verdict: a predecessor's state outside the window, or a class consumed up
// to the window's end while the registry's list goes on. Thread per class.
__global__ __launch_bounds__(256) void k_boundary_in(const ClassState* bounds, uint32_t rec,
                                                     uint32_t n_classes, uint32_t n_ranks,
                                                     uint32_t rank, uint32_t pass,
                                                     const uint32_t* winall,  // [rank][2 C]
                                                     const uint32_t* cls_begin_glob,
                                                     const uint32_t* meta, uint32_t stride,
                                                     const uint32_t* shift,
                                                     ClassState* boundary_local, DeviceParams* prm) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) {
    uint32_t any = 0, over = 0, wmiss = 0;
    for (uint32_t g = 0; g < n_ranks; ++g) {
      any |= bounds[(size_t)g * rec + n_classes].cursor;
      over |= bounds[(size_t)g * rec + n_classes].lo;
      wmiss |= bounds[(size_t)g * rec + n_classes].hown_lo;
    }
    prm->n_changed[pass & 63] = any ? 1u : 0u;
    if (over) prm->overflow = 1;
    if (wmiss) prm->window_miss = 1;
  }
  if (c >= n_classes) return;
  // The state before rank g's first request: the end state of the nearest rank below that has
  // requests (ranks without requests are transparent), or the start of the class's list.
  auto start_of = [&](uint32_t g) {
    const int q = rank_predecessor(meta, stride, g);
    ClassState st;
    if (q >= 0) {
      st = bounds[(size_t)q * rec + c];
    } else {
      st.cursor = st.lo = cls_begin_glob[c];
      st.hown_lo = st.hown_hi = kNone;
    }
    return st;
  };
  // The verdict on the windows is taken on final states only (a pass in which no rank changed
  // anything): speculative replays may run off a window without any consequence.
  uint32_t changed = 0;
  for (uint32_t g = 0; g < n_ranks; ++g) changed |= bounds[(size_t)g * rec + n_classes].cursor;
  bool miss = false;
  for (uint32_t g = 0; g < n_ranks && !changed; ++g) {
    if (meta[(size_t)g * stride + stride - 1] == 0) continue;  // no requests: touches nothing
    const uint32_t w0 = winall[(size_t)g * 2 * n_classes + 2 * c];
    const uint32_t w1 = winall[(size_t)g * 2 * n_classes + 2 * c + 1];
    const ClassState st = start_of(g);
    miss |= st.lo < w0 || st.cursor > w1 || st.lo > st.cursor;
    const ClassState en = bounds[(size_t)g * rec + c];
    miss |= en.cursor >= w1 && w1 < cls_begin_glob[c + 1];
  }
  if (miss) prm->window_miss = 1;  // (benign race: everybody writes 1)
  if (rank > 0) {
    // Into this rank's window first, then to local positions: a start state outside the
    // window is a miss (flagged above, on the final pass) — but the state handed to the
    // matching passes must be one they can reproduce exactly (they clamp what they are given,
    // and a chunk whose recorded start differs from its given start is never consistent).
    ClassState st = start_of(rank);
    const uint32_t w0 = winall[(size_t)rank * 2 * n_classes + 2 * c];
    const uint32_t w1 = winall[(size_t)rank * 2 * n_classes + 2 * c + 1];
    const uint32_t cur = min(max(st.cursor, w0), w1);
    const uint32_t lo = min(max(st.lo, w0), cur);
    if (lo == cur) st.hown_lo = st.hown_hi = kNone;  // (what a state without holes reports)
    st.cursor = cur - shift[c];
    st.lo = lo - shift[c];
    boundary_local[c] = st;
  }
}

// ---------------------------------------------------------------------------
// Inter-process all-gather without RCCL (ydc_group_init_ipc): every rank owns a MAILBOX — device
// memory opened by its peers through HIP IPC handles, or a shared host segment mapped by all of
// them — with one slot per (parity, sender). A sender writes its words into its slot of every
// peer's mailbox as 8-byte granules {word, exchange stamp} (system-scope atomic stores: every
// granule carries its own validity, so there is no flag, no fence and no ordering between
// granules — the hand-off pattern of match_kernel.h, across processes), a receiver polls the
// granules of its own mailbox until they carry the stamp of this exchange. Exchanges alternate
// between two sets of slots: a rank can only be two exchanges ahead of a peer that has not yet
// read the older one (it needs that peer's data of the exchange in between), so a slot is never
// overwritten before it was read. Bounded: a peer that does not deliver within `timeout_ticks`
// of the 100 MHz wall clock flags DeviceParams::exchange_timeout (and the words read as 0).
// Workgroup b serves peer b / blocks_per_peer: writes this rank's words to that peer, reads that
// peer's words from this rank's mailbox into recv (rank-major, like ncclAllGather; recv_stride
// words per rank: a long exchange is cut into pieces of at most slot_words words).
// ---------------------------------------------------------------------------
struct MailboxPeers {
  unsigned long long* box[16];  // mailbox of rank q as mapped into THIS process (own: box[rank])
};
constexpr uint32_t kMailboxMaxRanks = 16;

__global__ __launch_bounds__(256) void k_mailbox_all_gather(MailboxPeers peers, uint32_t rank,
                                                            uint32_t n_ranks, const uint32_t* send,
                                                            uint32_t* recv, uint32_t n_words,
                                                            uint32_t recv_stride,
                                                            uint32_t slot_words, uint32_t parity,
                                                            uint32_t stamp, uint32_t blocks_per_peer,
                                                            unsigned long long timeout_ticks,
                                                            DeviceParams* prm) {
  const uint32_t q = blockIdx.x / blocks_per_peer, part = blockIdx.x % blocks_per_peer;
  if (q >= n_ranks) return;
  const uint32_t stride = blocks_per_peer * blockDim.x;
  const uint32_t first = part * blockDim.x + threadIdx.x;
  if (q == rank) {  // own words: straight across
    for (uint32_t w = first; w < n_words; w += stride) recv[(size_t)rank * recv_stride + w] = send[w];
    return;
  }
  // Slot of sender r in a mailbox: [(parity * n_ranks + r) * slot_words, + slot_words).
  unsigned long long* out = peers.box[q] + ((size_t)parity * n_ranks + rank) * slot_words;
  const unsigned long long tag = (unsigned long long)stamp << 32;
  for (uint32_t w = first; w < n_words; w += stride)
    __hip_atomic_store(out + w, tag | send[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  unsigned long long* in = peers.box[rank] + ((size_t)parity * n_ranks + q) * slot_words;
  const unsigned long long t0 = wall_clock64();
  bool late = false;
  for (uint32_t w = first; w < n_words; w += stride) {
    unsigned long long v = __hip_atomic_load(in + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    while ((uint32_t)(v >> 32) != stamp && !late) {
      __builtin_amdgcn_s_sleep(8);
      v = __hip_atomic_load(in + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((uint32_t)(v >> 32) != stamp && wall_clock64() - t0 > timeout_ticks) late = true;
    }
    recv[(size_t)q * recv_stride + w] = (uint32_t)(v >> 32) == stamp ? (uint32_t)v : 0u;
  }
  if (late) prm->exchange_timeout = 1;
}

// Heartbeats of known servants (KeepServantAlive replaces the personality and keeps
// running_tasks, task_dispatcher.cc:195-201): thread per update; idx >= n_servants is padding.
struct ServantRowDev {  // == ydc_servant_row (include/yadcc_dispatch.h)
  uint32_t version, num_processors, current_load, max_tasks, flags, ip_id;
  uint64_t env_mask;
};
__global__ __launch_bounds__(256) void k_apply_rows(const uint32_t* idx, const ServantRowDev* rows,
                                                    uint32_t n, uint32_t n_servants,
                                                    uint32_t* version, uint32_t* nproc,
                                                    uint32_t* load, uint32_t* max_tasks,
                                                    uint32_t* flags) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = idx[i];
  if (s >= n_servants) return;
  const ServantRowDev r = rows[i];
  version[s] = r.version;
  nproc[s] = r.num_processors;
  load[s] = r.current_load;
  max_tasks[s] = r.max_tasks;
  flags[s] = r.flags;
}

// One streaming tick's registry deltas in one launch: workgroups [0, upd_blocks) apply the
// heartbeat rows (as k_apply_rows), the rest give released slots back (as k_release_slots).
__global__ __launch_bounds__(256) void k_apply_tick(const uint32_t* idx, const ServantRowDev* rows,
                                                    uint32_t n_upd, uint32_t upd_blocks,
                                                    const uint32_t* released, uint32_t n_rel,
                                                    uint32_t n_servants, uint32_t* version,
                                                    uint32_t* nproc, uint32_t* load,
                                                    uint32_t* max_tasks, uint32_t* flags,
                                                    uint32_t* running) {
  if (blockIdx.x < upd_blocks) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_upd) return;
    const uint32_t s = idx[i];
    if (s >= n_servants) return;
    const ServantRowDev r = rows[i];
    version[s] = r.version;
    nproc[s] = r.num_processors;
    load[s] = r.current_load;
    max_tasks[s] = r.max_tasks;
    flags[s] = r.flags;
  } else {
    const uint32_t i = (blockIdx.x - upd_blocks) * blockDim.x + threadIdx.x;
    if (i < n_rel && released[i] < n_servants) atomicSub(&running[released[i]], 1u);
  }
}

}  // namespace ydc

#include "bin_sort.h"
#include "match_kernel.h"
#include "wide_kernel.h"

#endif  // YADCC_AMD_KERNELS_H_

// zone_guess.h — k_zone_guess: start guesses for the chunks around the dedicated tier's end.
//
// The level guesses of the matching passes (match_kernel.h) assume that the n lowest-ranked
// slots are gone after n consuming requests. That holds inside a tier — every class's head
// stays near one utilisation level, the guesses are within a few list positions — and fails for
// a couple of thousand requests where the dedicated tier (task_dispatcher.cc:404-407: the
// slots of dedicated servants below half their cores sort before everything else) runs out:
// the classes' tier-0 parts end at different ranks, and requests whose classes are exhausted go
// on to tier 1 while a class that few requests can use still holds tier-0 slots. The chunks of
// that stretch start from wrong states, and the passes behind the first one repair them one
// after the other — one wave following ~2000 requests, more than half of cfg3's matching time
// (DESIGN.md §9.3).
//
// One wave walks that stretch BEFORE the passes, with nothing but the merge itself: a lane per
// class, the head rank of the class's list in a register (the lists' next entries in an LDS
// window), a request takes the lowest head among its eligible classes. No slots, no results, no
// own-servant rule (a request from a servant's host that meets its own servant at a head picks
// differently once in thousands of picks; the guesses need to be close, not exact — every start
// state is verified by the passes as before). Started a thousand requests before the tier's
// end from the level guess of that point, it is on the true track within a chunk (numpy
// restatement on cfg3: no deviation at any chunk boundary of the stretch, where the level
// guesses are off by up to 149 positions) and leaves the class cursors at the points where the
// stretch's chunks start their replays.
#ifndef YADCC_AMD_ZONE_GUESS_H_
#define YADCC_AMD_ZONE_GUESS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dispatch_core.h"

namespace ydc {

constexpr uint32_t kZoneMaxChunks = 32;  // chunks a zone may span (guess table rows)
constexpr uint32_t kZoneWindow = 256;    // list entries per class in LDS (refilled when used up)

struct ZoneArgs {
  ClassLists L;
  const uint64_t* mask;  // one word per request (<= 64 classes)
  uint32_t n_tasks, chunk_size, n_chunks;
  const uint32_t* before;  // [n_chunks + 1] consuming requests before every chunk (one part)
  const uint32_t* tail;    // warm-up mode: consuming requests among the last warm_len of every chunk
  uint32_t warm_len;       // 0: replays start at their chunk's first request
  const uint32_t* tile_tab;  // nullable: MatchBuffers::tile_tab (narrows the start searches)
  uint32_t tile_tab_tiles, tile_tab_elems;
  const uint2* sorted;   // the records in key order: {key, value}
  uint32_t tier_shift;   // key >> tier_shift != 0: tier 1
  uint32_t lead, trail;  // the zone: levels [T0 - lead, T0 + trail)
  uint32_t* hdr;         // [2] out: chunks [hdr[0], hdr[1]) have a row in `guess`
  ClassState* guess;     // [kZoneMaxChunks * n_classes]
};

__global__ __launch_bounds__(64) void k_zone_guess(ZoneArgs a, const DeviceParams* prm) {
  extern __shared__ uint32_t zwin[];  // [C][kZoneWindow] ranks of the classes' next entries
#ifdef YDC_PHASE_PROBE
#define YDC_ZSTAMP(i) do { if (threadIdx.x == 0) ydc_phase_probe[59990 + (i)] = wall_clock64(); } while (0)
#else
#define YDC_ZSTAMP(i) do { } while (0)
#endif
  YDC_ZSTAMP(0);
  const uint32_t lane = threadIdx.x, C = a.L.n_classes, K = a.n_chunks, cs = a.chunk_size;
  if (lane < 2) a.hdr[lane] = 0;  // (no zone unless the end of this kernel says so)
  const uint32_t M = prm->n_slots;
  if (M == 0 || K < 2 || a.L.list_p == nullptr) return;
  // ---- T0: the first slot of tier 1 in the global order (64 probes per round)
  uint32_t lo = 0, hi = M;  // T0 in [lo, hi]; hi == M or the slot at hi is tier 1
  while (lo < hi) {
    const uint32_t step = (hi - lo + 63) / 64;
    const bool in = lo + lane * step < hi;
    const uint32_t p = min(lo + (lane + 1) * step - 1, hi - 1);
    const bool t0 = in && (a.sorted[p].x >> a.tier_shift) == 0;
    const uint64_t act = __ballot(in), zero = __ballot(t0);
    // (tier 0 is a prefix of the order: the segments whose last slot is tier 0 are the first k)
    const uint32_t k = (uint32_t)__builtin_ctzll(~zero | (1ull << 63));
    if (zero == act) {  // every probe is tier 0: what is left of the range is, too
      lo = hi;
      break;
    }
    const uint32_t nlo = lo + k * step;
    hi = min(nlo + step - 1, hi - 1);  // (that probe is tier 1)
    lo = nlo;
  }
  const uint32_t T0 = lo;
  YDC_ZSTAMP(1);
  if (T0 == 0 || T0 >= M) return;  // one tier only
  // ---- the zone's first chunk: the first whose level is within `lead` of T0
  const uint32_t from = T0 > a.lead ? T0 - a.lead : 0u;
  uint32_t klo = 1, khi = K;  // first k in [1, K) with before[k] >= from, or K
  while (klo < khi) {
    const uint32_t step = (khi - klo + 63) / 64;
    const bool in = klo + lane * step < khi;
    const uint32_t p = min(klo + (lane + 1) * step - 1, khi - 1);
    const bool below = in && a.before[p] < from;
    const uint64_t act = __ballot(in), b = __ballot(below);
    const uint32_t k = (uint32_t)__builtin_ctzll(~b | (1ull << 63));
    if (b == act) {
      klo = khi;
      break;
    }
    const uint32_t nlo = klo + k * step;
    khi = min(nlo + step - 1, khi - 1);
    klo = nlo;
  }
  const uint32_t z_lo = klo;
  YDC_ZSTAMP(2);
  if (z_lo >= K) return;  // the batch ends before the tier does
  if (a.before[z_lo] >= T0 + a.trail) return;  // ... or starts behind it
  // ---- start: the level guess of the point where chunk z_lo's replay starts
  const uint32_t warm = a.warm_len && a.tail ? a.warm_len : 0u;
  const uint32_t t_start = z_lo * cs - warm;
  uint32_t level = a.before[z_lo] - (warm ? a.tail[z_lo - 1] : 0u);
  uint32_t x = 0, end = 0;  // this lane's class: cursor (list position), end of its list
  if (lane < C) {
    const uint32_t b = a.L.cls_begin[lane];
    end = a.L.cls_begin[lane + 1];
    uint32_t slo = b, shi = end;
    if (a.tile_tab) {
      const uint32_t nt = a.tile_tab_tiles, t = min(level / a.tile_tab_elems, nt - 1);
      const uint32_t* row = a.tile_tab + (size_t)lane * nt;
      slo = b + row[t];
      shi = t + 1 < nt ? b + row[t + 1] : end;
    }
    while (slo < shi) {  // first entry with rank >= level
      const uint32_t mid = (slo + shi) >> 1;
      if (list_rank(a.L, mid) < level) slo = mid + 1; else shi = mid;
    }
    x = slo;
  }
  YDC_ZSTAMP(3);
  // ---- rings: the layout match_fast_loop walks (match_kernel.h: MatchWave) — class c's list
  // entry p sits at word (c << 8) + (p & 255), as ~rank (0: no slot, also beyond the list's
  // end); a ring holds the entries [cursor, filled). Filled to the brim at the start, eight
  // classes at a time (their loads in flight together), and topped up before a block of 64
  // requests for every class that has fewer than 66 entries left (it can win 64 times).
  constexpr uint32_t kShift = 8, kRing = 1u << kShift, kMask = kRing - 1;
  static_assert(kRing == kZoneWindow, "ring size");
  if ((uint32_t)(uintptr_t)zwin != 0) __builtin_trap();  // (the loop's address stepping: rings aligned to their size)
  uint32_t filled = x;
  auto fill = [&](uint64_t which, uint32_t rounds) {  // (whole wave) `rounds` x 64 entries behind `filled`
    while (which) {
      uint32_t cl[8], v[8][4], f0[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        cl[q] = 64;
        if (which) {
          cl[q] = (uint32_t)__builtin_ctzll(which);
          which &= which - 1;
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        f0[q] = 0;
        if (cl[q] < 64) {
          f0[q] = (uint32_t)__builtin_amdgcn_readlane((int)filled, cl[q]);
          const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)end, cl[q]);
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t i = f0[q] + j * 64 + lane;
            v[q][j] = j < rounds && i < e ? ~list_rank(a.L, i) : 0u;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (cl[q] < 64) {
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j)
            if (j < rounds) zwin[(cl[q] << kShift) + ((f0[q] + j * 64 + lane) & kMask)] = v[q][j];
          if (lane == cl[q]) filled += rounds * 64;
        }
      }
    }
  };
  const uint64_t all_classes = C >= 64 ? ~0ull : (1ull << C) - 1;
  fill(all_classes, 2);  // (128 entries: what most classes lose in the whole stretch)
  __builtin_amdgcn_wave_barrier();
  const uint32_t base = (lane << kShift) << 2;  // LDS byte address of this lane's ring
  uint32_t hq = 0, nq = 0;  // ~rank of the entries at cursor / cursor + 1
  if (lane < C) {
    hq = zwin[(lane << kShift) + (x & kMask)];
    nq = zwin[(lane << kShift) + ((x + 1) & kMask)];
  }
  uint32_t steps = 1;  // DPP steps of the maximum over the class lanes
  while ((1u << steps) < C) ++steps;
  const uint32_t pair = steps >= 3 ? 1u : 0u;
  YDC_ZSTAMP(4);
  // ---- the walk: the matching loop itself (match_fast_loop: hand-scheduled, two requests per
  // iteration), a block of 64 requests per call, nobody "special" (no own-servant rule, no holes)
  uint32_t k_next = z_lo;  // next chunk whose start state is to be recorded
  const uint32_t k_cap = min(K, z_lo + kZoneMaxChunks);
  uint32_t z_hi = z_lo;
  const uint32_t t_end = min(a.n_tasks, k_cap * cs);
  uint32_t rec_at = t_start;  // (request index of the next record point: t_start + j * cs, block-aligned)
  // (a block's class masks are fetched a block ahead)
  uint64_t m_next = t_start + lane < t_end ? a.mask[t_start + lane] : 0ull;
  for (uint32_t tb = t_start; tb < t_end; tb += 64) {
    if (tb == rec_at) {
      // chunk k_next's replay starts here
      if (lane < C) a.guess[(size_t)(k_next - z_lo) * C + lane] = ClassState{x, x, kNone, kNone};
#ifdef YDC_PHASE_PROBE
      if (lane < C) ydc_phase_probe[60008 + (size_t)(k_next - z_lo) * 64 + lane] = x - a.L.cls_begin[lane];
#endif
      const uint32_t lvl = a.before[k_next] - (warm ? a.tail[k_next - 1] : 0u);
      z_hi = ++k_next;
      rec_at += cs;
      if (lvl >= T0 + a.trail || k_next >= k_cap) break;
    }
    {
      // (beyond the list's end the ring is padded with "no slot": the loop looks two entries ahead)
      const uint64_t low = __ballot(lane < C && filled - x < 66 && filled < end + 66);
      if (low) {
        __builtin_amdgcn_wave_barrier();
        fill(low, 2);  // (filled - cursor < 66: 128 more stay within the ring's 256)
        __builtin_amdgcn_wave_barrier();
      }
    }
    const uint64_t m = m_next;
    m_next = tb + 64 + lane < t_end ? a.mask[tb + 64 + lane] : 0ull;
    const uint32_t cnt = min(64u, t_end - tb);
    const BlockMasks bm = block_masks((uint32_t)m, (uint32_t)(m >> 32), steps > 5);
    uint32_t raw = 0, left = 0x7FFFFFFFu, i = 0;
    const uint32_t an0 = base + (((x + 1) & kMask) << 2);  // address of `next`
    uint32_t an = an0;
    while (i < cnt) {
      const uint32_t st = match_fast_loop<false>(i, cnt, bm, kNone, kNone, 0ull, 0ull, 0ull, raw, hq, nq, an,
                                                 (64u << kShift) << 2, (kRing << 2) - 1, steps, pair, left);
      if (st != 0) ++i;  // (cannot happen: nobody is special)
    }
    x += ((an - an0) & ((kRing << 2) - 1)) >> 2;  // this class's wins in the block
  }
  YDC_ZSTAMP(5);
  if (z_hi > z_lo && lane == 0) {
    a.hdr[0] = z_lo;
    a.hdr[1] = z_hi;
  }
#ifdef YDC_PHASE_PROBE
  if (lane == 0) {
    ydc_phase_probe[60000] = T0;
    ydc_phase_probe[60001] = z_lo;
    ydc_phase_probe[60002] = z_hi;
    ydc_phase_probe[60003] = t_start;
  }
#endif
}

}  // namespace ydc
#endif  // YADCC_AMD_ZONE_GUESS_H_

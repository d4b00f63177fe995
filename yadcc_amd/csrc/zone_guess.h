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
// One wave walks that stretch BESIDE the first pass — workgroup 0 of the launch of pass 0, the
// chunks' waves are the workgroups behind it — with nothing but the merge itself: a lane per class, the head rank of the class's list in a register (the
// lists' next entries in an LDS window), a request takes the lowest head among its eligible
// classes. No slots, no results, no own-servant rule (a request from a servant's host that meets
// its own servant at a head picks differently once in thousands of picks; the guesses need to be
// close, not exact — every start state is verified by the passes as before). Started a couple of
// chunks before the tier's end from the level guess of that point, it is on the true track within
// a chunk (numpy restatement on cfg3: no deviation at any chunk boundary of the stretch, where
// the level guesses are off by up to 149 positions) and publishes the class cursors at the points
// where the stretch's chunks start their replays, as 8-byte granules {cursor, batch number}: a
// chunk of the stretch (the header granules say which chunks those are) waits for its row
// instead of starting from its level guess — the sequential part of the batch runs while the
// other 1900 chunks replay, not behind them (walked in front of the passes it cost what it saved:
// DESIGN.md §9.3).
#ifndef YADCC_AMD_ZONE_GUESS_H_
#define YADCC_AMD_ZONE_GUESS_H_
// (included by match_kernel.h between match_fast_loop and k_match_pass)

namespace ydc {

constexpr uint32_t kZoneMaxChunks = 32;  // chunks a zone may span (rows of cursors)

__device__ __forceinline__ void zone_put(unsigned long long* g, uint32_t word, uint32_t seq) {
  __hip_atomic_store(g, ((unsigned long long)seq << 32) | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// zwin: the wave's LDS (at address 0, 2 * ring_total words): class c's list entry p sits at word
// (c << zshift) + (p & (2^zshift - 1)) as ~rank — the layout match_fast_loop walks (MatchWave's
// ring_p), 2^zshift = the largest power of two <= 2 * ring_total / C, at most 256.
__device__ __forceinline__ void zone_walk(const ClassLists& L, const uint64_t* mask, uint32_t n_tasks,
                                          uint32_t cs, uint32_t K, const MatchBuffers& B,
                                          DeviceParams* prm, uint32_t* zwin, uint32_t ring_total) {
#ifdef YDC_PHASE_PROBE
#define YDC_ZSTAMP(i) do { if (threadIdx.x == 0) ydc_phase_probe[59990 + (i)] = wall_clock64(); } while (0)
#else
#define YDC_ZSTAMP(i) do { } while (0)
#endif
  YDC_ZSTAMP(0);
  __builtin_amdgcn_s_setprio(3);  // (the batch waits for this wave: first on its SIMD)
  const uint32_t lane = threadIdx.x, C = L.n_classes;
  const uint32_t seq = prm->batch_seq;
  unsigned long long* box = B.zone_box;
  auto no_zone = [&]() {  // (the chunks look at the header: tell them not to wait)
    if (lane < 2) zone_put(box + lane, 0u, seq);
  };
  const uint32_t M = prm->n_slots;
  if (M == 0 || K < 2 || L.list_p == nullptr || B.before == nullptr || C == 0) return no_zone();
  // ---- T0: the first slot of tier 1 in the global order (64 probes per round)
  uint32_t lo = 0, hi = M;  // T0 in [lo, hi]; hi == M or the slot at hi is tier 1
  while (lo < hi) {
    const uint32_t step = (hi - lo + 63) / 64;
    const bool in = lo + lane * step < hi;
    const uint32_t p = min(lo + (lane + 1) * step - 1, hi - 1);
    const bool t0 = in && (B.zone_sorted[p].x >> B.zone_tier_shift) == 0;
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
  if (T0 == 0 || T0 >= M) return no_zone();  // one tier only
  // ---- the zone's first chunk: the first whose level is within `lead` of T0
  const uint32_t from = T0 > B.zone_lead ? T0 - B.zone_lead : 0u;
  uint32_t klo = 1, khi = K;  // first k in [1, K) with before[k] >= from, or K
  while (klo < khi) {
    const uint32_t step = (khi - klo + 63) / 64;
    const bool in = klo + lane * step < khi;
    const uint32_t p = min(klo + (lane + 1) * step - 1, khi - 1);
    const bool below = in && B.before[p] < from;
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
  if (z_lo >= K) return no_zone();  // the batch ends before the tier does
  if (B.before[z_lo] >= T0 + B.zone_trail) return no_zone();  // ... or starts behind it
  const uint32_t warm = B.warm_len && B.tail ? B.warm_len : 0u;
  // ---- ... and its end: the first chunk whose replay starts `trail` levels behind T0 (or the cap)
  const uint32_t k_cap = min(K, z_lo + kZoneMaxChunks);
  uint32_t z_hi = z_lo + 1;
  {
    const uint32_t k = z_lo + 1 + lane;  // (kZoneMaxChunks <= 64: one probe per lane)
    const bool in = k < k_cap && B.before[k] - (warm ? B.tail[k - 1] : 0u) < T0 + B.zone_trail;
    // (levels ascend: the chunks still inside are a prefix)
    z_hi = z_lo + 1 + (uint32_t)__builtin_ctzll(~__ballot(in));
  }
  if (lane == 0) {
    zone_put(box + 0, z_lo, seq);
    zone_put(box + 1, z_hi, seq);
    prm->zone_rows = z_hi - z_lo;  // (the host: did the chunks that were served come out consistent?)
  }
  YDC_ZSTAMP(2);
  if (z_hi - z_lo < 2) return;  // (row 0 is the chunk's own level guess: nobody waits for anything)
  // ---- start: the level guess of the point where chunk z_lo's replay starts
  const uint32_t t_start = z_lo * cs - warm;
  const uint32_t level = B.before[z_lo] - (warm ? B.tail[z_lo - 1] : 0u);
  uint32_t x = 0, end = 0;  // this lane's class: cursor (list position), end of its list
  if (lane < C) {
    const uint32_t b = L.cls_begin[lane];
    end = L.cls_begin[lane + 1];
    uint32_t slo = b, shi = end;
    if (B.tile_tab) {
      const uint32_t nt = B.tile_tab_tiles, t = min(level / B.tile_tab_elems, nt - 1);
      const uint32_t* row = B.tile_tab + (size_t)lane * nt;
      slo = b + row[t];
      shi = t + 1 < nt ? b + row[t + 1] : end;
    }
    while (slo < shi) {  // first entry with rank >= level
      const uint32_t mid = (slo + shi) >> 1;
      if (list_rank(L, mid) < level) slo = mid + 1; else shi = mid;
    }
    x = slo;
  }
  YDC_ZSTAMP(3);
  // ---- rings: a ring holds the entries [cursor, filled), ~rank each (0: no slot, also beyond the
  // list's end). Filled to the brim at the start, eight classes at a time (their loads in flight
  // together), and topped up to the brim before a block of 64 requests for every class that has
  // fewer than 66 entries left (it can win 64 times, and the loop looks two entries ahead).
  uint32_t zshift = 8;
  while (zshift > 7 && ((size_t)C << zshift) > 2 * (size_t)ring_total) --zshift;
  const uint32_t ring = 1u << zshift, zmask = ring - 1;
  if ((uint32_t)(uintptr_t)zwin != 0) __builtin_trap();  // (the loop's address stepping: rings aligned to their size)
  uint32_t filled = x;
  auto fill = [&](uint64_t which) {  // (whole wave) the rings of `which` up to cursor + ring
    while (which) {
      uint32_t cl[8], v[8][4], f0[8], f1[8];
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
        f0[q] = f1[q] = 0;
        if (cl[q] < 64) {
          f0[q] = (uint32_t)__builtin_amdgcn_readlane((int)filled, cl[q]);
          f1[q] = (uint32_t)__builtin_amdgcn_readlane((int)x, cl[q]) + ring;
          const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)end, cl[q]);
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t i = f0[q] + j * 64 + lane;
            v[q][j] = i < f1[q] && i < e ? ~list_rank(L, i) : 0u;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (cl[q] < 64) {
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t i = f0[q] + j * 64 + lane;
            if (i < f1[q]) zwin[(cl[q] << zshift) + (i & zmask)] = v[q][j];
          }
          if (lane == cl[q]) filled = f1[q];
        }
      }
    }
  };
  const uint64_t all_classes = C >= 64 ? ~0ull : (1ull << C) - 1;
  fill(all_classes);
  __builtin_amdgcn_wave_barrier();
  const uint32_t base = (lane << zshift) << 2;  // LDS byte address of this lane's ring
  uint32_t hq = 0, nq = 0;  // ~rank of the entries at cursor / cursor + 1
  if (lane < C) {
    hq = zwin[(lane << zshift) + (x & zmask)];
    nq = zwin[(lane << zshift) + ((x + 1) & zmask)];
  }
  uint32_t steps = 1;  // DPP steps of the maximum over the class lanes
  while ((1u << steps) < C) ++steps;
  const uint32_t pair = steps >= 3 ? 1u : 0u;
  YDC_ZSTAMP(4);
  // ---- the walk: the matching loop itself (match_fast_loop: hand-scheduled, two requests per
  // iteration), a block of 64 requests per call, nobody "special" (no own-servant rule, no holes)
  uint32_t k_next = z_lo;  // next chunk whose start state is to be recorded
  const uint32_t t_end = min(n_tasks, z_hi * cs);
  uint32_t rec_at = t_start;  // (request index of the next record point: t_start + j * cs, block-aligned)
  // (a block's class masks are fetched a block ahead)
  uint64_t m_next = t_start + lane < t_end ? mask[t_start + lane] : 0ull;
  for (uint32_t tb = t_start; tb < t_end; tb += 64) {
    if (tb == rec_at) {
      // chunk k_next's replay starts here
      if (lane < C) zone_put(box + 2 + (size_t)(k_next - z_lo) * C + lane, x, seq);
#ifdef YDC_PHASE_PROBE
      if (lane < C) ydc_phase_probe[60008 + (size_t)(k_next - z_lo) * 64 + lane] = x - L.cls_begin[lane];
      if (lane == 0) ydc_phase_probe[59900 + (k_next - z_lo)] = wall_clock64();
#endif
      ++k_next;
      rec_at += cs;
      if (k_next >= z_hi) break;
    }
    {
      const uint64_t low = __ballot(lane < C && filled - x < 66 && filled < end + 66);
      if (low) {
        __builtin_amdgcn_wave_barrier();
        fill(low);
        __builtin_amdgcn_wave_barrier();
      }
    }
    const uint64_t m = m_next;
    m_next = tb + 64 + lane < t_end ? mask[tb + 64 + lane] : 0ull;
    const uint32_t cnt = min(64u, t_end - tb);
    const BlockMasks bm = block_masks((uint32_t)m, (uint32_t)(m >> 32), steps > 5);
    uint32_t raw = 0, left = 0x7FFFFFFFu, i = 0;
    const uint32_t an0 = base + (((x + 1) & zmask) << 2);  // address of `next`
    uint32_t an = an0;
    while (i < cnt) {
      const uint32_t st = match_fast_loop<false>(i, cnt, bm, kNone, kNone, 0ull, 0ull, 0ull, raw, hq, nq, an,
                                                 (64u << zshift) << 2, (ring << 2) - 1, steps, pair, left);
      if (st != 0) ++i;  // (cannot happen: nobody is special)
    }
    x += ((an - an0) & ((ring << 2) - 1)) >> 2;  // this class's wins in the block
  }
  YDC_ZSTAMP(5);
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

// wide_kernel.h — k_sim_wide: the chunk replay for registries with MANY servant classes
// (more than kMaxWaveClasses = 256 distinct (environment set, version) signatures — a pool whose
// machines advertise individual compiler sets has about one class per servant; the reference
// has no limit, task_dispatcher.h:93-94, .cc:316-344).
//
// Same protocol as the thread-per-chunk k_sim_generic it replaces up to kMaxWideClasses (rounds
// of replay + k_update, checked by the host; results identical): one WAVE per chunk, lane l owns
// the classes l, l + 64, l + 128, ... and the whole class state lives in LDS as arrays
// (structure-of-arrays: a lane's reads of one field for consecutive classes hit consecutive
// banks). A request scans its eligible classes' head ranks — one LDS read per eligible class and
// lane, the request's class mask words are wave-uniform scalars —, the wave takes the minimum,
// and the winning lane advances its class from LDS (head <- next) while the list entry after
// next is fetched from memory behind the following requests (one pending fetch per lane, stored
// the next time that lane wins). Requests from a servant's own host and classes with holes take
// the general step, which runs the shared state machine of dispatch_core.h on the LDS state.
#ifndef YADCC_AMD_WIDE_KERNEL_H_
#define YADCC_AMD_WIDE_KERNEL_H_

#include "kernels.h"

namespace ydc {

constexpr uint32_t kMaxWideClasses = 3072;  // 36 B of class state + 8 B of staged mask words per class
constexpr uint32_t kWideFields = 9;
// The walk with prefetch waves keeps nine more words per class (a four-entry ring of upcoming list
// entries + how far it is filled): 72 B + 8 B of mask words per class.
constexpr uint32_t kMaxWalkPrefetchClasses = 1920;
constexpr uint32_t kWalkRing = 4;
constexpr size_t kGroupWalkMaxLds = 160 * 1024 - 512;  // k_walk_groups: one workgroup, a CU's LDS
// Dynamic LDS of k_sim_wide for C classes: the nine state arrays + the class masks of a block of
// 64 requests (64 x ceil(C / 64) 64-bit words).
// Seven state arrays of C words, the two the scan reads (head rank, head slot) padded to a
// multiple of 512 classes so that the scan needs no bounds tests; the class masks of a block of 64
// requests (rows of W8 = ceil(W / 8) * 8 words) and the "has holes" / "one servant" bit rows.
__host__ __device__ inline size_t wide_lds_bytes(uint32_t C, bool walk_prefetch = false) {
  const size_t W = (C + 63) / 64, W8 = (W + 7) & ~(size_t)7;
  return (size_t)(kWideFields - 2 + (walk_prefetch ? 1 + 2 * kWalkRing : 0)) * C * 4 + 2 * W8 * 64 * 4 + 16 +
         (64 * W8 + 2 * W) * 8;
}

// LDS accesses of the walker / prefetcher protocol, as DS instructions on the LDS offset: a
// `volatile` generic pointer makes the compiler emit FLAT loads, which count in vmcnt — every
// protocol read then waits for the wave's outstanding global stores (measured: no faster than
// fetching from memory). The "memory" clobber keeps the compiler from moving accesses across.
__device__ __forceinline__ uint32_t lds_load_u32(const uint32_t* p) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
  return v;
}
__device__ __forceinline__ void lds_store_u32(uint32_t* p, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" : : "v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
}

// Eligible-class lists (host_tables.h: elig_off / elig_cls, every row at most 64 classes) and the
// row every request falls into. row_of == NULL: the kernel scans the requests' class mask words.
struct WideLists {
  const uint32_t* row_of;
  const uint32_t* off;
  const uint32_t* cls;
};

struct WideState {
  uint32_t *cur, *lo, *hlo, *hhi, *end, *hp, *hg, *np, *ng;
  const uint64_t* singlew;  // LDS: bit c & 63 of word c / 64 = ClassLists::cls_single[c]
  __device__ __forceinline__ ClassRun run(const ClassLists& L, uint32_t c) const {
    ClassRun r;
    r.cursor = cur[c];
    r.lo = lo[c];
    r.hown_lo = hlo[c];
    r.hown_hi = hhi[c];
    r.end = end[c];
    r.head_p = hp[c];
    r.head_g = hg[c];
    r.single = (uint32_t)((singlew[c >> 6] >> (c & 63u)) & 1u);  // (not a memory access per class)
    return r;
  }
};

// The general step of one request on the LDS state — the shared state machine of dispatch_core.h:
// a request from a servant's own host whose servant shows at the head of an eligible class, a
// request one of whose classes has holes, a host that runs several servants. The whole wave works
// on the ONE request t (lane l looks at the classes l, l + 64, ... of its mask words); holew's bits
// follow the classes' holes.
// row_c: kNone - 1 = "no row": the lanes read the request's classes off its mask words (lane l the
// classes l, l + 64, ...); else this lane's class of the request's row of eligible classes (kNone:
// none; rows are ascending and hold a class once) — one class per lane, no loop over the words.
constexpr uint32_t kNoRow = 0xFFFFFFFEu;
// Head rank above the class id (k_walk_groups, ranks below 2^(32 - cbits) - 1): the lowest-ranked
// head of a row of classes and the class it belongs to are one unsigned minimum.
__device__ __forceinline__ uint32_t pack_head(uint32_t rank, uint32_t c, uint32_t cbits) {
  return rank == kNone ? kNone : (rank << cbits) | c;
}
__device__ __forceinline__ void wide_general_step(const ClassLists& L, const WideState& S, uint64_t* holew,
                                                  uint32_t W, const uint64_t* mask, uint32_t self_lo,
                                                  uint32_t self_hi, const SharedIpTable& shared,
                                                  uint32_t* __restrict__ slot_of, uint32_t t, uint32_t lane,
                                                  uint32_t row_c = kNoRow, uint32_t* hpk = nullptr,
                                                  uint32_t cbits = 0, uint32_t hpk_stride = 1) {
  const bool by_row = row_c != kNoRow;
  if (by_row) {
    if (__ballot(row_c != kNone) == 0) {
      if (lane == 0) slot_of[t] = kIdxEnvNotFound;
      return;
    }
  } else {
    uint64_t many = 0;
    for (uint32_t w = 0; w < W; ++w) many |= mask[w];
    if (many == 0) {
      if (lane == 0) slot_of[t] = kIdxEnvNotFound;
      return;
    }
  }
  if (self_hi == kSelfShared) {
    auto state_of = [&](uint32_t c, uint32_t& cursor, uint32_t& lo, uint32_t& hown_lo) {
      cursor = S.cur[c];
      lo = S.lo[c];
      hown_lo = S.hlo[c];
    };
    resolve_shared_self(mask, self_lo, &shared, state_of, self_lo, self_hi);
  }
  uint32_t bp = kNone, bi = 0, bg = 0, bc = kNone;
  if (by_row) {
    if (row_c != kNone) {
      uint32_t ci, cp, cg;
      if (class_candidate(L, S.run(L, row_c), self_lo, self_hi, ci, cp, cg)) {
        bp = cp;
        bi = ci;
        bg = cg;
        bc = row_c;
      }
    }
  } else {
    for (uint32_t w = 0; w < W; ++w) {
      if ((mask[w] >> lane) & 1u) {
        const uint32_t c = w * 64 + lane;
        uint32_t ci, cp, cg;
        if (class_candidate(L, S.run(L, c), self_lo, self_hi, ci, cp, cg) && cp < bp) {
          bp = cp;
          bi = ci;
          bg = cg;
          bc = c;
        }
      }
    }
  }
  const uint32_t mn = wave_min_u32(bp);
  bool win = mn != kNone && bp == mn;
  if (mn == kNone) {
    // task_dispatcher.cc:392-396: the requestor's own servant, first eligible class that has it
    uint32_t sc = kNone;
    if (self_lo != kNone && by_row) {
      uint32_t ci, cg;
      if (row_c != kNone && class_self_candidate(L, S.run(L, row_c), self_lo, self_hi, ci, cg)) {
        sc = row_c;
        bi = ci;
        bg = cg;
        bc = row_c;
      }
    } else if (self_lo != kNone) {
      for (uint32_t w = 0; w < W && sc == kNone; ++w) {
        if ((mask[w] >> lane) & 1u) {
          const uint32_t c = w * 64 + lane;
          uint32_t ci, cg;
          if (class_self_candidate(L, S.run(L, c), self_lo, self_hi, ci, cg)) {
            sc = c;
            bi = ci;
            bg = cg;
            bc = c;
          }
        }
      }
    }
    const uint32_t first = wave_min_u32(sc);
    if (first == kNone) {
      if (lane == 0) slot_of[t] = kIdxTimeout;
      return;
    }
    win = sc == first;
  }
  if (win) {
    slot_of[t] = bg;
    ClassRun r = S.run(L, bc);
    const bool had = r.lo < r.cursor;
    class_consume(L, r, bi, self_lo, self_hi);
    S.cur[bc] = r.cursor;
    S.lo[bc] = r.lo;
    S.hlo[bc] = r.hown_lo;
    S.hhi[bc] = r.hown_hi;
    S.hp[bc] = r.head_p;
    if (hpk) hpk[bc * hpk_stride] = pack_head(r.head_p, bc, cbits);  // (k_walk_groups: head rank and class in one word)
    S.hg[bc] = r.head_g;
    S.np[bc] = r.cursor + 1 < r.end ? list_rank(L, r.cursor + 1) : kNone;
    S.ng[bc] = r.cursor + 1 < r.end ? list_slot(L, r.cursor + 1) : kNone;
    const bool has = r.lo < r.cursor;
    if (has != had) holew[bc >> 6] ^= 1ull << (bc & 63u);
  }
}

// walk != 0 (launched with ONE workgroup): the speculation is not converging — pools of many
// small classes with sparse eligibility have no "level" the guesses could start from, and a
// correction then travels one chunk per round. The wave takes the first chunk whose start guess
// changed (everything before it is final: its guess IS its predecessor's final end state) and
// walks from there to the end of the batch with the state in LDS, leaving guesses and end
// states consistent behind it; the next k_update finds nothing to change.
// walk == 2 (256 threads): waves 1-3 are PREFETCHERS. A lone walker that fetches the list entry
// after next itself waits out a cold memory access per pick — a wave's vmcnt is wave-wide and in
// order —, 3.5 us per request. Here the walker never touches memory for the lists: every class
// has a four-entry ring of upcoming entries in LDS, which the prefetch waves keep filled behind
// its cursor (single-writer counters: the walker owns `cur`, the prefetchers own `fill`; a slot
// is position & 3, free once the walker has passed it; entries are written before `fill` moves).
__global__ __launch_bounds__(256) void k_sim_wide(ClassLists L, TaskTable T, uint32_t n_tasks,
                                                 uint32_t chunk_size, uint32_t n_chunks,
                                                 ClassState* __restrict__ guess,
                                                 ClassState* __restrict__ endst, uint8_t* dirty,
                                                 uint32_t* __restrict__ slot_of, SharedIpTable shared,
                                                 uint32_t round, DeviceParams* prm, uint32_t walk,
                                                 WideLists wl) {
  extern __shared__ uint32_t wsm[];
  if (blockIdx.x == 0 && threadIdx.x == 0) prm->n_changed[round & 63] = 0;
  uint32_t k = blockIdx.x;
  const uint32_t lane = threadIdx.x & 63u, wave_id = threadIdx.x >> 6, C = L.n_classes, W = T.words;
  const bool prefetched = walk == 2;
  // walk == 2 only: [fill | ring_p[4] | ring_g[4]] behind the state arrays and the block's masks
  // LDS: nine state arrays | "walk done" word | the block's class masks (8-byte aligned) | walk only:
  const uint32_t W8 = (W + 7u) & ~7u, Cpad = W8 * 64;
  const uint32_t state_words = (kWideFields - 2) * C + 2 * Cpad;
  const uint32_t mask_at = (state_words + 3u) & ~1u;
  uint32_t* const wdone = wsm + (size_t)state_words;
  uint32_t* const wfill = wsm + mask_at + (size_t)(64 * W8 + 2 * W) * 2;
  uint32_t* const wring_p = wfill + C;
  uint32_t* const wring_g = wring_p + (size_t)kWalkRing * C;
  const uint32_t* const vcur = wsm;  // == S.cur
  if (prefetched && wave_id != 0) {
    // ---- prefetch waves: class c belongs to wave 1 + (c / 64) % 3, lane c % 64
    __syncthreads();  // (the walker has set the state up)
    const uint32_t* endv = wsm + 4 * C;  // == S.end
    while (lds_load_u32(wdone) == 0) {
      for (uint32_t c = (wave_id - 1) * 64 + lane; c < C; c += 192) {
        const uint32_t cur = lds_load_u32(vcur + c), e = endv[c];
        uint32_t pos = lds_load_u32(wfill + c);
        pos = pos > cur + 2 ? pos : cur + 2;
        const uint32_t lim = cur + 2 + kWalkRing < e ? cur + 2 + kWalkRing : e;
        if (pos < lim) {
          uint32_t tp[kWalkRing], tg[kWalkRing];
#pragma unroll
          for (uint32_t u = 0; u < kWalkRing; ++u) {
            tp[u] = tg[u] = kNone;
            if (pos + u < lim) {
              tp[u] = list_rank(L, pos + u);
              tg[u] = list_slot(L, pos + u);
            }
          }
#pragma unroll
          for (uint32_t u = 0; u < kWalkRing; ++u) {
            if (pos + u < lim) {
              lds_store_u32(wring_p + (size_t)((pos + u) & (kWalkRing - 1)) * C + c, tp[u]);
              lds_store_u32(wring_g + (size_t)((pos + u) & (kWalkRing - 1)) * C + c, tg[u]);
            }
          }
          // (DS operations of a wave execute in order: the entries are in before `fill` moves)
          lds_store_u32(wfill + c, lim);
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    return;
  }
  if (walk) {
    k = n_chunks;
    for (uint32_t base = 0; base < n_chunks && k == n_chunks; base += 64) {
      const uint64_t m = __ballot(base + lane < n_chunks && dirty[base + lane] != 0);
      if (m) k = base + (uint32_t)__builtin_ctzll(m);
    }
    if (k >= n_chunks) {  // nothing left to walk (the prefetch waves are told, then let go)
      if (prefetched) {
        if (lane == 0) lds_store_u32(wdone, 1u);
        __syncthreads();
      }
      return;
    }
  } else if (k >= n_chunks || !dirty[k]) {
    return;
  }
  // (cur | lo | hlo | hhi | end | np | ng: C words each; hp | hg: Cpad words each, kNone beyond C)
  WideState S{wsm,         wsm + C,     wsm + 2 * C,        wsm + 3 * C,        wsm + 4 * C,
              wsm + 7 * C, wsm + 7 * C + Cpad, wsm + 5 * C, wsm + 6 * C,
              (const uint64_t*)(wsm + mask_at) + (size_t)64 * W8 + W};
  // The class masks of the current block of 64 requests, staged with coalesced loads: a
  // request's W words are then wave-uniform LDS reads instead of W memory round trips.
  uint64_t* const bmask = (uint64_t*)(wsm + mask_at);
  // Bit c & 63 of word c / 64: class c has holes. A request takes the general step only if one of
  // ITS classes has (with sparse eligibility some class nearly always has holes: a slot stepped
  // over by its servant's own host waits for one of the few requests that may use it).
  uint64_t* const holew = bmask + (size_t)64 * W8;
  const ClassState* start = guess + (size_t)k * C;
  // ---- start state (clamped like class_run_init: speculative states may be anything)
  for (uint32_t c = lane; c < C; c += 64) {
    const ClassState st = start[c];
    const uint32_t b = L.cls_begin[c], e = L.cls_begin[c + 1];
    uint32_t cur = st.cursor, lo = st.lo;
    cur = cur < b ? b : (cur > e ? e : cur);
    lo = lo < b ? b : (lo > cur ? cur : lo);
    S.cur[c] = cur;
    S.lo[c] = lo;
    S.hlo[c] = st.hown_lo;
    S.hhi[c] = st.hown_hi;
    S.end[c] = e;
  }
  uint32_t my_holes = 0;  // classes of this lane that have holes
  for (uint32_t c = lane; c < C; c += 64) {
    const uint32_t cur = S.cur[c], e = S.end[c];
    S.hp[c] = cur < e ? list_rank(L, cur) : kNone;
    S.hg[c] = cur < e ? list_slot(L, cur) : kNone;
    S.np[c] = cur + 1 < e ? list_rank(L, cur + 1) : kNone;
    S.ng[c] = cur + 1 < e ? list_slot(L, cur + 1) : kNone;
    my_holes += S.lo[c] < cur ? 1u : 0u;
  }
  for (uint32_t c = C + lane; c < Cpad; c += 64) {
    S.hp[c] = kNone;
    S.hg[c] = kNone;
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t w = 0; w < W; ++w) {
    const uint32_t c = w * 64 + lane;
    const uint64_t hb = __ballot(c < C && S.lo[c] < S.cur[c]);
    const uint64_t sb = __ballot(c < C && L.cls_single && L.cls_single[c] != 0);
    if (lane == 0) {
      holew[w] = hb;
      holew[W + w] = sb;  // (== S.singlew[w])
    }
  }
  (void)my_holes;
  __builtin_amdgcn_wave_barrier();
  if (prefetched) {
    for (uint32_t c = lane; c < C; c += 64) lds_store_u32(wfill + c, 0u);
    if (lane == 0) lds_store_u32(wdone, 0u);
    __syncthreads();  // the prefetch waves start
  }
  // The walker's "entry after next" of class c from the ring (waits for the prefetchers if the
  // class was picked faster than they refill, which takes several picks within one of their sweeps).
  auto ring_take = [&](uint32_t c, uint32_t pos, uint32_t& p, uint32_t& g) {
    uint32_t spins = 0;
    while (lds_load_u32(wfill + c) <= pos && ++spins < (1u << 20)) __builtin_amdgcn_s_sleep(1);
    if (lds_load_u32(wfill + c) > pos) {
      p = lds_load_u32(wring_p + (size_t)(pos & (kWalkRing - 1)) * C + c);
      g = lds_load_u32(wring_g + (size_t)(pos & (kWalkRing - 1)) * C + c);
    } else {  // (never expected: a prefetch wave that does not deliver; the walker fetches itself)
      p = list_rank(L, pos);
      g = list_slot(L, pos);
    }
  };
  // One fetch of "the entry after next" in flight per lane.
  uint32_t pend_c = 0, pend_p = kNone, pend_g = kNone;
  bool pend_on = false;
  auto flush = [&]() {
    if (pend_on) {
      S.np[pend_c] = pend_p;
      S.ng[pend_c] = pend_g;
      pend_on = false;
    }
  };

#ifdef YDC_PHASE_PROBE
  unsigned long long pr_stage = 0, pr_scan = 0, pr_take = 0, pr_gen = 0, pr_end = 0, pr_n_fast = 0, pr_n_gen = 0,
                     pr_t = 0, pr_total0 = wall_clock64();
#define YDC_WTICK() (pr_t = wall_clock64())
#define YDC_WACC(x) ((x) += wall_clock64() - pr_t)
#else
#define YDC_WTICK() ((void)0)
#define YDC_WACC(x) ((void)0)
#endif
  for (;;) {  // chunk k (and, walking, every chunk behind it)
  const uint32_t t0 = k * chunk_size, t1 = min(n_tasks, t0 + chunk_size);
  for (uint32_t tb = t0; tb < t1; tb += 64) {
    YDC_WTICK();
    // this block's own-servant ranges: lane i holds request tb + i
    const uint32_t tl = tb + lane;
    const uint32_t slo_v = tl < t1 ? T.self_lo[tl] : kNone, shi_v = tl < t1 ? T.self_hi[tl] : kNone;
    const uint32_t cnt = min(64u, t1 - tb);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i0 = 0; i0 < cnt && !wl.row_of; i0 += 8) {  // (rows of W8 words, zero beyond W; eight loads in flight)
      uint64_t mv[8];
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u)
        mv[u] = i0 + u < cnt && lane < W ? T.mask[(size_t)(tb + i0 + u) * W + lane] : 0;
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u)
        if (i0 + u < cnt && lane < W8) bmask[(size_t)(i0 + u) * W8 + lane] = mv[u];
    }
    // Lists: lane i holds the row of request tb + i (start and length of its eligible classes); the
    // class of lane l in request i + 1's list is fetched while request i is being placed.
    uint32_t off_v = 0, len_v = 0, nxt_c = kNone;
    if (wl.row_of) {
      const uint32_t r = tl < t1 ? wl.row_of[tl] : kNone;
      if (r != kNone) {
        off_v = wl.off[r];
        len_v = wl.off[r + 1] - off_v;
      }
      const uint32_t o = readlane_u32(off_v, 0), n0 = readlane_u32(len_v, 0);
      nxt_c = lane < n0 ? wl.cls[o + lane] : kNone;
    }
    __builtin_amdgcn_wave_barrier();
    YDC_WACC(pr_stage);
    for (uint32_t i = 0; i < cnt; ++i) {
      const uint32_t t = tb + i;
      // (lists: the mask words are only read by the rare general step, from memory)
      const uint64_t* mask = wl.row_of ? T.mask + (size_t)t * W : bmask + (size_t)i * W8;
      YDC_WTICK();
      uint32_t self_lo = readlane_u32(slo_v, i), self_hi = readlane_u32(shi_v, i);
      const uint32_t my_c = nxt_c;  // lists: this lane's class of THIS request's row
      bool general = self_hi == kSelfShared;
      if (wl.row_of)
        general |= __ballot(my_c != kNone && ((holew[my_c >> 6] >> (my_c & 63u)) & 1u)) != 0;
      else
        general |= __ballot(lane < W && (mask[lane < W ? lane : 0] & holew[lane < W ? lane : 0]) != 0) != 0;
      uint32_t best = kNone, bw = 0;
      uint64_t many = 0;
      if (wl.row_of && i + 1 < cnt) {
        const uint32_t o2 = readlane_u32(off_v, i + 1), n2 = readlane_u32(len_v, i + 1);
        nxt_c = lane < n2 ? wl.cls[o2 + lane] : kNone;
      }
      if (wl.row_of && !general) {
        // ---- list form of the fast path: one head rank per eligible class and lane
        const uint32_t n_el = readlane_u32(len_v, i);
        if (n_el == 0) {
          if (lane == 0) slot_of[t] = kIdxEnvNotFound;  // task_dispatcher.cc:105-108
          continue;
        }
        bool own = false;
        if (my_c != kNone) {
          best = S.hp[my_c];
          if (self_lo != kNone && S.hg[my_c] - self_lo < self_hi - self_lo) own = true;
        }
        const uint32_t mn = wave_min_u32(best);
        YDC_WACC(pr_scan);
        YDC_WTICK();
        if (__ballot(own) != 0 || (mn == kNone && self_lo != kNone)) {
          general = true;
        } else if (mn == kNone) {
          if (lane == 0) slot_of[t] = kIdxTimeout;  // :116-118 with timeout == now
          continue;
        } else {
          // The class's own lane (class & 63) keeps its pending fetch: it does the update.
          const uint32_t wlane = (uint32_t)__builtin_ctzll(__ballot(best == mn));
          const uint32_t c = readlane_u32(my_c, wlane);
          if (lane == (c & 63u)) {
            slot_of[t] = S.hg[c];
            flush();
            const uint32_t cur = S.cur[c] + 1;
            uint32_t p2 = kNone, g2 = kNone;
            const bool more_entries = cur + 1 < S.end[c];
            if (prefetched && more_entries) ring_take(c, cur + 1, p2, g2);
            S.cur[c] = cur;
            S.lo[c] = cur;  // (none of the request's classes has holes on this path)
            S.hp[c] = S.np[c];
            S.hg[c] = S.ng[c];
            if (!more_entries || prefetched) {
              S.np[c] = p2;
              S.ng[c] = g2;
            } else {
              pend_c = c;
              pend_p = list_rank(L, cur + 1);
              pend_g = list_slot(L, cur + 1);
              pend_on = true;
            }
          }
          __builtin_amdgcn_wave_barrier();
          YDC_WACC(pr_take);
#ifdef YDC_PHASE_PROBE
          ++pr_n_fast;
#endif
          continue;
        }
      } else if (!general) {
        bool own = false;
        // Eight mask words and the eight head ranks of this lane's classes under them per step, all
        // read unconditionally and side by side: an LDS read takes ~100 cycles to come back, and a
        // scan that tests a bit before it reads the head pays that twice per word (28 words:
        // 3.8 us per request, measured).
        const uint32_t self_len = self_hi - self_lo;
        for (uint32_t w0 = 0; w0 < W8; w0 += 8) {  // (rows and head arrays are padded: no bounds tests)
          uint64_t m[8];
          uint32_t v[8];
#pragma unroll
          for (uint32_t u = 0; u < 8; ++u) {
            m[u] = mask[w0 + u];  // wave-uniform
            v[u] = S.hp[(w0 + u) * 64 + lane];
          }
          if (self_lo != kNone) {  // (uniform) a request from a servant's host: is its own slot at a head?
            uint32_t g[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) g[u] = S.hg[(w0 + u) * 64 + lane];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u)
              if (((m[u] >> lane) & 1u) && g[u] - self_lo < self_len) own = true;
          }
#pragma unroll
          for (uint32_t u = 0; u < 8; ++u) {
            many |= m[u];
            const uint32_t cand = ((m[u] >> lane) & 1u) ? v[u] : kNone;
            if (cand < best) {
              best = cand;
              bw = w0 + u;
            }
          }
        }
        if (many == 0) {
          if (lane == 0) slot_of[t] = kIdxEnvNotFound;  // task_dispatcher.cc:105-108
          continue;
        }
        // An eligible class shows a slot of the requestor's own servant at its head (it has to be
        // stepped over), or nothing is left but possibly the own servant: the general step.
        const uint32_t mn = wave_min_u32(best);
        YDC_WACC(pr_scan);
        YDC_WTICK();
        if (__ballot(own) != 0 || (mn == kNone && self_lo != kNone)) {
          general = true;
        } else if (mn == kNone) {
          if (lane == 0) slot_of[t] = kIdxTimeout;  // :116-118 with timeout == now
          continue;
        } else {
          if (best == mn) {  // ranks are unique: exactly one lane
            const uint32_t c = bw * 64 + lane;
            slot_of[t] = S.hg[c];
            flush();
            const uint32_t cur = S.cur[c] + 1;
            // (the prefetch waves fill positions [cursor + 2, cursor + 6): the entry after next is
            // taken while the cursor still names the old head, or they would step over it)
            uint32_t p2 = kNone, g2 = kNone;
            const bool more_entries = cur + 1 < S.end[c];
            if (prefetched && more_entries) ring_take(c, cur + 1, p2, g2);
            S.cur[c] = cur;
            S.lo[c] = cur;  // (no holes anywhere on this path)
            S.hp[c] = S.np[c];
            S.hg[c] = S.ng[c];
            if (!more_entries || prefetched) {
              S.np[c] = p2;
              S.ng[c] = g2;
            } else {
              pend_c = c;
              pend_p = list_rank(L, cur + 1);
              pend_g = list_slot(L, cur + 1);
              pend_on = true;
            }
          }
          YDC_WACC(pr_take);
#ifdef YDC_PHASE_PROBE
          ++pr_n_fast;
#endif
          continue;
        }
      }
#ifdef YDC_PHASE_PROBE
      ++pr_n_gen;
#endif
      YDC_WTICK();
      // ---- general step (dispatch_core.h's state machine on the LDS state)
      flush();
      __builtin_amdgcn_wave_barrier();
      wide_general_step(L, S, holew, W, mask, self_lo, self_hi, shared, slot_of, t, lane);
      __builtin_amdgcn_wave_barrier();
      YDC_WACC(pr_gen);
    }
  }
  YDC_WTICK();
  flush();
  __builtin_amdgcn_wave_barrier();
  const bool more = walk && k + 1 < n_chunks;
  for (uint32_t c = lane; c < C; c += 64) {
    ClassState s;
    s.cursor = S.cur[c];
    s.lo = S.lo[c];
    const bool holes = s.lo < s.cursor;
    s.hown_lo = holes ? S.hlo[c] : kNone;
    s.hown_hi = holes ? S.hhi[c] : kNone;
    endst[(size_t)k * C + c] = s;
    if (more) guess[(size_t)(k + 1) * C + c] = s;  // (what chunk k + 1 is replayed from, right now)
  }
  if (lane == 0) {
    dirty[k] = 0;
    atomicAdd(&prm->chunk_sims, 1u);
  }
  YDC_WACC(pr_end);
  if (!more) break;
  ++k;
  }
#ifdef YDC_PHASE_PROBE
  if (walk && lane == 0) {
    ydc_phase_probe[0] = pr_stage;
    ydc_phase_probe[1] = pr_scan;
    ydc_phase_probe[2] = pr_take;
    ydc_phase_probe[3] = pr_gen;
    ydc_phase_probe[4] = pr_end;
    ydc_phase_probe[5] = pr_n_fast;
    ydc_phase_probe[6] = pr_n_gen;
    ydc_phase_probe[7] = wall_clock64() - pr_total0;
  }
#endif
  if (prefetched && lane == 0) lds_store_u32(wdone, 1u);
}


// ---------------------------------------------------------------------------
// k_walk_groups — the walk, sixty-four requests at a time (sparse eligibility: many small
// classes, a request may use a few dozen of them — host_tables.h: elig_off / elig_cls).
//
// The lone walker above places one request per step with the whole wave looking at its classes:
// ~0.6 us per request however few classes it has. Here LANE i holds request tb + i of a block
// and scans its own row of eligible classes in LDS (all rows are staged once); then the block is
// resolved in a few iterations, each of which commits every request whose pick is provably the
// one the sequential process makes:
//   * lane j's pick (its eligible class with the lowest-ranked head) is what sequence gives it
//     unless an EARLIER unresolved request may still take that very slot. Earlier requests that
//     commit in the same iteration take heads of other classes, which only raises those heads:
//     j's minimum stays where it is.
//   * so: every lane claims its winner class (LDS atomic min of the lane index); the lowest lane
//     per class is a first picker, the others wait for the next iteration ("losers"). Every
//     unresolved lane marks the classes of its row with its lane index (`taint`, atomic min);
//     a first picker whose winner class carries a mark from an earlier lane waits too and marks
//     its own row (repeated until no lane is added). What is left commits: distinct classes,
//     none of them reachable by anything unresolved before it.
//   * a request that needs the general step (its host's own servant at a head of one of its
//     classes, holes in one of them, a host with several servants, nothing left but perhaps its
//     own servant) is a barrier: the lanes before it resolve first, then the whole wave runs
//     the state machine on it (wide_general_step), then the lanes behind it continue.
//   * EnvironmentNotFound (empty row) and Timeout (every class of the row exhausted: classes only
//     lose entries inside a batch) are final the moment they are seen.
// Same protocol as the walker around it: starts at the first chunk whose start guess changed,
// leaves end states and guesses consistent behind it. One wave.
// ---------------------------------------------------------------------------
// (rows are padded to multiples of eight classes in LDS — a row is read eight ids at a time)
__host__ __device__ inline size_t group_walk_lds_bytes(uint32_t C, uint32_t n_rows, uint32_t n_list,
                                                       bool packed = false) {
  return wide_lds_bytes(C) + (size_t)(packed ? 3 : 2) * C * 4 + (packed ? 8 : 0) + ((size_t)n_rows + 4) * 4 + ((size_t)n_list + 8 * (size_t)n_rows) * 2 + 32 +
         2 * (((size_t)n_rows + 15) & ~(size_t)15);
}

// PACKED (ranks and class ids share a word: head_bits(C) + rank bits <= 32, the planner checks):
// hpk[c] = head rank << cbits | c beside S.hp — the scan of a row is one load and half a
// v_min3_u32 per class instead of a load, a compare and two selects (a lone wave is bound by the
// instructions it issues).
template <bool PACKED>
__global__ __launch_bounds__(64) void k_walk_groups(ClassLists L, TaskTable T, uint32_t n_tasks,
                                                    uint32_t chunk_size, uint32_t n_chunks,
                                                    ClassState* __restrict__ guess,
                                                    ClassState* __restrict__ endst, uint8_t* dirty,
                                                    uint32_t* __restrict__ slot_of, SharedIpTable shared,
                                                    uint32_t round, DeviceParams* prm, WideLists wl,
                                                    uint32_t n_rows, uint32_t n_list, uint32_t whole_batch,
                                                    uint32_t cbits, uint32_t plain_park) {
  extern __shared__ __attribute__((aligned(16))) uint32_t wsm[];
  const uint32_t lane = threadIdx.x, C = L.n_classes, W = T.words;
  if (lane == 0) prm->n_changed[round & 63] = 0;
  uint32_t k = n_chunks;
  for (uint32_t base = 0; base < n_chunks && k == n_chunks; base += 64) {
    const uint64_t m = __ballot(base + lane < n_chunks && dirty[base + lane] != 0);
    if (m) k = base + (uint32_t)__builtin_ctzll(m);
  }
  if (k >= n_chunks) return;  // nothing left to walk
  // LDS: the state arrays and bit rows of k_sim_wide, then claim[C] | taint[C] (or pairs) | row offsets | rows
  const uint32_t W8 = (W + 7u) & ~7u, Cpad = W8 * 64;
  const uint32_t state_words = (kWideFields - 2) * C + 2 * Cpad;
  const uint32_t mask_at = (state_words + 3u) & ~1u;
  WideState S{wsm,         wsm + C,     wsm + 2 * C,        wsm + 3 * C,        wsm + 4 * C,
              wsm + 7 * C, wsm + 7 * C + Cpad, wsm + 5 * C, wsm + 6 * C,
              (const uint64_t*)(wsm + mask_at) + (size_t)64 * W8 + W};
  uint64_t* const holew = (uint64_t*)(wsm + mask_at) + (size_t)64 * W8;
  uint32_t* const claim = wsm + wide_lds_bytes(C) / 4;
  // PACKED: {packed head, taint} pairs — one LDS address per class of a row serves the scan (the
  // head) and the marking (offset 4); else taint[C] alone
  // (PACKED: the first pair is nobody's — head kNone —: rows hold class id + 1, and the groups of
  // eight beyond a row's end read as zeros)
  constexpr uint32_t kTs = PACKED ? 2 : 1;  // words per class in that array
  uint32_t* const pairs = claim + C;
  uint32_t* const hpk = pairs + (PACKED ? 2 : 0);
  uint32_t* const taint = hpk + (PACKED ? 1 : 0);
  const uint32_t cmask = (1u << cbits) - 1;
  uint32_t* const row_off = hpk + kTs * C;  // n_rows + 1: starts of the PADDED rows, in groups of eight ids
  // (16-byte aligned: a row is read eight 16-bit class ids at a time)
  uint8_t* const row_len = (uint8_t*)(row_off + n_rows + 1);           // [n_rows] true lengths (<= 64)
  uint8_t* const row_holes = row_len + ((n_rows + 15) & ~15u);          // [n_rows] 1: a class of the row has holes
  // (an offset from the LDS base, not a rounded-up pointer value: through an integer the pointer
  // loses its address space and every row read becomes a FLAT load)
  const uint32_t rows_at = ((uint32_t)((row_holes + ((n_rows + 15) & ~15u)) - (uint8_t*)wsm) + 15u) & ~15u;
  uint16_t* const row_cls = (uint16_t*)((uint8_t*)wsm + rows_at);
  // claim / taint entries are (0xFFFFFF - iteration) << 6 | lane: an atomic min prefers the
  // current iteration's marks to any older one and the lowest lane among them — nothing is wiped.
  for (uint32_t c = lane; c < C; c += 64) claim[c] = taint[c * kTs] = kNone;
  if (PACKED && lane == 0) pairs[0] = pairs[1] = kNone;
  {
    // Padded row starts: a wave scan over ceil(len / 8), 64 rows per round.
    uint32_t carry = 0;
    for (uint32_t r0 = 0; r0 < n_rows; r0 += 64) {
      const uint32_t r = r0 + lane;
      const uint32_t g = r < n_rows ? (wl.off[r + 1] - wl.off[r] + 7) / 8 : 0u;
      uint32_t incl = g;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
        if ((int)lane >= d) incl += up;
      }
      if (r < n_rows) {
        row_off[r] = carry + incl - g;
        row_len[r] = (uint8_t)(wl.off[r + 1] - wl.off[r]);
      }
      carry += readlane_u32(incl, 63);
    }
    if (lane == 0) row_off[n_rows] = carry;
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t r0 = 0; r0 < n_rows; r0 += 8) {  // (a row is at most 64 classes: one lane per padded entry)
    // padding repeats the row's first class: looked at twice, chosen no differently
    uint32_t v[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {  // (eight rows' loads in flight)
      const uint32_t r = min(r0 + u, n_rows - 1);
      const uint32_t o = wl.off[r], len = wl.off[r + 1] - o;
      v[u] = len ? wl.cls[o + (lane < len ? lane : 0)] + 1u : 0u;  // (class id + 1: 0 is no class)
    }
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const uint32_t r = r0 + u;
      if (r < n_rows) {
        const uint32_t padded = (row_off[r + 1] - row_off[r]) * 8;
        if (lane < padded) row_cls[(size_t)row_off[r] * 8 + lane] = (uint16_t)v[u];
      }
    }
  }
  // ---- start state (clamped like class_run_init: speculative states may be anything)
  const ClassState* start = guess + (size_t)k * C;
  for (uint32_t c = lane; c < C; c += 64) {
    const ClassState st = start[c];
    const uint32_t b = L.cls_begin[c], e = L.cls_begin[c + 1];
    uint32_t cur = st.cursor, lo = st.lo;
    cur = cur < b ? b : (cur > e ? e : cur);
    lo = lo < b ? b : (lo > cur ? cur : lo);
    S.cur[c] = cur;
    S.lo[c] = lo;
    S.hlo[c] = st.hown_lo;
    S.hhi[c] = st.hown_hi;
    S.end[c] = e;
    S.hp[c] = cur < e ? list_rank(L, cur) : kNone;
    if (PACKED) hpk[c * kTs] = pack_head(S.hp[c], c, cbits);
    S.hg[c] = cur < e ? list_slot(L, cur) : kNone;
    S.np[c] = cur + 1 < e ? list_rank(L, cur + 1) : kNone;
    S.ng[c] = cur + 1 < e ? list_slot(L, cur + 1) : kNone;
  }
  for (uint32_t c = C + lane; c < Cpad; c += 64) {
    S.hp[c] = kNone;
    S.hg[c] = kNone;
  }
  __builtin_amdgcn_wave_barrier();
  uint32_t hole_classes = 0;  // (wave-uniform) classes that have holes
  for (uint32_t w = 0; w < W; ++w) {
    const uint32_t c = w * 64 + lane;
    const uint64_t hb = __ballot(c < C && S.lo[c] < S.cur[c]);
    const uint64_t sb = __ballot(c < C && L.cls_single && L.cls_single[c] != 0);
    hole_classes += (uint32_t)__popcll(hb);
    if (lane == 0) {
      holew[w] = hb;
      holew[W + w] = sb;  // (== S.singlew[w])
    }
  }
  __builtin_amdgcn_wave_barrier();
  // "A class of the row has holes" per row: what a request needs to know before it may take the
  // fast path. Recomputed when a general step changes some class's holes (only those do).
  auto mark_rows_with_holes = [&]() {
    for (uint32_t r = lane; r < n_rows; r += 64) {
      bool h = false;
      if (hole_classes) {
        const uint16_t* ids = row_cls + (size_t)row_off[r] * 8;
        for (uint32_t j = 0; j < row_len[r]; ++j) {
          const uint32_t c = ids[j] - 1u;
          h |= ((holew[c >> 6] >> (c & 63u)) & 1u) != 0;
        }
      }
      row_holes[r] = h ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
  };
  mark_rows_with_holes();
  // One fetch of "the entry after next" in flight per lane (issued when the lane commits, stored
  // right before the next commit or general step can need it: scans read heads only). The two
  // loads land in a254 / a255 — the last accumulation registers: one wave per SIMD has 512 registers
  // to itself, the compiler takes ~240 of them and would spill into a0, a1, ... if it had to — and
  // go from there to the LDS (`flush`), both in inline assembly: a loaded VALUE the compiler knows
  // of is copied to the register of the variable it merges into right behind the load, i.e. waited
  // for on the spot — a round trip to the L2 per iteration, a quarter of the walk.
  // plain_park (YDC_TUNE=walk_park=0): the same fetch in two ordinary variables — the compiler waits
  // for them where they are loaded, so the walk is a quarter slower, but nothing rests on registers
  // it does not know about. Kept selectable and run by the parity tests beside the default.
  uint32_t pend_c = 0, pend_i = 0, plain_p = 0, plain_g = 0;
  bool pend_on = false;
  // young: this iteration's first pickers have issued their fetches already (two loads, younger than
  // any this flush is for — the counter is in order): wait for all but those.
  auto flush = [&](bool young) {
    if (pend_on && plain_park) {
      S.ng[pend_c] = plain_g;
      S.np[pend_c] = L.list_p ? plain_p : pend_i;
      pend_on = false;
    }
    if (pend_on) {
      const uint32_t a_np = (uint32_t)(uintptr_t)(S.np + pend_c), a_ng = (uint32_t)(uintptr_t)(S.ng + pend_c);
      if (young) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (L.list_p) {
        asm volatile("ds_write_b32 %0, a254\n\tds_write_b32 %1, a255"
                     :: "v"(a_np), "v"(a_ng) : "a254", "a255", "memory");
      } else {  // (no class partition: a list entry's rank is its index)
        asm volatile("ds_write_b32 %0, a255" :: "v"(a_ng) : "a254", "a255", "memory");
        S.np[pend_c] = pend_i;
      }
      pend_on = false;
    }
  };
  uint32_t epoch = 0xFFFFFFu;  // counts down, one per iteration (a batch has far fewer than 2^24)
#ifdef YDC_PHASE_PROBE
  // Measurement build (tools/walk_probe.py groups): 100 MHz ticks per part of an iteration, counts.
  unsigned long long gp_block = 0, gp_scan = 0, gp_gen = 0, gp_claim = 0, gp_taint = 0, gp_commit = 0, gp_t = 0,
                     gp_iters = 0, gp_gens = 0, gp_rounds = 0, gp_commits = 0, gp_losers = 0, gp_blocked = 0,
                     gp_pending = 0, gp_total0 = wall_clock64();
#define YDC_GTICK() (gp_t = wall_clock64())
#define YDC_GACC(x) do { const unsigned long long now__ = wall_clock64(); (x) += now__ - gp_t; gp_t = now__; } while (0)
#define YDC_GCNT(x, v) ((x) += (v))
#else
#define YDC_GTICK() ((void)0)
#define YDC_GACC(x) ((void)0)
#define YDC_GCNT(x, v) ((void)0)
#endif
  // The block's request columns are fetched a block ahead.
  uint32_t nx_slo = kNone, nx_shi = kNone, nx_row = kNone;
  {
    const uint32_t t = k * chunk_size + lane;
    if (t < min(n_tasks, (k + 1) * chunk_size)) {
      nx_slo = T.self_lo[t];
      nx_shi = T.self_hi[t];
      nx_row = wl.row_of[t];
    }
  }
  for (;;) {  // chunk k and every chunk behind it
    const uint32_t t0 = k * chunk_size, t1 = min(n_tasks, t0 + chunk_size);
    for (uint32_t tb = t0; tb < t1; tb += 64) {
      const uint32_t t = tb + lane;
      const bool valid = t < t1;
      const uint32_t slo = nx_slo, shi = nx_shi, r = nx_row;
      YDC_GTICK();
      // The row: n groups of eight class ids, unpacked into registers for the block (a lone wave
      // is bound by the instructions it issues — ~4.5 cycles each —, not by the LDS: every scan
      // of every iteration walks these 8 n ids, so they are unpacked once, not per scan).
      uint32_t n = 0;
      uint32_t cid[64];
      if (valid && r != kNone) {
        const uint32_t o = row_off[r];
        n = row_off[r + 1] - o;
        const uint4* row = (const uint4*)row_cls + o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint4 q = (uint32_t)j < n ? row[j] : make_uint4(0, 0, 0, 0);
          cid[8 * j + 0] = q.x & 0xFFFFu;
          cid[8 * j + 1] = q.x >> 16;
          cid[8 * j + 2] = q.y & 0xFFFFu;
          cid[8 * j + 3] = q.y >> 16;
          cid[8 * j + 4] = q.z & 0xFFFFu;
          cid[8 * j + 5] = q.z >> 16;
          cid[8 * j + 6] = q.w & 0xFFFFu;
          cid[8 * j + 7] = q.w >> 16;
        }
      }
      if (PACKED) {
        // ... and turned into the LDS addresses of the classes' {head, taint} pairs: the scans and
        // the marking rounds use them as they are (the empty asm keeps the compiler from
        // holding id * 8 instead and adding the base in front of every access).
        const uint32_t pairs_at = (uint32_t)(uintptr_t)pairs;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
          cid[i] = pairs_at + cid[i] * 8u;
          asm volatile("" : "+v"(cid[i]));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 64; ++i) cid[i] -= 1u;  // (the rows hold class id + 1)
      }
      bool pending = valid;
      if (valid && n == 0) {
        slot_of[t] = kIdxEnvNotFound;  // task_dispatcher.cc:105-108
        pending = false;
      }
      bool my_holes = pending && row_holes[r] != 0;
      // (what the last block's last commits fetched: a lane fetches again at its first claim of
      // this block, into the same registers)
      flush(false);
      __builtin_amdgcn_wave_barrier();
      // (behind the flush, which waits for every load in flight)
      {
        // (the next block: of this chunk, or the first of the next one)
        uint32_t tn = tb + 64 + lane, tn_end = t1;
        if (tb + 64 >= t1) {
          tn = (k + 1) * chunk_size + lane;
          tn_end = min(n_tasks, (k + 2) * chunk_size);
        }
        const bool vn = tn < tn_end && tn < n_tasks;
        nx_slo = vn ? T.self_lo[tn] : kNone;
        nx_shi = vn ? T.self_hi[tn] : kNone;
        nx_row = vn ? wl.row_of[tn] : kNone;
      }
      YDC_GACC(gp_block);
      for (;;) {
        const uint64_t pend_mask = __ballot(pending);
        if (!pend_mask) break;
        --epoch;
        YDC_GCNT(gp_iters, 1);
        YDC_GCNT(gp_pending, __popcll(pend_mask));
        const uint32_t tag = epoch << 6;
        // ---- every unresolved lane: the lowest-ranked head among the classes of its row
        uint32_t best = kNone, bc = 0;
        bool gen = false;
        if (pending) {
          gen = shi == kSelfShared || my_holes;
          if (PACKED) {
            // (two groups per wait: a scan is as many LDS round trips as it has waits)
            uint32_t bk = kNone;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              if ((uint32_t)j < n) {
                uint32_t h[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) h[u] = *(const lds_u32_t*)(uintptr_t)cid[8 * j + u];
                const uint32_t a = min(min(min(bk, h[0]), min(h[1], h[2])), min(min(h[3], h[4]), min(h[5], min(h[6], h[7]))));
                const uint32_t b = min(min(min(h[8], h[9]), min(h[10], h[11])), min(min(h[12], h[13]), min(h[14], h[15])));
                bk = min(a, b);
              }
            }
            if (bk != kNone) {
              best = bk >> cbits;
              bc = bk & cmask;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if ((uint32_t)j < n) {
                uint32_t hp[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) hp[u] = S.hp[cid[8 * j + u]];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  if (hp[u] < best) {
                    best = hp[u];
                    bc = cid[8 * j + u];
                  }
                }
              }
            }
          }
          // From a servant's host: only its own servant at the head of the WINNING class matters —
          // at the head of any other class of the row it would be stepped over to a higher-ranked
          // entry, which cannot beat the winner (whose head is below that class's head already).
          if (slo != kNone && best != kNone && S.hg[bc] - slo < shi - slo) gen = true;
          if (best == kNone && slo != kNone) gen = true;  // nothing left but perhaps its own servant (:392-396)
        }
        const uint64_t gen_mask = __ballot(pending && gen);
        const uint32_t first_gen = gen_mask ? (uint32_t)__builtin_ctzll(gen_mask) : 64u;
        YDC_GACC(gp_scan);
        if (first_gen == (uint32_t)__builtin_ctzll(pend_mask)) {
          // The request that needs the state machine is the next one in sequence: the whole wave.
          const uint32_t tg = tb + first_gen;
          const uint32_t g_lo = readlane_u32(slo, first_gen), g_hi = readlane_u32(shi, first_gen);
          flush(false);
          __builtin_amdgcn_wave_barrier();
          // (the request's classes: lane l takes entry l of its row)
          const uint32_t g_row = readlane_u32(r, first_gen);
          const uint32_t g_len = row_len[g_row];
          const uint32_t g_c = lane < g_len ? (uint32_t)row_cls[(size_t)row_off[g_row] * 8 + lane] - 1u : kNone;
          wide_general_step(L, S, holew, W, T.mask + (size_t)tg * W, g_lo, g_hi, shared, slot_of, tg, lane, g_c,
                            PACKED ? hpk : nullptr, cbits, kTs);
          __builtin_amdgcn_wave_barrier();
          uint32_t holes_now = 0;
          for (uint32_t w = 0; w < W; ++w) holes_now += (uint32_t)__popcll(holew[w]);
          if (holes_now != hole_classes) {  // (a step changes one class's holes at most)
            hole_classes = holes_now;
            mark_rows_with_holes();
          }
          if (lane == first_gen) pending = false;
          my_holes = pending && row_holes[r] != 0;
          YDC_GACC(gp_gen);
          YDC_GCNT(gp_gens, 1);
          continue;
        }
        bool cand = pending && lane < first_gen;
        if (cand && best == kNone) {
          slot_of[t] = kIdxTimeout;  // :116-118 with timeout == now; classes only lose entries: final
          pending = false;
          cand = false;
        }
        if (cand) atomicMin(&claim[bc], tag | lane);
        __builtin_amdgcn_wave_barrier();
        // (the claim word, and with it — one LDS round trip — what a first picker's fetch needs)
        uint32_t claimed = kNone, cur0 = 0, end0 = 0;
        if (cand) {
          claimed = claim[bc];
          cur0 = S.cur[bc];
          end0 = S.end[bc];
        }
        const bool loser = cand && claimed != (tag | lane);
        bool blocked = false, mark = loser;
        // A first picker fetches the entry that becomes its class's `next` if it commits, now: the
        // marking rounds, the commit and the next iteration's scan and rounds pass before `flush`
        // wants it (issued at the commit it arrived a third of a microsecond late). A first picker
        // that ends up waiting has fetched for nothing.
        const bool fetching = cand && !loser && cur0 + 2 < end0;
        const bool young = __ballot(fetching) != 0 && !plain_park;
        if (fetching && plain_park) {
          plain_g = L.list_g[(size_t)(cur0 + 2) * L.stride];
          plain_p = L.list_p ? L.list_p[(size_t)(cur0 + 2) * L.stride] : 0u;
        } else if (fetching) {
          const uint32_t* ag = L.list_g + (size_t)(cur0 + 2) * L.stride;
          const uint32_t* ap = L.list_p ? L.list_p + (size_t)(cur0 + 2) * L.stride : ag;
          asm volatile("global_load_dword a254, %0, off\n\tglobal_load_dword a255, %1, off"
                       :: "v"(ap), "v"(ag) : "a254", "a255", "memory");
        }
        YDC_GACC(gp_claim);
        YDC_GCNT(gp_losers, __popcll(__ballot(loser)));
        for (;;) {
          YDC_GCNT(gp_rounds, 1);
          if (mark) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if ((uint32_t)j < n) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  if (PACKED)
                    __hip_atomic_fetch_min((lds_u32_t*)(uintptr_t)(cid[8 * j + u] + 4u), tag | lane, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                  else
                    atomicMin(&taint[cid[8 * j + u]], tag | lane);
                }
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
          // (a mark of this iteration from a lower lane: the tag matches and the value is smaller)
          mark = cand && !loser && !blocked && taint[bc * kTs] < (tag | lane);
          if (!__ballot(mark)) break;
          blocked |= mark;
        }
        YDC_GACC(gp_taint);
        YDC_GCNT(gp_blocked, __popcll(__ballot(blocked)));
        YDC_GCNT(gp_commits, __popcll(__ballot(cand && !loser && !blocked)));
        flush(young);  // (what the lanes that committed an iteration ago fetched)
        __builtin_amdgcn_wave_barrier();
        if (cand && !loser && !blocked) {
          // ---- commit: the head of class bc is this request's slot; no holes on this path
          slot_of[t] = S.hg[bc];
          const uint32_t cur = cur0 + 1;
          S.cur[bc] = cur;
          S.lo[bc] = cur;
          const uint32_t nhp = S.np[bc];
          S.hp[bc] = nhp;
          if (PACKED) hpk[bc * kTs] = pack_head(nhp, bc, cbits);
          S.hg[bc] = S.ng[bc];
          if (fetching) {  // (cur + 1 < end: entry cur + 1 is on its way, see the claim)
            pend_c = bc;
            pend_i = cur + 1;
            pend_on = true;
          } else {
            S.np[bc] = kNone;
            S.ng[bc] = kNone;
          }
          pending = false;
        }
        __builtin_amdgcn_wave_barrier();
        YDC_GACC(gp_commit);
      }
    }
    const bool more = k + 1 < n_chunks;
    if (!whole_batch || !more) {
      // (walking the whole batch from its first request, nobody reads the states in between)
      flush(false);
      __builtin_amdgcn_wave_barrier();
      for (uint32_t c = lane; c < C; c += 64) {
        ClassState s;
        s.cursor = S.cur[c];
        s.lo = S.lo[c];
        const bool holes = s.lo < s.cursor;
        s.hown_lo = holes ? S.hlo[c] : kNone;
        s.hown_hi = holes ? S.hhi[c] : kNone;
        endst[(size_t)k * C + c] = s;
        if (more) guess[(size_t)(k + 1) * C + c] = s;  // (what chunk k + 1 is replayed from, right now)
      }
    }
    if (lane == 0) {
      dirty[k] = 0;
      atomicAdd(&prm->chunk_sims, 1u);
    }
    if (!more) break;
    ++k;
  }
#ifdef YDC_PHASE_PROBE
  if (lane == 0) {
    const unsigned long long v[16] = {gp_block, gp_scan, gp_gen, gp_claim, gp_taint, gp_commit, wall_clock64() - gp_total0,
                                      gp_iters, gp_gens, gp_rounds, gp_commits, gp_losers, gp_blocked, gp_pending, 0, 0};
    for (int i = 0; i < 16; ++i) ydc_phase_probe[16 + i] = v[i];
  }
#endif
#undef YDC_GTICK
#undef YDC_GACC
#undef YDC_GCNT
}

}  // namespace ydc
#endif  // YADCC_AMD_WIDE_KERNEL_H_

// dispatch_core.h — placement arithmetic shared by every kernel of the MI355X
// task-dispatch path (and, compiled for the host, by tests/model/, which
// replays the exact same code single-threaded to check it against the oracle).
//
// What is computed (reference: yadcc/scheduler/task_dispatcher.cc):
//   * GetCapacityAvailable (:283-313) in closed form per (servant, running) "slot";
//   * the pick order of UnsafePickServantFor / UnsafeTryPickServantFor
//     (:362-451) as a total order on slots: (tier, running/capacity, servant);
//   * eligibility (:316-344) as compatibility between a task and a servant class;
//   * the requestor-avoids-itself rule (:372-396) as a per-task excluded slot range.
//
// N sequential WaitForStartingNewTask calls on a frozen registry consume, per
// class of servants, the class's slots in ascending order; a task takes the
// smallest unconsumed slot among the classes it is compatible with, skipping
// slots of its own host unless nothing else is left. sim_chunk() replays that
// for a contiguous chunk of tasks from a given per-class state.
#ifndef YADCC_AMD_DISPATCH_CORE_H_
#define YADCC_AMD_DISPATCH_CORE_H_

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define YDC_HD __host__ __device__ __forceinline__
#else
#define YDC_HD inline
#endif

namespace ydc {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kIdxTimeout = 0xFFFFFFFFu;      // WaitStatus::Timeout
constexpr uint32_t kIdxEnvNotFound = 0xFFFFFFFEu;  // WaitStatus::EnvironmentNotFound
constexpr uint32_t kSelfShared = 0xFFFFFFFEu;      // TaskInfo::self_hi marker: resolve `self` at run time
constexpr uint32_t kSelfServant = 0xFFFFFFFDu;     // TaskInfo::self_hi marker: self_lo is a servant index
constexpr uint32_t kFlagDedicated = 1u;
constexpr uint32_t kFlagLowMemory = 2u;
constexpr uint32_t kMaxWaveClasses = 256;  // lane-per-class kernel: 4 classes per lane
constexpr uint32_t kMaxExactCapBits = 21;  // 3*b <= 63: r << 2b fits in 64 bits

// ---------------------------------------------------------------------------
// Slots of one servant.
//
// free(r) <=> r < cap(r), cap(r) = min(max_tasks, max(nproc - max(load - r, 0), 0))
// (task_dispatcher.cc:283-313,353). Working the cases load >= nproc / r < load /
// r >= load through gives the closed form
//     free(r) <=> !low_memory && load < nproc && r < min(max_tasks, nproc)
// so a servant contributes the slots r = running .. min(max_tasks, nproc) - 1.
// ---------------------------------------------------------------------------
YDC_HD uint32_t servant_slot_count(uint32_t nproc, uint32_t load, uint32_t max_tasks,
                                   uint32_t running, uint32_t flags) {
  if ((flags & kFlagLowMemory) || max_tasks == 0 || load >= nproc) return 0;
  uint32_t top = max_tasks < nproc ? max_tasks : nproc;
  return running < top ? top - running : 0;
}

// Capacity seen by the pick made at running == r (only valid for a free slot).
YDC_HD uint32_t slot_capacity(uint32_t nproc, uint32_t load, uint32_t max_tasks, uint32_t r) {
  uint32_t foreign = load > r ? load - r : 0;  // :308-309
  uint32_t cap = nproc - foreign;              // > 0 because load < nproc
  return cap < max_tasks ? cap : max_tasks;    // :312
}

// 0 when UnsafeTryPickDedicatedServantFor's predicate holds (:404-407), else 1.
YDC_HD uint32_t slot_tier(uint32_t nproc, uint32_t flags, uint32_t r) {
  return ((flags & kFlagDedicated) && (uint64_t)r * 2 < nproc) ? 0u : 1u;
}

// Order-isomorphic integer image of (tier, double(r)/cap) for capacities below
// 2^cap_bits (cap_bits <= 21): floor(r * 4^cap_bits / cap) separates any two
// distinct fractions with denominators < 2^cap_bits (they differ by more than
// 4^-cap_bits), and IEEE double division separates and orders them the same
// way (denominators < 2^26). 2*cap_bits + 1 significant bits.
YDC_HD uint64_t slot_key_exact(uint32_t tier, uint32_t r, uint32_t cap, uint32_t cap_bits) {
  uint64_t q;
  if (cap_bits <= 10) {
    q = ((uint32_t)r << (2 * cap_bits)) / cap;  // < 2^30: 32-bit divide
  } else {
    q = ((uint64_t)r << (2 * cap_bits)) / cap;
  }
  return ((uint64_t)tier << (2 * cap_bits)) | q;
}

// Fallback for absurd capacities (>= 2^21): the reference's own double, bit for
// bit (:440-441); positive doubles order like their bit patterns.
YDC_HD uint64_t slot_key_fp64(uint32_t tier, uint32_t r, uint32_t cap) {
  double u = (double)r / (double)cap;
  uint64_t bits;
#if defined(__HIP_DEVICE_COMPILE__)
  bits = (uint64_t)__double_as_longlong(u);
#else
  __builtin_memcpy(&bits, &u, sizeof(bits));
#endif
  return ((uint64_t)tier << 63) | bits;
}

YDC_HD double slot_utilization(uint32_t r, uint32_t cap) { return (double)r / (double)cap; }

// Smallest r in [running, top] (top = first r that is not a slot) whose exact key — with the
// part id `part_key` already shifted into place — is >= K; top if there is none. The key is
// strictly increasing in r, so this is "how many of the servant's slots sort below K" + running:
// the closed-form count behind the per-rank key windows of the multi-GPU path.
// A float estimate (utilisation u = q / 4^b, r ~ u * capacity) lands within a slot or two; the
// exact integer key decides, so the estimate only costs steps, never correctness.
YDC_HD uint32_t first_slot_not_below(uint32_t nproc, uint32_t load, uint32_t max_tasks,
                                     uint32_t running, uint32_t flags, uint64_t part_key,
                                     uint64_t K, uint32_t cap_bits) {
  const uint32_t n = servant_slot_count(nproc, load, max_tasks, running, flags);
  if (n == 0) return running;
  const uint32_t top = running + n;
  auto key_at = [&](uint32_t r) {
    return part_key | slot_key_exact(slot_tier(nproc, flags, r), r,
                                     slot_capacity(nproc, load, max_tasks, r), cap_bits);
  };
  if (key_at(running) >= K) return running;
  if (key_at(top - 1) < K) return top;
  // Estimate inside the tier the threshold falls into.
  const uint64_t qmask = (1ull << (2 * cap_bits)) - 1;
  const float u = (float)(K & qmask) / (float)(qmask + 1);
  const uint32_t cmin = max_tasks < nproc ? max_tasks : nproc;
  float est = u * (float)cmin;  // r >= load (or capped by max_tasks): capacity is constant
  if (load > running) {
    const float a = (float)(nproc - load);  // r < load: capacity a + r while below max_tasks
    const float grow = u < 0.999f ? u * a / (1.0f - u) : (float)top;
    if (grow < (float)load && a + grow < (float)max_tasks) est = grow;
  }
  uint32_t r = est <= (float)running ? running : (est >= (float)(top - 1) ? top - 1 : (uint32_t)est);
  // Tier: all tier-0 slots sort below all tier-1 slots.
  const uint32_t ktier = (uint32_t)((K & ~part_key) >> (2 * cap_bits)) & 1u;
  if (slot_tier(nproc, flags, r) != ktier) {
    // first tier-1 slot of a dedicated servant: r * 2 >= nproc
    const uint32_t t1 = (nproc + 1) / 2;
    r = ktier ? (t1 < running ? running : (t1 > top - 1 ? top - 1 : t1))
              : (t1 == 0 || t1 - 1 < running ? running : (t1 - 1 > top - 1 ? top - 1 : t1 - 1));
  }
  // Bracket: key_at(lo) < K <= key_at(hi). Gallop away from the estimate, then bisect.
  uint32_t lo = running, hi = top - 1;
  if (r <= lo) r = lo + 1;
  if (r > hi) r = hi;
  if (key_at(r) >= K) {
    hi = r;
    for (uint32_t step = 1; hi - lo > 1; step *= 2) {
      const uint32_t probe = hi - lo > step ? hi - step : lo + 1;
      if (key_at(probe) >= K) {
        hi = probe;
      } else {
        lo = probe;
        break;
      }
    }
  } else {
    lo = r;
    for (uint32_t step = 1; hi - lo > 1; step *= 2) {
      const uint32_t probe = hi - lo > step ? lo + step : hi - 1;
      if (key_at(probe) < K) {
        lo = probe;
      } else {
        hi = probe;
        break;
      }
    }
  }
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (key_at(mid) >= K) hi = mid; else lo = mid;
  }
  return hi;
}

// The same answer without a search (the bin sort evaluates it once per servant and bin boundary,
// bin_sort.h). With cap(r) = a + r while r < g (a = nproc - load: the servant's own running
// tasks give capacity back, task_dispatcher.cc:308-309) and cap(r) = cB = min(nproc, max_tasks)
// from g on, "floor(r 4^b / cap(r)) >= q" turns into r (4^b - q) >= q a below g and
// r 4^b >= q cB from g on; tier 0 holds exactly for r < t1 = ceil(nproc / 2) on a dedicated
// servant (:404-407). The first r whose (tier, quotient) reaches K's follows from the monotone
// key; one division (32-bit for capacities below 2^10).
YDC_HD uint32_t first_slot_not_below_direct(uint32_t nproc, uint32_t load, uint32_t max_tasks,
                                            uint32_t running, uint32_t flags, uint64_t part_key,
                                            uint64_t K, uint32_t cap_bits) {
  const uint32_t n = servant_slot_count(nproc, load, max_tasks, running, flags);
  if (n == 0) return running;
  const uint32_t top = running + n;
  if (K <= part_key) return running;  // an earlier part, or the very first key of this one
  const uint64_t k = K - part_key;
  if (k >> (2 * cap_bits + 1)) return top;  // a later part
  const uint32_t ktier = (uint32_t)(k >> (2 * cap_bits));
  const uint64_t one = 1ull << (2 * cap_bits);
  const uint64_t q = k & (one - 1);
  const uint32_t t1 = (flags & kFlagDedicated) ? nproc / 2 + (nproc & 1u) : 0u;  // first tier-1 slot
  uint64_t ru = 0;  // first r >= 0 whose quotient is >= q
  if (q) {
    const uint32_t a = nproc - load;  // > 0: the servant has slots
    const uint32_t cB = max_tasks < nproc ? max_tasks : nproc;
    const uint32_t g = max_tasks > a ? (load < max_tasks - a ? load : max_tasks - a) : 0u;
    uint64_t rA = ~0ull;
    if (g) {  // (then a < 2^cap_bits: q a < 2^(3 cap_bits))
      const uint64_t den = one - q;
      if (cap_bits <= 10) rA = ((uint32_t)q * a + (uint32_t)den - 1) / (uint32_t)den;
      else rA = (q * a + den - 1) / den;
    }
    if (rA < g) {
      ru = rA;
    } else {
      const uint64_t rB = (q * cB + one - 1) >> (2 * cap_bits);
      ru = rB > g ? rB : g;
    }
  }
  const uint64_t first = ktier ? (ru > t1 ? ru : t1) : (ru < t1 ? ru : t1);
  return first <= running ? running : (first >= top ? top : (uint32_t)first);
}

// ... and in 32-bit arithmetic throughout, for capacities below 2^10 and keys (part id included)
// below 2^32 — what the bin sort meets on ordinary pools: q a and q cB stay below 2^30.
YDC_HD uint32_t first_slot_not_below_direct32(uint32_t nproc, uint32_t load, uint32_t max_tasks,
                                              uint32_t running, uint32_t flags, uint32_t part_key,
                                              uint32_t K, uint32_t cap_bits) {
  const uint32_t n = servant_slot_count(nproc, load, max_tasks, running, flags);
  if (n == 0) return running;
  const uint32_t top = running + n;
  if (K <= part_key) return running;
  const uint32_t k = K - part_key;
  if (k >> (2 * cap_bits + 1)) return top;
  const uint32_t ktier = k >> (2 * cap_bits);
  const uint32_t one = 1u << (2 * cap_bits);
  const uint32_t q = k & (one - 1);
  const uint32_t t1 = (flags & kFlagDedicated) ? nproc / 2 + (nproc & 1u) : 0u;
  uint32_t ru = 0;
  if (q) {
    const uint32_t a = nproc - load;
    const uint32_t cB = max_tasks < nproc ? max_tasks : nproc;
    const uint32_t g = max_tasks > a ? (load < max_tasks - a ? load : max_tasks - a) : 0u;
    uint32_t rA = 0xFFFFFFFFu;
    if (g) {
      const uint32_t den = one - q;
      rA = (q * a + den - 1) / den;
    }
    if (rA < g) {
      ru = rA;
    } else {
      const uint32_t rB = (q * cB + one - 1) >> (2 * cap_bits);
      ru = rB > g ? rB : g;
    }
  }
  const uint32_t first = ktier ? (ru > t1 ? ru : t1) : (ru < t1 ? ru : t1);
  return first <= running ? running : (first >= top ? top : first);
}

// The sort key of slot (servant, r), exact or fp64 format, the part id in place.
YDC_HD uint64_t slot_sort_key(uint32_t nproc, uint32_t load, uint32_t max_tasks, uint32_t flags,
                              uint32_t r, uint64_t part_key, bool exact, uint32_t cap_bits) {
  const uint32_t cap = slot_capacity(nproc, load, max_tasks, r), tier = slot_tier(nproc, flags, r);
  return part_key | (exact ? slot_key_exact(tier, r, cap, cap_bits) : slot_key_fp64(tier, r, cap));
}

// How many slots of servant `s` precede the list entry (head_key, head_servant) in its class
// list — i.e. sort before it: smaller key, or the same key on an earlier servant (ties go by
// registry index, the reference's first-wins rule, task_dispatcher.cc:440-447). A class list is
// consumed from the front, so with the entry at the class's cursor this is the number of
// slots the batch took on the servant — running_tasks needs no per-slot bookkeeping.
YDC_HD uint32_t servant_slots_before(uint32_t nproc, uint32_t load, uint32_t max_tasks,
                                     uint32_t running, uint32_t flags, uint32_t s,
                                     uint64_t part_key, uint64_t head_key, uint32_t head_servant,
                                     bool exact, uint32_t cap_bits) {
  const uint32_t n = servant_slot_count(nproc, load, max_tasks, running, flags);
  if (n == 0) return 0;
  const uint32_t top = running + n;
  uint32_t r;
  if (exact) {
    r = first_slot_not_below(nproc, load, max_tasks, running, flags, part_key, head_key, cap_bits);
  } else {
    uint32_t lo = running, hi = top;  // first r in [running, top] with key >= head_key
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (slot_sort_key(nproc, load, max_tasks, flags, mid, part_key, false, cap_bits) < head_key) lo = mid + 1;
      else hi = mid;
    }
    r = lo;
  }
  uint32_t cnt = r - running;
  if (r < top && s < head_servant &&
      slot_sort_key(nproc, load, max_tasks, flags, r, part_key, exact, cap_bits) == head_key)
    ++cnt;  // the tie on an earlier servant sorts first
  return cnt;
}

// ---------------------------------------------------------------------------
// Tasks.
// ---------------------------------------------------------------------------
// Per-task columns produced by the classify kernel.
//   mask[t * words + w]: bit c of word w <=> servant class 64*w + c is eligible;
//                        all zero => EnvironmentNotFound (:105-108).
//   [self_lo[t], self_hi[t]): generation-index range of the slots of the
//     requestor's own servant (kNone, kNone: none). self_hi == kSelfShared: several
//     servants share the requestor's host and self_lo is the first entry of the
//     host's group in the ip table (resolved at run time, see SharedIpTable).
//     self_hi == kSelfServant: self_lo is the index of the requestor's own servant; the
//     matching kernel turns it into the slot range when it stages the request (the bin sort's
//     front classifies requests in the launch that computes slot_base, bin_sort.h).
struct TaskTable {
  const uint64_t* mask;
  const uint32_t* self_lo;
  const uint32_t* self_hi;
  uint32_t words;  // ceil(n_classes / 64), >= 1
};

YDC_HD bool task_mask_empty(const TaskTable& T, uint32_t t) {
  uint64_t any = 0;
  for (uint32_t w = 0; w < T.words; ++w) any |= T.mask[(size_t)t * T.words + w];
  return any == 0;
}

// UnsafeEnumerateEligibleServants (:316-344) per class: the class advertises the
// digest and its version is not below min_version. Classes only contain servants
// with max_tasks != 0. cls_env holds env_words words per class (digest j = bit j % 64
// of word j / 64). Writes `words` mask words.
YDC_HD void task_class_mask(uint32_t env_id, uint32_t min_version, const uint64_t* cls_env,
                            const uint32_t* cls_ver, uint32_t n_classes, uint32_t words,
                            uint64_t* out, uint32_t env_words = 1) {
  for (uint32_t w = 0; w < words; ++w) {
    uint64_t m = 0;
    if (env_id < 64 * env_words) {
      uint32_t c0 = w * 64, c1 = c0 + 64 < n_classes ? c0 + 64 : n_classes;
      for (uint32_t c = c0; c < c1; ++c) {
        if (((cls_env[(size_t)c * env_words + (env_id >> 6)] >> (env_id & 63)) & 1u) &&
            cls_ver[c] >= min_version)
          m |= 1ull << (c - c0);
      }
    }
    out[w] = m;
  }
}

// lower_bound on a sorted u32 array.
YDC_HD uint32_t lower_bound_u32(const uint32_t* a, uint32_t n, uint32_t x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// Largest s with base[s] <= g, base ascending with base[0] == 0 (owner of slot g).
YDC_HD uint32_t owner_of_slot(const uint32_t* slot_base, uint32_t n_servants, uint32_t g) {
  uint32_t lo = 0, hi = n_servants;  // invariant: base[lo] <= g < base[hi]
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (slot_base[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------
// Chunk simulation.
// ---------------------------------------------------------------------------
// Class lists: entries [cls_begin[c], cls_begin[c+1]) of list_p / list_g hold the
// class's slots in ascending global rank; list_p = global rank of the slot in the
// (tier, utilisation, servant) order (NULL when there is a single class: rank ==
// index), list_g = generation index (servant-major) of the slot.
// stride: distance of consecutive entries in 32-bit words — 2 when rank and slot sit side by
// side in 8-byte records (the device's 32-bit-key sort), 1 for two plain arrays.
struct ClassLists {
  const uint32_t* list_p;
  const uint32_t* list_g;
  const uint32_t* cls_begin;  // [n_classes + 1]
  uint32_t n_classes;
  uint32_t stride = 1;
  const uint8_t* cls_single = nullptr;  // [n_classes] the class is one servant's (nullable)
};

// Consumption state of one class: everything below `cursor` is consumed except
// the "holes" — slots of ONE servant (generation range [hown_lo, hown_hi)) that
// requests from that servant's own host stepped over (:372-380) — of which `lo`
// is the smallest (lo == cursor: none). At any time all holes of a class belong
// to the same servant: a hole is only added by a task that finds the class's
// smallest unconsumed slot on its own host, and while holes exist that smallest
// slot is a hole.
struct ClassState {
  uint32_t cursor, lo, hown_lo, hown_hi;
};

// Run-time resolution of `self` when several servants share a host (`self` = first
// servant in registry order that is eligible for the task AND still free, :372-379).
// "Still free" is read off the class state, so a chunk replayed from any start state
// resolves it consistently with that state: a servant's slots sit in its class list in
// ascending order, hence it has an unconsumed slot iff its LAST slot (list position
// pos_last[s]) is at or after the cursor, or is one of the class's holes.
struct SharedIpTable {
  const uint32_t* ip;        // ip table sorted by (ip, servant)
  const uint32_t* servant;
  uint32_t n;
  const uint32_t* class_of;  // per servant (kNone: max_tasks == 0)
  const uint32_t* slot_base; // [n_servants + 1]
  uint32_t n_servants;
  const uint32_t* pos_last;  // per servant with slots: class-list position of its last slot
};

YDC_HD bool servant_still_free(uint32_t pos_last, uint32_t first_g, uint32_t cursor, uint32_t lo,
                               uint32_t hown_lo) {
  return pos_last >= cursor || (lo < cursor && hown_lo == first_g && pos_last >= lo);
}

// Live state of one class inside a simulation (one per lane on the GPU).
struct ClassRun {
  uint32_t cursor, lo, hown_lo, hown_hi;
  uint32_t end;     // cls_begin[c + 1]
  uint32_t head_p;  // rank / generation index of the entry at `cursor`
  uint32_t head_g;  // (kNone when the class is exhausted)
  uint32_t single;  // the class consists of one servant (ClassLists::cls_single)
};

YDC_HD uint32_t list_rank(const ClassLists& L, uint32_t i) {
  return L.list_p ? L.list_p[(size_t)i * L.stride] : i;
}
YDC_HD uint32_t list_slot(const ClassLists& L, uint32_t i) { return L.list_g[(size_t)i * L.stride]; }

YDC_HD void class_load_head(const ClassLists& L, ClassRun& r) {
  if (r.cursor < r.end) {
    r.head_p = list_rank(L, r.cursor);
    r.head_g = list_slot(L, r.cursor);
  } else {
    r.head_p = kNone;
    r.head_g = kNone;
  }
}

// Robust against arbitrary (speculative) start states: every index is clamped
// into the class range.
YDC_HD void class_run_init(const ClassLists& L, uint32_t c, const ClassState& start, ClassRun& r) {
  uint32_t b = L.cls_begin[c], e = L.cls_begin[c + 1];
  uint32_t cur = start.cursor, lo = start.lo;
  cur = cur < b ? b : (cur > e ? e : cur);
  lo = lo < b ? b : (lo > cur ? cur : lo);
  r.cursor = cur;
  r.lo = lo;
  r.hown_lo = start.hown_lo;
  r.hown_hi = start.hown_hi;
  r.end = e;
  r.single = L.cls_single ? L.cls_single[c] : 0u;
  class_load_head(L, r);
}

YDC_HD ClassState class_run_state(const ClassRun& r) {
  ClassState s;
  s.cursor = r.cursor;
  s.lo = r.lo;
  bool holes = r.lo < r.cursor;
  s.hown_lo = holes ? r.hown_lo : kNone;
  s.hown_hi = holes ? r.hown_hi : kNone;
  return s;
}

// The class's smallest unconsumed slot that is NOT on the requestor's own
// servant [self_lo, self_hi). Returns false if there is none.
// skip_to (optional): the caller has found out by itself — a wave looking at 64 entries at a time —
// that the head and the entries behind it up to skip_to are all the requestor's own (and that no
// foreign holes exist): the walk below starts there.
YDC_HD bool class_candidate(const ClassLists& L, const ClassRun& r, uint32_t self_lo,
                            uint32_t self_hi, uint32_t& ci, uint32_t& cp, uint32_t& cg,
                            uint32_t skip_to = kNone) {
  if (r.lo < r.cursor && r.hown_lo != self_lo) {
    // The smallest unconsumed slot is a hole somebody else's request may take.
    ci = r.lo;
    cp = list_rank(L, r.lo);
    cg = list_slot(L, r.lo);
    return true;
  }
  // No holes, or the holes are this requestor's own servant's: first entry
  // at/after the cursor that is not on the requestor's host.
  ci = r.cursor;
  cp = r.head_p;
  cg = r.head_g;
  // One servant's class and it is the requestor's own: nothing but own slots to walk over
  // (the reference looks at servants, not slots: task_dispatcher.cc:372-380 drops `self` once).
  if (r.single && ci < r.end && cg >= self_lo && cg < self_hi) return false;
  if (skip_to != kNone && skip_to > ci) {
    ci = skip_to < r.end ? skip_to : r.end;
    if (ci < r.end) {
      cp = list_rank(L, ci);
      cg = list_slot(L, ci);
    }
  }
  while (ci < r.end && cg >= self_lo && cg < self_hi) {
    ++ci;
    if (ci < r.end) {
      cp = list_rank(L, ci);
      cg = list_slot(L, ci);
    }
  }
  return ci < r.end;
}

// task_dispatcher.cc:392-396 — the requestor's own servant's next slot, if it
// lives in this class and has one left.
YDC_HD bool class_self_candidate(const ClassLists& L, const ClassRun& r, uint32_t self_lo,
                                 uint32_t self_hi, uint32_t& ci, uint32_t& cg) {
  if (r.lo < r.cursor) {
    if (r.hown_lo != self_lo) return false;
    ci = r.lo;
    cg = list_slot(L, r.lo);
    return true;
  }
  if (r.cursor < r.end && r.head_g >= self_lo && r.head_g < self_hi) {
    ci = r.cursor;
    cg = r.head_g;
    return true;
  }
  return false;
}

// Marks entry `ci` (obtained from one of the two functions above for the same
// self range) as consumed. Returns true when the cursor moved, in which case the
// caller must reload the head (class_load_head, or its own cached copy).
YDC_HD bool class_consume_state(const ClassLists& L, ClassRun& r, uint32_t ci, uint32_t self_lo,
                                uint32_t self_hi) {
  if (ci < r.cursor) {
    // Took the smallest hole: the next hole is the next slot of the same
    // servant below the cursor.
    uint32_t j = ci + 1;
    while (j < r.cursor) {
      uint32_t g = list_slot(L, j);
      if (g >= r.hown_lo && g < r.hown_hi) break;
      ++j;
    }
    r.lo = j;  // == cursor: no holes left
    return false;
  }
  if (r.lo == r.cursor) {
    if (ci > r.cursor) {
      // Stepped over own slots [cursor, ci): they become the holes.
      r.hown_lo = self_lo;
      r.hown_hi = self_hi;
    } else {
      r.lo = ci + 1;  // still no holes
    }
  }
  r.cursor = ci + 1;
  return true;
}

YDC_HD void class_consume(const ClassLists& L, ClassRun& r, uint32_t ci, uint32_t self_lo,
                          uint32_t self_hi) {
  if (class_consume_state(L, r, ci, self_lo, self_hi)) class_load_head(L, r);
}

// Resolves `self` for a task whose host runs several servants. state(c, cursor, lo, hown_lo)
// hands out the current state of class c (one implementation per caller: ClassRun array on
// the CPU / thread-per-chunk kernel, lane registers in the wave kernel).
template <typename StateOf>
YDC_HD void resolve_shared_self(const uint64_t* mask, uint32_t group_begin,
                                const SharedIpTable* shared, const StateOf& state,
                                uint32_t& self_lo, uint32_t& self_hi) {
  uint32_t i = group_begin;
  uint32_t ip = shared->ip[i];
  self_lo = self_hi = kNone;
  for (; i < shared->n && shared->ip[i] == ip; ++i) {
    uint32_t s = shared->servant[i];
    uint32_t c = shared->class_of[s];
    if (c == kNone || !((mask[c >> 6] >> (c & 63)) & 1u)) continue;  // not eligible (:324-338)
    const uint32_t b = shared->slot_base[s], e = shared->slot_base[s + 1];
    if (e == b) continue;  // offers nothing in this batch
    uint32_t cursor, lo, hown_lo;
    state(c, cursor, lo, hown_lo);
    if (!servant_still_free(shared->pos_last[s], b, cursor, lo, hown_lo)) continue;  // (:350-358)
    self_lo = b;
    self_hi = e;
    break;
  }
}

// Sequential replay of tasks [t0, t1) from `start` (n_classes entries): writes
// the generation index of the slot each task takes (or kIdxTimeout /
// kIdxEnvNotFound) to out_slot[t] and the end state to `end`. `runs` is scratch
// for n_classes ClassRun. (The wave kernel does the same with one lane per
// class; this form is used for the shared-host path and by the CPU model.)
YDC_HD void sim_chunk(const ClassLists& L, const TaskTable& T, uint32_t t0, uint32_t t1,
                      const ClassState* start, ClassState* end, uint32_t* out_slot,
                      ClassRun* runs, const SharedIpTable* shared) {
  const uint32_t C = L.n_classes;
  for (uint32_t c = 0; c < C; ++c) class_run_init(L, c, start[c], runs[c]);
  for (uint32_t t = t0; t < t1; ++t) {
    const uint64_t* mask = T.mask + (size_t)t * T.words;
    if (task_mask_empty(T, t)) {
      out_slot[t] = kIdxEnvNotFound;  // :105-108
      continue;
    }
    uint32_t self_lo = T.self_lo[t], self_hi = T.self_hi[t];
    if (self_hi == kSelfShared) {
      auto state = [&](uint32_t c, uint32_t& cursor, uint32_t& lo, uint32_t& hown_lo) {
        cursor = runs[c].cursor;
        lo = runs[c].lo;
        hown_lo = runs[c].hown_lo;
      };
      resolve_shared_self(mask, self_lo, shared, state, self_lo, self_hi);
    }
    uint32_t best_p = kNone, best_c = kNone, best_i = 0, best_g = 0;
    for (uint32_t w = 0; w < T.words; ++w) {
      for (uint64_t m = mask[w]; m; m &= m - 1) {
        uint32_t c = w * 64 + (uint32_t)__builtin_ctzll(m);
        uint32_t ci, cp, cg;
        if (class_candidate(L, runs[c], self_lo, self_hi, ci, cp, cg) && cp < best_p) {
          best_p = cp;
          best_c = c;
          best_i = ci;
          best_g = cg;
        }
      }
    }
    if (best_c == kNone && self_lo != kNone) {
      for (uint32_t w = 0; w < T.words && best_c == kNone; ++w) {
        for (uint64_t m = mask[w]; m; m &= m - 1) {
          uint32_t c = w * 64 + (uint32_t)__builtin_ctzll(m);
          uint32_t ci, cg;
          if (class_self_candidate(L, runs[c], self_lo, self_hi, ci, cg)) {
            best_c = c;
            best_i = ci;
            best_g = cg;
            break;
          }
        }
      }
    }
    if (best_c == kNone) {
      out_slot[t] = kIdxTimeout;  // :116-118 with timeout == now
      continue;
    }
    out_slot[t] = best_g;
    class_consume(L, runs[best_c], best_i, self_lo, self_hi);
  }
  for (uint32_t c = 0; c < C; ++c) end[c] = class_run_state(runs[c]);
}

YDC_HD bool class_state_equal(const ClassState& a, const ClassState& b) {
  return a.cursor == b.cursor && a.lo == b.lo && a.hown_lo == b.hown_lo && a.hown_hi == b.hown_hi;
}

// Level guess for the state before `consumed` slots were taken with no holes:
// the first `consumed` slots of the global order.
YDC_HD ClassState level_guess(const ClassLists& L, uint32_t c, uint32_t consumed) {
  uint32_t b = L.cls_begin[c], e = L.cls_begin[c + 1];
  uint32_t cur;
  if (L.list_p) {
    uint32_t lo = b, hi = e;  // first entry of the class whose rank is >= consumed
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (list_rank(L, mid) < consumed) lo = mid + 1; else hi = mid;
    }
    cur = lo;
  } else {
    cur = b + (consumed < e - b ? consumed : e - b);
  }
  ClassState s;
  s.cursor = cur;
  s.lo = cur;
  s.hown_lo = kNone;
  s.hown_hi = kNone;
  return s;
}

}  // namespace ydc
#endif  // YADCC_AMD_DISPATCH_CORE_H_

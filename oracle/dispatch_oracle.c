/* ORACLE — test infrastructure, NOT product code. See dispatch_oracle.h. */
#include "dispatch_oracle.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* GetCapacityAvailable, task_dispatcher.cc:283-313.                          */
/* The reference mixes size_t and int64_t (":308-312"): the size_t            */
/* subtraction wraps, is reinterpreted as int64_t and clamped at 0. Restated  */
/* here with signed 64-bit arithmetic, which is the same for all wire values  */
/* (u32 fields, api/scheduler.proto:76-97).                                   */
/* ------------------------------------------------------------------------- */
uint64_t oracle_capacity_available(uint32_t num_processors, uint32_t current_load,
                                   uint32_t max_tasks, uint64_t total_memory,
                                   uint64_t memory_available, uint64_t running,
                                   uint64_t min_memory_for_new_task) {
  if (total_memory != 0 && memory_available < min_memory_for_new_task) {
    return running; /* :286-292: low memory => capacity == running => never free */
  }
  int64_t foreign_load = (int64_t)current_load - (int64_t)running; /* :308-309 */
  if (foreign_load < 0) foreign_load = 0;
  int64_t capacity = (int64_t)num_processors - foreign_load; /* :310-311 */
  if (capacity < 0) capacity = 0;
  return (uint64_t)capacity < (uint64_t)max_tasks ? (uint64_t)capacity
                                                  : (uint64_t)max_tasks; /* :312 */
}

/* TryParseSize, yadcc/common/parse_size.cc:25-45 (suffix G/M/K/B, whole-string number). */
int oracle_try_parse_size(const char* s, uint64_t* out) {
  size_t len = strlen(s);
  if (len == 0) return -1;
  uint64_t scale = 1;
  char last = s[len - 1];
  if (last == 'G') { scale = 1ull << 30; --len; }
  else if (last == 'M') { scale = 1ull << 20; --len; }
  else if (last == 'K') { scale = 1ull << 10; --len; }
  else if (last == 'B') { --len; }
  if (len == 0) return -1;
  uint64_t v = 0;
  for (size_t i = 0; i < len; ++i) {
    if (s[i] < '0' || s[i] > '9') return -1;
    v = v * 10 + (uint64_t)(s[i] - '0');
  }
  *out = v * scale;
  return 0;
}

static int servant_has_env(const oracle_servants* sv, size_t s, uint32_t env_id) {
  /* ContainsEnvironmentSlow, :55-63, on interned digests. */
  uint32_t words = sv->env_words ? sv->env_words : 1;
  return env_id < 64 * words && ((sv->env_mask[s * words + (env_id >> 6)] >> (env_id & 63)) & 1u);
}

/* ------------------------------------------------------------------------- */
/* Literal restatement of WaitForStartingNewTask, one call per task.          */
/* ------------------------------------------------------------------------- */
size_t oracle_dispatch_scan(const oracle_servants* sv, uint64_t min_mem, const oracle_tasks* tk,
                            uint32_t* running, uint32_t* out_idx, double* out_util) {
  size_t S = sv->n, granted = 0;
  for (size_t t = 0; t < tk->n; ++t) {
    uint32_t env = tk->env_id[t], minv = tk->min_version[t], rip = tk->requestor_ip[t];
    int any_eligible = 0, any_free = 0;
    /* UnsafePickServantFor (:362-397) state. */
    size_t self = (size_t)-1;
    size_t best_ded = (size_t)-1, best_any = (size_t)-1;
    double util_ded = DBL_MAX, util_any = DBL_MAX; /* :424 */
    for (size_t s = 0; s < S; ++s) {
      /* UnsafeEnumerateEligibleServants, :316-344. */
      if (!servant_has_env(sv, s, env)) continue;  /* :326-329 */
      if (sv->max_tasks[s] == 0) continue;         /* :330-332 */
      if (sv->version[s] < minv) continue;         /* :333 (int vs uint32 => unsigned) */
      any_eligible = 1;
      /* UnsafeEnumerateFreeServants, :346-360. */
      uint64_t r = running[s];
      uint64_t cap = oracle_capacity_available(sv->num_processors[s], sv->current_load[s],
                                               sv->max_tasks[s], sv->total_memory[s],
                                               sv->memory_available[s], r, min_mem);
      if (r >= cap) continue; /* :353 */
      any_free = 1;
      /* :372-379: the FIRST free servant on the requestor's host is `self`
       * and is taken out of the candidate list. */
      if (self == (size_t)-1 && sv->ip[s] == rip) {
        self = s;
        continue;
      }
      double util = (double)r / (double)cap; /* :440-441 */
      /* :399-410 dedicated predicate. */
      if (sv->priority[s] == ORACLE_PRIORITY_DEDICATED && r * 2 < sv->num_processors[s]) {
        if (best_ded == (size_t)-1 || util < util_ded) { /* :444: strict <, first wins */
          util_ded = util;
          best_ded = s;
        }
      }
      if (best_any == (size_t)-1 || util < util_any) {
        util_any = util;
        best_any = s;
      }
    }
    if (!any_eligible) { /* :105-108 */
      out_idx[t] = ORACLE_IDX_ENV_NOT_FOUND;
      if (out_util) out_util[t] = -1.0;
      continue;
    }
    if (!any_free) { /* :116-118 with timeout == now */
      out_idx[t] = ORACLE_IDX_TIMEOUT;
      if (out_util) out_util[t] = -1.0;
      continue;
    }
    size_t pick;
    double util;
    if (best_ded != (size_t)-1) { pick = best_ded; util = util_ded; }      /* :383-385 */
    else if (best_any != (size_t)-1) { pick = best_any; util = util_any; } /* :389-391 */
    else { /* :392-396: only the requestor itself is free */
      pick = self;
      uint64_t cap = oracle_capacity_available(
          sv->num_processors[pick], sv->current_load[pick], sv->max_tasks[pick],
          sv->total_memory[pick], sv->memory_available[pick], running[pick], min_mem);
      util = (double)running[pick] / (double)cap;
    }
    running[pick] += 1; /* :123 */
    out_idx[t] = (uint32_t)pick;
    if (out_util) out_util[t] = util;
    ++granted;
  }
  return granted;
}

/* ------------------------------------------------------------------------- */
/* Slot-order formulation.                        */
/* ------------------------------------------------------------------------- */
typedef struct slot {
  uint32_t tier;    /* 0: DEDICATED and running*2 < nproc (:399-410), else 1 */
  double util;      /* double(running)/capacity at that running (:440-441) */
  uint32_t servant; /* registry index: first-in-registry wins ties (:444) */
  uint32_t running; /* the `running_tasks` value this slot is taken at */
} slot;

static int slot_cmp(const void* a, const void* b) {
  const slot* x = (const slot*)a;
  const slot* y = (const slot*)b;
  if (x->tier != y->tier) return x->tier < y->tier ? -1 : 1;
  if (x->util != y->util) return x->util < y->util ? -1 : 1;
  if (x->servant != y->servant) return x->servant < y->servant ? -1 : 1;
  return x->running < y->running ? -1 : (x->running > y->running);
}

typedef struct ipent {
  uint32_t ip, servant;
} ipent;
static int ipent_cmp(const void* a, const void* b) {
  const ipent* x = (const ipent*)a;
  const ipent* y = (const ipent*)b;
  if (x->ip != y->ip) return x->ip < y->ip ? -1 : 1;
  return x->servant < y->servant ? -1 : (x->servant > y->servant);
}

size_t oracle_dispatch_sorted(const oracle_servants* sv, uint64_t min_mem, const oracle_tasks* tk,
                              uint32_t* running, uint32_t* out_idx, double* out_util) {
  size_t S = sv->n, N = tk->n;
  if (sv->env_words > 1) return oracle_dispatch_scan(sv, min_mem, tk, running, out_idx, out_util);
  /* 1. Per-servant slot runs. A servant is free at r iff r < cap(r); cap grows
   * by at most one per extra running task, so "not free" is absorbing and a
   * servant's slots are r0, r0+1, ... until the first r with r >= cap(r). */
  size_t* first = (size_t*)malloc((S + 1) * sizeof(size_t));
  size_t M = 0;
  for (size_t s = 0; s < S; ++s) {
    first[s] = M;
    if (sv->max_tasks[s] == 0) continue; /* never eligible (:330-332) */
    uint64_t r = running[s];
    for (;;) {
      uint64_t cap = oracle_capacity_available(sv->num_processors[s], sv->current_load[s],
                                               sv->max_tasks[s], sv->total_memory[s],
                                               sv->memory_available[s], r, min_mem);
      if (r >= cap) break;
      ++M;
      ++r;
    }
  }
  first[S] = M;
  slot* slots = (slot*)malloc((M ? M : 1) * sizeof(slot));
  for (size_t s = 0; s < S; ++s) {
    uint64_t r = running[s];
    for (size_t i = first[s]; i < first[s + 1]; ++i, ++r) {
      uint64_t cap = oracle_capacity_available(sv->num_processors[s], sv->current_load[s],
                                               sv->max_tasks[s], sv->total_memory[s],
                                               sv->memory_available[s], r, min_mem);
      slots[i].tier =
          (sv->priority[s] == ORACLE_PRIORITY_DEDICATED && r * 2 < sv->num_processors[s]) ? 0 : 1;
      slots[i].util = (double)r / (double)cap;
      slots[i].servant = (uint32_t)s;
      slots[i].running = (uint32_t)r;
    }
  }
  qsort(slots, M, sizeof(slot), slot_cmp);

  /* 2. Servant classes by (env_mask, version); a task is compatible with a
   * class as a whole (:326-335). Only servants with max_tasks != 0 count. */
  uint32_t* cls_of = (uint32_t*)malloc((S ? S : 1) * sizeof(uint32_t));
  uint64_t* cls_mask = (uint64_t*)malloc((S ? S : 1) * sizeof(uint64_t));
  uint32_t* cls_ver = (uint32_t*)malloc((S ? S : 1) * sizeof(uint32_t));
  size_t C = 0;
  for (size_t s = 0; s < S; ++s) {
    cls_of[s] = UINT32_MAX;
    if (sv->max_tasks[s] == 0) continue;
    size_t c = 0;
    for (; c < C; ++c)
      if (cls_mask[c] == sv->env_mask[s] && cls_ver[c] == sv->version[s]) break;
    if (c == C) {
      cls_mask[C] = sv->env_mask[s];
      cls_ver[C] = sv->version[s];
      ++C;
    }
    cls_of[s] = (uint32_t)c;
  }
  /* Per-class list of global slot positions, ascending. */
  size_t* cls_begin = (size_t*)calloc(C + 2, sizeof(size_t));
  for (size_t p = 0; p < M; ++p) cls_begin[cls_of[slots[p].servant] + 2]++;
  for (size_t c = 0; c < C; ++c) cls_begin[c + 2] += cls_begin[c + 1];
  uint32_t* cls_list = (uint32_t*)malloc((M ? M : 1) * sizeof(uint32_t));
  for (size_t p = 0; p < M; ++p) cls_list[cls_begin[cls_of[slots[p].servant] + 1]++] = (uint32_t)p;
  /* now cls_begin[c] .. cls_begin[c+1] is class c's range */
  size_t* cursor = (size_t*)malloc((C ? C : 1) * sizeof(size_t));
  for (size_t c = 0; c < C; ++c) cursor[c] = cls_begin[c];
  uint8_t* consumed = (uint8_t*)calloc(M ? M : 1, 1);
  uint32_t* left = (uint32_t*)malloc((S ? S : 1) * sizeof(uint32_t)); /* unconsumed slots */
  for (size_t s = 0; s < S; ++s) left[s] = (uint32_t)(first[s + 1] - first[s]);

  /* Servants by ip for the `self` rule. */
  ipent* byip = (ipent*)malloc((S ? S : 1) * sizeof(ipent));
  for (size_t s = 0; s < S; ++s) {
    byip[s].ip = sv->ip[s];
    byip[s].servant = (uint32_t)s;
  }
  qsort(byip, S, sizeof(ipent), ipent_cmp);

  size_t granted = 0;
  for (size_t t = 0; t < N; ++t) {
    uint32_t env = tk->env_id[t], minv = tk->min_version[t], rip = tk->requestor_ip[t];
    /* Compatible classes. */
    int any_eligible = 0;
    /* self: first servant in registry order that is eligible, free and on the
     * requestor's host (:372-379 operates on the free list). */
    uint32_t self = UINT32_MAX;
    {
      size_t lo = 0, hi = S;
      while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (byip[mid].ip < rip) lo = mid + 1; else hi = mid;
      }
      for (size_t i = lo; i < S && byip[i].ip == rip; ++i) {
        uint32_t s = byip[i].servant;
        if (cls_of[s] == UINT32_MAX || left[s] == 0) continue;
        uint32_t c = cls_of[s];
        if (!(env < 64 && ((cls_mask[c] >> env) & 1u)) || cls_ver[c] < minv) continue;
        self = s;
        break;
      }
    }
    size_t best = (size_t)-1; /* smallest global position among candidates */
    for (size_t c = 0; c < C; ++c) {
      if (!(env < 64 && ((cls_mask[c] >> env) & 1u))) continue;
      if (cls_ver[c] < minv) continue;
      any_eligible = 1;
      /* First unconsumed slot of the class that does not belong to self. */
      while (cursor[c] < cls_begin[c + 1] && consumed[cls_list[cursor[c]]]) ++cursor[c];
      for (size_t i = cursor[c]; i < cls_begin[c + 1]; ++i) {
        uint32_t p = cls_list[i];
        if (consumed[p] || slots[p].servant == self) continue;
        if (p < best) best = p;
        break;
      }
    }
    if (!any_eligible) {
      out_idx[t] = ORACLE_IDX_ENV_NOT_FOUND;
      if (out_util) out_util[t] = -1.0;
      continue;
    }
    if (best == (size_t)-1 && self != UINT32_MAX) {
      /* :392-396: nothing but the requestor itself; take its next slot. */
      size_t c = cls_of[self];
      for (size_t i = cursor[c]; i < cls_begin[c + 1]; ++i) {
        uint32_t p = cls_list[i];
        if (!consumed[p] && slots[p].servant == self) { best = p; break; }
      }
    }
    if (best == (size_t)-1) {
      out_idx[t] = ORACLE_IDX_TIMEOUT;
      if (out_util) out_util[t] = -1.0;
      continue;
    }
    consumed[best] = 1;
    uint32_t s = slots[best].servant;
    left[s]--;
    running[s]++;
    out_idx[t] = s;
    if (out_util) out_util[t] = slots[best].util;
    ++granted;
  }
  free(first); free(slots); free(cls_of); free(cls_mask); free(cls_ver); free(cls_begin);
  free(cls_list); free(cursor); free(consumed); free(left); free(byip);
  return granted;
}

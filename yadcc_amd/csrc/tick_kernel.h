// tick_kernel.h — the small-batch path: ONE launch places a handful of requests.
//
// The reference's real call shape is one WaitForStartingTask RPC asking for `waiters + 1` grants
// (daemon/local/task_grant_keeper.cc:145-146; the loop at scheduler_service_impl.cc:234-264): one
// to a few dozen requests against the whole registry. The batch pipeline (slot generation, sort,
// speculative merge: four launches and more) costs 70-150 us whatever the batch holds; for a
// handful of requests the reference's own arg-min loop (task_dispatcher.cc:362-451) is the right
// algorithm — run by one workgroup:
//
//   * the registry's columns are read ONCE, coalesced, servant s by thread s mod 1024, and turned
//     into one 64-bit key per servant in registers: (tier << 63) | bits(double(running) / capacity)
//     — the reference's own double (task_dispatcher.cc:440-441), positive doubles order like their
//     bit patterns, tier 0 while DEDICATED and 2 * running < nproc (:399-410); ~0 = not free
//     (GetCapacityAvailable, :283-313, in the closed form of dispatch_core.h);
//   * eligibility (:316-344) comes from the servant's class — (environment set, version), as in
//     the batch pipeline — through one bit mask per request, built by the workgroup in LDS for 64
//     requests at a time (ballot per 64 classes);
//   * a pick is a min-reduction of (key, registry index) over the eligible free servants — first
//     wins on ties, the reference's strict `<` (:440-447) — six DPP steps per wave, one LDS hop
//     across the 16 waves, one barrier; the requestor's own servant (`self` = the first eligible
//     free servant on its host, :372-379) is left out of it and is the last resort (:392-396);
//     the winner's thread does `++running_tasks` (:123) and recomputes that one key;
//   * heartbeat rows that change no structure (load, capacities) and released grants (FreeTask's
//     `--running_tasks`, :181) ride in the same launch and are applied first; requests, deltas and
//     results of a typical call travel as kernel arguments / plain stores to page-locked host
//     memory — no copy command, no second launch, the host spins on a stamp the kernel stores last.
//
// Bit-exact with the batch pipeline and the oracle by construction (it IS the per-request scan);
// tests/test_tick_gpu.py forces whole test pools through it.
#ifndef YADCC_AMD_TICK_KERNEL_H_
#define YADCC_AMD_TICK_KERNEL_H_

#include "dispatch_core.h"
#include "kernels.h"

namespace ydc {

constexpr uint32_t kTickThreads = 1024;
constexpr uint32_t kTickWaves = kTickThreads / 64;
constexpr uint32_t kTickBlock = 64;          // requests staged (class masks built) at a time
constexpr uint32_t kTickInlineTasks = 64;    // requests that travel as kernel arguments
constexpr uint32_t kTickInlineUpd = 16;      // heartbeat rows ...
constexpr uint32_t kTickInlineRel = 64;      // released grants ...
constexpr uint32_t kTickMaxPerThread = 16;   // servants a thread keeps in registers
constexpr uint32_t kTickMaxServants = kTickThreads * kTickMaxPerThread;
constexpr uint32_t kTickMaxClasses = 2048;   // 64 requests x 32 mask words of LDS
constexpr uint64_t kTickNoKey = ~0ull;

// The columns of a heartbeat that changes no structure (KeepServantAlive, task_dispatcher.cc:195-201).
struct TickRow {
  uint32_t nproc, load, max_tasks, flags;
};

// Page-locked and coherent: the kernel's last stores; the host spins on `seq`.
struct TickDone {
  uint32_t seq, granted, timeouts, env_not_found;
};

struct TickArgs {
  // resident registry
  uint32_t *nproc, *load, *max_tasks, *flags;  // (written by the heartbeat rows of this tick)
  const uint32_t *class_of, *ip;
  uint32_t* running;  // resident column: released grants are taken off it; COMMIT adds the grants
  uint32_t* rw;       // the column the picks work on: == running with COMMIT, a scratch copy without
  uint32_t* run_out;  // nullable: running_tasks after the batch
  const uint64_t* cls_env;
  const uint32_t* cls_ver;
  uint32_t S, C, EW, W;  // servants, classes, mask words per class / per request
  // requests: the three columns (device or mapped host addresses), or NULL = the inline copies
  const uint32_t *t_env, *t_minv, *t_rip;
  uint32_t n_tasks;
  // deltas (NULL = inline)
  const uint32_t* upd_idx;
  const TickRow* upd_rows;
  const uint32_t* rel;
  uint32_t n_upd, n_rel;
  // results
  uint32_t* out_idx;
  double* out_util;  // nullable
  TickDone* done;
  uint32_t seq;
  uint32_t in_env[kTickInlineTasks], in_minv[kTickInlineTasks], in_rip[kTickInlineTasks];
  uint32_t in_upd_idx[kTickInlineUpd];
  TickRow in_upd[kTickInlineUpd];
  uint32_t in_rel[kTickInlineRel];
};

// What travels through a reduction: the best candidate that is not on the requestor's host, and
// the two lowest registry indexes among the candidates that are.
struct TickCand {
  uint32_t khi, klo, idx, own1, own2;
};

__device__ __forceinline__ void tick_merge(TickCand& a, const TickCand& b) {
  const uint64_t ka = ((uint64_t)a.khi << 32) | a.klo, kb = ((uint64_t)b.khi << 32) | b.klo;
  const bool take = kb < ka || (kb == ka && b.idx < a.idx);
  a.khi = take ? b.khi : a.khi;
  a.klo = take ? b.klo : a.klo;
  a.idx = take ? b.idx : a.idx;
  const uint32_t lo = min(a.own1, b.own1), hi = max(a.own1, b.own1);
  a.own2 = min(hi, min(a.own2, b.own2));
  a.own1 = lo;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void tick_dpp_step(TickCand& c) {
  TickCand o;
  o.khi = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, c.khi);
  o.klo = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, c.klo);
  o.idx = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, c.idx);
  o.own1 = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, c.own1);
  o.own2 = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, c.own2);
  tick_merge(c, o);
}

// Lane 15 of every row ends up with its row's result (disjoint ranges: nothing is merged twice,
// which the "two lowest" part relies on).
__device__ __forceinline__ void tick_row_reduce(TickCand& c) {
  tick_dpp_step<0x111, 0xf>(c);  // row_shr:1
  tick_dpp_step<0x112, 0xf>(c);  // row_shr:2
  tick_dpp_step<0x114, 0xf>(c);  // row_shr:4
  tick_dpp_step<0x118, 0xf>(c);  // row_shr:8
}

__device__ __forceinline__ TickCand tick_readlane(const TickCand& c, int lane) {
  TickCand r;
  r.khi = (uint32_t)__builtin_amdgcn_readlane((int)c.khi, lane);
  r.klo = (uint32_t)__builtin_amdgcn_readlane((int)c.klo, lane);
  r.idx = (uint32_t)__builtin_amdgcn_readlane((int)c.idx, lane);
  r.own1 = (uint32_t)__builtin_amdgcn_readlane((int)c.own1, lane);
  r.own2 = (uint32_t)__builtin_amdgcn_readlane((int)c.own2, lane);
  return r;
}

// The workgroup's result, the same in every thread. `part` = 5 x 16 words of LDS that nobody else
// touches until two reductions later (the caller alternates between two of them).
__device__ __forceinline__ TickCand tick_block_reduce(TickCand c, uint32_t* part) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  tick_row_reduce(c);
  tick_dpp_step<0x142, 0xa>(c);  // row_bcast:15
  tick_dpp_step<0x143, 0xc>(c);  // row_bcast:31
  if (lane == 63) {
    part[wave] = c.khi;
    part[16 + wave] = c.klo;
    part[32 + wave] = c.idx;
    part[48 + wave] = c.own1;
    part[64 + wave] = c.own2;
  }
  __syncthreads();
  TickCand w{0xFFFFFFFFu, 0xFFFFFFFFu, kNone, kNone, kNone};
  if (lane < kTickWaves) {
    w.khi = part[lane];
    w.klo = part[16 + lane];
    w.idx = part[32 + lane];
    w.own1 = part[48 + lane];
    w.own2 = part[64 + lane];
  }
  tick_row_reduce(w);
  return tick_readlane(w, 15);
}

// Key of servant state (dispatch_core.h closed forms; the fp64 key is the reference's own compare).
__device__ __forceinline__ uint64_t tick_key(uint32_t nproc, uint32_t load, uint32_t max_tasks,
                                             uint32_t flags, uint32_t r, bool in_class) {
  if (!in_class || servant_slot_count(nproc, load, max_tasks, r, flags) == 0) return kTickNoKey;
  return slot_key_fp64(slot_tier(nproc, flags, r), r, slot_capacity(nproc, load, max_tasks, r));
}

// K: servants per thread (servant k * 1024 + t is slot k of thread t). COLD: the columns a key is
// recomputed from stay in registers too (K <= 4); otherwise the winner's thread reads them again.
template <int K, bool COLD>
__global__ __launch_bounds__(kTickThreads) void k_tick(const TickArgs a) {
  extern __shared__ uint64_t s_mask[];  // [kTickBlock][W]
  __shared__ uint32_t s_env[kTickBlock], s_minv[kTickBlock], s_rip[kTickBlock], s_out[kTickBlock];
  __shared__ uint32_t s_any[kTickBlock];
  __shared__ double s_util[kTickBlock];
  __shared__ uint32_t s_part[2][80];
  const uint32_t t = threadIdx.x, lane = t & 63;
  const uint32_t S = a.S, W = a.W;

  // ---- registry deltas first: heartbeat rows, released grants ----
  if (a.n_upd | a.n_rel) {
    for (uint32_t u = t; u < a.n_upd; u += kTickThreads) {
      const uint32_t s = a.upd_idx ? a.upd_idx[u] : a.in_upd_idx[u];
      const TickRow r = a.upd_rows ? a.upd_rows[u] : a.in_upd[u];
      if (s < S) {
        a.nproc[s] = r.nproc;
        a.load[s] = r.load;
        a.max_tasks[s] = r.max_tasks;
        a.flags[s] = r.flags;
      }
    }
    for (uint32_t j = t; j < a.n_rel; j += kTickThreads) {
      const uint32_t s = a.rel ? a.rel[j] : a.in_rel[j];
      if (s < S) atomicSub(&a.running[s], 1u);
    }
    __syncthreads();  // (stores and atomics have reached the L2 this workgroup reads from)
  }

  // ---- the registry into registers ----
  uint64_t key[K];
  uint32_t cls[K], ip[K];
  uint32_t c_nproc[COLD ? K : 1], c_load[COLD ? K : 1], c_maxt[COLD ? K : 1], c_flags[COLD ? K : 1],
      c_run[COLD ? K : 1];
  const bool copy_run = a.rw != a.running;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t s = (uint32_t)k * kTickThreads + t;
    key[k] = kTickNoKey;
    cls[k] = 0;
    ip[k] = 0;
    if (COLD) c_nproc[k] = c_load[k] = c_maxt[k] = c_flags[k] = c_run[k] = 0;
    if (s < S) {
      const uint32_t np = a.nproc[s], ld = a.load[s], mt = a.max_tasks[s], fl = a.flags[s],
                     r = a.running[s], co = a.class_of[s];
      ip[k] = a.ip[s];
      cls[k] = co == kNone ? 0u : co;
      key[k] = tick_key(np, ld, mt, fl, r, co != kNone);
      if (copy_run) a.rw[s] = r;
      if (COLD) {
        c_nproc[k] = np;
        c_load[k] = ld;
        c_maxt[k] = mt;
        c_flags[k] = fl | (co != kNone ? 0x80000000u : 0u);
        c_run[k] = r;
      }
    }
  }

  uint32_t n_granted = 0, n_timeout = 0, n_envnf = 0;  // (thread 0's are reported)
  uint32_t red = 0;                                     // reductions so far (buffer parity)

  for (uint32_t base = 0; base < a.n_tasks; base += kTickBlock) {
    const uint32_t nb = min(kTickBlock, a.n_tasks - base);
    if (t < nb) {
      s_env[t] = a.t_env ? a.t_env[base + t] : a.in_env[base + t];
      s_minv[t] = a.t_env ? a.t_minv[base + t] : a.in_minv[base + t];
      s_rip[t] = a.t_env ? a.t_rip[base + t] : a.in_rip[base + t];
    }
    __syncthreads();
    // Class masks of the block: UnsafeEnumerateEligibleServants per class (task_dispatcher.cc:324-338).
    {
      const uint32_t cpad = W * 64, total = nb * cpad;
      for (uint32_t j = t; j < total; j += kTickThreads) {  // (whole waves: total is a multiple of 64)
        const uint32_t i = j / cpad, c = j - i * cpad;
        const uint32_t env = s_env[i];
        bool bit = false;
        if (c < a.C && env < 64 * a.EW)
          bit = ((a.cls_env[(size_t)c * a.EW + (env >> 6)] >> (env & 63)) & 1u) && a.cls_ver[c] >= s_minv[i];
        const uint64_t word = __ballot(bit);
        if (lane == 0) s_mask[i * W + (c >> 6)] = word;
      }
    }
    __syncthreads();
    if (t < nb) {
      uint64_t any = 0;
      for (uint32_t w = 0; w < W; ++w) any |= s_mask[t * W + w];
      s_any[t] = any != 0;
    }
    __syncthreads();

    for (uint32_t i = 0; i < nb; ++i) {
      if (!s_any[i]) {  // nobody advertises the environment at that version: :105-108
        if (t == 0) {
          s_out[i] = kIdxEnvNotFound;
          s_util[i] = -1.0;
        }
        ++n_envnf;
        continue;
      }
      const uint32_t rip = s_rip[i];
      const uint64_t* mrow = s_mask + i * W;
      const uint64_t m0 = mrow[0];
      TickCand mine{0xFFFFFFFFu, 0xFFFFFFFFu, kNone, kNone, kNone};
      uint64_t bk = kTickNoKey;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (key[k] == kTickNoKey) continue;
        const uint64_t m = W == 1 ? m0 : mrow[cls[k] >> 6];
        if (!((m >> (cls[k] & 63)) & 1u)) continue;
        const uint32_t s = (uint32_t)k * kTickThreads + t;
        if (ip[k] == rip) {  // a candidate on the requestor's own host (ascending s: first, second)
          if (mine.own1 == kNone) mine.own1 = s;
          else if (mine.own2 == kNone) mine.own2 = s;
        } else if (key[k] < bk) {
          bk = key[k];
          mine.idx = s;
        }
      }
      mine.khi = (uint32_t)(bk >> 32);
      mine.klo = (uint32_t)bk;
      TickCand best = tick_block_reduce(mine, s_part[red++ & 1]);
      if (best.own2 != kNone) {
        // Several eligible free servants on the requestor's host: only the first of them is `self`
        // (:372-379), the others compete like everybody else.
        TickCand again{0xFFFFFFFFu, 0xFFFFFFFFu, kNone, kNone, kNone};
        bk = kTickNoKey;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if (key[k] == kTickNoKey) continue;
          const uint64_t m = W == 1 ? m0 : mrow[cls[k] >> 6];
          if (!((m >> (cls[k] & 63)) & 1u)) continue;
          const uint32_t s = (uint32_t)k * kTickThreads + t;
          if (s != best.own1 && key[k] < bk) {
            bk = key[k];
            again.idx = s;
          }
        }
        again.khi = (uint32_t)(bk >> 32);
        again.klo = (uint32_t)bk;
        const TickCand b2 = tick_block_reduce(again, s_part[red++ & 1]);
        best.khi = b2.khi;
        best.klo = b2.klo;
        best.idx = b2.idx;
      }
      const uint32_t winner = best.idx != kNone ? best.idx : best.own1;  // :392-396
      if (winner == kNone) {  // eligible servants exist, none is free: Timeout with timeout == now (:116-118)
        if (t == 0) {
          s_out[i] = kIdxTimeout;
          s_util[i] = -1.0;
        }
        ++n_timeout;
        continue;
      }
      ++n_granted;
      if ((winner & (kTickThreads - 1)) == t) {
        const uint32_t wk = winner / kTickThreads;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if ((uint32_t)k != wk) continue;
          s_out[i] = winner;
          s_util[i] = __longlong_as_double((long long)(key[k] & 0x7FFFFFFFFFFFFFFFull));
          uint32_t np, ld, mt, fl, r;
          if (COLD) {
            np = c_nproc[k];
            ld = c_load[k];
            mt = c_maxt[k];
            fl = c_flags[k];
            r = c_run[k] + 1;
            c_run[k] = r;
          } else {
            np = a.nproc[winner];
            ld = a.load[winner];
            mt = a.max_tasks[winner];
            fl = a.flags[winner];
            r = a.rw[winner] + 1;
          }
          a.rw[winner] = r;  // ++pick->running_tasks (:123)
          key[k] = tick_key(np, ld, mt, fl & 3u, r, true);
        }
      }
    }
    __syncthreads();
    if (t < nb) {
      a.out_idx[base + t] = s_out[t];
      if (a.out_util) a.out_util[base + t] = s_util[t];
    }
    // (the next block's staging writes s_env / s_mask: every read of this block is behind the barrier above)
  }

  if (a.run_out) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t s = (uint32_t)k * kTickThreads + t;
      if (s < S) a.run_out[s] = COLD ? c_run[k] : a.rw[s];
    }
  }
  // Results first, then the stamp the host spins on.
  __threadfence_system();
  __syncthreads();
  if (t == 0) {
    a.done->granted = n_granted;
    a.done->timeouts = n_timeout;
    a.done->env_not_found = n_envnf;
    __hip_atomic_store(&a.done->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace ydc
#endif  // YADCC_AMD_TICK_KERNEL_H_

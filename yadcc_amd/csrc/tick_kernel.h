// tick_kernel.h — the small-batch path: ONE workgroup places a handful of requests.
//
// The reference's real call shape is one WaitForStartingTask RPC asking for `waiters + 1` grants
// (daemon/local/task_grant_keeper.cc:145-146; the loop at scheduler_service_impl.cc:234-264): one
// to a few dozen requests against the whole registry. The batch pipeline (slot generation, sort,
// speculative merge: four launches and more) costs 70-150 us whatever the batch holds; for a
// handful of requests the reference's own arg-min loop (task_dispatcher.cc:362-451) is the right
// algorithm — run by one workgroup, and, between calls, KEPT on its CU:
//
//   * the registry's columns are read ONCE, coalesced, servant s by thread s mod THREADS, and turned
//     into one 64-bit key per servant in registers: (tier << 63) | bits(double(running) / capacity)
//     — the reference's own double (task_dispatcher.cc:440-441), positive doubles order like their
//     bit patterns, tier 0 while DEDICATED and 2 * running < nproc (:399-410); ~0 = not free
//     (GetCapacityAvailable, :283-313, in the closed form of dispatch_core.h). As few waves as hold
//     the registry (256 threads up to 4096 servants): a pick costs a reduction, and every wave of a
//     SIMD pays for it again;
//   * eligibility (:316-344) comes from the servant's class — (environment set, version), as in
//     the batch pipeline — through a bit mask of the eligible classes per (digest, version
//     threshold), built by the workgroup in LDS (ballot per 64 classes) when that pair changes;
//     the servants' classes and hosts lie in LDS, read only then;
//   * a pick is a min-reduction of (key, registry index) over the eligible free servants — first
//     wins on ties, the reference's strict `<` (:440-447) — six DPP steps per wave, one LDS hop
//     across the waves, one LDS-only barrier. Every thread CACHES its best candidate: only the
//     thread whose servant was picked (or was touched by a heartbeat / a released grant) looks at
//     its servants again. The requestor's own servant (`self` = the first eligible free servant
//     on its host, :372-379) is left out and is the last resort (:392-396): a flag says whether
//     any thread holds such a candidate at all, and only then a second reduction finds it;
//   * the requests of one RPC are copies of one another: their n picks are ONE merge — every
//     thread offers its next two candidates, wave 0 merges the sorted lists (a wave-wide reduction
//     per pick, no workgroup barrier), rounds until all are placed;
//   * heartbeat rows that change no structure (load, capacities) and released grants (FreeTask's
//     `--running_tasks`, :181) ride with the requests: the thread that holds the servant patches
//     its registers; running_tasks and the rows go back to the columns after the answer;
//   * RESIDENT: the kernel of a COMMITting call does not end. It polls a page-locked mailbox
//     (TickBox: eight self-validating 8-byte granules = one 64-byte PCIe read per poll), takes the
//     next call's requests and deltas from it and answers the same way — no launch (7 us launch
//     to first store on this box), no column loads. It leaves when told (every other entry point
//     of the context ends it first) or when nobody has asked for 50 ms.
//
// Bit-exact with the batch pipeline and the oracle by construction (it IS the per-request scan);
// tests/test_tick_gpu.py forces whole test pools through it, launched and resident.
#ifndef YADCC_AMD_TICK_KERNEL_H_
#define YADCC_AMD_TICK_KERNEL_H_

#include "dispatch_core.h"
#include "kernels.h"

namespace ydc {

constexpr uint32_t kTickBlock = 64;          // results are flushed 64 requests at a time
constexpr uint32_t kTickInlineTasks = 64;    // requests that travel as kernel arguments
constexpr uint32_t kTickInlineUpd = 16;      // heartbeat rows ...
constexpr uint32_t kTickInlineRel = 64;      // released grants ...
constexpr uint32_t kTickMaxServants = 16384;  // 512 threads x 32 servants in registers
constexpr uint32_t kTickMaxClasses = 4096;    // eligible-class mask of a request: 64 words of LDS
constexpr uint64_t kTickNoKey = ~0ull;
// Which builds of k_tick carry the merge of identical requests (its lists: 32 bytes of LDS per
// thread, a dozen registers): the 256-thread ones and, with one-word candidates, 512 x 16.
__host__ __device__ constexpr bool tick_merges(int threads, int k, bool packed) {
  return threads <= 256 || (threads == 512 && packed);
}
constexpr uint32_t kTickMergeMin = 3;  // identical requests from which a command is placed as one merge

// The columns of a heartbeat that changes no structure (KeepServantAlive, task_dispatcher.cc:195-201).
struct TickRow {
  uint32_t nproc, load, max_tasks, flags;
};

// Page-locked and coherent: the kernel's last stores; the host spins on `seq`.
struct TickDone {
  uint32_t seq, granted, timeouts, env_not_found;
};

// The mailbox of a resident kernel (page-locked, coherent host memory). The host sends a command by
// storing its payload and then the eight granules {word, command number} of `head`; the kernel's
// wave 0 polls `head` (one 64-byte read per poll) and acts when all eight carry the number it
// waits for. What does not fit the head — requests beyond the first, released grants beyond four,
// heartbeat rows — lies in the arrays, stored before the head and read after it. The answer comes
// back the same way (`reply`, then the arrays for more than seven placements / utilisations).
constexpr uint32_t kTickCmdTick = 1, kTickCmdQuit = 2;
constexpr uint32_t kTickCmdSame = 0x80;  // flag on kTickCmdTick: every request is a copy of the first (one RPC's requests)
struct TickBox {
  // {cmd | n_tasks << 8 | n_upd << 16 | n_rel << 24}, env, minv, rip, rel[0 .. 3],
  // upd_idx[0], upd[0].{nproc, load, max_tasks, flags}, rel[4 .. 6]: two 64-byte lines — a request
  // with up to seven released grants and one heartbeat row needs no second read
  unsigned long long head[16];
  unsigned long long reply[8];  // {granted | timeouts << 8 | env_not_found << 16}, placement 0 .. 6
  uint32_t env[kTickInlineTasks], minv[kTickInlineTasks], rip[kTickInlineTasks];
  uint32_t rel[kTickInlineRel];
  uint32_t upd_idx[kTickInlineUpd];
  TickRow upd[kTickInlineUpd];
  uint32_t out_idx[kTickInlineTasks];
  double out_util[kTickInlineTasks];
  uint32_t alive;  // the kernel clears it when it leaves (QUIT, or nobody asked for idle_ticks)
};

struct TickArgs {
  uint32_t cap_bits, idx_bits;  // the one-word candidate: key format (dispatch_core.h), bits of a registry index
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
  TickBox* box;  // non-NULL: stay resident after this command and take the next ones from the mailbox
  unsigned long long idle_ticks;  // ... until nobody has sent one for this long (100 MHz ticks)
  uint32_t seq;
  uint32_t in_env[kTickInlineTasks], in_minv[kTickInlineTasks], in_rip[kTickInlineTasks];
  uint32_t in_upd_idx[kTickInlineUpd];
  TickRow in_upd[kTickInlineUpd];
  uint32_t in_rel[kTickInlineRel];
};

// What travels through the common reduction: the best free eligible servant that is not on the
// requestor's host — (key, registry index), first wins on equal keys.
struct TickBest {
  uint32_t khi, klo, idx;
};
// ... and through the rare one: the two lowest registry indexes among the free eligible servants
// that ARE on the requestor's host (the first is `self`, task_dispatcher.cc:372-379).
struct TickOwn {
  uint32_t own1, own2;
};

__device__ __forceinline__ void tick_merge(TickBest& a, const TickBest& b) {
  const uint64_t ka = ((uint64_t)a.khi << 32) | a.klo, kb = ((uint64_t)b.khi << 32) | b.klo;
  const bool take = kb < ka || (kb == ka && b.idx < a.idx);
  a.khi = take ? b.khi : a.khi;
  a.klo = take ? b.klo : a.klo;
  a.idx = take ? b.idx : a.idx;
}
__device__ __forceinline__ void tick_merge(TickOwn& a, const TickOwn& b) {
  const uint32_t lo = min(a.own1, b.own1), hi = max(a.own1, b.own1);
  a.own2 = min(hi, min(a.own2, b.own2));
  a.own1 = lo;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void tick_dpp_step(TickBest& c) {
  TickBest o;
  o.khi = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, c.khi);
  o.klo = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, c.klo);
  o.idx = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, c.idx);
  tick_merge(c, o);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void tick_dpp_step(TickOwn& c) {
  TickOwn o;
  o.own1 = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, c.own1);
  o.own2 = dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, c.own2);
  tick_merge(c, o);
}

// Inclusive scan steps inside a row of 16 lanes: lane n - 1 ends up with the result of lanes
// 0 .. n - 1 (disjoint ranges: nothing is merged twice, which the "two lowest" merge relies on).
template <int N, class T>
__device__ __forceinline__ void tick_row_reduce(T& c) {
  if (N > 1) tick_dpp_step<0x111, 0xf>(c);  // row_shr:1
  if (N > 2) tick_dpp_step<0x112, 0xf>(c);  // row_shr:2
  if (N > 4) tick_dpp_step<0x114, 0xf>(c);  // row_shr:4
  if (N > 8) tick_dpp_step<0x118, 0xf>(c);  // row_shr:8
}

// The one-word candidate (capacities below 2^10, which is every realistic pool): the exact integer
// key of dispatch_core.h (slot_key_exact: tier and floor(running * 4^b / capacity), order-isomorphic
// to the reference's double compare — DESIGN.md 2) above the registry index, 32 bits together.
// The minimum of the words IS the pick, first-wins tie-break included: a reduction step is one
// v_min_u32 with a DPP operand where the two-word key needs fourteen instructions.
struct TickWord {
  uint32_t w;
};
__device__ __forceinline__ void tick_merge(TickWord& a, const TickWord& b) { a.w = min(a.w, b.w); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void tick_dpp_step(TickWord& c) {
  c.w = min(c.w, dpp_u32<CTRL, ROW_MASK>(0xFFFFFFFFu, c.w));
}

template <bool PK>
struct TickTypes {
  using Key = uint64_t;
  using Cand = TickBest;
};
template <>
struct TickTypes<true> {
  using Key = uint32_t;
  using Cand = TickWord;
};
__device__ __forceinline__ TickBest tick_cand(uint64_t key, uint32_t s, uint32_t) {
  return TickBest{(uint32_t)(key >> 32), (uint32_t)key, key == kTickNoKey ? kNone : s};
}
__device__ __forceinline__ TickWord tick_cand(uint32_t key, uint32_t s, uint32_t ib) {
  return TickWord{key == 0xFFFFFFFFu ? 0xFFFFFFFFu : (key << ib) | s};
}
__device__ __forceinline__ uint32_t tick_cand_idx(const TickBest& c, uint32_t) { return c.idx; }
__device__ __forceinline__ uint32_t tick_cand_idx(const TickWord& c, uint32_t ib) {
  return c.w == 0xFFFFFFFFu ? kNone : c.w & ((1u << ib) - 1);
}
// (key, registry index) order: is a before b?
__device__ __forceinline__ bool tick_cand_less(const TickBest& a, const TickBest& b) {
  const uint64_t ka = ((uint64_t)a.khi << 32) | a.klo, kb = ((uint64_t)b.khi << 32) | b.klo;
  return ka < kb || (ka == kb && a.idx < b.idx);
}
__device__ __forceinline__ bool tick_cand_less(const TickWord& a, const TickWord& b) { return a.w < b.w; }

// A workgroup barrier that orders LDS traffic only. __syncthreads() also waits for every global
// store in flight (vmcnt(0)) — the winner's `++running_tasks` store of the pick before would be
// waited for by every pick (measured: 2 us per pick instead of 0.4).
__device__ __forceinline__ void tick_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ uint32_t tick_rl(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// The workgroup's result, the same in every thread. `part`: 3 x 16 (2 x 16) words of LDS that
// nobody else touches until two reductions later (the callers alternate between two of them).
template <int WAVES>
__device__ __forceinline__ TickBest tick_block_reduce(TickBest c, uint32_t* part) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  tick_row_reduce<16>(c);
  tick_dpp_step<0x142, 0xa>(c);  // row_bcast:15
  tick_dpp_step<0x143, 0xc>(c);  // row_bcast:31
  if (lane == 63) {
    part[wave] = c.khi;
    part[16 + wave] = c.klo;
    part[32 + wave] = c.idx;
  }
  tick_lds_barrier();
  TickBest w{0xFFFFFFFFu, 0xFFFFFFFFu, kNone};
  if (lane < WAVES) {
    w.khi = part[lane];
    w.klo = part[16 + lane];
    w.idx = part[32 + lane];
  }
  tick_row_reduce<WAVES>(w);
  return TickBest{tick_rl(w.khi, WAVES - 1), tick_rl(w.klo, WAVES - 1), tick_rl(w.idx, WAVES - 1)};
}
template <int WAVES>
__device__ __forceinline__ TickWord tick_block_reduce(TickWord c, uint32_t* part) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  tick_row_reduce<16>(c);
  tick_dpp_step<0x142, 0xa>(c);
  tick_dpp_step<0x143, 0xc>(c);
  if (lane == 63) part[wave] = c.w;
  tick_lds_barrier();
  TickWord w{0xFFFFFFFFu};
  if (lane < WAVES) w.w = part[lane];
  tick_row_reduce<WAVES>(w);
  return TickWord{tick_rl(w.w, WAVES - 1)};
}
template <int WAVES>
__device__ __forceinline__ TickOwn tick_block_reduce(TickOwn c, uint32_t* part) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  tick_row_reduce<16>(c);
  tick_dpp_step<0x142, 0xa>(c);
  tick_dpp_step<0x143, 0xc>(c);
  if (lane == 63) {
    part[wave] = c.own1;
    part[16 + wave] = c.own2;
  }
  tick_lds_barrier();
  TickOwn w{kNone, kNone};
  if (lane < WAVES) {
    w.own1 = part[lane];
    w.own2 = part[16 + lane];
  }
  tick_row_reduce<WAVES>(w);
  return TickOwn{tick_rl(w.own1, WAVES - 1), tick_rl(w.own2, WAVES - 1)};
}

// Key of a servant's state (dispatch_core.h closed forms; the fp64 key is the reference's own compare).
template <bool PK>
__device__ __forceinline__ typename TickTypes<PK>::Key tick_key(uint32_t nproc, uint32_t load, uint32_t max_tasks,
                                                                uint32_t flags, uint32_t r, bool in_class,
                                                                uint32_t cap_bits) {
  using Key = typename TickTypes<PK>::Key;
  if (!in_class || servant_slot_count(nproc, load, max_tasks, r, flags) == 0) return (Key) ~(Key)0;
  const uint32_t cap = slot_capacity(nproc, load, max_tasks, r), tier = slot_tier(nproc, flags, r);
  if (PK) return (Key)slot_key_exact(tier, r, cap, cap_bits);  // (cap_bits <= 10: 32-bit arithmetic)
  return (Key)slot_key_fp64(tier, r, cap);
}

// THREADS x K >= S: servant k * THREADS + t is slot k of thread t. As few waves as hold the
// registry: what a pick costs is the reduction, and every wave of a SIMD pays it again (measured:
// 1024 threads x 2 servants 1.7 us per pick, the issue slots of 4 waves per SIMD). COLD: the
// columns a key is recomputed from stay in registers too; otherwise the owner reads them again.
// No global store and (COLD) no global load sits inside the pick loop: the compiler's wait-count
// bookkeeping makes every pick wait for the stores of the pick before as soon as the loop touches a
// register that was once loaded (measured: 2 us per pick) — running_tasks goes back at the end.
#ifdef YDC_PHASE_PROBE
// Measurement build (`make probe`, tools/tick_probe.py): thread 0 leaves the 100 MHz wall clock at
// the kernel's phase boundaries in ydc_phase_probe[0 .. 31].
#define YDC_TICK_STAMP(slot)                                         \
  do {                                                               \
    if (threadIdx.x == 0) ydc_phase_probe[slot] = wall_clock64();    \
  } while (0)
#else
#define YDC_TICK_STAMP(slot) \
  do {                       \
  } while (0)
#endif

template <int THREADS, int K, bool COLD, bool PK>
__global__ __launch_bounds__(THREADS) void k_tick(const TickArgs a) {
  using KeyT = typename TickTypes<PK>::Key;
  using Cand = typename TickTypes<PK>::Cand;
  constexpr KeyT kNo = (KeyT) ~(KeyT)0;
  constexpr int WAVES = THREADS / 64;
  constexpr int G = K < 8 ? K : 8;  // servants whose columns are in flight together
  extern __shared__ uint64_t s_mask[];  // [W]: eligible classes of the current (digest, version threshold)
  __shared__ uint32_t s_out[kTickBlock];
  __shared__ double s_util[kTickBlock];
  __shared__ uint32_t s_part[2][48], s_part_own[2][32];
  __shared__ uint32_t s_own_flag[3];
  // The payload of a command — kernel arguments for the first, the mailbox for the ones a resident
  // kernel is sent later — staged once: the argument segment may live in host memory, where
  // every scalar load of it is a PCIe round trip.
  __shared__ uint32_t s_env[kTickBlock], s_minv[kTickBlock], s_rip[kTickBlock];
  __shared__ uint32_t s_rel[kTickInlineRel], s_uidx[kTickInlineUpd];
  __shared__ TickRow s_urow[kTickInlineUpd];
  __shared__ uint32_t s_cmd[4];  // resident kernel: {word 0 of the command head, 1 = leave}
  const uint32_t t = threadIdx.x, lane = t & 63;
  YDC_TICK_STAMP(0);
  // Everything the kernel reads from the head of its argument block, fetched in ONE go (the
  // compiler would fetch each field where it is first needed: a dozen dependent round trips).
  uint32_t* const p_nproc = a.nproc;
  uint32_t* const p_load = a.load;
  uint32_t* const p_maxt = a.max_tasks;
  uint32_t* const p_flags = a.flags;
  const uint32_t* const p_class_of = a.class_of;
  const uint32_t* const p_ip = a.ip;
  uint32_t* const p_running = a.running;
  uint32_t* const p_rw = a.rw;
  uint32_t* const p_run_out = a.run_out;
  const uint64_t* const p_cls_env = a.cls_env;
  const uint32_t* const p_cls_ver = a.cls_ver;
  const uint32_t S = a.S, C = a.C, EW = a.EW, W = a.W, cap_bits = a.cap_bits, ib = a.idx_bits;
  uint32_t n_tasks = a.n_tasks, n_upd = a.n_upd, n_rel = a.n_rel;
  const uint32_t *const p_tenv = a.t_env, *const p_tminv = a.t_minv, *const p_trip = a.t_rip;
  const uint32_t* const p_upd_idx = a.upd_idx;
  const TickRow* const p_upd_rows = a.upd_rows;
  const uint32_t* const p_rel = a.rel;
  uint32_t* const p_out_idx = a.out_idx;
  double* const p_out_util = a.out_util;
  TickDone* const p_done = a.done;
  TickBox* const box = a.box;
  const unsigned long long idle_ticks = a.idle_ticks;
  uint32_t seq = a.seq;
  asm volatile("" ::"s"(p_nproc), "s"(p_load), "s"(p_maxt), "s"(p_flags), "s"(p_class_of), "s"(p_ip),
               "s"(p_running), "s"(p_rw), "s"(p_run_out), "s"(p_cls_env), "s"(p_cls_ver), "s"(box), "s"(idle_ticks));
  asm volatile("" ::"s"(cap_bits), "s"(ib));
  asm volatile("" ::"s"(S), "s"(C), "s"(EW), "s"(W), "s"(n_tasks), "s"(n_upd), "s"(n_rel), "s"(p_tenv),
               "s"(p_tminv), "s"(p_trip), "s"(p_upd_idx), "s"(p_upd_rows), "s"(p_rel), "s"(p_out_idx),
               "s"(p_out_util), "s"(p_done), "s"(seq));
  YDC_TICK_STAMP(1);
  if (t < 3) s_own_flag[t] = 0;
  if (t < kTickBlock) {
    if (!p_tenv && t < n_tasks) {
      s_env[t] = a.in_env[t];
      s_minv[t] = a.in_minv[t];
      s_rip[t] = a.in_rip[t];
    }
    if (!p_rel && t < n_rel) s_rel[t] = a.in_rel[t];
    if (!p_upd_idx && t < n_upd) {
      s_uidx[t] = a.in_upd_idx[t];
      s_urow[t] = a.in_upd[t];
    }
  }

  // ---- long delta lists (beyond what travels as arguments): applied to the columns first ----
  if ((p_upd_idx && n_upd) || (p_rel && n_rel)) {
    if (p_upd_idx)
      for (uint32_t u = t; u < n_upd; u += THREADS) {
        const uint32_t s = p_upd_idx[u];
        const TickRow r = p_upd_rows[u];
        if (s < S) {
          p_nproc[s] = r.nproc;
          p_load[s] = r.load;
          p_maxt[s] = r.max_tasks;
          p_flags[s] = r.flags;
        }
      }
    if (p_rel)
      for (uint32_t j = t; j < n_rel; j += THREADS) {
        const uint32_t s = p_rel[j];
        if (s < S) atomicSub(&p_running[s], 1u);
      }
    __syncthreads();  // (stores and atomics have reached the L2 this workgroup reads from)
  }

  // ---- the registry into registers: G servants' columns in flight at a time ----
  // Per servant in registers: the key and running_tasks (and, COLD, the columns a key is made
  // of). Its class and its host are only looked at when the request signature changes: LDS.
  KeyT key[K];
  uint32_t c_run[K];
  uint32_t* const s_ip = (uint32_t*)(s_mask + W + (tick_merges(THREADS, K, PK) ? 4 * THREADS : 0));  // [K * THREADS], behind the merge's lists
  uint16_t* const s_cls = (uint16_t*)(s_ip + K * THREADS);                              // [K * THREADS]
  uint32_t c_nproc[COLD ? K : 1], c_load[COLD ? K : 1], c_maxt[COLD ? K : 1], c_flags[COLD ? K : 1];
  uint32_t in_cls = 0;   // bit k: servant k of this thread accepts tasks at all (max_tasks != 0)
  uint32_t changed = 0;  // bit k: running_tasks of servant k is to be written back
  const bool copy_run = p_rw != p_running;
#pragma unroll
  for (int g = 0; g < K; g += G) {
    uint32_t l_np[G], l_ld[G], l_mt[G], l_fl[G], l_r[G], l_co[G], l_ip[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const uint32_t s = (uint32_t)(g + j) * THREADS + t, sc = s < S ? s : 0;  // (S == 0: nothing is read)
      l_np[j] = l_ld[j] = l_mt[j] = l_fl[j] = l_r[j] = l_ip[j] = 0;
      l_co[j] = kNone;
      if (S) {
        l_np[j] = p_nproc[sc];
        l_ld[j] = p_load[sc];
        l_mt[j] = p_maxt[sc];
        l_fl[j] = p_flags[sc];
        l_r[j] = p_running[sc];
        l_co[j] = p_class_of[sc];
        l_ip[j] = p_ip[sc];
      }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int k = g + j;
      const uint32_t s = (uint32_t)k * THREADS + t;
      const bool live = s < S, inc = live && l_co[j] != kNone;
      s_ip[s] = live ? l_ip[j] : 0u;
      s_cls[s] = (uint16_t)(inc ? l_co[j] : 0u);
      in_cls |= (inc ? 1u : 0u) << k;
      c_run[k] = l_r[j];
      key[k] = tick_key<PK>(l_np[j], l_ld[j], l_mt[j], l_fl[j], l_r[j], inc, cap_bits);
      if (COLD) {
        c_nproc[k] = l_np[j];
        c_load[k] = l_ld[j];
        c_maxt[k] = l_mt[j];
        c_flags[k] = l_fl[j];
      }
      if (copy_run && live) p_rw[s] = l_r[j];
    }
  }
  YDC_TICK_STAMP(2);

  // (the thread index as the command loop sees it: opaque to the compiler once per turn, or it
  // hoists every per-servant index, address and mask out of the loop and spills them — 1.2 KB of
  // scratch per lane in the 16-servants-per-thread kernels)
  uint32_t tl = t;
  // What survives from one command of a resident kernel to the next: the eligible-class mask of
  // the last (digest, version threshold), the host the own-host bits belong to, and every
  // thread's cached candidates — a thread whose servants no command touches never rescans.
  uint32_t red = 0, pk = 0;  // reductions / picks so far (LDS buffer rotation)
  uint32_t p_env = 0, p_minv = 0, p_rip = 0;
  bool have_sig = false, any = false, dirty = true;
  uint32_t elig = 0, ownb = 0;  // bit k: servant k is eligible for / on the host of the signature
  Cand mine = tick_cand(kNo, kNone, ib);
  TickOwn mine_own{kNone, kNone};

  // The request signature the cached values belong to: one RPC's requests share it
  // (scheduler_service_impl.cc:228-264), and digests and version thresholds are few.
  auto set_signature = [&](uint32_t env, uint32_t minv, uint32_t rip) {
    if (!have_sig || env != p_env || minv != p_minv) {
      // Eligible classes: UnsafeEnumerateEligibleServants per class (task_dispatcher.cc:324-338).
      for (uint32_t c = t; c < W * 64; c += THREADS) {  // (whole waves)
        bool bit = false;
        if (c < C && env < 64 * EW)
          bit = ((p_cls_env[(size_t)c * EW + (env >> 6)] >> (env & 63)) & 1u) && p_cls_ver[c] >= minv;
        const uint64_t word = __ballot(bit);
        if (lane == 0) s_mask[c >> 6] = word;
      }
      tick_lds_barrier();
      uint64_t any_w = 0;
      for (uint32_t w = 0; w < W; ++w) any_w |= s_mask[w];
      any = any_w != 0;
      elig = 0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const uint32_t cl = s_cls[(uint32_t)k * THREADS + tl];
        const uint64_t m = s_mask[cl >> 6];
        elig |= (uint32_t)((m >> (cl & 63)) & ((in_cls >> k) & 1u)) << k;
      }
      dirty = true;
      tick_lds_barrier();  // (s_mask is rewritten at the next change)
    }
    if (!have_sig || rip != p_rip) {
      uint32_t nb = 0;
#pragma unroll
      for (int k = 0; k < K; ++k) nb |= (s_ip[(uint32_t)k * THREADS + tl] == rip ? 1u : 0u) << k;
      // (another requestor only matters to the threads that hold a servant of the old or the new host)
      if (nb | ownb) dirty = true;
      ownb = nb;
    }
    have_sig = true;
    p_env = env;
    p_minv = minv;
    p_rip = rip;
  };
  // A thread's next TWO candidates (the merge below): the best of its servants and what would be
  // its best once that one is taken — the runner-up, or the same servant at running + 1 (nk0).
  Cand e1 = tick_cand(kNo, kNone, ib);
  KeyT nk0 = kNo;
  uint32_t k0 = 0, k1 = 0;
  bool e1_same = false, list_valid = false;
  auto build_list = [&]() {
    KeyT b0 = kNo, b1 = kNo;
    uint32_t i0 = kNone, i1 = kNone;
    k0 = k1 = 0;
    mine_own.own1 = mine_own.own2 = kNone;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (!((elig >> k) & 1u) || key[k] == kNo) continue;
      const uint32_t s = (uint32_t)k * THREADS + tl;
      if ((ownb >> k) & 1u) {
        if (mine_own.own1 == kNone) mine_own.own1 = s;
        else if (mine_own.own2 == kNone) mine_own.own2 = s;
      } else if (key[k] < b0) {  // (ascending s: the first of equal keys stays in front)
        b1 = b0;
        i1 = i0;
        k1 = k0;
        b0 = key[k];
        i0 = s;
        k0 = (uint32_t)k;
      } else if (key[k] < b1) {
        b1 = key[k];
        i1 = s;
        k1 = (uint32_t)k;
      }
    }
    mine = tick_cand(b0, i0, ib);
    nk0 = kNo;
    if (i0 != kNone) {
      uint32_t np = 0, ld = 0, mt = 0, fl = 0, run = 0;
      if (!COLD) {
        np = p_nproc[i0];
        ld = p_load[i0];
        mt = p_maxt[i0];
        fl = p_flags[i0];
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const bool me = (uint32_t)k == k0;
        run = me ? c_run[k] : run;
        if (COLD) {
          np = me ? c_nproc[k] : np;
          ld = me ? c_load[k] : ld;
          mt = me ? c_maxt[k] : mt;
          fl = me ? c_flags[k] : fl;
        }
      }
      nk0 = tick_key<PK>(np, ld, mt, fl, run + 1, true, cap_bits);
    }
    e1_same = nk0 < b1 || (nk0 == b1 && i0 < i1);  // (both none: no second candidate either way)
    e1 = tick_cand(e1_same ? nk0 : b1, e1_same ? i0 : i1, ib);
    dirty = false;
    list_valid = true;
  };

  for (;;) {  // one command per turn (a kernel that is not resident takes one turn)
    asm volatile("" : "+v"(tl));
    tick_lds_barrier();  // (the staged command)
    YDC_TICK_STAMP(3);
    // ---- short delta lists: the owner patches its registers ----
    // Heartbeat rows: the columns first (stores only), then the registers.
    if (!p_upd_idx && n_upd) {
      for (uint32_t u = 0; u < n_upd; ++u) {
        const uint32_t s = s_uidx[u];
        if (s < S && s % THREADS == tl) {
          const TickRow r = s_urow[u];
          p_nproc[s] = r.nproc;
          p_load[s] = r.load;
          p_maxt[s] = r.max_tasks;
          p_flags[s] = r.flags;
        }
      }
      for (uint32_t u = 0; u < n_upd; ++u) {
        const uint32_t s = s_uidx[u];
        if (s < S && s % THREADS == tl) {
          const TickRow r = s_urow[u];
          const uint32_t wk = s / THREADS;
          uint32_t run = 0;
#pragma unroll
          for (int k = 0; k < K; ++k) run = (uint32_t)k == wk ? c_run[k] : run;
          const KeyT nk = tick_key<PK>(r.nproc, r.load, r.max_tasks, r.flags, run, (in_cls >> wk) & 1u, cap_bits);
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const bool me = (uint32_t)k == wk;
            key[k] = me ? nk : key[k];
            if (COLD) {
              c_nproc[k] = me ? r.nproc : c_nproc[k];
              c_load[k] = me ? r.load : c_load[k];
              c_maxt[k] = me ? r.max_tasks : c_maxt[k];
              c_flags[k] = me ? r.flags : c_flags[k];
            }
          }
          dirty = true;
        }
      }
    }
    // Released grants: FreeTask's --running_tasks (:181). This thread is the only one that knows servant s.
    if (!p_rel && n_rel) {
      uint32_t touched = 0;
      // (one LDS read per lane, then a lane read per released grant: no LDS round trip in the loop)
      const uint32_t rel_mine = lane < n_rel ? s_rel[lane] : kNone;
      for (uint32_t j = 0; j < n_rel; ++j) {
        const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)rel_mine, (int)j);
        if (s < S && s % THREADS == tl) {
          const uint32_t wk = s / THREADS;
#pragma unroll
          for (int k = 0; k < K; ++k) c_run[k] -= (uint32_t)k == wk ? 1u : 0u;
          touched |= 1u << wk;
        }
      }
      // (one key per touched servant, whatever the number of its released grants)
      if (touched) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if (!((touched >> k) & 1u)) continue;
          const uint32_t s = (uint32_t)k * THREADS + tl;
          uint32_t np, ld, mt, fl;
          if (COLD) {
            np = c_nproc[k];
            ld = c_load[k];
            mt = c_maxt[k];
            fl = c_flags[k];
          } else {  // (after this thread's own row stores above: same thread, same address)
            np = p_nproc[s];
            ld = p_load[s];
            mt = p_maxt[s];
            fl = p_flags[s];
          }
          key[k] = tick_key<PK>(np, ld, mt, fl, c_run[k], (in_cls >> k) & 1u, cap_bits);
        }
        changed |= touched;
        dirty = true;
      }
    }
    YDC_TICK_STAMP(4);

    uint32_t n_granted = 0, n_timeout = 0, n_envnf = 0;  // (thread 0's are reported)
    uint32_t i_start = 0;
    // ---- the requests of one RPC are copies of one another (scheduler_service_impl.cc:228-264):
    // n picks as ONE merge. Every thread offers its next two candidates; the picks of the
    // sequential process are the smallest (key, registry index) entries of the union of these
    // sorted lists, in order — a thread's own servants are picked in the order of its list
    // whatever the other threads do — so wave 0 alone merges them, a wave-wide reduction and no
    // workgroup barrier per pick, until the requests are placed or a list runs dry (then the
    // threads that were picked from make new lists: another round). Requests from a host that
    // runs an eligible free servant (`self`, :372-396) take the pick-by-pick loop below.
    // (256-thread kernels: the wider ones have no registers to spare for it)
    // (the one-word candidate does not hold the utilisation: a caller that wants it gets the loop below)
    if (tick_merges(THREADS, K, PK) && !p_tenv && n_tasks >= kTickMergeMin && n_tasks <= kTickBlock && !(PK && p_out_util)) {
      const bool differs = lane < n_tasks && (s_env[lane] != s_env[0] || s_minv[lane] != s_minv[0] || s_rip[lane] != s_rip[0]);
      if (__ballot(differs) == 0) {
        set_signature((uint32_t)__builtin_amdgcn_readfirstlane((int)s_env[0]),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)s_minv[0]),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)s_rip[0]));
        Cand* const s_lc = (Cand*)(s_mask + W);                      // [2][THREADS] candidates
        uint32_t* const s_cnt = (uint32_t*)(s_mask + W) + 6 * THREADS;  // [THREADS] entries taken this round
        if (!any) {  // :105-108, n times
          if (t < n_tasks) {
            s_out[t] = kIdxEnvNotFound;
            s_util[t] = -1.0;
          }
          n_envnf += n_tasks;
          i_start = n_tasks;
        } else {
          uint32_t placed = 0;
          bool own_seen = false;
          for (uint32_t round = 0; placed < n_tasks; ++round) {
            if (dirty || !list_valid) build_list();
            s_lc[t] = mine;
            s_lc[THREADS + t] = e1;
            const uint32_t fl_i = pk % 3;
            if (round == 0 && __ballot(mine_own.own1 != kNone) != 0 && lane == 0) s_own_flag[fl_i] = 1;
            tick_lds_barrier();
            if (round == 0) YDC_TICK_STAMP(27);
            if (round == 0) {
              own_seen = s_own_flag[fl_i] != 0;
              if (t == 0) s_own_flag[(pk + 2) % 3] = 0;
              ++pk;
              if (own_seen) break;  // (nothing has been placed: the loop below takes all of them)
            }
            if (t < 64) {
              constexpr int T = THREADS / 64;  // lists per lane: threads lane + 64 q
              uint32_t pos = 0;                // 2 bits per list: entries taken
              Cand lb = tick_cand(kNo, kNone, ib);
              auto local_best = [&]() {
                lb = tick_cand(kNo, kNone, ib);
#pragma unroll
                for (int q = 0; q < T; ++q) {
                  const uint32_t pq = (pos >> (2 * q)) & 3u;
                  if (pq < 2) {
                    const Cand cq = s_lc[pq * THREADS + lane + 64 * q];
                    if (tick_cand_less(cq, lb)) lb = cq;
                  }
                }
              };
              local_best();
              uint32_t stop = 0;  // 1: a list ran dry, 2: nothing is free any more
              while (placed < n_tasks && !stop) {
                Cand b = lb;
                tick_row_reduce<16>(b);
                tick_dpp_step<0x142, 0xa>(b);
                tick_dpp_step<0x143, 0xc>(b);
                uint32_t bidx;
                double butil = -1.0;
                if constexpr (PK) {
                  bidx = tick_cand_idx(TickWord{tick_rl(b.w, 63)}, ib);
                } else {
                  bidx = tick_rl(b.idx, 63);
                  butil = __longlong_as_double((long long)((((uint64_t)tick_rl(b.khi, 63) << 32) | tick_rl(b.klo, 63)) &
                                                           0x7FFFFFFFFFFFFFFFull));
                }
                if (bidx == kNone) {
                  stop = 2;
                  break;
                }
                if (lane == 0) {
                  s_out[placed] = bidx;
                  s_util[placed] = butil;
                }
                ++placed;
                const uint32_t tw = bidx % THREADS;
                bool dry = false;
                if (lane == (tw & 63u)) {
                  const uint32_t q2 = 2 * (tw >> 6);
                  pos += 1u << q2;
                  dry = ((pos >> q2) & 3u) == 2u;
                  local_best();
                }
                if (__ballot(dry) != 0) stop = 1;
              }
#pragma unroll
              for (int q = 0; q < T; ++q) s_cnt[lane + 64 * q] = (pos >> (2 * q)) & 3u;
              if (lane == 0) {
                s_cmd[2] = placed;
                s_cmd[3] = stop;
              }
            }
            tick_lds_barrier();
            if (round == 0) YDC_TICK_STAMP(28);
            const uint32_t took = s_cnt[t], stop = s_cmd[3];
            placed = s_cmd[2];
#ifdef YDC_PHASE_PROBE
            if (t == 0) {
              ydc_phase_probe[30] = round + 1;
              if (round == 0) ydc_phase_probe[31] = placed;
            }
#endif
            if (took) {
              // ++running_tasks (:123) of what was taken from this thread: its first candidate, and
              // with two its second — the same servant again, or the runner-up.
              const uint32_t add0 = 1u + (took == 2 && e1_same ? 1u : 0u), add1 = took == 2 && !e1_same ? 1u : 0u;
              uint32_t np0 = 0, ld0 = 0, mt0 = 0, fl0 = 0, r0 = 0, np1 = 0, ld1 = 0, mt1 = 0, fl1 = 0, r1 = 0;
              if (!COLD) {
                const uint32_t s0 = tick_cand_idx(mine, ib), s1 = tick_cand_idx(e1, ib);
                np0 = p_nproc[s0];
                ld0 = p_load[s0];
                mt0 = p_maxt[s0];
                fl0 = p_flags[s0];
                if (add1) {
                  np1 = p_nproc[s1];
                  ld1 = p_load[s1];
                  mt1 = p_maxt[s1];
                  fl1 = p_flags[s1];
                }
              }
#pragma unroll
              for (int k = 0; k < K; ++k) {
                const bool m0 = (uint32_t)k == k0, m1 = add1 && (uint32_t)k == k1;
                c_run[k] += m0 ? add0 : (m1 ? 1u : 0u);
                r0 = m0 ? c_run[k] : r0;
                r1 = m1 ? c_run[k] : r1;
                if (COLD) {
                  np0 = m0 ? c_nproc[k] : np0;
                  ld0 = m0 ? c_load[k] : ld0;
                  mt0 = m0 ? c_maxt[k] : mt0;
                  fl0 = m0 ? c_flags[k] : fl0;
                  np1 = m1 ? c_nproc[k] : np1;
                  ld1 = m1 ? c_load[k] : ld1;
                  mt1 = m1 ? c_maxt[k] : mt1;
                  fl1 = m1 ? c_flags[k] : fl1;
                }
              }
              const KeyT nka = add0 == 1 ? nk0 : tick_key<PK>(np0, ld0, mt0, fl0, r0, true, cap_bits);
              const KeyT nkb = add1 ? tick_key<PK>(np1, ld1, mt1, fl1, r1, true, cap_bits) : (KeyT)0;
#pragma unroll
              for (int k = 0; k < K; ++k) {
                const bool m0 = (uint32_t)k == k0, m1 = add1 && (uint32_t)k == k1;
                key[k] = m0 ? nka : (m1 ? nkb : key[k]);
              }
              changed |= (1u << k0) | (add1 ? 1u << k1 : 0u);
              dirty = true;
            }
            if (round == 0) YDC_TICK_STAMP(29);
            if (stop == 2) {  // Timeout (:116-118) for every request that is left
              if (t >= placed && t < n_tasks) {
                s_out[t] = kIdxTimeout;
                s_util[t] = -1.0;
              }
              n_timeout += n_tasks - placed;
              n_granted += placed;
              placed = n_tasks;
              i_start = n_tasks;
              break;
            }
            if (placed == n_tasks) {
              n_granted += placed;
              i_start = n_tasks;
            }
          }
        }
        if (i_start == n_tasks) {  // the results of a merged command (the loop below does not run)
          tick_lds_barrier();
          if (t < n_tasks) {
            p_out_idx[t] = s_out[t];
            if (p_out_util) p_out_util[t] = s_util[t];
          }
        }
      }
    }
    for (uint32_t i = i_start; i < n_tasks; ++i) {
      const uint32_t oi = i & (kTickBlock - 1);
      if (p_tenv && oi == 0) {  // the next 64 requests' columns (device or mapped host memory)
        if (t < kTickBlock && i + t < n_tasks) {
          s_env[t] = p_tenv[i + t];
          s_minv[t] = p_tminv[i + t];
          s_rip[t] = p_trip[i + t];
        }
        tick_lds_barrier();
      }
      // (the same in every thread: say so, the barriers and ballots below sit behind tests of them)
      const uint32_t env = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_env[oi]),
                     minv = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_minv[oi]),
                     rip = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_rip[oi]);
      set_signature(env, minv, rip);
      if (!any) {  // nobody advertises the environment at that version: :105-108
        if (t == 0) {
          s_out[oi] = kIdxEnvNotFound;
          s_util[oi] = -1.0;
        }
        ++n_envnf;
      } else {
        if (dirty) {  // this thread's candidates: its state (or the signature) changed
          KeyT bk = kNo;
          uint32_t bs = kNone;
          mine_own.own1 = mine_own.own2 = kNone;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            if (!((elig >> k) & 1u) || key[k] == kNo) continue;
            const uint32_t s = (uint32_t)k * THREADS + tl;
            if ((ownb >> k) & 1u) {  // on the requestor's own host (ascending s: first, second)
              if (mine_own.own1 == kNone) mine_own.own1 = s;
              else if (mine_own.own2 == kNone) mine_own.own2 = s;
            } else if (key[k] < bk) {
              bk = key[k];
              bs = s;
            }
          }
          mine = tick_cand(bk, bs, ib);
          dirty = false;
          list_valid = false;
        }
        if (i == 2) YDC_TICK_STAMP(24);
        // Own-host candidates are rare: a flag says whether the second reduction is needed at all.
        const uint32_t fl_i = pk % 3;
        if (__ballot(mine_own.own1 != kNone) != 0 && lane == 0) s_own_flag[fl_i] = 1;
        Cand best = tick_block_reduce<WAVES>(mine, s_part[red++ & 1]);
        if (i == 2) YDC_TICK_STAMP(25);
        const bool own_any = s_own_flag[fl_i] != 0;
        if (t == 0) s_own_flag[(pk + 2) % 3] = 0;
        ++pk;
        TickOwn own{kNone, kNone};
        if (own_any) {
          own = tick_block_reduce<WAVES>(mine_own, s_part_own[red++ & 1]);
          if (own.own2 != kNone) {
            // Several eligible free servants on the requestor's host: only the first of them is
            // `self` (:372-379), the others compete like everybody else.
            KeyT bk = kNo;
            uint32_t bs = kNone;
#pragma unroll
            for (int k = 0; k < K; ++k) {
              if (!((elig >> k) & 1u) || key[k] == kNo) continue;
              const uint32_t s = (uint32_t)k * THREADS + tl;
              if (s != own.own1 && key[k] < bk) {
                bk = key[k];
                bs = s;
              }
            }
            best = tick_block_reduce<WAVES>(tick_cand(bk, bs, ib), s_part[red++ & 1]);
          }
        }
        if (i == 2) YDC_TICK_STAMP(26);
        const uint32_t best_idx = tick_cand_idx(best, ib);
        const uint32_t winner = best_idx != kNone ? best_idx : own.own1;  // :392-396
        if (winner == kNone) {  // eligible servants exist, none is free: Timeout with timeout == now (:116-118)
          if (t == 0) {
            s_out[oi] = kIdxTimeout;
            s_util[oi] = -1.0;
          }
          ++n_timeout;
        } else {
          ++n_granted;
          if (winner % THREADS == tl) {
            const uint32_t wk = winner / THREADS;
            uint32_t np = 0, ld = 0, mt = 0, fl = 0, run = 0;
            if (!COLD) {
              np = p_nproc[winner];
              ld = p_load[winner];
              mt = p_maxt[winner];
              fl = p_flags[winner];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {  // (selects, not branches: the key is computed once below)
              const bool me = (uint32_t)k == wk;
              run = me ? c_run[k] : run;
              if (COLD) {
                np = me ? c_nproc[k] : np;
                ld = me ? c_load[k] : ld;
                mt = me ? c_maxt[k] : mt;
                fl = me ? c_flags[k] : fl;
              }
            }
            s_out[oi] = winner;
            // (the double the reference compares, task_dispatcher.cc:440-441: the same division)
            s_util[oi] = slot_utilization(run, slot_capacity(np, ld, mt, run));
            run += 1;  // ++pick->running_tasks (:123); written back at the end
            const KeyT nk = tick_key<PK>(np, ld, mt, fl, run, true, cap_bits);
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const bool me = (uint32_t)k == wk;
              key[k] = me ? nk : key[k];
              c_run[k] = me ? run : c_run[k];
            }
            changed |= 1u << wk;
            dirty = true;
          }
        }
      }
      if (i < 16) YDC_TICK_STAMP(8 + i);  // (pick i done)
      if (oi == kTickBlock - 1 || i + 1 == n_tasks) {
        tick_lds_barrier();
        const uint32_t first = i - oi;
        if (t <= oi) {
          p_out_idx[first + t] = s_out[t];
          if (p_out_util) p_out_util[first + t] = s_util[t];
        }
      }
    }

    if (box) {
      // Resident: the answer first. Granules {word, command number} — every granule says by itself
      // that it is this command's (8-byte stores are atomic; no fence, no order between them):
      // granule 0 the counters, granules 1 .. 7 the first seven placements. Longer answers (and
      // the utilisations) lie in the arrays stored above, ahead of the granules.
      if (t < 64) {  // (wave 0: the array stores above are this wave's own)
        if (n_tasks > 7 || p_out_util) __threadfence_system();
        if (t < 8) {
          const uint32_t word = t == 0 ? (n_granted | (n_timeout << 8) | (n_envnf << 16))
                                       : (t - 1 < n_tasks ? s_out[t - 1] : 0u);
          __hip_atomic_store(&box->reply[t], ((unsigned long long)seq << 32) | word, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    // running_tasks goes back: the servants this command touched (released grants, picks).
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t s = (uint32_t)k * THREADS + tl;
      if ((changed >> k) & 1u) p_rw[s] = c_run[k];
      if (p_run_out && s < S) p_run_out[s] = c_run[k];
    }
    changed = 0;
    if (!box) {
      // Results first, then the stamp the host spins on.
      YDC_TICK_STAMP(6);
      __threadfence_system();
      __syncthreads();
      YDC_TICK_STAMP(7);
      if (t == 0) {
        p_done->granted = n_granted;
        p_done->timeouts = n_timeout;
        p_done->env_not_found = n_envnf;
        __hip_atomic_store(&p_done->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }

    // ---- resident: wait for the next command (wave 0 polls the mailbox, the others wait at the barrier) ----
    ++seq;
    if (seq == 0) seq = 1;
    if (t < 64) {
      const unsigned long long t0 = wall_clock64();
      uint32_t word = 0, leave = 0;
      for (;;) {
        const unsigned long long g =
            __hip_atomic_load(&box->head[lane & 15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const bool mine_ok = (uint32_t)(g >> 32) == seq;
        if ((__ballot(mine_ok) & 0xFFFFull) == 0xFFFFull) {
          word = (uint32_t)g;
          break;
        }
        if (wall_clock64() - t0 > idle_ticks) {  // nobody has asked for a while: give the CU back
          leave = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      // head: {cmd | n_tasks << 8 | n_upd << 16 | n_rel << 24}, env, minv, rip, rel[0 .. 3]
      const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)word, 0);
      if (!leave) {
        if (lane == 1) s_env[0] = word;
        if (lane == 2) s_minv[0] = word;
        if (lane == 3) s_rip[0] = word;
        if (lane >= 4 && lane < 8) s_rel[lane - 4] = word;
        if (lane == 8) s_uidx[0] = word;
        if (lane == 9) s_urow[0].nproc = word;
        if (lane == 10) s_urow[0].load = word;
        if (lane == 11) s_urow[0].max_tasks = word;
        if (lane == 12) s_urow[0].flags = word;
        if (lane >= 13 && lane < 16) s_rel[lane - 9] = word;
        const uint32_t nt = (w0 >> 8) & 0xFF, nu = (w0 >> 16) & 0xFF, nr = w0 >> 24;
        // What does not fit the head was stored before it (release stores on the host). The reads of
        // it below are ordered behind the head's by an acquire fence — only when there is
        // something to read: a single request, or copies of one, travel in the head alone.
        if ((nt > 1 && !(w0 & kTickCmdSame)) || nr > 7 || nu > 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        if (nt > 1 && (w0 & kTickCmdSame)) {  // copies of the first request: nothing more to fetch
          const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)word, 1),
                         m0 = (uint32_t)__builtin_amdgcn_readlane((int)word, 2),
                         r0 = (uint32_t)__builtin_amdgcn_readlane((int)word, 3);
          if (lane < nt) {
            s_env[lane] = e0;
            s_minv[lane] = m0;
            s_rip[lane] = r0;
          }
        } else if (nt > 1 && lane < nt) {
          s_env[lane] = __hip_atomic_load(&box->env[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          s_minv[lane] = __hip_atomic_load(&box->minv[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          s_rip[lane] = __hip_atomic_load(&box->rip[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (nr > 7 && lane < nr) s_rel[lane] = __hip_atomic_load(&box->rel[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (nu > 1 && lane < nu) {
          s_uidx[lane] = __hip_atomic_load(&box->upd_idx[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          TickRow r;
          r.nproc = __hip_atomic_load(&box->upd[lane].nproc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          r.load = __hip_atomic_load(&box->upd[lane].load, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          r.max_tasks = __hip_atomic_load(&box->upd[lane].max_tasks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          r.flags = __hip_atomic_load(&box->upd[lane].flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          s_urow[lane] = r;
        }
      }
      if (lane == 0) {
        s_cmd[0] = w0;
        s_cmd[1] = leave ? 2u : ((w0 & 0x7F) == kTickCmdQuit ? 1u : 0u);
      }
    }
    tick_lds_barrier();  // (s_cmd; the barrier at the top of the next turn covers the staged payload again)
    const uint32_t w0 = s_cmd[0], going = s_cmd[1];
    if (going) {
      // Leaving: a QUIT is answered (the host waits for it), an idle exit is not. `alive` last.
      if (t == 0) {
        if (going == 1)
          __hip_atomic_store(&box->reply[0], ((unsigned long long)seq << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&box->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
    n_tasks = (w0 >> 8) & 0xFF;
    n_upd = (w0 >> 16) & 0xFF;
    n_rel = w0 >> 24;
  }
}

}  // namespace ydc
#endif  // YADCC_AMD_TICK_KERNEL_H_

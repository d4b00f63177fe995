// match_kernel.h — k_match_pass: the greedy merge of the pending requests over
// the per-class slot lists, chunk-parallel and speculative.
//
// One wave (= one workgroup) per chunk of requests, one lane per servant class
// (W classes per lane above 64).
//
//   pass 0   every chunk is replayed from its level guess (k_guess_init) and
//            leaves: the slot of every request, a checkpoint of the class states
//            before every block of 64 requests, its end state.
//   pass r   chunk k is *consistent* when the state it was last replayed from
//            (its first checkpoint) equals the end state of chunk k-1. An
//            inconsistent chunk is replayed from that end state; the replay
//            stops as soon as the states equal a checkpoint of the previous
//            replay (identical remainder). If its own end state changed, the
//            wave carries on into chunk k+1 unless another wave has claimed it
//            in this pass — a perturbation that needs thousands of requests to
//            die out is followed by one wave instead of one launch per chunk.
//   A pass that finds every chunk consistent changes nothing and proves that
//   every chunk was replayed from its predecessor's final state: the result is
//   the sequential one (chunk 0 always starts from the true state). The host
//   pre-launches a few passes; a pass returns at once when an earlier one found
//   nothing to do, and k_finalize only runs behind such a pass.
//
// End states are updated in place. A wave that reads its predecessor's end
// state while that is being rewritten replays from a mixed (meaningless but
// harmless) state; the inconsistency shows in the next pass.
//
// The pick of one request depends on the previous one, so the inner loop is a
// dependency chain. Per request: two v_readlane (the request's class mask), one
// v_cndmask that uses the mask as a lane mask, a 6-step DPP min, a ballot, a
// v_readlane of the winning slot; the winning lane advances its class from
// registers (head and next entry are kept in VGPRs, the LDS ring read for the
// entry after next is not waited for until the lane wins again).
#ifndef YADCC_AMD_MATCH_KERNEL_H_
#define YADCC_AMD_MATCH_KERNEL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dispatch_core.h"

namespace ydc {

constexpr uint32_t kPassSlots = 64;  // DeviceParams::n_changed is indexed by pass & 63

struct MatchBuffers {
  const ClassState* guess0;  // [K * C] level guesses (start states of pass 0)
  ClassState* endst;         // [K * C] end state of every chunk (in place)
  ClassState* checkpoint;    // [ceil(N / 64) * C] state before each block of 64 requests
  unsigned long long* claim; // [K] (batch, pass) stamp of the pass in which a wave took the chunk
  uint32_t* slot_of;         // [N] generation index of the slot each request takes
  // Multi-GPU: this rank's chunks continue the previous rank's. boundary_in (C
  // entries, or NULL) is the end state of the predecessor's last chunk.
  const ClassState* boundary_in;
};

// Everything a lane keeps about one of its classes.
struct LaneClass {
  uint32_t cursor, lo, hown_lo, hown_hi, end;
  uint32_t head_p, head_g;  // entry at `cursor`     (kNone past the end)
  uint32_t next_p, next_g;  // entry at `cursor + 1` (kNone past the end)
  uint32_t filled;          // ring holds list entries [cursor, filled)
};

template <int W>
struct MatchWave {
  const ClassLists& L;
  uint32_t* ring_p;  // [C][R]
  uint32_t* ring_g;
  uint32_t rmask;    // R - 1
  uint32_t rshift;   // log2 R
  uint32_t lane;
  LaneClass k[W];

  __device__ __forceinline__ uint32_t at(uint32_t cl, uint32_t i) const {
    return (cl << rshift) + (i & rmask);
  }
  __device__ __forceinline__ ClassRun as_run(int j) const {
    ClassRun r;
    r.cursor = k[j].cursor;
    r.lo = k[j].lo;
    r.hown_lo = k[j].hown_lo;
    r.hown_hi = k[j].hown_hi;
    r.end = k[j].end;
    r.head_p = k[j].head_p;
    r.head_g = k[j].head_g;
    return r;
  }
  __device__ __forceinline__ ClassState state(int j) const {
    ClassState s;
    s.cursor = k[j].cursor;
    s.lo = k[j].lo;
    const bool holes = k[j].lo < k[j].cursor;
    s.hown_lo = holes ? k[j].hown_lo : kNone;
    s.hown_hi = holes ? k[j].hown_hi : kNone;
    return s;
  }
  // Robust against arbitrary (speculative or torn) states: indexes are clamped.
  __device__ __forceinline__ void set_state(int j, const ClassState& st, uint32_t c, uint32_t C) {
    LaneClass& q = k[j];
    if (c < C) {
      const uint32_t b = L.cls_begin[c], e = L.cls_begin[c + 1];
      uint32_t cur = st.cursor, lo = st.lo;
      cur = cur < b ? b : (cur > e ? e : cur);
      lo = lo < b ? b : (lo > cur ? cur : lo);
      q.cursor = cur;
      q.lo = lo;
      q.hown_lo = st.hown_lo;
      q.hown_hi = st.hown_hi;
      q.end = e;
    } else {
      q.cursor = q.lo = q.end = 0;
      q.hown_lo = q.hown_hi = kNone;
    }
    q.head_p = q.head_g = q.next_p = q.next_g = kNone;
    q.filled = q.cursor;
  }
  // Wave-uniform: all lanes load list entries [from, to) of class cl (to - from <= 64).
  // Positions at or after the end of the class's list (`end`) get the sentinel kNone, so
  // that (head, next) read past the end of a list are "no slot" without a compare.
  __device__ __forceinline__ void fill(uint32_t cl, uint32_t from, uint32_t to, uint32_t end) {
    const uint32_t e = from + lane;
    if (e < to) {
      const bool real = e < end;
      ring_p[at(cl, e)] = real ? list_rank(L, e) : kNone;
      ring_g[at(cl, e)] = real ? L.list_g[e] : kNone;
    }
  }
  // First fill: every lane loads `n` (<= R) entries of its own classes,
  // 16 independent loads in flight per lane and array.
  __device__ __forceinline__ void init_rings(uint32_t C, uint32_t n, int W_) {
    for (int j = 0; j < W_; ++j) {
      const uint32_t cl = lane + 64 * j;
      LaneClass& q = k[j];
      if (cl < C) {
        for (uint32_t b0 = q.cursor; b0 < q.cursor + n; b0 += 16) {
          uint32_t tp[16], tg[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const uint32_t e = b0 + u;
            tp[u] = e < q.end ? list_rank(L, e) : kNone;
            tg[u] = e < q.end ? L.list_g[e] : kNone;
          }
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            if (b0 + u < q.cursor + n) {  // n may be below 16 (tiny rings)
              ring_p[at(cl, b0 + u)] = tp[u];
              ring_g[at(cl, b0 + u)] = tg[u];
            }
          }
        }
        q.filled = q.cursor + n;
      }
    }
  }
  // Entries the ring of this lane's class j still holds beyond the cursor; "plenty" once
  // the sentinels behind the end of the list are in.
  __device__ __forceinline__ uint32_t ring_left(int j) const {
    return k[j].filled >= k[j].end + 2 ? 0x7FFFFFFFu : k[j].filled - k[j].cursor;
  }
  // (head, next) of this lane's class j from the ring (filled >= cursor + 2 always).
  __device__ __forceinline__ void load_heads(int j) {
    const uint32_t cl = lane + 64 * j;
    LaneClass& q = k[j];
    q.head_p = ring_p[at(cl, q.cursor)];
    q.head_g = ring_g[at(cl, q.cursor)];
    q.next_p = ring_p[at(cl, q.cursor + 1)];
    q.next_g = ring_g[at(cl, q.cursor + 1)];
  }
};

// bit `lane` of m ? v : kNone — the request's class mask used as a lane mask.
__device__ __forceinline__ uint32_t select_by_lane_mask(uint64_t m, uint32_t v) {
  uint32_t out;
  asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(out) : "v"(kNone), "v"(v), "s"(m));
  return out;
}

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, uint32_t l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}


// The fast loop of one block of (up to 64) requests, W == 1, hand-scheduled. Runs up to
// n requests i, i+1, ... while each of them is "plain": no eligible class has holes and no
// eligible class shows a slot of the requestor's own servant at its head. Returns
//   0  n requests done,
//   1  request i needs the general step (i requests of this call were done).
// Requests whose eligible classes are all exhausted (or that have none) keep the default
// result in `res`. The caller guarantees that every ring holds at least n + 2 entries
// beyond its cursor, so the loop never looks at fill levels.
//
// Per request: the class mask (an aligned SGPR pair, fetched one request ahead in the
// wait states of the DPP chain) selects the eligible heads, a `steps`-step DPP min over the
// first 2^steps lanes finds the winner, the winning lane alone (exec = the compare mask)
// moves (next -> head), starts the LDS read of the entry after next and bumps its cursor;
// the result is parked in the winner's lane and copied to lane i off the critical path.
// Wait states (gfx940/gfx950): VALU-written SGPR -> VALU read 2, VALU-written VGPR -> DPP
// read 2, VALU-written VGPR -> v_readlane 1. s[90:95] are scratch.
#define YDC_DPP(ctrl) "v_min_u32_dpp %[t], %[t], %[t] " ctrl "\n"
// After step k: stop when k steps are enough (the compare + branch are the two wait states
// the next DPP step needs anyway).
#define YDC_DPP_STOP(k) "s_cmp_eq_u32 %[steps], " #k "\ns_cbranch_scc1 L_red%=\n"
#define YDC_DPP_SEQ                                                            \
  YDC_DPP("row_shr:1 row_mask:0xf bank_mask:0xf") YDC_DPP_STOP(1)              \
  YDC_DPP("row_shr:2 row_mask:0xf bank_mask:0xf") YDC_DPP_STOP(2)              \
  YDC_DPP("row_shr:4 row_mask:0xf bank_mask:0xf") YDC_DPP_STOP(3)              \
  YDC_DPP("row_shr:8 row_mask:0xf bank_mask:0xf") YDC_DPP_STOP(4)              \
  YDC_DPP("row_bcast:15 row_mask:0xa bank_mask:0xf") YDC_DPP_STOP(5)           \
  YDC_DPP("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 0\nL_red%=:\n"

#define YDC_FAST_LOOP(NAME)                                                    \
  __device__ __forceinline__ uint32_t NAME(                                                      \
      uint32_t& i, uint32_t n, uint32_t mlo, uint32_t mhi, uint32_t slo, uint32_t shi,           \
      uint64_t holes, uint64_t has_self, uint32_t& res, uint32_t& hp, uint32_t& hg,              \
      uint32_t& np, uint32_t& ng, uint32_t& cur, uint32_t off, uint32_t base, uint32_t rmask4,    \
      uint32_t steps, uint32_t last) {                                                          \
    uint32_t status, c, t, a, tkv, mn, tk, win, i1, s0, s1, m0save;                             \
    asm volatile(                                                                                \
        "s_mov_b32 %[m0s], m0\n"                                                                 \
        "s_mov_b32 %[st], 0\n"                                                                   \
        "s_nop 3\n"                                                                              \
        "s_cmp_eq_u32 %[n], 0\n"                                                                 \
        "s_cbranch_scc1 L_out%=\n"                                                               \
        "v_readlane_b32 s90, %[mlo], %[i]\n"                                                     \
        "v_readlane_b32 s91, %[mhi], %[i]\n"                                                     \
        "s_add_u32 %[i1], %[i], 1\n"                                                             \
        "L_loop%=:\n"                                                                            \
        "s_and_b64 s[92:93], s[90:91], %[holes]\n"                                               \
        "s_cbranch_scc1 L_slow%=\n"                                                              \
        "s_bitcmp1_b64 %[hs], %[i]\n"                                                            \
        "s_cbranch_scc1 L_self%=\n"                                                              \
        "L_cont%=:\n"                                                                            \
        "v_cndmask_b32 %[c], -1, %[hp], s[90:91]\n"                                              \
        "v_mov_b32 %[t], %[c]\n"                                                                 \
        "v_readlane_b32 s94, %[mlo], %[i1]\n"                                                    \
        "v_readlane_b32 s95, %[mhi], %[i1]\n" YDC_DPP_SEQ                                        \
        "v_readlane_b32 %[mn], %[t], %[last]\n"                                                  \
        "s_cmp_eq_u32 %[mn], -1\n"                                                               \
        "s_cbranch_scc1 L_tmo%=\n"                                                               \
        "v_cmp_eq_u32 vcc, %[mn], %[c]\n"                                                        \
        "s_mov_b64 exec, vcc\n"                                                                  \
        "v_mov_b32 %[tkv], %[hg]\n"                                                              \
        "s_waitcnt lgkmcnt(0)\n"                                                                 \
        "v_mov_b32 %[hp], %[np]\n"                                                               \
        "v_mov_b32 %[hg], %[ng]\n"                                                               \
        "v_add_u32 %[off], 4, %[off]\n"                                                          \
        "v_and_b32 %[off], %[rmask4], %[off]\n"                                                  \
        "v_add_u32 %[a], %[base], %[off]\n"                                                      \
        "ds_read_b32 %[np], %[a]\n"                                                              \
        "ds_read_b32 %[ng], %[a] offset:8192\n"                                                  \
        "v_add_u32 %[cur], 1, %[cur]\n"                                                          \
        "s_mov_b64 exec, -1\n"                                                                   \
        "s_ff1_i32_b64 %[win], vcc\n"                                                            \
        "s_mov_b32 m0, %[i]\n"                                                                   \
        "v_readlane_b32 %[tk], %[tkv], %[win]\n"                                                 \
        "s_mov_b64 s[90:91], s[94:95]\n"                                                         \
        "s_nop 0\n"                                                                              \
        "v_writelane_b32 %[res], %[tk], m0\n"                                                    \
        "L_next%=:\n"                                                                            \
        "s_add_u32 %[i], %[i], 1\n"                                                              \
        "s_add_u32 %[i1], %[i1], 1\n"                                                            \
        "s_add_u32 %[n], %[n], -1\n"                                                             \
        "s_cmp_lg_u32 %[n], 0\n"                                                                 \
        "s_cbranch_scc1 L_loop%=\n"                                                              \
        "s_branch L_out%=\n"                                                                     \
        "L_tmo%=:\n"                                                                             \
        "s_bitcmp1_b64 %[hs], %[i]\n"                                                            \
        "s_cbranch_scc1 L_slow%=\n"                                                              \
        "s_mov_b64 s[90:91], s[94:95]\n"                                                         \
        "s_branch L_next%=\n"                                                                    \
        "L_self%=:\n"                                                                            \
        "v_readlane_b32 %[s0], %[slo], %[i]\n"                                                   \
        "v_readlane_b32 %[s1], %[shi], %[i]\n"                                                   \
        "s_sub_u32 %[s1], %[s1], %[s0]\n"                                                        \
        "v_subrev_u32 %[a], %[s0], %[hg]\n"                                                      \
        "v_cmp_gt_u32 vcc, %[s1], %[a]\n"                                                        \
        "s_and_b64 s[92:93], vcc, s[90:91]\n"                                                    \
        "s_cbranch_scc0 L_cont%=\n"                                                              \
        "L_slow%=:\n"                                                                            \
        "s_mov_b32 %[st], 1\n"                                                                   \
        "L_out%=:\n"                                                                             \
        "s_waitcnt lgkmcnt(0)\n"                                                                 \
        "s_mov_b32 m0, %[m0s]\n"                                                                 \
        : [st] "=&s"(status), [i] "+s"(i), [n] "+s"(n), [res] "+v"(res), [hp] "+v"(hp),          \
          [hg] "+v"(hg), [np] "+v"(np), [ng] "+v"(ng), [cur] "+v"(cur), [off] "+v"(off),         \
          [c] "=&v"(c), [t] "=&v"(t), [a] "=&v"(a), [tkv] "=&v"(tkv), [mn] "=&s"(mn),            \
          [tk] "=&s"(tk), [win] "=&s"(win), [i1] "=&s"(i1), [s0] "=&s"(s0), [s1] "=&s"(s1),      \
          [m0s] "=&s"(m0save)                                                                    \
        : [mlo] "v"(mlo), [mhi] "v"(mhi), [slo] "v"(slo), [shi] "v"(shi), [holes] "s"(holes),    \
          [hs] "s"(has_self), [base] "v"(base), [rmask4] "s"(rmask4), [steps] "s"(steps),        \
          [last] "s"(last)                                                                       \
        : "vcc", "scc", "memory", "s90", "s91", "s92", "s93", "s94", "s95");                     \
    return status;                                                                               \
  }

YDC_FAST_LOOP(match_fast_loop)

template <int W>
__global__ __launch_bounds__(64) void k_match_pass(ClassLists L, TaskTable T, uint32_t n_tasks,
                                                   uint32_t chunk_size, uint32_t n_chunks,
                                                   MatchBuffers B, uint32_t pass,
                                                   uint32_t device_check, uint32_t rshift,
                                                   uint32_t init_fill, DeviceParams* prm) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_ring[];
  if (prm->need_shared) return;  // the batch went through the sequential path
  // An earlier pass found every chunk consistent: nothing to do.
  if (device_check && pass > 0 && prm->n_changed[(pass - 1) & (kPassSlots - 1)] == 0) return;
  const uint32_t lane = threadIdx.x;
  uint32_t kc = blockIdx.x;  // chunk
  if (kc >= n_chunks) return;
  const uint32_t C = L.n_classes;
  const bool multi = B.boundary_in != nullptr;
  // Unique per pass launch and ever growing (the batch counter lives on the device, so a
  // replayed graph gets fresh stamps too).
  const unsigned long long stamp = ((unsigned long long)prm->batch_seq << 16) | (pass + 1);

  // Fixed layout: ranks in the first 8 KB, generation indexes 8 KB further (the asm loop
  // addresses the second array with an immediate offset). C << rshift <= 2048.
  MatchWave<W> w{L, lds_ring, lds_ring + 2048, (1u << rshift) - 1, rshift, lane, {}};
  const uint32_t R = 1u << rshift;
  // A ring is topped up when no more than `thresh` entries are left in it.
  const uint32_t thresh = R / 4 < 4 ? 4 : (R / 4 > 12 ? 12 : R / 4);
  uint32_t steps = 1;  // DPP steps of the min over the class lanes
  while ((1u << steps) < C) ++steps;

  // ---- start state; is there anything to do? ----
  {
    const ClassState* start;
    if (pass == 0) {
      start = B.guess0 + (size_t)kc * C;
      if (kc == 0 && lane == 0) atomicAdd(&prm->n_changed[0], 1u);  // pass 0 always has work
    } else {
      if (kc == 0 && !multi) return;  // chunk 0 started from the true state
      start = kc == 0 ? B.boundary_in : B.endst + (size_t)(kc - 1) * C;
    }
    bool differs = false;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const uint32_t c = lane + 64 * j;
      ClassState st{};
      if (c < C) {
        st = start[c];
        if (pass != 0) {
          const ClassState used = B.checkpoint[(size_t)((kc * chunk_size) >> 6) * C + c];
          differs |= !class_state_equal(used, st);
        }
      }
      w.set_state(j, st, c, C);
    }
    if (pass != 0) {
      if (__ballot(differs) == 0) return;  // consistent
      // Inconsistent: this pass has work. One wave per chunk and pass.
      uint32_t taken = 0;
      if (lane == 0) {
        atomicAdd(&prm->n_changed[pass & (kPassSlots - 1)], 1u);
        taken = atomicMax(&B.claim[kc], stamp) == stamp;
      }
      if (readlane_u32(taken, 0)) return;  // a wave following its chain got here first
    }
  }

  bool ring_ready = false;
  uint64_t holes[W] = {};

  // Tops up the ring of class j of lane `cc`: one coalesced load of up to 64 entries.
  auto refill = [&](uint32_t cc, int bj) {
    uint32_t f_cl = 0, f_from = 0, f_to = 0, f_end = 0;
    if (lane == cc) {
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (j == bj) {
          LaneClass& q = w.k[j];
          if (q.filled < q.cursor) q.filled = q.cursor;
          f_cl = lane + 64 * j;
          f_from = q.filled;
          f_to = min(q.filled + 64u, q.cursor + R);
          f_end = q.end;
          q.filled = f_to;
        }
      }
    }
    w.fill(readlane_u32(f_cl, cc), readlane_u32(f_from, cc), readlane_u32(f_to, cc),
           readlane_u32(f_end, cc));
  };
  // Refills every ring that is running low; returns how many picks are safe before the
  // next look (every ring keeps the entry after next).
  auto top_up = [&]() -> uint32_t {
    uint32_t least = 0x7FFFFFFFu;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const bool real = lane + 64 * j < C;
      uint64_t need = __ballot(real && w.ring_left(j) <= thresh);
      while (need) {
        refill((uint32_t)__builtin_ctzll(need), j);
        need &= need - 1;
      }
      least = min(least, real ? w.ring_left(j) : 0x7FFFFFFFu);
    }
    __builtin_amdgcn_wave_barrier();
    return wave_min_u32(least) - 2;
  };

  for (;;) {  // chunk kc, then the chunks after it while the end states keep changing
    const uint32_t t0 = kc * chunk_size;
    const uint32_t t1 = min(n_tasks, t0 + chunk_size);
    bool stopped_early = false;

    // Requests of a block are staged one block ahead: lane l holds request tb + l.
    uint64_t nx_m[W];
    uint32_t nx_slo, nx_shi;
    ClassState nx_cp[W];
    auto stage = [&](uint32_t tb) {
      const uint32_t tl = tb + lane;
      nx_slo = nx_shi = kNone;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        nx_m[j] = (tl < t1 && (uint32_t)j < T.words) ? T.mask[(size_t)tl * T.words + j] : 0;
        const uint32_t c = lane + 64 * j;
        if (pass != 0 && tb < t1 && c < C) nx_cp[j] = B.checkpoint[(size_t)(tb >> 6) * C + c];
      }
      if (tl < t1) {
        nx_slo = T.self_lo[tl];
        nx_shi = T.self_hi[tl];
      }
    };
    stage(t0);

    for (uint32_t tb = t0; tb < t1; tb += 64) {
      // ---- checkpoint: stop if the previous replay was in the same state here ----
      {
        ClassState* cp = B.checkpoint + (size_t)(tb >> 6) * C;
        bool differs = false;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const uint32_t c = lane + 64 * j;
          if (c < C) {
            const ClassState s = w.state(j);
            if (pass == 0) {
              cp[c] = s;
            } else if (!class_state_equal(nx_cp[j], s)) {
              differs = true;
              cp[c] = s;
            }
          }
        }
        if (pass != 0 && __ballot(differs) == 0) {
          stopped_early = true;
          break;
        }
      }
      // ---- this block's requests; start fetching the next block's ----
      uint32_t mlo[W], mhi[W];
      uint64_t many = 0;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        mlo[j] = (uint32_t)nx_m[j];
        mhi[j] = (uint32_t)(nx_m[j] >> 32);
        many |= nx_m[j];
      }
      const uint32_t slo = nx_slo, shi = nx_shi;
      const uint32_t tl = tb + lane;
      stage(tb + 64);

      if (!ring_ready) {
        // First block that really runs: fill the rings.
        w.init_rings(C, init_fill, W);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < W; ++j) {
          if (lane + 64 * j < C) w.load_heads(j);
          holes[j] = __ballot(w.k[j].lo < w.k[j].cursor);
        }
        ring_ready = true;
      }

      const uint64_t has_self = __ballot(slo != kNone);
      uint32_t res = kIdxTimeout;
      const uint32_t cnt = min(64u, t1 - tb);

      // General step for request i: holes, own-servant heads, last-resort self pick. The
      // shared state machine (dispatch_core.h) advances the class; its ring is topped up
      // at the new cursor.
      auto general_step = [&](uint32_t i) {
        uint64_t mw[W];
#pragma unroll
        for (int j = 0; j < W; ++j)
          mw[j] = ((uint64_t)readlane_u32(mhi[j], i) << 32) | readlane_u32(mlo[j], i);
        const uint32_t self_lo = readlane_u32(slo, i);
        const uint32_t self_hi = readlane_u32(shi, i);
        uint32_t bp = kNone, bi = 0, bg = 0;
        int bj = 0;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          if ((mw[j] >> lane) & 1u) {
            uint32_t ci, cp, cg;
            if (class_candidate(L, w.as_run(j), self_lo, self_hi, ci, cp, cg) && cp < bp) {
              bp = cp;
              bi = ci;
              bg = cg;
              bj = j;
            }
          }
        }
        const uint32_t mn = wave_min_u32(bp);
        uint64_t winners;
        if (mn != kNone) {
          winners = __ballot(bp == mn);
        } else {
          bool ok = false;
          if (self_lo != kNone) {  // task_dispatcher.cc:392-396
#pragma unroll
            for (int j = 0; j < W; ++j) {
              if (!ok && ((mw[j] >> lane) & 1u)) {
                uint32_t ci, cg;
                if (class_self_candidate(L, w.as_run(j), self_lo, self_hi, ci, cg)) {
                  ok = true;
                  bi = ci;
                  bg = cg;
                  bj = j;
                }
              }
            }
          }
          winners = __ballot(ok);
        }
        if (winners == 0) return;  // Timeout (res default) — or no class at all (fixed below)
        const uint32_t win = (uint32_t)__builtin_ctzll(winners);
        const uint32_t taken = readlane_u32(bg, win);
        res = lane == i ? taken : res;
        bool moved = false;
        if (lane == win) {
#pragma unroll
          for (int j = 0; j < W; ++j) {
            if (j == bj) {
              ClassRun r = w.as_run(j);
              moved = class_consume_state(L, r, bi, self_lo, self_hi);
              LaneClass& q = w.k[j];
              q.cursor = r.cursor;
              q.lo = r.lo;
              q.hown_lo = r.hown_lo;
              q.hown_hi = r.hown_hi;
            }
          }
        }
        if (__ballot(moved)) {
          // The cursor moved (possibly past what the ring held): top the ring up at the
          // new cursor and reload (head, next).
          refill(win, (int)readlane_u32((uint32_t)bj, win));
          __builtin_amdgcn_wave_barrier();
          if (lane == win) {
#pragma unroll
            for (int j = 0; j < W; ++j)
              if (j == bj) w.load_heads(j);
          }
        }
#pragma unroll
        for (int j = 0; j < W; ++j) holes[j] = __ballot(w.k[j].lo < w.k[j].cursor);
      };

      if constexpr (W == 1) {
        LaneClass& q = w.k[0];
        const uint32_t base = (uint32_t)(uintptr_t)lds_ring + ((lane << rshift) << 2);
        const uint32_t rmask4 = (R << 2) - 1;
        uint32_t i = 0;
        while (i < cnt) {
          const uint32_t budget = top_up();
          const uint32_t n = min(cnt - i, budget);
          const uint32_t off = ((q.cursor + 1) & w.rmask) << 2;  // ring offset of `next`
          const uint32_t st = match_fast_loop(i, n, mlo[0], mhi[0], slo, shi, holes[0], has_self, res,
                                              q.head_p, q.head_g, q.next_p, q.next_g, q.cursor, off,
                                              base, rmask4, steps, (1u << steps) - 1);
          // Classes without holes were advanced with lo == cursor.
          if (!((holes[0] >> lane) & 1)) q.lo = q.cursor;
          if (st == 1) {
            general_step(i);
            ++i;
          }
        }
      } else {
        uint32_t budget = 0;
        for (uint32_t i = 0; i < cnt; ++i) {
          if (budget == 0) budget = top_up();
          uint64_t mw[W];
          bool general = false;
          uint64_t many_i = 0;
#pragma unroll
          for (int j = 0; j < W; ++j) {
            mw[j] = ((uint64_t)readlane_u32(mhi[j], i) << 32) | readlane_u32(mlo[j], i);
            general |= (mw[j] & holes[j]) != 0;  // an eligible class has holes
            many_i |= mw[j];
          }
          if (many_i == 0) continue;
          const bool self = (has_self >> i) & 1;
          if (self && !general) {
            const uint32_t self_lo = readlane_u32(slo, i);
            const uint32_t self_len = readlane_u32(shi, i) - self_lo;
            bool own = false;  // an eligible class shows a slot of the requestor's own servant
#pragma unroll
            for (int j = 0; j < W; ++j)
              own |= ((mw[j] >> lane) & 1u) && (w.k[j].head_g - self_lo < self_len);
            general = __ballot(own) != 0;
          }
          if (!general) {
            uint32_t bp = select_by_lane_mask(mw[0], w.k[0].head_p);
            uint32_t bg = w.k[0].head_g;
            int bj = 0;
#pragma unroll
            for (int j = 1; j < W; ++j) {
              const uint32_t c = select_by_lane_mask(mw[j], w.k[j].head_p);
              if (c < bp) {
                bp = c;
                bg = w.k[j].head_g;
                bj = j;
              }
            }
            const uint32_t mn = wave_min_u32(bp);
            if (mn != kNone) {
              const uint32_t win = (uint32_t)__builtin_ctzll(__ballot(bp == mn));
              const uint32_t taken = readlane_u32(bg, win);
              res = lane == i ? taken : res;
              if (lane == win) {
#pragma unroll
                for (int j = 0; j < W; ++j) {
                  if (j == bj) {
                    LaneClass& q = w.k[j];
                    const uint32_t cur = q.cursor + 1;
                    q.cursor = cur;
                    q.lo = cur;
                    q.head_p = q.next_p;
                    q.head_g = q.next_g;
                    q.next_p = w.ring_p[w.at(lane + 64 * j, cur + 1)];
                    q.next_g = w.ring_g[w.at(lane + 64 * j, cur + 1)];
                  }
                }
              }
              --budget;
              continue;
            }
            // Every eligible class is exhausted: Timeout (task_dispatcher.cc:116-118 with
            // timeout == now), unless the requestor's own servant may still serve it.
            if (!self) continue;
          }
          general_step(i);
          budget = 0;
        }
      }
      // No eligible class at all: EnvironmentNotFound (task_dispatcher.cc:105-108).
      if (many == 0) res = kIdxEnvNotFound;
      if (tl < t1) B.slot_of[tl] = res;
    }
    if (lane == 0) atomicAdd(&prm->chunk_sims, 1u);
    if (stopped_early) return;  // same remainder as last time: the end state stands

    // ---- end state ----
    bool differs = false;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const uint32_t c = lane + 64 * j;
      if (c < C) {
        const ClassState s = w.state(j);
        ClassState* e = B.endst + (size_t)kc * C + c;
        if (pass == 0) {
          *e = s;
        } else {
          const ClassState old = *e;
          if (!class_state_equal(old, s)) {
            differs = true;
            *e = s;
          }
        }
      }
    }
    if (pass == 0) return;
    if (__ballot(differs) == 0) return;  // nothing downstream is affected
    // The next chunk is now inconsistent. Follow the chain unless its own wave (or
    // another follower) has it in this pass; then the next pass picks it up.
    if (kc + 1 >= n_chunks) return;
    ++kc;
    uint32_t taken = 0;
    if (lane == 0) taken = atomicMax(&B.claim[kc], stamp) == stamp;
    if (readlane_u32(taken, 0)) return;
  }
}

}  // namespace ydc
#endif  // YADCC_AMD_MATCH_KERNEL_H_

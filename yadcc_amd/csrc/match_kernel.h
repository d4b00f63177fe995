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
  uint32_t* claim;           // [K] stamp of the pass in which a wave took the chunk
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


// The fast loop of one block of (up to 64) requests, W == 1, hand-scheduled. Runs
// requests i, i+1, ... while each of them is "plain": no eligible class has holes, no
// eligible class shows a slot of the requestor's own servant at its head, and the ring of
// the class that wins is not running low. Returns
//   0  all cnt requests done,
//   1  request i needs the general step,
//   2  request i-1 was served by lane `win`, whose ring must be topped up.
// Requests whose eligible classes are all exhausted (or that have none) keep the
// default result in `res`. Wait states (gfx940/gfx950): VALU-written SGPR -> VALU
// read 2, VALU-written VGPR -> DPP read 2, VALU-written VGPR -> v_readlane 1; s[90:93]
// are scratch (the 64-bit lane mask must be an aligned SGPR pair).
__device__ __forceinline__ uint32_t match_fast_loop(
    uint32_t& i, uint32_t cnt, uint32_t mlo, uint32_t mhi, uint32_t slo, uint32_t shi,
    uint64_t holes, uint64_t has_self, uint32_t& res, uint32_t& hp, uint32_t& hg, uint32_t& np,
    uint32_t& ng, uint32_t& cur, uint32_t& left, uint32_t base, uint32_t rmask, uint32_t low,
    uint32_t& win) {
  uint32_t status, c, t, a, mn, tk, sl, s0, s1, m0save;
  asm volatile(
      "s_mov_b32 %[m0s], m0\n"
      "s_mov_b32 %[st], 0\n"
      "s_nop 3\n"
      "L_loop%=:\n"
      "s_cmp_ge_u32 %[i], %[cnt]\n"
      "s_cbranch_scc1 L_out%=\n"
      "v_readlane_b32 s90, %[mlo], %[i]\n"
      "v_readlane_b32 s91, %[mhi], %[i]\n"
      "s_and_b64 s[92:93], s[90:91], %[holes]\n"
      "s_cbranch_scc1 L_slow%=\n"
      "s_bitcmp1_b64 %[hs], %[i]\n"
      "s_cbranch_scc1 L_self%=\n"
      "L_cont%=:\n"
      "v_cndmask_b32 %[c], -1, %[hp], s[90:91]\n"
      "v_mov_b32 %[t], %[c]\n"
      "s_nop 1\n"
      "v_min_u32_dpp %[t], %[t], %[t] row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_min_u32_dpp %[t], %[t], %[t] row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_min_u32_dpp %[t], %[t], %[t] row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_min_u32_dpp %[t], %[t], %[t] row_shr:8 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_min_u32_dpp %[t], %[t], %[t] row_bcast:15 row_mask:0xa bank_mask:0xf\n"
      "s_nop 1\n"
      "v_min_u32_dpp %[t], %[t], %[t] row_bcast:31 row_mask:0xc bank_mask:0xf\n"
      "s_nop 0\n"
      "v_readlane_b32 %[mn], %[t], 63\n"
      "s_cmp_eq_u32 %[mn], -1\n"
      "s_cbranch_scc1 L_tmo%=\n"
      "v_cmp_eq_u32 vcc, %[mn], %[c]\n"
      "s_ff1_i32_b64 %[win], vcc\n"
      "v_readlane_b32 %[tk], %[hg], %[win]\n"
      "s_mov_b32 m0, %[i]\n"
      "s_nop 0\n"
      "v_writelane_b32 %[res], %[tk], m0\n"
      "s_mov_b64 exec, vcc\n"
      "v_add_u32 %[cur], 1, %[cur]\n"
      "v_add_u32 %[a], 1, %[cur]\n"
      "v_and_b32 %[a], %[rmask], %[a]\n"
      "v_lshl_add_u32 %[a], %[a], 2, %[base]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_mov_b32 %[hp], %[np]\n"
      "v_mov_b32 %[hg], %[ng]\n"
      "ds_read_b32 %[np], %[a]\n"
      "ds_read_b32 %[ng], %[a] offset:8192\n"
      "v_subrev_u32 %[left], 1, %[left]\n"
      "s_mov_b64 exec, -1\n"
      "s_add_u32 %[i], %[i], 1\n"
      "v_readlane_b32 %[sl], %[left], %[win]\n"
      "s_cmp_le_u32 %[sl], %[low]\n"
      "s_cbranch_scc0 L_loop%=\n"
      "s_mov_b32 %[st], 2\n"
      "s_branch L_out%=\n"
      "L_tmo%=:\n"
      "s_bitcmp1_b64 %[hs], %[i]\n"
      "s_cbranch_scc1 L_slow%=\n"
      "s_add_u32 %[i], %[i], 1\n"
      "s_branch L_loop%=\n"
      "L_self%=:\n"
      "v_readlane_b32 %[s0], %[slo], %[i]\n"
      "v_readlane_b32 %[s1], %[shi], %[i]\n"
      "s_sub_u32 %[s1], %[s1], %[s0]\n"
      "v_subrev_u32 %[a], %[s0], %[hg]\n"
      "v_cmp_gt_u32 vcc, %[s1], %[a]\n"
      "s_and_b64 s[92:93], vcc, s[90:91]\n"
      "s_cbranch_scc0 L_cont%=\n"
      "L_slow%=:\n"
      "s_mov_b32 %[st], 1\n"
      "L_out%=:\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_mov_b32 m0, %[m0s]\n"
      : [st] "=&s"(status), [i] "+s"(i), [res] "+v"(res), [hp] "+v"(hp), [hg] "+v"(hg),
        [np] "+v"(np), [ng] "+v"(ng), [cur] "+v"(cur), [left] "+v"(left), [win] "+s"(win),
        [c] "=&v"(c), [t] "=&v"(t), [a] "=&v"(a), [mn] "=&s"(mn), [tk] "=&s"(tk), [sl] "=&s"(sl),
        [s0] "=&s"(s0), [s1] "=&s"(s1), [m0s] "=&s"(m0save)
      : [cnt] "s"(cnt), [mlo] "v"(mlo), [mhi] "v"(mhi), [slo] "v"(slo), [shi] "v"(shi),
        [holes] "s"(holes), [hs] "s"(has_self), [base] "v"(base), [rmask] "s"(rmask), [low] "s"(low)
      : "vcc", "scc", "memory", "s90", "s91", "s92", "s93");
  return status;
}

template <int W>
__global__ __launch_bounds__(64) void k_match_pass(ClassLists L, TaskTable T, uint32_t n_tasks,
                                                   uint32_t chunk_size, uint32_t n_chunks,
                                                   MatchBuffers B, uint32_t pass, uint32_t stamp,
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

  // Fixed layout: ranks in the first 8 KB, generation indexes 8 KB further (the asm loop
  // addresses the second array with an immediate offset). C << rshift <= 2048.
  MatchWave<W> w{L, lds_ring, lds_ring + 2048, (1u << rshift) - 1, rshift, lane, {}};
  const uint32_t R = 1u << rshift;
  // A class is topped up when fewer than low_water (>= 3) entries are left in its
  // ring, so the entry after next is always there when a lane advances.
  const uint32_t low_water = R / 4 < 3 ? 3 : (R / 4 > 16 ? 16 : R / 4);

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
      uint32_t old = 0;
      if (lane == 0) {
        atomicAdd(&prm->n_changed[pass & (kPassSlots - 1)], 1u);
        old = atomicMax(&B.claim[kc], stamp);
      }
      if (readlane_u32(old, 0) == stamp) return;  // a wave following its chain got here first
    }
  }

  bool ring_ready = false;
  uint64_t holes[W] = {};
  for (;;) {  // chunk kc, then the chunks after it while the end states keep changing
    const uint32_t t0 = kc * chunk_size;
    const uint32_t t1 = min(n_tasks, t0 + chunk_size);
    bool stopped_early = false;
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
            } else {
              const ClassState old = cp[c];
              if (!class_state_equal(old, s)) {
                differs = true;
                cp[c] = s;
              }
            }
          }
        }
        if (pass != 0 && __ballot(differs) == 0) {
          stopped_early = true;
          break;
        }
      }
      if (!ring_ready) {
        // First block that really runs: fill the rings (init_fill entries per class,
        // one coalesced load per class and array, all in flight together).
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const uint32_t nc = C > 64u * j ? min(64u, C - 64u * j) : 0u;
          for (uint32_t cc = 0; cc < nc; ++cc) {
            const uint32_t from = readlane_u32(w.k[j].cursor, cc);
            const uint32_t to = from + init_fill;
            w.fill(cc + 64 * j, from, to, readlane_u32(w.k[j].end, cc));
            if (lane == cc) w.k[j].filled = to;
          }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < W; ++j) {
          w.load_heads(j);
          holes[j] = __ballot(w.k[j].lo < w.k[j].cursor);
        }
        ring_ready = true;
      }

      // ---- stage the block's requests: lane l holds request tb + l ----
      uint32_t mlo[W], mhi[W], slo = kNone, shi = kNone;
      const uint32_t tl = tb + lane;
      uint64_t many = 0;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        uint64_t m = (tl < t1 && (uint32_t)j < T.words) ? T.mask[(size_t)tl * T.words + j] : 0;
        mlo[j] = (uint32_t)m;
        mhi[j] = (uint32_t)(m >> 32);
        many |= m;
      }
      if (tl < t1) {
        slo = T.self_lo[tl];
        shi = T.self_hi[tl];
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
        uint32_t m_cl = 0, m_from = 0, m_to = 0, m_end = 0;
        if (lane == win) {
#pragma unroll
          for (int j = 0; j < W; ++j) {
            if (j == bj) {
              ClassRun r = w.as_run(j);
              const bool cursor_moved = class_consume_state(L, r, bi, self_lo, self_hi);
              LaneClass& q = w.k[j];
              q.cursor = r.cursor;
              q.lo = r.lo;
              q.hown_lo = r.hown_lo;
              q.hown_hi = r.hown_hi;
              if (cursor_moved) {
                // Entries [cursor, filled) are still in the ring unless the cursor jumped
                // past them; top the ring up either way.
                if (q.filled < q.cursor) q.filled = q.cursor;
                moved = true;
                m_cl = lane + 64 * j;
                m_from = q.filled;
                m_to = min(q.filled + 64u, q.cursor + R);
                m_end = q.end;
                q.filled = m_to;
              }
            }
          }
        }
        if (__ballot(moved)) {
          w.fill(readlane_u32(m_cl, win), readlane_u32(m_from, win), readlane_u32(m_to, win),
                 readlane_u32(m_end, win));
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
      // Tops up the ring of class j of lane `win` (which has just picked from it).
      auto refill = [&](uint32_t win, int bj) {
        uint32_t f_cl = 0, f_from = 0, f_to = 0, f_end = 0;
        if (lane == win) {
#pragma unroll
          for (int j = 0; j < W; ++j) {
            if (j == bj) {
              LaneClass& q = w.k[j];
              f_cl = lane + 64 * j;
              f_from = q.filled;
              f_to = min(q.filled + 64u, q.cursor + R);
              f_end = q.end;
              q.filled = f_to;
            }
          }
        }
        w.fill(readlane_u32(f_cl, win), readlane_u32(f_from, win), readlane_u32(f_to, win),
               readlane_u32(f_end, win));
        __builtin_amdgcn_wave_barrier();
      };

      if constexpr (W == 1) {
        LaneClass& q = w.k[0];
        const uint32_t base = (uint32_t)(uintptr_t)lds_ring + ((lane << rshift) << 2);
        uint32_t i = 0, win = 0;
        for (;;) {
          uint32_t left = w.ring_left(0);
          const uint32_t st =
              match_fast_loop(i, cnt, mlo[0], mhi[0], slo, shi, holes[0], has_self, res, q.head_p,
                              q.head_g, q.next_p, q.next_g, q.cursor, left, base, w.rmask,
                              low_water, win);
          // Classes without holes were advanced with lo == cursor.
          if (!((holes[0] >> lane) & 1)) q.lo = q.cursor;
          if (st == 0) break;
          if (st == 2) {
            refill(win, 0);
          } else {
            general_step(i);
            ++i;
          }
        }
      } else {
        for (uint32_t i = 0; i < cnt; ++i) {
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
              uint32_t left = kNone;
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
                    left = w.ring_left(j);
                  }
                }
              }
              if (readlane_u32(left, win) <= low_water) refill(win, (int)readlane_u32((uint32_t)bj, win));
              continue;
            }
            // Every eligible class is exhausted: Timeout (task_dispatcher.cc:116-118 with
            // timeout == now), unless the requestor's own servant may still serve it.
            if (!self) continue;
          }
          general_step(i);
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
    uint32_t old = 0;
    if (lane == 0) old = atomicMax(&B.claim[kc], stamp);
    if (readlane_u32(old, 0) == stamp) return;
  }
}

}  // namespace ydc
#endif  // YADCC_AMD_MATCH_KERNEL_H_

// match_kernel.h — k_match_pass: the greedy merge of the pending requests over
// the per-class slot lists, chunk-parallel and speculative.
//
// One wave (= one workgroup) per chunk of requests, one lane per servant class
// (W classes per lane above 64).
//
//   pass 0   every chunk is replayed from its level guess (worked out here, or by
//            k_guess_init) and leaves: the slot of every request, a checkpoint of the
//            class states before every block of 64 requests and 16 requests into
//            the chunk, its end state.
//   pass r   chunk k is *consistent* when the state it was last replayed from
//            (its first checkpoint) equals the end state of chunk k-1. An
//            inconsistent chunk is replayed from that end state; the replay
//            stops as soon as the states equal a checkpoint of the previous
//            replay (identical remainder). If its own end state changed, the
//            wave carries on into chunk k+1 unless another wave has claimed it
//            in this pass; from pass 2 on a chunk behind an inconsistent chunk
//            leaves itself to that follower — a perturbation that needs thousands
//            of requests to die out is followed by one wave in one launch instead
//            of one launch per chunk.
//   pass 0 + 1 (one GPU): the launch of pass 0 does pass 1's work as well — a wave hands the
//            end state of its pass-0 replay to its successor through tagged 8-byte granules
//            (MatchBuffers::hand) and, where its own level guess missed its predecessor's end
//            state, replays its chunk from that state at once. What it leaves behind (flags of
//            pass 1, checkpoints, end states) is what a separate launch of pass 1 would have
//            left, minus the chain following; the host goes on with pass 2 if need be.
//            With <= 64 classes and one part, pass 0 also starts every chunk kWarmUp requests
//            early (from the level guess of that point; picks thrown away): a guess that is off
//            by a slot or two is back on track by the chunk's first request, and hardly any
//            chunk needs the second replay.
//   A pass in which no chunk's end state changed (pass 0: every end state equals the
//   next chunk's level guess) proves that every chunk's last replay started from its
//   predecessor's final end state: the result is the sequential one (chunk 0 always
//   starts from the true state). Such a pass leaves its flag in DeviceParams::
//   n_changed at 0. The host pre-launches a few passes; a pass returns at once when
//   an earlier one already was final, and k_finalize only runs behind a final pass.
//
// End states are updated in place. A wave that reads its predecessor's end
// state while that is being rewritten replays from a mixed (meaningless but
// harmless) state; the inconsistency shows in the next pass.
//
// The pick of one request depends on the previous one, so the inner loop is a
// dependency chain. A wave issues one instruction every ~4.5 cycles whatever its kind and
// however many waves share the SIMD, and stalls ~18 cycles wherever the scalar side waits for
// the vector side (tests/tools/issue_probe.hip): what counts is the number of instructions
// and of such hops per request, and the number of resident waves. The loop works on
// *inverted* ranks (~rank, 0 = no slot) so that the winner is a DPP max with zero fill (no
// identity moves), the request's class mask is used directly as a lane mask (v_cndmask with
// an SGPR pair), the result of a request is its slot's global rank (the reduced value itself;
// k_finalize maps rank -> slot -> servant), and the winning lane advances from registers
// (head / next kept in VGPRs, the LDS read of the entry after next is only waited for when the
// lane wins again). About 21 instructions per request at 4 classes, 29 at 32; with >= 5
// classes two requests share an iteration when their winners are different lanes.
#ifndef YADCC_AMD_MATCH_KERNEL_H_
#define YADCC_AMD_MATCH_KERNEL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dispatch_core.h"

namespace ydc {

constexpr uint32_t kPassSlots = 64;  // DeviceParams::n_changed is indexed by pass & 63

// Phase stamps of the matching kernel (measurement builds only: `make probe`,
// tools/phase_probe.py): lane 0 of every wave leaves the 100 MHz wall clock at its phase
// boundaries. Compiled out of the product library.
#ifdef YDC_PHASE_PROBE
#define YDC_PROBE(chunk, slot)                                                          \
  do {                                                                                  \
    if (threadIdx.x == 0 && (chunk) < kProbeChunks)                                     \
      ydc_phase_probe[(size_t)(chunk) * kProbeSlots + (slot)] = wall_clock64();         \
  } while (0)
// Accumulated inside the block loop of a chunk's first replay: ticks spent topping rings up and
// in the fast loop (low words), calls of the fast loop and general steps (high words).
#define YDC_PROBE_ACC(var, expr)                                                        \
  do {                                                                                  \
    const uint64_t ydc_t0 = wall_clock64();                                             \
    expr;                                                                               \
    var += wall_clock64() - ydc_t0;                                                     \
  } while (0)
#define YDC_PROBE_COUNT(var) (var += 1ull << 32)
#define YDC_PROBE_PUT(chunk, slot, v)                                                   \
  do {                                                                                  \
    if (threadIdx.x == 0 && (chunk) < kProbeChunks)                                     \
      ydc_phase_probe[(size_t)(chunk) * kProbeSlots + (slot)] = (v);                    \
  } while (0)
#else
#define YDC_PROBE(chunk, slot) do { (void)(chunk); } while (0)
#define YDC_PROBE_ACC(var, expr) do { expr; } while (0)
#define YDC_PROBE_COUNT(var) do { } while (0)
#define YDC_PROBE_PUT(chunk, slot, v) do { (void)(chunk); } while (0)
#endif

struct MatchBuffers {
  const ClassState* guess0;  // [K * C] level guesses (start states of pass 0)
  // Non-NULL (<= 64 classes, single GPU): pass 0 works its level guesses out itself from the
  // prefix of the chunks' consuming counts, and guess0 is not used. Counts are kept per part
  // of the registry (kernels.h: PartTable): before[chunk * n_parts + part].
  const uint32_t* before;
  const uint32_t* cls_comp;        // part of every class (n_parts > 1 only)
  const uint32_t* part_rank_base;  // first global rank of every part (n_parts > 1 only)
  uint32_t n_parts;
  // Multi-GPU with own guesses: the consuming-request counts of all ranks (gathered,
  // [rank * base_stride + part]); the guesses of rank base_rank start behind those of the ranks
  // before it. NULL on one GPU.
  const uint32_t* base_totals;
  uint32_t base_rank, base_stride;
  ClassState* endst;         // [K * C] end state of every chunk (in place)
  ClassState* checkpoint;    // [ceil(N / 64) * C] state before each block of 64 requests
  // [K * C] state after the first kEarlyAt requests of every chunk (<= 64 classes only): a
  // replay whose start was off by a slot or two is back on its previous track within a few
  // requests, and stops there instead of at the end of the block.
  ClassState* early;
  unsigned long long* claim; // [K] (batch, pass) stamp of the pass in which a wave took the chunk
  uint32_t* slot_of;         // [N] global rank of the slot each request takes (or kIdx*)
  // Multi-GPU: this rank's chunks continue the previous rank's. boundary_in (C
  // entries, or NULL) is the end state of the predecessor's last chunk.
  const ClassState* boundary_in;
  uint32_t has_successor;  // multi-GPU: another rank continues after this rank's last chunk
  // "pass r changed an end state" flags (index r & flag_mask): DeviceParams::n_changed.
  uint32_t* flags;
  uint32_t* sampled;  // sampled counts of changed end states, same indexing
  uint32_t flag_mask;
  // Non-NULL (one GPU): the launch of pass 0 is pass 1 as well. [K * C * 4] 8-byte granules
  // {state word, batch stamp}: a wave publishes the end state of its pass-0 replay here, waits
  // for its predecessor's, and — where its level guess was off — replays its chunk from that
  // state at once, the way the next launch would have (k_match_pass, "pass 0 + 1").
  unsigned long long* hand;
  // Non-NULL (with `hand`, one part, <= 64 classes): consuming requests among the last kWarmUp
  // requests of every chunk. Pass 0 then starts every chunk kWarmUp requests early, from the
  // level guess of THAT point, and throws the warm-up picks away: a level guess that is off by
  // a slot or two is back on the true track within a few requests, so the state it reaches at
  // the chunk's first request is almost always its predecessor's end state — and the chunk
  // needs no second replay.
  const uint32_t* tail;
  // Non-NULL (bin sort, one part): level table, entry (m, c) = position in class c's list of its
  // first slot whose global rank is >= 64 m (bin_sort.h). With <= 8 classes pass 0 turns a level
  // into class cursors with one lookup and one 64-entry window per class instead of a search.
  const uint32_t* level_tab;
  // Non-NULL (radix path, the class partition was a pass of its own): that pass's scanned
  // histogram table — entry [c * tile_tab_tiles + t] = slots of class c among the first
  // t * tile_tab_elems slots of the global order, i.e. a level table at tile granularity that
  // the sort leaves behind anyway (kernels.h: k_radix_scan). A lane's two level searches then run
  // over the class's entries of ONE tile (a few dozen) instead of its whole list.
  const uint32_t* tile_tab;
  uint32_t tile_tab_tiles, tile_tab_elems;
  uint32_t warm_len;  // the warm-up's length in requests (kWarmUp unless tuned; <= 64)
  // Checkpoints are taken before every cp_every-th block of 64 requests of a chunk (a power of
  // two; the chunk's first block always: it says what the chunk was replayed from). Every block
  // was round 5's choice: 16 B x C per 64 requests, 7.5 MB of a 1M-request batch with 30 classes,
  // written by every batch and read by the few chunks that are replayed a second time — whose
  // early stop mostly comes 16 requests in (`early`). Rows of blocks that are not checkpointed are
  // never read.
  uint32_t cp_every;
  uint32_t hand_tries;  // polls for the predecessor's granules before giving up (kHandTries; tests: 0)
  // Non-NULL (zone_guess.h): k_zone_guess walks the stretch where the dedicated tier runs out
  // as workgroup 0 of the launch of pass 0 (the chunks are the workgroups behind it) and publishes, as granules {word, batch number}, which chunks
  // that is ([0], [1]) and the class cursors their replays start from ([2 + row * C + class]);
  // those chunks wait for their row instead of using their level guess.
  unsigned long long* zone_box;
  const uint2* zone_sorted;   // the key-sorted records {key, value}: where tier 1 begins
  uint32_t zone_tier_shift;   // key >> zone_tier_shift != 0: tier 1
  uint32_t zone_lead, zone_trail;  // the stretch: levels [T0 - lead, T0 + trail)
};

constexpr uint32_t kWarmUp = 16;

constexpr uint32_t kZoneHeaderTries = 40;   // ~35 us
constexpr uint32_t kZoneRowTries = 4000;    // ~5 ms: the walk is running once its header is there
constexpr uint32_t kHandTries = 1500;  // polls for the predecessor's granules before giving up

constexpr uint32_t kEarlyAt = 16;  // requests into a chunk at which MatchBuffers::early is taken

// Everything a lane keeps about one of its classes.
struct LaneClass {
  uint32_t cursor, lo, hown_lo, hown_hi, end;
  uint32_t hq;      // ~rank of the entry at `cursor`     (0 past the end)
  uint32_t nq;      // ~rank of the entry at `cursor + 1` (0 past the end)
  uint32_t filled;  // ring holds list entries [cursor, filled)
  uint32_t single;  // the class consists of one servant
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
    r.single = k[j].single;
    r.head_p = ~k[j].hq;  // 0 -> kNone
    r.head_g = k[j].cursor < k[j].end ? ring_g[at(lane + 64 * j, k[j].cursor)] : kNone;
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
  // Generation index of the head of this lane's class j (kNone past the end).
  __device__ __forceinline__ uint32_t head_g(int j) const {
    return k[j].cursor < k[j].end ? ring_g[at(lane + 64 * j, k[j].cursor)] : kNone;
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
      q.single = L.cls_single ? L.cls_single[c] : 0u;
    } else {
      q.cursor = q.lo = q.end = 0;
      q.hown_lo = q.hown_hi = kNone;
      q.single = 0;
    }
    q.hq = q.nq = 0;
    q.filled = q.cursor;
  }
  // Wave-uniform: all lanes load list entries [from, to) of class cl (to - from <= 64).
  // ring_p holds ~rank. Positions at or after the end of the class's list (`end`) get the
  // sentinel 0 ("no slot"), so (head, next) read past the end need no compare.
  __device__ __forceinline__ void fill(uint32_t cl, uint32_t from, uint32_t to, uint32_t end) {
    const uint32_t e = from + lane;
    if (e < to) {
      const bool real = e < end;
      ring_p[at(cl, e)] = real ? ~list_rank(L, e) : 0u;
      ring_g[at(cl, e)] = real ? list_slot(L, e) : kNone;
    }
  }
  // First fill, n (<= R) entries per class. Few classes (C * ceil(n / 64) <= 16): one
  // coalesced 64-entry load per class and 64 entries, all of them in flight before the
  // first LDS store — one memory round trip. Otherwise every lane loads the entries of its
  // own classes, 16 independent loads in flight per lane and array.
  __device__ __forceinline__ void init_rings(uint32_t C, uint32_t n, int W_) {
    const uint32_t per = (n + 63) / 64;
    if (W_ == 1 && C * per <= 16) {
      uint32_t tp[16], tg[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        tp[u] = 0;
        tg[u] = kNone;
        if ((uint32_t)u < C * per) {
          const uint32_t cc = (uint32_t)u / per, part = (uint32_t)u % per;
          const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)k[0].cursor, (int)cc);
          const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)k[0].end, (int)cc);
          const uint32_t e = cur + part * 64 + lane;
          if (e < end && part * 64 + lane < n) {
            tp[u] = ~list_rank(L, e);
            tg[u] = list_slot(L, e);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if ((uint32_t)u < C * per) {
          const uint32_t cc = (uint32_t)u / per, part = (uint32_t)u % per;
          const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)k[0].cursor, (int)cc);
          if (part * 64 + lane < n) {
            ring_p[at(cc, cur + part * 64 + lane)] = tp[u];
            ring_g[at(cc, cur + part * 64 + lane)] = tg[u];
          }
        }
      }
      if (lane < C) k[0].filled = k[0].cursor + n;
      return;
    }
    if (W_ == 1 && n <= 64) {
      // One class per lane, up to 64 entries each: the wave loads the classes' entries together,
      // sixteen classes per round trip, one coalesced 64-entry load per class (a lane fetching
      // its own class's entries sixteen at a time took four dependent round trips of scattered
      // 8-byte reads: 22 of the 31 us a cfg3 wave spent before its first pick).
      for (uint32_t c0 = 0; c0 < C; c0 += 16) {
        uint32_t tp[16], tg[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          tp[u] = 0;
          tg[u] = kNone;
          if (c0 + u < C) {
            const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)k[0].cursor, (int)(c0 + u));
            const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)k[0].end, (int)(c0 + u));
            const uint32_t e = cur + lane;
            if (lane < n && e < end) {
              tp[u] = ~list_rank(L, e);
              tg[u] = list_slot(L, e);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (c0 + u < C && lane < n) {
            const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)k[0].cursor, (int)(c0 + u));
            ring_p[at(c0 + u, cur + lane)] = tp[u];
            ring_g[at(c0 + u, cur + lane)] = tg[u];
          }
        }
      }
      if (lane < C) k[0].filled = k[0].cursor + n;
      return;
    }
    for (int j = 0; j < W_; ++j) {
      const uint32_t cl = lane + 64 * j;
      LaneClass& q = k[j];
      if (cl < C) {
        for (uint32_t b0 = q.cursor; b0 < q.cursor + n; b0 += 16) {
          uint32_t tp[16], tg[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const uint32_t e = b0 + u;
            tp[u] = e < q.end ? ~list_rank(L, e) : 0u;
            tg[u] = e < q.end ? list_slot(L, e) : kNone;
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
    q.hq = ring_p[at(cl, q.cursor)];
    q.nq = ring_p[at(cl, q.cursor + 1)];
  }
};

// bit `lane` of m ? v : 0 — the request's class mask used as a lane mask.
__device__ __forceinline__ uint32_t select_by_lane_mask(uint64_t m, uint32_t v) {
  uint32_t out;
  asm volatile("v_cndmask_b32 %0, 0, %1, %2" : "=v"(out) : "v"(v), "s"(m));
  return out;
}

// Maximum over the 64 lanes (every lane gets it).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, dpp_u32<0x111>(0u, v));       // row_shr:1
  v = max(v, dpp_u32<0x112>(0u, v));       // row_shr:2
  v = max(v, dpp_u32<0x114>(0u, v));       // row_shr:4
  v = max(v, dpp_u32<0x118>(0u, v));       // row_shr:8
  v = max(v, dpp_u32<0x142, 0xa>(0u, v));  // row_bcast:15
  v = max(v, dpp_u32<0x143, 0xc>(0u, v));  // row_bcast:31
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, uint32_t l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}


// The fast loop of one block of (up to 64) requests, W == 1, hand-scheduled. Runs requests
// i, i+1, ... < end (i < end) while each of them is "plain". Returns
//   0  all of them done,
//   1  request i needs the general step (the requests before it were done),
//   2  a ring may run dry within the next few picks: top up, come back for request i.
// `special` marks the requests of the block that need a look first: `hole_hit` ones (an
// eligible class has holes) leave at once; `has_self` ones leave only if an eligible class
// shows a slot of the requestor's own servant at its head, or if nothing is left for them
// (last-resort self pick). Requests whose eligible classes are all exhausted leave their
// lane of `raw` alone; a served request gets ~(global rank of its slot) there (never 0; the
// caller turns it into the rank once per block). The caller guarantees that every ring holds
// at least end - i + 2 entries beyond its cursor.
//
// What bounds the loop is the number of instructions per request, whatever their kind
// (a wave issues one every 4-5 cycles): the loop control and the "is the next request
// plain" test are one bit test on a mask worked out before the loop (`go`: plain and below
// `end`, bit 0 clear so that index 64 — which wraps to bit 0 — ends the loop too; `pairok`:
// both i and i + 1 are), the requests' masks come from copies of the mask register shifted
// by one, two and three lanes (the lane select is the request index itself: no second
// counter), and results are written as they come out of the reduction.
//
// Registers: hq / nq = ~rank of head / next of the lane's class (0: none), `an` = LDS address
// of `next` in the lane's ring of ring_p (rings are aligned to their size, so stepping is an
// add + bit-field insert; the cursor is recovered from `an` by the caller), ring_g `goff` bytes
// further (behind all of ring_p). m0 = i, s[90:91] = class mask of request i (fetched one
// request ahead, in the wait states of the DPP chain), s[94:97] scratch. One loop body per DPP
// depth (2^steps >= classes).
// Wait states (gfx940/gfx950): VALU-written SGPR -> VALU read 2, VALU-written VGPR ->
// DPP read 2, VALU-written VGPR -> v_readlane 1.
#define YDC_MAXDPP(first, ctrl) "v_max_u32_dpp %[t], " first " " ctrl " bound_ctrl:0\n"
#define YDC_GAP "s_nop 1\n"
// NZ: the lanes that offer the request anything at all (eligible, not exhausted), worked out in
// a wait state of the reduction. The winner is looked for among them only, so a request nobody
// can serve (maximum 0 = "no slot") selects no lane, advances nothing and writes 0 = "not
// served" — the common path needs no test for it (a VALU -> SGPR -> compare -> branch hop costs
// ~40 cycles, tests/tools/issue_probe.hip; a plain instruction 4.5).
#define YDC_NZ "v_cmp_ne_u32_e64 s[94:95], 0, %[c]\n"
#define YDC_RED_1(K) YDC_MAXDPP("%[c], %[c]", "row_shr:1 row_mask:0xf bank_mask:0xf") YDC_NZ
// HI: how the upper mask word of the next request gets into s91 in the gap behind the first
// step — a v_readlane, or nothing at all with <= 32 classes (the word is always 0 and s91
// stays 0: one VALU instruction less per request).
#define YDC_HI_READ "v_readlane_b32 s91, %[m1hi], m0\n" YDC_NZ
#define YDC_HI_ZERO YDC_NZ "s_nop 0\n"
#define YDC_RED_2(K, HI) YDC_MAXDPP("%[c], %[c]", "row_shr:1 row_mask:0xf bank_mask:0xf") HI \
  YDC_MAXDPP("%[t], %[t]", "row_shr:2 row_mask:0xf bank_mask:0xf") "s_nop 0\n"
#define YDC_RED_3(K, HI) YDC_MAXDPP("%[c], %[c]", "row_shr:1 row_mask:0xf bank_mask:0xf") HI \
  YDC_MAXDPP("%[t], %[t]", "row_shr:2 row_mask:0xf bank_mask:0xf")                           \
  YDC_GAP YDC_MAXDPP("%[t], %[t]", "row_shr:4 row_mask:0xf bank_mask:0xf")
#define YDC_RED_4(K, HI) YDC_RED_3(K, HI) YDC_GAP YDC_MAXDPP("%[t], %[t]", "row_shr:8 row_mask:0xf bank_mask:0xf")
#define YDC_RED_5(K, HI) YDC_RED_4(K, HI) YDC_GAP YDC_MAXDPP("%[t], %[t]", "row_bcast:15 row_mask:0xa bank_mask:0xf")
#define YDC_RED_6(K, HI) YDC_RED_5(K, HI) YDC_GAP YDC_MAXDPP("%[t], %[t]", "row_bcast:31 row_mask:0xc bank_mask:0xf")
#define YDC_TAIL_NOP "s_nop 0\n"

// The winning lane(s) advance (exec = winners; the read of `next` issued by the lane's
// previous win has been waited for).
#define YDC_ADVANCE                                                                   \
  ".if %c[ck]\n"                                                                      \
  "v_add_u32 %[left], -1, %[left]\n"                                                  \
  ".endif\n"                                                                          \
  "v_mov_b32 %[hq], %[nq]\n"                                                          \
  "v_add_u32 %[a], 4, %[an]\n"                                                        \
  "v_bfi_b32 %[an], %[rmask4], %[a], %[an]\n"                                         \
  "ds_read_b32 %[nq], %[an]\n"                                                        \
  "s_mov_b64 exec, -1\n"

// Two requests per iteration (>= 5 classes): the two selections and DPP chains are
// independent until the winners are known, so they fill each other's wait states — and the
// slots left over fetch the masks of the two requests after them (a pair that follows a pair
// starts with both masks in place) and work out who offers each of the two anything. When the
// winners are different lanes both advance under one exec mask, otherwise only the first
// request is committed and the second one starts the next iteration. Plain requests only.
// s[92:93] = class mask of the second request, s[94:95] = its winner, s[98:99] / s[88:89] =
// the lanes with something to offer to the first / second; %[s0] doubles as the second
// request's maximum.
#define YDC_MAX2(dst, first, ctrl) "v_max_u32_dpp " dst ", " first " " ctrl " bound_ctrl:0\n"
#define YDC_P1                                                                        \
  YDC_MAX2("%[t]", "%[c], %[c]", "row_shr:1 row_mask:0xf bank_mask:0xf")              \
  "v_readlane_b32 s92, %[m3lo], m0\n"                                                 \
  YDC_MAX2("%[t1]", "%[c1], %[c1]", "row_shr:1 row_mask:0xf bank_mask:0xf")
#define YDC_PN(ctrl, FILL)                                                            \
  YDC_MAX2("%[t]", "%[t], %[t]", ctrl) FILL YDC_MAX2("%[t1]", "%[t1], %[t1]", ctrl)
#define YDC_PF_NOP "s_nop 0\n"
#define YDC_PF_HI91 "v_readlane_b32 s91, %[m2hi], m0\n"
#define YDC_PF_HI93 "v_readlane_b32 s93, %[m3hi], m0\n"
#define YDC_PF_NZ0 "v_cmp_ne_u32_e64 s[98:99], 0, %[c]\n"
#define YDC_PF_NZ1 "v_cmp_ne_u32_e64 s[88:89], 0, %[c1]\n"
#define YDC_PRED_3 YDC_P1 YDC_PN("row_shr:2 row_mask:0xf bank_mask:0xf", YDC_PF_NZ0)   \
  YDC_PN("row_shr:4 row_mask:0xf bank_mask:0xf", YDC_PF_NZ1)
#define YDC_PRED_4 YDC_PRED_3 YDC_PN("row_shr:8 row_mask:0xf bank_mask:0xf", YDC_PF_NOP)
#define YDC_PRED_5 YDC_PRED_4 YDC_PN("row_bcast:15 row_mask:0xa bank_mask:0xf", YDC_PF_NOP)
#define YDC_PRED_6 YDC_P1 YDC_PN("row_shr:2 row_mask:0xf bank_mask:0xf", YDC_PF_HI91)   \
  YDC_PN("row_shr:4 row_mask:0xf bank_mask:0xf", YDC_PF_HI93)                           \
  YDC_PN("row_shr:8 row_mask:0xf bank_mask:0xf", YDC_PF_NZ0)                            \
  YDC_PN("row_bcast:15 row_mask:0xa bank_mask:0xf", YDC_PF_NZ1)                         \
  YDC_PN("row_bcast:31 row_mask:0xc bank_mask:0xf", YDC_PF_NOP)
// How a pair that does not follow a pair gets its second mask (two wait states before its use).
#define YDC_PAIRPRE_LO "v_readlane_b32 s92, %[m1lo], m0\ns_nop 0\n"
#define YDC_PAIRPRE_HI "v_readlane_b32 s92, %[m1lo], m0\nv_readlane_b32 s93, %[m1hi], m0\ns_nop 0\n"
#define YDC_REREAD_LO ""
#define YDC_REREAD_HI "v_readlane_b32 s91, %[mhi], m0\n"
#define YDC_PAIRTEST(K)                                                               \
  "s_bitcmp1_b64 %[pairok], m0\n"                                                     \
  "s_cbranch_scc1 L" #K "_pairbody%=\n"
#define YDC_PAIRENTRY(K)                                                              \
  "s_cmp_eq_u32 %[entry], 2\n"                                                        \
  "s_cbranch_scc1 L" #K "_pairbody%=\n"
#define YDC_PAIR(K, PRED, LASTLANE, PAIRPRE, REREAD)                                  \
  "L" #K "_pairbody%=:\n" PAIRPRE                                                     \
  "L" #K "_pairchain%=:\n"                                                            \
  "v_cndmask_b32 %[c], 0, %[hq], s[90:91]\n"                                          \
  "v_cndmask_b32 %[c1], 0, %[hq], s[92:93]\n"                                         \
  "v_readlane_b32 s90, %[m2lo], m0\n" PRED                                            \
  "v_readlane_b32 %[mn], %[t], " LASTLANE "\n"                                        \
  "v_readlane_b32 %[s0], %[t1], " LASTLANE "\n"                                       \
  "s_waitcnt lgkmcnt(0)\n"                                                            \
  "v_cmp_eq_u32 vcc, %[mn], %[c]\n"                                                   \
  "v_cmp_eq_u32_e64 s[94:95], %[s0], %[c1]\n"                                         \
  "v_writelane_b32 %[raw], %[mn], m0\n"                                               \
  "s_add_u32 m0, m0, 1\n"                                                             \
  "v_writelane_b32 %[raw], %[s0], m0\n"                                               \
  "s_add_u32 m0, m0, 1\n"                                                             \
  "s_and_b64 vcc, vcc, s[98:99]\n"                                                    \
  "s_and_b64 s[94:95], s[94:95], s[88:89]\n"                                          \
  "s_and_b64 s[96:97], vcc, s[94:95]\n"                                               \
  "s_cbranch_scc1 L" #K "_conflict%=\n"                                               \
  "s_or_b64 exec, vcc, s[94:95]\n" YDC_ADVANCE                                        \
  "s_bitcmp1_b64 %[pairok], m0\n"                                                     \
  "s_cbranch_scc1 L" #K "_pairchain%=\n"                                              \
  "s_bitcmp1_b64 %[go], m0\n"                                                         \
  "s_cbranch_scc1 L" #K "_cont%=\n"                                                   \
  "s_branch L" #K "_check%=\n"                                                        \
  "L" #K "_conflict%=:\n"                                                             \
  "s_add_u32 m0, m0, -1\n"                                                            \
  "s_mov_b64 exec, vcc\n" YDC_ADVANCE                                                 \
  "v_writelane_b32 %[raw], 0, m0\n"                                                   \
  "v_readlane_b32 s90, %[mlo], m0\n" REREAD                                           \
  "s_bitcmp1_b64 %[pairok], m0\n"                                                     \
  "s_cbranch_scc1 L" #K "_pairbody%=\n"                                               \
  "s_nop 0\n"                                                                         \
  "s_branch L" #K "_cont%=\n"

// One request: select, reduce, (ZCHECK: leave if nobody serves it — only the requests that
// came through the `special` test need that: their last resort is the general step), commit.
#define YDC_ZCHECK_NONE ""
#define YDC_ZCHECK(K)                                                                 \
  "s_cmp_eq_u32 %[mn], 0\n"                                                           \
  "s_cbranch_scc1 L" #K "_tmo%=\n"
#define YDC_SINGLE(K, SFX, RED, LASTLANE, ZCHECK, CONTROL)                            \
  "L" #K "_cont" SFX "%=:\n"                                                          \
  "v_cndmask_b32 %[c], 0, %[hq], s[90:91]\n"                                          \
  "s_nop 0\n"                                                                         \
  "v_readlane_b32 s90, %[m1lo], m0\n" RED                                             \
  "v_readlane_b32 %[mn], %[t], " LASTLANE "\n"                                        \
  "s_mov_b64 exec, s[94:95]\n"                                                        \
  "s_waitcnt lgkmcnt(0)\n" ZCHECK                                                     \
  "v_cmp_eq_u32 vcc, %[mn], %[c]\n"                                                   \
  "v_writelane_b32 %[raw], %[mn], m0\n"                                               \
  "s_add_u32 m0, m0, 1\n" CONTROL
// Loop control behind a request. Without pairs the "is the next request plain" test sits in
// the shadow of the winner compare (nothing between it and the branch touches SCC).
#define YDC_CONTROL_PLAIN(K)                                                          \
  "s_bitcmp1_b64 %[go], m0\n"                                                         \
  "s_mov_b64 exec, vcc\n" YDC_ADVANCE                                                 \
  "s_cbranch_scc1 L" #K "_cont%=\n"                                                   \
  "s_branch L" #K "_check%=\n"
#define YDC_CONTROL_PAIRS(K)                                                          \
  "s_mov_b64 exec, vcc\n" YDC_ADVANCE                                                 \
  "s_bitcmp1_b64 %[pairok], m0\n"                                                     \
  "s_cbranch_scc1 L" #K "_pairbody%=\n"                                               \
  "s_bitcmp1_b64 %[go], m0\n"                                                         \
  "s_cbranch_scc1 L" #K "_cont%=\n"                                                   \
  "s_branch L" #K "_check%=\n"

#define YDC_LOOP_BODY(K, RED, LASTLANE, PAIRENTRY, CONTROL, PAIR)                  \
  "L" #K "_entry%=:\n"                                                             \
  "s_cmp_eq_u32 %[entry], 1\n"                                                     \
  "s_cbranch_scc1 L" #K "_cont%=\n" PAIRENTRY                                      \
  "s_branch L" #K "_special%=\n" PAIR                                              \
  YDC_SINGLE(K, "", RED, LASTLANE, YDC_ZCHECK_NONE, CONTROL)                       \
  YDC_SINGLE(K, "_self", RED, LASTLANE, YDC_ZCHECK(K), CONTROL)                    \
  "L" #K "_check%=:\n"                                                             \
  "s_cmp_ge_u32 m0, %[end]\n"                                                      \
  "s_cbranch_scc1 L_out%=\n"                                                       \
  "s_bitcmp1_b64 %[chk], m0\n"                                                     \
  "s_cbranch_scc0 L" #K "_special%=\n"                                             \
  "v_cmp_gt_u32 vcc, %[thr], %[left]\n"                                            \
  "s_cbranch_vccnz L_low%=\n"                                                      \
  "s_bitcmp1_b64 %[plain], m0\n"                                                   \
  "s_cbranch_scc1 L" #K "_cont%=\n"                                                \
  "L" #K "_special%=:\n"                                                           \
  "s_bitcmp1_b64 %[hh], m0\n"                                                      \
  "s_cbranch_scc1 L_slow%=\n"                                                      \
  "v_readlane_b32 %[s0], %[slo], m0\n"                                             \
  "v_readlane_b32 %[s1], %[shi], m0\n"                                             \
  "v_add_u32 %[a], -4, %[an]\n"                                                    \
  "v_bfi_b32 %[a], %[rmask4], %[a], %[an]\n"                                       \
  "v_add_u32 %[a], %[goff], %[a]\n"                                                \
  "ds_read_b32 %[a], %[a]\n"                                                       \
  "s_sub_u32 %[s1], %[s1], %[s0]\n"                                                \
  "s_waitcnt lgkmcnt(0)\n"                                                         \
  "v_subrev_u32 %[a], %[s0], %[a]\n"                                               \
  "v_cmp_gt_u32 vcc, %[s1], %[a]\n"                                                \
  "s_and_b64 s[96:97], vcc, s[90:91]\n"                                            \
  "s_cbranch_scc0 L" #K "_cont_self%=\n"                                           \
  "s_branch L_slow%=\n"                                                            \
  "L" #K "_tmo%=:\n"                                                               \
  "s_mov_b64 exec, -1\n"                                                           \
  "s_branch L_slow%=\n"

// Lane l of the result holds lane l + d of v (d = 1 .. 3; the last lanes get anything).
template <int D>
__device__ __forceinline__ uint32_t lanes_down(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x + D) & 63u) << 2), (int)v);
}

// The masks of a block's requests as the loop wants them: as staged, and shifted down by one,
// two and three lanes.
struct BlockMasks {
  uint32_t lo, hi, lo1, hi1, lo2, hi2, lo3, hi3;
};
__device__ __forceinline__ BlockMasks block_masks(uint32_t mlo, uint32_t mhi, bool wide) {
  BlockMasks m{mlo, mhi, lanes_down<1>(mlo), 0, lanes_down<2>(mlo), 0, lanes_down<3>(mlo), 0};
  if (wide) {  // (more than 32 classes)
    m.hi1 = lanes_down<1>(mhi);
    m.hi2 = lanes_down<2>(mhi);
    m.hi3 = lanes_down<3>(mhi);
  }
  return m;
}

template <bool kChecked>
__device__ __forceinline__ uint32_t match_fast_loop(
    uint32_t& i, uint32_t end, const BlockMasks& m, uint32_t slo, uint32_t shi,
    uint64_t special, uint64_t hole_hit, uint64_t has_self, uint32_t& raw, uint32_t& hq,
    uint32_t& nq, uint32_t& an, uint32_t goff, uint32_t rmask4, uint32_t steps, uint32_t pair,
    uint32_t& left) {
  constexpr uint32_t check_every = kChecked ? 8u : 0u;
  // Plain requests below `end`; index 64 wraps to bit 0, which therefore always says "stop".
  const uint64_t plain = ~special & (end >= 64 ? ~0ull : (1ull << end) - 1);
  // kChecked: the loop looks at the rings itself, before every request whose index is a
  // multiple of 8 — `left` = entries each lane's ring holds beyond its cursor, counted down as the
  // lane wins — and returns 2 when one of them could run dry within the next 8 picks. (Otherwise
  // the caller bounds `end` by what the emptiest ring allows.) With 30 classes sharing 1024 entries
  // that is a return every ~25 requests instead of every ~11 (cfg4).
  const uint64_t chk = check_every ? 0x0101010101010100ull : 0ull;
  const uint32_t thr = check_every + 2;
  const uint64_t go = plain & ~1ull & ~chk;
  const uint64_t pairok = pair ? go & (go >> 1) : 0ull;
  uint32_t entry = (uint32_t)(plain >> i) & 1u;
  if (entry && pair && i < 63 && ((plain >> (i + 1)) & 1)) entry = 2;
  uint32_t status, c, t, c1, t1, a, mn, sp, s0, s1, m0save;
  asm volatile(
      "s_mov_b32 %[m0s], m0\n"
      "s_mov_b32 %[st], 0\n"
      "s_mov_b32 m0, %[i]\n"
      "s_mov_b32 s93, 0\n"
      "s_nop 2\n"
      "v_readlane_b32 s90, %[mlo], m0\n"
      "v_readlane_b32 s91, %[mhi], m0\n"
      "s_cmp_eq_u32 %[steps], 1\n"
      "s_cbranch_scc1 L1_entry%=\n"
      "s_cmp_eq_u32 %[steps], 2\n"
      "s_cbranch_scc1 L2_entry%=\n"
      "s_cmp_eq_u32 %[steps], 3\n"
      "s_cbranch_scc1 L3_entry%=\n"
      "s_cmp_eq_u32 %[steps], 4\n"
      "s_cbranch_scc1 L4_entry%=\n"
      "s_cmp_eq_u32 %[steps], 5\n"
      "s_cbranch_scc1 L5_entry%=\n"
      "s_branch L6_entry%=\n"
      YDC_LOOP_BODY(1, YDC_RED_1(1), "1", "", YDC_CONTROL_PLAIN(1), "")
      YDC_LOOP_BODY(2, YDC_RED_2(2, YDC_HI_ZERO), "3", "", YDC_CONTROL_PLAIN(2), "")
      YDC_LOOP_BODY(3, YDC_RED_3(3, YDC_HI_ZERO) YDC_TAIL_NOP, "7", YDC_PAIRENTRY(3), YDC_CONTROL_PAIRS(3),
                    YDC_PAIR(3, YDC_PRED_3, "7", YDC_PAIRPRE_LO, YDC_REREAD_LO))
      YDC_LOOP_BODY(4, YDC_RED_4(4, YDC_HI_ZERO) YDC_TAIL_NOP, "15", YDC_PAIRENTRY(4), YDC_CONTROL_PAIRS(4),
                    YDC_PAIR(4, YDC_PRED_4, "15", YDC_PAIRPRE_LO, YDC_REREAD_LO))
      YDC_LOOP_BODY(5, YDC_RED_5(5, YDC_HI_ZERO) YDC_TAIL_NOP, "31", YDC_PAIRENTRY(5), YDC_CONTROL_PAIRS(5),
                    YDC_PAIR(5, YDC_PRED_5, "31", YDC_PAIRPRE_LO, YDC_REREAD_LO))
      YDC_LOOP_BODY(6, YDC_RED_6(6, YDC_HI_READ) YDC_TAIL_NOP, "63", YDC_PAIRENTRY(6), YDC_CONTROL_PAIRS(6),
                    YDC_PAIR(6, YDC_PRED_6, "63", YDC_PAIRPRE_HI, YDC_REREAD_HI))
      "L_low%=:\n"
      "s_mov_b32 %[st], 2\n"
      "s_branch L_out%=\n"
      "L_slow%=:\n"
      "s_mov_b32 %[st], 1\n"
      "L_out%=:\n"
      "s_mov_b32 %[i], m0\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_mov_b32 m0, %[m0s]\n"
      : [st] "=&s"(status), [i] "+s"(i), [raw] "+v"(raw), [hq] "+v"(hq), [nq] "+v"(nq),
        [an] "+v"(an), [left] "+v"(left), [c] "=&v"(c), [t] "=&v"(t), [c1] "=&v"(c1), [t1] "=&v"(t1),
        [a] "=&v"(a), [mn] "=&s"(mn),
        [sp] "=&s"(sp), [s0] "=&s"(s0), [s1] "=&s"(s1), [m0s] "=&s"(m0save)
      : [mlo] "v"(m.lo), [mhi] "v"(m.hi), [m1lo] "v"(m.lo1), [m1hi] "v"(m.hi1), [m2lo] "v"(m.lo2),
        [m2hi] "v"(m.hi2), [m3lo] "v"(m.lo3), [m3hi] "v"(m.hi3), [slo] "v"(slo), [shi] "v"(shi),
        [go] "s"(go), [pairok] "s"(pairok), [entry] "s"(entry), [end] "s"(end), [chk] "s"(chk),
        [plain] "s"(plain), [thr] "s"(thr), [ck] "n"(kChecked ? 1 : 0),
        [hh] "s"(hole_hit), [hs] "s"(has_self), [goff] "s"(goff), [rmask4] "s"(rmask4),
        [steps] "s"(steps)
      : "vcc", "scc", "memory", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97",
        "s98", "s99");
  return status;
}

// OCC: waves per SIMD the register allocation leaves room for. A wave's pace does not depend on
// the waves it shares the SIMD with (tests/tools/issue_probe.hip: one instruction every ~4.5
// cycles per wave at 1, 2 or 3 waves per SIMD), so where a batch has more than two chunks per
// SIMD to offer, a fourth resident wave is worth the two dozen registers spilled in the prologue
// (OCC = 4: 128 VGPRs instead of 152).
// CHECKED: rings of 32 entries, watched by the fast loop itself (match_fast_loop<true>).
}  // namespace ydc
#include "zone_guess.h"
namespace ydc {

template <int W, int OCC = 1, bool CHECKED = false>
__global__ __launch_bounds__(64, OCC) void k_match_pass(ClassLists L, TaskTable T, uint32_t n_tasks,
                                                   uint32_t chunk_size, uint32_t n_chunks,
                                                   MatchBuffers B, uint32_t pass_arg,
                                                   uint32_t flags, uint32_t rshift,
                                                   uint32_t init_fill, SharedIpTable shared,
                                                   DeviceParams* prm) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_ring[];
  uint32_t pass = pass_arg;  // (becomes 1 when the launch of pass 0 goes on with pass 1)
  const bool device_check = flags & 1u;  // an earlier consistent pass ends the work
  const bool count_sims = flags & 2u;   // debug: count replays (a same-address atomic each)
  // The host has given up on parallel repair (32 passes and the end states still change: every
  // chunk's true start state differs from any guess — e.g. a hole at every boundary).
  // Two launches replace the rest: a scout — every inconsistent chunk reports its index, the
  // lowest is the frontier (everything below it is final) — and the walk: the frontier's wave
  // replays its chunk from the true state and carries on through every chunk behind it.
  const bool walk_scout = flags & 8u, walk_run = flags & 16u;
  uint32_t kc = blockIdx.x;  // chunk
  if (W == 1 && B.zone_box != nullptr && pass_arg == 0) {
    // (zone_guess.h: one workgroup more than chunks, the first walks the stretch around the end
    // of the dedicated tier for the chunks there)
    if (blockIdx.x == 0) {
      zone_walk(L, T.mask, n_tasks, chunk_size, n_chunks, B, prm, lds_ring, flags >> 8);
      return;
    }
    kc = blockIdx.x - 1;
  }
  const uint32_t probe_kc = kc;
  [[maybe_unused]] uint64_t probe_topup = 0, probe_loop = 0, probe_gen = 0;  // (measurement builds only)
  if (pass_arg == 0) YDC_PROBE(probe_kc, 0);  // entry
  // Everything the wave needs to decide whether it has work, fetched in one round trip.
  const uint32_t batch_seq = prm->batch_seq;
  const uint32_t n_slots_all = prm->n_slots;
  const uint32_t walk_frontier = prm->reserved0;
  const uint32_t prev_changed = device_check && pass > 0 ? B.flags[(pass - 1) & B.flag_mask] : 1u;
  const bool own_guess = W == 1 && pass == 0 && B.before != nullptr;
  const bool warm_mode = own_guess && B.hand != nullptr && B.tail != nullptr && B.n_parts <= 1 &&
                         B.boundary_in == nullptr && chunk_size >= 64;
  // Level of this lane's class before / after the chunk: global rank of the first slot not yet
  // consumed if consumption followed the slot order of the class's part of the registry.
  uint32_t before0 = 0, before1 = 0;
  if (own_guess && kc < n_chunks) {
    const uint32_t G = B.n_parts;
    const uint32_t part = G > 1 && threadIdx.x < L.n_classes ? B.cls_comp[threadIdx.x] : 0u;
    before0 = B.before[(size_t)kc * G + part];
    before1 = B.before[(size_t)(kc + 1) * G + part];
    if (warm_mode && kc > 0) before0 -= B.tail[kc - 1];  // the level kWarmUp requests earlier
    if (B.base_totals) {
      uint32_t base = 0;
      for (uint32_t r = 0; r < B.base_rank; ++r) base += B.base_totals[(size_t)r * B.base_stride + part];
      before0 += base;
      before1 += base;
    }
    if (G > 1) {
      const uint32_t first = B.part_rank_base[part];
      before0 += first;
      before1 += first;
    }
  }
  if (prev_changed == 0) return;  // an earlier pass found every chunk consistent
  const uint32_t lane = threadIdx.x;
  if (kc >= n_chunks) return;
  const uint32_t C = L.n_classes;
  const bool multi = B.boundary_in != nullptr;
  // Unique per pass launch and ever growing (the batch counter lives on the device, so a
  // replayed graph gets fresh stamps too).
  unsigned long long stamp = ((unsigned long long)batch_seq << 16) | (pass + 1);
  // Pass 0 + 1 in one launch (MatchBuffers::hand).
  const bool fuse = B.hand != nullptr && pass_arg == 0 && !multi;
  bool fused_stage = false;   // this wave is in its pass-1 part
  ClassState start0[W] = {};  // the state the pass-0 replay started from (as recorded: clamped)

  // Layout: the ranks of all rings first, the generation indexes behind them. ring_total
  // entries per array (flags >> 8; C << rshift <= ring_total): fewer entries = less LDS per
  // wave = more waves of this latency-bound kernel per CU.
  const uint32_t ring_total = flags >> 8;
  MatchWave<W> w{L, lds_ring, lds_ring + ring_total, (1u << rshift) - 1, rshift, lane, {}};
  const uint32_t R = 1u << rshift;
  // A ring is topped up when no more than `thresh` entries are left in it.
  // Rings of 32 entries: the fast loop watches them itself (match_fast_loop<true>) and wants more
  // than 10 entries in every ring when it starts. With 64 entries or more the caller's bound on
  // the run length is rarely what ends a call, and the watching costs more than it saves (one
  // instruction per request: cfg3 379 -> 393 us); with 16 or fewer a look every 8 picks is too late.
  constexpr uint32_t check_every = CHECKED ? 8u : 0u;  // (the host: CHECKED <=> W == 1 and rshift == 5)
  const uint32_t thresh = check_every ? 12 : (R / 4 < 4 ? 4 : (R / 4 > 12 ? 12 : R / 4));
  uint32_t steps = 1;  // DPP steps of the min over the class lanes
  while ((1u << steps) < C) ++steps;
  // Two requests per iteration of the fast loop (match_fast_loop): with few classes the two
  // winners are mostly the same lane and the pairing only costs.
  const uint32_t pair_mode = (flags & 4u) && steps >= 3 ? 1u : 0u;

  // The first block's requests do not depend on the start state: fetched now, behind the loads
  // of the level guesses (pass 0 with own guesses, one mask word per request).
  bool pre_staged = false;
  uint64_t pre_m = 0;
  uint32_t pre_slo = kNone, pre_shi = kNone;
  if (own_guess && W == 1 && T.words == 1) {
    const uint32_t c_t0 = kc * chunk_size, c_t1 = min(n_tasks, c_t0 + chunk_size);
    const uint32_t tl = (warm_mode && kc > 0 ? c_t0 - B.warm_len : c_t0) + lane;
    if (tl < c_t1) {
      pre_m = T.mask[tl];
      pre_slo = T.self_lo[tl];
      pre_shi = T.self_hi[tl];
      if (pre_shi == kSelfServant) {  // (own servant by index, bin_sort.h: its slot range)
        const uint32_t b = shared.slot_base[pre_slo], e = shared.slot_base[pre_slo + 1];
        pre_slo = e > b ? b : kNone;
        pre_shi = e > b ? e : kNone;
      }
    }
    pre_staged = true;
  }
  bool rings_from_windows = false;  // the level table's windows filled the rings already
  uint32_t win_filled = 0;          // (this lane's class: list entries [.., win_filled) are in its ring)

  // ---- start state; is there anything to do? ----
  ClassState next_guess{};  // pass 0 with own guesses: the level guess of the next chunk
  if (own_guess) {
    // Level guesses of this chunk and the next: two lower bounds in the lane's class list,
    // walked together (the loads of the two searches overlap).
    ClassState st{};
    if (lane < C) {
      const uint32_t n0 = before0, n1 = before1;
      const uint32_t b = L.cls_begin[lane], e = L.cls_begin[lane + 1];
      uint32_t c0 = 0, c1 = 0;
      if (L.list_p && B.level_tab && C <= 8) {
        // (filled in below by the whole wave, from the level table)
      } else if (L.list_p && C <= 4) {
        // (filled in below by the whole wave)
      } else if (L.list_p) {
        // Every step of a search is a memory round trip (~1 us in a busy launch) and the picks
        // cannot start before the last one: the tile table narrows both searches to one sort
        // tile's worth of the class list, and each step probes the three quartile points of
        // what is left (independent loads, one round trip) — 1 + ~4 round trips where the plain
        // bisection of a 2^17-entry list took 18.
        uint32_t lo0 = b, hi0 = e, lo1 = b, hi1 = e;
        if (B.tile_tab) {
          const uint32_t off = prm->rank_offset, nt = B.tile_tab_tiles;
          const uint32_t t0 = min((n0 > off ? n0 - off : 0u) / B.tile_tab_elems, nt - 1);
          const uint32_t t1 = min((n1 > off ? n1 - off : 0u) / B.tile_tab_elems, nt - 1);
          const uint32_t* row = B.tile_tab + (size_t)lane * nt;
          const uint32_t a0 = row[t0], z0 = t0 + 1 < nt ? row[t0 + 1] : e - b;
          const uint32_t a1 = row[t1], z1 = t1 + 1 < nt ? row[t1 + 1] : e - b;
          lo0 = b + a0;
          hi0 = b + z0;
          lo1 = b + a1;
          hi1 = b + z1;
        }
        while (lo0 < hi0 || lo1 < hi1) {
          // quartile probes of [lo, hi): q1 <= q2 <= q3 < hi (they coincide in short ranges)
          const uint32_t s0 = (hi0 - lo0 + 3) >> 2, s1 = (hi1 - lo1 + 3) >> 2;
          uint32_t q0[3], q1[3], v0[3], v1[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            q0[k] = lo0 < hi0 ? min(lo0 + (k + 1) * s0 - 1, hi0 - 1) : 0u;
            q1[k] = lo1 < hi1 ? min(lo1 + (k + 1) * s1 - 1, hi1 - 1) : 0u;
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            v0[k] = lo0 < hi0 ? list_rank(L, q0[k]) : 0u;
            v1[k] = lo1 < hi1 ? list_rank(L, q1[k]) : 0u;
          }
          if (lo0 < hi0) {
            if (v0[0] >= n0) hi0 = q0[0];
            else if (v0[1] >= n0) { lo0 = q0[0] + 1; hi0 = q0[1]; }
            else if (v0[2] >= n0) { lo0 = q0[1] + 1; hi0 = q0[2]; }
            else lo0 = q0[2] + 1;
          }
          if (lo1 < hi1) {
            if (v1[0] >= n1) hi1 = q1[0];
            else if (v1[1] >= n1) { lo1 = q1[0] + 1; hi1 = q1[1]; }
            else if (v1[2] >= n1) { lo1 = q1[1] + 1; hi1 = q1[2]; }
            else lo1 = q1[2] + 1;
          }
        }
        c0 = lo0;
        c1 = lo1;
      } else {
        c0 = b + min(n0, e - b);
        c1 = b + min(n1, e - b);
      }
      st.cursor = st.lo = c0;
      st.hown_lo = st.hown_hi = kNone;
      next_guess = st;
      next_guess.cursor = next_guess.lo = c1;
    }
    if (L.list_p && B.level_tab && C <= 8) {
      // Level table: class c's first slot with rank >= 64 m is a lookup; the slots of the class
      // with a rank in [64 m, level) — at most 64 — are counted by the whole wave in one
      // 64-entry window of the list. Both levels of all classes side by side: two dependent
      // round trips instead of one per search step.
      uint32_t pos[2] = {0, 0}, lvl[2] = {before0, before1}, endc = 0;
      if (lane < C) {
        endc = L.cls_begin[lane + 1];
#pragma unroll
        for (int h = 0; h < 2; ++h)
          pos[h] = lvl[h] < n_slots_all ? B.level_tab[(size_t)(lvl[h] >> 6) * C + lane] : endc;
      }
      // <= 4 classes: the window of this chunk's own level is read with the slots beside the
      // ranks and two more windows behind it — the 192 list entries the rings start with, so the
      // rings need no round trip of their own (init_rings).
      const bool fill = C <= 4 && L.stride == 2 && (C << rshift) <= ring_total && (1u << rshift) >= 256;
      uint32_t v[16], fr[8], fg[8], g0[4];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        fr[q] = 0u;  // (ring sentinels: ~rank 0 = no slot)
        fg[q] = kNone;
        if (q < 4) g0[q] = kNone;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const uint32_t c = (uint32_t)q >> 1;
        v[q] = 0xFFFFFFFFu;  // "not below the level"
        if (c < C) {
          const uint32_t at = readlane_u32(pos[q & 1], c) + lane, e = readlane_u32(endc, c);
          if (at < e) {
            v[q] = list_rank(L, at);
            if (fill && !(q & 1)) g0[c] = list_slot(L, at);
          }
          if (fill && !(q & 1)) {
#pragma unroll
            for (int part = 1; part < 3; ++part) {
              const uint32_t a2 = at + 64 * part;
              if (a2 < e) {
                fr[2 * c + part - 1] = ~list_rank(L, a2);
                fg[2 * c + part - 1] = list_slot(L, a2);
              }
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const uint32_t c = (uint32_t)q >> 1;
        if (c < C) {
          const uint32_t tgt = readlane_u32(lvl[q & 1], c);
          const uint32_t cnt = (uint32_t)__popcll(__ballot(v[q] < tgt));  // a prefix: the list is sorted
          if (lane == c) {
            const uint32_t cur = pos[q & 1] + cnt;
            if (q & 1) next_guess.cursor = next_guess.lo = cur;
            else st.cursor = st.lo = cur;
          }
        }
      }
      if (lane < C) next_guess.hown_lo = next_guess.hown_hi = kNone;
      if (fill) {
        // (ring positions are list positions modulo the ring size: entries below the cursor land
        // in places the ring does not look at)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if ((uint32_t)c < C) {
            const uint32_t p0 = readlane_u32(pos[0], c) + lane;
            w.ring_p[w.at(c, p0)] = v[2 * c] == 0xFFFFFFFFu ? 0u : ~v[2 * c];
            w.ring_g[w.at(c, p0)] = g0[c];
#pragma unroll
            for (int part = 1; part < 3; ++part) {
              w.ring_p[w.at(c, p0 + 64 * part)] = fr[2 * c + part - 1];
              w.ring_g[w.at(c, p0 + 64 * part)] = fg[2 * c + part - 1];
            }
          }
        }
        rings_from_windows = true;
        win_filled = pos[0] + 192;
      }
    } else if (L.list_p && C <= 4) {
      // A handful of classes: the wave searches together, 64 probes per search and round
      // (three rounds for a list of 2^18 entries instead of eighteen dependent loads), the
      // 2 * C searches side by side (targets: the class's own level, see before0 / before1).
      uint32_t lo[8], hi[8], tgt[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        tgt[q] = readlane_u32((q & 1) ? before1 : before0, min((uint32_t)q >> 1, C - 1));
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t c = (uint32_t)q >> 1;
        lo[q] = hi[q] = 0;
        if (c < C) {
          lo[q] = L.cls_begin[c];
          hi[q] = L.cls_begin[c + 1];
        }
      }
      for (;;) {
        bool any = false;
        uint32_t v[8], step[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          step[q] = (hi[q] - lo[q] + 63) / 64;
          v[q] = 0xFFFFFFFFu;  // "not less": an empty segment
          if (hi[q] > lo[q]) {
            any = true;
            const uint32_t first = lo[q] + lane * step[q];
            if (first < hi[q]) v[q] = list_rank(L, min(first + step[q] - 1, hi[q] - 1));
          }
        }
        if (!any) break;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (hi[q] > lo[q]) {
            const uint32_t target = tgt[q];
            // Segments wholly below the target form a prefix (the list is sorted).
            const uint32_t k = (uint32_t)__popcll(__ballot(v[q] < target));
            const uint32_t nlo = lo[q] + k * step[q];
            if (nlo >= hi[q] || step[q] == 1) {
              lo[q] = hi[q] = min(nlo, hi[q]);  // found
            } else {
              lo[q] = nlo;  // the answer is in this segment, whose last entry is >= target
              hi[q] = min(hi[q], nlo + step[q]);
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (lane == ((uint32_t)q >> 1) && lane < C) {
          if (q & 1) next_guess.cursor = next_guess.lo = lo[q];
          else st.cursor = st.lo = next_guess.cursor = next_guess.lo = lo[q];
        }
      }
    }
    if (B.zone_box && kc > 0) {
      // The header is there long before any wave gets here (the walk starts before this launch and
      // finds its stretch in 5 us; the level guesses above took 15) — unless the walk's kernel has
      // not been given a place on the chip: then nobody waits for it.
      uint32_t z_lo = 0, z_hi = 0;
      for (uint32_t tries = 0; tries < kZoneHeaderTries; ++tries) {
        const unsigned long long h0 = __hip_atomic_load(B.zone_box + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long h1 = __hip_atomic_load(B.zone_box + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(h0 >> 32) == batch_seq && (uint32_t)(h1 >> 32) == batch_seq) {
          z_lo = (uint32_t)h0;
          z_hi = (uint32_t)h1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      if (kc > z_lo && kc < z_hi) {  // (row 0 is the level guess of chunk z_lo itself)
        const unsigned long long* g = B.zone_box + 2 + (size_t)(kc - z_lo) * C + lane;
        for (uint32_t tries = 0; tries < kZoneRowTries; ++tries) {
          unsigned long long v = (unsigned long long)batch_seq << 32;
          if (lane < C) v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__ballot((uint32_t)(v >> 32) != batch_seq) == 0) {
            if (lane < C) st.cursor = st.lo = (uint32_t)v;
            rings_from_windows = false;  // (filled around the level guess)
            __builtin_amdgcn_s_setprio(2);  // (the launch ends with these chunks)
            break;
          }
          __builtin_amdgcn_s_sleep(32);
        }
      }
    }
    w.set_state(0, st, lane, C);
    if (rings_from_windows && lane < C) w.k[0].filled = win_filled;
  } else {
    const ClassState* start;
    if (pass == 0) {
      start = B.guess0 + (size_t)kc * C;
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
      if (walk_scout) {
        if (lane == 0) {
          atomicMin(&prm->reserved0, kc);
          B.flags[pass & B.flag_mask] = 1;  // (not final)
        }
        return;
      }
      if (walk_run && kc != walk_frontier) return;
      if (walk_run) YDC_PROBE(0, 5);
      if (!walk_run && pass >= 2 && kc >= 1 && B.sampled[(pass - 1) & B.flag_mask] < 4) {
        // Few end states changed in the previous pass (sampled estimate < ~64): what is left
        // are chains — chunks whose predecessor's end state keeps changing. A chunk whose
        // predecessor is itself inconsistent would replay from a stale state; it leaves the
        // work to the wave that follows the chain from its head (below), and says that the
        // batch is not final yet. (With many changes everybody replays in parallel instead:
        // most of them settle within their own chunk.)
        const ClassState* pstart = kc == 1 ? B.boundary_in : B.endst + (size_t)(kc - 2) * C;
        bool pdiff = false;
        if (pstart) {
#pragma unroll
          for (int j = 0; j < W; ++j) {
            const uint32_t c = lane + 64 * j;
            if (c < C) {
              const ClassState used = B.checkpoint[(size_t)(((kc - 1) * chunk_size) >> 6) * C + c];
              pdiff |= !class_state_equal(used, pstart[c]);
            }
          }
        }
        if (__ballot(pdiff) != 0) {
          if (lane == 0) B.flags[pass & B.flag_mask] = 1;
          return;
        }
      }
      // Inconsistent: this pass has work. One wave per chunk and pass.
      uint32_t taken = 0;
      if (lane == 0) taken = atomicMax(&B.claim[kc], stamp) == stamp;
      if (readlane_u32(taken, 0) && !walk_run) return;  // a wave following its chain got here first
    }
  }

  if (pass_arg == 0) YDC_PROBE(probe_kc, 1);  // start state known (level guesses done)
  bool warm = warm_mode && kc > 0;  // this replay starts kWarmUp requests before the chunk
  bool ring_ready = false;
  uint64_t holes[W] = {};
  uint32_t followed = 0;  // chunks this wave has carried on into

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
  // Up to four rings in one memory round trip: all their loads are in flight before the first
  // LDS store (one refill at a time exposes a round trip per class, and classes that started
  // with equally full rings run low together).
  auto refill_group = [&](uint64_t& need, int bj) {
    uint32_t g_cl[4], g_from[4], g_to[4], g_end[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      g_cl[u] = g_from[u] = g_to[u] = g_end[u] = 0;
      if (need) {
        const uint32_t cc = (uint32_t)__builtin_ctzll(need);
        need &= need - 1;
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
        g_cl[u] = readlane_u32(f_cl, cc);
        g_from[u] = readlane_u32(f_from, cc);
        g_to[u] = readlane_u32(f_to, cc);
        g_end[u] = readlane_u32(f_end, cc);
      }
    }
    uint32_t tp[4], tg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t e = g_from[u] + lane;
      const bool real = e < g_to[u] && e < g_end[u];
      tp[u] = real ? ~list_rank(L, e) : 0u;
      tg[u] = real ? list_slot(L, e) : kNone;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t e = g_from[u] + lane;
      if (e < g_to[u]) {
        w.ring_p[w.at(g_cl[u], e)] = tp[u];
        w.ring_g[w.at(g_cl[u], e)] = tg[u];
      }
    }
  };
  // Refills every ring that is running low — and, while a round trip is being paid for anyway,
  // every ring that is half empty; returns how many picks are safe before the next look (every
  // ring keeps the entry after next).
  auto top_up = [&]() -> uint32_t {
    uint32_t least = 0x7FFFFFFFu;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const bool real = lane + 64 * j < C;
      uint64_t need = __ballot(real && w.ring_left(j) <= thresh);
      if (need) need = __ballot(real && w.ring_left(j) <= max(thresh, R / 2));
      while (need) refill_group(need, j);
      least = min(least, real ? w.ring_left(j) : 0x7FFFFFFFu);
    }
    __builtin_amdgcn_wave_barrier();
    return wave_min_u32(least) - 2;
  };

  for (;;) {  // chunk kc, then the chunks after it while the end states keep changing
    const uint32_t t0 = kc * chunk_size;
    const uint32_t t1 = min(n_tasks, t0 + chunk_size);
    bool stopped_early = false;

    // Is block tb of the chunk that starts at c0 one with a checkpoint in front?
    auto is_cp = [&](uint32_t tb, uint32_t c0) { return (((tb - c0) >> 6) & (B.cp_every - 1)) == 0; };
    // Requests of a block are staged one block ahead: lane l holds request tb + l.
    uint64_t nx_m[W];
    uint32_t nx_slo, nx_shi;
    ClassState nx_cp[W];
    auto stage = [&](uint32_t tb) {
      const uint32_t tl = tb + lane;
      nx_slo = nx_shi = kNone;
      if (pre_staged) {  // (the chunk's first block, fetched before the level guesses)
        pre_staged = false;
        nx_m[0] = pre_m;
        nx_slo = pre_slo;
        nx_shi = pre_shi;
      } else {
#pragma unroll
      for (int j = 0; j < W; ++j) {
        nx_m[j] = (tl < t1 && (uint32_t)j < T.words) ? T.mask[(size_t)tl * T.words + j] : 0;
        const uint32_t c = lane + 64 * j;
        if (pass != 0 && tb < t1 && c < C && is_cp(tb, t0)) nx_cp[j] = B.checkpoint[(size_t)(tb >> 6) * C + c];
      }
      if (tl < t1) {
        nx_slo = T.self_lo[tl];
        nx_shi = T.self_hi[tl];
      }
      }
    };
    // The second step of the staging, one block's work later (looking at what stage() fetched
    // any earlier would expose a memory round trip in every block): a request's own servant by
    // index (bin_sort.h) becomes its slot range, if it offers any.
    auto resolve_staged = [&]() {
      if (nx_shi == kSelfServant) {
        const uint32_t b = shared.slot_base[nx_slo], e = shared.slot_base[nx_slo + 1];
        nx_slo = e > b ? b : kNone;
        nx_shi = e > b ? e : kNone;
      }
    };
    stage(warm ? t0 - B.warm_len : t0);
    resolve_staged();
    ClassState early_cp{};
    if (W == 1 && pass != 0 && lane < C) early_cp = B.early[(size_t)kc * C + lane];

    for (uint32_t tb = warm ? t0 - B.warm_len : t0; tb < t1; tb = tb < t0 ? t0 : tb + 64) {
      const bool warm_blk = tb < t0;  // the warm-up requests: picks are made and thrown away
      // ---- checkpoint: stop if the previous replay was in the same state here ----
      if (!warm_blk && is_cp(tb, t0)) {
        ClassState* cp = B.checkpoint + (size_t)(tb >> 6) * C;
        bool differs = false;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const uint32_t c = lane + 64 * j;
          if (c < C) {
            const ClassState s = w.state(j);
            if (pass == 0) {
              cp[c] = s;
              if (tb == t0) start0[j] = s;  // what this chunk's replay starts from
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
      stage(warm_blk ? t0 : tb + 64);

      if (!ring_ready) {
        // First block that really runs: fill the rings (unless the level table's windows did).
        if (!rings_from_windows) w.init_rings(C, init_fill, W);
        rings_from_windows = false;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < W; ++j) {
          if (lane + 64 * j < C) w.load_heads(j);
          holes[j] = __ballot(w.k[j].lo < w.k[j].cursor);
        }
        ring_ready = true;
        if (pass_arg == 0 && !fused_stage) YDC_PROBE(probe_kc, 2);  // requests staged, rings filled
      }

      const uint64_t has_self = __ballot(slo != kNone);
      const uint64_t dyn_self = __ballot(shi == kSelfShared);
      uint32_t res = kIdxTimeout;
      uint32_t keep = 64;  // results of this block to store (fewer after an early stop)
      const uint32_t cnt = warm_blk ? B.warm_len : min(64u, t1 - tb);

      // General step for request i: holes, own-servant heads, last-resort self pick. The
      // shared state machine (dispatch_core.h) advances the class; its ring is topped up
      // at the new cursor.
      auto general_step = [&](uint32_t i) {
        uint64_t mw[W];
#pragma unroll
        for (int j = 0; j < W; ++j)
          mw[j] = ((uint64_t)readlane_u32(mhi[j], i) << 32) | readlane_u32(mlo[j], i);
        uint32_t self_lo = readlane_u32(slo, i);
        uint32_t self_hi = readlane_u32(shi, i);
        if (self_hi == kSelfShared) {
          // Several servants on the requestor's host: `self` = the first of them that is
          // eligible and still free in the CURRENT state (dispatch_core.h). Wave-uniform.
          auto state_of = [&](uint32_t c, uint32_t& cursor, uint32_t& lo, uint32_t& hown_lo) {
            const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(c & 63u));
            cursor = lo = hown_lo = 0;
#pragma unroll
            for (int j = 0; j < W; ++j) {
              const uint32_t cur_j = readlane_u32(w.k[j].cursor, l), lo_j = readlane_u32(w.k[j].lo, l),
                             ho_j = readlane_u32(w.k[j].hown_lo, l);
              if ((uint32_t)j == (c >> 6)) {
                cursor = cur_j;
                lo = lo_j;
                hown_lo = ho_j;
              }
            }
          };
          resolve_shared_self(mw, self_lo, &shared, state_of, self_lo, self_hi);
        }
        // A head that is the requestor's own slot is walked over (class_candidate) — an entry per
        // memory round trip. Registries of a few servants with tens of thousands of slots each have
        // runs of thousands of them (a request took 830 us there): where the entry behind the head
        // is the requestor's too, the wave looks for the end of the run together, 64 entries at a time.
        uint32_t skip[W];
#pragma unroll
        for (int j = 0; j < W; ++j) skip[j] = kNone;
        if (self_lo != kNone) {
#pragma unroll
          for (int j = 0; j < W; ++j) {
            const LaneClass& q = w.k[j];
            bool run2 = false;
            if (((mw[j] >> lane) & 1u) && q.cursor + 1 < q.end && !q.single &&
                !(q.lo < q.cursor && q.hown_lo != self_lo)) {
              const uint32_t g0 = w.ring_g[w.at(lane + 64 * j, q.cursor)];
              const uint32_t g1 = w.ring_g[w.at(lane + 64 * j, q.cursor + 1)];
              run2 = g0 >= self_lo && g0 < self_hi && g1 >= self_lo && g1 < self_hi;
            }
            uint64_t todo = __ballot(run2);
            while (todo) {
              const uint32_t cc = (uint32_t)__builtin_ctzll(todo);
              todo &= todo - 1;
              const uint32_t end_c = readlane_u32(q.end, cc);
              uint32_t pos = readlane_u32(q.cursor, cc) + 2, found = end_c;
              while (pos < end_c) {
                const uint32_t e = pos + lane;
                const uint32_t g = e < end_c ? list_slot(L, e) : kNone;
                const uint64_t other = __ballot(e >= end_c || g < self_lo || g >= self_hi);
                if (other) {
                  found = min(end_c, pos + (uint32_t)__builtin_ctzll(other));
                  break;
                }
                pos += 64;
              }
              if (lane == cc) skip[j] = found;
            }
          }
        }
        uint32_t bp = kNone, bi = 0;
        int bj = 0;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          if ((mw[j] >> lane) & 1u) {
            uint32_t ci, cp, cg;
            if (class_candidate(L, w.as_run(j), self_lo, self_hi, ci, cp, cg, skip[j]) && cp < bp) {
              bp = cp;
              bi = ci;
              bj = j;
            }
          }
        }
        const uint32_t mn = wave_min_u32(bp);
        uint64_t winners;
        uint32_t self_rank = kNone;
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
                  bj = j;
                  self_rank = list_rank(L, ci);
                }
              }
            }
          }
          winners = __ballot(ok);
        }
        if (winners == 0) return;  // Timeout (res default) — or no class at all (fixed below)
        const uint32_t win = (uint32_t)__builtin_ctzll(winners);
        // The result of a request is the global rank of its slot.
        const uint32_t taken = mn != kNone ? mn : readlane_u32(self_rank, win);
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
        // (The rings start at LDS address 0 — the kernel has no static LDS — so every ring is
        // aligned to its size, which the address stepping of the loop relies on.)
        if ((uint32_t)(uintptr_t)lds_ring & ((ring_total << 2) - 1)) __builtin_trap();
        const uint32_t base = (uint32_t)(uintptr_t)lds_ring + ((lane << rshift) << 2);
        const uint32_t rmask4 = (R << 2) - 1;
        const uint64_t my_mask = ((uint64_t)mhi[0] << 32) | mlo[0];
        const BlockMasks bm = block_masks(mlo[0], mhi[0], steps > 5);
        uint32_t raw = 0;  // ~rank of the slot of every request the fast loop served
        uint32_t i = 0;
        // The first block of a chunk pauses after kEarlyAt requests for the early checkpoint.
        uint32_t lim = tb == t0 && cnt > kEarlyAt ? kEarlyAt : cnt;
        for (;;) {
        while (i < lim) {
          uint32_t budget;
          YDC_PROBE_ACC(probe_topup, budget = top_up());
          YDC_PROBE_COUNT(probe_topup);
          const uint32_t n = check_every ? lim - i : min(lim - i, budget);
          uint32_t left = lane < C ? w.ring_left(0) : 0x7FFFFFFFu;
          const uint32_t an0 = base + (((q.cursor + 1) & w.rmask) << 2);  // address of `next`
          uint32_t an = an0;
          // Requests that need a look before the plain step: an eligible class has holes, or
          // the requestor's host runs several servants (`self` is resolved in the general step).
          const uint64_t hole_hit = (holes[0] ? __ballot((my_mask & holes[0]) != 0) : 0ull) | dyn_self;
          uint32_t st;
          const uint32_t pm = (uint32_t)__builtin_amdgcn_readfirstlane((int)pair_mode);
          YDC_PROBE_ACC(probe_loop, st = match_fast_loop<CHECKED>(i, i + n, bm, slo, shi, has_self | hole_hit,
                                                                  hole_hit, has_self, raw, q.hq, q.nq, an,
                                                                  ring_total << 2, rmask4, steps, pm, left));
          // Picks of this lane's class in the call (fewer than the ring holds): how far `next` moved.
          q.cursor += ((an - an0) & rmask4) >> 2;
          // Classes without holes were advanced with lo == cursor.
          if (!((holes[0] >> lane) & 1)) q.lo = q.cursor;
          if (st == 1) {
            YDC_PROBE_ACC(probe_gen, general_step(i));
            YDC_PROBE_COUNT(probe_loop);
            ++i;
          }
        }
        if (lim == cnt) break;
        {
          const ClassState sn = w.state(0);
          bool differs = false;
          if (lane < C) {
            if (pass == 0) {
              B.early[(size_t)kc * C + lane] = sn;
            } else if (!class_state_equal(early_cp, sn)) {
              differs = true;
              B.early[(size_t)kc * C + lane] = sn;
            }
          }
          if (pass != 0 && __ballot(differs) == 0) {
            stopped_early = true;  // the rest of the previous replay stands
            keep = kEarlyAt;
            break;
          }
        }
        lim = cnt;
        }
        if (raw != 0) res = ~raw;
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
          if ((dyn_self >> i) & 1) general = true;
          if (self && !general) {
            const uint32_t self_lo = readlane_u32(slo, i);
            const uint32_t self_len = readlane_u32(shi, i) - self_lo;
            bool own = false;  // an eligible class shows a slot of the requestor's own servant
#pragma unroll
            for (int j = 0; j < W; ++j)
              own |= ((mw[j] >> lane) & 1u) && (w.head_g(j) - self_lo < self_len);
            general = __ballot(own) != 0;
          }
          if (!general) {
            uint32_t bp = select_by_lane_mask(mw[0], w.k[0].hq);
            int bj = 0;
#pragma unroll
            for (int j = 1; j < W; ++j) {
              const uint32_t c = select_by_lane_mask(mw[j], w.k[j].hq);
              if (c > bp) {
                bp = c;
                bj = j;
              }
            }
            const uint32_t mx = wave_max_u32(bp);  // ~rank: the largest is the best slot
            if (mx != 0) {
              const uint32_t win = (uint32_t)__builtin_ctzll(__ballot(bp == mx));
              res = lane == i ? ~mx : res;
              if (lane == win) {
#pragma unroll
                for (int j = 0; j < W; ++j) {
                  if (j == bj) {
                    LaneClass& q = w.k[j];
                    const uint32_t cur = q.cursor + 1;
                    q.cursor = cur;
                    q.lo = cur;
                    q.hq = q.nq;
                    q.nq = w.ring_p[w.at(lane + 64 * j, cur + 1)];
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
      resolve_staged();  // (the next block's requests: fetched a block ago)
      if (pass_arg == 0 && !fused_stage) YDC_PROBE(probe_kc, warm_blk ? 3 : 4);  // warm-up / block done
      // No eligible class at all: EnvironmentNotFound (task_dispatcher.cc:105-108).
      if (many == 0) res = kIdxEnvNotFound;
      if (!warm_blk && tl < t1 && lane < keep) B.slot_of[tl] = res;
      if (stopped_early) break;
    }
    if (count_sims && lane == 0) atomicAdd(&prm->chunk_sims, 1u);
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
          // Is the next chunk's level guess what this chunk really ends in?
          if (kc + 1 < n_chunks)
            differs |= !class_state_equal(own_guess ? next_guess : B.guess0[(size_t)(kc + 1) * C + c], s);
        } else {
          const ClassState old = *e;
          if (!class_state_equal(old, s)) {
            differs = true;
            *e = s;
          }
        }
      }
    }
    bool changed = __ballot(differs) != 0;
    // With warm-ups the successor does not start from its level guess: pass 0 certifies nothing.
    if (pass == 0 && warm_mode && kc + 1 < n_chunks) changed = true;
    // The first chunk of the next rank starts from a guess of its own in pass 0.
    if (pass == 0 && kc + 1 == n_chunks && B.has_successor) changed = true;
    // "Not final yet": everybody stores the same 1 (no same-address atomics).
    if (changed && lane == 0) {
      B.flags[pass & B.flag_mask] = 1;
      if ((kc & 15u) == 0) atomicAdd(&B.sampled[pass & B.flag_mask], 1u);
    }
    if (pass_arg == 0 && !fused_stage) YDC_PROBE(probe_kc, 5);  // results + end state stored
    if (pass_arg == 0 && !fused_stage) {
      YDC_PROBE_PUT(probe_kc, 10, probe_topup);
      YDC_PROBE_PUT(probe_kc, 11, probe_loop);
    }
    if (pass_arg == 0 && fused_stage) YDC_PROBE(probe_kc, 9);   // second replay done
    if (walk_run) {  // (tools/cliff_probe.py: running totals of the walking wave, row 0)
      YDC_PROBE_PUT(0, 6, probe_topup);
      YDC_PROBE_PUT(0, 7, probe_loop);
      YDC_PROBE_PUT(0, 8, probe_gen);
      YDC_PROBE(0, 9);
    }
    if (pass == 0) {
      if (!fuse) return;
      // ---- pass 1 of this chunk, in the same launch. The end state of the pass-0 replay goes
      // out in 8-byte granules {word, batch stamp} (agent-scope atomics on both sides: every
      // granule carries its own validity, no fence, no ordering between granules needed) ...
      const uint32_t hstamp = batch_seq;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const uint32_t c = lane + 64 * j;
        if (c < C) {
          const ClassState s = w.state(j);
          // (word-major inside the chunk's block: the C lanes of one store write C consecutive
          // granules — class-major, every lane's 8 bytes went to a 32-byte sector of their own)
          unsigned long long* g = B.hand + (size_t)kc * C * 4 + c;
          const unsigned long long tag = (unsigned long long)hstamp << 32;
          __hip_atomic_store(g + 0 * (size_t)C, tag | s.cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(g + 1 * (size_t)C, tag | s.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(g + 2 * (size_t)C, tag | s.hown_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(g + 3 * (size_t)C, tag | s.hown_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      YDC_PROBE(probe_kc, 6);  // hand-off published
      if (kc == 0) return;  // chunk 0 started from the true state
      // ... and the predecessor's comes in the same way. Bounded: a wave that does not get it
      // (HIP promises nothing about dispatch order) says "not final" and leaves its chunk to
      // the next launch, which checks it like any other pass.
      ClassState pred[W];
      for (uint32_t tries = 0;; ++tries) {
        bool missing = false;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const uint32_t c = lane + 64 * j;
          pred[j] = ClassState{};
          if (c < C) {
            unsigned long long* g = B.hand + (size_t)(kc - 1) * C * 4 + c;
            const unsigned long long v0 = __hip_atomic_load(g + 0 * (size_t)C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long v1 = __hip_atomic_load(g + 1 * (size_t)C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long v2 = __hip_atomic_load(g + 2 * (size_t)C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long v3 = __hip_atomic_load(g + 3 * (size_t)C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            missing |= (uint32_t)(v0 >> 32) != hstamp || (uint32_t)(v1 >> 32) != hstamp ||
                       (uint32_t)(v2 >> 32) != hstamp || (uint32_t)(v3 >> 32) != hstamp;
            pred[j].cursor = (uint32_t)v0;
            pred[j].lo = (uint32_t)v1;
            pred[j].hown_lo = (uint32_t)v2;
            pred[j].hown_hi = (uint32_t)v3;
          }
        }
        if (__ballot(missing) == 0) break;
        if (tries >= B.hand_tries) {
          // Not verified: pass 1 is "not final", and counts as a changed end state for the
          // next pass's choice between parallel replays and chain following (sampled like one).
          if (lane == 0) {
            B.flags[1u & B.flag_mask] = 1;
            if ((kc & 15u) == 0) atomicAdd(&B.sampled[1u & B.flag_mask], 1u);
          }
          return;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      YDC_PROBE(probe_kc, 7);  // predecessor's granules arrived
      bool off = false;  // did this chunk's level guess miss the predecessor's end state?
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (lane + 64 * j < C) off |= !class_state_equal(start0[j], pred[j]);
      if (__ballot(off) == 0) return;  // consistent
      YDC_PROBE(probe_kc, 8);  // second replay starts
      pass = 1;
      stamp = ((unsigned long long)batch_seq << 16) | (pass + 1);
      fused_stage = true;
#pragma unroll
      for (int j = 0; j < W; ++j) w.set_state(j, pred[j], lane + 64 * j, C);
      ring_ready = false;
      warm = false;
      continue;  // replay chunk kc from the predecessor's end state, with pass 1's early stops
    }
    if (walk_run) {  // the walk: on to the next chunk, whatever this one's end state did
      if (kc + 1 >= n_chunks) return;
      ++kc;
      if (lane == 0) (void)atomicMax(&B.claim[kc], stamp);
      continue;
    }
    if (!changed) return;  // nothing downstream is affected
    // (Within the launch of pass 0 every chunk belongs to its own wave: no following.)
    if (fused_stage) return;
    // The next chunk is now inconsistent. Follow the chain unless its own wave (or
    // another follower) has it in this pass; then the next pass picks it up.
    if (kc + 1 >= n_chunks) return;
    if (++followed > 64) return;  // bound one wave's serial work; the next pass goes on
    ++kc;
    uint32_t taken = 0;
    if (lane == 0) taken = atomicMax(&B.claim[kc], stamp) == stamp;
    if (readlane_u32(taken, 0)) return;
  }
}

}  // namespace ydc
#endif  // YADCC_AMD_MATCH_KERNEL_H_

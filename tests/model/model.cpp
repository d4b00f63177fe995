// tests/model/model.cpp — TEST TOOL, not product code.
//
// Replays the device pipeline of yadcc_amd/csrc on the CPU, single-threaded,
// using the very same placement code (dispatch_core.h, host_tables.h): slot
// generation, stable key sort, class lists, task classification, chunked
// speculative matching with the same guess/update rule as the kernels, and
// finalisation. Lets pytest check the shared algorithm code (and how many
// speculation rounds it needs) against the oracle without a GPU. The product
// library never links this file.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../yadcc_amd/csrc/dispatch_core.h"
#include "../../yadcc_amd/csrc/host_tables.h"

using namespace ydc;

extern "C" {

struct model_stats {
  uint32_t n_slots, n_classes, key_bits, n_chunks, rounds, chunk_sims, force_fp64;
};

// Returns 0, or -5 if there are more than 64 classes.
int model_dispatch(uint32_t S, const uint32_t* version, const uint32_t* nproc, const uint32_t* load,
                   const uint32_t* max_tasks, const uint32_t* running, const uint32_t* flags,
                   const uint64_t* env_mask, const uint32_t* ip_id, uint32_t N,
                   const uint32_t* env_id, const uint32_t* min_version,
                   const uint32_t* requestor_ip, uint32_t chunk_size, int force_fp64,
                   uint32_t* out_idx, double* out_util, uint32_t* out_running,
                   model_stats* stats) {
  HostTables T;
  T.build(S, env_mask, version, max_tasks, nproc, ip_id);
  const uint32_t C = T.n_classes();
  KeyFormat kf = choose_key_format(force_fp64 ? 32 : T.cap_bits);

  // --- servant scan: slot counts and bases.
  std::vector<uint32_t> base(S + 1, 0);
  for (uint32_t s = 0; s < S; ++s) {
    uint32_t k = T.class_of[s] == kNone
                     ? 0
                     : servant_slot_count(nproc[s], load[s], max_tasks[s], running[s], flags[s]);
    base[s + 1] = base[s] + k;
  }
  const uint32_t M = base[S];

  // --- slot generation + stable sort by key (generation order breaks ties).
  std::vector<uint64_t> key(M);
  for (uint32_t s = 0; s < S; ++s) {
    for (uint32_t g = base[s]; g < base[s + 1]; ++g) {
      uint32_t r = running[s] + (g - base[s]);
      uint32_t cap = slot_capacity(nproc[s], load[s], max_tasks[s], r);
      uint32_t tier = slot_tier(nproc[s], flags[s], r);
      key[g] = kf.exact ? slot_key_exact(tier, r, cap, kf.cap_bits) : slot_key_fp64(tier, r, cap);
    }
  }
  std::vector<uint32_t> sorted_g(M);
  std::iota(sorted_g.begin(), sorted_g.end(), 0u);
  std::stable_sort(sorted_g.begin(), sorted_g.end(),
                   [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });

  // --- class lists (stable partition of ranks by class).
  std::vector<uint32_t> cls_begin(C + 1, 0), list_p(M), list_g(M);
  {
    std::vector<uint32_t> cnt(C + 1, 0);
    std::vector<uint32_t> owner(M);
    for (uint32_t p = 0; p < M; ++p) {
      owner[p] = owner_of_slot(base.data(), S, sorted_g[p]);
      cnt[T.class_of[owner[p]] + 1]++;
    }
    for (uint32_t c = 0; c < C; ++c) cls_begin[c + 1] = cls_begin[c] + cnt[c + 1];
    std::vector<uint32_t> fill(cls_begin.begin(), cls_begin.end() - 1);
    for (uint32_t p = 0; p < M; ++p) {
      uint32_t c = T.class_of[owner[p]];
      list_p[fill[c]] = p;
      list_g[fill[c]] = sorted_g[p];
      fill[c]++;
    }
  }
  ClassLists L;
  L.list_p = C > 1 ? list_p.data() : nullptr;
  L.list_g = list_g.data();
  L.cls_begin = cls_begin.data();
  L.n_classes = C;

  // --- task classification.
  const uint32_t W = std::max<uint32_t>(1, (C + 63) / 64);
  std::vector<uint64_t> tmask((size_t)N * W);
  std::vector<uint32_t> tself_lo(N, kNone), tself_hi(N, kNone);
  bool need_shared = false;
  for (uint32_t t = 0; t < N; ++t) {
    task_class_mask(env_id[t], min_version[t], T.cls_env.data(), T.cls_ver.data(), C, W,
                    &tmask[(size_t)t * W]);
    uint32_t i = lower_bound_u32(T.ip_sorted.data(), S, requestor_ip[t]);
    if (i < S && T.ip_sorted[i] == requestor_ip[t]) {
      if (i + 1 < S && T.ip_sorted[i + 1] == requestor_ip[t]) {
        tself_lo[t] = i;
        tself_hi[t] = kSelfShared;
        need_shared = true;
      } else {
        uint32_t s = T.ip_servant[i];
        if (base[s + 1] > base[s]) {
          tself_lo[t] = base[s];
          tself_hi[t] = base[s + 1];
        }
      }
    }
  }
  TaskTable ti{tmask.data(), tself_lo.data(), tself_hi.data(), W};

  // --- matching.
  std::vector<uint32_t> slot_of(N);
  if (chunk_size == 0 || need_shared) chunk_size = N ? N : 1;
  const uint32_t K = N ? (N + chunk_size - 1) / chunk_size : 0;
  std::vector<ClassRun> runs(std::max<uint32_t>(C, 1));
  std::vector<uint32_t> left;
  SharedIpTable sh{T.ip_sorted.data(), T.ip_servant.data(), S, T.class_of.data(),
                   base.data(),        S,                   nullptr};
  if (need_shared) {
    left.resize(S);
    for (uint32_t s = 0; s < S; ++s) left[s] = base[s + 1] - base[s];
    sh.left = left.data();
  }
  uint32_t rounds = 0, sims = 0;
  if (K) {
    // consuming tasks before each chunk
    std::vector<uint32_t> before(K + 1, 0);
    for (uint32_t k = 0; k < K; ++k) {
      uint32_t n = 0;
      for (uint32_t t = k * chunk_size; t < std::min(N, (k + 1) * chunk_size); ++t)
        n += !task_mask_empty(ti, t);
      before[k + 1] = before[k] + n;
    }
    std::vector<ClassState> guess((size_t)K * C), endst((size_t)K * C);
    for (uint32_t k = 0; k < K; ++k)
      for (uint32_t c = 0; c < C; ++c) guess[(size_t)k * C + c] = level_guess(L, c, before[k]);
    std::vector<uint8_t> dirty(K, 1);
    for (;;) {
      ++rounds;
      for (uint32_t k = 0; k < K; ++k) {
        if (!dirty[k]) continue;
        ++sims;
        sim_chunk(L, ti, k * chunk_size, std::min(N, (k + 1) * chunk_size),
                  &guess[(size_t)k * C], &endst[(size_t)k * C], slot_of.data(), runs.data(),
                  need_shared ? &sh : nullptr);
        dirty[k] = 0;
      }
      // update (same rule as k_update): the start guess of chunk k+1 becomes the
      // end state chunk k reached in its latest simulation.
      bool any = false;
      for (uint32_t k = 0; k + 1 < K; ++k) {
        for (uint32_t c = 0; c < C; ++c) {
          const ClassState& en = endst[(size_t)k * C + c];
          ClassState& g = guess[(size_t)(k + 1) * C + c];
          if (!class_state_equal(g, en)) {
            g = en;
            dirty[k + 1] = 1;
            any = true;
          }
        }
      }
      if (!any) break;
      if (rounds > K + 2) return -6;
    }
  }

  // --- finalise.
  std::vector<uint32_t> run(running, running + S);
  for (uint32_t t = 0; t < N; ++t) {
    uint32_t g = slot_of[t];
    if (g >= kIdxEnvNotFound) {
      out_idx[t] = g;
      if (out_util) out_util[t] = -1.0;
      continue;
    }
    uint32_t s = owner_of_slot(base.data(), S, g);
    uint32_t r = running[s] + (g - base[s]);
    out_idx[t] = s;
    if (out_util) out_util[t] = slot_utilization(r, slot_capacity(nproc[s], load[s], max_tasks[s], r));
    run[s]++;
  }
  if (out_running) std::memcpy(out_running, run.data(), S * sizeof(uint32_t));
  if (stats) {
    stats->n_slots = M;
    stats->n_classes = C;
    stats->key_bits = kf.key_bits;
    stats->n_chunks = K;
    stats->rounds = rounds;
    stats->chunk_sims = sims;
    stats->force_fp64 = !kf.exact;
  }
  return 0;
}

}  // extern "C"

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
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../yadcc_amd/csrc/dispatch_core.h"
#include "../../yadcc_amd/csrc/host_tables.h"

using namespace ydc;

// Analysis hook (tests/tools): per chunk, how far the level guess was from the start state the
// chunk really has — summed over the classes, and the largest single class (list positions).
static uint32_t* g_dev_sum = nullptr;
static uint32_t* g_dev_max = nullptr;
extern "C" void model_set_deviation_out(uint32_t* sum, uint32_t* mx) {
  g_dev_sum = sum;
  g_dev_max = mx;
}

extern "C" {

struct model_stats {
  uint32_t n_slots, n_classes, key_bits, n_chunks, rounds, chunk_sims, force_fp64;
};

// env_mask: env_words words per servant. Returns 0, or -6 if the rounds do not converge.
// n_alias further (host id, servant) entries of the requestor-address lookup table
// (ydc_set_host_aliases).
int model_dispatch_alias(uint32_t S, const uint32_t* version, const uint32_t* nproc, const uint32_t* load,
                         const uint32_t* max_tasks, const uint32_t* running, const uint32_t* flags,
                         const uint64_t* env_mask, uint32_t env_words, const uint32_t* ip_id, uint32_t N,
                         const uint32_t* env_id, const uint32_t* min_version,
                         const uint32_t* requestor_ip, uint32_t chunk_size, int force_fp64,
                         uint32_t* out_idx, double* out_util, uint32_t* out_running,
                         model_stats* stats, uint32_t n_alias, const uint32_t* alias_ip,
                         const uint32_t* alias_servant) {
  HostTables T;
  T.build(S, env_mask, version, max_tasks, nproc, ip_id, env_words, n_alias, alias_ip, alias_servant);
  const uint32_t NI = (uint32_t)T.ip_sorted.size();
  const uint32_t C = T.n_classes();
  uint32_t G = T.n_comp;  // independent parts of the registry (host_tables.h)
  KeyFormat kf = choose_key_format(force_fp64 ? 32 : T.cap_bits, 11, &G);
  auto comp_of = [&](uint32_t cls) { return G > 1 ? T.cls_comp[cls] : 0u; };

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
  std::vector<uint32_t> comp_rank_base(G + 1, 0);  // slots of the parts before g
  for (uint32_t s = 0; s < S; ++s) {
    for (uint32_t g = base[s]; g < base[s + 1]; ++g) {
      uint32_t r = running[s] + (g - base[s]);
      uint32_t cap = slot_capacity(nproc[s], load[s], max_tasks[s], r);
      uint32_t tier = slot_tier(nproc[s], flags[s], r);
      key[g] = kf.exact ? slot_key_exact(tier, r, cap, kf.cap_bits) : slot_key_fp64(tier, r, cap);
      if (G > 1) key[g] |= (uint64_t)comp_of(T.class_of[s]) << kf.comp_shift;  // part-major order
      comp_rank_base[comp_of(T.class_of[s]) + 1]++;
    }
  }
  for (uint32_t g = 0; g < G; ++g) comp_rank_base[g + 1] += comp_rank_base[g];
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
  // Cross-check of the bin sort's first launch (bin_sort.h: k_servant_scan_bins): the start of
  // every key bin in every class list, summed from per-servant closed forms, against the
  // sorted lists themselves.
  if (kf.exact && kf.key_bits <= 32 && M) {
    const BinFormat bf = choose_bins(kf.key_bits, T.max_slots);
    std::vector<uint32_t> closed(C), counted(C);
    for (uint32_t j = 1; j <= bf.n_bins; ++j) {
      const uint64_t K = (uint64_t)j << bf.shift;
      std::fill(closed.begin(), closed.end(), 0u);
      std::fill(counted.begin(), counted.end(), 0u);
      for (uint32_t s = 0; s < S; ++s) {
        const uint32_t c = T.class_of[s];
        if (c == kNone) continue;
        const uint64_t part_key = G > 1 ? (uint64_t)comp_of(c) << kf.comp_shift : 0ull;
        // (what bin_count_block evaluates: the 32-bit form where it applies, the last boundary
        // by the slot count)
        if (j == bf.n_bins)
          closed[c] += servant_slot_count(nproc[s], load[s], max_tasks[s], running[s], flags[s]);
        else if (kf.cap_bits <= 10)
          closed[c] += first_slot_not_below_direct32(nproc[s], load[s], max_tasks[s], running[s], flags[s],
                                                     (uint32_t)part_key, (uint32_t)K, kf.cap_bits) - running[s];
        else
          closed[c] += first_slot_not_below_direct(nproc[s], load[s], max_tasks[s], running[s], flags[s],
                                                   part_key, K, kf.cap_bits) - running[s];
      }
      for (uint32_t c = 0; c < C; ++c)
        for (uint32_t i = cls_begin[c]; i < cls_begin[c + 1]; ++i) counted[c] += key[list_g[i]] < K;
      if (closed != counted) return -8;
      if (j == bf.n_bins && std::accumulate(closed.begin(), closed.end(), 0u) != M) return -8;
    }
  }
  ClassLists L;
  L.list_p = C > 1 ? list_p.data() : nullptr;
  L.list_g = list_g.data();
  L.cls_begin = cls_begin.data();
  L.n_classes = C;
  L.cls_single = T.cls_single.data();

  // --- task classification.
  const uint32_t W = std::max<uint32_t>(1, (C + 63) / 64);
  std::vector<uint64_t> tmask((size_t)N * W);
  std::vector<uint32_t> tself_lo(N, kNone), tself_hi(N, kNone);
  bool need_shared = false;
  for (uint32_t t = 0; t < N; ++t) {
    task_class_mask(env_id[t], min_version[t], T.cls_env.data(), T.cls_ver.data(), C, W,
                    &tmask[(size_t)t * W], T.env_words);
    uint32_t i = lower_bound_u32(T.ip_sorted.data(), NI, requestor_ip[t]);
    if (i < NI && T.ip_sorted[i] == requestor_ip[t]) {
      if (i + 1 < NI && T.ip_sorted[i + 1] == requestor_ip[t]) {
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
  if (chunk_size == 0) chunk_size = N ? N : 1;
  const uint32_t K = N ? (N + chunk_size - 1) / chunk_size : 0;
  std::vector<ClassRun> runs(std::max<uint32_t>(C, 1));
  // Hosts that run several servants: `self` is resolved at replay time from the class state
  // (chunk-parallel like everything else), which needs every servant's last list position.
  std::vector<uint32_t> pos_last(S, 0);
  for (uint32_t i = 0; i < M; ++i) {
    const uint32_t g = list_g[i], s = owner_of_slot(base.data(), S, g);
    if (g + 1 == base[s + 1]) pos_last[s] = i;
  }
  SharedIpTable sh{T.ip_sorted.data(), T.ip_servant.data(), NI, T.class_of.data(),
                   base.data(),        S,                   pos_last.data()};
  uint32_t rounds = 0, sims = 0;
  std::vector<ClassState> final_state(C);
  if (K) {
    // consuming tasks before each chunk, per part of the registry
    std::vector<uint32_t> before((size_t)(K + 1) * G, 0);
    for (uint32_t k = 0; k < K; ++k) {
      for (uint32_t g = 0; g < G; ++g) before[(size_t)(k + 1) * G + g] = before[(size_t)k * G + g];
      for (uint32_t t = k * chunk_size; t < std::min(N, (k + 1) * chunk_size); ++t) {
        if (task_mask_empty(ti, t)) continue;
        uint32_t c = 0;  // any eligible class names the request's part
        for (uint32_t w = 0; w < W; ++w)
          if (tmask[(size_t)t * W + w]) {
            c = w * 64 + (uint32_t)__builtin_ctzll(tmask[(size_t)t * W + w]);
            break;
          }
        before[(size_t)(k + 1) * G + comp_of(c)]++;
      }
    }
    std::vector<ClassState> guess((size_t)K * C), endst((size_t)K * C);
    for (uint32_t k = 0; k < K; ++k)
      for (uint32_t c = 0; c < C; ++c)
        guess[(size_t)k * C + c] =
            level_guess(L, c, comp_rank_base[comp_of(c)] + before[(size_t)k * G + comp_of(c)]);
    const std::vector<ClassState> guess_level(guess);
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
      static const char* prefix_env = getenv("YDC_MODEL_PREFIX");
      const bool prefix_rule = prefix_env && atoi(prefix_env) != 0;
      if (prefix_rule) {
        // k_update_prefix: chunk k+1's cursor guess = true start + the sum of what the chunks
        // before it consumed in their latest replays (exact where the chain is consistent).
        std::vector<uint32_t> acc(C), from(C);  // from: the cursor chunk k was replayed from
        for (uint32_t c = 0; c < C; ++c) acc[c] = from[c] = guess[c].cursor;
        for (uint32_t k = 0; k + 1 < K; ++k) {
          for (uint32_t c = 0; c < C; ++c) {
            const ClassState& en = endst[(size_t)k * C + c];
            ClassState& g = guess[(size_t)(k + 1) * C + c];
            if (!class_state_equal(g, en)) any = true;  // (consistency is judged on the old guesses)
            acc[c] += en.cursor - from[c];
            from[c] = g.cursor;  // (before it is overwritten below)
            const uint32_t e = cls_begin[c + 1];
            const uint32_t cur = acc[c] > e ? e : acc[c];
            ClassState ng;
            if (cur == en.cursor) {
              ng = en;
            } else {
              ng.cursor = ng.lo = cur;
              ng.hown_lo = ng.hown_hi = kNone;
            }
            if (!class_state_equal(g, ng)) {
              g = ng;
              dirty[k + 1] = 1;
            }
          }
        }
      } else
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
    for (uint32_t c = 0; c < C; ++c) final_state[c] = endst[(size_t)(K - 1) * C + c];
    if (g_dev_sum)
      for (uint32_t k = 0; k < K; ++k) {
        uint32_t sum = 0, mx = 0;
        for (uint32_t c = 0; c < C; ++c) {
          const uint32_t a = guess_level[(size_t)k * C + c].cursor, b = guess[(size_t)k * C + c].cursor;
          const uint32_t d = a > b ? a - b : b - a;
          sum += d;
          mx = std::max(mx, d);
        }
        g_dev_sum[k] = sum;
        if (g_dev_max) g_dev_max[k] = mx;
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
  // Cross-check of what k_finalize does on the device: running_tasks in closed form from the
  // final class states (servant_slots_before) must equal the count over the placements.
  if (K && C) {
    for (uint32_t s = 0; s < S; ++s) {
      const uint32_t c = T.class_of[s];
      uint32_t taken = 0;
      if (c != kNone) {
        const ClassState& st = final_state[c];
        const bool holes_here = st.lo < st.cursor && st.hown_lo == base[s];
        const uint32_t x = holes_here ? st.lo : st.cursor;
        if (x >= cls_begin[c + 1]) {
          taken = servant_slot_count(nproc[s], load[s], max_tasks[s], running[s], flags[s]);
        } else {
          const uint32_t g = list_g[x], hs = owner_of_slot(base.data(), S, g);
          const uint32_t hr = running[hs] + (g - base[hs]);
          const uint64_t part_key = G > 1 ? (uint64_t)comp_of(c) << kf.comp_shift : 0ull;
          const uint64_t hkey = slot_sort_key(nproc[hs], load[hs], max_tasks[hs], flags[hs], hr,
                                              part_key, kf.exact, kf.cap_bits);
          taken = servant_slots_before(nproc[s], load[s], max_tasks[s], running[s], flags[s], s,
                                       part_key, hkey, hs, kf.exact, kf.cap_bits);
        }
      }
      if (running[s] + taken != run[s]) return -7;
    }
  }
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


int model_dispatch_wide(uint32_t S, const uint32_t* version, const uint32_t* nproc, const uint32_t* load,
                        const uint32_t* max_tasks, const uint32_t* running, const uint32_t* flags,
                        const uint64_t* env_mask, uint32_t env_words, const uint32_t* ip_id, uint32_t N,
                        const uint32_t* env_id, const uint32_t* min_version,
                        const uint32_t* requestor_ip, uint32_t chunk_size, int force_fp64,
                        uint32_t* out_idx, double* out_util, uint32_t* out_running,
                        model_stats* stats) {
  return model_dispatch_alias(S, version, nproc, load, max_tasks, running, flags, env_mask, env_words,
                              ip_id, N, env_id, min_version, requestor_ip, chunk_size, force_fp64,
                              out_idx, out_util, out_running, stats, 0, nullptr, nullptr);
}

int model_dispatch(uint32_t S, const uint32_t* version, const uint32_t* nproc, const uint32_t* load,
                   const uint32_t* max_tasks, const uint32_t* running, const uint32_t* flags,
                   const uint64_t* env_mask, const uint32_t* ip_id, uint32_t N,
                   const uint32_t* env_id, const uint32_t* min_version,
                   const uint32_t* requestor_ip, uint32_t chunk_size, int force_fp64,
                   uint32_t* out_idx, double* out_util, uint32_t* out_running,
                   model_stats* stats) {
  return model_dispatch_wide(S, version, nproc, load, max_tasks, running, flags, env_mask, 1, ip_id, N,
                             env_id, min_version, requestor_ip, chunk_size, force_fp64, out_idx,
                             out_util, out_running, stats);
}

// ---------------------------------------------------------------------------
// One rank of the multi-GPU sharding protocol, replayed on the CPU so that
// the protocol can be exercised by real processes exchanging over gloo
// (tests/test_sharded_protocol_gloo.py). Same steps as ydc_dispatch_sharded: every rank
// builds the same slot lists from the full servant table, classifies and replays only its
// own slice of the batch, publishes after every pass the end state of its last chunk and
// how many of its chunks were inconsistent, and finally its per-servant slot deltas.
// ---------------------------------------------------------------------------
struct model_shard {
  uint32_t S = 0, C = 0, N = 0, K = 0, W = 1, chunk = 256, G = 1;
  std::vector<uint32_t> cls_comp, comp_rank_base;  // parts of the registry (host_tables.h)
  std::vector<uint32_t> base, running, nproc, load, max_tasks;
  std::vector<uint32_t> cls_begin, list_p, list_g;
  std::vector<uint64_t> tmask;
  std::vector<uint32_t> tself_lo, tself_hi, slot_of, consuming_before;
  std::vector<ClassState> guess0, used_start, endst;
  std::vector<uint8_t> replayed_once;
  ClassLists L{};
  TaskTable ti{};
  uint32_t consuming_total = 0;
  // hosts that run several servants (run-time `self`)
  std::vector<uint32_t> ip_sorted, ip_servant, class_of, pos_last;
  SharedIpTable sh{};
  bool any_shared = false;
};

model_shard* model_shard_open(uint32_t S, const uint32_t* version, const uint32_t* nproc,
                              const uint32_t* load, const uint32_t* max_tasks,
                              const uint32_t* running, const uint32_t* flags,
                              const uint64_t* env_mask, const uint32_t* ip_id, uint32_t N,
                              const uint32_t* env_id, const uint32_t* min_version,
                              const uint32_t* requestor_ip, uint32_t chunk_size) {
  auto* m = new model_shard();
  HostTables T;
  T.build(S, env_mask, version, max_tasks, nproc, ip_id);
  const uint32_t C = T.n_classes();
  uint32_t G = T.n_comp;
  KeyFormat kf = choose_key_format(T.cap_bits, 11, &G);
  m->G = G;
  m->cls_comp.assign(C, 0);
  if (G > 1) m->cls_comp = T.cls_comp;
  m->comp_rank_base.assign(G + 1, 0);
  m->S = S;
  m->C = C;
  m->N = N;
  m->chunk = chunk_size ? chunk_size : 256;
  m->K = N ? (N + m->chunk - 1) / m->chunk : 0;
  m->running.assign(running, running + S);
  m->nproc.assign(nproc, nproc + S);
  m->load.assign(load, load + S);
  m->max_tasks.assign(max_tasks, max_tasks + S);
  m->base.assign(S + 1, 0);
  for (uint32_t s = 0; s < S; ++s) {
    uint32_t k = T.class_of[s] == kNone
                     ? 0
                     : servant_slot_count(nproc[s], load[s], max_tasks[s], running[s], flags[s]);
    m->base[s + 1] = m->base[s] + k;
  }
  const uint32_t M = m->base[S];
  std::vector<uint64_t> key(M);
  for (uint32_t s = 0; s < S; ++s)
    for (uint32_t g = m->base[s]; g < m->base[s + 1]; ++g) {
      uint32_t r = running[s] + (g - m->base[s]);
      uint32_t cap = slot_capacity(nproc[s], load[s], max_tasks[s], r);
      uint32_t tier = slot_tier(nproc[s], flags[s], r);
      key[g] = kf.exact ? slot_key_exact(tier, r, cap, kf.cap_bits) : slot_key_fp64(tier, r, cap);
      const uint32_t part = m->cls_comp[T.class_of[s]];
      if (G > 1) key[g] |= (uint64_t)part << kf.comp_shift;
      m->comp_rank_base[part + 1]++;
    }
  for (uint32_t g = 0; g < G; ++g) m->comp_rank_base[g + 1] += m->comp_rank_base[g];
  std::vector<uint32_t> sorted_g(M);
  std::iota(sorted_g.begin(), sorted_g.end(), 0u);
  std::stable_sort(sorted_g.begin(), sorted_g.end(),
                   [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
  m->cls_begin.assign(C + 1, 0);
  m->list_p.resize(M);
  m->list_g.resize(M);
  {
    std::vector<uint32_t> cnt(C + 1, 0), owner(M);
    for (uint32_t p = 0; p < M; ++p) {
      owner[p] = owner_of_slot(m->base.data(), S, sorted_g[p]);
      cnt[T.class_of[owner[p]] + 1]++;
    }
    for (uint32_t c = 0; c < C; ++c) m->cls_begin[c + 1] = m->cls_begin[c] + cnt[c + 1];
    std::vector<uint32_t> fill(m->cls_begin.begin(), m->cls_begin.end() - 1);
    for (uint32_t p = 0; p < M; ++p) {
      uint32_t c = T.class_of[owner[p]];
      m->list_p[fill[c]] = p;
      m->list_g[fill[c]] = sorted_g[p];
      fill[c]++;
    }
  }
  m->L.list_p = C > 1 ? m->list_p.data() : nullptr;
  m->L.list_g = m->list_g.data();
  m->L.cls_begin = m->cls_begin.data();
  m->L.n_classes = C;
  m->W = std::max<uint32_t>(1, (C + 63) / 64);
  m->tmask.assign((size_t)N * m->W, 0);
  m->tself_lo.assign(N, kNone);
  m->tself_hi.assign(N, kNone);
  for (uint32_t t = 0; t < N; ++t) {
    task_class_mask(env_id[t], min_version[t], T.cls_env.data(), T.cls_ver.data(), C, m->W,
                    &m->tmask[(size_t)t * m->W]);
    uint32_t i = lower_bound_u32(T.ip_sorted.data(), S, requestor_ip[t]);
    if (i < S && T.ip_sorted[i] == requestor_ip[t]) {
      if (i + 1 < S && T.ip_sorted[i + 1] == requestor_ip[t]) {
        m->tself_lo[t] = i;  // several servants on the host: resolved at replay time
        m->tself_hi[t] = kSelfShared;
      } else {
        uint32_t s = T.ip_servant[i];
        if (m->base[s + 1] > m->base[s]) {
          m->tself_lo[t] = m->base[s];
          m->tself_hi[t] = m->base[s + 1];
        }
      }
    }
  }
  m->ti = TaskTable{m->tmask.data(), m->tself_lo.data(), m->tself_hi.data(), m->W};
  m->any_shared = T.any_shared_ip;
  m->ip_sorted = T.ip_sorted;
  m->ip_servant = T.ip_servant;
  m->class_of = T.class_of;
  m->pos_last.assign(S, 0);
  for (uint32_t i = 0; i < M; ++i) {
    const uint32_t g = m->list_g[i], s = owner_of_slot(m->base.data(), S, g);
    if (g + 1 == m->base[s + 1]) m->pos_last[s] = i;
  }
  m->sh = SharedIpTable{m->ip_sorted.data(), m->ip_servant.data(), S, m->class_of.data(),
                        m->base.data(),      S,                    m->pos_last.data()};
  // consuming_before[k * G + g]: requests of part g in the chunks before k (row K: the totals)
  m->consuming_before.assign((size_t)(m->K + 1) * G, 0);
  for (uint32_t k = 0; k < m->K; ++k) {
    for (uint32_t g = 0; g < G; ++g)
      m->consuming_before[(size_t)(k + 1) * G + g] = m->consuming_before[(size_t)k * G + g];
    for (uint32_t t = k * m->chunk; t < std::min(N, (k + 1) * m->chunk); ++t) {
      if (task_mask_empty(m->ti, t)) continue;
      uint32_t c = 0;
      for (uint32_t w = 0; w < m->W; ++w)
        if (m->tmask[(size_t)t * m->W + w]) {
          c = w * 64 + (uint32_t)__builtin_ctzll(m->tmask[(size_t)t * m->W + w]);
          break;
        }
      m->consuming_before[(size_t)(k + 1) * G + m->cls_comp[c]]++;
    }
  }
  m->consuming_total = 0;
  for (uint32_t g = 0; g < G; ++g) m->consuming_total += m->consuming_before[(size_t)m->K * G + g];
  m->slot_of.assign(N, kIdxTimeout);
  m->guess0.resize((size_t)m->K * C);
  m->used_start.resize((size_t)m->K * C);
  m->endst.resize((size_t)m->K * C);
  return m;
}

uint32_t model_shard_n_classes(const model_shard* m) { return m->C; }
uint32_t model_shard_consuming(const model_shard* m) { return m->consuming_total; }
uint32_t model_shard_n_parts(const model_shard* m) { return m->G; }
// out[g] = consuming requests of part g in this rank's slice (what the ranks all-gather first).
void model_shard_consuming_parts(const model_shard* m, uint32_t* out) {
  for (uint32_t g = 0; g < m->G; ++g) out[g] = m->consuming_before[(size_t)m->K * m->G + g];
}

// One matching pass. base[g]: consuming requests of part g on the ranks before this one (pass 0).
// boundary_in: end state of the previous rank's last chunk (NULL on rank 0; unused in pass
// 0). out_end[C]: what this rank publishes. Returns the number of inconsistent chunks
// (pass 0: the number of chunks).
uint32_t model_shard_pass(model_shard* m, uint32_t pass, const uint32_t* base,
                          const ClassState* boundary_in, ClassState* out_end) {
  const uint32_t C = m->C, K = m->K;
  std::vector<ClassRun> runs(std::max<uint32_t>(C, 1));
  uint32_t busy = 0;
  for (uint32_t k = 0; k < K; ++k) {
    const ClassState* start;
    std::vector<ClassState> lvl;
    if (pass == 0) {
      lvl.resize(C);
      for (uint32_t c = 0; c < C; ++c) {
        const uint32_t g = m->cls_comp[c];
        lvl[c] = level_guess(m->L, c, m->comp_rank_base[g] + base[g] + m->consuming_before[(size_t)k * m->G + g]);
      }
      start = lvl.data();
    } else {
      if (k == 0 && !boundary_in) continue;  // rank 0, chunk 0: started from the true state
      start = k == 0 ? boundary_in : &m->endst[(size_t)(k - 1) * C];
      bool same = true;
      for (uint32_t c = 0; c < C; ++c)
        same &= class_state_equal(start[c], m->used_start[(size_t)k * C + c]);
      if (same) continue;  // consistent
    }
    ++busy;
    std::vector<ClassState> st(start, start + C);  // endst[k-1] may be rewritten below? no: k ascending
    for (uint32_t c = 0; c < C; ++c) m->used_start[(size_t)k * C + c] = st[c];
    sim_chunk(m->L, m->ti, k * m->chunk, std::min(m->N, (k + 1) * m->chunk), st.data(),
              &m->endst[(size_t)k * C], m->slot_of.data(), runs.data(), m->any_shared ? &m->sh : nullptr);
  }
  for (uint32_t c = 0; c < C; ++c) {
    if (K) {
      out_end[c] = m->endst[(size_t)(K - 1) * C + c];
    } else if (boundary_in) {
      out_end[c] = boundary_in[c];
    } else {
      out_end[c].cursor = out_end[c].lo = m->cls_begin[c];
      out_end[c].hown_lo = out_end[c].hown_hi = kNone;
    }
  }
  return busy;
}

void model_shard_finalize(const model_shard* m, uint32_t* out_idx, uint32_t* out_delta) {
  std::memset(out_delta, 0, m->S * sizeof(uint32_t));
  for (uint32_t t = 0; t < m->N; ++t) {
    uint32_t g = m->slot_of[t];
    if (g >= kIdxEnvNotFound) {
      out_idx[t] = g;
      continue;
    }
    uint32_t s = owner_of_slot(m->base.data(), m->S, g);
    out_idx[t] = s;
    out_delta[s]++;
  }
}

void model_shard_close(model_shard* m) { delete m; }

}  // extern "C"

// first_slot_not_below (dispatch_core.h: the closed-form count behind the key windows of the
// multi-GPU path) against a linear walk over the servant's slots. Returns the mismatches.
extern "C" uint32_t model_check_first_slot(uint32_t seed, uint32_t iterations) {
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
  auto rnd = [&]() {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return (uint32_t)(x >> 16);
  };
  uint32_t bad = 0;
  for (uint32_t it = 0; it < iterations; ++it) {
    // (mostly small capacities; one in 64 up to 2^17 - 1: the 64-bit arithmetic)
    const uint32_t cap_bits = rnd() % 64 ? 1 + rnd() % 12 : 13 + rnd() % 5, lim = (1u << cap_bits) - 1;
    const uint32_t nproc = 1 + rnd() % (lim * 2), load = rnd() % (nproc * 5 / 4 + 1);
    uint32_t mt = rnd() % (lim + 1);
    const uint32_t top = mt < nproc ? mt : nproc;
    const uint32_t running = rnd() % (top + 2), flags = rnd() % 4;
    const uint64_t part = (uint64_t)(rnd() % 3) << (2 * cap_bits + 1);
    uint64_t K = part | (((uint64_t)rnd() << 16 | rnd()) % (1ull << (2 * cap_bits + 1)));
    if (rnd() % 10 == 0) K = part;
    if (rnd() % 10 == 0) K = (uint64_t)(rnd() % 4) << (2 * cap_bits + 1);  // another part's first key
    if (rnd() % 16 == 0) K = (K & ~((1ull << (2 * cap_bits)) - 1)) | (rnd() % 4);  // tiny quotients
    if (rnd() % 16 == 0) K |= (1ull << (2 * cap_bits)) - 1 - rnd() % 4;              // quotients near 1
    const uint32_t n = servant_slot_count(nproc, load, mt, running, flags);
    uint32_t want = running + n;
    for (uint32_t r = running; r < running + n; ++r) {
      const uint64_t k = part | slot_key_exact(slot_tier(nproc, flags, r), r,
                                               slot_capacity(nproc, load, mt, r), cap_bits);
      if (k >= K) {
        want = r;
        break;
      }
    }
    bad += first_slot_not_below(nproc, load, mt, running, flags, part, K, cap_bits) != want;
    bad += first_slot_not_below_direct(nproc, load, mt, running, flags, part, K, cap_bits) != want;
    if (cap_bits <= 10 && K < (1ull << 32) && part < (1ull << 32))
      bad += first_slot_not_below_direct32(nproc, load, mt, running, flags, (uint32_t)part, (uint32_t)K,
                                           cap_bits) != want;
  }
  return bad;
}

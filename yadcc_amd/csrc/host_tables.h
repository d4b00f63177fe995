// host_tables.h — host-side derived tables of the resident servant registry.
//
// Built once per registry change (upload / heartbeat), never per dispatch:
//   * servant classes: servants with max_tasks != 0 grouped by the signature
//     (env_mask, version) that decides eligibility for every possible request
//     (reference yadcc/scheduler/task_dispatcher.cc:324-338);
//   * the ip table: servants sorted by (ip_id, registry index), used to find a
//     request's own servant (`self`, :372-379) with a binary search;
//   * cap_bits: bit width of the largest capacity any pick can see, which fixes
//     the integer sort-key format (dispatch_core.h: slot_key_exact).
#ifndef YADCC_AMD_HOST_TABLES_H_
#define YADCC_AMD_HOST_TABLES_H_

#include <algorithm>
#include <cstdint>
#include <unordered_map>
#include <utility>
#include <vector>

#include "dispatch_core.h"

namespace ydc {

constexpr uint32_t kMaxComponents = 16;

struct HostTables {
  std::vector<uint32_t> class_of;   // per servant, kNone when max_tasks == 0
  std::vector<uint64_t> cls_env;    // per class, env_words words each
  uint32_t env_words = 1;           // 64-bit words of an environment mask
  std::vector<uint32_t> cls_ver;    // per class
  std::vector<uint32_t> ip_sorted;  // ip table, sorted by (ip, servant); >= n entries (aliases)
  std::vector<uint32_t> ip_servant;
  bool any_shared_ip = false;       // some host runs more than one servant
  // The same table as an open-addressing hash (what the request classification probes: one
  // 8-byte load answers most lookups, where the binary search took log2(n) dependent ones).
  // Slot i = {ip_hash[2 i], ip_hash[2 i + 1]} = {host id, value}; value kNone: empty slot;
  // value < 2^31: the ONE servant on that host; value = 2^31 | k: several, the first of them is
  // entry k of ip_sorted. Home slot of a host id: (id * 0x9E3779B1) >> ip_hash_shift, linear
  // probing, at most half full.
  std::vector<uint32_t> ip_hash;
  uint32_t ip_hash_shift = 28;
  // One bit per host id in front of it (eight bits per slot, i.e. >= 16 per host: 32 KB for 16k
  // hosts — it stays in a CU's L1, where a probe of the table itself costs a cache line from
  // L2 per LANE): bit (id * 0x85EBCA6B) >> ip_filter_shift. Nine requestors in ten are nobody's
  // host and never touch the table.
  std::vector<uint32_t> ip_filter;
  uint32_t ip_filter_shift = 25;
  // Slot tiles of the bin sort's front (bin_sort.h): tile t = servants [bin_tile_start[t],
  // bin_tile_start[t + 1]), cut so that the tiles' slot BOUNDS (min(max_tasks, nproc) of the
  // servants that take tasks) are about equal: at most kTileSlots of them, at most kTileServants
  // servants. What a servant really offers in a batch is its bound minus running_tasks and foreign
  // load, so a tile never holds more than its bound.
  static constexpr uint32_t kTileSlots = 640, kTileSlotsMax = 2048, kTileServants = 32;
  std::vector<uint32_t> bin_tile_start;
  // Eligible-class masks by (digest bit, version threshold), for registries with few distinct
  // class versions (and a table of at most 2^20 words): ver_sorted = the distinct class versions ascending,
  // env_ver_mask[(env * (V + 1) + vi) * words + w] = classes advertising digest `env` (one of
  // the 64 * env_words bit numbers) whose version is >= ver_sorted[vi] (vi == V: none). A
  // request (env, min_version) looks up vi = number of entries of ver_sorted below
  // min_version. Empty when not built.
  std::vector<uint32_t> ver_sorted;
  std::vector<uint64_t> env_ver_mask;
  // The same lookup as LISTS, for registries with many classes and sparse eligibility (a pool
  // whose machines advertise individual compiler sets: a request may use 2 % of ~2000 classes):
  // row r = env * (V + 1) + vi holds the eligible classes elig_cls[elig_off[r] .. elig_off[r + 1])
  // in ascending order. Built with env_ver_mask when there are more than kMaxWaveClasses classes;
  // the wave-per-chunk kernel reads a request's classes from its row when the rows are short
  // (elig_max_len <= 64) instead of scanning the mask words.
  std::vector<uint32_t> elig_off, elig_cls;
  uint32_t elig_max_len = 0;  // longest row
  // Independent parts of the registry: classes are linked when some request could take either
  // (they share a digest), and a request only ever compares slots of ONE part. cls_comp[c] is
  // the part of class c (ids below kMaxComponents; several parts may share an id, which is
  // always exact — ids only group). Slots are ordered part-major (the id rides above the sort
  // key), and the level guesses count consumed slots per part: with disjoint environment
  // partitions every part is consumed at its own requests' rate, not at the global one.
  std::vector<uint32_t> cls_comp;
  uint32_t n_comp = 1;
  // 1: the class consists of ONE servant. A request from that servant's host then has no
  // candidate in the class at all (every entry is its own) — known without walking the list.
  std::vector<uint8_t> cls_single;
  uint32_t cap_bits = 1;            // max over servants of bits(min(max_tasks, nproc))
  uint64_t max_slots = 0;           // sum over servants of min(max_tasks, nproc): bound on slots

  // env_mask: words_per_servant words per servant (word w of servant s at
  // env_mask[s * words_per_servant + w]).
  // Aliases (n_alias entries, nullable): further (host id, servant) entries of the ip table — a
  // servant whose location has several ':' answers to every prefix that ends before one of them
  // (IsNetworkAddressEqual, task_dispatcher.cc:66-69), i.e. to more than one requestor address.
  void build(uint32_t n, const uint64_t* env_mask, const uint32_t* version,
             const uint32_t* max_tasks, const uint32_t* nproc, const uint32_t* ip_id,
             uint32_t words_per_servant = 1, uint32_t n_alias = 0, const uint32_t* alias_ip = nullptr,
             const uint32_t* alias_servant = nullptr) {
    env_words = std::max<uint32_t>(1, words_per_servant);
    const uint32_t EW = env_words;
    class_of.assign(n, kNone);
    cls_env.clear();
    cls_ver.clear();
    // Signature (environment set, version) -> class. Buckets by hash, exact compare inside.
    auto sig_hash = [&](uint32_t s) {
      uint64_t h = 0x9E3779B97F4A7C15ull ^ version[s];
      for (uint32_t w = 0; w < EW; ++w)
        h = (h ^ env_mask[(size_t)s * EW + w]) * 0x100000001B3ull + (h >> 29);
      return h;
    };
    std::unordered_multimap<uint64_t, uint32_t> ids;  // hash -> class
    uint32_t max_cap = 1;
    max_slots = 0;
    for (uint32_t s = 0; s < n; ++s) {
      if (max_tasks[s] == 0) continue;
      const uint64_t h = sig_hash(s);
      uint32_t cls = kNone;
      auto range = ids.equal_range(h);
      for (auto it = range.first; it != range.second; ++it) {
        const uint32_t c = it->second;
        if (cls_ver[c] == version[s] &&
            std::equal(&cls_env[(size_t)c * EW], &cls_env[(size_t)c * EW] + EW,
                       &env_mask[(size_t)s * EW])) {
          cls = c;
          break;
        }
      }
      if (cls == kNone) {
        cls = (uint32_t)cls_ver.size();
        ids.emplace(h, cls);
        cls_env.insert(cls_env.end(), &env_mask[(size_t)s * EW], &env_mask[(size_t)s * EW] + EW);
        cls_ver.push_back(version[s]);
      }
      class_of[s] = cls;
      uint32_t top = std::min(max_tasks[s], nproc[s]);
      max_cap = std::max(max_cap, top);
      max_slots += top;
    }
    {
      std::vector<uint32_t> members(cls_ver.size(), 0);
      for (uint32_t s = 0; s < n; ++s)
        if (class_of[s] != kNone) members[class_of[s]]++;
      cls_single.assign(cls_ver.size(), 0);
      for (size_t c = 0; c < members.size(); ++c) cls_single[c] = members[c] == 1;
    }
    cap_bits = 1;
    while (cap_bits < 32 && (max_cap >> cap_bits)) ++cap_bits;
    ver_sorted.assign(cls_ver.begin(), cls_ver.end());
    std::sort(ver_sorted.begin(), ver_sorted.end());
    ver_sorted.erase(std::unique(ver_sorted.begin(), ver_sorted.end()), ver_sorted.end());
    env_ver_mask.clear();
    const uint32_t C = (uint32_t)cls_ver.size(), V = (uint32_t)ver_sorted.size();
    // (built while it stays small: a few MB at most — 150 digests x 3 thresholds x 2000 classes
    // are 150 KB; registries beyond that classify with the loop over the classes)
    if (C && V <= 16 && (size_t)64 * EW * (V + 1) * ((C + 63) / 64) <= ((size_t)1 << 20)) {
      const uint32_t words = (C + 63) / 64;
      env_ver_mask.assign((size_t)64 * EW * (V + 1) * words, 0);
      for (uint32_t c = 0; c < C; ++c)
        for (uint32_t env = 0; env < 64 * EW; ++env)
          if ((cls_env[(size_t)c * EW + env / 64] >> (env % 64)) & 1u)
            for (uint32_t vi = 0; vi < V && ver_sorted[vi] <= cls_ver[c]; ++vi)
              env_ver_mask[((size_t)env * (V + 1) + vi) * words + c / 64] |= 1ull << (c % 64);
      elig_off.clear();
      elig_cls.clear();
      elig_max_len = 0;
      if (C > kMaxWaveClasses) {
        const size_t rows = (size_t)64 * EW * (V + 1);
        elig_off.assign(rows + 1, 0);
        for (size_t r = 0; r < rows; ++r) {
          uint32_t n_r = 0;
          for (uint32_t w = 0; w < words; ++w) n_r += (uint32_t)__builtin_popcountll(env_ver_mask[r * words + w]);
          elig_off[r + 1] = elig_off[r] + n_r;
          elig_max_len = std::max(elig_max_len, n_r);
        }
        elig_cls.resize(elig_off[rows]);
        for (size_t r = 0; r < rows; ++r) {
          uint32_t at = elig_off[r];
          for (uint32_t w = 0; w < words; ++w)
            for (uint64_t m = env_ver_mask[r * words + w]; m; m &= m - 1)
              elig_cls[at++] = w * 64 + (uint32_t)__builtin_ctzll(m);
        }
      }
    } else {
      ver_sorted.clear();
      elig_off.clear();
      elig_cls.clear();
      elig_max_len = 0;
    }

    // Parts: union-find over the classes through the digests they advertise.
    {
      std::vector<uint32_t> parent(C);
      for (uint32_t c = 0; c < C; ++c) parent[c] = c;
      auto find = [&](uint32_t x) {
        while (parent[x] != x) x = parent[x] = parent[parent[x]];
        return x;
      };
      std::vector<uint32_t> first_with(64 * (size_t)EW, kNone);
      for (uint32_t c = 0; c < C; ++c)
        for (uint32_t w = 0; w < EW; ++w)
          for (uint64_t m = cls_env[(size_t)c * EW + w]; m; m &= m - 1) {
            const uint32_t bit = w * 64 + (uint32_t)__builtin_ctzll(m);
            if (first_with[bit] == kNone) first_with[bit] = c;
            else parent[find(c)] = find(first_with[bit]);
          }
      cls_comp.assign(C, 0);
      std::unordered_map<uint32_t, uint32_t> id_of;
      for (uint32_t c = 0; c < C; ++c) {
        auto it = id_of.emplace(find(c), (uint32_t)id_of.size()).first;
        cls_comp[c] = it->second % kMaxComponents;
      }
      n_comp = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)id_of.size(), kMaxComponents));
    }

    std::vector<std::pair<uint32_t, uint32_t>> byip(n);
    for (uint32_t s = 0; s < n; ++s) byip[s] = {ip_id[s], s};
    for (uint32_t a = 0; a < n_alias; ++a)
      if (alias_servant[a] < n) byip.push_back({alias_ip[a], alias_servant[a]});
    std::sort(byip.begin(), byip.end());
    byip.erase(std::unique(byip.begin(), byip.end()), byip.end());
    const uint32_t n_ip = (uint32_t)byip.size();
    ip_sorted.resize(n_ip);
    ip_servant.resize(n_ip);
    any_shared_ip = false;
    for (uint32_t i = 0; i < n_ip; ++i) {
      ip_sorted[i] = byip[i].first;
      ip_servant[i] = byip[i].second;
      if (i && byip[i].first == byip[i - 1].first) any_shared_ip = true;
    }
    bin_tile_start.assign(1, 0u);
    {
      uint32_t in_tile = 0, first = 0;
      for (uint32_t s = 0; s < n; ++s) {
        const uint32_t ub = class_of[s] == kNone ? 0u : std::min(max_tasks[s], nproc[s]);
        if (s > first && (in_tile + ub > kTileSlots || s - first >= kTileServants)) {
          bin_tile_start.push_back(s);
          first = s;
          in_tile = 0;
        }
        in_tile += ub;
      }
      bin_tile_start.push_back(n);
    }
    uint32_t bits = 4;
    while ((1u << bits) < 2 * n_ip) ++bits;
    ip_hash_shift = 32 - bits;
    ip_hash.assign((size_t)2 << bits, kNone);
    for (uint32_t i = 0; i < n_ip;) {
      uint32_t j = i + 1;
      while (j < n_ip && ip_sorted[j] == ip_sorted[i]) ++j;
      uint32_t h = (ip_sorted[i] * 0x9E3779B1u) >> ip_hash_shift;
      while (ip_hash[2 * h + 1] != kNone) h = (h + 1) & ((1u << bits) - 1);
      ip_hash[2 * h] = ip_sorted[i];
      ip_hash[2 * h + 1] = j - i == 1 ? ip_servant[i] : (0x80000000u | i);
      i = j;
    }
    ip_filter_shift = 32 - (bits + 3);
    ip_filter.assign((size_t)1 << (bits + 3 - 5), 0u);
    for (uint32_t i = 0; i < n_ip; ++i) {
      const uint32_t f = (ip_sorted[i] * 0x85EBCA6Bu) >> ip_filter_shift;
      ip_filter[f >> 5] |= 1u << (f & 31);
    }
  }

  uint32_t n_classes() const { return (uint32_t)cls_ver.size(); }
};

// Sort-key format for a registry.
struct KeyFormat {
  bool exact;         // slot_key_exact (integer) vs slot_key_fp64
  uint32_t cap_bits;  // only for exact
  uint32_t key_bits;  // significant bits, the part id included
  uint32_t passes;    // LSD radix passes
  uint32_t bits_per_pass;
  uint32_t comp_shift;  // the part id sits at bits [comp_shift, key_bits) (== key_bits: no parts)
};

// n_comp > 1: the part id (HostTables::cls_comp) rides above the slot key. The fp64 key has no
// room for it: such registries are treated as one part (*n_comp is set to 1).
// fuse_cls_bits != 0: the class partition wants to ride on the LAST key pass (kernels.h:
// k_radix_scatter_classed — one scatter pass fewer), which needs its digit to leave that many
// bits free: the earlier passes are widened until it does, where the pass count allows
// (23-bit keys + 5 class bits: 9 + 9 + (5 + 5) instead of 8 + 8 + 7 and a pass of its own).
inline KeyFormat choose_key_format(uint32_t cap_bits, uint32_t max_radix_bits = 11,
                                   uint32_t* n_comp = nullptr, uint32_t fuse_cls_bits = 0) {
  KeyFormat f;
  f.exact = cap_bits <= kMaxExactCapBits;
  f.cap_bits = cap_bits;
  f.key_bits = f.exact ? 2 * cap_bits + 1 : 64;
  f.comp_shift = f.key_bits;
  if (n_comp && *n_comp > 1) {
    if (f.exact) {
      uint32_t b = 1;
      while ((1u << b) < *n_comp) ++b;
      f.key_bits += b;
    } else {
      *n_comp = 1;
    }
  }
  f.passes = (f.key_bits + max_radix_bits - 1) / max_radix_bits;
  f.bits_per_pass = (f.key_bits + f.passes - 1) / f.passes;
  if (fuse_cls_bits && fuse_cls_bits < max_radix_bits && f.passes >= 2) {
    const uint32_t room = max_radix_bits - fuse_cls_bits;  // key bits the last digit may hold
    if (f.key_bits - (f.passes - 1) * f.bits_per_pass > room) {
      const uint32_t bpp = (f.key_bits - room + f.passes - 2) / (f.passes - 1);
      if (bpp <= max_radix_bits && bpp * (f.passes - 1) < f.key_bits) f.bits_per_pass = bpp;
    }
  }
  return f;
}

// Geometry of the bin sort (bin_sort.h): B = 2^b equal bins over the key space, at most 512
// slots of the registry's bound per bin on average (the synthetic pools fill ~3/4 of the bound,
// and their fullest bin holds ~3.5x the average), 16 <= B <= max_bins (and B <= 2^key_bits).
struct BinFormat {
  uint32_t n_bins, shift;  // bin of a slot = key >> shift
};
inline BinFormat choose_bins(uint32_t key_bits, uint64_t slot_bound, uint32_t max_bins = 2048) {
  uint32_t b = 4;
  while ((2u << b) <= max_bins && ((uint64_t)512 << b) < slot_bound) ++b;
  if (b > key_bits) b = key_bits;
  return BinFormat{1u << b, key_bits - b};
}

}  // namespace ydc
#endif  // YADCC_AMD_HOST_TABLES_H_

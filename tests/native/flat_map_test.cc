// tests/native/flat_map_test.cc — FlatStringMap (yadcc_amd/csrc/flat_string_map.h) against
// std::unordered_map under random inserts, lookups and erases, with keys shaped like the ones
// the host class stores (locations that differ in their last characters, 64-character digests
// with a long common prefix, short requestor addresses). Prints FLAT-MAP-OK and exits 0.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include "flat_string_map.h"

#define CHECK(cond)                                                                \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

static std::string Key(std::mt19937_64& rng, int family) {
  const unsigned i = (unsigned)(rng() % 6000);
  char buf[96];
  switch (family) {
    case 0: std::snprintf(buf, sizeof buf, "10.%u.%u.%u:8335", i >> 16, (i >> 8) & 255, i & 255); break;
    case 1: std::snprintf(buf, sizeof buf, "c0ffee00c0ffee00c0ffee00c0ffee00c0ffee00c0ffee00%016x", i); break;
    case 2: std::snprintf(buf, sizeof buf, "%u", i % 700); break;
    default: std::snprintf(buf, sizeof buf, "[fe80::%x]:%u", i, 1000 + i % 7); break;
  }
  return buf;
}

int main() {
  std::mt19937_64 rng(12345);
  for (int round = 0; round < 4; ++round) {
    ydc::FlatStringMap<unsigned> flat;
    std::unordered_map<std::string, unsigned> ref;
    for (int step = 0; step < 200000; ++step) {
      const std::string k = Key(rng, (int)(rng() % 4));
      const unsigned op = (unsigned)(rng() % 10);
      if (op < 4) {
        const unsigned v = (unsigned)rng();
        auto [p, fresh] = flat.emplace(k, v);
        auto [it, fresh_ref] = ref.emplace(k, v);
        CHECK(fresh == fresh_ref && *p == it->second);
      } else if (op < 7) {
        const unsigned* p = flat.find(k);
        auto it = ref.find(k);
        CHECK((p != nullptr) == (it != ref.end()));
        if (p) CHECK(*p == it->second);
      } else if (op < 9) {
        CHECK(flat.erase(k) == (ref.erase(k) != 0));
      } else {
        flat[k] += 1;
        ref[k] += 1;
      }
      CHECK(flat.size() == ref.size());
    }
    std::size_t seen = 0;
    flat.for_each([&](const std::string& k, const unsigned& v) {
      auto it = ref.find(k);
      CHECK(it != ref.end() && it->second == v);
      ++seen;
    });
    CHECK(seen == ref.size());
    for (auto&& [k, v] : ref) CHECK(flat.find(k) && *flat.find(k) == v);
    if (round == 2) {
      flat.clear();
      CHECK(flat.empty() && !flat.find("10.0.0.1:8335"));
    }
  }
  // The hash must spread keys that differ only in their last characters (the index comes from
  // the low bits): a full table of consecutive locations stays within a few probes per miss.
  ydc::FlatStringMap<unsigned> t;
  for (unsigned i = 0; i < 16000; ++i) {
    char b[32];
    std::snprintf(b, sizeof b, "10.%u.%u.%u", i >> 16, (i >> 8) & 255, i & 255);
    t.emplace(b, i);
  }
  for (unsigned i = 0; i < 16000; ++i) {
    char b[32];
    std::snprintf(b, sizeof b, "10.%u.%u.%u", i >> 16, (i >> 8) & 255, i & 255);
    CHECK(t.find(b) && *t.find(b) == i);
    std::snprintf(b, sizeof b, "172.16.%u.%u", (i >> 8) & 255, i & 255);
    CHECK(!t.find(b));
  }
  std::puts("FLAT-MAP-OK");
  return 0;
}

// flat_string_map.h — the string -> small value tables of the host class (digest -> bit number,
// requestor address -> host id, location -> registry index).
//
// The reference finds a servant by comparing location strings in a loop and an environment by
// comparing 64-byte digests (task_dispatcher.cc:55-63,190-220); the host class looks both up
// once per request. std::unordered_map<std::string, …> costs a temporary std::string (a heap
// allocation for a 64-character digest), a byte-wise hash and a node chase per lookup —
// together most of the 119 ns per request the class spent on the host in round 3. This table is
// open addressing with linear probing (hashes in an array of their own, at most half full),
// looked up with a string_view, hashed eight bytes at a time; erase shifts the run back (no
// tombstones).
#ifndef YADCC_AMD_FLAT_STRING_MAP_H_
#define YADCC_AMD_FLAT_STRING_MAP_H_

#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

namespace ydc {

// Strings of more than 32 bytes (compiler digests: 64 hex characters of a hash) are hashed by
// their length, first 16 and last 16 bytes — the lookup compares the whole key anyway. No
// variable-length memcpy: tails are read as overlapping fixed-size loads.
inline std::uint64_t HashMix(std::uint64_t h, std::uint64_t w) {
  // 64 x 64 -> 128-bit multiply, halves folded: every input bit reaches the low bits the table
  // index is taken from (keys here differ in their LAST characters — "10.0.3.17" / "10.0.3.18" —
  // i.e. in the top bits of a word, which a 64-bit product only carries upwards).
  const unsigned __int128 m = (unsigned __int128)(h ^ w) * 0x9FB21C651E98DF25ull;
  return (std::uint64_t)m ^ (std::uint64_t)(m >> 64);
}
inline std::uint64_t HashLoad8(const char* p) {
  std::uint64_t w;
  std::memcpy(&w, p, 8);
  return w;
}
inline std::uint64_t HashBytes(const char* p, std::size_t n) {
  std::uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xFF51AFD7ED558CCDull);
  if (n > 32) {
    h = HashMix(h, HashLoad8(p));
    h = HashMix(h, HashLoad8(p + 8));
    h = HashMix(h, HashLoad8(p + n - 16));
    h = HashMix(h, HashLoad8(p + n - 8));
  } else if (n >= 8) {
    std::size_t i = 0;
    for (; i + 8 <= n; i += 8) h = HashMix(h, HashLoad8(p + i));
    if (i < n) h = HashMix(h, HashLoad8(p + n - 8));  // (overlaps the block before it)
  } else if (n >= 4) {
    std::uint32_t a, b;
    std::memcpy(&a, p, 4);
    std::memcpy(&b, p + n - 4, 4);
    h = HashMix(h, a | (std::uint64_t)b << 32);
  } else if (n) {
    h = HashMix(h, (std::uint64_t)(unsigned char)p[0] | (std::uint64_t)(unsigned char)p[n >> 1] << 8 |
                       (std::uint64_t)(unsigned char)p[n - 1] << 16);
  }
  h = HashMix(h, 0xD6E8FEB86659FD93ull);
  return h | 1;  // (0 marks an empty slot)
}

template <class V>
class FlatStringMap {
 public:
  FlatStringMap() { resize(16); }

  std::size_t size() const { return size_; }
  bool empty() const { return size_ == 0; }

  V* find(std::string_view key) { return const_cast<V*>(static_cast<const FlatStringMap*>(this)->find(key)); }
  const V* find(std::string_view key) const {
    const std::uint64_t h = HashBytes(key.data(), key.size());
    // Most lookups of the busiest table miss (nine requestor addresses in ten are no servant's
    // host): one bit per key in a filter sixteen times the size of the table answers those
    // without a probe sequence — and without the mispredicted branch that ends one.
    if (!(filter_[(h >> 24) & filter_mask_] >> ((h >> 18) & 63) & 1)) return nullptr;
    const std::size_t i = locate(key, h);
    return hashes_[i] ? &entries_[i].value : nullptr;
  }
  bool count(std::string_view key) const { return find(key) != nullptr; }

  // Inserts (key, value) unless the key is there; returns the stored value and whether it is new.
  std::pair<V*, bool> emplace(std::string_view key, V value) {
    if ((size_ + 1) * 2 > hashes_.size()) grow();  // at most half full: a miss ends after ~2 probes
    const std::uint64_t h = HashBytes(key.data(), key.size());
    const std::size_t i = locate(key, h);
    if (hashes_[i]) return {&entries_[i].value, false};
    hashes_[i] = h;
    filter_[(h >> 24) & filter_mask_] |= 1ull << ((h >> 18) & 63);
    entries_[i].key.assign(key.data(), key.size());
    entries_[i].value = std::move(value);
    ++size_;
    return {&entries_[i].value, true};
  }
  V& operator[](std::string_view key) { return *emplace(key, V{}).first; }

  bool erase(std::string_view key) {
    const std::size_t mask = hashes_.size() - 1;
    const std::size_t i = locate(key, HashBytes(key.data(), key.size()));
    if (!hashes_[i]) return false;
    // Backward shift: every entry of the run behind the hole moves up unless that would put it
    // in front of its home slot.
    std::size_t hole = i;
    for (std::size_t j = (i + 1) & mask; hashes_[j] != 0; j = (j + 1) & mask) {
      const std::size_t home = Home(hashes_[j]) & mask;
      // (cyclic) `home` outside (hole, j]: the entry may move into the hole
      const bool movable = hole <= j ? (home <= hole || home > j) : (home <= hole && home > j);
      if (movable) {
        hashes_[hole] = hashes_[j];
        entries_[hole] = std::move(entries_[j]);
        hole = j;
      }
    }
    hashes_[hole] = 0;
    entries_[hole].key.clear();
    entries_[hole].value = V{};
    --size_;
    return true;
  }

  void clear() {
    resize(16);
    size_ = 0;
  }

  template <class F>
  void for_each(F&& f) const {
    for (std::size_t i = 0; i != hashes_.size(); ++i)
      if (hashes_[i]) f(entries_[i].key, entries_[i].value);
  }

 private:
  struct Entry {
    std::string key;
    V value{};
  };
  // Bit 0 of a stored hash is forced to 1 (HashBytes): the home slot comes from the bits above it,
  // or every key would start on an odd slot.
  static std::size_t Home(std::uint64_t h) { return (std::size_t)(h >> 1); }

  // The slot of `key`, or the empty slot its probe sequence ends at. The hashes are an array of
  // their own: a probe that does not match touches eight of them per cache line and no key.
  std::size_t locate(std::string_view key, std::uint64_t h) const {
    const std::size_t mask = hashes_.size() - 1;
    for (std::size_t i = Home(h) & mask;; i = (i + 1) & mask) {
      if (hashes_[i] == 0) return i;
      if (hashes_[i] == h && entries_[i].key.size() == key.size() &&
          std::memcmp(entries_[i].key.data(), key.data(), key.size()) == 0)
        return i;
    }
  }
  void resize(std::size_t n) {
    hashes_.assign(n, 0);
    entries_.clear();
    entries_.resize(n);
    // 16 filter bits per slot (>= 32 per key); erased keys leave their bits behind until the
    // table grows — a stale bit only costs the probe it was meant to save.
    filter_.assign(n / 4, 0);
    filter_mask_ = n / 4 - 1;
  }
  void grow() {
    std::vector<std::uint64_t> old_hashes;
    std::vector<Entry> old_entries;
    old_hashes.swap(hashes_);
    old_entries.swap(entries_);
    resize(old_hashes.size() * 2);
    const std::size_t mask = hashes_.size() - 1;
    for (std::size_t k = 0; k != old_hashes.size(); ++k) {
      if (!old_hashes[k]) continue;
      std::size_t i = Home(old_hashes[k]) & mask;
      while (hashes_[i]) i = (i + 1) & mask;
      hashes_[i] = old_hashes[k];
      filter_[(old_hashes[k] >> 24) & filter_mask_] |= 1ull << ((old_hashes[k] >> 18) & 63);
      entries_[i] = std::move(old_entries[k]);
    }
  }

  std::vector<std::uint64_t> hashes_;  // 0: empty
  std::vector<std::uint64_t> filter_;
  std::size_t filter_mask_ = 0;
  std::vector<Entry> entries_;
  std::size_t size_ = 0;
};

}  // namespace ydc
#endif  // YADCC_AMD_FLAT_STRING_MAP_H_

// tests/native/td_linearize.cc — concurrent callers on the TaskDispatcher surface, natively (no
// interpreter between the threads and the library), through the C-ABI alone: links the CPU
// stand-in (libtd_stub.so; also built with -fsanitize=thread) or libydc.so on the GPU box.
//
//   td_linearize <out.json> [servants] [threads] [calls per thread] [max_tasks cap, 0 = none] [seed]
//
// Caller threads mix single requests (some parked with a deadline), small batches, FreeTask
// followed at once by the next request, lease renewals; a freer thread frees grants handed over to
// it; a registry thread sends heartbeats with new loads / changed machines and servant reports; a
// clock thread advances the injected clock and fires OnExpirationTimer. The dispatcher records the
// order in which the calls took effect (ydc_td_oplog_enable). Written to <out.json>: that log, the
// final DumpInternals and every thread's own calls in program order with the answers it got and
// the wall-clock interval of each call. tests/td_scenarios.py:verify_linearizable replays the log
// through the reference class (task_dispatcher.cc:93-140,142-277,498-536) and checks the three
// orders (program, real time, reference).
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "yadcc_dispatch.h"

namespace {
using Clk = std::chrono::steady_clock;
long long NowNs() { return std::chrono::duration_cast<std::chrono::nanoseconds>(Clk::now().time_since_epoch()).count(); }

struct Rng {
  unsigned long long s;
  unsigned next() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (unsigned)(s >> 33);
  }
  unsigned below(unsigned n) { return next() % n; }
};

struct Op {
  int kind;  // 0 wait, 1 free
  int st;
  unsigned long long id;
  std::string loc;
  long long t0, t1;
};

struct Pool {
  std::vector<std::string> locations;
  std::vector<std::vector<const char*>> envs;
  std::vector<ydc_td_servant> rows;
};

std::vector<std::string> g_digests;

void Personality(Rng& r, int i, unsigned cap, Pool* p) {
  ydc_td_servant& s = p->rows[i];
  const unsigned nprocs[] = {8, 16, 32, 64};
  s.version = r.below(4) ? 20 : 19;
  s.num_processors = nprocs[r.below(4)];
  const bool ded = r.below(10) < 3;
  s.priority = ded ? 1 : 2;
  s.max_tasks = r.below(20) == 0 ? 0 : s.num_processors * (ded ? 95 : 40) / 100;
  if (cap && s.max_tasks > cap) s.max_tasks = cap;
  s.current_load = r.below((unsigned)(s.num_processors * 5 / 4));
  s.total_memory_in_bytes = r.below(5) ? 64ull << 30 : 0;
  s.memory_available_in_bytes = r.below(14) ? 32ull << 30 : 1ull << 30;
  s.not_accepting_task_reason = 0;
  p->envs[i].clear();
  for (int d = 0; d < 4; ++d)
    if (r.below(10) < 6) p->envs[i].push_back(g_digests[d].c_str());
  if (p->envs[i].empty()) p->envs[i].push_back(g_digests[r.below(4)].c_str());
  s.env_digests = p->envs[i].data();
  s.n_envs = p->envs[i].size();
  s.observed_location = s.reported_location = p->locations[i].c_str();
}

void JsonStr(std::FILE* f, const std::string& s) {
  std::fputc('"', f);
  for (char c : s) {
    if (c == '"' || c == '\\') std::fputc('\\', f);
    std::fputc(c, f);
  }
  std::fputc('"', f);
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: td_linearize out.json [servants threads calls cap seed]\n");
    return 2;
  }
  const int n_servants = argc > 2 ? std::atoi(argv[2]) : 2000;
  const int n_threads = argc > 3 ? std::atoi(argv[3]) : 12;
  const int calls = argc > 4 ? std::atoi(argv[4]) : 2000;
  const unsigned cap = argc > 5 ? (unsigned)std::atoi(argv[5]) : 0;
  const unsigned long long seed = argc > 6 ? std::strtoull(argv[6], nullptr, 10) : 1;
  for (int i = 0; i < 4; ++i) {
    char b[80];
    std::snprintf(b, sizeof b, "c0ffee%04d%054d", i, 0);
    g_digests.push_back(b);
  }
  ydc_td* td = nullptr;
  if (ydc_td_create(0, nullptr, /*start_timer=*/0, /*fake_clock=*/1, &td) != YDC_OK || ydc_td_device_status(td) != YDC_OK) {
    std::fprintf(stderr, "no dispatcher / device\n");
    return 2;
  }
  std::atomic<long long> now_ns{0};
  ydc_td_set_clock_ns(td, 0);
  ydc_td_oplog_enable(td, 1);
  Pool pool;
  pool.locations.resize(n_servants);
  pool.envs.resize(n_servants);
  pool.rows.resize(n_servants);
  Rng r0{seed * 977 + 1};
  std::vector<long long> lease_ms(n_servants);
  for (int i = 0; i < n_servants; ++i) {
    // (every fifth servant shares its host with the one before: the `self` rule, task_dispatcher.cc:372-396)
    const int host = i % 5 == 4 ? i - 1 : i;
    pool.locations[i] = "10." + std::to_string(2 + host / 60000) + "." + std::to_string((host / 250) % 240) + "." +
                        std::to_string(host % 250) + ":" + std::to_string(8335 + (i - host));
    std::memset(&pool.rows[i], 0, sizeof pool.rows[i]);
    Personality(r0, i, cap, &pool);
    lease_ms[i] = i % 7 == 0 ? 3000 + (long long)r0.below(6000) : 3600000;
    if (ydc_td_keep_servant_alive(td, &pool.rows[i], lease_ms[i] * 1000000) != YDC_OK) return 3;
  }
  std::atomic<bool> stop{false};
  std::atomic<int> failed{0};
  std::mutex mu;  // handed_over, pool rows (registry thread only writes; callers never read rows)
  std::vector<unsigned long long> handed_over;
  std::vector<std::vector<Op>> hist(n_threads + 1);
  std::vector<std::string> ips(n_threads);

  auto caller = [&](int k) {
    Rng r{seed * 1000 + (unsigned)k};
    ips[k] = k % 3 ? "172.20." + std::to_string(k) + ".7"
                   : pool.locations[(std::size_t)k * 37 % n_servants].substr(0, pool.locations[(std::size_t)k * 37 % n_servants].find(':'));
    const std::string& ip = ips[k];
    std::vector<unsigned long long> mine;
    std::vector<Op>& h = hist[k];
    h.reserve(calls * 2);
    char loc[64];
    for (int c = 0; c < calls && !stop.load(std::memory_order_relaxed); ++c) {
      const unsigned ev = r.below(100);
      if (ev < 55 || mine.empty()) {
        const std::string dg = r.below(50) == 0 ? std::string("unknown") : g_digests[r.below(4)];
        const long long leases[] = {40, 200, 60000};
        const long long lease = leases[r.below(3)], timeout = r.below(4) == 0 ? 2 : 0;
        std::uint64_t id = 0;
        Op op{0, 0, 0, "", NowNs(), 0};
        const int rc = ydc_td_wait_for_starting_new_task(td, ip.c_str(), r.below(2) ? 20 : 0, dg.c_str(), lease * 1000000,
                                                         timeout * 1000000, 0, &id, loc, sizeof loc);
        op.t1 = NowNs();
        if (rc < 0) {
          ++failed;
          stop = true;
          return;
        }
        op.st = rc;
        if (rc == YDC_TD_GRANTED) {
          op.id = id;
          op.loc = loc;
          mine.push_back(id);
        }
        h.push_back(op);
      } else if (ev < 63) {
        const std::size_t n = 2 + r.below(7);
        const std::string& dg = g_digests[r.below(4)];
        const char* ipp[8];
        const char* dgp[8];
        std::uint32_t mv[8] = {0};
        std::int32_t st[8];
        std::uint64_t ids[8];
        char locs[8 * 64];
        for (std::size_t j = 0; j < n; ++j) ipp[j] = ip.c_str(), dgp[j] = dg.c_str();
        const long long t0 = NowNs();
        const int rc = ydc_td_wait_for_starting_new_tasks(td, n, ipp, mv, dgp, 60000ll * 1000000, nullptr, st, ids, locs, 64);
        const long long t1 = NowNs();
        if (rc < 0) {
          ++failed;
          stop = true;
          return;
        }
        for (std::size_t j = 0; j < n; ++j) {
          Op op{0, st[j], 0, "", t0, t1};
          if (st[j] == YDC_TD_GRANTED) {
            op.id = ids[j];
            op.loc = locs + j * 64;
            mine.push_back(ids[j]);
          }
          h.push_back(op);
        }
      } else if (ev < 90) {
        const std::size_t at = r.below((unsigned)mine.size());
        const unsigned long long id = mine[at];
        mine[at] = mine.back();
        mine.pop_back();
        if (r.below(5) == 0) {
          std::scoped_lock _(mu);
          handed_over.push_back(id);
        } else {
          Op op{1, 0, id, "", NowNs(), 0};
          ydc_td_free_task(td, id);
          op.t1 = NowNs();
          h.push_back(op);
        }
      } else {
        ydc_td_keep_task_alive(td, mine[r.below((unsigned)mine.size())], (r.below(2) ? 50ll : 5000ll) * 1000000);
      }
    }
    std::scoped_lock _(mu);
    handed_over.insert(handed_over.end(), mine.begin(), mine.end());
  };
  auto freer = [&] {
    std::vector<Op>& h = hist[n_threads];
    for (;;) {
      unsigned long long id = ~0ull;
      {
        std::scoped_lock _(mu);
        if (!handed_over.empty()) {
          id = handed_over.back();
          handed_over.pop_back();
        }
      }
      if (id == ~0ull) {
        if (stop.load()) return;
        std::this_thread::sleep_for(std::chrono::microseconds(100));
        continue;
      }
      Op op{1, 0, id, "", NowNs(), 0};
      ydc_td_free_task(td, id);
      op.t1 = NowNs();
      h.push_back(op);
    }
  };
  auto registry = [&] {
    Rng r{seed * 31 + 7};
    std::uint64_t unknown[8];
    while (!stop.load()) {
      const int i = (int)r.below((unsigned)n_servants);
      if (r.below(5)) {
        pool.rows[i].current_load = r.below((unsigned)(pool.rows[i].num_processors * 5 / 4));
      } else {
        Personality(r, i, cap, &pool);
      }
      ydc_td_keep_servant_alive(td, &pool.rows[i], lease_ms[i] * 1000000);
      if (r.below(3) == 0) {
        ydc_td_running_task rep[5];
        std::memset(rep, 0, sizeof rep);
        for (auto& t : rep) {
          t.task_grant_id = r.below(4000);
          t.servant_location = pool.locations[i].c_str();
          t.task_digest = "x";
        }
        ydc_td_notify_servant_running_tasks(td, pool.locations[i].c_str(), rep, 5, unknown, 8);
      }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  };
  auto clock = [&] {
    for (int tick = 1; !stop.load(); ++tick) {
      ydc_td_set_clock_ns(td, now_ns.fetch_add(1000000) + 1000000);
      if (tick % 25 == 0) ydc_td_on_expiration_timer(td);
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  };
  std::vector<std::thread> callers, others;
  others.emplace_back(freer);
  others.emplace_back(registry);
  others.emplace_back(clock);
  for (int k = 0; k < n_threads; ++k) callers.emplace_back(caller, k);
  for (auto& t : callers) t.join();
  stop = true;
  for (auto& t : others) t.join();
  if (failed.load()) {
    std::fprintf(stderr, "device error in a caller\n");
    return 1;
  }
  const std::string log = ydc_td_oplog_take(td);
  ydc_td_oplog_enable(td, 0);
  const std::string dump = ydc_td_dump_internals(td);
  ydc_td_stats hs{};
  ydc_td_host_stats(td, &hs);
  std::FILE* f = std::fopen(argv[1], "w");
  if (!f) return 4;
  std::fprintf(f, "{\"requests_per_device_turn\": %.3f, \"dump\": %s,\n\"threads\": [", (double)hs.requests / (hs.batches ? hs.batches : 1),
               dump.c_str());
  for (int k = 0; k <= n_threads; ++k) {
    std::fprintf(f, "%s{\"ip\": ", k ? ",\n" : "");
    if (k < n_threads) JsonStr(f, ips[k]); else std::fputs("null", f);
    std::fputs(", \"ops\": [", f);
    bool first = true;
    for (const Op& op : hist[k]) {
      std::fputs(first ? "" : ", ", f);
      first = false;
      if (op.kind == 0) {
        std::fprintf(f, "[\"wait\", %d, ", op.st);
        if (op.st == YDC_TD_GRANTED) {
          std::fprintf(f, "%llu, ", op.id);
          JsonStr(f, op.loc);
        } else {
          std::fputs("null, null", f);
        }
        std::fprintf(f, ", %lld, %lld]", op.t0, op.t1);
      } else {
        std::fprintf(f, "[\"free\", %llu, %lld, %lld]", op.id, op.t0, op.t1);
      }
    }
    std::fputs("]}", f);
  }
  std::fprintf(f, "],\n\"log\": %s}\n", log.c_str());
  std::fclose(f);
  ydc_td_destroy(td);
  std::printf("TD-LINEARIZE-WRITTEN %s\n", argv[1]);
  return 0;
}

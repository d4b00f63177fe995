// tests/native/td_concurrency_test.cc — the host class GpuTaskDispatcher under concurrent
// callers, built against the CPU stand-in of the device API (ydc_stub.cc) so that it runs
// under -fsanitize=thread / -fsanitize=address without a GPU (`make tsan`, `make asan`).
//
//  1. Wake-up order (reference task_dispatcher.cc:116-118,187,190-220): a parked waiter is
//     NOT woken by a heartbeat that adds capacity — a new caller takes that capacity — and IS
//     woken by FreeTask's notify_all.
//  2. A waiter completed by another thread's combined batch learns about it at once.
//  3. Many threads mixing all six public methods and the expiration timer: grant ids unique,
//     books balance.
// Prints TD-CONCURRENCY-OK and exits 0 when everything holds.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "gpu_task_dispatcher.h"

using namespace ydc;
using namespace std::literals;

#define CHECK(cond)                                                                  \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

static ServantPersonality Servant(const std::string& loc, std::size_t cap, const std::string& digest) {
  ServantPersonality s;
  s.version = 20;
  s.observed_location = s.reported_location = loc;
  s.environments = {digest};
  s.num_processors = 64;
  s.current_load = 0;
  s.max_tasks = cap;
  s.memory_available_in_bytes = 50ull << 30;
  s.priority = kServantPriorityUser;
  return s;
}

static void WakeOrder() {
  GpuTaskDispatcher::Options opt;
  opt.start_expiration_timer = false;
  GpuTaskDispatcher td(opt);
  CHECK(td.device_status() == 0);
  TaskPersonality t{"9.9.9.9", 0, "d"};
  td.KeepServantAlive(Servant("10.0.0.1:1", 1, "d"), 60s);
  auto g1 = td.WaitForStartingNewTask(t, 60s, td.Now(), false);
  CHECK(g1.ok && g1->servant_location == "10.0.0.1:1");

  std::atomic<int> waiter_done{0};
  WaitResult waiter_result;
  std::thread waiter([&] {
    waiter_result = td.WaitForStartingNewTask(t, 60s, td.Now() + 20s, false);
    waiter_done = 1;
  });
  std::this_thread::sleep_for(100ms);  // parked by now
  CHECK(!waiter_done);
  // A heartbeat adds capacity: wakes nobody (task_dispatcher.cc:190-220) ...
  td.KeepServantAlive(Servant("10.0.0.2:1", 1, "d"), 60s);
  std::this_thread::sleep_for(50ms);
  CHECK(!waiter_done);
  // ... and the next caller takes it.
  auto g2 = td.WaitForStartingNewTask(t, 60s, td.Now(), false);
  CHECK(g2.ok && g2->servant_location == "10.0.0.2:1");
  std::this_thread::sleep_for(50ms);
  CHECK(!waiter_done);
  // FreeTask wakes the waiter (:187), who gets the freed slot.
  auto t0 = std::chrono::steady_clock::now();
  td.FreeTask(g1->task_id);
  waiter.join();
  CHECK(std::chrono::steady_clock::now() - t0 < 5s);
  CHECK(waiter_result.ok && waiter_result->servant_location == "10.0.0.1:1");
}

// Two parked waiters, one FreeTask'd slot each freed back to back + a batch caller: whoever
// drains completes the others' requests; nobody may sleep on past that.
static void CompletedByOthers() {
  GpuTaskDispatcher::Options opt;
  opt.start_expiration_timer = false;
  GpuTaskDispatcher td(opt);
  TaskPersonality t{"9.9.9.9", 0, "d"};
  td.KeepServantAlive(Servant("10.0.0.1:1", 2, "d"), 60s);
  auto a = td.WaitForStartingNewTask(t, 60s, td.Now(), false);
  auto b = td.WaitForStartingNewTask(t, 60s, td.Now(), false);
  CHECK(a.ok && b.ok);
  std::atomic<int> done{0};
  std::vector<std::thread> ws;
  for (int i = 0; i < 2; ++i)
    ws.emplace_back([&] {
      auto r = td.WaitForStartingNewTask(t, 60s, td.Now() + 20s, false);
      CHECK(r.ok);
      ++done;
    });
  std::this_thread::sleep_for(100ms);
  auto t0 = std::chrono::steady_clock::now();
  td.FreeTask(a->task_id);
  td.FreeTask(b->task_id);
  // The batch flavour drains the queue first (earlier arrivals), then places its own (none fit).
  auto rs = td.WaitForStartingNewTasks({t, t}, 60s, {false, false});
  for (auto& w : ws) w.join();
  CHECK(done == 2);
  CHECK(std::chrono::steady_clock::now() - t0 < 5s);
  for (auto& r : rs) CHECK(!r.ok && r.status == WaitStatus::Timeout);
}

// A pool that is always full: eight threads take turns on two slots, every grant freed at once.
// Six of them are parked (asleep on the condition variable, or about to be) at any time, and every
// FreeTask — queued behind a device turn of somebody else's, or applied by the caller itself —
// must wake them (gpu_task_dispatcher.h: free_queue_, sleepers_): with no timer thread around, a
// lost wake-up would sleep until its deadline.
static void FullPoolChurn() {
  GpuTaskDispatcher::Options opt;
  opt.start_expiration_timer = false;
  GpuTaskDispatcher td(opt);
  td.KeepServantAlive(Servant("10.0.0.1:1", 2, "d"), 60s);
  const int kThreads = 8, kRounds = 150;
  std::atomic<int> granted{0}, timeouts{0};
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> ts;
  for (int t = 0; t < kThreads; ++t)
    ts.emplace_back([&, t] {
      TaskPersonality p{"9.9.9." + std::to_string(t), 0, "d"};
      for (int r = 0; r < kRounds; ++r) {
        auto g = td.WaitForStartingNewTask(p, 60s, td.Now() + 10s, false);
        if (!g.ok) {
          ++timeouts;
          continue;
        }
        ++granted;
        if (r & 1) std::this_thread::yield();
        td.FreeTask(g->task_id);
      }
    });
  for (auto& t : ts) t.join();
  CHECK(timeouts == 0);
  CHECK(granted == kThreads * kRounds);
  CHECK(std::chrono::steady_clock::now() - t0 < 30s);
  auto a = td.WaitForStartingNewTask(TaskPersonality{"9.9.9.9", 0, "d"}, 60s, td.Now(), false);
  auto b = td.WaitForStartingNewTask(TaskPersonality{"9.9.9.9", 0, "d"}, 60s, td.Now(), false);
  auto c = td.WaitForStartingNewTask(TaskPersonality{"9.9.9.9", 0, "d"}, 60s, td.Now(), false);
  CHECK(a.ok && b.ok && !c.ok);  // (the books balance: exactly the two slots are free again)
}

static void Storm() {
  GpuTaskDispatcher::Options opt;
  opt.start_expiration_timer = true;  // the dispatcher's own 1 s timer thread runs too
  GpuTaskDispatcher td(opt);
  const int kServants = 24, kThreads = 8;
  for (int i = 0; i < kServants; ++i)
    td.KeepServantAlive(Servant("10.1.0." + std::to_string(i) + ":8335", 4, "d" + std::to_string(i % 3)), 60s);
  std::mutex mu;
  std::set<std::uint64_t> live, ever;
  std::atomic<bool> stop{false};
  std::vector<std::thread> ths;
  for (int k = 0; k < kThreads; ++k)
    ths.emplace_back([&, k] {
      unsigned seed = 1234u + (unsigned)k;
      auto rnd = [&] { return seed = seed * 1664525u + 1013904223u, seed >> 8; };
      while (!stop) {
        const unsigned ev = rnd() % 100;
        TaskPersonality t{"172.16.0." + std::to_string(rnd() % 200), 0, "d" + std::to_string(rnd() % 3)};
        if (ev < 45) {
          auto r = td.WaitForStartingNewTask(t, 30s, td.Now() + std::chrono::milliseconds(rnd() % 3), false);
          if (r.ok) {
            std::scoped_lock _(mu);
            CHECK(ever.insert(r->task_id).second);  // ids are never handed out twice
            live.insert(r->task_id);
          }
        } else if (ev < 55) {
          auto rs = td.WaitForStartingNewTasks({t, t, t}, 30s, {false, true, false});
          std::scoped_lock _(mu);
          for (auto& r : rs)
            if (r.ok) {
              CHECK(ever.insert(r->task_id).second);
              live.insert(r->task_id);
            }
        } else if (ev < 85) {
          std::uint64_t id = ~0ull;
          {
            std::scoped_lock _(mu);
            if (!live.empty()) {
              id = *live.begin();
              live.erase(live.begin());
            }
          }
          if (id != ~0ull) td.FreeTask(id);
        } else if (ev < 90) {
          const int i = (int)(rnd() % kServants);
          td.KeepServantAlive(Servant("10.1.0." + std::to_string(i) + ":8335", 3 + rnd() % 3,
                                      "d" + std::to_string(i % 3)), 60s);
        } else if (ev < 94) {
          (void)td.KeepTaskAlive(rnd() % 1000, 30s);
        } else if (ev < 97) {
          (void)td.NotifyServantRunningTasks("10.1.0." + std::to_string(rnd() % kServants) + ":8335", {});
          (void)td.GetRunningTasks();
          // the copy-free column snapshot, held across other threads' reports
          auto cols = td.GetRunningTasksColumns();
          std::size_t sum = 0;
          for (std::size_t i = 0; i != cols->task_grant_ids.size(); ++i)
            sum += cols->strings[cols->location_off[i]] + cols->digest_len[i];
          CHECK(cols->location_off.size() == cols->task_grant_ids.size() && sum + 1 != 0);
        } else {
          td.OnExpirationTimer();
          (void)td.DumpInternals();
        }
      }
    });
  std::this_thread::sleep_for(1500ms);
  stop = true;
  for (auto& t : ths) t.join();
  // Books balance: every live grant frees exactly one slot; afterwards the whole pool is free.
  for (auto id : live) td.FreeTask(id);
  const std::string dump = td.DumpInternals();
  CHECK(dump.find("\"running_tasks\":0,\"capacity\"") != std::string::npos);
  CHECK(!ever.empty());
}

// Round-4 review: (a) a request whose digest is an empty view WITHOUT storage (data() == nullptr)
// must not match the unused entries of UnsafePlace's per-batch digest cache — it asks for a digest
// nobody advertises: EnvironmentNotFound (reference task_dispatcher.cc:105-108), like a non-null
// empty one; (b) the lease table stays bounded when a page empties while ids are still being
// handed out from it (serial grant / free over many page crossings).
static void NullDigestAndLeasePages() {
  GpuTaskDispatcher::Options opt;
  opt.start_expiration_timer = false;
  GpuTaskDispatcher td(opt);
  CHECK(td.device_status() == 0);
  td.KeepServantAlive(Servant("10.0.0.1:1", 4, "d"), 60s);
  {
    RequestView rq[2];
    rq[0].requestor_ip = "9.9.9.9";
    rq[0].compiler_digest = std::string_view{};  // null data, zero length
    rq[1].requestor_ip = "9.9.9.9";
    rq[1].compiler_digest = std::string_view{"", 0};
    std::int32_t status[2] = {-99, -99};
    std::uint64_t ids[2];
    char locs[2 * 32];
    td.WaitForStartingNewTasksInto(2, rq, 60s, status, ids, locs, 32);
    CHECK(status[0] == 1 && status[1] == 1);  // EnvironmentNotFound, twice
  }
  TaskPersonality t{"9.9.9.9", 0, "d"};
  for (int i = 0; i < 12 * 4096 + 17; ++i) {
    auto g = td.WaitForStartingNewTask(t, 60s, td.Now(), false);
    CHECK(g.ok);
    td.FreeTask(g->task_id);
  }
  CHECK(td.host_stats().lease_pages <= 2);
}

// Round 6: leases are filed under the second in which they run out, and the timer opens only the
// buckets that are due (gpu_task_dispatcher.h: LeaseWheel). Against a plain model — a lease is a
// zombie once a tick has seen expires_at < now, task_dispatcher.cc:523-535 —: random leases, renewals
// that move them forwards and backwards across seconds, frees, ticks at random times, with the
// sweep of stale entries forced to run all the time (lease_sweep_slack = 16).
static void LeaseWheelAgainstModel() {
  std::int64_t now_ns = 5'000'000'000;
  GpuTaskDispatcher::Options opt;
  opt.start_expiration_timer = false;
  opt.lease_sweep_slack = 16;
  opt.clock = [&now_ns] { return GpuTaskDispatcher::Clock::time_point(std::chrono::nanoseconds(now_ns)); };
  GpuTaskDispatcher td(opt);
  for (int i = 0; i < 24; ++i) td.KeepServantAlive(Servant("10.2.0." + std::to_string(i) + ":1", 64, "d"), 100000s);  // 24 x 64 slots
  struct Lease {
    std::uint64_t id;
    std::int64_t expires_ns;
    bool zombie;
  };
  std::vector<Lease> live;
  unsigned seed = 99;
  auto rnd = [&] { return seed = seed * 1664525u + 1013904223u, seed >> 8; };
  TaskPersonality t{"9.9.9.9", 0, "d"};
  for (int step = 0; step < 6000; ++step) {
    const unsigned ev = rnd() % 100;
    if (ev < 35 && live.size() < 1200) {
      const std::int64_t lease = 200'000'000ll + (std::int64_t)(rnd() % 6000) * 1'000'000;
      auto g = td.WaitForStartingNewTask(t, std::chrono::nanoseconds(lease), td.Now(), false);
      CHECK(g.ok);
      live.push_back({g->task_id, now_ns + lease, false});
    } else if (ev < 65 && !live.empty()) {
      Lease& l = live[rnd() % live.size()];
      const std::int64_t lease = 100'000'000ll + (std::int64_t)(rnd() % 5000) * 1'000'000;
      const bool ok = td.KeepTaskAlive(l.id, std::chrono::nanoseconds(lease));
      CHECK(ok == !l.zombie);  // :154-162: a zombie is not renewable
      if (ok) l.expires_ns = now_ns + lease;
    } else if (ev < 80 && !live.empty()) {
      const std::size_t at = rnd() % live.size();
      td.FreeTask(live[at].id);
      live[at] = live.back();
      live.pop_back();
    } else if (ev < 90) {
      now_ns += (std::int64_t)(rnd() % 900) * 1'000'000;
    } else {
      td.OnExpirationTimer();
      for (Lease& l : live)
        if (!l.zombie && l.expires_ns < now_ns) l.zombie = true;
      if (step % 7 == 0) {  // every zombie flag, through the dump
        const std::string dump = td.DumpInternals();
        std::size_t zombies = 0;
        for (std::size_t p = dump.find("\"zombie\":true"); p != std::string::npos; p = dump.find("\"zombie\":true", p + 1)) ++zombies;
        std::size_t want = 0;
        for (const Lease& l : live) want += l.zombie;
        CHECK(zombies == want);
      }
    }
  }
  // (the index holds the live leases and a bounded number of stale entries)
  CHECK(td.host_stats().lease_wheel_entries <= 4 * live.size() + 16 + 1);
}

int main() {
  LeaseWheelAgainstModel();
  NullDigestAndLeasePages();
  WakeOrder();
  CompletedByOthers();
  FullPoolChurn();
  Storm();
  std::printf("TD-CONCURRENCY-OK\n");
  return 0;
}

// Throughput of the preserved TaskDispatcher surface (ydc_td_*, include/yadcc_dispatch.h) measured
// natively — strings in, grant ids and location strings out, one caller thread, no Python
// marshalling in the way. bench.py runs it and puts the numbers into `td_surface` beside the
// verbatim reference's (oracle/_ref through oracle/refbind.py, bounded samples).
//
//   td_native_bench wait <servants> <batch> <reps>
//       WaitForStartingNewTask x batch as one ydc_td_wait_for_starting_new_tasks call, every grant
//       freed again (ydc_td_free_tasks) before the next batch; twice: every request from its own
//       requestor address with the digests interleaved (worst case for the lookups), and in runs
//       of 16 requests that share address and digest (what one WaitForStartingTask RPC asks for,
//       scheduler_service_impl.cc:228-264).
//   td_native_bench heartbeat <servants> <leases> <rounds>
//       `leases` live grants spread over the pool, then rounds x (KeepServantAlive +
//       NotifyServantRunningTasks of every servant, each reporting the grants it holds), then
//       GetRunningTasks polls (task_dispatcher.cc:190-277, running_task_bookkeeper.cc:36-43).
//   td_native_bench latency <servants> <samples>
//       the reference's real call shape (one WaitForStartingTask RPC asks for waiters + 1 grants,
//       daemon/local/task_grant_keeper.cc:145-146; the loop at scheduler_service_impl.cc:234-264):
//       p50 / p99 per CALL for one request through ydc_td_wait_for_starting_new_task and for
//       batches of 2 / 16 / 256 through ydc_td_wait_for_starting_new_tasks, registry warm, the
//       grants freed between two calls (so the release reaches the device inside the next call),
//       a servant's heartbeat with a new load figure before every fourth call.
//   td_native_bench concurrent <servants> <calls per thread>
//       1, 2, 4, 8, 16 and 32 caller threads (RPC handlers), each a loop of one
//       WaitForStartingNewTask + FreeTask from its own requestor address: calls/s of all threads
//       together and p50 / p99 per call. Callers that arrive while a batch is being placed are
//       placed together by whoever gets the lock next (gpu_task_dispatcher.cc: the queue).
//   td_native_bench timer <servants> <leases> <seconds>
//       single-request latency (as in `latency`) with `leases` live leases in the table, twice: the
//       dispatcher's own 1 s expiration timer off, and ON (task_dispatcher.cc:81-82,498-536 — the
//       reference always runs it): p50 / p99 / p99.9 / max per call over `seconds` seconds, and what
//       the ticks cost (lease entries looked at, lock hold time).
//   td_native_bench parked <samples> <seconds>
//       the reference's admitted scaling problem (task_dispatcher.h:281-288): a saturated pool, K =
//       100 / 1000 / 10000 waiters parked in WaitForStartingNewTask, a releaser freeing one slot at
//       a time — wake-to-grant latency and frees per second (tools/parked_workload.h; the reference
//       runs the same workload in oracle/ref_parked_bench.cc).
// Links libydc.so (GPU) or tests/native/libtd_stub.so (CPU model of the device API: host-side
// profiling without a GPU). Prints one JSON object.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "parked_workload.h"
#include "yadcc_dispatch.h"

using Clk = std::chrono::steady_clock;
static double Secs(Clk::time_point a, Clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

static std::vector<std::string> g_digests;
static const char* g_env_ptrs[4];

static std::string Location(int i) {
  return "10." + std::to_string(i >> 16) + "." + std::to_string((i >> 8) & 255) + "." + std::to_string(i & 255) + ":8335";
}

// The pool of SURVEY.md 8(d): core counts 64..256 (scaled by `scale`), 30 % dedicated, 95 % / 40 %
// of the cores offered, a random subset of 4 digests per servant.
static void Register(ydc_td* td, int n, std::mt19937_64& rng, int scale, std::vector<ydc_td_servant>* keep,
                     std::vector<std::string>* locations, std::vector<std::vector<const char*>>* envs) {
  const std::uint32_t nprocs[] = {64, 96, 128, 192, 256};
  keep->resize(n);
  locations->resize(n);
  envs->resize(n);
  for (int i = 0; i < n; ++i) {
    ydc_td_servant& s = (*keep)[i];
    std::memset(&s, 0, sizeof s);
    (*locations)[i] = Location(i);
    s.version = 20;
    s.observed_location = s.reported_location = (*locations)[i].c_str();
    s.num_processors = nprocs[rng() % 5] * scale;
    const bool dedicated = rng() % 10 < 3;
    s.priority = dedicated ? 1 : 2;
    s.max_tasks = s.num_processors * (dedicated ? 95 : 40) / 100;
    s.current_load = rng() % (s.num_processors / 2);
    s.total_memory_in_bytes = 256ull << 30;
    s.memory_available_in_bytes = 64ull << 30;
    for (int d = 0; d < 4; ++d)
      if (rng() % 2) (*envs)[i].push_back(g_env_ptrs[d]);
    if ((*envs)[i].empty()) (*envs)[i].push_back(g_env_ptrs[0]);
    s.env_digests = (*envs)[i].data();
    s.n_envs = (*envs)[i].size();
    if (ydc_td_keep_servant_alive(td, &s, 3600ll * 1000000000ll) != YDC_OK) std::exit(3);
  }
}

static ydc_td* Create(int start_timer = 0) {
  ydc_td* td = nullptr;
  if (ydc_td_create(0, nullptr, start_timer, /*fake_clock=*/0, &td) != YDC_OK || !td) {
    std::fprintf(stderr, "ydc_td_create failed\n");
    std::exit(2);
  }
  if (ydc_td_device_status(td) != YDC_OK) {
    std::fprintf(stderr, "no device (status %d)\n", ydc_td_device_status(td));
    std::exit(2);
  }
  return td;
}

struct WaitNumbers {
  double requests_per_s, ms_per_batch, free_per_s, granted_per_batch, host_ns_per_request, device_ns_per_request;
};

static WaitNumbers TimeWait(ydc_td* td, std::size_t batch, int reps, bool rpc_runs) {
  std::vector<std::string> ips(batch);
  std::vector<const char*> ip_ptrs(batch), digest_ptrs(batch);
  std::vector<std::uint32_t> minv(batch, 20);
  for (std::size_t i = 0; i < batch; ++i) {
    const std::size_t who = rpc_runs ? i / 16 : i;
    ips[i] = "172." + std::to_string(16 + ((who >> 16) & 15)) + "." + std::to_string((who >> 8) & 255) + "." +
             std::to_string(who & 255);
    // (runs share the pointers, as the requests of one RPC share the strings of its message)
    ip_ptrs[i] = rpc_runs && i % 16 ? ip_ptrs[i - 1] : ips[i].c_str();
    digest_ptrs[i] = g_env_ptrs[who % 4];
  }
  std::vector<std::int32_t> status(batch);
  std::vector<std::uint64_t> ids(batch), granted_ids;
  constexpr std::size_t kStride = 32;
  std::vector<char> locs(batch * kStride);
  granted_ids.reserve(batch);
  double wait_s = 0, free_s = 0;
  std::size_t granted = 0;
  ydc_td_stats s0{}, s1{};
  for (int rep = -2; rep < reps; ++rep) {  // two warm-up rounds
    if (rep == 0) ydc_td_host_stats(td, &s0);
    auto t0 = Clk::now();
    int rc = ydc_td_wait_for_starting_new_tasks(td, batch, ip_ptrs.data(), minv.data(), digest_ptrs.data(),
                                                15ll * 1000000000ll, nullptr, status.data(), ids.data(),
                                                locs.data(), kStride);
    auto t1 = Clk::now();
    if (rc < 0) {
      std::fprintf(stderr, "device error %d\n", rc);
      std::exit(1);
    }
    granted_ids.clear();
    for (std::size_t i = 0; i < batch; ++i)
      if (status[i] == YDC_TD_GRANTED) granted_ids.push_back(ids[i]);
    auto t2 = Clk::now();
    ydc_td_free_tasks(td, granted_ids.data(), granted_ids.size());
    auto t3 = Clk::now();
    if (rep >= 0) {
      wait_s += Secs(t0, t1);
      free_s += Secs(t2, t3);
      granted += granted_ids.size();
    }
  }
  ydc_td_host_stats(td, &s1);
  const double reqs = (double)(s1.requests - s0.requests);
  return {batch * reps / wait_s, 1e3 * wait_s / reps, granted / free_s, (double)granted / reps,
          reqs ? (s1.host_ns - s0.host_ns) / reqs : 0, reqs ? (s1.device_ns - s0.device_ns) / reqs : 0};
}

static void PrintWait(const char* name, const WaitNumbers& w, bool last) {
  std::printf("\"%s\": {\"requests_per_s\": %.0f, \"ms_per_batch\": %.4f, \"granted_per_batch\": %.1f, "
              "\"free_tasks_per_s\": %.0f, \"host_ns_per_request\": %.1f, \"device_ns_per_request\": %.1f}%s",
              name, w.requests_per_s, w.ms_per_batch, w.granted_per_batch, w.free_per_s, w.host_ns_per_request,
              w.device_ns_per_request, last ? "" : ", ");
}

static int WaitMode(int n_servants, std::size_t batch, int reps) {
  ydc_td* td = Create();
  std::mt19937_64 rng(1);
  std::vector<ydc_td_servant> sv;
  std::vector<std::string> locations;
  std::vector<std::vector<const char*>> envs;
  // capacity ~ 1.5 x the batch so that (almost) every request is granted
  const int scale = std::max<int>(1, (int)(batch * 3 / 2 / ((std::size_t)n_servants * 70) + 1));
  Register(td, n_servants, rng, scale, &sv, &locations, &envs);
  const WaitNumbers distinct = TimeWait(td, batch, reps, false);
  const WaitNumbers runs = TimeWait(td, batch, reps, true);
  std::printf("{\"mode\": \"wait\", \"servants\": %d, \"batch\": %zu, \"reps\": %d, ", n_servants, batch, reps);
  PrintWait("distinct_requestors", distinct, false);
  PrintWait("rpc_runs_of_16", runs, true);
  std::printf("}\n");
  ydc_td_destroy(td);
  return 0;
}

static int HeartbeatMode(int n_servants, std::size_t leases, int rounds) {
  ydc_td* td = Create();
  std::mt19937_64 rng(2);
  std::vector<ydc_td_servant> sv;
  std::vector<std::string> locations;
  std::vector<std::vector<const char*>> envs;
  const int scale = std::max<int>(1, (int)(leases * 2 / ((std::size_t)n_servants * 70) + 1));
  Register(td, n_servants, rng, scale, &sv, &locations, &envs);
  // Grants: batches of 100k requests until `leases` are live; the location string of every grant
  // says which servant holds it ("10.a.b.c:8335" -> index).
  std::vector<std::vector<ydc_td_running_task>> held(n_servants);
  std::string task_digest(64, 'a');
  {
    const std::size_t batch = std::min<std::size_t>(leases, 100000);
    std::vector<std::string> ips(batch);
    std::vector<const char*> ip_ptrs(batch), digest_ptrs(batch);
    std::vector<std::uint32_t> minv(batch, 20);
    for (std::size_t i = 0; i < batch; ++i) {
      ips[i] = "172.16." + std::to_string((i >> 8) & 255) + "." + std::to_string(i & 255);
      ip_ptrs[i] = ips[i].c_str();
      digest_ptrs[i] = g_env_ptrs[i % 4];
    }
    std::vector<std::int32_t> status(batch);
    std::vector<std::uint64_t> ids(batch);
    constexpr std::size_t kStride = 32;
    std::vector<char> locs(batch * kStride);
    std::size_t live = 0;
    while (live < leases) {
      const std::size_t n = std::min(batch, leases - live);
      int rc = ydc_td_wait_for_starting_new_tasks(td, n, ip_ptrs.data(), minv.data(), digest_ptrs.data(),
                                                  3600ll * 1000000000ll, nullptr, status.data(), ids.data(),
                                                  locs.data(), kStride);
      if (rc < 0) {
        std::fprintf(stderr, "device error %d\n", rc);
        return 1;
      }
      std::size_t got = 0;
      for (std::size_t i = 0; i < n; ++i) {
        if (status[i] != YDC_TD_GRANTED) continue;
        unsigned a, b, c, d;
        if (std::sscanf(locs.data() + i * kStride, "%u.%u.%u.%u", &a, &b, &c, &d) != 4) return 1;
        const int s = (int)((b << 16) | (c << 8) | d);
        ydc_td_running_task t{};
        t.servant_task_id = held[s].size();
        t.task_grant_id = ids[i];
        t.servant_location = locations[s].c_str();
        t.task_digest = task_digest.c_str();
        held[s].push_back(t);
        ++got;
      }
      if (!got) {
        std::fprintf(stderr, "pool exhausted at %zu leases\n", live);
        return 1;
      }
      live += got;
    }
  }
  std::vector<std::uint64_t> unknown(1024);
  // Round 0 fills the bookkeeper (every list is new); the following rounds report the same lists
  // with a new load figure, which is what steady-state heartbeats look like.
  double first_s = 0, steady_s = 0;
  std::size_t unknown_total = 0;
  for (int r = 0; r < rounds + 1; ++r) {
    auto t0 = Clk::now();
    for (int s = 0; s < n_servants; ++s) {
      sv[s].current_load = (sv[s].current_load + 1) % (sv[s].num_processors / 2);
      ydc_td_keep_servant_alive(td, &sv[s], 3600ll * 1000000000ll);
      const std::int64_t u = ydc_td_notify_servant_running_tasks(td, locations[s].c_str(), held[s].data(),
                                                               held[s].size(), unknown.data(), unknown.size());
      unknown_total += (std::size_t)std::max<std::int64_t>(u, 0);
    }
    const double dt = Secs(t0, Clk::now());
    if (r == 0) first_s = dt; else steady_s += dt;
  }
  // GetRunningTasks: ids only, and with the location strings.
  std::vector<std::uint64_t> st(leases), gr(leases);
  std::vector<char> locs(leases * 32);
  const int polls = 20;
  auto p0 = Clk::now();
  std::int64_t total = 0;
  for (int i = 0; i < polls; ++i) total = ydc_td_get_running_tasks(td, st.data(), gr.data(), nullptr, 0, nullptr, 0, leases);
  auto p1 = Clk::now();
  for (int i = 0; i < polls; ++i) total = ydc_td_get_running_tasks(td, st.data(), gr.data(), locs.data(), 32, nullptr, 0, leases);
  auto p2 = Clk::now();
  // ... and without the copy: a view into the shared snapshot (ydc_td_running_tasks_acquire / _release).
  const int views = 200000;
  std::uint64_t seen = 0;
  for (int i = 0; i < views; ++i) {
    void* h = nullptr;
    ydc_td_running_view v{};
    if (ydc_td_running_tasks_acquire(td, &h, &v) != YDC_OK) return 1;
    seen += v.n ? v.task_grant_ids[(std::size_t)i % v.n] != 0 : 0;
    ydc_td_running_tasks_release(h);
  }
  auto p3 = Clk::now();
  ydc_td_stats hs{};
  ydc_td_host_stats(td, &hs);
  std::printf("{\"mode\": \"heartbeat\", \"servants\": %d, \"leases\": %zu, \"rounds\": %d, "
              "\"heartbeats_per_s\": %.0f, \"us_per_heartbeat\": %.3f, \"first_round_heartbeats_per_s\": %.0f, "
              "\"reported_tasks_per_heartbeat\": %.1f, \"unknown_reported\": %zu, "
              "\"get_running_tasks_per_s\": %.1f, \"get_running_tasks_with_locations_per_s\": %.1f, "
              "\"running_tasks_views_per_s\": %.0f, "
              "\"running_tasks_listed\": %lld, \"bookkeeper_rebuilds\": %llu}\n",
              n_servants, leases, rounds, (double)n_servants * rounds / steady_s,
              1e6 * steady_s / ((double)n_servants * rounds), n_servants / first_s,
              (double)leases / n_servants, unknown_total, polls / Secs(p0, p1), polls / Secs(p1, p2),
              views / Secs(p2, p3) + 0.0 * seen, (long long)total, (unsigned long long)hs.bookkeeper_rebuilds);
  ydc_td_destroy(td);
  return 0;
}

static void Percentiles(std::vector<double>& us, double* p50, double* p99, double* mean) {
  std::sort(us.begin(), us.end());
  double sum = 0;
  for (double v : us) sum += v;
  *p50 = us[us.size() / 2];
  *p99 = us[std::min(us.size() - 1, us.size() * 99 / 100)];
  *mean = sum / us.size();
}

static int LatencyMode(int n_servants, int samples) {
  ydc_td* td = Create();
  std::mt19937_64 rng(3);
  std::vector<ydc_td_servant> sv;
  std::vector<std::string> locations;
  std::vector<std::vector<const char*>> envs;
  Register(td, n_servants, rng, 1, &sv, &locations, &envs);
  std::printf("{\"mode\": \"latency\", \"servants\": %d, \"samples\": %d, \"per_call_us\": {", n_servants, samples);
  const std::size_t batches[] = {1, 1, 2, 4, 8, 16, 32, 64, 128, 256};  // (the first 1: single-request entry point)
  bool first = true;
  for (int bi = 0; bi < 10; ++bi) {
    const std::size_t batch = batches[bi];
    const bool single_entry = bi == 0;
    std::vector<const char*> ip_ptrs(batch), digest_ptrs(batch);
    std::vector<std::uint32_t> minv(batch, 20);
    std::vector<std::int32_t> status(batch);
    std::vector<std::uint64_t> ids(batch), granted_ids;
    constexpr std::size_t kStride = 32;
    std::vector<char> locs(batch * kStride);
    std::vector<double> us, us_hb;
    std::size_t granted = 0;
    for (int r = -20; r < samples; ++r) {
      // One RPC: one requestor, one digest (scheduler_service_impl.cc:228-264).
      const std::string ip = "172.16." + std::to_string((r >> 8) & 255) + "." + std::to_string(r & 255);
      for (std::size_t i = 0; i < batch; ++i) {
        ip_ptrs[i] = ip.c_str();
        digest_ptrs[i] = g_env_ptrs[(unsigned)r % 4];
      }
      const bool heartbeat = (r & 3) == 3;
      if (heartbeat) {
        const int s = (int)(rng() % n_servants);
        sv[s].current_load = (sv[s].current_load + 1) % (sv[s].num_processors / 2);
        ydc_td_keep_servant_alive(td, &sv[s], 3600ll * 1000000000ll);
      }
      auto t0 = Clk::now();
      int rc;
      if (single_entry) {
        rc = ydc_td_wait_for_starting_new_task(td, ip_ptrs[0], 20, digest_ptrs[0], 15ll * 1000000000ll, 0, 0,
                                               &ids[0], locs.data(), kStride);
        status[0] = rc;
      } else {
        rc = ydc_td_wait_for_starting_new_tasks(td, batch, ip_ptrs.data(), minv.data(), digest_ptrs.data(),
                                                15ll * 1000000000ll, nullptr, status.data(), ids.data(),
                                                locs.data(), kStride);
      }
      auto t1 = Clk::now();
      if (rc < 0) {
        std::fprintf(stderr, "device error %d\n", rc);
        return 1;
      }
      granted_ids.clear();
      for (std::size_t i = 0; i < batch; ++i)
        if (status[i] == YDC_TD_GRANTED) granted_ids.push_back(ids[i]);
      ydc_td_free_tasks(td, granted_ids.data(), granted_ids.size());
      if (r >= 0) {
        (heartbeat ? us_hb : us).push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        granted += granted_ids.size();
      }
    }
    double p50, p99, mean, h50 = 0, h99 = 0, hmean = 0;
    Percentiles(us, &p50, &p99, &mean);
    if (!us_hb.empty()) Percentiles(us_hb, &h50, &h99, &hmean);
    std::printf("%s\"%s%zu\": {\"p50\": %.2f, \"p99\": %.2f, \"mean\": %.2f, \"behind_a_heartbeat_p50\": %.2f, "
                "\"behind_a_heartbeat_p99\": %.2f, \"granted_per_call\": %.2f}",
                first ? "" : ", ", single_entry ? "single_" : "batch_", batch, p50, p99, mean, h50, h99,
                (double)granted / samples);
    first = false;
  }
  std::printf("}}\n");
  ydc_td_destroy(td);
  return 0;
}

static int ConcurrentMode(int n_servants, int calls) {
  ydc_td* td = Create();
  std::mt19937_64 rng(3);
  std::vector<ydc_td_servant> sv;
  std::vector<std::string> locations;
  std::vector<std::vector<const char*>> envs;
  Register(td, n_servants, rng, 1, &sv, &locations, &envs);
  std::printf("{\"mode\": \"concurrent\", \"servants\": %d, \"calls_per_thread\": %d, \"threads\": {", n_servants, calls);
  bool first = true;
  for (int T : {1, 2, 4, 8, 16, 32}) {
    ydc_td_stats st0{}, st1{};
    ydc_td_host_stats(td, &st0);
    std::vector<std::vector<double>> lat(T);
    std::vector<double> free_us(T, 0.0);
    std::atomic<int> ready{0}, failed{0};
    std::atomic<bool> go{false};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        const std::string ip = "172.20." + std::to_string(t) + ".7";
        char loc[32];
        std::uint64_t id = 0;
        lat[t].reserve(calls);
        for (int r = -50; r < calls; ++r) {
          if (r == 0) {  // (the timed part starts when every thread has made its first calls)
            ++ready;
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
          }
          auto t0 = Clk::now();
          const int rc = ydc_td_wait_for_starting_new_task(td, ip.c_str(), 20, g_env_ptrs[(unsigned)(r + t) % 4],
                                                           15ll * 1000000000ll, 0, 0, &id, loc, sizeof loc);
          auto t1 = Clk::now();
          if (rc < 0) {
            ++failed;
            return;
          }
          if (rc == YDC_TD_GRANTED) ydc_td_free_task(td, id);
          auto t2 = Clk::now();
          if (r >= 0) {
            lat[t].push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            free_us[t] += std::chrono::duration<double, std::micro>(t2 - t1).count();
          }
        }
      });
    while (ready.load() < T) std::this_thread::yield();
    auto w0 = Clk::now();
    go.store(true, std::memory_order_release);
    for (auto& x : th) x.join();
    auto w1 = Clk::now();
    if (failed.load()) {
      std::fprintf(stderr, "device error in a caller thread\n");
      return 1;
    }
    std::vector<double> all;
    for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
    double p50, p99, mean;
    Percentiles(all, &p50, &p99, &mean);
    double free_sum = 0;
    for (double v : free_us) free_sum += v;
    ydc_td_host_stats(td, &st1);
    const double per_batch = (double)(st1.requests - st0.requests) / std::max<std::uint64_t>(1, st1.batches - st0.batches);
    const double dev_us = (double)(st1.device_ns - st0.device_ns) / 1e3 / std::max<std::uint64_t>(1, st1.batches - st0.batches);
    const double host_us = (double)(st1.host_ns - st0.host_ns) / 1e3 / std::max<std::uint64_t>(1, st1.batches - st0.batches);
    const double turns_per_s = (double)(st1.batches - st0.batches) / Secs(w0, w1);
    std::printf("%s\"%d\": {\"calls_per_s\": %.0f, \"p50_us\": %.2f, \"p99_us\": %.2f, \"mean_us\": %.2f, "
                "\"free_task_mean_us\": %.2f, \"requests_per_device_turn\": %.2f, \"device_us_per_turn\": %.2f, "
                "\"host_us_per_turn\": %.2f, \"us_between_turn_starts\": %.2f}",
                first ? "" : ", ", T, (double)T * calls / Secs(w0, w1), p50, p99, mean, free_sum / all.size(),
                per_batch, dev_us, host_us, 1e6 / turns_per_s);
    first = false;
  }
  std::printf("}}\n");
  ydc_td_destroy(td);
  return 0;
}

// `leases` live grants (one-hour leases) spread over the pool, in batches of 100k.
static bool Prefill(ydc_td* td, std::size_t leases) {
  const std::size_t batch = std::min<std::size_t>(leases, 100000);
  if (!batch) return true;
  std::vector<std::string> ips(batch);
  std::vector<const char*> ip_ptrs(batch), digest_ptrs(batch);
  std::vector<std::uint32_t> minv(batch, 20);
  for (std::size_t i = 0; i < batch; ++i) {
    ips[i] = "172.17." + std::to_string((i >> 8) & 255) + "." + std::to_string(i & 255);
    ip_ptrs[i] = ips[i].c_str();
    digest_ptrs[i] = g_env_ptrs[i % 4];
  }
  std::vector<std::int32_t> status(batch);
  std::vector<std::uint64_t> ids(batch);
  for (std::size_t live = 0; live < leases;) {
    const std::size_t n = std::min(batch, leases - live);
    if (ydc_td_wait_for_starting_new_tasks(td, n, ip_ptrs.data(), minv.data(), digest_ptrs.data(), 3600ll * 1000000000ll,
                                           nullptr, status.data(), ids.data(), nullptr, 0) < 0)
      return false;
    std::size_t got = 0;
    for (std::size_t i = 0; i < n; ++i) got += status[i] == YDC_TD_GRANTED;
    if (!got) return false;
    live += got;
  }
  return true;
}

static int TimerMode(int n_servants, std::size_t leases, double seconds) {
  std::printf("{\"mode\": \"timer\", \"servants\": %d, \"live_leases\": %zu, \"seconds_per_leg\": %.1f", n_servants, leases, seconds);
  for (int timer_on = 0; timer_on < 2; ++timer_on) {
    ydc_td* td = Create(timer_on);
    std::mt19937_64 rng(5);
    std::vector<ydc_td_servant> sv;
    std::vector<std::string> locations;
    std::vector<std::vector<const char*>> envs;
    const int scale = std::max<int>(1, (int)(leases * 2 / ((std::size_t)n_servants * 70) + 1));
    Register(td, n_servants, rng, scale, &sv, &locations, &envs);
    if (!Prefill(td, leases)) {
      std::fprintf(stderr, "prefill failed\n");
      return 1;
    }
    std::vector<double> us;
    us.reserve(1 << 20);
    char loc[32];
    std::uint64_t id = 0;
    const auto t_end = Clk::now() + std::chrono::duration<double>(seconds);
    for (long r = -200; Clk::now() < t_end; ++r) {
      const std::string ip = "172.16." + std::to_string((r >> 8) & 255) + "." + std::to_string(r & 255);
      if ((r & 3) == 3) {
        const int s = (int)(rng() % n_servants);
        sv[s].current_load = (sv[s].current_load + 1) % (sv[s].num_processors / 2);
        ydc_td_keep_servant_alive(td, &sv[s], 3600ll * 1000000000ll);
      }
      auto t0 = Clk::now();
      const int rc = ydc_td_wait_for_starting_new_task(td, ip.c_str(), 20, g_env_ptrs[(unsigned)r % 4], 15ll * 1000000000ll, 0,
                                                       0, &id, loc, sizeof loc);
      auto t1 = Clk::now();
      if (rc < 0) return 1;
      if (rc == YDC_TD_GRANTED) ydc_td_free_task(td, id);
      if (r >= 0) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    ydc_td_stats hs{};
    ydc_td_host_stats(td, &hs);
    std::sort(us.begin(), us.end());
    double sum = 0;
    for (double v : us) sum += v;
    auto q = [&](double f) { return us[std::min(us.size() - 1, (std::size_t)(f * us.size()))]; };
    std::size_t above50 = us.end() - std::upper_bound(us.begin(), us.end(), 50.0);
    std::printf(", \"timer_%s\": {\"calls\": %zu, \"p50\": %.2f, \"p99\": %.2f, \"p999\": %.2f, \"max\": %.2f, \"mean\": %.2f, "
                "\"calls_above_50us\": %zu, \"timer_ticks\": %llu, \"timer_max_us\": %.1f, \"timer_last_us\": %.1f, "
                "\"lease_entries_seen_last_tick\": %llu, \"lease_index_entries\": %llu}",
                timer_on ? "on" : "off", us.size(), q(0.5), q(0.99), q(0.999), us.back(), sum / us.size(), above50,
                (unsigned long long)hs.timer_ticks, hs.timer_max_ns / 1e3, hs.timer_last_ns / 1e3,
                (unsigned long long)hs.timer_lease_entries_seen, (unsigned long long)hs.lease_wheel_entries);
    ydc_td_destroy(td);
  }
  std::printf("}\n");
  return 0;
}

struct TdParkedAdapter {
  ydc_td* td;
  std::vector<std::string> ips;
  bool Wait(int waiter, long long timeout_ms, unsigned long long* id) {
    std::uint64_t got = 0;
    char loc[32];
    const int rc = ydc_td_wait_for_starting_new_task(td, ips[waiter].c_str(), 20, g_env_ptrs[0], 3600ll * 1000000000ll,
                                                     timeout_ms * 1000000ll, 0, &got, loc, sizeof loc);
    if (rc < 0) {
      std::fprintf(stderr, "device error %d\n", rc);
      std::exit(1);
    }
    *id = got;
    return rc == YDC_TD_GRANTED;
  }
  void Free(unsigned long long id) { ydc_td_free_task(td, id); }
};

static int ParkedMode(int samples, double seconds) {
  std::printf("{\"mode\": \"parked\", \"pool\": \"64 servants x 4 slots, all taken\", \"waiters\": {");
  const int ks[] = {100, 1000, 10000};
  for (int ki = 0; ki < 3; ++ki) {
    ydc_td* td = Create(/*start_timer=*/1);
    std::vector<std::string> locations(64);
    const char* env[1] = {g_env_ptrs[0]};
    for (int i = 0; i < 64; ++i) {
      ydc_td_servant s;
      std::memset(&s, 0, sizeof s);
      locations[i] = Location(i);
      s.version = 20;
      s.observed_location = s.reported_location = locations[i].c_str();
      s.num_processors = 64;
      s.priority = 2;
      s.max_tasks = 4;
      s.total_memory_in_bytes = 256ull << 30;
      s.memory_available_in_bytes = 64ull << 30;
      s.env_digests = env;
      s.n_envs = 1;
      if (ydc_td_keep_servant_alive(td, &s, 3600ll * 1000000000ll) != YDC_OK) return 3;
    }
    TdParkedAdapter a{td, {}};
    for (int k = 0; k < ks[ki]; ++k) a.ips.push_back("172.21." + std::to_string(k >> 8) + "." + std::to_string(k & 255));
    a.ips.push_back("172.22.0.1");
    std::vector<unsigned long long> initial;
    for (;;) {
      unsigned long long id;
      if (!a.Wait(ks[ki], 0, &id)) break;
      initial.push_back(id);
    }
    if (initial.size() != 256) {
      std::fprintf(stderr, "pool holds %zu grants, expected 256\n", initial.size());
      return 1;
    }
    parked::Run<TdParkedAdapter> run;
    run.a = &a;
    const parked::Result r = run.Go(ks[ki], initial, samples, seconds);
    parked::Print("ydc_td", r, ki == 2);
    ydc_td_destroy(td);
  }
  std::printf("}}\n");
  return 0;
}

int main(int argc, char** argv) {
  for (int i = 0; i < 4; ++i) {
    char b[80];
    std::snprintf(b, sizeof b, "%064x", 0xc0ffee + i);
    g_digests.push_back(b);
  }
  for (int i = 0; i < 4; ++i) g_env_ptrs[i] = g_digests[i].c_str();
  const std::string mode = argc > 1 ? argv[1] : "wait";
  if (mode == "wait")
    return WaitMode(argc > 2 ? std::atoi(argv[2]) : 2000, argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 10000,
                    argc > 4 ? std::atoi(argv[4]) : 20);
  if (mode == "heartbeat")
    return HeartbeatMode(argc > 2 ? std::atoi(argv[2]) : 16000, argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 1000000,
                         argc > 4 ? std::atoi(argv[4]) : 3);
  if (mode == "latency")
    return LatencyMode(argc > 2 ? std::atoi(argv[2]) : 2000, argc > 3 ? std::atoi(argv[3]) : 1000);
  if (mode == "concurrent")
    return ConcurrentMode(argc > 2 ? std::atoi(argv[2]) : 2000, argc > 3 ? std::atoi(argv[3]) : 2000);
  if (mode == "timer")
    return TimerMode(argc > 2 ? std::atoi(argv[2]) : 16000, argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 1000000,
                     argc > 4 ? std::atof(argv[4]) : 4.0);
  if (mode == "parked") return ParkedMode(argc > 2 ? std::atoi(argv[2]) : 200, argc > 3 ? std::atof(argv[3]) : 2.0);
  std::fprintf(stderr, "usage: td_native_bench wait|heartbeat|latency|concurrent|timer|parked ...\n");
  return 2;
}

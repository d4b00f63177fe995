// Oracle (test infrastructure, NOT product code): the CPU baseline of the parked-waiters
// measurement. The reference's own TaskDispatcher (task_dispatcher.cc compiled where it lies,
// against oracle/shims with -DORACLE_SHIM_REAL_THREADS: real clock, real condition variable, OS
// threads for fibers) under the workload of tools/parked_workload.h — K waiters parked in
// WaitForStartingNewTask on a saturated pool, a releaser freeing one slot at a time. This is the
// path the reference itself says "doesn't scale well" (task_dispatcher.h:281-288: notify_all per
// FreeTask, .cc:185-187; every waiter re-scans under the one lock, .cc:101-119).
// Executed by bench.py's cpu_baseline leg only; prints one JSON object.
//   ref_parked_bench <samples> <seconds> [K ...]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "flare/fiber/timer.h"
#include "parked_workload.h"
#include "yadcc/scheduler/task_dispatcher.h"  // the reference's own header

using namespace std::literals;
using yadcc::scheduler::ServantPersonality;
using yadcc::scheduler::TaskDispatcher;
using yadcc::scheduler::TaskPersonality;

struct RefAdapter {
  TaskDispatcher* td;
  std::vector<std::string> ips;
  std::string digest;
  bool Wait(int waiter, long long timeout_ms, unsigned long long* id) {
    TaskPersonality t;
    t.requestor_ip = ips[waiter];
    t.min_version = 20;
    t.env_desc.set_compiler_digest(digest);
    auto r = td->WaitForStartingNewTask(t, 3600s, flare::ReadCoarseSteadyClock() + std::chrono::milliseconds(timeout_ms),
                                        false);
    if (!r) return false;
    *id = r->task_id;
    return true;
  }
  void Free(unsigned long long id) { td->FreeTask(id); }
};

int main(int argc, char** argv) {
  const int samples = argc > 1 ? std::atoi(argv[1]) : 100;
  const double seconds = argc > 2 ? std::atof(argv[2]) : 2.0;
  std::vector<int> ks;
  for (int i = 3; i < argc; ++i) ks.push_back(std::atoi(argv[i]));
  if (ks.empty()) ks = {100, 1000, 10000};
  std::printf("{\"mode\": \"parked\", \"pool\": \"64 servants x 4 slots, all taken\", \"kind\": \"reference\", \"waiters\": {");
  for (std::size_t ki = 0; ki < ks.size(); ++ki) {
    TaskDispatcher td;
    // The reference's 1 s timer (task_dispatcher.cc:81-82): SetTimer is captured by the shim; this
    // thread is the timer.
    std::atomic<bool> stop_timer{false};
    std::thread timer([&] {
      while (!stop_timer.load()) {
        std::this_thread::sleep_for(1s);
        auto copy = flare::shim::Timers();
        for (auto&& [id, cb] : copy) cb();
      }
    });
    RefAdapter a{&td, {}, std::string(64, 'c')};
    for (int i = 0; i < 64; ++i) {
      ServantPersonality s{};
      s.version = 20;
      s.observed_location = s.reported_location =
          "10.0." + std::to_string(i >> 8) + "." + std::to_string(i & 255) + ":8335";
      s.environments.emplace_back().set_compiler_digest(a.digest);
      s.num_processors = 64;
      s.current_load = 0;
      s.total_memory_in_bytes = 256ull << 30;
      s.memory_available_in_bytes = 64ull << 30;
      s.priority = yadcc::scheduler::SERVANT_PRIORITY_USER;
      s.max_tasks = 4;
      td.KeepServantAlive(s, 3600s);
    }
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
    parked::Run<RefAdapter> run;
    run.a = &a;
    const parked::Result r = run.Go(ks[ki], initial, samples, seconds);
    parked::Print("reference", r, ki + 1 == ks.size());
    std::fflush(stdout);
    stop_timer = true;
    timer.join();
  }
  std::printf("}}\n");
  return 0;
}

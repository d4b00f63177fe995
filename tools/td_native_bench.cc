// Developer tool: throughput of the host class path in C++ (no Python marshalling):
// GpuTaskDispatcher::WaitForStartingNewTasks on a registry of 2000 servants with 4 compiler
// digests, batches of `batch` requests from distinct requestor hosts, every grant freed again
// before the next batch. Prints requests/s per stage. Needs the GPU.
//   tools/td_native_bench [batch=10000] [reps=20]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "gpu_task_dispatcher.h"

using namespace ydc;
using namespace std::literals;
using Clk = std::chrono::steady_clock;

int main(int argc, char** argv) {
  const std::size_t batch = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 10000;
  const int reps = argc > 2 ? std::atoi(argv[2]) : 20;
  GpuTaskDispatcher::Options opt;
  opt.device = 0;
  opt.start_expiration_timer = false;
  GpuTaskDispatcher td(opt);
  if (td.device_status() != 0) {
    std::fprintf(stderr, "no device: %s\n", td.device_error_message().c_str());
    return 2;
  }
  std::mt19937_64 rng(1);
  std::vector<std::string> digests;
  for (int i = 0; i < 4; ++i) {
    char b[80];
    std::snprintf(b, sizeof b, "%064x", 0xc0ffee + i);
    digests.push_back(b);
  }
  const std::uint32_t nprocs[] = {64, 96, 128, 192, 256};
  for (int i = 0; i < 2000; ++i) {
    ServantPersonality s;
    s.version = 20;
    s.observed_location = s.reported_location =
        "10." + std::to_string(i >> 16) + "." + std::to_string((i >> 8) & 255) + "." + std::to_string(i & 255) + ":8335";
    s.num_processors = nprocs[rng() % 5];
    const bool dedicated = rng() % 10 < 3;
    s.priority = dedicated ? kServantPriorityDedicated : kServantPriorityUser;
    s.max_tasks = s.num_processors * (dedicated ? 95 : 40) / 100;
    s.current_load = rng() % s.num_processors;
    s.total_memory_in_bytes = 256ull << 30;
    s.memory_available_in_bytes = 64ull << 30;
    for (auto&& d : digests)
      if (rng() % 2) s.environments.push_back(d);
    if (s.environments.empty()) s.environments.push_back(digests[0]);
    td.KeepServantAlive(s, 30s);
  }
  std::vector<TaskPersonality> reqs(batch);
  for (std::size_t i = 0; i < batch; ++i) {
    reqs[i].requestor_ip = "172.16." + std::to_string((i >> 8) & 255) + "." + std::to_string(i & 255);
    reqs[i].min_version = 20;
    reqs[i].compiler_digest = digests[i % 4];
  }
  const std::vector<bool> prefetching(batch, false);
  double wait_s = 0, free_s = 0;
  std::size_t granted = 0;
  for (int rep = -2; rep < reps; ++rep) {  // two warm-up rounds
    auto t0 = Clk::now();
    auto rs = td.WaitForStartingNewTasks(reqs, 15s, prefetching);
    auto t1 = Clk::now();
    std::size_t g = 0;
    for (auto&& r : rs) {
      if (r.device_error) {
        std::fprintf(stderr, "device error %d\n", r.device_error);
        return 1;
      }
      if (r) {
        td.FreeTask(r->task_id);
        ++g;
      }
    }
    auto t2 = Clk::now();
    if (rep >= 0) {
      wait_s += std::chrono::duration<double>(t1 - t0).count();
      free_s += std::chrono::duration<double>(t2 - t1).count();
      granted += g;
    }
  }
  std::printf("{\"batch\": %zu, \"reps\": %d, \"granted_per_batch\": %.1f, "
              "\"wait_requests_per_s\": %.0f, \"wait_ms_per_batch\": %.3f, \"free_tasks_per_s\": %.0f}\n",
              batch, reps, (double)granted / reps, batch * reps / wait_s, 1e3 * wait_s / reps,
              granted / free_s);
  return 0;
}

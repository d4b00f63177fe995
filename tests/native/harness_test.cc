// Native (C++) drive of the MI355X dispatcher through SchedulerHarness — the call pattern of
// the reference's SchedulerServiceImpl (scheduler_service_impl.cc:67-318) — without any
// Python in the loop. Scenarios: the reference's LoadBalanceCase golden vector
// (task_dispatcher_test.cc:216-298) issued as RPC-shaped calls, a multi-grant request, the
// argument / NAT / environment / quota status mapping, and the running-task report cycle.
// Prints HARNESS-OK and exits 0 when everything holds. Needs the GPU (no CPU placement).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "gpu_task_dispatcher.h"
#include "scheduler_harness.h"

using namespace ydc;
using namespace std::literals;

#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                         \
    }                                                                       \
  } while (0)

static GpuTaskDispatcher::Clock::time_point g_now{};

static HeartbeatRequest Beat(const std::string& location, std::uint32_t capacity, std::uint32_t nproc,
                             std::uint32_t load, const std::string& digest,
                             int priority = kServantPriorityUser) {
  HeartbeatRequest r;
  r.next_heartbeat_in_ms = 10000;
  r.version = 20;
  r.location = location;
  r.num_processors = nproc;
  r.current_load = load;
  r.servant_priority = priority;
  r.capacity = capacity;
  r.memory_available_in_bytes = 50ull << 30;
  r.env_descs = {digest};
  return r;
}

static std::string IpOf(const std::string& location) { return location.substr(0, location.find(':')); }

int main() {
  GpuTaskDispatcher::Options opt;
  opt.device = 0;
  opt.start_expiration_timer = false;
  opt.clock = [] { return g_now; };
  GpuTaskDispatcher dispatcher(opt);
  if (dispatcher.device_status() != 0) {
    std::fprintf(stderr, "no device: %s\n", dispatcher.device_error_message().c_str());
    return 2;
  }
  SchedulerHarness scheduler(&dispatcher);
  HeartbeatResponse hb;

  // ---- LoadBalanceCase (task_dispatcher_test.cc:216-298) as RPCs ----
  struct S { std::string loc; std::uint32_t cap, nproc, load; };
  std::vector<S> pool = {{"192.168.0.0:1000", 7, 16, 16}, {"192.168.0.1:1111", 7, 16, 1},
                         {"192.168.0.2:2222", 8, 16, 5}, {"192.168.0.3:3333", 6, 16, 12}};
  CHECK(scheduler.Heartbeat(IpOf(pool[0].loc), Beat(pool[0].loc, 7, 16, 16, "Load Balance"), &hb) ==
        kStatusSuccess);
  WaitForStartingTaskRequest one;
  one.compiler_digest = "Load Balance";
  one.immediate_reqs = 1;
  one.next_keep_alive_in_ms = 15000;
  one.min_version = 8;
  {
    WaitForStartingTaskResponse resp;  // the only servant is overloaded: busy, not "no environment"
    CHECK(scheduler.WaitForStartingTask("127.0.0.3", one, &resp) == kStatusNoQuotaAvailable);
  }
  for (std::size_t i = 1; i < pool.size(); ++i)
    CHECK(scheduler.Heartbeat(IpOf(pool[i].loc), Beat(pool[i].loc, pool[i].cap, pool[i].nproc,
                                                      pool[i].load, "Load Balance"), &hb) == kStatusSuccess);
  const int expect[] = {1, 2, 3, 2, 1, 2, 3};
  std::vector<std::uint64_t> grants;
  for (int want : expect) {
    WaitForStartingTaskResponse resp;
    CHECK(scheduler.WaitForStartingTask("127.0.0.3", one, &resp) == kStatusSuccess);
    CHECK(resp.grants.size() == 1);
    CHECK(resp.grants[0].servant_location == pool[want].loc);
    grants.push_back(resp.grants[0].task_grant_id);
    pool[want].load += 1;  // the reference test re-heartbeats the chosen servant with load + 1
    CHECK(scheduler.Heartbeat(IpOf(pool[want].loc), Beat(pool[want].loc, pool[want].cap, pool[want].nproc,
                                                         pool[want].load, "Load Balance"), &hb) ==
          kStatusSuccess);
  }
  for (std::size_t i = 0; i < grants.size(); ++i) CHECK(grants[i] == i);  // ids start at 0

  // ---- one RPC asking for 3 immediate + 2 prefetched grants: one device batch after the first ----
  {
    WaitForStartingTaskRequest many = one;
    many.immediate_reqs = 3;
    many.prefetch_reqs = 2;
    WaitForStartingTaskResponse resp;
    CHECK(scheduler.WaitForStartingTask("10.9.9.9", many, &resp) == kStatusSuccess);
    CHECK(resp.grants.size() == 5);
    for (std::size_t i = 0; i < 5; ++i) CHECK(resp.grants[i].task_grant_id == grants.size() + i);
    std::vector<std::uint64_t> ids;
    for (auto&& g : resp.grants) ids.push_back(g.task_grant_id);
    std::vector<bool> alive;
    CHECK(scheduler.KeepTaskAlive("", ids, 10000, &alive) == kStatusSuccess);
    for (bool a : alive) CHECK(a);
    CHECK(scheduler.KeepTaskAlive("", {424242}, 10000, &alive) == kStatusSuccess && !alive[0]);
    CHECK(scheduler.FreeTask("", ids) == kStatusSuccess);
  }

  // ---- status mapping ----
  {
    WaitForStartingTaskRequest bad = one;
    bad.compiler_digest = "nobody has this";
    WaitForStartingTaskResponse resp;
    CHECK(scheduler.WaitForStartingTask("10.9.9.9", bad, &resp) == kStatusEnvironmentNotAvailable);
    bad = one;
    bad.milliseconds_to_wait = 10001;
    CHECK(scheduler.WaitForStartingTask("10.9.9.9", bad, &resp) == kStatusInvalidArgument);
    HeartbeatRequest late = Beat("10.0.0.9:8335", 8, 16, 0, "Load Balance");
    late.next_heartbeat_in_ms = 30001;
    CHECK(scheduler.Heartbeat("10.0.0.9", late, &hb) == kStatusInvalidArgument);
    late.next_heartbeat_in_ms = 1000;
    late.location = "not an endpoint";
    CHECK(scheduler.Heartbeat("10.0.0.9", late, &hb) == kStatusInvalidArgument);
  }
  // A servant behind NAT (observed address != reported address) is registered but gets no task
  // (scheduler_service_impl.cc:146-153): the digest only it has is "not available".
  {
    CHECK(scheduler.Heartbeat("203.0.113.7", Beat("10.0.0.77:8335", 8, 16, 0, "nat only"), &hb) ==
          kStatusSuccess);
    WaitForStartingTaskRequest q = one;
    q.compiler_digest = "nat only";
    WaitForStartingTaskResponse resp;
    CHECK(scheduler.WaitForStartingTask("10.9.9.9", q, &resp) == kStatusEnvironmentNotAvailable);
    CHECK(dispatcher.DumpInternals().find("NOT_ACCEPTING_TASK_REASON_BEHIND_NAT") != std::string::npos);
  }

  // ---- running-task report cycle (Heartbeat -> NotifyServantRunningTasks -> GetRunningTasks) ----
  {
    HeartbeatRequest r = Beat(pool[1].loc, pool[1].cap, pool[1].nproc, pool[1].load, "Load Balance");
    RunningTask known, unknown;
    known.servant_task_id = 1;
    known.task_grant_id = grants[0];  // granted on pool[1] above
    known.servant_location = pool[1].loc;
    known.task_digest = "abc";
    unknown = known;
    unknown.task_grant_id = 999999;
    r.running_tasks = {known, unknown};
    CHECK(scheduler.Heartbeat(IpOf(pool[1].loc), r, &hb) == kStatusSuccess);
    CHECK(hb.expired_tasks.size() == 1 && hb.expired_tasks[0] == 999999);
    auto running = scheduler.GetRunningTasks();
    CHECK(running.size() == 1 && running[0].task_grant_id == grants[0] && running[0].task_digest == "abc");
  }
  // Leases run out -> zombies -> the servant's next report without them frees the slots.
  {
    g_now += 20s;
    dispatcher.OnExpirationTimer();  // grants[] leases (15 s) expired; servants' 10 s leases too
    std::vector<bool> alive;
    CHECK(scheduler.KeepTaskAlive("", {grants[0]}, 10000, &alive) == kStatusSuccess && !alive[0]);
  }
  std::printf("HARNESS-OK\n");
  return 0;
}

// tools/parked_workload.h — the "many parked waiters on a saturated pool" workload, once, for both
// sides: tools/td_native_bench.cc runs it on the TaskDispatcher surface of libydc.so, and
// oracle/ref_parked_bench.cc on the reference class itself (bench.py's cpu_baseline leg). Nothing of
// either implementation is in here: the dispatcher is an adapter with
//   bool Wait(int waiter, long long timeout_ms, unsigned long long* id)  // true: granted
//   void Free(unsigned long long id)
//
// The reference says of this path that it "doesn't scale well" (task_dispatcher.h:281-288): every
// FreeTask ends in notify_all (.cc:185-187), every parked waiter wakes up, takes the one lock and
// scans the registry (.cc:101-119). The workload: a pool whose every slot is taken, K waiter
// threads parked in WaitForStartingNewTask with a deadline seconds away; a releaser gives one slot
// back at a time.
//   phase 1 (latency): free one grant, wait until some waiter reports its grant — `samples` times;
//                      wake-to-grant = the waiter's return minus the start of that FreeTask.
//   phase 2 (rate):    the releaser frees whatever the waiters hand back as fast as it can for
//                      `seconds`: frees per second (= grants per second, the pool stays full).
#ifndef YDC_TOOLS_PARKED_WORKLOAD_H_
#define YDC_TOOLS_PARKED_WORKLOAD_H_
#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace parked {

using Clk = std::chrono::steady_clock;
inline long long NowNs() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(Clk::now().time_since_epoch()).count();
}

struct Result {
  int waiters = 0, samples = 0;
  double wake_to_grant_us_p50 = 0, wake_to_grant_us_p99 = 0, wake_to_grant_us_mean = 0;
  double free_call_us_mean = 0;  // what the FreeTask caller itself pays
  double frees_per_s = 0;
  long long rate_frees = 0;
  double rate_seconds = 0;
};

template <class Adapter>
struct Run {
  Adapter* a;
  std::atomic<bool> stop{false};
  std::atomic<long long> grants{0}, last_grant_ns{0};
  std::atomic<int> parked_once{0};
  std::mutex mu;
  std::deque<unsigned long long> held;  // grants waiting for the releaser

  struct Arg {
    Run* run;
    int k;
  };
  static void* Waiter(void* p) {
    Arg* arg = (Arg*)p;
    Run* r = arg->run;
    bool counted = false;
    while (!r->stop.load(std::memory_order_acquire)) {
      if (!counted) {
        counted = true;
        r->parked_once.fetch_add(1);
      }
      unsigned long long id = 0;
      if (!r->a->Wait(arg->k, 10000, &id)) continue;  // (deadline passed: park again)
      const long long t = NowNs();
      {
        std::scoped_lock _(r->mu);
        r->held.push_back(id);
      }
      r->last_grant_ns.store(t, std::memory_order_relaxed);
      r->grants.fetch_add(1, std::memory_order_release);
    }
    return nullptr;
  }

  bool Pop(unsigned long long* id) {
    std::scoped_lock _(mu);
    if (held.empty()) return false;
    *id = held.front();
    held.pop_front();
    return true;
  }

  // `initial`: the grants that fill the pool (taken by the caller before the waiters start).
  Result Go(int n_waiters, const std::vector<unsigned long long>& initial, int samples, double seconds) {
    held.assign(initial.begin(), initial.end());
    std::vector<pthread_t> th(n_waiters);
    std::vector<Arg> args(n_waiters);
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 256 * 1024);
    for (int k = 0; k < n_waiters; ++k) {
      args[k] = {this, k};
      if (pthread_create(&th[k], &attr, &Waiter, &args[k]) != 0) {
        std::fprintf(stderr, "pthread_create failed at waiter %d\n", k);
        std::exit(3);
      }
    }
    while (parked_once.load() < n_waiters) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    std::this_thread::sleep_for(std::chrono::milliseconds(300));  // (everybody is inside its wait by now)
    Result out;
    out.waiters = n_waiters;
    std::vector<double> us;
    double free_us = 0;
    for (int s = -5; s < samples; ++s) {
      unsigned long long id;
      while (!Pop(&id)) std::this_thread::yield();
      const long long before = grants.load(std::memory_order_acquire);
      const long long t0 = NowNs();
      a->Free(id);
      const long long t1 = NowNs();
      while (grants.load(std::memory_order_acquire) == before) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      if (s >= 0) {
        us.push_back((last_grant_ns.load(std::memory_order_relaxed) - t0) / 1e3);
        free_us += (t1 - t0) / 1e3;
      }
    }
    std::sort(us.begin(), us.end());
    out.samples = (int)us.size();
    double sum = 0;
    for (double v : us) sum += v;
    out.wake_to_grant_us_p50 = us[us.size() / 2];
    out.wake_to_grant_us_p99 = us[std::min(us.size() - 1, us.size() * 99 / 100)];
    out.wake_to_grant_us_mean = sum / us.size();
    out.free_call_us_mean = free_us / us.size();
    // phase 2
    const long long r0 = NowNs();
    long long n = 0;
    while ((NowNs() - r0) / 1e9 < seconds) {
      unsigned long long id;
      if (!Pop(&id)) {
        std::this_thread::yield();
        continue;
      }
      a->Free(id);
      ++n;
    }
    out.rate_seconds = (NowNs() - r0) / 1e9;
    out.rate_frees = n;
    out.frees_per_s = n / out.rate_seconds;
    // wind down: let every waiter see `stop` (a parked one returns at its deadline, <= 10 s, or with
    // a grant from the frees below)
    stop.store(true, std::memory_order_release);
    const long long w0 = NowNs();
    for (int k = 0; k < n_waiters; ++k) {
      for (;;) {
        unsigned long long id;
        while (Pop(&id)) a->Free(id);
        timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        ts.tv_nsec += 20 * 1000000;
        if (ts.tv_nsec >= 1000000000) ts.tv_sec++, ts.tv_nsec -= 1000000000;
        if (pthread_timedjoin_np(th[k], nullptr, &ts) == 0) break;
        if ((NowNs() - w0) / 1e9 > 60) {
          std::fprintf(stderr, "waiter %d did not return\n", k);
          std::exit(4);
        }
      }
    }
    return out;
  }
};

inline void Print(const char* side, const Result& r, bool last) {
  std::printf("\"%d\": {\"side\": \"%s\", \"wake_to_grant_us\": {\"p50\": %.1f, \"p99\": %.1f, \"mean\": %.1f, \"samples\": %d}, "
              "\"free_task_call_us\": %.1f, \"frees_per_s\": %.0f, \"rate_phase\": {\"frees\": %lld, \"seconds\": %.2f}}%s",
              r.waiters, side, r.wake_to_grant_us_p50, r.wake_to_grant_us_p99, r.wake_to_grant_us_mean, r.samples,
              r.free_call_us_mean, r.frees_per_s, r.rate_frees, r.rate_seconds, last ? "" : ", ");
}

}  // namespace parked
#endif

// gpu_task_dispatcher.cc — see gpu_task_dispatcher.h. Placement goes through the
// C-ABI only (ydc_upload_servants / ydc_update_servants / ydc_release_slots /
// ydc_dispatch); this file never computes a pick itself.
#include "gpu_task_dispatcher.h"

#include <algorithm>
#include <queue>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#if defined(__linux__)
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#endif

#include "../../include/yadcc_dispatch.h"

using namespace std::literals;

namespace ydc {

namespace {

// TryParseSize (reference yadcc/common/parse_size.cc:25-45): digits with an
// optional G/M/K/B suffix.
bool ParseSize(const std::string& s, std::size_t* out) {
  if (s.empty()) return false;
  std::uint64_t scale = 1;
  std::string body = s;
  switch (s.back()) {
    case 'G': scale = 1ull << 30; body.pop_back(); break;
    case 'M': scale = 1ull << 20; body.pop_back(); break;
    case 'K': scale = 1ull << 10; body.pop_back(); break;
    case 'B': body.pop_back(); break;
    default: break;
  }
  if (body.empty()) return false;
  std::uint64_t v = 0;
  for (char c : body) {
    if (c < '0' || c > '9') return false;
    v = v * 10 + (std::uint64_t)(c - '0');
  }
  *out = (std::size_t)(v * scale);
  return true;
}

// The requestor-is-the-servant test (IsNetworkAddressEqual, task_dispatcher.cc:66-69):
// `location` is longer than `requestor_ip`, has a ':' right behind it and starts with it. So a
// location answers to exactly the prefixes that end right before one of its ':' characters —
// "10.0.0.1:8335" to "10.0.0.1"; "[::1]:8335" (what scheduler_service_impl.cc:102-103 builds
// for IPv6 peers) to "[::1]", and also to "[:" and "[". Returns them longest first; empty for a
// location without any ':' (which equals no requestor address in the reference:
// `ip_port[ip2.size()] == ':'` can never hold).
std::vector<std::string> RequestorPrefixes(const std::string& location) {
  std::vector<std::string> out;
  for (std::size_t p = location.rfind(':'); p != std::string::npos; p = p ? location.rfind(':', p - 1) : p) {
    out.push_back(location.substr(0, p));
    if (p == 0) break;
  }
  return out;
}

// Sleeps while *word == expected, at most `ns` nanoseconds (spurious returns are fine).
inline void FutexWait(std::atomic<std::uint32_t>* word, std::uint32_t expected, long ns) {
#if defined(__linux__)
  timespec ts{0, ns};
  syscall(SYS_futex, reinterpret_cast<std::uint32_t*>(word), FUTEX_WAIT_PRIVATE, expected, &ts, nullptr, 0);
#else
  (void)word; (void)expected;
  std::this_thread::sleep_for(std::chrono::nanoseconds(ns));
#endif
}
inline void FutexWakeAll(std::atomic<std::uint32_t>* word) {
#if defined(__linux__)
  syscall(SYS_futex, reinterpret_cast<std::uint32_t*>(word), FUTEX_WAKE_PRIVATE, 0x7FFFFFFF, nullptr, nullptr, 0);
#else
  (void)word;
#endif
}

inline std::uint64_t NowNs() {
  return (std::uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

std::uint32_t Clamp32(std::size_t v) { return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (std::uint32_t)v; }

void JsonEscape(const std::string& s, std::string* out) {
  out->push_back('"');
  for (unsigned char c : s) {
    switch (c) {
      case '"': *out += "\\\""; break;
      case '\\': *out += "\\\\"; break;
      case '\n': *out += "\\n"; break;
      case '\r': *out += "\\r"; break;
      case '\t': *out += "\\t"; break;
      default:
        if (c < 0x20) {
          char buf[8];
          std::snprintf(buf, sizeof(buf), "\\u%04x", c);
          *out += buf;
        } else {
          out->push_back((char)c);
        }
    }
  }
  out->push_back('"');
}

const char* PriorityName(int p) {  // ServantPriority_Name, api/scheduler.proto:39-48
  switch (p) {
    case kServantPriorityDedicated: return "SERVANT_PRIORITY_DEDICATED";
    case kServantPriorityUser: return "SERVANT_PRIORITY_USER";
    case kServantPriorityUnknown: return "SERVANT_PRIORITY_UNKNOWN";
    default: return "";  // (protoc's *_Name() of a number without a name)
  }
}

const char* ReasonName(int r) {  // NotAcceptingTaskReason_Name, api/scheduler.proto:51-62
  switch (r) {
    case 1: return "NOT_ACCEPTING_TASK_REASON_USER_INSTRUCTED";
    case 2: return "NOT_ACCEPTING_TASK_REASON_POOR_MACHINE";
    case 3: return "NOT_ACCEPTING_TASK_REASON_CGROUPS_PRESENT";
    case 4: return "NOT_ACCEPTING_TASK_REASON_BEHIND_NAT";
    case 100: return "NOT_ACCEPTING_TASK_REASON_NOT_VERIFIED";
    case 0: return "NOT_ACCEPTING_TASK_REASON_UNKNOWN";
    default: return "";
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// RunningTaskBookkeeper
// ---------------------------------------------------------------------------
void RunningTaskBookkeeper::SetServantRunningTasks(const std::string& servant_location,
                                                   std::vector<RunningTask> tasks) {
  std::vector<RunningTaskView> views(tasks.size());
  std::vector<std::uint32_t> keep(tasks.size());
  for (std::size_t i = 0; i != tasks.size(); ++i) {
    views[i] = {tasks[i].servant_task_id, tasks[i].task_grant_id, tasks[i].servant_location,
                tasks[i].task_digest};
    keep[i] = (std::uint32_t)i;
  }
  SetServantRunningTasks(servant_location, views.data(), keep.data(), keep.size());
}

void RunningTaskBookkeeper::SetServantRunningTasks(std::string_view servant_location,
                                                   const RunningTaskView* tasks,
                                                   const std::uint32_t* keep, std::size_t n_keep) {
  std::scoped_lock _(lock_);
  std::vector<RunningTask>* have = running_tasks_.find(servant_location);
  if (have && have->size() == n_keep) {
    // The report of a second ago, most of the time: nothing to replace, nothing to invalidate
    // (the reference erases and re-inserts, running_task_bookkeeper.cc:24-29 — same content).
    bool same = true;
    for (std::size_t k = 0; k != n_keep && same; ++k) {
      const RunningTaskView& t = tasks[keep[k]];
      const RunningTask& h = (*have)[k];
      same = h.task_grant_id == t.task_grant_id && h.servant_task_id == t.servant_task_id &&
             h.servant_location == t.servant_location && h.task_digest == t.task_digest;
    }
    if (same) return;
  }
  if (!have) have = running_tasks_.emplace(servant_location, {}).first;
  have->resize(n_keep);
  for (std::size_t k = 0; k != n_keep; ++k) {
    const RunningTaskView& t = tasks[keep[k]];
    RunningTask& h = (*have)[k];
    h.servant_task_id = t.servant_task_id;
    h.task_grant_id = t.task_grant_id;
    h.servant_location.assign(t.servant_location.data(), t.servant_location.size());
    h.task_digest.assign(t.task_digest.data(), t.task_digest.size());
  }
  flattened_.reset();
  columns_.reset();
}

void RunningTaskBookkeeper::DropServant(std::string_view servant_location) {
  std::scoped_lock _(lock_);
  const std::vector<RunningTask>* have = running_tasks_.find(servant_location);
  if (!have) return;
  const bool had_tasks = !have->empty();
  running_tasks_.erase(servant_location);
  if (had_tasks) {
    flattened_.reset();
    columns_.reset();
  }
}

RunningTaskBookkeeper::Snapshot RunningTaskBookkeeper::GetRunningTasksShared() const {
  std::scoped_lock _(lock_);
  if (!flattened_) {
    // Order across servants is unspecified in the reference (hash-map order,
    // running_task_bookkeeper.cc:39-41); order within a servant is kept.
    auto flat = std::make_shared<std::vector<RunningTask>>();
    std::size_t total = 0;
    running_tasks_.for_each([&](const std::string&, const std::vector<RunningTask>& v) { total += v.size(); });
    flat->reserve(total);
    running_tasks_.for_each([&](const std::string&, const std::vector<RunningTask>& v) {
      flat->insert(flat->end(), v.begin(), v.end());
    });
    flattened_ = std::move(flat);
    ++rebuilds_;
  }
  return flattened_;
}

std::vector<RunningTask> RunningTaskBookkeeper::GetRunningTasks() const { return *GetRunningTasksShared(); }

RunningTaskBookkeeper::ColumnsSnapshot RunningTaskBookkeeper::GetRunningTasksColumns() const {
  std::scoped_lock _(lock_);
  if (!columns_) {
    auto cols = std::make_shared<RunningTaskColumns>();
    std::size_t total = 0;
    running_tasks_.for_each([&](const std::string&, const std::vector<RunningTask>& v) { total += v.size(); });
    cols->servant_task_ids.reserve(total);
    cols->task_grant_ids.reserve(total);
    cols->location_off.reserve(total);
    cols->digest_off.reserve(total);
    cols->location_len.reserve(total);
    cols->digest_len.reserve(total);
    FlatStringMap<std::uint32_t> pooled;  // string -> offset in the pool
    auto intern = [&](const std::string& str) {
      if (const std::uint32_t* at = pooled.find(str)) return *at;
      const std::uint32_t at = (std::uint32_t)cols->strings.size();
      cols->strings.append(str);
      cols->strings.push_back('\0');
      pooled.emplace(str, at);
      return at;
    };
    running_tasks_.for_each([&](const std::string&, const std::vector<RunningTask>& v) {
      const std::string* last_loc = nullptr;
      std::uint32_t last_at = 0;
      for (const RunningTask& t : v) {
        cols->servant_task_ids.push_back(t.servant_task_id);
        cols->task_grant_ids.push_back(t.task_grant_id);
        // (a servant's tasks carry its own location: asked once)
        if (!last_loc || *last_loc != t.servant_location) {
          last_at = intern(t.servant_location);
          last_loc = &t.servant_location;
        }
        cols->location_off.push_back(last_at);
        cols->location_len.push_back((std::uint32_t)t.servant_location.size());
        cols->digest_off.push_back(intern(t.task_digest));
        cols->digest_len.push_back((std::uint32_t)t.task_digest.size());
      }
    });
    columns_ = std::move(cols);
    ++rebuilds_;
  }
  return columns_;
}

// ---------------------------------------------------------------------------
// Task records
// ---------------------------------------------------------------------------
GpuTaskDispatcher::Task* GpuTaskDispatcher::TaskTable::find(std::uint64_t id) {
  const std::uint64_t page = id >> kPageBits;
  if (page < first_page_ || page - first_page_ >= pages_.size()) return nullptr;
  Page* p = pages_[page - first_page_].get();
  if (!p) return nullptr;
  Task* t = &p->tasks[id & ((1u << kPageBits) - 1)];
  return t->live ? t : nullptr;
}

GpuTaskDispatcher::Task* GpuTaskDispatcher::TaskTable::create(std::uint64_t id) {
  const std::uint64_t page = id >> kPageBits;
  if (pages_.empty()) first_page_ = page;
  while (page - first_page_ >= pages_.size()) pages_.emplace_back();
  auto& p = pages_[page - first_page_];
  if (!p) {
    if (!spare_.empty()) {
      p = std::move(spare_.back());
      spare_.pop_back();
    } else {
      p.reset(new Page);  // (records are initialised one by one as they are created)
    }
    p->live = 0;
  }
  Task* t = &p->tasks[id & ((1u << kPageBits) - 1)];
  t->prev = kNoTask;
  t->zombie = false;
  t->live = true;
  ++p->live;
  ++live_;
  next_id_ = id + 1;
  // The page before this one may have emptied while ids were still being handed out from it.
  if ((id & ((1u << kPageBits) - 1)) == 0 && page > first_page_) {
    auto& before = pages_[page - first_page_ - 1];
    if (before && before->live == 0) {
      if (spare_.size() < 64) spare_.push_back(std::move(before));
      before.reset();
      while (pages_.size() > 1 && !pages_.front()) {
        pages_.pop_front();
        ++first_page_;
      }
    }
  }
  return t;
}

void GpuTaskDispatcher::TaskTable::erase(std::uint64_t id) {
  Task* t = find(id);
  if (!t) return;
  t->live = false;
  --live_;
  auto& p = pages_[(id >> kPageBits) - first_page_];
  if (--p->live == 0) {
    // An emptied page goes, unless ids are still being handed out from it — decided by id, not
    // by its place in the deque: a full page that empties while it is still the back would
    // otherwise stay for good once the next page is appended behind it.
    if ((next_id_ >> kPageBits) != (id >> kPageBits)) {
      if (spare_.size() < 64) spare_.push_back(std::move(p));
      p.reset();
    }
    while (pages_.size() > 1 && !pages_.front()) {
      pages_.pop_front();
      ++first_page_;
    }
  }
}

std::uint32_t GpuTaskDispatcher::NamePool::intern(std::string_view s) {
  if (const std::uint32_t* id = ids_.find(s)) return *id;
  const std::uint32_t id = (std::uint32_t)names_.size();
  names_.emplace_back(s);
  ids_.emplace(s, id);
  return id;
}

// ---------------------------------------------------------------------------
// GpuTaskDispatcher
// ---------------------------------------------------------------------------
GpuTaskDispatcher::GpuTaskDispatcher(const Options& options) : options_(options) {
  if (!ParseSize(options_.servant_min_memory_for_accepting_new_task, &min_memory_for_new_task_)) {
    // The reference FLARE_CHECKs the flag (task_dispatcher.cc:83-86).
    std::fprintf(stderr, "GpuTaskDispatcher: cannot parse memory size [%s]\n",
                 options_.servant_min_memory_for_accepting_new_task.c_str());
    std::abort();
  }
  if (options_.device >= 0) {
    device_status_ = ydc_create(options_.device, 0, 0, 0, nullptr, &ctx_);
    if (device_status_ != YDC_OK) {
      device_error_ = std::string(ydc_strerror(device_status_)) + ": " + ydc_last_error(nullptr);
      ctx_ = nullptr;
    }
  } else {
    device_status_ = YDC_ERR_NO_DEVICE;
    device_error_ = "created without a device";
  }
  if (options_.start_expiration_timer) timer_ = std::thread([this] { TimerLoop(); });
}

GpuTaskDispatcher::~GpuTaskDispatcher() {
  {
    std::scoped_lock _(timer_lock_);
    stopping_ = true;
  }
  timer_cv_.notify_all();
  if (timer_.joinable()) timer_.join();
  if (ctx_) ydc_destroy(ctx_);
}

void GpuTaskDispatcher::TimerLoop() {
  std::unique_lock lk(timer_lock_);
  while (!stopping_) {
#if defined(__SANITIZE_THREAD__)
    if (timer_cv_.wait_until(lk, std::chrono::system_clock::now() + 1s, [this] { return stopping_; })) break;
#else
    if (timer_cv_.wait_for(lk, 1s, [this] { return stopping_; })) break;
#endif
    lk.unlock();
    OnExpirationTimer();
    lk.lock();
  }
}

GpuTaskDispatcher::HostStats GpuTaskDispatcher::host_stats() const {
  Section sec(const_cast<GpuTaskDispatcher*>(this));
  HostStats s = host_stats_;
  s.bookkeeper_rebuilds = running_task_bookkeeper_.rebuilds();
  s.lease_pages = tasks_.pages();
  s.lease_wheel_entries = lease_wheel_.entries();
  return s;
}

// ---------------------------------------------------------------------------
// Operation log (test switch, gpu_task_dispatcher.h)
// ---------------------------------------------------------------------------
namespace {
void LogKey(std::string* out, const char* key) {
  if (out->back() != '{') *out += ", ";
  out->push_back('"');
  *out += key;
  *out += "\": ";
}
void LogNum(std::string* out, const char* key, long long v) {
  LogKey(out, key);
  *out += std::to_string(v);
}
void LogStr(std::string* out, const char* key, std::string_view v) {
  LogKey(out, key);
  JsonEscape(std::string(v), out);
}
void LogOpen(std::string* out, const char* op) {
  *out += out->empty() ? "{" : ",\n{";
  LogStr(out, "op", op);
}
}  // namespace

void GpuTaskDispatcher::EnableOpLog(bool on) {
  Section sec(this);
  oplog_on_ = on;
}

std::string GpuTaskDispatcher::TakeOpLog() {
  Section sec(this);
  std::string out = "[" + oplog_ + "]";
  oplog_.clear();
  return out;
}

void GpuTaskDispatcher::LogWait(const RequestView& r, std::chrono::nanoseconds lease, Clock::time_point now,
                                int status, std::uint64_t id, const Servant* pick, std::uint32_t attempt) {
  LogOpen(&oplog_, "wait");
  LogNum(&oplog_, "try", attempt);
  LogNum(&oplog_, "now", now.time_since_epoch().count());
  LogStr(&oplog_, "ip", r.requestor_ip);
  LogStr(&oplog_, "digest", r.compiler_digest);
  LogNum(&oplog_, "minv", r.min_version);
  LogNum(&oplog_, "lease", lease.count());
  LogNum(&oplog_, "st", status);
  if (pick) {
    LogNum(&oplog_, "id", (long long)id);
    LogStr(&oplog_, "loc", pick->personality.observed_location);
  }
  oplog_ += "}";
}

std::size_t GpuTaskDispatcher::CapacityAvailable(const Servant& s) const {
  // GetCapacityAvailable, task_dispatcher.cc:283-313 (for DumpInternals only;
  // the dispatch path evaluates the same formula on the device).
  const auto& p = s.personality;
  if (p.total_memory_in_bytes != 0 && p.memory_available_in_bytes < min_memory_for_new_task_)
    return s.running_tasks;
  std::int64_t foreign = std::max<std::int64_t>((std::int64_t)p.current_load - (std::int64_t)s.running_tasks, 0);
  std::int64_t cap = std::max<std::int64_t>((std::int64_t)p.num_processors - foreign, 0);
  return std::min<std::size_t>(p.max_tasks, (std::size_t)cap);
}

std::uint32_t GpuTaskDispatcher::InternIp(std::string_view ip, bool create) {
  if (const std::uint32_t* id = ip_ids_.find(ip)) return *id;
  if (!create) return 0;  // 0: no servant lives there
  const std::uint32_t id = (std::uint32_t)ip_ids_.size() + 1;
  ip_ids_.emplace(ip, id);
  return id;
}

std::uint32_t GpuTaskDispatcher::RequestorId(std::string_view ip) {
  if (!shorter_prefix_refs_.empty()) {
    // Some location answers to `ip` through one of its shorter prefixes: from now on that is an
    // entry of the device's lookup table (next UnsafeSyncDevice).
    if (const std::uint32_t* a = alias_ids_.find(ip)) return *a;
    if (shorter_prefix_refs_.count(ip)) {
      const std::uint32_t id = InternIp(ip, true);
      alias_ids_.emplace(ip, id);
      aliases_dirty_ = true;
      ++registry_epoch_;
      return id;
    }
  }
  return InternIp(ip, false);
}

const GpuTaskDispatcher::EnvEntry* GpuTaskDispatcher::LookupEnv(std::string_view digest) const {
  return env_ids_.find(digest);  // null: nobody has it
}

// Interns the digests a servant advertises: one bit number per digest, as many 64-bit mask
// words as the live digests need (the reference keeps an unbounded vector of
// EnvironmentDesc per servant, task_dispatcher.h:93-94 — no limit here either). Returns
// the bit numbers in listing order; a digest listed twice holds two references, and
// ReleaseEnvBits walks the same list.
template <class Strings>
std::vector<std::uint32_t> GpuTaskDispatcher::AcquireEnvBits(const Strings& digests, std::size_t n) {
  std::vector<std::uint32_t> bits;
  bits.reserve(n);
  for (std::size_t i = 0; i != n; ++i) {
    const std::string_view d = digests[i];
    EnvEntry* e = env_ids_.find(d);
    if (!e) {
      EnvEntry fresh;
      if (!free_env_bits_.empty()) {
        fresh.bit = free_env_bits_.back();
        free_env_bits_.pop_back();
      } else {
        fresh.bit = next_env_bit_++;
      }
      fresh.name = names_.intern(d);
      e = env_ids_.emplace(d, fresh).first;
    }
    ++e->refs;
    bits.push_back(e->bit);
  }
  return bits;
}

void GpuTaskDispatcher::ReleaseEnvBits(const std::vector<std::string>& digests) {
  for (auto&& d : digests) {
    EnvEntry* e = env_ids_.find(d);
    if (!e) continue;
    if (--e->refs == 0) {
      free_env_bits_.push_back(e->bit);
      env_ids_.erase(d);
    }
  }
}

// ---------------------------------------------------------------------------
// Servant maintenance
// ---------------------------------------------------------------------------
void GpuTaskDispatcher::KeepServantAlive(const ServantPersonality& servant,
                                         std::chrono::nanoseconds expires_in) {
  std::vector<std::string_view> envs(servant.environments.begin(), servant.environments.end());
  ServantView v;
  v.version = servant.version;
  v.observed_location = servant.observed_location;
  v.reported_location = servant.reported_location;
  v.environments = envs.data();
  v.n_environments = envs.size();
  v.num_processors = servant.num_processors;
  v.current_load = servant.current_load;
  v.total_memory_in_bytes = servant.total_memory_in_bytes;
  v.memory_available_in_bytes = servant.memory_available_in_bytes;
  v.max_tasks = servant.max_tasks;
  v.priority = servant.priority;
  v.not_accepting_task_reason = servant.not_accepting_task_reason;
  KeepServantAlive(v, expires_in);
}

void GpuTaskDispatcher::KeepServantAlive(const ServantView& servant, std::chrono::nanoseconds expires_in) {
  Section sec(this);
  auto now = Now();
  ++host_stats_.heartbeats;
  if (oplog_on_) {
    LogOpen(&oplog_, "servant");
    LogNum(&oplog_, "now", now.time_since_epoch().count());
    LogNum(&oplog_, "lease", expires_in.count());
    LogStr(&oplog_, "location", servant.observed_location);
    LogStr(&oplog_, "reported", servant.reported_location);
    LogKey(&oplog_, "envs");
    oplog_ += "[";
    for (std::size_t k = 0; k != servant.n_environments; ++k) {
      if (k) oplog_ += ", ";
      JsonEscape(std::string(servant.environments[k]), &oplog_);
    }
    oplog_ += "]";
    LogNum(&oplog_, "version", servant.version);
    LogNum(&oplog_, "num_processors", (long long)servant.num_processors);
    LogNum(&oplog_, "current_load", (long long)servant.current_load);
    LogNum(&oplog_, "total_memory", (long long)servant.total_memory_in_bytes);
    LogNum(&oplog_, "memory_available", (long long)servant.memory_available_in_bytes);
    LogNum(&oplog_, "max_tasks", (long long)servant.max_tasks);
    LogNum(&oplog_, "priority", servant.priority);
    LogNum(&oplog_, "reason", servant.not_accepting_task_reason);
    oplog_ += "}";
  }
  auto assign_scalars = [&](ServantPersonality& p) {
    p.version = servant.version;
    p.num_processors = servant.num_processors;
    p.current_load = servant.current_load;
    p.total_memory_in_bytes = servant.total_memory_in_bytes;
    p.memory_available_in_bytes = servant.memory_available_in_bytes;
    p.max_tasks = servant.max_tasks;
    p.priority = servant.priority;
    p.not_accepting_task_reason = servant.not_accepting_task_reason;
  };
  std::uint32_t idx;
  if (const std::uint32_t* known = index_of_location_.find(servant.observed_location)) {
    // Renewal: the personality is replaced wholesale, running_tasks,
    // ever_assigned_tasks, discovered_at and the registry position stay
    // (task_dispatcher.cc:195-201). Most renewals repeat the environments (and often
    // everything else) of the last one: the digests are only interned anew when the list
    // changed, and a row whose device columns did not move is not sent again.
    idx = *known;
    Servant* e = servants_[idx].get();
    ServantPersonality& p = e->personality;
    e->expires_at = now + expires_in;
    servant_expires_at_[idx] = e->expires_at;
    bool same_envs = p.environments.size() == servant.n_environments;
    for (std::size_t k = 0; same_envs && k != servant.n_environments; ++k)
      same_envs = std::string_view(p.environments[k]) == servant.environments[k];
    const bool low_mem_was = p.total_memory_in_bytes != 0 && p.memory_available_in_bytes < min_memory_for_new_task_;
    const bool low_mem_is = servant.total_memory_in_bytes != 0 &&
                            servant.memory_available_in_bytes < min_memory_for_new_task_;
    const bool same_row = same_envs && p.version == servant.version &&
                          p.num_processors == servant.num_processors &&
                          p.current_load == servant.current_load && p.max_tasks == servant.max_tasks &&
                          p.priority == servant.priority && low_mem_was == low_mem_is;
    if (!same_envs) {
      ++registry_epoch_;
      auto bits = AcquireEnvBits(servant.environments, servant.n_environments);
      ReleaseEnvBits(p.environments);
      p.environments.assign(servant.environments, servant.environments + servant.n_environments);
      e->env_bits = std::move(bits);
    }
    assign_scalars(p);
    if (std::string_view(p.reported_location) != servant.reported_location)
      p.reported_location.assign(servant.reported_location.data(), servant.reported_location.size());
    if (same_row) {
      ++host_stats_.heartbeats_unchanged;
      return;
    }
  } else {
    idx = (std::uint32_t)servants_.size();
    ++registry_epoch_;
    auto added = std::make_unique<Servant>();
    added->uid = next_servant_uid_++;
    added->index = idx;
    ServantPersonality& p = added->personality;
    assign_scalars(p);
    p.observed_location.assign(servant.observed_location.data(), servant.observed_location.size());
    p.reported_location.assign(servant.reported_location.data(), servant.reported_location.size());
    if (p.observed_location.size() < sizeof(added->location_short)) {
      added->location_len = (std::uint8_t)p.observed_location.size();
      std::memcpy(added->location_short, p.observed_location.c_str(), p.observed_location.size() + 1);
    }
    p.environments.assign(servant.environments, servant.environments + servant.n_environments);
    added->discovered_at = now;
    added->expires_at = now + expires_in;
    added->running_tasks = 0;  // :210
    added->env_bits = AcquireEnvBits(servant.environments, servant.n_environments);
    auto prefixes = RequestorPrefixes(p.observed_location);
    // No ':' in the location: an id of its own that no requestor address can map to.
    added->ip_id = !prefixes.empty()
                       ? InternIp(prefixes.front(), true)
                       : InternIp(std::string("\0nohost#", 8) + std::to_string(added->uid), true);
    for (std::size_t k = 1; k < prefixes.size(); ++k) {
      ++shorter_prefix_refs_[prefixes[k]];
      if (alias_ids_.count(prefixes[k])) aliases_dirty_ = true;
      added->shorter_prefixes.push_back(std::move(prefixes[k]));
    }
    index_of_location_.emplace(p.observed_location, idx);
    servant_expires_at_.push_back(added->expires_at);
    servants_.push_back(std::move(added));
    row_is_dirty_.push_back(0);
  }
  if (!row_is_dirty_[idx]) {
    row_is_dirty_[idx] = 1;
    dirty_rows_.push_back(idx);
  }
  // The reference does not signal the condition variable here (task_dispatcher.cc:190-220):
  // a heartbeat wakes nobody. Parked waiters are retried when FreeTask wakes them
  // (UnsafeFreeTasks bumps wake_epoch_); a NEW caller sees the new capacity at once.
}

std::vector<std::uint64_t> GpuTaskDispatcher::NotifyServantRunningTasks(
    const std::string& servant_location, std::vector<RunningTask> tasks) {
  std::vector<RunningTaskView> views(tasks.size());
  for (std::size_t i = 0; i != tasks.size(); ++i)
    views[i] = {tasks[i].servant_task_id, tasks[i].task_grant_id, tasks[i].servant_location,
                tasks[i].task_digest};
  return NotifyServantRunningTasks(std::string_view(servant_location), views.data(), views.size());
}

std::vector<std::uint64_t> GpuTaskDispatcher::NotifyServantRunningTasks(std::string_view servant_location,
                                                                        const RunningTaskView* tasks,
                                                                        std::size_t n) {
  std::vector<std::uint64_t> unknown_tasks;
  Section sec(this);
  const std::uint32_t* known = index_of_location_.find(servant_location);
  auto log_report = [&] {
    if (!oplog_on_) return;
    LogOpen(&oplog_, "report");
    LogStr(&oplog_, "loc", servant_location);
    for (int which = 0; which != 2; ++which) {
      LogKey(&oplog_, which ? "unknown" : "ids");
      oplog_ += "[";
      const std::size_t m = which ? unknown_tasks.size() : n;
      for (std::size_t i = 0; i != m; ++i) {
        if (i) oplog_ += ", ";
        oplog_ += std::to_string(which ? unknown_tasks[i] : tasks[i].task_grant_id);
      }
      oplog_ += "]";
    }
    oplog_ += "}";
  };
  if (!known) {  // :241-243: the servant itself has expired — every reported id comes back
    unknown_tasks.reserve(n);
    for (std::size_t i = 0; i != n; ++i) unknown_tasks.push_back(tasks[i].task_grant_id);
    log_report();
    return unknown_tasks;
  }
  Servant* servant = servants_[*known].get();

  UnsafeSweepZombiesOf(servant, tasks, n);

  // Tasks reported by the servant but not (or no longer) granted on it are
  // returned, in report order (:256-273). The task record names its servant: no
  // scan over every task as in the reference.
  static thread_local std::vector<std::uint32_t> kept;
  kept.clear();
  for (std::size_t i = 0; i != n; ++i) {
    const Task* t = tasks_.find(tasks[i].task_grant_id);
    if (t && t->servant == servant && !t->zombie) {
      kept.push_back((std::uint32_t)i);
    } else {
      unknown_tasks.push_back(tasks[i].task_grant_id);
    }
  }
  running_task_bookkeeper_.SetServantRunningTasks(servant_location, tasks, kept.data(), kept.size());
  log_report();
  return unknown_tasks;
}

std::vector<RunningTask> GpuTaskDispatcher::GetRunningTasks() const {
  return running_task_bookkeeper_.GetRunningTasks();
}

// ---------------------------------------------------------------------------
// Leases
// ---------------------------------------------------------------------------
bool GpuTaskDispatcher::KeepTaskAlive(std::uint64_t task_id, std::chrono::nanoseconds new_expires_in) {
  Section sec(this);
  const auto now = Now();
  Task* t = tasks_.find(task_id);
  const bool ok = t && !t->zombie;  // :146-153, :154-162
  if (oplog_on_) {
    LogOpen(&oplog_, "renew");
    LogNum(&oplog_, "now", now.time_since_epoch().count());
    LogNum(&oplog_, "id", (long long)task_id);
    LogNum(&oplog_, "lease", new_expires_in.count());
    LogNum(&oplog_, "ok", ok);
    oplog_ += "}";
  }
  if (!ok) return false;
  const std::int64_t filed_under = LeaseWheel::SecondOf(t->expires_at);
  t->expires_at = now + new_expires_in;
  if (LeaseWheel::SecondOf(t->expires_at) != filed_under) FileLease(task_id, t->expires_at);
  return true;
}

void GpuTaskDispatcher::FileLease(std::uint64_t id, Clock::time_point expires_at) {
  lease_wheel_.File(id, expires_at);
  // Renewals and early frees leave entries behind; swept when they dominate.
  if (lease_wheel_.entries() > 4 * tasks_.size() + options_.lease_sweep_slack)
    lease_wheel_.Sweep([this](std::uint64_t i, std::int64_t second) {
      const Task* t = tasks_.find(i);
      return t && !t->zombie && LeaseWheel::SecondOf(t->expires_at) == second;
    });
}

bool GpuTaskDispatcher::UnsafeApplyQueuedFrees() {
  if (free_queued_.load(std::memory_order_seq_cst) == 0) return false;
  std::vector<std::uint64_t> ids;
  {
    std::scoped_lock _(queue_lock_);
    ids.swap(free_queue_);
    free_queued_.store(0, std::memory_order_seq_cst);
  }
  return UnsafeApplyFrees(ids);
}

bool GpuTaskDispatcher::UnsafeApplyFrees(const std::vector<std::uint64_t>& ids) {
  // (n calls, not one call with n ids: an unknown id ends a call, not the others — :176-180)
  for (std::uint64_t id : ids) {
    if (oplog_on_) {
      LogOpen(&oplog_, "free");
      LogNum(&oplog_, "id", (long long)id);
      oplog_ += "}";
    }
    UnsafeFreeTasks(&id, 1);
  }
  return !ids.empty();
}

void GpuTaskDispatcher::FreeTask(std::uint64_t task_id) { FreeTasks(&task_id, 1); }

void GpuTaskDispatcher::FreeTasks(const std::uint64_t* task_ids, std::size_t n) {
  if (n == 0) return;
  {
    std::scoped_lock _(queue_lock_);
    free_queue_.insert(free_queue_.end(), task_ids, task_ids + n);
    free_queued_.fetch_add((std::uint32_t)n, std::memory_order_seq_cst);
  }
  std::atomic_thread_fence(std::memory_order_seq_cst);  // (pairs with the one in ~Section)
  std::unique_lock lk(allocation_lock_, std::try_to_lock);
  if (!lk.owns_lock()) {
    // Somebody is in the middle of a device turn: it applies the ids when it is done (Section).
    // Unless threads sleep on the condition variable — they hold no lock, and the wake-up
    // (:187) is theirs.
    if (sleepers_.load(std::memory_order_seq_cst) == 0) return;
    lk.lock();
  }
  Section sec(this, std::move(lk));
}

void GpuTaskDispatcher::UnsafeFreeTasks(const std::uint64_t* task_ids, std::size_t n) {
  for (std::size_t i = 0; i != n; ++i) {
    const std::uint64_t id = task_ids[i];
    Task* t = tasks_.find(id);
    if (!t) return;  // quirk kept: bails out, no wake-up (:176-180)
    Servant* s = t->servant;
    --s->running_tasks;  // :181
    // unlink from the servant's grant list
    if (t->prev != kNoTask) tasks_.slot(t->prev)->next = t->next; else s->grants_head = t->next;
    if (t->next != kNoTask) tasks_.slot(t->next)->prev = t->prev;
    --s->n_grants;
    if (t->zombie) --s->n_zombies;
    if (!need_full_upload_ && !s->removed) pending_release_.push_back(s->index);
    tasks_.erase(id);
  }
  ++wake_epoch_;
  wake_pending_ = true;  // :187 notify_all — served by this thread before it lets go of the lock
}

// What the reference's woken waiters do one by one (re-run their loops, :101-119), done for all of
// them at once by the thread that woke them: the parked requests go through one device batch in
// arrival order; the owners of those that were served are woken, the others sleep on.
void GpuTaskDispatcher::UnsafeServeWoken() {
  if (!wake_pending_) return;
  wake_pending_ = false;
  if (parked_count_ != 0 || queued_.load(std::memory_order_relaxed) != 0) UnsafeDrainQueue();
}

void GpuTaskDispatcher::UnsafeSweepZombiesOf(Servant* servant, const RunningTaskView* reported,
                                             std::size_t n) {
  if (servant->n_zombies == 0) {
    // Nothing to sweep — but the reference's UnsafeFreeTasks still ends in notify_all for an
    // empty list (:187,453-476): every heartbeat of a known servant wakes the waiters.
    UnsafeFreeTasks(nullptr, 0);
    return;
  }
  std::unordered_set<std::uint64_t> running;
  running.reserve(n);
  for (std::size_t i = 0; i != n; ++i) running.insert(reported[i].task_grant_id);
  std::vector<std::uint64_t> sweeping;  // :453-476
  for (std::uint64_t id = servant->grants_head; id != kNoTask;) {
    const Task* t = tasks_.find(id);
    if (t->zombie && running.count(id) == 0) sweeping.push_back(id);
    id = t->next;
  }
  UnsafeFreeTasks(sweeping.data(), sweeping.size());
}

void GpuTaskDispatcher::OnExpirationTimer() {
  auto now = Now();
  Section sec(this);
  const std::uint64_t t_in = NowNs();
  if (oplog_on_) {
    LogOpen(&oplog_, "timer");
    LogNum(&oplog_, "now", now.time_since_epoch().count());
    oplog_ += "}";
  }

  // Expired servants leave the registry; the order of the others is kept
  // because it decides ties (:503-516).
  std::vector<std::uint32_t> expired;
  for (std::uint32_t i = 0; i != servant_expires_at_.size(); ++i)
    if (servant_expires_at_[i] < now) expired.push_back(i);
  std::vector<std::uint64_t> orphans;
  std::vector<std::unique_ptr<Servant>> removed;  // alive until their orphans are freed
  if (!expired.empty()) {
    ++registry_epoch_;
    // Device first, while the host rows still have their old positions: the deltas recorded
    // against those positions, then an order-preserving compaction of the resident columns on
    // the device (running_tasks of the survivors stays where it is; no table upload).
    if (ctx_ && !need_full_upload_) {
      int rc = UnsafeSyncDevice();
      if (rc == YDC_OK) rc = ydc_remove_servants(ctx_, expired.data(), (std::uint32_t)expired.size());
      if (rc != YDC_OK) need_full_upload_ = true;
    }
    std::size_t w = 0, next = 0;
    for (std::size_t i = 0; i != servants_.size(); ++i) {
      if (next < expired.size() && expired[next] == i) {
        ++next;
        Servant* s = servants_[i].get();
        running_task_bookkeeper_.DropServant(s->personality.observed_location);
        ReleaseEnvBits(s->personality.environments);
        for (std::uint64_t id = s->grants_head; id != kNoTask; id = tasks_.find(id)->next) orphans.push_back(id);
        for (auto&& p : s->shorter_prefixes) {
          std::uint32_t* refs = shorter_prefix_refs_.find(p);
          if (refs && --*refs == 0) shorter_prefix_refs_.erase(p);
        }
        index_of_location_.erase(s->personality.observed_location);
        s->removed = true;
        removed.push_back(std::move(servants_[i]));
      } else {
        if (w != i) {
          servants_[w] = std::move(servants_[i]);
          servant_expires_at_[w] = servant_expires_at_[i];
        }
        ++w;
      }
    }
    servants_.resize(w);
    servant_expires_at_.resize(w);
    for (std::uint32_t i = 0; i != servants_.size(); ++i) {
      servants_[i]->index = i;
      *index_of_location_.find(servants_[i]->personality.observed_location) = i;
    }
    dirty_rows_.clear();
    pending_release_.clear();
    row_is_dirty_.assign(servants_.size(), 0);
    if (!alias_ids_.empty()) aliases_dirty_ = true;  // (rows moved: the device dropped its aliases)
  }
  // UnsafeSweepOrphans (:478-496): tasks of vanished servants are forgotten at
  // once. Their ids are exactly the grant lists of the servants removed above. Called on every
  // tick, like the reference's (:518-520): UnsafeFreeTasks ends in notify_all even for an empty
  // list, so every parked waiter re-runs its loop at least once a second — that is how it gets
  // to see a new servant, a lighter load, or EnvironmentNotFound after the last eligible
  // servant expired (a heartbeat alone wakes nobody, :190-220).
  // (the servants are gone from the registry — Servant::removed —: no slot is released on
  // the device for them)
  UnsafeFreeTasks(orphans.data(), orphans.size());
  removed.clear();

  // Expired leases become zombies; they keep their slot until the servant's
  // next heartbeat no longer lists them (:523-535, task_dispatcher.h:207-214).
  // Only the leases filed under the seconds up to `now` are looked at (LeaseWheel), not every lease.
  std::uint64_t seen = 0;
  lease_wheel_.Due(
      now,
      [&](std::uint64_t id, std::int64_t second) {
        ++seen;
        const Task* t = tasks_.find(id);
        return t && !t->zombie && LeaseWheel::SecondOf(t->expires_at) == second;
      },
      [&](std::uint64_t id) {
        Task* t = tasks_.find(id);
        if (!(t->expires_at < now)) return false;
        t->zombie = true;
        ++t->servant->n_zombies;
        return true;
      });
  ++host_stats_.timer_ticks;
  host_stats_.timer_lease_entries_seen = seen;
  host_stats_.timer_last_ns = NowNs() - t_in;
  host_stats_.timer_max_ns = std::max(host_stats_.timer_max_ns, host_stats_.timer_last_ns);
}

// ---------------------------------------------------------------------------
// Placement
// ---------------------------------------------------------------------------
std::uint32_t GpuTaskDispatcher::DeviceFlags(const Servant& s) const {
  const auto& p = s.personality;
  std::uint32_t f = 0;
  if (p.priority == kServantPriorityDedicated) f |= YDC_SERVANT_DEDICATED;          // :405
  if (p.total_memory_in_bytes != 0 && p.memory_available_in_bytes < min_memory_for_new_task_)
    f |= YDC_SERVANT_LOW_MEMORY;                                                     // :286-287
  return f;
}

// The rows of dirty_rows_ (sorted: appended rows in index order) as the device API wants them,
// into sync_rows_ / sync_env_ (*env_words words per row).
void GpuTaskDispatcher::UnsafePackDirtyRows(std::uint32_t* env_words) {
  static_assert(sizeof(SyncRow) == sizeof(ydc_servant_row), "row layout");
  std::sort(dirty_rows_.begin(), dirty_rows_.end());
  const std::uint32_t ew = EnvWords();
  sync_rows_.resize(dirty_rows_.size());
  sync_env_.assign(dirty_rows_.size() * ew, 0);
  for (std::size_t k = 0; k != dirty_rows_.size(); ++k) {
    const Servant& s = *servants_[dirty_rows_[k]];
    SyncRow& r = sync_rows_[k];
    r.version = (std::uint32_t)s.personality.version;
    r.num_processors = Clamp32(s.personality.num_processors);
    r.current_load = Clamp32(s.personality.current_load);
    r.max_tasks = Clamp32(s.personality.max_tasks);
    r.flags = DeviceFlags(s);
    r.ip_id = s.ip_id;
    for (auto b : s.env_bits) sync_env_[k * ew + b / 64] |= 1ull << (b % 64);
    r.env_mask = sync_env_[k * ew];
  }
  *env_words = ew;
}

int GpuTaskDispatcher::UnsafeSyncDevice() {
  if (!ctx_) return device_status_ ? device_status_ : YDC_ERR_NO_DEVICE;
  auto flags_of = [this](const Servant& s) { return DeviceFlags(s); };
  if (need_full_upload_) {
    const std::size_t n = servants_.size();
    std::vector<std::uint32_t> version(n), nproc(n), load(n), max_tasks(n), running(n), flags(n), ip(n);
    const std::uint32_t ew = EnvWords();
    std::vector<std::uint64_t> env(n * ew, 0);
    for (std::size_t i = 0; i != n; ++i) {
      const Servant& s = *servants_[i];
      version[i] = (std::uint32_t)s.personality.version;  // compared as unsigned, :333
      nproc[i] = Clamp32(s.personality.num_processors);
      load[i] = Clamp32(s.personality.current_load);
      max_tasks[i] = Clamp32(s.personality.max_tasks);
      running[i] = Clamp32(s.running_tasks);
      flags[i] = flags_of(s);
      for (auto b : s.env_bits) env[i * ew + b / 64] |= 1ull << (b % 64);
      ip[i] = s.ip_id;
    }
    ydc_servant_soa soa{version.data(), nproc.data(), load.data(), max_tasks.data(),
                        running.data(), flags.data(),  env.data(),  ip.data(), ew};
    int rc = ydc_upload_servants(ctx_, &soa, (std::uint32_t)n);
    if (rc != YDC_OK) return rc;
    need_full_upload_ = false;
    dirty_rows_.clear();
    pending_release_.clear();
    row_is_dirty_.assign(n, 0);
    if (!alias_ids_.empty()) aliases_dirty_ = true;  // (a fresh table has no aliases)
    return UnsafeSyncAliases();
  }
  if (!dirty_rows_.empty()) {
    std::uint32_t ew = 1;
    UnsafePackDirtyRows(&ew);
    int rc = ydc_update_servants_wide(ctx_, dirty_rows_.data(), (const ydc_servant_row*)sync_rows_.data(),
                                      sync_env_.data(), ew, (std::uint32_t)dirty_rows_.size());
    if (rc != YDC_OK) return rc;
    for (auto i : dirty_rows_) row_is_dirty_[i] = 0;
    dirty_rows_.clear();
  }
  if (!pending_release_.empty()) {
    int rc = ydc_release_slots(ctx_, pending_release_.data(), (std::uint32_t)pending_release_.size());
    if (rc != YDC_OK) return rc;
    pending_release_.clear();
  }
  return UnsafeSyncAliases();
}

// The presented shorter prefixes as (host id, servant row) entries of the device's lookup table.
int GpuTaskDispatcher::UnsafeSyncAliases() {
  if (!aliases_dirty_) return YDC_OK;
  std::vector<std::uint32_t> ids, rows;
  for (std::uint32_t i = 0; i != servants_.size(); ++i)
    for (auto&& p : servants_[i]->shorter_prefixes) {
      if (const std::uint32_t* a = alias_ids_.find(p)) {
        ids.push_back(*a);
        rows.push_back(i);
      }
    }
  int rc = ydc_set_host_aliases(ctx_, ids.data(), rows.data(), (std::uint32_t)ids.size());
  if (rc == YDC_OK) aliases_dirty_ = false;
  return rc;
}

std::uint32_t* GpuTaskDispatcher::HostColumn::ensure(std::size_t n) {
  if (n <= cap) return p;
  if (p) {
    if (pinned) ydc_host_free(p); else std::free(p);
  }
  p = nullptr;
  cap = std::max<std::size_t>(n + n / 2, 1024);
  void* q = nullptr;
  pinned = ydc_host_alloc(cap * sizeof(std::uint32_t), &q) == YDC_OK;
  if (!pinned) q = std::malloc(cap * sizeof(std::uint32_t));  // (no device: the batch fails anyway)
  p = (std::uint32_t*)q;
  return p;
}

GpuTaskDispatcher::HostColumn::~HostColumn() {
  if (p) {
    if (pinned) ydc_host_free(p); else std::free(p);
  }
}


int GpuTaskDispatcher::UnsafePlace(const RequestSpan& batch) {
  const std::uint32_t n = (std::uint32_t)batch.n;
  std::uint32_t *env = col_env_.ensure(n), *minv = col_minv_.ensure(n), *rip = col_rip_.ensure(n),
                *out = col_out_.ensure(n);
  // (requestor ids first: an address presented for the first time may add table aliases, which
  // the sync below sends along). Consecutive requests of one RPC share the requestor and the
  // digest (scheduler_service_impl.cc:228-264): the last answer is tried first.
  std::string_view last_ip, last_digest;
  std::uint32_t last_ip_id = 0, last_env = 0xFFFFFFFFu, last_name = 0;
  col_digest_name_.resize(n);
  bool have_ip = false, have_digest = false;
  // (a parked request that is being placed again brings what was looked up for it last time)
  const bool cached = batch.pending && batch.cached;
  for (std::uint32_t i = 0; i != n && !cached; ++i) {
    const RequestView& r = batch[i];
    if (!have_ip || r.requestor_ip.data() != last_ip.data() || r.requestor_ip.size() != last_ip.size()) {
      last_ip = r.requestor_ip;
      last_ip_id = RequestorId(last_ip);
      have_ip = true;
    }
    rip[i] = last_ip_id;
  }
  // What reached the registry since the last batch — heartbeat rows, released grants — travels
  // WITH the batch (ydc_dispatch_tick: one launch for a handful of requests, deltas included);
  // a table that has to be uploaded whole, or new address aliases, go first and by themselves.
  const std::uint64_t t0 = NowNs();
  int rc = !ctx_ ? (device_status_ ? device_status_ : YDC_ERR_NO_DEVICE)
                 : (need_full_upload_ || aliases_dirty_ ? UnsafeSyncDevice() : YDC_OK);
  std::uint64_t device_ns = NowNs() - t0;
  if (rc == YDC_OK) {
    // Digests: the same view as the request before (one RPC's requests), else one of the last
    // four distinct string objects of this batch (same address and length within one call: same
    // bytes), else the table.
    struct Seen {
      const char* p = nullptr;
      std::size_t n = 0;
      std::uint32_t env = 0, name = 0;
    } seen[4];
    unsigned next_seen = 0;  // entries of seen[] in use: min(next_seen, 4) — an unused entry matches nothing
    for (std::uint32_t i = 0; i != n && cached; ++i) {
      const Pending& q = *batch.pending[i];
      rip[i] = q.col_rip;
      env[i] = q.col_env;
      col_digest_name_[i] = q.col_name;
      minv[i] = q.request.min_version;
    }
    for (std::uint32_t i = 0; i != n && !cached; ++i) {
      const RequestView& r = batch[i];
      const char* dp = r.compiler_digest.data();
      const std::size_t dn = r.compiler_digest.size();
      if (!(have_digest && dp == last_digest.data() && dn == last_digest.size())) {
        const Seen* hit = nullptr;
        for (unsigned k = 0, used = next_seen < 4 ? next_seen : 4; k != used; ++k)
          if (seen[k].p == dp && seen[k].n == dn) hit = &seen[k];
        if (hit) {
          last_env = hit->env;
          last_name = hit->name;
        } else {
          const EnvEntry* e = LookupEnv(r.compiler_digest);
          last_env = e ? e->bit : 0xFFFFFFFFu;  // unknown: nobody has it
          last_name = e ? e->name : 0;
          seen[next_seen++ & 3] = Seen{dp, dn, last_env, last_name};
        }
        last_digest = r.compiler_digest;
        have_digest = true;
      }
      env[i] = last_env;
      col_digest_name_[i] = last_name;
      minv[i] = r.min_version;
    }
    ydc_task_soa soa{env, minv, rip};
    const std::uint64_t t1 = NowNs();
    std::uint32_t ew = 1;
    if (!dirty_rows_.empty()) UnsafePackDirtyRows(&ew);
    rc = ydc_dispatch_tick(ctx_, dirty_rows_.data(), (const ydc_servant_row*)sync_rows_.data(), sync_env_.data(), ew,
                           (std::uint32_t)dirty_rows_.size(), pending_release_.data(),
                           (std::uint32_t)pending_release_.size(), &soa, n, YDC_DISPATCH_COMMIT, out, nullptr);
    if (rc == YDC_OK) {
      for (auto i : dirty_rows_) row_is_dirty_[i] = 0;
      dirty_rows_.clear();
      pending_release_.clear();
    }
    device_ns += NowNs() - t1;
  }
  host_stats_.device_ns += device_ns;
  host_stats_.requests += n;
  ++host_stats_.batches;
  if (rc != YDC_OK) need_full_upload_ = true;  // the resident running_tasks may be stale now
  return rc;
}

std::uint64_t GpuTaskDispatcher::UnsafeGrant(const RequestView& r, std::uint32_t digest_name,
                                             std::uint32_t servant_index,
                                             std::chrono::nanoseconds expires_in, Clock::time_point now) {
  Servant* pick = servants_[servant_index].get();
  ++pick->running_tasks;  // :123-124 (the device did the same on its column: COMMIT)
  ++pick->ever_assigned_tasks;
  const std::uint64_t task_id = next_task_id_++;  // :127
  Task* t = tasks_.create(task_id);
  t->servant = pick;
  t->started_at = now;
  t->expires_at = now + expires_in;
  t->is_prefetch = r.prefetching;
  t->digest_name = digest_name;  // (granted: some servant advertises the digest, so it has a name)
  const std::size_t ip_len = r.requestor_ip.size();
  const char* ip = r.requestor_ip.data();
  t->ip_len = (std::uint8_t)ip_len;
  if (ip_len >= 8 && ip_len <= 16) {  // (fixed-size overlapping copies: no call into memcpy)
    std::memcpy(t->ip_inline, ip, 8);
    std::memcpy(t->ip_inline + ip_len - 8, ip + ip_len - 8, 8);
  } else if (ip_len < 8) {
    for (std::size_t k = 0; k != ip_len; ++k) t->ip_inline[k] = ip[k];
  } else {
    t->ip_len = 0xFF;
    const std::uint32_t name = names_.intern(r.requestor_ip);
    std::memcpy(t->ip_inline, &name, 4);
  }
  t->next = pick->grants_head;
  if (pick->grants_head != kNoTask) tasks_.slot(pick->grants_head)->prev = task_id;
  pick->grants_head = task_id;
  ++pick->n_grants;
  FileLease(task_id, t->expires_at);
  return task_id;
}

std::string GpuTaskDispatcher::TaskRequestorIp(const Task& t) const {
  if (t.ip_len != 0xFF) return std::string(t.ip_inline, t.ip_len);
  std::uint32_t name;
  std::memcpy(&name, t.ip_inline, 4);
  return names_.name(name);
}

void GpuTaskDispatcher::UnsafeCacheColumns(Pending* r) {
  if (r->cols_epoch == registry_epoch_) return;
  r->col_rip = RequestorId(r->request.requestor_ip);  // (may create an alias: bumps registry_epoch_)
  const EnvEntry* e = LookupEnv(r->request.compiler_digest);
  r->col_env = e ? e->bit : 0xFFFFFFFFu;  // unknown: nobody has it
  r->col_name = e ? e->name : 0;
  if (sig_epoch_ != registry_epoch_) {
    sig_ids_.clear();
    sig_epoch_ = registry_epoch_;
  }
  r->sig = sig_ids_.try_emplace(SigKey{r->col_env, r->request.min_version, r->col_rip},
                                (std::uint32_t)sig_ids_.size()).first->second;
  r->cols_epoch = registry_epoch_;
}

void GpuTaskDispatcher::UnsafePark(Pending* r) {
  if (r->sig >= parked_.size()) parked_.resize((std::size_t)r->sig + 1);
  ParkedList& l = parked_[r->sig];
  r->park_prev = l.tail;
  r->park_next = nullptr;
  if (l.tail) l.tail->park_next = r; else l.head = r;
  l.tail = r;
  r->parked = true;
  ++parked_count_;
}

void GpuTaskDispatcher::UnsafeUnpark(Pending* r) {
  if (!r->parked) return;
  ParkedList& l = parked_[r->sig];
  if (r->park_prev) r->park_prev->park_next = r->park_next; else l.head = r->park_next;
  if (r->park_next) r->park_next->park_prev = r->park_prev; else l.tail = r->park_prev;
  r->park_prev = r->park_next = nullptr;
  r->parked = false;
  --parked_count_;
}

// The interning tables changed (a servant registered, changed its environments or expired): the
// parked requests' triples are numbered anew. Rare, O(parked).
void GpuTaskDispatcher::UnsafeReindexParked() {
  std::vector<Pending*> all;
  all.reserve(parked_count_);
  for (ParkedList& l : parked_)
    for (Pending* r = l.head; r; r = r->park_next) all.push_back(r);
  for (Pending* r : all) {
    r->park_prev = r->park_next = nullptr;
    r->parked = false;
  }
  parked_.clear();
  parked_count_ = 0;
  std::sort(all.begin(), all.end(), [](const Pending* a, const Pending* b) { return a->arrival < b->arrival; });
  for (Pending* r : all) UnsafeCacheColumns(r);
  for (Pending* r : all) UnsafeCacheColumns(r);  // (an alias created on the way invalidated earlier ones)
  for (Pending* r : all) UnsafePark(r);
  parked_epoch_ = registry_epoch_;
}

void GpuTaskDispatcher::UnsafeDispatch(const std::vector<Pending*>& batch) {
  if (batch.empty()) return;
  const std::uint64_t t0 = NowNs(), dev0 = host_stats_.device_ns;
  for (Pending* r : batch) UnsafeCacheColumns(r);
  for (Pending* r : batch) UnsafeCacheColumns(r);  // (an alias created on the way invalidated earlier ones)
  RequestSpan span;
  span.pending = batch.data();
  span.n = batch.size();
  span.cached = true;
  const int rc = UnsafePlace(span);
  if (rc != YDC_OK) {
    // Fail loudly: every request of the batch gets the device error.
    for (auto* r : batch) {
      r->done = true;
      r->result = WaitResult{};
      r->result.device_error = rc;
      r->result.status = WaitStatus::Timeout;
    }
    return;
  }
  const std::uint32_t* out = col_out_.p;
  auto now = Now();
  for (std::size_t i = 0; i != batch.size(); ++i) {
    Pending* r = batch[i];
    ++r->tries;
    if (out[i] == YDC_IDX_ENV_NOT_FOUND) {
      r->done = true;  // :105-108
      r->result.ok = false;
      r->result.status = WaitStatus::EnvironmentNotFound;
      if (oplog_on_) LogWait(r->request, r->expires_in, now, 1, 0, nullptr, r->tries);
    } else if (out[i] == YDC_IDX_TIMEOUT) {
      // (stays pending until its deadline, :116-118)
      if (oplog_on_) LogWait(r->request, r->expires_in, now, 2, 0, nullptr, r->tries);
    } else {
      r->done = true;
      r->result.ok = true;
      r->result.allocation.task_id = UnsafeGrant(r->request, col_digest_name_[i], out[i], r->expires_in, now);
      r->result.allocation.servant_location = servants_[out[i]]->personality.observed_location;
      if (oplog_on_) LogWait(r->request, r->expires_in, now, 0, r->result.allocation.task_id, servants_[out[i]].get(), r->tries);
    }
  }
  host_stats_.host_ns += (NowNs() - t0) - (host_stats_.device_ns - dev0);
}

void GpuTaskDispatcher::UnsafeDrainQueue() {
  // The queued frees and the queued requests are taken in ONE acquisition of queue_lock_, and the
  // frees applied first: a thread that called FreeTask and then WaitForStartingNewTask pushed them
  // in that order, so a turn that places its request has applied its free (the reference's
  // FreeTask is synchronous, task_dispatcher.cc:165-188 — program order must survive the queue).
  std::vector<std::uint64_t> frees;
  std::vector<Pending*> arrivals;
  {
    std::scoped_lock _(queue_lock_);
    if (!free_queue_.empty()) {
      frees.swap(free_queue_);
      free_queued_.store(0, std::memory_order_seq_cst);
    }
    arrivals.assign(queue_.begin(), queue_.end());
    queue_.clear();
    queued_.store(0, std::memory_order_relaxed);
  }
  UnsafeApplyFrees(frees);
  wake_pending_ = false;  // (this turn is the retry)
  // Arrival order: the parked requests (where the waiters were woken since they were last placed:
  // in the reference a waiter sleeps until notify_all or its deadline, and a heartbeat wakes
  // nobody, task_dispatcher.cc:116-118,187,190-220), then what has just come in.
  for (Pending* r : arrivals) UnsafeCacheColumns(r);
  for (Pending* r : arrivals) UnsafeCacheColumns(r);  // (an alias created on the way invalidated earlier ones)
  if (parked_count_ != 0 && parked_epoch_ != registry_epoch_) UnsafeReindexParked();
  if (parked_count_ == 0) parked_epoch_ = registry_epoch_;
  for (Pending* r : arrivals) r->arrival = next_arrival_++;
  const bool retry = parked_count_ != 0 && retried_epoch_ != wake_epoch_;
  retried_epoch_ = wake_epoch_;
  if (!retry && arrivals.empty()) return;
  sig_dead_.assign(sig_ids_.size() + 1, 0);
  std::vector<Pending*> served, seg;
  // Segments: a handful of requests first (one freed slot serves one waiter: 8 requests cost the
  // resident tick 13 us, 64 cost 40), four times as many while whole segments are granted.
  std::size_t cap = 8;
  const auto now = Now();
  // A request that has just come in, found no free servant and whose deadline has passed already
  // (the RPC's "do not wait": timeout == now) is answered here and now — its owner would otherwise
  // have to get hold of the lock only to find that out (task_dispatcher.cc:116-118: wait_until a
  // time that has passed returns at once).
  auto expire_now = [&](Pending* r) {
    if (r->deadline > now) return false;
    r->done = true;
    r->result = WaitResult{};
    r->result.status = WaitStatus::Timeout;
    served.push_back(r);
    return true;
  };
  auto flush = [&] {
    if (seg.empty()) return;
    UnsafeDispatch(seg);
    bool any_dead = false;
    for (Pending* r : seg) {
      if (r->done) {
        UnsafeUnpark(r);
        served.push_back(r);
      } else {  // Timeout: so will every later request of its triple in this turn
        sig_dead_[r->sig] = 1;
        any_dead = true;
        if (!r->parked && !expire_now(r)) UnsafePark(r);
      }
    }
    if (!any_dead) cap *= 4;
    seg.clear();
  };
  // Not sent, the answer is known: for the reference this was one more turn of the waiter's loop
  // that found no free servant (task_dispatcher.cc:109-118).
  auto known_timeout = [&](Pending* r) {
    ++r->tries;
    if (oplog_on_) LogWait(r->request, r->expires_in, now, 2, 0, nullptr, r->tries);
  };
  if (retry) {
    // The heads of the triples' lists, merged by arrival number.
    using Head = std::pair<std::uint64_t, Pending*>;
    std::priority_queue<Head, std::vector<Head>, std::greater<Head>> heads;
    for (ParkedList& l : parked_)
      if (l.head) heads.push({l.head->arrival, l.head});
    while (!heads.empty()) {
      Pending* r = heads.top().second;
      heads.pop();
      if (sig_dead_[r->sig]) {
        // (the rest of this triple's list stays where it is; walked only for the test log)
        if (oplog_on_)
          for (Pending* q = r; q; q = q->park_next) known_timeout(q);
        continue;
      }
      Pending* next = r->park_next;
      seg.push_back(r);
      if (seg.size() >= cap) {
        flush();
        // (the flush may have unparked `next`'s predecessors, never `next` itself: it was not in the segment)
      }
      if (next) heads.push({next->arrival, next});
    }
    // (a segment still open may hold requests whose successors are already in the heap: the heap is
    // empty here, so nothing is pending behind it)
  }
  for (Pending* r : arrivals) {
    if (sig_dead_[r->sig]) {
      known_timeout(r);
      if (!expire_now(r)) UnsafePark(r);
      continue;
    }
    seg.push_back(r);
    if (seg.size() >= cap) flush();
  }
  flush();
  // Requests of other threads may just have been completed by this one: wake exactly their owners
  // (those that have given up spinning sleep until their deadline). A sleeping owner cannot return
  // before this thread lets go of the lock; a spinning one may as soon as `published` is up.
  for (auto* r : served) {
    if (r->sleeping) r->cv.notify_one();
    r->published.store(true, std::memory_order_seq_cst);
  }
  WakeTurnSleepers();
}

// The end of a device turn (or of this thread's service): callers asleep for their answer look
// again. The counter moves first; a caller that is on its way to sleep either sees it moved (the
// futex compares) or is counted in turn_sleepers_ by the time it is read here.
void GpuTaskDispatcher::WakeTurnSleepers() {
  turn_seq_.fetch_add(1, std::memory_order_seq_cst);
  if (turn_sleepers_.load(std::memory_order_seq_cst) != 0) FutexWakeAll(&turn_seq_);
}

WaitResult GpuTaskDispatcher::WaitForStartingNewTask(const TaskPersonality& personality,
                                                     std::chrono::nanoseconds expires_in,
                                                     Clock::time_point timeout, bool prefetching) {
  Pending req;
  req.request = {personality.requestor_ip, personality.compiler_digest, personality.min_version, prefetching};
  req.expires_in = expires_in;
  req.deadline = timeout;
  {
    // Concurrent callers queue up here; whoever holds allocation_lock_ next
    // places all of them as one batch.
    std::scoped_lock _(queue_lock_);
    queue_.push_back(&req);
    queued_.fetch_add(1, std::memory_order_relaxed);
  }
  // Whoever holds allocation_lock_ next places everything that is queued, this request included:
  // its owner spins for the answer (one device turn away) instead of sleeping on the lock, and
  // takes the lock itself — placing everybody else's — when it is free.
  // A short spin (the answer is one device turn away); then sleep until a turn ends — the holder
  // serves the queue on its way out (Section), so a queued request is never stranded: either that
  // exit sees it, or the lock is free when this thread looks.
  std::unique_lock held(allocation_lock_, std::defer_lock);
  for (int spins = 0;;) {
    if (req.published.load(std::memory_order_acquire)) return req.result;
    if (!busy_.load(std::memory_order_relaxed) && held.try_lock()) break;
    if (++spins < options_.caller_spins) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      continue;
    }
    turn_sleepers_.fetch_add(1, std::memory_order_seq_cst);
    const std::uint32_t seen = turn_seq_.load(std::memory_order_seq_cst);
    if (!req.published.load(std::memory_order_seq_cst) && busy_.load(std::memory_order_seq_cst))
      FutexWait(&turn_seq_, seen, 200000);  // (bounded: a turn that never ends — a hung device — must not hide the lock)
    turn_sleepers_.fetch_sub(1, std::memory_order_seq_cst);
    spins = options_.caller_spins - options_.caller_spins / 4;
  }
  Section sec(this, std::move(held));
  std::unique_lock<std::mutex>& lk = sec.lock();
  for (;;) {
    if (!req.done) {
      UnsafeDrainQueue();  // (what comes in during the turn is served on the way out: Section)
    }
    if (req.done) return req.result;
    // About to sleep: say so first, then look for queued FreeTasks once more — a FreeTask that
    // comes later sees the sleeper and takes the lock itself (gpu_task_dispatcher.h: sleepers_).
    sleepers_.fetch_add(1, std::memory_order_seq_cst);
    if (UnsafeApplyQueuedFrees()) {
      sleepers_.fetch_sub(1, std::memory_order_seq_cst);
      continue;  // (waiters were woken: this one tries again, :116-118)
    }
    bool timed_out;
    busy_.store(false, std::memory_order_relaxed);  // (the waits below release the lock)
    req.sleeping = true;
    if (options_.clock) {
      // Injected (test) clock: poll it, do not sleep on the real one.
      timed_out = Now() >= req.deadline;
#if defined(__SANITIZE_THREAD__)
      if (!timed_out) req.cv.wait_until(lk, std::chrono::system_clock::now() + 1ms);
#else
      if (!timed_out) req.cv.wait_for(lk, 1ms);
#endif
    } else {
#if defined(__SANITIZE_THREAD__)
      // GCC 11's libtsan does not intercept pthread_cond_clockwait (what a steady_clock
      // wait compiles to) and then reports the lock as held across the wait; the sanitizer
      // build waits on the system clock (pthread_cond_timedwait) instead.
      timed_out = req.cv.wait_until(lk, std::chrono::system_clock::now() +
                                            (req.deadline - Clock::now())) == std::cv_status::timeout;
#else
      timed_out = req.cv.wait_until(lk, req.deadline) == std::cv_status::timeout;
#endif
    }
    req.sleeping = false;
    busy_.store(true, std::memory_order_relaxed);
    sleepers_.fetch_sub(1, std::memory_order_seq_cst);
    UnsafeApplyQueuedFrees();
    if (req.done) return req.result;  // a concurrent drain served us meanwhile
    if (timed_out) {
      UnsafeUnpark(&req);
      WaitResult r;
      r.status = WaitStatus::Timeout;  // :116-118
      return r;
    }
  }
}

// Places requests[0..n) as one device batch and registers the grants; sink(i, status, task id,
// the granted servant or null) sees every request in order, under the lock. status:
// 0 granted / 1 EnvironmentNotFound / 2 Timeout (timeout == now), or the negative YDC_ERR_* of a
// failed device batch. A sink that returns false gives the grant back at once.
template <class Sink>
int GpuTaskDispatcher::UnsafePlaceAndGrant(std::size_t n, const RequestView* requests,
                                           std::chrono::nanoseconds expires_in, Sink&& sink) {
  UnsafeDrainQueue();  // earlier arrivals first
  if (n == 0) return YDC_OK;
  const std::uint64_t t0 = NowNs(), dev0 = host_stats_.device_ns;
  RequestSpan span;
  span.views = requests;
  span.n = n;
  const int rc = UnsafePlace(span);
  if (rc != YDC_OK) {
    for (std::size_t i = 0; i != n; ++i) sink(i, rc, ~0ull, nullptr);
    return rc;
  }
  const std::uint32_t* out = col_out_.p;
  auto now = Now();
  int worst = YDC_OK;
  for (std::size_t i = 0; i != n; ++i) {
    if (out[i] >= YDC_IDX_ENV_NOT_FOUND) {
      // EnvironmentNotFound (:105-108) / Timeout — timeout == now (:116-118)
      if (oplog_on_) LogWait(requests[i], expires_in, now, out[i] == YDC_IDX_ENV_NOT_FOUND ? 1 : 2, 0, nullptr);
      sink(i, out[i] == YDC_IDX_ENV_NOT_FOUND ? 1 : 2, ~0ull, nullptr);
      continue;
    }
    const std::uint64_t id = UnsafeGrant(requests[i], col_digest_name_[i], out[i], expires_in, now);
    if (oplog_on_) LogWait(requests[i], expires_in, now, 0, id, servants_[out[i]].get());
    if (!sink(i, 0, id, servants_[out[i]].get())) {
      if (oplog_on_) {
        LogOpen(&oplog_, "free");
        LogNum(&oplog_, "id", (long long)id);
        oplog_ += "}";
      }
      UnsafeFreeTasks(&id, 1);
      worst = YDC_ERR_CAPACITY;
    }
  }
  host_stats_.host_ns += (NowNs() - t0) - (host_stats_.device_ns - dev0);
  return worst;
}

int GpuTaskDispatcher::WaitForStartingNewTasksInto(std::size_t n, const RequestView* requests,
                                                   std::chrono::nanoseconds expires_in,
                                                   std::int32_t* out_status, std::uint64_t* out_task_ids,
                                                   char* out_locations, std::size_t location_stride) {
  Section sec(this);
  return UnsafePlaceAndGrant(n, requests, expires_in,
                             [&](std::size_t i, int status, std::uint64_t id, const Servant* pick) {
    char* loc = out_locations && location_stride ? out_locations + i * location_stride : nullptr;
    out_status[i] = status;
    if (out_task_ids) out_task_ids[i] = id;
    if (!loc) return true;
    if (!pick) {
      loc[0] = 0;
      return true;
    }
    if (pick->location_len != 0xFF && location_stride >= sizeof(pick->location_short)) {
      std::memcpy(loc, pick->location_short, sizeof(pick->location_short));  // (fixed size, NUL inside)
      return true;
    }
    const std::string* where = &pick->personality.observed_location;
    if (where->size() < location_stride) {
      std::memcpy(loc, where->c_str(), where->size() + 1);
      return true;
    }
    // A grant whose servant the caller cannot name is useless: it is given back at once
    // instead of leaking the slot until the lease runs out.
    std::memcpy(loc, where->data(), location_stride - 1);
    loc[location_stride - 1] = 0;
    if (out_task_ids) out_task_ids[i] = ~0ull;
    out_status[i] = YDC_ERR_CAPACITY;
    return false;
  });
}

std::vector<WaitResult> GpuTaskDispatcher::WaitForStartingNewTasks(
    const std::vector<TaskPersonality>& personalities, std::chrono::nanoseconds expires_in,
    const std::vector<bool>& prefetching) {
  const std::size_t n = personalities.size();
  std::vector<RequestView> views(n);
  for (std::size_t i = 0; i != n; ++i)
    views[i] = {personalities[i].requestor_ip, personalities[i].compiler_digest,
                personalities[i].min_version, i < prefetching.size() && prefetching[i]};
  std::vector<WaitResult> out(n);
  Section sec(this);
  UnsafePlaceAndGrant(n, views.data(), expires_in,
                      [&](std::size_t i, int status, std::uint64_t id, const Servant* pick) {
    if (status == 0) {
      out[i].ok = true;
      out[i].allocation.task_id = id;
      out[i].allocation.servant_location = pick->personality.observed_location;
    } else if (status < 0) {
      out[i].device_error = status;
      out[i].status = WaitStatus::Timeout;
    } else {
      out[i].status = status == 1 ? WaitStatus::EnvironmentNotFound : WaitStatus::Timeout;
    }
    return true;
  });
  return out;
}

// ---------------------------------------------------------------------------
// DumpInternals (task_dispatcher.cc:538-614): same keys.
// ---------------------------------------------------------------------------
std::string GpuTaskDispatcher::DumpInternals() {
  Section sec(this);
  auto format_time = [this](Clock::time_point tp) {
    // steady -> system clock, like flare::internal::SystemClockView.
    auto sys = std::chrono::system_clock::now() +
               std::chrono::duration_cast<std::chrono::system_clock::duration>(tp - Now());
    std::time_t t = std::chrono::system_clock::to_time_t(sys);
    struct tm buf;
    char out[64] = "";
    if (localtime_r(&t, &buf)) std::strftime(out, sizeof(out), "%Y-%m-%d %H:%M:%S", &buf);
    return std::string(out);
  };
  std::string j = "{";
  std::uint64_t cluster_capacity = 0, capacity_unavailable = 0, total_running = 0;
  j += "\"servants\":[";
  for (std::size_t i = 0; i != servants_.size(); ++i) {
    const Servant& e = *servants_[i];
    const auto& p = e.personality;
    if (i) j += ",";
    j += "{\"version\":" + std::to_string(p.version);
    if (p.observed_location != p.reported_location) {
      j += ",\"observed_location\":";
      JsonEscape(p.observed_location, &j);
      j += ",\"reported_location\":";
      JsonEscape(p.reported_location, &j);
    } else {
      j += ",\"location\":";
      JsonEscape(p.observed_location, &j);
    }
    j += ",\"discovered_at\":";
    JsonEscape(format_time(e.discovered_at), &j);
    j += ",\"expires_at\":";
    JsonEscape(format_time(e.expires_at), &j);
    j += ",\"environments\":[";
    for (std::size_t k = 0; k != p.environments.size(); ++k) {
      if (k) j += ",";
      JsonEscape(p.environments[k], &j);
    }
    j += "],\"priority\":";
    JsonEscape(PriorityName(p.priority), &j);
    if (p.max_tasks) {
      j += ",\"max_tasks\":" + std::to_string(p.max_tasks);
    } else {
      j += ",\"not_accepting_task_reason\":";
      JsonEscape(ReasonName(p.not_accepting_task_reason), &j);
    }
    const std::size_t cap = CapacityAvailable(e);
    j += ",\"num_processors\":" + std::to_string(p.num_processors);
    j += ",\"current_load\":" + std::to_string(p.current_load);
    j += ",\"capacity_available\":" + std::to_string((std::int64_t)cap);
    j += ",\"total_memory_mb\":" + std::to_string(p.total_memory_in_bytes / 1024 / 1024);
    j += ",\"memory_available_mb\":" + std::to_string(p.memory_available_in_bytes / 1024 / 1024);
    j += ",\"running_tasks\":" + std::to_string(e.running_tasks);
    j += ",\"ever_assigned_tasks\":" + std::to_string(e.ever_assigned_tasks) + "}";
    total_running += e.running_tasks;
    cluster_capacity += p.max_tasks;
    capacity_unavailable += p.max_tasks - cap;
  }
  j += "],\"tasks\":{";
  bool first = true;
  tasks_.for_each([&](std::uint64_t k, Task& v) {
    if (!first) j += ",";
    first = false;
    j += "\"" + std::to_string(k) + "\":{\"task_id\":" + std::to_string(k);
    j += ",\"requestor_ip\":";
    JsonEscape(TaskRequestorIp(v), &j);
    j += ",\"compiler_digest\":";
    JsonEscape(names_.name(v.digest_name), &j);
    j += ",\"started_at\":";
    JsonEscape(format_time(v.started_at), &j);
    j += ",\"expires_at\":";
    JsonEscape(format_time(v.expires_at), &j);
    j += std::string(",\"prefetched_task\":") + (v.is_prefetch ? "true" : "false");
    j += ",\"servant_location\":";
    JsonEscape(v.servant->personality.observed_location, &j);
    j += std::string(",\"zombie\":") + (v.zombie ? "true" : "false") + "}";
  });
  j += "}";
  j += ",\"servants_up\":" + std::to_string(servants_.size());
  j += ",\"running_tasks\":" + std::to_string(total_running);
  j += ",\"capacity\":" + std::to_string(cluster_capacity);
  std::int64_t avail = (std::int64_t)(cluster_capacity - total_running - capacity_unavailable);
  j += ",\"capacity_available\":" + std::to_string(std::max<std::int64_t>(avail, 0));
  j += ",\"capacity_unavailable\":" + std::to_string(capacity_unavailable);
  j += ",\"gpu\":{\"device\":" + std::to_string(options_.device) +
       ",\"device_status\":" + std::to_string(device_status_) +
       ",\"environments_interned\":" + std::to_string(env_ids_.size()) +
       ",\"environment_mask_words\":" + std::to_string(EnvWords());
  ydc_stats ds;
  if (ctx_ && ydc_get_stats(ctx_, &ds) == YDC_OK)  // which device path served the calls so far
    j += ",\"tick_resident_calls\":" + std::to_string(ds.tick_resident_calls) +
         ",\"tick_launched_calls\":" + std::to_string(ds.tick_launched_calls) +
         ",\"pipeline_batches\":" + std::to_string(ds.pipeline_batches);
  j += "}";
  j += "}";
  return j;
}

}  // namespace ydc

// gpu_task_dispatcher.cc — see gpu_task_dispatcher.h. Placement goes through the
// C-ABI only (ydc_upload_servants / ydc_update_servants / ydc_release_slots /
// ydc_dispatch); this file never computes a pick itself.
#include "gpu_task_dispatcher.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

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
  std::scoped_lock _(lock_);
  running_tasks_[servant_location] = std::move(tasks);
  flattened_valid_ = false;
}

void RunningTaskBookkeeper::DropServant(const std::string& servant_location) {
  std::scoped_lock _(lock_);
  if (running_tasks_.erase(servant_location)) flattened_valid_ = false;
}

std::vector<RunningTask> RunningTaskBookkeeper::GetRunningTasks() const {
  std::scoped_lock _(lock_);
  if (!flattened_valid_) {
    // Order across servants is unspecified in the reference (hash-map order,
    // running_task_bookkeeper.cc:39-41); order within a servant is kept.
    flattened_.clear();
    for (auto&& [k, v] : running_tasks_) flattened_.insert(flattened_.end(), v.begin(), v.end());
    flattened_valid_ = true;
  }
  return flattened_;
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

std::uint32_t GpuTaskDispatcher::InternIp(const std::string& ip, bool create) {
  auto it = ip_ids_.find(ip);
  if (it != ip_ids_.end()) return it->second;
  if (!create) return 0;  // 0: no servant lives there
  std::uint32_t id = (std::uint32_t)ip_ids_.size() + 1;
  ip_ids_.emplace(ip, id);
  return id;
}

std::uint32_t GpuTaskDispatcher::RequestorId(const std::string& ip) {
  if (!shorter_prefix_refs_.empty()) {
    // Some location answers to `ip` through one of its shorter prefixes: from now on that is an
    // entry of the device's lookup table (next UnsafeSyncDevice).
    auto a = alias_ids_.find(ip);
    if (a != alias_ids_.end()) return a->second;
    if (shorter_prefix_refs_.count(ip)) {
      const std::uint32_t id = InternIp(ip, true);
      alias_ids_.emplace(ip, id);
      aliases_dirty_ = true;
      return id;
    }
  }
  return InternIp(ip, false);
}

std::uint32_t GpuTaskDispatcher::LookupEnv(const std::string& digest) const {
  auto it = env_ids_.find(digest);
  return it == env_ids_.end() ? 0xFFFFFFFFu : it->second.first;  // unknown: nobody has it
}

// Interns the digests a servant advertises: one bit number per digest, as many 64-bit mask
// words as the live digests need (the reference keeps an unbounded vector of
// EnvironmentDesc per servant, task_dispatcher.h:93-94 — no limit here either). Returns
// the bit numbers in listing order; a digest listed twice holds two references, and
// ReleaseEnvBits walks the same list.
std::vector<std::uint32_t> GpuTaskDispatcher::AcquireEnvBits(const std::vector<std::string>& digests) {
  std::vector<std::uint32_t> bits;
  bits.reserve(digests.size());
  for (auto&& d : digests) {
    auto it = env_ids_.find(d);
    if (it == env_ids_.end()) {
      std::uint32_t bit;
      if (!free_env_bits_.empty()) {
        bit = free_env_bits_.back();
        free_env_bits_.pop_back();
      } else {
        bit = next_env_bit_++;
      }
      it = env_ids_.emplace(d, std::make_pair(bit, 0u)).first;
    }
    ++it->second.second;
    bits.push_back(it->second.first);
  }
  return bits;
}

void GpuTaskDispatcher::ReleaseEnvBits(const std::vector<std::string>& digests) {
  for (auto&& d : digests) {
    auto it = env_ids_.find(d);
    if (it == env_ids_.end()) continue;
    if (--it->second.second == 0) {
      free_env_bits_.push_back(it->second.first);
      env_ids_.erase(it);
    }
  }
}

// ---------------------------------------------------------------------------
// Servant maintenance
// ---------------------------------------------------------------------------
void GpuTaskDispatcher::KeepServantAlive(const ServantPersonality& servant,
                                         std::chrono::nanoseconds expires_in) {
  std::scoped_lock _(allocation_lock_);
  auto now = Now();
  auto it = index_of_location_.find(servant.observed_location);
  std::uint32_t idx;
  if (it != index_of_location_.end()) {
    // Renewal: the personality is replaced wholesale, running_tasks,
    // ever_assigned_tasks, discovered_at and the registry position stay
    // (task_dispatcher.cc:195-201).
    idx = it->second;
    Servant* e = servants_[idx].get();
    auto bits = AcquireEnvBits(servant.environments);
    ReleaseEnvBits(e->personality.environments);
    e->personality = servant;
    e->env_bits = std::move(bits);
    e->expires_at = now + expires_in;
  } else {
    idx = (std::uint32_t)servants_.size();
    auto added = std::make_unique<Servant>();
    added->uid = next_servant_uid_++;
    added->personality = servant;
    added->discovered_at = now;
    added->expires_at = now + expires_in;
    added->running_tasks = 0;  // :210
    added->env_bits = AcquireEnvBits(servant.environments);
    auto prefixes = RequestorPrefixes(servant.observed_location);
    // No ':' in the location: an id of its own that no requestor address can map to.
    added->ip_id = !prefixes.empty()
                       ? InternIp(prefixes.front(), true)
                       : InternIp(std::string("\0nohost#", 8) + std::to_string(added->uid), true);
    for (std::size_t k = 1; k < prefixes.size(); ++k) {
      ++shorter_prefix_refs_[prefixes[k]];
      if (alias_ids_.count(prefixes[k])) aliases_dirty_ = true;
      added->shorter_prefixes.push_back(std::move(prefixes[k]));
    }
    index_of_location_.emplace(servant.observed_location, idx);
    index_of_uid_.emplace(added->uid, idx);
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
  std::vector<std::uint64_t> task_grant_ids;
  task_grant_ids.reserve(tasks.size());
  for (auto&& t : tasks) task_grant_ids.push_back(t.task_grant_id);

  std::scoped_lock _(allocation_lock_);
  auto it = index_of_location_.find(servant_location);
  if (it == index_of_location_.end()) return task_grant_ids;  // :241-243
  Servant* servant = servants_[it->second].get();

  UnsafeSweepZombiesOf(servant, {task_grant_ids.begin(), task_grant_ids.end()});

  // Tasks reported by the servant but not (or no longer) granted on it are
  // returned, in report order (:256-273). `grants` is the per-servant index the
  // reference rebuilds with a scan over every task.
  std::vector<std::uint64_t> unknown_tasks;
  std::vector<RunningTask> kept;
  kept.reserve(tasks.size());
  for (auto&& t : tasks) {
    bool permitted = false;
    if (servant->grants.count(t.task_grant_id)) {
      auto ti = tasks_.find(t.task_grant_id);
      permitted = ti != tasks_.end() && !ti->second.zombie;
    }
    if (permitted) {
      kept.push_back(std::move(t));
    } else {
      unknown_tasks.push_back(t.task_grant_id);
    }
  }
  running_task_bookkeeper_.SetServantRunningTasks(servant_location, std::move(kept));
  return unknown_tasks;
}

std::vector<RunningTask> GpuTaskDispatcher::GetRunningTasks() const {
  return running_task_bookkeeper_.GetRunningTasks();
}

// ---------------------------------------------------------------------------
// Leases
// ---------------------------------------------------------------------------
bool GpuTaskDispatcher::KeepTaskAlive(std::uint64_t task_id, std::chrono::nanoseconds new_expires_in) {
  std::scoped_lock _(allocation_lock_);
  auto it = tasks_.find(task_id);
  if (it == tasks_.end()) return false;  // :146-153
  if (it->second.zombie) return false;   // :154-162
  it->second.expires_at = Now() + new_expires_in;
  return true;
}

void GpuTaskDispatcher::FreeTask(std::uint64_t task_id) {
  std::scoped_lock _(allocation_lock_);
  UnsafeFreeTasks({task_id});
}

void GpuTaskDispatcher::UnsafeFreeTasks(const std::vector<std::uint64_t>& task_ids) {
  for (auto id : task_ids) {
    auto it = tasks_.find(id);
    if (it == tasks_.end()) return;  // quirk kept: bails out, no wake-up (:176-180)
    auto si = index_of_uid_.find(it->second.servant_uid);
    if (si != index_of_uid_.end()) {
      Servant* s = servants_[si->second].get();
      --s->running_tasks;  // :181
      s->grants.erase(id);
      if (!need_full_upload_) pending_release_.push_back(si->second);
    }
    tasks_.erase(it);
  }
  ++wake_epoch_;
  allocation_cv_.notify_all();  // :187
}

void GpuTaskDispatcher::UnsafeSweepZombiesOf(Servant* servant,
                                             const std::unordered_set<std::uint64_t>& running) {
  std::vector<std::uint64_t> sweeping;  // :453-476
  for (auto id : servant->grants) {
    auto ti = tasks_.find(id);
    if (ti != tasks_.end() && ti->second.zombie && running.count(id) == 0) sweeping.push_back(id);
  }
  UnsafeFreeTasks(sweeping);
}

void GpuTaskDispatcher::OnExpirationTimer() {
  auto now = Now();
  std::scoped_lock _(allocation_lock_);

  // Expired servants leave the registry; the order of the others is kept
  // because it decides ties (:503-516).
  std::vector<std::uint32_t> expired;
  for (std::uint32_t i = 0; i != servants_.size(); ++i)
    if (servants_[i]->expires_at < now) expired.push_back(i);
  std::vector<std::uint64_t> orphans;
  if (!expired.empty()) {
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
        orphans.insert(orphans.end(), s->grants.begin(), s->grants.end());
        for (auto&& p : s->shorter_prefixes) {
          auto it = shorter_prefix_refs_.find(p);
          if (it != shorter_prefix_refs_.end() && --it->second == 0) shorter_prefix_refs_.erase(it);
        }
        index_of_location_.erase(s->personality.observed_location);
        index_of_uid_.erase(s->uid);
      } else {
        if (w != i) servants_[w] = std::move(servants_[i]);
        ++w;
      }
    }
    servants_.resize(w);
    for (std::uint32_t i = 0; i != servants_.size(); ++i) {
      index_of_location_[servants_[i]->personality.observed_location] = i;
      index_of_uid_[servants_[i]->uid] = i;
    }
    dirty_rows_.clear();
    pending_release_.clear();
    row_is_dirty_.assign(servants_.size(), 0);
    if (!alias_ids_.empty()) aliases_dirty_ = true;  // (rows moved: the device dropped its aliases)
  }
  // UnsafeSweepOrphans (:478-496): tasks of vanished servants are forgotten at
  // once. Their ids are exactly the grant sets of the servants removed above. Called on every
  // tick, like the reference's (:518-520): UnsafeFreeTasks ends in notify_all even for an empty
  // list, so every parked waiter re-runs its loop at least once a second — that is how it gets
  // to see a new servant, a lighter load, or EnvironmentNotFound after the last eligible
  // servant expired (a heartbeat alone wakes nobody, :190-220).
  UnsafeFreeTasks(orphans);

  // Expired leases become zombies; they keep their slot until the servant's
  // next heartbeat no longer lists them (:523-535, task_dispatcher.h:207-214).
  for (auto&& [id, t] : tasks_) {
    if (t.expires_at < now) t.zombie = true;
  }
}

// ---------------------------------------------------------------------------
// Placement
// ---------------------------------------------------------------------------
int GpuTaskDispatcher::UnsafeSyncDevice() {
  if (!ctx_) return device_status_ ? device_status_ : YDC_ERR_NO_DEVICE;
  auto flags_of = [this](const Servant& s) {
    const auto& p = s.personality;
    std::uint32_t f = 0;
    if (p.priority == kServantPriorityDedicated) f |= YDC_SERVANT_DEDICATED;          // :405
    if (p.total_memory_in_bytes != 0 && p.memory_available_in_bytes < min_memory_for_new_task_)
      f |= YDC_SERVANT_LOW_MEMORY;                                                     // :286-287
    return f;
  };
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
    std::sort(dirty_rows_.begin(), dirty_rows_.end());  // appended rows in index order
    std::vector<ydc_servant_row> rows(dirty_rows_.size());
    const std::uint32_t ew = EnvWords();
    std::vector<std::uint64_t> env(rows.size() * ew, 0);
    for (std::size_t k = 0; k != dirty_rows_.size(); ++k) {
      const Servant& s = *servants_[dirty_rows_[k]];
      rows[k].version = (std::uint32_t)s.personality.version;
      rows[k].num_processors = Clamp32(s.personality.num_processors);
      rows[k].current_load = Clamp32(s.personality.current_load);
      rows[k].max_tasks = Clamp32(s.personality.max_tasks);
      rows[k].flags = flags_of(s);
      rows[k].ip_id = s.ip_id;
      for (auto b : s.env_bits) env[k * ew + b / 64] |= 1ull << (b % 64);
      rows[k].env_mask = env[k * ew];
    }
    int rc = ydc_update_servants_wide(ctx_, dirty_rows_.data(), rows.data(), env.data(), ew,
                                      (std::uint32_t)rows.size());
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
      auto a = alias_ids_.find(p);
      if (a != alias_ids_.end()) {
        ids.push_back(a->second);
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

void GpuTaskDispatcher::UnsafeDispatch(const std::vector<Pending*>& batch) {
  if (batch.empty()) return;
  const std::uint32_t n = (std::uint32_t)batch.size();
  std::uint32_t *env = col_env_.ensure(n), *minv = col_minv_.ensure(n), *rip = col_rip_.ensure(n),
                *out = col_out_.ensure(n);
  std::fill(out, out + n, YDC_IDX_TIMEOUT);
  // (requestor ids first: an address presented for the first time may add table aliases, which
  // the sync below sends along)
  for (std::uint32_t i = 0; i != n; ++i) rip[i] = RequestorId(batch[i]->personality->requestor_ip);
  int rc = UnsafeSyncDevice();
  if (rc == YDC_OK) {
    for (std::uint32_t i = 0; i != n; ++i) {
      const TaskPersonality& p = *batch[i]->personality;
      env[i] = LookupEnv(p.compiler_digest);
      minv[i] = p.min_version;
    }
    ydc_task_soa soa{env, minv, rip};
    rc = ydc_dispatch(ctx_, &soa, n, YDC_DISPATCH_COMMIT, out, nullptr, nullptr);
  }
  if (rc != YDC_OK) {
    // Fail loudly: every request of the batch gets the device error. The
    // resident running_tasks may be stale now; rebuild it from the host's.
    need_full_upload_ = true;
    for (auto* r : batch) {
      r->done = true;
      r->result = WaitResult{};
      r->result.device_error = rc;
      r->result.status = WaitStatus::Timeout;
    }
    return;
  }
  auto now = Now();
  for (std::uint32_t i = 0; i != n; ++i) {
    Pending* r = batch[i];
    if (out[i] == YDC_IDX_ENV_NOT_FOUND) {
      r->done = true;  // :105-108
      r->result.ok = false;
      r->result.status = WaitStatus::EnvironmentNotFound;
    } else if (out[i] == YDC_IDX_TIMEOUT) {
      r->tried_epoch = wake_epoch_;  // stays pending until its deadline (:116-118)
    } else {
      Servant* pick = servants_[out[i]].get();
      ++pick->running_tasks;  // :123-124 (the device did the same on its column: COMMIT)
      ++pick->ever_assigned_tasks;
      const std::uint64_t task_id = next_task_id_++;  // :127
      Task& t = tasks_[task_id];
      t.task_id = task_id;
      t.personality = *r->personality;
      t.servant_uid = pick->uid;
      t.started_at = now;
      t.expires_at = now + r->expires_in;
      t.is_prefetch = r->prefetching;
      pick->grants.insert(task_id);
      r->done = true;
      r->result.ok = true;
      r->result.allocation.task_id = task_id;
      r->result.allocation.servant_location = pick->personality.observed_location;
    }
  }
}

void GpuTaskDispatcher::UnsafeDrainQueue() {
  {
    std::scoped_lock _(queue_lock_);
    waiting_.insert(waiting_.end(), queue_.begin(), queue_.end());
    queue_.clear();
  }
  // One device batch, arrival order. A parked request is only retried after FreeTask has
  // woken the waiters (wake_epoch_): in the reference a waiter sleeps until notify_all or
  // its deadline, and a heartbeat wakes nobody (task_dispatcher.cc:116-118,187,190-220).
  std::vector<Pending*> batch;
  for (auto* r : waiting_)
    if (!r->done && r->tried_epoch != wake_epoch_) batch.push_back(r);
  UnsafeDispatch(batch);
  const std::size_t before = waiting_.size();
  waiting_.erase(std::remove_if(waiting_.begin(), waiting_.end(), [](Pending* r) { return r->done; }),
                 waiting_.end());
  // Requests of other threads may just have been completed by this one: wake their owners
  // (they sleep on the condition variable until their deadline otherwise).
  if (waiting_.size() != before) allocation_cv_.notify_all();
}

WaitResult GpuTaskDispatcher::WaitForStartingNewTask(const TaskPersonality& personality,
                                                     std::chrono::nanoseconds expires_in,
                                                     Clock::time_point timeout, bool prefetching) {
  Pending req;
  req.personality = &personality;
  req.expires_in = expires_in;
  req.deadline = timeout;
  req.prefetching = prefetching;
  {
    // Concurrent callers queue up here; whoever holds allocation_lock_ next
    // places all of them as one batch.
    std::scoped_lock _(queue_lock_);
    queue_.push_back(&req);
  }
  std::unique_lock lk(allocation_lock_);
  for (;;) {
    UnsafeDrainQueue();
    if (req.done) return req.result;
    bool timed_out;
    if (options_.clock) {
      // Injected (test) clock: poll it, do not sleep on the real one.
      timed_out = Now() >= req.deadline;
#if defined(__SANITIZE_THREAD__)
      if (!timed_out) allocation_cv_.wait_until(lk, std::chrono::system_clock::now() + 1ms);
#else
      if (!timed_out) allocation_cv_.wait_for(lk, 1ms);
#endif
    } else {
#if defined(__SANITIZE_THREAD__)
      // GCC 11's libtsan does not intercept pthread_cond_clockwait (what a steady_clock
      // wait compiles to) and then reports the lock as held across the wait; the sanitizer
      // build waits on the system clock (pthread_cond_timedwait) instead.
      timed_out = allocation_cv_.wait_until(lk, std::chrono::system_clock::now() +
                                                    (req.deadline - Clock::now())) == std::cv_status::timeout;
#else
      timed_out = allocation_cv_.wait_until(lk, req.deadline) == std::cv_status::timeout;
#endif
    }
    if (req.done) return req.result;  // a concurrent drain served us meanwhile
    if (timed_out) {
      waiting_.erase(std::remove(waiting_.begin(), waiting_.end(), &req), waiting_.end());
      WaitResult r;
      r.status = WaitStatus::Timeout;  // :116-118
      return r;
    }
  }
}

std::vector<WaitResult> GpuTaskDispatcher::WaitForStartingNewTasks(
    const std::vector<TaskPersonality>& personalities, std::chrono::nanoseconds expires_in,
    const std::vector<bool>& prefetching) {
  std::vector<Pending> reqs(personalities.size());
  std::unique_lock lk(allocation_lock_);
  auto now = Now();
  std::vector<Pending*> batch;
  batch.reserve(reqs.size());
  for (std::size_t i = 0; i != reqs.size(); ++i) {
    reqs[i].personality = &personalities[i];
    reqs[i].expires_in = expires_in;
    reqs[i].deadline = now;
    reqs[i].prefetching = i < prefetching.size() && prefetching[i];
    batch.push_back(&reqs[i]);
  }
  UnsafeDrainQueue();  // earlier arrivals first
  UnsafeDispatch(batch);
  std::vector<WaitResult> out(reqs.size());
  for (std::size_t i = 0; i != reqs.size(); ++i) {
    if (reqs[i].done) {
      out[i] = reqs[i].result;
    } else {
      out[i].status = WaitStatus::Timeout;  // timeout == now
    }
  }
  return out;
}

// ---------------------------------------------------------------------------
// DumpInternals (task_dispatcher.cc:538-614): same keys.
// ---------------------------------------------------------------------------
std::string GpuTaskDispatcher::DumpInternals() {
  std::scoped_lock _(allocation_lock_);
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
  for (auto&& [k, v] : tasks_) {
    if (!first) j += ",";
    first = false;
    j += "\"" + std::to_string(k) + "\":{\"task_id\":" + std::to_string(v.task_id);
    j += ",\"requestor_ip\":";
    JsonEscape(v.personality.requestor_ip, &j);
    j += ",\"compiler_digest\":";
    JsonEscape(v.personality.compiler_digest, &j);
    j += ",\"started_at\":";
    JsonEscape(format_time(v.started_at), &j);
    j += ",\"expires_at\":";
    JsonEscape(format_time(v.expires_at), &j);
    j += std::string(",\"prefetched_task\":") + (v.is_prefetch ? "true" : "false");
    auto si = index_of_uid_.find(v.servant_uid);
    j += ",\"servant_location\":";
    JsonEscape(si == index_of_uid_.end() ? std::string() : servants_[si->second]->personality.observed_location, &j);
    j += std::string(",\"zombie\":") + (v.zombie ? "true" : "false") + "}";
  }
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
       ",\"environment_mask_words\":" + std::to_string(EnvWords()) + "}";
  j += "}";
  return j;
}

}  // namespace ydc

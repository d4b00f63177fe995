// tests/native/ydc_stub.cc — TEST TOOL, not product code.
//
// A CPU stand-in for the few ydc_* entry points the host class GpuTaskDispatcher calls
// (include/yadcc_dispatch.h), placed with the CPU model of tests/model (which replays the
// device pipeline through the shared core headers). It exists so that the HOST class —
// locks, request combining, leases, interning, registry deltas — can be built with
// -fsanitize=thread / address and exercised on a box without a GPU (`make tsan`,
// `make asan`, tests/test_task_dispatcher_stub.py). libydc.so never contains this file, and
// the product has no CPU placement: without a device ydc_create fails with YDC_ERR_NO_DEVICE.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/yadcc_dispatch.h"

extern "C" int model_dispatch_alias(uint32_t S, const uint32_t* version, const uint32_t* nproc,
                                    const uint32_t* load, const uint32_t* max_tasks,
                                    const uint32_t* running, const uint32_t* flags,
                                    const uint64_t* env_mask, uint32_t env_words, const uint32_t* ip_id,
                                    uint32_t N, const uint32_t* env_id, const uint32_t* min_version,
                                    const uint32_t* requestor_ip, uint32_t chunk_size, int force_fp64,
                                    uint32_t* out_idx, double* out_util, uint32_t* out_running,
                                    void* stats, uint32_t n_alias, const uint32_t* alias_ip,
                                    const uint32_t* alias_servant);

struct ydc_context {
  uint32_t env_words = 1;
  std::vector<uint32_t> version, nproc, load, max_tasks, running, flags, ip;
  std::vector<uint64_t> env;
  std::vector<uint32_t> alias_ip, alias_servant;
  std::string last_error;
  uint32_t n() const { return (uint32_t)version.size(); }
  void resize(uint32_t m) {
    version.resize(m);
    nproc.resize(m);
    load.resize(m);
    max_tasks.resize(m);
    running.resize(m);
    flags.resize(m);
    ip.resize(m);
    env.resize((size_t)m * env_words);
  }
  void widen(uint32_t words) {
    if (words <= env_words) return;
    std::vector<uint64_t> wide((size_t)n() * words, 0);
    for (uint32_t s = 0; s < n(); ++s)
      for (uint32_t w = 0; w < env_words; ++w) wide[(size_t)s * words + w] = env[(size_t)s * env_words + w];
    env.swap(wide);
    env_words = words;
  }
};

extern "C" {

const char* ydc_strerror(int code) { return code == YDC_OK ? "ok" : "error (CPU stand-in)"; }
const char* ydc_last_error(const ydc_context* c) { return c ? c->last_error.c_str() : ""; }

int ydc_create(int, uint32_t, uint32_t, uint32_t, void*, ydc_context** out) {
  *out = new ydc_context();
  return YDC_OK;
}
int ydc_destroy(ydc_context* c) {
  delete c;
  return YDC_OK;
}

int ydc_upload_servants(ydc_context* c, const ydc_servant_soa* sv, uint32_t n) {
  c->env_words = sv && sv->env_words ? sv->env_words : 1;
  c->alias_ip.clear();
  c->alias_servant.clear();
  c->resize(0);
  c->resize(n);
  for (uint32_t s = 0; s < n; ++s) {
    c->version[s] = sv->version[s];
    c->nproc[s] = sv->num_processors[s];
    c->load[s] = sv->current_load[s];
    c->max_tasks[s] = sv->max_tasks[s];
    c->running[s] = sv->running_tasks[s];
    c->flags[s] = sv->flags[s];
    c->ip[s] = sv->ip_id[s];
  }
  if (n) std::memcpy(c->env.data(), sv->env_mask, (size_t)n * c->env_words * 8);
  return YDC_OK;
}

int ydc_update_servants_wide(ydc_context* c, const uint32_t* idx, const ydc_servant_row* rows,
                             const uint64_t* env_masks, uint32_t env_words, uint32_t n) {
  if (env_masks) c->widen(env_words);
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t s = idx[i];
    if (s > c->n()) return YDC_ERR_INVALID_ARGUMENT;
    if (s == c->n()) c->resize(s + 1);  // appended with running_tasks = 0
    c->version[s] = rows[i].version;
    c->nproc[s] = rows[i].num_processors;
    c->load[s] = rows[i].current_load;
    c->max_tasks[s] = rows[i].max_tasks;
    c->flags[s] = rows[i].flags;
    c->ip[s] = rows[i].ip_id;
    for (uint32_t w = 0; w < c->env_words; ++w)
      c->env[(size_t)s * c->env_words + w] =
          env_masks ? (w < env_words ? env_masks[(size_t)i * env_words + w] : 0)
                    : (w == 0 ? rows[i].env_mask : 0);
  }
  return YDC_OK;
}

int ydc_update_servants(ydc_context* c, const uint32_t* idx, const ydc_servant_row* rows, uint32_t n) {
  return ydc_update_servants_wide(c, idx, rows, nullptr, 1, n);
}

int ydc_remove_servants(ydc_context* c, const uint32_t* idx, uint32_t n) {
  uint32_t w = 0, next = 0;
  const uint32_t S = c->n(), EW = c->env_words;
  for (uint32_t s = 0; s < S; ++s) {
    if (next < n && idx[next] == s) {
      ++next;
      continue;
    }
    c->version[w] = c->version[s];
    c->nproc[w] = c->nproc[s];
    c->load[w] = c->load[s];
    c->max_tasks[w] = c->max_tasks[s];
    c->running[w] = c->running[s];
    c->flags[w] = c->flags[s];
    c->ip[w] = c->ip[s];
    for (uint32_t e = 0; e < EW; ++e) c->env[(size_t)w * EW + e] = c->env[(size_t)s * EW + e];
    ++w;
  }
  if (next != n) return YDC_ERR_INVALID_ARGUMENT;
  c->resize(w);
  c->alias_ip.clear();
  c->alias_servant.clear();
  return YDC_OK;
}

int ydc_set_host_aliases(ydc_context* c, const uint32_t* ip_id, const uint32_t* servant_idx, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i)
    if (servant_idx[i] >= c->n()) return YDC_ERR_INVALID_ARGUMENT;
  c->alias_ip.assign(ip_id, ip_id + n);
  c->alias_servant.assign(servant_idx, servant_idx + n);
  return YDC_OK;
}

int ydc_host_alloc(size_t bytes, void** out) {  // (no device: plain memory)
  *out = std::malloc(bytes ? bytes : 1);
  return *out ? YDC_OK : YDC_ERR_HIP;
}
int ydc_host_free(void* p) {
  std::free(p);
  return YDC_OK;
}

int ydc_release_slots(ydc_context* c, const uint32_t* servant_idx, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i)
    if (servant_idx[i] < c->n()) c->running[servant_idx[i]] -= 1;
  return YDC_OK;
}

int ydc_dispatch(ydc_context* c, const ydc_task_soa* tk, uint32_t N, uint32_t flags,
                 uint32_t* out_idx, double* out_util, uint32_t* out_running) {
  // Host-side profiling (tools/td_native_bench_stub): no placement at all — request i goes to
  // servant i mod S — so that what is left on the clock is the host class itself.
  static const bool round_robin = std::getenv("YDC_STUB_ROUND_ROBIN") != nullptr;
  if (round_robin && c->n()) {
    for (uint32_t i = 0; i < N; ++i) out_idx[i] = i % c->n();
    return YDC_OK;
  }
  std::vector<uint32_t> run(c->n());
  int rc = model_dispatch_alias(c->n(), c->version.data(), c->nproc.data(), c->load.data(),
                                c->max_tasks.data(), c->running.data(), c->flags.data(), c->env.data(),
                                c->env_words, c->ip.data(), N, tk ? tk->env_id : nullptr,
                                tk ? tk->min_version : nullptr, tk ? tk->requestor_ip : nullptr, 256, 0,
                                out_idx, out_util, run.data(), nullptr, (uint32_t)c->alias_ip.size(),
                                c->alias_ip.data(), c->alias_servant.data());
  if (rc) return YDC_ERR_NOT_CONVERGED;
  if (out_running) std::copy(run.begin(), run.end(), out_running);
  if (flags & YDC_DISPATCH_COMMIT) c->running = run;
  return YDC_OK;
}

int ydc_dispatch_tick(ydc_context* c, const uint32_t* upd_idx, const ydc_servant_row* upd_rows,
                      const uint64_t* upd_env_masks, uint32_t env_words, uint32_t n_upd,
                      const uint32_t* release_servant_idx, uint32_t n_rel, const ydc_task_soa* tasks,
                      uint32_t n_tasks, uint32_t flags, uint32_t* out_servant_idx, double* out_utilization) {
  if (n_upd)
    if (int rc = ydc_update_servants_wide(c, upd_idx, upd_rows, upd_env_masks, env_words, n_upd)) return rc;
  if (n_rel)
    if (int rc = ydc_release_slots(c, release_servant_idx, n_rel)) return rc;
  if (!n_tasks) return YDC_OK;
  return ydc_dispatch(c, tasks, n_tasks, flags, out_servant_idx, out_utilization, nullptr);
}

}  // extern "C"

extern "C" int ydc_get_stats(const ydc_context* c, ydc_stats* out) {  // (no device paths to tell apart here)
  if (!c || !out) return YDC_ERR_INVALID_ARGUMENT;
  std::memset(out, 0, sizeof *out);
  return YDC_OK;
}

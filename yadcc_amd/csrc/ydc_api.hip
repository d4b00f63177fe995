// ydc_api.hip — the extern "C" boundary (include/yadcc_dispatch.h): context,
// resident servant registry, per-batch launch sequence. gfx950 only; there is
// no CPU fallback — without a device every call fails with YDC_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/yadcc_dispatch.h"
#include "dispatch_core.h"
#include "host_tables.h"
#include "kernels.h"

using namespace ydc;

namespace {

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = std::max<size_t>(n, 16);
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct ydc_context {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  uint32_t max_servants = 0, max_tasks = 0, max_slots = 0;
  std::string last_error;

  // Host mirror of the registry columns the derived tables need.
  uint32_t n_servants = 0;
  std::vector<uint32_t> h_version, h_nproc, h_load, h_max_tasks, h_flags, h_ip;
  std::vector<uint64_t> h_env;
  HostTables tables;
  KeyFormat kf{};
  bool tables_dirty = true;

  // Resident registry.
  DevBuf<uint32_t> d_version, d_nproc, d_load, d_max_tasks, d_running, d_flags, d_class_of;
  DevBuf<uint32_t> d_ip_sorted, d_ip_servant, d_cls_ver;
  DevBuf<uint64_t> d_cls_env;

  // Per-batch workspace.
  DevBuf<uint32_t> d_slot_base, d_cls_begin, d_vals[2], d_hist, d_row_total;
  DevBuf<uint64_t> d_keys[2];  // viewed as u32 when the key fits
  DevBuf<uint16_t> d_cls_by_g;
  DevBuf<uint64_t> d_mask;
  DevBuf<uint32_t> d_self_lo, d_self_hi, d_chunk_consuming, d_before, d_slot_of, d_left;
  DevBuf<uint32_t> d_running_out;
  DevBuf<ClassState> d_guess[2], d_endst, d_checkpoint;
  DevBuf<uint32_t> d_claim;
  uint32_t pass_stamp = 0;  // unique id of every matching pass launched (claims)
  uint32_t round_hint = 3;  // passes to pre-launch before looking at the outcome
  DevBuf<ClassRun> d_runs;
  DevBuf<uint8_t> d_dirty;
  DevBuf<uint64_t> d_dbg;
  bool debug_sim = false;
  DevBuf<DeviceParams> d_prm;
  DeviceParams* h_prm = nullptr;  // pinned

  // Staging for the host-pointer entry point.
  DevBuf<uint32_t> d_t_env, d_t_minv, d_t_ip, d_out_idx, d_upd_idx;
  DevBuf<double> d_out_util;

  uint32_t opt_chunk_size = 0;     // 0: automatic
  uint32_t opt_target_chunks = 2048;
  uint32_t opt_rounds_per_check = 2;
  bool profiling = false;
  hipEvent_t ev[YDC_STAGE_COUNT + 1] = {};
  ydc_stats stats{};

  // Per-kernel timing (profiling only): one event pair per launch.
  struct KernelSample {
    const char* name;
    hipEvent_t a, b;
  };
  std::vector<KernelSample> ksamples;
  size_t ksamples_used = 0;
  std::string kprofile_json;
};

namespace {

std::string g_create_error;  // errors raised before a context exists

int fail(ydc_context* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ctx->last_error = buf;
  }
  return code;
}

#define HIP_TRY(ctx, expr)                                                                   \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess)                                                                   \
      return fail(ctx, YDC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                  __FILE__, __LINE__);                                                       \
  } while (0)

inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

int rebuild_tables(ydc_context* c) {
  const uint32_t n = c->n_servants;
  c->tables.build(n, c->h_env.data(), c->h_version.data(), c->h_max_tasks.data(),
                  c->h_nproc.data(), c->h_ip.data());
  c->kf = choose_key_format(c->tables.cap_bits, kRadixBits);
  const uint32_t C = c->tables.n_classes();
  if (C > 65535) return fail(c, YDC_ERR_TOO_MANY_CLASSES, "%u servant classes", C);
  HIP_TRY(c, c->d_class_of.reserve(n));
  HIP_TRY(c, c->d_ip_sorted.reserve(n));
  HIP_TRY(c, c->d_ip_servant.reserve(n));
  HIP_TRY(c, c->d_cls_env.reserve(C));
  HIP_TRY(c, c->d_cls_ver.reserve(C));
  HIP_TRY(c, c->d_cls_begin.reserve(C + 1));
  if (n) {
    HIP_TRY(c, hipMemcpyAsync(c->d_class_of.p, c->tables.class_of.data(), n * 4,
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_ip_sorted.p, c->tables.ip_sorted.data(), n * 4,
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_ip_servant.p, c->tables.ip_servant.data(), n * 4,
                              hipMemcpyHostToDevice, c->stream));
  }
  if (C) {
    HIP_TRY(c, hipMemcpyAsync(c->d_cls_env.p, c->tables.cls_env.data(), C * 8,
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_cls_ver.p, c->tables.cls_ver.data(), C * 4,
                              hipMemcpyHostToDevice, c->stream));
  }
  // The host vectors are pageable: the copies above are complete on return only
  // after a sync (the tables may be rebuilt before the next launch otherwise).
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->tables_dirty = false;
  return YDC_OK;
}

int reserve_registry(ydc_context* c, uint32_t n) {
  HIP_TRY(c, c->d_version.reserve(n));
  HIP_TRY(c, c->d_nproc.reserve(n));
  HIP_TRY(c, c->d_load.reserve(n));
  HIP_TRY(c, c->d_max_tasks.reserve(n));
  HIP_TRY(c, c->d_running.reserve(n));
  HIP_TRY(c, c->d_flags.reserve(n));
  HIP_TRY(c, c->d_running_out.reserve(n));
  HIP_TRY(c, c->d_slot_base.reserve((size_t)n + 1));
  HIP_TRY(c, c->d_left.reserve(n));
  return YDC_OK;
}

void mark(ydc_context* c, int stage) {
  if (c->profiling) (void)hipEventRecord(c->ev[stage], c->stream);
}

// Brackets one kernel launch with events when profiling is on.
struct KernelTimer {
  ydc_context* c;
  ydc_context::KernelSample* s = nullptr;
  KernelTimer(ydc_context* ctx, const char* name) : c(ctx) {
    if (!c->profiling) return;
    if (c->ksamples_used == c->ksamples.size()) {
      ydc_context::KernelSample n{name, nullptr, nullptr};
      if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return;
      c->ksamples.push_back(n);
    }
    s = &c->ksamples[c->ksamples_used++];
    s->name = name;
    (void)hipEventRecord(s->a, c->stream);
  }
  ~KernelTimer() {
    if (s) (void)hipEventRecord(s->b, c->stream);
  }
};
#define YDC_LAUNCH(ctx, name, ...)            \
  do {                                        \
    KernelTimer kt__(ctx, name);              \
    hipLaunchKernelGGL(__VA_ARGS__);          \
  } while (0)

template <typename KeyT>
int launch_sort_pass(ydc_context* c, const SortIn<KeyT>& in, uint32_t n_tiles, void* out_keys,
                     bool out_u32, uint32_t* out_vals) {
  YDC_LAUNCH(c, "k_radix_hist", k_radix_hist<KeyT>, dim3(n_tiles), dim3(kSortThreads), 0,
             c->stream, in, c->d_prm.p, n_tiles, c->d_hist.p);
  YDC_LAUNCH(c, "k_radix_scan", k_radix_scan, dim3(kRadix), dim3(256), 0, c->stream, n_tiles,
             c->d_hist.p, c->d_row_total.p);
  if (out_u32) {
    YDC_LAUNCH(c, "k_radix_scatter", (k_radix_scatter<KeyT, uint32_t>), dim3(n_tiles),
               dim3(kSortThreads), 0, c->stream, in, c->d_prm.p, n_tiles, c->d_hist.p,
               c->d_row_total.p, (uint32_t*)out_keys, out_vals);
  } else {
    YDC_LAUNCH(c, "k_radix_scatter", (k_radix_scatter<KeyT, uint64_t>), dim3(n_tiles),
               dim3(kSortThreads), 0, c->stream, in, c->d_prm.p, n_tiles, c->d_hist.p,
               c->d_row_total.p, (uint64_t*)out_keys, out_vals);
  }
  return YDC_OK;
}

}  // namespace

extern "C" {

const char* ydc_strerror(int code) {
  switch (code) {
    case YDC_OK: return "ok";
    case YDC_ERR_INVALID_ARGUMENT: return "invalid argument";
    case YDC_ERR_HIP: return "HIP runtime error";
    case YDC_ERR_NO_DEVICE: return "no usable gfx950 device (there is no CPU fallback)";
    case YDC_ERR_CAPACITY: return "capacity of the context exceeded";
    case YDC_ERR_TOO_MANY_CLASSES: return "too many servant classes";
    case YDC_ERR_NOT_CONVERGED: return "chunk states did not converge";
    default: return "unknown error";
  }
}

const char* ydc_last_error(const ydc_context* ctx) {
  return ctx ? ctx->last_error.c_str() : g_create_error.c_str();
}

int ydc_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    return 0;
  }
  return n;
}

int ydc_device_malloc(int device, size_t bytes, void** out) {
  if (!out) return YDC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (hipSetDevice(device) != hipSuccess) return YDC_ERR_NO_DEVICE;
  hipError_t e = hipMalloc(out, bytes ? bytes : 1);
  if (e != hipSuccess) {
    g_create_error = std::string("hipMalloc: ") + hipGetErrorString(e);
    return YDC_ERR_HIP;
  }
  return YDC_OK;
}

int ydc_device_free(void* p) { return hipFree(p) == hipSuccess ? YDC_OK : YDC_ERR_HIP; }

int ydc_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? YDC_OK : YDC_ERR_HIP;
}

int ydc_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? YDC_OK : YDC_ERR_HIP;
}
uint32_t ydc_abi_version(void) { return 1; }

int ydc_create(int device, uint32_t max_servants, uint32_t max_tasks, uint32_t max_slots,
               void* stream, ydc_context** out) {
  if (!out) return YDC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int n_dev = 0;
  hipError_t de = hipGetDeviceCount(&n_dev);
  if (de != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
    g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(de) + ", " +
                     std::to_string(n_dev) + " device(s), asked for " + std::to_string(device);
    return YDC_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) return YDC_ERR_NO_DEVICE;
  auto* c = new ydc_context();
  c->device = device;
  c->max_servants = max_servants;
  c->max_tasks = max_tasks;
  c->max_slots = max_slots;
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return YDC_ERR_HIP;
    }
    c->own_stream = true;
  }
  if (c->d_prm.reserve(1) != hipSuccess ||
      hipHostMalloc((void**)&c->h_prm, sizeof(DeviceParams)) != hipSuccess ||
      c->d_row_total.reserve(kRadix) != hipSuccess) {
    ydc_destroy(c);
    return YDC_ERR_HIP;
  }
  for (auto& e : c->ev) {
    if (hipEventCreate(&e) != hipSuccess) {
      ydc_destroy(c);
      return YDC_ERR_HIP;
    }
  }
  if (const char* s = getenv("YDC_DEBUG_SIM")) c->debug_sim = atoi(s) != 0;
  if (const char* s = getenv("YDC_CHUNK_SIZE")) c->opt_chunk_size = (uint32_t)atoi(s);
  if (const char* s = getenv("YDC_TARGET_CHUNKS")) c->opt_target_chunks = (uint32_t)atoi(s);
  if (const char* s = getenv("YDC_ROUNDS_PER_CHECK"))
    c->opt_rounds_per_check = std::max(1, atoi(s));
  *out = c;
  return YDC_OK;
}

int ydc_destroy(ydc_context* c) {
  if (!c) return YDC_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto* b : {&c->d_version, &c->d_nproc, &c->d_load, &c->d_max_tasks, &c->d_running,
                  &c->d_flags, &c->d_class_of, &c->d_ip_sorted, &c->d_ip_servant, &c->d_cls_ver,
                  &c->d_slot_base, &c->d_cls_begin, &c->d_vals[0], &c->d_vals[1], &c->d_hist,
                  &c->d_row_total, &c->d_self_lo, &c->d_self_hi, &c->d_chunk_consuming,
                  &c->d_before, &c->d_slot_of, &c->d_left, &c->d_running_out, &c->d_t_env,
                  &c->d_t_minv, &c->d_t_ip, &c->d_out_idx, &c->d_upd_idx})
    b->release();
  for (auto* b : {&c->d_cls_env, &c->d_keys[0], &c->d_keys[1], &c->d_mask}) b->release();
  c->d_cls_by_g.release();
  c->d_guess[0].release();
  c->d_guess[1].release();
  c->d_endst.release();
  c->d_checkpoint.release();
  c->d_claim.release();
  c->d_runs.release();
  c->d_dirty.release();
  c->d_dbg.release();
  c->d_prm.release();
  c->d_out_util.release();
  if (c->h_prm) (void)hipHostFree(c->h_prm);
  for (auto& e : c->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& k : c->ksamples) {
    if (k.a) (void)hipEventDestroy(k.a);
    if (k.b) (void)hipEventDestroy(k.b);
  }
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return YDC_OK;
}

int ydc_upload_servants(ydc_context* c, const ydc_servant_soa* sv, uint32_t n) {
  if (!c || (n && !sv)) return YDC_ERR_INVALID_ARGUMENT;
  if (c->max_servants && n > c->max_servants)
    return fail(c, YDC_ERR_CAPACITY, "%u servants > max_servants %u", n, c->max_servants);
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = reserve_registry(c, n)) return rc;
  c->n_servants = n;
  c->h_version.assign(sv ? sv->version : nullptr, sv ? sv->version + n : nullptr);
  c->h_nproc.assign(sv ? sv->num_processors : nullptr, sv ? sv->num_processors + n : nullptr);
  c->h_load.assign(sv ? sv->current_load : nullptr, sv ? sv->current_load + n : nullptr);
  c->h_max_tasks.assign(sv ? sv->max_tasks : nullptr, sv ? sv->max_tasks + n : nullptr);
  c->h_flags.assign(sv ? sv->flags : nullptr, sv ? sv->flags + n : nullptr);
  c->h_ip.assign(sv ? sv->ip_id : nullptr, sv ? sv->ip_id + n : nullptr);
  c->h_env.assign(sv ? sv->env_mask : nullptr, sv ? sv->env_mask + n : nullptr);
  if (n) {
    HIP_TRY(c, hipMemcpyAsync(c->d_version.p, sv->version, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_nproc.p, sv->num_processors, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_load.p, sv->current_load, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_max_tasks.p, sv->max_tasks, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_running.p, sv->running_tasks, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_flags.p, sv->flags, n * 4, hipMemcpyHostToDevice, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return rebuild_tables(c);
}

int ydc_update_servants(ydc_context* c, const uint32_t* idx, const ydc_servant_row* rows,
                        uint32_t n) {
  if (!c || (n && (!idx || !rows))) return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  // Appends first (they may need bigger buffers).
  uint32_t new_n = c->n_servants;
  for (uint32_t i = 0; i < n; ++i) {
    if (idx[i] > new_n) return fail(c, YDC_ERR_INVALID_ARGUMENT, "servant index %u out of order", idx[i]);
    if (idx[i] == new_n) ++new_n;
  }
  if (c->max_servants && new_n > c->max_servants)
    return fail(c, YDC_ERR_CAPACITY, "%u servants > max_servants %u", new_n, c->max_servants);
  if (new_n > c->d_version.cap) {
    // Grow: read the running column back, reallocate, re-upload everything.
    std::vector<uint32_t> run(c->n_servants);
    if (c->n_servants)
      HIP_TRY(c, hipMemcpy(run.data(), c->d_running.p, c->n_servants * 4, hipMemcpyDeviceToHost));
    run.resize(new_n, 0);
    uint32_t old_n = c->n_servants;
    (void)old_n;
    if (int rc = reserve_registry(c, std::max<uint32_t>(new_n, new_n + new_n / 2))) return rc;
    c->h_version.resize(new_n);
    c->h_nproc.resize(new_n);
    c->h_load.resize(new_n);
    c->h_max_tasks.resize(new_n);
    c->h_flags.resize(new_n);
    c->h_ip.resize(new_n);
    c->h_env.resize(new_n);
    HIP_TRY(c, hipMemcpy(c->d_running.p, run.data(), new_n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_version.p, c->h_version.data(), new_n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_nproc.p, c->h_nproc.data(), new_n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_load.p, c->h_load.data(), new_n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_max_tasks.p, c->h_max_tasks.data(), new_n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_flags.p, c->h_flags.data(), new_n * 4, hipMemcpyHostToDevice));
  } else if (new_n > c->n_servants) {
    c->h_version.resize(new_n);
    c->h_nproc.resize(new_n);
    c->h_load.resize(new_n);
    c->h_max_tasks.resize(new_n);
    c->h_flags.resize(new_n);
    c->h_ip.resize(new_n);
    c->h_env.resize(new_n);
    HIP_TRY(c, hipMemsetAsync(c->d_running.p + c->n_servants, 0, (new_n - c->n_servants) * 4, c->stream));
  }
  c->n_servants = new_n;
  bool structural = false;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t s = idx[i];
    const ydc_servant_row& r = rows[i];
    structural |= c->h_version[s] != r.version || c->h_env[s] != r.env_mask ||
                  c->h_ip[s] != r.ip_id || (c->h_max_tasks[s] == 0) != (r.max_tasks == 0) ||
                  std::min(c->h_max_tasks[s], c->h_nproc[s]) != std::min(r.max_tasks, r.num_processors);
    c->h_version[s] = r.version;
    c->h_nproc[s] = r.num_processors;
    c->h_load[s] = r.current_load;
    c->h_max_tasks[s] = r.max_tasks;
    c->h_flags[s] = r.flags;
    c->h_ip[s] = r.ip_id;
    c->h_env[s] = r.env_mask;
    // Small scattered writes; rows are contiguous per column on the device.
    HIP_TRY(c, hipMemcpyAsync(c->d_version.p + s, &c->h_version[s], 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_nproc.p + s, &c->h_nproc[s], 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_load.p + s, &c->h_load[s], 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_max_tasks.p + s, &c->h_max_tasks[s], 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_flags.p + s, &c->h_flags[s], 4, hipMemcpyHostToDevice, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (structural || c->tables_dirty) return rebuild_tables(c);
  return YDC_OK;
}

int ydc_release_slots(ydc_context* c, const uint32_t* servant_idx, uint32_t n) {
  if (!c || (n && !servant_idx)) return YDC_ERR_INVALID_ARGUMENT;
  if (!n) return YDC_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, c->d_upd_idx.reserve(n));
  HIP_TRY(c, hipMemcpyAsync(c->d_upd_idx.p, servant_idx, n * 4, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_release_slots, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream,
                     c->d_upd_idx.p, n, c->n_servants, c->d_running.p);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // servant_idx is pageable host memory
  return YDC_OK;
}

int ydc_set_running(ydc_context* c, const uint32_t* running, uint32_t n) {
  if (!c || n != c->n_servants || (n && !running)) return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  if (n) HIP_TRY(c, hipMemcpy(c->d_running.p, running, n * 4, hipMemcpyHostToDevice));
  return YDC_OK;
}

int ydc_get_running(ydc_context* c, uint32_t* out, uint32_t n) {
  if (!c || n != c->n_servants || (n && !out)) return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (n) HIP_TRY(c, hipMemcpy(out, c->d_running.p, n * 4, hipMemcpyDeviceToHost));
  return YDC_OK;
}

int ydc_dispatch_device(ydc_context* c, const ydc_task_soa* tk, uint32_t N, uint32_t flags,
                        uint32_t* d_out_idx, double* d_out_util, uint32_t* d_out_running) {
  if (!c || (N && !tk)) return YDC_ERR_INVALID_ARGUMENT;
  if (c->max_tasks && N > c->max_tasks)
    return fail(c, YDC_ERR_CAPACITY, "%u tasks > max_tasks %u", N, c->max_tasks);
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->tables_dirty)
    if (int rc = rebuild_tables(c)) return rc;
  const uint32_t S = c->n_servants;
  const uint32_t C = c->tables.n_classes();
  const uint32_t W = std::max<uint32_t>(1, ceil_div(C, 64));
  const uint64_t slot_bound64 = c->tables.max_slots;
  if (slot_bound64 > 0xFFFFFFF0ull || (c->max_slots && slot_bound64 > c->max_slots))
    return fail(c, YDC_ERR_CAPACITY, "registry can offer %llu slots > max_slots %u",
                (unsigned long long)slot_bound64, c->max_slots);
  const uint32_t slot_bound = (uint32_t)slot_bound64;
  const uint32_t n_tiles = std::max<uint32_t>(1, ceil_div(slot_bound, kSortTile));
  const bool any_shared = c->tables.any_shared_ip;
  const bool use_generic = C > kMaxWaveClasses;

  uint32_t cs = c->opt_chunk_size;
  if (cs == 0) {
    uint32_t want = ceil_div(std::max<uint32_t>(N, 1), std::max<uint32_t>(1, c->opt_target_chunks));
    cs = 64;
    while (cs < want && cs < 8192) cs <<= 1;
  }
  const uint32_t K = N ? ceil_div(N, cs) : 0;

  // Workspace.
  const bool key32 = c->kf.key_bits <= 32;
  HIP_TRY(c, c->d_keys[0].reserve(key32 ? (slot_bound + 1) / 2 : slot_bound));
  HIP_TRY(c, c->d_keys[1].reserve(key32 ? (slot_bound + 1) / 2 : slot_bound));
  HIP_TRY(c, c->d_vals[0].reserve(slot_bound));
  HIP_TRY(c, c->d_vals[1].reserve(slot_bound));
  HIP_TRY(c, c->d_hist.reserve((size_t)kRadix * n_tiles));
  if (C > 1) HIP_TRY(c, c->d_cls_by_g.reserve(slot_bound));
  HIP_TRY(c, c->d_mask.reserve((size_t)N * W));
  HIP_TRY(c, c->d_self_lo.reserve(N));
  HIP_TRY(c, c->d_self_hi.reserve(N));
  HIP_TRY(c, c->d_slot_of.reserve(N));
  HIP_TRY(c, c->d_chunk_consuming.reserve((size_t)K + 1));
  HIP_TRY(c, c->d_before.reserve((size_t)K + 1));
  HIP_TRY(c, c->d_dirty.reserve((size_t)K + 1));
  HIP_TRY(c, c->d_guess[0].reserve((size_t)K * C + 1));
  HIP_TRY(c, c->d_endst.reserve((size_t)K * C + 1));
  if (!use_generic) {
    HIP_TRY(c, c->d_checkpoint.reserve((size_t)ceil_div(std::max(N, 1u), 64) * C + 1));
    if ((size_t)K + 1 > c->d_claim.cap || c->pass_stamp > 0xFFFF0000u) {
      // Claims compare against ever-growing pass stamps: a fresh (or wrapped) array starts at 0.
      HIP_TRY(c, c->d_claim.reserve((size_t)K + 1));
      HIP_TRY(c, hipMemsetAsync(c->d_claim.p, 0, c->d_claim.cap * 4, c->stream));
      c->pass_stamp = 0;
    }
  }
  if (use_generic) HIP_TRY(c, c->d_runs.reserve((size_t)K * C + 1));
  else if (any_shared) HIP_TRY(c, c->d_runs.reserve((size_t)C + 1));

  ServantTable sv{c->d_version.p, c->d_nproc.p,  c->d_load.p,     c->d_max_tasks.p,
                  c->d_running.p, c->d_flags.p, c->d_class_of.p, S};
  DeviceParams* prm = c->d_prm.p;
  hipStream_t st = c->stream;

  c->ksamples_used = 0;
  mark(c, 0);
  // ---- servant scan
  YDC_LAUNCH(c, "k_servant_scan", k_servant_scan, dim3(1), dim3(1024), (C + 1) * sizeof(uint32_t), st, sv, C,
                     slot_bound, c->d_slot_base.p, c->d_cls_begin.p, prm);
  mark(c, 1);
  // ---- slot generation
  void* keys[2] = {c->d_keys[0].p, c->d_keys[1].p};
  uint32_t* vals[2] = {c->d_vals[0].p, c->d_vals[1].p};
  int cur = 0;
  if (slot_bound) {
    const uint32_t gen_blocks = ceil_div(slot_bound, 256);
    if (key32) {
      YDC_LAUNCH(c, "k_slot_gen", k_slot_gen<uint32_t>, dim3(gen_blocks), dim3(256), 0, st, sv,
                         c->d_slot_base.p, prm, (uint32_t)c->kf.exact, c->kf.cap_bits,
                         (uint32_t*)keys[0], vals[0], C > 1 ? c->d_cls_by_g.p : nullptr);
    } else {
      YDC_LAUNCH(c, "k_slot_gen", k_slot_gen<uint64_t>, dim3(gen_blocks), dim3(256), 0, st, sv,
                         c->d_slot_base.p, prm, (uint32_t)c->kf.exact, c->kf.cap_bits,
                         (uint64_t*)keys[0], vals[0], C > 1 ? c->d_cls_by_g.p : nullptr);
    }
  }
  mark(c, 2);
  // ---- sort by key
  const uint32_t key_passes = ceil_div(c->kf.key_bits, kRadixBits);
  if (slot_bound) {
    for (uint32_t p = 0; p < key_passes; ++p) {
      if (key32) {
        SortIn<uint32_t> in{(const uint32_t*)keys[cur], vals[cur], nullptr, p * kRadixBits};
        launch_sort_pass(c, in, n_tiles, keys[cur ^ 1], true, vals[cur ^ 1]);
      } else {
        SortIn<uint64_t> in{(const uint64_t*)keys[cur], vals[cur], nullptr, p * kRadixBits};
        launch_sort_pass(c, in, n_tiles, keys[cur ^ 1], false, vals[cur ^ 1]);
      }
      cur ^= 1;
    }
  }
  mark(c, 3);
  // ---- class lists: stable partition of ranks by class
  ClassLists L;
  L.n_classes = C;
  L.cls_begin = c->d_cls_begin.p;
  L.list_p = nullptr;
  L.list_g = vals[cur];
  if (C > 1 && slot_bound) {
    uint32_t cls_bits = 1;
    while ((1u << cls_bits) < C) ++cls_bits;
    const uint32_t cls_passes = ceil_div(cls_bits, kRadixBits);
    for (uint32_t p = 0; p < cls_passes; ++p) {
      // First pass: key == index (global rank). Later passes carry the rank along.
      SortIn<uint32_t> in{p == 0 ? nullptr : (const uint32_t*)keys[cur], vals[cur],
                          c->d_cls_by_g.p, p * kRadixBits};
      launch_sort_pass(c, in, n_tiles, keys[cur ^ 1], true, vals[cur ^ 1]);
      cur ^= 1;
    }
    L.list_p = (const uint32_t*)keys[cur];
    L.list_g = vals[cur];
  }
  mark(c, 4);
  // ---- task classification
  TaskTable T{c->d_mask.p, c->d_self_lo.p, c->d_self_hi.p, W};
  if (N) {
    HIP_TRY(c, hipMemsetAsync(c->d_chunk_consuming.p, 0, (size_t)K * 4, st));
    TaskColumns cols{tk->env_id, tk->min_version, tk->requestor_ip};
    YDC_LAUNCH(c, "k_task_classify", k_task_classify, dim3(ceil_div(N, 256)), dim3(256), 0, st, cols, N,
                       c->d_cls_env.p, c->d_cls_ver.p, C, W, c->d_ip_sorted.p, c->d_ip_servant.p,
                       S, c->d_slot_base.p, cs, c->d_mask.p, c->d_self_lo.p, c->d_self_hi.p,
                       c->d_chunk_consuming.p, prm);
    YDC_LAUNCH(c, "k_chunk_prefix", k_chunk_prefix, dim3(1), dim3(1024), 0, st, c->d_chunk_consuming.p, K,
                       c->d_before.p);
    if (C)
      YDC_LAUNCH(c, "k_guess_init", k_guess_init, dim3(ceil_div(K * C, 256)), dim3(256), 0, st, L,
                         c->d_before.p, K, c->d_guess[0].p, c->d_dirty.p);
  }
  mark(c, 5);
  // ---- matching
  uint32_t rounds = 0;
  SharedIpTable no_shared{};
  if (N && C == 0) {
    // No eligible servant at all: every request fails with EnvironmentNotFound
    // (task_dispatcher.cc:105-108).
    HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)c->d_slot_of.p, (int)kIdxEnvNotFound, N, st));
  }
  if (N && C) {
    if (any_shared) {
      // Whole batch sequentially, only if some task actually comes from a
      // host with several servants (decided on the device).
      SharedIpTable sh{c->d_ip_sorted.p, c->d_ip_servant.p, S, c->d_class_of.p,
                       c->d_slot_base.p, S, c->d_left.p};
      YDC_LAUNCH(c, "k_init_left", k_init_left, dim3(ceil_div(std::max(S, 1u), 256)), dim3(256), 0, st,
                         c->d_slot_base.p, S, c->d_left.p);
      YDC_LAUNCH(c, "k_sim_generic", k_sim_generic, dim3(1), dim3(64), 0, st, L, T, N, N, 1u,
                         c->d_guess[0].p, c->d_endst.p, c->d_dirty.p, c->d_slot_of.p,
                         c->d_runs.p, sh, 1u, 0u, prm);
    }
    if (use_generic) {
      // > kMaxWaveClasses classes: thread-per-chunk kernel, host-checked rounds.
      for (;;) {
        for (uint32_t b = 0; b < c->opt_rounds_per_check; ++b) {
          ClassState* gold = c->d_guess[0].p;
          YDC_LAUNCH(c, "k_sim_generic", k_sim_generic, dim3(ceil_div(K, 64)), dim3(64), 0, st, L, T, N, cs, K,
                             gold, c->d_endst.p, c->d_dirty.p, c->d_slot_of.p, c->d_runs.p,
                             no_shared, 0u, rounds, prm);
          YDC_LAUNCH(c, "k_update", k_update, dim3(std::max(1u, ceil_div(K * C, 256))), dim3(256), 0,
                     st, C, K, c->d_endst.p, gold, c->d_dirty.p, rounds, prm);
          ++rounds;
        }
        HIP_TRY(c, hipMemcpyAsync(c->h_prm, prm, sizeof(DeviceParams), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (c->h_prm->overflow)
          return fail(c, YDC_ERR_CAPACITY, "slot workspace overflow (bound %u)", slot_bound);
        if (c->h_prm->n_changed[(rounds - 1) & 63] == 0) break;
        if (rounds > K + 4) return fail(c, YDC_ERR_NOT_CONVERGED, "no fixpoint after %u rounds", rounds);
      }
    }
  }
  const bool wave_path = N && C && !use_generic;
  // ---- matching rounds + finalise. The rounds are pre-launched: a round returns at
  // once when an earlier one already proved the fixpoint, k_finalize only runs when
  // the last launched round did, and the host looks at the counters once.
  MatchBuffers mb{};
  uint32_t rshift = 4, init_fill = 8;
  if (wave_path) {
    mb.guess0 = c->d_guess[0].p;
    mb.endst = c->d_endst.p;
    mb.checkpoint = c->d_checkpoint.p;
    mb.claim = c->d_claim.p;
    mb.slot_of = c->d_slot_of.p;
    mb.boundary_in = nullptr;
    // Ring of R = 2^rshift entries per class; a wave's rings hold 2048 entries in all
    // (16 KB of LDS: ranks + generation indexes), see match_kernel.h.
    rshift = 3;
    while (rshift < 10 && ((size_t)C << (rshift + 1)) <= 2048) ++rshift;
    const uint32_t R = 1u << rshift;
    uint32_t want = 2 * cs / std::max(C, 1u);
    while (init_fill < want && init_fill < 64) init_fill <<= 1;
    init_fill = std::min(std::max(init_fill, 32u), R);
  }
  auto launch_round = [&](uint32_t r, uint32_t device_check) {
    const size_t lds = 16384;
    const uint32_t stamp = ++c->pass_stamp;
    if (W == 1) {
      YDC_LAUNCH(c, "k_match_pass", (k_match_pass<1>), dim3(K), dim3(64), lds, st, L, T, N, cs, K,
                 mb, r, stamp, device_check, rshift, init_fill, prm);
    } else if (W == 2) {
      YDC_LAUNCH(c, "k_match_pass", (k_match_pass<2>), dim3(K), dim3(64), lds, st, L, T, N, cs, K,
                 mb, r, stamp, device_check, rshift, init_fill, prm);
    } else {
      YDC_LAUNCH(c, "k_match_pass", (k_match_pass<4>), dim3(K), dim3(64), lds, st, L, T, N, cs, K,
                 mb, r, stamp, device_check, rshift, init_fill, prm);
    }
  };
  auto launch_finalize = [&](uint32_t check_slot) -> int {
    if (S) HIP_TRY(c, hipMemcpyAsync(c->d_running_out.p, c->d_running.p, (size_t)S * 4,
                                     hipMemcpyDeviceToDevice, st));
    if (N) {
      YDC_LAUNCH(c, "k_finalize", k_finalize, dim3(ceil_div(N, 256)), dim3(256), 0, st, sv, c->d_slot_base.p,
                         c->d_slot_of.p, N, d_out_idx, d_out_util, c->d_running_out.p, check_slot, prm);
    }
    // Not converged yet: k_finalize returned at once, running_out == running and the
    // copies below change nothing; they are repeated after the extra rounds.
    if ((flags & YDC_DISPATCH_COMMIT) && S)
      HIP_TRY(c, hipMemcpyAsync(c->d_running.p, c->d_running_out.p, (size_t)S * 4,
                                hipMemcpyDeviceToDevice, st));
    if (d_out_running && S)
      HIP_TRY(c, hipMemcpyAsync(d_out_running, c->d_running_out.p, (size_t)S * 4,
                                hipMemcpyDeviceToDevice, st));
    return YDC_OK;
  };
  mark(c, 6);
  if (wave_path) {
    uint32_t launched = 0;
    for (;;) {
      const uint32_t group = launched == 0 ? std::max(2u, std::min(c->round_hint, 16u)) : 4u;
      if (launched) {
        // Counter slots of the rounds to come (the first group's were cleared by
        // k_servant_scan). The stream is idle here: the host has just synchronised.
        for (uint32_t r = launched; r < launched + group; ++r)
          HIP_TRY(c, hipMemsetAsync(&prm->n_changed[r & 63], 0, 4, st));
      }
      for (uint32_t r = launched; r < launched + group; ++r) {
        launch_round(r, 1u);
      }
      launched += group;
      if (int rc = launch_finalize((launched - 1) & 63)) return rc;
      mark(c, 7);
      HIP_TRY(c, hipMemcpyAsync(c->h_prm, prm, sizeof(DeviceParams), hipMemcpyDeviceToHost, st));
      HIP_TRY(c, hipStreamSynchronize(st));
      HIP_TRY(c, hipGetLastError());
      if (c->h_prm->overflow)
        return fail(c, YDC_ERR_CAPACITY, "slot workspace overflow (bound %u)", slot_bound);
      if (c->h_prm->need_shared) {  // the sequential path produced the result
        rounds = 1;
        break;
      }
      if (c->h_prm->n_changed[(launched - 1) & 63] == 0) {
        // First round of this group that changed nothing.
        rounds = launched;
        for (uint32_t r = launched - group; r < launched; ++r)
          if (c->h_prm->n_changed[r & 63] == 0) {
            rounds = r + 1;
            break;
          }
        c->round_hint = rounds;
        if (c->debug_sim) {
          fprintf(stderr, "[ydc match] K=%u cs=%u R=%u fill=%u rounds=%u sims=%u changed:", K, cs,
                  1u << rshift, init_fill, rounds, c->h_prm->chunk_sims);
          for (uint32_t r = 0; r < rounds; ++r) fprintf(stderr, " %u", c->h_prm->n_changed[r & 63]);
          fprintf(stderr, "\n");
        }
        break;
      }
      if (launched > K + 4)
        return fail(c, YDC_ERR_NOT_CONVERGED, "no fixpoint after %u rounds", launched);
    }
  } else {
    if (int rc = launch_finalize(kNone)) return rc;
    mark(c, 7);
    HIP_TRY(c, hipMemcpyAsync(c->h_prm, prm, sizeof(DeviceParams), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipGetLastError());
    if (c->h_prm->overflow)
      return fail(c, YDC_ERR_CAPACITY, "slot workspace overflow (bound %u)", slot_bound);
  }

  ydc_stats& s = c->stats;
  std::memset(&s, 0, sizeof(s));
  s.n_tasks = N;
  s.n_servants = S;
  s.n_classes = C;
  s.n_slots = c->h_prm->n_slots;
  s.key_bits = c->kf.key_bits;
  s.radix_passes = key_passes;
  s.n_chunks = K;
  s.rounds = rounds;
  s.chunk_sims = c->h_prm->chunk_sims;
  s.granted = c->h_prm->granted;
  s.timeouts = c->h_prm->timeouts;
  s.env_not_found = c->h_prm->env_not_found;
  if (c->profiling) {
    for (int i = 0; i < 7; ++i) (void)hipEventElapsedTime(&s.stage_ms[i], c->ev[i], c->ev[i + 1]);
    (void)hipEventElapsedTime(&s.stage_ms[YDC_STAGE_TOTAL], c->ev[0], c->ev[7]);
    // Per-kernel totals of this dispatch as JSON: {"name": [launches, total_ms], ...}
    std::vector<std::pair<std::string, std::pair<int, double>>> acc;
    for (size_t i = 0; i < c->ksamples_used; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, c->ksamples[i].a, c->ksamples[i].b) != hipSuccess) continue;
      bool found = false;
      for (auto& e : acc)
        if (e.first == c->ksamples[i].name) {
          e.second.first++;
          e.second.second += ms;
          found = true;
        }
      if (!found) acc.push_back({c->ksamples[i].name, {1, ms}});
    }
    std::string j = "{";
    for (size_t i = 0; i < acc.size(); ++i) {
      char buf[160];
      snprintf(buf, sizeof(buf), "%s\"%s\": [%d, %.6f]", i ? ", " : "", acc[i].first.c_str(),
               acc[i].second.first, acc[i].second.second);
      j += buf;
    }
    c->kprofile_json = j + "}";
  }
  return YDC_OK;
}

int ydc_dispatch(ydc_context* c, const ydc_task_soa* tk, uint32_t N, uint32_t flags,
                 uint32_t* out_idx, double* out_util, uint32_t* out_running) {
  if (!c || (N && (!tk || !out_idx))) return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, c->d_t_env.reserve(N));
  HIP_TRY(c, c->d_t_minv.reserve(N));
  HIP_TRY(c, c->d_t_ip.reserve(N));
  HIP_TRY(c, c->d_out_idx.reserve(N));
  if (out_util) HIP_TRY(c, c->d_out_util.reserve(N));
  if (N) {
    HIP_TRY(c, hipMemcpyAsync(c->d_t_env.p, tk->env_id, (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_t_minv.p, tk->min_version, (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_t_ip.p, tk->requestor_ip, (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
  }
  ydc_task_soa d{c->d_t_env.p, c->d_t_minv.p, c->d_t_ip.p};
  int rc = ydc_dispatch_device(c, &d, N, flags, c->d_out_idx.p, out_util ? c->d_out_util.p : nullptr,
                               nullptr);
  if (rc) return rc;
  if (N) {
    HIP_TRY(c, hipMemcpyAsync(out_idx, c->d_out_idx.p, (size_t)N * 4, hipMemcpyDeviceToHost, c->stream));
    if (out_util)
      HIP_TRY(c, hipMemcpyAsync(out_util, c->d_out_util.p, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
  }
  if (out_running && c->n_servants)
    HIP_TRY(c, hipMemcpyAsync(out_running, c->d_running_out.p, (size_t)c->n_servants * 4,
                              hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return YDC_OK;
}

int ydc_synchronize(ydc_context* c) {
  if (!c) return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return YDC_OK;
}

int ydc_set_profiling(ydc_context* c, int on) {
  if (!c) return YDC_ERR_INVALID_ARGUMENT;
  c->profiling = on != 0;
  return YDC_OK;
}

const char* ydc_kernel_profile(const ydc_context* c) {
  return c ? c->kprofile_json.c_str() : "{}";
}

int ydc_get_stats(const ydc_context* c, ydc_stats* out) {
  if (!c || !out) return YDC_ERR_INVALID_ARGUMENT;
  *out = c->stats;
  return YDC_OK;
}

}  // extern "C"

// ydc_api.hip — the extern "C" boundary (include/yadcc_dispatch.h): context,
// resident servant registry, per-batch launch sequence. gfx950 only; there is
// no CPU fallback — without a device every call fails with YDC_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is resolved with dlopen at ydc_group_init

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
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
#include "tick_kernel.h"

using namespace ydc;

namespace {

// Owns one device allocation (freed with the context: `delete c` releases whatever
// ydc_destroy did not name).
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) {
    o.p = nullptr;
    o.cap = 0;
  }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      cap = o.cap;
      o.p = nullptr;
      o.cap = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
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

namespace {
// Sizes, workspace views and kernel argument blocks of one batch (plan_batch).
struct BatchPlan {
  uint32_t N = 0, S = 0, C = 0, W = 1, slot_bound = 0, n_tiles = 1, cs = 64, K = 0;
  uint32_t key_passes = 0, cls_passes = 0, cls_bits = 0, rshift = 4, init_fill = 8, sort_items = 8;
  uint32_t ring_total = 2048;  // entries of all rings of a matching wave (x 8 B of LDS)
  bool dense = false;  // the matching kernel's 4-waves-per-SIMD build (match_kernel.h: OCC)
  uint32_t fused_cls_bits = 0;  // class partition folded into the last key pass (kernels.h)
  uint32_t gbits = 0;  // != 0: the sort's values carry the class above gbits slot bits (SortIn)
  uint32_t slot_bound_glob = 0, win_margin = 0;  // (sharded sort: slot_bound is the window's)
  bool key32 = true, any_shared = false, use_generic = false, wave_path = false;
  bool packed = false;  // the sort moves 8-byte (key, value) records (32-bit keys)
  bool win = false;  // multi-GPU with a sharded sort: this rank only holds a key window of the slots
  // Bin sort (bin_sort.h) instead of the radix sort: n_bins equal bins over the key space.
  bool fuse01 = false;  // the launch of matching pass 0 is pass 1 as well (no launch for pass 1)
  bool wide_lists = false;  // > 256 classes with short eligible-class rows: k_sim_wide's list form
  bool binsort = false;
  bool zone = false;            // workgroup 0 of the launch of pass 0 walks the tier's end (zone_guess.h)
  bool zone_eligible = false;   // ... or would, if zone_decide had found it worth its while
  uint32_t n_bins = 0, bin_shift = 0, bin_slot_bits = 0, bin_cls_bits = 0;
  uint32_t bin_group = 0, bin_tiles = 0;  // servants per slot tile, slot tiles (k_front_bins)
  ServantTable sv{};
  ClassLists L{};
  TaskTable T{};
  MatchBuffers mb{};
  const uint32_t* rank_to_g = nullptr;  // slots in key order (before the class partition)
  uint32_t rank_stride = 1;             // 2: rank_to_g are the values of 8-byte sort records
  SharedIpTable shared{};  // pos_last != NULL: some host runs several servants
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
  std::vector<uint32_t> h_alias_ip, h_alias_servant;  // ydc_set_host_aliases: further ip table entries
  std::vector<uint64_t> h_env;  // env_words words per servant
  uint32_t env_words = 1;
  uint32_t n_parts = 1;         // independent parts of the registry (host_tables.h)
  HostTables tables;
  KeyFormat kf{};
  bool tables_dirty = true;

  // Resident registry.
  DevBuf<uint32_t> d_version, d_nproc, d_load, d_max_tasks, d_running, d_flags, d_class_of;
  DevBuf<uint32_t> d_spare[6];  // ydc_remove_servants compacts into these, then swaps
  DevBuf<uint32_t> d_ip_hash, d_ip_filter;
  DevBuf<uint32_t> d_bin_tile_start, d_bin_tile_base;  // slot tiles of the bin sort's front (host_tables.h)
  DevBuf<uint32_t> d_ip_sorted, d_ip_servant, d_cls_ver, d_ver_sorted, d_cls_comp, d_part_base;
  DevBuf<uint64_t> d_cls_env, d_env_ver_mask;
  DevBuf<uint8_t> d_cls_single;

  // Per-batch workspace.
  DevBuf<uint32_t> d_slot_base, d_cls_begin, d_vals[2], d_hist, d_row_total, d_tile_first;
  DevBuf<uint64_t> d_keys[2];  // viewed as u32 when the key fits
  DevBuf<uint16_t> d_cls_by_g;
  DevBuf<uint32_t> d_owner;     // servant of every slot (generation order)
  DevBuf<uint32_t> d_rank_to_g; // global rank -> slot when the class pass is fused into the sort
  DevBuf<unsigned long long> d_zone_box;  // zone_walk's granules: header + a row of cursors per chunk
  DevBuf<uint32_t> d_binbase, d_binruns;  // bin sort: starts of the bins per class, run table of the slot tiles
  DevBuf<uint32_t> d_level_tab;           // bin sort: class-list positions at every 64th global rank
  DevBuf<uint32_t> d_elig_off, d_elig_cls, d_row_of;  // > 256 classes: eligible-class lists, the requests' rows
  DevBuf<uint64_t> d_mask;
  DevBuf<uint32_t> d_self_lo, d_self_hi, d_chunk_consuming, d_before, d_slot_of, d_pos_last;
  DevBuf<uint32_t> d_chunk_tail;  // consuming requests among the last kWarmUp of every chunk
  DevBuf<uint32_t> d_running_out;
  DevBuf<ClassState> d_guess[1], d_endst, d_checkpoint, d_early;
  DevBuf<unsigned long long> d_claim, d_hand;  // d_hand: hand-off granules of the pass 0 + 1 launch
  uint32_t round_hint = 3;  // passes to pre-launch before looking at the outcome

  // Pipelined batches (ydc_dispatch_device_async / ydc_dispatch_wait): up to two batches are
  // enqueued before the host looks at the outcome of the older one.
  struct Pending {
    bool active = false, rerun = false;
    BatchPlan plan;
    ydc_task_soa tk{};
    uint32_t n = 0, flags = 0, launched = 0;
    uint32_t* out_idx = nullptr;
    double* out_util = nullptr;
    uint32_t* out_running = nullptr;
    DeviceParams* h_outcome = nullptr;  // pinned
    DeviceParams* d_h_outcome = nullptr;  // ... its device address
    hipEvent_t ev = nullptr;
  } pend[2];
  uint32_t pend_head = 0, pend_count = 0;
  bool enqueue_pipelined = false;  // the finalise being enqueued belongs to a pipelined batch
  uint64_t pipeline_misses = 0;

  // Multi-GPU group (ydc_group_*): this context is one rank of a sharded dispatcher.
  struct LocalHub;  // single-process transport: several contexts on one device
  struct Group {
    int rank = 0, n_ranks = 0;  // n_ranks == 0: not in a group
    void* rccl = nullptr;       // dlopen handle
    ncclComm_t comm = nullptr;
    decltype(&ncclAllGather) all_gather_fn = nullptr;
    decltype(&ncclCommDestroy) comm_destroy_fn = nullptr;
    decltype(&ncclCommCount) comm_count_fn = nullptr;
    decltype(&ncclGetErrorString) error_string_fn = nullptr;
    LocalHub* hub = nullptr;
    // Inter-process mailbox transport (ydc_group_ipc_export / ydc_group_init_ipc): this rank's
    // mailbox in both flavours — device memory behind a HIP IPC handle, a shared host segment —
    // and the peers' as mapped into this process (kernels.h: k_mailbox_all_gather).
    struct Mailbox {
      int kind = 0;  // 0: not in use; YDC_TRANSPORT_IPC_DEVICE / YDC_TRANSPORT_IPC_HOST once initialised
      int exported_ranks = 0, exported_rank = -1;
      void* own_dev = nullptr;        // device flavour
      bool own_dev_fine = false;
      bool have_handle = false;
      hipIpcMemHandle_t handle{};
      void* own_host = nullptr;       // host flavour (mmap of the shm segment, registered)
      char shm_name[64] = {};
      size_t bytes = 0;
      uint32_t slot_words = 0;
      MailboxPeers peers{};
      void* opened_dev[kMailboxMaxRanks] = {};   // hipIpcOpenMemHandle results to close
      void* opened_host[kMailboxMaxRanks] = {};  // peers' segments mapped here
      uint32_t seq = 0;                           // exchanges so far (the stamp; never 0)
      unsigned long long timeout_ticks = 30ull * 100000000ull;
    } box;
    DevBuf<uint32_t> d_totals, d_meta, d_base, d_delta, d_deltas;
    // Sharded sort (k_window): key-count table, per-servant windows, local prefix, class lists
    // of the whole registry, local -> registry-wide list position shifts, the ranks' windows.
    DevBuf<uint32_t> d_cum, d_r_first, d_lbase, d_cls_begin_glob, d_shift, d_winrec, d_winall;
    DevBuf<ClassState> d_bound_local;
    uint32_t margin_scale = 1;  // doubled after a batch whose window missed
    uint64_t windowed_batches = 0, window_misses = 0;
    DevBuf<uint32_t> d_pad, d_gather, d_all[3], d_all_idx;  // replicated fallback (whole batch)
    DevBuf<double> d_all_util;
    DevBuf<ClassState> d_send, d_bounds;
    ClassState* h_bounds = nullptr;  // pinned
    size_t h_bounds_cap = 0;
    uint32_t passes = 0;             // of the last sharded batch
    uint32_t pass_hint = 3;          // passes to pre-launch before looking at the outcome
  } group;

  // Streaming mode (ydc_stream_*): one tick = row updates + slot releases + one
  // committed batch, replayed from a captured graph.
  struct Stream {
    bool active = false, stale = true;
    uint32_t max_upd = 0, max_rel = 0, max_tasks = 0, passes = 0;
    // One pinned staging arena for everything a tick brings (heartbeat indexes and rows,
    // released slots, the three request columns) and its device mirror: one H2D copy per
    // tick. The typed pointers below point into the two arenas.
    uint8_t* h_in = nullptr;
    DevBuf<uint8_t> d_in;
    size_t in_bytes = 0;
    uint32_t *h_upd_idx = nullptr, *h_rel = nullptr, *h_env = nullptr, *h_minv = nullptr,
             *h_ip = nullptr, *h_out = nullptr;
    ydc_servant_row* h_upd_rows = nullptr;
    uint32_t *d_upd_idx = nullptr, *d_rel = nullptr, *d_env = nullptr, *d_minv = nullptr,
             *d_ip = nullptr;
    ServantRowDev* d_upd_rows = nullptr;
    // The same sections as the kernels of the captured step see them when they read the
    // page-locked arena in place (no H2D copy node), and the result array likewise.
    uint32_t *z_upd_idx = nullptr, *z_rel = nullptr, *z_env = nullptr, *z_minv = nullptr, *z_ip = nullptr,
             *z_out = nullptr;
    ServantRowDev* z_upd_rows = nullptr;
    bool zero_copy = false;  // the captured step in use reads / writes the arenas in place
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    BatchPlan plan;
    // COMMIT without a copy node (commit_swap): the step exists twice, captured with the two
    // running_tasks columns in either role; a tick that took effect makes its output THE column
    // and the next tick replays the other capture. run_a / run_b: the column each one reads.
    hipGraph_t graph_b = nullptr;
    hipGraphExec_t exec_b = nullptr;
    BatchPlan plan_b;
    const uint32_t *run_a = nullptr, *run_b = nullptr;
    bool swaps = false;
    uint64_t ticks = 0, recaptures = 0, eager_fallbacks = 0;
    bool eager_only = false;  // the registry's batches cannot be captured (> 256 classes): every tick runs eagerly
    // Passes to capture: one more than the last eager batch needed, to begin with; after 64
    // ticks that all needed fewer, exactly the most any of them needed (a pre-launched pass
    // that finds nothing to do still costs a launch); more again after a tick that ran out.
    uint32_t want_passes = 0, window_max = 0, window_ticks = 0;
  } stream_mode;
  DevBuf<ClassRun> d_runs;
  DevBuf<uint8_t> d_dirty;
  bool debug_sim = false;
  DevBuf<DeviceParams> d_prm;
  DeviceParams* h_prm = nullptr;  // pinned
  DeviceParams* d_h_prm = nullptr;  // ... its device address
  // Where the finalise being enqueued hands the batch's outcome to the host itself (NULL: the
  // caller reads d_prm back with a copy) — kernels.h: RunningArgs::host_outcome.
  DeviceParams* finalize_outcome = nullptr;
  bool opt_outcome_store = true;  // (outcome_store=0: always the copy)
  bool opt_release_counted = true;  // long release lists counted in LDS first (release_counted=0: an atomic per slot)
  bool opt_stream_zero_copy = true;  // streaming: the captured step reads / writes the page-locked arenas in place

  // Staging for the host-pointer entry point (ydc_dispatch): the three request columns in
  // one pinned arena and its device mirror (one H2D copy), the results (indexes |
  // running_tasks | utilisation) in another pair (one D2H copy, enqueued right behind the
  // finalise kernels so that the batch needs a single wait).
  uint8_t *h_in = nullptr, *h_res = nullptr;
  uint32_t* h_rel = nullptr;  // pinned staging of ydc_release_slots
  size_t h_rel_cap = 0;
  hipEvent_t h_rel_ev = nullptr;
  size_t h_in_cap = 0, h_res_cap = 0;
  DevBuf<uint8_t> d_in, d_res;
  struct {
    void* dst = nullptr;
    const void* src = nullptr;
    size_t bytes = 0;
  } post_copy;  // D2H copy to enqueue after every finalise of the current batch (bytes == 0: none)
  // ydc_dispatch: the request columns are only needed by the classification, so the slot
  // generation and the sort are enqueued first and run while the host stages the columns and
  // the H2D copy travels on a stream of its own (stage_host_requests).
  struct {
    bool active = false;
    bool direct = false;  // the caller's columns are page-locked: DMA straight from them
    const ydc_task_soa* tk = nullptr;
    uint32_t n = 0;
    size_t col = 0, bytes = 0;
  } host_in;
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_ev = nullptr;
  DevBuf<uint32_t> d_out_idx, d_upd_idx;
  DevBuf<ydc_servant_row> d_upd_rows;

  // Small-batch path (tick_kernel.h): one launch per call, requests / deltas / results as kernel
  // arguments and plain stores to page-locked memory.
  DevBuf<uint32_t> d_ip;            // resident copy of the ip_id column (rebuild_tables)
  TickDone* h_tick_done = nullptr;  // page-locked, coherent: the kernel's stamp + counters
  TickDone* d_tick_done = nullptr;  // ... its device address
  uint8_t *h_tick_io = nullptr, *d_tick_io = nullptr;  // page-locked arena: columns / deltas in, results out
  size_t tick_io_cap = 0;
  uint32_t tick_seq = 0;
  // Batches up to this many requests take it (small_batch=0: none does). kSmallBatchAuto: by
  // registry size — a pick costs ~1 us at 2k servants and ~4 us at 16k, the batch pipeline
  // ~90 / ~165 us whatever the batch holds (profiles/r05_td_latency_table.txt).
  static constexpr uint32_t kSmallBatchAuto = 0xFFFFFFFFu;
  uint32_t opt_small_batch = kSmallBatchAuto;
  // `same`: the requests are copies of one another (one RPC) and the kernel will place them as
  // one merge (tick_merges): ~0.5 us per request at 2k servants, ~1 us at 16k.
  uint32_t small_batch(bool same = false) const {
    if (opt_small_batch != kSmallBatchAuto) return opt_small_batch;
    return n_servants <= 4096 || same ? 64u : n_servants <= 8192 ? 48u : 32u;
  }
  uint64_t tick_batches = 0;
  // The resident form: the kernel of a COMMITting tick stays on its CU, the registry in its
  // registers, and takes the following ticks from a page-locked mailbox (tick_kernel.h: TickBox) —
  // no launch, no column loads. Every other use of the context ends it first (resident_stop).
  bool opt_resident = true;        // (resident=0: every tick is a launch)
  bool opt_tick_packed = true;     // (packed_tick=0: the two-word candidate everywhere)
  uint32_t opt_resident_idle_ms = 50;  // the kernel leaves by itself when nobody has asked for this long
  TickBox *h_box = nullptr, *d_box = nullptr;
  hipStream_t res_stream = nullptr;
  hipEvent_t res_ev = nullptr;
  bool res_live = false;  // a resident kernel was launched and has not been seen to leave
  bool res_util = false;  // ... and it stores utilisations (fixed at its launch: TickArgs::out_util)
  uint64_t tick_resident = 0, tick_launches = 0, pipeline_batches = 0;

  uint32_t opt_chunk_size = 0;     // 0: automatic
  uint32_t opt_target_chunks = 2048;
  // 16 KB of LDS per matching wave = 10 waves per CU. Smaller rings (more waves per CU, shorter
  // chunks) were measured and bring nothing: the waves saturate VALU issue at ~2 per SIMD.
  uint32_t opt_ring_total = 0;  // entries of a matching wave's rings; 0: chosen per batch (YDC_RING_TOTAL)
  uint32_t opt_xcd = 3;  // XCD-contiguous tile order: 1 slot generation, 2 histograms, 4 scatters (YDC_XCD_TILES)
  bool opt_scan_multi = true;  // (scan_multi=0: one workgroup loops over the slabs)
  bool opt_group_walk = true;  // sparse eligibility: the walk in groups of 64 requests (YDC_GROUP_WALK=0: one at a time)
  uint32_t opt_zone_guess = 1;  // start guesses around the dedicated tier's end from a walk of that stretch: 1 where it pays, 2 always, 0 never
  // lead: where the walk starts, in levels before the tier's end (the first chunk boundary inside
  // it: up to a chunk less). It has to be on the true track when the transient begins (cfg3: 840
  // levels before the end; a start 1349 before it is too late, 1861 is not), and every request of
  // the lead delays the stretch's last chunk by 54 ns: the default, and more after a batch whose
  // served chunks did not come out consistent (zone_feedback).
  uint32_t opt_zone_lead = 2304, opt_zone_trail = 1024, opt_zone_max_chunks = 2560;
  uint32_t zone_lead_cur = 0;   // (0: opt_zone_lead) what zone_feedback has raised the lead to
  uint32_t zone_fails = 0, zone_cooldown = 0;  // failures at the largest lead; batches without a walk
  // Whether the walk pays is the registry's business (three of four seeds of cfg3's pool have no
  // chain at the tier's end: the chunks there would wait 200 us for cursors their level guesses
  // already had): decided from what batches of this shape cost on the device with and without it
  // (zone_decide / zone_feedback; zone_guess=2 forces the walk, 0 forbids it).
  struct ZoneArm {
    uint32_t n = 0;
    float ticks = 0;  // (moving average, 100 MHz)
  } zone_on, zone_off;
  uint32_t zone_off_rounds = 0;  // rounds of the last batch without the walk
  bool zone_cold = true;         // the shape's first batch has not been seen yet (not counted)
  uint32_t zone_since_probe = 0;
  uint64_t zone_shape = 0;       // the plans the figures above are about
  bool opt_walk_packed = true; // ... with head rank and class id in one word where they fit (walk_packed=0: two arrays)
  bool opt_tile_tab = true;  // level searches narrowed by the class pass's histogram table (YDC_TILE_TAB=0)
  bool opt_classify_multi = true;  // (YDC_CLASSIFY_PER_THREAD=1: one request per thread everywhere)
  bool opt_split_gen = false;  // slot generation and request classification as two launches (YDC_SPLIT_GEN=1)
  bool opt_dense = true;  // 4-waves-per-SIMD matching kernel and twice the chunks where it pays (YDC_DENSE=0)
  bool opt_fused_class = true;
  bool opt_own_guess = true;
  bool opt_pair = true;
  bool opt_packed_class = true;
  bool opt_shard_sort = true;
  bool opt_packed_sort = true;  // 8-byte (key, value) sort records for 32-bit keys
  // Bin sort (three launches, bin_sort.h) for registries that offer at most this many slots;
  // a batch with a bin too large for LDS is repeated with the radix sort, which then stays
  // (binsort_blocked) until the registry changes structure.
  bool opt_fuse_passes = true;  // one GPU: the launch of pass 0 does pass 1 as well (match_kernel.h)
  uint32_t opt_warm_up = 0;  // requests a chunk of pass 0 starts early (1 .. 64; 0: by chunk size)
  uint32_t opt_cp_every = 4;  // checkpoints before every 4th block of a chunk (MatchBuffers::cp_every)
  uint32_t opt_hand_tries = kHandTries;  // (tests: 0 makes most waves give up and leave their chunk to pass 2)
  bool opt_binsort = true;
  bool opt_stream_graph = true;  // the streaming step is replayed from its hipGraph (0: enqueued eagerly)
  bool opt_walk_park = true;  // k_walk_groups parks its fetches in a254 / a255 (0: in plain variables)
  bool opt_group_binsort = true;  // multi-GPU: bin sort of the whole registry before windowed radix (YDC_GROUP_BINSORT=0)
  // The walk of the wide kernel with prefetch waves (YDC_WALK_PREFETCH=1). Off: measured slower
  // than the walker's own one-ahead fetch (57 against 35 ms of 100k picks; the scan was the cost).
  bool opt_walk_prefetch = false;
  bool opt_wide_lists = true;  // eligible-class lists for the wide kernel where every row is short (YDC_WIDE_LISTS=0: masks)
  bool opt_wide = true;  // > 256 classes: wave-per-chunk replay (YDC_WIDE=0: thread per chunk)
  bool opt_level_tab = true;  // bin sort leaves a level table for pass 0's guesses (YDC_LEVEL_TAB=0: search)
  // ydc_dispatch with page-locked caller buffers: no staging (YDC_ZERO_COPY=0 switches it off);
  // request columns read in place through the mapped pointer (YDC_HOST_IN=map) or copied by DMA
  // from where they lie (YDC_HOST_IN=copy).
  bool opt_zero_copy = true, opt_host_in_map = true;
  uint32_t opt_binsort_max_slots = 600000;
  bool binsort_blocked = false;
  bool debug_verify_binsort = false;  // YDC_BINSORT_VERIFY=1: check every bin sort against a host sort
  uint64_t binsort_misses = 0;
  int64_t opt_shard_margin = -1;  // >= 0: margin of the key windows in slots (tests)
  uint32_t opt_rounds_per_check = 2;
  // Passes after which the matching stops repairing chunks in parallel and lets one wave walk
  // each chain to its end (match_kernel.h: walk): registries whose every chunk boundary carries a
  // state no guess predicts (a handful of servants with tens of thousands of slots and requests
  // from their own hosts: 554 passes, 1.2 s; healthy batches need 2 - 4) — `walk_after` passes, then
  // a scout and the walk.
  // COMMIT by exchanging the resident running_tasks column with k_finalize's output instead of
  // copying it back (set around enqueue_finalize by callers that do the exchange; never while a
  // captured streaming step holds the two addresses).
  bool commit_by_swap = false;
  bool opt_commit_swap = true;  // (commit_swap=0: always the copy)
  uint32_t opt_walk_after = 12;
  uint32_t walk_flag = 0;
  uint32_t walked_at = 0;  // passes launched before the last batch's walk (0: it was not walked)
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

void stream_release(ydc_context* c);  // streaming mode, defined further down
void resident_stop(ydc_context* c);   // small-batch path's resident kernel, defined further down
void group_release(ydc_context* c);   // multi-GPU group, defined further down

// Errors raised before (or without) a context: process-wide, written from any thread.
std::mutex g_create_error_mu;
std::string g_create_error_text;
void set_create_error(std::string text) {
  std::lock_guard<std::mutex> lk(g_create_error_mu);
  g_create_error_text = std::move(text);
}
const char* create_error_cstr() {
  static thread_local std::string copy;  // (the caller reads it after the lock is gone)
  std::lock_guard<std::mutex> lk(g_create_error_mu);
  copy = g_create_error_text;
  return copy.c_str();
}

// Page-locked host ranges the device can address (ydc_host_register / ydc_host_alloc, or found
// pinned by the caller's own means): ydc_dispatch hands such buffers to the kernels as they are —
// request columns read and results written through the mapped pointer, no staging copy.
struct PinnedRange {
  const char* host;
  size_t bytes;
  char* dev;
  int kind;  // 0: registered here, 1: allocated here
};
std::mutex g_pinned_mu;
std::vector<PinnedRange> g_pinned;
// Pointers that were asked about and are NOT pinned (so that an unpinned caller does not pay a
// runtime query per batch); forgotten whenever a range is registered.
std::vector<const void*> g_not_pinned;

// Device address of [p, p + bytes) if it lies in a pinned range, else NULL.
void* pinned_device_pointer(const void* p, size_t bytes) {
  if (!p) return nullptr;
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  const char* q = (const char*)p;
  for (auto& r : g_pinned)
    if (q >= r.host && q + bytes <= r.host + r.bytes) return r.dev + (q - r.host);
  for (auto* np : g_not_pinned)
    if (np == p) return nullptr;
  // Pinned by the caller itself (hipHostMalloc / hipHostRegister outside this library)?
  // The whole range must lie in ONE pinned allocation: the last byte is asked about too and must
  // map where the first one's mapping continues (a column longer than its pinned part is staged).
  hipPointerAttribute_t a{}, z{};
  if (hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost && a.devicePointer) {
    if (bytes <= 1 ||
        (hipPointerGetAttributes(&z, q + bytes - 1) == hipSuccess && z.type == hipMemoryTypeHost &&
         z.devicePointer == (char*)a.devicePointer + (bytes - 1)))
      return a.devicePointer;  // (asked again next time: its owner may free it behind our back)
  }
  (void)hipGetLastError();
  if (g_not_pinned.size() >= 64) g_not_pinned.clear();
  g_not_pinned.push_back(p);
  return nullptr;
}

// Developer / test switchboard: ONE environment variable, YDC_TUNE="key=value,key=value", read
// when a context is created. It forces code paths the planner would not choose on its own (the
// parity tests cover the fallbacks with it: chunk and ring sizes, the radix pipeline on a
// registry the bin sort would take, the lone walker, ...) and splits launches for measurements.
// Not an interface: a scheduler never sets it, and nothing in it changes a placement.
const char* tune_value(const char* key) {
  static thread_local std::string value;
  const char* all = getenv("YDC_TUNE");
  if (!all) return nullptr;
  const size_t klen = std::strlen(key);
  for (const char* p = all; *p;) {
    const char* end = std::strchr(p, ',');
    const size_t len = end ? (size_t)(end - p) : std::strlen(p);
    if (len > klen && p[klen] == '=' && std::strncmp(p, key, klen) == 0) {
      value.assign(p + klen + 1, len - klen - 1);
      return value.c_str();
    }
    p += len + (end ? 1 : 0);
  }
  return nullptr;
}

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
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  const uint32_t n = c->n_servants;
  c->tables.build(n, c->h_env.data(), c->h_version.data(), c->h_max_tasks.data(),
                  c->h_nproc.data(), c->h_ip.data(), c->env_words, (uint32_t)c->h_alias_ip.size(),
                  c->h_alias_ip.data(), c->h_alias_servant.data());
  const uint32_t n_ip = (uint32_t)c->tables.ip_sorted.size();
  c->n_parts = c->tables.n_comp;
  const uint32_t C = c->tables.n_classes();
  // The class partition on the last key pass, which may be given room for it. Up to 8 classes:
  // with 30 (cfg4: 9 + 9 + (5 + 5) bits instead of 8 + 8 + 7 and a class pass) the three passes
  // took 188 us where the four take 155 — wider digits scatter worse, and the fused pass ranks
  // twice and writes a third array (profiles/r04: measured and rejected).
  uint32_t fuse_bits = 0;
  if (C > 1 && C <= kMaxFusedClasses && c->opt_fused_class)
    while ((1u << fuse_bits) < C) ++fuse_bits;
  c->kf = choose_key_format(c->tables.cap_bits, kMaxRadixBits, &c->n_parts, fuse_bits);
  if (C > 65535) return fail(c, YDC_ERR_TOO_MANY_CLASSES, "%u servant classes", C);
  HIP_TRY(c, c->d_class_of.reserve(n));
  HIP_TRY(c, c->d_ip.reserve(n));
  HIP_TRY(c, c->d_ip_sorted.reserve(n_ip));
  HIP_TRY(c, c->d_ip_servant.reserve(n_ip));
  HIP_TRY(c, c->d_ip_hash.reserve(c->tables.ip_hash.size()));
  HIP_TRY(c, hipMemcpyAsync(c->d_ip_hash.p, c->tables.ip_hash.data(), c->tables.ip_hash.size() * 4,
                            hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, c->d_bin_tile_start.reserve(c->tables.bin_tile_start.size()));
  HIP_TRY(c, c->d_bin_tile_base.reserve(c->tables.bin_tile_start.size()));
  HIP_TRY(c, hipMemcpyAsync(c->d_bin_tile_start.p, c->tables.bin_tile_start.data(),
                            c->tables.bin_tile_start.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, c->d_ip_filter.reserve(c->tables.ip_filter.size()));
  HIP_TRY(c, hipMemcpyAsync(c->d_ip_filter.p, c->tables.ip_filter.data(), c->tables.ip_filter.size() * 4,
                            hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, c->d_cls_env.reserve((size_t)C * c->env_words));
  HIP_TRY(c, c->d_cls_ver.reserve(C));
  HIP_TRY(c, c->d_cls_begin.reserve(C + 1));
  HIP_TRY(c, c->d_part_base.reserve(kMaxComponents + 1));
  HIP_TRY(c, c->d_cls_single.reserve(C ? C : 1));
  if (C)
    HIP_TRY(c, hipMemcpyAsync(c->d_cls_single.p, c->tables.cls_single.data(), C, hipMemcpyHostToDevice,
                              c->stream));
  if (c->n_parts > 1) {
    HIP_TRY(c, c->d_cls_comp.reserve(C));
    HIP_TRY(c, hipMemcpyAsync(c->d_cls_comp.p, c->tables.cls_comp.data(), (size_t)C * 4,
                              hipMemcpyHostToDevice, c->stream));
  }
  if (n) {
    HIP_TRY(c, hipMemcpyAsync(c->d_class_of.p, c->tables.class_of.data(), n * 4,
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_ip.p, c->h_ip.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_ip_sorted.p, c->tables.ip_sorted.data(), (size_t)n_ip * 4,
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_ip_servant.p, c->tables.ip_servant.data(), (size_t)n_ip * 4,
                              hipMemcpyHostToDevice, c->stream));
  }
  if (C) {
    HIP_TRY(c, hipMemcpyAsync(c->d_cls_env.p, c->tables.cls_env.data(), (size_t)C * c->env_words * 8,
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_cls_ver.p, c->tables.cls_ver.data(), C * 4,
                              hipMemcpyHostToDevice, c->stream));
  }
  if (!c->tables.elig_off.empty()) {
    HIP_TRY(c, c->d_elig_off.reserve(c->tables.elig_off.size()));
    HIP_TRY(c, c->d_elig_cls.reserve(std::max<size_t>(c->tables.elig_cls.size(), 1)));
    HIP_TRY(c, hipMemcpyAsync(c->d_elig_off.p, c->tables.elig_off.data(), c->tables.elig_off.size() * 4,
                              hipMemcpyHostToDevice, c->stream));
    if (!c->tables.elig_cls.empty())
      HIP_TRY(c, hipMemcpyAsync(c->d_elig_cls.p, c->tables.elig_cls.data(), c->tables.elig_cls.size() * 4,
                                hipMemcpyHostToDevice, c->stream));
  }
  if (!c->tables.env_ver_mask.empty()) {
    HIP_TRY(c, c->d_ver_sorted.reserve(c->tables.ver_sorted.size()));
    HIP_TRY(c, c->d_env_ver_mask.reserve(c->tables.env_ver_mask.size()));
    HIP_TRY(c, hipMemcpyAsync(c->d_ver_sorted.p, c->tables.ver_sorted.data(),
                              c->tables.ver_sorted.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_env_ver_mask.p, c->tables.env_ver_mask.data(),
                              c->tables.env_ver_mask.size() * 8, hipMemcpyHostToDevice, c->stream));
  }
  // The host vectors are pageable: the copies above are complete on return only
  // after a sync (the tables may be rebuilt before the next launch otherwise).
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->tables_dirty = false;
  c->binsort_blocked = false;   // another registry: the bins get another chance
  c->stream_mode.stale = true;  // a captured streaming step bakes the table sizes in
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
  HIP_TRY(c, c->d_pos_last.reserve(n));
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

// pa (nullable): the chunk prefix rides in the histogram launch as one more workgroup.
template <typename KeyT>
int launch_sort_pass(ydc_context* c, const SortIn<KeyT>& in_, uint32_t n_tiles, void* out_keys,
                     bool out_u32, uint32_t* out_vals, const PrefixArgs* pa = nullptr,
                     bool have_hist = false) {
  SortIn<KeyT> in = in_;
  in.xcd_hist = (c->opt_xcd >> 1) & 1;
  in.xcd_scatter = (c->opt_xcd >> 2) & 1;
  in.dbg = 0;
  const uint32_t radix = 1u << in.bits;
  if (!have_hist)  // (the first pass's tile histograms come out of k_slot_gen)
    YDC_LAUNCH(c, "k_radix_hist", k_radix_hist<KeyT>, dim3(xcd_grid(n_tiles) + (pa ? 1 : 0)), dim3(kSortThreads),
               radix * 4, c->stream, in, c->d_prm.p, n_tiles, c->d_hist.p, pa ? *pa : PrefixArgs{});
  YDC_LAUNCH(c, "k_radix_scan", k_radix_scan, dim3(radix), dim3(256), 0, c->stream, n_tiles,
             c->d_hist.p, c->d_row_total.p);
  const size_t lds = (size_t)(kSortWaves + 1) * radix * 4;
  if (in.fused_cls_bits) {
    const size_t lds2 = lds + (size_t)(kSortWaves + 1) * (radix >> in.fused_cls_bits) * 4;
    YDC_LAUNCH(c, "k_radix_scatter", (k_radix_scatter_classed<KeyT>), dim3(xcd_grid(n_tiles)),
               dim3(kSortThreads), lds2, c->stream, in, c->d_prm.p, n_tiles, c->d_hist.p,
               c->d_row_total.p, (uint32_t*)out_keys, out_vals, c->d_rank_to_g.p);
  } else if (out_u32) {
    YDC_LAUNCH(c, "k_radix_scatter", (k_radix_scatter<KeyT, uint32_t>), dim3(xcd_grid(n_tiles)),
               dim3(kSortThreads), lds, c->stream, in, c->d_prm.p, n_tiles, c->d_hist.p,
               c->d_row_total.p, (uint32_t*)out_keys, out_vals);
  } else {
    YDC_LAUNCH(c, "k_radix_scatter", (k_radix_scatter<KeyT, uint64_t>), dim3(xcd_grid(n_tiles)),
               dim3(kSortThreads), lds, c->stream, in, c->d_prm.p, n_tiles, c->d_hist.p,
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
  return ctx ? ctx->last_error.c_str() : create_error_cstr();
}

int ydc_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_create_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
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
    set_create_error(std::string("hipMalloc: ") + hipGetErrorString(e));
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
uint32_t ydc_abi_version(void) { return 6; }

int ydc_create(int device, uint32_t max_servants, uint32_t max_tasks, uint32_t max_slots,
               void* stream, ydc_context** out) {
  if (!out) return YDC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int n_dev = 0;
  hipError_t de = hipGetDeviceCount(&n_dev);
  if (de != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
    set_create_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(de) + ", " +
                     std::to_string(n_dev) + " device(s), asked for " + std::to_string(device));
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
      hipHostMalloc((void**)&c->h_prm, sizeof(DeviceParams), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->d_h_prm, c->h_prm, 0) != hipSuccess ||
      c->d_row_total.reserve(1u << kMaxRadixBits) != hipSuccess) {
    ydc_destroy(c);
    return YDC_ERR_HIP;
  }
  (void)hipMemset(c->d_prm.p, 0, sizeof(DeviceParams));  // batch_seq starts at 0
  for (auto& e : c->ev) {
    if (hipEventCreate(&e) != hipSuccess) {
      ydc_destroy(c);
      return YDC_ERR_HIP;
    }
  }
  if (const char* s = tune_value("debug_sim")) c->debug_sim = atoi(s) != 0;
  if (const char* s = tune_value("chunk_size")) c->opt_chunk_size = (uint32_t)atoi(s);
  if (const char* s = tune_value("target_chunks")) c->opt_target_chunks = (uint32_t)atoi(s);
  if (const char* s = tune_value("fused_class")) c->opt_fused_class = atoi(s) != 0;
  if (const char* s = tune_value("own_guess")) c->opt_own_guess = atoi(s) != 0;
  if (const char* s = tune_value("pair")) c->opt_pair = atoi(s) != 0;
  if (const char* s = tune_value("ring_total")) c->opt_ring_total = std::max(256u, (uint32_t)atoi(s));
  if (const char* s = tune_value("dense")) c->opt_dense = atoi(s) != 0;
  if (const char* s = tune_value("split_gen")) c->opt_split_gen = atoi(s) != 0;
  if (const char* s = tune_value("xcd_tiles")) c->opt_xcd = (uint32_t)atoi(s);
  if (const char* s = tune_value("tile_tab")) c->opt_tile_tab = atoi(s) != 0;
  if (const char* s = tune_value("group_walk")) c->opt_group_walk = atoi(s) != 0;
  if (const char* s = tune_value("walk_packed")) c->opt_walk_packed = atoi(s) != 0;
  if (const char* s = tune_value("zone_guess")) c->opt_zone_guess = (uint32_t)std::max(0, atoi(s));
  if (const char* s = tune_value("zone_lead")) c->opt_zone_lead = (uint32_t)atoi(s);
  if (const char* s = tune_value("zone_trail")) c->opt_zone_trail = (uint32_t)atoi(s);
  if (const char* s = tune_value("zone_max_chunks")) c->opt_zone_max_chunks = (uint32_t)atoi(s);
  if (const char* s = tune_value("scan_multi")) c->opt_scan_multi = atoi(s) != 0;
  if (const char* s = tune_value("classify_per_thread")) c->opt_classify_multi = atoi(s) != 1;
  if (const char* s = tune_value("packed_class")) c->opt_packed_class = atoi(s) != 0;
  if (const char* s = tune_value("shard_sort")) c->opt_shard_sort = atoi(s) != 0;
  if (const char* s = tune_value("packed_sort")) c->opt_packed_sort = atoi(s) != 0;
  if (const char* s = tune_value("binsort")) c->opt_binsort = atoi(s) != 0;
  if (const char* s = tune_value("stream_graph")) c->opt_stream_graph = atoi(s) != 0;
  if (const char* s = tune_value("walk_park")) c->opt_walk_park = atoi(s) != 0;
  if (const char* s = tune_value("fuse_passes")) c->opt_fuse_passes = atoi(s) != 0;
  if (const char* s = tune_value("warm_up")) c->opt_warm_up = (uint32_t)std::min(64, std::max(1, atoi(s)));
  if (const char* s = tune_value("hand_tries")) c->opt_hand_tries = (uint32_t)std::max(0, atoi(s));
  if (const char* s = tune_value("cp_every")) {
    uint32_t v = (uint32_t)std::max(1, atoi(s)), p2 = 1;
    while (p2 * 2 <= v && p2 < 1024) p2 *= 2;
    c->opt_cp_every = p2;
  }
  if (const char* s = tune_value("level_tab")) c->opt_level_tab = atoi(s) != 0;
  if (const char* s = tune_value("wide")) c->opt_wide = atoi(s) != 0;
  if (const char* s = tune_value("walk_prefetch")) c->opt_walk_prefetch = atoi(s) != 0;
  if (const char* s = tune_value("wide_lists")) c->opt_wide_lists = atoi(s) != 0;
  if (const char* s = tune_value("group_binsort")) c->opt_group_binsort = atoi(s) != 0;
  if (const char* s = tune_value("zero_copy")) c->opt_zero_copy = atoi(s) != 0;
  if (const char* s = tune_value("host_in")) c->opt_host_in_map = std::string(s) != "copy";
  if (const char* s = tune_value("binsort_verify")) c->debug_verify_binsort = atoi(s) != 0;
  if (const char* s = tune_value("binsort_max_slots")) c->opt_binsort_max_slots = (uint32_t)atoll(s);
  if (const char* s = tune_value("shard_margin")) c->opt_shard_margin = atoll(s);
  if (const char* s = tune_value("small_batch")) c->opt_small_batch = (uint32_t)std::max(0ll, atoll(s));
  if (const char* s = tune_value("resident")) c->opt_resident = atoi(s) != 0;
  if (const char* s = tune_value("packed_tick")) c->opt_tick_packed = atoi(s) != 0;
  if (const char* s = tune_value("resident_idle_ms")) c->opt_resident_idle_ms = (uint32_t)std::max(1, atoi(s));
  if (const char* s = tune_value("commit_swap")) c->opt_commit_swap = atoi(s) != 0;
  if (const char* s = tune_value("outcome_store")) c->opt_outcome_store = atoi(s) != 0;
  if (const char* s = tune_value("stream_zero_copy")) c->opt_stream_zero_copy = atoi(s) != 0;
  if (const char* s = tune_value("release_counted")) c->opt_release_counted = atoi(s) != 0;
  if (const char* s = tune_value("walk_after")) c->opt_walk_after = (uint32_t)std::max(2, atoi(s));
  if (const char* s = tune_value("rounds_per_check"))
    c->opt_rounds_per_check = std::max(1, atoi(s));
  *out = c;
  return YDC_OK;
}

int ydc_destroy(ydc_context* c) {
  if (!c) return YDC_OK;
  (void)hipSetDevice(c->device);
  resident_stop(c);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  stream_release(c);
  group_release(c);
  for (auto* b : {&c->d_version, &c->d_nproc, &c->d_load, &c->d_max_tasks, &c->d_running,
                  &c->d_flags, &c->d_class_of, &c->d_ip_hash, &c->d_ip_filter, &c->d_bin_tile_start, &c->d_bin_tile_base, &c->d_ip_sorted, &c->d_ip_servant, &c->d_cls_ver,
                  &c->d_cls_comp, &c->d_part_base,
                  &c->d_slot_base, &c->d_cls_begin, &c->d_vals[0], &c->d_vals[1], &c->d_hist, &c->d_tile_first,
                  &c->d_row_total, &c->d_self_lo, &c->d_self_hi, &c->d_chunk_consuming,
                  &c->d_before, &c->d_slot_of, &c->d_pos_last, &c->d_running_out, &c->d_out_idx,
                  &c->d_upd_idx})
    b->release();
  for (auto* b : {&c->d_cls_env, &c->d_env_ver_mask, &c->d_keys[0], &c->d_keys[1], &c->d_mask}) b->release();
  c->d_ver_sorted.release();
  c->d_cls_single.release();
  for (auto& b : c->d_spare) b.release();
  c->d_cls_by_g.release();
  c->d_owner.release();
  c->d_rank_to_g.release();
  c->d_zone_box.release();
  c->d_guess[0].release();
  c->d_endst.release();
  c->d_checkpoint.release();
  c->d_early.release();
  c->d_claim.release();
  c->d_runs.release();
  c->d_dirty.release();
  c->d_prm.release();
  if (c->h_prm) (void)hipHostFree(c->h_prm);
  for (auto& pd : c->pend) {
    if (pd.h_outcome) (void)hipHostFree(pd.h_outcome);
    if (pd.ev) (void)hipEventDestroy(pd.ev);
  }
  if (c->h_in) (void)hipHostFree(c->h_in);
  if (c->h_rel) (void)hipHostFree(c->h_rel);
  if (c->h_rel_ev) (void)hipEventDestroy(c->h_rel_ev);
  if (c->copy_ev) (void)hipEventDestroy(c->copy_ev);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->h_res) (void)hipHostFree(c->h_res);
  if (c->h_box) (void)hipHostFree(c->h_box);
  if (c->res_ev) (void)hipEventDestroy(c->res_ev);
  if (c->res_stream) (void)hipStreamDestroy(c->res_stream);
  if (c->h_tick_done) (void)hipHostFree(c->h_tick_done);
  if (c->h_tick_io) (void)hipHostFree(c->h_tick_io);
  c->d_ip.release();
  c->d_in.release();
  c->d_res.release();
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
  if (sv && sv->env_words > YDC_MAX_ENV_WORDS)
    return fail(c, YDC_ERR_INVALID_ARGUMENT, "env_words %u > %u", sv->env_words, YDC_MAX_ENV_WORDS);
  if (c->max_servants && n > c->max_servants)
    return fail(c, YDC_ERR_CAPACITY, "%u servants > max_servants %u", n, c->max_servants);
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // (released slots may still be on their way)
  if (int rc = reserve_registry(c, n)) return rc;
  c->n_servants = n;
  c->h_alias_ip.clear();  // (they name rows of the table that is being replaced)
  c->h_alias_servant.clear();
  c->h_version.assign(sv ? sv->version : nullptr, sv ? sv->version + n : nullptr);
  c->h_nproc.assign(sv ? sv->num_processors : nullptr, sv ? sv->num_processors + n : nullptr);
  c->h_load.assign(sv ? sv->current_load : nullptr, sv ? sv->current_load + n : nullptr);
  c->h_max_tasks.assign(sv ? sv->max_tasks : nullptr, sv ? sv->max_tasks + n : nullptr);
  c->h_flags.assign(sv ? sv->flags : nullptr, sv ? sv->flags + n : nullptr);
  c->h_ip.assign(sv ? sv->ip_id : nullptr, sv ? sv->ip_id + n : nullptr);
  c->env_words = sv && sv->env_words ? sv->env_words : 1;
  c->h_env.assign(sv ? sv->env_mask : nullptr, sv ? sv->env_mask + (size_t)n * c->env_words : nullptr);
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

namespace {
// Re-lays the host copy of the environment masks out for `words` words per servant (the
// device only holds per-class masks, which rebuild_tables derives from this copy).
void widen_env(ydc_context* c, uint32_t words) {
  if (words <= c->env_words) return;
  std::vector<uint64_t> wide((size_t)c->n_servants * words, 0);
  for (uint32_t s = 0; s < c->n_servants; ++s)
    for (uint32_t w = 0; w < c->env_words; ++w)
      wide[(size_t)s * words + w] = c->h_env[(size_t)s * c->env_words + w];
  c->h_env.swap(wide);
  c->env_words = words;
  c->tables_dirty = true;
}
}  // namespace

int ydc_update_servants_wide(ydc_context* c, const uint32_t* idx, const ydc_servant_row* rows,
                             const uint64_t* env_masks, uint32_t env_words, uint32_t n) {
  if (!c || (n && (!idx || !rows))) return YDC_ERR_INVALID_ARGUMENT;
  if (env_masks && (env_words == 0 || env_words > YDC_MAX_ENV_WORDS))
    return fail(c, YDC_ERR_INVALID_ARGUMENT, "env_words %u out of range", env_words);
  // A row carries one mask word: on a wider table it would silently clear the servant's other
  // environments (KeepServantAlive replaces the whole set, task_dispatcher.cc:195-201).
  if (!env_masks && n && c->env_words > 1)
    return fail(c, YDC_ERR_INVALID_ARGUMENT,
                "the table holds %u mask words per servant: use ydc_update_servants_wide", c->env_words);
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  // Appends first (they may need bigger buffers).
  uint32_t new_n = c->n_servants;
  for (uint32_t i = 0; i < n; ++i) {
    if (idx[i] > new_n) return fail(c, YDC_ERR_INVALID_ARGUMENT, "servant index %u out of order", idx[i]);
    if (idx[i] == new_n) ++new_n;
  }
  if (c->max_servants && new_n > c->max_servants)
    return fail(c, YDC_ERR_CAPACITY, "%u servants > max_servants %u", new_n, c->max_servants);
  if (env_masks) widen_env(c, env_words);
  const uint32_t EW = c->env_words;
  auto resize_host = [&](uint32_t m) {
    c->h_version.resize(m);
    c->h_nproc.resize(m);
    c->h_load.resize(m);
    c->h_max_tasks.resize(m);
    c->h_flags.resize(m);
    c->h_ip.resize(m);
    c->h_env.resize((size_t)m * EW);
  };
  if (new_n > c->d_version.cap) {
    // Grow: read the running column back, reallocate, re-upload everything. Released slots
    // (ydc_release_slots only enqueues) must have reached the column first.
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<uint32_t> run(c->n_servants);
    if (c->n_servants)
      HIP_TRY(c, hipMemcpy(run.data(), c->d_running.p, c->n_servants * 4, hipMemcpyDeviceToHost));
    run.resize(new_n, 0);
    if (int rc = reserve_registry(c, std::max<uint32_t>(new_n, new_n + new_n / 2))) return rc;
    resize_host(new_n);
    HIP_TRY(c, hipMemcpy(c->d_running.p, run.data(), new_n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_version.p, c->h_version.data(), new_n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_nproc.p, c->h_nproc.data(), new_n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_load.p, c->h_load.data(), new_n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_max_tasks.p, c->h_max_tasks.data(), new_n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_flags.p, c->h_flags.data(), new_n * 4, hipMemcpyHostToDevice));
  } else if (new_n > c->n_servants) {
    resize_host(new_n);
    HIP_TRY(c, hipMemsetAsync(c->d_running.p + c->n_servants, 0, (new_n - c->n_servants) * 4, c->stream));
  }
  bool structural = new_n != c->n_servants;
  c->n_servants = new_n;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t s = idx[i];
    const ydc_servant_row& r = rows[i];
    uint64_t* env = &c->h_env[(size_t)s * EW];
    bool env_changed = false;
    for (uint32_t w = 0; w < EW; ++w) {
      const uint64_t m = env_masks ? (w < env_words ? env_masks[(size_t)i * env_words + w] : 0)
                                   : (w == 0 ? r.env_mask : 0);
      env_changed |= env[w] != m;
      env[w] = m;
    }
    structural |= env_changed || c->h_version[s] != r.version ||
                  c->h_ip[s] != r.ip_id || (c->h_max_tasks[s] == 0) != (r.max_tasks == 0) ||
                  std::min(c->h_max_tasks[s], c->h_nproc[s]) != std::min(r.max_tasks, r.num_processors);
    c->h_version[s] = r.version;
    c->h_nproc[s] = r.num_processors;
    c->h_load[s] = r.current_load;
    c->h_max_tasks[s] = r.max_tasks;
    c->h_flags[s] = r.flags;
    c->h_ip[s] = r.ip_id;
  }
  if (n) {
    // One staged copy of the update list + one scatter kernel (rows are SoA on the device).
    static_assert(sizeof(ydc_servant_row) == sizeof(ServantRowDev), "row layout");
    HIP_TRY(c, c->d_upd_idx.reserve(n));
    HIP_TRY(c, c->d_upd_rows.reserve(n));
    HIP_TRY(c, hipMemcpyAsync(c->d_upd_idx.p, idx, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_upd_rows.p, rows, (size_t)n * sizeof(ydc_servant_row),
                              hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_apply_rows, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, c->d_upd_idx.p,
                       (const ServantRowDev*)c->d_upd_rows.p, n, c->n_servants, c->d_version.p,
                       c->d_nproc.p, c->d_load.p, c->d_max_tasks.p, c->d_flags.p);
    HIP_TRY(c, hipGetLastError());
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (structural || c->tables_dirty) return rebuild_tables(c);
  return YDC_OK;
}

int ydc_update_servants(ydc_context* c, const uint32_t* idx, const ydc_servant_row* rows,
                        uint32_t n) {
  return ydc_update_servants_wide(c, idx, rows, nullptr, 1, n);
}

int ydc_set_host_aliases(ydc_context* c, const uint32_t* ip_id, const uint32_t* servant_idx, uint32_t n) {
  if (!c || (n && (!ip_id || !servant_idx))) return YDC_ERR_INVALID_ARGUMENT;
  for (uint32_t i = 0; i < n; ++i)
    if (servant_idx[i] >= c->n_servants)
      return fail(c, YDC_ERR_INVALID_ARGUMENT, "alias %u names servant %u of %u", i, servant_idx[i], c->n_servants);
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->h_alias_ip.assign(ip_id, ip_id + n);
  c->h_alias_servant.assign(servant_idx, servant_idx + n);
  return rebuild_tables(c);
}

int ydc_remove_servants(ydc_context* c, const uint32_t* idx, uint32_t n) {
  if (!c || (n && !idx)) return YDC_ERR_INVALID_ARGUMENT;
  if (!n) return YDC_OK;
  for (uint32_t i = 0; i < n; ++i)
    if (idx[i] >= c->n_servants || (i && idx[i] <= idx[i - 1]))
      return fail(c, YDC_ERR_INVALID_ARGUMENT, "removed rows must be ascending and < %u", c->n_servants);
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  const uint32_t S = c->n_servants, kept = S - n, EW = c->env_words;
  // Device: order-preserving compaction of the six resident columns into spare buffers,
  // which then take their place (running_tasks of the survivors never leaves the device).
  HIP_TRY(c, c->d_upd_idx.reserve(n));
  for (auto& b : c->d_spare) HIP_TRY(c, b.reserve(c->d_version.cap));
  HIP_TRY(c, hipMemcpyAsync(c->d_upd_idx.p, idx, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  CompactCols in{{c->d_version.p, c->d_nproc.p, c->d_load.p, c->d_max_tasks.p, c->d_running.p, c->d_flags.p}};
  CompactCols out{{c->d_spare[0].p, c->d_spare[1].p, c->d_spare[2].p, c->d_spare[3].p, c->d_spare[4].p,
                   c->d_spare[5].p}};
  hipLaunchKernelGGL(k_compact_rows, dim3(ceil_div(S, 256)), dim3(256), 0, c->stream, in, out,
                     c->d_upd_idx.p, n, S);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // (idx is pageable; the swap below retires the old columns)
  DevBuf<uint32_t>* cols[6] = {&c->d_version, &c->d_nproc, &c->d_load, &c->d_max_tasks, &c->d_running,
                               &c->d_flags};
  for (int k = 0; k < 6; ++k) std::swap(*cols[k], c->d_spare[k]);
  // Host mirror.
  uint32_t w = 0, next = 0;
  for (uint32_t s = 0; s < S; ++s) {
    if (next < n && idx[next] == s) {
      ++next;
      continue;
    }
    if (w != s) {
      c->h_version[w] = c->h_version[s];
      c->h_nproc[w] = c->h_nproc[s];
      c->h_load[w] = c->h_load[s];
      c->h_max_tasks[w] = c->h_max_tasks[s];
      c->h_flags[w] = c->h_flags[s];
      c->h_ip[w] = c->h_ip[s];
      for (uint32_t e = 0; e < EW; ++e) c->h_env[(size_t)w * EW + e] = c->h_env[(size_t)s * EW + e];
    }
    ++w;
  }
  c->n_servants = kept;
  c->h_alias_ip.clear();  // (row numbers moved)
  c->h_alias_servant.clear();
  c->h_version.resize(kept);
  c->h_nproc.resize(kept);
  c->h_load.resize(kept);
  c->h_max_tasks.resize(kept);
  c->h_flags.resize(kept);
  c->h_ip.resize(kept);
  c->h_env.resize((size_t)kept * EW);
  return rebuild_tables(c);  // classes, the ip table and the slot bound follow the registry
}

// FreeTask's --running_tasks for a list of grants (servant indexes on the device).
static void launch_release(ydc_context* c, const uint32_t* d_idx, uint32_t n) {
  if (n >= 4096 && c->n_servants <= 16384 && c->opt_release_counted) {
    if ((size_t)c->n_servants * 4 > 48 * 1024)  // (above 48 KB of dynamic LDS the runtime wants to be told)
      (void)hipFuncSetAttribute((const void*)k_release_slots_counted, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)c->n_servants * 4));
    hipLaunchKernelGGL(k_release_slots_counted, dim3(ceil_div(n, kReleaseTile)), dim3(1024),
                       (size_t)c->n_servants * 4, c->stream, d_idx, n, c->n_servants, c->d_running.p);
  } else {
    hipLaunchKernelGGL(k_release_slots, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, d_idx, n,
                       c->n_servants, c->d_running.p);
  }
}

int ydc_release_slots(ydc_context* c, const uint32_t* servant_idx, uint32_t n) {
  if (!c || (n && !servant_idx)) return YDC_ERR_INVALID_ARGUMENT;
  if (!n) return YDC_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  HIP_TRY(c, c->d_upd_idx.reserve(n));
  // Through a pinned staging buffer, stream-ordered: no wait here (the next batch follows on
  // the same stream). The buffer is reused only after its previous copy has run.
  if (c->h_rel_ev) HIP_TRY(c, hipEventSynchronize(c->h_rel_ev));
  if ((size_t)n * 4 > c->h_rel_cap) {
    if (c->h_rel) (void)hipHostFree(c->h_rel);
    c->h_rel = nullptr;
    c->h_rel_cap = 0;
    const size_t want = std::max<size_t>((size_t)n * 6, 4096);
    HIP_TRY(c, hipHostMalloc((void**)&c->h_rel, want));
    c->h_rel_cap = want;
  }
  if (!c->h_rel_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->h_rel_ev, hipEventDisableTiming));
  std::memcpy(c->h_rel, servant_idx, (size_t)n * 4);
  HIP_TRY(c, hipMemcpyAsync(c->d_upd_idx.p, c->h_rel, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipEventRecord(c->h_rel_ev, c->stream));
  launch_release(c, c->d_upd_idx.p, n);
  HIP_TRY(c, hipGetLastError());
  return YDC_OK;
}

int ydc_release_slots_device(ydc_context* c, const uint32_t* d_servant_idx, uint32_t n) {
  if (!c || (n && !d_servant_idx)) return YDC_ERR_INVALID_ARGUMENT;
  if (!n) return YDC_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  launch_release(c, d_servant_idx, n);
  HIP_TRY(c, hipGetLastError());
  return YDC_OK;
}

int ydc_set_running(ydc_context* c, const uint32_t* running, uint32_t n) {
  if (!c || n != c->n_servants || (n && !running)) return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // (released slots may still be on their way)
  if (n) HIP_TRY(c, hipMemcpy(c->d_running.p, running, n * 4, hipMemcpyHostToDevice));
  return YDC_OK;
}

int ydc_get_running(ydc_context* c, uint32_t* out, uint32_t n) {
  if (!c || n != c->n_servants || (n && !out)) return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // (released slots may still be on their way)
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (n) HIP_TRY(c, hipMemcpy(out, c->d_running.p, n * 4, hipMemcpyDeviceToHost));
  return YDC_OK;
}

}  // extern "C" (reopened below)

// ---------------------------------------------------------------------------
// One batch = plan (sizes, workspace) + front (slots, sort, class lists, request
// classification, level guesses) + matching passes + finalise. The pieces only
// enqueue work on the context stream, so the streaming mode can capture them
// into a hipGraph; ydc_dispatch_device strings them together eagerly.
// ---------------------------------------------------------------------------
namespace {

// for_window: the plan of a multi-GPU batch whose slot sort is sharded (k_window generates only a
// key window of the slots: radix pipeline).
bool zone_decide(ydc_context* c, uint32_t N, uint32_t K, uint32_t C);

int plan_batch(ydc_context* c, uint32_t N, BatchPlan* out, bool for_window = false) {
  BatchPlan& p = *out;
  p = BatchPlan{};
  if (c->tables_dirty)
    if (int rc = rebuild_tables(c)) return rc;
  p.N = N;
  p.S = c->n_servants;
  p.C = c->tables.n_classes();
  p.W = std::max<uint32_t>(1, ceil_div(p.C, 64));
  const uint64_t slot_bound64 = c->tables.max_slots;
  if (slot_bound64 > 0xFFFFFFF0ull || (c->max_slots && slot_bound64 > c->max_slots))
    return fail(c, YDC_ERR_CAPACITY, "registry can offer %llu slots > max_slots %u",
                (unsigned long long)slot_bound64, c->max_slots);
  p.slot_bound = p.slot_bound_glob = (uint32_t)slot_bound64;
  // Sort tiles: 256 threads x `items` elements; fewer elements per thread while that still
  // leaves the chip short of workgroups (the passes are latency-bound at this size).
  p.sort_items = p.slot_bound <= 300000 ? 2 : (p.slot_bound <= 700000 ? 4 : 8);
  if (const char* e = tune_value("sort_items")) p.sort_items = std::min(8, std::max(1, atoi(e)));
  p.n_tiles = std::max<uint32_t>(1, ceil_div(p.slot_bound, kSortThreads * p.sort_items));
  p.any_shared = c->tables.any_shared_ip;
  p.use_generic = p.C > kMaxWaveClasses;
  p.wave_path = N && p.C && !p.use_generic;
  p.cs = c->opt_chunk_size;
  if (p.cs == 0) {
    uint32_t want = ceil_div(std::max<uint32_t>(N, 1), std::max<uint32_t>(1, c->opt_target_chunks));
    p.cs = 64;
    while (p.cs < want && p.cs < 8192) p.cs <<= 1;
    // Big batches: twice the chunks while they stay at 1024 requests or more — four waves per
    // SIMD instead of two (cfg4, 4M requests: k_match_pass 373 -> 294 us). Shorter chunks cost
    // more in wrong guesses and replays than the occupancy brings (cfg3 at 256 instead of 512
    // requests per chunk: 395 -> 465 us).
    if (c->opt_dense && p.W == 1 && p.cs >= 2048) p.cs >>= 1;
  }
  // The many-class kernels replay whole chunks per round (no checkpoints): chunks long enough
  // for a wrong start to heal inside them keep the rounds few (cfg2 with 10 digests, 947 classes:
  // 16 / 10 / 6 / 4 rounds at 64 / 128 / 256 / 512 requests; 3.0 / 2.6 / 2.5 / 2.9 ms).
  if (p.use_generic && c->opt_chunk_size == 0) p.cs = std::max(p.cs, 256u);
  p.K = N ? ceil_div(N, p.cs) : 0;
  const uint32_t C = p.C, K = p.K, W = p.W, slot_bound = p.slot_bound;

  // Workspace.
  p.key32 = c->kf.key_bits <= 32;
  // 32-bit keys: 8-byte (key, value) records in d_keys; 64-bit keys: keys there, values in d_vals.
  HIP_TRY(c, c->d_keys[0].reserve(slot_bound));
  HIP_TRY(c, c->d_keys[1].reserve(slot_bound));
  if (!p.key32 || !c->opt_packed_sort) {
    HIP_TRY(c, c->d_vals[0].reserve(slot_bound));
    HIP_TRY(c, c->d_vals[1].reserve(slot_bound));
  }
  p.packed = p.key32 && c->opt_packed_sort;
  HIP_TRY(c, c->d_hist.reserve(((size_t)1 << kMaxRadixBits) * p.n_tiles));
  HIP_TRY(c, c->d_tile_first.reserve((size_t)p.n_tiles + 2));
  if (C > 1) HIP_TRY(c, c->d_cls_by_g.reserve(slot_bound));
  p.key_passes = c->kf.passes;
  p.cls_passes = 0;
  p.fused_cls_bits = 0;
  p.gbits = 0;
  if (C > 1 && slot_bound) {
    uint32_t cls_bits = 1;
    while ((1u << cls_bits) < C) ++cls_bits;
    uint32_t gb = 1;
    while (gb < 32 && (slot_bound >> gb)) ++gb;
    if (gb + cls_bits <= 32 && c->opt_packed_class) p.gbits = gb;  // the class rides above the slot index
    // A handful of classes and room left in the last key digit: one pass does both.
    const uint32_t last_bits = c->kf.key_bits - (p.key_passes - 1) * c->kf.bits_per_pass;
    if (C <= kMaxFusedClasses && p.key_passes >= 1 && last_bits + cls_bits <= (uint32_t)kMaxRadixBits &&
        c->opt_fused_class) {
      p.fused_cls_bits = cls_bits;
      HIP_TRY(c, c->d_rank_to_g.reserve(slot_bound));
    } else {
      p.cls_passes = ceil_div(cls_bits, kMaxRadixBits);
      p.cls_bits = ceil_div(cls_bits, p.cls_passes);
    }
  }
  // Bin sort: 32-bit exact keys, the wave path's class limit, the class above the slot bits in
  // the value (or a single class), the whole slot order (a rank of a group takes it as well
  // unless it only sorts a key window). Decided from the registry alone.
  p.binsort = c->opt_binsort && !c->binsort_blocked && c->kf.exact && p.key32 && C >= 1 &&
              C <= kMaxWaveClasses && slot_bound && slot_bound <= c->opt_binsort_max_slots &&
              (C == 1 || p.gbits) && !for_window && p.S <= kBinMaxServants &&
              c->kf.cap_bits <= 11;
  if (p.binsort) {
    const BinFormat bf = choose_bins(c->kf.key_bits, slot_bound, kMaxBins);
    p.n_bins = bf.n_bins;
    p.bin_shift = bf.shift;
    // k_bin_sort keeps a record in one LDS word: key bits below the bin | slot | class.
    p.bin_slot_bits = 1;
    while (p.bin_slot_bits < 32 && (slot_bound >> p.bin_slot_bits)) ++p.bin_slot_bits;
    p.bin_cls_bits = 0;
    while ((1u << p.bin_cls_bits) < C) ++p.bin_cls_bits;
    // (more, narrower bins when the word is short of room for the key bits below the bin)
    while (p.bin_shift + p.bin_slot_bits + p.bin_cls_bits > 32 && p.n_bins < kMaxBins && p.bin_shift) {
      p.n_bins <<= 1;
      --p.bin_shift;
    }
    // Slot tiles: G consecutive servants, at most 2048 slots (a servant offers < 2^cap_bits).
    // (Smaller tiles were measured: more workgroups that each add up the servants before them
    // cost more than the fullest tile's shorter loop saves.)
    // Slot tiles: runs of consecutive servants cut by the host to about equal slot bounds
    // (host_tables.h: bin_tile_start; at most kBinMaxGroup servants and 2048 slots each).
    p.bin_group = kBinMaxGroup;
    p.bin_tiles = (uint32_t)c->tables.bin_tile_start.size() - 1;
    if (p.bin_shift + p.bin_slot_bits + p.bin_cls_bits > 32 || p.bin_tiles > kBinMaxTiles) p.binsort = false;
  }
  if (p.binsort) {
    p.cls_passes = 0;
    p.fused_cls_bits = 0;
    HIP_TRY(c, c->d_binbase.reserve((size_t)(p.n_bins + 1) * (C + 1)));
    HIP_TRY(c, c->d_binruns.reserve((size_t)p.bin_tiles * p.n_bins));
    HIP_TRY(c, c->d_rank_to_g.reserve(slot_bound));
    if (c->opt_level_tab) HIP_TRY(c, c->d_level_tab.reserve(((size_t)(slot_bound >> 6) + 2) * C));
    // (consuming counts per wave of 64 requests instead of per chunk)
    HIP_TRY(c, c->d_chunk_consuming.reserve(((size_t)ceil_div(std::max(N, 1u), 64) + 1) * c->n_parts));
  }
  HIP_TRY(c, c->d_owner.reserve(slot_bound));
  HIP_TRY(c, c->d_mask.reserve((size_t)N * W));
  HIP_TRY(c, c->d_self_lo.reserve(N));
  HIP_TRY(c, c->d_self_hi.reserve(N));
  HIP_TRY(c, c->d_slot_of.reserve(N));
  HIP_TRY(c, c->d_chunk_consuming.reserve(((size_t)K + 1) * c->n_parts));
  HIP_TRY(c, c->d_before.reserve(((size_t)K + 2) * c->n_parts));
  HIP_TRY(c, c->d_dirty.reserve((size_t)K + 1));
  HIP_TRY(c, c->d_guess[0].reserve((size_t)K * C + 1));
  HIP_TRY(c, c->d_endst.reserve((size_t)K * C + 1));
  if (!p.use_generic) {
    HIP_TRY(c, c->d_checkpoint.reserve((size_t)ceil_div(std::max(N, 1u), 64) * C + 1));
    HIP_TRY(c, c->d_early.reserve((size_t)K * C + 1));
    if ((size_t)K + 1 > c->d_claim.cap) {
      // Claims compare against ever-growing (batch, pass) stamps: a fresh array starts at 0.
      HIP_TRY(c, c->d_claim.reserve((size_t)K + 1));
      HIP_TRY(c, hipMemsetAsync(c->d_claim.p, 0, c->d_claim.cap * 8, c->stream));
    }
  }
  if (p.use_generic && !(C <= kMaxWideClasses && c->opt_wide)) HIP_TRY(c, c->d_runs.reserve((size_t)K * C + 1));
  // Sparse eligibility (every (digest, version threshold) row names at most 64 classes): the wide
  // kernel reads a request's classes from its row instead of scanning C / 64 mask words.
  p.wide_lists = p.use_generic && C <= kMaxWideClasses && c->opt_wide && c->opt_wide_lists &&
                 !c->tables.elig_off.empty() && c->tables.elig_max_len <= 64;
  if (p.wide_lists) HIP_TRY(c, c->d_row_of.reserve(std::max(N, 1u)));

  p.sv = ServantTable{c->d_version.p, c->d_nproc.p,  c->d_load.p,     c->d_max_tasks.p,
                      c->d_running.p, c->d_flags.p, c->d_class_of.p, p.S};
  // The sort ping-pongs between the two key/value buffers: where the lists end up.
  const int cur = (int)(((slot_bound ? p.key_passes : 0) + p.cls_passes) & 1);
  const int key_sorted = (int)((slot_bound ? p.key_passes : 0) & 1);
  p.L.n_classes = C;
  p.L.cls_begin = c->d_cls_begin.p;
  p.L.cls_single = c->d_cls_single.p;
  if (p.packed) {
    const uint32_t* rec = (const uint32_t*)c->d_keys[cur].p;  // {rank, slot} pairs
    p.L.list_p = p.cls_passes || p.fused_cls_bits ? rec : nullptr;
    p.L.list_g = rec + 1;
    p.L.stride = 2;
    p.rank_to_g = p.fused_cls_bits ? c->d_rank_to_g.p : (const uint32_t*)c->d_keys[key_sorted].p + 1;
    p.rank_stride = p.fused_cls_bits ? 1 : 2;
  } else {
    p.L.list_p = p.cls_passes || p.fused_cls_bits ? (const uint32_t*)c->d_keys[cur].p : nullptr;
    p.L.list_g = c->d_vals[cur].p;
    p.L.stride = 1;
    p.rank_to_g = p.fused_cls_bits ? c->d_rank_to_g.p : c->d_vals[key_sorted].p;
    p.rank_stride = 1;
  }
  if (p.binsort) {
    // Staging records in d_keys[0], the class lists ({rank, slot} records) in d_keys[1].
    const uint32_t* rec = (const uint32_t*)c->d_keys[1].p;
    p.L.list_p = rec;
    p.L.list_g = rec + 1;
    p.L.stride = 2;
    p.rank_to_g = c->d_rank_to_g.p;
    p.rank_stride = 1;
  }
  p.T = TaskTable{c->d_mask.p, c->d_self_lo.p, c->d_self_hi.p, W};
  p.shared = SharedIpTable{c->d_ip_sorted.p, c->d_ip_servant.p, (uint32_t)c->tables.ip_sorted.size(),
                           c->d_class_of.p, c->d_slot_base.p, p.S, p.any_shared ? c->d_pos_last.p : nullptr};
  p.mb = MatchBuffers{};
  if (p.wave_path) {
    p.mb.guess0 = c->d_guess[0].p;
    p.mb.endst = c->d_endst.p;
    p.mb.checkpoint = c->d_checkpoint.p;
    // <= 64 classes on one GPU: pass 0 computes its level guesses itself (no k_guess_init).
    p.mb.before = p.W == 1 && c->group.n_ranks <= 1 && c->opt_own_guess ? c->d_before.p : nullptr;
    p.mb.cls_comp = c->d_cls_comp.p;
    p.mb.part_rank_base = c->d_part_base.p;
    p.mb.n_parts = c->n_parts;
    p.mb.early = c->d_early.p;
    p.mb.cp_every = c->opt_cp_every;
    p.mb.claim = c->d_claim.p;
    p.mb.slot_of = c->d_slot_of.p;
    p.mb.boundary_in = nullptr;
    p.mb.flags = c->d_prm.p->n_changed;
    p.mb.sampled = c->d_prm.p->n_sampled;
    p.mb.flag_mask = 63;
    p.mb.level_tab = p.binsort && c->opt_level_tab && c->n_parts <= 1 ? c->d_level_tab.p : nullptr;
    // The class partition as a pass of its own leaves its scanned histogram table in d_hist:
    // digit == class, one column per sort tile (enqueue_sort; k_radix_scan).
    if (!p.binsort && p.cls_passes == 1 && p.slot_bound && c->opt_tile_tab) {
      p.mb.tile_tab = c->d_hist.p;
      p.mb.tile_tab_tiles = p.n_tiles;
      p.mb.tile_tab_elems = kSortThreads * p.sort_items;
    }
    // (a group of one rank is a single GPU with the exchanges of the protocol around it)
    p.fuse01 = c->opt_fuse_passes && c->group.n_ranks <= 1;
    if (p.fuse01) {
      if ((size_t)K * C * 4 > c->d_hand.cap) {
        // Granules are valid by their batch stamp (never 0): a fresh array starts at 0.
        HIP_TRY(c, c->d_hand.reserve((size_t)K * C * 4));
        HIP_TRY(c, hipMemsetAsync(c->d_hand.p, 0, c->d_hand.cap * 8, c->stream));
      }
      p.mb.hand = c->d_hand.p;
      p.mb.hand_tries = c->opt_hand_tries;
      HIP_TRY(c, c->d_chunk_tail.reserve((size_t)K + 1));
      p.mb.tail = c->n_parts <= 1 && p.mb.before ? c->d_chunk_tail.p : nullptr;
      // Warm-up length: an eighth of the chunk, 16 .. 64 requests (cfg2's chunks of 64: 16 is
      // enough for every chunk; cfg3's chunks of 512: 706 / 300 / 51 chunks still need a second
      // replay with 16 / 32 / 64, and the launch lasts as long as its slowest wave).
      p.mb.warm_len = c->opt_warm_up ? c->opt_warm_up : std::min(64u, std::max(kWarmUp, p.cs / 8));
    }
    // Ring of R = 2^rshift entries per class; a wave's rings hold 2048 entries in all
    // (16 KB of LDS: ranks + generation indexes), see match_kernel.h.
    p.ring_total = c->opt_ring_total;
    // The 4-waves-per-SIMD build of the matching kernel for chunks long enough to amortise its
    // spills (cfg2's chunks of 64 requests: 22.7 -> 23.1 us with it; cfg3's of 512: 426 -> 395),
    // and rings small enough for all of a CU's waves to be resident (160 KB of LDS, 16 waves).
    p.dense = c->opt_dense && p.W == 1 && p.cs >= 256;
    if (p.dense && c->opt_ring_total == 0) {
      const uint32_t per_cu = std::min<uint32_t>(16, std::max<uint32_t>(1, ceil_div(p.K, 256)));
      p.ring_total = 2048;
      while (p.ring_total > 256 && (size_t)per_cu * p.ring_total * 8 > 150 * 1024) p.ring_total >>= 1;
    } else if (c->opt_ring_total == 0) {
      p.ring_total = 2048;
    }
    while (p.ring_total < 2048 && ((size_t)C << 3) > p.ring_total) p.ring_total <<= 1;  // >= 8 per class
    p.rshift = 3;
    while (p.rshift < 10 && ((size_t)C << (p.rshift + 1)) <= p.ring_total) ++p.rshift;
    const uint32_t R = 1u << p.rshift;
    uint32_t want = 2 * p.cs / std::max(C, 1u);
    p.init_fill = 8;
    while (p.init_fill < want && p.init_fill < 64) p.init_fill <<= 1;
    p.init_fill = std::min(std::max(p.init_fill, 32u), R);
    // Few classes: the first fill is one coalesced load per class and 64 entries; take
    // enough for a whole block of 64 requests from one class.
    if (C <= 8) p.init_fill = std::min(128u, R);
    else if (C <= 16) p.init_fill = std::min(64u, R);
    // The stretch where the dedicated tier runs out, walked by one wave beside the first pass
    // (zone_guess.h: workgroup 0 of that launch). Where the passes are one round of
    // latency-bound waves — a chain behind the first launch then costs its full serial time
    // (cfg3: 224 of 384 us) —; a batch of more chunks hides its second replays behind the other
    // waves' work. (The walk's rings — ranks only, 128 entries per class at least — live in the
    // LDS a chunk's wave has.)
    p.zone = c->opt_zone_guess && p.W == 1 && p.mb.before && p.fuse01 && p.mb.tail && c->n_parts <= 1 &&
             !p.binsort && p.packed && c->kf.exact && p.key_passes >= 1 && C > 8 && C <= 64 && p.mb.tile_tab &&
             K >= 64 && K <= c->opt_zone_max_chunks && !for_window && slot_bound && !c->stream_mode.active &&
             ((size_t)C << 7) <= 2 * (size_t)p.ring_total;
    if (p.zone && c->zone_cooldown) {
      --c->zone_cooldown;
      p.zone = false;
    }
    p.zone_eligible = p.zone;
    if (p.zone && c->opt_zone_guess == 1) p.zone = zone_decide(c, N, K, C);
    if (p.zone) {
      const unsigned long long* had = c->d_zone_box.p;
      HIP_TRY(c, c->d_zone_box.reserve(2 + (size_t)kZoneMaxChunks * C));
      if (c->d_zone_box.p != had)  // (a granule is valid when it carries the batch's number: none does yet)
        HIP_TRY(c, hipMemsetAsync(c->d_zone_box.p, 0, c->d_zone_box.cap * 8, c->stream));
      p.mb.zone_box = c->d_zone_box.p;
      p.mb.zone_sorted = (const uint2*)c->d_keys[key_sorted].p;
      p.mb.zone_tier_shift = c->kf.key_bits - 1;
      p.mb.zone_lead = std::max(c->opt_zone_lead, c->zone_lead_cur);
      p.mb.zone_trail = c->opt_zone_trail;
    }
  }
  return YDC_OK;
}

// ---- the pieces of the front: servant scan | slot generation (+ request classification) |
// sort + class lists. enqueue_front_a strings them together; the multi-GPU path with a sharded
// sort puts its key-window selection between them (enqueue_front_windowed).

// Servant scan (also resets the per-batch device counters). cls_begin: where the class sizes
// of the whole registry go.
void enqueue_scan(ydc_context* c, const BatchPlan& p, uint32_t* cls_begin) {
  c->ksamples_used = 0;
  mark(c, 0);
  if (p.binsort) {  // (no scan: k_front_bins does without, see enqueue_gen)
    mark(c, 1);
    return;
  }
  // (one workgroup per slab of 1024 servants from 4k servants on: kernels.h, servant_scan_multi)
  const uint32_t scan_blocks = p.S > 4096 && c->opt_scan_multi ? ceil_div(p.S, 1024) : 1u;
  YDC_LAUNCH(c, "k_servant_scan", k_servant_scan, dim3(scan_blocks), dim3(1024), (p.C + 1) * sizeof(uint32_t),
             c->stream, p.sv, p.C, p.slot_bound_glob, c->d_slot_base.p, cls_begin,
             c->d_chunk_consuming.p, p.K, PartTable{c->d_cls_comp.p, c->n_parts, c->d_part_base.p},
             kSortThreads * p.sort_items, p.win ? nullptr : c->d_tile_first.p, c->d_prm.p);
  mark(c, 1);
}

// Slot generation (one workgroup per sort tile: it leaves the tile's histogram of the first
// sort pass behind as well, kernels.h) and / or the request classification (class masks,
// own-servant ranges, consuming counts per chunk), which rides in the same launch as
// workgroups [gen_blocks, gen_blocks + cls_blocks).
void enqueue_gen(ydc_context* c, const BatchPlan& p, const ydc_task_soa* tk, bool gen, bool classify) {
  const uint32_t N = p.N, S = p.S, C = p.C, W = p.W, cs = p.cs;
  ClassifyArgs ca{};
  if (N && classify) {
    ca = ClassifyArgs{TaskColumns{tk->env_id, tk->min_version, tk->requestor_ip}, N,
                      c->d_cls_env.p, c->d_cls_ver.p, C, W, c->env_words,
                      c->tables.env_ver_mask.empty() ? nullptr : c->d_ver_sorted.p,
                      c->tables.env_ver_mask.empty() ? nullptr : c->d_env_ver_mask.p,
                      (uint32_t)c->tables.ver_sorted.size(), c->d_ip_sorted.p, c->d_ip_servant.p, S,
                      c->d_slot_base.p, cs, c->d_mask.p, c->d_self_lo.p, c->d_self_hi.p,
                      c->d_chunk_consuming.p, c->d_cls_comp.p, c->n_parts,
                      p.mb.tail ? c->d_chunk_tail.p : nullptr, p.mb.warm_len,
                      p.binsort ? 1u : 0u, p.binsort ? 1u : 0u};
  }
  ca.n_ip = (uint32_t)c->tables.ip_sorted.size();
  ca.ip_hash = (const uint2*)c->d_ip_hash.p;
  ca.ip_hash_shift = c->tables.ip_hash_shift;
  ca.xcd_gen = c->opt_xcd & 1;
  ca.ip_filter = c->d_ip_filter.p;
  ca.ip_filter_shift = c->tables.ip_filter_shift;
  ca.row_out = p.wide_lists && N && classify ? c->d_row_of.p : nullptr;
  ca.cls_comp = c->d_cls_comp.p;  // (k_slot_gen reads them for the part id above the key)
  ca.n_parts = c->n_parts;
  const uint32_t bpp0 = c->kf.bits_per_pass;
  const uint32_t fused0 = p.key_passes == 1 ? p.fused_cls_bits : 0;
  const uint32_t bits0 = std::min(bpp0, c->kf.key_bits) + fused0;
  const uint32_t gen_blocks = gen && p.slot_bound ? p.n_tiles : 0;
  // Large batches of the radix path: four requests per thread (kernels.h: task_classify_block_multi;
  // the lookup form with one mask word). YDC_CLASSIFY_PER_THREAD=1 keeps one.
  ca.per_thread = classify && !p.binsort && N >= (1u << 18) && !c->tables.env_ver_mask.empty() && W == 1 &&
                          !ca.row_out && c->opt_classify_multi ? 4u : 1u;
  const uint32_t cls_blocks = classify ? ceil_div(N, 256 * ca.per_thread) : 0;
  if (gen_blocks + cls_blocks == 0) return;
  const size_t lds0 = ((size_t)4 << bits0);
  // Key window (sharded sort): local prefix, first local slot and registry-wide names.
  const uint32_t* base = p.win ? c->group.d_lbase.p : c->d_slot_base.p;
  const uint32_t* r_first = p.win ? c->group.d_r_first.p : nullptr;
  const uint32_t* gbase = p.win ? c->d_slot_base.p : nullptr;
  uint16_t* cls_by_g = C > 1 && !p.gbits ? c->d_cls_by_g.p : nullptr;
  if (p.binsort && gen) {
    // Bin boundaries | slot tiles | requests: one launch, no workgroup waits for another.
    const BinTable bt{p.n_bins, p.bin_shift, c->d_binbase.p, c->d_binruns.p, p.bin_group, p.bin_tiles,
                      c->d_bin_tile_start.p, c->d_bin_tile_base.p};
    const size_t lds_words = std::max<size_t>((size_t)5 * p.n_bins + 7 * p.bin_group + 1, C + 1);
    YDC_LAUNCH(c, "k_front_bins", k_front_bins, dim3(p.n_bins + p.bin_tiles + cls_blocks), dim3(256),
               lds_words * 4, c->stream, p.sv, C, p.slot_bound_glob, c->d_slot_base.p, c->d_cls_begin.p,
               PartTable{c->d_cls_comp.p, c->n_parts, c->d_part_base.p}, c->d_prm.p, c->kf.cap_bits,
               c->kf.comp_shift, p.gbits, bt, c->d_owner.p, (uint2*)c->d_keys[0].p, ca);
    return;
  }
  const char* gen_name = gen_blocks && cls_blocks ? "k_slot_gen" : (gen_blocks ? "k_slot_gen(slots)" : "k_slot_gen(requests)");
  if (p.key32) {
    YDC_LAUNCH(c, gen_name, k_slot_gen<uint32_t>, dim3(xcd_grid(gen_blocks) + cls_blocks), dim3(256), lds0,
               c->stream, p.sv, base, c->d_prm.p, (uint32_t)c->kf.exact, c->kf.cap_bits,
               (uint32_t*)c->d_keys[0].p, c->d_vals[0].p, cls_by_g, c->d_owner.p,
               gen_blocks, p.sort_items, bits0, fused0, p.gbits, c->d_hist.p, ca, c->kf.comp_shift,
               r_first, gbase, p.packed ? 1u : 0u, p.win ? nullptr : c->d_tile_first.p);
  } else {
    YDC_LAUNCH(c, gen_name, k_slot_gen<uint64_t>, dim3(xcd_grid(gen_blocks) + cls_blocks), dim3(256), lds0,
               c->stream, p.sv, base, c->d_prm.p, (uint32_t)c->kf.exact, c->kf.cap_bits,
               (uint64_t*)c->d_keys[0].p, c->d_vals[0].p, cls_by_g, c->d_owner.p,
               gen_blocks, p.sort_items, bits0, fused0, p.gbits, c->d_hist.p, ca, c->kf.comp_shift,
               r_first, gbase, p.packed ? 1u : 0u, p.win ? nullptr : c->d_tile_first.p);
  }
}

// What the chunk prefix reads: counts per chunk, or (bin sort) per wave of 64 requests.
PrefixArgs prefix_args(ydc_context* c, const BatchPlan& p) {
  return PrefixArgs{c->d_chunk_consuming.p, p.K, c->d_before.p, c->n_parts,
                    p.binsort ? p.cs / 64 : 0u, p.binsort ? ceil_div(p.N, 64) : 0u};
}

// Sort by key + class lists. prefix_pending: the chunk prefix of the consuming counts still
// has to be computed — it goes with the first histogram launch (one more workgroup).
int enqueue_sort(ydc_context* c, const BatchPlan& p, bool prefix_pending) {
  const uint32_t N = p.N;
  void* keys[2] = {c->d_keys[0].p, c->d_keys[1].p};
  uint32_t* vals[2] = {c->d_vals[0].p, c->d_vals[1].p};
  int cur = 0;
  const PrefixArgs pa = prefix_args(c, p);
  const PrefixArgs* pending_prefix = N && prefix_pending ? &pa : nullptr;
  mark(c, 2);
  if (p.binsort) {
    // One workgroup per bin (+ one for the chunk prefix): order inside the bins, global ranks,
    // class lists.
    BinSortArgs ba{(const uint2*)c->d_keys[0].p,
                   BinTable{p.n_bins, p.bin_shift, c->d_binbase.p, c->d_binruns.p, p.bin_group, p.bin_tiles,
                            c->d_bin_tile_start.p, c->d_bin_tile_base.p},
                   p.C, p.gbits, p.bin_slot_bits, p.bin_cls_bits, c->d_slot_base.p, c->d_cls_begin.p,
                   (uint2*)c->d_keys[1].p, c->d_rank_to_g.p,
                   c->opt_level_tab && c->n_parts <= 1 ? c->d_level_tab.p : nullptr};
    YDC_LAUNCH(c, "k_bin_sort", k_bin_sort, dim3(p.n_bins + (pending_prefix ? 1 : 0)), dim3(kBinThreads),
               (size_t)kBinLdsWords * 4, c->stream, ba, c->d_prm.p, pending_prefix ? pa : PrefixArgs{});
    mark(c, 3);
    mark(c, 4);
    return YDC_OK;
  }
  // ---- sort by key
  const uint32_t g_mask = p.gbits ? (1u << p.gbits) - 1 : 0xFFFFFFFFu;  // strips the class again
  const uint32_t bpp = c->kf.bits_per_pass;
  auto bits_of = [&](uint32_t q) { return std::min(bpp, c->kf.key_bits - q * bpp); };
  if (p.slot_bound) {
    for (uint32_t q = 0; q < p.key_passes; ++q) {
      // The last pass may carry the class above its key bits (k_radix_scatter_classed).
      const uint32_t fused = q + 1 == p.key_passes ? p.fused_cls_bits : 0;
      const uint16_t* cls = fused ? c->d_cls_by_g.p : nullptr;
      if (p.key32) {
        SortIn<uint32_t> in{(const uint32_t*)keys[cur], vals[cur], cls, q * bpp, bits_of(q) + fused,
                            p.sort_items, fused, p.gbits, fused ? g_mask : 0xFFFFFFFFu,
                            p.packed ? 1u : 0u, 0u};
        launch_sort_pass(c, in, p.n_tiles, keys[cur ^ 1], true, vals[cur ^ 1], q ? pending_prefix : nullptr,
                         q == 0);
      } else {
        SortIn<uint64_t> in{(const uint64_t*)keys[cur], vals[cur], cls, q * bpp, bits_of(q) + fused,
                            p.sort_items, fused, p.gbits, fused ? g_mask : 0xFFFFFFFFu, 0u, 0u};
        launch_sort_pass(c, in, p.n_tiles, keys[cur ^ 1], false, vals[cur ^ 1], q ? pending_prefix : nullptr,
                         q == 0);
      }
      if (q) pending_prefix = nullptr;
      cur ^= 1;
    }
  }
  mark(c, 3);
  // ---- class lists: stable partition of ranks by class
  for (uint32_t q = 0; q < p.cls_passes; ++q) {
    // First pass: key == index (global rank). Later passes carry the rank along.
    SortIn<uint32_t> in{q == 0 && !p.packed ? nullptr : (const uint32_t*)keys[cur], vals[cur],
                        c->d_cls_by_g.p, q * p.cls_bits, p.cls_bits, p.sort_items, 0u, p.gbits,
                        q + 1 == p.cls_passes ? g_mask : 0xFFFFFFFFu, p.packed ? 1u : 0u,
                        q == 0 ? 1u : 0u};
    launch_sort_pass(c, in, p.n_tiles, keys[cur ^ 1], true, vals[cur ^ 1], pending_prefix);
    pending_prefix = nullptr;
    cur ^= 1;
  }
  mark(c, 4);
  // Nothing was sorted (no free slot anywhere): the prefix gets a launch of its own.
  if (pending_prefix)
    YDC_LAUNCH(c, "k_chunk_prefix", k_chunk_prefix, dim3(1), dim3(1024), 0, c->stream, pa, c->d_prm.p);
  return YDC_OK;
}

// Everything before the level guesses: slots, sort, class lists, request classification.
int enqueue_front_a(ydc_context* c, const BatchPlan& p, const ydc_task_soa* tk) {
  enqueue_scan(c, p, c->d_cls_begin.p);
  if (c->opt_split_gen) {  // (measurement: the two halves of the launch timed apart)
    enqueue_gen(c, p, tk, true, false);
    enqueue_gen(c, p, tk, false, true);
  } else {
    enqueue_gen(c, p, tk, true, true);
  }
  return enqueue_sort(c, p, true);
}

// Level guesses of the chunks' start states (base: consuming requests of earlier ranks,
// multi-GPU only) and the two special cases that bypass the matching passes.
int enqueue_front_b(ydc_context* c, const BatchPlan& p, const uint32_t* d_base) {
  const uint32_t N = p.N, S = p.S, C = p.C, K = p.K;
  DeviceParams* prm = c->d_prm.p;
  hipStream_t st = c->stream;
  if (N && C && !(p.wave_path && p.mb.before && !d_base))
    YDC_LAUNCH(c, "k_guess_init", k_guess_init, dim3(ceil_div(K * C, 256)), dim3(256), 0, st, p.L,
               c->d_before.p, K, d_base, PartTable{c->d_cls_comp.p, c->n_parts, c->d_part_base.p},
               c->d_guess[0].p, c->d_dirty.p);
  mark(c, 5);
  if (N && C == 0) {
    // No eligible servant at all: every request fails with EnvironmentNotFound
    // (task_dispatcher.cc:105-108).
    HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)c->d_slot_of.p, (int)kIdxEnvNotFound, N, st));
  }
  if (N && C && p.any_shared && p.slot_bound) {
    // Hosts that run several servants: the replays resolve `self` from the class state, which
    // takes the list position of every servant's last slot (dispatch_core.h: SharedIpTable).
    YDC_LAUNCH(c, "k_pos_last", k_pos_last, dim3(ceil_div(p.slot_bound, 256)), dim3(256), 0, st,
               p.L, c->d_owner.p, c->d_slot_base.p, prm, c->d_pos_last.p);
  }
  (void)S;
  return YDC_OK;
}

// Host columns -> pinned arena -> device mirror, on the copy stream; the dispatch stream waits
// for the copy (only the kernels behind this point read the columns).
int stage_host_requests(ydc_context* c) {
  auto& h = c->host_in;
  if (h.direct) {
    const uint32_t* cols[3] = {h.tk->env_id, h.tk->min_version, h.tk->requestor_ip};
    for (int k = 0; k < 3; ++k)
      HIP_TRY(c, hipMemcpyAsync(c->d_in.p + k * h.col, cols[k], (size_t)h.n * 4, hipMemcpyHostToDevice,
                                c->copy_stream));
  } else {
    std::memcpy(c->h_in, h.tk->env_id, (size_t)h.n * 4);
    std::memcpy(c->h_in + h.col, h.tk->min_version, (size_t)h.n * 4);
    std::memcpy(c->h_in + 2 * h.col, h.tk->requestor_ip, (size_t)h.n * 4);
    HIP_TRY(c, hipMemcpyAsync(c->d_in.p, c->h_in, h.bytes, hipMemcpyHostToDevice, c->copy_stream));
  }
  HIP_TRY(c, hipEventRecord(c->copy_ev, c->copy_stream));
  HIP_TRY(c, hipStreamWaitEvent(c->stream, c->copy_ev, 0));
  return YDC_OK;
}

int enqueue_front(ydc_context* c, const BatchPlan& p, const ydc_task_soa* tk) {
  if (c->host_in.active && p.N) {
    // Host-buffer entry point: everything that does not read the requests first.
    enqueue_scan(c, p, c->d_cls_begin.p);
    enqueue_gen(c, p, tk, true, false);
    if (int rc = enqueue_sort(c, p, false)) return rc;
    if (int rc = stage_host_requests(c)) return rc;
    enqueue_gen(c, p, tk, false, true);
    const PrefixArgs pa = prefix_args(c, p);
    YDC_LAUNCH(c, "k_chunk_prefix", k_chunk_prefix, dim3(1), dim3(1024), 0, c->stream, pa, c->d_prm.p);
    return enqueue_front_b(c, p, nullptr);
  }
  if (int rc = enqueue_front_a(c, p, tk)) return rc;
  return enqueue_front_b(c, p, nullptr);
}

// One matching pass (match_kernel.h). device_check: return at once when the previous pass
// found every chunk consistent.
void enqueue_pass(ydc_context* c, const BatchPlan& p, uint32_t pass, uint32_t device_check) {
  if (p.fuse01 && pass == 1) return;  // (the launch of pass 0 did it)
  const size_t lds = (size_t)p.ring_total * 8;
  device_check |= c->debug_sim ? 2u : 0u;
  device_check |= c->opt_pair ? 4u : 0u;
  device_check |= c->walk_flag;  // (8: the walk's scout, 16: the walk itself — match_kernel.h)
  device_check |= p.ring_total << 8;
  DeviceParams* prm = c->d_prm.p;
  // (rings of 32 entries are watched by the fast loop itself: match_kernel.h, CHECKED)
  const bool checked = p.W == 1 && p.rshift == 5;
  // (pass 0 of a plan with a zone walk: one workgroup more, the walk is the first — zone_guess.h)
  const uint32_t grid1 = p.K + (p.mb.zone_box && pass == 0 ? 1u : 0u);
#define YDC_LAUNCH_MATCH1(OCC, CHECKED)                                                             \
  YDC_LAUNCH(c, "k_match_pass", (k_match_pass<1, OCC, CHECKED>), dim3(grid1), dim3(64), lds, c->stream, \
             p.L, p.T, p.N, p.cs, p.K, p.mb, pass, device_check, p.rshift, p.init_fill, p.shared, prm)
  if (p.W == 1) {
    if (p.dense && checked) YDC_LAUNCH_MATCH1(4, true);
    else if (p.dense) YDC_LAUNCH_MATCH1(4, false);
    else if (checked) YDC_LAUNCH_MATCH1(1, true);
    else YDC_LAUNCH_MATCH1(1, false);
#undef YDC_LAUNCH_MATCH1
  } else if (p.W == 2) {
    YDC_LAUNCH(c, "k_match_pass", (k_match_pass<2>), dim3(p.K), dim3(64), lds, c->stream, p.L, p.T,
               p.N, p.cs, p.K, p.mb, pass, device_check, p.rshift, p.init_fill, p.shared, prm);
  } else {
    YDC_LAUNCH(c, "k_match_pass", (k_match_pass<4>), dim3(p.K), dim3(64), lds, c->stream, p.L, p.T,
               p.N, p.cs, p.K, p.mb, pass, device_check, p.rshift, p.init_fill, p.shared, prm);
  }
}

// slot -> servant index, utilisation, running_tasks (one launch). check_slot != kNone: only
// takes effect when the pass with that counter slot found every chunk consistent. start_state
// (multi-GPU): class states before this rank's first request; d_taken: its slot deltas.
int enqueue_finalize(ydc_context* c, const BatchPlan& p, uint32_t flags, uint32_t* d_out_idx,
                     double* d_out_util, uint32_t* d_out_running, uint32_t check_slot,
                     uint32_t* d_taken = nullptr, const ClassState* start_state = nullptr) {
  const uint32_t S = p.S, N = p.N;
  const uint32_t req_blocks = ceil_div(N, 256), srv_blocks = ceil_div(S, 256);
  if (req_blocks + srv_blocks == 0) return YDC_OK;
  RunningArgs ra{};
  ra.end_state = N && p.C && p.K ? c->d_endst.p + (size_t)(p.K - 1) * p.C : nullptr;
  ra.start_state = ra.end_state ? start_state : nullptr;
  ra.L = p.L;
  ra.gslot_base = c->d_slot_base.p;
  ra.cls_comp = c->d_cls_comp.p;
  ra.n_parts = c->n_parts;
  ra.comp_shift = c->kf.comp_shift;
  ra.exact = c->kf.exact ? 1u : 0u;
  ra.cap_bits = c->kf.cap_bits;
  ra.n_servants = S;
  ra.running_out = c->d_running_out.p;
  ra.out_a = d_out_running;
  // The resident column is NOT written by this launch: its servant threads read the running
  // value of other servants (the head of their class list), so COMMIT is a copy behind it.
  ra.out_b = nullptr;
  ra.taken_out = d_taken;
  ra.pipelined = c->enqueue_pipelined ? 1u : 0u;
  ra.host_outcome = srv_blocks && c->opt_outcome_store ? c->finalize_outcome : nullptr;
  ra.srv_blocks = srv_blocks;
  YDC_LAUNCH(c, "k_finalize", k_finalize, dim3(req_blocks + srv_blocks), dim3(256), 0, c->stream, p.sv,
             c->d_slot_base.p, c->d_owner.p, p.rank_to_g, c->d_slot_of.p, N, p.wave_path ? 1u : 0u,
             d_out_idx, d_out_util, check_slot, c->d_prm.p, p.gbits ? (1u << p.gbits) - 1 : 0xFFFFFFFFu,
             req_blocks, ra, p.rank_stride);
  // COMMIT (`++pick->running_tasks`, task_dispatcher.cc:123): running_out -> the resident column.
  // When the passes have not converged yet running_out == running and the step is repeated.
  // Round 5: no copy where the caller can simply make running_out THE column afterwards
  // (commit_by_swap; a 4 us blit kernel per committed batch otherwise — kept where pointers are
  // baked into a captured step).
  if ((flags & YDC_DISPATCH_COMMIT) && S && !c->commit_by_swap)
    HIP_TRY(c, hipMemcpyAsync(c->d_running.p, c->d_running_out.p, (size_t)S * 4, hipMemcpyDeviceToDevice,
                              c->stream));
  return YDC_OK;
}

// Internal return code: a bin of the bin sort did not fit its LDS buffer (bin_sort.h) — the
// batch was gated out on the device and is to be repeated with the radix sort.
constexpr int kRetryRadix = 1 << 20;

// Reads the batch counters back and decides: 1 converged (rounds set), 0 more passes needed,
// 2 the batch has to be repeated with the radix sort.
int read_outcome(ydc_context* c, const BatchPlan& p, uint32_t first, uint32_t launched,
                 uint32_t* rounds, bool stored_by_finalize = false) {
  if (!stored_by_finalize)
    HIP_TRY(c, hipMemcpyAsync(c->h_prm, c->d_prm.p, sizeof(DeviceParams), hipMemcpyDeviceToHost,
                              c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  if (c->h_prm->overflow)
    return fail(c, YDC_ERR_CAPACITY, "slot workspace overflow (bound %u)", p.slot_bound);
  if (p.binsort && c->h_prm->window_miss) return 2;
  if (c->h_prm->n_changed[(launched - 1) & 63] != 0) return 0;
  *rounds = launched;
  for (uint32_t r = first; r < launched; ++r)
    if (c->h_prm->n_changed[r & 63] == 0) {
      *rounds = r + 1;  // first pass that found nothing to do
      break;
    }
  return 1;
}

void fill_stats(ydc_context* c, const BatchPlan& p, uint32_t rounds) {
  ydc_stats& s = c->stats;
  std::memset(&s, 0, sizeof(s));
  ++c->pipeline_batches;
  s.n_tasks = p.N;
  s.n_servants = p.S;
  s.n_classes = p.C;
  s.n_slots = c->h_prm->n_slots;
  s.key_bits = c->kf.key_bits;
  s.radix_passes = p.binsort ? 0 : p.key_passes;  // 0: the bin sort placed the slots
  s.n_chunks = p.K;
  s.rounds = rounds;
  s.chunk_sims = c->h_prm->chunk_sims;
  // Every request is exactly one of: granted, Timeout (eligible classes exist but are full),
  // EnvironmentNotFound (no eligible class).
  s.granted = c->h_prm->granted;
  s.shard_sort_batches = (uint32_t)c->group.windowed_batches;
  s.shard_sort_misses = (uint32_t)c->group.window_misses;
  s.zone_rows = p.zone ? c->h_prm->zone_rows : 0u;
  s.env_not_found = p.N - std::min(p.N, c->h_prm->consuming);
  s.timeouts = p.N - s.env_not_found - std::min(p.N - s.env_not_found, s.granted);
}

// Per-kernel totals of the dispatch just finished as JSON: {"name": [launches, total_ms], ...}
// (profiling only: one event pair per YDC_LAUNCH).
void collect_kernel_profile(ydc_context* c) {
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

// Walk or no walk for a batch of this shape? Without one until three batches (behind a first one
// that is not counted: cold caches) have shown a chain (more than two rounds); then three with it; then whichever costs less on the device
// (DeviceParams::t_begin .. t_end), the other one tried again every 64th batch.
bool zone_decide(ydc_context* c, uint32_t N, uint32_t K, uint32_t C) {
  const uint64_t shape = ((uint64_t)(N >> 12) << 40) | ((uint64_t)K << 16) | C;
  if (shape != c->zone_shape) {
    c->zone_shape = shape;
    c->zone_on = c->zone_off = ydc_context::ZoneArm{};
    c->zone_off_rounds = 0;
    c->zone_since_probe = 0;
    c->zone_cold = true;
  }
  if (c->zone_off.n < 3 || c->zone_off_rounds <= 2) return false;  // (no chain: nothing to cure)
  if (c->zone_on.n < 3) return true;
  const bool best = c->zone_on.ticks < c->zone_off.ticks;
  if (++c->zone_since_probe >= 64) {
    c->zone_since_probe = 0;
    return !best;
  }
  return best;
}

// After a batch with a zone walk (zone_guess.h): the chunks it served should have come out
// consistent in their first replay — two rounds. If not, the walk was not yet on the true track
// where the transient began: the next one starts earlier. (rows: DeviceParams::zone_rows.)
constexpr uint32_t kZoneLeadMax = 4096;
void zone_feedback(ydc_context* c, const BatchPlan& p, const DeviceParams& o, uint32_t rounds) {
  const uint32_t rows = o.zone_rows;
  if (p.zone_eligible) {
    const uint64_t t0 = ((uint64_t)o.t_begin_hi << 32) | o.t_begin_lo, t1 = ((uint64_t)o.t_end_hi << 32) | o.t_end_lo;
    ydc_context::ZoneArm& arm = p.zone ? c->zone_on : c->zone_off;
    if (c->zone_cold) {
      c->zone_cold = false;
    } else if (t1 > t0 && t1 - t0 < 100000000ull) {  // (stamped by this batch's first and last kernel)
      const float x = (float)(t1 - t0);
      arm.ticks = arm.n ? 0.75f * arm.ticks + 0.25f * x : x;
      ++arm.n;
    }
    if (!p.zone) c->zone_off_rounds = rounds;
  }
  if (!p.zone || rows < 2) return;
  if (rounds <= 2) {
    c->zone_fails = 0;
    return;
  }
  if (p.mb.zone_lead < kZoneLeadMax) {
    c->zone_lead_cur = std::min(p.mb.zone_lead + 512, kZoneLeadMax);
  } else if (++c->zone_fails >= 4) {  // this registry's transient is not one the walk tracks
    c->zone_fails = 0;
    c->zone_cooldown = 256;
  }
}

// Passes to launch before the first look at the outcome: what the last batch needed — and what
// the last batch WITHOUT the walk of the tier's end needed when this is one of those again
// (zone_decide tries the other arm now and then: it must not cost a pipeline miss).
uint32_t first_group(const ydc_context* c, const BatchPlan& p) {
  uint32_t hint = c->round_hint;
  if (p.zone_eligible && !p.zone) hint = std::max(hint, c->zone_off_rounds);
  return std::max(2u, std::min(hint, 16u));
}

// Passes [launched, ...) in groups until one finds every chunk consistent, each group
// followed by the (gated) finalise and one look at the counters.
int run_passes_until_consistent(ydc_context* c, const BatchPlan& p, uint32_t launched, uint32_t flags,
                                uint32_t* d_out_idx, double* d_out_util, uint32_t* d_out_running,
                                uint32_t* rounds) {
  bool walked = false;
  for (;;) {
    const uint32_t group = launched == 0 ? first_group(c, p) : 4u;
    if (launched) {
      // Counter slots of the passes to come (the first group's were cleared by
      // k_servant_scan). The stream is idle here: the host has just synchronised.
      for (uint32_t r = launched; r < launched + group; ++r) {
        HIP_TRY(c, hipMemsetAsync(&c->d_prm.p->n_changed[r & 63], 0, 4, c->stream));
        HIP_TRY(c, hipMemsetAsync(&c->d_prm.p->n_sampled[r & 63], 0, 4, c->stream));
      }
    }
    const uint32_t first = launched;
    // (k_finalize stores the outcome to the page-locked block itself where it has servant workgroups)
    const bool outcome_stored = c->opt_outcome_store && p.S != 0;
    c->finalize_outcome = outcome_stored ? c->d_h_prm : nullptr;
    if (launched >= c->opt_walk_after && !walked && group >= 4) {
      // Parallel repair is not getting anywhere (one chunk per pass): scout + walk, then two
      // ordinary passes that find everything consistent (match_kernel.h: walk_scout / walk_run).
      walked = true;
      c->walked_at = launched;
      HIP_TRY(c, hipMemsetAsync(&c->d_prm.p->reserved0, 0xFF, 4, c->stream));
      c->walk_flag = 8u;
      enqueue_pass(c, p, launched, 1u);
      c->walk_flag = 16u;
      enqueue_pass(c, p, launched + 1, 1u);
      c->walk_flag = 0;
      for (uint32_t r = launched + 2; r < launched + group; ++r) enqueue_pass(c, p, r, 1u);
    } else {
      for (uint32_t r = launched; r < launched + group; ++r) enqueue_pass(c, p, r, 1u);
    }
    launched += group;
    const bool by_swap = c->opt_commit_swap && !c->stream_mode.active && (flags & YDC_DISPATCH_COMMIT) && p.S;
    c->commit_by_swap = by_swap;
    const int frc = enqueue_finalize(c, p, flags, d_out_idx, d_out_util, d_out_running, (launched - 1) & 63);
    c->commit_by_swap = false;
    c->finalize_outcome = nullptr;
    if (frc) return frc;
    mark(c, 7);
    if (c->post_copy.bytes)  // a finalise that was gated out is repeated, and so is the copy
      HIP_TRY(c, hipMemcpyAsync(c->post_copy.dst, c->post_copy.src, c->post_copy.bytes,
                                hipMemcpyDeviceToHost, c->stream));
    int done = read_outcome(c, p, first, launched, rounds, outcome_stored);
    if (done < 0) return done;
    if (done == 2) return kRetryRadix;
    if (done) {
      // (the finalise just waited for was the final one: its output IS the column now)
      if (by_swap) std::swap(c->d_running, c->d_running_out);
      if (c->debug_sim) {
        fprintf(stderr, "[ydc match] K=%u cs=%u R=%u fill=%u rounds=%u sims=%u pass changed ends:", p.K,
                p.cs, 1u << p.rshift, p.init_fill, *rounds, c->h_prm->chunk_sims);
        for (uint32_t r = 0; r < *rounds; ++r) fprintf(stderr, " %u", c->h_prm->n_changed[r & 63]);
        fprintf(stderr, "\n");
      }
      return YDC_OK;
    }
    if (launched > p.K + 4)
      return fail(c, YDC_ERR_NOT_CONVERGED, "no fixpoint after %u passes", launched);
  }
}

// Debugging aid (YDC_BINSORT_VERIFY=1): the bin sort's outputs — global order, class lists —
// against a host sort of the very records k_front_bins staged. Waits for the stream.
int verify_binsort(ydc_context* c, const BatchPlan& p) {
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  DeviceParams prm;
  HIP_TRY(c, hipMemcpy(&prm, c->d_prm.p, sizeof(prm), hipMemcpyDeviceToHost));
  const uint32_t M = prm.n_slots, C = p.C, row = C + 1;
  if (prm.window_miss) {
    fprintf(stderr, "[ydc binsort verify] a bin overflowed (window_miss): nothing to compare\n");
    return YDC_OK;
  }
  std::vector<uint2> stage(M), list(M);
  std::vector<uint32_t> r2g(M), cls_begin(C + 1), base((size_t)(p.n_bins + 1) * row);
  HIP_TRY(c, hipMemcpy(stage.data(), c->d_keys[0].p, (size_t)M * 8, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(list.data(), c->d_keys[1].p, (size_t)M * 8, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(r2g.data(), c->d_rank_to_g.p, (size_t)M * 4, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(cls_begin.data(), c->d_cls_begin.p, (size_t)(C + 1) * 4, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(base.data(), c->d_binbase.p, base.size() * 4, hipMemcpyDeviceToHost));
  const uint32_t gmask = p.gbits ? (1u << p.gbits) - 1 : 0xFFFFFFFFu;
  uint32_t bad = 0, max_bin = 0;
  auto complain = [&](const char* what, uint32_t at, uint32_t got, uint32_t want) {
    if (++bad <= 12) fprintf(stderr, "[ydc binsort verify] %s at %u: device %u, host %u\n", what, at, got, want);
  };
  if (base[(size_t)p.n_bins * row + C] != M) complain("total of the last boundary row", p.n_bins, base[(size_t)p.n_bins * row + C], M);
  {
    // The closed-form bin starts against a count over the staged records (tile-major, any order).
    std::vector<uint32_t> cnt(p.n_bins + 1, 0);
    for (uint32_t i = 0; i < M; ++i) cnt[std::min(stage[i].x >> p.bin_shift, p.n_bins)]++;
    uint32_t acc = 0;
    for (uint32_t j = 0; j < p.n_bins; ++j) {
      if (base[(size_t)j * row + C] != acc) complain("start of bin", j, base[(size_t)j * row + C], acc);
      max_bin = std::max(max_bin, cnt[j]);
      acc += cnt[j];
    }
  }
  std::vector<uint2> sorted(stage);
  std::sort(sorted.begin(), sorted.end(), [&](const uint2& a, const uint2& b) {
    return a.x != b.x ? a.x < b.x : (a.y & gmask) < (b.y & gmask);
  });
  std::vector<uint32_t> next(cls_begin.begin(), cls_begin.end() - (C ? 1 : 0));
  for (uint32_t r = 0; r < M; ++r) {
    const uint32_t slot = sorted[r].y & gmask, cls = p.gbits ? sorted[r].y >> p.gbits : 0u;
    if (r2g[r] != slot) complain("rank_to_g", r, r2g[r], slot);
    if (cls < C) {
      const uint32_t at = next[cls]++;
      if (at < M && (list[at].x != r || list[at].y != slot)) {
        complain("class list rank", at, list[at].x, r);
        complain("class list slot", at, list[at].y, slot);
      }
    } else {
      complain("class of a record", r, cls, C);
    }
  }
  fprintf(stderr, "[ydc binsort verify] M=%u bins=%u shift=%u classes=%u fullest bin=%u: %u mismatches\n", M,
          p.n_bins, p.bin_shift, C, max_bin, bad);
  return bad ? fail(c, YDC_ERR_HIP, "bin sort verification failed (%u mismatches)", bad) : YDC_OK;
}

// Front, matching passes and finalise of a planned batch; returns when the results are there.
// kRetryRadix: the plan used the bin sort and a bin overflowed — nothing was committed.
int run_planned_batch(ydc_context* c, const BatchPlan& p, const ydc_task_soa* tk, uint32_t flags,
                      uint32_t* d_out_idx, double* d_out_util, uint32_t* d_out_running,
                      uint32_t* rounds_out) {
  const uint32_t N = p.N;
  if (int rc = enqueue_front(c, p, tk)) return rc;
  if (p.binsort && c->debug_verify_binsort)
    if (int rc = verify_binsort(c, p)) return rc;
  hipStream_t st = c->stream;
  DeviceParams* prm = c->d_prm.p;
  uint32_t rounds = 0;
  mark(c, 6);
  if (p.wave_path) {
    c->walked_at = 0;
    if (int rc = run_passes_until_consistent(c, p, 0, flags, d_out_idx, d_out_util, d_out_running,
                                             &rounds))
      return rc;
    // (a batch that had to be walked says nothing about how many passes the next one wants —
    // but if it is another of its kind, it should get to the walk as early)
    c->round_hint = c->walked_at ? std::min(c->walked_at, 3u) : rounds;
    zone_feedback(c, p, *c->h_prm, rounds);
  } else {
    if (N && p.C && p.use_generic) {
      // > kMaxWaveClasses classes: replay kernel + k_update per round, host-checked.
      const bool wide = p.C <= kMaxWideClasses && c->opt_wide;
      if (wide)  // (above 64 KB of dynamic LDS the runtime wants to be told)
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_sim_wide, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)std::max(wide_lds_bytes(kMaxWideClasses), wide_lds_bytes(kMaxWalkPrefetchClasses, true))));
      // The walk in groups of 64 requests (k_walk_groups) where the registry has eligible-class
      // lists and they fit the LDS beside the class states (YDC_GROUP_WALK=0: the lone walker).
      const uint32_t n_rows = p.wide_lists ? (uint32_t)c->tables.elig_off.size() - 1 : 0;
      const uint32_t n_list = p.wide_lists ? (uint32_t)c->tables.elig_cls.size() : 0;
      const bool group_walk = wide && p.wide_lists && c->opt_group_walk && p.C <= 65535 &&
                              group_walk_lds_bytes(p.C, n_rows, n_list) <= kGroupWalkMaxLds;
      // Head rank and class id in one word where both fit (and the extra array fits the LDS):
      // ranks are list positions of the whole registry here, below slot_bound.
      uint32_t walk_cbits = 1;
      while ((1u << walk_cbits) < p.C) ++walk_cbits;
      const bool walk_packed = group_walk && c->opt_walk_packed && c->group.n_ranks <= 1 &&
                               (uint64_t)p.slot_bound + 1 < ((uint64_t)1 << (32 - walk_cbits)) &&
                               group_walk_lds_bytes(p.C, n_rows, n_list, true) <= kGroupWalkMaxLds;
      if (group_walk)
        HIP_TRY(c, hipFuncSetAttribute(walk_packed ? (const void*)k_walk_groups<true> : (const void*)k_walk_groups<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGroupWalkMaxLds));
      uint32_t last_changed = 0xFFFFFFFFu;
      bool walked = false;
      if (group_walk) {
        // Sparse eligibility (eligible-class lists): no level to guess from, so no rounds of
        // speculation at all — the batch is walked from its first request, 64 requests at a time
        // (chunk 0's start state is the true one, k_guess_init; every chunk is still marked).
        if (walk_packed)
          YDC_LAUNCH(c, "k_walk_groups", k_walk_groups<true>, dim3(1), dim3(64),
                     group_walk_lds_bytes(p.C, n_rows, n_list, true), st, p.L, p.T, N, p.cs, p.K, c->d_guess[0].p,
                     c->d_endst.p, c->d_dirty.p, c->d_slot_of.p, p.shared, rounds, prm,
                     WideLists{c->d_row_of.p, c->d_elig_off.p, c->d_elig_cls.p}, n_rows, n_list, 1u, walk_cbits,
                     c->opt_walk_park ? 0u : 1u);
        else
          YDC_LAUNCH(c, "k_walk_groups", k_walk_groups<false>, dim3(1), dim3(64),
                     group_walk_lds_bytes(p.C, n_rows, n_list), st, p.L, p.T, N, p.cs, p.K, c->d_guess[0].p,
                     c->d_endst.p, c->d_dirty.p, c->d_slot_of.p, p.shared, rounds, prm,
                     WideLists{c->d_row_of.p, c->d_elig_off.p, c->d_elig_cls.p}, n_rows, n_list, 1u, walk_cbits,
                     c->opt_walk_park ? 0u : 1u);
        ++rounds;
        HIP_TRY(c, hipMemcpyAsync(c->h_prm, prm, sizeof(DeviceParams), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (c->h_prm->overflow)
          return fail(c, YDC_ERR_CAPACITY, "slot workspace overflow (bound %u)", p.slot_bound);
      }
      for (; !group_walk;) {
        for (uint32_t b = 0; b < c->opt_rounds_per_check; ++b) {
          ClassState* gold = c->d_guess[0].p;
          if (wide) {
            // One wave per chunk, the class states in LDS (wide_kernel.h) — or, once the rounds
            // have stopped making headway, one wave that walks the rest of the batch.
            // (the walk: with prefetch waves while their rings fit the LDS, YDC_WALK_PREFETCH=0: alone)
            const bool walk = walked, pf = walk && p.C <= kMaxWalkPrefetchClasses && c->opt_walk_prefetch;
            YDC_LAUNCH(c, walk ? "k_sim_wide(walk)" : "k_sim_wide", k_sim_wide, dim3(walk ? 1u : p.K),
                       dim3(pf ? 256 : 64), wide_lds_bytes(p.C, pf), st, p.L, p.T, N, p.cs, p.K, gold,
                       c->d_endst.p, c->d_dirty.p, c->d_slot_of.p, p.shared, rounds, prm,
                       walk ? (pf ? 2u : 1u) : 0u,
                       p.wide_lists ? WideLists{c->d_row_of.p, c->d_elig_off.p, c->d_elig_cls.p}
                                    : WideLists{nullptr, nullptr, nullptr});
          } else {
            YDC_LAUNCH(c, "k_sim_generic", k_sim_generic, dim3(ceil_div(p.K, 64)), dim3(64), 0, st, p.L,
                       p.T, N, p.cs, p.K, gold, c->d_endst.p, c->d_dirty.p, c->d_slot_of.p, c->d_runs.p,
                       p.shared, rounds, prm);
          }
          YDC_LAUNCH(c, "k_update", k_update, dim3(std::max(1u, ceil_div(p.K * p.C, 256))), dim3(256),
                     0, st, p.C, p.K, c->d_endst.p, gold, c->d_dirty.p, rounds, prm);
          ++rounds;
        }
        HIP_TRY(c, hipMemcpyAsync(c->h_prm, prm, sizeof(DeviceParams), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (c->h_prm->overflow)
          return fail(c, YDC_ERR_CAPACITY, "slot workspace overflow (bound %u)", p.slot_bound);
        const uint32_t changed = c->h_prm->n_changed[(rounds - 1) & 63];
        if (changed == 0) break;
        if (walked) return fail(c, YDC_ERR_NOT_CONVERGED, "the walk left %u inconsistent states", changed);
        // Guesses that are still changing almost as much as a check ago: corrections are
        // travelling chunk by chunk. Stop speculating.
        if (wide && rounds >= 4 && changed > last_changed / 4 * 3) walked = true;
        last_changed = changed;
        if (rounds > p.K + 4)
          return fail(c, YDC_ERR_NOT_CONVERGED, "no fixpoint after %u rounds", rounds);
      }
    }
    const bool by_swap = c->opt_commit_swap && !c->stream_mode.active && (flags & YDC_DISPATCH_COMMIT) && p.S;
    c->commit_by_swap = by_swap;
    const int frc = enqueue_finalize(c, p, flags, d_out_idx, d_out_util, d_out_running, kNone);
    c->commit_by_swap = false;
    if (frc) return frc;
    mark(c, 7);
    if (c->post_copy.bytes)
      HIP_TRY(c, hipMemcpyAsync(c->post_copy.dst, c->post_copy.src, c->post_copy.bytes,
                                hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(c->h_prm, prm, sizeof(DeviceParams), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipGetLastError());
    if (c->h_prm->overflow)
      return fail(c, YDC_ERR_CAPACITY, "slot workspace overflow (bound %u)", p.slot_bound);
    if (p.binsort && c->h_prm->window_miss) return kRetryRadix;
    if (by_swap) std::swap(c->d_running, c->d_running_out);  // (an ungated finalise: always final here)
  }
  *rounds_out = rounds;
  return YDC_OK;
}

// A bin of the bin sort overflowed: from now on (until the registry changes structure) the
// radix sort; `p` is planned again accordingly.
int fall_back_to_radix(ydc_context* c, uint32_t N, BatchPlan* p) {
  c->binsort_blocked = true;
  ++c->binsort_misses;
  c->stream_mode.stale = true;
  return plan_batch(c, N, p);
}

}  // namespace

// ---------------------------------------------------------------------------
// The small-batch path (tick_kernel.h): a handful of requests, the heartbeat rows that change no
// structure and the released grants that came in since the last call — one launch, no copy
// command; the host spins on the stamp the kernel stores last.
// ---------------------------------------------------------------------------
namespace {

// Registries and batches the one-workgroup kernel takes (tables must be current).
// The one-word candidate where the integer key and a registry index share 32 bits (capacities
// below 2^10 and a registry that leaves room: every realistic pool); the reference's double as
// the key otherwise (packed_tick=0: always).
bool tick_packed(const ydc_context* c, uint32_t* idx_bits_out = nullptr) {
  uint32_t idx_bits = 1;
  while ((1u << idx_bits) < std::max(c->n_servants, 2u)) ++idx_bits;
  if (idx_bits_out) *idx_bits_out = idx_bits;
  return c->opt_tick_packed && c->tables.cap_bits <= 10 && 2 * c->tables.cap_bits + 1 + idx_bits <= 32;
}

// Host request columns that are copies of their first entry (what one RPC sends).
bool tick_same_requests(const ydc_task_soa* tk, uint32_t n) {
  if (!tk || n < 2 || n > kTickBlock) return false;
  const uint32_t *e = (const uint32_t*)tk->env_id, *m = (const uint32_t*)tk->min_version,
                 *r = (const uint32_t*)tk->requestor_ip;
  for (uint32_t i = 1; i < n; ++i)
    if (e[i] != e[0] || m[i] != m[0] || r[i] != r[0]) return false;
  return true;
}

bool tick_takes(const ydc_context* c, uint32_t n_tasks, bool same = false) {
  same = same && tick_packed(c);  // (the wide builds of the kernel merge with one-word candidates only)
  return n_tasks <= c->small_batch(same) && c->small_batch() && c->n_servants <= kTickMaxServants &&
         c->tables.n_classes() <= kTickMaxClasses && c->h_alias_ip.empty() && c->group.n_ranks == 0 &&
         !c->stream_mode.active && c->pend_count == 0 && !c->debug_sim && !c->debug_verify_binsort;
}

// A heartbeat row that changes what the derived tables are built from (classes, the ip table,
// the slot bound): same test as ydc_update_servants_wide.
bool row_is_structural(const ydc_context* c, uint32_t s, const ydc_servant_row& r, const uint64_t* env_masks,
                       uint32_t env_words, uint32_t i) {
  if (s >= c->n_servants) return true;  // a new servant
  const uint32_t EW = c->env_words;
  if (env_masks && env_words > EW) return true;
  if (!env_masks && EW > 1) return true;  // (refused by ydc_update_servants_wide: let it say so)
  const uint64_t* env = &c->h_env[(size_t)s * EW];
  for (uint32_t w = 0; w < EW; ++w) {
    const uint64_t m = env_masks ? (w < env_words ? env_masks[(size_t)i * env_words + w] : 0) : (w == 0 ? r.env_mask : 0);
    if (env[w] != m) return true;
  }
  return c->h_version[s] != r.version || c->h_ip[s] != r.ip_id || (c->h_max_tasks[s] == 0) != (r.max_tasks == 0) ||
         std::min(c->h_max_tasks[s], c->h_nproc[s]) != std::min(r.max_tasks, r.num_processors);
}

// ---- the resident kernel (tick_kernel.h: TickBox) ----
// Contexts whose resident kernel may still be polling its mailbox: told to leave when the process
// exits without destroying them (the kernel reads page-locked memory the runtime is about to unmap).
std::mutex g_resident_mu;
std::vector<ydc_context*> g_resident;
bool g_resident_atexit = false;

inline unsigned long long box_load(const unsigned long long* p) {
  return __atomic_load_n(p, __ATOMIC_ACQUIRE);
}

// Waits until granule `g` of the reply carries command number `seq`; false when the kernel has left
// instead (idle exit racing with the command) or does not answer for a very long time.
bool box_wait(ydc_context* c, int g, uint32_t seq, uint32_t* word) {
  const unsigned long long* p = &c->h_box->reply[g];
  for (uint32_t spins = 0;; ++spins) {
    const unsigned long long v = box_load(p);
    if ((uint32_t)(v >> 32) == seq) {
      *word = (uint32_t)v;
      return true;
    }
    if ((spins & 0x3FFF) == 0x3FFF) {
      if (__atomic_load_n(&c->h_box->alive, __ATOMIC_ACQUIRE) == 0) {
        // (it may have answered right before leaving)
        const unsigned long long w = box_load(p);
        if ((uint32_t)(w >> 32) == seq) {
          *word = (uint32_t)w;
          return true;
        }
        return false;
      }
      // A live kernel that is slow (a preempted GPU, a debugger) is waited for: giving it up and
      // sending the command again could apply its releases and commit its picks twice. Only a
      // stream that has ended — finished or faulted — ends the wait.
      if ((spins & 0xFFFFF) == 0xFFFFF && hipStreamQuery(c->res_stream) != hipErrorNotReady) {
        const unsigned long long w = box_load(p);
        if ((uint32_t)(w >> 32) == seq) {
          *word = (uint32_t)w;
          return true;
        }
        return false;
      }
    }
  }
}

void resident_forget(ydc_context* c) {
  c->res_live = false;
  std::lock_guard<std::mutex> lk(g_resident_mu);
  g_resident.erase(std::remove(g_resident.begin(), g_resident.end(), c), g_resident.end());
}

// Ends the context's resident kernel (if any): QUIT through the mailbox, then its stream. After
// this the registry columns in HBM are what every other path expects (the kernel writes
// running_tasks and heartbeat rows through after every command).
void resident_stop(ydc_context* c) {
  if (!c->res_live) return;
  if (__atomic_load_n(&c->h_box->alive, __ATOMIC_ACQUIRE) != 0) {
    if (++c->tick_seq == 0) c->tick_seq = 1;
    const uint32_t seq = c->tick_seq;
    for (int g = 15; g >= 0; --g)
      __atomic_store_n(&c->h_box->head[g], ((unsigned long long)seq << 32) | (g == 0 ? kTickCmdQuit : 0u),
                       __ATOMIC_RELEASE);
    uint32_t word;
    (void)box_wait(c, 0, seq, &word);
  }
  (void)hipStreamSynchronize(c->res_stream);
  resident_forget(c);
}

void resident_atexit() {
  std::vector<ydc_context*> live;
  {
    std::lock_guard<std::mutex> lk(g_resident_mu);
    live = g_resident;
  }
  for (auto* c : live) resident_stop(c);
}

int resident_prepare(ydc_context* c) {
  if (!c->h_box) {
    HIP_TRY(c, hipHostMalloc((void**)&c->h_box, sizeof(TickBox), hipHostMallocCoherent | hipHostMallocMapped));
    std::memset(c->h_box, 0, sizeof(TickBox));
    HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_box, c->h_box, 0));
  }
  if (!c->res_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->res_stream, hipStreamNonBlocking));
  if (!c->res_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->res_ev, hipEventDisableTiming));
  return YDC_OK;
}

struct TickCall {
  const ydc_task_soa* tasks = nullptr;  // host columns, or device addresses (tasks_on_device)
  bool tasks_on_device = false;
  uint32_t n_tasks = 0;
  const uint32_t* upd_idx = nullptr;  // host; rows known to change no structure
  const ydc_servant_row* upd_rows = nullptr;
  uint32_t n_upd = 0;
  const uint32_t* rel = nullptr;  // host
  uint32_t n_rel = 0;
  uint32_t flags = 0;
  uint32_t* out_idx = nullptr;  // host, or device addresses (out_on_device)
  double* out_util = nullptr;
  uint32_t* out_running = nullptr;
  bool out_on_device = false;
};


// Fills the context's stats after a tick.
void tick_stats(ydc_context* c, uint32_t N, uint32_t granted, uint32_t timeouts, uint32_t env_not_found) {
  ++c->tick_batches;
  ydc_stats& st = c->stats;
  std::memset(&st, 0, sizeof(st));
  st.n_tasks = N;
  st.n_servants = c->n_servants;
  st.n_classes = c->tables.n_classes();
  st.key_bits = c->kf.key_bits;
  st.n_chunks = 1;
  st.rounds = 1;
  st.chunk_sims = 1;
  st.small_batch = 1;
  st.granted = granted;
  st.timeouts = timeouts;
  st.env_not_found = env_not_found;
}

// The answer of a resident kernel to command `seq`: counters, placements (tick_kernel.h: TickBox).
// false: the kernel left without taking the command.
bool resident_receive(ydc_context* c, const TickCall& io, uint32_t seq) {
  const uint32_t N = io.n_tasks;
  uint32_t counters;
  if (!box_wait(c, 0, seq, &counters)) return false;
  if (N <= 7 && !io.out_util) {
    for (uint32_t i = 0; i < N; ++i)
      if (!box_wait(c, 1 + (int)i, seq, &io.out_idx[i])) return false;
  } else {
    // (the arrays were stored, and fenced, ahead of the granules)
    for (uint32_t i = 0; i < N; ++i) io.out_idx[i] = __atomic_load_n(&c->h_box->out_idx[i], __ATOMIC_RELAXED);
    if (io.out_util) std::memcpy(io.out_util, c->h_box->out_util, (size_t)N * 8);
  }
  // granted: the requests that are neither of the two (the counters are bytes; N <= 64 fits)
  const uint32_t timeouts = (counters >> 8) & 0xFF, envnf = (counters >> 16) & 0xFF;
  tick_stats(c, N, N - timeouts - envnf, timeouts, envnf);
  return true;
}

int tick_run(ydc_context* c, const TickCall& io) {
  const uint32_t S = c->n_servants, C = c->tables.n_classes(), N = io.n_tasks;
  const uint32_t W = std::max<uint32_t>(1, ceil_div(C, 64));
  const bool commit0 = (io.flags & YDC_DISPATCH_COMMIT) != 0;
  // The resident form takes what a scheduler's turn looks like: COMMIT, host buffers, everything
  // within what travels as arguments.
  const bool resident = c->opt_resident && commit0 && !io.tasks_on_device && !io.out_on_device && !io.out_running &&
                        N <= kTickInlineTasks && io.n_upd <= kTickInlineUpd && io.n_rel <= kTickInlineRel &&
                        !c->profiling;
  if (c->res_live && !resident) resident_stop(c);
  // A resident kernel answers with or without utilisations for its whole life (its out_util is a
  // launch argument): a call that wants the other kind ends it and launches its own.
  if (c->res_live && (io.out_util != nullptr) != c->res_util) resident_stop(c);
  if (c->res_live) {
    TickBox* b = c->h_box;
    if (__atomic_load_n(&b->alive, __ATOMIC_ACQUIRE) != 0) {
      if (++c->tick_seq == 0) c->tick_seq = 1;
      const uint32_t seq = c->tick_seq;
      // Payload beyond the head first, the head's eight granules last. (One RPC's requests are
      // copies of one another, scheduler_service_impl.cc:228-264: then the head says so and holds all.)
      bool same = N > 1;
      for (uint32_t i = 1; i < N && same; ++i)
        same = io.tasks->env_id[i] == io.tasks->env_id[0] && io.tasks->min_version[i] == io.tasks->min_version[0] &&
               io.tasks->requestor_ip[i] == io.tasks->requestor_ip[0];
      if (N > 1 && !same)
        for (uint32_t i = 0; i < N; ++i) {
          b->env[i] = io.tasks->env_id[i];
          b->minv[i] = io.tasks->min_version[i];
          b->rip[i] = io.tasks->requestor_ip[i];
        }
      if (io.n_rel > 7) std::memcpy(b->rel, io.rel, (size_t)io.n_rel * 4);
      if (io.n_upd > 1)
        for (uint32_t i = 0; i < io.n_upd; ++i) {
          b->upd_idx[i] = io.upd_idx[i];
          b->upd[i] = TickRow{io.upd_rows[i].num_processors, io.upd_rows[i].current_load, io.upd_rows[i].max_tasks,
                              io.upd_rows[i].flags};
        }
      uint32_t words[16] = {kTickCmdTick | (same ? kTickCmdSame : 0u) | (N << 8) | (io.n_upd << 16) | (io.n_rel << 24),
                            N ? io.tasks->env_id[0] : 0u, N ? io.tasks->min_version[0] : 0u,
                            N ? io.tasks->requestor_ip[0] : 0u};
      for (uint32_t j = 0; j < 4 && j < io.n_rel; ++j) words[4 + j] = io.rel[j];
      for (uint32_t j = 4; j < 7 && j < io.n_rel; ++j) words[9 + j] = io.rel[j];
      if (io.n_upd) {
        words[8] = io.upd_idx[0];
        words[9] = io.upd_rows[0].num_processors;
        words[10] = io.upd_rows[0].current_load;
        words[11] = io.upd_rows[0].max_tasks;
        words[12] = io.upd_rows[0].flags;
      }
      for (int g = 15; g >= 0; --g)
        __atomic_store_n(&b->head[g], ((unsigned long long)seq << 32) | words[g], __ATOMIC_RELEASE);
      if (resident_receive(c, io, seq)) {
        ++c->tick_resident;
        return YDC_OK;
      }
    }
    // The kernel has left (nobody asked for a while): its stream is drained, a new one is launched
    // with this very command.
    HIP_TRY(c, hipStreamSynchronize(c->res_stream));
    resident_forget(c);
  }
  if (resident)
    if (int rc = resident_prepare(c)) return rc;
  if (!c->h_tick_done) {
    HIP_TRY(c, hipHostMalloc((void**)&c->h_tick_done, sizeof(TickDone), hipHostMallocCoherent | hipHostMallocMapped));
    std::memset(c->h_tick_done, 0, sizeof(TickDone));
    HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_tick_done, c->h_tick_done, 0));
  }
  auto pad = [](size_t b) { return (b + 63) & ~(size_t)63; };
  // Arena: [request columns] [heartbeat indexes | rows] [released] | [idx] [utilisation]
  const bool tasks_ptr = !io.tasks_on_device && N > kTickInlineTasks;
  const bool upd_ptr = io.n_upd > kTickInlineUpd, rel_ptr = io.n_rel > kTickInlineRel;
  const size_t o_env = 0, o_upd = o_env + (tasks_ptr ? 3 * pad((size_t)N * 4) : 0);
  const size_t o_rows = o_upd + (upd_ptr ? pad((size_t)io.n_upd * 4) : 0);
  const size_t o_rel = o_rows + (upd_ptr ? pad((size_t)io.n_upd * sizeof(TickRow)) : 0);
  const size_t o_idx = o_rel + (rel_ptr ? pad((size_t)io.n_rel * 4) : 0);
  const size_t o_util = o_idx + (io.out_on_device ? 0 : pad((size_t)N * 4));
  const size_t need = o_util + (!io.out_on_device && io.out_util ? pad((size_t)N * 8) : 0);
  if (need > c->tick_io_cap) {
    if (c->h_tick_io) (void)hipHostFree(c->h_tick_io);
    c->h_tick_io = c->d_tick_io = nullptr;
    c->tick_io_cap = 0;
    const size_t want = std::max<size_t>(need + need / 2, 8192);
    HIP_TRY(c, hipHostMalloc((void**)&c->h_tick_io, want, hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_tick_io, c->h_tick_io, 0));
    c->tick_io_cap = want;
  }
  const bool commit = (io.flags & YDC_DISPATCH_COMMIT) != 0;
  TickArgs a;
  a.nproc = c->d_nproc.p;
  a.load = c->d_load.p;
  a.max_tasks = c->d_max_tasks.p;
  a.flags = c->d_flags.p;
  a.class_of = c->d_class_of.p;
  a.ip = c->d_ip.p;
  a.running = c->d_running.p;
  a.cls_env = c->d_cls_env.p;
  a.cls_ver = c->d_cls_ver.p;
  a.S = S;
  a.C = C;
  a.EW = c->env_words;
  a.W = W;
  // running_tasks: the picks work on the resident column (COMMIT) or on a copy of it.
  uint32_t* dev_run_out = io.out_running ? (io.out_on_device ? io.out_running : c->d_running_out.p) : nullptr;
  a.rw = commit ? c->d_running.p : (dev_run_out ? dev_run_out : c->d_running_out.p);
  a.run_out = dev_run_out && dev_run_out != a.rw ? dev_run_out : nullptr;
  a.n_tasks = N;
  a.t_env = a.t_minv = a.t_rip = nullptr;
  if (io.tasks_on_device) {
    a.t_env = io.tasks->env_id;
    a.t_minv = io.tasks->min_version;
    a.t_rip = io.tasks->requestor_ip;
  } else if (tasks_ptr) {
    const size_t col = pad((size_t)N * 4);
    std::memcpy(c->h_tick_io + o_env, io.tasks->env_id, (size_t)N * 4);
    std::memcpy(c->h_tick_io + o_env + col, io.tasks->min_version, (size_t)N * 4);
    std::memcpy(c->h_tick_io + o_env + 2 * col, io.tasks->requestor_ip, (size_t)N * 4);
    a.t_env = (const uint32_t*)(c->d_tick_io + o_env);
    a.t_minv = (const uint32_t*)(c->d_tick_io + o_env + col);
    a.t_rip = (const uint32_t*)(c->d_tick_io + o_env + 2 * col);
  } else {
    for (uint32_t i = 0; i < N; ++i) {
      a.in_env[i] = io.tasks->env_id[i];
      a.in_minv[i] = io.tasks->min_version[i];
      a.in_rip[i] = io.tasks->requestor_ip[i];
    }
  }
  a.n_upd = io.n_upd;
  a.upd_idx = nullptr;
  a.upd_rows = nullptr;
  {
    uint32_t* idx = upd_ptr ? (uint32_t*)(c->h_tick_io + o_upd) : a.in_upd_idx;
    TickRow* rows = upd_ptr ? (TickRow*)(c->h_tick_io + o_rows) : a.in_upd;
    for (uint32_t i = 0; i < io.n_upd; ++i) {
      idx[i] = io.upd_idx[i];
      rows[i] = TickRow{io.upd_rows[i].num_processors, io.upd_rows[i].current_load, io.upd_rows[i].max_tasks,
                        io.upd_rows[i].flags};
    }
    if (upd_ptr) {
      a.upd_idx = (const uint32_t*)(c->d_tick_io + o_upd);
      a.upd_rows = (const TickRow*)(c->d_tick_io + o_rows);
    }
  }
  a.n_rel = io.n_rel;
  a.rel = nullptr;
  if (rel_ptr) {
    std::memcpy(c->h_tick_io + o_rel, io.rel, (size_t)io.n_rel * 4);
    a.rel = (const uint32_t*)(c->d_tick_io + o_rel);
  } else if (io.n_rel) {
    std::memcpy(a.in_rel, io.rel, (size_t)io.n_rel * 4);
  }
  a.out_idx = io.out_on_device ? io.out_idx : (uint32_t*)(c->d_tick_io + o_idx);
  a.out_util = io.out_util ? (io.out_on_device ? io.out_util : (double*)(c->d_tick_io + o_util)) : nullptr;
  a.done = c->d_tick_done;
  a.box = nullptr;
  a.idle_ticks = 0;
  if (++c->tick_seq == 0) c->tick_seq = 1;
  a.seq = c->tick_seq;
  hipStream_t launch_stream = c->stream;
  if (resident) {
    // This launch stays: it answers through the mailbox like every later command, on a stream of
    // its own, behind whatever the context's stream still has in flight.
    a.box = c->d_box;
    a.idle_ticks = (unsigned long long)c->opt_resident_idle_ms * 100000ull;
    a.out_idx = c->d_box->out_idx;
    a.out_util = io.out_util ? c->d_box->out_util : nullptr;
    __atomic_store_n(&c->h_box->alive, 1u, __ATOMIC_RELEASE);
    HIP_TRY(c, hipEventRecord(c->res_ev, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->res_stream, c->res_ev, 0));
    launch_stream = c->res_stream;
  }

  if (c->profiling) {
    c->ksamples_used = 0;
    for (int s = 0; s <= 6; ++s) mark(c, s);
  }
  // As few waves as hold the registry in registers, at most 16 servants per thread
  // (tick_kernel.h): 256 threads up to 4096 servants, 512 beyond (32 per thread above 8192).
  const uint32_t per_thread = std::max(1u, ceil_div(S, 256u));
  // As few waves as hold the registry in registers, at most 16 servants per thread
  // (tick_kernel.h): 256 threads up to 4096 servants, 512 beyond (32 per thread above 8192).
  // LDS: the eligible-class mask, the candidate lists of the merge (256-thread kernels, 32 B per
  // thread), the servants' hosts and classes (6 B per servant slot).
  auto launch = [&](auto kernel, uint32_t threads, uint32_t k, bool pk) -> int {
    const size_t lds = (size_t)W * 8 + (tick_merges((int)threads, (int)k, pk) ? (size_t)32 * threads : 0) +
                       (size_t)6 * threads * k;
    if (lds > 48 * 1024)
      HIP_TRY(c, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    YDC_LAUNCH(c, "k_tick", kernel, dim3(1), dim3(threads), lds, launch_stream, a);
    return YDC_OK;
  };
  uint32_t idx_bits = 1;
  const bool packed = tick_packed(c, &idx_bits);
  a.cap_bits = c->tables.cap_bits;
  a.idx_bits = idx_bits;
  int lrc;
#define YDC_TICK_LAUNCH(T, KK, COLD) \
  lrc = packed ? launch(k_tick<T, KK, COLD, true>, T, KK, true) : launch(k_tick<T, KK, COLD, false>, T, KK, false)
  if (per_thread <= 1) YDC_TICK_LAUNCH(256, 1, true);
  else if (per_thread <= 2) YDC_TICK_LAUNCH(256, 2, true);
  else if (per_thread <= 4) YDC_TICK_LAUNCH(256, 4, true);
  else if (per_thread <= 8) YDC_TICK_LAUNCH(256, 8, true);
  else if (per_thread <= 16) YDC_TICK_LAUNCH(256, 16, true);
  else if (per_thread <= 32) YDC_TICK_LAUNCH(512, 16, true);
  else YDC_TICK_LAUNCH(512, 32, false);
#undef YDC_TICK_LAUNCH
  if (lrc) return lrc;
  HIP_TRY(c, hipGetLastError());
  mark(c, 7);
  ++c->tick_launches;
  if (resident) {
    c->res_live = true;
    c->res_util = io.out_util != nullptr;
    {
      std::lock_guard<std::mutex> lk(g_resident_mu);
      g_resident.push_back(c);
      if (!g_resident_atexit) {
        g_resident_atexit = true;
        std::atexit(resident_atexit);
      }
    }
    if (!resident_receive(c, io, a.seq)) {
      (void)hipStreamSynchronize(c->res_stream);
      resident_forget(c);
      return fail(c, YDC_ERR_HIP, "the resident small-batch kernel left without answering its first command");
    }
    return YDC_OK;
  }

  // The kernel's last store is the stamp; spin on it (a launch-to-stamp round trip is a third
  // shorter than launch + hipStreamSynchronize). Never forever: the stream is asked now and then.
  {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
      if (__atomic_load_n(&c->h_tick_done->seq, __ATOMIC_ACQUIRE) == a.seq) break;
      if ((spins & 0xFFFF) == 0xFFFF) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q != hipSuccess && q != hipErrorNotReady)
          return fail(c, YDC_ERR_HIP, "the small-batch kernel failed: %s", hipGetErrorString(q));
        if (q == hipSuccess && __atomic_load_n(&c->h_tick_done->seq, __ATOMIC_ACQUIRE) != a.seq)
          return fail(c, YDC_ERR_HIP, "the small-batch kernel finished without its stamp");
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
          return fail(c, YDC_ERR_HIP, "no stamp from the small-batch kernel after 30 s");
      }
    }
  }
  if (io.out_on_device || (io.out_running && !io.out_on_device) || c->profiling)
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // (device outputs: complete when the call returns)
  if (!io.out_on_device) {
    if (N) std::memcpy(io.out_idx, c->h_tick_io + o_idx, (size_t)N * 4);
    if (N && io.out_util) std::memcpy(io.out_util, c->h_tick_io + o_util, (size_t)N * 8);
    if (io.out_running && S)
      HIP_TRY(c, hipMemcpy(io.out_running, dev_run_out, (size_t)S * 4, hipMemcpyDeviceToHost));
  }
  tick_stats(c, N, c->h_tick_done->granted, c->h_tick_done->timeouts, c->h_tick_done->env_not_found);
  ydc_stats& st = c->stats;
  if (c->profiling) {
    for (int i = 0; i < 7; ++i) (void)hipEventElapsedTime(&st.stage_ms[i], c->ev[i], c->ev[i + 1]);
    (void)hipEventElapsedTime(&st.stage_ms[YDC_STAGE_TOTAL], c->ev[0], c->ev[7]);
    collect_kernel_profile(c);
  }
  return YDC_OK;
}

}  // namespace

extern "C" {

int ydc_dispatch_device(ydc_context* c, const ydc_task_soa* tk, uint32_t N, uint32_t flags,
                        uint32_t* d_out_idx, double* d_out_util, uint32_t* d_out_running) {
  if (!c || (N && !tk)) return YDC_ERR_INVALID_ARGUMENT;
  if (c->max_tasks && N > c->max_tasks)
    return fail(c, YDC_ERR_CAPACITY, "%u tasks > max_tasks %u", N, c->max_tasks);
  if (c->pend_count && !c->pend[c->pend_head].rerun && c->pend[c->pend_head].active)
    return fail(c, YDC_ERR_INVALID_ARGUMENT, "pipelined batches outstanding: ydc_dispatch_wait first");
  HIP_TRY(c, hipSetDevice(c->device));
  if (N && N <= c->small_batch() && d_out_idx && !c->host_in.active && !c->post_copy.bytes) {
    // A handful of requests: one launch of the one-workgroup kernel (tick_kernel.h).
    if (c->tables_dirty)
      if (int rc = rebuild_tables(c)) return rc;
    if (tick_takes(c, N)) {
      TickCall io;
      io.tasks = tk;
      io.tasks_on_device = true;
      io.n_tasks = N;
      io.flags = flags;
      io.out_idx = d_out_idx;
      io.out_util = d_out_util;
      io.out_running = d_out_running;
      io.out_on_device = true;
      return tick_run(c, io);
    }
  }
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  BatchPlan p;
  if (int rc = plan_batch(c, N, &p)) return rc;
  uint32_t rounds = 0;
  int rc = run_planned_batch(c, p, tk, flags, d_out_idx, d_out_util, d_out_running, &rounds);
  if (rc == kRetryRadix) {
    if (int rc2 = fall_back_to_radix(c, N, &p)) return rc2;
    rc = run_planned_batch(c, p, tk, flags, d_out_idx, d_out_util, d_out_running, &rounds);
  }
  if (rc) return rc;
  fill_stats(c, p, rounds);
  ydc_stats& s = c->stats;
  if (c->profiling) {
    for (int i = 0; i < 7; ++i) (void)hipEventElapsedTime(&s.stage_ms[i], c->ev[i], c->ev[i + 1]);
    (void)hipEventElapsedTime(&s.stage_ms[YDC_STAGE_TOTAL], c->ev[0], c->ev[7]);
    collect_kernel_profile(c);
  }
  return YDC_OK;
}

// Pipelined form of ydc_dispatch_device: enqueues the whole batch (front, the matching passes the
// last batches needed, the gated finalise, the outcome read-back) and returns; the host looks at
// the outcome in ydc_dispatch_wait — by then the next batch is already queued behind this one, so
// the device does not idle while the host turns around. Exact whatever happens: a batch that is
// not final within its pre-launched passes (or whose bins overflowed) takes no effect, latches
// DeviceParams::pipeline_broken so that the batch behind it takes none either, and both are
// replayed in order by ydc_dispatch_wait.
int ydc_dispatch_device_async(ydc_context* c, const ydc_task_soa* tk, uint32_t N, uint32_t flags,
                              uint32_t* d_out_idx, double* d_out_util, uint32_t* d_out_running) {
  if (!c || (N && !tk)) return YDC_ERR_INVALID_ARGUMENT;
  if (c->pend_count == 2) return fail(c, YDC_ERR_INVALID_ARGUMENT, "two batches outstanding: ydc_dispatch_wait first");
  if (c->max_tasks && N > c->max_tasks)
    return fail(c, YDC_ERR_CAPACITY, "%u tasks > max_tasks %u", N, c->max_tasks);
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  auto& pd = c->pend[(c->pend_head + c->pend_count) & 1];
  if (!pd.h_outcome) {
    HIP_TRY(c, hipHostMalloc((void**)&pd.h_outcome, sizeof(DeviceParams), hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(c, hipHostGetDevicePointer((void**)&pd.d_h_outcome, pd.h_outcome, 0));
  }
  if (!pd.ev) HIP_TRY(c, hipEventCreateWithFlags(&pd.ev, hipEventDisableTiming));
  pd.active = true;
  pd.rerun = false;
  pd.tk = tk ? *tk : ydc_task_soa{};
  pd.n = N;
  pd.flags = flags;
  pd.out_idx = d_out_idx;
  pd.out_util = d_out_util;
  pd.out_running = d_out_running;
  pd.launched = 0;
  // A batch behind one that already has to be replayed is not worth enqueueing.
  const bool behind_rerun = c->pend_count == 1 && c->pend[c->pend_head].rerun;
  // A batch that cannot be enqueued completely is not outstanding: nothing of it can take effect
  // without its finalise, which is enqueued last; the slot is free again.
  auto give_up = [&](int rc) {
    pd.active = false;
    return rc;
  };
  if (int rc = plan_batch(c, N, &pd.plan)) return give_up(rc);
  if (behind_rerun || !pd.plan.wave_path || c->profiling || c->debug_verify_binsort) {
    // (registries without the wave path have host-checked rounds: placed when waited for)
    pd.rerun = true;
  } else {
    if (int rc = enqueue_front(c, pd.plan, &pd.tk)) return give_up(rc);
    const uint32_t group = first_group(c, pd.plan);
    for (uint32_t r = 0; r < group; ++r) enqueue_pass(c, pd.plan, r, 1u);
    pd.launched = group;
    c->enqueue_pipelined = true;
    // (the outcome block: stored by k_finalize's last servant workgroup; a registry without
    // servants has none, then it is read back with a copy)
    const bool outcome_stored = c->opt_outcome_store && pd.plan.S != 0;
    c->finalize_outcome = outcome_stored ? pd.d_h_outcome : nullptr;
    const bool by_swap = c->opt_commit_swap && !c->stream_mode.active && (flags & YDC_DISPATCH_COMMIT) && pd.plan.S;
    c->commit_by_swap = by_swap;
    int rc = enqueue_finalize(c, pd.plan, flags, d_out_idx, d_out_util, d_out_running, (group - 1) & 63);
    c->commit_by_swap = false;
    c->finalize_outcome = nullptr;
    c->enqueue_pipelined = false;
    if (rc) return give_up(rc);
    // (a batch that turns out not to be final wrote the column's own values: the exchange is
    // harmless then, and the replay plans with the pointers as they are)
    if (by_swap) std::swap(c->d_running, c->d_running_out);
    // (from here on the batch may take effect: a failing copy / event leaves it to be waited for
    // the slow way — a stream synchronise instead of the event)
    if ((!outcome_stored &&
         hipMemcpyAsync(pd.h_outcome, c->d_prm.p, sizeof(DeviceParams), hipMemcpyDeviceToHost, c->stream) != hipSuccess) ||
        hipEventRecord(pd.ev, c->stream) != hipSuccess) {
      (void)hipStreamSynchronize(c->stream);
      return give_up(fail(c, YDC_ERR_HIP, "could not enqueue the outcome read-back of a pipelined batch"));
    }
  }
  ++c->pend_count;
  return YDC_OK;
}

// Waits for the OLDEST outstanding batch; its results are final when this returns.
int ydc_dispatch_wait(ydc_context* c) {
  if (!c) return YDC_ERR_INVALID_ARGUMENT;
  if (c->pend_count == 0) return fail(c, YDC_ERR_INVALID_ARGUMENT, "no batch outstanding");
  HIP_TRY(c, hipSetDevice(c->device));
  auto& pd = c->pend[c->pend_head];
  auto pop = [&] {
    pd.active = false;
    c->pend_head ^= 1;
    --c->pend_count;
  };
  bool miss = pd.rerun;
  uint32_t rounds = 0;
  if (!miss) {
    HIP_TRY(c, hipEventSynchronize(pd.ev));
    const DeviceParams& o = *pd.h_outcome;
    if (o.overflow) {
      // Took no effect and latched the pipeline (k_finalize's gate includes overflow for
      // pipelined batches), so the batch behind it has not taken any either: drain, clear the
      // latch, leave that batch to be replayed when it is waited for, report this one.
      const uint32_t bound = pd.plan.slot_bound;
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      HIP_TRY(c, hipMemsetAsync(&c->d_prm.p->pipeline_broken, 0, 4, c->stream));
      if (c->pend_count == 2) c->pend[c->pend_head ^ 1].rerun = true;
      pop();
      return fail(c, YDC_ERR_CAPACITY, "slot workspace overflow (bound %u)", bound);
    }
    miss = o.pipeline_broken || (pd.plan.binsort && o.window_miss) || o.n_changed[(pd.launched - 1) & 63] != 0;
    if (!miss) {
      rounds = pd.launched;
      for (uint32_t r = 0; r < pd.launched; ++r)
        if (o.n_changed[r & 63] == 0) {
          rounds = r + 1;
          break;
        }
      *c->h_prm = o;
      c->round_hint = rounds;
      zone_feedback(c, pd.plan, o, rounds);
      fill_stats(c, pd.plan, rounds);
      pop();
      return YDC_OK;
    }
  }
  // Not final in the pipeline (or never enqueued): nothing of this batch — nor of the one behind
  // it — has taken effect. Drain, clear the latch, place it the synchronous way; the batch
  // behind it is replayed when it is waited for.
  ++c->pipeline_misses;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemsetAsync(&c->d_prm.p->pipeline_broken, 0, 4, c->stream));
  if (c->pend_count == 2) c->pend[c->pend_head ^ 1].rerun = true;
  const ydc_task_soa tk = pd.tk;
  const uint32_t n = pd.n, flags = pd.flags;
  uint32_t* oi = pd.out_idx;
  double* ou = pd.out_util;
  uint32_t* orun = pd.out_running;
  pop();
  return ydc_dispatch_device(c, &tk, n, flags, oi, ou, orun);
}

int ydc_dispatch(ydc_context* c, const ydc_task_soa* tk, uint32_t N, uint32_t flags,
                 uint32_t* out_idx, double* out_util, uint32_t* out_running) {
  if (!c || (N && (!tk || !out_idx))) return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  const bool same = N <= kTickBlock && tick_same_requests(tk, N);
  if (N && N <= c->small_batch(same)) {
    // A handful of requests: one launch of the one-workgroup kernel, the requests as kernel
    // arguments, the results stored to page-locked memory (tick_kernel.h).
    if (c->max_tasks && N > c->max_tasks)
      return fail(c, YDC_ERR_CAPACITY, "%u tasks > max_tasks %u", N, c->max_tasks);
    if (c->pend_count) return fail(c, YDC_ERR_INVALID_ARGUMENT, "pipelined batches outstanding: ydc_dispatch_wait first");
    if (c->tables_dirty)
      if (int rc = rebuild_tables(c)) return rc;
    if (tick_takes(c, N, same)) {
      TickCall io;
      io.tasks = tk;
      io.n_tasks = N;
      io.flags = flags;
      io.out_idx = out_idx;
      io.out_util = out_util;
      io.out_running = out_running;
      return tick_run(c, io);
    }
  }
  const uint32_t S = c->n_servants;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  // Page-locked caller buffers (ydc_host_register / ydc_host_alloc) are used as they are: the
  // classification reads the request columns and k_finalize writes the results through their
  // device addresses — no staging memcpy, no copy command on either side.
  const bool zero_copy = c->opt_zero_copy;
  const uint32_t* m_in[3] = {nullptr, nullptr, nullptr};
  if (zero_copy && N) {
    m_in[0] = (const uint32_t*)pinned_device_pointer(tk->env_id, (size_t)N * 4);
    m_in[1] = m_in[0] ? (const uint32_t*)pinned_device_pointer(tk->min_version, (size_t)N * 4) : nullptr;
    m_in[2] = m_in[1] ? (const uint32_t*)pinned_device_pointer(tk->requestor_ip, (size_t)N * 4) : nullptr;
  }
  const bool in_pinned = m_in[0] && m_in[1] && m_in[2];
  uint32_t* m_idx = zero_copy && N ? (uint32_t*)pinned_device_pointer(out_idx, (size_t)N * 4) : nullptr;
  uint32_t* m_run = zero_copy && out_running && S ? (uint32_t*)pinned_device_pointer(out_running, (size_t)S * 4) : nullptr;
  double* m_util = zero_copy && out_util && N ? (double*)pinned_device_pointer(out_util, (size_t)N * 8) : nullptr;
  const bool out_pinned = (!N || m_idx) && (!(out_running && S) || m_run) && (!(out_util && N) || m_util);
  // in: env | min_version | requestor_ip        out: idx | running_tasks | utilisation
  const size_t col = pad((size_t)N * 4), in_bytes = 3 * col;
  const size_t o_run = pad((size_t)N * 4), o_util = o_run + pad((size_t)(out_running ? S : 0) * 4);
  const size_t res_bytes = o_util + (out_util ? (size_t)N * 8 : 0);
  auto pinned = [&](uint8_t** q, size_t* cap, size_t want) -> hipError_t {
    if (want <= *cap) return hipSuccess;
    if (*q) (void)hipHostFree(*q);
    *q = nullptr;
    *cap = 0;
    want = std::max<size_t>(want + want / 2, 4096);
    hipError_t e = hipHostMalloc((void**)q, want);
    if (e == hipSuccess) *cap = want;
    return e;
  };
  // Request columns: read in place (in_pinned, YDC_HOST_IN=map), copied by DMA straight from
  // the caller's pinned columns (in_pinned, YDC_HOST_IN=copy), or staged through the context's
  // own pinned arena (pageable caller memory).
  const bool in_map = in_pinned && c->opt_host_in_map;
  ydc_task_soa d{};
  c->host_in.active = false;
  if (in_map) {
    d = ydc_task_soa{m_in[0], m_in[1], m_in[2]};
  } else if (N) {
    if (!in_pinned) HIP_TRY(c, pinned(&c->h_in, &c->h_in_cap, in_bytes));
    HIP_TRY(c, c->d_in.reserve(std::max<size_t>(in_bytes, 256)));
    if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->copy_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->copy_ev, hipEventDisableTiming));
    // The columns are staged and copied inside the batch, behind the launches that do not need
    // them (enqueue_front / stage_host_requests).
    c->host_in.active = true;
    c->host_in.direct = in_pinned;
    c->host_in.tk = tk;
    c->host_in.n = N;
    c->host_in.col = col;
    c->host_in.bytes = in_bytes;
    d = ydc_task_soa{(const uint32_t*)c->d_in.p, (const uint32_t*)(c->d_in.p + col),
                     (const uint32_t*)(c->d_in.p + 2 * col)};
  }
  int rc;
  if (out_pinned) {
    c->post_copy.bytes = 0;
    rc = ydc_dispatch_device(c, &d, N, flags, m_idx, out_util ? m_util : nullptr,
                             out_running && S ? m_run : nullptr);
    c->host_in.active = false;
    return rc;  // ydc_dispatch_device has waited for the stream: the results are in the caller's buffers
  }
  HIP_TRY(c, pinned(&c->h_res, &c->h_res_cap, res_bytes));
  HIP_TRY(c, c->d_res.reserve(std::max<size_t>(res_bytes, 256)));
  c->post_copy.dst = c->h_res;
  c->post_copy.src = c->d_res.p;
  c->post_copy.bytes = res_bytes;
  rc = ydc_dispatch_device(c, &d, N, flags, (uint32_t*)c->d_res.p,
                           out_util ? (double*)(c->d_res.p + o_util) : nullptr,
                           out_running && S ? (uint32_t*)(c->d_res.p + o_run) : nullptr);
  c->post_copy.bytes = 0;
  c->host_in.active = false;
  if (rc) return rc;
  // ydc_dispatch_device has waited for the stream: the results are in the pinned arena.
  if (N) std::memcpy(out_idx, c->h_res, (size_t)N * 4);
  if (out_running && S) std::memcpy(out_running, c->h_res + o_run, (size_t)S * 4);
  if (out_util && N) std::memcpy(out_util, c->h_res + o_util, (size_t)N * 8);
  return YDC_OK;
}

int ydc_dispatch_tick(ydc_context* c, const uint32_t* upd_idx, const ydc_servant_row* upd_rows,
                      const uint64_t* upd_env_masks, uint32_t env_words, uint32_t n_upd,
                      const uint32_t* release_servant_idx, uint32_t n_rel, const ydc_task_soa* tasks,
                      uint32_t n_tasks, uint32_t flags, uint32_t* out_servant_idx, double* out_utilization) {
  if (!c || (n_upd && (!upd_idx || !upd_rows)) || (n_rel && !release_servant_idx) ||
      (n_tasks && (!tasks || !out_servant_idx)))
    return YDC_ERR_INVALID_ARGUMENT;
  if (c->pend_count) return fail(c, YDC_ERR_INVALID_ARGUMENT, "pipelined batches outstanding: ydc_dispatch_wait first");
  if (c->max_tasks && n_tasks > c->max_tasks)
    return fail(c, YDC_ERR_CAPACITY, "%u tasks > max_tasks %u", n_tasks, c->max_tasks);
  HIP_TRY(c, hipSetDevice(c->device));
  // Heartbeats that change structure (a new servant, other environments / version / host /
  // capacity bound) take the general path, derived tables and all; so does a tick the kernel
  // does not take (a large batch, a registry beyond its limits).
  bool structural = false;
  for (uint32_t i = 0; i < n_upd && !structural; ++i)
    structural = row_is_structural(c, upd_idx[i], upd_rows[i], upd_env_masks, env_words, i);
  if (!structural && c->tables_dirty)
    if (int rc = rebuild_tables(c)) return rc;
  // (registry deltas ride in the launch only with COMMIT: running_tasks goes back once, into the
  // column the picks work on)
  const bool same = tick_same_requests(tasks, n_tasks);
  const bool fast = !structural && tick_takes(c, n_tasks, same) &&
                    ((flags & YDC_DISPATCH_COMMIT) || (!n_upd && !n_rel));
  if (!fast) {
    if (n_upd)
      if (int rc = ydc_update_servants_wide(c, upd_idx, upd_rows, upd_env_masks, env_words, n_upd)) return rc;
    n_upd = 0;
    if (c->tables_dirty)
      if (int rc = rebuild_tables(c)) return rc;
    if (!tick_takes(c, n_tasks, same) || (n_rel && !(flags & YDC_DISPATCH_COMMIT))) {
      if (n_rel)
        if (int rc = ydc_release_slots(c, release_servant_idx, n_rel)) return rc;
      if (!n_tasks) return YDC_OK;
      return ydc_dispatch(c, tasks, n_tasks, flags, out_servant_idx, out_utilization, nullptr);
    }
  }
  for (uint32_t i = 0; i < n_upd; ++i) {  // host mirror of the columns the rows replace
    const uint32_t s = upd_idx[i];
    c->h_nproc[s] = upd_rows[i].num_processors;
    c->h_load[s] = upd_rows[i].current_load;
    c->h_max_tasks[s] = upd_rows[i].max_tasks;
    c->h_flags[s] = upd_rows[i].flags;
  }
  TickCall io;
  io.tasks = tasks;
  io.n_tasks = n_tasks;
  io.upd_idx = upd_idx;
  io.upd_rows = upd_rows;
  io.n_upd = n_upd;
  io.rel = release_servant_idx;
  io.n_rel = n_rel;
  io.flags = flags;
  io.out_idx = out_servant_idx;
  io.out_util = out_utilization;
  return tick_run(c, io);
}

int ydc_host_register(void* p, size_t bytes) {
  if (!p || !bytes) return YDC_ERR_INVALID_ARGUMENT;
  hipError_t e = hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_create_error(std::string("hipHostRegister: ") + hipGetErrorString(e));
    return YDC_ERR_HIP;
  }
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess || !dev) {
    (void)hipGetLastError();
    (void)hipHostUnregister(p);
    set_create_error("hipHostGetDevicePointer failed for a registered range");
    return YDC_ERR_HIP;
  }
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  g_pinned.push_back(PinnedRange{(const char*)p, bytes, (char*)dev, 0});
  g_not_pinned.clear();
  return YDC_OK;
}

int ydc_host_unregister(void* p) {
  if (!p) return YDC_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  for (size_t i = 0; i < g_pinned.size(); ++i)
    if (g_pinned[i].host == (const char*)p && g_pinned[i].kind == 0) {
      g_pinned.erase(g_pinned.begin() + (long)i);
      return hipHostUnregister(p) == hipSuccess ? YDC_OK : YDC_ERR_HIP;
    }
  return YDC_ERR_INVALID_ARGUMENT;
}

int ydc_host_alloc(size_t bytes, void** out) {
  if (!out) return YDC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_create_error(std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? YDC_ERR_NO_DEVICE : YDC_ERR_HIP;
  }
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess || !dev) {
    (void)hipGetLastError();
    (void)hipHostFree(p);
    return YDC_ERR_HIP;
  }
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    g_pinned.push_back(PinnedRange{(const char*)p, bytes ? bytes : 1, (char*)dev, 1});
    g_not_pinned.clear();
  }
  *out = p;
  return YDC_OK;
}

int ydc_host_free(void* p) {
  if (!p) return YDC_OK;
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  for (size_t i = 0; i < g_pinned.size(); ++i)
    if (g_pinned[i].host == (const char*)p && g_pinned[i].kind == 1) {
      g_pinned.erase(g_pinned.begin() + (long)i);
      return hipHostFree(p) == hipSuccess ? YDC_OK : YDC_ERR_HIP;
    }
  return YDC_ERR_INVALID_ARGUMENT;
}

}  // extern "C" (reopened below)

// ---------------------------------------------------------------------------
// Multi-GPU: rank-range sharding of one batch. The global batch is the
// concatenation, in rank order, of the slices the ranks pass to ydc_dispatch_sharded;
// every rank holds the same servant table and computes the same sorted slot lists, and
// replays only its own slice. Exchanges (all-gathers, a few hundred bytes to S*4 bytes per
// rank): the slices' consuming-request counts (level guesses), after every matching pass
// the end state of each rank's last chunk + its count of inconsistent chunks, and finally
// the per-servant slot deltas. Transport: RCCL (librccl.so.1, resolved with dlopen so
// that a single-GPU scheduler has no RCCL dependency), or — for several contexts of one
// process on one device, which is how the protocol is tested on a single-GPU box — a
// barrier + device copies.
// ---------------------------------------------------------------------------
struct ydc_context::LocalHub {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  std::vector<const void*> send;
  int refs = 0;
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t gen = generation;
    if (++arrived == n) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};

namespace {

int group_all_gather(ydc_context* c, const void* send, void* recv, size_t bytes) {
  auto& g = c->group;
  if (g.n_ranks == 1 && !g.hub) {
    // A group of one: the gather is a copy (the communicator / mailbox exists, nothing to wait for).
    if (bytes) HIP_TRY(c, hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, c->stream));
    return YDC_OK;
  }
  if (g.box.kind) {
    if (bytes % 4) return fail(c, YDC_ERR_INVALID_ARGUMENT, "mailbox exchange of %zu bytes", bytes);
    const uint32_t words = (uint32_t)(bytes / 4), G = (uint32_t)g.n_ranks;
    for (uint32_t off = 0; off < words; off += g.box.slot_words) {
      const uint32_t n = std::min(g.box.slot_words, words - off);
      if (++g.box.seq == 0) ++g.box.seq;
      const uint32_t bpp = std::min(16u, std::max(1u, ceil_div(n, 1024)));
      YDC_LAUNCH(c, "k_mailbox_all_gather", k_mailbox_all_gather, dim3(G * bpp), dim3(256), 0, c->stream,
                 g.box.peers, (uint32_t)g.rank, G, (const uint32_t*)send + off, (uint32_t*)recv + off, n,
                 words, g.box.slot_words, g.box.seq & 1u, g.box.seq, bpp, g.box.timeout_ticks, c->d_prm.p);
    }
    return YDC_OK;
  }
  if (g.comm) {
    ncclResult_t r = g.all_gather_fn(send, recv, bytes, ncclUint8, g.comm, c->stream);
    if (r != ncclSuccess)
      return fail(c, YDC_ERR_HIP, "ncclAllGather: %s", g.error_string_fn ? g.error_string_fn(r) : "?");
    return YDC_OK;
  }
  if (g.hub) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // the send buffer is complete
    g.hub->send[g.rank] = send;
    g.hub->barrier();
    for (int r = 0; r < g.n_ranks; ++r)
      HIP_TRY(c, hipMemcpyAsync((char*)recv + (size_t)r * bytes, g.hub->send[r], bytes,
                                hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    g.hub->barrier();  // nobody rewrites its send buffer before everybody has read it
    return YDC_OK;
  }
  return fail(c, YDC_ERR_INVALID_ARGUMENT, "context is not part of a group");
}

// The mailbox transport reports a late peer only through DeviceParams::exchange_timeout (the
// words it waited for read as 0): whoever turns gathered words into sizes or inputs on the host
// asks here first. Waits for the stream.
int group_exchange_check(ydc_context* c) {
  auto& g = c->group;
  if (!g.box.kind) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return YDC_OK;
  }
  uint32_t late = 0;
  HIP_TRY(c, hipMemcpyAsync(&late, &c->d_prm.p->exchange_timeout, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (late)
    return fail(c, YDC_ERR_HIP, "a peer's data did not arrive within the mailbox time-out (rank %d of %d)",
                g.rank, g.n_ranks);
  return YDC_OK;
}

// What a rank hands its peers (ydc_group_ipc_export; YDC_IPC_HANDLE_BYTES).
struct MailboxBlob {
  uint32_t magic, abi;
  int32_t rank, n_ranks;
  uint32_t slot_words, have_device, have_host, device_fine;
  int32_t device, pid;
  uint64_t bytes, raw_dev, raw_host;  // raw_*: the owner's own pointers (peers inside the owner's process)
  hipIpcMemHandle_t handle;
  char shm_name[64];
};
static_assert(sizeof(MailboxBlob) <= YDC_IPC_HANDLE_BYTES, "blob must fit the published size");
constexpr uint32_t kMailboxMagic = 0x79646362u;  // "ydcb"

void mailbox_release(ydc_context* c) {
  auto& b = c->group.box;
  for (uint32_t q = 0; q < kMailboxMaxRanks; ++q) {
    if (b.opened_dev[q]) (void)hipIpcCloseMemHandle(b.opened_dev[q]);
    b.opened_dev[q] = nullptr;
    if (b.opened_host[q]) {
      (void)hipHostUnregister(b.opened_host[q]);
      (void)munmap(b.opened_host[q], b.bytes);
    }
    b.opened_host[q] = nullptr;
    b.peers.box[q] = nullptr;
  }
  if (b.own_dev) (void)hipFree(b.own_dev);
  b.own_dev = nullptr;
  b.have_handle = false;
  if (b.own_host) {
    (void)hipHostUnregister(b.own_host);
    (void)munmap(b.own_host, b.bytes);
    (void)shm_unlink(b.shm_name);
  }
  b.own_host = nullptr;
  b.shm_name[0] = 0;
  b.kind = 0;
  b.exported_ranks = 0;
  b.exported_rank = -1;
  b.bytes = 0;
  b.seq = 0;
}

void group_release(ydc_context* c) {
  auto& g = c->group;
  mailbox_release(c);
  if (g.comm && g.comm_destroy_fn) (void)g.comm_destroy_fn(g.comm);
  g.comm = nullptr;
  if (g.rccl) (void)dlclose(g.rccl);
  g.rccl = nullptr;
  if (g.hub) {
    bool last;
    {
      std::lock_guard<std::mutex> lk(g.hub->mu);
      last = --g.hub->refs == 0;
    }
    if (last) delete g.hub;
    g.hub = nullptr;
  }
  for (auto* b : {&g.d_totals, &g.d_meta, &g.d_base, &g.d_delta, &g.d_deltas, &g.d_pad, &g.d_gather,
                  &g.d_all[0], &g.d_all[1], &g.d_all[2], &g.d_all_idx, &g.d_cum, &g.d_r_first,
                  &g.d_lbase, &g.d_cls_begin_glob, &g.d_shift, &g.d_winrec, &g.d_winall})
    b->release();
  g.d_bound_local.release();
  g.d_all_util.release();
  g.d_send.release();
  g.d_bounds.release();
  if (g.h_bounds) (void)hipHostFree(g.h_bounds);
  g.h_bounds = nullptr;
  g.h_bounds_cap = 0;
  g.n_ranks = 0;
}

// librccl must sit on the SAME HIP runtime as this library: a process may hold two
// (e.g. /opt/rocm's and the one bundled with a PyTorch wheel), and RCCL calls on device
// memory of the other runtime fail. So look next to the libamdhip64 this library is bound
// to first, and only then fall back to the soname.
void* open_rccl(std::string* err) {
  std::vector<std::string> names;
  Dl_info info;
  if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      dir.resize(slash);
      names.push_back(dir + "/librccl.so.1");
      names.push_back(dir + "/librccl.so");
    }
  }
  names.push_back("librccl.so.1");
  names.push_back("librccl.so");
  std::string tried;
  for (auto& n : names) {
    if (void* h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL)) return h;
    tried += n + " ";
  }
  if (err) *err = "dlopen failed for: " + tried + "(" + (dlerror() ? dlerror() : "?") + ")";
  return nullptr;
}

}  // namespace

extern "C" {

int ydc_group_unique_id(void* out_id128) {
  if (!out_id128) return YDC_ERR_INVALID_ARGUMENT;
  static_assert(sizeof(ncclUniqueId) == 128, "ydc_group_unique_id hands out 128 bytes");
  std::string rccl_err;
  void* h = open_rccl(&rccl_err);
  if (!h) set_create_error(rccl_err);
  if (!h) return YDC_ERR_HIP;
  auto fn = (decltype(&ncclGetUniqueId))dlsym(h, "ncclGetUniqueId");
  if (!fn) {
    set_create_error("librccl has no ncclGetUniqueId");
    return YDC_ERR_HIP;
  }
  ncclUniqueId id;
  ncclResult_t r = fn(&id);
  if (r != ncclSuccess) {
    set_create_error("ncclGetUniqueId failed");
    return YDC_ERR_HIP;
  }
  std::memcpy(out_id128, &id, sizeof(id));
  return YDC_OK;  // the handle stays open: ydc_group_init reuses the loaded library
}

int ydc_group_init(ydc_context* c, const void* id128, int rank, int n_ranks) {
  if (!c || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  group_release(c);
  auto& g = c->group;
  std::string err;
  g.rccl = open_rccl(&err);
  if (!g.rccl) return fail(c, YDC_ERR_HIP, "%s", err.c_str());
  auto init_fn = (decltype(&ncclCommInitRank))dlsym(g.rccl, "ncclCommInitRank");
  g.all_gather_fn = (decltype(&ncclAllGather))dlsym(g.rccl, "ncclAllGather");
  g.comm_destroy_fn = (decltype(&ncclCommDestroy))dlsym(g.rccl, "ncclCommDestroy");
  g.error_string_fn = (decltype(&ncclGetErrorString))dlsym(g.rccl, "ncclGetErrorString");
  g.comm_count_fn = (decltype(&ncclCommCount))dlsym(g.rccl, "ncclCommCount");
  if (!init_fn || !g.all_gather_fn || !g.comm_destroy_fn)
    return fail(c, YDC_ERR_HIP, "librccl lacks ncclCommInitRank / ncclAllGather / ncclCommDestroy");
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclResult_t r = init_fn(&g.comm, n_ranks, id, rank);
  if (r != ncclSuccess) {
    g.comm = nullptr;
    return fail(c, YDC_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, n_ranks,
                g.error_string_fn ? g.error_string_fn(r) : "?");
  }
  g.rank = rank;
  g.n_ranks = n_ranks;
  return YDC_OK;
}

int ydc_group_init_local(ydc_context** ctxs, int n) {
  if (!ctxs || n < 1) return YDC_ERR_INVALID_ARGUMENT;
  auto* hub = new ydc_context::LocalHub();
  hub->n = n;
  hub->send.assign(n, nullptr);
  hub->refs = n;
  for (int r = 0; r < n; ++r) {
    if (!ctxs[r]) return YDC_ERR_INVALID_ARGUMENT;
    resident_stop(ctxs[r]);
    group_release(ctxs[r]);
    ctxs[r]->group.hub = hub;
    ctxs[r]->group.rank = r;
    ctxs[r]->group.n_ranks = n;
  }
  return YDC_OK;
}

int ydc_group_ipc_export(ydc_context* c, int rank, int n_ranks, void* out_handle) {
  if (!c || !out_handle || n_ranks < 1 || n_ranks > (int)kMailboxMaxRanks || rank < 0 || rank >= n_ranks)
    return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
  group_release(c);
  auto& b = c->group.box;
  // Slots of 64 KB of payload (a configs[3] registry's slot deltas: 16k servants x 4 B) unless
  // tuned; two sets of n_ranks slots of 8-byte granules.
  uint32_t slot_words = 16384;
  if (const char* e = tune_value("ipc_slot_words")) slot_words = (uint32_t)std::max(64, atoi(e));
  if (const char* e = tune_value("ipc_timeout_ms"))
    b.timeout_ticks = (unsigned long long)std::max(1, atoi(e)) * 100000ull;  // 100 MHz wall clock
  b.slot_words = slot_words;
  b.bytes = (size_t)2 * n_ranks * slot_words * 8;
  MailboxBlob blob{};
  blob.magic = kMailboxMagic;
  blob.abi = ydc_abi_version();
  blob.rank = rank;
  blob.n_ranks = n_ranks;
  blob.slot_words = slot_words;
  blob.device = c->device;
  blob.pid = (int32_t)getpid();
  blob.bytes = b.bytes;
  // Device flavour: fine-grained device memory where the runtime exports it (peer writes have to
  // be visible to a kernel that is running — across devices that takes fine-grained memory), plain
  // device memory otherwise (enough between processes that share one device).
  const char* coarse = tune_value("ipc_coarse");
  if (!(coarse && atoi(coarse))) {
    if (hipExtMallocWithFlags(&b.own_dev, b.bytes, hipDeviceMallocFinegrained) == hipSuccess) {
      if (hipIpcGetMemHandle(&b.handle, b.own_dev) == hipSuccess) {
        b.have_handle = true;
        b.own_dev_fine = true;
      } else {
        (void)hipFree(b.own_dev);
        b.own_dev = nullptr;
      }
    }
    (void)hipGetLastError();
  }
  if (!b.own_dev) {
    if (hipMalloc(&b.own_dev, b.bytes) == hipSuccess) {
      b.have_handle = hipIpcGetMemHandle(&b.handle, b.own_dev) == hipSuccess;
      b.own_dev_fine = false;
    } else {
      b.own_dev = nullptr;
    }
    (void)hipGetLastError();
  }
  if (b.own_dev) HIP_TRY(c, hipMemset(b.own_dev, 0, b.bytes));
  blob.have_device = b.own_dev != nullptr;  // (without a handle: peers inside this process only)
  blob.device_fine = b.own_dev_fine;
  blob.raw_dev = (uint64_t)(uintptr_t)b.own_dev;
  if (b.have_handle) blob.handle = b.handle;
  // Host flavour: a POSIX shared-memory segment, page-locked and mapped into the device's
  // address space by every process that opens it.
  snprintf(b.shm_name, sizeof(b.shm_name), "/ydc_box_%d_%llx", (int)getpid(),
           (unsigned long long)(uintptr_t)c & 0xFFFFFFFFFFull);
  (void)shm_unlink(b.shm_name);
  int fd = shm_open(b.shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd >= 0) {
    void* m = MAP_FAILED;
    if (ftruncate(fd, (off_t)b.bytes) == 0) m = mmap(nullptr, b.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    (void)close(fd);
    if (m != MAP_FAILED) {
      std::memset(m, 0, b.bytes);
      if (hipHostRegister(m, b.bytes, hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess) {
        b.own_host = m;
      } else {
        (void)hipGetLastError();
        (void)munmap(m, b.bytes);
      }
    }
    if (!b.own_host) (void)shm_unlink(b.shm_name);
  }
  blob.have_host = b.own_host != nullptr;
  blob.raw_host = (uint64_t)(uintptr_t)b.own_host;
  std::memcpy(blob.shm_name, b.shm_name, sizeof(blob.shm_name));
  if (!blob.have_device && !blob.have_host) {
    mailbox_release(c);
    return fail(c, YDC_ERR_HIP, "neither a device nor a host mailbox could be set up");
  }
  b.exported_ranks = n_ranks;
  b.exported_rank = rank;
  std::memset(out_handle, 0, YDC_IPC_HANDLE_BYTES);
  std::memcpy(out_handle, &blob, sizeof(blob));
  return YDC_OK;
}

int ydc_group_init_ipc(ydc_context* c, const void* handles, int rank, int n_ranks, int transport) {
  if (!c || !handles || n_ranks < 1 || n_ranks > (int)kMailboxMaxRanks || rank < 0 || rank >= n_ranks)
    return YDC_ERR_INVALID_ARGUMENT;
  if (transport != YDC_TRANSPORT_IPC_DEVICE && transport != YDC_TRANSPORT_IPC_HOST)
    return fail(c, YDC_ERR_INVALID_ARGUMENT, "transport %d is not a mailbox transport", transport);
  auto& g = c->group;
  auto& b = g.box;
  if (b.exported_ranks != n_ranks || b.exported_rank != rank)
    return fail(c, YDC_ERR_INVALID_ARGUMENT, "ydc_group_ipc_export(rank %d of %d) first", rank, n_ranks);
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  const bool host = transport == YDC_TRANSPORT_IPC_HOST;
  // (a second attempt with the other flavour: drop what the first one mapped)
  for (uint32_t q = 0; q < kMailboxMaxRanks; ++q) {
    if (b.opened_dev[q]) (void)hipIpcCloseMemHandle(b.opened_dev[q]);
    b.opened_dev[q] = nullptr;
    if (b.opened_host[q]) {
      (void)hipHostUnregister(b.opened_host[q]);
      (void)munmap(b.opened_host[q], b.bytes);
    }
    b.opened_host[q] = nullptr;
    b.peers.box[q] = nullptr;
  }
  b.kind = 0;
  for (int q = 0; q < n_ranks; ++q) {
    MailboxBlob pb;
    std::memcpy(&pb, (const char*)handles + (size_t)q * YDC_IPC_HANDLE_BYTES, sizeof(pb));
    if (pb.magic != kMailboxMagic || pb.rank != q || pb.n_ranks != n_ranks || pb.slot_words != b.slot_words ||
        pb.bytes != b.bytes)
      return fail(c, YDC_ERR_INVALID_ARGUMENT, "handle %d does not describe rank %d of %d with %u-word slots",
                  q, q, n_ranks, b.slot_words);
    void* dev_ptr = nullptr;
    if (!host) {
      if (!pb.have_device) return fail(c, YDC_ERR_HIP, "rank %d exported no device mailbox", q);
      if (q == rank) {
        dev_ptr = b.own_dev;
      } else if (pb.pid == (int32_t)getpid()) {
        dev_ptr = (void*)(uintptr_t)pb.raw_dev;  // a peer inside this process: its own pointer
      } else {
        hipError_t e = hipIpcOpenMemHandle(&dev_ptr, pb.handle, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
          (void)hipGetLastError();
          return fail(c, YDC_ERR_HIP, "hipIpcOpenMemHandle(rank %d): %s", q, hipGetErrorString(e));
        }
        b.opened_dev[q] = dev_ptr;
      }
    } else {
      if (!pb.have_host) return fail(c, YDC_ERR_HIP, "rank %d exported no host mailbox", q);
      void* m = nullptr;
      if (q == rank) {
        m = b.own_host;
      } else if (pb.pid == (int32_t)getpid()) {
        m = (void*)(uintptr_t)pb.raw_host;
      } else {
        int fd = shm_open(pb.shm_name, O_RDWR, 0600);
        if (fd < 0) return fail(c, YDC_ERR_HIP, "shm_open(%s) of rank %d failed", pb.shm_name, q);
        m = mmap(nullptr, b.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        (void)close(fd);
        if (m == MAP_FAILED) return fail(c, YDC_ERR_HIP, "mmap of rank %d's mailbox failed", q);
        hipError_t e = hipHostRegister(m, b.bytes, hipHostRegisterMapped | hipHostRegisterPortable);
        if (e != hipSuccess) {
          (void)hipGetLastError();
          (void)munmap(m, b.bytes);
          return fail(c, YDC_ERR_HIP, "hipHostRegister of rank %d's mailbox: %s", q, hipGetErrorString(e));
        }
        b.opened_host[q] = m;
      }
      HIP_TRY(c, hipHostGetDevicePointer(&dev_ptr, m, 0));
    }
    b.peers.box[q] = (unsigned long long*)dev_ptr;
  }
  b.kind = transport;
  b.seq = 0;
  g.rank = rank;
  g.n_ranks = n_ranks;
  return YDC_OK;
}

int ydc_group_transport(ydc_context* c) {
  if (!c) return YDC_ERR_INVALID_ARGUMENT;
  auto& g = c->group;
  if (g.n_ranks < 1) return YDC_TRANSPORT_NONE;
  if (g.comm) return YDC_TRANSPORT_RCCL;
  if (g.hub) return YDC_TRANSPORT_LOCAL;
  return g.box.kind ? g.box.kind : YDC_TRANSPORT_NONE;
}

int ydc_group_size(ydc_context* c, int* out_ranks, int* out_is_rccl) {
  if (!c || !out_ranks) return YDC_ERR_INVALID_ARGUMENT;
  auto& g = c->group;
  *out_ranks = g.n_ranks;
  if (out_is_rccl) *out_is_rccl = g.comm ? 1 : 0;
  if (g.comm && g.comm_count_fn) {
    // Ask the communicator itself (ncclCommCount), not our own bookkeeping.
    int n = 0;
    ncclResult_t r = g.comm_count_fn(g.comm, &n);
    if (r != ncclSuccess) return fail(c, YDC_ERR_HIP, "ncclCommCount failed");
    *out_ranks = n;
  }
  return YDC_OK;
}

int ydc_group_destroy(ydc_context* c) {
  if (!c) return YDC_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  group_release(c);
  return YDC_OK;
}

int ydc_dispatch_sharded(ydc_context* c, const ydc_task_soa* tk, uint32_t N, uint32_t flags,
                         uint32_t* d_out_idx, double* d_out_util, uint32_t* d_out_running) {
  if (!c || (N && !tk)) return YDC_ERR_INVALID_ARGUMENT;
  auto& g = c->group;
  if (g.n_ranks < 1) return fail(c, YDC_ERR_INVALID_ARGUMENT, "ydc_group_init first");
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  BatchPlan full_plan;
  if (int rc = plan_batch(c, N, &full_plan)) return rc;
  BatchPlan p = full_plan;
  if (p.use_generic) {
    // Registries the sharded matching does not take (> 256 classes: thread-per-chunk path):
    // every rank gathers the whole batch, places it redundantly with the single-GPU pipeline
    // — identical on all ranks — and keeps its own slice of the placement.
    const uint32_t G = (uint32_t)g.n_ranks;
    HIP_TRY(c, g.d_totals.reserve(G + 1));
    HIP_TRY(c, hipMemcpyAsync(g.d_totals.p + G, &N, 4, hipMemcpyHostToDevice, c->stream));
    if (int rc = group_all_gather(c, g.d_totals.p + G, g.d_totals.p, 4)) return rc;
    std::vector<uint32_t> sizes(G);
    HIP_TRY(c, hipMemcpyAsync(sizes.data(), g.d_totals.p, (size_t)G * 4, hipMemcpyDeviceToHost, c->stream));
    if (int rc = group_exchange_check(c)) return rc;  // (a late peer's size would read as 0)
    uint32_t max_n = 1, total = 0, my_off = 0;
    for (uint32_t r = 0; r < G; ++r) {
      max_n = std::max(max_n, sizes[r]);
      if (r < (uint32_t)g.rank) my_off += sizes[r];
      total += sizes[r];
    }
    HIP_TRY(c, g.d_pad.reserve(max_n));
    HIP_TRY(c, g.d_gather.reserve((size_t)max_n * G));
    HIP_TRY(c, g.d_all[0].reserve(total));
    HIP_TRY(c, g.d_all[1].reserve(total));
    HIP_TRY(c, g.d_all[2].reserve(total));
    HIP_TRY(c, g.d_all_idx.reserve(total));
    if (d_out_util) HIP_TRY(c, g.d_all_util.reserve(total));
    const uint32_t* cols[3] = {N ? tk->env_id : nullptr, N ? tk->min_version : nullptr,
                               N ? tk->requestor_ip : nullptr};
    for (int k = 0; k < 3; ++k) {
      if (N) HIP_TRY(c, hipMemcpyAsync(g.d_pad.p, cols[k], (size_t)N * 4, hipMemcpyDeviceToDevice, c->stream));
      if (int rc = group_all_gather(c, g.d_pad.p, g.d_gather.p, (size_t)max_n * 4)) return rc;
      uint32_t off = 0;
      for (uint32_t r = 0; r < G; ++r) {
        if (sizes[r])
          HIP_TRY(c, hipMemcpyAsync(g.d_all[k].p + off, g.d_gather.p + (size_t)r * max_n,
                                    (size_t)sizes[r] * 4, hipMemcpyDeviceToDevice, c->stream));
        off += sizes[r];
      }
    }
    // (... and a late peer's columns as zeros: nothing is placed, let alone committed, on them)
    if (int rc = group_exchange_check(c)) return rc;
    ydc_task_soa all{g.d_all[0].p, g.d_all[1].p, g.d_all[2].p};
    if (int rc = ydc_dispatch_device(c, &all, total, flags, g.d_all_idx.p,
                                     d_out_util ? g.d_all_util.p : nullptr, d_out_running))
      return rc;
    if (N && d_out_idx)
      HIP_TRY(c, hipMemcpyAsync(d_out_idx, g.d_all_idx.p + my_off, (size_t)N * 4,
                                hipMemcpyDeviceToDevice, c->stream));
    if (N && d_out_util)
      HIP_TRY(c, hipMemcpyAsync(d_out_util, g.d_all_util.p + my_off, (size_t)N * 8,
                                hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // Stats describe the whole batch in this mode; report this rank's share of the grants.
    c->stats.n_tasks = N;
    g.passes = c->stats.rounds;
    return YDC_OK;
  }
  const uint32_t G = (uint32_t)g.n_ranks, C = p.C, S = p.S, K = p.K;
  const size_t rec = (size_t)C + 1;  // ClassStates a rank publishes per pass
  const uint32_t P = c->n_parts;  // counts are exchanged per independent part of the registry
  const uint32_t MS = P + 1;  // words of a rank's k_rank_meta record
  HIP_TRY(c, g.d_totals.reserve((size_t)G * MS));
  HIP_TRY(c, g.d_meta.reserve(MS));
  HIP_TRY(c, g.d_bound_local.reserve(C ? C : 1));
  HIP_TRY(c, g.d_base.reserve(P));
  HIP_TRY(c, g.d_send.reserve(rec));
  HIP_TRY(c, g.d_bounds.reserve(rec * G));
  HIP_TRY(c, g.d_delta.reserve(S));
  HIP_TRY(c, g.d_deltas.reserve((size_t)S * G));
  if (g.h_bounds_cap < rec * G) {
    if (g.h_bounds) (void)hipHostFree(g.h_bounds);
    g.h_bounds = nullptr;
    HIP_TRY(c, hipHostMalloc((void**)&g.h_bounds, rec * G * sizeof(ClassState)));
    g.h_bounds_cap = rec * G;
  }
  hipStream_t st = c->stream;
  DeviceParams* prm = c->d_prm.p;

  // Sharded sort (kernels.h: k_key_count / k_window): this rank generates and
  // sorts only the key window its rank range can reach. Taken for integer keys, one part, a
  // servant per host, 2 .. 256 classes; anything else — and any batch whose window turns out
  // too small (window_miss, the same verdict on every rank) — runs with the full sort on
  // every rank. YDC_SHARD_SORT=0 switches it off.
  // (Decided from the registry alone — every rank must take the same branch, whatever its slice.)
  // A registry small enough for the bin sort is ordered whole on every rank (two launches, no
  // exchange) — cheaper than the key-count / window / radix sequence until the slot count is in
  // the millions; the windows are for the registries the bin sort does not take.
  bool windowed = G > 1 && c->opt_shard_sort && c->kf.exact && P == 1 && !p.use_generic && C >= 2 &&
                  !p.any_shared && !(full_plan.binsort && c->opt_group_binsort);
  BatchPlan win_plan;
  if (windowed)
    if (int rc = plan_batch(c, N, &win_plan, true)) return rc;
  uint32_t rounds = 0;
  for (;;) {
    p = windowed ? win_plan : full_plan;
    p.mb.has_successor = g.rank + 1 < g.n_ranks ? 1u : 0u;
    p.zone = false;  // (the walk of the tier's end is a single-context thing: zone_guess.h)
    p.mb.zone_box = nullptr;
    if (windowed) {
      // Slots of the window: this rank's requests + a margin on both sides (classes run ahead
      // of or behind the global level) + the granularity of the thresholds.
      const uint64_t margin = c->opt_shard_margin >= 0
                                  ? (uint64_t)c->opt_shard_margin
                                  : (uint64_t)g.margin_scale * std::max<uint64_t>(N / 8, 8192);
      const uint64_t bound = std::min<uint64_t>(
          win_plan.slot_bound, (uint64_t)N + 2 * margin + win_plan.slot_bound / 8 + 65536);
      p.win = true;
      p.win_margin = (uint32_t)std::min<uint64_t>(margin, 0x7FFFFFFFu);
      p.slot_bound = (uint32_t)bound;
      p.sort_items = p.slot_bound <= 300000 ? 2 : (p.slot_bound <= 700000 ? 4 : 8);
      p.n_tiles = std::max<uint32_t>(1, ceil_div(p.slot_bound, kSortThreads * p.sort_items));
      if (p.mb.tile_tab) {  // (the window's own tiles: ranks are registry-wide, rank_offset apart)
        p.mb.tile_tab_tiles = p.n_tiles;
        p.mb.tile_tab_elems = kSortThreads * p.sort_items;
      }
      HIP_TRY(c, g.d_cum.reserve(kWindowThresholds + 1));
      HIP_TRY(c, g.d_r_first.reserve(S));
      HIP_TRY(c, g.d_lbase.reserve((size_t)S + 1));
      HIP_TRY(c, g.d_cls_begin_glob.reserve((size_t)C + 1));
      HIP_TRY(c, g.d_shift.reserve(C));
      HIP_TRY(c, g.d_winrec.reserve((size_t)2 * C));
      HIP_TRY(c, g.d_winall.reserve((size_t)2 * C * G));
      ++g.windowed_batches;
    }
    // The chunks of this rank continue those of the nearest rank below that has requests
    // (k_global_flag / k_boundary_in copy its end state — sharded sort: translated into this
    // rank's local list positions — into d_bound_local after every exchange).
    p.mb.boundary_in = g.rank == 0 ? nullptr : g.d_bound_local.p;

    if (!windowed) {
      if (int rc = enqueue_front_a(c, p, tk)) return rc;
    } else {
      enqueue_scan(c, p, g.d_cls_begin_glob.p);
      enqueue_gen(c, p, tk, false, true);  // classification only
      if (N) {
        PrefixArgs pa{c->d_chunk_consuming.p, K, c->d_before.p, P, 0u, 0u};
        YDC_LAUNCH(c, "k_chunk_prefix", k_chunk_prefix, dim3(1), dim3(1024), 0, st, pa, prm);
      }
    }
    if (!N) HIP_TRY(c, hipMemsetAsync(c->d_before.p, 0, (size_t)4 * P, st));  // totals row of K == 0
    // Level guesses count the consuming requests of the ranks before this one; a rank without
    // requests is skipped by its successor (k_rank_meta: counts per part + number of requests).
    hipLaunchKernelGGL(k_rank_meta, dim3(1), dim3(64), 0, st, c->d_before.p + (size_t)K * P, P, N, g.d_meta.p);
    if (int rc = group_all_gather(c, g.d_meta.p, g.d_totals.p, (size_t)4 * MS)) return rc;
    if (windowed) {
      const uint32_t log_t = 7;  // kWindowThresholds == 128
      static_assert(kWindowThresholds == 128, "threshold shift");
      const uint32_t shift = c->kf.key_bits > log_t ? c->kf.key_bits - log_t : 0;
      YDC_LAUNCH(c, "k_key_count", k_key_count, dim3(kWindowThresholds - 1), dim3(256), 0, st, p.sv,
                 c->kf.cap_bits, shift, g.d_cum.p);
      WindowArgs wa{g.d_cum.p, c->kf.cap_bits, shift, g.d_totals.p, MS, (uint32_t)g.rank, G, p.win_margin,
                    p.slot_bound, c->d_slot_base.p, g.d_cls_begin_glob.p, C, g.d_r_first.p,
                    g.d_lbase.p, c->d_cls_begin.p, g.d_shift.p, g.d_winrec.p};
      YDC_LAUNCH(c, "k_window", k_window, dim3(1), dim3(1024), (size_t)2 * C * 4, st, p.sv, wa, prm);
      if (int rc = group_all_gather(c, g.d_winrec.p, g.d_winall.p, (size_t)2 * C * 4)) return rc;
      enqueue_gen(c, p, tk, true, false);  // the window's slots
      if (int rc = enqueue_sort(c, p, false)) return rc;
    }
    if (p.wave_path && p.W == 1 && c->opt_own_guess) {
      // Pass 0 works the guesses out itself, from the gathered counts.
      p.mb.before = c->d_before.p;
      p.mb.base_totals = g.d_totals.p;
      p.mb.base_rank = (uint32_t)g.rank;
      p.mb.base_stride = MS;
      if (int rc = enqueue_front_b(c, p, nullptr)) return rc;
    } else {
      hipLaunchKernelGGL(k_rank_base, dim3(1), dim3(64), 0, st, g.d_totals.p, (uint32_t)g.rank, P, MS, g.d_base.p);
      if (int rc = enqueue_front_b(c, p, g.d_base.p)) return rc;
    }
    mark(c, 6);

    // Matching passes, pre-launched in groups like on one GPU: after every pass the ranks
    // all-gather (end state of the last chunk, "changed an end state" flag); k_global_flag /
    // k_boundary_in turn the flags into one global flag per pass, which gates the following
    // passes on every rank alike. The host looks at the outcome once per group.
    uint32_t launched = 0;
    bool miss = false;
    for (;;) {
      const uint32_t group = launched == 0 ? std::max(2u, std::min(g.pass_hint, 12u)) : 3u;
      for (uint32_t r = launched; r < launched + group; ++r) {
        if (launched) {
          HIP_TRY(c, hipMemsetAsync(&prm->n_changed[r & 63], 0, 4, st));
          HIP_TRY(c, hipMemsetAsync(&prm->n_sampled[r & 63], 0, 4, st));
        }
        if (p.wave_path) enqueue_pass(c, p, r, 1u);
        hipLaunchKernelGGL(k_pack_boundary, dim3(ceil_div((uint32_t)rec, 256)), dim3(256), 0, st, p.L,
                           c->d_endst.p, p.wave_path ? K : 0u, p.mb.boundary_in, prm, r, g.d_send.p,
                           windowed ? g.d_shift.p : nullptr);
        if (int rc = group_all_gather(c, g.d_send.p, g.d_bounds.p, rec * sizeof(ClassState))) return rc;
        if (windowed) {
          hipLaunchKernelGGL(k_boundary_in, dim3(ceil_div(C, 256)), dim3(256), 0, st, g.d_bounds.p,
                             (uint32_t)rec, C, G, (uint32_t)g.rank, r, g.d_winall.p,
                             g.d_cls_begin_glob.p, g.d_totals.p, MS, g.d_shift.p, g.d_bound_local.p, prm);
        } else {
          hipLaunchKernelGGL(k_global_flag, dim3(std::max(1u, ceil_div(C, 256))), dim3(256), 0, st,
                             g.d_bounds.p, (uint32_t)rec, C, G, (uint32_t)g.rank, r, g.d_totals.p, MS,
                             c->d_cls_begin.p, g.d_bound_local.p, prm);
        }
      }
      const uint32_t first = launched;
      launched += group;
      // The tail behind the group, gated on the device by the last pass's global flag (the same
      // on every rank): placement of this rank's slice, then the global running_tasks from
      // everybody's slot deltas. Not converged yet (or a window missed): the deltas are zero
      // and nothing changes.
      if (int rc = enqueue_finalize(c, p, 0u, d_out_idx, d_out_util, nullptr, (launched - 1) & 63,
                                    g.d_delta.p, p.mb.boundary_in))
        return rc;
      if (S) {
        if (int rc = group_all_gather(c, g.d_delta.p, g.d_deltas.p, (size_t)S * 4)) return rc;
        hipLaunchKernelGGL(k_sum_deltas, dim3(ceil_div(S, 256)), dim3(256), 0, st, c->d_running.p,
                           g.d_deltas.p, S, G, c->d_running_out.p, d_out_running,
                           (flags & YDC_DISPATCH_COMMIT) ? c->d_running.p : nullptr, prm);
      }
      mark(c, 7);
      HIP_TRY(c, hipMemcpyAsync(c->h_prm, prm, sizeof(DeviceParams), hipMemcpyDeviceToHost, st));
      HIP_TRY(c, hipStreamSynchronize(st));
      HIP_TRY(c, hipGetLastError());
      if (c->debug_sim) {
        fprintf(stderr, "[ydc sharded] rank %d/%u windowed %d launched %u n_slots %u rank_offset %u miss %u changed:",
                g.rank, G, (int)windowed, launched, c->h_prm->n_slots, c->h_prm->rank_offset,
                c->h_prm->window_miss);
        for (uint32_t r = first; r < launched; ++r) fprintf(stderr, " %u", c->h_prm->n_changed[r & 63]);
        fprintf(stderr, "\n");
        if (g.rank == 0 && launched <= 30) {  // every rank's record of the last pass (registry-wide positions)
          (void)hipMemcpy(g.h_bounds, g.d_bounds.p, rec * G * sizeof(ClassState), hipMemcpyDeviceToHost);
          for (uint32_t q = 0; q < G; ++q) {
            fprintf(stderr, "   rank %u flag %u:", q, g.h_bounds[q * rec + C].cursor);
            for (uint32_t k = 0; k < std::min(C, 8u); ++k)
              fprintf(stderr, " (%u,%u,%x)", g.h_bounds[q * rec + k].cursor, g.h_bounds[q * rec + k].lo,
                      g.h_bounds[q * rec + k].hown_lo);
            fprintf(stderr, "\n");
          }
        }
      }
      if (c->h_prm->overflow) return fail(c, YDC_ERR_CAPACITY, "slot workspace overflow on some rank");
      if (c->h_prm->exchange_timeout)
        return fail(c, YDC_ERR_HIP, "a peer's data did not arrive within the mailbox time-out (rank %d of %u)",
                    g.rank, G);
      // (Belt and braces: a sharded sort that has not settled after 256 passes — the same count
      // on every rank — is treated like a missed window.)
      if (c->h_prm->window_miss || (windowed && launched >= 256)) {
        miss = true;
        break;
      }
      if (c->h_prm->n_changed[(launched - 1) & 63] == 0) {
        rounds = launched;
        for (uint32_t r = first; r < launched; ++r)
          if (c->h_prm->n_changed[r & 63] == 0) {
            rounds = r + 1;  // first pass in which no rank changed anything
            break;
          }
        g.pass_hint = rounds;
        break;
      }
      // Worst case one chunk per pass becomes final; K differs per rank, so bound it loosely.
      if (launched > 200000u) return fail(c, YDC_ERR_NOT_CONVERGED, "no fixpoint");
    }
    if (!miss) break;
    if (!windowed && p.binsort) {
      // A bin of the bin sort overflowed (bin_sort.h; the registry decides, so every rank met
      // the same): once more with the radix pipeline, which then stays.
      if (int rc = fall_back_to_radix(c, N, &full_plan)) return rc;
      continue;
    }
    if (!windowed) return fail(c, YDC_ERR_NOT_CONVERGED, "window miss flagged without a window");
    // Some rank's window did not cover what its requests reached (every rank saw the same
    // flag): once more with the full sort everywhere, and wider margins from now on.
    ++g.window_misses;
    g.margin_scale = std::min(g.margin_scale * 2, 64u);
    windowed = false;
  }
  g.passes = rounds;

  fill_stats(c, p, rounds);
  if (c->profiling) collect_kernel_profile(c);
  return YDC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Streaming mode (BASELINE.json configs[4]): a tick is
//   n_upd heartbeats of known servants  (KeepServantAlive, task_dispatcher.cc:195-201)
//   n_rel released grants               (FreeTask's --running_tasks, :181)
//   n_tasks requests, committed         (WaitForStartingNewTask x n, timeout == now)
// in this order. The whole step — staging copies, row scatter, release, the batch
// pipeline with a fixed number of pre-launched matching passes, the gated finalise and
// the result copy — is captured once into a hipGraph and replayed per tick. Lists shorter
// than the captured capacity are padded with no-ops (index 0xFFFFFFFF; requests for a
// digest nobody has), which changes nothing for the real entries. Heartbeats that change
// the registry's structure (new servant, other environments / version / host / capacity
// bound) are applied eagerly and the step is captured again.
// ---------------------------------------------------------------------------
namespace {

void stream_release(ydc_context* c) {
  auto& sm = c->stream_mode;
  if (sm.exec) (void)hipGraphExecDestroy(sm.exec);
  if (sm.graph) (void)hipGraphDestroy(sm.graph);
  if (sm.exec_b) (void)hipGraphExecDestroy(sm.exec_b);
  if (sm.graph_b) (void)hipGraphDestroy(sm.graph_b);
  sm.exec = sm.exec_b = nullptr;
  sm.graph = sm.graph_b = nullptr;
  if (sm.h_in) (void)hipHostFree(sm.h_in);
  if (sm.h_out) (void)hipHostFree(sm.h_out);
  sm.h_in = nullptr;
  sm.h_upd_idx = sm.h_rel = sm.h_env = sm.h_minv = sm.h_ip = sm.h_out = nullptr;
  sm.h_upd_rows = nullptr;
  sm.d_in.release();
  sm.active = false;
  sm.stale = true;
}

// The step itself, enqueued on the context's stream (inside a capture, or — stream_graph=0 — as it is).
int stream_enqueue_step(ydc_context* c, const BatchPlan& plan, bool by_swap) {
  auto& sm = c->stream_mode;
  hipStream_t st = c->stream;
  int rc = YDC_OK;
  auto cap = [&](hipError_t e) {
    if (e != hipSuccess && rc == YDC_OK)
      rc = fail(c, YDC_ERR_HIP, "streaming step: %s", hipGetErrorString(e));
  };
  const size_t T = sm.max_tasks;
  // Round 5: no copy node. The tick's inputs are read where the host put them (k_apply_tick and
  // the request classification read every word once), the placement is stored to the page-locked
  // result array by k_finalize, and so is the outcome block (stream_zero_copy=0: three copies).
  const bool zc = sm.zero_copy;
  if (!zc) cap(hipMemcpyAsync(sm.d_in.p, sm.h_in, sm.in_bytes, hipMemcpyHostToDevice, st));
  if (sm.max_upd + sm.max_rel) {
    const uint32_t upd_blocks = ceil_div(sm.max_upd, 256);
    hipLaunchKernelGGL(k_apply_tick, dim3(upd_blocks + ceil_div(sm.max_rel, 256)), dim3(256), 0, st,
                       zc ? sm.z_upd_idx : sm.d_upd_idx, zc ? sm.z_upd_rows : sm.d_upd_rows, sm.max_upd, upd_blocks,
                       zc ? sm.z_rel : sm.d_rel, sm.max_rel,
                       c->n_servants, c->d_version.p, c->d_nproc.p, c->d_load.p, c->d_max_tasks.p,
                       c->d_flags.p, c->d_running.p);
  }
  ydc_task_soa d{zc ? sm.z_env : sm.d_env, zc ? sm.z_minv : sm.d_minv, zc ? sm.z_ip : sm.d_ip};
  if (rc == YDC_OK) rc = enqueue_front(c, plan, &d);
  if (rc == YDC_OK && plan.wave_path)
    for (uint32_t r = 0; r < sm.passes; ++r) enqueue_pass(c, plan, r, 1u);
  const bool outcome_stored = zc && c->opt_outcome_store && plan.S != 0;
  if (rc == YDC_OK) {
    c->finalize_outcome = outcome_stored ? c->d_h_prm : nullptr;
    c->commit_by_swap = by_swap;
    rc = enqueue_finalize(c, plan, YDC_DISPATCH_COMMIT, zc ? sm.z_out : c->d_out_idx.p, nullptr, nullptr,
                          plan.wave_path ? (sm.passes - 1) & 63 : kNone);
    c->commit_by_swap = false;
    c->finalize_outcome = nullptr;
  }
  if (!zc) cap(hipMemcpyAsync(sm.h_out, c->d_out_idx.p, T * 4, hipMemcpyDeviceToHost, st));
  if (!outcome_stored) cap(hipMemcpyAsync(c->h_prm, c->d_prm.p, sizeof(DeviceParams), hipMemcpyDeviceToHost, st));
  return rc;
}

// One capture of the step with the columns as they are now (plan: made for them).
int stream_capture_one(ydc_context* c, const BatchPlan& plan, bool by_swap, hipGraph_t* g_out,
                       hipGraphExec_t* e_out) {
  hipStream_t st = c->stream;
  HIP_TRY(c, hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
  const int rc = stream_enqueue_step(c, plan, by_swap);
  hipGraph_t g = nullptr;
  hipError_t ee = hipStreamEndCapture(st, &g);
  if (ee != hipSuccess) return fail(c, YDC_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ee));
  if (rc != YDC_OK) {
    if (g) (void)hipGraphDestroy(g);
    return rc;
  }
  *g_out = g;
  HIP_TRY(c, hipGraphInstantiate(e_out, g, nullptr, nullptr, 0));
  return YDC_OK;
}

int stream_capture(ydc_context* c) {
  auto& sm = c->stream_mode;
  if (sm.exec) (void)hipGraphExecDestroy(sm.exec);
  if (sm.graph) (void)hipGraphDestroy(sm.graph);
  if (sm.exec_b) (void)hipGraphExecDestroy(sm.exec_b);
  if (sm.graph_b) (void)hipGraphDestroy(sm.graph_b);
  sm.exec = sm.exec_b = nullptr;
  sm.graph = sm.graph_b = nullptr;
  const bool was_profiling = c->profiling;
  c->profiling = false;  // no event pairs inside a capture
  // Sizes and workspace first (allocations and table uploads cannot be captured).
  if (int rc = plan_batch(c, sm.max_tasks, &sm.plan)) return rc;
  if (sm.plan.use_generic) {
    // More than 256 servant classes: the rounds of that path are checked by the host, which a
    // captured step cannot do — such ticks run eagerly (ydc_stream_tick_wide, below).
    c->profiling = was_profiling;
    sm.eager_only = true;
    sm.stale = false;
    return YDC_OK;
  }
  sm.eager_only = false;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  sm.passes = sm.want_passes ? sm.want_passes : std::max(2u, std::min(c->round_hint + 1, 12u));
  sm.window_max = sm.window_ticks = 0;
  sm.zero_copy = c->opt_stream_zero_copy;
  sm.swaps = c->opt_commit_swap && sm.plan.S != 0;
  sm.run_a = c->d_running.p;
  int rc = stream_capture_one(c, sm.plan, sm.swaps, &sm.graph, &sm.exec);
  if (rc == YDC_OK && sm.swaps) {
    // ... and once more with the two columns in each other's role.
    std::swap(c->d_running, c->d_running_out);
    sm.run_b = c->d_running.p;
    rc = plan_batch(c, sm.max_tasks, &sm.plan_b);
    if (rc == YDC_OK) rc = stream_capture_one(c, sm.plan_b, true, &sm.graph_b, &sm.exec_b);
    std::swap(c->d_running, c->d_running_out);
  }
  c->profiling = was_profiling;
  if (rc != YDC_OK) return rc;
  sm.stale = false;
  ++sm.recaptures;
  return YDC_OK;
}

}  // namespace

extern "C" {

int ydc_stream_begin(ydc_context* c, uint32_t max_updates, uint32_t max_releases,
                     uint32_t max_tasks) {
  if (!c || !max_tasks) return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  resident_stop(c);  // (the registry leaves the resident kernel's registers)
  stream_release(c);
  auto& sm = c->stream_mode;
  sm.max_upd = max_updates;
  sm.max_rel = max_releases;
  sm.max_tasks = max_tasks;
  // Arena layout (256 B aligned sections).
  size_t off = 0;
  auto section = [&](size_t bytes) {
    const size_t at = off;
    off += (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
    return at;
  };
  const size_t o_idx = section((size_t)max_updates * 4);
  const size_t o_rows = section((size_t)max_updates * sizeof(ydc_servant_row));
  const size_t o_rel = section((size_t)max_releases * 4);
  const size_t o_env = section((size_t)max_tasks * 4);
  const size_t o_minv = section((size_t)max_tasks * 4);
  const size_t o_ip = section((size_t)max_tasks * 4);
  sm.in_bytes = off;
  HIP_TRY(c, hipHostMalloc((void**)&sm.h_in, sm.in_bytes, hipHostMallocCoherent | hipHostMallocMapped));
  HIP_TRY(c, hipHostMalloc((void**)&sm.h_out, std::max<size_t>((size_t)max_tasks * 4, 16),
                           hipHostMallocCoherent | hipHostMallocMapped));
  {
    uint8_t* z_in = nullptr;
    HIP_TRY(c, hipHostGetDevicePointer((void**)&z_in, sm.h_in, 0));
    HIP_TRY(c, hipHostGetDevicePointer((void**)&sm.z_out, sm.h_out, 0));
    sm.z_upd_idx = (uint32_t*)(z_in + o_idx);
    sm.z_upd_rows = (ServantRowDev*)(z_in + o_rows);
    sm.z_rel = (uint32_t*)(z_in + o_rel);
    sm.z_env = (uint32_t*)(z_in + o_env);
    sm.z_minv = (uint32_t*)(z_in + o_minv);
    sm.z_ip = (uint32_t*)(z_in + o_ip);
  }
  HIP_TRY(c, sm.d_in.reserve(sm.in_bytes));
  sm.h_upd_idx = (uint32_t*)(sm.h_in + o_idx);
  sm.h_upd_rows = (ydc_servant_row*)(sm.h_in + o_rows);
  sm.h_rel = (uint32_t*)(sm.h_in + o_rel);
  sm.h_env = (uint32_t*)(sm.h_in + o_env);
  sm.h_minv = (uint32_t*)(sm.h_in + o_minv);
  sm.h_ip = (uint32_t*)(sm.h_in + o_ip);
  sm.d_upd_idx = (uint32_t*)(sm.d_in.p + o_idx);
  sm.d_upd_rows = (ServantRowDev*)(sm.d_in.p + o_rows);
  sm.d_rel = (uint32_t*)(sm.d_in.p + o_rel);
  sm.d_env = (uint32_t*)(sm.d_in.p + o_env);
  sm.d_minv = (uint32_t*)(sm.d_in.p + o_minv);
  sm.d_ip = (uint32_t*)(sm.d_in.p + o_ip);
  HIP_TRY(c, c->d_out_idx.reserve(max_tasks));
  sm.want_passes = sm.window_max = sm.window_ticks = 0;
  sm.active = true;
  sm.stale = true;
  return YDC_OK;
}

int ydc_stream_buffers_get(ydc_context* c, ydc_stream_buffers* out) {
  if (!c || !out || !c->stream_mode.active) return YDC_ERR_INVALID_ARGUMENT;
  auto& sm = c->stream_mode;
  out->upd_idx = sm.h_upd_idx;
  out->upd_rows = sm.h_upd_rows;
  out->release_servant_idx = sm.h_rel;
  out->env_id = sm.h_env;
  out->min_version = sm.h_minv;
  out->requestor_ip = sm.h_ip;
  out->out_servant_idx = sm.h_out;
  return YDC_OK;
}

int ydc_stream_end(ydc_context* c) {
  if (!c) return YDC_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  stream_release(c);
  return YDC_OK;
}

int ydc_stream_tick(ydc_context* c, const uint32_t* upd_idx, const ydc_servant_row* upd_rows,
                    uint32_t n_upd, const uint32_t* release_servant_idx, uint32_t n_rel,
                    const ydc_task_soa* tasks, uint32_t n_tasks, uint32_t* out_servant_idx) {
  return ydc_stream_tick_wide(c, upd_idx, upd_rows, nullptr, 1, n_upd, release_servant_idx, n_rel, tasks,
                              n_tasks, out_servant_idx);
}

int ydc_stream_tick_wide(ydc_context* c, const uint32_t* upd_idx, const ydc_servant_row* upd_rows,
                         const uint64_t* upd_env_masks, uint32_t env_words, uint32_t n_upd,
                         const uint32_t* release_servant_idx, uint32_t n_rel, const ydc_task_soa* tasks,
                         uint32_t n_tasks, uint32_t* out_servant_idx) {
  if (!c || !c->stream_mode.active) return YDC_ERR_INVALID_ARGUMENT;
  auto& sm = c->stream_mode;
  if (n_upd > sm.max_upd || n_rel > sm.max_rel || n_tasks > sm.max_tasks)
    return fail(c, YDC_ERR_CAPACITY, "tick (%u updates, %u releases, %u tasks) exceeds the capacity "
                "given to ydc_stream_begin (%u, %u, %u)", n_upd, n_rel, n_tasks, sm.max_upd,
                sm.max_rel, sm.max_tasks);
  if ((n_upd && (!upd_idx || !upd_rows)) || (n_rel && !release_servant_idx) ||
      (n_tasks && (!tasks || !out_servant_idx)))
    return YDC_ERR_INVALID_ARGUMENT;
  if (upd_env_masks && (env_words == 0 || env_words > YDC_MAX_ENV_WORDS))
    return fail(c, YDC_ERR_INVALID_ARGUMENT, "env_words %u out of range", env_words);
  HIP_TRY(c, hipSetDevice(c->device));
  // Heartbeats: structural ones (and new servants) take the eager path. With masks (the wide
  // form) a changed environment set is structural like any other change; without them, rows of
  // a table with several mask words cannot say what the servant advertises — it keeps its
  // environments, and a NEW servant (which would silently have none) is refused.
  const uint32_t EW = c->env_words;
  bool structural = false;
  if (!upd_env_masks && EW > 1)  // (every entry is looked at: the scan below stops at the first structural one)
    for (uint32_t i = 0; i < n_upd; ++i)
      if (upd_idx[i] >= c->n_servants)
        return fail(c, YDC_ERR_INVALID_ARGUMENT, "a tick that adds a servant to a table with %u mask "
                    "words needs its environments: use ydc_stream_tick_wide", EW);
  for (uint32_t i = 0; i < n_upd && !structural; ++i) {
    const uint32_t s = upd_idx[i];
    if (s >= c->n_servants) {
      structural = true;
      break;
    }
    const ydc_servant_row& r = upd_rows[i];
    bool env_changed = false;
    if (upd_env_masks) {
      for (uint32_t w = 0; w < std::max(EW, env_words); ++w) {
        const uint64_t have = w < EW ? c->h_env[(size_t)s * EW + w] : 0;
        const uint64_t want = w < env_words ? upd_env_masks[(size_t)i * env_words + w] : 0;
        env_changed |= have != want;
      }
    } else if (EW == 1) {
      env_changed = c->h_env[s] != r.env_mask;
    }
    structural = c->h_version[s] != r.version || env_changed ||
                 c->h_ip[s] != r.ip_id || (c->h_max_tasks[s] == 0) != (r.max_tasks == 0) ||
                 std::min(c->h_max_tasks[s], c->h_nproc[s]) != std::min(r.max_tasks, r.num_processors);
  }
  uint32_t graph_upd = n_upd;
  if (structural) {
    if (upd_env_masks) {
      if (int rc = ydc_update_servants_wide(c, upd_idx, upd_rows, upd_env_masks, env_words, n_upd)) return rc;
    } else if (EW == 1) {
      if (int rc = ydc_update_servants(c, upd_idx, upd_rows, n_upd)) return rc;
    } else {
      // Rows without masks on a wide table: the (known) servants keep their environments.
      std::vector<uint64_t> env((size_t)n_upd * EW, 0);
      for (uint32_t i = 0; i < n_upd; ++i)
        if (upd_idx[i] < c->n_servants)  // (new servants were refused above)
          std::copy_n(&c->h_env[(size_t)upd_idx[i] * EW], EW, &env[(size_t)i * EW]);
      if (int rc = ydc_update_servants_wide(c, upd_idx, upd_rows, env.data(), EW, n_upd)) return rc;
    }
    graph_upd = 0;
  } else {
    for (uint32_t i = 0; i < n_upd; ++i) {
      const uint32_t s = upd_idx[i];
      const ydc_servant_row& r = upd_rows[i];
      c->h_nproc[s] = r.num_processors;
      c->h_load[s] = r.current_load;
      c->h_max_tasks[s] = r.max_tasks;
      c->h_flags[s] = r.flags;
    }
  }
  if (sm.stale || c->tables_dirty)
    if (int rc = stream_capture(c)) return rc;
  // Stage the tick (padding = no-ops).
  // (a caller that filled the arena itself — ydc_stream_buffers_get — has nothing to copy)
  if (graph_upd) {
    if (upd_idx != sm.h_upd_idx) std::memcpy(sm.h_upd_idx, upd_idx, (size_t)graph_upd * 4);
    if (upd_rows != sm.h_upd_rows) std::memcpy(sm.h_upd_rows, upd_rows, (size_t)graph_upd * sizeof(ydc_servant_row));
  }
  for (uint32_t i = graph_upd; i < sm.max_upd; ++i) sm.h_upd_idx[i] = 0xFFFFFFFFu;
  if (n_rel && release_servant_idx != sm.h_rel) std::memcpy(sm.h_rel, release_servant_idx, (size_t)n_rel * 4);
  for (uint32_t i = n_rel; i < sm.max_rel; ++i) sm.h_rel[i] = 0xFFFFFFFFu;
  if (n_tasks) {
    if (tasks->env_id != sm.h_env) std::memcpy(sm.h_env, tasks->env_id, (size_t)n_tasks * 4);
    if (tasks->min_version != sm.h_minv) std::memcpy(sm.h_minv, tasks->min_version, (size_t)n_tasks * 4);
    if (tasks->requestor_ip != sm.h_ip) std::memcpy(sm.h_ip, tasks->requestor_ip, (size_t)n_tasks * 4);
  }
  for (uint32_t i = n_tasks; i < sm.max_tasks; ++i) {
    sm.h_env[i] = 0xFFFFFFFFu;  // a digest nobody has: EnvironmentNotFound, consumes nothing
    sm.h_minv[i] = 0;
    sm.h_ip[i] = 0;
  }
  if (sm.eager_only) {
    // The same step, enqueued instead of replayed: staging copy, registry deltas, the batch with
    // its host-checked rounds, COMMIT, results back.
    ++sm.eager_fallbacks;
    HIP_TRY(c, hipMemcpyAsync(sm.d_in.p, sm.h_in, sm.in_bytes, hipMemcpyHostToDevice, c->stream));
    if (sm.max_upd + sm.max_rel) {
      const uint32_t upd_blocks = ceil_div(sm.max_upd, 256);
      hipLaunchKernelGGL(k_apply_tick, dim3(upd_blocks + ceil_div(sm.max_rel, 256)), dim3(256), 0, c->stream,
                         sm.d_upd_idx, sm.d_upd_rows, sm.max_upd, upd_blocks, sm.d_rel, sm.max_rel,
                         c->n_servants, c->d_version.p, c->d_nproc.p, c->d_load.p, c->d_max_tasks.p,
                         c->d_flags.p, c->d_running.p);
    }
    BatchPlan pe;
    if (int rc = plan_batch(c, sm.max_tasks, &pe)) return rc;
    ydc_task_soa d{sm.d_env, sm.d_minv, sm.d_ip};
    uint32_t rounds_e = 0;
    if (int rc = run_planned_batch(c, pe, &d, YDC_DISPATCH_COMMIT, c->d_out_idx.p, nullptr, nullptr, &rounds_e))
      return rc;
    HIP_TRY(c, hipMemcpy(sm.h_out, c->d_out_idx.p, (size_t)sm.max_tasks * 4, hipMemcpyDeviceToHost));
    ++sm.ticks;
    fill_stats(c, pe, rounds_e);
    c->stats.n_tasks = n_tasks;
    c->stats.env_not_found -= std::min(c->stats.env_not_found, sm.max_tasks - n_tasks);  // padding
    if (n_tasks && out_servant_idx != sm.h_out) std::memcpy(out_servant_idx, sm.h_out, (size_t)n_tasks * 4);
    return YDC_OK;
  }
  const bool second = sm.swaps && c->d_running.p == sm.run_b;
  if (sm.swaps && !second && c->d_running.p != sm.run_a)
    return fail(c, YDC_ERR_NOT_CONVERGED, "streaming: the running_tasks column is neither of the captured ones");
  if (c->opt_stream_graph) {
    HIP_TRY(c, hipGraphLaunch(second ? sm.exec_b : sm.exec, c->stream));
  } else {
    // (measurement, stream_graph=0: the same step enqueued kernel by kernel instead of replayed)
    const bool was_profiling = c->profiling;
    c->profiling = false;
    const int erc = stream_enqueue_step(c, second ? sm.plan_b : sm.plan, sm.swaps);
    c->profiling = was_profiling;
    if (erc) return erc;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  ++sm.ticks;
  const BatchPlan& p = second ? sm.plan_b : sm.plan;
  // (a step that took effect left the registry's running_tasks in its output column)
  if (sm.swaps && !c->h_prm->overflow && !(p.binsort && c->h_prm->window_miss) &&
      (!p.wave_path || c->h_prm->n_changed[(sm.passes - 1) & 63] == 0))
    std::swap(c->d_running, c->d_running_out);
  if (c->h_prm->overflow)
    return fail(c, YDC_ERR_CAPACITY, "slot workspace overflow (bound %u)", p.slot_bound);
  uint32_t rounds = sm.passes;
  if (p.binsort && c->h_prm->window_miss) {
    // A bin of the bin sort overflowed (bin_sort.h): the tick's registry deltas are applied,
    // its batch was gated out. Place it eagerly with the radix sort; the step is captured
    // again, without the bin sort, on the next tick.
    ++sm.eager_fallbacks;
    BatchPlan p2;
    if (int rc = fall_back_to_radix(c, sm.max_tasks, &p2)) return rc;
    if (sm.zero_copy)  // (the captured step read the arena in place: the device copy is stale)
      HIP_TRY(c, hipMemcpyAsync(sm.d_in.p, sm.h_in, sm.in_bytes, hipMemcpyHostToDevice, c->stream));
    ydc_task_soa d{sm.d_env, sm.d_minv, sm.d_ip};
    if (int rc = run_planned_batch(c, p2, &d, YDC_DISPATCH_COMMIT, c->d_out_idx.p, nullptr, nullptr, &rounds))
      return rc;
    HIP_TRY(c, hipMemcpy(sm.h_out, c->d_out_idx.p, (size_t)sm.max_tasks * 4, hipMemcpyDeviceToHost));
    fill_stats(c, p2, rounds);
    c->stats.n_tasks = n_tasks;
    c->stats.env_not_found -= std::min(c->stats.env_not_found, sm.max_tasks - n_tasks);  // padding
    if (n_tasks && out_servant_idx != sm.h_out) std::memcpy(out_servant_idx, sm.h_out, (size_t)n_tasks * 4);
    return YDC_OK;
  }
  if (p.wave_path) {
    if (c->h_prm->n_changed[(sm.passes - 1) & 63] != 0) {
      // The captured passes were not enough (rare): finish eagerly and capture a longer
      // step next time.
      ++sm.eager_fallbacks;
      if (int rc = run_passes_until_consistent(c, p, sm.passes, YDC_DISPATCH_COMMIT, c->d_out_idx.p,
                                               nullptr, nullptr, &rounds))
        return rc;
      HIP_TRY(c, hipMemcpy(sm.h_out, c->d_out_idx.p, (size_t)sm.max_tasks * 4, hipMemcpyDeviceToHost));
      c->round_hint = rounds;
      sm.want_passes = std::min(rounds + 1, 12u);
      sm.stale = true;
    } else {
      for (uint32_t r = 0; r < sm.passes; ++r)
        if (c->h_prm->n_changed[r & 63] == 0) {
          rounds = r + 1;
          break;
        }
      sm.window_max = std::max(sm.window_max, rounds);
      if (++sm.window_ticks >= 64) {
        if (sm.window_max < sm.passes && sm.passes > 2) {
          sm.want_passes = std::max(2u, sm.window_max);
          sm.stale = true;
        }
        sm.window_max = sm.window_ticks = 0;
      }
    }
  }
  fill_stats(c, p, rounds);
  c->stats.n_tasks = n_tasks;
  c->stats.env_not_found -= std::min(c->stats.env_not_found, sm.max_tasks - n_tasks);  // padding
  if (n_tasks && out_servant_idx != sm.h_out) std::memcpy(out_servant_idx, sm.h_out, (size_t)n_tasks * 4);
  return YDC_OK;
}

}  // extern "C"

extern "C" {

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

#ifdef YDC_PHASE_PROBE
// Measurement builds only (`make probe`): the matching kernel's phase stamps, kProbeSlots per chunk.
int ydc_debug_phase_probe(unsigned long long* out, size_t n_words, int clear) {
  const size_t all = (size_t)kProbeChunks * kProbeSlots;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ydc_phase_probe), std::min(n_words, all) * 8) != hipSuccess)
    return YDC_ERR_HIP;
  if (clear) {
    std::vector<unsigned long long> z(all, 0);
    if (hipMemcpyToSymbol(HIP_SYMBOL(ydc_phase_probe), z.data(), all * 8) != hipSuccess) return YDC_ERR_HIP;
  }
  return YDC_OK;
}
#endif

int ydc_get_stats(const ydc_context* c, ydc_stats* out) {
  if (!c || !out) return YDC_ERR_INVALID_ARGUMENT;
  *out = c->stats;
  out->tick_resident_calls = (uint32_t)c->tick_resident;
  out->tick_launched_calls = (uint32_t)c->tick_launches;
  out->pipeline_batches = (uint32_t)c->pipeline_batches;
  return YDC_OK;
}

}  // extern "C"

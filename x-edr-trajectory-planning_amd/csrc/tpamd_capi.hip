// tpamd_capi.hip -- C-ABI of the engine (include/tpamd.h): workspace, launches,
// host-buffer convenience paths and per-kernel event timing.
#include "../../include/tpamd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "tpamd_kernels.h"
#include "tpamd_launch.h"
#include "tpamd_planner_set.h"
#include "tpamd_sweep_joint.h"   // LDS layout, tile size, k_rebuild_time; the kernel instances live in tpamd_sweep_inst.hip

using namespace tpamd;

namespace {

enum KernelIndex { KI_SETUP = 0, KI_SAMPLE_LP, KI_DETECT, KI_FINAL, KI_SWEEP, KI_EPILOGUE, KI_COUNT };
const char *kKernelNames[KI_COUNT] = {"k_setup", "k_sample_lp", "k_boundary_detect",
                                      "k_boundary_final", "k_sweep", "k_epilogue"};

struct EventPair {
  hipEvent_t start, stop;
  int kernel;
};

#define HIPCHK(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "[tpamd] HIP error %s at %s:%d: %s\n", #expr, __FILE__, \
                   __LINE__, hipGetErrorString(e_));                              \
      return TPAMD_E_HIP;                                                         \
    }                                                                             \
  } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// bump allocator over the staging buffer (first pass with base == null only sizes)
struct Stage {
  char *base;
  size_t off = 0;
  explicit Stage(void *b) : base((char *)b) {}
  template <typename T>
  T *take(size_t count) {
    T *p = base ? (T *)(base + off) : nullptr;
    off = align_up(off + count * sizeof(T), 256);
    return p;
  }
};

}  // namespace

struct tpamd_engine {
  int device = 0;
  void *ws_base = nullptr;
  size_t ws_bytes = 0;
  void *stage_base = nullptr;  // staging for the _host entry points
  size_t stage_bytes = 0;
  void *rows_base = nullptr;   // assembled constraint rows of Cartesian batches
  size_t rows_bytes = 0;
  Workspace ws{};
  int last_B = 0, last_N = 0;
  const double *last_time = nullptr;   // out->time of the last solve (tpamd_query_device's check)
  int profile = 0;             // 0 off, 1 every kernel, 2 the sweep kernel only
  bool force_generic = false;  // TPAMD_FORCE_GENERIC=1: A/B the specialised kernels
  bool keep_boundary = false;  // tpamd_debug_keep_boundary
  // Pipelined mode (tpamd_engine_set_pipelining): two workspaces used alternately, the front
  // stage (set-up + sampling/LP kernel) of a joint-space solve runs on the engine's own stream
  // so that it overlaps the sweep of the previous solve.
  int pipelining = 0;          // 0 off, 1 front stage on the engine's stream, 2 sweeps as well
  hipStream_t aux = nullptr;
  hipStream_t sweep_stream[2] = {nullptr, nullptr};
  hipEvent_t ev_front[2] = {nullptr, nullptr}, ev_sweep[2] = {nullptr, nullptr},
             ev_call[2] = {nullptr, nullptr};
  void *slot_base[2] = {nullptr, nullptr};   // ws_base / ws_bytes of the slot not in use
  size_t slot_bytes[2] = {0, 0};
  int slot = 0;                // workspace slot e->ws_base currently refers to
  // Concurrent groups (tpamd_time_joint_groups_*): each lane is a stream of the engine with a
  // workspace of its own; the groups of one call are spread over the lanes and run side by side.
  struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t front = nullptr;   // this lane's sampling/LP kernel has finished
    void *base = nullptr;
    size_t bytes = 0;
  };
  static constexpr int kMaxLanes = 8;
  Lane lanes[kMaxLanes];
  hipEvent_t ev_fork = nullptr;
  bool in_lane = false;        // a lane's workspace is swapped in: no slot bookkeeping
  hipEvent_t front_wait = nullptr, front_record = nullptr;   // see tpamd_time_joint_groups_device
  int phase = 0;               // joint solve: 0 all of it, 1 the front stage only, 2 the rest only
  std::vector<hipEvent_t> back_wait;   // phase 2: events the sweep's stream waits for first
  int max_lanes = 4;           // TPAMD_LANES: lanes used side by side (the runtime has 4 hardware queues)
  int chain_fronts = 2;        // TPAMD_CHAIN_FRONTS: 0 none, 1 every front stage behind the previous group's, 2 behind the heaviest group's (A/B)
  bool order_ragged = true;    // TPAMD_ORDER_RAGGED=0: A/B the longest-first order of ragged batches
  int k1_tpb = 0;              // TPAMD_K1_TPB: threads per block of the sampling/LP kernel (A/B)
  int k1_tpb_ragged = 64;      // TPAMD_K1_TPB_RAGGED: its upper limit for ragged batches (A/B)
  // Idle time in front of a pipelined front stage (TPAMD_FRONT_DELAY_US). The front stage of solve k+1
  // and the sweep of solve k become runnable at the same moment (the sweep of solve k-1 has ended); if
  // the sampling/LP kernel's 16 k blocks reach the CUs first, the sweep's workgroups (35 KB of LDS each)
  // wait for them and the overlap is lost: 0.54 instead of 0.49 ms per step, measured with a front
  // stage that starts 7.5 us after that moment; 10.5 us and everything above (up to 47) give the overlap.
  int front_delay_us = 10;
  // Event timing: pending (start, stop) pairs are folded into acc_ms/acc_n and their events
  // recycled through `pool` once kMaxPendingEvents are outstanding, so a long profiled run
  // holds a bounded number of HIP events.
  std::vector<EventPair> events;
  std::vector<EventPair> pool;
  double acc_ms[KI_COUNT] = {};
  int acc_n[KI_COUNT] = {};
};

namespace {

// Carve the workspace for (B, N, C). Returns the bytes needed; fills ws when base != null.
size_t carve_workspace(char *base, int B, int N, int C, Workspace *ws) {
  size_t off = 0;
  auto take = [&](size_t bytes) -> char * {
    char *p = base ? base + off : nullptr;
    off = align_up(off + bytes, 256);
    return p;
  };
  const size_t nb = (size_t)B, ns = (size_t)B * N;
  Workspace w{};
  w.ds = (double *)take(nb * 8);
  w.s_start = (double *)take(nb * 8);
  w.s_end = (double *)take(nb * 8);
  w.sd_start = (double *)take(nb * 8);
  w.sdd_start = (double *)take(nb * 8);
  w.t_start = (double *)take(nb * 8);
  w.delta = (double *)take(nb * 8);
  w.err_bits = (uint32_t *)take(nb * 4);
  w.lim = (double *)take(nb * 2 * C * 8);
  // + one tile of records: the sweep's tile prefetch always loads 32 whole records, so the
  // last path's partial tile reads up to 31 records past its end (tpamd_sweep_joint.h)
  w.q12 = (double *)take((ns + kTileSamples) * (C + 2) * 8);
  w.m0 = (double *)take(ns * 8);
  w.z0 = (double *)take(ns * 8);
  w.X0 = (double *)take(ns * 8);
  w.Y0 = (double *)take(ns * 8);
  w.Xz = (double *)take(ns * 8);
  w.Yz = (double *)take(ns * 8);
  w.at0 = (uint8_t *)take(ns);
  w.fix_flag = (uint8_t *)take(ns);
  w.fix_val = (double *)take(ns * 8);
  w.m = (double *)take(ns * 8);
  w.X = (double *)take(ns * 8);
  w.Y = (double *)take(ns * 8);
  w.type = (uint8_t *)take(ns);
  w.sd2 = (double *)take(ns * 8);
  w.diag = (long long *)take(nb * 64 * 8);
  w.order = (const int32_t *)take(nb * 4);
  if (ws) *ws = w;
  return off;
}

int ensure_workspace(tpamd_engine *e, int B, int N, int C) {
  const size_t need = carve_workspace(nullptr, B, N, C, nullptr);
  if (need > e->ws_bytes) {
    if (e->ws_base) HIPCHK(hipFree(e->ws_base));
    e->ws_base = nullptr;
    e->ws_bytes = 0;
    HIPCHK(hipMalloc(&e->ws_base, need));
    e->ws_bytes = need;
#ifdef TPAMD_K1_STUDY
    HIPCHK(hipMemset(e->ws_base, 0, need));
#endif
  }
  carve_workspace((char *)e->ws_base, B, N, C, &e->ws);
  e->ws.keep_boundary = e->keep_boundary ? 1 : 0;
#ifdef TPAMD_K1_STUDY
  {   // study build: both workspace slots share one timestamp buffer; the slot goes along in bit 1
    static void *g_study = nullptr;
    if (!g_study) { HIPCHK(hipMalloc(&g_study, 65536 * 8)); HIPCHK(hipMemset(g_study, 0, 65536 * 8)); }
    e->ws.diag = (long long *)g_study;
    e->ws.keep_boundary |= (e->slot & 1) << 1;
  }
#endif
  return 0;
}

// Pipelined mode: make workspace slot `slot` the current one (e->ws_base / e->ws_bytes).
void select_slot(tpamd_engine *e, int slot) {
  if (slot == e->slot) return;
  e->slot_base[e->slot] = e->ws_base;
  e->slot_bytes[e->slot] = e->ws_bytes;
  e->ws_base = e->slot_base[slot];
  e->ws_bytes = e->slot_bytes[slot];
  e->slot = slot;
}

bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
}

// Every entry point that touches the CURRENT workspace slot from the caller's stream while the
// engine has streams of its own (a pipelined mode is or was on) goes through this guard: the
// caller's stream first waits for the front stage and the sweep that last used the slot (they may
// still be running on the engine's streams), and the slot's "sweep done" event is re-recorded on
// the caller's stream afterwards, so that the next pipelined front stage that takes the slot is
// ordered behind this user. A stream that is being captured cannot wait for outside events: the
// caller must have fenced the engine before capturing (include/tpamd.h).
struct SlotGuard {
  tpamd_engine *e;
  hipStream_t st;
  int slot;
  bool on;
  SlotGuard(tpamd_engine *e_, hipStream_t st_, bool record_after = true, bool enable = true)
      : e(e_), st(st_), slot(e_->slot),
        on(enable && !e_->in_lane && e_->aux != nullptr && !stream_is_capturing(st_)) {
    if (!on) return;
    (void)hipStreamWaitEvent(st, e->ev_front[slot], 0);
    (void)hipStreamWaitEvent(st, e->ev_sweep[slot], 0);
    on = record_after;
  }
  ~SlotGuard() {
    if (on) (void)hipEventRecord(e->ev_sweep[slot], st);
  }
};

int ensure_stage(tpamd_engine *e, size_t need) {
  if (need > e->stage_bytes) {
    if (e->stage_base) HIPCHK(hipFree(e->stage_base));
    e->stage_base = nullptr;
    e->stage_bytes = 0;
    HIPCHK(hipMalloc(&e->stage_base, need));
    e->stage_bytes = need;
  }
  return 0;
}

int ensure_rows(tpamd_engine *e, size_t need) {
  if (need > e->rows_bytes) {
    if (e->rows_base) HIPCHK(hipFree(e->rows_base));
    e->rows_base = nullptr;
    e->rows_bytes = 0;
    HIPCHK(hipMalloc(&e->rows_base, need));
    e->rows_bytes = need;
  }
  return 0;
}

constexpr size_t kMaxPendingEvents = 512;

// Fold every pending event pair into the per-kernel sums (waits for the recorded work) and
// hand the events back to the pool.
void fold_events(tpamd_engine *e) {
  for (auto &ev : e->events) {
    float ms = 0.f;
    if (hipEventSynchronize(ev.stop) == hipSuccess &&
        hipEventElapsedTime(&ms, ev.start, ev.stop) == hipSuccess) {
      e->acc_ms[ev.kernel] += ms;
      e->acc_n[ev.kernel]++;
    }
    e->pool.push_back(ev);
  }
  e->events.clear();
}

struct Timer {
  tpamd_engine *e;
  hipStream_t st;
  int kernel;
  EventPair ev{};
  bool on;
  Timer(tpamd_engine *e_, hipStream_t st_, int k)
      : e(e_), st(st_), kernel(k), on(e_->profile == 1 || (e_->profile == 2 && k == KI_SWEEP)) {
    if (!on) return;
    if (e->events.size() >= kMaxPendingEvents) fold_events(e);
    if (!e->pool.empty()) {
      ev = e->pool.back();
      e->pool.pop_back();
    } else if (hipEventCreate(&ev.start) != hipSuccess || hipEventCreate(&ev.stop) != hipSuccess) {
      on = false;
      return;
    }
    ev.kernel = kernel;
    (void)hipEventRecord(ev.start, st);
  }
  ~Timer() {
    if (on) {
      (void)hipEventRecord(ev.stop, st);
      e->events.push_back(ev);
    }
  }
};

// Makes e->device current for the duration of an entry point and restores the caller's
// device afterwards (the caller may be PyTorch with another device current).
struct DeviceScope {
  int prev = -1;
  hipError_t err;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    err = (prev == device) ? hipSuccess : hipSetDevice(device);
    if (prev == device) prev = -1;   // nothing to restore
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define TPAMD_ON_DEVICE(e)          \
  DeviceScope device_scope_((e)->device); \
  HIPCHK(device_scope_.err)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: it is set
// for every kernel that can ask for more than 64 KB of LDS, once per device ordinal, when the
// first engine on that device is created (thread-safe; a failure is reported).
constexpr int kMaxDevices = 64;
std::mutex g_config_mutex;
bool g_device_configured[kMaxDevices] = {};

template <typename K>
hipError_t allow_big_lds(K kernel, int bytes = 160 * 1024) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// The current device must be `device`.
int configure_kernels_for_device(int device) {
  if (device < 0 || device >= kMaxDevices) return TPAMD_E_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(g_config_mutex);
  if (g_device_configured[device]) return 0;
#define TPAMD_BIG_LDS(...) HIPCHK(allow_big_lds(__VA_ARGS__))
  TPAMD_BIG_LDS(k_sample_lp_joint<1, 0>);
  TPAMD_BIG_LDS(k_sample_lp_joint<1, 3>);
  TPAMD_BIG_LDS(k_sample_lp_joint<1, 4>);
  TPAMD_BIG_LDS(k_sample_lp_joint<1, 5>);
  TPAMD_BIG_LDS(k_sample_lp_joint<1, 6>);
  TPAMD_BIG_LDS(k_sample_lp_joint<1, 7>);
  TPAMD_BIG_LDS(k_sample_lp_joint<1, 8>);
  TPAMD_BIG_LDS(k_sample_lp_joint<1, 14>);
  TPAMD_BIG_LDS(k_sample_lp_joint_wide<1, 14>);
  TPAMD_BIG_LDS(k_lp_rows<1>);
  TPAMD_BIG_LDS(k_lp_rows<2>);
  TPAMD_BIG_LDS(k_sweep<JointSource>);
  TPAMD_BIG_LDS(k_sweep<GenericSource>);
#define TPAMD_CONFIGURE_SWEEP(D, E) HIPCHK((configure_sweep_joint<D, E>()));
  TPAMD_SWEEP_INSTANCES(TPAMD_CONFIGURE_SWEEP)
#undef TPAMD_CONFIGURE_SWEEP
  TPAMD_BIG_LDS(k_cartesian_lp<1, 6>);
  TPAMD_BIG_LDS(k_cartesian_lp<1, 7>);
  TPAMD_BIG_LDS(k_resample_skip, 128 * 1024);   // int[N], N <= 32768, next to 8 B static
#undef TPAMD_BIG_LDS
  g_device_configured[device] = true;
  return 0;
}

// Joint counts with a specialised sweep kernel (which also runs the boundary passes 2-4 of
// its path and the planner epilogue).
bool has_joint_sweep(int D) { return (D >= 3 && D <= 8) || D == 14; }

// The sweep launch: joint-space batches with D in {3..8, 14} take the specialised kernel
// (6, 7, 14 are the BASELINE.json configurations), everything else the generic one.
// Returns true if the kernel also wrote qd/qdd (the planner epilogue).
template <class Source>
bool launch_sweep(hipStream_t st, int B, int N, int max_loops, const Source &src,
                  const Workspace &ws, const tpamd_path_outputs *out, bool /*force_generic*/) {
  const size_t lds = (2 * (size_t)N + 64) * sizeof(double);
  hipLaunchKernelGGL((k_sweep<Source>), dim3(B), dim3(64), lds, st, N, max_loops, src, ws,
                     out->time, out->s, out->sd, out->sdd, out->last_extremal_index,
                     out->max_time_increment, out->status);
  return false;
}

template <>
bool launch_sweep<JointSource>(hipStream_t st, int B, int N, int max_loops, const JointSource &src,
                               const Workspace &ws, const tpamd_path_outputs *out,
                               bool force_generic) {
#define TPAMD_LAUNCH_JOINT(DD)                                              \
  do {                                                                      \
    launch_sweep_joint<DD, 0>(B, st, N, max_loops, src, ws, out);           \
    return true;                                                            \
  } while (0)
  if (!force_generic) {
    switch (src.D) {
      case 3: TPAMD_LAUNCH_JOINT(3);
      case 4: TPAMD_LAUNCH_JOINT(4);
      case 5: TPAMD_LAUNCH_JOINT(5);
      case 6: TPAMD_LAUNCH_JOINT(6);
      case 7: TPAMD_LAUNCH_JOINT(7);
      case 8: TPAMD_LAUNCH_JOINT(8);
      case 14: TPAMD_LAUNCH_JOINT(14);
      default: break;
    }
  }
#undef TPAMD_LAUNCH_JOINT
  const size_t lds = (2 * (size_t)N + 64) * sizeof(double);
  hipLaunchKernelGGL((k_sweep<JointSource>), dim3(B), dim3(64), lds, st, N, max_loops, src, ws,
                     out->time, out->s, out->sd, out->sdd, out->last_extremal_index,
                     out->max_time_increment, out->status);
  return false;
}

template <class Source> bool sweep_runs_boundary_passes(const tpamd_engine *, const Source &) { return false; }
template <> bool sweep_runs_boundary_passes<JointSource>(const tpamd_engine *e, const JointSource &src) {
  return !e->force_generic && has_joint_sweep(src.D);
}

// Shared tail: boundary passes 2-4 (separate kernels unless the sweep kernel runs them itself)
// -> sweep (-> epilogue in joint mode).
template <class Source>
int run_boundary_and_sweep(tpamd_engine *e, hipStream_t st, int B, int N, int max_loops,
                           const Source &src, const tpamd_path_outputs *out,
                           bool *epilogue_done = nullptr) {
  e->ws.sd2_out = out->sd2;
  const Workspace &ws = e->ws;
  const dim3 grid_s((N + 255) / 256, B);
  if (!sweep_runs_boundary_passes(e, src)) {
    {
      Timer t(e, st, KI_DETECT);
      hipLaunchKernelGGL((k_boundary_zfit<Source>), grid_s, dim3(256), 0, st, N, src, ws);
      hipLaunchKernelGGL(k_boundary_detect, grid_s, dim3(256), 0, st, N, ws);
    }
    {
      Timer t(e, st, KI_FINAL);
      hipLaunchKernelGGL((k_boundary_final<Source>), grid_s, dim3(256), 0, st, N, src, ws);
    }
  }
  {
    Timer t(e, st, KI_SWEEP);
    const bool fused = launch_sweep(st, B, N, max_loops, src, ws, out, e->force_generic);
    if (epilogue_done) *epilogue_done = fused;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// Cartesian batches with D in {6, 7}: records carry two extra B-only rows.
template <int DD>
void launch_sweep_cartesian(hipStream_t st, int B, int N, int max_loops, const JointSource &src,
                            const Workspace &ws, const tpamd_path_outputs *out) {
  launch_sweep_joint<DD, 2>(B, st, N, max_loops, src, ws, out);
}

// LP boundary points of explicit rows, then the shared tail.
int run_rows(tpamd_engine *e, hipStream_t st, int B, int N, int C, int max_loops, const double *a,
             const double *b, const double *lower, const double *upper,
             const tpamd_path_outputs *out) {
  const Workspace &ws = e->ws;
  {
    Timer t(e, st, KI_SAMPLE_LP);
    const int tpb = 64;
    const size_t lds = 4 * (size_t)C * tpb * 8;
    const dim3 grid((N + tpb - 1) / tpb, B);
    if (C <= 32)
      hipLaunchKernelGGL((k_lp_rows<1>), grid, dim3(tpb), lds, st, N, C, a, b, lower, upper, ws);
    else
      hipLaunchKernelGGL((k_lp_rows<2>), grid, dim3(tpb), lds, st, N, C, a, b, lower, upper, ws);
  }
  GenericSource src;
  src.A = a; src.B = b; src.LO = lower; src.HI = upper; src.C = C;
  return run_boundary_and_sweep(e, st, B, N, max_loops, src, out);
}

}  // namespace

extern "C" {

int tpamd_version(void) { return TPAMD_VERSION; }

const char *tpamd_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case TPAMD_E_INVALID_ARGUMENT: return "invalid argument";
    case TPAMD_E_HIP: return "HIP runtime error";
    case TPAMD_E_UNSUPPORTED: return "unsupported size";
    case TPAMD_E_NO_DEVICE: return "no HIP device";
    case TPAMD_E_STALE: return "engine state belongs to a different solve";
    case TPAMD_PATH_INFEASIBLE_BOUNDS: return "infeasible bounds: no upper > lower at a sample";
    case TPAMD_PATH_S_RANGE: return "s_start must be < s_end";
    case TPAMD_PATH_SD_START_NEGATIVE: return "sd_start must be >= 0";
    case TPAMD_PATH_LOWER_GE_UPPER: return "constraints must satisfy lower < upper";
    case TPAMD_PATH_TOO_FEW_SAMPLES: return "need at least 2 samples";
    case TPAMD_PATH_NO_CONNECTION: return "could not connect from critical point to initial trajectory portion";
    case TPAMD_PATH_NAN_SD2: return "no solution found (NaN in sd2)";
    case TPAMD_PATH_NONZERO_END: return "non-zero terminal velocity";
    case TPAMD_PATH_CRIT_INDEX_ZERO: return "critical point search degenerated to index 0";
    default: return "unknown";
  }
}

int tpamd_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 0) return 0;
  return count;
}

void tpamd_shard_bounds(int num_paths, int num_shards, int32_t *begin) {
  if (!begin || num_shards <= 0) return;
  const int total = num_paths > 0 ? num_paths : 0;
  const int base = total / num_shards, rem = total % num_shards;
  for (int r = 0; r <= num_shards; r++) begin[r] = r * base + (r < rem ? r : rem);
}

void tpamd_shard_bounds_balanced(int num_paths, const double *cost, int num_shards, int32_t *begin) {
  if (!begin || num_shards <= 0) return;
  const int n = (num_paths > 0 && cost) ? num_paths : 0;
  double total = 0.0;
  for (int i = 0; i < n; i++) total += cost[i];
  int k = 1, lo = 0;
  double acc = 0.0;
  begin[0] = 0;
  for (int i = 0; i < n; i++) {
    acc += cost[i];
    const int remaining_items = n - (i + 1), remaining_blocks = num_shards - k;
    if (k < num_shards && (acc >= total * k / num_shards || remaining_items == remaining_blocks)) {
      begin[k] = i + 1;
      lo = i + 1;
      k++;
    }
  }
  (void)lo;
  for (; k <= num_shards; k++) begin[k] = n;
}

int tpamd_engine_create(int device_ordinal, tpamd_engine **out) {
  if (!out) return TPAMD_E_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    std::fprintf(stderr, "[tpamd] no HIP device available: the engine has no CPU fallback\n");
    return TPAMD_E_NO_DEVICE;
  }
  if (device_ordinal < 0 || device_ordinal >= count) return TPAMD_E_INVALID_ARGUMENT;
  DeviceScope scope(device_ordinal);
  HIPCHK(scope.err);
  const int rc = configure_kernels_for_device(device_ordinal);
  if (rc) return rc;
  tpamd_engine *e = new (std::nothrow) tpamd_engine();
  if (!e) return TPAMD_E_HIP;
  e->device = device_ordinal;
  {
    const char *fg = std::getenv("TPAMD_FORCE_GENERIC");
    e->force_generic = fg && fg[0] == '1';
    const char *tr = std::getenv("TPAMD_K1_TPB_RAGGED");
    if (tr && (atoi(tr) == 64 || atoi(tr) == 128 || atoi(tr) == 256)) e->k1_tpb_ragged = atoi(tr);
    const char *fd = std::getenv("TPAMD_FRONT_DELAY_US");
    if (fd && std::atoi(fd) >= 0 && std::atoi(fd) <= 1000) e->front_delay_us = std::atoi(fd);
    const char *ml = std::getenv("TPAMD_LANES");
    if (ml && std::atoi(ml) >= 1 && std::atoi(ml) <= tpamd_engine::kMaxLanes) e->max_lanes = std::atoi(ml);
    const char *cf = std::getenv("TPAMD_CHAIN_FRONTS");
    if (cf && cf[0] >= '0' && cf[0] <= '3') e->chain_fronts = cf[0] - '0';
    const char *orr = std::getenv("TPAMD_ORDER_RAGGED");
    e->order_ragged = !(orr && orr[0] == '0');
    const char *tb = std::getenv("TPAMD_K1_TPB");
    e->k1_tpb = tb ? std::atoi(tb) : 0;
  }
  *out = e;
  return 0;
}

void tpamd_engine_destroy(tpamd_engine *e) {
  if (!e) return;
  DeviceScope scope(e->device);
  for (auto *v : {&e->events, &e->pool})
    for (auto &ev : *v) { (void)hipEventDestroy(ev.start); (void)hipEventDestroy(ev.stop); }
  if (e->ws_base) (void)hipFree(e->ws_base);
  if (e->slot_base[1 - e->slot]) (void)hipFree(e->slot_base[1 - e->slot]);
  for (int k = 0; k < 2; k++) {
    if (e->ev_front[k]) (void)hipEventDestroy(e->ev_front[k]);
    if (e->ev_sweep[k]) (void)hipEventDestroy(e->ev_sweep[k]);
    if (e->ev_call[k]) (void)hipEventDestroy(e->ev_call[k]);
    if (e->sweep_stream[k]) (void)hipStreamDestroy(e->sweep_stream[k]);
  }
  if (e->aux) (void)hipStreamDestroy(e->aux);
  for (auto &ln : e->lanes) {
    if (ln.stream) (void)hipStreamDestroy(ln.stream);
    if (ln.done) (void)hipEventDestroy(ln.done);
    if (ln.front) (void)hipEventDestroy(ln.front);
    if (ln.base) (void)hipFree(ln.base);
  }
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->stage_base) (void)hipFree(e->stage_base);
  if (e->rows_base) (void)hipFree(e->rows_base);
  delete e;
}

int tpamd_engine_reserve(tpamd_engine *e, int B, int N, int C) {
  if (!e || B <= 0 || N <= 0 || C <= 0) return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  int rc = ensure_workspace(e, B, N, C);
  if (rc == 0 && e->pipelining) {       // both workspaces of the pipelined mode
    select_slot(e, 1 - e->slot);
    rc = ensure_workspace(e, B, N, C);
  }
  return rc;
}

size_t tpamd_engine_workspace_bytes(const tpamd_engine *e) { return e ? e->ws_bytes : 0; }

}  // extern "C"

namespace {
// The joint-space solve; `plan` (window chaining, tpamd_plan_joint_windows_host) adds two small
// kernels: skip marks after the set-up kernel, the start-velocity projection after K1.
int solve_joint(tpamd_engine *e, const tpamd_joint_batch *bt, const tpamd_joint_inputs *in,
                const tpamd_path_outputs *out, void *hip_stream, const PlanParams *plan,
                bool allow_pipelining = true) {
  if (!e || !bt || !in || !out) return TPAMD_E_INVALID_ARGUMENT;
  const int B = bt->num_paths, D = bt->num_dofs, N = bt->num_samples, P = bt->num_points;
  if (B <= 0) return B == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  if (D < 1 || D > 16 || N < 3 || N > 8192 || P < 3) return TPAMD_E_UNSUPPORTED;
  if (!in->knots || !in->control_points || !in->max_velocity || !in->max_acceleration ||
      !in->path_start || !in->delta || !in->sd_start || !in->time_start || !out->time ||
      !out->s || !out->sd || !out->sdd || !out->status)
    return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  hipStream_t st = (hipStream_t)hip_stream;
  const int C = 2 * D;
  // Pipelined mode: this solve takes the workspace the previous one did not use, and its front
  // stage goes to the engine's stream, ordered only behind the sweep that last used this
  // workspace -- not behind the caller's stream (see tpamd_engine_set_pipelining). A stream that
  // is being captured into a graph cannot fork into the engine's stream: plain order then.
  bool piped = e->pipelining != 0 && plan == nullptr && allow_pipelining;
  if (piped && stream_is_capturing(st)) piped = false;
  // the block size of the sampling/LP kernel and its LDS: checked before anything is launched or
  // the workspace slot changes
  int tpb = piped ? 128 : (D <= 7 ? 256 : (D < 14 ? 128 : 64));
  if (!piped && in->num_samples_per_path && tpb > e->k1_tpb_ragged) tpb = e->k1_tpb_ragged;
  if (e->k1_tpb == 64 || e->k1_tpb == 128 || e->k1_tpb == 256) tpb = e->k1_tpb;
  const size_t lds = ((size_t)(P + 3) + (size_t)P * D + 2 * C + 2 * (size_t)D * tpb) * 8;
  if (lds > 160 * 1024) return TPAMD_E_UNSUPPORTED;
  if (piped) select_slot(e, 1 - e->slot);
  const int slot = e->slot;
  // an unpipelined solve on an engine with streams of its own: ordered against them
  SlotGuard slot_guard(e, st, /*record_after=*/true, /*enable=*/!piped);
  int rc = ensure_workspace(e, B, N, C);
  if (rc) return rc;
  e->last_B = B; e->last_N = N; e->last_time = out->time;
  e->ws.ns = in->num_samples_per_path;
  e->ws.amax = in->max_acceleration;
  // ragged batches: sweep workgroups take the paths longest first (k_order_paths, below)
  int32_t *order = const_cast<int32_t *>(e->ws.order);
  const bool ordered = in->num_samples_per_path != nullptr && B > 1 && e->order_ragged && plan == nullptr;
  if (!ordered) e->ws.order = nullptr;
  const Workspace &ws = e->ws;
  const int max_loops = bt->max_solver_loops > 0 ? bt->max_solver_loops : 0;  // 0: per path, max(100, 10 n)
  hipStream_t fs = st;                  // stream of the front stage
  if (piped) {
    fs = e->aux;
    HIPCHK(hipStreamWaitEvent(fs, e->ev_sweep[slot], 0));
  }
  const bool do_front = e->phase != 2, do_back = e->phase != 1;
  if (do_front && piped && e->front_delay_us > 0)
    hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, fs, e->front_delay_us * 100);
  if (do_front) {
    Timer t(e, fs, KI_SETUP);
    hipLaunchKernelGGL(k_setup_joint, dim3((B + 127) / 128), dim3(128), 0, fs, B, N, D,
                       bt->constraint_safety, in->max_velocity, in->max_acceleration,
                       in->path_start, in->delta, in->sd_start, in->sdd_start, in->time_start,
                       ws);
    if (plan)
      hipLaunchKernelGGL(k_plan_mark_skipped, dim3((B + 127) / 128), dim3(128), 0, fs, *plan, ws);
    if (ordered)
      hipLaunchKernelGGL(k_order_paths, dim3(1), dim3(1024), 0, fs, B, N, in->num_samples_per_path, order);
  }
  if (do_front && e->front_wait) HIPCHK(hipStreamWaitEvent(fs, e->front_wait, 0));
  if (do_front) {
    Timer t(e, fs, KI_SAMPLE_LP);
    // (tpb, above) 128 threads (16 KB of LDS at D = 7) fit next to four resident sweep workgroups
    // of an earlier solve. Otherwise the block size follows the record width (measured, K1 alone:
    // D <= 7 0.166 / 0.174 / 0.186 ms at 256 / 128 / 64 threads; D = 14, N = 4000 0.765 / 0.734 /
    // 0.661 ms) and ragged batches take 64-thread blocks (fewer idle threads behind a path's end:
    // the mixed-DOF share of configs[4] 2.96 -> 2.60 ms).
    const dim3 grid((N + tpb - 1) / tpb, B);
#define TPAMD_K1(DD)                                                                         \
  hipLaunchKernelGGL((k_sample_lp_joint<1, DD>), grid, dim3(tpb), lds, fs, N, D, P, in->knots, \
                     in->control_points, out->q, ws)
    if (D == 7 && !e->force_generic) TPAMD_K1(7);
    else if (D == 6 && !e->force_generic) TPAMD_K1(6);
    else if (D == 14 && !e->force_generic)
      hipLaunchKernelGGL((k_sample_lp_joint_wide<1, 14>), grid, dim3(tpb), lds, fs, N, D, P, in->knots,
                         in->control_points, out->q, ws);
    else if (D == 3 && !e->force_generic) TPAMD_K1(3);
    else if (D == 4 && !e->force_generic) TPAMD_K1(4);
    else if (D == 5 && !e->force_generic) TPAMD_K1(5);
    else if (D == 8 && !e->force_generic) TPAMD_K1(8);
    else TPAMD_K1(0);
#undef TPAMD_K1
    if (plan) hipLaunchKernelGGL(k_plan_project, dim3((B + 127) / 128), dim3(128), 0, fs, *plan, ws);
  }
  if (do_front && e->front_record) HIPCHK(hipEventRecord(e->front_record, fs));
  if (!do_back) {
    HIPCHK(hipGetLastError());
    return 0;
  }
  for (hipEvent_t ev : e->back_wait) HIPCHK(hipStreamWaitEvent(st, ev, 0));
  // Mode 2: the sweep goes to one of two engine streams as well, so that it can start while the
  // previous solve's slowest paths are still running; it is ordered behind this call's position
  // in the caller's stream (the output buffers are free) and behind its own front stage. The
  // caller's stream is made to wait for the PREVIOUS solve only (tpamd_engine_fence for the rest).
  const bool deferred = piped && e->pipelining == 2;
  hipStream_t caller = st;
  if (piped) {
    HIPCHK(hipEventRecord(e->ev_front[slot], fs));
    if (deferred) {
      HIPCHK(hipEventRecord(e->ev_call[slot], caller));
      st = e->sweep_stream[slot];
      HIPCHK(hipStreamWaitEvent(st, e->ev_call[slot], 0));
    }
    HIPCHK(hipStreamWaitEvent(st, e->ev_front[slot], 0));
  }
  JointSource src;
  src.q12 = ws.q12; src.lim = ws.lim; src.D = D;
  bool fused = false;
  rc = run_boundary_and_sweep(e, st, B, N, max_loops, src, out, &fused);
  if (rc) {
    if (piped) (void)hipEventRecord(e->ev_sweep[slot], st);   // the slot's next user waits for what was launched
    return rc;
  }
  if (!fused && (out->qd || out->qdd)) {   // the specialised sweep kernels write qd/qdd themselves
    Timer t(e, st, KI_EPILOGUE);
    const size_t total = (size_t)B * N * D;
    hipLaunchKernelGGL(k_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, B, N,
                       D, ws.q12, out->sd, out->sdd, in->max_acceleration, out->status, ws.ns,
                       out->qd, out->qdd);
  }
  if (piped) HIPCHK(hipEventRecord(e->ev_sweep[slot], st));
  if (deferred) HIPCHK(hipStreamWaitEvent(caller, e->ev_sweep[1 - slot], 0));
  HIPCHK(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" {

int tpamd_time_joint_paths_device(tpamd_engine *e, const tpamd_joint_batch *bt,
                                  const tpamd_joint_inputs *in, const tpamd_path_outputs *out,
                                  void *hip_stream) {
  return solve_joint(e, bt, in, out, hip_stream, nullptr);
}

int tpamd_plan_joint_windows_host(tpamd_engine *e, const tpamd_plan_args *a) {
  if (!e || !a) return TPAMD_E_INVALID_ARGUMENT;
  if (a->num_planners <= 0) return a->num_planners == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  const size_t B = a->num_planners, D = a->num_dofs, N = a->num_samples, P = a->num_points,
               cap = a->history_capacity;
  if (D < 1 || D > 16 || N < 3 || N > 8192 || P < 3 || cap < N) return TPAMD_E_UNSUPPORTED;
  if (!a->knots || !a->control_points || !a->max_velocity || !a->max_acceleration || !a->delta ||
      !a->initial_velocity || !a->start_ns || !a->horizon_ns || !a->path_state || !a->planned_to_end ||
      !a->history_count || !a->history_time || !a->history_s || !a->history_sd || !a->history_sdd ||
      !a->history_q || !a->history_qd || !a->history_qdd || !a->path_horizon || !a->final_decel_start_ns ||
      !a->window_time || !a->window_s || !a->window_sd || !a->window_sdd || !a->window_sd2 || !a->window_q ||
      !a->window_q1 || !a->window_q2 || !a->window_path_start || !a->window_sd_start ||
      !a->window_time_start || !a->window_last_extremal_index || !a->window_max_time_increment ||
      !a->status || !a->windows || !a->loop_start_ns || !a->loop_count || !a->looping)
    return TPAMD_E_INVALID_ARGUMENT;
  for (size_t b = 0; b < B; b++)
    if (a->history_count[b] < 0 || (size_t)a->history_count[b] > cap) return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  hipStream_t st = nullptr;
  for (int pass = 0; pass < 2; pass++) {
    Stage s(pass ? e->stage_base : nullptr);
    double *d_knots = s.take<double>(B * (P + 3)), *d_cp = s.take<double>(B * P * D);
    double *d_vmax = s.take<double>(B * D), *d_amax = s.take<double>(B * D), *d_dl = s.take<double>(B);
    double *d_iv = s.take<double>(B * D);
    long long *d_start = s.take<long long>(B), *d_hor = s.take<long long>(B), *d_loop_start = s.take<long long>(B),
              *d_fds = s.take<long long>(B);
    int *d_state = s.take<int>(B), *d_count = s.take<int>(B), *d_pte = s.take<int>(B), *d_active = s.take<int>(B),
        *d_old = s.take<int>(B), *d_off = s.take<int>(B), *d_loop = s.take<int>(B), *d_app = s.take<int>(B),
        *d_win = s.take<int>(B), *d_status = s.take<int>(B), *d_nact = s.take<int>(1);
    double *h_t = s.take<double>(B * cap), *h_s = s.take<double>(B * cap), *h_sd = s.take<double>(B * cap),
           *h_sdd = s.take<double>(B * cap);
    double *h_q = s.take<double>(B * cap * D), *h_qd = s.take<double>(B * cap * D), *h_qdd = s.take<double>(B * cap * D);
    double *d_ph = s.take<double>(B), *d_ps = s.take<double>(B), *d_sd0 = s.take<double>(B), *d_t0 = s.take<double>(B),
           *d_sdd0 = s.take<double>(B);
    double *w_t = s.take<double>(B * N), *w_s = s.take<double>(B * N), *w_sd = s.take<double>(B * N),
           *w_sdd = s.take<double>(B * N), *w_sd2 = s.take<double>(B * N);
    double *w_q = s.take<double>(B * N * D), *w_qd = s.take<double>(B * N * D), *w_qdd = s.take<double>(B * N * D);
    double *w_q1 = s.take<double>(B * N * D), *w_q2 = s.take<double>(B * N * D), *w_dtm = s.take<double>(B);
    int32_t *w_lei = s.take<int32_t>(B), *w_st = s.take<int32_t>(B);
    if (!pass) {
      int rc = ensure_stage(e, s.off);
      if (rc) return rc;
      continue;
    }
    HIPCHK(hipMemcpyAsync(d_knots, a->knots, B * (P + 3) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_cp, a->control_points, B * P * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_vmax, a->max_velocity, B * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_amax, a->max_acceleration, B * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_dl, a->delta, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_iv, a->initial_velocity, B * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_start, a->start_ns, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_loop_start, a->resume ? a->loop_start_ns : a->start_ns, B * 8,
                          hipMemcpyHostToDevice, st));   // :630
    HIPCHK(hipMemcpyAsync(d_hor, a->horizon_ns, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_state, a->path_state, B * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_count, a->history_count, B * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_pte, a->planned_to_end, B * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_ph, a->path_horizon, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_fds, a->final_decel_start_ns, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(h_t, a->history_time, B * cap * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(h_s, a->history_s, B * cap * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(h_sd, a->history_sd, B * cap * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(h_sdd, a->history_sdd, B * cap * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(h_q, a->history_q, B * cap * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(h_qd, a->history_qd, B * cap * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(h_qdd, a->history_qdd, B * cap * D * 8, hipMemcpyHostToDevice, st));
    // loop state: a planner loops while it has not planned to the end (:632); nothing solved yet
    std::vector<int> act(B), zero(B, 0);
    int n_active = 0;
    for (size_t b = 0; b < B; b++) {
      act[b] = a->resume ? (a->looping[b] != 0) : (a->planned_to_end[b] ? 0 : 1);
      n_active += act[b];
    }
    HIPCHK(hipMemcpyAsync(d_active, act.data(), B * 4, hipMemcpyHostToDevice, st));
    for (int *z : {d_old, d_off, d_loop, d_app, d_win, d_status})
      HIPCHK(hipMemcpyAsync(z, zero.data(), B * 4, hipMemcpyHostToDevice, st));
    if (a->resume) HIPCHK(hipMemcpyAsync(d_loop, a->loop_count, B * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d_sdd0, 0, B * 8, st));
    HIPCHK(hipMemsetAsync(w_t, 0, B * N * 8, st));
    HIPCHK(hipMemsetAsync(d_ps, 0, B * 8, st));
    HIPCHK(hipMemsetAsync(d_sd0, 0, B * 8, st));
    HIPCHK(hipMemsetAsync(d_t0, 0, B * 8, st));
    HIPCHK(hipMemsetAsync(w_lei, 0, B * 4, st));
    HIPCHK(hipMemsetAsync(w_dtm, 0, B * 8, st));
    PlanParams p{};
    p.B = (int)B; p.N = (int)N; p.D = (int)D; p.K = (int)P + 3; p.cap = (int)cap;
    p.max_iterations = a->max_planning_iterations;
    p.max_initial_velocity_error = a->max_initial_velocity_error;
    p.knots = d_knots; p.delta = d_dl; p.initial_velocity = d_iv; p.start_ns = d_start; p.horizon_ns = d_hor;
    p.path_state = d_state; p.count = d_count; p.h_time = h_t; p.h_s = h_s; p.h_sd = h_sd; p.h_sdd = h_sdd;
    p.h_q = h_q; p.h_qd = h_qd; p.h_qdd = h_qdd; p.planned_to_end = d_pte; p.path_horizon = d_ph;
    p.final_decel_start_ns = d_fds; p.active = d_active; p.old_state = d_old; p.offset = d_off; p.loop = d_loop;
    p.append = d_app; p.windows = d_win; p.status = d_status; p.loop_start_ns = d_loop_start; p.num_active = d_nact;
    p.path_start = d_ps; p.sd_start = d_sd0; p.time_start = d_t0;
    p.w_time = w_t; p.w_s = w_s; p.w_sd = w_sd; p.w_sdd = w_sdd; p.w_q = w_q; p.w_qd = w_qd; p.w_qdd = w_qdd;
    p.w_status = w_st; p.w_lei = w_lei;
    tpamd_joint_batch bt{(int)B, (int)D, (int)N, (int)P, 0, 0, a->constraint_safety};
    tpamd_joint_inputs din{d_knots, d_cp, d_vmax, d_amax, d_ps, d_dl, d_sd0, d_sdd0, d_t0, nullptr};
    tpamd_path_outputs dout{w_t, w_s, w_sd, w_sdd, w_q, w_qd, w_qdd, w_lei, w_dtm, w_st, w_sd2};
    const unsigned gb = (unsigned)((B + 127) / 128);
    bool any_window = false;
    {
      bool full = false;
      for (size_t b = 0; b < B; b++)
        if (act[b] && (size_t)a->history_count[b] + N > cap) full = true;
      if (full) n_active = 0;           // reported as TPAMD_PLAN_MORE below (d_active is untouched)
    }
    while (n_active > 0) {
      // every planner's history must be able to take one more window wherever it connects
      hipLaunchKernelGGL(k_plan_begin, dim3(gb), dim3(128), 0, st, p, e->ws);
      int rc = solve_joint(e, &bt, &din, &dout, st, &p);
      if (rc) return rc;
      any_window = true;
      HIPCHK(hipMemsetAsync(d_nact, 0, 4, st));
      hipLaunchKernelGGL(k_plan_end, dim3(gb), dim3(128), 0, st, p);
      hipLaunchKernelGGL(k_plan_append, dim3((unsigned)((N + 127) / 128), (unsigned)B), dim3(128), 0, st, p);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(&n_active, d_nact, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (n_active > 0) {
        // capacity for the next round: count + N must fit for every looping planner
        std::vector<int> cnt(B), actv(B);
        HIPCHK(hipMemcpy(cnt.data(), d_count, B * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(actv.data(), d_active, B * 4, hipMemcpyDeviceToHost));
        bool full = false;
        for (size_t b = 0; b < B; b++)
          if (actv[b] && (size_t)cnt[b] + N > cap) full = true;
        if (full) break;      // the caller continues with a larger history (status stays 0, planners stay looping)
      }
    }
    if (any_window) {
      const size_t total = B * N * D;
      hipLaunchKernelGGL(k_unpack_records, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (int)B, (int)N,
                         (int)D, e->ws.q12, w_q1, w_q2);
      HIPCHK(hipGetLastError());
    } else {
      HIPCHK(hipMemsetAsync(w_q1, 0, B * N * D * 8, st));
      HIPCHK(hipMemsetAsync(w_q2, 0, B * N * D * 8, st));
    }
    HIPCHK(hipMemcpyAsync(a->path_state, d_state, B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->planned_to_end, d_pte, B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->history_count, d_count, B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->history_time, h_t, B * cap * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->history_s, h_s, B * cap * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->history_sd, h_sd, B * cap * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->history_sdd, h_sdd, B * cap * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->history_q, h_q, B * cap * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->history_qd, h_qd, B * cap * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->history_qdd, h_qdd, B * cap * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->path_horizon, d_ph, B * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->final_decel_start_ns, d_fds, B * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_time, w_t, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_s, w_s, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_sd, w_sd, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_sdd, w_sdd, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_sd2, w_sd2, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_q, w_q, B * N * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_q1, w_q1, B * N * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_q2, w_q2, B * N * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_path_start, d_ps, B * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_sd_start, d_sd0, B * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_time_start, d_t0, B * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_last_extremal_index, w_lei, B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->window_max_time_increment, w_dtm, B * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->status, d_status, B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->windows, d_win, B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(act.data(), d_active, B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->loop_start_ns, d_loop_start, B * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->loop_count, d_loop, B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // planners that are still looping stopped because a history is full
    for (size_t b = 0; b < B; b++) {
      a->looping[b] = act[b];
      if (a->status[b] == 0 && act[b]) a->status[b] = TPAMD_PLAN_MORE;
    }
  }
  return 0;
}

int tpamd_engine_fence(tpamd_engine *e, void *hip_stream) {
  if (!e) return TPAMD_E_INVALID_ARGUMENT;
  if (!e->aux) return 0;
  TPAMD_ON_DEVICE(e);
  for (int k = 0; k < 2; k++) HIPCHK(hipStreamWaitEvent((hipStream_t)hip_stream, e->ev_sweep[k], 0));
  return 0;
}

int tpamd_engine_set_pipelining(tpamd_engine *e, int on) {
  if (!e || on < 0 || on > 2) return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  if (on && !e->aux) {
    // lowest priority: the front stage of the NEXT solve fills what the running sweep leaves
    // free; it must not take workgroup slots from a sweep that is being dispatched
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const char *pr = std::getenv("TPAMD_AUX_PRIORITY");   // A/B: "high", "default"
    int prio = least;
    if (pr && pr[0] == 'h') prio = greatest;
    if (pr && pr[0] == 'd') prio = 0;
    HIPCHK(hipStreamCreateWithPriority(&e->aux, hipStreamNonBlocking, prio));
    for (int k = 0; k < 2; k++) {
      HIPCHK(hipEventCreateWithFlags(&e->ev_front[k], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&e->ev_sweep[k], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&e->ev_call[k], hipEventDisableTiming));
    }
  }
  // The sweep streams exist only once mode 2 is asked for: the runtime multiplexes streams onto a
  // few hardware queues, and one more idle stream was measured to put the engine's stream on the
  // caller's queue -- serialising exactly the two things mode 1 overlaps (0.66 -> 0.93 ms per step).
  if (on == 2)
    for (int k = 0; k < 2; k++)
      if (!e->sweep_stream[k]) HIPCHK(hipStreamCreateWithFlags(&e->sweep_stream[k], hipStreamNonBlocking));
  if (e->pipelining && on != e->pipelining) {   // leave the old mode with nothing in flight
    HIPCHK(hipStreamSynchronize(e->aux));
    for (int k = 0; k < 2; k++)
      if (e->sweep_stream[k]) HIPCHK(hipStreamSynchronize(e->sweep_stream[k]));
  }
  e->pipelining = on;
  return 0;
}

int tpamd_optimize_rows_device(tpamd_engine *e, const tpamd_rows_batch *bt,
                               const tpamd_rows_inputs *in, const tpamd_path_outputs *out,
                               void *hip_stream) {
  if (!e || !bt || !in || !out) return TPAMD_E_INVALID_ARGUMENT;
  const int B = bt->num_paths, N = bt->num_samples, C = bt->num_rows;
  if (B <= 0) return B == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  if (C < 1 || C > 64 || N < 3 || N > 8192) return TPAMD_E_UNSUPPORTED;
  if (!in->a || !in->b || !in->lower || !in->upper || !in->s_start || !in->s_end ||
      !in->sd_start || !in->time_start || !out->time || !out->s || !out->sd || !out->sdd ||
      !out->status)
    return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  hipStream_t st = (hipStream_t)hip_stream;
  SlotGuard slot_guard(e, st);
  int rc = ensure_workspace(e, B, N, 1);
  if (rc) return rc;
  e->last_B = B; e->last_N = N; e->last_time = out->time;
  e->ws.ns = nullptr;
  e->ws.order = nullptr;
  const Workspace &ws = e->ws;
  const int max_loops = bt->max_solver_loops > 0 ? bt->max_solver_loops : 100;
  {
    Timer t(e, st, KI_SETUP);
    hipLaunchKernelGGL(k_setup_rows, dim3((B + 127) / 128), dim3(128), 0, st, B, N, in->s_start,
                       in->s_end, in->sd_start, in->sdd_start, in->time_start, ws);
  }
  return run_rows(e, st, B, N, C, max_loops, in->a, in->b, in->lower, in->upper, out);
}

int tpamd_time_cartesian_paths_device(tpamd_engine *e, const tpamd_cartesian_batch *bt,
                                      const tpamd_cartesian_inputs *in,
                                      const tpamd_path_outputs *out, void *hip_stream) {
  if (!e || !bt || !in || !out) return TPAMD_E_INVALID_ARGUMENT;
  const int B = bt->num_paths, D = bt->num_dofs, N = bt->num_samples;
  if (B <= 0) return B == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  if (D < 1 || D > 16 || N < 3 || N > 8192) return TPAMD_E_UNSUPPORTED;
  if (!in->ik_positions || !in->jacobians || !in->max_velocity || !in->max_acceleration ||
      !in->max_translational_velocity || !in->max_rotational_velocity || !in->path_start ||
      !in->delta || !in->sd_start || !in->time_start || !out->time || !out->s || !out->sd ||
      !out->sdd || !out->status)
    return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  hipStream_t st = (hipStream_t)hip_stream;
  const int C = 2 * D + 2;
  const int max_loops = bt->max_solver_loops > 0 ? bt->max_solver_loops : 0;
  const bool fused = (D == 6 || D == 7) && !e->force_generic;
  SlotGuard slot_guard(e, st);
  int rc = ensure_workspace(e, B, N, C);
  if (rc) return rc;
  e->last_B = B; e->last_N = N; e->last_time = out->time;
  e->ws.ns = nullptr;
  e->ws.order = nullptr;
  e->ws.amax = in->max_acceleration;
  {
    Timer t(e, st, KI_SETUP);
    hipLaunchKernelGGL(k_setup_cartesian, dim3((B + 127) / 128), dim3(128), 0, st, B, N, D,
                       bt->constraint_safety, in->max_velocity, in->max_acceleration,
                       in->max_translational_velocity, in->max_rotational_velocity,
                       in->path_start, in->delta, in->sd_start, in->sdd_start, in->time_start,
                       e->ws);
  }
  if (fused) {
    // rows never materialised: K1 writes records, the joint-structured kernels do the rest
    const Workspace &ws = e->ws;
    {
      Timer t(e, st, KI_SAMPLE_LP);
      const int tpb = 256;
      const size_t lds = (2 * (size_t)C + (2 * (size_t)D + 2) * tpb) * 8;
      const dim3 grid((N + tpb - 1) / tpb, B);
      if (D == 6)
        hipLaunchKernelGGL((k_cartesian_lp<1, 6>), grid, dim3(tpb), lds, st, N, in->ik_positions,
                           in->jacobians, ws);
      else
        hipLaunchKernelGGL((k_cartesian_lp<1, 7>), grid, dim3(tpb), lds, st, N, in->ik_positions,
                           in->jacobians, ws);
    }
    JointSource src;
    src.q12 = ws.q12; src.lim = ws.lim; src.D = D; src.E = 2;
    e->ws.sd2_out = out->sd2;
    {
      Timer t(e, st, KI_SWEEP);
      if (D == 6) launch_sweep_cartesian<6>(st, B, N, max_loops, src, e->ws, out);
      else launch_sweep_cartesian<7>(st, B, N, max_loops, src, e->ws, out);
    }
    if (out->q)
      HIPCHK(hipMemcpyAsync(out->q, in->ik_positions, (size_t)B * N * D * 8,
                            hipMemcpyDeviceToDevice, st));
    HIPCHK(hipGetLastError());
    return 0;
  }
  const size_t nrow = align_up((size_t)B * N * C * 8, 256);
  rc = ensure_rows(e, 4 * nrow);
  if (rc) return rc;
  const Workspace &ws = e->ws;
  double *A = (double *)e->rows_base, *Bm = (double *)((char *)e->rows_base + nrow),
         *LO = (double *)((char *)e->rows_base + 2 * nrow),
         *HI = (double *)((char *)e->rows_base + 3 * nrow);
  {
    Timer t(e, st, KI_SETUP);
    hipLaunchKernelGGL(k_cartesian_rows, dim3((N + 127) / 128, B), dim3(128), 0, st, N, D,
                       bt->constraint_safety, in->ik_positions, in->jacobians, in->max_velocity,
                       in->max_acceleration, in->max_translational_velocity,
                       in->max_rotational_velocity, A, Bm, LO, HI, ws);
  }
  rc = run_rows(e, st, B, N, C, max_loops, A, Bm, LO, HI, out);
  if (rc) return rc;
  if (out->q)
    HIPCHK(hipMemcpyAsync(out->q, in->ik_positions, (size_t)B * N * D * 8,
                          hipMemcpyDeviceToDevice, st));
  if (out->qd || out->qdd) {
    Timer t(e, st, KI_EPILOGUE);
    const size_t total = (size_t)B * N * D;
    hipLaunchKernelGGL(k_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, B, N,
                       D, ws.q12, out->sd, out->sdd, in->max_acceleration, out->status, ws.ns,
                       out->qd, out->qdd);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int tpamd_time_cartesian_paths_host(tpamd_engine *e, const tpamd_cartesian_batch *bt,
                                    const tpamd_cartesian_inputs *in,
                                    const tpamd_path_outputs *out) {
  if (!e || !bt || !in || !out) return TPAMD_E_INVALID_ARGUMENT;
  if (bt->num_paths <= 0) return bt->num_paths == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  if (!in->ik_positions || !in->jacobians || !in->max_velocity || !in->max_acceleration ||
      !in->max_translational_velocity || !in->max_rotational_velocity || !in->path_start ||
      !in->delta || !in->sd_start || !in->time_start || !out->time || !out->s || !out->sd ||
      !out->sdd || !out->status)
    return TPAMD_E_INVALID_ARGUMENT;
  const size_t B = bt->num_paths, D = bt->num_dofs, N = bt->num_samples;
  TPAMD_ON_DEVICE(e);
  for (int pass = 0; pass < 2; pass++) {
    Stage s(pass ? e->stage_base : nullptr);
    double *d_q = s.take<double>(B * N * D), *d_J = s.take<double>(B * N * 6 * D);
    double *d_vmax = s.take<double>(B * D), *d_amax = s.take<double>(B * D);
    double *d_vt = s.take<double>(B), *d_vr = s.take<double>(B);
    double *d_ps = s.take<double>(B), *d_dl = s.take<double>(B), *d_sd0 = s.take<double>(B);
    double *d_sdd0 = s.take<double>(B), *d_t0 = s.take<double>(B);
    double *d_t = s.take<double>(B * N), *d_s = s.take<double>(B * N);
    double *d_sd = s.take<double>(B * N), *d_sdd = s.take<double>(B * N);
    double *d_qd = out->qd ? s.take<double>(B * N * D) : nullptr;
    double *d_qdd = out->qdd ? s.take<double>(B * N * D) : nullptr;
    int32_t *d_lei = s.take<int32_t>(B), *d_st = s.take<int32_t>(B);
    double *d_dtm = s.take<double>(B);
    double *d_sd2 = out->sd2 ? s.take<double>(B * N) : nullptr;
    if (!pass) {
      int rc = ensure_stage(e, s.off);
      if (rc) return rc;
      continue;
    }
    hipStream_t st = nullptr;
    HIPCHK(hipMemcpyAsync(d_q, in->ik_positions, B * N * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_J, in->jacobians, B * N * 6 * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_vmax, in->max_velocity, B * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_amax, in->max_acceleration, B * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_vt, in->max_translational_velocity, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_vr, in->max_rotational_velocity, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_ps, in->path_start, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_dl, in->delta, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_sd0, in->sd_start, B * 8, hipMemcpyHostToDevice, st));
    if (in->sdd_start)
      HIPCHK(hipMemcpyAsync(d_sdd0, in->sdd_start, B * 8, hipMemcpyHostToDevice, st));
    else
      HIPCHK(hipMemsetAsync(d_sdd0, 0, B * 8, st));
    HIPCHK(hipMemcpyAsync(d_t0, in->time_start, B * 8, hipMemcpyHostToDevice, st));
    tpamd_cartesian_inputs din{d_q, d_J, d_vmax, d_amax, d_vt, d_vr, d_ps, d_dl, d_sd0, d_sdd0, d_t0};
    tpamd_path_outputs dout{d_t, d_s, d_sd, d_sdd, nullptr, d_qd, d_qdd, d_lei, d_dtm, d_st, d_sd2};
    int rc = tpamd_time_cartesian_paths_device(e, bt, &din, &dout, st);
    if (rc) return rc;
    if (out->sd2) HIPCHK(hipMemcpyAsync(out->sd2, d_sd2, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->time, d_t, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->s, d_s, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->sd, d_sd, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->sdd, d_sdd, B * N * 8, hipMemcpyDeviceToHost, st));
    if (out->qd) HIPCHK(hipMemcpyAsync(out->qd, d_qd, B * N * D * 8, hipMemcpyDeviceToHost, st));
    if (out->qdd) HIPCHK(hipMemcpyAsync(out->qdd, d_qdd, B * N * D * 8, hipMemcpyDeviceToHost, st));
    if (out->last_extremal_index)
      HIPCHK(hipMemcpyAsync(out->last_extremal_index, d_lei, B * 4, hipMemcpyDeviceToHost, st));
    if (out->max_time_increment)
      HIPCHK(hipMemcpyAsync(out->max_time_increment, d_dtm, B * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->status, d_st, B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (out->q && out->q != in->ik_positions) std::memcpy(out->q, in->ik_positions, B * N * D * 8);
  }
  return 0;
}

// ---- host-buffer convenience paths ---------------------------------------

}  // extern "C"

namespace {
// Device copies of one joint batch's host arrays inside the staging buffer.
struct JointStage {
  double *knots, *cp, *vmax, *amax, *ps, *dl, *sd0, *sdd0, *t0;
  double *t, *s, *sd, *sdd, *q, *qd, *qdd, *dtm, *sd2;
  int32_t *lei, *st, *ns;
};

void carve_joint_stage(Stage &s, const tpamd_joint_batch *bt, const tpamd_joint_inputs *in,
                       const tpamd_path_outputs *out, JointStage *j) {
  const size_t B = bt->num_paths, D = bt->num_dofs, N = bt->num_samples, P = bt->num_points;
  j->knots = s.take<double>(B * (P + 3)); j->cp = s.take<double>(B * P * D);
  j->vmax = s.take<double>(B * D); j->amax = s.take<double>(B * D);
  j->ps = s.take<double>(B); j->dl = s.take<double>(B); j->sd0 = s.take<double>(B);
  j->sdd0 = s.take<double>(B); j->t0 = s.take<double>(B);
  j->t = s.take<double>(B * N); j->s = s.take<double>(B * N);
  j->sd = s.take<double>(B * N); j->sdd = s.take<double>(B * N);
  j->q = out->q ? s.take<double>(B * N * D) : nullptr;
  j->qd = out->qd ? s.take<double>(B * N * D) : nullptr;
  j->qdd = out->qdd ? s.take<double>(B * N * D) : nullptr;
  j->lei = s.take<int32_t>(B); j->st = s.take<int32_t>(B);
  j->dtm = s.take<double>(B);
  j->sd2 = out->sd2 ? s.take<double>(B * N) : nullptr;
  j->ns = in->num_samples_per_path ? s.take<int32_t>(B) : nullptr;
}

int upload_joint_stage(const JointStage &j, const tpamd_joint_batch *bt, const tpamd_joint_inputs *in,
                       hipStream_t st) {
  const size_t B = bt->num_paths, D = bt->num_dofs, P = bt->num_points;
  if (j.ns) HIPCHK(hipMemcpyAsync(j.ns, in->num_samples_per_path, B * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(j.knots, in->knots, B * (P + 3) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(j.cp, in->control_points, B * P * D * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(j.vmax, in->max_velocity, B * D * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(j.amax, in->max_acceleration, B * D * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(j.ps, in->path_start, B * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(j.dl, in->delta, B * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(j.sd0, in->sd_start, B * 8, hipMemcpyHostToDevice, st));
  if (in->sdd_start)
    HIPCHK(hipMemcpyAsync(j.sdd0, in->sdd_start, B * 8, hipMemcpyHostToDevice, st));
  else
    HIPCHK(hipMemsetAsync(j.sdd0, 0, B * 8, st));
  HIPCHK(hipMemcpyAsync(j.t0, in->time_start, B * 8, hipMemcpyHostToDevice, st));
  return 0;
}

int download_joint_stage(const JointStage &j, const tpamd_joint_batch *bt, const tpamd_path_outputs *out,
                         hipStream_t st) {
  const size_t B = bt->num_paths, D = bt->num_dofs, N = bt->num_samples;
  if (out->sd2) HIPCHK(hipMemcpyAsync(out->sd2, j.sd2, B * N * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(out->time, j.t, B * N * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(out->s, j.s, B * N * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(out->sd, j.sd, B * N * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(out->sdd, j.sdd, B * N * 8, hipMemcpyDeviceToHost, st));
  if (out->q) HIPCHK(hipMemcpyAsync(out->q, j.q, B * N * D * 8, hipMemcpyDeviceToHost, st));
  if (out->qd) HIPCHK(hipMemcpyAsync(out->qd, j.qd, B * N * D * 8, hipMemcpyDeviceToHost, st));
  if (out->qdd) HIPCHK(hipMemcpyAsync(out->qdd, j.qdd, B * N * D * 8, hipMemcpyDeviceToHost, st));
  if (out->last_extremal_index)
    HIPCHK(hipMemcpyAsync(out->last_extremal_index, j.lei, B * 4, hipMemcpyDeviceToHost, st));
  if (out->max_time_increment)
    HIPCHK(hipMemcpyAsync(out->max_time_increment, j.dtm, B * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(out->status, j.st, B * 4, hipMemcpyDeviceToHost, st));
  return 0;
}

bool joint_host_args_ok(const tpamd_joint_batch *bt, const tpamd_joint_inputs *in,
                        const tpamd_path_outputs *out) {
  return bt->num_dofs >= 1 && bt->num_samples >= 1 && bt->num_points >= 1 && in->knots &&
         in->control_points && in->max_velocity && in->max_acceleration && in->path_start && in->delta &&
         in->sd_start && in->time_start && out->time && out->s && out->sd && out->sdd && out->status;
}

// Lane k of the engine: created on first use. Lane 0 gets the highest stream priority (it is
// given the group with the longest critical path, see tpamd_time_joint_groups_device).
int ensure_lane(tpamd_engine *e, int k) {
  tpamd_engine::Lane &ln = e->lanes[k];
  if (ln.stream) return 0;
  int least = 0, greatest = 0;
  HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  const char *pr = std::getenv("TPAMD_LANE_PRIORITY");   // A/B: "0" = all lanes at the default priority
  const int prio = (k == 0 && !(pr && pr[0] == '0')) ? greatest : 0;
  HIPCHK(hipStreamCreateWithPriority(&ln.stream, hipStreamNonBlocking, prio));
  HIPCHK(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&ln.front, hipEventDisableTiming));
  if (!e->ev_fork) HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  return 0;
}

// Run `fn` with lane k's workspace in place of the engine's current one.
template <class F>
int with_lane_workspace(tpamd_engine *e, int k, F fn) {
  tpamd_engine::Lane &ln = e->lanes[k];
  std::swap(e->ws_base, ln.base);
  std::swap(e->ws_bytes, ln.bytes);
  const Workspace saved = e->ws;
  e->in_lane = true;
  const int rc = fn(ln.stream);
  e->in_lane = false;
  e->ws = saved;
  std::swap(e->ws_base, ln.base);
  std::swap(e->ws_bytes, ln.bytes);
  return rc;
}
}  // namespace

extern "C" {

int tpamd_time_joint_groups_device(tpamd_engine *e, int num_groups, const tpamd_joint_batch *batches,
                                   const tpamd_joint_inputs *inputs, const tpamd_path_outputs *outputs,
                                   void *hip_stream) {
  if (!e || num_groups < 0 || (num_groups > 0 && (!batches || !inputs || !outputs)))
    return TPAMD_E_INVALID_ARGUMENT;
  if (num_groups == 0) return 0;
  TPAMD_ON_DEVICE(e);
  hipStream_t caller = (hipStream_t)hip_stream;
  // heaviest first: the group whose longest path has the longest sweep (about stride x joints)
  std::vector<int> idx(num_groups);
  for (int g = 0; g < num_groups; g++) idx[g] = g;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
    return (long long)batches[a].num_samples * batches[a].num_dofs >
           (long long)batches[b].num_samples * batches[b].num_dofs;
  });
  if (num_groups == 1 || stream_is_capturing(caller)) {
    // nothing to overlap / a captured stream cannot fork into the engine's lanes: in order
    for (int g : idx) {
      const int rc = solve_joint(e, &batches[g], &inputs[g], &outputs[g], caller, nullptr, false);
      if (rc) return rc;
    }
    return 0;
  }
  const int nl = std::min(num_groups, e->max_lanes);
  for (int k = 0; k < nl; k++) {
    const int rc = ensure_lane(e, k);
    if (rc) return rc;
  }
  // this call uses the lanes' workspaces only, but an unpipelined solve that is still running on
  // the caller's stream owns nothing of theirs: fork the lanes from the caller's position
  HIPCHK(hipEventRecord(e->ev_fork, caller));
  for (int k = 0; k < nl; k++) HIPCHK(hipStreamWaitEvent(e->lanes[k].stream, e->ev_fork, 0));
  // The sampling/LP kernels of the groups run one after another in weight order, each with the
  // whole machine to itself, so that the heaviest group's sweep -- whose longest path is the
  // call's critical path -- starts as early as it can; the sweeps then overlap each other and the
  // later groups' sampling/LP kernels (a lane's front stage waits for the previous lane's).
  // (TPAMD_CHAIN_FRONTS, A/B: 0 no order among the front stages; 1 each behind the previous
  // group's; 2 all behind the heaviest group's; 3 as 2, and no sweep starts before every front
  // stage of the call is done -- for calls with at most one group per lane.)
  int rc_all = 0;
  hipEvent_t prev_front = nullptr;
  const bool two_phase = e->chain_fronts == 3 && num_groups <= nl;
  for (int n = 0; n < num_groups && rc_all == 0; n++) {
    const int g = idx[n];
    if (batches[g].num_paths <= 0) continue;
    const int k = n % nl;
    e->front_wait = e->chain_fronts ? prev_front : nullptr;
    e->front_record = e->lanes[k].front;
    e->phase = two_phase ? 1 : 0;
    rc_all = with_lane_workspace(e, k, [&](hipStream_t st) {
      return solve_joint(e, &batches[g], &inputs[g], &outputs[g], st, nullptr, false);
    });
    if (e->chain_fronts == 1 || prev_front == nullptr) prev_front = e->lanes[k].front;
  }
  e->front_wait = nullptr;
  e->front_record = nullptr;
  if (two_phase) {
    for (int n = 0; n < num_groups; n++)
      if (batches[idx[n]].num_paths > 0) e->back_wait.push_back(e->lanes[n % nl].front);
    e->phase = 2;
    for (int n = 0; n < num_groups && rc_all == 0; n++) {
      const int g = idx[n];
      if (batches[g].num_paths <= 0) continue;
      rc_all = with_lane_workspace(e, n % nl, [&](hipStream_t st) {
        return solve_joint(e, &batches[g], &inputs[g], &outputs[g], st, nullptr, false);
      });
    }
    e->back_wait.clear();
  }
  e->phase = 0;
  // join (also after an error: whatever was launched is ordered before the caller goes on)
  for (int k = 0; k < nl; k++) {
    HIPCHK(hipEventRecord(e->lanes[k].done, e->lanes[k].stream));
    HIPCHK(hipStreamWaitEvent(caller, e->lanes[k].done, 0));
  }
  e->last_B = 0; e->last_N = 0; e->last_time = nullptr;   // no single "last solve" to query
  return rc_all;
}

int tpamd_time_joint_groups_host(tpamd_engine *e, int num_groups, const tpamd_joint_batch *batches,
                                 const tpamd_joint_inputs *inputs, const tpamd_path_outputs *outputs) {
  if (!e || num_groups < 0 || (num_groups > 0 && (!batches || !inputs || !outputs)))
    return TPAMD_E_INVALID_ARGUMENT;
  std::vector<int> live;
  for (int g = 0; g < num_groups; g++) {
    if (batches[g].num_paths < 0) return TPAMD_E_INVALID_ARGUMENT;
    if (batches[g].num_paths == 0) continue;
    if (!joint_host_args_ok(&batches[g], &inputs[g], &outputs[g])) return TPAMD_E_INVALID_ARGUMENT;
    live.push_back(g);
  }
  if (live.empty()) return 0;
  TPAMD_ON_DEVICE(e);
  const size_t G = live.size();
  std::vector<JointStage> js(G);
  for (int pass = 0; pass < 2; pass++) {
    Stage s(pass ? e->stage_base : nullptr);
    for (size_t k = 0; k < G; k++) carve_joint_stage(s, &batches[live[k]], &inputs[live[k]], &outputs[live[k]], &js[k]);
    if (!pass) {
      int rc = ensure_stage(e, s.off);
      if (rc) return rc;
    }
  }
  hipStream_t st = nullptr;
  std::vector<tpamd_joint_batch> bts(G);
  std::vector<tpamd_joint_inputs> dins(G);
  std::vector<tpamd_path_outputs> douts(G);
  for (size_t k = 0; k < G; k++) {
    const JointStage &j = js[k];
    int rc = upload_joint_stage(j, &batches[live[k]], &inputs[live[k]], st);
    if (rc) return rc;
    bts[k] = batches[live[k]];
    dins[k] = tpamd_joint_inputs{j.knots, j.cp, j.vmax, j.amax, j.ps, j.dl, j.sd0, j.sdd0, j.t0, j.ns};
    douts[k] = tpamd_path_outputs{j.t, j.s, j.sd, j.sdd, j.q, j.qd, j.qdd, j.lei, j.dtm, j.st, j.sd2};
  }
  int rc = tpamd_time_joint_groups_device(e, (int)G, bts.data(), dins.data(), douts.data(), st);
  if (rc) return rc;
  for (size_t k = 0; k < G; k++) {
    rc = download_joint_stage(js[k], &batches[live[k]], &outputs[live[k]], st);
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

int tpamd_time_joint_paths_host(tpamd_engine *e, const tpamd_joint_batch *bt,
                                const tpamd_joint_inputs *in, const tpamd_path_outputs *out) {
  if (!e || !bt || !in || !out) return TPAMD_E_INVALID_ARGUMENT;
  if (bt->num_paths <= 0) return bt->num_paths == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  if (!joint_host_args_ok(bt, in, out)) return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  JointStage j{};
  for (int pass = 0; pass < 2; pass++) {
    Stage s(pass ? e->stage_base : nullptr);
    carve_joint_stage(s, bt, in, out, &j);
    if (!pass) {
      int rc = ensure_stage(e, s.off);
      if (rc) return rc;
    }
  }
  hipStream_t st = nullptr;
  int rc = upload_joint_stage(j, bt, in, st);
  if (rc) return rc;
  tpamd_joint_inputs din{j.knots, j.cp, j.vmax, j.amax, j.ps, j.dl, j.sd0, j.sdd0, j.t0, j.ns};
  tpamd_path_outputs dout{j.t, j.s, j.sd, j.sdd, j.q, j.qd, j.qdd, j.lei, j.dtm, j.st, j.sd2};
  // never pipelined: the inputs were just queued on this stream, the outputs are copied back
  // right behind the solve, and the staging buffer is reused by the next _host call
  rc = solve_joint(e, bt, &din, &dout, st, nullptr, /*allow_pipelining=*/false);
  if (rc) return rc;
  rc = download_joint_stage(j, bt, out, st);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

int tpamd_sample_joint_paths_host(tpamd_engine *e, int num_paths, int num_dofs, int num_samples,
                                  int num_points, const double *knots,
                                  const double *control_points, const double *path_start,
                                  const double *delta, double *q, double *q1, double *q2) {
  if (!e || !knots || !control_points || !path_start || !delta || !q || !q1 || !q2)
    return TPAMD_E_INVALID_ARGUMENT;
  if (num_paths <= 0) return num_paths == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  if (num_dofs < 1 || num_samples < 1 || num_points < 3) return TPAMD_E_UNSUPPORTED;
  const size_t B = num_paths, D = num_dofs, N = num_samples, P = num_points;
  TPAMD_ON_DEVICE(e);
  for (int pass = 0; pass < 2; pass++) {
    Stage s(pass ? e->stage_base : nullptr);
    double *d_knots = s.take<double>(B * (P + 3)), *d_cp = s.take<double>(B * P * D);
    double *d_ps = s.take<double>(B), *d_dl = s.take<double>(B);
    double *d_q = s.take<double>(B * N * D), *d_q1 = s.take<double>(B * N * D),
           *d_q2 = s.take<double>(B * N * D);
    if (!pass) {
      int rc = ensure_stage(e, s.off);
      if (rc) return rc;
      continue;
    }
    hipStream_t st = nullptr;
    HIPCHK(hipMemcpyAsync(d_knots, knots, B * (P + 3) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_cp, control_points, B * P * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_ps, path_start, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_dl, delta, B * 8, hipMemcpyHostToDevice, st));
    const size_t lds = (P + 3 + P * D) * 8;
    hipLaunchKernelGGL(k_sample_only, dim3((unsigned)((N + 255) / 256), (unsigned)B), dim3(256),
                       lds, st, (int)N, (int)D, (int)P, d_knots, d_cp, d_ps, d_dl, d_q, d_q1, d_q2);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(q, d_q, B * N * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(q1, d_q1, B * N * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(q2, d_q2, B * N * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return 0;
}

int tpamd_sample_pose_splines_device(tpamd_engine *e, int num_paths, int num_samples, int num_points,
                                     const double *knots, const double *translation_points,
                                     const double *rotation_points, const double *path_start,
                                     const double *delta, double *poses, void *hip_stream) {
  if (!e || !knots || !translation_points || !rotation_points || !path_start || !delta || !poses)
    return TPAMD_E_INVALID_ARGUMENT;
  if (num_paths <= 0) return num_paths == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  if (num_samples < 1 || num_points < 3) return TPAMD_E_UNSUPPORTED;
  const size_t lds = ((size_t)(num_points + 3) + 7 * (size_t)num_points) * 8;
  if (lds > 64 * 1024) return TPAMD_E_UNSUPPORTED;
  TPAMD_ON_DEVICE(e);
  hipLaunchKernelGGL(k_sample_pose_splines, dim3((num_samples + 255) / 256, num_paths), dim3(256), lds,
                     (hipStream_t)hip_stream, num_samples, num_points, knots, translation_points,
                     rotation_points, path_start, delta, poses);
  HIPCHK(hipGetLastError());
  return 0;
}

int tpamd_sample_pose_splines_host(tpamd_engine *e, int num_paths, int num_samples, int num_points,
                                   const double *knots, const double *translation_points,
                                   const double *rotation_points, const double *path_start,
                                   const double *delta, double *poses) {
  if (!e || !knots || !translation_points || !rotation_points || !path_start || !delta || !poses)
    return TPAMD_E_INVALID_ARGUMENT;
  if (num_paths <= 0) return num_paths == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  const size_t B = num_paths, N = num_samples, P = num_points;
  TPAMD_ON_DEVICE(e);
  for (int pass = 0; pass < 2; pass++) {
    Stage s(pass ? e->stage_base : nullptr);
    double *d_k = s.take<double>(B * (P + 3)), *d_t = s.take<double>(B * P * 3), *d_r = s.take<double>(B * P * 4);
    double *d_ps = s.take<double>(B), *d_dl = s.take<double>(B), *d_out = s.take<double>(B * N * 7);
    if (!pass) {
      int rc = ensure_stage(e, s.off);
      if (rc) return rc;
      continue;
    }
    hipStream_t st = nullptr;
    HIPCHK(hipMemcpyAsync(d_k, knots, B * (P + 3) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_t, translation_points, B * P * 3 * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_r, rotation_points, B * P * 4 * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_ps, path_start, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_dl, delta, B * 8, hipMemcpyHostToDevice, st));
    int rc = tpamd_sample_pose_splines_device(e, num_paths, num_samples, num_points, d_k, d_t, d_r, d_ps,
                                              d_dl, d_out, st);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(poses, d_out, B * N * 7 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return 0;
}

int tpamd_optimize_rows_host(tpamd_engine *e, const tpamd_rows_batch *bt,
                             const tpamd_rows_inputs *in, const tpamd_path_outputs *out) {
  if (!e || !bt || !in || !out) return TPAMD_E_INVALID_ARGUMENT;
  const size_t B = bt->num_paths, N = bt->num_samples, C = bt->num_rows;
  if (bt->num_paths <= 0) return bt->num_paths == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  for (int pass = 0; pass < 2; pass++) {
    Stage s(pass ? e->stage_base : nullptr);
    double *d_a = s.take<double>(B * N * C), *d_b = s.take<double>(B * N * C);
    double *d_lo = s.take<double>(B * N * C), *d_hi = s.take<double>(B * N * C);
    double *d_s0 = s.take<double>(B), *d_s1 = s.take<double>(B), *d_sd0 = s.take<double>(B);
    double *d_sdd0 = s.take<double>(B), *d_t0 = s.take<double>(B);
    double *d_t = s.take<double>(B * N), *d_s = s.take<double>(B * N);
    double *d_sd = s.take<double>(B * N), *d_sdd = s.take<double>(B * N);
    int32_t *d_lei = s.take<int32_t>(B), *d_st = s.take<int32_t>(B);
    double *d_dtm = s.take<double>(B);
    double *d_sd2 = out->sd2 ? s.take<double>(B * N) : nullptr;
    if (!pass) {
      int rc = ensure_stage(e, s.off);
      if (rc) return rc;
      continue;
    }
    hipStream_t st = nullptr;
    HIPCHK(hipMemcpyAsync(d_a, in->a, B * N * C * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_b, in->b, B * N * C * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_lo, in->lower, B * N * C * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_hi, in->upper, B * N * C * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_s0, in->s_start, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_s1, in->s_end, B * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_sd0, in->sd_start, B * 8, hipMemcpyHostToDevice, st));
    if (in->sdd_start)
      HIPCHK(hipMemcpyAsync(d_sdd0, in->sdd_start, B * 8, hipMemcpyHostToDevice, st));
    else
      HIPCHK(hipMemsetAsync(d_sdd0, 0, B * 8, st));
    HIPCHK(hipMemcpyAsync(d_t0, in->time_start, B * 8, hipMemcpyHostToDevice, st));
    tpamd_rows_inputs din{d_a, d_b, d_lo, d_hi, d_s0, d_s1, d_sd0, d_sdd0, d_t0};
    tpamd_path_outputs dout{d_t, d_s, d_sd, d_sdd, nullptr, nullptr, nullptr, d_lei, d_dtm, d_st,
                            d_sd2};
    int rc = tpamd_optimize_rows_device(e, bt, &din, &dout, st);
    if (rc) return rc;
    if (out->sd2) HIPCHK(hipMemcpyAsync(out->sd2, d_sd2, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->time, d_t, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->s, d_s, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->sd, d_sd, B * N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->sdd, d_sdd, B * N * 8, hipMemcpyDeviceToHost, st));
    if (out->last_extremal_index)
      HIPCHK(hipMemcpyAsync(out->last_extremal_index, d_lei, B * 4, hipMemcpyDeviceToHost, st));
    if (out->max_time_increment)
      HIPCHK(hipMemcpyAsync(out->max_time_increment, d_dtm, B * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out->status, d_st, B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return 0;
}

int tpamd_find_max_sd2_host(tpamd_engine *e, int num, int C, const double *a, const double *b,
                            const double *lower, const double *upper, double *sd2max,
                            double *sddmax, double *sd2zero) {
  if (!e || num < 0 || !a || !b || !lower || !upper || !sd2max || !sddmax || !sd2zero)
    return TPAMD_E_INVALID_ARGUMENT;
  if (num == 0) return 0;
  if (C < 1 || C > 64) return TPAMD_E_UNSUPPORTED;
  TPAMD_ON_DEVICE(e);
  const size_t n = (size_t)num, nc = n * C;
  for (int pass = 0; pass < 2; pass++) {
    Stage s(pass ? e->stage_base : nullptr);
    double *d_a = s.take<double>(nc), *d_b = s.take<double>(nc), *d_lo = s.take<double>(nc),
           *d_hi = s.take<double>(nc);
    double *d_o0 = s.take<double>(n), *d_o1 = s.take<double>(n), *d_o2 = s.take<double>(n);
    if (!pass) {
      int rc = ensure_stage(e, s.off);
      if (rc) return rc;
      continue;
    }
    hipStream_t st = nullptr;
    HIPCHK(hipMemcpyAsync(d_a, a, nc * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_b, b, nc * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_lo, lower, nc * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_hi, upper, nc * 8, hipMemcpyHostToDevice, st));
    const dim3 grid((num + 63) / 64);
    if (C <= 32)
      hipLaunchKernelGGL((k_lp_only<1>), grid, dim3(64), 0, st, num, C, d_a, d_b, d_lo, d_hi, d_o0,
                         d_o1, d_o2);
    else
      hipLaunchKernelGGL((k_lp_only<2>), grid, dim3(64), 0, st, num, C, d_a, d_b, d_lo, d_hi, d_o0,
                         d_o1, d_o2);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sd2max, d_o0, n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(sddmax, d_o1, n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(sd2zero, d_o2, n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return 0;
}

int tpamd_rebuild_time_device(tpamd_engine *e, int num_shards, int paths_per_shard, int N,
                              size_t shard_stride, const double *sd, const double *ds,
                              const double *time_start, const int32_t *ns, double *time_out,
                              void *hip_stream) {
  if (!e || !sd || !ds || !time_start || !time_out) return TPAMD_E_INVALID_ARGUMENT;
  if (num_shards < 0 || paths_per_shard < 0 || N < 2) return TPAMD_E_INVALID_ARGUMENT;
  const long long total = (long long)num_shards * paths_per_shard;
  if (total == 0) return 0;
  if (total > 0x7fffffffLL) return TPAMD_E_UNSUPPORTED;
  TPAMD_ON_DEVICE(e);
  hipLaunchKernelGGL(k_rebuild_time, dim3((unsigned)total), dim3(64), 0, (hipStream_t)hip_stream, N,
                     paths_per_shard, shard_stride, sd, ds, time_start, ns, time_out);
  HIPCHK(hipGetLastError());
  return 0;
}

int tpamd_query_device(tpamd_engine *e, int B, int N, int K, const double *time, const double *s,
                       const double *sd, const double *sd2, const int32_t *status,
                       const double *t_query, double *os, double *osd, double *osdd, int32_t *ok,
                       void *hip_stream) {
  if (!e || !time || !s || !sd || !t_query || !os || !osd || !osdd) return TPAMD_E_INVALID_ARGUMENT;
  if (B < 0 || K < 0 || N < 2) return TPAMD_E_INVALID_ARGUMENT;
  if (!sd2) {
    // fall back to the copy of sd2_ the engine keeps from its last solve -- only if that is
    // the solve these rows came from
    if (B != e->last_B || N != e->last_N || time != e->last_time) return TPAMD_E_STALE;
    sd2 = e->ws.sd2;
  }
  if (B == 0 || K == 0) return 0;
  TPAMD_ON_DEVICE(e);
  // the engine's copy of sd2_ sits in the current workspace slot: its sweep may be running on an
  // engine stream (mode 2), and the slot's next front stage must not overwrite it under this kernel
  SlotGuard slot_guard(e, (hipStream_t)hip_stream, /*record_after=*/true, /*enable=*/sd2 == e->ws.sd2);
  const size_t total = (size_t)B * K;
  hipLaunchKernelGGL(k_query, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)hip_stream, B, N, K, time, s, sd, sd2, status, t_query, os,
                     osd, osdd, ok);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"

namespace {
int resample_device(tpamd_engine *e, const tpamd_resample_args *a, void *hip_stream, bool skip_mode) {
  if (!e || !a) return TPAMD_E_INVALID_ARGUMENT;
  if (a->num_paths <= 0) return a->num_paths == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  if (a->num_samples < 2 || a->num_dofs < 1 || a->max_out < 1 || !(a->time_step > 0))
    return TPAMD_E_INVALID_ARGUMENT;
  if (!a->time || !a->s || !a->sd || !a->sdd || !a->q || !a->qd || !a->qdd ||
      !a->max_acceleration || !a->start_sec || !a->out_time || !a->out_s || !a->out_sd ||
      !a->out_sdd || !a->out_q || !a->out_qd || !a->out_qdd || !a->count)
    return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  ResampleParams p;
  p.B = a->num_paths; p.N = a->num_samples; p.D = a->num_dofs; p.max_out = a->max_out;
  p.time = a->time; p.s = a->s; p.sd = a->sd; p.sdd = a->sdd;
  p.q = a->q; p.qd = a->qd; p.qdd = a->qdd; p.amax = a->max_acceleration;
  p.start_sec = a->start_sec; p.time_step = a->time_step; p.status = a->status;
  p.ot = a->out_time; p.os = a->out_s; p.osd = a->out_sd; p.osdd = a->out_sdd;
  p.oq = a->out_q; p.oqd = a->out_qd; p.oqdd = a->out_qdd; p.count = a->count;
  if (skip_mode) {
    if (a->num_samples > 32768) return TPAMD_E_UNSUPPORTED;
    p.time_step = 0.95 * a->time_step;   // GetMinTimeDeltaToKeep, path_timing_trajectory.cc:893-900
    hipLaunchKernelGGL(k_resample_skip, dim3(a->num_paths), dim3(64), (size_t)a->num_samples * 4,
                       (hipStream_t)hip_stream, p);
  } else {
    hipLaunchKernelGGL(k_resample, dim3((a->max_out + 255) / 256, a->num_paths), dim3(256), 0,
                       (hipStream_t)hip_stream, p);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int resample_host(tpamd_engine *e, const tpamd_resample_args *a, bool skip_mode) {
  if (!e || !a) return TPAMD_E_INVALID_ARGUMENT;
  if (a->num_paths <= 0) return a->num_paths == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  const size_t B = a->num_paths, N = a->num_samples, D = a->num_dofs, M = a->max_out;
  TPAMD_ON_DEVICE(e);
  for (int pass = 0; pass < 2; pass++) {
    Stage s(pass ? e->stage_base : nullptr);
    double *d_t = s.take<double>(B * N), *d_s = s.take<double>(B * N), *d_sd = s.take<double>(B * N),
           *d_sdd = s.take<double>(B * N);
    double *d_q = s.take<double>(B * N * D), *d_qd = s.take<double>(B * N * D),
           *d_qdd = s.take<double>(B * N * D);
    double *d_am = s.take<double>(B * D), *d_st = s.take<double>(B);
    int32_t *d_status = a->status ? s.take<int32_t>(B) : nullptr;
    double *o_t = s.take<double>(B * M), *o_s = s.take<double>(B * M), *o_sd = s.take<double>(B * M),
           *o_sdd = s.take<double>(B * M);
    double *o_q = s.take<double>(B * M * D), *o_qd = s.take<double>(B * M * D),
           *o_qdd = s.take<double>(B * M * D);
    int32_t *o_cnt = s.take<int32_t>(B);
    if (!pass) {
      int rc = ensure_stage(e, s.off);
      if (rc) return rc;
      continue;
    }
    hipStream_t st = nullptr;
    HIPCHK(hipMemcpyAsync(d_t, a->time, B * N * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_s, a->s, B * N * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_sd, a->sd, B * N * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_sdd, a->sdd, B * N * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_q, a->q, B * N * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_qd, a->qd, B * N * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_qdd, a->qdd, B * N * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_am, a->max_acceleration, B * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_st, a->start_sec, B * 8, hipMemcpyHostToDevice, st));
    if (d_status) HIPCHK(hipMemcpyAsync(d_status, a->status, B * 4, hipMemcpyHostToDevice, st));
    tpamd_resample_args da = *a;
    da.time = d_t; da.s = d_s; da.sd = d_sd; da.sdd = d_sdd; da.q = d_q; da.qd = d_qd; da.qdd = d_qdd;
    da.max_acceleration = d_am; da.start_sec = d_st; da.status = d_status;
    da.out_time = o_t; da.out_s = o_s; da.out_sd = o_sd; da.out_sdd = o_sdd;
    da.out_q = o_q; da.out_qd = o_qd; da.out_qdd = o_qdd; da.count = o_cnt;
    int rc = resample_device(e, &da, st, skip_mode);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(a->out_time, o_t, B * M * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->out_s, o_s, B * M * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->out_sd, o_sd, B * M * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->out_sdd, o_sdd, B * M * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->out_q, o_q, B * M * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->out_qd, o_qd, B * M * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->out_qdd, o_qdd, B * M * D * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a->count, o_cnt, B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return 0;
}
}  // namespace

extern "C" {

int tpamd_resample_uniform_device(tpamd_engine *e, const tpamd_resample_args *a, void *hip_stream) {
  return resample_device(e, a, hip_stream, false);
}
int tpamd_resample_uniform_host(tpamd_engine *e, const tpamd_resample_args *a) {
  return resample_host(e, a, false);
}
int tpamd_resample_skip_device(tpamd_engine *e, const tpamd_resample_args *a, void *hip_stream) {
  return resample_device(e, a, hip_stream, true);
}
int tpamd_resample_skip_host(tpamd_engine *e, const tpamd_resample_args *a) {
  return resample_host(e, a, true);
}

int tpamd_debug_copy_boundary(tpamd_engine *e, int B, int N, double *sd2_max, double *sdd_max,
                              double *sdd_min, double *sd2_zero, uint8_t *type, double *sd2) {
  if (!e || B != e->last_B || N != e->last_N) return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  HIPCHK(hipDeviceSynchronize());
  const size_t n = (size_t)B * N;
  if (sd2_max) HIPCHK(hipMemcpy(sd2_max, e->ws.m, n * 8, hipMemcpyDeviceToHost));
  if (sdd_max) HIPCHK(hipMemcpy(sdd_max, e->ws.X, n * 8, hipMemcpyDeviceToHost));
  if (sdd_min) HIPCHK(hipMemcpy(sdd_min, e->ws.Y, n * 8, hipMemcpyDeviceToHost));
  if (sd2_zero) HIPCHK(hipMemcpy(sd2_zero, e->ws.z0, n * 8, hipMemcpyDeviceToHost));
  if (type) {
    HIPCHK(hipMemcpy(type, e->ws.type, n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) type[i] &= kBndTypeMask;  // drop the engine-internal cache bit
  }
  if (sd2) HIPCHK(hipMemcpy(sd2, e->ws.sd2, n * 8, hipMemcpyDeviceToHost));
  return 0;
}

void tpamd_debug_keep_boundary(tpamd_engine *e, int on) {
  if (e) e->keep_boundary = on != 0;
}

int tpamd_debug_kernel_vgprs(tpamd_engine *e, int which) {
  if (!e || which < 0 || which > 1) return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  hipFuncAttributes attr;
  if (which == 0)
    HIPCHK(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_sample_lp_joint<1, 7>)));
  else
    HIPCHK((sweep_joint_attributes<7, 0>(&attr)));
  return attr.numRegs;
}

int tpamd_debug_copy_diag(tpamd_engine *e, int B, long long *out) {
  if (!e || !out || B != e->last_B) return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, e->ws.diag, (size_t)B * 64 * 8, hipMemcpyDeviceToHost));
  return 0;
}

void tpamd_profile_enable(tpamd_engine *e, int enable) {
  if (e) e->profile = (enable == 2) ? 2 : (enable != 0 ? 1 : 0);
}

void tpamd_profile_reset(tpamd_engine *e) {
  if (!e) return;
  DeviceScope scope(e->device);
  fold_events(e);   // recycles the events
  for (int k = 0; k < KI_COUNT; k++) { e->acc_ms[k] = 0.0; e->acc_n[k] = 0; }
}

double tpamd_profile_mean_ms(tpamd_engine *e, int kernel_index, int *num_launches) {
  if (num_launches) *num_launches = 0;
  if (!e || kernel_index < 0 || kernel_index >= KI_COUNT) return 0.0;
  DeviceScope scope(e->device);
  fold_events(e);
  const int n = e->acc_n[kernel_index];
  if (num_launches) *num_launches = n;
  return n ? e->acc_ms[kernel_index] / n : 0.0;
}

const char *tpamd_profile_kernel_name(int k) { return (k >= 0 && k < KI_COUNT) ? kKernelNames[k] : ""; }
int tpamd_profile_num_kernels(void) { return KI_COUNT; }

}  // extern "C"

// ---------------------------------------------------------------- planner sets
struct tpamd_planner_set {
  tpamd_engine *e = nullptr;
  tpamd_planner_set_config cfg{};
  int cap = 0, tcap = 0;
  // fixed-size state (one allocation), history and trajectory (one allocation each: they grow)
  void *fixed = nullptr, *hist = nullptr, *traj = nullptr;
  size_t fixed_bytes = 0, hist_bytes = 0, traj_bytes = 0;
  PlannerSetState S{};
  PlanParams P{};
  // device arrays that are not part of S / P
  double *d_cp = nullptr, *d_vmax = nullptr, *d_delta = nullptr, *d_iv = nullptr, *d_sdd0 = nullptr;
  double *w_s = nullptr, *w_sd = nullptr, *w_sdd = nullptr, *w_q = nullptr, *w_qd = nullptr, *w_qdd = nullptr,
         *w_dtm = nullptr, *w_time = nullptr;
  int32_t *w_lei = nullptr, *w_st = nullptr;
  int *d_windows = nullptr;
  long long *d_start = nullptr, *d_horizon = nullptr, *d_loop_start = nullptr;
  PlannerSummaryDev *d_summary = nullptr;
  size_t last_h2d = 0, last_d2h = 0;
};

namespace {

size_t carve_history(char *base, size_t B, size_t cap, size_t D, tpamd_planner_set *ps) {
  size_t off = 0;
  auto take = [&](size_t n) { double *p = base ? (double *)(base + off) : nullptr; off = align_up(off + n * 8, 256); return p; };
  double *t = take(B * cap), *s = take(B * cap), *sd = take(B * cap), *sdd = take(B * cap);
  double *q = take(B * cap * D), *qd = take(B * cap * D), *qdd = take(B * cap * D);
  if (ps) {
    ps->S.h_time = t; ps->S.h_s = s; ps->S.h_sd = sd; ps->S.h_sdd = sdd; ps->S.h_q = q; ps->S.h_qd = qd; ps->S.h_qdd = qdd;
  }
  return off;
}
size_t carve_trajectory(char *base, size_t B, size_t tcap, size_t D, tpamd_planner_set *ps) {
  size_t off = 0;
  auto take = [&](size_t n) { double *p = base ? (double *)(base + off) : nullptr; off = align_up(off + n * 8, 256); return p; };
  double *t = take(B * tcap), *s = take(B * tcap), *sd = take(B * tcap), *sdd = take(B * tcap);
  double *q = take(B * tcap * D), *qd = take(B * tcap * D), *qdd = take(B * tcap * D);
  if (ps) {
    ps->S.t_time = t; ps->S.t_s = s; ps->S.t_sd = sd; ps->S.t_sdd = sdd; ps->S.t_q = q; ps->S.t_qd = qd; ps->S.t_qdd = qdd;
  }
  return off;
}

// PlanParams view of the set (the window-loop kernels of tpamd_kernels.h)
void refresh_plan_params(tpamd_planner_set *ps) {
  PlannerSetState &S = ps->S;
  PlanParams &p = ps->P;
  p.B = S.B; p.N = S.N; p.D = S.D; p.K = S.K; p.cap = ps->cap;
  p.max_iterations = ps->cfg.max_planning_iterations;
  p.max_initial_velocity_error = ps->cfg.max_initial_velocity_error;
  p.knots = S.knots; p.delta = ps->d_delta; p.initial_velocity = ps->d_iv;
  p.start_ns = ps->d_start; p.horizon_ns = ps->d_horizon;
  p.path_state = S.path_state; p.count = S.count;
  p.h_time = S.h_time; p.h_s = S.h_s; p.h_sd = S.h_sd; p.h_sdd = S.h_sdd; p.h_q = S.h_q; p.h_qd = S.h_qd; p.h_qdd = S.h_qdd;
  p.planned_to_end = S.planned_to_end; p.path_horizon = S.path_horizon; p.final_decel_start_ns = S.final_decel_start_ns;
  p.active = S.active; p.status = S.status; p.windows = ps->d_windows; p.loop_start_ns = ps->d_loop_start;
  p.num_active = S.num_active;
  p.path_start = S.path_start; p.sd_start = S.path_start_velocity; p.time_start = S.path_time_start;
  p.w_time = ps->w_time; p.w_s = ps->w_s; p.w_sd = ps->w_sd; p.w_sdd = ps->w_sdd; p.w_q = ps->w_q; p.w_qd = ps->w_qd;
  p.w_qdd = ps->w_qdd; p.w_status = ps->w_st; p.w_lei = ps->w_lei;
  S.cap = ps->cap; S.tcap = ps->tcap;
}

// Move the histories (or trajectories) to buffers with a larger per-planner capacity.
int grow_rows(tpamd_planner_set *ps, bool history, int new_cap, hipStream_t st) {
  const size_t B = ps->S.B, D = ps->S.D;
  PlannerSetState old = ps->S;
  void *old_base = history ? ps->hist : ps->traj;
  const int old_cap = history ? ps->cap : ps->tcap;
  const size_t need = history ? carve_history(nullptr, B, new_cap, D, nullptr) : carve_trajectory(nullptr, B, new_cap, D, nullptr);
  void *fresh = nullptr;
  HIPCHK(hipMalloc(&fresh, need));
  if (history) { carve_history((char *)fresh, B, new_cap, D, ps); ps->hist = fresh; ps->hist_bytes = need; ps->cap = new_cap; }
  else { carve_trajectory((char *)fresh, B, new_cap, D, ps); ps->traj = fresh; ps->traj_bytes = need; ps->tcap = new_cap; }
  const int *first = history ? nullptr : old.t_first;
  const int *count = history ? old.count : old.t_count;
  const dim3 g1((old_cap + 255) / 256, (unsigned)B), gD((unsigned)(((size_t)old_cap * D + 255) / 256), (unsigned)B);
  const double *src1[4] = {history ? old.h_time : old.t_time, history ? old.h_s : old.t_s, history ? old.h_sd : old.t_sd,
                           history ? old.h_sdd : old.t_sdd};
  double *dst1[4] = {history ? ps->S.h_time : ps->S.t_time, history ? ps->S.h_s : ps->S.t_s,
                     history ? ps->S.h_sd : ps->S.t_sd, history ? ps->S.h_sdd : ps->S.t_sdd};
  const double *srcD[3] = {history ? old.h_q : old.t_q, history ? old.h_qd : old.t_qd, history ? old.h_qdd : old.t_qdd};
  double *dstD[3] = {history ? ps->S.h_q : ps->S.t_q, history ? ps->S.h_qd : ps->S.t_qd, history ? ps->S.h_qdd : ps->S.t_qdd};
  for (int k = 0; k < 4; k++)
    hipLaunchKernelGGL(k_pset_regrow, g1, dim3(256), 0, st, (int)B, old_cap, new_cap, 1, first, count, src1[k], dst1[k]);
  for (int k = 0; k < 3; k++)
    hipLaunchKernelGGL(k_pset_regrow, gD, dim3(256), 0, st, (int)B, old_cap, new_cap, (int)D, first, count, srcD[k], dstD[k]);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipFree(old_base));
  refresh_plan_params(ps);
  return 0;
}

}  // namespace

extern "C" {

int tpamd_planner_set_create(tpamd_engine *e, const tpamd_planner_set_config *cfg, tpamd_planner_set **out) {
  if (!e || !cfg || !out) return TPAMD_E_INVALID_ARGUMENT;
  *out = nullptr;
  const size_t B = cfg->num_planners, D = cfg->num_dofs, N = cfg->num_samples, P = cfg->num_points;
  if (cfg->num_planners <= 0 || cfg->time_step_ns <= 0) return TPAMD_E_INVALID_ARGUMENT;
  if (D < 1 || D > 16 || N < 3 || N > 8192 || P < 3) return TPAMD_E_UNSUPPORTED;
  if (cfg->sampling_method != 0 && cfg->sampling_method != 1) return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(e);
  tpamd_planner_set *ps = new (std::nothrow) tpamd_planner_set();
  if (!ps) return TPAMD_E_HIP;
  ps->e = e;
  ps->cfg = *cfg;
  ps->cap = cfg->history_capacity > 0 ? std::max<int>(cfg->history_capacity, 2 * (int)N) : 8 * (int)N;
  ps->tcap = cfg->trajectory_capacity > 0 ? cfg->trajectory_capacity : 4096;
  const size_t K = P + 3;
  PlannerSetState &S = ps->S;
  for (int pass = 0; pass < 2; pass++) {
    Stage s(pass ? ps->fixed : nullptr);
    S.knots = s.take<double>(B * K); ps->d_cp = s.take<double>(B * P * D);
    ps->d_vmax = s.take<double>(B * D); S.amax = s.take<double>(B * D);
    ps->d_delta = s.take<double>(B); ps->d_iv = s.take<double>(B * D); ps->d_sdd0 = s.take<double>(B);
    S.path_state = s.take<int>(B); S.has_path = s.take<int>(B); S.count = s.take<int>(B);
    S.initial_plan = s.take<int>(B); S.planned_to_end = s.take<int>(B); S.target_reached = s.take<int>(B);
    S.path_horizon = s.take<double>(B); S.path_start = s.take<double>(B); S.path_start_velocity = s.take<double>(B);
    S.path_time_start = s.take<double>(B);
    S.start_time_ns = s.take<long long>(B); S.end_time_ns = s.take<long long>(B); S.final_decel_start_ns = s.take<long long>(B);
    ps->w_time = s.take<double>(B * N); ps->w_s = s.take<double>(B * N); ps->w_sd = s.take<double>(B * N);
    ps->w_sdd = s.take<double>(B * N); ps->w_q = s.take<double>(B * N * D); ps->w_qd = s.take<double>(B * N * D);
    ps->w_qdd = s.take<double>(B * N * D); ps->w_dtm = s.take<double>(B);
    ps->w_lei = s.take<int32_t>(B); ps->w_st = s.take<int32_t>(B);
    S.t_first = s.take<int>(B); S.t_count = s.take<int>(B);
    ps->d_start = s.take<long long>(B); ps->d_horizon = s.take<long long>(B); ps->d_loop_start = s.take<long long>(B);
    S.mode = s.take<int>(B); S.status = s.take<int>(B); S.active = s.take<int>(B); S.finish = s.take<int>(B);
    S.num_active = s.take<int>(2); S.resample_skip = s.take<int>(B); S.start_sec = s.take<double>(B);
    S.resample_count = s.take<int>(B);
    ps->d_windows = s.take<int>(B);
    ps->P.old_state = s.take<int>(B); ps->P.offset = s.take<int>(B); ps->P.loop = s.take<int>(B); ps->P.append = s.take<int>(B);
    ps->d_summary = s.take<PlannerSummaryDev>(B);
    if (!pass) {
      ps->fixed_bytes = s.off;
      if (hipMalloc(&ps->fixed, s.off) != hipSuccess) { delete ps; return TPAMD_E_HIP; }
    }
  }
  S.w_time = ps->w_time; S.w_lei = ps->w_lei;
  S.start_ns = ps->d_start; S.horizon_ns = ps->d_horizon;
  S.B = (int)B; S.N = (int)N; S.D = (int)D; S.K = (int)K;
  S.method = cfg->sampling_method; S.max_iterations = cfg->max_planning_iterations;
  S.time_step_sec = (double)cfg->time_step_ns / 1e9;                    // path_timing_trajectory.cc:206-211
  S.time_step_duration_ns = (long long)llround(S.time_step_sec * 1e9);  // absl::Seconds(time_step_sec_)
  ps->hist_bytes = carve_history(nullptr, B, ps->cap, D, nullptr);
  ps->traj_bytes = carve_trajectory(nullptr, B, ps->tcap, D, nullptr);
  if (hipMalloc(&ps->hist, ps->hist_bytes) != hipSuccess || hipMalloc(&ps->traj, ps->traj_bytes) != hipSuccess) {
    tpamd_planner_set_destroy(ps);
    return TPAMD_E_HIP;
  }
  carve_history((char *)ps->hist, B, ps->cap, D, ps);
  carve_trajectory((char *)ps->traj, B, ps->tcap, D, ps);
  if (hipMemset(ps->fixed, 0, ps->fixed_bytes) != hipSuccess) { tpamd_planner_set_destroy(ps); return TPAMD_E_HIP; }
  refresh_plan_params(ps);
  // ResetDerived :213-227: planned_to_end_ = true (all other scalars zero)
  std::vector<int> ones(B, 1);
  if (hipMemcpy(S.planned_to_end, ones.data(), B * 4, hipMemcpyHostToDevice) != hipSuccess) {
    tpamd_planner_set_destroy(ps);
    return TPAMD_E_HIP;
  }
  *out = ps;
  return 0;
}

void tpamd_planner_set_destroy(tpamd_planner_set *ps) {
  if (!ps) return;
  DeviceScope scope(ps->e->device);
  if (ps->fixed) (void)hipFree(ps->fixed);
  if (ps->hist) (void)hipFree(ps->hist);
  if (ps->traj) (void)hipFree(ps->traj);
  delete ps;
}

size_t tpamd_planner_set_device_bytes(const tpamd_planner_set *ps) {
  return ps ? ps->fixed_bytes + ps->hist_bytes + ps->traj_bytes : 0;
}

void tpamd_planner_set_last_plan_bytes(const tpamd_planner_set *ps, size_t *h2d, size_t *d2h) {
  if (h2d) *h2d = ps ? ps->last_h2d : 0;
  if (d2h) *d2h = ps ? ps->last_d2h : 0;
}

int tpamd_planner_set_upload_paths(tpamd_planner_set *ps, int count, const int32_t *ids, const double *knots,
                                   const double *cps, const double *vmax, const double *amax, const double *delta,
                                   const double *iv, const int32_t *path_state) {
  if (!ps || count < 0 || !knots || !cps || !vmax || !amax || !delta || !path_state) return TPAMD_E_INVALID_ARGUMENT;
  const size_t B = ps->S.B, D = ps->S.D, K = ps->S.K, P = K - 3;
  if ((size_t)count > B) return TPAMD_E_INVALID_ARGUMENT;
  TPAMD_ON_DEVICE(ps->e);
  const int one = 1;
  std::vector<double> zero(D, 0.0);
  if (!ids && count > 0) {     // planners 0 .. count-1: one copy per array
    for (int k = 0; k < count; k++)
      if (path_state[k] != 1 && path_state[k] != 2) return TPAMD_E_INVALID_ARGUMENT;
    const size_t n = (size_t)count;
    std::vector<int> ones(n, 1);
    std::vector<double> zeros(iv ? 0 : n * D, 0.0);
    HIPCHK(hipMemcpyAsync((double *)ps->S.knots, knots, n * K * 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->d_cp, cps, n * P * D * 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->d_vmax, vmax, n * D * 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync((double *)ps->S.amax, amax, n * D * 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->d_delta, delta, n * 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->d_iv, iv ? iv : zeros.data(), n * D * 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->S.path_state, path_state, n * 4, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->S.has_path, ones.data(), n * 4, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
  }
  for (int k = 0; k < count; k++) {
    const size_t b = ids ? (size_t)ids[k] : (size_t)k;
    if (b >= B || (path_state[k] != 1 && path_state[k] != 2)) return TPAMD_E_INVALID_ARGUMENT;
    HIPCHK(hipMemcpyAsync((double *)ps->S.knots + b * K, knots + (size_t)k * K, K * 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->d_cp + b * P * D, cps + (size_t)k * P * D, P * D * 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->d_vmax + b * D, vmax + (size_t)k * D, D * 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync((double *)ps->S.amax + b * D, amax + (size_t)k * D, D * 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->d_delta + b, delta + k, 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->d_iv + b * D, iv ? iv + (size_t)k * D : zero.data(), D * 8, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->S.path_state + b, path_state + k, 4, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(ps->S.has_path + b, &one, 4, hipMemcpyHostToDevice, nullptr));
  }
  HIPCHK(hipStreamSynchronize(nullptr));
  return 0;
}

int tpamd_planner_set_reset(tpamd_planner_set *ps, int count, const int32_t *ids) {
  if (!ps || count < 0) return TPAMD_E_INVALID_ARGUMENT;
  const size_t B = ps->S.B;
  TPAMD_ON_DEVICE(ps->e);
  PlannerSetState &S = ps->S;
  const int one = 1;
  const int n = ids ? count : (int)B;
  for (int k = 0; k < n; k++) {
    const size_t b = ids ? (size_t)ids[k] : (size_t)k;
    if (b >= B) return TPAMD_E_INVALID_ARGUMENT;
    for (int *a : {S.path_state, S.has_path, S.count, S.initial_plan, S.target_reached, S.t_first, S.t_count})
      HIPCHK(hipMemsetAsync(a + b, 0, 4, nullptr));
    for (double *a : {S.path_horizon, S.path_start, S.path_start_velocity, S.path_time_start})
      HIPCHK(hipMemsetAsync(a + b, 0, 8, nullptr));
    for (long long *a : {S.start_time_ns, S.end_time_ns, S.final_decel_start_ns})
      HIPCHK(hipMemsetAsync(a + b, 0, 8, nullptr));
    HIPCHK(hipMemcpyAsync(S.planned_to_end + b, &one, 4, hipMemcpyHostToDevice, nullptr));
  }
  HIPCHK(hipStreamSynchronize(nullptr));
  return 0;
}

int tpamd_planner_set_plan(tpamd_planner_set *ps, const int64_t *start_ns, const int64_t *horizon_ns,
                           tpamd_planner_summary *summary) {
  if (!ps || !start_ns || !horizon_ns) return TPAMD_E_INVALID_ARGUMENT;
  static_assert(sizeof(tpamd_planner_summary) == sizeof(PlannerSummaryDev), "summary layouts must agree");
  tpamd_engine *e = ps->e;
  TPAMD_ON_DEVICE(e);
  PlannerSetState &S = ps->S;
  const size_t B = S.B, N = S.N, D = S.D, P = S.K - 3;
  hipStream_t st = nullptr;
  ps->last_h2d = ps->last_d2h = 0;
  HIPCHK(hipMemcpyAsync(ps->d_start, start_ns, B * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ps->d_horizon, horizon_ns, B * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ps->d_loop_start, start_ns, B * 8, hipMemcpyHostToDevice, st));   // :630
  ps->last_h2d += 3 * B * 8;
  for (int *z : {ps->P.old_state, ps->P.offset, ps->P.loop, ps->P.append, ps->d_windows})
    HIPCHK(hipMemsetAsync(z, 0, B * 4, st));
  HIPCHK(hipMemsetAsync(S.num_active, 0, 8, st));
  const unsigned gb = (unsigned)((B + 127) / 128);
  hipLaunchKernelGGL(k_pset_prologue, dim3(gb), dim3(128), 0, st, S);
  hipLaunchKernelGGL(k_pset_check_capacity, dim3(gb), dim3(128), 0, st, S);
  int na[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(na, S.num_active, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  ps->last_d2h += 8;
  tpamd_joint_batch bt{(int)B, (int)D, (int)N, (int)P, 0, 0, ps->cfg.constraint_safety};
  for (int guard = 0; na[0] > 0; guard++) {
    if (guard > 100000) return TPAMD_E_UNSUPPORTED;
    if (na[1]) {          // a looping planner's history is full: double the histories, then go on
      const int rc = grow_rows(ps, /*history=*/true, 2 * ps->cap, st);
      if (rc) return rc;
    }
    tpamd_joint_inputs din{S.knots, ps->d_cp, ps->d_vmax, S.amax, S.path_start, ps->d_delta, S.path_start_velocity,
                           ps->d_sdd0, S.path_time_start, nullptr};
    tpamd_path_outputs dout{ps->w_time, ps->w_s, ps->w_sd, ps->w_sdd, ps->w_q, ps->w_qd, ps->w_qdd, ps->w_lei,
                            ps->w_dtm, ps->w_st, nullptr};
    hipLaunchKernelGGL(k_plan_begin, dim3(gb), dim3(128), 0, st, ps->P, e->ws);
    const int rc = solve_joint(e, &bt, &din, &dout, st, &ps->P);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(S.num_active, 0, 8, st));
    hipLaunchKernelGGL(k_plan_end, dim3(gb), dim3(128), 0, st, ps->P);
    hipLaunchKernelGGL(k_plan_append, dim3((unsigned)((N + 127) / 128), (unsigned)B), dim3(128), 0, st, ps->P);
    hipLaunchKernelGGL(k_pset_check_capacity, dim3(gb), dim3(128), 0, st, S);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(na, S.num_active, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    ps->last_d2h += 8;
  }
  // ResampleTrajectory(start) :660 and the bookkeeping of :662-684; repeated with larger trajectory
  // buffers if a planner's resampled trajectory does not fit
  for (int attempt = 0; attempt < 8; attempt++) {
    hipLaunchKernelGGL(k_pset_before_resample, dim3(gb), dim3(128), 0, st, S);
    ResampleParams rp;
    rp.B = (int)B; rp.N = ps->cap; rp.D = (int)D; rp.max_out = ps->tcap;
    rp.time = S.h_time; rp.s = S.h_s; rp.sd = S.h_sd; rp.sdd = S.h_sdd; rp.q = S.h_q; rp.qd = S.h_qd; rp.qdd = S.h_qdd;
    rp.amax = S.amax; rp.start_sec = S.start_sec; rp.status = S.resample_skip; rp.ns = S.count;
    rp.ot = S.t_time; rp.os = S.t_s; rp.osd = S.t_sd; rp.osdd = S.t_sdd; rp.oq = S.t_q; rp.oqd = S.t_qd; rp.oqdd = S.t_qdd;
    rp.count = S.resample_count;
    if (S.method == 0) {
      rp.time_step = S.time_step_sec;
      hipLaunchKernelGGL(k_resample, dim3((ps->tcap + 255) / 256, (unsigned)B), dim3(256), 0, st, rp);
    } else {
      if (ps->cap > 32768) return TPAMD_E_UNSUPPORTED;
      rp.time_step = 0.95 * S.time_step_sec;            // GetMinTimeDeltaToKeep :893-900
      hipLaunchKernelGGL(k_resample_skip, dim3((unsigned)B), dim3(64), (size_t)ps->cap * 4, st, rp);
    }
    HIPCHK(hipMemsetAsync(S.num_active, 0, 8, st));
    hipLaunchKernelGGL(k_pset_epilogue, dim3(gb), dim3(128), 0, st, S);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(na, S.num_active, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    ps->last_d2h += 8;
    if (na[1] <= ps->tcap) break;
    const int rc = grow_rows(ps, /*history=*/false, std::max(2 * ps->tcap, na[1] + 64), st);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_pset_summary, dim3(gb), dim3(128), 0, st, S, ps->d_windows, ps->d_summary);
  HIPCHK(hipGetLastError());
  if (summary) {
    HIPCHK(hipMemcpyAsync(summary, ps->d_summary, B * sizeof(PlannerSummaryDev), hipMemcpyDeviceToHost, st));
    ps->last_d2h += B * sizeof(PlannerSummaryDev);
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

int tpamd_planner_set_download_trajectory(tpamd_planner_set *ps, int planner, int first, int count, double *time,
                                          double *s, double *sd, double *sdd, double *q, double *qd, double *qdd) {
  if (!ps || planner < 0 || planner >= ps->S.B || first < 0 || count < 0) return TPAMD_E_INVALID_ARGUMENT;
  if (count == 0) return 0;
  TPAMD_ON_DEVICE(ps->e);
  const PlannerSetState &S = ps->S;
  int fc[2];
  HIPCHK(hipMemcpy(&fc[0], S.t_first + planner, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&fc[1], S.t_count + planner, 4, hipMemcpyDeviceToHost));
  if (first + count > fc[1]) return TPAMD_E_INVALID_ARGUMENT;
  const size_t o = (size_t)planner * ps->tcap + fc[0] + first, D = S.D, n = count;
  if (time) HIPCHK(hipMemcpy(time, S.t_time + o, n * 8, hipMemcpyDeviceToHost));
  if (s) HIPCHK(hipMemcpy(s, S.t_s + o, n * 8, hipMemcpyDeviceToHost));
  if (sd) HIPCHK(hipMemcpy(sd, S.t_sd + o, n * 8, hipMemcpyDeviceToHost));
  if (sdd) HIPCHK(hipMemcpy(sdd, S.t_sdd + o, n * 8, hipMemcpyDeviceToHost));
  if (q) HIPCHK(hipMemcpy(q, S.t_q + o * D, n * D * 8, hipMemcpyDeviceToHost));
  if (qd) HIPCHK(hipMemcpy(qd, S.t_qd + o * D, n * D * 8, hipMemcpyDeviceToHost));
  if (qdd) HIPCHK(hipMemcpy(qdd, S.t_qdd + o * D, n * D * 8, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"

// tpamd_multi.cc -- several devices from one process (include/tpamd_multi.h): path blocks per
// device, one host thread per device, one RCCL gather to the root. Host code only; the kernels
// are libtpamd.so's.
#include "../../include/tpamd_multi.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace {

#define HIPCHK(expr)                                                                    \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      std::fprintf(stderr, "[tpamd multi] HIP error %s at %s:%d: %s\n", #expr, __FILE__, \
                   __LINE__, hipGetErrorString(e_));                                    \
      return TPAMD_E_HIP;                                                               \
    }                                                                                   \
  } while (0)
#define NCCLCHK(expr)                                                                    \
  do {                                                                                   \
    ncclResult_t r_ = (expr);                                                            \
    if (r_ != ncclSuccess) {                                                             \
      std::fprintf(stderr, "[tpamd multi] RCCL error %s at %s:%d: %s\n", #expr, __FILE__, \
                   __LINE__, ncclGetErrorString(r_));                                    \
      return TPAMD_E_RCCL;                                                               \
    }                                                                                    \
  } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Bump {   // sizing pass with base == null, then the real one
  char *base;
  size_t off = 0;
  explicit Bump(void *b) : base((char *)b) {}
  template <typename T>
  T *take(size_t count) {
    T *p = base ? (T *)(base + off) : nullptr;
    off = align_up(off + count * sizeof(T), 256);
    return p;
  }
};

}  // namespace

struct tpamd_multi {
  int n = 0;
  bool rccl = false;
  std::vector<int> dev;
  std::vector<tpamd_engine *> eng;
  std::vector<hipStream_t> st;
  std::vector<ncclComm_t> comm;
  std::vector<void *> buf;        // per device: inputs and (non-root) outputs of its block
  std::vector<size_t> buf_bytes;
  void *root_aux = nullptr;       // root: ds, time_start of all paths (compact payload)
  size_t root_aux_bytes = 0;
};

namespace {

int ensure_buffer(tpamd_multi *m, int k, size_t need) {
  if (need <= m->buf_bytes[k]) return 0;
  if (m->buf[k]) HIPCHK(hipFree(m->buf[k]));
  m->buf[k] = nullptr;
  m->buf_bytes[k] = 0;
  HIPCHK(hipMalloc(&m->buf[k], need));
  m->buf_bytes[k] = need;
  return 0;
}

// What device k holds for its block while a gather is pending.
struct Block {
  int lo = 0, count = 0;
  double *knots, *cp, *vmax, *amax, *ps, *dl, *sd0, *sdd0, *t0;
  int32_t *ns;
  double *t, *s, *sd, *sdd, *q, *qd, *qdd, *dtm;
  int32_t *lei, *st;
  int rc = 0;
};

void carve_block(Bump &b, size_t Bk, size_t D, size_t N, size_t P, bool ragged, bool own_outputs,
                 bool want_q, bool want_derivs, Block *blk) {
  blk->knots = b.take<double>(Bk * (P + 3)); blk->cp = b.take<double>(Bk * P * D);
  blk->vmax = b.take<double>(Bk * D); blk->amax = b.take<double>(Bk * D);
  blk->ps = b.take<double>(Bk); blk->dl = b.take<double>(Bk); blk->sd0 = b.take<double>(Bk);
  blk->sdd0 = b.take<double>(Bk); blk->t0 = b.take<double>(Bk);
  blk->ns = ragged ? b.take<int32_t>(Bk) : nullptr;
  blk->s = b.take<double>(Bk * N);
  blk->dtm = b.take<double>(Bk);
  blk->qd = want_derivs ? b.take<double>(Bk * N * D) : nullptr;
  blk->qdd = want_derivs ? b.take<double>(Bk * N * D) : nullptr;
  if (own_outputs) {
    blk->t = b.take<double>(Bk * N); blk->sd = b.take<double>(Bk * N); blk->sdd = b.take<double>(Bk * N);
    blk->q = want_q ? b.take<double>(Bk * N * D) : nullptr;
    blk->lei = b.take<int32_t>(Bk); blk->st = b.take<int32_t>(Bk);
  }
}

}  // namespace

extern "C" {

size_t tpamd_gather_bytes_per_path(int payload, int N, int D) {
  if (N < 0 || D < 0) return 0;
  switch (payload) {
    case TPAMD_GATHER_COMPACT: return 16 * (size_t)N + 16;
    case TPAMD_GATHER_PROFILE: return 24 * (size_t)N;
    case TPAMD_GATHER_FULL: return 24 * (size_t)N + 8 * (size_t)N * D;
    default: return 0;
  }
}

int tpamd_multi_create(int num_devices, const int *device_ordinals, int force_rccl, tpamd_multi **out) {
  if (!out) return TPAMD_E_INVALID_ARGUMENT;
  *out = nullptr;
  const int visible = tpamd_device_count();
  if (visible <= 0) {
    std::fprintf(stderr, "[tpamd multi] no HIP device available: there is no CPU fallback\n");
    return TPAMD_E_NO_DEVICE;
  }
  if (num_devices <= 0 || num_devices > visible) return TPAMD_E_INVALID_ARGUMENT;
  tpamd_multi *m = new tpamd_multi();
  m->n = num_devices;
  m->dev.resize(num_devices);
  for (int k = 0; k < num_devices; k++) {
    m->dev[k] = device_ordinals ? device_ordinals[k] : k;
    if (m->dev[k] < 0 || m->dev[k] >= visible) { delete m; return TPAMD_E_INVALID_ARGUMENT; }
    for (int j = 0; j < k; j++)
      if (m->dev[j] == m->dev[k]) { delete m; return TPAMD_E_INVALID_ARGUMENT; }   // RCCL: one rank per device
  }
  m->eng.assign(num_devices, nullptr);
  m->st.assign(num_devices, nullptr);
  m->buf.assign(num_devices, nullptr);
  m->buf_bytes.assign(num_devices, 0);
  int prev = 0;
  (void)hipGetDevice(&prev);
  int rc = 0;
  for (int k = 0; k < num_devices && rc == 0; k++) {
    rc = tpamd_engine_create(m->dev[k], &m->eng[k]);
    if (rc == 0 && (hipSetDevice(m->dev[k]) != hipSuccess ||
                    hipStreamCreateWithFlags(&m->st[k], hipStreamNonBlocking) != hipSuccess))
      rc = TPAMD_E_HIP;
  }
  if (rc == 0 && (num_devices > 1 || force_rccl)) {
    m->comm.assign(num_devices, nullptr);
    if (ncclCommInitAll(m->comm.data(), num_devices, m->dev.data()) != ncclSuccess) {
      std::fprintf(stderr, "[tpamd multi] ncclCommInitAll failed for %d device(s)\n", num_devices);
      m->comm.clear();
      rc = TPAMD_E_RCCL;
    } else {
      m->rccl = true;
    }
  }
  (void)hipSetDevice(prev);
  if (rc != 0) {
    tpamd_multi_destroy(m);
    return rc;
  }
  *out = m;
  return 0;
}

void tpamd_multi_destroy(tpamd_multi *m) {
  if (!m) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (auto c : m->comm)
    if (c) (void)ncclCommDestroy(c);
  for (int k = 0; k < m->n; k++) {
    (void)hipSetDevice(m->dev[k]);
    if (m->st[k]) (void)hipStreamDestroy(m->st[k]);
    if (m->buf[k]) (void)hipFree(m->buf[k]);
    if (k == 0 && m->root_aux) (void)hipFree(m->root_aux);
    if (m->eng[k]) tpamd_engine_destroy(m->eng[k]);
  }
  (void)hipSetDevice(prev);
  delete m;
}

int tpamd_multi_num_devices(const tpamd_multi *m) { return m ? m->n : 0; }
int tpamd_multi_device(const tpamd_multi *m, int k) { return (m && k >= 0 && k < m->n) ? m->dev[k] : -1; }
tpamd_engine *tpamd_multi_engine(tpamd_multi *m, int k) { return (m && k >= 0 && k < m->n) ? m->eng[k] : nullptr; }
int tpamd_multi_uses_rccl(const tpamd_multi *m) { return (m && m->rccl) ? 1 : 0; }

int tpamd_multi_time_joint_paths_host(tpamd_multi *m, const tpamd_joint_batch *bt,
                                      const tpamd_joint_inputs *in, const int32_t *shard_begin,
                                      const tpamd_path_outputs *host_out, int payload,
                                      const tpamd_path_outputs *root_out) {
  if (!m || !bt || !in) return TPAMD_E_INVALID_ARGUMENT;
  const int B = bt->num_paths;
  if (B <= 0) return B == 0 ? 0 : TPAMD_E_INVALID_ARGUMENT;
  if (payload < TPAMD_GATHER_COMPACT || payload > TPAMD_GATHER_FULL) return TPAMD_E_INVALID_ARGUMENT;
  if (!in->knots || !in->control_points || !in->max_velocity || !in->max_acceleration || !in->path_start ||
      !in->delta || !in->sd_start || !in->time_start)
    return TPAMD_E_INVALID_ARGUMENT;
  if (host_out && (!host_out->time || !host_out->s || !host_out->sd || !host_out->sdd || !host_out->status))
    return TPAMD_E_INVALID_ARGUMENT;
  if (root_out) {
    if (!root_out->sd || !root_out->sdd || !root_out->time) return TPAMD_E_INVALID_ARGUMENT;
    if (payload == TPAMD_GATHER_FULL && !root_out->q) return TPAMD_E_INVALID_ARGUMENT;
    if (m->n > 1 && !m->rccl) return TPAMD_E_RCCL;
  }
  const size_t D = bt->num_dofs, N = bt->num_samples, P = bt->num_points;
  std::vector<int32_t> begin(m->n + 1);
  if (shard_begin) {
    for (int k = 0; k <= m->n; k++) begin[k] = shard_begin[k];
    if (begin[0] != 0 || begin[m->n] != B) return TPAMD_E_INVALID_ARGUMENT;
    for (int k = 0; k < m->n; k++)
      if (begin[k + 1] < begin[k]) return TPAMD_E_INVALID_ARGUMENT;
  } else {
    tpamd_shard_bounds(B, m->n, begin.data());
  }
  bool equal_blocks = true;
  for (int k = 1; k < m->n; k++) equal_blocks = equal_blocks && (begin[k + 1] - begin[k] == begin[1] - begin[0]);
  const bool ragged = in->num_samples_per_path != nullptr;
  const bool want_q = (host_out && host_out->q) || (root_out && payload == TPAMD_GATHER_FULL);
  const bool want_derivs = host_out && (host_out->qd || host_out->qdd);
  std::vector<Block> blocks(m->n);

  // ---- the solve: one host thread per device, no collective --------------------------------
  auto work = [&](int k) -> int {
    Block &blk = blocks[k];
    blk.lo = begin[k];
    blk.count = begin[k + 1] - begin[k];
    if (blk.count == 0) return 0;
    const size_t lo = blk.lo, Bk = blk.count;
    tpamd_joint_batch bk = *bt;
    bk.num_paths = (int)Bk;
    tpamd_joint_inputs ik{in->knots + lo * (P + 3), in->control_points + lo * P * D,
                          in->max_velocity + lo * D, in->max_acceleration + lo * D,
                          in->path_start + lo, in->delta + lo, in->sd_start + lo,
                          in->sdd_start ? in->sdd_start + lo : nullptr, in->time_start + lo,
                          ragged ? in->num_samples_per_path + lo : nullptr};
    if (!root_out) {
      // results go straight to the host: the engine's own host-buffer path
      tpamd_path_outputs ok{host_out->time + lo * N, host_out->s + lo * N, host_out->sd + lo * N,
                            host_out->sdd + lo * N, host_out->q ? host_out->q + lo * N * D : nullptr,
                            host_out->qd ? host_out->qd + lo * N * D : nullptr,
                            host_out->qdd ? host_out->qdd + lo * N * D : nullptr,
                            host_out->last_extremal_index ? host_out->last_extremal_index + lo : nullptr,
                            host_out->max_time_increment ? host_out->max_time_increment + lo : nullptr,
                            host_out->status + lo, host_out->sd2 ? host_out->sd2 + lo * N : nullptr};
      return tpamd_time_joint_paths_host(m->eng[k], &bk, &ik, &ok);
    }
    // results stay on the device for the gather: the root's block is written in place
    HIPCHK(hipSetDevice(m->dev[k]));
    const bool own = k != 0;
    for (int pass = 0; pass < 2; pass++) {
      Bump b(pass ? m->buf[k] : nullptr);
      carve_block(b, Bk, D, N, P, ragged, /*own_outputs=*/true, want_q, want_derivs, &blk);
      if (!pass) {
        const int rc = ensure_buffer(m, k, b.off);
        if (rc) return rc;
      }
    }
    if (!own) {   // root: the gathered arrays are the outputs of its own block
      blk.t = root_out->time + lo * N; blk.sd = root_out->sd + lo * N; blk.sdd = root_out->sdd + lo * N;
      if (payload == TPAMD_GATHER_FULL) blk.q = root_out->q + lo * N * D;
      if (root_out->status) blk.st = root_out->status + lo;
      if (root_out->last_extremal_index) blk.lei = root_out->last_extremal_index + lo;
    }
    hipStream_t st = m->st[k];
    HIPCHK(hipMemcpyAsync(blk.knots, ik.knots, Bk * (P + 3) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(blk.cp, ik.control_points, Bk * P * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(blk.vmax, ik.max_velocity, Bk * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(blk.amax, ik.max_acceleration, Bk * D * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(blk.ps, ik.path_start, Bk * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(blk.dl, ik.delta, Bk * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(blk.sd0, ik.sd_start, Bk * 8, hipMemcpyHostToDevice, st));
    if (ik.sdd_start) HIPCHK(hipMemcpyAsync(blk.sdd0, ik.sdd_start, Bk * 8, hipMemcpyHostToDevice, st));
    else HIPCHK(hipMemsetAsync(blk.sdd0, 0, Bk * 8, st));
    HIPCHK(hipMemcpyAsync(blk.t0, ik.time_start, Bk * 8, hipMemcpyHostToDevice, st));
    if (ragged) HIPCHK(hipMemcpyAsync(blk.ns, ik.num_samples_per_path, Bk * 4, hipMemcpyHostToDevice, st));
    tpamd_joint_inputs din{blk.knots, blk.cp, blk.vmax, blk.amax, blk.ps, blk.dl, blk.sd0, blk.sdd0, blk.t0, blk.ns};
    tpamd_path_outputs dout{blk.t, blk.s, blk.sd, blk.sdd, blk.q, blk.qd, blk.qdd, blk.lei, blk.dtm, blk.st, nullptr};
    const int rc = tpamd_time_joint_paths_device(m->eng[k], &bk, &din, &dout, st);
    if (rc) return rc;
    if (host_out) {
      HIPCHK(hipMemcpyAsync(host_out->time + lo * N, blk.t, Bk * N * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(host_out->s + lo * N, blk.s, Bk * N * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(host_out->sd + lo * N, blk.sd, Bk * N * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(host_out->sdd + lo * N, blk.sdd, Bk * N * 8, hipMemcpyDeviceToHost, st));
      if (host_out->q) HIPCHK(hipMemcpyAsync(host_out->q + lo * N * D, blk.q, Bk * N * D * 8, hipMemcpyDeviceToHost, st));
      if (host_out->qd) HIPCHK(hipMemcpyAsync(host_out->qd + lo * N * D, blk.qd, Bk * N * D * 8, hipMemcpyDeviceToHost, st));
      if (host_out->qdd) HIPCHK(hipMemcpyAsync(host_out->qdd + lo * N * D, blk.qdd, Bk * N * D * 8, hipMemcpyDeviceToHost, st));
      if (host_out->last_extremal_index)
        HIPCHK(hipMemcpyAsync(host_out->last_extremal_index + lo, blk.lei, Bk * 4, hipMemcpyDeviceToHost, st));
      if (host_out->max_time_increment)
        HIPCHK(hipMemcpyAsync(host_out->max_time_increment + lo, blk.dtm, Bk * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(host_out->status + lo, blk.st, Bk * 4, hipMemcpyDeviceToHost, st));
    }
    if (!m->rccl || m->n == 1) HIPCHK(hipStreamSynchronize(st));   // (else the gather follows on the same stream)
    return 0;
  };
  {
    std::vector<std::thread> threads;
    for (int k = 1; k < m->n; k++) threads.emplace_back([&, k] { blocks[k].rc = work(k); });
    blocks[0].rc = work(0);
    for (auto &t : threads) t.join();
    for (int k = 0; k < m->n; k++)
      if (blocks[k].rc) return blocks[k].rc;
  }
  if (!root_out) return 0;

  // ---- ONE gather to the root: every array of the payload inside one RCCL group -------------
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (m->rccl) {
    auto gather = [&](auto get_ptr, void *root_base, size_t per_path, ncclDataType_t type, size_t elem) -> int {
      // get_ptr(k): the block's array on device k; root_base: [B]-shaped array on the root
      for (int k = 0; k < m->n; k++) {
        const size_t cnt = (size_t)blocks[k].count * per_path;
        if (equal_blocks) {
          // the root's own block already sits at its offset: in place (sendbuff == recvbuff + rank * count)
          NCCLCHK(ncclGather(k == 0 ? (char *)root_base : (char *)get_ptr(k), k == 0 ? root_base : nullptr, cnt,
                             type, 0, m->comm[k], m->st[k]));
        } else if (k != 0 && cnt) {
          NCCLCHK(ncclSend(get_ptr(k), cnt, type, 0, m->comm[k], m->st[k]));
          NCCLCHK(ncclRecv((char *)root_base + (size_t)blocks[k].lo * per_path * elem, cnt, type, k, m->comm[0], m->st[0]));
        }
      }
      return 0;
    };
    NCCLCHK(ncclGroupStart());
    int rc = 0;
    if (payload != TPAMD_GATHER_COMPACT)
      rc = gather([&](int k) { return (void *)blocks[k].t; }, root_out->time, N, ncclDouble, 8);
    if (!rc) rc = gather([&](int k) { return (void *)blocks[k].sd; }, root_out->sd, N, ncclDouble, 8);
    if (!rc) rc = gather([&](int k) { return (void *)blocks[k].sdd; }, root_out->sdd, N, ncclDouble, 8);
    if (!rc && payload == TPAMD_GATHER_FULL)
      rc = gather([&](int k) { return (void *)blocks[k].q; }, root_out->q, N * D, ncclDouble, 8);
    if (!rc && root_out->status)
      rc = gather([&](int k) { return (void *)blocks[k].st; }, root_out->status, 1, ncclInt32, 4);
    if (!rc && root_out->last_extremal_index)
      rc = gather([&](int k) { return (void *)blocks[k].lei; }, root_out->last_extremal_index, 1, ncclInt32, 4);
    NCCLCHK(ncclGroupEnd());
    if (rc) return rc;
  }
  HIPCHK(hipSetDevice(m->dev[0]));
  if (payload == TPAMD_GATHER_COMPACT) {
    // the root rebuilds time from the gathered sd: ds and time_start of every path are known to
    // this process (one process drives all devices), so they go up from the host -- the
    // operations of k_setup_joint: s_end = path_start + delta (n - 1), ds = (s_end - s_start) / (n - 1)
    std::vector<double> aux(2 * (size_t)B);
    for (int b = 0; b < B; b++) {
      const int nb = ragged ? (in->num_samples_per_path[b] < (int)N ? in->num_samples_per_path[b] : (int)N) : (int)N;
      const double s0 = in->path_start[b];
      const double s1 = s0 + in->delta[b] * (nb - 1);
      aux[b] = (s1 - s0) / (nb - 1);
      aux[B + b] = in->time_start[b];
    }
    const size_t need = align_up(2 * (size_t)B * 8, 256) + (ragged ? align_up((size_t)B * 4, 256) : 0);
    if (need > m->root_aux_bytes) {
      if (m->root_aux) HIPCHK(hipFree(m->root_aux));
      m->root_aux = nullptr;
      m->root_aux_bytes = 0;
      HIPCHK(hipMalloc(&m->root_aux, need));
      m->root_aux_bytes = need;
    }
    double *d_aux = (double *)m->root_aux;
    int32_t *d_ns = ragged ? (int32_t *)((char *)m->root_aux + align_up(2 * (size_t)B * 8, 256)) : nullptr;
    HIPCHK(hipMemcpyAsync(d_aux, aux.data(), 2 * (size_t)B * 8, hipMemcpyHostToDevice, m->st[0]));
    if (ragged)
      HIPCHK(hipMemcpyAsync(d_ns, in->num_samples_per_path, (size_t)B * 4, hipMemcpyHostToDevice, m->st[0]));
    const int rc = tpamd_rebuild_time_device(m->eng[0], 1, B, (int)N, 0, root_out->sd, d_aux, d_aux + B, d_ns,
                                             root_out->time, m->st[0]);
    if (rc) return rc;
  }
  for (int k = 0; k < m->n; k++) {
    HIPCHK(hipSetDevice(m->dev[k]));
    HIPCHK(hipStreamSynchronize(m->st[k]));
  }
  (void)hipSetDevice(prev);
  return 0;
}

int tpamd_multi_time_joint_groups_host(tpamd_multi *m, int num_groups, const tpamd_joint_batch *batches,
                                       const tpamd_joint_inputs *inputs, const tpamd_path_outputs *outputs,
                                       const int32_t *group_device) {
  if (!m || num_groups < 0 || (num_groups > 0 && (!batches || !inputs || !outputs || !group_device)))
    return TPAMD_E_INVALID_ARGUMENT;
  for (int g = 0; g < num_groups; g++)
    if (group_device[g] < 0 || group_device[g] >= m->n) return TPAMD_E_INVALID_ARGUMENT;
  std::vector<int> rcs(m->n, 0);
  auto work = [&](int k) {
    std::vector<tpamd_joint_batch> b;
    std::vector<tpamd_joint_inputs> i;
    std::vector<tpamd_path_outputs> o;
    for (int g = 0; g < num_groups; g++)
      if (group_device[g] == k) { b.push_back(batches[g]); i.push_back(inputs[g]); o.push_back(outputs[g]); }
    if (!b.empty()) rcs[k] = tpamd_time_joint_groups_host(m->eng[k], (int)b.size(), b.data(), i.data(), o.data());
  };
  std::vector<std::thread> threads;
  for (int k = 1; k < m->n; k++) threads.emplace_back(work, k);
  work(0);
  for (auto &t : threads) t.join();
  for (int rc : rcs)
    if (rc) return rc;
  return 0;
}

}  // extern "C"

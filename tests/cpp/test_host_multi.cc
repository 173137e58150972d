// GPU test of the multi-device pieces (run by tests/test_gpu_multi.py): the engine pool behind the
// host mirror, BatchPathTiming's device set and sample buckets, and the one-process sharded solve
// with its RCCL gather (include/tpamd_multi.h) on a device set of size one -- the only size a
// one-GPU box offers; ncclCommInitAll / ncclGather still run, with one rank. The oracle is linked
// only as the checker.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/tpamd_multi.h"
#include "../../oracle/tp_oracle.h"
#include "../../x-edr-trajectory-planning_amd/host/batch_path_timing.h"
#include "../../x-edr-trajectory-planning_amd/host/engine_handle.h"
#include "../../x-edr-trajectory-planning_amd/host/timeable_path_joint_spline.h"

using namespace trajectory_planning;

static std::atomic<int> g_fail{0};
#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); g_fail++; } \
  } while (0)

static std::shared_ptr<TimeableJointSplinePath> MakePath(int N, const std::vector<VectorXd> &wps,
                                                         double vmax, double amax, double *delta_out) {
  const size_t D = wps[0].size();
  auto probe = std::make_shared<TimeableJointSplinePath>(JointPathOptions().set_num_dofs(D).set_num_path_samples(N));
  probe->SetWaypoints({wps.data(), wps.size()});
  const double delta = probe->knots().back() / (N - 1);
  auto path = std::make_shared<TimeableJointSplinePath>(
      JointPathOptions().set_num_dofs(D).set_num_path_samples(N).set_delta_parameter(delta));
  std::vector<double> v(D, vmax), a(D, amax);
  CHECK(path->SetMaxJointVelocity({v.data(), v.size()}).ok());
  CHECK(path->SetMaxJointAcceleration({a.data(), a.size()}).ok());
  CHECK(path->SetWaypoints({wps.data(), wps.size()}).ok());
  if (delta_out) *delta_out = delta;
  return path;
}

struct Lcg {
  unsigned long long seed;
  double operator()() {
    seed = seed * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(seed >> 11) / 9007199254740992.0;
  }
};

static std::vector<std::shared_ptr<TimeableJointSplinePath>> MixedPaths(int B, unsigned long long seed) {
  const int dofs[3] = {6, 7, 14};
  Lcg rnd{seed};
  std::vector<std::shared_ptr<TimeableJointSplinePath>> paths;
  for (int b = 0; b < B; b++) {
    const int D = dofs[b % 3], N = 120 + (int)(rnd() * 900), W = (b % 5 == 0) ? 4 : 6;
    std::vector<VectorXd> wps;
    for (int i = 0; i < W; i++) {
      VectorXd v(D);
      for (int d = 0; d < D; d++) v[d] = 4.0 * rnd() - 2.0;
      wps.push_back(v);
    }
    paths.push_back(MakePath(N, wps, 1.0 + rnd(), 2.0 + 2.0 * rnd(), nullptr));
  }
  return paths;
}

static bool SameResult(const BatchTimingResult &a, const BatchTimingResult &b) {
  return a.status == b.status && a.last_extremal_index == b.last_extremal_index && a.time == b.time && a.s == b.s &&
         a.sd == b.sd && a.sdd == b.sdd && a.q == b.q && a.qd == b.qd && a.qdd == b.qdd &&
         a.sample_offset == b.sample_offset;
}

// The pool: every engine call of the mirror leases an engine; two threads at work at once hold two
// engines (they no longer queue behind one global lock), and the results are those of one thread.
static void TestEnginePoolServesThreadsSideBySide() {
  const auto paths_a = MixedPaths(45, 11), paths_b = MixedPaths(45, 23);
  BatchTimingResult ref_a, ref_b;
  {
    BatchPathTiming ba, bb;
    CHECK(ba.SetPaths(paths_a).ok() && bb.SetPaths(paths_b).ok());
    CHECK(ba.ComputeTimingProfiles(0.5, &ref_a).ok());
    CHECK(bb.ComputeTimingProfiles(0.5, &ref_b).ok());
  }
  CHECK(tpamd::engines_idle(0) >= 1);
  std::atomic<int> holding{0}, max_holding{0};
  auto worker = [&](const std::vector<std::shared_ptr<TimeableJointSplinePath>> &paths, const BatchTimingResult &ref) {
    for (int rep = 0; rep < 6; rep++) {
      {   // leases overlap in time: a second engine appears instead of a wait
        tpamd::EngineLease lease = tpamd::acquire_engine(0);
        CHECK((bool)lease);
        const int h = ++holding;
        int m = max_holding.load();
        while (h > m && !max_holding.compare_exchange_weak(m, h)) {}
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        --holding;
      }
      BatchPathTiming b;
      BatchTimingResult r;
      CHECK(b.SetPaths(paths).ok());
      CHECK(b.ComputeTimingProfiles(0.5, &r).ok());
      CHECK(SameResult(r, ref));
    }
  };
  std::thread t1(worker, std::cref(paths_a), std::cref(ref_a)), t2(worker, std::cref(paths_b), std::cref(ref_b));
  t1.join();
  t2.join();
  CHECK(max_holding.load() == 2);
  CHECK(tpamd::engines_created(0) >= 2);
  CHECK(tpamd::engines_idle(0) == tpamd::engines_created(0));   // everything came back
}

// Device set and sample buckets: same bits whatever the grouping; a device the box does not have
// is refused.
static void TestBatchDeviceSetAndBuckets() {
  const auto paths = MixedPaths(60, 5);
  BatchPathTiming b0;
  BatchTimingResult r0;
  CHECK(b0.SetPaths(paths).ok());
  CHECK(b0.ComputeTimingProfiles(0.25, &r0).ok());
  for (size_t b = 0; b < paths.size(); b++) CHECK(r0.status[b] == 0);
  BatchPathTiming b1;
  BatchTimingResult r1;
  CHECK(b1.SetPaths(paths).ok());
  CHECK(b1.SetDevices({0}).ok());
  b1.SetSampleBucket(256);
  CHECK(b1.ComputeTimingProfiles(0.25, &r1).ok());
  CHECK(SameResult(r0, r1));
  const int visible = tpamd::device_count();
  CHECK(visible >= 1);
  CHECK(!b1.SetDevices({visible}).ok());
  CHECK(!b1.SetDevices({0, 0}).ok());
  if (visible >= 2) {   // (an 8-GPU node; never true on the one-GPU test boxes)
    BatchPathTiming b2;
    BatchTimingResult r2;
    CHECK(b2.SetPaths(paths).ok());
    std::vector<int> all;
    for (int d = 0; d < visible; d++) all.push_back(d);
    CHECK(b2.SetDevices(all).ok());
    CHECK(b2.ComputeTimingProfiles(0.25, &r2).ok());
    CHECK(SameResult(r0, r2));
  }
}

// The one-process sharded solve and its gather, on every device the box has (one here).
static void TestMultiGatherPayloads() {
  const int B = 40, D = 7, N = 400, W = 6, P = 3 * W - 2;
  Lcg rnd{99};
  std::vector<double> knots((size_t)B * (P + 3)), cps((size_t)B * P * D), vmax((size_t)B * D), amax((size_t)B * D),
      ps(B, 0.0), dl(B), sd0(B, 0.0), t0(B, 0.75);
  for (int b = 0; b < B; b++) {
    std::vector<VectorXd> wps;
    for (int i = 0; i < W; i++) {
      VectorXd v(D);
      for (int d = 0; d < D; d++) v[d] = 4.0 * rnd() - 2.0;
      wps.push_back(v);
    }
    const double vm = 1.0 + rnd(), am = 2.0 + 2.0 * rnd();
    auto p = MakePath(N, wps, vm, am, &dl[b]);
    std::copy(p->knots().begin(), p->knots().end(), knots.begin() + (size_t)b * (P + 3));
    std::copy(p->packed_control_points().begin(), p->packed_control_points().end(), cps.begin() + (size_t)b * P * D);
    for (int d = 0; d < D; d++) { vmax[b * D + d] = vm; amax[b * D + d] = am; }
  }
  std::vector<int32_t> ns(B);
  for (int b = 0; b < B; b++) ns[b] = (b % 3 == 0) ? N : 150 + (int)(rnd() * (N - 150));
  const int visible = tpamd::device_count();
  for (int ragged = 0; ragged < 2; ragged++) {
    std::vector<double> dlr = dl;
    if (ragged)
      for (int b = 0; b < B; b++) dlr[b] = knots[(size_t)b * (P + 3) + P + 2] / (ns[b] - 1);
    tpamd_joint_batch bt{B, D, N, P, 0, 0, 0.8};
    tpamd_joint_inputs in{knots.data(), cps.data(), vmax.data(), amax.data(), ps.data(), dlr.data(), sd0.data(),
                          nullptr, t0.data(), ragged ? ns.data() : nullptr};
    // reference: one engine, host buffers
    std::vector<double> rt((size_t)B * N, -7), rs((size_t)B * N, -7), rsd((size_t)B * N, -7), rsdd((size_t)B * N, -7),
        rq((size_t)B * N * D, -7), rqd((size_t)B * N * D, -7), rqdd((size_t)B * N * D, -7);
    std::vector<int32_t> rst(B, -1), rlei(B, 0);
    {
      tpamd::EngineLease lease = tpamd::acquire_engine(0);
      tpamd_path_outputs out{rt.data(), rs.data(), rsd.data(), rsdd.data(), rq.data(), rqd.data(), rqdd.data(),
                             rlei.data(), nullptr, rst.data(), nullptr};
      CHECK(tpamd_time_joint_paths_host(lease.get(), &bt, &in, &out) == 0);
      for (int b = 0; b < B; b++) CHECK(rst[b] == 0);
    }
    tpamd_multi *multi = nullptr;
    CHECK(tpamd_multi_create(visible, nullptr, /*force_rccl=*/1, &multi) == 0);
    if (!multi) return;
    CHECK(tpamd_multi_num_devices(multi) == visible && tpamd_multi_uses_rccl(multi) == 1);
    // (a) straight to the host, unequal blocks when there is more than one device
    {
      std::vector<double> t((size_t)B * N, -7), s((size_t)B * N, -7), sd((size_t)B * N, -7), sdd((size_t)B * N, -7),
          q((size_t)B * N * D, -7), qd((size_t)B * N * D, -7), qdd((size_t)B * N * D, -7);
      std::vector<int32_t> st(B, -1), lei(B, 0);
      tpamd_path_outputs out{t.data(), s.data(), sd.data(), sdd.data(), q.data(), qd.data(), qdd.data(),
                             lei.data(), nullptr, st.data(), nullptr};
      std::vector<double> cost(B);
      for (int b = 0; b < B; b++) cost[b] = (ragged ? ns[b] : N) * 196.0;
      std::vector<int32_t> begin(visible + 1);
      tpamd_shard_bounds_balanced(B, cost.data(), visible, begin.data());
      CHECK(tpamd_multi_time_joint_paths_host(multi, &bt, &in, begin.data(), &out, TPAMD_GATHER_FULL, nullptr) == 0);
      CHECK(st == rst && lei == rlei);
      for (int b = 0; b < B; b++) {
        const size_t n = ragged ? ns[b] : N;
        for (size_t i = 0; i < n; i++) {
          const size_t o = (size_t)b * N + i;
          CHECK(t[o] == rt[o] && s[o] == rs[o] && sd[o] == rsd[o] && sdd[o] == rsdd[o]);
          for (int d = 0; d < D; d++) CHECK(q[o * D + d] == rq[o * D + d] && qdd[o * D + d] == rqdd[o * D + d]);
        }
      }
    }
    // (b) the gather to the root, every payload
    hipSetDevice(tpamd_multi_device(multi, 0));
    double *d_t, *d_sd, *d_sdd, *d_q;
    int32_t *d_st, *d_lei;
    hipMalloc((void **)&d_t, (size_t)B * N * 8); hipMalloc((void **)&d_sd, (size_t)B * N * 8);
    hipMalloc((void **)&d_sdd, (size_t)B * N * 8); hipMalloc((void **)&d_q, (size_t)B * N * D * 8);
    hipMalloc((void **)&d_st, B * 4); hipMalloc((void **)&d_lei, B * 4);
    for (int payload : {TPAMD_GATHER_COMPACT, TPAMD_GATHER_PROFILE, TPAMD_GATHER_FULL}) {
      hipMemset(d_t, 0xff, (size_t)B * N * 8); hipMemset(d_sd, 0xff, (size_t)B * N * 8);
      hipMemset(d_sdd, 0xff, (size_t)B * N * 8); hipMemset(d_q, 0xff, (size_t)B * N * D * 8);
      hipMemset(d_st, 0xff, B * 4);
      tpamd_path_outputs root{d_t, nullptr, d_sd, d_sdd, d_q, nullptr, nullptr, d_lei, nullptr, d_st, nullptr};
      CHECK(tpamd_multi_time_joint_paths_host(multi, &bt, &in, nullptr, nullptr, payload, &root) == 0);
      std::vector<double> t((size_t)B * N), sd((size_t)B * N), sdd((size_t)B * N), q((size_t)B * N * D);
      std::vector<int32_t> st(B), lei(B);
      hipMemcpy(t.data(), d_t, t.size() * 8, hipMemcpyDeviceToHost);
      hipMemcpy(sd.data(), d_sd, sd.size() * 8, hipMemcpyDeviceToHost);
      hipMemcpy(sdd.data(), d_sdd, sdd.size() * 8, hipMemcpyDeviceToHost);
      hipMemcpy(q.data(), d_q, q.size() * 8, hipMemcpyDeviceToHost);
      hipMemcpy(st.data(), d_st, B * 4, hipMemcpyDeviceToHost);
      hipMemcpy(lei.data(), d_lei, B * 4, hipMemcpyDeviceToHost);
      CHECK(st == rst && lei == rlei);
      for (int b = 0; b < B; b++) {
        const size_t n = ragged ? ns[b] : N;
        for (size_t i = 0; i < n; i++) {
          const size_t o = (size_t)b * N + i;
          CHECK(t[o] == rt[o]);                 // COMPACT: rebuilt on the root from sd, same bits
          CHECK(sd[o] == rsd[o] && sdd[o] == rsdd[o]);
          if (payload == TPAMD_GATHER_FULL)
            for (int d = 0; d < D; d++) CHECK(q[o * D + d] == rq[o * D + d]);
        }
      }
      CHECK(tpamd_gather_bytes_per_path(payload, N, D) ==
            (payload == TPAMD_GATHER_COMPACT ? 16u * N + 16u : payload == TPAMD_GATHER_PROFILE ? 24u * N : 24u * N + 8u * N * D));
    }
    hipFree(d_t); hipFree(d_sd); hipFree(d_sdd); hipFree(d_q); hipFree(d_st); hipFree(d_lei);
    // (c) joint groups per device
    {
      std::vector<double> t((size_t)B * N, -7), s((size_t)B * N, -7), sd((size_t)B * N, -7), sdd((size_t)B * N, -7);
      std::vector<int32_t> st(B, -1);
      tpamd_path_outputs out{t.data(), s.data(), sd.data(), sdd.data(), nullptr, nullptr, nullptr, nullptr, nullptr,
                             st.data(), nullptr};
      const int32_t dev0 = 0;
      CHECK(tpamd_multi_time_joint_groups_host(multi, 1, &bt, &in, &out, &dev0) == 0);
      CHECK(st == rst);
      for (int b = 0; b < B; b++)
        for (size_t i = 0; i < (size_t)(ragged ? ns[b] : N); i++) CHECK(t[(size_t)b * N + i] == rt[(size_t)b * N + i]);
    }
    tpamd_multi_destroy(multi);
  }
  // argument checks
  tpamd_multi *bad = nullptr;
  CHECK(tpamd_multi_create(visible + 1, nullptr, 0, &bad) == TPAMD_E_INVALID_ARGUMENT && bad == nullptr);
  const int twice[2] = {0, 0};
  if (visible >= 2) CHECK(tpamd_multi_create(2, twice, 0, &bad) == TPAMD_E_INVALID_ARGUMENT);
}

int main() {
  TestEnginePoolServesThreadsSideBySide();
  TestBatchDeviceSetAndBuckets();
  TestMultiGatherPayloads();
  tpamd::release_idle_engines();
  if (g_fail == 0) std::printf("ALL OK\n");
  else std::printf("%d CHECKS FAILED\n", g_fail.load());
  return g_fail == 0 ? 0 : 1;
}

// Micro-benchmark: ordered running sum t_i = t_{i-1} + dt_i over n doubles in LDS, one wave each
// (time_optimal_path_timing.cc:453-454 must be summed left to right).
//   variant 0: every lane the same serial arithmetic (reference for the bits)
//   variant 1: lane pipeline -- one add and a 64-bit wave_shr:1 per element (sweep tail, round 2)
//   variant 2: v_fmac_f64 with a DPP row_newbcast operand: acc = dt[lane c of the row] * 1.0 + acc,
//              one instruction per element; EXEC shrinks by one lane per element so that lane c
//              stops after its own element; rows chained through row_bcast:15
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o chain_bench chain_bench.hip
// Run:   ./chain_bench [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__device__ __forceinline__ long long stamp() {
  long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                          __builtin_amdgcn_readlane(__double2loint(v), l));
}

// ---- variant 2 -------------------------------------------------------------------------
// One 16-lane row: lanes c..15 of the row add element c. HALF = "lo" / "hi": which half of EXEC
// holds the row, SH: the row's position inside that half (0 or 16).
#define TP_STEP(HALF, SH, C)                                                              \
  "s_mov_b32 exec_" HALF ", %[m" #C "]\n\t"                                               \
  "v_fmac_f64_dpp %[acc], %[dt], %[one] row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t"
template <int ROW>
__device__ __forceinline__ void ordered_row(double &acc, double dt, double one) {
  constexpr unsigned SH = (ROW & 1) ? 16u : 0u;
#define TP_M(C) [m##C] "n"((int)(((0xffffu << C) & 0xffffu) << SH))
  unsigned long long save;
  if (ROW < 2) {
    asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_hi, 0\n\t"
                 TP_STEP("lo", SH, 0) TP_STEP("lo", SH, 1) TP_STEP("lo", SH, 2) TP_STEP("lo", SH, 3)
                 TP_STEP("lo", SH, 4) TP_STEP("lo", SH, 5) TP_STEP("lo", SH, 6) TP_STEP("lo", SH, 7)
                 TP_STEP("lo", SH, 8) TP_STEP("lo", SH, 9) TP_STEP("lo", SH, 10) TP_STEP("lo", SH, 11)
                 TP_STEP("lo", SH, 12) TP_STEP("lo", SH, 13) TP_STEP("lo", SH, 14) TP_STEP("lo", SH, 15)
                 "s_mov_b64 exec, %[sv]\n\t"
                 : [acc] "+v"(acc), [sv] "=&s"(save)
                 : [dt] "v"(dt), [one] "v"(one), TP_M(0), TP_M(1), TP_M(2), TP_M(3), TP_M(4), TP_M(5), TP_M(6),
                   TP_M(7), TP_M(8), TP_M(9), TP_M(10), TP_M(11), TP_M(12), TP_M(13), TP_M(14), TP_M(15));
  } else {
    asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, 0\n\t"
                 TP_STEP("hi", SH, 0) TP_STEP("hi", SH, 1) TP_STEP("hi", SH, 2) TP_STEP("hi", SH, 3)
                 TP_STEP("hi", SH, 4) TP_STEP("hi", SH, 5) TP_STEP("hi", SH, 6) TP_STEP("hi", SH, 7)
                 TP_STEP("hi", SH, 8) TP_STEP("hi", SH, 9) TP_STEP("hi", SH, 10) TP_STEP("hi", SH, 11)
                 TP_STEP("hi", SH, 12) TP_STEP("hi", SH, 13) TP_STEP("hi", SH, 14) TP_STEP("hi", SH, 15)
                 "s_mov_b64 exec, %[sv]\n\t"
                 : [acc] "+v"(acc), [sv] "=&s"(save)
                 : [dt] "v"(dt), [one] "v"(one), TP_M(0), TP_M(1), TP_M(2), TP_M(3), TP_M(4), TP_M(5), TP_M(6),
                   TP_M(7), TP_M(8), TP_M(9), TP_M(10), TP_M(11), TP_M(12), TP_M(13), TP_M(14), TP_M(15));
  }
#undef TP_M
}
// lane 15 of row ROW-1 -> every lane of row ROW (row_bcast:15 writes the next row only)
template <int ROW>
__device__ __forceinline__ void row_carry(double &acc) {
  int lo = __double2loint(acc), hi = __double2hiint(acc);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 1 << ROW, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 1 << ROW, 0xf, false);
  acc = __hiloint2double(hi, lo);
}
// lane L returns t + dt_0 + ... + dt_L, added left to right (dt_i = dtv of lane i)
__device__ __forceinline__ double ordered_block64(double t, double dtv) {
  double acc = t;
  const double one = 1.0;
  ordered_row<0>(acc, dtv, one);
  row_carry<1>(acc);
  ordered_row<1>(acc, dtv, one);
  row_carry<2>(acc);
  ordered_row<2>(acc, dtv, one);
  row_carry<3>(acc);
  ordered_row<3>(acc, dtv, one);
  return acc;
}

template <int VARIANT>
__global__ void __launch_bounds__(128) k_chain(const double *dt_g, double *t_g, long long *cycles, int n) {
  extern __shared__ double tl[];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  double *my = tl + (size_t)w * n;
  const double *src = dt_g + (size_t)blockIdx.x * n;
  for (int i = lane; i < n; i += 64) my[i] = src[i];
  __syncthreads();
  const long long t0 = stamp();
  double t = 0.25;
  if (VARIANT == 0) {
    for (int k = 0; k < n; k++) { t = t + my[k]; if (lane == 0) my[k] = t; }
  } else if (VARIANT == 1) {
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const double dtv = (i < n) ? my[i] : 0.0;
      double x = t, y = 0.0;
#pragma unroll
      for (int r = 0; r < 64; r++) {
        y = x + dtv;
        if (r < 63) {
          int lo = __double2loint(x), hi = __double2hiint(x);
          lo = __builtin_amdgcn_update_dpp(lo, __double2loint(y), 0x138, 0xf, 0xf, false);
          hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(y), 0x138, 0xf, 0xf, false);
          x = __hiloint2double(hi, lo);
        }
      }
      if (i < n) my[i] = y;
      t = readlane_f64(y, 63);
    }
  } else {
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const double dtv = (i < n) ? my[i] : 0.0;
      const double tv = ordered_block64(t, dtv);
      if (i < n) my[i] = tv;
      t = readlane_f64(tv, 63);
    }
  }
  const long long t1 = stamp();
  __syncthreads();
  for (int i = lane; i < n; i += 64) t_g[(size_t)(blockIdx.x * 2 + w) * n + i] = my[i];
  if (lane == 0) cycles[blockIdx.x * 2 + w] = t1 - t0;
}

int main(int argc, char **argv) {
  const int n = 2000, wgs = argc > 1 ? atoi(argv[1]) : 1024;
  std::vector<double> dt((size_t)wgs * n);
  unsigned long long s = 88172645463325252ull;
  for (auto &v : dt) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = 1e-3 + (double)(s >> 11) * (1.0 / 9007199254740992.0) * 1e-2; }
  // a few special values: zeros, a huge and an infinite increment (later sums are inf, earlier ones are not)
  dt[5] = 0.0; dt[6] = 0.0; dt[(size_t)n + 700] = 1e300; dt[(size_t)n + 701] = 1e308; dt[(size_t)2 * n + 1234] = 1.0 / 0.0;
  double *d_dt, *d_t; long long *d_c;
  hipMalloc(&d_dt, dt.size() * 8); hipMalloc(&d_t, dt.size() * 2 * 8); hipMalloc(&d_c, (size_t)wgs * 2 * 8);
  hipMemcpy(d_dt, dt.data(), dt.size() * 8, hipMemcpyHostToDevice);
  std::vector<double> r[3];
  std::vector<long long> c((size_t)wgs * 2);
  const size_t lds = (size_t)2 * n * 8;
  for (int v = 0; v < 3; v++) {
    r[v].resize(dt.size() * 2);
    for (int rep = 0; rep < 2; rep++) {
      if (v == 0) hipLaunchKernelGGL(k_chain<0>, dim3(wgs), dim3(128), lds, 0, d_dt, d_t, d_c, n);
      else if (v == 1) hipLaunchKernelGGL(k_chain<1>, dim3(wgs), dim3(128), lds, 0, d_dt, d_t, d_c, n);
      else hipLaunchKernelGGL(k_chain<2>, dim3(wgs), dim3(128), lds, 0, d_dt, d_t, d_c, n);
      hipDeviceSynchronize();
    }
    hipMemcpy(r[v].data(), d_t, r[v].size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : c) mean += (double)x; mean /= c.size();
    printf("variant %d: %d workgroups x 2 waves, %.1f cycles per element (mean over waves)\n", v, wgs, mean / n);
  }
  printf("variant 1 bit-identical to 0: %s\n", memcmp(r[0].data(), r[1].data(), r[0].size() * 8) == 0 ? "yes" : "NO");
  printf("variant 2 bit-identical to 0: %s\n", memcmp(r[0].data(), r[2].data(), r[0].size() * 8) == 0 ? "yes" : "NO");
  return 0;
}

// Micro-benchmark: ordered running sum t_i = t_{i-1} + dt_i over n doubles in LDS, one wave each.
//   variant 0: every lane the same arithmetic, dt read from LDS in 16-byte pairs (the sweep tail of round 2)
//   variant 1: 64 elements per block, one per lane; v_readlane feeds the add and EXEC shrinks by one
//              lane per step, so lane L ends up with the sum through element L
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o chain_bench chain_bench.hip
// Run:   ./chain_bench [workgroups]   (prints cycles per element for both variants and checks bit equality)
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

// lane L returns t + dt_0 + ... + dt_L, added left to right (dt_i = dtv of lane i)
__device__ __forceinline__ double chain_block64(double t, double dtv) {
  const int lo = __double2loint(dtv), hi = __double2hiint(dtv);
  double tv = t;
  asm volatile(
      "s_mov_b64 s[20:21], exec\n\t"
      "v_readlane_b32 s22, %1, 0\n\t"
      "v_readlane_b32 s23, %2, 0\n\t"
      "v_readlane_b32 s24, %1, 1\n\t"
      "v_readlane_b32 s25, %2, 1\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 2\n\t"
      "v_readlane_b32 s23, %2, 2\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 3\n\t"
      "v_readlane_b32 s25, %2, 3\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 4\n\t"
      "v_readlane_b32 s23, %2, 4\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 5\n\t"
      "v_readlane_b32 s25, %2, 5\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 6\n\t"
      "v_readlane_b32 s23, %2, 6\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 7\n\t"
      "v_readlane_b32 s25, %2, 7\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 8\n\t"
      "v_readlane_b32 s23, %2, 8\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 9\n\t"
      "v_readlane_b32 s25, %2, 9\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 10\n\t"
      "v_readlane_b32 s23, %2, 10\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 11\n\t"
      "v_readlane_b32 s25, %2, 11\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 12\n\t"
      "v_readlane_b32 s23, %2, 12\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 13\n\t"
      "v_readlane_b32 s25, %2, 13\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 14\n\t"
      "v_readlane_b32 s23, %2, 14\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 15\n\t"
      "v_readlane_b32 s25, %2, 15\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 16\n\t"
      "v_readlane_b32 s23, %2, 16\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 17\n\t"
      "v_readlane_b32 s25, %2, 17\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 18\n\t"
      "v_readlane_b32 s23, %2, 18\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 19\n\t"
      "v_readlane_b32 s25, %2, 19\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 20\n\t"
      "v_readlane_b32 s23, %2, 20\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 21\n\t"
      "v_readlane_b32 s25, %2, 21\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 22\n\t"
      "v_readlane_b32 s23, %2, 22\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 23\n\t"
      "v_readlane_b32 s25, %2, 23\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 24\n\t"
      "v_readlane_b32 s23, %2, 24\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 25\n\t"
      "v_readlane_b32 s25, %2, 25\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 26\n\t"
      "v_readlane_b32 s23, %2, 26\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 27\n\t"
      "v_readlane_b32 s25, %2, 27\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 28\n\t"
      "v_readlane_b32 s23, %2, 28\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 29\n\t"
      "v_readlane_b32 s25, %2, 29\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 30\n\t"
      "v_readlane_b32 s23, %2, 30\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 31\n\t"
      "v_readlane_b32 s25, %2, 31\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 32\n\t"
      "v_readlane_b32 s23, %2, 32\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 33\n\t"
      "v_readlane_b32 s25, %2, 33\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 34\n\t"
      "v_readlane_b32 s23, %2, 34\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 35\n\t"
      "v_readlane_b32 s25, %2, 35\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 36\n\t"
      "v_readlane_b32 s23, %2, 36\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 37\n\t"
      "v_readlane_b32 s25, %2, 37\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 38\n\t"
      "v_readlane_b32 s23, %2, 38\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 39\n\t"
      "v_readlane_b32 s25, %2, 39\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 40\n\t"
      "v_readlane_b32 s23, %2, 40\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 41\n\t"
      "v_readlane_b32 s25, %2, 41\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 42\n\t"
      "v_readlane_b32 s23, %2, 42\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 43\n\t"
      "v_readlane_b32 s25, %2, 43\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 44\n\t"
      "v_readlane_b32 s23, %2, 44\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 45\n\t"
      "v_readlane_b32 s25, %2, 45\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 46\n\t"
      "v_readlane_b32 s23, %2, 46\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 47\n\t"
      "v_readlane_b32 s25, %2, 47\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 48\n\t"
      "v_readlane_b32 s23, %2, 48\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 49\n\t"
      "v_readlane_b32 s25, %2, 49\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 50\n\t"
      "v_readlane_b32 s23, %2, 50\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 51\n\t"
      "v_readlane_b32 s25, %2, 51\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 52\n\t"
      "v_readlane_b32 s23, %2, 52\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 53\n\t"
      "v_readlane_b32 s25, %2, 53\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 54\n\t"
      "v_readlane_b32 s23, %2, 54\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 55\n\t"
      "v_readlane_b32 s25, %2, 55\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 56\n\t"
      "v_readlane_b32 s23, %2, 56\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 57\n\t"
      "v_readlane_b32 s25, %2, 57\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 58\n\t"
      "v_readlane_b32 s23, %2, 58\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 59\n\t"
      "v_readlane_b32 s25, %2, 59\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 60\n\t"
      "v_readlane_b32 s23, %2, 60\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 61\n\t"
      "v_readlane_b32 s25, %2, 61\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s22, %1, 62\n\t"
      "v_readlane_b32 s23, %2, 62\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "v_readlane_b32 s24, %1, 63\n\t"
      "v_readlane_b32 s25, %2, 63\n\t"
      "v_add_f64 %0, %0, s[22:23]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "s_nop 1\n\t"
      "v_add_f64 %0, %0, s[24:25]\n\t"
      "s_lshl_b64 exec, exec, 1\n\t"
      "s_mov_b64 exec, s[20:21]\n\t"
      "s_nop 1\n\t"
      : "+v"(tv)
      : "v"(lo), "v"(hi)
      : "s20", "s21", "s22", "s23", "s24", "s25");
  return tv;
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
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 *tl2 = reinterpret_cast<f64x2 *>(my);
    constexpr int kBlk = 8;
    f64x2 d[kBlk], dn[kBlk];
    int k = 0;
    if (2 * kBlk <= n)
      for (int i = 0; i < kBlk; i++) d[i] = tl2[i];
    for (; k + 2 * kBlk <= n; k += 2 * kBlk) {
      const bool more = k + 4 * kBlk <= n;
      if (more)
        for (int i = 0; i < kBlk; i++) dn[i] = tl2[(k >> 1) + kBlk + i];
#pragma unroll
      for (int i = 0; i < kBlk; i++) {
        t = t + d[i].x; d[i].x = t;
        t = t + d[i].y; d[i].y = t;
      }
      if (lane == 0)
        for (int i = 0; i < kBlk; i++) tl2[(k >> 1) + i] = d[i];
      for (int i = 0; i < kBlk; i++) d[i] = dn[i];
    }
    for (; k < n; k++) { t = t + my[k]; if (lane == 0) my[k] = t; }
  } else {
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const double dtv = (i < n) ? my[i] : 0.0;
      const double tv = chain_block64(t, dtv);
      if (i < n) my[i] = tv;
      t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tv), 63),
                           __builtin_amdgcn_readlane(__double2loint(tv), 63));
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
  double *d_dt, *d_t; long long *d_c;
  hipMalloc(&d_dt, dt.size() * 8); hipMalloc(&d_t, dt.size() * 2 * 8); hipMalloc(&d_c, (size_t)wgs * 2 * 8);
  hipMemcpy(d_dt, dt.data(), dt.size() * 8, hipMemcpyHostToDevice);
  std::vector<double> r0(dt.size() * 2), r1(dt.size() * 2);
  std::vector<long long> c((size_t)wgs * 2);
  const size_t lds = (size_t)2 * n * 8;
  for (int v = 0; v < 2; v++) {
    for (int rep = 0; rep < 2; rep++) {
      if (v == 0) hipLaunchKernelGGL(k_chain<0>, dim3(wgs), dim3(128), lds, 0, d_dt, d_t, d_c, n);
      else hipLaunchKernelGGL(k_chain<1>, dim3(wgs), dim3(128), lds, 0, d_dt, d_t, d_c, n);
      hipDeviceSynchronize();
    }
    hipMemcpy((v ? r1 : r0).data(), d_t, r0.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : c) mean += (double)x; mean /= c.size();
    printf("variant %d: %d workgroups x 2 waves, %.1f cycles per element (mean over waves)\n", v, wgs, mean / n);
  }
  printf("results bit-identical: %s\n", memcmp(r0.data(), r1.data(), r0.size() * 8) == 0 ? "yes" : "NO");
  return 0;
}

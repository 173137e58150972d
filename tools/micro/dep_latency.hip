// Micro-benchmark: issue-to-issue latency of DEPENDENT VALU operations on one wave (cycles by s_memtime).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o dep_latency dep_latency.hip ; run: ./dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ long long stamp() {
  long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void k_dep(double *out, long long *cycles, double a, double b, int iters) {
  double v = a + threadIdx.x * 1e-9;
  float f = (float)a;
  double w = b;
  const long long t0 = stamp();
  for (int it = 0; it < iters; it++) {
    if (OP == 0) asm volatile(REP64("v_add_f64 %0, %0, %1\n\t") : "+v"(v) : "v"(b));
    if (OP == 1) asm volatile(REP64("v_mul_f64 %0, %0, %1\n\t") : "+v"(v) : "v"(b));
    if (OP == 2) asm volatile(REP64("v_fma_f64 %0, %0, %1, %1\n\t") : "+v"(v) : "v"(b));
    if (OP == 3) asm volatile(REP64("v_add_f32 %0, %0, %1\n\t") : "+v"(f) : "v"((float)b));
    if (OP == 4) asm volatile(REP64("v_add_f64 %0, %0, %2\n\tv_add_f64 %1, %1, %2\n\t") : "+v"(v), "+v"(w) : "v"(b));   // two independent chains
    if (OP == 5) asm volatile(REP64("v_mov_b32 %0, %0\n\t") : "+v"(f));
    if (OP == 6) asm volatile(REP64("v_rcp_f64 %0, %0\n\t") : "+v"(v));
  }
  const long long t1 = stamp();
  out[blockIdx.x * blockDim.x + threadIdx.x] = v + f + w;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  double *d_out; long long *d_c;
  hipMalloc(&d_out, 1 << 20); hipMalloc(&d_c, 1 << 16);
  const char *names[] = {"v_add_f64 dependent", "v_mul_f64 dependent", "v_fma_f64 dependent", "v_add_f32 dependent",
                         "two interleaved v_add_f64 chains (per pair)", "v_mov_b32 dependent", "v_rcp_f64 dependent"};
  for (int op = 0; op < 7; op++) {
    for (int waves = 1; waves <= 2; waves++) {   // waves per SIMD: blocks of 256 threads = 4 waves = one per SIMD
      const int iters = 32, blocks = 256 * waves;
      for (int rep = 0; rep < 2; rep++) {
        switch (op) {
          case 0: hipLaunchKernelGGL(k_dep<0>, dim3(blocks), dim3(256), 0, 0, d_out, d_c, 1.0, 1e-3, iters); break;
          case 1: hipLaunchKernelGGL(k_dep<1>, dim3(blocks), dim3(256), 0, 0, d_out, d_c, 1.0, 1.0000001, iters); break;
          case 2: hipLaunchKernelGGL(k_dep<2>, dim3(blocks), dim3(256), 0, 0, d_out, d_c, 1.0, 0.5, iters); break;
          case 3: hipLaunchKernelGGL(k_dep<3>, dim3(blocks), dim3(256), 0, 0, d_out, d_c, 1.0, 1e-3, iters); break;
          case 4: hipLaunchKernelGGL(k_dep<4>, dim3(blocks), dim3(256), 0, 0, d_out, d_c, 1.0, 1e-3, iters); break;
          case 5: hipLaunchKernelGGL(k_dep<5>, dim3(blocks), dim3(256), 0, 0, d_out, d_c, 1.0, 1e-3, iters); break;
          case 6: hipLaunchKernelGGL(k_dep<6>, dim3(blocks), dim3(256), 0, 0, d_out, d_c, 1.5, 1e-3, iters); break;
        }
        hipDeviceSynchronize();
      }
      long long c[2048];
      hipMemcpy(c, d_c, blocks * 8, hipMemcpyDeviceToHost);
      double mean = 0; for (int i = 0; i < blocks; i++) mean += (double)c[i]; mean /= blocks;
      printf("%-48s %d wave(s)/SIMD: %6.2f cycles per instruction\n", names[op], waves, mean / (64.0 * iters) / (op == 4 ? 1 : 1));
    }
  }
  return 0;
}

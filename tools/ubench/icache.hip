// Microbenchmark: does straight-line code far larger than the instruction cache run at full
// issue rate with one wave per SIMD?  (design input for model-specialised kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define BODY8 \
  "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n" \
  "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"

template <int REPT, int LOOPS>
__global__ __launch_bounds__(64) void k(double* out, double x, double y, long long* cyc) {
  double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int l = 0; l < LOOPS; ++l) {
    if (REPT == 16) asm volatile(".rept 16\n" BODY8 ".endr\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if (REPT == 512) asm volatile(".rept 512\n" BODY8 ".endr\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if (REPT == 1536) asm volatile(".rept 1536\n" BODY8 ".endr\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if (REPT == 2048) asm volatile(".rept 2048\n" BODY8 ".endr\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if (REPT == 8192) asm volatile(".rept 8192\n" BODY8 ".endr\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int REPT, int LOOPS>
void run(int waves) {
  double* out; long long* cyc;
  CK(hipMalloc(&out, waves * 64 * 8)); CK(hipMalloc(&cyc, waves * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0));
    k<REPT, LOOPS><<<waves, 64>>>(out, 1.0000001, 0.5, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c0; CK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
  const double ninst = 8.0 * REPT * LOOPS;
  printf("code %7d KB  loops %5d  waves %5d : %8.3f ms  %6.2f ns/inst  wave0 %.2f cnt/inst\n", REPT * 64 / 1024, LOOPS, waves, ms,
         ms * 1e6 / ninst, c0 / ninst);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  for (int waves : {1024, 2048}) {
    run<16, 4096>(waves);     // 1 KB loop
    run<512, 128>(waves);     // 32 KB
    run<1536, 32>(waves);     // 96 KB
    run<8192, 1>(waves);      // 512 KB, single pass (cold)
    run<2048, 1>(waves);      // 128 KB single pass
  }
  return 0;
}

// Microbenchmark: issue rate of DEPENDENT fp64 FMA chains with one wave per SIMD
// (how many independent chains does the generated code need to interleave?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int CH>
__global__ __launch_bounds__(64) void k(double* out, double x, double y, long long* cyc) {
  double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int l = 0; l < 256; ++l) {
    if (CH == 1) asm volatile(".rept 64\n v_fma_f64 %0, %1, %2, %0\n .endr\n" : "+v"(a0) : "v"(x), "v"(y));
    if (CH == 2) asm volatile(".rept 32\n v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1\n .endr\n" : "+v"(a0), "+v"(a1) : "v"(x), "v"(y));
    if (CH == 3) asm volatile(".rept 21\n v_fma_f64 %0, %3, %4, %0\n v_fma_f64 %1, %3, %4, %1\n v_fma_f64 %2, %3, %4, %2\n .endr\n v_fma_f64 %0, %3, %4, %0\n" : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(x), "v"(y));
    if (CH == 4) asm volatile(".rept 16\n v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3\n .endr\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
    if (CH == 11) asm volatile(".rept 64\n v_mul_f64 %0, %1, %0\n .endr\n" : "+v"(a0) : "v"(x));
    if (CH == 12) asm volatile(".rept 64\n v_add_f64 %0, %1, %0\n .endr\n" : "+v"(a0) : "v"(x));
    if (CH == 13) asm volatile(".rept 32\n v_fma_f64 %0, %2, %3, %0\n v_add_f64 %1, %2, %1\n .endr\n" : "+v"(a0), "+v"(a1) : "v"(x), "v"(y));
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
void run(const char* what) {
  double* out; long long* cyc;
  CK(hipMalloc(&out, 1024 * 64 * 8)); CK(hipMalloc(&cyc, 1024 * 8));
  for (int it = 0; it < 2; ++it) { k<CH><<<1024, 64>>>(out, 1.0000001, 0.5, cyc); CK(hipDeviceSynchronize()); }
  long long c0; CK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
  printf("%-40s %.2f cycles/inst\n", what, c0 / (256.0 * 64));
  CK(hipFree(out)); CK(hipFree(cyc));
}
int main() {
  run<1>("fma f64, 1 dependent chain");
  run<2>("fma f64, 2 interleaved chains");
  run<3>("fma f64, 3 interleaved chains");
  run<4>("fma f64, 4 interleaved chains");
  run<11>("mul f64, 1 dependent chain");
  run<12>("add f64, 1 dependent chain");
  run<13>("fma f64 chain interleaved with an add f64 chain");
  return 0;
}

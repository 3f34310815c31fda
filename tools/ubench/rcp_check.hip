// spec_rcp (brbd_spec.hpp) against the IEEE division on the device: max error in ulps over
// log-uniform magnitudes 1e-150 .. 1e150, both signs.
// hipcc -O3 --offload-arch=gfx950 -I pinocchio_amd/csrc -I include -o tools/ubench/rcp_check tools/ubench/rcp_check.hip
#include "brbd_spec.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k(const double* x, double* fast, double* ieee, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { fast[i] = brbd::spec_rcp(x[i]); ieee[i] = 1.0 / x[i]; }
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), f(n), g(n);
  srand(7);
  const double ranges[3] = {3.0, 12.0, 150.0};
  for (int r = 0; r < 3; ++r) {
    for (int i = 0; i < n; ++i) {
      const double u = (double)rand() / RAND_MAX * 2 - 1, m = 1.0 + (double)rand() / RAND_MAX;
      x[i] = ((rand() & 1) ? -m : m) * pow(10.0, u * ranges[r]);
    }
    double *dx, *df, *dg;
    CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&df, n * 8)); CK(hipMalloc(&dg, n * 8));
    CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    k<<<n / 256, 256>>>(dx, df, dg, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(f.data(), df, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(g.data(), dg, n * 8, hipMemcpyDeviceToHost));
    long long worst = 0, differ = 0;
    for (int i = 0; i < n; ++i) {
      long long a, b; memcpy(&a, &f[i], 8); memcpy(&b, &g[i], 8);
      const long long d = llabs(a - b);
      if (d > worst) worst = d;
      differ += d != 0;
      if (g[i] != 1.0 / x[i]) { printf("device division differs from the host's at %g\n", x[i]); return 1; }
    }
    printf("|x| in 1e+-%g: max %lld ulp, %.2f %% of results differ from the IEEE quotient\n", ranges[r], worst, 100.0 * differ / n);
    CK(hipFree(dx)); CK(hipFree(df)); CK(hipFree(dg));
  }
  return 0;
}

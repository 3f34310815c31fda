/* CPU copy of spec_sincos (pinocchio_amd/csrc/brbd_spec.hpp), operation for operation, against
   glibc's sincos: gcc -O2 -o /tmp/sincos_check tools/ubench/sincos_check.c -lm && /tmp/sincos_check */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static void fast_sincos(double x, double* so, double* co) {
  const double MAGIC = 6755399441055744.0; /* 1.5 * 2^52 */
  const double t = fma(x, 0.63661977236758138, MAGIC);
  const double n = t - MAGIC;
  uint64_t tb; memcpy(&tb, &t, 8);
  const uint32_t q = (uint32_t)tb;
  double r = fma(-n, 1.5707963267948966, x);
  r = fma(-n, 6.123233995736766e-17, r);
  r = (fabs(x) < 1125899906842624.0) ? r : NAN;
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06); ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03); ps = fma(z, ps, -1.66666666666666324348e-01);
  ps = fma(r * z, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07); pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03); pc = fma(z, pc, 4.16666666666666019037e-02);
  pc = fma(z * z, pc, fma(z, -0.5, 1.0));
  double s = (q & 1) ? pc : ps, c = (q & 1) ? ps : pc;
  uint64_t sb, cb; memcpy(&sb, &s, 8); memcpy(&cb, &c, 8);
  sb ^= (uint64_t)((q << 30) & 0x80000000u) << 32;
  cb ^= (uint64_t)(((q + 1) << 30) & 0x80000000u) << 32;
  memcpy(so, &sb, 8); memcpy(co, &cb, 8);
}
int main() {
  srand(1);
  double scales[] = {1e-3, 1, 10, 1000, 1e6, 5e8, 1e12, 1e15};
  for (int k = 0; k < 8; ++k) {
    double ws = 0, wc = 0;
    for (int i = 0; i < 4000000; ++i) {
      double x = ((double)rand() / RAND_MAX * 2 - 1) * scales[k];
      double s, c, s0, c0; fast_sincos(x, &s, &c); sincos(x, &s0, &c0);
      double es = fabs(s - s0), ec = fabs(c - c0);
      if (es > ws) ws = es; if (ec > wc) wc = ec;
    }
    printf("scale %g: max abs err sin %.3e cos %.3e\n", scales[k], ws, wc);
  }
  double ws = 0, wc = 0;
  for (int k = -100000; k <= 100000; ++k) for (int d = -2; d <= 2; ++d) {
    double x = k * 0.78539816339744831; for (int j = 0; j < abs(d); ++j) x = nextafter(x, d > 0 ? 1e300 : -1e300);
    double s, c, s0, c0; fast_sincos(x, &s, &c); sincos(x, &s0, &c0);
    double es = fabs(s - s0), ec = fabs(c - c0);
    if (es > ws) ws = es; if (ec > wc) wc = ec;
  }
  printf("k pi/4 +- ulps: max abs err sin %.3e cos %.3e\n", ws, wc);
  double s, c; fast_sincos(2e15, &s, &c); printf("2e15 -> %g %g\n", s, c);
  fast_sincos(INFINITY, &s, &c); printf("inf -> %g %g\n", s, c);
  fast_sincos(NAN, &s, &c); printf("nan -> %g %g\n", s, c);
  fast_sincos(0.0, &s, &c); printf("0 -> %g %g\n", s, c);
  fast_sincos(-0.0, &s, &c); printf("-0 -> %g %g\n", s, c);
  return 0;
}

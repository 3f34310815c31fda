#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
static void fast_sincos(double x, double* so, double* co) {
  const double n = rint(x * 0.63661977236758138);
  double r = fma(-n, 1.5707963267948966, x);
  r = fma(-n, 6.123233995736766e-17, r);
  r = fma(-n, -1.4973849048591698e-33, r);
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
  const int q = (int)fma(-4.0, rint(n * 0.25), n);
  double s = (q & 1) ? pc : ps, c = (q & 1) ? ps : pc;
  *so = (q & 2) ? -s : s; *co = ((q + 1) & 2) ? -c : c;
}
int main() {
  srand(1);
  double scales[] = {1, 10, 1000, 1e6, 5e8, 1e12, 1e15};
  for (int k = 0; k < 7; ++k) {
    double ws = 0, wc = 0;
    for (int i = 0; i < 2000000; ++i) {
      double x = ((double)rand() / RAND_MAX * 2 - 1) * scales[k];
      double s, c, s0, c0; fast_sincos(x, &s, &c); sincos(x, &s0, &c0);
      double es = fabs(s - s0), ec = fabs(c - c0);
      if (es > ws) ws = es; if (ec > wc) wc = ec;
    }
    printf("scale %g: max abs err sin %.3e cos %.3e\n", scales[k], ws, wc);
  }
  double s, c; fast_sincos(2e15, &s, &c); printf("2e15 -> %g %g\n", s, c);
  fast_sincos(INFINITY, &s, &c); printf("inf -> %g %g\n", s, c);
  return 0;
}

// count_flops.cpp -- algorithmic flops per eval, COUNTED: the reference-structured CPU
// restatement (oracle/brbd_oracle.c) is compiled once more with `double` replaced by a counting
// scalar, and every add / sub / mul / div it performs on one instance is tallied (an fma would be
// its mul + its add = 2, as 8d asks; unary minus, copies and comparisons are free).  sincos and sqrt
// calls are counted separately.  The figures are frozen in BASELINE.md section 3 and used by bench.py.
//
// TEST / MEASUREMENT INFRASTRUCTURE: this links the oracle, so it lives under tools/ beside the
// other harness programs and is never part of the product.
//
//   g++ -O1 -fpermissive -w -o count_flops count_flops.cpp && ./count_flops model.txt
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double real_t;
static long long g_add = 0, g_mul = 0, g_div = 0, g_sincos = 0, g_sqrt = 0;

struct cnt {
  real_t x;
  cnt() = default;
  cnt(real_t v) : x(v) {}
  cnt(int v) : x(v) {}
  explicit operator real_t() const { return x; }
  cnt& operator+=(const cnt& o) { ++g_add; x += o.x; return *this; }
  cnt& operator-=(const cnt& o) { ++g_add; x -= o.x; return *this; }
  cnt& operator*=(const cnt& o) { ++g_mul; x *= o.x; return *this; }
  cnt& operator/=(const cnt& o) { ++g_div; x /= o.x; return *this; }
  cnt operator-() const { return cnt(-x); }
};
static inline cnt operator+(const cnt& a, const cnt& b) { ++g_add; return cnt(a.x + b.x); }
static inline cnt operator-(const cnt& a, const cnt& b) { ++g_add; return cnt(a.x - b.x); }
static inline cnt operator*(const cnt& a, const cnt& b) { ++g_mul; return cnt(a.x * b.x); }
static inline cnt operator/(const cnt& a, const cnt& b) { ++g_div; return cnt(a.x / b.x); }
static inline bool operator<(const cnt& a, const cnt& b) { return a.x < b.x; }
static inline bool operator>(const cnt& a, const cnt& b) { return a.x > b.x; }
static inline bool operator<=(const cnt& a, const cnt& b) { return a.x <= b.x; }
static inline bool operator>=(const cnt& a, const cnt& b) { return a.x >= b.x; }
static inline bool operator==(const cnt& a, const cnt& b) { return a.x == b.x; }
static inline bool operator!=(const cnt& a, const cnt& b) { return a.x != b.x; }
static inline void sincos(const cnt& a, cnt* s, cnt* c) { ++g_sincos; s->x = std::sin(a.x); c->x = std::cos(a.x); }
static inline cnt sqrt(const cnt& a) { ++g_sqrt; return cnt(std::sqrt(a.x)); }
static inline cnt fabs(const cnt& a) { return cnt(std::fabs(a.x)); }

#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#define double cnt
#include "../../oracle/brbd_oracle.c"
#undef double

struct Counts { long long add, mul, div, sc, sq; };
static Counts snap() { return Counts{g_add, g_mul, g_div, g_sincos, g_sqrt}; }
static void report(const char* name, Counts a, Counts b) {
  const long long ad = b.add - a.add, mu = b.mul - a.mul, di = b.div - a.div;
  printf("\"%s\": {\"flops\": %lld, \"add\": %lld, \"mul\": %lld, \"div\": %lld, \"sincos\": %lld, \"sqrt\": %lld}",
         name, ad + mu + di, ad, mu, di, b.sc - a.sc, b.sq - a.sq);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "r");
  if (!f) return 2;
  int n, nq, nv;
  if (fscanf(f, "%d %d %d", &n, &nq, &nv) != 3) return 2;
  std::vector<int32_t> parents(n), jtype(n), idx_q(n), idx_v(n);
  std::vector<cnt> placement(12 * n), inertia(10 * n), axis(3 * n), arm(nv > 0 ? nv : 1), q(nq), v(nv), a(nv);
  auto rd_i = [&](std::vector<int32_t>& x) { for (auto& e : x) { int t; if (fscanf(f, "%d", &t) != 1) exit(2); e = t; } };
  auto rd_d = [&](std::vector<cnt>& x, int k) { for (int i = 0; i < k; ++i) { real_t t; if (fscanf(f, "%lf", &t) != 1) exit(2); x[i].x = t; } };
  rd_i(parents); rd_i(jtype); rd_i(idx_q); rd_i(idx_v);
  rd_d(placement, 12 * n); rd_d(inertia, 10 * n); rd_d(axis, 3 * n); rd_d(arm, nv);
  brbd_model_desc d;
  memset(&d, 0, sizeof d);
  d.njoints = n; d.nq = nq; d.nv = nv;
  d.parents = parents.data(); d.jtype = jtype.data(); d.idx_q = idx_q.data(); d.idx_v = idx_v.data();
  d.placement = placement.data(); d.inertia = inertia.data(); d.axis = axis.data(); d.armature = arm.data();
  real_t g[3];
  if (fscanf(f, "%lf %lf %lf", &g[0], &g[1], &g[2]) != 3) return 2;
  for (int k = 0; k < 3; ++k) d.gravity[k].x = g[k];
  rd_d(q, nq); rd_d(v, nv); rd_d(a, nv);
  fclose(f);
  std::vector<cnt> tau(nv), ddq(nv), J(6 * nv), oMi(12 * n), dq(nv * nv), dv(nv * nv), da(nv * nv);
  printf("{");
  Counts c0 = snap();
  if (brbd_oracle_rnea(&d, q.data(), v.data(), a.data(), tau.data())) return 3;
  Counts c1 = snap(); report("rnea", c0, c1); printf(", ");
  if (brbd_oracle_aba_world(&d, q.data(), v.data(), tau.data(), ddq.data())) return 3;
  Counts c2 = snap(); report("aba_world", c1, c2); printf(", ");
  if (brbd_oracle_aba_local(&d, q.data(), v.data(), tau.data(), ddq.data())) return 3;
  Counts c3 = snap(); report("aba_local", c2, c3); printf(", ");
  if (brbd_oracle_fk_jacobians(&d, q.data(), J.data(), oMi.data())) return 3;
  Counts c4 = snap(); report("fk_jacobians", c3, c4);
  Counts c5 = snap();
  if (brbd_oracle_rnea_derivatives(&d, q.data(), v.data(), a.data(), dq.data(), dv.data(), da.data(), tau.data()) == 0) {
    c5 = snap(); printf(", "); report("rnea_derivatives", c4, c5);
  }
  printf("}\n");
  return 0;
}

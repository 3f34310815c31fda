// A C++ caller of include/brbd_parallel.hpp that produces numbers: brbd::rneaInParallel /
// brbd::abaInParallel on column-major host matrices (rows / cols / data / outerStride, the
// interface of Eigen::MatrixXd, Eigen::Map and Eigen::Ref), as a user of the reference calls
// pinocchio::rneaInParallel (algorithm/parallel/rnea.hpp:39-83).  Input: a text file written by
// tests/test_gpu_cpp_caller.py with the model tables, a batch (q, v, a) and the expected tau with
// the tolerance of each column (the reference's unittest/value.cpp vectors and the golden vectors).
// Checks, per column with Eigen isApprox semantics: tau == expected; aba(q, v, tau) == a
// (unittest/aba.cpp:143-154); the same through strided views (outerStride > rows) over exactly
// sized storage whose padding must stay untouched.  Exit code 0 = all good.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "brbd_parallel.hpp"

struct Mat {  // owning column-major matrix, like Eigen::MatrixXd
  long r, c;
  std::vector<double> d;
  Mat(long r_, long c_, double fill = 0.0) : r(r_), c(c_), d((size_t)(r_ * c_), fill) {}
  long rows() const { return r; }
  long cols() const { return c; }
  long outerStride() const { return r; }
  const double* data() const { return d.data(); }
  double& operator()(long i, long j) { return d[(size_t)(j * r + i)]; }
  double operator()(long i, long j) const { return d[(size_t)(j * r + i)]; }
};

struct View {  // non-owning strided view, like Eigen::Map<MatrixXd, 0, OuterStride<>>
  long r, c, ld;
  double* p;
  long rows() const { return r; }
  long cols() const { return c; }
  long outerStride() const { return ld; }
  const double* data() const { return p; }
};

static double col_rel_err(const double* x, const double* y, long n) {  // ||x - y|| / min(||x||, ||y||)
  double num = 0, nx = 0, ny = 0;
  for (long i = 0; i < n; ++i) {
    num += (x[i] - y[i]) * (x[i] - y[i]);
    nx += x[i] * x[i];
    ny += y[i] * y[i];
  }
  const double den = std::sqrt(nx < ny ? nx : ny);
  return den > 0 ? std::sqrt(num) / den : std::sqrt(num);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "r");
  if (!f) return 2;
  auto rd = [&](double& x) { if (std::fscanf(f, "%lf", &x) != 1) { std::fprintf(stderr, "short input\n"); std::exit(2); } };
  auto rdi = [&](int32_t& x) { int t; if (std::fscanf(f, "%d", &t) != 1) { std::fprintf(stderr, "short input\n"); std::exit(2); } x = t; };
  brbd::ModelDesc m;
  int32_t n, nq, nv, B;
  rdi(n); rdi(nq); rdi(nv); rdi(B);
  m.njoints = n; m.nq = nq; m.nv = nv;
  m.parents.assign(n, 0); m.jtype.assign(n, 0); m.idx_q.assign(n, 0); m.idx_v.assign(n, 0);
  for (auto* t : {&m.parents, &m.jtype, &m.idx_q, &m.idx_v})
    for (auto& e : *t) rdi(e);
  m.placement.assign(12 * (size_t)n, 0.0); m.inertia.assign(10 * (size_t)n, 0.0); m.axis.assign(3 * (size_t)n, 0.0);
  m.armature.assign((size_t)nv, 0.0);
  for (auto* t : {&m.placement, &m.inertia, &m.axis, &m.armature})
    for (auto& e : *t) rd(e);
  for (int k = 0; k < 3; ++k) rd(m.gravity[k]);
  Mat q(nq, B), v(nv, B), a(nv, B), expected(nv, B), tau(nv, B, NAN), ddq(nv, B, NAN);
  std::vector<double> tol((size_t)B);
  for (auto* t : {&q.d, &v.d, &a.d, &expected.d, &tol})
    for (auto& e : *t) rd(e);
  std::fclose(f);

  int fails = 0;
  try {
    brbd::ModelPool pool(m, 4);
    brbd::rneaInParallel(4, pool, q, v, a, tau);
    double worst_r = 0, worst_a = 0;
    for (long b = 0; b < B; ++b) {
      const double e = col_rel_err(&tau.d[(size_t)(b * nv)], &expected.d[(size_t)(b * nv)], nv);
      if (!(e <= tol[(size_t)b])) { ++fails; std::fprintf(stderr, "rnea column %ld: rel err %.3e > %.1e\n", b, e, tol[(size_t)b]); }
      if (e / tol[(size_t)b] > worst_r) worst_r = e / tol[(size_t)b];
    }
    brbd::abaInParallel(4, pool, q, v, tau, ddq);
    for (long b = 0; b < B; ++b) {
      double na = 0;
      for (long i = 0; i < nv; ++i) na += a(i, b) * a(i, b);
      if (na == 0) continue;  // (a = 0 columns: isApprox against zero is meaningless)
      const double e = col_rel_err(&ddq.d[(size_t)(b * nv)], &a.d[(size_t)(b * nv)], nv);
      if (!(e <= 1e-9)) { ++fails; std::fprintf(stderr, "aba column %ld: rel err %.3e\n", b, e); }
      if (e > worst_a) worst_a = e;
    }
    // strided views over exactly sized storage: padding rows keep their canary
    const long pad = 3;
    auto strided = [&](const Mat& src, std::vector<double>& store) {
      const long ld = src.r + pad;
      store.assign((size_t)((src.c - 1) * ld + src.r), -1.5e300);
      for (long j = 0; j < src.c; ++j)
        for (long i = 0; i < src.r; ++i) store[(size_t)(j * ld + i)] = src(i, j);
      return View{src.r, src.c, ld, store.data()};
    };
    std::vector<double> sq, sv, sa, st;
    Mat nanm(nv, B, NAN);
    const View vq = strided(q, sq), vv = strided(v, sv), va = strided(a, sa), vt = strided(nanm, st);
    brbd::rneaInParallel(1, pool, vq, vv, va, vt);
    for (long j = 0; j < B; ++j) {
      for (long i = 0; i < nv; ++i)
        if (st[(size_t)(j * vt.ld + i)] != tau(i, j)) { ++fails; std::fprintf(stderr, "strided tau differs at (%ld, %ld)\n", i, j); i = nv; }
      if (j + 1 < B)
        for (long i = nv; i < vt.ld; ++i)
          if (st[(size_t)(j * vt.ld + i)] != -1.5e300) { ++fails; std::fprintf(stderr, "padding overwritten at column %ld\n", j); i = vt.ld; }
    }
    // one process, several devices: the column split over a MultiDevicePool (here: two handles on
    // GPU 0, one host thread each) gives the same numbers as the single-device call
    {
      brbd::MultiDevicePool multi(m, 4, {0, 0});
      Mat tau2(nv, B, NAN), ddq2(nv, B, NAN);
      brbd::rneaInParallel(2, multi, q, v, a, tau2);
      brbd::abaInParallel(2, multi, q, v, tau2, ddq2);
      for (size_t k = 0; k < tau.d.size(); ++k)
        if (tau2.d[k] != tau.d[k] || ddq2.d[k] != ddq.d[k]) { ++fails; std::fprintf(stderr, "multi-device result differs at %zu\n", k); break; }
    }
    // argument checks still throw std::invalid_argument before any device work
    try {
      brbd::rneaInParallel(5, pool, q, v, a, tau);
      ++fails;
    } catch (const std::invalid_argument&) {
    }
    std::printf("cpp caller: %d instances, rnea worst err/tol %.3g, aba worst rel err %.3g, fails %d\n", (int)B, worst_r,
                worst_a, fails);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
  return fails ? 1 : 0;
}

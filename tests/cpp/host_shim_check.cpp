// Compile-and-run check of include/brbd_parallel.hpp with a minimal Eigen-like matrix (no GPU is
// needed: only the argument checks that must throw BEFORE any device work are exercised, as the
// reference's PINOCCHIO_CHECK_* do, algorithm/parallel/rnea.hpp:53-68).
#include <cstdio>
#include <vector>

#include "brbd_parallel.hpp"

struct Mat {  // column-major, like Eigen::MatrixXd
  long r, c;
  std::vector<double> d;
  Mat(long r_, long c_) : r(r_), c(c_), d((size_t)(r_ * c_), 0.0) {}
  long rows() const { return r; }
  long cols() const { return c; }
  long outerStride() const { return r; }
  const double* data() const { return d.data(); }
};

template <class F>
static bool throws_invalid(F f) {
  try {
    f();
  } catch (const std::invalid_argument&) {
    return true;
  } catch (...) {
    return false;
  }
  return false;
}

int main() {
  brbd::ModelDesc m;  // single RZ joint
  m.njoints = 2; m.nq = 1; m.nv = 1;
  m.parents = {0, 0}; m.jtype = {0, BRBD_JOINT_RZ}; m.idx_q = {0, 0}; m.idx_v = {0, 0};
  m.placement.assign(24, 0.0);
  m.placement[12] = m.placement[16] = m.placement[20] = 1.0;
  m.inertia.assign(20, 0.0);
  m.inertia[10] = 1.0; m.inertia[14] = m.inertia[16] = m.inertia[19] = 0.1;
  m.axis.assign(6, 0.0);
  Mat q(1, 4), v(1, 4), a(1, 4), tau(1, 4), bad(2, 4), few(1, 3);
  int fails = 0;
  {
    brbd::ModelPool empty(m, 0);
    fails += !throws_invalid([&] { brbd::rneaInParallel(1, empty, q, v, a, tau); });
  }
  brbd::ModelPool pool(m, 2);
  fails += !throws_invalid([&] { brbd::rneaInParallel(3, pool, q, v, a, tau); });
  fails += !throws_invalid([&] { brbd::rneaInParallel(1, pool, bad, v, a, tau); });
  fails += !throws_invalid([&] { brbd::abaInParallel(1, pool, q, few, a, tau); });
  fails += !throws_invalid([&] { brbd::abaInParallel(1, pool, q, v, bad, a); });
  fails += !throws_invalid([&] { pool.getModel(7); });
  // external forces / joint Jacobian wrappers: (6 * njoints) x B and (6 * nv) x B matrices
  Mat fext(12, 4), fbad(6, 4), J(6, 4), Jbad(5, 4);
  fails += !throws_invalid([&] { brbd::rnea(pool, q, v, a, fbad, tau); });
  fails += !throws_invalid([&] { brbd::aba(pool, q, v, tau, fext, few); });
  fails += !throws_invalid([&] { brbd::getJointJacobian(pool, q, 1, 1, Jbad); });
  fails += !throws_invalid([&] { brbd::rnea(pool, bad, v, a, fext, tau); });
  (void)J;
  std::printf("host shim checks: %s\n", fails ? "FAILED" : "ok");
  return fails;
}

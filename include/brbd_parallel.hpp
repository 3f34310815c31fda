// brbd_parallel.hpp -- header-only C++ host mirror of Pinocchio's parallel entry points on top of
// the C ABI (include/brbd.h).  Same names, argument order and error behaviour as
//   pinocchio::rneaInParallel  (include/pinocchio/algorithm/parallel/rnea.hpp:31-83)
//   pinocchio::abaInParallel   (include/pinocchio/algorithm/parallel/aba.hpp:32-84)
//   pinocchio::ModelPoolTpl    (include/pinocchio/multibody/pool/model.hpp:19-165)
// so that a caller switches by changing the namespace.  Matrices are any column-major type with
// rows(), cols(), data() and outerStride() (Eigen::MatrixXd, Eigen::Map, Eigen::Ref ...): one
// instance per contiguous column = BRBD_INSTANCE_MAJOR with ld = outerStride(), host memory.
//
// When Pinocchio's headers are visible, brbd::make_desc(pinocchio::Model) flattens a real model
// (fields of SURVEY 8a row a13; joint kind from shortname(), e.g. joint-revolute.hpp:848-851).
#pragma once
#include <cstring>
#include <exception>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "brbd.h"

namespace brbd {

// Owning, plain-table description of a model (what brbd_model_desc points into).
struct ModelDesc {
  int njoints = 1, nq = 0, nv = 0;
  std::vector<int32_t> parents{0}, jtype{0}, idx_q{0}, idx_v{0};
  std::vector<double> placement = std::vector<double>(12, 0.0), inertia = std::vector<double>(10, 0.0),
                      axis = std::vector<double>(3, 0.0), armature;
  double gravity[3] = {0.0, 0.0, -9.81};

  brbd_model_desc view() const {
    brbd_model_desc d;
    d.njoints = njoints; d.nq = nq; d.nv = nv;
    d.parents = parents.data(); d.jtype = jtype.data(); d.idx_q = idx_q.data(); d.idx_v = idx_v.data();
    d.placement = placement.data(); d.inertia = inertia.data(); d.axis = axis.data();
    d.armature = armature.empty() ? nullptr : armature.data();
    for (int k = 0; k < 3; ++k) d.gravity[k] = gravity[k];
    return d;
  }
};

inline void throw_status(int rc) {
  if (rc == BRBD_OK) return;
  const std::string msg = brbd_last_error();
  if (rc == BRBD_ERR_INVALID_ARGUMENT || rc == BRBD_ERR_UNSUPPORTED) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

// Counterpart of ModelPoolTpl: `size` replicas in the reference, one flattened BatchModel per GPU
// here; size() only feeds the reference's argument checks.
class ModelPool {
 public:
  explicit ModelPool(const ModelDesc& model, size_t pool_size = 1, int device = 0)
      : desc_(model), size_(pool_size), device_(device) {}
  ~ModelPool() { release(); }
  ModelPool(const ModelPool&) = delete;
  ModelPool& operator=(const ModelPool&) = delete;

  size_t size() const { return size_; }
  void resize(size_t n) { size_ = n; }
  const ModelDesc& getModel(size_t index = 0) const {
    if (index >= (size_ ? size_ : 1)) throw std::invalid_argument("Index greater than the size of the model vector.");
    return desc_;
  }
  void update(const ModelDesc& model) {  // pool/model.hpp:68-73
    desc_ = model;
    release();
  }
  brbd_model* handle() {
    if (!h_) {
      const brbd_model_desc d = desc_.view();
      throw_status(brbd_model_create(&d, device_, &h_));
    }
    return h_;
  }
  void release() {
    if (h_) brbd_model_destroy(h_);
    h_ = nullptr;
  }

 private:
  ModelDesc desc_;
  size_t size_;
  int device_;
  brbd_model* h_ = nullptr;
};

namespace detail {
inline void check(bool ok, const char* msg) {
  if (!ok) throw std::invalid_argument(msg);  // PINOCCHIO_CHECK_INPUT_ARGUMENT, macros.hpp:201-215
}
}  // namespace detail

// tau = rnea(q, v, a) column by column; `tau` is the LAST argument and is written through a const
// reference exactly like the reference does with const_cast_derived() (parallel/rnea.hpp:59).
template <class ConfigPool, class Tangent1, class Tangent2, class Tangent3>
void rneaInParallel(const size_t num_threads, ModelPool& pool, const ConfigPool& q, const Tangent1& v,
                    const Tangent2& a, const Tangent3& tau) {
  detail::check(pool.size() > 0, "The pool should have at least one element");  // rnea.hpp:53
  detail::check(num_threads <= pool.size(), "The pool is too small");            // rnea.hpp:54
  const ModelDesc& m = pool.getModel(0);
  detail::check(q.rows() == m.nq, "q.rows() != model.nq");                       // rnea.hpp:61-68
  detail::check(v.rows() == m.nv && a.rows() == m.nv && tau.rows() == m.nv, "rows != model.nv");
  detail::check(q.cols() == v.cols() && q.cols() == a.cols() && q.cols() == tau.cols(), "column mismatch");
  throw_status(brbd_rnea(pool.handle(), (int64_t)q.cols(), q.data(), (int64_t)q.outerStride(), v.data(),
                         (int64_t)v.outerStride(), a.data(), (int64_t)a.outerStride(),
                         const_cast<double*>(tau.data()), (int64_t)tau.outerStride(), BRBD_INSTANCE_MAJOR,
                         BRBD_HOST, nullptr));
}

// a = aba(q, v, tau, Convention::WORLD) column by column (parallel/aba.hpp:40-84).
template <class ConfigPool, class Tangent1, class Tangent2, class Tangent3>
void abaInParallel(const size_t num_threads, ModelPool& pool, const ConfigPool& q, const Tangent1& v,
                   const Tangent2& tau, const Tangent3& a) {
  detail::check(pool.size() > 0, "The pool should have at least one element");
  detail::check(num_threads <= pool.size(), "The pool is too small");
  const ModelDesc& m = pool.getModel(0);
  detail::check(q.rows() == m.nq, "q.rows() != model.nq");
  detail::check(v.rows() == m.nv && a.rows() == m.nv && tau.rows() == m.nv, "rows != model.nv");
  detail::check(q.cols() == v.cols() && q.cols() == a.cols() && q.cols() == tau.cols(), "column mismatch");
  throw_status(brbd_aba(pool.handle(), (int64_t)q.cols(), q.data(), (int64_t)q.outerStride(), v.data(),
                        (int64_t)v.outerStride(), tau.data(), (int64_t)tau.outerStride(),
                        const_cast<double*>(a.data()), (int64_t)a.outerStride(), BRBD_INSTANCE_MAJOR,
                        BRBD_HOST, nullptr));
}

// ---- one process, several GPUs ---------------------------------------------------------------------
// The static column split of the reference's OpenMP loop (parallel/rnea.hpp:74-82) with one host
// thread + one device model per GPU: slice k = columns [k B / G, (k + 1) B / G) runs on devices[k].
// No collective and no peer copy: every slice is staged by its own device.
class MultiDevicePool {
 public:
  MultiDevicePool(const ModelDesc& model, size_t pool_size, const std::vector<int>& devices) : size_(pool_size) {
    if (devices.empty()) throw std::invalid_argument("devices must name at least one GPU");
    for (int d : devices) pools_.emplace_back(new ModelPool(model, pool_size, d));
  }
  size_t size() const { return size_; }
  size_t devices() const { return pools_.size(); }
  ModelPool& on(size_t k) { return *pools_[k]; }

 private:
  size_t size_;
  std::vector<std::unique_ptr<ModelPool>> pools_;
};

namespace detail {
struct ColumnSlice {  // a block of columns of a column-major matrix: same outerStride, shifted data
  long r, c, ld;
  const double* p;
  long rows() const { return r; }
  long cols() const { return c; }
  long outerStride() const { return ld; }
  const double* data() const { return p; }
};
template <class Mat>
ColumnSlice columns(const Mat& m, long lo, long hi) {
  return ColumnSlice{(long)m.rows(), hi - lo, (long)m.outerStride(), m.data() + lo * (long)m.outerStride()};
}
template <class F>
void on_every_device(MultiDevicePool& pool, long B, F f) {
  const long G = (long)pool.devices();
  std::vector<std::thread> threads;
  std::vector<std::exception_ptr> errs((size_t)G);
  for (long k = 0; k < G; ++k)
    threads.emplace_back([&, k] {
      try {
        const long lo = k * B / G, hi = (k + 1) * B / G;
        if (hi > lo) f(pool.on((size_t)k), lo, hi);
      } catch (...) {
        errs[(size_t)k] = std::current_exception();
      }
    });
  for (auto& t : threads) t.join();
  for (auto& e : errs)
    if (e) std::rethrow_exception(e);
}
}  // namespace detail

template <class ConfigPool, class Tangent1, class Tangent2, class Tangent3>
void rneaInParallel(const size_t num_threads, MultiDevicePool& pool, const ConfigPool& q, const Tangent1& v,
                    const Tangent2& a, const Tangent3& tau) {
  detail::check(pool.size() > 0, "The pool should have at least one element");
  detail::check(num_threads <= pool.size(), "The pool is too small");
  detail::check(q.cols() == v.cols() && q.cols() == a.cols() && q.cols() == tau.cols(), "column mismatch");
  detail::on_every_device(pool, (long)q.cols(), [&](ModelPool& p, long lo, long hi) {
    rneaInParallel(num_threads, p, detail::columns(q, lo, hi), detail::columns(v, lo, hi), detail::columns(a, lo, hi),
                   detail::columns(tau, lo, hi));
  });
}
template <class ConfigPool, class Tangent1, class Tangent2, class Tangent3>
void abaInParallel(const size_t num_threads, MultiDevicePool& pool, const ConfigPool& q, const Tangent1& v,
                   const Tangent2& tau, const Tangent3& a) {
  detail::check(pool.size() > 0, "The pool should have at least one element");
  detail::check(num_threads <= pool.size(), "The pool is too small");
  detail::check(q.cols() == v.cols() && q.cols() == a.cols() && q.cols() == tau.cols(), "column mismatch");
  detail::on_every_device(pool, (long)q.cols(), [&](ModelPool& p, long lo, long hi) {
    abaInParallel(num_threads, p, detail::columns(q, lo, hi), detail::columns(v, lo, hi), detail::columns(tau, lo, hi),
                  detail::columns(a, lo, hi));
  });
}

// Batched rnea(model, data, q, v, a, fext) / aba(model, data, q, v, tau, fext) (algorithm/rnea.hxx:171-219,
// algorithm/aba.hxx:296-354, WORLD convention): `fext` is (6 * njoints) x B, column b holding the
// reference's aligned_vector<Force> of instance b, entry i = [linear; angular] in joint i's frame.
template <class ConfigPool, class Tangent1, class Tangent2, class ForcePool, class Tangent3>
void rnea(ModelPool& pool, const ConfigPool& q, const Tangent1& v, const Tangent2& a, const ForcePool& fext,
          const Tangent3& tau) {
  detail::check(pool.size() > 0, "The pool should have at least one element");
  const ModelDesc& m = pool.getModel(0);
  detail::check(q.rows() == m.nq, "q.rows() != model.nq");
  detail::check(v.rows() == m.nv && a.rows() == m.nv && tau.rows() == m.nv, "rows != model.nv");
  detail::check(fext.rows() == 6 * m.njoints, "fext.rows() != 6 * model.njoints");  // rnea.hxx:179
  detail::check(q.cols() == v.cols() && q.cols() == a.cols() && q.cols() == tau.cols() && q.cols() == fext.cols(),
                "column mismatch");
  throw_status(brbd_rnea_fext(pool.handle(), (int64_t)q.cols(), q.data(), (int64_t)q.outerStride(), v.data(),
                              (int64_t)v.outerStride(), a.data(), (int64_t)a.outerStride(), fext.data(),
                              (int64_t)fext.outerStride(), const_cast<double*>(tau.data()),
                              (int64_t)tau.outerStride(), BRBD_INSTANCE_MAJOR, BRBD_HOST, nullptr));
}

template <class ConfigPool, class Tangent1, class Tangent2, class ForcePool, class Tangent3>
void aba(ModelPool& pool, const ConfigPool& q, const Tangent1& v, const Tangent2& tau, const ForcePool& fext,
         const Tangent3& a) {
  detail::check(pool.size() > 0, "The pool should have at least one element");
  const ModelDesc& m = pool.getModel(0);
  detail::check(q.rows() == m.nq, "q.rows() != model.nq");
  detail::check(v.rows() == m.nv && a.rows() == m.nv && tau.rows() == m.nv, "rows != model.nv");
  detail::check(fext.rows() == 6 * m.njoints, "fext.rows() != 6 * model.njoints");
  detail::check(q.cols() == v.cols() && q.cols() == a.cols() && q.cols() == tau.cols() && q.cols() == fext.cols(),
                "column mismatch");
  throw_status(brbd_aba_fext(pool.handle(), (int64_t)q.cols(), q.data(), (int64_t)q.outerStride(), v.data(),
                             (int64_t)v.outerStride(), tau.data(), (int64_t)tau.outerStride(), fext.data(),
                             (int64_t)fext.outerStride(), const_cast<double*>(a.data()),
                             (int64_t)a.outerStride(), BRBD_INSTANCE_MAJOR, BRBD_HOST, nullptr));
}

// Batched computeJointJacobians(q) + getJointJacobian(joint_id, rf, J) (algorithm/jacobian.hxx:175-283):
// J is (6 * nv) x B, column b = the 6 x nv Jacobian of instance b, column-major; rf as
// pinocchio::ReferenceFrame (0 WORLD, 1 LOCAL, 2 LOCAL_WORLD_ALIGNED).
template <class ConfigPool, class JacobianPool>
void getJointJacobian(ModelPool& pool, const ConfigPool& q, int joint_id, int rf, const JacobianPool& J) {
  detail::check(pool.size() > 0, "The pool should have at least one element");
  const ModelDesc& m = pool.getModel(0);
  detail::check(q.rows() == m.nq, "q.rows() != model.nq");
  detail::check(J.rows() == 6 * m.nv, "J.rows() != 6 * model.nv");
  detail::check(q.cols() == J.cols(), "column mismatch");
  throw_status(brbd_joint_jacobian(pool.handle(), (int64_t)q.cols(), q.data(), (int64_t)q.outerStride(), joint_id, rf,
                                   const_cast<double*>(J.data()), (int64_t)J.outerStride(), BRBD_INSTANCE_MAJOR,
                                   BRBD_HOST, nullptr));
}

}  // namespace brbd

// ------------------------------------------------------------------------------------------------
// Optional adapter, compiled only where Pinocchio (and therefore Eigen/Boost) is installed.
#if defined(__has_include)
#if __has_include(<pinocchio/multibody/model.hpp>)
#include <pinocchio/multibody/model.hpp>
namespace brbd {
inline int joint_type_from_shortname(const std::string& s) {
  static const char* names[] = {"JointModelRX", "JointModelRY", "JointModelRZ", "JointModelPX", "JointModelPY",
                                "JointModelPZ", "JointModelFreeFlyer", "JointModelSpherical",
                                "JointModelRevoluteUnaligned", "JointModelPrismaticUnaligned", "JointModelRUBX",
                                "JointModelRUBY", "JointModelRUBZ", "JointModelRevoluteUnboundedUnaligned"};
  static_assert(sizeof(names) / sizeof(names[0]) == BRBD_JOINT_TYPE_COUNT, "one shortname per brbd_joint_type");
  for (int k = 0; k < BRBD_JOINT_TYPE_COUNT; ++k)
    if (s == names[k]) return k;
  throw std::invalid_argument("joint kind " + s + " is not supported by the brbd engine");
}
// axis of the unaligned joint kinds (public member, e.g. joint-revolute-unaligned.hpp:734)
template <class JointModelKind>
inline bool axis_of(const pinocchio::JointModel& joint, double* out) {
  if (const JointModelKind* p = boost::get<JointModelKind>(&joint.toVariant())) {
    for (int k = 0; k < 3; ++k) out[k] = p->axis[k];
    return true;
  }
  return false;
}
inline ModelDesc make_desc(const pinocchio::Model& model) {
  ModelDesc d;
  const int n = model.njoints;
  d.njoints = n; d.nq = model.nq; d.nv = model.nv;
  d.parents.assign(n, 0); d.jtype.assign(n, 0); d.idx_q.assign(n, 0); d.idx_v.assign(n, 0);
  d.placement.assign(12 * (size_t)n, 0.0); d.inertia.assign(10 * (size_t)n, 0.0); d.axis.assign(3 * (size_t)n, 0.0);
  d.armature.assign(model.armature.data(), model.armature.data() + model.nv);
  for (int k = 0; k < 3; ++k) d.gravity[k] = model.gravity.linear()[k];
  for (int i = 1; i < n; ++i) {
    d.parents[i] = (int32_t)model.parents[(size_t)i];
    d.jtype[i] = joint_type_from_shortname(model.joints[(size_t)i].shortname());
    d.idx_q[i] = model.idx_qs[(size_t)i];
    d.idx_v[i] = model.idx_vs[(size_t)i];
    const auto& P = model.jointPlacements[(size_t)i];
    std::memcpy(&d.placement[12 * (size_t)i], P.rotation().data(), 9 * sizeof(double));  // column-major
    std::memcpy(&d.placement[12 * (size_t)i + 9], P.translation().data(), 3 * sizeof(double));
    const auto& Y = model.inertias[(size_t)i];
    d.inertia[10 * (size_t)i] = Y.mass();
    std::memcpy(&d.inertia[10 * (size_t)i + 1], Y.lever().data(), 3 * sizeof(double));
    std::memcpy(&d.inertia[10 * (size_t)i + 4], Y.inertia().data().data(), 6 * sizeof(double));
    double* ax = &d.axis[3 * (size_t)i];
    (void)(axis_of<pinocchio::JointModelRevoluteUnaligned>(model.joints[(size_t)i], ax) ||
           axis_of<pinocchio::JointModelPrismaticUnaligned>(model.joints[(size_t)i], ax) ||
           axis_of<pinocchio::JointModelRevoluteUnboundedUnaligned>(model.joints[(size_t)i], ax));
  }
  return d;
}
}  // namespace brbd
#endif

// ------------------------------------------------------------------------------------------------
// Drop-in overloads on Pinocchio's OWN pool type (multibody/pool/model.hpp:19-165): a caller keeps
//     pinocchio::ModelPool pool(model, n);  brbd::rneaInParallel(n, pool, q, v, a, tau);
// and only the namespace of the call changes.  The flattened BatchModel lives in a per-pool device
// cache keyed by the pool's address (SURVEY 8b "Ownership"); Pinocchio has no change notification,
// so every call re-flattens pool.getModel(0) (a few KB of tables) and rebuilds the device model
// when any table differs; brbd::forget(pool) drops the entry (call it before the pool dies).
// NOT COMPILED in the build container (Pinocchio, Eigen and Boost are absent there): written
// against the reference's headers, exercised only where they exist.
#if __has_include(<pinocchio/multibody/pool/model.hpp>)
#include <map>
#include <memory>
#include <mutex>
#include <type_traits>
#include <pinocchio/multibody/pool/model.hpp>
namespace brbd {
namespace detail {
struct PoolEntry {
  ModelDesc desc;
  std::unique_ptr<ModelPool> pool;
};
inline bool same_tables(const ModelDesc& a, const ModelDesc& b) {
  return a.njoints == b.njoints && a.nq == b.nq && a.nv == b.nv && a.parents == b.parents && a.jtype == b.jtype &&
         a.idx_q == b.idx_q && a.idx_v == b.idx_v && a.placement == b.placement && a.inertia == b.inertia &&
         a.axis == b.axis && a.armature == b.armature && a.gravity[0] == b.gravity[0] &&
         a.gravity[1] == b.gravity[1] && a.gravity[2] == b.gravity[2];
}
inline std::map<const void*, PoolEntry>& pool_cache() {
  static std::map<const void*, PoolEntry> c;
  return c;
}
inline std::mutex& pool_cache_mutex() {
  static std::mutex m;
  return m;
}
template <class PinPool>
inline ModelPool& device_pool(PinPool& pool) {
  PINOCCHIO_CHECK_INPUT_ARGUMENT(pool.size() > 0, "The pool should have at least one element");
  ModelDesc now = make_desc(pool.getModel(0));
  std::lock_guard<std::mutex> g(pool_cache_mutex());
  PoolEntry& e = pool_cache()[static_cast<const void*>(&pool)];
  if (!e.pool || !same_tables(e.desc, now)) {  // first use, or the model was changed / pool.update() called
    e.desc = now;
    e.pool.reset(new ModelPool(e.desc, (size_t)pool.size()));
  }
  e.pool->resize((size_t)pool.size());
  return *e.pool;
}
}  // namespace detail

template <typename Scalar, int Options, template <typename, int> class JointCollectionTpl>
inline void forget(const pinocchio::ModelPoolTpl<Scalar, Options, JointCollectionTpl>& pool) {
  std::lock_guard<std::mutex> g(detail::pool_cache_mutex());
  detail::pool_cache().erase(static_cast<const void*>(&pool));
}

// Argument adaptation.  The reference takes ANY Eigen::MatrixBase (parallel/rnea.hpp:39-45): plain
// matrices, blocks, maps with strides, row-major matrices, expressions such as 2 * q.  The C ABI
// wants column-major storage with unit inner stride (element (d, b) at x[b * ld + d]).  An argument
// that already is that (DirectAccessBit, not row-major unless it is a column vector, inner stride 1 at
// compile time) is passed through by reference; anything else is evaluated into a plain
// column-major matrix first (inputs), or computed into one and assigned back (the output).
namespace detail {
typedef Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::ColMajor> PlainColMajor;
template <class D>
struct passes_through {
  static const bool value = (int(Eigen::internal::traits<D>::Flags) & int(Eigen::DirectAccessBit)) != 0 &&
                            (!bool(D::IsRowMajor) || int(D::ColsAtCompileTime) == 1) &&
                            int(Eigen::internal::inner_stride_at_compile_time<D>::ret) == 1;
};
template <class D, bool Through = passes_through<D>::value>
struct InArg {  // input: the argument itself
  const D& m;
  explicit InArg(const Eigen::MatrixBase<D>& x) : m(x.derived()) {}
  const D& get() const { return m; }
};
template <class D>
struct InArg<D, false> {  // input: evaluated copy (also turns row-major storage into column-major)
  PlainColMajor m;
  explicit InArg(const Eigen::MatrixBase<D>& x) : m(x) {}
  const PlainColMajor& get() const { return m; }
};
template <class D, bool Through = passes_through<D>::value>
struct OutArg {  // output written in place, through const like the reference's const_cast_derived()
  D& m;
  explicit OutArg(const Eigen::MatrixBase<D>& x) : m(x.const_cast_derived()) {}
  D& get() { return m; }
  void commit() {}
};
template <class D>
struct OutArg<D, false> {  // output computed into a plain matrix, then assigned to the caller's expression
  D& target;
  PlainColMajor m;
  explicit OutArg(const Eigen::MatrixBase<D>& x) : target(x.const_cast_derived()), m(x.rows(), x.cols()) {}
  PlainColMajor& get() { return m; }
  void commit() { target = m; }
};
}  // namespace detail

// same template header as algorithm/parallel/rnea.hpp:31-45, for Scalar = double
template <int Options, template <typename, int> class JointCollectionTpl, typename ConfigVectorPool,
          typename TangentVectorPool1, typename TangentVectorPool2, typename TangentVectorPool3>
void rneaInParallel(const size_t num_threads, pinocchio::ModelPoolTpl<double, Options, JointCollectionTpl>& pool,
                    const Eigen::MatrixBase<ConfigVectorPool>& q, const Eigen::MatrixBase<TangentVectorPool1>& v,
                    const Eigen::MatrixBase<TangentVectorPool2>& a, const Eigen::MatrixBase<TangentVectorPool3>& tau) {
  static_assert(std::is_same<typename ConfigVectorPool::Scalar, double>::value &&
                    std::is_same<typename TangentVectorPool3::Scalar, double>::value,
                "the MI355X engine computes in double");
  detail::InArg<ConfigVectorPool> qe(q);
  detail::InArg<TangentVectorPool1> ve(v);
  detail::InArg<TangentVectorPool2> ae(a);
  detail::OutArg<TangentVectorPool3> te(tau);
  rneaInParallel(num_threads, detail::device_pool(pool), qe.get(), ve.get(), ae.get(), te.get());
  te.commit();
}
template <int Options, template <typename, int> class JointCollectionTpl, typename ConfigVectorPool,
          typename TangentVectorPool1, typename TangentVectorPool2, typename TangentVectorPool3>
void abaInParallel(const size_t num_threads, pinocchio::ModelPoolTpl<double, Options, JointCollectionTpl>& pool,
                   const Eigen::MatrixBase<ConfigVectorPool>& q, const Eigen::MatrixBase<TangentVectorPool1>& v,
                   const Eigen::MatrixBase<TangentVectorPool2>& tau, const Eigen::MatrixBase<TangentVectorPool3>& a) {
  static_assert(std::is_same<typename ConfigVectorPool::Scalar, double>::value &&
                    std::is_same<typename TangentVectorPool3::Scalar, double>::value,
                "the MI355X engine computes in double");
  detail::InArg<ConfigVectorPool> qe(q);
  detail::InArg<TangentVectorPool1> ve(v);
  detail::InArg<TangentVectorPool2> te(tau);
  detail::OutArg<TangentVectorPool3> ae(a);
  abaInParallel(num_threads, detail::device_pool(pool), qe.get(), ve.get(), te.get(), ae.get());
  ae.commit();
}
}  // namespace brbd
#endif
#endif

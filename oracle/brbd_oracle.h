/*
 * brbd_oracle.h -- CPU restatement of Pinocchio's rnea / aba / forwardKinematics
 * / computeJointJacobians / computeRNEADerivatives for ONE instance, plus
 * OpenMP batch loops shaped like rneaInParallel / abaInParallel.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may call this.  The product (libbrbd_hip.so)
 * never links or loads it.
 *
 * HOW PARITY IS PINNED.  The reference cannot be compiled or imported in the
 * build container (needs Eigen, Boost, urdfdom), so there is no
 * oracle/_ref.  This restatement is pinned
 * (0) BY VALUE for RNEA: the numbers the reference's own tests hold --
 *     unittest/value.cpp:31-179 (five RNEA vectors on simple_humanoid.urdf +
 *     free-flyer, the headline model, tolerances 1e-12 .. 1e-6) and
 *     unittest/joint-prismatic.cpp:137-216 -- committed as
 *     tests/golden/reference_value_cpp.json, tests/test_reference_values.py;
 * (1) by the reference's own test invariants re-expressed in
 *     tests/test_oracle_invariants.py (aba(rnea(a)) = a ties ABA to the
 *     value-pinned RNEA; derivatives vs finite differences of it; J qdot =
 *     oMi.act(v)); and
 * (2) by agreement <= 1e-12 with the structurally independent numpy
 *     implementation in oracle/naive_np.py.
 * ABA, the Jacobians and the derivatives have no stored numbers in the
 * reference: for them parity is pinned through (1) and (2) only.
 */
#ifndef BRBD_ORACLE_H
#define BRBD_ORACLE_H

#include "../include/brbd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* all q/v/a/tau pointers: one instance, contiguous (nq or nv doubles) */

/* algorithm/rnea.hxx:117-161 */
int brbd_oracle_rnea(const brbd_model_desc* m, const double* q, const double* v, const double* a,
                     double* tau);
/* algorithm/aba.hxx:243-294 (WORLD; what abaInParallel calls, parallel/aba.hpp:82) */
int brbd_oracle_aba_world(const brbd_model_desc* m, const double* q, const double* v,
                          const double* tau, double* ddq);
/* algorithm/aba.hxx:488-542 (LOCAL; cross-check, unittest/aba.cpp:157-159) */
int brbd_oracle_aba_local(const brbd_model_desc* m, const double* q, const double* v,
                          const double* tau, double* ddq);
/* algorithm/kinematics.hxx:52-97 + algorithm/jacobian.hxx:38-95.
 * J: 6 x nv column-major (may be NULL); oMi: njoints x 12, rotation column-major
 * then translation (may be NULL). */
/* rnea / aba with external forces (rnea.hxx:171-219, aba.hxx:296-354 WORLD, :544-603 LOCAL):
 * fext = 6 doubles [lin; ang] per joint (joint 0 included, ignored), in the joint frame */
int brbd_oracle_rnea_fext(const brbd_model_desc* m, const double* q, const double* v, const double* a,
                          const double* fext, double* tau);
int brbd_oracle_aba_fext(const brbd_model_desc* m, const double* q, const double* v, const double* tau,
                         const double* fext, int convention, double* ddq);
/* getJointJacobian after computeJointJacobians (jacobian.hxx:175-283): rf 0 WORLD, 1 LOCAL,
 * 2 LOCAL_WORLD_ALIGNED; J (6 x nv, column-major) must be zeroed by the caller */
int brbd_oracle_joint_jacobian(const brbd_model_desc* m, const double* q, int joint, int rf, double* J);
int brbd_oracle_fk_jacobians(const brbd_model_desc* m, const double* q, double* J, double* oMi);
/* algorithm/rnea-derivatives.hxx:472-541.  Outputs nv x nv column-major, fully
 * written (zeros included); dtau_da upper triangle only.  dtau_da / tau may be NULL. */
int brbd_oracle_rnea_derivatives(const brbd_model_desc* m, const double* q, const double* v,
                                 const double* a, double* dtau_dq, double* dtau_dv, double* dtau_da,
                                 double* tau);

/* algorithm/aba.hxx:841-877 computeMinverse: Minv nv x nv column-major, UPPER triangle (plus the
 * diagonal blocks of multi-dof joints), strictly lower part zero.  Depth-first numbering required. */
int brbd_oracle_minverse(const brbd_model_desc* m, const double* q, double* Minv);
/* algorithm/joint-configuration.hxx integrate(model, q, v, qout): qout = q (+) v per joint
 * (liegroup/special-euclidean.hpp:773-812, special-orthogonal.hpp:164-184, 495-510) */
int brbd_oracle_integrate(const brbd_model_desc* m, const double* q, const double* v, double* qout);
/* algorithm/aba-derivatives.hxx:381-454: ddq_dq, ddq_dv, ddq_dtau = Minv (full, symmetric), each
 * nv x nv column-major; ddq (nv) may be NULL */
int brbd_oracle_aba_derivatives(const brbd_model_desc* m, const double* q, const double* v, const double* tau,
                                double* ddq_dq, double* ddq_dv, double* Minv, double* ddq);

/* Batch loops: `#pragma omp parallel for schedule(static)` over instance-major
 * columns with per-thread scratch, as algorithm/parallel/rnea.hpp:70-82.
 * num_threads <= 0 means omp_get_max_threads(). Returns threads used or <0. */
int brbd_oracle_rnea_batch(const brbd_model_desc* m, int64_t B, const double* q, const double* v,
                           const double* a, double* tau, int num_threads);
int brbd_oracle_aba_batch(const brbd_model_desc* m, int64_t B, const double* q, const double* v,
                          const double* tau, double* ddq, int num_threads);
int brbd_oracle_fk_jacobians_batch(const brbd_model_desc* m, int64_t B, const double* q, double* J,
                                   double* oMi, int num_threads);
int brbd_oracle_rnea_derivatives_batch(const brbd_model_desc* m, int64_t B, const double* q,
                                       const double* v, const double* a, double* dq, double* dv,
                                       double* da, double* tau, int num_threads);
int brbd_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif

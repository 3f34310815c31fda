/*
 * brbd.h -- C ABI of the MI355X batched rigid-body-dynamics engine
 * (libbrbd_hip.so).  This is the drop-in boundary for the data-parallel path
 * of Pinocchio: everything under the `#pragma omp parallel for` of
 *   include/pinocchio/algorithm/parallel/rnea.hpp:74-82
 *   include/pinocchio/algorithm/parallel/aba.hpp:75-83
 * plus batched forwardKinematics / computeJointJacobians / computeRNEADerivatives.
 *
 * Plain pointers and sizes only; no torch / Eigen / STL types cross this line.
 * Every function returns BRBD_OK (0) or a negative status and never throws;
 * brbd_last_error() returns a thread-local message for the last failure.
 *
 * Conventions (identical to the reference):
 *   - spatial vectors are [linear(3); angular(3)]
 *     (multibody/joint/joint-revolute.hpp:56-60)
 *   - free-flyer q = [x y z, qx qy qz qw], v = [v_lin_body, w_body]
 *     (multibody/joint/joint-free-flyer.hpp:259-282); quaternions are NOT
 *     normalised by the engine (joint-free-flyer.hpp:338-339)
 *   - all arithmetic is IEEE fp64
 */
#ifndef BRBD_H
#define BRBD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------ */
#define BRBD_OK 0
#define BRBD_ERR_INVALID_ARGUMENT (-1) /* reference: std::invalid_argument       */
#define BRBD_ERR_UNSUPPORTED (-2)      /* joint type outside the supported set   */
#define BRBD_ERR_RUNTIME (-3)          /* HIP failure (reference: n/a)           */
#define BRBD_ERR_NO_DEVICE (-4)        /* no HIP device / extension not usable   */

/* ---- joint kinds (JointModelVariant alternatives,
 *      multibody/joint/joint-collection.hpp:86-112) ----------------------- */
enum brbd_joint_type {
  BRBD_JOINT_RX = 0, /* JointModelRX  joint-revolute.hpp   */
  BRBD_JOINT_RY = 1,
  BRBD_JOINT_RZ = 2,
  BRBD_JOINT_PX = 3, /* JointModelPX  joint-prismatic.hpp  */
  BRBD_JOINT_PY = 4,
  BRBD_JOINT_PZ = 5,
  BRBD_JOINT_FREEFLYER = 6,          /* joint-free-flyer.hpp              */
  BRBD_JOINT_SPHERICAL = 7,          /* joint-spherical.hpp               */
  BRBD_JOINT_REVOLUTE_UNALIGNED = 8, /* joint-revolute-unaligned.hpp      */
  BRBD_JOINT_PRISMATIC_UNALIGNED = 9,
  BRBD_JOINT_RUBX = 10, /* JointModelRUBX joint-revolute-unbounded.hpp (nq=2: cos,sin) */
  BRBD_JOINT_RUBY = 11,
  BRBD_JOINT_RUBZ = 12,
  /* joint-revolute-unbounded-unaligned.hpp:190-200 (nq=2: cos,sin; axis from `axis`) */
  BRBD_JOINT_REVOLUTE_UNBOUNDED_UNALIGNED = 13,
  BRBD_JOINT_TYPE_COUNT = 14
};

/* ---- matrix layout at the boundary ------------------------------------- */
enum brbd_layout {
  /* Pinocchio / Eigen column-major (dof x B): element (d, b) at x[b*ld + d],
   * ld >= dof.  This is what rneaInParallel receives
   * (algorithm/parallel/rnea.hpp:39-45). */
  BRBD_INSTANCE_MAJOR = 0,
  /* Engine-native SoA: element (d, b) at x[d*ld + b], ld >= B.  The batch is
   * the fastest-varying dimension so one lane per instance coalesces. */
  BRBD_BATCH_MAJOR = 1
};

enum brbd_memspace {
  BRBD_HOST = 0,  /* pointers are host memory: the call stages H2D/D2H itself */
  BRBD_DEVICE = 1 /* pointers are device memory on the model's device         */
};

/* ---- constant model description ---------------------------------------
 * Mirrors the ModelTpl fields the hot path reads (multibody/model.hpp:107-132,
 * 154,176,248).  Joint 0 is the universe; entries [0] of the per-joint arrays
 * are ignored.  parents[i] < i must hold (multibody/model.hxx:144-151). */
typedef struct brbd_model_desc {
  int32_t njoints; /* model.njoints (universe included)                      */
  int32_t nq;      /* model.nq                                               */
  int32_t nv;      /* model.nv                                               */
  const int32_t* parents; /* [njoints]  model.parents                        */
  const int32_t* jtype;   /* [njoints]  enum brbd_joint_type                 */
  const int32_t* idx_q;   /* [njoints]  model.idx_qs                         */
  const int32_t* idx_v;   /* [njoints]  model.idx_vs                         */
  /* [njoints][12] model.jointPlacements: rotation COLUMN-major (Eigen Matrix3
   * memory order) in [0..8], translation in [9..11]                         */
  const double* placement;
  /* [njoints][10] model.inertias: mass, lever c[3], Symmetric3 data
   * (xx,xy,yy,xz,yz,zz) of the rotational inertia about the COM
   * (spatial/inertia.hpp:286-291, spatial/symmetric3.hpp:43-52)             */
  const double* inertia;
  /* [njoints][3] unit axis for *_UNALIGNED joints, ignored otherwise; may be
   * NULL when the model has no unaligned joint                              */
  const double* axis;
  const double* armature; /* [nv] model.armature; NULL = zeros               */
  double gravity[3];      /* model.gravity.linear(), default (0,0,-9.81)     */
} brbd_model_desc;

typedef struct brbd_model brbd_model; /* opaque: flattened BatchModel + device table */

/* Last error message of the calling thread ("" if none). */
const char* brbd_last_error(void);

/* Library / device probes (never touch the GPU before brbd_model_create). */
const char* brbd_version(void);
int brbd_device_count(void);

/* Flatten a model description into the device-constant BatchModel table on
 * HIP device `device` (replaces one ModelPoolTpl replica set,
 * multibody/pool/model.hpp:41-46).  Rejects unsupported joint types with
 * BRBD_ERR_UNSUPPORTED, malformed tables with BRBD_ERR_INVALID_ARGUMENT. */
int brbd_model_create(const brbd_model_desc* desc, int device, brbd_model** out);

/* What the table-driven kernels see (host only, no device needed; diagnostics and tests).
 * Unaligned joints (joint-revolute-unaligned.hpp:667-708, joint-prismatic-unaligned.hpp) are
 * canonicalised by the flattener: with Q a rotation taking e_z to the axis, the joint becomes an
 * RZ / PZ joint in the frame (described frame) * Q; its body inertia and its children's placements
 * are re-expressed in that frame.  q, v, a, tau, the world-frame Jacobian and the derivatives are
 * unchanged; forwardKinematics' oMi is the canonical one times `post` (= Q^T).  Outputs, each
 * caller-allocated: jtype[njoints]; placement[njoints][12] and inertia[njoints][10] laid out like
 * brbd_model_desc; post[njoints][9] row-major (identity for joints left as described). */
int brbd_model_canonical_form(const brbd_model_desc* desc, int32_t* jtype, double* placement,
                              double* inertia, double* post);
int brbd_model_destroy(brbd_model* m);
/* Per-stream buffers.  The handle keeps, per stream it has been called on, the kernels' workspace and
 * (for BRBD_HOST / INSTANCE_MAJOR nv x nv calls) staging buffers: up to 1 GiB of batch-major copies,
 * 256 MiB of device staging and as much pinned host memory, all alive until brbd_model_destroy.  A
 * caller that uses many short-lived streams on a long-lived handle releases a stream's share here:
 * the call waits for the work queued on `stream` and frees its buffers (they are re-created on the
 * next call on that stream).  No reference counterpart. */
int brbd_model_release_stream(brbd_model* m, void* stream);
int brbd_model_dims(const brbd_model* m, int32_t* njoints, int32_t* nq, int32_t* nv);

/* Bytes of device workspace the table-driven kernels spill to for a batch of B instances
 * (reported separately from algorithmic I/O in the roofline); -1 for an algorithm without a
 * workspace of its own (BRBD_ALGO_ABA_DERIV_PRODUCTS). */
int64_t brbd_workspace_bytes(const brbd_model* m, int algo, int64_t B);
enum brbd_algo { BRBD_ALGO_RNEA = 0, BRBD_ALGO_ABA = 1, BRBD_ALGO_FK_JAC = 2, BRBD_ALGO_RNEA_DERIV = 3,
                 BRBD_ALGO_MINVERSE = 4, BRBD_ALGO_ABA_DERIV_PRODUCTS = 5, BRBD_ALGO_JOINT_JAC = 6 };

/* tau = RNEA(q, v, a) for B instances.
 * Replaces the loop body of rneaInParallel (algorithm/parallel/rnea.hpp:74-82)
 * = impl::rnea (algorithm/rnea.hxx:117-161).
 * `stream` is a hipStream_t (NULL = default stream).  With BRBD_DEVICE the call
 * is asynchronous on `stream`; with BRBD_HOST it returns after the results are
 * in `tau` (the batch is pipelined in column chunks over two internal streams
 * ordered after `stream`; caller padding, ld > rows, is neither read nor written).
 * Re-entrancy: workspace and staging are kept per (handle, stream), so calls on
 * different streams of one handle may overlap; one stream of one handle must be
 * driven by one host thread at a time. */
int brbd_rnea(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
              const double* a, int64_t lda, double* tau, int64_t ldtau, int layout, int memspace,
              void* stream);

/* ddq = ABA(q, v, tau), WORLD convention, for B instances.
 * Replaces the loop body of abaInParallel (algorithm/parallel/aba.hpp:75-83)
 * = impl::abaWorldConvention (algorithm/aba.hxx:243-294). */
int brbd_aba(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
             const double* tau, int64_t ldtau, double* ddq, int64_t ldddq, int layout, int memspace,
             void* stream);

/* rnea / aba with external forces: rnea(model, data, q, v, a, fext) (algorithm/rnea.hxx:171-219,
 * f[i] -= fext[i]) and aba(model, data, q, v, tau, fext) in the WORLD convention
 * (algorithm/aba.hxx:296-354, of[i] -= oMi[i].act(fext[i])).  fext is a (6 * njoints) x B matrix
 * in the layout of the call: rows 6i .. 6i+5 = the spatial force [linear; angular] on joint i in
 * that joint's frame (the reference's aligned_vector<Force> of model.joints.size() entries, the
 * universe's entry included and ignored).  The parallel wrappers of the reference have no such
 * overload; these two are what a batched caller with contact forces binds instead. */
int brbd_rnea_fext(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
                   const double* a, int64_t lda, const double* fext, int64_t ldfext, double* tau,
                   int64_t ldtau, int layout, int memspace, void* stream);
int brbd_aba_fext(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
                  const double* tau, int64_t ldtau, const double* fext, int64_t ldfext, double* ddq,
                  int64_t ldddq, int layout, int memspace, void* stream);

/* brbd_aba for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that must not call hipMalloc -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  Same arguments and results as brbd_aba with BRBD_DEVICE
 * pointers (asynchronous on `stream`), but it always runs the table-driven ABA kernel and that
 * kernel's workspace (pass-2 -> pass-3 hand-off records, live-state stack overflow) is a fixed
 * 64 MiB file-scope device array instead of a buffer the handle allocates and grows.  The call
 * allocates and frees nothing (no hipMalloc / hipFree / events).  Limits: B <= 1 048 576 per call
 * (BRBD_ERR_INVALID_ARGUMENT beyond), and a model whose workspace for ONE wave of 64 instances
 * exceeds 64 MiB is rejected; a batch larger than the array holds at once runs with fewer waves
 * (the kernel strides over the tiles).  Every GPU the library runs on holds its own copy of the
 * array until the process ends; all calls on one GPU share it, so calls that overlap on one GPU
 * (different streams or handles) race: serialise them. */
int brbd_aba_noalloc(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
                     const double* tau, int64_t ldtau, double* ddq, int64_t ldddq, int layout, void* stream);

/* Page-lock / release a host range (hipHostRegister / hipHostUnregister) so that BRBD_HOST calls on
 * matrices inside it copy by DMA and overlap upload, kernels and download; the caller must
 * unregister before freeing the memory.  Optional: pageable memory works (the library packs it
 * into its own pinned bounce buffers with a few worker threads), about 20 % slower.  No
 * reference counterpart (the reference never leaves host memory). */
int brbd_host_register(void* ptr, size_t bytes);
int brbd_host_unregister(void* ptr);

/* forwardKinematics(q) + computeJointJacobians(q) for B instances
 * (algorithm/kinematics.hxx:52-97, algorithm/jacobian.hxx:38-95).
 * J: per instance 6 x nv column-major (row 0-2 linear, 3-5 angular), WORLD
 *    frame; element (r, c) of instance b is
 *      INSTANCE_MAJOR: J[b*ldJ + c*6 + r]   (ldJ >= 6*nv)
 *      BATCH_MAJOR:    J[(c*6 + r)*ldJ + b] (ldJ >= B)
 * oMi: per instance njoints x 12 (rotation column-major then translation),
 *    joint 0 = identity; same two layouts with dof = 12*njoints.
 * Either of J / oMi may be NULL to skip that output. */
int brbd_fk_jacobians(brbd_model* m, int64_t B, const double* q, int64_t ldq, double* J, int64_t ldJ,
                      double* oMi, int64_t ldoMi, int layout, int memspace, void* stream);

/* computeJointJacobians(q) followed by getJointJacobian(joint, rf, J) per instance
 * (algorithm/jacobian.hxx:38-95, 175-283, 315-326): J is (6 * nv) x B, per instance a column-major
 * 6 x nv matrix; the columns of the joint's support (the joint and its ancestors) hold the joint
 * Jacobian expressed in rf (pinocchio::ReferenceFrame: 0 WORLD, 1 LOCAL, 2 LOCAL_WORLD_ALIGNED),
 * every other column is written as zero (the reference leaves them to the caller's setZero()).
 * Models with compiled kernels and at most 64 joints run the model-compiled kernel
 * (pinocchio_amd/codegen/jjac.py: one kernel per model and frame, the joint a run-time argument). */
int brbd_joint_jacobian(brbd_model* m, int64_t B, const double* q, int64_t ldq, int joint, int rf,
                        double* J, int64_t ldJ, int layout, int memspace, void* stream);

/* computeRNEADerivatives (algorithm/rnea-derivatives.hxx:472-541) for B
 * instances.  Each output is nv x nv column-major per instance (element (r,c)
 * at offset c*nv + r), in the two layouts above with dof = nv*nv.  Unlike the
 * reference (rnea-derivatives.hpp:104-106) the outputs need not be pre-zeroed:
 * the engine writes every element (structural zeros included); dtau_da holds
 * the UPPER triangle of M like the reference (plus the full nvj x nvj diagonal
 * block of multi-dof joints, as rnea-derivatives.hxx:420-421 writes it).
 * dtau_da and tau may be NULL.
 * BRBD_DEVICE + BRBD_INSTANCE_MAJOR with B >= 4096 (also brbd_minverse,
 * brbd_aba_derivatives): the call runs on batch-major copies of column chunks
 * that the handle keeps per stream (up to 1 GiB) and transposes the results
 * into the caller's matrices; same values as the BRBD_BATCH_MAJOR call. */
int brbd_rnea_derivatives(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v,
                          int64_t ldv, const double* a, int64_t lda, double* dtau_dq, int64_t lddq,
                          double* dtau_dv, int64_t lddv, double* dtau_da, int64_t ldda, double* tau,
                          int64_t ldtau, int layout, int memspace, void* stream);

/* integrate(model, q, v, qout) per instance (algorithm/joint-configuration.hxx; per joint the
 * Lie-group integrate_impl: liegroup/vector-space.hpp q + v, special-orthogonal.hpp:164-184 (revolute
 * unbounded), :495-510 (spherical), special-euclidean.hpp:773-812 (free-flyer: q * exp6(v), sign
 * kept on the hemisphere of q, first-order re-normalisation)).  qout is nq x B; it may not alias q. */
int brbd_integrate(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv, double* qout,
                   int64_t ldqout, int layout, int memspace, void* stream);

/* computeMinverse(model, data, q) per instance (algorithm/aba.hxx:841-877): Minv is (nv * nv) x B,
 * per instance column-major nv x nv; like data.Minv in the reference only the UPPER triangle (and the
 * diagonal blocks of multi-dof joints) is computed, the strictly lower part is written as zero.
 * Needs depth-first joint numbering: every subtree a contiguous index range, which is what the
 * reference assumes implicitly through data.nvSubtree (multibody/data.hxx:205-230) and what
 * Model::addJoint produces for trees built limb by limb; BRBD_ERR_UNSUPPORTED otherwise (also
 * brbd_rnea_derivatives and brbd_aba_derivatives).  Models with compiled
 * kernels run the model-compiled computeMinverse (pinocchio_amd/codegen/minv.py) on BATCH_MAJOR
 * device matrices (INSTANCE_MAJOR ones go through batch-major copies, see brbd_rnea_derivatives). */
int brbd_minverse(brbd_model* m, int64_t B, const double* q, int64_t ldq, double* Minv, int64_t ldminv, int layout,
                  int memspace, void* stream);

/* computeABADerivatives(model, data, q, v, tau, ddq_dq, ddq_dv, ddq_dtau) per instance
 * (algorithm/aba-derivatives.hxx:381-454): three (nv * nv) x B matrices, per instance column-major;
 * ddq_dtau = Minv, FULL and symmetric (:449-450); ddq_dq = -Minv dtau_dq, ddq_dv = -Minv dtau_dv at
 * the acceleration ddq = aba(q, v, tau) (:452-453), which is returned in `ddq` (nv x B) unless NULL.
 * Every element is written.  Depth-first numbering (see brbd_minverse); nv <= 96. */
int brbd_aba_derivatives(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
                         const double* tau, int64_t ldtau, double* ddq_dq, int64_t lddq, double* ddq_dv, int64_t lddv,
                         double* ddq_dtau, int64_t lddtau, double* ddq, int64_t ldddq, int layout, int memspace,
                         void* stream);

/* Layout helpers (device): instance-major <-> batch-major transposes used by
 * the INSTANCE_MAJOR entry points; exposed for harnesses.
 * dst/src are device pointers; dof x B logical matrix. */
int brbd_transpose_to_batch_major(int device, int64_t dof, int64_t B, const double* src, int64_t ld_src,
                                  double* dst, int64_t ld_dst, void* stream);
int brbd_transpose_to_instance_major(int device, int64_t dof, int64_t B, const double* src,
                                     int64_t ld_src, double* dst, int64_t ld_dst, void* stream);

/* Model-compiled kernels.  For models registered at build time (pinocchio_amd/codegen/build.py)
 * the library also holds straight-line kernels generated for that very model: the recursion
 * unrolled over its tree with every structural zero of its constants folded away ("<name>_p"),
 * or with all geometric / inertial constants generic ("<name>_d": any model with the same tree and
 * joint kinds).  brbd_model_create selects them when the model's signature matches; every other
 * model runs on the table-driven kernels.  Both produce the results of the same recursions.
 * brbd_model_compiled_kernels returns the matched entry's name ("" = table-driven);
 * brbd_model_use_compiled(m, 0) forces the table-driven kernels (BRBD_DISABLE_COMPILED=1 does so
 * for every model).  No reference counterpart (closest: the CppADCodeGen path, out of scope). */
const char* brbd_model_compiled_kernels(const brbd_model* m);
/* Compiled kernels for a model that was NOT known when the library was built: the model compiler runs
 * on any model (`python -m pinocchio_amd.codegen.plugin --urdf robot.urdf --out libbrbd_spec_robot.so`,
 * a few minutes of hipcc) and links that model's kernels into a small shared library.  Loading it
 * (before brbd_model_create; or list it in BRBD_COMPILED_PLUGINS=a.so:b.so) makes models with that
 * signature run on them.  Returns the number of kernel sets added (>= 0) or a negative status
 * (BRBD_ERR_INVALID_ARGUMENT: not loadable, not a plugin, or built against another kernel ABI). */
int brbd_load_compiled_kernels(const char* path);
int brbd_model_use_compiled(brbd_model* m, int enabled);

/* Output contract of brbd_rnea_derivatives.  The reference requires the caller to zero the three
 * matrices before the call and fills only the entries that couple a joint with its own subtree
 * and root path (include/pinocchio/algorithm/rnea-derivatives.hpp:104-106: "must be first
 * initialized with zeros").  Default here (0): every element is written, structural zeros
 * included, so the outputs need no clearing.  1 = the reference's contract: the caller has zeroed
 * dtau_dq / dtau_dv / dtau_da and the structural zeros are not written again, which saves their
 * HBM traffic (41% of the stores for a quadruped).  Honoured by the model-compiled kernels; the
 * table-driven kernel writes every element in both modes (same result). */
int brbd_model_outputs_prezeroed(brbd_model* m, int enabled);

/* Checking build only (pinocchio_amd/libbrbd_hip_check.so, `make -C pinocchio_amd/csrc check`): the
 * model-compiled kernels address memory through range-checked buffer resources, and the hardware
 * silently DROPS an out-of-range access; the checking build compares every offset with its range
 * first and counts violations.  Returns that count after synchronising the device (0 = every
 * access of every launch on this handle was in range), or -1 in the regular build.  No reference
 * counterpart. */
int64_t brbd_debug_oob_count(brbd_model* m);

/* Kernel timing for the roofline (bench.py): with timing enabled every kernel launch of the
 * engine is bracketed by hipEvents recorded on the launch stream.  brbd_kernel_time() waits for
 * the recorded launches of `algo` (enum brbd_algo), returns how many there were and their summed
 * device time in milliseconds, and forgets them.  enabled = 2 ("chained"): the event that closes
 * one launch also opens the next launch of the same handle on the same stream, i.e. one event
 * per launch instead of two (an event costs the stream about 3 us); a launch's time then includes
 * the gap to its predecessor, and whatever else the caller put on the stream in between.
 * enabled = 3: a call that is ONE dispatch of a model-compiled kernel carries its own begin / end
 * timestamps (hipExtLaunchKernelGGL): the time is the kernel's alone, as a profiler reports it, but
 * the stream pays more per launch than with chained events (MI355X: +7 us per RNEA + ABA step
 * against +4.5); every other launch is timed as with 2. */
int brbd_set_timing(brbd_model* m, int enabled);
/* Time only every period-th launch of each algorithm (default 1: all of them), counted per handle
 * from this call on: a long run then pays for the events of a sample of its launches only (MI355X,
 * RNEA + ABA step: 3.2 us per step with chained events on every launch).  With enabled = 2 the
 * timed launches of different algorithms still share events when they follow each other directly
 * (launch k of RNEA and launch k of ABA are timed together); an untimed launch in between breaks
 * the chain, so a timed launch never includes an untimed one.  No reference counterpart. */
int brbd_set_timing_period(brbd_model* m, int period);
/* (every enum brbd_algo value is timed, compiled and table-driven kernels alike; brbd_integrate
 * has no enum value and is not timed) */
int brbd_kernel_time(brbd_model* m, int algo, int32_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* BRBD_H */

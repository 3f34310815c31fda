// brbd_device.hpp -- device-side constant tables and 6-D spatial algebra for gfx950.
//
// One lane = one robot instance; every value below lives in VGPRs (arrays are only ever
// indexed with compile-time constants after unrolling).  The constant model table is read
// with wave-uniform addresses, i.e. through the scalar data path into SGPRs, and fp64 VALU
// instructions consume those SGPR pairs directly.
//
// Conventions (identical to the reference): 6-vectors are [lin(3); ang(3)]
// (multibody/joint/joint-revolute.hpp:56-60); rotations are row-major double[9].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DI __device__ __forceinline__

namespace brbd {

// ---- joint classes the kernels specialise on ------------------------------------------------
enum : int {
  JT_RX = 0, JT_RY, JT_RZ, JT_PX, JT_PY, JT_PZ, JT_FF, JT_SPH, JT_RU, JT_PU, JT_RUBX, JT_RUBY, JT_RUBZ
};

// ---- per-joint constant record (BatchModel table row) -----------------------------------------
// 368 bytes, 16-byte aligned so the scalar unit can fetch it with s_load_dwordx8/x16.
struct alignas(16) JointConst {
  double R[9];    // jointPlacement rotation, row-major
  double t[3];    // jointPlacement translation
  double m;       // inertia: mass
  double c[3];    //          lever
  double I[6];    //          Symmetric3 (xx,xy,yy,xz,yz,zz) about the COM
  double arm[6];  // armature of this joint's dofs
  double axis[3]; // unit axis (unaligned kinds)
  double post[9]; // canonicalised unaligned joints: oMi = (oMi of the canonical frame) * post (row-major)
  int32_t jtype, parent, idx_q, idx_v;
  int32_t hand_off;  // ABA: first slot of this joint's pass2->pass3 hand-off record
  int32_t comp;      // 1-dof aligned kinds: index of the non-zero component of S (0..5)
  int32_t nvj, nqj;
};

// The table and the program are immutable during a launch: read them through the CONSTANT
// address space so that every field is fetched by the scalar unit (s_load) into SGPRs instead of
// 64 identical per-lane vector loads.
#define BRBD_CONSTANT __attribute__((address_space(4)))
template <typename T>
DI const T BRBD_CONSTANT* as_constant(const T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (const T BRBD_CONSTANT*)p;
#pragma clang diagnostic pop
}
DI void load_joint(const JointConst* tab, int i, JointConst& o) {
  const JointConst BRBD_CONSTANT* t = as_constant(tab) + i;
#pragma unroll
  for (int k = 0; k < 9; ++k) { o.R[k] = t->R[k]; o.post[k] = t->post[k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) { o.t[k] = t->t[k]; o.c[k] = t->c[k]; o.axis[k] = t->axis[k]; }
  o.m = t->m;
#pragma unroll
  for (int k = 0; k < 6; ++k) { o.I[k] = t->I[k]; o.arm[k] = t->arm[k]; }
  o.jtype = t->jtype; o.parent = t->parent; o.idx_q = t->idx_q; o.idx_v = t->idx_v;
  o.hand_off = t->hand_off; o.comp = t->comp; o.nvj = t->nvj; o.nqj = t->nqj;
}

// ---- one step of the depth-first program ----------------------------------------------------
struct alignas(16) Op {
  int32_t joint;
  int32_t flags;
  int32_t depth;  // slot in the depth-indexed stack
  int32_t bslot;  // branch slot of the PARENT (to read its state / reach its accumulator);
                  // the joint's own branch slot is in bits 16.. of flags
  // integer description of the joint, duplicated here so that a compile-time program carries
  // everything the control flow depends on (the numeric constants stay in JointConst)
  int32_t jtype, idx_q, idx_v, nvj;
  int32_t comp, hand_off, pad0, pad1;
};
enum : int {
  OP_BWD = 1,          // backward visit of a non-leaf joint
  F_FROM_REGS = 2,     // FWD: parent state is still in registers (previous op was FWD(parent))
  F_PARENT_ROOT = 4,   // parent == 0 (universe)
  F_PUSH_BRANCH = 8,   // FWD: joint has >= 2 children: push its state to its branch slot
  F_LEAF = 16,         // FWD: joint has no children: do the backward visit in the same step
  F_CARRY_REGS = 32,   // backward visit: hand the contribution to the parent in registers
                       //   (parent has exactly one child) instead of read-modify-write
  F_CARRY_IN = 64,     // BWD: exactly one child: its contribution is in the carry registers
  F_MULTI = 128,       // BWD: >= 2 children: accumulator slot already holds own + children
  F_POST = 256         // FK output: the joint frame was canonicalised (unaligned axis): oMi = oMi' * post
};
DI int op_own_bslot(int flags) { return flags >> 16; }
DI Op load_op(const Op* prog, int s) {
  const Op BRBD_CONSTANT* p = as_constant(prog) + s;
  Op o;
  o.joint = p->joint; o.flags = p->flags; o.depth = p->depth; o.bslot = p->bslot;
  o.jtype = p->jtype; o.idx_q = p->idx_q; o.idx_v = p->idx_v; o.nvj = p->nvj;
  o.comp = p->comp; o.hand_off = p->hand_off; o.pad0 = 0; o.pad1 = 0;
  return o;
}
DI Op op_from_joint(const JointConst& jc, int joint) {
  Op o;
  o.joint = joint; o.flags = 0; o.depth = 0; o.bslot = 0;
  o.jtype = jc.jtype; o.idx_q = jc.idx_q; o.idx_v = jc.idx_v; o.nvj = jc.nvj;
  o.comp = jc.comp; o.hand_off = jc.hand_off; o.pad0 = 0; o.pad1 = 0;
  return o;
}

// value select (never a select between addresses, which would defeat scalar replacement)
DI double sel3(int k, double a, double b, double c) {
  double r = c;
  r = (k == 1) ? b : r;
  r = (k == 0) ? a : r;
  return r;
}

// ---- 3-vectors / rotations -------------------------------------------------------------------
DI void cross3(const double* a, const double* b, double* o) {
  const double x = a[1] * b[2] - a[2] * b[1];
  const double y = a[2] * b[0] - a[0] * b[2];
  const double z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
DI void rot_vec(const double* R, const double* x, double* o) {  // o = R x
  const double a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  const double b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  const double c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  o[0] = a; o[1] = b; o[2] = c;
}
DI void rotT_vec(const double* R, const double* x, double* o) {  // o = R^T x
  const double a = R[0] * x[0] + R[3] * x[1] + R[6] * x[2];
  const double b = R[1] * x[0] + R[4] * x[1] + R[7] * x[2];
  const double c = R[2] * x[0] + R[5] * x[1] + R[8] * x[2];
  o[0] = a; o[1] = b; o[2] = c;
}
DI void rot_mul(const double* A, const double* B, double* C) {  // C = A B (no aliasing)
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// ---- SE3 actions (spatial/motion-dense.hpp:261-282, spatial/force-dense.hpp:217-239) ----------
DI void se3_act_motion(const double* R, const double* t, const double* m, double* o) {
  double ang[3], lin[3], txa[3];
  rot_vec(R, m + 3, ang);
  rot_vec(R, m, lin);
  cross3(t, ang, txa);
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = lin[k] + txa[k]; o[3 + k] = ang[k]; }
}
DI void se3_actinv_motion(const double* R, const double* t, const double* m, double* o) {
  double txa[3], tmp[3];
  cross3(t, m + 3, txa);
#pragma unroll
  for (int k = 0; k < 3; ++k) tmp[k] = m[k] - txa[k];
  rotT_vec(R, tmp, o);
  rotT_vec(R, m + 3, o + 3);
}
DI void se3_act_force(const double* R, const double* t, const double* f, double* o) {
  double lin[3], ang[3], txl[3];
  rot_vec(R, f, lin);
  rot_vec(R, f + 3, ang);
  cross3(t, lin, txl);
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = lin[k]; o[3 + k] = ang[k] + txl[k]; }
}
// v x m (spatial/motion-dense.hpp:224-229)
DI void motion_cross(const double* v, const double* m, double* o) {
  double a[3], b[3], c[3];
  cross3(v, m + 3, a);
  cross3(v + 3, m, b);
  cross3(v + 3, m + 3, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = a[k] + b[k]; o[3 + k] = c[k]; }
}
// v x* f (spatial/force-dense.hpp:180-185)
DI void force_cross(const double* v, const double* f, double* o) {
  double a[3], b[3], c[3];
  cross3(v + 3, f, a);
  cross3(v + 3, f + 3, b);
  cross3(v, f, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = a[k]; o[3 + k] = b[k] + c[k]; }
}
// f = Y v with Y = (m, c, I sym3) (spatial/inertia.hpp:741-748)
DI void inertia_mul(double m, const double* c, const double* I, const double* v, double* f) {
  double cxw[3], lin[3], cxf[3];
  cross3(c, v + 3, cxw);
#pragma unroll
  for (int k = 0; k < 3; ++k) lin[k] = m * (v[k] - cxw[k]);
  const double a0 = I[0] * v[3] + I[1] * v[4] + I[3] * v[5];
  const double a1 = I[1] * v[3] + I[2] * v[4] + I[4] * v[5];
  const double a2 = I[3] * v[3] + I[4] * v[4] + I[5] * v[5];
  cross3(c, lin, cxf);
  f[0] = lin[0]; f[1] = lin[1]; f[2] = lin[2];
  f[3] = a0 + cxf[0]; f[4] = a1 + cxf[1]; f[5] = a2 + cxf[2];
}

// ---- packed symmetric 6x6 (21 entries, upper triangle row-major) ------------------------------
__host__ __device__ constexpr int sidx(int i, int j) {
  return i <= j ? (i * (13 - i)) / 2 + (j - i) : (j * (13 - j)) / 2 + (i - j);
}
DI void sym6_mulvec(const double* A, const double* x, double* o) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) s += A[sidx(i, j)] * x[j];
    o[i] = s;
  }
}
DI double dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

// World inertia of a body: Yw = oMi.act(Y) = (m, t + R c, R I R^T) (spatial/inertia.hpp:885-893)
DI void inertia_to_world(const double* R, const double* t, double m, const double* c, const double* I,
                         double* cw, double* Iw) {
  double Rc[3];
  rot_vec(R, c, Rc);
#pragma unroll
  for (int k = 0; k < 3; ++k) cw[k] = t[k] + Rc[k];
  // T = R * I (I symmetric), Iw = T * R^T, only the 6 unique entries
  double T[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    T[3 * i + 0] = R[3 * i] * I[0] + R[3 * i + 1] * I[1] + R[3 * i + 2] * I[3];
    T[3 * i + 1] = R[3 * i] * I[1] + R[3 * i + 1] * I[2] + R[3 * i + 2] * I[4];
    T[3 * i + 2] = R[3 * i] * I[3] + R[3 * i + 1] * I[4] + R[3 * i + 2] * I[5];
  }
  Iw[0] = T[0] * R[0] + T[1] * R[1] + T[2] * R[2];  // xx
  Iw[1] = T[0] * R[3] + T[1] * R[4] + T[2] * R[5];  // xy
  Iw[2] = T[3] * R[3] + T[4] * R[4] + T[5] * R[5];  // yy
  Iw[3] = T[0] * R[6] + T[1] * R[7] + T[2] * R[8];  // xz
  Iw[4] = T[3] * R[6] + T[4] * R[7] + T[5] * R[8];  // yz
  Iw[5] = T[6] * R[6] + T[7] * R[7] + T[8] * R[8];  // zz
}
// Inertia::matrix() (spatial/inertia.hpp:480-491) into packed sym6:
// [[m 1, -m[c]x], [m[c]x, I - m[c]x[c]x]]
DI void inertia_to_sym6(double m, const double* c, const double* I, double* A) {
  const double mcx = m * c[0], mcy = m * c[1], mcz = m * c[2];
  A[sidx(0, 0)] = m;  A[sidx(0, 1)] = 0.0; A[sidx(0, 2)] = 0.0;
  A[sidx(1, 1)] = m;  A[sidx(1, 2)] = 0.0; A[sidx(2, 2)] = m;
  // -m [c]x = [[0, mcz, -mcy], [-mcz, 0, mcx], [mcy, -mcx, 0]]
  A[sidx(0, 3)] = 0.0;  A[sidx(0, 4)] = mcz;  A[sidx(0, 5)] = -mcy;
  A[sidx(1, 3)] = -mcz; A[sidx(1, 4)] = 0.0;  A[sidx(1, 5)] = mcx;
  A[sidx(2, 3)] = mcy;  A[sidx(2, 4)] = -mcx; A[sidx(2, 5)] = 0.0;
  // I - m [c]x [c]x = I + m (|c|^2 1 - c c^T)
  A[sidx(3, 3)] = I[0] + (mcy * c[1] + mcz * c[2]);
  A[sidx(3, 4)] = I[1] - mcx * c[1];
  A[sidx(3, 5)] = I[3] - mcx * c[2];
  A[sidx(4, 4)] = I[2] + (mcx * c[0] + mcz * c[2]);
  A[sidx(4, 5)] = I[4] - mcy * c[2];
  A[sidx(5, 5)] = I[5] + (mcx * c[0] + mcy * c[1]);
}

// ---- quaternion -> rotation, Eigen formula, no normalisation ----------------------------------
// (used at multibody/joint/joint-free-flyer.hpp:338-339, joint-spherical.hpp:528-529)
DI void quat_to_R(double x, double y, double z, double w, double* R) {
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// placement * Rot_K(q): two rotated columns + one cross product, translation unchanged
// (multibody/joint/joint-revolute.hpp:132-166)
template <int K>
DI void revolute_liMi(const double* P, double s, double c, double* R) {
  constexpr int KA = (K + 1) % 3, KB = (K + 2) % 3;
  double ck[3], na[3], nb[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    ck[r] = P[3 * r + K];
    na[r] = c * P[3 * r + KA] + s * P[3 * r + KB];
  }
  cross3(ck, na, nb);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    R[3 * r + K] = ck[r];
    R[3 * r + KA] = na[r];
    R[3 * r + KB] = nb[r];
  }
}

// Symmetric positive-definite inverse (Cholesky), N in {3, 6}; replaces matrix_inversion
// (math/matrix-inverse.hpp:142-147).  A, Ainv: full row-major N x N.
template <int N>
DI void spd_inverse(const double* A, double* Ainv) {
  double L[N * N], Li[N * N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double d = A[N * j + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[N * j + k] * L[N * j + k];
    const double rd = rsqrt(d);
    L[N * j + j] = d * rd;
    Li[N * j + j] = rd;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double s = A[N * i + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[N * i + k] * L[N * j + k];
      L[N * i + j] = s * rd;
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j)
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double s = 0.0;
#pragma unroll
      for (int k = j; k < i; ++k) s -= L[N * i + k] * Li[N * k + j];
      Li[N * i + j] = s * Li[N * i + i];
    }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = i; j < N; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = j; k < N; ++k) s += Li[N * k + i] * Li[N * k + j];
      Ainv[N * i + j] = s;
      Ainv[N * j + i] = s;
    }
}

}  // namespace brbd

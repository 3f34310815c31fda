// brbd_flatten.hpp -- host-side BatchModel flattener.
//
// Turns the constant tree description (include/brbd.h: brbd_model_desc, i.e. the fields of
// pinocchio::ModelTpl listed in SURVEY 8a row a13) into
//   * the per-joint constant table (JointConst rows), and
//   * the depth-first visit PROGRAM the kernels interpret.
//
// Children of a joint are visited in DECREASING index order, so that the child->parent
// accumulations (f[parent] += ..., oYaba[parent] += ...) happen in exactly the order of the
// reference's `for (i = njoints-1; i > 0; --i)` loops (rnea.hxx:152-155, aba.hxx:276-279).
#pragma once
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/brbd.h"
#include "brbd_kernels.hpp"

namespace brbd {

static const int kJointNq[BRBD_JOINT_TYPE_COUNT] = {1, 1, 1, 1, 1, 1, 7, 4, 1, 1, 2, 2, 2, 2};
static const int kJointNv[BRBD_JOINT_TYPE_COUNT] = {1, 1, 1, 1, 1, 1, 6, 3, 1, 1, 1, 1, 1, 1};

struct FlatModel {
  int njoints = 0, nq = 0, nv = 0;
  int maxdepth = 1, maxbranch = 1;
  int hand_total = 0;  // doubles per lane of ABA hand-off records
  double gravity[3] = {0, 0, -9.81};
  std::vector<JointConst> tab;       // what the table-driven kernels read (unaligned joints canonicalised)
  std::vector<JointConst> tab_desc;  // as described (signature / constants of the model-compiled kernels)
  std::vector<Op> prog;   // FWD/BWD program (RNEA, ABA passes 1+2)
  std::vector<Op> prog3;  // FWD-only program (ABA pass 3, FK/Jacobians)
  bool depth_first = true;  // subtrees are contiguous index ranges (needed by RNEA derivatives)
  std::vector<int> nv_subtree;

  // workspace doubles per lane
  int stack_rnea() const { return maxdepth * 8 + maxbranch * 12; }
  int stack_aba() const { return maxdepth * 2 + maxbranch * 27; }
  int ws_aba() const { return stack_aba() + hand_total; }
  int ws_fkjac() const { return maxbranch * 12; }
};

// Unaligned joints (joint-revolute-unaligned.hpp:667-708, joint-prismatic-unaligned.hpp,
// joint-revolute-unbounded-unaligned.hpp:190-200) run on the
// kernels of the aligned ones.  With Q a rotation that takes e_z to the axis u, the joint transform
// is M(q) = Q M_z(q) Q^T (rotation about / translation along z), so the joint is an RZ / PZ joint
// whose frame is the described one times Q: placement' = placement Q, the body inertia and the
// children's placements are re-expressed in that frame (Q^T . Y, Q^T . placement_child), q / v / a /
// tau are the same numbers.  RNEA, ABA, the world-frame Jacobian and the derivatives do not
// depend on the choice of joint frame; forwardKinematics' oMi does: oMi = oMi' Q^T (F_POST).
inline std::string canonicalise_unaligned(struct FlatModel& M, const std::vector<std::vector<int>>& children);

// returns "" on success, else an error message; *status gets the BRBD_ERR_* code
inline std::string flatten(const brbd_model_desc* d, FlatModel* out, int* status) {
  *status = BRBD_ERR_INVALID_ARGUMENT;
  if (!d) return "model description is NULL";
  if (d->njoints < 1) return "njoints must be >= 1 (joint 0 is the universe)";
  if (!d->parents || !d->jtype || !d->idx_q || !d->idx_v || !d->placement || !d->inertia)
    return "model description has NULL tables";
  const int n = d->njoints;
  FlatModel& M = *out;
  M = FlatModel();
  M.njoints = n;
  M.nq = d->nq;
  M.nv = d->nv;
  for (int k = 0; k < 3; ++k) M.gravity[k] = d->gravity[k];
  M.tab.resize(n);
  std::vector<std::vector<int>> children(n);
  std::vector<char> qcover(d->nq > 0 ? d->nq : 0, 0), vcover(d->nv > 0 ? d->nv : 0, 0);
  int hand = 0;
  for (int i = 0; i < n; ++i) {
    JointConst& jc = M.tab[i];
    std::memset(&jc, 0, sizeof jc);
    if (i == 0) continue;
    const int p = d->parents[i], jt = d->jtype[i];
    if (p < 0 || p >= i)
      return "parents[" + std::to_string(i) + "] must satisfy 0 <= parent < joint index";
    if (jt < 0 || jt >= BRBD_JOINT_TYPE_COUNT) return "jtype[" + std::to_string(i) + "] is not a brbd_joint_type";
    if ((jt == BRBD_JOINT_REVOLUTE_UNALIGNED || jt == BRBD_JOINT_PRISMATIC_UNALIGNED ||
         jt == BRBD_JOINT_REVOLUTE_UNBOUNDED_UNALIGNED) && !d->axis)
      return "joint " + std::to_string(i) + " is unaligned but the description has no axis table";
    const int nqj = kJointNq[jt], nvj = kJointNv[jt];
    if (d->idx_q[i] < 0 || d->idx_q[i] + nqj > d->nq) return "idx_q[" + std::to_string(i) + "] out of range";
    if (d->idx_v[i] < 0 || d->idx_v[i] + nvj > d->nv) return "idx_v[" + std::to_string(i) + "] out of range";
    for (int k = 0; k < nqj; ++k) {
      if (qcover[d->idx_q[i] + k]) return "idx_q ranges overlap at joint " + std::to_string(i);
      qcover[d->idx_q[i] + k] = 1;
    }
    for (int k = 0; k < nvj; ++k) {
      if (vcover[d->idx_v[i] + k]) return "idx_v ranges overlap at joint " + std::to_string(i);
      vcover[d->idx_v[i] + k] = 1;
    }
    children[p].push_back(i);
    const double* P = d->placement + 12 * (size_t)i;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) jc.R[3 * r + c] = P[3 * c + r];  // column-major in, row-major table
    jc.t[0] = P[9]; jc.t[1] = P[10]; jc.t[2] = P[11];
    const double* Y = d->inertia + 10 * (size_t)i;
    jc.m = Y[0];
    jc.c[0] = Y[1]; jc.c[1] = Y[2]; jc.c[2] = Y[3];
    for (int k = 0; k < 6; ++k) jc.I[k] = Y[4 + k];
    for (int k = 0; k < nvj; ++k) jc.arm[k] = d->armature ? d->armature[d->idx_v[i] + k] : 0.0;
    if (d->axis)
      for (int k = 0; k < 3; ++k) jc.axis[k] = d->axis[3 * (size_t)i + k];
    jc.jtype = jt;
    jc.parent = p;
    jc.idx_q = d->idx_q[i];
    jc.idx_v = d->idx_v[i];
    jc.nvj = nvj;
    jc.nqj = nqj;
    jc.comp = 0;
    if (jt <= BRBD_JOINT_RZ) jc.comp = 3 + jt;
    else if (jt <= BRBD_JOINT_PZ) jc.comp = jt - BRBD_JOINT_PX;
    else if (jt >= BRBD_JOINT_RUBX && jt <= BRBD_JOINT_RUBZ) jc.comp = 3 + (jt - BRBD_JOINT_RUBX);
    jc.hand_off = hand;
    hand += hand_size(nvj);
  }
  for (char c : qcover)
    if (!c) return "idx_q ranges do not cover nq";
  for (char c : vcover)
    if (!c) return "idx_v ranges do not cover nv";
  M.hand_total = hand;
  M.tab_desc = M.tab;
  std::string err = canonicalise_unaligned(M, children);
  if (!err.empty()) return err;

  // depth, branch slots
  std::vector<int> depth(n, -1), bslot(n, -1);
  int maxdepth = 0, maxbranch = 0;
  for (int i = 1; i < n; ++i) {
    const int p = d->parents[i];
    depth[i] = p == 0 ? 0 : depth[p] + 1;
    if (depth[i] + 1 > maxdepth) maxdepth = depth[i] + 1;
  }
  // own branch slot = number of branching proper ancestors (unique along any root path)
  for (int i = 1; i < n; ++i) {
    const int p = d->parents[i];
    const int anc = p == 0 ? 0 : (bslot[p] + (children[p].size() >= 2 ? 1 : 0));
    bslot[i] = anc;  // slot this joint WOULD use if it branches
    if (children[i].size() >= 2 && anc + 1 > maxbranch) maxbranch = anc + 1;
  }
  M.maxdepth = maxdepth > 0 ? maxdepth : 1;
  M.maxbranch = maxbranch > 0 ? maxbranch : 1;

  // depth-first program, children in decreasing index order
  struct Frame { int joint; size_t next; };
  std::vector<Frame> stack;
  int last_fwd = -1;  // joint whose forward state is in registers
  auto fill_ints = [&](Op& op, int i) {
    const JointConst& jc = M.tab[i];
    op.jtype = jc.jtype; op.idx_q = jc.idx_q; op.idx_v = jc.idx_v; op.nvj = jc.nvj;
    op.comp = jc.comp; op.hand_off = jc.hand_off; op.pad0 = 0; op.pad1 = 0;
  };
  auto emit_fwd = [&](int i) {
    const int p = d->parents[i];
    Op op;
    op.joint = i;
    op.depth = depth[i];
    op.bslot = p == 0 ? 0 : bslot[p];
    fill_ints(op, i);
    int fl = 0;
    if (p == 0) fl |= F_PARENT_ROOT;
    else if (last_fwd == p) fl |= F_FROM_REGS;
    if (children[i].size() >= 2) fl |= F_PUSH_BRANCH;
    if (children[i].empty()) {
      fl |= F_LEAF;
      if (p != 0 && children[p].size() == 1) fl |= F_CARRY_REGS;
    }
    if (M.tab_desc[i].jtype != M.tab[i].jtype) fl |= F_POST;
    fl |= bslot[i] << 16;
    op.flags = fl;
    M.prog.push_back(op);
    M.prog3.push_back(op);
    last_fwd = i;
  };
  auto emit_bwd = [&](int i) {
    const int p = d->parents[i];
    Op op;
    op.joint = i;
    op.depth = depth[i];
    op.bslot = p == 0 ? 0 : bslot[p];
    fill_ints(op, i);
    int fl = OP_BWD;
    if (p == 0) fl |= F_PARENT_ROOT;
    else if (children[p].size() == 1) fl |= F_CARRY_REGS;
    if (children[i].size() == 1) fl |= F_CARRY_IN;
    else fl |= F_MULTI;
    if (M.tab_desc[i].jtype != M.tab[i].jtype) fl |= F_POST;
    fl |= bslot[i] << 16;
    op.flags = fl;
    M.prog.push_back(op);
    last_fwd = -1;
  };
  for (int ri = (int)children[0].size() - 1; ri >= 0; --ri) {
    const int r = children[0][ri];
    emit_fwd(r);
    if (children[r].empty()) continue;
    stack.push_back({r, children[r].size()});
    while (!stack.empty()) {
      Frame& f = stack.back();
      if (f.next == 0) {
        emit_bwd(f.joint);
        stack.pop_back();
        continue;
      }
      const int c = children[f.joint][--f.next];
      emit_fwd(c);
      if (!children[c].empty()) stack.push_back({c, children[c].size()});
    }
  }

  // subtree contiguity (multibody/data.hxx:205-230 assumes it)
  M.nv_subtree.assign(n, 0);
  std::vector<int> last(n);
  for (int i = 0; i < n; ++i) last[i] = i;
  for (int i = n - 1; i >= 1; --i) {
    const int p = d->parents[i];
    if (last[i] > last[p]) last[p] = last[i];
  }
  M.depth_first = true;
  for (int i = 1; i < n && M.depth_first; ++i)
    for (int j = i + 1; j <= last[i]; ++j) {
      int p = j;
      while (p > i) p = d->parents[p];
      if (p != i) { M.depth_first = false; break; }
    }
  for (int i = 1; i < n; ++i)
    M.nv_subtree[i] = d->idx_v[last[i]] + kJointNv[d->jtype[last[i]]] - d->idx_v[i];
  *status = BRBD_OK;
  return "";
}

inline std::string canonicalise_unaligned(FlatModel& M, const std::vector<std::vector<int>>& children) {
  auto mul = [](const double* A, const double* B, double* C) {  // row-major 3x3
    double T[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) T[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
    for (int k = 0; k < 9; ++k) C[k] = T[k];
  };
  for (int i = 1; i < M.njoints; ++i) {
    JointConst& jc = M.tab[i];
    const bool ru = jc.jtype == BRBD_JOINT_REVOLUTE_UNALIGNED, pu = jc.jtype == BRBD_JOINT_PRISMATIC_UNALIGNED;
    const bool rubu = jc.jtype == BRBD_JOINT_REVOLUTE_UNBOUNDED_UNALIGNED;  // -> RUBZ: q = (cos, sin) as is
    if (!ru && !pu && !rubu) continue;
    double u[3] = {jc.axis[0], jc.axis[1], jc.axis[2]};
    const double nrm = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (!(nrm > 0.0)) return "joint " + std::to_string(i) + ": unaligned joint with a zero axis";
    for (int k = 0; k < 3; ++k) u[k] /= nrm;
    // right-handed orthonormal basis (a, b, u): a = u x e_min normalised, b = u x a
    int kmin = 0;
    for (int k = 1; k < 3; ++k)
      if (std::fabs(u[k]) < std::fabs(u[kmin])) kmin = k;
    double e[3] = {0, 0, 0}, a[3], b[3];
    e[kmin] = 1.0;
    a[0] = u[1] * e[2] - u[2] * e[1]; a[1] = u[2] * e[0] - u[0] * e[2]; a[2] = u[0] * e[1] - u[1] * e[0];
    const double na = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    for (int k = 0; k < 3; ++k) a[k] /= na;
    b[0] = u[1] * a[2] - u[2] * a[1]; b[1] = u[2] * a[0] - u[0] * a[2]; b[2] = u[0] * a[1] - u[1] * a[0];
    double Q[9], Qt[9];
    for (int r = 0; r < 3; ++r) { Q[3 * r] = a[r]; Q[3 * r + 1] = b[r]; Q[3 * r + 2] = u[r]; }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Qt[3 * r + c] = Q[3 * c + r];
    mul(jc.R, Q, jc.R);  // placement' = placement Q
    // inertia in the new frame: c' = Q^T c, I' = Q^T I Q
    const double c0[3] = {jc.c[0], jc.c[1], jc.c[2]};
    for (int r = 0; r < 3; ++r) jc.c[r] = Qt[3 * r] * c0[0] + Qt[3 * r + 1] * c0[1] + Qt[3 * r + 2] * c0[2];
    double I3[9] = {jc.I[0], jc.I[1], jc.I[3], jc.I[1], jc.I[2], jc.I[4], jc.I[3], jc.I[4], jc.I[5]}, T[9];
    mul(Qt, I3, T);
    mul(T, Q, I3);
    jc.I[0] = I3[0]; jc.I[1] = 0.5 * (I3[1] + I3[3]); jc.I[2] = I3[4];
    jc.I[3] = 0.5 * (I3[2] + I3[6]); jc.I[4] = 0.5 * (I3[5] + I3[7]); jc.I[5] = I3[8];
    for (int k = 0; k < 9; ++k) jc.post[k] = Qt[k];
    jc.jtype = ru ? BRBD_JOINT_RZ : (rubu ? BRBD_JOINT_RUBZ : BRBD_JOINT_PZ);
    jc.comp = pu ? 2 : 5;
    for (int ch : children[i]) {  // placement_child' = Q^T . placement_child
      JointConst& cc = M.tab[ch];
      mul(Qt, cc.R, cc.R);
      const double t0[3] = {cc.t[0], cc.t[1], cc.t[2]};
      for (int r = 0; r < 3; ++r) cc.t[r] = Qt[3 * r] * t0[0] + Qt[3 * r + 1] * t0[1] + Qt[3 * r + 2] * t0[2];
    }
  }
  return "";
}

}  // namespace brbd

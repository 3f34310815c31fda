// brbd_kernels.hpp -- batch kernels: RNEA, ABA (WORLD), FK + Jacobians.
//
// Execution model
//   * one lane = one instance, one wave (64 lanes) = one workgroup, grid-stride over tiles of 64
//     instances, so any per-wave spill workspace is bounded by the grid, not by the batch;
//   * the kinematic tree is walked by a depth-first PROGRAM built by the flattener
//     (brbd_flatten.hpp): a joint's forward visit, its subtree, then its backward visit.  Only the
//     joints on the current root path are live; chains hand state parent->child and
//     child->parent in registers;
//   * every control decision (op kind, joint kind, flags) is wave-uniform: no divergence.
//
// These are the TABLE-DRIVEN kernels: they interpret the program at run time for any model; the
// live path state is kept in LDS and a wave-private, coalesced global workspace.  Models registered
// at build time (or through a kernel plugin) run on the model-compiled kernels instead
// (brbd_spec.hpp + pinocchio_amd/codegen): the same recursions as straight-line code.
//
// Reference recursions: algorithm/rnea.hxx:45-161, algorithm/aba.hxx:102-294,
// algorithm/kinematics.hxx:52-97, algorithm/jacobian.hxx:38-95.
#pragma once
#include "brbd_device.hpp"

namespace brbd {

constexpr int WAVE = 64;

struct KArgs {
  const JointConst* tab;
  const Op* prog;   // pass 1+2 program (FWD/BWD ops)
  const Op* prog3;  // forward-only program (ABA pass 3, FK/Jacobians)
  int32_t nops, nops3;
  int32_t njoints, nq, nv;
  int32_t maxdepth, maxbranch;
  int64_t B;
  int64_t ntiles;
  double* ws;         // spill workspace
  int32_t lds_cap;    // live-state slots held in LDS (HybridStore), rest in ws
  int32_t rf;         // jjac_kernel: reference frame
  int64_t ws_stride;  // doubles per wave
  double gx, gy, gz;  // gravity
  const double* q; int64_t ldq;
  const double* v; int64_t ldv;
  const double* a; int64_t lda;  // RNEA: a, ABA: tau
  double* out; int64_t ldout;    // RNEA: tau, ABA: ddq, FKJ: J
  double* out2; int64_t ldout2;  // FKJ: oMi
  const double* fext; int64_t ldfext;  // external forces, 6 rows [lin; ang] per joint in its frame, or NULL
  const double* ctab;            // model-compiled kernels (brbd_spec.hpp): packed constant table
  int64_t tile0;                 //   first tile of this launch (the host launches in chunks)
  double* out3; int64_t ldout3;  //   RNEA derivatives: dtau/da
  double* out4; int64_t ldout4;  //   RNEA derivatives: tau
  uint32_t nb[8];                //   bytes spanned by q, v, a/tau, out, out2, out3, out4, fext (0 = absent)
  uint32_t ws_bytes;             //   bytes of the global park area of this launch
  unsigned long long* oob;       //   checking build (-DBRBD_SPEC_CHECK): out-of-range access counter, else NULL
  unsigned long long chain;      //   compiled getJointJacobian: bit i = joint i is the joint or one of its ancestors
};

// ---- boundary matrices -------------------------------------------------------------------------
// Element (d, b) of a dof x B matrix in the two layouts of include/brbd.h.  Lanes past the end of
// the batch are clamped onto the last instance: they recompute and re-store exactly its values,
// which keeps every load/store unpredicated and the control flow uniform.
template <int LAYOUT>
struct Mat;
template <>
struct Mat<1> {  // BATCH_MAJOR: uniform row base (SGPR pair) + 32-bit lane offset (saddr addressing)
  double* p;     // base + tile * 64
  int64_t ld;
  unsigned lo8;  // clamped lane, in bytes (zero-extended 32-bit offset -> saddr addressing)
  int64_t t0;    // tile * 64
  DI Mat(double* base, int64_t ld_, int64_t tile, int64_t B, int lane) : p(base + tile * WAVE), ld(ld_), t0(tile * WAVE) {
    const int64_t left = B - tile * WAVE;  // >= 1
    lo8 = ((int64_t)lane < left ? (unsigned)lane : (unsigned)(left - 1)) * 8u;
  }
  DI Mat other(const double* base, int64_t ld_) const {  // another matrix, same tile and lane
    Mat m = *this;
    m.p = const_cast<double*>(base) + t0;
    m.ld = ld_;
    return m;
  }
  DI double get(int d) const {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(p + (int64_t)d * ld) + lo8);
  }
  DI void put(int d, double x) const {
    *reinterpret_cast<double*>(reinterpret_cast<char*>(p + (int64_t)d * ld) + lo8) = x;
  }
};
template <>
struct Mat<0> {  // INSTANCE_MAJOR: per-lane row pointer, uniform element index
  double* row;
  int64_t b;
  DI Mat(double* base, int64_t ld_, int64_t tile, int64_t B, int lane) {
    b = tile * WAVE + lane;
    if (b >= B) b = B - 1;
    row = base + b * ld_;
  }
  DI Mat other(const double* base, int64_t ld_) const {  // another matrix, same instance
    Mat m = *this;
    m.row = const_cast<double*>(base) + b * ld_;
    return m;
  }
  DI double get(int d) const { return row[d]; }
  DI void put(int d, double x) const { row[d] = x; }
};

// External force on the joint of `op`, in the frame the kernels use for it (rnea.hxx:205,
// aba.hxx:338).  The caller gives it in the described joint frame; a canonicalised unaligned joint
// (brbd_flatten.hpp) lives in that frame times Q, so the force is rotated by post = Q^T.
template <class MatT>
DI void load_fext(const KArgs& A, const Op op, const JointConst& jc, const MatT& like, double* f) {
  const MatT fx = like.other(A.fext, A.ldfext);
#pragma unroll
  for (int k = 0; k < 6; ++k) f[k] = fx.get(6 * op.joint + k);
  if (op.flags & F_POST) {
    double r[6];
    rot_vec(jc.post, f, r);
    rot_vec(jc.post, f + 3, r + 3);
#pragma unroll
    for (int k = 0; k < 6; ++k) f[k] = r[k];
  }
}

// ---- live-state storage policies -----------------------------------------------------------------
// GlobalStore: slot s of this lane lives at base[s * 64 + lane]: one coalesced 512-byte row per
// slot, uniform base + lane offset.
struct GlobalStore {
  double* base;
  unsigned lane8;  // lane * 8 bytes
  DI GlobalStore(double* b, unsigned lane) : base(b), lane8(lane * 8u) {}
  DI double ld(int slot) const {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base + (int64_t)slot * WAVE) + lane8);
  }
  DI void st(int slot, double x) const {
    *reinterpret_cast<double*>(reinterpret_cast<char*>(base + (int64_t)slot * WAVE) + lane8) = x;
  }
};
// RegStore: slots are elements of a local array; with a compile-time program every index is a
// constant after inlining, SROA turns the array into registers and the register allocator keeps
// exactly the live path in VGPRs (anything beyond 512 VGPRs falls back to scratch by itself).
template <int N>
struct RegStore {
  double r[N > 0 ? N : 1];
  DI double ld(int slot) const { return r[slot]; }
  DI void st(int slot, double x) { r[slot] = x; }
};

// HybridStore: the first `cap` slots live in LDS (one 512-byte row per slot, ~100-cycle access
// instead of an L2 round trip), the rest in the wave-private global workspace.  `cap` is chosen by
// the host so that the LDS footprint never lowers the occupancy the VGPR budget allows.
struct HybridStore {
  double* lds;  // &smem[lane]; slot s at lds[s * 64]
  int cap;
  GlobalStore g;
  DI HybridStore(double* smem, int cap_, double* gbase, unsigned lane) : lds(smem + lane), cap(cap_), g(gbase, lane) {}
  DI double ld(int slot) const { return slot < cap ? lds[slot * WAVE] : g.ld(slot - cap); }
  DI void st(int slot, double x) const {
    if (slot < cap) lds[slot * WAVE] = x;
    else g.st(slot - cap, x);
  }
};

// First element of q, v and a (RNEA) / tau (ABA) of an op's joint, fetched one op ahead of its
// use so that the HBM latency of the inputs hides behind the previous op's arithmetic.
struct Pre {
  double q0, v0, x0;
};
template <class MatT>
DI Pre prefetch(const Op& op, const MatT& q, const MatT& v, const MatT& x) {
  Pre p;
  p.q0 = q.get(op.idx_q);
  p.v0 = v.get(op.idx_v);
  p.x0 = x.get(op.idx_v);
  return p;
}

// ---- joint kinematics: liMi = placement * M(q) ---------------------------------------------------
// have_sc: (s, c) already known (backward visit / pass 3), skip sincos.  q0 = q[idx_q] (prefetched).
template <class MatT>
DI void joint_liMi(const JointConst& jc, const Op& op, const MatT& q, double q0, bool have_sc, double& s,
                   double& c, double* R, double* t) {
  const int jt = op.jtype, iq = op.idx_q;
  switch (jt) {
    case JT_RX: case JT_RY: case JT_RZ: case JT_RUBX: case JT_RUBY: case JT_RUBZ: {
      if (!have_sc) {
        if (jt <= JT_RZ) {
          sincos(q0, &s, &c);  // math/sincos.hpp:47-61
        } else {
          c = q0;  // joint-revolute-unbounded.hpp:153-162
          s = q.get(iq + 1);
        }
      }
      const int k = jt <= JT_RZ ? jt - JT_RX : jt - JT_RUBX;
      if (k == 0) revolute_liMi<0>(jc.R, s, c, R);
      else if (k == 1) revolute_liMi<1>(jc.R, s, c, R);
      else revolute_liMi<2>(jc.R, s, c, R);
      t[0] = jc.t[0]; t[1] = jc.t[1]; t[2] = jc.t[2];
      break;
    }
    case JT_PX: case JT_PY: case JT_PZ: {
      // joint-prismatic.hpp:262-272
      const int k = jt - JT_PX;
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = jc.R[i];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double r0 = jc.R[3 * r], r1 = jc.R[3 * r + 1], r2 = jc.R[3 * r + 2];
        t[r] = jc.t[r] + q0 * sel3(k, r0, r1, r2);
      }
      break;
    }
    case JT_FF: {
      // joint-free-flyer.hpp:332-374
      double Rq[9], p[3], Rp[3];
      quat_to_R(q.get(iq + 3), q.get(iq + 4), q.get(iq + 5), q.get(iq + 6), Rq);
      p[0] = q0; p[1] = q.get(iq + 1); p[2] = q.get(iq + 2);
      rot_mul(jc.R, Rq, R);
      rot_vec(jc.R, p, Rp);
      t[0] = jc.t[0] + Rp[0]; t[1] = jc.t[1] + Rp[1]; t[2] = jc.t[2] + Rp[2];
      break;
    }
    default: {  // JT_SPH, joint-spherical.hpp:525-589
      double Rq[9];
      quat_to_R(q0, q.get(iq + 1), q.get(iq + 2), q.get(iq + 3), Rq);
      rot_mul(jc.R, Rq, R);
      t[0] = jc.t[0]; t[1] = jc.t[1]; t[2] = jc.t[2];
      break;
    }
  }
}

// Dense S * x for the joint (x = joint velocity or acceleration read from `m`; x0 = m[idx_v])
template <class MatT>
DI void joint_S_times(const Op& op, const MatT& m, double x0, double* o) {
  const int iv = op.idx_v;
  if (op.nvj == 1) {
    const int comp = op.comp;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = (k == comp) ? x0 : 0.0;
  } else if (op.nvj == 6) {
    o[0] = x0;
#pragma unroll
    for (int k = 1; k < 6; ++k) o[k] = m.get(iv + k);
  } else {
    o[0] = o[1] = o[2] = 0.0;
    o[3] = x0; o[4] = m.get(iv + 1); o[5] = m.get(iv + 2);
  }
}

// =================================================================================== RNEA
// live-state slots per lane: F[depth][8] = f(6) + (s, c);  VA[bslot][12] = v(6) + a_gf(6)
struct RneaState {
  double v[6], a[6], carry[6];
};
DI void rnea_state_init(RneaState& st) {
#pragma unroll
  for (int k = 0; k < 6; ++k) st.v[k] = st.a[k] = st.carry[k] = 0.0;
}

template <class MatT, class Stack>
DI void rnea_step(const KArgs& A, const Op op, const Pre pre, Stack& S, const int VA0, const MatT& q,
                  const MatT& vin, const MatT& ain, const MatT& tau, RneaState& st) {
  JointConst jc;
  load_joint(A.tab, op.joint, jc);
  const int fl = op.flags;
  double R[9], t[3], f[6], sn = 0.0, cs = 1.0;
  if (!(fl & OP_BWD)) {
    // ---------------- forward visit (rnea.hxx:45-79) ----------------
    double vp[6], ap[6];
    if (fl & F_PARENT_ROOT) {  // rnea.hxx:137-138: v[0] = 0, a_gf[0] = -gravity
#pragma unroll
      for (int k = 0; k < 6; ++k) vp[k] = ap[k] = 0.0;
      ap[0] = -A.gx; ap[1] = -A.gy; ap[2] = -A.gz;
    } else if (fl & F_FROM_REGS) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { vp[k] = st.v[k]; ap[k] = st.a[k]; }
    } else {
      const int base = VA0 + op.bslot * 12;
#pragma unroll
      for (int k = 0; k < 6; ++k) { vp[k] = S.ld(base + k); ap[k] = S.ld(base + 6 + k); }
    }
    joint_liMi(jc, op, q, pre.q0, false, sn, cs, R, t);
    double vJ[6], Sa[6], tmp[6];
    joint_S_times(op, vin, pre.v0, vJ);
    joint_S_times(op, ain, pre.x0, Sa);
    if (fl & F_PARENT_ROOT) {  // :63-65, velocity propagation guarded by parent > 0
#pragma unroll
      for (int k = 0; k < 6; ++k) st.v[k] = vJ[k];
    } else {
      se3_actinv_motion(R, t, vp, tmp);
#pragma unroll
      for (int k = 0; k < 6; ++k) st.v[k] = vJ[k] + tmp[k];
    }
    motion_cross(st.v, vJ, st.a);      // :67 (c = 0)
    se3_actinv_motion(R, t, ap, tmp);  // :69 (not guarded)
#pragma unroll
    for (int k = 0; k < 6; ++k) st.a[k] = (st.a[k] + Sa[k]) + tmp[k];
    double h[6];
    inertia_mul(jc.m, jc.c, jc.I, st.v, h);  // :73
    inertia_mul(jc.m, jc.c, jc.I, st.a, f);  // :74
    force_cross(st.v, h, tmp);               // :75
#pragma unroll
    for (int k = 0; k < 6; ++k) f[k] += tmp[k];
    if (A.fext) {  // rnea.hxx:205
      load_fext(A, op, jc, q, tmp);
#pragma unroll
      for (int k = 0; k < 6; ++k) f[k] -= tmp[k];
    }
    if (fl & F_PUSH_BRANCH) {
      const int base = VA0 + op_own_bslot(fl) * 12;
#pragma unroll
      for (int k = 0; k < 6; ++k) { S.st(base + k, st.v[k]); S.st(base + 6 + k, st.a[k]); }
    }
    if (!(fl & F_LEAF)) {
      const int base = op.depth * 8;
#pragma unroll
      for (int k = 0; k < 6; ++k) S.st(base + k, f[k]);
      S.st(base + 6, sn);
      S.st(base + 7, cs);
      return;
    }
  } else {
    // ---------------- backward visit of a non-leaf joint (rnea.hxx:92-107) ----------------
    const int base = op.depth * 8;
#pragma unroll
    for (int k = 0; k < 6; ++k) f[k] = S.ld(base + k);
    sn = S.ld(base + 6);
    cs = S.ld(base + 7);
    if (fl & F_CARRY_IN) {
#pragma unroll
      for (int k = 0; k < 6; ++k) f[k] += st.carry[k];
    }
    joint_liMi(jc, op, q, pre.q0, true, sn, cs, R, t);
  }
  // ---- common tail: tau = S^T f (+ armature * a), hand f to the parent ----
  const int iv = op.idx_v;
  if (op.nvj == 1) {
    const int comp = op.comp;
    double x = f[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) {
      const double fk = f[k];
      x = (k == comp) ? fk : x;
    }
    x += jc.arm[0] * pre.x0;  // rnea.hxx:158
    tau.put(iv, x);
  } else if (op.nvj == 6) {
#pragma unroll
    for (int k = 0; k < 6; ++k) tau.put(iv + k, f[k] + jc.arm[k] * ain.get(iv + k));
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) tau.put(iv + k, f[3 + k] + jc.arm[k] * ain.get(iv + k));
  }
  if (!(fl & F_PARENT_ROOT)) {  // :105-106
    double up[6];
    se3_act_force(R, t, f, up);
    if (fl & F_CARRY_REGS) {
#pragma unroll
      for (int k = 0; k < 6; ++k) st.carry[k] = up[k];
    } else {
      const int pbase = (op.depth - 1) * 8;
#pragma unroll
      for (int k = 0; k < 6; ++k) S.st(pbase + k, S.ld(pbase + k) + up[k]);
    }
  }
}

extern __shared__ double brbd_smem[];

template <int LAYOUT>
__global__ __launch_bounds__(WAVE) void rnea_kernel(const KArgs A) {
  const int lane = threadIdx.x;
  HybridStore S(brbd_smem, A.lds_cap, A.ws + (int64_t)blockIdx.x * A.ws_stride, (unsigned)lane);
  const int VA0 = A.maxdepth * 8;
  for (int64_t tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    const Mat<LAYOUT> q(const_cast<double*>(A.q), A.ldq, tile, A.B, lane);
    const Mat<LAYOUT> vin(const_cast<double*>(A.v), A.ldv, tile, A.B, lane);
    const Mat<LAYOUT> ain(const_cast<double*>(A.a), A.lda, tile, A.B, lane);
    const Mat<LAYOUT> tau(A.out, A.ldout, tile, A.B, lane);
    RneaState st;
    rnea_state_init(st);
    Op op = load_op(A.prog, 0);
    Pre pre = prefetch(op, q, vin, ain);
    for (int s = 0; s < A.nops; ++s) {
      const Op nxt = load_op(A.prog, s + 1 < A.nops ? s + 1 : s);
      const Pre pn = prefetch(nxt, q, vin, ain);
      rnea_step(A, op, pre, S, VA0, q, vin, ain, tau, st);
      op = nxt;
      pre = pn;
    }
  }
}

// =================================================================================== ABA (WORLD)
// J columns of a joint in the world frame: oMi.act(S) (aba.hxx:124, jacobian.hxx:59)
// 1-dof aligned: revolute k -> [t x R[:,k]; R[:,k]], prismatic k -> [R[:,k]; 0]
DI void jcol_1dof(const Op& op, const double* R, const double* t, double* J) {
  const int comp = op.comp;
  const int k = comp >= 3 ? comp - 3 : comp;
  double col[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // select among VALUES: a ternary on the array lvalues becomes a dynamically indexed load and
    // drags the whole state struct into scratch memory
    const double r0 = R[3 * r], r1 = R[3 * r + 1], r2 = R[3 * r + 2];
    col[r] = sel3(k, r0, r1, r2);
  }
  if (comp >= 3) {
    cross3(t, col, J);
    J[3] = col[0]; J[4] = col[1]; J[5] = col[2];
  } else {
    J[0] = col[0]; J[1] = col[1]; J[2] = col[2];
    J[3] = J[4] = J[5] = 0.0;
  }
}
// multi-dof: FF (6 columns: action matrix, se3-tpl.hpp:178-191) or spherical (3 angular columns)
template <int NV>
DI void jcols_multi(const double* R, const double* t, double* J /* NV x 6 */) {
  constexpr int first_ang = NV == 6 ? 3 : 0;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    double col[3];
    const int j = c >= first_ang ? c - first_ang : c;
#pragma unroll
    for (int r = 0; r < 3; ++r) col[r] = R[3 * r + j];
    if (c >= first_ang) {
      cross3(t, col, J + 6 * c);
      J[6 * c + 3] = col[0]; J[6 * c + 4] = col[1]; J[6 * c + 5] = col[2];
    } else {
      J[6 * c + 0] = col[0]; J[6 * c + 1] = col[1]; J[6 * c + 2] = col[2];
      J[6 * c + 3] = J[6 * c + 4] = J[6 * c + 5] = 0.0;
    }
  }
}

// oMi = oMp * liMi (se3-tpl.hpp:311-315); root: oMi = liMi (aba.hxx:119-122)
DI void compose(bool root, const double* Rp, const double* tp, const double* Rl, const double* tl, double* R,
                double* t) {
  if (root) {
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rl[i];
    t[0] = tl[0]; t[1] = tl[1]; t[2] = tl[2];
  } else {
    double Rt[3];
    rot_mul(Rp, Rl, R);
    rot_vec(Rp, tl, Rt);
    t[0] = tp[0] + Rt[0]; t[1] = tp[1] + Rt[1]; t[2] = tp[2] + Rt[2];
  }
}

// own world inertia (packed sym6) and bias force of = ov x* (Y ov)  (aba.hxx:134-138)
template <class MatT>
DI void aba_own_inertia(const KArgs& A, const Op op, const MatT& q, const JointConst& jc, const double* R,
                        const double* t, const double* ov, double* Ia, double* of) {
  double cw[3], Iw[6], h[6];
  inertia_to_world(R, t, jc.m, jc.c, jc.I, cw, Iw);
  inertia_to_sym6(jc.m, cw, Iw, Ia);
  inertia_mul(jc.m, cw, Iw, ov, h);
  force_cross(ov, h, of);
  if (A.fext) {  // aba.hxx:338: of[i] -= oMi.act(fext[i])
    double fe[6], fw[6];
    load_fext(A, op, jc, q, fe);
    se3_act_force(R, t, fe, fw);
#pragma unroll
    for (int k = 0; k < 6; ++k) of[k] -= fw[k];
  }
}

// live-state slots per lane (ABA), "stack" part (LDS first, see HybridStore):
//   SC[depth][2]     = (sin, cos) of the non-leaf joints on the current root path
//   ACC[bslot][27]   = Ia (21) + of (6) of the branching joints on the path: own + children so far;
//                      pass 3 reuses the same rows for K3[bslot][24] = oMi (12) + ov (6) + a (6)
// and, always in the global workspace (it lives from pass 2 to pass 3 for every joint):
//   HAND[...]        = per joint hand-off record (hand_off from the flattener)
//
// The world placement oMi and velocity ov of the joints on the path are NOT stacked: on the way
// back up they are recovered from the child's by undoing the composition
//   oMp = oMi * liMi^-1,  ov_p = ov_i - oMi.act(vJ_i)
// (45 flops instead of an 18-double round trip through memory per joint; the difference to the
// values of the forward sweep is rounding-level, ~1e-16 per joint).
struct AbaLayout {
  int sc0, acc0, end;
};
__host__ __device__ constexpr AbaLayout aba_layout(int maxdepth, int maxbranch) {
  AbaLayout L{};
  L.sc0 = 0;
  L.acc0 = L.sc0 + maxdepth * 2;
  L.end = L.acc0 + maxbranch * 27;
  return L;
}

// hand-off record sizes (doubles) by joint dof count; +2 for the cached (s, c) at the end
__host__ __device__ constexpr int hand_size(int nvj) {
  return (nvj == 1 ? 7 : 6 * nvj + nvj * nvj + nvj) + 2;
}

// Backward visit arithmetic for an NV-dof joint (aba.hxx:153-193).  Ia/of hold the accumulated
// articulated inertia / bias force of the joint on entry and the contribution for the parent on
// exit.  Writes the pass-3 hand-off record, or ddq directly for joints attached to the universe.
template <int NV, class MatT>
DI void aba_backward(const KArgs& A, const JointConst& jc, const Op& op, const GlobalStore& H, bool root,
                     const double* J, const double* bias, const MatT& tauin, double tau0, const MatT& ddq,
                     double* Ia, double* of) {
  const int iv = op.idx_v;
  double u[NV], U[NV * 6], D[NV * NV], Dinv[NV * NV], UD[NV * 6];
#pragma unroll
  for (int c = 0; c < NV; ++c) u[c] = (c == 0 ? tau0 : tauin.get(iv + c)) - dot6(J + 6 * c, of);  // :173
#pragma unroll
  for (int c = 0; c < NV; ++c) sym6_mulvec(Ia, J + 6 * c, U + 6 * c);  // :175
#pragma unroll
  for (int r = 0; r < NV; ++r)
#pragma unroll
    for (int c = r; c < NV; ++c) {
      double d = dot6(J + 6 * r, U + 6 * c);  // :176
      if (r == c) d += jc.arm[r];             // :179
      D[NV * r + c] = d;
      D[NV * c + r] = d;
    }
  if (NV == 1) Dinv[0] = 1.0 / D[0];
  else spd_inverse<NV>(D, Dinv);  // :181
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < NV; ++r) s += U[6 * r + k] * Dinv[NV * r + c];
      UD[6 * c + k] = s;  // :182
    }
  if (root) {
    // parent is the universe: a = a_gf[0] + bias(=0) = -gravity is known now, so pass 3 of this
    // joint collapses to ddq = Dinv u - UDinv^T a (aba.hxx:223-226)
    double a0[6] = {-A.gx, -A.gy, -A.gz, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < NV; ++c) s += Dinv[NV * r + c] * u[c];
      s -= dot6(UD + 6 * r, a0);
      ddq.put(iv + r, s);
    }
    return;
  }
  // hand-off record for pass 3: UDinv (6 NV), then NV==1: Dinv*u ; NV>1: Dinv (NV*NV), u (NV)
  const int h = op.hand_off;
#pragma unroll
  for (int k = 0; k < 6 * NV; ++k) H.st(h + k, UD[k]);
  if (NV == 1) {
    H.st(h + 6, Dinv[0] * u[0]);
  } else {
#pragma unroll
    for (int k = 0; k < NV * NV; ++k) H.st(h + 6 * NV + k, Dinv[k]);
#pragma unroll
    for (int k = 0; k < NV; ++k) H.st(h + 6 * NV + NV * NV + k, u[k]);
  }
  // :186 Ia -= UDinv U^T
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < NV; ++c) s += UD[6 * c + i] * U[6 * c + j];
      Ia[sidx(i, j)] -= s;
    }
  // :188-189 of += Ia * a_gf + UDinv * u
  double t1[6];
  sym6_mulvec(Ia, bias, t1);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double s = t1[k];
#pragma unroll
    for (int c = 0; c < NV; ++c) s += UD[6 * c + k] * u[c];
    of[k] += s;
  }
}

// Joints attached to the universe (typically the free-flyer): no parent to update and
// a = a_gf[0] + bias(= 0) = -gravity is already known, so pass 3 of the joint collapses into the
// backward visit: ddq = Dinv (u - U^T a) (aba.hxx:173-181,223-226), solved by a Cholesky
// factorisation of D = J^T Ia J + armature without forming Dinv / UDinv.
template <int NV, class MatT>
DI void aba_root_solve(const KArgs& A, const JointConst& jc, const Op& op, const double* J, const MatT& tauin,
                       double tau0, const MatT& ddq, const double* Ia, const double* of) {
  const int iv = op.idx_v;
  double U[NV * 6], L[NV * NV], x[NV];
  const double a0[6] = {-A.gx, -A.gy, -A.gz, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    sym6_mulvec(Ia, J + 6 * c, U + 6 * c);
    x[c] = ((c == 0 ? tau0 : tauin.get(iv + c)) - dot6(J + 6 * c, of)) - dot6(U + 6 * c, a0);
  }
  // Cholesky D = L L^T, lower triangle, row-major
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    double d = dot6(J + 6 * j, U + 6 * j) + jc.arm[j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[NV * j + k] * L[NV * j + k];
    const double rd = NV == 1 ? 1.0 / d : rsqrt(d);
    L[NV * j + j] = rd;  // 1 / L_jj (NV == 1: 1 / D)
#pragma unroll
    for (int i = j + 1; i < NV; ++i) {
      double sv = dot6(J + 6 * i, U + 6 * j);
#pragma unroll
      for (int k = 0; k < j; ++k) sv -= L[NV * i + k] * L[NV * j + k];
      L[NV * i + j] = sv * rd;
    }
  }
  if (NV == 1) {
    ddq.put(iv, x[0] * L[0]);
    return;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {  // L y = x
    double sv = x[i];
#pragma unroll
    for (int k = 0; k < i; ++k) sv -= L[NV * i + k] * x[k];
    x[i] = sv * L[NV * i + i];
  }
#pragma unroll
  for (int i = NV - 1; i >= 0; --i) {  // L^T z = y
    double sv = x[i];
#pragma unroll
    for (int k = i + 1; k < NV; ++k) sv -= L[NV * k + i] * x[k];
    x[i] = sv * L[NV * i + i];
  }
#pragma unroll
  for (int r = 0; r < NV; ++r) ddq.put(iv + r, x[r]);
}

// Multi-dof joints in the MIDDLE of a tree (spherical shoulders, nested free-flyers) are rare; their
// backward visit is kept out of line so that its 6x6 temporaries do not set the register budget of
// the 1-dof main path.  Arguments travel through private memory on purpose.
template <int NV, class MatT>
__device__ __noinline__ void aba_backward_multi(const KArgs* A, const JointConst* jc, const Op* op,
                                                const GlobalStore* H, const double* R, const double* t,
                                                const double* bias, const MatT* tauin, double tau0,
                                                const MatT* ddq, double* Ia, double* of) {
  double J[6 * NV];
  jcols_multi<NV>(R, t, J);
  aba_backward<NV>(*A, *jc, *op, *H, false, J, bias, *tauin, tau0, *ddq, Ia, of);
}

template <class MatT>
DI void aba_backward_any(const KArgs& A, const JointConst& jc, const Op& op, const GlobalStore& H, bool root,
                         const double* R, const double* t, const double* bias, const MatT& tauin, double tau0,
                         const MatT& ddq, double* Ia, double* of) {
  if (op.nvj == 1) {
    double J[6];
    jcol_1dof(op, R, t, J);
    if (root) aba_root_solve<1>(A, jc, op, J, tauin, tau0, ddq, Ia, of);
    else aba_backward<1>(A, jc, op, H, false, J, bias, tauin, tau0, ddq, Ia, of);
  } else if (root) {
    if (op.nvj == 6) {
      double J[36];
      jcols_multi<6>(R, t, J);
      aba_root_solve<6>(A, jc, op, J, tauin, tau0, ddq, Ia, of);
    } else {
      double J[18];
      jcols_multi<3>(R, t, J);
      aba_root_solve<3>(A, jc, op, J, tauin, tau0, ddq, Ia, of);
    }
  } else {
    // private copies: only these escape to the out-of-line call
    double cR[9], ct[3], cb[6], cIa[21], cof[6];
#pragma unroll
    for (int k = 0; k < 9; ++k) cR[k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) ct[k] = t[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { cb[k] = bias[k]; cof[k] = of[k]; }
#pragma unroll
    for (int k = 0; k < 21; ++k) cIa[k] = Ia[k];
    const KArgs cA = A;
    const JointConst cj = jc;
    const Op cop = op;
    const GlobalStore cH = H;
    const MatT ctau = tauin, cddq = ddq;
    if (op.nvj == 6) aba_backward_multi<6>(&cA, &cj, &cop, &cH, cR, ct, cb, &ctau, tau0, &cddq, cIa, cof);
    else aba_backward_multi<3>(&cA, &cj, &cop, &cH, cR, ct, cb, &ctau, tau0, &cddq, cIa, cof);
#pragma unroll
    for (int k = 0; k < 21; ++k) Ia[k] = cIa[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) of[k] = cof[k];
  }
}

struct AbaState {
  double R[9], t[3], ov[6];  // world placement / velocity of the joint the walk currently stands on
  double cIa[21], cof[6];    // carry registers child -> parent
};
DI void aba_state_init(AbaState& st) {
#pragma unroll
  for (int k = 0; k < 9; ++k) st.R[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) st.t[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) st.ov[k] = st.cof[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 21; ++k) st.cIa[k] = 0.0;
}

// one op of passes 1 + 2 (depth-first).  Invariant: on entry of FWD(i) the state holds the
// kinematics of parent(i); on entry of BWD(i) it holds those of i; on exit of a backward visit it
// holds those of parent(i).
template <class MatT, class Stack>
DI void aba_step12(const KArgs& A, const Op op, const Pre pre, const AbaLayout L, Stack& S, const GlobalStore& H,
                   const MatT& q, const MatT& vin, const MatT& tauin, const MatT& ddq, AbaState& st) {
  JointConst jc;
  load_joint(A.tab, op.joint, jc);
  const int fl = op.flags;
  const bool root = fl & F_PARENT_ROOT;
  double Ia[21], of[6], bias[6], Rl[9], tl[3], vJ[6], ovJ[6];
  double sn = 0.0, cs = 1.0;
  joint_S_times(op, vin, pre.v0, vJ);
  if (!(fl & OP_BWD)) {
    // ---------------- forward visit (aba.hxx:102-139) ----------------
    joint_liMi(jc, op, q, pre.q0, false, sn, cs, Rl, tl);
    double Rp[9], tp[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rp[k] = st.R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) tp[k] = st.t[k];
    compose(root, Rp, tp, Rl, tl, st.R, st.t);  // :118-122
    se3_act_motion(st.R, st.t, vJ, ovJ);        // :126
    if (root) {
#pragma unroll
      for (int k = 0; k < 6; ++k) st.ov[k] = ovJ[k];
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) st.ov[k] += ovJ[k];  // :127-128
    }
    // (s, c) go to the hand-off record for pass 3 ...
    {
      const int h = op.hand_off + hand_size(op.nvj) - 2;
      H.st(h, sn);
      H.st(h + 1, cs);
    }
    if (fl & F_PUSH_BRANCH) {
      // accumulator starts with the joint's own inertia and bias force (aba.hxx:134-138)
      aba_own_inertia(A, op, q, jc, st.R, st.t, st.ov, Ia, of);
      const int ac = L.acc0 + op_own_bslot(fl) * 27;
#pragma unroll
      for (int k = 0; k < 21; ++k) S.st(ac + k, Ia[k]);
#pragma unroll
      for (int k = 0; k < 6; ++k) S.st(ac + 21 + k, of[k]);
    }
    if (!(fl & F_LEAF)) {
      // ... and to the path stack for the backward visit
      S.st(L.sc0 + 2 * op.depth, sn);
      S.st(L.sc0 + 2 * op.depth + 1, cs);
      return;
    }
    aba_own_inertia(A, op, q, jc, st.R, st.t, st.ov, Ia, of);
  } else {
    // ---------------- backward visit of a non-leaf joint (aba.hxx:153-193) ----------------
    sn = S.ld(L.sc0 + 2 * op.depth);
    cs = S.ld(L.sc0 + 2 * op.depth + 1);
    joint_liMi(jc, op, q, pre.q0, true, sn, cs, Rl, tl);
    se3_act_motion(st.R, st.t, vJ, ovJ);
    if (fl & F_MULTI) {
      const int ac = L.acc0 + op_own_bslot(fl) * 27;
#pragma unroll
      for (int k = 0; k < 21; ++k) Ia[k] = S.ld(ac + k);
#pragma unroll
      for (int k = 0; k < 6; ++k) of[k] = S.ld(ac + 21 + k);
    } else {
      aba_own_inertia(A, op, q, jc, st.R, st.t, st.ov, Ia, of);
#pragma unroll
      for (int k = 0; k < 21; ++k) Ia[k] += st.cIa[k];  // aba.hxx:190-191 (single child)
#pragma unroll
      for (int k = 0; k < 6; ++k) of[k] += st.cof[k];
    }
  }
  // ---- backward arithmetic; bias = ov_parent x ov with ov_parent = ov - oMi.act(vJ) (:126-132) ----
  double ovp[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) ovp[k] = root ? 0.0 : st.ov[k] - ovJ[k];
  if (root) {
#pragma unroll
    for (int k = 0; k < 6; ++k) bias[k] = 0.0;
  } else {
    motion_cross(ovp, st.ov, bias);
  }
  aba_backward_any(A, jc, op, H, root, st.R, st.t, bias, tauin, pre.x0, ddq, Ia, of);
  if (root) return;
  // ---- hand the articulated contribution to the parent (aba.hxx:190-191) ----
  if (fl & F_CARRY_REGS) {
#pragma unroll
    for (int k = 0; k < 21; ++k) st.cIa[k] = Ia[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) st.cof[k] = of[k];
  } else {
    const int ac = L.acc0 + op.bslot * 27;  // parent's accumulator
#pragma unroll
    for (int k = 0; k < 21; ++k) S.st(ac + k, S.ld(ac + k) + Ia[k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) S.st(ac + 21 + k, S.ld(ac + 21 + k) + of[k]);
  }
  // ---- step back onto the parent: oMp = oMi * liMi^-1, ov_p = ov_i - oMi.act(vJ_i) ----
  {
    double Rn[9], Rt[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        Rn[3 * r + c] = st.R[3 * r] * Rl[3 * c] + st.R[3 * r + 1] * Rl[3 * c + 1] + st.R[3 * r + 2] * Rl[3 * c + 2];
    rot_vec(Rn, tl, Rt);
#pragma unroll
    for (int k = 0; k < 9; ++k) st.R[k] = Rn[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) st.t[k] -= Rt[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) st.ov[k] = ovp[k];
  }
}

struct Aba3State {
  double R[9], t[3], ov[6], acc[6];
};
DI void aba3_state_init(Aba3State& st) {
#pragma unroll
  for (int k = 0; k < 9; ++k) st.R[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) st.t[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) st.ov[k] = st.acc[k] = 0.0;
}

// pass-3 prefetch: joint velocity + the 1-dof hand-off record (UDinv 6, Dinv*u, s, c)
struct Pre3 {
  double q0, v0, h[9];
};
template <class MatT>
DI Pre3 prefetch3(const Op& op, const GlobalStore& H, const MatT& q, const MatT& v) {
  Pre3 p;
  p.q0 = q.get(op.idx_q);
  p.v0 = v.get(op.idx_v);
  const int h = op.hand_off + (op.nvj == 1 ? 0 : hand_size(op.nvj) - 9);  // last 9 slots hold (.., s, c)
#pragma unroll
  for (int k = 0; k < 9; ++k) p.h[k] = H.ld(h + k);
  return p;
}

// one op of pass 3 (aba.hxx:207-232), forward
template <class MatT, class Stack>
DI void aba_step3(const KArgs& A, const Op op, const Pre3 pre, const AbaLayout L, Stack& S, const GlobalStore& H,
                  const MatT& q, const MatT& vin, const MatT& ddq, Aba3State& st) {
  JointConst jc;
  load_joint(A.tab, op.joint, jc);
  const int fl = op.flags;
  const bool root = fl & F_PARENT_ROOT;
  double Rp[9], tp[3], ovp[6], ap[6];
  if (root) {
#pragma unroll
    for (int k = 0; k < 9; ++k) Rp[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) tp[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) ovp[k] = ap[k] = 0.0;
    ap[0] = -A.gx; ap[1] = -A.gy; ap[2] = -A.gz;  // aba.hxx:261
  } else if (fl & F_FROM_REGS) {
#pragma unroll
    for (int k = 0; k < 9; ++k) Rp[k] = st.R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) tp[k] = st.t[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { ovp[k] = st.ov[k]; ap[k] = st.acc[k]; }
  } else {
    const int base = L.acc0 + op.bslot * 27;
#pragma unroll
    for (int k = 0; k < 9; ++k) Rp[k] = S.ld(base + k);
#pragma unroll
    for (int k = 0; k < 3; ++k) tp[k] = S.ld(base + 9 + k);
#pragma unroll
    for (int k = 0; k < 6; ++k) { ovp[k] = S.ld(base + 12 + k); ap[k] = S.ld(base + 18 + k); }
  }
  double sn = pre.h[7], cs = pre.h[8];
  double Rl[9], tl[3], vJ[6], ovJ[6], bias[6];
  joint_liMi(jc, op, q, pre.q0, true, sn, cs, Rl, tl);
  compose(root, Rp, tp, Rl, tl, st.R, st.t);
  joint_S_times(op, vin, pre.v0, vJ);
  se3_act_motion(st.R, st.t, vJ, ovJ);
  if (root) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { st.ov[k] = ovJ[k]; bias[k] = 0.0; }
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) st.ov[k] = ovJ[k] + ovp[k];
    motion_cross(ovp, st.ov, bias);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) st.acc[k] = ap[k] + bias[k];  // :223 (oa_gf[i] += oa_gf[parent])
  const int iv = op.idx_v, h = op.hand_off;
  const bool leaf = fl & F_LEAF;
  if (op.nvj == 1) {
    double dd;
    if (root) {
      dd = ddq.get(iv);  // written by the backward visit of this lane
    } else {
      dd = pre.h[6] - dot6(pre.h, st.acc);  // :224-226
      ddq.put(iv, dd);
    }
    if (!leaf) {
      double J[6];
      jcol_1dof(op, st.R, st.t, J);
#pragma unroll
      for (int k = 0; k < 6; ++k) st.acc[k] += J[k] * dd;  // :227
    }
  } else if (op.nvj == 6) {
    double dd[6], J[36];
    if (root) {
#pragma unroll
      for (int r = 0; r < 6; ++r) dd[r] = ddq.get(iv + r);
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double sum = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) sum += H.ld(h + 36 + 6 * r + c) * H.ld(h + 72 + c);
#pragma unroll
        for (int k = 0; k < 6; ++k) sum -= H.ld(h + 6 * r + k) * st.acc[k];
        dd[r] = sum;
        ddq.put(iv + r, sum);
      }
    }
    jcols_multi<6>(st.R, st.t, J);
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int k = 0; k < 6; ++k) st.acc[k] += J[6 * c + k] * dd[c];
  } else {
    double dd[3], J[18];
    if (root) {
#pragma unroll
      for (int r = 0; r < 3; ++r) dd[r] = ddq.get(iv + r);
    } else {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        double sum = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) sum += H.ld(h + 18 + 3 * r + c) * H.ld(h + 27 + c);
#pragma unroll
        for (int k = 0; k < 6; ++k) sum -= H.ld(h + 6 * r + k) * st.acc[k];
        dd[r] = sum;
        ddq.put(iv + r, sum);
      }
    }
    jcols_multi<3>(st.R, st.t, J);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k = 0; k < 6; ++k) st.acc[k] += J[6 * c + k] * dd[c];
  }
  if (fl & F_PUSH_BRANCH) {
    const int base = L.acc0 + op_own_bslot(fl) * 27;
#pragma unroll
    for (int k = 0; k < 9; ++k) S.st(base + k, st.R[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) S.st(base + 9 + k, st.t[k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) { S.st(base + 12 + k, st.ov[k]); S.st(base + 18 + k, st.acc[k]); }
  }
}

template <int LAYOUT>
__global__ __launch_bounds__(WAVE) void aba_kernel(const KArgs A) {
  const int lane = threadIdx.x;
  double* const wbase = A.ws + (int64_t)blockIdx.x * A.ws_stride;
  const AbaLayout L = aba_layout(A.maxdepth, A.maxbranch);
  const int gstack = L.end > A.lds_cap ? L.end - A.lds_cap : 0;  // stack slots that overflow LDS
  HybridStore S(brbd_smem, A.lds_cap, wbase, (unsigned)lane);
  const GlobalStore H(wbase + (int64_t)gstack * WAVE, (unsigned)lane);
  for (int64_t tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    const Mat<LAYOUT> q(const_cast<double*>(A.q), A.ldq, tile, A.B, lane);
    const Mat<LAYOUT> vin(const_cast<double*>(A.v), A.ldv, tile, A.B, lane);
    const Mat<LAYOUT> tauin(const_cast<double*>(A.a), A.lda, tile, A.B, lane);
    const Mat<LAYOUT> ddq(A.out, A.ldout, tile, A.B, lane);
    AbaState st;
    aba_state_init(st);
    {
      Op op = load_op(A.prog, 0);
      Pre pre = prefetch(op, q, vin, tauin);
      for (int s = 0; s < A.nops; ++s) {
        const Op nxt = load_op(A.prog, s + 1 < A.nops ? s + 1 : s);
        const Pre pn = prefetch(nxt, q, vin, tauin);
        aba_step12(A, op, pre, L, S, H, q, vin, tauin, ddq, st);
        op = nxt;
        pre = pn;
      }
    }
    Aba3State st3;
    aba3_state_init(st3);
    {
      Op op = load_op(A.prog3, 0);
      Pre3 pre = prefetch3(op, H, q, vin);
      for (int s = 0; s < A.nops3; ++s) {
        const Op nxt = load_op(A.prog3, s + 1 < A.nops3 ? s + 1 : s);
        const Pre3 pn = prefetch3(nxt, H, q, vin);
        aba_step3(A, op, pre, L, S, H, q, vin, ddq, st3);
        op = nxt;
        pre = pn;
      }
    }
  }
}

// =================================================================================== FK + Jacobians
// forwardKinematics(q) (kinematics.hxx:52-97) + computeJointJacobians(q) (jacobian.hxx:38-95).
// live state: KB[bslot][12] = oMi of branching joints.  Chains (arms) never spill.
struct FkState {
  double R[9], t[3];
};
template <class MatT, class Stack>
DI void fkjac_step(const KArgs& A, const Op op, Stack& S, const MatT& q, const MatT& Jout, const MatT& Mout,
                   bool wantJ, bool wantM, FkState& st) {
  JointConst jc;
  load_joint(A.tab, op.joint, jc);
  const int fl = op.flags;
  const bool root = fl & F_PARENT_ROOT;
  double Rl[9], tl[3], sn, cs;
  joint_liMi(jc, op, q, q.get(op.idx_q), false, sn, cs, Rl, tl);
  if (root) {
#pragma unroll
    for (int k = 0; k < 9; ++k) st.R[k] = Rl[k];
    st.t[0] = tl[0]; st.t[1] = tl[1]; st.t[2] = tl[2];
  } else {
    double Rp[9], tp[3], Rt[3];
    if (fl & F_FROM_REGS) {
#pragma unroll
      for (int k = 0; k < 9; ++k) Rp[k] = st.R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) tp[k] = st.t[k];
    } else {
      const int base = op.bslot * 12;
#pragma unroll
      for (int k = 0; k < 9; ++k) Rp[k] = S.ld(base + k);
#pragma unroll
      for (int k = 0; k < 3; ++k) tp[k] = S.ld(base + 9 + k);
    }
    rot_mul(Rp, Rl, st.R);
    rot_vec(Rp, tl, Rt);
    st.t[0] = tp[0] + Rt[0]; st.t[1] = tp[1] + Rt[1]; st.t[2] = tp[2] + Rt[2];
  }
  if (fl & F_PUSH_BRANCH) {
    const int base = op_own_bslot(fl) * 12;
#pragma unroll
    for (int k = 0; k < 9; ++k) S.st(base + k, st.R[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) S.st(base + 9 + k, st.t[k]);
  }
  if (wantM) {
    const int o = 12 * op.joint;
    if (fl & F_POST) {  // back from the canonical joint frame to the described one (brbd_flatten.hpp)
      double Ro[9];
      rot_mul(st.R, jc.post, Ro);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Mout.put(o + 3 * c + r, Ro[3 * r + c]);
    } else {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Mout.put(o + 3 * c + r, st.R[3 * r + c]);  // column-major rotation
    }
    Mout.put(o + 9, st.t[0]); Mout.put(o + 10, st.t[1]); Mout.put(o + 11, st.t[2]);
  }
  if (wantJ) {
    const int o = 6 * op.idx_v;
    if (op.nvj == 1) {
      double J[6];
      jcol_1dof(op, st.R, st.t, J);
#pragma unroll
      for (int k = 0; k < 6; ++k) Jout.put(o + k, J[k]);
    } else if (op.nvj == 6) {
      double J[36];
      jcols_multi<6>(st.R, st.t, J);
#pragma unroll
      for (int k = 0; k < 36; ++k) Jout.put(o + k, J[k]);
    } else {
      double J[18];
      jcols_multi<3>(st.R, st.t, J);
#pragma unroll
      for (int k = 0; k < 18; ++k) Jout.put(o + k, J[k]);
    }
  }
}

template <int LAYOUT>
__global__ __launch_bounds__(WAVE) void fkjac_kernel(const KArgs A) {
  const int lane = threadIdx.x;
  GlobalStore S(A.ws + (int64_t)blockIdx.x * A.ws_stride, (unsigned)lane);
  const bool wantJ = A.out != nullptr, wantM = A.out2 != nullptr;
  for (int64_t tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    const Mat<LAYOUT> q(const_cast<double*>(A.q), A.ldq, tile, A.B, lane);
    const Mat<LAYOUT> Jout(A.out, A.ldout, tile, A.B, lane);
    const Mat<LAYOUT> Mout(A.out2, A.ldout2, tile, A.B, lane);
    if (wantM) {  // joint 0 = identity
#pragma unroll
      for (int k = 0; k < 12; ++k) Mout.put(k, (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0);
    }
    FkState st;
#pragma unroll
    for (int k = 0; k < 9; ++k) st.R[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) st.t[k] = 0.0;
    for (int s = 0; s < A.nops3; ++s) fkjac_step(A, load_op(A.prog3, s), S, q, Jout, Mout, wantJ, wantM, st);
  }
}

// ============================================================================ joint Jacobian
// computeJointJacobians + getJointJacobian(joint, rf) (jacobian.hxx:38-95,175-283) for ONE joint:
// A.prog3 holds the joint's support chain, root first (A.nops3 entries); A.rf = 0 WORLD, 1 LOCAL,
// 2 LOCAL_WORLD_ALIGNED.  Two walks over the chain: the first for the joint's placement, the
// second writes each support column in the requested frame; the other columns are zero.
template <int NV, class MatT>
DI void jjac_put(int rf, const double* Rj, const double* tj, const double* J, int idx_v, const MatT& Jout) {
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const double* col = J + 6 * c;
    double o[6];
    if (rf == 1) {          // LOCAL: oMjoint.actInv(col)
      se3_actinv_motion(Rj, tj, col, o);
    } else if (rf == 2) {   // LOCAL_WORLD_ALIGNED: linear -= t x angular
      double tx[3];
      cross3(tj, col + 3, tx);
      o[0] = col[0] - tx[0]; o[1] = col[1] - tx[1]; o[2] = col[2] - tx[2];
      o[3] = col[3]; o[4] = col[4]; o[5] = col[5];
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) o[k] = col[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) Jout.put(6 * (idx_v + c) + k, o[k]);
  }
}

template <int LAYOUT>
__global__ __launch_bounds__(WAVE) void jjac_kernel(const KArgs A) {
  const int lane = threadIdx.x;
  for (int64_t tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    const Mat<LAYOUT> q(const_cast<double*>(A.q), A.ldq, tile, A.B, lane);
    const Mat<LAYOUT> Jout(A.out, A.ldout, tile, A.B, lane);
    // columns off the support chain are zero (A.chain: bit c = dof c is on the chain, for nv <= 64;
    // 0 = unknown: zero everything first, the chain's columns are then written twice)
    for (int c = 0; c < A.nv; ++c)
      if (A.chain == 0ull || !((A.chain >> c) & 1ull))
        for (int r = 0; r < 6; ++r) Jout.put(6 * c + r, 0.0);
    double Rj[9], tj[3];
    for (int pass = 0; pass < 2; ++pass) {
      double R[9], t[3];
      for (int s = 0; s < A.nops3; ++s) {
        const Op op = load_op(A.prog3, s);
        JointConst jc;
        load_joint(A.tab, op.joint, jc);
        double Rl[9], tl[3], sn, cs, Rn[9], tn[3];
        joint_liMi(jc, op, q, q.get(op.idx_q), false, sn, cs, Rl, tl);
        compose(s == 0, R, t, Rl, tl, Rn, tn);
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = Rn[k];
        t[0] = tn[0]; t[1] = tn[1]; t[2] = tn[2];
        if (pass == 0) {
          if (s == A.nops3 - 1) {  // the joint itself
            if (op.flags & F_POST) rot_mul(R, jc.post, Rj);  // described frame of a canonicalised joint
            else {
#pragma unroll
              for (int k = 0; k < 9; ++k) Rj[k] = R[k];
            }
            tj[0] = t[0]; tj[1] = t[1]; tj[2] = t[2];
          }
          continue;
        }
        if (op.nvj == 1) {
          double J[6];
          jcol_1dof(op, R, t, J);
          jjac_put<1>(A.rf, Rj, tj, J, op.idx_v, Jout);
        } else if (op.nvj == 6) {
          double J[36];
          jcols_multi<6>(R, t, J);
          jjac_put<6>(A.rf, Rj, tj, J, op.idx_v, Jout);
        } else {
          double J[18];
          jcols_multi<3>(R, t, J);
          jjac_put<3>(A.rf, Rj, tj, J, op.idx_v, Jout);
        }
      }
    }
  }
}

// =================================================================================== layout helpers
// dof x B instance-major (x[b*ld + d]) <-> batch-major (y[d*ld + b]) through a padded LDS tile.
// 64 x 64 tiles, 512-byte runs on both sides: brbd_transpose_to_*, and the transposes the library
// itself uses to run an INSTANCE_MAJOR call of an output-bound algorithm on the BATCH_MAJOR kernel
// (brbd_capi.hip, via_batch_major).  TO_BM: src is instance-major (B x dof, ld_src), dst
// batch-major (dof x B).  One workgroup owns 64 instances and walks over ALL their dofs: the
// instance-major rows are rarely aligned to cache lines, and the partial lines where two 512-byte
// runs meet are then written by the same CU within a few hundred cycles (one L2 merges them).
template <int TO_BM>
static __global__ __launch_bounds__(256) void xpose64_kernel(const double* __restrict__ src, int64_t ld_src,
                                                             double* __restrict__ dst, int64_t ld_dst, int64_t dof,
                                                             int64_t B) {
  __shared__ double tile[64][65];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;  // 4 waves: wave w moves rows w, w+4, ...
  const int64_t b0 = (int64_t)blockIdx.x * 64;
  // software pipeline: the loads of tile t+1 are in flight while tile t goes through LDS and out
  double r[16];
  auto load = [&](int64_t d0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = w + 4 * k;
      if (TO_BM) {
        const int64_t b = b0 + i, d = d0 + l;
        r[k] = (b < B && d < dof) ? src[b * ld_src + d] : 0.0;
      } else {
        const int64_t d = d0 + i, b = b0 + l;
        r[k] = (b < B && d < dof) ? src[d * ld_src + b] : 0.0;
      }
    }
  };
  load(0);
  for (int64_t d0 = 0; d0 < dof; d0 += 64) {
#pragma unroll
    for (int k = 0; k < 16; ++k) tile[w + 4 * k][l] = r[k];
    __syncthreads();
    if (d0 + 64 < dof) load(d0 + 64);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = w + 4 * k;
      if (TO_BM) {
        const int64_t d = d0 + i, b = b0 + l;
        if (b < B && d < dof) dst[d * ld_dst + b] = tile[l][i];
      } else {
        const int64_t b = b0 + i, d = d0 + l;
        if (b < B && d < dof) dst[b * ld_dst + d] = tile[l][i];
      }
    }
    __syncthreads();
  }
}

}  // namespace brbd

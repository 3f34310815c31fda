// brbd_f4.hpp -- the callers either side of the hot path for sampling-based MPC rollouts
// (SURVEY 8f rank 4), batched, one lane per instance, table-driven (any model):
//   integrate_kernel     integrate(model, q, v, qout)   algorithm/joint-configuration.hxx
//                        (liegroup/special-euclidean.hpp:773-812, special-orthogonal.hpp:164-184, 495-510)
//   minverse_kernel      computeMinverse(model, data, q)  algorithm/aba.hxx:841-877
//   aba_deriv_product_kernel   the closing products of computeABADerivatives
//                        (algorithm/aba-derivatives.hxx:449-453): ddq_dq = -Minv dtau_dq, ddq_dv = -Minv dtau_dv
// First correct versions: the 6 x nv column blocks of computeMinverse live in the wave-private
// coalesced global workspace like those of rnea_deriv_kernel (brbd_deriv.hpp); both matrix kernels
// are bound by that workspace traffic and by their nv x nv outputs.
#pragma once
#include "brbd_flatten.hpp"
#include "brbd_kernels.hpp"

namespace brbd {

struct F4Args {
  const JointConst* tab;
  const Op* prog3;  // forward depth-first program (ops carry the depth)
  const int* nvsub;
  int32_t nops3, njoints, nq, nv, maxdepth;
  int32_t full;  // minverse: also fill the strictly lower triangle (aba-derivatives.hxx:449-450)
  int64_t B, ntiles;
  double* ws;
  int64_t ws_stride;
  const double* q; int64_t ldq;
  const double* v; int64_t ldv;
  double* out; int64_t ldout;    // integrate: qout; minverse: Minv; product: ddq_dq (in: dtau_dq)
  double* out2; int64_t ldout2;  // product: ddq_dv (in: dtau_dv)
  const double* minv; int64_t ldminv;  // product: Minv, full symmetric
};

// ---------------------------------------------------------------------------------------- integrate
DI void f4_quat_exp3(const double* v, double* qo) {  // explog-quaternion.hpp:25-66, (x, y, z, w)
  const double eps = 2.220446049250313e-16;
  const double prec = 1.220703125e-4;  // eps^(1/4), math/taylor-expansion.hpp:31-36
  const double t2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const double t = sqrt(t2 + eps * eps);
  double s, c;
  sincos(0.5 * t, &s, &c);
  const double it = 1.0 / t;
  const double t2_2 = t2 * 0.25;
  const double a = 0.5 * (1.0 - t2_2 / 6.0 + t2_2 * t2_2 / 120.0);
  const bool big = t2 > prec;
#pragma unroll
  for (int k = 0; k < 3; ++k) qo[k] = big ? s * (v[k] * it) : a * v[k];
  qo[3] = big ? c : 1.0 - t2_2 * 0.5 + t2_2 * t2_2 / 24.0;
}
DI void f4_quat_mul(const double* a, const double* b, double* o) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
DI void f4_quat_normalize1(double* q) {  // math/quaternion.hpp:89-112 firstOrderNormalize
  const double alpha = (3.0 - (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])) * 0.5;
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] *= alpha;
}

template <int LAYOUT>
__global__ __launch_bounds__(WAVE) void integrate_kernel(const F4Args A) {
  const int lane = threadIdx.x;
  for (int64_t tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    const Mat<LAYOUT> q(const_cast<double*>(A.q), A.ldq, tile, A.B, lane);
    const Mat<LAYOUT> v(const_cast<double*>(A.v), A.ldv, tile, A.B, lane);
    const Mat<LAYOUT> o(A.out, A.ldout, tile, A.B, lane);
    for (int i = 1; i < A.njoints; ++i) {
      const JointConst BRBD_CONSTANT* t = as_constant(A.tab) + i;
      const int jt = t->jtype, iq = t->idx_q, iv = t->idx_v;
      if (jt == JT_FF) {  // special-euclidean.hpp:773-812
        double lin[3], w[3], quat[4];
#pragma unroll
        for (int k = 0; k < 3; ++k) { lin[k] = v.get(iv + k); w[k] = v.get(iv + 3 + k); }
#pragma unroll
        for (int k = 0; k < 4; ++k) quat[k] = q.get(iq + 3 + k);
        const double eps = 2.220446049250313e-16, prec = 1.220703125e-4;
        const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + eps * eps;  // exp6 :106-107
        const double th = sqrt(t2);
        double st, ct;
        sincos(th, &st, &ct);
        const double inv_t2 = 1.0 / t2;
        const double a_wxv = th < prec ? 0.5 - t2 / 24.0 : (1.0 - ct) * inv_t2;          // :119-123
        const double a_w2 = th < prec ? 1.0 / 6.0 - t2 / 120.0 : (th - st) * inv_t2 / th;  // :125-126
        double wxv[3], wwxv[3], trans[3], e3[4], R[9], rt[3], res[4];
        cross3(w, lin, wxv);
        cross3(w, wxv, wwxv);
#pragma unroll
        for (int k = 0; k < 3; ++k) trans[k] = lin[k] + a_wxv * wxv[k] + a_w2 * wwxv[k];  // :130
        f4_quat_exp3(w, e3);                                                              // :135
        quat_to_R(quat[0], quat[1], quat[2], quat[3], R);
        rot_vec(R, trans, rt);
#pragma unroll
        for (int k = 0; k < 3; ++k) o.put(iq + k, rt[k] + q.get(iq + k));  // :796
        f4_quat_mul(quat, e3, res);                                        // :799
        const double dotp = res[0] * quat[0] + res[1] * quat[1] + res[2] * quat[2] + res[3] * quat[3];
        const double sg = dotp < 0.0 ? -1.0 : 1.0;                         // :801-807
#pragma unroll
        for (int k = 0; k < 4; ++k) res[k] *= sg;
        f4_quat_normalize1(res);                                           // :812
#pragma unroll
        for (int k = 0; k < 4; ++k) o.put(iq + 3 + k, res[k]);
      } else if (jt == JT_SPH) {  // special-orthogonal.hpp:495-510
        double w[3], quat[4], e3[4], res[4];
#pragma unroll
        for (int k = 0; k < 3; ++k) w[k] = v.get(iv + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) quat[k] = q.get(iq + k);
        f4_quat_exp3(w, e3);
        f4_quat_mul(quat, e3, res);
        f4_quat_normalize1(res);
#pragma unroll
        for (int k = 0; k < 4; ++k) o.put(iq + k, res[k]);
      } else if (jt >= JT_RUBX && jt <= JT_RUBZ) {  // special-orthogonal.hpp:164-184 (q = cos, sin)
        const double ca = q.get(iq), sa = q.get(iq + 1);
        double so, co;
        sincos(v.get(iv), &so, &co);
        const double o0 = co * ca - so * sa, o1 = so * ca + co * sa;
        const double f = (3.0 - (o0 * o0 + o1 * o1)) * 0.5;
        o.put(iq, o0 * f);
        o.put(iq + 1, o1 * f);
      } else {  // vector space: q + v
        o.put(iq, q.get(iq) + v.get(iv));
      }
    }
  }
}

// ---------------------------------------------------------------------------------- computeMinverse
// workspace slots per lane
struct MinvLayout {
  int OMI, J, YA, U, UD, F0, FC, total;
  __host__ __device__ MinvLayout(int n, int nv, int maxdepth) {
    OMI = 0;                 // n x 12: oMi of every joint
    J = OMI + 12 * n;        // 6 x nv: world Jacobian columns (aba.hxx:650)
    YA = J + 6 * nv;         // n x 21: oYaba, packed symmetric
    U = YA + 21 * n;         // 6 x nv
    UD = U + 6 * nv;         // 6 x nv: UDinv
    F0 = UD + 6 * nv;        // 6 x nv: data.Fcrb[0]
    FC = F0 + 6 * nv;        // maxdepth x 6 x nv: data.Fcrb[i] of the joints on the current path
    total = FC + (maxdepth + 1) * 6 * nv;
  }
};
inline int minverse_ws_per_lane(const FlatModel& F) { return MinvLayout(F.njoints, F.nv, F.maxdepth).total; }

template <int NV, class MatT>
DI void minv_backward(const F4Args& A, const GlobalStore& W, const MinvLayout& L, const JointConst& jc, int i,
                      const MatT& M) {
  const int nv = A.nv, iv = jc.idx_v, p = jc.parent;
  const int nsub = as_constant(A.nvsub)[i];
  double Ia[21], Jc[NV][6], U[NV][6], UD[NV][6], D[NV * NV], Dinv[NV * NV];
#pragma unroll
  for (int k = 0; k < 21; ++k) Ia[k] = W.ld(L.YA + 21 * i + k);
#pragma unroll
  for (int c = 0; c < NV; ++c) {
#pragma unroll
    for (int k = 0; k < 6; ++k) Jc[c][k] = W.ld(L.J + 6 * (iv + c) + k);
    sym6_mulvec(Ia, Jc[c], U[c]);  // :689
  }
#pragma unroll
  for (int r = 0; r < NV; ++r)
#pragma unroll
    for (int c = 0; c < NV; ++c) D[NV * r + c] = dot6(Jc[r], U[c]) + (r == c ? jc.arm[r] : 0.0);  // :690-693
  if constexpr (NV == 1) Dinv[0] = 1.0 / D[0];
  else spd_inverse<NV>(D, Dinv);  // :695
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < NV; ++r) s += U[r][k] * Dinv[NV * r + c];
      UD[c][k] = s;  // :696
      W.st(L.U + 6 * (iv + c) + k, U[c][k]);
      W.st(L.UD + 6 * (iv + c) + k, s);
    }
#pragma unroll
  for (int r = 0; r < NV; ++r)
#pragma unroll
    for (int c = 0; c < NV; ++c) M.put((iv + c) * nv + iv + r, Dinv[NV * r + c]);  // :698
  const int nch = nsub - NV;
  if (nch > 0) {
    double SD[NV][6];
#pragma unroll
    for (int c = 0; c < NV; ++c)
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < NV; ++r) s += Jc[r][k] * Dinv[NV * r + c];
        SD[c][k] = s;  // :703
      }
    for (int c = NV; c < nsub; ++c) {  // :705-707
      double f[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) f[k] = W.ld(L.F0 + 6 * (iv + c) + k);
#pragma unroll
      for (int r = 0; r < NV; ++r) M.put((iv + c) * nv + iv + r, -dot6(SD[r], f));
    }
  }
  if (nch <= 0 || p > 0) {  // :709-721: leaf: Fcrb = U * Minv block; else (moving parent): Fcrb += U * Minv block
    for (int c = 0; c < nsub; ++c) {
      double mrc[NV];
#pragma unroll
      for (int r = 0; r < NV; ++r) mrc[r] = M.get((iv + c) * nv + iv + r);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        double s = nch > 0 ? W.ld(L.F0 + 6 * (iv + c) + k) : 0.0;
#pragma unroll
        for (int r = 0; r < NV; ++r) s += U[r][k] * mrc[r];
        W.st(L.F0 + 6 * (iv + c) + k, s);
      }
    }
  }
  if (p > 0) {  // :723-727
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 6; ++b) {
        double s = Ia[sidx(a, b)];
#pragma unroll
        for (int c = 0; c < NV; ++c) s -= UD[c][a] * U[c][b];
        const int slot = L.YA + 21 * p + sidx(a, b);
        W.st(slot, W.ld(slot) + s);
      }
  }
}

template <int NV, class MatT>
DI void minv_forward2(const F4Args& A, const GlobalStore& W, const MinvLayout& L, const JointConst& jc, int depth,
                      const MatT& M) {
  const int nv = A.nv, iv = jc.idx_v, p = jc.parent;
  const int Fi = L.FC + depth * 6 * nv, Fp = L.FC + (depth - 1) * 6 * nv;  // (depth >= 1 when p > 0)
  double Jc[NV][6], UD[NV][6];
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      Jc[c][k] = W.ld(L.J + 6 * (iv + c) + k);
      UD[c][k] = W.ld(L.UD + 6 * (iv + c) + k);
    }
  for (int c = iv; c < nv; ++c) {
    double fp[6], mrc[NV];
#pragma unroll
    for (int k = 0; k < 6; ++k) fp[k] = p > 0 ? W.ld(Fp + 6 * c + k) : 0.0;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      double x = M.get(c * nv + iv + r);
      if (p > 0) {  // :819-825
        x -= dot6(UD[r], fp);
        M.put(c * nv + iv + r, x);
      }
      mrc[r] = x;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {  // :827-832
      double s = fp[k];
#pragma unroll
      for (int r = 0; r < NV; ++r) s += Jc[r][k] * mrc[r];
      W.st(Fi + 6 * c + k, s);
    }
  }
}

template <int LAYOUT>
__global__ __launch_bounds__(WAVE) void minverse_kernel(const F4Args A) {
  const int lane = threadIdx.x;
  const GlobalStore W(A.ws + (int64_t)blockIdx.x * A.ws_stride, (unsigned)lane);
  const int nv = A.nv, n = A.njoints;
  const MinvLayout L(n, nv, A.maxdepth);
  for (int64_t tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    const Mat<LAYOUT> q(const_cast<double*>(A.q), A.ldq, tile, A.B, lane);
    const Mat<LAYOUT> M(A.out, A.ldout, tile, A.B, lane);
    for (int e = 0; e < nv * nv; ++e) M.put(e, 0.0);                 // :853 (data.Minv starts as zeros)
    for (int e = 0; e < 6 * nv; ++e) W.st(L.F0 + e, 0.0);            // :863
    // ------------------------------------------------------------------ pass 1 (:626-654)
    for (int i = 1; i < n; ++i) {
      JointConst jc;
      load_joint(A.tab, i, jc);
      const Op op = op_from_joint(jc, i);
      const int p = jc.parent, iv = jc.idx_v;
      double Rl[9], tl[3], sn, cs, R[9], t[3];
      joint_liMi(jc, op, q, q.get(op.idx_q), false, sn, cs, Rl, tl);
      if (p > 0) {
        double Rp[9], tp[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rp[k] = W.ld(L.OMI + 12 * p + k);
#pragma unroll
        for (int k = 0; k < 3; ++k) tp[k] = W.ld(L.OMI + 12 * p + 9 + k);
        compose(false, Rp, tp, Rl, tl, R, t);
      } else {
        compose(true, nullptr, nullptr, Rl, tl, R, t);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) W.st(L.OMI + 12 * i + k, R[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) W.st(L.OMI + 12 * i + 9 + k, t[k]);
      if (jc.nvj == 1) {
        double J[6];
        jcol_1dof(op, R, t, J);
#pragma unroll
        for (int k = 0; k < 6; ++k) W.st(L.J + 6 * iv + k, J[k]);
      } else if (jc.nvj == 6) {
        double J[36];
        jcols_multi<6>(R, t, J);
#pragma unroll
        for (int k = 0; k < 36; ++k) W.st(L.J + 6 * iv + k, J[k]);
      } else {
        double J[18];
        jcols_multi<3>(R, t, J);
#pragma unroll
        for (int k = 0; k < 18; ++k) W.st(L.J + 6 * iv + k, J[k]);
      }
      double cw[3], Iw[6], Ya[21];
      inertia_to_world(R, t, jc.m, jc.c, jc.I, cw, Iw);  // :652
      inertia_to_sym6(jc.m, cw, Iw, Ya);                 // :653
#pragma unroll
      for (int k = 0; k < 21; ++k) W.st(L.YA + 21 * i + k, Ya[k]);
    }
    // ------------------------------------------------------------------ pass 2 (:668-728)
    for (int i = n - 1; i > 0; --i) {
      JointConst jc;
      load_joint(A.tab, i, jc);
      if (jc.nvj == 1) minv_backward<1>(A, W, L, jc, i, M);
      else if (jc.nvj == 6) minv_backward<6>(A, W, L, jc, i, M);
      else minv_backward<3>(A, W, L, jc, i, M);
    }
    // ------------------------------------------------------------------ pass 3 (:802-833), depth-first
    for (int s = 0; s < A.nops3; ++s) {
      const Op op = load_op(A.prog3, s);
      JointConst jc;
      load_joint(A.tab, op.joint, jc);
      const int depth = op.depth + 1;  // level 0 is never read (universe)
      if (jc.nvj == 1) minv_forward2<1>(A, W, L, jc, depth, M);
      else if (jc.nvj == 6) minv_forward2<6>(A, W, L, jc, depth, M);
      else minv_forward2<3>(A, W, L, jc, depth, M);
    }
    if (A.full)  // aba-derivatives.hxx:449-450: strictly lower = transpose of strictly upper
      for (int c = 0; c < nv; ++c)
        for (int r = c + 1; r < nv; ++r) M.put(c * nv + r, M.get(r * nv + c));
  }
}

// ---------------------------------------------------------- closing products of computeABADerivatives
// In place: out (holding dtau_dq) <- -Minv * out, out2 (dtau_dv) <- -Minv * out2; Minv full symmetric.
// Column c of a product depends on column c of the factor only: CB columns of both factors are held
// in registers (2 * CB * NVMAX doubles) while Minv streams past once per column block -- Minv is the
// operand that is re-read (nv / CB times; from L2 / HBM: 64 instances' matrices do not fit a cache).
template <int LAYOUT, int NVMAX, int CB>
__global__ __launch_bounds__(WAVE) void aba_deriv_product_kernel(const F4Args A) {
  const int lane = threadIdx.x;
  const int nv = A.nv;
  for (int64_t tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    const Mat<LAYOUT> Mi(const_cast<double*>(A.minv), A.ldminv, tile, A.B, lane);
    const Mat<LAYOUT> D1(A.out, A.ldout, tile, A.B, lane);
    const Mat<LAYOUT> D2(A.out2, A.ldout2, tile, A.B, lane);
    for (int c0 = 0; c0 < nv; c0 += CB) {
      double c1[CB][NVMAX], c2[CB][NVMAX];
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const bool col = c0 + j < nv;
#pragma unroll
        for (int k = 0; k < NVMAX; ++k) {
          c1[j][k] = (col && k < nv) ? D1.get((c0 + j) * nv + k) : 0.0;
          c2[j][k] = (col && k < nv) ? D2.get((c0 + j) * nv + k) : 0.0;
        }
      }
      for (int r = 0; r < nv; ++r) {
        double s1[CB], s2[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) s1[j] = s2[j] = 0.0;
#pragma unroll
        for (int k = 0; k < NVMAX; ++k) {
          const double mrk = k < nv ? Mi.get(k * nv + r) : 0.0;  // Minv(r, k) = Minv(k, r)
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            s1[j] += mrk * c1[j][k];
            s2[j] += mrk * c2[j][k];
          }
        }
#pragma unroll
        for (int j = 0; j < CB; ++j)
          if (c0 + j < nv) {
            D1.put((c0 + j) * nv + r, -s1[j]);
            D2.put((c0 + j) * nv + r, -s2[j]);
          }
      }
    }
  }
}

}  // namespace brbd

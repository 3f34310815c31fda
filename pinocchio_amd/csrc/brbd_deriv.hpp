// brbd_deriv.hpp -- batched computeRNEADerivatives (algorithm/rnea-derivatives.hxx:472-541).
//
// One lane = one instance.  The per-dof 6 x nv column blocks (J, dVdq, dAdq, dAdv, dFda, dFdq,
// dFdv, dYtJ) and the per-joint world quantities do not fit on chip, so this kernel keeps them in
// the wave-private, coalesced global workspace and follows the reference's two sweeps
// (forward :263-338, backward :378-459).  It is bound by its nv x nv outputs and the workspace.
#pragma once
#include "brbd_flatten.hpp"
#include "brbd_kernels.hpp"

namespace brbd {

struct DerivArgs {
  const JointConst* tab;
  const int* nvsub;
  int32_t njoints, nq, nv;
  int64_t B, ntiles;
  double* ws;
  int64_t ws_stride;
  double gx, gy, gz;
  const double* q; int64_t ldq;
  const double* v; int64_t ldv;
  const double* a; int64_t lda;
  double* dq; int64_t lddq;
  double* dv; int64_t lddv;
  double* da; int64_t ldda;    // may be NULL
  double* tau; int64_t ldtau;  // may be NULL
};

// per-joint record: oMi 12, v 6, a 6, ov 6, oa_gf 6, of 6, Ycrb 10, doYcrb 36
constexpr int DJ_OMI = 0, DJ_V = 12, DJ_A = 18, DJ_OV = 24, DJ_OAGF = 30, DJ_OF = 36, DJ_Y = 42,
              DJ_DY = 52, DJ_SIZE = 88;
// column blocks (6 x nv each)
enum { DC_J = 0, DC_DVDQ, DC_DADQ, DC_DADV, DC_DFDA, DC_DFDQ, DC_DFDV, DC_DYTJ, DC_COUNT };

inline int deriv_ws_per_lane(const FlatModel& F) { return DC_COUNT * 6 * F.nv + DJ_SIZE * F.njoints; }

DI void ld6(const GlobalStore& w, int slot, double* o) {
#pragma unroll
  for (int k = 0; k < 6; ++k) o[k] = w.ld(slot + k);
}
DI void st6(const GlobalStore& w, int slot, const double* x) {
#pragma unroll
  for (int k = 0; k < 6; ++k) w.st(slot + k, x[k]);
}
DI void skew3(const double* v, double* S) {
  S[0] = 0; S[1] = -v[2]; S[2] = v[1];
  S[3] = v[2]; S[4] = 0; S[5] = -v[0];
  S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
DI void sym3_to_mat(const double* I, double* M) {
  M[0] = I[0]; M[1] = I[1]; M[2] = I[3];
  M[3] = I[1]; M[4] = I[2]; M[5] = I[4];
  M[6] = I[3]; M[7] = I[4]; M[8] = I[5];
}
// Inertia::variation(v) + addForceCrossMatrix(oh) (inertia.hpp:762-789, rnea-derivatives.hxx:335-351)
// -> full row-major 6x6
DI void inertia_variation(double m, const double* c, const double* I, const double* v, const double* oh,
                          double* res) {
  double mv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) mv[k] = v[k] * m;
  double s_mvl[9], s_mva[9], s_c[9], s_w[9], A[9], Bm[9], LA[9], AA[9], LL[9], T1[9], T2[9], I3[9];
  skew3(mv, s_mvl);
  skew3(mv + 3, s_mva);
  skew3(c, s_c);
  skew3(v + 3, s_w);
  rot_mul(s_mva, s_c, A);
  rot_mul(s_c, s_mva, Bm);
#pragma unroll
  for (int k = 0; k < 9; ++k) LA[k] = -s_mvl[k] - A[k] + Bm[k];
  rot_mul(s_mvl, s_c, A);
  rot_mul(s_c, s_mvl, Bm);
#pragma unroll
  for (int k = 0; k < 9; ++k) AA[k] = -A[k] - Bm[k];
  sym3_to_mat(I, I3);
  rot_mul(s_c, s_c, A);
#pragma unroll
  for (int k = 0; k < 9; ++k) LL[k] = I3[k] - m * A[k];
  rot_mul(LL, s_w, T1);
  rot_mul(s_w, LL, T2);
#pragma unroll
  for (int k = 0; k < 9; ++k) AA[k] += -T1[k] + T2[k];
  double sl[9], sa[9], nl[3] = {-oh[0], -oh[1], -oh[2]}, na[3] = {-oh[3], -oh[4], -oh[5]};
  skew3(nl, sl);
  skew3(na, sa);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      res[6 * i + j] = 0.0;
      res[6 * i + 3 + j] = LA[3 * i + j] + sl[3 * i + j];
      res[6 * (3 + i) + j] = LA[3 * j + i] + sl[3 * i + j];
      res[6 * (3 + i) + 3 + j] = AA[3 * i + j] + sa[3 * i + j];
    }
}
DI void mat6_mulvec(const double* M, const double* x, double* o) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) s += M[6 * i + j] * x[j];
    o[i] = s;
  }
}

template <int LAYOUT>
__global__ __launch_bounds__(WAVE) void rnea_deriv_kernel(const DerivArgs A) {
  const int lane = threadIdx.x;
  const GlobalStore W(A.ws + (int64_t)blockIdx.x * A.ws_stride, (unsigned)lane);
  const int nv = A.nv, n = A.njoints;
  const int colsz = 6 * nv;
  const int J0 = 0;  // column block b starts at b * colsz
  const int JR0 = DC_COUNT * colsz;
  for (int64_t tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    const bool valid = true;  // lanes past the batch are clamped onto the last instance (Mat)
    const Mat<LAYOUT> q(const_cast<double*>(A.q), A.ldq, tile, A.B, lane);
    const Mat<LAYOUT> vin(const_cast<double*>(A.v), A.ldv, tile, A.B, lane);
    const Mat<LAYOUT> ain(const_cast<double*>(A.a), A.lda, tile, A.B, lane);
    const Mat<LAYOUT> pdq(A.dq, A.lddq, tile, A.B, lane);
    const Mat<LAYOUT> pdv(A.dv, A.lddv, tile, A.B, lane);
    const Mat<LAYOUT> pda(A.da, A.ldda, tile, A.B, lane);
    const Mat<LAYOUT> ptau(A.tau, A.ldtau, tile, A.B, lane);
    const bool wantM = A.da != nullptr, wantTau = A.tau != nullptr;
    // structural zeros (the reference asks the caller to pre-zero, rnea-derivatives.hpp:104-106)
    if (valid)
      for (int e = 0; e < nv * nv; ++e) {
        pdq.put(e, 0.0);
        pdv.put(e, 0.0);
        if (wantM) pda.put(e, 0.0);
      }

    // ------------------------------------------------------------ forward sweep (:263-338)
    for (int i = 1; i < n; ++i) {
      JointConst jc;
      load_joint(A.tab, i, jc);
      const int p = jc.parent, iv = jc.idx_v, nvj = jc.nvj;
      const int rec = JR0 + DJ_SIZE * i, prec = JR0 + DJ_SIZE * p;
      double Rl[9], tl[3], sn, cs, R[9], t[3];
      const Op op = op_from_joint(jc, i);
      joint_liMi(jc, op, q, q.get(op.idx_q), false, sn, cs, Rl, tl);
      double vJ[6], Sa[6], vloc[6], aloc[6], tmp[6];
      joint_S_times(op, vin, vin.get(op.idx_v), vJ);
      joint_S_times(op, ain, ain.get(op.idx_v), Sa);
      double ovp[6], oagfp[6];
      if (p > 0) {
        double Rp[9], tp[3], Rt[3], vp[6], ap[6];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rp[k] = W.ld(prec + DJ_OMI + k);
#pragma unroll
        for (int k = 0; k < 3; ++k) tp[k] = W.ld(prec + DJ_OMI + 9 + k);
        rot_mul(Rp, Rl, R);
        rot_vec(Rp, tl, Rt);
        t[0] = tp[0] + Rt[0]; t[1] = tp[1] + Rt[1]; t[2] = tp[2] + Rt[2];
        ld6(W, prec + DJ_V, vp);
        ld6(W, prec + DJ_A, ap);
        se3_actinv_motion(Rl, tl, vp, tmp);
#pragma unroll
        for (int k = 0; k < 6; ++k) vloc[k] = vJ[k] + tmp[k];
        motion_cross(vloc, vJ, aloc);
        se3_actinv_motion(Rl, tl, ap, tmp);
#pragma unroll
        for (int k = 0; k < 6; ++k) aloc[k] = (Sa[k] + aloc[k]) + tmp[k];
        ld6(W, prec + DJ_OV, ovp);
        ld6(W, prec + DJ_OAGF, oagfp);
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = Rl[k];
        t[0] = tl[0]; t[1] = tl[1]; t[2] = tl[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) vloc[k] = vJ[k];
        motion_cross(vloc, vJ, aloc);
#pragma unroll
        for (int k = 0; k < 6; ++k) { aloc[k] = Sa[k] + aloc[k]; ovp[k] = 0.0; oagfp[k] = 0.0; }
        oagfp[0] = -A.gx; oagfp[1] = -A.gy; oagfp[2] = -A.gz;  // :504
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) W.st(rec + DJ_OMI + k, R[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) W.st(rec + DJ_OMI + 9 + k, t[k]);
      st6(W, rec + DJ_V, vloc);
      st6(W, rec + DJ_A, aloc);
      double cw[3], Iw[6], ov[6], oa[6], oh[6], of[6];
      inertia_to_world(R, t, jc.m, jc.c, jc.I, cw, Iw);  // :302
      se3_act_motion(R, t, vloc, ov);                    // :303
      se3_act_motion(R, t, aloc, oa);                    // :304
      oa[0] -= A.gx; oa[1] -= A.gy; oa[2] -= A.gz;       // :305 (oa_gf)
      inertia_mul(jc.m, cw, Iw, ov, oh);                 // :307
      inertia_mul(jc.m, cw, Iw, oa, of);                 // :308
      force_cross(ov, oh, tmp);
#pragma unroll
      for (int k = 0; k < 6; ++k) of[k] += tmp[k];
      st6(W, rec + DJ_OV, ov);
      st6(W, rec + DJ_OAGF, oa);
      st6(W, rec + DJ_OF, of);
      W.st(rec + DJ_Y, jc.m);
#pragma unroll
      for (int k = 0; k < 3; ++k) W.st(rec + DJ_Y + 1 + k, cw[k]);
#pragma unroll
      for (int k = 0; k < 6; ++k) W.st(rec + DJ_Y + 4 + k, Iw[k]);
      // columns (:319-332)
      for (int c = 0; c < nvj; ++c) {
        double Sc[6], Jc[6], dJ[6], dVdq[6], dAdq[6], dAdv[6];
        if (nvj == 1) {
#pragma unroll
          for (int k = 0; k < 6; ++k) Sc[k] = (k == jc.comp) ? 1.0 : 0.0;
        } else {
          const int comp = nvj == 6 ? c : 3 + c;
#pragma unroll
          for (int k = 0; k < 6; ++k) Sc[k] = (k == comp) ? 1.0 : 0.0;
        }
        se3_act_motion(R, t, Sc, Jc);
        motion_cross(ov, Jc, dJ);
        motion_cross(oagfp, Jc, dAdq);
        if (p > 0) {
          motion_cross(ovp, Jc, dVdq);
          motion_cross(ovp, dVdq, tmp);
#pragma unroll
          for (int k = 0; k < 6; ++k) { dAdq[k] += tmp[k]; dAdv[k] = dJ[k] + dVdq[k]; }
        } else {
#pragma unroll
          for (int k = 0; k < 6; ++k) { dVdq[k] = 0.0; dAdv[k] = dJ[k]; }
        }
        const int col = 6 * (iv + c);
        st6(W, J0 + DC_J * colsz + col, Jc);
        st6(W, J0 + DC_DVDQ * colsz + col, dVdq);
        st6(W, J0 + DC_DADQ * colsz + col, dAdq);
        st6(W, J0 + DC_DADV * colsz + col, dAdv);
      }
      double dY[36];
      inertia_variation(jc.m, cw, Iw, ov, oh, dY);  // :335-337
#pragma unroll
      for (int k = 0; k < 36; ++k) W.st(rec + DJ_DY + k, dY[k]);
    }

    // ------------------------------------------------------------ backward sweep (:378-459)
    for (int i = n - 1; i > 0; --i) {
      JointConst jc;
      load_joint(A.tab, i, jc);
      const int p = jc.parent, iv = jc.idx_v, nvj = jc.nvj;
      const int nvs = as_constant(A.nvsub)[i];
      const int ivp = iv + nvj, nvsp = nvs - nvj;
      const int rec = JR0 + DJ_SIZE * i, prec = JR0 + DJ_SIZE * p;
      double of[6], Ym, Yc[3], YI[6], dY[36];
      ld6(W, rec + DJ_OF, of);
      Ym = W.ld(rec + DJ_Y);
#pragma unroll
      for (int k = 0; k < 3; ++k) Yc[k] = W.ld(rec + DJ_Y + 1 + k);
#pragma unroll
      for (int k = 0; k < 6; ++k) YI[k] = W.ld(rec + DJ_Y + 4 + k);
#pragma unroll
      for (int k = 0; k < 36; ++k) dY[k] = W.ld(rec + DJ_DY + k);
      // per own column: the 6-vectors that only depend on this joint
      for (int c = 0; c < nvj; ++c) {
        const int col = 6 * (iv + c);
        double Jc[6], dVdq[6], dAdq[6], dAdv[6], dFda[6], dFdq[6], dFdv[6], dYtJ[6], tmp[6];
        ld6(W, DC_J * colsz + col, Jc);
        ld6(W, DC_DVDQ * colsz + col, dVdq);
        ld6(W, DC_DADQ * colsz + col, dAdq);
        ld6(W, DC_DADV * colsz + col, dAdv);
        if (wantTau) {  // :409-410 (+ armature, :538-539)
          const double x = dot6(Jc, of) + as_constant(A.tab)[i].arm[c] * ain.get(iv + c);
          if (valid) ptau.put(iv + c, x);
        }
        inertia_mul(Ym, Yc, YI, Jc, dFda);  // :419
        inertia_mul(Ym, Yc, YI, dAdq, dFdq);  // :424-430
        if (p > 0) {
          mat6_mulvec(dY, dVdq, tmp);
#pragma unroll
          for (int k = 0; k < 6; ++k) dFdq[k] += tmp[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {  // :432 dYtJ_col = doYcrb^T J_col
          double s = 0.0;
#pragma unroll
          for (int r = 0; r < 6; ++r) s += Jc[r] * dY[6 * r + k];
          dYtJ[k] = s;
        }
        mat6_mulvec(dY, Jc, dFdv);  // :443-444
        inertia_mul(Ym, Yc, YI, dAdv, tmp);
#pragma unroll
        for (int k = 0; k < 6; ++k) dFdv[k] += tmp[k];
        st6(W, DC_DFDA * colsz + col, dFda);
        st6(W, DC_DFDQ * colsz + col, dFdq);
        st6(W, DC_DFDV * colsz + col, dFdv);
        st6(W, DC_DYTJ * colsz + col, dYtJ);
      }
      // row blocks: rows = own dofs, columns = whole subtree (:420-421, :437-438, :450-451)
      for (int r = 0; r < nvj; ++r) {
        double Jr[6];
        ld6(W, DC_J * colsz + 6 * (iv + r), Jr);
        for (int c = 0; c < nvs; ++c) {
          double x[6];
          const int col = 6 * (iv + c);
          const int e = (iv + c) * nv + iv + r;
          if (wantM) {
            ld6(W, DC_DFDA * colsz + col, x);
            double mval = dot6(Jr, x);
            if (c == r) mval += as_constant(A.tab)[i].arm[r];  // :540
            if (valid) pda.put(e, mval);
          }
          ld6(W, DC_DFDQ * colsz + col, x);
          const double dqv = dot6(Jr, x);
          ld6(W, DC_DFDV * colsz + col, x);
          const double dvv = dot6(Jr, x);
          if (valid) { pdq.put(e, dqv); pdv.put(e, dvv); }
        }
      }
      // column blocks: rows = strict descendants, columns = own dofs (:433-435, :446-448)
      for (int c = 0; c < nvj; ++c) {
        double Jc[6], dVdq[6], dAdq[6], dAdv[6];
        const int col = 6 * (iv + c);
        ld6(W, DC_J * colsz + col, Jc);
        ld6(W, DC_DVDQ * colsz + col, dVdq);
        ld6(W, DC_DADQ * colsz + col, dAdq);
        ld6(W, DC_DADV * colsz + col, dAdv);
        for (int r = 0; r < nvsp; ++r) {
          double fa[6], yj[6];
          ld6(W, DC_DFDA * colsz + 6 * (ivp + r), fa);
          ld6(W, DC_DYTJ * colsz + 6 * (ivp + r), yj);
          const int e = (iv + c) * nv + ivp + r;
          const double dqv = dot6(fa, dAdq) + dot6(yj, dVdq);
          const double dvv = dot6(fa, dAdv) + dot6(yj, Jc);
          if (valid) { pdq.put(e, dqv); pdv.put(e, dvv); }
        }
        // :440 dFdq_cols += J_cols x* of   (after the dq row block used dFdq)
        double dFdq[6], tmp[6];
        ld6(W, DC_DFDQ * colsz + col, dFdq);
        force_cross(Jc, of, tmp);
#pragma unroll
        for (int k = 0; k < 6; ++k) dFdq[k] += tmp[k];
        st6(W, DC_DFDQ * colsz + col, dFdq);
      }
      if (p > 0) {  // :453-458
        // Ycrb[p] += Ycrb[i] (inertia.hpp:672-686)
        double Pm = W.ld(prec + DJ_Y), Pc[3], PI[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) Pc[k] = W.ld(prec + DJ_Y + 1 + k);
#pragma unroll
        for (int k = 0; k < 6; ++k) PI[k] = W.ld(prec + DJ_Y + 4 + k);
        const double eps = 2.220446049250313e-16;
        const double mab = Pm + Ym;
        const double mab_inv = 1.0 / (mab > eps ? mab : eps);
        double AB[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) AB[k] = Pc[k] - Yc[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) Pc[k] = Pc[k] * (Pm * mab_inv) + (Ym * mab_inv) * Yc[k];
        const double coef = Pm * Ym * mab_inv;
        // [AB]x[AB]x = AB AB^T - |AB|^2 1
        const double n2 = AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2];
        PI[0] = (PI[0] + YI[0]) - coef * (AB[0] * AB[0] - n2);
        PI[1] = (PI[1] + YI[1]) - coef * (AB[0] * AB[1]);
        PI[2] = (PI[2] + YI[2]) - coef * (AB[1] * AB[1] - n2);
        PI[3] = (PI[3] + YI[3]) - coef * (AB[0] * AB[2]);
        PI[4] = (PI[4] + YI[4]) - coef * (AB[1] * AB[2]);
        PI[5] = (PI[5] + YI[5]) - coef * (AB[2] * AB[2] - n2);
        W.st(prec + DJ_Y, mab);
#pragma unroll
        for (int k = 0; k < 3; ++k) W.st(prec + DJ_Y + 1 + k, Pc[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) W.st(prec + DJ_Y + 4 + k, PI[k]);
#pragma unroll
        for (int k = 0; k < 36; ++k) W.st(prec + DJ_DY + k, W.ld(prec + DJ_DY + k) + dY[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) W.st(prec + DJ_OF + k, W.ld(prec + DJ_OF + k) + of[k]);
      }
    }
  }
}

}  // namespace brbd

/*
 * brbd_oracle.c -- scalar C restatement of the reference's hot path.
 * TEST INFRASTRUCTURE ONLY (see brbd_oracle.h).  RNEA pinned by value against the reference's
 * unittest/value.cpp vectors (tests/test_reference_values.py); the rest by invariants.
 *
 * Every function cites the reference file:line whose arithmetic it follows.
 * Citations are relative to /root/reference/include/pinocchio/.
 *
 * Conventions: 6-vectors are [lin(3); ang(3)]; SE3 = (R row-major[9], t[3]);
 * Inertia = (m, c[3], I[6]) with I in Symmetric3 order xx,xy,yy,xz,yz,zz
 * (spatial/symmetric3.hpp:43-52).
 */
#define _GNU_SOURCE
#include "brbd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  double R[9];
  double t[3];
} se3_t;

typedef struct {
  double m;
  double c[3];
  double I[6];
} inertia_t;

/* ------------------------------------------------------------------ 3-vectors */
static inline void cross3(const double* a, const double* b, double* o) {
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  o[0] = x;
  o[1] = y;
  o[2] = z;
}
static inline void matvec3(const double* R, const double* x, double* o) {
  double a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  double b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  double c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  o[0] = a;
  o[1] = b;
  o[2] = c;
}
static inline void matTvec3(const double* R, const double* x, double* o) {
  double a = R[0] * x[0] + R[3] * x[1] + R[6] * x[2];
  double b = R[1] * x[0] + R[4] * x[1] + R[7] * x[2];
  double c = R[2] * x[0] + R[5] * x[1] + R[8] * x[2];
  o[0] = a;
  o[1] = b;
  o[2] = c;
}
static inline void matmul3(const double* A, const double* B, double* C) {
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, T, sizeof T);
}
static inline void skew3(const double* v, double* S) {
  S[0] = 0;
  S[1] = -v[2];
  S[2] = v[1];
  S[3] = v[2];
  S[4] = 0;
  S[5] = -v[0];
  S[6] = -v[1];
  S[7] = v[0];
  S[8] = 0;
}
static inline void sym3_to_mat(const double* I, double* M) {
  M[0] = I[0];
  M[1] = I[1];
  M[2] = I[3];
  M[3] = I[1];
  M[4] = I[2];
  M[5] = I[4];
  M[6] = I[3];
  M[7] = I[4];
  M[8] = I[5];
}
static inline void sym3_mulvec(const double* I, const double* v, double* o) {
  /* spatial/symmetric3.hpp:490-503 rhsMult */
  o[0] = I[0] * v[0] + I[1] * v[1] + I[3] * v[2];
  o[1] = I[1] * v[0] + I[2] * v[1] + I[4] * v[2];
  o[2] = I[3] * v[0] + I[4] * v[1] + I[5] * v[2];
}

/* ------------------------------------------------------------------ SE3 */
/* spatial/se3-tpl.hpp:311-315  A*B = (RA RB, tA + RA tB) */
static inline void se3_mul(const se3_t* A, const se3_t* B, se3_t* C) {
  se3_t T;
  matmul3(A->R, B->R, T.R);
  matvec3(A->R, B->t, T.t);
  for (int k = 0; k < 3; ++k) T.t[k] += A->t[k];
  *C = T;
}
/* spatial/motion-dense.hpp:261-266 */
static inline void se3_act_motion(const se3_t* M, const double* m, double* o) {
  double ang[3], lin[3], txa[3];
  matvec3(M->R, m + 3, ang);
  matvec3(M->R, m, lin);
  cross3(M->t, ang, txa);
  for (int k = 0; k < 3; ++k) {
    o[k] = lin[k] + txa[k];
    o[3 + k] = ang[k];
  }
}
/* spatial/motion-dense.hpp:276-282 */
static inline void se3_actinv_motion(const se3_t* M, const double* m, double* o) {
  double txa[3], tmp[3], lin[3], ang[3];
  cross3(M->t, m + 3, txa);
  for (int k = 0; k < 3; ++k) tmp[k] = m[k] - txa[k];
  matTvec3(M->R, tmp, lin);
  matTvec3(M->R, m + 3, ang);
  for (int k = 0; k < 3; ++k) {
    o[k] = lin[k];
    o[3 + k] = ang[k];
  }
}
/* spatial/force-dense.hpp:217-223 */
static inline void se3_act_force(const se3_t* M, const double* f, double* o) {
  double lin[3], ang[3], txl[3];
  matvec3(M->R, f, lin);
  matvec3(M->R, f + 3, ang);
  cross3(M->t, lin, txl);
  for (int k = 0; k < 3; ++k) {
    o[k] = lin[k];
    o[3 + k] = ang[k] + txl[k];
  }
}
/* spatial/motion-dense.hpp:224-229  v x m */
static inline void motion_cross(const double* v, const double* m, double* o) {
  double a[3], b[3], c[3];
  cross3(v, m + 3, a);
  cross3(v + 3, m, b);
  cross3(v + 3, m + 3, c);
  for (int k = 0; k < 3; ++k) {
    o[k] = a[k] + b[k];
    o[3 + k] = c[k];
  }
}
/* spatial/force-dense.hpp:180-185  v x* f */
static inline void force_cross(const double* v, const double* f, double* o) {
  double a[3], b[3], c[3];
  cross3(v + 3, f, a);
  cross3(v + 3, f + 3, b);
  cross3(v, f, c);
  for (int k = 0; k < 3; ++k) {
    o[k] = a[k];
    o[3 + k] = b[k] + c[k];
  }
}

/* ------------------------------------------------------------------ Inertia */
/* spatial/inertia.hpp:741-748 */
static inline void inertia_mul(const inertia_t* Y, const double* v, double* f) {
  double cxw[3], lin[3], ang[3], cxf[3];
  cross3(Y->c, v + 3, cxw);
  for (int k = 0; k < 3; ++k) lin[k] = Y->m * (v[k] - cxw[k]);
  sym3_mulvec(Y->I, v + 3, ang);
  cross3(Y->c, lin, cxf);
  for (int k = 0; k < 3; ++k) {
    f[k] = lin[k];
    f[3 + k] = ang[k] + cxf[k];
  }
}
/* spatial/inertia.hpp:480-491: [[m 1, -m[c]x],[m[c]x, I - m[c]x[c]x]] row-major 6x6 */
static void inertia_matrix(const inertia_t* Y, double* M) {
  double cx[9], cxcx[9], I3[9];
  skew3(Y->c, cx);
  matmul3(cx, cx, cxcx);
  sym3_to_mat(Y->I, I3);
  memset(M, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; ++i) {
    M[6 * i + i] = Y->m;
    for (int j = 0; j < 3; ++j) {
      M[6 * i + 3 + j] = -Y->m * cx[3 * i + j];
      M[6 * (3 + i) + j] = Y->m * cx[3 * i + j];
      M[6 * (3 + i) + 3 + j] = I3[3 * i + j] - Y->m * cxcx[3 * i + j];
    }
  }
}
/* spatial/inertia.hpp:885-893 + symmetric3.hpp:560-601: (m, t + R c, R I R^T) */
static void se3_act_inertia(const se3_t* M, const inertia_t* Y, inertia_t* O) {
  double I3[9], T[9], RT[9], RIRt[9], Rc[3];
  sym3_to_mat(Y->I, I3);
  matmul3(M->R, I3, T);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) RT[3 * i + j] = M->R[3 * j + i];
  matmul3(T, RT, RIRt);
  matvec3(M->R, Y->c, Rc);
  O->m = Y->m;
  for (int k = 0; k < 3; ++k) O->c[k] = M->t[k] + Rc[k];
  O->I[0] = RIRt[0];
  O->I[1] = RIRt[1];
  O->I[2] = RIRt[4];
  O->I[3] = RIRt[2];
  O->I[4] = RIRt[5];
  O->I[5] = RIRt[8];
}
/* spatial/inertia.hpp:672-686 __pequ__ */
static void inertia_pequ(inertia_t* Ya, const inertia_t* Yb) {
  const double eps = 2.220446049250313e-16;
  double mab = Ya->m + Yb->m;
  double mab_inv = 1.0 / (mab > eps ? mab : eps);
  double AB[3], sx[9], sxsx[9];
  for (int k = 0; k < 3; ++k) AB[k] = Ya->c[k] - Yb->c[k];
  for (int k = 0; k < 3; ++k) {
    Ya->c[k] *= (Ya->m * mab_inv);
    Ya->c[k] += (Yb->m * mab_inv) * Yb->c[k];
  }
  skew3(AB, sx);
  matmul3(sx, sx, sxsx);
  double coef = Ya->m * Yb->m * mab_inv;
  const int idx[6] = {0, 1, 4, 2, 5, 8};
  for (int k = 0; k < 6; ++k) {
    Ya->I[k] += Yb->I[k];
    Ya->I[k] -= coef * sxsx[idx[k]];
  }
  Ya->m = mab;
}
/* spatial/inertia.hpp:762-789 variation(v) -> row-major 6x6 */
static void inertia_variation(const inertia_t* Y, const double* v, double* res) {
  double mv[6];
  for (int k = 0; k < 6; ++k) mv[k] = v[k] * Y->m;
  double s_mvl[9], s_mva[9], s_c[9], s_w[9], A[9], Bm[9], LA[9], AA[9], LL[9], T1[9], T2[9], I3[9];
  skew3(mv, s_mvl);
  skew3(mv + 3, s_mva);
  skew3(Y->c, s_c);
  skew3(v + 3, s_w);
  matmul3(s_mva, s_c, A);  /* skewSquare(mv.ang, c) */
  matmul3(s_c, s_mva, Bm); /* skewSquare(c, mv.ang) */
  for (int k = 0; k < 9; ++k) LA[k] = -s_mvl[k] - A[k] + Bm[k];
  matmul3(s_mvl, s_c, A);
  matmul3(s_c, s_mvl, Bm);
  for (int k = 0; k < 9; ++k) AA[k] = -A[k] - Bm[k];
  sym3_to_mat(Y->I, I3);
  matmul3(s_c, s_c, A);
  for (int k = 0; k < 9; ++k) LL[k] = I3[k] - Y->m * A[k];
  matmul3(LL, s_w, T1);
  matmul3(s_w, LL, T2);
  for (int k = 0; k < 9; ++k) AA[k] += -T1[k] + T2[k];
  memset(res, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      res[6 * i + 3 + j] = LA[3 * i + j];
      res[6 * (3 + i) + j] = LA[3 * j + i];
      res[6 * (3 + i) + 3 + j] = AA[3 * i + j];
    }
}

/* ------------------------------------------------------------------ dense 6x6 */
static inline void mat6_mulvec(const double* M, const double* x, double* o) {
  double t[6];
  for (int i = 0; i < 6; ++i) {
    double s = 0;
    for (int j = 0; j < 6; ++j) s += M[6 * i + j] * x[j];
    t[i] = s;
  }
  memcpy(o, t, sizeof t);
}
static inline double dot6(const double* a, const double* b) {
  double s = 0;
  for (int k = 0; k < 6; ++k) s += a[k] * b[k];
  return s;
}

/* Symmetric positive-definite inverse, n <= 6 (math/matrix-inverse.hpp:142-147:
 * Eigen .inverse() for n<=4, generated LDL^T-style sequence for n=6; any exact
 * symmetric inverse is equivalent at 1e-10).  Cholesky here. */
static int spd_inverse(int n, const double* A, double* Ainv) {
  double L[36], Li[36];
  memset(L, 0, sizeof L);
  memset(Li, 0, sizeof Li);
  for (int j = 0; j < n; ++j) {
    double d = A[n * j + j];
    for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
    if (!(d > 0)) return -1;
    d = sqrt(d);
    L[6 * j + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[n * i + j];
      for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
      L[6 * i + j] = s / d;
    }
  }
  for (int j = 0; j < n; ++j) {
    Li[6 * j + j] = 1.0 / L[6 * j + j];
    for (int i = j + 1; i < n; ++i) {
      double s = 0;
      for (int k = j; k < i; ++k) s -= L[6 * i + k] * Li[6 * k + j];
      Li[6 * i + j] = s / L[6 * i + i];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0;
      int k0 = i > j ? i : j;
      for (int k = k0; k < n; ++k) s += Li[6 * k + i] * Li[6 * k + j];
      Ainv[n * i + j] = s;
    }
  return 0;
}

/* ------------------------------------------------------------------ joints */
static const int JOINT_NQ[BRBD_JOINT_TYPE_COUNT] = {1, 1, 1, 1, 1, 1, 7, 4, 1, 1, 2, 2, 2, 2};
static const int JOINT_NV[BRBD_JOINT_TYPE_COUNT] = {1, 1, 1, 1, 1, 1, 6, 3, 1, 1, 1, 1, 1, 1};

/* Eigen Quaternion::toRotationMatrix (third-party, standard form, no
 * normalisation; used at joint-free-flyer.hpp:338-339, joint-spherical.hpp:528-529) */
static void quat_to_R(const double* qq, double* R) {
  const double x = qq[0], y = qq[1], z = qq[2], w = qq[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1 - (txx + tyy);
}

typedef struct {
  int nv;
  double S[6][6]; /* S[k] = k-th column of the motion subspace, joint frame */
  double vJ[6];   /* S * v_joint */
} jdata_t;

static inline void col3(const double* R, int c, double* o) {
  o[0] = R[c];
  o[1] = R[3 + c];
  o[2] = R[6 + c];
}
static inline void setcol3(double* R, int c, const double* v) {
  R[c] = v[0];
  R[3 + c] = v[1];
  R[6 + c] = v[2];
}

/* liMi = jointPlacement * M(q), S, vJ.  vj may be NULL (q-only variants). */
static int joint_calc(int jt, const se3_t* P, const double* ax, const double* q, const double* vj,
                      se3_t* liMi, jdata_t* jd) {
  memset(jd, 0, sizeof *jd);
  jd->nv = JOINT_NV[jt];
  switch (jt) {
    case BRBD_JOINT_RX:
    case BRBD_JOINT_RY:
    case BRBD_JOINT_RZ:
    case BRBD_JOINT_RUBX:
    case BRBD_JOINT_RUBY:
    case BRBD_JOINT_RUBZ: {
      double s, c;
      int k;
      if (jt <= BRBD_JOINT_RZ) {
        k = jt - BRBD_JOINT_RX;
        /* joint-revolute.hpp:801-830 calc -> math/sincos.hpp:47-61 (glibc sincos) */
        sincos(q[0], &s, &c);
      } else {
        k = jt - BRBD_JOINT_RUBX;
        /* joint-revolute-unbounded.hpp:153-162: q = (cos, sin) */
        c = q[0];
        s = q[1];
      }
      /* joint-revolute.hpp:132-166 TransformRevolute::se3Action(placement):
       * two rotated columns + one cross product, translation unchanged */
      const int ka = (k + 1) % 3, kb = (k + 2) % 3;
      double ck[3], ca[3], cb[3], na[3], nb[3];
      col3(P->R, k, ck);
      col3(P->R, ka, ca);
      col3(P->R, kb, cb);
      for (int r = 0; r < 3; ++r) na[r] = c * ca[r] + s * cb[r];
      cross3(ck, na, nb);
      setcol3(liMi->R, k, ck);
      setcol3(liMi->R, ka, na);
      setcol3(liMi->R, kb, nb);
      memcpy(liMi->t, P->t, sizeof liMi->t);
      jd->S[0][3 + k] = 1.0; /* joint-revolute.hpp:462-471 */
      if (vj) jd->vJ[3 + k] = vj[0];
      return 0;
    }
    case BRBD_JOINT_PX:
    case BRBD_JOINT_PY:
    case BRBD_JOINT_PZ: {
      /* joint-prismatic.hpp:262-272: R' = R_p, t' = t_p + q R_p[:,k] */
      const int k = jt - BRBD_JOINT_PX;
      memcpy(liMi->R, P->R, sizeof liMi->R);
      for (int r = 0; r < 3; ++r) liMi->t[r] = P->t[r] + q[0] * P->R[3 * r + k];
      jd->S[0][k] = 1.0;
      if (vj) jd->vJ[k] = vj[0];
      return 0;
    }
    case BRBD_JOINT_FREEFLYER: {
      /* joint-free-flyer.hpp:332-374 */
      se3_t M;
      quat_to_R(q + 3, M.R);
      M.t[0] = q[0];
      M.t[1] = q[1];
      M.t[2] = q[2];
      se3_mul(P, &M, liMi);
      for (int k = 0; k < 6; ++k) jd->S[k][k] = 1.0;
      if (vj)
        for (int k = 0; k < 6; ++k) jd->vJ[k] = vj[k];
      return 0;
    }
    case BRBD_JOINT_SPHERICAL: {
      /* joint-spherical.hpp:525-589 */
      se3_t M;
      quat_to_R(q, M.R);
      M.t[0] = M.t[1] = M.t[2] = 0;
      se3_mul(P, &M, liMi);
      for (int k = 0; k < 3; ++k) jd->S[k][3 + k] = 1.0;
      if (vj)
        for (int k = 0; k < 3; ++k) jd->vJ[3 + k] = vj[k];
      return 0;
    }
    case BRBD_JOINT_REVOLUTE_UNALIGNED:
    case BRBD_JOINT_REVOLUTE_UNBOUNDED_UNALIGNED: {
      /* joint-revolute-unaligned.hpp:667-708: R = toRotationMatrix(axis, cos, sin)
       * (math/rotation.hpp, Rodrigues); joint-revolute-unbounded-unaligned.hpp:190-200: the same
       * with q = (cos, sin) */
      double s, c;
      if (jt == BRBD_JOINT_REVOLUTE_UNALIGNED) {
        sincos(q[0], &s, &c);
      } else {
        c = q[0];
        s = q[1];
      }
      se3_t M;
      const double x = ax[0], y = ax[1], z = ax[2], c1 = 1 - c;
      double sx = s * x, sy = s * y, sz = s * z;
      double c1x = c1 * x, c1y = c1 * y, c1z = c1 * z;
      M.R[0] = c + c1x * x;
      M.R[1] = c1x * y - sz;
      M.R[2] = c1x * z + sy;
      M.R[3] = c1x * y + sz;
      M.R[4] = c + c1y * y;
      M.R[5] = c1y * z - sx;
      M.R[6] = c1x * z - sy;
      M.R[7] = c1y * z + sx;
      M.R[8] = c + c1z * z;
      M.t[0] = M.t[1] = M.t[2] = 0;
      se3_mul(P, &M, liMi);
      for (int k = 0; k < 3; ++k) jd->S[0][3 + k] = ax[k];
      if (vj)
        for (int k = 0; k < 3; ++k) jd->vJ[3 + k] = ax[k] * vj[0];
      return 0;
    }
    case BRBD_JOINT_PRISMATIC_UNALIGNED: {
      se3_t M;
      memset(&M, 0, sizeof M);
      M.R[0] = M.R[4] = M.R[8] = 1;
      for (int k = 0; k < 3; ++k) M.t[k] = ax[k] * q[0];
      se3_mul(P, &M, liMi);
      for (int k = 0; k < 3; ++k) jd->S[0][k] = ax[k];
      if (vj)
        for (int k = 0; k < 3; ++k) jd->vJ[k] = ax[k] * vj[0];
      return 0;
    }
    default:
      return -1;
  }
}

/* ------------------------------------------------------------------ model access */
static void get_placement(const brbd_model_desc* m, int i, se3_t* P) {
  const double* p = m->placement + 12 * (size_t)i;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) P->R[3 * r + c] = p[3 * c + r]; /* column-major in */
  P->t[0] = p[9];
  P->t[1] = p[10];
  P->t[2] = p[11];
}
static void get_inertia(const brbd_model_desc* m, int i, inertia_t* Y) {
  const double* p = m->inertia + 10 * (size_t)i;
  Y->m = p[0];
  Y->c[0] = p[1];
  Y->c[1] = p[2];
  Y->c[2] = p[3];
  memcpy(Y->I, p + 4, 6 * sizeof(double));
}
static const double ZERO3[3] = {0, 0, 0};
static const double* get_axis(const brbd_model_desc* m, int i) {
  return m->axis ? m->axis + 3 * (size_t)i : ZERO3;
}
static int check_model(const brbd_model_desc* m) {
  if (!m || m->njoints < 1 || !m->parents || !m->jtype || !m->idx_q || !m->idx_v || !m->placement ||
      !m->inertia)
    return -1;
  for (int i = 1; i < m->njoints; ++i) {
    if (m->parents[i] < 0 || m->parents[i] >= i) return -1;
    if (m->jtype[i] < 0 || m->jtype[i] >= BRBD_JOINT_TYPE_COUNT) return -1;
    if (m->idx_q[i] < 0 || m->idx_q[i] + JOINT_NQ[m->jtype[i]] > m->nq) return -1;
    if (m->idx_v[i] < 0 || m->idx_v[i] + JOINT_NV[m->jtype[i]] > m->nv) return -1;
  }
  return 0;
}

/* ------------------------------------------------------------------ workspace */
typedef struct {
  int n, nv;
  se3_t* liMi;
  se3_t* oMi;
  jdata_t* jd;
  double* v;     /* n x 6 */
  double* a;     /* n x 6 */
  double* f;     /* n x 6 */
  double* h;     /* n x 6 */
  double* Yaba;  /* n x 36 */
  double* U;     /* n x 36 (6 x nvj, column k at U + 6k) */
  double* UDinv; /* n x 36 */
  double* Dinv;  /* n x 36 */
  double* u;     /* nv */
  double* J;     /* 6 x nv col-major */
  inertia_t* Ycrb;
  /* derivatives */
  double *dJ, *dVdq, *dAdq, *dAdv, *dFdq, *dFdv, *dFda, *dYtJ; /* 6 x nv each */
  double* doYcrb;                                               /* n x 36 */
  double* oa_gf;                                                /* n x 6 */
  double* oa;                                                   /* n x 6 */
  int* nvSubtree;
} ws_t;

static ws_t* ws_alloc(const brbd_model_desc* m) {
  ws_t* w = (ws_t*)calloc(1, sizeof(ws_t));
  size_t n = (size_t)m->njoints, nv = (size_t)(m->nv > 0 ? m->nv : 1);
  w->n = m->njoints;
  w->nv = m->nv;
  w->liMi = (se3_t*)calloc(n, sizeof(se3_t));
  w->oMi = (se3_t*)calloc(n, sizeof(se3_t));
  w->jd = (jdata_t*)calloc(n, sizeof(jdata_t));
  w->v = (double*)calloc(n * 6, 8);
  w->a = (double*)calloc(n * 6, 8);
  w->f = (double*)calloc(n * 6, 8);
  w->h = (double*)calloc(n * 6, 8);
  w->Yaba = (double*)calloc(n * 36, 8);
  w->U = (double*)calloc(n * 36, 8);
  w->UDinv = (double*)calloc(n * 36, 8);
  w->Dinv = (double*)calloc(n * 36, 8);
  w->u = (double*)calloc(nv, 8);
  w->J = (double*)calloc(6 * nv, 8);
  w->Ycrb = (inertia_t*)calloc(n, sizeof(inertia_t));
  w->dJ = (double*)calloc(6 * nv, 8);
  w->dVdq = (double*)calloc(6 * nv, 8);
  w->dAdq = (double*)calloc(6 * nv, 8);
  w->dAdv = (double*)calloc(6 * nv, 8);
  w->dFdq = (double*)calloc(6 * nv, 8);
  w->dFdv = (double*)calloc(6 * nv, 8);
  w->dFda = (double*)calloc(6 * nv, 8);
  w->dYtJ = (double*)calloc(6 * nv, 8);
  w->doYcrb = (double*)calloc(n * 36, 8);
  w->oa_gf = (double*)calloc(n * 6, 8);
  w->oa = (double*)calloc(n * 6, 8);
  w->nvSubtree = (int*)calloc(n, sizeof(int));
  /* multibody/data.hxx:205-230: nvSubtree from the last joint of the subtree
   * (subtree = contiguous index range when joints are in depth-first order) */
  for (int i = (int)n - 1; i >= 1; --i) {
    int last = i;
    for (int j = i + 1; j < (int)n; ++j) {
      int p = j;
      while (p > i) p = m->parents[p];
      if (p == i) last = j;
    }
    w->nvSubtree[i] = m->idx_v[last] + JOINT_NV[m->jtype[last]] - m->idx_v[i];
  }
  return w;
}
static void ws_free(ws_t* w) {
  if (!w) return;
  free(w->liMi);
  free(w->oMi);
  free(w->jd);
  free(w->v);
  free(w->a);
  free(w->f);
  free(w->h);
  free(w->Yaba);
  free(w->U);
  free(w->UDinv);
  free(w->Dinv);
  free(w->u);
  free(w->J);
  free(w->Ycrb);
  free(w->dJ);
  free(w->dVdq);
  free(w->dAdq);
  free(w->dAdv);
  free(w->dFdq);
  free(w->dFdv);
  free(w->dFda);
  free(w->dYtJ);
  free(w->doYcrb);
  free(w->oa_gf);
  free(w->oa);
  free(w->nvSubtree);
  free(w);
}

/* ================================================================== RNEA */
/* algorithm/rnea.hxx:117-161; forward step :45-79, backward step :92-107 */
/* fext (may be NULL): 6 doubles [lin; ang] per joint, joint 0 included and ignored, expressed in
 * the joint frame: rnea.hxx:171-219 (data.f[i] -= fext[i], :205) */
static void rnea_ws(const brbd_model_desc* m, ws_t* w, const double* q, const double* v,
                    const double* a, const double* fext, double* tau) {
  const int n = m->njoints;
  for (int k = 0; k < m->nv; ++k) tau[k] = 0.0; /* :136 */
  for (int k = 0; k < 6; ++k) w->v[k] = 0.0;     /* :137 */
  for (int k = 0; k < 3; ++k) {                  /* :138 a_gf[0] = -gravity */
    w->a[k] = -m->gravity[k];
    w->a[3 + k] = 0.0;
  }
  for (int i = 1; i < n; ++i) {
    const int p = m->parents[i], jt = m->jtype[i];
    se3_t P;
    inertia_t Y;
    get_placement(m, i, &P);
    get_inertia(m, i, &Y);
    jdata_t* jd = &w->jd[i];
    joint_calc(jt, &P, get_axis(m, i), q + m->idx_q[i], v + m->idx_v[i], &w->liMi[i], jd); /* :59-61 */
    double* vi = w->v + 6 * i;
    double* ai = w->a + 6 * i;
    double tmp[6];
    memcpy(vi, jd->vJ, 6 * sizeof(double)); /* :63 */
    if (p > 0) {                            /* :64-65 */
      se3_actinv_motion(&w->liMi[i], w->v + 6 * p, tmp);
      for (int k = 0; k < 6; ++k) vi[k] += tmp[k];
    }
    motion_cross(vi, jd->vJ, ai); /* :67 (c = 0 for every supported joint) */
    for (int c = 0; c < jd->nv; ++c) /* :68 */
      for (int k = 0; k < 6; ++k) ai[k] += jd->S[c][k] * a[m->idx_v[i] + c];
    se3_actinv_motion(&w->liMi[i], w->a + 6 * p, tmp); /* :69 (not guarded by parent>0) */
    for (int k = 0; k < 6; ++k) ai[k] += tmp[k];
    inertia_mul(&Y, vi, w->h + 6 * i); /* :73 */
    inertia_mul(&Y, ai, w->f + 6 * i); /* :74 */
    force_cross(vi, w->h + 6 * i, tmp); /* :75 */
    for (int k = 0; k < 6; ++k) w->f[6 * i + k] += tmp[k];
    if (fext)
      for (int k = 0; k < 6; ++k) w->f[6 * i + k] -= fext[6 * i + k]; /* :205 */
  }
  for (int i = n - 1; i > 0; --i) {
    const int p = m->parents[i];
    jdata_t* jd = &w->jd[i];
    for (int c = 0; c < jd->nv; ++c) tau[m->idx_v[i] + c] += dot6(jd->S[c], w->f + 6 * i); /* :103 */
    if (p > 0) {                                                                          /* :105-106 */
      double tmp[6];
      se3_act_force(&w->liMi[i], w->f + 6 * i, tmp);
      for (int k = 0; k < 6; ++k) w->f[6 * p + k] += tmp[k];
    }
  }
  if (m->armature)
    for (int k = 0; k < m->nv; ++k) tau[k] += m->armature[k] * a[k]; /* :158 */
}

/* ================================================================== ABA (WORLD) */
/* algorithm/aba.hxx:243-294; pass1 :102-139, pass2 :153-193, pass3 :207-232 */
/* fext as for rnea_ws: aba.hxx:296-354 (data.of[i] -= data.oMi[i].act(fext[i]), :338) */
static int aba_world_ws(const brbd_model_desc* m, ws_t* w, const double* q, const double* v,
                        const double* tau, const double* fext, double* ddq) {
  const int n = m->njoints;
  double* ov = w->v;
  double* oa_gf = w->a;
  double* of = w->f;
  for (int k = 0; k < 3; ++k) { /* :261 */
    oa_gf[k] = -m->gravity[k];
    oa_gf[3 + k] = 0.0;
  }
  for (int k = 0; k < 6; ++k) of[k] = 0.0;           /* :262 */
  for (int k = 0; k < m->nv; ++k) w->u[k] = tau[k]; /* :263 */

  for (int i = 1; i < n; ++i) { /* pass 1 */
    const int p = m->parents[i], jt = m->jtype[i];
    se3_t P;
    inertia_t Y;
    get_placement(m, i, &P);
    get_inertia(m, i, &Y);
    jdata_t* jd = &w->jd[i];
    joint_calc(jt, &P, get_axis(m, i), q + m->idx_q[i], v + m->idx_v[i], &w->liMi[i], jd); /* :115-118 */
    if (p > 0) /* :119-122 */
      se3_mul(&w->oMi[p], &w->liMi[i], &w->oMi[i]);
    else
      w->oMi[i] = w->liMi[i];
    for (int c = 0; c < jd->nv; ++c) /* :124 */
      se3_act_motion(&w->oMi[i], jd->S[c], w->J + 6 * (m->idx_v[i] + c));
    double* ovi = ov + 6 * i;
    se3_act_motion(&w->oMi[i], jd->vJ, ovi); /* :126 */
    if (p > 0)                               /* :127-128 */
      for (int k = 0; k < 6; ++k) ovi[k] += ov[6 * p + k];
    for (int k = 0; k < 6; ++k) oa_gf[6 * i + k] = 0.0; /* :130 (c = 0) */
    if (p > 0) {                                        /* :131-132 */
      double tmp[6];
      motion_cross(ov + 6 * p, ovi, tmp);
      for (int k = 0; k < 6; ++k) oa_gf[6 * i + k] += tmp[k];
    }
    se3_act_inertia(&w->oMi[i], &Y, &w->Ycrb[i]);      /* :134 */
    inertia_matrix(&w->Ycrb[i], w->Yaba + 36 * i);     /* :135 */
    inertia_mul(&w->Ycrb[i], ovi, w->h + 6 * i);       /* :137 */
    force_cross(ovi, w->h + 6 * i, of + 6 * i);        /* :138 */
    if (fext) {                                         /* :338 */
      double tmp[6];
      se3_act_force(&w->oMi[i], fext + 6 * i, tmp);
      for (int k = 0; k < 6; ++k) of[6 * i + k] -= tmp[k];
    }
  }

  for (int i = n - 1; i > 0; --i) { /* pass 2 */
    const int p = m->parents[i];
    const int nvj = w->jd[i].nv, iv = m->idx_v[i];
    double* Ia = w->Yaba + 36 * i;
    double* Jc = w->J + 6 * iv;
    double* fi = of + 6 * i;
    double* U = w->U + 36 * i;
    double* UD = w->UDinv + 36 * i;
    double* Dinv = w->Dinv + 36 * i;
    double StU[36];
    for (int c = 0; c < nvj; ++c) w->u[iv + c] -= dot6(Jc + 6 * c, fi); /* :173 */
    for (int c = 0; c < nvj; ++c) mat6_mulvec(Ia, Jc + 6 * c, U + 6 * c); /* :175 */
    for (int r = 0; r < nvj; ++r)                                        /* :176 */
      for (int c = 0; c < nvj; ++c) StU[nvj * r + c] = dot6(Jc + 6 * r, U + 6 * c);
    if (m->armature)
      for (int c = 0; c < nvj; ++c) StU[nvj * c + c] += m->armature[iv + c]; /* :179 */
    if (spd_inverse(nvj, StU, Dinv)) return -1;                             /* :181 */
    for (int c = 0; c < nvj; ++c)                                           /* :182 */
      for (int k = 0; k < 6; ++k) {
        double s = 0;
        for (int r = 0; r < nvj; ++r) s += U[6 * r + k] * Dinv[nvj * r + c];
        UD[6 * c + k] = s;
      }
    if (p > 0) {
      for (int r = 0; r < 6; ++r) /* :186 */
        for (int c = 0; c < 6; ++c) {
          double s = 0;
          for (int k = 0; k < nvj; ++k) s += UD[6 * k + r] * U[6 * k + c];
          Ia[6 * r + c] -= s;
        }
      double t1[6];
      mat6_mulvec(Ia, oa_gf + 6 * i, t1); /* :188-189 */
      for (int k = 0; k < 6; ++k) {
        double s = t1[k];
        for (int c = 0; c < nvj; ++c) s += UD[6 * c + k] * w->u[iv + c];
        fi[k] += s;
      }
      for (int k = 0; k < 36; ++k) w->Yaba[36 * p + k] += Ia[k]; /* :190 */
      for (int k = 0; k < 6; ++k) of[6 * p + k] += fi[k];        /* :191 */
    }
  }

  for (int i = 1; i < n; ++i) { /* pass 3 */
    const int p = m->parents[i];
    const int nvj = w->jd[i].nv, iv = m->idx_v[i];
    double* Jc = w->J + 6 * iv;
    double* UD = w->UDinv + 36 * i;
    double* Dinv = w->Dinv + 36 * i;
    double* ai = oa_gf + 6 * i;
    for (int k = 0; k < 6; ++k) ai[k] += oa_gf[6 * p + k]; /* :223 */
    for (int r = 0; r < nvj; ++r) {                        /* :224-226 */
      double s = 0;
      for (int c = 0; c < nvj; ++c) s += Dinv[nvj * r + c] * w->u[iv + c];
      ddq[iv + r] = s - dot6(UD + 6 * r, ai);
    }
    for (int c = 0; c < nvj; ++c) /* :227 */
      for (int k = 0; k < 6; ++k) ai[k] += Jc[6 * c + k] * ddq[iv + c];
  }
  return 0;
}

/* ================================================================== ABA (LOCAL) */
/* algorithm/aba.hxx:488-542; pass1 :378-403, pass2 :417-443, pass3 :457-477;
 * calc_aba e.g. joint-revolute.hpp:832-846; SE3actOn aba.hxx:25-76 */
/* fext as for rnea_ws: aba.hxx:544-603 (data.f[i] -= fext[i], :587) */
static int aba_local_ws(const brbd_model_desc* m, ws_t* w, const double* q, const double* v,
                        const double* tau, const double* fext, double* ddq) {
  const int n = m->njoints;
  for (int k = 0; k < 6; ++k) w->v[k] = 0.0;
  for (int k = 0; k < 3; ++k) {
    w->a[k] = -m->gravity[k];
    w->a[3 + k] = 0.0;
  }
  for (int k = 0; k < 6; ++k) w->f[k] = 0.0;
  for (int k = 0; k < m->nv; ++k) w->u[k] = tau[k];
  for (int i = 1; i < n; ++i) {
    const int p = m->parents[i], jt = m->jtype[i];
    se3_t P;
    inertia_t Y;
    get_placement(m, i, &P);
    get_inertia(m, i, &Y);
    jdata_t* jd = &w->jd[i];
    joint_calc(jt, &P, get_axis(m, i), q + m->idx_q[i], v + m->idx_v[i], &w->liMi[i], jd);
    double* vi = w->v + 6 * i;
    double tmp[6];
    memcpy(vi, jd->vJ, 6 * sizeof(double));
    if (p > 0) {
      se3_actinv_motion(&w->liMi[i], w->v + 6 * p, tmp);
      for (int k = 0; k < 6; ++k) vi[k] += tmp[k];
    }
    motion_cross(vi, jd->vJ, w->a + 6 * i); /* :398 */
    inertia_matrix(&Y, w->Yaba + 36 * i);   /* :400 */
    inertia_mul(&Y, vi, w->h + 6 * i);      /* :401 */
    force_cross(vi, w->h + 6 * i, w->f + 6 * i);
    if (fext)
      for (int k = 0; k < 6; ++k) w->f[6 * i + k] -= fext[6 * i + k]; /* :587 */
  }
  for (int i = n - 1; i > 0; --i) {
    const int p = m->parents[i];
    const int nvj = w->jd[i].nv, iv = m->idx_v[i];
    jdata_t* jd = &w->jd[i];
    double* Ia = w->Yaba + 36 * i;
    double* U = w->U + 36 * i;
    double* UD = w->UDinv + 36 * i;
    double* Dinv = w->Dinv + 36 * i;
    double StU[36];
    for (int c = 0; c < nvj; ++c) w->u[iv + c] -= dot6(jd->S[c], w->f + 6 * i); /* :431 */
    for (int c = 0; c < nvj; ++c) mat6_mulvec(Ia, jd->S[c], U + 6 * c);        /* calc_aba */
    for (int r = 0; r < nvj; ++r)
      for (int c = 0; c < nvj; ++c) StU[nvj * r + c] = dot6(jd->S[r], U + 6 * c);
    if (m->armature)
      for (int c = 0; c < nvj; ++c) StU[nvj * c + c] += m->armature[iv + c];
    if (spd_inverse(nvj, StU, Dinv)) return -1;
    for (int c = 0; c < nvj; ++c)
      for (int k = 0; k < 6; ++k) {
        double s = 0;
        for (int r = 0; r < nvj; ++r) s += U[6 * r + k] * Dinv[nvj * r + c];
        UD[6 * c + k] = s;
      }
    if (p > 0) {
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
          double s = 0;
          for (int k = 0; k < nvj; ++k) s += UD[6 * k + r] * U[6 * k + c];
          Ia[6 * r + c] -= s;
        }
      double* pa = w->f + 6 * i;
      double t1[6];
      mat6_mulvec(Ia, w->a + 6 * i, t1); /* :438-439 */
      for (int k = 0; k < 6; ++k) {
        double s = t1[k];
        for (int c = 0; c < nvj; ++c) s += UD[6 * c + k] * w->u[iv + c];
        pa[k] += s;
      }
      /* :440 Yaba[parent] += X Ia X^T with X = force action matrix of liMi
       * = [[R, 0], [[t]x R, R]] (aba.hxx:25-76) */
      const se3_t* M = &w->liMi[i];
      double X[36], T[36], tx[9], txR[9];
      skew3(M->t, tx);
      matmul3(tx, M->R, txR);
      memset(X, 0, sizeof X);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          X[6 * r + c] = M->R[3 * r + c];
          X[6 * (3 + r) + c] = txR[3 * r + c];
          X[6 * (3 + r) + 3 + c] = M->R[3 * r + c];
        }
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
          double s = 0;
          for (int k = 0; k < 6; ++k) s += X[6 * r + k] * Ia[6 * k + c];
          T[6 * r + c] = s;
        }
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
          double s = 0;
          for (int k = 0; k < 6; ++k) s += T[6 * r + k] * X[6 * c + k];
          w->Yaba[36 * p + 6 * r + c] += s;
        }
      double t2[6];
      se3_act_force(M, pa, t2); /* :441 */
      for (int k = 0; k < 6; ++k) w->f[6 * p + k] += t2[k];
    }
  }
  for (int i = 1; i < n; ++i) {
    const int p = m->parents[i];
    const int nvj = w->jd[i].nv, iv = m->idx_v[i];
    jdata_t* jd = &w->jd[i];
    double* UD = w->UDinv + 36 * i;
    double* Dinv = w->Dinv + 36 * i;
    double* ai = w->a + 6 * i;
    double tmp[6];
    se3_actinv_motion(&w->liMi[i], w->a + 6 * p, tmp); /* :468 */
    for (int k = 0; k < 6; ++k) ai[k] += tmp[k];
    for (int r = 0; r < nvj; ++r) { /* :469-471 */
      double s = 0;
      for (int c = 0; c < nvj; ++c) s += Dinv[nvj * r + c] * w->u[iv + c];
      ddq[iv + r] = s - dot6(UD + 6 * r, ai);
    }
    for (int c = 0; c < nvj; ++c) /* :472 */
      for (int k = 0; k < 6; ++k) ai[k] += jd->S[c][k] * ddq[iv + c];
  }
  return 0;
}

/* ================================================================== FK + Jacobians */
/* algorithm/kinematics.hxx:52-97, algorithm/jacobian.hxx:38-95 */
static void fkjac_ws(const brbd_model_desc* m, ws_t* w, const double* q, double* J, double* oMi) {
  const int n = m->njoints;
  if (oMi) {
    memset(oMi, 0, 12 * sizeof(double));
    oMi[0] = oMi[4] = oMi[8] = 1.0;
  }
  for (int i = 1; i < n; ++i) {
    const int p = m->parents[i], jt = m->jtype[i];
    se3_t P;
    get_placement(m, i, &P);
    jdata_t* jd = &w->jd[i];
    joint_calc(jt, &P, get_axis(m, i), q + m->idx_q[i], NULL, &w->liMi[i], jd); /* jacobian.hxx:50-52 */
    if (p > 0) /* :54-57 */
      se3_mul(&w->oMi[p], &w->liMi[i], &w->oMi[i]);
    else
      w->oMi[i] = w->liMi[i];
    if (J)
      for (int c = 0; c < jd->nv; ++c) /* :59 */
        se3_act_motion(&w->oMi[i], jd->S[c], J + 6 * (m->idx_v[i] + c));
    if (oMi) {
      double* o = oMi + 12 * i;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[3 * c + r] = w->oMi[i].R[3 * r + c]; /* column-major out */
      o[9] = w->oMi[i].t[0];
      o[10] = w->oMi[i].t[1];
      o[11] = w->oMi[i].t[2];
    }
  }
}

/* ================================================================== RNEA derivatives */
/* algorithm/rnea-derivatives.hxx:472-541; forward :263-338, backward :378-459 */
static void mat6_mul_cols(const double* M, const double* X, int ncols, double* Y, int add) {
  for (int c = 0; c < ncols; ++c) {
    double t[6];
    mat6_mulvec(M, X + 6 * c, t);
    for (int k = 0; k < 6; ++k) Y[6 * c + k] = (add ? Y[6 * c + k] : 0.0) + t[k];
  }
}
static void inertia_action_cols(const inertia_t* Y, const double* X, int ncols, double* F, int add) {
  /* spatial/act-on-set.hpp motionSet::inertiaAction */
  for (int c = 0; c < ncols; ++c) {
    double t[6];
    inertia_mul(Y, X + 6 * c, t);
    for (int k = 0; k < 6; ++k) F[6 * c + k] = (add ? F[6 * c + k] : 0.0) + t[k];
  }
}
static void rnea_derivatives_ws(const brbd_model_desc* m, ws_t* w, const double* q, const double* v,
                                const double* a, double* pdq, double* pdv, double* pda,
                                double* tau_out) {
  const int n = m->njoints, nv = m->nv;
  double* ov = w->h; /* reuse: world velocities */
  double* of = w->f;
  double* oh = w->U; /* reuse first 6 of each 36 block */
  double* tau = w->u;
  for (size_t k = 0; k < (size_t)nv * nv; ++k) {
    pdq[k] = 0.0;
    pdv[k] = 0.0;
    pda[k] = 0.0;
  }
  for (int k = 0; k < 3; ++k) { /* :504 */
    w->oa_gf[k] = -m->gravity[k];
    w->oa_gf[3 + k] = 0.0;
  }
  for (int i = 1; i < n; ++i) {
    const int p = m->parents[i], jt = m->jtype[i], iv = m->idx_v[i];
    se3_t P;
    inertia_t Y;
    get_placement(m, i, &P);
    get_inertia(m, i, &Y);
    jdata_t* jd = &w->jd[i];
    joint_calc(jt, &P, get_axis(m, i), q + m->idx_q[i], v + iv, &w->liMi[i], jd); /* :281-283 */
    const int nvj = jd->nv;
    double* vi = w->v + 6 * i;
    double* ai = w->a + 6 * i;
    double tmp[6];
    memcpy(vi, jd->vJ, 6 * sizeof(double)); /* :285 */
    if (p > 0) {                            /* :287-293 */
      se3_mul(&w->oMi[p], &w->liMi[i], &w->oMi[i]);
      se3_actinv_motion(&w->liMi[i], w->v + 6 * p, tmp);
      for (int k = 0; k < 6; ++k) vi[k] += tmp[k];
    } else
      w->oMi[i] = w->liMi[i];
    motion_cross(vi, jd->vJ, tmp); /* :295-296 */
    for (int k = 0; k < 6; ++k) {
      double s = 0;
      for (int c = 0; c < nvj; ++c) s += jd->S[c][k] * a[iv + c];
      ai[k] = s + tmp[k];
    }
    if (p > 0) { /* :297-300 */
      se3_actinv_motion(&w->liMi[i], w->a + 6 * p, tmp);
      for (int k = 0; k < 6; ++k) ai[k] += tmp[k];
    }
    se3_act_inertia(&w->oMi[i], &Y, &w->Ycrb[i]);          /* :302 */
    se3_act_motion(&w->oMi[i], vi, ov + 6 * i);            /* :303 */
    se3_act_motion(&w->oMi[i], ai, w->oa + 6 * i);         /* :304 */
    for (int k = 0; k < 6; ++k) w->oa_gf[6 * i + k] = w->oa[6 * i + k]; /* :305 */
    for (int k = 0; k < 3; ++k) w->oa_gf[6 * i + k] -= m->gravity[k];
    inertia_mul(&w->Ycrb[i], ov + 6 * i, oh + 36 * i);     /* :307 */
    inertia_mul(&w->Ycrb[i], w->oa_gf + 6 * i, of + 6 * i); /* :308 */
    force_cross(ov + 6 * i, oh + 36 * i, tmp);
    for (int k = 0; k < 6; ++k) of[6 * i + k] += tmp[k];
    double* Jc = w->J + 6 * iv;
    double* dJc = w->dJ + 6 * iv;
    double* dVdqc = w->dVdq + 6 * iv;
    double* dAdqc = w->dAdq + 6 * iv;
    double* dAdvc = w->dAdv + 6 * iv;
    for (int c = 0; c < nvj; ++c) {
      se3_act_motion(&w->oMi[i], jd->S[c], Jc + 6 * c);            /* :319 */
      motion_cross(ov + 6 * i, Jc + 6 * c, dJc + 6 * c);           /* :320 */
      motion_cross(w->oa_gf + 6 * p, Jc + 6 * c, dAdqc + 6 * c);   /* :321 */
      memcpy(dAdvc + 6 * c, dJc + 6 * c, 6 * sizeof(double));      /* :322 */
      if (p > 0) {                                                 /* :323-328 */
        motion_cross(ov + 6 * p, Jc + 6 * c, dVdqc + 6 * c);
        motion_cross(ov + 6 * p, dVdqc + 6 * c, tmp);
        for (int k = 0; k < 6; ++k) {
          dAdqc[6 * c + k] += tmp[k];
          dAdvc[6 * c + k] += dVdqc[6 * c + k];
        }
      } else
        for (int k = 0; k < 6; ++k) dVdqc[6 * c + k] = 0.0; /* :331 */
    }
    double* dY = w->doYcrb + 36 * i;
    inertia_variation(&w->Ycrb[i], ov + 6 * i, dY); /* :335 */
    /* :337, :340-351 addForceCrossMatrix(oh): addSkew(-f.lin) on LA and AL, addSkew(-f.ang) on AA */
    double sl[9], sa[9], nl[3], na[3];
    for (int k = 0; k < 3; ++k) {
      nl[k] = -oh[36 * i + k];
      na[k] = -oh[36 * i + 3 + k];
    }
    skew3(nl, sl);
    skew3(na, sa);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        dY[6 * r + 3 + c] += sl[3 * r + c];
        dY[6 * (3 + r) + c] += sl[3 * r + c];
        dY[6 * (3 + r) + 3 + c] += sa[3 * r + c];
      }
  }

  for (int i = n - 1; i > 0; --i) {
    const int p = m->parents[i], iv = m->idx_v[i];
    const int nvj = w->jd[i].nv;
    const int nvs = w->nvSubtree[i];
    const int ivp = iv + nvj, nvsp = nvs - nvj;
    double* Jc = w->J + 6 * iv;
    double* dVdqc = w->dVdq + 6 * iv;
    double* dAdqc = w->dAdq + 6 * iv;
    double* dAdvc = w->dAdv + 6 * iv;
    double* dFdqc = w->dFdq + 6 * iv;
    double* dFdvc = w->dFdv + 6 * iv;
    double* dFdac = w->dFda + 6 * iv;
    double* dYtJc = w->dYtJ + 6 * iv;
    double* dY = w->doYcrb + 36 * i;
    for (int c = 0; c < nvj; ++c) tau[iv + c] = dot6(Jc + 6 * c, of + 6 * i); /* :409-410 */
    inertia_action_cols(&w->Ycrb[i], Jc, nvj, dFdac, 0);                      /* :419 */
    for (int r = 0; r < nvj; ++r)                                             /* :420-421 */
      for (int c = 0; c < nvs; ++c)
        pda[(size_t)(iv + c) * nv + iv + r] = dot6(Jc + 6 * r, w->dFda + 6 * (iv + c));
    if (p > 0) { /* :424-430 */
      mat6_mul_cols(dY, dVdqc, nvj, dFdqc, 0);
      inertia_action_cols(&w->Ycrb[i], dAdqc, nvj, dFdqc, 1);
    } else
      inertia_action_cols(&w->Ycrb[i], dAdqc, nvj, dFdqc, 0);
    for (int c = 0; c < nvj; ++c) /* :432 dYtJ_cols^T = J_cols^T doYcrb */
      for (int k = 0; k < 6; ++k) {
        double s = 0;
        for (int r = 0; r < 6; ++r) s += Jc[6 * c + r] * dY[6 * r + k];
        dYtJc[6 * c + k] = s;
      }
    for (int r = 0; r < nvsp; ++r) /* :433-435 */
      for (int c = 0; c < nvj; ++c)
        pdq[(size_t)(iv + c) * nv + ivp + r] =
          dot6(w->dFda + 6 * (ivp + r), dAdqc + 6 * c) + dot6(w->dYtJ + 6 * (ivp + r), dVdqc + 6 * c);
    for (int r = 0; r < nvj; ++r) /* :437-438 */
      for (int c = 0; c < nvs; ++c)
        pdq[(size_t)(iv + c) * nv + iv + r] = dot6(Jc + 6 * r, w->dFdq + 6 * (iv + c));
    for (int c = 0; c < nvj; ++c) { /* :440 motionSet::act<ADDTO>(J_cols, of, dFdq_cols): m x* f */
      double t[6];
      force_cross(Jc + 6 * c, of + 6 * i, t);
      for (int k = 0; k < 6; ++k) dFdqc[6 * c + k] += t[k];
    }
    mat6_mul_cols(dY, Jc, nvj, dFdvc, 0);                  /* :443 */
    inertia_action_cols(&w->Ycrb[i], dAdvc, nvj, dFdvc, 1); /* :444 */
    for (int r = 0; r < nvsp; ++r)                         /* :446-448 */
      for (int c = 0; c < nvj; ++c)
        pdv[(size_t)(iv + c) * nv + ivp + r] =
          dot6(w->dFda + 6 * (ivp + r), dAdvc + 6 * c) + dot6(w->dYtJ + 6 * (ivp + r), Jc + 6 * c);
    for (int r = 0; r < nvj; ++r) /* :450-451 */
      for (int c = 0; c < nvs; ++c)
        pdv[(size_t)(iv + c) * nv + iv + r] = dot6(Jc + 6 * r, w->dFdv + 6 * (iv + c));
    if (p > 0) { /* :453-458 */
      inertia_pequ(&w->Ycrb[p], &w->Ycrb[i]);
      for (int k = 0; k < 36; ++k) w->doYcrb[36 * p + k] += dY[k];
      for (int k = 0; k < 6; ++k) of[6 * p + k] += of[6 * i + k];
    }
  }
  /* :538-540 armature */
  if (m->armature)
    for (int k = 0; k < nv; ++k) {
      tau[k] += m->armature[k] * a[k];
      pda[(size_t)k * nv + k] += m->armature[k];
    }
  if (tau_out) memcpy(tau_out, tau, nv * sizeof(double));
}

/* ================================================================== public API */
int brbd_oracle_rnea(const brbd_model_desc* m, const double* q, const double* v, const double* a,
                     double* tau) {
  if (check_model(m)) return -1;
  ws_t* w = ws_alloc(m);
  rnea_ws(m, w, q, v, a, NULL, tau);
  ws_free(w);
  return 0;
}
int brbd_oracle_aba_world(const brbd_model_desc* m, const double* q, const double* v,
                          const double* tau, double* ddq) {
  if (check_model(m)) return -1;
  ws_t* w = ws_alloc(m);
  int rc = aba_world_ws(m, w, q, v, tau, NULL, ddq);
  ws_free(w);
  return rc;
}
int brbd_oracle_aba_local(const brbd_model_desc* m, const double* q, const double* v,
                          const double* tau, double* ddq) {
  if (check_model(m)) return -1;
  ws_t* w = ws_alloc(m);
  int rc = aba_local_ws(m, w, q, v, tau, NULL, ddq);
  ws_free(w);
  return rc;
}
/* with external forces, one instance (convention: 0 = WORLD, 1 = LOCAL) */
int brbd_oracle_rnea_fext(const brbd_model_desc* m, const double* q, const double* v, const double* a,
                          const double* fext, double* tau) {
  if (check_model(m)) return -1;
  ws_t* w = ws_alloc(m);
  rnea_ws(m, w, q, v, a, fext, tau);
  ws_free(w);
  return 0;
}
int brbd_oracle_aba_fext(const brbd_model_desc* m, const double* q, const double* v, const double* tau,
                         const double* fext, int convention, double* ddq) {
  if (check_model(m)) return -1;
  ws_t* w = ws_alloc(m);
  int rc = convention ? aba_local_ws(m, w, q, v, tau, fext, ddq) : aba_world_ws(m, w, q, v, tau, fext, ddq);
  ws_free(w);
  return rc;
}
/* computeJointJacobians (jacobian.hxx:38-95) + getJointJacobian(joint, rf) (:175-283,315-326):
 * the columns of the joint's support (itself and its ancestors) of the world Jacobian, expressed
 * in rf: 0 WORLD (copied), 1 LOCAL (oMjoint.actInv), 2 LOCAL_WORLD_ALIGNED (linear -= t x angular);
 * J is 6 x nv column-major and must come zeroed (the other columns are not written). */
int brbd_oracle_joint_jacobian(const brbd_model_desc* m, const double* q, int joint, int rf, double* J) {
  if (check_model(m) || joint < 1 || joint >= m->njoints || rf < 0 || rf > 2) return -1;
  ws_t* w = ws_alloc(m);
  double* Jw = (double*)calloc((size_t)6 * m->nv + 6, sizeof(double));
  fkjac_ws(m, w, q, Jw, NULL);
  const se3_t* oM = &w->oMi[joint];
  for (int i = joint; i > 0; i = m->parents[i])
    for (int c = 0; c < JOINT_NV[m->jtype[i]]; ++c) {
      const double* in = Jw + 6 * (m->idx_v[i] + c);
      double* out = J + 6 * (m->idx_v[i] + c);
      if (rf == 0) {
        memcpy(out, in, 6 * sizeof(double));
      } else if (rf == 1) {
        se3_actinv_motion(oM, in, out);
      } else {
        double tx[3];
        cross3(oM->t, in + 3, tx);
        for (int k = 0; k < 3; ++k) {
          out[k] = in[k] - tx[k];
          out[3 + k] = in[3 + k];
        }
      }
    }
  free(Jw);
  ws_free(w);
  return 0;
}
int brbd_oracle_fk_jacobians(const brbd_model_desc* m, const double* q, double* J, double* oMi) {
  if (check_model(m)) return -1;
  ws_t* w = ws_alloc(m);
  if (J) memset(J, 0, sizeof(double) * 6 * (size_t)m->nv);
  fkjac_ws(m, w, q, J, oMi);
  ws_free(w);
  return 0;
}
int brbd_oracle_rnea_derivatives(const brbd_model_desc* m, const double* q, const double* v,
                                 const double* a, double* dq, double* dv, double* da, double* tau) {
  if (check_model(m)) return -1;
  ws_t* w = ws_alloc(m);
  double* da_tmp = da ? da : (double*)malloc(sizeof(double) * (size_t)m->nv * m->nv + 8);
  rnea_derivatives_ws(m, w, q, v, a, dq, dv, da_tmp, tau);
  if (!da) free(da_tmp);
  ws_free(w);
  return 0;
}

/* ================================================================== computeMinverse */
/* algorithm/aba.hxx:841-877; pass 1 :626-654, pass 2 :668-728, pass 3 :802-833.  Minv is nv x nv,
 * written here COLUMN-major like the other per-instance matrices (element (r, c) at c * nv + r);
 * as in the reference only the UPPER triangle (and the diagonal blocks of multi-dof joints) is
 * computed, the strictly lower part is zero (data.Minv starts as zeros, :853).  Needs depth-first
 * numbering (nvSubtree column ranges, like the derivatives). */
static int minverse_ws(const brbd_model_desc* m, ws_t* w, const double* q, double* Minv) {
  const int n = m->njoints, nv = m->nv;
  double* Fcrb = (double*)calloc((size_t)n * 6 * (size_t)(nv > 0 ? nv : 1), sizeof(double)); /* Fcrb[i]: 6 x nv col-major */
  double* SDinv = (double*)calloc((size_t)6 * (size_t)(nv > 0 ? nv : 1), sizeof(double));
  if (!Fcrb || !SDinv) return -1;
#define MINV(r, c) Minv[(size_t)(c) * nv + (r)]
  for (int k = 0; k < nv * nv; ++k) Minv[k] = 0.0; /* :853 */
  for (int i = 1; i < n; ++i) {                    /* pass 1 */
    const int p = m->parents[i];
    se3_t P;
    inertia_t Y;
    get_placement(m, i, &P);
    get_inertia(m, i, &Y);
    jdata_t* jd = &w->jd[i];
    joint_calc(m->jtype[i], &P, get_axis(m, i), q + m->idx_q[i], NULL, &w->liMi[i], jd); /* :636-639 */
    if (p > 0) se3_mul(&w->oMi[p], &w->liMi[i], &w->oMi[i]);                             /* :641-644 */
    else w->oMi[i] = w->liMi[i];
    for (int c = 0; c < jd->nv; ++c) se3_act_motion(&w->oMi[i], jd->S[c], w->J + 6 * (m->idx_v[i] + c)); /* :650 */
    se3_act_inertia(&w->oMi[i], &Y, &w->Ycrb[i]);  /* :652 */
    inertia_matrix(&w->Ycrb[i], w->Yaba + 36 * i); /* :653 */
  }
  double* F0 = Fcrb; /* data.Fcrb[0].setZero() :863 (calloc) */
  for (int i = n - 1; i > 0; --i) { /* pass 2 */
    const int p = m->parents[i];
    const int nvj = w->jd[i].nv, iv = m->idx_v[i], nsub = w->nvSubtree[i];
    double* Ia = w->Yaba + 36 * i;
    double* Jc = w->J + 6 * iv;
    double* U = w->U + 36 * i;
    double* UD = w->UDinv + 36 * i;
    double* Dinv = w->Dinv + 36 * i;
    double StU[36];
    for (int c = 0; c < nvj; ++c) mat6_mulvec(Ia, Jc + 6 * c, U + 6 * c); /* :689 */
    for (int r = 0; r < nvj; ++r)                                        /* :690 */
      for (int c = 0; c < nvj; ++c) StU[nvj * r + c] = dot6(Jc + 6 * r, U + 6 * c);
    if (m->armature)
      for (int c = 0; c < nvj; ++c) StU[nvj * c + c] += m->armature[iv + c]; /* :693 */
    if (spd_inverse(nvj, StU, Dinv)) { free(Fcrb); free(SDinv); return -1; } /* :695 */
    for (int c = 0; c < nvj; ++c)                                           /* :696 */
      for (int k = 0; k < 6; ++k) {
        double s = 0;
        for (int r = 0; r < nvj; ++r) s += U[6 * r + k] * Dinv[nvj * r + c];
        UD[6 * c + k] = s;
      }
    for (int r = 0; r < nvj; ++r) /* :698 */
      for (int c = 0; c < nvj; ++c) MINV(iv + r, iv + c) = Dinv[nvj * r + c];
    const int nch = nsub - nvj; /* :699 */
    if (nch > 0) {
      for (int c = 0; c < nvj; ++c) /* :703 SDinv_cols = J_cols * Dinv */
        for (int k = 0; k < 6; ++k) {
          double s = 0;
          for (int r = 0; r < nvj; ++r) s += Jc[6 * r + k] * Dinv[nvj * r + c];
          SDinv[6 * (iv + c) + k] = s;
        }
      for (int r = 0; r < nvj; ++r) /* :705-707 */
        for (int c = 0; c < nch; ++c) MINV(iv + r, iv + nvj + c) = -dot6(SDinv + 6 * (iv + r), F0 + 6 * (iv + nvj + c));
      if (p > 0) /* :709-715 Fcrb[0].middleCols(iv, nsub) += U * Minv.block(iv, iv, nvj, nsub) */
        for (int c = 0; c < nsub; ++c)
          for (int k = 0; k < 6; ++k) {
            double s = 0;
            for (int r = 0; r < nvj; ++r) s += U[6 * r + k] * MINV(iv + r, iv + c);
            F0[6 * (iv + c) + k] += s;
          }
    } else { /* :717-721 */
      for (int c = 0; c < nsub; ++c)
        for (int k = 0; k < 6; ++k) {
          double s = 0;
          for (int r = 0; r < nvj; ++r) s += U[6 * r + k] * MINV(iv + r, iv + c);
          F0[6 * (iv + c) + k] = s;
        }
    }
    if (p > 0) { /* :723-727 */
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
          double s = 0;
          for (int k = 0; k < nvj; ++k) s += UD[6 * k + r] * U[6 * k + c];
          Ia[6 * r + c] -= s;
        }
      for (int k = 0; k < 36; ++k) w->Yaba[36 * p + k] += Ia[k];
    }
  }
  for (int i = 1; i < n; ++i) { /* pass 3 */
    const int p = m->parents[i];
    const int nvj = w->jd[i].nv, iv = m->idx_v[i];
    double* Jc = w->J + 6 * iv;
    double* UD = w->UDinv + 36 * i;
    double* Fi = Fcrb + (size_t)i * 6 * nv;
    double* Fp = Fcrb + (size_t)p * 6 * nv;
    if (p > 0) /* :819-825 Minv.middleRows(iv, nvj).rightCols(nv - iv) -= UDinv^T * Fcrb[parent].rightCols(nv - iv) */
      for (int r = 0; r < nvj; ++r)
        for (int c = iv; c < nv; ++c) MINV(iv + r, c) -= dot6(UD + 6 * r, Fp + 6 * c);
    for (int c = iv; c < nv; ++c) /* :827-829 Fcrb[i].rightCols = J_cols * Minv.middleRows(iv, nvj).rightCols */
      for (int k = 0; k < 6; ++k) {
        double s = 0;
        for (int r = 0; r < nvj; ++r) s += Jc[6 * r + k] * MINV(iv + r, c);
        Fi[6 * c + k] = s;
      }
    if (p > 0) /* :830-832 */
      for (int c = iv; c < nv; ++c)
        for (int k = 0; k < 6; ++k) Fi[6 * c + k] += Fp[6 * c + k];
  }
#undef MINV
  free(Fcrb);
  free(SDinv);
  return 0;
}

int brbd_oracle_minverse(const brbd_model_desc* m, const double* q, double* Minv) {
  if (check_model(m) || !q || !Minv) return -1;
  ws_t* w = ws_alloc(m);
  const int rc = minverse_ws(m, w, q, Minv);
  ws_free(w);
  return rc;
}

/* ================================================================== integrate */
/* algorithm/joint-configuration.hxx integrate(model, q, v, qout): per joint the Lie-group
 * integrate_impl of its configuration space.
 *   vector spaces (revolute, prismatic, unaligned): q + v
 *   SO(2), revolute unbounded: multibody/liegroup/special-orthogonal.hpp:164-184
 *   SO(3), spherical:          special-orthogonal.hpp:495-510
 *   SE(3), free-flyer:         special-euclidean.hpp:773-812
 * quaternion::exp3 spatial/explog-quaternion.hpp:25-66, exp6 :92-136,
 * firstOrderNormalize math/quaternion.hpp:89-112, Taylor precision math/taylor-expansion.hpp:31-36 */
static double ts_prec3(void) { return sqrt(sqrt(2.220446049250313e-16)); } /* eps^(1/4) */

static void quat_exp3(const double* v, double* qo) { /* explog-quaternion.hpp:25-66; qo = (x, y, z, w) */
  const double eps = 2.220446049250313e-16;
  const double t2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const double t = sqrt(t2 + eps * eps);
  if (t2 > ts_prec3()) { /* Eigen::AngleAxis(t, v / t) -> quaternion */
    double s, c;
    sincos(0.5 * t, &s, &c);
    for (int k = 0; k < 3; ++k) qo[k] = s * (v[k] / t);
    qo[3] = c;
  } else { /* order-4 Taylor expansion */
    const double t2_2 = t2 / 4;
    const double a = 0.5 * (1.0 - t2_2 / 6.0 + t2_2 * t2_2 / 120.0);
    for (int k = 0; k < 3; ++k) qo[k] = a * v[k];
    qo[3] = 1.0 - t2_2 / 2 + t2_2 * t2_2 / 24;
  }
}
static void quat_mul(const double* a, const double* b, double* o) { /* Eigen quaternion product, (x, y, z, w) */
  const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  const double ww = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = ww;
}
static void quat_first_order_normalize(double* q) { /* math/quaternion.hpp:89-112 */
  const double N2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double alpha = (3.0 - N2) / 2.0;
  for (int k = 0; k < 4; ++k) q[k] *= alpha;
}

int brbd_oracle_integrate(const brbd_model_desc* m, const double* q, const double* v, double* qout) {
  if (check_model(m) || !q || !v || !qout) return -1;
  for (int i = 1; i < m->njoints; ++i) {
    const int jt = m->jtype[i], iq = m->idx_q[i], iv = m->idx_v[i];
    if (jt == BRBD_JOINT_FREEFLYER) { /* special-euclidean.hpp:773-812 */
      const double* lin = v + iv;
      const double* wv = v + iv + 3;
      const double eps = 2.220446049250313e-16;
      const double t2 = wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2] + eps * eps; /* exp6 :106-107 */
      const double t = sqrt(t2);
      double st, ct;
      sincos(t, &st, &ct);
      const double inv_t2 = 1.0 / t2;
      const double a_wxv = t < ts_prec3() ? 0.5 - t2 / 24.0 : (1.0 - ct) * inv_t2;            /* :119-123 */
      const double a_w2 = t < ts_prec3() ? 1.0 / 6.0 - t2 / 120.0 : (t - st) * inv_t2 / t;     /* :125-126 */
      double wxv[3], wwxv[3], trans[3], e3[4];
      cross3(wv, lin, wxv);
      cross3(wv, wxv, wwxv);
      for (int k = 0; k < 3; ++k) trans[k] = lin[k] + a_wxv * wxv[k] + a_w2 * wwxv[k]; /* :130 */
      quat_exp3(wv, e3);                                                               /* :135 */
      double R[9], rt[3];
      quat_to_R(q + iq + 3, R); /* quat * vec3: Eigen rotates through the rotation matrix of quat */
      matvec3(R, trans, rt);
      for (int k = 0; k < 3; ++k) qout[iq + k] = rt[k] + q[iq + k]; /* :796 */
      double res[4];
      quat_mul(q + iq + 3, e3, res);                                /* :799 */
      const double dotp = res[0] * q[iq + 3] + res[1] * q[iq + 4] + res[2] * q[iq + 5] + res[3] * q[iq + 6];
      if (dotp < 0) /* :801-807 */
        for (int k = 0; k < 4; ++k) res[k] = -res[k];
      quat_first_order_normalize(res); /* :812 */
      for (int k = 0; k < 4; ++k) qout[iq + 3 + k] = res[k];
    } else if (jt == BRBD_JOINT_SPHERICAL) { /* special-orthogonal.hpp:495-510 */
      double e3[4], res[4];
      quat_exp3(v + iv, e3);
      quat_mul(q + iq, e3, res);
      quat_first_order_normalize(res);
      for (int k = 0; k < 4; ++k) qout[iq + k] = res[k];
    } else if ((jt >= BRBD_JOINT_RUBX && jt <= BRBD_JOINT_RUBZ) || jt == BRBD_JOINT_REVOLUTE_UNBOUNDED_UNALIGNED) {
      /* special-orthogonal.hpp:164-184 */
      const double ca = q[iq], sa = q[iq + 1];
      double so, co;
      sincos(v[iv], &so, &co);
      double o0 = co * ca - so * sa, o1 = so * ca + co * sa;
      const double norm2 = o0 * o0 + o1 * o1;
      const double f = (3 - norm2) / 2;
      qout[iq] = o0 * f;
      qout[iq + 1] = o1 * f;
    } else { /* vector space: liegroup/vector-space.hpp integrate_impl: q + v */
      qout[iq] = q[iq] + v[iv];
    }
  }
  return 0;
}

/* ================================================================== computeABADerivatives */
/* algorithm/aba-derivatives.hxx:381-454.  The reference computes Minv and ddq (passes 1-3), the
 * partial derivatives of RNEA at that acceleration (pass 4), symmetrises Minv (:449-450) and ends
 * with (:452-453)   ddq_dq = -Minv * dtau_dq,   ddq_dv = -Minv * dtau_dv,   ddq_dtau = Minv.
 * Restated with the building blocks above: aba (WORLD), computeRNEADerivatives at a = ddq,
 * computeMinverse.  All matrices nv x nv column-major; ddq may be NULL. */
int brbd_oracle_aba_derivatives(const brbd_model_desc* m, const double* q, const double* v, const double* tau,
                                double* ddq_dq, double* ddq_dv, double* Minv, double* ddq_out) {
  if (check_model(m) || !q || !v || !tau || !ddq_dq || !ddq_dv || !Minv) return -1;
  const int nv = m->nv;
  const size_t nn = (size_t)nv * nv;
  double* ddq = (double*)calloc((size_t)nv + 1, sizeof(double));
  double* dq = (double*)calloc(nn + 1, sizeof(double));
  double* dv = (double*)calloc(nn + 1, sizeof(double));
  int rc = brbd_oracle_aba_world(m, q, v, tau, ddq);
  if (!rc) rc = brbd_oracle_rnea_derivatives(m, q, v, ddq, dq, dv, NULL, NULL);
  if (!rc) rc = brbd_oracle_minverse(m, q, Minv);
  if (!rc) {
    for (int r = 0; r < nv; ++r) /* :449-450 */
      for (int c = 0; c < r; ++c) Minv[(size_t)c * nv + r] = Minv[(size_t)r * nv + c];
    for (int c = 0; c < nv; ++c) /* :452-453 */
      for (int r = 0; r < nv; ++r) {
        double s1 = 0, s2 = 0;
        for (int k = 0; k < nv; ++k) {
          s1 += Minv[(size_t)k * nv + r] * dq[(size_t)c * nv + k];
          s2 += Minv[(size_t)k * nv + r] * dv[(size_t)c * nv + k];
        }
        ddq_dq[(size_t)c * nv + r] = -s1;
        ddq_dv[(size_t)c * nv + r] = -s2;
      }
    if (ddq_out) memcpy(ddq_out, ddq, (size_t)nv * sizeof(double));
  }
  free(ddq);
  free(dq);
  free(dv);
  return rc;
}

int brbd_oracle_max_threads(void) {
#ifdef _OPENMP
  /* the batch loops call omp_set_num_threads() like the reference
   * (utils/openmp.hpp:31-35), which changes omp_get_max_threads(): remember the
   * value the process started with (OMP_NUM_THREADS or the core count) */
  static int initial = 0;
  if (initial == 0) initial = omp_get_max_threads();
  return initial;
#else
  return 1;
#endif
}

static int resolve_threads(int num_threads) {
  int mx = brbd_oracle_max_threads();
  if (num_threads <= 0 || num_threads > mx) num_threads = mx;
  return num_threads;
}

/* algorithm/parallel/rnea.hpp:70-82: setDefaultOpenMPSettings + static schedule,
 * one scratch replica per thread (multibody/pool/model.hpp:41-46) */
int brbd_oracle_rnea_batch(const brbd_model_desc* m, int64_t B, const double* q, const double* v,
                           const double* a, double* tau, int num_threads) {
  if (check_model(m)) return -1;
  const int nt = resolve_threads(num_threads);
  ws_t** pool = (ws_t**)calloc((size_t)nt, sizeof(ws_t*));
  for (int t = 0; t < nt; ++t) pool[t] = ws_alloc(m);
#ifdef _OPENMP
  omp_set_num_threads(nt);
  omp_set_dynamic(0);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < B; ++i) {
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
    rnea_ws(m, pool[tid], q + i * m->nq, v + i * m->nv, a + i * m->nv, NULL, tau + i * m->nv);
  }
  for (int t = 0; t < nt; ++t) ws_free(pool[t]);
  free(pool);
  return nt;
}

/* algorithm/parallel/aba.hpp:71-83 */
int brbd_oracle_aba_batch(const brbd_model_desc* m, int64_t B, const double* q, const double* v,
                          const double* tau, double* ddq, int num_threads) {
  if (check_model(m)) return -1;
  const int nt = resolve_threads(num_threads);
  ws_t** pool = (ws_t**)calloc((size_t)nt, sizeof(ws_t*));
  for (int t = 0; t < nt; ++t) pool[t] = ws_alloc(m);
  int bad = 0;
#ifdef _OPENMP
  omp_set_num_threads(nt);
  omp_set_dynamic(0);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < B; ++i) {
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
    if (aba_world_ws(m, pool[tid], q + i * m->nq, v + i * m->nv, tau + i * m->nv, NULL, ddq + i * m->nv)) {
#pragma omp atomic write
      bad = 1;
    }
  }
  for (int t = 0; t < nt; ++t) ws_free(pool[t]);
  free(pool);
  return bad ? -2 : nt;
}

int brbd_oracle_fk_jacobians_batch(const brbd_model_desc* m, int64_t B, const double* q, double* J,
                                   double* oMi, int num_threads) {
  if (check_model(m)) return -1;
  const int nt = resolve_threads(num_threads);
  ws_t** pool = (ws_t**)calloc((size_t)nt, sizeof(ws_t*));
  for (int t = 0; t < nt; ++t) pool[t] = ws_alloc(m);
#ifdef _OPENMP
  omp_set_num_threads(nt);
  omp_set_dynamic(0);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < B; ++i) {
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
    double* Ji = J ? J + i * 6 * (int64_t)m->nv : NULL;
    if (Ji) memset(Ji, 0, sizeof(double) * 6 * (size_t)m->nv);
    fkjac_ws(m, pool[tid], q + i * m->nq, Ji, oMi ? oMi + i * 12 * (int64_t)m->njoints : NULL);
  }
  for (int t = 0; t < nt; ++t) ws_free(pool[t]);
  free(pool);
  return nt;
}

int brbd_oracle_rnea_derivatives_batch(const brbd_model_desc* m, int64_t B, const double* q,
                                       const double* v, const double* a, double* dq, double* dv,
                                       double* da, double* tau, int num_threads) {
  if (check_model(m)) return -1;
  const int nt = resolve_threads(num_threads);
  const int64_t nn = (int64_t)m->nv * m->nv;
  ws_t** pool = (ws_t**)calloc((size_t)nt, sizeof(ws_t*));
  double** datmp = (double**)calloc((size_t)nt, sizeof(double*));
  for (int t = 0; t < nt; ++t) {
    pool[t] = ws_alloc(m);
    datmp[t] = (double*)malloc(sizeof(double) * (size_t)nn + 8);
  }
#ifdef _OPENMP
  omp_set_num_threads(nt);
  omp_set_dynamic(0);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < B; ++i) {
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
    rnea_derivatives_ws(m, pool[tid], q + i * m->nq, v + i * m->nv, a + i * m->nv, dq + i * nn,
                        dv + i * nn, da ? da + i * nn : datmp[tid], tau ? tau + i * m->nv : NULL);
  }
  for (int t = 0; t < nt; ++t) {
    ws_free(pool[t]);
    free(datmp[t]);
  }
  free(pool);
  free(datmp);
  return nt;
}

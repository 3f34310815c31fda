// brbd_spec.hpp -- kernel shells of the MODEL-COMPILED kernels.
//
// pinocchio_amd/codegen turns one constant model into straight-line bodies (gen/*.inc): the
// depth-first recursion of RNEA / ABA / FK+Jacobians fully unrolled, every structural zero of the
// model folded away, one C statement per surviving fp64 instruction.  This header gives those
// bodies their environment on gfx950:
//   * CT(k): the kernel's packed constant table, read through the CONSTANT address space with
//     compile-time offsets -> s_load_dwordx2..x16 into SGPRs, consumed directly by fp64 VALU ops;
//   * IN_* / OUT*: the boundary matrices in either layout of include/brbd.h (Mat<LAYOUT>);
//   * PK_LDS_* / PK_GLB_*: the park areas the generator's storage plan assigned: LDS
//     ([pair][lane] double2 -> ds_read/write_b128) and a wave-private global workspace
//     ([pair][lane] double2 -> global_load/store_dwordx4, coalesced 1 KiB rows);
//   * VISIT_END(): scheduling barrier between visits, so that the early loads the generator
//     placed one visit ahead stay there and register pressure stays per-visit.
// One lane = one instance, one wave = one workgroup, grid-stride over tiles of 64 instances.
#pragma once
#include "brbd_kernels.hpp"

namespace brbd {

typedef void (*SpecFn)(const KArgs);

struct SpecKernel {
  SpecFn fn[2][4];        // [layout][variant]; variant: fkjac output mask (1 = J, 2 = oMi), else 0
  const int32_t* gather;  // nconst x (joint, field, index): how to pack the constant table
  int32_t nconst;
  int32_t lds_doubles;    // per lane
  int32_t glb_doubles;    // per lane
};
enum : int { SF_R = 0, SF_T, SF_M, SF_C, SF_I, SF_ARM, SF_AXIS, SF_G };

struct SpecEntry {
  const char* name;
  const char* signature;  // tree + joint kinds + 0/1/-1 pattern of the constants (codegen/joints.py)
  SpecKernel algo[3];     // BRBD_ALGO_RNEA, BRBD_ALGO_ABA, BRBD_ALGO_FK_JAC
};

// ---- boundary matrices of the model-compiled kernels ------------------------------------------
// BATCH_MAJOR: one buffer resource per matrix and tile (base = first instance of the tile), element
// (d, lane) at byte d*ld8 + lane8: `buffer_load_dwordx2 v, v_lane8, s[rsrc], s_off offen` with
// s_off = d * ld8 one s_mul_i32 away -- two issue slots per element and no 64-bit address math.
// The host only selects these kernels when dof * ld * 8 < 2^32 for every matrix.
// Lanes past the end of the batch are clamped onto the last instance (same values re-stored).
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <int LAYOUT>
struct SpecMat;
template <>
struct SpecMat<1> {
  __amdgpu_buffer_rsrc_t r;
  unsigned lane8;
  DI SpecMat(const double* base, int64_t ld, int64_t tile, int64_t B, int lane) {
    (void)ld;
    r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base + tile * WAVE), 0, 0xFFFFFFFFu, 0x00020000);
    const int64_t left = B - tile * WAVE;  // >= 1
    lane8 = ((int64_t)lane < left ? (unsigned)lane : (unsigned)(left - 1)) * 8u;
  }
  DI double get(int d, unsigned ld8) const {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane8, (unsigned)d * ld8, 0));
  }
  DI void put(int d, unsigned ld8, double x) const {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x), r, lane8, (unsigned)d * ld8, 0);
  }
};
template <>
struct SpecMat<0> {  // INSTANCE_MAJOR: per-lane row pointer, element index is an immediate offset
  double* row;
  DI SpecMat(const double* base, int64_t ld, int64_t tile, int64_t B, int lane) {
    int64_t b = tile * WAVE + lane;
    if (b >= B) b = B - 1;
    row = const_cast<double*>(base) + b * ld;
  }
  DI double get(int d, unsigned) const { return row[d]; }
  DI void put(int d, unsigned, double x) const { row[d] = x; }
};

// sin and cos of a joint angle in ~40 instructions: 3-term Cody-Waite reduction by pi/2 carried by
// FMAs (exact products), fdlibm's minimax kernels on [-pi/4, pi/4] (k_sin.c / k_cos.c coefficients),
// quadrant fix-up.  Max abs error 2.3e-16 against glibc for |x| < 1e9 (tools/ubench/sincos_check.c);
// larger arguments, inf and nan take the device library's path.  (reference: math/sincos.hpp:47-61
// calls glibc sincos.)
DI void spec_sincos(double x, double& so, double& co) {
  if (__builtin_expect(!(__builtin_fabs(x) < 1.0e9), 0)) {
    sincos(x, &so, &co);
    return;
  }
  const double n = __builtin_rint(x * 0.63661977236758138);
  double r = __builtin_fma(-n, 1.5707963267948966, x);
  r = __builtin_fma(-n, 6.123233995736766e-17, r);
  r = __builtin_fma(-n, -1.4973849048591698e-33, r);
  const double z = r * r;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  ps = __builtin_fma(r * z, ps, r);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  pc = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
  const int q = (int)n;
  const double s = (q & 1) ? pc : ps, c = (q & 1) ? ps : pc;
  so = (q & 2) ? -s : s;
  co = ((q + 1) & 2) ? -c : c;
}

// ---- the macros the generated bodies use (codegen/emit.py: MACROS) ----
#define CT(k) (ct[k])
#define IN_Q(k) (mq.get(k, ldq8))
#define IN_V(k) (mv.get(k, ldv8))
#define IN_X(k) (mx.get(k, ldx8))
#define OUT(k, val) do { if (WANT & 1) mo.put(k, ldo8, val); } while (0)
#define OUT2(k, val) do { if (WANT & 2) mo2.put(k, ldo28, val); } while (0)
#define SINCOS(a, s, c) spec_sincos(a, s, c)
#define RSQRT(a) rsqrt(a)
#define RCP(a) (1.0 / (a))
#define FMA(a, b, c) __builtin_fma(a, b, c)
#define PK_LDS_ST2(p, a, b) lds2[(p) * WAVE] = make_double2(a, b)
#define PK_LDS_LD2(p, a, b) do { const double2 t_ = lds2[(p) * WAVE]; a = t_.x; b = t_.y; } while (0)
#define PK_GLB_ST2(p, a, b) \
  *reinterpret_cast<double2*>(reinterpret_cast<char*>(ws2 + (p) * WAVE) + lane16) = make_double2(a, b)
#define PK_GLB_LD2(p, a, b)                                                                              \
  do {                                                                                                   \
    const double2 t_ = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(ws2 + (p) * WAVE) + lane16); \
    a = t_.x; b = t_.y;                                                                                  \
  } while (0)
// Visit boundary: nothing is scheduled across it, and the uniform values every scalar load / offset
// is derived from are made opaque again, so neither LICM (the tile loop) nor the scheduler can hoist
// the table loads of later visits up front (that is what cost 2000 SGPR spills in the template-
// unrolled experiment).
#define VISIT_END()                                                                                  \
  do {                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    asm volatile("" : "+s"(ct), "+s"(ldq8), "+s"(ldv8), "+s"(ldx8), "+s"(ldo8), "+s"(ldo28));        \
  } while (0)

// everything before the tile loop
#define BRBD_SPEC_PROLOGUE(LDS_PAIRS)                                                                \
  const int lane = threadIdx.x;                                                                      \
  const double BRBD_CONSTANT* ct = as_constant(A.ctab);                                              \
  __shared__ double2 lds_area[(LDS_PAIRS) > 0 ? (LDS_PAIRS) * WAVE : 1];                             \
  double2* const lds2 = lds_area + lane;                                                             \
  double2* const ws2 = reinterpret_cast<double2*>(A.ws + (int64_t)blockIdx.x * A.ws_stride);         \
  const unsigned lane16 = (unsigned)lane * 16u;                                                      \
  unsigned ldq8 = (unsigned)A.ldq * 8u, ldv8 = (unsigned)A.ldv * 8u, ldx8 = (unsigned)A.lda * 8u;    \
  unsigned ldo8 = (unsigned)A.ldout * 8u, ldo28 = (unsigned)A.ldout2 * 8u;                           \
  (void)lds2; (void)ws2; (void)lane16;

#define BRBD_SPEC_TILE(LAYOUT)                                                                       \
  const SpecMat<LAYOUT> mq(A.q, A.ldq, tile, A.B, lane);                                             \
  const SpecMat<LAYOUT> mv(A.v ? A.v : A.q, A.ldv, tile, A.B, lane);                                 \
  const SpecMat<LAYOUT> mx(A.a ? A.a : A.q, A.lda, tile, A.B, lane);                                 \
  const SpecMat<LAYOUT> mo(A.out ? A.out : A.out2, A.ldout, tile, A.B, lane);                        \
  const SpecMat<LAYOUT> mo2(A.out2 ? A.out2 : A.out, A.ldout2, tile, A.B, lane);                     \
  asm volatile("" : "+s"(ct), "+s"(ldq8), "+s"(ldv8), "+s"(ldx8), "+s"(ldo8), "+s"(ldo28));          \
  (void)mv; (void)mx; (void)mo; (void)mo2;

}  // namespace brbd

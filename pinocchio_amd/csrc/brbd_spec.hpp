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
//     ([pair][lane] double2 -> buffer_load/store_dwordx4, coalesced 1 KiB rows);
//   * VISIT_END(): scheduling barrier between visits, so that the early loads the generator
//     placed one visit ahead stay there and register pressure stays per-visit.
// One lane = one instance, one wave = one workgroup = one tile of 64 instances; the host launches
// at most `chunk` tiles at a time so that the wave-private global park area stays bounded.
#pragma once
#include "brbd_kernels.hpp"

namespace brbd {

typedef void (*SpecFn)(const KArgs);

struct SpecKernel {
  // [layout][variant]; layout 0 / 1 as in include/brbd.h, 2 = INSTANCE_MAJOR with tight leading
  // dimensions, staged through LDS (RNEA / ABA only); variant: fkjac output mask (1 = J, 2 = oMi),
  // else index into codegen/build.py: WANTS
  SpecFn fn[3][4];
  const int32_t* gather;  // nconst x 4 ints: how to pack the constant table (brbd_capi.hip: pack_constants)
  int32_t nconst;
  int32_t lds_doubles;    // per lane
  int32_t glb_doubles;    // per lane
  int32_t stage_doubles;  // per lane: global staging rows of the layout-2 kernels (0 = staged in registers)
};
enum : int { SF_R = 0, SF_T, SF_M, SF_C, SF_I, SF_ARM, SF_AXIS, SF_G };

// Layout version of KArgs / SpecKernel / SpecEntry: kernels compiled separately (plugins built by
// pinocchio_amd/codegen/plugin.py for models unknown when the library was built, loaded with
// brbd_load_compiled_kernels) must agree with the library that launches them.
#define BRBD_SPEC_ABI 7

struct SpecEntry {
  const char* name;
  const char* signature;  // tree + joint kinds + 0/1/-1 pattern of the constants (codegen/joints.py)
  SpecKernel algo[7];     // BRBD_ALGO_RNEA, _ABA, _FK_JAC, _RNEA_DERIV, _MINVERSE, _ABA_DERIV_PRODUCTS, _JOINT_JAC
};

// ---- memory access of the model-compiled kernels ------------------------------------------------
// Every vector memory access goes through a buffer resource with an exact byte range
// (num_records): `buffer_load/store v, v_off, s[rsrc], s_off offen offset:imm`.  Two issue slots
// per element at most (an s_mul_i32 for s_off), no 64-bit address arithmetic, and an offset that
// is out of range is dropped by the hardware instead of faulting.  The host only selects these
// kernels when every matrix spans less than 4 GiB (32-bit byte offsets).
// Lanes past the end of the batch are clamped onto the last instance (same values re-stored).
// Cache policy of the boundary matrices (aux operand of the buffer instructions; 2 = nt).  Inputs
// are read once and outputs written once: in BATCH_MAJOR every access is a full 512-byte row
// segment, and streaming it past the L2 is worth 13 % on the headline (RNEA 18.9 -> 16.4 us, ABA
// 44.9 -> 40.1 us) and 20 % on the output-bound kernels.  INSTANCE_MAJOR keeps the default policy:
// its lanes touch one cache line each and live on L2 hits (nt there: 1.08e9 -> 0.43e9 evals/s).
// The kernel template sets SPEC_IN_AUX / SPEC_OUT_AUX (codegen/build.py: nt loads slow the
// output-bound derivative and FK + Jacobian kernels down, so they stream only their stores).
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
DI __amdgpu_buffer_rsrc_t spec_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
// Checking build (make check -> libbrbd_hip_check.so, -DBRBD_SPEC_CHECK): the hardware drops an
// access whose offset is outside the buffer's byte range, which would HIDE an indexing bug.  In
// this build every access of a compiled kernel is held against the range; violations are counted
// in the handle's counter (brbd_debug_oob_count).  tests/test_gpu_bounds_check.py runs the GPU
// suite's kernels once under it, and provokes violations on purpose to see them counted.
//
// How: every access is (per-lane offset fixed at construction) + (wave-uniform offset of the site),
// and the site's part is a LITERAL of the generated body (a row number, a park slot).  An accessor
// therefore only notes the largest literal any of its sites uses -- the compiler folds that to a
// constant -- and compares it ONCE, when it goes out of scope at the end of the tile, with the
// largest one that fits: the range minus the access size minus the maximum of the lane offsets
// over the wave (one reduction per accessor), divided by the leading dimension where the literal
// is a row of a BATCH_MAJOR matrix.  The instrumented kernels are the product kernels plus a few
// dozen instructions.  History (round 3): the first version compared at every site and called an
// out-of-line reporter; that put a stack frame under kernels that already spill SGPRs, the
// combination hipcc 7.2 miscompiles -- the instrumented talos_like k_aba<1,33> returned NaN while
// the product kernel was right.  Per-site compares into a per-lane record pushed the same kernel
// into scratch; per-site scalar compares spilled thousands of SGPRs (LLVM reassociates the counter).
// A site inside a conditional block counts whether or not it executes: stricter, and still exact
// for the straight-line bodies.
#ifdef BRBD_SPEC_CHECK
struct SpecBad {
  unsigned n, kinds;     // wave-uniform: violating accessors, their kinds (bit mask) ...
  int off, limit;        // ... and the last one: its largest literal, and the largest that fits
  int ws_seen, ws_limit; // the wave-private park area (PK_GLB_*)
};
typedef SpecBad* SpecOob;
DI unsigned spec_wave_max(unsigned x) {
#pragma unroll
  for (int s = 1; s < WAVE; s <<= 1) {
    const unsigned y = (unsigned)__shfl_xor((int)x, s, WAVE);
    x = x > y ? x : y;
  }
  return (unsigned)__builtin_amdgcn_readfirstlane((int)x);
}
// Largest uniform offset a site may add to the lane offsets: INT_MAX = not checked (bytes == 0: an
// absent optional matrix, dropped by design), -1 = the range cannot even hold offset 0 of every
// lane (any site at all is then a violation; an accessor without sites has seen == -1 and passes).
DI int spec_slack(unsigned voff, unsigned size, unsigned bytes) {
  const unsigned need = spec_wave_max(voff) + size;
  if (bytes == 0u) return 0x7fffffff;
  if (need > bytes || need < size) return -1;
  const unsigned slack = bytes - need;
  return slack > 0x7fffffffu ? 0x7fffffff : (int)slack;
}
// (BATCH_MAJOR: a site adds (row literal) x (leading dimension) -- the last row that fits)
DI int spec_rows(int slack, unsigned ld8) {
  return (slack < 0 || slack == 0x7fffffff || ld8 == 0u) ? slack
                                                          : __builtin_amdgcn_readfirstlane((int)((unsigned)slack / ld8));
}
DI void spec_check(SpecBad* s, int seen, int limit, unsigned what) {
  const bool bad = seen > limit;
  s->n += bad ? 1u : 0u;
  s->kinds |= bad ? 1u << what : 0u;
  s->off = bad ? seen : s->off;
  s->limit = bad ? limit : s->limit;
}
struct SpecReport {  // declared before the accessors, so destroyed after them
  unsigned long long* oob;
  SpecBad* s;
  DI ~SpecReport() {
    spec_check(s, s->ws_seen, s->ws_limit, 5);
    if (!oob || s->n == 0u || threadIdx.x != 0) return;
    atomicAdd(oob, (unsigned long long)s->n);
    oob[1] = s->kinds;
    oob[2] = (unsigned long long)(long long)s->off;
    oob[3] = (unsigned long long)(long long)s->limit;
  }
};
#define SPEC_SLACK(oob, voff, size, bytes, what) spec_slack(voff, size, bytes)
#define SPEC_ROWS(slack, ld8) spec_rows(slack, ld8)
#define SPEC_SEEN mutable int seen = -1;
#define SPEC_NOTE(x) (seen = seen > (int)(x) ? seen : (int)(x))
#define SPEC_FINAL(T, limit, what) DI ~T() { spec_check(oob, seen, limit, what); }
#define SPEC_NOTE_WS(x) (spec_bad.ws_seen = spec_bad.ws_seen > (int)(x) ? spec_bad.ws_seen : (int)(x))
#define SPEC_OOB (&spec_bad)
#define SPEC_OOB_DECL SpecBad spec_bad = {0u, 0u, 0, 0, -1, 0x7fffffff}; const SpecReport spec_report{A.oob, &spec_bad}
#define SPEC_WS_LIMIT(wsoff, bytes) (spec_bad.ws_limit = spec_slack(wsoff, 16u, bytes))
#else
typedef unsigned long long* SpecOob;
#define SPEC_SLACK(oob, voff, size, bytes, what) 0
#define SPEC_ROWS(slack, ld8) 0
#define SPEC_SEEN
#define SPEC_NOTE(x) do { } while (0)
#define SPEC_FINAL(T, limit, what)
#define SPEC_NOTE_WS(x) do { } while (0)
#define SPEC_OOB A.oob
#define SPEC_OOB_DECL int spec_bad = 0
#define SPEC_WS_LIMIT(wsoff, bytes) do { } while (0)
#endif
template <int LAYOUT, int AUX>
struct SpecMat;
template <int AUX>
struct SpecMat<1, AUX> {  // BATCH_MAJOR: element (d, b) at byte d * ld8 + b * 8
  __amdgpu_buffer_rsrc_t r;
  unsigned voff;
  int dmax;  // (checking build: the last row whose elements are in range for every lane)
  SpecOob oob;
  SPEC_SEEN
  SPEC_FINAL(SpecMat, dmax, 1)
  DI SpecMat(const double* base, unsigned bytes, unsigned ld8, int64_t b, SpecOob oob_ = nullptr)
      : r(spec_rsrc(base, bytes)), oob(oob_) {
    voff = (unsigned)b * 8u;
    dmax = SPEC_ROWS(SPEC_SLACK(oob, voff, 8u, bytes, 1), ld8);
  }
  DI double get(int d, unsigned ld8) const {
    SPEC_NOTE(d);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, (unsigned)d * ld8, AUX));
  }
  DI void put(int d, unsigned ld8, double x) const {
    SPEC_NOTE(d);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x), r, voff, (unsigned)d * ld8, AUX);
  }
};
template <int AUX>
struct SpecMat<0, AUX> {  // INSTANCE_MAJOR: element (d, b) at byte b * ld8 + d * 8 (immediate offset)
  __amdgpu_buffer_rsrc_t r;
  unsigned voff;
  int slack;  // (checking build)
  SpecOob oob;
  SPEC_SEEN
  SPEC_FINAL(SpecMat, slack, 3)
  DI SpecMat(const double* base, unsigned bytes, unsigned ld8, int64_t b, SpecOob oob_ = nullptr)
      : r(spec_rsrc(base, bytes)), oob(oob_) {
    voff = (unsigned)b * ld8;
    slack = SPEC_SLACK(oob, voff, 8u, bytes, 3);
  }
  DI double get(int d, unsigned) const {
    SPEC_NOTE(d * 8);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff + (unsigned)d * 8u, 0, 0));
  }
  DI void put(int d, unsigned, double x) const {
    SPEC_NOTE(d * 8);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x), r, voff + (unsigned)d * 8u, 0, 0);
  }
};
DI void spec_park_st(__amdgpu_buffer_rsrc_t r, unsigned voff, int p, double a, double b) {
  const v2u x = __builtin_bit_cast(v2u, a), y = __builtin_bit_cast(v2u, b);
  const v4u w = {x.x, x.y, y.x, y.y};
  __builtin_amdgcn_raw_buffer_store_b128(w, r, voff + (unsigned)p * (WAVE * 16u), 0, 0);
}
DI void spec_park_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, int p, double& a, double& b) {
  const v4u w = __builtin_amdgcn_raw_buffer_load_b128(r, voff + (unsigned)p * (WAVE * 16u), 0, 0);
  const v2u x = {w.x, w.y}, y = {w.z, w.w};
  a = __builtin_bit_cast(double, x);
  b = __builtin_bit_cast(double, y);
}

// ---- INSTANCE_MAJOR with LDS-staged transposition (kernel layout 2) ---------------------------------
// The reference's own layout (Eigen column-major, one instance per contiguous column,
// algorithm/parallel/rnea.hpp:39-45) is the worst case for one lane per instance: a per-lane 8-byte
// load touches one cache line per lane, and each line is touched 16 times over the program (round 1
// measured 4.9x the input bytes fetched and the waves waiting on the texture addresser 3/4 of the
// time).  With a tight leading dimension the 64 columns of a tile are ONE contiguous run of 64 x N
// doubles: the wave reads it with N fully coalesced 512-byte loads, transposes it through LDS (the
// park area is not in use yet) and every lane ends up with its own instance's N values -- in
// registers (MODE 0: the kernel has the room, e.g. RNEA) or in N coalesced rows of the wave's
// private global staging area (MODE 1: the body then reads them like a BATCH_MAJOR matrix).  The
// output goes the other way at the end of the kernel.  The host selects layout 2 when every staged
// matrix of the call has ld == rows; anything else runs the per-lane kernels (layout 0).
// One wave per workgroup: its LDS operations execute in issue order, so a transposition needs no
// hardware barrier between its write and read phases, only the compiler kept from reordering them.
// (BRBD_SPEC_LDS_BARRIER: the __syncthreads() of rounds 1-2, which also waits for every load in flight.)
#ifdef BRBD_SPEC_LDS_BARRIER
#define SPEC_LDS_ORDER() __syncthreads()
#else
#define SPEC_LDS_ORDER() __builtin_amdgcn_wave_barrier()
#endif
struct SpecStage {
  int64_t tile;
  int lane;
  double* lds;   // >= 64 * N doubles of LDS, free at the start and at the end of the kernel
  double* rows;  // this wave's global staging rows ([row][64 lanes]), MODE 1
  unsigned rows_bytes;
  double* win;   // MODE 3: LDS window of SPEC_WIN_ROWS rows of SPEC_WIN_STRIDE doubles, behind the park area
};
constexpr int SPEC_WIN_ROWS = 8, SPEC_WIN_STRIDE = 68;
// Two phases, so that the loads of ALL staged matrices are in flight before the first one is waited
// for (one exposed HBM latency per kernel instead of one per matrix): issue() = the N coalesced
// loads of the tile's flat run; commit() = through LDS, each lane receives its own N values.
template <int N, int AUX>
DI void spec_stage_issue(const double* base, unsigned bytes, const SpecStage& sc, double* x) {
  const __amdgpu_buffer_rsrc_t r = spec_rsrc(base, bytes);
  // flat run of the tile: element j * 64 + lane; past the end of the batch the range check returns 0
  const unsigned t0 = (unsigned)sc.tile * (unsigned)(WAVE * N * 8) + (unsigned)sc.lane * 8u;
#pragma unroll
  for (int j = 0; j < N; ++j)
    x[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, t0, (unsigned)(j * WAVE * 8), AUX));
}
template <int N, typename Sink>
DI void spec_stage_commit(const SpecStage& sc, const double* x, Sink&& sink) {
  SPEC_LDS_ORDER();
#pragma unroll
  for (int j = 0; j < N; ++j) sc.lds[j * WAVE + sc.lane] = x[j];
  SPEC_LDS_ORDER();
#pragma unroll
  for (int d = 0; d < N; ++d) sink(d, sc.lds[sc.lane * N + d]);
}
// The same with 16 bytes per lane per instruction: pair p = j * 64 + lane of the tile's flat run =
// elements 2p, 2p + 1 -- half the load and LDS-write instructions (x2 / x4 buffer accesses are
// range-checked per dword, so a pair that straddles the end of the matrix loses only its far half).
// N odd: the last instruction's upper 32 lanes lie past the tile and take no part.  MODE 4 of the
// staged inputs; measured per kernel (humanoid, 65 536: RNEA 17.3 -> 16.4 us, ABA 25.1 -> 26.3 us, so
// RNEA uses it and ABA does not: codegen/build.py, STAGE_MODE).
template <int N>
constexpr int spec_stage_regs() { return 2 * ((N + 1) / 2); }
template <int N>
DI bool spec_stage_lane_on(int j, int lane) { return N % 2 == 0 || j < N / 2 || lane < WAVE / 2; }
template <int N, int AUX>
DI void spec_stage_issue_wide(const double* base, unsigned bytes, const SpecStage& sc, double* x) {
  const __amdgpu_buffer_rsrc_t r = spec_rsrc(base, bytes);
  const unsigned t0 = (unsigned)sc.tile * (unsigned)(WAVE * N * 8) + (unsigned)sc.lane * 16u;
#pragma unroll
  for (int j = 0; j < (N + 1) / 2; ++j) {
    const v4u w = __builtin_amdgcn_raw_buffer_load_b128(r, t0, (unsigned)(j * WAVE * 16), AUX);
    const v2u lo = {w.x, w.y}, hi = {w.z, w.w};
    x[2 * j] = __builtin_bit_cast(double, lo);
    x[2 * j + 1] = __builtin_bit_cast(double, hi);
  }
}
template <int N, typename Sink>
DI void spec_stage_commit_wide(const SpecStage& sc, const double* x, Sink&& sink) {
  double2* const l2 = reinterpret_cast<double2*>(sc.lds);
  SPEC_LDS_ORDER();
#pragma unroll
  for (int j = 0; j < (N + 1) / 2; ++j)
    if (spec_stage_lane_on<N>(j, sc.lane)) l2[j * WAVE + sc.lane] = make_double2(x[2 * j], x[2 * j + 1]);
  SPEC_LDS_ORDER();
#pragma unroll
  for (int d = 0; d < N; ++d) sink(d, sc.lds[sc.lane * N + d]);
}
template <int LAYOUT, int AUX, int N, int MODE>
struct SpecIn : SpecMat<LAYOUT, AUX> {
  DI SpecIn(const double* base, unsigned bytes, unsigned ld8, int64_t b, SpecOob oob_, const SpecStage&, int)
      : SpecMat<LAYOUT, AUX>(base, bytes, ld8, b, oob_) {}
  DI void commit(const SpecStage&) {}
  template <typename G> DI void grp(int, G&) const {}
  template <typename G> DI void grpc(int, G&) const {}
  template <typename G> DI double get2(int d, unsigned ld8, const G&, int) const { return this->get(d, ld8); }
};
template <int AUX, int N>
struct SpecIn<2, AUX, N, 0> {  // staged into registers
  double r[N > 0 ? N : 1];    // (issue: the lane's share of the flat run; after commit: its own row)
  DI SpecIn(const double* base, unsigned bytes, unsigned, int64_t, SpecOob, const SpecStage& sc, int) {
    spec_stage_issue<N, AUX>(base, bytes, sc, r);
  }
  DI void commit(const SpecStage& sc) {
    double x[N > 0 ? N : 1];
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = r[j];
    spec_stage_commit<N>(sc, x, [&](int d, double y) { r[d] = y; });
  }
  DI double get(int d, unsigned) const { return r[d]; }
  template <typename G> DI void grp(int, G&) const {}
  template <typename G> DI void grpc(int, G&) const {}
  template <typename G> DI double get2(int d, unsigned ld8, const G&, int) const { return get(d, ld8); }
};
template <int AUX, int N>
struct SpecIn<2, AUX, N, 4> {  // staged into registers, 16 bytes per lane per instruction
  static constexpr int NX = spec_stage_regs<N>();
  double r[NX > 0 ? NX : 1];  // (issue: the lane's share of the flat run; after commit: its own row)
  DI SpecIn(const double* base, unsigned bytes, unsigned, int64_t, SpecOob, const SpecStage& sc, int) {
    spec_stage_issue_wide<N, AUX>(base, bytes, sc, r);
  }
  DI void commit(const SpecStage& sc) {
    double x[NX > 0 ? NX : 1];
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] = r[j];
    spec_stage_commit_wide<N>(sc, x, [&](int d, double y) { r[d] = y; });
  }
  DI double get(int d, unsigned) const { return r[d]; }
  template <typename G> DI void grp(int, G&) const {}
  template <typename G> DI void grpc(int, G&) const {}
  template <typename G> DI double get2(int d, unsigned ld8, const G&, int) const { return get(d, ld8); }
};
template <int AUX, int N>
struct SpecIn<2, AUX, N, 1> {  // staged into the wave's global rows row0 .. row0 + N - 1
  __amdgpu_buffer_rsrc_t rr;
  unsigned voff;
  int slack, row0;
  SpecOob oob;
  SPEC_SEEN
  SPEC_FINAL(SpecIn, slack, 7)
  double pend[N > 0 ? N : 1];
  DI SpecIn(const double* base, unsigned bytes, unsigned, int64_t, SpecOob oob_, const SpecStage& sc, int row0_)
      : rr(spec_rsrc(sc.rows, sc.rows_bytes)), voff((unsigned)sc.lane * 8u), row0(row0_), oob(oob_) {
    slack = SPEC_SLACK(oob, voff, 8u, sc.rows_bytes, 7);
    spec_stage_issue<N, AUX>(base, bytes, sc, pend);
  }
  DI void commit(const SpecStage& sc) {
    spec_stage_commit<N>(sc, pend, [&](int d, double x) {
      SPEC_NOTE((row0 + d) * WAVE * 8);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x), rr, voff, (unsigned)((row0 + d) * WAVE * 8), 0);
    });
  }
  DI double get(int d, unsigned) const {
    SPEC_NOTE((row0 + d) * WAVE * 8);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rr, voff, (unsigned)((row0 + d) * WAVE * 8), 0));
  }
  template <typename G> DI void grp(int, G&) const {}
  template <typename G> DI void grpc(int, G&) const {}
  template <typename G> DI double get2(int d, unsigned ld8, const G&, int) const { return get(d, ld8); }
};
// MODE 2: no staging at all -- every lane fetches ITS OWN column a cache line at a time.  In the
// reference's layout elements 16 g .. 16 g + 15 of a lane's instance are 128 consecutive bytes: 8
// back-to-back b128 loads per lane (GRP_Q(g, n) in the generated body, codegen/emit.py:
// group_inputs) bring the line in once and consume it whole, when the walk first needs one of
// its elements, so the loads overlap the arithmetic as they do in BATCH_MAJOR instead of all
// being waited for before the first visit.  Default cache policy: the 8 loads of a group and the
// neighbouring lanes' groups share lines and meet in the L1 / L2.  Works for any leading dimension.
template <int KIND>  // 0: none, 2: a 128-byte line of the lane's own column, 3: 8 dofs of the tile (MODE 3)
struct SpecGrp {};
template <>
struct SpecGrp<2> {
  double2 v[8];
};
template <>
struct SpecGrp<3> {
  double v[8];
};
template <int AUX, int N>
struct SpecIn<2, AUX, N, 2> {
  __amdgpu_buffer_rsrc_t r;
  unsigned voff;
  int slack;
  SpecOob oob;
  SPEC_SEEN
  SPEC_FINAL(SpecIn, slack, 10)
  DI SpecIn(const double* base, unsigned bytes, unsigned ld8, int64_t b, SpecOob oob_, const SpecStage&, int)
      : r(spec_rsrc(base, bytes)), voff((unsigned)b * ld8), oob(oob_) {
    slack = SPEC_SLACK(oob, voff, 8u, bytes, 10);
  }
  DI void commit(const SpecStage&) {}
  DI void grp(int g, SpecGrp<2>& G) const {  // (g is a literal at every call site: the bounds fold)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (16 * g + 2 * i < N) {
        const unsigned off = voff + (unsigned)(128 * g + 16 * i);
        // (the last pair of an odd N reaches 8 bytes into the next column; past the very end of the
        // matrix the range check of the buffer returns zeros for that half)
        SPEC_NOTE(128 * g + 16 * i);
        const v4u w = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        const v2u x = {w.x, w.y}, y = {w.z, w.w};
        G.v[i] = make_double2(__builtin_bit_cast(double, x), __builtin_bit_cast(double, y));
      }
  }
  DI double get2(int, unsigned, const SpecGrp<2>& G, int j) const { return (j & 1) ? G.v[j >> 1].y : G.v[j >> 1].x; }
  template <typename G> DI void grpc(int, G&) const {}
};
// MODE 3: chunked staging.  Nothing is loaded up front.  GRP_Q(g, n) (placed a prefetch distance
// ahead of the visit that needs the group, codegen/emit.py: group_inputs) issues 8 loads that
// together cover dofs 8 g .. 8 g + 7 of all 64 instances of the tile: lane l takes dof 8 g + l % 8
// of instances 8 k + l / 8 (k = 0 .. 7), i.e. every load instruction reads 8 runs of 64 contiguous
// bytes -- coalesced enough for the texture addresser (8 - 16 lines per instruction against 64 for
// a per-lane column fetch), and in flight while the walk computes.  GRPC_Q(g, n), where the first
// element is consumed, passes the 8 x 64 block through the LDS window ([dof][instance], rows padded
// to 68 doubles so that the 32 lanes of a half-wave write 32 different banks) and every lane
// receives the 8 elements of its own instance.  The window lies behind the park area, which is in
// use all along: these bodies park at most 70 doubles per lane in LDS (codegen/compile.py:
// CHUNK_LDS).  Requires a tight leading dimension (the host selects layout 2 only then); a group
// that reaches past row N reads the first dofs of the next instance (or, at the very end of the
// matrix, the zeros of the range check), which nothing consumes.
// (default cache policy unless overridden: a 128-byte line is shared by two groups fetched at different times)
#ifndef SPEC_CHUNK_AUX
#define SPEC_CHUNK_AUX 0
#endif
template <int AUX, int N>
struct SpecIn<2, AUX, N, 3> {
  __amdgpu_buffer_rsrc_t r;
  unsigned voff;
  double* win;
  int lane;
  DI SpecIn(const double* base, unsigned bytes, unsigned, int64_t, SpecOob, const SpecStage& sc, int)
      : r(spec_rsrc(base, bytes)),
        voff((unsigned)sc.tile * (unsigned)(WAVE * N * 8) + (unsigned)(((sc.lane >> 3) * N + (sc.lane & 7)) * 8)),
        win(sc.win), lane(sc.lane) {}
  DI void commit(const SpecStage&) {}
  DI void grp(int g, SpecGrp<3>& G) const {  // (g is a literal at every call site)
#pragma unroll
    for (int k = 0; k < 8; ++k)
      G.v[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, (unsigned)((k * 8 * N + g * 8) * 8), SPEC_CHUNK_AUX));
  }
  DI void grpc(int, SpecGrp<3>& G) const {
    // No __syncthreads() here: it would also wait for every global load in flight (the next groups,
    // the parks), which is the overlap this mode exists for.  One wave per workgroup: its LDS
    // operations execute in issue order, so the reads below see the writes above, and the next
    // commit's writes come after these reads; the compiler keeps the order (it cannot tell the
    // addresses apart) and the wave barriers keep it from moving anything across.
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 8; ++k) win[(lane & 7) * SPEC_WIN_STRIDE + 8 * k + (lane >> 3)] = G.v[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int d = 0; d < 8; ++d) G.v[d] = win[d * SPEC_WIN_STRIDE + lane];
    __builtin_amdgcn_wave_barrier();
  }
  DI double get2(int, unsigned, const SpecGrp<3>& G, int j) const { return G.v[j]; }
};

template <int LAYOUT, int AUX, int N>
struct SpecOut : SpecMat<LAYOUT, AUX> {
  DI SpecOut(double* base, unsigned bytes, unsigned ld8, int64_t b, SpecOob oob_)
      : SpecMat<LAYOUT, AUX>(base, bytes, ld8, b, oob_) {}
  DI void flush(double*, unsigned, const SpecStage&) const {}
};
template <int AUX, int N>
struct SpecOut<2, AUX, N> {  // collected in registers, transposed through LDS and stored coalesced at the end
  mutable double r[N > 0 ? N : 1];
  DI SpecOut(double*, unsigned, unsigned, int64_t, SpecOob) {}
  DI void put(int d, unsigned, double x) const { r[d] = x; }
  DI void flush(double* base, unsigned bytes, const SpecStage& sc) const {
    const __amdgpu_buffer_rsrc_t rs = spec_rsrc(base, bytes);
    const unsigned t0 = (unsigned)sc.tile * (unsigned)(WAVE * N * 8) + (unsigned)sc.lane * 8u;
    SPEC_LDS_ORDER();
#pragma unroll
    for (int d = 0; d < N; ++d) sc.lds[sc.lane * N + d] = r[d];
    SPEC_LDS_ORDER();
#pragma unroll
    for (int j = 0; j < N; ++j)  // (stores past the end of the batch are dropped by the range check)
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, sc.lds[j * WAVE + sc.lane]), rs, t0,
                                            (unsigned)(j * WAVE * 8), AUX);
  }
};

// Outputs whose rows arrive in increasing order (FK + Jacobians, codegen/fkjac.py): CH rows are
// collected in registers and leave through LDS as soon as the chunk is complete -- element
// (row, instance) of the chunk goes to LDS as [instance][row], and 64 consecutive elements of that
// flat order are one store: runs of CH * 8 bytes per instance.  `d` is a literal at every call
// site, so the chunk bookkeeping folds away.
template <int LAYOUT, int AUX, int N, int STREAM>
struct SpecOutSel : SpecOut<LAYOUT, AUX, N> {
  DI SpecOutSel(double* base, unsigned bytes, unsigned ld8, int64_t b, SpecOob oob_, const SpecStage&)
      : SpecOut<LAYOUT, AUX, N>(base, bytes, ld8, b, oob_) {}
};
template <int AUX, int N>
struct SpecOutSel<2, AUX, N, 1> {
  static constexpr int CH = 64;
  mutable double r[CH];
  __amdgpu_buffer_rsrc_t rs;
  unsigned t0, nbytes;
  int lane;
  double* lds;
  SpecOob oob;
  DI SpecOutSel(double* base, unsigned bytes, unsigned, int64_t, SpecOob oob_, const SpecStage& sc)
      : rs(spec_rsrc(base, bytes)), t0((unsigned)sc.tile * (unsigned)(WAVE * N * 8)), nbytes(bytes), lane(sc.lane),
        lds(sc.lds), oob(oob_) {}
  DI void flush_chunk(int c0, int nr) const {  // (c0, nr: literals once put() is inlined)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CH; ++k)
      if (k < nr) lds[lane * nr + k] = r[k];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (j < nr) {
        const unsigned f = (unsigned)(j * WAVE + lane), inst = f / (unsigned)nr, row = f % (unsigned)nr;
        const unsigned off = t0 + inst * (unsigned)(N * 8) + (unsigned)(c0 * 8) + row * 8u;
        // (no check here: the range is tested explicitly, and offsets and range are multiples of 8)
        if (off + 8u <= nbytes)  // (instances past the end of the batch)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, lds[f]), rs, off, 0, AUX);
      }
  }
  DI void put(int d, unsigned, double x) const {
    r[d % CH] = x;
    const int c0 = d - d % CH;                  // first row of d's chunk
    const int nr = N - c0 < CH ? N - c0 : CH;   // rows of that chunk
    if (d == c0 + nr - 1) flush_chunk(c0, nr);  // rows arrive in increasing order: the chunk is complete
  }
  DI void flush(double*, unsigned, const SpecStage&) const {}
};
// defaults for the kernel templates that do not stream their outputs (shadowed in codegen/fkjac's)
constexpr int SPEC_NO2 = 0, SPEC_OUT_STREAM = 0;

// sin and cos of a joint angle in ~35 straight-line instructions: 2-term Cody-Waite reduction by
// pi/2 carried by FMAs (exact products; the third term, 1.5e-33 n, is below 1e-17 for the whole
// domain), fdlibm's minimax kernels on [-pi/4, pi/4] (k_sin.c / k_cos.c coefficients), quadrant
// fix-up on the integer bits.  n = rint(x 2/pi) comes from one FMA with 1.5 * 2^52: the sum's ulp
// is 1, so the FMA rounds x 2/pi to the nearest integer in one rounding and leaves n mod 2^32 (two's
// complement) in the low mantissa word -- no rint, no float -> int conversion.  Max abs error
// 2.3e-16 against glibc for |x| <= 1e12, 4.5e-16 at 1e15 (tools/ubench/sincos_check.c).
// Deliberately BRANCH-FREE: the compiled kernels stream through their code once, so every taken
// branch (e.g. around an inlined large-argument path) lands on an instruction-cache miss --
// measured at ~1000 cycles per joint, more than the joint's arithmetic.
// Valid for |x| < 2^50 rad; beyond that (and for inf / nan) the result is NaN, where the reference
// (math/sincos.hpp:47-61 -> glibc) returns the sine of the rounded double.  Models whose inputs can
// be that large run on the table-driven kernels (brbd_model_use_compiled(m, 0)).
DI void spec_sincos(double x, double& so, double& co) {
  const double t = __builtin_fma(x, 0.63661977236758138, 6755399441055744.0);
  const double n = t - 6755399441055744.0;
  const unsigned q = (unsigned)__double2loint(t);  // n mod 2^32
  double r = __builtin_fma(-n, 1.5707963267948966, x);
  r = __builtin_fma(-n, 6.123233995736766e-17, r);
  r = (__builtin_fabs(x) < 1125899906842624.0) ? r : __builtin_nan("");
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
  const bool odd = q & 1u;
  const double s = odd ? pc : ps, c = odd ? ps : pc;
  // sin changes sign in quadrants 2, 3 (bit 1 of q), cos in 1, 2 (bit 1 of q + 1): xor into the sign bit
  so = __hiloint2double(__double2hiint(s) ^ (int)((q << 30) & 0x80000000u), __double2loint(s));
  co = __hiloint2double(__double2hiint(c) ^ (int)(((q + 1u) << 30) & 0x80000000u), __double2loint(c));
}

// 1 / x for the joint-space inertias D (aba.hxx:165, 695): the hardware estimate and two Newton
// steps, 5 instructions where the IEEE division takes 11 (v_div_scale x2, v_div_fmas, v_div_fixup:
// the scaling only matters for results near the denormal range, and the last step only for the
// correctly rounded last bit).  Within 1 ulp of 1.0 / x for normal x (tools/ubench/rcp_check.hip);
// x = 0 gives NaN instead of inf, a singular D either way.
DI double spec_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

// ---- the macros the generated bodies use (codegen/emit.py: MACROS) ----
#define CT(k) (ct[k])
#define IN_Q(k) (mq.get(k, ldq8))
#define IN_V(k) (mv.get(k, ldv8))
#define IN_X(k) (mx.get(k, ldx8))
// grouped fetches (codegen/emit.py: group_inputs): real in the layout-2 kernels with SPEC_STAGE_MODE 2,
// empty everywhere else, where IN_*2 is the plain element load
#define GRP_Q(g, n) SpecGrp<SPEC_GROUPED> qg_##g##_##n; mq.grp(g, qg_##g##_##n)
#define GRP_V(g, n) SpecGrp<SPEC_GROUPED> vg_##g##_##n; mv.grp(g, vg_##g##_##n)
#define GRP_X(g, n) SpecGrp<SPEC_GROUPED> xg_##g##_##n; mx.grp(g, xg_##g##_##n)
#define GRPC_Q(g, n) mq.grpc(g, qg_##g##_##n)
#define GRPC_V(g, n) mv.grpc(g, vg_##g##_##n)
#define GRPC_X(g, n) mx.grpc(g, xg_##g##_##n)
#define IN_Q2(k, g, j, n) (mq.get2(k, ldq8, qg_##g##_##n, j))
#define IN_V2(k, g, j, n) (mv.get2(k, ldv8, vg_##g##_##n, j))
#define IN_X2(k, g, j, n) (mx.get2(k, ldx8, xg_##g##_##n, j))
// external forces: only in the kernel variants with WANT bit 5 (a compile-time constant)
#define IN_F(k) ((WANT & 32) ? mf.get(k, ldf8) : 0.0)
#define OUT(k, val) do { if (WANT & 1) mo.put(k, ldo8, val); } while (0)
#define OUT2(k, val) do { if (WANT & 2) mo2.put(k, ldo28, val); } while (0)
// output matrices read (kernels that update them in place, or read the third one: codegen/prod.py)
#define LD1(k) (mo.get(k, ldo8))
#define LD2(k) (mo2.get(k, ldo28))
#define LD3(k) (mo3.get(k, ldo38))
// mirror image of an element of the first output (computeMinverse: the strictly lower triangle):
// the value in the variants with WANT bit 1, a structural zero in the others
#define OUTSYM(k, val) do { if (WANT & 2) mo.put(k, ldo8, val); else if ((WANT & 17) == 17) mo.put(k, ldo8, 0.0); } while (0)
#define OUT3(k, val) do { if (WANT & 4) mo3.put(k, ldo38, val); } while (0)
#define OUT4(k, val) do { if (WANT & 8) mo4.put(k, ldo48, val); } while (0)
// structural zeros of the derivative matrices: only in the kernels that do not assume that the
// caller zeroed the outputs (WANT bit 4), as the reference does (rnea-derivatives.hpp:104-106)
#define OUTZ(k) do { if ((WANT & 17) == 17) mo.put(k, ldo8, 0.0); } while (0)
#define OUT2Z(k) do { if ((WANT & 18) == 18) mo2.put(k, ldo28, 0.0); } while (0)
#define OUT3Z(k) do { if ((WANT & 20) == 20) mo3.put(k, ldo38, 0.0); } while (0)
// getJointJacobian (codegen/jjac.py): is joint i on the support chain the host passed?  (wave-uniform)
#define ON_CHAIN(i) ((A.chain >> (i)) & 1ull)
#define SINCOS(a, s, c) spec_sincos(a, s, c)
#define RSQRT(a) rsqrt(a)
#define RCP(a) spec_rcp(a)
#define FMA(a, b, c) __builtin_fma(a, b, c)
// Experiment (tools/cg_variant.sh EXTRA=-DBRBD_TILE_LANES=32 + BRBD_HALF_TILE=1 at run time, BATCH_MAJOR
// RNEA only): tiles of 32 instances, the upper half of every wave idle, so that 65 536 instances are
// 2 048 waves = TWO per SIMD.  It measures what a second resident wave hides of the first one's
// waits without shortening anybody's program -- the upper bound of what splitting an instance over
// two lanes could win on top of its shorter programs.
#ifndef BRBD_TILE_LANES
#define BRBD_TILE_LANES WAVE
#endif
#define PK_LDS_ST2(p, a, b) lds2[(p) * BRBD_TILE_LANES] = make_double2(a, b)
#define PK_LDS_LD2(p, a, b) do { const double2 t_ = lds2[(p) * BRBD_TILE_LANES]; a = t_.x; b = t_.y; } while (0)
#define PK_GLB_ST2(p, a, b) do { SPEC_NOTE_WS((p) * (WAVE * 16)); spec_park_st(wsr, wsoff, p, a, b); } while (0)
#define PK_GLB_LD2(p, a, b) do { SPEC_NOTE_WS((p) * (WAVE * 16)); spec_park_ld(wsr, wsoff, p, a, b); } while (0)
// Visit boundary: nothing is scheduled across it, so the table loads of later visits are not
// hoisted up front (that is what cost 2000 SGPR spills in the template-unrolled experiment).
#define OPAQUE(x) asm volatile("" : "+v"(x))
// (Making the uniform bases opaque at every visit -- asm volatile("" : "+s"(ct), ...) -- kept LICM
// from hoisting the table loads out of a tile loop, but tied up SGPRs and, with hipcc 7.2, produced
// kernels that miscomputed under SGPR spilling.  The shells now run ONE tile per workgroup: there
// is no loop to hoist out of, and the scheduling barrier alone keeps the loads in their visit.)
#define BRBD_LAUNDER() do { } while (0)
#define BRBD_CAT2(a, b) a##b
#define BRBD_CAT(a, b) BRBD_CAT2(a, b)
#ifdef BRBD_SPEC_TRACE
// profiling build (tools/cg_variant.sh EXTRA=-DBRBD_SPEC_TRACE): wave 0 stamps the cycle counter at
// every visit boundary into the buffer the host passes in A.prog3 (BRBD_TRACE=1 dumps it)
#define VISIT_END()                                                                                  \
  do {                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    if (blockIdx.x == 0 && lane == 0 && A.prog3)                                                     \
      reinterpret_cast<long long*>(const_cast<Op*>(A.prog3))[trace_ix] = __builtin_readcyclecounter(); \
    ++trace_ix;                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  } while (0)
#else
#define VISIT_END()                                                                                  \
  do {                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    BRBD_LAUNDER();                                                                                  \
  } while (0)
#endif

// everything before the body: one tile of 64 instances per workgroup
#define BRBD_SPEC_PROLOGUE(LDS_PAIRS)                                                                \
  const int lane = threadIdx.x;                                                                      \
  const double BRBD_CONSTANT* ct = as_constant(A.ctab);                                              \
  __shared__ double2 lds_area[(LDS_PAIRS) > 0 ? (LDS_PAIRS) * BRBD_TILE_LANES : 1];                  \
  double2* const lds2 = lds_area + lane;                                                             \
  const __amdgpu_buffer_rsrc_t wsr = spec_rsrc(A.ws, A.ws_bytes);                                    \
  const unsigned wsoff = blockIdx.x * (unsigned)(A.ws_stride * 8) + (unsigned)lane * 16u;            \
  const unsigned ldq8 = (unsigned)A.ldq * 8u, ldv8 = (unsigned)A.ldv * 8u, ldx8 = (unsigned)A.lda * 8u; \
  const unsigned ldo8 = (unsigned)A.ldout * 8u, ldo28 = (unsigned)A.ldout2 * 8u;                     \
  const unsigned ldo38 = (unsigned)A.ldout3 * 8u, ldo48 = (unsigned)A.ldout4 * 8u;                   \
  const unsigned ldf8 = (unsigned)A.ldfext * 8u;                                                     \
  int trace_ix = 0;                                                                                  \
  SPEC_OOB_DECL; (void)spec_bad;                                                                     \
  SPEC_WS_LIMIT(wsoff, A.ws_bytes);                                                                  \
  (void)lds2; (void)wsr; (void)wsoff; (void)trace_ix;

// SPEC_NQ / SPEC_NV / SPEC_NX / SPEC_NO (rows of q, v, third input, first output), SPEC_STAGE_MODE and
// SPEC_GLB (doubles per lane of the global park area, after which the staging rows start) are
// constants of the generated kernel template (codegen/build.py).
#define BRBD_SPEC_TILE(LAYOUT)                                                                       \
  if (BRBD_TILE_LANES < WAVE && lane >= BRBD_TILE_LANES) return;                                     \
  int64_t inst = tile * BRBD_TILE_LANES + lane;                                                      \
  if (inst >= A.B) inst = A.B - 1;                                                                   \
  constexpr int LAY_ = (LAYOUT) == 2 ? 0 : (LAYOUT);  /* matrices that are not staged: per lane */   \
  constexpr int SPEC_GROUPED = ((LAYOUT) == 2 && SPEC_STAGE_MODE >= 2) ? SPEC_STAGE_MODE : 0;        \
  const SpecStage sc{tile, lane, reinterpret_cast<double*>(lds_area),                                \
                     A.ws + (int64_t)blockIdx.x * A.ws_stride + (int64_t)SPEC_GLB * WAVE,            \
                     (unsigned)((SPEC_NQ + SPEC_NV + SPEC_NX) * WAVE * 8),                            \
                     reinterpret_cast<double*>(lds_area) + SPEC_LDS_PARK * BRBD_TILE_LANES};         \
  SpecIn<LAYOUT, SPEC_IN_AUX, SPEC_NQ, SPEC_STAGE_MODE> mq(A.q, A.nb[0], ldq8, inst, SPEC_OOB, sc, 0);                \
  SpecIn<LAYOUT, SPEC_IN_AUX, SPEC_NV, SPEC_STAGE_MODE> mv(A.v, A.nb[1], ldv8, inst, SPEC_OOB, sc, SPEC_NQ);          \
  SpecIn<LAYOUT, SPEC_IN_AUX, SPEC_NX, SPEC_STAGE_MODE> mx(A.a, A.nb[2], ldx8, inst, SPEC_OOB, sc, SPEC_NQ + SPEC_NV); \
  mq.commit(sc); mv.commit(sc); mx.commit(sc); /* layout 2: all loads were issued above */           \
  if ((LAYOUT) == 2) __syncthreads();                                                                \
  const SpecOutSel<LAYOUT, SPEC_OUT_AUX, SPEC_NO, SPEC_OUT_STREAM> mo(A.out, A.nb[3], ldo8, inst, SPEC_OOB, sc);      \
  const SpecOutSel<(SPEC_OUT_STREAM ? (LAYOUT) : LAY_), SPEC_OUT_AUX, SPEC_NO2, SPEC_OUT_STREAM> mo2(A.out2, A.nb[4], ldo28, inst, SPEC_OOB, sc); \
  const SpecMat<LAY_, SPEC_OUT_AUX> mo3(A.out3, A.nb[5], ldo38, inst, SPEC_OOB);                        \
  const SpecMat<LAY_, SPEC_OUT_AUX> mo4(A.out4, A.nb[6], ldo48, inst, SPEC_OOB);                        \
  const SpecMat<LAY_, SPEC_IN_AUX> mf(A.fext, A.nb[7], ldf8, inst, SPEC_OOB);                           \
  (void)mv; (void)mx; (void)mo; (void)mo2; (void)mo3; (void)mo4; (void)ldo38; (void)ldo48; (void)mf; (void)ldf8; (void)sc;

// after the body: layout 2 stores its collected output
#define BRBD_SPEC_EPILOGUE() mo.flush(A.out, A.nb[3], sc)

}  // namespace brbd

// brbd_capi.hip -- C ABI (include/brbd.h) of the MI355X batched rigid-body-dynamics engine.
//
// Host side of the drop-in boundary: argument validation with the reference's error behaviour
// (algorithm/parallel/rnea.hpp:53-68, parallel/aba.hpp:54-69 -> std::invalid_argument, here
// BRBD_ERR_INVALID_ARGUMENT), the BatchModel flattener, workspace ownership and kernel launches.
// There is NO CPU fallback in this library: without a HIP device every compute entry point
// fails with BRBD_ERR_NO_DEVICE / BRBD_ERR_RUNTIME.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/brbd.h"
#include "brbd_flatten.hpp"
#include "brbd_kernels.hpp"
#include "brbd_deriv.hpp"
#include "brbd_f4.hpp"
#include "brbd_spec.hpp"
#include "gen/registry.inc"

using namespace brbd;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (void)hipGetLastError(); /* reported here: do not leave it for an unrelated later call */ \
      return fail(BRBD_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    }                                                                                       \
  } while (0)

// Compiled kernels loaded at run time (brbd_load_compiled_kernels): entries of plugin libraries,
// consulted after the ones this library was built with.
std::mutex g_plugin_mu;
std::vector<const brbd::SpecEntry*> g_plugin_entries;
bool g_plugin_env_done = false;

struct Ev {  // a hipEvent shared by the launches it delimits
  hipEvent_t e = nullptr;
  ~Ev() { if (e) (void)hipEventDestroy(e); }
};
struct TimedLaunch {
  std::shared_ptr<Ev> start, stop;
};

}  // namespace

// Everything a call needs beyond its arguments, per (handle, stream): the kernels' workspace, the
// device staging of the host-memspace path and the two internal streams its pipeline runs on.
// Calls on different streams of one handle never share any of it (SURVEY 8b "Threading": the C
// ABI is re-entrant per handle + stream); two host threads must not drive the SAME stream of the
// same handle at once, as with HIP itself.
struct StreamCtx {
  double* d_ws = nullptr;
  size_t ws_doubles = 0;
  double* d_stage = nullptr;  // host-memspace path: two chunk slots of device staging
  size_t stage_doubles = 0;
  double* d_tmp = nullptr;    // brbd_aba_derivatives: the acceleration when the caller does not want it
  size_t tmp_doubles = 0;
  double* d_xp = nullptr;     // via_batch_major: batch-major copies of one column chunk
  size_t xp_doubles = 0;
  double* h_stage = nullptr;  // pageable host matrices: two chunk slots of pinned bounce buffers
  size_t hstage_doubles = 0;
  hipStream_t pipe[2] = {nullptr, nullptr};  // internal streams of the host-memspace pipeline
  hipEvent_t ev_in = nullptr, ev_done[2] = {nullptr, nullptr};
};

struct brbd_model {
  FlatModel flat;
  int device = 0;
  int num_cu = 0;
  JointConst* d_tab = nullptr;
  Op* d_prog = nullptr;
  Op* d_prog3 = nullptr;
  int* d_nvsub = nullptr;
  std::map<hipStream_t, std::unique_ptr<StreamCtx>> ctxs;
  unsigned long long* d_oob = nullptr;  // checking build: out-of-range access counter of the compiled kernels
  double* scratch = nullptr;  // this device's copy of the file-scope scratch array (brbd_aba_noalloc)
  // per-algorithm launch geometry (waves resident on the device)
  // (resolved on first use; atomics: two host threads may drive two streams of one handle, both
  // would store the same value)
  std::atomic<int> max_waves[7] = {};
  std::atomic<int> max_waves_spec[7] = {};
  std::atomic<int> max_waves_jjac{0};
  std::atomic<int> max_waves_f4[5] = {};  // minverse, integrate, aba-derivative products (nv <= 24, <= 48, <= 96)
  std::map<int, std::pair<Op*, int>> chains;  // brbd_joint_jacobian: support chain of a joint (device)
  // model-compiled kernels (brbd_spec.hpp), when the model's signature is registered
  const SpecEntry* spec = nullptr;
  double* d_ct[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // packed constant table per algorithm
  bool prezeroed = false;  // brbd_model_outputs_prezeroed: derivative kernels skip structural zeros
  bool use_spec = true;  // BRBD_DISABLE_COMPILED=1 or brbd_model_use_compiled(m, 0): table-driven kernels
  bool use_lds = true;
  std::atomic<bool> warned_4g{false};
  int timing = 0;  // 0 off, 1 two events per launch, 2 chained (one event per launch)
  int timing_period = 1;            // brbd_set_timing_period: every period-th launch of an algorithm is timed
  uint64_t timing_count[7] = {0, 0, 0, 0, 0, 0, 0};
  std::shared_ptr<Ev> last_stop;
  hipStream_t last_stream = nullptr;
  std::vector<TimedLaunch> timed[7];
  std::mutex mu;
};

namespace {

// Signature of a model as pinocchio_amd/codegen/joints.py: ModelView.signature() builds it:
// tree, joint kinds, and per constant one of 0 / 1 / m(-1) / g(eneric); `dense`: geometry and
// inertias all generic (armature and gravity keep their pattern).
double model_constant(const FlatModel& F, int joint, int field, int index) {
  const JointConst& jc = F.tab_desc[joint];  // (as described: the compiled kernels know every joint kind)
  switch (field) {
    case SF_R: return jc.R[index];
    case SF_T: return jc.t[index];
    case SF_M: return jc.m;
    case SF_C: return jc.c[index];
    case SF_I: return jc.I[index];
    case SF_ARM: {  // the generator names an armature entry by its dof index; idx_v is part of the
      const int k = index - jc.idx_v;  // signature, so k is in range for a matched model -- checked anyway
      return (k >= 0 && k < jc.nvj) ? jc.arm[k] : 0.0;
    }
    case SF_AXIS: return jc.axis[index];
    default: return F.gravity[index];
  }
}
// mode 0: the exact pattern ("<name>_p"); 1: armature of 1-dof joints generic ("<name>_a": rotor
// inertias are run-time constants); 2: geometry and inertias generic too ("<name>_d").  Gravity is
// generic in every variant; the armature of multi-dof joints always keeps its pattern.
std::string model_signature(const FlatModel& F, int mode) {
  std::string s = "n" + std::to_string(F.njoints) + ";p";
  for (int i = 1; i < F.njoints; ++i) s += (i > 1 ? "," : "") + std::to_string(F.tab_desc[i].parent);
  s += ";j";
  for (int i = 1; i < F.njoints; ++i) s += (i > 1 ? "," : "") + std::to_string(F.tab_desc[i].jtype);
  // the generated bodies hard-code the rows of q / v / tau each joint reads and writes
  s += ";q";
  for (int i = 1; i < F.njoints; ++i) s += (i > 1 ? "," : "") + std::to_string(F.tab_desc[i].idx_q);
  s += ";v";
  for (int i = 1; i < F.njoints; ++i) s += (i > 1 ? "," : "") + std::to_string(F.tab_desc[i].idx_v);
  s += ";";
  auto put = [&](double x, bool generic) {
    char c = x == 0.0 ? '0' : (x == 1.0 ? '1' : (x == -1.0 ? 'm' : 'g'));
    s += generic ? 'g' : c;
  };
  const bool dense = mode == 2;
  for (int k = 0; k < 3; ++k) put(F.gravity[k], true);
  for (int i = 1; i < F.njoints; ++i) {
    const JointConst& jc = F.tab_desc[i];
    for (int k = 0; k < 9; ++k) put(jc.R[k], dense);
    for (int k = 0; k < 3; ++k) put(jc.t[k], dense);
    put(jc.m, dense);
    for (int k = 0; k < 3; ++k) put(jc.c[k], dense);
    for (int k = 0; k < 6; ++k) put(jc.I[k], dense);
    for (int k = 0; k < 3; ++k) put(jc.axis[k], dense);
    for (int k = 0; k < jc.nvj; ++k) put(jc.arm[k], mode >= 1 && jc.nvj == 1);
  }
  return s;
}
// A kernel's packed constant table from its gather list (codegen/emit.py: encode_const), 4 ints per
// entry: a model constant (field, joint, index, 0); a literal (15, low word, high word, 0); or an
// expression of EARLIER entries that the model compiler hoisted out of the kernel body (16 + op,
// operands as 2 * index + negated; op 0 a*b, 1 a+b, 2 a*b+c, 3 1/a, 4 1/sqrt(a)) -- e.g. the 6 x 6
// matrix of a body inertia, or a leaf joint's whole articulated-body projection, evaluated once here
// instead of once per wave per launch.
std::vector<double> pack_constants(const FlatModel& F, const SpecKernel& k) {
  std::vector<double> ct((size_t)k.nconst + 16, 0.0);  // padded: merged s_loads may read past the end
  auto ref = [&](int32_t r) { const double x = ct[(size_t)(r >> 1)]; return (r & 1) ? -x : x; };
  for (int c = 0; c < k.nconst; ++c) {
    const int32_t* g = k.gather + 4 * c;
    if (g[0] < 15) {
      ct[c] = model_constant(F, g[1], g[0], g[2]);
    } else if (g[0] == 15) {
      const uint64_t bits = (uint64_t)(uint32_t)g[1] | ((uint64_t)(uint32_t)g[2] << 32);
      std::memcpy(&ct[c], &bits, sizeof(double));
    } else {
      const double a = ref(g[1]);
      switch (g[0] - 16) {
        case 0: ct[c] = a * ref(g[2]); break;
        case 1: ct[c] = a + ref(g[2]); break;
        case 2: ct[c] = a * ref(g[2]) + ref(g[3]); break;
        case 3: ct[c] = 1.0 / a; break;
        default: ct[c] = 1.0 / std::sqrt(a); break;
      }
    }
  }
  return ct;
}

// The context of (handle, stream); created on first use.
StreamCtx* get_ctx(brbd_model* m, hipStream_t s) {
  std::lock_guard<std::mutex> g(m->mu);
  std::unique_ptr<StreamCtx>& c = m->ctxs[s];
  if (!c) c.reset(new StreamCtx());
  return c.get();
}

// Grow a per-stream device buffer.  Work already queued on `s` may still use the old one: wait for
// THAT stream (not the device) before freeing it.  Never reached once the buffer is warm.
int ensure_buf(double** buf, size_t* have, size_t doubles, hipStream_t s) {
  if (doubles <= *have) return BRBD_OK;
  if (*buf) {
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipFree(*buf));
  }
  *buf = nullptr;
  *have = 0;
  HIP_TRY(hipMalloc(buf, doubles * sizeof(double)));
  *have = doubles;
  return BRBD_OK;
}

void free_ctx(StreamCtx& c) {
  if (c.d_ws) (void)hipFree(c.d_ws);
  if (c.d_stage) (void)hipFree(c.d_stage);
  if (c.d_tmp) (void)hipFree(c.d_tmp);
  if (c.d_xp) (void)hipFree(c.d_xp);
  if (c.h_stage) (void)hipHostFree(c.h_stage);
  for (int k = 0; k < 2; ++k) {
    if (c.ev_done[k]) (void)hipEventDestroy(c.ev_done[k]);
    if (c.pipe[k]) (void)hipStreamDestroy(c.pipe[k]);
  }
  if (c.ev_in) (void)hipEventDestroy(c.ev_in);
}

template <typename K>
int resident_waves(brbd_model* m, K kernel, int block) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1)
    per_cu = 4;
  if (per_cu > 32) per_cu = 32;
  return per_cu * m->num_cu;
}

// Validate one dof x B matrix argument against the layout (reference: row/col checks,
// algorithm/parallel/rnea.hpp:61-68).
int check_mat(const char* name, const void* p, int64_t ld, int64_t dof, int64_t B, int layout) {
  if (!p) return fail(BRBD_ERR_INVALID_ARGUMENT, std::string(name) + " is NULL");
  const int64_t need = layout == BRBD_BATCH_MAJOR ? B : dof;
  if (ld < need)
    return fail(BRBD_ERR_INVALID_ARGUMENT, std::string("leading dimension of ") + name + " is " +
                                               std::to_string(ld) + ", needs >= " + std::to_string(need));
  return BRBD_OK;
}

int check_common(brbd_model* m, int64_t B, int layout, int memspace) {
  if (!m) return fail(BRBD_ERR_INVALID_ARGUMENT, "model handle is NULL");
  if (B < 0) return fail(BRBD_ERR_INVALID_ARGUMENT, "batch size is negative");
  if (layout != BRBD_INSTANCE_MAJOR && layout != BRBD_BATCH_MAJOR)
    return fail(BRBD_ERR_INVALID_ARGUMENT, "layout must be BRBD_INSTANCE_MAJOR or BRBD_BATCH_MAJOR");
  if (memspace != BRBD_HOST && memspace != BRBD_DEVICE)
    return fail(BRBD_ERR_INVALID_ARGUMENT, "memspace must be BRBD_HOST or BRBD_DEVICE");
  return BRBD_OK;
}

// Doubles a dof x B matrix really spans: the last row / column is not padded out to ld.
double mat_span_bytes(int64_t ld, int64_t dof, int64_t B, int layout) {
  const double outer = layout == BRBD_BATCH_MAJOR ? (double)dof : (double)B;
  const double inner = layout == BRBD_BATCH_MAJOR ? (double)B : (double)dof;
  return outer < 1 ? 0.0 : 8.0 * ((outer - 1) * (double)ld + inner);
}

void fill_common(const brbd_model* m, KArgs& A, int64_t B) {
  const FlatModel& F = m->flat;
  A.tab = m->d_tab;
  A.prog = m->d_prog;
  A.prog3 = m->d_prog3;
  A.nops = (int)F.prog.size();
  A.nops3 = (int)F.prog3.size();
  A.njoints = F.njoints;
  A.nq = F.nq;
  A.nv = F.nv;
  A.maxdepth = F.maxdepth;
  A.maxbranch = F.maxbranch;
  A.B = B;
  A.ntiles = (B + WAVE - 1) / WAVE;
  A.gx = F.gravity[0];
  A.gy = F.gravity[1];
  A.gz = F.gravity[2];
  A.oob = m->d_oob;
}

// brbd_set_timing_period: is this launch of `algo` one of the timed ones?  An untimed launch breaks
// the chain of mode 2 (the next timed launch opens with an event of its own).
static bool timing_sampled(brbd_model* m, int algo) {
  if (!m->timing || algo < 0) return false;
  if (m->timing_period <= 1) return true;
  std::lock_guard<std::mutex> g(m->mu);
  const bool yes = m->timing_count[algo]++ % (uint64_t)m->timing_period == 0;
  if (!yes) m->last_stop.reset();
  return yes;
}

struct Timer {
  brbd_model* m;
  int algo;
  hipStream_t s;
  TimedLaunch t{};
  bool on = false;
  int begin() {
    if (!timing_sampled(m, algo)) return BRBD_OK;
    // chained timing (brbd_set_timing(m, 2)): the event that closed the previous launch of this
    // handle on this stream opens this one: one event per launch instead of two
    {
      std::lock_guard<std::mutex> g(m->mu);  // (Timer::end and timing_sampled write these under the lock)
      if (m->timing >= 2 && m->last_stop && m->last_stream == s) t.start = m->last_stop;
    }
    if (!t.start) {
      t.start = std::make_shared<Ev>();
      HIP_TRY(hipEventCreate(&t.start->e));
      HIP_TRY(hipEventRecord(t.start->e, s));
    }
    t.stop = std::make_shared<Ev>();
    HIP_TRY(hipEventCreate(&t.stop->e));
    on = true;
    return BRBD_OK;
  }
  int end() {
    if (!on) return BRBD_OK;
    HIP_TRY(hipEventRecord(t.stop->e, s));
    std::lock_guard<std::mutex> g(m->mu);
    m->timed[algo].push_back(t);
    m->last_stop = t.stop;
    m->last_stream = s;
    return BRBD_OK;
  }
};

// Caller-provided workspace (brbd_aba_noalloc): table-driven kernels only, nothing allocated,
// no timing events; the grid is capped to what the workspace holds.
struct FixedWs {
  double* ptr = nullptr;
  size_t doubles = 0;
};

// One algorithm launch on device pointers.
// opt: RNEA derivatives: the output contract (-1 the handle's, 0 write every element, 1 outputs
// pre-zeroed by the caller); computeMinverse: 1 = symmetric fill.
int launch_algo(brbd_model* m, int algo, KArgs& A, int layout, hipStream_t stream, const FixedWs* fixed = nullptr,
                int opt = -1) {
  const FlatModel& F = m->flat;
  if (A.B == 0) return BRBD_OK;
  int per_lane = 0;
  switch (algo) {
    case BRBD_ALGO_RNEA: per_lane = F.stack_rnea(); break;
    case BRBD_ALGO_ABA: per_lane = F.ws_aba(); break;
    case BRBD_ALGO_FK_JAC: per_lane = F.ws_fkjac(); break;
    case BRBD_ALGO_RNEA_DERIV: per_lane = 1; break;  // (only reached when a compiled kernel exists)
    case BRBD_ALGO_MINVERSE: per_lane = 1; break;    // (likewise)
    case BRBD_ALGO_ABA_DERIV_PRODUCTS: per_lane = 1; break;
    case BRBD_ALGO_JOINT_JAC: per_lane = 1; break;
    default: return fail(BRBD_ERR_INVALID_ARGUMENT, "unknown algorithm");
  }
  // model-compiled kernel?  (BATCH_MAJOR addresses elements with 32-bit byte offsets)
  const SpecKernel* sk = (m->spec && m->use_spec && !fixed) ? &m->spec->algo[algo] : nullptr;
  int variant = 0, klayout = layout;
  if (sk) {
    if (algo == BRBD_ALGO_FK_JAC) variant = (A.out ? 1 : 0) | (A.out2 ? 2 : 0);
    else if (algo == BRBD_ALGO_RNEA_DERIV) variant = (opt < 0 ? m->prezeroed : opt != 0) ? 1 : 0;
    else if (algo == BRBD_ALGO_MINVERSE) variant = opt > 0 ? 1 : 0;
    else if (algo == BRBD_ALGO_JOINT_JAC) variant = A.rf;  // pinocchio::ReferenceFrame
    else variant = A.fext ? 1 : 0;  // rnea / aba: kernel variant with external forces
    // buffer ranges: every matrix must span less than 4 GiB (32-bit byte offsets)
    const int64_t nn = (int64_t)F.nv * F.nv;
    const bool der = algo == BRBD_ALGO_RNEA_DERIV || algo == BRBD_ALGO_MINVERSE || algo == BRBD_ALGO_ABA_DERIV_PRODUCTS;
    const bool jac = algo == BRBD_ALGO_FK_JAC || algo == BRBD_ALGO_JOINT_JAC;
    const int64_t dofs[8] = {F.nq, F.nv, F.nv, der ? nn : (jac ? 6 * (int64_t)F.nv : F.nv),
                             der ? nn : 12 * (int64_t)F.njoints, nn, F.nv, 6 * (int64_t)F.njoints};
    const void* ptrs[8] = {A.q, A.v, A.a, A.out, A.out2, A.out3, A.out4, A.fext};
    const int64_t lds_[8] = {A.ldq, A.ldv, A.lda, A.ldout, A.ldout2, A.ldout3, A.ldout4, A.ldfext};
    for (int k = 0; k < 8 && sk; ++k) {
      A.nb[k] = 0;
      if (!ptrs[k]) continue;
      const double span = mat_span_bytes(lds_[k], dofs[k], A.B, layout);  // exact: the range check is the bound
      if (span >= 4294967295.0) {
        // 32-bit byte offsets: a matrix of 4 GiB or more runs on the (slower) table-driven kernels.
        // Loud, once per handle: callers should split the batch by columns instead.
        if (!m->warned_4g.exchange(true)) {
          fprintf(stderr, "brbd: a boundary matrix of this call spans >= 4 GiB: the model-compiled kernels "
                          "(32-bit offsets) do not apply, using the table-driven kernels; split the batch by "
                          "columns to stay on the compiled ones\n");
        }
        sk = nullptr;
      } else {
        A.nb[k] = (uint32_t)span;
      }
    }
#ifdef BRBD_SPEC_CHECK
    {  // positive control of the checking build (tests/test_gpu_bounds_check.py): BRBD_CHECK_SHRINK=k
       // declares matrix k (q, v, a, out, out2, out3, out4, fext) 8 bytes short of what the kernel
       // addresses, so its last element is out of range -- counted by the checks, dropped by the hardware
      const char* e = getenv("BRBD_CHECK_SHRINK");
      const int k = e ? atoi(e) : -1;
      if (sk && k >= 0 && k < 8 && A.nb[k] >= 8) A.nb[k] -= 8;
    }
#endif
    // INSTANCE_MAJOR with tight leading dimensions (what Eigen hands the reference): the kernels
    // that stage the tile through LDS (brbd_spec.hpp, kernel layout 2)
    static const bool no_staged = getenv("BRBD_DISABLE_STAGED") && getenv("BRBD_DISABLE_STAGED")[0] == '1';
    if (sk && layout == BRBD_INSTANCE_MAJOR && !no_staged && (algo == BRBD_ALGO_RNEA || algo == BRBD_ALGO_ABA) &&
        sk->fn[2][variant] && A.ldq == F.nq && A.ldv == F.nv && A.lda == F.nv && A.ldout == F.nv)
      klayout = 2;
    // (forward kinematics + Jacobians: q staged in, J and oMi streamed out in chunks of 64 rows)
    if (sk && layout == BRBD_INSTANCE_MAJOR && !no_staged && jac && sk->fn[2][variant] &&
        A.ldq == F.nq && (!A.out || A.ldout == 6 * (int64_t)F.nv) && (!A.out2 || A.ldout2 == 12 * (int64_t)F.njoints))
      klayout = 2;
    if (sk && !sk->fn[klayout][variant]) sk = nullptr;
  }
  if ((algo == BRBD_ALGO_RNEA_DERIV || algo == BRBD_ALGO_MINVERSE || algo == BRBD_ALGO_ABA_DERIV_PRODUCTS ||
       algo == BRBD_ALGO_JOINT_JAC) && !sk) return 1;  // caller falls back to the table-driven kernel
  const bool stat = sk != nullptr;
  SpecFn sfn = stat ? sk->fn[klayout][variant] : nullptr;
  {  // experiment (brbd_spec.hpp: BRBD_TILE_LANES): kernels built with 32-instance tiles
    static const bool half = getenv("BRBD_HALF_TILE") && getenv("BRBD_HALF_TILE")[0] == '1';
    if (half && stat && algo == BRBD_ALGO_RNEA && klayout == 1) A.ntiles = (A.B + 31) / 32;
  }
  const int stage = stat && klayout == 2 ? sk->stage_doubles : 0;
  std::atomic<int>& mw_cache = stat ? m->max_waves_spec[algo] : m->max_waves[algo];
  int mw = mw_cache.load(std::memory_order_relaxed);
  if (mw == 0) {
    if (stat) mw = resident_waves(m, sfn, WAVE);
    else if (algo == BRBD_ALGO_RNEA) mw = resident_waves(m, rnea_kernel<1>, WAVE);
    else if (algo == BRBD_ALGO_ABA) mw = resident_waves(m, aba_kernel<1>, WAVE);
    else mw = resident_waves(m, fkjac_kernel<1>, WAVE);
    mw_cache.store(mw, std::memory_order_relaxed);  // register/LDS-limited resident waves on the whole device
  }
  // LDS part of the live-state stack (generic RNEA / ABA): as many slots as fit without lowering
  // the number of waves per CU below what this launch can use.
  int lds_cap = 0;
  size_t lds_bytes = 0;
  int64_t resident = mw;
  if (!stat && (algo == BRBD_ALGO_RNEA || algo == BRBD_ALGO_ABA)) {
    const int stack = algo == BRBD_ALGO_RNEA ? F.stack_rnea() : F.stack_aba();
    const int64_t per_cu_vgpr = mw / m->num_cu;
    int64_t want_per_cu = (A.ntiles + m->num_cu - 1) / m->num_cu;
    if (want_per_cu > per_cu_vgpr) want_per_cu = per_cu_vgpr;
    if (want_per_cu < 1) want_per_cu = 1;
    const int64_t lds_per_cu = 160 * 1024;
    int64_t cap = lds_per_cu / (want_per_cu * WAVE * 8);
    if (cap > stack) cap = stack;
    if (cap > 96) cap = 96;  // 48 KiB dynamic LDS per workgroup at most
    if (!m->use_lds) cap = 0;
    lds_cap = (int)cap;
    lds_bytes = (size_t)lds_cap * WAVE * 8;
    per_lane = (stack - lds_cap) + (algo == BRBD_ALGO_ABA ? F.hand_total : 0);
    if (per_lane < 1) per_lane = 1;
  } else if (stat) {
    per_lane = sk->glb_doubles + stage > 0 ? sk->glb_doubles + stage : 1;  // global park area + staging rows
    A.ctab = m->d_ct[algo];
  }
  A.lds_cap = lds_cap;
  // table-driven kernels: grid-stride over the tiles with as many waves as are resident.
  // model-compiled kernels: one tile per workgroup; launched in chunks when they park state in
  // the wave-private global area, so that the area stays bounded (8 rounds of resident waves).
  int64_t grid = A.ntiles < resident ? A.ntiles : resident;
  if (stat) {
    static const int rounds = getenv("BRBD_CHUNK_ROUNDS") ? atoi(getenv("BRBD_CHUNK_ROUNDS")) : 8;
    // (test hook: BRBD_CHUNK_ROUNDS = -n launches n TILES at a time, whatever the kernel parks)
    int64_t chunk = rounds < 0 ? -rounds : (sk->glb_doubles + stage > 0 ? resident * (rounds > 0 ? rounds : 1) : A.ntiles);
    const int64_t cap = (int64_t)(4000000000.0 / ((double)per_lane * WAVE * 8));  // 32-bit park offsets
    if (chunk > cap) chunk = cap;
    grid = A.ntiles < chunk ? A.ntiles : chunk;
  }
  A.ws_stride = (int64_t)per_lane * WAVE;
  int rc = BRBD_OK;
  if (fixed) {
    const int64_t fit = (int64_t)(fixed->doubles / (size_t)A.ws_stride);
    if (fit < 1)
      return fail(BRBD_ERR_INVALID_ARGUMENT, "this model needs " + std::to_string(A.ws_stride * 8) +
                  " bytes of workspace per wave, more than the fixed scratch area holds");
    if (grid > fit) grid = fit;  // the table-driven kernels stride over the tiles: fewer waves, more rounds
    A.ws = fixed->ptr;
  } else {
    StreamCtx* ctx = get_ctx(m, stream);
    if ((rc = ensure_buf(&ctx->d_ws, &ctx->ws_doubles, (size_t)grid * A.ws_stride, stream))) return rc;
    A.ws = ctx->d_ws;
  }
  A.ws_bytes = (uint32_t)((size_t)grid * A.ws_stride * 8);
#ifdef BRBD_SPEC_CHECK
  {  // (BRBD_CHECK_SHRINK=8: the wave-private park area, 16 bytes short)
    const char* e = getenv("BRBD_CHECK_SHRINK");
    if (stat && e && atoi(e) == 8 && A.ws_bytes >= 16) A.ws_bytes -= 16;
  }
#endif
  // timing mode 3 applies to calls that are ONE dispatch of a compiled kernel; anything else is timed
  // between chained events like mode 2
  const bool dispatch_mode = m->timing == 3 && stat && !fixed && A.ntiles <= grid;
  const bool dispatch_timed = dispatch_mode && timing_sampled(m, algo);
  Timer tm{m, (fixed || dispatch_mode) ? -1 : algo, stream};
  if ((rc = tm.begin())) return rc;
  const dim3 g((unsigned)grid), b(WAVE);
  if (stat) {
    // profiling builds only (-DBRBD_SPEC_TRACE): per-visit cycle stamps of wave 0, dumped to stderr
    static const bool trace = getenv("BRBD_TRACE") && getenv("BRBD_TRACE")[0] == '1';
    long long* d_trace = nullptr;
    A.prog3 = nullptr;
    if (trace && hipMalloc(&d_trace, 4096 * sizeof(long long)) == hipSuccess) {
      (void)hipMemset(d_trace, 0, 4096 * sizeof(long long));
      A.prog3 = reinterpret_cast<const Op*>(d_trace);
    }
    if (dispatch_timed) {
      // brbd_set_timing(m, 3): the one dispatch of this call carries its own begin / end timestamps
      // (hipExtLaunchKernelGGL): no marker packets between the launches of a stream
      TimedLaunch t{};
      t.start = std::make_shared<Ev>();
      t.stop = std::make_shared<Ev>();
      HIP_TRY(hipEventCreate(&t.start->e));
      HIP_TRY(hipEventCreate(&t.stop->e));
      A.tile0 = 0;
      hipExtLaunchKernelGGL(sfn, dim3((unsigned)A.ntiles), b, 0, stream, t.start->e, t.stop->e, 0, A);
      std::lock_guard<std::mutex> g(m->mu);
      m->timed[algo].push_back(t);
    } else
    for (int64_t t0 = 0; t0 < A.ntiles; t0 += grid) {
      A.tile0 = t0;
      const int64_t n = A.ntiles - t0 < grid ? A.ntiles - t0 : grid;
      hipLaunchKernelGGL(sfn, dim3((unsigned)n), b, 0, stream, A);
    }
    if (d_trace) {
      std::vector<long long> h(4096);
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpy(h.data(), d_trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
      (void)hipFree(d_trace);
      fprintf(stderr, "BRBD_TRACE algo %d layout %d:", algo, layout);
      for (size_t k = 0; k < h.size() && h[k]; ++k) fprintf(stderr, " %lld", k ? h[k] - h[k - 1] : 0LL);
      fprintf(stderr, "\n");
    }
  } else if (algo == BRBD_ALGO_RNEA) {
    if (layout == BRBD_BATCH_MAJOR) hipLaunchKernelGGL(rnea_kernel<1>, g, b, lds_bytes, stream, A);
    else hipLaunchKernelGGL(rnea_kernel<0>, g, b, lds_bytes, stream, A);
  } else if (algo == BRBD_ALGO_ABA) {
    if (layout == BRBD_BATCH_MAJOR) hipLaunchKernelGGL(aba_kernel<1>, g, b, lds_bytes, stream, A);
    else hipLaunchKernelGGL(aba_kernel<0>, g, b, lds_bytes, stream, A);
  } else {
    if (layout == BRBD_BATCH_MAJOR) hipLaunchKernelGGL(fkjac_kernel<1>, g, b, 0, stream, A);
    else hipLaunchKernelGGL(fkjac_kernel<0>, g, b, 0, stream, A);
  }
  HIP_TRY(hipGetLastError());
  return tm.end();
}

struct DevGuard {
  int prev = -1;
  bool ok = false;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
  }
  ~DevGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace

extern "C" {

const char* brbd_last_error(void) { return g_err.c_str(); }

const char* brbd_version(void) { return "brbd 0.1 (gfx950, hip)"; }

int brbd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int brbd_model_canonical_form(const brbd_model_desc* desc, int32_t* jtype, double* placement,
                              double* inertia, double* post) {
  if (!jtype || !placement || !inertia || !post) return fail(BRBD_ERR_INVALID_ARGUMENT, "output table is NULL");
  FlatModel F;
  int status = BRBD_OK;
  const std::string err = flatten(desc, &F, &status);
  if (status != BRBD_OK) return fail(status, err);
  for (int i = 0; i < F.njoints; ++i) {
    const JointConst& jc = F.tab[i];
    const bool canon = i > 0 && jc.jtype != F.tab_desc[i].jtype;
    jtype[i] = jc.jtype;
    double* P = placement + 12 * (size_t)i;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) P[3 * c + r] = i ? jc.R[3 * r + c] : (r == c ? 1.0 : 0.0);
    for (int k = 0; k < 3; ++k) P[9 + k] = jc.t[k];
    double* Y = inertia + 10 * (size_t)i;
    Y[0] = jc.m;
    for (int k = 0; k < 3; ++k) Y[1 + k] = jc.c[k];
    for (int k = 0; k < 6; ++k) Y[4 + k] = jc.I[k];
    for (int k = 0; k < 9; ++k) post[9 * (size_t)i + k] = canon ? jc.post[k] : ((k % 4 == 0) ? 1.0 : 0.0);
  }
  return BRBD_OK;
}

static int resolve_scratch(brbd_model* m);  // (no-alloc variant, below)

// Kernels compiled for a model AFTER this library was built: pinocchio_amd/codegen/plugin.py runs the
// model compiler on any model (a URDF, a pinocchio::Model's tables) and links the kernels into a
// small shared library; loading it here makes brbd_model_create select them for models with that
// signature, exactly like the built-in ones.  Returns the number of entries added (>= 0) or a
// negative status.
int brbd_load_compiled_kernels(const char* path) {
  if (!path || !*path) return fail(BRBD_ERR_INVALID_ARGUMENT, "path is NULL or empty");
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return fail(BRBD_ERR_INVALID_ARGUMENT, std::string("cannot load ") + path + ": " + dlerror());
  typedef const SpecEntry* const* (*plugin_fn)(int*, int*);
  plugin_fn fn = (plugin_fn)dlsym(h, "brbd_spec_plugin");
  if (!fn) {
    dlclose(h);
    return fail(BRBD_ERR_INVALID_ARGUMENT, std::string(path) + " is not a brbd kernel plugin (no brbd_spec_plugin)");
  }
  int count = 0, info[3] = {0, 0, 0};
  const SpecEntry* const* entries = fn(&count, info);
  if (info[0] != BRBD_SPEC_ABI || info[1] != (int)sizeof(KArgs) || info[2] != (int)sizeof(SpecEntry)) {
    dlclose(h);
    return fail(BRBD_ERR_INVALID_ARGUMENT, std::string(path) + " was built against another layout of the kernel "
                "arguments (ABI " + std::to_string(info[0]) + ", library " + std::to_string(BRBD_SPEC_ABI) + "): rebuild it");
  }
  int added = 0;
  std::lock_guard<std::mutex> g(g_plugin_mu);
  for (int k = 0; k < count; ++k) {
    bool dup = false;
    for (const SpecEntry* e : g_plugin_entries) dup = dup || std::string(e->signature) == entries[k]->signature;
    if (!dup) {
      g_plugin_entries.push_back(entries[k]);
      ++added;
    }
  }
  return added;  // (the library stays loaded for the life of the process: its kernels are in use)
}

static void load_plugins_from_env() {  // BRBD_COMPILED_PLUGINS = a.so:b.so, read once
  {
    std::lock_guard<std::mutex> g(g_plugin_mu);
    if (g_plugin_env_done) return;
    g_plugin_env_done = true;
  }
  const char* env = getenv("BRBD_COMPILED_PLUGINS");
  if (!env) return;
  std::string list(env);
  size_t pos = 0;
  while (pos <= list.size()) {
    const size_t end = list.find(':', pos);
    const std::string one = list.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
    if (!one.empty() && brbd_load_compiled_kernels(one.c_str()) < 0)
      fprintf(stderr, "brbd: BRBD_COMPILED_PLUGINS: %s\n", g_err.c_str());
    if (end == std::string::npos) break;
    pos = end + 1;
  }
}

int brbd_model_create(const brbd_model_desc* desc, int device, brbd_model** out) {
  if (!out) return fail(BRBD_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  FlatModel F;
  int status = BRBD_OK;
  const std::string err = flatten(desc, &F, &status);
  if (status != BRBD_OK) return fail(status, err);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BRBD_ERR_NO_DEVICE, "no HIP device available: the engine has no CPU fallback");
  if (device < 0 || device >= ndev)
    return fail(BRBD_ERR_INVALID_ARGUMENT, "device index " + std::to_string(device) + " out of range");
  DevGuard dg(device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  brbd_model* m = new brbd_model();
  m->flat = F;
  m->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete m;
    return fail(BRBD_ERR_RUNTIME, "hipGetDeviceProperties failed");
  }
  m->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  {
    const char* env = getenv("BRBD_DISABLE_COMPILED");
    if (env && env[0] == '1') m->use_spec = false;
    env = getenv("BRBD_DISABLE_LDS");
    if (env && env[0] == '1') m->use_lds = false;
  }
  auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  hipError_t e = upload((void**)&m->d_tab, m->flat.tab.data(), m->flat.tab.size() * sizeof(JointConst));
  if (e == hipSuccess) e = upload((void**)&m->d_prog, m->flat.prog.data(), m->flat.prog.size() * sizeof(Op));
  if (e == hipSuccess) e = upload((void**)&m->d_prog3, m->flat.prog3.data(), m->flat.prog3.size() * sizeof(Op));
  if (e == hipSuccess)
    e = upload((void**)&m->d_nvsub, m->flat.nv_subtree.data(), m->flat.nv_subtree.size() * sizeof(int));
  if (e == hipSuccess) {
    // model-compiled kernels: exact zero pattern first, then the same pattern with run-time
    // armature, then the dense kernels of the same tree; the kernels built into the library first,
    // then those of plugins loaded at run time
    load_plugins_from_env();
    const std::string sig[3] = {model_signature(m->flat, 0), model_signature(m->flat, 1), model_signature(m->flat, 2)};
    std::lock_guard<std::mutex> g(g_plugin_mu);
    for (int pass = 0; pass < 3 && !m->spec; ++pass) {
      for (int k = 0; k < spec_entry_count && !m->spec; ++k)
        if (sig[pass] == spec_entries[k]->signature) m->spec = spec_entries[k];
      for (size_t k = 0; k < g_plugin_entries.size() && !m->spec; ++k)
        if (sig[pass] == g_plugin_entries[k]->signature) m->spec = g_plugin_entries[k];
    }
    for (int a = 0; m->spec && a < 7 && e == hipSuccess; ++a) {
      const std::vector<double> ct = pack_constants(m->flat, m->spec->algo[a]);
      e = upload((void**)&m->d_ct[a], ct.data(), ct.size() * sizeof(double));
    }
  }
  if (e != hipSuccess) {
    brbd_model_destroy(m);
    return fail(BRBD_ERR_RUNTIME, std::string("uploading the model table: ") + hipGetErrorString(e));
  }
#ifdef BRBD_SPEC_CHECK
  if (hipMalloc(&m->d_oob, 8 * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(m->d_oob, 0, 8 * sizeof(unsigned long long)) != hipSuccess) {
    brbd_model_destroy(m);
    return fail(BRBD_ERR_RUNTIME, "allocating the bounds-check counter");
  }
#endif
  // no-alloc variant: this device's address of the file-scope scratch array, and the launch
  // geometry of its kernel, resolved here so that brbd_aba_noalloc itself only launches
  if (resolve_scratch(m) != BRBD_OK) {
    const std::string msg = g_err;
    brbd_model_destroy(m);
    return fail(BRBD_ERR_RUNTIME, msg);
  }
  m->max_waves[BRBD_ALGO_ABA].store(resident_waves(m, aba_kernel<1>, WAVE));
  *out = m;
  return BRBD_OK;
}

int brbd_model_destroy(brbd_model* m) {
  if (!m) return BRBD_OK;
  DevGuard dg(m->device);
  for (auto& c : m->chains) (void)hipFree(c.second.first);
  for (auto& v : m->timed) v.clear();  // (events are destroyed with their last holder)
  m->last_stop.reset();
  if (m->d_tab) (void)hipFree(m->d_tab);
  if (m->d_prog) (void)hipFree(m->d_prog);
  if (m->d_prog3) (void)hipFree(m->d_prog3);
  if (m->d_nvsub) (void)hipFree(m->d_nvsub);
  for (double* p : m->d_ct)
    if (p) (void)hipFree(p);
  for (auto& c : m->ctxs)
    if (c.second) free_ctx(*c.second);
  if (m->d_oob) (void)hipFree(m->d_oob);
  delete m;
  return BRBD_OK;
}

int brbd_model_release_stream(brbd_model* m, void* stream) {
  if (!m) return fail(BRBD_ERR_INVALID_ARGUMENT, "model handle is NULL");
  DevGuard dg(m->device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  std::unique_ptr<StreamCtx> ctx;
  {
    std::lock_guard<std::mutex> g(m->mu);
    auto it = m->ctxs.find((hipStream_t)stream);
    if (it == m->ctxs.end()) return BRBD_OK;
    ctx = std::move(it->second);
    m->ctxs.erase(it);
    if (m->last_stream == (hipStream_t)stream) m->last_stop.reset();
  }
  if (!ctx) return BRBD_OK;
  // everything queued with these buffers must have finished before they go
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  for (int k = 0; k < 2; ++k)
    if (ctx->pipe[k]) (void)hipStreamSynchronize(ctx->pipe[k]);
  free_ctx(*ctx);
  return BRBD_OK;
}

int brbd_model_dims(const brbd_model* m, int32_t* njoints, int32_t* nq, int32_t* nv) {
  if (!m) return fail(BRBD_ERR_INVALID_ARGUMENT, "model handle is NULL");
  if (njoints) *njoints = m->flat.njoints;
  if (nq) *nq = m->flat.nq;
  if (nv) *nv = m->flat.nv;
  return BRBD_OK;
}

int64_t brbd_workspace_bytes(const brbd_model* m, int algo, int64_t B) {
  if (!m || B < 0) return -1;
  const FlatModel& F = m->flat;
  int per_lane = 0;
  switch (algo) {
    case BRBD_ALGO_RNEA: per_lane = F.stack_rnea(); break;
    case BRBD_ALGO_ABA: per_lane = F.ws_aba(); break;
    case BRBD_ALGO_FK_JAC: per_lane = F.ws_fkjac(); break;
    case BRBD_ALGO_RNEA_DERIV: per_lane = deriv_ws_per_lane(F); break;
    case BRBD_ALGO_MINVERSE: per_lane = minverse_ws_per_lane(F); break;
    default: return -1;  // (BRBD_ALGO_ABA_DERIV_PRODUCTS: no workspace of its own)
  }
  const int64_t tiles = (B + WAVE - 1) / WAVE;
  return tiles * WAVE * (int64_t)per_lane * 8;
}

// ---- host-memspace plumbing ---------------------------------------------------------------------
// The reference's calling convention: every matrix lives in host memory (Eigen::MatrixXd,
// algorithm/parallel/rnea.hpp:39-45).  The batch is cut into column chunks that go through two
// device staging slots on two internal streams: while chunk k computes, chunk k + 1 uploads and
// chunk k - 1 downloads (PCIe is full duplex).  Copies are 2-D (pitch = the caller's leading
// dimension, width = the rows / columns that exist), so caller padding is neither read nor written
// and the device side is always tight.  With pageable host memory the copies block the host and
// only the kernels overlap; with pinned memory (hipHostMalloc, hipHostRegister / brbd_host_register,
// torch pin_memory) all three stages overlap.  Results are valid when the call returns.
extern "C++" {
namespace {

// ---- INSTANCE_MAJOR device matrices through the BATCH_MAJOR kernels -------------------------------
// An output-bound algorithm (the RNEA derivatives: 3 nv x nv matrices per instance) writes its
// results one element at a time; in the reference's own layout that is, per store, 64 lanes
// hitting 64 different cache lines several KiB apart, and the memory system merges almost none of
// it (measured: 7x the BATCH_MAJOR time).  Instead: a column chunk of the inputs is transposed
// into per-stream batch-major copies, the BATCH_MAJOR kernel runs on those, and the outputs are
// transposed back with 512-byte runs on both sides (xpose64_kernel).  Chunks of up to 1 GiB of
// copies (BRBD_VIA_BM_CHUNK = instances per chunk overrides): measured, larger chunks are faster --
// the launches of a small chunk do not fill the device.  Same kernels, same arithmetic, same bits.
struct XpMat {
  const double* in;  // device matrix read by the algorithm (NULL: not an input)
  double* out;       // device matrix written by it (NULL: not an output)
  int64_t ld, dof;
};

bool via_bm_enabled() {
  static const bool off = getenv("BRBD_DISABLE_VIA_BM") && getenv("BRBD_DISABLE_VIA_BM")[0] == '1';
  return !off;
}

template <typename Launch>
int via_batch_major(brbd_model* m, int64_t B, const XpMat* mats, int nm, hipStream_t st, Launch launch) {
  int64_t tot = 0;
  for (int i = 0; i < nm; ++i) tot += mats[i].dof;
  static const int64_t forced = getenv("BRBD_VIA_BM_CHUNK") ? atoll(getenv("BRBD_VIA_BM_CHUNK")) : 0;
  int64_t C = forced > 0 ? forced : (int64_t)((1024u << 20) / 8) / (tot > 0 ? tot : 1);  // 1 GiB of copies
  C = (C / WAVE) * WAVE;
  if (C < 1024) C = 1024;
  {  // equal chunks: a short last one would leave most of the device idle for three launches
    const int64_t n = (B + C - 1) / C;
    C = (((B + n - 1) / n + WAVE - 1) / WAVE) * WAVE;
  }
  StreamCtx* ctx = get_ctx(m, st);
  int rc = ensure_buf(&ctx->d_xp, &ctx->xp_doubles, (size_t)C * (size_t)tot, st);
  if (rc) return rc;
  double* d[8];
  int64_t ld[8];
  {
    size_t off = 0;
    for (int i = 0; i < nm; ++i) {
      d[i] = mats[i].dof ? ctx->d_xp + off : nullptr;
      ld[i] = C;
      off += (size_t)mats[i].dof * (size_t)C;
    }
  }
  for (int64_t c0 = 0; c0 < B; c0 += C) {
    const int64_t Bc = B - c0 < C ? B - c0 : C;
    const unsigned gx = (unsigned)((Bc + 63) / 64);
    for (int i = 0; i < nm; ++i)
      if (mats[i].in && mats[i].dof)
        hipLaunchKernelGGL(xpose64_kernel<1>, dim3(gx), dim3(256), 0, st,
                           mats[i].in + c0 * mats[i].ld, mats[i].ld, d[i], C, mats[i].dof, Bc);
    HIP_TRY(hipGetLastError());
    if ((rc = launch(Bc, d, ld, st))) return rc;
    for (int i = 0; i < nm; ++i)
      if (mats[i].out && mats[i].dof)
        hipLaunchKernelGGL(xpose64_kernel<0>, dim3(gx), dim3(256), 0, st, d[i], C,
                           mats[i].out + c0 * mats[i].ld, mats[i].ld, mats[i].dof, Bc);
    HIP_TRY(hipGetLastError());
  }
  return BRBD_OK;
}

struct HostMat {
  const double* src;  // host matrix (uploaded when `up`)
  double* dst;        // same matrix as an output (downloaded when `down`), else NULL
  int64_t ld, dof;
  bool up, down;
};

size_t padded(size_t doubles) { return (doubles + 31) & ~size_t(31); }

// ---- pageable host matrices: packed by worker threads into pinned bounce buffers ------------------
// hipMemcpyAsync from pageable memory is staged by the runtime on the calling thread (measured:
// 2.0e7 evals/s for the humanoid step against 4.5e7 from pinned memory).  The reference's callers
// hold plain Eigen matrices, so the library does that staging itself, in parallel: worker threads
// copy the chunk's columns into a pinned image (and the results back out of one) while the DMA
// engines and the kernels work on the neighbouring chunk.
struct CopyTask {
  char* dst;
  const char* src;
  size_t dpitch, spitch, width, rows;
};

class CopyPool {  // one per process; parallel_run calls are serialised
 public:
  static CopyPool& get() {
    static CopyPool* p = new CopyPool();  // (never destroyed: workers may outlive static destructors)
    return *p;
  }
  int threads() const { return nthreads_; }
  void run(const std::vector<CopyTask>& tasks) {
    if (tasks.empty()) return;
    if (nthreads_ <= 1 || tasks.size() == 1) {
      for (const CopyTask& t : tasks) do_task(t);
      return;
    }
    std::lock_guard<std::mutex> serial(run_mu_);
    {
      std::lock_guard<std::mutex> g(mu_);
      tasks_ = &tasks;
      next_.store(0);
      pending_ = (int)workers_.size();
      ++generation_;
    }
    cv_.notify_all();
    work();  // the calling thread takes tasks too
    std::unique_lock<std::mutex> g(mu_);
    done_cv_.wait(g, [&] { return pending_ == 0; });
    tasks_ = nullptr;
  }

 private:
  CopyPool() {
    const char* e = getenv("BRBD_HOST_THREADS");
    int n = e ? atoi(e) : 6;  // (measured: 4 saturate the copies of one GPU)
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && n > hw) n = hw;
    if (n < 0) n = 0;
    nthreads_ = n;
    for (int k = 1; k < n; ++k) {
      workers_.emplace_back([this] { loop(); });
      workers_.back().detach();
    }
  }
  static void do_task(const CopyTask& t) {
    for (size_t r = 0; r < t.rows; ++r) memcpy(t.dst + r * t.dpitch, t.src + r * t.spitch, t.width);
  }
  void work() {
    const std::vector<CopyTask>& T = *tasks_;
    for (size_t k = next_.fetch_add(1); k < T.size(); k = next_.fetch_add(1)) do_task(T[k]);
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return generation_ != seen; });
        seen = generation_;
      }
      work();
      std::lock_guard<std::mutex> g(mu_);
      if (--pending_ == 0) done_cv_.notify_one();
    }
  }
  int nthreads_ = 0;
  std::vector<std::thread> workers_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_, done_cv_;
  const std::vector<CopyTask>* tasks_ = nullptr;
  std::atomic<size_t> next_{0};
  int pending_ = 0;
  unsigned long long generation_ = 0;
};

// split a pitched 2-D copy into tasks of about 128 KiB
void add_copy_tasks(std::vector<CopyTask>& out, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width,
                    size_t rows) {
  const size_t grain = 128 * 1024;
  if (width == 0 || rows == 0) return;
  if (dpitch == width && spitch == width) {  // one contiguous block
    width *= rows;
    rows = 1;
  }
  if (width >= grain) {
    for (size_t r = 0; r < rows; ++r)
      for (size_t c = 0; c < width; c += grain)
        out.push_back({(char*)dst + r * dpitch + c, (const char*)src + r * spitch + c, 0, 0,
                       width - c < grain ? width - c : grain, 1});
  } else {
    const size_t per = grain / width;
    for (size_t r = 0; r < rows; r += per)
      out.push_back({(char*)dst + r * dpitch, (const char*)src + r * spitch, dpitch, spitch, width,
                     rows - r < per ? rows - r : per});
  }
}

bool is_device_visible_host(const void* p) {  // pinned / registered host memory?
  if (!p) return true;
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // (plain malloc memory: invalid value on older runtimes)
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

int ensure_host_buf(double** buf, size_t* have, size_t doubles) {
  if (doubles <= *have) return BRBD_OK;
  if (*buf) HIP_TRY(hipHostFree(*buf));
  *buf = nullptr;
  *have = 0;
  HIP_TRY(hipHostMalloc((void**)buf, doubles * sizeof(double), hipHostMallocDefault));
  *have = doubles;
  return BRBD_OK;
}

int64_t host_chunk(int64_t B) {
  static const int64_t forced = getenv("BRBD_HOST_CHUNK") ? atoll(getenv("BRBD_HOST_CHUNK")) : 0;
  if (forced > 0) return ((forced + WAVE - 1) / WAVE) * WAVE;
  int64_t c = ((B / 4 + WAVE - 1) / WAVE) * WAVE;  // >= 4 chunks in flight for a large batch
  if (c < 4096) c = 4096;
  if (c > 32768) c = 32768;
  return c;
}

// launch(Bc, dev, ldd, stream): run the kernels on one chunk; dev[i] / ldd[i] = device image of mats[i]
template <typename Launch>
int host_pipeline(brbd_model* m, int64_t B, int layout, const HostMat* mats, int nm, hipStream_t user,
                  Launch launch) {
  StreamCtx* ctx = get_ctx(m, user);
  int64_t C = host_chunk(B) < B ? host_chunk(B) : ((B + WAVE - 1) / WAVE) * WAVE;
  {  // at most 128 MiB per staging slot (the derivative matrices: 3 nv^2 doubles per instance)
    int64_t tot = 0;
    for (int i = 0; i < nm; ++i) tot += mats[i].dof;
    int64_t cap = ((int64_t)(128u << 20) / 8 / (tot > 0 ? tot : 1) / WAVE) * WAVE;
    if (cap < 256) cap = 256;
    if (C > cap) C = cap;
  }
  const int64_t nchunks = (B + C - 1) / C;
  size_t off[16], slot = 0;
  for (int i = 0; i < nm; ++i) {
    off[i] = slot;
    slot += padded((size_t)mats[i].dof * (size_t)C);
  }
  int rc = ensure_buf(&ctx->d_stage, &ctx->stage_doubles, (nchunks > 1 ? 2 : 1) * slot, user);
  if (rc) return rc;
  for (int k = 0; k < 2; ++k) {
    if (!ctx->pipe[k]) HIP_TRY(hipStreamCreateWithFlags(&ctx->pipe[k], hipStreamNonBlocking));
    if (!ctx->ev_done[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_done[k], hipEventDisableTiming));
  }
  if (!ctx->ev_in) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming));
  // Whatever way this call ends, nothing of it may still be in flight when it returns: an error path
  // would otherwise leave queued copies into the caller's matrices (or the pinned bounce slots the
  // next call reuses) running after the BRBD_HOST call has reported its status.
  struct Drain {
    StreamCtx* c;
    ~Drain() {
      for (int k = 0; k < 2; ++k)
        if (c->pipe[k]) (void)hipStreamSynchronize(c->pipe[k]);
    }
  } drain{ctx};
  // order the pipeline after what the caller already queued on its stream
  HIP_TRY(hipEventRecord(ctx->ev_in, user));
  for (int k = 0; k < 2; ++k) HIP_TRY(hipStreamWaitEvent(ctx->pipe[k], ctx->ev_in, 0));

  const bool bm = layout == BRBD_BATCH_MAJOR;
  auto copy = [&](int i, int64_t c0, int64_t Bc, double* dev, bool to_device, hipStream_t st) -> hipError_t {
    const HostMat& M = mats[i];
    // host block: columns [c0, c0 + Bc); device image tight (ld = dof, or C in BATCH_MAJOR)
    const size_t spitch = (size_t)M.ld * 8, dpitch = (size_t)(bm ? C : M.dof) * 8;
    const size_t width = (size_t)(bm ? Bc : M.dof) * 8, height = (size_t)(bm ? M.dof : Bc);
    const bool flat = spitch == width && dpitch == width;  // tight on both sides: one contiguous block
    if (to_device) {
      const double* h = M.src + (bm ? c0 : c0 * M.ld);
      if (flat) return hipMemcpyAsync(dev, h, width * height, hipMemcpyHostToDevice, st);
      return hipMemcpy2DAsync(dev, dpitch, h, spitch, width, height, hipMemcpyHostToDevice, st);
    }
    double* h = M.dst + (bm ? c0 : c0 * M.ld);
    if (flat) return hipMemcpyAsync(h, dev, width * height, hipMemcpyDeviceToHost, st);
    return hipMemcpy2DAsync(h, spitch, dev, dpitch, width, height, hipMemcpyDeviceToHost, st);
  };
  auto stage_in = [&](int64_t k) -> int {  // upload + kernels of chunk k
    const int sl = (int)(k & 1);
    const int64_t c0 = k * C, Bc = B - c0 < C ? B - c0 : C;
    double* base = ctx->d_stage + (size_t)sl * slot;
    double* dev[16];
    int64_t ldd[16];
    for (int i = 0; i < nm; ++i) {
      dev[i] = base + off[i];
      ldd[i] = bm ? C : mats[i].dof;
      if (mats[i].up) HIP_TRY(copy(i, c0, Bc, dev[i], true, ctx->pipe[sl]));
    }
    return launch(Bc, dev, ldd, ctx->pipe[sl]);
  };
  auto stage_out = [&](int64_t k) -> int {
    const int sl = (int)(k & 1);
    const int64_t c0 = k * C, Bc = B - c0 < C ? B - c0 : C;
    double* base = ctx->d_stage + (size_t)sl * slot;
    for (int i = 0; i < nm; ++i)
      if (mats[i].down) HIP_TRY(copy(i, c0, Bc, base + off[i], false, ctx->pipe[sl]));
    return BRBD_OK;
  };
  // pageable host memory, enough of it: the library's own pinned bounce buffers (see CopyPool)
  bool pageable = false;
  size_t total_bytes = 0;
  for (int i = 0; i < nm; ++i) {
    if (mats[i].up && !is_device_visible_host(mats[i].src)) pageable = true;
    if (mats[i].down && !is_device_visible_host(mats[i].dst)) pageable = true;
    total_bytes += (size_t)mats[i].dof * (size_t)B * 8 * ((mats[i].up ? 1 : 0) + (mats[i].down ? 1 : 0));
  }
  if (pageable && CopyPool::get().threads() >= 1 && total_bytes >= (size_t)(4u << 20)) {
    if ((rc = ensure_host_buf(&ctx->h_stage, &ctx->hstage_doubles, (nchunks > 1 ? 2 : 1) * slot))) return rc;
    std::vector<CopyTask> tasks;
    auto geometry = [&](int i, int64_t Bc, size_t& hpitch, size_t& ipitch, size_t& width, size_t& height) {
      hpitch = (size_t)mats[i].ld * 8;                 // the caller's matrix
      ipitch = (size_t)(bm ? C : mats[i].dof) * 8;     // its tight image (pinned and device alike)
      width = (size_t)(bm ? Bc : mats[i].dof) * 8;
      height = (size_t)(bm ? mats[i].dof : Bc);
    };
    auto pack = [&](int64_t k) {  // caller's columns of chunk k -> pinned image
      const int sl = (int)(k & 1);
      const int64_t c0 = k * C, Bc = B - c0 < C ? B - c0 : C;
      tasks.clear();
      for (int i = 0; i < nm; ++i) {
        if (!mats[i].up || !mats[i].dof) continue;
        size_t hp, ip, w, h;
        geometry(i, Bc, hp, ip, w, h);
        add_copy_tasks(tasks, ctx->h_stage + (size_t)sl * slot + off[i], ip, mats[i].src + (bm ? c0 : c0 * mats[i].ld), hp, w, h);
      }
      CopyPool::get().run(tasks);
    };
    auto submit = [&](int64_t k) -> int {  // pinned image -> device, kernels, results -> pinned image
      const int sl = (int)(k & 1);
      const int64_t c0 = k * C, Bc = B - c0 < C ? B - c0 : C;
      double* dbase = ctx->d_stage + (size_t)sl * slot;
      double* hbase = ctx->h_stage + (size_t)sl * slot;
      double* dev[16];
      int64_t ldd[16];
      for (int i = 0; i < nm; ++i) {
        dev[i] = dbase + off[i];
        ldd[i] = bm ? C : mats[i].dof;
        if (!mats[i].dof) continue;
        size_t hp, ip, w, h;
        geometry(i, Bc, hp, ip, w, h);
        if (mats[i].up) HIP_TRY(hipMemcpyAsync(dev[i], hbase + off[i], ip * h, hipMemcpyHostToDevice, ctx->pipe[sl]));
      }
      int r = launch(Bc, dev, ldd, ctx->pipe[sl]);
      if (r) return r;
      for (int i = 0; i < nm; ++i) {
        if (!mats[i].down || !mats[i].dof) continue;
        size_t hp, ip, w, h;
        geometry(i, Bc, hp, ip, w, h);
        HIP_TRY(hipMemcpyAsync(hbase + off[i], dev[i], ip * h, hipMemcpyDeviceToHost, ctx->pipe[sl]));
      }
      HIP_TRY(hipEventRecord(ctx->ev_done[sl], ctx->pipe[sl]));
      return BRBD_OK;
    };
    auto unpack = [&](int64_t k) -> int {  // pinned image -> caller's columns, once the chunk is done
      const int sl = (int)(k & 1);
      const int64_t c0 = k * C, Bc = B - c0 < C ? B - c0 : C;
      HIP_TRY(hipEventSynchronize(ctx->ev_done[sl]));
      tasks.clear();
      for (int i = 0; i < nm; ++i) {
        if (!mats[i].down || !mats[i].dof) continue;
        size_t hp, ip, w, h;
        geometry(i, Bc, hp, ip, w, h);
        add_copy_tasks(tasks, mats[i].dst + (bm ? c0 : c0 * mats[i].ld), hp, ctx->h_stage + (size_t)sl * slot + off[i], ip, w, h);
      }
      CopyPool::get().run(tasks);
      return BRBD_OK;
    };
    pack(0);
    if ((rc = submit(0))) return rc;
    for (int64_t k = 0; k < nchunks; ++k) {
      if (k + 1 < nchunks) {  // (slot (k+1)&1 was unpacked in the previous iteration)
        pack(k + 1);
        if ((rc = submit(k + 1))) return rc;
      }
      if ((rc = unpack(k))) return rc;
    }
    for (int k = 0; k < 2; ++k) HIP_TRY(hipStreamSynchronize(ctx->pipe[k]));
    return BRBD_OK;
  }
  // software pipeline: the upload of chunk k + 1 is queued before the download of chunk k, so that a
  // host thread blocked in a pageable copy still leaves the GPU with work
  if ((rc = stage_in(0))) return rc;
  for (int64_t k = 0; k < nchunks; ++k) {
    if (k + 1 < nchunks && (rc = stage_in(k + 1))) return rc;
    if ((rc = stage_out(k))) return rc;
  }
  for (int k = 0; k < 2; ++k) {
    HIP_TRY(hipEventRecord(ctx->ev_done[k], ctx->pipe[k]));
    HIP_TRY(hipStreamWaitEvent(user, ctx->ev_done[k], 0));
  }
  for (int k = 0; k < 2; ++k) HIP_TRY(hipStreamSynchronize(ctx->pipe[k]));
  return BRBD_OK;
}

}  // namespace
}  // extern "C++"

// rnea with (fext != NULL) or without external forces
static int rnea_impl(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
                     const double* a, int64_t lda, const double* fext, int64_t ldf, bool with_fext,
                     double* tau, int64_t ldtau, int layout, int memspace, void* stream) {
  int rc = check_common(m, B, layout, memspace);
  if (rc) return rc;
  const FlatModel& F = m->flat;
  if ((rc = check_mat("q", q, ldq, F.nq, B, layout))) return rc;
  if ((rc = check_mat("v", v, ldv, F.nv, B, layout))) return rc;
  if ((rc = check_mat("a", a, lda, F.nv, B, layout))) return rc;
  if (with_fext && (rc = check_mat("fext", fext, ldf, 6 * (int64_t)F.njoints, B, layout))) return rc;
  if ((rc = check_mat("tau", tau, ldtau, F.nv, B, layout))) return rc;
  if (B == 0) return BRBD_OK;
  DevGuard dg(m->device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  hipStream_t s = (hipStream_t)stream;
  if (memspace == BRBD_DEVICE) {
    KArgs A{};
    fill_common(m, A, B);
    A.ldfext = ldf;
    A.q = q; A.ldq = ldq; A.v = v; A.ldv = ldv; A.a = a; A.lda = lda; A.out = tau; A.ldout = ldtau;
    A.fext = with_fext ? fext : nullptr;
    return launch_algo(m, BRBD_ALGO_RNEA, A, layout, s);
  }
  const HostMat mats[5] = {{q, nullptr, ldq, F.nq, true, false}, {v, nullptr, ldv, F.nv, true, false},
                           {a, nullptr, lda, F.nv, true, false}, {nullptr, tau, ldtau, F.nv, false, true},
                           {fext, nullptr, ldf, 6 * (int64_t)F.njoints, true, false}};
  return host_pipeline(m, B, layout, mats, with_fext ? 5 : 4, s,
                       [&](int64_t Bc, double** d, const int64_t* ld, hipStream_t st) -> int {
                         KArgs A{};
                         fill_common(m, A, Bc);
                         A.q = d[0]; A.ldq = ld[0]; A.v = d[1]; A.ldv = ld[1]; A.a = d[2]; A.lda = ld[2];
                         A.out = d[3]; A.ldout = ld[3];
                         A.fext = with_fext ? d[4] : nullptr; A.ldfext = with_fext ? ld[4] : 0;
                         return launch_algo(m, BRBD_ALGO_RNEA, A, layout, st);
                       });
}

int brbd_rnea(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
              const double* a, int64_t lda, double* tau, int64_t ldtau, int layout, int memspace,
              void* stream) {
  return rnea_impl(m, B, q, ldq, v, ldv, a, lda, nullptr, 0, false, tau, ldtau, layout, memspace, stream);
}

int brbd_rnea_fext(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
                   const double* a, int64_t lda, const double* fext, int64_t ldfext, double* tau,
                   int64_t ldtau, int layout, int memspace, void* stream) {
  return rnea_impl(m, B, q, ldq, v, ldv, a, lda, fext, ldfext, true, tau, ldtau, layout, memspace, stream);
}

static int aba_check(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
                     const double* tau, int64_t ldtau, const double* fext, int64_t ldf, bool with_fext,
                     double* ddq, int64_t ldddq, int layout, int memspace) {
  int rc = check_common(m, B, layout, memspace);
  if (rc) return rc;
  const FlatModel& F = m->flat;
  if ((rc = check_mat("q", q, ldq, F.nq, B, layout))) return rc;
  if ((rc = check_mat("v", v, ldv, F.nv, B, layout))) return rc;
  // the reference forgets this check (algorithm/parallel/aba.hpp:62-69 tests `a` only); we add it
  if ((rc = check_mat("tau", tau, ldtau, F.nv, B, layout))) return rc;
  if (with_fext && (rc = check_mat("fext", fext, ldf, 6 * (int64_t)F.njoints, B, layout))) return rc;
  return check_mat("a", ddq, ldddq, F.nv, B, layout);
}

static int aba_impl(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
                    const double* tau, int64_t ldtau, const double* fext, int64_t ldf, bool with_fext,
                    double* ddq, int64_t ldddq, int layout, int memspace, void* stream) {
  int rc = aba_check(m, B, q, ldq, v, ldv, tau, ldtau, fext, ldf, with_fext, ddq, ldddq, layout, memspace);
  if (rc) return rc;
  const FlatModel& F = m->flat;
  if (B == 0) return BRBD_OK;
  DevGuard dg(m->device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  hipStream_t s = (hipStream_t)stream;
  if (memspace == BRBD_DEVICE) {
    KArgs A{};
    fill_common(m, A, B);
    A.ldfext = ldf;
    A.q = q; A.ldq = ldq; A.v = v; A.ldv = ldv; A.a = tau; A.lda = ldtau; A.out = ddq; A.ldout = ldddq;
    A.fext = with_fext ? fext : nullptr;
    return launch_algo(m, BRBD_ALGO_ABA, A, layout, s);
  }
  const HostMat mats[5] = {{q, nullptr, ldq, F.nq, true, false}, {v, nullptr, ldv, F.nv, true, false},
                           {tau, nullptr, ldtau, F.nv, true, false}, {nullptr, ddq, ldddq, F.nv, false, true},
                           {fext, nullptr, ldf, 6 * (int64_t)F.njoints, true, false}};
  return host_pipeline(m, B, layout, mats, with_fext ? 5 : 4, s,
                       [&](int64_t Bc, double** d, const int64_t* ld, hipStream_t st) -> int {
                         KArgs A{};
                         fill_common(m, A, Bc);
                         A.q = d[0]; A.ldq = ld[0]; A.v = d[1]; A.ldv = ld[1]; A.a = d[2]; A.lda = ld[2];
                         A.out = d[3]; A.ldout = ld[3];
                         A.fext = with_fext ? d[4] : nullptr; A.ldfext = with_fext ? ld[4] : 0;
                         return launch_algo(m, BRBD_ALGO_ABA, A, layout, st);
                       });
}

int brbd_aba(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
             const double* tau, int64_t ldtau, double* ddq, int64_t ldddq, int layout, int memspace,
             void* stream) {
  return aba_impl(m, B, q, ldq, v, ldv, tau, ldtau, nullptr, 0, false, ddq, ldddq, layout, memspace, stream);
}

int brbd_aba_fext(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
                  const double* tau, int64_t ldtau, const double* fext, int64_t ldfext, double* ddq,
                  int64_t ldddq, int layout, int memspace, void* stream) {
  return aba_impl(m, B, q, ldq, v, ldv, tau, ldtau, fext, ldfext, true, ddq, ldddq, layout, memspace, stream);
}

int brbd_fk_jacobians(brbd_model* m, int64_t B, const double* q, int64_t ldq, double* J, int64_t ldJ,
                      double* oMi, int64_t ldoMi, int layout, int memspace, void* stream) {
  int rc = check_common(m, B, layout, memspace);
  if (rc) return rc;
  const FlatModel& F = m->flat;
  if ((rc = check_mat("q", q, ldq, F.nq, B, layout))) return rc;
  if (!J && !oMi) return fail(BRBD_ERR_INVALID_ARGUMENT, "both outputs are NULL");
  const int64_t jd = 6 * (int64_t)F.nv, md = 12 * (int64_t)F.njoints;
  if (J && (rc = check_mat("J", J, ldJ, jd, B, layout))) return rc;
  if (oMi && (rc = check_mat("oMi", oMi, ldoMi, md, B, layout))) return rc;
  if (B == 0) return BRBD_OK;
  DevGuard dg(m->device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  hipStream_t s = (hipStream_t)stream;
  if (memspace == BRBD_DEVICE) {
    KArgs A{};
    fill_common(m, A, B);
    A.q = q; A.ldq = ldq; A.out = J; A.ldout = ldJ; A.out2 = oMi; A.ldout2 = ldoMi;
    return launch_algo(m, BRBD_ALGO_FK_JAC, A, layout, s);
  }
  const HostMat mats[3] = {{q, nullptr, ldq, F.nq, true, false}, {nullptr, J, ldJ, J ? jd : 0, false, J != nullptr},
                           {nullptr, oMi, ldoMi, oMi ? md : 0, false, oMi != nullptr}};
  return host_pipeline(m, B, layout, mats, 3, s, [&](int64_t Bc, double** d, const int64_t* ld, hipStream_t st) -> int {
    KArgs A{};
    fill_common(m, A, Bc);
    A.q = d[0]; A.ldq = ld[0];
    A.out = J ? d[1] : nullptr; A.ldout = ld[1];
    A.out2 = oMi ? d[2] : nullptr; A.ldout2 = ld[2];
    return launch_algo(m, BRBD_ALGO_FK_JAC, A, layout, st);
  });
}

int brbd_joint_jacobian(brbd_model* m, int64_t B, const double* q, int64_t ldq, int joint, int rf,
                        double* J, int64_t ldJ, int layout, int memspace, void* stream) {
  int rc = check_common(m, B, layout, memspace);
  if (rc) return rc;
  const FlatModel& F = m->flat;
  if (joint < 1 || joint >= F.njoints) return fail(BRBD_ERR_INVALID_ARGUMENT, "joint index out of range");
  if (rf < 0 || rf > 2) return fail(BRBD_ERR_INVALID_ARGUMENT, "rf must be 0 (WORLD), 1 (LOCAL) or 2 (LOCAL_WORLD_ALIGNED)");
  const int64_t jd = 6 * (int64_t)F.nv;
  if ((rc = check_mat("q", q, ldq, F.nq, B, layout))) return rc;
  if ((rc = check_mat("J", J, ldJ, jd, B, layout))) return rc;
  if (B == 0) return BRBD_OK;
  DevGuard dg(m->device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  hipStream_t s = (hipStream_t)stream;
  // support chain of the joint, root first, as forward ops (cached per joint)
  Op* d_chain = nullptr;
  int nchain = 0;
  {
    std::lock_guard<std::mutex> g(m->mu);
    auto it = m->chains.find(joint);
    if (it == m->chains.end()) {
      std::vector<Op> chain;
      for (int i = joint; i > 0; i = F.tab[i].parent) {
        Op op{};
        for (const Op& o : F.prog3)
          if (o.joint == i) op = o;
        chain.insert(chain.begin(), op);
      }
      Op* d = nullptr;
      HIP_TRY(hipMalloc(&d, chain.size() * sizeof(Op)));
      HIP_TRY(hipMemcpy(d, chain.data(), chain.size() * sizeof(Op), hipMemcpyHostToDevice));
      it = m->chains.emplace(joint, std::make_pair(d, (int)chain.size())).first;
    }
    d_chain = it->second.first;
    nchain = it->second.second;
  }
  auto run = [&](int64_t Bc, const double* dq, int64_t ldq_, double* dJ, int64_t ldJ_, hipStream_t st) -> int {
    if (m->spec && m->use_spec && F.njoints <= 64) {  // model-compiled kernel (codegen/jjac.py)
      KArgs K{};
      fill_common(m, K, Bc);
      K.rf = rf;
      K.q = dq; K.ldq = ldq_; K.out = dJ; K.ldout = ldJ_;
      for (int i = joint; i > 0; i = F.tab[i].parent) K.chain |= 1ull << i;
      const int r = launch_algo(m, BRBD_ALGO_JOINT_JAC, K, layout, st);
      if (r <= 0) return r;  // 1 = no compiled kernel applies
    }
    KArgs A{};
    fill_common(m, A, Bc);
    A.prog3 = d_chain;
    A.nops3 = nchain;
    A.rf = rf;
    if (F.nv <= 64)  // dofs of the support chain: the table-driven kernel zeroes only the other columns
      for (int i = joint; i > 0; i = F.tab[i].parent)
        for (int k = 0; k < F.tab[i].nvj; ++k) A.chain |= 1ull << (F.tab[i].idx_v + k);
    A.q = dq; A.ldq = ldq_; A.out = dJ; A.ldout = ldJ_;
    if (m->max_waves_jjac.load() == 0) m->max_waves_jjac.store(resident_waves(m, jjac_kernel<1>, WAVE));
    const int64_t mwj = m->max_waves_jjac.load();
    const int64_t grid = A.ntiles < mwj ? A.ntiles : mwj;
    const dim3 g((unsigned)grid), b(WAVE);
    Timer tm{m, BRBD_ALGO_JOINT_JAC, st};
    int rt = tm.begin();
    if (rt) return rt;
    if (layout == BRBD_BATCH_MAJOR) hipLaunchKernelGGL(jjac_kernel<1>, g, b, 0, st, A);
    else hipLaunchKernelGGL(jjac_kernel<0>, g, b, 0, st, A);
    HIP_TRY(hipGetLastError());
    return tm.end();
  };
  if (memspace == BRBD_DEVICE) return run(B, q, ldq, J, ldJ, s);
  const HostMat mats[2] = {{q, nullptr, ldq, F.nq, true, false}, {nullptr, J, ldJ, jd, false, true}};
  return host_pipeline(m, B, layout, mats, 2, s, [&](int64_t Bc, double** d, const int64_t* ld, hipStream_t st) -> int {
    return run(Bc, d[0], ld[0], d[1], ld[1], st);
  });
}

// computeRNEADerivatives on device matrices: in[0..2] = q, v, a; out[0..3] = dtau_dq, dtau_dv, dtau_da, tau
// (the last two may be NULL).  prezeroed: -1 = the handle's output contract, 0 / 1 = this call's.
extern "C++" {
namespace {
int deriv_device(brbd_model* m, int64_t Bc, const double* const* in, const int64_t* ldi, double* const* out,
                 const int64_t* ldo, hipStream_t st, int layout, int pz) {
  const FlatModel& F = m->flat;
  if (m->spec && m->use_spec) {  // model-compiled kernel (codegen/deriv.py), when registered
    KArgs A{};
    fill_common(m, A, Bc);
    A.q = in[0]; A.ldq = ldi[0]; A.v = in[1]; A.ldv = ldi[1]; A.a = in[2]; A.lda = ldi[2];
    A.out = out[0]; A.ldout = ldo[0]; A.out2 = out[1]; A.ldout2 = ldo[1]; A.out3 = out[2]; A.ldout3 = ldo[2];
    A.out4 = out[3]; A.ldout4 = ldo[3];
    const int r = launch_algo(m, BRBD_ALGO_RNEA_DERIV, A, layout, st, nullptr, pz);
    if (r <= 0) return r;  // 1 = no compiled kernel applies
  }
  DerivArgs D{};
  D.tab = m->d_tab;
  D.nvsub = m->d_nvsub;
  D.njoints = F.njoints; D.nq = F.nq; D.nv = F.nv;
  D.B = Bc;
  D.ntiles = (Bc + WAVE - 1) / WAVE;
  D.gx = F.gravity[0]; D.gy = F.gravity[1]; D.gz = F.gravity[2];
  D.q = in[0]; D.ldq = ldi[0]; D.v = in[1]; D.ldv = ldi[1]; D.a = in[2]; D.lda = ldi[2];
  D.dq = out[0]; D.lddq = ldo[0]; D.dv = out[1]; D.lddv = ldo[1]; D.da = out[2]; D.ldda = ldo[2];
  D.tau = out[3]; D.ldtau = ldo[3];
  if (m->max_waves[BRBD_ALGO_RNEA_DERIV].load() == 0)
    m->max_waves[BRBD_ALGO_RNEA_DERIV].store(resident_waves(m, rnea_deriv_kernel<1>, WAVE));
  const int64_t mwd = m->max_waves[3].load();
  const int64_t grid = D.ntiles < mwd ? D.ntiles : mwd;
  D.ws_stride = (int64_t)deriv_ws_per_lane(F) * WAVE;
  StreamCtx* ctx = get_ctx(m, st);
  int r = ensure_buf(&ctx->d_ws, &ctx->ws_doubles, (size_t)grid * D.ws_stride, st);
  if (r) return r;
  D.ws = ctx->d_ws;
  Timer tm{m, BRBD_ALGO_RNEA_DERIV, st};
  if ((r = tm.begin())) return r;
  const dim3 g((unsigned)grid), b(WAVE);
  if (layout == BRBD_BATCH_MAJOR) hipLaunchKernelGGL(rnea_deriv_kernel<1>, g, b, 0, st, D);
  else hipLaunchKernelGGL(rnea_deriv_kernel<0>, g, b, 0, st, D);
  HIP_TRY(hipGetLastError());
  return tm.end();
}
}  // namespace
}  // extern "C++"

int brbd_rnea_derivatives(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v,
                          int64_t ldv, const double* a, int64_t lda, double* dtau_dq, int64_t lddq,
                          double* dtau_dv, int64_t lddv, double* dtau_da, int64_t ldda, double* tau,
                          int64_t ldtau, int layout, int memspace, void* stream) {
  int rc = check_common(m, B, layout, memspace);
  if (rc) return rc;
  const FlatModel& F = m->flat;
  const int64_t nn = (int64_t)F.nv * F.nv;
  if ((rc = check_mat("q", q, ldq, F.nq, B, layout))) return rc;
  if ((rc = check_mat("v", v, ldv, F.nv, B, layout))) return rc;
  if ((rc = check_mat("a", a, lda, F.nv, B, layout))) return rc;
  if ((rc = check_mat("dtau_dq", dtau_dq, lddq, nn, B, layout))) return rc;
  if ((rc = check_mat("dtau_dv", dtau_dv, lddv, nn, B, layout))) return rc;
  if (dtau_da && (rc = check_mat("dtau_da", dtau_da, ldda, nn, B, layout))) return rc;
  if (tau && (rc = check_mat("tau", tau, ldtau, F.nv, B, layout))) return rc;
  if (!F.depth_first)
    return fail(BRBD_ERR_UNSUPPORTED,
                "computeRNEADerivatives needs depth-first joint numbering (contiguous subtrees), as "
                "pinocchio's Data::computeNvSubtree assumes");
  if (B == 0) return BRBD_OK;
  DevGuard dg(m->device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  hipStream_t s = (hipStream_t)stream;
  auto run = [&](int64_t Bc, const double* const* in, const int64_t* ldi, double* const* out, const int64_t* ldo,
                 hipStream_t st, int lay, int pz) -> int { return deriv_device(m, Bc, in, ldi, out, ldo, st, lay, pz); };
  if (memspace == BRBD_DEVICE) {
    const double* in[3] = {q, v, a};
    const int64_t ldi[3] = {ldq, ldv, lda};
    double* out[4] = {dtau_dq, dtau_dv, dtau_da, tau};
    const int64_t ldo[4] = {lddq, lddv, ldda, ldtau};
    if (layout == BRBD_INSTANCE_MAJOR && B >= 4096 && via_bm_enabled()) {
      // the reference's own layout: through the BATCH_MAJOR kernel (see via_batch_major).  The
      // copies are not pre-zeroed, so the kernel variant that writes the structural zeros runs.
      const XpMat mats[7] = {{q, nullptr, ldq, F.nq}, {v, nullptr, ldv, F.nv}, {a, nullptr, lda, F.nv},
                             {nullptr, dtau_dq, lddq, nn}, {nullptr, dtau_dv, lddv, nn},
                             {nullptr, dtau_da, ldda, dtau_da ? nn : 0}, {nullptr, tau, ldtau, tau ? F.nv : 0}};
      return via_batch_major(m, B, mats, 7, s, [&](int64_t Bc, double** d, const int64_t* ld, hipStream_t st) -> int {
        const double* bin[3] = {d[0], d[1], d[2]};
        double* bout[4] = {d[3], d[4], dtau_da ? d[5] : nullptr, tau ? d[6] : nullptr};
        return run(Bc, bin, ld, bout, ld + 3, st, BRBD_BATCH_MAJOR, 0);
      });
    }
    return run(B, in, ldi, out, ldo, s, layout, -1);
  }
  // under the prezeroed contract the kernel leaves the structural zeros alone: the caller's zeros
  // must be on the device too, so those outputs are uploaded as well
  const bool pz = m->prezeroed;
  const HostMat mats[7] = {{q, nullptr, ldq, F.nq, true, false}, {v, nullptr, ldv, F.nv, true, false},
                           {a, nullptr, lda, F.nv, true, false},
                           {dtau_dq, dtau_dq, lddq, nn, pz, true}, {dtau_dv, dtau_dv, lddv, nn, pz, true},
                           {dtau_da, dtau_da, ldda, dtau_da ? nn : 0, pz && dtau_da, dtau_da != nullptr},
                           {nullptr, tau, ldtau, tau ? F.nv : 0, false, tau != nullptr}};
  return host_pipeline(m, B, layout, mats, 7, s, [&](int64_t Bc, double** d, const int64_t* ld, hipStream_t st) -> int {
    const double* in[3] = {d[0], d[1], d[2]};
    double* out[4] = {d[3], d[4], dtau_da ? d[5] : nullptr, tau ? d[6] : nullptr};
    return run(Bc, in, ld, out, ld + 3, st, layout, -1);
  });
}

// ---- integrate, computeMinverse, computeABADerivatives ---------------------------
extern "C++" {
namespace {
void fill_f4(const brbd_model* m, F4Args& A, int64_t B) {
  const FlatModel& F = m->flat;
  A.tab = m->d_tab;
  A.prog3 = m->d_prog3;
  A.nvsub = m->d_nvsub;
  A.nops3 = (int)F.prog3.size();
  A.njoints = F.njoints; A.nq = F.nq; A.nv = F.nv; A.maxdepth = F.maxdepth;
  A.B = B;
  A.ntiles = (B + WAVE - 1) / WAVE;
}
template <typename K>
int f4_grid(brbd_model* m, std::atomic<int>& cache, K kernel, int64_t ntiles) {
  int w = cache.load(std::memory_order_relaxed);
  if (w == 0) {
    w = resident_waves(m, kernel, WAVE);
    cache.store(w, std::memory_order_relaxed);
  }
  return (int)(ntiles < w ? ntiles : w);
}
int minverse_device(brbd_model* m, int64_t B, const double* q, int64_t ldq, double* Minv, int64_t ldm, int full,
                    int layout, hipStream_t st) {
  const FlatModel& F = m->flat;
  if (m->spec && m->use_spec && layout == BRBD_BATCH_MAJOR) {  // model-compiled kernel (codegen/minv.py)
    KArgs K{};
    fill_common(m, K, B);
    K.q = q; K.ldq = ldq; K.out = Minv; K.ldout = ldm;
    const int r = launch_algo(m, BRBD_ALGO_MINVERSE, K, layout, st, nullptr, full ? 1 : 0);
    if (r <= 0) return r;  // 1 = no compiled kernel applies
  }
  F4Args A{};
  fill_f4(m, A, B);
  A.full = full;
  A.q = q; A.ldq = ldq; A.out = Minv; A.ldout = ldm;
  const int grid = f4_grid(m, m->max_waves_f4[0], minverse_kernel<1>, A.ntiles);
  A.ws_stride = (int64_t)minverse_ws_per_lane(F) * WAVE;
  StreamCtx* ctx = get_ctx(m, st);
  int rc = ensure_buf(&ctx->d_ws, &ctx->ws_doubles, (size_t)grid * A.ws_stride, st);
  if (rc) return rc;
  A.ws = ctx->d_ws;
  Timer tm{m, BRBD_ALGO_MINVERSE, st};
  if ((rc = tm.begin())) return rc;
  if (layout == BRBD_BATCH_MAJOR) hipLaunchKernelGGL(minverse_kernel<1>, dim3(grid), dim3(WAVE), 0, st, A);
  else hipLaunchKernelGGL(minverse_kernel<0>, dim3(grid), dim3(WAVE), 0, st, A);
  HIP_TRY(hipGetLastError());
  return tm.end();
}
}  // namespace
}  // extern "C++"

int brbd_integrate(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv, double* qout,
                   int64_t ldqout, int layout, int memspace, void* stream) {
  int rc = check_common(m, B, layout, memspace);
  if (rc) return rc;
  const FlatModel& F = m->flat;
  if ((rc = check_mat("q", q, ldq, F.nq, B, layout))) return rc;
  if ((rc = check_mat("v", v, ldv, F.nv, B, layout))) return rc;
  if ((rc = check_mat("qout", qout, ldqout, F.nq, B, layout))) return rc;
  if (B == 0) return BRBD_OK;
  DevGuard dg(m->device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  auto run = [&](int64_t Bc, const double* dq, int64_t l0, const double* dv, int64_t l1, double* dout, int64_t l2,
                 hipStream_t st) -> int {
    F4Args A{};
    fill_f4(m, A, Bc);
    A.q = dq; A.ldq = l0; A.v = dv; A.ldv = l1; A.out = dout; A.ldout = l2;
    const int grid = f4_grid(m, m->max_waves_f4[1], integrate_kernel<1>, A.ntiles);
    if (layout == BRBD_BATCH_MAJOR) hipLaunchKernelGGL(integrate_kernel<1>, dim3(grid), dim3(WAVE), 0, st, A);
    else hipLaunchKernelGGL(integrate_kernel<0>, dim3(grid), dim3(WAVE), 0, st, A);
    HIP_TRY(hipGetLastError());
    return BRBD_OK;
  };
  if (memspace == BRBD_DEVICE) return run(B, q, ldq, v, ldv, qout, ldqout, (hipStream_t)stream);
  const HostMat mats[3] = {{q, nullptr, ldq, F.nq, true, false}, {v, nullptr, ldv, F.nv, true, false},
                           {nullptr, qout, ldqout, F.nq, false, true}};
  return host_pipeline(m, B, layout, mats, 3, (hipStream_t)stream,
                       [&](int64_t Bc, double** d, const int64_t* ld, hipStream_t st) -> int {
                         return run(Bc, d[0], ld[0], d[1], ld[1], d[2], ld[2], st);
                       });
}

int brbd_minverse(brbd_model* m, int64_t B, const double* q, int64_t ldq, double* Minv, int64_t ldminv, int layout,
                  int memspace, void* stream) {
  int rc = check_common(m, B, layout, memspace);
  if (rc) return rc;
  const FlatModel& F = m->flat;
  const int64_t nn = (int64_t)F.nv * F.nv;
  if ((rc = check_mat("q", q, ldq, F.nq, B, layout))) return rc;
  if ((rc = check_mat("Minv", Minv, ldminv, nn, B, layout))) return rc;
  if (!F.depth_first)
    return fail(BRBD_ERR_UNSUPPORTED, "computeMinverse needs depth-first joint numbering (contiguous subtrees), as "
                                      "pinocchio's Data::computeNvSubtree assumes");
  if (B == 0) return BRBD_OK;
  DevGuard dg(m->device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  if (memspace == BRBD_DEVICE && layout == BRBD_INSTANCE_MAJOR && B >= 4096 && via_bm_enabled()) {
    const XpMat mats[2] = {{q, nullptr, ldq, F.nq}, {nullptr, Minv, ldminv, (int64_t)F.nv * F.nv}};
    return via_batch_major(m, B, mats, 2, (hipStream_t)stream,
                           [&](int64_t Bc, double** d, const int64_t* ld, hipStream_t st) -> int {
                             return minverse_device(m, Bc, d[0], ld[0], d[1], ld[1], 0, BRBD_BATCH_MAJOR, st);
                           });
  }
  if (memspace == BRBD_DEVICE) return minverse_device(m, B, q, ldq, Minv, ldminv, 0, layout, (hipStream_t)stream);
  const HostMat mats[2] = {{q, nullptr, ldq, F.nq, true, false}, {nullptr, Minv, ldminv, nn, false, true}};
  return host_pipeline(m, B, layout, mats, 2, (hipStream_t)stream,
                       [&](int64_t Bc, double** d, const int64_t* ld, hipStream_t st) -> int {
                         return minverse_device(m, Bc, d[0], ld[0], d[1], ld[1], 0, layout, st);
                       });
}

int brbd_aba_derivatives(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
                         const double* tau, int64_t ldtau, double* ddq_dq, int64_t lddq, double* ddq_dv, int64_t lddv,
                         double* ddq_dtau, int64_t lddtau, double* ddq, int64_t ldddq, int layout, int memspace,
                         void* stream) {
  int rc = check_common(m, B, layout, memspace);
  if (rc) return rc;
  const FlatModel& F = m->flat;
  const int64_t nn = (int64_t)F.nv * F.nv;
  if ((rc = check_mat("q", q, ldq, F.nq, B, layout))) return rc;
  if ((rc = check_mat("v", v, ldv, F.nv, B, layout))) return rc;
  if ((rc = check_mat("tau", tau, ldtau, F.nv, B, layout))) return rc;
  if ((rc = check_mat("ddq_dq", ddq_dq, lddq, nn, B, layout))) return rc;
  if ((rc = check_mat("ddq_dv", ddq_dv, lddv, nn, B, layout))) return rc;
  if ((rc = check_mat("ddq_dtau", ddq_dtau, lddtau, nn, B, layout))) return rc;
  if (ddq && (rc = check_mat("ddq", ddq, ldddq, F.nv, B, layout))) return rc;
  if (!F.depth_first)
    return fail(BRBD_ERR_UNSUPPORTED, "computeABADerivatives needs depth-first joint numbering (contiguous subtrees)");
  if (F.nv > 96) return fail(BRBD_ERR_UNSUPPORTED, "computeABADerivatives: models with nv > 96 are not supported yet");
  if (B == 0) return BRBD_OK;
  DevGuard dg(m->device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  // aba-derivatives.hxx:381-454 with the engine's own building blocks: ddq = aba(q, v, tau); the RNEA
  // partial derivatives at that acceleration (pass 4 of the reference); Minv, symmetrised (:449-450);
  // the closing products (:452-453).  All on one stream, in order.
  auto run = [&](int64_t Bc, const double* dq, int64_t l0, const double* dv, int64_t l1, const double* dt, int64_t l2,
                 double* o1, int64_t lo1, double* o2, int64_t lo2, double* o3, int64_t lo3, double* acc, int64_t la,
                 hipStream_t st, int layout) -> int {
    int r = brbd_aba(m, Bc, dq, l0, dv, l1, dt, l2, acc, la, layout, BRBD_DEVICE, st);
    if (r) return r;
    {  // every element written (prezeroed = 0): the products read both matrices in full
      const double* in[3] = {dq, dv, acc};
      const int64_t ldi[3] = {l0, l1, la};
      double* out[4] = {o1, o2, nullptr, nullptr};
      const int64_t ldo[4] = {lo1, lo2, 0, 0};
      if ((r = deriv_device(m, Bc, in, ldi, out, ldo, st, layout, 0))) return r;
    }
    if ((r = minverse_device(m, Bc, dq, l0, o3, lo3, 1, layout, st))) return r;
    if (m->spec && m->use_spec && layout == BRBD_BATCH_MAJOR) {  // model-compiled products (codegen/prod.py)
      KArgs K{};
      fill_common(m, K, Bc);
      K.out = o1; K.ldout = lo1; K.out2 = o2; K.ldout2 = lo2; K.out3 = o3; K.ldout3 = lo3;
      r = launch_algo(m, BRBD_ALGO_ABA_DERIV_PRODUCTS, K, layout, st);
      if (r <= 0) return r;  // 1 = no compiled kernel applies
    }
    F4Args A{};
    fill_f4(m, A, Bc);
    A.out = o1; A.ldout = lo1; A.out2 = o2; A.ldout2 = lo2; A.minv = o3; A.ldminv = lo3;
    // (4 columns of both factors in registers for nv <= 24, 2 for nv <= 48, 1 for nv <= 96: about 200 doubles)
    const int cls = F.nv <= 24 ? 0 : (F.nv <= 48 ? 1 : 2);
    const int grid = cls == 0 ? f4_grid(m, m->max_waves_f4[2], aba_deriv_product_kernel<1, 24, 4>, A.ntiles)
                   : cls == 1 ? f4_grid(m, m->max_waves_f4[3], aba_deriv_product_kernel<1, 48, 2>, A.ntiles)
                              : f4_grid(m, m->max_waves_f4[4], aba_deriv_product_kernel<1, 96, 1>, A.ntiles);
    Timer tm{m, BRBD_ALGO_ABA_DERIV_PRODUCTS, st};
    if ((r = tm.begin())) return r;
    const dim3 g_(grid), b_(WAVE);
    if (layout == BRBD_BATCH_MAJOR) {
      if (cls == 0) hipLaunchKernelGGL((aba_deriv_product_kernel<1, 24, 4>), g_, b_, 0, st, A);
      else if (cls == 1) hipLaunchKernelGGL((aba_deriv_product_kernel<1, 48, 2>), g_, b_, 0, st, A);
      else hipLaunchKernelGGL((aba_deriv_product_kernel<1, 96, 1>), g_, b_, 0, st, A);
    } else {
      if (cls == 0) hipLaunchKernelGGL((aba_deriv_product_kernel<0, 24, 4>), g_, b_, 0, st, A);
      else if (cls == 1) hipLaunchKernelGGL((aba_deriv_product_kernel<0, 48, 2>), g_, b_, 0, st, A);
      else hipLaunchKernelGGL((aba_deriv_product_kernel<0, 96, 1>), g_, b_, 0, st, A);
    }
    HIP_TRY(hipGetLastError());
    return tm.end();
  };
  if (memspace == BRBD_DEVICE && layout == BRBD_INSTANCE_MAJOR && B >= 4096 && via_bm_enabled()) {
    // every stage is output-bound: the whole composition runs on batch-major copies
    const XpMat mats[7] = {{q, nullptr, ldq, F.nq}, {v, nullptr, ldv, F.nv}, {tau, nullptr, ldtau, F.nv},
                           {nullptr, ddq_dq, lddq, nn}, {nullptr, ddq_dv, lddv, nn}, {nullptr, ddq_dtau, lddtau, nn},
                           {nullptr, ddq, ldddq, F.nv}};  // (ddq NULL: the copy is the temporary)
    return via_batch_major(m, B, mats, 7, (hipStream_t)stream,
                           [&](int64_t Bc, double** d, const int64_t* ld, hipStream_t st) -> int {
                             return run(Bc, d[0], ld[0], d[1], ld[1], d[2], ld[2], d[3], ld[3], d[4], ld[4], d[5],
                                        ld[5], d[6], ld[6], st, BRBD_BATCH_MAJOR);
                           });
  }
  if (memspace == BRBD_DEVICE) {
    hipStream_t st = (hipStream_t)stream;
    double* acc = ddq;
    int64_t la = ldddq;
    if (!acc) {  // the acceleration is needed anyway: a per-stream temporary
      StreamCtx* ctx = get_ctx(m, st);
      const int64_t ld = layout == BRBD_BATCH_MAJOR ? B : F.nv;
      if ((rc = ensure_buf(&ctx->d_tmp, &ctx->tmp_doubles, (size_t)F.nv * (size_t)B, st))) return rc;
      acc = ctx->d_tmp;
      la = ld;
    }
    return run(B, q, ldq, v, ldv, tau, ldtau, ddq_dq, lddq, ddq_dv, lddv, ddq_dtau, lddtau, acc, la, st, layout);
  }
  const HostMat mats[7] = {{q, nullptr, ldq, F.nq, true, false}, {v, nullptr, ldv, F.nv, true, false},
                           {tau, nullptr, ldtau, F.nv, true, false}, {nullptr, ddq_dq, lddq, nn, false, true},
                           {nullptr, ddq_dv, lddv, nn, false, true}, {nullptr, ddq_dtau, lddtau, nn, false, true},
                           {nullptr, ddq, ldddq, F.nv, false, ddq != nullptr}};
  return host_pipeline(m, B, layout, mats, 7, (hipStream_t)stream,
                       [&](int64_t Bc, double** d, const int64_t* ld, hipStream_t st) -> int {
                         return run(Bc, d[0], ld[0], d[1], ld[1], d[2], ld[2], d[3], ld[3], d[4], ld[4], d[5], ld[5],
                                    d[6], ld[6], st, layout);
                       });
}

// ---- no-alloc variant ---------------------------------------------------------------------------
// brbd_aba for callers that cannot allocate (HIP graph capture, serving loops, harnesses that
// forbid hipMalloc): device memory only, always the table-driven ABA kernel, whose workspace (the
// pass-2 -> pass-3 hand-off records and the part of the live-state stack that does not fit LDS)
// lives in the file-scope array below instead of a hipMalloc'ed buffer.  Nothing is allocated or
// freed here: no hipMalloc / hipFree, no events, no host containers.  Each GPU holds its own copy
// of the array (64 MiB) from module load to process exit; its address is resolved once per device
// in brbd_model_create (hipGetSymbolAddress) and kept in the handle.  Every call on a GPU shares
// that GPU's array: calls that overlap on one GPU (two streams, two handles) race -- serialise
// them.  Nothing in the array needs clearing: the kernel writes every slot before it reads it.
#define BRBD_NOALLOC_SCRATCH_DOUBLES (8u * 1024u * 1024u)
#define BRBD_NOALLOC_MAX_BATCH (1 << 20)
static __device__ __align__(256) double scratch[BRBD_NOALLOC_SCRATCH_DOUBLES];

static int resolve_scratch(brbd_model* m) {  // (device m->device is current)
  double* ptr = nullptr;
  HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
  if (!ptr) return fail(BRBD_ERR_RUNTIME, "hipGetSymbolAddress(scratch) returned NULL");
  m->scratch = ptr;
  return BRBD_OK;
}

int brbd_aba_noalloc(brbd_model* m, int64_t B, const double* q, int64_t ldq, const double* v, int64_t ldv,
                     const double* tau, int64_t ldtau, double* ddq, int64_t ldddq, int layout, void* stream) {
  int rc = aba_check(m, B, q, ldq, v, ldv, tau, ldtau, nullptr, 0, false, ddq, ldddq, layout, BRBD_DEVICE);
  if (rc) return rc;
  if (B > BRBD_NOALLOC_MAX_BATCH)
    return fail(BRBD_ERR_INVALID_ARGUMENT, "brbd_aba_noalloc: batch " + std::to_string(B) + " exceeds the maximum of " +
                                               std::to_string((int64_t)BRBD_NOALLOC_MAX_BATCH) + " instances per call");
  if (B == 0) return BRBD_OK;
  if (!m->scratch) return fail(BRBD_ERR_RUNTIME, "the scratch array of this device was not resolved at model creation");
  DevGuard dg(m->device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  KArgs A{};
  fill_common(m, A, B);
  A.q = q; A.ldq = ldq; A.v = v; A.ldv = ldv; A.a = tau; A.lda = ldtau; A.out = ddq; A.ldout = ldddq;
  const FixedWs fixed{m->scratch, (size_t)BRBD_NOALLOC_SCRATCH_DOUBLES};
  return launch_algo(m, BRBD_ALGO_ABA, A, layout, (hipStream_t)stream, &fixed);
}

int64_t brbd_debug_oob_count(brbd_model* m) {
#ifdef BRBD_SPEC_CHECK
  if (!m || !m->d_oob) return -1;
  DevGuard dg(m->device);
  unsigned long long n[5] = {0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(n, m->d_oob, sizeof n, hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  if (n[0])
    fprintf(stderr, "brbd check: %llu accessors with out-of-range sites; kinds (bit mask, brbd_spec.hpp) 0x%llx; last: "
                    "literal %lld where at most %lld fits\n", n[0], n[1], (long long)n[2], (long long)n[3]);
  return (int64_t)n[0];
#else
  (void)m;
  return -1;
#endif
}

int brbd_host_register(void* ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(BRBD_ERR_INVALID_ARGUMENT, "NULL pointer or zero size");
  // whole pages: the runtime rejects ranges that are not page aligned
  const uintptr_t page = 4096, lo = (uintptr_t)ptr & ~(page - 1), hi = ((uintptr_t)ptr + bytes + page - 1) & ~(page - 1);
  HIP_TRY(hipHostRegister((void*)lo, (size_t)(hi - lo), hipHostRegisterDefault));
  return BRBD_OK;
}

int brbd_host_unregister(void* ptr) {
  if (!ptr) return fail(BRBD_ERR_INVALID_ARGUMENT, "NULL pointer");
  HIP_TRY(hipHostUnregister((void*)((uintptr_t)ptr & ~(uintptr_t)4095)));
  return BRBD_OK;
}

static int transpose_common(int device, int64_t dof, int64_t B, const double* src, int64_t ld_src,
                            double* dst, int64_t ld_dst, void* stream, int to_bm) {
  if (!src || !dst) return fail(BRBD_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (dof < 0 || B < 0) return fail(BRBD_ERR_INVALID_ARGUMENT, "negative size");
  if (dof == 0 || B == 0) return BRBD_OK;
  const int64_t need_src = to_bm ? dof : B, need_dst = to_bm ? B : dof;
  if (ld_src < need_src || ld_dst < need_dst) return fail(BRBD_ERR_INVALID_ARGUMENT, "leading dimension too small");
  DevGuard dg(device);
  if (!dg.ok) return fail(BRBD_ERR_RUNTIME, "hipSetDevice failed");
  const dim3 g((unsigned)((B + 63) / 64)), b(256);
  if (to_bm) hipLaunchKernelGGL(xpose64_kernel<1>, g, b, 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, dof, B);
  else hipLaunchKernelGGL(xpose64_kernel<0>, g, b, 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, dof, B);
  HIP_TRY(hipGetLastError());
  return BRBD_OK;
}

int brbd_transpose_to_batch_major(int device, int64_t dof, int64_t B, const double* src, int64_t ld_src,
                                  double* dst, int64_t ld_dst, void* stream) {
  return transpose_common(device, dof, B, src, ld_src, dst, ld_dst, stream, 1);
}
int brbd_transpose_to_instance_major(int device, int64_t dof, int64_t B, const double* src,
                                     int64_t ld_src, double* dst, int64_t ld_dst, void* stream) {
  return transpose_common(device, dof, B, src, ld_src, dst, ld_dst, stream, 0);
}

const char* brbd_model_compiled_kernels(const brbd_model* m) {
  return (m && m->spec) ? m->spec->name : "";
}

int brbd_model_use_compiled(brbd_model* m, int enabled) {
  if (!m) return fail(BRBD_ERR_INVALID_ARGUMENT, "model handle is NULL");
  m->use_spec = enabled != 0;
  return BRBD_OK;
}

int brbd_model_outputs_prezeroed(brbd_model* m, int enabled) {
  if (!m) return fail(BRBD_ERR_INVALID_ARGUMENT, "model handle is NULL");
  m->prezeroed = enabled != 0;
  return BRBD_OK;
}

int brbd_set_timing(brbd_model* m, int enabled) {
  if (!m) return fail(BRBD_ERR_INVALID_ARGUMENT, "model handle is NULL");
  m->timing = (enabled == 2 || enabled == 3) ? enabled : (enabled != 0 ? 1 : 0);
  {
    std::lock_guard<std::mutex> g(m->mu);
    m->last_stop.reset();
  }
  return BRBD_OK;
}

int brbd_set_timing_period(brbd_model* m, int period) {
  if (!m || period < 1) return fail(BRBD_ERR_INVALID_ARGUMENT, "model handle is NULL or period < 1");
  std::lock_guard<std::mutex> g(m->mu);
  m->timing_period = period;
  for (auto& c : m->timing_count) c = 0;
  m->last_stop.reset();
  return BRBD_OK;
}

int brbd_kernel_time(brbd_model* m, int algo, int32_t* launches, double* total_ms) {
  if (!m || algo < 0 || algo > 6) return fail(BRBD_ERR_INVALID_ARGUMENT, "bad model handle or algo");
  DevGuard dg(m->device);
  std::vector<TimedLaunch> v;
  {
    std::lock_guard<std::mutex> g(m->mu);
    v.swap(m->timed[algo]);
  }
  double sum = 0.0;
  for (auto& t : v) {
    HIP_TRY(hipEventSynchronize(t.stop->e));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, t.start->e, t.stop->e));
    sum += ms;
  }
  if (launches) *launches = (int32_t)v.size();
  if (total_ms) *total_ms = sum;
  return BRBD_OK;
}

}  // extern "C"

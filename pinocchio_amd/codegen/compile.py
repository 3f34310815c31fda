"""Driver of the model compiler: model -> kernel bodies (.inc), packed constant table, signature.

``build_kernel`` runs an algorithm's generator twice (dry pass to learn the lifetimes of the
parked values, storage plan, final pass) and returns the emitted body with the metadata the host
needs: the gather list of the packed constant table and the sizes of the LDS / global park areas.

``cpu_library`` wraps the same bodies in a scalar C function and compiles it with gcc: that is
how tests/test_codegen_cpu.py checks the generator against the oracle without a GPU.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import re
import subprocess
import tempfile
from typing import Dict, Optional

import numpy as np

from .. import model as M
from . import aba, aba_local, deriv, fkjac, jjac, joints, minv, prod, rnea
from .emit import AHEAD, Emitter, eval_const_expr, plan_storage

ALGOS = {"rnea": rnea.generate, "aba": aba.generate, "aba_local": aba_local.generate, "fkjac": fkjac.generate, "deriv": deriv.generate,
         "minv": minv.generate, "prod": prod.generate, "jjac": jjac.generate}

# storage budgets (doubles per lane) of the park planner, per algorithm:
# (registers for short-lived parks, LDS, registers for long-lived parks)
# LDS: 160 KiB per CU, 4 resident waves of 64 lanes at one wave per SIMD -> 80 doubles per lane
BUDGETS = {"rnea": (160, 80, 0), "aba": (48, 80, 100, 64), "aba_local": (48, 80, 100, 64), "fkjac": (36, 80, 0), "deriv": (64, 80, 32, 64),
           "minv": (64, 80, 64, 64), "prod": (0, 0, 0), "jjac": (0, 0, 0)}  # (deriv: first rung of LADDER)
for _a in list(BUDGETS):  # experiment hook: BRBD_CG_ABA="64,80"
    _v = os.environ.get("BRBD_CG_" + _a.upper())
    if _v:
        BUDGETS[_a] = tuple(int(x) for x in _v.split(","))


# prefetch distances that differ from emit.AHEAD (the derivative kernel's pair visits consume a
# parked value every other statement: a long distance keeps hundreds of them in flight)
AHEAD_ALGO = {"deriv": {"glb": 120, "in": 200}, "minv": {"glb": 120, "in": 200}, "prod": {"in": 40}}
for _a in list(ALGOS):  # experiment hook: BRBD_CG_AHEAD_DERIV="glb=200,in=100"
    for _kv in os.environ.get("BRBD_CG_AHEAD_" + _a.upper(), "").split(","):
        if "=" in _kv:
            AHEAD_ALGO.setdefault(_a, {})[_kv.split("=")[0]] = int(_kv.split("=")[1])


# algorithms whose INSTANCE_MAJOR kernels fetch their inputs a cache line per lane at a time (Emitter.group_inputs)
# (experiment, off by default: measured slower than staging the tile, see codegen/build.py: STAGE_MODE)
GROUPED = set(x for x in os.environ.get("BRBD_CG_GROUPED", "").split(",") if x)

# algorithms whose early loads are placed at statement granularity (emit.Emitter.text, ``fine``)
FINE = set(os.environ.get("BRBD_CG_FINE", "deriv,minv,prod").split(","))


# Budget ladder: the budgets above suit the models they were tuned on; a larger model has enough
# parks to fill every class to its limit at once.  ``build_kernel`` walks down the ladder until
# the body's register-pressure estimate (doubles live in program order) fits: hipcc spills to
# scratch beyond ~200 doubles (512 registers per lane at one wave per SIMD, minus addressing).
LADDER = {"aba": [(48, 80, 100, 64), (32, 80, 100, 64), (16, 80, 100, 64), (16, 80, 80, 64)],
          "aba_local": [(48, 80, 100, 64), (32, 80, 100, 64), (16, 80, 100, 64), (16, 80, 80, 64)],
          # rnea: the walk itself keeps about 90 doubles live, so its path state (the humanoid's: 76 doubles)
          # is parked in REGISTERS, not in LDS -- the humanoid's kernel then touches no LDS at all (150
          # doubles live, 282 registers) and runs 16.2 -> 14.5 us (INSTANCE_MAJOR 19.4 -> 17.2; MI355X,
          # same box, alternating; the (40, 80, 0) of rounds 1-2 dated from a walk with twice the live
          # state).  Larger trees overflow into LDS, then get more registers rather than the wave's
          # global area (first rung without global parks).
          "rnea": [(160, 80, 0), (200, 80, 0), (240, 80, 0)],
          "deriv": [(64, 80, 32, 64), (40, 80, 32, 64), (24, 80, 16, 64), (8, 80, 0, 64), (0, 80, 0, 64),
                    (0, 80, 0, 48), (0, 80, 0, 24)],
          "minv": [(64, 80, 64, 64), (32, 80, 32, 64), (16, 80, 16, 64), (0, 80, 0, 64)]}
LIVE_MAX = {"aba": 230, "aba_local": int(os.environ.get("BRBD_CG_LIVEMAX", "240")), "deriv": 182, "minv": 200}

_IDENT = re.compile(r"\b([a-z][a-z]?_?[0-9]+(?:_[0-9]+)?|[sc]_t[0-9]+)\b")


def liveness_peak(text: str) -> int:
    """Most doubles live after any statement of a generated body, in program order (constants of
    the scalar table excluded).  tools/cg_liveness.py prints the same per visit."""
    # (chunked staging: a fetched group is 8 doubles from its GRP_* to the last element read from it)
    text = re.sub(r"GRP_([QVX])\((\d+), (\d+)\);",
                  lambda m: "double " + ", ".join("g%s%d_%d" % (m.group(1).lower(), int(m.group(2)) * 100 + int(m.group(3)), j)
                                                  for j in range(8)) + ";", text)
    text = re.sub(r"IN_([QVX])2\(\d+, (\d+), (\d+), (\d+)\)",
                  lambda m: "g%s%d_%d" % (m.group(1).lower(), int(m.group(2)) * 100 + int(m.group(4)), int(m.group(3))), text)
    stm, uses = [], {}
    for ln, line in enumerate(text.splitlines()):
        if line.startswith("/*") or line.startswith("VISIT_END"):
            stm.append(((), ()))
            continue
        ids = _IDENT.findall(line)
        m = re.match(r"(?:const )?double ([^=;]+?)(?: =|;)", line)
        d = [x.strip() for x in m.group(1).split(",")] if m else []
        u = [x for x in ids if x not in d]
        stm.append((d, u))
        for x in u:
            uses[x] = ln
    live, peak = set(), 0
    for ln, (d, u) in enumerate(stm):
        for x in d:
            if x in uses and not x.startswith("k"):
                live.add(x)
        for x in u:
            if uses.get(x) == ln:
                live.discard(x)
        peak = max(peak, len(live))
    return peak


class Kernel:
    def __init__(self, algo, text, const_list, lds, glb, stats, nvisits):
        self.algo, self.text, self.const_list = algo, text, const_list
        self.lds, self.glb, self.stats, self.nvisits = lds, glb, stats, nvisits


# ABA exists in both conventions of the reference (aba.hxx:102-294 WORLD, :378-542 LOCAL; same ddq,
# unittest/aba.cpp:157-159): the compiler emits both and keeps the shorter program.
ABA_CONVENTION = os.environ.get("BRBD_CG_ABA_CONV", "auto")  # auto | world | local


def kernel_cost(k: "Kernel") -> int:
    """Issue slots of a body, roughly: one per fp64 statement, ~35 per sincos / ~6 per rcp, rsqrt."""
    return k.stats["fp"] + 30 * k.stats["trans"]


# Chunked staging of the INSTANCE_MAJOR kernels (csrc/brbd_spec.hpp: SpecIn<2, ., ., 3>): the inputs
# arrive 8 dofs x 64 instances at a time through an LDS window of 8 rows of 68 doubles (4 352 bytes
# per wave), issued a prefetch distance ahead of the visit that consumes them.  The window comes out
# of the 40 KiB of LDS a wave owns at four waves per CU, so these bodies park at most 70 doubles
# per lane in LDS instead of 80.
CHUNK_WIDTH = 8
CHUNK_WINDOW_PAIRS = 5   # double2 rows of 64 lanes: 5 * 1024 bytes >= 8 * 68 * 8
CHUNK_LDS = 70
CHUNK_LIVE_MARGIN = int(os.environ.get("BRBD_CG_CHUNK_MARGIN", "0"))
CHUNK_GAP = int(os.environ.get("BRBD_CG_CHUNK_GAP", "600"))   # statements after which a group is fetched again


def build_kernel(mv: joints.ModelView, algo: str, reg_budget: Optional[int] = None,
                 lds_budget: Optional[int] = None, chunked: bool = False, like: Optional[Kernel] = None) -> Kernel:
    """``chunked``: the body of the layout-2 kernels with chunked staging; ``like``: the kernel whose
    constant table it must share (same convention and tail, or None is returned)."""
    if algo == "aba":
        conv = {"world": ["aba"], "local": ["aba_local"]}.get(ABA_CONVENTION, ["aba", "aba_local"])
        if like is not None:
            conv = ["aba_local" if like.convention == "local" else "aba"]
        best = min((_build_one(mv, a, reg_budget, lds_budget, chunked, getattr(like, "tail", None)) for a in conv),
                   key=kernel_cost)
        best.convention = "local" if best.algo == "aba_local" else "world"
        best.algo = "aba"
    else:
        best = _build_one(mv, algo, reg_budget, lds_budget, chunked)
    if like is not None and list(best.const_list) != list(like.const_list):
        return None
    return best


# aba_local: (tail length, budgets) in order of preference.  Anything parked in the wave's global
# area costs more than a longer tail (MI355X, humanoid: 40 doubles per lane in global parks 29.3 us,
# none 25.1 us; tails of 3 instead of 2 joints 26.2 us), so the first rung that keeps every park on
# chip within the register estimate wins.
LADDER_LOCAL = [(t, b) for t in (0, 1, 2, 3, 4, 5) for b in ((64, 80, 140, 64), (48, 80, 140, 64), (48, 80, 120, 64), (48, 80, 100, 64))]


def _build_one(mv: joints.ModelView, algo: str, reg_budget: Optional[int] = None,
               lds_budget: Optional[int] = None, chunked: bool = False, only_tail=None) -> Kernel:
    gen = ALGOS[algo]
    ahead = dict(AHEAD, **AHEAD_ALGO.get(algo, {}))
    ladder = [(None, BUDGETS[algo])]
    if reg_budget is None and lds_budget is None and ("BRBD_CG_" + algo.upper()) not in os.environ:
        if algo == "aba_local" and "BRBD_CG_LTAIL" not in os.environ:
            ladder = LADDER_LOCAL
        else:
            ladder = [(None, b) for b in LADDER.get(algo, [BUDGETS[algo]])]
    if only_tail is not None:
        ladder = [(t, b) for t, b in ladder if t == only_tail]
    live_max = LIVE_MAX.get(algo, 0)
    if chunked:
        # the window takes LDS from the parks: first the same rungs with the smaller LDS, then (what no
        # longer fits goes to registers, not to global memory) with the register budgets raised by as
        # much (the groups in flight count in the liveness estimate)
        def capped(b, add):
            lost = max(0, b[1] - CHUNK_LDS)
            return (b[0] + (lost if add else 0), b[1] - lost, b[2] + (lost if add and b[2] else 0)) + tuple(b[3:])
        tails = []
        for t, _ in ladder:
            if t not in tails:
                tails.append(t)
        ladder = [(t, capped(b, add)) for t in tails for add in (False, True) for tt, b in ladder if tt == t]
        live_max -= CHUNK_LIVE_MARGIN
    dry, best = {}, None
    for tail, budget in ladder:
        kw = {} if tail is None else {"tail": tail}
        if tail not in dry:
            dry[tail] = gen(mv, Emitter(), **kw)
        rb, lb, rl = budget[:3]
        rt = budget[3] if len(budget) > 3 else 64
        rb = rb if reg_budget is None else reg_budget
        lb = lb if lds_budget is None else lds_budget
        plan, lds, glb = plan_storage(dry[tail].parks, rb, lb, rl, rt, strict=(algo == "minv"))
        e = gen(mv, Emitter(plan), **kw)
        if chunked:
            e.group_inputs(gap=CHUNK_GAP, width=CHUNK_WIDTH, commit=True)
        elif algo in GROUPED:
            e.group_inputs()
        text = e.text(ahead, algo in FINE)
        k = Kernel(algo, text, list(e.const_list), lds, glb, dict(e.stats), len(e.visits))
        k.tail = tail
        if len(ladder) == 1:
            return k
        if algo == "rnea":
            if glb == 0:
                return k
            best = k if best is None or glb < best.glb else best
            continue
        fits = liveness_peak(text) <= live_max
        if algo != "aba_local":
            best = k
            if fits:
                break
            continue
        if fits and glb == 0:
            return k
        if fits and (best is None or glb < best.glb):
            best = k
    return best if best is not None else k


def pack_constants(mv: joints.ModelView, const_list) -> np.ndarray:
    """The packed constant table: model constants (joint, field, index) and the expressions of
    earlier entries the emitter hoisted out of the body (emit.eval_const_expr)."""
    vals = []
    for ident in const_list:
        vals.append(eval_const_expr(ident, vals) if ident[0] == "expr" else
                    (ident[1] if ident[0] == "lit" else mv.value(ident)))
    return np.array(vals + [0.0], dtype=np.float64)


_CPU_SHELL = r"""
#define _GNU_SOURCE
#include <math.h>
#define CT(k) ct[k]
#define IN_Q(k) q[k]
#define IN_V(k) v[k]
#define IN_X(k) x[k]
#define IN_F(k) fx[k]
#define GRP_Q(g, n) do { } while (0)
#define GRP_V(g, n) do { } while (0)
#define GRP_X(g, n) do { } while (0)
#define GRPC_Q(g, n) do { } while (0)
#define GRPC_V(g, n) do { } while (0)
#define GRPC_X(g, n) do { } while (0)
#define IN_Q2(k, g, j, n) q[k]
#define IN_V2(k, g, j, n) v[k]
#define IN_X2(k, g, j, n) x[k]
#ifndef WANT
#define WANT 0
#endif
#define OUT(k, val) out[k] = (val)
#define OUT2(k, val) out2[k] = (val)
#define OUT3(k, val) out3[k] = (val)
#define OUT4(k, val) out4[k] = (val)
#define OUTSYM(k, val) out[k] = (WANT & 2) ? (val) : 0.0
#define LD1(k) out[k]
#define LD2(k) out2[k]
#define LD3(k) out3[k]
#ifdef SKIP_ZEROS
#define OUTZ(k) do { } while (0)
#define OUT2Z(k) do { } while (0)
#define OUT3Z(k) do { } while (0)
#else
#define OUTZ(k) out[k] = 0.0
#define OUT2Z(k) out2[k] = 0.0
#define OUT3Z(k) out3[k] = 0.0
#endif
#define SINCOS(a, s, c) sincos(a, &s, &c)
#define RSQRT(a) (1.0 / sqrt(a))
#define RCP(a) (1.0 / (a))
#define FMA(a, b, c) fma(a, b, c)
#define PK_LDS_ST2(p, a, b) do { lds[2 * (p)] = (a); lds[2 * (p) + 1] = (b); } while (0)
#define PK_LDS_LD2(p, a, b) do { a = lds[2 * (p)]; b = lds[2 * (p) + 1]; } while (0)
#define PK_GLB_ST2(p, a, b) do { glb[2 * (p)] = (a); glb[2 * (p) + 1] = (b); } while (0)
#define PK_GLB_LD2(p, a, b) do { a = glb[2 * (p)]; b = glb[2 * (p) + 1]; } while (0)
#define VISIT_END() do { } while (0)
#define OPAQUE(x) do { } while (0)
unsigned long long spec_chain = 0;  /* support chain of the joint (codegen/jjac.py), set by the test */
#define ON_CHAIN(i) ((spec_chain >> (i)) & 1ull)
static void one(const double* ct, const double* q, const double* v, const double* x, const double* fx,
                double* out, double* out2, double* out3, double* out4, double* lds, double* glb) {
  (void)ct; (void)q; (void)v; (void)x; (void)fx; (void)out; (void)out2; (void)out3; (void)out4; (void)lds; (void)glb;
#include "BODY"
}
void spec_batch(const double* ct, long B, const double* q, long ldq, const double* v, long ldv,
                const double* x, long ldx, double* out, long ldo, double* out2, long ldo2, double* out3,
                long ldo3, double* out4, long ldo4, const double* fx, long ldf) {
  double lds[LDS_N + 2], glb[GLB_N + 2];
  for (long b = 0; b < B; ++b)
    one(ct, q + b * ldq, v ? v + b * ldv : 0, x ? x + b * ldx : 0, fx ? fx + b * ldf : 0, out + b * ldo, out2 ? out2 + b * ldo2 : 0,
        out3 ? out3 + b * ldo3 : 0, out4 ? out4 + b * ldo4 : 0, lds, glb);
}
"""


class CpuKernel:
    """The generated body compiled for the host (checker of the generator, never a product path)."""

    def __init__(self, mv: joints.ModelView, algo: str, skip_zeros: bool = False, want: int = 0, **kw):
        self.mv, self.kernel = mv, build_kernel(mv, algo, **kw)
        self.ct = pack_constants(mv, self.kernel.const_list)
        self.dir = tempfile.mkdtemp(prefix="brbd_cg_")
        body = os.path.join(self.dir, "body.inc")
        with open(body, "w") as f:
            f.write(self.kernel.text)
        src = os.path.join(self.dir, "shell.c")
        with open(src, "w") as f:
            f.write(_CPU_SHELL.replace("BODY", body).replace("LDS_N", str(self.kernel.lds)).replace(
                "GLB_N", str(self.kernel.glb)))
        so = os.path.join(self.dir, "spec.so")
        # -ffp-contract=off: only the FMAs the generator asked for
        subprocess.check_call(["gcc", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src, "-lm"]
                              + (["-DSKIP_ZEROS"] if skip_zeros else []) + ["-DWANT=%d" % want])
        self.lib = ctypes.CDLL(so)
        dp, l = ctypes.POINTER(ctypes.c_double), ctypes.c_long
        self.lib.spec_batch.argtypes = [dp, l, dp, l, dp, l, dp, l, dp, l, dp, l, dp, l, dp, l, dp, l]

    def run(self, q, v=None, x=None, out_rows=(0, 0, 0), fext=None, init_outs=None):
        out_rows = tuple(out_rows) + (0,) * (4 - len(out_rows))
        def P(a):
            return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))

        q = np.asfortranarray(q, dtype=np.float64)
        B = q.shape[1]
        v = None if v is None else np.asfortranarray(v, dtype=np.float64)
        x = None if x is None else np.asfortranarray(x, dtype=np.float64)
        fext = None if fext is None else np.asfortranarray(fext, dtype=np.float64)
        outs = [np.zeros((r, B), order="F") if r else None for r in out_rows]
        for o, src in zip(outs, init_outs or ()):  # (kernels that update their outputs in place)
            if o is not None and src is not None:
                o[...] = src
        self.lib.spec_batch(P(self.ct), B, P(q), q.shape[0], P(v), 0 if v is None else v.shape[0], P(x),
                            0 if x is None else x.shape[0], P(outs[0]), out_rows[0], P(outs[1]), out_rows[1],
                            P(outs[2]), out_rows[2], P(outs[3]), out_rows[3], P(fext),
                            0 if fext is None else fext.shape[0])
        return outs


def signature_hash(sig: str) -> str:
    return hashlib.sha1(sig.encode()).hexdigest()[:16]

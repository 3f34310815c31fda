"""Generate the model-compiled kernels of the registered models.

    python -m pinocchio_amd.codegen.build [--out pinocchio_amd/csrc/gen] [--models a,b,...]

Variants of a registered model: ``p`` (pattern: exact zeros / +-1 of the model's constants
folded; serves models with the same tree, joint kinds and zero pattern), ``a`` (the same with the
armature of every 1-dof joint a run-time constant: robots with rotor inertias; shares the kernels
the armature does not enter with ``p``) and, on request, ``d`` (dense: every geometric / inertial
constant generic; serves any model with the same tree and joint kinds).  Gravity is a run-time
constant in all of them.  Output per variant: ``spec_<model>_<v>.hip`` (kernel shells + registry entry)
and the bodies ``<model>_<v>_<algo>.inc``; plus ``registry.inc`` listing all entries.
pinocchio_amd/csrc/Makefile compiles each ``.hip`` as its own translation unit.
"""
from __future__ import annotations

import argparse
import os
import sys

from .. import sample_models as S
from . import compile as C
from . import joints

# Models that get compiled kernels by default.  Selection rule (tools/spec_usage.py after a build):
# no scratch traffic and at most a few dozen spilled SGPRs in every kernel.  baxter_simple (rpy
# placements: ~3x the constants per visit) and the dense variants of the larger trees exceed the
# SGPR budget of hipcc 7.2's allocator by hundreds of spills and stay on the table-driven kernels
# until the generator feeds their constants through another path.
REGISTERED = ["simple_humanoid", "talos_like", "quadruped", "arm7", "manipulator"]
FIELD = {"R": 0, "t": 1, "m": 2, "c": 3, "I": 4, "arm": 5, "axis": 6, "g": 7}
ALGO_ORDER = ["rnea", "aba", "fkjac", "deriv", "minv", "prod", "jjac"]


# kernel variants (WANT masks) per algorithm; fkjac: bit 0 = J, bit 1 = oMi; deriv: all four outputs
# compiled in (an absent one has a zero-length buffer range, so its stores are dropped by the
# hardware), bit 4 = also write the structural zeros; rnea / aba: bit 5 = external forces
# minv: bit 4 = write the structural zeros (always), bit 1 = symmetric fill (computeABADerivatives)
# jjac (getJointJacobian): bit 0 = write J, bit 1 = LOCAL, bit 2 = LOCAL_WORLD_ALIGNED (else WORLD): the
# variant index is pinocchio::ReferenceFrame (0 WORLD, 1 LOCAL, 2 LOCAL_WORLD_ALIGNED)
WANTS = {"rnea": [1, 33], "aba": [1, 33], "fkjac": [1, 2, 3], "deriv": [31, 15], "minv": [17, 19], "prod": [3],
         "jjac": [1, 3, 5]}
# Kernel layouts per algorithm: 0 INSTANCE_MAJOR per lane, 1 BATCH_MAJOR, 2 INSTANCE_MAJOR staged through
# LDS (csrc/brbd_spec.hpp: SpecIn / SpecOut; inputs q, v, a | tau and the one output are small
# enough to transpose in one go).  Staging mode of the inputs: 0 = registers, 1 = the wave's global
# staging rows.  Round 2's world-frame ABA had no registers to spare and used mode 1 (106 rows out to
# global memory and back: 2.6x the algorithmic traffic); the local-frame ABA stages into registers like
# RNEA: the staged values die as the walk consumes them while the hand-off records are born, and the
# 20 doubles per lane hipcc cannot place go to scratch (humanoid, MI355X: 2.05e9 -> 2.49e9 evals/s in
# the reference's layout; lower park budgets instead of the spills: 2.39e9).
# Mode 2 (experiment: BRBD_CG_STAGE_MODE="rnea=2,aba=2" with BRBD_CG_GROUPED=rnea,aba,aba_local): no
# staging, every lane fetches its own column a 128-byte line at a time when the walk first needs it
# (csrc/brbd_spec.hpp: SpecIn<2, ., ., 2>).  Measured slower than register staging (humanoid, 65 536:
# RNEA 22.9 against 20.1 us, ABA 34.6 against 28.7 us): 64 lanes x 64 different lines per load
# instruction is texture-addresser time that the coalesced tile loads do not pay.
# (minv: BATCH_MAJOR only -- INSTANCE_MAJOR calls of the nv x nv outputs run on batch-major copies anyway)
# fkjac: layout 2 for models whose outputs are emitted in increasing element order (fkjac.monotone):
# both outputs are streamed out in chunks of 64 rows
LAYOUTS = {"rnea": (0, 1, 2), "aba": (0, 1, 2), "fkjac": (0, 1, 2), "deriv": (0, 1), "minv": (1,), "prod": (1,),
           "jjac": (0, 1, 2)}
# Mode 3 (experiment: BRBD_CG_STAGE_MODE="rnea=3,aba=3"): chunked staging -- the inputs arrive 8 dofs x 64
# instances at a time through a small LDS window, a prefetch distance ahead of the visit that consumes
# them, so that the loads overlap the walk as they do in BATCH_MAJOR (csrc/brbd_spec.hpp:
# SpecIn<2, ., ., 3>; these kernels run a body of their own, compile.build_kernel(chunked=True), with
# the LDS parks capped at 70 doubles per lane, that shares the constant table of the other layouts').
# Measured much slower than staging the tile up front (humanoid, 65 536, MI355X: 1.92e9 against
# 2.96e9 evals/s for the RNEA + ABA step): every group fetch touches a 64-byte piece of EVERY cache
# line of the tile, 4.5 times over the kernel with other traffic in between, so the lines are
# fetched again and again, while the up-front staging reads each line exactly once.
# Mode 4 = mode 0 with 16-byte loads and LDS writes (half the instructions of the prologue); measured per
# kernel on the humanoid at 65 536: RNEA 17.3 -> 16.4 us, ABA 25.1 -> 26.3 us (its register file is full,
# the two padding registers per matrix and the different schedule cost more than the prologue saves).
STAGE_MODE = {"rnea": 4, "aba": 0, "fkjac": 0, "deriv": 0, "minv": 0, "prod": 0, "jjac": 0}
STAGED_WANTS = [1]  # (the external-force variants read 6 * njoints more rows per lane anyway: not staged)


def _layouts(algo: str, model) -> tuple:
    if algo in ("fkjac", "jjac"):
        from . import fkjac
        if not fkjac.monotone(joints.ModelView(model)):
            return (0, 1)
    return LAYOUTS[algo]


def _stage_mode(algo: str, model) -> int:
    """Register staging of the layout-2 ABA kernel only while the staged tile (nq + 2 nv doubles per
    lane) fits next to the kernel's register parks: simple_humanoid (106) spills 20 doubles per lane
    and no SGPR; talos_like (115) spills 57 AND 12 SGPRs, the combination that is never shipped
    (codegen/usage.py) -- such models keep the global staging rows."""
    if algo == "aba" and STAGE_MODE[algo] == 0 and model.nq + 2 * model.nv > 108:
        return 1
    return STAGE_MODE[algo]


def _fallback_mode(algo: str, model) -> int:
    """staging mode of a model whose chunked body cannot share the constant table"""
    return 1 if (algo == "aba" and model.nq + 2 * model.nv > 108) else 0


def _staged_wants(algo: str):
    return WANTS[algo] if algo in ("fkjac", "jjac") else STAGED_WANTS
for _kv in os.environ.get("BRBD_CG_STAGE_MODE", "").split(","):  # experiment hook: "aba=0"
    if "=" in _kv:
        STAGE_MODE[_kv.split("=")[0]] = int(_kv.split("=")[1])


# models handed to the generator as objects (codegen/plugin.py: a URDF, a user's tables) next to
# the built-in ones that are looked up by name
EXTRA_MODELS = {}


def _model(name: str):
    return EXTRA_MODELS[name] if name in EXTRA_MODELS else S.build(name)


def _write(path: str, text: str):
    """Write only when the content changes, so that make recompiles only what really changed."""
    try:
        with open(path) as f:
            if f.read() == text:
                return
    except OSError:
        pass
    with open(path, "w") as f:
        f.write(text)


def _emit_algo(args):
    """One algorithm of one model variant: the body (.inc), the kernel template around it
    (.kern.inc) and one translation unit per boundary layout and output mask that instantiates it."""
    out_dir, name, variant, algo = args
    model = _model(name)
    mv = joints.ModelView(model, dense=(variant == "d"), arm_generic=(variant == "a"))
    ident = "%s_%s" % (name, variant)
    k = C.build_kernel(mv, algo)
    nq, nv = model.nq, model.nv
    layouts = _layouts(algo, model)
    staged = 2 in layouts
    # rows of the staged matrices: q, v, third input, first output (fkjac: q in, the Jacobian out)
    rows = (nq, 0, 0, 6 * nv) if algo in ("fkjac", "jjac") else (nq, nv, nv, nv)
    rows2 = 12 * model.njoints if algo == "fkjac" else 0          # second output, when it is streamed too
    stream = 1 if algo in ("fkjac", "jjac") else 0                # outputs leave in chunks of 64 rows
    stage_pairs = (max(rows[:3] + ((64,) if stream else rows[3:])) + 1) // 2
    stage_mode = _stage_mode(algo, model)
    k2 = None
    if staged and stage_mode == 3:
        k2 = C.build_kernel(mv, algo, chunked=True, like=k)
        if k2 is None:
            stage_mode = _fallback_mode(algo, model)
    stage_rows = (nq + 2 * nv) if (staged and stage_mode == 1) else 0
    body = "%s_%s.inc" % (ident, algo)
    _write(os.path.join(out_dir, body), k.text)
    body2 = "%s_%s.chunk.inc" % (ident, algo)
    if k2 is not None:
        _write(os.path.join(out_dir, body2), k2.text)
    elif os.path.exists(os.path.join(out_dir, body2)):
        os.remove(os.path.join(out_dir, body2))
    glb = max(k.glb, k2.glb if k2 is not None else 0)
    out_pairs = (max(((64,) if stream else rows[3:])) + 1) // 2   # the output leaves through LDS at the end
    lds2_pairs = max(k2.lds // 2 + C.CHUNK_WINDOW_PAIRS, out_pairs) if k2 is not None else max(k.lds // 2, stage_pairs)
    kern = "%s_%s.kern.inc" % (ident, algo)
    kern_text = ("\n".join([
            "// GENERATED by pinocchio_amd/codegen/build.py -- do not edit.",
            "// %s: %d fp64 ops, %d transcendental, %d loads, %d stores, %d constants; park: %d LDS + %d global doubles/lane"
            % (algo, k.stats["fp"], k.stats["trans"], k.stats["ld"], k.stats["st"], len(k.const_list), k.lds, k.glb),
            "template <int LAYOUT, int WANT>",
            "__global__ __launch_bounds__(WAVE) void k_%s(const KArgs A) {" % algo,
            "  constexpr int SPEC_IN_AUX = %d, SPEC_OUT_AUX = 2;  // cache policy, see brbd_spec.hpp" % (
                2 if algo in ("rnea", "aba") else 0),
            "  // rows of q, v, the third input and the first output; staging of the layout-2 kernels",
            "  constexpr int SPEC_NQ = %d, SPEC_NV = %d, SPEC_NX = %d, SPEC_NO = %d, SPEC_STAGE_MODE = %d, SPEC_GLB = %d;" % (
                rows + (stage_mode, glb)),
            "  constexpr int SPEC_LDS_PARK = LAYOUT == 2 ? %d : %d;  // doubles per lane of the park area (the staging window follows it)"
            % (k2.lds if k2 is not None else k.lds, k.lds),
            ] + (["  constexpr int SPEC_NO2 = %d, SPEC_OUT_STREAM = %d;  // (shadow the defaults of brbd_spec.hpp)" % (rows2, stream)]
                 if stream else []) + [
            "  BRBD_SPEC_PROLOGUE(LAYOUT == 2 ? %d : %d)" % (lds2_pairs, k.lds // 2),
            "  {  // one tile of 64 instances per workgroup: no loop, nothing for LICM to hoist",
            "    const int64_t tile = A.tile0 + blockIdx.x;",
            "    BRBD_SPEC_TILE(LAYOUT)",
            ] + (['#include "%s"' % body] if k2 is None else [
            "    if constexpr (LAYOUT == 2) {  // chunked staging: a body of its own, same constant table",
            '#include "%s"' % body2,
            "    } else {",
            '#include "%s"' % body,
            "    }"]) + [
            "    BRBD_SPEC_EPILOGUE();",
            "  }",
            "}",
            ""]))
    _write(os.path.join(out_dir, kern), kern_text)
    import hashlib
    digest = hashlib.sha1((k.text + (k2.text if k2 is not None else "") + kern_text).encode()).hexdigest()[:16]
    for layout in layouts:
        for w in (WANTS[algo] if layout != 2 else _staged_wants(algo)):
            _write(os.path.join(out_dir, "spec_%s_%s_%d_%d.hip" % (ident, algo, layout, w)), "\n".join([
                "// GENERATED by pinocchio_amd/codegen/build.py -- do not edit.",
                "// body + template digest %s (this file changes, and is recompiled, exactly when they do)" % digest,
                '#include "../brbd_spec.hpp"',
                "namespace brbd {",
                "namespace spec_%s {" % ident,
                '#include "%s"' % kern,
                "template __global__ void k_%s<%d, %d>(const KArgs A);" % (algo, layout, w),
                "}  // namespace spec_%s" % ident,
                "}  // namespace brbd",
                ""]))
    from .emit import encode_const
    gather = ", ".join("%d,%d,%d,%d" % encode_const(ident, FIELD) for ident in k.const_list) or "0"
    return algo, gather, len(k.const_list), k.lds, glb, stage_rows, layouts


# the armature enters these algorithms only; the "a" variant (armature of 1-dof joints generic) shares
# the kernels of the others with the "p" variant of the same model
ARMATURE_ALGOS = ("rnea", "aba", "deriv", "minv")


def emit_variant(out_dir: str, name: str, variant: str, pool=None) -> str:
    """The registry translation unit of one model variant (kernel declarations, gather lists of the
    packed constant tables, SpecEntry); the kernels themselves are compiled in TUs of their own."""
    mv = joints.ModelView(_model(name), dense=(variant == "d"), arm_generic=(variant == "a"))
    ident = "%s_%s" % (name, variant)

    def owner(algo):  # variant whose kernels serve this algorithm
        return "p" if (variant == "a" and algo not in ARMATURE_ALGOS) else variant

    jobs = [(out_dir, name, owner(algo), algo) for algo in ALGO_ORDER]
    results = pool.map(_emit_algo, jobs) if pool is not None else [_emit_algo(j) for j in jobs]
    lines = [
        "// GENERATED by pinocchio_amd/codegen/build.py -- do not edit.",
        '#include "../brbd_spec.hpp"',
        "namespace brbd {",
    ]
    for v in sorted({owner(a) for a in ALGO_ORDER}):
        lines.append("namespace spec_%s_%s {" % (name, v))
        for algo in ALGO_ORDER:
            if owner(algo) == v:
                lines.append("template <int LAYOUT, int WANT> __global__ void k_%s(const KArgs A);  // spec_%s_%s_%s_<layout>_<want>.hip"
                             % (algo, name, v, algo))
        lines.append("}  // namespace spec_%s_%s" % (name, v))
    lines.append("namespace spec_%s {" % ident)
    entries = []
    for algo, gather, nconst, lds, glb, stage_rows, layouts in results:
        ns = "spec_%s_%s::" % (name, owner(algo))
        lines += [
            "static const int32_t gather_%s[] = {%s};" % (algo, gather),
        ]
        rows = []
        for layout in (0, 1, 2):
            if layout not in layouts:
                fn = ["nullptr"] * 4
            elif algo == "fkjac":   # variant index = WANT mask
                fn = ["nullptr"] + ["%sk_fkjac<%d, %d>" % (ns, layout, w) for w in (1, 2, 3)]
            else:                 # variant index = position in WANTS
                fn = [("%sk_%s<%d, %d>" % (ns, algo, layout, w)) if (layout != 2 or w in _staged_wants(algo)) else "nullptr"
                      for w in WANTS[algo]]
                fn += ["nullptr"] * (4 - len(fn))
            rows.append("{" + ", ".join(fn) + "}")
        entries.append("{{%s}, spec_%s::gather_%s, %d, %d, %d, %d}" % (", ".join(rows), ident, algo, nconst, lds, glb, stage_rows))
    lines += [
        "}  // namespace spec_%s" % ident,
        "// host pass only: the table holds the kernels' host-side handles; their device code lives in",
        "// the translation units that instantiate them",
        "#ifndef __HIP_DEVICE_COMPILE__",
        'extern const SpecEntry spec_entry_%s = {"%s", "%s", {%s}};' % (
            ident, ident, mv.signature(), ",\n    ".join(entries)),
        "#endif",
        "}  // namespace brbd",
        "",
    ]
    _write(os.path.join(out_dir, "spec_%s.hip" % ident), "\n".join(lines))
    return ident


def main(argv=None):
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.abspath(__file__))
    ap.add_argument("--out", default=os.path.join(here, "..", "csrc", "gen"))
    ap.add_argument("--models", default=",".join(REGISTERED))
    ap.add_argument("--variants", default=os.environ.get("BRBD_SPEC_VARIANTS", "p,a"))
    a = ap.parse_args(argv)
    os.makedirs(a.out, exist_ok=True)
    idents = []
    import multiprocessing
    with multiprocessing.Pool(min(8, os.cpu_count() or 1)) as pool:
        variants = [v for v in a.variants.split(",") if v]
        if "a" in variants and "p" not in variants:
            variants.insert(0, "p")  # (the "a" variant borrows the armature-free kernels of "p")
        for name in [m for m in a.models.split(",") if m]:
            for v in variants:
                idents.append(emit_variant(a.out, name, v, pool))
                print("generated", idents[-1], file=sys.stderr)
    reg = "// GENERATED by pinocchio_amd/codegen/build.py -- do not edit.\nnamespace brbd {\n"
    for i in idents:
        reg += "extern const SpecEntry spec_entry_%s;\n" % i
    reg += "static const SpecEntry* const spec_entries[] = {%s};\n" % ", ".join("&spec_entry_" + i for i in idents)
    reg += "static const int spec_entry_count = %d;\n}  // namespace brbd\n" % len(idents)
    _write(os.path.join(a.out, "registry.inc"), reg)
    # translation units of kernels that are no longer generated would still be linked: remove them
    import glob
    keep = set()
    for i in idents:
        keep.add("spec_%s.hip" % i)
    for path in glob.glob(os.path.join(a.out, "spec_*_*_*_*.hip")):
        base = os.path.basename(path)
        name = base[len("spec_"):]
        if not any(name.startswith(i + "_") for i in idents):
            continue
        stem = base[:-4]
        parts = stem.split("_")
        algo, layout, w = parts[-3], int(parts[-2]), int(parts[-1])
        ok = algo in LAYOUTS and layout in LAYOUTS[algo] and w in (WANTS[algo] if layout != 2 else _staged_wants(algo))
        if not ok:
            for ext in (".hip", ".o", ".usage"):
                try:
                    os.remove(os.path.join(a.out, stem + ext))
                except OSError:
                    pass


if __name__ == "__main__":
    main()

"""Straight-line code emitter with structural constant folding.

The model compiler (rnea.py / aba.py / fkjac.py) runs a recursion ONCE, symbolically, for one
constant model, on values of class ``V``.  Every arithmetic helper folds exact zeros and +-1
(structural zeros of joint subspaces, the universe's zero velocity, identity placements, zero
centre-of-mass components ...) and emits one C statement per surviving floating-point
instruction.  The result is a basic block per visit of the depth-first program: no interpreter,
no table decode, no selects -- on a SIMD that holds a single wave every instruction, scalar or
vector, costs one 4-cycle issue slot, so the instruction count IS the run time
(DESIGN.md section 4).

The emitted text is plain C that is also valid HIP device code: the same ``.inc`` body is
compiled by gcc for the CPU-side check of the generator (tests/test_codegen_cpu.py) and by hipcc
inside the kernel shells of ``csrc/brbd_spec.hpp``.  It only uses the macros documented in
``MACROS`` below.

Long-lived values are *parked*: the compiler decides nothing about them.  A first, dry pass over
the program records every park with its lifetime; ``plan_storage`` then assigns each one to a
storage class (registers / LDS / wave-private global workspace) under explicit budgets, and the
second pass emits the loads of a parked value one visit ahead of its use.
"""
from __future__ import annotations

import math
import re
import os
from typing import Dict, List, Optional, Sequence, Tuple

MACROS = """
  CT(k)                 k-th double of this kernel's packed constant table (wave-uniform -> s_load)
  IN_Q(k) IN_V(k) IN_X(k)   element k of q / v / third input (a or tau) of this lane's instance
  GRP_Q(g, n) / IN_Q2(k, g, j, n)   (also V, X) fetch n of the 16-element group g of a matrix / its element
                        j = element k of the matrix (Emitter.group_inputs; plain element loads outside layout 2)
  OUT(k, x)  OUT2(k, x)  OUT3(k, x)  store element k of the 1st / 2nd / 3rd output
  LD1(k) LD2(k) LD3(k)  element k of the 1st / 2nd / 3rd output matrix, read (in-place kernels)
  OUTSYM(k, x)          element k of the 1st output, mirror image of a stored one (WANT bit 1: x, else 0)
  SINCOS(x, s, c)       s = sin(x), c = cos(x)
  RSQRT(x)  RCP(x)      1/sqrt(x), 1/x
  FMA(a, b, c)          a*b + c, one rounding
  PK_LDS_ST2(p, a, b) / PK_LDS_LD2(p, a, b)   pair slot p of the LDS park area
  PK_GLB_ST2(p, a, b) / PK_GLB_LD2(p, a, b)   pair slot p of the global park area
  VISIT_END()           boundary between two visits (scheduling barrier on the GPU)
  OPAQUE(x)             make the compiler forget how x was computed (no instruction)
"""


class V:
    """A value: either an exact number known at generation time, or +-(a C variable)."""

    __slots__ = ("name", "neg", "num")

    def __init__(self, name: Optional[str] = None, neg: bool = False, num: float = 0.0):
        self.name, self.neg, self.num = name, neg, num

    @property
    def is_num(self) -> bool:
        return self.name is None

    def is_zero(self) -> bool:
        return self.name is None and self.num == 0.0

    def __neg__(self) -> "V":
        if self.name is None:
            return V(num=-self.num)
        return V(self.name, not self.neg)

    def code(self) -> str:
        if self.name is None:
            return _lit(self.num)
        return ("-" if self.neg else "") + self.name

    def key(self):
        return (self.name, self.neg) if self.name is not None else ("#", self.num)

    def __repr__(self):
        return "V(" + self.code() + ")"


def _lit(x: float) -> str:
    if x == math.floor(x) and abs(x) < 1e15:
        s = "%.1f" % x
    else:
        s = repr(float(x))
    return "(" + s + ")" if x < 0 else s


# prefetch distance in statements per kind of early load (experiment hook: BRBD_CG_AHEAD="in=800")
AHEAD = {"ct": 60, "in": 400, "inf": 100, "lds": 0, "glb": 400}
for _kv in os.environ.get("BRBD_CG_AHEAD", "").split(","):
    if "=" in _kv:
        AHEAD[_kv.split("=")[0]] = int(_kv.split("=")[1])

_NAME = re.compile(r"\b[a-z][a-z]?_?[0-9]+(?:_[0-9]+)?\b")
_DECL = re.compile(r"(?:const )?double ([^=;]+?)(?: =|;)")

ZERO = V(num=0.0)
ONE = V(num=1.0)

# Expressions of model constants alone are evaluated on the host (experiment hook: BRBD_CG_HOIST=0)
HOIST_CONST = os.environ.get("BRBD_CG_HOIST", "1") == "1"
# ... while the visit has fewer than this many table entries: they travel in SGPR pairs, about 30 of
# which are free next to the buffer resources; hipcc spills the rest into VGPR lanes
HOIST_CAP = int(os.environ.get("BRBD_CG_HOISTCAP", "24"))
_KNAME = re.compile(r"-?k[0-9]+$")


def _is_const(x) -> bool:
    return not isinstance(x, str) or _KNAME.match(x) is not None


CONST_OPS = {"mul": 0, "add": 1, "fma": 2, "rcp": 3, "rsqrt": 4}


def eval_const_expr(ident, values) -> float:
    """Value of a hoisted constant expression ("expr", op, (index, negated)...) given the values of
    the table entries before it.  csrc/brbd_capi.hip: pack_constants evaluates the same program."""
    op, a = ident[1], [(-values[i] if neg else values[i]) for i, neg in ident[2:]]
    if op == "mul":
        return a[0] * a[1]
    if op == "add":
        return a[0] + a[1]
    if op == "fma":
        return a[0] * a[1] + a[2]
    if op == "rcp":
        return 1.0 / a[0]
    if op == "rsqrt":
        return 1.0 / math.sqrt(a[0])
    raise KeyError(op)


def encode_const(ident, field_codes) -> Tuple[int, int, int, int]:
    """One entry of a kernel's gather list (csrc/brbd_spec.hpp: SpecKernel::gather): a model constant
    (field, joint, index, 0), a literal (15, low word, high word, 0) or an expression of earlier
    entries (16 + op, operands as 2 * index + negated)."""
    import struct
    if ident[0] == "lit":
        lo, hi = struct.unpack("<ii", struct.pack("<d", ident[1]))
        return (15, lo, hi, 0)
    if ident[0] == "expr":
        r = [2 * i + (1 if neg else 0) for i, neg in ident[2:]] + [0, 0]
        return (16 + CONST_OPS[ident[1]], r[0], r[1], r[2])
    j, f, i = ident
    return (field_codes[f], j, i, 0)


def num(x: float) -> V:
    return V(num=float(x))


class Park:
    __slots__ = ("pid", "vals", "start", "end", "hint", "cls", "slot")

    def __init__(self, pid, vals, start, hint):
        self.pid, self.vals, self.start, self.end, self.hint = pid, vals, start, start, hint
        self.cls, self.slot = "reg", 0


class Visit:
    __slots__ = ("pre", "body", "label")

    def __init__(self, label):
        self.pre: List[str] = []
        self.body: List[str] = []
        self.label = label


class Emitter:
    """Collects the statements of one kernel body.  ``plan`` = None: dry run (records parks)."""

    def __init__(self, plan: Optional[Dict[int, Tuple[str, int]]] = None):
        self.plan = plan
        self.visits: List[Visit] = []
        self.n = 0
        self.cse: Dict[tuple, V] = {}
        self.parks: List[Park] = []
        self.const_list: List[tuple] = []  # packed constant table: (joint, field, index)
        self.const_cache: Dict[tuple, V] = {}
        self.in_cache: Dict[tuple, V] = {}
        self.stats = {"fp": 0, "ld": 0, "st": 0, "ct": 0, "trans": 0}
        self.has_blocks = False
        self.fillers: List[str] = []  # stores that depend on nothing (structural zeros)
        self.lit_index: Dict[float, int] = {}
        self.opaque_regs = False  # experiment: made register pressure worse with hipcc 7.2

    # ---- visits --------------------------------------------------------------------------
    def begin_visit(self, label: str):
        self.visits.append(Visit(label))
        self.cse = {}
        self.const_cache = {}
        self.in_cache = {}

    @property
    def vi(self) -> int:
        return len(self.visits) - 1

    def _tmp(self) -> str:
        self.n += 1
        return "t%d" % self.n

    def stmt(self, line: str):
        self.visits[-1].body.append(line)

    def pre(self, line: str, kind: str = "ct", earliest: int = 0):
        """A load that may be issued ahead of the current visit: ``kind`` selects the prefetch
        distance ("ct" scalar table, "in" boundary matrices, "lds" / "glb" park areas), ``earliest``
        is the first visit it may be issued in (a park's loads come after its stores)."""
        self.visits[-1].pre.append((line, kind, earliest))

    def comment(self, text: str):
        self.stmt("/* " + text + " */")

    # ---- leaves ----------------------------------------------------------------------------
    def const(self, value: float, ident: tuple, dense: bool = False) -> V:
        """A model constant.  Exact 0 / +-1 fold unless ``dense``; the rest is read from the packed
        table (one contiguous run per visit, so the scalar loads merge into s_load_dwordx8/x16)."""
        if not dense:
            if value == 0.0:
                return ZERO
            if value == 1.0:
                return ONE
            if value == -1.0:
                return V(num=-1.0)
        if ident in self.const_cache:
            return self.const_cache[ident]
        k = len(self.const_list)
        self.const_list.append(ident)
        name = "k%d" % k
        self.pre("const double %s = CT(%d);" % (name, k))
        self.stats["ct"] += 1
        v = V(name)
        self.const_cache[ident] = v
        return v

    def inp(self, which: str, k: int) -> V:
        key = (which, k)
        if key in self.in_cache:
            return self.in_cache[key]
        name = "%s%d_%d" % (which.lower(), k, self.vi)
        # (external forces: 6 more loads per joint, only live in the fext kernel variants, which are the
        # ones closest to the register limit: a shorter prefetch distance keeps fewer of them in flight)
        self.pre("const double %s = IN_%s(%d);" % (name, which, k), "inf" if which == "F" else "in")
        self.stats["ld"] += 1
        v = V(name)
        self.in_cache[key] = v
        return v

    def inp_out(self, which: int, k: int) -> V:
        """Element k of output matrix `which` as an INPUT (kernels that update a matrix in place or
        read a third one): a fresh load every time it is asked for."""
        name = "ld%d_%d" % (self._next_io(), which)  # (shape of the names the text passes look for)
        self.pre("const double %s = LD%d(%d);" % (name, which, k), "in")
        self.stats["ld"] += 1
        return V(name)

    def _next_io(self) -> int:
        self.n_io = getattr(self, "n_io", 0) + 1
        return self.n_io

    def out(self, k: int, x: V, which: int = 1):
        mac = "OUT" if which == 1 else "OUT%d" % which
        self.stmt("%s(%d, %s);" % (mac, k, x.code()))
        self.stats["st"] += 1

    def out_sym(self, k: int, x: V):
        """Element k of the first output as the mirror image of an element already written: stored
        by the kernel variants with WANT bit 1 (symmetric fill), zero (or left alone) in the others."""
        self.stmt("OUTSYM(%d, %s);" % (k, x.code()))
        self.stats["st"] += 1

    def cond_sub(self, vals: Sequence[V], thunk, bit: int = 32) -> List[V]:
        """vals - thunk() in the kernel variants whose WANT mask has ``bit`` (external forces),
        vals unchanged in the others: the statements thunk() emits sit in an ``if (WANT & bit)``
        block, a compile-time constant in every instantiation."""
        if os.environ.get("BRBD_CG_NOFEXT") == "1":  # experiment hook: bodies without the blocks
            return list(vals)
        names = []
        for v in vals:
            n = self._tmp()
            self.stmt("double %s = %s;" % (n, v.code()))
            names.append(n)
        saved = dict(self.cse)
        self.stmt("if (WANT & %d) {" % bit)
        for n, d in zip(names, thunk()):
            if not d.is_zero():
                self.stmt("%s -= %s;" % (n, d.code()))
                self.stats["fp"] += 1
        self.stmt("}")
        self.cse = saved  # (what the block computed is out of scope after it)
        self.has_blocks = True
        return [V(n) for n in names]

    def block(self, cond: str):
        """``with e.block("cond"):`` -- the statements emitted inside sit in ``if (cond) { ... }``
        (a wave-uniform or compile-time condition).  Temporaries of the block are out of scope after
        it, so the common-subexpression table is restored on exit; loads and table constants asked for
        inside are issued ahead of the visit as usual, i.e. unconditionally."""
        em = self

        class _Block:
            def __enter__(self_b):
                self_b.saved = dict(em.cse)
                em.stmt("if (%s) {" % cond)
                em.has_blocks = True

            def __exit__(self_b, *exc):
                em.stmt("}")
                em.cse = self_b.saved
                return False

        return _Block()

    def out_filler(self, k: int, x: V, which: int = 1):
        """An output that depends on no computed value (a structural zero): text() spreads these
        evenly over the whole body, so that they drain while the arithmetic runs.  All waves of a
        launch run the same program in step: a burst of stores at one place is a phase in which
        every wave waits for HBM and no wave computes."""
        assert x.is_zero()
        mac = "OUTZ" if which == 1 else "OUT%dZ" % which
        self.fillers.append("%s(%d);" % (mac, k))
        self.stats["st"] += 1

    # ---- arithmetic with folding -------------------------------------------------------------
    def _emit(self, key: tuple, expr: str, neg: bool = False, kind: str = "fp", cdef=None) -> V:
        if key in self.cse:
            v = self.cse[key]
            return -v if neg else v
        if (cdef is not None and HOIST_CONST and all(_is_const(x) for x in cdef[1])
                and sum(1 for p in self.visits[-1].pre if p[1] == "ct") < HOIST_CAP):
            # every operand is a model constant (or a literal): the value is one too.  It becomes an
            # entry of the packed constant table, evaluated once on the host (brbd_model_create)
            # instead of once per wave per launch.
            refs = []
            for x in cdef[1]:
                if isinstance(x, str):
                    refs.append((int(x.lstrip("-")[1:]), x[0] == "-"))
                else:  # a literal operand is a table entry of its own (never loaded by the kernel)
                    if float(x) not in self.lit_index:
                        self.lit_index[float(x)] = len(self.const_list)
                        self.const_list.append(("lit", float(x)))
                    refs.append((self.lit_index[float(x)], False))
            ident = ("expr", cdef[0]) + tuple(refs)
            k = len(self.const_list)
            self.const_list.append(ident)
            name = "k%d" % k
            self.pre("const double %s = CT(%d);" % (name, k))
            self.stats["ct"] += 1
            self.stats["hoisted"] = self.stats.get("hoisted", 0) + 1
            v = V(name)
            self.cse[key] = v
            return -v if neg else v
        name = self._tmp()
        self.stmt("const double %s = %s;" % (name, expr))
        self.stats[kind] += 1
        v = V(name)
        self.cse[key] = v
        return -v if neg else v

    def mul(self, a: V, b: V) -> V:
        if a.is_num and b.is_num:
            return num(a.num * b.num)
        if a.is_num:
            a, b = b, a
        if b.is_num:
            if b.num == 0.0:
                return ZERO
            if b.num == 1.0:
                return a
            if b.num == -1.0:
                return -a
            neg = a.neg != (b.num < 0)
            c = abs(b.num)
            return self._emit(("*", a.name, ("#", c)), "%s * %s" % (a.name, _lit(c)), neg, cdef=("mul", [a.name, c]))
        neg = a.neg != b.neg
        x, y = sorted((a.name, b.name))
        return self._emit(("*", x, y), "%s * %s" % (x, y), neg, cdef=("mul", [x, y]))

    def add(self, a: V, b: V) -> V:
        if a.is_num and b.is_num:
            return num(a.num + b.num)
        if a.is_zero():
            return b
        if b.is_zero():
            return a
        if a.is_num:
            a, b = b, a
        if b.is_num:  # var + literal
            if a.neg:  # -(x) + c = -(x - c)
                return self._emit(("+#", a.name, -b.num), "%s + %s" % (a.name, _lit(-b.num)), True,
                                  cdef=("add", [a.name, -b.num]))
            return self._emit(("+#", a.name, b.num), "%s + %s" % (a.name, _lit(b.num)), cdef=("add", [a.name, b.num]))
        if a.neg and b.neg:
            x, y = sorted((a.name, b.name))
            return self._emit(("+", x, y), "%s + %s" % (x, y), True, cdef=("add", [x, y]))
        if not a.neg and not b.neg:
            x, y = sorted((a.name, b.name))
            return self._emit(("+", x, y), "%s + %s" % (x, y), cdef=("add", [x, y]))
        if a.neg:
            a, b = b, a  # a positive, b negative: a - b
        if a.name == b.name:
            return ZERO
        return self._emit(("-", a.name, b.name), "%s - %s" % (a.name, b.name), cdef=("add", [a.name, "-" + b.name]))

    def sub(self, a: V, b: V) -> V:
        return self.add(a, -b)

    def fma(self, a: V, b: V, c: V) -> V:
        """a*b + c with one rounding (folds when a factor is 0 / +-1 or c is 0)."""
        if c.is_zero():
            return self.mul(a, b)
        if a.is_num and b.is_num:
            return self.add(num(a.num * b.num), c)
        if a.is_num:
            a, b = b, a
        if b.is_num and b.num in (0.0, 1.0, -1.0):
            return self.add(self.mul(a, b), c)
        # normalise signs: result = s * FMA(x, y, z)
        pneg = a.neg != (b.neg if not b.is_num else (b.num < 0))
        bx = b.name if not b.is_num else _lit(abs(b.num))
        if c.is_num:
            cneg, cx = c.num < 0, _lit(abs(c.num))
        else:
            cneg, cx = c.neg, c.name
        x, y = (a.name, bx) if b.is_num else tuple(sorted((a.name, bx)))
        yv = abs(b.num) if b.is_num else y
        cv = abs(c.num) if c.is_num else cx
        if pneg and cneg:
            return self._emit(("fma", x, y, cx, 0), "FMA(%s, %s, %s)" % (x, y, cx), True, cdef=("fma", [x, yv, cv]))
        if not pneg and not cneg:
            return self._emit(("fma", x, y, cx, 0), "FMA(%s, %s, %s)" % (x, y, cx), cdef=("fma", [x, yv, cv]))
        if pneg:  # -x*y + c
            return self._emit(("fma", x, y, cx, 1), "FMA(-%s, %s, %s)" % (x, y, cx), cdef=("fma", ["-" + x, yv, cv]))
        return self._emit(("fma", x, y, cx, 1), "FMA(-%s, %s, %s)" % (x, y, cx), True,
                          cdef=("fma", ["-" + x, yv, cv]))  # x*y - c = -(-x*y + c)

    def dot(self, a: Sequence[V], b: Sequence[V], init: V = ZERO) -> V:
        """init + sum a_k b_k as an FMA chain (terms with an exact zero factor vanish)."""
        acc = init
        for x, y in zip(a, b):
            acc = self.fma(x, y, acc)
        return acc

    def rcp(self, a: V) -> V:
        if a.is_num:
            return num(1.0 / a.num)
        return self._emit(("rcp", a.name), "RCP(%s)" % a.name, a.neg, "trans", cdef=("rcp", [a.name]))

    def rsqrt(self, a: V) -> V:
        assert not a.is_num and not a.neg
        return self._emit(("rsq", a.name), "RSQRT(%s)" % a.name, False, "trans", cdef=("rsqrt", [a.name]))

    def sincos(self, a: V) -> Tuple[V, V]:
        if a.is_num:
            return num(math.sin(a.num)), num(math.cos(a.num))
        assert not a.neg
        n = self._tmp()
        s, c = "s_" + n, "c_" + n
        self.stmt("double %s, %s; SINCOS(%s, %s, %s);" % (s, c, a.name, s, c))
        self.stats["trans"] += 1
        return V(s), V(c)

    # ---- parking -------------------------------------------------------------------------------
    tag_parks = False  # checker hook (tests/test_codegen_cpu.py): name the park behind every park access

    def _tag(self, pk: "Park") -> str:
        return " /*P%d*/" % pk.pid if Emitter.tag_parks else ""

    def park(self, vals: Sequence[V], hint: str = "stack") -> Park:
        """Hand values to the storage plan until a later visit asks for them back."""
        pk = Park(len(self.parks), list(vals), self.vi, hint)
        self.parks.append(pk)
        if self.plan is not None:
            pk.cls, pk.slot = self.plan[pk.pid]
            if pk.cls != "reg":
                live = [v for v in pk.vals if not v.is_num]
                mac = "PK_LDS_ST2" if pk.cls == "lds" else "PK_GLB_ST2"
                for j in range(0, len(live), 2):
                    a = live[j].name
                    b = live[j + 1].name if j + 1 < len(live) else a
                    self.stmt("%s(%d, %s, %s);%s" % (mac, pk.slot // 2 + j // 2, a, b, self._tag(pk)))
                    self.stats["st"] += 1
        return pk

    def unpark(self, pk: Park) -> List[V]:
        """Values of a park, valid in the current visit (may be called in several visits)."""
        pk.end = max(pk.end, self.vi)
        if self.plan is None or pk.cls == "reg" or self.vi <= pk.start + 1:
            # (a value consumed by the very next visit is still in registers: its early load would
            # be issued before the store)
            if self.plan is None or self.vi <= pk.start + 1 or not self.opaque_regs:
                return list(pk.vals)
            # A register-parked value comes back OPAQUE: otherwise the compiler proves that what is
            # recomputed from it (a joint rotation from its sin / cos ...) equals what the forward
            # visit computed and keeps all of that alive instead.
            out = []
            for v in pk.vals:
                if v.is_num:
                    out.append(v)
                    continue
                n = self._tmp()
                self.stmt("double %s = %s; OPAQUE(%s);" % (n, v.name, n))
                out.append(V(n, v.neg))
            return out
        live = [v for v in pk.vals if not v.is_num]
        mac = "PK_LDS_LD2" if pk.cls == "lds" else "PK_GLB_LD2"
        names = []
        for j in range(0, len(live), 2):
            a = self._tmp()
            b = self._tmp()
            self.pre("double %s, %s; %s(%d, %s, %s);%s" % (a, b, mac, pk.slot // 2 + j // 2, a, b, self._tag(pk)), pk.cls,
                     pk.start + 1)
            self.stats["ld"] += 1
            names += [a, b]
        out, j = [], 0
        for v in pk.vals:
            if v.is_num:
                out.append(v)
            else:
                out.append(V(names[j], v.neg))
                j += 1
        return out

    # ---- grouped input loads -----------------------------------------------------------------------
    def group_inputs(self, gap: int = 1500, width: int = 16, commit: bool = False):
        """Rewrite the element loads of q / v / the third input so that the INSTANCE_MAJOR kernels can
        fetch them a group of ``width`` consecutive elements at a time, when the walk first needs one
        of them, instead of staging the whole tile before the first visit.  Every element load
        `IN_Q(k)` becomes `IN_Q2(k, g, j, n)` = element j of fetch n of group g, preceded (at the
        usual prefetch distance) by one `GRP_Q(g, n)` per fetch; in the other layouts GRP_* is empty
        and IN_*2 is the plain element load.  A group whose uses lie more than ``gap`` statements
        apart is fetched again (n counts the fetches) rather than held in registers in between.

        width 16 (csrc/brbd_spec.hpp: SpecIn<2, ., ., 2>): elements 16 g .. 16 g + 15 of a lane's
        column are 128 consecutive bytes, read with 8 back-to-back b128 loads per lane.
        ``commit`` (SpecIn<2, ., ., 3>, width 8): GRP_* issues the group's coalesced loads of the whole
        tile (8 dofs x 64 instances, each lane receives a share of it), and `GRPC_Q(g, n)` -- placed
        where the first element is consumed, like the element reads themselves -- passes them
        through the LDS window so that each lane ends up with its own instance's 8 elements."""
        pat = re.compile(r"^const double (\w+) = IN_([QVX])\((\d+)\);$")
        start, pos = [], 0
        for v in self.visits:
            start.append(pos)
            pos += len(v.body)
        last: Dict[tuple, Tuple[int, int]] = {}   # (matrix, group) -> (position of the last use, fetch number)
        for vi, v in enumerate(self.visits):
            new_pre = []
            for line, kind, earliest in v.pre:
                m = pat.match(line)
                if not m:
                    new_pre.append((line, kind, earliest))
                    continue
                name, which, k = m.group(1), m.group(2), int(m.group(3))
                g, j = divmod(k, width)
                key = (which, g)
                if key not in last or start[vi] - last[key][0] > gap:
                    n = last[key][1] + 1 if key in last else 0
                    new_pre.append(("GRP_%s(%d, %d);" % (which, g, n), kind, earliest))
                    if commit:
                        new_pre.append(("GRPC_%s(%d, %d);" % (which, g, n), "use", vi))
                else:
                    n = last[key][1]
                last[key] = (start[vi], n)
                new_pre.append(("const double %s = IN_%s2(%d, %d, %d, %d);" % (name, which, k, g, j, n),
                                "use" if commit else kind, vi if commit else earliest))
            v.pre = new_pre

    # ---- output ----------------------------------------------------------------------------------
    def text(self, ahead: Optional[Dict[str, int]] = None, fine: bool = False) -> str:
        """Concatenate the visits.  An early load of visit k is issued at the start of the latest
        visit that lies at least ahead[kind] STATEMENTS (~ instructions, 4 cycles each with one
        wave per SIMD) before visit k, but not before its ``earliest`` visit.  With a single wave per
        SIMD nothing else hides memory latency: HBM wants thousands of cycles of distance, LDS and
        the scalar cache a few hundred.  (Visits differ in length by 20x, so a distance counted in
        visits either starves the long ones or floods the registers in the short ones.)
        ``fine``: place each early load exactly ahead[kind] statements before its first use, inside
        the body of the earlier visit (or of its own), instead of bunching them at visit starts
        (lower register peaks when many short visits each consume a parked block)."""
        ahead = ahead or AHEAD
        n = len(self.visits)
        size = [len(v.body) for v in self.visits]
        if fine:
            assert not self.has_blocks, "statement-level placement could land inside a conditional block"
            start = [0] * (n + 1)
            for k in range(n):
                start[k + 1] = start[k] + size[k]
            ins: Dict[int, List[str]] = {}
            for k in range(n):
                # first statement of the visit that reads each name
                first: Dict[str, int] = {}
                for j, st in enumerate(self.visits[k].body):
                    for name in _NAME.findall(st):
                        first.setdefault(name, j)
                for line, kind, earliest in self.visits[k].pre:
                    m = _DECL.match(line)
                    names = [x.strip() for x in m.group(1).split(",")] if m else []
                    use = min([first[x] for x in names if x in first] or [0])
                    pos = max(start[min(max(earliest, 0), k)], start[k] + use - ahead.get(kind, 0))
                    ins.setdefault(pos, []).append(line)
            lines = []
            for k in range(n):
                lines += ins.get(start[k], []) if size[k] else []
                lines.append("/* ---- visit %d: %s ---- */" % (k, self.visits[k].label))
                for j, st in enumerate(self.visits[k].body):
                    if j > 0:
                        lines += ins.get(start[k] + j, [])
                    lines.append(st)
                if not size[k]:  # (an empty visit shares its position with the next one)
                    pass
                lines.append("VISIT_END();")
            assert not ins.get(start[n]), "early load after the last statement"
            return "\n".join(self._spread(lines)) + "\n"
        early: List[List[str]] = [[] for _ in range(n)]
        for k in range(n):
            for line, kind, earliest in self.visits[k].pre:
                want = ahead.get(kind, 0)
                j, dist = k, 0
                while j > 0 and j > earliest and dist < want:
                    j -= 1
                    dist += size[j]
                early[max(j, 0)].append(line)
        lines: List[str] = []
        for k in range(n):
            lines += early[k]
            lines.append("/* ---- visit %d: %s ---- */" % (k, self.visits[k].label))
            lines += self.visits[k].body
            lines.append("VISIT_END();")
        return "\n".join(self._spread(lines)) + "\n"

    def _spread(self, lines: List[str], window: int = 128) -> List[str]:
        """Place the filler stores where the body has the least memory traffic of its own
        (water-filling over windows of ``window`` statements; a b128 park access counts as two)."""
        if not self.fillers:
            return lines
        at = [n for n, ln in enumerate(lines) if not ln.startswith("/*") and not ln.startswith("VISIT_END")]
        nwin = max(1, len(at) // window)
        load = [0.0] * nwin
        for j, n in enumerate(at):
            ln = lines[n]
            w = min(j * nwin // len(at), nwin - 1)
            if "PK_GLB_" in ln:
                load[w] += 2
            elif ln.startswith("OUT") or "IN_" in ln:  # (fillers are not in ``lines`` yet)
                load[w] += 1
        import heapq
        heap = [(load[w], w) for w in range(nwin)]
        heapq.heapify(heap)
        count = [0] * nwin
        for _ in self.fillers:
            l, w = heapq.heappop(heap)
            count[w] += 1
            heapq.heappush(heap, (l + 1, w))
        where: Dict[int, List[str]] = {}
        it = iter(self.fillers)
        for w in range(nwin):
            lo, hi = w * len(at) // nwin, (w + 1) * len(at) // nwin
            for j in range(count[w]):
                where.setdefault(at[lo + (j * (hi - lo)) // count[w]], []).append(next(it))
        out: List[str] = []
        for n, ln in enumerate(lines):
            out.append(ln)
            out += where.get(n, [])
        return out


def plan_storage(parks: Sequence[Park], reg_short: int, lds_budget: int, reg_long: int = 0, reg_tail: int = 64,
                 strict: bool = False):
    """Assign every park a storage class: ("reg" | "lds" | "glb", first slot).

    * A park consumed by the very next visit stays in registers (chains hand state over for free).
    * Short-lived parks (hint "stack": path state, contributions of finished subtrees) are placed
      first: registers while fewer than ``reg_short`` such doubles are live, then LDS, then the
      wave-private global workspace.
    * Long-lived parks (hint "hand": what ABA's pass 3 needs from pass 2) then fill what is left:
      up to ``reg_long`` live doubles in registers (the allocator keeps values with distant uses in
      AGPRs), then the LDS slots the stack leaves free during their lifetime, then global.
    LDS / global slots are allocated with interval first-fit at pair granularity; a slot may be
    reused by a park that starts in the visit its previous owner was last read in (those loads are
    issued one visit earlier).  ``strict``: no such sharing of a visit -- for bodies that read a
    park late in a visit in which they have already stored new ones (computeMinverse's backward
    visits), with loads placed at statement granularity.  Returns (plan, lds_doubles, glb_doubles)."""
    plan: Dict[int, Tuple[str, int]] = {}
    used = {"lds": [], "glb": []}  # (slot, size, start, end)
    top = {"lds": 0, "glb": 0}

    def size_of(pk):
        n = sum(1 for v in pk.vals if not v.is_num)
        return n + (n & 1)

    def first_fit(cls, size, start, end, limit):
        if strict:
            busy = sorted((s, sz) for s, sz, a, b in used[cls] if a <= end and start <= b)
        else:
            busy = sorted((s, sz) for s, sz, a, b in used[cls] if a < end and start < b)
        pos = 0
        for s, sz in busy:
            if pos + size <= s:
                break
            pos = max(pos, s + sz)
        if limit is not None and pos + size > limit:
            return -1
        used[cls].append((pos, size, start, end))
        top[cls] = max(top[cls], pos + size)
        return pos

    def reg_live(pk, placed):
        # register-parked doubles live at any time during pk's lifetime (upper bound: all overlapping)
        return sum(size_of(p) for p in placed if p.start < pk.end and pk.start < p.end)

    # ("tail": what ABA's pass 3 keeps while it re-runs the backward visits of a tail; short-lived,
    # at a time when the visits are light on registers)
    for hint, budget in (("stack", reg_short), ("tail", reg_tail), ("hand", reg_long)):
        placed = []
        for pk in sorted((p for p in parks if p.hint == hint), key=lambda p: (p.start, p.pid)):
            size = size_of(pk)
            if size == 0 or pk.end <= pk.start + 1:
                plan[pk.pid] = ("reg", 0)
                continue
            if reg_live(pk, placed) + size <= budget:
                plan[pk.pid] = ("reg", 0)
                placed.append(pk)
                continue
            pos = first_fit("lds", size, pk.start, pk.end, lds_budget)
            if pos >= 0:
                plan[pk.pid] = ("lds", pos)
                continue
            plan[pk.pid] = ("glb", first_fit("glb", size, pk.start, pk.end, None))
    return plan, top["lds"], top["glb"]

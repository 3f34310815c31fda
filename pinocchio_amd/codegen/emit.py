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
from typing import Dict, List, Optional, Sequence, Tuple

MACROS = """
  CT(k)                 k-th double of this kernel's packed constant table (wave-uniform -> s_load)
  IN_Q(k) IN_V(k) IN_X(k)   element k of q / v / third input (a or tau) of this lane's instance
  OUT(k, x)  OUT2(k, x)  OUT3(k, x)  store element k of the 1st / 2nd / 3rd output
  SINCOS(x, s, c)       s = sin(x), c = cos(x)
  RSQRT(x)  RCP(x)      1/sqrt(x), 1/x
  FMA(a, b, c)          a*b + c, one rounding
  PK_LDS_ST2(p, a, b) / PK_LDS_LD2(p, a, b)   pair slot p of the LDS park area
  PK_GLB_ST2(p, a, b) / PK_GLB_LD2(p, a, b)   pair slot p of the global park area
  VISIT_END()           boundary between two visits (scheduling barrier on the GPU)
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


ZERO = V(num=0.0)
ONE = V(num=1.0)


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
        self._pending_pre: List[str] = []  # early loads requested before their visit exists

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

    def pre(self, line: str):
        """A load that may be issued one visit ahead of the current one."""
        self.visits[-1].pre.append(line)

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
        self.pre("const double %s = IN_%s(%d);" % (name, which, k))
        self.stats["ld"] += 1
        v = V(name)
        self.in_cache[key] = v
        return v

    def out(self, k: int, x: V, which: int = 1):
        mac = "OUT" if which == 1 else "OUT%d" % which
        self.stmt("%s(%d, %s);" % (mac, k, x.code()))
        self.stats["st"] += 1

    # ---- arithmetic with folding -------------------------------------------------------------
    def _emit(self, key: tuple, expr: str, neg: bool = False, kind: str = "fp") -> V:
        if key in self.cse:
            v = self.cse[key]
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
            return self._emit(("*", a.name, ("#", c)), "%s * %s" % (a.name, _lit(c)), neg)
        neg = a.neg != b.neg
        x, y = sorted((a.name, b.name))
        return self._emit(("*", x, y), "%s * %s" % (x, y), neg)

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
                return self._emit(("+#", a.name, -b.num), "%s + %s" % (a.name, _lit(-b.num)), True)
            return self._emit(("+#", a.name, b.num), "%s + %s" % (a.name, _lit(b.num)))
        if a.neg and b.neg:
            x, y = sorted((a.name, b.name))
            return self._emit(("+", x, y), "%s + %s" % (x, y), True)
        if not a.neg and not b.neg:
            x, y = sorted((a.name, b.name))
            return self._emit(("+", x, y), "%s + %s" % (x, y))
        if a.neg:
            a, b = b, a  # a positive, b negative: a - b
        if a.name == b.name:
            return ZERO
        return self._emit(("-", a.name, b.name), "%s - %s" % (a.name, b.name))

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
        if pneg and cneg:
            return self._emit(("fma", x, y, cx, 0), "FMA(%s, %s, %s)" % (x, y, cx), True)
        if not pneg and not cneg:
            return self._emit(("fma", x, y, cx, 0), "FMA(%s, %s, %s)" % (x, y, cx))
        if pneg:  # -x*y + c
            return self._emit(("fma", x, y, cx, 1), "FMA(-%s, %s, %s)" % (x, y, cx))
        return self._emit(("fma", x, y, cx, 1), "FMA(-%s, %s, %s)" % (x, y, cx), True)  # x*y - c = -(-x*y + c)

    def dot(self, a: Sequence[V], b: Sequence[V], init: V = ZERO) -> V:
        """init + sum a_k b_k as an FMA chain (terms with an exact zero factor vanish)."""
        acc = init
        for x, y in zip(a, b):
            acc = self.fma(x, y, acc)
        return acc

    def rcp(self, a: V) -> V:
        if a.is_num:
            return num(1.0 / a.num)
        return self._emit(("rcp", a.name), "RCP(%s)" % a.name, a.neg, "trans")

    def rsqrt(self, a: V) -> V:
        assert not a.is_num and not a.neg
        return self._emit(("rsq", a.name), "RSQRT(%s)" % a.name, False, "trans")

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
                    self.stmt("%s(%d, %s, %s);" % (mac, pk.slot // 2 + j // 2, a, b))
                    self.stats["st"] += 1
        return pk

    def unpark(self, pk: Park) -> List[V]:
        """Values of a park, valid in the current visit (may be called in several visits)."""
        pk.end = max(pk.end, self.vi)
        if self.plan is None or pk.cls == "reg" or self.vi <= pk.start + 1:
            # (a value consumed by the very next visit is still in registers: its early load would
            # be issued before the store)
            return list(pk.vals)
        live = [v for v in pk.vals if not v.is_num]
        mac = "PK_LDS_LD2" if pk.cls == "lds" else "PK_GLB_LD2"
        names = []
        for j in range(0, len(live), 2):
            a = self._tmp()
            b = self._tmp()
            self.pre("double %s, %s; %s(%d, %s, %s);" % (a, b, mac, pk.slot // 2 + j // 2, a, b))
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

    # ---- output ----------------------------------------------------------------------------------
    def text(self, ahead: int = 1) -> str:
        """Concatenate the visits; the early loads of visit k are issued at the start of visit
        k - ahead."""
        n = len(self.visits)
        lines: List[str] = []
        for k in range(n):
            if k == 0:
                for j in range(0, min(ahead, n)):
                    lines += self.visits[j].pre
            if k + ahead < n:
                lines += self.visits[k + ahead].pre
            lines.append("/* ---- visit %d: %s ---- */" % (k, self.visits[k].label))
            lines += self.visits[k].body
            lines.append("VISIT_END();")
        return "\n".join(lines) + "\n"


def plan_storage(parks: Sequence[Park], reg_budget: int, lds_budget: int,
                 reg_hints: Sequence[str] = ("stack", "hand"), order: Sequence[str] = ()):
    """Assign every park a storage class.  Greedy over start visits: registers while the live
    register-parked doubles stay under ``reg_budget``, then LDS (first-fit interval allocation,
    pair granularity), then the wave-private global workspace.  A slot may be reused by a park that
    starts in the visit its previous owner was last read in (those loads are issued earlier).
    Returns (plan, lds_doubles, glb_doubles)."""
    plan: Dict[int, Tuple[str, int]] = {}
    live_reg: List[Park] = []
    lds_used: List[Tuple[int, int, int]] = []  # (slot, size, end)
    glb_used: List[Tuple[int, int, int]] = []
    lds_max = glb_max = 0

    def first_fit(used, size, start, limit):
        used[:] = [u for u in used if u[2] > start]
        pos = 0
        for s, sz, _ in sorted(used):
            if pos + size <= s:
                break
            pos = max(pos, s + sz)
        if limit is not None and pos + size > limit:
            return -1
        return pos

    for pk in sorted(parks, key=lambda p: (p.start, p.pid)):
        size = sum(1 for v in pk.vals if not v.is_num)
        size += size & 1
        if size == 0 or pk.end <= pk.start + 1:
            # consumed by the very next visit (or never): a register hand-over
            plan[pk.pid] = ("reg", 0)
            continue
        live_reg = [p for p in live_reg if p.end > pk.start]
        reg_now = sum(sum(1 for v in p.vals if not v.is_num) for p in live_reg)
        if pk.hint in reg_hints and reg_now + size <= reg_budget:
            plan[pk.pid] = ("reg", 0)
            live_reg.append(pk)
            continue
        pos = first_fit(lds_used, size, pk.start, lds_budget)
        if pos >= 0:
            plan[pk.pid] = ("lds", pos)
            lds_used.append((pos, size, pk.end))
            lds_max = max(lds_max, pos + size)
            continue
        pos = first_fit(glb_used, size, pk.start, None)
        plan[pk.pid] = ("glb", pos)
        glb_used.append((pos, size, pk.end))
        glb_max = max(glb_max, pos + size)
    return plan, lds_max, glb_max

"""Model compiler: the closing products of computeABADerivatives (algorithm/aba-derivatives.hxx:
449-453), in place:  OUT <- -Minv * OUT (dtau/dq -> ddq/dq),  OUT2 <- -Minv * OUT2 (dtau/dv ->
ddq/dv),  Minv = OUT3 (read only, full symmetric).  All three nv x nv column-major per instance.

What the generator knows and a table-driven kernel does not: dtau/dq and dtau/dv couple dof k with
dof c only when their joints lie on one root path, so column c of either factor has its non-zeros
in the rows of c's ancestors and of its subtree (40 % of a humanoid's rows).  One lane per instance
cannot keep three nv x nv matrices on chip, so one of them streams:

* small models (the upper triangle of Minv fits the register file): Minv is read ONCE and stays in
  registers; the factors stream past column by column, each element read once and its product
  written once;
* larger ones: the columns of a limb share their non-zero rows K: groups of such columns keep their
  factor block (|K| x |C| x 2 doubles) in registers while the rows of Minv restricted to K stream
  past -- Minv is read once per group instead of once per two columns.
"""
from __future__ import annotations

from typing import List

from . import joints as J
from .emit import ZERO, Emitter

import os

MINV_REGS = 180   # doubles of Minv's upper triangle that may stay in registers
BLOCK_REGS = int(os.environ.get("BRBD_CG_PROD_BLOCK", "150"))    # doubles of a column group's factor blocks (both matrices)


def _coupled(mv: J.ModelView):
    """nz[c] = sorted rows k with a structural non-zero in column c of dtau/dq, dtau/dv"""
    nvt = mv.model.nv
    joint_of = [0] * nvt
    for i in range(1, mv.n):
        for k in range(mv.nv(i)):
            joint_of[mv.idx_v[i] + k] = i

    def above(a, b):
        while b > 0:
            if b == a:
                return True
            b = mv.parents[b]
        return False

    return [[k for k in range(nvt) if above(joint_of[k], joint_of[c]) or above(joint_of[c], joint_of[k])]
            for c in range(nvt)]


def generate(mv: J.ModelView, e: Emitter) -> Emitter:
    nvt = mv.model.nv
    nz = _coupled(mv)

    def minv(r, k):  # Minv(r, k) of the symmetric matrix: always read from the upper triangle
        r, k = min(r, k), max(r, k)
        return e.inp_out(3, k * nvt + r)

    def column(which, c, rows):
        return {k: e.inp_out(which, c * nvt + k) for k in rows}

    if nvt * (nvt + 1) // 2 <= MINV_REGS:
        e.begin_visit("MINV")
        M = {(r, k): minv(r, k) for k in range(nvt) for r in range(k + 1)}
        for which in (1, 2):
            for c in range(nvt):
                e.begin_visit("COL %d of %d" % (c, which))
                d = column(which, c, nz[c])
                for r in range(nvt):
                    s = ZERO
                    for k in nz[c]:
                        s = e.fma(M[(min(r, k), max(r, k))], d[k], s)
                    e.out(c * nvt + r, -s, which)
        return e
    # column groups: consecutive columns, greedily, while the union of their rows keeps the block small
    groups: List[List[int]] = []
    cur: List[int] = []
    rows: set = set()
    for c in range(nvt):
        u = rows | set(nz[c])
        if cur and 2 * len(u) * (len(cur) + 1) > BLOCK_REGS:
            groups.append(cur)
            cur, u = [], set(nz[c])
        cur.append(c)
        rows = u
    if cur:
        groups.append(cur)
    for g in groups:
        K = sorted(set(k for c in g for k in nz[c]))
        e.begin_visit("GROUP %d..%d" % (g[0], g[-1]))
        d1 = {c: column(1, c, nz[c]) for c in g}
        d2 = {c: column(2, c, nz[c]) for c in g}
        for r0 in range(0, nvt, 2):
            e.begin_visit("GROUP %d..%d rows %d.." % (g[0], g[-1], r0))
            for r in range(r0, min(r0 + 2, nvt)):
                m = {k: minv(r, k) for k in K}
                for c in g:
                    for which, d in ((1, d1[c]), (2, d2[c])):
                        s = ZERO
                        for k in nz[c]:
                            s = e.fma(m[k], d[k], s)
                        e.out(c * nvt + r, -s, which)
    return e

"""Model compiler: computeMinverse (algorithm/aba.hxx:841-877; pass 1 :626-654, pass 2 :668-728,
pass 3 :802-833) as straight-line code.  Output OUT: Minv, nv x nv column-major per instance
(element (r, c) at c * nv + r): the upper triangle and the full diagonal block of every multi-dof
joint, like ``data.Minv``; the strictly lower part is zero, or -- kernel variants with WANT bit 1,
what computeABADerivatives needs (aba-derivatives.hxx:449-450) -- the mirror image of the upper
part (OUTSYM).

The reference keeps one 6 x nv block ``Fcrb`` per joint.  Here the columns of those blocks travel
as parks of 6 * nv_j doubles, one per (block owner, joint j whose dofs are the columns):

* passes 1 + 2 fused in one depth-first walk like the ABA kernel.  Backward visit of joint i:
  U = Ia J, D, Dinv, UDinv; Minv(i, i) = Dinv; for every joint j below i:
  Minv(i, j) = -(J Dinv)^T F0[j] (:705-707) and, unless i hangs off the universe,
  F0[j] += U Minv(i, j) (:709-715; F0 of i's own columns is U Dinv = UDinv, :717-721); the
  articulated inertia Ia - UDinv U^T goes to the parent (:723-727).  A joint attached to the
  universe has its rows final here (pass 3 does not change them) and starts pass 3 at once:
  F[j] = J Minv(i, j) (:827-829).
* pass 3, parents before children: Minv(i, j) -= UDinv^T F_parent[j] for every joint j >= i of
  the tree (:819-825: also the columns right of i's subtree, which pass 2 left at zero), then
  F_i[j] = F_parent[j] + J Minv(i, j) for the joints j that i's children will ask for (:827-832).

The work per column is independent.  Pass 2 takes one sub-visit per (i, j) with the joint's own
U / UDinv / J in registers; pass 3 walks a whole chain of single-child joints (a limb) per column
block, so a column of F is read once per limb and parked again only where the tree branches.
Needs depth-first numbering (every subtree's dofs contiguous), like the derivatives.
"""
from __future__ import annotations

from . import algebra as A
from . import joints as J
from .aba import spd_inverse
from .emit import ZERO, Emitter


def generate(mv: J.ModelView, e: Emitter) -> Emitter:
    n, nvt = mv.n, mv.model.nv
    last = list(range(n))  # last joint of i's subtree
    for i in range(n - 1, 0, -1):
        if mv.parents[i] > i:
            raise ValueError("computeMinverse kernels need parents numbered before their children")
        p = mv.parents[i]
        last[p] = max(last[p], last[i])
    for i in range(1, n):  # depth-first numbering: subtree of i = joints i .. last[i], dofs contiguous
        sub = [j for j in range(1, n) if _is_below(mv, j, i)]
        if sub != list(range(i, last[i] + 1)) or any(
                mv.idx_v[j + 1] != mv.idx_v[j] + mv.nv(j) for j in range(i, last[i])):
            raise ValueError("computeMinverse kernels need depth-first joint numbering")
    written = set()

    def put(r, c, val, mirror):
        e.out(c * nvt + r, val, 1)
        written.add((r, c))
        if mirror:
            e.out_sym(r * nvt + c, val)
            written.add((c, r))

    kin = {}      # joint -> Park R (9), t (3): read by the children's forward and its own backward visit
    contrib = {}  # parent -> Park of the running sum of its finished children's articulated inertias (21)
    f0 = {}       # joint i -> {j: Park 6 nv_j}: F0 columns of the joints of i's subtree, after BWD(i)
    mrow = {}     # joint i -> Park [j][c][r]: Minv(i rows, columns of j) for j in subtree(i), pass-2 values
    hand = {}     # joint i -> Park UDinv (6 nv_i) + J (6 nv_i), for pass 3
    f3 = {}       # joint i -> {j: Park 6 nv_j}: pass-3 F_i columns of the joints j its children ask for

    def cols(j):
        return range(mv.idx_v[j], mv.idx_v[j] + mv.nv(j))

    def backward(i, R, t):
        p = mv.parents[i]
        nv, iv = mv.nv(i), mv.idx_v[i]
        Jc = [A.act_motion(e, R, t, col) for col in J.subspace(e, mv, i)]       # :650
        m, c, I = mv.inertia(e, i)
        cw, Iw = A.inertia_to_world(e, R, t, c, I)                               # :652
        Ia = A.inertia_to_sym6(e, m, cw, Iw)                                     # :653
        if i in contrib:
            Ia = A.vadd(e, Ia, e.unpark(contrib.pop(i)))
        U = [A.sym6_mulvec(e, Ia, Jc[k]) for k in range(nv)]                     # :689
        D = [[ZERO] * nv for _ in range(nv)]
        for r in range(nv):
            for cc in range(r, nv):
                d = A.dot6(e, Jc[r], U[cc])                                      # :690
                D[r][cc] = D[cc][r] = e.add(d, mv.c(e, i, "arm", iv + r)) if r == cc else d  # :693
        Dinv = [[e.rcp(D[0][0])]] if nv == 1 else spd_inverse(e, D)              # :695
        UD = [[e.dot([U[r][k] for r in range(nv)], [Dinv[r][cc] for r in range(nv)]) for k in range(6)]
              for cc in range(nv)]                                               # :696
        below = [j for j in range(i + 1, last[i] + 1)]
        SD = [[e.dot([Jc[r][k] for r in range(nv)], [Dinv[r][cc] for r in range(nv)]) for k in range(6)]
              for cc in range(nv)] if below else None                            # :703
        if p == 0:
            for r in range(nv):                                                  # :698
                for cc in range(nv):
                    put(iv + r, iv + cc, Dinv[r][cc], False)
        fin = {}
        for ch in mv.children[i]:
            fin.update(f0.pop(ch))
        fout, f3out, keep = {}, {}, []
        if p > 0:
            keep += [Dinv[r][cc] for cc in range(nv) for r in range(nv)]        # (own block, [c][r])
        for j in below:
            e.begin_visit("BWD %d x %d" % (i, j))
            f = e.unpark(fin[j])
            fnew, f3new = [], []
            for k, cj in enumerate(cols(j)):
                fc = f[6 * k:6 * k + 6]
                x = [-A.dot6(e, SD[r], fc) for r in range(nv)]                   # :705-707
                if p == 0:
                    for r in range(nv):
                        put(iv + r, cj, x[r], True)
                    g = A.zeros(6)
                    for r in range(nv):
                        g = A.vaxpy(e, x[r], Jc[r], g)                           # :827-829
                    f3new += g
                else:
                    keep += x
                    g = fc
                    for r in range(nv):
                        g = A.vaxpy(e, x[r], U[r], g)                            # :709-715
                    fnew += g
            if p == 0:
                f3out[j] = e.park(f3new, "stack")
            else:
                fout[j] = e.park(fnew, "stack")
        if p == 0:
            if below:
                f3[i] = f3out
            return
        fout[i] = e.park([x for cc in range(nv) for x in UD[cc]], "stack")       # :717-721 (U Dinv = UDinv)
        f0[i] = fout
        mrow[i] = e.park(keep, "hand")
        hand[i] = e.park([x for cc in range(nv) for x in UD[cc]] + [x for cc in range(nv) for x in Jc[cc]], "hand")
        Ia2 = list(Ia)
        for a_ in range(6):
            for b_ in range(a_, 6):
                s = Ia[A.sidx(a_, b_)]
                for cc in range(nv):
                    s = e.fma(-UD[cc][a_], U[cc][b_], s)                         # :723-725
                Ia2[A.sidx(a_, b_)] = s
        if p in contrib:
            Ia2 = A.vadd(e, e.unpark(contrib[p]), Ia2)                           # :726
        contrib[p] = e.park(Ia2, "stack")

    for kind, i in J.visit_program(mv):
        p = mv.parents[i]
        leaf = not mv.children[i]
        if kind == "F":
            e.begin_visit("FWD %d%s" % (i, " + BWD (leaf)" if leaf else ""))
            Rl, tl, _ = J.joint_transform(e, mv, i)                              # :636-639
            if p == 0:
                R, t = Rl, tl
            else:
                st = e.unpark(kin[p])
                Rp, tp = [st[0:3], st[3:6], st[6:9]], st[9:12]
                R, t = A.rot_mul(e, Rp, Rl), A.rot_vec(e, Rp, tl, tp)            # :641-644
            if leaf:
                backward(i, R, t)
            else:
                kin[i] = e.park([x for row in R for x in row] + list(t), "stack")
        elif not leaf:
            e.begin_visit("BWD %d" % i)
            st = e.unpark(kin[i])
            backward(i, [st[0:3], st[3:6], st[6:9]], st[9:12])

    # ---- pass 3 (:802-833), parents before children.  A chain of joints with one child each (a limb)
    # is walked per COLUMN BLOCK: the block's F column comes in once from the chain's parent, every
    # joint of the chain updates its rows of Minv and the column in registers, and the column is
    # parked again only below the chain's last joint if that one branches.  (Joint by joint, every
    # joint of a limb would read and write the F columns of everything to its right.)
    CHAIN_MAX = 8
    done = set()
    for i0 in range(1, n):
        if mv.parents[i0] == 0 or i0 in done:
            continue
        chain = [i0]
        while len(mv.children[chain[-1]]) == 1 and len(chain) < CHAIN_MAX:
            chain.append(mv.children[chain[-1]][0])
        done.update(chain)
        p = mv.parents[i0]
        r_ = i0
        while mv.parents[r_] > 0:
            r_ = mv.parents[r_]
        tree_last = last[r_]
        tail = chain[-1]
        e.begin_visit("PASS3 %d..%d" % (i0, tail))
        UD, Jc, mr, pos = {}, {}, {}, {}
        for i in chain:
            nv = mv.nv(i)
            h = e.unpark(hand[i])
            UD[i] = [h[6 * k:6 * k + 6] for k in range(nv)]
            Jc[i] = [h[6 * nv + 6 * k:6 * nv + 6 * k + 6] for k in range(nv)]
            mr[i] = e.unpark(mrow[i])
            pos[i], off = {}, 0
            for j in range(i, last[i] + 1):  # order of `keep`: own block first, then the joints below
                pos[i][j] = off
                off += nv * mv.nv(j)
        fout = {}
        first = True
        for j in range(i0, tree_last + 1):
            if not first:
                e.begin_visit("PASS3 %d..%d x %d" % (i0, tail, j))
            first = False
            fcols = e.unpark(f3[p][j])
            want_f = bool(mv.children[tail]) and j > tail
            fnew = []
            for k, cj in enumerate(cols(j)):
                fc = fcols[6 * k:6 * k + 6]
                for i in chain:
                    if j < i:      # (columns left of joint i: not its business, nor its successors')
                        break
                    nv, iv = mv.nv(i), mv.idx_v[i]
                    x = []
                    for r in range(nv):
                        base = mr[i][pos[i][j] + k * nv + r] if j <= last[i] else ZERO
                        x.append(e.sub(base, A.dot6(e, UD[i][r], fc)))           # :819-825
                        put(iv + r, cj, x[r], j != i)
                    if i != tail or want_f:
                        for r in range(nv):
                            fc = A.vaxpy(e, x[r], Jc[i][r], fc)                  # :827-832
                if want_f:
                    fnew += fc
            if want_f:
                fout[j] = e.park(fnew, "stack")
        if fout:
            f3[tail] = fout
    # everything else is structurally zero: other trees, and (unless mirrored) the strictly lower part
    for cc in range(nvt):
        for r in range(nvt):
            if (r, cc) not in written:
                e.out_filler(cc * nvt + r, ZERO, 1)
    return e


def _is_below(mv: J.ModelView, j: int, i: int) -> bool:
    while j > 0:
        if j == i:
            return True
        j = mv.parents[j]
    return False

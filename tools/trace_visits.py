#!/usr/bin/env python3
"""Per-visit cycle stamps of wave 0 (a -DBRBD_SPEC_TRACE build, BRBD_TRACE=1) next to the number of
statements the generator emitted for each visit.  Usage: trace_visits.py run   (on the GPU box, with
BRBD_LIB pointing at the trace build) | trace_visits.py table <stderr log> <rnea.inc> <aba.inc>"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np, torch
    from pinocchio_amd import parallel, sample_models as S
    m = S.simple_humanoid(); B = int(os.environ.get("TRACE_B", "65536"))
    pool = parallel.ModelPool(m, 1)
    rng = np.random.default_rng(0)
    q = torch.from_numpy(np.ascontiguousarray(m.random_configuration(rng, B))).cuda()
    v = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
    a = torch.from_numpy(rng.uniform(-1, 1, (m.nv, B))).cuda()
    for _ in range(3):
        tau = parallel.rneaInParallel(1, pool, q, v, a)
        parallel.abaInParallel(1, pool, q, v, tau)
    torch.cuda.synchronize()


def table(log, rnea_inc, aba_inc):
    traces = {}
    for ln in open(log):
        mm = re.match(r"BRBD_TRACE algo (\d) layout (\d):(.*)", ln)
        if mm:
            traces[int(mm.group(1))] = [int(x) for x in mm.group(3).split()]   # (last call wins)
    for algo, inc in ((0, rnea_inc), (1, aba_inc)):
        visits, cur = [], None
        for ln in open(inc):
            mm = re.match(r"/\* ---- visit (\d+): (.*) ---- \*/", ln)
            if mm:
                cur = [mm.group(2), 0, 0]
                visits.append(cur)
            elif cur is not None and not ln.startswith("VISIT_END") and not ln.startswith("/*"):
                cur[1] += 1
                cur[2] += 1 if ("PK_GLB" in ln or "IN_" in ln or ln.startswith("OUT")) else 0
        tr = traces.get(algo, [])
        # stamp k is taken at the END of visit k; tr[k] = stamp[k] - stamp[k-1] (tr[0] = 0)
        print("algo %d: %d visits, %d stamps, %d statements, %d cycles between first and last stamp" %
              (algo, len(visits), len(tr), sum(v[1] for v in visits), sum(tr)))
        rows = []
        for k, v in enumerate(visits):
            c = tr[k] if k < len(tr) else 0
            rows.append((k, v[0], v[1], v[2], c, c / max(v[1], 1)))
        for r in rows:
            print("%4d %-34s stmts %5d mem %4d cycles %7d  %.1f c/stmt" % r)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        table(*sys.argv[2:5])

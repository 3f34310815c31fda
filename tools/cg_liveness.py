"""Register-pressure profile of a generated body in program order (doubles live after each
statement), per visit.  Usage: python tools/cg_liveness.py file.inc"""
import re, sys
txt = open(sys.argv[1]).read().splitlines()
ident = re.compile(r"\b([a-z][a-z]?_?[0-9]+(?:_[0-9]+)?|[sc]_t[0-9]+)\b")
defs, uses = {}, {}
stm = []
for ln, line in enumerate(txt):
    if line.startswith("/*") or line.startswith("VISIT_END"):
        stm.append((line, [], [])); continue
    ids = ident.findall(line)
    d = []
    m = re.match(r"(?:const )?double ([^=;]+?)(?: =|;)", line)
    if m:
        d = [x.strip() for x in m.group(1).split(",")]
    u = [x for x in ids if x not in d]
    stm.append((line, d, u))
    for x in d: defs.setdefault(x, ln)
    for x in u: uses[x] = ln
live = set(); prof = []; visit = "pre"; vmax = {}
for ln, (line, d, u) in enumerate(stm):
    if line.startswith("/* ---- visit"):
        visit = line
    for x in d:
        if x in uses and not x.startswith("k"): live.add(x)
    for x in u:
        if uses.get(x) == ln: live.discard(x)
    vmax[visit] = max(vmax.get(visit, 0), len(live))
for v, m in vmax.items():
    print(m, v)
print("max", max(vmax.values()))
if len(sys.argv) > 2:
    # long-lived values: defined in visit a, last used in visit b > a + 1
    vis_of = []; cur = -1
    for line, d, u in stm:
        if line.startswith("/* ---- visit"):
            cur = int(line.split()[3].rstrip(":"))
        vis_of.append(cur)
    from collections import Counter
    span = Counter()
    for x, ln in defs.items():
        if x in uses and not x.startswith("k"):
            a, b = vis_of[ln], vis_of[uses[x]]
            if b > a + 1: span[(a, b)] += 1
    for (a, b), n in sorted(span.items()):
        print("def visit", a, "last use visit", b, "count", n)

#!/usr/bin/env python3
"""Digest of a BRBD_TRACE dump: cycles per visit next to the visit labels and statement counts of
the generated body.  Usage: trace_digest.py body.inc trace.log [algo-number]"""
import re, sys, collections
body, log = sys.argv[1], sys.argv[2]
algo = sys.argv[3] if len(sys.argv) > 3 else None
labels, sizes = [], []
for l in open(body):
    m = re.match(r"/\* ---- visit \d+: (.*) ---- \*/", l)
    if m:
        labels.append(m.group(1)); sizes.append(0)
    elif labels and not l.startswith("VISIT_END"):
        sizes[-1] += 1
last = None
for l in open(log):
    if l.startswith("BRBD_TRACE") and (algo is None or l.startswith("BRBD_TRACE algo %s " % algo)):
        last = l
d = [int(x) for x in last.split(":")[1].split()]
d = d[1:]  # (the first stamp is the end of visit 0)
print("visits", len(labels), "stamps", len(d) + 1, "total cycles", sum(d))
kind = collections.defaultdict(lambda: [0, 0, 0])
for k in range(1, min(len(labels), len(d) + 1)):
    key = labels[k].split()[0]
    kind[key][0] += 1; kind[key][1] += d[k - 1]; kind[key][2] += sizes[k]
for key, (n, cyc, st) in kind.items():
    print("%-6s n %4d cycles %8d statements %7d  cycles/statement %.2f" % (key, n, cyc, st, cyc / max(st, 1)))
if "-v" in sys.argv:
    for k in range(1, min(len(labels), len(d) + 1)):
        print("%4d %-28s %6d st %7d cyc  %.1f" % (k, labels[k], sizes[k], d[k - 1], d[k - 1] / max(sizes[k], 1)))

#!/bin/bash
# One PMC pass with the LDS counters over bench.py (both boundary layouts run in it): are the
# INSTANCE_MAJOR kernels' staging transpositions bound by LDS bank conflicts?
# Usage (GPU box, repo root): tools/pmc_lds.sh TAG
TAG=${1:-lds}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS \
  --output-format csv -d "$OUT/lds" -o p -- python3 bench.py --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline > "$OUT/lds.json" 2> "$OUT/lds.err" || { echo "pass failed"; tail -3 "$OUT/lds.err"; exit 1; }
python3 - "$OUT/lds" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_rnea" not in k and "k_aba" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in sorted(acc.items()):
    d = {x: c[x] / max(n[(k, x)], 1) for x in c}
    w = d.get("SQ_WAVE_CYCLES", 1)
    import re
    print(re.search(r"k_\w+<[^>]*>", k).group(0), {x.replace("SQ_", ""): round(v / w, 4) if x != "SQ_WAVE_CYCLES" else round(v) for x, v in d.items()})
PY

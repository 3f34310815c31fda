#!/bin/bash
# A/B: run bench.py (no CPU baseline) against each gpurun_out/lib_<tag>.so given; prints kernel times
cd "${GRAFT_REPO_ROOT:-.}"
for t in "$@"; do
  BRBD_LIB=$PWD/ab_libs/lib_$t.so timeout -k 10 120 python bench.py --steps 600 --warmup 300 --no-cpu-baseline > gpurun_out/ab_$t.json 2> gpurun_out/ab_$t.err || { echo "$t FAILED"; tail -3 gpurun_out/ab_$t.err; continue; }
  python - "$t" <<'PY'
import json,sys
t=sys.argv[1]; d=json.load(open(f"gpurun_out/ab_{t}.json"))
c=d["config"]; print(f"{t}: rnea {c['rnea_kernel_ms']*1e3:.1f} us  aba {c['aba_kernel_ms']*1e3:.1f} us  value {d['value']:.3e}  im {c['instance_major_evals_per_s']:.3e} err {c['max_rel_err_aba_of_rnea']:.1e}")
PY
done

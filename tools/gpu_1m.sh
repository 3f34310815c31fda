#!/bin/bash
# The large-batch regime (BASELINE config [3]'s per-GPU slice): kernel trace + PMC passes of the ABA
# kernel at B = 2^20 for simple_humanoid and talos_like.  GRBM_GUI_ACTIVE / duration = effective clock.
# Usage (GPU box, repo root): tools/gpu_1m.sh TAG
TAG=${1:-r03}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
for M in simple_humanoid talos_like; do
  T=${TAG}_1m_$M
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$T -o p -- \
    python3 bench.py --model $M --batch 1048576 --algo aba --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline \
    > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
  rc=$?; echo "$M trace rc=$rc"
  case $rc in 124|134|137|139) exit $rc;; esac
  find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${T}_kernel_stats.csv
  head -4 gpurun_out/${T}_kernel_stats.csv | cut -c1-160
  tools/pmc_passes.sh $T --model $M --batch 1048576 --algo aba || exit 1
  python3 tools/pmc_summary.py $T > gpurun_out/${T}_pmc_summary.json && python3 tools/pmc_digest.py gpurun_out/${T}_pmc_summary.json 16384 > gpurun_out/${T}_pmc_digest.txt
  cat gpurun_out/${T}_pmc_digest.txt | cut -c1-400
done

#!/bin/bash
# rocprofv3 kernel trace + stats of bench.py, then the PMC passes (tools/pmc_passes.sh) and their
# summaries.  Usage (GPU box, repo root): tools/gpu_profile.sh TAG
TAG=${1:-p}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o p -- \
  python3 bench.py --steps 300 --warmup 100 --no-cpu-baseline > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_${TAG}.err
rc=$?; echo "trace rc=$rc"
case $rc in 124|134|137|139) exit $rc;; esac
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats.csv
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
tools/pmc_passes.sh $TAG || exit 1
python3 tools/pmc_summary.py $TAG > gpurun_out/${TAG}_pmc_summary.json && python3 tools/pmc_digest.py gpurun_out/${TAG}_pmc_summary.json > gpurun_out/${TAG}_pmc_digest.txt
python3 tools/pmc_traffic.py gpurun_out/${TAG}_pmc_summary.json gpurun_out/${TAG}_pmc_traffic.json > /dev/null
cat gpurun_out/${TAG}_pmc_digest.txt | head -60

#!/bin/bash
# One GPU session: the GPU test suite (plus any extra test files named on the command line), then
# bench.py; every step under its own time limit; after a fault / abort / time limit nothing else
# is started.  Usage (on the GPU box, from the repo root): tools/gpu_session.sh TAG [extra test files]
TAG=${1:-s}; shift
mkdir -p gpurun_out
bad() { case "$1" in 124|134|137|139) return 0;; *) return 1;; esac; }
timeout -k 10 1500 python -u -m pytest tests "$@" -m gpu -q --timeout 600 --timeout-method thread -p no:cacheprovider \
  > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/${TAG}_tests.log
if bad $rc; then echo "stopping after rc=$rc"; exit $rc; fi
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rc2=$?; echo "bench rc=$rc2"; tail -c 3000 gpurun_out/${TAG}_bench.json; tail -5 gpurun_out/${TAG}_bench.err
exit $(( rc != 0 ? rc : rc2 ))

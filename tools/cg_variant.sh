#!/bin/bash
# Build one experimental variant of the model-compiled kernels for $MODEL (pattern) into
# ab_libs/lib_<tag>.so and print its register usage.  Usage: tools/cg_variant.sh tag [ENV=...]
# (EXTRA = extra hipcc flags, e.g. -DBRBD_SPEC_TRACE; ALGOS = which usage lines to print)
set -e
TAG=$1; shift
MODEL=${MODEL:-simple_humanoid}
cd "$(dirname "$0")/.."
D=/tmp/cgv_$TAG; rm -rf $D; mkdir -p $D/gen
env "$@" python -m pinocchio_amd.codegen.build --out $D/gen --models $MODEL --variants p 2>/dev/null
# kernels whose generated text equals the main build's (pinocchio_amd/csrc/gen, same digest line) reuse
# its objects -- only when no flag / text experiment below is active
REUSE=""
if [ -z "$EXTRA$KATTR$NOBAR$DEFER_OUT$HALF" ]; then
  for f in $D/gen/spec_*.hip; do
    b=$(basename $f)
    if cmp -s $f pinocchio_amd/csrc/gen/$b && [ -f pinocchio_amd/csrc/gen/${b%.hip}.o ]; then
      cp pinocchio_amd/csrc/gen/${b%.hip}.o $D/gen/$b.o
      cp pinocchio_amd/csrc/gen/${b%.hip}.usage $D/gen/$b.usage 2>/dev/null || true
      REUSE="$REUSE $b"
      mv $f $f.reused
    fi
  done
fi
FLAGS="$EXTRA -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -I pinocchio_amd/csrc"
ls $D/gen/spec_*.hip > /dev/null 2>&1 && sed -i 's#"../brbd_spec.hpp"#"brbd_spec.hpp"#' $D/gen/spec_*.hip
# KATTR: extra kernel attribute, e.g. KATTR='__attribute__((amdgpu_waves_per_eu(2,2)))'
# NOBAR="rnea aba": drop the scheduling barriers between the visits of these algorithms' bodies
for alg in $NOBAR; do sed -i 's/^VISIT_END();$//' $D/gen/*_$alg.inc; done
# DEFER_OUT="aba": move the first output's stores of these bodies to the end of the body (experiment:
# do stores interleaved with pass 3's loads stall the in-order vmcnt waits?)
for alg in $DEFER_OUT; do for f in $D/gen/*_$alg.inc; do python3 - "$f" <<'PY'
import sys
p = sys.argv[1]; L = open(p).read().split("\n")
outs = [x for x in L if x.startswith("OUT(")]
rest = [x for x in L if not x.startswith("OUT(")]
while rest and rest[-1] == "": rest.pop()
assert rest[-1] == "VISIT_END();"
open(p, "w").write("\n".join(rest[:-1] + outs + ["VISIT_END();", ""]))
PY
done; done
# HALF="rnea": tiles of 32 instances for these algorithms' kernels (brbd_spec.hpp: BRBD_TILE_LANES; run with BRBD_HALF_TILE=1)
for alg in $HALF; do for f in $D/gen/*_$alg.kern.inc; do sed -i '1i #undef BRBD_TILE_LANES\n#define BRBD_TILE_LANES 32' $f; done; done
if [ -n "$KATTR" ]; then sed -i "s/__launch_bounds__(WAVE)/__launch_bounds__(WAVE) $KATTR/" $D/gen/*.kern.inc; fi
sed -e 's#"gen/registry.inc"#"'$D'/gen/registry.inc"#' -e 's#"../../include/brbd.h"#"brbd.h"#' pinocchio_amd/csrc/brbd_capi.hip > $D/gen/spec_capi.hip
(ls $D/gen/spec_*.hip 2>/dev/null || true) | xargs -r -P 8 -I{} sh -c "/opt/rocm/bin/hipcc $FLAGS -I include -I $D/gen -c -o {}.o {} -Rpass-analysis=kernel-resource-usage 2> {}.usage || { grep error {}.usage; exit 255; }"
cat $D/gen/*.usage | grep -E "Function Name|VGPRs:|AGPRs|Scratch|SGPRs Spill|VGPRs Spill" | grep -A5 "k_rneaIL\|k_abaIL\|k_derivIL" | sed 's/.*remark: *//;s/\[-Rpass.*//' | tr "\n" " " | sed "s/Function Name: _ZN4brbd[0-9]*spec_${MODEL}_p/\n/g"; echo
mkdir -p ab_libs
/opt/rocm/bin/hipcc $FLAGS -shared -o ab_libs/lib_$TAG.so $D/gen/*.o 2>/dev/null
echo "built ab_libs/lib_$TAG.so"

#!/bin/bash
# Composition kernel at 1..4 waves per SIMD (register budget 512/W), on the GPU box.  Every variant is built in a scratch
# copy of the tree (the in-tree objects and libstark252_hip.so are never touched), compiler errors stop the sweep.
set -euo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
for W in 1 2 3 4; do
  V=/tmp/sp_comp_waves_$W
  rm -rf "$V"; mkdir -p "$V"
  cp -r "$ROOT/lambdaworks_cairo_prover_amd" "$ROOT/include" "$ROOT/tools" "$V/"
  ( cd "$V/lambdaworks_cairo_prover_amd/csrc" && rm -f stark_kernels.o ../libstark252_hip.so &&
    make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -DSP_COMP_WAVES=$W" )
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/cw$W -o p -- python3 "$V/tools/prove_bench.py" 149000 8 80 20 > /tmp/cw$W.log 2>&1 )
  echo "waves $W: $(python3 "$ROOT/tools/rocprof_summary.py" "$(find /tmp/cw$W -name '*results.db' | head -1)" | grep 'cairo_composition_kernel<false>' | awk '{print $(NF-3), $(NF-2), $(NF-1)}')  $(tail -1 /tmp/cw$W.log | cut -c1-60)"
done

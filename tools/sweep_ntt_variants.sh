#!/bin/bash
# A/B of compile-time NTT variants on the GPU box: every variant is built in a scratch copy of the tree (the in-tree objects
# and libstark252_hip.so are never touched), compiler errors stop the sweep.
# usage: tools/sweep_ntt_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...   (an empty string = the default build)
set -euo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
i=0
for FLAGS in "$@"; do
  V=/tmp/sp_ntt_variant_$i; i=$((i+1))
  rm -rf "$V"; mkdir -p "$V"
  cp -r "$ROOT/lambdaworks_cairo_prover_amd" "$ROOT/include" "$ROOT/tools" "$V/"
  ( cd "$V/lambdaworks_cairo_prover_amd/csrc" && rm -f ntt.o ../libstark252_hip.so &&
    make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 $FLAGS" )
  echo "== variant [$FLAGS]"
  for rep in 1 2; do
    python3 "$V/tools/ntt_batch_bench.py" 22 1 40 2>&1 | tail -1
    python3 "$V/tools/ntt_batch_bench.py" 22 34 2>&1 | tail -1
  done
  python3 "$V/tools/prove_bench.py" 149000 8 80 20 2>&1 | tail -1
done

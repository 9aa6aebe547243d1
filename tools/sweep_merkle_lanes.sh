#!/bin/bash
# A/B of compile-time Merkle variants on the GPU box (scratch copies of the tree, like sweep_ntt_variants.sh).
# usage: tools/sweep_merkle_lanes.sh "<flags of variant 1>" ...   (an empty string = the default build; e.g. -DSP_MK_LANES_MAX_NODES=8192)
set -euo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
i=0
for FLAGS in "$@"; do
  V=/tmp/sp_mk_variant_$i; i=$((i+1))
  rm -rf "$V"; mkdir -p "$V"
  cp -r "$ROOT/lambdaworks_cairo_prover_amd" "$ROOT/include" "$ROOT/tools" "$ROOT/tests" "$V/"
  ( cd "$V/lambdaworks_cairo_prover_amd/csrc" && rm -f merkle.o ../libstark252_hip.so &&
    make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 $FLAGS" )
  echo "== variant [$FLAGS]"
  python3 "$V/tools/merkle_tail_bench.py" 2>&1 | tail -10
  python3 "$V/tools/merkle_width_bench.py" 2>&1 | tail -6
  python3 "$V/tools/prove_bench.py" 149000 8 80 20 2>&1 | tail -2
done

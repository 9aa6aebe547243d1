#!/bin/bash
# Same-box, interleaved A/B of the round-6 product (a row's reduction through two multiply-adds, csrc/fp.h sp_row_reduce) against the
# carry-chain form of rounds 2 - 5 (-DSP_FE_NO_MAD_REDUCTION): the NTT metric of bench.py and whole proofs at configs[2] / configs[3].
# Run on the GPU box from the repo root:  tools/ab_mad_reduction.sh > gpurun_out/r06_ab_mad_reduction.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SO=lambdaworks_cairo_prover_amd/libstark252_hip.so
cp $SO /tmp/new.so
make -s -C lambdaworks_cairo_prover_amd/csrc clean
make -s -j16 -C lambdaworks_cairo_prover_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -DSP_FE_NO_MAD_REDUCTION" 2>&1 | grep -E "error" | head -3
cp $SO /tmp/old.so
line() { python3 bench.py --proof 0 --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'NTT 2^22: %.4f ms  %.4e butterflies/s  frac %.4f   Merkle 2^23 x 34: %s ms' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('roofline_merkle', {}).get('avg_launch_ms')))"; }
proof() { python3 tools/prove_bench.py $2 $3 80 20 2>&1 | grep "^warm" | sed "s/^/$1 fib($2) blowup $3: /"; }
for rep in 1 2 3; do
  cp /tmp/old.so $SO; line "carry-chain  "; proof "carry-chain  " 149000 8; proof "carry-chain  " 70000 4
  cp /tmp/new.so $SO; line "two-mad      "; proof "two-mad      " 149000 8; proof "two-mad      " 70000 4
done
cp /tmp/new.so $SO

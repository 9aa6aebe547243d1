#!/bin/bash
# composition kernel at 1..4 waves per SIMD (register budget 512/W): rebuilds stark_kernels.o on the GPU box
cd $GRAFT_REPO_ROOT/lambdaworks_cairo_prover_amd/csrc
for W in 2 3 4 1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -DSP_COMP_WAVES=$W -c stark_kernels.hip -o stark_kernels.o 2>/dev/null
  make -s 2>/dev/null
  cd /tmp; export TMPDIR=/tmp
  rm -rf /tmp/cw$W; rocprofv3 --kernel-trace --stats -d /tmp/cw$W -o p -- python3 $GRAFT_REPO_ROOT/tools/prove_bench.py 149000 8 80 20 > /tmp/cw$W.log 2>&1
  echo "waves $W: $(python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/cw$W -name '*results.db' | head -1) 2>/dev/null | grep 'cairo_composition_kernel<false>' | awk '{print $(NF-3), $(NF-2), $(NF-1)}')  $(tail -1 /tmp/cw$W.log | cut -c1-60)"
  cd $GRAFT_REPO_ROOT/lambdaworks_cairo_prover_amd/csrc
done

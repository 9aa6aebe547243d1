#!/bin/bash
# strided DIT passes: fused first pair (144+ VGPRs, three work-groups per CU) vs none (four work-groups with the 40 KB LDS tile)
cd $GRAFT_REPO_ROOT/lambdaworks_cairo_prover_amd/csrc
for F in 1 0 1 0; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -DSP_NTT_FUSE_LD_BUILD=$F -c ntt.hip -o ntt.o 2>/dev/null
  make -s 2>/dev/null
  cd $GRAFT_REPO_ROOT
  echo "fuse_ld build=$F"
  python tools/ntt_batch_bench.py 22 34 2>&1 | tail -1
  python tools/ntt_batch_bench.py 20 272 2>&1 | tail -1
  python tools/prove_bench.py 149000 8 80 20 2>&1 | tail -1
  cd $GRAFT_REPO_ROOT/lambdaworks_cairo_prover_amd/csrc
done

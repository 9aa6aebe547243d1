#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers rest on (run on the GPU box from the repo root):
#   tools/profile_round.sh <tag>   ->  gpurun_out/<tag>/{ntt_kt,ntt_pf,ntt_pw,prove_kt}  + text summaries
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --proof 0 --no-cpu-baseline --steps 20 --warmup 3"   # (bench.py warms itself for 300 ms before the timed steps)
# kernel trace of the DEFAULT bench command (200 timed steps: settled clocks, the durations bench.py itself reports)
rocprofv3 --kernel-trace --stats -d $O/ntt_kt -o ntt -- python3 $R/bench.py --proof 0 --no-cpu-baseline > $O/ntt_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/ntt_pf -o f -- $B > $O/ntt_pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/ntt_pw -o w -- $B > $O/ntt_pw.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prove_kt -o prove -- python3 $R/tools/prove_bench.py 149000 8 80 20 > $O/prove_kt.log 2>&1
cd $R
python3 tools/rocprof_summary.py $(find $O/ntt_kt -name "*results.db" | head -1) $(find $O/ntt_pf -name "*results.db" | head -1) $(find $O/ntt_pw -name "*results.db" | head -1) --bench-log $O/ntt_kt.log > $O/ntt_summary.txt 2>&1
python3 tools/rocprof_summary.py $(find $O/prove_kt -name "*results.db" | head -1) > $O/prove_summary.txt 2>&1
python3 tools/pmc_traffic.py $(find $O/ntt_pf -name "*results.db" | head -1) $(find $O/ntt_pw -name "*results.db" | head -1) 22 $O/ntt22_traffic.json $O/merkle_traffic.json > /dev/null 2>$O/traffic.err
head -12 $O/ntt_summary.txt; head -16 $O/prove_summary.txt; cat $O/ntt22_traffic.json $O/merkle_traffic.json

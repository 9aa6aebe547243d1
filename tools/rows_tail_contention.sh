#!/bin/bash
# the tail of the row-major host-table path (sp_cairo_prove at config #3) with the per-host thread budget in place -
# one context alone; the same with the budget of eight ranks (SP_HOST_RANKS=8: 4 gather threads instead of 24); and with seven more
# contexts alive on the box (each has proved once from rows - its gather pool exists and sleeps - and then idles).
# usage (GPU box, repo root): tools/rows_tail_contention.sh [iterations=60]  ->  gpurun_out/r05_rows_tail.txt
set -u
IT=${1:-60}
O=gpurun_out/r05_rows_tail.txt
mkdir -p gpurun_out
{
echo "## one context, default budget (host ranks 1)"
python3 tools/rows_tail.py 149000 8 $IT
echo "## one context, SP_HOST_RANKS=8 (the budget one of eight ranks gets)"
SP_HOST_RANKS=8 python3 tools/rows_tail.py 149000 8 $IT
echo "## eight contexts alive: seven idle ones (pool created by one small rows proof, then asleep) beside the measured one, SP_HOST_RANKS=8 everywhere"
pids=()
for i in 1 2 3 4 5 6 7; do
    SP_HOST_RANKS=8 SP_UPLOAD_MIN_MB=0 python3 - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
ctx = api.Context()
run = api.CairoRun.fibonacci(4000)
ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(4, 8, 3, 4))
time.sleep(150)
PY
    pids+=($!)
done
sleep 20
SP_HOST_RANKS=8 python3 tools/rows_tail.py 149000 8 $IT
for p in "${pids[@]}"; do kill $p 2>/dev/null; done
wait 2>/dev/null
echo "## threads of a rank: default budget vs SP_HOST_RANKS=8"
for r in 1 8; do
SP_HOST_RANKS=$r python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
from lambdaworks_cairo_prover_amd import api
ctx = api.Context()
before = len(os.listdir("/proc/self/task"))
run = api.CairoRun.fibonacci(70000)
ctx.cairo_prove(run.main_trace(), run.public_inputs_c, api.ProofOptions(4, 80, 3, 20))
print("host cpus", api.host_cpus(), "budget, ranks", api.host_cpu_budget(), "threads before / after the first rows proof", before, len(os.listdir("/proc/self/task")), "upload", ctx.last_upload_stats())
PY
done
} > $O 2>&1
grep -E "^#|median:|host cpus" $O

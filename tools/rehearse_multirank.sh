#!/bin/bash
# Rehearsal of the driver's N > 1 command lines on a box with ONE GPU (ranks share it, gloo-staged exchange): functional checks of the
# launcher, the control plane and the sharded proofs at their real sizes - NOT scaling measurements.
# usage: tools/rehearse_multirank.sh   (on the GPU box; writes gpurun_out/r05_bench_{4,8}ranks_shared_gpu.json)
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 1700 python3 bench.py --gpus 4 --steps 20 --warmup 5 > gpurun_out/r05_bench_4ranks_shared_gpu.json 2> gpurun_out/r05_bench_4ranks.err
echo "4 ranks (plain launcher): rc $? in $(( $(date +%s) - t0 )) s"
t0=$(date +%s)
timeout 1700 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 20 --warmup 5 \
    > gpurun_out/r05_bench_8ranks_shared_gpu.out 2> gpurun_out/r05_bench_8ranks.err
echo "8 ranks (torch.distributed.run, the driver's command): rc $? in $(( $(date +%s) - t0 )) s"
grep '^{' gpurun_out/r05_bench_8ranks_shared_gpu.out | tail -1 > gpurun_out/r05_bench_8ranks_shared_gpu.json
for f in gpurun_out/r05_bench_4ranks_shared_gpu.json gpurun_out/r05_bench_8ranks_shared_gpu.json; do
    python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d.get("summary", {})
    print(sys.argv[1], "value", d.get("value"), "n_gpus", d.get("n_gpus"), "error", d.get("error"),
          "cfg3", s.get("cfg3", {}).get("sha"), s.get("cfg3", {}).get("resident_ms"), "cfg4", s.get("cfg4", {}).get("sha"), s.get("cfg4", {}).get("resident_ms"), "rccl", s.get("rccl"))
except Exception as e:
    print(sys.argv[1], "NOT VALID JSON:", e)
PY
done
tail -n 5 gpurun_out/r05_bench_4ranks.err; tail -n 5 gpurun_out/r05_bench_8ranks.err

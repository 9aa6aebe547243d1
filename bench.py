#!/usr/bin/env python3
"""bench.py — Stark252 NTT throughput (BASELINE.json configs[1]: NTT size 2^22 on one MI355X) with roofline and
CPU-baseline objects.  One "step" = one forward natural-order NTT of 2^22 elements per GPU, inputs resident in HBM.

python bench.py --gpus N --steps K --warmup W   (N > 1: launched by torch.distributed.run, one rank per GPU; the path
shards by column with no data-path collective -> weak scaling, value = butterflies of all ranks / max time).

The "proof" object of the same JSON line is BASELINE.json's second figure: whole-proof generation of the 2^20-row Cairo
fibonacci trace (configs[2]) on the N GPUs - in process for N = 1, for N > 1 coset-sharded over the library's RCCL
communicator in one child process per rank under a time limit (proof_isolated), so that the headline line survives
whatever happens there.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 22
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_MUL_CEILING = 1.97e11
# registers-only chain of the butterfly the passes execute (fe_mul_lazy + fe_add_raw + fe_sub_add_2p), one MI355X, 25 ms
# kernels: profiles/r01_mulvar_ubench_long.txt (1.49e11 for the fully reduced butterfly of the first version)
VALU_BUTTERFLY_CEILING = 1.72e11


def cpu_baseline(log_n=22, reps=3):
    """CPU oracle (faithful restatement of the reference's radix-2 FFT) timed on this host, 1 thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "1"  # the oracle's codec loops are OpenMP-parallel; the reported baseline is 1 core
    import numpy as np
    import oracle_lib as oracle
    n = 1 << log_n
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    x[:, 0] &= 0x07
    oracle.ntt(x[:1024])  # load + warm
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle.ntt(x)
    dt = (time.perf_counter() - t0) / reps
    return {"value": (n // 2) * log_n / dt, "unit": "butterflies/s", "cores": 1, "kind": "port",
            "sample": f"{reps} x forward NTT 2^{log_n} through oracle_ntt (includes 32-byte BE codec), single thread"}


VALU_KECCAK_CEILING = 1.01e10  # Keccak-f[1600]/s, measured registers-only permutation rate (profiles/r01_keccak_ubench.txt)


def merkle_roofline(torch, ctx, dev, log_leaves=23, cols=34, reps=5):
    """The hash passes (BASELINE north_star: 'achieved HBM GB/s against the chip's peak for the NTT and hash passes'):
    one batched Keccak-256 Merkle commitment of the configs[2] main-trace shape - 2^23 leaves of 34 field elements read
    from the column-major LDE in HBM, all 2^24 - 1 nodes written - timed with HIP events on the context stream."""
    n = 1 << log_leaves
    data = torch.randint(0, 2**31 - 1, (cols, n, 8), dtype=torch.int32, device=dev)   # any residues: hashed after Montgomery -> canonical
    data[..., 7] &= 0x07FFFFFF
    nodes = torch.empty((2 * n - 1, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.merkle_build_dev(data.data_ptr(), n, cols, n, nodes.data_ptr())
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        ctx.merkle_build_dev(data.data_ptr(), n, cols, n, nodes.data_ptr())
    ms = ctx.timer_stop() / reps
    ctx.sync()
    algo_bytes = n * (32 * cols + 64)                         # leaf bytes read + 32 B per node written (+ re-read)
    perms = n * ((32 * cols + 1 + 135) // 136) + (n - 1)
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    del data, nodes
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "kernel": "leaf_hash_kernel + node_hash levels of one batched Merkle build", "avg_launch_ms": ms,
            "workload": f"2^{log_leaves} leaves x {cols} field elements (configs[2] main-trace commitment)",
            "keccak_f_per_s": perms / (ms * 1e-3), "valu_ceiling_keccak_f_per_s": VALU_KECCAK_CEILING,
            "valu_frac": perms / (ms * 1e-3) / VALU_KECCAK_CEILING}


def cpu_proof_baseline(api, ctx):
    """The CPU oracle's whole prover (OpenMP over columns / LDE points, the reference's rayon decomposition) on a bounded
    sample of the proof workload - the same Cairo fibonacci program at 2^14 trace rows, same options as configs[2] - next to
    the device prover on that very input; the two proofs must be the same bytes."""
    import ctypes
    import oracle_lib as oracle
    cores = os.cpu_count() or 1
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        cores = 1
    run = api.CairoRun.fibonacci(2330)
    trace = run.main_trace()
    opts = (8, 80, 3, 20)
    t0 = time.perf_counter()
    want = oracle.cairo_prove(trace, run.public_inputs_c, opts)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    gpu_ms = []
    for _ in range(3):
        t0 = time.perf_counter()
        got = ctx.cairo_prove(trace, run.public_inputs_c, api.ProofOptions(*opts))
        gpu_ms.append((time.perf_counter() - t0) * 1e3)
    return {"sample": f"whole proof, Cairo fibonacci trace {run.n_rows} rows x 52 columns, blowup 8, 80 queries, grinding 20",
            "cpu_ms": cpu_ms, "cores": cores, "kind": "port", "gpu_ms_same_input": min(gpu_ms), "identical_bytes": got == want}


def proof_benchmark(api, ctx, args, world, dist, force_rccl=False):
    """Whole-proof generation (BASELINE configs[2] shape by default: fib trace 2^20 rows, blowup 8, 80 queries, grinding 20)
    on the coset-sharded device prover; with N > 1 ranks the shards exchange through the library's RCCL communicator."""
    run = api.CairoRun.fibonacci(args.proof_fib)
    trace = run.main_trace()
    opt = api.ProofOptions(args.proof_blowup, 80, 3, 20)
    if world > 1 or force_rccl:
        if dist.get_backend() == "nccl":
            ctx.init_rccl()                                   # the library's own RCCL communicator (xGMI)
        else:                                                 # development aid: ranks sharing one GPU, host-staged exchange
            ctx.set_collective(world, dist.get_rank(), api.StagedAllGather())
    import torch
    dev_trace = torch.from_numpy(trace).to(torch.device(f"cuda:{torch.cuda.current_device()}"))  # input resident in HBM
    torch.cuda.synchronize()
    n, cols = trace.shape[0], trace.shape[1]
    proof = ctx.cairo_prove_dev(dev_trace.data_ptr(), n, cols, run.public_inputs_c, opt)  # warm-up: allocations, tables
    times = []
    for _ in range(3):
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        proof = ctx.cairo_prove_dev(dev_trace.data_ptr(), n, cols, run.public_inputs_c, opt)
        dt = (time.perf_counter() - t0) * 1e3
        if dist is not None and world > 1:   # max over ranks
            t = torch.tensor([dt], dtype=torch.float64, device=dev_trace.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        times.append(dt)
    rounds_dev = ctx.last_round_ms()
    t0 = time.perf_counter()
    proof_h = ctx.cairo_prove(trace, run.public_inputs_c, opt)
    pcie_ms = (time.perf_counter() - t0) * 1e3
    assert proof_h == proof
    import hashlib
    return {"proof_gen_ms": min(times), "proof_gen_ms_all": times, "device_round_ms": rounds_dev, "trace_rows": run.n_rows,
            "trace_cols": 52, "blowup": args.proof_blowup, "fri_queries": 80, "grinding": 20, "proof_bytes": len(proof),
            "proof_sha256": hashlib.sha256(proof).hexdigest(), "n_gpus": world,
            "proof_gen_ms_from_host_buffer": pcie_ms,
            "note": "wall time of sp_cairo_prove_dev (main trace resident in HBM); the *_from_host_buffer figure adds the PCIe copy"}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _device_index(local_rank):
    # SP_BENCH_FORCE_DEVICE: development aid (several ranks on the one GPU of the test box)
    return int(os.environ.get("SP_BENCH_FORCE_DEVICE", local_rank))


def _backend():
    return os.environ.get("SP_BENCH_BACKEND", "nccl")


def proof_child(args):
    """Runs in a child process of every rank (see proof_isolated): its own process group on its own port and its own RCCL
    communicator inside the library; rank 0 writes the result as JSON to args.proof_child."""
    import torch
    from lambdaworks_cairo_prover_amd import api
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist
    dev_index = _device_index(local_rank)
    torch.cuda.set_device(dev_index)
    if _backend() == "nccl":
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"))
    else:
        dist.init_process_group(_backend())
    result = None
    try:
        ctx = api.Context(device=dev_index)
        result = proof_benchmark(api, ctx, args, world, dist, force_rccl=True)
        ctx.close()
    except Exception as e:
        result = {"error": repr(e)}
    if rank == 0:
        with open(args.proof_child, "w") as f:
            json.dump(result, f)
    dist.destroy_process_group()


def proof_isolated(args, rank, local_rank, world, dist):
    """Whole-proof timing for N > 1 GPUs.  The sharded prover exchanges leaf digests through the library's RCCL communicator;
    every rank runs it in a CHILD process under a time limit, so that a failure or a hang of that path cannot take the
    headline measurement with it.  Returns the child's result (rank 0) or an error object."""
    import subprocess
    import tempfile
    box = [None, None]
    if rank == 0:
        fd, path = tempfile.mkstemp(prefix="sp_proof_", suffix=".json")
        os.close(fd)
        os.unlink(path)
        box = [_free_port(), path]
    if dist is not None:
        dist.broadcast_object_list(box, src=0)
    port, path = box
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC")}
    env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, os.path.abspath(__file__), "--proof-child", path, "--proof-fib", str(args.proof_fib),
           "--proof-blowup", str(args.proof_blowup)]
    err = tempfile.TemporaryFile()
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=err)
    status = "ok"
    try:
        rc = child.wait(timeout=args.proof_timeout)
        if rc != 0:
            status = f"child exit code {rc}"
    except subprocess.TimeoutExpired:
        child.kill()          # exactly the process started above
        child.wait()
        status = f"timeout after {args.proof_timeout} s"
    if rank != 0:
        return None
    try:
        with open(path) as f:
            res = json.load(f)
        os.unlink(path)
        if status != "ok" and isinstance(res, dict):
            res["child_status"] = status
        return res
    except Exception:
        err.seek(0)
        tail = err.read().decode(errors="replace")[-600:]
        return {"error": f"sharded proof child: {status}", "stderr_tail": tail}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--proof", type=int, default=-1, help="also time whole-proof generation (default: on; N > 1 in child processes)")
    ap.add_argument("--proof-isolated", action="store_true", help="run the proof timing in a child process also for 1 GPU")
    ap.add_argument("--proof-timeout", type=int, default=240, help="seconds the child processes of the N > 1 proof timing may take")
    ap.add_argument("--proof-child", type=str, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--proof-fib", type=int, default=149000, help="fibonacci index of the proved Cairo program (149000 -> 2^20 rows)")
    ap.add_argument("--proof-blowup", type=int, default=8)
    args = ap.parse_args()
    if args.proof_child:
        return proof_child(args)

    import numpy as np
    import torch
    from lambdaworks_cairo_prover_amd import api

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    dev_index = _device_index(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if _backend() == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group(_backend())
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")

    n = 1 << args.log_n
    # synthetic input: uniformly random residues < 2^251 (< p), written directly in the device layout
    # (8 x u32 little-endian Montgomery limbs) so that the timed region starts with the data resident in HBM.
    g = torch.Generator(device="cpu").manual_seed(0x5EED0000 + rank)
    host = torch.randint(0, 2**31 - 1, (n, 8), dtype=torch.int64, generator=g).to(torch.int32)
    host2 = torch.randint(0, 2, (n, 8), dtype=torch.int64, generator=g).to(torch.int32)
    host = host | (host2 << 31)
    host[:, 7] &= 0x07FFFFFF
    data = host.to(dev).contiguous()
    ctx = api.Context(device=dev_index)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.sync()

    for _ in range(args.warmup):
        ctx.ntt_dev(data.data_ptr(), n)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()                    # HIP events on the context stream bracket the timed region
    for _ in range(args.steps):
        ctx.ntt_dev(data.data_ptr(), n)  # asynchronous launches, back to back on the context stream
    kernel_ms = [ctx.timer_stop() / args.steps]
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev if _backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    traffic = None  # HBM bytes per NTT from the committed PMC profile (same command, same size)
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_ntt22_traffic.json")))
        if tj.get("log_n") == args.log_n:
            traffic = tj["traffic_bytes_per_ntt"]
    except Exception:
        pass
    butterflies = (n // 2) * args.log_n
    value = butterflies * args.steps * world / dt
    avg_ms = sum(kernel_ms) / len(kernel_ms)
    algo_bytes = 64.0 * n  # read once + write once
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
    out = {
        "metric": "stark252_ntt_field_ops_per_s", "value": value, "unit": "butterflies/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u256 mod p (8 x u32 Montgomery limbs)", "data": "synthetic",
        "config": {"workload": f"Stark252 forward NTT 2^{args.log_n}, natural order in/out, one vector per GPU (BASELINE configs[1])",
                   "log_n": args.log_n, "parallelism": f"replicas x{world} (column sharding, no collective)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": "ntt_pass_kernel chain of one NTT (all passes)", "avg_launch_ms": avg_ms,
                     "mulmod_per_s": butterflies / (avg_ms * 1e-3),
                     # the pass kernels are VALU-issue bound: the honest ceiling is the sustained rate
                     # of a registers-only butterfly (lazy mul + add + sub) chain, profiles/r01_mulvar_ubench_long.txt
                     "valu_ceiling_butterflies_per_s": VALU_BUTTERFLY_CEILING,
                     "valu_frac": butterflies / (avg_ms * 1e-3) / VALU_BUTTERFLY_CEILING},
    }
    if rank == 0:
        try:
            out["roofline_merkle"] = merkle_roofline(torch, ctx, dev)
        except Exception as e:
            out["roofline_merkle"] = {"error": repr(e)}
    if args.proof != 0:
        try:
            if world == 1 and not args.proof_isolated:
                out["proof"] = proof_benchmark(api, ctx, args, world, dist)
            else:
                out["proof"] = proof_isolated(args, rank, local_rank, world, dist)
        except Exception as e:  # the headline metric must survive a failure of the secondary one
            out["proof"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
        try:
            out["cpu_baseline"]["proof"] = cpu_proof_baseline(api, ctx)
        except Exception as e:
            out["cpu_baseline"]["proof"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — Stark252 NTT throughput (BASELINE.json configs[1]: NTT size 2^22 on one MI355X) with roofline and
CPU-baseline objects.  One "step" = one forward natural-order NTT of 2^22 elements per GPU, inputs resident in HBM.

python bench.py --gpus N --steps K --warmup W
  * started by torch.distributed.run (WORLD_SIZE in the environment): one rank per GPU, as launched;
  * started as a plain command with N > 1: this process spawns the N ranks itself, BEFORE anything touches the GPU, and
    passes rank 0's JSON line through.
The NTT workload shards by column with no data-path collective (weak scaling: value = butterflies of all ranks / max time).
Before the timed region every leg runs until at least WARM_MS of device time has passed, whatever --warmup says: a cold
MI355X needs ~100 ms to reach its clocks and a 20-step run would otherwise measure the ramp.

The "proof" objects of the same JSON line are BASELINE.json's second figure: whole-proof generation of the 2^20-row Cairo
fibonacci trace (configs[2]) and of the 70k-step program of benches/criterion_prover_70k.rs (configs[3], "proof_cfg4") on the
N GPUs - in process for N = 1; for N > 1 sharded over the library's RCCL communicator in one child process per rank under
a time limit, so that the headline line survives whatever happens there.  "rccl" reports what that communicator moved.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 22
WARM_MS = 300.0
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# registers-only chain of the butterfly the passes execute (fe_mul_lazy + fe_add_raw + fe_sub_add_2p), one MI355X, 47 ms kernels:
# profiles/r06_mul3_v2_ubench.txt - 1.82e11 since round 6 (a row's reduction through two multiply-adds; 1.74e11 with the carry-chain
# form of rounds 2 - 5, 1.49e11 for the fully reduced butterfly of the first version)
VALU_BUTTERFLY_CEILING = 1.82e11
# independent ceiling: the 80 v_mad_u64_u32 of one Montgomery product (64 of the product, 16 of the reduction; 72 before round 6) at the
# measured issue time of that instruction alone (2.299 ns per wave-instruction, 8 waves per SIMD, profiles/r01_instruction_ubench3.txt)
# on 1024 SIMDs x 64 lanes
MAD_ISSUE_NS = 2.299
MADS_PER_PRODUCT = 80
MUL_ISSUE_CEILING = 1024 * 64 / (MADS_PER_PRODUCT * MAD_ISSUE_NS * 1e-9)
VALU_KECCAK_CEILING = 1.01e10  # Keccak-f[1600]/s, measured registers-only permutation rate (profiles/r01_keccak_ubench.txt)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_ntt22_traffic.json")
MERKLE_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_merkle_traffic.json")


def ntt_source_sha16():
    h = hashlib.sha256()
    for f in ("ntt.hip", "ntt.h", "fp.h"):
        h.update(open(os.path.join(ROOT, "lambdaworks_cairo_prover_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def merkle_source_sha16():
    h = hashlib.sha256()
    for f in ("merkle.hip", "merkle.h", "keccak.h", "fp.h"):
        h.update(open(os.path.join(ROOT, "lambdaworks_cairo_prover_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def host_cpus():
    """CPUs this process may really use (sp_host_cpus: hardware threads cut down by the affinity mask and the cgroup CPU quota -
    the GPU boxes of the pool give a container 16 CPUs' worth of time on a 256-thread host)."""
    try:
        from lambdaworks_cairo_prover_amd import api
        return max(1, api.host_cpus())
    except Exception:
        return os.cpu_count() or 1


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def warm_until(ctx, fn, min_ms=WARM_MS, chunk=10, max_calls=100000):
    """Run fn() until the HIP-event time of the calls adds up to min_ms (clock ramp), whatever the caller's --warmup."""
    total, calls = 0.0, 0
    while total < min_ms and calls < max_calls:
        ctx.timer_start()
        for _ in range(chunk):
            fn()
        total += ctx.timer_stop()
        calls += chunk
    return calls


def cpu_baseline(log_n=22):
    """CPU oracle (faithful restatement of the reference's radix-2 FFT) timed on this host: one thread, and one vector per
    hardware thread (the reference's rayon decomposition is one column per worker); the 32-byte codec apart."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as oracle
    lib = oracle.load()
    n = 1 << log_n
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    x[:, 0] &= 0x07
    cores = host_cpus()
    out = (ctypes.c_double * 3)()
    xp = x.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    lib.oracle_ntt_bench(xp, ctypes.c_uint64(1 << 12), 1, 1, out)          # load + warm
    assert lib.oracle_ntt_bench(xp, ctypes.c_uint64(n), 2, 1, out) == 0
    one, codec_s = (n // 2) * log_n * 2 / out[1], out[0]
    vectors = min(cores, 64)            # 128 MiB of vectors and temporaries per thread: bounded
    assert lib.oracle_ntt_bench(xp, ctypes.c_uint64(n), vectors, vectors, out) == 0
    allc = (n // 2) * log_n * vectors / out[1]
    return {"value": one, "unit": "butterflies/s", "cores": 1, "kind": "port",
            "sample": f"2 x forward NTT 2^{log_n} in the oracle's field representation, single thread (codec excluded)",
            "codec_s_per_vector": codec_s, "cpu_model": cpu_model(), "nproc": os.cpu_count() or 1, "usable_cpus": cores,
            "all_cores": {"value": allc, "unit": "butterflies/s", "cores": vectors,
                          "sample": f"{vectors} x forward NTT 2^{log_n}, one vector per thread (OpenMP): {vectors} threads of the "
                                    f"{cores} CPUs this process may use (affinity mask and cgroup CPU quota) on a host with "
                                    f"{os.cpu_count()} hardware threads"}}


def merkle_roofline(torch, ctx, dev, log_leaves=23, cols=34, reps=5):
    """The hash passes (BASELINE north_star: 'achieved HBM GB/s against the chip's peak for the NTT and hash passes'):
    one batched Keccak-256 Merkle commitment of the configs[2] main-trace shape - 2^23 leaves of 34 field elements read
    from the column-major LDE in HBM, all 2^24 - 1 nodes written - timed with HIP events on the context stream."""
    n = 1 << log_leaves
    data = torch.randint(0, 2**31 - 1, (cols, n, 8), dtype=torch.int32, device=dev)   # any residues: hashed after Montgomery -> canonical
    data[..., 7] &= 0x07FFFFFF
    nodes = torch.empty((2 * n - 1, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    warm_until(ctx, lambda: ctx.merkle_build_dev(data.data_ptr(), n, cols, n, nodes.data_ptr()), chunk=4)
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
    traffic, note = None, "no PMC profile committed"
    try:   # HBM-side bytes per build from the committed PMC passes, valid only for the kernel sources they were taken from
        tj = json.load(open(MERKLE_TRAFFIC_FILE))
        if log_leaves == 23 and cols == 34:
            if tj.get("merkle_source_sha16") == merkle_source_sha16():
                traffic, note = tj["traffic_bytes_per_build"], os.path.relpath(MERKLE_TRAFFIC_FILE, ROOT)
            else:
                note = f"stale: {os.path.relpath(MERKLE_TRAFFIC_FILE, ROOT)} was taken from other kernel sources ({tj.get('traffic_bytes_per_build')} B)"
    except Exception:
        pass
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": note, "algorithmic_bytes": algo_bytes,
            "kernel": "leaf_hash_kernel + node_hash levels of one batched Merkle build", "avg_launch_ms": ms,
            "workload": f"2^{log_leaves} leaves x {cols} field elements (configs[2] main-trace commitment)",
            "keccak_f_per_s": perms / (ms * 1e-3), "valu_ceiling_keccak_f_per_s": VALU_KECCAK_CEILING,
            "valu_frac": perms / (ms * 1e-3) / VALU_KECCAK_CEILING}


def cpu_proof_sample(api, ctx):
    """The CPU oracle's whole prover (OpenMP over columns / LDE points, the reference's rayon decomposition) on a bounded
    sample of the proof workload - the same Cairo fibonacci program at 2^14 trace rows, same options as configs[2] - next to
    the device prover on that very input; the two proofs must be the same bytes."""
    import ctypes
    import oracle_lib as oracle
    cores = host_cpus()
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


def cpu_proof_child(args):
    """Child of cpu_proof_cfg4: the oracle's whole prover on configs[3]'s real shape, result as JSON on a file."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as oracle
    from lambdaworks_cairo_prover_amd import api
    cores = host_cpus()
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        cores = 1
    fib, blowup = args.cpu_proof_shape
    run = api.CairoRun.fibonacci(fib)
    trace = run.main_trace()
    t0 = time.perf_counter()
    proof, rounds = oracle.cairo_prove(trace, run.public_inputs_c, (blowup, 80, 3, 20), want_timings=True)
    ms = (time.perf_counter() - t0) * 1e3
    # kernel-level CPU rate beside the whole proof: Keccak-f permutations per second of the oracle's
    # Merkle build on a 2^16-leaf x 34-column tree (9 permutations a leaf - 1088 bytes + padding over a 136-byte rate - and one a node)
    import numpy as np
    leaves = np.random.default_rng(5).integers(0, 256, size=(1 << 16, 34, 32), dtype=np.uint8)
    leaves[:, :, 0] &= 0x07
    t1 = time.perf_counter()
    oracle.merkle_build(leaves)
    keccak_s = time.perf_counter() - t1
    perms = (1 << 16) * 9 + (1 << 16) - 1
    with open(args.cpu_proof_child, "w") as f:
        json.dump({"cpu_ms": ms, "cpu_round_ms": [round(x * 1e3, 1) for x in rounds], "cores": cores, "nproc": os.cpu_count() or 1,
                   "cpu_keccak_f_per_s": perms / keccak_s, "proof_sha256": hashlib.sha256(proof).hexdigest(), "proof_bytes": len(proof),
                   "trace_rows": run.n_rows}, f)


def cpu_proof_cfg4(args, device_proof):
    """configs[3] (2^19 rows, blowup 4, 80 queries, grinding 20 - benches/criterion_prover_70k.rs:46-57) on the CPU oracle at
    FULL size under a time budget (a child process, killed when the budget runs out), beside the device time."""
    fd, path = tempfile.mkstemp(prefix="sp_cpu_proof_", suffix=".json")
    os.close(fd)
    os.unlink(path)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-proof-child", path, "--cpu-proof-shape", str(args.cfg4_fib), str(args.cfg4_blowup)]
    child = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, SP_HOST_PINNED="0"))
    base = {"sample": f"whole proof, fib({args.cfg4_fib}) program, blowup {args.cfg4_blowup}, 80 queries, grinding 20 (configs[3], full size)",
            "kind": "port", "cpu_model": cpu_model(), "budget_s": args.cpu_proof_budget}
    try:
        rc = child.wait(timeout=args.cpu_proof_budget)
    except subprocess.TimeoutExpired:
        child.kill()
        child.wait()
        base["timeout"] = True
        return base
    try:
        res = json.load(open(path))
        os.unlink(path)
    except Exception:
        base["error"] = f"child exit code {rc}"
        return base
    base.update(res)
    # BASELINE.md section 3's third kernel-level CPU rate: LDE points through round 2 (constraint evaluation on every point, interpolation
    # of the composition polynomial, its commitment) per second - beside the device's round 2 on the same input, which evaluates 2n of them
    try:
        pts = res["trace_rows"] * args.cfg4_blowup
        base["cpu_round2_lde_points_per_s"] = pts / (res["cpu_round_ms"][1] * 1e-3)
        if isinstance(device_proof, dict) and device_proof.get("device_round_ms"):
            # like for like: the device evaluates the constraints on 2n points when the trace check was clean (composition
            # path 1 - deg H < 2n), on all N otherwise; the rate counts the points it EVALUATED.  What the CPU figure
            # is comparable with as a job - all N points' worth of round 2 per second - is the *_equivalent key.
            path = device_proof.get("composition_path", 1)
            evaluated = 2 * res["trace_rows"] if path == 1 else pts
            r2 = device_proof["device_round_ms"][2] * 1e-3
            base["gpu_round2_lde_points_per_s"] = evaluated / r2
            base["gpu_round2_lde_points_evaluated"] = evaluated
            base["gpu_round2_lde_points_equivalent_per_s"] = pts / r2
    except Exception:
        pass
    if isinstance(device_proof, dict) and "proof_sha256" in device_proof:
        base["gpu_ms_same_input"] = device_proof.get("proof_gen_ms_from_host_buffer")
        base["identical_bytes"] = device_proof["proof_sha256"] == res["proof_sha256"]
    return base


# The two FULL runs of the CPU oracle that exist on one and the same host (the build container, 8 cores, OMP_NUM_THREADS=8,
# tools/oracle_config3.py): configs[2] (2^20 rows, blowup 8) 460 s - tests/golden/README_config3.md - and configs[3] (the 70k program,
# 2^19 rows, blowup 4) 84 s.  Their ratio carries the first to whatever host timed the second.
CFG3_OVER_CFG4_SAME_HOST = 460.0 / 84.0


def extrapolate_cfg3_cpu(c4, p3, p4):
    """configs[2] itself takes the oracle minutes (460 s on the build container's 8 cores), beyond what a default run may spend.  Two
    estimates from the FULL configs[3] run this process just made on this host's CPUs, both labelled:
      * cpu_ms - that run x the measured same-host ratio of the two full runs (460 s / 84 s = 5.48: what the bigger shape really costs the
        oracle, memory effects included);
      * cpu_ms_by_round_laws - each round of that run scaled by its own law: rounds 1, 2 and 4 are dominated by the
        size-N transforms per column / per FRI layer (N log2 N, N = LDE points), round 3 evaluates the trace polynomials at the
        out-of-domain points (rows x columns); 52 columns and 80 queries in both shapes."""
    import math
    pts3, pts4 = p3["trace_rows"] * p3["blowup"], p4["trace_rows"] * p4["blowup"]
    nlogn = pts3 * math.log2(pts3) / (pts4 * math.log2(pts4))
    rows = p3["trace_rows"] / p4["trace_rows"]
    rounds = c4.get("cpu_round_ms") or []
    by_law = None
    if len(rounds) == 4:
        scaled = [rounds[0] * nlogn, rounds[1] * nlogn, rounds[2] * rows, rounds[3] * nlogn]
        by_law = sum(scaled) + max(0.0, c4["cpu_ms"] - sum(rounds)) * nlogn
    return {"kind": "extrapolated", "cpu_ms": c4["cpu_ms"] * CFG3_OVER_CFG4_SAME_HOST, "cores": c4.get("cores"), "scale": CFG3_OVER_CFG4_SAME_HOST,
            "cpu_ms_by_round_laws": by_law, "round_laws": {"rounds_1_2_4": nlogn, "round_3": rows},
            "full_runs_same_host": {"host": "build container, 8 cores", "cfg3_s": 460, "cfg4_s": 84},
            "from": "cpu_baseline.proof_cfg4 (a full run of the CPU oracle on this host's CPUs, identical bytes) x the ratio of the two full "
                    "oracle runs made on one host (460 s / 84 s, 8 cores); cpu_ms_by_round_laws: every round of that run by its own scaling law"}


def cold_child(args):
    """Child of cold_start: a fresh process proves once - context, setup, upload pipeline and all (the reference CLI proves once
    per process, src/main.rs:85-108) - then twice more; result as JSON on a file."""
    t_imp = time.perf_counter()
    import torch
    torch.cuda.init()
    from lambdaworks_cairo_prover_amd import api
    t_imp = (time.perf_counter() - t_imp) * 1e3
    fib, blowup = args.cold_shape
    opt = api.ProofOptions(blowup, 80, 3, 20)
    t_start = time.perf_counter()                # the one-shot wall clock: from here to the first proof's bytes
    ctx_on_thread = "+ctx" in args.cold_path     # 'run+ctx+prewarm': sp_ctx_create on the pre-warm's thread too, beside the VM
    t0 = time.perf_counter()
    ctx = None if ctx_on_thread else api.Context()
    t_ctx = (time.perf_counter() - t0) * 1e3
    t_setup = None
    rows = args.cold_path.startswith("rows")
    if args.cold_path.endswith("+prewarm"):
        # sp_prewarm on a thread of its own WHILE the front-end runs the program (ctypes releases the GIL): what a one-proof-per-process
        # caller does (src/main.rs:85-108: VM first, then the proof) - arena, tables, streams, first launches, clocks
        import threading
        n_rows = 1 << (7 * fib + 9).bit_length()   # 7 fib + 9 steps and a little padding: 2^20 for 149000, 2^19 for 70000
        box = {}

        box["cancel"] = False

        def warm():
            t = time.perf_counter()
            if ctx_on_thread:
                box["ctx"] = api.Context()
                box["ctx_ms"] = (time.perf_counter() - t) * 1e3
                if box["cancel"]:                         # (the VM was done before the context existed)
                    box["ctx"].prewarm_cancel()
            c = box["ctx"] if ctx_on_thread else ctx
            t = time.perf_counter()
            c.prewarm(n_rows, 34, 18, False, opt, api.SP_PREWARM_ALL if rows else (api.SP_PREWARM_KERNELS | api.SP_PREWARM_CLOCKS))
            box["ms"] = (time.perf_counter() - t) * 1e3

        th = threading.Thread(target=warm)
        t0 = time.perf_counter()
        th.start()
        run = api.CairoRun.fibonacci(fib)
        t_run = (time.perf_counter() - t0) * 1e3
        trace = run.main_trace() if rows else None        # (the caller's row-major table exists before the proof call, like the run)
        box["cancel"] = True
        c = box.get("ctx") if ctx_on_thread else ctx
        if c is not None:
            c.prewarm_cancel()                            # the trace exists: whatever is left of the clock ramp is cut short
        th.join()
        if ctx_on_thread:
            ctx, t_ctx = box["ctx"], box.get("ctx_ms")
        t_both = (time.perf_counter() - t0) * 1e3
        t_setup = box.get("ms")
    else:
        t_both = None
        t0 = time.perf_counter()
        run = api.CairoRun.fibonacci(fib)
        t_run = (time.perf_counter() - t0) * 1e3
        trace = run.main_trace() if rows else None
    ms = []
    t_one_shot = None
    for _ in range(4):
        t0 = time.perf_counter()
        proof = ctx.cairo_prove(trace, run.public_inputs_c, opt) if rows else ctx.cairo_prove_run(run, opt)
        ms.append((time.perf_counter() - t0) * 1e3)
        if t_one_shot is None:
            t_one_shot = (time.perf_counter() - t_start) * 1e3
    ms = [ms[0], ms[1], min(ms[2:])]
    with open(args.cold_child, "w") as f:
        json.dump({"first_call_ms": ms[0], "second_call_ms": ms[1], "third_call_ms": ms[2], "context_create_ms": t_ctx, "prewarm_ms": t_setup,
                   "front_end_and_prewarm_ms": t_both, "context_to_first_proof_ms": t_one_shot,
                   "trace_rows": run.n_rows, "front_end_run_ms": t_run, "front_end_split": run.timings(), "import_torch_and_library_ms": t_imp,
                   "proof_sha256": hashlib.sha256(proof).hexdigest()}, f)
    ctx.close()


def cold_start(args, fib, blowup, path):
    """First proof of a fresh process through `path` ("rows": sp_cairo_prove on a pageable row-major table, "run":
    sp_cairo_prove_run): first_call_ms includes sp_prove_setup's allocations and tables, the upload pipeline's threads and pinned
    ring and every first kernel launch; context_create_ms (the library's HIP runtime and code objects) is reported beside it."""
    fd, out = tempfile.mkstemp(prefix="sp_cold_", suffix=".json")
    os.close(fd)
    os.unlink(out)
    cmd = [sys.executable, os.path.abspath(__file__), "--cold-child", out, "--cold-shape", str(fib), str(blowup), "--cold-path", path]
    try:
        rc = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120).returncode
        res = json.load(open(out))
        os.unlink(out)
        return res
    except Exception as e:
        return {"error": repr(e)}


def proof_benchmark(api, ctx, fib, blowup, world, dist):
    """Whole-proof generation (fib trace, 80 queries, grinding 20) on the sharded device prover; N > 1 ranks exchange through
    whatever collective the caller installed on ctx."""
    import torch
    run = api.CairoRun.fibonacci(fib)
    trace = run.main_trace()
    opt = api.ProofOptions(blowup, 80, 3, 20)
    dev_trace = torch.from_numpy(trace).to(torch.device(f"cuda:{torch.cuda.current_device()}"))  # input resident in HBM
    torch.cuda.synchronize()
    n, cols = trace.shape[0], trace.shape[1]
    calls = [1]
    proof = ctx.cairo_prove_dev(dev_trace.data_ptr(), n, cols, run.public_inputs_c, opt)  # allocations, tables
    t_w = time.perf_counter()
    while (time.perf_counter() - t_w) * 1e3 < WARM_MS and world == 1:                     # clock ramp (ranks must stay in step: N = 1 only)
        ctx.cairo_prove_dev(dev_trace.data_ptr(), n, cols, run.public_inputs_c, opt)
        calls[0] += 1
    if world > 1:
        for _ in range(3):
            ctx.cairo_prove_dev(dev_trace.data_ptr(), n, cols, run.public_inputs_c, opt)
        calls[0] += 3
    calls[0] += 5
    times = []
    for _ in range(5):
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        proof = ctx.cairo_prove_dev(dev_trace.data_ptr(), n, cols, run.public_inputs_c, opt)
        dt = (time.perf_counter() - t0) * 1e3
        if dist is not None and world > 1:   # max over ranks
            t = torch.tensor([dt], dtype=torch.float64)       # (the control plane is a gloo group: CPU tensors)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        times.append(dt)
    rounds_dev = ctx.last_round_ms()
    info = ctx.last_proof_info()
    host_ms, run_ms, runcols_ms, up_rows, up_run = [], [], [], None, None
    ctx.cairo_prove(trace, run.public_inputs_c, opt)        # (first call: pinned staging ring, gather threads)
    ctx.cairo_prove_run(run, opt)
    calls[0] += 2
    ctx.set_option(api.SP_OPT_DEVICE_TRACE, 0)              # the run's host table, column group by column group (round 3's path)
    try:
        ctx.cairo_prove_run(run, opt)
        for _ in range(3):
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            ctx.cairo_prove_run(run, opt)
            runcols_ms.append((time.perf_counter() - t0) * 1e3)
        calls[0] += 4
    finally:
        ctx.set_option(api.SP_OPT_DEVICE_TRACE, 1)
    for _ in range(4):
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        proof_h = ctx.cairo_prove(trace, run.public_inputs_c, opt)
        host_ms.append((time.perf_counter() - t0) * 1e3)
        if host_ms[-1] == min(host_ms):
            up_rows = ctx.last_upload_stats()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        proof_r = ctx.cairo_prove_run(run, opt)
        run_ms.append((time.perf_counter() - t0) * 1e3)
        if run_ms[-1] == min(run_ms):
            up_run = ctx.last_upload_stats()
    assert proof_h == proof and proof_r == proof
    return {"proof_gen_ms": min(times), "proof_gen_ms_all": times, "device_round_ms": rounds_dev, "trace_rows": run.n_rows,
            "trace_cols": 52, "blowup": blowup, "fri_queries": 80, "grinding": 20, "proof_bytes": len(proof),
            "proof_sha256": hashlib.sha256(proof).hexdigest(), "n_gpus": world,
            "interpolation_sharded": info.get("interpolation_sharded"), "groups": info.get("groups"), "composition_path": info.get("composition_path"),
            "proof_gen_ms_from_host_buffer": min(host_ms), "proof_gen_ms_from_host_buffer_all": host_ms, "upload": up_rows,
            "proof_gen_ms_from_run": min(run_ms), "proof_gen_ms_from_run_all": run_ms, "upload_run": up_run,
            "proof_gen_ms_from_run_host_table_all": runcols_ms, "proofs_run": calls[0] + 8,
            "note": "proof_gen_ms: wall time of sp_cairo_prove_dev (main trace resident in HBM).  *_from_host_buffer: sp_cairo_prove, the "
                    "drop-in call on the reference's row-major TraceTable in pageable memory (host threads gather column groups into a "
                    "pinned ring, DMA, transforms of the groups overlapped).  *_from_run: sp_cairo_prove_run - the run's register states "
                    "and memory go up and the device builds the trace (SP_OPT_DEVICE_TRACE); *_from_run_host_table: the same call with the "
                    "option off (the run's page-locked column-major table, plain DMA per column group).  upload / upload_run: sp_last_upload_stats of the "
                    "fastest of those calls - exposed_ms is how long the compute stream waited for column groups"}


def poseidon_benchmark(api, torch, fib, blowup):
    """configs[4] names Poseidon Merkle trees: the optional backend (SP_OPT_MERKLE_BACKEND; NO reference counterpart, so no bit-exactness
    claim against the reference - the bytes equal the CPU oracle's under the same backend, tests/test_gpu_poseidon.py).  A context of
    its own (the option re-shapes the prover); the proof is checked by the library's host verifier under that backend."""
    run = api.CairoRun.fibonacci(fib)
    trace = run.main_trace()
    opt = api.ProofOptions(blowup, 80, 3, 20)
    dev_trace = torch.from_numpy(trace).to(torch.device(f"cuda:{torch.cuda.current_device()}"))
    torch.cuda.synchronize()
    n, cols = trace.shape[0], trace.shape[1]
    with api.Context(device=torch.cuda.current_device()) as ctx:
        ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_POSEIDON)
        proof = ctx.cairo_prove_dev(dev_trace.data_ptr(), n, cols, run.public_inputs_c, opt)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            proof = ctx.cairo_prove_dev(dev_trace.data_ptr(), n, cols, run.public_inputs_c, opt)
            times.append((time.perf_counter() - t0) * 1e3)
        rounds = ctx.last_round_ms()
        # the rate of the commitment kernel alone: 2^20 rows of 34 columns -> 18 Hades permutations per leaf + one per node
        leaves, width = 1 << 20, 34
        data = torch.randint(0, 2**31 - 1, (width, leaves, 8), dtype=torch.int32, device=dev_trace.device)
        data[..., 7] &= 0x07FFFFFF
        nodes = torch.empty((2 * leaves - 1, 32), dtype=torch.uint8, device=dev_trace.device)
        ctx.merkle_build_dev(data.data_ptr(), leaves, width, leaves, nodes.data_ptr())
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.merkle_build_dev(data.data_ptr(), leaves, width, leaves, nodes.data_ptr())
        ctx.sync()
        tree_ms = (time.perf_counter() - t0) * 1e3 / 3
    perms = leaves * ((width + 2) // 2) + leaves - 1
    t0 = time.perf_counter()
    ok = api.cairo_verify(proof, run.public_inputs_c, opt, api.SP_MERKLE_POSEIDON)
    verify_ms = (time.perf_counter() - t0) * 1e3
    return {"merkle_backend": "poseidon (Starknet Hades over Stark252, csrc/poseidon.h)", "proof_gen_ms": min(times), "proof_gen_ms_all": times,
            "device_round_ms": rounds, "trace_rows": run.n_rows, "trace_cols": 52, "blowup": blowup, "fri_queries": 80, "grinding": 20,
            "proof_bytes": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(), "verified_by_host_verifier": bool(ok),
            "host_verify_ms": verify_ms,
            "commitment_kernel": {"workload": "2^20 leaves x 34 columns, whole tree", "ms": tree_ms, "hades_permutations_per_s": perms / (tree_ms * 1e-3),
                                  "field_products_per_s": 214 * perms / (tree_ms * 1e-3),
                                  "note": "214 Montgomery products (83 + 24 squarings, 83 + 24 products) per permutation: the kernel runs at the "
                                          "multiplier-bound rate of the NTT butterflies (roofline.mulmod_per_s)"},
            "note": "not the reference's configuration (config.rs:10-20 fixes Keccak256 trees): reported beside `proof`, never instead of it"}


def _r(x, nd=2):
    return round(x, nd) if isinstance(x, (int, float)) else x


def _med_min(xs):
    import statistics
    xs = [x for x in (xs or []) if isinstance(x, (int, float))]
    return [_r(statistics.median(xs)), _r(min(xs))] if xs else None


def compact_line(full):
    """The ONE JSON line the driver records.  The driver keeps the contract keys, `roofline`, `cpu_baseline` and a 2000-character tail
    of stdout, so everything bulky (per-repetition arrays, the children of the first-call measurements, upload statistics, prose
    notes) goes to a side file and the line ENDS with a compact `summary` of the north-star figures: whole-proof times of configs[2]
    ("cfg3": 2^20 rows, blowup 8) and configs[3] ("cfg4": the 70k program, blowup 4) as [median, min] over the repetitions, first
    proof of a fresh process, the Merkle roofline of this very run, the CPU oracle beside them, sha256 prefixes of the proofs."""
    detail_dir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT
    detail_path = os.path.join(detail_dir, "bench_detail.json")
    try:
        with open(detail_path, "w") as f:
            json.dump(full, f, indent=1)
    except OSError:
        detail_path = None
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")
    line = {k: full[k] for k in keep if k in full}
    rf = full.get("roofline", {})
    line["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "mulmod_per_s", "mul_issue_frac", "valu_frac")}
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "usable_cpus")}
        if isinstance(cb.get("all_cores"), dict):
            line["cpu_baseline"]["all_cores_value"] = cb["all_cores"].get("value")
            line["cpu_baseline"]["all_cores_threads"] = cb["all_cores"].get("cores")
    rm = full.get("roofline_merkle")
    if isinstance(rm, dict) and "frac" in rm:
        line["roofline_merkle"] = {k: _r(rm.get(k), 4) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "valu_frac")}
    line["detail_file"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    summ = {}
    for key, name in (("proof", "cfg3"), ("proof_cfg4", "cfg4")):
        p = full.get(key)
        if not isinstance(p, dict):
            continue
        if "proof_gen_ms" not in p:
            summ[name] = {"error": str(p.get("error", p))[:160]}
            continue
        s = {"rows": p.get("trace_rows"), "blowup": p.get("blowup"), "n_gpus": p.get("n_gpus"),
             "resident_ms": _med_min(p.get("proof_gen_ms_all")), "from_rows_ms": _med_min(p.get("proof_gen_ms_from_host_buffer_all")),
             "from_run_ms": _med_min(p.get("proof_gen_ms_from_run_all")), "from_run_host_table_ms": _med_min(p.get("proof_gen_ms_from_run_host_table_all")),
             "first_call_ms": _r(p.get("first_call_ms")), "prewarmed_first_call_ms": _r(p.get("prewarmed_first_call_ms")),
             "warm_same_child_ms": _r(p.get("prewarmed_child_warm_ms")),
             "one_shot_ms": [_r(x, 0) for x in p["one_shot_ms"]] if isinstance(p.get("one_shot_ms"), list) else None,
             "comm_ms": p.get("comm_ms_per_proof"), "sha": (p.get("proof_sha256") or "")[:8]}
        summ[name] = {k: v for k, v in s.items() if v is not None}
    if isinstance(cb, dict):
        c4 = cb.get("proof_cfg4")
        if isinstance(c4, dict) and "cpu_ms" in c4 and "cfg4" in summ:
            summ["cfg4"].update(cpu_ms=_r(c4["cpu_ms"], 0), cpu_round_ms=c4.get("cpu_round_ms"), cpu_cores=c4.get("cores"), cpu_identical=c4.get("identical_bytes"))
        c3 = cb.get("proof_cfg3_extrapolated")
        if isinstance(c3, dict) and "cpu_ms" in c3 and "cfg3" in summ:
            summ["cfg3"].update(cpu_ms_extrapolated=_r(c3["cpu_ms"], 0), cpu_ms_by_round_laws=_r(c3.get("cpu_ms_by_round_laws"), 0), cpu_cores=c3.get("cores"),
                                cpu_full_run_elsewhere="460 s on 8 cores of the build container (84 s for cfg4 there)")
    if isinstance(rm, dict) and "frac" in rm:
        summ["merkle"] = {"frac": _r(rm["frac"], 4), "gbs": _r(rm.get("achieved"), 0), "ms": _r(rm.get("avg_launch_ms"), 3), "keccak_valu_frac": _r(rm.get("valu_frac"), 3)}
    pj = full.get("projected")
    if isinstance(pj, dict):
        for key, name in (("proof", "cfg3"), ("proof_cfg4", "cfg4")):
            q = pj.get(key)
            if isinstance(q, dict) and "best" in q:
                summ.setdefault("projected_not_measured", {})[f"{name}x{q.get('ranks')}"] = q["best"]
    c5 = full.get("cfg5_projected")
    if isinstance(c5, dict):
        summ["cfg5_projected"] = dict(c5.get("summary") or {"error": str(c5.get("error"))[:160]}, projection=True)
    pp = full.get("proof_poseidon")
    if isinstance(pp, dict) and "proof_gen_ms" in pp:
        summ["cfg3_poseidon_ms"] = _r(pp["proof_gen_ms"], 1)
    ap = full.get("air_prove")
    if isinstance(ap, dict) and "ms" in ap:
        summ["air_prove"] = {k: _r(ap.get(k)) for k in ("air", "rows", "blowup", "ms")}
    if isinstance(full.get("rccl"), dict):
        lk = full["rccl"].get("link") or {}
        pf = full["rccl"].get("preflight")
        if isinstance(pf, dict) and not pf.get("ok"):
            summ["rccl_preflight_failed"] = pf.get("failures") or True
        summ["rccl"] = dict({k: full["rccl"].get(k) for k in ("world", "backend", "devices_shared", "interpolation_sharded")}, selftest=full.get("transport_selftest"),
                            link_gbs=[_r(lk.get("allgather_gbs_per_link"), 1), _r(lk.get("alltoall_gbs_per_link"), 1)], link_ms=[_r(lk.get("allgather_ms"), 3), _r(lk.get("alltoall_ms"), 3)])
    line["summary"] = summ
    return line


def air_prove_benchmark(api, ctx, log_n=20, blowup=8):
    """sp_air_prove on an AIR other than Cairo - the reference's fibonacci_2_columns example (src/starks/example/fibonacci_2_columns.rs)
    at 2^20 rows: two columns, two constraints run by the INTERPRETED composition kernel (the constraint program of an sp_air_desc),
    from a row-major host table; the proof is checked by the library's host verifier."""
    import numpy as np
    from lambdaworks_cairo_prover_amd import air
    n, P = 1 << log_n, api.P
    a, b = 1, 1
    buf = bytearray(n * 64)
    for i in range(n):                       # c0[i+1] = c0[i] + c1[i], c1[i+1] = c1[i] + c0[i+1]
        buf[64 * i:64 * i + 32] = a.to_bytes(32, "big")
        buf[64 * i + 32:64 * i + 64] = b.to_bytes(32, "big")
        a = (a + b) % P
        b = (b + a) % P
    trace = np.frombuffer(bytes(buf), dtype=np.uint8).reshape(n, 2, 32)
    desc, keep = air.fibonacci_2_columns(1, 1).build()
    opt = api.ProofOptions(blowup, 80, 3, 20)
    proof = ctx.air_prove(desc, trace, opt)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        proof = ctx.air_prove(desc, trace, opt)
        times.append((time.perf_counter() - t0) * 1e3)
    info = ctx.last_proof_info()
    t0 = time.perf_counter()
    ok = api.air_verify(proof, desc, opt)
    return {"air": "fibonacci_2_columns", "rows": n, "cols": 2, "blowup": blowup, "fri_queries": 80, "grinding": 20, "ms": min(times), "ms_all": times,
            "composition_path": info["composition_path"], "proof_bytes": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(),
            "verified_by_host_verifier": bool(ok), "host_verify_ms": (time.perf_counter() - t0) * 1e3,
            "note": "sp_air_prove: host table in (67 MB), air_composition_kernel interprets the AIR's constraint program per point"}


XGMI_LINK_GBS_PER_DIRECTION = 76.8   # /opt/skills/guides/MI355X_MICROARCH.md: 7 links x ~153 GB/s bidirectional per GPU
XGMI_LINK_EFFICIENCY = 0.6           # what RCCL's point-to-point and ring kernels are assumed to reach of a link (not measured here)
COLLECTIVE_LATENCY_MS = 0.03


def project_ranks(api, fib, blowup, ranks, single_gpu_ms, single_rows_ms=None, single_run_ms=None):
    """PROJECTION, not a measurement: rank 0's share of a `ranks`-way sharded proof on this one GPU over the library's timing-only
    transport (sp_comm_init_null: nothing is exchanged, received blocks are zero-filled), which gives the per-rank compute time with
    every kernel at its real size, plus a model of the xGMI time of the collectives it issued (their byte counts are exact):
    received bytes / ((ranks - 1) links x 76.8 GB/s x 0.6) + 30 us per call.  Both interpolation modes are timed (by column with a
    coefficient all-gather / on every rank), and - in the mode the library picks by itself - the two host-input legs: the
    reference's row-major table (every rank uploads 1/ranks of it, the trace columns are all-gathered) and the run (registers +
    memory up, the trace built on every rank).  The proof bytes of such runs are meaningless."""
    import torch
    run = api.CairoRun.fibonacci(fib)
    trace = run.main_trace()
    opt = api.ProofOptions(blowup, 80, 3, 20)
    dev_trace = torch.from_numpy(trace).to(torch.device(f"cuda:{torch.cuda.current_device()}"))
    torch.cuda.synchronize()
    n, cols = trace.shape[0], trace.shape[1]

    def model(per, groups):
        ingest_gbs = max(1, groups - 1) * XGMI_LINK_GBS_PER_DIRECTION * XGMI_LINK_EFFICIENCY
        return per["received_bytes"] / (ingest_gbs * 1e9) * 1e3 + (per["allgather_calls"] + per["alltoall_calls"]) * COLLECTIVE_LATENCY_MS, ingest_gbs

    def leg(ctx, call, reps=3, warm=2):
        for _ in range(warm):
            call()
        before = ctx.comm_stats()
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            call()
            times.append((time.perf_counter() - t0) * 1e3)
        after = ctx.comm_stats()
        per = {k: (after[k] - before[k]) / reps for k in ("allgather_calls", "allgather_bytes", "alltoall_calls", "alltoall_bytes", "received_bytes")}
        return min(times), per

    out = {"projection": True, "ranks": ranks, "single_gpu_ms": single_gpu_ms, "modes": {}}
    for mode, name in ((0, "replicated_interpolation"), (1, "sharded_interpolation")):
        ctx = api.Context(device=torch.cuda.current_device())
        try:
            ctx.init_null(ranks, 0)
            ctx.set_option(api.SP_OPT_SHARD_INTERPOLATION, mode)
            compute_ms, per = leg(ctx, lambda: ctx.cairo_prove_dev(dev_trace.data_ptr(), n, cols, run.public_inputs_c, opt), warm=3)
            info, dev_bytes, rounds = ctx.last_proof_info(), ctx.prover_device_bytes(), ctx.last_round_ms()
        finally:
            ctx.close()
        groups = max(1, info["groups"])
        comm_ms, ingest = model(per, groups)
        out["modes"][name] = {"compute_ms": compute_ms, "comm_ms_model": comm_ms, "device_round_ms": rounds, "collectives_per_proof": per,
                              "device_bytes_per_rank": dev_bytes, "speedup_ceiling": single_gpu_ms / (compute_ms + comm_ms) if single_gpu_ms else None,
                              "speedup_if_comm_hidden": single_gpu_ms / compute_ms if single_gpu_ms else None}
        out["groups"], out["assumed_ingest_gbs"] = groups, ingest
    # the mode the library chooses by itself (SP_OPT_SHARD_INTERPOLATION = 2: the link model), and the host-input legs in it
    ctx = api.Context(device=torch.cuda.current_device())
    try:
        ctx.init_null(ranks, 0)
        ctx.cairo_prove_dev(dev_trace.data_ptr(), n, cols, run.public_inputs_c, opt)
        chosen = "sharded_interpolation" if ctx.last_proof_info()["interpolation_sharded"] else "replicated_interpolation"
        run_ms, run_per = leg(ctx, lambda: ctx.cairo_prove_run(run, opt))
        run_up = ctx.last_upload_stats()
    finally:
        ctx.close()
    # the row-major table: the LAST rank's share - its window of the table holds none of the flag columns that cross PCIe as bitmaps,
    # so it is the rank the others wait for in the all-gather of the trace
    ctx = api.Context(device=torch.cuda.current_device())
    try:
        ctx.init_null(ranks, ranks - 1)
        rows_ms, rows_per = leg(ctx, lambda: ctx.cairo_prove(trace, run.public_inputs_c, opt))
        rows_up = ctx.last_upload_stats()
    finally:
        ctx.close()
    out["default_mode"] = chosen
    best = out["modes"][chosen]
    out["compute_ms"], out["comm_ms_model"] = best["compute_ms"], best["comm_ms_model"]
    out["speedup_ceiling"], out["speedup_if_comm_hidden"] = best["speedup_ceiling"], best["speedup_if_comm_hidden"]
    rows_comm, _ = model(rows_per, out["groups"])
    run_comm, _ = model(run_per, out["groups"])
    out["from_host_rows"] = {"compute_and_upload_ms": rows_ms, "comm_ms_model": rows_comm, "uploaded_bytes_per_rank": rows_up["bytes"], "upload": rows_up,
                             "single_gpu_ms": single_rows_ms, "speedup_ceiling": single_rows_ms / (rows_ms + rows_comm) if single_rows_ms else None}
    out["from_run"] = {"compute_and_upload_ms": run_ms, "comm_ms_model": run_comm, "uploaded_bytes_per_rank": run_up["bytes"],
                       "single_gpu_ms": single_run_ms, "speedup_ceiling": single_run_ms / (run_ms + run_comm) if single_run_ms else None}
    out["best"] = {"mode": chosen, "compute_ms": _r(best["compute_ms"]), "comm_ms_model": _r(best["comm_ms_model"]), "x_no_overlap": _r(out["speedup_ceiling"]),
                   "x_comm_hidden": _r(out["speedup_if_comm_hidden"]),
                   "other_mode_x_no_overlap": _r(out["modes"]["sharded_interpolation" if chosen.startswith("repl") else "replicated_interpolation"]["speedup_ceiling"]),
                   "from_rows_x": _r(out["from_host_rows"]["speedup_ceiling"]), "from_run_x": _r(out["from_run"]["speedup_ceiling"])}
    out["note"] = ("NOT a measurement of a multi-GPU run: rank 0's compute share timed on one GPU with a null transport (bytes not exchanged, proof "
                   "bytes meaningless) + a bandwidth model of the exact collective byte counts; compute_ms + comm_ms_model assumes NO overlap "
                   "(the blocking collectives of this run), speedup_if_comm_hidden all of it")
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _placement(local_rank, world):
    """(device index, transport, devices shared): one GPU per rank and the library's own RCCL communicator ("rccl") when the box has
    the GPUs; otherwise (development box) the ranks share the devices there are and exchange through host-staged gloo hooks
    ("gloo-staged") - reported as devices_shared in the JSON line.  The CONTROL plane (rendezvous, barriers, the max over ranks, the
    128-byte RCCL id) is always a gloo group: it has monitored barriers with a timeout, needs no GPU, and keeps the headline line alive
    whatever the fabric does - every byte of the data path goes through sp_comm_init_rccl's communicator inside the library."""
    import torch
    count = torch.cuda.device_count()          # does not initialise the GPU
    if "SP_BENCH_FORCE_DEVICE" in os.environ:
        return int(os.environ["SP_BENCH_FORCE_DEVICE"]), os.environ.get("SP_BENCH_TRANSPORT", "gloo-staged"), True
    if count >= world:
        return local_rank, os.environ.get("SP_BENCH_TRANSPORT", "rccl"), False
    return local_rank % max(count, 1), "gloo-staged", True


def _init_control_plane(timeout_s):
    """torch.distributed over gloo on the loopback interface (one node by contract; the container's hostname may not resolve)."""
    import datetime
    import socket
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "GLOO_SOCKET_IFNAME" not in os.environ:
        try:
            socket.gethostbyname(socket.gethostname())
        except OSError:
            os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))
    return dist


def proof_child(args):
    """Runs in a child process of every rank (see proof_isolated): its own process group on its own port and its own RCCL
    communicator inside the library; rank 0 writes the result as JSON to args.proof_child."""
    import torch
    from lambdaworks_cairo_prover_amd import api
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    dev_index, transport, shared = _placement(local_rank, world)
    torch.cuda.set_device(dev_index)
    dist = _init_control_plane(INIT_TIMEOUT_S)
    result = {}
    try:
        ctx = api.Context(device=dev_index)
        if transport == "rccl":
            ctx.init_rccl()                                   # the library's own RCCL communicator (xGMI)
        else:                                                 # development aid: ranks sharing one GPU, host-staged exchange
            ctx.set_collective(world, rank, api.StagedAllGather())
        try:                                                  # rank-stamped blocks through every installed primitive, blocking and stream-ordered
            ctx.comm_selftest(1 << 20)
            result["transport_selftest"] = "ok"
        except Exception as e:
            result["transport_selftest"] = repr(e)
        for key, fib, blowup in (("proof", args.proof_fib, args.proof_blowup), ("proof_cfg4", args.cfg4_fib, args.cfg4_blowup)):
            try:
                before, t_before = ctx.comm_stats(), ctx.comm_time_ms()
                result[key] = proof_benchmark(api, ctx, fib, blowup, world, dist)
                after, t_after = ctx.comm_stats(), ctx.comm_time_ms()
                proofs_run = max(1, result[key].get("proofs_run", 1))
                # measured, not modelled: what the exchanges occupied their streams for (event pairs; the wait for the slowest peer included)
                # and the wall time inside blocking hooks, averaged over every proof of this leg (all input forms)
                result[key]["comm_ms_per_proof"] = {"stream_ordered": round((t_after[0] - t_before[0]) / proofs_run, 3), "blocking": round((t_after[1] - t_before[1]) / proofs_run, 3)}
                result[key]["collective_bytes_per_proof"] = {k: (after[k] - before[k]) // max(1, result[key].get("proofs_run", 1))
                                                             for k in ("allgather_bytes", "alltoall_bytes", "received_bytes")}
            except Exception as e:
                result[key] = {"error": repr(e)}
        stats = ctx.comm_stats()
        result["rccl"] = {"world": stats["world"], "backend": "rccl" if transport == "rccl" else "gloo-staged hook", "devices_shared": shared,
                          "allgather_calls": stats["allgather_calls"], "allgather_bytes": stats["allgather_bytes"],
                          "alltoall_calls": stats["alltoall_calls"], "alltoall_bytes": stats["alltoall_bytes"],
                          # sp_comm_measure (sp_comm_init_rccl runs it once, 64 MB per rank): GB/s per link and direction, minimum over the
                          # ranks - what SP_OPT_SHARD_INTERPOLATION = 2 decided from; zeros under the staged hooks (nothing measured)
                          "link": ctx.comm_measure(0),
                          "interpolation_sharded": {k: result[k].get("interpolation_sharded") for k in ("proof", "proof_cfg4") if isinstance(result.get(k), dict)}}
        ctx.close()
    except Exception as e:
        result["error"] = repr(e)
    if rank == 0:
        with open(args.proof_child, "w") as f:
            json.dump(result, f)
    dist.destroy_process_group()


def rccl_preflight_child(args):
    """Child of rccl_preflight: the library's RCCL communicator is built, checked (rank-stamped 1 MB blocks through every primitive) and
    timed in a process of its own; every rank writes {"ok": …} to its own file."""
    import torch
    from lambdaworks_cairo_prover_amd import api
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    dev_index, transport, shared = _placement(local_rank, world)
    torch.cuda.set_device(dev_index)
    dist = _init_control_plane(INIT_TIMEOUT_S)
    res = {"ok": False}
    try:
        ctx = api.Context(device=dev_index)
        ctx.init_rccl()
        ctx.comm_selftest(1 << 20)
        res = {"ok": True, "link": ctx.comm_measure(0)}
        ctx.close()
    except Exception as e:
        res["error"] = repr(e)[:300]
    with open(args.rccl_preflight, "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


def _die_with_parent(sig):
    """preexec_fn for children of a process that is single-threaded and has never touched the GPU (the launcher): the child gets `sig`
    when the process that started it dies (PR_SET_PDEATHSIG) - no rank outlives a launcher that was killed outright.  NOT for children of
    a rank: between fork and exec the closure runs Python (import, dlopen, allocator) in a copy of a multi-threaded process whose HIP or
    gloo threads may hold the loader or malloc lock at fork time - CPython documents that as deadlock-prone; those children
    arm the signal themselves (`_arm_parent_death_signal`)."""
    def fn():
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(sig), 0, 0, 0)      # PR_SET_PDEATHSIG = 1
        except Exception:
            pass
    return fn


def _arm_parent_death_signal():
    """First statement of a child of a RANK (proof child, RCCL pre-flight child): the parent is multi-threaded and owns a GPU, so nothing
    may run between its fork and the exec - the child itself asks for SP_BENCH_PDEATHSIG when the process named in SP_BENCH_PARENT_PID
    dies, and leaves at once if that process is already gone (the window between the exec and this call)."""
    want = os.environ.get("SP_BENCH_PARENT_PID")
    if not want:
        return
    try:
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(os.environ.get("SP_BENCH_PDEATHSIG", "9")), 0, 0, 0)      # PR_SET_PDEATHSIG = 1
    except Exception:
        pass
    if os.getppid() != int(want):
        os._exit(86)


def _spawn_rank_child(flag, path, rank, local_rank, world, port, extra_args=(), extra_env=None):
    import signal
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC")}
    env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               SP_BENCH_PARENT_PID=str(os.getpid()), SP_BENCH_PDEATHSIG=str(int(signal.SIGKILL)))
    env.update(extra_env or {})
    err = tempfile.TemporaryFile()
    # (no preexec_fn: this process has initialised torch, HIP and gloo - see _die_with_parent; close_fds + plain exec only)
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), flag, path, *extra_args], env=env, stdout=subprocess.DEVNULL, stderr=err)
    return child, err


def _wait_child(child, timeout):
    try:
        rc = child.wait(timeout=timeout)
        return "ok" if rc == 0 else f"child exit code {rc}"
    except subprocess.TimeoutExpired:
        child.kill()          # exactly the process started above
        child.wait()
        return f"timeout after {timeout:.0f} s"


def rccl_preflight(rank, local_rank, world, dist, timeout=150.0):
    """Before the proofs go over RCCL for the first time on this box: a throw-away child per rank builds the communicator and runs the
    selftest under a short time limit.  The ranks agree on the outcome (minimum over ranks on the control plane): "rccl", or - a failure or
    a hang anywhere - "gloo-staged", the host-staged hooks on the same devices, so that the N-GPU proof figures and their byte parity
    exist whatever the fabric does (reported as such)."""
    import torch
    box = [None]
    if rank == 0:
        box = [_free_port()]
    dist.broadcast_object_list(box, src=0)
    fd, path = tempfile.mkstemp(prefix=f"sp_preflight_{rank}_", suffix=".json")
    os.close(fd)
    os.unlink(path)
    child, err = _spawn_rank_child("--rccl-preflight", path, rank, local_rank, world, box[0])
    status = _wait_child(child, timeout)
    res = {"ok": False, "error": status}
    try:
        with open(path) as f:
            res = json.load(f)
        os.unlink(path)
    except Exception:
        err.seek(0)
        res["stderr_tail"] = err.read().decode(errors="replace")[-300:]
    ok = torch.tensor([1 if res.get("ok") else 0], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    reasons = [None] * world
    dist.all_gather_object(reasons, None if res.get("ok") else (res.get("error") or "failed"))
    return ("rccl" if int(ok.item()) == 1 else "gloo-staged"), {"ok": bool(ok.item()), "rank0": res, "failures": {str(r): x for r, x in enumerate(reasons) if x}}


def proof_isolated(args, rank, local_rank, world, dist):
    """Whole-proof timing for N > 1 GPUs.  The sharded prover exchanges digests and coefficients through the library's RCCL
    communicator; every rank runs it in a CHILD process under a time limit, so that a failure or a hang of that path cannot
    take the headline measurement with it.  Returns the child's result (rank 0) or an error object."""
    _, transport, _ = _placement(local_rank, world)
    preflight = None
    if transport == "rccl" and dist is not None and os.environ.get("SP_BENCH_NO_PREFLIGHT") is None:
        transport, preflight = rccl_preflight(rank, local_rank, world, dist)
    box = [None, None]
    if rank == 0:
        fd, path = tempfile.mkstemp(prefix="sp_proof_", suffix=".json")
        os.close(fd)
        os.unlink(path)
        box = [_free_port(), path]
    if dist is not None:
        dist.broadcast_object_list(box, src=0)
    port, path = box
    child, err = _spawn_rank_child("--proof-child", path, rank, local_rank, world, port,
                                   ["--proof-fib", str(args.proof_fib), "--proof-blowup", str(args.proof_blowup), "--cfg4-fib", str(args.cfg4_fib),
                                    "--cfg4-blowup", str(args.cfg4_blowup)], {"SP_BENCH_TRANSPORT": transport, "SP_COMM_LOG": "1"})
    status = _wait_child(child, args.proof_timeout)
    if rank != 0:
        return None
    try:
        with open(path) as f:
            res = json.load(f)
        os.unlink(path)
        if status != "ok" and isinstance(res, dict):
            res["child_status"] = status
        if isinstance(res, dict):      # what the library said it chose and from which rate (SP_COMM_LOG: one line per set-up shape)
            err.seek(0)
            said = [l for l in err.read().decode(errors="replace").splitlines() if l.startswith("[stark252 rank")]
            res.setdefault("rccl", {})["interpolation_decisions"] = said[:8]
        if preflight is not None and isinstance(res, dict):
            res.setdefault("rccl", {})["preflight"] = preflight
        return res
    except Exception:
        err.seek(0)
        tail = err.read().decode(errors="replace")[-600:]
        return {"error": f"sharded proof child: {status}", "stderr_tail": tail, "rccl": {"preflight": preflight}}


BENCH_DEADLINE_S = float(os.environ.get("SP_BENCH_DEADLINE_S", "1500"))          # the whole N > 1 job (the driver's own limit is 1800 s)
INIT_TIMEOUT_S = float(os.environ.get("SP_BENCH_INIT_TIMEOUT_S", "300"))          # rendezvous of the control plane (a fresh box pages torch in for 1-2 min)
BARRIER_TIMEOUT_S = float(os.environ.get("SP_BENCH_BARRIER_TIMEOUT_S", "300"))    # one barrier of the timed region


def error_line(args, world, message, **extra):
    """The ONE JSON line of a run that did not get to its measurement: the contract keys with value null and the reason."""
    line = {"metric": "stark252_ntt_field_ops_per_s", "value": None, "unit": "butterflies/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u256 mod p (8 x u32 Montgomery limbs)", "data": "synthetic",
            "config": {"workload": f"Stark252 forward NTT 2^{args.log_n}, natural order in/out, one vector per GPU (BASELINE configs[1])"},
            "error": message}
    line.update(extra)
    return line


_final_line_printed = False


def print_final(line):
    """Exactly one JSON line per job on stdout (whoever gets there first: the measurement, the watchdog or the SIGTERM handler)."""
    global _final_line_printed
    if _final_line_printed:
        return
    _final_line_printed = True
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()


class RankGuard:
    """What keeps an N > 1 rank from waiting for ever: a deadline on a timer thread (a collective or a kernel that never returns cannot
    be interrupted from Python) and a SIGTERM handler (torch.distributed.run terminates the other workers when one dies).  Either way
    rank 0 still prints the one JSON line - with the reason - and the process ends with a non-zero code."""

    def __init__(self, args, rank, world):
        import signal
        import threading
        self.args, self.rank, self.world, self.stage = args, rank, world, "start"
        self._timer = threading.Timer(BENCH_DEADLINE_S, self._expired)
        self._timer.daemon = True
        self._timer.start()
        try:
            signal.signal(signal.SIGTERM, self._terminated)
        except ValueError:      # (not the main thread)
            pass

    def _say(self, message):
        if self.rank == 0:
            print_final(error_line(self.args, self.world, message, stage=self.stage))

    def _expired(self):
        self._say(f"deadline of {BENCH_DEADLINE_S:.0f} s passed in stage '{self.stage}'")
        os._exit(4)

    def _terminated(self, signum, frame):
        self._say(f"terminated (SIGTERM) in stage '{self.stage}': another rank failed or the launcher gave up")
        os._exit(143)

    def fail(self, message):
        self._say(message)
        os._exit(5)

    def done(self):
        self._timer.cancel()


def launch_ranks(args):
    """`python bench.py --gpus N` as a plain command: start the N ranks as fresh child processes (this parent never touches the GPU),
    watch ALL of them, pass rank 0's JSON line through.  One rank that exits with an error - or an overall deadline - ends the job:
    the remaining children (each its own process group, so their proof children go with them) are terminated, ONE JSON line is printed
    whatever happened (rank 0's if it got that far, else an error line) and the return code is non-zero."""
    import signal
    port = _free_port()
    procs, out0 = [], tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True,
                                      preexec_fn=_die_with_parent(signal.SIGTERM), stdout=out0 if r == 0 else subprocess.DEVNULL))
    t_end = time.monotonic() + BENCH_DEADLINE_S + 30.0       # (the ranks' own deadline comes first and leaves an error line)
    reason = None
    asked = []                                               # a SIGTERM / SIGINT to the launcher ends the job like a failed rank does
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sig, lambda signum, frame: asked.append(signum))
        except ValueError:
            pass
    while True:
        if asked and reason is None:
            reason = f"launcher received signal {asked[0]}"
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if reason:
            pass
        elif bad:
            reason = f"rank {bad[0][0]} exited with code {bad[0][1]}"
        elif time.monotonic() > t_end:
            reason = f"launcher deadline of {BENCH_DEADLINE_S + 30:.0f} s passed"
        if reason:
            for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
                for p in procs:
                    if p.poll() is None:
                        try:
                            os.killpg(p.pid, sig)        # exactly the process groups started above
                        except ProcessLookupError:
                            pass
                t_grace = time.monotonic() + grace
                while time.monotonic() < t_grace and any(p.poll() is None for p in procs):
                    time.sleep(0.05)
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    rc = next((c for c in codes if c != 0), 0) or (1 if asked else 0)
    out0.seek(0)
    text = out0.read().decode(errors="replace")
    line = None
    for l in reversed(text.splitlines()):                  # (gloo prints its connection banner on stdout)
        if l.startswith("{"):
            try:
                line = json.loads(l)
                break
            except ValueError:
                continue
    if line is None:
        line = error_line(args, args.gpus, reason or f"rank 0 printed no JSON line (exit codes {codes})", exit_codes=codes)
        rc = rc or 1
    elif reason and "error" not in line:
        line["launcher_note"] = reason
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--proof", type=int, default=-1, help="also time whole-proof generation (default: on; N > 1 in child processes)")
    ap.add_argument("--proof-timeout", type=int, default=360, help="seconds the child processes of the N > 1 proof timing may take")
    ap.add_argument("--proof-child", type=str, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--rccl-preflight", type=str, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--proof-fib", type=int, default=149000, help="fibonacci index of the proved Cairo program (149000 -> 2^20 rows)")
    ap.add_argument("--proof-blowup", type=int, default=8)
    ap.add_argument("--cfg4-fib", type=int, default=70000, help="configs[3]: the 70k program of benches/criterion_prover_70k.rs (2^19 rows)")
    ap.add_argument("--cfg4-blowup", type=int, default=4)
    ap.add_argument("--cpu-proof-budget", type=int, default=150, help="seconds the CPU oracle may take for the full-size configs[3] proof (0: skip)")
    ap.add_argument("--cpu-proof-child", type=str, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-proof-shape", type=int, nargs=2, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cold-child", type=str, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cold-shape", type=int, nargs=2, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cold-path", type=str, default="rows", help=argparse.SUPPRESS)
    ap.add_argument("--no-poseidon", action="store_true", help="skip the proof with Poseidon Merkle trees (the optional backend of configs[4])")
    ap.add_argument("--no-cold-start", action="store_true", help="skip the first-proof-of-a-fresh-process measurements")
    ap.add_argument("--project-cfg5", action="store_true", help="N = 1 only: also run ONE rank's share of BASELINE configs[4] at its own size (2^24 rows, blowup 16, 8 ranks; "
                    "Keccak and Poseidon; ~169 GB of HBM, ~40 s) over the timing-only transport -> summary.cfg5_projected (a projection; tools/project_cfg5.py)")
    ap.add_argument("--project-ranks", type=int, default=8, help="N = 1 only: also PROJECT (not measure) an N-rank sharded proof from rank 0's share on this GPU (0: off)")
    args = ap.parse_args()
    if args.cpu_proof_child:
        return cpu_proof_child(args)
    if args.cold_child:
        return cold_child(args)
    if args.rccl_preflight:
        _arm_parent_death_signal()
        return rccl_preflight_child(args)
    if args.proof_child:
        _arm_parent_death_signal()
        return proof_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    guard = RankGuard(args, rank, world) if world > 1 else None
    if world > 1 and os.environ.get("SP_BENCH_FAULT_RANK") == str(rank):     # fault injection (tests/test_bench_launcher.py): this rank never joins
        if os.environ.get("SP_BENCH_FAULT", "exit") == "hang":
            time.sleep(1e6)
        sys.exit(3)

    def body():
        import torch
        from lambdaworks_cairo_prover_amd import api

        dist = None
        dev_index, transport, shared = _placement(local_rank, world)
        if world > 1:
            guard.stage = "rendezvous"
            try:
                dist = _init_control_plane(INIT_TIMEOUT_S)
            except Exception as e:
                guard.fail(f"rendezvous of the {world} ranks failed: {e!r}")
            os.environ.setdefault("SP_HOST_RANKS", os.environ.get("LOCAL_WORLD_SIZE", str(world)))   # the library's host-thread budget (sp_set_option SP_OPT_HOST_RANKS)
            guard.stage = "setup"
        torch.cuda.set_device(dev_index)
        dev = torch.device(f"cuda:{dev_index}")

        # one process per GPU, kept on the CPUs of its GPU's NUMA node (what `numactl --cpunodebind` does for a deployment on these
        # two-socket hosts): the tables this process builds are then first-touched beside the page-locked staging of the library
        numa_node = -1
        if os.environ.get("SP_BENCH_NO_NUMA_BIND") is None:
            try:
                numa_node = api.host_bind_to_device(dev_index)
            except Exception:
                numa_node = -1
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

        def barrier(stage="barrier"):
            """cuda synchronize + barrier; returns the moment THIS rank's device work was done (before it waited for the others)."""
            torch.cuda.synchronize()
            ctx.sync()
            t_synced = time.perf_counter()
            if dist is not None:
                import datetime
                guard.stage = stage
                try:        # bounded: names the rank that did not arrive instead of waiting for it for ever
                    dist.monitored_barrier(timeout=datetime.timedelta(seconds=BARRIER_TIMEOUT_S))
                except Exception as e:
                    guard.fail(f"{stage}: {e!r}"[:600])
            return t_synced

        self_warm = warm_until(ctx, lambda: ctx.ntt_dev(data.data_ptr(), n))   # clock ramp, independent of --warmup
        for _ in range(args.warmup):
            ctx.ntt_dev(data.data_ptr(), n)
        barrier("barrier before the timed region")
        t0 = time.perf_counter()
        ctx.timer_start()                    # HIP events on the context stream bracket the timed region
        for _ in range(args.steps):
            ctx.ntt_dev(data.data_ptr(), n)  # asynchronous launches, back to back on the context stream
        kernel_ms = [ctx.timer_stop() / args.steps]
        # each rank's K steps end when ITS device is idle; the closing barrier's own latency (a gloo round over N ranks, ~1 ms beside a
        # 7 ms region at --steps 20) is not NTT time.  The figure reported is the MAX of these over the ranks.
        dt = barrier("barrier behind the timed region") - t0
        if dist is not None:
            guard.stage = "max over ranks"
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            guard.stage = "proofs"

        # HBM bytes per NTT from the committed PMC passes of this very kernel source (tools/profile_round.sh); a profile taken
        # from other sources is reported as stale instead of being passed off as a measurement of this build
        traffic, traffic_note = None, "no PMC profile committed for this size"
        try:
            tj = json.load(open(TRAFFIC_FILE))
            if tj.get("log_n") == args.log_n:
                if tj.get("ntt_source_sha16") == ntt_source_sha16():
                    traffic, traffic_note = tj["traffic_bytes_per_ntt"], os.path.relpath(TRAFFIC_FILE, ROOT)
                else:
                    traffic_note = f"stale: {os.path.relpath(TRAFFIC_FILE, ROOT)} was taken from other kernel sources ({tj.get('traffic_bytes_per_ntt')} B)"
        except Exception:
            pass
        butterflies = (n // 2) * args.log_n
        value = butterflies * args.steps * world / dt
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        algo_bytes = 64.0 * n  # read once + write once
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        rate = butterflies / (avg_ms * 1e-3)
        out = {
            "metric": "stark252_ntt_field_ops_per_s", "value": value, "unit": "butterflies/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u256 mod p (8 x u32 Montgomery limbs)", "data": "synthetic",
            "config": {"workload": f"Stark252 forward NTT 2^{args.log_n}, natural order in/out, one vector per GPU (BASELINE configs[1])",
                       "log_n": args.log_n, "parallelism": f"replicas x{world} (column sharding, no collective)",
                       "self_warmup_steps": self_warm, "devices_shared": shared,
                       "host_numa_node": numa_node},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "ntt_pass_kernel chain of one NTT (all passes)", "avg_launch_ms": avg_ms,
                         "mulmod_per_s": rate,
                         # the pass kernels are VALU-issue bound.  Two ceilings: the multiplier alone (72
                         # v_mad_u64_u32 per product at that instruction's measured issue time - independent of this code), and
                         # the sustained rate of a registers-only chain of this code's butterfly (lazy mul + add + sub)
                         "mul_issue_ceiling_per_s": MUL_ISSUE_CEILING, "mul_issue_frac": rate / MUL_ISSUE_CEILING,
                         "valu_ceiling_butterflies_per_s": VALU_BUTTERFLY_CEILING, "valu_frac": rate / VALU_BUTTERFLY_CEILING},
        }
        if rank == 0:
            try:
                out["roofline_merkle"] = merkle_roofline(torch, ctx, dev)
            except Exception as e:
                out["roofline_merkle"] = {"error": repr(e)}
        if args.proof != 0:
            try:
                if world == 1:
                    out["proof"] = proof_benchmark(api, ctx, args.proof_fib, args.proof_blowup, 1, None)
                    out["proof_cfg4"] = proof_benchmark(api, ctx, args.cfg4_fib, args.cfg4_blowup, 1, None)
                    if not args.no_poseidon:
                        try:
                            out["proof_poseidon"] = poseidon_benchmark(api, torch, args.proof_fib, args.proof_blowup)
                        except Exception as e:
                            out["proof_poseidon"] = {"error": repr(e)}
                    try:
                        out["air_prove"] = air_prove_benchmark(api, ctx)
                    except Exception as e:
                        out["air_prove"] = {"error": repr(e)}
                    if args.project_ranks > 1:
                        out["projected"] = {}
                        for key, fib, blowup in (("proof", args.proof_fib, args.proof_blowup), ("proof_cfg4", args.cfg4_fib, args.cfg4_blowup)):
                            try:
                                # (more ranks than LDE cosets only replicate roles: project blowup-many ranks at most)
                                out["projected"][key] = project_ranks(api, fib, blowup, min(args.project_ranks, blowup), out[key].get("proof_gen_ms"),
                                                                      out[key].get("proof_gen_ms_from_host_buffer"), out[key].get("proof_gen_ms_from_run"))
                            except Exception as e:
                                out["projected"][key] = {"error": repr(e)}
                    if args.project_cfg5:
                        try:
                            sys.path.insert(0, os.path.join(ROOT, "tools"))
                            import project_cfg5
                            ctx.close()                                   # (the share needs the GPU's memory to itself: 169 of 288 GB)
                            out["cfg5_projected"] = project_cfg5.project(api)
                            ctx = api.Context(device=dev_index)
                        except Exception as e:
                            out["cfg5_projected"] = {"error": repr(e)}
                    if not args.no_cold_start:   # fresh child processes, one proof path each (this process keeps its own context)
                        for key, fib, blowup in (("proof", args.proof_fib, args.proof_blowup), ("proof_cfg4", args.cfg4_fib, args.cfg4_blowup)):
                            order = ["run", "rows", "run+prewarm", "rows+prewarm", "run+ctx+prewarm"]
                            if key == "proof":
                                # (a GPU that has idled for seconds - the projection's host work just did that - costs the next fresh process
                                # ~0.5 s once, whatever it runs: a throw-away child takes that instead of the first measured one)
                                out[key]["first_child_after_idle_ms"] = cold_start(args, fib, blowup, "run").get("context_to_first_proof_ms")
                            runs = {path: dict(cold_start(args, fib, blowup, path), path=path) for path in order}
                            out[key]["first_call_ms"] = runs["rows"].get("first_call_ms")
                            pw = runs["run+prewarm"]
                            out[key]["prewarmed_first_call_ms"] = pw.get("first_call_ms")
                            out[key]["prewarmed_child_warm_ms"] = pw.get("third_call_ms")
                            # sp_ctx_create -> VM -> first proof's bytes: everything in sequence / pre-warm beside the VM / context and pre-warm beside the VM
                            out[key]["one_shot_ms"] = [runs[p].get("context_to_first_proof_ms") for p in ("run", "run+prewarm", "run+ctx+prewarm")]
                            out[key]["first_call"] = dict(runs, note=(
                                "fresh child processes, one entry point each (rows: sp_cairo_prove on a pageable row-major table, run: "
                                "sp_cairo_prove_run); '+prewarm': sp_prewarm on a thread of its own while the front-end runs the program, then "
                                "the proof - prewarm_ms is that call alone, front_end_and_prewarm_ms the two together, third_call_ms the warm "
                                "proof of the same child.  first_call_ms = the rows child without a pre-warm"))
                else:
                    res = proof_isolated(args, rank, local_rank, world, dist)
                    if rank == 0:
                        for key in ("proof", "proof_cfg4", "rccl", "transport_selftest"):
                            if isinstance(res, dict) and key in res:
                                out[key] = res[key]
                        if not isinstance(res, dict) or "proof" not in res:
                            out["proof"] = res
            except Exception as e:  # the headline metric must survive a failure of the secondary one
                out["proof"] = {"error": repr(e)}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            try:
                out["cpu_baseline"]["proof"] = cpu_proof_sample(api, ctx)
            except Exception as e:
                out["cpu_baseline"]["proof"] = {"error": repr(e)}
            if args.cpu_proof_budget > 0 and args.proof != 0:
                try:
                    out["cpu_baseline"]["proof_cfg4"] = cpu_proof_cfg4(args, out.get("proof_cfg4"))
                except Exception as e:
                    out["cpu_baseline"]["proof_cfg4"] = {"error": repr(e)}
                c4, p3, p4 = out["cpu_baseline"]["proof_cfg4"], out.get("proof"), out.get("proof_cfg4")
                if isinstance(c4, dict) and "cpu_ms" in c4 and isinstance(p3, dict) and isinstance(p4, dict) and "trace_rows" in p3 and "trace_rows" in p4:
                    out["cpu_baseline"]["proof_cfg3_extrapolated"] = extrapolate_cfg3_cpu(c4, p3, p4)
                    if isinstance(out.get("cfg5_projected"), dict) and "summary" in out["cfg5_projected"]:
                        # configs[4]'s 447 GB of LDE columns do not fit the oracle's host ("n/a, exceeds host
                        # memory"): the configs[2] figure above carried on by LDE points x log2(points) (2^28 x 28 against 2^23 x 23) -
                        # an extrapolation of an extrapolation, labelled as such
                        scale5 = (2**28 * 28) / (2**23 * 23)
                        out["cfg5_projected"]["summary"]["cpu_ms_extrapolated"] = round(out["cpu_baseline"]["proof_cfg3_extrapolated"]["cpu_ms"] * scale5)
                        out["cfg5_projected"]["summary"]["cpu_note"] = (f"CPU oracle: n/a at this size (447 GB of LDE columns); configs[2]'s extrapolated "
                                                                          f"figure x {scale5:.1f} (N log N), {c4.get('cores')} CPUs")
        if rank == 0:
            print_final(compact_line(out))
        if guard is not None:
            guard.stage = "shutdown"
        ctx.close()
        if dist is not None:
            try:
                dist.monitored_barrier(timeout=__import__("datetime").timedelta(seconds=BARRIER_TIMEOUT_S))   # nobody leaves while a peer may still need the store
                dist.destroy_process_group()
            except Exception:
                pass
        if guard is not None:
            guard.done()
        return 0

    try:
        return body()
    except SystemExit:
        raise
    except BaseException as e:      # an N > 1 rank that fails anywhere still leaves the ONE line (rank 0) and a non-zero code
        if guard is None:
            raise
        import traceback
        traceback.print_exc()
        guard.fail(f"rank {rank} failed in stage '{guard.stage}': {e!r}"[:600])


if __name__ == "__main__":
    sys.exit(main() or 0)

#!/usr/bin/env python3
"""BASELINE.json configs[4] AT ITS OWN SIZE, one rank's share: trace 2^24 rows x 52 columns, blowup 16, 8 ranks (two LDE cosets each), Keccak and
Poseidon trees - rank 0 and rank 7 of the eight on ONE MI355X over the timing-only transport (sp_comm_init_null: nothing exchanged, received
blocks zero-filled, proof bytes meaningless), every kernel at its real size.  Measured: the prover's device bytes (replaces the "~190 GB by
linear extrapolation" of earlier rounds), per-round device ms, wall ms of the proof from the run (sp_cairo_prove_run: registers + memory up, the
trace built on the device - the path all eight ranks of a real job would take), the exact collective byte counts.  Modelled (stated, NOT
measured): the xGMI time of those bytes, (G - 1) links x 76.8 GB/s x 0.6 + 30 us per call, no overlap assumed.

usage: project_cfg5.py [--log-n 24] [--blowup 16] [--ranks 8] [--proofs 3] [--no-poseidon] [--out profiles/r05_cfg5_rank_share.txt]
The result is a PROJECTION of a multi-GPU run from one GPU, labelled as such wherever it is quoted."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

XGMI_LINK_GBS_PER_DIRECTION = 76.8
XGMI_LINK_EFFICIENCY = 0.6
COLLECTIVE_LATENCY_MS = 0.03


def fib_index_for_rows(log_n):
    """largest fibonacci index whose run still fits 2^log_n rows (the run of fib(k) has 7k + c rows: 149000 -> 2^20)"""
    return int(149000 * (1 << (log_n - 20)) * 1.0025) if log_n >= 20 else 149000 >> (20 - log_n)


def share(api, run, opt, ranks, rank, backend, proofs, shard_interp=2):
    ctx = api.Context(device=0)
    try:
        ctx.init_null(ranks, rank)
        ctx.set_option(api.SP_OPT_SHARD_INTERPOLATION, shard_interp)
        if backend == "poseidon":
            ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_POSEIDON)
        t0 = time.perf_counter()
        ctx.cairo_prove_run(run, opt)                      # allocations, tables, first launches
        first_ms = (time.perf_counter() - t0) * 1e3
        before = ctx.comm_stats()
        times = []
        for _ in range(proofs):
            t0 = time.perf_counter()
            ctx.cairo_prove_run(run, opt)
            times.append((time.perf_counter() - t0) * 1e3)
        after = ctx.comm_stats()
        per = {k: (after[k] - before[k]) / proofs for k in ("allgather_calls", "allgather_bytes", "alltoall_calls", "alltoall_bytes", "received_bytes")}
        info = ctx.last_proof_info()
        groups = max(1, info["groups"])
        ingest = max(1, groups - 1) * XGMI_LINK_GBS_PER_DIRECTION * XGMI_LINK_EFFICIENCY
        comm_ms = per["received_bytes"] / (ingest * 1e9) * 1e3 + (per["allgather_calls"] + per["alltoall_calls"]) * COLLECTIVE_LATENCY_MS
        return {"rank": rank, "ranks": ranks, "merkle": backend, "compute_ms": [round(statistics.median(times), 1), round(min(times), 1)],
                "first_proof_ms": round(first_ms, 1), "device_round_ms": [round(x, 1) for x in ctx.last_round_ms()],
                "device_gb": round(ctx.prover_device_bytes() / 1e9, 2), "collectives_per_proof": per, "comm_ms_model": round(comm_ms, 1),
                "assumed_ingest_gbs": round(ingest, 1), "interpolation_sharded": info["interpolation_sharded"], "groups": groups,
                "fri_sharded_layers": info.get("fri_sharded_layers"), "upload": ctx.last_upload_stats()}
    finally:
        ctx.close()


def project(api, log_n=24, blowup=16, ranks=8, proofs=3, poseidon=True, which_ranks=(0, 7), shard_interp=2):
    t0 = time.perf_counter()
    fib = fib_index_for_rows(log_n)
    run = api.CairoRun.fibonacci(fib)
    while run.n_rows > (1 << log_n):                        # (the estimate overshot: step back)
        fib = int(fib * 0.999)
        run = api.CairoRun.fibonacci(fib)
    assert run.n_rows == 1 << log_n, (fib, run.n_rows)
    front_end_s = time.perf_counter() - t0
    opt = api.ProofOptions(blowup, 80, 3, 20)
    out = {"projection": True, "config": f"BASELINE configs[4]: fib({fib}) -> 2^{log_n} rows x 52 columns, blowup {blowup}, 80 queries, grinding 20, {ranks} ranks",
           "front_end_s": round(front_end_s, 2), "front_end_split": run.timings(), "shares": []}
    for backend in (("keccak", "poseidon") if poseidon else ("keccak",)):
        for r in which_ranks:
            out["shares"].append(share(api, run, opt, ranks, min(r, ranks - 1), backend, proofs, shard_interp))
    k = [s for s in out["shares"] if s["merkle"] == "keccak"]
    slow = max(k, key=lambda s: s["compute_ms"][0])
    out["summary"] = {"compute_ms": slow["compute_ms"][0], "comm_ms_model": slow["comm_ms_model"], "device_gb": max(s["device_gb"] for s in out["shares"]),
                      "proof_ms_no_overlap": round(slow["compute_ms"][0] + slow["comm_ms_model"], 1), "slowest_rank": slow["rank"]}
    p = [s for s in out["shares"] if s["merkle"] == "poseidon"]
    if p:
        ps = max(p, key=lambda s: s["compute_ms"][0])
        out["summary"]["poseidon_compute_ms"] = ps["compute_ms"][0]
        out["summary"]["poseidon_proof_ms_no_overlap"] = round(ps["compute_ms"][0] + ps["comm_ms_model"], 1)
    out["note"] = ("NOT a measurement of an 8-GPU run: one rank's compute share timed on one GPU over a null transport (proof bytes meaningless) + a "
                   "bandwidth model of the exact collective byte counts, no overlap assumed; device_gb IS a measurement (sp_prover_device_bytes)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--blowup", type=int, default=16)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--proofs", type=int, default=3)
    ap.add_argument("--no-poseidon", action="store_true")
    ap.add_argument("--shard-interp", type=int, default=2, help="SP_OPT_SHARD_INTERPOLATION: 0 every rank, 1 by column + coefficient all-gather, 2 the link model")
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    from lambdaworks_cairo_prover_amd import api
    res = project(api, args.log_n, args.blowup, args.ranks, args.proofs, not args.no_poseidon, shard_interp=args.shard_interp)
    text = json.dumps(res, indent=1)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()

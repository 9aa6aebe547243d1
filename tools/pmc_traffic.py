#!/usr/bin/env python3
"""HBM traffic per NTT from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) -> profiles/<name>.json, the file bench.py reads for roofline.traffic.
usage: pmc_traffic.py <fetch_results.db> <write_results.db> <log_n> <out.json> [<merkle_out.json>]"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ntt_source_sha16():
    """Same digest as bench.py: the traffic figure is only valid for the kernel sources it was measured on."""
    h = hashlib.sha256()
    for f in ("ntt.hip", "ntt.h", "fp.h"):
        h.update(open(os.path.join(ROOT, "lambdaworks_cairo_prover_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def merkle_source_sha16():
    h = hashlib.sha256()
    for f in ("merkle.hip", "merkle.h", "keccak.h", "fp.h"):
        h.update(open(os.path.join(ROOT, "lambdaworks_cairo_prover_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    q = "select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name"
    return {name: (cnt, tot) for name, cnt, tot in db.execute(q, (counter,))}


def main():
    fpath, wpath, log_n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, w = per_kernel(fpath, "FETCH_SIZE"), per_kernel(wpath, "WRITE_SIZE")
    fp = {k: v for k, v in f.items() if "ntt_pass_kernel" in k}
    wp = {k: v for k, v in w.items() if "ntt_pass_kernel" in k}
    assert fp and set(fp) == set(wp), "the two PMC passes must have run the same pass kernels"

    def launches_per_ntt(passes):
        # every NTT launches each of its pass kernels a fixed number of times: transforms of the run = the smallest dispatch count
        # (bench.py's warm_until() runs a different number of transforms under each profiler pass, so each pass is normalised by
        # ITS OWN count - dividing the WRITE_SIZE total by the FETCH pass's count was the 1.28 GB bug of round 2)
        ntts = min(c for c, _ in passes.values())
        mult = {}
        for k, (c, _) in passes.items():
            assert c % ntts == 0, f"{k}: {c} dispatches is not a multiple of {ntts} transforms"
            mult[k] = c // ntts
        return ntts, mult

    ntts_f, mult_f = launches_per_ntt(fp)
    ntts_w, mult_w = launches_per_ntt(wp)
    assert mult_f == mult_w, (mult_f, mult_w)
    launches = sum(mult_f.values())
    fetch_kb = sum(mult_f[k] * t / c for k, (c, t) in fp.items())     # per-dispatch average x dispatches per transform
    write_kb = sum(mult_w[k] * t / c for k, (c, t) in wp.items())
    passes = fp
    res = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --proof 0 --steps 20 --warmup 3`",
        "log_n": log_n,
        "launches_per_ntt": launches,
        "transforms_in_fetch_pass": ntts_f,
        "transforms_in_write_pass": ntts_w,
        "fetch_size_kb_per_ntt": fetch_kb,
        "write_size_kb_per_ntt": write_kb,
        "ntt_source_sha16": ntt_source_sha16(),
        "correction": "FETCH_SIZE x 2, WRITE_SIZE x 1: calibrated on known byte counts in the access patterns of the passes "
                      "(tools/fetch_calibration.hip, profiles/r02_fetch_calibration.txt: contiguous and 128/256-byte-row reads of 1 GiB "
                      "report 0.5 GiB, writes report 1:1).  The figure includes the twiddle fetches (32-byte gathers, mostly served by "
                      "L2 / the 256 MiB infinity cache, which FETCH_SIZE counts).",
        "traffic_bytes_per_ntt": (2 * fetch_kb + write_kb) * 1024,
        "traffic_over_algorithmic": (2 * fetch_kb + write_kb) * 1024 / (64.0 * (1 << log_n)),
        "per_kernel_fetch_kb_avg": {k: t / c for k, (c, t) in fp.items()},
        "per_kernel_write_kb_avg": {k: t / c for k, (c, t) in wp.items()},
        "per_kernel_launches_per_ntt": mult_f,
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
    # the Merkle leg of the same bench.py run (roofline_merkle: one batched build of 2^23 leaves x 34 elements): every build
    # launches the leaf kernel once, so builds = launches of the leaf kernel; all hash kernels of the run belong to that leg
    if len(sys.argv) > 5:
        leaf = [k for k in f if "leaf_hash" in k]
        if leaf:
            builds_f = sum(f[k][0] for k in leaf)
            builds_w = sum(w[k][0] for k in leaf if k in w) or builds_f
            fetch_kb = sum(t for k, (_, t) in f.items() if "hash" in k) / builds_f
            write_kb = sum(t for k, (_, t) in w.items() if "hash" in k) / builds_w
            m = {"source": res["source"] + " (its Merkle leg)", "workload": "2^23 leaves x 34 field elements",
                 "fetch_size_kb_per_build": fetch_kb, "write_size_kb_per_build": write_kb, "merkle_source_sha16": merkle_source_sha16(),
                 "correction": "FETCH_SIZE x 2, WRITE_SIZE x 1 (profiles/r02_fetch_calibration.txt)",
                 "traffic_bytes_per_build": (2 * fetch_kb + write_kb) * 1024}
            json.dump(m, open(sys.argv[5], "w"), indent=1)
            print(json.dumps(m, indent=1))


if __name__ == "__main__":
    main()

"""bench.py's pure pieces (no GPU): the compact JSON line keeps the contract keys and ends with `summary`; the configs[2] CPU figure is the
same-host ratio of the two full oracle runs; the error line carries the contract keys."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def test_compact_line_of_a_recorded_run(tmp_path, monkeypatch):
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_cmd_detail.json")))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))          # (the detail file goes beside the line: not into the repository)
    full["cfg5_projected"] = {"summary": {"compute_ms": 433.7, "comm_ms_model": 15.7, "device_gb": 168.86, "proof_ms_no_overlap": 449.4}}
    full["rccl"] = {"world": 8, "backend": "rccl", "devices_shared": False, "link": {"allgather_gbs_per_link": 55.5, "alltoall_gbs_per_link": 44.4, "allgather_ms": 1.2, "alltoall_ms": 0.9},
                    "interpolation_sharded": {"proof": 0, "proof_cfg4": 0}, "preflight": {"ok": True}}
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert all(k in line for k in CONTRACT) and list(line)[-1] == "summary"
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"]) and line["roofline"]["bound"] == "hbm"
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"]) and line["cpu_baseline"]["kind"] == "port"
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-12
    summ = line["summary"]
    assert summ["cfg3"]["sha"] == "3b115b1a" and summ["cfg4"]["sha"] == "25b2708a" and summ["cfg4"]["cpu_identical"] is True
    assert summ["cfg5_projected"]["projection"] is True and summ["rccl"]["link_gbs"] == [55.5, 44.4] and "rccl_preflight_failed" not in summ
    assert len(text) < 4096 and len(json.dumps(summ)) < 2000          # the driver keeps a 2000-character tail: the summary must fit it


def test_cfg3_cpu_figure_scales_the_full_cfg4_run():
    c4 = {"cpu_ms": 18000.0, "cpu_round_ms": [8700.0, 5900.0, 100.0, 2900.0], "cores": 16}
    out = bench.extrapolate_cfg3_cpu(c4, {"trace_rows": 1 << 20, "blowup": 8}, {"trace_rows": 1 << 19, "blowup": 4})
    assert abs(out["cpu_ms"] - 18000.0 * 460 / 84) < 1e-6 and out["kind"] == "extrapolated" and out["cores"] == 16
    nlogn = (2**23 * 23) / (2**21 * 21)
    want = (8700 + 5900 + 2900) * nlogn + 100 * 2 + (18000 - 17600) * nlogn
    assert abs(out["cpu_ms_by_round_laws"] - want) < 1e-6


def test_error_line_has_the_contract_keys():
    args = argparse.Namespace(steps=20, warmup=5, log_n=22)
    line = bench.error_line(args, 8, "rank 3 exited with code 1", stage="rendezvous")
    assert all(k in line for k in CONTRACT) and line["value"] is None and line["n_gpus"] == 8 and line["error"] and line["stage"] == "rendezvous"

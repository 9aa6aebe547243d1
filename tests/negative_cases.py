"""The reference's negative integration tests (tests/integration_tests.rs:206-357) as data: each case yields
(main trace, public inputs to prove with, options to prove with, public inputs to verify with, options to verify with);
the proof must be REJECTED.  Used with the oracle prover on the CPU and with the device prover on the GPU."""
import ctypes

import cairo_asm as A
import oracle_lib
from lambdaworks_cairo_prover_amd import api

DEFAULT = (4, 3, 3, 1)      # ProofOptions::default_test_options


def _pub_fields(run):
    pi = run.public_inputs_c
    g = lambda name: int.from_bytes(bytes(getattr(pi, name)), "big")   # noqa: E731
    types = list(ctypes.string_at(pi.segment_types, pi.n_segments)) if pi.n_segments else []
    ranges = list((ctypes.c_uint64 * (2 * pi.n_segments)).from_address(pi.segment_ranges)) if pi.n_segments else []
    return dict(pc_init=g("pc_init"), ap_init=g("ap_init"), fp_init=g("fp_init"), pc_final=g("pc_final"), ap_final=g("ap_final"),
                rc_min=pi.range_check_min, rc_max=pi.range_check_max, public_memory=run.public_memory(), num_steps=pi.num_steps,
                segments=[(types[i], ranges[2 * i], ranges[2 * i + 1]) for i in range(len(types))])


def _make(f):
    return oracle_lib.make_public_inputs(f["pc_init"], f["ap_init"], f["fp_init"], f["pc_final"], f["ap_final"], f["rc_min"], f["rc_max"],
                                         f["public_memory"], f["num_steps"], f["segments"])


def cases():
    out = {}
    # :207 a slightly different program in the verifier's public memory
    run = api.CairoRun.fibonacci(20)
    f = _pub_fields(run)
    good, k1 = _make(f)
    pm = dict(f["public_memory"]); pm[1] = 5; pm[3] = 5
    bad, k2 = _make(dict(f, public_memory=sorted(pm.items())))
    out["slightly_different_program"] = (run.main_trace(), good, DEFAULT, bad, DEFAULT, (run, k1, k2))
    # :227 different range bounds
    bad1, k3 = _make(dict(f, rc_min=f["rc_min"] + 1))
    bad2, k4 = _make(dict(f, rc_max=f["rc_max"] - 1))
    out["range_check_min_plus_one"] = (run.main_trace(), good, DEFAULT, bad1, DEFAULT, (run, k1, k3))
    out["range_check_max_minus_one"] = (run.main_trace(), good, DEFAULT, bad2, DEFAULT, (run, k1, k4))
    # :244 changed range-check value in the trace (the decomposition constraint fails)
    words, entry = A.rc_program()
    rc = api.CairoRun.from_program_builtins(words, entry_pc=entry)
    frc = _pub_fields(rc)
    prc, k5 = _make(frc)
    t = rc.main_trace().copy()
    t[0, t.shape[1] - 1] = api.felts_to_bytes([35])[0]
    out["changed_range_check_value"] = (t, prc, DEFAULT, prc, DEFAULT, (rc, k5))
    # :269 a value above 2^128 in the range-check column (its 8 x 16-bit limbs only cover the low 128 bits)
    t2 = rc.main_trace().copy()
    t2[0, t2.shape[1] - 1] = api.felts_to_bytes([2**128 + 1])[0]
    out["overflowing_range_check_value"] = (t2, prc, DEFAULT, prc, DEFAULT, (rc, k5))
    # :305 changed output: an output cell's value in the trace differs from the public memory
    wo, eo = A.output_rc_program()
    ro = api.CairoRun.from_program_builtins(wo, entry_pc=eo, output=True)
    fo = _pub_fields(ro)
    po, k6 = _make(fo)
    t3 = ro.main_trace().copy()
    out_seg = [s for s in fo["segments"] if s[0] == 1][0]
    hit = None
    for i in range(t3.shape[0]):                     # the row whose op1 address is the first output cell
        if int.from_bytes(t3[i, 22].tobytes(), "big") == out_seg[1]:
            hit = i
            break
    assert hit is not None
    t3[hit, 26] = api.felts_to_bytes([100])[0]       # op1 value
    out["changed_output"] = (t3, po, DEFAULT, po, DEFAULT, (ro, k6))
    # :341 different security parameters on the verifier's side (Conjecturable80Bits vs 128Bits, blowup 8: options.rs:37-72)
    out["different_security_params"] = (ro.main_trace(), po, (8, 31, 3, 20), po, (8, 55, 3, 20), (ro, k6))
    return out

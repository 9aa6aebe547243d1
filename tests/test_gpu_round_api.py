"""The round-level C ABI (what a Rust shim inside src/starks/prover.rs would call) driven from Python with
the transcript on the caller's side: every root / value it returns equals the corresponding field of the oracle's proof.
Also: the SP_FE_MONT_LIMBS context encoding (lambdaworks' in-memory FieldElement layout) gives the same proof bytes."""
import ctypes
import struct

import numpy as np
import pytest

from lambdaworks_cairo_prover_amd import _lib, api

pytestmark = pytest.mark.gpu
P = api.P


class Boundary(ctypes.Structure):
    _fields_ = [("col", ctypes.c_uint32), ("step", ctypes.c_uint64), ("value", ctypes.c_uint8 * 32)]


class Openings(ctypes.Structure):
    _fields_ = [("n_queries", ctypes.c_uint32), ("n_layers", ctypes.c_uint32), ("n_cols", ctypes.c_uint32), ("depth0", ctypes.c_uint32)] + \
               [(k, ctypes.c_void_p) for k in ("trace_evals", "comp_evals", "main_paths", "aux_paths", "comp_paths", "fri_evals",
                                               "fri_evals_sym", "fri_paths", "fri_paths_sym")]


def fe(x):
    return int(x).to_bytes(32, "big")


def parse_proof(b):
    """Minimal reader of the proof layout (reference src/starks/proof/stark.rs:161-218)."""
    p = 0

    def u64():
        nonlocal p
        v = struct.unpack(">Q", b[p:p + 8])[0]
        p += 8
        return v

    def take(n):
        nonlocal p
        v = b[p:p + n]
        p += n
        return v

    out = {"trace_length": u64()}
    out["trace_roots"] = [take(32) for _ in range(u64())]
    u64()
    ne = u64(); u64()
    out["ood"] = [take(32) for _ in range(ne)]
    out["row_width"] = u64()
    out["comp_root"] = take(32); u64()
    out["h1z"], out["h2z"] = take(32), take(32)
    out["fri_roots"] = [take(32) for _ in range(u64())]
    out["fri_last"] = take(32)
    out["rest"] = b[p:-8]
    out["nonce"] = struct.unpack(">Q", b[-8:])[0]
    return out


def test_round_level_calls_reproduce_the_oracle_proof(hip_lib, oracle):
    lib = hip_lib
    run = api.CairoRun.fibonacci(50)
    trace = run.main_trace()
    n, cols = trace.shape[0], trace.shape[1]
    options = (4, 4, 3, 2)
    want = oracle.cairo_prove(trace, run.public_inputs_c, options)
    W = parse_proof(want)
    ctx = api.Context(device=0)
    h = ctx._h
    opt = api.ProofOptions(*options).to_c()
    root = (ctypes.c_uint8 * 32)()
    tr = oracle.Transcript()
    u8p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    # round 1
    _lib.check(lib.sp_prove_setup(h, ctypes.c_uint64(n), 34, 18, 0, ctypes.byref(opt)))
    _lib.check(lib.sp_commit_trace(h, 0, u8p(trace), ctypes.c_uint64(n), 34, root))
    assert bytes(root) == W["trace_roots"][0]
    tr.append(bytes(root))
    rap = [tr.to_field() for _ in range(3)]
    rap_b = b"".join(fe(x) for x in rap)
    _lib.check(lib.sp_cairo_commit_aux(h, rap_b, ctypes.byref(run.public_inputs_c), root))
    assert bytes(root) == W["trace_roots"][1]
    tr.append(bytes(root))
    # out-of-order call is refused, state is not corrupted
    assert lib.sp_ood(h, fe(5), (ctypes.c_uint8 * (32 * 106))()) == _lib.SP_E_STATE
    # round 2: boundary constraints as the reference builds them (src/cairo/air.rs:777-849)
    pi = run.public_inputs_c
    alpha, z = rap[0], rap[1]
    prod = 1
    for a, v in run.public_memory():
        prod = prod * ((z - (a + alpha * v)) % P) % P
    perm_final = pow(z, len(run.public_memory()), P) * pow(prod, P - 2, P) % P
    val = lambda name: int.from_bytes(bytes(getattr(pi, name)), "big")
    bcs = [(19, 0, val("pc_init")), (17, 0, val("ap_init")), (19, pi.num_steps - 1, val("pc_final")), (17, pi.num_steps - 1, val("ap_final")),
           (48, n - 1, perm_final), (51, n - 1, 1), (34, 0, pi.range_check_min), (36, n - 1, pi.range_check_max)]
    barr = (Boundary * 8)()
    for i, (c, s, v) in enumerate(bcs):
        barr[i].col, barr[i].step = c, s
        ctypes.memmove(barr[i].value, fe(v), 32)
    coeffs = [tr.to_field() for _ in range(8)] + [tr.to_field() for _ in range(8)] + [tr.to_field() for _ in range(49)] + [tr.to_field() for _ in range(49)]
    _lib.check(lib.sp_composition(h, rap_b, barr, 8, b"".join(fe(x) for x in coeffs), 49, root))
    assert bytes(root) == W["comp_root"]
    tr.append(bytes(root))
    # round 3
    hinv = pow(3, P - 2, P)
    while True:
        zz = tr.to_field()
        if pow(zz * hinv % P, n * 4, P) != 1 and pow(zz, n, P) != 1:
            break
    ood = (ctypes.c_uint8 * (32 * (2 + 2 * 52)))()
    _lib.check(lib.sp_ood(h, fe(zz), ood))
    ood = bytes(ood)
    assert ood[:32] == W["h1z"] and ood[32:64] == W["h2z"]
    assert [ood[64 + 32 * i:96 + 32 * i] for i in range(104)] == W["ood"]
    tr.append(ood)
    # round 4
    gammas = [tr.to_field() for _ in range(2 + 104)]
    _lib.check(lib.sp_deep_fri_commit_begin(h, b"".join(fe(x) for x in gammas), root))
    fri_roots = [bytes(root)]
    tr.append(bytes(root))
    is_last = ctypes.c_int(0)
    while True:
        zeta = tr.to_field()
        _lib.check(lib.sp_fri_fold_commit(h, fe(zeta), root, ctypes.byref(is_last)))
        if is_last.value:
            break
        fri_roots.append(bytes(root))
        tr.append(bytes(root))
    assert fri_roots == W["fri_roots"]
    assert bytes(root) == W["fri_last"]
    tr.append(bytes(root))
    ch = tr.challenge()
    nonce = ctypes.c_uint64()
    _lib.check(lib.sp_grind(h, ch, ctypes.c_uint8(options[3]), ctypes.byref(nonce)))
    assert nonce.value == W["nonce"]
    tr.append(struct.pack(">Q", nonce.value))
    iotas = [tr.to_usize() % (n * 4) for _ in range(options[1])]
    op = Openings()
    _lib.check(lib.sp_open(h, (ctypes.c_uint64 * len(iotas))(*iotas), len(iotas), ctypes.byref(op)))
    assert (op.n_queries, op.n_cols) == (len(iotas), 52)
    # every opened value / path appears, in order, in the oracle's serialized openings
    rest = W["rest"]
    tev = ctypes.string_at(op.trace_evals, 32 * 52 * len(iotas))
    for s in range(len(iotas)):
        assert tev[s * 52 * 32:(s + 1) * 52 * 32] in rest
    mp = ctypes.string_at(op.main_paths, 32 * op.depth0 * len(iotas))
    for s in range(len(iotas)):
        assert mp[s * op.depth0 * 32:(s + 1) * op.depth0 * 32] in rest
    fe0 = ctypes.string_at(op.fri_evals, 32 * op.n_layers * len(iotas))
    for s in range(len(iotas)):
        assert fe0[s * op.n_layers * 32:(s + 1) * op.n_layers * 32] in rest
    ctx.close()


def test_mont_limbs_encoding_gives_the_same_proof(hip_lib, oracle):
    run = api.CairoRun.fibonacci(40)
    options = (4, 3, 3, 1)
    want = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
    trace_lw = run.main_trace(fe_encoding=api.SP_FE_MONT_LIMBS)  # lambdaworks FieldElement memory layout
    assert not np.array_equal(trace_lw, run.main_trace())
    with api.Context(device=0, fe_encoding=api.SP_FE_MONT_LIMBS) as ctx:
        got = ctx.cairo_prove(trace_lw, run.public_inputs_c, api.ProofOptions(*options))
    assert got == want


def test_commit_trace_columns_gives_the_root_of_the_row_major_call(hip_lib):
    """sp_commit_trace_columns (`trace.cols()`, what interpolate_and_commit starts from, prover.rs:130): same root as
    sp_commit_trace on the row-major table - in the context encoding, and strided in the device layout."""
    import numpy as np
    lib = hip_lib
    run = api.CairoRun.fibonacci(200)
    trace = run.main_trace()
    n = trace.shape[0]
    opt = api.ProofOptions(4, 4, 3, 2).to_c()
    u8p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    roots = []
    cols_be = np.ascontiguousarray(trace.transpose(1, 0, 2))
    wide = np.zeros((34, n + 16, 32), dtype=np.uint8)
    wide[:, :n] = api.fe_to_device(cols_be.reshape(-1, 32)).reshape(34, n, 32)
    with api.Context(device=0) as ctx:
        for how in ("rows", "columns", "device-layout columns"):
            root = (ctypes.c_uint8 * 32)()
            _lib.check(lib.sp_prove_setup(ctx._h, ctypes.c_uint64(n), 34, 18, 0, ctypes.byref(opt)))
            if how == "rows":
                _lib.check(lib.sp_commit_trace(ctx._h, 0, u8p(trace), ctypes.c_uint64(n), 34, root))
            elif how == "columns":
                _lib.check(lib.sp_commit_trace_columns(ctx._h, 0, u8p(cols_be), ctypes.c_uint64(n), 34, ctypes.c_uint64(0), 0, root))
            else:
                _lib.check(lib.sp_commit_trace_columns(ctx._h, 0, u8p(wide), ctypes.c_uint64(n), 34, ctypes.c_uint64(n + 16), 1, root))
            roots.append(bytes(root))
        assert lib.sp_commit_trace_columns(ctx._h, 0, u8p(cols_be), ctypes.c_uint64(n), 34, ctypes.c_uint64(0), 0, root) == _lib.SP_E_STATE   # segment 0 is committed
    assert roots[0] == roots[1] == roots[2]

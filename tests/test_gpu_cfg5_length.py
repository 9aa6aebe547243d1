"""BASELINE configs[4]'s trace LENGTH (n = 2^24) on the one GPU of the test box: the fine-grained transforms at 2^23 and 2^24
against the CPU oracle element for element (pass schedules of log n = 23, 24 - four passes - are code paths no smaller size
takes), a whole single-GPU proof of a 2^24-row Cairo trace (2^26-key sorts in the auxiliary trace, 2^26-leaf trees, the
2n-point composition at 2^25 points), and a second point for the per-rank device-memory extrapolation of
tests/test_gpu_cfg5_shape.py (rank 0 of eight at n = 2^22).  The eight-GPU run itself needs hardware this box does not have."""
import ctypes
import hashlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from lambdaworks_cairo_prover_amd import _lib, api

pytestmark = pytest.mark.gpu


def _free_device_bytes():
    hip = ctypes.CDLL("libamdhip64.so")          # the runtime the library itself uses
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipSetDevice(0) == 0 and hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def _rand_felts(seed, n):
    raw = np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)   # (random.randbytes overflows above 2^28 bytes)
    raw[:, 0] &= 0x07  # < 2^251 < p
    return raw


@pytest.mark.parametrize("k", [23, 24])
def test_ntt_and_lde_at_cfg5_lengths_match_oracle(hip_ctx, oracle, k):
    """sp_ntt forward / inverse / coset forward / coset inverse and sp_lde (blowup 2) at n = 2^23, 2^24: every element equals
    the oracle's (the oracle calls run side by side on host threads: ctypes releases the GIL)."""
    n = 1 << k
    x = _rand_felts(5000 + k, n)
    h = api.felts_to_bytes([3])
    with ThreadPoolExecutor(max_workers=5) as pool:
        want = {"fwd": pool.submit(oracle.ntt, x), "inv": pool.submit(oracle.ntt, x, True), "cfwd": pool.submit(oracle.ntt, x, False, 3),
                "cinv": pool.submit(oracle.ntt, x, True, 3), "lde": pool.submit(oracle.lde, x, 2, 3)}
        got_fwd = hip_ctx.ntt(x)
        assert np.array_equal(hip_ctx.ntt(got_fwd, inverse=True), x)                       # round trip on the device
        got_inv = hip_ctx.ntt(x, inverse=True)
        got_cfwd = hip_ctx.ntt(x, coset=h)
        got_cinv = hip_ctx.ntt(x, inverse=True, coset=h)
        got_lde = hip_ctx.lde(x.reshape(1, n, 32), 2, h)[0]
        assert np.array_equal(got_fwd, want["fwd"].result()), "forward NTT"
        assert np.array_equal(got_inv, want["inv"].result()), "inverse NTT"
        assert np.array_equal(got_cfwd, want["cfwd"].result()), "coset forward NTT"
        assert np.array_equal(got_cinv, want["cinv"].result()), "coset inverse NTT"
        assert np.array_equal(got_lde, want["lde"].result()), "LDE (blowup 2)"


def test_whole_proof_of_a_2_24_row_trace_on_one_gpu(hip_lib):
    """n = 2^24 rows x 52 columns (BASELINE configs[4]'s trace length; blowup 4: N = 2^26 LDE points, 112 GB of LDE columns):
    the product verifier accepts the proof and rejects it after a byte flip; the run itself (sp_cairo_prove_run: trace built on
    the device, and the run's page-locked columns with that option off), the reference's row-major host table (sp_cairo_prove) and the device-resident table
    (sp_cairo_prove_dev) give the same bytes.  Blowup 2 when less than 250 GB of device memory is free."""
    free = _free_device_bytes()
    if free < 150e9:
        pytest.skip("needs >= 150 GB of free device memory")
    blowup = 4 if free >= 250e9 else 2
    opt = api.ProofOptions(blowup, 30, 3, 16)
    with api.Context(device=0) as ctx:
        run = api.CairoRun.fibonacci(2390000)            # 16 730 009 steps -> 2^24 rows
        assert run.n_rows == 1 << 24 and run.n_cols == 34
        proof = ctx.cairo_prove_run(run, opt)            # the trace built on the device: 0.4 GB of registers + 0.7 GB of memory go up
        info, up, dev_bytes, rounds = ctx.last_proof_info(), ctx.last_upload_stats(), ctx.prover_device_bytes(), ctx.last_round_ms()
        assert info["composition_path"] == 1             # the 2n-point composition (2^25 points) after a clean trace check
        assert up["kind"].startswith("run image") and up["bytes"] < 0.1 * (1 << 24) * 34 * 32
        ctx.set_option(api.SP_OPT_DEVICE_TRACE, 0)       # and the run's own page-locked table (18 GB), column group by column group
        assert ctx.cairo_prove_run(run, opt) == proof
        up_cols = ctx.last_upload_stats()
        assert up_cols["kind"].startswith("host columns") and up_cols["bytes"] == (1 << 24) * 34 * 32
        ctx.set_option(api.SP_OPT_DEVICE_TRACE, 1)
        assert api.cairo_verify(proof, run.public_inputs_c, opt)
        bad = bytearray(proof)
        bad[len(bad) // 3] ^= 0x10
        assert not api.cairo_verify(bytes(bad), run.public_inputs_c, opt)
        assert not api.cairo_verify(proof, run.public_inputs_c, api.ProofOptions(blowup * 2, 30, 3, 16))
        want = hashlib.sha256(proof).hexdigest()
        print(f"\n2^24 rows, blowup {blowup}: {sum(rounds):.0f} ms of device time (rounds {[round(r) for r in rounds[1:]]}), "
              f"{dev_bytes / 1e9:.0f} GB held by the prover, upload {up['dma_gbs']} GB/s, exposed {up['exposed_ms']} ms")
        # configs[4] names Poseidon Merkle trees: the optional backend at this length (2^26-leaf trees of field-element digests)
        ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_POSEIDON)
        pproof = ctx.cairo_prove_run(run, opt)
        prounds = ctx.last_round_ms()
        assert len(pproof) == len(proof) and pproof != proof
        assert api.cairo_verify(pproof, run.public_inputs_c, opt, api.SP_MERKLE_POSEIDON)
        assert not api.cairo_verify(pproof, run.public_inputs_c, opt)
        bad = bytearray(pproof)
        bad[len(bad) // 3] ^= 0x10
        assert not api.cairo_verify(bytes(bad), run.public_inputs_c, opt, api.SP_MERKLE_POSEIDON)
        print(f"the same with Poseidon trees: {sum(prounds):.0f} ms of device time (rounds {[round(r) for r in prounds[1:]]})")
        ctx.set_option(api.SP_OPT_MERKLE_BACKEND, api.SP_MERKLE_KECCAK256)
        trace = run.main_trace()                         # the reference's row-major TraceTable: 18 GB of pageable memory
        assert hashlib.sha256(ctx.cairo_prove(trace, run.public_inputs_c, opt)).hexdigest() == want
        if _free_device_bytes() >= trace.nbytes + (2 << 30):
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
            hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            hip.hipFree.argtypes = [ctypes.c_void_p]
            dev = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(dev), trace.nbytes) == 0
            try:
                assert hip.hipMemcpy(dev, trace.ctypes.data, trace.nbytes, 1) == 0
                assert hashlib.sha256(ctx.cairo_prove_dev(dev.value, 1 << 24, 34, run.public_inputs_c, opt)).hexdigest() == want
            finally:
                hip.hipFree(dev)


def test_rank0_device_bytes_second_point_for_the_cfg5_extrapolation(hip_lib):
    """tests/test_gpu_cfg5_shape.py measures 24.0 GB per rank at n = 2^21 (blowup 16, eight ranks) and extrapolates linearly to
    n = 2^24.  A second point: the buffers rank 0 of eight allocates for n = 2^22 (sp_prove_setup only - no collective runs)
    must be twice those for n = 2^21 to within 1 %, and eight times the latter must fit the 288 GB of an MI355X."""
    if _free_device_bytes() < 60e9:
        pytest.skip("needs ~50 GB of free device memory")
    lib = hip_lib
    hook = api.ALLGATHER_FN(lambda user, send, recv, nbytes: -1)      # never called: setup allocates, it does not communicate
    sizes = {}
    for logn in (21, 22):
        with api.Context(device=0) as ctx:
            _lib.check(lib.sp_set_collective(ctx._h, 8, 0, hook, None))
            opt = api.ProofOptions(16, 30, 3, 12).to_c()
            _lib.check(lib.sp_prove_setup(ctx._h, ctypes.c_uint64(1 << logn), 34, 18, 0, ctypes.byref(opt)))
            sizes[logn] = ctx.prover_device_bytes()
    print(f"\nrank 0 of 8, blowup 16: {sizes[21] / 1e9:.2f} GB at n = 2^21, {sizes[22] / 1e9:.2f} GB at n = 2^22 "
          f"(ratio {sizes[22] / sizes[21]:.4f}); x4 -> {4 * sizes[22] / 1e9:.0f} GB at n = 2^24")
    assert abs(sizes[22] / sizes[21] - 2.0) < 0.02
    assert 4 * sizes[22] < 0.9 * 288e9

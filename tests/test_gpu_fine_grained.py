"""Parity of the fine-grained HIP entry points (the lambdaworks seam of SURVEY.md §8(b)) against the CPU oracle.
Bit-exact: integer arithmetic mod p, compared as canonical 32-byte big-endian records."""
import random

import numpy as np
import pytest

from lambdaworks_cairo_prover_amd import api

pytestmark = pytest.mark.gpu
P = api.P


def rand_felts(rng, n):
    return api.felts_to_bytes([rng.randrange(P) for _ in range(n)])


@pytest.mark.parametrize("k", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 16, 18])
def test_ntt_forward_inverse_matches_oracle(hip_ctx, oracle, k):
    rng = random.Random(1000 + k)
    x = rand_felts(rng, 1 << k)
    want = oracle.ntt(x)
    got = hip_ctx.ntt(x)
    assert np.array_equal(got, want)
    back = hip_ctx.ntt(got, inverse=True)
    assert np.array_equal(back, x)
    assert np.array_equal(hip_ctx.ntt(x, inverse=True), oracle.ntt(x, inverse=True))


@pytest.mark.parametrize("k", [1, 4, 9, 12, 14])
def test_ntt_coset_variants(hip_ctx, oracle, k):
    rng = random.Random(2000 + k)
    x = rand_felts(rng, 1 << k)
    h = api.felts_to_bytes([3])
    assert np.array_equal(hip_ctx.ntt(x, coset=h), oracle.ntt(x, coset=3))
    assert np.array_equal(hip_ctx.ntt(x, inverse=True, coset=h), oracle.ntt(x, inverse=True, coset=3))


def test_ntt_edge_inputs(hip_ctx, oracle):
    # all zero, all p-1, delta
    n = 1 << 11
    for vals in ([0] * n, [P - 1] * n, [1] + [0] * (n - 1), [0] * (n - 1) + [P - 1]):
        x = api.felts_to_bytes(vals)
        assert np.array_equal(hip_ctx.ntt(x), oracle.ntt(x))


def test_ntt_rejects_non_power_of_two(hip_ctx):
    with pytest.raises(api.SpError):
        hip_ctx.ntt(api.felts_to_bytes([1, 2, 3]))


def test_ntt_2_22_properties(hip_ctx, oracle):
    """BASELINE config #2 size: round trip and linearity (size-independent properties) + spot values vs a direct sum."""
    rng = random.Random(22)
    n = 1 << 22
    raw = np.frombuffer(rng.randbytes(32 * n), dtype=np.uint8).reshape(n, 32).copy()
    raw[:, 0] &= 0x07  # < 2^251 < p
    y = hip_ctx.ntt(raw)
    assert np.array_equal(hip_ctx.ntt(y, inverse=True), raw)
    assert np.array_equal(y, oracle.ntt(raw))


@pytest.mark.parametrize("k,blowup,cols", [(3, 2, 1), (6, 4, 3), (10, 4, 2), (12, 8, 2), (13, 16, 1)])
def test_lde_matches_oracle(hip_ctx, oracle, k, blowup, cols):
    rng = random.Random(3000 + k)
    n = 1 << k
    coeffs = np.stack([rand_felts(rng, n) for _ in range(cols)])
    got = hip_ctx.lde(coeffs, blowup, api.felts_to_bytes([3]))
    for j in range(cols):
        assert np.array_equal(got[j], oracle.lde(coeffs[j], blowup, 3)), (k, blowup, j)


@pytest.mark.parametrize("n,width", [(1, 1), (2, 1), (4, 1), (8, 2), (64, 18), (256, 34), (1024, 1), (512, 43), (128, 5), (16384, 2), (32768, 1)])
def test_merkle_matches_oracle(hip_ctx, oracle, n, width):
    rng = random.Random(4000 + n + width)
    rows = rand_felts(rng, n * width).reshape(n, width, 32)
    root, nodes = hip_ctx.merkle_build(rows, want_nodes=True)
    oroot, onodes = oracle.merkle_build(rows, want_nodes=True)
    assert root == oroot
    assert np.array_equal(nodes, onodes)


def test_batch_inverse(hip_ctx, oracle):
    rng = random.Random(5)
    for n in (1, 2, 63, 64, 65, 5000, 1 << 16, (1 << 16) + 16, (1 << 16) + 5, 3 << 16):   # >= 2^16 and a multiple of 16: two-level form
        x = api.felts_to_bytes([rng.randrange(1, P) for _ in range(n)])
        assert np.array_equal(hip_ctx.batch_inverse(x), oracle.batch_inverse(x))
    with pytest.raises(api.SpError):
        hip_ctx.batch_inverse(api.felts_to_bytes([5, 0, 7]))
    big = [rng.randrange(1, P) for _ in range(1 << 16)]
    for zero_at in (0, 1, 17, (1 << 16) - 1):
        y = list(big)
        y[zero_at] = 0
        with pytest.raises(api.SpError):
            hip_ctx.batch_inverse(api.felts_to_bytes(y))


class _HipBuffer:
    """Device memory through the HIP runtime the library itself is linked against (importing torch after the library has
    initialised HIP would bring a second runtime into the process)."""

    def __init__(self, host: np.ndarray):
        import ctypes
        self._ct = ctypes
        self._hip = ctypes.CDLL("libamdhip64.so")
        self.nbytes = host.nbytes
        self.ptr = ctypes.c_void_p()
        assert self._hip.hipMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(self.nbytes)) == 0
        assert self._hip.hipMemcpy(self.ptr, host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(self.nbytes), 1) == 0

    def to_host(self, dtype=np.uint8):
        out = np.empty(self.nbytes, dtype=np.uint8)
        assert self._hip.hipMemcpy(out.ctypes.data_as(self._ct.c_void_p), self.ptr, self._ct.c_size_t(self.nbytes), 2) == 0
        return out

    def free(self):
        self._hip.hipFree(self.ptr)


@pytest.mark.parametrize("k,batch", [(10, 1), (13, 5), (22, 1)])
def test_ntt_dev_batched_device_entry_point(hip_ctx, oracle, k, batch):
    """sp_ntt_dev (the entry point bench.py times): asynchronous, batched, device layout in place - the CPU oracle's values element
    for element, at 2^22 on full-range residues with a block at p - 1 ... p - 2^16; the event timers return plausible durations."""
    n = 1 << k
    rng = random.Random(100 + k)
    def full_range(seed):
        """uniform residues below 2^251 as raw big-endian bytes, and - the values a lazy reduction is most likely to get wrong - a block
        of 2^16 elements counting down from p - 1 (this entry point used to see values below 2^62 at this size)"""
        raw = np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
        raw[:, 0] &= 0x07
        top = np.frombuffer(b"".join((api.P - 1 - i).to_bytes(32, "big") for i in range(1 << 16)), dtype=np.uint8).reshape(-1, 32)
        raw[n // 2:n // 2 + (1 << 16)] = top
        return raw
    cols = [api.felts_to_bytes([rng.randrange(api.P) for _ in range(n)]) if k <= 13 else full_range(k + v) for v in range(batch)]
    dev_layout = np.ascontiguousarray(np.concatenate([api.fe_to_device(c) for c in cols]))
    d = _HipBuffer(dev_layout)
    try:
        hip_ctx.timer_start()
        hip_ctx.ntt_dev(d.ptr.value, n, batch)
        region_ms = hip_ctx.timer_stop()
        call_ms = hip_ctx.last_kernel_ms()
        hip_ctx.sync()
        got = api.fe_from_device(d.to_host().reshape(-1, 32)).reshape(batch, n, 32)
        for v in range(batch):
            assert np.array_equal(got[v], oracle.ntt(cols[v]))      # the oracle itself, at every size (2^22: the size bench.py times)
        assert 0.0 < call_ms <= region_ms * 1.5 + 0.05
        hip_ctx.ntt_dev(d.ptr.value, n, batch, inverse=True)   # the inverse restores the input
        hip_ctx.sync()
        assert np.array_equal(d.to_host().reshape(-1, 32), dev_layout.reshape(-1, 32))
    finally:
        d.free()


@pytest.mark.parametrize("k,width", [(6, 1), (11, 2), (13, 34)])
def test_merkle_build_dev_device_entry_point(hip_ctx, oracle, k, width):
    """sp_merkle_build_dev (the entry point bench.py times for the hash passes): column-major device-layout columns in,
    every node of the tree out - the same nodes as the oracle's MerkleTree::build over the same rows."""
    n = 1 << k
    rng = np.random.default_rng(7 * k + width)
    cols = [api.felts_to_bytes([int(x) for x in rng.integers(0, 2**62, size=n)] if j % 3 else
                               [random.Random(j * 1000 + i).randrange(api.P) for i in range(n)]) for j in range(width)]
    dev_layout = np.ascontiguousarray(np.concatenate([api.fe_to_device(c) for c in cols]))
    d = _HipBuffer(dev_layout)
    nodes = _HipBuffer(np.zeros((2 * n - 1) * 32, dtype=np.uint8))
    try:
        hip_ctx.merkle_build_dev(d.ptr.value, n, width, n, nodes.ptr.value)
        hip_ctx.sync()
        got = nodes.to_host().reshape(2 * n - 1, 32)
        rows = np.stack(cols, axis=1)                      # (n, width, 32)
        want_root, want_nodes = oracle.merkle_build(rows, want_nodes=True)
        assert got[0].tobytes() == want_root
        assert np.array_equal(got, want_nodes)
    finally:
        d.free()
        nodes.free()


@pytest.mark.parametrize("k", [13, 18])
def test_ntt_extreme_values_across_passes(hip_ctx, oracle, k):
    """The passes defer reduction (values up to (2 + 2 stages) p inside a pass, < 2p between passes): inputs at the top of the
    range (p - 1 everywhere, alternating 0 / p - 1, p - 1 on one residue class) must still come out canonical and
    equal to the oracle through two- and three-pass transforms, forward, inverse and on a coset."""
    n = 1 << k
    pm1 = api.felts_to_bytes([P - 1])[0]
    zero = api.felts_to_bytes([0])[0]
    patterns = []
    patterns.append(np.tile(pm1, (n, 1)))
    alt = np.tile(zero, (n, 1)); alt[::2] = pm1
    patterns.append(alt)
    sparse = np.tile(zero, (n, 1)); sparse[3::64] = pm1
    patterns.append(sparse)
    h = api.felts_to_bytes([3])
    for x in patterns:
        assert np.array_equal(hip_ctx.ntt(x), oracle.ntt(x))
        assert np.array_equal(hip_ctx.ntt(x, inverse=True), oracle.ntt(x, inverse=True))
        assert np.array_equal(hip_ctx.ntt(x, coset=h), oracle.ntt(x, coset=3))


def test_lde_extreme_coefficients(hip_ctx, oracle):
    k, blowup, cols = 12, 8, 3
    n = 1 << k
    pm1 = api.felts_to_bytes([P - 1])[0]
    coeffs = np.tile(pm1, (cols * n, 1)).reshape(cols, n, 32).copy()
    coeffs[1, ::2] = 0
    coeffs[2, 1:] = 0
    got = hip_ctx.lde(coeffs, blowup, api.felts_to_bytes([3]))
    want = np.stack([oracle.lde(coeffs[c], blowup, 3) for c in range(cols)])
    assert np.array_equal(got, want)


def test_plain_c_caller_runs_on_the_gpu(hip_ctx, tmp_path):
    """examples/c_abi_smoke.c: a C99 program through the C ABI - an NTT round trip on the device, then the whole path of the reference's
    CLI (run a program, pre-warm, prove from the run with the trace built on the device, verify, frame the proof file)."""
    import subprocess
    from test_capi import _build_c_example
    out = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "ntt round trip: rc 0, identical" in out.stdout, out.stdout + out.stderr
    assert "cairo proof: rc 0 (ok), 709 steps, 1024 x 34 trace" in out.stdout and "verifier accepts" in out.stdout, out.stdout + out.stderr


# ---- sp_fe_mul: the kernels' own Montgomery product and square on operands the test chooses -------------------------------------
P_ = api.P
R_ = 1 << 256
RINV_ = pow(R_, -1, P_)


def _to_lw(vals):
    """Montgomery representatives (integers < p) as lambdaworks' in-memory FieldElement: 4 x u64, most significant limb first."""
    out = np.empty((len(vals), 32), dtype=np.uint8)
    for k, v in enumerate(vals):
        for i in range(4):
            out[k, 8 * i:8 * i + 8] = np.frombuffer(((v >> (64 * (3 - i))) & (2**64 - 1)).to_bytes(8, "little"), dtype=np.uint8)
    return out


def _from_lw(arr):
    return [sum(int.from_bytes(bytes(rec[8 * i:8 * i + 8]), "little") << (64 * (3 - i)) for i in range(4)) for rec in arr]


def _rows_whose_reduction_overflows(a, b, square=False):
    """The CIOS rows of csrc/fp.h on Python integers: the rows where E = 17 m + {x7, x6} exceeds 64 bits, i.e. where the device's
    reduction-through-two-multiply-adds needs v_mad_u64_u32's carry-out (about one row in 2^27 on random operands)."""
    hits, T = [], 0
    for i in range(8):
        ai = (a >> (32 * i)) & 0xFFFFFFFF
        if square:      # row i of fe_sqr_lazy: the diagonal term once (at column i), every product with a higher limb doubled
            S = T + ai * ((ai << (32 * i)) + 2 * ((a >> (32 * (i + 1))) << (32 * (i + 1))))
        else:
            S = T + ai * b
        m = (-S) & 0xFFFFFFFF
        X = ((S + m) >> 192) & (2**64 - 1)
        if 17 * m + X >= 2**64:
            hits.append(i)
        assert (S + m * P_) % 2**32 == 0
        T = (S + m * P_) >> 32
    return hits, T


def test_fe_mul_matches_python_integers(hip_ctx):
    """FieldElement Mul / square through sp_fe_mul: random and extreme canonical operands against Python integers."""
    rng = random.Random(61)
    edge = [0, 1, 2, P_ - 1, P_ - 2, 2**251, 2**251 - 1, 2**192, 17 * 2**192, 2**64 - 1, 2**128 + 1, (P_ - 1) // 2, 3]
    a = edge + [rng.randrange(P_) for _ in range(3000)]
    b = [rng.choice(edge) for _ in edge] + [rng.randrange(P_) for _ in range(3000)]
    got = api.bytes_to_felts(hip_ctx.fe_mul(api.felts_to_bytes(a), api.felts_to_bytes(b)))
    assert got == [x * y % P_ for x, y in zip(a, b)]
    got = api.bytes_to_felts(hip_ctx.fe_mul(api.felts_to_bytes(a)))
    assert got == [x * x % P_ for x in a]


def test_fe_mul_rows_whose_reduction_carries_out():
    """Round 6: the device reduces a CIOS row through two multiply-adds, E = 17 m + {x7, x6} and F = 2^27 m + {x8, hi(E)}, and takes the
    overflow of E from v_mad_u64_u32's carry-out (inline assembly, csrc/fp.h sp_row_reduce).  Random operands overflow E about once in 2^27
    rows - no random test ever walks that path - so the operands are BUILT: lower limbs of a zero, so that row i starts from an empty
    accumulator, and b = K div a_i with the bits 192..255 of K all ones: x6 = x7 = 0xffffffff in that row.  Every row 0..7 for the product
    (with junk in the limbs above), row 0 for the square; the Python model of the rows confirms that each case really overflows where
    it should, and the device's Montgomery limbs must equal a b R^-1 mod p."""
    rng = random.Random(62)
    A, B, want, hit_rows = [], [], [], []
    for i in range(8):
        for _ in range(40):
            ai = rng.randrange(1 << 8, 1 << (27 if i == 7 else 32)) | 1
            j = rng.randrange(0, max(1, ai // 64))
            K = (((j << 64) | (2**64 - 1)) << 192) | rng.randrange(2**191, 2**192)
            b = K // ai
            a = (ai << (32 * i)) | (rng.randrange(2**256) >> (32 * (i + 1)) << (32 * (i + 1)))
            a &= (1 << 251) - 1
            a |= ai << (32 * i)
            if not (b < P_ and a < P_):
                continue
            hits, T = _rows_whose_reduction_overflows(a, b)
            if i not in hits:
                continue
            A.append(a); B.append(b); hit_rows.append(i)
            want.append(a * b * RINV_ % P_)
            assert T % P_ == want[-1]
    assert set(hit_rows) == set(range(8)) and len(A) >= 100, (sorted(set(hit_rows)), len(A))
    # squares: row 0 only (the operand's upper limbs are what is solved for)
    SQ, sq_want = [], []
    for _ in range(60):
        a0 = rng.randrange(1 << 28, 1 << 32) | 1
        K = (((rng.randrange(0, 4) << 64) | (2**64 - 1)) << 192) | rng.randrange(2**191, 2**192)
        U = (K - a0 * a0) // (2 * a0 << 32)
        a = a0 + (U << 32)
        if a >= P_:
            continue
        hits, T = _rows_whose_reduction_overflows(a, a, square=True)
        if 0 not in hits:
            continue
        SQ.append(a); sq_want.append(a * a * RINV_ % P_)
    assert len(SQ) >= 20
    with api.Context(device=0, fe_encoding=api.SP_FE_MONT_LIMBS) as ctx:
        got = _from_lw(ctx.fe_mul(_to_lw(A), _to_lw(B)))
        bad = [(hit_rows[k], hex(A[k]), hex(B[k])) for k in range(len(A)) if got[k] != want[k]]
        assert not bad, bad[:3]
        got = _from_lw(ctx.fe_mul(_to_lw(B), _to_lw(A)))          # the other operand order: no carry-out row, same product
        assert got == want
        got = _from_lw(ctx.fe_mul(_to_lw(SQ)))
        assert got == sq_want

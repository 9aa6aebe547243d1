"""csrc/fp.h on the host: the deferred-reduction helpers of the NTT passes (fe_reduce_lazy_2p, fe_canonical_lazy,
fe_sub_add_kp, fe_sub_add_2p, fe_neg_one) and the Montgomery product on operands beyond p, against Python integers.
The same header is compiled for gfx950; the GPU suites pin the kernels, this pins the arithmetic they rely on."""
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2**251 + 17 * 2**192 + 1
R = 2**256
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    exe = str(tmp_path_factory.mktemp("fp") / "fp_host_check")
    subprocess.check_call([HIPCC, "-O2", "-x", "hip", "--cuda-host-only", "-include", "cstring",
                           "-I", os.path.join(ROOT, "lambdaworks_cairo_prover_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_src", "fp_host_check.cpp"), "-o", exe], stderr=subprocess.DEVNULL)

    def run(cases):
        text = "".join(f"{op} {a:064x} {b:064x} {k}\n" for op, a, b, k in cases)
        out = subprocess.run([exe], input=text.encode(), capture_output=True, check=True).stdout.decode().split()
        assert len(out) == len(cases)
        return [int(x, 16) for x in out]
    return run


def edge_values():
    vals = [0, 1, P - 1, P, P + 1, 2 * P - 1, 2 * P, 2**251 - 1, 2**251, 2**255, R - 1, 31 * P, 31 * P + 5, 22 * P - 1, 16 * P, 30 * P - 1]
    for k in range(32):
        vals += [k * P, k * P + 1, max(k * P - 1, 0), k * 2**251, min(k * 2**251 + 2**251 - 1, R - 1)]
    return vals


def test_quotient_estimate_reductions(probe):
    rng = random.Random(11)
    vals = edge_values() + [rng.randrange(R) for _ in range(4000)]
    lazy = probe([("lazy2p", v, 0, 0) for v in vals])
    canon = probe([("canon", v, 0, 0) for v in vals])
    for v, r, c in zip(vals, lazy, canon):
        assert r % P == v % P and r < 2 * P, hex(v)
        assert c == v % P, hex(v)


def test_biased_differences(probe):
    rng = random.Random(12)
    cases, want = [], []
    for k in (2, 4, 8):
        for _ in range(1500):
            a, b = rng.randrange(k * P), rng.randrange(k * P)
            cases.append(("subkp", a, b, k)); want.append(a - b + k * P)
        for a, b in ((0, k * P - 1), (k * P - 1, 0), (k * P - 1, k * P - 1), (0, 0)):
            cases.append(("subkp", a, b, k)); want.append(a - b + k * P)
    for _ in range(1500):   # DIT: u up to 28p, t < 2p
        a, b = rng.randrange(28 * P), rng.randrange(2 * P)
        cases.append(("sub2p", a, b, 0)); want.append(a - b + 2 * P)
    got = probe(cases)
    for (op, a, b, k), g, w in zip(cases, got, want):
        assert 0 <= w < R and g == w, (op, hex(a), hex(b), k)


def test_lazy_product_accepts_any_256_bit_operand(probe):
    rng = random.Random(13)
    rinv = pow(R, -1, P)
    cases = [("mullazy", rng.randrange(R), rng.randrange(P), 0) for _ in range(3000)]
    cases += [("mullazy", R - 1, P - 1, 0), ("mullazy", R - 1, 0, 0), ("mullazy", 0, P - 1, 0), ("mullazy", 31 * P, P - 1, 0)]
    for (op, a, b, k), g in zip(cases, probe(cases)):
        assert g < 2 * P and g % P == a * b * rinv % P
    cases = [("mul", rng.randrange(P), rng.randrange(P), 0) for _ in range(1000)]
    for (op, a, b, k), g in zip(cases, probe(cases)):
        assert g == a * b * rinv % P


def test_dedicated_square(probe):
    """fe_sqr_lazy (triangular CIOS rows: 36 + 8 multiply-adds): a^2 / R mod p in [0, 2p) for every a < 2^253, fe_sqr canonical."""
    rng = random.Random(16)
    vals = [0, 1, P - 1, P, P + 1, 2 * P - 1, 2**251, 2**252, 2**253 - 1, 2**32 - 1, 2**64 - 1, (2**253 - 1) ^ (2**31), 0x80000000 * (2**32 + 1)]
    vals += [(1 << k) - 1 for k in range(1, 254, 7)] + [((1 << 31) << (32 * j)) for j in range(7)] + [rng.randrange(2**253) for _ in range(4000)]
    rinv = pow(R, -1, P)
    for a, g in zip(vals, probe([("sqrlazy", a, 0, 0) for a in vals])):
        assert g < 2 * P and g % P == a * a * rinv % P, hex(a)
    canon = [v % P for v in vals]
    for a, g in zip(canon, probe([("sqr", a, 0, 0) for a in canon])):
        assert g == a * a * rinv % P, hex(a)


def test_minus_one(probe):
    assert probe([("negone", 0, 0, 0)])[0] == (P - R % P) % P


def test_inverse_addition_chain(probe):
    """fe_inv (Montgomery in, Montgomery out): the division-step inversion (600 Bernstein-Yang steps on nine 30-bit limbs) and the
    Fermat chain it replaced (a^(p-2) through the 2^192 - 1 doubling chain) against Python integers, on edge values - powers of two,
    small values, values next to p, operands in [p, 2p) - and random ones; zero maps to zero."""
    rng = random.Random(14)
    vals = [1, 2, 3, P - 1, P - 2, R % P, 3 * R % P, (P + 1) // 2, 2**251, 2**250, 2**30 - 1, 2**30, 2**60, 2**240 + 1]
    vals += [2**k for k in range(0, 251, 7)] + [P - 2**k for k in range(0, 251, 11)] + list(range(1, 60))
    vals += [rng.randrange(1, P) for _ in range(3000)]
    for op in ("inv", "invfermat"):
        for a, g in zip(vals, probe([(op, a, 0, 0) for a in vals])):
            # a = x R, result = x^-1 R  ->  a * g = R^2 (mod p)
            assert g < P and a * g % P == R * R % P, (op, hex(a))
    assert probe([("inv", 0, 0, 0)]) == [0]
    lazy = [P + 5, 2 * P - 1, P + rng.randrange(P)]          # lazily reduced operands are accepted
    for a, g in zip(lazy, probe([("inv", a, 0, 0) for a in lazy])):
        assert g < P and a * g % P == R * R % P


def test_from_montgomery_by_64_bit_rounds(probe):
    """fe_from_mont: a R^-1 mod p, canonical, for every canonical input (the conversion in front of every leaf hash)."""
    rng = random.Random(15)
    rinv = pow(R, -1, P)
    vals = [0, 1, 2**64 - 1, 2**64, 2**128 - 1, 2**192, P - 1, P - 2**64, R % P] + [rng.randrange(P) for _ in range(4000)]
    vals += [(k << 64) | lo for k in (0, 1, 2**187 - 1) for lo in (0, 1, 2**64 - 1)]
    for a, g in zip(vals, probe([("frommont", a, 0, 0) for a in vals])):
        assert g == a * rinv % P, hex(a)

"""Pins the CPU oracle against every known-answer test the reference holds for this path."""
import random

import numpy as np
import pytest

P = 2**251 + 17 * 2**192 + 1


def test_keccak256_kats(oracle):
    # original Keccak padding, as sha3::Keccak256 (reference src/starks/grinding.rs:1,25)
    assert oracle.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert oracle.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"


def test_grinding_known_answer(oracle):
    # reference src/starks/grinding.rs:54-77: factor 10 -> nonce 33
    ch = bytes([226, 27, 133, 168, 62, 203, 20, 59, 122, 230, 227, 33, 76, 44, 53, 150, 200, 45, 136, 162, 249, 239, 142,
                90, 204, 191, 45, 4, 53, 22, 103, 240])
    assert oracle.grinding_nonce(ch, 10) == 33


def test_field_ops_against_python_ints(oracle):
    rng = random.Random(1)
    for _ in range(300):
        a, b = rng.randrange(P), rng.randrange(P)
        assert oracle.fe_op(0, a, b) == (a + b) % P
        assert oracle.fe_op(1, a, b) == (a - b) % P
        assert oracle.fe_op(2, a, b) == a * b % P
        assert oracle.fe_op(5, a) == (-a) % P
    for a in (1, 2, P - 1, rng.randrange(1, P)):
        assert oracle.fe_op(3, a) == pow(a, P - 2, P)
    for a, b in ((0, 0), (P - 1, P - 1), (P - 1, 1), (1, P - 1), (2**251, 2**251)):
        assert oracle.fe_op(2, a, b) == a * b % P
        assert oracle.fe_op(0, a, b) == (a + b) % P
        assert oracle.fe_op(1, a, b) == (a - b) % P


def test_permutation_column_known_answer(oracle):
    # reference src/cairo/air.rs:1410-1452 (generate_memory_permutation_argument_column): a = [3,1,2], v = [5,1,2],
    # sorted a' = [1,2,3], v' = [1,2,5], alpha = 15, z = 10  ->  two published hex field elements, then 1.
    z, alpha = 10, 15
    a, v, ap, vp = [3, 1, 2], [5, 1, 2], [1, 2, 3], [1, 2, 5]
    prod, out = 1, []
    for i in range(3):
        num = oracle.fe_op(1, z, oracle.fe_op(0, a[i], oracle.fe_op(2, alpha, v[i])))
        den = oracle.fe_op(3, oracle.fe_op(1, z, oracle.fe_op(0, ap[i], oracle.fe_op(2, alpha, vp[i]))))
        prod = oracle.fe_op(2, prod, oracle.fe_op(2, num, den))
        out.append(prod)
    assert out == [0x2aaaaaaaaaaaab0555555555555555555555555555555555555555555555561,
                   0x1745d1745d174602e8ba2e8ba2e8ba2e8ba2e8ba2e8ba2e8ba2e8ba2e8ba2ec, 1]


def test_roots_of_unity(oracle):
    # SURVEY.md Appendix A (confirmed by the golden proofs)
    assert oracle.primitive_root(10) == 0x659d83946a03edd72406af6711825f5653d9e35dc125289a206c054ec89c4f1
    assert oracle.primitive_root(22) == 0x3e4383531eeac7c9822fb108d24a344d841544dd6482f17ead331453e3a2f4b
    for k in (1, 5, 19, 28):
        w = oracle.primitive_root(k)
        assert pow(w, 1 << k, P) == 1 and pow(w, 1 << (k - 1), P) == P - 1


def test_lde_order_matches_direct_evaluation(oracle):
    # reference src/starks/prover.rs:837-882: lde[i] == p(h * w^i), w of order log2(N)
    from oracle_lib import fe
    rng = random.Random(3)
    n, blowup, h = 8, 4, 3
    coeffs = [rng.randrange(P) for _ in range(n)]
    arr = np.frombuffer(b"".join(fe(c) for c in coeffs), dtype=np.uint8).reshape(n, 32)
    ev = oracle.lde(arr, blowup, h)
    w = oracle.primitive_root(5)
    for i in range(n * blowup):
        x = h * pow(w, i, P) % P
        want = sum(c * pow(x, k, P) for k, c in enumerate(coeffs)) % P
        assert int.from_bytes(ev[i].tobytes(), "big") == want


def test_ntt_matches_naive_dft(oracle):
    from oracle_lib import fe
    rng = random.Random(4)
    for k in (0, 1, 4, 6):
        n = 1 << k
        x = [rng.randrange(P) for _ in range(n)]
        arr = np.frombuffer(b"".join(fe(c) for c in x), dtype=np.uint8).reshape(n, 32)
        w = oracle.primitive_root(k) if k else 1
        got = [int.from_bytes(r.tobytes(), "big") for r in oracle.ntt(arr)]
        assert got == [sum(x[j] * pow(w, i * j, P) for j in range(n)) % P for i in range(n)]
        back = oracle.ntt(oracle.ntt(arr), inverse=True)
        assert np.array_equal(back, arr)


def test_transcript_masks_and_reversal(oracle):
    # reference src/starks/transcript.rs:13-51 + lambdaworks DefaultTranscript
    t = oracle.Transcript()
    t.append(b"\x01\x02\x03")
    d = oracle.keccak256(b"\x01\x02\x03")
    c1 = t.challenge()
    assert c1 == d[::-1]
    d2 = oracle.keccak256(c1)[::-1]
    f = t.to_field()
    masked = bytes([d2[0] & 0x07]) + d2[1:]
    assert f == int.from_bytes(masked, "big") and f < 2**251
    d3 = oracle.keccak256(d2)[::-1]
    assert t.to_usize() == int.from_bytes(d3[:8], "big")


def test_merkle_tree_layout(oracle):
    from oracle_lib import fe
    vals = [5, 6, 7, 8]
    rows = np.frombuffer(b"".join(fe(v) for v in vals), dtype=np.uint8).reshape(4, 1, 32)
    root, nodes = oracle.merkle_build(rows, want_nodes=True)
    leaves = [oracle.keccak256(fe(v)) for v in vals]
    l01 = oracle.keccak256(leaves[0] + leaves[1])
    l23 = oracle.keccak256(leaves[2] + leaves[3])
    assert root == oracle.keccak256(l01 + l23)
    assert nodes[0].tobytes() == root and nodes[1].tobytes() == l01 and nodes[2].tobytes() == l23
    assert [nodes[3 + i].tobytes() for i in range(4)] == leaves
    # batched leaf: Keccak over the concatenated big-endian row
    rows2 = np.frombuffer(b"".join(fe(v) for v in vals), dtype=np.uint8).reshape(2, 2, 32)
    root2 = oracle.merkle_build(rows2)
    assert root2 == oracle.keccak256(oracle.keccak256(fe(5) + fe(6)) + oracle.keccak256(fe(7) + fe(8)))


def test_batch_inverse_and_zero(oracle):
    from oracle_lib import fe
    arr = np.frombuffer(b"".join(fe(v) for v in [2, 3, 5]), dtype=np.uint8).reshape(3, 32)
    out = oracle.batch_inverse(arr)
    assert [int.from_bytes(o.tobytes(), "big") for o in out] == [pow(v, P - 2, P) for v in [2, 3, 5]]
    with pytest.raises(ZeroDivisionError):
        oracle.batch_inverse(np.frombuffer(fe(0) + fe(1), dtype=np.uint8).reshape(2, 32))


def test_fri_fold_f293_known_answer():
    # reference src/starks/fri/fri_functions.rs:38-63: fold on F_293 (pure arithmetic restatement of fold_polynomial)
    def fold(p, beta, m=293):
        even, odd = p[0::2], [c * beta % m for c in p[1::2]]
        out = [0] * max(len(even), len(odd))
        for i, c in enumerate(even):
            out[i] = c
        for i, c in enumerate(odd):
            out[i] = (out[i] + c) % m
        return out
    p1 = fold([3, 1, 2, 7, 3, 5], 4)
    assert p1 == [7, 30, 23]
    p2 = fold(p1, 3)
    assert p2 == [97, 23]
    assert fold(p2, 2) == [143]

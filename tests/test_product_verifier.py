"""The verifier shipped in the product library (host C++, no GPU): accepts the reference's golden proof, agrees with the
oracle's verifier on accept/reject, and never crashes on malformed bytes (in the spirit of the reference's fuzz target,
fuzz/fuzz_targets/deserialize.rs, and of tests/integration_tests.rs:206-357)."""
import ctypes
import os
import random
import struct

from lambdaworks_cairo_prover_amd import api
from test_oracle_golden import parse_proof_file

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OPT = api.ProofOptions.default_test_options()


def test_accepts_reference_golden_proof(hip_lib):
    golden, pi = parse_proof_file(os.path.join(GOLDEN, "fibonacci_70000.proof"))
    words = [pi["public_memory"][a] for a in sorted(pi["public_memory"])]
    run = api.CairoRun.from_program(words)
    assert api.cairo_verify(golden, run.public_inputs_c, OPT)
    # the legacy-protocol fixtures must NOT verify under the current protocol (different boundary term)
    legacy, pi2 = parse_proof_file(os.path.join(GOLDEN, "fibonacci_500.proof"))
    run2 = api.CairoRun.from_program([pi2["public_memory"][a] for a in sorted(pi2["public_memory"])])
    assert not api.cairo_verify(legacy, run2.public_inputs_c, OPT)


def test_agrees_with_oracle_and_rejects_tampering(hip_lib, oracle):
    run = api.CairoRun.fibonacci(30)
    proof = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, (4, 3, 3, 1))
    assert api.cairo_verify(proof, run.public_inputs_c, OPT)
    rng = random.Random(5)
    flipped_rejected = 0
    for _ in range(60):
        bad = bytearray(proof)
        i = rng.randrange(len(bad))
        bad[i] ^= 1 << rng.randrange(8)
        got = api.cairo_verify(bytes(bad), run.public_inputs_c, OPT)
        want = oracle.cairo_verify(bytes(bad), run.public_inputs_c, (4, 3, 3, 1))
        assert got == want
        flipped_rejected += (not got)
    assert flipped_rejected == 60  # every single-bit flip is rejected (length prefixes and the frame's row width included)
    # wrong options / public inputs
    assert not api.cairo_verify(proof, run.public_inputs_c, api.ProofOptions(4, 4, 3, 1))   # more queries than the proof holds
    assert not api.cairo_verify(proof, run.public_inputs_c, api.ProofOptions(4, 3, 3, 30))  # grinding not satisfied
    pub2 = type(run.public_inputs_c)()
    ctypes.memmove(ctypes.byref(pub2), ctypes.byref(run.public_inputs_c), ctypes.sizeof(pub2))
    pub2.num_steps -= 1
    assert not api.cairo_verify(proof, pub2, OPT)


def test_malformed_inputs_do_not_crash(hip_lib, oracle):
    run = api.CairoRun.fibonacci(10)
    proof = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, (4, 3, 3, 1))
    rng = random.Random(9)
    for cut in (0, 1, 7, 8, 100, len(proof) - 1):
        assert not api.cairo_verify(proof[:cut] if cut else b"\x00", run.public_inputs_c, OPT)
    for _ in range(40):
        junk = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 400)))
        assert not api.cairo_verify(junk, run.public_inputs_c, OPT)
    huge_len = struct.pack(">Q", 2**63) + proof[8:]
    assert not api.cairo_verify(huge_len, run.public_inputs_c, OPT)


def test_proof_file_framing(hip_lib, oracle):
    run = api.CairoRun.fibonacci(10)
    proof = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, (4, 3, 3, 1))
    blob = api.proof_file_bytes(proof, run)
    got, pi = parse_proof_file_bytes(blob)
    assert got == proof
    assert pi["num_steps"] == run.num_steps and pi["rc"] == [run.public_inputs_c.range_check_min, run.public_inputs_c.range_check_max]
    assert [pi["public_memory"][a] for a in sorted(pi["public_memory"])] == [v for _, v in sorted(run.public_memory())]


def parse_proof_file_bytes(blob):
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".proof", delete=False) as f:
        f.write(blob)
        name = f.name
    try:
        return parse_proof_file(name)
    finally:
        os.unlink(name)


def test_every_single_bit_of_a_proof_matters(hip_lib, oracle):
    """Bits 0 and 7 of every byte of a valid proof, one at a time: the product's verifier and the oracle's reject each of the 2 x 21 664
    mutants - including the length prefixes (frame, element, query and opening lengths) that the reference's deserializer slices by
    (proof/stark.rs:225-440) and the out-of-domain frame's row width that its verifier takes the rows by (verifier.rs:136-137)."""
    run = api.CairoRun.fibonacci(10)
    options = (4, 3, 3, 1)
    proof = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
    accepted_by_product, accepted_by_oracle = [], []
    for pos in range(len(proof)):
        for bit in (0, 7):
            bad = bytearray(proof)
            bad[pos] ^= 1 << bit
            if api.cairo_verify(bytes(bad), run.public_inputs_c, OPT):
                accepted_by_product.append((pos, bit))
            if oracle.cairo_verify(bytes(bad), run.public_inputs_c, options):
                accepted_by_oracle.append((pos, bit))
    assert accepted_by_product == [] and accepted_by_oracle == []


def test_padding_the_reference_deserializer_tolerates_is_refused(hip_lib, oracle):
    """The reference parses every length-prefixed part inside the slice the prefix announces and reads the nonce from the last eight
    bytes (stark.rs:225-440): bytes appended INSIDE the last opening's slice (prefix enlarged to match) leave every parsed value and
    the nonce where they were - the oracle's verifier, which restates that parse, still accepts.  The product accepts the
    serializer's framing only."""
    run = api.CairoRun.fibonacci(10)
    options = (4, 3, 3, 1)
    proof = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
    assert oracle.cairo_verify(proof, run.public_inputs_c, options) and api.cairo_verify(proof, run.public_inputs_c, OPT)
    # find the last opening's length prefix: walk the openings from the count field (three queries -> three openings of equal size)
    nonce = proof[-8:]
    body = proof[:-8]
    n_open = 3
    # the three openings have equal size s: [.. count=3][s][opening][s][opening][s][opening]; solve for s from the tail
    for s in range(64, len(body)):
        if struct.unpack(">Q", body[-s - 8:-s])[0] == s and struct.unpack(">Q", body[-2 * (s + 8) - 0: -2 * (s + 8) + 8])[0] == s:
            break
    else:
        raise AssertionError("opening size not found")
    assert struct.unpack(">Q", body[-n_open * (s + 8) - 8:-n_open * (s + 8)])[0] == n_open
    pad = bytes(range(1, 25))
    padded = body[:-s - 8] + struct.pack(">Q", s + len(pad)) + body[-s:] + pad + nonce
    assert oracle.cairo_verify(padded, run.public_inputs_c, options)           # the reference's parse: padding inside a slice is ignored
    assert not api.cairo_verify(padded, run.public_inputs_c, OPT)              # the product: a prefix must equal its part
    assert api.last_error().startswith("non-canonical framing:")               # ... and says that this is why: not "rejected:"
    assert not api.cairo_verify(proof + b"\0", run.public_inputs_c, OPT) and api.last_error().startswith("non-canonical framing:")
    tampered = bytearray(proof); tampered[40] ^= 1                             # a bit of the first trace root: well-formed, invalid
    assert not api.cairo_verify(bytes(tampered), run.public_inputs_c, OPT) and api.last_error().startswith("rejected:")
    assert not api.cairo_verify(proof[:100], run.public_inputs_c, OPT) and api.last_error().startswith("malformed:")
    assert api.cairo_verify(proof, run.public_inputs_c, OPT) and api.last_error() == ""
    assert not oracle.cairo_verify(body + pad + nonce[:-1] + bytes([nonce[-1] ^ 1]), run.public_inputs_c, options)   # (bytes behind the openings move nothing either - but a changed nonce never passes)


def test_verdicts_agree_on_mixed_mutations(hip_lib, oracle):
    """Flips, truncations, appended bytes, zeroed and swapped 32-byte words on proofs of three programs and option sets: the product's
    verdict is the oracle's on every one of them (the oracle parses like the reference - part by part inside the announced slices,
    the nonce from the last eight bytes - so appended bytes change the nonce it reads and the proof falls with it)."""
    import cairo_asm as A
    cases = []
    for seed in range(3):
        rng = random.Random(seed)
        if seed == 0:
            run = api.CairoRun.fibonacci(10)
        else:
            words, entry = A.random_program(seed, 30)
            run = api.CairoRun.from_program(words, entry_pc=entry)
        options = [(4, 3, 3, 1), (2, 5, 3, 2), (8, 4, 3, 0)][seed]
        proof = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, options)
        assert oracle.cairo_verify(proof, run.public_inputs_c, options) and api.cairo_verify(proof, run.public_inputs_c, api.ProofOptions(*options))
        cases.append((run, options, proof))
    rejected = 0
    for i in range(450):
        rng = random.Random(1000 + i)
        run, options, proof = cases[i % 3]
        b = bytearray(proof)
        kind = rng.choice(["flip", "flip", "flip", "trunc", "extend", "zero", "swap"])
        if kind == "flip":
            for _ in range(rng.choice([1, 1, 2, 5])):
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif kind == "trunc":
            b = b[:rng.randrange(len(b))]
        elif kind == "extend":
            b += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 64)))
        elif kind == "zero":
            j = rng.randrange(len(b))
            b[j:j + 32] = bytes(min(32, len(b) - j))
        else:
            j, k = rng.randrange(len(b) - 32), rng.randrange(len(b) - 32)
            b[j:j + 32], b[k:k + 32] = b[k:k + 32], b[j:j + 32]
        want = oracle.cairo_verify(bytes(b), run.public_inputs_c, options)
        got = api.cairo_verify(bytes(b), run.public_inputs_c, api.ProofOptions(*options))
        assert got == want, (i, kind)
        rejected += not got
    assert rejected >= 440          # (a swap of two equal words or a zeroed all-zero word leaves the proof as it was)


def test_cli_verify_on_proof_files(hip_lib, oracle):
    """sp_proof_file_verify = the reference CLI's `verify` (src/main.rs:113-143): the file the reference itself wrote
    (benches/proofs/fibonacci_70000.proof) is accepted AS IT IS - proof and public inputs both come from its bytes, nothing is
    re-derived by running the program; the legacy-protocol fixtures are well-formed and rejected; files written by
    sp_proof_file_encode (rc-builtin and output segments included) round-trip; damage to the file is rejected, malformed or - inside
    slack the reference's parsers ignore - harmless, never a crash."""
    golden = open(os.path.join(GOLDEN, "fibonacci_70000.proof"), "rb").read()
    assert api.proof_file_verify(golden, OPT) and api.last_error() == ""
    for legacy in ("fibonacci_500.proof", "fibonacci_1000.proof"):
        assert not api.proof_file_verify(open(os.path.join(GOLDEN, legacy), "rb").read(), OPT) and api.last_error().startswith("rejected:")
    assert not api.proof_file_verify(golden, api.ProofOptions(8, 3, 3, 1))            # the verifier's options are not the prover's
    assert api.proof_file_verify(golden + b"\x00" * 5, OPT)                          # PublicInputs::deserialize stops behind num_steps (air.rs:428-450)
    # truncations: every cut inside the public inputs and a sample inside the proof
    plen = struct.unpack(">Q", golden[:8])[0]
    for cut in list(range(8 + plen, len(golden))) + [0, 7, 8, 9, 100, plen // 2, 8 + plen - 1]:
        assert not api.proof_file_verify(golden[:cut], OPT) and api.last_error().startswith(("malformed:", "non-canonical framing:", "rejected:")), cut
    # one flipped bit in the public inputs: rejected or malformed - every field is bound by a boundary constraint or by the permutation
    # product - with ONE exception the reference has too: fp_init appears in no boundary constraint (air.rs:777-849 pins pc and ap at both
    # ends, the permutation product and the two range-check bounds), so the verifier never looks at it
    rng = random.Random(5)
    fp_init = range(8 + plen + 8 + 64, 8 + plen + 8 + 96)
    # (and num_steps only enters as the exponent of the trace-domain generator, order 2^19 here: its bits from 19 up change nothing -
    # in the reference neither, boundary.rs evaluates g^step)
    for _ in range(300):
        pos, bit = rng.randrange(8 + plen, len(golden)), rng.randrange(8)
        bad = bytearray(golden)
        bad[pos] ^= 1 << bit
        ok = api.proof_file_verify(bytes(bad), OPT)
        steps_bit = 8 * (len(golden) - 1 - pos) + bit if pos >= len(golden) - 8 else None
        if pos in fp_init:
            assert ok or api.last_error().startswith("malformed:"), pos              # (a flip of its top bits can leave the field: >= p)
        elif steps_bit is not None and steps_bit >= 19:
            assert ok, (pos, bit)
        else:
            assert not ok, (pos, bit)
    unconstrained = bytearray(golden)
    unconstrained[fp_init[-1]] ^= 1
    assert api.proof_file_verify(bytes(unconstrained), OPT)
    # files of this library's own writer, builtin segments included
    from test_rc_builtin import run_of
    for run in (api.CairoRun.fibonacci(10), run_of("rc_program"), run_of("output_and_rc")):
        proof = oracle.cairo_prove(run.main_trace(), run.public_inputs_c, (4, 3, 3, 1))
        blob = api.proof_file_bytes(proof, run)
        assert api.proof_file_verify(blob, OPT)
        tampered = bytearray(blob)
        tampered[-1] ^= 1                                                            # num_steps
        assert not api.proof_file_verify(bytes(tampered), OPT)
    for _ in range(100):                                                             # junk never crashes
        junk = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 400)))
        assert not api.proof_file_verify(junk or b"\x00", OPT)

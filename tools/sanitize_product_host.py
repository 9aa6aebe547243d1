#!/usr/bin/env python3
"""Host-side code of the product library (Cairo front-end, verifier with both Merkle backends on valid, tampered, truncated and random inputs, host Poseidon, NUMA helper) under AddressSanitizer - no GPU needed (tools/sanitize_host.sh)."""
import sys, os, random, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from lambdaworks_cairo_prover_amd import _lib
_lib.LIB_PATH = os.environ.get('SP_SANITIZE_DIR', '/tmp/sp_sanitize') + '/libstark252_hip.so'
from lambdaworks_cairo_prover_amd import api
import oracle_lib as O
import poseidon_ref as pr
lib = _lib.load()
rnd = random.Random(2)
for n in (1, 2, 5, 34, 43):
    v = [rnd.randrange(pr.P) for _ in range(n)]
    assert api.poseidon_host(0, v) == pr.hash_many(v)
assert api.poseidon_host(3, [1, 2, 3]) == pr.hades([1, 2, 3])
# front-end: runs, traces, columns, builtins
for idx in (1, 10, 100, 1000):
    run = api.CairoRun.fibonacci(idx)
    t = run.main_trace(); t2 = run.main_trace(api.SP_FE_MONT_LIMBS); run.columns(); run.public_memory()
opts = (4, 3, 3, 1); OPT = api.ProofOptions(*opts)
run = api.CairoRun.fibonacci(20)
for backend in (0, 1):
    O.set_merkle_backend(backend)
    p = O.cairo_prove(run.main_trace(), run.public_inputs_c, opts)
    O.set_merkle_backend(0)
    assert api.cairo_verify(p, run.public_inputs_c, OPT, backend)
    assert not api.cairo_verify(p, run.public_inputs_c, OPT, 1 - backend)
    for _ in range(60):
        bad = bytearray(p); i = rnd.randrange(len(bad)); bad[i] ^= 1 << rnd.randrange(8)
        api.cairo_verify(bytes(bad), run.public_inputs_c, OPT, backend)
    for cut in (0, 1, 7, 8, 9, 100, 1000, len(p) - 1):
        api.cairo_verify(p[:cut] or b'\0', run.public_inputs_c, OPT, backend)
    for _ in range(40):
        junk = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 600)))
        api.cairo_verify(junk, run.public_inputs_c, OPT, backend)
blob = api.proof_file_bytes(p, run)
n = ctypes.c_int(); lib.sp_host_cpus(ctypes.byref(n)); api.host_bind_to_device(0)
# round 4: the inputs of the random tests - bit-flipped programs through the VM, damaged dumps and arrays through the readers, every
# length prefix of a proof through the verifier
import numpy as np
import cairo_asm as A
G = os.path.join(ROOT, 'tests', 'golden')
t0, m0 = open(G + '/program.trace', 'rb').read(), open(G + '/program.memory', 'rb').read()
for seed in range(400):
    r = random.Random(seed)
    words, entry = A.random_program(seed, 20)
    words = list(words)
    for _ in range(r.randrange(0, 4)):
        j = r.randrange(len(words)); words[j] = (words[j] ^ (1 << r.randrange(63))) % A.P
    try:
        rn = api.CairoRun.from_program(words, entry_pc=entry, max_steps=r.choice([16, 256, 4096])); rn.main_trace(); regs, addrs, vals = rn.export()
    except api.SpError:
        continue
    if r.random() < 0.5:
        addrs = addrs.copy(); addrs[r.randrange(len(addrs))] ^= np.uint64(1 << r.randrange(12))
    try: api.CairoRun.from_arrays(regs, addrs, vals, len(words)).main_trace()
    except api.SpError: pass
for seed in range(300):
    r = random.Random(seed); t, m = bytearray(t0), bytearray(m0)
    k = r.choice(['t', 'm', 'tt', 'mt'])
    if k == 't': t[r.randrange(len(t))] ^= 1 << r.randrange(8)
    elif k == 'm': m[r.randrange(len(m))] ^= 1 << r.randrange(8)
    elif k == 'tt': t = t[:r.randrange(len(t))]
    else: m = m[:r.randrange(len(m))]
    try: api.CairoRun.from_dumps(bytes(t), bytes(m), program_size=r.choice([1, 5, len(m0) // 40])).main_trace()
    except api.SpError: pass
O.set_merkle_backend(0)
p = O.cairo_prove(run.main_trace(), run.public_inputs_c, opts)
for pos in range(0, len(p), 3):
    bad = bytearray(p); bad[pos] ^= 1 << (pos % 8)
    assert not api.cairo_verify(bytes(bad), run.public_inputs_c, OPT, 0)
print("product host code under ASan: ok")
# round 5: the reference unit-test vectors through the host builder (crafted runs of tests/test_reference_unit_kats.py), duplicate and
# conflicting memory cells through sp_cairo_run_from_arrays (flat-array and sparse paths), the verifier's refusal reasons
import test_reference_unit_kats as K
run, v, values = K.rc_decompose_run(); K.check_rc_decompose(run.main_trace(), v, values)
run, v = K.rc_holes_run(); K.check_rc_holes(run, run.main_trace(), v)
run, v = K.missing_offsets_run(); K.check_missing_offsets(run.main_trace(), v)
for name in ("no_codelen", "inside_program_section", "outside_program_section"):
    run, v = K.memory_holes_run(name); K.check_memory_holes(run, run.main_trace(), v)
run, v = K.fill_memory_holes_run(); K.check_fill_memory_holes(run, run.main_trace(), v)
import test_dump_reader_fuzz as F
F.test_conflicting_duplicate_address_is_an_error()
import test_product_verifier as V
V.test_padding_the_reference_deserializer_tolerates_is_refused(lib, O)
V.test_cli_verify_on_proof_files(lib, O)
print("budget", api.host_cpu_budget(), "rule", api.model_shard_interpolation(70.0, 8, 20))
print("sanitize_product_host round-5 additions ok")


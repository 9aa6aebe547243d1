"""ctypes binding of the CPU oracle (oracle/liboracle_stark252.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle_stark252.so")

P = 2**251 + 17 * 2**192 + 1


class PublicInputsC(ctypes.Structure):
    _fields_ = [("pc_init", ctypes.c_uint8 * 32), ("ap_init", ctypes.c_uint8 * 32), ("fp_init", ctypes.c_uint8 * 32),
                ("pc_final", ctypes.c_uint8 * 32), ("ap_final", ctypes.c_uint8 * 32),
                ("range_check_min", ctypes.c_uint16), ("range_check_max", ctypes.c_uint16),
                ("n_segments", ctypes.c_uint32), ("segment_types", ctypes.c_void_p), ("segment_ranges", ctypes.c_void_p),
                ("n_public_memory", ctypes.c_uint64), ("public_memory", ctypes.c_void_p), ("num_steps", ctypes.c_uint64)]


class ProofOptionsC(ctypes.Structure):
    _fields_ = [("blowup_factor", ctypes.c_uint8), ("fri_number_of_queries", ctypes.c_uint64),
                ("coset_offset", ctypes.c_uint64), ("grinding_factor", ctypes.c_uint8)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def usable_cpus():
    """CPUs this process can really have: the affinity mask cut down by the cgroup's CPU quota (the GPU boxes give a container
    16 CPUs' worth of time on a 256-thread host; an OpenMP team of 256 spinning threads there makes a 128-row proof take 5 s)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def load():
    global _lib
    if _lib is None:
        # the oracle's OpenMP team: no larger than the CPUs there are, and sleeping rather than spinning between parallel regions
        os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        if not os.path.exists(LIB_PATH):
            build()
        try:
            _lib = ctypes.CDLL(LIB_PATH)
        except OSError:
            build()
            _lib = ctypes.CDLL(LIB_PATH)
        if "SP_ORACLE_THREADS_SET" not in os.environ:          # (libgomp may have been loaded - by torch - before the defaults above were set)
            try:
                ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "0")) or usable_cpus())
            except (OSError, ValueError):
                pass
        _lib.oracle_grinding_nonce.restype = ctypes.c_uint64
        _lib.oracle_transcript_new.restype = ctypes.c_void_p
        _lib.oracle_transcript_to_usize.restype = ctypes.c_uint64
    return _lib


def _u8p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def fe(x):
    return int(x).to_bytes(32, "big")


def keccak256(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    load().oracle_keccak256(data, ctypes.c_uint64(len(data)), out)
    return out.raw


def fe_op(op, a, b=0):
    out = ctypes.create_string_buffer(32)
    rc = load().oracle_fe_op(op, fe(a), fe(b), out)
    if rc != 0:
        raise ZeroDivisionError("oracle_fe_op failed")
    return int.from_bytes(out.raw, "big")


def primitive_root(order):
    out = ctypes.create_string_buffer(32)
    load().oracle_primitive_root(order, out)
    return int.from_bytes(out.raw, "big")


def ntt(arr, inverse=False, coset=None):
    a = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1, 32).copy()
    rc = load().oracle_ntt(_u8p(a), ctypes.c_uint64(a.shape[0]), int(bool(inverse)), None if coset is None else fe(coset))
    assert rc == 0
    return a


def lde(coeffs, blowup, coset):
    a = np.ascontiguousarray(coeffs, dtype=np.uint8).reshape(-1, 32)
    out = np.empty((a.shape[0] * blowup, 32), dtype=np.uint8)
    rc = load().oracle_lde(_u8p(a), ctypes.c_uint64(a.shape[0]), ctypes.c_uint32(blowup), fe(coset), _u8p(out))
    assert rc == 0
    return out


def merkle_build(rows, want_nodes=False):
    a = np.ascontiguousarray(rows, dtype=np.uint8)
    n, w = a.shape[0], a.shape[1]
    root = ctypes.create_string_buffer(32)
    nodes = np.empty((2 * n - 1, 32), dtype=np.uint8) if want_nodes else None
    rc = load().oracle_merkle_build(_u8p(a), ctypes.c_uint64(n), ctypes.c_uint32(w), root, _u8p(nodes) if want_nodes else None)
    assert rc == 0
    return (root.raw, nodes) if want_nodes else root.raw


def set_merkle_backend(backend):
    """0 = Keccak256 trees (the reference), 1 = Starknet Poseidon trees (oracle/poseidon.hpp); process-wide, reset it after use."""
    assert load().oracle_set_merkle_backend(int(backend)) == 0


def poseidon(mode, values):
    """mode 0 hash_many, 1 hash(x, y), 2 hash_single(x), 3 the Hades permutation (three outputs); canonical integers."""
    buf = b"".join(fe(v) for v in values)
    out = ctypes.create_string_buffer(96)
    assert load().oracle_poseidon(int(mode), buf, ctypes.c_uint64(len(values)), out) == 0
    r = [int.from_bytes(out.raw[32 * k:32 * k + 32], "big") for k in range(3)]
    return r if mode == 3 else r[0]


def batch_inverse(arr):
    a = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1, 32).copy()
    rc = load().oracle_batch_inverse(_u8p(a), ctypes.c_uint64(a.shape[0]))
    if rc != 0:
        raise ZeroDivisionError("batch inverse of zero")
    return a


class Transcript:
    def __init__(self):
        self._l = load()
        self._h = ctypes.c_void_p(self._l.oracle_transcript_new())

    def append(self, b: bytes):
        self._l.oracle_transcript_append(self._h, b, ctypes.c_uint64(len(b)))

    def challenge(self) -> bytes:
        out = ctypes.create_string_buffer(32)
        self._l.oracle_transcript_challenge(self._h, out)
        return out.raw

    def to_field(self) -> int:
        out = ctypes.create_string_buffer(32)
        self._l.oracle_transcript_to_field(self._h, out)
        return int.from_bytes(out.raw, "big")

    def to_usize(self) -> int:
        return self._l.oracle_transcript_to_usize(self._h)

    def __del__(self):
        try:
            self._l.oracle_transcript_free(self._h)
        except Exception:
            pass


def grinding_nonce(challenge: bytes, factor: int) -> int:
    return load().oracle_grinding_nonce(challenge, ctypes.c_uint8(factor))


def make_public_inputs(pc_init, ap_init, fp_init, pc_final, ap_final, rc_min, rc_max, public_memory, num_steps, segments=()):
    """public_memory: list of (address, value) ints. Returns (struct, keepalive)."""
    pi = PublicInputsC()
    for name, v in (("pc_init", pc_init), ("ap_init", ap_init), ("fp_init", fp_init), ("pc_final", pc_final), ("ap_final", ap_final)):
        ctypes.memmove(getattr(pi, name), fe(v), 32)
    pi.range_check_min, pi.range_check_max = rc_min, rc_max
    types = (ctypes.c_uint8 * max(1, len(segments)))(*[s[0] for s in segments])
    ranges = (ctypes.c_uint64 * max(1, 2 * len(segments)))(*[x for s in segments for x in s[1:3]])
    pm = ctypes.create_string_buffer(b"".join(fe(a) + fe(v) for a, v in public_memory), 64 * len(public_memory) or 1)
    pi.n_segments = len(segments)
    pi.segment_types = ctypes.cast(types, ctypes.c_void_p)
    pi.segment_ranges = ctypes.cast(ranges, ctypes.c_void_p)
    pi.n_public_memory = len(public_memory)
    pi.public_memory = ctypes.cast(pm, ctypes.c_void_p)
    pi.num_steps = num_steps
    return pi, (types, ranges, pm)


def cairo_prove(main_trace, pub, options, legacy_boundary=False, want_timings=False):
    """main_trace: (n, cols, 32) uint8; pub: PublicInputsC (oracle or product struct — same layout)."""
    a = np.ascontiguousarray(main_trace, dtype=np.uint8)
    n, cols = a.shape[0], a.shape[1]
    opt = ProofOptionsC(*options)
    out = ctypes.POINTER(ctypes.c_uint8)()
    ln = ctypes.c_uint64()
    tm = (ctypes.c_double * 4)()
    rc = load().oracle_cairo_prove(_u8p(a), ctypes.c_uint64(n), ctypes.c_uint32(cols), ctypes.byref(pub), ctypes.byref(opt),
                                   int(legacy_boundary), ctypes.byref(out), ctypes.byref(ln), tm)
    if rc != 0:
        raise RuntimeError(f"oracle_cairo_prove failed: {rc}")
    proof = ctypes.string_at(out, ln.value)
    load().oracle_free(out)
    return (proof, list(tm)) if want_timings else proof


def cairo_verify(proof: bytes, pub, options) -> bool:
    opt = ProofOptionsC(*options)
    rc = load().oracle_cairo_verify(proof, ctypes.c_uint64(len(proof)), ctypes.byref(pub), ctypes.byref(opt))
    return rc == 1


def cairo_aux_trace(main_trace, pub, rap):
    a = np.ascontiguousarray(main_trace, dtype=np.uint8)
    n, cols = a.shape[0], a.shape[1]
    out = np.empty((n, 18, 32), dtype=np.uint8)
    rc = load().oracle_cairo_aux_trace(_u8p(a), ctypes.c_uint64(n), ctypes.c_uint32(cols), ctypes.byref(pub), b"".join(fe(x) for x in rap), _u8p(out))
    assert rc == 0
    return out


def cairo_transition(frame, rap, has_rc_builtin=False):
    a = np.ascontiguousarray(frame, dtype=np.uint8)
    cols = a.shape[1]
    ncons = 50 if has_rc_builtin else 49
    out = np.empty((ncons, 32), dtype=np.uint8)
    rc = load().oracle_cairo_transition(_u8p(a), ctypes.c_uint32(cols), int(has_rc_builtin), b"".join(fe(x) for x in rap), _u8p(out))
    assert rc == 0
    return out


# ---- example AIRs (reference src/starks/example) and the program AIR ----------------------------------------------------
EXAMPLE_KINDS = {"simple_fibonacci": 0, "fibonacci_2_columns": 1, "quadratic": 2, "fibonacci_rap": 3, "dummy": 4}


def _params(values):
    return b"".join(fe(v) for v in (list(values) + [1, 1])[:2])


def example_trace(kind, length, params=(1, 1)):
    """The reference's trace generator for the example; (rows, cols, 32) uint8. `length` = trace length (steps for fibonacci_rap)."""
    l = load()
    l.oracle_example_trace.restype = ctypes.c_uint64
    cols = ctypes.c_uint32()
    n = l.oracle_example_trace(EXAMPLE_KINDS[kind], _params(params), ctypes.c_uint64(length), None, ctypes.byref(cols))
    assert n > 0
    out = np.empty((n, cols.value, 32), dtype=np.uint8)
    assert l.oracle_example_trace(EXAMPLE_KINDS[kind], _params(params), ctypes.c_uint64(length), _u8p(out), ctypes.byref(cols)) == n
    return out


def _take_proof(out, ln):
    proof = ctypes.string_at(out, ln.value)
    load().oracle_free(out)
    return proof


def example_prove(kind, trace, options, params=(1, 1), steps=0):
    a = np.ascontiguousarray(trace, dtype=np.uint8)
    opt = ProofOptionsC(*options)
    out, ln = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_uint64()
    rc = load().oracle_example_prove(EXAMPLE_KINDS[kind], _params(params), ctypes.c_uint64(steps), _u8p(a), ctypes.c_uint64(a.shape[0]),
                                     ctypes.c_uint32(a.shape[1]), ctypes.byref(opt), ctypes.byref(out), ctypes.byref(ln))
    if rc != 0:
        raise RuntimeError(f"oracle_example_prove failed: {rc}")
    return _take_proof(out, ln)


def example_verify(kind, proof, options, params=(1, 1), steps=0):
    opt = ProofOptionsC(*options)
    return load().oracle_example_verify(EXAMPLE_KINDS[kind], _params(params), ctypes.c_uint64(steps), proof, ctypes.c_uint64(len(proof)), ctypes.byref(opt)) == 1


def program_air_prove(desc, trace, options):
    """desc: lambdaworks_cairo_prover_amd.air.AirDescC (same layout as oracle_air_desc)."""
    a = np.ascontiguousarray(trace, dtype=np.uint8)
    opt = ProofOptionsC(*options)
    out, ln = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_uint64()
    rc = load().oracle_program_air_prove(ctypes.byref(desc), _u8p(a), ctypes.c_uint64(a.shape[0]), ctypes.byref(opt), ctypes.byref(out), ctypes.byref(ln))
    if rc != 0:
        raise RuntimeError(f"oracle_program_air_prove failed: {rc}")
    return _take_proof(out, ln)


def program_air_verify(desc, proof, options):
    opt = ProofOptionsC(*options)
    return load().oracle_program_air_verify(ctypes.byref(desc), proof, ctypes.c_uint64(len(proof)), ctypes.byref(opt)) == 1

"""Host-side mirror of the reference interface for the proving hot path, on top of the C ABI.

Names follow the reference: `ProofOptions` (src/starks/proof/options.rs:21-26), `PublicInputs` (src/cairo/air.rs:163-181),
`generate_prover_args` (src/cairo/runner/run.rs:242-263), `generate_cairo_proof` (src/cairo/air.rs:1165-1171), and the
lambdaworks seam `interpolate_fft / evaluate_offset_fft / MerkleTree::build / inplace_batch_inverse`.
Field elements cross this layer as canonical 32-byte big-endian records in numpy uint8 arrays of shape (..., 32).
"""
import ctypes
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import (SP_FE_CANON_BE, SP_FE_MONT_LIMBS, CairoPublicInputsC, ConfigC, ProofOptionsC, SpError, check)

__all__ = ["ProofOptions", "Context", "CairoRun", "generate_prover_args_fibonacci", "SpError", "P", "felts_to_bytes",
           "bytes_to_felts", "SP_FE_CANON_BE", "SP_FE_MONT_LIMBS"]

P = 2**251 + 17 * 2**192 + 1


def felts_to_bytes(values):
    """list of python ints (< p) -> uint8 array (len, 32), canonical big-endian."""
    return np.frombuffer(b"".join(int(v).to_bytes(32, "big") for v in values), dtype=np.uint8).reshape(-1, 32).copy()


def bytes_to_felts(arr):
    b = np.ascontiguousarray(arr, dtype=np.uint8).tobytes()
    return [int.from_bytes(b[i:i + 32], "big") for i in range(0, len(b), 32)]


@dataclass
class ProofOptions:
    """reference src/starks/proof/options.rs:21-26"""
    blowup_factor: int = 4
    fri_number_of_queries: int = 3
    coset_offset: int = 3
    grinding_factor: int = 1

    @staticmethod
    def default_test_options():  # options.rs:144-151
        return ProofOptions(4, 3, 3, 1)

    # SecurityLevel (options.rs:5-12)
    CONJECTURABLE_80, CONJECTURABLE_100, CONJECTURABLE_128, PROVABLE_80, PROVABLE_100, PROVABLE_128 = range(6)

    @staticmethod
    def _from_c(c):
        return ProofOptions(int(c.blowup_factor), int(c.fri_number_of_queries), int(c.coset_offset), int(c.grinding_factor))

    @staticmethod
    def new_secure(security_level, coset_offset):  # options.rs:35-75 (sp_proof_options_new_secure)
        out = ProofOptionsC()
        check(_lib.load().sp_proof_options_new_secure(int(security_level), ctypes.c_uint64(coset_offset), ctypes.byref(out)))
        return ProofOptions._from_c(out)

    @staticmethod
    def new_with_checked_security(blowup_factor, fri_number_of_queries, coset_offset, grinding_factor, security_target, field_bits=252, provable=False):
        """options.rs:78-102 (provable=True: new_with_checked_provable_security, :107-129); raises SpError whose message is the
        reference's InsecureOptionError variant.  field_bits: F::field_bit_size() (252 for Stark252)."""
        out = ProofOptionsC()
        check(_lib.load().sp_proof_options_checked(ctypes.c_uint8(blowup_factor), ctypes.c_uint64(fri_number_of_queries), ctypes.c_uint64(coset_offset),
                                                   ctypes.c_uint8(grinding_factor), ctypes.c_uint8(security_target), int(bool(provable)), ctypes.c_uint32(field_bits), ctypes.byref(out)))
        return ProofOptions._from_c(out)

    def to_c(self):
        return ProofOptionsC(self.blowup_factor, self.fri_number_of_queries, self.coset_offset, self.grinding_factor)


def _u8p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


class Context:
    """One device context (sp_ctx): owns a HIP stream, twiddle tables and, during a proof, all device buffers."""

    def __init__(self, device=0, fe_encoding=SP_FE_CANON_BE):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        cfg = ConfigC(device, fe_encoding)
        check(self._lib.sp_ctx_create(ctypes.byref(self._h), ctypes.byref(cfg)))
        self.fe_encoding = fe_encoding

    def close(self):
        if self._h:
            self._lib.sp_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- lambdaworks seam -------------------------------------------------------------------------------
    def ntt(self, data, inverse=False, coset=None):
        """evaluate_fft / interpolate_fft / evaluate_offset_fft / interpolate_offset_fft on a (n, 32) array."""
        a = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, 32).copy()
        c = None if coset is None else _u8p(np.ascontiguousarray(coset, dtype=np.uint8))
        check(self._lib.sp_ntt(self._h, _u8p(a), ctypes.c_uint64(a.shape[0]), int(bool(inverse)), c))
        return a

    def lde(self, coeffs, blowup, coset):
        """evaluate_polynomial_on_lde_domain for (cols, n, 32) coefficient columns -> (cols, n*blowup, 32)."""
        a = np.ascontiguousarray(coeffs, dtype=np.uint8)
        cols, n = a.shape[0], a.shape[1]
        out = np.empty((cols, n * blowup, 32), dtype=np.uint8)
        cs = np.ascontiguousarray(coset, dtype=np.uint8)
        check(self._lib.sp_lde(self._h, _u8p(a), ctypes.c_uint64(n), ctypes.c_uint32(cols), ctypes.c_uint32(blowup), _u8p(cs), _u8p(out)))
        return out

    def merkle_build(self, rows, want_nodes=False):
        """MerkleTree::build over (n_leaves, fe_per_leaf, 32) rows. Returns root bytes (and the (2n-1, 32) node array)."""
        a = np.ascontiguousarray(rows, dtype=np.uint8)
        n, w = a.shape[0], a.shape[1]
        root = np.empty(32, dtype=np.uint8)
        nodes = np.empty((2 * n - 1, 32), dtype=np.uint8) if want_nodes else None
        check(self._lib.sp_merkle_build(self._h, _u8p(a), ctypes.c_uint64(n), ctypes.c_uint32(w), _u8p(root),
                                        _u8p(nodes) if want_nodes else None))
        return (root.tobytes(), nodes) if want_nodes else root.tobytes()

    def merkle_build_dev(self, cols_ptr, n_leaves, fe_per_leaf, col_stride, nodes_ptr):
        """sp_merkle_build_dev: column-major device-layout columns -> (2n-1) x 32-byte nodes on the device (asynchronous)."""
        check(self._lib.sp_merkle_build_dev(self._h, ctypes.c_void_p(cols_ptr), ctypes.c_uint64(n_leaves), ctypes.c_uint32(fe_per_leaf),
                                            ctypes.c_uint64(col_stride), ctypes.c_void_p(nodes_ptr)))

    def fe_mul(self, a, b=None):
        """sp_fe_mul: element-wise products (b None: squares) by the device's own Montgomery routines, in the context's encoding."""
        x = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32)
        out = np.empty_like(x)
        y = None if b is None else np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 32)
        assert y is None or y.shape == x.shape
        check(self._lib.sp_fe_mul(self._h, _u8p(x), None if y is None else _u8p(y), ctypes.c_uint64(x.shape[0]), _u8p(out)))
        return out

    def batch_inverse(self, data):
        a = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, 32).copy()
        check(self._lib.sp_batch_inverse(self._h, _u8p(a), ctypes.c_uint64(a.shape[0])))
        return a

    # ---- whole proof -----------------------------------------------------------------------------------
    def cairo_prove(self, main_trace, public_inputs_c, options):
        """generate_cairo_proof (reference src/cairo/air.rs:1165-1171) + serialize: returns the proof bytes.
        main_trace: (n, cols, 32) uint8 in the context encoding; public_inputs_c: CairoPublicInputsC."""
        a = np.ascontiguousarray(main_trace, dtype=np.uint8)
        n, cols = a.shape[0], a.shape[1]
        opt = options.to_c()
        out = ctypes.POINTER(ctypes.c_uint8)()
        ln = ctypes.c_uint64()
        check(self._lib.sp_cairo_prove(self._h, _u8p(a), ctypes.c_uint64(n), ctypes.c_uint32(cols), ctypes.byref(public_inputs_c),
                                       ctypes.byref(opt), ctypes.byref(out), ctypes.byref(ln)))
        proof = ctypes.string_at(out, ln.value)
        self._lib.sp_free(out)
        return proof

    def cairo_prove_dev(self, trace_dev_ptr, n, cols, public_inputs_c, options):
        """Same as cairo_prove with the (n, cols, 32)-byte main trace already in device memory at `trace_dev_ptr`."""
        opt = options.to_c()
        out = ctypes.POINTER(ctypes.c_uint8)()
        ln = ctypes.c_uint64()
        check(self._lib.sp_cairo_prove_dev(self._h, ctypes.c_void_p(trace_dev_ptr), ctypes.c_uint64(n), ctypes.c_uint32(cols),
                                           ctypes.byref(public_inputs_c), ctypes.byref(opt), ctypes.byref(out), ctypes.byref(ln)))
        proof = ctypes.string_at(out, ln.value)
        self._lib.sp_free(out)
        return proof

    def _take_proof(self, out, ln):
        proof = ctypes.string_at(out, ln.value)
        self._lib.sp_free(out)
        return proof

    def cairo_prove_run(self, run, options):
        """sp_cairo_prove_run: the run's own column-major (page-locked) main trace goes up by DMA - no host-side gather."""
        opt = options.to_c()
        out = ctypes.POINTER(ctypes.c_uint8)()
        ln = ctypes.c_uint64()
        check(self._lib.sp_cairo_prove_run(self._h, run._h, ctypes.byref(opt), ctypes.byref(out), ctypes.byref(ln)))
        return self._take_proof(out, ln)

    def cairo_prove_columns(self, cols_ptr, n, cols, public_inputs_c, options, col_stride=0, device_layout=False):
        """sp_cairo_prove_columns: host COLUMNS at `cols_ptr` (int address or (cols, n, 32) uint8 array)."""
        keep = None
        if isinstance(cols_ptr, np.ndarray):
            keep = np.ascontiguousarray(cols_ptr, dtype=np.uint8)
            cols_ptr = keep.ctypes.data
        opt = options.to_c()
        out = ctypes.POINTER(ctypes.c_uint8)()
        ln = ctypes.c_uint64()
        check(self._lib.sp_cairo_prove_columns(self._h, ctypes.c_void_p(cols_ptr), ctypes.c_uint64(n), ctypes.c_uint32(cols), ctypes.c_uint64(col_stride),
                                               int(bool(device_layout)), ctypes.byref(public_inputs_c), ctypes.byref(opt), ctypes.byref(out), ctypes.byref(ln)))
        del keep
        return self._take_proof(out, ln)

    def prewarm(self, n_rows, main_cols=34, aux_cols=18, has_rc_builtin=False, options=None, flags=0):
        """sp_prewarm: everything a first proof of this shape would otherwise pay on its critical path (arena, tables, streams and
        page-locked slots, the first launch of every kernel family, the device's clocks) - callable on a thread of its own while the
        Cairo VM is still running (ctypes releases the GIL for the duration).  flags: SP_PREWARM_* (0 = all)."""
        opt = (options or ProofOptions()).to_c()
        check(self._lib.sp_prewarm(self._h, ctypes.c_uint64(n_rows), ctypes.c_uint32(main_cols), ctypes.c_uint32(aux_cols), int(bool(has_rc_builtin)),
                                   ctypes.byref(opt), ctypes.c_uint32(flags)))

    def prewarm_cancel(self):
        """sp_prewarm_cancel: from any thread - the prewarm running (or about to run) on this context skips the rest of its clock ramp."""
        check(self._lib.sp_prewarm_cancel(self._h))

    def last_upload_stats(self):
        """sp_last_upload_stats of the last proof's main-trace upload."""
        v = (ctypes.c_double * 10)()
        check(self._lib.sp_last_upload_stats(self._h, v))
        kinds = {0: "single copy / resident", 1: "row-major host buffer, gathered by host threads", 2: "host columns, DMA from page-locked memory",
                 3: "host columns in PAGEABLE memory (staged by the runtime)", 4: "run image (register states + memory, page-locked), trace built on the device",
                 5: "run image in PAGEABLE memory, trace built on the device"}
        return {"kind": kinds.get(int(v[0]), "?"), "groups": int(v[1]), "bytes": int(v[2]), "gather_ms": round(v[3], 3), "gather_gbs": round(v[4], 1),
                "dma_ms": round(v[5], 3), "dma_gbs": round(v[6], 1), "exposed_ms": round(v[7], 3), "max_stall_ms": round(v[8], 3), "host_ms": round(v[9], 3)}

    def air_prove(self, desc, main_trace, options):
        """sp_air_prove: desc = lambdaworks_cairo_prover_amd.air.AirDescC (AirBuilder.build()[0]); main_trace (n, cols, 32)."""
        a = np.ascontiguousarray(main_trace, dtype=np.uint8)
        n, cols = a.shape[0], a.shape[1]
        assert cols == desc.main_cols
        opt = options.to_c()
        out = ctypes.POINTER(ctypes.c_uint8)()
        ln = ctypes.c_uint64()
        check(self._lib.sp_air_prove(self._h, ctypes.byref(desc), _u8p(a), ctypes.c_uint64(n), ctypes.byref(opt), ctypes.byref(out), ctypes.byref(ln)))
        proof = ctypes.string_at(out, ln.value)
        self._lib.sp_free(out)
        return proof

    def last_proof_info(self):
        """{'composition_path': 1 (2n points) | 2 (whole domain) | 3 (whole domain, deg H >= 2n), 'fri_sharded_layers', 'groups', ...}"""
        out = (ctypes.c_uint32 * 4)()
        check(self._lib.sp_last_proof_info(self._h, out))
        return {"composition_path": out[0], "fri_sharded_layers": out[1], "groups": out[2], "interpolation_sharded": out[3]}

    def prover_device_bytes(self):
        out = ctypes.c_uint64()
        check(self._lib.sp_prover_device_bytes(self._h, ctypes.byref(out)))
        return out.value

    def last_round_ms(self):
        ms = (ctypes.c_float * 5)()
        check(self._lib.sp_last_round_ms(self._h, ms))
        return list(ms)

    # ---- device-resident variants used by bench.py -------------------------------------------------------
    def ntt_dev(self, data_ptr, n, batch=1, inverse=False):
        check(self._lib.sp_ntt_dev(self._h, ctypes.c_void_p(data_ptr), ctypes.c_uint64(n), ctypes.c_uint32(batch), int(bool(inverse)), None))

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        check(self._lib.sp_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def timer_start(self):
        check(self._lib.sp_timer_start(self._h))

    def timer_stop(self):
        ms = ctypes.c_float()
        check(self._lib.sp_timer_stop(self._h, ctypes.byref(ms)))
        return ms.value

    def sync(self):
        check(self._lib.sp_sync(self._h))


def host_cpus():
    """sp_host_cpus: hardware threads cut down by the affinity mask and the cgroup CPU quota."""
    n = ctypes.c_int()
    check(_lib.load().sp_host_cpus(ctypes.byref(n)))
    return n.value


def host_cpu_budget():
    """sp_host_cpu_budget: (CPUs of this process' share, ranks sharing the host) - SP_OPT_HOST_RANKS / SP_HOST_RANKS / LOCAL_WORLD_SIZE."""
    b, r = ctypes.c_int(), ctypes.c_int()
    check(_lib.load().sp_host_cpu_budget(ctypes.byref(b), ctypes.byref(r)))
    return b.value, r.value


def host_bind_to_device(device=0):
    """sp_host_bind_to_device: keep this thread (and the threads it creates from now on) on the CPUs of the GPU's NUMA node; returns
    the node or -1 when unknown (nothing changed)."""
    node = ctypes.c_int(-1)
    check(_lib.load().sp_host_bind_to_device(int(device), ctypes.byref(node)))
    return node.value


def fe_to_device(values_be, fe_encoding=SP_FE_CANON_BE):
    """ABI-encoded (n, 32) array -> device layout bytes (8 x u32 little-endian Montgomery), on the host."""
    a = np.ascontiguousarray(values_be, dtype=np.uint8).reshape(-1, 32)
    out = np.empty_like(a)
    check(_lib.load().sp_fe_to_device(fe_encoding, _u8p(a), ctypes.c_uint64(a.shape[0]), _u8p(out)))
    return out


def fe_from_device(dev_bytes, fe_encoding=SP_FE_CANON_BE):
    a = np.ascontiguousarray(dev_bytes, dtype=np.uint8).reshape(-1, 32)
    out = np.empty_like(a)
    check(_lib.load().sp_fe_from_device(fe_encoding, _u8p(a), ctypes.c_uint64(a.shape[0]), _u8p(out)))
    return out


class CairoRun:
    """Register trace + memory + PublicInputs + main trace of one Cairo execution (host side).

    Mirrors `generate_prover_args` (reference src/cairo/runner/run.rs:242-263): run -> PublicInputs::from_regs_and_mem
    -> build_main_trace.  Only hint-free, builtin-free programs run in the built-in VM; dumps of cairo-run are also accepted.
    """

    def __init__(self, handle):
        self._lib = _lib.load()
        self._h = handle
        n, c, s = ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_uint64()
        check(self._lib.sp_cairo_run_shape(self._h, ctypes.byref(n), ctypes.byref(c), ctypes.byref(s)))
        self.n_rows, self.n_cols, self.num_steps = n.value, c.value, s.value
        self.public_inputs_c = CairoPublicInputsC()
        check(self._lib.sp_cairo_run_public_inputs(self._h, ctypes.byref(self.public_inputs_c)))

    @staticmethod
    def fibonacci(fib_index):
        lib = _lib.load()
        h = ctypes.c_void_p()
        check(lib.sp_cairo_run_fibonacci(ctypes.c_uint64(fib_index), ctypes.byref(h)))
        return CairoRun(h)

    @staticmethod
    def from_program(words, max_steps=1 << 26, entry_pc=1):
        lib = _lib.load()
        h = ctypes.c_void_p()
        a = felts_to_bytes(words)
        check(lib.sp_cairo_run_program_at(_u8p(a), ctypes.c_uint64(a.shape[0]), ctypes.c_uint64(entry_pc),
                                          ctypes.c_uint64(max_steps), ctypes.byref(h)))
        return CairoRun(h)

    @staticmethod
    def from_program_builtins(words, output=False, range_check=True, max_steps=1 << 26, entry_pc=1):
        """A hint-free program that declares the output and / or range_check builtins (sp_cairo_run_program_builtins)."""
        lib = _lib.load()
        h = ctypes.c_void_p()
        a = felts_to_bytes(words)
        mask = (1 if output else 0) | (2 if range_check else 0)
        check(lib.sp_cairo_run_program_builtins(_u8p(a), ctypes.c_uint64(a.shape[0]), ctypes.c_uint64(entry_pc),
                                                ctypes.c_uint64(max_steps), ctypes.c_uint32(mask), ctypes.byref(h)))
        return CairoRun(h)

    @staticmethod
    def from_dumps(trace_bytes, memory_bytes, program_size):
        lib = _lib.load()
        h = ctypes.c_void_p()
        check(lib.sp_cairo_run_from_dumps(trace_bytes, ctypes.c_uint64(len(trace_bytes)), memory_bytes,
                                          ctypes.c_uint64(len(memory_bytes)), ctypes.c_uint64(program_size), ctypes.byref(h)))
        return CairoRun(h)

    @staticmethod
    def from_arrays(regs, addrs, values, program_size, segments=(), fe_encoding=SP_FE_CANON_BE):
        """sp_cairo_run_from_arrays: regs (steps, 3) uint64 rows (ap, fp, pc); addrs (n,) uint64; values (n, 32) uint8 in `fe_encoding`;
        segments: (type, start, end) triples (0 RangeCheck, 1 Output) - cairo-vm's relocated outputs as the reference receives them."""
        lib = _lib.load()
        h = ctypes.c_void_p()
        r = np.ascontiguousarray(regs, dtype=np.uint64).reshape(-1, 3)
        a = np.ascontiguousarray(addrs, dtype=np.uint64)
        v = np.ascontiguousarray(values, dtype=np.uint8).reshape(-1, 32)
        st = np.array([s[0] for s in segments], dtype=np.uint8)
        sr = np.array([x for s in segments for x in s[1:]], dtype=np.uint64)
        check(lib.sp_cairo_run_from_arrays(r.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), ctypes.c_uint64(r.shape[0]),
                                           a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), _u8p(v), int(fe_encoding), ctypes.c_uint64(a.shape[0]),
                                           ctypes.c_uint64(program_size), _u8p(st) if len(segments) else None,
                                           sr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)) if len(segments) else None, ctypes.c_uint32(len(segments)), ctypes.byref(h)))
        return CairoRun(h)

    def export(self, fe_encoding=SP_FE_CANON_BE):
        """sp_cairo_run_export: (regs (steps, 3) uint64, addrs (n,) uint64, values (n, 32) uint8) of this run."""
        steps, cells = ctypes.c_uint64(), ctypes.c_uint64()
        check(self._lib.sp_cairo_run_export(self._h, int(fe_encoding), ctypes.byref(steps), ctypes.byref(cells), None, None, None))
        regs = np.empty((steps.value, 3), dtype=np.uint64)
        addrs = np.empty(cells.value, dtype=np.uint64)
        values = np.empty((cells.value, 32), dtype=np.uint8)
        check(self._lib.sp_cairo_run_export(self._h, int(fe_encoding), None, None, regs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                            addrs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), _u8p(values)))
        return regs, addrs, values

    def main_trace(self, fe_encoding=SP_FE_CANON_BE):
        out = np.empty((self.n_rows, self.n_cols, 32), dtype=np.uint8)
        check(self._lib.sp_cairo_run_main_trace(self._h, fe_encoding, _u8p(out)))
        return out

    def timings(self):
        """sp_cairo_run_timings: ms spent in the VM, the shape pass, the upload image and (once built) the host table."""
        v = (ctypes.c_double * 4)()
        check(self._lib.sp_cairo_run_timings(self._h, v))
        return {"vm_ms": round(v[0], 2), "trace_shape_ms": round(v[1], 2), "device_image_ms": round(v[2], 2), "host_table_ms": round(v[3], 2)}

    def main_trace_dev(self, ctx, fe_encoding=SP_FE_CANON_BE):
        """sp_cairo_run_main_trace_dev: the same table, built by the device from the run's register states and memory."""
        out = np.empty((self.n_rows, self.n_cols, 32), dtype=np.uint8)
        check(self._lib.sp_cairo_run_main_trace_dev(ctx._h, self._h, fe_encoding, _u8p(out)))
        return out

    def columns(self):
        """(address, n_rows, n_cols, pinned) of the run's own column-major main trace in the device layout (sp_cairo_run_columns)."""
        p, n, c, pin = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_int()
        check(self._lib.sp_cairo_run_columns(self._h, ctypes.byref(p), ctypes.byref(n), ctypes.byref(c), ctypes.byref(pin)))
        return p.value, n.value, c.value, bool(pin.value)

    def public_memory(self):
        pi = self.public_inputs_c
        raw = ctypes.string_at(pi.public_memory, 64 * pi.n_public_memory)
        return [(int.from_bytes(raw[64 * i:64 * i + 32], "big"), int.from_bytes(raw[64 * i + 32:64 * i + 64], "big"))
                for i in range(pi.n_public_memory)]

    def close(self):
        if self._h:
            self._lib.sp_cairo_run_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cairo_verify(proof, public_inputs_c, options, merkle_backend=0):
    """verify_cairo_proof (reference src/cairo/air.rs:1176-1182) on the host CPU through the library (no GPU needed).
    merkle_backend: SP_MERKLE_KECCAK256 (the reference's trees) or SP_MERKLE_POSEIDON (sp_cairo_verify_backend)."""
    lib = _lib.load()
    opt = options.to_c()
    return lib.sp_cairo_verify_backend(proof, ctypes.c_uint64(len(proof)), ctypes.byref(public_inputs_c), ctypes.byref(opt), int(merkle_backend)) == 1


def last_error():
    """sp_last_error(): the message behind the last non-OK return of this thread's calls; after a 0 from the verifiers its first word
    tells "rejected:" / "malformed:" / "non-canonical framing:" apart (include/stark252_hip.h, sp_cairo_verify)."""
    return _lib.load().sp_last_error().decode(errors="replace")


def air_verify(proof, desc, options, merkle_backend=0):
    """sp_air_verify(_backend): the library's CPU verifier for an AIR given as a constraint program."""
    lib = _lib.load()
    opt = options.to_c()
    return lib.sp_air_verify_backend(proof, ctypes.c_uint64(len(proof)), ctypes.byref(desc), ctypes.byref(opt), int(merkle_backend)) == 1


def poseidon_host(mode, values):
    """sp_poseidon_host on canonical integers: mode 0 hash_many, 1 hash(x, y), 2 hash_single(x), 3 the Hades permutation (three outputs)."""
    lib = _lib.load()
    buf = b"".join(int(v).to_bytes(32, "big") for v in values)
    out = ctypes.create_string_buffer(96)
    check(lib.sp_poseidon_host(SP_FE_CANON_BE, int(mode), buf, ctypes.c_uint64(len(values)), out))
    r = [int.from_bytes(out.raw[32 * k:32 * k + 32], "big") for k in range(3)]
    return r if mode == 3 else r[0]


def proof_file_verify(file_bytes, options, merkle_backend=0):
    """sp_proof_file_verify: the reference CLI's `verify` command (src/main.rs:113-143) on the bytes of a proof file."""
    lib = _lib.load()
    opt = options.to_c()
    return lib.sp_proof_file_verify_backend(file_bytes, ctypes.c_uint64(len(file_bytes)), ctypes.byref(opt), int(merkle_backend)) == 1


def proof_file_bytes(proof, run):
    """u64_be(len) || proof || PublicInputs, the file format of the reference CLI (src/main.rs:98-102)."""
    lib = _lib.load()
    out = ctypes.POINTER(ctypes.c_uint8)()
    ln = ctypes.c_uint64()
    check(lib.sp_proof_file_encode(proof, ctypes.c_uint64(len(proof)), run._h, ctypes.byref(out), ctypes.byref(ln)))
    b = ctypes.string_at(out, ln.value)
    lib.sp_free(out)
    return b


def generate_prover_args_fibonacci(fib_index):
    """(main_trace, public_inputs) for fib(1, 1, fib_index), the shape of the reference's benches."""
    return CairoRun.fibonacci(fib_index)


# ---- multi-GPU: coset sharding ---------------------------------------------------------------
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64)


def shard_global_index(i_loc, logb, shard_log, shard_rank):
    """Global LDE index of local index i_loc on the rank holding the cosets c = c_loc * 2^shard_log + shard_rank
    (same map as `shard_global_index` in csrc/stark_kernels.hip)."""
    lb_loc = logb - shard_log
    c_loc, q = i_loc & ((1 << lb_loc) - 1), i_loc >> lb_loc
    return (q << logb) + (c_loc << shard_log) + shard_rank


def interleave_shards(gathered, n, logb, shard_log):
    """numpy mirror of `interleave_shards_kernel`: gathered[r][q*b_loc + c_loc] -> out[q*b + c_loc*G + r]."""
    world = 1 << shard_log
    g = np.asarray(gathered)
    n_loc = g.shape[1]
    out = np.empty((world * n_loc,) + g.shape[2:], dtype=g.dtype)
    for r in range(world):
        idx = np.array([shard_global_index(i, logb, shard_log, r) for i in range(n_loc)])
        out[idx] = g[r]
    return out


class StagedAllGather:
    """Blocking all-gather hook over torch.distributed with host staging (works with the gloo backend, i.e. also when
    several ranks share one GPU). Production multi-GPU runs use the library's own RCCL communicator instead
    (Context.init_rccl). `device_memory=False` treats the pointers as host memory (CPU-only protocol tests)."""

    def __init__(self, group=None, device_memory=True):
        import torch.distributed as dist
        self.dist, self.group, self.device_memory = dist, group, device_memory
        self.world = dist.get_world_size(group)
        if device_memory:
            self.hip = ctypes.CDLL("libamdhip64.so")
            self.hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.cfn = ALLGATHER_FN(self._call)
        self.a2a_cfn = ALLGATHER_FN(self._call_a2a)   # sp_alltoall_fn has the same C signature

    def _copy(self, dst, src, nbytes, kind):
        if self.device_memory:
            return self.hip.hipMemcpy(dst, src, nbytes, kind)  # 1 = H2D, 2 = D2H
        ctypes.memmove(dst, src, nbytes)
        return 0

    def _call(self, user, send, recv, nbytes):
        try:
            import torch
            host = np.empty(nbytes, dtype=np.uint8)
            if self._copy(host.ctypes.data, send, nbytes, 2) != 0:
                return -1
            outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
            self.dist.all_gather(outs, torch.from_numpy(host), group=self.group)
            allb = torch.cat(outs).numpy()
            if self._copy(recv, allb.ctypes.data, allb.nbytes, 1) != 0:
                return -2
            return 0
        except Exception:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return -3


def _staged_call_a2a(self, user, send, recv, nbytes):
    """sp_alltoall_fn: block d of send goes to rank d, block s of recv came from rank s (host-staged all_to_all_single)."""
    try:
        import torch
        total = nbytes * self.world
        host = np.empty(total, dtype=np.uint8)
        if self._copy(host.ctypes.data, send, total, 2) != 0:
            return -1
        out = torch.empty(total, dtype=torch.uint8)
        self.dist.all_to_all_single(out, torch.from_numpy(host), group=self.group)
        o = out.numpy()
        if self._copy(recv, o.ctypes.data, total, 1) != 0:
            return -2
        return 0
    except Exception:  # never let an exception cross the C boundary
        import traceback
        traceback.print_exc()
        return -3


StagedAllGather._call_a2a = _staged_call_a2a


ALLGATHER_ASYNC_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p)
_HOST_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p)


class StagedAsyncAllGather:
    """Test aid for the stream-ordered hooks (sp_set_collective_async, sp_set_alltoall_async) when the ranks share one GPU: an exchange
    is a device-to-host copy, a host function (hipLaunchHostFunc) that runs the gloo collective on a process group OF ITS OWN - one
    per stream the prover enqueues on (its communication stream, its compute stream), beside the blocking hooks' collectives of the
    main thread - and a host-to-device copy, all enqueued on the stream the prover hands over.  Production runs use the library's
    RCCL communicator (ncclAllGather / grouped ncclSend + ncclRecv on that stream)."""

    def __init__(self, alltoall=True):
        import torch.distributed as dist
        self.dist = dist
        self.groups = [dist.new_group(backend="gloo"), dist.new_group(backend="gloo")]
        self.stream_slot = {}                  # stream handle -> group index, in the order the streams are first seen (the same on every rank)
        self.world = dist.get_world_size()
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        self.hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        self.hip.hipLaunchHostFunc.argtypes = [ctypes.c_void_p, _HOST_FN, ctypes.c_void_p]
        self.hip.hipHostFree.argtypes = [ctypes.c_void_p]
        self.pending, self.calls = {}, 0
        self.pool = {}                         # free page-locked staging buffers by size
        self.live, self.ran = [], {}           # staging buffers in flight; per stream slot, the last call whose host function has run
        self.cfn = ALLGATHER_ASYNC_FN(self._call)
        self.a2a_cfn = ALLGATHER_ASYNC_FN(self._call_a2a) if alltoall else None     # sp_alltoall_async_fn has the same C signature
        self.hostfn = _HOST_FN(self._host)

    def _enqueue(self, kind, send, recv, send_bytes, recv_bytes, stream):
        try:
            key = int(stream or 0)
            if key not in self.stream_slot:
                self.stream_slot[key] = len(self.stream_slot)
            slot = self.stream_slot[key]
            # (a stream runs in order: once the host function of a LATER call on it has run, the copies of an earlier one are done)
            done = [b for b in self.live if b[0] == slot and b[1] < self.ran.get(slot, 0)]
            for b in done:                     # back to the pool (page-locking memory costs ~0.1 ms per MB: never per call)
                self.pool.setdefault(b[4], []).append(b[2]); self.pool.setdefault(b[5], []).append(b[3])
                self.live.remove(b)

            def take(nbytes):
                free = self.pool.get(nbytes)
                if free:
                    return free.pop()
                p = ctypes.c_void_p()
                return p.value if self.hip.hipHostMalloc(ctypes.byref(p), nbytes, 0) == 0 else None
            hs_v, hr_v = take(send_bytes), take(recv_bytes)
            if hs_v is None or hr_v is None:
                return -1
            hs, hr = ctypes.c_void_p(hs_v), ctypes.c_void_p(hr_v)
            self.calls += 1
            self.live.append((slot, self.calls, hs.value, hr.value, send_bytes, recv_bytes))
            self.pending[self.calls] = (kind, hs.value, hr.value, send_bytes, recv_bytes, slot)
            if self.hip.hipMemcpyAsync(hs, send, send_bytes, 2, stream) != 0:
                return -2
            if self.hip.hipLaunchHostFunc(stream, self.hostfn, ctypes.c_void_p(self.calls)) != 0:
                return -3
            if self.hip.hipMemcpyAsync(recv, hr, recv_bytes, 1, stream) != 0:
                return -4
            return 0
        except Exception:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return -5

    def _call(self, user, send, recv, nbytes, stream):
        return self._enqueue("allgather", send, recv, nbytes, nbytes * self.world, stream)

    def _call_a2a(self, user, send, recv, nbytes, stream):
        return self._enqueue("alltoall", send, recv, nbytes * self.world, nbytes * self.world, stream)

    def _host(self, key):
        try:
            import torch
            kind, hs, hr, send_bytes, recv_bytes, slot = self.pending.pop(int(key))
            group = self.groups[slot]
            mine = torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * send_bytes).from_address(hs)).copy())
            if kind == "allgather":
                outs = [torch.empty(send_bytes, dtype=torch.uint8) for _ in range(self.world)]
                self.dist.all_gather(outs, mine, group=group)
                for i, o in enumerate(outs):
                    ctypes.memmove(hr + i * send_bytes, o.data_ptr(), send_bytes)
            else:
                out = torch.empty(recv_bytes, dtype=torch.uint8)
                self.dist.all_to_all_single(out, mine, group=group)
                ctypes.memmove(hr, out.data_ptr(), recv_bytes)
            self.ran[slot] = max(self.ran.get(slot, 0), int(key))
        except Exception:
            import traceback
            traceback.print_exc()


def _ctx_set_collective_async(self, hook):
    """Install a stream-ordered all-gather hook (sp_set_collective_async); call after set_collective."""
    self._async_hook = hook  # keep the callbacks alive
    check(self._lib.sp_set_collective_async(self._h, hook.cfn if hook is not None else None))
    if hook is not None and getattr(hook, "a2a_cfn", None) is not None:
        check(self._lib.sp_set_alltoall_async(self._h, hook.a2a_cfn))


def _ctx_set_collective(self, world, rank, hook, alltoall=True):
    """Install the blocking all-gather hook (and, unless alltoall=False, the all-to-all hook of the digest exchange)."""
    self._hook = hook  # keep the callbacks alive
    check(self._lib.sp_set_collective(self._h, world, rank, hook.cfn if hook is not None else None, None))
    if hook is not None and alltoall and hasattr(hook, "a2a_cfn"):
        check(self._lib.sp_set_alltoall(self._h, hook.a2a_cfn))


def _ctx_init_null(self, world, rank):
    """sp_comm_init_null: timing-only transport (projection of one rank's share on a single GPU; the proof bytes are meaningless)."""
    check(self._lib.sp_comm_init_null(self._h, world, rank))


def _ctx_comm_selftest(self, bytes_per_block=1 << 20):
    check(self._lib.sp_comm_selftest(self._h, ctypes.c_uint64(bytes_per_block)))


def _ctx_comm_measure(self, bytes_per_rank=64 << 20):
    """sp_comm_measure (every rank calls it): the transport's all-gather and all-to-all timed at bytes_per_rank; 0 = read back."""
    out = (ctypes.c_double * 6)()
    check(self._lib.sp_comm_measure(self._h, ctypes.c_uint64(int(bytes_per_rank)), out))
    return {"allgather_ms": out[0], "allgather_gbs_per_link": out[1], "alltoall_ms": out[2], "alltoall_gbs_per_link": out[3],
            "bytes_per_rank": int(out[4]), "world": int(out[5])}


LINK_MEASURED_MARGIN = 1.25      # SP_LINK_MEASURED_MARGIN: mode 2 divides a MEASURED all-gather rate by this before it applies the rule


def model_shard_interpolation(link_gbs, groups, log2_rows):
    """sp_model_shard_interpolation: the decision rule of SP_OPT_SHARD_INTERPOLATION = 2 (no GPU needed)."""
    lib = _lib.load()
    lib.sp_model_shard_interpolation.argtypes = [ctypes.c_double, ctypes.c_uint32, ctypes.c_uint32]
    return int(lib.sp_model_shard_interpolation(float(link_gbs), int(groups), int(log2_rows)))


def _ctx_comm_time_ms(self):
    """sp_comm_time_ms: (ms in stream-ordered collectives, ms in blocking collectives) since the context was created."""
    out = (ctypes.c_double * 2)()
    check(self._lib.sp_comm_time_ms(self._h, out))
    return out[0], out[1]


def _ctx_comm_stats(self):
    out = (ctypes.c_uint64 * 6)()
    check(self._lib.sp_comm_stats(self._h, out))
    return {"world": out[0], "allgather_calls": out[1], "allgather_bytes": out[2], "alltoall_calls": out[3],
            "alltoall_bytes": out[4], "received_bytes": out[5]}


SP_OPT_FRI_SHARD_MIN_LOG, SP_OPT_SHARD_INTERPOLATION, SP_OPT_UPLOAD_THREADS, SP_OPT_MERKLE_BACKEND, SP_OPT_MERKLE_ONE_COLUMN_ROWS, SP_OPT_DEVICE_TRACE, SP_OPT_LINK_GBS, SP_OPT_HOST_RANKS = 1, 2, 3, 4, 5, 6, 7, 8
SP_MERKLE_KECCAK256, SP_MERKLE_POSEIDON = 0, 1
SP_PREWARM_KERNELS, SP_PREWARM_CLOCKS, SP_PREWARM_HOST_ROWS, SP_PREWARM_ALL = 1, 2, 4, 7


def _ctx_set_option(self, key, value):
    check(self._lib.sp_set_option(self._h, int(key), ctypes.c_int64(int(value))))


def _ctx_init_rccl(self, group=None):
    """Native path: ncclAllGather over xGMI on the context stream; the 128-byte id travels over torch.distributed."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    idb = (ctypes.c_uint8 * 128)()
    if rank == 0:
        check(self._lib.sp_comm_unique_id(idb))
    obj = [bytes(idb)]
    dist.broadcast_object_list(obj, src=0, group=group)
    check(self._lib.sp_comm_init_rccl(self._h, obj[0], world, rank))


Context.set_collective = _ctx_set_collective
Context.init_rccl = _ctx_init_rccl
Context.init_null = _ctx_init_null
Context.set_collective_async = _ctx_set_collective_async
Context.comm_stats = _ctx_comm_stats
Context.comm_selftest = _ctx_comm_selftest
Context.comm_measure = _ctx_comm_measure
Context.comm_time_ms = _ctx_comm_time_ms
Context.set_option = _ctx_set_option
__all__ += ["cairo_verify", "proof_file_bytes", "proof_file_verify", "StagedAllGather", "StagedAsyncAllGather", "shard_global_index", "interleave_shards",
            "SP_OPT_FRI_SHARD_MIN_LOG", "SP_OPT_SHARD_INTERPOLATION", "SP_OPT_UPLOAD_THREADS", "SP_OPT_MERKLE_BACKEND", "SP_OPT_MERKLE_ONE_COLUMN_ROWS", "SP_OPT_DEVICE_TRACE", "SP_OPT_LINK_GBS", "SP_OPT_HOST_RANKS", "host_cpu_budget", "last_error", "model_shard_interpolation",
            "SP_MERKLE_KECCAK256", "SP_MERKLE_POSEIDON", "poseidon_host", "host_bind_to_device",
            "SP_PREWARM_KERNELS", "SP_PREWARM_CLOCKS", "SP_PREWARM_HOST_ROWS", "SP_PREWARM_ALL"]

"""Builder for `sp_air_desc` (include/stark252_hip.h): an AIR other than Cairo, given as a straight-line program over
frame cells.  Mirrors what an implementor of the reference's `AIR` trait provides (src/starks/traits.rs:15-119):
context (columns, transition offsets / degrees / exemptions), `compute_transition`, `boundary_constraints`,
`composition_poly_degree_bound`, the number of RAP challenges and the auxiliary-trace builder (by kind).

    b = AirBuilder(main_cols=1, offsets=[0, 1, 2], degree_bound_factor=1)
    r0, r1, r2 = b.load(0, 0), b.load(1, 0), b.load(2, 0)
    b.constraint(r2 - r1 - r0, degree=1, exemptions=2)
    b.boundary(col=0, step=0, value=1); b.boundary(0, 1, 1)
    desc = b.build()
"""
import ctypes

P = 2**251 + 17 * 2**192 + 1

OP_LOAD, OP_CONST, OP_ADD, OP_SUB, OP_MUL, OP_OUT = range(6)
MAX_OFFSETS, MAX_TRANSITIONS = 8, 64
AUX_TRACE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint8))


class AirOpC(ctypes.Structure):
    _fields_ = [("op", ctypes.c_uint8), ("pad", ctypes.c_uint8), ("a", ctypes.c_uint16), ("b", ctypes.c_uint16), ("pad2", ctypes.c_uint16)]


class AirBoundaryC(ctypes.Structure):
    _fields_ = [("col", ctypes.c_uint32), ("pad", ctypes.c_uint32), ("step", ctypes.c_uint64), ("value", ctypes.c_uint8 * 32)]


class AirDescC(ctypes.Structure):
    _fields_ = [("main_cols", ctypes.c_uint32), ("aux_cols", ctypes.c_uint32),
                ("n_offsets", ctypes.c_uint32), ("offsets", ctypes.c_uint32 * MAX_OFFSETS),
                ("n_transitions", ctypes.c_uint32), ("degrees", ctypes.c_uint32 * MAX_TRANSITIONS),
                ("exemptions", ctypes.c_uint32 * MAX_TRANSITIONS),
                ("num_transition_exemptions", ctypes.c_uint32), ("degree_bound_factor", ctypes.c_uint32),
                ("n_ops", ctypes.c_uint32), ("ops", ctypes.POINTER(AirOpC)),
                ("n_consts", ctypes.c_uint32), ("consts", ctypes.c_void_p),
                ("n_rap", ctypes.c_uint32), ("aux_kind", ctypes.c_uint32),
                ("n_boundary", ctypes.c_uint32), ("boundary", ctypes.POINTER(AirBoundaryC)),
                ("aux_fn", AUX_TRACE_FN), ("aux_user", ctypes.c_void_p)]


def _check_layout():
    """The library reports the size of its sp_air_desc: a mirror with other field sizes must not reach sp_air_prove."""
    from . import _lib
    want = _lib.load().sp_air_desc_size()
    if ctypes.sizeof(AirDescC) != want:
        raise ImportError(f"AirDescC is {ctypes.sizeof(AirDescC)} bytes, the library's sp_air_desc {want}: the binding is out of date")


AUX_NONE, AUX_FIBONACCI_RAP, AUX_CALLBACK = 0, 1, 2
_RAP_TAG = 0x8000


class Value:
    """An SSA value of the constraint program; arithmetic operators emit ops into the owning builder."""

    def __init__(self, builder, index):
        self.b, self.i = builder, index

    def _lift(self, other):
        return other if isinstance(other, Value) else self.b.const(other)

    def __add__(self, o): return self.b._emit(OP_ADD, self.i, self._lift(o).i)
    def __sub__(self, o): return self.b._emit(OP_SUB, self.i, self._lift(o).i)
    def __mul__(self, o): return self.b._emit(OP_MUL, self.i, self._lift(o).i)
    __radd__ = __add__
    __rmul__ = __mul__

    def __rsub__(self, o): return self._lift(o) - self


class AirBuilder:
    def __init__(self, main_cols, offsets, degree_bound_factor, aux_cols=0, n_rap=0, aux_kind=AUX_NONE, num_transition_exemptions=1,
                 aux_builder=None):
        """aux_builder(rap: list[int]) -> (n, aux_cols) nested list of ints: the AIR's build_auxiliary_trace (aux_kind AUX_CALLBACK)."""
        assert 1 <= len(offsets) <= MAX_OFFSETS
        self.aux_builder = aux_builder
        self.main_cols, self.aux_cols, self.offsets = main_cols, aux_cols, list(offsets)
        self.degree_bound_factor, self.n_rap, self.aux_kind = degree_bound_factor, n_rap, aux_kind
        self.num_transition_exemptions = num_transition_exemptions
        self.ops, self.consts, self.degrees, self.exemptions, self.bcs = [], [], [], [], []

    def _emit(self, op, a, b):
        self.ops.append((op, a, b))
        return Value(self, len(self.ops) - 1)

    def load(self, row, col):
        """Cell (frame row `row` = index into the transition offsets, column `col` of main||aux)."""
        assert row < len(self.offsets) and col < self.main_cols + self.aux_cols
        return self._emit(OP_LOAD, row, col)

    def const(self, v):
        v %= P
        if v not in self.consts:
            self.consts.append(v)
        return self._emit(OP_CONST, self.consts.index(v), 0)

    def rap(self, i):
        assert i < self.n_rap
        return self._emit(OP_CONST, _RAP_TAG | i, 0)

    def constraint(self, value, degree, exemptions):
        assert len(self.degrees) < MAX_TRANSITIONS
        self._emit(OP_OUT, len(self.degrees), value.i)
        self.degrees.append(degree)
        self.exemptions.append(exemptions)

    def boundary(self, col, step, value):
        self.bcs.append((col, step, value % P))

    def build(self):
        """Returns (AirDescC, keepalive)."""
        _check_layout()
        d = AirDescC()
        d.main_cols, d.aux_cols = self.main_cols, self.aux_cols
        d.n_offsets = len(self.offsets)
        for i, o in enumerate(self.offsets):
            d.offsets[i] = o
        d.n_transitions = len(self.degrees)
        for i, (g, e) in enumerate(zip(self.degrees, self.exemptions)):
            d.degrees[i], d.exemptions[i] = g, e
        d.num_transition_exemptions = self.num_transition_exemptions
        d.degree_bound_factor = self.degree_bound_factor
        ops = (AirOpC * len(self.ops))()
        for i, (op, a, b) in enumerate(self.ops):
            if op == OP_CONST and (a & _RAP_TAG):
                a = len(self.consts) + (a & ~_RAP_TAG)   # RAP challenges follow the constants
            ops[i].op, ops[i].a, ops[i].b = op, a, b
        consts = ctypes.create_string_buffer(b"".join(c.to_bytes(32, "big") for c in self.consts), max(1, 32 * len(self.consts)))
        bcs = (AirBoundaryC * max(1, len(self.bcs)))()
        for i, (col, step, value) in enumerate(self.bcs):
            bcs[i].col, bcs[i].step = col, step
            ctypes.memmove(bcs[i].value, value.to_bytes(32, "big"), 32)
        d.n_ops, d.ops = len(self.ops), ctypes.cast(ops, ctypes.POINTER(AirOpC))
        d.n_consts, d.consts = len(self.consts), ctypes.cast(consts, ctypes.c_void_p)
        d.n_rap, d.aux_kind = self.n_rap, self.aux_kind
        d.n_boundary, d.boundary = len(self.bcs), ctypes.cast(bcs, ctypes.POINTER(AirBoundaryC))
        cb = None
        if self.aux_kind == AUX_CALLBACK:
            builder, aux_cols = self.aux_builder, self.aux_cols

            def _aux(user, rap_ptr, n_rap, out_ptr):   # canonical big-endian contexts (the default encoding)
                try:
                    raw = ctypes.string_at(rap_ptr, 32 * n_rap)
                    rows = builder([int.from_bytes(raw[32 * i:32 * i + 32], "big") for i in range(n_rap)])
                    flat = b"".join((int(v) % P).to_bytes(32, "big") for row in rows for v in row)
                    assert len(flat) == 32 * aux_cols * len(rows)
                    ctypes.memmove(out_ptr, flat, len(flat))
                    return 0
                except Exception:  # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return -1
            cb = AUX_TRACE_FN(_aux)
            d.aux_fn = cb
        return d, (ops, consts, bcs, cb)


# ---- the reference's example AIRs in program form (src/starks/example/*.rs) ------------------------------------------
def simple_fibonacci(a0=1, a1=1):
    b = AirBuilder(1, [0, 1, 2], 1)
    b.constraint(b.load(2, 0) - b.load(1, 0) - b.load(0, 0), degree=1, exemptions=2)
    b.boundary(0, 0, a0); b.boundary(0, 1, a1)
    return b


def fibonacci_2_columns(a0=1, a1=1):
    b = AirBuilder(2, [0, 1], 1)
    b.constraint(b.load(1, 0) - b.load(0, 0) - b.load(0, 1), 1, 1)
    b.constraint(b.load(1, 1) - b.load(0, 1) - b.load(1, 0), 1, 1)
    b.boundary(0, 0, a0); b.boundary(1, 0, a1)
    return b


def quadratic(a0=3):
    b = AirBuilder(1, [0, 1], 2)
    x = b.load(0, 0)
    b.constraint(b.load(1, 0) - x * x, 2, 1)
    b.boundary(0, 0, a0)
    return b


def dummy():
    b = AirBuilder(2, [0, 1, 2], 1)
    f = b.load(0, 0)
    b.constraint(f * (f - 1), 2, 0)
    b.constraint(b.load(2, 1) - b.load(1, 1) - b.load(0, 1), 1, 2)
    b.boundary(1, 0, 1); b.boundary(1, 1, 1)
    return b


def fibonacci_rap(trace_length, steps):
    b = AirBuilder(2, [0, 1, 2], 1, aux_cols=1, n_rap=1, aux_kind=AUX_FIBONACCI_RAP, num_transition_exemptions=2)
    b.constraint(b.load(2, 0) - b.load(1, 0) - b.load(0, 0), 1, 3 + trace_length - steps - 1)
    gamma = b.rap(0)
    b.constraint(b.load(1, 2) * (b.load(0, 1) + gamma) - b.load(0, 2) * (b.load(0, 0) + gamma), 2, 1)
    b.boundary(0, 0, 1); b.boundary(0, 1, 1); b.boundary(2, 0, 1)
    return b

"""sp_cairo_run_from_dumps (the two binary files of `cairo-run`: reference src/cairo/cairo_mem.rs:35-61, register_states.rs:51-78) on
damaged input - flipped bits, truncated files, huge addresses and register values, duplicated cells, impossible program sizes: the
reader either reports an error code or returns a run with a well-formed main trace; it never crashes, hangs or raises anything else."""
import os
import random

import pytest

from lambdaworks_cairo_prover_amd import api

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _read(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


@pytest.mark.parametrize("block", range(4))
def test_damaged_dumps_give_an_error_or_a_well_formed_run(block):
    t0, m0, t1, m1 = _read("program.trace"), _read("program.memory"), _read("mul_trace.out"), _read("mul_mem.out")
    accepted = refused = 0
    for seed in range(100 * block, 100 * block + 100):
        rng = random.Random(seed)
        t, m, ps = (bytearray(t0), bytearray(m0), rng.choice([1, 5, len(m0) // 40])) if rng.random() < 0.6 else (bytearray(t1), bytearray(m1), 5)
        kind = rng.choice(["tbyte", "mbyte", "ttrunc", "mtrunc", "tbig", "mbig", "mdup", "ps"])
        if kind == "tbyte":
            for _ in range(rng.randrange(1, 4)):
                t[rng.randrange(len(t))] ^= 1 << rng.randrange(8)
        elif kind == "mbyte":
            for _ in range(rng.randrange(1, 4)):
                m[rng.randrange(len(m))] ^= 1 << rng.randrange(8)
        elif kind == "ttrunc":
            t = t[:rng.randrange(len(t))]
        elif kind == "mtrunc":
            m = m[:rng.randrange(len(m))]
        elif kind == "tbig":
            i = rng.randrange(len(t) // 8) * 8
            t[i:i + 8] = rng.choice([2**63, 2**64 - 1, 2**40]).to_bytes(8, "little")
        elif kind == "mbig":
            i = rng.randrange(len(m) // 40) * 40
            m[i:i + 8] = rng.choice([2**63, 2**64 - 1, 2**40, 2**34]).to_bytes(8, "little")
        elif kind == "mdup":
            i = rng.randrange(len(m) // 40) * 40
            m += m[i:i + 40]
        else:
            ps = rng.choice([0, 2**32, 2**63, len(m) // 40 + 5])
        try:
            run = api.CairoRun.from_dumps(bytes(t), bytes(m), program_size=ps)
        except api.SpError:
            refused += 1
            continue
        trace = run.main_trace()
        assert trace.shape[1] == 34 and trace.shape[0] == run.n_rows and run.n_rows & (run.n_rows - 1) == 0 and run.n_rows >= run.num_steps
        accepted += 1
    assert accepted > 0 and refused > 0


def test_damaged_arrays_give_an_error_or_a_well_formed_run():
    """sp_cairo_run_from_arrays (cairo-vm's relocated trace and memory handed over in memory): a shuffled or duplicated memory list
    gives the very same main trace; flipped registers, addresses and values, dropped cells, huge addresses and impossible program sizes
    give an error code or a well-formed run - never a crash."""
    import numpy as np

    import cairo_asm as A
    base = []
    for seed in range(4):
        words, entry = A.random_program(seed, 25)
        run = api.CairoRun.from_program(words, entry_pc=entry)
        base.append((run, len(words)) + run.export())
    same = accepted = refused = 0
    for i in range(600):
        rng = random.Random(i)
        run0, psize, regs, addrs, vals = base[i % len(base)]
        regs, addrs, vals, ps = regs.copy(), addrs.copy(), vals.copy(), psize
        kind = rng.choice(["none", "reg", "addr", "val", "drop", "dup", "psize", "shuffle", "huge"])
        if kind == "reg":
            regs[rng.randrange(len(regs)), rng.randrange(3)] ^= np.uint64(1 << rng.randrange(20))
        elif kind == "addr":
            addrs[rng.randrange(len(addrs))] ^= np.uint64(1 << rng.randrange(12))
        elif kind == "val":
            vals[rng.randrange(len(vals)), rng.randrange(32)] ^= 1 << rng.randrange(8)
        elif kind == "drop":
            j = rng.randrange(len(addrs))
            addrs, vals = np.delete(addrs, j), np.delete(vals, j, axis=0)
        elif kind == "dup":
            j = rng.randrange(len(addrs))
            addrs, vals = np.append(addrs, addrs[j]), np.append(vals, vals[j:j + 1], axis=0)
        elif kind == "psize":
            ps = rng.choice([0, 1, psize + 3, 2**40])
        elif kind == "shuffle":
            perm = np.array(rng.sample(range(len(addrs)), len(addrs)))
            addrs, vals = addrs[perm], vals[perm]
        elif kind == "huge":
            addrs[rng.randrange(len(addrs))] = np.uint64(rng.choice([2**40, 2**63, 2**64 - 1]))
        try:
            run = api.CairoRun.from_arrays(regs, addrs, vals, ps)
        except api.SpError:
            refused += 1
            continue
        trace = run.main_trace()
        assert trace.shape[1] == 34 and trace.shape[0] & (trace.shape[0] - 1) == 0
        accepted += 1
        if kind in ("none", "shuffle", "dup"):
            assert np.array_equal(trace, run0.main_trace()), (i, kind)
            same += 1
    assert same > 100 and refused > 100


def test_conflicting_duplicate_address_is_an_error():
    """Cairo memory is write-once: the same address twice with the same value is one cell, with two values it is
    InconsistentMemory - in the flat-array path (filled by several threads: no schedule-dependent winner) and in the sparse one."""
    import numpy as np

    import cairo_asm as A
    words, entry = A.random_program(1, 25)
    run0 = api.CairoRun.from_program(words, entry_pc=entry)
    regs, addrs, vals = run0.export()
    for far in (False, True):
        a, v = addrs.copy(), vals.copy()
        if far:     # one cell far away sends the whole list cell by cell through CairoMemory::set
            a, v = np.append(a, np.uint64(2**40)), np.append(v, v[:1], axis=0)
        for j in (0, len(addrs) // 2, len(addrs) - 1):
            same = api.CairoRun.from_arrays(regs, np.append(a, a[j]), np.append(v, v[j:j + 1], axis=0), len(words))
            assert np.array_equal(same.main_trace(), run0.main_trace())
            other = v[j:j + 1].copy()
            other[0, 31] ^= 1
            for order in (0, 1):    # the conflicting entry behind or in front of the original
                aa = np.append(a, a[j]) if order == 0 else np.insert(a, 0, a[j])
                vv = np.append(v, other, axis=0) if order == 0 else np.insert(v, 0, other, axis=0)
                with pytest.raises(api.SpError, match="InconsistentMemory"):
                    api.CairoRun.from_arrays(regs, aa, vv, len(words))

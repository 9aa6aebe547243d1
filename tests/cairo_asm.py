"""A few Cairo instructions assembled by hand (Cairo whitepaper section 4.5; decoder: reference src/cairo/decode/
instruction_flags.rs:1-77, instruction_offsets.rs:18-56) and small hint-free programs that use the range_check and output
builtins - the shapes of the reference's `rc_program`, `lt_comparison` and `signed_div_rem` tests
(tests/integration_tests.rs:151-172), whose compiled forms need cairo-vm hints and are not available here."""
P = 2**251 + 17 * 2**192 + 1

DST_FP, OP0_FP, OP1_IMM, OP1_FP, OP1_AP, RES_ADD, RES_MUL, PC_ABS, PC_REL, PC_JNZ, AP_ADD, AP_ADD1, CALL, RET, ASSERT = (1 << i for i in range(15))


def word(flags, off_dst=0, off_op0=-1, off_op1=-1):
    enc = lambda o: (o + 0x8000) & 0xFFFF   # noqa: E731
    return enc(off_dst) | (enc(off_op0) << 16) | (enc(off_op1) << 32) | (flags << 48)


def push_imm(v):                      # [ap] = v; ap++
    return [word(OP0_FP | OP1_IMM | AP_ADD1 | ASSERT, 0, -1, 1), v % P]


def push_fp(k):                       # [ap] = [fp + k]; ap++
    return [word(OP0_FP | OP1_FP | AP_ADD1 | ASSERT, 0, -1, k)]


def push_fp_plus(k, imm):             # [ap] = [fp + k] + imm; ap++
    return [word(OP0_FP | OP1_IMM | RES_ADD | AP_ADD1 | ASSERT, 0, k, 1), imm % P]


def assert_fp_eq_deref_fp(kd, kp, off=0):   # [fp + kd] = [[fp + kp] + off]
    return [word(DST_FP | OP0_FP | ASSERT, kd, kp, off)]


def assert_ap_eq_deref_fp(kd, kp, off=0):   # [ap + kd] = [[fp + kp] + off]
    return [word(OP0_FP | ASSERT, kd, kp, off)]


def call_rel(delta):
    return [word(OP1_IMM | PC_REL | CALL, 0, 1, 1), delta % P]


def jnz_fp(k, delta):                 # jmp rel delta if [fp + k] != 0
    return [word(DST_FP | OP0_FP | OP1_IMM | PC_JNZ, k, -1, 1), delta % P]


def ret():
    return [word(DST_FP | OP0_FP | OP1_FP | PC_ABS | RET, -2, -1, -1)]


def rc_program():
    """%builtins range_check; main{range_check_ptr}: assert_nn(5); assert_nn(2)  (cairo_programs/cairo0/rc_program.cairo).
    Returns (words, entry_pc)."""
    assert_nn = assert_fp_eq_deref_fp(-3, -4) + push_fp_plus(-4, 1) + ret()          # a = [range_check_ptr]; return range_check_ptr + 1
    main_at = 1 + len(assert_nn)
    main = push_fp(-3) + push_imm(5)
    main += call_rel(1 - (main_at + len(main)))
    main += push_imm(2)
    main += call_rel(1 - (main_at + len(main)))
    main += ret()
    return assert_nn + main, main_at


def rc_loop_program(count, start=3, step=5):
    """%builtins range_check; range-checks start, start + step, ... (count values) in a recursive loop.  Returns (words, entry_pc)."""
    # loop(range_check_ptr [fp-5], x [fp-4], n [fp-3]) -> range_check_ptr
    head = jnz_fp(-3, 4) + push_fp(-5) + ret()                                       # n == 0: return range_check_ptr
    body = assert_fp_eq_deref_fp(-4, -5) + push_fp_plus(-5, 1) + push_fp_plus(-4, step) + push_fp_plus(-3, -1)
    body += call_rel(1 - (1 + len(head) + len(body)))
    body += ret()
    loop = head + body
    main_at = 1 + len(loop)
    main = push_fp(-3) + push_imm(start) + push_imm(count)
    main += call_rel(1 - (main_at + len(main)))
    main += ret()
    return loop + main, main_at


def output_rc_program():
    """%builtins output range_check; writes two output cells, range-checks one of the values, returns both pointers."""
    main = push_imm(7) + assert_ap_eq_deref_fp(-1, -4, 0)        # assert [output_ptr] = 7
    main += push_imm(2**100 + 9) + assert_ap_eq_deref_fp(-1, -4, 1)   # assert [output_ptr + 1] = 2^100 + 9
    main += assert_ap_eq_deref_fp(-1, -3, 0)                     # assert [range_check_ptr] = 2^100 + 9
    main += assert_ap_eq_deref_fp(-3, -3, 1)                     # assert [range_check_ptr + 1] = 7
    main += push_fp_plus(-4, 2) + push_fp_plus(-3, 2) + ret()
    return main, 1


def holes_program():
    """A builtin-free program whose run leaves memory holes (ap jumps over 60 cells it never writes) and range-check holes (offsets
    -61 .. 1 with most values in between unused), with a taken and a not-taken jnz.  Returns (words, entry_pc)."""
    main = push_imm(5) + push_imm(0)
    main += [word(OP0_FP | OP1_IMM | AP_ADD, -1, -1, 1), 60]                     # ap += 60          (dst, op0: cells that exist)
    main += [word(OP0_FP | OP1_AP | AP_ADD1 | ASSERT, 0, -1, -62)]               # [ap] = [ap - 62]  (= 5); ap++
    main += [word(OP0_FP | OP1_IMM | PC_JNZ, -1, -1, 1), 3]                      # jmp rel 3 if [ap - 1] != 0   (taken: skips the next push)
    main += push_imm(11)[:1]                                                     # (one word: jumped over together with ...)
    main += [word(OP0_FP | OP1_IMM | PC_JNZ, -62, -1, 1), 2]                     # jmp rel 2 if [ap - 62] != 0  ([ap-62] = 0: not taken)
    main += ret()
    return main, 1


def random_program(seed, length=40):
    """A random hint-free, builtin-free program: pushes of immediates (0, 1, small, near p), sums and products of earlier cells
    (ap-based op0 / op1 and immediates), `ap += k` (memory holes; far offsets make range-check holes), taken and not-taken forward
    `jnz`s over words that are never executed, and calls of a two-instruction function.  The generator tracks ap and every value it
    wrote, so it knows which way each jump goes and what the following offsets mean.  Returns (words, entry_pc)."""
    import random
    rng = random.Random(seed)
    func = push_fp_plus(-3, 7) + [word(OP1_AP | RES_MUL | AP_ADD1 | ASSERT, 0, -1, -1)] + ret()   # f(x): [ap] = x + 7; [ap] = [ap-1]^2; ret
    func_at = 1
    main_at = func_at + len(func)
    main = []
    cells = []                                     # values at fp + 0 .. (cells[i] is None for a cell the run never writes)

    def val(back):
        return cells[len(cells) - back]

    def usable(limit=100):
        return [b for b in range(1, min(limit, len(cells)) + 1) if isinstance(val(b), int)]

    def pick_imm():
        return rng.choice([0, 1, 2, rng.randrange(1 << 16), rng.randrange(P), P - 1, P - rng.randrange(1, 1 << 20)])

    main += push_imm(rng.randrange(1, 1 << 30)); cells.append(main[-1])
    main += push_imm(rng.randrange(P)); cells.append(main[-1])
    for _ in range(length):
        kind = rng.choice(["imm", "add", "mul", "addi", "muli", "hole", "jnz", "call", "add", "mul"])
        u = usable()
        if kind == "imm" or not u:
            v = pick_imm()
            main += push_imm(v); cells.append(v % P)
        elif kind in ("add", "mul"):
            i, j = rng.choice(u), rng.choice(u)
            main += [word(OP1_AP | (RES_ADD if kind == "add" else RES_MUL) | AP_ADD1 | ASSERT, 0, -i, -j)]
            cells.append((val(i) + val(j)) % P if kind == "add" else (val(i) * val(j)) % P)
        elif kind in ("addi", "muli"):
            i, v = rng.choice(u), pick_imm()
            main += [word(OP1_IMM | (RES_ADD if kind == "addi" else RES_MUL) | AP_ADD1 | ASSERT, 0, -i, 1), v % P]
            cells.append((val(i) + v) % P if kind == "addi" else (val(i) * v) % P)
        elif kind == "hole":
            i, k = rng.choice(u), rng.choice([1, 2, 3, 17, 60])
            main += [word(OP1_IMM | AP_ADD, -i, -i, 1), k]                     # ap += k  (dst and op0: a cell that exists)
            cells.extend([None] * k)
        elif kind == "jnz":
            if not isinstance(val(1), int):                                    # (the fall-through doubles [ap - 1]: it has to be a value)
                continue
            i, j = rng.choice(u), rng.choice(u)
            skipped = rng.randrange(1, 4)
            main += [word(OP1_IMM | PC_JNZ, -i, -j, 1), 2 + skipped]           # jmp rel 2 + skipped if [ap - i] != 0
            junk = [word(OP1_AP | RES_ADD | AP_ADD1 | ASSERT, 0, -1, -1)] * skipped
            if val(i) != 0:
                main += junk                                                   # jumped over: never executed
            else:
                main += junk                                                   # not taken: these DO run ([ap] = 2 [ap-1], skipped times)
                for _ in range(skipped):
                    cells.append((2 * val(1)) % P)
        elif kind == "call":
            i = rng.choice(u)
            main += [word(OP1_AP | AP_ADD1 | ASSERT, 0, -i, -i)]               # [ap] = [ap - i] (the argument; op0 is read and not used)
            x = val(i); cells.append(x)
            here = main_at + len(main)
            main += call_rel(func_at - here)
            cells.extend(["fp", "pc"])                                         # the frame: caller's fp, return pc (never used as operands below)
            y = (x + 7) % P
            cells.extend([y, (y * y) % P])
    main += ret()
    return func + main, main_at

// See cairo_host.h. Host-only C++ (compiled by hipcc as plain host code).
#include "cairo_host.h"
#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace sp {

static inline uint64_t fe_low_u64(const fe& a) {
    fe r = fe_from_mont(a);
    return (uint64_t)r.v[0] | ((uint64_t)r.v[1] << 32);
}
// field element -> signed 64-bit offset (x or -(p - x)); throws when neither fits
static inline int64_t fe_to_i64(const fe& a) {
    fe r = fe_from_mont(a);
    bool small = true;
    for (int i = 2; i < 8; ++i) if (r.v[i]) small = false;
    if (small) return (int64_t)((uint64_t)r.v[0] | ((uint64_t)r.v[1] << 32));
    fe n = fe_from_mont(fe_neg(a));
    for (int i = 2; i < 8; ++i) if (n.v[i]) throw std::runtime_error("offset does not fit in 64 bits");
    return -(int64_t)((uint64_t)n.v[0] | ((uint64_t)n.v[1] << 32));
}

bool parse_trace_le(const uint8_t* b, size_t len, std::vector<RegisterState>& out) {
    if (len % 24 != 0) return false;
    out.resize(len / 24);
    for (size_t i = 0; i < out.size(); ++i) {
        uint64_t v[3];
        std::memcpy(v, b + 24 * i, 24);  // little-endian host
        out[i] = RegisterState{v[0], v[1], v[2]};
    }
    return true;
}
bool parse_memory_le(const uint8_t* b, size_t len, CairoMemory& out) {
    if (len % 40 != 0) return false;
    for (size_t i = 0; i < len / 40; ++i) {
        uint64_t addr;
        std::memcpy(&addr, b + 40 * i, 8);
        uint8_t be[32];
        for (int k = 0; k < 32; ++k) be[k] = b[40 * i + 8 + 31 - k];  // value is 32 bytes little-endian
        out.data[addr] = fe_from_bytes_be(be);
    }
    return true;
}

PublicInputs public_inputs_from_regs_and_mem(const std::vector<RegisterState>& regs, const CairoMemory& mem,
                                             size_t program_size, const std::vector<MemorySegment>& segments) {
    PublicInputs p;
    p.memory_segments = segments;
    for (uint64_t i = 1; i <= program_size; ++i) {
        const fe* v = mem.get(i);
        if (!v) throw std::runtime_error("program word missing from memory");
        p.public_memory.push_back({i, *v});
    }
    if (const MemorySegment* out = p.segment(1))
        for (uint64_t a = out->start; a < out->end; ++a) {
            const fe* v = mem.get(a);
            if (!v) throw std::runtime_error("output cell missing from memory");
            p.public_memory.push_back({a, *v});
        }
    const RegisterState& last = regs.back();
    p.pc_init = fe_from_u64(regs[0].pc); p.ap_init = fe_from_u64(regs[0].ap); p.fp_init = fe_from_u64(regs[0].fp);
    p.pc_final = fe_from_u64(last.pc); p.ap_final = fe_from_u64(last.ap);
    p.num_steps = regs.size();
    return p;
}

namespace {
struct Decoded {
    uint32_t flags;  // 15 bits
    uint32_t off_dst, off_op0, off_op1;  // biased 16-bit fields
    int dst_reg, op0_reg, op1_src, res_logic, pc_update, ap_update, opcode;
};
Decoded decode(const fe& word) {
    uint64_t w = fe_low_u64(word);
    Decoded d;
    d.off_dst = (uint32_t)(w & 0xffff); d.off_op0 = (uint32_t)((w >> 16) & 0xffff); d.off_op1 = (uint32_t)((w >> 32) & 0xffff);
    uint32_t f = (uint32_t)(w >> 48);
    d.flags = f & 0x7fff;
    d.dst_reg = f & 1; d.op0_reg = (f >> 1) & 1; d.op1_src = (f >> 2) & 7; d.res_logic = (f >> 5) & 3;
    d.pc_update = (f >> 7) & 7; d.ap_update = (f >> 10) & 3; d.opcode = (f >> 12) & 7;
    auto one_hot_or_zero = [](int v) { return v == 0 || v == 1 || v == 2 || v == 4; };
    if (!one_hot_or_zero(d.op1_src)) throw std::runtime_error("InvalidOp1Src");
    if (d.res_logic == 3) throw std::runtime_error("InvalidResLogic");
    if (!one_hot_or_zero(d.pc_update)) throw std::runtime_error("InvalidPcUpdate");
    if (d.ap_update == 3) throw std::runtime_error("InvalidApUpdate");
    if (!one_hot_or_zero(d.opcode)) throw std::runtime_error("InvalidOpcode");
    return d;
}
inline uint64_t add_signed(uint64_t base, uint32_t biased_off) { return base + (uint64_t)biased_off - 0x8000ULL; }
const fe& mem_at(const CairoMemory& m, uint64_t a) {
    const fe* v = m.get(a);
    if (!v) throw std::runtime_error("memory cell not found");
    return *v;
}
}  // namespace

std::vector<fe> build_main_trace(const std::vector<RegisterState>& regs, const CairoMemory& mem, PublicInputs& pub,
                                 size_t* n_rows, size_t* n_cols) {
    const size_t steps = regs.size();
    const MemorySegment* rc_seg = pub.segment(0);
    const size_t cols = rc_seg ? 43 : 34;
    const fe zero = fe_zero(), one = fe_one();
    std::vector<fe> t;
    t.reserve((steps + steps / 8 + 64) * cols);
    t.resize(steps * cols, zero);
    // build_cairo_execution_trace (execution_trace.rs:261-356)
    for (size_t i = 0; i < steps; ++i) {
        const RegisterState& r = regs[i];
        const fe& inst = mem_at(mem, r.pc);
        Decoded d = decode(inst);
        fe* row = &t[i * cols];
        for (int k = 0; k < 15; ++k) row[k] = ((d.flags >> k) & 1) ? one : zero;
        uint64_t dst_addr = add_signed(d.dst_reg ? r.fp : r.ap, d.off_dst);
        uint64_t op0_addr = add_signed(d.op0_reg ? r.fp : r.ap, d.off_op0);
        fe dst = mem_at(mem, dst_addr);
        fe op0 = mem_at(mem, op0_addr);
        uint64_t op1_base = d.op1_src == 0 ? fe_low_u64(op0) : d.op1_src == 1 ? r.pc : d.op1_src == 2 ? r.fp : r.ap;
        uint64_t op1_addr = add_signed(op1_base, d.off_op1);
        fe op1 = mem_at(mem, op1_addr);
        fe res;
        if (d.pc_update == 4) {  // jnz: res holds dst^-1 (execution_trace.rs:382-440)
            if (!(d.res_logic == 0 && d.opcode == 0 && d.ap_update != 1)) throw std::runtime_error("Undefined Behavior");
            res = fe_is_zero(dst) ? dst : fe_inv(dst);
        } else {
            res = d.res_logic == 0 ? op1 : d.res_logic == 1 ? fe_add(op0, op1) : fe_mul(op0, op1);
        }
        // update_values (execution_trace.rs:572-592)
        if (d.opcode == 1) { op0 = fe_from_u64(r.pc + (d.op1_src == 1 ? 2 : 1)); dst = fe_from_u64(r.fp); }
        else if (d.opcode == 4) { res = dst; }
        row[16] = res; row[17] = fe_from_u64(r.ap); row[18] = fe_from_u64(r.fp); row[19] = fe_from_u64(r.pc);
        row[20] = fe_from_u64(dst_addr); row[21] = fe_from_u64(op0_addr); row[22] = fe_from_u64(op1_addr);
        row[23] = inst; row[24] = dst; row[25] = op0; row[26] = op1;
        row[27] = fe_from_u64(d.off_dst); row[28] = fe_from_u64(d.off_op0); row[29] = fe_from_u64(d.off_op1);
        fe t0 = ((d.flags >> 9) & 1) ? dst : zero;
        row[30] = t0; row[31] = fe_mul(t0, res); row[32] = fe_mul(op0, op1);
        row[33] = (i + 1 == steps) ? zero : one;
    }
    if (rc_seg) {  // add_rc_builtin_columns (execution_trace.rs:358-379, :604-624)
        size_t k = 0;
        for (uint64_t a = rc_seg->start; a < rc_seg->end && k < steps; ++a, ++k) {
            const fe& v = mem_at(mem, a);
            fe raw = fe_from_mont(v);
            fe* row = &t[k * cols];
            for (int c = 0; c < 8; ++c) row[34 + c] = fe_from_u64((raw.v[c / 2] >> (16 * (c & 1))) & 0xffff);
            row[42] = v;
        }
    }
    // sorted addresses of the execution trace, before any padding (execution_trace.rs:64-67)
    std::vector<uint64_t> addrs;
    addrs.reserve(4 * steps);
    for (size_t i = 0; i < steps; ++i)
        for (int c = 19; c <= 22; ++c) addrs.push_back(fe_low_u64(t[i * cols + c]));
    std::sort(addrs.begin(), addrs.end());
    // get_rc_holes / fill_rc_holes (execution_trace.rs:136-185)
    {
        std::vector<uint16_t> offs;
        offs.reserve(3 * steps);
        for (size_t i = 0; i < steps; ++i)
            for (int c = 27; c <= 29; ++c) offs.push_back((uint16_t)fe_low_u64(t[i * cols + c]));
        std::sort(offs.begin(), offs.end());
        std::vector<uint16_t> missing;
        for (size_t i = 1; i < offs.size(); ++i)
            if (offs[i] != offs[i - 1])
                for (uint32_t v = (uint32_t)offs[i - 1] + 1; v < offs[i]; ++v) missing.push_back((uint16_t)v);
        size_t pad = ((missing.size() + 2) / 3) * 3 - missing.size();
        for (size_t i = 0; i < pad; ++i) missing.push_back(offs.back());
        pub.range_check_min = offs.front(); pub.range_check_max = offs.back();
        pub.has_rc_min = pub.has_rc_max = true;
        for (size_t i = 0; i < missing.size(); i += 3) {
            size_t base = t.size();
            t.resize(base + cols, zero);
            for (int k = 0; k < 3; ++k) t[base + 27 + k] = fe_from_u64(missing[i + k]);
        }
    }
    // get_memory_holes / fill_memory_holes (execution_trace.rs:195-255)
    {
        uint64_t codelen = pub.public_memory.size();
        std::vector<uint64_t> holes;
        uint64_t prev = addrs[0];
        for (uint64_t a : addrs) {
            uint64_t diff = a - prev;
            if (diff != 1 && diff != 0 && a > codelen)
                for (uint64_t h = prev + 1; h < a; ++h) if (h > codelen) holes.push_back(h);
            prev = a;
        }
        if (!holes.empty()) {
            std::vector<fe> last(t.end() - cols, t.end());
            size_t hi = 0;
            size_t rows = (holes.size() + 3) / 4;
            for (size_t r = 0; r < rows; ++r) {
                std::vector<fe> row = last;
                for (int c = 19; c <= 22; ++c) if (hi < holes.size()) row[c] = fe_from_u64(holes[hi++]);
                t.insert(t.end(), row.begin(), row.end());
            }
        }
    }
    // add_pub_memory_dummy_accesses (execution_trace.rs:91-96, :112-127)
    {
        std::vector<fe> last(t.end() - cols, t.end());
        for (int c = 19; c <= 26; ++c) last[c] = zero;
        size_t rows = (pub.public_memory.size() >> 2) + 1;
        for (size_t r = 0; r < rows; ++r) t.insert(t.end(), last.begin(), last.end());
    }
    // pad_with_last_row to the next power of two (execution_trace.rs:82-84)
    {
        size_t n = t.size() / cols, p2 = 1;
        while (p2 < n) p2 <<= 1;
        std::vector<fe> last(t.end() - cols, t.end());
        t.reserve(p2 * cols);
        for (size_t r = n; r < p2; ++r) t.insert(t.end(), last.begin(), last.end());
        *n_rows = p2;
    }
    *n_cols = cols;
    return t;
}

// cairo-run's non-proof-mode layout (reference src/cairo/runner/run.rs:64-240 drives cairo-vm 0.6.0 that way): program at
// 1..L, execution segment behind it, then one segment per builtin the program declares (layout order: output, range_check),
// then the two empty segments whose address is the return fp / end pc of main.  main's initial stack is
// [builtin bases ..., return fp, end pc]; it returns the advanced builtin pointers on top of the stack, which give the
// used length of every builtin segment (cairo-vm's read_return_values -> stop_ptr; run.rs:211-222 for range_check).
// The bases depend on the length of the execution segment, known only after the run: a first run with far-away
// placeholder bases measures the segments, the second one uses the real addresses (no hint-free program can branch on them).
void run_program_builtins(const std::vector<fe>& program, uint32_t builtins_mask, std::vector<RegisterState>& regs, CairoMemory& mem,
                          uint64_t max_steps, uint64_t entry_pc, std::vector<MemorySegment>& segments_out) {
    const uint64_t L = program.size();
    std::vector<uint8_t> kinds;   // segment types in stack order (air.rs:156-160: 0 RangeCheck, 1 Output)
    if (builtins_mask & 1u) kinds.push_back(1);
    if (builtins_mask & 2u) kinds.push_back(0);
    const size_t nb = kinds.size();
    std::vector<uint64_t> base(nb), used(nb, 0);
    for (size_t b = 0; b < nb; ++b) base[b] = (1ULL << 40) * (b + 1);
    uint64_t end_marker = ~0ULL >> 8;
    for (int pass = 0; pass < 2; ++pass) {
        mem.data.clear();
        regs.clear();
        for (uint64_t i = 0; i < L; ++i) mem.data[i + 1] = program[i];
        for (size_t b = 0; b < nb; ++b) mem.data[L + 1 + b] = fe_from_u64(base[b]);
        mem.data[L + 1 + nb] = fe_from_u64(end_marker);  // return fp
        mem.data[L + 2 + nb] = fe_from_u64(end_marker);  // return pc
        uint64_t pc = entry_pc, ap = L + 3 + nb, fp = L + 3 + nb;
        const uint64_t fp0 = fp;
        bool done = false;
        while (!done) {
            if (regs.size() >= max_steps) throw std::runtime_error("step limit exceeded");
            regs.push_back(RegisterState{ap, fp, pc});
            Decoded d = decode(mem_at(mem, pc));
            uint64_t size = d.op1_src == 1 ? 2 : 1;
            uint64_t dst_addr = add_signed(d.dst_reg ? fp : ap, d.off_dst);
            uint64_t op0_addr = add_signed(d.op0_reg ? fp : ap, d.off_op0);
            if (d.opcode == 1) {  // call: [ap] = fp, [ap+1] = pc + size
                mem.data[ap] = fe_from_u64(fp);
                mem.data[ap + 1] = fe_from_u64(pc + size);
            }
            const fe* op0p = mem.get(op0_addr);
            uint64_t op1_base;
            if (d.op1_src == 0) { if (!op0p) throw std::runtime_error("op0 unknown"); op1_base = fe_low_u64(*op0p); }
            else op1_base = d.op1_src == 1 ? pc : d.op1_src == 2 ? fp : ap;
            uint64_t op1_addr = add_signed(op1_base, d.off_op1);
            const fe* op1p = mem.get(op1_addr);
            const fe* dstp = mem.get(dst_addr);
            if (d.opcode == 4) {  // assert_eq with operand deduction
                if (!dstp) {
                    if (!op1p || (d.res_logic != 0 && !op0p)) throw std::runtime_error("cannot deduce dst");
                    fe r = d.res_logic == 0 ? *op1p : d.res_logic == 1 ? fe_add(*op0p, *op1p) : fe_mul(*op0p, *op1p);
                    mem.data[dst_addr] = r;
                } else if (!op1p) {
                    fe r;
                    if (d.res_logic == 0) r = *dstp;
                    else if (!op0p) throw std::runtime_error("cannot deduce op1");
                    else if (d.res_logic == 1) r = fe_sub(*dstp, *op0p);
                    else r = fe_mul(*dstp, fe_inv(*op0p));
                    mem.data[op1_addr] = r;
                } else if (!op0p && d.res_logic != 0) {
                    mem.data[op0_addr] = d.res_logic == 1 ? fe_sub(*dstp, *op1p) : fe_mul(*dstp, fe_inv(*op1p));
                }
                dstp = mem.get(dst_addr); op0p = mem.get(op0_addr); op1p = mem.get(op1_addr);
            }
            if (!dstp || !op1p) throw std::runtime_error("operand unknown");
            fe res = fe_zero();
            if (d.pc_update != 4) {
                if (d.res_logic != 0 && !op0p) throw std::runtime_error("op0 unknown");
                res = d.res_logic == 0 ? *op1p : d.res_logic == 1 ? fe_add(*op0p, *op1p) : fe_mul(*op0p, *op1p);
            }
            if (d.opcode == 4 && !fe_eq(*dstp, res)) throw std::runtime_error("assert_eq failed");
            uint64_t next_pc;
            switch (d.pc_update) {
                case 0: next_pc = pc + size; break;
                case 1: next_pc = (uint64_t)fe_to_i64(res); break;
                case 2: next_pc = pc + (uint64_t)fe_to_i64(res); break;
                default: next_pc = fe_is_zero(*dstp) ? pc + size : pc + (uint64_t)fe_to_i64(*op1p); break;
            }
            uint64_t next_ap = ap, next_fp = fp;
            if (d.ap_update == 1) next_ap = ap + (uint64_t)fe_to_i64(res);
            else if (d.ap_update == 2) next_ap = ap + 1;
            if (d.opcode == 1) { next_ap = ap + 2; next_fp = ap + 2; }
            else if (d.opcode == 2) {
                next_fp = fe_low_u64(*dstp);
                if (fp == fp0) done = true;  // main's own ret
            }
            pc = next_pc; ap = next_ap; fp = next_fp;
        }
        // main leaves the advanced builtin pointers on top of its stack (read_return_values)
        const uint64_t final_ap = regs.back().ap;
        for (size_t b = 0; b < nb; ++b) {
            const fe* stop = mem.get(final_ap - nb + b);
            if (!stop) throw std::runtime_error("builtin stop pointer missing from the final stack");
            const uint64_t sp = fe_low_u64(*stop);
            if (sp < base[b] || sp - base[b] > (1ULL << 32)) throw std::runtime_error("invalid builtin stop pointer");
            used[b] = sp - base[b];
        }
        uint64_t next = final_ap;   // the execution segment ends where ap ends
        for (size_t b = 0; b < nb; ++b) { base[b] = next; next += used[b]; }
        end_marker = next;
    }
    segments_out.clear();
    for (size_t b = 0; b < nb; ++b) {
        // every cell of a builtin segment must have been written; range-checked values lie in [0, 2^128)
        for (uint64_t a = base[b]; a < base[b] + used[b]; ++a) {
            const fe* v = mem.get(a);
            if (!v) throw std::runtime_error("hole in a builtin segment");
            if (kinds[b] == 0) {
                fe raw = fe_from_mont(*v);
                for (int k = 4; k < 8; ++k) if (raw.v[k]) throw std::runtime_error("range-check builtin: value out of [0, 2^128)");
            }
        }
        segments_out.push_back(MemorySegment{kinds[b], base[b], base[b] + used[b]});
    }
}

void run_program_plain(const std::vector<fe>& program, std::vector<RegisterState>& regs, CairoMemory& mem, uint64_t max_steps, uint64_t entry_pc) {
    std::vector<MemorySegment> none;
    run_program_builtins(program, 0, regs, mem, max_steps, entry_pc, none);
}

std::vector<fe> fibonacci_program(uint64_t fib_index) {
    static const char* words[22] = {
        "480680017fff8000", "1", "480680017fff8000", "1", "480680017fff8000", nullptr, "1104800180018000", "3",
        "208b7fff7fff7ffe", "20780017fff7ffd", "5", "480a7ffc7fff8000", "480a7ffc7fff8000", "208b7fff7fff7ffe",
        "482a7ffc7ffb8000", "480a7ffc7fff8000", "48127ffe7fff8000", "482680017ffd8000",
        "800000000000011000000000000000000000000000000000000000000000000",  // -1
        "1104800180018000",
        "800000000000010fffffffffffffffffffffffffffffffffffffffffffffff7",  // -10 (relative call offset)
        "208b7fff7fff7ffe"};
    std::vector<fe> prog(22);
    for (int i = 0; i < 22; ++i) {
        if (!words[i]) { prog[i] = fe_from_u64(fib_index); continue; }
        std::string h(words[i]);
        h = std::string(64 - h.size(), '0') + h;
        uint8_t b[32];
        for (int k = 0; k < 32; ++k) b[k] = (uint8_t)std::stoul(h.substr(2 * k, 2), nullptr, 16);
        prog[i] = fe_from_bytes_be(b);
    }
    return prog;
}

std::vector<uint8_t> serialize_public_inputs(const PublicInputs& p) {
    std::vector<uint8_t> b;
    auto u64 = [&](uint64_t v) { for (int i = 7; i >= 0; --i) b.push_back((uint8_t)(v >> (8 * i))); };
    auto felt = [&](const fe& x) { uint8_t t[32]; fe_to_bytes_be(x, t); b.insert(b.end(), t, t + 32); };
    u64(32);
    felt(p.pc_init); felt(p.ap_init); felt(p.fp_init); felt(p.pc_final); felt(p.ap_final);
    if (p.has_rc_min) { b.push_back(1); b.push_back((uint8_t)(p.range_check_min >> 8)); b.push_back((uint8_t)p.range_check_min); } else b.push_back(0);
    if (p.has_rc_max) { b.push_back(1); b.push_back((uint8_t)(p.range_check_max >> 8)); b.push_back((uint8_t)p.range_check_max); } else b.push_back(0);
    u64(p.memory_segments.size());
    for (auto& s : p.memory_segments) { b.push_back(s.type); u64(s.start); u64(s.end); }
    u64(p.public_memory.size());
    for (auto& kv : p.public_memory) { felt(fe_from_u64(kv.first)); felt(kv.second); }
    u64(p.num_steps);
    return b;
}

}  // namespace sp

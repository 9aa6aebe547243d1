// Host-side Cairo front-end pieces the prover boundary needs (all C++, no device work):
//   * PublicInputs                      reference src/cairo/air.rs:163-276
//   * binary .trace / .memory readers   reference src/cairo/register_states.rs:51-78, src/cairo/cairo_mem.rs:35-61
//   * instruction decode                reference src/cairo/decode/instruction_flags.rs:1-77, instruction_offsets.rs:18-56
//   * build_main_trace                  reference src/cairo/execution_trace.rs:57-87, :261-356 (+ helpers)
//   * a small Cairo VM for hint-free programs, with the output and range_check builtins (stands in for cairo-vm 0.6.0 that
//     reference src/cairo/runner/run.rs:64-240 drives; non-proof-mode layout)
#pragma once
#include "fp.h"
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace sp {

struct RegisterState { uint64_t ap, fp, pc; };

struct CairoMemory {
    std::unordered_map<uint64_t, fe> data;  // values in Montgomery form
    const fe* get(uint64_t addr) const { auto it = data.find(addr); return it == data.end() ? nullptr : &it->second; }
};

struct MemorySegment { uint8_t type; uint64_t start, end; };  // type 0 RangeCheck, 1 Output (air.rs:156-160)

struct PublicInputs {
    fe pc_init, ap_init, fp_init, pc_final, ap_final;
    bool has_rc_min = false, has_rc_max = false;
    uint16_t range_check_min = 0, range_check_max = 0;
    std::vector<MemorySegment> memory_segments;
    std::vector<std::pair<uint64_t, fe>> public_memory;  // (address, value); addresses are small integers
    uint64_t num_steps = 0;
    const MemorySegment* segment(uint8_t type) const {
        for (auto& s : memory_segments) if (s.type == type) return &s;
        return nullptr;
    }
};

// register_states.rs:51-78 / cairo_mem.rs:35-61. Return false on a malformed length.
bool parse_trace_le(const uint8_t* bytes, size_t len, std::vector<RegisterState>& out);
bool parse_memory_le(const uint8_t* bytes, size_t len, CairoMemory& out);

// air.rs:189-220 (range_check_min/max are filled by build_main_trace, as in the reference)
PublicInputs public_inputs_from_regs_and_mem(const std::vector<RegisterState>& regs, const CairoMemory& mem,
                                             size_t program_size, const std::vector<MemorySegment>& segments);

// execution_trace.rs:57-87. Returns the row-major n x cols main trace (cols = 34, or 43 with the rc builtin),
// n a power of two; sets pub.range_check_min/max. Throws std::runtime_error on undecodable instructions.
std::vector<fe> build_main_trace(const std::vector<RegisterState>& regs, const CairoMemory& mem, PublicInputs& pub,
                                 size_t* n_rows, size_t* n_cols);

// Runs `program` (field elements, address 1..L) from pc = 1 in cairo-run's non-proof-mode layout until main returns.
// Fills the relocated register trace and memory. Supports every hint-free, builtin-free instruction.
void run_program_plain(const std::vector<fe>& program, std::vector<RegisterState>& regs, CairoMemory& mem, uint64_t max_steps, uint64_t entry_pc = 1);
// The same for a program that declares builtins (builtins_mask: bit 0 output, bit 1 range_check): their base pointers are
// main's implicit arguments, main returns the advanced pointers; segments_out receives the used range of each builtin
// segment (what run.rs:211-222 reads back from cairo-vm).  Hint-free programs only; range-checked values must be < 2^128.
void run_program_builtins(const std::vector<fe>& program, uint32_t builtins_mask, std::vector<RegisterState>& regs, CairoMemory& mem,
                          uint64_t max_steps, uint64_t entry_pc, std::vector<MemorySegment>& segments_out);

// The 22-word fibonacci program of tests/golden/fibonacci_70000.proof with the index replaced by `fib_index`
// (fib(1, 1, fib_index), no final assert): 7*fib_index + 9 steps.
std::vector<fe> fibonacci_program(uint64_t fib_index);

// air.rs:223-276
std::vector<uint8_t> serialize_public_inputs(const PublicInputs& p);

}  // namespace sp

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
#include <functional>
#include <mutex>
#include <algorithm>
#include <vector>

namespace sp {

struct RegisterState { uint64_t ap, fp, pc; };

// Relocated Cairo memory, values in Montgomery form.  Addresses of a run are small consecutive integers (program, execution and
// builtin segments follow each other), so cells live in a flat array that grows with the highest address written; an address far
// beyond it (the placeholder bases of the first builtin pass, a hostile dump) goes to a hash map.
struct CairoMemory {
    static constexpr uint64_t ALWAYS_DENSE = 1ULL << 26;   // below this address a cell always lives in the flat array
    std::vector<fe> dense;
    std::vector<uint8_t> present;
    std::unordered_map<uint64_t, fe> sparse;
    const fe* get(uint64_t addr) const {
        if (addr < dense.size() && present[addr]) return &dense[addr];
        if (sparse.empty()) return nullptr;
        auto it = sparse.find(addr);
        return it == sparse.end() ? nullptr : &it->second;
    }
    void set(uint64_t addr, const fe& v) {
        if (addr >= dense.size()) {
            // the flat array follows contiguous growth (at most doubling per step); a far-away address goes to the map
            if (addr >= ALWAYS_DENSE && addr > 2 * (uint64_t)dense.size()) { sparse[addr] = v; return; }
            const uint64_t want = std::max<uint64_t>(std::max<uint64_t>(addr + 1, dense.size() * 2), 1024);
            dense.resize(want);
            present.resize(want, 0);
        }
        if (!sparse.empty()) sparse.erase(addr);
        dense[addr] = v; present[addr] = 1;
    }
    void clear() { dense.clear(); present.clear(); sparse.clear(); }
};

// The main trace in the device's own layout: column-major [cols][n] Montgomery field elements, in page-locked host memory when
// the HIP runtime can provide it (a column group is then one plain DMA - sp_cairo_prove_run), ordinary memory otherwise.
struct TraceColumns {
    fe* data = nullptr;
    size_t n_rows = 0, n_cols = 0;
    bool pinned = false;
    // A pageable table that try_pin() has copied into page-locked memory stays allocated until the run is freed: its address has
    // been handed out (sp_cairo_run_columns: "lives as long as the run") and another prover may still be reading or DMA-ing from it.
    fe* retired = nullptr;
    mutable std::mutex pin_mutex;              // try_pin() from two provers at once; current() from readers
    TraceColumns() = default;
    TraceColumns(const TraceColumns&) = delete;
    TraceColumns& operator=(const TraceColumns&) = delete;
    ~TraceColumns() { release(); }
    void allocate(size_t rows, size_t cols);   // throws std::bad_alloc
    bool try_pin();                            // copies a pageable table into page-locked memory (false: the runtime has none to give)
    const fe* current(bool* pinned_out = nullptr) const {   // the table to read from now on (both copies hold the same, immutable trace)
        std::lock_guard<std::mutex> lk(pin_mutex);
        if (pinned_out) *pinned_out = pinned;
        return data;
    }
    void release();
    fe& at(size_t row, size_t col) { return data[col * n_rows + row]; }
    const fe& at(size_t row, size_t col) const { return data[col * n_rows + row]; }
};

// Set by sp_ctx_create: host-only entry points never initialise the HIP runtime by themselves (a process may want PyTorch's
// copy of it to come first), so a run built before any context keeps its trace in ordinary memory.
void hip_runtime_mark_in_use();
bool hip_runtime_in_use();

// hardware threads, affinity mask and cgroup CPU quota taken together
unsigned host_effective_cpus();
// Ranks (processes with a context each, one per GPU) that share this host and its CPUs: sp_set_option(SP_OPT_HOST_RANKS), else the
// environment (SP_HOST_RANKS, then torch.distributed.run's LOCAL_WORLD_SIZE), else 1.  Every host-side thread count of the library
// (gather pool of the row-major upload, host_parallel_for) is sized from host_cpu_budget() = host_effective_cpus() / host_ranks(),
// at least 1; with more ranks than CPUs (host_oversubscribed) waits block instead of polling.
unsigned host_ranks();
void set_host_ranks(unsigned ranks);
unsigned host_cpu_budget();
bool host_oversubscribed();
// fn(begin, end) over [0, n) on up to SP_HOST_THREADS (default: host_cpu_budget(), at most 64) host threads
void host_parallel_for(size_t n, size_t min_chunk, const std::function<void(size_t, size_t)>& fn);

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

// What build_main_trace decides before it writes a single row (pass A): shape, extra rows, and that every cell it will read exists.
struct TracePlan {
    size_t steps = 0, cols = 0, n = 0;          // cols = 34, or 43 with the range-check builtin; n = rows, a power of two
    size_t r_rc = 0, r_holes = 0, r_dummy = 0;  // first row of: range-check holes, memory holes, public-memory dummies + padding
    std::vector<uint16_t> missing;              // unused offsets, three per row, padded with the largest one (execution_trace.rs:136-185)
    std::vector<uint64_t> holes;                // unused addresses, four per row (execution_trace.rs:195-255)
    uint64_t rc_start = 0, rc_count = 0;        // rows 0 .. rc_count-1 carry the range-checked values at rc_start + row (rc builtin)
    uint64_t mem_cells = 0;                     // 1 + the highest address any row reads
    bool dense = false;                         // all of them live in CairoMemory::dense: the device builder can index the memory
};
void plan_main_trace(const std::vector<RegisterState>& regs, const CairoMemory& mem, PublicInputs& pub, TracePlan& plan);
void fill_main_trace(const std::vector<RegisterState>& regs, const CairoMemory& mem, const TracePlan& plan, TraceColumns& out);

// Inputs of the device-side builder (csrc/trace_kernels.hip): [register states 24 B x steps][memory 32 B x mem_cells][missing
// offsets u16][memory holes u64], each part 256-byte aligned, in one host buffer.
struct TraceImage {
    uint8_t* base = nullptr;
    uint8_t* retired = nullptr;                // the pageable copy try_pin() replaced (kept until release(): it may be in use)
    uint64_t bytes = 0, off_regs = 0, off_mem = 0, off_missing = 0, off_holes = 0;
    bool pinned = false;
    mutable std::mutex pin_mutex;
    const uint8_t* current(bool* pinned_out = nullptr) const {   // what to upload from (both copies hold the same bytes)
        std::lock_guard<std::mutex> lk(pin_mutex);
        if (pinned_out) *pinned_out = pinned;
        return base;
    }
    TraceImage() = default;
    TraceImage(const TraceImage&) = delete;
    TraceImage& operator=(const TraceImage&) = delete;
    ~TraceImage() { release(); }
    void build(const std::vector<RegisterState>& regs, const CairoMemory& mem, const TracePlan& plan);   // nothing when !plan.dense
    bool try_pin();
    void release();
};

// execution_trace.rs:57-87. Builds the n x cols main trace (cols = 34, or 43 with the rc builtin), n a power of two, column-major
// in `out`; sets pub.range_check_min/max. Throws std::runtime_error on undecodable instructions.
void build_main_trace(const std::vector<RegisterState>& regs, const CairoMemory& mem, PublicInputs& pub, TraceColumns& out);

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
PublicInputs deserialize_public_inputs(const uint8_t* bytes, size_t len);   // air.rs:278-450; throws std::runtime_error("malformed: ...")

}  // namespace sp

// The Cairo main trace built on the device from the run's register states and memory.  Replaces, for callers that prove a run of the
// front-end, build_main_trace (reference src/cairo/execution_trace.rs:57-87) and its helpers: build_cairo_execution_trace
// (:261-356), update_values (:572-592), compute_res / the jnz inverse (:382-440), add_rc_builtin_columns (:358-379, :604-624),
// fill_rc_holes (:136-185), fill_memory_holes (:195-255), add_pub_memory_dummy_accesses (:91-127), pad_with_last_row (:82-84).
// What decides the SHAPE of the trace (and everything that can fail) stays on the host (plan_main_trace, cairo_host.cpp); the
// device writes the n x cols table - 1.1 GB at 2^20 rows - from 24 B per step and 32 B per memory cell.
#pragma once
#include "common.h"

namespace sp {

struct MainTraceArgs {
    const uint64_t* regs;       // [steps][ap, fp, pc]                       (register_states.rs:51-78)
    const fe* mem;              // [cells] Montgomery values, index = address (cairo_mem.rs:35-61)
    const uint16_t* missing;    // [3 * (r_holes - r_rc)] unused offsets
    const uint64_t* holes;      // [n_holes] unused addresses
    uint64_t steps, cells, n, r_rc, r_holes, r_dummy, n_holes;
    uint64_t rc_start, rc_count;
    uint32_t cols;              // 34 | 43
    fe* trace;                  // out: [cols][n], natural row order
};

// scratch: 2 * steps field elements + steps bytes (the jnz denominators, the batch inversion's prefixes, the row mask).
// *flag_dev is set when a row reads beyond `cells` (cannot happen for a plan the host validated).
static inline size_t main_trace_scratch_bytes(uint64_t steps) { return (size_t)steps * (2 * sizeof(fe) + 1) + 256; }
int cairo_main_trace_device(hipStream_t st, const MainTraceArgs& a, void* scratch, int* flag_dev);
// The same in pieces, for a caller that uploads the register states in chunks behind the memory: the rows of the steps [s0, s1) (needs
// the whole memory and regs[s0 .. s1)), then - once every step has its row - the jnz inverses, the builtin columns and the rows
// behind the steps.
int cairo_main_trace_steps(hipStream_t st, const MainTraceArgs& a, void* scratch, int* flag_dev, uint64_t s0, uint64_t s1);
int cairo_main_trace_finish(hipStream_t st, const MainTraceArgs& a, void* scratch, int* flag_dev);

}  // namespace sp

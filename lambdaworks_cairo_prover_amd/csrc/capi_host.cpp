// C ABI: host-only entry points (Cairo front-end, encoding helpers). See include/stark252_hip.h.
#include "../../include/stark252_hip.h"
#include "cairo_host.h"
#include "cairo_air_host.h"
#include "poseidon.h"
#include <array>
#include "common.h"
#include <cstring>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <exception>

struct sp_cairo_run {
    std::vector<sp::RegisterState> regs;
    sp::CairoMemory mem;
    sp::PublicInputs pub;
    sp::TracePlan plan;            // shape and extra rows of the main trace (pass A of build_main_trace: validated at run creation)
    sp::TraceImage image;          // register states + memory + hole lists in one buffer: what the device-side builder uploads
    // The n x cols table itself, column-major in the device layout (pinned when the HIP runtime provides it).  Built on first use:
    // sp_cairo_prove_run builds the trace ON THE DEVICE from `image` (70 MB instead of 1.1 GB at 2^20 rows), so a caller that only
    // proves never pays for the host table; sp_cairo_run_main_trace / sp_cairo_run_columns and the host-column proof path do.
    sp::TraceColumns main_trace;
    std::once_flag main_trace_once;
    std::exception_ptr main_trace_error;
    const sp::TraceColumns& host_trace() {
        std::call_once(main_trace_once, [this] {
            const auto t0 = std::chrono::steady_clock::now();
            try { sp::fill_main_trace(regs, mem, plan, main_trace); } catch (...) { main_trace_error = std::current_exception(); }
            t_table_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        });
        if (main_trace_error) std::rethrow_exception(main_trace_error);
        return main_trace;
    }
    size_t n_rows = 0, n_cols = 0;
    double t_vm_ms = 0, t_plan_ms = 0, t_image_ms = 0, t_table_ms = 0;   // sp_cairo_run_timings
    // flattened views handed out by sp_cairo_run_public_inputs
    std::vector<uint8_t> seg_types;
    std::vector<uint64_t> seg_ranges;
    std::vector<uint8_t> pm_bytes;
};

namespace sp {   // for capi_prove.cpp (sp_cairo_prove_run)
const PublicInputs& cairo_run_public_inputs(const sp_cairo_run* run) { return run->pub; }
const TraceColumns& cairo_run_columns(const sp_cairo_run* run) { return const_cast<sp_cairo_run*>(run)->host_trace(); }
// (nullptr when the memory of the run is not one flat array: the caller falls back to the host table)
bool cairo_run_device_inputs(const sp_cairo_run* run, const TracePlan** plan, TraceImage** image) {
    if (!run->image.base) return false;
    *plan = &run->plan; *image = &const_cast<sp_cairo_run*>(run)->image;
    return true;
}
}
static thread_local std::string g_last_error;
void sp_set_error(const std::string& s) { g_last_error = s; }
namespace sp {
int air_verify_host(const uint8_t* proof_bytes, size_t len, uint32_t main_cols, uint32_t aux_cols, const std::vector<uint32_t>& offsets,
                    const std::vector<uint32_t>& degrees, const std::vector<uint32_t>& exemptions, uint32_t bound_factor,
                    const std::vector<std::array<uint16_t, 3>>& ops, const std::vector<fe>& consts, uint32_t n_rap,
                    const std::vector<BoundaryConstraint>& boundary, uint8_t blowup, uint64_t queries, uint64_t coset_offset, uint8_t grinding);
}
namespace sp { void set_verify_merkle_backend(int backend); int host_bind_calling_thread_to_device_node(int device, int* node_out); }
namespace sp { int cairo_verify_host(const uint8_t* proof_bytes, size_t len, const PublicInputs& pub, uint8_t blowup, uint64_t queries, uint64_t coset_offset, uint8_t grinding); }

namespace sp {
// C view -> host PublicInputs, shared by the prover entry points and sp_cairo_verify so that both reject the same inputs.
PublicInputs public_inputs_from_c(const sp_cairo_public_inputs* p) {
    PublicInputs r;
    r.pc_init = fe_from_bytes_be(p->pc_init); r.ap_init = fe_from_bytes_be(p->ap_init); r.fp_init = fe_from_bytes_be(p->fp_init);
    r.pc_final = fe_from_bytes_be(p->pc_final); r.ap_final = fe_from_bytes_be(p->ap_final);
    r.has_rc_min = r.has_rc_max = true;
    r.range_check_min = p->range_check_min; r.range_check_max = p->range_check_max;
    for (uint32_t i = 0; i < p->n_segments; ++i)
        r.memory_segments.push_back({p->segment_types[i], p->segment_ranges[2 * i], p->segment_ranges[2 * i + 1]});
    for (uint64_t i = 0; i < p->n_public_memory; ++i) {
        fe a = fe_from_mont(fe_from_bytes_be(p->public_memory + 64 * i));
        for (int k = 2; k < 8; ++k) if (a.v[k]) throw std::runtime_error("public memory address does not fit in 64 bits");
        r.public_memory.push_back({(uint64_t)a.v[0] | ((uint64_t)a.v[1] << 32), fe_from_bytes_be(p->public_memory + 64 * i + 32)});
    }
    r.num_steps = p->num_steps;
    return r;
}
}  // namespace sp

extern "C" {

const char* sp_version(void) { return "stark252-hip 0.3 (gfx950)"; }
int sp_abi_version(void) { return SP_ABI_VERSION; }
uint64_t sp_air_desc_size(void) { return sizeof(sp_air_desc); }
const char* sp_last_error(void) { return g_last_error.c_str(); }
int sp_host_bind_to_device(int device, int* node_out) {
    if (device < 0) return SP_E_INVALID_ARG;
    return sp::host_bind_calling_thread_to_device_node(device, node_out);
}
int sp_host_cpus(int* count_out) {
    if (!count_out) return SP_E_INVALID_ARG;
    *count_out = (int)sp::host_effective_cpus();
    return SP_OK;
}

// ProofOptions::new_secure (reference src/starks/proof/options.rs:35-75)
int sp_proof_options_new_secure(int security_level, uint64_t coset_offset, sp_proof_options* out) {
    static const uint64_t queries[6] = {31, 41, 55, 80, 104, 140};
    if (!out || security_level < 0 || security_level > 5) return SP_E_INVALID_ARG;
    out->blowup_factor = 4; out->fri_number_of_queries = queries[security_level]; out->coset_offset = coset_offset; out->grinding_factor = 20;
    return SP_OK;
}

// new_with_checked_security (options.rs:78-102) / new_with_checked_provable_security (:107-129), check_field_security (:131-141)
int sp_proof_options_checked(uint8_t blowup_factor, uint64_t fri_number_of_queries, uint64_t coset_offset, uint8_t grinding_factor,
                             uint8_t security_target, int provable, uint32_t field_bits, sp_proof_options* out) {
    if (!out) return SP_E_INVALID_ARG;
    const uint64_t EXTENSION_DEGREE = 1, NUM_BITS_MAX_DOMAIN_SIZE = 40;
    if ((uint64_t)field_bits * EXTENSION_DEGREE <= (uint64_t)security_target + NUM_BITS_MAX_DOMAIN_SIZE) { sp_set_error("InsecureOptionError::FieldSize"); return SP_E_INVALID_ARG; }
    bool insecure;
    if (!provable) {
        const uint64_t bits = blowup_factor ? (uint64_t)__builtin_ctz((unsigned)blowup_factor) : 8;          // u8::trailing_zeros
        insecure = (uint64_t)security_target >= (uint64_t)grinding_factor + bits * fri_number_of_queries - 1;   // (wraps like the reference's usize only for 0 queries and grinding 0)
    } else {
        const uint64_t bits = blowup_factor ? (uint64_t)(__builtin_clz((unsigned)blowup_factor) - 24) : 8;   // u8::leading_zeros, as the reference has it
        insecure = (uint64_t)security_target < (uint64_t)grinding_factor + bits * fri_number_of_queries / 2;
    }
    if (insecure) { sp_set_error("InsecureOptionError::SecurityBits"); return SP_E_INVALID_ARG; }
    out->blowup_factor = blowup_factor; out->fri_number_of_queries = fri_number_of_queries; out->coset_offset = coset_offset; out->grinding_factor = grinding_factor;
    return SP_OK;
}

int sp_host_cpu_budget(int* budget_out, int* ranks_out) {
    if (!budget_out) return SP_E_INVALID_ARG;
    *budget_out = (int)sp::host_cpu_budget();
    if (ranks_out) *ranks_out = (int)sp::host_ranks();
    return SP_OK;
}

int sp_fe_to_device(int enc, const uint8_t* in, uint64_t n, uint8_t* out) {
    if (!in || !out) return SP_E_INVALID_ARG;
    fe* o = reinterpret_cast<fe*>(out);
    if (enc == SP_FE_CANON_BE) {
        for (uint64_t i = 0; i < n; ++i) o[i] = fe_from_bytes_be(in + 32 * i);
    } else if (enc == SP_FE_MONT_LIMBS) {
        for (uint64_t i = 0; i < n; ++i) { uint64_t l[4]; std::memcpy(l, in + 32 * i, 32); o[i] = fe_from_lw_limbs(l); }
    } else return SP_E_INVALID_ARG;
    return SP_OK;
}
int sp_fe_from_device(int enc, const uint8_t* in, uint64_t n, uint8_t* out) {
    if (!in || !out) return SP_E_INVALID_ARG;
    const fe* a = reinterpret_cast<const fe*>(in);
    if (enc == SP_FE_CANON_BE) {
        for (uint64_t i = 0; i < n; ++i) fe_to_bytes_be(a[i], out + 32 * i);
    } else if (enc == SP_FE_MONT_LIMBS) {
        for (uint64_t i = 0; i < n; ++i) { uint64_t l[4]; fe_to_lw_limbs(a[i], l); std::memcpy(out + 32 * i, l, 32); }
    } else return SP_E_INVALID_ARG;
    return SP_OK;
}

static int finish_run(sp_cairo_run* r, size_t program_size, sp_cairo_run** out, const std::vector<sp::MemorySegment>& segments = {}) {
    const auto t0 = std::chrono::steady_clock::now();
    r->pub = sp::public_inputs_from_regs_and_mem(r->regs, r->mem, program_size, segments);
    sp::plan_main_trace(r->regs, r->mem, r->pub, r->plan);      // shape, range_check_min / max, every check the fill relies on
    const auto t1 = std::chrono::steady_clock::now();
    r->image.build(r->regs, r->mem, r->plan);
    r->t_plan_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    r->t_image_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    r->n_rows = r->plan.n; r->n_cols = r->plan.cols;
    static const bool eager = std::getenv("SP_RUN_EAGER_TRACE") != nullptr;
    if (eager || !r->image.base) (void)r->host_trace();         // (no flat memory: the host table is the only form there is)
    *out = r;
    return SP_OK;
}

int sp_cairo_run_program_at(const uint8_t* words, uint64_t n_words, uint64_t entry_pc, uint64_t max_steps, sp_cairo_run** out) {
    if (!words || !out || n_words == 0 || entry_pc == 0 || entry_pc > n_words) return SP_E_INVALID_ARG;
    sp_cairo_run* r = new sp_cairo_run();
    try {
        std::vector<fe> prog(n_words);
        for (uint64_t i = 0; i < n_words; ++i) prog[i] = fe_from_bytes_be(words + 32 * i);
        { const auto tv = std::chrono::steady_clock::now(); sp::run_program_plain(prog, r->regs, r->mem, max_steps, entry_pc); r->t_vm_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv).count(); }
        return finish_run(r, n_words, out);
    } catch (const std::exception& e) { sp_set_error(e.what()); delete r; return SP_E_PROGRAM; }
}

int sp_cairo_run_program_builtins(const uint8_t* words, uint64_t n_words, uint64_t entry_pc, uint64_t max_steps, uint32_t builtins_mask,
                                  sp_cairo_run** out) {
    if (!words || !out || n_words == 0 || entry_pc == 0 || entry_pc > n_words || (builtins_mask & ~3u)) return SP_E_INVALID_ARG;
    sp_cairo_run* r = new sp_cairo_run();
    try {
        std::vector<fe> prog(n_words);
        for (uint64_t i = 0; i < n_words; ++i) prog[i] = fe_from_bytes_be(words + 32 * i);
        std::vector<sp::MemorySegment> segs;
        { const auto tv = std::chrono::steady_clock::now(); sp::run_program_builtins(prog, builtins_mask, r->regs, r->mem, max_steps, entry_pc, segs); r->t_vm_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv).count(); }
        return finish_run(r, n_words, out, segs);
    } catch (const std::exception& e) { sp_set_error(e.what()); delete r; return SP_E_PROGRAM; }
}

int sp_cairo_run_program(const uint8_t* words, uint64_t n_words, uint64_t max_steps, sp_cairo_run** out) {
    return sp_cairo_run_program_at(words, n_words, 1, max_steps, out);
}

int sp_cairo_run_fibonacci(uint64_t fib_index, sp_cairo_run** out) {
    if (!out) return SP_E_INVALID_ARG;
    sp_cairo_run* r = new sp_cairo_run();
    try {
        std::vector<fe> prog = sp::fibonacci_program(fib_index);
        { const auto tv = std::chrono::steady_clock::now(); sp::run_program_plain(prog, r->regs, r->mem, 7 * fib_index + 64); r->t_vm_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv).count(); }
        return finish_run(r, prog.size(), out);
    } catch (const std::exception& e) { sp_set_error(e.what()); delete r; return SP_E_PROGRAM; }
}

int sp_cairo_run_from_dumps(const uint8_t* trace, uint64_t trace_len, const uint8_t* memory, uint64_t memory_len,
                            uint64_t program_size, sp_cairo_run** out) {
    if (!trace || !memory || !out) return SP_E_INVALID_ARG;
    sp_cairo_run* r = new sp_cairo_run();
    try {
        if (!sp::parse_trace_le(trace, trace_len, r->regs) || r->regs.empty()) throw std::runtime_error("IncorrectNumberOfBytes (trace)");
        if (!sp::parse_memory_le(memory, memory_len, r->mem)) throw std::runtime_error("IncorrectNumberOfBytes (memory)");
        return finish_run(r, program_size, out);
    } catch (const std::exception& e) { sp_set_error(e.what()); delete r; return SP_E_PROGRAM; }
}

// generate_prover_args (reference src/cairo/runner/run.rs:242-263) from what cairo-vm hands the reference in memory - the relocated
// register states and memory of `run_program` (run.rs:64-240) and the builtin segments it reads back (run.rs:211-222) - without the
// detour through the binary dump files: what a Rust shim that keeps cairo-vm (hints, Cairo 1) calls.
int sp_cairo_run_from_arrays(const uint64_t* regs, uint64_t steps, const uint64_t* addrs, const uint8_t* values, int enc, uint64_t n_cells,
                             uint64_t program_size, const uint8_t* seg_types, const uint64_t* seg_ranges, uint32_t n_segments, sp_cairo_run** out) {
    if (!regs || !addrs || !values || !out || steps == 0 || n_cells == 0 || (n_segments && (!seg_types || !seg_ranges)) ||
        (enc != SP_FE_CANON_BE && enc != SP_FE_MONT_LIMBS)) return SP_E_INVALID_ARG;
    sp_cairo_run* r = new sp_cairo_run();
    try {
        r->regs.resize(steps);
        sp::host_parallel_for(steps, 1 << 16, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) r->regs[i] = sp::RegisterState{regs[3 * i], regs[3 * i + 1], regs[3 * i + 2]};
        });
        auto cell = [&](uint64_t k) {
            fe v;
            if (enc == SP_FE_CANON_BE) v = fe_from_bytes_be(values + 32 * k);
            else { uint64_t l[4]; std::memcpy(l, values + 32 * k, 32); v = fe_from_lw_limbs(l); }
            return v;
        };
        // a relocated memory is one run of addresses from 1 on: the flat array is sized once and filled by all threads (the wire
        // encoding costs a Montgomery product per cell); anything else goes cell by cell through CairoMemory::set
        uint64_t hi = 0;
        for (uint64_t k = 0; k < n_cells; ++k) hi = std::max(hi, addrs[k]);
        if (hi < 8 * n_cells + 1024 && hi + 1 < (1ULL << 32)) {
            r->mem.dense.assign(hi + 1, fe_zero());
            r->mem.present.assign(hi + 1, 0);
            // Cairo memory is write-once: the first thread to claim an address writes it, a second entry for the same address is
            // set aside and must carry the same value (checked once every first write has landed) - otherwise the winner would
            // depend on the thread schedule
            std::mutex dup_m;
            std::vector<uint64_t> dups;
            sp::host_parallel_for(n_cells, 1 << 14, [&](size_t b, size_t e) {
                for (size_t k = b; k < e; ++k) {
                    if (__atomic_exchange_n(&r->mem.present[addrs[k]], (uint8_t)1, __ATOMIC_ACQ_REL)) { std::lock_guard<std::mutex> lk(dup_m); dups.push_back(k); }
                    else r->mem.dense[addrs[k]] = cell(k);
                }
            });
            for (uint64_t k : dups)
                if (!fe_eq(r->mem.dense[addrs[k]], cell(k))) throw std::runtime_error("InconsistentMemory: two values for one address");
        } else {
            for (uint64_t k = 0; k < n_cells; ++k) {
                const fe v = cell(k);
                if (const fe* seen = r->mem.get(addrs[k])) { if (!fe_eq(*seen, v)) throw std::runtime_error("InconsistentMemory: two values for one address"); }
                else r->mem.set(addrs[k], v);
            }
        }
        std::vector<sp::MemorySegment> segs;
        for (uint32_t i = 0; i < n_segments; ++i) {
            if (seg_types[i] > 1 || seg_ranges[2 * i + 1] < seg_ranges[2 * i]) throw std::runtime_error("malformed memory segment");
            segs.push_back(sp::MemorySegment{seg_types[i], seg_ranges[2 * i], seg_ranges[2 * i + 1]});
        }
        return finish_run(r, program_size, out, segs);
    } catch (const std::exception& e) { sp_set_error(e.what()); delete r; return SP_E_PROGRAM; }
}

// The register states and the memory of a run as arrays (the inverse of sp_cairo_run_from_arrays; cells in increasing address order).
// Call with null pointers for the counts, then with buffers of steps x 3 and n_cells (x 32 bytes) entries.
int sp_cairo_run_export(const sp_cairo_run* run, int enc, uint64_t* steps_out, uint64_t* n_cells_out, uint64_t* regs_out, uint64_t* addrs_out, uint8_t* values_out) {
    if (!run || (enc != SP_FE_CANON_BE && enc != SP_FE_MONT_LIMBS)) return SP_E_INVALID_ARG;
    std::vector<uint64_t> addrs;
    for (uint64_t a = 0; a < run->mem.dense.size(); ++a) if (run->mem.present[a]) addrs.push_back(a);
    std::vector<uint64_t> far;
    for (auto& kv : run->mem.sparse) far.push_back(kv.first);
    std::sort(far.begin(), far.end());
    addrs.insert(addrs.end(), far.begin(), far.end());
    if (steps_out) *steps_out = run->regs.size();
    if (n_cells_out) *n_cells_out = addrs.size();
    if (regs_out)
        for (size_t i = 0; i < run->regs.size(); ++i) { regs_out[3 * i] = run->regs[i].ap; regs_out[3 * i + 1] = run->regs[i].fp; regs_out[3 * i + 2] = run->regs[i].pc; }
    if (addrs_out && values_out)
        for (size_t k = 0; k < addrs.size(); ++k) {
            addrs_out[k] = addrs[k];
            const fe& v = *run->mem.get(addrs[k]);
            if (enc == SP_FE_CANON_BE) fe_to_bytes_be(v, values_out + 32 * k);
            else { uint64_t l[4]; fe_to_lw_limbs(v, l); std::memcpy(values_out + 32 * k, l, 32); }
        }
    return SP_OK;
}

void sp_cairo_run_free(sp_cairo_run* run) { delete run; }

int sp_cairo_run_timings(const sp_cairo_run* run, double out[4]) {
    if (!run || !out) return SP_E_INVALID_ARG;
    out[0] = run->t_vm_ms; out[1] = run->t_plan_ms; out[2] = run->t_image_ms; out[3] = run->t_table_ms;
    return SP_OK;
}

int sp_cairo_run_shape(const sp_cairo_run* run, uint64_t* n_rows, uint32_t* n_cols, uint64_t* num_steps) {
    if (!run) return SP_E_INVALID_ARG;
    if (n_rows) *n_rows = run->n_rows;
    if (n_cols) *n_cols = (uint32_t)run->n_cols;
    if (num_steps) *num_steps = run->pub.num_steps;
    return SP_OK;
}

int sp_cairo_run_main_trace(const sp_cairo_run* run, int enc, uint8_t* out) {
    if (!run || !out || (enc != SP_FE_CANON_BE && enc != SP_FE_MONT_LIMBS)) return SP_E_INVALID_ARG;
    // row-major n x cols in the ABI encoding (what the reference's TraceTable holds, trace.rs:9-13) from the column-major store
    const sp::TraceColumns* Tp = nullptr;
    try { Tp = &const_cast<sp_cairo_run*>(run)->host_trace(); } catch (const std::exception& e) { sp_set_error(e.what()); return SP_E_PROGRAM; }
    const sp::TraceColumns& T = *Tp;
    sp::host_parallel_for(T.n_rows, 1024, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i)
            for (size_t c = 0; c < T.n_cols; ++c) {
                uint8_t* o = out + (i * T.n_cols + c) * 32;
                if (enc == SP_FE_CANON_BE) fe_to_bytes_be(T.at(i, c), o);
                else { uint64_t l[4]; fe_to_lw_limbs(T.at(i, c), l); std::memcpy(o, l, 32); }
            }
    });
    return SP_OK;
}

// The same trace as it is stored: column-major [cols][n_rows] field elements in the DEVICE layout (8 x u32 little-endian
// Montgomery limbs, SP_FE_DEVICE), page-locked when *pinned comes back 1.  The pointer lives as long as the run.
int sp_cairo_run_columns(const sp_cairo_run* run, const void** cols_out, uint64_t* n_rows, uint32_t* n_cols, int* pinned) {
    if (!run || !cols_out) return SP_E_INVALID_ARG;
    bool is_pinned = false;
    const sp::TraceColumns* Tp = nullptr;
    try { Tp = &const_cast<sp_cairo_run*>(run)->host_trace(); } catch (const std::exception& e) { sp_set_error(e.what()); return SP_E_PROGRAM; }
    *cols_out = Tp->current(&is_pinned);
    if (n_rows) *n_rows = Tp->n_rows;
    if (n_cols) *n_cols = (uint32_t)Tp->n_cols;
    if (pinned) *pinned = is_pinned ? 1 : 0;
    return SP_OK;
}

// Page-locked host memory for trace tables that will be handed to sp_cairo_prove / sp_cairo_prove_columns / sp_commit_trace:
// a buffer from here crosses PCIe by DMA without the staging copy a pageable buffer needs.
int sp_host_alloc(uint64_t bytes, void** out) {
    if (!out) return SP_E_INVALID_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {   // (initialises the HIP runtime: see INTEGRATION.md section 6 on PyTorch)
        (void)hipGetLastError();
        *out = nullptr;
        sp_set_error("sp_host_alloc: hipHostMalloc failed (" + std::to_string(bytes) + " bytes)");
        return SP_E_ALLOC;
    }
    return SP_OK;
}
void sp_host_free(void* p) { if (p) (void)hipHostFree(p); }

int sp_cairo_run_public_inputs(const sp_cairo_run* crun, sp_cairo_public_inputs* pi) {
    if (!crun || !pi) return SP_E_INVALID_ARG;
    sp_cairo_run* run = const_cast<sp_cairo_run*>(crun);
    const sp::PublicInputs& p = run->pub;
    fe_to_bytes_be(p.pc_init, pi->pc_init); fe_to_bytes_be(p.ap_init, pi->ap_init); fe_to_bytes_be(p.fp_init, pi->fp_init);
    fe_to_bytes_be(p.pc_final, pi->pc_final); fe_to_bytes_be(p.ap_final, pi->ap_final);
    pi->range_check_min = p.range_check_min; pi->range_check_max = p.range_check_max;
    run->seg_types.clear(); run->seg_ranges.clear();
    for (auto& s : p.memory_segments) { run->seg_types.push_back(s.type); run->seg_ranges.push_back(s.start); run->seg_ranges.push_back(s.end); }
    run->pm_bytes.resize(64 * p.public_memory.size());
    for (size_t i = 0; i < p.public_memory.size(); ++i) {
        fe_to_bytes_be(fe_from_u64(p.public_memory[i].first), &run->pm_bytes[64 * i]);
        fe_to_bytes_be(p.public_memory[i].second, &run->pm_bytes[64 * i + 32]);
    }
    pi->n_segments = (uint32_t)p.memory_segments.size();
    pi->segment_types = run->seg_types.data();
    pi->segment_ranges = run->seg_ranges.data();
    pi->n_public_memory = p.public_memory.size();
    pi->public_memory = run->pm_bytes.data();
    pi->num_steps = p.num_steps;
    return SP_OK;
}

// verify_cairo_proof (reference src/cairo/air.rs:1176-1182): 1 = accept, 0 = reject (also for malformed proofs).
int sp_cairo_verify(const uint8_t* proof, uint64_t proof_len, const sp_cairo_public_inputs* p, const sp_proof_options* opt) {
    if (!proof || !p || !opt) return SP_E_INVALID_ARG;
    try {
        sp::PublicInputs r = sp::public_inputs_from_c(p);
        const int ok = sp::cairo_verify_host(proof, proof_len, r, opt->blowup_factor, opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor);
        sp_set_error(ok == 1 ? "" : "rejected: a verification step failed");
        return ok;
    } catch (const std::exception& e) { sp_set_error(e.what()); return 0; }
}

namespace {
struct VerifyBackendScope {
    explicit VerifyBackendScope(int b) { sp::set_verify_merkle_backend(b); }
    ~VerifyBackendScope() { sp::set_verify_merkle_backend(SP_MERKLE_KECCAK256); }
};
}
int sp_cairo_verify_backend(const uint8_t* proof, uint64_t proof_len, const sp_cairo_public_inputs* p, const sp_proof_options* opt, int merkle_backend) {
    if (merkle_backend != SP_MERKLE_KECCAK256 && merkle_backend != SP_MERKLE_POSEIDON) return SP_E_INVALID_ARG;
    VerifyBackendScope scope(merkle_backend);
    return sp_cairo_verify(proof, proof_len, p, opt);
}

// verify::<Stark252PrimeField, A> (reference src/starks/verifier.rs:559-657) for a program AIR: 1 = accept, 0 = reject
// (also for malformed proofs or descriptors).
int sp_air_verify(const uint8_t* proof, uint64_t proof_len, const sp_air_desc* d, const sp_proof_options* opt) {
    if (!proof || !d || !opt) return SP_E_INVALID_ARG;
    try {
        if (d->n_offsets == 0 || d->n_offsets > 8 || d->n_transitions == 0 || d->n_transitions > 64 || (d->n_ops && !d->ops) ||
            (d->n_consts && !d->consts) || (d->n_boundary && !d->boundary)) { sp_set_error("malformed: AIR descriptor"); return 0; }
        std::vector<uint32_t> offsets(d->offsets, d->offsets + d->n_offsets), degrees(d->degrees, d->degrees + d->n_transitions),
            exemptions(d->exemptions, d->exemptions + d->n_transitions);
        std::vector<std::array<uint16_t, 3>> ops;
        for (uint32_t i = 0; i < d->n_ops; ++i) ops.push_back({d->ops[i].op, d->ops[i].a, d->ops[i].b});
        std::vector<fe> consts;
        for (uint32_t i = 0; i < d->n_consts; ++i) consts.push_back(fe_from_bytes_be(d->consts + 32 * (size_t)i));
        std::vector<sp::BoundaryConstraint> bcs;
        for (uint32_t i = 0; i < d->n_boundary; ++i) bcs.push_back(sp::BoundaryConstraint{d->boundary[i].col, d->boundary[i].step, fe_from_bytes_be(d->boundary[i].value)});
        const int ok = sp::air_verify_host(proof, proof_len, d->main_cols, d->aux_cols, offsets, degrees, exemptions, d->degree_bound_factor, ops, consts,
                                           d->n_rap, bcs, opt->blowup_factor, opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor);
        sp_set_error(ok == 1 ? "" : "rejected: a verification step failed");
        return ok;
    } catch (const std::exception& e) { sp_set_error(e.what()); return 0; }
}

// The `verify` command of the reference CLI (src/main.rs:113-143): u64_be(len(proof)) || proof || PublicInputs::serialize, parsed and
// handed to verify_cairo_proof.  1 = accepted; 0 otherwise, sp_last_error() as for sp_cairo_verify.
int sp_proof_file_verify_backend(const uint8_t* file, uint64_t file_len, const sp_proof_options* opt, int merkle_backend) {
    if (!file || !opt) return SP_E_INVALID_ARG;
    if (merkle_backend != SP_MERKLE_KECCAK256 && merkle_backend != SP_MERKLE_POSEIDON) return SP_E_INVALID_ARG;
    try {
        if (file_len < 8) throw std::runtime_error("malformed: InvalidAmountOfBytes (proof file)");
        uint64_t plen = 0;
        for (int i = 0; i < 8; ++i) plen = (plen << 8) | file[i];
        if (plen > file_len - 8) throw std::runtime_error("malformed: the proof length exceeds the file");
        const sp::PublicInputs pub = sp::deserialize_public_inputs(file + 8 + plen, (size_t)(file_len - 8 - plen));
        VerifyBackendScope scope(merkle_backend);
        const int ok = sp::cairo_verify_host(file + 8, (size_t)plen, pub, opt->blowup_factor, opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor);
        sp_set_error(ok == 1 ? "" : "rejected: a verification step failed");
        return ok;
    } catch (const std::exception& e) { sp_set_error(e.what()); return 0; }
}
int sp_proof_file_verify(const uint8_t* file, uint64_t file_len, const sp_proof_options* opt) {
    return sp_proof_file_verify_backend(file, file_len, opt, SP_MERKLE_KECCAK256);
}

// CLI proof file of the reference (src/main.rs:98-102): u64_be(len(proof)) || proof || PublicInputs::serialize.
// *out is malloc'd (sp_free). The public-memory order of the reference is HashMap order; this writer uses address order.
int sp_proof_file_encode(const uint8_t* proof, uint64_t proof_len, const sp_cairo_run* run, uint8_t** out, uint64_t* out_len) {
    if (!proof || !run || !out || !out_len) return SP_E_INVALID_ARG;
    std::vector<uint8_t> pi = sp::serialize_public_inputs(run->pub);
    uint64_t total = 8 + proof_len + pi.size();
    uint8_t* b = (uint8_t*)std::malloc(total);
    if (!b) return SP_E_ALLOC;
    for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(proof_len >> (56 - 8 * i));
    std::memcpy(b + 8, proof, proof_len);
    std::memcpy(b + 8 + proof_len, pi.data(), pi.size());
    *out = b; *out_len = total;
    return SP_OK;
}

int sp_air_verify_backend(const uint8_t* proof, uint64_t proof_len, const sp_air_desc* d, const sp_proof_options* opt, int merkle_backend) {
    if (merkle_backend != SP_MERKLE_KECCAK256 && merkle_backend != SP_MERKLE_POSEIDON) return SP_E_INVALID_ARG;
    VerifyBackendScope scope(merkle_backend);
    return sp_air_verify(proof, proof_len, d, opt);
}

// Host-side Poseidon (csrc/poseidon.h) for known-answer tests of the optional Merkle backend.
int sp_poseidon_host(int enc, int mode, const uint8_t* in, uint64_t n, uint8_t* out) {
    if (!in || !out || (enc != SP_FE_MONT_LIMBS && enc != SP_FE_CANON_BE) || mode < 0 || mode > 3) return SP_E_INVALID_ARG;
    if ((mode == 1 && n != 2) || (mode == 2 && n != 1) || (mode == 3 && n != 3) || n > (1u << 24)) return SP_E_INVALID_ARG;
    std::vector<fe> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i] = enc == SP_FE_CANON_BE ? fe_from_bytes_be(in + 32 * i) : fe_from_lw_limbs(reinterpret_cast<const uint64_t*>(in + 32 * i));
    fe r[3];
    int n_out = 1;
    if (mode == 0) r[0] = sp::poseidon_hash_many(v.data(), 1, (uint32_t)n);
    else if (mode == 1) r[0] = sp::poseidon_hash2(v[0], v[1]);
    else if (mode == 2) r[0] = sp::poseidon_hash1(v[0]);
    else { r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; sp::poseidon_permute(r[0], r[1], r[2]); for (int k = 0; k < 3; ++k) r[k] = fe_reduce_once(r[k]); n_out = 3; }
    for (int k = 0; k < n_out; ++k) {
        if (enc == SP_FE_CANON_BE) fe_to_bytes_be(r[k], out + 32 * k);
        else fe_to_lw_limbs(r[k], reinterpret_cast<uint64_t*>(out + 32 * k));
    }
    return SP_OK;
}

}  // extern "C"

// C ABI: round-level prover entry points and the whole-proof call. See include/stark252_hip.h.
#include "prover.h"
#include "cairo_host.h"
#include "prover_internal.h"
#include <cstdlib>
#include <cstring>

namespace sp {   // capi_host.cpp
const PublicInputs& cairo_run_public_inputs(const sp_cairo_run* run);
const TraceColumns& cairo_run_columns(const sp_cairo_run* run);
bool cairo_run_device_inputs(const sp_cairo_run* run, const TracePlan** plan, TraceImage** image);
}
using namespace sp;

namespace {
ProverHolder* holder(sp_ctx* c, bool create) { return prover_holder(c, create); }
int dec(sp_ctx* c, const uint8_t* in, uint64_t n, fe* out) { return sp_fe_to_device(c->enc, in, n, reinterpret_cast<uint8_t*>(out)); }
int enc(sp_ctx* c, const fe* in, uint64_t n, uint8_t* out) { return sp_fe_from_device(c->enc, reinterpret_cast<const uint8_t*>(in), n, out); }

PublicInputs to_host_pub(const sp_cairo_public_inputs* p) { return public_inputs_from_c(p); }
}  // namespace

extern "C" {

int sp_prove_setup(sp_ctx* c, uint64_t n, uint32_t main_cols, uint32_t aux_cols, int has_rc, const sp_proof_options* opt) {
    if (!c || !opt) return SP_E_INVALID_ARG;
    ProverHolder* h = holder(c, true);
    ProofOptionsHost o{opt->blowup_factor, opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor};
    return h->prover.setup(n, main_cols, aux_cols, has_rc != 0, o);
}

// A small VALID Cairo proof (fib(1100): 2^13 rows) through the four input forms on this context: every kernel family of the proof
// path - transforms, hashing, sorts and scans of the auxiliary trace, constraint check and composition, out-of-domain folds, DEEP,
// the FRI chain, grinding, openings, the decode / transpose kernels of the uploads - takes its first launch (code-object load,
// kernel-object set-up) here instead of inside the caller's first proof.
static int warm_small_proofs(sp_ctx* c, const ProofOptionsHost& o) {
    try {
        std::vector<fe> prog = fibonacci_program(1100);
        std::vector<RegisterState> regs;
        CairoMemory mem;
        run_program_plain(prog, regs, mem, 1u << 20);
        PublicInputs pub = public_inputs_from_regs_and_mem(regs, mem, prog.size(), {});
        TraceColumns T;
        TracePlan plan;
        plan_main_trace(regs, mem, pub, plan);
        fill_main_trace(regs, mem, plan, T);
        const uint64_t n = T.n_rows;
        const uint32_t cols = (uint32_t)T.n_cols;
        std::vector<uint8_t> bytes;
        float ms[5];
        // Three of the four input forms only as far as their commitment (round 1 of the main trace: what differs between them);
        // the fourth as a whole proof.
        CairoAirInfo air = cairo_air_info(pub);
        auto commit_only = [&](const uint8_t* p, StarkProver::TraceSource src, int col_enc, uint64_t col_stride) -> int {
            StarkProver* P = &prover_holder(c, true)->prover;
            SP_TRY(P->setup(n, air.main_columns, air.aux_columns, air.has_rc_builtin, o));
            uint8_t root[32];
            return P->commit_trace(0, p, cols, root, src, col_enc, col_stride);
        };
        // (1) host columns in the device layout: DMA per column group on the copy stream
        SP_TRY(commit_only(reinterpret_cast<const uint8_t*>(T.data), StarkProver::TRACE_HOST_COLUMNS, -1, n));
        // (2) host columns in the context encoding: the in-place decode in front of the transforms (what the row-major pipeline runs too)
        // (the two tables below in page-locked memory: a pageable buffer of this size is registered with the driver for the copy, and
        // giving it back to the system afterwards - the first munmap of a process - evicts the process' queues: the next launch
        // waited 20 - 28 ms, tools/prewarm_split.py)
        struct Pinned {
            uint8_t* p = nullptr;
            explicit Pinned(size_t bytes) { if (hipHostMalloc(reinterpret_cast<void**>(&p), bytes, hipHostMallocDefault) != hipSuccess) { p = nullptr; (void)hipGetLastError(); } }
            ~Pinned() { if (p) (void)hipHostFree(p); }
        };
        Pinned enc_cols((size_t)n * cols * 32), rows((size_t)n * cols * 32);
        if (!enc_cols.p || !rows.p) { sp_set_error("sp_prewarm: page-locked allocation failed"); return SP_E_ALLOC; }
        SP_TRY(sp_fe_from_device(c->enc, reinterpret_cast<const uint8_t*>(T.data), n * cols, enc_cols.p));
        SP_TRY(commit_only(enc_cols.p, StarkProver::TRACE_HOST_COLUMNS, c->enc, n));
        // (3) a row-major table (below the pipeline's threshold: one copy + rows -> columns, the kernel of sp_cairo_prove_dev)
        for (uint64_t i = 0; i < n; ++i)
            for (uint32_t j = 0; j < cols; ++j) std::memcpy(&rows.p[(i * cols + j) * 32], &enc_cols.p[((size_t)j * n + i) * 32], 32);
        SP_TRY(commit_only(rows.p, StarkProver::TRACE_HOST_ROWS, 0, 0));
        // (4) the run itself: register states + memory up, the table written by the device (sp_cairo_prove_run's default)
        TraceImage image;
        image.build(regs, mem, plan);
        if (image.base) {
            StarkProver::TraceBuildInput in{&plan, &image};
            SP_TRY(cairo_prove(c, reinterpret_cast<const uint8_t*>(&in), n, cols, pub, o, bytes, ms, StarkProver::TRACE_DEVICE_BUILD));
        } else
            SP_TRY(cairo_prove(c, reinterpret_cast<const uint8_t*>(T.data), n, cols, pub, o, bytes, ms, StarkProver::TRACE_HOST_COLUMNS, -1, n));
        return SP_OK;
    } catch (const std::exception& e) { sp_set_error(std::string("sp_prewarm: ") + e.what()); return SP_E_INVALID_ARG; }
}

int sp_prewarm(sp_ctx* c, uint64_t n, uint32_t main_cols, uint32_t aux_cols, int has_rc, const sp_proof_options* opt, uint32_t flags) {
    if (!c || !opt) return SP_E_INVALID_ARG;
    if (flags == 0) flags = SP_PREWARM_ALL;
    ProverHolder* h = holder(c, true);
    ProofOptionsHost o{opt->blowup_factor, opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor};
    // the shape first: ONE arena allocation of the real size (the small proofs below are carved out of it, nothing is re-allocated)
    sp_ctx* ctx = c;
    double _tp = wall_ms();
    SP_TRY(h->prover.setup(n, main_cols, aux_cols, has_rc != 0, o));
    SP_TIMEPOINT("prewarm: setup");
    // (the page-locked ring and its parked threads on a thread of their own, beside the launches below, made the prewarm 10 ms shorter
    // when nothing else ran and the first proof from a row-major table 4 - 14 ms slower in three runs of three: inline it is)
    SP_TRY(h->prover.warm_plumbing((flags & SP_PREWARM_HOST_ROWS) != 0));
    SP_TIMEPOINT("prewarm: streams, events, ring");
    if (flags & SP_PREWARM_KERNELS) {
        ProofOptionsHost small = o;
        small.fri_number_of_queries = std::min<uint64_t>(std::max<uint64_t>(o.fri_number_of_queries, 1), 64);
        small.grinding_factor = std::min<uint8_t>(o.grinding_factor, 16);
        SP_TRY(warm_small_proofs(c, small));
        SP_TIMEPOINT("prewarm: small proofs");
        SP_TRY(h->prover.setup(n, main_cols, aux_cols, has_rc != 0, o));
        SP_TIMEPOINT("prewarm: setup again");
    }
    int rc = SP_OK;
    if (flags & SP_PREWARM_CLOCKS) rc = h->prover.warm_round1();
    SP_TIMEPOINT("prewarm: round 1 at the real shape");
    c->prewarm_cancel.store(0, std::memory_order_release);      // (a request is spent on the call it reaches)
    return rc;
}

int sp_prewarm_cancel(sp_ctx* c) {
    if (!c) return SP_E_INVALID_ARG;
    c->prewarm_cancel.store(1, std::memory_order_release);
    return SP_OK;
}

int sp_commit_trace(sp_ctx* c, int segment, const uint8_t* rows, uint64_t n, uint32_t cols, uint8_t root_out[32]) {
    if (!c) return SP_E_INVALID_ARG;
    ProverHolder* h = holder(c, false);
    if (!h) { sp_set_error("sp_prove_setup not called"); return SP_E_STATE; }
    if (n != h->prover.n()) return SP_E_INVALID_ARG;
    return h->prover.commit_trace(segment, rows, cols, root_out);
}

int sp_commit_trace_columns(sp_ctx* c, int segment, const uint8_t* cols, uint64_t n, uint32_t n_cols, uint64_t col_stride, int device_layout,
                            uint8_t root_out[32]) {
    if (!c) return SP_E_INVALID_ARG;
    ProverHolder* h = holder(c, false);
    if (!h) { sp_set_error("sp_prove_setup not called"); return SP_E_STATE; }
    if (n != h->prover.n()) return SP_E_INVALID_ARG;
    return h->prover.commit_trace(segment, cols, n_cols, root_out, StarkProver::TRACE_HOST_COLUMNS, device_layout ? -1 : c->enc, col_stride);
}

int sp_cairo_commit_aux(sp_ctx* c, const uint8_t* rap, const sp_cairo_public_inputs* pub, uint8_t root_out[32]) {
    if (!c || !rap || !pub || !root_out) return SP_E_INVALID_ARG;
    ProverHolder* h = holder(c, false);
    if (!h) { sp_set_error("sp_prove_setup not called"); return SP_E_STATE; }
    try {
        fe r[3];
        SP_TRY(dec(c, rap, 3, r));
        PublicInputs p = to_host_pub(pub);
        return h->prover.commit_aux_cairo(p, r, root_out);
    } catch (const std::exception& e) { sp_set_error(e.what()); return SP_E_INVALID_ARG; }
}

int sp_composition(sp_ctx* c, const uint8_t* rap, const sp_boundary_constraint* bc, uint32_t nb, const uint8_t* coeffs, uint32_t T, uint8_t root_out[32]) {
    if (!c || !rap || (!bc && nb) || !coeffs || !root_out) return SP_E_INVALID_ARG;
    ProverHolder* h = holder(c, false);
    if (!h) { sp_set_error("sp_prove_setup not called"); return SP_E_STATE; }
    try {
        fe r[3];
        SP_TRY(dec(c, rap, 3, r));
        std::vector<BoundaryConstraint> bcs(nb);
        for (uint32_t j = 0; j < nb; ++j) { bcs[j].col = bc[j].col; bcs[j].step = bc[j].step; SP_TRY(dec(c, bc[j].value, 1, &bcs[j].value)); }
        std::vector<fe> all(2 * (size_t)(nb + T));
        SP_TRY(dec(c, coeffs, all.size(), all.data()));
        std::vector<fe> ba(all.begin(), all.begin() + nb), bb(all.begin() + nb, all.begin() + 2 * nb);
        std::vector<fe> ta(all.begin() + 2 * nb, all.begin() + 2 * nb + T), tb(all.begin() + 2 * nb + T, all.end());
        PublicInputs dummy;
        if (T == 50) dummy.memory_segments.push_back({0, 0, 0});
        CairoAirInfo air = cairo_air_info(dummy);
        if (air.num_transition_constraints != T) { sp_set_error("sp_composition: the Cairo AIR has 49 (50 with the range-check builtin) transition constraints"); return SP_E_UNSUPPORTED; }
        return h->prover.composition(r, bcs, ba, bb, ta, tb, air.transition_degrees, air.transition_exemptions, root_out);
    } catch (const std::exception& e) { sp_set_error(e.what()); return SP_E_INVALID_ARG; }
}

int sp_ood(sp_ctx* c, const uint8_t z[32], uint8_t* out) {
    if (!c || !z || !out) return SP_E_INVALID_ARG;
    ProverHolder* h = holder(c, false);
    if (!h) { sp_set_error("sp_prove_setup not called"); return SP_E_STATE; }
    fe zz, h1, h2;
    SP_TRY(dec(c, z, 1, &zz));
    std::vector<fe> tr;
    SP_TRY(h->prover.ood(zz, &h1, &h2, tr));
    SP_TRY(enc(c, &h1, 1, out));
    SP_TRY(enc(c, &h2, 1, out + 32));
    return enc(c, tr.data(), tr.size(), out + 64);
}

int sp_deep_fri_commit_begin(sp_ctx* c, const uint8_t* gammas, uint8_t root0_out[32]) {
    if (!c || !gammas || !root0_out) return SP_E_INVALID_ARG;
    ProverHolder* h = holder(c, false);
    if (!h) { sp_set_error("sp_prove_setup not called"); return SP_E_STATE; }
    std::vector<fe> g(2 + 2 * (size_t)h->prover.cols());
    SP_TRY(dec(c, gammas, g.size(), g.data()));
    std::vector<fe> tg(g.begin() + 2, g.end());
    return h->prover.deep_fri_begin(g[0], g[1], tg, root0_out);
}

int sp_fri_fold_commit(sp_ctx* c, const uint8_t zeta[32], uint8_t out[32], int* is_last) {
    if (!c || !zeta || !out || !is_last) return SP_E_INVALID_ARG;
    ProverHolder* h = holder(c, false);
    if (!h) { sp_set_error("sp_prove_setup not called"); return SP_E_STATE; }
    fe zt, last;
    SP_TRY(dec(c, zeta, 1, &zt));
    SP_TRY(h->prover.fri_fold_commit(zt, out, &last, is_last));
    if (*is_last) return enc(c, &last, 1, out);
    return SP_OK;
}

int sp_grind(sp_ctx* c, const uint8_t challenge[32], uint8_t factor, uint64_t* nonce_out) {
    if (!c || !challenge || !nonce_out) return SP_E_INVALID_ARG;
    ProverHolder* h = holder(c, false);   // never replaces a prover another entry point owns
    if (!h || !h->prover.ready()) { sp_set_error("sp_prove_setup not called"); return SP_E_STATE; }
    return h->prover.grind(challenge, factor, nonce_out);
}

int sp_open(sp_ctx* c, const uint64_t* iotas, uint32_t q, sp_openings* out) {
    if (!c || !iotas || !out) return SP_E_INVALID_ARG;
    ProverHolder* h = holder(c, false);
    if (!h) { sp_set_error("sp_prove_setup not called"); return SP_E_STATE; }
    std::vector<uint64_t> io(iotas, iotas + q);
    SP_TRY(h->prover.open(io, h->open));
    const Openings& o = h->open;
    auto encv = [&](const std::vector<fe>& v, std::vector<uint8_t>& dst) { dst.resize(v.size() * 32); return enc(c, v.data(), v.size(), dst.data()); };
    SP_TRY(encv(o.trace_evals, h->trace_evals)); SP_TRY(encv(o.comp_evals, h->comp_evals));
    SP_TRY(encv(o.fri_evals, h->fri_evals)); SP_TRY(encv(o.fri_evals_sym, h->fri_evals_sym));
    out->n_queries = o.n_queries; out->n_layers = o.n_layers; out->n_cols = o.n_cols; out->depth0 = o.depth0;
    out->trace_evals = h->trace_evals.data(); out->comp_evals = h->comp_evals.data();
    out->main_paths = reinterpret_cast<const uint8_t*>(o.main_paths.data());
    out->aux_paths = reinterpret_cast<const uint8_t*>(o.aux_paths.data());
    out->comp_paths = reinterpret_cast<const uint8_t*>(o.comp_paths.data());
    out->fri_evals = h->fri_evals.data(); out->fri_evals_sym = h->fri_evals_sym.data();
    out->fri_paths = reinterpret_cast<const uint8_t*>(o.fri_paths.data());
    out->fri_paths_sym = reinterpret_cast<const uint8_t*>(o.fri_paths_sym.data());
    return SP_OK;
}

static int cairo_prove_impl(sp_ctx* c, const uint8_t* main_trace, uint64_t n, uint32_t cols, const PublicInputs& p,
                            const sp_proof_options* opt, uint8_t** proof_out, uint64_t* proof_len, StarkProver::TraceSource src,
                            int col_enc = -1, uint64_t col_stride = 0);

int sp_cairo_prove(sp_ctx* c, const uint8_t* main_trace, uint64_t n, uint32_t cols, const sp_cairo_public_inputs* pub,
                   const sp_proof_options* opt, uint8_t** proof_out, uint64_t* proof_len) {
    if (!pub) return SP_E_INVALID_ARG;
    try { return cairo_prove_impl(c, main_trace, n, cols, to_host_pub(pub), opt, proof_out, proof_len, StarkProver::TRACE_HOST_ROWS); }
    catch (const std::exception& e) { sp_set_error(e.what()); return SP_E_INVALID_ARG; }
}
int sp_cairo_prove_dev(sp_ctx* c, const void* main_trace_dev, uint64_t n, uint32_t cols, const sp_cairo_public_inputs* pub,
                       const sp_proof_options* opt, uint8_t** proof_out, uint64_t* proof_len) {
    if (!pub) return SP_E_INVALID_ARG;
    try { return cairo_prove_impl(c, static_cast<const uint8_t*>(main_trace_dev), n, cols, to_host_pub(pub), opt, proof_out, proof_len, StarkProver::TRACE_DEVICE_ROWS); }
    catch (const std::exception& e) { sp_set_error(e.what()); return SP_E_INVALID_ARG; }
}
int sp_cairo_prove_columns(sp_ctx* c, const uint8_t* main_trace_cols, uint64_t n, uint32_t cols, uint64_t col_stride, int device_layout,
                           const sp_cairo_public_inputs* pub, const sp_proof_options* opt, uint8_t** proof_out, uint64_t* proof_len) {
    if (!pub || !c) return SP_E_INVALID_ARG;
    try {
        return cairo_prove_impl(c, main_trace_cols, n, cols, to_host_pub(pub), opt, proof_out, proof_len, StarkProver::TRACE_HOST_COLUMNS,
                                device_layout ? -1 : c->enc, col_stride);
    } catch (const std::exception& e) { sp_set_error(e.what()); return SP_E_INVALID_ARG; }
}
int sp_cairo_prove_run(sp_ctx* c, const sp_cairo_run* run, const sp_proof_options* opt, uint8_t** proof_out, uint64_t* proof_len) {
    if (!run || !c) return SP_E_INVALID_ARG;
    try {
        // The trace built on the device from the run's register states and memory (SP_OPT_DEVICE_TRACE, the default): 24 B per step and
        // 32 B per memory cell cross PCIe instead of the n x cols table, and the host never builds that table at all.
        const TracePlan* plan = nullptr;
        TraceImage* image = nullptr;
        if (c->opt_device_trace && cairo_run_device_inputs(run, &plan, &image)) {
            StarkProver::TraceBuildInput in{plan, image};
            return cairo_prove_impl(c, reinterpret_cast<const uint8_t*>(&in), plan->n, (uint32_t)plan->cols, cairo_run_public_inputs(run), opt, proof_out,
                                    proof_len, StarkProver::TRACE_DEVICE_BUILD);
        }
        // the host table (built on first use), column-major in the device layout.  A run built before this process had a context keeps
        // it in pageable memory: copied into page-locked memory once (the pageable copy stays valid until sp_cairo_run_free - its
        // address may have been handed out by sp_cairo_run_columns)
        const_cast<TraceColumns&>(cairo_run_columns(run)).try_pin();
        const TraceColumns& T = cairo_run_columns(run);
        return cairo_prove_impl(c, reinterpret_cast<const uint8_t*>(T.current()), T.n_rows, (uint32_t)T.n_cols, cairo_run_public_inputs(run), opt, proof_out,
                                proof_len, StarkProver::TRACE_HOST_COLUMNS, -1, T.n_rows);
    } catch (const std::exception& e) { sp_set_error(e.what()); return SP_E_PROGRAM; }
}

// build_main_trace on the device, for callers (and tests) that want the table itself: out = row-major n x cols in `fe_encoding`,
// what sp_cairo_run_main_trace produces on the host.
int sp_cairo_run_main_trace_dev(sp_ctx* c, const sp_cairo_run* run, int enc, uint8_t* out) {
    if (!c || !run || !out || (enc != SP_FE_CANON_BE && enc != SP_FE_MONT_LIMBS)) return SP_E_INVALID_ARG;
    const TracePlan* plan = nullptr;
    TraceImage* image = nullptr;
    if (!cairo_run_device_inputs(run, &plan, &image)) { sp_set_error("sp_cairo_run_main_trace_dev: the memory of this run is not one flat array"); return SP_E_UNSUPPORTED; }
    SP_HIP_CHECK(hipSetDevice(c->device));
    const TracePlan& P = *plan;
    struct Buf { void* p = nullptr; ~Buf() { if (p) (void)hipFree(p); } } img, table, enc_dev, flag;
    const size_t table_bytes = (size_t)P.n * P.cols * 32;
    if (hipMalloc(&img.p, image->bytes + main_trace_scratch_bytes(P.steps)) != hipSuccess || hipMalloc(&table.p, table_bytes) != hipSuccess ||
        hipMalloc(&enc_dev.p, table_bytes) != hipSuccess || hipMalloc(&flag.p, sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return SP_E_ALLOC; }
    uint8_t* stage = static_cast<uint8_t*>(img.p);
    SP_HIP_CHECK(hipMemcpyAsync(stage, image->current(), image->bytes, hipMemcpyHostToDevice, c->stream));
    SP_HIP_CHECK(hipMemsetAsync(flag.p, 0, sizeof(int), c->stream));
    MainTraceArgs a{};
    a.regs = reinterpret_cast<const uint64_t*>(stage + image->off_regs);
    a.mem = reinterpret_cast<const fe*>(stage + image->off_mem);
    a.missing = reinterpret_cast<const uint16_t*>(stage + image->off_missing);
    a.holes = reinterpret_cast<const uint64_t*>(stage + image->off_holes);
    a.steps = P.steps; a.cells = P.mem_cells; a.n = P.n; a.r_rc = P.r_rc; a.r_holes = P.r_holes; a.r_dummy = P.r_dummy; a.n_holes = P.holes.size();
    a.rc_start = P.rc_start; a.rc_count = P.rc_count; a.cols = (uint32_t)P.cols; a.trace = static_cast<fe*>(table.p);
    SP_TRY(cairo_main_trace_device(c->stream, a, stage + image->bytes, static_cast<int*>(flag.p)));
    SP_TRY(encode_elements(c->stream, enc, static_cast<const fe*>(table.p), (uint64_t)P.n * P.cols, static_cast<uint8_t*>(enc_dev.p)));
    std::vector<uint8_t> cols_host(table_bytes);
    int f = 0;
    SP_HIP_CHECK(hipMemcpyAsync(cols_host.data(), enc_dev.p, table_bytes, hipMemcpyDeviceToHost, c->stream));
    SP_HIP_CHECK(hipMemcpyAsync(&f, flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    SP_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (f) { sp_set_error("sp_cairo_run_main_trace_dev: a row reads beyond the memory image"); return SP_E_INVALID_ARG; }
    host_parallel_for(P.n, 1024, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i)
            for (size_t j = 0; j < P.cols; ++j) std::memcpy(out + (i * P.cols + j) * 32, &cols_host[(j * P.n + i) * 32], 32);
    });
    return SP_OK;
}

int sp_last_upload_stats(sp_ctx* c, double out[10]) {
    if (!c || !out) return SP_E_INVALID_ARG;
    std::memcpy(out, c->upload_stats, sizeof(double) * 10);
    return SP_OK;
}

static int cairo_prove_impl(sp_ctx* c, const uint8_t* main_trace, uint64_t n, uint32_t cols, const PublicInputs& p,
                            const sp_proof_options* opt, uint8_t** proof_out, uint64_t* proof_len, StarkProver::TraceSource src,
                            int col_enc, uint64_t col_stride) {
    if (!c || !main_trace || !opt || !proof_out || !proof_len) return SP_E_INVALID_ARG;
    c->prewarm_cancel.store(0, std::memory_order_release);   // (a sp_prewarm_cancel that found no prewarm to stop ends here, not in some later prewarm)
    try {
        ProofOptionsHost o{opt->blowup_factor, opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor};
        std::vector<uint8_t> bytes;
        float ms[5] = {0, 0, 0, 0, 0};
        int rc = cairo_prove(c, main_trace, n, cols, p, o, bytes, ms, src, col_enc, col_stride);
        if (rc != SP_OK) return rc;
        std::memcpy(c->round_ms, ms, sizeof(ms));
        *proof_out = (uint8_t*)std::malloc(bytes.size());
        if (!*proof_out) return SP_E_ALLOC;
        std::memcpy(*proof_out, bytes.data(), bytes.size());
        *proof_len = bytes.size();
        return SP_OK;
    } catch (const std::exception& e) { sp_set_error(e.what()); return SP_E_INVALID_ARG; }
}

void sp_free(void* p) { std::free(p); }

int sp_last_proof_info(sp_ctx* c, uint32_t out[4]) {
    if (!c || !out) return SP_E_INVALID_ARG;
    std::memcpy(out, c->proof_info, sizeof(uint32_t) * 4);
    return SP_OK;
}

int sp_prover_device_bytes(sp_ctx* c, uint64_t* bytes_out) {
    if (!c || !bytes_out) return SP_E_INVALID_ARG;
    *bytes_out = c->prover_device_bytes;
    return SP_OK;
}

int sp_last_round_ms(sp_ctx* c, float out[5]) {
    if (!c || !out) return SP_E_INVALID_ARG;
    std::memcpy(out, c->round_ms, sizeof(float) * 5);
    return SP_OK;
}

int sp_air_prove(sp_ctx* c, const sp_air_desc* d, const uint8_t* main_trace, uint64_t n, const sp_proof_options* opt,
                 uint8_t** proof_out, uint64_t* proof_len) {
    if (!c || !d || !main_trace || !opt || !proof_out || !proof_len) return SP_E_INVALID_ARG;
    c->prewarm_cancel.store(0, std::memory_order_release);
    if (d->n_offsets == 0 || d->n_offsets > 8 || d->n_transitions == 0 || d->n_transitions > 64 || (d->n_ops && !d->ops) ||
        (d->n_consts && !d->consts) || (d->n_boundary && !d->boundary)) { sp_set_error("sp_air_prove: malformed descriptor"); return SP_E_INVALID_ARG; }
    sp::AirDescHost a;
    a.main_cols = d->main_cols; a.aux_cols = d->aux_cols;
    a.offsets.assign(d->offsets, d->offsets + d->n_offsets);
    a.degrees.assign(d->degrees, d->degrees + d->n_transitions);
    a.exemptions.assign(d->exemptions, d->exemptions + d->n_transitions);
    a.num_transition_exemptions = d->num_transition_exemptions;
    a.degree_bound_factor = d->degree_bound_factor;
    for (uint32_t i = 0; i < d->n_ops; ++i) a.ops.push_back(sp::AirOpHost{d->ops[i].op, d->ops[i].a, d->ops[i].b});
    for (uint32_t i = 0; i < d->n_consts; ++i) a.consts.push_back(fe_from_bytes_be(d->consts + 32 * (size_t)i));
    a.n_rap = d->n_rap; a.aux_kind = d->aux_kind; a.aux_fn = d->aux_fn; a.aux_user = d->aux_user;
    for (uint32_t i = 0; i < d->n_boundary; ++i)
        a.boundary.push_back(sp::BoundaryConstraint{d->boundary[i].col, d->boundary[i].step, fe_from_bytes_be(d->boundary[i].value)});
    sp::ProofOptionsHost o{opt->blowup_factor, opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor};
    std::vector<uint8_t> proof;
    int rc = sp::air_prove(c, a, main_trace, n, o, proof);
    if (rc != SP_OK) return rc;
    uint8_t* out = (uint8_t*)std::malloc(proof.size());
    if (!out) return SP_E_ALLOC;
    std::memcpy(out, proof.data(), proof.size());
    *proof_out = out; *proof_len = proof.size();
    return SP_OK;
}

}  // extern "C"

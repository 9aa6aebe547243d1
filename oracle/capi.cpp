// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/fp.hpp header).
// extern "C" surface of the CPU oracle for ctypes (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
// All field elements cross this boundary as canonical 32-byte big-endian (the reference wire format).
#include "fp.hpp"
#include "keccak.hpp"
#include "poly.hpp"
#include "merkle.hpp"
#include "air.hpp"
#include "cairo_air.hpp"
#include "stark.hpp"
#include "example_airs.hpp"
#include <memory>
#include <omp.h>
#include <cstdlib>
#include <cstdio>

using namespace oracle;

static std::vector<Fp> load_felts(const uint8_t* in, size_t n) {
    std::vector<Fp> v(n);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) v[i] = Fp::from_bytes_be(in + 32 * (size_t)i);
    return v;
}
static void store_felts(const std::vector<Fp>& v, uint8_t* out) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)v.size(); ++i) v[i].to_bytes_be(out + 32 * (size_t)i);
}

extern "C" {

void oracle_keccak256(const uint8_t* data, uint64_t len, uint8_t out[32]) { keccak256(data, len, out); }

// op: 0 add, 1 sub, 2 mul, 3 inv(a), 4 pow(a, b as u64 in the low 8 bytes), 5 neg(a)
int oracle_fe_op(int op, const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
    try {
        Fp x = Fp::from_bytes_be(a), y = Fp::from_bytes_be(b), r;
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            case 3: r = x.inv(); break;
            case 4: { uint64_t e = 0; for (int i = 24; i < 32; ++i) e = (e << 8) | b[i]; r = x.pow(e); break; }
            case 5: r = -x; break;
            default: return -1;
        }
        r.to_bytes_be(out);
        return 0;
    } catch (...) { return -2; }
}

void oracle_primitive_root(uint32_t order, uint8_t out[32]) { primitive_root(order).to_bytes_be(out); }

// In-place batch inverse of n elements. Returns -2 if any element is zero (lambdaworks panics).
int oracle_batch_inverse(uint8_t* data, uint64_t n) {
    try { auto v = load_felts(data, n); batch_inverse(v); store_felts(v, data); return 0; } catch (...) { return -2; }
}

// Natural-order DFT of n = 2^k elements, in place.  inverse != 0: inverse DFT (includes 1/n).
// coset (nullable): forward -> evaluate on coset*<w> (coefficient k scaled by coset^k first);
//                   inverse -> interpolate from coset evaluations (coefficient k scaled by coset^-k after).
int oracle_ntt(uint8_t* data, uint64_t n, int inverse, const uint8_t* coset) {
    // in place: the caller's buffer holds n elements, and the reference's evaluate_fft answers a length that is not a power of two
    // with next_power_of_two(n) evaluations (interpolate_fft refuses it)
    if (n == 0 || (n & (n - 1))) return -1;
    try {
        std::vector<Fp> a = load_felts(data, n);
        if (!inverse) {
            Fp off = coset ? Fp::from_bytes_be(coset) : Fp::one();
            Poly p(a.begin(), a.end());
            std::vector<Fp> ev = evaluate_offset_fft(p, 1, n, off);
            store_felts(ev, data);
        } else {
            Poly c = coset ? interpolate_offset_fft(a, Fp::from_bytes_be(coset)) : interpolate_fft(a);
            c.resize(n, Fp::zero());
            store_felts(c, data);
        }
        return 0;
    } catch (...) { return -2; }
}

// bench.py cpu_baseline: `vectors` forward NTTs of the same n elements, one vector per OpenMP thread (the reference runs
// its column transforms under rayon the same way, prover.rs:171-172).  The 32-byte big-endian codec is timed apart:
// out[0] = seconds to decode + encode one vector (1 thread), out[1] = wall seconds of all transforms, out[2] = threads.
int oracle_ntt_bench(const uint8_t* data, uint64_t n, uint32_t vectors, uint32_t threads, double out[3]) {
    try {
        double t0 = omp_get_wtime();
        std::vector<Fp> a = load_felts(data, n);
        std::vector<uint8_t> back(n * 32);
        store_felts(a, back.data());
        out[0] = omp_get_wtime() - t0;
        Poly p(a.begin(), a.end());
        if (threads == 0) threads = 1;
        std::vector<Fp> keep(vectors);
        double t1 = omp_get_wtime();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (int64_t v = 0; v < (int64_t)vectors; ++v) {
            std::vector<Fp> ev = evaluate_offset_fft(p, 1, n, Fp::one());
            keep[v] = ev[(size_t)v % n];
        }
        out[1] = omp_get_wtime() - t1;
        out[2] = (double)threads;
        return keep.empty() ? -1 : 0;   // (keep: the transforms cannot be optimised away)
    } catch (...) { return -2; }
}

// `evaluate_polynomial_on_lde_domain` of one column: n coefficients -> n*blowup evaluations on coset*<w_N>
int oracle_lde(const uint8_t* coeffs, uint64_t n, uint32_t blowup, const uint8_t coset[32], uint8_t* out) {
    if (n == 0 || (n & (n - 1)) || blowup == 0 || (blowup & (blowup - 1))) return -1;      // (out holds n x blowup elements)
    try {
        std::vector<Fp> a = load_felts(coeffs, n);
        Poly p(a.begin(), a.end()); trim(p);
        std::vector<Fp> ev = evaluate_polynomial_on_lde_domain(p, blowup, n, Fp::from_bytes_be(coset));
        store_felts(ev, out);
        return 0;
    } catch (...) { return -2; }
}

// Merkle tree over n_leaves rows of fe_per_leaf elements (row-major). nodes_out (nullable): (2n-1)*32 bytes.
int oracle_merkle_build(const uint8_t* rows, uint64_t n_leaves, uint32_t fe_per_leaf, uint8_t root_out[32], uint8_t* nodes_out) {
    try {
        std::vector<Fp> v = load_felts(rows, n_leaves * fe_per_leaf);
        MerkleTree t = MerkleTree::build_batched(v.data(), n_leaves, fe_per_leaf, fe_per_leaf == 1);
        std::memcpy(root_out, t.root.data(), 32);
        if (nodes_out) for (size_t i = 0; i < t.nodes.size(); ++i) std::memcpy(nodes_out + 32 * i, t.nodes[i].data(), 32);
        return 0;
    } catch (...) { return -2; }
}

// Transcript replay helper: feed `n_ops` operations; op kinds: 0 append(bytes), 1 challenge -> 32 bytes,
// 2 to_field -> 32 bytes, 3 to_usize -> 8 bytes BE.
struct OracleTranscript { Transcript t; };
// Merkle backend of every later call: 0 Keccak256 (reference), 1 Starknet Poseidon (merkle.hpp)
int oracle_set_merkle_backend(int backend) {
    if (backend != 0 && backend != 1) return -1;
    merkle_backend() = backend;
    return 0;
}
// Poseidon known answers: mode 0 hash_many(in[0..n)), 1 hash(in[0], in[1]), 2 hash_single(in[0]), 3 permutation of in[0..3) (96 bytes out)
int oracle_poseidon(int mode, const uint8_t* in, uint64_t n, uint8_t* out) {
    std::vector<Fp> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i] = Fp::from_bytes_be(in + 32 * i);
    const Poseidon& ps = Poseidon::get();
    if (mode == 0) ps.hash_many(v.data(), n).to_bytes_be(out);
    else if (mode == 1 && n == 2) ps.hash(v[0], v[1]).to_bytes_be(out);
    else if (mode == 2 && n == 1) ps.hash_single(v[0]).to_bytes_be(out);
    else if (mode == 3 && n == 3) { Fp s[3] = {v[0], v[1], v[2]}; ps.permute(s); for (int k = 0; k < 3; ++k) s[k].to_bytes_be(out + 32 * k); }
    else return -1;
    return 0;
}

void* oracle_transcript_new() { return new OracleTranscript(); }
void oracle_transcript_free(void* h) { delete (OracleTranscript*)h; }
void oracle_transcript_append(void* h, const uint8_t* d, uint64_t n) { ((OracleTranscript*)h)->t.append(d, n); }
void oracle_transcript_challenge(void* h, uint8_t out[32]) { Digest d = ((OracleTranscript*)h)->t.challenge(); std::memcpy(out, d.data(), 32); }
void oracle_transcript_to_field(void* h, uint8_t out[32]) { ((OracleTranscript*)h)->t.to_field().to_bytes_be(out); }
uint64_t oracle_transcript_to_usize(void* h) { return ((OracleTranscript*)h)->t.to_usize(); }

uint64_t oracle_grinding_nonce(const uint8_t challenge[32], uint8_t factor) {
    Digest d; std::memcpy(d.data(), challenge, 32);
    return grinding_nonce(d, factor);
}

// ---- Cairo public inputs as a flat struct (mirrors reference src/cairo/air.rs:163-181)
struct oracle_cairo_public_inputs {
    uint8_t pc_init[32], ap_init[32], fp_init[32], pc_final[32], ap_final[32];
    uint16_t range_check_min, range_check_max;
    uint32_t n_segments;              // memory_segments
    const uint8_t* segment_types;     // 0 = RangeCheck, 1 = Output
    const uint64_t* segment_ranges;   // start,end pairs
    uint64_t n_public_memory;
    const uint8_t* public_memory;     // (address, value) pairs, 64 bytes each
    uint64_t num_steps;
};

static PublicInputs to_pi(const oracle_cairo_public_inputs* c) {
    PublicInputs p;
    p.pc_init = Fp::from_bytes_be(c->pc_init); p.ap_init = Fp::from_bytes_be(c->ap_init);
    p.fp_init = Fp::from_bytes_be(c->fp_init); p.pc_final = Fp::from_bytes_be(c->pc_final);
    p.ap_final = Fp::from_bytes_be(c->ap_final);
    p.has_rc_min = p.has_rc_max = true;
    p.rc_min = c->range_check_min; p.rc_max = c->range_check_max;
    for (uint32_t i = 0; i < c->n_segments; ++i)
        p.memory_segments.push_back({c->segment_types[i], c->segment_ranges[2 * i], c->segment_ranges[2 * i + 1]});
    for (uint64_t i = 0; i < c->n_public_memory; ++i)
        p.public_memory.push_back({Fp::from_bytes_be(c->public_memory + 64 * i), Fp::from_bytes_be(c->public_memory + 64 * i + 32)});
    p.num_steps = c->num_steps;
    return p;
}

struct oracle_proof_options { uint8_t blowup_factor; uint64_t fri_number_of_queries; uint64_t coset_offset; uint8_t grinding_factor; };

// `generate_cairo_proof` (reference src/cairo/air.rs:1165-1171) + `Serializable::serialize`.
// main_trace: row-major n x main_cols canonical BE.  proof_out is malloc'd; free with oracle_free.
// timings_out (nullable): 4 doubles, seconds per round 1..4.
int oracle_cairo_prove(const uint8_t* main_trace, uint64_t n, uint32_t main_cols, const oracle_cairo_public_inputs* pub,
                       const oracle_proof_options* opt, int legacy_boundary, uint8_t** proof_out, uint64_t* proof_len,
                       double* timings_out) {
    try {
        PublicInputs pi = to_pi(pub);
        ProofOptions o{opt->blowup_factor, (size_t)opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor};
        CairoAir air(n, pi, o);
        std::vector<Fp> tr = load_felts(main_trace, n * main_cols);
        Prover pr(air, legacy_boundary != 0);
        StarkProof proof = pr.prove(tr, main_cols);
        std::vector<uint8_t> bytes = serialize_proof(proof);
        *proof_out = (uint8_t*)std::malloc(bytes.size());
        std::memcpy(*proof_out, bytes.data(), bytes.size());
        *proof_len = bytes.size();
        if (timings_out) { timings_out[0] = pr.timings.round1; timings_out[1] = pr.timings.round2; timings_out[2] = pr.timings.round3; timings_out[3] = pr.timings.round4; }
        return 0;
    } catch (const std::exception& e) { std::fprintf(stderr, "oracle_cairo_prove: %s\n", e.what()); return -2; }
}

// `verify_cairo_proof` (reference src/cairo/air.rs:1176-1182). Returns 1 accept, 0 reject, <0 malformed.
int oracle_cairo_verify(const uint8_t* proof, uint64_t proof_len, const oracle_cairo_public_inputs* pub, const oracle_proof_options* opt) {
    try {
        StarkProof p = deserialize_proof(proof, proof_len);
        PublicInputs pi = to_pi(pub);
        ProofOptions o{opt->blowup_factor, (size_t)opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor};
        if (p.trace_length == 0 || (p.trace_length & (p.trace_length - 1)) || p.trace_length > (1ULL << 30)) return 0;
        CairoAir air(p.trace_length, pi, o);
        return verify(air, p) ? 1 : 0;
    } catch (...) { return -3; }
}

// Auxiliary trace only (reference src/cairo/air.rs:660-729): rap = 3 felts; out = n x 18 row-major.
int oracle_cairo_aux_trace(const uint8_t* main_trace, uint64_t n, uint32_t main_cols, const oracle_cairo_public_inputs* pub,
                           const uint8_t* rap, uint8_t* out) {
    try {
        PublicInputs pi = to_pi(pub);
        CairoAir air(n, pi, ProofOptions::default_test_options());
        std::vector<Fp> tr = load_felts(main_trace, n * main_cols);
        std::vector<Fp> r = load_felts(rap, 3);
        store_felts(air.build_auxiliary_trace(tr, main_cols, r), out);
        return 0;
    } catch (...) { return -2; }
}

// Transition constraints on one 2-row frame (reference src/cairo/air.rs:743-767): frame = 2 x cols, out = 49|50.
int oracle_cairo_transition(const uint8_t* frame, uint32_t cols, int has_rc_builtin, const uint8_t* rap, uint8_t* out) {
    try {
        PublicInputs pi;
        if (has_rc_builtin) pi.memory_segments.push_back({0, 0, 0});
        CairoAir air(2, pi, ProofOptions::default_test_options());
        if (air.ctx.trace_columns != cols) return -1;
        std::vector<Fp> f = load_felts(frame, 2 * cols), r = load_felts(rap, 3), c(air.ctx.num_transition_constraints);
        air.compute_transition(f.data(), r, c.data());
        store_felts(c, out);
        return 0;
    } catch (...) { return -2; }
}

void oracle_free(void* p) { std::free(p); }


// ---- example AIRs of the reference (src/starks/example) and the "program" AIR -----------------------------------------
// kind: 0 simple_fibonacci, 1 fibonacci_2_columns, 2 quadratic, 3 fibonacci_rap, 4 dummy.
// params: up to two 32-byte BE field elements (a0, a1 / a0); steps only for fibonacci_rap.
static std::unique_ptr<Air> make_example(int kind, size_t n, const uint8_t* params, uint64_t steps, const ProofOptions& o) {
    Fp a0 = params ? Fp::from_bytes_be(params) : Fp::one();
    Fp a1 = params ? Fp::from_bytes_be(params + 32) : Fp::one();
    switch (kind) {
        case 0: return std::unique_ptr<Air>(new FibonacciAir(n, a0, a1, o));
        case 1: return std::unique_ptr<Air>(new Fibonacci2ColsAir(n, a0, a1, o));
        case 2: return std::unique_ptr<Air>(new QuadraticAir(n, a0, o));
        case 3: return std::unique_ptr<Air>(new FibonacciRapAir(n, (size_t)steps, o));
        case 4: return std::unique_ptr<Air>(new DummyAir(n, o));
        default: throw std::runtime_error("unknown example AIR");
    }
}
static int finish_proof(const StarkProof& proof, uint8_t** proof_out, uint64_t* proof_len) {
    std::vector<uint8_t> bytes = serialize_proof(proof);
    *proof_out = (uint8_t*)std::malloc(bytes.size());
    std::memcpy(*proof_out, bytes.data(), bytes.size());
    *proof_len = bytes.size();
    return 0;
}

// The reference's trace generators; out = n x cols row-major BE.  Returns the number of rows (0 on error); call with
// out = nullptr to query the size.  `len` = trace length (steps for fibonacci_rap).
uint64_t oracle_example_trace(int kind, const uint8_t* params, uint64_t len, uint8_t* out, uint32_t* cols_out) {
    try {
        Fp a0 = params ? Fp::from_bytes_be(params) : Fp::one();
        Fp a1 = params ? Fp::from_bytes_be(params + 32) : Fp::one();
        std::vector<Fp> rows; size_t n = len; uint32_t cols = 1;
        switch (kind) {
            case 0: rows = fibonacci_trace(a0, a1, len); break;
            case 1: rows = fibonacci_trace_2_columns(a0, a1, len); cols = 2; break;
            case 2: rows = quadratic_trace(a0, len); break;
            case 3: rows = fibonacci_rap_trace(a0, a1, len, &n); cols = 2; break;
            case 4: rows = dummy_trace(len); cols = 2; break;
            default: return 0;
        }
        if (cols_out) *cols_out = cols;
        if (out) store_felts(rows, out);
        return n;
    } catch (...) { return 0; }
}

int oracle_example_prove(int kind, const uint8_t* params, uint64_t steps, const uint8_t* main_trace, uint64_t n, uint32_t main_cols,
                         const oracle_proof_options* opt, uint8_t** proof_out, uint64_t* proof_len) {
    try {
        ProofOptions o{opt->blowup_factor, (size_t)opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor};
        std::unique_ptr<Air> air = make_example(kind, n, params, steps, o);
        std::vector<Fp> tr = load_felts(main_trace, n * main_cols);
        Prover pr(*air, false);
        return finish_proof(pr.prove(tr, main_cols), proof_out, proof_len);
    } catch (const std::exception& e) { std::fprintf(stderr, "oracle_example_prove: %s\n", e.what()); return -2; }
}
int oracle_example_verify(int kind, const uint8_t* params, uint64_t steps, const uint8_t* proof, uint64_t proof_len, const oracle_proof_options* opt) {
    try {
        StarkProof p = deserialize_proof(proof, proof_len);
        ProofOptions o{opt->blowup_factor, (size_t)opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor};
        if (p.trace_length == 0 || (p.trace_length & (p.trace_length - 1)) || p.trace_length > (1ULL << 30)) return 0;
        std::unique_ptr<Air> air = make_example(kind, p.trace_length, params, steps, o);
        return verify(*air, p) ? 1 : 0;
    } catch (...) { return -3; }
}

// Program AIR descriptor: must match sp_air_desc of include/stark252_hip.h field for field.
struct oracle_air_boundary { uint32_t col; uint32_t pad; uint64_t step; uint8_t value[32]; };
struct oracle_air_desc {
    uint32_t main_cols, aux_cols;
    uint32_t n_offsets; uint32_t offsets[8];
    uint32_t n_transitions; uint32_t degrees[64]; uint32_t exemptions[64];
    uint32_t num_transition_exemptions;
    uint32_t degree_bound_factor;
    uint32_t n_ops; const AirOp* ops;
    uint32_t n_consts; const uint8_t* consts;
    uint32_t n_rap;
    uint32_t aux_kind;
    uint32_t n_boundary; const oracle_air_boundary* boundary;
    int (*aux_fn)(void* user, const uint8_t* rap, uint32_t n_rap, uint8_t* aux_rows_out); void* aux_user;   // aux_kind 2
};
static std::unique_ptr<ProgramAir> make_program_air(const oracle_air_desc* d, size_t n, const ProofOptions& o) {
    std::unique_ptr<ProgramAir> a(new ProgramAir());
    a->trace_len = n;
    a->ctx.proof_options = o;
    a->ctx.trace_columns = d->main_cols + d->aux_cols;
    for (uint32_t i = 0; i < d->n_offsets; ++i) a->ctx.transition_offsets.push_back(d->offsets[i]);
    for (uint32_t i = 0; i < d->n_transitions; ++i) { a->ctx.transition_degrees.push_back(d->degrees[i]); a->ctx.transition_exemptions.push_back(d->exemptions[i]); }
    a->ctx.num_transition_constraints = d->n_transitions;
    a->ctx.num_transition_exemptions = d->num_transition_exemptions;
    a->ops.assign(d->ops, d->ops + d->n_ops);
    a->consts = load_felts(d->consts, d->n_consts);
    a->n_rap = d->n_rap; a->aux_cols = d->aux_cols; a->aux_kind = d->aux_kind; a->bound_factor = d->degree_bound_factor;
    a->aux_fn = d->aux_fn; a->aux_user = d->aux_user;
    for (uint32_t i = 0; i < d->n_boundary; ++i) a->bcs.push_back({d->boundary[i].col, (size_t)d->boundary[i].step, Fp::from_bytes_be(d->boundary[i].value)});
    return a;
}
int oracle_program_air_prove(const oracle_air_desc* d, const uint8_t* main_trace, uint64_t n, const oracle_proof_options* opt,
                             uint8_t** proof_out, uint64_t* proof_len) {
    try {
        ProofOptions o{opt->blowup_factor, (size_t)opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor};
        std::unique_ptr<ProgramAir> air = make_program_air(d, n, o);
        std::vector<Fp> tr = load_felts(main_trace, n * d->main_cols);
        Prover pr(*air, false);
        return finish_proof(pr.prove(tr, d->main_cols), proof_out, proof_len);
    } catch (const std::exception& e) { std::fprintf(stderr, "oracle_program_air_prove: %s\n", e.what()); return -2; }
}
int oracle_program_air_verify(const oracle_air_desc* d, const uint8_t* proof, uint64_t proof_len, const oracle_proof_options* opt) {
    try {
        StarkProof p = deserialize_proof(proof, proof_len);
        ProofOptions o{opt->blowup_factor, (size_t)opt->fri_number_of_queries, opt->coset_offset, opt->grinding_factor};
        if (p.trace_length == 0 || (p.trace_length & (p.trace_length - 1)) || p.trace_length > (1ULL << 30)) return 0;
        std::unique_ptr<ProgramAir> air = make_program_air(d, p.trace_length, o);
        return verify(*air, p) ? 1 : 0;
    } catch (...) { return -3; }
}
}  // extern "C"

// Whole-proof drivers on top of the round-level prover (see prover.h): the host side of the Fiat-Shamir transcript, the proof
// writer, generate_cairo_proof (reference src/cairo/air.rs:1165-1171) and prove::<F, A> for program AIRs (src/starks/prover.rs:532-766).
#include "prover_internal.h"
#include "keccak.h"
#include <array>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace sp {

// ============================================================================================ whole proof (host driver)
namespace {

// DefaultTranscript of lambdaworks-crypto @ a17b951 and the sampling rules of
// reference src/starks/transcript.rs:13-79.
struct HostTranscript {
    std::vector<uint8_t> buf;
    void append(const uint8_t* d, size_t n) { buf.insert(buf.end(), d, d + n); }
    void append_felt(const fe& x) { uint8_t b[32]; fe_to_bytes_be(x, b); append(b, 32); }
    void challenge(uint8_t out[32]) {
        uint8_t d[32];
        sp_keccak256_host(buf.data(), buf.size(), d);
        for (int i = 0; i < 32; ++i) out[i] = d[31 - i];
        buf.assign(out, out + 32);
    }
    fe to_field() {
        uint8_t r[32];
        challenge(r);
        r[0] &= 0x07;  // 251 random bits (transcript.rs:24-43)
        return fe_from_bytes_be(r);
    }
    uint64_t to_usize() {
        uint8_t r[32];
        challenge(r);
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v = (v << 8) | r[i];
        return v;
    }
};

// Writes the proof in one pass into a buffer of its final size (every length of the format is known before the first byte).
struct ProofWriter {
    std::vector<uint8_t> b;
    size_t at = 0;
    explicit ProofWriter(size_t total) : b(total) {}
    void u64(uint64_t v) { for (int i = 7; i >= 0; --i) b[at++] = (uint8_t)(v >> (8 * i)); }
    void felt(const fe& x) { fe_to_bytes_be(x, &b[at]); at += 32; }
    // an opened value: Montgomery limbs, or already the 32 wire bytes (Openings::values_canonical_be)
    void opened(const fe& x, bool canonical_be) { if (canonical_be) { std::memcpy(&b[at], &x, 32); at += 32; } else felt(x); }
    void raw(const void* p, size_t n) { std::memcpy(&b[at], p, n); at += n; }
    void path(const digest32* p, uint32_t depth) { u64(depth); raw(p, (size_t)depth * 32); }
};

bool z_in_domains(const fe& z, const fe& hinv, uint32_t logn, uint32_t logN) {  // transcript.rs:53-69
    fe a = fe_mul(z, hinv), b = z;
    for (uint32_t i = 0; i < logN; ++i) a = fe_sqr(a);
    for (uint32_t i = 0; i < logn; ++i) b = fe_sqr(b);
    return fe_eq(a, fe_one()) || fe_eq(b, fe_one());
}

}  // namespace

// StarkProof serialization (reference proof/stark.rs:161-218, fri/fri_decommit.rs:24-45, frame.rs:86-106).
// roots: the trace-segment roots (one or two); ood: frame rows x C evaluations.
static void serialize_proof(uint64_t n, const std::vector<std::array<uint8_t, 32>>& roots, uint32_t C, const std::vector<fe>& ood,
                            const uint8_t comp_root[32], const fe& h1z, const fe& h2z, const std::vector<std::vector<uint8_t>>& fri_roots,
                            const fe& last_value, const std::vector<uint64_t>& iotas, const Openings& o, uint64_t nonce,
                            std::vector<uint8_t>& proof_out) {
    const uint32_t L = o.n_layers, d0 = o.depth0;
    const bool be = o.values_canonical_be;
    const size_t Q = iotas.size(), R = roots.size();
    size_t path_total = 0;
    for (uint32_t k = 0; k < L; ++k) path_total += d0 - k;
    // sizes of the nested records (each is preceded by its byte length)
    const size_t frame_bytes = 16 + 32 * ood.size() + 8;
    const size_t paths_bytes = (size_t)L * 8 + path_total * 32;                       // L paths: depth word + digests
    const size_t query_bytes = 8 + paths_bytes + 8 + 8 + (size_t)L * 32 + 8 + (size_t)L * 32 + 8 + paths_bytes;
    const size_t opening_bytes = (8 + (size_t)d0 * 32) + 8 + 64 + 8 + R * (8 + (size_t)d0 * 32) + 8 + (size_t)C * 32;
    const size_t total = 8 + 8 + R * 32 + 8 + frame_bytes + 32 + 8 + 64 + 8 + fri_roots.size() * 32 + 32 + 8 + Q * (8 + query_bytes) + 8 + Q * (8 + opening_bytes) + 8;
    ProofWriter w(total);
    w.u64(n);
    w.u64(R);
    for (auto& r : roots) w.raw(r.data(), 32);
    w.u64(frame_bytes);
    w.u64(ood.size()); w.u64(32);
    for (auto& e : ood) w.felt(e);
    w.u64(C);
    w.raw(comp_root, 32);
    w.u64(32); w.felt(h1z); w.felt(h2z);
    w.u64(fri_roots.size());
    for (auto& r : fri_roots) w.raw(r.data(), 32);
    w.felt(last_value);
    w.u64(Q);
    for (size_t s = 0; s < Q; ++s) {
        w.u64(query_bytes);
        w.u64(L);
        size_t po = 0;
        for (uint32_t k = 0; k < L; ++k) { w.path(&o.fri_paths_sym[s * path_total + po], d0 - k); po += d0 - k; }
        w.u64(32);
        w.u64(L);
        for (uint32_t k = 0; k < L; ++k) w.opened(o.fri_evals_sym[s * L + k], be);
        w.u64(L);
        for (uint32_t k = 0; k < L; ++k) w.opened(o.fri_evals[s * L + k], be);
        w.u64(L);
        po = 0;
        for (uint32_t k = 0; k < L; ++k) { w.path(&o.fri_paths[s * path_total + po], d0 - k); po += d0 - k; }
    }
    w.u64(Q);
    for (size_t s = 0; s < Q; ++s) {
        w.u64(opening_bytes);
        w.path(&o.comp_paths[s * d0], d0);
        w.u64(32);
        w.opened(o.comp_evals[s * 2], be); w.opened(o.comp_evals[s * 2 + 1], be);
        w.u64(R);
        w.path(&o.main_paths[s * d0], d0);
        if (R > 1) w.path(&o.aux_paths[s * d0], d0);
        w.u64(C);
        for (uint32_t j = 0; j < C; ++j) w.opened(o.trace_evals[s * C + j], be);
    }
    w.u64(nonce);
    if (w.at != total) throw std::runtime_error("serialize_proof: size bookkeeping is off");
    proof_out.swap(w.b);
}

ProverHolder* prover_holder(sp_ctx* c, bool create) {
    ProverHolder* h = dynamic_cast<ProverHolder*>(c->prover_state_deleter_holder);
    if (!h && create) {
        delete c->prover_state_deleter_holder;
        h = new ProverHolder(c);
        c->prover_state_deleter_holder = h;
    }
    return h;
}

int cairo_prove(sp_ctx* ctx, const uint8_t* main_trace, uint64_t n, uint32_t cols, const PublicInputs& pub,
                const ProofOptionsHost& opt, std::vector<uint8_t>& proof_out, float round_ms[5],
                StarkProver::TraceSource src, int col_enc, uint64_t col_stride) {
    try {
        CairoAirInfo air = cairo_air_info(pub);
        if (cols != air.main_columns) { sp_set_error("cairo_prove: main trace must have 34 columns (43 with the range-check builtin)"); return SP_E_INVALID_ARG; }
        StarkProver* P = &prover_holder(ctx, true)->prover;   // kept (with its device buffers) across proofs of the same shape on this context
        struct Events {   // released on every exit path
            hipEvent_t e[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
            ~Events() { for (auto& x : e) if (x) (void)hipEventDestroy(x); }
        } evs;
        hipEvent_t* ev = evs.e;
        for (auto& e : evs.e) SP_HIP_CHECK(hipEventCreate(&e));
        double _tp = wall_ms();
        static const bool tail_timing = std::getenv("SP_TAIL_TIMING") != nullptr;
        const double t_entry = wall_ms();
        if (opt.fri_number_of_queries == 0) { sp_set_error("prove: fri_number_of_queries must be at least 1 (the reference emits a proof without openings for 0; this prover does not)"); return SP_E_INVALID_ARG; }
        SP_TRY(P->setup(n, air.main_columns, air.aux_columns, air.has_rc_builtin, opt));
        SP_TIMEPOINT("setup (alloc + tables)");
        HostTranscript tr;
        uint8_t root[32];
        // ---- round 1 (reference prover.rs:187-224)
        SP_HIP_CHECK(hipEventRecord(ev[0], ctx->stream));
        const double t_ev0 = wall_ms();
        P->request_aux_presort(pub);                    // the sorts of the auxiliary trace: beside round 1 too
        if (pub.num_steps >= 1 && pub.num_steps <= n)   // round 2's boundary denominators need no challenge: beside round 1
            SP_TRY(P->prefetch_boundary_inverses({0, pub.num_steps - 1, n - 1}));
        P->hint_binary_columns(16);                     // the instruction flags (air.rs:29-46): one bit per cell over PCIe from a row-major host table
        SP_TRY(P->commit_trace(0, main_trace, cols, root, src, col_enc, col_stride));
        uint8_t main_root[32]; std::memcpy(main_root, root, 32);
        SP_TIMEPOINT("r1 commit main (H2D+iNTT+LDE+Merkle)");
        tr.append(root, 32);
        fe rap[3] = {tr.to_field(), tr.to_field(), tr.to_field()};
        SP_TRY(P->commit_aux_cairo(pub, rap, root));
        SP_TIMEPOINT("r1 aux trace + commit (device)");
        uint8_t aux_root[32]; std::memcpy(aux_root, root, 32);
        tr.append(root, 32);
        SP_HIP_CHECK(hipEventRecord(ev[1], ctx->stream));
        // ---- round 2 (reference prover.rs:597-635)
        std::vector<BoundaryConstraint> bcs = boundary_constraints(pub, rap, n, air.has_rc_builtin);
        const uint32_t T = air.num_transition_constraints;
        SP_TRY(P->composition_precheck(rap, bcs, T));   // runs while the challenges below are sampled
        std::vector<fe> b_alpha(bcs.size()), b_beta(bcs.size()), t_alpha(T), t_beta(T);
        for (auto& x : b_alpha) x = tr.to_field();
        for (auto& x : b_beta) x = tr.to_field();
        for (auto& x : t_alpha) x = tr.to_field();
        for (auto& x : t_beta) x = tr.to_field();
        SP_TRY(P->composition(rap, bcs, b_alpha, b_beta, t_alpha, t_beta, air.transition_degrees, air.transition_exemptions, root));
        uint8_t comp_root[32]; std::memcpy(comp_root, root, 32);
        SP_TIMEPOINT("r2 composition");
        tr.append(root, 32);
        SP_HIP_CHECK(hipEventRecord(ev[2], ctx->stream));
        // ---- round 3 (reference prover.rs:652-684)
        const uint32_t logn = (uint32_t)sp_log2_exact(n), logN = logn + (uint32_t)sp_log2_exact(opt.blowup_factor);
        fe hinv = fe_inv(fe_from_u64(opt.coset_offset));
        fe z;
        do { z = tr.to_field(); } while (z_in_domains(z, hinv, logn, logN));
        fe h1z, h2z;
        std::vector<fe> ood;
        SP_TRY(P->ood(z, &h1z, &h2z, ood));
        SP_TIMEPOINT("r3 ood");
        tr.append_felt(h1z); tr.append_felt(h2z);
        for (auto& e : ood) tr.append_felt(e);
        SP_HIP_CHECK(hipEventRecord(ev[3], ctx->stream));
        // ---- round 4 (reference prover.rs:327-404)
        fe gamma = tr.to_field(), gamma_p = tr.to_field();
        std::vector<fe> tg(2 * (size_t)P->cols());
        for (auto& x : tg) x = tr.to_field();
        SP_TRY(P->deep_fri_begin(gamma, gamma_p, tg, root));
        std::vector<std::vector<uint8_t>> fri_roots;
        fri_roots.emplace_back(root, root + 32);
        tr.append(root, 32);
        fe last_value;
        for (;;) {
            fe zeta = tr.to_field();
            // From the first layer this rank holds whole (layer 0 on one GPU; behind the sharded layers otherwise) the layers follow
            // each other on the device without a host round trip; the transcript catches up afterwards.
            if (P->fri_chain_available()) {
                std::vector<std::array<uint8_t, 32>> rest;
                SP_TRY(P->fri_commit_chain(zeta, tr.buf.data(), rest, &last_value));
                for (auto& r : rest) {
                    fri_roots.emplace_back(r.begin(), r.end());
                    tr.append(r.data(), 32);
                    (void)tr.to_field();      // zeta_k: the device sampled the same value
                }
                break;
            }
            int is_last = 0;
            SP_TRY(P->fri_fold_commit(zeta, root, &last_value, &is_last));
            if (is_last) break;
            fri_roots.emplace_back(root, root + 32);
            tr.append(root, 32);
        }
        SP_TIMEPOINT("r4 deep + fri commit");
        tr.append_felt(last_value);
        uint8_t gch[32];
        tr.challenge(gch);
        uint64_t nonce = 0;
        SP_TRY(P->grind(gch, opt.grinding_factor, &nonce));
        {
            uint8_t nb[8];
            for (int i = 0; i < 8; ++i) nb[i] = (uint8_t)(nonce >> (56 - 8 * i));
            tr.append(nb, 8);
        }
        SP_TIMEPOINT("r4 grinding");
        std::vector<uint64_t> iotas(opt.fri_number_of_queries);
        for (auto& x : iotas) x = tr.to_usize() % P->N();
        Openings& o = prover_holder(ctx, true)->open;     // (kept with the prover: its arrays are reused by the next proof)
        SP_TRY(P->open(iotas, o, true));   // opened values as wire bytes (encoded on the device)
        SP_TIMEPOINT("r4 openings");
        const double t_open = wall_ms();
        SP_HIP_CHECK(hipEventRecord(ev[4], ctx->stream));
        SP_HIP_CHECK(hipEventSynchronize(ev[4]));
        if (round_ms) {
            round_ms[0] = 0.f;
            for (int r = 0; r < 4; ++r) SP_HIP_CHECK(hipEventElapsedTime(&round_ms[r + 1], ev[r], ev[r + 1]));
        }
        std::vector<std::array<uint8_t, 32>> roots(2);
        std::memcpy(roots[0].data(), main_root, 32); std::memcpy(roots[1].data(), aux_root, 32);
        const double t_ser0 = wall_ms();
        serialize_proof(n, roots, P->cols(), ood, comp_root, h1z, h2z, fri_roots, last_value, iotas, o, nonce, proof_out);
        if (tail_timing) std::fprintf(stderr, "[sp_tail] before the first event %.3f ms, events + bookkeeping after open() %.3f ms, serialize %.3f ms\n", t_ev0 - t_entry,
                                      t_ser0 - t_open, wall_ms() - t_ser0);
        return SP_OK;
    } catch (const std::exception& e) {
        sp_set_error(std::string("cairo_prove: ") + e.what());
        return SP_E_INVALID_ARG;
    }
}

// prove::<F, A> for a program AIR (reference src/starks/prover.rs:532-766): same rounds, the AIR-specific parts come from
// the descriptor - RAP challenges (n_rap field samples), auxiliary trace (by kind, built on the host: the example AIRs are
// tiny), boundary constraints, transition program.
int air_prove(sp_ctx* ctx, const AirDescHost& air, const uint8_t* main_trace, uint64_t n, const ProofOptionsHost& opt,
              std::vector<uint8_t>& proof_out) {
    try {
        if (air.main_cols == 0 || air.main_cols + air.aux_cols > 64) { sp_set_error("air_prove: column count out of range"); return SP_E_INVALID_ARG; }
        StarkProver* P = &prover_holder(ctx, true)->prover;   // kept (with its device buffers) across proofs of the same shape on this context
        if (opt.fri_number_of_queries == 0) { sp_set_error("prove: fri_number_of_queries must be at least 1 (the reference emits a proof without openings for 0; this prover does not)"); return SP_E_INVALID_ARG; }
        SP_TRY(P->setup(n, air.main_cols, air.aux_cols, false, opt));
        HostTranscript tr;
        uint8_t root[32];
        std::vector<std::array<uint8_t, 32>> roots;
        // ---- round 1 (reference prover.rs:187-224)
        SP_TRY(P->commit_trace(0, main_trace, air.main_cols, root));
        roots.emplace_back(); std::memcpy(roots.back().data(), root, 32);
        tr.append(root, 32);
        std::vector<fe> rap(air.n_rap);
        for (auto& x : rap) x = tr.to_field();
        if (air.aux_cols && air.aux_kind == 2) {
            // build_auxiliary_trace of the caller's AIR (traits.rs:25-29): row-major n x aux_cols from the RAP challenges
            if (!air.aux_fn) { sp_set_error("air_prove: aux_kind 2 needs aux_fn"); return SP_E_INVALID_ARG; }
            std::vector<uint8_t> rap_bytes(std::max<size_t>(1, rap.size()) * 32), aux_rows((size_t)n * air.aux_cols * 32);
            if (!rap.empty()) SP_TRY(sp_fe_from_device(ctx->enc, reinterpret_cast<const uint8_t*>(rap.data()), rap.size(), rap_bytes.data()));
            if (air.aux_fn(air.aux_user, rap_bytes.data(), (uint32_t)rap.size(), aux_rows.data()) != 0) { sp_set_error("air_prove: the auxiliary-trace callback failed"); return SP_E_INVALID_ARG; }
            SP_TRY(P->commit_trace(1, aux_rows.data(), air.aux_cols, root));
            roots.emplace_back(); std::memcpy(roots.back().data(), root, 32);
            tr.append(root, 32);
        } else if (air.aux_cols) {
            if (air.aux_kind != 1 || air.aux_cols != 1 || air.main_cols < 2 || air.n_rap < 1) {
                sp_set_error("air_prove: unknown auxiliary-trace kind (1 = fibonacci_rap permutation column, 2 = caller-supplied)");
                return SP_E_UNSUPPORTED;
            }
            // fibonacci_rap.rs:69-93: z_0 = 1, z_i = z_(i-1) (a_(i-1) + gamma) / (b_(i-1) + gamma)
            std::vector<fe> den(n), num(n);
            for (uint64_t i = 0; i < n; ++i) {
                fe a, b;
                const uint8_t* row = main_trace + (size_t)i * air.main_cols * 32;
                if (ctx->enc == SP_FE_CANON_BE) { a = fe_from_bytes_be(row); b = fe_from_bytes_be(row + 32); }
                else { uint64_t l[4]; std::memcpy(l, row, 32); a = fe_from_lw_limbs(l); std::memcpy(l, row + 32, 32); b = fe_from_lw_limbs(l); }
                num[i] = fe_add(a, rap[0]); den[i] = fe_add(b, rap[0]);
            }
            for (auto& d : den) if (fe_is_zero(d)) { sp_set_error("air_prove: zero denominator in the permutation column"); return SP_E_ZERO_INVERSE; }
            host_batch_inverse(den);
            std::vector<uint8_t> aux_rows((size_t)n * 32);
            fe zacc = fe_one();
            for (uint64_t i = 0; i < n; ++i) {
                if (i > 0) zacc = fe_mul(zacc, fe_mul(num[i - 1], den[i - 1]));
                if (ctx->enc == SP_FE_CANON_BE) fe_to_bytes_be(zacc, &aux_rows[(size_t)i * 32]);
                else { uint64_t l[4]; fe_to_lw_limbs(zacc, l); std::memcpy(&aux_rows[(size_t)i * 32], l, 32); }
            }
            SP_TRY(P->commit_trace(1, aux_rows.data(), 1, root));
            roots.emplace_back(); std::memcpy(roots.back().data(), root, 32);
            tr.append(root, 32);
        }
        // ---- round 2 (reference prover.rs:597-635)
        const size_t B = air.boundary.size(), T = air.degrees.size();
        std::vector<fe> b_alpha(B), b_beta(B), t_alpha(T), t_beta(T);
        for (auto& x : b_alpha) x = tr.to_field();
        for (auto& x : b_beta) x = tr.to_field();
        for (auto& x : t_alpha) x = tr.to_field();
        for (auto& x : t_beta) x = tr.to_field();
        SP_TRY(P->composition_air(air, rap, b_alpha, b_beta, t_alpha, t_beta, root));
        uint8_t comp_root[32]; std::memcpy(comp_root, root, 32);
        tr.append(root, 32);
        // ---- round 3 (reference prover.rs:652-684)
        const uint32_t logn = (uint32_t)sp_log2_exact(n), logN = logn + (uint32_t)sp_log2_exact(opt.blowup_factor);
        fe hinv = fe_inv(fe_from_u64(opt.coset_offset));
        fe z;
        do { z = tr.to_field(); } while (z_in_domains(z, hinv, logn, logN));
        fe h1z, h2z;
        std::vector<fe> ood;
        SP_TRY(P->ood(z, &h1z, &h2z, ood));
        tr.append_felt(h1z); tr.append_felt(h2z);
        for (auto& e : ood) tr.append_felt(e);
        // ---- round 4 (reference prover.rs:327-404)
        fe gamma = tr.to_field(), gamma_p = tr.to_field();
        std::vector<fe> tg((size_t)P->frame_rows() * P->cols());
        for (auto& x : tg) x = tr.to_field();
        SP_TRY(P->deep_fri_begin(gamma, gamma_p, tg, root));
        std::vector<std::vector<uint8_t>> fri_roots;
        fri_roots.emplace_back(root, root + 32);
        tr.append(root, 32);
        fe last_value;
        for (;;) {
            fe zeta = tr.to_field();
            // From the first layer this rank holds whole (layer 0 on one GPU; behind the sharded layers otherwise) the layers follow
            // each other on the device without a host round trip; the transcript catches up afterwards.
            if (P->fri_chain_available()) {
                std::vector<std::array<uint8_t, 32>> rest;
                SP_TRY(P->fri_commit_chain(zeta, tr.buf.data(), rest, &last_value));
                for (auto& r : rest) {
                    fri_roots.emplace_back(r.begin(), r.end());
                    tr.append(r.data(), 32);
                    (void)tr.to_field();      // zeta_k: the device sampled the same value
                }
                break;
            }
            int is_last = 0;
            SP_TRY(P->fri_fold_commit(zeta, root, &last_value, &is_last));
            if (is_last) break;
            fri_roots.emplace_back(root, root + 32);
            tr.append(root, 32);
        }
        tr.append_felt(last_value);
        uint8_t gch[32];
        tr.challenge(gch);
        uint64_t nonce = 0;
        SP_TRY(P->grind(gch, opt.grinding_factor, &nonce));
        {
            uint8_t nb[8];
            for (int i = 0; i < 8; ++i) nb[i] = (uint8_t)(nonce >> (56 - 8 * i));
            tr.append(nb, 8);
        }
        std::vector<uint64_t> iotas(opt.fri_number_of_queries);
        for (auto& x : iotas) x = tr.to_usize() % P->N();
        Openings& o = prover_holder(ctx, true)->open;     // (kept with the prover: its arrays are reused by the next proof)
        SP_TRY(P->open(iotas, o, true));   // opened values as wire bytes (encoded on the device)
        SP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        serialize_proof(n, roots, P->cols(), ood, comp_root, h1z, h2z, fri_roots, last_value, iotas, o, nonce, proof_out);
        return SP_OK;
    } catch (const std::exception& e) {
        sp_set_error(std::string("air_prove: ") + e.what());
        return SP_E_INVALID_ARG;
    }
}

}  // namespace sp

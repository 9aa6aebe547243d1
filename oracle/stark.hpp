// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/fp.hpp header).
// CPU restatement of the reference STARK engine, written to follow the reference algorithm step by step
// (zero-padded size-N FFT per column, row-major transposes, per-point `pow`, coefficient-form FRI with a fresh
// FFT per layer, Ruffini division for the DEEP polynomial), NOT the GPU-friendly reformulation:
//   prove                      reference src/starks/prover.rs:532-766 (rounds: :126-224, :226-286, :288-325, :327-529)
//   ConstraintEvaluator        reference src/starks/constraints/evaluator.rs:38-260, :299-323
//   Domain                     reference src/starks/domain.rs:20-56
//   FRI                        reference src/starks/fri/mod.rs:20-127, fri_commitment.rs:30-47, fri_functions.rs:4-27
//   transcript sampling        reference src/starks/transcript.rs:13-79
//   serialization              reference src/starks/proof/stark.rs:53-77,161-218, fri/fri_decommit.rs:19-46,
//                              frame.rs:86-106, utils.rs:6-13
//   verify                     reference src/starks/verifier.rs:59-657
// `legacy_boundary` reproduces the older commit that generated tests/golden/fibonacci_{500,1000}.proof
// (SURVEY.md §8(c)): one (alpha,beta) pair per trace column and a per-column interpolated boundary polynomial.
#pragma once
#include "air.hpp"
#include "poly.hpp"
#include <memory>

namespace oracle {

struct FriDecommitment {
    std::vector<std::vector<Digest>> layers_auth_paths_sym;
    std::vector<Fp> layers_evaluations_sym;
    std::vector<std::vector<Digest>> layers_auth_paths;
    std::vector<Fp> layers_evaluations;
};
struct DeepPolynomialOpenings {
    std::vector<Digest> lde_composition_poly_proof;
    Fp lde_composition_poly_even_evaluation, lde_composition_poly_odd_evaluation;
    std::vector<std::vector<Digest>> lde_trace_merkle_proofs;
    std::vector<Fp> lde_trace_evaluations;
};
struct StarkProof {
    uint64_t trace_length;
    std::vector<Digest> lde_trace_merkle_roots;
    std::vector<Fp> trace_ood_frame_data;  // row-major [offset k][column j]
    size_t trace_ood_row_width;
    Digest composition_poly_root;
    Fp composition_poly_even_ood_evaluation, composition_poly_odd_ood_evaluation;
    std::vector<Digest> fri_layers_merkle_roots;
    Fp fri_last_value;
    std::vector<FriDecommitment> query_list;
    std::vector<DeepPolynomialOpenings> deep_poly_openings;
    uint64_t nonce;
};

// ---------------------------------------------------------------- serialization
struct ByteWriter {
    std::vector<uint8_t> b;
    void u64(uint64_t v) { for (int i = 7; i >= 0; --i) b.push_back((uint8_t)(v >> (8 * i))); }
    void felt(const Fp& x) { uint8_t t[32]; x.to_bytes_be(t); b.insert(b.end(), t, t + 32); }
    void digest(const Digest& d) { b.insert(b.end(), d.begin(), d.end()); }
    void path(const std::vector<Digest>& p) { u64(p.size()); for (auto& d : p) digest(d); }
    void bytes(const std::vector<uint8_t>& v) { b.insert(b.end(), v.begin(), v.end()); }
};

inline std::vector<uint8_t> serialize_fri_decommitment(const FriDecommitment& q) {
    ByteWriter w;
    w.u64(q.layers_auth_paths_sym.size());
    for (auto& p : q.layers_auth_paths_sym) w.path(p);
    w.u64(32);
    w.u64(q.layers_evaluations_sym.size());
    for (auto& e : q.layers_evaluations_sym) w.felt(e);
    w.u64(q.layers_evaluations.size());
    for (auto& e : q.layers_evaluations) w.felt(e);
    w.u64(q.layers_auth_paths.size());
    for (auto& p : q.layers_auth_paths) w.path(p);
    return w.b;
}
inline std::vector<uint8_t> serialize_deep_openings(const DeepPolynomialOpenings& o) {
    ByteWriter w;
    w.path(o.lde_composition_poly_proof);
    w.u64(32);
    w.felt(o.lde_composition_poly_even_evaluation);
    w.felt(o.lde_composition_poly_odd_evaluation);
    w.u64(o.lde_trace_merkle_proofs.size());
    for (auto& p : o.lde_trace_merkle_proofs) w.path(p);
    w.u64(o.lde_trace_evaluations.size());
    for (auto& e : o.lde_trace_evaluations) w.felt(e);
    return w.b;
}
inline std::vector<uint8_t> serialize_proof(const StarkProof& p) {
    ByteWriter w;
    w.u64(p.trace_length);
    w.u64(p.lde_trace_merkle_roots.size());
    for (auto& r : p.lde_trace_merkle_roots) w.digest(r);
    {
        ByteWriter f;
        f.u64(p.trace_ood_frame_data.size());
        f.u64(p.trace_ood_frame_data.empty() ? 0 : 32);
        for (auto& e : p.trace_ood_frame_data) f.felt(e);
        f.u64(p.trace_ood_row_width);
        w.u64(f.b.size());
        w.bytes(f.b);
    }
    w.digest(p.composition_poly_root);
    w.u64(32);
    w.felt(p.composition_poly_even_ood_evaluation);
    w.felt(p.composition_poly_odd_ood_evaluation);
    w.u64(p.fri_layers_merkle_roots.size());
    for (auto& r : p.fri_layers_merkle_roots) w.digest(r);
    w.felt(p.fri_last_value);
    w.u64(p.query_list.size());
    for (auto& q : p.query_list) { auto qb = serialize_fri_decommitment(q); w.u64(qb.size()); w.bytes(qb); }
    w.u64(p.deep_poly_openings.size());
    for (auto& o : p.deep_poly_openings) { auto ob = serialize_deep_openings(o); w.u64(ob.size()); w.bytes(ob); }
    w.u64(p.nonce);
    return w.b;
}

struct ByteReader {
    const uint8_t* p; size_t n, pos;
    ByteReader(const uint8_t* d, size_t len) : p(d), n(len), pos(0) {}
    void need(size_t k) { if (pos + k > n) throw std::runtime_error("InvalidAmountOfBytes"); }
    uint64_t u64() { need(8); uint64_t v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | p[pos + i]; pos += 8; return v; }
    uint8_t u8() { need(1); return p[pos++]; }
    uint16_t u16() { need(2); uint16_t v = (uint16_t)((p[pos] << 8) | p[pos + 1]); pos += 2; return v; }
    Fp felt() { need(32); Fp x = Fp::from_bytes_be(p + pos); pos += 32; return x; }
    Digest digest() { need(32); Digest d; std::memcpy(d.data(), p + pos, 32); pos += 32; return d; }
    std::vector<Digest> path() { uint64_t k = u64(); if (k > n) throw std::runtime_error("bad len"); std::vector<Digest> v; for (uint64_t i = 0; i < k; ++i) v.push_back(digest()); return v; }
};

// StarkProof::deserialize (proof/stark.rs:225-440), with its sub-deserializers Frame::deserialize (frame.rs:113-156),
// FriDecommitment::deserialize (fri/fri_decommit.rs:53-145) and the Merkle-path reader (stark.rs: deserialize_proof), statement by
// statement: every length-prefixed part is parsed INSIDE the slice its prefix announces (bytes left over in a slice are ignored, a slice
// that is too short is an error), a field element is the first 32 of `felt_len` bytes (fewer than 32: an error; lambdaworks'
// UnsignedInteger::from_bytes_be reads `0..32`), and the nonce is the last eight bytes of whatever follows the openings.
inline Fp read_felt(ByteReader& r, uint64_t felt_len) {
    if (felt_len > r.n - r.pos) throw std::runtime_error("InvalidAmountOfBytes");
    if (felt_len < 32) throw std::runtime_error("FromBEBytesError");
    Fp x = Fp::from_bytes_be(r.p + r.pos);
    r.pos += felt_len;
    return x;
}
inline ByteReader sub_slice(ByteReader& r, uint64_t len) {
    if (len > r.n - r.pos) throw std::runtime_error("InvalidAmountOfBytes");
    ByteReader s(r.p + r.pos, (size_t)len);
    r.pos += len;
    return s;
}
inline StarkProof deserialize_proof(const uint8_t* data, size_t len) {
    ByteReader r(data, len);
    StarkProof p;
    p.trace_length = r.u64();
    uint64_t nroots = r.u64();
    if (nroots > len) throw std::runtime_error("InvalidAmountOfBytes");
    for (uint64_t i = 0; i < nroots; ++i) p.lde_trace_merkle_roots.push_back(r.digest());
    {   // Frame::deserialize on the announced slice
        ByteReader f = sub_slice(r, r.u64());
        uint64_t nel = f.u64();
        uint64_t felt_len = f.u64();
        if (nel > len) throw std::runtime_error("InvalidAmountOfBytes");
        for (uint64_t i = 0; i < nel; ++i) p.trace_ood_frame_data.push_back(read_felt(f, felt_len));
        p.trace_ood_row_width = f.u64();
    }
    p.composition_poly_root = r.digest();
    const uint64_t felt_len = r.u64();
    p.composition_poly_even_ood_evaluation = read_felt(r, felt_len);
    p.composition_poly_odd_ood_evaluation = read_felt(r, felt_len);
    uint64_t nfri = r.u64();
    if (nfri > len) throw std::runtime_error("InvalidAmountOfBytes");
    for (uint64_t i = 0; i < nfri; ++i) p.fri_layers_merkle_roots.push_back(r.digest());
    p.fri_last_value = read_felt(r, felt_len);
    uint64_t nq = r.u64();
    if (nq > len) throw std::runtime_error("InvalidAmountOfBytes");
    for (uint64_t i = 0; i < nq; ++i) {
        ByteReader q = sub_slice(r, r.u64());
        FriDecommitment d;
        uint64_t k = q.u64(); if (k > len) throw std::runtime_error("InvalidAmountOfBytes");
        for (uint64_t j = 0; j < k; ++j) d.layers_auth_paths_sym.push_back(q.path());
        const uint64_t fl = q.u64();
        k = q.u64(); if (k > len) throw std::runtime_error("InvalidAmountOfBytes");
        for (uint64_t j = 0; j < k; ++j) d.layers_evaluations_sym.push_back(read_felt(q, fl));
        k = q.u64(); if (k > len) throw std::runtime_error("InvalidAmountOfBytes");
        for (uint64_t j = 0; j < k; ++j) d.layers_evaluations.push_back(read_felt(q, fl));
        k = q.u64(); if (k > len) throw std::runtime_error("InvalidAmountOfBytes");
        for (uint64_t j = 0; j < k; ++j) d.layers_auth_paths.push_back(q.path());
        p.query_list.push_back(d);
    }
    uint64_t no = r.u64();
    if (no > len) throw std::runtime_error("InvalidAmountOfBytes");
    for (uint64_t i = 0; i < no; ++i) {
        ByteReader q = sub_slice(r, r.u64());
        DeepPolynomialOpenings o;
        o.lde_composition_poly_proof = q.path();
        const uint64_t fl = q.u64();
        o.lde_composition_poly_even_evaluation = read_felt(q, fl);
        o.lde_composition_poly_odd_evaluation = read_felt(q, fl);
        uint64_t k = q.u64(); if (k > len) throw std::runtime_error("InvalidAmountOfBytes");
        for (uint64_t j = 0; j < k; ++j) o.lde_trace_merkle_proofs.push_back(q.path());
        k = q.u64(); if (k > len) throw std::runtime_error("InvalidAmountOfBytes");
        for (uint64_t j = 0; j < k; ++j) o.lde_trace_evaluations.push_back(read_felt(q, fl));
        p.deep_poly_openings.push_back(o);
    }
    // the nonce is the LAST eight bytes of what is left (stark.rs:410-422: `bytes.len() - 8 ..`), whatever lies between the openings and them
    r.need(8);
    r.pos = r.n - 8;
    p.nonce = r.u64();
    return p;
}

// ---------------------------------------------------------------- domain (domain.rs:20-56)
struct Domain {
    unsigned root_order, lde_root_order;
    std::vector<Fp> lde_roots_of_unity_coset, trace_roots_of_unity;
    Fp trace_primitive_root, coset_offset;
    size_t blowup_factor, interpolation_domain_size;
    explicit Domain(const Air& air) {
        blowup_factor = air.ctx.proof_options.blowup_factor;
        coset_offset = Fp::from_u64(air.ctx.proof_options.coset_offset);
        interpolation_domain_size = air.trace_len;
        root_order = log2_exact(air.trace_len);
        trace_primitive_root = primitive_root(root_order);
        trace_roots_of_unity = root_coset(root_order, interpolation_domain_size, Fp::one());
        lde_root_order = log2_exact(air.trace_len * blowup_factor);
        lde_roots_of_unity_coset = root_coset(lde_root_order, air.trace_len * blowup_factor, coset_offset);
    }
};

inline std::vector<Fp> batch_sample_challenges(size_t k, Transcript& t) {
    std::vector<Fp> v(k);
    for (size_t i = 0; i < k; ++i) v[i] = t.to_field();
    return v;
}
// transcript.rs:53-69 (membership tested against both domains)
inline Fp sample_z_ood(const Domain& d, Transcript& t) {
    for (;;) {
        Fp v = t.to_field();
        // v in LDE coset  <=>  (v/h)^N == 1 ;  v in trace domain <=> v^n == 1   (exact, cheaper than a scan)
        Fp a = v * d.coset_offset.inv();
        Fp b = v;
        for (unsigned i = 0; i < d.lde_root_order; ++i) a = a.square();
        for (unsigned i = 0; i < d.root_order; ++i) b = b.square();
        if (a != Fp::one() && b != Fp::one()) return v;
    }
}

// prover.rs:106-123
inline std::vector<Fp> evaluate_polynomial_on_lde_domain(const Poly& p, size_t blowup, size_t domain_size, const Fp& offset) {
    std::vector<Fp> ev = evaluate_offset_fft(p, blowup, domain_size, offset);
    size_t step = ev.size() / (domain_size * blowup);
    if (step == 1) return ev;
    std::vector<Fp> out;
    for (size_t i = 0; i < ev.size(); i += step) out.push_back(ev[i]);
    return out;
}

// ---------------------------------------------------------------- prover
struct ProverTimings { double round1 = 0, round2 = 0, round3 = 0, round4 = 0; };

struct Prover {
    const Air& air;
    bool legacy_boundary;
    Domain domain;
    Transcript transcript;
    ProverTimings timings;
    // round 1
    std::vector<Poly> trace_polys;
    std::vector<Fp> lde_trace;  // row-major N x C
    size_t C = 0, N = 0, n = 0;
    std::vector<MerkleTree> trace_trees;
    std::vector<Digest> trace_roots;
    std::vector<Fp> rap;

    Prover(const Air& a, bool legacy = false) : air(a), legacy_boundary(legacy), domain(a) {
        n = a.trace_len; N = n * domain.blowup_factor;
    }

    // interpolate_and_commit (prover.rs:126-159): returns column-major LDE evaluations of this segment
    std::vector<std::vector<Fp>> interpolate_and_commit(const std::vector<Fp>& table, size_t cols) {
        std::vector<Poly> polys(cols);
        std::vector<std::vector<Fp>> evals(cols);
#pragma omp parallel for schedule(dynamic)
        for (long j = 0; j < (long)cols; ++j) {
            std::vector<Fp> col(n);
            for (size_t i = 0; i < n; ++i) col[i] = table[i * cols + j];
            polys[j] = interpolate_fft(col);
            evals[j] = evaluate_polynomial_on_lde_domain(polys[j], domain.blowup_factor, domain.interpolation_domain_size, domain.coset_offset);
        }
        // TraceTable::new_from_cols + rows() + batch_commit
        std::vector<Fp> rows(N * cols);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)N; ++i)
            for (size_t j = 0; j < cols; ++j) rows[(size_t)i * cols + j] = evals[j][i];
        MerkleTree tree = MerkleTree::build_batched(rows.data(), N, cols);
        transcript.append_digest(tree.root);
        trace_roots.push_back(tree.root);
        trace_trees.push_back(std::move(tree));
        for (auto& p : polys) trace_polys.push_back(std::move(p));
        return evals;
    }

    StarkProof prove(const std::vector<Fp>& main_trace, size_t main_cols);
};

static inline double now_s() {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

inline StarkProof Prover::prove(const std::vector<Fp>& main_trace, size_t main_cols) {
    const AirContext& ctx = air.ctx;
    const size_t b = domain.blowup_factor;
    double t0 = now_s();
    // ================= Round 1 (prover.rs:187-224)
    std::vector<std::vector<Fp>> evals = interpolate_and_commit(main_trace, main_cols);
    rap = air.build_rap_challenges(transcript);
    std::vector<Fp> aux = air.build_auxiliary_trace(main_trace, main_cols, rap);
    if (!aux.empty()) {
        size_t ac = air.number_auxiliary_rap_columns();
        std::vector<std::vector<Fp>> ev2 = interpolate_and_commit(aux, ac);
        for (auto& e : ev2) evals.push_back(std::move(e));
    }
    C = evals.size();
    lde_trace.resize(N * C);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; ++i)
        for (size_t j = 0; j < C; ++j) lde_trace[(size_t)i * C + j] = evals[j][i];
    evals.clear(); evals.shrink_to_fit();
    double t1 = now_s(); timings.round1 = t1 - t0;

    // ================= Round 2 (prover.rs:597-635, :226-286)
    std::vector<BoundaryConstraint> bcs = air.boundary_constraints(rap);
    size_t nbcoef = legacy_boundary ? C : bcs.size();
    std::vector<Fp> b_alpha = batch_sample_challenges(nbcoef, transcript);
    std::vector<Fp> b_beta = batch_sample_challenges(nbcoef, transcript);
    std::vector<Fp> t_alpha = batch_sample_challenges(ctx.num_transition_constraints, transcript);
    std::vector<Fp> t_beta = batch_sample_challenges(ctx.num_transition_constraints, transcript);

    const std::vector<Fp>& xs = domain.lde_roots_of_unity_coset;
    const size_t D = air.composition_poly_degree_bound();
    // boundary term (evaluator.rs:58-115)
    std::vector<Fp> boundary_evaluation(N, Fp::zero());
    {
        std::vector<Fp> d_adj(N);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)N; ++i) d_adj[i] = xs[i].pow(D - n);
        if (!legacy_boundary) {
            for (size_t k = 0; k < bcs.size(); ++k) {
                Fp point = domain.trace_primitive_root.pow(bcs[k].step);
                std::vector<Fp> zinv(N);
                for (size_t i = 0; i < N; ++i) zinv[i] = xs[i] - point;
                batch_inverse(zinv);
#pragma omp parallel for schedule(static)
                for (long i = 0; i < (long)N; ++i)
                    boundary_evaluation[i] += zinv[i] * (b_alpha[k] * d_adj[i] + b_beta[k]) * (lde_trace[(size_t)i * C + bcs[k].col] - bcs[k].value);
            }
        } else {
            // legacy scheme: per column, (alpha x^(D-n) + beta) (t_col - I_col) / Z_col
            for (size_t col = 0; col < C; ++col) {
                std::vector<Fp> px, py;
                for (auto& bc : bcs) if (bc.col == col) { px.push_back(domain.trace_primitive_root.pow(bc.step)); py.push_back(bc.value); }
                if (px.empty()) {
#pragma omp parallel for schedule(static)
                    for (long i = 0; i < (long)N; ++i)
                        boundary_evaluation[i] += (b_alpha[col] * d_adj[i] + b_beta[col]) * lde_trace[(size_t)i * C + col];
                    continue;
                }
                // Lagrange interpolant evaluated pointwise: I(x) = sum_k y_k prod_{m!=k} (x - x_m)/(x_k - x_m)
                std::vector<Fp> wk(px.size());
                for (size_t k = 0; k < px.size(); ++k) {
                    Fp d = Fp::one();
                    for (size_t m = 0; m < px.size(); ++m) if (m != k) d = d * (px[k] - px[m]);
                    wk[k] = py[k] * d.inv();
                }
                std::vector<Fp> zinv(N);
                for (size_t i = 0; i < N; ++i) { Fp zz = Fp::one(); for (auto& q : px) zz = zz * (xs[i] - q); zinv[i] = zz; }
                batch_inverse(zinv);
#pragma omp parallel for schedule(static)
                for (long i = 0; i < (long)N; ++i) {
                    Fp I = Fp::zero();
                    for (size_t k = 0; k < px.size(); ++k) {
                        Fp t = wk[k];
                        for (size_t m = 0; m < px.size(); ++m) if (m != k) t = t * (xs[i] - px[m]);
                        I += t;
                    }
                    boundary_evaluation[i] += (b_alpha[col] * d_adj[i] + b_beta[col]) * (lde_trace[(size_t)i * C + col] - I) * zinv[i];
                }
            }
        }
    }
    // transition exemptions (traits.rs:49-79, evaluator.rs:299-323): unique nonzero exemption counts
    std::vector<size_t> uniq_ex;
    for (size_t e : ctx.transition_exemptions) if (e > 0 && std::find(uniq_ex.begin(), uniq_ex.end(), e) == uniq_ex.end()) uniq_ex.push_back(e);
    std::vector<std::vector<Fp>> exemption_evals;
    for (size_t cant : uniq_ex) {
        Poly ep(1, Fp::one());
        for (size_t k = 0; k < cant; ++k) {
            Fp root = domain.trace_roots_of_unity[n - 1 - k];
            Poly nx(ep.size() + 1, Fp::zero());
            for (size_t i = 0; i < ep.size(); ++i) { nx[i + 1] += ep[i]; nx[i] -= ep[i] * root; }
            ep = nx;
        }
        exemption_evals.push_back(evaluate_polynomial_on_lde_domain(ep, b, n, domain.coset_offset));
    }
    size_t max_deg = *std::max_element(ctx.transition_degrees.begin(), ctx.transition_degrees.end());
    std::vector<std::vector<Fp>> degree_adjustments(max_deg);
    for (size_t deg = 1; deg <= max_deg; ++deg) {
        degree_adjustments[deg - 1].resize(N);
        size_t e = D - n * (deg - 1);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)N; ++i) degree_adjustments[deg - 1][i] = xs[i].pow(e);
    }
    // zerofier 1/(x^n - 1): blowup distinct values (evaluator.rs:156-171)
    std::vector<Fp> zerofier = root_coset(log2_exact(b), b, domain.coset_offset.pow(n));
    for (auto& v : zerofier) v = v - Fp::one();
    batch_inverse(zerofier);

    std::vector<Fp> evaluations_acc(N);
    const size_t T = ctx.num_transition_constraints;
    const size_t nofs = ctx.transition_offsets.size();
#pragma omp parallel
    {
        std::vector<Fp> frame(nofs * C), cons(T);
#pragma omp for schedule(static)
        for (long i = 0; i < (long)N; ++i) {
            for (size_t r = 0; r < nofs; ++r) {
                size_t row = ((size_t)i + ctx.transition_offsets[r] * b) % N;  // frame.rs:40-59
                std::memcpy(&frame[r * C], &lde_trace[row * C], C * sizeof(Fp));
            }
            air.compute_transition(frame.data(), rap, cons.data());
            const Fp& zf = zerofier[(size_t)i % b];
            Fp acc = Fp::zero();
            for (size_t k = 0; k < T; ++k) {
                Fp term = zf * (t_alpha[k] * degree_adjustments[ctx.transition_degrees[k] - 1][i] + t_beta[k]) * cons[k];
                size_t ex = ctx.transition_exemptions[k];
                if (ex != 0) {
                    size_t idx = 0;
                    if (ctx.num_transition_exemptions != 1) idx = std::find(uniq_ex.begin(), uniq_ex.end(), ex) - uniq_ex.begin();
                    term = term * exemption_evals[idx][i];
                }
                acc += term;
            }
            evaluations_acc[i] = acc + boundary_evaluation[i];
        }
    }
    boundary_evaluation.clear(); boundary_evaluation.shrink_to_fit();
    degree_adjustments.clear(); exemption_evals.clear();
    // composition poly (evaluation_table.rs:27-33, prover.rs:250-286)
    Poly H = interpolate_offset_fft(evaluations_acc, domain.coset_offset);
    evaluations_acc.clear(); evaluations_acc.shrink_to_fit();
    Poly H1, H2;
    even_odd_decomposition(H, H1, H2);
    std::vector<Fp> H1_lde = evaluate_polynomial_on_lde_domain(H1, b, n, domain.coset_offset);
    std::vector<Fp> H2_lde = evaluate_polynomial_on_lde_domain(H2, b, n, domain.coset_offset);
    std::vector<Fp> comp_rows(2 * N);
    for (size_t i = 0; i < N; ++i) { comp_rows[2 * i] = H1_lde[i]; comp_rows[2 * i + 1] = H2_lde[i]; }
    MerkleTree comp_tree = MerkleTree::build_batched(comp_rows.data(), N, 2);
    transcript.append_digest(comp_tree.root);
    double t2 = now_s(); timings.round2 = t2 - t1;

    // ================= Round 3 (prover.rs:652-684, :288-325)
    Fp z = sample_z_ood(domain, transcript);
    Fp z2 = z.square();
    Fp H1_z2 = poly_eval(H1, z2), H2_z2 = poly_eval(H2, z2);
    std::vector<std::vector<Fp>> ood(nofs, std::vector<Fp>(C));
    for (size_t r = 0; r < nofs; ++r) {
        Fp pt = z * domain.trace_primitive_root.pow(ctx.transition_offsets[r]);
#pragma omp parallel for schedule(dynamic)
        for (long j = 0; j < (long)C; ++j) ood[r][j] = poly_eval(trace_polys[j], pt);
    }
    transcript.append_felt(H1_z2);
    transcript.append_felt(H2_z2);
    for (auto& row : ood) for (auto& e : row) transcript.append_felt(e);
    double t3 = now_s(); timings.round3 = t3 - t2;

    // ================= Round 4 (prover.rs:327-404)
    Fp gamma = transcript.to_field(), gamma_p = transcript.to_field();
    std::vector<Fp> trace_gammas = batch_sample_challenges(nofs * C, transcript);
    // compute_deep_composition_poly (prover.rs:410-482)
    Poly deep;
    {
        Poly h1t = poly_scale(poly_sub_const(H1, H1_z2), gamma);
        ruffini_division_inplace(h1t, z2);
        Poly h2t = poly_scale(poly_sub_const(H2, H2_z2), gamma_p);
        ruffini_division_inplace(h2t, z2);
        std::vector<Poly> per_col(C);
#pragma omp parallel for schedule(dynamic)
        for (long j = 0; j < (long)C; ++j) {
            Poly agg;
            for (size_t r = 0; r < nofs; ++r) {
                Fp zs = z * domain.trace_primitive_root.pow(ctx.transition_offsets[r]);
                Poly q = poly_sub_const(trace_polys[j], ood[r][j]);
                ruffini_division_inplace(q, zs);
                agg = poly_add(agg, poly_scale(q, trace_gammas[(size_t)j * nofs + r]));
            }
            per_col[j] = agg;
        }
        Poly trace_term;
        for (size_t j = 0; j < C; ++j) trace_term = poly_add(trace_term, per_col[j]);
        deep = poly_add(poly_add(h1t, h2t), trace_term);
    }
    // fri_commit_phase (fri/mod.rs:20-72)
    struct Layer { std::vector<Fp> evaluation; MerkleTree tree; size_t domain_size; };
    std::vector<Layer> layers;
    size_t number_layers = domain.root_order;
    Fp fri_last_value;
    {
        size_t dsize = N;
        Fp offs = domain.coset_offset;
        Poly cur = deep;
        auto make_layer = [&](const Poly& p, const Fp& off, size_t ds) {
            Layer L;
            L.evaluation = evaluate_offset_fft(p, 1, ds, off);
            L.tree = MerkleTree::build_single(L.evaluation.data(), L.evaluation.size());
            L.domain_size = ds;
            return L;
        };
        layers.push_back(make_layer(cur, offs, dsize));
        transcript.append_digest(layers.back().tree.root);
        auto fold = [&](const Poly& p, const Fp& beta) {  // fri_functions.rs:4-27
            Poly even, odd;
            for (size_t i = 0; i < p.size(); ++i) { if (i & 1) odd.push_back(p[i] * beta); else even.push_back(p[i]); }
            return poly_add(even, odd);
        };
        for (size_t k = 1; k < number_layers; ++k) {
            Fp zeta = transcript.to_field();
            offs = offs.square();
            dsize /= 2;
            cur = fold(cur, zeta);
            layers.push_back(make_layer(cur, offs, dsize));
            transcript.append_digest(layers.back().tree.root);
        }
        Fp zeta = transcript.to_field();
        Poly last = fold(cur, zeta);
        fri_last_value = last.empty() ? Fp::zero() : last[0];
        transcript.append_felt(fri_last_value);
    }
    // grinding (prover.rs:380-385)
    Digest gch = transcript.challenge();
    uint64_t nonce = grinding_nonce(gch, ctx.proof_options.grinding_factor);
    {
        uint8_t nb[8];
        for (int i = 0; i < 8; ++i) nb[i] = (uint8_t)(nonce >> (56 - 8 * i));
        transcript.append(nb, 8);
    }
    // fri_query_phase (fri/mod.rs:74-127)
    StarkProof proof;
    std::vector<size_t> iotas;
    if (!layers.empty()) {
        for (size_t s = 0; s < ctx.proof_options.fri_number_of_queries; ++s) iotas.push_back((size_t)(transcript.to_usize() % N));
        for (size_t iota : iotas) {
            FriDecommitment q;
            for (auto& L : layers) {
                size_t index = iota % L.domain_size;
                size_t index_sym = (iota + L.domain_size / 2) % L.domain_size;
                q.layers_auth_paths_sym.push_back(L.tree.proof(index_sym));
                q.layers_evaluations_sym.push_back(L.evaluation[index_sym]);
                q.layers_evaluations.push_back(L.evaluation[index]);
                q.layers_auth_paths.push_back(L.tree.proof(index));
            }
            proof.query_list.push_back(std::move(q));
        }
    }
    // open_deep_composition_poly (prover.rs:484-529)
    for (size_t iota : iotas) {
        size_t index = iota % N;
        DeepPolynomialOpenings o;
        o.lde_composition_poly_proof = comp_tree.proof(index);
        o.lde_composition_poly_even_evaluation = H1_lde[index];
        o.lde_composition_poly_odd_evaluation = H2_lde[index];
        for (auto& t : trace_trees) o.lde_trace_merkle_proofs.push_back(t.proof(index));
        o.lde_trace_evaluations.assign(&lde_trace[index * C], &lde_trace[index * C] + C);
        proof.deep_poly_openings.push_back(std::move(o));
    }
    double t4 = now_s(); timings.round4 = t4 - t3;

    proof.trace_length = n;
    proof.lde_trace_merkle_roots = trace_roots;
    for (auto& row : ood) for (auto& e : row) proof.trace_ood_frame_data.push_back(e);
    proof.trace_ood_row_width = C;
    proof.composition_poly_root = comp_tree.root;
    proof.composition_poly_even_ood_evaluation = H1_z2;
    proof.composition_poly_odd_ood_evaluation = H2_z2;
    for (auto& L : layers) proof.fri_layers_merkle_roots.push_back(L.tree.root);
    proof.fri_last_value = fri_last_value;
    proof.nonce = nonce;
    return proof;
}

// ---------------------------------------------------------------- verifier (verifier.rs:59-657)
inline bool verify(const Air& air, const StarkProof& proof) {
    const AirContext& ctx = air.ctx;
    if (proof.query_list.size() < ctx.proof_options.fri_number_of_queries) return false;
    if (proof.trace_length != air.trace_len) return false;
    Domain domain(air);
    const size_t n = air.trace_len, N = n * domain.blowup_factor, C = ctx.trace_columns;
    const size_t nofs = ctx.transition_offsets.size();
    if (proof.trace_ood_frame_data.size() != nofs * C || proof.lde_trace_merkle_roots.empty()) return false;
    // the reference takes the frame's rows by the row width the proof states (verifier.rs:136-137, 237, 533-541): with any other width
    // than the AIR's it replays a different transcript or indexes past a row - never an accepted proof
    if (proof.trace_ood_row_width != C) return false;
    Transcript t;
    // step 1
    t.append_digest(proof.lde_trace_merkle_roots[0]);
    std::vector<Fp> rap = air.build_rap_challenges(t);
    if (proof.lde_trace_merkle_roots.size() > 1) t.append_digest(proof.lde_trace_merkle_roots[1]);
    std::vector<BoundaryConstraint> bcs = air.boundary_constraints(rap);
    std::vector<Fp> b_alpha = batch_sample_challenges(bcs.size(), t), b_beta = batch_sample_challenges(bcs.size(), t);
    std::vector<Fp> t_alpha = batch_sample_challenges(ctx.num_transition_constraints, t);
    std::vector<Fp> t_beta = batch_sample_challenges(ctx.num_transition_constraints, t);
    t.append_digest(proof.composition_poly_root);
    Fp z = sample_z_ood(domain, t);
    t.append_felt(proof.composition_poly_even_ood_evaluation);
    t.append_felt(proof.composition_poly_odd_ood_evaluation);
    for (auto& e : proof.trace_ood_frame_data) t.append_felt(e);
    Fp gamma_even = t.to_field(), gamma_odd = t.to_field();
    std::vector<Fp> trace_term_coeffs = batch_sample_challenges(C * nofs, t);  // [col][row]
    std::vector<Fp> zetas;
    for (auto& r : proof.fri_layers_merkle_roots) { t.append_digest(r); zetas.push_back(t.to_field()); }
    t.append_felt(proof.fri_last_value);
    Digest gch = t.challenge();
    uint8_t lz = grinding_trailing_zeros(gch, proof.nonce);
    {
        uint8_t nb[8];
        for (int i = 0; i < 8; ++i) nb[i] = (uint8_t)(proof.nonce >> (56 - 8 * i));
        t.append(nb, 8);
    }
    std::vector<size_t> iotas;
    for (size_t s = 0; s < ctx.proof_options.fri_number_of_queries; ++s) iotas.push_back((size_t)(t.to_usize() % N));
    if (lz < ctx.proof_options.grinding_factor) return false;

    // step 2 (verifier.rs:208-317)
    {
        size_t D = air.composition_poly_degree_bound();
        Fp bdz = z.pow(D - n);
        Fp bq = Fp::zero();
        for (size_t k = 0; k < bcs.size(); ++k) {
            Fp point = domain.trace_primitive_root.pow(bcs[k].step);
            Fp num = proof.trace_ood_frame_data[bcs[k].col] - bcs[k].value;
            Fp den = (z - point).inv();
            bq += num * den * (b_alpha[k] * bdz + b_beta[k]);
        }
        std::vector<Fp> cons(ctx.num_transition_constraints);
        air.compute_transition(proof.trace_ood_frame_data.data(), rap, cons.data());
        Fp denominator = (z.pow(n) - Fp::one()).inv();
        size_t max_ex = *std::max_element(ctx.transition_exemptions.begin(), ctx.transition_exemptions.end());
        Fp last_root = domain.trace_roots_of_unity[n - 1];
        std::vector<Fp> exemption;  // transition_exemptions_verifier (traits.rs:97-118)
        for (size_t idx = 1; idx <= max_ex; ++idx) {
            Fp v = Fp::one();
            for (size_t k = 1; k <= idx; ++k) v = v * (z - last_root.pow(k));
            exemption.push_back(v);
        }
        size_t max_deg = *std::max_element(ctx.transition_degrees.begin(), ctx.transition_degrees.end());
        std::vector<Fp> dadj;
        for (size_t d = 1; d <= max_deg; ++d) dadj.push_back(z.pow(D - n * (d - 1)));
        Fp sum = Fp::zero();
        for (size_t k = 0; k < ctx.num_transition_constraints; ++k) {
            Fp ex = ctx.transition_exemptions[k] ? exemption[ctx.transition_exemptions[k] - 1] : Fp::one();
            sum += denominator * cons[k] * (t_alpha[k] * dadj[ctx.transition_degrees[k] - 1] + t_beta[k]) * ex;
        }
        Fp lhs = proof.composition_poly_even_ood_evaluation + z * proof.composition_poly_odd_ood_evaluation;
        if (lhs != bq + sum) return false;
    }
    // step 3 (verifier.rs:319-356, :443-523)
    if (proof.query_list.size() != iotas.size() && proof.query_list.size() < iotas.size()) return false;
    Fp two_inv = Fp::from_u64(2).inv();
    bool ok = true;
    for (size_t s = 0; s < iotas.size(); ++s) {
        const FriDecommitment& q = proof.query_list[s];
        size_t L = proof.fri_layers_merkle_roots.size();
        if (q.layers_auth_paths.size() != L || q.layers_evaluations.size() != L || q.layers_auth_paths_sym.size() != L || q.layers_evaluations_sym.size() != L) return false;
        size_t iota = iotas[s];
        Fp ep_inv = domain.lde_roots_of_unity_coset[iota].inv();
        Fp v = q.layers_evaluations[0];
        for (size_t k = 0; k < L; ++k) {
            size_t dl = size_t(1) << (domain.lde_root_order - k);
            size_t isym = (iota + dl / 2) % dl;
            ok &= merkle_verify(q.layers_auth_paths_sym[k], proof.fri_layers_merkle_roots[k], isym, &q.layers_evaluations_sym[k], 1, true);
            ok &= merkle_verify(q.layers_auth_paths[k], proof.fri_layers_merkle_roots[k], iota, &q.layers_evaluations[k], 1, true);
            const Fp& es = q.layers_evaluations_sym[k];
            v = (v + es) * two_inv + zetas[k] * (v - es) * two_inv * ep_inv;
            ep_inv = ep_inv.square();
            if (k + 1 < L) ok &= (v == q.layers_evaluations[k + 1]);
            else ok &= (v == proof.fri_last_value);
        }
    }
    if (!ok) return false;
    // step 4 (verifier.rs:358-441, :525-557) — unlike the reference, the trace-opening Merkle checks are NOT discarded
    if (proof.deep_poly_openings.size() < iotas.size()) return false;
    size_t aux_cols = air.number_auxiliary_rap_columns();
    size_t main_cols = C - aux_cols;
    Fp z2 = z.square();
    for (size_t s = 0; s < iotas.size(); ++s) {
        size_t iota = iotas[s];
        const DeepPolynomialOpenings& o = proof.deep_poly_openings[s];
        if (o.lde_trace_evaluations.size() != C) return false;
        Fp hh[2] = {o.lde_composition_poly_even_evaluation, o.lde_composition_poly_odd_evaluation};
        ok &= merkle_verify(o.lde_composition_poly_proof, proof.composition_poly_root, iota, hh, 2);
        if (o.lde_trace_merkle_proofs.size() != proof.lde_trace_merkle_roots.size()) return false;
        ok &= merkle_verify(o.lde_trace_merkle_proofs[0], proof.lde_trace_merkle_roots[0], iota, o.lde_trace_evaluations.data(), main_cols);
        if (proof.lde_trace_merkle_roots.size() > 1)
            ok &= merkle_verify(o.lde_trace_merkle_proofs[1], proof.lde_trace_merkle_roots[1], iota, o.lde_trace_evaluations.data() + main_cols, aux_cols);
        const Fp& x = domain.lde_roots_of_unity_coset[iota];
        Fp denom_inv = (x - z2).inv();
        Fp trace_term = Fp::zero();
        for (size_t r = 0; r < nofs; ++r) {
            Fp div = (x - z * domain.trace_primitive_root.pow(r)).inv();
            for (size_t j = 0; j < C; ++j)
                trace_term += (o.lde_trace_evaluations[j] - proof.trace_ood_frame_data[r * C + j]) * div * trace_term_coeffs[j * nofs + r];
        }
        Fp h1 = (o.lde_composition_poly_even_evaluation - proof.composition_poly_even_ood_evaluation) * denom_inv;
        Fp h2 = (o.lde_composition_poly_odd_evaluation - proof.composition_poly_odd_ood_evaluation) * denom_inv;
        Fp deep = trace_term + h1 * gamma_even + h2 * gamma_odd;
        ok &= (deep == proof.query_list[s].layers_evaluations[0]);
    }
    return ok;
}

}  // namespace oracle

// CPU verifier shipped with the library: `verify` of reference src/starks/verifier.rs:559-657
// (steps 1-4 :59-557) for the Cairo AIR, plus the CLI proof-file framing of reference src/main.rs:98-102.
// O(queries * log N) hashes — host code, no device work. Unlike the reference (verifier.rs:411-422 discards the fold
// result) the trace-opening Merkle checks are enforced.
#include "cairo_air_host.h"
#include "common.h"
#include "keccak.h"
#include "poseidon.h"
#include <algorithm>
#include <array>
#include <cstring>
#include <functional>
#include <stdexcept>

namespace sp {

// CairoAIR::compute_transition on a 2-row out-of-domain frame (reference src/cairo/air.rs:743-767, helpers :869-1160)
void cairo_transition_host(const fe* frame, uint32_t cols, bool has_rc, const fe rap[3], fe* c) {
    const fe* cur = frame;
    const fe* nxt = frame + cols;
    const uint32_t A = has_rc ? 43 : 34;
    const fe one = fe_one(), two = fe_from_u64(2);
    auto add = [](const fe& a, const fe& b) { return fe_add(a, b); };
    auto sub = [](const fe& a, const fe& b) { return fe_sub(a, b); };
    auto mul = [](const fe& a, const fe& b) { return fe_mul(a, b); };
    for (int k = 0; k < 15; ++k) c[k] = mul(cur[k], sub(cur[k], one));
    c[15] = cur[15];
    const fe b16 = fe_from_u64(1ULL << 16), b32 = fe_from_u64(1ULL << 32), b48 = fe_from_u64(1ULL << 48), b15 = fe_from_u64(1ULL << 15);
    fe f0s = fe_zero();
    for (int k = 14; k >= 0; --k) f0s = add(cur[k], add(f0s, f0s));
    c[16] = sub(add(add(add(cur[27], mul(b16, cur[28])), mul(b32, cur[29])), mul(b48, f0s)), cur[23]);
    const fe &ap = cur[17], &fp = cur[18], &pc = cur[19];
    c[17] = sub(add(add(mul(cur[0], fp), mul(sub(one, cur[0]), ap)), sub(cur[27], b15)), cur[20]);
    c[18] = sub(add(add(mul(cur[1], fp), mul(sub(one, cur[1]), ap)), sub(cur[28], b15)), cur[21]);
    c[19] = sub(add(add(add(add(mul(cur[2], pc), mul(cur[4], ap)), mul(cur[3], fp)),
                        mul(sub(sub(sub(one, cur[2]), cur[4]), cur[3]), cur[25])), sub(cur[29], b15)), cur[22]);
    const fe size = add(cur[2], one);
    c[20] = sub(add(add(add(ap, mul(cur[10], cur[16])), cur[11]), mul(cur[12], two)), nxt[17]);
    c[21] = sub(add(add(mul(cur[13], cur[24]), mul(cur[12], add(ap, two))), mul(sub(sub(one, cur[13]), cur[12]), fp)), nxt[18]);
    c[22] = mul(sub(cur[31], cur[9]), sub(nxt[19], add(pc, size)));
    c[23] = sub(add(mul(cur[30], sub(nxt[19], add(pc, cur[26]))), mul(sub(one, cur[9]), nxt[19])),
                add(add(mul(sub(sub(sub(one, cur[7]), cur[8]), cur[9]), add(pc, size)), mul(cur[7], cur[16])), mul(cur[8], add(pc, cur[16]))));
    c[24] = sub(mul(cur[9], cur[24]), cur[30]);
    c[25] = sub(mul(cur[30], cur[16]), cur[31]);
    c[26] = sub(cur[32], mul(cur[25], cur[26]));
    c[27] = sub(add(add(mul(cur[5], add(cur[25], cur[26])), mul(cur[6], cur[32])), mul(sub(sub(sub(one, cur[5]), cur[6]), cur[9]), cur[26])),
                mul(sub(one, cur[9]), cur[16]));
    c[28] = mul(cur[12], sub(cur[24], fp));
    c[29] = mul(cur[12], sub(cur[25], add(pc, size)));
    c[30] = mul(cur[14], sub(cur[24], cur[16]));
    for (int k = 16; k <= 30; ++k) c[k] = mul(c[k], cur[33]);
    const fe &alpha = rap[0], &z = rap[1], &zrc = rap[2];
    const fe* as = cur + A + 3; const fe* vs = cur + A + 7; const fe* pp = cur + A + 11;
    const fe &as0n = nxt[A + 3], &vs0n = nxt[A + 7], &p0n = nxt[A + 11];
    for (int k = 0; k < 3; ++k) {
        fe step = sub(sub(as[k + 1], as[k]), one);
        c[31 + k] = mul(sub(as[k], as[k + 1]), step);
        c[35 + k] = mul(sub(vs[k], vs[k + 1]), step);
        c[39 + k] = sub(mul(sub(z, add(as[k + 1], mul(alpha, vs[k + 1]))), pp[k + 1]), mul(sub(z, add(cur[20 + k], mul(alpha, cur[24 + k]))), pp[k]));
    }
    {
        fe step = sub(sub(as0n, as[3]), one);
        c[34] = mul(sub(as[3], as0n), step);
        c[38] = mul(sub(vs[3], vs0n), step);
        c[42] = sub(mul(sub(z, add(as0n, mul(alpha, vs0n))), p0n), mul(sub(z, add(nxt[19], mul(alpha, nxt[23]))), pp[3]));
    }
    const fe* rc = cur + A; const fe& rc0n = nxt[A]; const fe* q = cur + A + 15; const fe& q0n = nxt[A + 15];
    c[43] = mul(sub(rc[0], rc[1]), sub(sub(rc[1], rc[0]), one));
    c[44] = mul(sub(rc[1], rc[2]), sub(sub(rc[2], rc[1]), one));
    c[45] = mul(sub(rc[2], rc0n), sub(sub(rc0n, rc[2]), one));
    c[46] = sub(mul(sub(zrc, rc[1]), q[1]), mul(sub(zrc, cur[28]), q[0]));
    c[47] = sub(mul(sub(zrc, rc[2]), q[2]), mul(sub(zrc, cur[29]), q[1]));
    c[48] = sub(mul(sub(zrc, rc0n), q0n), mul(sub(zrc, nxt[27]), q[2]));
    if (has_rc) {
        fe acc = fe_zero();
        for (int k = 7; k >= 0; --k) acc = add(mul(acc, b16), cur[34 + k]);
        c[49] = sub(acc, cur[42]);
    }
}

namespace {
typedef std::array<uint8_t, 32> Dig;
struct Reader {
    const uint8_t* p; size_t n, pos = 0;
    Reader(const uint8_t* d, size_t len) : p(d), n(len) {}
    void need(size_t k) { if (k > n - pos) throw std::runtime_error("malformed: InvalidAmountOfBytes"); }
    uint64_t u64() { need(8); uint64_t v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | p[pos + i]; pos += 8; return v; }
    uint64_t count(size_t unit) { uint64_t k = u64(); if (k > (n - pos) / (unit ? unit : 1)) throw std::runtime_error("malformed: a length field exceeds the proof"); return k; }
    fe felt() {
        need(32);
        // reject non-canonical encodings (>= p)
        static const uint8_t PBE[32] = {0x08, 0, 0, 0, 0, 0, 0, 0x11, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
        if (std::memcmp(p + pos, PBE, 32) >= 0) throw std::runtime_error("malformed: field element out of range");
        fe x = fe_from_bytes_be(p + pos); pos += 32; return x;
    }
    Dig dig() { need(32); Dig d; std::memcpy(d.data(), p + pos, 32); pos += 32; return d; }
    std::vector<Dig> path() { uint64_t k = count(32); std::vector<Dig> v(k); for (auto& d : v) d = dig(); return v; }
};
struct FriDecommitment { std::vector<std::vector<Dig>> paths_sym, paths; std::vector<fe> evals_sym, evals; };
struct Opening { std::vector<Dig> comp_path; fe h1, h2; std::vector<std::vector<Dig>> trace_paths; std::vector<fe> trace_evals; };
struct Proof {
    uint64_t trace_length; std::vector<Dig> trace_roots; std::vector<fe> ood; uint64_t row_width; Dig comp_root; fe h1z, h2z;
    std::vector<Dig> fri_roots; fe fri_last; std::vector<FriDecommitment> queries; std::vector<Opening> openings; uint64_t nonce;
};
// The wire format of StarkProof::serialize (reference proof/stark.rs:161-218), accepted in exactly the form that function writes:
// every length prefix must equal the bytes its part occupies, every element length must be 32, elements must be canonical (< p) and
// the nonce must end the buffer.  The reference's own deserializer (stark.rs:225-440) parses each part inside the slice its prefix
// announces and takes the nonce from the last eight bytes, so it tolerates padding inside and behind the parts; a proof that uses that
// freedom is refused here.  What both refuse: a prefix that disagrees with its part.
Proof parse(const uint8_t* data, size_t len) {
    Reader r(data, len);
    Proof p;
    auto slice = [&](void) { const uint64_t l = r.u64(); r.need(l); return r.pos + (size_t)l; };          // end position of a length-prefixed part
    // (messages that start with "non-canonical framing:" mark what ONLY this strict parser refuses - sp_last_error() after a 0 from
    // sp_cairo_verify / sp_air_verify lets a caller that re-frames proofs tell that from an invalid proof, include/stark252_hip.h)
    auto close = [&](size_t end) {
        if (r.pos < end) throw std::runtime_error("non-canonical framing: padding inside a length-prefixed part");
        if (r.pos > end) throw std::runtime_error("malformed: a part overruns its length prefix");
    };
    auto felt_len = [&](void) { if (r.u64() != 32) throw std::runtime_error("non-canonical framing: element length is not 32"); };
    p.trace_length = r.u64();
    uint64_t nr = r.count(32);
    for (uint64_t i = 0; i < nr; ++i) p.trace_roots.push_back(r.dig());
    {
        const size_t end = slice();                        // Frame (frame.rs:93-111): count, element length, elements, row width
        uint64_t ne = r.count(32); felt_len();
        for (uint64_t i = 0; i < ne; ++i) p.ood.push_back(r.felt());
        p.row_width = r.u64();
        close(end);
    }
    p.comp_root = r.dig(); felt_len();
    p.h1z = r.felt(); p.h2z = r.felt();
    uint64_t nf = r.count(32);
    for (uint64_t i = 0; i < nf; ++i) p.fri_roots.push_back(r.dig());
    p.fri_last = r.felt();
    uint64_t nq = r.count(8);
    for (uint64_t i = 0; i < nq; ++i) {
        const size_t end = slice();                        // FriDecommitment (fri_decommit.rs:20-51)
        FriDecommitment q;
        uint64_t k = r.count(8); for (uint64_t j = 0; j < k; ++j) q.paths_sym.push_back(r.path());
        felt_len();
        k = r.count(32); for (uint64_t j = 0; j < k; ++j) q.evals_sym.push_back(r.felt());
        k = r.count(32); for (uint64_t j = 0; j < k; ++j) q.evals.push_back(r.felt());
        k = r.count(8); for (uint64_t j = 0; j < k; ++j) q.paths.push_back(r.path());
        close(end);
        p.queries.push_back(std::move(q));
    }
    uint64_t no = r.count(8);
    for (uint64_t i = 0; i < no; ++i) {
        const size_t end = slice();                        // DeepPolynomialOpenings (stark.rs:49-82)
        Opening o;
        o.comp_path = r.path(); felt_len();
        o.h1 = r.felt(); o.h2 = r.felt();
        uint64_t k = r.count(8); for (uint64_t j = 0; j < k; ++j) o.trace_paths.push_back(r.path());
        k = r.count(32); for (uint64_t j = 0; j < k; ++j) o.trace_evals.push_back(r.felt());
        close(end);
        p.openings.push_back(std::move(o));
    }
    p.nonce = r.u64();
    if (r.pos != len) throw std::runtime_error("non-canonical framing: trailing bytes behind the nonce");
    return p;
}
struct Tr {
    std::vector<uint8_t> buf;
    void append(const uint8_t* d, size_t n) { buf.insert(buf.end(), d, d + n); }
    void felt(const fe& x) { uint8_t b[32]; fe_to_bytes_be(x, b); append(b, 32); }
    void challenge(uint8_t out[32]) { uint8_t d[32]; sp_keccak256_host(buf.data(), buf.size(), d); for (int i = 0; i < 32; ++i) out[i] = d[31 - i]; buf.assign(out, out + 32); }
    fe field() { uint8_t r[32]; challenge(r); r[0] &= 0x07; return fe_from_bytes_be(r); }
    uint64_t usize() { uint8_t r[32]; challenge(r); uint64_t v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | r[i]; return v; }
};
// the hash of the commitments being checked (sp_*_verify_backend; thread-local: the entry points are context-free)
thread_local int t_merkle_backend = SP_MERKLE_KECCAK256;
Dig poseidon_dig(const fe& h) {
    Dig d; uint64_t w[4];
    poseidon_digest_from_fe(h, w);
    std::memcpy(d.data(), w, 32);
    return d;
}
fe poseidon_fe(const Dig& d) {
    uint64_t w[4];
    std::memcpy(w, d.data(), 32);
    return poseidon_fe_from_digest(w);
}
// A Poseidon digest is the canonical big-endian encoding of a field element: bytes >= p (d and d + p name the same element) are
// not a digest.  lambdaworks' Poseidon trees carry field elements, so only canonical encodings exist there; accepting the others
// would make proofs of this backend byte-malleable, which the Keccak256 backend is not.
bool poseidon_digest_canonical(const Dig& d) {
    static const uint8_t P_BE[32] = {0x08, 0, 0, 0, 0, 0, 0, 0x11, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x01};
    return std::memcmp(d.data(), P_BE, 32) < 0;
}
Dig hash_felts(const fe* v, size_t k, bool single_element_tree) {
    if (t_merkle_backend == SP_MERKLE_POSEIDON) return poseidon_dig(single_element_tree ? poseidon_hash1(v[0]) : poseidon_hash_many(v, 1, (uint32_t)k));
    std::vector<uint8_t> b(32 * k);
    for (size_t i = 0; i < k; ++i) fe_to_bytes_be(v[i], &b[32 * i]);
    Dig d; sp_keccak256_host(b.data(), b.size(), d.data()); return d;
}
bool merkle_ok(const std::vector<Dig>& path, const Dig& root, uint64_t index, const fe* v, size_t k, bool single_element_tree = false) {
    Dig h = hash_felts(v, k, single_element_tree);
    for (const Dig& s : path) {
        if (t_merkle_backend == SP_MERKLE_POSEIDON) {
            if (!poseidon_digest_canonical(s)) throw std::runtime_error("non-canonical framing: a Poseidon path digest is not below p");
            h = poseidon_dig((index & 1) ? poseidon_hash2(poseidon_fe(s), poseidon_fe(h)) : poseidon_hash2(poseidon_fe(h), poseidon_fe(s)));
            index >>= 1;
            continue;
        }
        uint8_t b[64];
        if (index & 1) { std::memcpy(b, s.data(), 32); std::memcpy(b + 32, h.data(), 32); } else { std::memcpy(b, h.data(), 32); std::memcpy(b + 32, s.data(), 32); }
        sp_keccak256_host(b, 64, h.data());
        index >>= 1;
    }
    return h == root;
}
}  // namespace

void set_verify_merkle_backend(int backend) { t_merkle_backend = backend; }

// What `verify` needs from an AIR (reference src/starks/traits.rs:15-119).
struct VerifySpec {
    uint32_t main_cols = 0, aux_cols = 0;
    std::vector<uint32_t> offsets, degrees, exemptions;
    uint32_t bound_factor = 2;   // composition_poly_degree_bound / trace_length
    uint32_t n_rap = 0;
    std::function<std::vector<BoundaryConstraint>(const std::vector<fe>& rap)> boundary;
    std::function<void(const fe* frame /*[rows][C]*/, const std::vector<fe>& rap, fe* out)> transition;
};

// returns 1 accept, 0 reject; throws on malformed input
static int verify_host(const uint8_t* proof_bytes, size_t len, const VerifySpec& air, uint8_t blowup, uint64_t queries, uint64_t coset_offset, uint8_t grinding) {
    Proof pr = parse(proof_bytes, len);
    if (pr.queries.size() < queries) return 0;
    const uint64_t n = pr.trace_length;
    int k = sp_log2_exact(n), lb = sp_log2_exact(blowup);
    if (k < 1 || lb < 1 || k + lb > 40) return 0;
    const uint32_t C = air.main_cols + air.aux_cols, T = (uint32_t)air.degrees.size(), R = (uint32_t)air.offsets.size();
    const uint32_t f = air.bound_factor;
    const size_t n_roots = air.aux_cols ? 2 : 1;
    const uint64_t N = n << lb;
    // (the reference slices the out-of-domain frame by the row width the PROOF states, verifier.rs:136-137, 533-541: any other width than
    // the AIR's makes it replay a different transcript)
    if (pr.ood.size() != (size_t)R * C || pr.row_width != C || pr.trace_roots.size() != n_roots || pr.fri_roots.size() != (size_t)k) return 0;
    for (uint32_t c = 0; c < T; ++c) if (air.degrees[c] < 1 || air.degrees[c] > f + 1 || air.exemptions[c] >= n) return 0;
    const fe h = fe_from_u64(coset_offset), hinv = fe_inv(h);
    auto root_of = [&](int order) { fe w = fe_from_bytes_be((const uint8_t*)"\x00\x52\x82\xdb\x87\x52\x9c\xfa\x3f\x04\x64\x51\x9c\x8b\x0f\xa5\xad\x18\x71\x48\xe1\x1a\x61\x61\x60\x70\x02\x4f\x42\xf8\xef\x94"); for (int i = order; i < 192; ++i) w = fe_sqr(w); return w; };
    const fe g = root_of(k), w = root_of(k + lb);
    // ---- step 1: replay the transcript (verifier.rs:59-206)
    Tr t;
    t.append(pr.trace_roots[0].data(), 32);
    std::vector<fe> rap(air.n_rap);
    for (auto& x : rap) x = t.field();
    if (n_roots > 1) t.append(pr.trace_roots[1].data(), 32);
    std::vector<BoundaryConstraint> bcs = air.boundary(rap);
    std::vector<fe> ba(bcs.size()), bb(bcs.size()), ta(T), tb(T);
    for (auto& x : ba) x = t.field();
    for (auto& x : bb) x = t.field();
    for (auto& x : ta) x = t.field();
    for (auto& x : tb) x = t.field();
    t.append(pr.comp_root.data(), 32);
    fe z;
    for (;;) {
        z = t.field();
        fe a = fe_mul(z, hinv), b = z;
        for (int i = 0; i < k + lb; ++i) a = fe_sqr(a);
        for (int i = 0; i < k; ++i) b = fe_sqr(b);
        if (!fe_eq(a, fe_one()) && !fe_eq(b, fe_one())) break;
    }
    t.felt(pr.h1z); t.felt(pr.h2z);
    for (auto& e : pr.ood) t.felt(e);
    fe gamma = t.field(), gamma_p = t.field();
    std::vector<fe> tg((size_t)R * C);
    for (auto& x : tg) x = t.field();
    std::vector<fe> zetas;
    for (auto& r : pr.fri_roots) { t.append(r.data(), 32); zetas.push_back(t.field()); }
    t.felt(pr.fri_last);
    uint8_t gch[32];
    t.challenge(gch);
    {
        uint8_t data[40], dg[32];
        std::memcpy(data, gch, 32);
        for (int i = 0; i < 8; ++i) data[32 + i] = (uint8_t)(pr.nonce >> (8 * i));
        sp_keccak256_host(data, 40, dg);
        uint64_t head = 0;
        for (int i = 0; i < 8; ++i) head = (head << 8) | dg[i];
        int tz = head == 0 ? 64 : __builtin_ctzll(head);
        uint8_t nb[8];
        for (int i = 0; i < 8; ++i) nb[i] = (uint8_t)(pr.nonce >> (56 - 8 * i));
        t.append(nb, 8);
        if (tz < (int)grinding) return 0;
    }
    std::vector<uint64_t> iotas(queries);
    for (auto& x : iotas) x = t.usize() % N;
    // ---- step 2: composition polynomial at z (verifier.rs:208-317)
    {
        fe zn = fe_pow_u64(z, n);
        fe bdz = fe_pow_u64(zn, f - 1);                       // z^(D - n)
        fe bq = fe_zero();
        for (size_t j = 0; j < bcs.size(); ++j) {
            if (bcs[j].col >= C) return 0;
            fe den = fe_sub(z, fe_pow_u64(g, bcs[j].step));
            if (fe_is_zero(den)) return 0;
            fe num = fe_sub(pr.ood[bcs[j].col], bcs[j].value);
            bq = fe_add(bq, fe_mul(fe_mul(num, fe_inv(den)), fe_add(fe_mul(ba[j], bdz), bb[j])));
        }
        std::vector<fe> cons(T);
        air.transition(pr.ood.data(), rap, cons.data());
        fe zden = fe_sub(zn, fe_one());
        if (fe_is_zero(zden)) return 0;
        fe zf = fe_inv(zden);
        // transition_exemptions_verifier (traits.rs:97-118): ex_e(z) = prod_{i=1..e} (z - g^(n-i))
        uint32_t max_ex = 0;
        for (uint32_t e : air.exemptions) max_ex = std::max(max_ex, e);
        std::vector<fe> ex(max_ex + 1, fe_one());
        for (uint32_t e = 1; e <= max_ex; ++e) ex[e] = fe_mul(ex[e - 1], fe_sub(z, fe_pow_u64(g, n - e)));
        fe sum = fe_zero();
        for (uint32_t c = 0; c < T; ++c) {
            fe adj = fe_pow_u64(zn, f - air.degrees[c] + 1);   // z^(D - n (deg - 1))
            fe term = fe_mul(fe_mul(zf, cons[c]), fe_add(fe_mul(ta[c], adj), tb[c]));
            term = fe_mul(term, ex[air.exemptions[c]]);
            sum = fe_add(sum, term);
        }
        if (!fe_eq(fe_add(pr.h1z, fe_mul(z, pr.h2z)), fe_add(bq, sum))) return 0;
    }
    // ---- step 3: FRI (verifier.rs:319-356, :443-523)
    const fe half = fe_inv(fe_from_u64(2));
    const size_t L = pr.fri_roots.size();
    // The queries are independent: checked on the host's threads (80 queries of a 2^20-row proof are ~10^5 hashes - 15 ms of Keccak
    // or 250 ms of Poseidon on one thread).  status per query: 1 accepted, 0 a check failed, -1 malformed (the verdict is reject
    // either way; the counts keep the single-threaded order of the decisions irrelevant).
    const int backend = t_merkle_backend;
    std::vector<int8_t> st_fri(queries, 1), st_deep(queries, 1);
    host_parallel_for(queries, 1, [&](size_t b, size_t e) {
        t_merkle_backend = backend;   // (thread-local: the workers of this call take the caller's setting)
        for (size_t s = b; s < e; ++s) {
            const FriDecommitment& q = pr.queries[s];
            if (q.paths.size() != L || q.paths_sym.size() != L || q.evals.size() != L || q.evals_sym.size() != L) { st_fri[s] = -1; continue; }
            bool ok = true;
            fe xinv = fe_inv(fe_mul(h, fe_pow_u64(w, iotas[s])));
            fe v = q.evals[0];
            for (size_t l = 0; l < L; ++l) {
                uint64_t dl = N >> l, isym = (iotas[s] + dl / 2) % dl;
                ok &= merkle_ok(q.paths_sym[l], pr.fri_roots[l], isym, &q.evals_sym[l], 1, true);
                ok &= merkle_ok(q.paths[l], pr.fri_roots[l], iotas[s], &q.evals[l], 1, true);
                const fe& es = q.evals_sym[l];
                v = fe_add(fe_mul(fe_add(v, es), half), fe_mul(fe_mul(fe_mul(zetas[l], fe_sub(v, es)), half), xinv));
                xinv = fe_sqr(xinv);
                ok &= fe_eq(v, l + 1 < L ? q.evals[l + 1] : pr.fri_last);
            }
            st_fri[s] = ok ? 1 : 0;
        }
    });
    for (int8_t v : st_fri) if (v != 1) return 0;
    // ---- step 4: DEEP consistency and openings (verifier.rs:358-441, :525-557)
    if (pr.openings.size() < queries) return 0;
    const fe z2 = fe_sqr(z);
    std::vector<fe> zk(R);
    for (uint32_t r = 0; r < R; ++r) zk[r] = fe_mul(z, fe_pow_u64(g, air.offsets[r]));
    host_parallel_for(queries, 1, [&](size_t b, size_t e) {
        t_merkle_backend = backend;
        for (size_t s = b; s < e; ++s) {
            const Opening& o = pr.openings[s];
            if (o.trace_evals.size() != C || o.trace_paths.size() != n_roots) { st_deep[s] = -1; continue; }
            bool ok = true;
            fe hh[2] = {o.h1, o.h2};
            ok &= merkle_ok(o.comp_path, pr.comp_root, iotas[s], hh, 2);
            ok &= merkle_ok(o.trace_paths[0], pr.trace_roots[0], iotas[s], o.trace_evals.data(), air.main_cols);
            if (n_roots > 1) ok &= merkle_ok(o.trace_paths[1], pr.trace_roots[1], iotas[s], o.trace_evals.data() + air.main_cols, air.aux_cols);
            fe x = fe_mul(h, fe_pow_u64(w, iotas[s]));
            fe d2 = fe_sub(x, z2);
            if (fe_is_zero(d2)) { st_deep[s] = -1; continue; }
            fe i2 = fe_inv(d2);
            fe acc = fe_zero();
            bool zero_den = false;
            for (uint32_t r = 0; r < R && !zero_den; ++r) {
                fe d = fe_sub(x, zk[r]);
                if (fe_is_zero(d)) { zero_den = true; break; }
                fe ir = fe_inv(d);
                for (uint32_t j = 0; j < C; ++j)
                    acc = fe_add(acc, fe_mul(fe_mul(fe_sub(o.trace_evals[j], pr.ood[(size_t)r * C + j]), ir), tg[(size_t)R * j + r]));
            }
            if (zero_den) { st_deep[s] = -1; continue; }
            acc = fe_add(acc, fe_mul(fe_mul(fe_sub(o.h1, pr.h1z), i2), gamma));
            acc = fe_add(acc, fe_mul(fe_mul(fe_sub(o.h2, pr.h2z), i2), gamma_p));
            ok &= fe_eq(acc, pr.queries[s].evals[0]);
            st_deep[s] = ok ? 1 : 0;
        }
    });
    for (int8_t v : st_deep) if (v != 1) return 0;
    return 1;
}

int cairo_verify_host(const uint8_t* proof_bytes, size_t len, const PublicInputs& pub, uint8_t blowup, uint64_t queries, uint64_t coset_offset, uint8_t grinding) {
    CairoAirInfo info = cairo_air_info(pub);
    VerifySpec spec;
    spec.main_cols = info.main_columns; spec.aux_cols = info.aux_columns;
    spec.offsets = {0, 1};
    spec.degrees = info.transition_degrees; spec.exemptions = info.transition_exemptions;
    spec.bound_factor = 2; spec.n_rap = 3;
    // the trace length is only known from the proof: read it first (boundary steps depend on it)
    if (len < 8) throw std::runtime_error("malformed: InvalidAmountOfBytes");
    uint64_t n = 0;
    for (int i = 0; i < 8; ++i) n = (n << 8) | proof_bytes[i];
    const bool has_rc = info.has_rc_builtin;
    const uint32_t C = info.trace_columns;
    spec.boundary = [&pub, n, has_rc](const std::vector<fe>& rap) { fe r[3] = {rap[0], rap[1], rap[2]}; return boundary_constraints(pub, r, n, has_rc); };
    spec.transition = [C, has_rc](const fe* frame, const std::vector<fe>& rap, fe* out) { fe r[3] = {rap[0], rap[1], rap[2]}; cairo_transition_host(frame, C, has_rc, r, out); };
    return verify_host(proof_bytes, len, spec, blowup, queries, coset_offset, grinding);
}

// `verify::<F, A>` for a program AIR (include/stark252_hip.h sp_air_desc); ops as in AirOpDev of stark_kernels.h.
int air_verify_host(const uint8_t* proof_bytes, size_t len, uint32_t main_cols, uint32_t aux_cols, const std::vector<uint32_t>& offsets,
                    const std::vector<uint32_t>& degrees, const std::vector<uint32_t>& exemptions, uint32_t bound_factor,
                    const std::vector<std::array<uint16_t, 3>>& ops /*op, a, b*/, const std::vector<fe>& consts, uint32_t n_rap,
                    const std::vector<BoundaryConstraint>& boundary, uint8_t blowup, uint64_t queries, uint64_t coset_offset, uint8_t grinding) {
    const uint32_t C = main_cols + aux_cols, T = (uint32_t)degrees.size(), R = (uint32_t)offsets.size();
    if (T == 0 || R == 0 || exemptions.size() != T || bound_factor < 1) return 0;
    for (size_t t = 0; t < ops.size(); ++t) {   // same well-formedness rules as the prover
        const uint16_t op = ops[t][0], a = ops[t][1], b = ops[t][2];
        bool ok = true;
        switch (op) {
            case 0: ok = a < R && b < C; break;
            case 1: ok = a < consts.size() + n_rap; break;
            case 2: case 3: case 4: ok = a < t && b < t && ops[a][0] != 5 && ops[b][0] != 5; break;
            case 5: ok = a < T && b < t && ops[b][0] != 5; break;
            default: ok = false;
        }
        if (!ok) throw std::runtime_error("malformed: constraint program");
    }
    VerifySpec spec;
    spec.main_cols = main_cols; spec.aux_cols = aux_cols; spec.offsets = offsets; spec.degrees = degrees; spec.exemptions = exemptions;
    spec.bound_factor = bound_factor; spec.n_rap = n_rap;
    spec.boundary = [&boundary](const std::vector<fe>&) { return boundary; };
    spec.transition = [&ops, &consts, C, T](const fe* frame, const std::vector<fe>& rap, fe* out) {
        std::vector<fe> v(ops.size(), fe_zero());
        for (uint32_t k = 0; k < T; ++k) out[k] = fe_zero();
        for (size_t t = 0; t < ops.size(); ++t) {
            const uint16_t op = ops[t][0], a = ops[t][1], b = ops[t][2];
            switch (op) {
                case 0: v[t] = frame[(size_t)a * C + b]; break;
                case 1: v[t] = a < consts.size() ? consts[a] : rap[a - consts.size()]; break;
                case 2: v[t] = fe_add(v[a], v[b]); break;
                case 3: v[t] = fe_sub(v[a], v[b]); break;
                case 4: v[t] = fe_mul(v[a], v[b]); break;
                default: out[a] = v[b]; break;
            }
        }
    };
    return verify_host(proof_bytes, len, spec, blowup, queries, coset_offset, grinding);
}

}  // namespace sp

// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/fp.hpp header).
// Restatement of the lambdaworks-crypto (rev a17b951, un-vendored) Merkle tree as configured by the reference
// (src/starks/config.rs:10-20): FRI layers use `Keccak256Tree` (leaf = Keccak256(32-byte BE element)), trace and
// composition commitments use `BatchKeccak256Tree` (leaf = Keccak256(concatenated 32-byte BE row elements)),
// parent = Keccak256(left || right), 32-byte commitments; `get_proof_by_pos` returns the sibling path bottom-up
// (used at reference src/starks/prover.rs:500,515 and src/starks/fri/mod.rs:105-107); `Proof::verify` walks the
// index parity.  Conventions: SURVEY.md §8(c) item 4 (confirmed by the golden proofs).  And DefaultTranscript
// (SURVEY.md §8(c) item 5).
#pragma once
#include "fp.hpp"
#include "keccak.hpp"
#include "poseidon.hpp"
#include <array>
#include <vector>

namespace oracle {

typedef std::array<uint8_t, 32> Digest;

// The hash of the trees: 0 = Keccak256 (the reference, config.rs:10-20), 1 = Starknet Poseidon (poseidon.hpp - the optional backend of
// BASELINE.json configs[4], NO reference counterpart: leaf of a row = hash_many, leaf of a single-element tree = hash_single,
// node = hash(left, right), digest = canonical big-endian element).  A process-wide switch: the oracle is test infrastructure.
inline int& merkle_backend() { static int b = 0; return b; }
inline Digest poseidon_digest(const Fp& x) { Digest d; x.to_bytes_be(d.data()); return d; }

inline Digest hash_felts(const Fp* row, size_t cols, bool single_element_tree = false) {
    if (merkle_backend() == 1) return poseidon_digest(single_element_tree ? Poseidon::get().hash_single(row[0]) : Poseidon::get().hash_many(row, cols));
    Keccak256 k;
    uint8_t b[32];
    for (size_t j = 0; j < cols; ++j) { row[j].to_bytes_be(b); k.update(b, 32); }
    Digest d; k.finalize(d.data());
    return d;
}
inline Digest hash_pair(const Digest& l, const Digest& r) {
    if (merkle_backend() == 1) return poseidon_digest(Poseidon::get().hash(Fp::from_bytes_be(l.data()), Fp::from_bytes_be(r.data())));
    uint8_t buf[64];
    std::memcpy(buf, l.data(), 32); std::memcpy(buf + 32, r.data(), 32);
    Digest d; keccak256(buf, 64, d.data());
    return d;
}

struct MerkleTree {
    // nodes[0] = root; children of i are 2i+1, 2i+2; leaves occupy nodes[n-1 .. 2n-2]
    std::vector<Digest> nodes;
    size_t n_leaves;
    Digest root;

    void build_from_leaves(std::vector<Digest>&& leaves) {
        n_leaves = leaves.size();
        if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) throw std::runtime_error("leaf count must be a power of two");
        nodes.assign(2 * n_leaves - 1, Digest());
        for (size_t i = 0; i < n_leaves; ++i) nodes[n_leaves - 1 + i] = leaves[i];
        for (size_t lo = n_leaves / 2; lo >= 1; lo /= 2) {   // level by level (independent nodes: threads help the Poseidon backend)
#pragma omp parallel for schedule(static) if (lo >= 64)
            for (long i = (long)lo - 1; i < (long)(2 * lo - 1); ++i) nodes[i] = hash_pair(nodes[2 * i + 1], nodes[2 * i + 2]);
        }
        root = nodes[0];
    }
    // rows: row-major n x cols
    static MerkleTree build_batched(const Fp* rows, size_t n, size_t cols, bool single_element_tree = false) {
        std::vector<Digest> leaves(n);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)n; ++i) leaves[i] = hash_felts(rows + (size_t)i * cols, cols, single_element_tree);
        MerkleTree t; t.build_from_leaves(std::move(leaves));
        return t;
    }
    static MerkleTree build_single(const Fp* vals, size_t n) { return build_batched(vals, n, 1, true); }

    std::vector<Digest> proof(size_t pos) const {
        std::vector<Digest> path;
        size_t p = pos + n_leaves - 1;
        while (p != 0) {
            size_t sib = (p & 1) ? p + 1 : p - 1;
            path.push_back(nodes[sib]);
            p = (p - 1) / 2;
        }
        return path;
    }
};

// lambdaworks `Proof::verify::<Backend>(root, index, value)`
inline bool merkle_verify(const std::vector<Digest>& path, const Digest& root, size_t index, const Fp* value, size_t cols,
                          bool single_element_tree = false) {
    Digest h = hash_felts(value, cols, single_element_tree);
    for (const Digest& sib : path) {
        h = (index & 1) ? hash_pair(sib, h) : hash_pair(h, sib);
        index >>= 1;
    }
    return h == root;
}

// lambdaworks `DefaultTranscript`: append -> buffer ||= bytes; challenge -> d = Keccak256(buffer),
// r = reverse(d), buffer := r, return r.
struct Transcript {
    std::vector<uint8_t> buf;
    void append(const uint8_t* d, size_t n) { buf.insert(buf.end(), d, d + n); }
    void append_digest(const Digest& d) { append(d.data(), 32); }
    void append_felt(const Fp& x) { uint8_t b[32]; x.to_bytes_be(b); append(b, 32); }
    Digest challenge() {
        Digest d; keccak256(buf.data(), buf.size(), d.data());
        std::reverse(d.begin(), d.end());
        buf.assign(d.begin(), d.end());
        return d;
    }
    // reference src/starks/transcript.rs:13-43 (251 random bits)
    Fp to_field() {
        Digest r = challenge();
        r[0] &= 0x07;
        return Fp::from_bytes_be(r.data());
    }
    // reference src/starks/transcript.rs:45-51
    uint64_t to_usize() {
        Digest r = challenge();
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v = (v << 8) | r[i];
        return v;
    }
};

// reference src/starks/grinding.rs:17-29
inline uint8_t grinding_trailing_zeros(const Digest& challenge, uint64_t nonce) {
    uint8_t data[40];
    std::memcpy(data, challenge.data(), 32);
    for (int i = 0; i < 8; ++i) data[32 + i] = (uint8_t)(nonce >> (8 * i));  // LE
    uint8_t dig[32];
    keccak256(data, 40, dig);
    uint64_t head = 0;
    for (int i = 0; i < 8; ++i) head = (head << 8) | dig[i];  // BE
    return head == 0 ? 64 : (uint8_t)__builtin_ctzll(head);
}
// reference src/starks/grinding.rs:40-48
inline uint64_t grinding_nonce(const Digest& challenge, uint8_t factor) {
    for (uint64_t n = 0;; ++n)
        if (grinding_trailing_zeros(challenge, n) >= factor) return n;
}

}  // namespace oracle

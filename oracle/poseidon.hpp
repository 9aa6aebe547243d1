// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/fp.hpp header).
// Starknet Poseidon over Stark252 for the optional Poseidon Merkle backend (BASELINE.json configs[4]).  PARITY NOTE: the reference
// has no Poseidon backend (src/starks/config.rs:10-20 fixes Keccak256), so there is no reference artefact to pin this against; it
// is pinned against PUBLIC Starknet known answers instead (tests/test_poseidon.py) and restates the published definition plainly:
// the 91 x 3 round keys sha256("Hades" + index) mod p computed here at start-up (own SHA-256 below), every round adding its three
// keys, x^3 on all lanes in the 4 + 4 full rounds and on the last lane in the 83 partial ones, then the mix
// [[3,1,1],[1,-1,1],[1,1,-2]].  The product (csrc/poseidon.h) uses a generated table of 107 compressed constants and lazily
// reduced arithmetic instead - the two share nothing but the definition.
#pragma once
#include "fp.hpp"
#include <array>
#include <cstring>
#include <string>
#include <vector>

namespace oracle {

// FIPS 180-4 SHA-256 of a short message (only used for the round keys)
inline void sha256(const uint8_t* msg, size_t len, uint8_t out[32]) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
        0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
        0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    std::vector<uint8_t> m(msg, msg + len);
    m.push_back(0x80);
    while (m.size() % 64 != 56) m.push_back(0);
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 7; i >= 0; --i) m.push_back((uint8_t)(bits >> (8 * i)));
    auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    for (size_t off = 0; off < m.size(); off += 64) {
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)m[off + 4 * i] << 24) | ((uint32_t)m[off + 4 * i + 1] << 16) | ((uint32_t)m[off + 4 * i + 2] << 8) | m[off + 4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
            const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}

struct Poseidon {
    static constexpr int FULL_HALF = 4, PARTIAL = 83, ROUNDS = 91;
    std::vector<std::array<Fp, 3>> keys;
    Poseidon() {
        keys.resize(ROUNDS);
        for (int r = 0; r < ROUNDS; ++r)
            for (int j = 0; j < 3; ++j) {
                const std::string name = "Hades" + std::to_string(3 * r + j);
                uint8_t d[32];
                sha256(reinterpret_cast<const uint8_t*>(name.data()), name.size(), d);
                keys[r][j] = Fp::from_bytes_be(d);   // reduces the 256-bit digest modulo p
            }
    }
    static const Poseidon& get() { static const Poseidon p; return p; }

    void permute(Fp s[3]) const {
        const Fp two = Fp::from_u64(2), three = Fp::from_u64(3);
        for (int r = 0; r < ROUNDS; ++r) {
            for (int j = 0; j < 3; ++j) s[j] += keys[r][j];
            if (r < FULL_HALF || r >= FULL_HALF + PARTIAL) { for (int j = 0; j < 3; ++j) s[j] = s[j] * s[j] * s[j]; }
            else s[2] = s[2] * s[2] * s[2];
            const Fp t = s[0] + s[1] + s[2];
            const Fp a = t + two * s[0], b = t - two * s[1], c = t - three * s[2];
            s[0] = a; s[1] = b; s[2] = c;
        }
    }
    Fp hash(const Fp& x, const Fp& y) const { Fp s[3] = {x, y, Fp::from_u64(2)}; permute(s); return s[0]; }
    Fp hash_single(const Fp& x) const { Fp s[3] = {x, Fp::zero(), Fp::one()}; permute(s); return s[0]; }
    Fp hash_many(const Fp* v, size_t n) const {
        std::vector<Fp> m(v, v + n);
        m.push_back(Fp::one());
        if (m.size() % 2) m.push_back(Fp::zero());
        Fp s[3] = {Fp::zero(), Fp::zero(), Fp::zero()};
        for (size_t i = 0; i < m.size(); i += 2) { s[0] += m[i]; s[1] += m[i + 1]; permute(s); }
        return s[0];
    }
};

}  // namespace oracle

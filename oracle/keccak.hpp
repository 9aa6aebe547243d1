// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/fp.hpp header).
// Keccak-256 with the ORIGINAL Keccak padding (0x01 .. 0x80), i.e. `sha3::Keccak256` (sha3 0.10.6,
// reference Cargo.toml:17) as used by reference src/starks/grinding.rs:1,25 and by the lambdaworks-crypto
// Merkle backends / DefaultTranscript (rev a17b951, not vendored).  KAT: keccak256("") =
// c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace oracle {

static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

static inline uint64_t rotl64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

inline void keccak_f1600(uint64_t s[25]) {
    static const int rho[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) s[i] ^= d[i % 5];
        // rho + pi: B[y, 2x+3y] = rot(A[x,y], r[x,y]); index = x + 5*y
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) {
                int nx = y, ny = (2 * x + 3 * y) % 5;
                b[nx + 5 * ny] = rotl64(s[x + 5 * y], rho[x + 5 * y]);
            }
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) s[x + 5 * y] = b[x + 5 * y] ^ ((~b[(x + 1) % 5 + 5 * y]) & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= KECCAK_RC[round];
    }
}

struct Keccak256 {
    uint64_t st[25];
    uint8_t buf[136];
    size_t pos;
    Keccak256() : pos(0) { std::memset(st, 0, sizeof(st)); }
    void absorb_block(const uint8_t* blk) {
        for (int i = 0; i < 17; ++i) {
            uint64_t v; std::memcpy(&v, blk + 8 * i, 8);  // little-endian host
            st[i] ^= v;
        }
        keccak_f1600(st);
    }
    void update(const uint8_t* data, size_t len) {
        while (len) {
            if (pos == 0 && len >= 136) { absorb_block(data); data += 136; len -= 136; continue; }
            size_t take = 136 - pos; if (take > len) take = len;
            std::memcpy(buf + pos, data, take); pos += take; data += take; len -= take;
            if (pos == 136) { absorb_block(buf); pos = 0; }
        }
    }
    void finalize(uint8_t out[32]) {
        std::memset(buf + pos, 0, 136 - pos);
        buf[pos] ^= 0x01; buf[135] ^= 0x80;
        absorb_block(buf);
        std::memcpy(out, st, 32);
    }
};

inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
    Keccak256 k; k.update(data, len); k.finalize(out);
}

}  // namespace oracle

// Keccak-f[1600] / Keccak-256 (original 0x01..0x80 padding, i.e. sha3::Keccak256 as used by reference
// src/starks/grinding.rs:25 and the lambdaworks-crypto Merkle backends selected at src/starks/config.rs:10-20),
// usable from gfx950 kernels (state held in 50 VGPRs; every index is compile-time) and from host code
// (Fiat-Shamir transcript, which is sequential and stays on the CPU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPK_HD __host__ __device__ __forceinline__

__device__ __constant__ const uint64_t SP_KECCAK_RC_DEV[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const uint64_t SP_KECCAK_RC_HOST[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

SPK_HD uint64_t sp_rotl64(uint64_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    // two v_alignbit_b32 on the 32-bit halves (a 64-bit shift pair is three slower VALU ops on gfx950)
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    uint32_t rlo, rhi;
    if (n == 32) { rlo = hi; rhi = lo; }
    else if (n < 32) { rhi = __builtin_amdgcn_alignbit(hi, lo, 32 - n); rlo = __builtin_amdgcn_alignbit(lo, hi, 32 - n); }
    else { rhi = __builtin_amdgcn_alignbit(lo, hi, 64 - n); rlo = __builtin_amdgcn_alignbit(hi, lo, 64 - n); }
    return ((uint64_t)rhi << 32) | rlo;
#else
    return (x << n) | (x >> (64 - n));
#endif
}
SPK_HD uint64_t sp_bswap64(uint64_t x) {
    x = ((x & 0x00ff00ff00ff00ffULL) << 8) | ((x >> 8) & 0x00ff00ff00ff00ffULL);
    x = ((x & 0x0000ffff0000ffffULL) << 16) | ((x >> 16) & 0x0000ffff0000ffffULL);
    return (x << 32) | (x >> 32);
}

// One round, fully scalarised (no arrays indexed at run time -> stays in registers on the GPU).
#define SP_KECCAK_ROUND(rc)                                                                       \
    {                                                                                             \
        uint64_t c0 = s[0] ^ s[5] ^ s[10] ^ s[15] ^ s[20];                                        \
        uint64_t c1 = s[1] ^ s[6] ^ s[11] ^ s[16] ^ s[21];                                        \
        uint64_t c2 = s[2] ^ s[7] ^ s[12] ^ s[17] ^ s[22];                                        \
        uint64_t c3 = s[3] ^ s[8] ^ s[13] ^ s[18] ^ s[23];                                        \
        uint64_t c4 = s[4] ^ s[9] ^ s[14] ^ s[19] ^ s[24];                                        \
        uint64_t d0 = c4 ^ sp_rotl64(c1, 1), d1 = c0 ^ sp_rotl64(c2, 1), d2 = c1 ^ sp_rotl64(c3, 1), \
                 d3 = c2 ^ sp_rotl64(c4, 1), d4 = c3 ^ sp_rotl64(c0, 1);                          \
        uint64_t b0 = s[0] ^ d0;                                                                  \
        uint64_t b1 = sp_rotl64(s[6] ^ d1, 44);                                                   \
        uint64_t b2 = sp_rotl64(s[12] ^ d2, 43);                                                  \
        uint64_t b3 = sp_rotl64(s[18] ^ d3, 21);                                                  \
        uint64_t b4 = sp_rotl64(s[24] ^ d4, 14);                                                  \
        uint64_t b5 = sp_rotl64(s[3] ^ d3, 28);                                                   \
        uint64_t b6 = sp_rotl64(s[9] ^ d4, 20);                                                   \
        uint64_t b7 = sp_rotl64(s[10] ^ d0, 3);                                                   \
        uint64_t b8 = sp_rotl64(s[16] ^ d1, 45);                                                  \
        uint64_t b9 = sp_rotl64(s[22] ^ d2, 61);                                                  \
        uint64_t b10 = sp_rotl64(s[1] ^ d1, 1);                                                   \
        uint64_t b11 = sp_rotl64(s[7] ^ d2, 6);                                                   \
        uint64_t b12 = sp_rotl64(s[13] ^ d3, 25);                                                 \
        uint64_t b13 = sp_rotl64(s[19] ^ d4, 8);                                                  \
        uint64_t b14 = sp_rotl64(s[20] ^ d0, 18);                                                 \
        uint64_t b15 = sp_rotl64(s[4] ^ d4, 27);                                                  \
        uint64_t b16 = sp_rotl64(s[5] ^ d0, 36);                                                  \
        uint64_t b17 = sp_rotl64(s[11] ^ d1, 10);                                                 \
        uint64_t b18 = sp_rotl64(s[17] ^ d2, 15);                                                 \
        uint64_t b19 = sp_rotl64(s[23] ^ d3, 56);                                                 \
        uint64_t b20 = sp_rotl64(s[2] ^ d2, 62);                                                  \
        uint64_t b21 = sp_rotl64(s[8] ^ d3, 55);                                                  \
        uint64_t b22 = sp_rotl64(s[14] ^ d4, 39);                                                 \
        uint64_t b23 = sp_rotl64(s[15] ^ d0, 41);                                                 \
        uint64_t b24 = sp_rotl64(s[21] ^ d1, 2);                                                  \
        s[0] = b0 ^ (~b1 & b2) ^ (rc);                                                            \
        s[1] = b1 ^ (~b2 & b3);                                                                   \
        s[2] = b2 ^ (~b3 & b4);                                                                   \
        s[3] = b3 ^ (~b4 & b0);                                                                   \
        s[4] = b4 ^ (~b0 & b1);                                                                   \
        s[5] = b5 ^ (~b6 & b7);                                                                   \
        s[6] = b6 ^ (~b7 & b8);                                                                   \
        s[7] = b7 ^ (~b8 & b9);                                                                   \
        s[8] = b8 ^ (~b9 & b5);                                                                   \
        s[9] = b9 ^ (~b5 & b6);                                                                   \
        s[10] = b10 ^ (~b11 & b12);                                                               \
        s[11] = b11 ^ (~b12 & b13);                                                               \
        s[12] = b12 ^ (~b13 & b14);                                                               \
        s[13] = b13 ^ (~b14 & b10);                                                               \
        s[14] = b14 ^ (~b10 & b11);                                                               \
        s[15] = b15 ^ (~b16 & b17);                                                               \
        s[16] = b16 ^ (~b17 & b18);                                                               \
        s[17] = b17 ^ (~b18 & b19);                                                               \
        s[18] = b18 ^ (~b19 & b15);                                                               \
        s[19] = b19 ^ (~b15 & b16);                                                               \
        s[20] = b20 ^ (~b21 & b22);                                                               \
        s[21] = b21 ^ (~b22 & b23);                                                               \
        s[22] = b22 ^ (~b23 & b24);                                                               \
        s[23] = b23 ^ (~b24 & b20);                                                               \
        s[24] = b24 ^ (~b20 & b21);                                                               \
    }

__device__ __forceinline__ void sp_keccak_f1600_dev(uint64_t s[25]) {
#pragma unroll 1
    for (int r = 0; r < 24; ++r) SP_KECCAK_ROUND(SP_KECCAK_RC_DEV[r])
}
inline void sp_keccak_f1600_host(uint64_t s[25]) {
    for (int r = 0; r < 24; ++r) SP_KECCAK_ROUND(SP_KECCAK_RC_HOST[r])
}

// Host-side one-shot Keccak-256.
inline void sp_keccak256_host(const uint8_t* data, size_t len, uint8_t out[32]) {
    uint64_t s[25];
    for (int i = 0; i < 25; ++i) s[i] = 0;
    while (len >= 136) {
        for (int i = 0; i < 17; ++i) { uint64_t v; __builtin_memcpy(&v, data + 8 * i, 8); s[i] ^= v; }
        sp_keccak_f1600_host(s);
        data += 136; len -= 136;
    }
    uint8_t blk[136];
    for (int i = 0; i < 136; ++i) blk[i] = 0;
    for (size_t i = 0; i < len; ++i) blk[i] = data[i];
    blk[len] ^= 0x01; blk[135] ^= 0x80;
    for (int i = 0; i < 17; ++i) { uint64_t v; __builtin_memcpy(&v, blk + 8 * i, 8); s[i] ^= v; }
    sp_keccak_f1600_host(s);
    __builtin_memcpy(out, s, 32);
}

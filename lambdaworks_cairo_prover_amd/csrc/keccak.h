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

// 3-input logic: gfx950 has v_bitop3_b32 (any boolean function of three inputs in one VALU op), which turns the five-way
// theta parities into two ops, folds the theta application into one op per lane half and chi into one op
// (measured: 10.1 vs 6.5 G permutations/s, tools/experiments/ubench_keccak.hip, profiles/r01_keccak_ubench.txt).
SPK_HD uint64_t sp_xor3(uint64_t a, uint64_t b, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0x96);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0x96);
    return ((uint64_t)hi << 32) | lo;
#else
    return a ^ b ^ c;
#endif
}
SPK_HD uint64_t sp_chi(uint64_t a, uint64_t b, uint64_t c) {  // a ^ (~b & c)
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0xd2);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0xd2);
    return ((uint64_t)hi << 32) | lo;
#else
    return a ^ (~b & c);
#endif
}

// One round, fully scalarised (no arrays indexed at run time -> stays in registers on the GPU).
// theta: c_x = parity of column x, lane (x, y) ^= c_(x-1) ^ rotl(c_(x+1), 1); rho/pi: b = rotl(lane, r); chi; iota.
#define SP_KECCAK_ROUND(rc)                                                                                   \
    {                                                                                                         \
        const uint64_t c0 = sp_xor3(sp_xor3(s[0], s[5], s[10]), s[15], s[20]);                                \
        const uint64_t c1 = sp_xor3(sp_xor3(s[1], s[6], s[11]), s[16], s[21]);                                \
        const uint64_t c2 = sp_xor3(sp_xor3(s[2], s[7], s[12]), s[17], s[22]);                                \
        const uint64_t c3 = sp_xor3(sp_xor3(s[3], s[8], s[13]), s[18], s[23]);                                \
        const uint64_t c4 = sp_xor3(sp_xor3(s[4], s[9], s[14]), s[19], s[24]);                                \
        const uint64_t r0 = sp_rotl64(c0, 1), r1 = sp_rotl64(c1, 1), r2 = sp_rotl64(c2, 1),                   \
                       r3 = sp_rotl64(c3, 1), r4 = sp_rotl64(c4, 1);                                          \
        const uint64_t b0 = sp_xor3(s[0], c4, r1);                                                            \
        const uint64_t b1 = sp_rotl64(sp_xor3(s[6], c0, r2), 44);                                             \
        const uint64_t b2 = sp_rotl64(sp_xor3(s[12], c1, r3), 43);                                            \
        const uint64_t b3 = sp_rotl64(sp_xor3(s[18], c2, r4), 21);                                            \
        const uint64_t b4 = sp_rotl64(sp_xor3(s[24], c3, r0), 14);                                            \
        const uint64_t b5 = sp_rotl64(sp_xor3(s[3], c2, r4), 28);                                             \
        const uint64_t b6 = sp_rotl64(sp_xor3(s[9], c3, r0), 20);                                             \
        const uint64_t b7 = sp_rotl64(sp_xor3(s[10], c4, r1), 3);                                             \
        const uint64_t b8 = sp_rotl64(sp_xor3(s[16], c0, r2), 45);                                            \
        const uint64_t b9 = sp_rotl64(sp_xor3(s[22], c1, r3), 61);                                            \
        const uint64_t b10 = sp_rotl64(sp_xor3(s[1], c0, r2), 1);                                             \
        const uint64_t b11 = sp_rotl64(sp_xor3(s[7], c1, r3), 6);                                             \
        const uint64_t b12 = sp_rotl64(sp_xor3(s[13], c2, r4), 25);                                           \
        const uint64_t b13 = sp_rotl64(sp_xor3(s[19], c3, r0), 8);                                            \
        const uint64_t b14 = sp_rotl64(sp_xor3(s[20], c4, r1), 18);                                           \
        const uint64_t b15 = sp_rotl64(sp_xor3(s[4], c3, r0), 27);                                            \
        const uint64_t b16 = sp_rotl64(sp_xor3(s[5], c4, r1), 36);                                            \
        const uint64_t b17 = sp_rotl64(sp_xor3(s[11], c0, r2), 10);                                           \
        const uint64_t b18 = sp_rotl64(sp_xor3(s[17], c1, r3), 15);                                           \
        const uint64_t b19 = sp_rotl64(sp_xor3(s[23], c2, r4), 56);                                           \
        const uint64_t b20 = sp_rotl64(sp_xor3(s[2], c1, r3), 62);                                            \
        const uint64_t b21 = sp_rotl64(sp_xor3(s[8], c2, r4), 55);                                            \
        const uint64_t b22 = sp_rotl64(sp_xor3(s[14], c3, r0), 39);                                           \
        const uint64_t b23 = sp_rotl64(sp_xor3(s[15], c4, r1), 41);                                           \
        const uint64_t b24 = sp_rotl64(sp_xor3(s[21], c0, r2), 2);                                            \
        s[0] = sp_chi(b0, b1, b2) ^ (rc);                                                                     \
        s[1] = sp_chi(b1, b2, b3); s[2] = sp_chi(b2, b3, b4); s[3] = sp_chi(b3, b4, b0);                      \
        s[4] = sp_chi(b4, b0, b1);                                                                            \
        s[5] = sp_chi(b5, b6, b7); s[6] = sp_chi(b6, b7, b8); s[7] = sp_chi(b7, b8, b9);                      \
        s[8] = sp_chi(b8, b9, b5); s[9] = sp_chi(b9, b5, b6);                                                 \
        s[10] = sp_chi(b10, b11, b12); s[11] = sp_chi(b11, b12, b13); s[12] = sp_chi(b12, b13, b14);          \
        s[13] = sp_chi(b13, b14, b10); s[14] = sp_chi(b14, b10, b11);                                         \
        s[15] = sp_chi(b15, b16, b17); s[16] = sp_chi(b16, b17, b18); s[17] = sp_chi(b17, b18, b19);          \
        s[18] = sp_chi(b18, b19, b15); s[19] = sp_chi(b19, b15, b16);                                         \
        s[20] = sp_chi(b20, b21, b22); s[21] = sp_chi(b21, b22, b23); s[22] = sp_chi(b22, b23, b24);          \
        s[23] = sp_chi(b23, b24, b20); s[24] = sp_chi(b24, b20, b21);                                         \
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

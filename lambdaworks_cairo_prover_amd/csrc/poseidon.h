// Starknet Poseidon over Stark252 for the optional Poseidon Merkle backend (BASELINE.json configs[4]).
//
// NO COUNTERPART IN THE REFERENCE: src/starks/config.rs:10-20 fixes Keccak256 trees, and the pinned lambdaworks-crypto revision has
// no Poseidon backend.  What is built is what later lambdaworks-crypto versions configure for this field (PoseidonCairoStark252 with
// TreePoseidon / BatchPoseidonTree), stated here from the published definition of the hash:
//   Hades permutation, state of three elements, x^3, 4 full + 83 partial + 4 full rounds, mix M = [[3,1,1],[1,-1,1],[1,1,-2]],
//   round keys sha256("Hades" + index) mod p (tools/gen_poseidon_constants.py: compressed to 107 additions);
//   hash(x, y)      = permute(x, y, 2)[0]                                   - a tree node over its two children,
//   hash_single(x)  = permute(x, 0, 1)[0]                                   - the leaf of a single-element tree (FRI layers),
//   hash_many(v)    = sponge of rate 2 over v || 1 || 0*, first state word  - the leaf of a row of trace / composition columns.
// Pinned by the public Starknet vectors (tests/test_poseidon.py) through three independent implementations (pure Python, the
// oracle's, this one); a digest is the canonical 32-byte big-endian encoding of the resulting element.
//
// Arithmetic: the state stays lazily reduced (every lane in [0, 2p)) between rounds.  A partial round is
//   lane 2 += constant (< 3p < 2^253);  square (36 + 8 multiply-adds) and product (72) -> [0, 2p);  the mix in raw 256-bit adds
//   (3 s0 + s1 + s2 < 10p, s0 - s1 + s2 + 4p < 10p, s0 + s1 - 2 s2 + 8p < 14p: all below 32p < 2^256) and one quotient-estimate
//   reduction per lane (fe_reduce_lazy_2p) - about 650 issue slots of which the two products are 500.
#pragma once
#include "fp.h"
#include "poseidon_constants.h"

namespace sp {

constexpr int POSEIDON_FULL_HALF = 4, POSEIDON_PARTIAL = 83;

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ const fe SP_POSEIDON_RC_DEV[SP_POSEIDON_N_RC] = {SP_POSEIDON_RC_TABLE};
#define SP_POSEIDON_RC SP_POSEIDON_RC_DEV
#else
static const fe SP_POSEIDON_RC_HOST[SP_POSEIDON_N_RC] = {SP_POSEIDON_RC_TABLE};
#define SP_POSEIDON_RC SP_POSEIDON_RC_HOST
#endif

// x^3 for x < 3p (< 2^253): result in [0, 2p)
SP_HD fe poseidon_cube(const fe& x) { return fe_mul_lazy(fe_sqr_lazy(x), x); }

// (s0, s1, s2) <- M (s0, s1, s2), inputs in [0, 2p), outputs in [0, 2p)
SP_HD void poseidon_mix(fe& s0, fe& s1, fe& s2) {
    const fe t = fe_add_raw(fe_add_raw(s0, s1), s2);                        // < 6p
    const fe a0 = fe_add_raw(t, fe_add_raw(s0, s0));                        // t + 2 s0 < 10p
    const fe a1 = fe_sub_add_kp(t, fe_add_raw(s1, s1), 4u);                 // t - 2 s1 + 4p in (0, 10p)
    const fe a2 = fe_sub_add_kp(t, fe_add_raw(fe_add_raw(s2, s2), s2), 8u);  // t - 3 s2 + 8p in (0, 14p)
    s0 = fe_reduce_lazy_2p(a0);
    s1 = fe_reduce_lazy_2p(a1);
    s2 = fe_reduce_lazy_2p(a2);
}

// The Hades permutation on Montgomery-form lanes in [0, 2p); the lanes come back in [0, 2p).
SP_HD void poseidon_permute(fe& s0, fe& s1, fe& s2) {
    const fe* rc = SP_POSEIDON_RC;
#pragma unroll 1
    for (int r = 0; r < POSEIDON_FULL_HALF; ++r, rc += 3) {
        s0 = poseidon_cube(fe_add_raw(s0, rc[0]));
        s1 = poseidon_cube(fe_add_raw(s1, rc[1]));
        s2 = poseidon_cube(fe_add_raw(s2, rc[2]));
        poseidon_mix(s0, s1, s2);
    }
#pragma unroll 1
    for (int r = 0; r < POSEIDON_PARTIAL; ++r, ++rc) {
        s2 = poseidon_cube(fe_add_raw(s2, rc[0]));
        poseidon_mix(s0, s1, s2);
    }
#pragma unroll 1
    for (int r = 0; r < POSEIDON_FULL_HALF; ++r, rc += 3) {
        s0 = poseidon_cube(fe_add_raw(s0, rc[0]));
        s1 = poseidon_cube(fe_add_raw(s1, rc[1]));
        s2 = poseidon_cube(fe_add_raw(s2, rc[2]));
        poseidon_mix(s0, s1, s2);
    }
}

SP_HD fe poseidon_two() {   // 2 in Montgomery form
    const fe one = fe_one();
    return fe_add(one, one);
}

// hash(x, y): Montgomery-form inputs below 2p, canonical Montgomery-form output
SP_HD fe poseidon_hash2(const fe& x, const fe& y) {
    fe s0 = x, s1 = y, s2 = poseidon_two();
    poseidon_permute(s0, s1, s2);
    return fe_reduce_once(s0);
}
SP_HD fe poseidon_hash1(const fe& x) {
    fe s0 = x, s1 = fe_zero(), s2 = fe_one();
    poseidon_permute(s0, s1, s2);
    return fe_reduce_once(s0);
}
// hash_many over a strided sequence: element j at p[j * stride]
SP_HD fe poseidon_hash_many(const fe* p, uint64_t stride, uint32_t n) {
    fe s0 = fe_zero(), s1 = fe_zero(), s2 = fe_zero();
    uint32_t j = 0;
#pragma unroll 1
    for (; j + 2 <= n; j += 2) {
        s0 = fe_reduce_lazy_2p(fe_add_raw(s0, p[(uint64_t)j * stride]));
        s1 = fe_reduce_lazy_2p(fe_add_raw(s1, p[(uint64_t)(j + 1) * stride]));
        poseidon_permute(s0, s1, s2);
    }
    if (j < n) {   // odd length: the last element and the padding 1 share a block
        s0 = fe_reduce_lazy_2p(fe_add_raw(s0, p[(uint64_t)j * stride]));
        s1 = fe_reduce_lazy_2p(fe_add_raw(s1, fe_one()));
    } else {       // even length: a block of its own for 1, 0
        s0 = fe_reduce_lazy_2p(fe_add_raw(s0, fe_one()));
    }
    poseidon_permute(s0, s1, s2);
    return fe_reduce_once(s0);
}

// a digest = the canonical big-endian bytes of an element, held as four little-endian 64-bit words like a Keccak digest
SP_HD uint64_t poseidon_bswap64(uint64_t x) {
    x = ((x & 0x00ff00ff00ff00ffULL) << 8) | ((x >> 8) & 0x00ff00ff00ff00ffULL);
    x = ((x & 0x0000ffff0000ffffULL) << 16) | ((x >> 16) & 0x0000ffff0000ffffULL);
    return (x << 32) | (x >> 32);
}
SP_HD void poseidon_digest_from_fe(const fe& a, uint64_t w[4]) {
    const fe raw = fe_from_mont(a);
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = poseidon_bswap64((uint64_t)raw.v[2 * (3 - k)] | ((uint64_t)raw.v[2 * (3 - k) + 1] << 32));
}
// (a digest is below p by construction; bytes that are not - a forged authentication path - are reduced by the Montgomery
// product like any 256-bit operand, so the verifier simply computes a different root)
SP_HD fe poseidon_fe_from_digest(const uint64_t w[4]) {
    fe raw;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t limb = poseidon_bswap64(w[k]);
        raw.v[2 * (3 - k)] = (uint32_t)limb; raw.v[2 * (3 - k) + 1] = (uint32_t)(limb >> 32);
    }
    return fe_reduce_once(fe_mul_lazy(raw, fe_r2()));
}

}  // namespace sp

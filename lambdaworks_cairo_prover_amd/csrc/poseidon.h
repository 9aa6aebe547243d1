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
//   lane 2 += constant (< 3p < 2^253);  square (36 + 8 multiply-adds) and product (72) -> [0, 2p);  the linear layer in raw 256-bit
//   adds and shifts (everything stays below 32p < 2^256) with two quotient-estimate reductions (fe_reduce_lazy_2p) - see
//   poseidon_partial_rounds: 116 multiply-adds and ~420 other vector instructions per round.
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

// a * 2^K as a 256-bit integer (caller guarantees no overflow)
template <int K>
SP_HD fe poseidon_shl(const fe& a) {
    fe r;
    r.v[0] = a.v[0] << K;
#pragma unroll
    for (int i = 1; i < 8; ++i) r.v[i] = (a.v[i] << K) | (a.v[i - 1] >> (32 - K));
    return r;
}

// The partial rounds touch lanes 0 and 1 only linearly, and the next round needs of them only their sum: with v = s0 and
// u = s0 + s1 a round is
//     c = (s2 + k)^3;   s2 <- u - 2c;   (u, v) <- (4v + 2c, 2v + u + c)
// (from s0' = t + 2 s0, s1' = t - 2 s1, s2' = t - 3c with t = u + c).  u is recomputed from v and c every round, so it never
// needs a reduction (u < 12p); v and s2 take one each - 98 instructions of linear work per round instead of 151.
SP_HD void poseidon_partial_rounds(fe& s0, fe& s1, fe& s2, const fe* rc) {
    fe v = s0, u = fe_add_raw(s0, s1);                                         // u < 4p
#pragma unroll 1
    for (int r = 0; r < POSEIDON_PARTIAL; ++r) {
        const fe c = poseidon_cube(fe_add_raw(s2, rc[r]));                    // [0, 2p)
        const fe c2 = fe_add_raw(c, c);                                        // < 4p
        s2 = fe_reduce_lazy_2p(fe_sub_add_kp(u, c2, 4u));                      // u - 2c + 4p < 16p
        const fe vn = fe_add_raw(fe_add_raw(poseidon_shl<1>(v), u), c);        // 2v + u + c < 18p
        u = fe_add_raw(poseidon_shl<2>(v), c2);                                // 4v + 2c < 12p
        v = fe_reduce_lazy_2p(vn);
    }
    s0 = v;
    s1 = fe_reduce_lazy_2p(fe_sub_add_kp(u, v, 2u));                           // u - v + 2p < 14p
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
    poseidon_partial_rounds(s0, s1, s2, rc);
    rc += POSEIDON_PARTIAL;
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

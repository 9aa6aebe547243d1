// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped MI355X path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// CPU restatement of the Stark252 prime field used by the reference prover
// (`FieldElement<Stark252PrimeField>`, alias `FE` at reference src/lib.rs:12-13).  The reference's field
// lives in lambdaworks-math @ rev a17b951 (reference Cargo.toml:11), which is NOT vendored under
// /root/reference; this file restates its published algorithm (4x64-bit Montgomery, R = 2^256) and is
// pinned by the golden proofs under tests/golden/ (see oracle/README.md for the parity status).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>

namespace oracle {

typedef unsigned __int128 u128;

// p = 2^251 + 17*2^192 + 1, little-endian 64-bit limbs.
static const uint64_t P[4] = {1ULL, 0ULL, 0ULL, 0x0800000000000011ULL};

struct Fp {
    uint64_t l[4];  // Montgomery form a*R mod p, little-endian limbs, always canonical (< p)

    static inline bool geq_p(const uint64_t* a) {
        for (int i = 3; i >= 0; --i) {
            if (a[i] > P[i]) return true;
            if (a[i] < P[i]) return false;
        }
        return true;
    }
    static inline void sub_p(uint64_t* a) {
        u128 br = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)a[i] - P[i] - br;
            a[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }

    static inline Fp zero() { Fp r; r.l[0] = r.l[1] = r.l[2] = r.l[3] = 0; return r; }

    // Montgomery product: CIOS with -p^{-1} mod 2^64 = 2^64-1 (p = 1 mod 2^64).
    static inline Fp mont_mul(const Fp& a, const Fp& b) {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) {
                u128 s = (u128)a.l[j] * b.l[i] + t[j] + (uint64_t)c;
                t[j] = (uint64_t)s;
                c = s >> 64;
            }
            u128 s = (u128)t[4] + (uint64_t)c;
            t[4] = (uint64_t)s;
            t[5] = (uint64_t)(s >> 64);
            uint64_t m = (uint64_t)(0 - t[0]);  // t[0] * (-p^{-1}) mod 2^64
            // t += m * p, then shift one limb.  p limbs: {1, 0, 0, P3}
            u128 s0 = (u128)t[0] + m;  // low limb becomes 0
            uint64_t cy = (uint64_t)(s0 >> 64);
            u128 s1 = (u128)t[1] + cy;
            t[0] = (uint64_t)s1; cy = (uint64_t)(s1 >> 64);
            u128 s2 = (u128)t[2] + cy;
            t[1] = (uint64_t)s2; cy = (uint64_t)(s2 >> 64);
            u128 s3 = (u128)m * P[3] + t[3] + cy;
            t[2] = (uint64_t)s3; cy = (uint64_t)(s3 >> 64);
            u128 s4 = (u128)t[4] + cy;
            t[3] = (uint64_t)s4;
            t[4] = t[5] + (uint64_t)(s4 >> 64);
            t[5] = 0;
        }
        Fp r;
        r.l[0] = t[0]; r.l[1] = t[1]; r.l[2] = t[2]; r.l[3] = t[3];
        if (t[4] || geq_p(r.l)) sub_p(r.l);
        return r;
    }

    static Fp R2() {  // R^2 mod p
        Fp r;
        r.l[3] = 0x07ffd4ab5e008810ULL; r.l[2] = 0xffffffffff6f8000ULL;
        r.l[1] = 0x00000001330fffffULL; r.l[0] = 0xfffffd737e000401ULL;
        return r;
    }
    static Fp one() {  // R mod p
        Fp r;
        r.l[3] = 0x07fffffffffffdf0ULL; r.l[2] = 0xffffffffffffffffULL;
        r.l[1] = 0xffffffffffffffffULL; r.l[0] = 0xffffffffffffffe1ULL;
        return r;
    }
    static Fp from_raw(const uint64_t v[4]) {  // integer (little-endian limbs, < p) -> Montgomery
        Fp a; std::memcpy(a.l, v, 32);
        return mont_mul(a, R2());
    }
    static Fp from_u64(uint64_t v) { uint64_t x[4] = {v, 0, 0, 0}; return from_raw(x); }
    void to_raw(uint64_t v[4]) const {  // Montgomery -> canonical integer
        Fp o; o.l[0] = 1; o.l[1] = o.l[2] = o.l[3] = 0;
        Fp r = mont_mul(*this, o);
        std::memcpy(v, r.l, 32);
    }
    // 32-byte big-endian canonical representative (lambdaworks `to_bytes_be`)
    void to_bytes_be(uint8_t out[32]) const {
        uint64_t v[4]; to_raw(v);
        for (int i = 0; i < 4; ++i)
            for (int b = 0; b < 8; ++b) out[(3 - i) * 8 + b] = (uint8_t)(v[i] >> (56 - 8 * b));
    }
    // `from_bytes_be`: value is reduced mod p if >= p (only values < p occur on this path).
    static Fp from_bytes_be(const uint8_t in[32]) {
        uint64_t v[4];
        for (int i = 0; i < 4; ++i) {
            uint64_t x = 0;
            for (int b = 0; b < 8; ++b) x = (x << 8) | in[(3 - i) * 8 + b];
            v[i] = x;
        }
        while (geq_p(v)) sub_p(v);
        return from_raw(v);
    }
    static Fp from_hex(const std::string& hs) {
        std::string h = hs;
        if (h.size() >= 2 && h[0] == '0' && (h[1] == 'x' || h[1] == 'X')) h = h.substr(2);
        if (h.size() > 64) throw std::runtime_error("hex too long");
        h = std::string(64 - h.size(), '0') + h;
        uint8_t b[32];
        for (int i = 0; i < 32; ++i) b[i] = (uint8_t)std::stoul(h.substr(2 * i, 2), nullptr, 16);
        return from_bytes_be(b);
    }
    std::string to_hex() const {
        uint8_t b[32]; to_bytes_be(b);
        static const char* d = "0123456789abcdef";
        std::string s;
        for (int i = 0; i < 32; ++i) { s.push_back(d[b[i] >> 4]); s.push_back(d[b[i] & 15]); }
        return s;
    }

    inline Fp operator+(const Fp& o) const {
        Fp r; u128 c = 0;
        for (int i = 0; i < 4; ++i) { u128 s = (u128)l[i] + o.l[i] + (uint64_t)c; r.l[i] = (uint64_t)s; c = s >> 64; }
        if (c || geq_p(r.l)) sub_p(r.l);
        return r;
    }
    inline Fp operator-(const Fp& o) const {
        Fp r; u128 br = 0;
        for (int i = 0; i < 4; ++i) { u128 d = (u128)l[i] - o.l[i] - br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
        if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { u128 s = (u128)r.l[i] + P[i] + (uint64_t)c; r.l[i] = (uint64_t)s; c = s >> 64; } }
        return r;
    }
    inline Fp operator-() const { return zero() - *this; }
    inline Fp operator*(const Fp& o) const { return mont_mul(*this, o); }
    inline Fp& operator+=(const Fp& o) { *this = *this + o; return *this; }
    inline Fp& operator-=(const Fp& o) { *this = *this - o; return *this; }
    inline Fp& operator*=(const Fp& o) { *this = *this * o; return *this; }
    inline bool operator==(const Fp& o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
    inline bool operator!=(const Fp& o) const { return !(*this == o); }
    inline bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    inline Fp square() const { return mont_mul(*this, *this); }

    Fp pow(uint64_t e) const {
        Fp r = one(), b = *this;
        while (e) { if (e & 1) r = r * b; b = b.square(); e >>= 1; }
        return r;
    }
    Fp pow_limbs(const uint64_t e[4]) const {
        Fp r = one();
        for (int i = 255; i >= 0; --i) {
            r = r.square();
            if ((e[i / 64] >> (i % 64)) & 1) r = r * *this;
        }
        return r;
    }
    Fp inv() const {  // Fermat: a^(p-2); panics on zero like lambdaworks
        if (is_zero()) throw std::runtime_error("inverse of zero");
        uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};  // p-2 (P[0]=1 -> borrow)
        // P[0] - 2 underflows: p - 2 = {0xffff...ffff, 0xffff...ffff, 0xffff...ffff, P3 - 1}
        e[0] = 0xffffffffffffffffULL; e[1] = 0xffffffffffffffffULL; e[2] = 0xffffffffffffffffULL; e[3] = P[3] - 1;
        return pow_limbs(e);
    }
    // canonical integer value as 4 limbs (the lambdaworks `representative()`)
    void representative(uint64_t v[4]) const { to_raw(v); }
    uint64_t low_u64() const { uint64_t v[4]; to_raw(v); return v[0]; }
};

// lambdaworks `FieldElement::inplace_batch_inverse` (Montgomery's trick); panics on a zero element.
inline void batch_inverse(Fp* a, size_t n) {
    if (n == 0) return;
    std::vector<Fp> pre(n);
    Fp acc = Fp::one();
    for (size_t i = 0; i < n; ++i) { pre[i] = acc; acc = acc * a[i]; }
    Fp inv = acc.inv();
    for (size_t i = n; i-- > 0;) { Fp t = inv * pre[i]; inv = inv * a[i]; a[i] = t; }
}
inline void batch_inverse(std::vector<Fp>& a) { batch_inverse(a.data(), a.size()); }

// 2-adic generator of order 2^192 (lambdaworks Stark252PrimeField::TWO_ADIC_PRIMITVE_ROOT_OF_UNITY),
// confirmed by the byte-identical golden proofs.
inline Fp two_adic_root() { return Fp::from_hex("5282db87529cfa3f0464519c8b0fa5ad187148e11a61616070024f42f8ef94"); }
// `get_primitive_root_of_unity(order)`: W^(2^(192-order))
inline Fp primitive_root(unsigned order) {
    Fp w = two_adic_root();
    for (unsigned i = order; i < 192; ++i) w = w.square();
    return w;
}

}  // namespace oracle

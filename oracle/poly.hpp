// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/fp.hpp header).
// Restatement of the lambdaworks-math (rev a17b951, un-vendored) FFT/Polynomial surface that the reference
// calls: `Polynomial::interpolate_fft` (reference src/starks/trace.rs:107), `interpolate_offset_fft`
// (src/starks/constraints/evaluation_table.rs:32), `evaluate_offset_fft` (src/starks/prover.rs:117,
// src/starks/fri/fri_commitment.rs:36), `evaluate` (prover.rs:301-304, frame.rs:67-83),
// `ruffini_division_inplace` (prover.rs:436-473), `even_odd_decomposition` (prover.rs:252),
// `get_powers_of_primitive_root_coset` (src/starks/domain.rs:30-44).  Conventions: SURVEY.md §8(c) items 2-3,
// all confirmed by the byte-identical golden proofs.
#pragma once
#include "fp.hpp"
#include <vector>
#include <algorithm>

namespace oracle {

typedef std::vector<Fp> Poly;  // coefficients, lowest degree first; trailing zeros trimmed

inline void trim(Poly& p) { while (!p.empty() && p.back().is_zero()) p.pop_back(); }

inline unsigned log2_exact(size_t n) {
    unsigned k = 0; while ((size_t(1) << k) < n) ++k;
    if ((size_t(1) << k) != n) throw std::runtime_error("size not a power of two");
    return k;
}
inline size_t next_pow2(size_t n) { size_t m = 1; while (m < n) m <<= 1; return m; }

// powers w^0 .. w^(count-1) times offset: lambdaworks get_powers_of_primitive_root_coset(order, count, offset)
inline std::vector<Fp> root_coset(unsigned order, size_t count, const Fp& offset) {
    Fp w = primitive_root(order);
    std::vector<Fp> r(count);
    Fp cur = offset;
    for (size_t i = 0; i < count; ++i) { r[i] = cur; cur = cur * w; }
    return r;
}

// In-place radix-2 DFT, natural order in and out: a[i] <- sum_k a[k] w^(ik), w a primitive len-th root.
inline void dft_inplace(std::vector<Fp>& a, const Fp& w) {
    size_t n = a.size();
    if (n <= 1) return;
    unsigned lg = log2_exact(n);
    // bit reversal
    for (size_t i = 0; i < n; ++i) {
        size_t j = 0;
        for (unsigned b = 0; b < lg; ++b) if (i >> b & 1) j |= size_t(1) << (lg - 1 - b);
        if (i < j) std::swap(a[i], a[j]);
    }
    std::vector<Fp> tw(n / 2);
    Fp cur = Fp::one();
    for (size_t i = 0; i < n / 2; ++i) { tw[i] = cur; cur = cur * w; }
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len / 2, step = n / len;
        for (size_t s = 0; s < n; s += len)
            for (size_t j = 0; j < half; ++j) {
                Fp u = a[s + j], v = a[s + j + half] * tw[j * step];
                a[s + j] = u + v;
                a[s + j + half] = u - v;
            }
    }
}

// Polynomial::evaluate_fft-style forward transform of `coeffs` zero-padded to `size`.
inline std::vector<Fp> evaluate_fft_size(const Poly& coeffs, size_t size) {
    std::vector<Fp> a(size, Fp::zero());
    std::copy(coeffs.begin(), coeffs.end(), a.begin());
    dft_inplace(a, primitive_root(log2_exact(size)));
    return a;
}

// `poly.evaluate_offset_fft(blowup, Some(domain_size), offset)`: coefficient k times offset^k,
// zero-pad to max(len, domain_size).next_pow2() * blowup, forward DFT, natural order.
inline std::vector<Fp> evaluate_offset_fft(const Poly& p, size_t blowup, size_t domain_size, const Fp& offset) {
    Poly scaled(p.size());
    Fp cur = Fp::one();
    for (size_t k = 0; k < p.size(); ++k) { scaled[k] = p[k] * cur; cur = cur * offset; }
    size_t len = next_pow2(std::max(p.size(), domain_size)) * blowup;
    return evaluate_fft_size(scaled, len);
}

// `Polynomial::interpolate_fft(evals)`: inverse DFT over <g>, natural order; result trimmed.
inline Poly interpolate_fft(const std::vector<Fp>& evals) {
    std::vector<Fp> a = evals;
    size_t n = a.size();
    Fp w = primitive_root(log2_exact(n)).inv();
    dft_inplace(a, w);
    Fp ninv = Fp::from_u64(n).inv();
    for (auto& x : a) x = x * ninv;
    trim(a);
    return a;
}

// `Polynomial::interpolate_offset_fft(evals, offset)`: inverse DFT then coefficient k times offset^-k.
inline Poly interpolate_offset_fft(const std::vector<Fp>& evals, const Fp& offset) {
    std::vector<Fp> a = evals;
    size_t n = a.size();
    Fp w = primitive_root(log2_exact(n)).inv();
    dft_inplace(a, w);
    Fp ninv = Fp::from_u64(n).inv();
    Fp oinv = offset.inv();
    Fp cur = ninv;
    for (auto& x : a) { x = x * cur; cur = cur * oinv; }
    trim(a);
    return a;
}

// Horner (`Polynomial::evaluate`)
inline Fp poly_eval(const Poly& p, const Fp& x) {
    Fp acc = Fp::zero();
    for (size_t i = p.size(); i-- > 0;) acc = acc * x + p[i];
    return acc;
}

// p <- (p - p(b)) / (X - b)  (`ruffini_division_inplace`; the remainder is dropped)
inline void ruffini_division_inplace(Poly& p, const Fp& b) {
    if (p.empty()) return;
    size_t d = p.size() - 1;
    Fp c = p[d];  // q_{d-1}
    for (size_t i = d; i-- > 0;) {
        Fp t = p[i] + b * c;  // q_{i-1} (the remainder when i == 0)
        p[i] = c;             // q_i
        c = t;
    }
    p.pop_back();
    trim(p);
}

inline void even_odd_decomposition(const Poly& p, Poly& even, Poly& odd) {
    even.clear(); odd.clear();
    for (size_t i = 0; i < p.size(); ++i) (i & 1 ? odd : even).push_back(p[i]);
    trim(even); trim(odd);
}

inline Poly poly_add(const Poly& a, const Poly& b) {
    Poly r(std::max(a.size(), b.size()), Fp::zero());
    for (size_t i = 0; i < a.size(); ++i) r[i] = a[i];
    for (size_t i = 0; i < b.size(); ++i) r[i] = r[i] + b[i];
    trim(r);
    return r;
}
inline Poly poly_scale(const Poly& a, const Fp& s) {
    Poly r(a.size());
    for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] * s;
    trim(r);
    return r;
}
inline Poly poly_sub_const(const Poly& a, const Fp& c) {
    Poly r = a;
    if (r.empty()) r.push_back(Fp::zero());
    r[0] = r[0] - c;
    trim(r);
    return r;
}

}  // namespace oracle

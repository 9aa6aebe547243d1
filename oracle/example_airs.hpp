// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/fp.hpp header).
// The example AIRs of the reference (src/starks/example/*.rs), restated one by one, plus a "program" AIR whose
// transition constraints are a small straight-line program over frame cells (the form the device library accepts for
// AIRs other than Cairo, include/stark252_hip.h sp_air_desc).  tests/test_oracle_example_airs.py checks that the program
// form of every example gives the same proof bytes as its hand-written class, which pins the program definitions.
#pragma once
#include "air.hpp"
#include <stdexcept>

namespace oracle {

// ---- simple_fibonacci.rs:37-128 ----------------------------------------------------------------------------------
struct FibonacciAir : Air {
    Fp a0, a1;
    FibonacciAir(size_t n, const Fp& a0_, const Fp& a1_, const ProofOptions& opt) : a0(a0_), a1(a1_) {
        trace_len = n;
        ctx = AirContext{opt, 1, {1}, {0, 1, 2}, {2}, 1, 1};
    }
    std::vector<Fp> build_auxiliary_trace(const std::vector<Fp>&, size_t, const std::vector<Fp>&) const override { return {}; }
    std::vector<Fp> build_rap_challenges(Transcript&) const override { return {}; }
    size_t number_auxiliary_rap_columns() const override { return 0; }
    size_t composition_poly_degree_bound() const override { return trace_len; }
    void compute_transition(const Fp* f, const std::vector<Fp>&, Fp* out) const override { out[0] = f[2] - f[1] - f[0]; }
    std::vector<BoundaryConstraint> boundary_constraints(const std::vector<Fp>&) const override { return {{0, 0, a0}, {0, 1, a1}}; }
};
inline std::vector<Fp> fibonacci_trace(const Fp& a0, const Fp& a1, size_t n) {  // simple_fibonacci.rs:111-128
    std::vector<Fp> t{a0, a1};
    for (size_t i = 2; i < n; ++i) t.push_back(t[i - 1] + t[i - 2]);
    return t;
}

// ---- fibonacci_2_columns.rs:25-130 ------------------------------------------------------------------------------
struct Fibonacci2ColsAir : Air {
    Fp a0, a1;
    Fibonacci2ColsAir(size_t n, const Fp& a0_, const Fp& a1_, const ProofOptions& opt) : a0(a0_), a1(a1_) {
        trace_len = n;
        ctx = AirContext{opt, 2, {1, 1}, {0, 1}, {1, 1}, 2, 1};
    }
    std::vector<Fp> build_auxiliary_trace(const std::vector<Fp>&, size_t, const std::vector<Fp>&) const override { return {}; }
    std::vector<Fp> build_rap_challenges(Transcript&) const override { return {}; }
    size_t number_auxiliary_rap_columns() const override { return 0; }
    size_t composition_poly_degree_bound() const override { return trace_len; }
    void compute_transition(const Fp* f, const std::vector<Fp>&, Fp* out) const override {
        const Fp* r0 = f; const Fp* r1 = f + 2;
        out[0] = r1[0] - r0[0] - r0[1];
        out[1] = r1[1] - r0[1] - r1[0];
    }
    std::vector<BoundaryConstraint> boundary_constraints(const std::vector<Fp>&) const override { return {{0, 0, a0}, {1, 0, a1}}; }
};
inline std::vector<Fp> fibonacci_trace_2_columns(const Fp& a0, const Fp& a1, size_t n) {  // row-major n x 2
    std::vector<Fp> c0{a0}, c1{a1};
    for (size_t i = 1; i < n; ++i) { Fp nv = c0[i - 1] + c1[i - 1]; c0.push_back(nv); c1.push_back(nv + c1[i - 1]); }
    std::vector<Fp> rows(2 * n);
    for (size_t i = 0; i < n; ++i) { rows[2 * i] = c0[i]; rows[2 * i + 1] = c1[i]; }
    return rows;
}

// ---- quadratic_air.rs:30-125 ------------------------------------------------------------------------------------
struct QuadraticAir : Air {
    Fp a0;
    QuadraticAir(size_t n, const Fp& a0_, const ProofOptions& opt) : a0(a0_) {
        trace_len = n;
        ctx = AirContext{opt, 1, {2}, {0, 1}, {1}, 1, 1};
    }
    std::vector<Fp> build_auxiliary_trace(const std::vector<Fp>&, size_t, const std::vector<Fp>&) const override { return {}; }
    std::vector<Fp> build_rap_challenges(Transcript&) const override { return {}; }
    size_t number_auxiliary_rap_columns() const override { return 0; }
    size_t composition_poly_degree_bound() const override { return 2 * trace_len; }
    void compute_transition(const Fp* f, const std::vector<Fp>&, Fp* out) const override { out[0] = f[1] - f[0] * f[0]; }
    std::vector<BoundaryConstraint> boundary_constraints(const std::vector<Fp>&) const override { return {{0, 0, a0}}; }
};
inline std::vector<Fp> quadratic_trace(const Fp& a0, size_t n) {
    std::vector<Fp> t{a0};
    for (size_t i = 1; i < n; ++i) t.push_back(t[i - 1] * t[i - 1]);
    return t;
}

// ---- dummy_air.rs:20-118 ----------------------------------------------------------------------------------------
struct DummyAir : Air {
    DummyAir(size_t n, const ProofOptions& opt) {
        trace_len = n;
        ctx = AirContext{opt, 2, {2, 1}, {0, 1, 2}, {0, 2}, 2, 1};
    }
    std::vector<Fp> build_auxiliary_trace(const std::vector<Fp>&, size_t, const std::vector<Fp>&) const override { return {}; }
    std::vector<Fp> build_rap_challenges(Transcript&) const override { return {}; }
    size_t number_auxiliary_rap_columns() const override { return 0; }
    size_t composition_poly_degree_bound() const override { return trace_len; }
    void compute_transition(const Fp* f, const std::vector<Fp>&, Fp* out) const override {
        const Fp* r0 = f; const Fp* r1 = f + 2; const Fp* r2 = f + 4;
        out[0] = r0[0] * (r0[0] - Fp::one());
        out[1] = r2[1] - r1[1] - r0[1];
    }
    std::vector<BoundaryConstraint> boundary_constraints(const std::vector<Fp>&) const override { return {{1, 0, Fp::one()}, {1, 1, Fp::one()}}; }
};
inline std::vector<Fp> dummy_trace(size_t n) {  // row-major n x 2: column 0 all ones, column 1 fibonacci
    std::vector<Fp> fib{Fp::one(), Fp::one()};
    for (size_t i = 2; i < n; ++i) fib.push_back(fib[i - 1] + fib[i - 2]);
    std::vector<Fp> rows(2 * n);
    for (size_t i = 0; i < n; ++i) { rows[2 * i] = Fp::one(); rows[2 * i + 1] = fib[i]; }
    return rows;
}

// ---- fibonacci_rap.rs:23-194 ------------------------------------------------------------------------------------
inline std::vector<Fp> fibonacci_rap_aux_column(const std::vector<Fp>& main, size_t mc, const Fp& gamma) {
    size_t n = main.size() / mc;
    std::vector<Fp> aux(n);
    for (size_t i = 0; i < n; ++i) {
        if (i == 0) aux[i] = Fp::one();
        else aux[i] = aux[i - 1] * ((main[(i - 1) * mc + 0] + gamma) * (main[(i - 1) * mc + 1] + gamma).inv());
    }
    return aux;
}
struct FibonacciRapAir : Air {
    size_t steps;
    FibonacciRapAir(size_t n, size_t steps_, const ProofOptions& opt) : steps(steps_) {
        trace_len = n;
        size_t exemptions = 3 + n - steps - 1;
        ctx = AirContext{opt, 3, {1, 2}, {0, 1, 2}, {exemptions, 1}, 2, 2};
    }
    std::vector<Fp> build_auxiliary_trace(const std::vector<Fp>& main, size_t mc, const std::vector<Fp>& rap) const override {
        return fibonacci_rap_aux_column(main, mc, rap[0]);
    }
    std::vector<Fp> build_rap_challenges(Transcript& t) const override { return {t.to_field()}; }
    size_t number_auxiliary_rap_columns() const override { return 1; }
    size_t composition_poly_degree_bound() const override { return trace_len; }
    void compute_transition(const Fp* f, const std::vector<Fp>& rap, Fp* out) const override {
        const Fp* r0 = f; const Fp* r1 = f + 3; const Fp* r2 = f + 6;
        const Fp& gamma = rap[0];
        out[0] = r2[0] - r1[0] - r0[0];
        out[1] = r1[2] * (r0[1] + gamma) - r0[2] * (r0[0] + gamma);
    }
    std::vector<BoundaryConstraint> boundary_constraints(const std::vector<Fp>&) const override {
        return {{0, 0, Fp::one()}, {0, 1, Fp::one()}, {2, 0, Fp::one()}};
    }
};
inline std::vector<Fp> fibonacci_rap_trace(const Fp& a0, const Fp& a1, size_t steps, size_t* n_out) {  // row-major n x 2
    std::vector<Fp> fib{a0, a1};
    for (size_t i = 2; i < steps; ++i) fib.push_back(fib[i - 1] + fib[i - 2]);
    std::vector<Fp> perm = fib;
    perm[0] = fib[steps - 1];
    perm[steps - 1] = a0;
    fib.push_back(Fp::zero()); perm.push_back(Fp::zero());
    size_t n = 1;
    while (n < fib.size()) n <<= 1;     // resize_to_next_power_of_two
    fib.resize(n, Fp::zero()); perm.resize(n, Fp::zero());
    std::vector<Fp> rows(2 * n);
    for (size_t i = 0; i < n; ++i) { rows[2 * i] = fib[i]; rows[2 * i + 1] = perm[i]; }
    *n_out = n;
    return rows;
}

// ---- program AIR ------------------------------------------------------------------------------------------------
// op: 0 LOAD (a = index into transition_offsets, b = column), 1 CONST (a = constant index; indices >= n_consts are the
// RAP challenges), 2 ADD, 3 SUB, 4 MUL (a, b = indices of earlier ops), 5 OUT (a = constraint index, b = op index).
struct AirOp { uint8_t op; uint8_t pad; uint16_t a, b; uint16_t pad2; };
struct ProgramAir : Air {
    std::vector<AirOp> ops;
    std::vector<Fp> consts;
    size_t n_rap = 0, aux_cols = 0, aux_kind = 0, bound_factor = 1;
    std::vector<BoundaryConstraint> bcs;
    int (*aux_fn)(void* user, const uint8_t* rap, uint32_t n_rap, uint8_t* aux_rows_out) = nullptr;   // aux_kind 2: the caller's build_auxiliary_trace
    void* aux_user = nullptr;
    std::vector<Fp> build_auxiliary_trace(const std::vector<Fp>& main, size_t mc, const std::vector<Fp>& rap) const override {
        if (aux_kind == 0) return {};
        if (aux_kind == 1) return fibonacci_rap_aux_column(main, mc, rap[0]);
        if (aux_kind == 2) {   // canonical big-endian in and out (the oracle's only encoding)
            if (!aux_fn) throw std::runtime_error("aux_kind 2 without a callback");
            const size_t n = main.size() / mc;
            std::vector<uint8_t> rb(std::max<size_t>(1, rap.size()) * 32), rows(n * aux_cols * 32);
            for (size_t i = 0; i < rap.size(); ++i) rap[i].to_bytes_be(&rb[32 * i]);
            if (aux_fn(aux_user, rb.data(), (uint32_t)rap.size(), rows.data()) != 0) throw std::runtime_error("auxiliary-trace callback failed");
            std::vector<Fp> out(n * aux_cols);
            for (size_t i = 0; i < out.size(); ++i) out[i] = Fp::from_bytes_be(&rows[32 * i]);
            return out;
        }
        throw std::runtime_error("unknown aux kind");
    }
    std::vector<Fp> build_rap_challenges(Transcript& t) const override {
        std::vector<Fp> r;
        for (size_t i = 0; i < n_rap; ++i) r.push_back(t.to_field());
        return r;
    }
    size_t number_auxiliary_rap_columns() const override { return aux_cols; }
    size_t composition_poly_degree_bound() const override { return bound_factor * trace_len; }
    void compute_transition(const Fp* f, const std::vector<Fp>& rap, Fp* out) const override {
        std::vector<Fp> v(ops.size());
        const size_t cols = ctx.trace_columns;
        for (size_t i = 0; i < ops.size(); ++i) {
            const AirOp& o = ops[i];
            switch (o.op) {
                case 0: v[i] = f[(size_t)o.a * cols + o.b]; break;
                case 1: v[i] = o.a < consts.size() ? consts[o.a] : rap.at(o.a - consts.size()); break;
                case 2: v[i] = v[o.a] + v[o.b]; break;
                case 3: v[i] = v[o.a] - v[o.b]; break;
                case 4: v[i] = v[o.a] * v[o.b]; break;
                case 5: out[o.a] = v[o.b]; break;
                default: throw std::runtime_error("bad op");
            }
        }
    }
    std::vector<BoundaryConstraint> boundary_constraints(const std::vector<Fp>&) const override { return bcs; }
};

}  // namespace oracle

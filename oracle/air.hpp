// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/fp.hpp header).
// AIR interface restated from reference src/starks/traits.rs:15-119, src/starks/context.rs:4-18,
// src/starks/proof/options.rs:21-26,144-151 and src/starks/constraints/boundary.rs:13-33.
#pragma once
#include "fp.hpp"
#include "merkle.hpp"
#include <vector>
#include <cstddef>

namespace oracle {

struct ProofOptions {
    uint8_t blowup_factor;
    size_t fri_number_of_queries;
    uint64_t coset_offset;
    uint8_t grinding_factor;
    static ProofOptions default_test_options() { return ProofOptions{4, 3, 3, 1}; }  // options.rs:144-151
};

struct AirContext {
    ProofOptions proof_options;
    size_t trace_columns;
    std::vector<size_t> transition_degrees;
    std::vector<size_t> transition_offsets;
    std::vector<size_t> transition_exemptions;
    size_t num_transition_constraints;
    size_t num_transition_exemptions;
};

struct BoundaryConstraint { size_t col, step; Fp value; };

struct Air {
    AirContext ctx;
    size_t trace_len;
    virtual ~Air() {}
    // row-major n x aux_cols table (empty if the AIR has no auxiliary trace)
    virtual std::vector<Fp> build_auxiliary_trace(const std::vector<Fp>& main_trace, size_t main_cols,
                                                  const std::vector<Fp>& rap) const = 0;
    virtual std::vector<Fp> build_rap_challenges(Transcript& t) const = 0;
    virtual size_t number_auxiliary_rap_columns() const = 0;
    virtual size_t composition_poly_degree_bound() const = 0;
    // frame: offsets.size() rows of trace_columns elements, row-major
    virtual void compute_transition(const Fp* frame, const std::vector<Fp>& rap, Fp* out) const = 0;
    virtual std::vector<BoundaryConstraint> boundary_constraints(const std::vector<Fp>& rap) const = 0;
};

}  // namespace oracle

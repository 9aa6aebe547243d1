// Host-side pieces of the Cairo AIR that sit between the device rounds (reference src/cairo/air.rs):
// CairoAIR::new (:587-658), boundary_constraints (:777-849).  (build_auxiliary_trace runs on the device: aux_kernels.hip.)
#pragma once
#include "cairo_host.h"
#include "../../include/stark252_hip.h"
#include <vector>

namespace sp {

struct BoundaryConstraint { uint32_t col; uint64_t step; fe value; };

struct CairoAirInfo {
    uint32_t trace_columns, main_columns, aux_columns;
    uint32_t num_transition_constraints;
    bool has_rc_builtin;
    std::vector<uint32_t> transition_degrees, transition_exemptions;
};

CairoAirInfo cairo_air_info(const PublicInputs& pub);

std::vector<BoundaryConstraint> boundary_constraints(const PublicInputs& pub, const fe rap[3], uint64_t trace_length, bool has_rc_builtin);

void host_batch_inverse(std::vector<fe>& a);  // throws std::runtime_error on a zero element

// sp_cairo_public_inputs (C ABI view) -> PublicInputs; throws std::runtime_error on addresses beyond 64 bits (capi_host.cpp)
PublicInputs public_inputs_from_c(const sp_cairo_public_inputs* p);

}  // namespace sp

// See cairo_air_host.h.
#include "cairo_air_host.h"
#include <algorithm>
#include <numeric>
#include <stdexcept>
#include <unordered_map>

namespace sp {

CairoAirInfo cairo_air_info(const PublicInputs& pub) {
    CairoAirInfo a;
    a.has_rc_builtin = !pub.memory_segments.empty();  // air.rs:623 ("hacky solution" kept as is)
    a.main_columns = a.has_rc_builtin ? 43 : 34;
    a.aux_columns = 18;
    a.trace_columns = a.main_columns + a.aux_columns;
    a.num_transition_constraints = a.has_rc_builtin ? 50 : 49;
    static const uint32_t deg[49] = {2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3,
                                     2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2};
    static const uint32_t ex[49] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0,
                                    0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0};
    a.transition_degrees.assign(deg, deg + 49);
    a.transition_exemptions.assign(ex, ex + 49);
    if (a.has_rc_builtin) { a.transition_degrees.push_back(1); a.transition_exemptions.push_back(0); }
    return a;
}

void host_batch_inverse(std::vector<fe>& a) {
    if (a.empty()) return;
    std::vector<fe> pre(a.size());
    fe acc = fe_one();
    for (size_t i = 0; i < a.size(); ++i) { pre[i] = acc; acc = fe_mul(acc, a[i]); }
    if (fe_is_zero(acc)) throw std::runtime_error("batch inverse of a zero element");
    fe inv = fe_inv(acc);
    for (size_t i = a.size(); i-- > 0;) { fe t = fe_mul(inv, pre[i]); inv = fe_mul(inv, a[i]); a[i] = t; }
}

std::vector<BoundaryConstraint> boundary_constraints(const PublicInputs& pub, const fe rap[3], uint64_t n, bool has_rc) {
    const fe &alpha = rap[0], &z = rap[1];
    const uint32_t bo = has_rc ? 0 : 9;  // BUILTIN_OFFSET (air.rs:152-154)
    fe prod = fe_one();
    for (auto& kv : pub.public_memory) prod = fe_mul(prod, fe_sub(z, fe_add(fe_from_u64(kv.first), fe_mul(alpha, kv.second))));
    if (fe_is_zero(prod)) throw std::runtime_error("permutation boundary value: inverse of zero");
    fe permutation_final = fe_mul(fe_pow_u64(z, pub.public_memory.size()), fe_inv(prod));
    std::vector<BoundaryConstraint> b;
    b.push_back({19, 0, pub.pc_init});
    b.push_back({17, 0, pub.ap_init});
    b.push_back({19, pub.num_steps - 1, pub.pc_final});
    b.push_back({17, pub.num_steps - 1, pub.ap_final});
    b.push_back({57 - bo, n - 1, permutation_final});
    b.push_back({60 - bo, n - 1, fe_one()});
    b.push_back({43 - bo, 0, fe_from_u64(pub.range_check_min)});
    b.push_back({45 - bo, n - 1, fe_from_u64(pub.range_check_max)});
    return b;
}

}  // namespace sp

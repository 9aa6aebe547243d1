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

static inline bool raw_less(const fe& x, const fe& y) {  // compares canonical integers
    for (int k = 7; k >= 0; --k) if (x.v[k] != y.v[k]) return x.v[k] < y.v[k];
    return false;
}

std::vector<fe> build_auxiliary_trace(const fe* main, uint64_t n, uint32_t mc, const PublicInputs& pub, const fe rap[3]) {
    const fe &alpha = rap[0], &z = rap[1], &zrc = rap[2];
    const size_t M = 4 * n;
    std::vector<fe> a_orig(M), v_orig(M);
    for (uint64_t i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) { a_orig[4 * i + k] = main[i * mc + 19 + k]; v_orig[4 * i + k] = main[i * mc + 23 + k]; }
    // add_pub_memory_in_public_input_section (air.rs:475-517)
    std::vector<fe> a_aux = a_orig, v_aux = v_orig;
    const size_t pm = pub.public_memory.size();
    if (pm > M) throw std::runtime_error("public memory larger than the trace");
    const size_t section = M - pm;
    std::vector<uint64_t> pm_addrs;
    if (const MemorySegment* out = pub.segment(1)) {
        uint64_t output_section = out->end - out->start, program_section = pm - output_section;
        for (uint64_t i = 1; i <= program_section; ++i) pm_addrs.push_back(i);
        for (uint64_t a = out->start; a < out->end; ++a) pm_addrs.push_back(a);
    } else {
        for (uint64_t i = 1; i <= pm; ++i) pm_addrs.push_back(i);
    }
    std::unordered_map<uint64_t, fe> mm;
    for (auto& kv : pub.public_memory) mm[kv.first] = kv.second;
    for (size_t i = 0; i < pm; ++i) {
        a_aux[section + i] = fe_from_u64(pm_addrs[i]);
        auto it = mm.find(pm_addrs[i]);
        if (it == mm.end()) throw std::runtime_error("public memory address missing");
        v_aux[section + i] = it->second;
    }
    // stable sort by the address representative (air.rs:519-523)
    std::vector<fe> reps(M);
    for (size_t i = 0; i < M; ++i) reps[i] = fe_from_mont(a_aux[i]);
    std::vector<uint32_t> idx(M);
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return raw_less(reps[x], reps[y]); });
    std::vector<fe> a_s(M), v_s(M);
    for (size_t i = 0; i < M; ++i) { a_s[i] = a_aux[idx[i]]; v_s[i] = v_aux[idx[i]]; }
    // memory permutation column (air.rs:525-551)
    std::vector<fe> den(M);
    for (size_t i = 0; i < M; ++i) den[i] = fe_sub(z, fe_add(a_s[i], fe_mul(alpha, v_s[i])));
    host_batch_inverse(den);
    std::vector<fe> perm(M);
    fe prod = fe_one();
    for (size_t i = 0; i < M; ++i) {
        prod = fe_mul(prod, fe_mul(fe_sub(z, fe_add(a_orig[i], fe_mul(alpha, v_orig[i]))), den[i]));
        perm[i] = prod;
    }
    // range check (air.rs:685-703, :552-572)
    const size_t M3 = 3 * n;
    std::vector<fe> off_orig(M3);
    std::vector<uint16_t> off_sorted(M3);
    for (uint64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            off_orig[3 * i + k] = main[i * mc + 27 + k];
            off_sorted[3 * i + k] = (uint16_t)fe_from_mont(off_orig[3 * i + k]).v[0];
        }
    std::sort(off_sorted.begin(), off_sorted.end());
    std::vector<fe> off_s(M3), rden(M3), rperm(M3);
    fe last_val = fe_zero(); uint32_t last_key = 0x10000;
    for (size_t i = 0; i < M3; ++i) {
        if (off_sorted[i] != last_key) { last_key = off_sorted[i]; last_val = fe_from_u64(last_key); }
        off_s[i] = last_val;
        rden[i] = fe_sub(zrc, last_val);
    }
    host_batch_inverse(rden);
    prod = fe_one();
    for (size_t i = 0; i < M3; ++i) { prod = fe_mul(fe_mul(prod, fe_sub(zrc, off_orig[i])), rden[i]); rperm[i] = prod; }
    // wide format (air.rs:705-728)
    std::vector<fe> aux(n * 18);
    for (uint64_t i = 0; i < n; ++i) {
        fe* r = &aux[i * 18];
        for (int k = 0; k < 3; ++k) r[k] = off_s[3 * i + k];
        for (int k = 0; k < 4; ++k) r[3 + k] = a_s[4 * i + k];
        for (int k = 0; k < 4; ++k) r[7 + k] = v_s[4 * i + k];
        for (int k = 0; k < 4; ++k) r[11 + k] = perm[4 * i + k];
        for (int k = 0; k < 3; ++k) r[15 + k] = rperm[3 * i + k];
    }
    return aux;
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

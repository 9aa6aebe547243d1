// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/fp.hpp header).
// Restatement of the Cairo AIR of the reference: src/cairo/air.rs (column ids :29-154, PublicInputs :163-276,
// add_pub_memory_in_public_input_section :475-494, get_pub_memory_addrs :500-517, sort :519-523, permutation
// columns :525-572, CairoAIR::new :587-658, build_auxiliary_trace :660-729, build_rap_challenges :731-737,
// compute_transition :743-767, boundary_constraints :777-849, constraint helpers :869-1160).
#pragma once
#include "air.hpp"
#include <algorithm>
#include <map>
#include <unordered_map>

namespace oracle {

// frame / main-trace column ids (air.rs:73-154)
enum {
    F_DST_FP = 0, F_OP_0_FP = 1, F_OP_1_VAL = 2, F_OP_1_FP = 3, F_OP_1_AP = 4, F_RES_ADD = 5, F_RES_MUL = 6,
    F_PC_ABS = 7, F_PC_REL = 8, F_PC_JNZ = 9, F_AP_ADD = 10, F_AP_ONE = 11, F_OPC_CALL = 12, F_OPC_RET = 13,
    F_OPC_AEQ = 14,
    FRAME_RES = 16, FRAME_AP = 17, FRAME_FP = 18, FRAME_PC = 19, FRAME_DST_ADDR = 20, FRAME_OP0_ADDR = 21,
    FRAME_OP1_ADDR = 22, FRAME_INST = 23, FRAME_DST = 24, FRAME_OP0 = 25, FRAME_OP1 = 26, OFF_DST = 27,
    OFF_OP0 = 28, OFF_OP1 = 29, FRAME_T0 = 30, FRAME_T1 = 31, FRAME_MUL = 32, FRAME_SELECTOR = 33,
    RC_0 = 34, RC_VALUE = 42,
    RANGE_CHECK_COL_1 = 43, RANGE_CHECK_COL_2 = 44, RANGE_CHECK_COL_3 = 45,
    MEMORY_ADDR_SORTED_0 = 46, MEMORY_VALUES_SORTED_0 = 50, PERMUTATION_ARGUMENT_COL_0 = 54,
    PERMUTATION_ARGUMENT_RANGE_CHECK_COL_1 = 58,
    MEM_P_TRACE_OFFSET = 17, MEM_A_TRACE_OFFSET = 19, BUILTIN_OFFSET = 9
};
// constraint ids (air.rs:29-71)
enum {
    C_INST = 16, C_DST_ADDR = 17, C_OP0_ADDR = 18, C_OP1_ADDR = 19, C_NEXT_AP = 20, C_NEXT_FP = 21, C_NEXT_PC_1 = 22,
    C_NEXT_PC_2 = 23, C_T0 = 24, C_T1 = 25, C_MUL_1 = 26, C_MUL_2 = 27, C_CALL_1 = 28, C_CALL_2 = 29, C_ASSERT_EQ = 30,
    C_MEMORY_INCREASING_0 = 31, C_MEMORY_CONSISTENCY_0 = 35, C_PERMUTATION_ARGUMENT_0 = 39,
    C_RANGE_CHECK_INCREASING_0 = 43, C_RANGE_CHECK_0 = 46, C_RANGE_CHECK_BUILTIN = 49
};

struct MemorySegment { uint8_t type; uint64_t start, end; };  // type 0 = RangeCheck, 1 = Output (air.rs:156-160)

struct PublicInputs {
    Fp pc_init, ap_init, fp_init, pc_final, ap_final;
    bool has_rc_min = false, has_rc_max = false;
    uint16_t rc_min = 0, rc_max = 0;
    std::vector<MemorySegment> memory_segments;
    std::vector<std::pair<Fp, Fp>> public_memory;  // (address, value); the reference keeps a HashMap
    uint64_t num_steps = 0;

    const MemorySegment* segment(uint8_t type) const {
        for (auto& s : memory_segments) if (s.type == type) return &s;
        return nullptr;
    }
    std::unordered_map<uint64_t, Fp> memory_map() const {
        std::unordered_map<uint64_t, Fp> m;
        for (auto& kv : public_memory) m[kv.first.low_u64()] = kv.second;
        return m;
    }
};

struct CairoAir : public Air {
    PublicInputs pub;
    bool has_rc_builtin;

    CairoAir(size_t trace_length, const PublicInputs& pi, const ProofOptions& opt) : pub(pi) {
        trace_len = trace_length;
        size_t trace_columns = 34 + 3 + 12 + 3;
        std::vector<size_t> deg;
        for (int i = 0; i < 15; ++i) deg.push_back(2);
        deg.push_back(1);
        for (int i = 0; i < 15; ++i) deg.push_back(3);
        for (int i = 0; i < 18; ++i) deg.push_back(2);
        static const size_t ex[49] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0,
                                      0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0};
        std::vector<size_t> exv(ex, ex + 49);
        size_t ncons = 49;
        has_rc_builtin = !pi.memory_segments.empty();
        if (has_rc_builtin) { trace_columns += 9; deg.push_back(1); exv.push_back(0); ncons += 1; }
        ctx = AirContext{opt, trace_columns, deg, {0, 1}, exv, ncons, 1};
    }
    size_t builtin_offset() const { return has_rc_builtin ? 0 : BUILTIN_OFFSET; }
    size_t number_auxiliary_rap_columns() const override { return 12 + 3 + 3; }
    size_t composition_poly_degree_bound() const override { return 2 * trace_len; }

    std::vector<Fp> build_rap_challenges(Transcript& t) const override {
        std::vector<Fp> r(3);
        r[0] = t.to_field();  // alpha_memory
        r[1] = t.to_field();  // z_memory
        r[2] = t.to_field();  // z_range_check
        return r;
    }

    std::vector<Fp> build_auxiliary_trace(const std::vector<Fp>& main, size_t mc, const std::vector<Fp>& rap) const override {
        size_t n = main.size() / mc;
        const Fp &alpha = rap[0], &z = rap[1], &zrc = rap[2];
        std::vector<Fp> a_orig(4 * n), v_orig(4 * n);
        static const int acols[4] = {FRAME_PC, FRAME_DST_ADDR, FRAME_OP0_ADDR, FRAME_OP1_ADDR};
        static const int vcols[4] = {FRAME_INST, FRAME_DST, FRAME_OP0, FRAME_OP1};
        for (size_t i = 0; i < n; ++i)
            for (int k = 0; k < 4; ++k) { a_orig[4 * i + k] = main[i * mc + acols[k]]; v_orig[4 * i + k] = main[i * mc + vcols[k]]; }
        // add_pub_memory_in_public_input_section (air.rs:475-494)
        std::vector<Fp> a_aux = a_orig, v_aux = v_orig;
        size_t pm = pub.public_memory.size();
        size_t section = a_orig.size() - pm;
        std::vector<Fp> pm_addrs;
        const MemorySegment* out = pub.segment(1);
        if (out) {
            uint64_t output_section = out->end - out->start;
            uint64_t program_section = pm - output_section;
            for (uint64_t i = 1; i <= program_section; ++i) pm_addrs.push_back(Fp::from_u64(i));
            for (uint64_t a = out->start; a < out->end; ++a) pm_addrs.push_back(Fp::from_u64(a));
        } else {
            for (uint64_t i = 1; i <= pm; ++i) pm_addrs.push_back(Fp::from_u64(i));
        }
        auto mm = pub.memory_map();
        for (size_t i = 0; i < pm; ++i) {
            a_aux[section + i] = pm_addrs[i];
            v_aux[section + i] = mm.at(pm_addrs[i].low_u64());
        }
        // stable sort by address representative (air.rs:519-523)
        std::vector<std::array<uint64_t, 4>> reps(4 * n);
        for (size_t i = 0; i < 4 * n; ++i) a_aux[i].representative(reps[i].data());
        std::vector<uint32_t> idx(4 * n);
        for (size_t i = 0; i < 4 * n; ++i) idx[i] = (uint32_t)i;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) {
            for (int k = 3; k >= 0; --k) if (reps[x][k] != reps[y][k]) return reps[x][k] < reps[y][k];
            return false;
        });
        std::vector<Fp> a_s(4 * n), v_s(4 * n);
        for (size_t i = 0; i < 4 * n; ++i) { a_s[i] = a_aux[idx[i]]; v_s[i] = v_aux[idx[i]]; }
        // memory permutation column (air.rs:525-551)
        std::vector<Fp> den(4 * n);
        for (size_t i = 0; i < 4 * n; ++i) den[i] = z - (a_s[i] + alpha * v_s[i]);
        batch_inverse(den);
        std::vector<Fp> perm(4 * n);
        Fp prod = Fp::one();
        for (size_t i = 0; i < 4 * n; ++i) { prod = prod * ((z - (a_orig[i] + alpha * v_orig[i])) * den[i]); perm[i] = prod; }
        // range check (air.rs:685-703, 552-572)
        std::vector<Fp> off_orig(3 * n);
        std::vector<uint16_t> off_sorted(3 * n);
        for (size_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) { off_orig[3 * i + k] = main[i * mc + OFF_DST + k]; off_sorted[3 * i + k] = (uint16_t)off_orig[3 * i + k].low_u64(); }
        std::sort(off_sorted.begin(), off_sorted.end());
        std::vector<Fp> off_s(3 * n), rden(3 * n), rperm(3 * n);
        for (size_t i = 0; i < 3 * n; ++i) { off_s[i] = Fp::from_u64(off_sorted[i]); rden[i] = zrc - off_s[i]; }
        batch_inverse(rden);
        prod = Fp::one();
        for (size_t i = 0; i < 3 * n; ++i) { prod = prod * (zrc - off_orig[i]) * rden[i]; rperm[i] = prod; }
        // wide format (air.rs:705-728)
        std::vector<Fp> aux(n * 18);
        for (size_t i = 0; i < n; ++i) {
            Fp* r = &aux[i * 18];
            for (int k = 0; k < 3; ++k) r[k] = off_s[3 * i + k];
            for (int k = 0; k < 4; ++k) r[3 + k] = a_s[4 * i + k];
            for (int k = 0; k < 4; ++k) r[7 + k] = v_s[4 * i + k];
            for (int k = 0; k < 4; ++k) r[11 + k] = perm[4 * i + k];
            for (int k = 0; k < 3; ++k) r[15 + k] = rperm[3 * i + k];
        }
        return aux;
    }

    void compute_transition(const Fp* frame, const std::vector<Fp>& rap, Fp* c) const override {
        size_t W = ctx.trace_columns;
        const Fp* curr = frame;
        const Fp* next = frame + W;
        size_t bo = builtin_offset();
        Fp one = Fp::one(), two = Fp::from_u64(2);
        for (size_t i = 0; i < ctx.num_transition_constraints; ++i) c[i] = Fp::zero();
        // compute_instr_constraints (air.rs:869-897)
        for (int i = 0; i < 15; ++i) c[i] = curr[i] * (curr[i] - one);
        c[15] = curr[15];
        Fp b16 = two.pow(16), b32 = two.pow(32), b48 = two.pow(48);
        Fp f0s = Fp::zero();
        for (int i = 14; i >= 0; --i) f0s = curr[i] + two * f0s;
        c[C_INST] = curr[OFF_DST] + b16 * curr[OFF_OP0] + b32 * curr[OFF_OP1] + b48 * f0s - curr[FRAME_INST];
        // compute_operand_constraints (air.rs:899-924)
        const Fp &ap = curr[FRAME_AP], &fp = curr[FRAME_FP], &pc = curr[FRAME_PC];
        Fp b15 = two.pow(15);
        c[C_DST_ADDR] = curr[F_DST_FP] * fp + (one - curr[F_DST_FP]) * ap + (curr[OFF_DST] - b15) - curr[FRAME_DST_ADDR];
        c[C_OP0_ADDR] = curr[F_OP_0_FP] * fp + (one - curr[F_OP_0_FP]) * ap + (curr[OFF_OP0] - b15) - curr[FRAME_OP0_ADDR];
        c[C_OP1_ADDR] = curr[F_OP_1_VAL] * pc + curr[F_OP_1_AP] * ap + curr[F_OP_1_FP] * fp +
                        (one - curr[F_OP_1_VAL] - curr[F_OP_1_AP] - curr[F_OP_1_FP]) * curr[FRAME_OP0] +
                        (curr[OFF_OP1] - b15) - curr[FRAME_OP1_ADDR];
        // compute_register_constraints (air.rs:926-959)
        Fp inst_size = curr[F_OP_1_VAL] + one;
        c[C_NEXT_AP] = curr[FRAME_AP] + curr[F_AP_ADD] * curr[FRAME_RES] + curr[F_AP_ONE] + curr[F_OPC_CALL] * two - next[FRAME_AP];
        c[C_NEXT_FP] = curr[F_OPC_RET] * curr[FRAME_DST] + curr[F_OPC_CALL] * (curr[FRAME_AP] + two) +
                       (one - curr[F_OPC_RET] - curr[F_OPC_CALL]) * curr[FRAME_FP] - next[FRAME_FP];
        c[C_NEXT_PC_1] = (curr[FRAME_T1] - curr[F_PC_JNZ]) * (next[FRAME_PC] - (curr[FRAME_PC] + inst_size));
        c[C_NEXT_PC_2] = curr[FRAME_T0] * (next[FRAME_PC] - (curr[FRAME_PC] + curr[FRAME_OP1])) +
                         (one - curr[F_PC_JNZ]) * next[FRAME_PC] -
                         ((one - curr[F_PC_ABS] - curr[F_PC_REL] - curr[F_PC_JNZ]) * (curr[FRAME_PC] + inst_size) +
                          curr[F_PC_ABS] * curr[FRAME_RES] + curr[F_PC_REL] * (curr[FRAME_PC] + curr[FRAME_RES]));
        c[C_T0] = curr[F_PC_JNZ] * curr[FRAME_DST] - curr[FRAME_T0];
        c[C_T1] = curr[FRAME_T0] * curr[FRAME_RES] - curr[FRAME_T1];
        // compute_opcode_constraints (air.rs:961-978)
        c[C_MUL_1] = curr[FRAME_MUL] - curr[FRAME_OP0] * curr[FRAME_OP1];
        c[C_MUL_2] = curr[F_RES_ADD] * (curr[FRAME_OP0] + curr[FRAME_OP1]) + curr[F_RES_MUL] * curr[FRAME_MUL] +
                     (one - curr[F_RES_ADD] - curr[F_RES_MUL] - curr[F_PC_JNZ]) * curr[FRAME_OP1] -
                     (one - curr[F_PC_JNZ]) * curr[FRAME_RES];
        c[C_CALL_1] = curr[F_OPC_CALL] * (curr[FRAME_DST] - curr[FRAME_FP]);
        c[C_CALL_2] = curr[F_OPC_CALL] * (curr[FRAME_OP0] - (curr[FRAME_PC] + inst_size));
        c[C_ASSERT_EQ] = curr[F_OPC_AEQ] * (curr[FRAME_DST] - curr[FRAME_RES]);
        // enforce_selector (air.rs:980-985)
        for (int i = C_INST; i <= C_ASSERT_EQ; ++i) c[i] = c[i] * curr[FRAME_SELECTOR];
        // memory_is_increasing (air.rs:987-1043)
        const Fp* as = curr + (MEMORY_ADDR_SORTED_0 - bo);
        const Fp* vs = curr + (MEMORY_VALUES_SORTED_0 - bo);
        const Fp& as0n = next[MEMORY_ADDR_SORTED_0 - bo];
        const Fp& vs0n = next[MEMORY_VALUES_SORTED_0 - bo];
        for (int k = 0; k < 3; ++k) {
            c[C_MEMORY_INCREASING_0 + k] = (as[k] - as[k + 1]) * (as[k + 1] - as[k] - one);
            c[C_MEMORY_CONSISTENCY_0 + k] = (vs[k] - vs[k + 1]) * (as[k + 1] - as[k] - one);
        }
        c[C_MEMORY_INCREASING_0 + 3] = (as[3] - as0n) * (as0n - as[3] - one);
        c[C_MEMORY_CONSISTENCY_0 + 3] = (vs[3] - vs0n) * (as0n - as[3] - one);
        // permutation_argument (air.rs:1045-1090)
        const Fp &alpha = rap[0], &z = rap[1], &zrc = rap[2];
        const Fp* p = curr + (PERMUTATION_ARGUMENT_COL_0 - bo);
        const Fp& p0n = next[PERMUTATION_ARGUMENT_COL_0 - bo];
        const Fp* a = curr + FRAME_PC;    // pc, dst_addr, op0_addr, op1_addr
        const Fp* v = curr + FRAME_INST;  // inst, dst, op0, op1
        for (int k = 1; k <= 3; ++k)
            c[C_PERMUTATION_ARGUMENT_0 + k - 1] = (z - (as[k] + alpha * vs[k])) * p[k] - (z - (a[k] + alpha * v[k])) * p[k - 1];
        c[C_PERMUTATION_ARGUMENT_0 + 3] = (z - (as0n + alpha * vs0n)) * p0n - (z - (next[FRAME_PC] + alpha * next[FRAME_INST])) * p[3];
        // permutation_argument_range_check (air.rs:1092-1135)
        const Fp* rc = curr + (RANGE_CHECK_COL_1 - bo);
        const Fp& rc0n = next[RANGE_CHECK_COL_1 - bo];
        c[C_RANGE_CHECK_INCREASING_0] = (rc[0] - rc[1]) * (rc[1] - rc[0] - one);
        c[C_RANGE_CHECK_INCREASING_0 + 1] = (rc[1] - rc[2]) * (rc[2] - rc[1] - one);
        c[C_RANGE_CHECK_INCREASING_0 + 2] = (rc[2] - rc0n) * (rc0n - rc[2] - one);
        const Fp* q = curr + (PERMUTATION_ARGUMENT_RANGE_CHECK_COL_1 - bo);
        const Fp& q0n = next[PERMUTATION_ARGUMENT_RANGE_CHECK_COL_1 - bo];
        c[C_RANGE_CHECK_0] = (zrc - rc[1]) * q[1] - (zrc - curr[OFF_OP0]) * q[0];
        c[C_RANGE_CHECK_0 + 1] = (zrc - rc[2]) * q[2] - (zrc - curr[OFF_OP1]) * q[1];
        c[C_RANGE_CHECK_0 + 2] = (zrc - rc0n) * q0n - (zrc - next[OFF_DST]) * q[2];
        // range_check_builtin (air.rs:1141-1160)
        if (has_rc_builtin) {
            Fp acc = Fp::zero();
            for (int k = 7; k >= 0; --k) acc = acc * b16 + curr[RC_0 + k];
            c[C_RANGE_CHECK_BUILTIN] = acc - curr[RC_VALUE];
        }
    }

    std::vector<BoundaryConstraint> boundary_constraints(const std::vector<Fp>& rap) const override {
        const Fp &alpha = rap[0], &z = rap[1];
        size_t bo = builtin_offset();
        size_t final_index = trace_len - 1;
        Fp prod = Fp::one();
        for (auto& kv : pub.public_memory) prod = prod * (z - (kv.first + alpha * kv.second));
        Fp permutation_final = z.pow(pub.public_memory.size()) * prod.inv();
        std::vector<BoundaryConstraint> b;
        b.push_back({MEM_A_TRACE_OFFSET, 0, pub.pc_init});
        b.push_back({MEM_P_TRACE_OFFSET, 0, pub.ap_init});
        b.push_back({MEM_A_TRACE_OFFSET, (size_t)pub.num_steps - 1, pub.pc_final});
        b.push_back({MEM_P_TRACE_OFFSET, (size_t)pub.num_steps - 1, pub.ap_final});
        b.push_back({PERMUTATION_ARGUMENT_COL_0 + 3 - bo, final_index, permutation_final});
        b.push_back({PERMUTATION_ARGUMENT_RANGE_CHECK_COL_1 + 2 - bo, final_index, Fp::one()});
        b.push_back({RANGE_CHECK_COL_1 - bo, 0, Fp::from_u64(pub.rc_min)});
        b.push_back({RANGE_CHECK_COL_3 - bo, final_index, Fp::from_u64(pub.rc_max)});
        return b;
    }
};

}  // namespace oracle

// Device kernels of the STARK rounds (see stark_kernels.h).
#include "stark_kernels.h"
#include "keccak.h"

namespace sp {

__device__ __forceinline__ fe sk_ld(const fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void sk_st(fe* p, const fe& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
__device__ __forceinline__ fe operator+(const fe& a, const fe& b) { return fe_add(a, b); }
__device__ __forceinline__ fe operator-(const fe& a, const fe& b) { return fe_sub(a, b); }
__device__ __forceinline__ fe operator*(const fe& a, const fe& b) { return fe_mul(a, b); }

// w_N^e (e in [0, N)) from the half table
__device__ __forceinline__ fe root_pow(const fe* tw, uint32_t e, uint32_t logN) {
    uint32_t half = 1u << (logN - 1);
    fe w = sk_ld(tw + (e & (half - 1)));
    return (e & half) ? fe_neg(w) : w;
}

// ---------------------------------------------------------------------------------------------- power tables
struct PowTableArgs { fe pw[32]; fe c; };
__global__ void __launch_bounds__(256) power_table_kernel(fe* out, uint64_t count, uint32_t bitrev_bits, PowTableArgs a) {
    uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= count) return;
    uint32_t e = bitrev_bits ? (__brev((uint32_t)q) >> (32 - bitrev_bits)) : (uint32_t)q;
    fe acc = a.c;
    for (uint32_t i = 0; i < 32; ++i)
        if ((e >> i) & 1) acc = fe_mul(acc, a.pw[i]);
    sk_st(out + q, acc);
}
int gen_power_table(hipStream_t st, fe* out, uint64_t count, uint32_t bitrev_bits, const fe& base, const fe& c) {
    PowTableArgs a;
    fe cur = base;
    for (int i = 0; i < 32; ++i) { a.pw[i] = cur; cur = fe_sqr(cur); }
    a.c = c;
    hipLaunchKernelGGL(power_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, out, count, bitrev_bits, a);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// ---------------------------------------------------------------------------------------------- x_i - point
struct PointsArgs { fe h; fe pt[AIR_MAX_OFFSETS + 1]; };
__device__ __forceinline__ uint32_t shard_global_index(uint32_t i_loc, ShardMap m) {
    const uint32_t lb_loc = m.logb - m.shard_log;
    const uint32_t c_loc = i_loc & ((1u << lb_loc) - 1u), q = i_loc >> lb_loc;
    return (q << m.logb) + (c_loc << m.shard_log) + m.shard_rank;
}
__global__ void __launch_bounds__(256) coset_minus_points_kernel(fe* den, uint64_t N, uint32_t logN, const fe* roots, uint32_t ndist, PointsArgs a, ShardMap sm) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    fe x = fe_mul(root_pow(roots, shard_global_index((uint32_t)i, sm), logN), a.h);
    for (uint32_t d = 0; d < ndist; ++d) sk_st(den + (uint64_t)d * N + i, fe_sub(x, a.pt[d]));
}
int coset_minus_points(hipStream_t st, fe* den, uint64_t N, uint32_t logN, const fe* roots_N, const fe& h, const fe* points_host, uint32_t ndist, ShardMap sm) {
    if (ndist > AIR_MAX_OFFSETS + 1) return SP_E_INVALID_ARG;
    PointsArgs a;
    a.h = h;
    for (uint32_t d = 0; d < ndist; ++d) a.pt[d] = points_host[d];
    hipLaunchKernelGGL(coset_minus_points_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, den, N, logN, roots_N, ndist, a, sm);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// ---------------------------------------------------------------------------------------------- Cairo composition
// Column ids of the main trace (reference src/cairo/air.rs:73-110); aux columns start at main_cols:
// +0..2 sorted offsets, +3..6 sorted addresses, +7..10 sorted values, +11..14 memory permutation, +15..17 rc permutation.
enum {
    K_RES = 16, K_AP = 17, K_FP = 18, K_PC = 19, K_DST_ADDR = 20, K_OP0_ADDR = 21, K_OP1_ADDR = 22, K_INST = 23, K_DST = 24,
    K_OP0 = 25, K_OP1 = 26, K_OFF_DST = 27, K_OFF_OP0 = 28, K_OFF_OP1 = 29, K_T0 = 30, K_T1 = 31, K_MUL = 32, K_SEL = 33,
    K_RC0 = 34, K_RCV = 42
};

// Register pressure: left alone, the compiler hoists every column load of this kernel (> 60 field elements) to the
// top, needs all 256 VGPRs plus scratch and runs at one wave per SIMD (round-1 profile: 16.5 ms at n = 2^20, b = 8,
// i.e. < half of the arithmetic rate).  The constraint groups are therefore evaluated as PHASES: the loads of a
// phase are addressed through an offset that is an opaque function (always zero) of the previous phase's
// accumulators, so they cannot be scheduled before that phase has finished; values shared between phases are
// simply re-loaded (L1/L2 hits).
__device__ __forceinline__ uint32_t phase_gate(const fe& a, const fe& b) {
    uint32_t z;
    asm volatile("v_and_b32 %0, 0, %1" : "=v"(z) : "v"(a.v[0] ^ b.v[0]));
    return z;  // == 0, but only the hardware knows
}

// CHECK = false: composition evaluations.  Point i of the launch is element e = i << stride_log of every LDE column
//   (stride_log > 0: only the cosets 0 and b/2, i.e. the 2n points h w_2n^i, which fix a composition polynomial of
//   degree < 2n); the frame's second row is element e + b_loc; binv = [ndist][count]; out[i].
// CHECK = true: the same constraint expressions on the TRACE itself (cols = natural-order columns of n rows, second
//   row = i + 1 mod n): *flag |= 1 if any transition constraint is non-zero on a row it is enforced on, or a boundary
//   value differs.  A clean flag means every quotient C_k / Z_k is a polynomial, hence deg H < 2n.
#ifndef SP_COMP_WAVES
#define SP_COMP_WAVES 3
#endif
// bytes of LDS the composition kernel may take for its per-coset coefficient table: 128 KB of gfx950's 160 KB a CU, and never more than
// the device grants one work-group (asked once per device: a part with 64 KB per group then takes the table from global memory)
static size_t comp_lds_limit() {
    static size_t cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!cached[dev]) {
        int per_block = 0;
        if (hipDeviceGetAttribute(&per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || per_block <= 0) per_block = 64 << 10;
        cached[dev] = std::min<size_t>(128u << 10, (size_t)per_block);
    }
    return cached[dev];
}
// STAGED: the per-coset coefficient table is copied into LDS first (every blowup factor up to 64); not STAGED (blowup 128: the table
// would take 237 KB): read from the constant block in global memory.  A template parameter, not a run-time choice: a pointer that may
// point into either address space becomes a flat pointer, and the flat accesses to LDS raised memory-aperture violations.
template <bool CHECK, bool STAGED = !CHECK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SP_COMP_WAVES, SP_COMP_WAVES)))
cairo_composition_kernel(const fe* __restrict__ cols, uint64_t count, uint64_t col_len, uint32_t stride_log, uint32_t logN, uint32_t logb,
                         const fe* __restrict__ roots, const CompositionConsts* __restrict__ K, const fe* __restrict__ binv,
                         fe* __restrict__ out, int* __restrict__ flag, uint32_t shard_log, uint32_t shard_rank, uint64_t row0, uint64_t row_end) {
    extern __shared__ __attribute__((aligned(16))) uint4 sh_raw[];
    fe* sh_coef = reinterpret_cast<fe*>(sh_raw);  // [b][T + B]
    const uint32_t b = 1u << logb;
    const uint32_t T = K->n_transitions, B = K->n_boundary, W = T + B;
    if (STAGED) {
        for (uint32_t t = threadIdx.x; t < b * W; t += 256) {
            uint32_t c = t / W, k = t % W;
            sh_coef[t] = K->coef[c][k];
        }
        __syncthreads();
    }
    // (CHECK: rows row0 .. row_end - 1 of the count-row trace - a rank of a sharded prover checks its own slice of the rows)
    const uint64_t i = (CHECK ? row0 : 0) + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (CHECK ? row_end : count)) return;
    // element indices are LOCAL under coset sharding (this rank holds b_loc = b >> shard_log cosets); logN, b, c are global
    const uint32_t b_loc = b >> shard_log;
    const ShardMap sm{logb, shard_log, shard_rank};
    const uint64_t el = CHECK ? i : (i << stride_log);   // local natural-order index of the point
    const uint32_t iglob = CHECK ? 0u : shard_global_index((uint32_t)el, sm);
    const uint32_t c = iglob & (b - 1);
    // frame row offset 1 = LDE index + blowup (reference frame.rs:40-59): same coset; on the trace: the next row
    const uint64_t elnext = (el + (CHECK ? 1u : b_loc)) & (col_len - 1);
    // the LDE columns are coset-major (the trace of CHECK mode is a plain array): e, enext = storage indices
    const LdeOrder ord{CHECK ? 0u : 1u, logb - shard_log, logN - logb};
    const uint64_t e = ord.at(el), enext = ord.at(elnext);
    auto coef_at = [&](uint32_t k) -> fe { if (STAGED) return sh_coef[c * W + k]; else return K->coef[c][k]; };
    const uint32_t A = K->main_cols;
    uint32_t gate = 0;  // see phase_gate
    auto cur = [&](uint32_t col) { return sk_ld(cols + (uint64_t)col * col_len + e + gate); };
    auto nxt = [&](uint32_t col) { return sk_ld(cols + (uint64_t)col * col_len + enext + gate); };

    const fe one = fe_one();
    fe S0 = fe_zero(), S1 = fe_zero(), S2 = fe_zero(), S3 = fe_zero();
    // S0: no selector, no exemption; S1: exempted; S2: selector; S3: selector and exempted.
    // CHECK mode: S_g.v[0] collects "some constraint of group g is non-zero here" instead of the weighted sum.
    auto acc = [&](fe& S, uint32_t k, const fe& v) {
        if (CHECK) { uint32_t o = 0; for (int l = 0; l < 8; ++l) o |= v.v[l]; S.v[0] |= o; }
        else S = S + coef_at(k) * v;
    };

    // --- phase 0: flags (air.rs:869-881) and the instruction word (air.rs:883-896)
    {
        fe f0s = fe_zero();
#pragma unroll 1
        for (int k = 14; k >= 0; --k) {
            fe f = cur(k);
            acc(S0, k, fe_sqr(f) - f);      // (the dedicated square: 36 + 8 multiply-adds)
            f0s = f + (f0s + f0s);
        }
        acc(S0, 15, cur(15));
        fe c16 = cur(K_OFF_DST) + K->b16 * cur(K_OFF_OP0) + K->b32 * cur(K_OFF_OP1) + K->b48 * f0s - cur(K_INST);
        acc(S2, 16, c16);
    }
    gate = phase_gate(S0, S2);
    // --- phase 1: operand constraints (air.rs:899-924)
    {
        const fe ap = cur(K_AP), fp = cur(K_FP);
        fe d = fp - ap;
        acc(S2, 17, ap + cur(0) * d + (cur(K_OFF_DST) - K->b15) - cur(K_DST_ADDR));
        acc(S2, 18, ap + cur(1) * d + (cur(K_OFF_OP0) - K->b15) - cur(K_OP0_ADDR));
        fe f2 = cur(2), f3 = cur(3), f4 = cur(4);
        fe c19 = f2 * cur(K_PC) + f4 * ap + f3 * fp + (one - f2 - f4 - f3) * cur(K_OP0) + (cur(K_OFF_OP1) - K->b15) - cur(K_OP1_ADDR);
        acc(S2, 19, c19);
    }
    gate = phase_gate(S2, S2);
    // --- phase 2: register constraints ap, fp (air.rs:926-938)
    {
        const fe ap = cur(K_AP), fp = cur(K_FP);
        fe f12 = cur(12), f13 = cur(13);
        fe c20 = ap + cur(10) * cur(K_RES) + cur(11) + (f12 + f12) - nxt(K_AP);
        acc(S3, 20, c20);
        fe c21 = f13 * cur(K_DST) + f12 * (ap + K->two) + (one - f13 - f12) * fp - nxt(K_FP);
        acc(S3, 21, c21);
    }
    gate = phase_gate(S3, S3);
    // --- phase 3: register constraints pc, t0, t1 (air.rs:940-959)
    {
        const fe pc = cur(K_PC), res = cur(K_RES), f9 = cur(9);
        const fe pc_size = pc + (cur(2) + one);  // frame_inst_size (air.rs:1137-1139)
        fe npc = nxt(K_PC);
        fe t0 = cur(K_T0), t1 = cur(K_T1);
        acc(S3, 22, (t1 - f9) * (npc - pc_size));
        fe f7 = cur(7), f8 = cur(8);
        fe c23 = t0 * (npc - (pc + cur(K_OP1))) + (one - f9) * npc -
                 ((one - f7 - f8 - f9) * pc_size + f7 * res + f8 * (pc + res));
        acc(S3, 23, c23);
        acc(S2, 24, f9 * cur(K_DST) - t0);
        acc(S2, 25, t0 * res - t1);
    }
    gate = phase_gate(S2, S3);
    // --- phase 4: opcode constraints (air.rs:961-978)
    {
        const fe res = cur(K_RES), f9 = cur(9);
        fe op0 = cur(K_OP0), op1 = cur(K_OP1), mul = cur(K_MUL);
        fe f5 = cur(5), f6 = cur(6);
        acc(S2, 26, mul - op0 * op1);
        fe c27 = f5 * (op0 + op1) + f6 * mul + (one - f5 - f6 - f9) * op1 - (one - f9) * res;
        acc(S2, 27, c27);
        fe f12 = cur(12), dst = cur(K_DST);
        acc(S2, 28, f12 * (dst - cur(K_FP)));
        acc(S2, 29, f12 * (op0 - (cur(K_PC) + (cur(2) + one))));
        acc(S2, 30, cur(14) * (dst - res));
    }
    gate = phase_gate(S2, S2);
    // --- phase 5: memory (air.rs:987-1043) and permutation argument (air.rs:1045-1090)
    {
        const fe alpha = K->rap[0], z = K->rap[1];
        fe a_prev = cur(A + 3), v_prev = cur(A + 7), p_prev = cur(A + 11);
#pragma unroll 1
        for (uint32_t k = 1; k <= 4; ++k) {
            // k = 1..3: next sorted cell of this row; k = 4: first sorted cell of the next row
            const uint64_t row = ((k < 4) ? e : enext) + gate;
            const uint32_t kk = (k < 4) ? k : 0;
            fe a_k = sk_ld(cols + (uint64_t)(A + 3 + kk) * col_len + row);
            fe v_k = sk_ld(cols + (uint64_t)(A + 7 + kk) * col_len + row);
            fe p_k = sk_ld(cols + (uint64_t)(A + 11 + kk) * col_len + row);
            fe step = a_k - a_prev - one;
            fe inc = (a_prev - a_k) * step;           // MEMORY_INCREASING_{k-1}
            fe cons = (v_prev - v_k) * step;          // MEMORY_CONSISTENCY_{k-1}
            // original (unsorted) access k: (dst_addr,dst), (op0_addr,op0), (op1_addr,op1), then next row's (pc,inst)
            fe a_o = sk_ld(cols + (uint64_t)(K_PC + kk) * col_len + row);
            fe v_o = sk_ld(cols + (uint64_t)(K_INST + kk) * col_len + row);
            fe perm = (z - (a_k + alpha * v_k)) * p_k - (z - (a_o + alpha * v_o)) * p_prev;  // PERMUTATION_ARGUMENT_{k-1}
            if (k < 4) {
                acc(S0, 31 + k - 1, inc); acc(S0, 35 + k - 1, cons); acc(S0, 39 + k - 1, perm);
            } else {
                acc(S1, 34, inc); acc(S1, 38, cons); acc(S1, 42, perm);
            }
            a_prev = a_k; v_prev = v_k; p_prev = p_k;
        }
    }
    gate = phase_gate(S0, S1);
    // --- phase 6: range check (air.rs:1092-1135)
    {
        const fe zrc = K->rap[2];
        fe rc0 = cur(A + 0), rc1 = cur(A + 1), rc2 = cur(A + 2), rc0n = nxt(A + 0);
        acc(S0, 43, (rc0 - rc1) * (rc1 - rc0 - one));
        acc(S0, 44, (rc1 - rc2) * (rc2 - rc1 - one));
        acc(S1, 45, (rc2 - rc0n) * (rc0n - rc2 - one));
        fe q0 = cur(A + 15), q1 = cur(A + 16), q2 = cur(A + 17), q0n = nxt(A + 15);
        acc(S0, 46, (zrc - rc1) * q1 - (zrc - cur(K_OFF_OP0)) * q0);
        acc(S0, 47, (zrc - rc2) * q2 - (zrc - cur(K_OFF_OP1)) * q1);
        acc(S0, 48, (zrc - rc0n) * q0n - (zrc - nxt(K_OFF_DST)) * q2);
    }
    gate = phase_gate(S0, S1);
    // --- range-check builtin (air.rs:1141-1160)
    if (K->has_rc_builtin) {
        fe accv = fe_zero();
#pragma unroll 1
        for (int k = 7; k >= 0; --k) accv = accv * K->b16 + cur(K_RC0 + k);
        acc(S0, 49, accv - cur(K_RCV));
        gate = phase_gate(S0, S0);
    }
    const fe sel = cur(K_SEL);
    if (CHECK) {
        // enforced rows: every row for S0/S2 (S2 only where the selector is non-zero), every row but the last for S1/S3
        const bool last = (i == count - 1);
        const bool selnz = !fe_is_zero(sel);
        bool bad = S0.v[0] != 0 || (!last && S1.v[0] != 0) || (selnz && (S2.v[0] != 0 || (!last && S3.v[0] != 0)));
        for (uint32_t j = 0; j < B; ++j)
            if (i == K->bstep[j] && !fe_eq(cur(K->bcol[j]), K->bvalue[j])) bad = true;
        if (bad) atomicOr(flag, 1);
        return;
    }
    // --- combine (evaluator.rs:205-253): zerofier * (sum + exemption * sum_exempted)
    const fe x = root_pow(roots, iglob, logN) * K->h;
    fe total = K->zerofier[c] * ((S0 + sel * S2) + (x - K->g_last) * (S1 + sel * S3));
    // --- boundary term (evaluator.rs:58-115)
#pragma unroll 1
    for (uint32_t j = 0; j < B; ++j) {
        fe num = cur(K->bcol[j]) - K->bvalue[j];
        total = total + coef_at(T + j) * num * sk_ld(binv + (uint64_t)K->bden[j] * count + i);
    }
    sk_st(out + i, total);
}

int cairo_composition(hipStream_t st, const fe* lde, uint64_t count, uint64_t col_len, uint32_t stride_log, uint32_t logN, uint32_t logb,
                      const fe* roots_N, const CompositionConsts* consts_dev, const fe* binv, fe* out, uint32_t shard_log, uint32_t shard_rank) {
    if ((1u << logb) > CAIRO_MAX_BLOWUP) { sp_set_error("composition: blowup factor > 128 unsupported"); return SP_E_UNSUPPORTED; }
    size_t lds = (size_t)(1u << logb) * (CAIRO_MAX_TRANSITIONS + CAIRO_MAX_BOUNDARY) * sizeof(fe);
    if (lds > comp_lds_limit())
        hipLaunchKernelGGL((cairo_composition_kernel<false, false>), dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, lde, count, col_len, stride_log,
                           logN, logb, roots_N, consts_dev, binv, out, (int*)nullptr, shard_log, shard_rank, (uint64_t)0, count);
    else
    hipLaunchKernelGGL((cairo_composition_kernel<false, true>), dim3((unsigned)((count + 255) / 256)), dim3(256), lds, st, lde, count, col_len, stride_log,
                       logN, logb, roots_N, consts_dev, binv, out, (int*)nullptr, shard_log, shard_rank, (uint64_t)0, count);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

int cairo_trace_check(hipStream_t st, const fe* trace, uint64_t n, const CompositionConsts* consts_dev, int* flag_dev, uint64_t row0, uint64_t rows) {
    if (rows == 0) rows = n - row0;
    if (row0 + rows > n) return SP_E_INVALID_ARG;
    // rows row0 .. row0 + rows - 1 (the frame's next row wraps modulo n as on the whole trace)
    hipLaunchKernelGGL((cairo_composition_kernel<true, false>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, trace, n, n, 0u, 0u, 0u,
                       (const fe*)nullptr, consts_dev, (const fe*)nullptr, (fe*)nullptr, flag_dev, 0u, 0u, row0, row0 + rows);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// ---------------------------------------------------------------------------------------------- composition split
struct SplitArgs { fe hinv; };
__global__ void __launch_bounds__(256) split_composition_kernel(const fe* X, uint64_t n, uint32_t logb, const fe* t2, SplitArgs a, fe* H1s, fe* H2s) {
    uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    uint64_t N = n << logb;
    uint64_t pos = logb ? (q << (logb - 1)) : 0;
    fe t = sk_ld(t2 + q);
    if (logb == 0) {
        // blowup 1: X has n entries in bit-reversed order of k; even k -> first half. Not used by the prover (b >= 2).
        return;
    }
    sk_st(H1s + q, fe_mul(sk_ld(X + pos), t));
    sk_st(H2s + q, fe_mul(fe_mul(sk_ld(X + (N >> 1) + pos), t), a.hinv));
}
int split_composition(hipStream_t st, const fe* X, uint64_t n, uint32_t logb, const fe* t2, const fe& hinv, fe* H1s, fe* H2s) {
    if (logb == 0) { sp_set_error("split_composition: blowup factor must be >= 2"); return SP_E_UNSUPPORTED; }
    SplitArgs a; a.hinv = hinv;
    hipLaunchKernelGGL(split_composition_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, n, logb, t2, a, H1s, H2s);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// flag |= 1 when the composition polynomial has a non-zero coefficient of degree >= 2n (a trace that violates its
// constraints): X = unscaled bit-reversed size-N inverse transform; coefficient k sits at rev_N(k), so "k >= 2n" means
// "some of the low log2(b)-1 bits of the position inside its half are set".
__global__ void __launch_bounds__(256) high_coeff_check_kernel(const fe* X, uint64_t N, uint32_t logb, int* flag) {
    uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= N) return;
    if (logb < 2) return;  // b = 2: every coefficient index below N is < 2n
    uint64_t low = q & ((1ULL << (logb - 1)) - 1ULL);
    if (low == 0) return;
    if (!fe_is_zero(sk_ld(X + q))) atomicOr(flag, 1);
}
int high_coeff_check(hipStream_t st, const fe* X, uint64_t N, uint32_t logb, int* flag_dev) {
    hipLaunchKernelGGL(high_coeff_check_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, X, N, logb, flag_dev);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}
// General split (no degree assumption): H1f[q] = X[q] * t[q], H2f[q] = X[N/2 + q] * t[q] * hinv for q < N/2,
// t[q] = N^-1 h^(-rev_{N/2}(q)); both are h-scaled bit-reversed coefficient arrays of N/2 entries.
__global__ void __launch_bounds__(256) split_full_kernel(const fe* X, uint64_t half, const fe* t, SplitArgs a, fe* H1f, fe* H2f) {
    uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= half) return;
    fe tq = sk_ld(t + q);
    sk_st(H1f + q, fe_mul(sk_ld(X + q), tq));
    sk_st(H2f + q, fe_mul(fe_mul(sk_ld(X + half + q), tq), a.hinv));
}
int split_composition_full(hipStream_t st, const fe* X, uint64_t N, const fe* t_half, const fe& hinv, fe* H1f, fe* H2f) {
    SplitArgs a; a.hinv = hinv;
    uint64_t half = N >> 1;
    hipLaunchKernelGGL(split_full_kernel, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, st, X, half, t_half, a, H1f, H2f);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// ---------------------------------------------------------------------------------------------- OOD fold
__global__ void __launch_bounds__(256) fold_eval_kernel(const fe* in, uint64_t in_vec_stride, uint32_t in_points, uint64_t Mq, uint32_t T,
                                                        const fe* yp, uint32_t points, fe* out) {
    uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= Mq) return;
    const uint32_t v = blockIdx.y, p = blockIdx.z;
    const fe* src = in + (uint64_t)v * in_vec_stride + (in_points > 1 ? (uint64_t)p * Mq * T : 0);
    // lazily reduced sum: products in [0, 2p), eight of them on top of an accumulator below 2p stay below 18p < 2^256
    fe acc = fe_zero();
    for (uint32_t t = 0; t < T; ++t) {
        acc = fe_add_raw(acc, fe_mul_lazy(sk_ld(src + (uint64_t)t * Mq + q), yp[p * T + t]));
        if ((t & 7u) == 7u) acc = fe_reduce_lazy_2p(acc);
    }
    sk_st(out + ((uint64_t)v * points + p) * Mq + q, fe_canonical_lazy(acc));
}
int fold_eval_level(hipStream_t st, const fe* in, uint64_t in_vec_stride, uint32_t in_points, uint64_t M, uint32_t l,
                    const fe* yp, uint32_t points, uint32_t vectors, fe* out) {
    uint64_t Mq = M >> l;
    dim3 grid((unsigned)((Mq + 255) / 256), vectors, points);
    hipLaunchKernelGGL(fold_eval_kernel, grid, dim3(256), 0, st, in, in_vec_stride, in_points, Mq, 1u << l, yp, points, out);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// ---------------------------------------------------------------------------------------------- DEEP composition
// Point q of the launch is element (q << shift) of every column (shift > 0: one coset of the LDE domain only);
// inv holds the three inverse arrays with `count` entries each, out[q] the value.
template <int MAXR>
__global__ void __launch_bounds__(256) deep_kernel(const fe* __restrict__ lde, const fe* __restrict__ h1, const fe* __restrict__ h2, uint64_t count,
                                                   uint64_t col_stride, uint32_t shift, const DeepConsts* __restrict__ K,
                                                   const fe* __restrict__ inv, fe* __restrict__ out, LdeOrder ord) {
    uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= count) return;
    const uint64_t i = ord.at(q << shift);   // coset-major columns: the one coset of the n-point evaluation is contiguous
    const uint32_t C = K->cols, R = K->rows;
    fe a[MAXR];
#pragma unroll
    for (int k = 0; k < MAXR; ++k) a[k] = fe_zero();
    // lazily reduced sums over the columns: products in [0, 2p), eight of them on top of an accumulator below 2p stay below 18p < 2^256
    for (uint32_t j = 0; j < C; ++j) {
        fe t = sk_ld(lde + (uint64_t)j * col_stride + i);
#pragma unroll
        for (int k = 0; k < MAXR; ++k)
            if ((uint32_t)k < R) {
                a[k] = fe_add_raw(a[k], fe_mul_lazy(t, K->gammas[k][j]));
                if ((j & 7u) == 7u) a[k] = fe_reduce_lazy_2p(a[k]);
            }
    }
#pragma unroll
    for (int k = 0; k < MAXR; ++k) a[k] = fe_canonical_lazy(a[k]);
    fe hh = K->gamma_h1 * sk_ld(h1 + i) + K->gamma_h2 * sk_ld(h2 + i) - K->c_h;
    fe r = hh * sk_ld(inv + (uint64_t)R * count + q);
#pragma unroll
    for (int k = 0; k < MAXR; ++k)
        if ((uint32_t)k < R) r = r + (a[k] - K->c_t[k]) * sk_ld(inv + (uint64_t)k * count + q);
    sk_st(out + q, r);
}
int deep_composition(hipStream_t st, const fe* lde, const fe* h1, const fe* h2, uint64_t count, uint64_t col_stride, uint32_t shift,
                     const DeepConsts* consts_dev, const fe* inv, fe* out, LdeOrder order, uint32_t frame_rows) {
    // (the accumulators of unused frame rows would still take registers: two instantiations)
    if (frame_rows <= 2) hipLaunchKernelGGL(deep_kernel<2>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, lde, h1, h2, count, col_stride, shift, consts_dev, inv, out, order);
    else hipLaunchKernelGGL(deep_kernel<AIR_MAX_OFFSETS>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, lde, h1, h2, count, col_stride, shift, consts_dev, inv, out, order);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// ---------------------------------------------------------------------------------------------- FRI fold
struct FriArgs { fe half, c; };
// Sharded layers: this rank holds the elements with global index i = (local << shard_log) | shard_rank; the partner
// i + M/2 has the same residue, so the fold is local and only the twiddle exponent needs the global index.
__global__ void __launch_bounds__(256) fri_fold_kernel(const fe* cur, fe* next, uint64_t Mh, uint32_t logN, uint32_t layer, const fe* roots, FriArgs a,
                                                       uint32_t shard_log, uint32_t shard_rank, const fe* c_dev) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Mh) return;
    if (c_dev) a.c = sk_ld(c_dev);   // zeta * half / offset left in device memory by the previous layer's root launch (merkle.h, FriChallenge)
    fe x = sk_ld(cur + i), y = sk_ld(cur + Mh + i);
    uint32_t Nm = (1u << logN) - 1;
    const uint32_t ig = ((uint32_t)i << shard_log) | shard_rank;
    uint32_t e = (0u - (ig << layer)) & Nm;  // w_M^-i = w_N^(-i 2^layer)
    fe w = root_pow(roots, e, logN);
    sk_st(next + i, a.half * (x + y) + a.c * (w * (x - y)));
}
// The fold and the leaf hash of the layer it produces in one launch (FriLayer::new, fri_commitment.rs:30-47: the leaf of a FRI tree
// is Keccak256 of the element's canonical 32-byte big-endian encoding): the folded element never travels to HBM and back before it is
// hashed, and every layer of the commit phase is one launch and one dependent latency shorter.
__global__ void __launch_bounds__(256) fri_fold_hash_kernel(const fe* cur, fe* next, uint64_t Mh, uint32_t logN, uint32_t layer, const fe* roots, FriArgs a,
                                                            const fe* c_dev, digest32* leaves_out) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Mh) return;
    if (c_dev) a.c = sk_ld(c_dev);
    fe x = sk_ld(cur + i), y = sk_ld(cur + Mh + i);
    uint32_t Nm = (1u << logN) - 1;
    uint32_t e = (0u - ((uint32_t)i << layer)) & Nm;
    fe w = root_pow(roots, e, logN);
    const fe v = a.half * (x + y) + a.c * (w * (x - y));
    sk_st(next + i, v);
    const fe raw = fe_from_mont(v);
    uint64_t s[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) s[k] = 0;
#pragma unroll
    for (uint32_t l = 0; l < 4; ++l) s[l] = sp_bswap64((uint64_t)raw.v[2 * (3 - l)] | ((uint64_t)raw.v[2 * (3 - l) + 1] << 32));
    s[4] = 0x01ULL;
    s[16] = 0x8000000000000000ULL;
    sp_keccak_f1600_dev(s);
    digest32 d;
    d.w[0] = s[0]; d.w[1] = s[1]; d.w[2] = s[2]; d.w[3] = s[3];
    leaves_out[i] = d;
}
int fri_fold_hash(hipStream_t st, const fe* cur, fe* next, uint64_t M, uint32_t logN, uint32_t layer, const fe* roots_N, const fe& half, const fe& c,
                  const fe* c_dev, digest32* leaves_out) {
    uint64_t Mh = M >> 1;
    FriArgs a; a.half = half; a.c = c;
    hipLaunchKernelGGL(fri_fold_hash_kernel, dim3((unsigned)((Mh + 255) / 256)), dim3(256), 0, st, cur, next, Mh, logN, layer, roots_N, a, c_dev, leaves_out);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

int fri_fold(hipStream_t st, const fe* cur, fe* next, uint64_t M, uint32_t logN, uint32_t layer, const fe* roots_N, const fe& half, const fe& c,
             uint32_t shard_log, uint32_t shard_rank, const fe* c_dev) {
    uint64_t Mh = M >> 1;   // M = elements this rank holds
    FriArgs a; a.half = half; a.c = c;
    hipLaunchKernelGGL(fri_fold_kernel, dim3((unsigned)((Mh + 255) / 256)), dim3(256), 0, st, cur, next, Mh, logN, layer, roots_N, a, shard_log, shard_rank, c_dev);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// ---------------------------------------------------------------------------------------------- grinding
struct GrindArgs { uint64_t ch[4]; };
__global__ void __launch_bounds__(256) grind_kernel(GrindArgs a, uint32_t factor, uint64_t start, uint64_t count, unsigned long long* result) {
    uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    // ranges are queued in increasing order without host synchronisation: once an earlier range has produced a nonce,
    // nothing in a later range can be the minimum
    if (*reinterpret_cast<volatile unsigned long long*>(result) < start) return;
    uint64_t nonce = start + t;
    uint64_t s[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) s[k] = 0;
    s[0] = a.ch[0]; s[1] = a.ch[1]; s[2] = a.ch[2]; s[3] = a.ch[3];
    s[4] = nonce;             // 8 bytes little-endian
    s[5] = 0x01ULL;           // Keccak padding starts at byte 40
    s[16] = 0x8000000000000000ULL;
    sp_keccak_f1600_dev(s);
    uint64_t head = sp_bswap64(s[0]);  // first 8 digest bytes read big-endian
    uint64_t mask = factor >= 64 ? ~0ULL : ((1ULL << factor) - 1ULL);
    if ((head & mask) == 0) atomicMin(result, (unsigned long long)nonce);
}
int grind_range(hipStream_t st, const uint8_t challenge[32], uint8_t factor, uint64_t start, uint64_t count, unsigned long long* result_dev) {
    GrindArgs a;
    __builtin_memcpy(a.ch, challenge, 32);
    hipLaunchKernelGGL(grind_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, a, (uint32_t)factor, start, count, result_dev);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// ---------------------------------------------------------------------------------------------- gathers
__global__ void __launch_bounds__(128) gather_jobs_kernel(const GatherJob* jobs, const uint64_t* idx, fe* out) {
    const GatherJob jb = jobs[blockIdx.y];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= jb.count * jb.width) return;
    const uint32_t r = t / jb.width, j = t % jb.width;
    const uint64_t at = idx[jb.idx_off + r];
    if (jb.kind == 0) {
        sk_st(out + jb.out_off + t, sk_ld(static_cast<const fe*>(jb.base) + (uint64_t)j * jb.stride_or_leaves + at));
    } else {
        uint64_t p = at + jb.stride_or_leaves - 1;
        for (uint32_t k = 0; k < j; ++k) p = (p - 1) >> 1;
        const uint64_t sib = (p & 1) ? p + 1 : p - 1;
        sk_st(out + jb.out_off + t, sk_ld(static_cast<const fe*>(jb.base) + sib));
    }
}
int gather_jobs(hipStream_t st, const GatherJob* jobs_dev, uint32_t njobs, uint32_t max_items, const uint64_t* idx_dev, fe* out) {
    if (njobs == 0 || max_items == 0) return SP_OK;
    hipLaunchKernelGGL(gather_jobs_kernel, dim3((max_items + 127) / 128, njobs), dim3(128), 0, st, jobs_dev, idx_dev, out);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// dst column v [coset-major, the `len` evaluations this rank holds] = src column v [natural order, whole domain]: local
// natural index l is the global index (l << shard_log) | shard_rank  (exceptional paths only)
__global__ void __launch_bounds__(256) natural_to_coset_major_kernel(const fe* src, uint64_t src_stride, fe* dst, uint64_t len, LdeOrder ord,
                                                                     uint32_t shard_log, uint32_t shard_rank) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    sk_st(dst + (uint64_t)blockIdx.y * len + ord.at(i), sk_ld(src + (uint64_t)blockIdx.y * src_stride + ((i << shard_log) | shard_rank)));
}
int natural_to_coset_major(hipStream_t st, const fe* src, uint64_t src_stride, fe* dst, uint64_t len, uint32_t ncols, LdeOrder order,
                           uint32_t shard_log, uint32_t shard_rank) {
    hipLaunchKernelGGL(natural_to_coset_major_kernel, dim3((unsigned)((len + 255) / 256), ncols), dim3(256), 0, st, src, src_stride, dst, len, order,
                       shard_log, shard_rank);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// ---------------------------------------------------------------------------------------------- shard reassembly
__global__ void __launch_bounds__(256) interleave_shards_kernel(const uint4* gathered, uint4* out, uint64_t n, ShardMap m) {
    const uint32_t lb_loc = m.logb - m.shard_log;
    const uint64_t Nl = n << lb_loc, N = n << m.logb;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // global index
    if (i >= N) return;
    uint32_t c = (uint32_t)i & ((1u << m.logb) - 1u);
    uint64_t q = i >> m.logb;
    uint32_t r = c & ((1u << m.shard_log) - 1u), c_loc = c >> m.shard_log;
    uint64_t src = (uint64_t)r * Nl + (q << lb_loc) + c_loc;
    out[2 * i] = gathered[2 * src];
    out[2 * i + 1] = gathered[2 * src + 1];
}
int interleave_shards(hipStream_t st, const void* gathered, void* out, uint64_t n, ShardMap m) {
    uint64_t N = n << m.logb;
    hipLaunchKernelGGL(interleave_shards_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, (const uint4*)gathered, (uint4*)out, n, m);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// ---------------------------------------------------------------------------------------------- program AIRs
// The constraint program runs once per point with its values in a per-thread array (scratch): these AIRs are the small
// examples of the reference (a handful of ops), throughput is not the point here, identical field elements are.
template <bool CHECK>
__global__ void __launch_bounds__(256) air_composition_kernel(const fe* __restrict__ cols, uint64_t count, uint64_t col_len, uint32_t stride_log,
                                                              uint32_t logN, uint32_t logb, const fe* __restrict__ roots,
                                                              const CompositionConsts* __restrict__ K, const AirProgram* __restrict__ Pg,
                                                              const fe* __restrict__ ex_roots, const fe* __restrict__ binv,
                                                              fe* __restrict__ out, int* __restrict__ flag, uint32_t shard_log, uint32_t shard_rank) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const uint32_t b = 1u << logb, b_loc = b >> shard_log;
    const ShardMap sm{logb, shard_log, shard_rank};
    const uint64_t e = CHECK ? i : (i << stride_log);   // local natural-order index; storage index = ord.at(.)
    const LdeOrder ord{CHECK ? 0u : 1u, logb - shard_log, logN - logb};
    const uint32_t iglob = CHECK ? 0u : shard_global_index((uint32_t)e, sm);
    const uint32_t c = iglob & (b - 1);
    const uint32_t T = K->n_transitions, B = K->n_boundary;
    fe v[AIR_MAX_LIVE];
    fe cons[AIR_MAX_TRANSITIONS];
    for (uint32_t k = 0; k < T; ++k) cons[k] = fe_zero();
    const uint32_t n_ops = Pg->n_ops;
    for (uint32_t t = 0; t < n_ops; ++t) {
        const AirOpDev o = Pg->ops[t];
        fe r = fe_zero();
        switch (o.op) {
            case 0: {   // frame row = trace row + offset: LDE index + offset * blowup (frame.rs:40-59), same coset
                const uint64_t row = (e + (uint64_t)Pg->offsets[o.a] * (CHECK ? 1u : b_loc)) & (col_len - 1);
                r = sk_ld(cols + (uint64_t)o.b * col_len + ord.at(row));
                break;
            }
            case 1: r = Pg->consts[o.a]; break;
            case 2: r = v[o.a] + v[o.b]; break;
            case 3: r = v[o.a] - v[o.b]; break;
            case 4: r = v[o.a] * v[o.b]; break;
            default: cons[o.a] = v[o.b]; continue;
        }
        v[o.dst] = r;
    }
    if (CHECK) {
        bool bad = false;
        for (uint32_t k = 0; k < T; ++k)
            if (i + Pg->ex_rows[k] < count && !fe_is_zero(cons[k])) bad = true;   // enforced on rows 0 .. n - 1 - exemptions
        for (uint32_t j = 0; j < B; ++j)
            if (i == K->bstep[j] && !fe_eq(sk_ld(cols + (uint64_t)K->bcol[j] * col_len + i), K->bvalue[j])) bad = true;
        if (bad) atomicOr(flag, 1);
        return;
    }
    const fe x = root_pow(roots, iglob, logN) * K->h;
    fe exv[AIR_MAX_EXEMPT_KINDS];
    for (int q = 0; q < AIR_MAX_EXEMPT_KINDS; ++q) {
        fe p = fe_one();
        for (uint32_t j = 0; j < Pg->ex_count[q]; ++j) p = p * (x - sk_ld(ex_roots + j));
        exv[q] = p;
    }
    fe acc = fe_zero();
    for (uint32_t k = 0; k < T; ++k) {
        fe term = K->coef[c][k] * cons[k];
        const uint32_t ek = Pg->ex_kind[k];
        if (ek) term = term * exv[ek - 1];
        acc = acc + term;
    }
    fe total = K->zerofier[c] * acc;
    for (uint32_t j = 0; j < B; ++j) {
        fe num = sk_ld(cols + (uint64_t)K->bcol[j] * col_len + ord.at(e)) - K->bvalue[j];
        total = total + K->coef[c][T + j] * num * sk_ld(binv + (uint64_t)K->bden[j] * count + i);
    }
    sk_st(out + i, total);
}

int air_composition(hipStream_t st, const fe* lde, uint64_t count, uint64_t col_len, uint32_t stride_log, uint32_t logN, uint32_t logb,
                    const fe* roots_N, const CompositionConsts* consts_dev, const AirProgram* prog_dev, const fe* ex_roots,
                    const fe* binv, fe* out, uint32_t shard_log, uint32_t shard_rank) {
    if ((1u << logb) > CAIRO_MAX_BLOWUP) { sp_set_error("composition: blowup factor > 128 unsupported"); return SP_E_UNSUPPORTED; }
    hipLaunchKernelGGL(air_composition_kernel<false>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, lde, count, col_len, stride_log,
                       logN, logb, roots_N, consts_dev, prog_dev, ex_roots, binv, out, (int*)nullptr, shard_log, shard_rank);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}
int air_trace_check(hipStream_t st, const fe* trace, uint64_t n, const CompositionConsts* consts_dev, const AirProgram* prog_dev, int* flag_dev) {
    hipLaunchKernelGGL(air_composition_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, trace, n, n, 0u, 0u, 0u,
                       (const fe*)nullptr, consts_dev, prog_dev, (const fe*)nullptr, (const fe*)nullptr, (fe*)nullptr, flag_dev, 0u, 0u);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

}  // namespace sp

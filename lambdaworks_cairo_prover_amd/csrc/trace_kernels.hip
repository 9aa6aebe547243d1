// See trace_kernels.h.  One lane per trace row; a wave writes 64 consecutive rows of one column = 2 KiB contiguous.
#include "trace_kernels.h"
#include "field_kernels.h"

namespace sp {

namespace {

__device__ __forceinline__ fe tk_ld(const fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void tk_st(fe* p, const fe& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
__device__ __forceinline__ uint64_t tk_low64(const fe& mont) {
    const fe r = fe_from_mont(mont);
    return (uint64_t)r.v[0] | ((uint64_t)r.v[1] << 32);
}
__device__ __forceinline__ fe tk_cell(const MainTraceArgs& a, uint64_t addr, int* flag) {
    if (addr >= a.cells) { atomicExch(flag, 2); return fe_zero(); }
    return tk_ld(a.mem + addr);
}

constexpr int TK_THREADS = 256;

// build_cairo_execution_trace (execution_trace.rs:261-356): decode (instruction_flags.rs:1-77, instruction_offsets.rs:18-56), the
// three operand addresses, res (compute_res, :382-440), update_values (:572-592), t0 / t1 / mul, the selector column.
// jnz rows with dst != 0 need dst^-1: the denominator goes to inv[i] (1 elsewhere), the row is marked, jnz_fix_kernel finishes it.
__global__ void __launch_bounds__(TK_THREADS) step_rows_kernel(MainTraceArgs a, fe* inv, uint8_t* deferred_mask, int* flag, uint64_t s0, uint64_t s1) {
    const uint64_t i = s0 + (uint64_t)blockIdx.x * TK_THREADS + threadIdx.x;
    if (i >= s1) return;
    const uint64_t ap = a.regs[3 * i], fp = a.regs[3 * i + 1], pc = a.regs[3 * i + 2];
    const fe inst = tk_cell(a, pc, flag);
    const uint64_t w = tk_low64(inst);
    const uint32_t off_dst = (uint32_t)(w & 0xffff), off_op0 = (uint32_t)((w >> 16) & 0xffff), off_op1 = (uint32_t)((w >> 32) & 0xffff);
    const uint32_t f = (uint32_t)(w >> 48) & 0x7fff;
    const uint32_t dst_reg = f & 1, op0_reg = (f >> 1) & 1, op1_src = (f >> 2) & 7, res_logic = (f >> 5) & 3, pc_update = (f >> 7) & 7, opcode = (f >> 12) & 7;
    const uint64_t dst_addr = (dst_reg ? fp : ap) + off_dst - 0x8000ull;
    const uint64_t op0_addr = (op0_reg ? fp : ap) + off_op0 - 0x8000ull;
    fe dst = tk_cell(a, dst_addr, flag);
    fe op0 = tk_cell(a, op0_addr, flag);
    const uint64_t op1_base = op1_src == 0 ? tk_low64(op0) : op1_src == 1 ? pc : op1_src == 2 ? fp : ap;
    const uint64_t op1_addr = op1_base + off_op1 - 0x8000ull;
    const fe op1 = tk_cell(a, op1_addr, flag);
    const fe zero = fe_zero(), one = fe_one();
    fe* T = a.trace + i;
    const uint64_t n = a.n;
#pragma unroll
    for (int k = 0; k < 15; ++k) tk_st(T + (uint64_t)k * n, ((f >> k) & 1) ? one : zero);
    tk_st(T + 15 * n, zero);
    fe res;
    bool deferred = false;
    if (pc_update == 4) {
        res = dst;
        deferred = !fe_is_zero(dst);
    } else {
        res = res_logic == 0 ? op1 : res_logic == 1 ? fe_add(op0, op1) : fe_mul(op0, op1);
    }
    tk_st(inv + i, deferred ? dst : one);
    deferred_mask[i] = deferred ? 1 : 0;
    if (opcode == 1) { op0 = fe_from_u64(pc + (op1_src == 1 ? 2 : 1)); dst = fe_from_u64(fp); }
    else if (opcode == 4) res = dst;
    tk_st(T + 16 * n, res);
    tk_st(T + 17 * n, fe_from_u64(ap)); tk_st(T + 18 * n, fe_from_u64(fp)); tk_st(T + 19 * n, fe_from_u64(pc));
    tk_st(T + 20 * n, fe_from_u64(dst_addr)); tk_st(T + 21 * n, fe_from_u64(op0_addr)); tk_st(T + 22 * n, fe_from_u64(op1_addr));
    tk_st(T + 23 * n, inst); tk_st(T + 24 * n, dst); tk_st(T + 25 * n, op0); tk_st(T + 26 * n, op1);
    tk_st(T + 27 * n, fe_from_u64(off_dst)); tk_st(T + 28 * n, fe_from_u64(off_op0)); tk_st(T + 29 * n, fe_from_u64(off_op1));
    const fe t0 = ((f >> 9) & 1) ? dst : zero;
    tk_st(T + 30 * n, t0);
    tk_st(T + 31 * n, deferred ? zero : fe_mul(t0, res));
    tk_st(T + 32 * n, fe_mul(op0, op1));
    tk_st(T + 33 * n, (i + 1 == a.steps) ? zero : one);
    for (uint32_t c = 34; c < a.cols; ++c) tk_st(T + (uint64_t)c * n, zero);
}

__global__ void __launch_bounds__(TK_THREADS) jnz_fix_kernel(MainTraceArgs a, const fe* inv, const uint8_t* deferred_mask) {
    const uint64_t i = (uint64_t)blockIdx.x * TK_THREADS + threadIdx.x;
    if (i >= a.steps || !deferred_mask[i]) return;
    const fe r = tk_ld(inv + i);
    tk_st(a.trace + 16 * a.n + i, r);
    tk_st(a.trace + 31 * a.n + i, fe_mul(tk_ld(a.trace + 30 * a.n + i), r));
}

// add_rc_builtin_columns (execution_trace.rs:358-379, :604-624): the eight 16-bit limbs of the value, least significant first, and the value
__global__ void __launch_bounds__(TK_THREADS) rc_builtin_kernel(MainTraceArgs a, int* flag) {
    const uint64_t k = (uint64_t)blockIdx.x * TK_THREADS + threadIdx.x;
    if (k >= a.rc_count) return;
    const fe v = tk_cell(a, a.rc_start + k, flag);
    const fe raw = fe_from_mont(v);
#pragma unroll
    for (int c = 0; c < 8; ++c) tk_st(a.trace + (uint64_t)(34 + c) * a.n + k, fe_from_u64((raw.v[c / 2] >> (16 * (c & 1))) & 0xffff));
    tk_st(a.trace + 42ull * a.n + k, v);
}

// The rows behind the steps: fill_rc_holes, fill_memory_holes, add_pub_memory_dummy_accesses + pad_with_last_row.  Every row is a
// function of row A = r_holes - 1 (the last range-check row if there is one - zeros and three offsets - else the last step's row,
// which the kernels before this one have completed), so no row of this launch reads another one.
__global__ void __launch_bounds__(TK_THREADS) tail_rows_kernel(MainTraceArgs a) {
    const uint64_t r = a.steps + (uint64_t)blockIdx.x * TK_THREADS + threadIdx.x;
    if (r >= a.n) return;
    const fe zero = fe_zero();
    const bool rc_rows = a.r_holes > a.r_rc;
    for (uint32_t c = 0; c < a.cols; ++c) {
        fe v;
        if (r < a.r_holes) {                                        // a range-check row
            v = (c >= 27 && c <= 29) ? fe_from_u64(a.missing[3 * (r - a.r_rc) + (c - 27)]) : zero;
        } else {
            // row A, column c
            if (rc_rows) v = (c >= 27 && c <= 29) ? fe_from_u64(a.missing[3 * (a.r_holes - 1 - a.r_rc) + (c - 27)]) : zero;
            else v = tk_ld(a.trace + (uint64_t)c * a.n + (a.steps - 1));
            // a memory-hole row: A with four unused addresses; beyond them: row r_dummy - 1 with the memory columns zeroed
            const uint64_t hr = r < a.r_dummy ? r : (a.r_dummy > a.r_holes ? a.r_dummy - 1 : ~0ull);
            if (hr != ~0ull && c >= 19 && c <= 22) {
                const uint64_t q = 4 * (hr - a.r_holes) + (c - 19);
                if (q < a.n_holes) v = fe_from_u64(a.holes[q]);
            }
            if (r >= a.r_dummy && c >= 19 && c <= 26) v = zero;
        }
        tk_st(a.trace + (uint64_t)c * a.n + r, v);
    }
}

}  // namespace

static bool tk_args_ok(const MainTraceArgs& a, const void* scratch, const int* flag_dev) {
    return a.regs && a.mem && a.trace && scratch && flag_dev && a.steps != 0 && a.steps <= a.n && (a.cols == 34 || a.cols == 43);
}
int cairo_main_trace_steps(hipStream_t st, const MainTraceArgs& a, void* scratch, int* flag_dev, uint64_t s0, uint64_t s1) {
    if (!tk_args_ok(a, scratch, flag_dev) || s0 > s1 || s1 > a.steps) return SP_E_INVALID_ARG;
    if (s0 == s1) return SP_OK;
    fe* inv = static_cast<fe*>(scratch);
    uint8_t* mask = reinterpret_cast<uint8_t*>(inv + 2 * a.steps);
    hipLaunchKernelGGL(step_rows_kernel, dim3((uint32_t)((s1 - s0 + TK_THREADS - 1) / TK_THREADS)), dim3(TK_THREADS), 0, st, a, inv, mask, flag_dev, s0, s1);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}
int cairo_main_trace_finish(hipStream_t st, const MainTraceArgs& a, void* scratch, int* flag_dev) {
    if (!tk_args_ok(a, scratch, flag_dev)) return SP_E_INVALID_ARG;
    fe* inv = static_cast<fe*>(scratch);
    fe* inv_scratch = inv + a.steps;
    uint8_t* mask = reinterpret_cast<uint8_t*>(inv_scratch + a.steps);
    const uint32_t blocks = (uint32_t)((a.steps + TK_THREADS - 1) / TK_THREADS);
    SP_TRY(batch_inverse(st, inv, inv_scratch, a.steps, flag_dev));        // (no element is zero: rows without a jnz carry a one)
    hipLaunchKernelGGL(jnz_fix_kernel, dim3(blocks), dim3(TK_THREADS), 0, st, a, inv, mask);
    if (a.rc_count) hipLaunchKernelGGL(rc_builtin_kernel, dim3((uint32_t)((a.rc_count + TK_THREADS - 1) / TK_THREADS)), dim3(TK_THREADS), 0, st, a, flag_dev);
    if (a.n > a.steps) hipLaunchKernelGGL(tail_rows_kernel, dim3((uint32_t)((a.n - a.steps + TK_THREADS - 1) / TK_THREADS)), dim3(TK_THREADS), 0, st, a);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}
int cairo_main_trace_device(hipStream_t st, const MainTraceArgs& a, void* scratch, int* flag_dev) {
    SP_TRY(cairo_main_trace_steps(st, a, scratch, flag_dev, 0, a.steps));
    return cairo_main_trace_finish(st, a, scratch, flag_dev);
}

}  // namespace sp

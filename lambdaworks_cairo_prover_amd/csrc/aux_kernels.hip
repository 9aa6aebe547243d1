// Cairo auxiliary trace on gfx950 (see aux_kernels.h).
#include "aux_kernels.h"
#include "field_kernels.h"
#include <cstring>
#include "sort_kernels.h"

namespace sp {

__device__ __forceinline__ fe ax_ld(const fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void ax_st(fe* p, const fe& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

struct AuxConsts { fe z, alpha, zrc; };

// ---- memory part -------------------------------------------------------------------------------------------
// keys / sorted pairs first (they need no challenge: cairo_aux_presort), numerators and denominators once alpha and z are known
__global__ void __launch_bounds__(256) aux_keys_kernel(const fe* mem_cols, uint64_t n, const fe* pm_addr, const fe* pm_val, uint64_t pm,
                                                       fe* a_aux, fe* v_aux, uint64_t* keys, uint32_t* idx, int* flag, uint32_t key_bits, int* wide_flag, int all_limbs) {
    uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= 4 * n) return;
    uint64_t i = e >> 2; uint32_t k = (uint32_t)e & 3;
    fe a, v;
    if (e >= 4 * n - pm) {  // last |pm| accesses are replaced by the public memory (air.rs:475-494)
        uint64_t j = e - (4 * n - pm);
        a = ax_ld(pm_addr + j); v = ax_ld(pm_val + j);
    } else {
        a = ax_ld(mem_cols + (uint64_t)k * n + i);
        v = ax_ld(mem_cols + (uint64_t)(4 + k) * n + i);
    }
    ax_st(a_aux + e, a); ax_st(v_aux + e, v);
    fe raw = fe_from_mont(a);
    // an address beyond 2^64 (no Cairo VM produces one; the reference sorts whatever the table holds by its 256-bit value,
    // air.rs:519-523): flag 2 sends the caller to the four-limb sort below, which passes all_limbs = 1
    if (!all_limbs && (raw.v[2] | raw.v[3] | raw.v[4] | raw.v[5] | raw.v[6] | raw.v[7])) atomicExch(flag, 2);
    const uint64_t key = (uint64_t)raw.v[0] | ((uint64_t)raw.v[1] << 32);
    if (key_bits < 64 && (key >> key_bits)) atomicExch(wide_flag, 1);   // beyond the bits the presort looks at: the caller sorts again, all 64
    keys[e] = key;
    idx[e] = (uint32_t)e;
}
// limb `limb` (64 bits) of the addresses in the order the sort has reached so far: the next key of the four-limb LSD sort
__global__ void __launch_bounds__(256) aux_limb_keys_kernel(const fe* a_aux, const uint32_t* order, uint64_t M, uint32_t limb, uint64_t* keys, uint32_t* idx) {
    uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M) return;
    const uint32_t s = order[e];
    const fe raw = fe_from_mont(ax_ld(a_aux + s));
    keys[e] = (uint64_t)raw.v[2 * limb] | ((uint64_t)raw.v[2 * limb + 1] << 32);
    idx[e] = s;
}
__global__ void __launch_bounds__(256) aux_gather_pairs_kernel(const fe* a_aux, const fe* v_aux, const uint32_t* idx, uint64_t M, fe* a_s, fe* v_s) {
    uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M) return;
    uint32_t s = idx[e];
    ax_st(a_s + e, ax_ld(a_aux + s)); ax_st(v_s + e, ax_ld(v_aux + s));
}
// num[e] = z - (a + alpha v) of the ORIGINAL access e (air.rs:543-550), den[e] = the same of the sorted pair e
__global__ void __launch_bounds__(256) aux_num_den_kernel(const fe* mem_cols, uint64_t n, const fe* a_s, const fe* v_s, AuxConsts K, fe* num, fe* den) {
    uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= 4 * n) return;
    uint64_t i = e >> 2; uint32_t k = (uint32_t)e & 3;
    const fe a = ax_ld(mem_cols + (uint64_t)k * n + i), v = ax_ld(mem_cols + (uint64_t)(4 + k) * n + i);
    ax_st(num + e, fe_sub(K.z, fe_add(a, fe_mul(K.alpha, v))));
    ax_st(den + e, fe_sub(K.z, fe_add(ax_ld(a_s + e), fe_mul(K.alpha, ax_ld(v_s + e)))));
}

__global__ void __launch_bounds__(256) mul_inplace_kernel(fe* x, const fe* y, uint64_t M) {
    uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M) return;
    ax_st(x + e, fe_mul(ax_ld(x + e), ax_ld(y + e)));
}

// ---- prefix product ----------------------------------------------------------------------------------------
constexpr int PP_PER_THREAD = 8;
constexpr int PP_BLOCK = 256 * PP_PER_THREAD;  // 2048 elements per block

// inclusive scan of the 256 per-thread products of a block through LDS; returns this thread's EXCLUSIVE prefix
__device__ __forceinline__ fe block_exclusive_scan(fe mine, fe* sh, fe* block_total) {
    const uint32_t t = threadIdx.x;
    sh[t] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        fe other = sh[t >= off ? t - off : 0];
        __syncthreads();
        if (t >= off) sh[t] = fe_mul(other, sh[t]);
        __syncthreads();
    }
    fe excl = t == 0 ? fe_one() : sh[t - 1];
    if (block_total) *block_total = sh[255];
    return excl;
}

__global__ void __launch_bounds__(256) pp_block_totals_kernel(const fe* data, uint64_t M, fe* block_tot) {
    __shared__ fe sh[256];
    uint64_t base = (uint64_t)blockIdx.x * PP_BLOCK + (uint64_t)threadIdx.x * PP_PER_THREAD;
    fe acc = fe_one();
    for (int k = 0; k < PP_PER_THREAD; ++k)
        if (base + k < M) acc = fe_mul(acc, ax_ld(data + base + k));
    fe tot;
    (void)block_exclusive_scan(acc, sh, &tot);
    if (threadIdx.x == 0) ax_st(block_tot + blockIdx.x, tot);
}

// single block: in-place inclusive scan of `count` block totals, converted to EXCLUSIVE prefixes
__global__ void __launch_bounds__(256) pp_scan_totals_kernel(fe* block_tot, uint64_t count) {
    __shared__ fe sh[256];
    uint64_t per = (count + 255) / 256;
    uint64_t base = (uint64_t)threadIdx.x * per;
    fe acc = fe_one();
    for (uint64_t k = 0; k < per; ++k)
        if (base + k < count) acc = fe_mul(acc, ax_ld(block_tot + base + k));
    fe run = block_exclusive_scan(acc, sh, nullptr);
    for (uint64_t k = 0; k < per; ++k)
        if (base + k < count) {
            fe cur = ax_ld(block_tot + base + k);
            ax_st(block_tot + base + k, run);  // exclusive prefix of block (base + k)
            run = fe_mul(run, cur);
        }
}

__global__ void __launch_bounds__(256) pp_apply_kernel(fe* data, uint64_t M, const fe* block_prefix) {
    __shared__ fe sh[256];
    uint64_t base = (uint64_t)blockIdx.x * PP_BLOCK + (uint64_t)threadIdx.x * PP_PER_THREAD;
    fe vals[PP_PER_THREAD];
    fe acc = fe_one();
#pragma unroll
    for (int k = 0; k < PP_PER_THREAD; ++k) {
        vals[k] = (base + k < M) ? ax_ld(data + base + k) : fe_one();
        acc = fe_mul(acc, vals[k]);
    }
    fe run = fe_mul(ax_ld(block_prefix + blockIdx.x), block_exclusive_scan(acc, sh, nullptr));
#pragma unroll
    for (int k = 0; k < PP_PER_THREAD; ++k) {
        run = fe_mul(run, vals[k]);
        if (base + k < M) ax_st(data + base + k, run);
    }
}

int prefix_product(hipStream_t st, fe* data, uint64_t M, fe* block_tot) {
    if (M == 0) return SP_OK;
    uint64_t blocks = (M + PP_BLOCK - 1) / PP_BLOCK;
    hipLaunchKernelGGL(pp_block_totals_kernel, dim3((unsigned)blocks), dim3(256), 0, st, data, M, block_tot);
    hipLaunchKernelGGL(pp_scan_totals_kernel, dim3(1), dim3(256), 0, st, block_tot, blocks);
    hipLaunchKernelGGL(pp_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, data, M, block_tot);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// ---- range-check part --------------------------------------------------------------------------------------
// the 3n offsets as 16-bit keys in row-major (long) order; they are then sorted by counting (sort_kernels.hip: a Cairo trace uses
// only a handful of distinct offsets, so a wave adds each class of equal keys to the histogram with one atomic)
__global__ void __launch_bounds__(256) rc_keys_kernel(const fe* off_cols, uint64_t n, uint16_t* keys, int* flag) {
    uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= 3 * n) return;
    uint64_t i = e / 3; uint32_t k = (uint32_t)(e % 3);
    fe raw = fe_from_mont(ax_ld(off_cols + (uint64_t)k * n + i));
    keys[e] = (uint16_t)raw.v[0];     // the low 16 bits whatever the cell holds, like the reference (air.rs:689-692: `representative().into()` to u16)
    (void)flag;
}
__global__ void __launch_bounds__(256) rc_den_kernel(fe* den, AuxConsts K) {
    uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= 65536) return;
    fe raw = fe_zero(); raw.v[0] = v;
    ax_st(den + v, fe_sub(K.zrc, fe_to_mont(raw)));
}
__global__ void __launch_bounds__(256) rc_terms_kernel(const fe* off_cols, uint64_t n, const uint16_t* sorted, const fe* dinv, AuxConsts K, fe* terms) {
    uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= 3 * n) return;
    uint64_t i = e / 3; uint32_t k = (uint32_t)(e % 3);
    fe o = ax_ld(off_cols + (uint64_t)k * n + i);
    ax_st(terms + e, fe_mul(fe_sub(K.zrc, o), ax_ld(dinv + sorted[e])));
}

// ---- wide format (air.rs:705-728), straight into natural-order columns -------------------------------------
// columns 0-2 sorted offsets, 3-6 sorted addresses, 7-10 sorted values: they need the sorts only, no challenge
__global__ void __launch_bounds__(256) aux_sorted_columns_kernel(uint64_t n, const uint16_t* rc_sorted, const fe* a_s, const fe* v_s, fe* out) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (uint32_t k = 0; k < 3; ++k) {
        fe raw = fe_zero(); raw.v[0] = rc_sorted[3 * i + k];
        ax_st(out + (uint64_t)k * n + i, fe_to_mont(raw));
    }
    for (uint32_t k = 0; k < 4; ++k) {
        ax_st(out + (uint64_t)(3 + k) * n + i, ax_ld(a_s + 4 * i + k));
        ax_st(out + (uint64_t)(7 + k) * n + i, ax_ld(v_s + 4 * i + k));
    }
}
// columns 11-14 memory permutation, 15-17 range-check permutation (the two prefix products)
__global__ void __launch_bounds__(256) aux_permutation_columns_kernel(uint64_t n, const fe* perm, const fe* rperm, fe* out) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (uint32_t k = 0; k < 4; ++k) ax_st(out + (uint64_t)(11 + k) * n + i, ax_ld(perm + 4 * i + k));
    for (uint32_t k = 0; k < 3; ++k) ax_st(out + (uint64_t)(15 + k) * n + i, ax_ld(rperm + 3 * i + k));
}

// ---- host side ---------------------------------------------------------------------------------------------
static size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

size_t aux_workspace_bytes(uint64_t n, uint64_t pm_cap, size_t* sort_tmp_bytes) {
    const size_t tmp = radix_sort_workspace_bytes(4 * n);
    *sort_tmp_bytes = tmp;
    size_t b = 0;
    b += 7 * align_up(sizeof(fe) * 4 * n);          // a_aux v_aux num a_s v_s den inv_scratch
    b += align_up(sizeof(fe) * 3 * n);              // rc_terms
    b += align_up(sizeof(uint16_t) * 3 * n);
    b += 2 * align_up(sizeof(uint64_t) * 4 * n) + 2 * align_up(sizeof(uint32_t) * 4 * n);
    b += align_up(sizeof(uint32_t) * 65537);
    b += 2 * align_up(sizeof(fe) * 65536);
    b += align_up(sizeof(fe) * (4 * n / PP_BLOCK + 8));
    b += 2 * align_up(sizeof(fe) * (pm_cap + 1));
    b += 2 * align_up(tmp);
    b += align_up(sizeof(uint16_t) * 3 * n);
    b += align_up(sizeof(fe) * (4 * n / PP_BLOCK + 8));
    return b;
}

void aux_workspace_carve(AuxWorkspace& w, void* base, uint64_t n, uint64_t pm_cap, size_t sort_tmp_bytes) {
    uint8_t* p = reinterpret_cast<uint8_t*>(base);
    auto take = [&](size_t bytes) { void* r = p; p += align_up(bytes); return r; };
    w.a_aux = (fe*)take(sizeof(fe) * 4 * n); w.v_aux = (fe*)take(sizeof(fe) * 4 * n); w.num = (fe*)take(sizeof(fe) * 4 * n);
    w.a_s = (fe*)take(sizeof(fe) * 4 * n); w.v_s = (fe*)take(sizeof(fe) * 4 * n); w.den = (fe*)take(sizeof(fe) * 4 * n);
    w.inv_scratch = (fe*)take(sizeof(fe) * 4 * n);
    w.rc_terms = (fe*)take(sizeof(fe) * 3 * n);
    w.rc_sorted = (uint16_t*)take(sizeof(uint16_t) * 3 * n);
    w.keys_in = (uint64_t*)take(sizeof(uint64_t) * 4 * n); w.keys_out = (uint64_t*)take(sizeof(uint64_t) * 4 * n);
    w.idx_in = (uint32_t*)take(sizeof(uint32_t) * 4 * n); w.idx_out = (uint32_t*)take(sizeof(uint32_t) * 4 * n);
    w.hist = (uint32_t*)take(sizeof(uint32_t) * 65537);
    w.rc_den = (fe*)take(sizeof(fe) * 65536); w.rc_den_scratch = (fe*)take(sizeof(fe) * 65536);
    w.block_tot = (fe*)take(sizeof(fe) * (4 * n / PP_BLOCK + 8));
    w.pm_addr = (fe*)take(sizeof(fe) * (pm_cap + 1)); w.pm_val = (fe*)take(sizeof(fe) * (pm_cap + 1));
    w.sort_tmp = take(sort_tmp_bytes); w.sort_tmp_bytes = sort_tmp_bytes;
    w.sort_tmp_rc = take(sort_tmp_bytes);
    w.rc_keys = (uint16_t*)take(sizeof(uint16_t) * 3 * n);
    w.block_tot_rc = (fe*)take(sizeof(fe) * (4 * n / PP_BLOCK + 8));
    w.n = n; w.pm_cap = pm_cap;
}

int cairo_aux_presort(hipStream_t st, AuxWorkspace& w, const fe* mem_cols, uint64_t n, const fe* pm_addr_host, const fe* pm_val_host,
                      uint64_t pm, int* flag, int* wide_flag) {
    if (n != w.n || pm > w.pm_cap || pm > 4 * n) { sp_set_error("aux trace: workspace too small"); return SP_E_INVALID_ARG; }
    const uint64_t M = 4 * n, M3 = 3 * n;
    auto blocks = [](uint64_t x) { return dim3((unsigned)((x + 255) / 256)); };
    if (pm) {
        SP_HIP_CHECK(hipMemcpyAsync(w.pm_addr, pm_addr_host, sizeof(fe) * pm, hipMemcpyHostToDevice, st));
        SP_HIP_CHECK(hipMemcpyAsync(w.pm_val, pm_val_host, sizeof(fe) * pm, hipMemcpyHostToDevice, st));
    }
    // The memory of a valid run is continuous (every address up to the largest is accessed, execution_trace.rs:195-255 fills the
    // holes), so its 4n addresses are below 4n + |public memory|: the sort looks at log2(8n) key bits - three 8-bit digit passes
    // at 2^20 rows instead of eight - and a key beyond them raises *wide_flag, on which the caller falls back to all 64.
    uint32_t key_bits = 4;
    while ((1ULL << key_bits) < 8 * n && key_bits < 64) ++key_bits;
    hipLaunchKernelGGL(aux_keys_kernel, blocks(M), dim3(256), 0, st, mem_cols, n, w.pm_addr, w.pm_val, pm, w.a_aux, w.v_aux, w.keys_in, w.idx_in, flag,
                       key_bits, wide_flag, 0);
    SP_TRY(radix_sort_pairs_u64(st, w.keys_in, w.keys_out, w.idx_in, w.idx_out, M, key_bits, w.sort_tmp));
    hipLaunchKernelGGL(aux_gather_pairs_kernel, blocks(M), dim3(256), 0, st, w.a_aux, w.v_aux, w.idx_out, M, w.a_s, w.v_s);
    hipLaunchKernelGGL(rc_keys_kernel, blocks(M3), dim3(256), 0, st, mem_cols + 8 * n, n, w.rc_keys, flag);
    SP_TRY(counting_sort_u16(st, w.rc_keys, w.rc_sorted, M3, w.hist));
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

static dim3 ax_blocks(uint64_t x) { return dim3((unsigned)((x + 255) / 256)); }

int cairo_aux_sorted_columns(hipStream_t st, AuxWorkspace& w, uint64_t n, fe* out) {
    hipLaunchKernelGGL(aux_sorted_columns_kernel, ax_blocks(n), dim3(256), 0, st, n, w.rc_sorted, w.a_s, w.v_s, out);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}
int cairo_aux_memory_permutation(hipStream_t st, AuxWorkspace& w, const fe* mem_cols, uint64_t n, const fe rap[3], int* flag) {
    AuxConsts K; K.alpha = rap[0]; K.z = rap[1]; K.zrc = rap[2];
    const uint64_t M = 4 * n;
    hipLaunchKernelGGL(aux_num_den_kernel, ax_blocks(M), dim3(256), 0, st, mem_cols, n, w.a_s, w.v_s, K, w.num, w.den);
    SP_TRY(batch_inverse(st, w.den, w.inv_scratch, M, flag));
    hipLaunchKernelGGL(mul_inplace_kernel, ax_blocks(M), dim3(256), 0, st, w.num, w.den, M);
    SP_TRY(prefix_product(st, w.num, M, w.block_tot));
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}
int cairo_aux_rc_permutation(hipStream_t st, AuxWorkspace& w, const fe* mem_cols, uint64_t n, const fe rap[3], int* flag) {
    AuxConsts K; K.alpha = rap[0]; K.z = rap[1]; K.zrc = rap[2];
    const uint64_t M3 = 3 * n;
    const fe* off_cols = mem_cols + 8 * n;
    hipLaunchKernelGGL(rc_den_kernel, dim3(256), dim3(256), 0, st, w.rc_den, K);
    SP_TRY(batch_inverse(st, w.rc_den, w.rc_den_scratch, 65536, flag));
    hipLaunchKernelGGL(rc_terms_kernel, ax_blocks(M3), dim3(256), 0, st, off_cols, n, w.rc_sorted, w.rc_den, K, w.rc_terms);
    SP_TRY(prefix_product(st, w.rc_terms, M3, w.block_tot_rc));
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}
int cairo_aux_permutation_columns(hipStream_t st, AuxWorkspace& w, uint64_t n, fe* out) {
    hipLaunchKernelGGL(aux_permutation_columns_kernel, ax_blocks(n), dim3(256), 0, st, n, w.num, w.rc_terms, out);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

int cairo_aux_trace_device(hipStream_t st, AuxWorkspace& w, const fe* mem_cols, uint64_t n, const fe* pm_addr_host, const fe* pm_val_host,
                           uint64_t pm, const fe rap[3], fe* out, int* flag, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join, bool presorted, bool all_limbs) {
    if (n != w.n || pm > w.pm_cap || pm > 4 * n) { sp_set_error("aux trace: workspace too small"); return SP_E_INVALID_ARG; }
    AuxConsts K; K.alpha = rap[0]; K.z = rap[1]; K.zrc = rap[2];
    const uint64_t M = 4 * n, M3 = 3 * n;
    const bool fork = side && ev_fork && ev_join;
    hipStream_t rs = fork ? side : st;            // stream of the range-check half
    if (!presorted) {
        if (pm) {
            SP_HIP_CHECK(hipMemcpyAsync(w.pm_addr, pm_addr_host, sizeof(fe) * pm, hipMemcpyHostToDevice, st));
            SP_HIP_CHECK(hipMemcpyAsync(w.pm_val, pm_val_host, sizeof(fe) * pm, hipMemcpyHostToDevice, st));
        }
        // memory: substitute, sort (stable, by address); range check: counting sort of the 3n 16-bit offsets
        int* no_wide = nullptr;
        hipLaunchKernelGGL(aux_keys_kernel, ax_blocks(M), dim3(256), 0, st, mem_cols, n, w.pm_addr, w.pm_val, pm, w.a_aux, w.v_aux, w.keys_in, w.idx_in, flag, 64u, no_wide,
                           all_limbs ? 1 : 0);
        SP_TRY(radix_sort_pairs_u64(st, w.keys_in, w.keys_out, w.idx_in, w.idx_out, M, 64, w.sort_tmp));
        // 256-bit addresses: three more stable sorts, by the next limb each, of the order reached so far (LSD over four 64-bit digits
        // = the reference's stable sort by `representative()`)
        for (uint32_t limb = 1; all_limbs && limb < 4; ++limb) {
            hipLaunchKernelGGL(aux_limb_keys_kernel, ax_blocks(M), dim3(256), 0, st, w.a_aux, w.idx_out, M, limb, w.keys_in, w.idx_in);
            SP_TRY(radix_sort_pairs_u64(st, w.keys_in, w.keys_out, w.idx_in, w.idx_out, M, 64, w.sort_tmp));
        }
        hipLaunchKernelGGL(aux_gather_pairs_kernel, ax_blocks(M), dim3(256), 0, st, w.a_aux, w.v_aux, w.idx_out, M, w.a_s, w.v_s);
        hipLaunchKernelGGL(rc_keys_kernel, ax_blocks(M3), dim3(256), 0, st, mem_cols + 8 * n, n, w.rc_keys, flag);
        SP_TRY(counting_sort_u16(st, w.rc_keys, w.rc_sorted, M3, w.hist));
    }
    if (fork) {
        SP_HIP_CHECK(hipEventRecord(ev_fork, st));   // the trace columns, the sorts and the cleared flag are behind this point
        SP_HIP_CHECK(hipStreamWaitEvent(side, ev_fork, 0));
    }
    // the two permutation arguments: two chains of dependent latencies (a batch inversion and a scan each) side by side
    SP_TRY(cairo_aux_memory_permutation(st, w, mem_cols, n, rap, flag));
    SP_TRY(cairo_aux_rc_permutation(rs, w, mem_cols, n, rap, flag));
    if (fork) {
        SP_HIP_CHECK(hipEventRecord(ev_join, side));
        SP_HIP_CHECK(hipStreamWaitEvent(st, ev_join, 0));
    }
    SP_TRY(cairo_aux_sorted_columns(st, w, n, out));
    SP_TRY(cairo_aux_permutation_columns(st, w, n, out));
    (void)K;
    return SP_OK;
}

}  // namespace sp

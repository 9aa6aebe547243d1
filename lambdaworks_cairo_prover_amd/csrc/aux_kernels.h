// Cairo auxiliary (RAP) trace on the device. Replaces CairoAIR::build_auxiliary_trace (reference src/cairo/air.rs:660-729)
// and its helpers: add_pub_memory_in_public_input_section (:475-494), sort_columns_by_memory_address (:519-523),
// generate_memory_permutation_argument_column (:525-551), generate_range_check_permutation_argument_column (:552-572).
#pragma once
#include "common.h"

namespace sp {

struct AuxWorkspace {
    // 4n-element arrays
    fe *a_aux, *v_aux, *num, *a_s, *v_s, *den, *inv_scratch;
    // 3n-element arrays (range check)
    fe *rc_terms;
    uint16_t* rc_sorted;      // 3n
    uint64_t *keys_in, *keys_out;  // 4n
    uint32_t *idx_in, *idx_out;    // 4n
    uint32_t* hist;           // 65537
    fe *rc_den, *rc_den_scratch;   // 65536 each
    fe* block_tot;            // prefix-product block totals (>= 4n / 2048 + 1, two levels)
    fe *pm_addr, *pm_val;     // public memory (capacity pm_cap)
    void* sort_tmp; size_t sort_tmp_bytes;
    // the range-check half runs on its own stream beside the memory half: its own keys, sort workspace and block totals
    uint16_t* rc_keys;        // 3n
    void* sort_tmp_rc;
    fe* block_tot_rc;
    uint64_t n; uint64_t pm_cap;
};

size_t aux_workspace_bytes(uint64_t n, uint64_t pm_cap, size_t* sort_tmp_bytes);
// carve the workspace out of one allocation of aux_workspace_bytes(n, pm_cap)
void aux_workspace_carve(AuxWorkspace& w, void* base, uint64_t n, uint64_t pm_cap, size_t sort_tmp_bytes);

// mem_cols: the 11 natural-order main-trace columns 19..29 (pc, dst_addr, op0_addr, op1_addr, inst, dst, op0, op1,
// off_dst, off_op0, off_op1), column k at mem_cols + k*n.  pm_addr_host/pm_val_host: the public-memory (address, value)
// list in the order of get_pub_memory_addrs (air.rs:500-517).  rap = alpha_memory, z_memory, z_range_check.
// aux_cols_out: 18 natural-order columns at stride n (sorted offsets 0-2, sorted addresses 3-6, sorted values 7-10,
// memory permutation 11-14, range-check permutation 15-17).  *flag_dev: 1 = a zero permutation denominator (the reference's batch
// inversion fails there too), 2 = an address beyond 2^64 met by the 64-bit sort - the caller repeats the call with all_limbs = true
// (four stable 64-bit sorts, least significant limb first: the reference's stable sort by the 256-bit value, air.rs:519-523).
// Offsets enter the range-check sort as their low 16 bits whatever the cell holds (air.rs:689-692).
// side / ev_fork / ev_join: when side is not null the range-check half (sort of the offsets, its table of inverses, its prefix
// product) is queued there - two chains of dependent latencies (a batch inversion each) side by side instead of in a row.
int cairo_aux_trace_device(hipStream_t st, AuxWorkspace& w, const fe* mem_cols, uint64_t n, const fe* pm_addr_host, const fe* pm_val_host,
                           uint64_t pm, const fe rap[3], fe* aux_cols_out, int* flag_dev, hipStream_t side = nullptr,
                           hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr, bool presorted = false, bool all_limbs = false);

// The part of the auxiliary trace that needs no challenge, for a stream of its own while round 1 extends and hashes the main
// trace: public-memory substitution, the stable sort of the 4n accesses by address with the gather of the sorted (address,
// value) pairs, and the sort of the 3n offsets.  A later cairo_aux_trace_device(..., presorted = true) on the same workspace
// starts from there.  *flag_dev is set on malformed input like there.
// The address sort looks at log2(8n) key bits only (the memory of a valid run is continuous); *wide_flag_dev is set when an
// address lies beyond them - the presort's order is then not the stable order by address and the caller must not use it.
int cairo_aux_presort(hipStream_t st, AuxWorkspace& w, const fe* mem_cols, uint64_t n, const fe* pm_addr_host, const fe* pm_val_host,
                      uint64_t pm, int* flag_dev, int* wide_flag_dev);

// The pieces of cairo_aux_trace_device after the sorts, for callers that run them on streams of their own (prover.cpp: the sorted
// columns and their transforms on the compute stream beside the two permutation chains):
//   sorted columns 0-10 of the auxiliary trace (no challenge needed);
//   memory permutation -> w.num (4n prefix products), range-check permutation -> w.rc_terms (3n);  *flag_dev: zero denominator;
//   permutation columns 11-17 from those.
int cairo_aux_sorted_columns(hipStream_t st, AuxWorkspace& w, uint64_t n, fe* aux_cols_out);
int cairo_aux_memory_permutation(hipStream_t st, AuxWorkspace& w, const fe* mem_cols, uint64_t n, const fe rap[3], int* flag_dev);
int cairo_aux_rc_permutation(hipStream_t st, AuxWorkspace& w, const fe* mem_cols, uint64_t n, const fe rap[3], int* flag_dev);
int cairo_aux_permutation_columns(hipStream_t st, AuxWorkspace& w, uint64_t n, fe* aux_cols_out);

// In-place inclusive prefix product of M elements (block_tot: workspace of >= M/2048 + 2 elements).
int prefix_product(hipStream_t st, fe* data, uint64_t M, fe* block_tot);

}  // namespace sp

// Sorting primitives of the Cairo auxiliary trace (reference src/cairo/air.rs:519-523 `sort_columns_by_memory_address`, :552-572):
// a stable least-significant-digit radix sort of (64-bit key, 32-bit index) pairs and a counting sort of 16-bit keys, written for
// gfx950 (one wave per work-group, stable ranking by wave-wide digit matching).  They replace rocPRIM's device radix sort, which
// was the one vendor-library primitive on the proof path.
#pragma once
#include "common.h"

namespace sp {

// bytes of workspace radix_sort_pairs_u64 needs for n pairs
size_t radix_sort_workspace_bytes(uint64_t n);
// Stable sort of n pairs by the low key_bits bits of the key (8-bit digits, ceil(key_bits / 8) passes).  The sorted pairs end up
// in keys_out / vals_out; keys_in / vals_in are overwritten.  n < 2^32.
int radix_sort_pairs_u64(hipStream_t st, uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, uint64_t n, uint32_t key_bits,
                         void* workspace);
// out = the n 16-bit keys in ascending order (counting sort); hist: 65537 x uint32 of workspace.  n < 2^32.
int counting_sort_u16(hipStream_t st, const uint16_t* keys, uint16_t* out, uint64_t n, uint32_t* hist);

}  // namespace sp

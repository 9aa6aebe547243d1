// Element-wise field kernels (batch inversion, layout/encoding conversion). See field_kernels.hip.
#pragma once
#include "common.h"

namespace sp {

// In-place batch inverse of n device elements; scratch = n elements; *zero_flag_dev is set to 1 when an element is 0.
int batch_inverse(hipStream_t st, fe* data, fe* scratch, uint64_t n, int* zero_flag_dev);
// out[i] = a[i] * b[i] (b == nullptr: a[i]^2), canonical Montgomery values
int mul_elements(hipStream_t st, const fe* a, const fe* b, uint64_t n, fe* out);
// rows_dev: n_rows x n_cols row-major in ABI encoding `enc` (device memory) -> cols[c*col_stride + r] device layout
int rows_to_columns(hipStream_t st, int enc, const uint8_t* rows_dev, uint64_t n_rows, uint32_t n_cols, fe* cols, uint64_t col_stride);
// dst_dev[0 .. bytes) = src_pinned_host[0 .. bytes) by a kernel reading the page-locked host memory (16-byte aligned, bytes % 16 == 0)
int pull_copy(hipStream_t st, const void* src_pinned_host, void* dst_dev, size_t bytes);
int encode_elements(hipStream_t st, int enc, const fe* in, uint64_t n, uint8_t* out_dev);
int decode_elements(hipStream_t st, int enc, const uint8_t* in_dev, uint64_t n, fe* out);
// `cols` columns of n cells each (n a multiple of 64) given as bitmaps of n / 64 words per column -> field elements 0 / 1, column-major
int expand_bit_columns(hipStream_t st, const uint64_t* bits_dev, uint64_t n, uint32_t cols, fe* out);

}  // namespace sp

// Stable LSD radix sort of 64-bit keys on gfx950 (cb_sort.hip): the one-off sorts of the graph ingest (edge_index -> CSR: key = row << bits |
// col, GNN_model/GCN.py:92-95; symmetrisation of utils.py:667-674).  Hand-written like everything else in this library (rounds 1-3 used
// rocPRIM's onesweep here: 60 % of the shared object's size for two call sites).
#pragma once
#include "cb_common.h"

namespace cb {

// bytes of scratch sort_u64 needs for n keys
size_t sort_u64_temp_bytes(int64_t n);

// keys_out = keys_in sorted ascending by bits [0, end_bit) (stable); keys_in is used as the second buffer and destroyed.  n < 2^32.
int sort_u64(void* temp, size_t temp_bytes, uint64_t* keys_in, uint64_t* keys_out, int64_t n, int end_bit, hipStream_t st);

}  // namespace cb

// rm_hitsort_dev.h -- the hit records into the reference's output order on the device.
//
// Same result as rma::sort_hits() (rm_hitsort.h, which says what the order is and why): the five
// header words packed into one 64-bit key of the widths this database and descriptor need, a
// stable radix sort of (key, slot in the hit buffer) pairs -- rocPRIM's, a plain library sort --
// then one kernel that moves the records and renumbers the order word.  The copy back is then
// the final, ordered stream and the host does not touch the records at all (its sort + gather took
// 0.3 ms of a 4.4 ms step at 100 Mbase, 1.9 of 38 ms at 1 Gbase).  A record whose header does not
// fit the key is flagged and the caller falls back to the host sort.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>

namespace rma {

struct DevHitSort {
	unsigned long long	*keys[ 2 ] = { nullptr, nullptr };
	unsigned	*vals[ 2 ] = { nullptr, nullptr };
	void	*tmp = nullptr;
	size_t	tmp_bytes = 0;
	int32_t	*d_out = nullptr;
	int	*d_flag = nullptr;
	int64_t	cap = 0;
	int	stride = 0;
	int	w_ord = 20;	// bits for the order word (pieces of items count in their own ranges, rm_scan_kernel.h PIECE_ORDER_BITS); widened after a scan whose order words did not fit

	// room for cap records of stride words
	hipError_t	reserve( int64_t cap, int stride );
	// Enqueue on s: d_out[ n ][ stride ] = d_hits[ n ][ stride ] in order, order word renumbered;
	// *d_flag != 0 afterwards if some header word did not fit (then d_out is not to be used).
	// Returns hipErrorInvalidValue without enqueuing anything when the widths leave no room for
	// the order word.
	hipError_t	run( const int32_t *d_hits, int64_t n, int w_seq, int w_pos, int w_rank, hipStream_t s );
	void	release();
};

// word 0 of every record (the entry's number within one scan's database) becomes index[ word 0 ], the
// entry's number in the whole database: what a rank of a multi-GPU search does before its records
// travel (rm_gather.cpp).  Records whose word 0 is outside 0 .. n_index-1 are left as they are.
hipError_t	relabel_entries( int32_t *d_hits, int64_t n, int stride, const int32_t *d_index, int32_t n_index, hipStream_t s );

}	// namespace rma

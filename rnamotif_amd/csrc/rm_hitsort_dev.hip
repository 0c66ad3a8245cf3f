// rm_hitsort_dev.hip -- see rm_hitsort_dev.h
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "rm_hitsort_dev.h"

namespace rma {

namespace {

struct Widths { int seq, pos, rank, ord; };

__global__ void __launch_bounds__( 256 )
hit_keys_kernel( const int32_t *hits, long long n, int stride, Widths w, unsigned long long *keys, unsigned *vals, int *flag )
{
	const long long	i = blockIdx.x * 256ll + threadIdx.x;
	if( i >= n )
		return;
	const int32_t	*x = hits + i * stride;
	const unsigned	s = unsigned( x[ 0 ] ), c = unsigned( x[ 1 ] ) & 1u, p = unsigned( x[ 2 ] ), r = unsigned( x[ 3 ] ), o = unsigned( x[ 4 ] );
	// (every width is below 32)
	if( ( s >> w.seq ) | ( p >> w.pos ) | ( r >> w.rank ) | ( o >> w.ord ) )
		atomicOr( flag, 1 );
	unsigned long long	k = ( ( unsigned long long )s << 1 ) | c;
	k = ( k << w.pos ) | p;
	k = ( k << w.rank ) | r;
	k = ( k << w.ord ) | o;
	keys[ i ] = k;
	vals[ i ] = unsigned( i );
}

// one thread per word of the ordered stream; the order word becomes the record's distance from the
// first record of its (entry, strand, start, rank) group, found by bisection in the sorted keys
__global__ void __launch_bounds__( 256 )
hit_gather_kernel( const int32_t *hits, const unsigned *perm, const unsigned long long *skeys, int w_ord,
	long long n, int stride, int32_t *out )
{
	const long long	i = blockIdx.x * 4ll + ( threadIdx.x >> 6 );	// a wave per record
	if( i >= n )
		return;
	const int32_t	*x = hits + ( long long )perm[ i ] * stride;
	int32_t	*o = out + i * stride;
	for( int w = threadIdx.x & 63; w < stride; w += 64 ){
		int32_t	v = x[ w ];
		if( w == 4 ){
			const unsigned long long	g = skeys[ i ] >> w_ord;
			long long	lo = 0, hi = i;
			while( lo < hi ){
				const long long	mid = ( lo + hi ) >> 1;
				if( ( skeys[ mid ] >> w_ord ) < g )
					lo = mid + 1;
				else
					hi = mid;
			}
			v = int32_t( i - lo );
		}
		o[ w ] = v;
	}
}

__global__ void __launch_bounds__( 256 )
hit_relabel_kernel( int32_t *hits, long long n, int stride, const int32_t *index, int n_index )
{
	const long long	i = blockIdx.x * 256ll + threadIdx.x;
	if( i >= n )
		return;
	const int32_t	e = hits[ i * stride ];
	if( e >= 0 && e < n_index )
		hits[ i * stride ] = index[ e ];
}

}	// namespace

hipError_t relabel_entries( int32_t *d_hits, int64_t n, int stride, const int32_t *d_index, int32_t n_index, hipStream_t s )
{
	if( n <= 0 )
		return hipSuccess;
	hipLaunchKernelGGL( hit_relabel_kernel, dim3( unsigned( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, s,
		d_hits, ( long long )n, stride, d_index, int( n_index ) );
	return hipGetLastError();
}

void DevHitSort::release()
{
	for( int i = 0; i < 2; i++ ){
		if( keys[ i ] ) ( void )hipFree( keys[ i ] );
		if( vals[ i ] ) ( void )hipFree( vals[ i ] );
		keys[ i ] = nullptr;
		vals[ i ] = nullptr;
	}
	if( tmp ) ( void )hipFree( tmp );
	if( d_out ) ( void )hipFree( d_out );
	if( d_flag ) ( void )hipFree( d_flag );
	tmp = nullptr;
	d_out = nullptr;
	d_flag = nullptr;
	cap = 0;
	tmp_bytes = 0;
}

hipError_t DevHitSort::reserve( int64_t want, int stride_ )
{
	if( want <= cap && stride_ == stride )
		return hipSuccess;
	const int	keep = w_ord;
	release();
	w_ord = keep;
	hipError_t	e;
	for( int i = 0; i < 2; i++ ){
		if( ( e = hipMalloc( &keys[ i ], size_t( want ) * sizeof( unsigned long long ) ) ) != hipSuccess ) return e;
		if( ( e = hipMalloc( &vals[ i ], size_t( want ) * sizeof( unsigned ) ) ) != hipSuccess ) return e;
	}
	if( ( e = hipMalloc( &d_out, size_t( want ) * stride_ * sizeof( int32_t ) ) ) != hipSuccess ) return e;
	if( ( e = hipMalloc( &d_flag, sizeof( int ) ) ) != hipSuccess ) return e;
	size_t	bytes = 0;
	if( ( e = rocprim::radix_sort_pairs( nullptr, bytes, keys[ 0 ], keys[ 1 ], vals[ 0 ], vals[ 1 ], size_t( want ), 0u, 64u ) ) != hipSuccess ) return e;
	if( bytes == 0 )
		bytes = 16;
	if( ( e = hipMalloc( &tmp, bytes ) ) != hipSuccess ) return e;
	tmp_bytes = bytes;
	cap = want;
	stride = stride_;
	return hipSuccess;
}

hipError_t DevHitSort::run( const int32_t *d_hits, int64_t n, int w_seq, int w_pos, int w_rank, hipStream_t s )
{
	if( n <= 0 || n > cap || n > int64_t( 0x7fffffff ) )
		return hipErrorInvalidValue;
	Widths	w{ w_seq, w_pos, w_rank, w_ord };
	if( w.seq > 31 || w.pos > 31 || w.rank > 31 )
		return hipErrorInvalidValue;
	if( w.seq + 1 + w.pos + w.rank + w.ord > 64 )
		w.ord = 64 - ( w.seq + 1 + w.pos + w.rank );
	if( w.ord > 31 )
		w.ord = 31;
	if( w.ord < 4 )
		return hipErrorInvalidValue;
	hipError_t	e;
	if( ( e = hipMemsetAsync( d_flag, 0, sizeof( int ), s ) ) != hipSuccess ) return e;
	hipLaunchKernelGGL( hit_keys_kernel, dim3( unsigned( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, s,
		d_hits, ( long long )n, stride, w, keys[ 0 ], vals[ 0 ], d_flag );
	if( ( e = hipGetLastError() ) != hipSuccess ) return e;
	// All 64 bits of the key, whatever the fields in use: the library picks its kernels by the bit range,
	// and a range that changes with the database means kernels that are loaded in the middle of a search
	// (11 ms in the first batch, measured) instead of by the scanner's warm-up.  The extra passes over a
	// few thousand keys cost microseconds.
	const unsigned	bits = 64u;
	size_t	bytes = 0;
	if( ( e = rocprim::radix_sort_pairs( nullptr, bytes, keys[ 0 ], keys[ 1 ], vals[ 0 ], vals[ 1 ], size_t( n ), 0u, bits, s ) ) != hipSuccess ) return e;
	if( bytes > tmp_bytes ){
		// (the library picks its algorithm by size: what a small sort needs is not bounded by the largest one's)
		if( ( e = hipStreamSynchronize( s ) ) != hipSuccess ) return e;
		( void )hipFree( tmp );
		tmp = nullptr;
		tmp_bytes = 0;
		if( ( e = hipMalloc( &tmp, bytes ) ) != hipSuccess ) return e;
		tmp_bytes = bytes;
	}
	bytes = tmp_bytes;
	if( ( e = rocprim::radix_sort_pairs( tmp, bytes, keys[ 0 ], keys[ 1 ], vals[ 0 ], vals[ 1 ], size_t( n ), 0u, bits, s ) ) != hipSuccess ) return e;
	hipLaunchKernelGGL( hit_gather_kernel, dim3( unsigned( ( n + 3 ) / 4 ) ), dim3( 256 ), 0, s,
		d_hits, vals[ 1 ], keys[ 1 ], w.ord, ( long long )n, stride, d_out );
	return hipGetLastError();
}

}	// namespace rma

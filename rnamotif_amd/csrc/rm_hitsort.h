// rm_hitsort.h -- hit records into the reference's output order.
//
// The reference prints candidates as its walk meets them: entry by entry, strand by strand, start
// position by start position (find_motif.c:164-215), and at one start the alternatives of the outer
// helix from the longest down (find_motif.c:370-411), each alternative's candidates in the order of
// its own depth-first walk.  The kernels append records to the hit buffer in whatever order their
// lanes finish; the five header words (entry, strand, start, rank of the alternative, order within
// it) are that walk's coordinates.  sort_hits() orders records by them -- ties (the candidates of one
// continuation, whose slots in the buffer ascend) keep their buffer order -- and renumbers the order
// word to 0, 1, ... within (entry, strand, start, rank), which is what the walk would have counted.
//
// A stable least-significant-digit radix sort over the bits the header words really use: the scan
// of 100 Mbase against trna.descr returns 5.8 K records, of 1 Gbase 59 K, and the comparison sort
// that stood here before took a tenth of such a step.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace rma {

struct HitKey { uint64_t k; uint32_t i, order; };

static inline int bits_of( uint32_t x ){ int b = 0; while( x ){ b++; x >>= 1; } return b; }

// stable LSD radix sort of keys[].k over its low `bits` bits
inline void hit_radix( std::vector<HitKey> &keys, std::vector<HitKey> &tmp, int bits )
{
	const size_t	n = keys.size();
	if( bits <= 0 || n < 2 )
		return;
	const int	passes = ( bits + 11 ) / 12;
	const int	digit = ( bits + passes - 1 ) / passes;
	const uint32_t	mask = ( 1u << digit ) - 1;
	tmp.resize( n );
	std::vector<uint32_t>	cnt( size_t( 1 ) << digit );
	HitKey	*src = keys.data(), *dst = tmp.data();
	for( int p = 0, sh = 0; p < passes; p++, sh += digit ){
		std::fill( cnt.begin(), cnt.end(), 0u );
		for( size_t i = 0; i < n; i++ )
			cnt[ ( src[ i ].k >> sh ) & mask ]++;
		uint32_t	at = 0;
		for( uint32_t &c : cnt ){
			const uint32_t	k = c;
			c = at;
			at += k;
		}
		for( size_t i = 0; i < n; i++ )
			dst[ cnt[ ( src[ i ].k >> sh ) & mask ]++ ] = src[ i ];
		std::swap( src, dst );
	}
	if( src != keys.data() )
		memcpy( keys.data(), src, n * sizeof( HitKey ) );
}

// The permutation into the reference's order and the renumbered order word of every record.
// The five header words are packed into one 64-bit key of just the bits this set of records uses
// (7+1+20+6+1 for 100 entries of a megabase against trna.descr: three passes of 12 bits); a set
// whose fields need more than 64 bits together is sorted by comparison.
inline void hit_order( const int32_t *raw, int64_t n, int stride, std::vector<HitKey> &keys, std::vector<HitKey> &tmp )
{
	keys.resize( static_cast<size_t>( n ) );
	// one pass over the records (they are 20 + 8 per element bytes apart): header words aside
	tmp.resize( static_cast<size_t>( n ) * 2 );
	uint32_t	*hdr = reinterpret_cast<uint32_t *>( tmp.data() );	// 5 words per record
	uint32_t	m[ 5 ] = { 0, 0, 0, 0, 0 };
	bool	sorted = true;
	for( int64_t i = 0; i < n; i++ ){
		const int32_t	*x = raw + i * stride;
		uint32_t	*h = hdr + i * 5;
		h[ 0 ] = uint32_t( x[ 0 ] );
		h[ 1 ] = uint32_t( x[ 1 ] ) & 1u;
		h[ 2 ] = uint32_t( x[ 2 ] );
		h[ 3 ] = uint32_t( x[ 3 ] );
		h[ 4 ] = uint32_t( x[ 4 ] );
		m[ 0 ] |= h[ 0 ];
		m[ 2 ] |= h[ 2 ];
		m[ 3 ] |= h[ 3 ];
		m[ 4 ] |= h[ 4 ];
		if( i ){
			const uint32_t	*w = h - 5;
			if( w[ 0 ] != h[ 0 ] ? w[ 0 ] > h[ 0 ] : w[ 1 ] != h[ 1 ] ? w[ 1 ] > h[ 1 ] : w[ 2 ] != h[ 2 ] ? w[ 2 ] > h[ 2 ] :
				w[ 3 ] != h[ 3 ] ? w[ 3 ] > h[ 3 ] : w[ 4 ] > h[ 4 ] )
				sorted = false;
		}
	}
	const int	w0 = bits_of( m[ 0 ] ), w2 = bits_of( m[ 2 ] ), w3 = bits_of( m[ 3 ] ), w4 = bits_of( m[ 4 ] );
	const int	bits = w0 + 1 + w2 + w3 + w4;
	if( bits <= 64 ){
		// (a field of 0 bits is 0 in every record: shifts stay below 64 because bits <= 64 and the
		// strand bit is always there)
		for( int64_t i = 0; i < n; i++ ){
			const uint32_t	*h = hdr + i * 5;
			uint64_t	k = h[ 0 ];
			k = ( k << 1 ) | h[ 1 ];
			k = ( k << w2 ) | h[ 2 ];
			k = ( k << w3 ) | h[ 3 ];
			keys[ i ].k = k;	// without the order word: the group a record is renumbered in
			keys[ i ].i = uint32_t( i );
			keys[ i ].order = h[ 4 ];
		}
		if( !sorted ){
			if( n < 1024 )
				std::stable_sort( keys.begin(), keys.end(), []( const HitKey &x, const HitKey &y ){
					return x.k != y.k ? x.k < y.k : x.order < y.order; } );
			else{
				for( int64_t i = 0; i < n; i++ )
					keys[ i ].k = w4 ? ( keys[ i ].k << w4 ) | keys[ i ].order : keys[ i ].k;
				hit_radix( keys, tmp, bits );
				if( w4 )
					for( int64_t i = 0; i < n; i++ )
						keys[ i ].k >>= w4;
			}
		}
	}else{
		// entry numbers, positions, ranks and order words that together need more than 64 bits
		std::vector<uint32_t>	hd( hdr, hdr + n * 5 );
		for( int64_t i = 0; i < n; i++ )
			keys[ i ].i = uint32_t( i );
		auto	cmp = [ &hd ]( const HitKey &p, const HitKey &q ){
			return std::lexicographical_compare( &hd[ size_t( p.i ) * 5 ], &hd[ size_t( p.i ) * 5 + 5 ],
				&hd[ size_t( q.i ) * 5 ], &hd[ size_t( q.i ) * 5 + 5 ] ); };
		if( !sorted )
			std::stable_sort( keys.begin(), keys.end(), cmp );
		// groups numbered instead of packed
		uint64_t	g = 0;
		for( int64_t i = 0; i < n; i++ ){
			if( i && memcmp( &hd[ size_t( keys[ i ].i ) * 5 ], &hd[ size_t( keys[ i - 1 ].i ) * 5 ], 4 * sizeof( uint32_t ) ) )
				g++;
			keys[ i ].k = g;
		}
	}
	// the order word: 0, 1, ... within (entry, strand, start, rank)
	uint64_t	pk = ~0ull;
	uint32_t	order = 0;
	for( int64_t i = 0; i < n; i++ ){
		order = ( i && keys[ i ].k == pk ) ? order + 1 : 0;
		pk = keys[ i ].k;
		keys[ i ].order = order;
	}
}

// raw[n][stride] -> out[n][stride] in the reference's order, order word renumbered
inline void sort_hits( const int32_t *raw, int64_t n, int stride, int32_t *out,
		std::vector<HitKey> &keys, std::vector<HitKey> &tmp )
{
	hit_order( raw, n, stride, keys, tmp );
	// the records themselves: 20 bytes + 8 per descriptor element each, from all over the buffer --
	// memory bound, so above a few thousand records several threads share the copy
	auto	copy = [&]( int64_t lo, int64_t hi ){
		for( int64_t i = lo; i < hi; i++ ){
			int32_t	*o = out + size_t( i ) * stride;
			memcpy( o, raw + int64_t( keys[ i ].i ) * stride, size_t( stride ) * sizeof( int32_t ) );
			o[ 4 ] = int32_t( keys[ i ].order );
		}
	};
	const int64_t	per = 8192;
	unsigned	nt = unsigned( std::min<int64_t>( n / per, 8 ) );
	nt = std::min( nt, std::max( 1u, std::thread::hardware_concurrency() ) );
	if( nt < 2 ){
		copy( 0, n );
		return;
	}
	std::vector<std::thread>	pool;
	for( unsigned t = 1; t < nt; t++ )
		pool.emplace_back( copy, n * t / nt, n * ( t + 1 ) / nt );
	copy( 0, n / nt );
	for( std::thread &t : pool )
		t.join();
}

}	// namespace rma

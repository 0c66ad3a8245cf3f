// rm_scan_core.h -- the per-start-position motif search as an explicit state
// machine: what find_motif()/find_1_motif()/find_ss()/find_wchlx()/find_pknot*()
// /find_phlx()/find_triplex()/find_4plex*() do by recursion
// (/root/reference/src/find_motif.c:245-973) is done here with one frame per
// search level, so a GPU lane can run it without a call stack.
//
// Invariant used throughout: the reference only ever descends from search k to
// search k+1 (s_forward, s_inner and "searchno + 1" all name it, compile.c:
// 3128-3296), and each level's [zero,dollar] window has been written by an
// earlier level before it is entered.  A level is a resumable generator of
// alternatives; the driver loop below advances the deepest level, descends when
// it yields, and pops when it is exhausted.
//
// The file is plain C++: the HIP kernel includes it with RMD_FN = __device__,
// tests/hostsim compiles it for the CPU to debug the state machine without a
// GPU.  Sequence access is through a byte array of base codes (0..4).
#pragma once
#include "rm_dev_program.h"

#ifndef RMD_FN
#define RMD_FN	static inline
#endif
// rarely taken paths (constraints, multi-strand helices, end-of-list checks): kept
// out of line in the kernel so the common ss / proper-helix path stays small
#ifndef RMD_COLD
#define RMD_COLD	static
#endif

#ifndef RMD_FN_MEMBER
#define RMD_FN_MEMBER	inline
#endif
// the level generators of the general path (big, each with its own working set)
#ifndef RMD_GEN_FN
#define RMD_GEN_FN	RMD_COLD
#endif
#define RMD_UNDEF	(-1)

// host-only instrumentation for tests/hostsim (never defined in the kernel build)
#ifdef RMD_STATS
static long long	rmd_stat[ 16 ];
#define RMD_COUNT( i )	( rmd_stat[ i ]++ )
#else
#define RMD_COUNT( i )	( ( void )0 )
#endif

// What a complete structural match leaves behind (s_matchoff/s_matchlen/s_n_mispairs/
// s_n_mismatches of every element, rnamot.h:235-238, plus the context spans): rebuilt from the
// search records only when a candidate reaches the end of the search list, then run through
// chk_motif / set_context / chk_sites and written out.  The search itself never touches it.
struct rmd_lane_t {
	int32_t	moff[ RMD_MAX_ELEMS ], mlen[ RMD_MAX_ELEMS ];	// s_matchoff/s_matchlen
	int16_t	mpr[ RMD_MAX_ELEMS ], mm[ RMD_MAX_ELEMS ];	// s_n_mispairs/s_n_mismatches
	int32_t	l_off, l_len, r_off, r_len, l_mm, r_mm;
	int32_t	slen, szero;
	int32_t	rank, order;
	// (the end-of-list checks below read the table through these, so that a kernel can keep it
	// spread over the lanes of a wave instead: rm_scan_kernel.h WaveTable)
	RMD_FN_MEMBER int	off( int d ) const { return moff[ d ]; }
	RMD_FN_MEMBER int	len( int d ) const { return mlen[ d ]; }
	RMD_FN_MEMBER int	wtype( const rmd_program_t *P, int pos, int undef_is_ss ) const;
};

// sequence view: code of absolute position p is sq[ p - sq0 ]
struct rmd_seq_t {
	const uint8_t	*sq;
	int32_t	sq0;
};

RMD_FN int rmd_code( const rmd_seq_t &s, int p ) { return s.sq[ p - s.sq0 ]; }

// A window of a strand as 4-bit codes, 8 per dword in forward-strand order, dwords STRIDE apart
// (the pooled pass B of the kernel keeps one column per lane in LDS; the bases come straight from
// the packed database, complemented for the other strand).  Position p of the strand is nibble
// ( p ^ flip ) + bias: flip 0 / bias -first for the forward strand, flip -1 / bias slen - first
// for the reverse one (~p = -p - 1).
template< int STRIDE >
struct rmd_nibseq_t {
	const uint32_t	*w;
	int32_t	flip, bias;
};
template< int STRIDE >
RMD_FN int rmd_code( const rmd_nibseq_t<STRIDE> &s, int p )
{
	const int	i = ( p ^ s.flip ) + s.bias;
	return int( ( s.w[ ( i >> 3 ) * STRIDE ] >> ( ( i & 7 ) * 4 ) ) & 15u );
}

RMD_FN int rmd_paired( const rmd_program_t *P, int ps, int b5, int b3 )		// RM_paired :1291
{
	return ( rmd_pairsets( P )[ ps ].mat2 >> ( b5 * 5 + b3 ) ) & 1;
}
RMD_FN int rmd_triple( const rmd_program_t *P, int ps, int b1, int b2, int b3 )	// RM_triple :1304
{
	int	ix = ( b1 * 5 + b2 ) * 5 + b3;
	return ( rmd_pairsets( P )[ ps ].mat3[ ix >> 5 ] >> ( ix & 31 ) ) & 1;
}
RMD_FN int rmd_quad( const rmd_program_t *P, int ps, int b1, int b2, int b3, int b4 )	// RM_quad :1318
{
	int	ix = ( ( b1 * 5 + b2 ) * 5 + b3 ) * 5 + b4;
	return ( rmd_pairsets( P )[ ps ].mat4[ ix >> 5 ] >> ( ix & 31 ) ) & 1;
}

RMD_FN int rmd_popc64( uint64_t x )
{
#if defined( __HIP_DEVICE_COMPILE__ )
	return __popcll( x );
#else
	return __builtin_popcountll( x );
#endif
}
RMD_FN int rmd_ctz64( uint64_t x )
{
#if defined( __HIP_DEVICE_COMPILE__ )
	return __ffsll( ( unsigned long long )x ) - 1;
#else
	return __builtin_ctzll( x );
#endif
}

RMD_FN int rmd_clz64( uint64_t x )	// x != 0
{
#if defined( __HIP_DEVICE_COMPILE__ )
	return __clzll( ( long long )x );
#else
	return __builtin_clzll( x );
#endif
}

// ---------------------------------------------------------------- seq= constraints
// step() semantics (regexp.c:389-664) as a set-of-positions automaton: after each
// base, `act` holds the pattern positions that just consumed it.
RMD_FN uint64_t rmd_re_close( const rmd_regex_t &re, uint64_t f )
{
	for( int k = 0; k < re.n_close; k++ )
		f |= ( f & re.opt ) << 1;
	return f;
}

template< class SQ >
RMD_COLD int rmd_re_step( const rmd_regex_t &re, const SQ &sq, int off, int len )
{
	uint64_t	act = 0, endbit = 1ull << re.n_states;
	for( int pos = 0; ; pos++ ){
		uint64_t	f = ( act << 1 ) | ( act & re.star );
		if( pos == 0 || !re.anchored )
			f |= 1;
		f = rmd_re_close( re, f );
		if( ( f & endbit ) && ( !re.dollar || pos == len ) )
			return 1;
		if( pos == len )
			return 0;
		act = f & re.accept[ rmd_code( sq, off + pos ) ];
		if( act == 0 && re.anchored )
			return 0;
	}
}

// mm_step()/mm_advance() (mm_regexp.c:353-469): fixed length expressions only.
// *n_mm is left as the last attempt left it.
template< class SQ >
RMD_COLD int rmd_re_mm_step( const rmd_regex_t &re, const SQ &sq, int off, int len, int l_mm, int *n_mm )
{
	int	n = re.n_states;
	for( int st = 0; ; st++ ){
		int	cnt = 0, ok = 1;
		for( int j = 0; j < n; j++ ){
			if( st + j >= len ){
				ok = 0;
				break;
			}
			uint64_t	bit = 1ull << j;
			if( re.dot & bit )
				continue;
			if( !( re.accept[ rmd_code( sq, off + st + j ) ] & bit ) ){
				if( ++cnt > l_mm ){
					ok = 0;
					break;
				}
			}
		}
		if( ok && re.dollar && st + n != len )
			ok = 0;
		*n_mm = cnt;
		if( ok )
			return 1;
		if( re.anchored || st >= len )
			return 0;
	}
}

// The same two for expressions of 64 to 127 states (rmd_regex2_t): a set of states is two words.
struct rmd_set2_t { uint64_t lo, hi; };
RMD_FN rmd_set2_t rmd_s2( const uint64_t *w ) { return rmd_set2_t{ w[ 0 ], w[ 1 ] }; }
RMD_FN rmd_set2_t rmd_s2_and( rmd_set2_t a, rmd_set2_t b ) { return rmd_set2_t{ a.lo & b.lo, a.hi & b.hi }; }
RMD_FN rmd_set2_t rmd_s2_or( rmd_set2_t a, rmd_set2_t b ) { return rmd_set2_t{ a.lo | b.lo, a.hi | b.hi }; }
RMD_FN rmd_set2_t rmd_s2_shl1( rmd_set2_t a ) { return rmd_set2_t{ a.lo << 1, ( a.hi << 1 ) | ( a.lo >> 63 ) }; }
RMD_FN int rmd_s2_bit( rmd_set2_t a, int i ) { return int( ( ( i < 64 ? a.lo : a.hi ) >> ( i & 63 ) ) & 1 ); }

template< class SQ >
RMD_COLD int rmd_re2_step( const rmd_regex_t &re, const rmd_regex2_t &r2, const SQ &sq, int off, int len )
{
	rmd_set2_t	act{ 0, 0 };
	for( int pos = 0; ; pos++ ){
		rmd_set2_t	f = rmd_s2_or( rmd_s2_shl1( act ), rmd_s2_and( act, rmd_s2( r2.star ) ) );
		if( pos == 0 || !re.anchored )
			f.lo |= 1;
		for( int k = 0; k < r2.n_close; k++ )
			f = rmd_s2_or( f, rmd_s2_shl1( rmd_s2_and( f, rmd_s2( r2.opt ) ) ) );
		if( rmd_s2_bit( f, r2.n_states ) && ( !re.dollar || pos == len ) )
			return 1;
		if( pos == len )
			return 0;
		act = rmd_s2_and( f, rmd_s2( r2.accept[ rmd_code( sq, off + pos ) ] ) );
		if( ( act.lo | act.hi ) == 0 && re.anchored )
			return 0;
	}
}

template< class SQ >
RMD_COLD int rmd_re2_mm_step( const rmd_regex_t &re, const rmd_regex2_t &r2, const SQ &sq, int off, int len, int l_mm, int *n_mm )
{
	int	n = r2.n_states;
	for( int st = 0; ; st++ ){
		int	cnt = 0, ok = 1;
		for( int j = 0; j < n; j++ ){
			if( st + j >= len ){
				ok = 0;
				break;
			}
			if( rmd_s2_bit( rmd_s2( r2.dot ), j ) )
				continue;
			if( !rmd_s2_bit( rmd_s2( r2.accept[ rmd_code( sq, off + st + j ) ] ), j ) ){
				if( ++cnt > l_mm ){
					ok = 0;
					break;
				}
			}
		}
		if( ok && re.dollar && st + n != len )
			ok = 0;
		*n_mm = cnt;
		if( ok )
			return 1;
		if( re.anchored || st >= len )
			return 0;
	}
}

// chk_seq(), find_motif.c:1810
template< class SQ >
RMD_FN int rmd_chk_seq( const rmd_program_t *P, const rmd_elem_t &e, const SQ &sq, int off, int len, int *n_mm )
{
	const rmd_regex_t	&re = rmd_regexes( P )[ e.re ];
	if( re.wide >= 0 ){
		const rmd_regex2_t	&r2 = rmd_regexes2( P )[ re.wide ];
		if( e.mismatch > 0 )
			return rmd_re2_mm_step( re, r2, sq, off, len, e.mismatch, n_mm );
		return rmd_re2_step( re, r2, sq, off, len );
	}
	if( e.mismatch > 0 )
		return rmd_re_mm_step( re, sq, off, len, e.mismatch, n_mm );
	return rmd_re_step( re, sq, off, len );
}

// Necessary condition for chk_seq( e, s5, len ) with any len >= e.minlen: the
// leading mandatory positions of an anchored seq= must accept the bases at s5.
// Used to drop start positions before any search state is touched.
template< class SQ >
RMD_FN int rmd_prefix_ok( const rmd_program_t *P, const rmd_elem_t &e, const SQ &sq, int s5 )
{
	if( e.re < 0 || e.mismatch > 0 )
		return 1;
	const rmd_regex_t	&re = rmd_regexes( P )[ e.re ];
	int	n = re.n_prefix < e.minlen ? re.n_prefix : e.minlen;
	for( int i = 0; i < n; i++ )
		if( !( ( re.accept[ rmd_code( sq, s5 + i ) ] >> i ) & 1 ) )
			return 0;
	return 1;
}

// ---------------------------------------------------------------- sets of helix lengths
// The candidates of a helix at one pair of ends are a set of lengths (the reference's h3[ 101 ] / hlen[] / n_mpr[] arrays,
// find_motif.c:406): one 64-bit word for helices of up to 63 base pairs -- every descriptor of the reference's tests --
// and two words (rmd_lset_t, round 4: lengths to 127) in the one kernel instance compiled for longer ones (RMD_KIND_WIDE).
struct rmd_lset_t { uint64_t lo, hi; };
RMD_FN void rmd_ls_clear( uint64_t &s ) { s = 0; }
RMD_FN void rmd_ls_clear( rmd_lset_t &s ) { s.lo = s.hi = 0; }
RMD_FN bool rmd_ls_none( uint64_t s ) { return s == 0; }
RMD_FN bool rmd_ls_none( const rmd_lset_t &s ) { return ( s.lo | s.hi ) == 0; }
RMD_FN void rmd_ls_add( uint64_t &s, int b ) { s |= 1ull << b; }
RMD_FN void rmd_ls_add( rmd_lset_t &s, int b ) { if( b < 64 ) s.lo |= 1ull << b; else s.hi |= 1ull << ( b - 64 ); }
RMD_FN int rmd_ls_first( uint64_t s ) { return rmd_ctz64( s ); }		// (s is not empty)
RMD_FN int rmd_ls_first( const rmd_lset_t &s ) { return s.lo ? rmd_ctz64( s.lo ) : 64 + rmd_ctz64( s.hi ); }
RMD_FN void rmd_ls_drop_first( uint64_t &s ) { s &= s - 1; }
RMD_FN void rmd_ls_drop_first( rmd_lset_t &s ) { if( s.lo ) s.lo &= s.lo - 1; else s.hi &= s.hi - 1; }
RMD_FN void rmd_ls_keep_above( uint64_t &s, int hl ) { s = hl >= 63 ? 0 : s & ~( ( 2ull << hl ) - 1 ); }	// lengths > hl
RMD_FN void rmd_ls_keep_above( rmd_lset_t &s, int hl )
{
	if( hl >= 127 )
		s.lo = s.hi = 0;
	else if( hl >= 63 ){
		s.lo = 0;
		s.hi = hl == 63 ? s.hi : s.hi & ~( ( 2ull << ( hl - 64 ) ) - 1 );
	}else
		s.lo &= ~( ( 2ull << hl ) - 1 );
}
RMD_FN void rmd_ls_keep_range( uint64_t &s, int lo, int hi )		// lengths lo .. hi
{
	if( lo > 0 )
		s = lo >= 64 ? 0 : s & ~( ( 1ull << lo ) - 1 );
	if( hi < 63 )
		s = hi < 0 ? 0 : s & ( ( 2ull << hi ) - 1 );
}
RMD_FN void rmd_ls_keep_range( rmd_lset_t &s, int lo, int hi )
{
	if( lo > 0 ){
		if( lo >= 128 ) s.lo = s.hi = 0;
		else if( lo >= 64 ){ s.lo = 0; s.hi &= ~( ( 1ull << ( lo - 64 ) ) - 1 ); }
		else s.lo &= ~( ( 1ull << lo ) - 1 );
	}
	if( hi < 127 ){
		if( hi < 0 ) s.lo = s.hi = 0;
		else if( hi < 63 ){ s.hi = 0; s.lo &= ( 2ull << hi ) - 1; }
		else if( hi == 63 ) s.hi = 0;
		else s.hi &= ( 2ull << ( hi - 64 ) ) - 1;
	}
}
RMD_FN void rmd_ls_keep_only( uint64_t &s, int b ) { s &= 1ull << b; }
RMD_FN void rmd_ls_keep_only( rmd_lset_t &s, int b ) { if( b < 64 ){ s.lo &= 1ull << b; s.hi = 0; }else{ s.lo = 0; s.hi &= 1ull << ( b - 64 ); } }
RMD_FN int rmd_ls_count_below( uint64_t s, int hl ) { return rmd_popc64( hl >= 64 ? s : s & ( ( 1ull << hl ) - 1 ) ); }	// members < hl
RMD_FN int rmd_ls_count_below( const rmd_lset_t &s, int hl )
{
	if( hl <= 64 )
		return rmd_popc64( hl == 64 ? s.lo : s.lo & ( ( 1ull << hl ) - 1 ) );
	return rmd_popc64( s.lo ) + rmd_popc64( hl >= 128 ? s.hi : s.hi & ( ( 1ull << ( hl - 64 ) ) - 1 ) );
}

// ---------------------------------------------------------------- helix matchers
// match_wchlx(), find_motif.c:975.  Every candidate ends at s3, so the result
// is the set of accepted lengths (bit hl of *cand) and the mispair positions.
// mm5/mm3: s_n_mismatches of the two strands, in (current value) and out.
template< class SQ, class LS = uint64_t >
RMD_FN int rmd_match_wchlx_mm( const rmd_program_t *P, const SQ &sq,
	int d5, int d3, int s5, int s3, int s3lim, LS *cand, LS *mis, int *pmm5, int *pmm3 )
{
	const rmd_elem_t	&stp = P->elems[ d5 ], &stp3 = P->elems[ d3 ];
	RMD_COUNT( 4 );
	LS	c, m;
	rmd_ls_clear( c );
	rmd_ls_clear( m );
	int	hl, mpr, l_bpr, mm5 = *pmm5, mm3 = *pmm3;
	const uint32_t	mat2 = rmd_pairsets( P )[ stp.pairset ].mat2;	// (RM_paired :1291 for every pair below: the table once)

	if( stp.minlen == 0 ){
		int	ok = 1;
		if( stp.re >= 0 && !rmd_chk_seq( P, stp, sq, s5, 0, &mm5 ) )
			ok = 0;
		if( ok && ( stp3.re < 0 || rmd_chk_seq( P, stp3, sq, s3 + 1, 0, &mm3 ) ) )
			rmd_ls_add( c, 0 );
	}
	if( ( ( mat2 >> ( rmd_code( sq, s5 ) * 5 + rmd_code( sq, s3 ) ) ) & 1u ) ){
		hl = 1;
		mpr = 0;
		l_bpr = 1;
	}else if( !( stp.ends & RMA_5PAIRED ) ){
		hl = 1;
		mpr = 1;
		l_bpr = 0;
		rmd_ls_add( m, 0 );
	}else{
		*pmm5 = mm5;
		*pmm3 = mm3;
		*cand = c;
		*mis = m;
		return !rmd_ls_none( c );
	}
	for( ; ; ){
		if( hl >= stp.minlen ){
			int	ok = 1;
			if( !l_bpr && ( stp.ends & RMA_3PAIRED ) )
				ok = 0;
			else if( stp.pfrac && mpr > rmd_rules( P )[ stp.rule ].pf_maxmpr[ hl ] )
				ok = 0;
			else if( stp.re >= 0 && !rmd_chk_seq( P, stp, sq, s5, hl, &mm5 ) )
				ok = 0;
			else if( stp3.re >= 0 && !rmd_chk_seq( P, stp3, sq, s3 - hl + 1, hl, &mm3 ) )
				ok = 0;
			if( ok )
				rmd_ls_add( c, hl );
		}
		if( !( s3 - hl + 1 >= s3lim ) )
			break;
		if( hl >= stp.maxlen )
			break;
		if( ( ( mat2 >> ( rmd_code( sq, s5 + hl ) * 5 + rmd_code( sq, s3 - hl ) ) ) & 1u ) )
			l_bpr = 1;
		else{
			mpr++;
			if( mpr > stp.mplim )
				break;
			l_bpr = 0;
			rmd_ls_add( m, hl );
		}
		hl++;
	}
	*pmm5 = mm5;
	*pmm3 = mm3;
	*cand = c;
	*mis = m;
	return !rmd_ls_none( c );
}

// match_phlx(), find_motif.c:1114.  mm5/mm3: s_n_mismatches of the two strands, in and out.
RMD_COLD int rmd_match_phlx( const rmd_program_t *P, const rmd_seq_t &sq,
	int d5, int d3, int s5, int s3, int s5hi, int s5lo, int *hlen, int *n_mpr, int *mm5, int *mm3 )
{
	const rmd_elem_t	&stp = P->elems[ d5 ], &stp3 = P->elems[ d3 ];
	int	b3 = rmd_code( sq, s3 );
	for( int s = s5hi; s >= s5lo; s-- ){
		int	hl, mpr, l_pr;
		if( rmd_paired( P, stp.pairset, rmd_code( sq, s ), b3 ) ){
			hl = 1;
			mpr = 0;
			l_pr = 1;
		}else if( !( stp.ends & RMA_5PAIRED ) ){
			hl = 1;
			mpr = 1;
			l_pr = 0;
		}else
			continue;
		for( int s1 = s - 1; s1 >= s5; s1-- ){
			if( rmd_paired( P, stp.pairset, rmd_code( sq, s1 ), rmd_code( sq, s3 - hl ) ) )
				l_pr = 1;
			else{
				l_pr = 0;
				if( ++mpr > stp.mplim )
					return 0;
			}
			hl++;
		}
		if( !l_pr && ( stp.ends & RMA_3PAIRED ) )
			return 0;
		if( hl < stp.minlen || hl > stp.maxlen )
			return 0;
		if( stp.pfrac && mpr > rmd_rules( P )[ stp.rule ].pf_maxmpr[ hl ] )
			return 0;
		if( stp.re >= 0 && !rmd_chk_seq( P, stp, sq, s5, hl, mm5 ) )
			return 0;
		if( stp3.re >= 0 && !rmd_chk_seq( P, stp3, sq, s3 - hl + 1, hl, mm3 ) )
			return 0;
		*hlen = hl;
		*n_mpr = mpr;
		return 1;
	}
	return 0;
}

// match_triplex(), find_motif.c:1183.  mm1: s_n_mismatches of the middle strand.
RMD_COLD int rmd_match_triplex( const rmd_program_t *P, const rmd_seq_t &sq,
	int d, int d1, int s1, int s2, int s3, int tlen, int *n_mpr, int *mm1 )
{
	const rmd_elem_t	&stp = P->elems[ d ], &stp1 = P->elems[ d1 ];
	int	mplim = rmd_rules( P )[ stp.rule ].tq_mplim[ tlen ], mpr, l_pr;
	if( rmd_triple( P, stp.pairset, rmd_code( sq, s1 ), rmd_code( sq, s2 ), rmd_code( sq, s3 - tlen + 1 ) ) ){
		mpr = 0;
		l_pr = 1;
	}else if( !( stp.ends & RMA_5PAIRED ) ){
		mpr = 1;
		l_pr = 0;
	}else
		return 0;
	for( int t = 1; t < tlen; t++ ){
		if( !rmd_triple( P, stp.pairset, rmd_code( sq, s1 + t ), rmd_code( sq, s2 - t ), rmd_code( sq, s3 - tlen + 1 + t ) ) ){
			l_pr = 0;
			if( ++mpr > mplim )
				return 0;
		}else
			l_pr = 1;
	}
	if( !l_pr && ( stp.ends & RMA_3PAIRED ) )
		return 0;
	if( stp1.re >= 0 && !rmd_chk_seq( P, stp1, sq, s2 - tlen + 1, tlen, mm1 ) )
		return 0;
	*n_mpr = mpr;
	return 1;
}

// match_4plex(), find_motif.c:1234.  mm1/mm2: s_n_mismatches of the two inner strands.
RMD_COLD int rmd_match_4plex( const rmd_program_t *P, const rmd_seq_t &sq,
	int d1, int d2, int s1, int s2, int s3, int s4, int qlen, int *n_mpr, int *mm1, int *mm2 )
{
	const rmd_elem_t	&stp1 = P->elems[ d1 ], &stp2 = P->elems[ d2 ];
	int	mplim = rmd_rules( P )[ stp1.rule ].tq_mplim[ qlen ], mpr, l_pr;
	if( rmd_quad( P, stp1.pairset, rmd_code( sq, s1 + qlen - 1 ), rmd_code( sq, s2 ), rmd_code( sq, s3 ), rmd_code( sq, s4 - qlen + 1 ) ) )
		l_pr = 1;
	else if( !( stp1.ends & RMA_5PAIRED ) )
		l_pr = 0;
	else
		return 0;
	mpr = 0;	// (sic) find_motif.c:1260 restarts the count after the first quad
	for( int q = 1; q < qlen; q++ ){
		if( !rmd_quad( P, stp1.pairset, rmd_code( sq, s1 + qlen - 1 - q ), rmd_code( sq, s2 + q ),
			rmd_code( sq, s3 - q ), rmd_code( sq, s4 - qlen + 1 + q ) ) ){
			l_pr = 0;
			if( ++mpr > mplim )
				return 0;
		}else
			l_pr = 1;
	}
	if( !l_pr && ( stp1.ends & RMA_3PAIRED ) )
		return 0;
	if( stp1.re >= 0 && !rmd_chk_seq( P, stp1, sq, s2, qlen, mm1 ) )
		return 0;
	if( stp2.re >= 0 && !rmd_chk_seq( P, stp2, sq, s3 - qlen + 1, qlen, mm2 ) )
		return 0;
	*n_mpr = mpr;
	return 1;
}

// Does match_wchlx( s5, s3 ) have any candidate, seq= constraints aside?  A
// superset test for the pre-filter pass: same pairing, end and pairfrac rules
// (find_motif.c:1010-1109), no state written.
template< class SQ >
RMD_FN int rmd_quick_wchlx( const rmd_program_t *P, const rmd_elem_t &stp, const SQ &sq, int s5, int s3, int s3lim )
{
	if( stp.minlen == 0 )
		return 1;
	int	hl = 1, mpr = 0, l_bpr = 1;
	const uint32_t	mat2 = rmd_pairsets( P )[ stp.pairset ].mat2;
	if( !( ( mat2 >> ( rmd_code( sq, s5 ) * 5 + rmd_code( sq, s3 ) ) ) & 1u ) ){
		if( stp.ends & RMA_5PAIRED )
			return 0;
		mpr = 1;
		l_bpr = 0;
	}
	for( ; ; ){
		if( hl >= stp.minlen && ( l_bpr || !( stp.ends & RMA_3PAIRED ) ) && ( !stp.pfrac || mpr <= rmd_rules( P )[ stp.rule ].pf_maxmpr[ hl ] ) )
			return 1;
		if( !( s3 - hl + 1 >= s3lim ) || hl >= stp.maxlen )
			return 0;
		if( ( mat2 >> ( rmd_code( sq, s5 + hl ) * 5 + rmd_code( sq, s3 - hl ) ) ) & 1u )
			l_bpr = 1;
		else{
			if( ++mpr > stp.mplim )
				return 0;
			l_bpr = 0;
		}
		hl++;
	}
}

// The end positions a helix level walks down (find_motif :273) mostly fail at the first pair; when
// the helix must have its 5' end paired that pair decides alone (find_motif.c:1010-1021), and the 5'
// base is the same for every end: the partners it accepts as a 5-bit row, one code and one bit test
// per end.  Returns the first sd' <= sd whose base pairs with the one at s5, or lo - 1.
template< class SQ >
RMD_FN int rmd_skip_unpaired_ends( const rmd_program_t *P, const rmd_elem_t &stp, const SQ &sq, int z, int s5, int sd, int lo )
{
	if( stp.minlen == 0 || !( stp.ends & RMA_5PAIRED ) )
		return sd;
	const unsigned	row = ( rmd_pairsets( P )[ stp.pairset ].mat2 >> ( rmd_code( sq, z + s5 ) * 5 ) ) & 31u;
	while( sd >= lo && !( ( row >> rmd_code( sq, z + sd ) ) & 1u ) )
		sd--;
	return sd;
}


// ---------------------------------------------------------------- terminal checks
// fm_window[] lookup (find_motif.c:1333-1385 marks): type of the element that
// covers position pos, RMA_T_SS when nothing does and undef_is_ss, else -1.
RMD_FN_MEMBER int rmd_lane_t::wtype( const rmd_program_t *P, int pos, int undef_is_ss ) const
{
	for( int d = 0; d < P->n_elems; d++ ){
		if( mlen[ d ] > 0 && pos >= moff[ d ] && pos < moff[ d ] + mlen[ d ] )
			return P->elems[ d ].type;
	}
	return undef_is_ss ? RMA_T_SS : -1;
}

template< class TB, class SQ >
RMD_COLD int rmd_chk_motif( const rmd_program_t *P, const TB &L, const SQ &sq )	// chk_motif :1406
{
	for( int d = 0; d < P->n_elems; d++ ){
		const rmd_elem_t	&stp = P->elems[ d ];
		if( !stp.strict )
			continue;
		if( stp.type == RMA_T_H5 ){			// chk_wchlx :1441
			int	d3 = stp.mates[ 0 ];
			int	h5_5 = L.off( d ), h5_3 = h5_5 + L.len( d ) - 1;
			int	h3_5 = L.off( d3 ), h3_3 = h3_5 + L.len( d3 ) - 1;
			if( ( stp.strict & RMA_5STRICT ) && h5_5 > 0 && h3_3 < L.slen - 1 ){
				if( L.wtype( P, h5_5 - 1, 1 ) == RMA_T_SS && L.wtype( P, h3_3 + 1, 1 ) == RMA_T_SS &&
					rmd_paired( P, stp.pairset, rmd_code( sq, h5_5 - 1 ), rmd_code( sq, h3_3 + 1 ) ) )
					return 0;
			}
			if( stp.strict & RMA_3STRICT ){
				if( L.wtype( P, h5_3 + 1, 0 ) == RMA_T_SS && L.wtype( P, h3_5 - 1, 0 ) == RMA_T_SS &&
					rmd_paired( P, stp.pairset, rmd_code( sq, h5_3 + 1 ), rmd_code( sq, h3_5 - 1 ) ) )
					return 0;
			}
		}else if( stp.type == RMA_T_T1 ){		// chk_triplex :1557
			int	d1 = stp.mates[ 0 ], d2 = stp.mates[ 1 ];
			int	t1_5 = L.off( d ), t1_3 = t1_5 + L.len( d ) - 1;
			int	t2_5 = L.off( d1 ), t2_3 = t2_5 + L.len( d1 ) - 1;
			int	t3_5 = L.off( d2 ), t3_3 = t3_5 + L.len( d2 ) - 1;
			if( ( stp.strict & RMA_5STRICT ) && t1_5 > 0 ){
				if( L.wtype( P, t1_5 - 1, 1 ) == RMA_T_SS && L.wtype( P, t2_3 + 1, 0 ) == RMA_T_SS &&
					L.wtype( P, t3_5 - 1, 0 ) == RMA_T_SS &&
					rmd_triple( P, stp.pairset, rmd_code( sq, t1_5 - 1 ), rmd_code( sq, t2_3 + 1 ), rmd_code( sq, t3_5 - 1 ) ) )
					return 0;
			}
			if( ( stp.strict & RMA_3STRICT ) && t3_3 < L.slen - 1 ){
				if( L.wtype( P, t1_3 + 1, 0 ) == RMA_T_SS && L.wtype( P, t2_5 - 1, 0 ) == RMA_T_SS &&
					L.wtype( P, t3_3 + 1, 1 ) == RMA_T_SS &&
					rmd_triple( P, stp.pairset, rmd_code( sq, t1_3 + 1 ), rmd_code( sq, t2_5 - 1 ), rmd_code( sq, t3_3 + 1 ) ) )
					return 0;
			}
		}else if( stp.type == RMA_T_Q1 ){		// chk_4plex :1629
			int	d1 = stp.mates[ 0 ], d2 = stp.mates[ 1 ], d3 = stp.mates[ 2 ];
			int	q1_5 = L.off( d ), q1_3 = q1_5 + L.len( d ) - 1;
			int	q2_5 = L.off( d1 ), q2_3 = q2_5 + L.len( d1 ) - 1;
			int	q3_5 = L.off( d2 ), q3_3 = q3_5 + L.len( d2 ) - 1;
			int	q4_5 = L.off( d3 ), q4_3 = q4_5 + L.len( d3 ) - 1;
			if( ( stp.strict & RMA_5STRICT ) && q1_5 > 0 && q4_3 < L.slen - 1 ){
				if( L.wtype( P, q1_5 - 1, 1 ) == RMA_T_SS && L.wtype( P, q2_3 + 1, 0 ) == RMA_T_SS &&
					L.wtype( P, q3_5 - 1, 0 ) == RMA_T_SS && L.wtype( P, q4_3 + 1, 1 ) == RMA_T_SS &&
					rmd_quad( P, stp.pairset, rmd_code( sq, q1_5 - 1 ), rmd_code( sq, q2_3 + 1 ),
						rmd_code( sq, q3_5 - 1 ), rmd_code( sq, q4_3 + 1 ) ) )
					return 0;
			}
			if( stp.strict & RMA_3STRICT ){		// (sic) :1706 never looks at the q4 side
				if( L.wtype( P, q1_3 + 1, 0 ) == RMA_T_SS && L.wtype( P, q2_5 - 1, 0 ) == RMA_T_SS &&
					L.wtype( P, q3_3 + 1, 0 ) == RMA_T_SS &&
					rmd_quad( P, stp.pairset, rmd_code( sq, q1_3 + 1 ), rmd_code( sq, q2_5 - 1 ),
						rmd_code( sq, q3_3 + 1 ), rmd_code( sq, q4_5 - 1 ) ) )
					return 0;
			}
		}
		// RMA_T_P5: chk_phlx :1500 returns TRUE on every path
	}
	return 1;
}

template< class TB, class SQ >
RMD_COLD int rmd_set_context( const rmd_program_t *P, TB &L, const SQ &sq )	// set_context :1720
{
	if( P->has_lctx ){
		int	off = L.off( 0 ) - P->lctx.maxlen;
		if( off < 0 )
			off = 0;
		L.l_off = off;
		L.l_len = L.off( 0 ) - off;
		if( L.l_len < P->lctx.minlen )
			return 0;
		if( P->lctx.re >= 0 && !rmd_chk_seq( P, P->lctx, sq, off, L.l_len, &L.l_mm ) )
			return 0;
	}
	if( P->has_rctx ){
		int	n = P->n_elems - 1;
		L.r_off = L.off( n ) + L.len( n );
		int	end = L.r_off + P->rctx.maxlen;
		if( end > L.slen )
			end = L.slen;
		L.r_len = end - L.r_off;
		if( L.r_len < P->rctx.minlen )
			return 0;
		if( P->rctx.re >= 0 ){
			// (sic) :1749-1751 matches from the END of the context; the C string
			// it builds stops at the end of the sequence
			int	len = L.r_len;
			if( len > L.slen - end )
				len = L.slen - end;
			if( !rmd_chk_seq( P, P->rctx, sq, end, len, &L.r_mm ) )
				return 0;
		}
	}
	return 1;
}

template< class TB, class SQ >
RMD_COLD int rmd_chk_sites( const rmd_program_t *P, const TB &L, const SQ &sq )	// chk_sites :1758
{
	for( int s = 0; s < P->n_sites; s++ ){
		const rmd_site_t	&si = rmd_sites( P )[ s ];
		int	b[ 4 ] = { 0, 0, 0, 0 };
		for( int k = 0; k < si.n_pos; k++ ){
			int	d = si.elem[ k ], pos;
			if( si.l2r[ k ] ){
				if( si.offset[ k ] > L.len( d ) )
					return 0;
				pos = L.off( d ) + si.offset[ k ] - 1;
			}else if( si.offset[ k ] >= L.len( d ) )
				return 0;
			else
				pos = L.off( d ) + L.len( d ) - si.offset[ k ] - 1;
			b[ k ] = rmd_code( sq, pos );
		}
		int	rv = 0;
		if( si.n_pos == 2 )
			rv = rmd_paired( P, si.pairset, b[ 0 ], b[ 1 ] );
		else if( si.n_pos == 3 )
			rv = rmd_triple( P, si.pairset, b[ 0 ], b[ 1 ], b[ 2 ] );
		else if( si.n_pos == 4 )
			rv = rmd_quad( P, si.pairset, b[ 0 ], b[ 1 ], b[ 2 ], b[ 3 ] );
		if( !rv )
			return 0;
	}
	return 1;
}

// ---------------------------------------------------------------- helpers of the level generators
RMD_FN int rmd_imin( int a, int b ) { return a < b ? a : b; }
RMD_FN int rmd_imax( int a, int b ) { return a > b ? a : b; }

RMD_FN int rmd_s3lim( int szero, int sdollar, int i_minl, int h_maxl )	// find_motif.c:426-429
{
	int	v = sdollar - szero + 1;
	v = ( v - i_minl ) / 2;
	v = rmd_imin( v, h_maxl );
	return sdollar - v + 1;
}

// phlx/triplex end bounds, find_motif.c:730-739, :801-810
RMD_FN void rmd_phlx_bounds( int szero, int slen, int h_minl, int h_maxl, int i_minl, int i_maxsum, int *s5hi, int *s5lo )
{
	int	hi = rmd_imin( ( slen - i_minl ) / 2, h_maxl );
	*s5hi = szero + hi - 1;
	int	ilen = rmd_imin( slen - 2 * h_minl, i_maxsum );
	int	lo = slen - ilen;
	if( lo & 1 )
		lo++;
	lo = rmd_imin( lo / 2, h_maxl );
	*s5lo = szero + lo - 1;
}

// Write one candidate (find_ss :373-392 up to the RM_score() call).
RMD_COLD void rmd_fill_hit( const rmd_program_t *P, const rmd_lane_t *L, int seq, int comp, int szero, int32_t *w )
{
	w[ 0 ] = seq;
	w[ 1 ] = comp;
	w[ 2 ] = szero;
	w[ 3 ] = L->rank;
	w[ 4 ] = L->order;
	for( int d = 0; d < P->n_elems; d++ ){
		w[ RMA_HIT_HDR + 4 * d + 0 ] = L->moff[ d ];
		w[ RMA_HIT_HDR + 4 * d + 1 ] = L->mlen[ d ];
		w[ RMA_HIT_HDR + 4 * d + 2 ] = L->mpr[ d ];
		w[ RMA_HIT_HDR + 4 * d + 3 ] = L->mm[ d ];
	}
	int	k = RMA_HIT_HDR + 4 * P->n_elems;
	w[ k + 0 ] = P->has_lctx ? L->l_off : 0;
	w[ k + 1 ] = P->has_lctx ? L->l_len : 0;
	w[ k + 2 ] = P->has_rctx ? L->r_off : 0;
	w[ k + 3 ] = P->has_rctx ? L->r_len : 0;
	for( int e = 0; e < P->n_efn; e++ )
		w[ k + 4 + e ] = RMA_EFN_INFINITY;	// filled by the efn pass
}

// The search for one start position (one iteration of RM_find_motif's loops,
// find_motif.c:184-205).  Sink::put( lane ) stores the candidate.
// Range of end positions of the first search element at start szero
// (find_motif.c:266-269 with RM_find_motif's window, :179-205): hi downto lo.
RMD_FN void rmd_level0_range( const rmd_program_t *P, int szero, int slen, int *hi, int *lo )
{
	const rmd_elem_t	&stp = P->elems[ P->searches[ 0 ] ];
	int	d0 = rmd_imin( szero + P->w_winsize - 1, slen - 1 );
	if( stp.maxglen != RMA_UNBOUNDED && szero + stp.maxglen - 1 < d0 )
		d0 = szero + stp.maxglen - 1;
	*hi = d0;
	*lo = szero + stp.minglen - 1;
}

// r0/cnt select which end positions of the first element are searched: ranks
// r0 .. r0+cnt-1 counted from the largest; ( 0, RMD_ALL_RANKS ) is the whole position.
#define RMD_ALL_RANKS	0x7fffffff

// ---------------------------------------------------------------- lean path
// Descriptors whose search levels are all ss elements and proper Watson-Crick
// helices (the common case: hairpins, cloverleaves, ...) do not need the general
// frames.  Per level the whole iterator is 8 bytes -- window start, saved window
// end, next end position to try, chosen helix length, phase -- kept in LDS; helix
// candidate sets are recomputed when a level is resumed and the full element
// table is rebuilt only when a candidate reaches the end of the search list.  The
// transitions are the ss / proper-helix cases of rmd_next_general() (find_ss :332,
// find_wchlx :400, find_motif :245); tests/hostsim runs both against the oracle.
struct alignas( 8 ) rmd_lrec_t {
	int16_t	zero;		// window start, relative to the item's start position
	int16_t	osd;		// window end on entry (o_sdollar), relative
	int16_t	sd;		// next end position to try, relative
	uint8_t	hl;		// helix length of the current alternative
	uint8_t	ph;		// 0: looking for an end position, 1: alternative applied
};

struct rmd_lean_t {
	int32_t	szero, slen;
	int32_t	hi0, lo0;	// first level: end position of rank 0, lowest end position allowed
	int32_t	rank, order;
	int32_t	pretested;	// the item is one end position that already passed the first-pairs test
	// The 3' ends level hm_level (the first helix of the first element's interior) is allowed to take, as the
	// caller's tests found them (the kernel's pass A', from pair rows and stem-loop cores: necessary
	// conditions, so the ends left out lead to no candidate): bit j of hmask[ i ] = end number j, counted from
	// the shortest the helix may have, when the first element has its i-th length.  hm_level < 0: none.
	int32_t	hm_level;
	uint32_t	hmask[ 2 ];
	// >= 0: the walk takes this one length of the first element only (the caller walks the others as items of
	// their own; the candidates' order words tell the pieces apart, see the kernel's drain)
	int32_t	only_hl;
	// rmd_lean_step< ..., DEFER = true >() does not run the end-of-list checks of a complete structural
	// match itself (rmd_lean_emit): it sets this and returns, the caller runs them (the kernel does, all
	// lanes of the wave on one match: wave_emit), clears it and counts `order` up if the candidate was stored.
	int32_t	pending;
};

// The interior [a, b] (relative to z) of helix stp ends with a proper helix whose 3' strand
// must end at b: can any of its admissible 5' ends start it?
template< class SQ >
RMD_FN bool rmd_tail_ok( const rmd_program_t *P, const rmd_elem_t &stp, const SQ &sq, int z, int a, int b )
{
	const rmd_elem_t	&t = P->elems[ P->searches[ stp.tail_s ] ];
	int	s_hi = b - t.minglen + 1;
	int	s_lo = t.maxglen == RMA_UNBOUNDED ? a : b - t.maxglen + 1;
	if( a + stp.tail_pre_min > s_lo )
		s_lo = a + stp.tail_pre_min;
	if( stp.tail_pre_max >= 0 && a + stp.tail_pre_max < s_hi )
		s_hi = a + stp.tail_pre_max;
	// (a helix whose 5' end must pair: the 5' bases its 3' base accepts, one bit test per start)
	unsigned	col = 31u;
	if( t.minlen > 0 && ( t.ends & RMA_5PAIRED ) ){
		const unsigned	m = rmd_pairsets( P )[ t.pairset ].mat2 >> rmd_code( sq, z + b );
		col = ( m & 1u ) | ( ( m >> 4 ) & 2u ) | ( ( m >> 8 ) & 4u ) | ( ( m >> 12 ) & 8u ) | ( ( m >> 16 ) & 16u );
	}
	for( int s = s_hi; s >= s_lo; s-- )
		if( ( ( col >> rmd_code( sq, z + s ) ) & 1u ) &&
			rmd_quick_wchlx( P, t, sq, z + s, z + b, rmd_s3lim( s, b, t.q_iminl, t.maxlen ) + z ) )
			return true;
	return false;
}

RMD_FN rmd_lrec_t rmd_lean_open( const rmd_program_t *P, int level, int zero, int osd )	// rmd_enter()
{
	const rmd_elem_t	&stp = P->elems[ P->searches[ level ] ];
	rmd_lrec_t	r;
	int	hi = osd;
	if( stp.loop && stp.maxglen != RMA_UNBOUNDED && zero + stp.maxglen - 1 < hi )
		hi = zero + stp.maxglen - 1;
	if( osd - stp.rem_min < hi )		// the groups that follow need room
		hi = osd - stp.rem_min;
	r.zero = int16_t( zero );
	r.osd = int16_t( osd );
	r.sd = int16_t( hi );
	r.hl = 0;
	r.ph = 0;
	return r;
}

template< class LR >
RMD_FN int rmd_lean_begin( const rmd_program_t *P, LR &lr, rmd_lean_t &st, int szero, int slen, int r0, int cnt )
{
	st.szero = szero;
	st.slen = slen;
	st.rank = -1;
	st.order = 0;
	st.hm_level = -1;
	st.only_hl = -1;
	st.pending = 0;
	int	d0 = rmd_imin( szero + P->w_winsize - 1, slen - 1 ) - szero;
	rmd_lrec_t	r = rmd_lean_open( P, 0, 0, d0 );
	const rmd_elem_t	&stp = P->elems[ P->searches[ 0 ] ];
	st.hi0 = r.sd;
	st.lo0 = stp.minglen - 1;
	st.pretested = cnt == 1 && stp.quick;	// (rank items are only made from end positions that passed it)
	if( r0 > 0 || cnt != RMD_ALL_RANKS ){
		r.sd = int16_t( st.hi0 - r0 );
		if( cnt < st.hi0 - st.lo0 + 1 && r.sd - cnt + 1 > st.lo0 )
			st.lo0 = r.sd - cnt + 1;
	}
	lr.set( 0, r );
	return 0;
}

// Rebuild the element table of the current path into L and run the end-of-list
// checks (find_ss :362-393); called only for complete structural matches.
template< class LR, class Sink, class SQ >
RMD_FN void rmd_lean_emit( const rmd_program_t *P, LR &lr, rmd_lean_t &st, const SQ &sq, rmd_lane_t *L, Sink &sink )
{
	const int	z = st.szero;
	for( int k = 0; k < P->n_searches; k++ ){
		const rmd_lrec_t	r = lr.get( k );
		const int	d = P->searches[ k ];
		const rmd_elem_t	&stp = P->elems[ d ];
		const int	zero = z + r.zero, cur = z + r.sd + 1;
		if( stp.type == RMA_T_SS ){
			int	mm = 0;
			if( stp.re >= 0 && stp.mismatch > 0 )
				rmd_chk_seq( P, stp, sq, zero, cur - zero + 1, &mm );
			L->moff[ d ] = zero;
			L->mlen[ d ] = cur - zero + 1;
			L->mpr[ d ] = 0;
			L->mm[ d ] = int16_t( mm );
		}else{
			const int	d3 = stp.mates[ 0 ], hl = r.hl;
			uint64_t	cand, mis;
			int	mm5 = 0, mm3 = 0;
			rmd_match_wchlx_mm( P, sq, d, d3, zero, cur, rmd_s3lim( zero, cur, stp.minilen, stp.maxlen ), &cand, &mis, &mm5, &mm3 );
			int	mpr = rmd_popc64( mis & ( ( 1ull << hl ) - 1 ) );
			L->moff[ d ] = zero;
			L->mlen[ d ] = hl;
			L->moff[ d3 ] = cur - hl + 1;
			L->mlen[ d3 ] = hl;
			L->mpr[ d ] = L->mpr[ d3 ] = int16_t( mpr );
			L->mm[ d ] = int16_t( mm5 );
			L->mm[ d3 ] = int16_t( mm3 );
		}
	}
	L->slen = st.slen;
	L->szero = z;
	L->rank = st.rank;
	L->order = st.order;
	L->l_mm = L->r_mm = RMD_UNDEF;
	L->l_off = L->l_len = L->r_off = L->r_len = 0;
	if( P->strict_helices && !rmd_chk_motif( P, *L, sq ) )
		return;
	if( !rmd_set_context( P, *L, sq ) )
		return;
	if( !rmd_chk_sites( P, *L, sq ) )
		return;
	sink.put( P, L, z );
	st.order++;
}

// One transition at level k; returns the next level, -1 when the item is done.
// A caller may know a faster way to run the tail test of a level (the kernel does, from the
// bit vectors of its pre-filter): tail( stp, z, a, b, &result ) returns false when it does not.
struct rmd_no_accel_t {
	RMD_FN_MEMBER bool	tail( const rmd_elem_t &, int, int, int, bool * ) const { return false; }
};

template< class LR, class Sink, class SQ, class Accel = rmd_no_accel_t, bool DEFER = false >
RMD_FN int rmd_lean_step( const rmd_program_t *P, LR &lr, rmd_lean_t &st, const SQ &sq, int k,
	rmd_lane_t *L, Sink &sink, const Accel &accel = Accel() )
{
	const int	d = P->searches[ k ];
	// (a copy: the element's members are fetched together, at once, not one by one as the walk's branches ask for them --
	// a step is a chain of dependent LDS reads, and a third of them were these)
	const rmd_elem_t	stp = P->elems[ d ];
	const int	is_ss = stp.type == RMA_T_SS;
	const int	z = st.szero;
	rmd_lrec_t	r = lr.get( k );
	uint64_t	cand = 0, mis = 0;
	int	cur;

	if( r.ph != 0 ){
		r.ph = 0;
		if( !is_ss ){
			// back at a helix: its remaining lengths at the same end position
			cur = r.sd + 1;
			int	mm5 = 0, mm3 = 0;
			rmd_match_wchlx_mm( P, sq, d, stp.mates[ 0 ], z + r.zero, z + cur,
				rmd_s3lim( r.zero, cur, stp.minilen, stp.maxlen ) + z, &cand, &mis, &mm5, &mm3 );
			cand = r.hl >= 63 ? 0 : cand & ~( ( 2ull << r.hl ) - 1 );
			if( k == 0 && st.only_hl >= 0 )
				cand &= 1ull << st.only_hl;
		}
	}
	for( ; ; ){
		if( cand == 0 ){
			// next end position, find_motif :273
			int	lo = stp.loop ? r.zero + stp.minglen - 1 : r.osd;
			if( stp.rem_max >= 0 && r.osd - stp.rem_max > lo )	// ... and must reach the chain's end
				lo = r.osd - stp.rem_max;
			if( k == 0 && st.lo0 > lo )
				lo = st.lo0;
			if( stp.quick && !( k == 0 && st.pretested ) ){
				for( ; ; ){
					r.sd = int16_t( rmd_skip_unpaired_ends( P, stp, sq, z, r.zero, r.sd, lo ) );
					if( r.sd < lo || rmd_quick_wchlx( P, stp, sq, z + r.zero, z + r.sd,
						rmd_s3lim( r.zero, r.sd, stp.q_iminl, stp.maxlen ) + z ) )
						break;
					r.sd--;
				}
			}
			if( r.sd < lo )
				return stp.lean_back_s;		// (k - 1, or further back past levels that have no other alternative)
			cur = r.sd--;
			if( k == st.hm_level ){
				// an end the caller's tests have ruled out for this length of the first element
				const int	i0 = int( lr.get( 0 ).hl ) - P->elems[ P->searches[ 0 ] ].minlen, j = cur - ( r.zero + stp.minglen - 1 );
				if( i0 >= 0 && i0 < 2 && j >= 0 && j < 32 && !( ( st.hmask[ i0 ] >> j ) & 1u ) )
					continue;
			}
			if( stp.loop ){
				if( k == 0 ){
					st.rank = st.hi0 - cur;
					st.order = 0;
				}
				if( stp.next_s >= 0 )
					lr.set( stp.next_s, rmd_lean_open( P, stp.next_s, cur + 1, r.osd ) );
			}
			if( is_ss ){
				const int	len = cur - r.zero + 1;
				if( len < stp.minlen || len > stp.maxlen )
					continue;
				if( stp.re >= 0 ){
					int	mm = 0;
					if( !rmd_chk_seq( P, stp, sq, z + r.zero, len, &mm ) )
						continue;
				}
				if( k < P->n_searches - 1 ){
					// the next level's window may have been written long ago (end of an
					// inner chain) and its iterator used up since: start it afresh
					rmd_lrec_t	c = lr.get( k + 1 );
					c = rmd_lean_open( P, k + 1, c.zero, c.osd );
					// look ahead: if the next level is a helix and none of its end
					// positions can start it, this length of the ss leads nowhere
					const rmd_elem_t	nx = P->elems[ P->searches[ k + 1 ] ];
					if( nx.quick ){
						const int	nlo = nx.loop ? c.zero + nx.minglen - 1 : c.osd;
						for( ; ; ){
							c.sd = int16_t( rmd_skip_unpaired_ends( P, nx, sq, z, c.zero, c.sd, nlo ) );
							if( c.sd < nlo || rmd_quick_wchlx( P, nx, sq, z + c.zero, z + c.sd,
								rmd_s3lim( c.zero, c.sd, nx.q_iminl, nx.maxlen ) + z ) )
								break;
							c.sd--;
						}
						if( c.sd < nlo )
							continue;
					}
					r.ph = 1;
					lr.set( k, r );
					lr.set( k + 1, c );
					return k + 1;
				}
				r.ph = 1;
				lr.set( k, r );
				if constexpr( DEFER )
					st.pending = 1;
				else
					rmd_lean_emit( P, lr, st, sq, L, sink );
				return k;
			}
			int	mm5 = 0, mm3 = 0;
			if( !rmd_match_wchlx_mm( P, sq, d, stp.mates[ 0 ], z + r.zero, z + cur,
				rmd_s3lim( r.zero, cur, stp.minilen, stp.maxlen ) + z, &cand, &mis, &mm5, &mm3 ) ){
				cand = 0;
				continue;
			}
			if( k == 0 && st.only_hl >= 0 ){
				cand &= 1ull << st.only_hl;
				if( cand == 0 )
					continue;
			}
		}
		// helix: next accepted length, find_wchlx :435-460
		const int	hl = rmd_ctz64( cand );
		cand &= cand - 1;
		if( cur - r.zero - 2 * hl + 1 > stp.maxilen )
			continue;
		if( stp.tail_s >= 0 ){
			bool	ok;
			if( !accel.tail( stp, z, r.zero + hl, cur - hl, &ok ) )
				ok = rmd_tail_ok( P, stp, sq, z, r.zero + hl, cur - hl );
			if( !ok )
				continue;
		}
		r.hl = uint8_t( hl );
		r.ph = 1;
		lr.set( k, r );
		lr.set( stp.inner_s, rmd_lean_open( P, stp.inner_s, r.zero + hl, cur - hl ) );
		return k + 1;
	}
}

// ---------------------------------------------------------------- general path
// Every element type: ss, proper and improper (pseudoknot) Watson-Crick helices, parallel
// helices, triplexes, 4-plexes.  What find_motif()/find_1_motif()/find_ss()/find_wchlx()/
// find_pknot*()/find_phlx()/find_triplex()/find_4plex*() do by recursion
// (/root/reference/src/find_motif.c:245-973) is done with one 12-byte record per search level,
// kept in LDS by the kernel: the level's window (written by the levels above it, piecewise
// for the interiors of a pseudoknot: upd_pksearches :667), the next end position of
// find_motif's loop, the loop variables of the level's own type and the helix length of the
// alternative in use.  A level is a resumable generator: rmd_gen_step() advances the deepest
// level, descends when it yields and pops when it is exhausted.  Candidate sets of a helix
// (the reference's h3[]/hlen[]/n_mpr[] arrays) are recomputed when a level is resumed; the
// element table (s_matchoff ...) is rebuilt from the records only for complete matches
// (rmd_gen_emit).  All positions in a record are relative to the item's start position.
struct rmd_grec_t {
	int16_t	zero;		// window start
	int16_t	osd;		// window end on entry (o_sdollar)
	int16_t	sd;		// next end position to try; while ph != 0 the one in use is sd + 1
	int16_t	a;		// pknot: next 5' start (in use: a - 1); triplex: next end of the middle
				// strand (in use: a + 1); 4-plex: start of the second strand in use
	int16_t	c;		// pknot: next 3' end (in use: c + 1); 4-plex: next end of the third strand (in use: c + 1)
	uint8_t	hl;		// helix length of the alternative in use
	uint8_t	ph;		// generator phase (0: looking for an end position)
};

// the iterator part of a record as two words (how the kernel keeps it in LDS, and what a
// continuation carries): next end position | first loop variable, second loop variable | helix length | phase
RMD_FN uint32_t rmd_grec_word1( const rmd_grec_t &v ) { return uint32_t( uint16_t( v.sd ) ) | ( uint32_t( uint16_t( v.a ) ) << 16 ); }
RMD_FN uint32_t rmd_grec_word2( const rmd_grec_t &v )
{
	return uint32_t( uint16_t( v.c ) ) | ( uint32_t( v.hl ) << 16 ) | ( uint32_t( v.ph ) << 24 );
}
RMD_FN void rmd_grec_set_words( rmd_grec_t &r, uint32_t d1, uint32_t d2 )
{
	r.sd = int16_t( d1 & 0xffffu );
	r.a = int16_t( d1 >> 16 );
	r.c = int16_t( d2 & 0xffffu );
	r.hl = uint8_t( ( d2 >> 16 ) & 0xffu );
	r.ph = uint8_t( d2 >> 24 );
}

struct rmd_gen_t {
	int32_t	szero, slen;
	int32_t	hi0, lo0;	// first level: end position of rank 0, lowest end position allowed
	int32_t	rank, order;
	int32_t	pretested;	// the item is one end position that already passed the first-pairs test
	int32_t	wend;		// last position of the item's window (relative)
	int32_t	tag;		// >= 0: what candidates carry as their order (an alternative of the split level, below)
	// A step does a bounded amount of work: the loops of the generators count their iterations down
	// from budget0 and, at a point where the level's record says where they are, stop with
	// paused set; the next step of the level goes on from there.  Lanes of a wave that share a
	// level then leave a step together, whatever their items hold, and the ones that are done
	// take new items -- instead of all waiting for the longest scan among them.
	int32_t	budget, budget0, paused;
};
// (rmd_program_t::step_budget, at least 4: a step must get past the ticks of the nested loop heads --
// generator, end position, scan of 3' ends -- to the one that comes after some progress)
#define RMD_TICK( st_ )	( --( st_ ).budget < 0 ? ( ( st_ ).paused = 1 ) : 0 )

// find_minlen()/find_maxlen() over range q of improper helix pk, find_motif.c:642-665
template< class GR >
RMD_FN void rmd_pk_len( const rmd_pk_t &pk, const GR &gr, int q, int *mn, int *mx )
{
	int	a = pk.bmin[ q ], b = pk.bmax[ q ];
	for( unsigned m = pk.mask[ q ]; m; m &= m - 1 ){
		const int	hl = gr.hl( pk.lvl[ rmd_ctz64( m ) ] );
		a += hl;
		b += hl;
	}
	*mn = a;
	*mx = b;
}

// rmd_enter(): start the iterator of level k over its window (find_motif :266-272)
template< class GR >
RMD_FN void rmd_gen_open( const rmd_program_t *P, GR &gr, int k )
{
	const rmd_elem_t	&stp = P->elems[ P->searches[ k ] ];
	rmd_grec_t	r = gr.get( k );
	int	hi = r.osd;
	if( stp.loop && stp.maxglen != RMA_UNBOUNDED && r.zero + stp.maxglen - 1 < hi )
		hi = r.zero + stp.maxglen - 1;
	r.sd = int16_t( hi );
	r.ph = 0;
	gr.set( k, r );
}

template< class GR >
RMD_FN int rmd_gen_begin( const rmd_program_t *P, GR &gr, rmd_gen_t &st, int szero, int slen, int r0, int cnt )
{
	st.szero = szero;
	st.slen = slen;
	st.rank = -1;
	st.order = 0;
	st.tag = -1;
	st.budget0 = st.budget = P->step_budget;
	st.paused = 0;
	rmd_grec_t	r;
	r.zero = 0;
	r.osd = int16_t( rmd_imin( szero + P->w_winsize - 1, slen - 1 ) - szero );
	st.wend = r.osd;
	r.sd = r.a = r.c = 0;
	r.hl = r.ph = 0;
	gr.set( 0, r );
	rmd_gen_open( P, gr, 0 );
	const rmd_elem_t	&stp = P->elems[ P->searches[ 0 ] ];
	st.hi0 = gr.get( 0 ).sd;
	st.lo0 = stp.minglen - 1;
	st.pretested = cnt == 1 && stp.quick;	// (rank items of such a level are only made from end positions that passed it)
	if( r0 > 0 || cnt != RMD_ALL_RANKS ){
		r = gr.get( 0 );
		r.sd = int16_t( st.hi0 - r0 );
		if( cnt < st.hi0 - st.lo0 + 1 && r.sd - cnt + 1 > st.lo0 )
			st.lo0 = r.sd - cnt + 1;
		gr.set( 0, r );
	}
	return 0;
}

// A level pins the window of a later level (start and/or end, relative to z; RMD_NOPIN: not this
// one).  If an ss heads that level, what its anchored seq= demands there can be tested at once
// instead of after every level in between: the reference's chk_seq() on the final string
// (find_ss :346-351) can only succeed if these hold, so an alternative dropped here produces no
// candidate there (pk1.descr: the gaaa loop closed by the first helix' 3' strand).
#define RMD_NOPIN	( -0x7fff )
// limit: last position the level's window can reach (its end if known, else the item's)
RMD_FN bool rmd_pin_ok( const rmd_program_t *P, const rmd_seq_t &sq, int z, int level, int zero, int osd, int limit )
{
	const rmd_elem_t	&e = P->elems[ P->searches[ level ] ];
	if( e.type != RMA_T_SS )
		return true;
	if( zero != RMD_NOPIN && e.pin_start ){
		if( zero + e.minlen - 1 > limit || !rmd_prefix_ok( P, e, sq, z + zero ) )
			return false;
	}
	if( osd != RMD_NOPIN && e.pin_end_n > 0 ){
		const int	n = e.pin_end_n;
		int	mm = 0;
		if( osd - n + 1 < 0 || !rmd_chk_seq( P, e, sq, z + osd - n + 1, n, &mm ) )
			return false;
	}
	return true;
}

// A caller may know a faster way to find the 3' ends that can start a Watson-Crick helix (the
// kernel does, from bit vectors over the tile): ends( stp, s5, top, lo, &mask ) sets bit i of
// mask for every end position top-63+i >= lo (absolute positions) at which the first minlen
// pairs of helix stp with its 5' end at s5 stay within the mispair limit -- a superset of the
// ends where match_wchlx() finds a candidate -- and returns false when it cannot tell.
// The accelerator type also says which kinds of element the instance is compiled for (a kernel
// instance per class of descriptor keeps the code, and with it the registers, of the others out):
#define RMD_KIND_PK	1	// improper (pseudoknot) helices
#define RMD_KIND_TQ	2	// parallel helices, triplexes, 4-plexes
#define RMD_KIND_WIDE	4	// helices of 64 to 127 base pairs: sets of lengths are two words (rmd_lset_t)
template< bool WIDE > struct rmd_lset_sel { using type = uint64_t; };
template<> struct rmd_lset_sel<true> { using type = rmd_lset_t; };
struct rmd_no_ends_t {
	static constexpr int	kinds = RMD_KIND_PK | RMD_KIND_TQ;
	RMD_FN_MEMBER bool	ends( const rmd_elem_t &, int, int, int, uint64_t * ) const { return false; }
};

// Skip the 3' ends e, lo <= e <= *top (relative to z), at which a helix from s5 cannot start.
// Returns with *top at the next end worth a full match, or below lo.
template< class Accel >
RMD_FN void rmd_gen_skip_ends( const rmd_program_t *P, const rmd_seq_t &sq, const rmd_elem_t &stp, const Accel &accel,
	int z, int s5, int i_minl, int lo, int *top, rmd_gen_t &st )
{
	int	e = *top;
	while( e >= lo ){
		if( RMD_TICK( st ) )
			break;		// (paused: *top says how far the scan got)
		uint64_t	m;
		if( accel.ends( stp, z + s5, z + e, z + lo, &m ) ){
			if( m != 0 ){
				e -= rmd_clz64( m );	// the highest end position that passes (bit i: end e-63+i)
				break;
			}
			e -= 64;
			continue;
		}
		if( rmd_quick_wchlx( P, stp, sq, z + s5, z + e, rmd_s3lim( s5, e, i_minl, stp.maxlen ) + z ) )
			break;
		e--;
	}
	*top = e;
}

// Next end position of level k (find_motif :273-280): false when there is none left.
template< class GR, class Accel >
RMD_FN bool rmd_gen_next_sd( const rmd_program_t *P, GR &gr, rmd_gen_t &st, const rmd_seq_t &sq, int k,
	const rmd_elem_t &stp, rmd_grec_t &r, int *cur, const Accel &accel )
{
	const int	z = st.szero;
	int	lo = stp.loop ? r.zero + stp.minglen - 1 : r.osd;
	if( k == 0 && st.lo0 > lo )
		lo = st.lo0;
	for( ; ; ){
		if( stp.quick && !( k == 0 && st.pretested ) ){
			// end positions whose first base pairs cannot start this helix are skipped at once: they
			// have no effect that outlives the iteration (find_motif.c:273-280, 1010-1021)
			int	e = r.sd;
			rmd_gen_skip_ends( P, sq, stp, accel, z, r.zero, stp.q_iminl, lo, &e, st );
			r.sd = int16_t( e < lo ? lo - 1 : e );
			if( st.paused )
				return false;
		}
		if( r.sd < lo )
			return false;
		if( RMD_TICK( st ) )
			return false;
		*cur = r.sd--;
		// an end position after which the next group's anchored seq= cannot start leads nowhere
		if( !stp.loop || stp.next_s < 0 || rmd_pin_ok( P, sq, z, stp.next_s, *cur + 1, RMD_NOPIN, r.osd ) )
			break;
	}
	if( stp.loop ){
		if( k == 0 ){
			st.rank = st.hi0 - *cur;
			st.order = 0;
		}
		if( stp.next_s >= 0 )
			gr.set_window( stp.next_s, *cur + 1, r.osd );
	}
	return true;
}

// Advance level k to its next alternative: true with the alternative recorded in r and the
// windows of the levels it opens set, false when the level is exhausted.
template< class GR, class Accel >
RMD_GEN_FN bool rmd_gen_ss( const rmd_program_t *P, GR &gr, rmd_gen_t &st, const rmd_seq_t &sq, int k,
	const rmd_elem_t &stp, rmd_grec_t &r, const Accel &accel )		// find_ss :332
{
	r.ph = 0;
	for( int cur; rmd_gen_next_sd( P, gr, st, sq, k, stp, r, &cur, accel ); ){
		const int	len = cur - r.zero + 1;
		if( len < stp.minlen || len > stp.maxlen )
			continue;
		if( stp.re >= 0 ){
			int	mm = 0;
			if( !rmd_chk_seq( P, stp, sq, st.szero + r.zero, len, &mm ) )
				continue;
		}
		r.ph = 1;
		return true;
	}
	return false;
}

template< class GR, class Accel >
RMD_GEN_FN bool rmd_gen_wchlx( const rmd_program_t *P, GR &gr, rmd_gen_t &st, const rmd_seq_t &sq, int k,
	const rmd_elem_t &stp, rmd_grec_t &r, const Accel &accel )		// find_wchlx :400
{
	const int	z = st.szero, d = P->searches[ k ];
	typename rmd_lset_sel<( Accel::kinds & RMD_KIND_WIDE ) != 0>::type	cand, mis;
	rmd_ls_clear( cand );
	rmd_ls_clear( mis );
	int	cur = r.sd + 1, mm5 = 0, mm3 = 0;
	if( r.ph != 0 ){
		// back at the helix: its remaining lengths at the same end position
		r.ph = 0;
		rmd_match_wchlx_mm( P, sq, d, stp.mates[ 0 ], z + r.zero, z + cur,
			rmd_s3lim( r.zero, cur, stp.minilen, stp.maxlen ) + z, &cand, &mis, &mm5, &mm3 );
		rmd_ls_keep_above( cand, r.hl );
	}
	for( ; ; ){
		if( rmd_ls_none( cand ) ){
			if( !rmd_gen_next_sd( P, gr, st, sq, k, stp, r, &cur, accel ) )
				return false;
			mm5 = mm3 = 0;
			if( !rmd_match_wchlx_mm( P, sq, d, stp.mates[ 0 ], z + r.zero, z + cur,
				rmd_s3lim( r.zero, cur, stp.minilen, stp.maxlen ) + z, &cand, &mis, &mm5, &mm3 ) ){
				rmd_ls_clear( cand );
				continue;
			}
		}
		const int	hl = rmd_ls_first( cand );		// find_wchlx :435-460
		rmd_ls_drop_first( cand );
		if( cur - r.zero - 2 * hl + 1 > stp.maxilen )
			continue;
		if( !rmd_pin_ok( P, sq, z, stp.inner_s, r.zero + hl, cur - hl, cur - hl ) )
			continue;
		r.hl = uint8_t( hl );
		r.ph = 1;
		gr.set_window( stp.inner_s, r.zero + hl, cur - hl );
		return true;
	}
}

// Improper helix: find_pknot :465, find_pknot5 :495, find_pknot3 :530.
// Phases: 1 next 5' start, 2 next 3' end, 3 next helix length at (s5, s3).
// The knot's second helix must leave interiors of admissible length on both sides of the first
// helix' 3' strand (the hlx == 2 test, :613-627).  The reference tests that per matched length;
// here the same inequalities bound the 5' starts, 3' ends and lengths that are tried at all.
template< class GR, class Accel >
RMD_GEN_FN bool rmd_gen_pknot( const rmd_program_t *P, GR &gr, rmd_gen_t &st, const rmd_seq_t &sq, int k,
	const rmd_elem_t &stp, rmd_grec_t &r, const Accel &accel )
{
	const int	z = st.szero, d = P->searches[ k ], d3 = stp.mates[ 0 ];
	const rmd_pk_t	&pk = rmd_pks( P )[ stp.pk ];
	using LS = typename rmd_lset_sel<( Accel::kinds & RMD_KIND_WIDE ) != 0>::type;
	LS	cand, mis;
	rmd_ls_clear( cand );
	rmd_ls_clear( mis );
	int	cur = r.sd + 1;			// end position in use (ph != 0)
	int	i_minl = 0, i_maxl, l_s5 = 0, l_s3 = 0;
	int	iL_last = 0, iR_last = 0, iL_minl = 0, iL_maxl = 0, iR_minl = 0, iR_maxl = 0;	// second helix, :571-598
	int	hlo = stp.minlen, hhi = stp.maxlen;	// lengths the interiors allow at the 5' start in use
	if( pk.hlx2 ){
		const rmd_grec_t	h1 = gr.get( pk.lvl[ 0 ] );
		const int	s3_h1 = h1.c + 1;
		iL_last = s3_h1 - h1.hl;
		iR_last = s3_h1 + 1;
		rmd_pk_len( pk, gr, RMD_PK_IL, &iL_minl, &iL_maxl );
		rmd_pk_len( pk, gr, RMD_PK_IR, &iR_minl, &iR_maxl );
	}
	// the loop limits are functions of the helices matched above this level: recomputed on resume
	auto s5_limits = [ & ]( int *f_s5 ) -> bool {		// find_pknot5 :510-522
		int	p_minl, p_maxl, r_minl, r_maxl;
		rmd_pk_len( pk, gr, RMD_PK_P, &p_minl, &p_maxl );
		rmd_pk_len( pk, gr, RMD_PK_R, &r_minl, &r_maxl );
		const int	slen = cur - r.zero + 1;
		if( p_maxl + r_maxl < slen )
			return false;
		*f_s5 = r.zero + p_minl;
		l_s5 = r.zero + rmd_imin( p_maxl, slen - r_minl );
		return true;
	};
	auto s3_limits = [ & ]( int s5, int *f_s3 ) -> bool {	// find_pknot3 :555-568
		int	s_minl, s_maxl;
		rmd_pk_len( pk, gr, RMD_PK_I, &i_minl, &i_maxl );
		rmd_pk_len( pk, gr, RMD_PK_S, &s_minl, &s_maxl );
		const int	slen3 = cur - s5 + 1, g_minl = 2 * stp.minlen + i_minl;
		if( g_minl + s_minl > slen3 )
			return false;
		*f_s3 = cur - s_minl;
		l_s3 = cur - rmd_imin( slen3 - g_minl, s_maxl );
		if( pk.hlx2 ){
			// iL = iL_last - ( s5 + hl - 1 ) in [ iL_minl, iL_maxl ] bounds hl; with it
			// iR = ( s3 - hl + 1 ) - iR_last in [ iR_minl, iR_maxl ] bounds s3
			hlo = rmd_imax( stp.minlen, iL_last - s5 + 1 - iL_maxl );
			hhi = rmd_imin( stp.maxlen, iL_last - s5 + 1 - iL_minl );
			if( hlo > hhi )
				return false;
			*f_s3 = rmd_imin( *f_s3, hhi - 1 + iR_last + iR_maxl );
			l_s3 = rmd_imax( l_s3, hlo - 1 + iR_last + iR_minl );
		}
		return true;
	};
	// lengths hlo .. hhi of a candidate set
	auto clip = [ & ]( LS c ) -> LS {
		rmd_ls_keep_range( c, hlo, hhi );
		return c;
	};
	{
		// back at the level: the limits it was left with
		int	dummy;
		if( r.ph >= 1 )
			s5_limits( &dummy );
		if( r.ph >= 2 )
			s3_limits( r.a - 1, &dummy );
		if( r.ph == 3 ){
			const int	s5 = r.a - 1, s3 = r.c + 1;
			int	mm5 = 0, mm3 = 0;
			rmd_match_wchlx_mm( P, sq, d, d3, z + s5, z + s3, rmd_s3lim( s5, s3, i_minl, stp.maxlen ) + z, &cand, &mis, &mm5, &mm3 );
			cand = clip( cand );
			rmd_ls_keep_above( cand, r.hl );
		}
	}
	for( ; ; ){
		if( r.ph != 3 && RMD_TICK( st ) )
			return false;		// (phases 0..2 are all in the record)
		if( r.ph == 0 ){
			if( !rmd_gen_next_sd( P, gr, st, sq, k, stp, r, &cur, accel ) )
				return false;
			if( stp.scope == 0 ){
				// the other helices of the knot start from the knot's window, :476-487
				for( int s = 1; s < stp.n_scopes; s++ )
					if( P->elems[ stp.scopes[ s ] ].type == RMA_T_H5 )
						gr.set_window( pk.lvl[ s ], r.zero, cur );
			}
			int	f_s5;
			if( !s5_limits( &f_s5 ) )
				continue;
			r.a = int16_t( f_s5 );
			r.ph = 1;
		}
		if( r.ph == 1 ){		// next 5' start, find_pknot5 :523 / find_pknot3 :548-568
			if( r.a > l_s5 ){
				r.ph = 0;
				continue;
			}
			const int	s5 = r.a++;
			if( !rmd_prefix_ok( P, stp, sq, z + s5 ) )
				continue;
			int	f_s3;
			if( !s3_limits( s5, &f_s3 ) )
				continue;
			if( pk.w_osd5 >= 0 && !rmd_pin_ok( P, sq, z, pk.w_osd5, RMD_NOPIN, s5 - 1, s5 - 1 ) )
				continue;
			r.c = int16_t( f_s3 );
			r.ph = 2;
		}
		if( r.ph == 2 ){		// next 3' end, find_pknot3 :600
			const int	s5 = r.a - 1;
			// 3' ends whose first pairs cannot start the helix change nothing: skip them
			{
				int	e = r.c;
				rmd_gen_skip_ends( P, sq, stp, accel, z, s5, i_minl, l_s3, &e, st );
				r.c = int16_t( e < l_s3 ? l_s3 - 1 : e );
				if( st.paused )
					return false;
			}
			if( r.c < l_s3 ){
				r.ph = 1;
				continue;
			}
			const int	s3 = r.c--;
			if( pk.w_zero3 >= 0 && !rmd_pin_ok( P, sq, z, pk.w_zero3, s3 + 1, RMD_NOPIN, st.wend ) )
				continue;
			int	mm5 = 0, mm3 = 0;
			if( !rmd_match_wchlx_mm( P, sq, d, d3, z + s5, z + s3, rmd_s3lim( s5, s3, i_minl, stp.maxlen ) + z,
				&cand, &mis, &mm5, &mm3 ) )
				continue;
			cand = clip( cand );
			r.ph = 3;
		}
		// ph == 3: next helix length at (s5, s3), find_pknot3 :607
		const int	s5 = r.a - 1, s3 = r.c + 1;
		bool	found = false;
		int	hl = 0;
		while( !rmd_ls_none( cand ) ){
			hl = rmd_ls_first( cand );
			rmd_ls_drop_first( cand );
			if( ( s3 - s5 + 1 ) - 2 * hl < i_minl ){
				rmd_ls_clear( cand );	// break: longer helices only get worse
				break;
			}
			if( pk.hlx2 ){		// :613-627
				const int	il = iL_last - ( s5 + hl - 1 ), ir = ( s3 - hl + 1 ) - iR_last;
				if( il < iL_minl || il > iL_maxl || ir < iR_minl || ir > iR_maxl )
					continue;
			}
			// upd_pksearches(), :667: the interiors this choice pins
			if( pk.w_zero5 >= 0 && !rmd_pin_ok( P, sq, z, pk.w_zero5, s5 + hl, RMD_NOPIN, st.wend ) )
				continue;
			if( pk.w_osd3 >= 0 && !rmd_pin_ok( P, sq, z, pk.w_osd3, RMD_NOPIN, s3 - hl, s3 - hl ) )
				continue;
			found = true;
			break;
		}
		if( !found ){
			r.ph = 2;
			continue;
		}
		r.hl = uint8_t( hl );
		if( pk.w_osd5 >= 0 )
			gr.set_osd( pk.w_osd5, s5 - 1 );
		if( pk.w_zero5 >= 0 )
			gr.set_zero( pk.w_zero5, s5 + hl );
		if( pk.w_osd3 >= 0 )
			gr.set_osd( pk.w_osd3, s3 - hl );
		if( pk.w_zero3 >= 0 )
			gr.set_zero( pk.w_zero3, s3 + 1 );
		return true;
	}
}

template< class GR, class Accel >
RMD_GEN_FN bool rmd_gen_phlx( const rmd_program_t *P, GR &gr, rmd_gen_t &st, const rmd_seq_t &sq, int k,
	const rmd_elem_t &stp, rmd_grec_t &r, const Accel &accel )		// find_phlx :703
{
	const int	z = st.szero, d = P->searches[ k ];
	r.ph = 0;
	for( int cur; rmd_gen_next_sd( P, gr, st, sq, k, stp, r, &cur, accel ); ){
		int	s5hi, s5lo, hlen, n_mpr, mm5 = 0, mm3 = 0;
		rmd_phlx_bounds( z + r.zero, cur - r.zero + 1, stp.minlen, stp.maxlen, stp.minilen, stp.maxilen, &s5hi, &s5lo );
		if( !rmd_match_phlx( P, sq, d, stp.mates[ 0 ], z + r.zero, z + cur, s5hi, s5lo, &hlen, &n_mpr, &mm5, &mm3 ) )
			continue;
		if( cur - r.zero - 2 * hlen + 1 > stp.maxilen )
			continue;
		if( !rmd_pin_ok( P, sq, z, stp.inner_s, r.zero + hlen, cur - hlen, cur - hlen ) )
			continue;
		r.hl = uint8_t( hlen );
		r.ph = 1;
		gr.set_window( stp.inner_s, r.zero + hlen, cur - hlen );
		return true;
	}
	return false;
}

template< class GR, class Accel >
RMD_GEN_FN bool rmd_gen_triplex( const rmd_program_t *P, GR &gr, rmd_gen_t &st, const rmd_seq_t &sq, int k,
	const rmd_elem_t &stp, rmd_grec_t &r, const Accel &accel )		// find_triplex :763
{
	const int	z = st.szero, d = P->searches[ k ], d1 = stp.scopes[ 1 ], d2 = stp.scopes[ 2 ];
	const rmd_elem_t	&stp1 = P->elems[ d1 ];
	int	cur = r.sd + 1;
	for( ; ; ){
		if( r.ph == 0 ){
			if( !rmd_gen_next_sd( P, gr, st, sq, k, stp, r, &cur, accel ) )
				return false;
			int	s5hi, s5lo, hlen, n_mpr, mm5 = 0, mm3 = 0;
			rmd_phlx_bounds( z + r.zero, cur - r.zero + 1, stp.minlen, stp.maxlen, stp.minilen + stp1.minilen,
				stp.maxilen + stp.minlen + stp1.maxilen, &s5hi, &s5lo );
			if( !rmd_match_phlx( P, sq, d, d2, z + r.zero, z + cur, s5hi, s5lo, &hlen, &n_mpr, &mm5, &mm3 ) )
				continue;
			if( cur - r.zero - 2 * hlen + 1 > stp.maxilen + stp1.maxilen + hlen )
				continue;
			r.hl = uint8_t( hlen );
			// first end of the middle strand, :821 -- not beyond what the first interior's maximum
			// allows (:833, tested by the reference after the match)
			r.a = int16_t( cur - stp1.minilen - hlen );
			if( stp.maxilen < ( 1 << 28 ) && r.a > r.zero + 2 * hlen - 1 + stp.maxilen )
				r.a = int16_t( r.zero + 2 * hlen - 1 + stp.maxilen );
			r.ph = 1;
		}
		const int	hlen = r.hl;
		// ... and not below what the second interior's maximum allows (:835)
		const int	last = rmd_imax( r.zero + 2 * hlen + stp.minilen - 1, stp1.maxilen < ( 1 << 28 ) ? cur - hlen - stp1.maxilen : -( 1 << 28 ) );
		// middle strand ends whose base cannot complete the first triple are passed over
		// (match_triplex :1198-1206 returns at once when the 5' end must be paired)
		unsigned	m2 = 0x1f;
		if( ( stp.ends & RMA_5PAIRED ) && stp.tup >= 0 )
			m2 = rmd_tups( P )[ stp.tup ].t2[ rmd_code( sq, z + r.zero ) * 5 + rmd_code( sq, z + cur - hlen + 1 ) ];
		while( r.a >= last ){
			if( RMD_TICK( st ) )
				return false;
			const int	s = r.a--;
			if( !( ( m2 >> rmd_code( sq, z + s ) ) & 1 ) )
				continue;
			int	n_mpr, mm1 = 0;
			if( !rmd_match_triplex( P, sq, d, d1, z + r.zero, z + s, z + cur, hlen, &n_mpr, &mm1 ) )
				continue;
			if( s - 2 * hlen - r.zero + 1 > stp.maxilen )
				continue;
			if( cur - hlen - s > stp1.maxilen )
				continue;
			if( !rmd_pin_ok( P, sq, z, stp.inner_s, r.zero + hlen, s - hlen, s - hlen ) ||
				!rmd_pin_ok( P, sq, z, stp1.inner_s, s + 1, cur - hlen, cur - hlen ) )
				continue;
			gr.set_window( stp.inner_s, r.zero + hlen, s - hlen );
			gr.set_window( stp1.inner_s, s + 1, cur - hlen );
			return true;
		}
		r.ph = 0;
	}
}

// Phases: 1 next length of the outer helix, 2 next (s1, s2) of the inner strands.
template< class GR, class Accel >
RMD_GEN_FN bool rmd_gen_4plex( const rmd_program_t *P, GR &gr, rmd_gen_t &st, const rmd_seq_t &sq, int k,
	const rmd_elem_t &stp, rmd_grec_t &r, const Accel &accel )		// find_4plex :851, find_4plex_inner :902
{
	const int	z = st.szero, d = P->searches[ k ];
	const int	d1 = stp.mates[ 0 ], d2 = stp.mates[ 1 ], d3 = stp.mates[ 2 ];
	const rmd_elem_t	&stp1 = P->elems[ d1 ], &stp2 = P->elems[ d2 ];
	const int	i_minl = stp.minilen + stp1.minilen + stp2.minilen + 2 * stp.minlen;
	typename rmd_lset_sel<( Accel::kinds & RMD_KIND_WIDE ) != 0>::type	cand, mis;
	rmd_ls_clear( cand );
	rmd_ls_clear( mis );
	int	cur = r.sd + 1;
	bool	have_cand = false;		// cand holds the outer helix' lengths beyond r.hl
	for( ; ; ){
		if( r.ph == 0 ){
			if( !rmd_gen_next_sd( P, gr, st, sq, k, stp, r, &cur, accel ) )
				return false;
			int	mm5 = 0, mm3 = 0;
			if( !rmd_match_wchlx_mm( P, sq, d, d3, z + r.zero, z + cur, rmd_s3lim( r.zero, cur, i_minl, stp.maxlen ) + z,
				&cand, &mis, &mm5, &mm3 ) )
				continue;
			have_cand = true;
			r.ph = 1;
		}
		if( r.ph == 1 ){		// next outer helix length, find_4plex :893
			if( !have_cand ){
				int	mm5 = 0, mm3 = 0;
				rmd_match_wchlx_mm( P, sq, d, d3, z + r.zero, z + cur, rmd_s3lim( r.zero, cur, i_minl, stp.maxlen ) + z,
					&cand, &mis, &mm5, &mm3 );
				rmd_ls_keep_above( cand, r.hl );
				have_cand = true;
			}
			if( rmd_ls_none( cand ) ){
				r.ph = 0;
				have_cand = false;
				continue;
			}
			const int	hl = rmd_ls_first( cand );
			rmd_ls_drop_first( cand );
			r.hl = uint8_t( hl );
			r.a = int16_t( r.zero + hl + stp.minilen );		// s1
			r.c = int16_t( cur - hl - stp2.minilen );		// s2
			r.ph = 2;
		}
		// ph == 2: find_4plex_inner :902, s1 upwards, s2 downwards
		const int	hl = r.hl;
		const int	s1lim = cur - 3 * hl - stp2.minilen - stp1.minilen;
		// The scan of (s1, s2), start of the second strand and end of the third, with what
		// match_4plex() and the interior limits tested after it (:945-968) imply for each alone --
		// necessary conditions, so only pairs that cannot be accepted are passed over:
		//  * a quad holds only if its second base can complete it for some third (rmd_tup_t::q2);
		//    quads after the first that cannot hold count against the mispair limit, the first
		//    ends the match at once when the 5' end must be paired (:1249-1268);
		//  * the same for the third base once the second strand is fixed (q3), first quad only
		//    (the rest is match_4plex itself);
		//  * interiors longer than their maxima: s1 beyond the first, s2 outside the other two.
		const bool	masks = stp.tup >= 0, first5 = ( stp1.ends & RMA_5PAIRED ) != 0;
		const rmd_tup_t	&tup = rmd_tups( P )[ masks ? stp.tup : 0 ];
		const int	mplim = rmd_rules( P )[ stp1.rule ].tq_mplim[ hl ];
		const int	b14 = rmd_code( sq, z + r.zero + hl - 1 ) * 5 + rmd_code( sq, z + cur - hl + 1 );
		const int	s2_first = cur - hl - stp2.minilen;
		const int	big = 1 << 28;		// (an interior without upper limit: RMA_UNBOUNDED)
		const int	s1_last = stp.maxilen < big ? rmd_imin( s1lim, r.zero + hl - 1 + stp.maxilen ) : s1lim;
		while( r.a <= s1_last ){
			if( RMD_TICK( st ) )
				return false;
			const int	s1 = r.a;
			const int	b2 = rmd_code( sq, z + s1 );
			if( r.c == s2_first && masks ){
				// arriving at this s1: can the second strand stand at all?
				int	bad = 0;
				bool	ok = true;
				for( int q = 0; q < hl && ok; q++ ){
					const unsigned	m = tup.q2[ rmd_code( sq, z + r.zero + hl - 1 - q ) * 5 + rmd_code( sq, z + cur - hl + 1 + q ) ];
					if( !( ( m >> rmd_code( sq, z + s1 + q ) ) & 1 ) )
						ok = q == 0 ? !first5 : ++bad <= mplim;
				}
				if( !ok ){
					r.a++;
					continue;
				}
			}
			// s2 from the top of what the third interior's maximum allows down to what the second's and the
			// strands' room allow
			const int	s2_lo = stp2.maxilen < big ? rmd_imax( s1 + 2 * hl + stp1.minilen, cur - hl + 1 - stp2.maxilen ) : s1 + 2 * hl + stp1.minilen;
			if( stp1.maxilen < big && r.c > s1 + 2 * hl - 1 + stp1.maxilen )
				r.c = int16_t( s1 + 2 * hl - 1 + stp1.maxilen );
			if( r.c < s2_lo ){
				r.a++;
				r.c = int16_t( s2_first );
				continue;
			}
			const int	s2 = r.c--;
			if( masks && first5 && !( ( tup.q3[ ( b14 / 5 * 5 + b2 ) * 5 + b14 % 5 ] >> rmd_code( sq, z + s2 ) ) & 1 ) )
				continue;
			int	n_mpr, mm1 = 0, mm2 = 0;
			if( !rmd_match_4plex( P, sq, d1, d2, z + r.zero, z + s1, z + s2, z + cur, hl, &n_mpr, &mm1, &mm2 ) )
				continue;
			if( s1 - r.zero - hl + 1 > stp.maxilen )
				continue;
			if( s2 - s1 - 2 * hl + 1 > stp1.maxilen )
				continue;
			if( cur - s2 - hl + 1 > stp2.maxilen )
				continue;
			if( !rmd_pin_ok( P, sq, z, stp.inner_s, r.zero + hl, s1 - 1, s1 - 1 ) ||
				!rmd_pin_ok( P, sq, z, stp1.inner_s, s1 + hl, s2 - hl, s2 - hl ) ||
				!rmd_pin_ok( P, sq, z, stp2.inner_s, s2 + 1, cur - hl, cur - hl ) )
				continue;
			gr.set_window( stp.inner_s, r.zero + hl, s1 - 1 );
			gr.set_window( stp1.inner_s, s1 + hl, s2 - hl );
			gr.set_window( stp2.inner_s, s2 + 1, cur - hl );
			return true;
		}
		r.ph = 1;
		have_cand = false;
	}
}

// Rebuild the element table of the current path into L (every level holds an alternative)
// and run the end-of-list checks (find_ss :362-393).
template< int KINDS, class GR, class Sink >
RMD_GEN_FN void rmd_gen_emit( const rmd_program_t *P, GR &gr, rmd_gen_t &st, const rmd_seq_t &sq, rmd_lane_t *L, Sink &sink )
{
	constexpr bool	PK = ( KINDS & RMD_KIND_PK ) != 0, TQ = ( KINDS & RMD_KIND_TQ ) != 0;
	const int	z = st.szero;
	for( int k = 0; k < P->n_searches; k++ ){
		const rmd_grec_t	r = gr.get( k );
		const int	d = P->searches[ k ];
		const rmd_elem_t	&stp = P->elems[ d ];
		const int	zero = z + r.zero, cur = z + r.sd + 1;
		switch( stp.type ){
		case RMA_T_SS : {
			int	mm = 0;
			if( stp.re >= 0 && stp.mismatch > 0 )
				rmd_chk_seq( P, stp, sq, zero, cur - zero + 1, &mm );
			L->moff[ d ] = zero;
			L->mlen[ d ] = cur - zero + 1;
			L->mpr[ d ] = 0;
			L->mm[ d ] = int16_t( mm );
			break;
		}
		case RMA_T_H5 : {
			const int	d3 = stp.mates[ 0 ], hl = r.hl;
			typename rmd_lset_sel<( KINDS & RMD_KIND_WIDE ) != 0>::type	cand, mis;
			int	s5 = zero, s3 = cur, i_minl = stp.minilen, mm5 = 0, mm3 = 0;
			if constexpr( PK ){
			if( !stp.proper ){
				// s_n_mismatches of a knot's strands is not reset per attempt (find_pknot3): what
				// the search started with unless this match's chk_seq() calls count them
				int	i_maxl;
				s5 = z + r.a - 1;
				s3 = z + r.c + 1;
				rmd_pk_len( rmd_pks( P )[ stp.pk ], gr, RMD_PK_I, &i_minl, &i_maxl );
				mm5 = mm3 = RMD_UNDEF;
			}
			}
			rmd_match_wchlx_mm( P, sq, d, d3, s5, s3, rmd_s3lim( s5, s3, i_minl, stp.maxlen ), &cand, &mis, &mm5, &mm3 );
			const int	mpr = rmd_ls_count_below( mis, hl );
			L->moff[ d ] = s5;
			L->mlen[ d ] = hl;
			L->moff[ d3 ] = s3 - hl + 1;
			L->mlen[ d3 ] = hl;
			L->mpr[ d ] = L->mpr[ d3 ] = int16_t( mpr );
			L->mm[ d ] = int16_t( mm5 );
			L->mm[ d3 ] = int16_t( mm3 );
			break;
		}
		case RMA_T_P5 : if constexpr( TQ ){
			const int	d3 = stp.mates[ 0 ];
			int	s5hi, s5lo, hlen = 0, n_mpr = 0, mm5 = 0, mm3 = 0;
			rmd_phlx_bounds( zero, cur - zero + 1, stp.minlen, stp.maxlen, stp.minilen, stp.maxilen, &s5hi, &s5lo );
			rmd_match_phlx( P, sq, d, d3, zero, cur, s5hi, s5lo, &hlen, &n_mpr, &mm5, &mm3 );
			L->moff[ d ] = zero;
			L->mlen[ d ] = hlen;
			L->moff[ d3 ] = cur - hlen + 1;
			L->mlen[ d3 ] = hlen;
			L->mpr[ d ] = L->mpr[ d3 ] = int16_t( n_mpr );
			L->mm[ d ] = int16_t( mm5 );
			L->mm[ d3 ] = int16_t( mm3 );
			break;
		}
		case RMA_T_T1 : if constexpr( TQ ){
			const int	d1 = stp.scopes[ 1 ], d2 = stp.scopes[ 2 ];
			const rmd_elem_t	&stp1 = P->elems[ d1 ];
			int	s5hi, s5lo, hlen = 0, n_mpr = 0, mm5 = 0, mm3 = 0, mm1 = 0;
			rmd_phlx_bounds( zero, cur - zero + 1, stp.minlen, stp.maxlen, stp.minilen + stp1.minilen,
				stp.maxilen + stp.minlen + stp1.maxilen, &s5hi, &s5lo );
			rmd_match_phlx( P, sq, d, d2, zero, cur, s5hi, s5lo, &hlen, &n_mpr, &mm5, &mm3 );
			const int	s = z + r.a + 1;
			rmd_match_triplex( P, sq, d, d1, zero, s, cur, hlen, &n_mpr, &mm1 );
			L->moff[ d ] = zero;
			L->moff[ d1 ] = s - hlen + 1;
			L->moff[ d2 ] = cur - hlen + 1;
			L->mlen[ d ] = L->mlen[ d1 ] = L->mlen[ d2 ] = hlen;
			L->mpr[ d ] = L->mpr[ d1 ] = L->mpr[ d2 ] = int16_t( n_mpr );
			L->mm[ d ] = int16_t( mm5 );
			L->mm[ d1 ] = int16_t( mm1 );
			L->mm[ d2 ] = int16_t( mm3 );
			break;
		}
		case RMA_T_Q1 : if constexpr( TQ ){
			const int	d1 = stp.mates[ 0 ], d2 = stp.mates[ 1 ], d3 = stp.mates[ 2 ], hl = r.hl;
			const int	i_minl = stp.minilen + P->elems[ d1 ].minilen + P->elems[ d2 ].minilen + 2 * stp.minlen;
			typename rmd_lset_sel<( KINDS & RMD_KIND_WIDE ) != 0>::type	cand, mis;
			int	n_mpr = 0, mm5 = 0, mm3 = 0, mm1 = 0, mm2 = 0;
			rmd_match_wchlx_mm( P, sq, d, d3, zero, cur, rmd_s3lim( zero, cur, i_minl, stp.maxlen ), &cand, &mis, &mm5, &mm3 );
			const int	s1 = z + r.a, s2 = z + r.c + 1;
			rmd_match_4plex( P, sq, d1, d2, zero, s1, s2, cur, hl, &n_mpr, &mm1, &mm2 );
			L->moff[ d ] = zero;
			L->moff[ d1 ] = s1;
			L->moff[ d2 ] = s2 - hl + 1;
			L->moff[ d3 ] = cur - hl + 1;
			L->mlen[ d ] = L->mlen[ d1 ] = L->mlen[ d2 ] = L->mlen[ d3 ] = hl;
			L->mpr[ d ] = L->mpr[ d1 ] = L->mpr[ d2 ] = L->mpr[ d3 ] = int16_t( n_mpr );
			L->mm[ d ] = int16_t( mm5 );
			L->mm[ d1 ] = int16_t( mm1 );
			L->mm[ d2 ] = int16_t( mm2 );
			L->mm[ d3 ] = int16_t( mm3 );
			break;
		}
		default :
			break;
		}
	}
	L->slen = st.slen;
	L->szero = z;
	L->rank = st.rank;
	L->order = st.tag >= 0 ? st.tag : st.order;
	L->l_mm = L->r_mm = RMD_UNDEF;
	L->l_off = L->l_len = L->r_off = L->r_len = 0;
	if( P->strict_helices && !rmd_chk_motif( P, *L, sq ) )
		return;
	if( !rmd_set_context( P, *L, sq ) )
		return;
	if( !rmd_chk_sites( P, *L, sq ) )
		return;
	sink.put( P, L, z );
	if( st.tag < 0 )
		st.order++;
}

// The next alternative of level k (find_1_motif's dispatch, :289-330)
template< class GR, class Accel >
RMD_FN bool rmd_gen_next( const rmd_program_t *P, GR &gr, rmd_gen_t &st, const rmd_seq_t &sq, int k,
	const rmd_elem_t &stp, rmd_grec_t &r, const Accel &accel )
{
	constexpr bool	PK = ( Accel::kinds & RMD_KIND_PK ) != 0, TQ = ( Accel::kinds & RMD_KIND_TQ ) != 0;
	switch( stp.type ){
	case RMA_T_SS :
		return rmd_gen_ss( P, gr, st, sq, k, stp, r, accel );
	case RMA_T_H5 :
		if constexpr( PK ){
			if( !stp.proper )
				return rmd_gen_pknot( P, gr, st, sq, k, stp, r, accel );
		}
		return rmd_gen_wchlx( P, gr, st, sq, k, stp, r, accel );
	case RMA_T_P5 :
		if constexpr( TQ )
			return rmd_gen_phlx( P, gr, st, sq, k, stp, r, accel );
		return false;
	case RMA_T_T1 :
		if constexpr( TQ )
			return rmd_gen_triplex( P, gr, st, sq, k, stp, r, accel );
		return false;
	case RMA_T_Q1 :
		if constexpr( TQ )
			return rmd_gen_4plex( P, gr, st, sq, k, stp, r, accel );
		return false;
	default :
		return false;
	}
}

// Down from level k through the levels that have one alternative (an ss that is the whole of an
// interior: find_ss :332-351 on its window): the next level with a choice, n_searches when the
// match is complete, -1 when one of them fails.
template< class GR >
RMD_FN int rmd_gen_descend( const rmd_program_t *P, GR &gr, const rmd_gen_t &st, const rmd_seq_t &sq, int k )
{
	int	j = k + 1;
	for( ; j < P->n_searches; j++ ){
		const rmd_elem_t	&e = P->elems[ P->searches[ j ] ];
		if( e.type != RMA_T_SS || e.loop )
			break;
		rmd_grec_t	c = gr.get( j );
		const int	len = c.osd - c.zero + 1;
		int	mm = 0;
		if( len < e.minlen || len > e.maxlen || ( e.re >= 0 && !rmd_chk_seq( P, e, sq, st.szero + c.zero, len, &mm ) ) )
			return -1;
		c.sd = int16_t( c.osd - 1 );		// (as rmd_gen_ss() leaves it: the one end position, taken)
		c.ph = 1;
		gr.set_iter( j, c );
	}
	return j;
}

// The search tree of an item is narrow at the top and, now and then, deep: most lanes of a wave
// work on the first levels (the helices that head the search list) while the few that found
// something there walk the rest alone, each on its own code path.  With a split level S =
// rmd_program_t::split_s, an alternative of level S that survives is not walked by the lane that
// found it: it is handed over as a continuation -- the item, and for each level 0..S the iterator
// its generator was resumed from, which reproduces the alternative (generators are functions of
// the records above them and the sequence) -- and walked later next to other continuations
// (rmd_gen_resume, the kernel's second round over a tile).  Candidates then carry the number of
// their level-S alternative within (start, rank) as their order; the lane that walks an
// alternative emits its candidates in the reference's order, so the order of the hit buffer
// breaks the ties and the host renumbers (rma_scan).
struct rmd_no_split_t {
	RMD_FN_MEMBER int	level() const { return -1; }
	template< class GR >
	RMD_FN_MEMBER bool	push( const rmd_gen_t &, GR &, int ) const { return false; }
};

// One transition at level k; returns the next level, -1 when the item is done.  Levels whose
// element has a single alternative take no transition of their own: they are checked on the way
// down (rmd_gen_descend) and skipped on the way back (rmd_elem_t::back_s), so a lane spends its
// steps on the levels that have a choice -- and so do the lanes next to it.
// A step goes on across levels while its budget lasts (chain): down into the level an
// alternative opens, back to the level below when one is exhausted -- a lane that alternates
// between two levels (an ss of several lengths and the helix behind it) does not need a round
// for each visit.  It hands back at a level <= floor (the continuation's end, rmd_gen_resume).
template< class GR, class Sink, class Accel = rmd_no_ends_t, class Split = rmd_no_split_t >
RMD_FN int rmd_gen_step( const rmd_program_t *P, GR &gr, rmd_gen_t &st, const rmd_seq_t &sq, int k, rmd_lane_t *L, Sink &sink,
	const Accel &accel = Accel(), const Split &split = Split(), int floor = -1, bool chain = true )
{
	const int	S = split.level();
	st.budget = st.budget0;
	st.paused = 0;
	for( ; ; ){
		const rmd_elem_t	&stp = P->elems[ P->searches[ k ] ];
		rmd_grec_t	r = gr.get( k );
		int	next;
		for( ; ; ){
			if( S >= 0 && k <= S )
				gr.set_before( k, r );		// what this alternative's generator is resumed from
			if( !rmd_gen_next( P, gr, st, sq, k, stp, r, accel ) ){
				if( st.paused ){
					gr.set_iter( k, r );	// out of budget: the level goes on from here at its next step
					return k;
				}
				next = stp.back_s;
				break;
			}
			gr.set_iter( k, r );
			const int	j = rmd_gen_descend( P, gr, st, sq, k );
			if( j < 0 )
				continue;		// this alternative of level k leads nowhere: its next one
			if( k == S ){
				// an alternative of the split level: numbered, and walked by whoever takes the continuation
				const int	alt = st.order++;
				if( j < P->n_searches && split.push( st, gr, alt ) )
					continue;
				st.tag = alt;
			}
			if( j < P->n_searches ){
				rmd_gen_open( P, gr, j );
				next = j;
				break;
			}
			rmd_gen_emit<Accel::kinds>( P, gr, st, sq, L, sink );
			if( !chain || st.budget <= 0 )
				return k;
		}
		if( !chain || next <= floor || st.budget <= 0 )
			return next;
		k = next;
	}
}

// Take up a continuation: item (szero, r0, cnt) as rmd_gen_begin() takes it, before[ 2 * j ],
// before[ 2 * j + 1 ] the iterator (GR's packing of rmd_grec_t's sd | a and c | hl | ph) level
// j <= S was resumed from, alt the alternative's number.  Returns the level to go on with; the
// continuation is finished when rmd_gen_step() comes back to a level <= S.
template< class GR, class Accel >
RMD_FN int rmd_gen_resume( const rmd_program_t *P, GR &gr, rmd_gen_t &st, const rmd_seq_t &sq,
	int szero, int slen, int r0, int cnt, int S, const uint32_t *before, int alt, const Accel &accel )
{
	rmd_gen_begin( P, gr, st, szero, slen, r0, cnt );
	st.budget = 0x7fffffff;		// (reproducing an alternative is not a step to be cut short)
	for( int j = 0; j <= S; j++ ){
		gr.set_iter_words( j, before[ 2 * j ], before[ 2 * j + 1 ] );
		rmd_grec_t	r = gr.get( j );
		const rmd_elem_t	&stp = P->elems[ P->searches[ j ] ];
		if( r.ph != 0 ){
			// resumed within an end position: what taking that end position had set up
			// (rmd_gen_next_sd, and find_pknot :476-487 for the first helix of a knot)
			const int	cur = r.sd + 1;
			if( stp.loop && stp.next_s >= 0 )
				gr.set_window( stp.next_s, cur + 1, r.osd );
			if( ( Accel::kinds & RMD_KIND_PK ) && stp.type == RMA_T_H5 && !stp.proper && stp.scope == 0 ){
				const rmd_pk_t	&pk = rmd_pks( P )[ stp.pk ];
				for( int x = 1; x < stp.n_scopes; x++ )
					if( P->elems[ stp.scopes[ x ] ].type == RMA_T_H5 )
						gr.set_window( pk.lvl[ x ], r.zero, cur );
			}
		}
		if( !rmd_gen_next( P, gr, st, sq, j, stp, r, accel ) )
			return -1;		// (cannot happen: the alternative was found from this very state)
		gr.set_iter( j, r );
		if( j < S )
			rmd_gen_open( P, gr, j + 1 );
	}
	st.tag = alt;
	st.rank = st.hi0 - ( gr.get( 0 ).sd + 1 );	// (the first level may have been resumed within an end position)
	const int	j = rmd_gen_descend( P, gr, st, sq, S );
	if( j < 0 || j >= P->n_searches )
		return -1;			// (only alternatives with levels left to walk are handed over)
	rmd_gen_open( P, gr, j );
	return j;
}

// The search for one start position (one iteration of RM_find_motif's loops,
// find_motif.c:184-205), restricted to ranks r0 .. r0+cnt-1 of the first element.
template< class GR, class Sink, class Accel = rmd_no_ends_t >
RMD_FN void rmd_gen_position( const rmd_program_t *P, GR &gr, rmd_lane_t *L, const rmd_seq_t &sq,
	int szero, int slen, int r0, int cnt, Sink &sink, const Accel &accel = Accel() )
{
	rmd_gen_t	st;
	int	k = rmd_gen_begin( P, gr, st, szero, slen, r0, cnt );
	while( k >= 0 )
		k = rmd_gen_step( P, gr, st, sq, k, L, sink, accel );
}

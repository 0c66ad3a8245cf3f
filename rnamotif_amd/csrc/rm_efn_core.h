// rm_efn_core.h -- nearest neighbour energy of one candidate, for the device.
//
// RM_efn() with ef_stack/ef_ibloop/ef_hploop/ef_dangle/ef_aupen
// (/root/reference/src/efn.c:1162-1607) with the recursion over exterior and
// multi-branch loops replaced by an explicit interval stack, the base code and
// base pair arrays of setupefn()/setbp() (/root/reference/src/score.c:3128-3250)
// computed on demand from the candidate's element table instead of being
// materialised, and the tables read as int16 (they live in LDS in the kernel).
// Energies are integers in 1/100 kcal/mol; the loop-size logarithm comes from a
// host table (rma_efndata_t::loginc).
//
// Plain C++ like rm_scan_core.h: compiled for the device by the kernel and for
// the CPU by tests/hostsim.
#pragma once
#include "rm_dev_program.h"

#ifndef RMD_FN
#define RMD_FN	static inline
#endif
#ifndef RMD_FN_MEMBER
#define RMD_FN_MEMBER	inline
#endif

// offsets of the tables inside the flat int16 image
enum {
	RME_INTER = 0,				// [31]
	RME_BULGE = RME_INTER + 31,		// [31]
	RME_HAIRPIN = RME_BULGE + 31,		// [31]
	RME_DANGLE = RME_HAIRPIN + 31,		// [5][5][5][2]
	RME_POPPEN = RME_DANGLE + 250,		// [5]
	RME_EPARAM = RME_POPPEN + 5,		// [16]
	RME_MISC = RME_EPARAM + 16,		// maxpen auend gubonus cslope cint c3 gail ntriloops ntloops
	RME_TRIKEY = RME_MISC + 9,		// [50] 15-bit keys
	RME_TRIVAL = RME_TRIKEY + 50,		// [50]
	RME_TLVAL = RME_TRIVAL + 50,		// [100]
	RME_STACK = RME_TLVAL + 100,		// [5][5][5][5]
	RME_TSTKH = RME_STACK + 625,
	RME_TSTKI = RME_TSTKH + 625,
	RME_SINT2 = RME_TSTKI + 625,		// [6][6][5][5]
	RME_ASINT = RME_SINT2 + 900,		// [6][6][5][5][5]
	RME_SINT4 = RME_ASINT + 4500,		// [6][6][5][5][5][5]
	RME_N16 = RME_SINT4 + 22500
};
enum { RME_MAXPEN, RME_AUEND, RME_GUBONUS, RME_CSLOPE, RME_CINT, RME_C3, RME_GAIL, RME_NTRI, RME_NTL };

struct rme_tables_t {
	const int16_t	*t;		// RME_N16 entries (LDS in the kernel)
	const int32_t	*tlkey;		// [100] 18-bit tetraloop keys
	const int32_t	*loginc;	// [RMA_EFN_LOGINC]
};

// The candidate as efn sees it: bc( i ) = base code, bp( i ) = partner or -1,
// both indexed from the first base of the call (setupefn's i).
// BIG: a call over more than 15 helices (the descriptor language allows fifty): stacks and tables of the loops' walks sized
// for it -- an instance of the energy kernel of its own, chosen per descriptor (rmd_program_t::efn_big); the usual
// instance keeps the small ones (a few hundred bytes of scratch memory a lane instead of twelve thousand).
template< class Cand, int BIG = 0 >
struct rme_ctx_t {
	static constexpr int	stk = BIG ? 128 : 48;
	const rme_tables_t	*T;
	const Cand	*C;
	int	l_base;
	RMD_FN_MEMBER int	bc( int i ) const { return C->bc( i ); }
	RMD_FN_MEMBER int	bp( int i ) const { return C->bp( i ); }
	RMD_FN_MEMBER int	tab( int off ) const { return T->t[ off ]; }
};

#define RME_INF	RMA_EFN_INFINITY

template< class X > RMD_FN int rme_dangle( const X &x, int i, int j, int ip, int jp )	// ef_dangle :1582
{
	return x.tab( RME_DANGLE + ( ( x.bc( i ) * 5 + x.bc( j ) ) * 5 + x.bc( ip ) ) * 2 + jp );
}

template< class X > RMD_FN int rme_aupen( const X &x, int i, int j )	// ef_aupen :1591
{
	int	bi = x.bc( i ), bj = x.bc( j );
	int	pen = ( bi == RMA_BC_A && bj == RMA_BC_T ) || ( bi == RMA_BC_G && bj == RMA_BC_T ) ||
		( bi == RMA_BC_T && ( bj == RMA_BC_A || bj == RMA_BC_G ) );
	return pen * x.tab( RME_MISC + RME_AUEND );
}

template< class X > RMD_FN int rme_tab4( const X &x, int base, int a, int b, int c, int d )
{
	return x.tab( base + ( ( a * 5 + b ) * 5 + c ) * 5 + d );
}

template< class X > RMD_FN int rme_loginc( const X &x, int size )
{
	return x.T->loginc[ size < RMA_EFN_LOGINC ? size : RMA_EFN_LOGINC - 1 ];
}

template< class X > RMD_FN int rme_stack( const X &x, int i, int j )	// ef_stack :1337
{
	if( i == x.l_base || j == x.l_base + 1 )
		return RME_INF;
	return rme_tab4( x, RME_STACK, x.bc( i ), x.bc( j ), x.bc( i + 1 ), x.bc( j - 1 ) ) + x.tab( RME_EPARAM + 0 );
}

RMD_FN int rme_wc( int i, int j ) { return i + j == 3; }
RMD_FN int rme_gu( int i, int j ) { return i == RMA_BC_G && j == RMA_BC_T; }

template< class X > RMD_FN int rme_ibloop( const X &x, int i, int j, int ip, int jp )	// ef_ibloop :1351
{
	if( ( i <= x.l_base && ip > x.l_base ) || ( jp <= x.l_base && j > x.l_base ) )
		return RME_INF;
	int	size1 = ip - i - 1, size2 = j - jp - 1, size = size1 + size2;
	int	m = size1 < size2 ? size1 : size2;
	int	min4 = m < 4 ? m : 4;
	int	si = x.bc( i ), sj = x.bc( j ), sip = x.bc( ip ), sjp = x.bc( jp );
	int	rval = 0;
	if( size1 == 0 || size2 == 0 ){
		if( size == 1 )
			return rme_tab4( x, RME_STACK, si, sj, sip, sjp ) + x.tab( RME_BULGE + size ) + x.tab( RME_EPARAM + 1 );
		rval = rme_aupen( x, i, j ) + rme_aupen( x, ip, jp );
		if( size > 30 )
			rval += x.tab( RME_BULGE + 30 ) + rme_loginc( x, size ) + x.tab( RME_EPARAM + 1 );
		else
			rval += x.tab( RME_BULGE + size ) + x.tab( RME_EPARAM + 1 );
		return rval;
	}
	int	lopsid = size1 > size2 ? size1 - size2 : size2 - size1;
	int	pen = lopsid * x.tab( RME_POPPEN + min4 );
	if( pen > x.tab( RME_MISC + RME_MAXPEN ) )
		pen = x.tab( RME_MISC + RME_MAXPEN );
	int	gail = ( size1 == 1 || size2 == 1 ) && x.tab( RME_MISC + RME_GAIL ) == 1;
	if( size > 30 ){
		if( gail )
			rval += rme_tab4( x, RME_TSTKI, si, sj, RMA_BC_A, RMA_BC_A ) + rme_tab4( x, RME_TSTKI, sjp, sip, RMA_BC_A, RMA_BC_A );
		else
			rval += rme_tab4( x, RME_TSTKI, si, sj, x.bc( i + 1 ), x.bc( j - 1 ) ) +
				rme_tab4( x, RME_TSTKI, sjp, sip, x.bc( jp + 1 ), x.bc( ip - 1 ) );
		rval += x.tab( RME_INTER + 30 ) + rme_loginc( x, size ) + x.tab( RME_EPARAM + 2 ) + pen;
	}else if( lopsid == 1 && size == 3 ){
		int	lf, rt;
		if( size1 < size2 ){
			if( rme_wc( si, sj ) ) lf = si;
			else if( rme_gu( si, sj ) ) lf = 4;
			else if( rme_gu( sj, si ) ) lf = 5;
			else return RME_INF;
			if( rme_wc( sip, sjp ) ) rt = sip;
			else if( rme_gu( sip, sjp ) ) rt = 4;
			else if( rme_gu( sjp, sip ) ) rt = 5;
			else return RME_INF;
			rval += x.tab( RME_EPARAM + 2 ) + x.tab( RME_ASINT + ( ( ( lf * 6 + rt ) * 5 + x.bc( i + 1 ) ) * 5 + x.bc( j - 1 ) ) * 5 + x.bc( jp + 1 ) );
		}else{
			if( rme_wc( sjp, sip ) ) lf = sjp;
			else if( rme_gu( sjp, sip ) ) lf = 4;
			else if( rme_gu( sip, sjp ) ) lf = 5;
			else return RME_INF;
			if( rme_wc( sj, si ) ) rt = sj;
			else if( rme_gu( sj, si ) ) rt = 4;
			else if( rme_gu( si, sj ) ) rt = 5;
			else return RME_INF;
			rval += x.tab( RME_EPARAM + 2 ) + x.tab( RME_ASINT + ( ( ( lf * 6 + rt ) * 5 + x.bc( jp + 1 ) ) * 5 + x.bc( ip - 1 ) ) * 5 + x.bc( i + 1 ) );
		}
	}else if( lopsid == 0 && size <= 4 ){
		int	lf, rt;
		if( rme_wc( si, sj ) ) lf = si;
		else if( rme_gu( si, sj ) || rme_gu( sj, si ) ) lf = si + 2;
		else return RME_INF;
		if( rme_wc( sip, sjp ) ) rt = sip;
		else if( rme_gu( sip, sjp ) || rme_gu( sjp, sip ) ) rt = sip + 2;
		else return RME_INF;
		if( size == 2 )
			rval += x.tab( RME_EPARAM + 2 ) + x.tab( RME_SINT2 + ( ( lf * 6 + rt ) * 5 + x.bc( i + 1 ) ) * 5 + x.bc( j - 1 ) );
		else if( size == 4 )
			rval += x.tab( RME_EPARAM + 2 ) + x.tab( RME_SINT4 + ( ( ( ( lf * 6 + rt ) * 5 + x.bc( i + 1 ) ) * 5 + x.bc( j - 1 ) ) * 5 + x.bc( ip - 1 ) ) * 5 + x.bc( jp + 1 ) );
	}else{
		if( gail )
			rval += rme_tab4( x, RME_TSTKI, si, sj, RMA_BC_A, RMA_BC_A ) + rme_tab4( x, RME_TSTKI, sjp, sip, RMA_BC_A, RMA_BC_A );
		else
			rval += rme_tab4( x, RME_TSTKI, si, sj, x.bc( i + 1 ), x.bc( j - 1 ) ) +
				rme_tab4( x, RME_TSTKI, sjp, sip, x.bc( jp + 1 ), x.bc( ip - 1 ) );
		rval += x.tab( RME_EPARAM + 2 ) + x.tab( RME_INTER + ( size > 30 ? 30 : size ) ) + pen;
	}
	return rval;
}

template< class X > RMD_FN int rme_hploop( const X &x, int i, int j )	// ef_hploop :1498
{
	if( i <= x.l_base && j > x.l_base )
		return RME_INF;
	int	size = j - i - 1, rval = 0, ccnt = 0;
	for( int k = i + i; k < j; k++ ){		// (sic) efn.c:1511 starts the poly-C scan at i + i
		if( x.bc( k ) == RMA_BC_C )
			ccnt++;
		else
			break;
	}
	if( ccnt == size )
		rval = size == 3 ? x.tab( RME_MISC + RME_C3 ) : x.tab( RME_MISC + RME_CINT ) + size * x.tab( RME_MISC + RME_CSLOPE );
	if( i > 1 && j <= x.l_base ){
		if( x.bc( i ) == RMA_BC_G && x.bc( i - 1 ) == RMA_BC_G && x.bc( i - 2 ) == RMA_BC_G && x.bc( j ) == RMA_BC_T )
			rval += x.tab( RME_MISC + RME_GUBONUS );
	}
	if( size <= 3 ){
		int	lval = 0;
		if( size == 3 ){
			int	key = x.bc( i + size + 1 );
			for( int k = size + 1; k >= 0; k-- )
				key = ( key << 3 ) + x.bc( i + k );
			int	n = x.tab( RME_MISC + RME_NTRI );
			for( int k = 0; k < n; k++ ){
				// the reference compares full ints; a key with a code 4 base
				// can exceed 15 bits but then matches no table entry either way
				if( x.tab( RME_TRIKEY + k ) == key ){
					lval = x.tab( RME_TRIVAL + k );
					break;
				}
			}
		}
		rval += x.tab( RME_HAIRPIN + size ) + x.tab( RME_EPARAM + 3 ) + rme_aupen( x, i, j ) + lval;
	}else if( size <= 30 ){
		int	lval = 0;
		if( size == 4 ){
			int	key = x.bc( i + size + 1 );
			for( int k = size; k >= 0; k-- )
				key = ( key << 3 ) + x.bc( i + k );
			int	n = x.tab( RME_MISC + RME_NTL );
			for( int k = 0; k < n; k++ ){
				if( x.T->tlkey[ k ] == key ){
					lval = x.tab( RME_TLVAL + k );
					break;
				}
			}
		}
		rval += rme_tab4( x, RME_TSTKH, x.bc( i ), x.bc( j ), x.bc( i + 1 ), x.bc( j - 1 ) ) +
			x.tab( RME_HAIRPIN + size ) + x.tab( RME_EPARAM + 3 ) + lval;
	}else
		rval += rme_tab4( x, RME_TSTKH, x.bc( i ), x.bc( j ), x.bc( i + 1 ), x.bc( j - 1 ) ) +
			x.tab( RME_HAIRPIN + 30 ) + rme_loginc( x, size ) + x.tab( RME_EPARAM + 3 );
	return rval;
}


// RM_efn( 0, l_base, 1 ), efn.c:1162
template< class X > RMD_FN int rme_efn( const X &x )
{
	constexpr int	RME_STK = X::stk;
	int	stk_i[ RME_STK ], stk_j[ RME_STK ];
	unsigned long long	stk_open[ ( RME_STK + 63 ) / 64 ] = { 0 };
	int	sp = 0, e = 0;
	stk_i[ 0 ] = 0;
	stk_j[ 0 ] = x.l_base;
	stk_open[ 0 ] = 1;
	sp = 1;
	const int	fbp = x.tab( RME_EPARAM + 5 ), helixp = x.tab( RME_EPARAM + 8 );
	while( sp > 0 ){
		sp--;
		int	i = stk_i[ sp ], j = stk_j[ sp ];
		int	open = int( ( stk_open[ sp >> 6 ] >> ( sp & 63 ) ) & 1 );
		int	fb = open ? 0 : fbp;
		int	done = 0;
		// (RM_efn returns EFN_INFINITY itself from a call that meets a "knot", efn.c:1218,1262: what the call had
		// added up before that -- dangles, the helix' stacks -- is dropped, what its callers hold is not)
		const int	e_call = e;
		if( x.bp( i ) == -1 || x.bp( j ) == -1 ){
			while( x.bp( i ) == -1 && x.bp( i + 1 ) == -1 ){
				i++;
				e += fb;
				if( i >= j - 1 ){
					done = 1;
					break;
				}
			}
			if( done )
				continue;
			while( x.bp( j ) == -1 && x.bp( j - 1 ) == -1 ){
				j--;
				e += fb;
				if( i >= j - 1 ){
					done = 1;
					break;
				}
			}
			if( done )
				continue;
			if( x.bp( i ) == -1 && x.bp( i + 1 ) > i + 1 ){
				int	dg = rme_dangle( x, x.bp( i + 1 ), i + 1, i, 1 );
				e += ( dg < 0 ? dg : 0 ) + fb;
				i++;
			}
			if( x.bp( j ) == -1 && x.bp( j - 1 ) != -1 && x.bp( j - 1 ) < j - 1 ){
				int	dg = rme_dangle( x, j - 1, x.bp( j - 1 ), j, 0 );
				e += ( dg < 0 ? dg : 0 ) + fb;
				j--;
			}
		}
		if( x.bp( i ) != j ){
			int	k = x.bp( i ), kp = x.bp( j );
			if( k >= kp || sp + 2 > RME_STK ){
				e = e_call + RME_INF;		// "knot": pairs by a pair set of the descriptor's own (efn_usestdbp = 0) can make one
				continue;
			}
			int	cut;			// first interval ends at cut
			if( x.bp( k + 1 ) != -1 )
				cut = k;
			else if( x.bp( k + 2 ) == -1 )
				cut = k + 1;
			else if( rme_dangle( x, k, i, k + 1, 0 ) <= rme_dangle( x, x.bp( k + 2 ), k + 2, k + 1, 1 ) )
				cut = k + 1;
			else
				cut = k;
			// push the right part first so the left one is evaluated first
			stk_i[ sp ] = cut + 1;
			stk_j[ sp ] = j;
			stk_open[ sp >> 6 ] = ( stk_open[ sp >> 6 ] & ~( 1ull << ( sp & 63 ) ) ) | ( ( unsigned long long )open << ( sp & 63 ) );
			sp++;
			stk_i[ sp ] = i;
			stk_j[ sp ] = cut;
			stk_open[ sp >> 6 ] = ( stk_open[ sp >> 6 ] & ~( 1ull << ( sp & 63 ) ) ) | ( ( unsigned long long )open << ( sp & 63 ) );
			sp++;
			continue;
		}
		if( !open )
			e += helixp;
		e += rme_aupen( x, i, j );
		for( ; ; ){
			if( x.bp( i + 1 ) == j - 1 ){
				e += rme_stack( x, i, j );
				i++;
				j--;
				continue;
			}
			int	sum = 0, ip = 0, jp = 0, bad = 0;
			for( int k = i + 1; k < j; ){
				int	b = x.bp( k );
				if( b > k ){
					sum++;
					ip = k;
					k = b + 1;
					jp = k - 1;
					if( k > j ){
						bad = 1;
						break;
					}
				}else if( b == -1 )
					k++;
				else{
					bad = 1;
					break;
				}
			}
			if( bad ){
				e = e_call + RME_INF;
				break;
			}
			if( sum == 0 ){
				e += rme_hploop( x, i, j );
				break;
			}
			if( sum == 1 ){
				e += rme_ibloop( x, i, j, ip, jp );
				i = ip;
				j = jp;
				continue;
			}
			int	is = i + 1, js = j - 1;
			e += x.tab( RME_EPARAM + 4 ) + helixp + rme_aupen( x, i, j );
			if( x.bp( i + 1 ) == -1 && x.bp( i + 2 ) != -1 ){
				int	dg = rme_dangle( x, i, j, i + 1, 0 );
				if( dg <= rme_dangle( x, x.bp( i + 2 ), i + 2, i + 1, 1 ) ){
					is = i + 2;
					e += ( dg < 0 ? dg : 0 ) + fbp;
				}
			}
			if( x.bp( i + 1 ) == -1 && x.bp( i + 2 ) == -1 ){
				int	dg = rme_dangle( x, i, j, i + 1, 0 );
				is = i + 2;
				e += ( dg < 0 ? dg : 0 ) + fbp;
			}
			if( x.bp( j - 1 ) == -1 && x.bp( j - 2 ) != -1 ){
				int	dg = rme_dangle( x, i, j, j - 1, 1 );
				if( dg <= rme_dangle( x, j - 2, x.bp( j - 2 ), j - 1, 0 ) ){
					js = j - 2;
					e += ( dg < 0 ? dg : 0 ) + fbp;
				}
			}
			if( x.bp( j - 1 ) == -1 && x.bp( j - 2 ) == -1 ){
				int	dg = rme_dangle( x, i, j, j - 1, 1 );
				js = j - 2;
				e += ( dg < 0 ? dg : 0 ) + fbp;
			}
			if( sp + 1 > RME_STK ){
				e += RME_INF;
				break;
			}
			stk_i[ sp ] = is;
			stk_j[ sp ] = js;
			stk_open[ sp >> 6 ] &= ~( 1ull << ( sp & 63 ) );
			sp++;
			break;
		}
	}
	return e;
}

// The candidate view: elements idx..idx2 of a hit record laid over a sequence.
// Seq::code( p ) returns the base code at strand position p.
template< class Seq >
struct rme_cand_t {
	const rmd_program_t	*P;
	const int32_t	*w;		// hit record
	const Seq	*sq;
	int	idx, idx2, pos, pos2;	// pos2 resolved (last base of idx2 included)
	int	off5, start, len;
	// optional per-candidate arrays (the kernel keeps them in LDS for calls of up to
	// RME_CACHE bases): what setupefn() materialises as rm_bcseq[] / rm_basepr[]
	int16_t	*cbp = nullptr;
	uint8_t	*cbc = nullptr;

	RMD_FN_MEMBER int	moff( int d ) const { return w[ RMA_HIT_HDR + 4 * d ]; }
	RMD_FN_MEMBER int	mlen( int d ) const { return w[ RMA_HIT_HDR + 4 * d + 1 ]; }

	// setupefn :3128: returns 0 when the call is one the reference rejects
	RMD_FN_MEMBER int	setup( const rma_efn_site_t &es )
	{
		idx = es.idx;
		idx2 = es.idx2;
		pos = es.pos;
		pos2 = es.pos2 < 0 ? mlen( idx2 ) - 1 : es.pos2;
		off5 = moff( idx );
		start = off5 + pos;
		len = 0;
		for( int d = idx; d <= idx2; d++ ){
			int	t = P->elems[ d ].type;
			if( t != RMA_T_SS && t != RMA_T_H5 && t != RMA_T_H3 )
				return 0;
			len += mlen( d );
		}
		len -= pos;
		len -= mlen( idx2 ) - ( pos2 + 1 );
		if( len <= 0 )
			return 0;
		// setbp :3216 on every helix base: all partners must fall inside the call
		for( int d = idx; d <= idx2; d++ ){
			const rmd_elem_t	&st = P->elems[ d ];
			if( st.type == RMA_T_SS || ( st.type == RMA_T_H5 && d == idx2 ) || ( st.type == RMA_T_H3 && d == idx ) )
				continue;
			if( !st.proper )
				return 0;
			int	m = st.mates[ 0 ];
			int	p0 = d == idx ? pos : 0, p1 = d == idx2 ? pos2 + 1 : mlen( d );
			if( p0 >= p1 )
				continue;
			int	lo = ( mlen( m ) - ( p1 - 1 ) - 1 ) + moff( m ) - off5;
			int	hi = ( mlen( m ) - p0 - 1 ) + moff( m ) - off5;
			if( lo < 0 || hi >= len )
				return 0;
		}
		return 1;
	}
	// fill cbc[] / cbp[] for the whole call: one pass over the elements
	RMD_FN_MEMBER void	fill_cache( int16_t *bpbuf, uint8_t *bcbuf )
	{
		for( int d = idx; d <= idx2; d++ ){
			const rmd_elem_t	&st = P->elems[ d ];
			const int	p0 = d == idx ? pos : 0, p1 = d == idx2 ? pos2 + 1 : mlen( d );
			const bool	ss = st.type == RMA_T_SS || ( st.type == RMA_T_H5 && d == idx2 ) || ( st.type == RMA_T_H3 && d == idx );
			const int	m = st.mates[ 0 ];
			const int	ps = P->efn_usestdbp ? P->efn_stdbp : st.pairset;
			for( int pq = p0; pq < p1; pq++ ){
				const int	p = moff( d ) + pq, i = p - start;
				const int	b = sq->code( p );
				bcbuf[ i ] = uint8_t( b );
				int	partner = -1;
				if( !ss ){
					const int	q1 = mlen( m ) - pq - 1;
					const int	b1 = sq->code( q1 + moff( m ) );
					if( ( rmd_pairsets( P )[ ps ].mat2 >> ( b * 5 + b1 ) ) & 1 )
						partner = q1 + moff( m ) - off5;
				}
				bpbuf[ i ] = int16_t( partner );
			}
		}
		cbp = bpbuf;
		cbc = bcbuf;
	}
	RMD_FN_MEMBER int	bc( int i ) const
	{
		if( i < 0 || i >= len )
			return RMA_BC_N;	// the reference reads stale rm_bcseq[] there; never reached for nested helices
		if( cbc != nullptr )
			return cbc[ i ];
		return sq->code( start + i );
	}
	RMD_FN_MEMBER int	bp( int i ) const
	{
		if( i < 0 || i >= len )
			return -1;
		if( cbp != nullptr )
			return cbp[ i ];
		int	p = start + i;
		for( int d = idx; d <= idx2; d++ ){
			int	o = moff( d ), l = mlen( d );
			if( p < o || p >= o + l )
				continue;
			const rmd_elem_t	&st = P->elems[ d ];
			if( st.type == RMA_T_SS || ( st.type == RMA_T_H5 && d == idx2 ) || ( st.type == RMA_T_H3 && d == idx ) )
				return -1;
			int	pq = p - o;
			int	m = st.mates[ 0 ];
			int	q1 = mlen( m ) - pq - 1;
			int	ps = P->efn_usestdbp ? P->efn_stdbp : st.pairset;
			int	b = sq->code( p ), b1 = sq->code( q1 + moff( m ) );
			return ( ( rmd_pairsets( P )[ ps ].mat2 >> ( b * 5 + b1 ) ) & 1 ) ? q1 + moff( m ) - off5 : -1;
		}
		return -1;
	}
};

// Energy of efn site k for the hit record w; what do_sc_efnx() returns before
// the 0.01 scaling (score.c:1672-1679).
// (bpbuf/bcbuf: room for the base codes and partners of a call of up to cache bases; longer calls
// compute them on demand from the hit record)
template< class Seq, int BIG = 0 >
RMD_FN int rme_site_energy( const rmd_program_t *P, const rme_tables_t *T, const Seq *sq, const int32_t *w, int k,
	int16_t *bpbuf = nullptr, uint8_t *bcbuf = nullptr, int cache = 0 )
{
	rme_cand_t<Seq>	c;
	c.P = P;
	c.w = w;
	c.sq = sq;
	if( !c.setup( rmd_efn_sites( P )[ k ] ) )
		return RME_INF;
	if( bpbuf != nullptr && c.len <= cache )
		c.fill_cache( bpbuf, bcbuf );
	rme_ctx_t< rme_cand_t<Seq>, BIG >	x;
	x.T = T;
	x.C = &c;
	x.l_base = c.len - 1;
	return rme_efn( x );
}

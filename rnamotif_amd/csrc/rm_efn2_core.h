// rm_efn2_core.h -- efn2(): nearest neighbour energy with coaxial stacking, for the device.
//
// RM_efn2() with ef2_stack / ef2_ibloop / ef2_hploop / ef2_dangle / ef2_aupen and the four
// coaxial table look-ups (/root/reference/src/efn2.c:1103-1777) over the candidate view of
// rm_efn_core.h (base codes and partners computed on demand from the hit record, as
// setupefn()/setbp() would fill them, score.c:3128-3250).  The tables (rma_efn2data_t,
// 1.9 MB with the 5^8 entries of the 2x2 interior loops) stay in global memory; the two
// logarithms come from host tables.  Energies are integers in 1/100 kcal/mol.
//
// The reference walks the exterior loop with "rm_basepr[ip]==0" as its test for an
// unpaired base although unpaired is -1 in its arrays (efn2.c:1337, efn2_drv.c:143-146), and
// leaves them unless every exterior helix starts exactly where that test stops.  The same
// steps are taken here; where the reference would index outside its arrays the result is
// RMA_EFN2_INFINITY (the oracle, oracle/rm_oracle_efn2.c, flags the same inputs).
#pragma once
#include "rm_efn_core.h"

#define RME2_INF	RMA_EFN2_INFINITY
// branches of one loop: 16 in the usual instance, 52 in the one for calls over more than 15 helices (BIG; the reference
// allows 100, a descriptor of RMD_MAX_ELEMS elements cannot have more than fifty helices)
template< class Cand, int BIG = 0 >
struct rme2_ctx_t {
	static constexpr int	max_helix = BIG ? 52 : 16, stk = BIG ? 128 : 32;
	const rma_efn2data_t	*E;
	const Cand	*C;
	int	l_base;
	RMD_FN_MEMBER int	bc( int i ) const { return C->bc( i ); }
	RMD_FN_MEMBER int	bp( int i ) const { return C->bp( i ); }
};

RMD_FN int rme2_min( int a, int b ) { return a < b ? a : b; }

template< class X > RMD_FN int rme2_aupen( const X &x, int i, int j )	// ef2_aupen :1728
{
	return ( x.bc( i ) == RMA_BC_T || x.bc( j ) == RMA_BC_T ) ? x.E->auend : 0;
}
template< class X > RMD_FN int rme2_dangle( const X &x, int i, int j, int ip, int jp )	// ef2_dangle :1716
{
	return x.E->dangle[ x.bc( i ) ][ x.bc( j ) ][ x.bc( ip ) ][ jp ];
}
template< class X > RMD_FN int rme2_loginc( const X &x, int size )
{
	return x.E->loginc[ size < RMA_EFN_LOGINC ? size : RMA_EFN_LOGINC - 1 ];
}
#define RME2_T4( tab, a, b, c, d )	( x.E->tab[ x.bc( a ) ][ x.bc( b ) ][ x.bc( c ) ][ x.bc( d ) ] )

template< class X > RMD_FN int rme2_ibloop( const X &x, int i, int j, int ip, int jp )	// ef2_ibloop :1553
{
	const rma_efn2data_t	*e = x.E;
	const int	size1 = ip - i - 1, size2 = j - jp - 1, size = size1 + size2;
	if( size1 == 0 || size2 == 0 ){
		if( size == 1 )
			return RME2_T4( stack, i, j, ip, jp ) + e->bulge[ size ] + e->eparam[ 2 ];
		return e->bulge[ size > 30 ? 30 : size ] + ( size > 30 ? rme2_loginc( x, size ) : 0 ) + e->eparam[ 2 ] +
			rme2_aupen( x, i, j ) + rme2_aupen( x, jp, ip );
	}
	const int	lopsid = size1 > size2 ? size1 - size2 : size2 - size1;
	const int	pen = rme2_min( e->maxpen, lopsid * e->poppen[ rme2_min( 2, rme2_min( size1, size2 ) ) ] );
	const bool	gail = ( size1 == 1 || size2 == 1 ) && e->gail;
	if( size <= 30 ){
		if( size1 == 2 && size2 == 2 )
			return e->iloop22[ x.bc( i ) ][ x.bc( ip ) ][ x.bc( j ) ][ x.bc( jp ) ][ x.bc( i + 1 ) ][ x.bc( i + 2 ) ][ x.bc( j - 1 ) ][ x.bc( j - 2 ) ];
		if( size1 == 1 && size2 == 2 )
			return e->iloop21[ x.bc( i ) ][ x.bc( j ) ][ x.bc( i + 1 ) ][ x.bc( j - 1 ) ][ x.bc( jp + 1 ) ][ x.bc( ip ) ][ x.bc( jp ) ];
		if( size1 == 2 && size2 == 1 )
			return e->iloop21[ x.bc( jp ) ][ x.bc( ip ) ][ x.bc( jp + 1 ) ][ x.bc( ip - 1 ) ][ x.bc( i + 1 ) ][ x.bc( j ) ][ x.bc( i ) ];
		if( size == 2 )
			return e->iloop11[ x.bc( i ) ][ x.bc( i + 1 ) ][ x.bc( ip ) ][ x.bc( j ) ][ x.bc( j - 1 ) ][ x.bc( jp ) ];
	}
	int	energy = gail ?
		e->tstki[ x.bc( i ) ][ x.bc( j ) ][ 1 ][ 1 ] + e->tstki[ x.bc( jp ) ][ x.bc( ip ) ][ 1 ][ 1 ] :
		RME2_T4( tstki, i, j, i + 1, j - 1 ) + RME2_T4( tstki, jp, ip, jp + 1, ip - 1 );
	energy += size > 30 ? e->inter[ 30 ] + rme2_loginc( x, size ) : e->inter[ size ];
	return energy + e->eparam[ 3 ] + pen;
}

template< class X > RMD_FN int rme2_hploop( const X &x, int i, int j )	// ef2_hploop :1642
{
	const rma_efn2data_t	*e = x.E;
	const int	size = j - i - 1;
	const int	bi = x.bc( i ), bj = x.bc( j );
	int	energy;
	if( size > 30 )
		energy = RME2_T4( tstkh, i, j, i + 1, j - 1 ) + e->hairpin[ 30 ] + rme2_loginc( x, size ) + e->eparam[ 4 ];
	else if( size < 3 ){
		energy = e->hairpin[ size < 0 ? 0 : size ] + e->eparam[ 4 ];
		if( bi == 4 || bj == 4 )
			energy += 6;
	}else if( size == 4 ){
		int	tlink = 0;
		const int	key = bj * 3125 + x.bc( i + 4 ) * 625 + x.bc( i + 3 ) * 125 + x.bc( i + 2 ) * 25 + x.bc( i + 1 ) * 5 + bi;
		for( int c = 1; c <= e->ntloops && tlink == 0; c++ )
			if( key == e->tloop[ c ][ 0 ] )
				tlink = e->tloop[ c ][ 1 ];
		energy = RME2_T4( tstkh, i, j, i + 1, j - 1 ) + e->hairpin[ size ] + e->eparam[ 4 ] + tlink;
	}else if( size == 3 ){
		int	tlink = 0;
		const int	key = bj * 625 + x.bc( i + 3 ) * 125 + x.bc( i + 2 ) * 25 + x.bc( i + 1 ) * 5 + bi;
		for( int c = 1; c <= e->ntriloops && tlink == 0; c++ )
			if( key == e->triloop[ c ][ 0 ] )
				tlink = e->triloop[ c ][ 1 ];
		energy = e->hairpin[ size ] + e->eparam[ 4 ] + tlink + rme2_aupen( x, i, j );	// (no stacking term, :1679-1682)
	}else
		energy = RME2_T4( tstkh, i, j, i + 1, j - 1 ) + e->hairpin[ size ] + e->eparam[ 4 ];
	if( bi == RMA_BC_G && bj == RMA_BC_T && i > 1 && i < x.l_base )
		if( x.bc( i - 1 ) == RMA_BC_G && x.bc( i - 2 ) == RMA_BC_G )
			energy += e->gubonus;
	bool	polyc = true;
	for( int k = 1; k <= size && polyc; k++ )
		polyc = x.bc( i + k ) == RMA_BC_C;
	if( polyc )
		energy += size == 3 ? e->c3 : e->cint + size * e->cslope;
	return energy;
}

// Stacking of the branches of one loop (efn2.c:1214-1301 closed, :1349-1436 exterior).
// hx[h] = { 3' base, 5' base } as the reference's helix[h][0..1]; closed loops carry the
// closing stem as branch 0 and again as branch n.
template< class X > RMD_FN int rme2_branches( const X &x, int ( *hx )[ 2 ], int n, bool closed )
{
	int	coax[ X::max_helix + 1 ][ X::max_helix + 1 ];
	const int	l_base = x.l_base;
	const int	m = closed ? n : n - 1;		// last index of the diagonal in use
	for( int a = 0; a <= m; a++ )
		for( int b = 0; b <= m; b++ )
			coax[ a ][ b ] = 0;
	for( int h = 0; h < n; h++ ){
		bool	gap3, gap5;
		if( closed ){
			gap3 = hx[ h + 1 ][ 1 ] - hx[ h ][ 0 ] > 1;
			gap5 = h == 0 ? hx[ 0 ][ 1 ] - hx[ n - 1 ][ 0 ] > 1 : hx[ h ][ 1 ] - hx[ h - 1 ][ 0 ] > 1;
		}else{
			gap3 = h < n - 1 ? hx[ h + 1 ][ 1 ] - hx[ h ][ 0 ] > 1 : l_base - hx[ h ][ 0 ] >= 1;
			gap5 = h == 0 ? hx[ 0 ][ 1 ] > 1 : hx[ h ][ 1 ] - hx[ h - 1 ][ 0 ] >= 1;
		}
		int	v = 0;
		if( closed && gap3 && gap5 )
			v = RME2_T4( tstkm, hx[ h ][ 0 ], hx[ h ][ 1 ], hx[ h ][ 0 ] + 1, hx[ h ][ 1 ] - 1 );
		else{
			if( gap3 )
				v = rme2_min( 0, rme2_dangle( x, hx[ h ][ 0 ], hx[ h ][ 1 ], hx[ h ][ 0 ] + 1, 0 ) );
			if( gap5 )
				v += rme2_min( 0, rme2_dangle( x, hx[ h ][ 0 ], hx[ h ][ 1 ], hx[ h ][ 1 ] - 1, 1 ) );
		}
		coax[ h ][ h ] = v;
	}
	if( closed )
		coax[ n ][ n ] = coax[ 0 ][ 0 ];
	const int	npair = closed ? n : n - 1;
	for( int h = 0; h < npair; h++ ){
		const int	d = hx[ h + 1 ][ 1 ] - hx[ h ][ 0 ];
		int	v = coax[ h ][ h ] + coax[ h + 1 ][ h + 1 ];
		if( d == 1 )
			v = rme2_min( v, RME2_T4( coax, hx[ h ][ 1 ], hx[ h ][ 0 ], hx[ h + 1 ][ 1 ], hx[ h + 1 ][ 0 ] ) );
		else if( d == 2 ){
			bool	g5, g3;
			if( closed ){
				g5 = h != 0 ? hx[ h ][ 1 ] - hx[ h - 1 ][ 0 ] > 1 : hx[ 0 ][ 1 ] - hx[ n - 1 ][ 0 ] > 1;
				g3 = h != n - 1 ? hx[ h + 2 ][ 1 ] - hx[ h + 1 ][ 0 ] > 1 : hx[ 1 ][ 1 ] - hx[ 0 ][ 0 ] > 1;
			}else{
				g5 = h != 0 ? hx[ h ][ 1 ] - hx[ h - 1 ][ 0 ] > 1 : hx[ 0 ][ 1 ] > 1;
				g3 = h != n - 2 ? hx[ h + 2 ][ 1 ] - hx[ h + 1 ][ 0 ] > 1 : hx[ n - 1 ][ 0 ] < l_base;
			}
			if( g5 ){
				const int	t = h != 0 ?
					RME2_T4( tstackcoax, hx[ h ][ 0 ], hx[ h ][ 1 ], hx[ h ][ 0 ] + 1, hx[ h ][ 1 ] - 1 ) :
					RME2_T4( tstackcoax, hx[ h ][ 1 ], hx[ h ][ 0 ], hx[ h ][ 0 ] + 1, hx[ h ][ 1 ] - 1 );
				v = rme2_min( v, t + RME2_T4( coaxstack, hx[ h ][ 0 ] + 1, hx[ h ][ 1 ] - 1, hx[ h + 1 ][ 1 ], hx[ h + 1 ][ 0 ] ) );
			}
			if( g3 )
				v = rme2_min( v, RME2_T4( tstackcoax, hx[ h ][ 0 ] + 1, hx[ h + 1 ][ 0 ] + 1, hx[ h + 1 ][ 1 ], hx[ h + 1 ][ 0 ] ) +
					RME2_T4( coaxstack, hx[ h ][ 0 ], hx[ h ][ 1 ], hx[ h ][ 0 ] + 1, hx[ h + 1 ][ 0 ] + 1 ) );
		}
		coax[ h ][ h + 1 ] = v;
	}
	for( int h = 2; h <= m && h < n; h++ )
		for( int a = 0; a + h <= m; a++ ){
			int	v = coax[ a ][ a ] + coax[ a + 1 ][ a + h ];
			for( int b = 1; b < h; b++ )
				v = rme2_min( v, coax[ a ][ a + b ] + coax[ a + b + 1 ][ a + h ] );
			coax[ a ][ a + h ] = v;
		}
	if( closed )
		return rme2_min( coax[ 0 ][ n - 1 ], coax[ 1 ][ n ] );
	return n >= 1 ? coax[ 0 ][ n - 1 ] : 0;
}

template< class X > RMD_FN int rme2_efn2( const X &x )	// RM_efn2 :1103
{
	const rma_efn2data_t	*e = x.E;
	const int	l_base = x.l_base;
	int	stk_i[ X::stk ], stk_j[ X::stk ], sp = 0;
	int	hx[ X::max_helix + 1 ][ 2 ];
	int	energy = 0;
	stk_i[ sp ] = 0;
	stk_j[ sp ] = l_base;
	sp++;
	while( sp > 0 ){
		sp--;
		int	i = stk_i[ sp ], j = stk_j[ sp ];
		if( i < 0 || j < 0 || i > l_base || j > l_base )
			return RME2_INF;
		if( x.bp( i ) == j ){
			if( i >= j )
				return RME2_INF;
			for( bool again = true; again; ){
				again = false;
				while( x.bp( i + 1 ) == j - 1 ){
					energy += RME2_T4( stack, i, j, i + 1, j - 1 ) + e->eparam[ 1 ];
					i++;
					j--;
				}
				int	n_helix = 0, ip = 0, jp = 0;
				for( int k = i + 1; k < j; ){
					const int	q = x.bp( k );
					if( q > k ){
						n_helix++;
						ip = k;
						k = q + 1;
						jp = q;
					}else if( q == -1 )
						k++;
					else
						return RME2_INF;
				}
				if( n_helix == 0 )
					energy += rme2_hploop( x, i, j );
				else if( n_helix == 1 ){
					energy += rme2_ibloop( x, i, j, ip, jp );
					i = ip;
					j = jp;
					again = true;
				}else{
					n_helix++;
					if( n_helix >= X::max_helix )
						return RME2_INF;
					hx[ 0 ][ 0 ] = i;
					hx[ 0 ][ 1 ] = j;
					int	n_upn = 0;
					for( int h = 1; h < n_helix; h++ ){
						int	p = hx[ h - 1 ][ 0 ] + 1;
						while( x.bp( p ) == -1 )
							p++;
						const int	q = x.bp( p );
						energy += rme2_aupen( x, p, q );
						hx[ h ][ 1 ] = p;
						hx[ h ][ 0 ] = q;
						if( sp >= X::stk )
							return RME2_INF;
						stk_i[ sp ] = p;
						stk_j[ sp ] = q;
						sp++;
						n_upn += p - hx[ h - 1 ][ 0 ] - 1;
					}
					hx[ n_helix ][ 0 ] = hx[ 0 ][ 0 ];
					hx[ n_helix ][ 1 ] = hx[ 0 ][ 1 ];
					n_upn += hx[ n_helix ][ 1 ] - hx[ n_helix - 1 ][ 0 ] - 1;
					energy += e->efn2a + n_helix * e->efn2c;
					energy += n_upn <= 6 ? n_upn * e->efn2b :
						6 * e->efn2b + e->mbl_log[ n_upn < RMA_EFN_LOGINC ? n_upn : RMA_EFN_LOGINC - 1 ];
					energy += rme2_branches( x, hx, n_helix, true );
				}
			}
			continue;
		}
		// exterior loop
		int	n_helix = 0;
		while( i < l_base ){
			const int	q = x.bp( i );
			if( q != -1 ){
				n_helix++;
				i = q;
			}
			i++;
		}
		if( n_helix >= X::max_helix )
			return RME2_INF;
		int	p = 1;
		for( int h = 0; h < n_helix; h++ ){
			while( p <= l_base && x.bp( p ) == 0 )
				p++;
			const int	q = p <= l_base ? x.bp( p ) : -1;
			if( q < 0 )
				return RME2_INF;	// the reference leaves its arrays from here on
			energy += rme2_aupen( x, p, q );
			hx[ h ][ 1 ] = p;
			hx[ h ][ 0 ] = q;
			if( sp >= X::stk )
				return RME2_INF;
			stk_i[ sp ] = p;
			stk_j[ sp ] = q;
			sp++;
			p = q + 1;
		}
		energy += rme2_branches( x, hx, n_helix, false );
	}
	return energy;
}

// Energy of efn2 site k for the hit record w (do_sc_efnx, score.c:1672-1679, before the 0.01).
template< class Seq, int BIG = 0 >
RMD_FN int rme2_site_energy( const rmd_program_t *P, const rma_efn2data_t *E, const Seq *sq, const int32_t *w, int k,
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
	rme2_ctx_t< rme_cand_t<Seq>, BIG >	x;
	x.E = E;
	x.C = &c;
	x.l_base = c.len - 1;
	return rme2_efn2( x );
}

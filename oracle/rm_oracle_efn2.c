/*
 * rm_oracle_efn2.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar restatement of RM_efn2() and its helpers, /root/reference/src/efn2.c:1103-1777
 * (ef2_stack :1544, ef2_ibloop :1553, ef2_hploop :1642, ef2_dangle :1716, ef2_aupen :1728,
 * ef2_tstkm/coax/tstackcoax/coaxstack :1738-1777), over the tables of rma_efn2data_t.
 * Pinned by tests/test_efn2_oracle.py against the reference's own efn2_drv
 * (oracle/_ref/efn2_drv, built from the reference's sources as they lie).
 *
 * The reference's walk of the exterior loop tests "rm_basepr[ip]==0" for "unpaired"
 * (efn2.c:1337) although its arrays are 0-based with -1 for unpaired (efn2_drv.c:143-146):
 * unless every exterior helix starts exactly where that test leaves ip, it pushes an
 * interval ( ip, -1 ) and goes on to index its arrays with negative numbers.  Those inputs
 * have no defined answer; this restatement follows the same steps and returns
 * RMA_EFN2_INFINITY with *undefined = 1 the moment the reference would leave its arrays.
 */
#include <stdlib.h>
#include <string.h>
#include "rm_oracle.h"

#define INF	RMA_EFN2_INFINITY
#define MAXHELIX	100
#define STK	51
#define MIN( a, b )	( ( a ) < ( b ) ? ( a ) : ( b ) )

typedef struct {
	const rma_efn2data_t	*ed;
	const int	*bc, *bp;
	int	l_base;
} ctx_t;

static int stack_e( const ctx_t *x, int i, int j, int ip, int jp )		/* :1544 */
{
	return x->ed->stack[ x->bc[ i ] ][ x->bc[ j ] ][ x->bc[ ip ] ][ x->bc[ jp ] ] + x->ed->eparam[ 1 ];
}

static int aupen( const ctx_t *x, int i, int j )				/* :1728 */
{
	return ( x->bc[ i ] == RMA_BC_T || x->bc[ j ] == RMA_BC_T ) ? x->ed->auend : 0;
}

static int dangle( const ctx_t *x, int i, int j, int ip, int jp )		/* :1716 */
{
	return x->ed->dangle[ x->bc[ i ] ][ x->bc[ j ] ][ x->bc[ ip ] ][ jp ];
}

static int loginc( const ctx_t *x, int size )
{
	return x->ed->loginc[ size < RMA_EFN_LOGINC ? size : RMA_EFN_LOGINC - 1 ];
}

static int ibloop( const ctx_t *x, int i, int j, int ip, int jp )		/* :1553 */
{
	const rma_efn2data_t	*e = x->ed;
	const int	*bc = x->bc;
	int	size1 = ip - i - 1, size2 = j - jp - 1, size = size1 + size2;
	int	energy;

	if( size1 == 0 || size2 == 0 ){
		if( size == 1 )
			energy = e->stack[ bc[ i ] ][ bc[ j ] ][ bc[ ip ] ][ bc[ jp ] ] + e->bulge[ size ] + e->eparam[ 2 ];
		else if( size > 30 )
			energy = e->bulge[ 30 ] + loginc( x, size ) + e->eparam[ 2 ] + aupen( x, i, j ) + aupen( x, jp, ip );
		else
			energy = e->bulge[ size ] + e->eparam[ 2 ] + aupen( x, i, j ) + aupen( x, jp, ip );
		return energy;
	}
	{
		int	lopsid = abs( size1 - size2 );
		int	pen = MIN( e->maxpen, lopsid * e->poppen[ MIN( 2, MIN( size1, size2 ) ) ] );
		int	gail = ( size1 == 1 || size2 == 1 ) && e->gail;
		if( size > 30 ){
			if( gail )
				energy = e->tstki[ bc[ i ] ][ bc[ j ] ][ 1 ][ 1 ] + e->tstki[ bc[ jp ] ][ bc[ ip ] ][ 1 ][ 1 ];
			else
				energy = e->tstki[ bc[ i ] ][ bc[ j ] ][ bc[ i + 1 ] ][ bc[ j - 1 ] ] +
					e->tstki[ bc[ jp ] ][ bc[ ip ] ][ bc[ jp + 1 ] ][ bc[ ip - 1 ] ];
			energy += e->inter[ 30 ] + loginc( x, size ) + e->eparam[ 3 ] + pen;
		}else if( size1 == 2 && size2 == 2 )
			energy = e->iloop22[ bc[ i ] ][ bc[ ip ] ][ bc[ j ] ][ bc[ jp ] ][ bc[ i + 1 ] ][ bc[ i + 2 ] ][ bc[ j - 1 ] ][ bc[ j - 2 ] ];
		else if( size1 == 1 && size2 == 2 )
			energy = e->iloop21[ bc[ i ] ][ bc[ j ] ][ bc[ i + 1 ] ][ bc[ j - 1 ] ][ bc[ jp + 1 ] ][ bc[ ip ] ][ bc[ jp ] ];
		else if( size1 == 2 && size2 == 1 )
			energy = e->iloop21[ bc[ jp ] ][ bc[ ip ] ][ bc[ jp + 1 ] ][ bc[ ip - 1 ] ][ bc[ i + 1 ] ][ bc[ j ] ][ bc[ i ] ];
		else if( size == 2 )
			energy = e->iloop11[ bc[ i ] ][ bc[ i + 1 ] ][ bc[ ip ] ][ bc[ j ] ][ bc[ j - 1 ] ][ bc[ jp ] ];
		else{
			if( gail )
				energy = e->tstki[ bc[ i ] ][ bc[ j ] ][ 1 ][ 1 ] + e->tstki[ bc[ jp ] ][ bc[ ip ] ][ 1 ][ 1 ];
			else
				energy = e->tstki[ bc[ i ] ][ bc[ j ] ][ bc[ i + 1 ] ][ bc[ j - 1 ] ] +
					e->tstki[ bc[ jp ] ][ bc[ ip ] ][ bc[ jp + 1 ] ][ bc[ ip - 1 ] ];
			energy += e->inter[ size ] + e->eparam[ 3 ] + pen;
		}
	}
	return energy;
}

static int hploop( const ctx_t *x, int i, int j )				/* :1642 */
{
	const rma_efn2data_t	*e = x->ed;
	const int	*bc = x->bc;
	int	size = j - i - 1, energy, tlink, count, key, k;

	if( size > 30 )
		energy = e->tstkh[ bc[ i ] ][ bc[ j ] ][ bc[ i + 1 ] ][ bc[ j - 1 ] ] + e->hairpin[ 30 ] + loginc( x, size ) + e->eparam[ 4 ];
	else if( size < 3 ){
		energy = e->hairpin[ size ] + e->eparam[ 4 ];
		if( bc[ i ] == 4 || bc[ j ] == 4 )
			energy += 6;
	}else if( size == 4 ){
		tlink = 0;
		key = bc[ j ] * 3125 + bc[ i + 4 ] * 625 + bc[ i + 3 ] * 125 + bc[ i + 2 ] * 25 + bc[ i + 1 ] * 5 + bc[ i ];
		for( count = 1; count <= e->ntloops && tlink == 0; count++ )
			if( key == e->tloop[ count ][ 0 ] )
				tlink = e->tloop[ count ][ 1 ];
		energy = e->tstkh[ bc[ i ] ][ bc[ j ] ][ bc[ i + 1 ] ][ bc[ j - 1 ] ] + e->hairpin[ size ] + e->eparam[ 4 ] + tlink;
	}else if( size == 3 ){
		tlink = 0;
		key = bc[ j ] * 625 + bc[ i + 3 ] * 125 + bc[ i + 2 ] * 25 + bc[ i + 1 ] * 5 + bc[ i ];
		for( count = 1; count <= e->ntriloops && tlink == 0; count++ )
			if( key == e->triloop[ count ][ 0 ] )
				tlink = e->triloop[ count ][ 1 ];
		/* (the stacking term is computed and thrown away, :1679-1682) */
		energy = e->hairpin[ size ] + e->eparam[ 4 ] + tlink + aupen( x, i, j );
	}else
		energy = e->tstkh[ bc[ i ] ][ bc[ j ] ][ bc[ i + 1 ] ][ bc[ j - 1 ] ] + e->hairpin[ size ] + e->eparam[ 4 ];

	/* GU closure preceded by GG */
	if( bc[ i ] == RMA_BC_G && bc[ j ] == RMA_BC_T && i > 1 && i < x->l_base )
		if( bc[ i - 1 ] == RMA_BC_G && bc[ i - 2 ] == RMA_BC_G )
			energy += e->gubonus;
	/* poly-C loop */
	tlink = 1;
	for( k = 1; k <= size && tlink == 1; k++ )
		if( bc[ i + k ] != RMA_BC_C )
			tlink = 0;
	if( tlink == 1 )
		energy += size == 3 ? e->c3 : e->cint + size * e->cslope;
	return energy;
}

#define T4( tab, a, b, c, d )	( x->ed->tab[ x->bc[ a ] ][ x->bc[ b ] ][ x->bc[ c ] ][ x->bc[ d ] ] )

int rmo_efn2( const rma_efn2data_t *ed, const int *bcseq, const int *basepr, int l_base, int *undefined )
{
	static int	coax[ MAXHELIX + 1 ][ MAXHELIX + 1 ], helix[ MAXHELIX + 1 ][ 2 ];
	int	stk[ STK ][ 2 ], sp = 0;
	ctx_t	cx = { ed, bcseq, basepr, l_base }, *x = &cx;
	const int	*bp = basepr;
	int	energy = 0, i, j, ip = 0, jp = 0, k, h, h1, n_helix, n_upn;

	*undefined = 0;
	sp++;
	stk[ sp ][ 0 ] = 0;
	stk[ sp ][ 1 ] = l_base;
	for( ; ; ){
		if( sp == 0 )
			return energy;
		i = stk[ sp ][ 0 ];
		j = stk[ sp ][ 1 ];
		sp--;
		if( i < 0 || j < 0 || i > l_base || j > l_base ){
			*undefined = 1;
			return INF;
		}
		if( bp[ i ] == j ){
			/* a closed interval: walk down the helix, classify the loop it closes */
			if( i >= j ){	/* an exterior "helix" read off a 3' base: negative loop sizes follow */
				*undefined = 1;
				return INF;
			}
			int	again = 1;
			while( again ){
				again = 0;
				while( bp[ i + 1 ] == j - 1 ){
					energy += stack_e( x, i, j, i + 1, j - 1 );
					i++;
					j--;
				}
				n_helix = 0;
				for( k = i + 1; k < j; ){
					if( bp[ k ] > k ){
						n_helix++;
						ip = k;
						k = bp[ k ] + 1;
						jp = k - 1;
					}else if( bp[ k ] == -1 )
						k++;
					else{		/* not a nested structure: the reference would spin here */
						*undefined = 1;
						return INF;
					}
				}
				if( n_helix == 0 )
					energy += hploop( x, i, j );
				else if( n_helix == 1 ){
					energy += ibloop( x, i, j, ip, jp );
					i = ip;
					j = jp;
					again = 1;	/* "while( rm_basepr[i] == j )" holds again */
				}else{
					n_helix++;	/* include the closing stem */
					if( n_helix >= MAXHELIX )
						return INF;
					for( h = 0; h <= n_helix; h++ )
						memset( coax[ h ], 0, ( n_helix + 1 ) * sizeof( int ) );
					helix[ 0 ][ 0 ] = i;
					helix[ 0 ][ 1 ] = j;
					n_upn = 0;
					for( h = 1; h < n_helix; h++ ){
						ip = helix[ h - 1 ][ 0 ] + 1;
						while( bp[ ip ] == -1 )
							ip++;
						energy += aupen( x, ip, bp[ ip ] );
						helix[ h ][ 1 ] = ip;
						helix[ h ][ 0 ] = bp[ ip ];
						if( sp + 1 >= STK ){
							*undefined = 1;
							return INF;
						}
						sp++;
						stk[ sp ][ 0 ] = ip;
						stk[ sp ][ 1 ] = bp[ ip ];
						n_upn += ip - helix[ h - 1 ][ 0 ] - 1;
					}
					helix[ n_helix ][ 0 ] = helix[ 0 ][ 0 ];
					helix[ n_helix ][ 1 ] = helix[ 0 ][ 1 ];
					n_upn += helix[ n_helix ][ 1 ] - helix[ n_helix - 1 ][ 0 ] - 1;
					energy += ed->efn2a + n_helix * ed->efn2c;
					if( n_upn <= 6 )
						energy += n_upn * ed->efn2b;
					else
						energy += 6 * ed->efn2b + ed->mbl_log[ n_upn < RMA_EFN_LOGINC ? n_upn : RMA_EFN_LOGINC - 1 ];
					/* stacking of the branches, :1214-1301 */
					for( h1 = 0; h1 < n_helix; h1++ ){
						int	gap5 = h1 == 0 ? helix[ 0 ][ 1 ] - helix[ n_helix - 1 ][ 0 ] > 1 :
							helix[ h1 ][ 1 ] - helix[ h1 - 1 ][ 0 ] > 1;
						int	gap3 = helix[ h1 + 1 ][ 1 ] - helix[ h1 ][ 0 ] > 1;
						coax[ h1 ][ h1 ] = 0;
						if( gap3 && gap5 )
							coax[ h1 ][ h1 ] = T4( tstkm, helix[ h1 ][ 0 ], helix[ h1 ][ 1 ], helix[ h1 ][ 0 ] + 1, helix[ h1 ][ 1 ] - 1 );
						else{
							if( gap3 )
								coax[ h1 ][ h1 ] = MIN( 0, dangle( x, helix[ h1 ][ 0 ], helix[ h1 ][ 1 ], helix[ h1 ][ 0 ] + 1, 0 ) );
							if( gap5 )
								coax[ h1 ][ h1 ] += MIN( 0, dangle( x, helix[ h1 ][ 0 ], helix[ h1 ][ 1 ], helix[ h1 ][ 1 ] - 1, 1 ) );
						}
					}
					coax[ n_helix ][ n_helix ] = coax[ 0 ][ 0 ];
					for( h1 = 0; h1 < n_helix; h1++ ){
						int	d = helix[ h1 + 1 ][ 1 ] - helix[ h1 ][ 0 ];
						if( d == 1 )
							coax[ h1 ][ h1 + 1 ] = MIN( coax[ h1 ][ h1 ] + coax[ h1 + 1 ][ h1 + 1 ],
								T4( coax, helix[ h1 ][ 1 ], helix[ h1 ][ 0 ], helix[ h1 + 1 ][ 1 ], helix[ h1 + 1 ][ 0 ] ) );
						else if( d == 2 ){
							int	g5 = h1 != 0 ? helix[ h1 ][ 1 ] - helix[ h1 - 1 ][ 0 ] > 1 : helix[ 0 ][ 1 ] - helix[ n_helix - 1 ][ 0 ] > 1;
							int	g3 = h1 != n_helix - 1 ? helix[ h1 + 2 ][ 1 ] - helix[ h1 + 1 ][ 0 ] > 1 : helix[ 1 ][ 1 ] - helix[ 0 ][ 0 ] > 1;
							coax[ h1 ][ h1 + 1 ] = coax[ h1 ][ h1 ] + coax[ h1 + 1 ][ h1 + 1 ];
							if( g5 ){
								int	t = h1 != 0 ?
									T4( tstackcoax, helix[ h1 ][ 0 ], helix[ h1 ][ 1 ], helix[ h1 ][ 0 ] + 1, helix[ h1 ][ 1 ] - 1 ) :
									T4( tstackcoax, helix[ h1 ][ 1 ], helix[ h1 ][ 0 ], helix[ h1 ][ 0 ] + 1, helix[ h1 ][ 1 ] - 1 );
								coax[ h1 ][ h1 + 1 ] = MIN( coax[ h1 ][ h1 + 1 ], t +
									T4( coaxstack, helix[ h1 ][ 0 ] + 1, helix[ h1 ][ 1 ] - 1, helix[ h1 + 1 ][ 1 ], helix[ h1 + 1 ][ 0 ] ) );
							}
							if( g3 )
								coax[ h1 ][ h1 + 1 ] = MIN( coax[ h1 ][ h1 + 1 ],
									T4( tstackcoax, helix[ h1 ][ 0 ] + 1, helix[ h1 + 1 ][ 0 ] + 1, helix[ h1 + 1 ][ 1 ], helix[ h1 + 1 ][ 0 ] ) +
									T4( coaxstack, helix[ h1 ][ 0 ], helix[ h1 ][ 1 ], helix[ h1 ][ 0 ] + 1, helix[ h1 + 1 ][ 0 ] + 1 ) );
						}else
							coax[ h1 ][ h1 + 1 ] = coax[ h1 ][ h1 ] + coax[ h1 + 1 ][ h1 + 1 ];
					}
					for( h = 2; h < n_helix; h++ ){
						int	a, b;
						for( a = 0; a + h <= n_helix; a++ ){
							coax[ a ][ a + h ] = coax[ a ][ a ] + coax[ a + 1 ][ a + h ];
							for( b = 1; b < h; b++ )
								coax[ a ][ a + h ] = MIN( coax[ a ][ a + h ], coax[ a ][ a + b ] + coax[ a + b + 1 ][ a + h ] );
						}
					}
					energy += MIN( coax[ 0 ][ n_helix - 1 ], coax[ 1 ][ n_helix ] );
				}
			}
			continue;
		}
		/* exterior loop, :1316-1440 */
		n_helix = 0;
		while( i < l_base ){
			if( bp[ i ] != -1 ){
				n_helix++;
				i = bp[ i ];
			}
			i++;
		}
		if( n_helix >= MAXHELIX )
			return INF;
		for( h = 0; h < n_helix; h++ )
			memset( coax[ h ], 0, n_helix * sizeof( int ) );
		ip = 1;
		for( h = 0; h < n_helix; h++ ){
			while( ip <= l_base && bp[ ip ] == 0 )
				ip++;
			if( ip > l_base || bp[ ip ] < 0 ){	/* the reference leaves its arrays from here on */
				*undefined = 1;
				return INF;
			}
			energy += aupen( x, ip, bp[ ip ] );
			helix[ h ][ 1 ] = ip;
			helix[ h ][ 0 ] = bp[ ip ];
			if( sp + 1 >= STK ){
				*undefined = 1;
				return INF;
			}
			sp++;
			stk[ sp ][ 0 ] = ip;
			stk[ sp ][ 1 ] = bp[ ip ];
			ip = bp[ ip ] + 1;
		}
		for( h1 = 0; h1 < n_helix; h1++ ){
			coax[ h1 ][ h1 ] = 0;
			if( h1 < n_helix - 1 ? helix[ h1 + 1 ][ 1 ] - helix[ h1 ][ 0 ] > 1 : l_base - helix[ h1 ][ 0 ] >= 1 )
				coax[ h1 ][ h1 ] = MIN( 0, dangle( x, helix[ h1 ][ 0 ], helix[ h1 ][ 1 ], helix[ h1 ][ 0 ] + 1, 0 ) );
			if( h1 == 0 ? helix[ 0 ][ 1 ] > 1 : helix[ h1 ][ 1 ] - helix[ h1 - 1 ][ 0 ] >= 1 )
				coax[ h1 ][ h1 ] += MIN( 0, dangle( x, helix[ h1 ][ 0 ], helix[ h1 ][ 1 ], helix[ h1 ][ 1 ] - 1, 1 ) );
		}
		for( h1 = 0; h1 < n_helix - 1; h1++ ){
			int	d = helix[ h1 + 1 ][ 1 ] - helix[ h1 ][ 0 ];
			if( d == 1 )
				coax[ h1 ][ h1 + 1 ] = MIN( coax[ h1 ][ h1 ] + coax[ h1 + 1 ][ h1 + 1 ],
					T4( coax, helix[ h1 ][ 1 ], helix[ h1 ][ 0 ], helix[ h1 + 1 ][ 1 ], helix[ h1 + 1 ][ 0 ] ) );
			else if( d == 2 ){
				int	g5 = h1 != 0 ? helix[ h1 ][ 1 ] - helix[ h1 - 1 ][ 0 ] > 1 : helix[ 0 ][ 1 ] > 1;
				int	g3 = h1 != n_helix - 2 ? helix[ h1 + 2 ][ 1 ] - helix[ h1 + 1 ][ 0 ] > 1 : helix[ n_helix - 1 ][ 0 ] < l_base;
				coax[ h1 ][ h1 + 1 ] = coax[ h1 ][ h1 ] + coax[ h1 + 1 ][ h1 + 1 ];
				if( g5 ){
					int	t = h1 != 0 ?
						T4( tstackcoax, helix[ h1 ][ 0 ], helix[ h1 ][ 1 ], helix[ h1 ][ 0 ] + 1, helix[ h1 ][ 1 ] - 1 ) :
						T4( tstackcoax, helix[ h1 ][ 1 ], helix[ h1 ][ 0 ], helix[ h1 ][ 0 ] + 1, helix[ h1 ][ 1 ] - 1 );
					coax[ h1 ][ h1 + 1 ] = MIN( coax[ h1 ][ h1 + 1 ], t +
						T4( coaxstack, helix[ h1 ][ 0 ] + 1, helix[ h1 ][ 1 ] - 1, helix[ h1 + 1 ][ 1 ], helix[ h1 + 1 ][ 0 ] ) );
				}
				if( g3 )
					coax[ h1 ][ h1 + 1 ] = MIN( coax[ h1 ][ h1 + 1 ],
						T4( tstackcoax, helix[ h1 ][ 0 ] + 1, helix[ h1 + 1 ][ 0 ] + 1, helix[ h1 + 1 ][ 1 ], helix[ h1 + 1 ][ 0 ] ) +
						T4( coaxstack, helix[ h1 ][ 0 ], helix[ h1 ][ 1 ], helix[ h1 ][ 0 ] + 1, helix[ h1 + 1 ][ 0 ] + 1 ) );
			}else
				coax[ h1 ][ h1 + 1 ] = coax[ h1 ][ h1 ] + coax[ h1 + 1 ][ h1 + 1 ];
		}
		for( h = 2; h < n_helix; h++ ){
			int	a, b;
			for( a = 0; a + h < n_helix; a++ ){
				coax[ a ][ a + h ] = coax[ a ][ a ] + coax[ a + 1 ][ a + h ];
				for( b = 1; b < h; b++ )
					coax[ a ][ a + h ] = MIN( coax[ a ][ a + h ], coax[ a ][ a + b ] + coax[ a + b + 1 ][ a + h ] );
			}
		}
		if( n_helix >= 1 )
			energy += coax[ 0 ][ n_helix - 1 ];
	}
}

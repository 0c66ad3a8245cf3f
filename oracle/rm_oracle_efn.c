/*
 * rm_oracle_efn.c -- TEST INFRASTRUCTURE (see rm_oracle.h).
 *
 * CPU restatement of the reference's nearest neighbour energy function:
 * table readers (RM_getefndata and helpers, /root/reference/src/efn.c:157-918)
 * and RM_efn with ef_stack / ef_ibloop / ef_hploop / ef_dangle / ef_aupen
 * (efn.c:1162-1607).  Energies are integers in 1/100 kcal/mol.
 * Checked against the reference's own efn_drv binary (oracle/_ref) in
 * tests/test_efn_oracle.py.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "rm_oracle.h"

#define	UNDEF	(-1)
#define	INF	RMA_EFN_INFINITY
#define	MIN(a,b)	((a)<(b)?(a):(b))
#define	MAX(a,b)	((a)>(b)?(a):(b))
#define	NINT(x)		((int)((x)>=0?(x)+.5:(x)-.5))		/* efn.c:25 */
#define	WC(i,j)		((i)+(j) == 3)				/* efn.c:27 */
#define	GU(i,j)		((i)==RMA_BC_G && (j)==RMA_BC_T)	/* efn.c:28 */

/* ------------------------------------------------------------------ readers */
static	int	skipto( FILE *fp, const char *str, char *line, int s_line )	/* efn.c:954 */
{
	while( fgets( line, s_line, fp ) )
		if( strstr( line, str ) )
			return( 1 );
	return( 0 );
}

/* split( line, fields, " \t\n" ), split.c:26-48, into a fixed table */
static	int	fields_of( char *line, char *fields[], int maxf )
{
	int	nf = 0;
	char	*sp = line;

	for( ; ; ){
		sp += strspn( sp, " \t\n" );
		if( !*sp || nf == maxf )
			return( nf );
		fields[ nf++ ] = sp;
		sp += strcspn( sp, " \t\n" );
		if( *sp )
			*sp++ = '\0';
	}
}

static	int	cell( const char *f )
{
	return( *f == '.' ? INF : NINT( 100.0 * atof( f ) ) );
}

static	FILE	*dopen( const char *dir, const char *name )
{
	char	path[ 1024 ];
	FILE	*fp;

	snprintf( path, sizeof( path ), "%s/%s", dir, name );
	if( ( fp = fopen( path, "r" ) ) == NULL )
		fprintf( stderr, "rmo_load_efndata: can't read '%s'.\n", path );
	return( fp );
}

static	int	packloop( const char *loop )	/* efn.c:920 */
{
	int	i, num = 0;

	for( i = ( int )strlen( loop ) - 1; i >= 0; i-- ){
		int	bc;
		switch( loop[ i ] ){
		case 'A' : case 'a' : bc = RMA_BC_A; break;
		case 'C' : case 'c' : bc = RMA_BC_C; break;
		case 'G' : case 'g' : bc = RMA_BC_G; break;
		case 'T' : case 't' : case 'U' : case 'u' : bc = RMA_BC_T; break;
		default : return( -1 );
		}
		num = ( num << 3 ) + bc;
	}
	return( num );
}

static	int	get_loops( const char *dir, const char *name, int maxn, int32_t tab[][ 2 ], int32_t *n )	/* :209, :249 */
{
	FILE	*fp = dopen( dir, name );
	char	line[ 256 ], loop[ 256 ] = "";
	float	energy = 0;
	int	t = 0;

	if( fp == NULL )
		return( 0 );
	if( !skipto( fp, "---", line, sizeof( line ) ) ){
		fclose( fp );
		*n = 0;
		return( 0 );
	}
	for( ; fgets( line, sizeof( line ), fp ); t++ ){
		sscanf( line, "%255s %f", loop, &energy );
		if( t < maxn ){
			tab[ t ][ 0 ] = packloop( loop );
			tab[ t ][ 1 ] = NINT( 100.0 * energy );
		}
	}
	fclose( fp );
	*n = t > maxn ? maxn : t;
	return( 1 );
}

static	int	get_stack( const char *dir, const char *name, int32_t st[ 5 ][ 5 ][ 5 ][ 5 ], int defval )	/* :568 */
{
	FILE	*fp = dopen( dir, name );
	char	line[ 256 ], *f[ 16 ];
	int	v1, v2, v3, v4, k, nf;

	if( fp == NULL )
		return( 0 );
	for( v1 = 0; v1 < 5; v1++ ) for( v2 = 0; v2 < 5; v2++ ) for( v3 = 0; v3 < 5; v3++ ) for( v4 = 0; v4 < 5; v4++ )
		st[ v1 ][ v2 ][ v3 ][ v4 ] = defval;
	for( v1 = 0; v1 < 4; v1++ ){
		if( !skipto( fp, "<--", line, sizeof( line ) ) ){
			fclose( fp );
			return( 0 );
		}
		for( v3 = 0; v3 < 4; v3++ ){
			if( !fgets( line, sizeof( line ), fp ) )
				break;
			nf = fields_of( line, f, 16 );
			for( k = 0; k < nf; k++ )
				st[ v1 ][ k / 4 ][ v3 ][ k % 4 ] = cell( f[ k ] );
		}
	}
	fclose( fp );
	return( 1 );
}

static	int	get_misc( const char *dir, rma_efndata_t *ed )	/* getmiscloop :290 */
{
	FILE	*fp = dopen( dir, "miscloop.dat" );
	char	line[ 256 ];
	float	f1 = 0, f2 = 0, f3 = 0, f4 = 0;
	int32_t	*terms[ 6 ];
	int	k, rv = 0;

	if( fp == NULL )
		return( 0 );
#define	NEXT()	( skipto( fp, "-->", line, sizeof( line ) ) && fgets( line, sizeof( line ), fp ) )
	if( !NEXT() ) goto DONE;
	sscanf( line, "%f", &ed->prelog );
	ed->prelog *= 10.0;
	if( !NEXT() ) goto DONE;
	sscanf( line, "%f", &f1 );
	ed->maxpen = NINT( 100.0 * f1 );
	if( !NEXT() ) goto DONE;
	sscanf( line, "%f %f %f %f", &f1, &f2, &f3, &f4 );
	ed->poppen[ 0 ] = 0;
	ed->poppen[ 1 ] = NINT( 100.0 * f1 );
	ed->poppen[ 2 ] = NINT( 100.0 * f2 );
	ed->poppen[ 3 ] = NINT( 100.0 * f3 );
	ed->poppen[ 4 ] = NINT( 100.0 * f4 );
	ed->eparam[ 6 ] = 30;
	ed->eparam[ 7 ] = 30;
	if( !NEXT() ) goto DONE;
	sscanf( line, "%f %f %f", &f1, &f2, &f3 );
	ed->eparam[ 4 ] = NINT( 100.0 * f1 );
	ed->eparam[ 5 ] = NINT( 100.0 * f2 );
	ed->eparam[ 8 ] = NINT( 100.0 * f3 );
	rv = 1;
	if( !NEXT() ) goto DONE;	/* efn2 multibranch terms: absent in old files */
	terms[ 0 ] = &ed->auend; terms[ 1 ] = &ed->gubonus; terms[ 2 ] = &ed->cslope;
	terms[ 3 ] = &ed->cint; terms[ 4 ] = &ed->c3; terms[ 5 ] = &ed->init;
	for( k = 0; k < 6; k++ ){
		if( !NEXT() ){ rv = 0; goto DONE; }
		sscanf( line, "%f", &f1 );
		*terms[ k ] = NINT( 100.0 * f1 );
	}
	if( !NEXT() ){ rv = 0; goto DONE; }
	sscanf( line, "%d", &ed->gail );
#undef NEXT
DONE : ;
	fclose( fp );
	return( rv );
}

int	rmo_load_efndata( const char *dir, rma_efndata_t *ed )	/* RM_getefndata :157 */
{
	FILE	*fp;
	char	line[ 256 ], *f[ 24 ];
	int	v1, v2, v3, v4, v5, v6, k, nf, worst, rval = 1;

	memset( ed, 0, sizeof( *ed ) );
	if( dir == NULL || *dir == '\0' )
		return( 0 );
	if( !get_loops( dir, "tloop.dat", 100, ed->tloops, &ed->ntloops ) ) rval = 0;
	if( !get_loops( dir, "triloop.dat", 50, ed->triloops, &ed->ntriloops ) ) rval = 0;
	if( !get_misc( dir, ed ) ) rval = 0;

	if( ( fp = dopen( dir, "dangle.dat" ) ) == NULL )	/* getdangle :467 */
		rval = 0;
	else{
		for( v4 = 0; v4 < 2; v4++ ){
			for( v1 = 0; v1 < 4; v1++ ){
				if( !skipto( fp, "<--", line, sizeof( line ) ) || !fgets( line, sizeof( line ), fp ) ){
					rval = 0;
					v4 = 2;
					break;
				}
				nf = fields_of( line, f, 16 );
				for( k = 0; k < nf; k++ )
					ed->dangle[ v1 ][ k / 4 ][ k % 4 ][ v4 ] = cell( f[ k ] );
			}
		}
		fclose( fp );
	}

	if( ( fp = dopen( dir, "loop.dat" ) ) == NULL )		/* getibhloop :517 */
		rval = 0;
	else{
		if( !skipto( fp, "---", line, sizeof( line ) ) )
			rval = 0;
		else for( k = 1; k <= RMA_EFN_MAXLOOP; k++ ){
			if( !fgets( line, sizeof( line ), fp ) )
				break;
			if( fields_of( line, f, 4 ) < 4 )
				continue;
			ed->inter[ k ] = cell( f[ 1 ] );
			ed->bulge[ k ] = cell( f[ 2 ] );
			ed->hairpin[ k ] = cell( f[ 3 ] );
		}
		fclose( fp );
	}

	if( !get_stack( dir, "stack.dat", ed->stack, INF ) ) rval = 0;
	for( v1 = 0; v1 < 4; v1++ ) for( v2 = 0; v2 < 4; v2++ ) for( v3 = 0; v3 < 4; v3++ ) for( v4 = 0; v4 < 4; v4++ )
		if( ed->stack[ v1 ][ v2 ][ v3 ][ v4 ] != ed->stack[ v4 ][ v3 ][ v2 ][ v1 ] )
			rval = 0;				/* stacktest :622 */
	if( !get_stack( dir, "tstackh.dat", ed->tstkh, 0 ) ) rval = 0;
	if( !get_stack( dir, "tstacki.dat", ed->tstki, 0 ) ) rval = 0;

	if( ( fp = dopen( dir, "sint2.dat" ) ) == NULL )	/* getsymint :646 */
		rval = 0;
	else{
		if( !skipto( fp, "<--", line, sizeof( line ) ) )
			rval = 0;
		else for( v1 = 0; v1 < 6; v1++ ){
			if( !skipto( fp, "<--", line, sizeof( line ) ) ){
				rval = 0;
				break;
			}
			for( v3 = 0; v3 < 4; v3++ ){
				if( !fgets( line, sizeof( line ), fp ) )
					break;
				nf = fields_of( line, f, 24 );
				for( k = 0; k < nf; k++ )
					ed->sint2[ v1 ][ k / 4 ][ v3 ][ k % 4 ] = NINT( 100.0 * atof( f[ k ] ) );
			}
		}
		fclose( fp );
		for( v1 = 0; v1 < 6; v1++ ) for( v2 = 0; v2 < 6; v2++ ){
			worst = -999;
			for( v3 = 0; v3 < 4; v3++ ) for( v4 = 0; v4 < 4; v4++ )
				worst = MAX( worst, ed->sint2[ v1 ][ v2 ][ v3 ][ v4 ] );
			for( v3 = 0; v3 < 5; v3++ ){
				ed->sint2[ v1 ][ v2 ][ v3 ][ 4 ] = worst;
				ed->sint2[ v1 ][ v2 ][ 4 ][ v3 ] = worst;
			}
		}
	}
	if( ( fp = dopen( dir, "sint4.dat" ) ) == NULL )
		rval = 0;
	else{
		if( !skipto( fp, "<--", line, sizeof( line ) ) )
			rval = 0;
		else for( v1 = 0; v1 < 6 && rval; v1++ ) for( v2 = 0; v2 < 6; v2++ ){
			if( !skipto( fp, "<--", line, sizeof( line ) ) ){
				rval = 0;
				break;
			}
			for( v3 = 0; v3 < 4; v3++ ) for( v4 = 0; v4 < 4; v4++ ){
				if( !fgets( line, sizeof( line ), fp ) )
					break;
				nf = fields_of( line, f, 16 );
				for( k = 0; k < nf; k++ )
					ed->sint4[ v1 ][ v2 ][ v3 ][ v4 ][ k / 4 ][ k % 4 ] = NINT( 100.0 * atof( f[ k ] ) );
			}
		}
		fclose( fp );
		for( v1 = 0; v1 < 6; v1++ ) for( v2 = 0; v2 < 6; v2++ ){
			worst = -999;
			for( v3 = 0; v3 < 4; v3++ ) for( v4 = 0; v4 < 4; v4++ ) for( v5 = 0; v5 < 4; v5++ ) for( v6 = 0; v6 < 4; v6++ )
				worst = MAX( worst, ed->sint4[ v1 ][ v2 ][ v3 ][ v4 ][ v5 ][ v6 ] );
			for( v3 = 0; v3 < 5; v3++ ) for( v4 = 0; v4 < 5; v4++ ) for( v5 = 0; v5 < 5; v5++ ){
				ed->sint4[ v1 ][ v2 ][ v3 ][ v4 ][ v5 ][ 4 ] = worst;
				ed->sint4[ v1 ][ v2 ][ v3 ][ v4 ][ 4 ][ v5 ] = worst;
				ed->sint4[ v1 ][ v2 ][ v3 ][ 4 ][ v4 ][ v5 ] = worst;
				ed->sint4[ v1 ][ v2 ][ 4 ][ v3 ][ v4 ][ v5 ] = worst;
			}
		}
	}

	if( ( fp = dopen( dir, "asint1x2.dat" ) ) == NULL )	/* getasymint :826 */
		rval = 0;
	else{
		for( v1 = 0; v1 < 6; v1++ ) for( v2 = 0; v2 < 6; v2++ ) for( v3 = 0; v3 < 5; v3++ ) for( v4 = 0; v4 < 5; v4++ ) for( v5 = 0; v5 < 5; v5++ )
			ed->asint1x2[ v1 ][ v2 ][ v3 ][ v4 ][ v5 ] = INF;
		if( !skipto( fp, "<--", line, sizeof( line ) ) )
			rval = 0;
		else for( v1 = 0; v1 < 6 && rval; v1++ ) for( v5 = 0; v5 < 4; v5++ ){
			if( !skipto( fp, "<--", line, sizeof( line ) ) ){
				rval = 0;
				break;
			}
			for( v3 = 0; v3 < 4; v3++ ){
				if( !fgets( line, sizeof( line ), fp ) )
					break;
				nf = fields_of( line, f, 24 );
				for( k = 0; k < nf; k++ )
					ed->asint1x2[ v1 ][ k / 4 ][ v3 ][ k % 4 ][ v5 ] = NINT( 100.0 * atof( f[ k ] ) );
			}
		}
		fclose( fp );
	}
	for( k = 0; k < RMA_EFN_LOGINC; k++ )
		ed->loginc[ k ] = k > 30 ? NINT( ed->prelog * log( k / 30.0 ) ) : 0;
	return( rval );
}

/* ------------------------------------------------------------------ energy */
typedef struct efn_t {
	const rma_efndata_t	*ed;
	const int	*bcseq, *basepr;
	int	l_base;
} efn_t;

static	int	big_loop( const efn_t *e, int size )
{
	if( size < RMA_EFN_LOGINC )
		return( e->ed->loginc[ size ] );
	return( NINT( e->ed->prelog * log( size / 30.0 ) ) );
}

static	int	ef_dangle( const efn_t *e, int i, int j, int ip, int jp )	/* :1582 */
{
	return( e->ed->dangle[ e->bcseq[ i ] ][ e->bcseq[ j ] ][ e->bcseq[ ip ] ][ jp ] );
}

static	int	ef_aupen( const efn_t *e, int i, int j )	/* :1591 */
{
	int	bi = e->bcseq[ i ], bj = e->bcseq[ j ];
	int	pen = ( bi == RMA_BC_A && bj == RMA_BC_T ) || ( bi == RMA_BC_G && bj == RMA_BC_T ) ||
			( bi == RMA_BC_T && ( bj == RMA_BC_A || bj == RMA_BC_G ) );
	return( pen * e->ed->auend );
}

static	int	ef_stack( const efn_t *e, int i, int j )	/* :1337 */
{
	const int	*s = e->bcseq;

	if( i == e->l_base || j == e->l_base + 1 )
		return( INF );
	return( e->ed->stack[ s[ i ] ][ s[ j ] ][ s[ i + 1 ] ][ s[ j - 1 ] ] + e->ed->eparam[ 0 ] );
}

static	int	ef_ibloop( const efn_t *e, int i, int j, int ip, int jp )	/* :1351 */
{
	const rma_efndata_t	*ed = e->ed;
	const int	*s = e->bcseq;
	int	size, size1, size2, min4, lopsid, loginc, lf, rt, rval = 0;

	if( ( i <= e->l_base && ip > e->l_base ) || ( jp <= e->l_base && j > e->l_base ) )
		return( INF );
	size1 = ip - i - 1;
	size2 = j - jp - 1;
	size = size1 + size2;
	min4 = MIN( 4, MIN( size1, size2 ) );
	if( size1 == 0 || size2 == 0 ){
		if( size == 1 )
			rval += ed->stack[ s[ i ] ][ s[ j ] ][ s[ ip ] ][ s[ jp ] ] + ed->bulge[ size ] + ed->eparam[ 1 ];
		else{
			rval += ef_aupen( e, i, j ) + ef_aupen( e, ip, jp );
			if( size > 30 )
				rval += ed->bulge[ 30 ] + big_loop( e, size ) + ed->eparam[ 1 ];
			else
				rval += ed->bulge[ size ] + ed->eparam[ 1 ];
		}
		return( rval );
	}
	lopsid = abs( size1 - size2 );
	if( size > 30 ){
		loginc = big_loop( e, size );
		if( ( size1 == 1 || size2 == 1 ) && ed->gail == 1 )
			rval += ed->tstki[ s[ i ] ][ s[ j ] ][ RMA_BC_A ][ RMA_BC_A ] +
				ed->tstki[ s[ jp ] ][ s[ ip ] ][ RMA_BC_A ][ RMA_BC_A ];
		else
			rval += ed->tstki[ s[ i ] ][ s[ j ] ][ s[ i + 1 ] ][ s[ j - 1 ] ] +
				ed->tstki[ s[ jp ] ][ s[ ip ] ][ s[ jp + 1 ] ][ s[ ip - 1 ] ];
		rval += ed->inter[ 30 ] + loginc + ed->eparam[ 2 ] + MIN( ed->maxpen, lopsid * ed->poppen[ min4 ] );
	}else if( lopsid == 1 && size == 3 ){
		if( size1 < size2 ){
			if( WC( s[ i ], s[ j ] ) ) lf = s[ i ];
			else if( GU( s[ i ], s[ j ] ) ) lf = 4;
			else if( GU( s[ j ], s[ i ] ) ) lf = 5;
			else return( INF );
			if( WC( s[ ip ], s[ jp ] ) ) rt = s[ ip ];
			else if( GU( s[ ip ], s[ jp ] ) ) rt = 4;
			else if( GU( s[ jp ], s[ ip ] ) ) rt = 5;
			else return( INF );
			rval += ed->eparam[ 2 ] + ed->asint1x2[ lf ][ rt ][ s[ i + 1 ] ][ s[ j - 1 ] ][ s[ jp + 1 ] ];
		}else{
			if( WC( s[ jp ], s[ ip ] ) ) lf = s[ jp ];
			else if( GU( s[ jp ], s[ ip ] ) ) lf = 4;
			else if( GU( s[ ip ], s[ jp ] ) ) lf = 5;
			else return( INF );
			if( WC( s[ j ], s[ i ] ) ) rt = s[ j ];
			else if( GU( s[ j ], s[ i ] ) ) rt = 4;
			else if( GU( s[ i ], s[ j ] ) ) rt = 5;
			else return( INF );
			rval += ed->eparam[ 2 ] + ed->asint1x2[ lf ][ rt ][ s[ jp + 1 ] ][ s[ ip - 1 ] ][ s[ i + 1 ] ];
		}
	}else if( lopsid == 0 && size <= 4 ){
		if( WC( s[ i ], s[ j ] ) ) lf = s[ i ];
		else if( GU( s[ i ], s[ j ] ) || GU( s[ j ], s[ i ] ) ) lf = s[ i ] + 2;
		else return( INF );
		if( WC( s[ ip ], s[ jp ] ) ) rt = s[ ip ];
		else if( GU( s[ ip ], s[ jp ] ) || GU( s[ jp ], s[ ip ] ) ) rt = s[ ip ] + 2;
		else return( INF );
		if( size == 2 )
			rval += ed->eparam[ 2 ] + ed->sint2[ lf ][ rt ][ s[ i + 1 ] ][ s[ j - 1 ] ];
		else if( size == 4 )
			rval += ed->eparam[ 2 ] + ed->sint4[ lf ][ rt ][ s[ i + 1 ] ][ s[ j - 1 ] ][ s[ ip - 1 ] ][ s[ jp + 1 ] ];
	}else{
		if( ( size1 == 1 || size2 == 1 ) && ed->gail == 1 )
			rval += ed->tstki[ s[ i ] ][ s[ j ] ][ RMA_BC_A ][ RMA_BC_A ] +
				ed->tstki[ s[ jp ] ][ s[ ip ] ][ RMA_BC_A ][ RMA_BC_A ];
		else
			rval += ed->tstki[ s[ i ] ][ s[ j ] ][ s[ i + 1 ] ][ s[ j - 1 ] ] +
				ed->tstki[ s[ jp ] ][ s[ ip ] ][ s[ jp + 1 ] ][ s[ ip - 1 ] ];
		rval += ed->eparam[ 2 ] + ed->inter[ size > 30 ? 30 : size ] +
			MIN( ed->maxpen, lopsid * ed->poppen[ min4 ] );
	}
	return( rval );
}

static	int	ef_hploop( const efn_t *e, int i, int j )	/* :1498 */
{
	const rma_efndata_t	*ed = e->ed;
	const int	*s = e->bcseq;
	int	size, ccnt, k, key, lval, rval = 0;

	if( i <= e->l_base && j > e->l_base )
		return( INF );
	size = j - i - 1;
	for( ccnt = 0, k = i + i; k < j; k++ ){		/* (sic) :1511 starts at i + i */
		if( s[ k ] == RMA_BC_C )
			ccnt++;
		else
			break;
	}
	if( ccnt == size )
		rval = size == 3 ? ed->c3 : ed->cint + size * ed->cslope;
	if( i > 1 && j <= e->l_base ){
		if( s[ i ] == RMA_BC_G && s[ i - 1 ] == RMA_BC_G && s[ i - 2 ] == RMA_BC_G && s[ j ] == RMA_BC_T )
			rval += ed->gubonus;
	}
	if( size <= 3 ){
		lval = 0;
		if( size == 3 ){
			key = s[ i + size + 1 ];
			for( k = size + 1; k >= 0; k-- )
				key = ( key << 3 ) + s[ i + k ];
			for( k = 0; k < ed->ntriloops; k++ ){
				if( ed->triloops[ k ][ 0 ] == key ){
					lval = ed->triloops[ k ][ 1 ];
					break;
				}
			}
		}
		rval += ed->hairpin[ size ] + ed->eparam[ 3 ] + ef_aupen( e, i, j ) + lval;
	}else if( size <= 30 ){
		lval = 0;
		if( size == 4 ){
			key = s[ i + size + 1 ];
			for( k = size; k >= 0; k-- )
				key = ( key << 3 ) + s[ i + k ];
			for( k = 0; k < ed->ntloops; k++ ){
				if( ed->tloops[ k ][ 0 ] == key ){
					lval = ed->tloops[ k ][ 1 ];
					break;
				}
			}
		}
		rval += ed->tstkh[ s[ i ] ][ s[ j ] ][ s[ i + 1 ] ][ s[ j - 1 ] ] + ed->hairpin[ size ] + ed->eparam[ 3 ] + lval;
	}else
		rval += ed->tstkh[ s[ i ] ][ s[ j ] ][ s[ i + 1 ] ][ s[ j - 1 ] ] + ed->hairpin[ 30 ] +
			big_loop( e, size ) + ed->eparam[ 3 ];
	return( rval );
}

static	int	efn_rec( const efn_t *E, int i, int j, int open )	/* RM_efn :1162 */
{
	const rma_efndata_t	*ed = E->ed;
	const int	*bp = E->basepr;
	int	e = 0, ip = 0, jp = 0, is, js, k, kp, sum;
	int	fb = open ? 0 : ed->eparam[ 5 ];	/* free base penalty inside multiloops */

	if( bp[ i ] == UNDEF || bp[ j ] == UNDEF ){
		while( bp[ i ] == UNDEF && bp[ i + 1 ] == UNDEF ){
			i++;
			e += fb;
			if( i >= j - 1 )
				return( e );
		}
		while( bp[ j ] == UNDEF && bp[ j - 1 ] == UNDEF ){
			j--;
			e += fb;
			if( i >= j - 1 )
				return( e );
		}
		if( bp[ i ] == UNDEF && bp[ i + 1 ] > i + 1 ){
			e += MIN( 0, ef_dangle( E, bp[ i + 1 ], i + 1, i, 1 ) ) + fb;
			i++;
		}
		if( bp[ j ] == UNDEF && bp[ j - 1 ] != UNDEF && bp[ j - 1 ] < j - 1 ){
			e += MIN( 0, ef_dangle( E, j - 1, bp[ j - 1 ], j, 0 ) ) + fb;
			j--;
		}
	}

	if( bp[ i ] != j ){
		k = bp[ i ];
		kp = bp[ j ];
		if( k >= kp )
			return( INF );		/* "knot" */
		if( bp[ k + 1 ] != UNDEF ){
			e += efn_rec( E, i, k, open );
			e += efn_rec( E, k + 1, j, open );
		}else if( bp[ k + 2 ] == UNDEF ){
			e += efn_rec( E, i, k + 1, open );
			e += efn_rec( E, k + 2, j, open );
		}else if( ef_dangle( E, k, i, k + 1, 0 ) <= ef_dangle( E, bp[ k + 2 ], k + 2, k + 1, 1 ) ){
			e += efn_rec( E, i, k + 1, open );
			e += efn_rec( E, k + 2, j, open );
		}else{
			e += efn_rec( E, i, k, open );
			e += efn_rec( E, k + 1, j, open );
		}
		return( e );
	}

	if( !open )
		e += ed->eparam[ 8 ];
	e += ef_aupen( E, i, j );
	for( ; ; ){
		if( bp[ i + 1 ] == j - 1 ){
			e += ef_stack( E, i, j );
			i++;
			j--;
			continue;
		}
		for( sum = 0, k = i + 1; k < j; ){
			if( bp[ k ] > k ){
				sum++;
				ip = k;
				k = bp[ k ] + 1;
				jp = k - 1;
				if( k > j )
					return( INF );
			}else if( bp[ k ] == UNDEF )
				k++;
			else
				return( INF );	/* the reference would spin here; cannot happen for nested pairs */
		}
		if( sum == 0 ){
			e += ef_hploop( E, i, j );
			return( e );
		}
		if( sum == 1 ){
			e += ef_ibloop( E, i, j, ip, jp );
			i = ip;
			j = jp;
			continue;
		}
		is = i + 1;
		js = j - 1;
		e += ed->eparam[ 4 ] + ed->eparam[ 8 ] + ef_aupen( E, i, j );
		if( bp[ i + 1 ] == UNDEF && bp[ i + 2 ] != UNDEF ){
			if( ef_dangle( E, i, j, i + 1, 0 ) <= ef_dangle( E, bp[ i + 2 ], i + 2, i + 1, 1 ) ){
				is = i + 2;
				e += MIN( 0, ef_dangle( E, i, j, i + 1, 0 ) ) + ed->eparam[ 5 ];
			}
		}
		if( bp[ i + 1 ] == UNDEF && bp[ i + 2 ] == UNDEF ){
			is = i + 2;
			e += MIN( 0, ef_dangle( E, i, j, i + 1, 0 ) ) + ed->eparam[ 5 ];
		}
		if( bp[ j - 1 ] == UNDEF && bp[ j - 2 ] != UNDEF ){
			if( ef_dangle( E, i, j, j - 1, 1 ) <= ef_dangle( E, j - 2, bp[ j - 2 ], j - 1, 0 ) ){
				js = j - 2;
				e += MIN( 0, ef_dangle( E, i, j, j - 1, 1 ) ) + ed->eparam[ 5 ];
			}
		}
		if( bp[ j - 1 ] == UNDEF && bp[ j - 2 ] == UNDEF ){
			js = j - 2;
			e += MIN( 0, ef_dangle( E, i, j, j - 1, 1 ) ) + ed->eparam[ 5 ];
		}
		e += efn_rec( E, is, js, 0 );
		return( e );
	}
}

int	rmo_efn( const rma_efndata_t *ed, const int *bcseq, const int *basepr, int l_base )
{
	efn_t	E;

	E.ed = ed;
	E.bcseq = bcseq;
	E.basepr = basepr;
	E.l_base = l_base;
	return( efn_rec( &E, 0, l_base, 1 ) );
}

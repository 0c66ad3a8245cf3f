/*
 * rm_oracle_scan.c -- TEST INFRASTRUCTURE (see rm_oracle.h).
 *
 * Recursive, scalar restatement of /root/reference/src/find_motif.c over the
 * flattened motif program.  Every routine names the reference routine and
 * line it follows.  Sequence constraints are matched on the reduced atom form
 * with a backtracking matcher that mirrors step()/advance()
 * (/root/reference/src/regexp.c:389-664) and mm_step()/mm_advance()
 * (/root/reference/src/mm_regexp.c:353-469).
 *
 * One deliberate difference: the reference's fm_window[] mark array is
 * malloc'ed and never cleared (find_motif.c:129), so a position that has not
 * been marked yet holds whatever malloc returned.  Here such positions read as
 * UNDEF, which is also what every marked-then-unmarked position holds.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rm_oracle.h"

#define	UNDEF	(-1)
#define	MIN(a,b)	((a)<(b)?(a):(b))
#define	MAX(a,b)	((a)>(b)?(a):(b))
#define	ODD(i)		((i)&0x1)
#define	EPS		1e-6
#define	MAXH		101		/* find_motif.c:406 h3[ 101 ] */

typedef struct ctx_t {
	const rma_program_t	*p;
	const rma_efndata_t	*ed;
	const char	*sbuf;
	int	slen, comp, seq;
	int	szero;				/* fm_szero			*/
	int	windowsize;
	int	zero[ RMA_MAX_ELEMS ], dollar[ RMA_MAX_ELEMS ];	/* SEARCH_T s_zero/s_dollar */
	int	moff[ RMA_MAX_ELEMS ], mlen[ RMA_MAX_ELEMS ];	/* s_matchoff/s_matchlen	*/
	int	mpr[ RMA_MAX_ELEMS ], mm[ RMA_MAX_ELEMS ];	/* s_n_mispairs/_mismatches	*/
	int	l_off, l_len, r_off, r_len;	/* rm_lctx / rm_rctx match	*/
	int	l_mm, r_mm;
	int	*winbuf, *window;		/* fm_winbuf / fm_window	*/
	int	rank, order;
	rmo_hits_t	*hits;
	int	err;
	int	b2bc[ 256 ];
} ctx_t;

static	int	find_motif( ctx_t *, int );

/* ------------------------------------------------------------------ pairing */
static	int	paired2( const ctx_t *c, int ps, int b5, int b3 )	/* RM_paired :1291 */
{
	int	ix = c->b2bc[ ( unsigned char )b5 ] * 5 + c->b2bc[ ( unsigned char )b3 ];
	return( ( c->p->pairsets[ ps ].mat2 >> ix ) & 1 );
}

static	int	triple( const ctx_t *c, int ps, int b1, int b2, int b3 )	/* RM_triple :1304 */
{
	int	ix = ( c->b2bc[ ( unsigned char )b1 ] * 5 + c->b2bc[ ( unsigned char )b2 ] ) * 5 +
			c->b2bc[ ( unsigned char )b3 ];
	return( ( c->p->pairsets[ ps ].mat3[ ix >> 5 ] >> ( ix & 31 ) ) & 1 );
}

static	int	quad( const ctx_t *c, int ps, int b1, int b2, int b3, int b4 )	/* RM_quad :1318 */
{
	int	ix = ( ( c->b2bc[ ( unsigned char )b1 ] * 5 + c->b2bc[ ( unsigned char )b2 ] ) * 5 +
			c->b2bc[ ( unsigned char )b3 ] ) * 5 + c->b2bc[ ( unsigned char )b4 ];
	return( ( c->p->pairsets[ ps ].mat4[ ix >> 5 ] >> ( ix & 31 ) ) & 1 );
}

/* ------------------------------------------------------------------ seq= matcher */
typedef struct restr_t {
	const rma_regex_t	*re;
	const unsigned char	*s;	/* base codes, terminated by 255	*/
} restr_t;

#define	EOS	255
static	int	atom_ok( const rma_re_atom_t *a, int code )
{
	return( code != EOS && ( ( a->mask >> code ) & 1 ) );
}

/* advance(), regexp.c:426: greedy repeats, longest first */
static	int	re_advance( const restr_t *r, int lp, int ia )
{
	const rma_regex_t	*re = r->re;

	for( ; ; ia++ ){
		const rma_re_atom_t	*a;
		int	lo, extra, cur;

		if( ia == re->n_atoms ){
			if( re->dollar )
				return( r->s[ lp ] == EOS );
			return( 1 );
		}
		a = &re->atoms[ ia ];
		if( a->lo == 1 && a->hi == 1 ){
			if( !atom_ok( a, r->s[ lp ] ) )
				return( 0 );
			lp++;
			continue;
		}
		lo = a->lo;
		extra = a->hi == 255 ? 0x7fffffff : a->hi - a->lo;
		for( ; lo > 0; lo-- ){
			if( !atom_ok( a, r->s[ lp ] ) )
				return( 0 );
			lp++;
		}
		cur = lp;
		for( ; extra > 0 && atom_ok( a, r->s[ lp ] ); extra-- )
			lp++;
		for( ; lp >= cur; lp-- ){
			if( re_advance( r, lp, ia + 1 ) )
				return( 1 );
		}
		return( 0 );
	}
}

static	int	re_step( const restr_t *r )		/* step(), regexp.c:389 */
{
	int	p1 = 0;

	if( r->re->anchored )
		return( re_advance( r, 0, 0 ) );
	do{
		if( re_advance( r, p1, 0 ) )
			return( 1 );
	}while( r->s[ p1++ ] != EOS );
	return( 0 );
}

/* mm_advance(), mm_regexp.c:369 */
static	int	re_mm_advance( const restr_t *r, int lp, int l_mm, int *n_mm )
{
	const rma_regex_t	*re = r->re;
	int	ia, k;

	*n_mm = 0;
	for( ia = 0; ia < re->n_atoms; ia++ ){
		const rma_re_atom_t	*a = &re->atoms[ ia ];
		for( k = 0; k < a->lo; k++ ){
			int	code = r->s[ lp++ ];
			if( code == EOS )
				return( 0 );
			if( a->kind == 1 )		/* CDOT never counts	*/
				continue;
			if( !( ( a->mask >> code ) & 1 ) ){
				( *n_mm )++;
				if( *n_mm > l_mm )
					return( 0 );
			}
		}
	}
	if( re->dollar && r->s[ lp ] != EOS )
		return( 0 );
	return( 1 );
}

static	int	re_mm_step( const restr_t *r, int l_mm, int *n_mm )	/* mm_step(), mm_regexp.c:353 */
{
	int	p1 = 0;

	if( r->re->anchored )
		return( re_mm_advance( r, 0, l_mm, n_mm ) );
	do{
		if( re_mm_advance( r, p1, l_mm, n_mm ) )
			return( 1 );
	}while( r->s[ p1++ ] != EOS );
	return( 0 );
}

/* chk_seq(), find_motif.c:1810.  *n_mm is written only when mismatch > 0. */
static	int	chk_seq( ctx_t *c, const rma_elem_t *e, int off, int len, int *n_mm )
{
	static	unsigned char	*buf = NULL;
	static	int	s_buf = 0;
	restr_t	r;
	int	i;

	if( len + 1 > s_buf ){
		s_buf = len + 1 + 1024;
		buf = ( unsigned char * )realloc( buf, s_buf );
	}
	for( i = 0; i < len; i++ )
		buf[ i ] = ( unsigned char )c->b2bc[ ( unsigned char )c->sbuf[ off + i ] ];
	buf[ len ] = EOS;
	r.re = &c->p->regexes[ e->re ];
	r.s = buf;
	if( e->mismatch > 0 )
		return( re_mm_step( &r, e->mismatch, n_mm ) );
	return( re_step( &r ) );
}

/* ------------------------------------------------------------------ marks */
static	void	mark_ss( ctx_t *c, int d, int s5, int slen )	/* :1333 */
{
	int	s;

	c->moff[ d ] = s5;
	c->mlen[ d ] = slen;
	for( s = 0; s < slen; s++ )
		c->window[ s5 + s - c->szero ] = d;
}

static	void	unmark_ss( ctx_t *c, int d, int s5, int slen )	/* :1344 */
{
	int	s;

	c->moff[ d ] = UNDEF;
	c->mlen[ d ] = UNDEF;
	for( s = 0; s < slen; s++ )
		c->window[ s5 + s - c->szero ] = UNDEF;
}

static	void	mark_duplex( ctx_t *c, int d5, int h5, int d3, int h3, int hlen )	/* :1355 */
{
	int	h;

	c->moff[ d5 ] = h5;
	c->mlen[ d5 ] = hlen;
	c->moff[ d3 ] = h3 - hlen + 1;
	c->mlen[ d3 ] = hlen;
	for( h = 0; h < hlen; h++ ){
		c->window[ h5 + h - c->szero ] = d5;
		c->window[ h3 - h - c->szero ] = d5;
	}
}

static	void	unmark_duplex( ctx_t *c, int d5, int h5, int d3, int h3, int hlen )	/* :1371 */
{
	int	h;

	c->moff[ d5 ] = UNDEF;
	c->mlen[ d5 ] = UNDEF;
	c->moff[ d3 ] = UNDEF;
	c->mlen[ d3 ] = UNDEF;
	for( h = 0; h < hlen; h++ ){
		c->window[ h5 + h - c->szero ] = UNDEF;
		c->window[ h3 - h - c->szero ] = UNDEF;
	}
}

/* ------------------------------------------------------------------ helix matchers */
static	int	mplim_of( const ctx_t *c, const rma_elem_t *e, int *pfrac )	/* :1023-1033, :1122-1130 */
{
	int	mplim;

	*pfrac = 0;
	if( e->mispair > 0 )
		mplim = e->mispair;
	else if( e->pairfrac < 1.0 ){
		mplim = ( 1. - e->pairfrac ) * MIN( e->maxlen, c->windowsize ) + 0.5;
		*pfrac = 1;
	}else
		mplim = 0;
	return( mplim );
}

/* match_wchlx(), find_motif.c:975 */
static	int	match_wchlx( ctx_t *c, int d5, int d3, int s5, int s3, int s3lim,
	int h3[], int hlen[], int n_mpr[] )
{
	const rma_elem_t	*stp = &c->p->elems[ d5 ], *stp3 = &c->p->elems[ d3 ];
	int	nh, hl, mpr, l_bpr, mplim, pfrac;
	int	b5, b3;

#define	EMIT()	do{ if( nh >= MAXH ){ c->err = 1; return( nh ); } \
		h3[ nh ] = s3; hlen[ nh ] = hl; n_mpr[ nh ] = mpr; nh++; }while( 0 )

	nh = 0;
	b5 = c->sbuf[ s5 ];
	b3 = c->sbuf[ s3 ];
	if( stp->minlen == 0 ){
		hl = 0;
		mpr = 0;
		if( stp->re >= 0 && !chk_seq( c, stp, s5, hl, &c->mm[ d5 ] ) )
			goto REAL_HELIX;
		if( stp3->re >= 0 ){
			if( chk_seq( c, stp3, s3 - hl + 1, hl, &c->mm[ d3 ] ) )
				EMIT();
		}else
			EMIT();
	}

REAL_HELIX : ;
	if( paired2( c, stp->pairset, b5, b3 ) ){
		hl = 1;
		mpr = 0;
		l_bpr = 1;
	}else if( !( stp->ends & RMA_5PAIRED ) ){
		hl = 1;
		mpr = 1;
		l_bpr = 0;
	}else if( stp->minlen == 0 )
		return( 1 );		/* (sic) :1018-1019 */
	else
		return( 0 );

	mplim = mplim_of( c, stp, &pfrac );

	if( hl >= stp->minlen ){
		if( !l_bpr && ( stp->ends & RMA_3PAIRED ) )
			goto SKIP;
		if( pfrac && 1. * ( hl - mpr ) / hl < stp->pairfrac - EPS )
			goto SKIP;
		if( stp->re >= 0 && !chk_seq( c, stp, s5, hl, &c->mm[ d5 ] ) )
			goto SKIP;
		if( stp3->re >= 0 ){
			if( chk_seq( c, stp3, s3 - hl + 1, hl, &c->mm[ d3 ] ) )
				EMIT();
		}else
			EMIT();
	}
SKIP : ;

	for( ; s3 - hl + 1 >= s3lim; ){
		if( hl >= stp->maxlen )
			break;
		b5 = c->sbuf[ s5 + hl ];
		b3 = c->sbuf[ s3 - hl ];
		if( paired2( c, stp->pairset, b5, b3 ) )
			l_bpr = 1;
		else{
			mpr++;
			if( mpr > mplim )
				break;
			l_bpr = 0;
		}
		hl++;
		if( hl >= stp->minlen ){
			if( !l_bpr && ( stp->ends & RMA_3PAIRED ) )
				continue;
			if( pfrac && ( 1. * hl - mpr ) / hl < stp->pairfrac - EPS )
				continue;
			if( stp->re >= 0 && !chk_seq( c, stp, s5, hl, &c->mm[ d5 ] ) )
				continue;
			if( stp3->re >= 0 ){
				if( chk_seq( c, stp3, s3 - hl + 1, hl, &c->mm[ d3 ] ) )
					EMIT();
			}else
				EMIT();
		}
	}
	return( nh );
#undef EMIT
}

/* match_phlx(), find_motif.c:1114 */
static	int	match_phlx( ctx_t *c, int d5, int d3, int s5, int s3, int s5hi, int s5lo,
	int *hlen, int *n_mpr )
{
	const rma_elem_t	*stp = &c->p->elems[ d5 ], *stp3 = &c->p->elems[ d3 ];
	int	s, s1, b5, b3, mplim, l_pr, pfrac;

	mplim = mplim_of( c, stp, &pfrac );
	b3 = c->sbuf[ s3 ];
	for( s = s5hi; s >= s5lo; s-- ){
		b5 = c->sbuf[ s ];
		if( paired2( c, stp->pairset, b5, b3 ) ){
			*hlen = 1;
			*n_mpr = 0;
			l_pr = 1;
		}else if( !( stp->ends & RMA_5PAIRED ) ){
			*hlen = 1;
			*n_mpr = 1;
			l_pr = 0;
		}else
			continue;
		for( s1 = s - 1; s1 >= s5; s1-- ){
			b5 = c->sbuf[ s1 ];
			b3 = c->sbuf[ s3 - *hlen ];
			if( paired2( c, stp->pairset, b5, b3 ) )
				l_pr = 1;
			else{
				l_pr = 0;
				( *n_mpr )++;
				if( *n_mpr > mplim )
					return( 0 );
			}
			( *hlen )++;
		}
		if( !l_pr && ( stp->ends & RMA_3PAIRED ) )
			return( 0 );
		if( *hlen < stp->minlen || *hlen > stp->maxlen )
			return( 0 );
		if( pfrac && 1. * ( *hlen - *n_mpr ) / ( *hlen ) < stp->pairfrac - EPS )
			return( 0 );
		if( stp->re >= 0 && !chk_seq( c, stp, s5, *hlen, &c->mm[ d5 ] ) )
			return( 0 );
		if( stp3->re >= 0 && !chk_seq( c, stp3, s3 - *hlen + 1, *hlen, &c->mm[ d3 ] ) )
			return( 0 );
		return( 1 );
	}
	return( 0 );
}

/* match_triplex(), find_motif.c:1183 */
static	int	match_triplex( ctx_t *c, int d, int d1, int s1, int s2, int s3, int tlen, int *n_mpr )
{
	const rma_elem_t	*stp = &c->p->elems[ d ], *stp1 = &c->p->elems[ d1 ];
	int	t, mplim, l_pr;

	mplim = 0;
	if( stp->mispair > 0 )
		mplim = stp->mispair;
	else if( stp->pairfrac < 1.0 )
		mplim = ( 1. - stp->pairfrac ) * tlen + 0.5;

	if( triple( c, stp->pairset, c->sbuf[ s1 ], c->sbuf[ s2 ], c->sbuf[ s3 - tlen + 1 ] ) ){
		*n_mpr = 0;
		l_pr = 1;
	}else if( !( stp->ends & RMA_5PAIRED ) ){
		*n_mpr = 1;
		l_pr = 0;
	}else
		return( 0 );
	for( t = 1; t < tlen; t++ ){
		if( !triple( c, stp->pairset, c->sbuf[ s1 + t ], c->sbuf[ s2 - t ], c->sbuf[ s3 - tlen + 1 + t ] ) ){
			l_pr = 0;
			( *n_mpr )++;
			if( *n_mpr > mplim )
				return( 0 );
		}else
			l_pr = 1;
	}
	if( !l_pr && ( stp->ends & RMA_3PAIRED ) )
		return( 0 );
	if( stp1->re >= 0 && !chk_seq( c, stp1, s2 - tlen + 1, tlen, &c->mm[ d1 ] ) )
		return( 0 );
	return( 1 );
}

/* match_4plex(), find_motif.c:1234 */
static	int	match_4plex( ctx_t *c, int d1, int d2, int s1, int s2, int s3, int s4, int qlen, int *n_mpr )
{
	const rma_elem_t	*stp1 = &c->p->elems[ d1 ], *stp2 = &c->p->elems[ d2 ];
	int	q, mplim, l_pr;

	mplim = 0;
	if( stp1->mispair > 0 )
		mplim = stp1->mispair;
	else if( stp1->pairfrac < 1.0 )
		mplim = ( 1. - stp1->pairfrac ) * qlen + 0.5;

	if( quad( c, stp1->pairset, c->sbuf[ s1 + qlen - 1 ], c->sbuf[ s2 ], c->sbuf[ s3 ], c->sbuf[ s4 - qlen + 1 ] ) ){
		*n_mpr = 0;
		l_pr = 1;
	}else if( !( stp1->ends & RMA_5PAIRED ) ){
		*n_mpr = 1;
		l_pr = 0;
	}else
		return( 0 );
	for( *n_mpr = 0, q = 1; q < qlen; q++ ){	/* (sic) :1260 forgets the first mispair */
		if( !quad( c, stp1->pairset, c->sbuf[ s1 + qlen - 1 - q ], c->sbuf[ s2 + q ],
			c->sbuf[ s3 - q ], c->sbuf[ s4 - qlen + 1 + q ] ) ){
			l_pr = 0;
			( *n_mpr )++;
			if( *n_mpr > mplim )
				return( 0 );
		}else
			l_pr = 1;
	}
	if( !l_pr && ( stp1->ends & RMA_3PAIRED ) )
		return( 0 );
	if( stp1->re >= 0 && !chk_seq( c, stp1, s2, qlen, &c->mm[ d1 ] ) )
		return( 0 );
	if( stp2->re >= 0 && !chk_seq( c, stp2, s3 - qlen + 1, qlen, &c->mm[ d2 ] ) )
		return( 0 );
	return( 1 );
}

/* ------------------------------------------------------------------ strict helices */
static	int	wtype( const ctx_t *c, int pos, int undef_is_ss )
{
	int	d = c->window[ pos - c->szero ];

	if( d == UNDEF )
		return( undef_is_ss ? RMA_T_SS : -1 );
	return( c->p->elems[ d ].type );
}

static	int	chk_wchlx( ctx_t *c, int d )	/* :1441 */
{
	const rma_elem_t	*stp = &c->p->elems[ d ];
	int	d3 = stp->mates[ 0 ];
	int	h5_5 = c->moff[ d ], h5_3 = h5_5 + c->mlen[ d ] - 1;
	int	h3_5 = c->moff[ d3 ], h3_3 = h3_5 + c->mlen[ d3 ] - 1;

	if( stp->strict & RMA_5STRICT ){
		if( h5_5 > 0 && h3_3 < c->slen - 1 ){
			if( wtype( c, h5_5 - 1, 1 ) == RMA_T_SS && wtype( c, h3_3 + 1, 1 ) == RMA_T_SS ){
				if( paired2( c, stp->pairset, c->sbuf[ h5_5 - 1 ], c->sbuf[ h3_3 + 1 ] ) )
					return( 0 );
			}
		}
	}
	if( stp->strict & RMA_3STRICT ){
		if( wtype( c, h5_3 + 1, 0 ) == RMA_T_SS && wtype( c, h3_5 - 1, 0 ) == RMA_T_SS ){
			if( paired2( c, stp->pairset, c->sbuf[ h5_3 + 1 ], c->sbuf[ h3_5 - 1 ] ) )
				return( 0 );
		}
	}
	return( 1 );
}

static	int	chk_triplex( ctx_t *c, int d )	/* :1557 */
{
	const rma_elem_t	*stp = &c->p->elems[ d ];
	int	d1 = stp->mates[ 0 ], d2 = stp->mates[ 1 ];
	int	t1_5 = c->moff[ d ], t1_3 = t1_5 + c->mlen[ d ] - 1;
	int	t2_5 = c->moff[ d1 ], t2_3 = t2_5 + c->mlen[ d1 ] - 1;
	int	t3_5 = c->moff[ d2 ], t3_3 = t3_5 + c->mlen[ d2 ] - 1;

	if( ( stp->strict & RMA_5STRICT ) && t1_5 > 0 ){
		if( wtype( c, t1_5 - 1, 1 ) == RMA_T_SS && wtype( c, t2_3 + 1, 0 ) == RMA_T_SS &&
			wtype( c, t3_5 - 1, 0 ) == RMA_T_SS ){
			if( triple( c, stp->pairset, c->sbuf[ t1_5 - 1 ], c->sbuf[ t2_3 + 1 ], c->sbuf[ t3_5 - 1 ] ) )
				return( 0 );
		}
	}
	if( ( stp->strict & RMA_3STRICT ) && t3_3 < c->slen - 1 ){
		if( wtype( c, t1_3 + 1, 0 ) == RMA_T_SS && wtype( c, t2_5 - 1, 0 ) == RMA_T_SS &&
			wtype( c, t3_3 + 1, 1 ) == RMA_T_SS ){
			if( triple( c, stp->pairset, c->sbuf[ t1_3 + 1 ], c->sbuf[ t2_5 - 1 ], c->sbuf[ t3_3 + 1 ] ) )
				return( 0 );
		}
	}
	return( 1 );
}

static	int	chk_4plex( ctx_t *c, int d )	/* :1629 */
{
	const rma_elem_t	*stp = &c->p->elems[ d ];
	int	d1 = stp->mates[ 0 ], d2 = stp->mates[ 1 ], d3 = stp->mates[ 2 ];
	int	q1_5 = c->moff[ d ], q1_3 = q1_5 + c->mlen[ d ] - 1;
	int	q2_5 = c->moff[ d1 ], q2_3 = q2_5 + c->mlen[ d1 ] - 1;
	int	q3_5 = c->moff[ d2 ], q3_3 = q3_5 + c->mlen[ d2 ] - 1;
	int	q4_5 = c->moff[ d3 ], q4_3 = q4_5 + c->mlen[ d3 ] - 1;

	if( stp->strict & RMA_5STRICT ){
		if( q1_5 > 0 && q4_3 < c->slen - 1 ){
			if( wtype( c, q1_5 - 1, 1 ) == RMA_T_SS && wtype( c, q2_3 + 1, 0 ) == RMA_T_SS &&
				wtype( c, q3_5 - 1, 0 ) == RMA_T_SS && wtype( c, q4_3 + 1, 1 ) == RMA_T_SS ){
				if( quad( c, stp->pairset, c->sbuf[ q1_5 - 1 ], c->sbuf[ q2_3 + 1 ],
					c->sbuf[ q3_5 - 1 ], c->sbuf[ q4_3 + 1 ] ) )
					return( 0 );
			}
		}
	}
	if( stp->strict & RMA_3STRICT ){
		/* (sic) :1706-1707 tests st3 twice and never st4 */
		if( wtype( c, q1_3 + 1, 0 ) == RMA_T_SS && wtype( c, q2_5 - 1, 0 ) == RMA_T_SS &&
			wtype( c, q3_3 + 1, 0 ) == RMA_T_SS ){
			if( quad( c, stp->pairset, c->sbuf[ q1_3 + 1 ], c->sbuf[ q2_5 - 1 ],
				c->sbuf[ q3_3 + 1 ], c->sbuf[ q4_5 - 1 ] ) )
				return( 0 );
		}
	}
	return( 1 );
}

static	int	chk_motif( ctx_t *c )	/* :1406; chk_phlx :1500 always returns TRUE */
{
	int	d;

	for( d = 0; d < c->p->n_elems; d++ ){
		const rma_elem_t	*stp = &c->p->elems[ d ];
		if( !stp->strict )
			continue;
		switch( stp->type ){
		case RMA_T_H5 :
			if( !chk_wchlx( c, d ) )
				return( 0 );
			break;
		case RMA_T_T1 :
			if( !chk_triplex( c, d ) )
				return( 0 );
			break;
		case RMA_T_Q1 :
			if( !chk_4plex( c, d ) )
				return( 0 );
			break;
		default :
			break;
		}
	}
	return( 1 );
}

/* ------------------------------------------------------------------ context, sites */
static	int	set_context( ctx_t *c )	/* :1720 */
{
	const rma_program_t	*p = c->p;
	int	offset, length;

	if( !p->has_lctx ){
		if( !p->has_rctx )
			return( 1 );
	}else{
		offset = c->l_off = MAX( c->moff[ 0 ] - p->lctx.maxlen, 0 );
		length = c->l_len = c->moff[ 0 ] - c->l_off;
		if( length < p->lctx.minlen )
			return( 0 );
		if( p->lctx.re >= 0 && !chk_seq( c, &p->lctx, offset, length, &c->l_mm ) )
			return( 0 );
	}
	if( p->has_rctx ){
		int	n = p->n_elems - 1;
		c->r_off = c->moff[ n ] + c->mlen[ n ];
		offset = MIN( c->r_off + p->rctx.maxlen, c->slen );
		length = c->r_len = offset - c->r_off;
		if( length < p->rctx.minlen )
			return( 0 );
		/* (sic) :1749-1751 hands the end of the context, not its start, to chk_seq;
		 * the copy stops at the sequence's terminating NUL */
		if( p->rctx.re >= 0 ){
			int	avail = c->slen - offset;
			if( !chk_seq( c, &p->rctx, offset, MIN( length, avail ), &c->r_mm ) )
				return( 0 );
		}
	}
	return( 1 );
}

static	int	chk_sites( ctx_t *c )	/* :1758, chk_1_site :1769 */
{
	int	s, k;

	for( s = 0; s < c->p->n_sites; s++ ){
		const rma_site_t	*sip = &c->p->sites[ s ];
		int	b[ 4 ], rv = 0;
		for( k = 0; k < sip->n_pos; k++ ){
			const rma_site_pos_t	*pp = &sip->pos[ k ];
			int	d = pp->elem, pos;
			if( pp->l2r ){
				if( pp->offset > c->mlen[ d ] )
					return( 0 );
				pos = c->moff[ d ] + pp->offset - 1;
			}else if( pp->offset >= c->mlen[ d ] )
				return( 0 );
			else
				pos = c->moff[ d ] + c->mlen[ d ] - pp->offset - 1;
			b[ k ] = c->sbuf[ pos ];
		}
		if( sip->n_pos == 2 )
			rv = paired2( c, sip->pairset, b[ 0 ], b[ 1 ] );
		else if( sip->n_pos == 3 )
			rv = triple( c, sip->pairset, b[ 0 ], b[ 1 ], b[ 2 ] );
		else if( sip->n_pos == 4 )
			rv = quad( c, sip->pairset, b[ 0 ], b[ 1 ], b[ 2 ], b[ 3 ] );
		if( !rv )
			return( 0 );
	}
	return( 1 );
}

/* ------------------------------------------------------------------ efn at the candidate */
static	const rma_efn2data_t	*rmo_efn2data = NULL;	/* tables for efn2() sites, set by the tests */
void	rmo_set_efn2data( const rma_efn2data_t *ed ) { rmo_efn2data = ed; }

static	int	efn_site( ctx_t *c, const rma_efn_site_t *es )	/* setupefn/setbp, score.c:3128-3250 */
{
	const rma_program_t	*p = c->p;
	static	int	*bcseq = NULL, *basepr = NULL;
	static	int	s_buf = 0;
	int	idx = es->idx, idx2 = es->idx2, pos = es->pos;
	int	pos2 = es->pos2 < 0 ? c->mlen[ idx2 ] - 1 : es->pos2;
	int	off5 = c->moff[ idx ];
	int	len, d, i, pq;
	int	ps;

	for( len = 0, d = idx; d <= idx2; d++ )
		len += c->mlen[ d ];
	len -= pos;
	len -= c->mlen[ idx2 ] - ( pos2 + 1 );
	if( len <= 0 )
		return( RMA_EFN_INFINITY );
	if( len + 8 > s_buf ){
		s_buf = len + 1024;
		bcseq = ( int * )realloc( bcseq, s_buf * sizeof( int ) );
		basepr = ( int * )realloc( basepr, s_buf * sizeof( int ) );
	}
	for( i = 0; i < len + 8; i++ ){
		bcseq[ i ] = RMA_BC_N;
		basepr[ i ] = UNDEF;
	}
	i = 0;
	for( d = idx; d <= idx2; d++ ){
		const rma_elem_t	*stp = &p->elems[ d ];
		int	p0 = d == idx ? pos : 0;
		int	p1 = d == idx2 ? pos2 + 1 : c->mlen[ d ];
		for( pq = p0; pq < p1; pq++, i++ ){
			int	dopair;
			bcseq[ i ] = c->b2bc[ ( unsigned char )c->sbuf[ c->moff[ d ] + pq ] ];
			if( stp->type == RMA_T_H5 )
				dopair = d != idx2;		/* a trailing h5 is treated as ss */
			else if( stp->type == RMA_T_H3 )
				dopair = d != idx;		/* a leading h3 is treated as ss  */
			else if( stp->type == RMA_T_SS )
				dopair = 0;
			else
				return( RMA_EFN_INFINITY );	/* reference: fatal error */
			basepr[ i ] = UNDEF;
			if( dopair ){
				int	m = stp->mates[ 0 ];
				int	q1 = c->mlen[ m ] - pq - 1;
				int	bp1 = q1 + c->moff[ m ] - off5;
				int	b = c->sbuf[ pq + c->moff[ d ] ];
				int	b1 = c->sbuf[ q1 + c->moff[ m ] ];
				if( !stp->proper || bp1 < 0 || bp1 >= len )
					return( RMA_EFN_INFINITY );	/* reference: fatal error */
				ps = p->efn_usestdbp ? p->efn_stdbp : stp->pairset;
				basepr[ i ] = paired2( c, ps, b, b1 ) ? bp1 : UNDEF;
			}
		}
	}
	if( es->kind == RMA_EFN_KIND_EFN2 ){
		int	undefined;
		return( rmo_efn2data ? rmo_efn2( rmo_efn2data, bcseq, basepr, len - 1, &undefined ) : RMA_EFN2_INFINITY );
	}
	return( rmo_efn( c->ed, bcseq, basepr, len - 1 ) );
}

/* ------------------------------------------------------------------ candidate emission */
static	void	emit_hit( ctx_t *c )	/* find_ss :373-392 up to RM_score() */
{
	const rma_program_t	*p = c->p;
	rmo_hits_t	*h = c->hits;
	int32_t	*w;
	int	d, k;

	if( h->n == h->cap ){
		h->cap = h->cap ? 2 * h->cap : 1024;
		h->data = ( int32_t * )realloc( h->data, h->cap * h->stride * sizeof( int32_t ) );
	}
	w = h->data + h->n * h->stride;
	w[ 0 ] = c->seq;
	w[ 1 ] = c->comp;
	w[ 2 ] = c->szero;
	w[ 3 ] = c->rank;
	w[ 4 ] = c->order++;
	for( d = 0; d < p->n_elems; d++ ){
		w[ RMA_HIT_HDR + 4 * d + 0 ] = c->moff[ d ];
		w[ RMA_HIT_HDR + 4 * d + 1 ] = c->mlen[ d ];
		w[ RMA_HIT_HDR + 4 * d + 2 ] = c->mpr[ d ];
		w[ RMA_HIT_HDR + 4 * d + 3 ] = c->mm[ d ];
	}
	k = rma_hit_ctx_off( p );
	w[ k + 0 ] = p->has_lctx ? c->l_off : 0;
	w[ k + 1 ] = p->has_lctx ? c->l_len : 0;
	w[ k + 2 ] = p->has_rctx ? c->r_off : 0;
	w[ k + 3 ] = p->has_rctx ? c->r_len : 0;
	k = rma_hit_efn_off( p );
	for( d = 0; d < p->n_efn_sites; d++ )
		w[ k + d ] = ( c->ed || p->efn_sites[ d ].kind == RMA_EFN_KIND_EFN2 ) ? efn_site( c, &p->efn_sites[ d ] ) : RMA_EFN_INFINITY;
	h->n++;
}

/* ------------------------------------------------------------------ the search */
static	int	find_ss( ctx_t *c, int s )	/* :332 */
{
	const rma_program_t	*p = c->p;
	int	d = p->searches[ s ];
	const rma_elem_t	*stp = &p->elems[ d ];
	int	szero = c->zero[ s ], sdollar = c->dollar[ s ];
	int	slen = sdollar - szero + 1;
	int	rv;

	c->mm[ d ] = 0;
	c->mpr[ d ] = 0;
	if( slen < stp->minlen || slen > stp->maxlen )
		return( 0 );
	if( stp->re >= 0 && !chk_seq( c, stp, szero, slen, &c->mm[ d ] ) )
		return( 0 );
	mark_ss( c, d, szero, slen );
	if( s + 1 < p->n_searches )		/* srp->s_forward */
		rv = find_motif( c, s + 1 );
	else{
		rv = 1;
		if( p->strict_helices && !chk_motif( c ) )
			rv = 0;
		else if( !set_context( c ) )
			rv = 0;
		else if( !chk_sites( c ) )
			rv = 0;
		else
			emit_hit( c );		/* RM_score()/print_match() happen on the host */
	}
	unmark_ss( c, d, szero, slen );
	return( rv );
}

static	int	s3lim_of( int szero, int sdollar, int i_minl, int h_maxl )	/* :426-429 */
{
	int	s3lim = sdollar - szero + 1;

	s3lim = ( s3lim - i_minl ) / 2;
	s3lim = MIN( s3lim, h_maxl );
	return( sdollar - s3lim + 1 );
}

static	int	find_wchlx( ctx_t *c, int s )	/* :400 */
{
	const rma_program_t	*p = c->p;
	int	d = p->searches[ s ];
	const rma_elem_t	*stp = &p->elems[ d ];
	int	d3 = stp->mates[ 0 ];
	int	szero = c->zero[ s ], sdollar = c->dollar[ s ];
	int	h3[ MAXH ], hlen[ MAXH ], n_mpr[ MAXH ];
	int	h, n_h3, rv = 0, s3lim, i_len, is;

	c->mm[ d ] = c->mpr[ d ] = 0;
	c->mm[ d3 ] = c->mpr[ d3 ] = 0;
	s3lim = s3lim_of( szero, sdollar, stp->minilen, stp->maxlen );
	if( ( n_h3 = match_wchlx( c, d, d3, szero, sdollar, s3lim, h3, hlen, n_mpr ) ) ){
		for( h = 0; h < n_h3; h++ ){
			i_len = h3[ h ] - szero - 2 * hlen[ h ] + 1;
			if( i_len > stp->maxilen )
				continue;
			c->mpr[ d ] = c->mpr[ d3 ] = n_mpr[ h ];
			mark_duplex( c, d, szero, d3, h3[ h ], hlen[ h ] );
			is = p->elems[ stp->inner ].searchno;
			c->zero[ is ] = szero + hlen[ h ];
			c->dollar[ is ] = h3[ h ] - hlen[ h ];
			rv |= find_motif( c, is );
			unmark_duplex( c, d, szero, d3, h3[ h ], hlen[ h ] );
		}
	}
	return( rv );
}

static	int	find_minlen( const ctx_t *c, int fd, int ld )	/* :642 */
{
	int	minl = 0, d;

	for( d = fd; d <= ld; d++ )
		minl += c->mlen[ d ] != UNDEF ? c->mlen[ d ] : c->p->elems[ d ].minlen;
	return( minl );
}

static	int	find_maxlen( const ctx_t *c, int fd, int ld )	/* :655 */
{
	int	maxl = 0, d;

	for( d = fd; d <= ld; d++ )
		maxl += c->mlen[ d ] != UNDEF ? c->mlen[ d ] : c->p->elems[ d ].maxlen;
	return( maxl );
}

static	void	upd_pksearches( ctx_t *c, int d, int h5, int h3, int hlen )	/* :667 */
{
	const rma_program_t	*p = c->p;
	const rma_elem_t	*stp = &p->elems[ d ], *stp3;
	int	i;

	if( stp->scope > 0 ){
		i = p->elems[ stp->scopes[ stp->scope - 1 ] ].inner;
		if( i >= 0 )
			c->dollar[ p->elems[ i ].searchno ] = h5 - 1;
	}
	i = stp->inner;
	if( i >= 0 )
		c->zero[ p->elems[ i ].searchno ] = h5 + hlen;
	stp3 = &p->elems[ stp->mates[ 0 ] ];
	i = p->elems[ stp3->scopes[ stp3->scope - 1 ] ].inner;
	if( i >= 0 )
		c->dollar[ p->elems[ i ].searchno ] = h3 - hlen;
	if( stp3->scope < stp3->n_scopes - 1 ){
		i = stp3->inner;
		if( i >= 0 )
			c->zero[ p->elems[ i ].searchno ] = h3 + 1;
	}
}

static	int	find_pknot3( ctx_t *c, int s, int s5 )	/* :530 */
{
	const rma_program_t	*p = c->p;
	int	d5 = p->searches[ s ];
	const rma_elem_t	*stp5 = &p->elems[ d5 ];
	int	d3 = stp5->mates[ 0 ];
	int	dn = stp5->scopes[ stp5->n_scopes - 1 ];
	int	sdollar = c->dollar[ s ], slen = sdollar - s5 + 1;
	int	h_minl = stp5->minlen, h_maxl = stp5->maxlen;
	int	i_minl, g_minl, s_minl, s_maxl, f_s3, l_s3, s3, s3lim, hlx;
	int	iL_minl = 0, iL_maxl = 0, iL_last = 0, iR_minl = 0, iR_maxl = 0, iR_last = 0;
	int	h3[ MAXH ], hlen[ MAXH ], n_mpr[ MAXH ];
	int	h, n_h3, rv = 0;

	i_minl = find_minlen( c, d5 + 1, d3 - 1 );
	g_minl = 2 * h_minl + i_minl;
	s_minl = find_minlen( c, d3 + 1, dn );
	s_maxl = find_maxlen( c, d3 + 1, dn );
	if( g_minl + s_minl > slen )
		return( 0 );
	f_s3 = sdollar - s_minl;
	l_s3 = sdollar - MIN( slen - g_minl, s_maxl );

	hlx = d5 == stp5->scopes[ 1 ] ? 2 : 1;
	if( hlx == 2 ){
		int	d3_h1 = p->elems[ stp5->scopes[ 0 ] ].mates[ 0 ];
		int	s_left, e_left, s_right, e_right;
		iL_last = c->moff[ d3_h1 ] - 1;
		iR_last = c->moff[ d3_h1 ] + c->mlen[ d3_h1 ];
		s_left = d5 + 1;
		e_left = d3_h1 - 1;
		if( s_left <= e_left ){
			iL_minl = find_minlen( c, s_left, e_left );
			iL_maxl = find_maxlen( c, s_left, e_left );
		}
		s_right = d3_h1 + 1;
		e_right = d3 - 1;
		if( s_right <= e_right ){
			iR_minl = find_minlen( c, s_right, e_right );
			iR_maxl = find_maxlen( c, s_right, e_right );
		}
	}

	for( s3 = f_s3; s3 >= l_s3; s3-- ){
		s3lim = s3lim_of( s5, s3, i_minl, h_maxl );
		if( ( n_h3 = match_wchlx( c, d5, d3, s5, s3, s3lim, h3, hlen, n_mpr ) ) ){
			for( h = 0; h < n_h3; h++ ){
				if( ( s3 - s5 + 1 ) - 2 * hlen[ h ] < i_minl )
					break;
				if( hlx == 2 ){
					if( iL_last - ( s5 + hlen[ h ] - 1 ) < iL_minl )
						continue;
					if( iL_last - ( s5 + hlen[ h ] - 1 ) > iL_maxl )
						continue;
					if( ( s3 - hlen[ h ] + 1 ) - iR_last < iR_minl )
						continue;
					if( ( s3 - hlen[ h ] + 1 ) - iR_last > iR_maxl )
						continue;
				}
				c->mpr[ d5 ] = c->mpr[ d3 ] = n_mpr[ h ];
				mark_duplex( c, d5, s5, d3, h3[ h ], hlen[ h ] );
				upd_pksearches( c, d5, s5, h3[ h ], hlen[ h ] );
				rv |= find_motif( c, s + 1 );
				unmark_duplex( c, d5, s5, d3, h3[ h ], hlen[ h ] );
			}
		}
	}
	return( rv );
}

static	int	find_pknot5( ctx_t *c, int s )	/* :495 */
{
	const rma_program_t	*p = c->p;
	int	d5 = p->searches[ s ];
	const rma_elem_t	*stp5 = &p->elems[ d5 ];
	int	d0 = stp5->scopes[ 0 ], dn = stp5->scopes[ stp5->n_scopes - 1 ];
	int	szero = c->zero[ s ], sdollar = c->dollar[ s ], slen = sdollar - szero + 1;
	int	p_minl, p_maxl, r_minl, r_maxl, s5, f_s5, l_s5, rv = 0;

	p_minl = find_minlen( c, d0, d5 - 1 );
	p_maxl = find_maxlen( c, d0, d5 - 1 );
	r_minl = find_minlen( c, d5, dn );
	r_maxl = find_maxlen( c, d5, dn );
	if( p_maxl + r_maxl < slen )
		return( 0 );
	f_s5 = szero + p_minl;
	l_s5 = szero + MIN( p_maxl, slen - r_minl );
	for( s5 = f_s5; s5 <= l_s5; s5++ )
		rv |= find_pknot3( c, s, s5 );
	return( rv );
}

static	int	find_pknot( ctx_t *c, int s )	/* :465 */
{
	const rma_program_t	*p = c->p;
	int	d = p->searches[ s ];
	const rma_elem_t	*stp = &p->elems[ d ];
	int	k;

	if( stp->scope == 0 ){
		for( k = 1; k < stp->n_scopes; k++ ){
			int	d1 = stp->scopes[ k ];
			if( p->elems[ d1 ].type == RMA_T_H5 ){
				int	s1 = p->elems[ d1 ].searchno;
				c->moff[ d1 ] = UNDEF;
				c->mlen[ d1 ] = UNDEF;
				c->zero[ s1 ] = c->zero[ s ];
				c->dollar[ s1 ] = c->dollar[ s ];
			}
		}
	}
	return( find_pknot5( c, s ) );
}

static	void	phlx_bounds( int szero, int slen, int h_minl, int h_maxl, int i_minl, int i_maxsum,
	int *s5hi, int *s5lo )	/* :730-739, :801-810 */
{
	int	ilen;

	*s5hi = MIN( ( slen - i_minl ) / 2, h_maxl );
	*s5hi = szero + *s5hi - 1;
	ilen = slen - 2 * h_minl;
	ilen = MIN( ilen, i_maxsum );
	*s5lo = slen - ilen;
	if( ODD( *s5lo ) )
		( *s5lo )++;
	*s5lo = MIN( *s5lo / 2, h_maxl );
	*s5lo = szero + *s5lo - 1;
}

static	int	find_phlx( ctx_t *c, int s )	/* :703 */
{
	const rma_program_t	*p = c->p;
	int	d = p->searches[ s ];
	const rma_elem_t	*stp = &p->elems[ d ];
	int	d3 = stp->mates[ 0 ];
	int	szero = c->zero[ s ], sdollar = c->dollar[ s ], slen = sdollar - szero + 1;
	int	s5hi, s5lo, hlen, n_mpr, i_len, is, rv = 0;

	c->mm[ d ] = c->mpr[ d ] = 0;
	c->mm[ d3 ] = c->mpr[ d3 ] = 0;
	phlx_bounds( szero, slen, stp->minlen, stp->maxlen, stp->minilen, stp->maxilen, &s5hi, &s5lo );
	if( match_phlx( c, d, d3, szero, sdollar, s5hi, s5lo, &hlen, &n_mpr ) ){
		i_len = sdollar - szero - 2 * hlen + 1;
		if( i_len > stp->maxilen )
			return( 0 );
		c->mpr[ d ] = c->mpr[ d3 ] = n_mpr;
		mark_duplex( c, d, szero, d3, sdollar, hlen );
		is = p->elems[ stp->inner ].searchno;
		c->zero[ is ] = szero + hlen;
		c->dollar[ is ] = sdollar - hlen;
		rv = find_motif( c, is );
		unmark_duplex( c, d, szero, d3, sdollar, hlen );
	}
	return( rv );
}

static	int	find_triplex( ctx_t *c, int s )	/* :763 */
{
	const rma_program_t	*p = c->p;
	int	d = p->searches[ s ];
	const rma_elem_t	*stp = &p->elems[ d ];
	int	d1 = stp->scopes[ 1 ], d2 = stp->scopes[ 2 ];
	const rma_elem_t	*stp1 = &p->elems[ d1 ];
	int	szero = c->zero[ s ], sdollar = c->dollar[ s ], slen = sdollar - szero + 1;
	int	i1_minl = stp->minilen, i1_maxl = stp->maxilen;
	int	i2_minl = stp1->minilen, i2_maxl = stp1->maxilen;
	int	i1s = p->elems[ stp->inner ].searchno, i2s = p->elems[ stp1->inner ].searchno;
	int	s5hi, s5lo, hlen, n_mpr, i_len, i1_len, i2_len, sp, rv = 0;

	c->mm[ d ] = c->mpr[ d ] = 0;
	c->mm[ d1 ] = c->mpr[ d1 ] = 0;
	c->mm[ d2 ] = c->mpr[ d2 ] = 0;
	phlx_bounds( szero, slen, stp->minlen, stp->maxlen, i1_minl + i2_minl,
		i1_maxl + stp->minlen + i2_maxl, &s5hi, &s5lo );
	if( match_phlx( c, d, d2, szero, sdollar, s5hi, s5lo, &hlen, &n_mpr ) ){
		i_len = sdollar - szero - 2 * hlen + 1;
		if( i_len > i1_maxl + i2_maxl + hlen )
			return( 0 );
		mark_duplex( c, d, szero, d2, sdollar, hlen );
		for( sp = sdollar - i2_minl - hlen; sp >= szero + 2 * hlen + i1_minl - 1; sp-- ){
			if( match_triplex( c, d, d1, szero, sp, sdollar, hlen, &n_mpr ) ){
				i1_len = sp - 2 * hlen - szero + 1;
				if( i1_len > i1_maxl )
					continue;
				i2_len = sdollar - hlen - sp;
				if( i2_len > i2_maxl )
					continue;
				c->mpr[ d ] = c->mpr[ d1 ] = c->mpr[ d2 ] = n_mpr;
				mark_ss( c, d1, sp - hlen + 1, hlen );
				c->zero[ i1s ] = szero + hlen;
				c->dollar[ i1s ] = sp - hlen;
				c->zero[ i2s ] = sp + 1;
				c->dollar[ i2s ] = sdollar - hlen;
				rv |= find_motif( c, i1s );
				unmark_ss( c, d1, sp - hlen + 1, hlen );
			}
		}
		unmark_duplex( c, d, szero, d2, sdollar, hlen );
	}
	return( rv );
}

static	int	find_4plex_inner( ctx_t *c, int s, int s3, int hlen )	/* :902 */
{
	const rma_program_t	*p = c->p;
	int	d = p->searches[ s ];
	const rma_elem_t	*stp = &p->elems[ d ];
	int	d1 = stp->mates[ 0 ], d2 = stp->mates[ 1 ], d3 = stp->mates[ 2 ];
	const rma_elem_t	*stp1 = &p->elems[ d1 ], *stp2 = &p->elems[ d2 ];
	int	szero = c->zero[ s ];
	int	i1_minl = stp->minilen, i1_maxl = stp->maxilen;
	int	i2_minl = stp1->minilen, i2_maxl = stp1->maxilen;
	int	i3_minl = stp2->minilen, i3_maxl = stp2->maxilen;
	int	i1s = p->elems[ stp->inner ].searchno;
	int	i2s = p->elems[ stp1->inner ].searchno;
	int	i3s = p->elems[ stp2->inner ].searchno;
	int	s1, s1lim, s2, s2lim, n_mpr, rv = 0;

	s1lim = s3 - 3 * hlen - i3_minl - i2_minl;
	for( s1 = szero + hlen + i1_minl; s1 <= s1lim; s1++ ){
		s2lim = s1 + 2 * hlen + i2_minl;
		for( s2 = s3 - hlen - i3_minl; s2 >= s2lim; s2-- ){
			if( match_4plex( c, d1, d2, szero, s1, s2, s3, hlen, &n_mpr ) ){
				if( s1 - szero - hlen + 1 > i1_maxl )
					continue;
				if( s2 - s1 - 2 * hlen + 1 > i2_maxl )
					continue;
				if( s3 - s2 - hlen + 1 > i3_maxl )
					continue;
				c->mpr[ d ] = c->mpr[ d1 ] = c->mpr[ d2 ] = c->mpr[ d3 ] = n_mpr;
				mark_duplex( c, d1, s1, d2, s2, hlen );
				c->zero[ i1s ] = szero + hlen;
				c->dollar[ i1s ] = s1 - 1;
				c->zero[ i2s ] = s1 + hlen;
				c->dollar[ i2s ] = s2 - hlen;
				c->zero[ i3s ] = s2 + 1;
				c->dollar[ i3s ] = s3 - hlen;
				rv |= find_motif( c, i1s );
				unmark_duplex( c, d1, s1, d2, s2, hlen );
			}
		}
	}
	return( rv );
}

static	int	find_4plex( ctx_t *c, int s )	/* :851 */
{
	const rma_program_t	*p = c->p;
	int	d = p->searches[ s ];
	const rma_elem_t	*stp = &p->elems[ d ];
	int	d1 = stp->mates[ 0 ], d2 = stp->mates[ 1 ], d3 = stp->mates[ 2 ];
	int	szero = c->zero[ s ], sdollar = c->dollar[ s ];
	int	h3[ MAXH ], hlen[ MAXH ], n_mpr[ MAXH ];
	int	h, n_h3, i_minl, s3lim, rv = 0;

	c->mm[ d ] = c->mpr[ d ] = 0;
	c->mm[ d1 ] = c->mpr[ d1 ] = 0;
	c->mm[ d2 ] = c->mpr[ d2 ] = 0;
	c->mm[ d3 ] = c->mpr[ d3 ] = 0;
	i_minl = stp->minilen + p->elems[ d1 ].minilen + p->elems[ d2 ].minilen + 2 * stp->minlen;
	s3lim = s3lim_of( szero, sdollar, i_minl, stp->maxlen );
	if( ( n_h3 = match_wchlx( c, d, d3, szero, sdollar, s3lim, h3, hlen, n_mpr ) ) ){
		for( h = 0; h < n_h3; h++ ){
			mark_duplex( c, d, szero, d3, h3[ h ], hlen[ h ] );
			rv |= find_4plex_inner( c, s, h3[ h ], hlen[ h ] );
			unmark_duplex( c, d, szero, d3, h3[ h ], hlen[ h ] );
		}
	}
	return( rv );
}

static	int	find_1_motif( ctx_t *c, int s )	/* :289 */
{
	const rma_elem_t	*stp = &c->p->elems[ c->p->searches[ s ] ];

	switch( stp->type ){
	case RMA_T_SS :
		return( find_ss( c, s ) );
	case RMA_T_H5 :
		return( stp->proper ? find_wchlx( c, s ) : find_pknot( c, s ) );
	case RMA_T_P5 :
		return( find_phlx( c, s ) );
	case RMA_T_T1 :
		return( find_triplex( c, s ) );
	case RMA_T_Q1 :
		return( find_4plex( c, s ) );
	default :
		c->err = 2;
		return( 0 );
	}
}

static	int	find_motif( ctx_t *c, int s )	/* :245 */
{
	const rma_program_t	*p = c->p;
	const rma_elem_t	*stp = &p->elems[ p->searches[ s ] ];
	int	n_s, loop, sdollar, o_sdollar, f_sdollar, l_sdollar, rv = 0;

	if( c->err )
		return( 0 );
	if( stp->next >= 0 ){
		n_s = p->elems[ stp->next ].searchno;
		loop = 1;
	}else if( stp->outer < 0 ){
		n_s = -1;
		loop = 1;
	}else{
		n_s = -1;
		loop = 0;
	}
	o_sdollar = c->dollar[ s ];
	if( stp->maxglen == RMA_UNBOUNDED )
		f_sdollar = c->dollar[ s ];
	else
		f_sdollar = MIN( c->dollar[ s ], c->zero[ s ] + stp->maxglen - 1 );
	l_sdollar = c->zero[ s ] + stp->minglen - 1;
	if( loop ){
		for( sdollar = f_sdollar; sdollar >= l_sdollar; sdollar-- ){
			if( s == 0 ){		/* boundary sort key, not in the reference */
				c->rank = f_sdollar - sdollar;
				c->order = 0;
			}
			c->dollar[ s ] = sdollar;
			if( n_s >= 0 ){
				c->zero[ n_s ] = sdollar + 1;
				c->dollar[ n_s ] = o_sdollar;
			}
			rv |= find_1_motif( c, s );
		}
	}else
		rv = find_1_motif( c, s );
	c->dollar[ s ] = o_sdollar;
	return( rv );
}

/* ------------------------------------------------------------------ entry points */
void	rmo_hits_init( rmo_hits_t *h, const rma_program_t *p )
{
	h->data = NULL;
	h->n = h->cap = 0;
	h->stride = rma_hit_stride( p );
}

void	rmo_hits_free( rmo_hits_t *h )
{
	free( h->data );
	h->data = NULL;
	h->n = h->cap = 0;
}

void	rmo_revcomp( char *sbuf, int slen )	/* mk_rcmp, rnamot.c:193 */
{
	char	*sp, *cp;
	int	c1, c2;

#define	WC_CMP(c)	((c)=='a'||(c)=='A'?'t':(c)=='c'||(c)=='C'?'g':(c)=='g'||(c)=='G'?'c': \
			 (c)=='t'||(c)=='T'||(c)=='u'||(c)=='U'?'a':'n')
	for( sp = sbuf, cp = &sbuf[ slen - 1 ]; sp <= cp; sp++, cp-- ){
		c1 = WC_CMP( *sp );
		c2 = WC_CMP( *cp );
		*sp = c2;
		*cp = c1;
	}
#undef WC_CMP
}

int	rmo_scan( const rma_program_t *p, const rma_efndata_t *ed, int seq_index,
	const char *sbuf, int slen, int comp, rmo_hits_t *hits )	/* RM_find_motif :164 */
{
	ctx_t	*c;
	int	w_winsize, l_szero, i, szero, rv;

	c = ( ctx_t * )calloc( 1, sizeof( ctx_t ) );
	c->p = p;
	c->ed = ed;
	c->sbuf = sbuf;
	c->slen = slen;
	c->comp = comp;
	c->seq = seq_index;
	c->hits = hits;
	c->windowsize = p->windowsize;
	for( i = 0; i < 256; i++ )
		c->b2bc[ i ] = RMA_BC_N;
	c->b2bc[ 'a' ] = c->b2bc[ 'A' ] = RMA_BC_A;
	c->b2bc[ 'c' ] = c->b2bc[ 'C' ] = RMA_BC_C;
	c->b2bc[ 'g' ] = c->b2bc[ 'G' ] = RMA_BC_G;
	c->b2bc[ 't' ] = c->b2bc[ 'T' ] = RMA_BC_T;
	c->b2bc[ 'u' ] = c->b2bc[ 'U' ] = RMA_BC_T;
	for( i = 0; i < RMA_MAX_ELEMS; i++ ){
		c->moff[ i ] = c->mlen[ i ] = UNDEF;
		c->mpr[ i ] = c->mm[ i ] = UNDEF;
		c->zero[ i ] = c->dollar[ i ] = UNDEF;
	}
	c->l_mm = c->r_mm = UNDEF;

	w_winsize = p->dmaxlen < p->windowsize ? p->dmaxlen : p->windowsize;
	c->winbuf = ( int * )malloc( ( ( size_t )w_winsize + p->windowsize + 4 ) * sizeof( int ) );
	for( i = 0; i < w_winsize + p->windowsize + 4; i++ )
		c->winbuf[ i ] = UNDEF;
	c->window = &c->winbuf[ 1 ];

	l_szero = slen - w_winsize;
	for( szero = 0; szero < l_szero && !c->err; szero++ ){
		c->szero = szero;
		c->zero[ 0 ] = szero;
		c->dollar[ 0 ] = MIN( szero + w_winsize - 1, slen - 1 );
		c->window[ -1 ] = UNDEF;
		c->window[ c->dollar[ 0 ] + 1 - szero ] = UNDEF;
		find_motif( c, 0 );
	}
	l_szero = slen - p->dminlen;
	c->dollar[ 0 ] = slen - 1;
	for( ; szero <= l_szero && !c->err; szero++ ){
		c->szero = szero;
		c->zero[ 0 ] = szero;
		find_motif( c, 0 );
	}
	rv = c->err ? -1 : 0;
	free( c->winbuf );
	free( c );
	return( rv );
}

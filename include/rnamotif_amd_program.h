/*
 * rnamotif_amd_program.h -- the flattened "motif program" and the hit record.
 *
 * This is the data half of the drop-in boundary around the reference's scan
 * path.  The reference hands RM_find_motif() its compiled descriptor through
 * process globals (rm_descr[], rm_searches[], rm_sites, rm_lctx/rm_rctx,
 * rm_dminlen/rm_dmaxlen; /root/reference/src/find_motif.c:17-43, types in
 * src/rnamot.h:145-274).  Here the same information is one position-independent
 * POD blob that can be copied to HBM as is, and what the scan leaves behind
 * (s_matchoff/s_matchlen/s_n_mispairs/s_n_mismatches, rnamot.h:235-238) is a
 * fixed-stride hit record.
 *
 * Plain C, no pointers inside the blob: every cross reference is an index.
 */
#ifndef RNAMOTIF_AMD_PROGRAM_H
#define RNAMOTIF_AMD_PROGRAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMA_MAGIC		0x524d4131u	/* "RMA1" */
#define RMA_UNDEF		(-1)
#define RMA_UNBOUNDED		0x7fffffff	/* rnamot.h:40 */
#define RMA_EFN_INFINITY	16000		/* rnamot.h:41 */

#define RMA_MAX_ELEMS		100		/* compile.c:49 RM_DESCR_SIZE */
#define RMA_MAX_SITES		16
#define RMA_MAX_EFN_SITES	16
#define RMA_MAX_RE		(RMA_MAX_ELEMS + 2)
#define RMA_MAX_RE_ATOMS	128
#define RMA_MAX_PAIRSETS	(RMA_MAX_ELEMS + RMA_MAX_SITES + 2)

/* base codes, rnamot.h:138-143 (every non-acgtu letter is RMA_BC_N) */
#define RMA_BC_A	0
#define RMA_BC_C	1
#define RMA_BC_G	2
#define RMA_BC_T	3
#define RMA_BC_N	4

/* element types (the reference uses its yacc token numbers, rmgrm.y:34-47) */
enum rma_type {
	RMA_T_CTX = 0,
	RMA_T_SS, RMA_T_H5, RMA_T_H3, RMA_T_P5, RMA_T_P3,
	RMA_T_T1, RMA_T_T2, RMA_T_T3,
	RMA_T_Q1, RMA_T_Q2, RMA_T_Q3, RMA_T_Q4,
	RMA_T_SE		/* score section only */
};

/* ends / strict attribute bits, rnamot.h:210-214 */
#define RMA_5PAIRED	01
#define RMA_3PAIRED	02
#define RMA_5STRICT	01
#define RMA_3STRICT	02

/* One atom of a compiled seq= constraint.  The reference's ed-style byte code
 * (regexp.c:74-88) is reduced to what can occur on a nucleotide alphabet:
 * a 5-bit class over {a,c,g,t,other} and a repeat count.  hi == 255 means
 * unbounded (regexp.c getrnge: sizecode 255). */
typedef struct rma_re_atom {
	uint8_t	mask;	/* bit c set: base code c is accepted		*/
	uint8_t	lo;	/* minimum repeats				*/
	uint8_t	hi;	/* maximum repeats, 255 = no limit		*/
	uint8_t	kind;	/* 0 chr, 1 dot, 2 class, 3 negated class	*/
} rma_re_atom_t;

typedef struct rma_regex {
	int32_t	anchored;	/* seq[0] == '^' (circf, find_motif.c:1818)	*/
	int32_t	dollar;		/* trailing $ (CDOL)				*/
	int32_t	n_atoms;
	int32_t	fixed_len;	/* sum of lo if every atom has lo == hi, else -1 */
	/* 1: the atoms accept MORE than the expression does -- a back reference \1 stands as ".*", a literal or class
	 * member that is not one of acgt (iupac = 0) as "any letter that is not acgt", \< and \> as nothing: what the
	 * packed database can tell.  The scan's records are then candidates by a necessary condition; the host applies
	 * the whole expression to the text when it replays them (rma_replay_*, Replayer::one_hit), as chk_seq() would
	 * have (find_motif.c:1810; step / advance, regexp.c:389-664). */
	int32_t	loose;
	rma_re_atom_t	atoms[ RMA_MAX_RE_ATOMS ];
} rma_regex_t;

/* Pairing tables as bit sets over 5^n base code tuples
 * (BP_MAT_T/BT_MAT_T/BQ_MAT_T, rnamot.h:145-147).  Index = ((b1*5+b2)*5+b3)*5+b4.
 * For 3- and 4-base sets mat2 is the reference's "rbmat" (first,last) projection
 * (compile.c:2581-2625) used by match_wchlx/match_phlx on the outer strands. */
typedef struct rma_pairset {
	int32_t		n_bases;	/* 2, 3 or 4 */
	uint32_t	mat2;		/* 25 bits  */
	uint32_t	mat3[ 4 ];	/* 125 bits */
	uint32_t	mat4[ 20 ];	/* 625 bits */
} rma_pairset_t;

typedef struct rma_elem {
	int32_t	type;			/* enum rma_type		*/
	int32_t	proper;			/* s_attr[SA_PROPER]		*/
	int32_t	ends;			/* s_attr[SA_ENDS]		*/
	int32_t	strict;			/* s_attr[SA_STRICT]		*/
	int32_t	index;			/* s_index			*/
	int32_t	searchno;		/* s_searchno or -1		*/
	int32_t	next, prev, inner, outer;	/* element indices or -1 */
	int32_t	n_mates;
	int32_t	mates[ 3 ];
	int32_t	n_scopes;
	int32_t	scope;			/* s_scope or -1		*/
	int32_t	scopes[ 8 ];		/* s_scopes[] as element indices; pknots are
					 * limited to 4 helices on this build	*/
	int32_t	minlen, maxlen;
	int32_t	minglen, maxglen;
	int32_t	minilen, maxilen;
	int32_t	mismatch;
	int32_t	mispair;
	double	pairfrac;
	int32_t	pairset;		/* index into pairsets[] or -1	*/
	int32_t	re;			/* index into regexes[] or -1	*/
} rma_elem_t;

typedef struct rma_site_pos {
	int32_t	elem;			/* p_descr as element index	*/
	int32_t	l2r;			/* a_l2r			*/
	int32_t	offset;			/* a_offset			*/
} rma_site_pos_t;

typedef struct rma_site {
	int32_t	n_pos;
	rma_site_pos_t	pos[ 4 ];
	int32_t	pairset;
} rma_site_t;

/* One static efn() call site of the score program (score.c:1567-1682):
 * elements idx..idx2 (0-based into the descriptor), pos/pos2 already resolved
 * the way do_sc_efnx does (pos 0-based; pos2 < 0 means "last base of idx2"). */
typedef struct rma_efn_site {
	int32_t	idx, pos, idx2, pos2;
	int32_t	kind;			/* 0: efn() (efn.c:1162), 1: efn2() (efn2.c:1103) */
} rma_efn_site_t;
#define RMA_EFN_KIND_EFN	0
#define RMA_EFN_KIND_EFN2	1

typedef struct rma_program {
	uint32_t	magic;
	uint32_t	size;			/* sizeof( rma_program_t )	*/
	int32_t	n_elems;
	int32_t	n_searches;
	int32_t	searches[ RMA_MAX_ELEMS ];	/* s_descr of each SEARCH_T	*/
	int32_t	dminlen, dmaxlen;		/* rm_dminlen / rm_dmaxlen	*/
	int32_t	windowsize;			/* find_motif.c:114-128		*/
	int32_t	strict_helices;			/* rm_args->a_strict_helices	*/
	int32_t	chk_both_strs;			/* rnamot.c:113-117		*/
	int32_t	has_lctx, has_rctx;		/* rm_lctx / rm_rctx != NULL	*/
	rma_elem_t	lctx, rctx;
	rma_elem_t	elems[ RMA_MAX_ELEMS ];
	int32_t	n_sites;
	rma_site_t	sites[ RMA_MAX_SITES ];
	int32_t	n_pairsets;
	rma_pairset_t	pairsets[ RMA_MAX_PAIRSETS ];
	int32_t	n_regexes;
	rma_regex_t	regexes[ RMA_MAX_RE ];
	int32_t	n_efn_sites;
	rma_efn_site_t	efn_sites[ RMA_MAX_EFN_SITES ];
	int32_t	efn_usestdbp;			/* score.c:1591-1592		*/
	int32_t	efn_stdbp;			/* pairset index of efn_stdbp	*/
} rma_program_t;

/* Nearest-neighbour energy tables, the in-memory form of EFNDATA_T
 * (efn.c:54-88) in integer 1/100 kcal/mol.  loginc[n] holds the reference's
 * NINT( prelog*log( n/30. ) ) (efn.c:1375,1384,1572) for loop sizes up to the
 * window, computed once on the host in the reference's float arithmetic so
 * the device never evaluates log(). */
#define RMA_EFN_MAXLOOP	30
#define RMA_EFN_LOGINC	8192
typedef struct rma_efndata {
	int32_t	inter[ 31 ], bulge[ 31 ], hairpin[ 31 ];
	int32_t	dangle[ 5 ][ 5 ][ 5 ][ 2 ];
	int32_t	maxpen;
	int32_t	poppen[ 5 ];
	int32_t	eparam[ 16 ];
	int32_t	auend, gubonus, cslope, cint, c3, init, gail;
	int32_t	asint1x2[ 6 ][ 6 ][ 5 ][ 5 ][ 5 ];
	int32_t	sint2[ 6 ][ 6 ][ 5 ][ 5 ];
	int32_t	sint4[ 6 ][ 6 ][ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	tloops[ 100 ][ 2 ];
	int32_t	ntloops;
	int32_t	triloops[ 50 ][ 2 ];
	int32_t	ntriloops;
	int32_t	stack[ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	tstkh[ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	tstki[ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	loginc[ RMA_EFN_LOGINC ];
	float	prelog;
} rma_efndata_t;

/* Tables of efn2(), the in-memory form of EFN2DATA_T (efn2.c:22-66), integer 1/100
 * kcal/mol with EFN2_INFINITY = 9999999.  loginc[n] = (int)( prelog*log( n/30. ) )
 * (truncated, efn2.c:1570,1583,1650) and mbl_log[n] = (int)( 11.*log( n/6. ) + 0.5 )
 * (efn2.c:1209) are computed on the host in the reference's arithmetic. */
#define RMA_EFN2_INFINITY	9999999
#define RMA_EFN2_MAXTLOOP	100
typedef struct rma_efn2data {
	int32_t	inter[ 31 ], bulge[ 31 ], hairpin[ 31 ];
	int32_t	dangle[ 5 ][ 5 ][ 5 ][ 2 ];
	int32_t	maxpen;
	int32_t	poppen[ 5 ];
	int32_t	eparam[ 11 ];
	int32_t	efn2a, efn2b, efn2c, auend, gubonus, cslope, cint, c3, init, gail;
	int32_t	tloop[ RMA_EFN2_MAXTLOOP + 1 ][ 2 ];	/* 1-based, base-5 keys (efn2.c:1050) */
	int32_t	ntloops;
	int32_t	triloop[ RMA_EFN2_MAXTLOOP + 1 ][ 2 ];
	int32_t	ntriloops;
	int32_t	stack[ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	tstkh[ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	tstki[ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	coax[ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	tstackcoax[ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	coaxstack[ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	tstack[ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	tstkm[ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	iloop11[ 5 ][ 5 ][ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	iloop21[ 5 ][ 5 ][ 5 ][ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	iloop22[ 5 ][ 5 ][ 5 ][ 5 ][ 5 ][ 5 ][ 5 ][ 5 ];
	int32_t	loginc[ RMA_EFN_LOGINC ];
	int32_t	mbl_log[ RMA_EFN_LOGINC ];
	float	prelog;
} rma_efn2data_t;

/* Hit record: int32 words, stride = rma_hit_stride( prog ).
 *   [0] seq     index of the sequence in the batch
 *   [1] comp    0 = strand as given, 1 = reverse complement
 *   [2] szero   start offset of the scan position (fm_szero)
 *   [3] rank    which end position of the first search element this hit
 *               came from: 0 for the largest (tried first), counting up
 *               (find_motif.c:273 loops sdollar downwards)
 *   [4] order   emission counter within (seq,comp,szero,rank); hits sorted
 *               by (seq,comp,szero,rank,order) are in the reference's output
 *               order (find_motif.c:184-205,273,435,523,600,821,938)
 *   [5 + 4*e .. ] per element e: matchoff, matchlen, n_mispairs, n_mismatches
 *   then lctx off,len, rctx off,len (0,0 when absent)
 *   then one word per efn site: energy in 1/100 kcal/mol (RM_efn's int)
 */
#define RMA_HIT_HDR	5
static inline int rma_hit_stride( const rma_program_t *p )
{
	return( RMA_HIT_HDR + 4 * p->n_elems + 4 + p->n_efn_sites );
}
static inline int rma_hit_ctx_off( const rma_program_t *p )
{
	return( RMA_HIT_HDR + 4 * p->n_elems );
}
static inline int rma_hit_efn_off( const rma_program_t *p )
{
	return( RMA_HIT_HDR + 4 * p->n_elems + 4 );
}

#ifdef __cplusplus
}
#endif
#endif

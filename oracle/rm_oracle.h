/*
 * rm_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's scan path (find_motif.c, the
 * step()/mm_step() matchers, efn.c) over the flattened motif program of
 * include/rnamotif_amd_program.h.  It exists so that the HIP scanner can be
 * checked record by record.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use anything in this directory.
 *
 * Pinning: tests/test_golden_stdout.py runs the host front end + this oracle
 * over the reference's own test database and compares the raw output with the
 * md5 sums recorded from the reference and, through the
 * reference's own rmfmt (oracle/_ref), with the reference's test .chk files;
 * the energy function is compared with the reference's efn_drv (oracle/_ref).
 */
#ifndef RM_ORACLE_H
#define RM_ORACLE_H

#include <stdint.h>
#include "rnamotif_amd_program.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rmo_hits {
	int32_t	*data;		/* n * stride words, malloc'ed, grows	*/
	int64_t	n, cap;
	int	stride;
} rmo_hits_t;

void	rmo_hits_init( rmo_hits_t *h, const rma_program_t *p );
void	rmo_hits_free( rmo_hits_t *h );

/* RM_find_motif( ..., comp, slen, sbuf ) for one strand of one sequence
 * (find_motif.c:164).  sbuf is what the reference's caller passes: lower case
 * letters, for comp = 1 already reverse complemented (rmo_revcomp).  ed may be
 * NULL when the program has no efn sites.  Returns 0, or -1 on an internal
 * limit (helix candidate list overflow). */
int	rmo_scan( const rma_program_t *p, const rma_efndata_t *ed, int seq_index,
		const char *sbuf, int slen, int comp, rmo_hits_t *hits );

/* mk_rcmp(), rnamot.c:193-216 */
void	rmo_revcomp( char *sbuf, int slen );

/* RM_getefndata(), efn.c:157-918; returns 1 on success like the reference */
int	rmo_load_efndata( const char *dir, rma_efndata_t *ed );

/* RM_efn( 0, l_base, 1 ) on bcseq/basepr[0..l_base], efn.c:1162 */
int	rmo_efn( const rma_efndata_t *ed, const int *bcseq, const int *basepr, int l_base );

/* RM_efn2() on bcseq/basepr[0..l_base] (basepr -1 = unpaired), efn2.c:1103; tables as
 * the product's loader fills them (rma_efn2data_load).  *undefined is set, and
 * RMA_EFN2_INFINITY returned, where the reference would index outside its arrays. */
void	rmo_set_efn2data( const rma_efn2data_t *ed );	/* used by rmo_scan for efn2() sites */
int	rmo_efn2( const rma_efn2data_t *ed, const int *bcseq, const int *basepr, int l_base, int *undefined );

#ifdef __cplusplus
}
#endif
#endif

/*
 * rnamotif_amd.h -- C ABI of the MI355X scan path.
 *
 * The reference has no plug-in API; the boundary this library replaces is the
 * set of calls main() makes around the scan (/root/reference/src/rnamot.c:49-188,
 * prototypes in /root/reference/src/rnamot.h:305-381):
 *
 *   RM_init() + yyparse() + SE_link() + RM_linkscore()   -> rma_descr_compile()
 *   RM_fm_init()                  (find_motif.c:109)      -> rma_scanner_create()
 *   FN_fgetseq() into sbuf        (dbutil.c:42)           -> rma_db_create()
 *   RM_find_motif() x2 per entry  (find_motif.c:164)      -> rma_scan()
 *   RM_score() + print_match()    (score.c:608,
 *                                  find_motif.c:1826)     -> rma_replay_*()
 *
 * Plain C: pointers, sizes, integer status codes.  Every function that can
 * fail returns 0 on success and non-zero with a message in err[] otherwise;
 * nothing in the library calls exit().  The scan runs on the GPU only: there is
 * no CPU fallback, rma_scanner_create() fails when no HIP device is usable.
 */
#ifndef RNAMOTIF_AMD_H
#define RNAMOTIF_AMD_H

#include <stddef.h>
#include <stdint.h>
#include "rnamotif_amd_program.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rma_descr	rma_descr_t;	/* compiled descriptor + score program (host)	*/
typedef struct rma_scanner	rma_scanner_t;	/* motif program + buffers on one GPU		*/
typedef struct rma_db		rma_db_t;	/* packed sequences resident in HBM		*/
typedef struct rma_replay	rma_replay_t;	/* score VM + hit printer state			*/

const char	*rma_version( void );
int	rma_device_count( void );

/* ---- descriptor front end (host).  argv is the rnamotif command line
 * (argv[0] = program name, "-descr file", "-Dname=value", "-sh", "-context", ...);
 * database file names in it are remembered for rma_descr_dbfiles(). */
int	rma_descr_compile( int argc, const char *const *argv, rma_descr_t **out, char *err, size_t errlen );
void	rma_descr_free( rma_descr_t *d );
const rma_program_t	*rma_descr_program( const rma_descr_t *d );
const rma_efndata_t	*rma_descr_efndata( const rma_descr_t *d );	/* NULL: no efn() in the score section */
const rma_efn2data_t	*rma_descr_efn2data( const rma_descr_t *d );	/* NULL: no efn2() in the score section */
int	rma_descr_minlen( const rma_descr_t *d );			/* rm_dminlen */
int	rma_descr_maxlen( const rma_descr_t *d );			/* rm_dmaxlen, RMA_UNBOUNDED if open */
/* for bindings that do not want to mirror the struct: n_elems, n_searches, hit stride,
 * ctx offset, efn offset, n_efn_sites, chk_both_strs, windowsize */
void	rma_program_info( const rma_program_t *prog, int32_t info[ 8 ] );

/* ---- energy tables on their own: RM_getefndata() efn.c:157 / RM_getefn2data() efn2.c:130
 * from the directory dir (the reference's efn_datadir / $EFNDATA). */
int	rma_efndata_load( const char *dir, rma_efndata_t *out, char *err, size_t errlen );
int	rma_efn2data_load( const char *dir, rma_efn2data_t *out, char *err, size_t errlen );

/* ---- scanner.  prog (and efn, may be NULL) are copied. */
int	rma_scanner_create( const rma_program_t *prog, const rma_efndata_t *efn, int device,
		rma_scanner_t **out, char *err, size_t errlen );
/* tables for the program's efn2() sites (RM_getefn2data, efn2.c:130); copied to the device */
int	rma_scanner_set_efn2data( rma_scanner_t *sc, const rma_efn2data_t *efn2, char *err, size_t errlen );
/* Launch-shape and diagnostic switches (DESIGN.md has the table).  The RNAMOTIF_* environment is read
 * once, by rma_scanner_create(); the switches that may change between scans change through this call
 * only: "dbg", "pool", "pool_min", "pool_refill", "drain", "glist", "drain_waves", "search_wgs", "flush", "efn_light", "host_sort", "timing", "short".
 * None of them changes the records a scan returns. */
int	rma_scanner_set_option( rma_scanner_t *sc, const char *name, int value, char *err, size_t errlen );
/* One scan of eight start positions, thrown away: what the runtime sets up on first use (code objects,
 * the first allocations, the ordering's kernels) is paid here and not in the first batch of a search.
 * The command line program calls it once the scanner is complete; the library never does by itself. */
int	rma_scanner_warmup( rma_scanner_t *sc, char *err, size_t errlen );
void	rma_scanner_destroy( rma_scanner_t *sc );

/* ---- database: n sequences of lower case letters as the reference's readers
 * deliver them (dbutil.c: every alpha character kept, u -> t).  They are packed
 * 2 bits + 1 ambiguity bit per base and uploaded; the host text is not kept.  A database
 * lives on the device of the scanner named at its creation (sc may be NULL: device 0) and holds
 * nothing that depends on a descriptor: every scanner of that device can scan it, side by side if
 * they like (rma_db_attach, rma_scan_begin) -- one upload, many descriptors.  Device memory of a
 * destroyed database is kept for the next one of about its size (no allocation per batch). */
int	rma_db_create( rma_scanner_t *sc, const char *const *seqs, const int32_t *slens, int32_t n,
		rma_db_t **out, char *err, size_t errlen );
/* The same, answering only for start positions pos_lo[i] <= szero < pos_hi[i] of each strand
 * of entry i (RM_find_motif's szero, find_motif.c:184-205): a long entry can be searched by
 * several devices, each holding the whole entry and a slice of its start positions; the union
 * of the hit records, sorted by their first five words, is the whole entry's hit list. */
int	rma_db_create_ranges( rma_scanner_t *sc, const char *const *seqs, const int32_t *slens,
		const int32_t *pos_lo, const int32_t *pos_hi, int32_t n,
		rma_db_t **out, char *err, size_t errlen );
void	rma_db_destroy( rma_db_t *db );
int64_t	rma_db_bases( const rma_db_t *db );
/* Lay db out for scanner sc (the tiling of its entries for that descriptor's tile size) ahead of sc's
 * first scan of it, which would otherwise do it.  A database may be attached to any number of scanners
 * of its device. */
int	rma_db_attach( rma_scanner_t *sc, rma_db_t *db, char *err, size_t errlen );
/* wait until the upload of db is complete (rma_db_create_packed_async) */
int	rma_db_wait( rma_db_t *db, char *err, size_t errlen );

/* ---- packed database on disk (no counterpart in the reference, which re-reads the text
 * on every run, rnamot.c:157-183): the readers' output -- names, definition lines and
 * sequences as FN_/PIR_/GB_fgetseq deliver them (dbutil.c:42,130,226) -- stored in the
 * layout rma_db_create() makes, plus the letters at ambiguous positions, so that a scan
 * can start from it directly and print_match()'s text can still be rebuilt. */
typedef struct rma_pack	rma_pack_t;
int	rma_pack_write( const char *path, const char *const *sids, const char *const *sdefs,
		const char *const *seqs, const int32_t *slens, int32_t n, char *err, size_t errlen );
int	rma_pack_open( const char *path, rma_pack_t **out, char *err, size_t errlen );
void	rma_pack_close( rma_pack_t *pk );
int32_t	rma_pack_count( const rma_pack_t *pk );
int64_t	rma_pack_bases( const rma_pack_t *pk );
const char	*rma_pack_sid( const rma_pack_t *pk, int32_t i );
const char	*rma_pack_sdef( const rma_pack_t *pk, int32_t i );
int32_t	rma_pack_slen( const rma_pack_t *pk, int32_t i );
/* buf must hold rma_pack_slen( pk, i ) + 1 bytes */
int	rma_pack_seq( const rma_pack_t *pk, int32_t i, char *buf );
/* entries [first, first+count) straight into HBM; hit records count entries from first */
int	rma_db_create_packed( rma_scanner_t *sc, const rma_pack_t *pk, int32_t first, int32_t count,
		rma_db_t **out, char *err, size_t errlen );
/* The same without waiting for the copies: they run on the device's upload stream, under whatever the
 * scanners' streams are doing; a scan of the database waits for them on the device.  The pack must
 * stay as it is until rma_db_wait() or the end of a scan of the database.  From page-locked memory
 * (rma_pack_pin) the call returns at once; from pageable memory the runtime stages the words first. */
int	rma_db_create_packed_async( rma_scanner_t *sc, const rma_pack_t *pk, int32_t first, int32_t count,
		rma_db_t **out, char *err, size_t errlen );
/* page-lock the packed words of pk so that uploads from it are plain DMA (undone by rma_pack_close) */
int	rma_pack_pin( rma_pack_t *pk, char *err, size_t errlen );
/* The same database in memory, read from sequence files the way rnamotif reads them (DB_fnext,
 * dbutil.c:12-40; fmt "fastn" | "pir" | "gb" or NULL; maxslen = rnamotif's -N, 0 for its default):
 * FASTA files through the parallel reader, everything else -- and every entry a reader has a
 * diagnostic for -- through the restatements of FN_/PIR_/GB_fgetseq, which print what the
 * reference prints.  Files that are packed databases already are appended as they are. */
int	rma_pack_read( const char *const *paths, int32_t n_paths, const char *fmt, int32_t maxslen, int32_t threads,
		rma_pack_t **out, char *err, size_t errlen );
/* A rank's share of a database without reading the rest (one process per GPU).
 * rma_database_index(): the entries of the files in order and an upper bound of each one's length --
 * its bytes in a FASTA file ('>' found by all threads, nothing parsed), its length in a packed
 * database -- for the ranks to divide among themselves; *extent is to be released with rma_free().
 * rma_pack_read_entries(): the entries with the given numbers (ascending), read and packed; the
 * other entries' bytes are not touched.  Both return 2, with nothing made, when the files can only be
 * read whole -- -fmt pir | gb, a file that cannot be mapped, an entry the serial reader has a
 * diagnostic for: the caller then reads everything with rma_pack_read(). */
int	rma_database_index( const char *const *paths, int32_t n_paths, const char *fmt, int32_t threads,
		int64_t **extent, int32_t *n_entries, char *err, size_t errlen );
int	rma_pack_read_entries( const char *const *paths, int32_t n_paths, const char *fmt, int32_t maxslen, int32_t threads,
		const int32_t *entry, int32_t n, rma_pack_t **out, char *err, size_t errlen );
void	rma_free( void *p );
/* Any n entries of a packed database, entry[i] with start positions pos_lo[i] <= szero < pos_hi[i]
 * (NULL: all), straight into HBM: what one rank of a multi-GPU search takes of a database every
 * rank has read.  Hit records number the entries 0 .. n-1 in the order given. */
int	rma_db_create_packed_ranges( rma_scanner_t *sc, const rma_pack_t *pk, const int32_t *entry,
		const int32_t *pos_lo, const int32_t *pos_hi, int32_t n, rma_db_t **out, char *err, size_t errlen );

/* ---- scan every sequence of db (both strands when the program says so).
 * *hits receives *n_hits records of rma_hit_stride( prog ) words, sorted by
 * (seq, comp, szero, rank, order) = the reference's output order; the memory
 * belongs to the scanner and is valid until its next scan or destruction.
 * Energies of the program's efn sites are filled in. */
int	rma_scan( rma_scanner_t *sc, const rma_db_t *db, const int32_t **hits, int64_t *n_hits,
		char *err, size_t errlen );

/* rma_scan() in two halves.  rma_scan_begin() puts the search kernel on the scanner's stream and
 * returns; rma_scan_end() waits for it, runs the energy kernel and the ordering, copies the records
 * back and returns them as rma_scan() does.  Between the two the host is free: to begin the scan of
 * another scanner (two descriptors over one database run side by side), or to upload the next
 * database (rma_db_create_packed_async).  One scan in flight per scanner. */
int	rma_scan_begin( rma_scanner_t *sc, const rma_db_t *db, char *err, size_t errlen );
int	rma_scan_end( rma_scanner_t *sc, const int32_t **hits, int64_t *n_hits, char *err, size_t errlen );
/* rma_scan_end() that leaves the ordered records in HBM (*d_hits is a device pointer, valid until
 * the scanner's next scan): for rma_gather_hits(), which sends them from there. */
int	rma_scan_end_on_device( rma_scanner_t *sc, const int32_t **d_hits, int64_t *n_hits, char *err, size_t errlen );

/* The device part of rma_scan() alone (search kernel + efn kernel, no copy back,
 * no sort), for measurement: returns the candidate count and the time of the
 * search kernel as measured with HIP events on the scanner's stream. */
int	rma_scan_device( rma_scanner_t *sc, const rma_db_t *db, int64_t *n_hits, float *search_ms,
		float *efn_ms, char *err, size_t errlen );
/* The kernels of the scanner's last search (HIP events on its stream): ms[0] the search kernel, ms[1] the
 * drain kernel that walked the items the search kernel left in its list (0: the search kernel walked
 * them itself), ms[2] the efn kernel (0: none).  search_ms above is ms[0] + ms[1]. */
int	rma_scanner_last_kernel_ms( rma_scanner_t *sc, float ms[ 3 ], char *err, size_t errlen );

/* Records from several scans (the slices of one database searched by several GPUs or hosts, word 0
 * already the database-wide entry number) into the reference's output order: by the five header
 * words, ties in the order given; the order word is renumbered within (entry, strand, start, rank).
 * out[n][stride] must not overlap hits.  Host only -- what rank 0 of mrnamotif does with the
 * MT_RESULT messages it receives (mrnamotif.c:733-760). */
int	rma_sort_hits( const int32_t *hits, int64_t n_hits, int32_t stride, int32_t *out, char *err, size_t errlen );

/* ---- the one exchange of a multi-GPU search: the records of every rank's last scan to one rank,
 * over RCCL (xGMI), device to device -- mrnamotif's MT_RESULT messages, mrnamotif.c:733-760 and
 * :898-917.  One process per GPU.  Rank 0 makes an id (rma_comm_unique_id) and hands it to the
 * others by whatever the job has (MPI_Bcast, a torch.distributed broadcast, a file); every rank then
 * calls rma_comm_create() -- collectively.  RCCL is loaded at run time (librccl.so.1); a world of
 * one rank needs none. */
#define RMA_COMM_ID_BYTES	128
typedef struct rma_comm	rma_comm_t;
int	rma_comm_unique_id( uint8_t id[ RMA_COMM_ID_BYTES ], char *err, size_t errlen );
int	rma_comm_create( const uint8_t id[ RMA_COMM_ID_BYTES ], int rank, int world, int device,
		rma_comm_t **out, char *err, size_t errlen );
void	rma_comm_destroy( rma_comm_t *comm );
/* The collective calls rma_gather_hits() makes, as a table.  rma_comm_create() fills it with RCCL's
 * (ncclAllGather of int64, ncclSend / ncclRecv of int32, ncclGroupStart / ncclGroupEnd, ncclCommCount); a caller
 * with another transport of the same semantics -- MPI between nodes where mrnamotif.c has MPI_Send / MPI_Recv, or
 * a test that makes a call fail -- hands its own to rma_comm_create_on().  Every call returns 0 or a code that
 * error_string() puts into words; stream is the hipStream_t the buffers are valid on; buffers are in HBM. */
typedef struct rma_transport {
	int	( *all_gather )( void *ctx, const void *send, void *recv, size_t n_int64_per_rank, void *stream );
	int	( *send )( void *ctx, const void *buf, size_t n_int32, int peer, void *stream );
	int	( *recv )( void *ctx, void *buf, size_t n_int32, int peer, void *stream );
	int	( *group_start )( void *ctx );
	int	( *group_end )( void *ctx );
	const char	*( *error_string )( void *ctx, int code );	/* may be null */
	int	( *comm_count )( void *ctx, int *count );		/* may be null */
	void	*ctx;
} rma_transport_t;
int	rma_comm_create_on( const rma_transport_t *transport, int rank, int world, int device,
		rma_comm_t **out, char *err, size_t errlen );
/* The number of ranks the transport itself reports (RCCL: ncclCommCount; a world of one: 1). */
int	rma_comm_count( rma_comm_t *comm, int *count, char *err, size_t errlen );
/* Collective.  Every rank has ended a scan of its shard with rma_scan_end_on_device() (or rma_scan_end:
 * the records are still in HBM).  global_index[ i ], i < n_index, is the number in the whole database
 * of entry i of this rank's shard: word 0 of the records is rewritten to it on the device (once per scan: a second
 * gather of the same records leaves them).  An all-gather of the counts and of a flag per rank (16 bytes per rank: a
 * rank that cannot take part -- records ordered on the host, no entry numbers -- says so there, and every rank
 * returns an error instead of waiting), a second one only when the root's buffers have to grow (could they?), then one
 * grouped send/receive, whose group is closed on every path: on rank `root`,
 * *hits / *n_hits are all records, rank by rank, each rank's part in the reference's order (when the
 * ranks hold consecutive runs of entries that is the whole job's order; otherwise rma_sort_hits()
 * merges); elsewhere *n_hits = 0.  counts, if not NULL, receives every rank's count on every rank.
 * The memory belongs to the communicator and is valid until its next gather. */
int	rma_gather_hits( rma_comm_t *comm, rma_scanner_t *sc, const int32_t *global_index, int32_t n_index, int root,
		const int32_t **hits, int64_t *n_hits, int64_t *counts, char *err, size_t errlen );

/* ---- replay: run the score program over candidates and print accepted hits
 * in the reference's format to a stdio stream opened on path ("-" = stdout). */
int	rma_replay_open( rma_descr_t *d, const char *path, rma_replay_t **out, char *err, size_t errlen );
/* one batch: the same sequences (ids, definition lines, text) the db was made from */
int	rma_replay_batch( rma_replay_t *rp, const char *const *sids, const char *const *sdefs,
		const char *const *seqs, const int32_t *slens, int32_t n,
		const int32_t *hits, int64_t n_hits, int64_t *n_printed, char *err, size_t errlen );
/* the same over a packed database: word 0 of a record is the entry's number in pk minus first; the
 * text of an entry is rebuilt for the span of each hit only */
int	rma_replay_pack( rma_replay_t *rp, const rma_pack_t *pk, int32_t first,
		const int32_t *hits, int64_t n_hits, int64_t *n_printed, char *err, size_t errlen );
int	rma_replay_close( rma_replay_t *rp, char *err, size_t errlen );	/* runs the END program */

#ifdef __cplusplus
}
#endif
#endif

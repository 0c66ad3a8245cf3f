// rm_kernels.h -- what the host half of the scanner (rm_scanner.cpp) and the kernel translation
// units (rm_scan_inst_*.hip, rm_scan_kernel.h) share: the views of a database and of the hit buffer
// a launch gets, the launch-shape constants, and one launcher per kernel instance.  The kernels
// are templates (rm_scan_kernel.h); every instance lives in a translation unit of its own so that
// they compile side by side.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include "rm_dev_program.h"

// ---------------------------------------------------------------- device views
struct DbView {
	const uint32_t	*codes, *amask;
	const int64_t	*base_off;	// [n_seq]   first base of sequence s (multiple of 32)
	const int32_t	*slen;		// [n_seq]
	const int64_t	*tile_start;	// [n_seq+1] prefix sum of tiles over sequences
	const int32_t	*tile_seq;	// [n_tiles] sequence of every tile (saves a search per tile)
	const int32_t	*tile_meta;	// [n_tiles][RMK_META_WORDS] one tile per workgroup pass: the tile's entry, strand, first start position ... in one line
	const int32_t	*pos_lo, *pos_hi;	// [n_seq] or null: only start positions lo <= szero < hi (each strand)
	int32_t	n_seq, strands, tile_t;
	int64_t	n_tiles;
	int64_t	concat_bases;		// > 0: the tiles lie over the concatenation of the entries, this many bases a strand (rm_scanner.cpp, Layout::concat)
};

struct HitBuf {
	int32_t	*hits;
	unsigned long long	*count;		// candidates found (may exceed cap)
	unsigned long long	*ticket;	// next tile
	int64_t	cap;
	unsigned	*spill;			// [gridDim.x][spill_cap] work queue items that did not fit the LDS queue
	int	spill_cap;
	unsigned	*pool;			// [glist_cap + gridDim.x * pool_cap][RMK_POOL_WORDS] pooled instance: items that passed the tile's tests
	int	pool_cap, pool_min;	// ... searched once pool_min of them have come together
	int	pool_refill;		// idle lanes of a wave that pop together
	// pooled instance with a drain kernel (glist_cap > 0): the device-wide list the workgroups' pools are flushed to
	// is the first glist_cap items at `pool`, the workgroups' own areas come after it; ticket[ RMK_GCTL - 1 ] counts
	// the items reserved in it (may exceed glist_cap), ticket[ RMK_GCTL ] those taken (rma_drain_kernel)
	int	glist_cap;
};

// words of a tile's line in DbView::tile_meta
enum { RMK_META_SEQ = 0, RMK_META_COMP, RMK_META_Z0, RMK_META_SLEN, RMK_META_OFF_LO, RMK_META_OFF_HI, RMK_META_POS_HI, RMK_META_PAD, RMK_META_WORDS };

// ---------------------------------------------------------------- launch-shape constants
#ifndef QCAP
#define QCAP		1024		// work queue entries per workgroup
#endif
// Lean path records of one lane in LDS, 6 bytes per level (LdsRecs, rm_scan_kernel.h)
#define LEAN_REC_BYTES	6
// General path records of one lane in LDS (rmd_grec_t, 12 bytes per level: LdsGRecs)
#define GEN_REC_BYTES	12
// queue of continuations of the general instance (LdsSplit)
#define DEEP_QUEUE	128
#ifndef FLUSH_WAVES_PER_SIMD
#define FLUSH_WAVES_PER_SIMD	5	// the pooled lean instance that walks nothing (rma_search_kernel, WALK = false): what it is compiled and tiled for ...
#endif
#ifndef TICKET_TILES
#define TICKET_TILES		4	// tiles a ticket is good for (rma_search_kernel, one tile per pass) ...
#endif
#ifndef TICKET_TAIL
#define TICKET_TAIL		3	// ... but for the last so many per workgroup, which go one by one
#endif
#ifndef DRAIN_WAVES_PER_SIMD
#define DRAIN_WAVES_PER_SIMD	4	// what the drain kernel is compiled for (its registers: 128 at four)
#endif
#ifndef FLUSH_WGS_PER_CU
#define FLUSH_WGS_PER_CU	5	// ... and the workgroups a CU gets of it (rma_scan_begin)
#endif
#ifndef SEARCH_WAVES_PER_SIMD
#define SEARCH_WAVES_PER_SIMD	4
#endif
#ifndef GENERAL_WAVES_PER_SIMD
#define GENERAL_WAVES_PER_SIMD	3
#endif
#ifndef RMD_KIND_PK
#define RMD_KIND_PK	1	// improper (pseudoknot) helices
#define RMD_KIND_TQ	2	// parallel helices, triplexes, 4-plexes
#define RMD_KIND_WIDE	4	// helices of 64 to 127 base pairs
#endif
// (... three with 168 registers for descriptors with triplexes / 4-plexes: qu+tr 46.4 -> 39.2 ms, where
// pk1 went 7.5 -> 8.6 ms in round 2, with workgroups of four waves.  With workgroups of one wave three and four
// measure alike -- pk1 3.61 / 3.63 ms, pk_j1+2 67.4 / 68.3 -- and three spill 7 registers where four spill 80:
// the general instances' scratch traffic is what the fourth wave cost.)
#define GENERAL_WAVES( kinds_ )	( ( ( kinds_ ) & RMD_KIND_TQ ) ? 3 : GENERAL_WAVES_PER_SIMD )
#ifndef SHORT_GROUP
#define SHORT_GROUP		16	// tiles per workgroup pass for databases of short entries
#endif
#define SHORT_ENTRY_MEAN	4000	// ... which are those whose entries average less than this
#define SPILL_ITEMS		8192	// queue items per workgroup that may overflow into HBM (32 KB each, 64 MB in all)
#define RMK_POOL_WORDS		5	// entry, start, rank | strand, and the 3' ends the first helix of the interior may take (two outer lengths)
#define RMK_N_COUNTERS		128	// 64-bit counters behind a launch: [0] candidates, [1] ticket, diagnostics, [RMK_GCTL] the list's two
#define RMK_GCTL		100
#define PIECE_ORDER_BITS	12	// candidates a piece of an item may find; more, and the search is repeated with whole items (ticket[ 2 ])
#define GLIST_BELOW		128	// what a workgroup's pool holds at the end goes to the list when it is less than this
#define SEARCH_BLOCK		256	// lanes of a search workgroup of the lean instances
// ... and of the general instances: ONE wave.  The waves of a four-wave workgroup met at the end of
// every tile, and the tile's second round -- a few dozen continuations, each a long walk -- kept one of
// them busy while three waited: 23 % (pk1), 44 % (qu+tr) and 32 % (pk_j1+2) of all wave cycles,
// RNAMOTIF_DBG=34.  A wave that is its own workgroup waits for nobody; tiles are a quarter the size.
#ifndef GENERAL_BLOCK
#define GENERAL_BLOCK		64
#endif
#define EFN_BLOCK		256	// lanes of an efn workgroup

// ---------------------------------------------------------------- kernel instances
// Which instance of rma_search_kernel a launch takes (rma_scan_device picks it from the descriptor
// and the database's shape).
enum rmk_instance {
	RMK_LEAN_POOL = 0,	// lean, pass B over a pool of survivors (the headline instance)
	RMK_LEAN_CONCAT,	// ... with the tiles over the concatenation of the entries (databases of short entries)
	RMK_LEAN_FLUSH,		// lean, every survivor of pass A' straight to the drain kernel's list: the search kernel walks nothing
	RMK_LEAN_CONCAT_FLUSH,	// ... with the tiles over the concatenation of the entries
	RMK_LEAN_GROUP,		// lean, groups of SHORT_GROUP small tiles (databases of short entries)
	RMK_LEAN_TILE,		// lean, pass B tile by tile
	RMK_GEN_PLAIN,		// general, no pseudoknot, no triplex / 4-plex
	RMK_GEN_PK,
	RMK_GEN_TQ,
	RMK_GEN_PKTQ,
	RMK_GEN_WIDE,		// every element kind, helices of up to 127 base pairs (two-word sets of lengths)
	RMK_GEN_PLAIN_CONCAT,	// the general instances over tiles that lie over the concatenation of the entries
	RMK_GEN_PK_CONCAT,
	RMK_GEN_TQ_CONCAT,
	RMK_GEN_PKTQ_CONCAT,
	RMK_N_INSTANCES
};

struct rmk_search_args {
	const rmd_program_t	*d_prog;
	int	prog_bytes, qcap;
	DbView	db;
	HitBuf	hb;
	int	tile_bytes, dbg;
};

// Launch instance `inst` with `grid` workgroups of SEARCH_BLOCK lanes and lds bytes of dynamic LDS on s.
hipError_t	rmk_launch_search( int inst, int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
// the launchers behind it, one translation unit each (rm_scan_inst_*.hip)
hipError_t	rmk_launch_lean_pool( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_lean_concat( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_lean_flush( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_lean_concat_flush( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_lean_drain( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );	// (qcap: not used; tile_bytes: window dwords per lane)
hipError_t	rmk_launch_lean_group( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_lean_tile( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_gen_plain( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_gen_pk( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_gen_tq( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_gen_pktq( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_gen_wide( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_gen_plain_concat( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_gen_pk_concat( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_gen_tq_concat( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );
hipError_t	rmk_launch_gen_pktq_concat( int grid, size_t lds, hipStream_t s, const rmk_search_args &a );

struct rmk_efn_args {
	const rmd_program_t	*d_prog;
	DbView	db;
	int32_t	*hits;
	long long	n_hits;
	const int16_t	*t16;		// efn's tables as int16 (RME_N16 entries, padded to 8), or null
	const int32_t	*tlkey, *loginc;
	const rma_efn2data_t	*e2;	// efn2's tables, or null
};
hipError_t	rmk_launch_efn( int grid, hipStream_t s, const rmk_efn_args &a );
hipError_t	rmk_launch_efn_light( int grid, hipStream_t s, const rmk_efn_args &a );	// (grid: workgroups of 64 lanes)
hipError_t	rmk_launch_efn_big( int grid, hipStream_t s, const rmk_efn_args &a );	// (calls over more than 15 helices: rmd_program_t::efn_big)
hipError_t	rmk_preload_efn( void );

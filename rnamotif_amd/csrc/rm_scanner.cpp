// rm_scanner.cpp -- the scanner / database half of the C ABI (include/rnamotif_amd.h) on the
// host: device memory, streams, launch shapes.  The kernels are in rm_scan_kernel.h, reached
// through the launchers of rm_kernels.h.
//
// What belongs to whom:
//   DevCtx (one per GPU)  the upload stream and a cache of device blocks: a database that is
//                         destroyed gives its block back, the next one of about that size takes
//                         it -- no hipMalloc / hipFree per database (the loop over batches of the
//                         command line, rnamot.c:158-185, makes one per batch)
//   rma_db                the packed bases of some entries in HBM -- codes, ambiguity mask,
//                         offsets, lengths, start-position ranges: nothing in it depends on a
//                         descriptor -- plus, per launch shape that has scanned it, the tiling
//                         (tile_start / tile_seq), made when a scanner first meets the database
//   rma_scanner           the motif program of one descriptor on one GPU, its stream, hit buffer,
//                         work areas and ordering stage
// A database is uploaded on the device's upload stream; a scan waits for that on its own stream
// (an event), so the upload of the next database runs under the scan of this one.
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#define RMD_FN		static inline
#define RMD_FN_MEMBER	inline
#include "rm_kernels.h"
#include "rm_efn_core.h"
#include "rm_efndata.h"
#include "rm_fasta.h"
#include "rm_pack.h"
#include "rm_hitsort.h"
#include "rm_hitsort_dev.h"
#include "rnamotif_amd.h"

#define HIPCHK( call )	do{ hipError_t e_ = ( call ); if( e_ != hipSuccess ){ \
		snprintf( err, errlen, "%s: %s", #call, hipGetErrorString( e_ ) ); return 1; } }while( 0 )

namespace {

// ---------------------------------------------------------------- per-device context
struct Block {
	void	*p = nullptr;
	size_t	bytes = 0;
};

struct DevCtx {
	int	device = 0;
	std::mutex	mu;
	hipStream_t	upload = nullptr;
	std::vector<Block>	spare;		// blocks of destroyed databases, waiting for the next one
	size_t	spare_bytes = 0;
	// Take a block of at least `bytes`: a spare one that is not more than twice that, else a new one.
	hipError_t	take( size_t bytes, Block *out )
	{
		bytes = std::max<size_t>( ( bytes + 255 ) & ~size_t( 255 ), 256 );
		{
			std::lock_guard<std::mutex>	lk( mu );
			int	best = -1;
			for( size_t i = 0; i < spare.size(); i++ )
				if( spare[ i ].bytes >= bytes && spare[ i ].bytes <= 2 * bytes + ( 1 << 20 ) &&
					( best < 0 || spare[ i ].bytes < spare[ size_t( best ) ].bytes ) )
					best = int( i );
			if( best >= 0 ){
				*out = spare[ size_t( best ) ];
				spare_bytes -= out->bytes;
				spare.erase( spare.begin() + best );
				return hipSuccess;
			}
		}
		// (a little head room: the batches of one search differ by an entry or two)
		const size_t	want = bytes + bytes / 16;
		hipError_t	e = hipMalloc( &out->p, want );
		if( e != hipSuccess ){
			// the spare blocks are memory too: give them back and try again
			drop_spare();
			( void )hipGetLastError();
			e = hipMalloc( &out->p, want );
		}
		out->bytes = e == hipSuccess ? want : 0;
		return e;
	}
	void	give( Block b )
	{
		if( b.p == nullptr )
			return;
		{
			std::lock_guard<std::mutex>	lk( mu );
			// at most eight blocks and 4 GB wait here
			if( spare.size() < 8 && spare_bytes + b.bytes <= ( size_t( 4 ) << 30 ) ){
				spare.push_back( b );
				spare_bytes += b.bytes;
				return;
			}
		}
		( void )hipFree( b.p );
	}
	void	drop_spare()
	{
		std::lock_guard<std::mutex>	lk( mu );
		for( Block &b : spare )
			( void )hipFree( b.p );
		spare.clear();
		spare_bytes = 0;
	}
};

std::mutex	g_ctx_mu;
std::vector<std::unique_ptr<DevCtx>>	g_ctx;

// (the caller has made `device` current)
DevCtx *dev_ctx( int device, char *err, size_t errlen )
{
	std::lock_guard<std::mutex>	lk( g_ctx_mu );
	for( auto &c : g_ctx )
		if( c->device == device )
			return c.get();
	std::unique_ptr<DevCtx>	c( new DevCtx );
	c->device = device;
	hipError_t	e = hipStreamCreateWithFlags( &c->upload, hipStreamNonBlocking );
	if( e != hipSuccess ){
		snprintf( err, errlen, "hipStreamCreate (upload stream): %s", hipGetErrorString( e ) );
		return nullptr;
	}
	g_ctx.push_back( std::move( c ) );
	return g_ctx.back().get();
}

// launch-shape and diagnostic switches: read from the environment once, when the scanner is
// created (DESIGN.md has the table), changed afterwards only through rma_scanner_set_option()
struct Options {
	int	dbg = 0;
	int	pool = -1;		// -1: by the descriptor, 0: pass B tile by tile
	int	pool_min = 1024, pool_refill = 48;
	int	drain = 1;		// pooled instance: the items are walked by a kernel of their own (0: by the workgroup that found them)
	int	glist = 0;		// > 0: items of the drain kernel's list (tests: a list that overflows), 0: by the database's size
	int	drain_waves = 6;	// workgroups (of one wave) of the drain kernel per CU; 0: what LDS and registers allow (16).  Six: the kernel alone
				// is as fast as with 16 (profiles/overlap_try.py: 0.29 ms), and the next scan's search kernel starts beside it
	int	flush = -1;		// pooled instance that walks nothing (RMK_LEAN_FLUSH): -1 where the descriptor has a look-ahead chain, 0 never, 1 wherever the pooled instance runs
	int	efn_light = -1;		// the energy kernel in workgroups of one wave that stage no tables (rma_efn_light_kernel): -1 by the scan's instance, 0 never, 1 always
	int	search_wgs = 0;		// > 0: workgroups of a lean search kernel per CU (fewer than fit: another scanner's drain kernel runs beside it)
	int	host_sort = 0, timing = 0;
	int	short_force = -1;	// -1: by the mean entry length, 0 never, 1 always groups of small tiles, 2 always tiles over the concatenation
	int	tile = 0, qcap = 0;	// forced tile size / queue entries, 0: computed
	int	spill = -1;		// forced spill area, -1: SPILL_ITEMS
	int	budget = 0;
	static int	env_int( const char *name, int dflt )
	{
		const char	*v = getenv( name );
		return v != nullptr && v[ 0 ] != '\0' ? atoi( v ) : dflt;
	}
	void	latch()
	{
		dbg = env_int( "RNAMOTIF_DBG", 0 );
		pool = env_int( "RNAMOTIF_POOL", -1 );
		pool_min = std::max( 1, env_int( "RNAMOTIF_POOL_MIN", 1024 ) );
		pool_refill = env_int( "RNAMOTIF_POOL_REFILL", 48 );
		drain = env_int( "RNAMOTIF_DRAIN", 1 );
		glist = env_int( "RNAMOTIF_GLIST", 0 );
		drain_waves = env_int( "RNAMOTIF_DRAIN_WAVES", 6 );
		search_wgs = env_int( "RNAMOTIF_SEARCH_WGS", 0 );
		flush = env_int( "RNAMOTIF_FLUSH", -1 );
		efn_light = env_int( "RNAMOTIF_EFN_LIGHT", -1 );
		host_sort = env_int( "RNAMOTIF_HOSTSORT", 0 );
		timing = getenv( "RNAMOTIF_TIMING" ) != nullptr;
		if( const char *f = getenv( "RNAMOTIF_SHORT" ) )
			short_force = f[ 0 ] == '1' ? 1 : f[ 0 ] == '2' ? 2 : 0;
		tile = env_int( "RNAMOTIF_TILE", 0 );
		if( tile < 0 || tile > 16384 )
			tile = 0;
		qcap = env_int( "RNAMOTIF_QCAP", 0 );
		spill = env_int( "RNAMOTIF_SPILL", -1 );
		budget = env_int( "RNAMOTIF_BUDGET", 0 );
	}
};

// the tiling of a database for one launch shape
struct Layout {
	int	tile_t = 0, dminlen = 0, strands = 0, group = 1, qcap = 0;
	bool	flush = false;		// tiles of the size of the pooled instance that walks nothing (RMK_LEAN_FLUSH)
	// Tiles over the CONCATENATION of the entries (round 4; databases of short entries, pooled lean instance): a
	// strand of the whole packed array -- the entries one after the other, each padded to 32 bases -- is tiled as
	// if it were one long entry, so that the vectors of a tile are full whatever the entries' lengths; what a
	// tile's tests let through is brought back to its entry when it enters the pool (super_convert in the kernel).
	bool	concat = false;
	int64_t	concat_bases = 0;
	Block	blk;
	int64_t	*d_tile_start = nullptr;
	int32_t	*d_tile_seq = nullptr;
	int64_t	n_tiles = 0;
	std::vector<int64_t>	h_tile_start;	// (what the copies read: alive as long as the layout)
	std::vector<int32_t>	h_tile_seq;
	// one tile per workgroup pass: all a workgroup needs to know of tile t in one 32-byte line (RMK_META_*), so that it
	// is one load -- made a tile ahead, straight into LDS -- instead of three dependent ones at the tile's start
	int32_t	*d_tile_meta = nullptr;
	std::vector<int32_t>	h_tile_meta;
	hipEvent_t	ready = nullptr;	// the copies are complete: every scan waits for it on its stream
	~Layout(){ if( ready != nullptr ) ( void )hipEventDestroy( ready ); }
};

}	// namespace

struct rma_scanner {
	rma_program_t	prog;
	rmd_program_t	dprog;
	Options	opt;
	int	device = 0;
	DevCtx	*ctx = nullptr;
	hipStream_t	stream = nullptr;
	hipEvent_t	ev[ 5 ] = { nullptr, nullptr, nullptr, nullptr, nullptr };	// search kernel's start / end, efn kernel's, [4]: the search kernel's end when a drain kernel follows
	bool	drained = false;		// the last launch had a drain kernel
	bool	searched = false, efn_ran = false;	// a search kernel was launched at all; the last scan had an efn kernel
	rma_efn2data_t	*d_efn2 = nullptr;	// efn2() tables, global memory
	bool	need_efn2 = false;
	rmd_program_t	*d_prog = nullptr;	// compact image, prog_bytes long
	int	prog_bytes = 0;
	int	qcap = QCAP;		// work queue entries per workgroup
	int16_t	*d_t16 = nullptr;
	int32_t	*d_tlkey = nullptr, *d_loginc = nullptr;
	bool	have_efn = false;
	int32_t	*d_hits = nullptr;
	int64_t	hit_cap = 0;
	unsigned long long	*d_counters = nullptr;	// [0] count, [1] ticket
	unsigned	*d_spill = nullptr;		// [grid_blocks][spill_cap] queue overflow of every workgroup
	int	spill_cap = 0;
	bool	whole_items = false;		// ... which takes the items whole, not in pieces (see search_finish)
	int	glist_cap = 0;			// pooled instance: items of the list the drain kernel walks (the head of d_pool)
	int	glist_need = 0;			// ... and what a scan of the instance that walks nothing asked for (search_finish)
	bool	flush = false;			// long entries are searched by RMK_LEAN_FLUSH, on tiles of its own size
	int	tile_t_flush = 0, qcap_flush = 0;
	unsigned	*d_pool = nullptr;		// [grid_blocks][pool_cap][3] pooled instance: items waiting for pass B
	int	pool_cap = 0;
	int32_t	*h_raw = nullptr;		// pinned
	size_t	h_raw_cap = 0;
	std::vector<int32_t>	h_sorted;
	std::vector<rma::HitKey>	keys, keys_tmp;
	rma::DevHitSort	dsort;		// ordering on the device (rm_hitsort_dev.h)
	unsigned long long	*h_ctr = nullptr;	// pinned: the counters a launch leaves
	int	tile_t = 2048;
	int	drain_grid = 0, drain_nib = 0;
	size_t	drain_lds = 0;
	int	grid_blocks = 0;		// most workgroups of a launch of a lean instance (eight of four waves per CU)
	int	spill_blocks = 0;		// workgroups d_spill has areas for
	int	kinds = 0;			// RMD_KIND_* of the descriptor
	// the scan between rma_scan_begin() and rma_scan_end()
	struct InFlight {
		const rma_db	*db = nullptr;
		const Layout	*lay = nullptr;
		int	inst = 0, grid = 0, tile_bytes = 0;
		size_t	lds = 0;
		bool	lean = false, grouped = false;
	}	fly;
	// what the last scan left on the device, in order (rma_scan_end): for rma_gather_hits()
	const int32_t	*d_last = nullptr;
	int64_t	n_last = 0;
	int	last_state = 0;			// (rma_scanner_last_state)
	bool	last_relabelled = false;	// rma_gather_hits has put database-wide entry numbers into d_last's records
};

struct rma_db {
	int	device = 0;
	DevCtx	*ctx = nullptr;
	Block	blk;			// codes | amask | base_off | slen | pos_lo | pos_hi
	uint32_t	*d_codes = nullptr, *d_amask = nullptr;
	int64_t	*d_base_off = nullptr;
	int32_t	*d_slen = nullptr, *d_pos_lo = nullptr, *d_pos_hi = nullptr;
	std::vector<int32_t>	h_slen, h_pos_lo, h_pos_hi;	// (host copies: the tilings are made from them)
	std::vector<int64_t>	h_base_off;
	int32_t	n_seq = 0, max_slen = 0;
	int64_t	total_bases = 0, sum_slen = 0;
	int64_t	padded_bases = 0;	// bases the packed arrays hold, padding between the entries included
	bool	ascending = true;	// the entries lie in the packed arrays in their order, none overlapping
	hipEvent_t	ready = nullptr;	// the upload is complete (recorded on the upload stream)
	std::mutex	mu;			// layouts, busy
	std::vector<std::unique_ptr<Layout>>	layouts;
	std::vector<rma_scanner *>	busy;		// scanners with a scan of this database in flight
};

extern "C" void rma_db_destroy( rma_db_t *db );
extern "C" void rma_scanner_destroy( rma_scanner_t *sc );
extern "C" int rma_scan_end( rma_scanner_t *sc, const int32_t **hits, int64_t *n_hits, char *err, size_t errlen );

extern "C" int rma_device_count( void )
{
	int	n = 0;
	if( hipGetDeviceCount( &n ) != hipSuccess )
		return 0;
	return n;
}

hipError_t rmk_launch_search( int inst, int grid, size_t lds, hipStream_t s, const rmk_search_args &a )
{
	switch( inst ){
	case RMK_LEAN_POOL :	return rmk_launch_lean_pool( grid, lds, s, a );
	case RMK_LEAN_CONCAT :	return rmk_launch_lean_concat( grid, lds, s, a );
	case RMK_LEAN_FLUSH :	return rmk_launch_lean_flush( grid, lds, s, a );
	case RMK_LEAN_CONCAT_FLUSH :	return rmk_launch_lean_concat_flush( grid, lds, s, a );
	case RMK_LEAN_GROUP :	return rmk_launch_lean_group( grid, lds, s, a );
	case RMK_LEAN_TILE :	return rmk_launch_lean_tile( grid, lds, s, a );
	case RMK_GEN_PLAIN :	return rmk_launch_gen_plain( grid, lds, s, a );
	case RMK_GEN_PK :	return rmk_launch_gen_pk( grid, lds, s, a );
	case RMK_GEN_TQ :	return rmk_launch_gen_tq( grid, lds, s, a );
	case RMK_GEN_PKTQ :	return rmk_launch_gen_pktq( grid, lds, s, a );
	case RMK_GEN_WIDE :	return rmk_launch_gen_wide( grid, lds, s, a );
	case RMK_GEN_PLAIN_CONCAT :	return rmk_launch_gen_plain_concat( grid, lds, s, a );
	case RMK_GEN_PK_CONCAT :	return rmk_launch_gen_pk_concat( grid, lds, s, a );
	case RMK_GEN_TQ_CONCAT :	return rmk_launch_gen_tq_concat( grid, lds, s, a );
	case RMK_GEN_PKTQ_CONCAT :	return rmk_launch_gen_pktq_concat( grid, lds, s, a );
	}
	return hipErrorInvalidValue;
}

// LDS of one search workgroup: program image | queue | tile | 6 bit vectors | lean records
static size_t search_lds_bytes( int prog_bytes, const rmd_program_t &dp, int tile_t, bool lean, int qcap, int group = 1, bool flush = false )
{
	const int	tile_bytes = tile_t + dp.w_winsize + dp.lmargin + dp.rmargin + 80;
	// (bit vectors of a tile: the literal's, five per pair-row set, four of a leading 4-plex' strand filter, five more when a triplex follows it)
	// (lean with a look-ahead chain: one more -- the start positions that remain; the chain's other vectors
	// borrow the place of the search records, which pass A does not use)
	// (... and five when the descriptor has a best literal: where each base stands, for the literal's occurrence vector)
	const size_t	pb_bytes = ( ( lean ? 6 + ( ( dp.chain.on || dp.lit_re >= 0 ) && group == 1 ? 1 : 0 ) : 1 + 5 * size_t( dp.n_rowsets ) + ( dp.q1f.on ? ( dp.q1f.t_on ? 9 : 4 ) : 0 ) + ( dp.lit_re >= 0 ? 1 : 0 ) ) +
			( dp.lit_re >= 0 && group == 1 ? 5 : 0 ) ) *
		( size_t( tile_bytes + 63 ) / 64 + 3 ) * sizeof( unsigned long long );
	size_t	lds = size_t( prog_bytes ) + size_t( qcap ) * sizeof( unsigned ) +
		size_t( group ) * ( ( ( size_t( tile_bytes ) + 15 ) & ~size_t( 15 ) ) + pb_bytes );
	// (the instance that walks nothing has no records: only the look-ahead chain's ten working vectors, which elsewhere borrow their place)
	if( flush )
		lds += dp.chain.on ? size_t( 10 ) * ( size_t( tile_bytes + 63 ) / 64 + 3 ) * sizeof( unsigned long long ) : 0;
	else
		lds += lean ? size_t( dp.n_searches ) * SEARCH_BLOCK * LEAN_REC_BYTES : size_t( dp.n_rec_dwords ) * GENERAL_BLOCK * 4;
	if( lean && group > 1 && dp.lit_re >= 0 )	// (groups: the literal's five vectors once per wave, behind the records)
		lds += 8 + size_t( SEARCH_BLOCK / 64 ) * 6 * ( size_t( tile_bytes + 63 ) / 64 + 3 ) * sizeof( unsigned long long );	// (the sixth: the literal's start positions)
	if( !lean && dp.split_s >= 0 )		// resume states of the levels up to the split level, queue of continuations
		lds += size_t( dp.split_s + 1 ) * GENERAL_BLOCK * 8 + size_t( DEEP_QUEUE ) * ( 2 + 2 * ( dp.split_s + 1 ) ) * 4;
	return lds;
}

extern "C" int rma_scanner_create( const rma_program_t *prog, const rma_efndata_t *efn, int device,
	rma_scanner_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	rma_scanner	*sc = new rma_scanner;
	// every early return below releases the scanner and what it holds by then
	struct ScGuard { rma_scanner *p; ~ScGuard(){ if( p ) rma_scanner_destroy( p ); } }	guard{ sc };
	sc->prog = *prog;
	sc->opt.latch();
	// (host work first: a descriptor outside the device limits is refused with its reason whether
	// or not a device is there to refuse it for)
	if( rmd_build( prog, &sc->dprog, err, errlen ) )
		return 1;
	int	ndev = 0;
	if( hipGetDeviceCount( &ndev ) != hipSuccess || ndev <= 0 ){
		snprintf( err, errlen, "no HIP device available: the rnamotif scan path runs on the GPU only" );
		return 1;
	}
	if( device < 0 || device >= ndev ){
		snprintf( err, errlen, "device %d out of range (0..%d)", device, ndev - 1 );
		return 1;
	}
	if( sc->opt.budget > 0 )		// launch-shape switch: iterations per step
		sc->dprog.step_budget = std::max( 4, sc->opt.budget );
	for( int k = 0; k < prog->n_efn_sites; k++ ){
		if( prog->efn_sites[ k ].kind == RMA_EFN_KIND_EFN2 )
			sc->need_efn2 = true;	// tables come with rma_scanner_set_efn2data(), checked at the first scan
		else if( efn == nullptr ){
			snprintf( err, errlen, "the program has efn() call sites but no energy tables were given" );
			return 1;
		}
	}
	for( int k = 0; k < sc->dprog.n_searches; k++ ){
		const rmd_elem_t	&e = sc->dprog.elems[ sc->dprog.searches[ k ] ];
		if( e.type == RMA_T_H5 && !e.proper )
			sc->kinds |= RMD_KIND_PK;
		if( e.type == RMA_T_P5 || e.type == RMA_T_T1 || e.type == RMA_T_Q1 )
			sc->kinds |= RMD_KIND_TQ;
	}
	sc->device = device;
	HIPCHK( hipSetDevice( device ) );
	sc->ctx = dev_ctx( device, err, errlen );
	if( sc->ctx == nullptr )
		return 1;
	HIPCHK( hipStreamCreateWithFlags( &sc->stream, hipStreamNonBlocking ) );
	for( int i = 0; i < 5; i++ )
		HIPCHK( hipEventCreate( &sc->ev[ i ] ) );
	{
		// the device gets the compact image; sc->dprog stays the full struct for the host
		std::vector<char>	img( sizeof( rmd_program_t ) );
		sc->prog_bytes = int( rmd_make_image( &sc->dprog, img.data() ) );
		HIPCHK( hipMalloc( &sc->d_prog, size_t( sc->prog_bytes ) ) );
		HIPCHK( hipMemcpy( sc->d_prog, img.data(), size_t( sc->prog_bytes ), hipMemcpyHostToDevice ) );
	}
	HIPCHK( hipMalloc( &sc->d_counters, RMK_N_COUNTERS * sizeof( unsigned long long ) ) );
	if( efn != nullptr ){
		std::vector<int16_t>	t16;
		std::vector<int32_t>	tlkey;
		rma::efn_tables16( efn, t16, tlkey );
		HIPCHK( hipMalloc( &sc->d_t16, t16.size() * sizeof( int16_t ) ) );
		HIPCHK( hipMemcpy( sc->d_t16, t16.data(), t16.size() * sizeof( int16_t ), hipMemcpyHostToDevice ) );
		HIPCHK( hipMalloc( &sc->d_tlkey, tlkey.size() * sizeof( int32_t ) ) );
		HIPCHK( hipMemcpy( sc->d_tlkey, tlkey.data(), tlkey.size() * sizeof( int32_t ), hipMemcpyHostToDevice ) );
		HIPCHK( hipMalloc( &sc->d_loginc, RMA_EFN_LOGINC * sizeof( int32_t ) ) );
		HIPCHK( hipMemcpy( sc->d_loginc, efn->loginc, RMA_EFN_LOGINC * sizeof( int32_t ), hipMemcpyHostToDevice ) );
		sc->have_efn = true;
	}
	hipDeviceProp_t	prop;
	HIPCHK( hipGetDeviceProperties( &prop, device ) );
	sc->grid_blocks = prop.multiProcessorCount * 8;
	// (the general instances run workgroups of one wave, four times as many, on tiles a quarter the size)
	const int	per_wave0 = sc->dprog.lean_ok ? 1 : SEARCH_BLOCK / GENERAL_BLOCK;
	sc->spill_blocks = sc->grid_blocks * per_wave0;
	sc->spill_cap = sc->opt.spill >= 0 ? sc->opt.spill : SPILL_ITEMS / per_wave0;	// (tests: 0 = overflow searched in place)
	HIPCHK( hipMalloc( &sc->d_spill, std::max<size_t>( size_t( sc->spill_blocks ) * sc->spill_cap, 1 ) * sizeof( unsigned ) ) );
	if( sc->dprog.lean_ok ){
		// The search of a tile ends with a few long-running items on a few lanes, so fewer,
		// larger tiles are better as long as four workgroups still share a CU's 160 KB of LDS
		// (trna.descr, ms per 100 Mbase: T = 2048 6.97, 4096 5.87, 6144 5.40 with 8-byte records;
		// 6656 4.38, 9984 3.99 with 6-byte records; one step further only three fit: 5.0) and the
		// work queue still holds what the pre-filter lets through: on random sequence a start
		// position yields n_rank * P( first minlen pairs hold, at most lim mispairs ) items.
		const rmd_program_t	&dp = sc->dprog;
		const rmd_elem_t	&e0 = dp.elems[ dp.searches[ 0 ] ];
		double	density = 1.0;
		if( e0.type == RMA_T_H5 && e0.pairset >= 0 && e0.minlen >= 1 ){
			const uint32_t	m2 = rmd_pairsets( &dp )[ e0.pairset ].mat2;
			int	np = 0;
			for( int a = 0; a < 4; a++ )
				for( int b = 0; b < 4; b++ )
					np += ( m2 >> ( a * 5 + b ) ) & 1;
			const double	pp = np / 16.0;
			const int	lim = ( e0.ends & RMA_5PAIRED ) ? e0.mplim : std::max( e0.mplim, 1 );
			double	p = 0, comb = 1;
			for( int m = 0; m <= lim && m <= e0.minlen; m++ ){
				p += comb * std::pow( pp, e0.minlen - m ) * std::pow( 1 - pp, m );
				comb = comb * ( e0.minlen - m ) / ( m + 1 );
			}
			const int	w = dp.w_winsize;
			const int	n_rank = ( e0.maxglen != RMA_UNBOUNDED && e0.maxglen < w ? e0.maxglen : w ) - e0.minglen + 1;
			density = std::min( 1.0, p ) * std::max( 1, n_rank );
		}
		if( dp.lit_re >= 0 ){
			// ... and only where the best literal occurs at an allowed offset
			const rmd_regex_t	&lre = rmd_regexes( &dp )[ dp.lit_re ];
			double	pl = 1.0;
			for( int j = 0; j < lre.n_states; j++ ){
				int	n = 0;
				for( int c = 0; c < 4; c++ )
					n += int( ( lre.accept[ c ] >> j ) & 1 );
				pl *= n / 4.0;
			}
			density *= std::min( 1.0, pl * ( dp.lit_hi - dp.lit_lo + 1 ) );
		}
		const size_t	budget = ( 160 * 1024 ) / SEARCH_WAVES_PER_SIMD - 1152;	// (static __shared__: the waves' buffers of start positions that passed the look-ahead, 1 KB, and 40 bytes)
		// What the LDS queue cannot hold spills to HBM at 4 bytes per item, so LDS goes to the tile
		// first and the queue gets what is left, up to the expected number of items (trna.descr:
		// queue 1024 / T 9984 3.99 ms, 512 / 11008 3.94, 256 / 11520 3.91 -- the last spills a
		// third of its items for that 1 %: the queue starts at 512).  A tile should still not
		// produce more than half the spill area on average.
		const int	q_min = 512;
		sc->tile_t = 2048;
		for( int t = 16384; t >= 2048; t -= 256 )
			if( search_lds_bytes( sc->prog_bytes, dp, t, true, q_min ) <= budget &&
				density * t * 1.1 <= q_min + std::max( sc->spill_cap, 2 * 512 ) / 2 ){
				sc->tile_t = t;
				break;
			}
		sc->qcap = q_min;
		const int	q_want = int( std::min( 8192.0, std::ceil( density * sc->tile_t * 1.2 / 256 ) * 256 ) );
		while( sc->qcap + 256 <= q_want && search_lds_bytes( sc->prog_bytes, dp, sc->tile_t, true, sc->qcap + 256 ) <= budget )
			sc->qcap += 256;
		// The pooled instance that walks nothing (RMK_LEAN_FLUSH: every survivor of pass A' goes to the drain kernel's list):
		// FLUSH_WAVES_PER_SIMD workgroups a CU, tiles as large as its smaller share of LDS holds without the search records.
		// For descriptors with a look-ahead chain -- a few dozen long walks per workgroup, which the drain kernel takes anyway;
		// hundreds of cheap items (ire.descr, mp.ends.descr) are walked best where they are found.
		// (sc->flush: it can run; use_flush(): the options of the moment want it)
		sc->flush = ( dp.w_winsize + dp.lmargin + dp.rmargin + 14 ) / 8 <= 32;
		if( sc->flush ){
			const size_t	budget_f = ( 160 * 1024 ) / FLUSH_WAVES_PER_SIMD - 1152;
			sc->tile_t_flush = 0;
			// (no larger than one pass of the workgroup's 256 lanes decodes, 32 bases a lane: profiles/flush_matrix.sh -- tiles of
			// 7936 positions 0.639 ms, of 8192, a second pass for ten lanes, 0.757; of 10752 0.658)
			const int	t_one = ( 254 * 32 - 61 - ( dp.w_winsize + dp.lmargin + dp.rmargin ) ) / 256 * 256;
			for( int t = std::max( 2048, std::min( 16384, t_one ) ); t >= 2048; t -= 256 )
				if( search_lds_bytes( sc->prog_bytes, dp, t, true, q_min, 1, true ) <= budget_f &&
					density * t * 1.1 <= q_min + std::max( sc->spill_cap, 2 * 512 ) / 2 ){
					sc->tile_t_flush = t;
					break;
				}
			sc->qcap_flush = q_min;
			const int	q_want_f = int( std::min( 8192.0, std::ceil( density * sc->tile_t_flush * 1.2 / 256 ) * 256 ) );
			while( sc->tile_t_flush > 0 && sc->qcap_flush + 256 <= q_want_f &&
				search_lds_bytes( sc->prog_bytes, dp, sc->tile_t_flush, true, sc->qcap_flush + 256, 1, true ) <= budget_f )
				sc->qcap_flush += 256;
			if( sc->tile_t_flush == 0 )
				sc->flush = false;
		}
	}
	if( !sc->dprog.lean_ok ){
		// general instance: workgroups of one wave (GENERAL_BLOCK), as many per SIMD as the registers allow
		// (GENERAL_WAVES) and as still leave every one of them a tile of a thousand positions or more next
		// to its records -- 12 bytes per level and lane -- and its queue (what the queue cannot hold spills
		// to HBM)
		const int	per_wave = SEARCH_BLOCK / GENERAL_BLOCK;	// workgroups where one of four waves stood
		sc->qcap = 512 / per_wave < 128 ? 128 : 512 / per_wave;
		sc->tile_t = 1024;
		bool	found = false;
		// (descriptors with triplexes / 4-plexes: two workgroups per SIMD on tiles of three thousand positions
		// rather than three on a thousand -- a tile's second round then has a few dozen continuations for
		// its lanes instead of five: qu+tr 19.6 -> 16.7 ms; pk1 and pk_j1+2 are best at 2048, four per SIMD)
		const int	wg0 = ( sc->kinds & RMD_KIND_TQ ) ? 2 : GENERAL_WAVES( 0 );
		for( int wg = wg0; wg >= 1 && !found; wg-- ){
			const size_t	budget = ( 160 * 1024 ) / ( wg * per_wave ) - ( per_wave > 1 ? 1024 : 2560 );	// (static __shared__ -- the pre-filter's wave buffers -- and allocation granules)
			for( int t = ( per_wave > 1 ? 4096 : 8192 ); t >= ( wg > 1 ? 3072 / per_wave : 1024 / per_wave ); t -= 256 )
				if( search_lds_bytes( sc->prog_bytes, sc->dprog, t, false, sc->qcap ) <= budget ){
					sc->tile_t = t;
					found = true;
					break;
				}
		}
	}
	if( sc->opt.qcap >= 64 && sc->opt.qcap <= 16384 )
		sc->qcap = sc->qcap_flush = ( sc->opt.qcap + 3 ) & ~3;	// (what follows the queue in LDS is read 8 bytes at a time)
	if( sc->opt.tile > 0 )
		sc->tile_t = sc->opt.tile;
	// room for the candidates of a few hundred Mbase at the densities of the reference's descriptors
	// (63 per Mbase for trna.descr); a scan that finds more is repeated into a buffer of the right
	// size (count-then-emit, rma_scan_end)
	sc->hit_cap = 1 << 17;
	HIPCHK( hipMalloc( &sc->d_hits, size_t( sc->hit_cap ) * sc->dprog.hit_stride * sizeof( int32_t ) ) );
	HIPCHK( hipHostMalloc( reinterpret_cast<void **>( &sc->h_ctr ), ( RMK_GCTL + 3 ) * sizeof( unsigned long long ), hipHostMallocDefault ) );
	// page-locked room for the records of a usual batch (16 K of them) now, not in the first scan
	sc->h_raw_cap = size_t( 16384 ) * sc->dprog.hit_stride;
	HIPCHK( hipHostMalloc( reinterpret_cast<void **>( &sc->h_raw ), sc->h_raw_cap * sizeof( int32_t ), hipHostMallocDefault ) );
	if( sc->dsort.reserve( sc->hit_cap, sc->dprog.hit_stride ) != hipSuccess )
		( void )hipGetLastError();	// (no room for the ordering's buffers: rma_scan_end orders on the host)
	guard.p = nullptr;
	*out = sc;
	return 0;
}

extern "C" int rma_scanner_set_option( rma_scanner_t *sc, const char *name, int value, char *err, size_t errlen )
{
	const std::string	n = name ? name : "";
	Options	&o = sc->opt;
	if( n == "dbg" ) o.dbg = value;
	else if( n == "pool" ) o.pool = value;
	else if( n == "pool_min" ) o.pool_min = std::max( 1, value );
	else if( n == "pool_refill" ) o.pool_refill = value;
	else if( n == "drain" ) o.drain = value;
	else if( n == "glist" ) o.glist = std::max( 0, value );
	else if( n == "drain_waves" ) o.drain_waves = std::max( 0, value );
	else if( n == "search_wgs" ) o.search_wgs = std::max( 0, value );
	else if( n == "flush" ) o.flush = value;
	else if( n == "efn_light" ) o.efn_light = value;
	else if( n == "host_sort" ) o.host_sort = value;
	else if( n == "timing" ) o.timing = value;
	else if( n == "short" ) o.short_force = value;
	else if( n == "forget_last" ){
		// (not a switch: the last scan's records are no longer there for rma_gather_hits() -- a rank whose share of a
		// round is empty sends nothing, whatever the round before left)
		sc->d_last = nullptr;
		sc->n_last = 0;
		sc->last_state = 0;
		sc->last_relabelled = false;
	}else{
		snprintf( err, errlen, "rma_scanner_set_option: no option '%s' that can change after creation "
			"(dbg, pool, pool_min, pool_refill, drain, glist, drain_waves, search_wgs, flush, efn_light, host_sort, timing, short; forget_last)", n.c_str() );
		return 1;
	}
	return 0;
}

extern "C" int rma_scanner_set_efn2data( rma_scanner_t *sc, const rma_efn2data_t *efn2, char *err, size_t errlen )
{
	HIPCHK( hipSetDevice( sc->device ) );
	if( sc->d_efn2 == nullptr )
		HIPCHK( hipMalloc( &sc->d_efn2, sizeof( rma_efn2data_t ) ) );
	HIPCHK( hipMemcpy( sc->d_efn2, efn2, sizeof( rma_efn2data_t ), hipMemcpyHostToDevice ) );
	return 0;
}

extern "C" void rma_scanner_destroy( rma_scanner_t *sc )
{
	if( sc == nullptr )
		return;
	if( sc->d_prog == nullptr && sc->stream == nullptr ){	// (refused before anything was set up on a device)
		delete sc;
		return;
	}
	( void )hipSetDevice( sc->device );
	if( sc->fly.db != nullptr ){		// (a scan begun and never ended)
		const int32_t	*h = nullptr;
		int64_t	n = 0;
		char	e[ 256 ];
		( void )rma_scan_end( sc, &h, &n, e, sizeof( e ) );
	}
	if( sc->stream )
		( void )hipStreamSynchronize( sc->stream );
	( void )hipFree( sc->d_prog );
	( void )hipFree( sc->d_efn2 );
	if( sc->h_raw != nullptr )
		( void )hipHostFree( sc->h_raw );
	if( sc->h_ctr != nullptr )
		( void )hipHostFree( sc->h_ctr );
	sc->dsort.release();
	( void )hipFree( sc->d_t16 );
	( void )hipFree( sc->d_tlkey );
	( void )hipFree( sc->d_loginc );
	( void )hipFree( sc->d_hits );
	( void )hipFree( sc->d_counters );
	( void )hipFree( sc->d_spill );
	( void )hipFree( sc->d_pool );
	for( int i = 0; i < 5; i++ )
		if( sc->ev[ i ] )
			( void )hipEventDestroy( sc->ev[ i ] );
	if( sc->stream )
		( void )hipStreamDestroy( sc->stream );
	delete sc;
}

// ---------------------------------------------------------------- databases
static size_t align256( size_t x ) { return ( x + 255 ) & ~size_t( 255 ); }

// Upload n packed entries: `pieces` are runs of whole words of the source arrays that follow each
// other in the device arrays (one run for a slice of a pack, one per entry for chosen entries);
// base_off[] are the entries' offsets in bases (multiples of 32) in the device arrays.  wait:
// return when the copies are complete (the source may then go away); otherwise the caller keeps
// the source as it is until rma_db_wait() or the end of the first scan.
struct Piece { const uint32_t *codes, *amask; size_t mask_words; };

static int db_upload( int device, const std::vector<Piece> &pieces, const int64_t *base_off, const int32_t *slen, int32_t n,
	const int32_t *pos_lo, const int32_t *pos_hi, bool wait, rma_db_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	if( pos_lo != nullptr )
		for( int i = 0; i < n; i++ )
			if( pos_lo[ i ] < 0 || pos_hi[ i ] < pos_lo[ i ] ){
				snprintf( err, errlen, "entry %d: start positions [%d, %d) are not a range", i, pos_lo[ i ], pos_hi[ i ] );
				return 1;
			}
	HIPCHK( hipSetDevice( device ) );
	DevCtx	*ctx = dev_ctx( device, err, errlen );
	if( ctx == nullptr )
		return 1;
	rma_db	*db = new rma_db;
	db->device = device;
	db->ctx = ctx;
	db->n_seq = n;
	// every early return below frees what has been allocated so far
	struct DbGuard { rma_db *p; ~DbGuard(){ if( p ) rma_db_destroy( p ); } }	guard{ db };
	db->h_slen.assign( slen, slen + n );
	db->h_base_off.assign( base_off, base_off + n );
	if( pos_lo != nullptr ){
		db->h_pos_lo.assign( pos_lo, pos_lo + n );
		db->h_pos_hi.assign( pos_hi, pos_hi + n );
	}
	for( int i = 0; i < n; i++ ){
		db->sum_slen += slen[ i ];
		db->max_slen = std::max( db->max_slen, slen[ i ] );
		if( pos_lo != nullptr )		// this database answers for a slice of the entry's start positions only
			db->total_bases += std::max<int64_t>( 0, std::min<int64_t>( pos_hi[ i ], slen[ i ] ) - pos_lo[ i ] );
		else
			db->total_bases += slen[ i ];
	}
	size_t	n_mask = 0;
	for( const Piece &p : pieces )
		n_mask += p.mask_words;
	db->padded_bases = int64_t( n_mask ) * 32;
	for( int i = 0; i < n; i++ )
		if( base_off[ i ] < 0 || base_off[ i ] + slen[ i ] > db->padded_bases || ( i + 1 < n && base_off[ i ] + slen[ i ] > base_off[ i + 1 ] ) )
			db->ascending = false;
	const size_t	nn = size_t( std::max( n, 1 ) );
	const size_t	o_codes = 0, o_amask = align256( std::max<size_t>( 2 * n_mask, 1 ) * 4 );
	const size_t	o_off = o_amask + align256( std::max<size_t>( n_mask, 1 ) * 4 ), o_slen = o_off + align256( nn * 8 );
	const size_t	o_lo = o_slen + align256( nn * 4 ), o_hi = o_lo + ( pos_lo ? align256( nn * 4 ) : 0 );
	const size_t	total = o_hi + ( pos_lo ? align256( nn * 4 ) : 0 );
	HIPCHK( ctx->take( total, &db->blk ) );
	char	*base = static_cast<char *>( db->blk.p );
	db->d_codes = reinterpret_cast<uint32_t *>( base + o_codes );
	db->d_amask = reinterpret_cast<uint32_t *>( base + o_amask );
	db->d_base_off = reinterpret_cast<int64_t *>( base + o_off );
	db->d_slen = reinterpret_cast<int32_t *>( base + o_slen );
	if( pos_lo != nullptr ){
		db->d_pos_lo = reinterpret_cast<int32_t *>( base + o_lo );
		db->d_pos_hi = reinterpret_cast<int32_t *>( base + o_hi );
	}
	HIPCHK( hipEventCreateWithFlags( &db->ready, hipEventDisableTiming ) );
	hipStream_t	up = ctx->upload;
	size_t	at = 0;		// mask words uploaded so far
	for( const Piece &p : pieces ){
		if( p.mask_words == 0 )
			continue;
		HIPCHK( hipMemcpyAsync( db->d_codes + 2 * at, p.codes, 2 * p.mask_words * 4, hipMemcpyHostToDevice, up ) );
		HIPCHK( hipMemcpyAsync( db->d_amask + at, p.amask, p.mask_words * 4, hipMemcpyHostToDevice, up ) );
		at += p.mask_words;
	}
	if( n > 0 ){
		// (the small tables come from the host copies the database keeps: they outlive the call)
		HIPCHK( hipMemcpyAsync( db->d_base_off, db->h_base_off.data(), size_t( n ) * 8, hipMemcpyHostToDevice, up ) );
		HIPCHK( hipMemcpyAsync( db->d_slen, db->h_slen.data(), size_t( n ) * 4, hipMemcpyHostToDevice, up ) );
		if( pos_lo != nullptr ){
			HIPCHK( hipMemcpyAsync( db->d_pos_lo, db->h_pos_lo.data(), size_t( n ) * 4, hipMemcpyHostToDevice, up ) );
			HIPCHK( hipMemcpyAsync( db->d_pos_hi, db->h_pos_hi.data(), size_t( n ) * 4, hipMemcpyHostToDevice, up ) );
		}
	}
	HIPCHK( hipEventRecord( db->ready, up ) );
	if( wait )
		HIPCHK( hipEventSynchronize( db->ready ) );
	guard.p = nullptr;
	*out = db;
	return 0;
}

// the device a database made for scanner sc lives on (sc may be null: device 0)
static int device_of( const rma_scanner_t *sc ) { return sc != nullptr ? sc->device : 0; }

static const Layout *layout_for( rma_scanner_t *sc, const rma_db_t *cdb, char *err, size_t errlen, hipStream_t on = nullptr );

// a database made for a scanner gets that scanner's tiling with it, behind its words on the upload stream
static int tile_at_creation( rma_scanner_t *sc, rma_db_t **out, char *err, size_t errlen )
{
	if( sc == nullptr || layout_for( sc, *out, err, errlen, ( *out )->ctx->upload ) != nullptr )
		return 0;
	rma_db_destroy( *out );
	*out = nullptr;
	return 1;
}

static int db_from_text( rma_scanner_t *sc, const char *const *seqs, const int32_t *slens, const int32_t *pos_lo, const int32_t *pos_hi,
	int32_t n, rma_db_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	rma::PackedDb	pk;
	for( int i = 0; i < n; i++ )
		pk.add( seqs[ i ], slens[ i ] < 0 ? 0 : slens[ i ] );
	std::vector<Piece>	pieces{ Piece{ pk.codes.data(), pk.amask.data(), pk.amask.size() } };
	if( db_upload( device_of( sc ), pieces, pk.base_off.data(), pk.slen.data(), n, pos_lo, pos_hi, true, out, err, errlen ) )
		return 1;
	return tile_at_creation( sc, out, err, errlen );
}

extern "C" int rma_db_create_ranges( rma_scanner_t *sc, const char *const *seqs, const int32_t *slens,
	const int32_t *pos_lo, const int32_t *pos_hi, int32_t n, rma_db_t **out, char *err, size_t errlen )
{
	return db_from_text( sc, seqs, slens, pos_lo, pos_hi, n, out, err, errlen );
}

extern "C" int rma_db_create( rma_scanner_t *sc, const char *const *seqs, const int32_t *slens, int32_t n,
	rma_db_t **out, char *err, size_t errlen )
{
	return db_from_text( sc, seqs, slens, nullptr, nullptr, n, out, err, errlen );
}

const rma::PackFile *rma_pack_file( const rma_pack_t *pk );	// rm_capi.cpp

static int db_from_pack( rma_scanner_t *sc, const rma_pack_t *pack, int32_t first, int32_t count, bool wait,
	rma_db_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	const rma::PackFile	&pf = *rma_pack_file( pack );
	if( first < 0 || count < 0 || first + count > pf.count() ){
		snprintf( err, errlen, "entries [%d, %d) are outside the packed database (%d entries)", first, first + count, pf.count() );
		return 1;
	}
	std::vector<Piece>	pieces;
	std::vector<int64_t>	rel( static_cast<size_t>( count ), 0 );
	if( count > 0 ){
		const int64_t	b0 = pf.base_off[ first ];
		const int	last = first + count - 1;
		const int64_t	b1 = pf.base_off[ last ] + ( ( int64_t( pf.slen[ last ] ) + 31 ) / 32 ) * 32;
		for( int i = 0; i < count; i++ )
			rel[ i ] = pf.base_off[ first + i ] - b0;
		pieces.push_back( Piece{ pf.codes.data() + b0 / 16, pf.amask.data() + b0 / 32, size_t( ( b1 - b0 ) / 32 ) } );
	}
	if( db_upload( device_of( sc ), pieces, rel.data(), pf.slen.data() + first, count, nullptr, nullptr, wait, out, err, errlen ) )
		return 1;
	return tile_at_creation( sc, out, err, errlen );
}

extern "C" int rma_db_create_packed( rma_scanner_t *sc, const rma_pack_t *pack, int32_t first, int32_t count,
	rma_db_t **out, char *err, size_t errlen )
{
	return db_from_pack( sc, pack, first, count, true, out, err, errlen );
}

extern "C" int rma_db_create_packed_async( rma_scanner_t *sc, const rma_pack_t *pack, int32_t first, int32_t count,
	rma_db_t **out, char *err, size_t errlen )
{
	return db_from_pack( sc, pack, first, count, false, out, err, errlen );
}

extern "C" int rma_db_create_packed_ranges( rma_scanner_t *sc, const rma_pack_t *pack, const int32_t *entry,
	const int32_t *pos_lo, const int32_t *pos_hi, int32_t n, rma_db_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	const rma::PackFile	&pf = *rma_pack_file( pack );
	// the chosen entries side by side, each copied from where it lies in the pack (every entry starts
	// on a 32-base boundary: whole words); entries that follow each other in the pack go as one run
	std::vector<Piece>	pieces;
	std::vector<int64_t>	rel( size_t( std::max( n, 0 ) ) );
	std::vector<int32_t>	slen( size_t( std::max( n, 0 ) ) );
	int64_t	at = 0;		// mask words so far
	for( int i = 0; i < n; i++ ){
		const int	e = entry[ i ];
		if( e < 0 || e >= pf.count() ){
			snprintf( err, errlen, "entry %d is outside the packed database (%d entries)", e, pf.count() );
			return 1;
		}
		const int64_t	w1 = pf.base_off[ e ] / 32, nw1 = ( int64_t( pf.slen[ e ] ) + 31 ) / 32;
		rel[ i ] = at * 32;
		slen[ i ] = pf.slen[ e ];
		if( !pieces.empty() && pieces.back().amask + pieces.back().mask_words == pf.amask.data() + w1 )
			pieces.back().mask_words += size_t( nw1 );
		else if( nw1 > 0 )
			pieces.push_back( Piece{ pf.codes.data() + 2 * w1, pf.amask.data() + w1, size_t( nw1 ) } );
		at += nw1;
	}
	if( db_upload( device_of( sc ), pieces, rel.data(), slen.data(), n, pos_lo, pos_hi, true, out, err, errlen ) )
		return 1;
	return tile_at_creation( sc, out, err, errlen );
}

extern "C" int rma_db_wait( rma_db_t *db, char *err, size_t errlen )
{
	HIPCHK( hipSetDevice( db->device ) );
	HIPCHK( hipEventSynchronize( db->ready ) );
	return 0;
}

extern "C" void rma_db_destroy( rma_db_t *db )
{
	if( db == nullptr )
		return;
	( void )hipSetDevice( db->device );
	// scans of this database that were begun and not ended: their kernels read it
	std::vector<rma_scanner *>	busy;
	{
		std::lock_guard<std::mutex>	lk( db->mu );
		busy = db->busy;
	}
	for( rma_scanner *sc : busy )
		( void )hipStreamSynchronize( sc->stream );
	if( db->ready != nullptr ){
		( void )hipEventSynchronize( db->ready );	// (an upload still on its way into the block)
		( void )hipEventDestroy( db->ready );
	}
	// (a tiling's two copies may still be on their way -- made by rma_db_attach() on a scanner's stream, or behind the
	// words on the upload stream after db->ready was recorded: they read the layout's host vectors and write its block)
	for( auto &l : db->layouts )
		if( l->ready != nullptr )
			( void )hipEventSynchronize( l->ready );
	if( db->ctx != nullptr ){
		db->ctx->give( db->blk );
		for( auto &l : db->layouts )
			db->ctx->give( l->blk );
	}
	delete db;
}

extern "C" int64_t rma_db_bases( const rma_db_t *db ) { return db->total_bases; }

// ---------------------------------------------------------------- pinned host memory
// A packed database in memory whose words are page-locked uploads by DMA, without a staging copy,
// and asynchronously (rma_db_create_packed_async).  Locking costs about as much as one upload, so it
// pays for a pack that is uploaded more than once or under a running scan.
extern "C" int rma_pack_pin( rma_pack_t *pack, char *err, size_t errlen )
{
	rma::PackFile	&pf = *const_cast<rma::PackFile *>( rma_pack_file( pack ) );
	if( pf.pin.unreg != nullptr )
		return 0;
	void	*c = pf.codes.empty() ? nullptr : pf.codes.data(), *m = pf.amask.empty() ? nullptr : pf.amask.data();
	if( c != nullptr )
		HIPCHK( hipHostRegister( c, pf.codes.size() * 4, hipHostRegisterDefault ) );
	if( m != nullptr ){
		hipError_t	e = hipHostRegister( m, pf.amask.size() * 4, hipHostRegisterDefault );
		if( e != hipSuccess ){
			if( c != nullptr )
				( void )hipHostUnregister( c );
			snprintf( err, errlen, "hipHostRegister: %s", hipGetErrorString( e ) );
			return 1;
		}
	}
	pf.pin.c = c;
	pf.pin.m = m;
	pf.pin.unreg = []( void *p ){ ( void )hipHostUnregister( p ); };
	return 0;
}

// the pooled lean instance (see the kernel): when the window of an item, four bits a base, fits the
// column a lane gets of the tile's place in LDS
// the pooled instance that walks nothing: where it can run (rma_scanner_create) and the options do not ask for the other one
static bool use_flush( const rma_scanner_t *sc )
{
	const Options	&o = sc->opt;
	return sc->flush && o.drain != 0 && o.pool != 0 && !( o.dbg & ( 16 | 2048 | 8388608 ) ) &&
		( o.flush < 0 ? sc->dprog.chain.on != 0 : o.flush != 0 );
}

static bool pooled_fits( const rma_scanner_t *sc, int tile_t, bool flush = false )
{
	const rmd_program_t	&dp = sc->dprog;
	if( !dp.lean_ok || ( sc->opt.dbg & 16 ) )
		return false;
	if( flush )		// (the instance that walks nothing lays out no window: rma_scanner_create asked what the drain kernel asks)
		return use_flush( sc );
	const int	tile_bytes = tile_t + dp.w_winsize + dp.lmargin + dp.rmargin + 80;
	const int	n_dw = ( dp.w_winsize + dp.lmargin + dp.rmargin + 14 ) / 8;
	const size_t	room = size_t( ( tile_bytes + 15 ) & ~15 ) + size_t( 6 + ( dp.chain.on ? 1 : 0 ) ) * ( ( tile_bytes + 63 ) / 64 + 3 ) * sizeof( unsigned long long );
	bool	pooled = n_dw <= 32 && size_t( n_dw ) * SEARCH_BLOCK * sizeof( uint32_t ) <= room;
	if( sc->opt.pool >= 0 )		// 0: pass B tile by tile (tests, profiles/pool_matrix.py)
		pooled = pooled && sc->opt.pool != 0;
	return pooled;
}

// ---------------------------------------------------------------- tilings
// The launch shape of database db under scanner sc, and its tiling.  Long entries: the scanner's
// tile, one per workgroup pass.  A database of many short entries (GenBank divisions, transcript
// sets) never fills such a tile, and a few dozen queue items cannot occupy 256 lanes: it gets small
// tiles in groups of SHORT_GROUP per workgroup pass (rma_search_kernel<.., G>), if the descriptor is
// lean and the group fits the LDS budget.
static const Layout *layout_for( rma_scanner_t *sc, const rma_db_t *cdb, char *err, size_t errlen, hipStream_t on )
{
	if( on == nullptr )
		on = sc->stream;
	rma_db	*db = const_cast<rma_db *>( cdb );
	int	tile_t = sc->tile_t, qcap = sc->qcap, group = 1;
	const int	n = db->n_seq;
	// (cloverleaf-like descriptors -- a look-ahead chain whose first helix is tested jointly with the stem-loop behind
	// it -- do better tile by tile even there: chain, pass A' and the drain kernel are the one-tile instance's;
	// trna.descr over the reference's test database x 20: 3.36 against 3.70 ms.  bulge.descr, a chain of one
	// stem-loop, stays with the groups: 1.57 against 2.30.)
	bool	grouped = n >= 64 && db->sum_slen / n < SHORT_ENTRY_MEAN && sc->opt.tile == 0 && !( sc->dprog.chain.on && sc->dprog.chain.hn_on );
	if( sc->opt.short_force >= 0 )		// 0 never, 1 always (tests)
		grouped = sc->opt.short_force == 1;
	// Short entries and a descriptor the pooled instance takes: tiles over the concatenation of the entries (round 4)
	// -- whole start positions only (no slices), entries in order in the packed arrays, positions within 30 bits.
	const bool	short_db = n >= 64 && db->sum_slen / n < SHORT_ENTRY_MEAN && sc->opt.tile == 0;
	bool	concat = ( sc->opt.short_force < 0 ? short_db : sc->opt.short_force == 2 ) && n >= 1 &&
		db->h_pos_lo.empty() && db->ascending && db->padded_bases < ( int64_t( 1 ) << 30 ) &&
		( pooled_fits( sc, tile_t ) || !sc->dprog.lean_ok || ( sc->opt.dbg & 16 ) ) && !sc->dprog.wide;		// (the pooled lean instance, or a general one)
	if( concat )
		grouped = false;
	// (long entries, a descriptor with a look-ahead chain: the instance that walks nothing, on tiles of its own size)
	const bool	flush = use_flush( sc ) && !grouped;
	if( flush ){
		tile_t = sc->opt.tile > 0 ? sc->opt.tile : sc->tile_t_flush;
		qcap = sc->qcap_flush;
	}
	if( grouped && sc->dprog.lean_ok && !( sc->opt.dbg & 16 ) ){
		const size_t	budget = ( 160 * 1024 ) / SEARCH_WAVES_PER_SIMD - 64 - SHORT_GROUP * 32;
		// (tiles of 1024 positions measured slower than of 768 where both fit: mp.ends 1.56 / 1.40 ms)
		for( int t = 768; t >= 256; t -= 256 ){
			int	q = 256;	// LDS goes to the slots; what a group queues beyond this spills
			if( sc->opt.qcap > 0 )	// tests: force the overflow path
				q = std::max( 64, sc->opt.qcap );
			if( search_lds_bytes( sc->prog_bytes, sc->dprog, t, true, q, SHORT_GROUP ) <= budget ){
				tile_t = t;
				qcap = q;
				group = SHORT_GROUP;
				break;
			}
		}
	}
	const int	strands = sc->prog.chk_both_strs ? 2 : 1, dminlen = sc->prog.dminlen;
	// A general instance's tile is one wave's, walks and all, for milliseconds: a database of fewer tiles than the device
	// holds workgroups (eight a CU) gets smaller ones -- down to 256 positions -- so that a short database, or one heavy
	// region of it, is not the work of a handful of waves (a 6 000 base database of repeats: 3 tiles, 90 s).
	if( !sc->dprog.lean_ok && group == 1 && sc->opt.tile == 0 ){
		const int64_t	positions = ( concat ? db->padded_bases : db->sum_slen ) * strands, slots = sc->grid_blocks;
		if( positions / tile_t < slots )
			tile_t = int( std::min<int64_t>( tile_t, std::max<int64_t>( 256, ( positions / slots + 63 ) / 64 * 64 ) ) );
	}
	std::lock_guard<std::mutex>	lk( db->mu );
	for( auto &l : db->layouts )
		if( l->tile_t == tile_t && l->dminlen == dminlen && l->strands == strands && l->group == group && l->qcap == qcap && l->concat == concat && l->flush == flush )
			return l.get();
	std::unique_ptr<Layout>	l( new Layout );
	l->tile_t = tile_t;
	l->dminlen = dminlen;
	l->strands = strands;
	l->group = group;
	l->qcap = qcap;
	l->concat = concat;
	l->flush = flush;
	l->concat_bases = db->padded_bases;
	std::vector<int64_t>	&tile_start = l->h_tile_start;
	tile_start.assign( size_t( n ) + 1, 0 );
	if( concat ){
		// one "entry" of padded_bases bases per strand; a tile's line names the entries its start positions fall into
		// (RMK_META_SEQ: the first, RMK_META_PAD: how many -- the search of an item's entry stays within them)
		const int64_t	total = db->padded_bases, nsz = total - dminlen + 1;
		const int64_t	nt = nsz > 0 ? ( nsz + tile_t - 1 ) / tile_t : 0;
		l->n_tiles = nt * strands;
		tile_start[ n ] = l->n_tiles;		// (nothing reads the per-entry sums of such a tiling)
		l->h_tile_seq.assign( 1, 0 );
		std::vector<int32_t>	&meta = l->h_tile_meta;
		meta.assign( size_t( std::max<int64_t>( l->n_tiles, 1 ) ) * RMK_META_WORDS, 0 );
		const std::vector<int64_t>	&bo = db->h_base_off;
		auto entry_at = [ & ]( int64_t g ) -> int {	// the last entry that begins at or before base g of the arrays
			const int	k = int( std::upper_bound( bo.begin(), bo.end(), g ) - bo.begin() ) - 1;
			return k < 0 ? 0 : k;
		};
		for( int64_t t = 0; t < l->n_tiles; t++ ){
			const int	comp = int( t / nt );
			const int64_t	z0 = ( t % nt ) * tile_t, z1 = std::min<int64_t>( z0 + tile_t, nsz ) - 1;
			const int64_t	g_lo = comp ? total - 1 - z1 : z0, g_hi = comp ? total - 1 - z0 : z1;
			const int	k_lo = entry_at( g_lo ), k_hi = entry_at( g_hi );
			int32_t	*m = &meta[ size_t( t ) * RMK_META_WORDS ];
			m[ RMK_META_SEQ ] = k_lo;
			m[ RMK_META_COMP ] = comp;
			m[ RMK_META_Z0 ] = int32_t( z0 );
			m[ RMK_META_SLEN ] = int32_t( total );
			m[ RMK_META_OFF_LO ] = m[ RMK_META_OFF_HI ] = 0;
			m[ RMK_META_POS_HI ] = 0x7fffffff;
			m[ RMK_META_PAD ] = k_hi - k_lo + 1;
		}
	}else
	for( int i = 0; i < n; i++ ){
		int64_t	nsz = int64_t( db->h_slen[ i ] ) - dminlen + 1;	// start positions of a strand
		if( !db->h_pos_lo.empty() )
			nsz = std::min<int64_t>( nsz, db->h_pos_hi[ i ] ) - db->h_pos_lo[ i ];
		const int64_t	nt = nsz > 0 ? ( nsz + tile_t - 1 ) / tile_t : 0;
		tile_start[ i + 1 ] = tile_start[ i ] + nt * strands;
	}
	l->n_tiles = tile_start[ n ];
	std::vector<int32_t>	&tile_seq = l->h_tile_seq;
	if( !concat ){
		tile_seq.resize( size_t( std::max<int64_t>( l->n_tiles, 1 ) ) );
		for( int i = 0; i < n; i++ )
			for( int64_t t = tile_start[ i ]; t < tile_start[ i + 1 ]; t++ )
				tile_seq[ size_t( t ) ] = i;
	}
	std::vector<int32_t>	&tile_meta = l->h_tile_meta;
	if( group == 1 && !concat ){
		tile_meta.assign( size_t( std::max<int64_t>( l->n_tiles, 1 ) ) * RMK_META_WORDS, 0 );
		for( int i = 0; i < n; i++ ){
			const int64_t	per_strand = ( tile_start[ i + 1 ] - tile_start[ i ] ) / strands;
			const int	lo = db->h_pos_lo.empty() ? 0 : db->h_pos_lo[ i ], hi = db->h_pos_hi.empty() ? 0x7fffffff : db->h_pos_hi[ i ];
			for( int64_t t = tile_start[ i ]; t < tile_start[ i + 1 ]; t++ ){
				int32_t	*m = &tile_meta[ size_t( t ) * RMK_META_WORDS ];
				const int64_t	local = t - tile_start[ i ];
				m[ RMK_META_SEQ ] = i;
				m[ RMK_META_COMP ] = int32_t( local / per_strand );
				m[ RMK_META_Z0 ] = lo + int32_t( local % per_strand ) * tile_t;
				m[ RMK_META_SLEN ] = db->h_slen[ i ];
				m[ RMK_META_OFF_LO ] = int32_t( uint64_t( db->h_base_off[ i ] ) & 0xffffffffu );
				m[ RMK_META_OFF_HI ] = int32_t( uint64_t( db->h_base_off[ i ] ) >> 32 );
				m[ RMK_META_POS_HI ] = hi;
			}
		}
	}
	const size_t	o_seq = align256( tile_start.size() * 8 ), o_meta = align256( o_seq + tile_seq.size() * 4 );
	hipError_t	e = db->ctx->take( o_meta + tile_meta.size() * 4, &l->blk );
	// On the stream given: the upload stream when the database is being made for a scanner (its first
	// scan then finds the tiling there), else the scanner's.  The host copies stay with the layout and
	// an event says when the copies are done, so nothing waits here.
	if( e == hipSuccess ){
		l->d_tile_start = static_cast<int64_t *>( l->blk.p );
		l->d_tile_seq = reinterpret_cast<int32_t *>( static_cast<char *>( l->blk.p ) + o_seq );
		e = hipMemcpyAsync( l->d_tile_start, tile_start.data(), tile_start.size() * 8, hipMemcpyHostToDevice, on );
	}
	if( e == hipSuccess )
		e = hipMemcpyAsync( l->d_tile_seq, tile_seq.data(), tile_seq.size() * 4, hipMemcpyHostToDevice, on );
	if( e == hipSuccess && !tile_meta.empty() ){
		l->d_tile_meta = reinterpret_cast<int32_t *>( static_cast<char *>( l->blk.p ) + o_meta );
		e = hipMemcpyAsync( l->d_tile_meta, tile_meta.data(), tile_meta.size() * 4, hipMemcpyHostToDevice, on );
	}
	if( e == hipSuccess )
		e = hipEventCreateWithFlags( &l->ready, hipEventDisableTiming );
	if( e == hipSuccess )
		e = hipEventRecord( l->ready, on );
	if( e != hipSuccess ){
		db->ctx->give( l->blk );
		snprintf( err, errlen, "tiling of the database: %s", hipGetErrorString( e ) );
		return nullptr;
	}
	db->layouts.push_back( std::move( l ) );
	return db->layouts.back().get();
}

extern "C" int rma_db_attach( rma_scanner_t *sc, rma_db_t *db, char *err, size_t errlen )
{
	if( db->device != sc->device ){
		snprintf( err, errlen, "the database lives on device %d, the scanner on device %d", db->device, sc->device );
		return 1;
	}
	HIPCHK( hipSetDevice( sc->device ) );
	return layout_for( sc, db, err, errlen ) != nullptr ? 0 : 1;
}

static DbView view_of( const rma_db *db, const Layout *l )
{
	DbView	v;
	v.codes = db->d_codes;
	v.amask = db->d_amask;
	v.base_off = db->d_base_off;
	v.slen = db->d_slen;
	v.tile_start = l->d_tile_start;
	v.tile_seq = l->d_tile_seq;
	v.tile_meta = l->d_tile_meta;
	v.concat_bases = l->concat ? l->concat_bases : 0;
	v.pos_lo = db->d_pos_lo;
	v.pos_hi = db->d_pos_hi;
	v.n_seq = db->n_seq;
	v.strands = l->strands;
	v.tile_t = l->tile_t;
	v.n_tiles = l->n_tiles;
	return v;
}

// ---------------------------------------------------------------- scans
static int launch_search( rma_scanner_t *sc, char *err, size_t errlen )
{
	const rma_scanner::InFlight	&f = sc->fly;
	HIPCHK( hipMemsetAsync( sc->d_counters, 0, RMK_N_COUNTERS * sizeof( unsigned long long ), sc->stream ) );
	rmk_search_args	a;
	a.d_prog = sc->d_prog;
	a.prog_bytes = sc->prog_bytes;
	a.qcap = f.lay->qcap;
	a.db = view_of( f.db, f.lay );
	const bool	listed = f.inst == RMK_LEAN_POOL || f.inst == RMK_LEAN_CONCAT || f.inst == RMK_LEAN_FLUSH || f.inst == RMK_LEAN_CONCAT_FLUSH;
	const bool	drain = listed && sc->glist_cap > 0;
	a.hb = HitBuf{ sc->d_hits, sc->d_counters, sc->d_counters + 1, sc->hit_cap, sc->d_spill, sc->spill_cap, sc->d_pool, sc->pool_cap,
		sc->opt.pool_min, sc->opt.pool_refill, listed ? sc->glist_cap : 0 };
	a.tile_bytes = f.tile_bytes;
	a.dbg = sc->opt.dbg | ( sc->whole_items ? 2097152 : 0 );
	HIPCHK( hipEventRecord( sc->ev[ 0 ], sc->stream ) );
	HIPCHK( rmk_launch_search( f.inst, f.grid, f.lds, sc->stream, a ) );
	sc->drained = drain;
	sc->searched = true;
	sc->efn_ran = false;
	if( drain ){	// the items the search kernel left in the list: walked by a kernel of their own
		HIPCHK( hipEventRecord( sc->ev[ 4 ], sc->stream ) );
		a.tile_bytes = sc->drain_nib;
		HIPCHK( rmk_launch_lean_drain( sc->drain_grid, sc->drain_lds, sc->stream, a ) );
	}
	HIPCHK( hipEventRecord( sc->ev[ 1 ], sc->stream ) );
	// [0] candidates, [3] queue overflow of the general instance, [RMK_GCTL] items reserved in the drain kernel's list,
	// [RMK_GCTL + 2] queue overflow of the lean instance that walks nothing: one copy, one wait
	HIPCHK( hipMemcpyAsync( sc->h_ctr, sc->d_counters, ( RMK_GCTL + 3 ) * sizeof( unsigned long long ), hipMemcpyDeviceToHost, sc->stream ) );
	return 0;
}

static void debug_report( rma_scanner_t *sc, unsigned long long count )
{
	const rma_scanner::InFlight	&f = sc->fly;
	const rmd_program_t	&dp = sc->dprog;
	const int	dbg = sc->opt.dbg;
	unsigned long long	q = 0;
	( void )hipMemcpy( &q, sc->d_counters + 2, sizeof( q ), hipMemcpyDeviceToHost );
	fprintf( stderr, "[dbg] queued items: %llu, candidates %llu (tile %d x %d, queue %d, LDS %zu, %lld tiles)\n", q, count,
		f.lay->tile_t, f.grouped ? f.lay->group : 1, f.lay->qcap, f.lds, ( long long )f.lay->n_tiles );
	if( ( dbg & 1048576 ) && f.lean ){
		unsigned long long	tl[ 5 ];
		( void )hipMemcpy( tl, sc->d_counters + 1 + 87, sizeof( tl ), hipMemcpyDeviceToHost );
		const double	t0 = double( ~tl[ 1 ] ), g = double( tl[ 0 ] );
		fprintf( stderr, "[dbg] workgroups that had tiles (%.0f of %d): out of tiles after %.1f us (mean), done after %.1f us (mean), %.1f us (last)\n", g, f.grid,
			( double( tl[ 2 ] ) / g - t0 ) * 0.01, ( double( tl[ 4 ] ) / g - t0 ) * 0.01, ( double( tl[ 3 ] ) - t0 ) * 0.01 );
	}
	if( ( dbg & 536870912 ) && sc->drained ){
		unsigned long long	bins[ 32 ];
		( void )hipMemcpy( bins, sc->d_counters + 1 + 23, sizeof( bins ), hipMemcpyDeviceToHost );
		fprintf( stderr, "[dbg] drain: waves through by 16 us from the first wave's start:" );
		for( int b = 0; b < 32; b++ )
			if( bins[ b ] )
				fprintf( stderr, " %d:%llu", b, bins[ b ] );
		fprintf( stderr, "\n" );
	}
	if( dbg & 32 ){
		unsigned long long	ph[ 6 ];
		( void )hipMemcpy( ph, sc->d_counters + 4, sizeof( ph ), hipMemcpyDeviceToHost );
		double	tot = 0;
		for( int i = 0; i < 6; i++ )
			tot += double( ph[ i ] );
		unsigned long long	lv[ 80 ];
		( void )hipMemcpy( lv, sc->d_counters + 16, sizeof( lv ), hipMemcpyDeviceToHost );
		const bool	listed = f.inst == RMK_LEAN_POOL || f.inst == RMK_LEAN_CONCAT || f.inst == RMK_LEAN_FLUSH || f.inst == RMK_LEAN_CONCAT_FLUSH;
		if( f.lean && !( listed && sc->glist_cap > 0 ) )
			fprintf( stderr, "[dbg] pool sessions: %.3g wave cycles popping (%.0f per round), %.3g stepping (%.0f per step)\n",
				double( lv[ 4 ] ), lv[ 0 ] ? double( lv[ 4 ] ) / lv[ 0 ] : 0.0, double( lv[ 5 ] ), lv[ 2 ] ? double( lv[ 5 ] ) / lv[ 2 ] : 0.0 ),
			fprintf( stderr, "[dbg] longest step %.3g cycles, most stepping in one wave (one session) %.3g cycles\n", double( lv[ 6 ] ), double( lv[ 7 ] ) );
		if( listed && sc->glist_cap > 0 ){
			// (the drain kernel's items)
			unsigned long long	g[ 2 ];
			( void )hipMemcpy( g, sc->d_counters + RMK_GCTL, sizeof( g ), hipMemcpyDeviceToHost );
			fprintf( stderr, "[dbg] drain: %llu items in the list (%llu taken), %llu walked: %.0f cycles and %.1f steps each; longest %.3g cycles, most steps %llu\n",
				g[ 0 ], g[ 1 ], lv[ 2 ], lv[ 2 ] ? double( lv[ 5 ] ) / lv[ 2 ] : 0.0, lv[ 2 ] ? double( lv[ 3 ] ) / lv[ 2 ] : 0.0, double( lv[ 6 ] ), lv[ 7 ] );
			{
				unsigned long long	lap[ 6 ];
				( void )hipMemcpy( lap, sc->d_counters + 1 + 93, sizeof( lap ), hipMemcpyDeviceToHost );
				const double	all = double( lap[ 1 ] + lap[ 2 ] + lap[ 3 ] + lap[ 4 ] ) + 1;
				fprintf( stderr, "[dbg] drain: %llu wave rounds of %.1f lanes; wave cycles taking items %.1f%%, stepping %.1f%%, complete matches %.1f%%, hand-overs %.1f%%; %.0f cycles a round\n",
					lap[ 5 ], lap[ 5 ] ? double( lap[ 0 ] ) / lap[ 5 ] : 0.0, 100 * lap[ 1 ] / all, 100 * lap[ 2 ] / all, 100 * lap[ 3 ] / all, 100 * lap[ 4 ] / all,
					lap[ 5 ] ? all / lap[ 5 ] : 0.0 );
			}
			fprintf( stderr, "[dbg] drain: items by log2( cycles ):" );
			for( int b = 8; b < 32; b++ )
				if( lv[ 8 + b ] )
					fprintf( stderr, " %d:%llu", b, lv[ 8 + b ] );
			fprintf( stderr, "\n[dbg] drain: items by complete matches (0, 1, 2-3, 4-7, ...: count, mean cycles):" );
			for( int kk = 0; kk < 16; kk++ )
				if( lv[ 45 + kk ] )
					fprintf( stderr, " %llu,%.0f", lv[ 45 + kk ], double( lv[ 61 + kk ] ) / lv[ 45 + kk ] );
			fprintf( stderr, "\n" );
		}else if( f.lean ){
			fprintf( stderr, "[dbg] steps by log2( cycles ):" );
			for( int b = 8; b < 32; b++ )
				if( lv[ 8 + b ] )
					fprintf( stderr, " %d:%llu", b, lv[ 8 + b ] );
			fprintf( stderr, "\n[dbg] complete matches: %llu, %.0f cycles each", lv[ 78 ], lv[ 78 ] ? double( lv[ 77 ] ) / lv[ 78 ] : 0.0 );
			fprintf( stderr, "\n[dbg] steps by deepest level (count, mean cycles):" );
			for( int kk = 0; kk < 16; kk++ )
				if( lv[ 61 + kk ] )
					fprintf( stderr, " %d:%llu,%.0f", kk, lv[ 61 + kk ], double( lv[ 45 + kk ] ) / lv[ 61 + kk ] );
			fprintf( stderr, "\n" );
		}
		if( f.lean && !( listed && sc->glist_cap > 0 ) )
			fprintf( stderr, "[dbg] pass B: %llu pop rounds of %.1f lanes, %llu steps of %.1f lanes; wave cycles popping %.1f%%, stepping %.1f%%\n",
				lv[ 0 ], lv[ 0 ] ? double( lv[ 1 ] ) / lv[ 0 ] : 0.0, lv[ 2 ], lv[ 2 ] ? double( lv[ 3 ] ) / lv[ 2 ] : 0.0,
				100.0 * lv[ 4 ] / double( lv[ 4 ] + lv[ 5 ] + 1 ), 100.0 * lv[ 5 ] / double( lv[ 4 ] + lv[ 5 ] + 1 ) );
		for( int kk = 0; kk < dp.n_searches && kk < 32 && !f.lean; kk++ )
			fprintf( stderr, "[dbg] level %2d (element %2d, type %d): %llu wave rounds, %.1f lanes each\n", kk, dp.searches[ kk ],
				dp.elems[ dp.searches[ kk ] ].type, lv[ 2 * kk ], lv[ 2 * kk ] ? double( lv[ 2 * kk + 1 ] ) / lv[ 2 * kk ] : 0.0 );
		fprintf( stderr, "[dbg] wave cycles: decode %.1f%%, literal %.1f%%, rows %.1f%%, pre-filter %.1f%%, search %.1f%%, waiting %.1f%%\n",
			100 * ph[ 0 ] / tot, 100 * ph[ 1 ] / tot, 100 * ph[ 2 ] / tot, 100 * ph[ 3 ] / tot, 100 * ph[ 4 ] / tot, 100 * ph[ 5 ] / tot );
	}
}

// The search kernel of a scan is on its way when this returns; rma_scan_end() (or search_finish())
// waits for it.  Two scanners that have begun run side by side on their own streams.
extern "C" int rma_scan_begin( rma_scanner_t *sc, const rma_db_t *db, char *err, size_t errlen )
{
	HIPCHK( hipSetDevice( sc->device ) );
	if( sc->fly.db != nullptr ){
		snprintf( err, errlen, "rma_scan_begin: the scanner has a scan in flight (rma_scan_end() ends it)" );
		return 1;
	}
	if( db->device != sc->device ){
		snprintf( err, errlen, "the database lives on device %d, the scanner on device %d", db->device, sc->device );
		return 1;
	}
	if( sc->need_efn2 && sc->d_efn2 == nullptr ){
		snprintf( err, errlen, "the program has efn2() call sites but rma_scanner_set_efn2data() was not called" );
		return 1;
	}
	sc->d_last = nullptr;
	sc->n_last = 0;
	sc->last_state = 0;
	sc->last_relabelled = false;
	const Layout	*lay = layout_for( sc, db, err, errlen );
	if( lay == nullptr )
		return 1;
	rma_scanner::InFlight	&f = sc->fly;
	f.db = db;
	f.lay = lay;
	{
		rma_db	*mdb = const_cast<rma_db *>( db );
		std::lock_guard<std::mutex>	lk( mdb->mu );
		mdb->busy.push_back( sc );
	}
	struct Unfly { rma_scanner *sc; bool armed; ~Unfly(){ if( armed ){
			rma_db	*mdb = const_cast<rma_db *>( sc->fly.db );
			std::lock_guard<std::mutex>	lk( mdb->mu );
			mdb->busy.erase( std::find( mdb->busy.begin(), mdb->busy.end(), sc ) );
			sc->fly.db = nullptr;
		} } }	unfly{ sc, true };
	if( lay->n_tiles == 0 ){
		f.grid = 0;
		unfly.armed = false;
		return 0;
	}
	const rmd_program_t	&dp = sc->dprog;
	f.lean = dp.lean_ok && !( sc->opt.dbg & 16 );
	f.grouped = f.lean && lay->group > 1;
	f.tile_bytes = lay->tile_t + dp.w_winsize + dp.lmargin + dp.rmargin + 80;
	f.lds = search_lds_bytes( sc->prog_bytes, dp, lay->tile_t, f.lean, lay->qcap, f.grouped ? SHORT_GROUP : 1, f.lean && lay->flush );
	if( f.lds > 150 * 1024 ){
		snprintf( err, errlen, "window of %d bases does not fit the LDS tile (%zu bytes needed)", dp.w_winsize, f.lds );
		return 1;
	}
	// the pooled lean instance (see the kernel): when the window of an item, four bits a base, fits the
	// column a lane gets of the tile's place in LDS
	const bool	pooled = f.lean && !f.grouped && pooled_fits( sc, lay->tile_t, lay->flush );
	if( lay->concat && f.lean && !pooled ){
		snprintf( err, errlen, "a tiling over the concatenation of the entries is for the pooled lean instance and the general ones" );	// (layout_for asks pooled_fits too)
		return 1;
	}
	if( pooled ){
		// The list of the drain kernel: room for an item per 32 bases (trna.descr leaves one per 70 before the
		// stem-loop tests and one per 4500 after them); a workgroup that finds it full walks its own items.
		int	want = !sc->opt.drain ? 0 : sc->opt.glist > 0 ? sc->opt.glist :
			int( std::min<long long>( std::max<long long>( db->sum_slen / 32, 1 << 18 ), 1 << 24 ) );
		if( lay->flush )		// (what an earlier scan reserved beyond the list's end: search_finish)
			want = std::max( want, sc->glist_need );
		const int	cap = sc->opt.pool_min + lay->qcap + sc->spill_cap;
		if( cap > sc->pool_cap || want > sc->glist_cap || ( want == 0 && sc->glist_cap != 0 ) || ( sc->opt.glist > 0 && want != sc->glist_cap && !lay->flush ) ){
			HIPCHK( hipStreamSynchronize( sc->stream ) );
			( void )hipFree( sc->d_pool );
			sc->d_pool = nullptr;
			const int	cap1 = std::max( cap, sc->pool_cap );
			sc->pool_cap = sc->glist_cap = 0;
			HIPCHK( hipMalloc( &sc->d_pool, ( size_t( want ) + size_t( sc->grid_blocks ) * cap1 ) * RMK_POOL_WORDS * sizeof( unsigned ) ) );
			sc->pool_cap = cap1;
			sc->glist_cap = want;
		}
		// the drain kernel: one wave per workgroup -- the program, a window column and the records of 64 lanes
		sc->drain_nib = ( dp.w_winsize + dp.lmargin + dp.rmargin + 14 ) / 8;	// (n_dw above: at most 32)
		sc->drain_lds = size_t( sc->prog_bytes ) + size_t( sc->drain_nib + dp.n_searches ) * 64 * sizeof( uint32_t ) + size_t( dp.n_searches ) * 64 * sizeof( uint16_t );
		const int	per_cu = int( std::min<size_t>( 4 * SEARCH_WAVES_PER_SIMD, ( 160 * 1024 ) / ( sc->drain_lds + 64 ) ) );
		sc->drain_grid = ( sc->grid_blocks / 8 ) * std::max( 1, sc->opt.drain_waves > 0 ? std::min( sc->opt.drain_waves, per_cu ) : per_cu );
	}
	// the kernel instance: lean (pooled, one tile or a group of small ones per pass), or the general one
	// compiled for the kinds of element the descriptor has
	f.inst = pooled ? ( lay->concat ? ( lay->flush ? RMK_LEAN_CONCAT_FLUSH : RMK_LEAN_CONCAT ) : lay->flush ? RMK_LEAN_FLUSH : RMK_LEAN_POOL ) : f.grouped ? RMK_LEAN_GROUP : f.lean ? RMK_LEAN_TILE :
		dp.wide ? RMK_GEN_WIDE :
		lay->concat ? ( sc->kinds == 0 ? RMK_GEN_PLAIN_CONCAT : sc->kinds == RMD_KIND_PK ? RMK_GEN_PK_CONCAT : sc->kinds == RMD_KIND_TQ ? RMK_GEN_TQ_CONCAT : RMK_GEN_PKTQ_CONCAT ) :
		sc->kinds == 0 ? RMK_GEN_PLAIN : sc->kinds == RMD_KIND_PK ? RMK_GEN_PK : sc->kinds == RMD_KIND_TQ ? RMK_GEN_TQ : RMK_GEN_PKTQ;
	const int64_t	n_units = f.grouped ? ( lay->n_tiles + SHORT_GROUP - 1 ) / SHORT_GROUP : lay->n_tiles;
	f.grid = int( std::min<int64_t>( n_units, f.lean ? sc->grid_blocks : sc->spill_blocks ) );
	if( f.lean && sc->opt.search_wgs > 0 )		// (option search_wgs: workgroups of a lean search kernel per CU -- room for another scanner's drain kernel beside it)
		f.grid = std::min( f.grid, sc->opt.search_wgs * ( sc->grid_blocks / 8 ) );
	else if( f.inst == RMK_LEAN_FLUSH || f.inst == RMK_LEAN_CONCAT_FLUSH )
		// The instance that walks nothing is compiled for five workgroups a CU (96 registers, a fifth of the LDS) and runs
		// four: a fifth measures the same (0.642 against 0.639 ms), and what it would take -- 33 KB of LDS, a wave's registers
		// on every SIMD -- is where the drain kernel and the energy kernel of the scanner that had the step before run
		// meanwhile (two scanners in turns).
		f.grid = std::min( f.grid, FLUSH_WGS_PER_CU * ( sc->grid_blocks / 8 ) );
	// the database's upload and the tiling's, on the device's upload stream, come first
	HIPCHK( hipStreamWaitEvent( sc->stream, db->ready, 0 ) );
	HIPCHK( hipStreamWaitEvent( sc->stream, lay->ready, 0 ) );
	if( launch_search( sc, err, errlen ) )
		return 1;
	unfly.armed = false;
	return 0;
}

// Wait for the search kernel of the scan in flight; repeat it while it asks for a larger spill area
// or hit buffer (count-then-emit).  On return the candidates are in d_hits, unordered, no energies.
static int search_finish( rma_scanner_t *sc, int64_t *n_hits, float *search_ms, char *err, size_t errlen )
{
	rma_scanner::InFlight	&f = sc->fly;
	*n_hits = 0;
	if( f.grid == 0 )
		return 0;
	const rmd_program_t	&dp = sc->dprog;
	unsigned long long	count = 0;
	for( int attempt = 0; ; attempt++ ){
		HIPCHK( hipStreamSynchronize( sc->stream ) );
		count = sc->h_ctr[ 0 ];
		if( sc->opt.dbg )
			debug_report( sc, count );
		bool	again = false;
		if( !f.lean ){
			// the general instance does not search queue overflow in place: a larger spill area, and again
			const unsigned long long	need = sc->h_ctr[ 3 ];
			if( need > 0 ){
				if( attempt == 3 ){
					snprintf( err, errlen, "work queue overflow after regrow (%llu items in a tile)", need );
					return 1;
				}
				( void )hipFree( sc->d_spill );
				sc->d_spill = nullptr;
				sc->spill_cap = int( need ) + 1024;
				HIPCHK( hipMalloc( &sc->d_spill, size_t( sc->spill_blocks ) * sc->spill_cap * sizeof( unsigned ) ) );
				again = true;
			}
		}
		if( f.inst == RMK_LEAN_FLUSH || f.inst == RMK_LEAN_CONCAT_FLUSH ){
			// the instance that walks nothing reports what it had no room for -- a tile's items beyond queue and spill area, the
			// list's items beyond its end -- and the scan is repeated with room for them
			const unsigned long long	q_need = sc->h_ctr[ RMK_GCTL + 2 ], l_need = sc->h_ctr[ RMK_GCTL ];
			if( ( q_need > 0 || l_need > ( unsigned long long )sc->glist_cap ) && attempt == 3 ){
				snprintf( err, errlen, "work queue or item list overflow after regrow (%llu items in a tile, %llu in the list)", q_need, l_need );
				return 1;
			}
			if( q_need > 0 ){
				( void )hipFree( sc->d_spill );
				sc->d_spill = nullptr;
				sc->spill_cap = int( q_need ) + 1024;
				HIPCHK( hipMalloc( &sc->d_spill, size_t( sc->spill_blocks ) * sc->spill_cap * sizeof( unsigned ) ) );
				again = true;
			}
			if( l_need > ( unsigned long long )sc->glist_cap ){
				if( l_need > ( 1ull << 28 ) ){
					snprintf( err, errlen, "%llu items for the drain kernel's list: more than it can be made to hold", l_need );
					return 1;
				}
				( void )hipFree( sc->d_pool );
				sc->d_pool = nullptr;
				sc->glist_need = int( l_need + l_need / 8 ) + 1024;
				const int	cap1 = sc->pool_cap;
				sc->pool_cap = sc->glist_cap = 0;
				HIPCHK( hipMalloc( &sc->d_pool, ( size_t( sc->glist_need ) + size_t( sc->grid_blocks ) * cap1 ) * RMK_POOL_WORDS * sizeof( unsigned ) ) );
				sc->pool_cap = cap1;
				sc->glist_cap = sc->glist_need;
				again = true;
			}
		}
		if( f.lean && sc->h_ctr[ 3 ] != 0 && !sc->whole_items ){
			// a piece of an item found more candidates than the order words of the pieces leave room for
			// (PIECE_ORDER_BITS): once more, and from now on, with whole items
			sc->whole_items = true;
			again = true;
		}
		if( !again && int64_t( count ) > sc->hit_cap ){
			if( attempt == 3 ){
				snprintf( err, errlen, "hit buffer overflow after regrow (%llu candidates)", count );
				return 1;
			}
			// count-then-emit: the first pass told us how many records there are
			( void )hipFree( sc->d_hits );
			sc->d_hits = nullptr;
			sc->hit_cap = int64_t( count ) + 1024;
			HIPCHK( hipMalloc( &sc->d_hits, size_t( sc->hit_cap ) * dp.hit_stride * sizeof( int32_t ) ) );
			again = true;
		}
		if( !again )
			break;
		if( launch_search( sc, err, errlen ) )
			return 1;
	}
	if( search_ms )
		HIPCHK( hipEventElapsedTime( search_ms, sc->ev[ 0 ], sc->ev[ 1 ] ) );
	*n_hits = int64_t( count );
	return 0;
}

static int launch_efn( rma_scanner_t *sc, int64_t count, char *err, size_t errlen )
{
	const rmd_program_t	&dp = sc->dprog;
	if( !( ( sc->have_efn || sc->d_efn2 != nullptr ) && dp.n_efn > 0 && count > 0 ) )
		return 0;
	// one workgroup per CU at most (its LDS), each striding over the candidates
	const int64_t	blocks = std::min<int64_t>( ( count + EFN_BLOCK - 1 ) / EFN_BLOCK, sc->grid_blocks / 8 );
	rmk_efn_args	a{ sc->d_prog, view_of( sc->fly.db, sc->fly.lay ), sc->d_hits, ( long long )count,
		sc->have_efn ? sc->d_t16 : nullptr, sc->d_tlkey, sc->d_loginc, sc->d_efn2 };
	sc->efn_ran = true;
	HIPCHK( hipEventRecord( sc->ev[ 2 ], sc->stream ) );
	// (behind the search instance that walks nothing: workgroups of one wave that find room next to another scanner's search
	// kernel -- the staged form's 136 KB of LDS wait until that kernel is through)
	const bool	light = !sc->dprog.efn_big && ( sc->opt.efn_light < 0 ? ( sc->fly.inst == RMK_LEAN_FLUSH || sc->fly.inst == RMK_LEAN_CONCAT_FLUSH ) : sc->opt.efn_light != 0 );
	if( light )
		HIPCHK( rmk_launch_efn_light( int( std::min<int64_t>( ( count + 63 ) / 64, sc->grid_blocks * 2 ) ), sc->stream, a ) );
	else
	HIPCHK( sc->dprog.efn_big ? rmk_launch_efn_big( int( blocks ), sc->stream, a ) : rmk_launch_efn( int( blocks ), sc->stream, a ) );
	HIPCHK( hipEventRecord( sc->ev[ 3 ], sc->stream ) );
	return 0;
}

static void scan_done( rma_scanner_t *sc )
{
	if( sc->fly.db == nullptr )
		return;
	rma_db	*mdb = const_cast<rma_db *>( sc->fly.db );
	{
		std::lock_guard<std::mutex>	lk( mdb->mu );
		auto	it = std::find( mdb->busy.begin(), mdb->busy.end(), sc );
		if( it != mdb->busy.end() )
			mdb->busy.erase( it );
	}
	sc->fly.db = nullptr;
	sc->fly.lay = nullptr;
}

extern "C" int rma_scan_device( rma_scanner_t *sc, const rma_db_t *db, int64_t *n_hits, float *search_ms,
	float *efn_ms, char *err, size_t errlen )
{
	*n_hits = 0;
	if( search_ms ) *search_ms = 0;
	if( efn_ms ) *efn_ms = 0;
	if( rma_scan_begin( sc, db, err, errlen ) )
		return 1;
	struct Done { rma_scanner *sc; ~Done(){ scan_done( sc ); } }	done{ sc };
	int64_t	n = 0;
	if( search_finish( sc, &n, search_ms, err, errlen ) )
		return 1;
	*n_hits = n;
	const bool	has_efn = ( sc->have_efn || sc->d_efn2 != nullptr ) && sc->dprog.n_efn > 0 && n > 0;
	if( launch_efn( sc, n, err, errlen ) )
		return 1;
	HIPCHK( hipStreamSynchronize( sc->stream ) );
	if( efn_ms && has_efn )
		HIPCHK( hipEventElapsedTime( efn_ms, sc->ev[ 2 ], sc->ev[ 3 ] ) );
	return 0;
}

// The kernels of the scanner's last search, by HIP events on its stream: ms[ 0 ] the search kernel, ms[ 1 ] the
// drain kernel that walked what it left in the list (0: there was none), ms[ 2 ] the efn kernel (0: none).
extern "C" int rma_scanner_last_kernel_ms( rma_scanner_t *sc, float ms[ 3 ], char *err, size_t errlen )
{
	HIPCHK( hipSetDevice( sc->device ) );
	ms[ 0 ] = ms[ 1 ] = ms[ 2 ] = 0;
	HIPCHK( hipStreamSynchronize( sc->stream ) );
	if( !sc->searched )
		return 0;		// (nothing was launched yet)
	if( sc->drained ){
		HIPCHK( hipEventElapsedTime( &ms[ 0 ], sc->ev[ 0 ], sc->ev[ 4 ] ) );
		HIPCHK( hipEventElapsedTime( &ms[ 1 ], sc->ev[ 4 ], sc->ev[ 1 ] ) );
	}else
		HIPCHK( hipEventElapsedTime( &ms[ 0 ], sc->ev[ 0 ], sc->ev[ 1 ] ) );
	if( sc->efn_ran )
		HIPCHK( hipEventElapsedTime( &ms[ 2 ], sc->ev[ 2 ], sc->ev[ 3 ] ) );
	return 0;
}

// Energies, reference order, copy back.  With on_device_only the ordered records stay in HBM
// (rma_scanner::d_last, for rma_gather_hits) and *hits is not set.
static int scan_end( rma_scanner_t *sc, const int32_t **hits, int64_t *n_hits, bool copy_back, char *err, size_t errlen )
{
	*n_hits = 0;
	if( hits )
		*hits = nullptr;
	if( sc->fly.db == nullptr ){
		snprintf( err, errlen, "rma_scan_end: no scan in flight" );
		return 1;
	}
	HIPCHK( hipSetDevice( sc->device ) );
	struct Done { rma_scanner *sc; ~Done(){ scan_done( sc ); } }	done{ sc };
	const bool	timing = sc->opt.timing != 0;
	auto	t0 = std::chrono::steady_clock::now();
	auto lap = [&]( const char *what ){
		if( timing ){
			auto	t1 = std::chrono::steady_clock::now();
			fprintf( stderr, "[timing] %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>( t1 - t0 ).count() );
			t0 = t1;
		}
	};
	int64_t	n = 0;
	if( search_finish( sc, &n, nullptr, err, errlen ) )
		return 1;
	lap( "search" );
	*n_hits = n;
	sc->last_state = 1;		// (no records: nothing to be anywhere)
	if( n == 0 )
		return 0;
	sc->last_state = 2;		// (until the ordered records are known to be in HBM)
	if( launch_efn( sc, n, err, errlen ) )
		return 1;
	if( timing ){
		HIPCHK( hipStreamSynchronize( sc->stream ) );
		lap( "energies" );
	}
	const rma_db	*db = sc->fly.db;
	const int	stride = sc->dprog.hit_stride;
	// pinned staging buffer: the copy back is a single DMA
	const size_t	words = size_t( n ) * stride;
	if( copy_back && words > sc->h_raw_cap ){
		if( sc->h_raw != nullptr )
			( void )hipHostFree( sc->h_raw );
		sc->h_raw = nullptr;
		sc->h_raw_cap = 0;
		HIPCHK( hipHostMalloc( reinterpret_cast<void **>( &sc->h_raw ), words * 2 * sizeof( int32_t ), hipHostMallocDefault ) );
		sc->h_raw_cap = words * 2;
	}
	// Reference order -- (entry, strand, start, rank, order), order word renumbered -- on the device,
	// behind the efn kernel on the same stream: what comes back is the final stream (rm_hitsort_dev.h).
	// Header words that do not fit the 64-bit key (or host_sort): the host's sort_hits().
	bool	on_device = false;
	if( !sc->opt.host_sort && n >= 2 ){
		if( sc->dsort.reserve( sc->hit_cap, stride ) == hipSuccess &&
			sc->dsort.run( sc->d_hits, n, rma::bits_of( unsigned( db->n_seq > 0 ? db->n_seq - 1 : 0 ) ), rma::bits_of( unsigned( db->max_slen ) ),
				rma::bits_of( unsigned( sc->dprog.w_winsize ) ), sc->stream ) == hipSuccess ){
			int	flag = 1;
			if( timing ){
				HIPCHK( hipStreamSynchronize( sc->stream ) );
				lap( "sorted" );
			}
			if( copy_back )
				HIPCHK( hipMemcpyAsync( sc->h_raw, sc->dsort.d_out, words * sizeof( int32_t ), hipMemcpyDeviceToHost, sc->stream ) );
			HIPCHK( hipMemcpyAsync( sc->h_ctr, sc->dsort.d_flag, sizeof( int ), hipMemcpyDeviceToHost, sc->stream ) );
			HIPCHK( hipStreamSynchronize( sc->stream ) );
			memcpy( &flag, sc->h_ctr, sizeof( flag ) );
			on_device = flag == 0;
			if( !on_device && sc->dsort.w_ord < 31 )
				sc->dsort.w_ord = 31;	// (order words above 255: room for them from now on, if the other fields leave it)
		}else
			( void )hipGetLastError();
	}
	if( on_device ){
		lap( "ordered" );
		sc->d_last = sc->dsort.d_out;
		sc->n_last = n;
		sc->last_state = 1;
		if( hits )
			*hits = copy_back ? sc->h_raw : nullptr;
		return 0;
	}
	if( n == 1 && !sc->opt.host_sort ){
		// (a single record is in order; its order word -- the kernels leave the number of the walk's choices
		// there, or a count within a piece of an item -- is 0 as the sorts would make it)
		HIPCHK( hipMemsetAsync( sc->d_hits + 4, 0, sizeof( int32_t ), sc->stream ) );
		if( copy_back )
			HIPCHK( hipMemcpyAsync( sc->h_raw, sc->d_hits, words * sizeof( int32_t ), hipMemcpyDeviceToHost, sc->stream ) );
		HIPCHK( hipStreamSynchronize( sc->stream ) );
		sc->d_last = sc->d_hits;
		sc->n_last = 1;
		sc->last_state = 1;
		if( hits )
			*hits = copy_back ? sc->h_raw : nullptr;
		return 0;
	}
	if( words > sc->h_raw_cap ){		// (not asked to copy back, but the host has to order)
		if( sc->h_raw != nullptr )
			( void )hipHostFree( sc->h_raw );
		sc->h_raw = nullptr;
		sc->h_raw_cap = 0;
		HIPCHK( hipHostMalloc( reinterpret_cast<void **>( &sc->h_raw ), words * 2 * sizeof( int32_t ), hipHostMallocDefault ) );
		sc->h_raw_cap = words * 2;
	}
	HIPCHK( hipMemcpyAsync( sc->h_raw, sc->d_hits, words * sizeof( int32_t ), hipMemcpyDeviceToHost, sc->stream ) );
	HIPCHK( hipStreamSynchronize( sc->stream ) );
	lap( "copy back" );
	sc->h_sorted.resize( words );
	rma::sort_hits( sc->h_raw, n, stride, sc->h_sorted.data(), sc->keys, sc->keys_tmp );
	lap( "ordering" );
	if( !copy_back ){
		// the ordered records back where a device-side consumer finds them
		HIPCHK( hipMemcpyAsync( sc->d_hits, sc->h_sorted.data(), words * sizeof( int32_t ), hipMemcpyHostToDevice, sc->stream ) );
		HIPCHK( hipStreamSynchronize( sc->stream ) );
		sc->d_last = sc->d_hits;
		sc->n_last = n;
		sc->last_state = 1;
	}
	// (copy_back with the ordering on the host: the ordered records are on the host only, last_state stays 2 and
	// rma_gather_hits() says so instead of sending nothing)
	if( hits )
		*hits = sc->h_sorted.data();
	return 0;
}

extern "C" int rma_scan_end( rma_scanner_t *sc, const int32_t **hits, int64_t *n_hits, char *err, size_t errlen )
{
	return scan_end( sc, hits, n_hits, true, err, errlen );
}

extern "C" int rma_scan( rma_scanner_t *sc, const rma_db_t *db, const int32_t **hits, int64_t *n_hits,
	char *err, size_t errlen )
{
	*hits = nullptr;
	*n_hits = 0;
	if( rma_scan_begin( sc, db, err, errlen ) )
		return 1;
	return scan_end( sc, hits, n_hits, true, err, errlen );
}

// the ordered records of a scan left in HBM (for a collective that sends them from there)
extern "C" int rma_scan_end_on_device( rma_scanner_t *sc, const int32_t **d_hits, int64_t *n_hits, char *err, size_t errlen )
{
	*d_hits = nullptr;
	if( scan_end( sc, nullptr, n_hits, false, err, errlen ) )
		return 1;
	*d_hits = *n_hits > 0 ? sc->d_last : nullptr;
	return 0;
}

// One scan of eight start positions, so that what the runtime sets up on first use (code objects of
// the kernel instance this descriptor takes, the first device allocations of a database, the ordering's
// kernels) is paid before the first batch of a search: 15-50 ms there.  All 'a': for most
// descriptors nothing pairs; whatever is found is thrown away.  Called by the command line program
// once the scanner is complete (efn2 tables attached); the library never runs it by itself.
extern "C" int rma_scanner_warmup( rma_scanner_t *sc, char *err, size_t errlen )
{
	if( sc->prog.dminlen > 2000 )
		return 0;
	HIPCHK( hipSetDevice( sc->device ) );
	const Options	keep = sc->opt;
	sc->opt.dbg = 0;		// (no diagnostics of the warm-up)
	sc->opt.timing = 0;
	const std::string	warm( size_t( std::max( sc->prog.dminlen, 1 ) + 7 ), 'a' );
	const char	*seqs[ 1 ] = { warm.c_str() };
	const int32_t	lens[ 1 ] = { int32_t( warm.size() ) };
	rma_db_t	*wdb = nullptr;
	int	rc = rma_db_create( sc, seqs, lens, 1, &wdb, err, errlen );
	if( rc == 0 ){
		const int32_t	*wh = nullptr;
		int64_t	wn = 0;
		rc = rma_scan( sc, wdb, &wh, &wn, err, errlen );
		rma_db_destroy( wdb );
	}
	if( sc->dprog.n_efn > 0 ){
		// ... and the energy kernel, which the scan above did not reach (no candidate): one launch over no
		// records -- its code object, and the scratch memory its interval stack makes the runtime set
		// aside at a kernel's first launch (9 ms in the first batch, measured)
		( void )rmk_preload_efn();
		rmk_efn_args	a{ sc->d_prog, DbView{}, sc->d_hits, 0ll, sc->have_efn ? sc->d_t16 : nullptr, sc->d_tlkey, sc->d_loginc, sc->d_efn2 };
		( void )rmk_launch_efn( 1, sc->stream, a );
		( void )hipStreamSynchronize( sc->stream );
		( void )hipGetLastError();
	}
	if( rc == 0 && sc->dsort.cap >= 4096 ){
		// ... and one pass of the ordering over a cleared hit buffer
		( void )hipMemsetAsync( sc->d_hits, 0, size_t( 4096 ) * sc->dprog.hit_stride * sizeof( int32_t ), sc->stream );
		( void )sc->dsort.run( sc->d_hits, 4096, 10, 20, 8, sc->stream );
		// ... and the way back of the records, as a scan ends: the first copy of this size from the device
		// sets up the engine that makes it
		if( sc->h_raw_cap >= size_t( 4096 ) * sc->dprog.hit_stride )
			( void )hipMemcpyAsync( sc->h_raw, sc->dsort.d_out, size_t( 4096 ) * sc->dprog.hit_stride * sizeof( int32_t ), hipMemcpyDeviceToHost, sc->stream );
		( void )hipStreamSynchronize( sc->stream );
		( void )hipGetLastError();
	}
	sc->opt = keep;
	return rc;
}

// ---------------------------------------------------------------- for rm_gather.cpp
int rma_scanner_device( const rma_scanner_t *sc ) { return sc->device; }
hipStream_t rma_scanner_stream( const rma_scanner_t *sc ) { return sc->stream; }
int rma_scanner_stride( const rma_scanner_t *sc ) { return sc->dprog.hit_stride; }
void rma_scanner_last( const rma_scanner_t *sc, const int32_t **d_hits, int64_t *n ) { *d_hits = sc->d_last; *n = sc->n_last; }
// where the last scan's ordered records are: 0 no scan has ended, 1 in HBM (d_last; also a scan without records), 2 on the host only
int rma_scanner_last_state( const rma_scanner_t *sc ) { return sc->last_state; }
bool rma_scanner_last_relabelled( const rma_scanner_t *sc ) { return sc->last_relabelled; }
void rma_scanner_set_relabelled( rma_scanner_t *sc ) { sc->last_relabelled = true; }

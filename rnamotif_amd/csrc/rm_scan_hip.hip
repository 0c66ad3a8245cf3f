// rm_scan_hip.hip -- the scan path on MI355X (gfx950): search kernel, efn
// kernel, and the scanner / database halves of the C ABI (include/rnamotif_amd.h).
//
// What runs here is the reference's RM_find_motif() for every start position of
// every sequence and strand (/root/reference/src/find_motif.c:164-207), one
// lane per start position, and RM_efn() (/root/reference/src/efn.c:1162) for
// every candidate and efn() call site, one lane per candidate.
//
// Data layout in HBM (details in DESIGN.md):
//   codes  2 bits/base, 16 bases per uint32; amask 1 bit/base, 32 per uint32;
//          every sequence starts on a 32-base boundary
//   hits   fixed-stride int32 records, appended through one atomic counter
// A workgroup owns a tile of T consecutive start positions of one strand; it
// decodes the T + w - 1 (+ context margins) bases the tile can touch into LDS
// as one byte per base (codes 0..4, reverse strand complemented on the fly) and
// keeps the motif program in LDS as well.  Tiles are handed out through an
// atomic ticket so that long searches do not stall a fixed schedule.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <chrono>
#include <cmath>
#include <string>
#include <vector>

#define RMD_HD		__host__ __device__ inline
#define RMD_FN		static __device__ inline
#define RMD_COLD	static __device__ inline
#define RMD_FN_MEMBER	__device__ inline
#include "rm_scan_core.h"
#include "rm_efn_core.h"
#include "rm_efn2_core.h"
#include "rm_fasta.h"
#include "rm_pack.h"
#include "rm_hitsort.h"
#include "rm_hitsort_dev.h"
#include "rnamotif_amd.h"

// ---------------------------------------------------------------- device views
struct DbView {
	const uint32_t	*codes, *amask;
	const int64_t	*base_off;	// [n_seq]   first base of sequence s (multiple of 32)
	const int32_t	*slen;		// [n_seq]
	const int64_t	*tile_start;	// [n_seq+1] prefix sum of tiles over sequences
	const int32_t	*tile_seq;	// [n_tiles] sequence of every tile (saves a search per tile)
	const int32_t	*pos_lo, *pos_hi;	// [n_seq] or null: only start positions lo <= szero < hi (each strand)
	int32_t	n_seq, strands, tile_t;
	int64_t	n_tiles;
};

struct HitBuf {
	int32_t	*hits;
	unsigned long long	*count;		// candidates found (may exceed cap)
	unsigned long long	*ticket;	// next tile
	int64_t	cap;
	unsigned	*spill;			// [gridDim.x][spill_cap] work queue items that did not fit the LDS queue
	int	spill_cap;
	unsigned	*pool;			// [gridDim.x][pool_cap][3] pooled instance: items that passed the tile's tests
	int	pool_cap, pool_min;	// ... searched once pool_min of them have come together
	int	pool_refill;		// idle lanes of a wave that pop together
};

__device__ inline int db_code( const DbView &db, int64_t base )	// forward strand code of absolute base
{
	uint32_t	am = db.amask[ base >> 5 ];
	if( ( am >> ( base & 31 ) ) & 1 )
		return RMA_BC_N;
	return ( db.codes[ base >> 4 ] >> ( 2 * ( base & 15 ) ) ) & 3;
}

// code at strand position p of sequence (off,slen), strand comp (mk_rcmp, rnamot.c:193)
__device__ inline int db_strand_code( const DbView &db, int64_t off, int slen, int comp, int p )
{
	int	c = db_code( db, off + ( comp ? slen - 1 - p : p ) );
	return ( comp && c < 4 ) ? 3 - c : c;
}

struct DevSink {
	HitBuf	hb;
	int	seq, comp, stride;
	__device__ inline void put( const rmd_program_t *P, const rmd_lane_t *L, int szero )
	{
		unsigned long long	slot = atomicAdd( hb.count, 1ull );
		if( slot < ( unsigned long long )hb.cap )
			rmd_fill_hit( P, L, seq, comp, szero, hb.hits + slot * stride );
	}
};

#ifndef QCAP
#define QCAP		1024		// work queue entries per workgroup
#endif

// Lean path records of one lane in LDS, 6 bytes per level: windows of lean descriptors are
// shorter than 4096 (rmd_build), so window start and saved end take 12 bits each, the next
// end position (down to -2) 13, the helix length 6, the phase 1.
// Level k: a dword at lo[ k * BLOCK ] and a half word at hi[ k * BLOCK ], lane-contiguous.
#define LEAN_REC_BYTES	6
template< int BLOCK >
struct LdsRecs {
	uint32_t	*lo;
	uint16_t	*hi;
	__device__ inline rmd_lrec_t	get( int k ) const
	{
		const uint32_t	a = lo[ k * BLOCK ];
		const uint32_t	b = hi[ k * BLOCK ];
		rmd_lrec_t	r;
		r.zero = int16_t( a & 0xfffu );
		r.osd = int16_t( int( ( a >> 12 ) & 0xfffu ) - 1 );
		r.sd = int16_t( int( ( a >> 24 ) | ( ( b & 0x1fu ) << 8 ) ) - 2 );
		r.hl = uint8_t( ( b >> 5 ) & 0x3fu );
		r.ph = uint8_t( b >> 11 );
		return r;
	}
	__device__ inline void	set( int k, rmd_lrec_t v )
	{
		// saved end >= -1 (empty interior at the window start), next end >= -2 (one below an
		// element that may be empty at position 0): stored with offsets 1 and 2
		const uint32_t	sd1 = uint32_t( int( v.sd ) + 2 ) & 0x1fffu;
		lo[ k * BLOCK ] = ( uint32_t( v.zero ) & 0xfffu ) | ( ( uint32_t( int( v.osd ) + 1 ) & 0xfffu ) << 12 ) | ( sd1 << 24 );
		hi[ k * BLOCK ] = uint16_t( ( ( sd1 >> 8 ) & 0x1fu ) | ( uint32_t( v.hl ) << 5 ) | ( uint32_t( v.ph ) << 11 ) );
	}
};

// General path records of one lane in LDS (rmd_grec_t, 12 bytes per level): three dwords at
// w[ ( 3 * k + j ) * BLOCK ], lane-contiguous, so a wave's access is conflict free whatever
// levels its lanes are on.  Dword 0: window start | saved window end; dword 1: next end position |
// first loop variable; dword 2: second loop variable | helix length | phase.
#define GEN_REC_BYTES	12
template< int BLOCK >
struct LdsGRecs {
	uint32_t	*w;
	// levels up to the split level: the iterator each one's alternative was resumed from (rmd_gen_step)
	uint32_t	*bw;
	// rmd_program_t::rec_off (in LDS): where level k's record starts; levels with a single
	// alternative keep the window dword only (their iterator is "the whole window, taken")
	const int16_t	*off;
	__device__ inline rmd_grec_t	get( int k ) const
	{
		const int	o = off[ k ];
		const uint32_t	d0 = w[ ( o & 0x7fff ) * BLOCK ];
		rmd_grec_t	r;
		r.zero = int16_t( d0 & 0xffffu );
		r.osd = int16_t( d0 >> 16 );
		if( o < 0 ){
			r.sd = int16_t( r.osd - 1 );
			r.a = r.c = 0;
			r.hl = 0;
			r.ph = 1;
			return r;
		}
		rmd_grec_set_words( r, w[ ( o + 1 ) * BLOCK ], w[ ( o + 2 ) * BLOCK ] );
		return r;
	}
	__device__ inline void	set_iter_words( int k, uint32_t d1, uint32_t d2 )
	{
		const int	o = off[ k ];
		if( o >= 0 ){
			w[ ( o + 1 ) * BLOCK ] = d1;
			w[ ( o + 2 ) * BLOCK ] = d2;
		}
	}
	__device__ inline void	set_iter( int k, rmd_grec_t v ) { set_iter_words( k, rmd_grec_word1( v ), rmd_grec_word2( v ) ); }
	__device__ inline void	set_window( int k, int zero, int osd )
	{
		w[ ( off[ k ] & 0x7fff ) * BLOCK ] = ( uint32_t( zero ) & 0xffffu ) | ( uint32_t( osd ) << 16 );
	}
	__device__ inline void	set( int k, rmd_grec_t v )
	{
		set_window( k, v.zero, v.osd );
		set_iter( k, v );
	}
	__device__ inline void	set_zero( int k, int zero )
	{
		uint32_t	&d0 = w[ ( off[ k ] & 0x7fff ) * BLOCK ];
		d0 = ( d0 & 0xffff0000u ) | ( uint32_t( zero ) & 0xffffu );
	}
	__device__ inline void	set_osd( int k, int osd )
	{
		uint32_t	&d0 = w[ ( off[ k ] & 0x7fff ) * BLOCK ];
		d0 = ( d0 & 0xffffu ) | ( uint32_t( osd ) << 16 );
	}
	__device__ inline int	hl( int k ) const { return int( ( w[ ( off[ k ] + 2 ) * BLOCK ] >> 16 ) & 0xffu ); }	// (helix levels only)
	__device__ inline void	set_before( int k, rmd_grec_t v )
	{
		bw[ ( 2 * k ) * BLOCK ] = rmd_grec_word1( v );
		bw[ ( 2 * k + 1 ) * BLOCK ] = rmd_grec_word2( v );
	}
	__device__ inline uint32_t	before_word( int i ) const { return bw[ i * BLOCK ]; }
};

// rmd_gen_step()'s hand-over of the alternatives of the split level: a queue of continuations in
// LDS, walked in the tile's second round.  Entry: the work item, the alternative's number, and
// two words per level 0..S (LdsGRecs::set_before).  A full queue refuses: the lane walks on itself.
#define DEEP_QUEUE	128
struct LdsSplit {
	int	S;
	uint32_t	*dq;
	int	*dq_n;
	const unsigned	*item;
	__device__ inline int	level() const { return S; }
	__device__ inline int	entry_words() const { return 2 + 2 * ( S + 1 ); }
	template< class GR >
	__device__ inline bool	push( const rmd_gen_t &, GR &gr, int alt ) const
	{
		const int	slot = atomicAdd( dq_n, 1 );
		if( slot >= DEEP_QUEUE )
			return false;
		uint32_t	*e = dq + slot * entry_words();
		e[ 0 ] = *item;
		e[ 1 ] = uint32_t( alt );
		for( int i = 0; i < 2 * ( S + 1 ); i++ )
			e[ 2 + i ] = gr.before_word( i );
		return true;
	}
};

// 64 bits of a bit vector starting at bit q: three dwords through two v_alignbit_b32
__device__ inline unsigned long long bits64( const unsigned long long *row, int q )
{
	const uint32_t	*r = reinterpret_cast<const uint32_t *>( row ) + ( q >> 5 );
	const uint32_t	d0 = r[ 0 ], d1 = r[ 1 ], d2 = r[ 2 ];
	const uint32_t	lo = __builtin_amdgcn_alignbit( d1, d0, uint32_t( q & 31 ) );
	const uint32_t	hi = __builtin_amdgcn_alignbit( d2, d1, uint32_t( q & 31 ) );
	return ( ( unsigned long long )hi << 32 ) | lo;
}

// The tail test of rmd_lean_step() from the pre-filter's bit vectors.  pb[c][q] says "the base
// at q pairs with 5' base c"; when the pair table is symmetric it also says "the base at q,
// as the 5' partner, pairs with 3' base c".  For a tail helix that allows no mispair and must
// have both ends paired, "some admissible 5' end s starts it against the 3' end b" is then an
// AND over its first minlen pairs of windows of those rows -- a few dozen instructions
// instead of one first-pairs test per admissible s.
struct TailAccel {
	const rmd_program_t	*P;
	const unsigned long long	*pb;
	const uint8_t	*tile;
	int	pb_words, p_lo, vec_bits;
	int	usable_for;		// element index (level-0 helix) the rows were built for, or -1
	__device__ inline bool	tail( const rmd_elem_t &stp, int z, int a, int b, bool *res ) const
	{
		if( usable_for < 0 || stp.searchno != 0 )
			return false;
		const rmd_elem_t	&t = P->elems[ P->searches[ stp.tail_s ] ];
		int	s_hi = b - t.minglen + 1, s_lo = b - t.maxglen + 1;
		if( a + stp.tail_pre_min > s_lo )
			s_lo = a + stp.tail_pre_min;
		if( stp.tail_pre_max >= 0 && a + stp.tail_pre_max < s_hi )
			s_hi = a + stp.tail_pre_max;
		const int	n = s_hi - s_lo + 1;
		if( n <= 0 ){
			*res = false;
			return true;
		}
		if( n > 64 )
			return false;
		unsigned long long	m = n == 64 ? ~0ull : ( 1ull << n ) - 1;
		for( int j = 0; j < t.minlen && m; j++ ){
			const int	c3 = tile[ z + b - j - p_lo ];
			const int	q = z + s_lo + j - p_lo + 64;		// bit of 5' position s_lo + j
			if( c3 > 4 || q < 0 || q + 64 > vec_bits )
				return false;
			m &= bits64( pb + c3 * pb_words, q );
		}
		*res = m != 0;
		return true;
	}
};

// Pair rows of one pair table over a tile: rows5[ b * pb_words ] has a bit per tile position (64
// pad bits in front) whose base can be the 3' partner of 5' base b.  End positions w0 .. w0+63 of
// a helix whose 5' strand starts at s5: bit i of the result is set when the first hl0 pairs of
// (s5, w0+i) hold with at most lim mispairs (and, with ends5, the first pair itself holds) --
// match_wchlx's rule for reaching length minlen, find_motif.c:1010-1033,1065-1080; positions
// below lo are cleared.
__device__ inline unsigned long long rows_win( const unsigned long long *rows5, int pb_words, const uint8_t *tile, int p_lo,
	int hl0, int lim, bool ends5, int s5, int w0, int lo )
{
	unsigned long long	W;
	if( lim == 0 ){
		// no mispair allowed: a plain AND of the shifted rows, done as soon as no end position is left
		W = ~0ull;
		for( int h = 0; h < hl0 && W; h++ ){
			const int	qq = w0 - h - p_lo + 64;	// bit index into the padded vector
			W &= qq >= 0 ? bits64( rows5 + tile[ s5 + h - p_lo ] * pb_words, qq ) : 0ull;
		}
	}else{
		unsigned long long	c1 = 0, c2 = 0, c3 = 0, c4 = 0, first = 0;	// >= 1/2/3/4 mispairs
		for( int h = 0; h < hl0; h++ ){
			const int	qq = w0 - h - p_lo + 64;
			const unsigned long long	mis = ~( qq >= 0 ? bits64( rows5 + tile[ s5 + h - p_lo ] * pb_words, qq ) : 0ull );
			if( h == 0 )
				first = mis;
			c4 |= c3 & mis;
			c3 |= c2 & mis;
			c2 |= c1 & mis;
			c1 |= mis;
		}
		W = ~( lim == 1 ? c2 : lim == 2 ? c3 : c4 );
		if( ends5 )
			W &= ~first;
	}
	const int	imin = lo - w0;
	if( imin > 0 )
		W = imin >= 64 ? 0 : W & ( ~0ull << imin );
	return W;
}

// rmd_gen_skip_ends()'s accelerator: the 3' ends that can start helix stp, from the pair rows of
// its pair table (rmd_elem_t::rows names the row set; -1: none, the core tests end by end)
template< int KINDS >
struct RowEnds {
	static constexpr int	kinds = KINDS;	// element kinds the instance is compiled for (rm_scan_core.h)
	const unsigned long long	*rows;		// row set 0; set j at rows + 5 * j * pb_words
	const uint8_t	*tile;
	int	pb_words, p_lo, vec_bits;
	__device__ inline bool	ends( const rmd_elem_t &stp, int s5, int top, int lo, uint64_t *mask ) const
	{
		if( stp.rows < 0 )
			return false;
		const int	w0 = top - 63, q_hi = w0 - p_lo + 64;
		if( q_hi + 96 > vec_bits || s5 < p_lo )		// (bits64 reads three dwords from its first bit)
			return false;
		const int	lim = ( stp.ends & RMA_5PAIRED ) ? stp.mplim : ( stp.mplim > 1 ? stp.mplim : 1 );
		*mask = rows_win( rows + 5 * stp.rows * pb_words, pb_words, tile, p_lo, stp.minlen, lim,
			( stp.ends & RMA_5PAIRED ) != 0, s5, w0, lo );
		return true;
	}
};

// ---------------------------------------------------------------- general instance, pass B
// What the search of a tile needs of the kernel's state: pass B of the general instance.  (The
// pre-filter no longer searches queue overflow in place -- three inlined copies of the state
// machine in its loops made them 2-3 times slower; the host repeats the launch with a larger
// spill area instead, rma_scan_device.)
struct GenTile {
	const rmd_program_t	*P;
	uint32_t	*recs, *before, *deep;		// LDS: records, resume states, queue of continuations
	const unsigned	*queue, *spill;
	int	qcap, nq;
	int	*qhead, *dqn, *dqhead;			// LDS counters
	const unsigned long long	*pb;
	const uint8_t	*tile;
	int	pb_words, p_lo, vec_words, slen, z0, split_s, dbg;
};

// (inlined: as a function of its own, called once per tile, it ran 15-25 % slower -- profiles/matrix_r2.sh)
#ifndef PASS_B_ATTR
#define PASS_B_ATTR	__attribute__(( always_inline ))
#endif
template< int BLOCK, int KINDS >
__device__ PASS_B_ATTR void general_pass_b( const GenTile gt, DevSink sink )
{
	const rmd_program_t	*const P = gt.P;
	const int	lane_id = threadIdx.x & 63, dbg = gt.dbg, qcap = gt.qcap, nq = gt.nq, split_s = gt.split_s;
	const int	slen = gt.slen, z0 = gt.z0, p_lo = gt.p_lo, pb_words = gt.pb_words, vec_words = gt.vec_words;
	const unsigned long long	lt_mask = ( 1ull << lane_id ) - 1;
	const unsigned	*const queue = gt.queue, *const spill = gt.spill;
	const unsigned long long	*const pb = gt.pb;
	const uint8_t	*const tile = gt.tile;
	uint32_t	*const g_deep = gt.deep;
	int	&s_qhead = *gt.qhead, &s_dqn = *gt.dqn, &s_dqhead = *gt.dqhead;
	const HitBuf	&hb = sink.hb;
	LdsGRecs<BLOCK>	gr{ gt.recs + threadIdx.x, gt.before + threadIdx.x, P->rec_off };
	rmd_seq_t	sq{ tile, p_lo };
	rmd_lane_t	lane;
	int	k = -1;
	bool	dry = false;
			// every element type: 12 bytes of search state per level, in LDS (rmd_grec_t)
			rmd_gen_t	st;
			unsigned	cur_item = 0;
			const RowEnds<KINDS>	ends{ pb, tile, pb_words, p_lo, ( dbg & 64 ) ? 0 : vec_words * 64 };	// (bit 64: end by end, no rows)
			const LdsSplit	split{ split_s, g_deep, &s_dqn, &cur_item };
			// Two rounds over the tile.  Round 0: the work items, down to the split level; what
			// survives there is queued as a continuation (LdsSplit), so the lanes stay together on the
			// first levels.  Round 1: the continuations, each walked from below the split level to
			// its end by one lane.  Without a split level round 0 walks everything.
			for( int round = 0; round < ( split_s >= 0 ? 2 : 1 ); round++ ){
				if( round == 1 ){
					__syncthreads();
					k = -1;
					dry = false;
				}
				const int	n_work = round == 0 ? nq : ( s_dqn < DEEP_QUEUE ? s_dqn : DEEP_QUEUE );
				int	*const head = round == 0 ? &s_qhead : &s_dqhead;
				for( ; ; ){
					const unsigned long long	want = __ballot( k < 0 && !dry );
					if( want ){
						int	base = 0;
						if( lane_id == __ffsll( want ) - 1 )
							base = atomicAdd( head, __popcll( want ) );
						base = __shfl( base, __ffsll( want ) - 1 );
						if( k < 0 && !dry ){
							const int	i = base + __popcll( want & lt_mask );
							if( i >= n_work )
								dry = true;
							else if( round == 0 ){
								cur_item = i < qcap ? queue[ i ] :
									__hip_atomic_load( spill + ( i - qcap ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
								const int	r = int( cur_item & 0xffffu );
								k = rmd_gen_begin( P, gr, st, z0 + int( cur_item >> 16 ), slen,
									r == 0xffff ? 0 : r, r == 0xffff ? RMD_ALL_RANKS : 1 );
							}else{
								const uint32_t	*e = g_deep + i * split.entry_words();
								cur_item = e[ 0 ];
								const int	r = int( cur_item & 0xffffu );
								k = rmd_gen_resume( P, gr, st, sq, z0 + int( cur_item >> 16 ), slen,
									r == 0xffff ? 0 : r, r == 0xffff ? RMD_ALL_RANKS : 1, split_s, e + 2, int( e[ 1 ] ), ends );
							}
						}
					}
					if( __ballot( k >= 0 ) == 0 )
						break;
					if( dbg & 32 ){
						// diagnostic: per search level, wave rounds with a lane on it and lanes served
						for( int kk = 0; kk < P->n_searches; kk++ ){
							const unsigned long long	mk = __ballot( k == kk );
							if( mk && lane_id == 0 ){
								atomicAdd( hb.ticket + 15 + 2 * kk, 1ull );
								atomicAdd( hb.ticket + 16 + 2 * kk, ( unsigned long long )__popcll( mk ) );
							}
						}
					}
					// Lanes on different levels run different code, one level after the other: serve the
					// level most lanes are on and let the others wait for company (they are served once the
					// lanes ahead of them have gone dry or caught up), instead of a round per level for a
					// lane or two each.
					int	serve = k;
					if( !( dbg & 128 ) ){
						int	most = 0;
						for( int kk = 0; kk < P->n_searches; kk++ ){
							const int	n = __popcll( __ballot( k == kk ) );
							if( n > most ){
								most = n;
								serve = kk;
							}
						}
					}
					if( k >= 0 && k == serve ){
						if( round == 0 )
							k = rmd_gen_step( P, gr, st, sq, k, &lane, sink, ends, split, -1, !( dbg & 512 ) );
						else{
							k = rmd_gen_step( P, gr, st, sq, k, &lane, sink, ends, rmd_no_split_t(), split_s, !( dbg & 512 ) );
							if( k <= split_s )
								k = -1;		// back at the split level: this alternative is done
						}
					}
				}
			}
		}

// ---------------------------------------------------------------- search kernel
#ifndef SEARCH_WAVES_PER_SIMD
#define SEARCH_WAVES_PER_SIMD	4
#endif
// LEAN: the descriptor has only ss and proper helices (rmd_program_t::lean_ok) -- pass B keeps
// 8 bytes of state per level in LDS; the general state machine is not compiled into that
// instance at all (no scratch frames, fewer registers).
// (the general instance is bound by the latency of its dependent LDS accesses: four workgroups per
// CU where its records, 12 bytes per level and lane, leave room for them -- measured against three
// at 168 registers and against out-of-line level generators, profiles/matrix_r2.sh)
#ifndef GENERAL_WAVES_PER_SIMD
#define GENERAL_WAVES_PER_SIMD	4
#endif
// (... three with 168 registers for descriptors with triplexes / 4-plexes: qu+tr 46.4 -> 39.2 ms, where
// pk1 goes 7.5 -> 8.6 ms)
#define GENERAL_WAVES( kinds_ )	( ( ( kinds_ ) & RMD_KIND_TQ ) ? 3 : GENERAL_WAVES_PER_SIMD )
#ifndef SHORT_GROUP
#define SHORT_GROUP		16	// tiles per workgroup pass for databases of short entries
#endif
#define SHORT_ENTRY_MEAN	4000
#define SPILL_ITEMS		8192	// queue items per workgroup that may overflow into HBM (32 KB each, 64 MB in all)	// ... which are those whose entries average less than this
// G: tiles per workgroup pass.  G == 1: one tile, all lanes on it.  G > 1 (databases of short
// entries, lean descriptors only): a group of G small tiles, each in its own LDS slot and
// pre-filtered by one wave, feeding ONE work queue -- a tile of a 500 base entry yields a few
// dozen items, far too few for 256 lanes, and pass B is where the time goes.
// KINDS (general instance): the element kinds it is compiled for, RMD_KIND_PK | RMD_KIND_TQ -- an
// instance per class of descriptor, so that a pseudoknot search does not carry the 4-plex code.
// POOL (lean, G == 1): pass B does not run tile by tile.  A tile of 10 K start positions leaves some
// hundred items that pass the first-pairs and tail tests -- 256 lanes each walk one for a few steps
// and then wait for the slowest (14 of 64 lanes busy, measured).  The pooled instance appends the
// survivors of every tile to a pool in the workgroup's HBM area instead and searches them once
// pool_min have come together: a lane pops an item, rebuilds its window from the packed database
// into a column of LDS (4 bits per base; the tile and its bit vectors are not needed then and lend
// their place), walks it, and pops the next -- every lane busy until the pool runs dry.
template< int BLOCK, bool LEAN, int G, int KINDS = 0, bool POOL = false >
__global__ void __launch_bounds__( BLOCK, LEAN ? SEARCH_WAVES_PER_SIMD : GENERAL_WAVES( KINDS ) )
rma_search_kernel( const rmd_program_t *gP, int prog_bytes, int qcap, DbView db, HitBuf hb, int tile_bytes, int dbg )
{
	static_assert( G == 1 || ( LEAN && G % ( BLOCK / 64 ) == 0 && G <= 32 ), "tile groups: lean path, whole rounds of waves" );
	static_assert( !POOL || ( LEAN && G == 1 ), "pooled pass B: lean path, one tile per pass" );
	extern __shared__ __align__( 16 ) unsigned char	smem[];
	rmd_program_t	*P = reinterpret_cast<rmd_program_t *>( smem );
	// gP is the compact image (rmd_make_image): prog_bytes of it, a multiple of 16
	unsigned	*queue = reinterpret_cast<unsigned *>( smem + prog_bytes );
	uint8_t	*const tile0 = smem + prog_bytes + qcap * sizeof( unsigned );
	const int	slot_bytes = ( tile_bytes + 15 ) & ~15;
	__shared__ long long	s_tile;
	__shared__ int	s_seq, s_qn, s_qhead, s_dqn, s_dqhead;
	__shared__ int	s_pool_n, s_pool_head;
	__shared__ int	s_ctx[ G ][ G > 1 ? 8 : 1 ];	// G > 1: seq, comp, slen, z0, p_lo, vec_words of every slot
	const int	tid = threadIdx.x;
	// lanes that share a tile in pass A: the workgroup, or one wave per slot
	constexpr int	UNIT = G > 1 ? 64 : BLOCK;
	const int	utid = G > 1 ? ( tid & 63 ) : tid;
	const int	ubase = G > 1 ? 0 : ( tid >> 6 ) * 64;

	for( unsigned i = tid; i < unsigned( prog_bytes ) / 4; i += BLOCK )
		reinterpret_cast<uint32_t *>( P )[ i ] = reinterpret_cast<const uint32_t *>( gP )[ i ];
	if( tid == 0 ){
		s_pool_n = 0;
		s_pool_head = 0;
	}
	__syncthreads();

	const int	T = db.tile_t;
	const int	w = P->w_winsize, lm = P->lmargin, rm = P->rmargin;
	rmd_lane_t	lane;

	// pre-filter set-up: first search element a proper helix (find_wchlx) or a 4-plex
	const rmd_elem_t	&e0 = P->elems[ P->searches[ 0 ] ];
	int	i_minl0 = e0.minilen;
	if( e0.type == RMA_T_Q1 )
		i_minl0 += P->elems[ e0.mates[ 0 ] ].minilen + P->elems[ e0.mates[ 1 ] ].minilen + 2 * e0.minlen;
	int	n_rank = ( e0.maxglen != RMA_UNBOUNDED && e0.maxglen < w ? e0.maxglen : w ) - e0.minglen + 1;
	const bool	quick = ( ( e0.type == RMA_T_H5 && e0.proper ) || e0.type == RMA_T_Q1 ) && n_rank < 0xffff;
	// helices that allow no mispair at all (find_motif.c:1010-1033 with mplim == 0) take
	// the bit-parallel pre-filter
	// first helix of a pseudoknot whose 5' strand starts at the start position
	const bool	pk0 = e0.type == RMA_T_H5 && !e0.proper && e0.scope == 0 && !( dbg & 4 ) &&
		e0.mplim <= 3 && e0.minlen >= 1;
	const bool	bitpar = ( ( quick && !( dbg & 4 ) ) || pk0 ) && e0.mplim <= 3 && e0.minlen >= 1 && e0.rows == 0;
	const unsigned	e0_mat2 = e0.pairset >= 0 ? rmd_pairsets( P )[ e0.pairset ].mat2 : 0;
	const bool	e0_at_szero = e0.type == RMA_T_P5 || e0.type == RMA_T_T1 || e0.type == RMA_T_Q1 ||
		( e0.type == RMA_T_H5 && ( e0.proper || e0.scope == 0 ) );
	const int	pb_words = ( tile_bytes + 63 ) / 64 + 3;
	// bit vectors of a slot: where the best literal occurs, then five pair rows per row set (the lean
	// instance keeps one set, the first element's; the general one a set per pair table its helices use)
	const int	n_rs = LEAN ? 1 : P->n_rowsets;
	const int	n_vec = 1 + 5 * n_rs;
	unsigned long long	*const pb0 = reinterpret_cast<unsigned long long *>( tile0 + size_t( G ) * slot_bytes );
	uint32_t	*lean_lo = reinterpret_cast<uint32_t *>( pb0 + size_t( G ) * n_vec * pb_words );
	uint16_t	*lean_hi = reinterpret_cast<uint16_t *>( lean_lo + P->n_searches * BLOCK );
	LdsRecs<BLOCK>	lr{ lean_lo + threadIdx.x, lean_hi + threadIdx.x };
	// (the general instance's records take the same place; behind them the resume states of the
	// levels up to the split level and the queue of continuations)
	const int	split_s = LEAN || ( dbg & 256 ) ? -1 : P->split_s;
	uint32_t	*const g_before = lean_lo + ( LEAN ? 0 : P->n_rec_dwords ) * BLOCK;
	uint32_t	*const g_deep = g_before + 2 * ( split_s + 1 ) * BLOCK;
	LdsGRecs<BLOCK>	gr{ lean_lo + threadIdx.x, g_before + threadIdx.x, P->rec_off };
	const bool	lit = P->lit_re >= 0 && !( dbg & 8 );
	const int	lit_n = lit ? rmd_regexes( P )[ P->lit_re ].n_states : 0;
	const int	lit_hi = lit ? ( P->lit_hi < w - lit_n ? P->lit_hi : w - lit_n ) : 0;
	const bool	split_ranks = !quick && lit && n_rank > 1 && n_rank < 0xffff;
	// TailAccel applies: symmetric pair table, tail helix on the same table without mispairs
	bool	tail_from_rows = false;
	if( LEAN && bitpar && e0.tail_s >= 0 ){
		const rmd_elem_t	&te = P->elems[ P->searches[ e0.tail_s ] ];
		bool	sym = true;
		for( int x = 0; x < 5; x++ )
			for( int y = 0; y < 5; y++ )
				sym = sym && ( ( ( e0_mat2 >> ( x * 5 + y ) ) ^ ( e0_mat2 >> ( y * 5 + x ) ) ) & 1 ) == 0;
		tail_from_rows = sym && te.pairset == e0.pairset && te.mplim == 0 && !te.pfrac && te.minlen >= 1 &&
			( te.ends & RMA_5PAIRED ) && ( te.ends & RMA_3PAIRED ) && te.maxglen != RMA_UNBOUNDED;
	}

	// the same for the first helix of the first element's interior (pooled instance)
	const bool	head_from_rows = POOL && bitpar && e0.head_s >= 0 && !( dbg & 4096 ) &&
		P->elems[ P->searches[ e0.head_s >= 0 ? e0.head_s : 0 ] ].rows == 0;

	// a slot belongs to one wave when G > 1: its phases are ordered within the wave (LDS
	// executes a wave's accesses in order), the waves need not march in step
#define SLOT_SYNC()	do{ \
		if constexpr( G > 1 ){ \
			__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" ); \
			__builtin_amdgcn_wave_barrier(); \
			__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" ); \
		}else \
			__syncthreads(); \
	}while( 0 )
	// Queue overflow (the queue is sized for the expected density; real sequence clusters) goes to
	// this workgroup's spill area in HBM and is popped after the LDS part -- 4 bytes out and in
	// per item.  The stores are plain (write-through L1, merged in L2; the barrier before pass B
	// orders them), the loads bypass L1 (agent scope: the area is reused tile after tile and L1
	// may hold the previous tile's lines).  Only what exceeds that too is searched in place by
	// the lane that found it.
	unsigned	*const spill = hb.spill + size_t( blockIdx.x ) * hb.spill_cap;
	const int	qtotal = qcap + hb.spill_cap;
	// diagnostic (RNAMOTIF_DBG bit 32): wave cycles per phase, summed over all waves, into the counters behind
	// the ticket: 0 ticket + decode, 1 literal vector, 2 pair rows, 3 pre-filter loop, 4 search, 5 waiting for the tile's end
	unsigned long long	t_ph = ( dbg & 32 ) ? __builtin_amdgcn_s_memtime() : 0;
#define PHASE( i_ )	do{ if( dbg & 32 ){ \
			const unsigned long long	now_ = __builtin_amdgcn_s_memtime(); \
			if( ( threadIdx.x & 63 ) == 0 ) \
				atomicAdd( hb.ticket + 3 + ( i_ ), now_ - t_ph ); \
			t_ph = now_; \
		} }while( 0 )
	const long long	n_units = G > 1 ? ( db.n_tiles + G - 1 ) / G : db.n_tiles;
	for( ; ; ){
		if( tid == 0 ){
			const long long	t = ( long long )atomicAdd( hb.ticket, 1ull );
			const int	s = G == 1 && t < db.n_tiles ? db.tile_seq[ t ] : 0;
			s_tile = t;
			s_seq = s;
			s_qn = 0;
			s_qhead = 0;
			s_dqn = 0;
			s_dqhead = 0;
		}
		__syncthreads();
		const long long	t = s_tile;
		bool	last = false;
		if( t >= n_units ){
			// (pooled: one more round, over a tile without start positions, for what the pool still holds)
			if constexpr( POOL )
				last = true;
			else
				break;
		}
		// what pass B needs of the tile (G > 1: of the last slot; pass B reloads per item)
		int	seq = 0, slen = 0, z0 = 0, p_lo = 0, vec_words = 0;
		uint8_t	*tile = tile0;
		unsigned long long	*pb = pb0 + pb_words;
		rmd_seq_t	sq{ tile0, 0 };
		DevSink	sink{ hb, 0, 0, P->hit_stride };
		const int	lane_id = tid & 63;
		const unsigned long long	lt_mask = ( 1ull << lane_id ) - 1;
		// (every wave makes the same number of rounds)
		for( int slot = G > 1 ? ( tid >> 6 ) : 0; slot < G; slot += G > 1 ? BLOCK / 64 : 1 ){
		const long long	tt = G > 1 ? t * G + slot : t;
		const bool	live = tt < db.n_tiles;
		const unsigned	slot_bits = G > 1 ? unsigned( slot ) << 26 : 0u;
		seq = G > 1 ? ( live ? db.tile_seq[ tt ] : 0 ) : s_seq;
		slen = db.slen[ seq ];
		tile = tile0 + size_t( slot ) * slot_bytes;
		unsigned long long	*const occ = pb0 + size_t( slot ) * n_vec * pb_words;	// where the best literal occurs (bit per start)
		pb = occ + pb_words;		// row set 0
		const int64_t	off = db.base_off[ seq ];
		const int	local = live ? int( tt - db.tile_start[ seq ] ) : 0;
		const int	per_strand = live ? int( ( db.tile_start[ seq + 1 ] - db.tile_start[ seq ] ) / db.strands ) : 1;
		const int	comp = local / per_strand;
		const int	pos_lo = db.pos_lo ? db.pos_lo[ seq ] : 0;
		const int	pos_hi = db.pos_hi ? db.pos_hi[ seq ] : 0x7fffffff;
		z0 = pos_lo + ( local % per_strand ) * T;

		// decode the bases this tile can touch: [ z0 - lm, z0 + T + w - 1 + rm )
		p_lo = z0 - lm;
		int	p_from = p_lo < 0 ? 0 : p_lo;
		int	p_to = z0 + T + w - 1 + rm;
		if( p_to > slen )
			p_to = slen;
		if( !live )
			p_to = p_from;		// slot past the last tile: nothing to decode, no start position
		// one packed word (16 bases) per lane and step; the reverse strand is the same words
		// read backwards and complemented (mk_rcmp, rnamot.c:193)
		if( p_from < p_to ){
			const int	f_lo = comp ? slen - p_to : p_from, f_hi = comp ? slen - 1 - p_from : p_to - 1;
			const int64_t	w_lo = ( off + f_lo ) >> 4, w_hi = ( off + f_hi ) >> 4;
			for( int64_t wi = w_lo + utid; wi <= w_hi; wi += UNIT ){
				const uint32_t	cw = db.codes[ wi ];
				const uint32_t	am = db.amask[ wi >> 1 ] >> ( ( wi & 1 ) * 16 );
				const int	f0 = int( ( wi << 4 ) - off );
				for( int k = 0; k < 16; k++ ){
					const int	f = f0 + k;
					if( f < f_lo || f > f_hi )
						continue;
					int	c = ( am >> k ) & 1 ? RMA_BC_N : int( ( cw >> ( 2 * k ) ) & 3 );
					if( comp && c < 4 )
						c = 3 - c;
					tile[ ( comp ? slen - 1 - f : f ) - p_lo ] = uint8_t( c );
				}
			}
		}
		SLOT_SYNC();
		PHASE( 0 );
		// short entries fill only part of a tile: the loops below run over what is there
		const int	pos_end = rmd_imin( slen - P->dminlen + 1, pos_hi );
		const int	n_pos = live ? rmd_imin( T, pos_end - z0 ) : 0;		// start positions of this tile
		vec_words = rmd_imin( pb_words, ( p_to - p_lo + 64 + 63 ) / 64 + 1 );	// bit vector words in use
		if constexpr( G > 1 ){
			if( utid == 0 ){
				int	*c = s_ctx[ slot ];
				c[ 0 ] = seq; c[ 1 ] = comp; c[ 2 ] = slen; c[ 3 ] = z0; c[ 4 ] = p_lo; c[ 5 ] = vec_words;
			}
		}

		// ---- pass A: pre-filter.  Where the first search element is a proper helix
		// or a 4-plex, almost every (start, end) pair dies at its first base pairs
		// (find_motif.c:1010-1021); test that here in registers, no search state,
		// and compact the survivors into the LDS work queue with one wave ballot +
		// prefix count per step.  Other first elements queue the whole position.
		sq = rmd_seq_t{ tile, p_lo };
		sink.seq = seq;
		sink.comp = comp;
		// Best-literal filter (the reference's -O skip scan, find_motif.c:209-243, as a
		// necessary condition): occ has a bit for every tile position where the literal
		// starts; a start position is searched only if one lies at an allowed offset.
		if( lit ){
			const rmd_regex_t	&lre = rmd_regexes( P )[ P->lit_re ];
			const int	n_valid = p_to - p_lo;
			for( int base = ubase; base < vec_words * 64; base += UNIT ){
				const int	q = base + lane_id - 64;
				bool	ok = q >= p_from - p_lo && q + lit_n <= n_valid;
				for( int jj = 0; ok && jj < lit_n; jj++ )
					ok = ( lre.accept[ tile[ q + jj ] ] >> jj ) & 1;
				const unsigned long long	m = __ballot( ok );
				if( lane_id == 0 )
					occ[ base >> 6 ] = m;
			}
			SLOT_SYNC();
			PHASE( 1 );
		}
		// does the literal start anywhere in [a_, b_] (absolute positions)?
#define LIT_IN( a_, b_, res_ )	do{ \
		res_ = true; \
		if( lit ){ \
			res_ = false; \
			const int	b1_ = ( b_ ) - p_lo + 64; \
			for( int b0_ = ( a_ ) - p_lo + 64; b0_ <= b1_ && !res_; b0_ += 64 ){ \
				const int	wi_ = b0_ >> 6, sh_ = b0_ & 63; \
				unsigned long long	x_ = sh_ ? ( occ[ wi_ ] >> sh_ ) | ( occ[ wi_ + 1 ] << ( 64 - sh_ ) ) : occ[ wi_ ]; \
				const int	nb_ = b1_ - b0_ + 1; \
				if( nb_ < 64 ) \
					x_ &= ( 1ull << nb_ ) - 1; \
				res_ = x_ != 0; \
			} \
		} }while( 0 )
#define LIT_OK( szero_, res_ )	LIT_IN( ( szero_ ) + P->lit_lo, ( szero_ ) + lit_hi, res_ )

		// push( pred, item ): one ballot + prefix count per call; overflow is searched in place
#define QPUSH( pred, item, szero_, r0_, cnt_ )	do{ \
		const unsigned long long	m_ = __ballot( pred ); \
		if( m_ ){ \
			int	base_ = 0; \
			if( lane_id == __ffsll( m_ ) - 1 ) \
				base_ = atomicAdd( &s_qn, __popcll( m_ ) ); \
			base_ = __shfl( base_, __ffsll( m_ ) - 1 ); \
			if( pred ){ \
				const int	slot_ = base_ + __popcll( m_ & lt_mask ); \
				if( slot_ < qcap ) \
					queue[ slot_ ] = ( item ) | slot_bits; \
				else if( slot_ < qtotal ) \
					spill[ slot_ - qcap ] = ( item ) | slot_bits; \
				else if constexpr( LEAN ){ \
					rmd_lean_t	st_; \
					int	k_ = rmd_lean_begin( P, lr, st_, szero_, slen, r0_, cnt_ ); \
					while( k_ >= 0 ) \
						k_ = rmd_lean_step( P, lr, st_, sq, k_, &lane, sink ); \
				} \
				/* (general instance: the launch is repeated with a spill area that holds the tile's items) */ \
			} \
		} }while( 0 )

		// Pair rows: for every row set, rows[ b ] has a bit per tile position that can pair with 5'
		// base b (one ballot per 64 positions and row), so "the first minlen pairs of (start, end)
		// hold" is an AND of minlen shifted 64-bit windows, 64 end positions at a time -- for the
		// pre-filter below and for the helices of the search itself (RowEnds).
		if( LEAN ? bitpar : n_rs > 0 ){
			const int	n_valid = p_to - p_lo;
			for( int rs = 0; rs < n_rs; rs++ ){
				const unsigned	mat2 = rmd_pairsets( P )[ P->rowset_ps[ rs ] ].mat2;
				unsigned long long	*const rows = pb + 5 * rs * pb_words;
				for( int base = ubase; base < vec_words * 64; base += UNIT ){
					const int	q = base + lane_id - 64;		// one pad word in front
					const int	code = ( q >= p_from - p_lo && q < n_valid ) ? tile[ q ] : 7;
					for( int b5 = 0; b5 < 5; b5++ ){
						const unsigned long long	m = __ballot( code < 5 && ( ( mat2 >> ( b5 * 5 + code ) ) & 1 ) );
						if( lane_id == b5 )
							rows[ b5 * pb_words + ( base >> 6 ) ] = m;
					}
				}
			}
			SLOT_SYNC();
			PHASE( 2 );
		}
		if( bitpar ){
			const int	hl0 = e0.minlen;
			// superset of match_wchlx's rule (find_motif.c:1010-1033,1065-1080): at most mplim
			// mispairs among the first minlen pairs; an unpaired first pair is allowed only if
			// the 5' end may be unpaired (then it is the one mispair that is always tolerated)
			const int	lim = ( e0.ends & RMA_5PAIRED ) ? e0.mplim : ( e0.mplim > 1 ? e0.mplim : 1 );
			// end positions top-r0-63 .. top-r0 of a helix starting at szero: bit i set when
			// end position top-r0-63+i passes; positions below lo are cleared
			const bool	ends5 = ( e0.ends & RMA_5PAIRED ) != 0;
			auto	win = [ & ]( int szero, int top, int r0, int lo ) -> unsigned long long {
				return rows_win( pb, pb_words, tile, p_lo, hl0, lim, ends5, szero, top - r0 - 63, lo );
			};
			int	n_wait = 0;		// pseudoknot pre-filter: start positions of this wave that wait in cbuf
			for( int j = 0; j < n_pos; j += UNIT ){
				const int	rel0 = j + utid;
				bool	valid0 = rel0 < T && z0 + rel0 <= slen - P->dminlen && z0 + rel0 < pos_hi;
				if( valid0 )
					LIT_OK( z0 + rel0, valid0 );
				if constexpr( !LEAN ){
				if( pk0 ){
					// The literal and the anchored prefix of the first helix leave one start position in
					// sixteen (pk1.descr: "^tg", "gaaa"): the row windows below would run for four lanes of
					// a wave.  Survivors are collected, per wave, and taken 64 at a time.
					__shared__ uint16_t	s_cbuf[ BLOCK / 64 ][ 128 ];
					uint16_t	*const cbuf = s_cbuf[ tid >> 6 ];
					valid0 = valid0 && rmd_prefix_ok( P, e0, sq, z0 + rel0 );
					const unsigned long long	mv = __ballot( valid0 );
					if( valid0 )
						cbuf[ n_wait + __popcll( mv & lt_mask ) ] = uint16_t( rel0 );
					n_wait += __popcll( mv );
					const bool	last_j = j + UNIT >= n_pos;
					while( n_wait >= 64 || ( last_j && n_wait > 0 ) ){
						__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
						__builtin_amdgcn_wave_barrier();
						__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
						const int	n_take = n_wait < 64 ? n_wait : 64;
						const bool	valid = lane_id < n_take;
						const int	rel = valid ? int( cbuf[ lane_id ] ) : 0;
						const int	rest = n_wait - n_take;
						const int	moved = lane_id < rest ? int( cbuf[ n_take + lane_id ] ) : 0;
						__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
						__builtin_amdgcn_wave_barrier();
						__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
						if( lane_id < rest )
							cbuf[ lane_id ] = uint16_t( moved );
						n_wait = rest;
						const int	szero = z0 + rel;
						int	hi = 0, lo = 1;
						if( valid )
							rmd_level0_range( P, szero, slen, &hi, &lo );
						do{
							// first helix of a pseudoknot (find_pknot5/find_pknot3 :495-640): its 3' end
							// lies between s5 + 2*minlen + interior - 1 and the window end minus what must
							// follow it; search the position only if some end there can start the helix
							bool	any = false;
							unsigned long long	W0 = 0;		// 3' ends top-63 .. top that can, when the whole range is one word
							const int	top = hi - e0.q_sminl, bot = szero + 2 * e0.minlen + e0.q_iminl - 1;
							const bool	one_word = top - bot < 64 && e0.q_smaxl >= 0;
							if( valid ){
								if( one_word ){
									W0 = top >= bot ? win( szero, top, 0, bot ) : 0;
									any = W0 != 0;
								}else
									for( int r0 = 0; !any && r0 <= top - bot; r0 += 64 )
										any = win( szero, top, r0, bot ) != 0;
							}
							if( __ballot( any ) == 0 )
								continue;		// (nothing to queue for these 64 positions)
							if( split_ranks ){
								// Lanes whose end range is one word: "rank r leaves the 3' strand the ends
								// hi-r-q_smaxl .. hi-r-q_sminl" (find_pknot3 :566-568) = some bit of W0 among
								// 63-r-(q_smaxl-q_sminl) .. 63-r = bit 63-r of W0 smeared upwards over that many
								// places; the ranks to queue are the bits of one word, taken one per round -- a few
								// rounds instead of one per rank of the window.
								unsigned long long	Rm = 0;
								if( any && one_word ){
									unsigned long long	D = W0;
									for( int have = 1, need = e0.q_smaxl - e0.q_sminl + 1; have < need; ){
										const int	st = rmd_imin( have, need - have );
										D |= D << st;
										have += st;
									}
									const int	rmax = rmd_imin( rmd_imin( n_rank - 1, hi - lo ), 63 );
									Rm = rmax >= 0 ? D & ( ~0ull << ( 63 - rmax ) ) : 0ull;
								}
								while( __ballot( Rm != 0 ) ){
									bool	pred = Rm != 0;
									const int	i = pred ? 63 - __builtin_clzll( Rm ) : 0;
									const int	r = 63 - i;
									if( pred )
										Rm &= ~( 1ull << i );
									if( pred && P->lit_ehi >= 0 ){
										const int	e_ = hi - r;
										const int	a_ = rmd_imax( szero + P->lit_lo, e_ - P->lit_ehi );
										const int	b_ = rmd_imin( szero + lit_hi, e_ - P->lit_elo );
										if( a_ > b_ )
											pred = false;
										else
											LIT_IN( a_, b_, pred );
									}
									if( __ballot( pred ) == 0 )
										continue;
									QPUSH( pred, ( unsigned( rel ) << 16 ) | unsigned( r ), szero, r, 1 );
								}
								// (end ranges wider than a word: rank by rank)
								const bool	slow = any && !one_word;
								if( __ballot( slow ) )
								for( int r = 0; r < n_rank; r++ ){
									bool	pred = slow && r <= hi - lo;
									if( pred && one_word ){
										// this rank's end leaves the 3' strand the ends hi-r-q_smaxl .. hi-r-q_sminl
										// (find_pknot3 :566-568): bits 63-r-(q_smaxl-q_sminl) .. 63-r of W0
										const int	i_hi = 63 - r, i_lo = rmd_imax( i_hi - ( e0.q_smaxl - e0.q_sminl ), 0 );
										pred = i_hi >= 0 && ( ( W0 >> i_lo ) & ( i_hi - i_lo >= 63 ? ~0ull : ( 2ull << ( i_hi - i_lo ) ) - 1 ) ) != 0;
									}
									if( pred && P->lit_ehi >= 0 ){
										// the rank fixes the end of the knot: the literal must also sit at
										// an admissible distance from that end
										const int	e_ = hi - r;
										const int	a_ = rmd_imax( szero + P->lit_lo, e_ - P->lit_ehi );
										const int	b_ = rmd_imin( szero + lit_hi, e_ - P->lit_elo );
										if( a_ > b_ )
											pred = false;
										else
											LIT_IN( a_, b_, pred );
									}
									if( __ballot( pred ) == 0 )
										continue;
									QPUSH( pred, ( unsigned( rel ) << 16 ) | unsigned( r ), szero, r, 1 );
								}
							}else
								QPUSH( any, ( unsigned( rel ) << 16 ) | 0xffffu, szero, 0, RMD_ALL_RANKS );
						}while( 0 );
					}
					continue;
				}
				}
				const int	rel = rel0;
				const int	szero = z0 + rel;
				const bool	valid = valid0;
				int	hi = 0, lo = 1;
				if( valid )
					rmd_level0_range( P, szero, slen, &hi, &lo );
				for( int r0 = 0; r0 < n_rank; r0 += 64 ){
					unsigned long long	W = 0;
					if( valid && r0 <= hi - lo )
						W = win( szero, hi, r0, lo );
					while( __ballot( W != 0 ) ){
						const bool	has = W != 0;
						const int	i = has ? __ffsll( W ) - 1 : 0;
						const int	r = r0 + 63 - i;
						// (the pinned tail helix of the interior, rmd_tail_ok(), is left to pass B:
						// tested here it costs more in this divergent loop than it saves there)
						QPUSH( has, ( unsigned( rel ) << 16 ) | unsigned( r ), szero, r, 1 );
						W &= W - 1;
					}
				}
			}
		}else
		for( int j = 0; j < n_pos; j += UNIT ){
			const int	rel = j + utid;
			const int	szero = z0 + rel;
			bool	valid = rel < T && szero <= slen - P->dminlen && szero < pos_hi;
			if( valid )
				LIT_OK( szero, valid );
			int	hi = 0, lo = 1;
			if( valid && quick )
				rmd_level0_range( P, szero, slen, &hi, &lo );
			if( valid && split_ranks )
				rmd_level0_range( P, szero, slen, &hi, &lo );
			const int	steps = ( quick || split_ranks ) ? n_rank : 1;
			for( int r = 0; r < steps; r++ ){
				if( split_ranks ){
					// few start positions survive the filters: hand their end positions
					// out one by one so that the lanes of pass B all get work
					bool	pred = valid && r <= hi - lo && ( r > 0 || !e0_at_szero || rmd_prefix_ok( P, e0, sq, szero ) );
					if( r == 0 && valid && !pred )
						valid = false;
					if( pred && P->lit_ehi >= 0 ){
						// this rank fixes the end of the first element's group: the literal must
						// also sit at an admissible distance from that end
						const int	e_ = hi - r;
						const int	a_ = rmd_imax( szero + P->lit_lo, e_ - P->lit_ehi );
						const int	b_ = rmd_imin( szero + lit_hi, e_ - P->lit_elo );
						if( a_ > b_ )
							pred = false;
						else
							LIT_IN( a_, b_, pred );
					}
					QPUSH( pred, ( unsigned( rel ) << 16 ) | unsigned( r ), szero, r, 1 );
				}else if( quick ){
					const int	sd = hi - r;
					const bool	pred = valid && sd >= lo &&
						rmd_quick_wchlx( P, e0, sq, szero, sd, rmd_s3lim( szero, sd, i_minl0, e0.maxlen ) );
					QPUSH( pred, ( unsigned( rel ) << 16 ) | unsigned( r ), szero, r, 1 );
				}else{
					// helices whose 5' strand starts at the start position: its anchored
					// seq= prefix must be there
					const bool	pred = valid && ( !e0_at_szero || rmd_prefix_ok( P, e0, sq, szero ) );
					QPUSH( pred, ( unsigned( rel ) << 16 ) | 0xffffu, szero, 0, RMD_ALL_RANKS );
				}
			}
		}
#undef QPUSH
#undef LIT_OK
#undef LIT_IN
		PHASE( 3 );
		}	// slots
#undef SLOT_SYNC
		__syncthreads();
		PHASE( 5 );

		// ---- pass B: the full search.  Lanes are persistent within the tile: a lane
		// that finishes its item pops the next one at once (wave-aggregated pop), so a
		// wave lasts as long as its share of the work, not as its slowest item times
		// the number of rounds.
		if constexpr( !LEAN ){
			// more items than queue and spill area hold: none is searched twice or dropped silently --
			// the host repeats the launch with an area of the size asked for here
			if( tid == 0 && s_qn > qtotal )
				atomicMax( hb.ticket + 2, ( unsigned long long )s_qn );
		}
		const int	nq = ( dbg & 1 ) ? 0 : ( s_qn < qtotal ? s_qn : qtotal );
		if( ( dbg & 2 ) && tid == 0 )
			atomicAdd( hb.ticket + 1, ( unsigned long long )s_qn );
		int	k = -1;
		bool	dry = false;
		if constexpr( POOL ){
			// ---- pass A': every queued item takes the tail test (for every helix length its end
			// position allows: is the pinned helix that closes the interior there?  nine in ten
			// are not), all lanes busy; what passes goes to the pool with its entry and strand
			unsigned	*const pool = hb.pool + size_t( blockIdx.x ) * hb.pool_cap * 3;
			{
				TailAccel	ac{ P, pb, tile, pb_words, p_lo, vec_words * 64, 0 };
				for( int c = 0; c < nq; c += BLOCK ){
					const int	i = c + tid;
					unsigned	item = 0;
					bool	keep = false;
					if( i < nq ){
						item = i < qcap ? queue[ i ] : __hip_atomic_load( spill + ( i - qcap ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
						const int	r = int( item & 0xffffu );
						keep = true;
						if( ( tail_from_rows || head_from_rows ) && r != 0xffff ){
							const int	szero = z0 + int( item >> 16 );
							int	hi, lo;
							rmd_level0_range( P, szero, slen, &hi, &lo );
							const int	span = hi - r - szero + 1;
							keep = false;
							for( int hl = e0.minlen; hl <= e0.maxlen && !keep; hl++ ){
								const int	ilen = span - 2 * hl;
								if( ilen < e0.minilen )
									break;
								if( ilen > e0.maxilen )
									continue;
								bool	ok = true, t;
								if( tail_from_rows && ac.tail( e0, szero, hl, span - 1 - hl, &t ) )
									ok = t;
								if( ok && head_from_rows ){
									// the first helix of the interior, 5' strand at one of a few offsets from the
									// interior's start: can any 3' end start it? (rows_win: its first minlen pairs)
									const rmd_elem_t	&H = P->elems[ P->searches[ e0.head_s ] ];
									const int	lim = ( H.ends & RMA_5PAIRED ) ? H.mplim : ( H.mplim > 1 ? H.mplim : 1 );
									const int	b = szero + span - 1 - hl;		// last position of the interior
									ok = false;
									for( int pre = e0.head_pre_min; pre <= e0.head_pre_max && !ok; pre++ ){
										const int	s5 = szero + hl + pre;
										const int	top = rmd_imin( s5 + H.maxglen - 1, b ), bot = s5 + H.minglen - 1;
										if( top < bot )
											continue;
										const int	w0 = top - 63, q_hi = w0 - p_lo + 64;
										// (undecided -- a range wider than a word, bits the vectors do not hold: kept)
										if( top - bot >= 64 || q_hi - H.minlen < 0 || q_hi + 96 > vec_words * 64 )
											ok = true;
										else
											ok = rows_win( pb, pb_words, tile, p_lo, H.minlen, lim, ( H.ends & RMA_5PAIRED ) != 0, s5, w0, bot ) != 0;
									}
								}
								keep = ok;
							}
						}
					}
					const unsigned long long	m = __ballot( keep );
					if( m ){
						int	base = 0;
						if( lane_id == __ffsll( m ) - 1 )
							base = atomicAdd( &s_pool_n, __popcll( m ) );
						base = __shfl( base, __ffsll( m ) - 1 );
						if( keep ){
							// (at most pool_min - 1 items wait when a tile starts and a tile queues at most
							// qtotal: the pool holds pool_min + qtotal)
							unsigned	*e = pool + 3 * size_t( base + __popcll( m & lt_mask ) );
							e[ 0 ] = unsigned( seq );
							e[ 1 ] = unsigned( z0 + int( item >> 16 ) );
							e[ 2 ] = ( item & 0xffffu ) | ( unsigned( sink.comp ) << 16 );
						}
					}
				}
			}
			PHASE( 3 );
			__syncthreads();
			const int	n_pool = s_pool_n;
			if( n_pool > 0 && ( last || n_pool >= hb.pool_min ) && ( dbg & 2048 ) ){
				// (diagnostic: the pool is filled and thrown away)
				__syncthreads();
				if( tid == 0 )
					s_pool_n = 0;
			}else if( n_pool > 0 && ( last || n_pool >= hb.pool_min ) ){
				// ---- pass B over the pool
				constexpr int	NIB_MAX = 32;		// window dwords per lane the host has checked room for
				uint32_t	*const col = reinterpret_cast<uint32_t *>( tile0 ) + tid;
				rmd_nibseq_t<BLOCK>	nsq{ col, 0, 0 };
				rmd_lean_t	st;
				const rmd_no_accel_t	none;
				for( ; ; ){
					const unsigned long long	want = __ballot( k < 0 && !dry );
					const unsigned long long	busy = __ballot( k >= 0 );
					// idle lanes pop together: a window costs some hundred instructions to rebuild, and
					// a round for one lane costs the wave as much as a round for sixteen
					if( want && ( busy == 0 || __popcll( want ) >= hb.pool_refill ) ){
						if( ( dbg & 32 ) && lane_id == 0 ){
							atomicAdd( hb.ticket + 15, 1ull );
							atomicAdd( hb.ticket + 16, ( unsigned long long )__popcll( want ) );
						}
						int	base = 0;
						if( lane_id == __ffsll( want ) - 1 )
							base = atomicAdd( &s_pool_head, __popcll( want ) );
						base = __shfl( base, __ffsll( want ) - 1 );
						if( k < 0 && !dry ){
							const int	i = base + __popcll( want & lt_mask );
							if( i < n_pool ){
								const unsigned	*e = pool + 3 * size_t( i );
								const int	iseq = int( __hip_atomic_load( e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) );
								const int	szero = int( __hip_atomic_load( e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) );
								const unsigned	rc = __hip_atomic_load( e + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
								const int	r = int( rc & 0xffffu ), icomp = int( rc >> 16 ) & 1;
								const int	islen = db.slen[ iseq ];
								const int64_t	off = db.base_off[ iseq ];
								// the window's bases [ p0, p1 ) of the strand = [ f_lo, f_hi ) of the entry as stored
								const int	p0 = rmd_imax( szero - lm, 0 ), p1 = rmd_imin( szero + w + rm, islen );
								const int	f_lo = icomp ? islen - p1 : p0, f_hi = icomp ? islen - p0 : p1;
								const int64_t	g0 = ( off + f_lo ) & ~int64_t( 7 );
								const int	n_dw = int( ( off + f_hi - g0 + 7 ) >> 3 );
								for( int j = 0; j < n_dw && j < NIB_MAX; j++ ){
									const int64_t	g = g0 + 8 * j;		// eight bases: half a word of codes, a byte of the mask
									const uint32_t	cw = ( db.codes[ g >> 4 ] >> ( ( g & 8 ) * 2 ) ) & 0xffffu;
									const uint32_t	am = ( db.amask[ g >> 5 ] >> ( g & 24 ) ) & 0xffu;
									uint32_t	x = ( cw | ( cw << 8 ) ) & 0x00ff00ffu;
									x = ( x | ( x << 4 ) ) & 0x0f0f0f0fu;
									x = ( x | ( x << 2 ) ) & 0x33333333u;
									if( icomp )
										x ^= 0x33333333u;		// (mk_rcmp, rnamot.c:193: 3 - code)
									uint32_t	n = ( am | ( am << 12 ) ) & 0x000f000fu;
									n = ( n | ( n << 6 ) ) & 0x03030303u;
									n = ( n | ( n << 3 ) ) & 0x11111111u;
									col[ j * BLOCK ] = ( x & ~( n * 3u ) ) | ( n << 2 );	// RMA_BC_N = 4
								}
								nsq.flip = icomp ? -1 : 0;
								nsq.bias = int( off - g0 ) + ( icomp ? islen : 0 );
								sink.seq = iseq;
								sink.comp = icomp;
								k = rmd_lean_begin( P, lr, st, szero, islen, r == 0xffff ? 0 : r, r == 0xffff ? RMD_ALL_RANKS : 1 );
							}else
								dry = true;
						}
						continue;
					}
					if( busy == 0 )
						break;
					if( ( dbg & 32 ) && lane_id == 0 ){
						atomicAdd( hb.ticket + 17, 1ull );
						atomicAdd( hb.ticket + 18, ( unsigned long long )__popcll( busy ) );
					}
					if( k >= 0 )
						k = rmd_lean_step( P, lr, st, nsq, k, &lane, sink, none );
				}
				__syncthreads();
				if( tid == 0 ){
					s_pool_n = 0;
					s_pool_head = 0;
				}
			}
			if( last ){
				PHASE( 4 );
				break;
			}
		}else if constexpr( LEAN ){

			// ss / proper-helix descriptors: 8 bytes of search state per level, in LDS
			rmd_lean_t	st;
			// the pre-filter's rows serve the tail test of level 0 when the tail helix pairs by
			// the same (symmetric) table, allows no mispair and has both ends paired
			TailAccel	accel{ P, pb, tile, pb_words, p_lo, vec_words * 64, tail_from_rows ? 0 : -1 };
			for( ; ; ){
				// lanes without work pop until they hold an item that survives the tail test
				unsigned long long	t_b0 = ( dbg & 32 ) ? __builtin_amdgcn_s_memtime() : 0;
				for( unsigned long long want; ( want = __ballot( k < 0 && !dry ) ) != 0; ){
					if( ( dbg & 32 ) && lane_id == 0 ){
						atomicAdd( hb.ticket + 15, 1ull );
						atomicAdd( hb.ticket + 16, ( unsigned long long )__popcll( want ) );
					}
					int	base = 0;
					if( lane_id == __ffsll( want ) - 1 )
						base = atomicAdd( &s_qhead, __popcll( want ) );
					base = __shfl( base, __ffsll( want ) - 1 );
					if( k < 0 && !dry ){
						const int	i = base + __popcll( want & lt_mask );
						if( i < nq ){
							const unsigned	item = i < qcap ? queue[ i ] :
								__hip_atomic_load( spill + ( i - qcap ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
							const int	r = int( item & 0xffffu );
							if constexpr( G > 1 ){
								// the item's own tile: its slot of LDS, its entry and strand
								const int	sl = int( item >> 26 );
								const int	*c = s_ctx[ sl ];
								slen = c[ 2 ];
								z0 = c[ 3 ];
								p_lo = c[ 4 ];
								tile = tile0 + size_t( sl ) * slot_bytes;
								pb = pb0 + ( size_t( sl ) * n_vec + 1 ) * pb_words;
								sq = rmd_seq_t{ tile, p_lo };
								sink.seq = c[ 0 ];
								sink.comp = c[ 1 ];
								accel.pb = pb;
								accel.tile = tile;
								accel.p_lo = p_lo;
								accel.vec_bits = c[ 5 ] * 64;
							}
							const int	szero = z0 + int( ( item >> 16 ) & ( G > 1 ? 0x3ffu : 0xffffu ) );
							bool	drop = false;
							if( accel.usable_for >= 0 && r != 0xffff ){
								// the tail test for every helix length the end position allows,
								// before any search state is set up: nine items in ten end here
								int	hi, lo;
								rmd_level0_range( P, szero, slen, &hi, &lo );
								const int	span = hi - r - szero + 1;
								drop = true;
								for( int hl = e0.minlen; hl <= e0.maxlen && drop; hl++ ){
									const int	ilen = span - 2 * hl;
									if( ilen < e0.minilen )
										break;
									bool	ok;
									if( ilen <= e0.maxilen && ( !accel.tail( e0, szero, hl, span - 1 - hl, &ok ) || ok ) )
										drop = false;
								}
							}
							if( !drop )
								k = rmd_lean_begin( P, lr, st, szero, slen,
									r == 0xffff ? 0 : r, r == 0xffff ? RMD_ALL_RANKS : 1 );
						}else
							dry = true;
					}
				}
				const unsigned long long	busy = __ballot( k >= 0 );
				unsigned long long	t_b1 = 0;
				if( dbg & 32 ){
					t_b1 = __builtin_amdgcn_s_memtime();
					if( lane_id == 0 ){
						atomicAdd( hb.ticket + 19, t_b1 - t_b0 );
						atomicAdd( hb.ticket + 17, 1ull );
						atomicAdd( hb.ticket + 18, ( unsigned long long )__popcll( busy ) );
					}
				}
				if( busy == 0 )
					break;
				if( k >= 0 )
					k = rmd_lean_step( P, lr, st, sq, k, &lane, sink, accel );
				if( ( dbg & 32 ) && lane_id == 0 )
					atomicAdd( hb.ticket + 20, __builtin_amdgcn_s_memtime() - t_b1 );
			}
		}else{
			GenTile	gt{ P, lean_lo, g_before, g_deep, queue, spill, qcap, nq, &s_qhead, &s_dqn, &s_dqhead,
				pb, tile, pb_words, p_lo, vec_words, slen, z0, split_s, dbg };
			general_pass_b<BLOCK, KINDS>( gt, sink );
		}
		PHASE( 4 );
		__syncthreads();
		PHASE( 5 );
	}
#undef PHASE
}

// ---------------------------------------------------------------- efn kernel
struct DevSeq {
	DbView	db;
	int64_t	off;
	int	slen, comp;
	__device__ inline int code( int p ) const { return db_strand_code( db, off, slen, comp, p ); }
};

#define RME_N16_PAD	( ( RME_N16 + 7 ) / 8 * 8 )
// One workgroup of 256 lanes per CU: efn's tables (60.7 KB as int16) are staged once per
// workgroup, every lane keeps the base codes and partners of its call in LDS while it is no
// longer than EFN_CACHE bases (a cloverleaf is under 96) -- 136 KB in all -- and the candidates
// are taken in a grid-stride loop, so the staging is paid once per CU and four waves share it.
#define EFN_BLOCK	256
#define EFN_CACHE	96
template< int BLOCK >
__global__ void __launch_bounds__( BLOCK )
rma_efn_kernel( const rmd_program_t *gP, DbView db, int32_t *hits, long long n_hits,
	const int16_t *g16, const int32_t *tlkey, const int32_t *loginc, const rma_efn2data_t *e2 )
{
	// tables staged 16 bytes per lane and step (the device copy is padded to a multiple of 8 entries)
	__shared__ __align__( 16 ) int16_t	t16[ RME_N16_PAD ];
	if( g16 != nullptr )
		for( int i = threadIdx.x; i < RME_N16_PAD / 8; i += BLOCK )
			reinterpret_cast<uint4 *>( t16 )[ i ] = reinterpret_cast<const uint4 *>( g16 )[ i ];
	__syncthreads();
	// per lane: base codes and partners of the call, when it is short enough
	__shared__ int16_t	s_bp[ BLOCK ][ EFN_CACHE + 1 ];
	__shared__ uint8_t	s_bc[ BLOCK ][ EFN_CACHE + 4 ];
	rme_tables_t	T{ t16, tlkey, loginc };
	int16_t	*bpbuf = s_bp[ threadIdx.x ];
	uint8_t	*bcbuf = s_bc[ threadIdx.x ];
	const int	efn_off = RMA_HIT_HDR + 4 * gP->n_elems + 4;
	for( long long h = ( long long )blockIdx.x * BLOCK + threadIdx.x; h < n_hits; h += ( long long )gridDim.x * BLOCK ){
		int32_t	*w = hits + h * gP->hit_stride;
		DevSeq	sq{ db, db.base_off[ w[ 0 ] ], db.slen[ w[ 0 ] ], w[ 1 ] };
		for( int k = 0; k < gP->n_efn; k++ ){
			if( gP->efn_sites[ k ].kind == RMA_EFN_KIND_EFN2 )
				w[ efn_off + k ] = e2 != nullptr ? rme2_site_energy( gP, e2, &sq, w, k, bpbuf, bcbuf, EFN_CACHE ) : RME2_INF;
			else if( g16 != nullptr )
				w[ efn_off + k ] = rme_site_energy( gP, &T, &sq, w, k, bpbuf, bcbuf, EFN_CACHE );
		}
	}
}

// ---------------------------------------------------------------- host side
#define HIPCHK( call )	do{ hipError_t e_ = ( call ); if( e_ != hipSuccess ){ \
		snprintf( err, errlen, "%s: %s", #call, hipGetErrorString( e_ ) ); return 1; } }while( 0 )

struct rma_scanner {
	rma_program_t	prog;
	rmd_program_t	dprog;
	int	device = 0;
	hipStream_t	stream = nullptr;
	hipEvent_t	ev[ 4 ] = { nullptr, nullptr, nullptr, nullptr };
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
	unsigned	*d_pool = nullptr;		// [grid_blocks][pool_cap][3] pooled instance: items waiting for pass B
	int	pool_cap = 0, pool_min = 1024;
	int32_t	*h_raw = nullptr;		// pinned
	size_t	h_raw_cap = 0;
	std::vector<int32_t>	h_sorted;
	std::vector<rma::HitKey>	keys, keys_tmp;
	rma::DevHitSort	dsort;		// ordering on the device (rm_hitsort_dev.h)
	unsigned long long	*h_ctr = nullptr;	// pinned: the counters a launch leaves
	int	tile_t = 2048;
	int	grid_blocks = 0;
};

struct rma_db {
	rma_scanner	*sc;
	int	device = 0;		// (kept here: a database may outlive its scanner)
	uint32_t	*d_codes = nullptr, *d_amask = nullptr;
	int64_t	*d_base_off = nullptr, *d_tile_start = nullptr;
	int32_t	*d_tile_seq = nullptr;
	int	tile_t = 0, qcap = 0, group = 1;	// launch shape of this database (see db_upload)
	int32_t	*d_slen = nullptr, *d_pos_lo = nullptr, *d_pos_hi = nullptr;
	int32_t	n_seq = 0, max_slen = 0;
	int64_t	n_tiles = 0, total_bases = 0;
	int	strands = 2;
};

extern "C" void rma_db_destroy( rma_db_t *db );
extern "C" void rma_scanner_destroy( rma_scanner_t *sc );

static void build_tables16( const rma_efndata_t *ed, std::vector<int16_t> &t16, std::vector<int32_t> &tlkey )
{
	t16.assign( ( RME_N16 + 7 ) / 8 * 8, 0 );	// padded for 16-byte staging loads
	tlkey.assign( 100, -1 );
	auto put = [&]( int off, const int32_t *src, int n ){
		for( int i = 0; i < n; i++ ){
			int	v = src[ i ];
			t16[ off + i ] = int16_t( v > 32767 ? 32767 : v < -32768 ? -32768 : v );
		}
	};
	put( RME_INTER, ed->inter, 31 );
	put( RME_BULGE, ed->bulge, 31 );
	put( RME_HAIRPIN, ed->hairpin, 31 );
	put( RME_DANGLE, &ed->dangle[ 0 ][ 0 ][ 0 ][ 0 ], 250 );
	put( RME_POPPEN, ed->poppen, 5 );
	put( RME_EPARAM, ed->eparam, 16 );
	int32_t	misc[ 9 ] = { ed->maxpen, ed->auend, ed->gubonus, ed->cslope, ed->cint, ed->c3, ed->gail,
		ed->ntriloops, ed->ntloops };
	put( RME_MISC, misc, 9 );
	for( int k = 0; k < 50; k++ ){
		// a key that does not fit 15 bits can never equal a computed key's low part
		// by accident: store -1 (no computed key is negative)
		int	key = k < ed->ntriloops ? ed->triloops[ k ][ 0 ] : -1;
		t16[ RME_TRIKEY + k ] = int16_t( key >= 0 && key <= 32767 ? key : -1 );
		t16[ RME_TRIVAL + k ] = int16_t( k < ed->ntriloops ? ed->triloops[ k ][ 1 ] : 0 );
	}
	for( int k = 0; k < 100; k++ ){
		tlkey[ k ] = k < ed->ntloops ? ed->tloops[ k ][ 0 ] : -1;
		t16[ RME_TLVAL + k ] = int16_t( k < ed->ntloops ? ed->tloops[ k ][ 1 ] : 0 );
	}
	put( RME_STACK, &ed->stack[ 0 ][ 0 ][ 0 ][ 0 ], 625 );
	put( RME_TSTKH, &ed->tstkh[ 0 ][ 0 ][ 0 ][ 0 ], 625 );
	put( RME_TSTKI, &ed->tstki[ 0 ][ 0 ][ 0 ][ 0 ], 625 );
	put( RME_SINT2, &ed->sint2[ 0 ][ 0 ][ 0 ][ 0 ], 900 );
	put( RME_ASINT, &ed->asint1x2[ 0 ][ 0 ][ 0 ][ 0 ][ 0 ], 4500 );
	put( RME_SINT4, &ed->sint4[ 0 ][ 0 ][ 0 ][ 0 ][ 0 ][ 0 ], 22500 );
}

extern "C" int rma_device_count( void )
{
	int	n = 0;
	if( hipGetDeviceCount( &n ) != hipSuccess )
		return 0;
	return n;
}

// LDS of one search workgroup: program image | queue | tile | 6 bit vectors | lean records
static size_t search_lds_bytes( int prog_bytes, const rmd_program_t &dp, int tile_t, bool lean, int qcap, int group = 1 )
{
	const int	tile_bytes = tile_t + dp.w_winsize + dp.lmargin + dp.rmargin + 16;
	const size_t	pb_bytes = ( lean ? 6 : 1 + 5 * size_t( dp.n_rowsets ) ) * ( size_t( tile_bytes + 63 ) / 64 + 3 ) * sizeof( unsigned long long );
	size_t	lds = size_t( prog_bytes ) + size_t( qcap ) * sizeof( unsigned ) +
		size_t( group ) * ( ( ( size_t( tile_bytes ) + 15 ) & ~size_t( 15 ) ) + pb_bytes );
	lds += lean ? size_t( dp.n_searches ) * 256 * LEAN_REC_BYTES : size_t( dp.n_rec_dwords ) * 256 * 4;
	if( !lean && dp.split_s >= 0 )		// resume states of the levels up to the split level, queue of continuations
		lds += size_t( dp.split_s + 1 ) * 256 * 8 + size_t( DEEP_QUEUE ) * ( 2 + 2 * ( dp.split_s + 1 ) ) * 4;
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
	if( const char *sb = getenv( "RNAMOTIF_BUDGET" ) )		// launch-shape switch: iterations per step
		sc->dprog.step_budget = std::max( 4, atoi( sb ) );
	for( int k = 0; k < prog->n_efn_sites; k++ ){
		if( prog->efn_sites[ k ].kind == RMA_EFN_KIND_EFN2 )
			sc->need_efn2 = true;	// tables come with rma_scanner_set_efn2data(), checked at the first scan
		else if( efn == nullptr ){
			snprintf( err, errlen, "the program has efn() call sites but no energy tables were given" );
			return 1;
		}
	}
	sc->device = device;
	HIPCHK( hipSetDevice( device ) );
	HIPCHK( hipStreamCreate( &sc->stream ) );
	for( int i = 0; i < 4; i++ )
		HIPCHK( hipEventCreate( &sc->ev[ i ] ) );
	{
		// the device gets the compact image; sc->dprog stays the full struct for the host
		std::vector<char>	img( sizeof( rmd_program_t ) );
		sc->prog_bytes = int( rmd_make_image( &sc->dprog, img.data() ) );
		HIPCHK( hipMalloc( &sc->d_prog, size_t( sc->prog_bytes ) ) );
		HIPCHK( hipMemcpy( sc->d_prog, img.data(), size_t( sc->prog_bytes ), hipMemcpyHostToDevice ) );
	}
	HIPCHK( hipMalloc( &sc->d_counters, 96 * sizeof( unsigned long long ) ) );
	if( efn != nullptr ){
		std::vector<int16_t>	t16;
		std::vector<int32_t>	tlkey;
		build_tables16( efn, t16, tlkey );
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
	sc->spill_cap = SPILL_ITEMS;
	if( const char *sp = getenv( "RNAMOTIF_SPILL" ) )	// tests: 0 = overflow searched in place
		sc->spill_cap = std::max( 0, atoi( sp ) );
	HIPCHK( hipMalloc( &sc->d_spill, std::max<size_t>( size_t( sc->grid_blocks ) * sc->spill_cap, 1 ) * sizeof( unsigned ) ) );
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
		const size_t	budget = ( 160 * 1024 ) / SEARCH_WAVES_PER_SIMD - 64;	// (static __shared__: 32 bytes)
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
	}
	if( !sc->dprog.lean_ok ){
		// general instance: its records take 12 bytes per level and lane of LDS next to the tile;
		// as many workgroups per CU as still leave a tile of a few thousand positions (what the
		// queue cannot hold spills to HBM)
		sc->qcap = 512;
		sc->tile_t = 1024;
		bool	found = false;
		int	kinds = 0;
		for( int k = 0; k < sc->dprog.n_searches; k++ ){
			const int	t = sc->dprog.elems[ sc->dprog.searches[ k ] ].type;
			if( t == RMA_T_P5 || t == RMA_T_T1 || t == RMA_T_Q1 )
				kinds |= RMD_KIND_TQ;
		}
		for( int wg = GENERAL_WAVES( kinds ); wg >= 1 && !found; wg-- ){
			const size_t	budget = ( 160 * 1024 ) / wg - 2560;	// (static __shared__ -- 1 KB of it the pre-filter's wave buffers -- and allocation granules)
			for( int t = 8192; t >= ( wg > 1 ? 3072 : 1024 ); t -= 256 )
				if( search_lds_bytes( sc->prog_bytes, sc->dprog, t, false, sc->qcap ) <= budget ){
					sc->tile_t = t;
					found = true;
					break;
				}
		}
	}
	if( const char *qq = getenv( "RNAMOTIF_QCAP" ) )
		if( atoi( qq ) >= 64 && atoi( qq ) <= 16384 )
			sc->qcap = ( atoi( qq ) + 3 ) & ~3;	// (what follows the queue in LDS is read 8 bytes at a time)
	const char	*tt = getenv( "RNAMOTIF_TILE" );
	if( tt != nullptr && atoi( tt ) > 0 && atoi( tt ) <= 16384 )
		sc->tile_t = atoi( tt );
	// room for the candidates of a few hundred Mbase at the densities of the reference's descriptors
	// (63 per Mbase for trna.descr); a scan that finds more is repeated into a buffer of the right
	// size (count-then-emit, rma_scan_device)
	sc->hit_cap = 1 << 17;
	HIPCHK( hipMalloc( &sc->d_hits, size_t( sc->hit_cap ) * sc->dprog.hit_stride * sizeof( int32_t ) ) );
	HIPCHK( hipHostMalloc( reinterpret_cast<void **>( &sc->h_ctr ), 4 * sizeof( unsigned long long ), hipHostMallocDefault ) );
	// the ordering's buffers, and one pass over whatever the hit buffer holds so that its kernels are
	// loaded: 12 ms that would otherwise fall into the first scan
	if( sc->dsort.reserve( sc->hit_cap, sc->dprog.hit_stride ) == hipSuccess ){
		( void )sc->dsort.run( sc->d_hits, 4096, 10, 20, 8, sc->stream );
		( void )hipStreamSynchronize( sc->stream );
	}
	( void )hipGetLastError();
	guard.p = nullptr;
	*out = sc;
	// One scan of eight start positions, so that what the runtime sets up on first use (code objects of
	// the kernel instance this descriptor takes, the first device allocations of a database) belongs to
	// the creation of the scanner and not to the first batch of a search: 15-50 ms there.  All 'a':
	// nothing pairs, whatever the descriptor, so no helix is ever walked.
	if( !getenv( "RNAMOTIF_NO_WARMUP" ) && sc->prog.dminlen <= 2000 ){
		const std::string	warm( size_t( std::max( sc->prog.dminlen, 1 ) + 7 ), 'a' );
		const char	*seqs[ 1 ] = { warm.c_str() };
		const int32_t	lens[ 1 ] = { int32_t( warm.size() ) };
		rma_db_t	*wdb = nullptr;
		char	werr[ 256 ];
		if( rma_db_create( sc, seqs, lens, 1, &wdb, werr, sizeof( werr ) ) == 0 ){
			const int32_t	*wh = nullptr;
			int64_t	wn = 0;
			( void )rma_scan( sc, wdb, &wh, &wn, werr, sizeof( werr ) );	// (efn2 tables not set yet: refused, harmless)
			rma_db_destroy( wdb );
		}
		( void )hipGetLastError();
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
	for( int i = 0; i < 4; i++ )
		if( sc->ev[ i ] )
			( void )hipEventDestroy( sc->ev[ i ] );
	if( sc->stream )
		( void )hipStreamDestroy( sc->stream );
	delete sc;
}

// Upload n packed entries: codes/amask hold n_code_words/n_mask_words words, base_off[] are
// offsets in bases (multiples of 32) relative to the first word of the arrays.
static int db_upload( rma_scanner_t *sc, const uint32_t *codes, size_t n_code_words, const uint32_t *amask,
	size_t n_mask_words, const int64_t *base_off, const int32_t *slen, int32_t n, rma_db_t **out, char *err, size_t errlen,
	const int32_t *pos_lo = nullptr, const int32_t *pos_hi = nullptr )
{
	if( pos_lo != nullptr )
		for( int i = 0; i < n; i++ )
			if( pos_lo[ i ] < 0 || pos_hi[ i ] < pos_lo[ i ] ){
				snprintf( err, errlen, "entry %d: start positions [%d, %d) are not a range", i, pos_lo[ i ], pos_hi[ i ] );
				return 1;
			}
	rma_db	*db = new rma_db;
	db->sc = sc;
	db->device = sc->device;
	db->n_seq = n;
	// every early return below frees what has been allocated so far
	struct DbGuard { rma_db *p; ~DbGuard(){ if( p ) rma_db_destroy( p ); } }	guard{ db };
	db->total_bases = 0;
	db->strands = sc->prog.chk_both_strs ? 2 : 1;
	std::vector<int64_t>	tile_start( size_t( n ) + 1, 0 );
	// Launch shape.  Long entries: the scanner's tile, one per workgroup pass.  A database of
	// many short entries (GenBank divisions, transcript sets) never fills such a tile, and a
	// few dozen queue items cannot occupy 256 lanes: it gets small tiles in groups of
	// SHORT_GROUP per workgroup pass (rma_search_kernel<.., G>), if the descriptor is lean and
	// the group fits the LDS budget.
	db->tile_t = sc->tile_t;
	db->qcap = sc->qcap;
	db->group = 1;
	{
		int64_t	tot = 0;
		for( int i = 0; i < n; i++ ){
			tot += slen[ i ];
			db->max_slen = std::max( db->max_slen, slen[ i ] );
		}
		bool	grouped = n >= 64 && tot / n < SHORT_ENTRY_MEAN && !getenv( "RNAMOTIF_TILE" );
		if( const char *force = getenv( "RNAMOTIF_SHORT" ) )	// "0" never, "1" always (tests)
			grouped = force[ 0 ] == '1';
		if( grouped && sc->dprog.lean_ok ){
			const size_t	budget = ( 160 * 1024 ) / SEARCH_WAVES_PER_SIMD - 64 - SHORT_GROUP * 32;
			// (tiles of 1024 positions measured slower than of 768 where both fit: mp.ends 1.56 / 1.40 ms)
			for( int t = 768; t >= 256; t -= 256 ){
				int	q = 256;	// LDS goes to the slots; what a group queues beyond this spills
				if( const char *qq = getenv( "RNAMOTIF_QCAP" ) )	// tests: force the overflow path
					q = std::max( 64, atoi( qq ) );
				if( search_lds_bytes( sc->prog_bytes, sc->dprog, t, true, q, SHORT_GROUP ) <= budget ){
					db->tile_t = t;
					db->qcap = q;
					db->group = SHORT_GROUP;
					break;
				}
			}
		}
	}
	const int	T = db->tile_t;
	for( int i = 0; i < n; i++ ){
		int64_t	nsz = int64_t( slen[ i ] ) - sc->prog.dminlen + 1;	// start positions of a strand
		if( pos_lo != nullptr ){
			// this database answers for a slice of the entry's start positions only
			db->total_bases += std::max<int64_t>( 0, std::min<int64_t>( pos_hi[ i ], slen[ i ] ) - pos_lo[ i ] );
			nsz = std::min<int64_t>( nsz, pos_hi[ i ] ) - pos_lo[ i ];
		}else
			db->total_bases += slen[ i ];
		int64_t	nt = nsz > 0 ? ( nsz + T - 1 ) / T : 0;
		tile_start[ i + 1 ] = tile_start[ i ] + nt * db->strands;
	}
	db->n_tiles = tile_start[ n ];
	HIPCHK( hipSetDevice( sc->device ) );
	size_t	nc = std::max<size_t>( n_code_words, 1 ), na = std::max<size_t>( n_mask_words, 1 );
	HIPCHK( hipMalloc( &db->d_codes, nc * sizeof( uint32_t ) ) );
	HIPCHK( hipMalloc( &db->d_amask, na * sizeof( uint32_t ) ) );
	HIPCHK( hipMalloc( &db->d_base_off, std::max<size_t>( n, 1 ) * sizeof( int64_t ) ) );
	HIPCHK( hipMalloc( &db->d_slen, std::max<size_t>( n, 1 ) * sizeof( int32_t ) ) );
	HIPCHK( hipMalloc( &db->d_tile_start, ( size_t( n ) + 1 ) * sizeof( int64_t ) ) );
	if( n_code_words > 0 ){
		HIPCHK( hipMemcpy( db->d_codes, codes, n_code_words * sizeof( uint32_t ), hipMemcpyHostToDevice ) );
		HIPCHK( hipMemcpy( db->d_amask, amask, n_mask_words * sizeof( uint32_t ), hipMemcpyHostToDevice ) );
	}
	if( n > 0 ){
		HIPCHK( hipMemcpy( db->d_base_off, base_off, size_t( n ) * sizeof( int64_t ), hipMemcpyHostToDevice ) );
		HIPCHK( hipMemcpy( db->d_slen, slen, size_t( n ) * sizeof( int32_t ), hipMemcpyHostToDevice ) );
	}
	HIPCHK( hipMemcpy( db->d_tile_start, tile_start.data(), tile_start.size() * sizeof( int64_t ), hipMemcpyHostToDevice ) );
	{
		std::vector<int32_t>	tile_seq( size_t( std::max<int64_t>( db->n_tiles, 1 ) ) );
		for( int i = 0; i < n; i++ )
			for( int64_t t = tile_start[ i ]; t < tile_start[ i + 1 ]; t++ )
				tile_seq[ size_t( t ) ] = i;
		HIPCHK( hipMalloc( &db->d_tile_seq, tile_seq.size() * sizeof( int32_t ) ) );
		HIPCHK( hipMemcpy( db->d_tile_seq, tile_seq.data(), tile_seq.size() * sizeof( int32_t ), hipMemcpyHostToDevice ) );
	}
	if( pos_lo != nullptr && n > 0 ){
		HIPCHK( hipMalloc( &db->d_pos_lo, size_t( n ) * sizeof( int32_t ) ) );
		HIPCHK( hipMalloc( &db->d_pos_hi, size_t( n ) * sizeof( int32_t ) ) );
		HIPCHK( hipMemcpy( db->d_pos_lo, pos_lo, size_t( n ) * sizeof( int32_t ), hipMemcpyHostToDevice ) );
		HIPCHK( hipMemcpy( db->d_pos_hi, pos_hi, size_t( n ) * sizeof( int32_t ), hipMemcpyHostToDevice ) );
	}
	guard.p = nullptr;
	*out = db;
	return 0;
}

extern "C" int rma_db_create_ranges( rma_scanner_t *sc, const char *const *seqs, const int32_t *slens,
	const int32_t *pos_lo, const int32_t *pos_hi, int32_t n, rma_db_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	rma::PackedDb	pk;
	for( int i = 0; i < n; i++ )
		pk.add( seqs[ i ], slens[ i ] < 0 ? 0 : slens[ i ] );
	return db_upload( sc, pk.codes.data(), pk.codes.size(), pk.amask.data(), pk.amask.size(),
		pk.base_off.data(), pk.slen.data(), n, out, err, errlen, pos_lo, pos_hi );
}

extern "C" int rma_db_create( rma_scanner_t *sc, const char *const *seqs, const int32_t *slens, int32_t n,
	rma_db_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	rma::PackedDb	pk;
	for( int i = 0; i < n; i++ )
		pk.add( seqs[ i ], slens[ i ] < 0 ? 0 : slens[ i ] );
	return db_upload( sc, pk.codes.data(), pk.codes.size(), pk.amask.data(), pk.amask.size(),
		pk.base_off.data(), pk.slen.data(), n, out, err, errlen );
}

const rma::PackFile *rma_pack_file( const rma_pack_t *pk );	// rm_capi.cpp

extern "C" int rma_db_create_packed( rma_scanner_t *sc, const rma_pack_t *pack, int32_t first, int32_t count,
	rma_db_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	const rma::PackFile	&pf = *rma_pack_file( pack );
	if( first < 0 || count < 0 || first + count > pf.count() ){
		snprintf( err, errlen, "entries [%d, %d) are outside the packed database (%d entries)", first, first + count, pf.count() );
		return 1;
	}
	if( count == 0 )
		return db_upload( sc, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, out, err, errlen );
	const int64_t	b0 = pf.base_off[ first ];
	const int	last = first + count - 1;
	const int64_t	b1 = pf.base_off[ last ] + ( ( int64_t( pf.slen[ last ] ) + 31 ) / 32 ) * 32;
	std::vector<int64_t>	rel;
	rel.resize( size_t( count ) );
	for( int i = 0; i < count; i++ )
		rel[ i ] = pf.base_off[ first + i ] - b0;
	return db_upload( sc, pf.codes.data() + b0 / 16, size_t( ( b1 - b0 ) / 16 ), pf.amask.data() + b0 / 32,
		size_t( ( b1 - b0 ) / 32 ), rel.data(), pf.slen.data() + first, count, out, err, errlen );
}

extern "C" int rma_db_create_packed_ranges( rma_scanner_t *sc, const rma_pack_t *pack, const int32_t *entry,
	const int32_t *pos_lo, const int32_t *pos_hi, int32_t n, rma_db_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	const rma::PackFile	&pf = *rma_pack_file( pack );
	// the chosen entries side by side (every entry starts on a 32-base boundary: whole words)
	std::vector<uint32_t>	codes, amask;
	std::vector<int64_t>	rel( size_t( std::max( n, 0 ) ) );
	std::vector<int32_t>	slen( size_t( std::max( n, 0 ) ) );
	for( int i = 0; i < n; i++ ){
		const int	e = entry[ i ];
		if( e < 0 || e >= pf.count() ){
			snprintf( err, errlen, "entry %d is outside the packed database (%d entries)", e, pf.count() );
			return 1;
		}
		const int64_t	w1 = pf.base_off[ e ] / 32, nw1 = ( int64_t( pf.slen[ e ] ) + 31 ) / 32;
		rel[ i ] = int64_t( amask.size() ) * 32;
		slen[ i ] = pf.slen[ e ];
		codes.insert( codes.end(), pf.codes.begin() + 2 * w1, pf.codes.begin() + 2 * ( w1 + nw1 ) );
		amask.insert( amask.end(), pf.amask.begin() + w1, pf.amask.begin() + w1 + nw1 );
	}
	return db_upload( sc, codes.data(), codes.size(), amask.data(), amask.size(), rel.data(), slen.data(), n, out, err, errlen,
		pos_lo, pos_hi );
}

extern "C" void rma_db_destroy( rma_db_t *db )
{
	if( db == nullptr )
		return;
	( void )hipSetDevice( db->device );
	( void )hipFree( db->d_codes );
	( void )hipFree( db->d_amask );
	( void )hipFree( db->d_base_off );
	( void )hipFree( db->d_slen );
	( void )hipFree( db->d_tile_start );
	( void )hipFree( db->d_tile_seq );
	( void )hipFree( db->d_pos_lo );
	( void )hipFree( db->d_pos_hi );
	delete db;
}

extern "C" int64_t rma_db_bases( const rma_db_t *db ) { return db->total_bases; }

static DbView view_of( const rma_scanner *sc, const rma_db *db )
{
	DbView	v;
	v.codes = db->d_codes;
	v.amask = db->d_amask;
	v.base_off = db->d_base_off;
	v.slen = db->d_slen;
	v.tile_start = db->d_tile_start;
	v.tile_seq = db->d_tile_seq;
	v.pos_lo = db->d_pos_lo;
	v.pos_hi = db->d_pos_hi;
	v.n_seq = db->n_seq;
	v.strands = db->strands;
	v.tile_t = db->tile_t;
	v.n_tiles = db->n_tiles;
	return v;
}

// wait = false: the efn kernel is left running on the scanner's stream (rma_scan() queues the ordering
// and the copy back behind it and waits once)
static int scan_device( rma_scanner_t *sc, const rma_db_t *db, int64_t *n_hits, float *search_ms,
	float *efn_ms, char *err, size_t errlen, bool wait )
{
	constexpr int	BLOCK = 256;
	HIPCHK( hipSetDevice( sc->device ) );
	*n_hits = 0;
	if( search_ms ) *search_ms = 0;
	if( efn_ms ) *efn_ms = 0;
	if( db->sc != sc ){
		snprintf( err, errlen, "the database was created for another scanner (tiles are laid out per scanner)" );
		return 1;
	}
	if( sc->need_efn2 && sc->d_efn2 == nullptr ){
		snprintf( err, errlen, "the program has efn2() call sites but rma_scanner_set_efn2data() was not called" );
		return 1;
	}
	if( db->n_tiles == 0 )
		return 0;
	DbView	v = view_of( sc, db );
	const rmd_program_t	&dp = sc->dprog;
	const int	dbg = getenv( "RNAMOTIF_DBG" ) ? atoi( getenv( "RNAMOTIF_DBG" ) ) : 0;
	const bool	lean = dp.lean_ok && !( dbg & 16 );
	const bool	grouped = lean && db->group > 1;
	int	tile_bytes = db->tile_t + dp.w_winsize + dp.lmargin + dp.rmargin + 16;
	size_t	lds = search_lds_bytes( sc->prog_bytes, dp, db->tile_t, lean, db->qcap, grouped ? SHORT_GROUP : 1 );
	if( lds > 150 * 1024 ){
		snprintf( err, errlen, "window of %d bases does not fit the LDS tile (%zu bytes needed)", dp.w_winsize, lds );
		return 1;
	}
	// the kernel instance: lean (one tile or a group of small ones per pass), or the general one
	// compiled for the kinds of element the descriptor has
	int	kinds = 0;
	for( int k = 0; k < dp.n_searches; k++ ){
		const rmd_elem_t	&e = dp.elems[ dp.searches[ k ] ];
		if( e.type == RMA_T_H5 && !e.proper )
			kinds |= RMD_KIND_PK;
		if( e.type == RMA_T_P5 || e.type == RMA_T_T1 || e.type == RMA_T_Q1 )
			kinds |= RMD_KIND_TQ;
	}
	// the pooled lean instance (see the kernel): when the window of an item, four bits a base, fits the
	// column a lane gets of the tile's place in LDS
	bool	pooled = false;
	if( lean && !grouped ){
		const int	n_dw = ( dp.w_winsize + dp.lmargin + dp.rmargin + 14 ) / 8;
		const size_t	room = size_t( ( tile_bytes + 15 ) & ~15 ) + size_t( 6 ) * ( ( tile_bytes + 63 ) / 64 + 3 ) * sizeof( unsigned long long );
		pooled = n_dw <= 32 && size_t( n_dw ) * BLOCK * sizeof( uint32_t ) <= room;
		if( const char *pl = getenv( "RNAMOTIF_POOL" ) )	// "0": pass B tile by tile (tests, profiles/pool_matrix.py)
			pooled = pooled && atoi( pl ) != 0;
	}
	if( pooled ){
		if( const char *pm = getenv( "RNAMOTIF_POOL_MIN" ) )
			sc->pool_min = std::max( 1, atoi( pm ) );
		const int	cap = sc->pool_min + db->qcap + sc->spill_cap;
		if( cap > sc->pool_cap ){
			( void )hipFree( sc->d_pool );
			sc->d_pool = nullptr;
			sc->pool_cap = 0;
			HIPCHK( hipMalloc( &sc->d_pool, size_t( sc->grid_blocks ) * cap * 3 * sizeof( unsigned ) ) );
			sc->pool_cap = cap;
		}
	}
	typedef void	( *kernel_t )( const rmd_program_t *, int, int, DbView, HitBuf, int, int );
	const kernel_t	kernel = pooled ? &rma_search_kernel<BLOCK, true, 1, 0, true> :
		grouped ? &rma_search_kernel<BLOCK, true, SHORT_GROUP> : lean ? &rma_search_kernel<BLOCK, true, 1> :
		kinds == 0 ? &rma_search_kernel<BLOCK, false, 1, 0> : kinds == RMD_KIND_PK ? &rma_search_kernel<BLOCK, false, 1, RMD_KIND_PK> :
		kinds == RMD_KIND_TQ ? &rma_search_kernel<BLOCK, false, 1, RMD_KIND_TQ> : &rma_search_kernel<BLOCK, false, 1, RMD_KIND_PK | RMD_KIND_TQ>;
	HIPCHK( hipFuncSetAttribute( reinterpret_cast<const void *>( kernel ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t	n_units = grouped ? ( db->n_tiles + SHORT_GROUP - 1 ) / SHORT_GROUP : db->n_tiles;
	int	grid = int( std::min<int64_t>( n_units, sc->grid_blocks ) );
	unsigned long long	count = 0;
	for( int attempt = 0; attempt < 4; attempt++ ){
		HIPCHK( hipMemsetAsync( sc->d_counters, 0, 96 * sizeof( unsigned long long ), sc->stream ) );
		HitBuf	hb{ sc->d_hits, sc->d_counters, sc->d_counters + 1, sc->hit_cap, sc->d_spill, sc->spill_cap, sc->d_pool, sc->pool_cap, sc->pool_min,
			getenv( "RNAMOTIF_POOL_REFILL" ) ? atoi( getenv( "RNAMOTIF_POOL_REFILL" ) ) : 48 };
		HIPCHK( hipEventRecord( sc->ev[ 0 ], sc->stream ) );
		hipLaunchKernelGGL( kernel, dim3( grid ), dim3( BLOCK ), lds, sc->stream,
			sc->d_prog, sc->prog_bytes, db->qcap, v, hb, tile_bytes, dbg );
		HIPCHK( hipGetLastError() );
		HIPCHK( hipEventRecord( sc->ev[ 1 ], sc->stream ) );
		// [0] candidates, [3] queue overflow of the general instance: one copy, one wait
		HIPCHK( hipMemcpyAsync( sc->h_ctr, sc->d_counters, 4 * sizeof( unsigned long long ), hipMemcpyDeviceToHost, sc->stream ) );
		HIPCHK( hipStreamSynchronize( sc->stream ) );
		count = sc->h_ctr[ 0 ];
		if( getenv( "RNAMOTIF_DBG" ) ){
			unsigned long long	q = 0;
			( void )hipMemcpy( &q, sc->d_counters + 2, sizeof( q ), hipMemcpyDeviceToHost );
			fprintf( stderr, "[dbg] queued items: %llu, candidates %llu (tile %d x %d, queue %d, LDS %zu, %lld tiles)\n", q, count,
				db->tile_t, grouped ? db->group : 1, db->qcap, lds, ( long long )db->n_tiles );
			if( dbg & 32 ){
				unsigned long long	ph[ 6 ];
				( void )hipMemcpy( ph, sc->d_counters + 4, sizeof( ph ), hipMemcpyDeviceToHost );
				double	tot = 0;
				for( int i = 0; i < 6; i++ )
					tot += double( ph[ i ] );
				unsigned long long	lv[ 64 ];
				( void )hipMemcpy( lv, sc->d_counters + 16, sizeof( lv ), hipMemcpyDeviceToHost );
				if( lean )
					fprintf( stderr, "[dbg] pass B: %llu pop rounds of %.1f lanes, %llu steps of %.1f lanes; wave cycles popping %.1f%%, stepping %.1f%%\n",
						lv[ 0 ], lv[ 0 ] ? double( lv[ 1 ] ) / lv[ 0 ] : 0.0, lv[ 2 ], lv[ 2 ] ? double( lv[ 3 ] ) / lv[ 2 ] : 0.0,
						100.0 * lv[ 4 ] / double( lv[ 4 ] + lv[ 5 ] + 1 ), 100.0 * lv[ 5 ] / double( lv[ 4 ] + lv[ 5 ] + 1 ) );
				for( int kk = 0; kk < dp.n_searches && kk < 32 && !lean; kk++ )
					fprintf( stderr, "[dbg] level %2d (element %2d, type %d): %llu wave rounds, %.1f lanes each\n", kk, dp.searches[ kk ],
						dp.elems[ dp.searches[ kk ] ].type, lv[ 2 * kk ], lv[ 2 * kk ] ? double( lv[ 2 * kk + 1 ] ) / lv[ 2 * kk ] : 0.0 );
				fprintf( stderr, "[dbg] wave cycles: decode %.1f%%, literal %.1f%%, rows %.1f%%, pre-filter %.1f%%, search %.1f%%, waiting %.1f%%\n",
					100 * ph[ 0 ] / tot, 100 * ph[ 1 ] / tot, 100 * ph[ 2 ] / tot, 100 * ph[ 3 ] / tot, 100 * ph[ 4 ] / tot, 100 * ph[ 5 ] / tot );
			}
		}
		if( !lean ){
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
				HIPCHK( hipMalloc( &sc->d_spill, size_t( sc->grid_blocks ) * sc->spill_cap * sizeof( unsigned ) ) );
				continue;
			}
		}
		if( int64_t( count ) <= sc->hit_cap )
			break;
		if( attempt == 3 ){
			snprintf( err, errlen, "hit buffer overflow after regrow (%llu candidates)", count );
			return 1;
		}
		// count-then-emit: the first pass told us how many records there are
		( void )hipFree( sc->d_hits );
		sc->d_hits = nullptr;
		sc->hit_cap = int64_t( count ) + 1024;
		HIPCHK( hipMalloc( &sc->d_hits, size_t( sc->hit_cap ) * dp.hit_stride * sizeof( int32_t ) ) );
	}
	if( search_ms )
		HIPCHK( hipEventElapsedTime( search_ms, sc->ev[ 0 ], sc->ev[ 1 ] ) );
	*n_hits = int64_t( count );
	if( ( sc->have_efn || sc->d_efn2 != nullptr ) && dp.n_efn > 0 && count > 0 ){
		constexpr int	EB = EFN_BLOCK;
		// one workgroup per CU at most (its LDS), each striding over the candidates
		const int64_t	blocks = std::min<int64_t>( ( int64_t( count ) + EB - 1 ) / EB, sc->grid_blocks / 8 );
		HIPCHK( hipEventRecord( sc->ev[ 2 ], sc->stream ) );
		hipLaunchKernelGGL( rma_efn_kernel<EB>, dim3( unsigned( blocks ) ), dim3( EB ), 0, sc->stream,
			sc->d_prog, v, sc->d_hits, ( long long )count, sc->have_efn ? sc->d_t16 : nullptr, sc->d_tlkey, sc->d_loginc,
			sc->d_efn2 );
		HIPCHK( hipGetLastError() );
		HIPCHK( hipEventRecord( sc->ev[ 3 ], sc->stream ) );
		if( wait || efn_ms )
			HIPCHK( hipStreamSynchronize( sc->stream ) );
		if( efn_ms )
			HIPCHK( hipEventElapsedTime( efn_ms, sc->ev[ 2 ], sc->ev[ 3 ] ) );
	}
	return 0;
}

extern "C" int rma_scan_device( rma_scanner_t *sc, const rma_db_t *db, int64_t *n_hits, float *search_ms,
	float *efn_ms, char *err, size_t errlen )
{
	return scan_device( sc, db, n_hits, search_ms, efn_ms, err, errlen, true );
}

extern "C" int rma_scan( rma_scanner_t *sc, const rma_db_t *db, const int32_t **hits, int64_t *n_hits,
	char *err, size_t errlen )
{
	*hits = nullptr;
	int64_t	n = 0;
	const bool	timing = getenv( "RNAMOTIF_TIMING" ) != nullptr;
	auto	t0 = std::chrono::steady_clock::now();
	auto lap = [&]( const char *what ){
		if( timing ){
			auto	t1 = std::chrono::steady_clock::now();
			fprintf( stderr, "[timing] %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>( t1 - t0 ).count() );
			t0 = t1;
		}
	};
	if( scan_device( sc, db, &n, nullptr, nullptr, err, errlen, false ) )
		return 1;
	lap( "search" );
	*n_hits = n;
	if( n == 0 )
		return 0;
	const int	stride = sc->dprog.hit_stride;
	// pinned staging buffer: the copy back is a single DMA
	const size_t	words = size_t( n ) * stride;
	if( words > sc->h_raw_cap ){
		if( sc->h_raw != nullptr )
			( void )hipHostFree( sc->h_raw );
		sc->h_raw = nullptr;
		sc->h_raw_cap = 0;
		HIPCHK( hipHostMalloc( reinterpret_cast<void **>( &sc->h_raw ), words * 2 * sizeof( int32_t ), hipHostMallocDefault ) );
		sc->h_raw_cap = words * 2;
	}
	// Reference order -- (entry, strand, start, rank, order), order word renumbered -- on the device,
	// behind the efn kernel on the same stream: what comes back is the final stream (rm_hitsort_dev.h).
	// Header words that do not fit the 64-bit key (or RNAMOTIF_HOSTSORT=1): the host's sort_hits().
	const bool	host_sort = getenv( "RNAMOTIF_HOSTSORT" ) != nullptr && atoi( getenv( "RNAMOTIF_HOSTSORT" ) ) != 0;
	bool	on_device = false;
	if( !host_sort && n >= 2 ){
		auto	bits_of = []( unsigned x ){ int b = 0; while( x ){ b++; x >>= 1; } return b; };
		if( sc->dsort.reserve( sc->hit_cap, stride ) == hipSuccess &&
			sc->dsort.run( sc->d_hits, n, bits_of( unsigned( db->n_seq > 0 ? db->n_seq - 1 : 0 ) ), bits_of( unsigned( db->max_slen ) ),
				bits_of( unsigned( sc->dprog.w_winsize ) ), sc->stream ) == hipSuccess ){
			int	flag = 1;
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
		*hits = sc->h_raw;
		return 0;
	}
	HIPCHK( hipMemcpyAsync( sc->h_raw, sc->d_hits, words * sizeof( int32_t ), hipMemcpyDeviceToHost, sc->stream ) );
	HIPCHK( hipStreamSynchronize( sc->stream ) );
	lap( "copy back" );
	sc->h_sorted.resize( words );
	rma::sort_hits( sc->h_raw, n, stride, sc->h_sorted.data(), sc->keys, sc->keys_tmp );
	lap( "ordering" );
	*hits = sc->h_sorted.data();
	return 0;
}

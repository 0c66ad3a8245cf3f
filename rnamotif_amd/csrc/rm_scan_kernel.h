// rm_scan_kernel.h -- device code of the scan path on MI355X (gfx950): the search kernel and
// the efn kernel as templates.  Included by the rm_scan_inst_*.hip translation units only, each
// of which instantiates some of the kernel's instances behind the launchers rm_kernels.h declares
// (one file per class of descriptor, so that the instances compile side by side).
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
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#define RMD_HD		__host__ __device__ inline
#define RMD_FN		static __device__ inline
#define RMD_COLD	static __device__ inline
#define RMD_FN_MEMBER	__device__ inline
#include "rm_scan_core.h"
#include "rm_efn_core.h"
#include "rm_efn2_core.h"
#include "rm_kernels.h"

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


// Lean path records of one lane in LDS, 6 bytes per level: windows of lean descriptors are
// shorter than 4096 (rmd_build), so window start and saved end take 12 bits each, the next
// end position (down to -2) 13, the helix length 6, the phase 1.
// Level k: a dword at lo[ k * BLOCK ] and a half word at hi[ k * BLOCK ], lane-contiguous.
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

// ---------------------------------------------------------------- a complete match, the whole wave on it
// What rmd_lean_emit() does for one lane -- rebuild the element table from the search records, run the
// end-of-list checks, store the candidate (find_ss :362-393) -- takes that lane some 150 000 cycles: the
// table (s_matchoff ... of up to 100 elements, indexed at run time) lives in scratch memory, and
// chk_motif's fm_window lookups walk it element by element, four times per strict helix; the other lanes
// of the wave wait meanwhile, and an item with twenty alternatives holds its wave for milliseconds
// (RNAMOTIF_DBG=34: 38 % of pass B's wave cycles over trna.descr, and all of its tail).  Here the
// wave does it together: lane j rebuilds search level j from the records of the lane that
// found the match (levels j and up to 16), lane e keeps element e of the table in registers, a window lookup is a
// ballot, the candidate's words go out side by side.  No scratch memory, a few thousand cycles.
struct WaveTable {
	int	moff, mlen, type, mpr, mm;		// of element `lane` (lean descriptors have at most 32)
	int	l_off, l_len, r_off, r_len, l_mm, r_mm;		// (the same in every lane)
	int	slen;
	// (d is the same in every lane wherever the checks call these: a scalar read of lane d's register)
	__device__ inline int	off( int d ) const { return __builtin_amdgcn_readlane( moff, __builtin_amdgcn_readfirstlane( d ) ); }
	__device__ inline int	len( int d ) const { return __builtin_amdgcn_readlane( mlen, __builtin_amdgcn_readfirstlane( d ) ); }
	__device__ inline int	wtype( const rmd_program_t *, int pos, int undef_is_ss ) const	// rmd_lane_t::wtype: the first element that covers pos
	{
		const unsigned long long	b = __ballot( mlen > 0 && pos >= moff && pos < moff + mlen );
		if( b == 0 )
			return undef_is_ss ? RMA_T_SS : -1;
		return __builtin_amdgcn_readlane( type, __builtin_amdgcn_readfirstlane( __ffsll( b ) - 1 ) );
	}
};

// All 64 lanes call this with the same arguments: the records (lr), the sequence view and the search
// state of the lane whose search reached the end of the list.  True when the candidate passed the checks
// (and was stored, room permitting): the caller counts that lane's `order` up.
template< int BLOCK, class LR, class SQ >
__device__ __forceinline__ bool wave_emit_body( const rmd_program_t *P, const LR lr, const SQ sq, int z, int slen, int rank, int order,
	int seq, int comp, int32_t *hits, unsigned long long *count, long long cap, int lane_id )
{
	// lane j: level j, as rmd_lean_emit's loop has it
	int	lz = 0, lc = 0, lx = 0, lm = 0;
	int	digits = 0;	// (the level's share of the order word, rmd_elem_t::ord_stride)
	if( lane_id < P->n_searches ){
		const int	kk = lane_id;
		const rmd_lrec_t	r = lr.get( kk );
		const int	d = P->searches[ kk ];
		const rmd_elem_t	&stp = P->elems[ d ];
		const int	zero = z + r.zero, cur = z + r.sd + 1;
		lz = zero;
		lc = cur;
		{
			const int	first = ( kk == 0 || !stp.loop ) ? int( r.sd ) + 1 : int( rmd_lean_open( P, kk, r.zero, r.osd ).sd );
			digits = ( ( first - ( int( r.sd ) + 1 ) ) * stp.ord_nlen + ( stp.type == RMA_T_SS ? 0 : int( r.hl ) - stp.minlen ) ) * stp.ord_stride;
		}
		if( stp.type == RMA_T_SS ){
			int	mm = 0;
			if( stp.re >= 0 && stp.mismatch > 0 )
				rmd_chk_seq( P, stp, sq, zero, cur - zero + 1, &mm );
			lm = mm & 0xffff;
		}else{
			const int	hl = r.hl;
			uint64_t	cand, mis;
			int	mm5 = 0, mm3 = 0;
			rmd_match_wchlx_mm( P, sq, d, stp.mates[ 0 ], zero, cur, rmd_s3lim( zero, cur, stp.minilen, stp.maxlen ), &cand, &mis, &mm5, &mm3 );
			lx = hl | ( rmd_popc64( mis & ( ( 1ull << hl ) - 1 ) ) << 8 );
			lm = ( mm5 & 0xffff ) | ( mm3 << 16 );
		}
	}
	// lane e: element e, from the level that placed it (a 3' strand: its helix's)
	WaveTable	tb;
	{
		const int	e = lane_id;
		const bool	in = e < P->n_elems;
		const rmd_elem_t	&el = P->elems[ in ? e : 0 ];
		const int	ty = el.type;
		const int	lv_ = ty == RMA_T_H3 ? P->elems[ el.mates[ 0 ] ].searchno : el.searchno;
		const bool	placed = in && lv_ >= 0;
		const int	src = placed ? lv_ : 0;
		// (every lane takes part in the exchanges)
		const int	zero = __shfl( lz, src ), cur = __shfl( lc, src ), x = __shfl( lx, src ), m = __shfl( lm, src );
		const int	hl = x & 0xff;
		tb.type = in ? ty : -1;
		tb.moff = !placed ? 0 : ty == RMA_T_H3 ? cur - hl + 1 : zero;
		tb.mlen = !placed ? 0 : ty == RMA_T_SS ? cur - zero + 1 : hl;
		tb.mpr = !placed || ty == RMA_T_SS ? 0 : x >> 8;
		tb.mm = !placed ? 0 : ty == RMA_T_H3 ? ( m >> 16 ) : int( int16_t( m & 0xffff ) );
	}
	if( P->ord_ok ){
		for( int o = 32; o > 0; o >>= 1 )
			digits += __shfl_xor( digits, o );
		order = digits;
	}
	tb.slen = slen;
	tb.l_mm = tb.r_mm = RMD_UNDEF;
	tb.l_off = tb.l_len = tb.r_off = tb.r_len = 0;
	if( P->strict_helices && !rmd_chk_motif( P, tb, sq ) )
		return false;
	if( !rmd_set_context( P, tb, sq ) )
		return false;
	if( !rmd_chk_sites( P, tb, sq ) )
		return false;
	unsigned long long	slot = 0;
	if( lane_id == 0 )
		slot = atomicAdd( count, 1ull );
	slot = __shfl( slot, 0 );
	if( slot < ( unsigned long long )cap ){
		int32_t	*w = hits + slot * P->hit_stride;		// (rmd_fill_hit's layout)
		const int	k = RMA_HIT_HDR + 4 * P->n_elems;
		if( lane_id == 0 ){
			w[ 0 ] = seq;
			w[ 1 ] = comp;
			w[ 2 ] = z;
			w[ 3 ] = rank;
			w[ 4 ] = order;
			w[ k + 0 ] = P->has_lctx ? tb.l_off : 0;
			w[ k + 1 ] = P->has_lctx ? tb.l_len : 0;
			w[ k + 2 ] = P->has_rctx ? tb.r_off : 0;
			w[ k + 3 ] = P->has_rctx ? tb.r_len : 0;
		}
		if( lane_id < P->n_elems ){
			w[ RMA_HIT_HDR + 4 * lane_id + 0 ] = tb.moff;
			w[ RMA_HIT_HDR + 4 * lane_id + 1 ] = tb.mlen;
			w[ RMA_HIT_HDR + 4 * lane_id + 2 ] = tb.mpr;
			w[ RMA_HIT_HDR + 4 * lane_id + 3 ] = tb.mm;
		}
		for( int e = lane_id; e < P->n_efn; e += 64 )
			w[ k + 4 + e ] = RMA_EFN_INFINITY;	// filled by the efn pass
	}
	return true;
}

// (out of line where complete matches are rare -- the search kernel's own walks; in line in the drain kernel)
template< int BLOCK, class LR, class SQ >
__device__ __noinline__ bool wave_emit( const rmd_program_t *P, const LR lr, const SQ sq, int z, int slen, int rank, int order,
	int seq, int comp, int32_t *hits, unsigned long long *count, long long cap, int lane_id )
{
	return wave_emit_body<BLOCK>( P, lr, sq, z, slen, rank, order, seq, comp, hits, count, cap, lane_id );
}

// The matches the lanes of a wave have pending after a step (rmd_lean_t::pending), one at a time;
// sq_of( l ): the sequence view lane l searches in.  Called by all 64 lanes.
template< int BLOCK, bool INLINE = false, class SQOF >
__device__ inline void wave_emit_pending( const rmd_program_t *P, const LdsRecs<BLOCK> &lr, rmd_lean_t &st, int k, const SQOF &sq_of,
	int seq, int comp, const HitBuf &hb, int lane_id, int order_base = 0 )
{
	for( unsigned long long em = __ballot( k >= 0 && st.pending ); em; em &= em - 1 ){
		const int	l = __ffsll( em ) - 1;
		const LdsRecs<BLOCK>	lrl{ lr.lo + ( l - lane_id ), lr.hi + ( l - lane_id ) };
		if( lane_id == l && st.only_hl >= 0 && !P->ord_ok && st.order >= ( 1 << PIECE_ORDER_BITS ) )
			atomicMax( hb.ticket + 2, 1ull );	// (the pieces' order words would run into each other)
		bool	stored;
		if constexpr( INLINE )
			stored = wave_emit_body<BLOCK>( P, lrl, sq_of( l ), __shfl( st.szero, l ), __shfl( st.slen, l ), __shfl( st.rank, l ),
				__shfl( st.order + order_base, l ), __shfl( seq, l ), __shfl( comp, l ), hb.hits, hb.count, hb.cap, lane_id );
		else
			stored = wave_emit<BLOCK>( P, lrl, sq_of( l ), __shfl( st.szero, l ), __shfl( st.slen, l ), __shfl( st.rank, l ),
				__shfl( st.order + order_base, l ), __shfl( seq, l ), __shfl( comp, l ), hb.hits, hb.count, hb.cap, lane_id );
		if( lane_id == l ){
			st.order += stored;
			st.pending = 0;
		}
	}
}

// Tiles over the concatenation of the entries (DbView::concat_bases > 0; databases of short entries): pass A works on a
// strand of the whole packed array as if it were one long entry -- full vectors whatever the entries' lengths, and no
// knowledge of where an entry ends: all its tests are necessary conditions on the bases between a start and an end
// position, which for a true candidate lie in one entry.  What they let through is brought back to its entry here, when
// it leaves the tile: start position szero_u and end rank r_u (0xffff: all) of the concatenation's strand comp -> the
// entry, the start position within it and the rank among ITS end positions (rmd_level0_range with the entry's length);
// false for a start in the padding between entries, too close to its entry's end for the motif, or an end position
// beyond it.  k_lo, k_n: the entries the tile's start positions fall into (DbView::tile_meta).  start_u: where the
// entry's strand begins in the concatenation's (the tile's) coordinates.
__device__ inline bool super_convert( const rmd_program_t *P, const int64_t *base_off, const int32_t *slens, long long total, int comp, int k_lo, int k_n, int szero_u, int r_u,
	int *seq, int *szero_e, int *r_e, int *start_u, int *slen_e )
{
	const int64_t	g = comp ? int64_t( total ) - 1 - szero_u : szero_u;	// the start position's base in the packed arrays
	int	lo = k_lo, hi = k_lo + k_n;		// the last entry of [ lo, hi ) that begins at or before g
	while( hi - lo > 1 ){
		const int	mid = ( lo + hi ) >> 1;
		if( base_off[ mid ] <= g )
			lo = mid;
		else
			hi = mid;
	}
	const int64_t	off = base_off[ lo ];
	const int	sl = slens[ lo ];
	if( g < off || g >= off + sl )
		return false;
	const int	sz = comp ? int( off + sl - 1 - g ) : int( g - off );
	if( sz > sl - P->dminlen )
		return false;
	*start_u = szero_u - sz;
	*r_e = 0xffff;
	if( r_u != 0xffff ){
		int	hi_u, lo_u, hi_e, lo_e;
		rmd_level0_range( P, szero_u, int( total ), &hi_u, &lo_u );
		rmd_level0_range( P, sz, sl, &hi_e, &lo_e );
		const int	end_e = hi_u - r_u - *start_u;
		if( end_e > hi_e || end_e < lo_e )
			return false;
		*r_e = hi_e - end_e;
	}
	*seq = lo;
	*szero_e = sz;
	*slen_e = sl;
	return true;
}

// The lanes with `over` set search the item ( szero, r0, cnt ) each has got, start to end (what a tile's pass
// A does with items that found room neither in the queue nor in its spill area; rare, kept out of line).
// Called by all 64 lanes.
// cc: tiles over a concatenation of entries (super_convert) -- the item is brought to its entry's coordinates first, the tile's
// bytes are seen from there (sq0, slen, seq then differ from lane to lane); cc.total == 0: not such a tiling.
struct ConcatCtx { const int64_t *base_off; const int32_t *slens; long long total; int k_lo, k_n; };
template< int BLOCK, bool CONCAT >
__device__ __forceinline__ void lean_search_here_body( const rmd_program_t *P, uint32_t *lo, uint16_t *hi, const uint8_t *sq_bytes, int sq0,
	bool over, int szero, int slen, int r0, int cnt, int seq, int comp, int32_t *hits, unsigned long long *count, long long cap, int lane_id,
	const ConcatCtx cc )
{
	LdsRecs<BLOCK>	lr{ lo, hi };
	if( CONCAT && over ){
		int	st_u = 0, re = 0;
		over = super_convert( P, cc.base_off, cc.slens, cc.total, comp, cc.k_lo, cc.k_n, szero, cnt == 1 ? r0 : 0xffff, &seq, &szero, &re, &st_u, &slen );
		sq0 -= st_u;
		if( cnt == 1 )
			r0 = re;
	}
	const rmd_seq_t	sq{ sq_bytes, sq0 };
	HitBuf	hb{};
	hb.hits = hits;
	hb.count = count;
	hb.cap = cap;
	DevSink	sink{ hb, seq, comp, P->hit_stride };
	rmd_lean_t	st;
	int	k = -1;
	if( over )
		k = rmd_lean_begin( P, lr, st, szero, slen, r0, cnt );
	while( __ballot( k >= 0 ) ){
		if( k >= 0 )
			k = rmd_lean_step<LdsRecs<BLOCK>, DevSink, rmd_seq_t, rmd_no_accel_t, true>( P, lr, st, sq, k, nullptr, sink );
		wave_emit_pending<BLOCK>( P, lr, st, k, [ & ]( int l ){ return rmd_seq_t{ sq_bytes, __shfl( sq0, l ) }; }, seq, comp, hb, lane_id );
	}
}
template< int BLOCK >
__device__ __noinline__ void lean_search_here( const rmd_program_t *P, uint32_t *lo, uint16_t *hi, const uint8_t *sq_bytes, int sq0,
	bool over, int szero, int slen, int r0, int cnt, int seq, int comp, int32_t *hits, unsigned long long *count, long long cap, int lane_id )
{
	lean_search_here_body<BLOCK, false>( P, lo, hi, sq_bytes, sq0, over, szero, slen, r0, cnt, seq, comp, hits, count, cap, lane_id, ConcatCtx{} );
}

// General path records of one lane in LDS (rmd_grec_t, 12 bytes per level): three dwords at
// w[ ( 3 * k + j ) * BLOCK ], lane-contiguous, so a wave's access is conflict free whatever
// levels its lanes are on.  Dword 0: window start | saved window end; dword 1: next end position |
// first loop variable; dword 2: second loop variable | helix length | phase.
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
		// (a copy of the loop per helix length, unrolled so that the LDS reads of all its steps are issued
		// together, measured the same: trna.descr 3.26 / 1.94 ms against 3.25 / 1.92)
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
	ConcatCtx	cc;		// (CONCAT instances: tiles over a concatenation of entries)
};

// (inlined: as a function of its own, called once per tile, it ran 15-25 % slower -- profiles/matrix_r2.sh)
#ifndef PASS_B_ATTR
#define PASS_B_ATTR	__attribute__(( always_inline ))
#endif
template< int BLOCK, int KINDS, bool CONCAT = false >
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
			RowEnds<KINDS>	ends{ pb, tile, pb_words, p_lo, ( dbg & 64 ) ? 0 : vec_words * 64 };	// (bit 64: end by end, no rows)
			// Tiles over a concatenation of entries (CONCAT): an item that is popped is brought to its entry's coordinates
			// (super_convert) and walked there -- the tile's bytes and rows seen from the entry's first base, the entry's
			// length and number; an item that belongs to no entry is dropped.  szero / r of the item at hand:
			auto	item_begin = [ & ]( unsigned item, int *szero, int *r, int *islen ) -> bool {
				*szero = z0 + int( item >> 16 );
				*r = int( item & 0xffffu );
				*islen = slen;
				if constexpr( CONCAT ){
					int	seq_e = 0, st_u = 0;
					if( !super_convert( P, gt.cc.base_off, gt.cc.slens, gt.cc.total, sink.comp, gt.cc.k_lo, gt.cc.k_n, *szero, *r, &seq_e, szero, r, &st_u, islen ) )
						return false;
					sq.sq0 = p_lo - st_u;
					ends.p_lo = p_lo - st_u;
					sink.seq = seq_e;
				}
				return true;
			};
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
								int	i_sz, r, i_sl;
								if( item_begin( cur_item, &i_sz, &r, &i_sl ) )
									k = rmd_gen_begin( P, gr, st, i_sz, i_sl, r == 0xffff ? 0 : r, r == 0xffff ? RMD_ALL_RANKS : 1 );
							}else{
								const uint32_t	*e = g_deep + i * split.entry_words();
								cur_item = e[ 0 ];
								int	i_sz, r, i_sl;
								if( item_begin( cur_item, &i_sz, &r, &i_sl ) )		// (always: it was walked to the split level from here)
									k = rmd_gen_resume( P, gr, st, sq, i_sz, i_sl,
										r == 0xffff ? 0 : r, r == 0xffff ? RMD_ALL_RANKS : 1, split_s, e + 2, int( e[ 1 ] ), ends );
							}
						}
					}
					if( __ballot( k >= 0 ) == 0 ){
						// (tiles over a concatenation: the items a round popped may all belong to no entry -- starts in the padding
						// between two entries come one after the other in the queue; only lanes that found the queue empty are done)
						if( CONCAT && __ballot( !dry ) != 0 )
							continue;
						break;
					}
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

// ---------------------------------------------------------------- pooled pass B: one item
// A lane takes pool item e (RMK_POOL_WORDS words: entry, start position, rank | strand << 16, the 3' ends
// left to the first helix of the interior for two outer lengths): its window, four bits a base, from the
// packed database into the lane's column of LDS (dwords STRIDE apart), the search state at its first
// level.  Returns that level.
template< int STRIDE, class LR >
__device__ inline int pool_item_begin( const rmd_program_t *P, const DbView &db, const unsigned *e, uint32_t *col, int nib_max,
	LR &lr, rmd_lean_t &st, rmd_nibseq_t<STRIDE> &nsq, int &seq_out, int &comp_out, int &order_base )
{
	const int	w = P->w_winsize, lm = P->lmargin, rm = P->rmargin;
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
	for( int j = 0; j < n_dw && j < nib_max; j++ ){
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
		col[ j * STRIDE ] = ( x & ~( n * 3u ) ) | ( n << 2 );	// RMA_BC_N = 4
	}
	nsq.flip = icomp ? -1 : 0;
	nsq.bias = int( off - g0 ) + ( icomp ? islen : 0 );
	seq_out = iseq;
	comp_out = icomp;
	const int	k = rmd_lean_begin( P, lr, st, szero, islen, r == 0xffff ? 0 : r, r == 0xffff ? RMD_ALL_RANKS : 1 );
	st.hmask[ 0 ] = __hip_atomic_load( e + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
	st.hmask[ 1 ] = __hip_atomic_load( e + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
	st.hm_level = ( st.hmask[ 0 ] & st.hmask[ 1 ] ) == ~0u ? -1 : P->elems[ P->searches[ 0 ] ].head_s;
	// a piece of an item (pool_sub_count): one length of the first element, one of the ends left to the first helix
	// of its interior or all of them.  Its candidates' order words start where the walk of the whole item
	// would have counted them at least: pieces in the order that walk takes them, lengths up, ends down.
	const int	piece = int( rc >> 17 ) & 3;
	order_base = 0;
	if( piece ){
		const unsigned	m = st.hmask[ piece - 1 ];
		st.only_hl = P->elems[ P->searches[ 0 ] ].minlen + piece - 1;
		order_base = ( ( piece - 1 ) * 32 + ( __popc( m ) == 1 ? 31 - ( __ffs( int( m ) ) - 1 ) : 0 ) ) << PIECE_ORDER_BITS;
	}
	return k;
}

// The pieces pool item ( rank | strand, masks h0, h1 ) is cut into for the drain kernel's list; piece p of them
// as the words 2 .. 4 of a list item.  An item whose masks say nothing (~0: no test was made) stays whole; one
// with masks goes length by length of the first element -- that length's ends one by one when they are few
// (each then is a walk of a handful of steps: the tail of the drain kernel is its longest walk, 159 steps
// over trna.descr when items stay whole).
#define POOL_SUB_ENDS	8
__device__ inline int pool_sub_count( unsigned rc, unsigned h0, unsigned h1 )
{
	if( ( h0 & h1 ) == ~0u || ( rc & 0xffffu ) == 0xffffu )
		return 1;
	const int	c0 = h0 == 0 ? 0 : ( h0 == ~0u || __popc( h0 ) > POOL_SUB_ENDS ) ? 1 : __popc( h0 );
	const int	c1 = h1 == 0 ? 0 : ( h1 == ~0u || __popc( h1 ) > POOL_SUB_ENDS ) ? 1 : __popc( h1 );
	return c0 + c1;
}
__device__ inline void pool_sub_piece( unsigned rc, unsigned h0, unsigned h1, int p, unsigned *w2, unsigned *w3, unsigned *w4 )
{
	*w2 = rc;
	*w3 = h0;
	*w4 = h1;
	if( ( h0 & h1 ) == ~0u || ( rc & 0xffffu ) == 0xffffu )
		return;
	const int	c0 = h0 == 0 ? 0 : ( h0 == ~0u || __popc( h0 ) > POOL_SUB_ENDS ) ? 1 : __popc( h0 );
	const int	i = p < c0 ? 0 : 1;
	unsigned	m = i == 0 ? h0 : h1;
	if( !( m == ~0u || __popc( m ) > POOL_SUB_ENDS ) ){
		// its ( p - first )-th end, from the highest down
		for( int q = i == 0 ? p : p - c0; q > 0; q-- )
			m &= ~( 1u << ( 31 - __clz( int( m ) ) ) );
		m = 1u << ( 31 - __clz( int( m ) ) );
	}
	*w2 = rc | ( unsigned( i + 1 ) << 17 );
	*w3 = i == 0 ? m : 0u;
	*w4 = i == 0 ? 0u : m;
}

// The drain of the pooled instance.  The search kernel leaves the items that passed its tests in ONE list
// in HBM (HitBuf::glist; a workgroup's pool is flushed there) instead of walking them itself: some twenty
// per workgroup over trna.descr and 100 Mbases, each a walk of tens of microseconds with a heavy tail -- a
// workgroup that walked its own kept the device waiting for 0.7 ms after the last tile was done
// (RNAMOTIF_DBG bit 1048576).  This kernel comes after it on the same stream: workgroups of ONE wave (no
// barrier anywhere), every wave taking its fair share of the list, then more as lanes come free.
#define DRAIN_BLOCK	64
template< int BLOCK >
__global__ void __launch_bounds__( BLOCK, DRAIN_WAVES_PER_SIMD )
rma_drain_kernel( const rmd_program_t *gP, int prog_bytes, DbView db, HitBuf hb, int n_nib, int dbg )
{
	static_assert( BLOCK == 64, "one wave per workgroup" );
	extern __shared__ __align__( 16 ) unsigned char	smem[];
	rmd_program_t	*P = reinterpret_cast<rmd_program_t *>( smem );
	const int	tid = threadIdx.x, lane_id = tid & 63;
	for( unsigned i = tid; i < unsigned( prog_bytes ) / 4; i += BLOCK )
		reinterpret_cast<uint32_t *>( P )[ i ] = reinterpret_cast<const uint32_t *>( gP )[ i ];
	__syncthreads();
	uint32_t	*const col = reinterpret_cast<uint32_t *>( smem + prog_bytes ) + tid;
	uint32_t	*const lean_lo = reinterpret_cast<uint32_t *>( smem + prog_bytes ) + n_nib * BLOCK;	// (n_nib: window dwords per lane)
	uint16_t	*const lean_hi = reinterpret_cast<uint16_t *>( lean_lo + P->n_searches * BLOCK );
	LdsRecs<BLOCK>	lr{ lean_lo + tid, lean_hi + tid };
	const unsigned long long	lt_mask = ( 1ull << lane_id ) - 1;
	const long long	reserved = ( long long )__hip_atomic_load( hb.ticket + ( RMK_GCTL - 1 ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
	const long long	total = reserved < hb.glist_cap ? reserved : hb.glist_cap;
	const long long	n_waves = ( long long )gridDim.x;
	const long long	share = ( total + n_waves - 1 ) / n_waves;
	const int	fair = share < 1 ? 1 : share > 64 ? 64 : int( share );
	const int	refill = rmd_imax( 1, rmd_imin( hb.pool_refill, fair / 2 ) );
	rmd_nibseq_t<BLOCK>	nsq{ col, 0, 0 };
	// (the lane whose column holds this lane's window: its own, or -- a subtree it was handed -- the one it came from)
	int	wlane = lane_id;
	const uint32_t	*const col0 = col - tid;
	rmd_lean_t	st;
	DevSink	sink{ hb, 0, 0, P->hit_stride };
	const rmd_no_accel_t	none;
	int	k = -1, n_steps = 0, n_emit = 0, obase = 0, floor_ = 0;
	const bool	forks = P->ord_ok && !( dbg & 4194304 );
	unsigned long long	t_item = 0;
	bool	dry = total == 0;
	// (diagnostic, RNAMOTIF_DBG bit 536870912: when the waves are through, in bins of 16 us from the first wave's start)
	if( ( dbg & 536870912 ) && lane_id == 0 )
		atomicMax( hb.ticket + 55, ~( unsigned long long )wall_clock64() );
	for( ; ; ){
		const unsigned long long	want = __ballot( k < 0 && !dry );
		const unsigned long long	busy = __ballot( k >= 0 );
		// lanes that came free take items once enough of them have (a round costs the wave the same for one
		// lane as for sixteen), and no more than leaves the other waves their share
		const int	n = rmd_imin( __popcll( want ), fair - __popcll( busy ) );
		// (diagnostic, RNAMOTIF_DBG bit 32: the wave's cycles by what it does -- taking items, stepping, complete matches, hand-overs)
		unsigned long long	t_d0 = ( dbg & 32 ) ? __builtin_amdgcn_s_memtime() : 0;
#define DRAIN_LAP( slot_ )	do{ if( dbg & 32 ){ \
			const unsigned long long	now_ = __builtin_amdgcn_s_memtime(); \
			if( lane_id == 0 ) \
				atomicAdd( hb.ticket + ( slot_ ), now_ - t_d0 ); \
			t_d0 = now_; \
		} }while( 0 )
		if( n > 0 && ( busy == 0 || n >= refill ) ){
			unsigned long long	base = 0;
			if( lane_id == __ffsll( want ) - 1 )
				base = atomicAdd( hb.ticket + RMK_GCTL, ( unsigned long long )n );
			base = __shfl( base, __ffsll( want ) - 1 );
			if( k < 0 && !dry ){
				const long long	i = ( long long )base + __popcll( want & lt_mask );
				if( i < ( long long )base + n && i < total ){
					const unsigned	*e = hb.pool + RMK_POOL_WORDS * size_t( i );
					// (a workgroup whose items found no room in the list left its share of it void)
					if( __hip_atomic_load( e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) != 0xffffffffu ){
						k = pool_item_begin<BLOCK>( P, db, e, col, n_nib, lr, st, nsq, sink.seq, sink.comp, obase );
						nsq.w = col;
						wlane = lane_id;
						floor_ = 0;
						if( dbg & 268435456 )		// (ablation: the item's window is laid out and the item dropped)
							k = -1;
						t_item = ( dbg & 32 ) ? __builtin_amdgcn_s_memtime() : 0;
						n_steps = n_emit = 0;
					}
				}else if( ( long long )base + n >= total )
					dry = true;
			}
			DRAIN_LAP( 94 );
			continue;
		}
		if( busy == 0 )
			break;
		const bool	was = k >= 0;
		const int	k_was = k;
		if( ( dbg & 32 ) && lane_id == 0 ){
			atomicAdd( hb.ticket + 98, 1ull );
			atomicAdd( hb.ticket + 93, ( unsigned long long )__popcll( busy ) );
		}
		if( k >= 0 ){
			k = rmd_lean_step<LdsRecs<BLOCK>, DevSink, rmd_nibseq_t<BLOCK>, rmd_no_accel_t, true>( P, lr, st, nsq, k, nullptr, sink, none );
			n_steps++;
			n_emit += st.pending;
		}
		DRAIN_LAP( 95 );
		wave_emit_pending<BLOCK, true>( P, lr, st, k, [ & ]( int l ){
			return rmd_nibseq_t<BLOCK>{ col0 + __shfl( wlane, l ), __shfl( nsq.flip, l ), __shfl( nsq.bias, l ) }; },
			sink.seq, sink.comp, hb, lane_id, obase );
		DRAIN_LAP( 96 );
		if( k >= 0 && k < floor_ )
			k = -1;		// (the subtree this lane was given is done)
		if( forks ){
			// Lanes with nothing left to take are given subtrees: a lane that has just gone down a level hands
			// that level's walk -- its records, window and state, copied -- to an idle lane and goes on with the
			// alternatives of its own level, as if the subtree had been walked.  (The candidates' order words do
			// not depend on who finds them when: rmd_elem_t::ord_stride.)  The longest walk of trna.descr's
			// items, 73 steps one after the other, is what the drain kernel took its time from.
			const unsigned long long	idle = __ballot( k < 0 && dry );
			const unsigned long long	down = __ballot( k >= 0 && k == k_was + 1 );
			const int	n_f = rmd_imin( __popcll( idle ), __popcll( down ) );
			if( n_f > 0 ){
				const bool	taker = k < 0 && dry && __popcll( idle & lt_mask ) < n_f;
				const bool	giver = k >= 0 && k == k_was + 1 && __popcll( down & lt_mask ) < n_f;
				int	src = lane_id;
				if( taker ){
					unsigned long long	m = down;
					for( int q = __popcll( idle & lt_mask ); q > 0; q-- )
						m &= m - 1;
					src = __ffsll( m ) - 1;
				}
				// (every lane takes part in the exchanges; a lane that is not a taker reads its own values)
				const int	k_src = __shfl( k, src );
				st.szero = __shfl( st.szero, src );
				st.slen = __shfl( st.slen, src );
				st.hi0 = __shfl( st.hi0, src );
				st.lo0 = __shfl( st.lo0, src );
				st.rank = __shfl( st.rank, src );
				st.order = __shfl( st.order, src );
				st.pretested = __shfl( st.pretested, src );
				st.hm_level = __shfl( st.hm_level, src );
				st.hmask[ 0 ] = __shfl( st.hmask[ 0 ], src );
				st.hmask[ 1 ] = __shfl( st.hmask[ 1 ], src );
				st.only_hl = __shfl( st.only_hl, src );
				nsq.flip = __shfl( nsq.flip, src );
				nsq.bias = __shfl( nsq.bias, src );
				sink.seq = __shfl( sink.seq, src );
				sink.comp = __shfl( sink.comp, src );
				obase = __shfl( obase, src );
				// The window is not copied: the taker reads it where it lies.  (A lane is idle for good -- `dry` -- only once the list's
				// counter has passed its end, and from then on no lane of any wave begins an item: the columns stay as they are.)
				wlane = __shfl( wlane, src );
				nsq.w = col0 + wlane;
				if( taker ){
					// the records, four levels' loads ahead of their stores (the two may be the same LDS for all the compiler knows)
					const uint32_t	*lo_s = lr.lo + ( src - lane_id );
					const uint16_t	*hi_s = lr.hi + ( src - lane_id );
					const int	ns = P->n_searches;
					for( int j = 0; j < ns; j += 4 ){
						uint32_t	a[ 4 ];
						uint16_t	b[ 4 ];
						for( int q = 0; q < 4; q++ ){
							const int	jq = rmd_imin( j + q, ns - 1 );
							a[ q ] = lo_s[ jq * BLOCK ];
							b[ q ] = hi_s[ jq * BLOCK ];
						}
						for( int q = 0; q < 4; q++ )
							if( j + q < ns ){
								lr.lo[ ( j + q ) * BLOCK ] = a[ q ];
								lr.hi[ ( j + q ) * BLOCK ] = b[ q ];
							}
					}
					st.pending = 0;
					k = k_src;
					floor_ = k_src;
					t_item = ( dbg & 32 ) ? __builtin_amdgcn_s_memtime() : 0;
					n_steps = n_emit = 0;
				}
				if( giver )
					k = k_was;	// (back at its own level, the subtree below as good as walked)
			}
		}
		DRAIN_LAP( 97 );
		if( ( dbg & 32 ) && was && k < 0 ){
			// (diagnostic: how long the items take, how many steps, how many complete matches)
			const unsigned long long	dt = __builtin_amdgcn_s_memtime() - t_item;
			atomicAdd( hb.ticket + 23 + ( 63 - __clzll( dt | 1ull ) ), 1ull );
			atomicMax( hb.ticket + 21, dt );
			atomicAdd( hb.ticket + 20, dt );
			atomicAdd( hb.ticket + 17, 1ull );
			atomicAdd( hb.ticket + 18, ( unsigned long long )n_steps );
			atomicMax( hb.ticket + 22, ( unsigned long long )n_steps );
			atomicAdd( hb.ticket + 60 + rmd_imin( 31 - __clz( n_emit | 1 ) + ( n_emit > 0 ), 15 ), 1ull );
			atomicAdd( hb.ticket + 76 + rmd_imin( 31 - __clz( n_emit | 1 ) + ( n_emit > 0 ), 15 ), dt );
		}
	}
#undef DRAIN_LAP
	if( ( dbg & 536870912 ) && lane_id == 0 ){
		const unsigned long long	t0 = ~__hip_atomic_load( hb.ticket + 55, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
		const unsigned long long	dt = wall_clock64() - t0;		// (100 MHz)
		atomicAdd( hb.ticket + 23 + rmd_imin( int( dt / 1600 ), 31 ), 1ull );
	}
}

// ---------------------------------------------------------------- search kernel
// LEAN: the descriptor has only ss and proper helices (rmd_program_t::lean_ok) -- pass B keeps
// 8 bytes of state per level in LDS; the general state machine is not compiled into that
// instance at all (no scratch frames, fewer registers).
// (the general instance is bound by the latency of its dependent LDS accesses: four workgroups per
// CU where its records, 12 bytes per level and lane, leave room for them -- measured against three
// at 168 registers and against out-of-line level generators, profiles/matrix_r2.sh)
// (... three with 168 registers for descriptors with triplexes / 4-plexes: qu+tr 46.4 -> 39.2 ms, where
// pk1 goes 7.5 -> 8.6 ms)
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
// CONCAT (pooled instance): the tiles lie over the concatenation of the entries (DbView::concat_bases, super_convert) -- an
// instance of its own, so that the instance for databases of long entries is, to the register, what it was without it
// (the headline kernel sits at its 128 registers: a dozen more live values in pass A' cost it a quarter of its speed).
// WALK = false (pooled instance): the search kernel walks nothing.  What passes pass A' goes straight to the drain kernel's list,
// cut into its pieces, a wave reserving its items' room with one atomic; queue overflow and a full list are REPORTED
// (ticket[ RMK_GCTL + 1 ], ticket[ RMK_GCTL - 1 ] > glist_cap) and the host repeats the scan with larger areas, as the general
// instances do -- no pool, no walk, no in-place search in the kernel: fewer registers, fewer barriers a tile.
template< int BLOCK, bool LEAN, int G, int KINDS = 0, bool POOL = false, bool CONCAT = false, bool WALK = true >
__global__ void __launch_bounds__( BLOCK, LEAN ? ( WALK ? SEARCH_WAVES_PER_SIMD : FLUSH_WAVES_PER_SIMD ) : GENERAL_WAVES( KINDS ) )
rma_search_kernel( const rmd_program_t *gP, int prog_bytes, int qcap, DbView db, HitBuf hb, int tile_bytes, int dbg )
{
	static_assert( G == 1 || ( LEAN && G % ( BLOCK / 64 ) == 0 && G <= 32 ), "tile groups: lean path, whole rounds of waves" );
	static_assert( !POOL || ( LEAN && G == 1 ), "pooled pass B: lean path, one tile per pass" );
	static_assert( !CONCAT || POOL || !LEAN, "tiles over a concatenation of entries: the pooled lean instance and the general ones" );
	static_assert( WALK || POOL, "a search kernel that walks nothing: the pooled lean instance" );
	extern __shared__ __align__( 16 ) unsigned char	smem[];
	rmd_program_t	*P = reinterpret_cast<rmd_program_t *>( smem );
	// gP is the compact image (rmd_make_image): prog_bytes of it, a multiple of 16
	unsigned	*queue = reinterpret_cast<unsigned *>( smem + prog_bytes );
	uint8_t	*const tile0 = smem + prog_bytes + qcap * sizeof( unsigned );
	const int	slot_bytes = ( tile_bytes + 15 ) & ~15;
	__shared__ long long	s_tile;
	__shared__ int	s_seq, s_qn, s_qhead, s_dqn, s_dqhead;
	__shared__ int	s_pool_n, s_pool_head, s_glist_full, s_fl_total, s_fl_pos;
	__shared__ long long	s_gstart;
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
		s_glist_full = 0;
		s_fl_total = 0;
		s_fl_pos = 0;
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
	// (a 4-plex at the head of the search list: four more, rmd_q1filter_t)
	const bool	q1f_vecs = !LEAN && ( KINDS & RMD_KIND_TQ ) != 0 && P->q1f.on;
	const bool	q1f = q1f_vecs && !( dbg & 8192 );
	// (lean, one tile per pass, with a look-ahead chain: eight more, rmd_chain_t)
	const bool	chain_on = LEAN && G == 1 && P->chain.on;
	// (tiles over the concatenation of the entries: super_convert)
	constexpr bool	concat = CONCAT;
	// (a vector of the start positions worth a look: what the look-ahead chain leaves, or where the best literal is within reach)
	const bool	sv_on = ( LEAN && G == 1 && ( P->chain.on || P->lit_re >= 0 ) ) || ( !LEAN && P->lit_re >= 0 );
	// (... and, last, where each base stands -- five vectors -- when there is a best literal to look for)
	const int	n_vec = 1 + 5 * n_rs + ( !LEAN && ( KINDS & RMD_KIND_TQ ) != 0 && P->q1f.on ? ( P->q1f.t_on ? 9 : 4 ) : 0 ) + ( sv_on ? 1 : 0 ) +
		( P->lit_re >= 0 && G == 1 ? 5 : 0 );
	unsigned long long	*const pb0 = reinterpret_cast<unsigned long long *>( tile0 + size_t( G ) * slot_bytes );
	uint32_t	*lean_lo = reinterpret_cast<uint32_t *>( pb0 + size_t( G ) * n_vec * pb_words );
	uint16_t	*lean_hi = reinterpret_cast<uint16_t *>( lean_lo + P->n_searches * BLOCK );
	// (groups of small tiles: those five vectors live only from a slot's decode to its literal vector -- one set per
	// wave, behind the records, not one per slot: sixteen sets cost ire.descr its tiles of 768 positions)
	unsigned long long	*const lit_scratch = reinterpret_cast<unsigned long long *>(
		( reinterpret_cast<uintptr_t>( lean_hi + P->n_searches * BLOCK ) + 7 ) & ~uintptr_t( 7 ) );
	// the chain's working vectors -- where each base stands (5), two groups' vectors, a stem-loop's cores --
	// take the place of the search records, which are not in use before pass B
	unsigned long long	*const tv = reinterpret_cast<unsigned long long *>( lean_lo );
	// (the instance that walks nothing has no records: the host gave the vectors their own room)
	const bool	chain_vecs = chain_on && ( !WALK || size_t( P->n_searches ) * BLOCK * LEAN_REC_BYTES >= size_t( 10 ) * pb_words * sizeof( unsigned long long ) );
	__shared__ int	s_inplace;		// this tile's pre-filter searched queue overflow in place: the records, and with them tv, were written
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
	// (the ticket of the tile after this one is asked for while this one is worked on: the one counter all
	// workgroups share answers in microseconds, and nobody should wait for it with a tile in hand)
	// One tile per pass: the tickets run TWO tiles ahead, so that the tile's line of DbView::tile_meta -- entry, strand,
	// first start position, length, place in the packed arrays: all the kernel asks of a tile -- is fetched a tile ahead,
	// by two lanes, straight into LDS (no register holds it meanwhile), two buffers in turns.  Until round 4 a tile began
	// with three dependent loads (tile_seq[ t ]; the entry's slen / base_off / tile_start; the packed words): the phase
	// counters had 42 % of the headline kernel's wave cycles in its "decode" phase, most of it those round trips.
	__shared__ __align__( 16 ) int	s_meta[ 2 ][ RMK_META_WORDS ];
	__shared__ long long	s_tile_nx;
	long long	t_next = 0, t_next2 = 0;
	// One tile per pass: a ticket is good for TICKET_TILES tiles in a row, and only the last TICKET_TAIL tiles per workgroup go
	// one by one, to even out the end.  Every ticket is an atomic on ONE word that all workgroups of all XCDs share; 25 000 of
	// them (trna.descr, 100 Mbase) took the device 0.35 ms to serve, one after the other -- what profiles/flush_stages.py showed
	// as the kernel's "decode" stage (0.346 ms with nothing but decode in it, 0.14 with the tiles dealt out in advance; that,
	// though, leaves a workgroup that starts late -- behind another scanner's kernels -- with all its tiles still to do).
	const long long	n_big = ( G == 1 && !( dbg & 67108864 ) && n_units > ( long long )TICKET_TAIL * gridDim.x ) ?
		( n_units - ( long long )TICKET_TAIL * gridDim.x ) / TICKET_TILES : 0ll;
	long long	t_run = 0, t_run_end = 0;
	auto	take_tile = [ & ]() -> long long {
		if( t_run < t_run_end )
			return t_run++;
		const long long	c = ( long long )atomicAdd( hb.ticket, 1ull );
		if( c >= n_big )
			return n_big * TICKET_TILES + ( c - n_big );
		t_run = c * TICKET_TILES + 1;
		t_run_end = ( c + 1 ) * TICKET_TILES;
		return c * TICKET_TILES;
	};
	if( tid == 0 ){
		t_next = take_tile();
		if constexpr( G == 1 ){
			t_next2 = take_tile();
			if( t_next < db.n_tiles )
				for( int k = 0; k < RMK_META_WORDS; k++ )
					s_meta[ 0 ][ k ] = db.tile_meta[ t_next * RMK_META_WORDS + k ];
		}
	}
	bool	had_tiles = false;
	if( ( dbg & 1048576 ) && tid == 0 )
		atomicMax( hb.ticket + 88, ~( unsigned long long )wall_clock64() );	// (the first workgroup's start)
	for( int pass = 0; ; pass++ ){
		if( tid == 0 ){
			const long long	t = t_next;
			if constexpr( G == 1 ){
				t_next = t_next2;
				if( t < n_units )
					t_next2 = take_tile();
				s_tile_nx = t_next;
			}else{
				if( t < n_units )
					t_next = take_tile();
			}
			s_tile = t;
			s_seq = 0;
			s_qn = 0;
			s_qhead = 0;
			s_dqn = 0;
			s_dqhead = 0;
		}
		__syncthreads();
		const long long	t = s_tile;
		if constexpr( G == 1 ){
			// the next tile's line, on its way into the other buffer (read at the next pass, many barriers from here)
			const long long	tn = s_tile_nx;
			if( tn < db.n_tiles && tid < RMK_META_WORDS / 4 )
				__builtin_amdgcn_global_load_lds( ( const __attribute__(( address_space( 1 ) )) void * )( db.tile_meta + tn * RMK_META_WORDS + tid * 4 ),
					( __attribute__(( address_space( 3 ) )) void * )( &s_meta[ ( pass + 1 ) & 1 ][ 0 ] ), 16, 0, 0 );
		}
		bool	last = false;
		if( t >= n_units ){
			// (pooled: one more round, over a tile without start positions, for what the pool still holds)
			if constexpr( POOL && WALK )
				last = true;
			else
				break;
			if( ( dbg & 1048576 ) && tid == 0 && had_tiles ){
				atomicAdd( hb.ticket + 89, ( unsigned long long )wall_clock64() );	// (out of tiles)
				atomicAdd( hb.ticket + 87, 1ull );
			}
		}else
			had_tiles = true;
		// what pass B needs of the tile (G > 1: of the last slot; pass B reloads per item)
		int	seq = 0, slen = 0, z0 = 0, p_lo = 0, vec_words = 0;
		int	ent_n = 0;	// (tiles over a concatenation: the entries from `seq` on that the tile's start positions fall into)
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
		int64_t	off = 0;
		int	comp = 0, pos_hi = 0x7fffffff;
		if constexpr( G == 1 ){
			// (the tile's line, fetched a pass ago)
			const int	*const m = s_meta[ pass & 1 ];
			ent_n = live ? m[ RMK_META_PAD ] : 0;
			seq = live ? m[ RMK_META_SEQ ] : 0;
			comp = live ? m[ RMK_META_COMP ] : 0;
			z0 = live ? m[ RMK_META_Z0 ] : 0;
			slen = live ? m[ RMK_META_SLEN ] : 0;
			off = live ? int64_t( ( uint64_t( uint32_t( m[ RMK_META_OFF_HI ] ) ) << 32 ) | uint32_t( m[ RMK_META_OFF_LO ] ) ) : 0;
			pos_hi = live ? m[ RMK_META_POS_HI ] : 0;
		}else{
			seq = live ? db.tile_seq[ tt ] : 0;
			slen = db.slen[ seq ];
		}
		tile = tile0 + size_t( slot ) * slot_bytes;
		unsigned long long	*const occ = pb0 + size_t( slot ) * n_vec * pb_words;	// where the best literal occurs (bit per start)
		pb = occ + pb_words;		// row set 0
		unsigned long long	*const xv_of_slot = pb + 5 * n_rs * pb_words;
		unsigned long long	*const lvp = G > 1 ? lit_scratch + size_t( tid >> 6 ) * 6 * pb_words : occ + size_t( n_vec - 5 ) * pb_words;	// (only with a literal)
		unsigned long long	*const lsv = G > 1 ? lvp + 5 * pb_words : xv_of_slot;		// where the literal's start positions go (G > 1: the wave's sixth vector)
		if constexpr( G > 1 ){
			off = db.base_off[ seq ];
			const int	local = live ? int( tt - db.tile_start[ seq ] ) : 0;
			const int	per_strand = live ? int( ( db.tile_start[ seq + 1 ] - db.tile_start[ seq ] ) / db.strands ) : 1;
			comp = local / per_strand;
			const int	pos_lo = db.pos_lo ? db.pos_lo[ seq ] : 0;
			pos_hi = db.pos_hi ? db.pos_hi[ seq ] : 0x7fffffff;
			z0 = pos_lo + ( local % per_strand ) * T;
		}

		// decode the bases this tile can touch: [ z0 - lm, z0 + T + w - 1 + rm )
		p_lo = z0 - lm;
		int	p_from = p_lo < 0 ? 0 : p_lo;
		int	p_to = z0 + T + w - 1 + rm;
		if( p_to > slen )
			p_to = slen;
		if( !live )
			p_to = p_from;		// slot past the last tile: nothing to decode, no start position
		// Tile position 0 is moved back (by less than 32) to where a group of 32 bases of the packed database
		// begins -- counted from the entry's start for strand 0, from its end for strand 1, the same words
		// read backwards and complemented (mk_rcmp, rnamot.c:193) -- so that one lane turns one such group
		// into 32 positions of everything the tile holds: the byte per base, the bit vectors of where each
		// base stands and, ORed from those by the pair table, the pair rows.  (Until round 3 a lane unpacked 16
		// bases byte by byte and the rows were made by ten wave ballots per 64 positions: 0.78 of trna.descr's
		// 2.4 ms.)
		p_lo -= comp ? ( ( ( p_lo - slen ) % 32 ) + 32 ) % 32 : ( ( p_lo % 32 ) + 32 ) % 32;
		vec_words = rmd_imin( pb_words, ( p_to - p_lo + 64 + 63 ) / 64 + 1 );	// bit vector words in use
		{
			const int	n_dw = ( p_to - p_lo + 31 ) / 32;
			uint32_t	*const tile32 = reinterpret_cast<uint32_t *>( tile );
			auto	even16 = []( uint32_t x ) -> uint32_t {		// bits 0, 2, 4 ... of x side by side
				x &= 0x55555555u;
				x = ( x | ( x >> 1 ) ) & 0x33333333u;
				x = ( x | ( x >> 2 ) ) & 0x0f0f0f0fu;
				x = ( x | ( x >> 4 ) ) & 0x00ff00ffu;
				return ( x | ( x >> 8 ) ) & 0x0000ffffu;
			};
			auto	spread4 = []( uint32_t x ) -> uint32_t {	// bits 0 .. 3 of x to the low bits of four bytes
				return __umul24( x & 15u, 0x00204081u ) & 0x01010101u;
			};
			for( int dd = utid; dd < vec_words * 2; dd += UNIT ){
				const int	d = dd - 2;		// (64 pad bits in front of every vector)
				uint32_t	b0 = 0, b1 = 0, nn = 0, valid = 0;
				if( d >= 0 && d < n_dw ){
					const int	q0 = 32 * d;
					const int	fa = comp ? slen - 1 - ( p_lo + q0 ) : p_lo + q0;	// forward position of tile position q0
					const int	f0 = comp ? fa - 31 : fa;				// ... of the group's first base
					if( f0 >= 0 && f0 < slen ){
						const int64_t	g = ( off + f0 ) >> 5;
						const uint32_t	cw0 = db.codes[ 2 * g ], cw1 = db.codes[ 2 * g + 1 ];
						nn = db.amask[ g ];
						b0 = even16( cw0 ) | ( even16( cw1 ) << 16 );
						b1 = even16( cw0 >> 1 ) | ( even16( cw1 >> 1 ) << 16 );
						if( comp ){
							b0 = ~__brev( b0 );
							b1 = ~__brev( b1 );
							nn = __brev( nn );
						}
					}
					const int	v_lo = rmd_imax( 0, p_from - ( p_lo + q0 ) ), v_hi = rmd_imin( 32, p_to - ( p_lo + q0 ) );
					if( v_hi > v_lo )
						valid = ( v_hi - v_lo >= 32 ? ~0u : ( ( 1u << ( v_hi - v_lo ) ) - 1u ) ) << v_lo;
					// the byte per base: 0 .. 3, 4 for a letter that is not acgt, 7 outside the entry
					for( int k = 0; k < 8; k++ ){
						const uint32_t	s0 = spread4( b0 >> ( 4 * k ) ), s1 = spread4( b1 >> ( 4 * k ) );
						const uint32_t	sn = spread4( nn >> ( 4 * k ) ), sv = spread4( valid >> ( 4 * k ) );
						uint32_t	by = ( ( s0 | ( s1 << 1 ) ) & ~( sn * 3u ) ) | ( sn << 2 );
						by = ( by & ( sv * 7u ) ) | ( ( sv ^ 0x01010101u ) * 7u );
						tile32[ 8 * d + k ] = by;
					}
				}
				// where each base stands
				const uint32_t	acgt = valid & ~nn;
				const uint32_t	is[ 5 ] = { acgt & ~b1 & ~b0, acgt & ~b1 & b0, acgt & b1 & ~b0, acgt & b1 & b0, valid & nn };
				// pair rows: rows[ b ] has a bit per tile position that can pair with 5' base b -- for the
				// pre-filter's windows and for the helices of the search itself (RowEnds)
				if( LEAN ? bitpar : n_rs > 0 )
					for( int rs = 0; rs < n_rs; rs++ ){
						const unsigned	mat2 = rmd_pairsets( P )[ P->rowset_ps[ rs ] ].mat2;
						uint32_t	*const rows = reinterpret_cast<uint32_t *>( pb + 5 * rs * pb_words );
						for( int b5 = 0; b5 < 5; b5++ ){
							uint32_t	m = 0;
							for( int c = 0; c < 5; c++ )
								m |= is[ c ] & ( 0u - ( ( mat2 >> ( b5 * 5 + c ) ) & 1u ) );
							rows[ b5 * pb_words * 2 + dd ] = m;
						}
					}
				if( chain_vecs )
					for( int b5 = 0; b5 < 5; b5++ )
						reinterpret_cast<uint32_t *>( tv + b5 * pb_words )[ dd ] = is[ b5 ];
				if( P->lit_re >= 0 )
					for( int b5 = 0; b5 < 5; b5++ )
						reinterpret_cast<uint32_t *>( lvp + b5 * pb_words )[ dd ] = is[ b5 ];
				if( q1f_vecs ){
					// strand filter of a leading 4-plex (rmd_q1filter_t): where a base stands that some quad has in
					// second / third place, and the same for the triples of a triplex that follows
					const rmd_q1filter_t	&F = P->q1f;
					uint32_t	*const x32 = reinterpret_cast<uint32_t *>( pb + 5 * n_rs * pb_words );
					auto	any_of = [ & ]( int mask ) -> uint32_t {
						uint32_t	m = 0;
						for( int c = 0; c < 5; c++ )
							m |= is[ c ] & ( 0u - ( ( unsigned( mask ) >> c ) & 1u ) );
						return m;
					};
					x32[ dd ] = any_of( F.m2 );
					x32[ pb_words * 2 + dd ] = any_of( F.m3 );
					if( F.t_on ){
						x32[ 4 * pb_words * 2 + dd ] = any_of( F.tm1 );
						x32[ 5 * pb_words * 2 + dd ] = any_of( F.tm2 );
						x32[ 6 * pb_words * 2 + dd ] = any_of( F.tm3 );
					}
				}
			}
		}
		SLOT_SYNC();
		PHASE( 0 );
		PHASE( 2 );
		// short entries fill only part of a tile: the loops below run over what is there
		const int	pos_end = rmd_imin( slen - P->dminlen + 1, pos_hi );
		const int	n_pos = live ? rmd_imin( T, pos_end - z0 ) : 0;		// start positions of this tile
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
		// (Round 3: a lane per 64 positions -- the bases each state of the literal takes, as the OR of their
		// "stands here" vectors, shifted into place and ANDed -- where rounds 1 and 2 had a lane per position
		// read the literal's bases one by one: a quarter of pk1.descr's wave cycles.  Positions outside the
		// entry have no bit in any vector.)
		if( lit ){
			const rmd_regex_t	&lre = rmd_regexes( P )[ P->lit_re ];
			const unsigned long long	*const lv = lvp;
			for( int wi = utid; wi < vec_words; wi += UNIT ){
				unsigned long long	acc = ~0ull;
				for( int jj = 0; jj < lit_n; jj++ ){
					unsigned long long	lo = 0, hi = 0;
					for( int c = 0; c < 5; c++ )
						if( ( lre.accept[ c ] >> jj ) & 1 ){
							lo |= lv[ c * pb_words + wi ];
							hi |= wi + 1 < pb_words ? lv[ c * pb_words + wi + 1 ] : 0ull;
						}
					acc &= jj ? ( lo >> jj ) | ( hi << ( 64 - jj ) ) : lo;
				}
				occ[ wi ] = acc;
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
		// LIT_OK for the 64 start positions from bit x of the vectors on, or more (undecided near the vectors' end: kept)
		auto	lit_starts = [ & ]( int x ) -> unsigned long long {
			if( !lit )
				return ~0ull;
			const int	n = lit_hi - P->lit_lo + 1;
			unsigned long long	r = 0;
			for( int k = 0; k < n && ~r; k += 64 ){
				const int	idx = x + P->lit_lo + k, m = rmd_imin( 64, n - k );
				if( idx < 0 || ( idx >> 6 ) + 2 >= pb_words )
					return ~0ull;
				unsigned long long	lo = bits64( occ, idx ), hi = bits64( occ, idx + 64 );
				// (bit i: any of the bits i .. i + m - 1 of hi:lo)
				for( int have = 1; have < m; ){
					const int	st = rmd_imin( have, m - have );
					lo |= ( lo >> st ) | ( hi << ( 64 - st ) );
					hi |= hi >> st;
					have += st;
				}
				r |= lo;
			}
			return r;
		};

		// push( pred, item ): one ballot + prefix count per call; overflow is searched in place
#define QPUSH( pred, item, szero_, r0_, cnt_ )	do{ \
		const unsigned long long	m_ = __ballot( pred ); \
		if( m_ ){ \
			bool	over_ = false; \
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
				else \
					over_ = true; \
				/* (general instance: the launch is repeated with a spill area that holds the tile's items) */ \
			} \
			if constexpr( LEAN && WALK ){ \
				if( __ballot( over_ ) ){ \
					/* items beyond the queue and its spill area are searched here and now */ \
					s_inplace = 1; \
					if constexpr( CONCAT ){ \
						/* (tiles over a concatenation: the item in its entry's coordinates, the tile's bytes seen from there) */ \
						int	o_sq0_ = sq.sq0, o_sz_ = szero_, o_sl_ = slen, o_r_ = r0_, o_seq_ = sink.seq; \
						if( over_ ){ \
							int	st_u_ = 0, re_ = 0; \
							over_ = super_convert( P, db.base_off, db.slen, db.concat_bases, sink.comp, seq, ent_n, szero_, ( cnt_ ) == 1 ? ( r0_ ) : 0xffff, &o_seq_, &o_sz_, &re_, &st_u_, &o_sl_ ); \
							o_sq0_ = sq.sq0 - st_u_; \
							o_r_ = ( cnt_ ) == 1 ? re_ : ( r0_ ); \
						} \
						lean_search_here<BLOCK>( P, lr.lo, lr.hi, sq.sq, o_sq0_, over_, o_sz_, o_sl_, o_r_, cnt_, o_seq_, sink.comp, hb.hits, hb.count, hb.cap, lane_id ); \
					}else \
						lean_search_here<BLOCK>( P, lr.lo, lr.hi, sq.sq, sq.sq0, over_, szero_, slen, r0_, cnt_, sink.seq, sink.comp, hb.hits, hb.count, hb.cap, lane_id ); \
				} \
			} \
		} }while( 0 )

		if constexpr( G == 1 ){
			if( dbg & 65536 )		// (ablation: the tile is decoded and its rows are built; nothing is queued)
				continue;
		}
		// Strand filter of a leading 4-plex (rmd_q1filter_t): G2 / G3 -- where a base stands that some quad
		// has in second / third place -- then A -- start positions from which a second strand can be reached --
		// and B -- end positions before which a third strand can stand.  Bit q + 64 of a vector is tile position q.
		unsigned long long	*const xv = pb + 5 * n_rs * pb_words;
		if( q1f ){
			const rmd_q1filter_t	F = P->q1f;
			const bool	tri = F.t_on && !( dbg & 16384 );
			const int	n_valid = p_to - p_lo, vec_bits = vec_words * 64;
			// 64 positions from bit x on: can a strand stand there?  dir +1: read from its start onwards;
			// -1: from its end backwards.  Bits the vectors do not hold: undecided, kept.
			auto	stand = [ & ]( const unsigned long long *gv, int x, int dir, int nmin, int badmax, bool first5 ) -> unsigned long long {
				if( x - nmin < 0 || x + nmin + 96 > vec_bits )
					return ~0ull;
				unsigned long long	c1 = 0, c2 = 0, c3 = 0, c4 = 0;
				for( int q = 1; q < nmin; q++ ){
					const unsigned long long	mis = ~bits64( gv, x + dir * q );
					c4 |= c3 & mis;
					c3 |= c2 & mis;
					c2 |= c1 & mis;
					c1 |= mis;
				}
				unsigned long long	ok = ~( badmax == 0 ? c1 : badmax == 1 ? c2 : badmax == 2 ? c3 : c4 );
				if( first5 )
					ok &= bits64( gv, x );
				return ok;
			};
			// 64 bits of a finished vector from bit x on (outside it: undecided)
			auto	peek = [ & ]( const unsigned long long *v, int x ) -> unsigned long long {
				return x >= 0 && x + 96 <= vec_bits ? bits64( v, x ) : ~0ull;
			};
			for( int wi = utid; wi < vec_words; wi += UNIT ){
				const int	x = wi * 64;
				unsigned long long	a = 0, b = 0;
				for( int d = F.a_lo; d <= F.a_hi && ~a; d++ )
					a |= stand( xv, x + d, 1, F.nmin, F.badmax, F.first5 );
				for( int d = F.b_hi; d <= F.b_lo && ~b; d++ )
					b |= stand( xv + pb_words, x - d, -1, F.nmin, F.badmax, F.first5 );
				xv[ 2 * pb_words + wi ] = a;
				xv[ 3 * pb_words + wi ] = b;
				if( tri ){
					// ends of a second strand of the triplex with a third one within reach behind them
					unsigned long long	r2 = 0;
					for( int d = F.r2_lo; d <= F.r2_hi && ~r2; d++ )
						r2 |= stand( xv + 6 * pb_words, x + d, 1, F.tnmin, F.tbad, F.tfirst5 );
					xv[ 7 * pb_words + wi ] = r2 & stand( xv + 5 * pb_words, x, -1, F.tnmin, F.tbad, F.tfirst5 );
				}
			}
			SLOT_SYNC();
			if( tri ){
				// starts of a first strand with such a second one within reach ...
				for( int wi = utid; wi < vec_words; wi += UNIT ){
					const int	x = wi * 64;
					unsigned long long	r1 = 0;
					for( int d = F.r1_lo; d <= F.r1_hi && ~r1; d++ )
						r1 |= peek( xv + 7 * pb_words, x + d );
					xv[ 8 * pb_words + wi ] = r1 & stand( xv + 4 * pb_words, x, 1, F.tnmin, F.tbad, F.tfirst5 );
				}
				SLOT_SYNC();
				// ... and the end positions of the 4-plex' group that have one of those behind them
				for( int wi = utid; wi < vec_words; wi += UNIT ){
					const int	x = wi * 64;
					unsigned long long	f = 0;
					for( int d = F.f_lo; d <= F.f_hi && ~f; d++ )
						f |= peek( xv + 8 * pb_words, x + d );
					xv[ 3 * pb_words + wi ] &= f;
				}
				SLOT_SYNC();
			}
		}
		// Look-ahead chain of the first element's interior (rmd_chain_t): start positions at which the
		// interior's stem-loops cannot all be there are not searched.  tv[ 0 .. 4 ]: where each base stands;
		// tv[ 5 ], tv[ 6 ]: a group's vector and the next group's, in turns; tv[ 7 ], tv[ 8 ]: the cores of the
		// first two shapes of stem-loop (kept for pass A'), tv[ 9 ]: those of any other; xv[ 0 ]: the start
		// positions that remain.
		const bool	chain = chain_vecs && bitpar && !( dbg & 32768 );
		if( chain ){
			const rmd_chain_t	&C = P->chain;
			const int	vec_bits = vec_words * 64;
			const unsigned	mat2 = rmd_pairsets( P )[ P->rowset_ps[ 0 ] ].mat2;
			const int	nb = ( ( mat2 >> 20 ) & 31u ) || ( mat2 & 0x0108421u << 4 ) ? 5 : 4;		// (n pairs with nothing in the usual tables)
			// 64 bits of a vector from bit x on; outside it: undecided, kept
			auto	peek = [ & ]( const unsigned long long *v, int x ) -> unsigned long long {
				return x >= 0 && x + 96 <= vec_bits ? bits64( v, x ) : ~0ull;
			};
			// the bases at x .. x+63 pair with the ones d further on
			auto	pairs = [ & ]( int x, int d ) -> unsigned long long {
				unsigned long long	m = 0;
				for( int b = 0; b < nb; b++ )
					m |= peek( tv + b * pb_words, x ) & peek( pb + b * pb_words, x + d );
				return m;
			};
			if( tid == 0 )
				s_inplace = 0;
			int	cur = 0, slot_key[ 3 ] = { -1, -1, -1 };		// (stem-loops of one shape -- trna.descr's anticodon and T arms -- share their cores)
			for( int k = C.n - 1; k >= 0; k-- ){
				const rmd_chain_sib_t	sb = C.sib[ k ];
				const int	slot = sb.core_slot >= 0 ? sb.core_slot : 2;
				unsigned long long	*const dst = tv + ( 5 + cur ) * pb_words, *const core = tv + ( 7 + slot ) * pb_words;
				const unsigned long long	*const nxt = tv + ( 5 + ( cur ^ 1 ) ) * pb_words;
				const int	key = ( int( sb.hmin ) << 20 ) | ( int( sb.lmin ) << 10 ) | int( sb.lmax );
				if( sb.leaf && key != slot_key[ slot ] ){
					slot_key[ slot ] = key;
					// the innermost hmin pairs of the stem-loop, for one of its loop lengths
					for( int wi = tid; wi < vec_words; wi += BLOCK ){
						const int	x = wi * 64;
						unsigned long long	f = 0;
						for( int L = sb.lmin; L <= sb.lmax && ~f; L++ ){
							unsigned long long	w1 = ~0ull;
							for( int j = 0; j < sb.hmin && w1; j++ )
								w1 &= pairs( x + j, 2 * sb.hmin + L - 1 - 2 * j );
							f |= w1;
						}
						core[ wi ] = f;
					}
					__syncthreads();
				}
				for( int wi = tid; wi < vec_words; wi += BLOCK ){
					const int	x = wi * 64;
					unsigned long long	f = ~0ull;
					if( sb.leaf ){
						// ... which lie as far in as its helix is longer than the shortest
						f = 0;
						for( int t = 0; t <= sb.tmax && ~f; t++ )
							f |= peek( core, x + t );
					}
					if( k < C.n - 1 && f ){
						// ... and the next group one of this group's lengths later
						unsigned long long	r = 0;
						for( int len = sb.len_lo; len <= sb.len_hi && ~r; len++ )
							r |= peek( nxt, x + len );
						f &= r;
					}
					dst[ wi ] = f;
				}
				__syncthreads();
				cur ^= 1;
			}
			const unsigned long long	*const g1 = tv + ( 5 + ( cur ^ 1 ) ) * pb_words;
			for( int wi = tid; wi < vec_words; wi += BLOCK ){
				const int	x = wi * 64;
				unsigned long long	r = 0;
				for( int h = C.s_lo; h <= C.s_hi && ~r; h++ )
					r |= peek( g1, x + h );
				xv[ wi ] = r & lit_starts( x );
			}
			__syncthreads();
		}
		if constexpr( G == 1 ){
			if( dbg & 131072 )		// (ablation: ... and the look-ahead chain; nothing is queued)
				continue;
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
			// one start position per lane: the end positions that pass the first-pairs test are queued
			auto	body = [ & ]( const int rel, const bool valid_in ){
				const int	szero = z0 + rel;
				bool	valid = valid_in;
				if( q1f && valid ){
					// no place within reach for the 4-plex' second strand: nothing starts here
					const int	bq = szero - p_lo + 64;
					valid = ( xv[ 2 * pb_words + ( bq >> 6 ) ] >> ( bq & 63 ) ) & 1;
				}
				int	hi = 0, lo = 1;
				if( valid )
					rmd_level0_range( P, szero, slen, &hi, &lo );
				for( int r0 = 0; r0 < n_rank; r0 += 64 ){
					unsigned long long	W = 0;
					if( valid && r0 <= hi - lo ){
						W = win( szero, hi, r0, lo );
						if( q1f && W ){
							// ... and end positions before which no third strand can stand (bit i: end hi - r0 - 63 + i)
							const int	bq = hi - r0 - 63 - p_lo + 64;
							if( bq >= 0 && bq + 96 <= vec_words * 64 )
								W &= bits64( xv + 3 * pb_words, bq );
						}
					}
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
			};
			bool	by_words = false;
			{
			bool	start_vec = chain;
			const unsigned long long	*src = xv;		// the vector of start positions worth a look
			if constexpr( !LEAN ){
				if( q1f && !pk0 && !( dbg & 33554432 ) ){
					// a 4-plex at the head of the search list: the start positions from which its second strand can be
					// reached (the strand filter's A), with the best literal within reach if there is one
					src = xv + 2 * pb_words;
					if( lit && sv_on ){
						unsigned long long	*const sv = occ + size_t( n_vec - 6 ) * pb_words;
						for( int wi = utid; wi < vec_words; wi += UNIT )
							sv[ wi ] = xv[ 2 * pb_words + wi ] & lit_starts( wi * 64 );
						SLOT_SYNC();
						src = sv;
					}
					start_vec = true;
				}
			}
			if( LEAN && !chain && lit && ( G > 1 || sv_on ) && !( dbg & 33554432 ) ){
				// no look-ahead chain, but a best literal: the start positions that have it within reach, as a vector
				// (ire.descr, mp.ends.descr: one position in some hundred -- the same rounds saved)
				for( int wi = utid; wi < vec_words; wi += UNIT )
					lsv[ wi ] = lit_starts( wi * 64 );
				SLOT_SYNC();
				src = lsv;
				start_vec = true;
			}
			if( start_vec ){
				// The look-ahead leaves a start position in fourteen (trna.descr).  They are taken from the words
				// of its vector -- a lane looks at 16 positions at once and hands on the bits that are set --
				// instead of position by position (36 rounds of the wave per tile, each for four or five
				// survivors: the rounds, not the survivors, were where the pre-filter's time went), collected
				// per wave and searched 64 at a time, so that the row windows run with their lanes full.
				by_words = true;
				__shared__ uint16_t	s_cbuf2[ BLOCK / 64 ][ 128 ];
				uint16_t	*const cbuf = s_cbuf2[ tid >> 6 ];
				auto	take = [ & ](){
					__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
					const int	n_take = n_wait < 64 ? n_wait : 64;
					const bool	have = lane_id < n_take;
					const int	rel = have ? int( cbuf[ lane_id ] ) : 0;
					const int	rest = n_wait - n_take;
					const int	moved = lane_id < rest ? int( cbuf[ n_take + lane_id ] ) : 0;
					__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" );
					if( lane_id < rest )
						cbuf[ lane_id ] = uint16_t( moved );
					n_wait = rest;
					body( rel, have );
				};
				for( int q0 = 0; q0 < n_pos; q0 += UNIT * 16 ){
					const int	rel_lo = q0 + utid * 16;
					unsigned	m = 0;
					if( rel_lo < n_pos ){
						// (n_pos covers what the position-by-position test asks: inside the tile, the entry and the slice)
						m = unsigned( bits64( src, z0 + rel_lo - p_lo + 64 ) ) & 0xffffu;
						if( n_pos - rel_lo < 16 )
							m &= ( 1u << ( n_pos - rel_lo ) ) - 1u;
					}
					while( __ballot( m != 0 ) ){
						bool	ok = m != 0;
						const int	rel = rel_lo + ( ok ? __ffs( int( m ) ) - 1 : 0 );
						m &= m - 1;
						if( ok )
							LIT_OK( z0 + rel, ok );
						const unsigned long long	mv = __ballot( ok );
						if( ok )
							cbuf[ n_wait + __popcll( mv & lt_mask ) ] = uint16_t( rel );
						n_wait += __popcll( mv );
						while( n_wait >= 64 )
							take();
					}
				}
				while( n_wait > 0 )
					take();
			}
			}
			// (general instance, a pseudoknot's first helix at the start position and a best literal: the start positions
			// worth a look come from the words of a vector -- the literal within reach AND the helix' anchored seq= prefix,
			// both from the "stands here" vectors -- one bit per lane and round, instead of position by position)
			bool	pk_words = false, pk_done = false;
			unsigned	pk_m = 0;
			int	pk_q0 = 0, pk_lo = 0;
			if constexpr( !LEAN ){
			if( pk0 && lit && sv_on && !( dbg & 33554432 ) ){
				unsigned long long	*const sv = occ + size_t( n_vec - 6 ) * pb_words;
				const unsigned long long	*const lv = occ + size_t( n_vec - 5 ) * pb_words;
				const bool	pre = e0.re >= 0 && e0.mismatch == 0;
				const rmd_regex_t	&pre_re = rmd_regexes( P )[ pre ? e0.re : 0 ];
				const int	n_pre = pre ? rmd_imin( pre_re.n_prefix, e0.minlen ) : 0;
				for( int wi = utid; wi < vec_words; wi += UNIT ){
					unsigned long long	acc = lit_starts( wi * 64 );
					for( int jj = 0; jj < n_pre && acc; jj++ ){
						unsigned long long	lo = 0, hi = 0;
						for( int c = 0; c < 5; c++ )
							if( ( pre_re.accept[ c ] >> jj ) & 1 ){
								lo |= lv[ c * pb_words + wi ];
								hi |= wi + 1 < pb_words ? lv[ c * pb_words + wi + 1 ] : ~0ull;
							}
						acc &= jj ? ( lo >> jj ) | ( hi << ( 64 - jj ) ) : lo;
					}
					sv[ wi ] = acc;
				}
				SLOT_SYNC();
				pk_words = true;
			}
			}
			for( int j = 0; ; ){
				int	rel0 = 0;
				bool	valid0 = false, last_j = false;
				if( pk_words ){
					if( pk_done )
						break;
					const unsigned long long	*const sv = occ + size_t( n_vec - 6 ) * pb_words;
					while( __ballot( pk_m != 0 ) == 0 && pk_q0 < n_pos ){
						pk_lo = pk_q0 + utid * 16;
						pk_m = 0;
						if( pk_lo < n_pos ){
							pk_m = unsigned( bits64( sv, z0 + pk_lo - p_lo + 64 ) ) & 0xffffu;
							if( n_pos - pk_lo < 16 )
								pk_m &= ( 1u << ( n_pos - pk_lo ) ) - 1u;
						}
						pk_q0 += UNIT * 16;
					}
					if( __ballot( pk_m != 0 ) == 0 ){
						pk_done = true;		// (one more round, without positions: what still waits is searched)
						last_j = true;
					}else{
						valid0 = pk_m != 0;
						rel0 = pk_lo + ( valid0 ? __ffs( int( pk_m ) ) - 1 : 0 );
						pk_m &= pk_m - 1;
					}
				}else{
					if( j >= ( by_words ? 0 : n_pos ) )
						break;
					rel0 = j + utid;
					valid0 = rel0 < T && z0 + rel0 <= slen - P->dminlen && z0 + rel0 < pos_hi;
					last_j = j + UNIT >= n_pos;
					j += UNIT;
				}
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
				body( rel0, valid0 );
			}
		}else{
		bool	by_words = false;
		if constexpr( LEAN ){
		if( lit && ( G > 1 || sv_on ) && !quick && !split_ranks && !( dbg & 33554432 ) ){
			// whole start positions are queued, and only those with the best literal within reach (ire.descr: one in
			// 170): taken from the words of that vector, as above, not position by position
			by_words = true;
			for( int wi = utid; wi < vec_words; wi += UNIT )
				lsv[ wi ] = lit_starts( wi * 64 );
			SLOT_SYNC();
			for( int q0 = 0; q0 < n_pos; q0 += UNIT * 16 ){
				const int	rel_lo = q0 + utid * 16;
				unsigned	m = 0;
				if( rel_lo < n_pos ){
					m = unsigned( bits64( lsv, z0 + rel_lo - p_lo + 64 ) ) & 0xffffu;
					if( n_pos - rel_lo < 16 )
						m &= ( 1u << ( n_pos - rel_lo ) ) - 1u;
				}
				while( __ballot( m != 0 ) ){
					bool	pred = m != 0;
					const int	rel = rel_lo + ( pred ? __ffs( int( m ) ) - 1 : 0 ), szero = z0 + rel;
					m &= m - 1;
					if( pred )
						LIT_OK( szero, pred );
					pred = pred && ( !e0_at_szero || rmd_prefix_ok( P, e0, sq, szero ) );
					QPUSH( pred, ( unsigned( rel ) << 16 ) | 0xffffu, szero, 0, RMD_ALL_RANKS );
				}
			}
		}
		}
		for( int j = 0; j < ( by_words ? 0 : n_pos ); j += UNIT ){
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
		if constexpr( !WALK ){
			if( tid == 0 && s_qn > qtotal )
				atomicMax( hb.ticket + ( RMK_GCTL + 1 ), ( unsigned long long )s_qn );
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
			unsigned	*const pool = hb.pool + ( size_t( hb.glist_cap ) + size_t( blockIdx.x ) * hb.pool_cap ) * RMK_POOL_WORDS;
			{
				TailAccel	ac{ P, pb, tile, pb_words, p_lo, vec_words * 64, 0 };
				// (the chain's cores are where it left them unless the pre-filter had to search in place)
				const bool	head_next = chain_vecs && bitpar && P->chain.hn_on && !( dbg & ( 32768 | 262144 ) ) && !last && s_inplace == 0;
				for( int c = 0; c < nq; c += BLOCK ){
					const int	i = c + tid;
					unsigned	item = 0, hm[ 2 ] = { ~0u, ~0u };	// (the 3' ends left to the first helix of the interior, per outer length)
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
							// (every outer length is looked at: each leaves its own ends to the interior's first helix)
							const bool	want_masks = head_next && e0.head_pre_min == e0.head_pre_max;
							if( want_masks )
								hm[ 0 ] = hm[ 1 ] = 0;
							for( int hl = e0.minlen; hl <= e0.maxlen && ( !keep || want_masks ); hl++ ){
								const int	ilen = span - 2 * hl;
								if( ilen < e0.minilen )
									break;
								if( ilen > e0.maxilen )
									continue;
								unsigned	ends_left = ~0u;
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
										else{
											unsigned long long	Wd = rows_win( pb, pb_words, tile, p_lo, H.minlen, lim, ( H.ends & RMA_5PAIRED ) != 0, s5, w0, bot );
											if( !( dbg & 134217728 ) ){
												// ... of which the walk takes only those that leave the groups behind the helix their room and let
												// them reach the interior's end (rmd_lean_open, rmd_lean_step: rem_min, rem_max): bit i = end w0 + i
												const int	e_hi = rmd_imin( top, b - H.rem_min ) - w0, e_lo = ( H.rem_max >= 0 ? rmd_imax( bot, b - H.rem_max ) : bot ) - w0;
												if( e_hi < 63 )
													Wd &= e_hi < 0 ? 0ull : ( 2ull << e_hi ) - 1;
												if( e_lo > 0 )
													Wd &= e_lo > 63 ? 0ull : ~0ull << e_lo;
											}
											if( Wd && head_next ){
												// ... and one of those ends must have the next stem-loop's core at the right distance
												// behind it (bit i of Wd: end w0 + i)
												const rmd_chain_t	&C = P->chain;
												unsigned long long	cw = 0;
												bool	all = true;
												for( int g = C.hn_glo + 1; g <= C.hn_ghi + 1 + C.hn_tmax; g++ ){
													const int	x = w0 + g - p_lo + 64;
													if( x < 0 || x + 96 > vec_words * 64 )
														all = false;		// (undecided)
													else
														cw |= bits64( tv + ( 7 + C.hn_slot ) * pb_words, x );
												}
												if( all )
													Wd &= cw;
												// (bit j of the mask: end bot + j)
												if( all && top - bot < 32 )
													ends_left = unsigned( Wd >> ( bot - w0 ) );
											}
											ok = Wd != 0;
										}
									}
								}
								if( ok && hl - e0.minlen < 2 )
									hm[ hl - e0.minlen ] = want_masks ? ends_left : ~0u;
								if( ok && hl - e0.minlen >= 2 )
									hm[ 0 ] = hm[ 1 ] = ~0u, keep = true;		// (a third length: no room for its mask; no restriction)
								keep = keep || ok;
							}
							// (a length that failed its tests keeps the mask 0: the walk finds nothing there either way)
						}
					}
					// what the pool holds of an item: its entry, its start position there, its rank among the end positions
					int	p_seq = seq, p_szero = z0 + int( item >> 16 ), p_r = int( item & 0xffffu );
					if constexpr( CONCAT ){
						if( keep ){
							int	st_u_, sl_;
							keep = super_convert( P, db.base_off, db.slen, db.concat_bases, sink.comp, seq, ent_n, p_szero, p_r, &p_seq, &p_szero, &p_r, &st_u_, &sl_ );
						}
					}
					if constexpr( !WALK ){
						// straight to the drain kernel's list, piece by piece (pool_sub_count): the wave's items side by side
						const unsigned	rc = unsigned( p_r ) | ( unsigned( sink.comp ) << 16 );
						const bool	cut = !( dbg & 2097152 );		// (diagnostic: items go whole)
						const int	n_p = !keep ? 0 : !cut ? 1 : pool_sub_count( rc, hm[ 0 ], hm[ 1 ] );
						int	incl = n_p;
						for( int o = 1; o < 64; o <<= 1 ){
							const int	v = __shfl_up( incl, o );
							if( lane_id >= o )
								incl += v;
						}
						const int	tot = __shfl( incl, 63 );
						if( tot > 0 ){
							unsigned long long	g0 = 0;
							if( lane_id == 0 )
								g0 = atomicAdd( hb.ticket + ( RMK_GCTL - 1 ), ( unsigned long long )tot );
							g0 = __shfl( g0, 0 );
							// (a full list: what has room is written -- every place below the list's end has one owner --, the
							// host sees more reserved than there is and repeats the scan with a list that holds it)
							const long long	at = ( long long )g0 + incl - n_p;
							for( int p = 0; p < n_p; p++ )
								if( at + p < hb.glist_cap ){
									unsigned	*o = hb.pool + ( at + p ) * RMK_POOL_WORDS;
									o[ 0 ] = unsigned( p_seq );
									o[ 1 ] = unsigned( p_szero );
									if( cut )
										pool_sub_piece( rc, hm[ 0 ], hm[ 1 ], p, o + 2, o + 3, o + 4 );
									else
										o[ 2 ] = rc, o[ 3 ] = hm[ 0 ], o[ 4 ] = hm[ 1 ];
								}
						}
						continue;
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
							unsigned	*e = pool + RMK_POOL_WORDS * size_t( base + __popcll( m & lt_mask ) );
							e[ 3 ] = hm[ 0 ];
							e[ 4 ] = hm[ 1 ];
							e[ 0 ] = unsigned( p_seq );
							e[ 1 ] = unsigned( p_szero );
							e[ 2 ] = unsigned( p_r ) | ( unsigned( sink.comp ) << 16 );
						}
					}
				}
			}
			PHASE( 3 );
			if constexpr( WALK ){
			__syncthreads();
			// What the workgroup's pool still holds at the end goes to the device-wide list the drain kernel works on
			// (rma_drain_kernel) when it is too little to keep the workgroup's lanes busy: GLIST_BELOW items.  (Many
			// items, as ire.descr and mp.ends.descr leave them -- some hundred per workgroup, a step or two each --
			// are walked best where they are, 256 lanes on them while other workgroups still filter: 0.05 ms against
			// 0.28 in the drain kernel.  trna.descr leaves some twenty per workgroup, walks of dozens of steps.)
			if( hb.glist_cap > 0 && s_glist_full == 0 && s_pool_n > 0 && last && ( s_pool_n < GLIST_BELOW || ( dbg & 8388608 ) ) && !( dbg & 2048 ) ){
				const int	n_fl = s_pool_n;
				const bool	cut = !( dbg & 2097152 );		// (diagnostic: items go whole)
				// how many list items the pool's make (pool_sub_count) ...
				int	mine = 0;
				for( int i = tid; i < n_fl; i += BLOCK ){
					const unsigned	*e = pool + RMK_POOL_WORDS * size_t( i );
					mine += !cut ? 1 : pool_sub_count( __hip_atomic_load( e + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ),
						__hip_atomic_load( e + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ), __hip_atomic_load( e + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) );
				}
				if( mine )
					atomicAdd( &s_fl_total, mine );
				__syncthreads();
				// ... their place in the list, all or none ...
				if( tid == 0 )
					s_gstart = ( long long )atomicAdd( hb.ticket + ( RMK_GCTL - 1 ), ( unsigned long long )s_fl_total );
				__syncthreads();
				const long long	g0 = s_gstart;
				const int	n_list = s_fl_total;
				const bool	fits = g0 + n_list <= hb.glist_cap;
				if( fits ){
					// ... and the pieces, in whatever order the lanes get to them
					for( int i = tid; i < n_fl; i += BLOCK ){
						const unsigned	*e = pool + RMK_POOL_WORDS * size_t( i );
						const unsigned	w0 = __hip_atomic_load( e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ), w1 = __hip_atomic_load( e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
						const unsigned	rc = __hip_atomic_load( e + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
						const unsigned	h0 = __hip_atomic_load( e + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ), h1 = __hip_atomic_load( e + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
						const int	n_p = !cut ? 1 : pool_sub_count( rc, h0, h1 );
						if( n_p == 0 )
							continue;
						const int	at = atomicAdd( &s_fl_pos, n_p );
						for( int p = 0; p < n_p; p++ ){
							unsigned	*o = hb.pool + ( g0 + at + p ) * RMK_POOL_WORDS;
							o[ 0 ] = w0;
							o[ 1 ] = w1;
							if( cut )
								pool_sub_piece( rc, h0, h1, p, o + 2, o + 3, o + 4 );
							else
								o[ 2 ] = rc, o[ 3 ] = h0, o[ 4 ] = h1;
						}
					}
				}else{
					// no room: what was reserved of the list stays void, the workgroup walks its items itself from now on
					const int	n_void = hb.glist_cap > g0 ? int( hb.glist_cap - g0 < n_list ? hb.glist_cap - g0 : n_list ) : 0;
					for( int i = tid; i < n_void; i += BLOCK )
						hb.pool[ ( g0 + i ) * RMK_POOL_WORDS ] = 0xffffffffu;
				}
				__syncthreads();
				if( tid == 0 ){
					if( fits )
						s_pool_n = 0;
					else
						s_glist_full = 1;
					s_fl_total = 0;
					s_fl_pos = 0;
				}
				__syncthreads();
			}
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
				unsigned long long	t_wave = 0;
				rmd_lean_t	st;
				const rmd_no_accel_t	none;
				for( ; ; ){
					const unsigned long long	want = __ballot( k < 0 && !dry );
					const unsigned long long	busy = __ballot( k >= 0 );
					// idle lanes pop together: a window costs some hundred instructions to rebuild, and
					// a round for one lane costs the wave as much as a round for sixteen
					if( want && ( busy == 0 || __popcll( want ) >= hb.pool_refill ) ){
						const unsigned long long	t_p0 = ( dbg & 32 ) ? __builtin_amdgcn_s_memtime() : 0;
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
								int	whole;		// (the workgroup's own items are whole: their order words start at 0)
								k = pool_item_begin<BLOCK>( P, db, pool + RMK_POOL_WORDS * size_t( i ), col, NIB_MAX, lr, st, nsq, sink.seq, sink.comp, whole );
							}else
								dry = true;
						}
						if( ( dbg & 32 ) && lane_id == 0 )
							atomicAdd( hb.ticket + 19, __builtin_amdgcn_s_memtime() - t_p0 );
						continue;
					}
					if( busy == 0 )
						break;
					const unsigned long long	t_s0 = ( dbg & 32 ) ? __builtin_amdgcn_s_memtime() : 0;
					if( ( dbg & 32 ) && lane_id == 0 ){
						atomicAdd( hb.ticket + 17, 1ull );
						atomicAdd( hb.ticket + 18, ( unsigned long long )__popcll( busy ) );
					}
					// (diagnostic: the deepest level any lane of the wave steps at)
					int	k_in_ = 0;
					if( dbg & 32 ){
						k_in_ = k;
						for( int o = 32; o > 0; o >>= 1 )
							k_in_ = rmd_imax( k_in_, __shfl_xor( k_in_, o ) );
					}
					if( k >= 0 )
						k = rmd_lean_step<LdsRecs<BLOCK>, DevSink, rmd_nibseq_t<BLOCK>, rmd_no_accel_t, true>( P, lr, st, nsq, k, &lane, sink, none );
					// complete matches, one at a time, the whole wave on each
					const unsigned long long	t_e0 = ( dbg & 32 ) ? __builtin_amdgcn_s_memtime() : 0;
					const int	n_em_ = ( dbg & 32 ) ? __popcll( __ballot( k >= 0 && st.pending ) ) : 0;
					wave_emit_pending<BLOCK>( P, lr, st, k, [ & ]( int l ){
						return rmd_nibseq_t<BLOCK>{ nsq.w + ( l - lane_id ), __shfl( nsq.flip, l ), __shfl( nsq.bias, l ) }; },
						sink.seq, sink.comp, hb, lane_id );
					if( ( dbg & 32 ) && lane_id == 0 && n_em_ ){
						atomicAdd( hb.ticket + 92, __builtin_amdgcn_s_memtime() - t_e0 );
						atomicAdd( hb.ticket + 93, ( unsigned long long )n_em_ );
					}
					if( ( dbg & 32 ) && lane_id == 0 ){
						const unsigned long long	dt_ = __builtin_amdgcn_s_memtime() - t_s0;
						atomicAdd( hb.ticket + 20, dt_ );
						atomicMax( hb.ticket + 21, dt_ );		// the longest single step
						atomicAdd( hb.ticket + 23 + ( 63 - __clzll( dt_ | 1ull ) ), 1ull );
						atomicAdd( hb.ticket + 60 + rmd_imin( rmd_imax( k_in_, 0 ), 15 ), dt_ );
						atomicAdd( hb.ticket + 76 + rmd_imin( rmd_imax( k_in_, 0 ), 15 ), 1ull );
						t_wave += dt_;
						atomicMax( hb.ticket + 22, t_wave );	// the most any wave spent stepping
					}
				}
				__syncthreads();
				if( tid == 0 ){
					s_pool_n = 0;
					s_pool_head = 0;
				}
			}
			if( last ){
				PHASE( 4 );
				if( ( dbg & 1048576 ) && tid == 0 && had_tiles ){
					atomicMax( hb.ticket + 90, ( unsigned long long )wall_clock64() );
					atomicAdd( hb.ticket + 91, ( unsigned long long )wall_clock64() );
				}
				break;
			}
			}	// WALK
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
					k = rmd_lean_step<LdsRecs<BLOCK>, DevSink, rmd_seq_t, TailAccel, true>( P, lr, st, sq, k, &lane, sink, accel );
				wave_emit_pending<BLOCK>( P, lr, st, k, [ & ]( int l ){
					return rmd_seq_t{ tile0 + __shfl( int( sq.sq - tile0 ), l ), __shfl( sq.sq0, l ) }; },
					sink.seq, sink.comp, hb, lane_id );
				if( ( dbg & 32 ) && lane_id == 0 )
					atomicAdd( hb.ticket + 20, __builtin_amdgcn_s_memtime() - t_b1 );
			}
		}else{
			GenTile	gt{ P, lean_lo, g_before, g_deep, queue, spill, qcap, nq, &s_qhead, &s_dqn, &s_dqhead,
				pb, tile, pb_words, p_lo, vec_words, slen, z0, split_s, dbg,
				ConcatCtx{ db.base_off, db.slen, CONCAT ? db.concat_bases : 0ll, seq, ent_n } };
			general_pass_b<BLOCK, KINDS, CONCAT>( gt, sink );
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
#define EFN_CACHE	96
#define EFN_LIGHT_LDS	( 64 * ( ( EFN_CACHE + 2 ) * 2 + EFN_CACHE + 4 ) )	// dynamic LDS of rma_efn_light_kernel
// BIG: for descriptors with an efn() / efn2() call over more than 15 helices (rmd_program_t::efn_big): the loops' stacks sized
// for fifty (rm_efn_core.h)
// STAGE = false, BLOCK = 64 (rma_scan_begin's pipelined callers: two scanners in turns): a workgroup of one wave that reads the
// tables where they are -- 61 KB that the caches hold -- and keeps 19 KB of LDS for its lanes' calls, at no more than 128
// registers: it finds room on a CU next to the workgroups of a search kernel (the instance that walks nothing leaves 33 KB and
// a wave per SIMD), where the staged form -- 136 KB -- waits until the other scanner's search kernel is through.
template< int BLOCK, int BIG, bool STAGE >
__device__ __forceinline__ void efn_body( const rmd_program_t *gP, const DbView &db, int32_t *hits, long long n_hits,
	const int16_t *g16, const int32_t *tlkey, const int32_t *loginc, const rma_efn2data_t *e2 )
{
	// tables staged 16 bytes per lane and step (the device copy is padded to a multiple of 8 entries)
	__shared__ __align__( 16 ) int16_t	t16_s[ STAGE ? RME_N16_PAD : 8 ];
	const int16_t	*t16 = g16;
	if constexpr( STAGE ){
		if( g16 != nullptr )
			for( int i = threadIdx.x; i < RME_N16_PAD / 8; i += BLOCK )
				reinterpret_cast<uint4 *>( t16_s )[ i ] = reinterpret_cast<const uint4 *>( g16 )[ i ];
		__syncthreads();
		t16 = t16_s;
	}
	// per lane: base codes and partners of the call, when it is short enough
	// (STAGE = false: in dynamic LDS -- the compiler, not knowing how much there is, keeps to the registers the launch bounds ask for)
	__shared__ int16_t	s_bp[ STAGE ? BLOCK : 1 ][ EFN_CACHE + 1 ];
	__shared__ uint8_t	s_bc[ STAGE ? BLOCK : 1 ][ EFN_CACHE + 4 ];
	rme_tables_t	T{ t16, tlkey, loginc };
	int16_t	*bpbuf = s_bp[ STAGE ? threadIdx.x : 0 ];
	uint8_t	*bcbuf = s_bc[ STAGE ? threadIdx.x : 0 ];
	if constexpr( !STAGE ){
		extern __shared__ __align__( 16 ) unsigned char	efn_dyn[];
		bpbuf = reinterpret_cast<int16_t *>( efn_dyn ) + threadIdx.x * ( EFN_CACHE + 2 );
		bcbuf = efn_dyn + BLOCK * ( EFN_CACHE + 2 ) * sizeof( int16_t ) + threadIdx.x * ( EFN_CACHE + 4 );
	}
	const int	efn_off = RMA_HIT_HDR + 4 * gP->n_elems + 4;
	for( long long h = ( long long )blockIdx.x * BLOCK + threadIdx.x; h < n_hits; h += ( long long )gridDim.x * BLOCK ){
		int32_t	*w = hits + h * gP->hit_stride;
		DevSeq	sq{ db, db.base_off[ w[ 0 ] ], db.slen[ w[ 0 ] ], w[ 1 ] };
		for( int k = 0; k < gP->n_efn; k++ ){
			if( rmd_efn_sites( gP )[ k ].kind == RMA_EFN_KIND_EFN2 )
				w[ efn_off + k ] = e2 != nullptr ? rme2_site_energy<DevSeq, BIG>( gP, e2, &sq, w, k, bpbuf, bcbuf, EFN_CACHE ) : RME2_INF;
			else if( g16 != nullptr )
				w[ efn_off + k ] = rme_site_energy<DevSeq, BIG>( gP, &T, &sq, w, k, bpbuf, bcbuf, EFN_CACHE );
		}
	}
}

template< int BLOCK, int BIG = 0 >
__global__ void __launch_bounds__( BLOCK )
rma_efn_kernel( const rmd_program_t *gP, DbView db, int32_t *hits, long long n_hits,
	const int16_t *g16, const int32_t *tlkey, const int32_t *loginc, const rma_efn2data_t *e2 )
{
	efn_body<BLOCK, BIG, true>( gP, db, hits, n_hits, g16, tlkey, loginc, e2 );
}

// (120 registers: four workgroups of the search instance that walks nothing, at 96 each, leave a SIMD 128)
template< int BIG = 0 >
__global__ void __launch_bounds__( 64, 4 )
rma_efn_light_kernel( const rmd_program_t *gP, DbView db, int32_t *hits, long long n_hits,
	const int16_t *g16, const int32_t *tlkey, const int32_t *loginc, const rma_efn2data_t *e2 )
{
	efn_body<64, BIG, false>( gP, db, hits, n_hits, g16, tlkey, loginc, e2 );
}

// ---------------------------------------------------------------- launchers
// One per translation unit (rm_scan_inst_*.hip): the instance's dynamic LDS limit and its launch.
#define RMK_DEFINE_LAUNCHER( name_, LEAN_, G_, KINDS_, POOL_, ... ) \
hipError_t name_( int grid, size_t lds, hipStream_t s, const rmk_search_args &a ) \
{ \
	constexpr int	BLOCK_ = ( LEAN_ ) ? SEARCH_BLOCK : GENERAL_BLOCK; \
	auto	kernel = &rma_search_kernel<BLOCK_, LEAN_, G_, KINDS_, POOL_ __VA_OPT__(,) __VA_ARGS__>; \
	hipError_t	e = hipFuncSetAttribute( reinterpret_cast<const void *>( kernel ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ); \
	if( e != hipSuccess ) \
		return e; \
	/* the workgroups are persistent (tiles by ticket): no more of them than the device holds at once -- the others \
	   would start when the first ones leave, find no tile and go (trna.descr: 2048 asked for, 1024 resident) */ \
	{ \
		static thread_local size_t	for_lds = ~size_t( 0 ); \
		static thread_local int	resident = 0, for_dev = -1; \
		int	dev = 0; \
		( void )hipGetDevice( &dev ); \
		if( for_lds != lds || for_dev != dev ){ \
			int	per_cu = 0, cus = 0; \
			resident = 0; \
			if( hipOccupancyMaxActiveBlocksPerMultiprocessor( &per_cu, kernel, BLOCK_, lds ) == hipSuccess && \
				hipDeviceGetAttribute( &cus, hipDeviceAttributeMultiprocessorCount, dev ) == hipSuccess ) \
				resident = per_cu * cus; \
			for_lds = lds; \
			for_dev = dev; \
		} \
		if( resident > 0 && grid > resident ) \
			grid = resident; \
	} \
	hipLaunchKernelGGL( kernel, dim3( grid ), dim3( BLOCK_ ), lds, s, \
		a.d_prog, a.prog_bytes, a.qcap, a.db, a.hb, a.tile_bytes, a.dbg ); \
	return hipGetLastError(); \
}

#define RMK_DEFINE_DRAIN_LAUNCHER( name_ ) \
hipError_t name_( int grid, size_t lds, hipStream_t s, const rmk_search_args &a ) \
{ \
	auto	kernel = &rma_drain_kernel<DRAIN_BLOCK>; \
	hipError_t	e = hipFuncSetAttribute( reinterpret_cast<const void *>( kernel ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ); \
	if( e != hipSuccess ) \
		return e; \
	hipLaunchKernelGGL( kernel, dim3( grid ), dim3( DRAIN_BLOCK ), lds, s, a.d_prog, a.prog_bytes, a.db, a.hb, a.tile_bytes, a.dbg ); \
	return hipGetLastError(); \
}

// rm_dev_program.cpp -- reduce the boundary blob to the device form
// (see rm_dev_program.h).  Host code.
#include "rm_dev_program.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace {

const double	EPS = 1e-6;	// find_motif.c:15

// 0: too many states; 1: *out holds the expression; 2: *wide does (64 to 127 states) and *out only its flags
int build_regex( const rma_regex_t &re, rmd_regex_t *out, rmd_regex2_t *wide )
{
	memset( out, 0, sizeof( *out ) );
	memset( wide, 0, sizeof( *wide ) );
	out->anchored = re.anchored;
	out->dollar = re.dollar;
	out->fixed_len = re.fixed_len;
	out->wide = -1;
	int	n = 0, run = 0, longest = 0;
	auto add = [&]( const rma_re_atom_t &a, bool opt, bool star ) -> bool {
		if( n >= 127 )
			return false;
		const int	w = n >> 6;
		const uint64_t	bit = 1ull << ( n & 63 );
		for( int c = 0; c < 5; c++ )
			if( ( a.mask >> c ) & 1 )
				wide->accept[ c ][ w ] |= bit;
		if( opt ){
			wide->opt[ w ] |= bit;
			run++;
			longest = std::max( longest, run );
		}else
			run = 0;
		if( star )
			wide->star[ w ] |= bit;
		if( a.kind == 1 )
			wide->dot[ w ] |= bit;
		n++;
		return true;
	};
	for( int i = 0; i < re.n_atoms; i++ ){
		const rma_re_atom_t	&a = re.atoms[ i ];
		for( int k = 0; k < a.lo; k++ )
			if( !add( a, false, false ) )
				return 0;
		if( a.hi == 255 ){
			if( !add( a, true, true ) )
				return 0;
		}else for( int k = a.lo; k < a.hi; k++ )
			if( !add( a, true, false ) )
				return 0;
	}
	wide->n_states = n;
	wide->n_close = longest;
	if( n > 63 )
		return 2;
	// one word is enough: the form every test of the kernels reads
	for( int c = 0; c < 5; c++ )
		out->accept[ c ] = wide->accept[ c ][ 0 ];
	out->opt = wide->opt[ 0 ];
	out->star = wide->star[ 0 ];
	out->dot = wide->dot[ 0 ];
	out->n_states = n;
	out->n_close = longest;
	if( re.anchored ){
		while( out->n_prefix < n && !( ( out->opt >> out->n_prefix ) & 1 ) )
			out->n_prefix++;
	}
	return 1;
}

}	// namespace

int rmd_build( const rma_program_t *p, rmd_program_t *out, char *err, size_t errlen )
{
#define FAIL( ... )	do{ snprintf( err, errlen, __VA_ARGS__ ); return -1; }while( 0 )
	memset( out, 0, sizeof( *out ) );
	if( p->magic != RMA_MAGIC || p->size != sizeof( rma_program_t ) )
		FAIL( "not a motif program (bad magic or size)" );
	if( p->n_elems > RMD_MAX_ELEMS )
		FAIL( "descriptor has %d elements; the device scanner takes at most %d", p->n_elems, RMD_MAX_ELEMS );
	if( p->n_sites > RMD_MAX_SITES )
		FAIL( "descriptor has %d sites; the device scanner takes at most %d", p->n_sites, RMD_MAX_SITES );
	if( p->n_regexes > RMD_MAX_RE )
		FAIL( "too many seq= expressions for the device scanner" );
	out->n_elems = p->n_elems;
	out->n_searches = p->n_searches;
	out->step_budget = 128;	// (32 .. 128 measure alike, profiles/matrix_r2.sh; with levels chained within a step the longer one is ahead)
	out->dminlen = p->dminlen;
	out->w_winsize = p->dmaxlen < p->windowsize ? p->dmaxlen : p->windowsize;
	out->strict_helices = p->strict_helices;
	out->has_lctx = p->has_lctx;
	out->has_rctx = p->has_rctx;
	out->n_sites = p->n_sites;
	out->n_efn = p->n_efn_sites;
	out->efn_usestdbp = p->efn_usestdbp;
	out->hit_stride = rma_hit_stride( p );
	for( int s = 0; s < p->n_searches; s++ )
		out->searches[ s ] = int8_t( p->searches[ s ] );

	// pair sets: identical tables share a slot
	int	n_ps = 0;
	int	psmap[ RMA_MAX_PAIRSETS ];
	for( int i = 0; i < p->n_pairsets; i++ ){
		rmd_pairset_t	d;
		d.mat2 = p->pairsets[ i ].mat2;
		memcpy( d.mat3, p->pairsets[ i ].mat3, sizeof( d.mat3 ) );
		memcpy( d.mat4, p->pairsets[ i ].mat4, sizeof( d.mat4 ) );
		int	k;
		for( k = 0; k < n_ps; k++ )
			if( !memcmp( &out->pairsets[ k ], &d, sizeof( d ) ) )
				break;
		if( k == n_ps ){
			if( n_ps == RMD_MAX_PS )
				FAIL( "too many distinct pair sets for the device scanner" );
			out->pairsets[ n_ps++ ] = d;
		}
		psmap[ i ] = k;
	}
	out->efn_stdbp = p->efn_stdbp >= 0 ? psmap[ p->efn_stdbp ] : 0;

	out->n_regexes2 = 0;
	for( int i = 0; i < p->n_regexes; i++ ){
		rmd_regex2_t	wide;
		const int	kind = build_regex( p->regexes[ i ], &out->regexes[ i ], &wide );
		if( kind == 0 )
			FAIL( "a seq= expression expands to more than 127 positions" );
		if( kind == 2 ){
			if( out->n_regexes2 == RMD_MAX_RE2 )
				FAIL( "more than %d seq= expressions of more than 63 positions", RMD_MAX_RE2 );
			out->regexes[ i ].wide = out->n_regexes2;
			out->regexes2[ out->n_regexes2++ ] = wide;
		}
	}

	int	n_rules = 0;
	auto cvt = [&]( const rma_elem_t &e, rmd_elem_t *d ) -> int {
		d->type = int8_t( e.type );
		d->proper = int8_t( e.proper );
		d->ends = int8_t( e.ends );
		d->strict = int8_t( e.strict );
		d->loop = e.next >= 0 || e.outer < 0;
		d->next_s = e.next >= 0 ? int8_t( p->elems[ e.next ].searchno ) : -1;
		d->inner = int8_t( e.inner );
		d->inner_s = e.inner >= 0 ? int8_t( p->elems[ e.inner ].searchno ) : -1;
		d->searchno = int8_t( e.searchno );
		d->n_mates = int8_t( e.n_mates );
		for( int m = 0; m < 3; m++ )
			d->mates[ m ] = int8_t( e.mates[ m ] );
		d->n_scopes = int8_t( e.n_scopes );
		d->scope = int8_t( e.scope );
		for( int s = 0; s < 8; s++ )
			d->scopes[ s ] = int8_t( e.scopes[ s ] );
		d->pairset = e.pairset >= 0 ? int8_t( psmap[ e.pairset ] ) : -1;
		d->re = int8_t( e.re );
		d->minlen = e.minlen;
		d->maxlen = e.maxlen;
		d->minglen = e.minglen;
		d->maxglen = e.maxglen;
		d->minilen = e.minilen;
		d->maxilen = e.maxilen;
		d->mismatch = e.mismatch;
		d->quick = ( e.type == RMA_T_H5 && e.proper ) || e.type == RMA_T_Q1;
		d->q_iminl = e.minilen;
		if( e.type == RMA_T_Q1 )
			d->q_iminl += p->elems[ e.mates[ 0 ] ].minilen + p->elems[ e.mates[ 1 ] ].minilen + 2 * e.minlen;
		if( e.type == RMA_T_H5 && !e.proper ){
			out->need_init = 1;
			// find_pknot3 :558-562 with nothing matched yet: plain sums of minlen
			d->q_iminl = 0;
			for( int k = e.index + 1; k < e.mates[ 0 ]; k++ )
				d->q_iminl += p->elems[ k ].minlen;
			d->q_sminl = 0;
			int64_t	smax = 0;
			for( int k = e.mates[ 0 ] + 1; k <= e.scopes[ e.n_scopes - 1 ]; k++ ){
				d->q_sminl += p->elems[ k ].minlen;
				smax += p->elems[ k ].maxlen;
			}
			d->q_smaxl = smax < 30000 ? int32_t( smax ) : -1;
		}
		// every strand of a helix carries the group's rules (match_4plex reads them from q2)
		bool	helix = e.type != RMA_T_SS && e.type != RMA_T_CTX;
		if( helix ){
			if( e.maxlen > RMD_MAX_HLEN_WIDE )
				return 1;
			if( e.maxlen > RMD_MAX_HLEN )
				out->wide = 1;
			// find_motif.c:1023-1033
			if( e.mispair > 0 ){
				d->mplim = e.mispair;
				d->pfrac = 0;
			}else if( e.pairfrac < 1.0 ){
				d->mplim = int( ( 1. - e.pairfrac ) * std::min( e.maxlen, p->windowsize ) + 0.5 );
				d->pfrac = 1;
			}else{
				d->mplim = 0;
				d->pfrac = 0;
			}
			// one rule table per distinct (mispair, pairfrac) pair
			rmd_rule_t	rule;
			memset( &rule, 0, sizeof( rule ) );
			for( int hl = 0; hl <= RMD_MAX_HLEN_WIDE; hl++ ){
				int	best = 0;
				if( hl == 0 )
					best = 255;
				else for( int mpr = 0; mpr <= hl; mpr++ ){
					if( !( 1. * ( hl - mpr ) / hl < e.pairfrac - EPS ) )	// :1040,:1086
						best = mpr;
				}
				rule.pf_maxmpr[ hl ] = uint8_t( best );
				int	tq = 0;		// :1190-1194, :1241-1245
				if( e.mispair > 0 )
					tq = e.mispair;
				else if( e.pairfrac < 1.0 )
					tq = int( ( 1. - e.pairfrac ) * hl + 0.5 );
				rule.tq_mplim[ hl ] = uint8_t( std::min( tq, 255 ) );
			}
			int	k;
			for( k = 0; k < n_rules; k++ )
				if( !memcmp( &out->rules[ k ], &rule, sizeof( rule ) ) )
					break;
			if( k == n_rules ){
				if( n_rules == RMD_MAX_RULES )
					return 2;
				out->rules[ n_rules++ ] = rule;
			}
			d->rule = k;
		}
		return 0;
	};
	for( int i = 0; i < p->n_elems; i++ ){
		int	rc = cvt( p->elems[ i ], &out->elems[ i ] );
		if( rc == 2 )
			FAIL( "more than %d distinct helix mispair/pairfrac rules", RMD_MAX_RULES );
		if( rc )
			FAIL( "helix element %d allows %d base pairs; the device scanner takes at most %d",
				i + 1, p->elems[ i ].maxlen, RMD_MAX_HLEN_WIDE );
	}
	if( p->has_lctx )
		cvt( p->lctx, &out->lctx );
	if( p->has_rctx )
		cvt( p->rctx, &out->rctx );

	// structure the search relies on (the reference dereferences these unchecked)
	if( p->n_searches < 1 || p->elems[ p->searches[ p->n_searches - 1 ] ].type != RMA_T_SS )
		FAIL( "the last element searched must be an ss element" );
	for( int s = 0; s < p->n_searches; s++ ){
		const rma_elem_t	&e = p->elems[ p->searches[ s ] ];
		switch( e.type ){
		case RMA_T_SS :
			break;
		case RMA_T_H5 :
			if( e.proper && e.inner < 0 )
				FAIL( "helix element %d has no interior", e.index + 1 );
			if( !e.proper && ( e.n_scopes < 2 || e.n_scopes > 8 ) )
				FAIL( "pseudoknot with %d strands is not supported", e.n_scopes );
			break;
		case RMA_T_P5 :
			if( e.inner < 0 )
				FAIL( "helix element %d has no interior", e.index + 1 );
			break;
		case RMA_T_T1 :
			if( e.inner < 0 || p->elems[ e.scopes[ 1 ] ].inner < 0 )
				FAIL( "triplex element %d has an empty interior", e.index + 1 );
			break;
		case RMA_T_Q1 :
			if( e.inner < 0 || p->elems[ e.mates[ 0 ] ].inner < 0 || p->elems[ e.mates[ 1 ] ].inner < 0 )
				FAIL( "4-plex element %d has an empty interior", e.index + 1 );
			break;
		default :
			FAIL( "element %d cannot head a search", e.index + 1 );
		}
	}
	// (window below 4096: the lean records pack positions into 12 bits, rm_scan_hip.hip)
	out->lean_ok = p->n_searches <= RMD_LEAN_LEVELS && out->w_winsize < 4096 && p->n_elems <= 64 &&	// (64: an element per lane, the kernel's WaveTable)
		!out->wide;		// (the lean records hold a helix length in six bits)
	for( int s = 0; s < p->n_searches; s++ ){
		const rma_elem_t	&e = p->elems[ p->searches[ s ] ];
		if( !( e.type == RMA_T_SS || ( e.type == RMA_T_H5 && e.proper ) ) )
			out->lean_ok = 0;
	}
	// ss elements that head a level: early tests of an anchored seq= (rmd_elem_t::pin_start/pin_end_n)
	for( int d = 0; d < p->n_elems; d++ ){
		rmd_elem_t	&e = out->elems[ d ];
		e.pin_start = e.pin_end_n = 0;
		if( e.type != RMA_T_SS || e.re < 0 || e.mismatch != 0 || p->elems[ d ].searchno < 0 )
			continue;
		const rmd_regex_t	&re = out->regexes[ e.re ];
		e.pin_start = re.anchored && re.n_prefix > 0;
		if( !e.loop && re.dollar && re.fixed_len > 0 && re.fixed_len <= 63 && re.opt == 0 && re.star == 0 )
			e.pin_end_n = int8_t( re.fixed_len );
	}
	// an ss that is the whole of an interior (no loop over its end, find_motif :266-272) has one
	// alternative: the general path passes through such levels without a step of their own
	{
		int	back = -1;
		for( int s = 0; s < p->n_searches; s++ ){
			rmd_elem_t	&e = out->elems[ p->searches[ s ] ];
			e.back_s = int8_t( back );
			if( !( e.type == RMA_T_SS && !e.loop ) )
				back = s;
		}
	}
	// ... and the lean path's own (rmd_elem_t::lean_back_s): also a single strand of one length that is followed by others
	{
		int	back = -1;
		for( int s = 0; s < p->n_searches; s++ ){
			rmd_elem_t	&e = out->elems[ p->searches[ s ] ];
			e.lean_back_s = int8_t( back );
			e.ord_pad_ = 0;
			const bool	one = e.type == RMA_T_SS && ( !e.loop || ( s > 0 && e.maxglen != RMA_UNBOUNDED && e.minglen == e.maxglen ) );
			if( !one )
				back = s;
		}
	}
	// split level: the helices at the head of the search list are where most start positions die
	{
		int	run = 0;
		while( run < p->n_searches && out->elems[ p->searches[ run ] ].type != RMA_T_SS )
			run++;
		bool	more = false;		// a level with a choice after them
		for( int s = run; s < p->n_searches; s++ ){
			const rmd_elem_t	&e = out->elems[ p->searches[ s ] ];
			more = more || !( e.type == RMA_T_SS && !e.loop );
		}
		out->split_s = run >= 1 && run <= 4 && more ? run - 1 : -1;
	}
	{
		int	o = 0;
		for( int s = 0; s < p->n_searches; s++ ){
			const rmd_elem_t	&e = out->elems[ p->searches[ s ] ];
			const bool	single = s > 0 && e.type == RMA_T_SS && !e.loop;
			out->rec_off[ s ] = int16_t( single ? ( o | 0x8000 ) : o );
			o += single ? 1 : 3;
		}
		out->n_rec_dwords = o;
	}
	// first-tuple masks of the triplex / 4-plex pair tables
	int	n_tups = 0;
	for( int d = 0; d < p->n_elems; d++ )
		out->elems[ d ].tup = -1;
	for( int s = 0; s < p->n_searches; s++ ){
		rmd_elem_t	&e = out->elems[ p->searches[ s ] ];
		if( !( e.type == RMA_T_T1 || e.type == RMA_T_Q1 ) || e.pairset < 0 || n_tups == RMD_MAX_TUP )
			continue;
		// (match_4plex reads the pair table of the second strand, match_triplex that of the first:
		// one table per group, compile.c hands every strand the group's)
		const int	ps = e.type == RMA_T_Q1 ? out->elems[ e.mates[ 0 ] ].pairset : e.pairset;
		if( ps < 0 )
			continue;
		const rmd_pairset_t	&t = out->pairsets[ ps ];
		rmd_tup_t	&u = out->tups[ n_tups ];
		memset( &u, 0, sizeof( u ) );
		auto tri = [&]( int a, int b, int c ){ const int ix = ( a * 5 + b ) * 5 + c; return ( t.mat3[ ix >> 5 ] >> ( ix & 31 ) ) & 1; };
		auto quad = [&]( int a, int b, int c, int dd ){ const int ix = ( ( a * 5 + b ) * 5 + c ) * 5 + dd; return ( t.mat4[ ix >> 5 ] >> ( ix & 31 ) ) & 1; };
		for( int a = 0; a < 5; a++ )
			for( int c = 0; c < 5; c++ )
				for( int b = 0; b < 5; b++ ){
					if( tri( a, b, c ) )
						u.t2[ a * 5 + c ] |= uint8_t( 1u << b );
					for( int x = 0; x < 5; x++ )
						if( quad( a, b, x, c ) ){
							u.q2[ a * 5 + c ] |= uint8_t( 1u << b );
							u.q3[ ( a * 5 + b ) * 5 + c ] |= uint8_t( 1u << x );
						}
				}
		e.tup = int8_t( n_tups++ );
	}
	out->n_tups = n_tups;
	// a 4-plex at the head of the search list: what pass A can ask of its inner strands (rmd_q1filter_t)
	memset( &out->q1f, 0, sizeof( out->q1f ) );
	{
		const rmd_elem_t	&e0 = out->elems[ p->searches[ 0 ] ];
		if( e0.type == RMA_T_Q1 && e0.tup >= 0 && e0.n_mates == 3 && e0.minlen >= 1 && e0.minlen <= 12 ){
			const rmd_elem_t	&e1 = out->elems[ e0.mates[ 0 ] ], &e2 = out->elems[ e0.mates[ 1 ] ];
			const rmd_tup_t	&u = out->tups[ e0.tup ];
			rmd_q1filter_t	&f = out->q1f;
			for( int i = 0; i < 25; i++ )
				f.m2 |= int8_t( u.q2[ i ] );
			for( int i = 0; i < 125; i++ )
				f.m3 |= int8_t( u.q3[ i ] );
			f.first5 = ( e1.ends & RMA_5PAIRED ) != 0;
			f.nmin = int8_t( e0.minlen );
			int	bad = 0;
			for( int hl = e0.minlen; hl <= e0.maxlen && hl <= RMD_MAX_HLEN; hl++ )
				bad = std::max( bad, int( out->rules[ e1.rule ].tq_mplim[ hl ] ) );
			f.badmax = int8_t( std::min( bad, 3 ) );
			const int	big = 1 << 14;
			f.a_lo = int16_t( e0.minlen + e0.minilen );
			f.a_hi = int16_t( e0.maxilen < big && e0.maxlen < big ? e0.maxlen - 1 + e0.maxilen : -1 );
			f.b_hi = int16_t( e0.minlen + e2.minilen );
			f.b_lo = int16_t( e2.maxilen < big && e0.maxlen < big ? e0.maxlen - 1 + e2.maxilen : -1 );
			// (reach of a few words at most: what lies farther is left to the search)
			f.on = f.a_hi >= f.a_lo && f.b_lo >= f.b_hi && f.a_hi - f.a_lo < 64 && f.b_lo - f.b_hi < 64 && f.a_lo < 4000 && f.b_hi < 4000;
			// ... followed by a triplex, with at most one single strand in between?
			int	ss_lo = 0, ss_hi = 0, ts = e0.next_s;
			if( f.on && ts >= 0 && out->elems[ p->searches[ ts ] ].type == RMA_T_SS ){
				const rmd_elem_t	&ss = out->elems[ p->searches[ ts ] ];
				ss_lo = ss.minlen;
				ss_hi = ss.maxlen;
				ts = ss.next_s;
			}
			if( f.on && ts >= 0 && ss_hi < big && out->elems[ p->searches[ ts ] ].type == RMA_T_T1 ){
				const rmd_elem_t	&t = out->elems[ p->searches[ ts ] ];
				if( t.tup >= 0 && t.n_scopes == 3 && t.minlen >= 1 && t.minlen <= 12 && t.maxlen < big &&
					t.maxilen < big && out->elems[ t.scopes[ 1 ] ].maxilen < big ){
					const rmd_elem_t	&t2 = out->elems[ t.scopes[ 1 ] ];
					const rmd_tup_t	&tu = out->tups[ t.tup ];
					for( int b1 = 0; b1 < 5; b1++ )
						for( int b3 = 0; b3 < 5; b3++ )
							if( tu.t2[ b1 * 5 + b3 ] ){
								f.tm1 |= int8_t( 1 << b1 );
								f.tm3 |= int8_t( 1 << b3 );
								f.tm2 |= int8_t( tu.t2[ b1 * 5 + b3 ] );
							}
					f.tfirst5 = ( t.ends & RMA_5PAIRED ) != 0;
					f.tnmin = int8_t( t.minlen );
					int	tb = 0;
					for( int hl = t.minlen; hl <= t.maxlen && hl <= RMD_MAX_HLEN; hl++ )
						tb = std::max( tb, int( out->rules[ t.rule ].tq_mplim[ hl ] ) );
					f.tbad = int8_t( std::min( tb, 3 ) );
					f.f_lo = int16_t( 1 + ss_lo );
					f.f_hi = int16_t( 1 + ss_hi );
					f.r1_lo = int16_t( 2 * t.minlen + t.minilen - 1 );
					f.r1_hi = int16_t( 2 * t.maxlen + t.maxilen - 1 );
					f.r2_lo = int16_t( 1 + t2.minilen );
					f.r2_hi = int16_t( 1 + t2.maxilen );
					f.t_on = f.f_hi - f.f_lo < 64 && f.r1_hi - f.r1_lo < 64 && f.r2_hi - f.r2_lo < 64 && f.r1_hi < 2000;
				}
			}
		}
	}
	// pair row sets: one per distinct pair table of the helices that are matched with
	// match_wchlx() at a search level and whose first-pairs rule fits the bit-parallel test
	out->n_rowsets = 0;
	for( int d = 0; d < p->n_elems; d++ )
		out->elems[ d ].rows = -1;
	for( int s = 0; s < p->n_searches; s++ ){
		rmd_elem_t	&d = out->elems[ p->searches[ s ] ];
		if( !( d.type == RMA_T_H5 || d.type == RMA_T_Q1 ) || d.pairset < 0 || d.minlen < 1 || d.mplim > 3 )
			continue;
		int	k;
		for( k = 0; k < out->n_rowsets; k++ )
			if( out->rowset_ps[ k ] == d.pairset )
				break;
		if( k == out->n_rowsets ){
			if( out->n_rowsets == 4 )
				continue;
			out->rowset_ps[ out->n_rowsets++ ] = d.pairset;
		}
		d.rows = int8_t( k );
	}
	// improper helices: the sums find_pknot5/find_pknot3 take over ranges of the knot (rmd_pk_t)
	int	n_pks = 0;
	for( int d = 0; d < p->n_elems; d++ )
		out->elems[ d ].pk = -1;
	for( int s = 0; s < p->n_searches; s++ ){
		const rma_elem_t	&e = p->elems[ p->searches[ s ] ];
		if( e.type != RMA_T_H5 || e.proper )
			continue;
		if( n_pks == RMD_MAX_PK )
			FAIL( "more than %d pseudoknot helices", RMD_MAX_PK );
		rmd_pk_t	&pk = out->pks[ n_pks ];
		memset( &pk, 0, sizeof( pk ) );
		const int	d = e.index, d3 = e.mates[ 0 ], d0 = e.scopes[ 0 ], dn = e.scopes[ e.n_scopes - 1 ];
		// the helix a strand of the knot belongs to, as the search level of its 5' strand
		auto level_of = [&]( int el ) -> int {
			const rma_elem_t	&x = p->elems[ el ];
			return x.type == RMA_T_H5 ? x.searchno : p->elems[ x.mates[ 0 ] ].searchno;
		};
		for( int k = 0; k < 8; k++ )
			pk.lvl[ k ] = k < e.n_scopes ? int8_t( level_of( e.scopes[ k ] ) ) : int8_t( -1 );
		pk.hlx2 = d == e.scopes[ 1 ];
		const int	d3_h1 = p->elems[ e.scopes[ 0 ] ].mates[ 0 ];
		const int	lo[ RMD_PK_N ] = { d0, d, d + 1, d3 + 1, d + 1, d3_h1 + 1 };
		const int	hi[ RMD_PK_N ] = { d - 1, dn, d3 - 1, dn, d3_h1 - 1, d3 - 1 };
		for( int q = 0; q < RMD_PK_N; q++ ){
			if( ( q == RMD_PK_IL || q == RMD_PK_IR ) && !pk.hlx2 )
				continue;
			int64_t	mn = 0, mx = 0;
			for( int el = lo[ q ]; el <= hi[ q ]; el++ ){
				int	sc = -1;
				for( int k = 0; k < e.n_scopes; k++ )
					if( e.scopes[ k ] == el )
						sc = k;
				if( sc >= 0 && level_of( el ) < e.searchno ){
					pk.mask[ q ] |= uint8_t( 1u << sc );
					continue;
				}
				mn += p->elems[ el ].minlen;
				mx += p->elems[ el ].maxlen;
			}
			pk.bmin[ q ] = int32_t( std::min<int64_t>( mn, 1000000000 ) );
			pk.bmax[ q ] = int32_t( std::min<int64_t>( mx, 1000000000 ) );
		}
		// upd_pksearches(), find_motif.c:667-701
		pk.w_osd5 = pk.w_zero5 = pk.w_osd3 = pk.w_zero3 = -1;
		auto inner_level = [&]( int el ) -> int8_t {
			const int	in = p->elems[ el ].inner;
			return in >= 0 ? int8_t( p->elems[ in ].searchno ) : int8_t( -1 );
		};
		if( e.scope > 0 )
			pk.w_osd5 = inner_level( e.scopes[ e.scope - 1 ] );
		pk.w_zero5 = inner_level( d );
		const rma_elem_t	&e3 = p->elems[ d3 ];
		if( e3.scope > 0 )
			pk.w_osd3 = inner_level( e3.scopes[ e3.scope - 1 ] );
		if( e3.scope < e3.n_scopes - 1 )
			pk.w_zero3 = inner_level( d3 );
		out->elems[ d ].pk = int8_t( n_pks++ );
	}
	out->n_pks = n_pks;
	for( int d = 0; d < p->n_elems; d++ ){
		out->elems[ d ].rem_min = 0;
		out->elems[ d ].rem_max = -1;
		out->elems[ d ].tail_s = -1;
		out->elems[ d ].tail_pre_min = 0;
		out->elems[ d ].tail_pre_max = -1;
		out->elems[ d ].head_s = -1;
		out->elems[ d ].head_pre_min = 0;
		out->elems[ d ].head_pre_max = 0;
	}
	if( out->lean_ok ){
		// The reference walks a chain left to right and only learns at its end that the
		// remaining groups do not fit.  Both bounds below are implied by the hard length
		// limits find_motif() itself applies (find_motif.c:266-273, :423-440), so cutting
		// with them only skips branches that cannot produce a candidate.
		for( int s = 1; s < p->n_searches; s++ ){
			rmd_elem_t	*d = &out->elems[ p->searches[ s ] ];
			long	mn = 0, mx = 0;
			bool	unb = false;
			int	last = s;
			for( int j = d->next_s; j >= 0; j = out->elems[ p->searches[ j ] ].next_s ){
				const rmd_elem_t	&ej = out->elems[ p->searches[ j ] ];
				mn += ej.minglen;
				if( ej.maxglen == RMA_UNBOUNDED )
					unb = true;
				else
					mx += ej.maxglen;
				last = j;
			}
			const bool	closed = !out->elems[ p->searches[ last ] ].loop;
			if( d->loop && mn < 30000 )
				d->rem_min = int16_t( mn );
			if( d->loop && closed && !unb && mx < 30000 )
				d->rem_max = int16_t( mx );
		}
		for( int s = 0; s < p->n_searches; s++ ){
			rmd_elem_t	*d = &out->elems[ p->searches[ s ] ];
			if( d->type != RMA_T_H5 || d->inner_s < 0 )
				continue;
			long	mn = 0, mx = 0;
			bool	unb = false;
			int	t = d->inner_s;
			for( ; ; ){
				const rmd_elem_t	&ej = out->elems[ p->searches[ t ] ];
				if( ej.next_s < 0 )
					break;
				mn += ej.minglen;
				if( ej.maxglen == RMA_UNBOUNDED )
					unb = true;
				else
					mx += ej.maxglen;
				t = ej.next_s;
			}
			// the first helix of the interior, behind single strands of bounded length only
			{
				long	hmn = 0, hmx = 0;
				for( int h = d->inner_s; h >= 0; h = out->elems[ p->searches[ h ] ].next_s ){
					const rmd_elem_t	&eh = out->elems[ p->searches[ h ] ];
					if( eh.type == RMA_T_H5 ){
						if( eh.quick && eh.minlen >= 1 && eh.maxglen != RMA_UNBOUNDED && hmx - hmn <= 3 && hmx < 30000 ){
							d->head_s = int8_t( h );
							d->head_pre_min = int16_t( hmn );
							d->head_pre_max = int16_t( hmx );
						}
						break;
					}
					if( eh.type != RMA_T_SS || eh.maxglen == RMA_UNBOUNDED )
						break;
					hmn += eh.minglen;
					hmx += eh.maxglen;
				}
			}
			const rmd_elem_t	&te = out->elems[ p->searches[ t ] ];
			if( t != d->inner_s && te.quick && te.type == RMA_T_H5 && !te.loop && te.minlen >= 1 && mn < 30000 ){
				d->tail_s = int8_t( t );
				d->tail_pre_min = int16_t( mn );
				d->tail_pre_max = ( unb || mx >= 30000 ) ? int16_t( -1 ) : int16_t( mx );
			}
		}
	}
	// order words as numbers of the walk's choices (rmd_elem_t::ord_stride)
	out->ord_ok = 0;
	out->ord_bits = 0;
	for( int d = 0; d < p->n_elems; d++ ){
		out->elems[ d ].ord_stride = 0;
		out->elems[ d ].ord_nlen = 1;
		out->elems[ d ].ord_pad_ = 0;
	}
	if( out->lean_ok ){
		long long	weight = 1;
		bool	fits = true;
		for( int s = p->n_searches - 1; s >= 0 && fits; s-- ){
			rmd_elem_t	*d = &out->elems[ p->searches[ s ] ];
			// (the first level's end position is the candidate's rank: no digit for it)
			const long long	n_end = ( s == 0 || !d->loop ) ? 1 :
				( d->maxglen == RMA_UNBOUNDED ? ( long long )out->w_winsize : ( long long )d->maxglen ) - d->minglen + 1;
			const long long	n_len = d->type == RMA_T_SS ? 1 : ( long long )d->maxlen - d->minlen + 1;
			if( n_end < 1 || n_len < 1 || n_len > 30000 )
				fits = false;
			else{
				d->ord_stride = int32_t( weight );
				d->ord_nlen = int16_t( n_len );
				weight *= n_end * n_len;
				if( weight > ( 1ll << 31 ) )
					fits = false;
			}
		}
		if( fits ){
			out->ord_ok = 1;
			while( ( 1ll << out->ord_bits ) < weight )
				out->ord_bits++;
		}
	}
	// the look-ahead chain of the first element's interior (rmd_chain_t)
	memset( &out->chain, 0, sizeof( out->chain ) );
	if( out->lean_ok ){
		const rmd_elem_t	&e0 = out->elems[ p->searches[ 0 ] ];
		rmd_chain_t	&ch = out->chain;
		if( e0.type == RMA_T_H5 && e0.proper && e0.inner_s >= 0 && e0.rows == 0 && e0.minlen >= 1 && e0.maxlen < 1000 ){
			ch.s_lo = int16_t( e0.minlen );
			ch.s_hi = int16_t( e0.maxlen );
			bool	any_leaf = false;
			int	level_of[ RMD_MAX_CHAIN ];
			for( int s = e0.inner_s; s >= 0 && ch.n < RMD_MAX_CHAIN; s = out->elems[ p->searches[ s ] ].next_s ){
				const rmd_elem_t	&g = out->elems[ p->searches[ s ] ];
				if( g.maxglen == RMA_UNBOUNDED || g.maxglen >= 2000 )
					break;		// (what lies behind a group of any length is not pinned down by the start)
				level_of[ ch.n ] = s;
				rmd_chain_sib_t	&sb = ch.sib[ ch.n++ ];
				sb.len_lo = int16_t( g.minglen );
				sb.len_hi = int16_t( g.maxglen );
				if( g.type == RMA_T_H5 && g.proper && g.pairset == e0.pairset && g.mplim == 0 && !g.pfrac &&
					( g.ends & RMA_5PAIRED ) && ( g.ends & RMA_3PAIRED ) &&
					g.minlen >= 2 && g.minlen <= 16 && g.maxlen < 64 && g.inner_s >= 0 && g.maxilen != RMA_UNBOUNDED && g.maxilen - g.minilen < 32 ){
					// a stem-loop: nothing but single strands inside
					bool	only_ss = true;
					for( int t = g.inner_s; t >= 0; t = out->elems[ p->searches[ t ] ].next_s )
						only_ss = only_ss && out->elems[ p->searches[ t ] ].type == RMA_T_SS;
					// ... and worth its price: a stem-loop that most positions could start (a short helix with many
					// loop lengths: trna.descr's D arm, 3 pairs and 8 loop lengths, stands at one position in two)
					// costs a pass over the tile per loop length and filters little
					double	p_pair = 0;
					{
						const uint32_t	m2 = out->pairsets[ e0.pairset ].mat2;
						for( int a = 0; a < 4; a++ )
							for( int b = 0; b < 4; b++ )
								p_pair += ( ( m2 >> ( a * 5 + b ) ) & 1 ) / 16.0;
					}
					const double	p_leaf = ( g.maxlen - g.minlen + 1 ) * ( g.maxilen - g.minilen + 1 ) * std::pow( p_pair, g.minlen );
					if( only_ss && p_leaf <= 0.25 ){
						sb.leaf = 1;
						sb.hmin = int8_t( g.minlen );
						sb.tmax = int16_t( g.maxlen - g.minlen );
						sb.lmin = int16_t( g.minilen );
						sb.lmax = int16_t( g.maxilen );
						any_leaf = true;
					}
				}
			}
			// (drop groups behind the last stem-loop: they say nothing)
			while( ch.n > 0 && !ch.sib[ ch.n - 1 ].leaf )
				ch.n--;
			// the cores of the first two shapes of stem-loop are kept for the candidate test of pass A'
			{
				int	keys[ 2 ] = { -1, -1 };
				for( int k = ch.n - 1; k >= 0; k-- ){		// (the order the kernel computes them in)
					rmd_chain_sib_t	&sb = ch.sib[ k ];
					sb.core_slot = -1;
					if( !sb.leaf )
						continue;
					const int	key = ( int( sb.hmin ) << 20 ) | ( int( sb.lmin ) << 10 ) | int( sb.lmax );
					for( int q = 0; q < 2; q++ )
						if( keys[ q ] == key || keys[ q ] < 0 ){
							keys[ q ] = key;
							sb.core_slot = int8_t( q );
							break;
						}
				}
			}
			// the first helix of the interior and the stem-loop that follows it behind single strands only
			if( e0.head_s >= 0 )
				for( int k = 0; k < ch.n; k++ ){
					if( level_of[ k ] != e0.head_s )
						continue;
					long	glo = 0, ghi = 0;
					for( int j = k + 1; j < ch.n; j++ ){
						const rmd_chain_sib_t	&sj = ch.sib[ j ];
						if( sj.leaf ){
							if( sj.core_slot >= 0 && ghi - glo < 24 && ghi < 1000 ){
								ch.hn_on = 1;
								ch.hn_slot = sj.core_slot;
								ch.hn_tmax = sj.tmax;
								ch.hn_glo = int16_t( glo );
								ch.hn_ghi = int16_t( ghi );
							}
							break;
						}
						if( out->elems[ p->searches[ level_of[ j ] ] ].type != RMA_T_SS )
							break;
						glo += sj.len_lo;
						ghi += sj.len_hi;
					}
					break;
				}
			ch.on = any_leaf && ch.n > 0;
			// Groups that are no stem-loops only shift and smear the vector of the group behind them: runs of them
			// become one group (the sum of their length ranges), and a run in front of the first stem-loop goes into
			// the range of the first element's own lengths -- a pass over the tile and a barrier less for each
			// (trna.descr: eight passes -> five).
			if( ch.on ){
				rmd_chain_sib_t	m[ RMD_MAX_CHAIN ];
				int	nm = 0;
				for( int k = 0; k < ch.n; k++ ){
					const rmd_chain_sib_t	&sb = ch.sib[ k ];
					if( !sb.leaf && nm > 0 && !m[ nm - 1 ].leaf && int( m[ nm - 1 ].len_hi ) + sb.len_hi < 30000 ){
						m[ nm - 1 ].len_lo = int16_t( m[ nm - 1 ].len_lo + sb.len_lo );
						m[ nm - 1 ].len_hi = int16_t( m[ nm - 1 ].len_hi + sb.len_hi );
					}else
						m[ nm++ ] = sb;
				}
				int	first = 0;
				if( nm > 1 && !m[ 0 ].leaf && int( ch.s_hi ) + m[ 0 ].len_hi < 30000 ){
					ch.s_lo = int16_t( ch.s_lo + m[ 0 ].len_lo );
					ch.s_hi = int16_t( ch.s_hi + m[ 0 ].len_hi );
					first = 1;
				}
				ch.n = int8_t( nm - first );
				for( int k = 0; k < ch.n; k++ )
					ch.sib[ k ] = m[ first + k ];
			}
		}
	}
	for( int s = 0; s < p->n_sites; s++ ){
		rmd_site_t	*d = &out->sites[ s ];
		d->n_pos = int8_t( p->sites[ s ].n_pos );
		d->pairset = int8_t( psmap[ p->sites[ s ].pairset ] );
		for( int k = 0; k < 4; k++ ){
			d->elem[ k ] = int8_t( p->sites[ s ].pos[ k ].elem );
			d->l2r[ k ] = int8_t( p->sites[ s ].pos[ k ].l2r );
			d->offset[ k ] = int16_t( p->sites[ s ].pos[ k ].offset );
		}
	}
	for( int k = 0; k < p->n_efn_sites; k++ ){
		out->efn_sites[ k ] = p->efn_sites[ k ];
		// the energy functions walk loops with fixed stacks (rm_efn_core.h rme_ctx_t::stk, rm_efn2_core.h rme2_ctx_t::max_helix):
		// a call over more helices than the usual instance holds takes the instance sized for fifty
		int	lo = p->efn_sites[ k ].idx, hi = p->efn_sites[ k ].idx2, helices = 0;
		for( int d = std::max( 0, std::min( lo, hi ) ); d <= std::max( lo, hi ) && d < p->n_elems; d++ )
			helices += p->elems[ d ].type == RMA_T_H5;
		if( helices > 15 )
			out->efn_big = 1;		// (the energy kernel's instance with stacks for fifty helices)
	}
	// Best literal (optimize_query, compile.c:3315-3392; mm_classccnt, mm_regexp.c:232):
	// the fixed-length seq= with the most "effective characters" whose offset from the
	// start of the motif is bounded.  Necessary condition only, so output neutral.
	out->lit_re = -1;
	{
		double	best = 0;
		int	lmin = 0, lmax = 0;		// offset range of element i from the motif start
		for( int i = 0; i < p->n_elems; i++ ){
			const rma_elem_t	&e = p->elems[ i ];
			if( e.re >= 0 && e.mismatch == 0 && lmax != RMA_UNBOUNDED && e.maxlen != RMA_UNBOUNDED ){
				const rma_regex_t	&re = p->regexes[ e.re ];
				const rmd_regex_t	&dre = out->regexes[ e.re ];
				if( re.fixed_len > 0 && dre.opt == 0 && dre.n_states == re.fixed_len ){
					double	ecnt = 0;
					for( int a = 0; a < re.n_atoms; a++ ){
						int	bits = __builtin_popcount( re.atoms[ a ].mask & 0xf );
						double	c = bits >= 4 ? 0 : bits == 3 ? 0.25 : bits == 2 ? 0.5 : bits == 1 ? 1.0 : 0;
						if( re.atoms[ a ].mask & 0x10 )
							c = 0;		// accepts ambiguity letters too: no information
						ecnt += c * re.atoms[ a ].lo;
					}
					int	n = re.fixed_len;
					int	lo = lmin, hi = lmax;
					if( re.anchored && re.dollar ){
						// whole element
					}else if( re.anchored ){
						// at the element's first base
					}else if( re.dollar ){
						lo = lmin + e.minlen - n;
						hi = lmax + e.maxlen - n;
					}else
						hi = lmax + e.maxlen - n;
					if( ecnt >= 2.5 && ecnt > best && lo >= 0 && hi >= lo && hi - lo < 4096 ){
						best = ecnt;
						out->lit_re = e.re;
						out->lit_lo = lo;
						out->lit_hi = hi;
						// the same literal seen from the end of the first search element's group
						// (its end position is what a rank of the first level fixes): distance
						// from the literal's first base to that end
						out->lit_elo = 0;
						out->lit_ehi = -1;
						const rma_elem_t	&g0 = p->elems[ p->searches[ 0 ] ];
						const int	g_last = g0.n_scopes > 0 ? g0.scopes[ g0.n_scopes - 1 ] : g0.index;
						if( i >= g0.index && i <= g_last ){
							long	smin = 0, smax = 0;
							bool	unb = false;
							for( int j = i + 1; j <= g_last; j++ ){
								smin += p->elems[ j ].minlen;
								if( p->elems[ j ].maxlen == RMA_UNBOUNDED )
									unb = true;
								else
									smax += p->elems[ j ].maxlen;
							}
							long	dlo, dhi;
							if( re.anchored && !re.dollar ){
								dlo = e.minlen - 1 + smin;
								dhi = e.maxlen - 1 + smax;
							}else if( re.dollar ){
								dlo = n - 1 + smin;
								dhi = n - 1 + smax;
							}else{
								dlo = n - 1 + smin;
								dhi = e.maxlen - 1 + smax;
							}
							if( !unb && dhi >= dlo && dhi < 30000 ){
								out->lit_elo = int32_t( dlo );
								out->lit_ehi = int32_t( dhi );
							}
						}
					}
				}
			}
			lmin += e.minlen;
			if( lmax != RMA_UNBOUNDED )
				lmax = e.maxlen == RMA_UNBOUNDED ? RMA_UNBOUNDED : lmax + e.maxlen;
		}
	}
	out->lmargin = 1;
	out->rmargin = 1;
	if( p->has_lctx && p->lctx.re >= 0 )
		out->lmargin = std::max( 1, p->lctx.maxlen );
	if( p->has_rctx && p->rctx.re >= 0 )
		out->rmargin = std::max( 1, 2 * p->rctx.maxlen );
	out->n_regexes = p->n_regexes;
	out->n_rules = n_rules;
	out->n_pairsets = n_ps;
	out->off_regexes = int32_t( offsetof( rmd_program_t, regexes ) );
	out->off_regexes2 = int32_t( offsetof( rmd_program_t, regexes2 ) );
	out->off_rules = int32_t( offsetof( rmd_program_t, rules ) );
	out->off_pairsets = int32_t( offsetof( rmd_program_t, pairsets ) );
	out->off_pks = int32_t( offsetof( rmd_program_t, pks ) );
	out->off_tups = int32_t( offsetof( rmd_program_t, tups ) );
	out->off_sites = int32_t( offsetof( rmd_program_t, sites ) );
	out->off_efn = int32_t( offsetof( rmd_program_t, efn_sites ) );
	out->image_bytes = int32_t( sizeof( rmd_program_t ) );
	return 0;
#undef FAIL
}

size_t rmd_make_image( const rmd_program_t *full, void *img )
{
	char	*out = static_cast<char *>( img );
	auto align = []( size_t v, size_t a ){ return ( v + a - 1 ) / a * a; };
	// everything up to the last element in use ...
	size_t	n = offsetof( rmd_program_t, elems ) + size_t( full->n_elems ) * sizeof( rmd_elem_t );
	memcpy( out, full, n );
	rmd_program_t	*hdr = reinterpret_cast<rmd_program_t *>( out );	// (only its leading members are valid)
	// ... then the pools, most strictly aligned first
	n = align( n, alignof( rmd_regex_t ) );
	hdr->off_regexes = int32_t( n );
	memcpy( out + n, full->regexes, size_t( full->n_regexes ) * sizeof( rmd_regex_t ) );
	n += size_t( full->n_regexes ) * sizeof( rmd_regex_t );
	hdr->off_regexes2 = int32_t( n );		// (same alignment)
	memcpy( out + n, full->regexes2, size_t( full->n_regexes2 ) * sizeof( rmd_regex2_t ) );
	n += size_t( full->n_regexes2 ) * sizeof( rmd_regex2_t );
	n = align( n, alignof( rmd_rule_t ) );
	hdr->off_rules = int32_t( n );
	memcpy( out + n, full->rules, size_t( full->n_rules ) * sizeof( rmd_rule_t ) );
	n += size_t( full->n_rules ) * sizeof( rmd_rule_t );
	n = align( n, alignof( rmd_pairset_t ) );
	hdr->off_pairsets = int32_t( n );
	memcpy( out + n, full->pairsets, size_t( full->n_pairsets ) * sizeof( rmd_pairset_t ) );
	n += size_t( full->n_pairsets ) * sizeof( rmd_pairset_t );
	n = align( n, alignof( rmd_pk_t ) );
	hdr->off_pks = int32_t( n );
	memcpy( out + n, full->pks, size_t( full->n_pks ) * sizeof( rmd_pk_t ) );
	n += size_t( full->n_pks ) * sizeof( rmd_pk_t );
	hdr->off_tups = int32_t( n );
	memcpy( out + n, full->tups, size_t( full->n_tups ) * sizeof( rmd_tup_t ) );
	n += size_t( full->n_tups ) * sizeof( rmd_tup_t );
	n = align( n, alignof( rma_efn_site_t ) );
	hdr->off_efn = int32_t( n );
	memcpy( out + n, full->efn_sites, size_t( full->n_efn ) * sizeof( rma_efn_site_t ) );
	n += size_t( full->n_efn ) * sizeof( rma_efn_site_t );
	n = align( n, alignof( rmd_site_t ) );
	hdr->off_sites = int32_t( n );
	memcpy( out + n, full->sites, size_t( full->n_sites ) * sizeof( rmd_site_t ) );
	n += size_t( full->n_sites ) * sizeof( rmd_site_t );
	n = align( n, 16 );
	hdr->image_bytes = int32_t( n );
	return n;
}

// hostsim_check.cpp -- TEST INFRASTRUCTURE.
//
// Compiles the device search state machine (rnamotif_amd/csrc/rm_scan_core.h)
// for the CPU and checks it, record by record, against the scalar oracle
// (oracle/rm_oracle_scan.c) on a FASTA file.  This is how the state machine is
// debugged in a container without a GPU; the package never loads it.
//
//   hostsim_check [rnamotif options] -descr file.descr db.fastn
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "rm_cli.h"
#include "rm_oracle.h"
#define RMD_FN static inline
#define RMD_FN_MEMBER inline
// (HOSTSIM_BUDGET: iterations a step of the general path may take before it pauses; small values make
// every loop of the generators stop and resume)
static int	hostsim_budget = 0;
#include "rm_scan_core.h"
#include "rm_efn_core.h"
#include "rm_efn2_core.h"
#include "rm_efndata.h"

// strand view for the energy cores: code( p ) of the strand the hit lies on
struct HostSeq {
	const char	*sbuf;
	int	code( int p ) const
	{
		switch( sbuf[ p ] ){
		case 'a' : return 0;
		case 'c' : return 1;
		case 'g' : return 2;
		case 't' : case 'u' : return 3;
		default : return 4;
		}
	}
};

struct VecSink {
	std::vector<int32_t>	*out;
	int	seq, comp, stride;
	void put( const rmd_program_t *P, const rmd_lane_t *L, int szero )
	{
		size_t	o = out->size();
		out->resize( o + stride );
		rmd_fill_hit( P, L, seq, comp, szero, out->data() + o );
	}
};

struct ArrRecs {
	rmd_lrec_t	r[ RMD_LEAN_LEVELS ];
	rmd_lrec_t	get( int k ) const { return r[ k ]; }
	void	set( int k, rmd_lrec_t v ) { r[ k ] = v; }
};

// the general path's records, one per search level (the kernel keeps them packed in LDS)
struct GenRecs {
	rmd_grec_t	r[ RMD_MAX_ELEMS ];
	uint32_t	before[ 2 * RMD_MAX_ELEMS ];
	rmd_grec_t	get( int k ) const { return r[ k ]; }
	void	set( int k, rmd_grec_t v ) { r[ k ] = v; }
	void	set_iter( int k, rmd_grec_t v ) { r[ k ].sd = v.sd; r[ k ].a = v.a; r[ k ].c = v.c; r[ k ].hl = v.hl; r[ k ].ph = v.ph; }
	void	set_iter_words( int k, uint32_t d1, uint32_t d2 ) { rmd_grec_set_words( r[ k ], d1, d2 ); }
	void	set_before( int k, rmd_grec_t v ) { before[ 2 * k ] = rmd_grec_word1( v ); before[ 2 * k + 1 ] = rmd_grec_word2( v ); }
	void	set_window( int k, int zero, int osd ) { r[ k ].zero = int16_t( zero ); r[ k ].osd = int16_t( osd ); }
	void	set_zero( int k, int zero ) { r[ k ].zero = int16_t( zero ); }
	void	set_osd( int k, int osd ) { r[ k ].osd = int16_t( osd ); }
	int	hl( int k ) const { return r[ k ].hl; }
};

// continuations of the split level, as the kernel queues them (every third one is refused, as a
// full queue would: the lane then walks the alternative itself)
struct HostCont { int alt; uint32_t before[ 2 * RMD_MAX_ELEMS ]; };
struct HostSplit {
	int	S;
	std::vector<HostCont>	*q;
	int	*n_offered;
	int	level() const { return S; }
	bool	push( const rmd_gen_t &, GenRecs &gr, int alt ) const
	{
		if( ++*n_offered % 3 == 0 )
			return false;
		HostCont	c;
		c.alt = alt;
		memcpy( c.before, gr.before, sizeof( c.before ) );
		q->push_back( c );
		return true;
	}
};

// rmd_gen_skip_ends()'s accelerator as the kernel defines it (RowEnds, rm_scan_hip.hip), bit by
// bit instead of from bit vectors: the first minlen pairs of (s5, end) within the mispair limit
struct HostEnds {
	// (WIDE: the host runs the two-word sets of helix lengths -- the instance for helices of 64 to 127 base pairs -- on every descriptor)
	static constexpr int	kinds = RMD_KIND_PK | RMD_KIND_TQ | RMD_KIND_WIDE;
	const rmd_program_t	*P;
	rmd_seq_t	sq;
	int	slen;
	bool	ends( const rmd_elem_t &stp, int s5, int top, int lo, uint64_t *mask ) const
	{
		if( stp.rows < 0 || getenv( "HOSTSIM_NOENDS" ) )
			return false;
		const int	lim = ( stp.ends & RMA_5PAIRED ) ? stp.mplim : ( stp.mplim > 1 ? stp.mplim : 1 );
		uint64_t	m = 0;
		for( int i = 0; i < 64; i++ ){
			const int	e = top - 63 + i;
			if( e < lo || e >= slen )
				continue;
			int	mis = 0;
			bool	ok = true;
			for( int h = 0; h < stp.minlen && ok; h++ ){
				const bool	pr = e - h >= 0 && s5 + h < slen &&
					rmd_paired( P, stp.pairset, rmd_code( sq, s5 + h ), rmd_code( sq, e - h ) );
				if( !pr ){
					mis++;
					if( h == 0 && ( stp.ends & RMA_5PAIRED ) )
						ok = false;
				}
			}
			if( ok && mis <= lim )
				m |= 1ull << i;
		}
		*mask = m;
		return true;
	}
};

// one work item, through the lean path when the descriptor allows it
static void sim_item( const rmd_program_t *dp, rmd_lane_t *lane, const rmd_seq_t &sq, int szero, int slen, int r0, int cnt, VecSink &sink )
{
	if( dp->lean_ok && !getenv( "HOSTSIM_NOLEAN" ) ){
		ArrRecs	recs;
		rmd_lean_t	st;
		int	k = rmd_lean_begin( dp, recs, st, szero, slen, r0, cnt );
		// (the kernels' order words, where the descriptor allows them -- rmd_elem_t::ord_stride: the number of the walk's
		// choices -- must grow from one candidate of a (start, rank) to the next as this walk finds them; checked here,
		// for every candidate, as wave_emit computes them from the records)
		long long	last_key = -1;
		int	last_rank = -1;
		while( k >= 0 ){
			const size_t	had = sink.out->size();
			k = rmd_lean_step( dp, recs, st, sq, k, lane, sink );
			if( dp->ord_ok && sink.out->size() > had ){
				long long	key = 0;
				for( int kk = 0; kk < dp->n_searches; kk++ ){
					const rmd_lrec_t	r = recs.get( kk );
					const rmd_elem_t	&stp = dp->elems[ dp->searches[ kk ] ];
					const int	first = ( kk == 0 || !stp.loop ) ? int( r.sd ) + 1 : int( rmd_lean_open( dp, kk, r.zero, r.osd ).sd );
					const long long	digit = ( long long )( first - ( int( r.sd ) + 1 ) ) * stp.ord_nlen + ( stp.type == RMA_T_SS ? 0 : int( r.hl ) - stp.minlen );
					if( digit < 0 ){
						fprintf( stderr, "hostsim: negative order digit at level %d (start %d)\n", kk, szero );
						exit( 3 );
					}
					key += digit * stp.ord_stride;
				}
				if( key >= ( 1ll << 31 ) || ( st.rank == last_rank && key <= last_key ) ){
					fprintf( stderr, "hostsim: order word %lld after %lld at start %d rank %d\n", key, last_key, szero, st.rank );
					exit( 3 );
				}
				last_key = key;
				last_rank = st.rank;
			}
		}
	}else{
		GenRecs	recs;
		memset( &recs, 0x55, sizeof( recs ) );	// (windows are written before they are read: any garbage must do)
		HostEnds	ends{ dp, sq, slen };
		if( dp->split_s < 0 || getenv( "HOSTSIM_NOSPLIT" ) ){
			rmd_gen_position( dp, recs, lane, sq, szero, slen, r0, cnt, sink, ends );
			return;
		}
		// as the kernel does it: the first levels now, what they hand over afterwards; then the
		// candidates of the item in the order the host restores (rank, alternative, emission)
		std::vector<int32_t>	mine;
		VecSink	tmp{ &mine, sink.seq, sink.comp, sink.stride };
		std::vector<HostCont>	conts;
		int	n_offered = 0;
		HostSplit	split{ dp->split_s, &conts, &n_offered };
		rmd_gen_t	st;
		int	k = rmd_gen_begin( dp, recs, st, szero, slen, r0, cnt );
		while( k >= 0 )
			k = rmd_gen_step( dp, recs, st, sq, k, lane, tmp, ends, split );
		for( const HostCont &c : conts ){
			memset( &recs, 0x55, sizeof( recs ) );
			k = rmd_gen_resume( dp, recs, st, sq, szero, slen, r0, cnt, dp->split_s, c.before, c.alt, ends );
			while( k > dp->split_s )
				k = rmd_gen_step( dp, recs, st, sq, k, lane, tmp, ends, rmd_no_split_t(), dp->split_s );
		}
		const int	stride = sink.stride;
		std::vector<int>	idx( mine.size() / stride );
		for( size_t i = 0; i < idx.size(); i++ )
			idx[ i ] = int( i );
		std::stable_sort( idx.begin(), idx.end(), [&]( int x, int y ){
			const int32_t	*a = &mine[ size_t( x ) * stride ], *b = &mine[ size_t( y ) * stride ];
			return a[ 3 ] != b[ 3 ] ? a[ 3 ] < b[ 3 ] : a[ 4 ] < b[ 4 ]; } );
		int	prev_rank = -1, order = 0;
		for( int i : idx ){
			int32_t	*w = &mine[ size_t( i ) * stride ];
			order = w[ 3 ] == prev_rank ? order + 1 : 0;
			prev_rank = w[ 3 ];
			w[ 4 ] = order;
			sink.out->insert( sink.out->end(), w, w + stride );
		}
	}
}

static void sim_scan( const rmd_program_t *dp, int seq, const char *sbuf, int slen, int comp, std::vector<int32_t> &out )
{
	std::vector<uint8_t>	codes( slen + 1 );
	for( int i = 0; i < slen; i++ ){
		switch( sbuf[ i ] ){
		case 'a' : codes[ i ] = 0; break;
		case 'c' : codes[ i ] = 1; break;
		case 'g' : codes[ i ] = 2; break;
		case 't' : case 'u' : codes[ i ] = 3; break;
		default : codes[ i ] = 4; break;
		}
	}
	rmd_seq_t	sq{ codes.data(), 0 };
	rmd_lane_t	lane;
	VecSink	sink{ &out, seq, comp, dp->hit_stride };
	// same decomposition as the kernel: a pre-filter over the end positions of the
	// first element where it is a proper helix or a 4-plex, one item per survivor
	const rmd_elem_t	&e0 = dp->elems[ dp->searches[ 0 ] ];
	bool	quick = ( e0.type == RMA_T_H5 && e0.proper ) || e0.type == RMA_T_Q1;
	int	i_minl0 = e0.minilen;
	if( e0.type == RMA_T_Q1 )
		i_minl0 += dp->elems[ e0.mates[ 0 ] ].minilen + dp->elems[ e0.mates[ 1 ] ].minilen + 2 * e0.minlen;
	for( int szero = 0; szero <= slen - dp->dminlen; szero++ ){
		if( dp->lit_re >= 0 && !getenv( "HOSTSIM_NOQUICK" ) ){
			// best-literal filter, as in the kernel
			const rmd_regex_t	&lre = rmd_regexes( dp )[ dp->lit_re ];
			int	n = lre.n_states, hi = std::min( dp->lit_hi, dp->w_winsize - n );
			bool	found = false;
			for( int q = szero + dp->lit_lo; q <= szero + hi && !found; q++ ){
				if( q + n > slen )
					break;
				bool	ok = true;
				for( int j = 0; ok && j < n; j++ )
					ok = ( lre.accept[ codes[ q + j ] ] >> j ) & 1;
				found = ok;
			}
			if( !found )
				continue;
		}
		if( !quick || getenv( "HOSTSIM_NOQUICK" ) ){
			bool	at_szero = e0.type == RMA_T_P5 || e0.type == RMA_T_T1 || e0.type == RMA_T_Q1 ||
				( e0.type == RMA_T_H5 && ( e0.proper || e0.scope == 0 ) );
			if( !getenv( "HOSTSIM_NOQUICK" ) && at_szero && !rmd_prefix_ok( dp, e0, sq, szero ) )
				continue;
			sim_item( dp, &lane, sq, szero, slen, 0, RMD_ALL_RANKS, sink );
			continue;
		}
		int	hi, lo;
		rmd_level0_range( dp, szero, slen, &hi, &lo );
		for( int sd = hi; sd >= lo; sd-- ){
			int	s3lim = rmd_s3lim( szero, sd, i_minl0, e0.maxlen );
			if( rmd_quick_wchlx( dp, e0, sq, szero, sd, s3lim ) )
				sim_item( dp, &lane, sq, szero, slen, hi - sd, 1, sink );
		}
	}
}

int main( int argc, char **argv )
{
	try{
		if( getenv( "HOSTSIM_BUDGET" ) )
			hostsim_budget = std::max( 4, atoi( getenv( "HOSTSIM_BUDGET" ) ) );	// (rm_scan_core.h: at least 4)
		rma::Args	args = rma::parse_args( argc, argv );
		rma::Prepared	pr = rma::prepare( args );
		fprintf( stderr, "%s: %d elements, %d searches, stride %d\n", args.dfname.c_str(),
			pr.prog->n_elems, pr.prog->n_searches, rma_hit_stride( pr.prog.get() ) );
		rmd_program_t	dp;
		char	err[ 512 ];
		if( rmd_build( pr.prog.get(), &dp, err, sizeof( err ) ) ){
			fprintf( stderr, "rmd_build: %s\n", err );
			return 2;
		}
		if( hostsim_budget > 0 )
			dp.step_budget = hostsim_budget;
		int	stride = dp.hit_stride, n_cmp = rma_hit_efn_off( pr.prog.get() );
		int64_t	total = 0, bad = 0, n_efn2 = 0, n_efn = 0;
		// efn()'s tables as the energy kernel gets them (rma::efn_tables16)
		std::vector<int16_t>	t16;
		std::vector<int32_t>	tlkey;
		if( pr.efn )
			rma::efn_tables16( pr.efn.get(), t16, tlkey );
		const rme_tables_t	T16{ t16.data(), tlkey.data(), pr.efn ? pr.efn->loginc : nullptr };
		int	seq = 0;
		for( const std::string &fn : args.dbfnames ){
			FILE	*fp = fopen( fn.c_str(), "r" );
			if( !fp ){ perror( fn.c_str() ); return 2; }
			rma::FastaReader	rd( fp );
			rma::SeqRecord	rec;
			while( rd.next( rec ) ){
				std::vector<char>	buf( rec.seq.begin(), rec.seq.end() );
				buf.push_back( 0 );
				int	slen = int( rec.seq.size() );
				for( int comp = 0; comp < ( pr.prog->chk_both_strs ? 2 : 1 ); comp++ ){
					if( comp )
						rmo_revcomp( buf.data(), slen );
					rmo_hits_t	oh;
					rmo_hits_init( &oh, pr.prog.get() );
					rmo_set_efn2data( pr.efn2.get() );
					rmo_scan( pr.prog.get(), pr.efn.get(), seq, buf.data(), slen, comp, &oh );
					std::vector<int32_t>	sh;
					sim_scan( &dp, seq, buf.data(), slen, comp, sh );
					int64_t	ns = int64_t( sh.size() ) / stride;
					total += oh.n;
					if( ns != oh.n ){
						if( bad < 10 )
							fprintf( stderr, "seq %d (%s) comp %d: oracle %lld hits, sim %lld\n", seq, rec.sid.c_str(), comp, ( long long )oh.n, ( long long )ns );
						bad++;
					}
					for( int64_t h = 0; h < std::min( ns, oh.n ); h++ ){
						if( memcmp( oh.data + h * stride, sh.data() + h * stride, n_cmp * sizeof( int32_t ) ) ){
							if( bad < 10 ){
								fprintf( stderr, "seq %d (%s) comp %d hit %lld differs:\n  oracle:", seq, rec.sid.c_str(), comp, ( long long )h );
								for( int k = 0; k < n_cmp; k++ ) fprintf( stderr, " %d", oh.data[ h * stride + k ] );
								fprintf( stderr, "\n  sim   :" );
								for( int k = 0; k < n_cmp; k++ ) fprintf( stderr, " %d", sh[ h * stride + k ] );
								fprintf( stderr, "\n" );
							}
							bad++;
							break;
						}
						// efn() sites: rm_efn_core.h compiled for the host, with the tables as the kernel has them,
						// without and with the per-lane cache of base codes and partners
						for( int k = 0; pr.efn && k < dp.n_efn; k++ ){
							if( dp.efn_sites[ k ].kind == RMA_EFN_KIND_EFN2 )
								continue;
							HostSeq	hs{ buf.data() };
							int16_t	bpbuf[ 97 ];
							uint8_t	bcbuf[ 100 ];
							const int	got = rme_site_energy( &dp, &T16, &hs, sh.data() + h * stride, k );
							const int	got_c = rme_site_energy( &dp, &T16, &hs, sh.data() + h * stride, k, bpbuf, bcbuf, 96 );
							const int	want = oh.data[ h * stride + n_cmp + k ];
							n_efn++;
							if( got != want || got_c != want ){
								if( bad < 10 ){
									fprintf( stderr, "seq %d comp %d hit %lld efn site %d: oracle %d, device core %d, with cache %d\n  record:", seq, comp, ( long long )h, k, want, got, got_c );
									for( int q = 0; q < n_cmp; q++ ) fprintf( stderr, " %d", oh.data[ h * stride + q ] );
									fprintf( stderr, "\n" );
								}
								bad++;
							}
						}
						// efn2() sites: the device core, compiled for the host, against the oracle's value
						for( int k = 0; pr.efn2 && k < dp.n_efn; k++ ){
							if( dp.efn_sites[ k ].kind != RMA_EFN_KIND_EFN2 )
								continue;
							HostSeq	hs{ buf.data() };
							const int	got = rme2_site_energy( &dp, pr.efn2.get(), &hs, sh.data() + h * stride, k );
							const int	want = oh.data[ h * stride + n_cmp + k ];
							n_efn2++;
							if( got != want ){
								if( bad < 10 )
									fprintf( stderr, "seq %d comp %d hit %lld efn2 site %d: oracle %d, device core %d\n", seq, comp, ( long long )h, k, want, got );
								bad++;
							}
						}
					}
					rmo_hits_free( &oh );
				}
				seq++;
			}
			fclose( fp );
		}
#ifdef RMD_STATS
		fprintf( stderr, "stats: match_wchlx %lld\n", rmd_stat[ 4 ] );
#endif
		printf( "%s: %lld candidates, %lld mismatching strands", args.dfname.c_str(), ( long long )total, ( long long )bad );
		if( n_efn2 > 0 )
			printf( " (%lld efn2 energies compared)", ( long long )n_efn2 );
		if( n_efn > 0 )
			printf( " (%lld efn energies compared)", ( long long )n_efn );
		printf( "\n" );
		return bad ? 1 : 0;
	}catch( rma::Error &e ){
		fprintf( stderr, "%s\n", e.what() );
		return 2;
	}
}

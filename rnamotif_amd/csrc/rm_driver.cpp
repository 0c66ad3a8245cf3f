// rm_driver.cpp -- see rm_driver.h.
#include "rm_driver.h"
#include "rm_score.h"
#include "rm_pack.h"
#include <cstring>

namespace rma {

Replayer::Replayer( Descriptor &d, const rma_program_t &prog, FILE *out )
	: d_( d ), prog_( prog ), out_( out ), printer_( d, out )
{
	d_.score->out = out;
}

void Replayer::begin()
{
	d_.score->setprog( P_BEGIN );
	d_.score->run( 0, 0, nullptr, nullptr, nullptr );
	d_.score->setprog( P_MAIN );
}

void Replayer::end()
{
	d_.score->setprog( P_END );
	d_.score->run( 0, 0, nullptr, nullptr, nullptr );
	d_.score->setprog( P_MAIN );
}

static void revcomp( std::string &s )	// mk_rcmp, rnamot.c:193-216
{
	auto cmp = []( char c ) -> char {
		switch( c ){
		case 'a' : case 'A' : return 't';
		case 'c' : case 'C' : return 'g';
		case 'g' : case 'G' : return 'c';
		case 't' : case 'T' : case 'u' : case 'U' : return 'a';
		default : return 'n';
		}
	};
	std::string	r( s.size(), 'n' );
	for( size_t i = 0, n = s.size(); i < n; i++ )
		r[ n - 1 - i ] = cmp( s[ i ] );
	s.swap( r );
}

void Replayer::replay( const std::vector<SeqRecord> &batch, const int32_t *hits, int64_t n, SearchStats &st )
{
	int	stride = rma_hit_stride( &prog_ );
	int	ctx_off = rma_hit_ctx_off( &prog_ ), efn_off = rma_hit_efn_off( &prog_ );
	int	cur_seq = -1;
	std::string	rc;
	for( int64_t h = 0; h < n; h++ ){
		const int32_t	*w = hits + h * stride;
		int	seq = w[ 0 ], comp = w[ 1 ];
		if( seq < 0 || seq >= int( batch.size() ) )
			fail( "scanner returned a hit for sequence %d of a batch of %d.", seq, int( batch.size() ) );
		const SeqRecord	&rec = batch[ seq ];
		if( comp && seq != cur_seq ){
			rc = rec.seq;
			revcomp( rc );
			cur_seq = seq;
		}
		const char	*sbuf = comp ? rc.c_str() : rec.seq.c_str();
		int	slen = int( rec.seq.size() );
		// restore what the search leaves in rm_descr[] at the end of the search list
		for( int e = 0; e < prog_.n_elems; e++ ){
			Strel	&s = d_.descr[ e ];
			s.matchoff = w[ RMA_HIT_HDR + 4 * e ];
			s.matchlen = w[ RMA_HIT_HDR + 4 * e + 1 ];
			s.n_mispairs = w[ RMA_HIT_HDR + 4 * e + 2 ];
			s.n_mismatches = w[ RMA_HIT_HDR + 4 * e + 3 ];
		}
		if( d_.lctx ){
			d_.lctx->matchoff = w[ ctx_off ];
			d_.lctx->matchlen = w[ ctx_off + 1 ];
		}
		if( d_.rctx ){
			d_.rctx->matchoff = w[ ctx_off + 2 ];
			d_.rctx->matchlen = w[ ctx_off + 3 ];
		}
		// find_ss, find_motif.c:373-392
		d_.nval->pval = ( void * )rec.sid.c_str();
		d_.cval->ival = comp;
		d_.pval->ival = comp ? slen - d_.descr[ 0 ].matchoff : d_.descr[ 0 ].matchoff + 1;
		int	len = 0;
		for( int e = 0; e < prog_.n_elems; e++ )
			len += d_.descr[ e ].matchlen;
		d_.lval->ival = len;
		st.n_candidates++;
		Ident	*h_id = nullptr;
		if( d_.score->run( comp, slen, sbuf, &h_id, prog_.n_efn_sites ? w + efn_off : nullptr ) != SA_REJECT ){
			printer_.print( rec.sid.c_str(), rec.sdef.c_str(), comp, slen, sbuf, h_id );
			st.n_hits++;
		}
	}
}

int run_search( Descriptor &d, const rma_program_t &prog, ScanBackend &be, FILE *out,
	int64_t batch_bases, SearchStats *stats )
{
	SearchStats	st;
	Replayer	rp( d, prog, out );
	rp.begin();
	std::vector<FILE *>	files;
	bool	use_stdin = d.args.dbfnames.empty();
	size_t	nfiles = use_stdin ? 1 : d.args.dbfnames.size();
	std::vector<SeqRecord>	batch;
	int64_t	in_batch = 0;
	char	err[ 1024 ];
	int	ecnt = 0;
	const int	show_progress = d.int_global( "show_progress", 0 );
	auto flush = [&](){
		if( batch.empty() )
			return;
		std::vector<const char *>	seqs;
		std::vector<int32_t>	slens;
		for( const SeqRecord &r : batch ){
			seqs.push_back( r.seq.c_str() );
			slens.push_back( int32_t( r.seq.size() ) );
		}
		const int32_t	*hits = nullptr;
		int64_t	n_hits = 0;
		err[ 0 ] = '\0';
		if( be.scan( be.self, seqs.data(), slens.data(), int( batch.size() ), &hits, &n_hits, err, sizeof( err ) ) )
			fail( "scan failed: %s", err );
		rp.replay( batch, hits, n_hits, st );
		batch.clear();
		in_batch = 0;
	};
	// A packed database (rm_pack.h) takes the place of a text file: same entries, same order.
	auto scan_pack = [&]( const std::string &path ){
		PackFile	pk;
		std::string	perr;
		if( !pk.load( path, perr ) )
			fail( "%s", perr.c_str() );
		int	first = 0;
		while( first < pk.count() ){
			int	count = 0;
			int64_t	bases = 0;
			while( first + count < pk.count() && ( count == 0 || bases < batch_bases ) ){
				bases += pk.slen[ first + count ];
				count++;
			}
			for( int i = 0; i < count; i++ ){
				ecnt++;
				if( show_progress > 0 && ecnt % show_progress == 0 )
					fprintf( stderr, "%s: %7d: %s\n", d.args.argv0.c_str(), ecnt, pk.sid( first + i ) );
			}
			st.n_seqs += count;
			st.n_bases += bases;
			const int32_t	*hits = nullptr;
			int64_t	n_hits = 0;
			err[ 0 ] = '\0';
			std::vector<SeqRecord>	recs;
			recs.resize( size_t( count ) );
			if( be.scan_packed != nullptr ){
				if( be.scan_packed( be.self, &pk, first, count, &hits, &n_hits, err, sizeof( err ) ) )
					fail( "scan failed: %s", err );
			}else{
				std::vector<const char *>	seqs;
				std::vector<int32_t>	slens;
				for( int i = 0; i < count; i++ ){
					recs[ i ].seq = pk.unpack( first + i );
					seqs.push_back( recs[ i ].seq.c_str() );
					slens.push_back( pk.slen[ first + i ] );
				}
				if( be.scan( be.self, seqs.data(), slens.data(), count, &hits, &n_hits, err, sizeof( err ) ) )
					fail( "scan failed: %s", err );
			}
			// text only for the entries that have candidates
			const int	stride = rma_hit_stride( &prog );
			for( int64_t h = 0; h < n_hits; h++ ){
				const int	i = hits[ h * stride ];
				if( i < 0 || i >= count )
					fail( "scanner returned a hit for sequence %d of a batch of %d.", i, count );
				SeqRecord	&r = recs[ i ];
				if( r.sid.empty() ){
					r.sid = pk.sid( first + i );
					r.sdef = pk.sdef( first + i );
					if( r.seq.empty() )
						r.seq = pk.unpack( first + i );
				}
			}
			rp.replay( recs, hits, n_hits, st );
			first += count;
		}
	};
	for( size_t f = 0; f < nfiles; f++ ){
		FILE	*fp = stdin;
		if( !use_stdin && PackFile::is_pack( d.args.dbfnames[ f ] ) ){
			flush();
			scan_pack( d.args.dbfnames[ f ] );
			if( f + 1 < nfiles )
				ecnt++;		// the EOF that switches files is counted, rnamot.c:160-168
			continue;
		}
		if( !use_stdin ){
			fp = fopen( d.args.dbfnames[ f ].c_str(), "r" );
			if( fp == nullptr ){
				// DB_fnext, dbutil.c:12-40: report and stop
				fprintf( stderr, "DB_fnext: can't read seq file '%s'.\n", d.args.dbfnames[ f ].c_str() );
				break;
			}
		}
		FastaReader	rd( fp, d.args.maxslen, seq_format_of( d.args.dbfmt ) );
		SeqRecord	rec;
		for( ; ; ){
			const bool	got = rd.next( rec );
			// rnamot.c:160-176: the entry counter also ticks for the EOF that switches to
			// the next file (with an empty name), not for the one that ends the run
			if( !got && ( use_stdin || f + 1 >= nfiles ) )
				break;
			ecnt++;
			if( show_progress > 0 && ecnt % show_progress == 0 )
				fprintf( stderr, "%s: %7d: %s\n", d.args.argv0.c_str(), ecnt, got ? rec.sid.c_str() : "" );
			if( !got )
				break;
			st.n_seqs++;
			st.n_bases += int64_t( rec.seq.size() );
			in_batch += int64_t( rec.seq.size() );
			batch.push_back( std::move( rec ) );
			if( in_batch >= batch_bases )
				flush();
		}
		if( fp != stdin )
			fclose( fp );
	}
	flush();
	rp.end();
	if( stats )
		*stats = st;
	return 0;
}

}	// namespace rma

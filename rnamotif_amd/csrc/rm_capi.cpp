// rm_capi.cpp -- host half of the C ABI (include/rnamotif_amd.h): descriptor
// compilation and candidate replay.  The scanner/database half lives in
// rm_scan_hip.hip.
#include "rnamotif_amd.h"
#include "rm_cli.h"
#include "rm_efndata.h"
#include "rm_pack.h"
#include "rm_stream.h"
#include "rm_hitsort.h"
#include <cctype>
#include <cstddef>
#include <cstring>
#include <cstdlib>
#include <memory>

struct rma_descr {
	rma::Prepared	pr;
};

struct rma_replay {
	rma_descr	*d;
	FILE	*fp;
	bool	own_fp;
	rma::Replayer	*rp;
	rma::SearchStats	st;
};

static int set_err( char *err, size_t errlen, const char *msg )
{
	if( err != nullptr && errlen > 0 ){
		strncpy( err, msg, errlen - 1 );
		err[ errlen - 1 ] = '\0';
	}
	return 1;
}

extern "C" const char *rma_version( void ) { return "rnamotif_amd 0.1 (rnamotif v3.1.1 scan path for gfx950)"; }

extern "C" int rma_descr_compile( int argc, const char *const *argv, rma_descr_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	try{
		rma::Args	args = rma::parse_args( argc, const_cast<char **>( argv ) );
		if( !args.have_dfname && !args.have_xdfname )
			return set_err( err, errlen, "no descriptor: use -descr file or -xdescr file" );
		rma_descr	*d = new rma_descr;
		try{
			d->pr = rma::prepare( args );
		}catch( ... ){
			delete d;
			throw;
		}
		*out = d;
		return 0;
	}catch( rma::Error &e ){
		return set_err( err, errlen, e.what() );
	}catch( std::exception &e ){
		return set_err( err, errlen, e.what() );
	}
}

extern "C" void rma_descr_free( rma_descr_t *d ) { delete d; }
extern "C" const rma_efn2data_t *rma_descr_efn2data( const rma_descr_t *d ) { return d->pr.efn2.get(); }
extern "C" const rma_program_t *rma_descr_program( const rma_descr_t *d ) { return d->pr.prog.get(); }
extern "C" const rma_efndata_t *rma_descr_efndata( const rma_descr_t *d ) { return d->pr.efn.get(); }
extern "C" int rma_descr_minlen( const rma_descr_t *d ) { return d->pr.descr->dminlen; }
extern "C" int rma_descr_maxlen( const rma_descr_t *d ) { return d->pr.descr->dmaxlen; }

extern "C" void rma_program_info( const rma_program_t *p, int32_t info[ 8 ] )
{
	info[ 0 ] = p->n_elems;
	info[ 1 ] = p->n_searches;
	info[ 2 ] = rma_hit_stride( p );
	info[ 3 ] = rma_hit_ctx_off( p );
	info[ 4 ] = rma_hit_efn_off( p );
	info[ 5 ] = p->n_efn_sites;
	info[ 6 ] = p->chk_both_strs;
	info[ 7 ] = p->windowsize;
}

extern "C" int rma_replay_open( rma_descr_t *d, const char *path, rma_replay_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	try{
		rma_replay	*rp = new rma_replay;
		rp->d = d;
		if( path == nullptr || !strcmp( path, "-" ) ){
			rp->fp = stdout;
			rp->own_fp = false;
		}else{
			rp->fp = fopen( path, "w" );
			rp->own_fp = true;
			if( rp->fp == nullptr ){
				delete rp;
				return set_err( err, errlen, "can't open the output file" );
			}
		}
		rp->rp = new rma::Replayer( *d->pr.descr, *d->pr.prog, rp->fp );
		rp->rp->begin();
		*out = rp;
		return 0;
	}catch( rma::Error &e ){
		return set_err( err, errlen, e.what() );
	}catch( std::exception &e ){
		return set_err( err, errlen, e.what() );
	}
}

extern "C" int rma_replay_batch( rma_replay_t *rp, const char *const *sids, const char *const *sdefs,
	const char *const *seqs, const int32_t *slens, int32_t n,
	const int32_t *hits, int64_t n_hits, int64_t *n_printed, char *err, size_t errlen )
{
	try{
		std::vector<rma::SeqRecord>	batch( n );
		for( int i = 0; i < n; i++ ){
			batch[ i ].sid = sids[ i ];
			batch[ i ].sdef = sdefs[ i ];
			batch[ i ].seq.assign( seqs[ i ], size_t( slens[ i ] < 0 ? 0 : slens[ i ] ) );
		}
		int64_t	before = rp->st.n_hits;
		rp->rp->replay( batch, hits, n_hits, rp->st );
		if( n_printed )
			*n_printed = rp->st.n_hits - before;
		fflush( rp->fp );
		return 0;
	}catch( rma::Error &e ){
		return set_err( err, errlen, e.what() );
	}
}

struct rma_pack {
	rma::PackFile	pf;
};

extern "C" int rma_replay_pack( rma_replay_t *rp, const rma_pack_t *pk, int32_t first,
	const int32_t *hits, int64_t n_hits, int64_t *n_printed, char *err, size_t errlen )
{
	try{
		int64_t	before = rp->st.n_hits;
		rp->rp->replay_packed( pk->pf, first, hits, n_hits, rp->st );
		if( n_printed )
			*n_printed = rp->st.n_hits - before;
		fflush( rp->fp );
		return 0;
	}catch( rma::Error &e ){
		return set_err( err, errlen, e.what() );
	}
}

extern "C" int rma_sort_hits( const int32_t *hits, int64_t n_hits, int32_t stride, int32_t *out, char *err, size_t errlen )
{
	if( n_hits < 0 || n_hits > int64_t( 0xffffffffu ) || stride < 5 || ( n_hits && ( !hits || !out ) ) )
		return set_err( err, errlen, "rma_sort_hits: records of at least 5 words, at most 2^32-1 of them" );
	try{
		std::vector<rma::HitKey>	keys, tmp;
		rma::sort_hits( hits, n_hits, stride, out, keys, tmp );
		return 0;
	}catch( std::exception &e ){
		return set_err( err, errlen, e.what() );
	}
}

extern "C" int rma_replay_close( rma_replay_t *rp, char *err, size_t errlen )
{
	int	rv = 0;
	try{
		rp->rp->end();
	}catch( rma::Error &e ){
		rv = set_err( err, errlen, e.what() );
	}
	fflush( rp->fp );
	if( rp->own_fp )
		fclose( rp->fp );
	delete rp->rp;
	delete rp;
	return rv;
}

// ---------------------------------------------------------------- packed database

extern "C" int rma_pack_write( const char *path, const char *const *sids, const char *const *sdefs,
	const char *const *seqs, const int32_t *slens, int32_t n, char *err, size_t errlen )
{
	rma::PackFile	pf;
	rma::SeqRecord	rec;
	for( int i = 0; i < n; i++ ){
		rec.sid = sids && sids[ i ] ? sids[ i ] : "";
		rec.sdef = sdefs && sdefs[ i ] ? sdefs[ i ] : "";
		rec.seq.assign( seqs[ i ], size_t( slens[ i ] < 0 ? 0 : slens[ i ] ) );
		for( char &c : rec.seq ){		// what the readers do, dbutil.c:112-113
			c = char( tolower( ( unsigned char )c ) );
			if( c == 'u' )
				c = 't';
		}
		pf.add( rec );
	}
	std::string	e;
	if( !pf.save( path, e ) )
		return set_err( err, errlen, e.c_str() );
	return 0;
}

extern "C" int rma_pack_open( const char *path, rma_pack_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	rma_pack	*pk = new rma_pack;
	std::string	e;
	try{
		if( !pk->pf.load( path, e ) ){
			delete pk;
			return set_err( err, errlen, e.c_str() );
		}
	}catch( std::exception &x ){
		delete pk;
		return set_err( err, errlen, x.what() );
	}
	*out = pk;
	return 0;
}

extern "C" int rma_pack_read( const char *const *paths, int32_t n_paths, const char *fmt, int32_t maxslen, int32_t threads,
	rma_pack_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	rma_pack	*pk = new rma_pack;
	try{
	const int	lim = maxslen > 0 ? maxslen + 1 : 30000000 + 1;		// getargs.c: -N n reads n letters
	const rma::SeqFormat	sf = rma::seq_format_of( fmt ? fmt : "" );
	for( int f = 0; f < n_paths; f++ ){
		const std::string	path = paths[ f ];
		if( rma::PackFile::is_pack( path ) ){
			rma::PackFile	one;
			std::string	e;
			if( !one.load( path, e ) ){
				delete pk;
				return set_err( err, errlen, e.c_str() );
			}
			for( int i = 0; i < one.count(); i++ ){
				const int64_t	w1 = one.base_off[ i ] / 32, nw1 = ( int64_t( one.slen[ i ] ) + 31 ) / 32;
				const int64_t	x0 = one.exc_off[ i ], x1 = i + 1 < one.count() ? one.exc_off[ i + 1 ] : int64_t( one.exc.size() );
				pk->pf.append_packed( one.sid( i ), one.sdef( i ),
					std::vector<uint32_t>( one.codes.begin() + 2 * w1, one.codes.begin() + 2 * ( w1 + nw1 ) ),
					std::vector<uint32_t>( one.amask.begin() + w1, one.amask.begin() + w1 + nw1 ),
					std::vector<char>( one.exc.begin() + x0, one.exc.begin() + x1 ), one.slen[ i ] );
			}
			continue;
		}
		long	resume_at = 0;
		if( sf == rma::FMT_FASTN ){
			rma::FastaStream	fs;
			if( fs.open( path, lim, threads > 0 ? threads : 8 ) ){
				while( std::unique_ptr<rma::PackFile> b = fs.next( int64_t( 1 ) << 40 ) ){
					if( pk->pf.count() == 0 )
						pk->pf = std::move( *b );
					else for( int i = 0; i < b->count(); i++ ){
						const int64_t	w1 = b->base_off[ i ] / 32, nw1 = ( int64_t( b->slen[ i ] ) + 31 ) / 32;
						const int64_t	x0 = b->exc_off[ i ], x1 = i + 1 < b->count() ? b->exc_off[ i + 1 ] : int64_t( b->exc.size() );
						pk->pf.append_packed( b->sid( i ), b->sdef( i ),
							std::vector<uint32_t>( b->codes.begin() + 2 * w1, b->codes.begin() + 2 * ( w1 + nw1 ) ),
							std::vector<uint32_t>( b->amask.begin() + w1, b->amask.begin() + w1 + nw1 ),
							std::vector<char>( b->exc.begin() + x0, b->exc.begin() + x1 ), b->slen[ i ] );
					}
				}
				if( fs.stopped_at() < 0 )
					continue;
				resume_at = long( fs.stopped_at() );
			}
		}
		FILE	*fp = fopen( path.c_str(), "r" );
		if( fp == nullptr ){
			// DB_fnext, dbutil.c:33-37: report and stop reading
			fprintf( stderr, "DB_fnext: can't read seq file '%s'.\n", path.c_str() );
			break;
		}
		if( resume_at > 0 )
			fseek( fp, resume_at, SEEK_SET );
		rma::FastaReader	rd( fp, lim, sf );
		rma::SeqRecord	rec;
		while( rd.next( rec ) )
			pk->pf.add( rec );
		fclose( fp );
	}
	}catch( std::exception &x ){
		delete pk;
		return set_err( err, errlen, x.what() );
	}
	*out = pk;
	return 0;
}

// ---------------------------------------------------------------- a rank's share of a database
// What every rank of a multi-GPU search can know of the database without reading it: its entries and
// an upper bound of each one's length -- from the tables of a packed database, or from the '>' of a
// FASTA file (found by all threads at once; nothing is parsed).
extern "C" int rma_database_index( const char *const *paths, int32_t n_paths, const char *fmt, int32_t threads,
	int64_t **extent, int32_t *n_entries, char *err, size_t errlen )
{
	*extent = nullptr;
	*n_entries = 0;
	try{
		if( rma::seq_format_of( fmt ? fmt : "" ) != rma::FMT_FASTN )
			return 2;		// (pir, gb: through the serial readers only)
		std::vector<int64_t>	ext;
		for( int f = 0; f < n_paths; f++ ){
			const std::string	path = paths[ f ];
			if( rma::PackFile::is_pack( path ) ){
				rma::PackFile	pf;
				std::string	e;
				if( !pf.open( path, e ) )
					return set_err( err, errlen, e.c_str() );
				for( int i = 0; i < pf.count(); i++ )
					ext.push_back( pf.slen[ i ] );
				continue;
			}
			rma::FastaStream	fs;
			if( !fs.open( path, 30000001, threads > 0 ? threads : 8 ) )
				return 2;
			for( size_t i = 0; i < fs.n_entries(); i++ )
				ext.push_back( fs.extent( i ) );
		}
		if( ext.size() > size_t( 0x7fffffff ) )
			return 2;
		int64_t	*out = static_cast<int64_t *>( malloc( std::max<size_t>( ext.size(), 1 ) * sizeof( int64_t ) ) );
		if( out == nullptr )
			return set_err( err, errlen, "out of memory" );
		memcpy( out, ext.data(), ext.size() * sizeof( int64_t ) );
		*extent = out;
		*n_entries = int32_t( ext.size() );
		return 0;
	}catch( std::exception &x ){
		return set_err( err, errlen, x.what() );
	}
}

extern "C" void rma_free( void *p ) { free( p ); }

// Entries entry[ 0 .. n ) (numbers in the whole database as rma_database_index() counts them,
// ascending) read, packed, and nothing else.  2: one of them needs the serial reader's diagnostics.
extern "C" int rma_pack_read_entries( const char *const *paths, int32_t n_paths, const char *fmt, int32_t maxslen, int32_t threads,
	const int32_t *entry, int32_t n, rma_pack_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	if( rma::seq_format_of( fmt ? fmt : "" ) != rma::FMT_FASTN )
		return 2;
	for( int i = 1; i < n; i++ )
		if( entry[ i ] <= entry[ i - 1 ] )
			return set_err( err, errlen, "rma_pack_read_entries: entry numbers must ascend" );
	std::unique_ptr<rma_pack>	pk( new rma_pack );
	const int	lim = maxslen > 0 ? maxslen + 1 : 30000000 + 1;
	try{
		int64_t	first_of_file = 0;
		int	at = 0;		// next of entry[] to serve
		for( int f = 0; f < n_paths && at < n; f++ ){
			const std::string	path = paths[ f ];
			if( rma::PackFile::is_pack( path ) ){
				rma::PackFile	one;
				std::string	e;
				if( !one.open( path, e ) )
					return set_err( err, errlen, e.c_str() );
				while( at < n && entry[ at ] < first_of_file + one.count() ){
					// a run of consecutive entries: one read
					const int	i0 = int( entry[ at ] - first_of_file );
					int	cnt = 1;
					while( at + cnt < n && entry[ at + cnt ] == entry[ at ] + cnt && i0 + cnt < one.count() )
						cnt++;
					if( !one.ensure_range( i0, cnt, e ) )
						return set_err( err, errlen, e.c_str() );
					for( int i = i0; i < i0 + cnt; i++ ){
						const int64_t	w1 = one.base_off[ i ] / 32, nw1 = ( int64_t( one.slen[ i ] ) + 31 ) / 32;
						const int64_t	x0 = one.exc_off[ i ], x1 = i + 1 < one.count() ? one.exc_off[ i + 1 ] : int64_t( one.exc.size() );
						pk->pf.append_packed( one.sid( i ), one.sdef( i ),
							std::vector<uint32_t>( one.codes.begin() + 2 * w1, one.codes.begin() + 2 * ( w1 + nw1 ) ),
							std::vector<uint32_t>( one.amask.begin() + w1, one.amask.begin() + w1 + nw1 ),
							std::vector<char>( one.exc.begin() + x0, one.exc.begin() + x1 ), one.slen[ i ] );
					}
					at += cnt;
				}
				first_of_file += one.count();
				continue;
			}
			rma::FastaStream	fs;
			if( !fs.open( path, lim, threads > 0 ? threads : 8 ) )
				return 2;
			std::vector<int32_t>	local;
			while( at < n && entry[ at ] < first_of_file + int64_t( fs.n_entries() ) )
				local.push_back( int32_t( entry[ at++ ] - first_of_file ) );
			if( !fs.read_entries( local.data(), local.size(), pk->pf ) )
				return 2;
			first_of_file += int64_t( fs.n_entries() );
		}
		if( at < n )
			return set_err( err, errlen, "rma_pack_read_entries: entry number beyond the database" );
	}catch( std::exception &x ){
		return set_err( err, errlen, x.what() );
	}
	*out = pk.release();
	return 0;
}

extern "C" void rma_pack_close( rma_pack_t *pk ) { delete pk; }
extern "C" int32_t rma_pack_count( const rma_pack_t *pk ) { return pk->pf.count(); }
extern "C" int64_t rma_pack_bases( const rma_pack_t *pk ) { return pk->pf.total_bases; }
extern "C" const char *rma_pack_sid( const rma_pack_t *pk, int32_t i )
{
	return i >= 0 && i < pk->pf.count() ? pk->pf.sid( i ) : nullptr;
}
extern "C" const char *rma_pack_sdef( const rma_pack_t *pk, int32_t i )
{
	return i >= 0 && i < pk->pf.count() ? pk->pf.sdef( i ) : nullptr;
}
extern "C" int32_t rma_pack_slen( const rma_pack_t *pk, int32_t i )
{
	return i >= 0 && i < pk->pf.count() ? pk->pf.slen[ i ] : -1;
}
extern "C" int rma_pack_seq( const rma_pack_t *pk, int32_t i, char *buf )
{
	if( i < 0 || i >= pk->pf.count() )
		return 1;
	std::string	s = pk->pf.unpack( i );
	memcpy( buf, s.c_str(), s.size() + 1 );
	return 0;
}

// for rm_scan_hip.hip
const rma::PackFile *rma_pack_file( const rma_pack_t *pk ) { return &pk->pf; }
const rma_pack_t *rma_pack_wrap( const rma::PackFile *pf )
{
	static_assert( offsetof( rma_pack, pf ) == 0, "rma_pack wraps exactly one PackFile" );
	return reinterpret_cast<const rma_pack_t *>( pf );
}

// ---------------------------------------------------------------- energy tables
extern "C" int rma_efndata_load( const char *dir, rma_efndata_t *out, char *err, size_t errlen )
{
	std::string	e;
	if( !rma::load_efndata( dir ? dir : "", out, e ) )
		return set_err( err, errlen, e.c_str() );
	return 0;
}

extern "C" int rma_efn2data_load( const char *dir, rma_efn2data_t *out, char *err, size_t errlen )
{
	std::string	e;
	try{
		if( !rma::load_efn2data( dir ? dir : "", out, e ) )
			return set_err( err, errlen, e.c_str() );
	}catch( rma::Error &x ){
		return set_err( err, errlen, x.what() );
	}
	return 0;
}

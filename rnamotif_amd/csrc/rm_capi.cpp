// rm_capi.cpp -- host half of the C ABI (include/rnamotif_amd.h): descriptor
// compilation and candidate replay.  The scanner/database half lives in
// rm_scan_hip.hip.
#include "rnamotif_amd.h"
#include "rm_cli.h"
#include <cstring>

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

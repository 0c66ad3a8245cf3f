// rm_cli.cpp -- see rm_cli.h.
#include "rm_cli.h"
#include "rm_score.h"
#include "rm_efndata.h"
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <exception>

namespace rma {

Prepared prepare( const Args &args )
{
	Prepared	pr;
	pr.descr = compile_descriptor( args );
	Descriptor	&d = *pr.descr;
	d.score->linkscore();
	pr.prog.reset( new rma_program_t );
	d.to_program( pr.prog.get() );
	const std::vector<EfnCall>	&ec = d.score->efn_calls();
	if( ec.size() > RMA_MAX_EFN_SITES )
		fail( "more than %d efn() calls in the score section.", RMA_MAX_EFN_SITES );
	pr.prog->n_efn_sites = int( ec.size() );
	for( size_t k = 0; k < ec.size(); k++ )
		pr.prog->efn_sites[ k ] = ec[ k ].site;
	bool	any_efn = false, any_efn2 = false;
	for( const EfnCall &c : ec )
		( c.site.kind == RMA_EFN_KIND_EFN2 ? any_efn2 : any_efn ) = true;
	if( any_efn && !args.copt ){
		pr.efn.reset( new rma_efndata_t );
		std::string	err, dir = find_efndata_dir( d );
		if( !load_efndata( dir, pr.efn.get(), err ) ){
			// score.c:1593: rm_efndataok = 0; the reference goes on and
			// scores with whatever was read.  Report and continue likewise.
			fputs( err.c_str(), stderr );
		}
	}
	if( any_efn2 && !args.copt ){
		pr.efn2.reset( new rma_efn2data_t );
		std::string	err, dir = find_efndata_dir( d );
		if( !load_efn2data( dir, pr.efn2.get(), err ) )	// score.c:1611, likewise not fatal there
			fputs( err.c_str(), stderr );
	}
	return pr;
}

int cli_main( int argc, char **argv, BackendFactory make_backend )
{
	const auto	t_start = std::chrono::steady_clock::now();
	auto lap = [&]( const char *what ){
		if( getenv( "RNAMOTIF_TIMING" ) )
			fprintf( stderr, "[timing] %-28s at %8.1f ms\n", what,
				std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t_start ).count() );
	};
	try{
		Args	args = parse_args( argc, argv );
		if( args.vopt )				// rnamot.c:56-65
			fprintf( stderr, "%s: %s.\n", argv[ 0 ], VERSION_STR );
		if( args.sopt ){
			std::unique_ptr<Descriptor>	d0 = init_only( args );
			dump_descriptor( *d0, stderr, 2, 0, 0, 0 );
		}
		if( args.vopt || args.sopt )
			return 0;
		if( !args.have_dfname && !args.have_xdfname ){
			fprintf( stderr, USAGE_FMT, argv[ 0 ] );
			return 1;
		}
		Prepared	pr = prepare( args );
		Descriptor	&d = *pr.descr;
		if( !d.stderr_text.empty() ){
			fputs( d.stderr_text.c_str(), stderr );
			d.stderr_text.clear();
		}
		if( args.have_dfname ){		// rnamot.c:89-97
			fprintf( stderr, "%s: complete descr length: min/max = %d/", args.dfname.c_str(), d.dminlen );
			if( d.dmaxlen == UNBOUNDED )
				fprintf( stderr, "UNBND\n" );
			else
				fprintf( stderr, "%d\n", d.dmaxlen );
		}
		if( args.dopt || args.hopt )		// rnamot.c:101-106
			dump_descriptor( d, stderr, args.dopt, args.dopt, args.dopt, args.hopt );
		if( args.dopt || args.popt )
			d.score->dump( stderr );
		if( args.copt )
			return 0;
		lap( "descriptor compiled" );
		ScanBackend	be = make_backend( pr.prog.get(), pr.efn.get(), pr.efn2.get() );
		lap( "scanner created" );
		const char	*bb = getenv( "RNAMOTIF_BATCH_BASES" );
		// (64 Mbase per batch: sixteen batches to a gigabase, each two or three milliseconds in every
		// stage of the pipeline -- reading, upload, kernels, replay -- so the stages overlap from the
		// first tenth of the database on; 268 Mbase, the value until round 2, gave the pipeline four)
		int64_t	batch_bases = bb ? atoll( bb ) : ( int64_t( 1 ) << 26 );
		run_search( d, *pr.prog, be, stdout, batch_bases, nullptr );
		lap( "search done" );
		return 0;
	}catch( Error &e ){
		const char	*m = e.what();
		fputs( m, stderr );
		size_t	n = strlen( m );
		if( n == 0 || m[ n - 1 ] != '\n' )
			fputc( '\n', stderr );
		return 1;
	}catch( std::exception &e ){		// (out of memory and the like: reported, not std::terminate)
		fprintf( stderr, "%s: %s\n", argc > 0 ? argv[ 0 ] : "rnamotif", e.what() );
		return 1;
	}
}

}	// namespace rma

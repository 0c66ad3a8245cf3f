// rm_pack_main.cpp -- rnamotif_pack: write a sequence database (fastn, pir or gb,
// read exactly as rnamotif reads it) as a packed database (rm_pack.h) that rnamotif
// accepts in place of the text file.
//   usage: rnamotif_pack [ -fmt fastn|pir|gb ] [ -N maxslen ] out.rmdb [ seq-file ... ]
#include "rm_pack.h"
#include <cstdlib>
#include <cstring>

int main( int argc, char **argv )
{
	std::string	fmt, out;
	std::vector<std::string>	files;
	int	maxslen = 30000000 + 1;
	for( int ac = 1; ac < argc; ac++ ){
		if( !strcmp( argv[ ac ], "-fmt" ) && ac + 1 < argc )
			fmt = argv[ ++ac ];
		else if( !strcmp( argv[ ac ], "-N" ) && ac + 1 < argc )
			maxslen = atoi( argv[ ++ac ] ) + 1;
		else if( argv[ ac ][ 0 ] == '-' || ( !fmt.empty() && fmt != "fastn" && fmt != "pir" && fmt != "gb" ) ){
			fprintf( stderr, "usage: %s [ -fmt fastn|pir|gb ] [ -N maxslen ] out.rmdb [ seq-file ... ]\n", argv[ 0 ] );
			return 1;
		}else if( out.empty() )
			out = argv[ ac ];
		else
			files.push_back( argv[ ac ] );
	}
	if( out.empty() ){
		fprintf( stderr, "usage: %s [ -fmt fastn|pir|gb ] [ -N maxslen ] out.rmdb [ seq-file ... ]\n", argv[ 0 ] );
		return 1;
	}
	rma::PackFile	pf;
	rma::SeqRecord	rec;
	const size_t	nf = files.empty() ? 1 : files.size();
	for( size_t f = 0; f < nf; f++ ){
		FILE	*fp = files.empty() ? stdin : fopen( files[ f ].c_str(), "r" );
		if( fp == nullptr ){
			fprintf( stderr, "DB_fnext: can't read seq file '%s'.\n", files[ f ].c_str() );
			return 1;
		}
		rma::FastaReader	rd( fp, maxslen, rma::seq_format_of( fmt ) );
		while( rd.next( rec ) )
			pf.add( rec );
		if( fp != stdin )
			fclose( fp );
	}
	std::string	err;
	if( !pf.save( out, err ) ){
		fprintf( stderr, "%s\n", err.c_str() );
		return 1;
	}
	fprintf( stderr, "%s: %d entries, %lld bases, %zu ambiguous.\n", out.c_str(), pf.count(),
		( long long )pf.total_bases, pf.exc.size() );
	return 0;
}

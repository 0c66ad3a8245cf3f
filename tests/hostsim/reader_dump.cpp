// reader_dump.cpp -- TEST INFRASTRUCTURE.
//
// The product's database readers (rnamotif_amd/csrc/rm_fasta.cpp, and rm_stream.cpp for FASTA
// files) printing every entry the way oracle/ref_dbutil_drv.c prints what the reference's
// dbutil.c reads: name, definition, length, letters; <EOF> where the next file is opened.
//
//   reader_dump fastn|pir|gb maxslen serial|stream [file ...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "rm_fasta.h"
#include "rm_stream.h"

static void put( const char *sid, const char *sdef, const std::string &seq )
{
	printf( "%s\n%s\n%d\n%s\n", sid, sdef, int( seq.size() ), seq.c_str() );
}

int main( int argc, char **argv )
{
	if( argc < 4 ){
		fprintf( stderr, "usage: reader_dump fastn|pir|gb maxslen serial|stream [file ...]\n" );
		return 2;
	}
	const rma::SeqFormat	fmt = rma::seq_format_of( argv[ 1 ] );
	const int	lim = atoi( argv[ 2 ] ) + 1;
	const bool	stream = !strcmp( argv[ 3 ], "stream" );
	const int	nf = argc - 4;
	// the EOF of file f, as DB_fnext() and the main loop handle it (dbutil.c:12-40, rnamot.c:160-168)
	bool	stop = false;
	auto next_file = [&]( int f ){
		if( f + 1 >= nf )
			return;
		FILE	*t = fopen( argv[ 4 + f + 1 ], "r" );
		if( t == nullptr ){
			fprintf( stderr, "DB_fnext: can't read seq file '%s'.\n", argv[ 4 + f + 1 ] );
			stop = true;
			return;
		}
		fclose( t );
		printf( "<EOF>\n" );
	};
	for( int f = 0; f < std::max( nf, 1 ) && !stop; f++ ){
		FILE	*fp = stdin;
		long	resume = 0;
		if( nf > 0 && stream && fmt == rma::FMT_FASTN ){
			rma::FastaStream	fs;
			if( fs.open( argv[ 4 + f ], lim, 3 ) ){
				while( std::unique_ptr<rma::PackFile> pk = fs.next( 1000 ) )
					for( int i = 0; i < pk->count(); i++ )
						put( pk->sid( i ), pk->sdef( i ), pk->unpack( i ) );
				if( fs.stopped_at() < 0 ){
					next_file( f );
					continue;
				}
				resume = long( fs.stopped_at() );
			}
		}
		if( nf > 0 ){
			fp = fopen( argv[ 4 + f ], "r" );
			if( fp == nullptr ){
				fprintf( stderr, "DB_fnext: can't read seq file '%s'.\n", argv[ 4 + f ] );
				return f == 0 ? 1 : 0;
			}
			fseek( fp, resume, SEEK_SET );
		}
		rma::FastaReader	rd( fp, lim, fmt );
		rma::SeqRecord	rec;
		while( rd.next( rec ) )
			put( rec.sid.c_str(), rec.sdef.c_str(), rec.seq );
		if( fp != stdin )
			fclose( fp );
		next_file( f );
	}
	return 0;
}

/* ref_dbutil_drv.c -- TEST INFRASTRUCTURE.
 *
 * A driver of this repository's own around the reference's database readers, compiled together
 * with /root/reference/src/dbutil.c where it lies (oracle/Makefile, target ref; never copied,
 * never committed as a binary): DB_fnext() over the files given and FN_/PIR_/GB_fgetseq() for
 * every entry, called the way main() calls them (rnamot.c:125-179: SDEF_SIZE, a buffer of
 * a_maxslen bytes), each entry printed as four lines -- name, definition, length, letters.  The
 * product's readers (rnamotif_amd/csrc/rm_fasta.cpp, rm_stream.cpp) are pinned against it
 * (tests/test_reader_pins.py).
 *
 *   ref_dbutil_drv fastn|pir|gb maxslen [file ...]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dbutil.h"

#define UNDEF		(-1)		/* rmdefs.h */
#define SID_SIZE	100		/* rnamot.h:44 */
#define SDEF_SIZE	20000		/* rnamot.h:45 */

int main( int argc, char *argv[] )
{
	int	( *fgetseq )( FILE *, char *, int, char *, int, char * ) = FN_fgetseq;
	static char	sid[ SID_SIZE ], sdef[ SDEF_SIZE ];
	char	*sbuf;
	int	s_sbuf, slen, c_fname = UNDEF, n_fname;
	FILE	*fp = NULL;

	if( argc < 3 ){
		fprintf( stderr, "usage: %s fastn|pir|gb maxslen [file ...]\n", argv[ 0 ] );
		return 2;
	}
	if( !strcmp( argv[ 1 ], "pir" ) )
		fgetseq = PIR_fgetseq;
	else if( !strcmp( argv[ 1 ], "gb" ) )
		fgetseq = GB_fgetseq;
	s_sbuf = atoi( argv[ 2 ] ) + 1;		/* getargs.c: a_maxslen = N + 1 */
	sbuf = ( char * )malloc( ( size_t )s_sbuf );
	n_fname = argc - 3;
	fp = DB_fnext( fp, &c_fname, n_fname, &argv[ 3 ] );
	if( fp == NULL )
		return 1;
	for( ; ; ){
		slen = fgetseq( fp, sid, SDEF_SIZE, sdef, s_sbuf, sbuf );
		if( slen == EOF ){
			fp = DB_fnext( fp, &c_fname, n_fname, &argv[ 3 ] );
			if( fp == NULL )
				break;
			printf( "<EOF>\n" );	/* (the main loop searches an empty entry here, rnamot.c:160-179) */
			continue;
		}
		printf( "%s\n%s\n%d\n%.*s\n", sid, sdef, slen, slen, sbuf );
	}
	return 0;
}

// rm_fasta.cpp -- see rm_fasta.h.
#include "rm_fasta.h"
#include <cctype>

namespace rma {

static int skipbl2nl( FILE *fp )	// dbutil.c:336-345
{
	int	c;
	while( isspace( c = getc( fp ) ) )
		if( c == '\n' )
			break;
	return c;
}

bool FastaReader::next( SeqRecord &rec )	// FN_fgetseq, dbutil.c:42-128
{
	rec.sid.clear();
	rec.sdef.clear();
	rec.seq.clear();
	rec.eof = false;
	int	c = getc( fp_ );
	if( c == EOF ){
		rec.eof = true;
		return false;
	}
	if( c != '>' ){
		fprintf( stderr, "FN_fgetseq: fastn file does not begin with '>'.\n" );
		rec.eof = true;
		return false;
	}
	if( ( c = skipbl2nl( fp_ ) ) == EOF || c == '\n' ){
		fprintf( stderr, "FN_fgetseq: fastn file has an unnamed entry.\n" );
		rec.eof = true;
		return false;
	}
	rec.sid.push_back( char( c ) );
	while( ( c = getc( fp_ ) ) != EOF ){
		if( isspace( c ) )
			break;
		if( rec.sid.size() < 99 )	// SID_SIZE; the reference does not check
			rec.sid.push_back( char( c ) );
	}
	if( c == EOF )
		return true;
	if( c != '\n' ){
		if( ( c = skipbl2nl( fp_ ) ) == EOF )
			return true;
	}
	if( c != '\n' ){
		const unsigned	s_sdef = 20000;		// SDEF_SIZE
		unsigned	cnt = 1;
		rec.sdef.push_back( char( c ) );
		while( ( c = getc( fp_ ) ) != 0 ){
			if( c == '\n' || c == EOF )
				break;
			cnt++;
			if( cnt < s_sdef )
				rec.sdef.push_back( char( c ) );
		}
		if( cnt >= s_sdef )
			fprintf( stderr, "FN_fgetseq: entry: '%s': def len: %d, truncated to %d.\n",
				rec.sid.c_str(), cnt, s_sdef - 1 );
	}
	if( c == EOF )
		return true;
	unsigned	cnt = 0;
	while( ( c = getc( fp_ ) ) != EOF ){
		if( c == '>' ){
			ungetc( c, fp_ );
			break;
		}
		if( isalpha( c ) ){
			cnt++;
			if( cnt < unsigned( maxslen_ ) ){
				c = tolower( c );
				rec.seq.push_back( c == 'u' ? 't' : char( c ) );
			}
		}
	}
	if( cnt > unsigned( maxslen_ ) )
		fprintf( stderr, "FN_fgetseq: entry: '%s': seq len: %d, truncated to %d.\n",
			rec.sid.c_str(), cnt, maxslen_ - 1 );
	return true;
}

void PackedDb::add( const char *seq, int n )
{
	base_off.push_back( padded_bases() );
	slen.push_back( n );
	total_bases += n;
	size_t	w2 = codes.size(), w1 = amask.size();
	size_t	nw1 = ( size_t( n ) + 31 ) / 32;
	codes.resize( w2 + nw1 * 2, 0 );
	amask.resize( w1 + nw1, 0 );
	for( int i = 0; i < n; i++ ){
		unsigned	code;
		switch( seq[ i ] ){
		case 'a' : case 'A' : code = 0; break;
		case 'c' : case 'C' : code = 1; break;
		case 'g' : case 'G' : code = 2; break;
		case 't' : case 'T' : case 'u' : case 'U' : code = 3; break;
		default :
			code = 0;
			amask[ w1 + ( i >> 5 ) ] |= 1u << ( i & 31 );
			break;
		}
		codes[ w2 + ( i >> 4 ) ] |= code << ( 2 * ( i & 15 ) );
	}
}

}	// namespace rma

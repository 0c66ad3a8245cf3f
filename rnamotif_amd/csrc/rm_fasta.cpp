// rm_fasta.cpp -- see rm_fasta.h.
#include "rm_fasta.h"
#include <cctype>
#include <cstring>

namespace rma {

static int skipbl2nl( FILE *fp )	// dbutil.c:336-345
{
	int	c;
	while( isspace( c = getc( fp ) ) )
		if( c == '\n' )
			break;
	return c;
}

SeqFormat seq_format_of( const std::string &name )
{
	if( name == "pir" )
		return FMT_PIR;
	if( name == "gb" )
		return FMT_GENBANK;
	return FMT_FASTN;
}

bool FastaReader::next( SeqRecord &rec )
{
	switch( fmt_ ){
	case FMT_PIR : return next_pir( rec );
	case FMT_GENBANK : return next_gb( rec );
	default : return next_fastn( rec );
	}
}

// Sequence letters up to the next '>' (FN_/PIR_fgetseq share this loop, dbutil.c:104-126,197-223)
static void read_letters( FILE *fp, const char *who, int maxslen, SeqRecord &rec )
{
	unsigned	cnt = 0;
	int	c;
	while( ( c = getc( fp ) ) != EOF ){
		if( c == '>' ){
			ungetc( c, fp );
			break;
		}
		if( isalpha( c ) ){
			cnt++;
			if( cnt < unsigned( maxslen ) ){
				c = tolower( c );
				rec.seq.push_back( c == 'u' ? 't' : char( c ) );
			}
		}
	}
	if( cnt > unsigned( maxslen ) )
		fprintf( stderr, "%s: entry: '%s': seq len: %d, truncated to %d.\n",
			who, rec.sid.c_str(), cnt, maxslen - 1 );
}

bool FastaReader::next_pir( SeqRecord &rec )	// PIR_fgetseq, dbutil.c:130-224
{
	rec.sid.clear();
	rec.sdef.clear();
	rec.seq.clear();
	rec.eof = true;
	int	c = getc( fp_ );
	if( c == EOF )
		return false;
	if( c != '>' ){
		fprintf( stderr, "PIR_fgetseq: pir file does not begin with '>'.\n" );
		return false;
	}
	if( ( c = skipbl2nl( fp_ ) ) == EOF || c == '\n' ){
		fprintf( stderr, "PIR_fgetseq: pir file has an unnamed entry.\n" );
		return false;
	}
	rec.sid.push_back( char( c ) );
	while( ( c = getc( fp_ ) ) != EOF ){
		if( isspace( c ) )
			break;
		if( rec.sid.size() < 99 )
			rec.sid.push_back( char( c ) );
	}
	if( c == EOF ){
		fprintf( stderr, "PIR_fgetseq: entry: '%s': no title line.\n", rec.sid.c_str() );
		return false;
	}
	if( c != '\n' ){
		fprintf( stderr, "PIR_fgetseq: entry: '%s': extra chars on ID line ignored.\n", rec.sid.c_str() );
		while( ( c = getc( fp_ ) ) != EOF )
			if( c == '\n' )
				break;
	}
	if( c != '\n' ){
		fprintf( stderr, "PIR_fgetseq: entry: '%s': no title line.\n", rec.sid.c_str() );
		return false;
	}
	// the title line: its first character is taken whatever it is (dbutil.c:181-183)
	const unsigned	s_sdef = 20000;
	unsigned	cnt = 1;
	c = getc( fp_ );
	if( c != EOF )		// (a newline too: an empty title line swallows the line after it)
		rec.sdef.push_back( char( c ) );
	while( ( c = getc( fp_ ) ) != 0 ){
		if( c == '\n' || c == EOF )
			break;
		cnt++;
		if( cnt < s_sdef )
			rec.sdef.push_back( char( c ) );
	}
	if( cnt >= s_sdef )
		fprintf( stderr, "PIR_fgetseq: entry: '%s': title len: %d, truncated to %d.\n",
			rec.sid.c_str(), cnt, s_sdef - 1 );
	rec.eof = false;
	read_letters( fp_, "PIR_fgetseq", maxslen_, rec );
	return true;
}

bool FastaReader::next_gb( SeqRecord &rec )	// GB_fgetseq, dbutil.c:226-334
{
	rec.sid.clear();
	rec.sdef.clear();
	rec.seq.clear();
	rec.eof = true;
	char	line[ 256 ], locus[ 256 ], acc[ 256 ], gid[ 256 ];
	const unsigned	s_sdef = 20000;
	*locus = '\0';
	while( fgets( line, sizeof( line ), fp_ ) ){
		if( !strncmp( line, "LOCUS", 5 ) ){
			sscanf( line, "LOCUS %255s", locus );
			break;
		}
	}
	if( *locus == '\0' )
		return false;
	unsigned	cnt = 0;
	// everything between LOCUS and ACCESSION is appended to the definition,
	// newlines as blanks; the blank of the last line is cut (dbutil.c:250-275)
	while( fgets( line, sizeof( line ), fp_ ) ){
		if( !strncmp( line, "ACCESSION", 9 ) ){
			if( !rec.sdef.empty() )
				rec.sdef.pop_back();
			break;
		}
		for( const char *lp = line; *lp; lp++ ){
			cnt++;
			if( cnt < s_sdef )
				rec.sdef.push_back( *lp == '\n' ? ' ' : *lp );
		}
	}
	if( rec.sdef.empty() ){
		fprintf( stderr, "GB_fgetseq: missing DEFINITION line.\n" );
		return false;
	}
	if( cnt >= s_sdef )
		fprintf( stderr, "GB_fgetseq: entry: '%s': def len: %d, truncated to %d.\n", "", cnt, s_sdef - 1 );
	*acc = *gid = '\0';
	while( fgets( line, sizeof( line ), fp_ ) ){
		if( !strncmp( line, "VERSION", 7 ) ){
			sscanf( line, "VERSION %255s GI:%255s", acc, gid );
			break;
		}
	}
	if( *acc == '\0' ){
		fprintf( stderr, "GB_fgetseq: missing VERSION line.\n" );
		return false;
	}
	if( char *dp = strchr( acc, '.' ) )
		*dp = '\0';
	rec.sid = std::string( "gi|" ) + gid + "|gb|" + acc + "|" + locus;
	bool	origin = false;
	while( fgets( line, sizeof( line ), fp_ ) ){
		if( !strncmp( line, "ORIGIN", 6 ) ){
			origin = true;
			break;
		}
	}
	if( !origin ){
		fprintf( stderr, "GB_fgetseq: missing ORIGIN line.\n" );
		return false;
	}
	cnt = 0;
	bool	slashes = false;
	while( fgets( line, sizeof( line ), fp_ ) ){
		if( !strncmp( line, "//", 2 ) ){
			slashes = true;
			break;
		}
		for( const char *lp = line; *lp; lp++ ){
			if( isalpha( ( unsigned char )*lp ) ){
				cnt++;
				if( cnt < unsigned( maxslen_ ) )	// no u -> t here (dbutil.c:312-313)
					rec.seq.push_back( char( tolower( ( unsigned char )*lp ) ) );
			}
		}
	}
	if( !slashes ){
		fprintf( stderr, "GB_fgetseq: missing // line.\n" );
		return false;
	}
	if( cnt > unsigned( maxslen_ ) )
		fprintf( stderr, "GB_fgetseq: entry: '%s': seq len: %d, truncated to %d.\n",
			rec.sid.c_str(), cnt, maxslen_ - 1 );
	rec.eof = false;
	return true;
}

bool FastaReader::next_fastn( SeqRecord &rec )	// FN_fgetseq, dbutil.c:42-128
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

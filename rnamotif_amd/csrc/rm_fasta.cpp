// rm_fasta.cpp -- see rm_fasta.h.
#include "rm_fasta.h"
#include <cctype>
#include <cstring>

namespace rma {

// skipbl2nl(), dbutil.c:336-345, over the reader's own get()
#define SKIPBL2NL( c_ )	do{ while( isspace( ( c_ ) = get() ) ) if( ( c_ ) == '\n' ) break; }while( 0 )

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

// Sequence letters up to the next '>' (FN_/PIR_fgetseq share this loop, dbutil.c:104-126,197-223):
// every alphabetic character, lower case, u -> t; straight over the block buffer
void FastaReader::read_letters( const char *who, SeqRecord &rec )
{
	static unsigned char	tab[ 256 ];
	static bool	init = false;
	if( !init ){
		for( int c = 0; c < 256; c++ ){
			int	l = isalpha( c ) ? tolower( c ) : 0;
			tab[ c ] = ( unsigned char )( l == 'u' ? 't' : l );
		}
		init = true;
	}
	unsigned	cnt = 0;
	const unsigned	lim = unsigned( maxslen_ );
	for( bool more = true; more; ){
		if( pos_ == len_ ){
			len_ = fread( buf_.data(), 1, buf_.size(), fp_ );
			pos_ = 0;
			if( len_ == 0 )
				break;
		}
		const char	*p = buf_.data() + pos_, *e = buf_.data() + len_;
		const size_t	o0 = rec.seq.size();
		rec.seq.resize( o0 + size_t( e - p ) );
		char	*o = &rec.seq[ o0 ];
		for( ; p < e; p++ ){
			const unsigned char	t = tab[ ( unsigned char )*p ];
			if( t ){
				cnt++;
				if( cnt < lim )
					*o++ = char( t );
			}else if( *p == '>' ){
				more = false;
				break;
			}
		}
		rec.seq.resize( size_t( o - rec.seq.data() ) );
		pos_ = size_t( p - buf_.data() );	// at the '>' (left unread) or at the end of the block
	}
	if( cnt > lim )
		fprintf( stderr, "%s: entry: '%s': seq len: %d, truncated to %d.\n",
			who, rec.sid.c_str(), cnt, maxslen_ - 1 );
}

bool FastaReader::next_pir( SeqRecord &rec )	// PIR_fgetseq, dbutil.c:130-224
{
	rec.sid.clear();
	rec.sdef.clear();
	rec.seq.clear();
	rec.eof = true;
	int	c = get();
	if( c == EOF )
		return false;
	if( c != '>' ){
		fprintf( stderr, "PIR_fgetseq: pir file does not begin with '>'.\n" );
		return false;
	}
	SKIPBL2NL( c );
	if( c == EOF || c == '\n' ){
		fprintf( stderr, "PIR_fgetseq: pir file has an unnamed entry.\n" );
		return false;
	}
	rec.sid.push_back( char( c ) );
	while( ( c = get() ) != EOF ){
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
		while( ( c = get() ) != EOF )
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
	c = get();
	// (a newline too: an empty title line swallows the line after it; and the EOF of a file that
	// ends with the name line, stored as the character it converts to -- pinned against the
	// reference's reader, tests/test_reader_pins.py)
	rec.sdef.push_back( char( c ) );
	while( ( c = get() ) != 0 ){
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
	read_letters( "PIR_fgetseq", rec );
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
	int	c = get();
	if( c == EOF ){
		rec.eof = true;
		return false;
	}
	if( c != '>' ){
		fprintf( stderr, "FN_fgetseq: fastn file does not begin with '>'.\n" );
		rec.eof = true;
		return false;
	}
	SKIPBL2NL( c );
	if( c == EOF || c == '\n' ){
		fprintf( stderr, "FN_fgetseq: fastn file has an unnamed entry.\n" );
		rec.eof = true;
		return false;
	}
	rec.sid.push_back( char( c ) );
	while( ( c = get() ) != EOF ){
		if( isspace( c ) )
			break;
		if( rec.sid.size() < 99 )	// SID_SIZE; the reference does not check
			rec.sid.push_back( char( c ) );
	}
	if( c == EOF )
		return true;
	if( c != '\n' ){
		SKIPBL2NL( c );
		if( c == EOF )
			return true;
	}
	if( c != '\n' ){
		const unsigned	s_sdef = 20000;		// SDEF_SIZE
		unsigned	cnt = 1;
		rec.sdef.push_back( char( c ) );
		while( ( c = get() ) != 0 ){
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
	read_letters( "FN_fgetseq", rec );
	return true;
}

void PackedDb::add( const char *seq, int n )
{
	// letter -> 2-bit code (bits 0-1) and ambiguity flag (bit 2)
	static unsigned char	lut[ 256 ];
	static bool	init = false;
	if( !init ){
		for( int c = 0; c < 256; c++ )
			lut[ c ] = 4;
		lut[ 'a' ] = lut[ 'A' ] = 0;
		lut[ 'c' ] = lut[ 'C' ] = 1;
		lut[ 'g' ] = lut[ 'G' ] = 2;
		lut[ 't' ] = lut[ 'T' ] = lut[ 'u' ] = lut[ 'U' ] = 3;
		init = true;
	}
	base_off.push_back( padded_bases() );
	slen.push_back( n );
	total_bases += n;
	const size_t	w2 = codes.size(), w1 = amask.size();
	const size_t	nw1 = ( size_t( n ) + 31 ) / 32;
	codes.resize( w2 + nw1 * 2, 0 );
	amask.resize( w1 + nw1, 0 );
	const unsigned char	*p = reinterpret_cast<const unsigned char *>( seq );
	uint32_t	*cw = codes.data() + w2, *mw = amask.data() + w1;
	int	i = 0;
	for( ; i + 32 <= n; i += 32, p += 32 ){	// one mask word, two code words per step
		uint32_t	c0 = 0, c1 = 0, m = 0;
		for( int k = 0; k < 16; k++ ){
			const unsigned	a = lut[ p[ k ] ], b = lut[ p[ 16 + k ] ];
			c0 |= ( a & 3u ) << ( 2 * k );
			c1 |= ( b & 3u ) << ( 2 * k );
			m |= ( ( a >> 2 ) << k ) | ( ( b >> 2 ) << ( 16 + k ) );
		}
		cw[ i >> 4 ] = c0;
		cw[ ( i >> 4 ) + 1 ] = c1;
		mw[ i >> 5 ] = m;
	}
	for( ; i < n; i++, p++ ){
		const unsigned	a = lut[ *p ];
		cw[ i >> 4 ] |= ( a & 3u ) << ( 2 * ( i & 15 ) );
		mw[ i >> 5 ] |= ( a >> 2 ) << ( i & 31 );
	}
}

}	// namespace rma

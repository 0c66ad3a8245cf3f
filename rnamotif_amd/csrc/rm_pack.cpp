// rm_pack.cpp -- see rm_pack.h.
#include "rm_pack.h"
#include <cstdio>
#include <cstring>

namespace rma {

static const char	PACK_MAGIC[ 9 ] = "RMAPACK1";

void PackFile::add( const SeqRecord &rec )
{
	const int	n = int( rec.seq.size() );
	const size_t	w2 = codes.size(), w1 = amask.size();
	const size_t	nw1 = ( size_t( n ) + 31 ) / 32;
	base_off.push_back( int64_t( w1 ) * 32 );
	exc_off.push_back( int64_t( exc.size() ) );
	slen.push_back( n );
	total_bases += n;
	codes.resize( w2 + nw1 * 2, 0 );
	amask.resize( w1 + nw1, 0 );
	for( int i = 0; i < n; i++ ){
		unsigned	code = 0;
		switch( rec.seq[ i ] ){
		case 'a' : code = 0; break;
		case 'c' : code = 1; break;
		case 'g' : code = 2; break;
		case 't' : case 'u' : code = 3; break;	// (only the GenBank reader can deliver a 'u', dbutil.c:312)
		default :
			amask[ w1 + ( i >> 5 ) ] |= 1u << ( i & 31 );
			exc.push_back( rec.seq[ i ] );
			break;
		}
		codes[ w2 + ( i >> 4 ) ] |= code << ( 2 * ( i & 15 ) );
	}
	sid_off.push_back( int64_t( text.size() ) );
	text.insert( text.end(), rec.sid.begin(), rec.sid.end() );
	text.push_back( '\0' );
	sdef_off.push_back( int64_t( text.size() ) );
	text.insert( text.end(), rec.sdef.begin(), rec.sdef.end() );
	text.push_back( '\0' );
}

void PackFile::append_packed( const std::string &sid, const std::string &sdef, const std::vector<uint32_t> &c,
	const std::vector<uint32_t> &m, const std::vector<char> &x, int32_t n )
{
	base_off.push_back( int64_t( amask.size() ) * 32 );
	exc_off.push_back( int64_t( exc.size() ) );
	slen.push_back( n );
	total_bases += n;
	codes.insert( codes.end(), c.begin(), c.end() );
	amask.insert( amask.end(), m.begin(), m.end() );
	exc.insert( exc.end(), x.begin(), x.end() );
	sid_off.push_back( int64_t( text.size() ) );
	text.insert( text.end(), sid.c_str(), sid.c_str() + strlen( sid.c_str() ) + 1 );
	sdef_off.push_back( int64_t( text.size() ) );
	text.insert( text.end(), sdef.c_str(), sdef.c_str() + strlen( sdef.c_str() ) + 1 );
}

void PackFile::window( int i, int comp, int lo, int hi, char *out ) const
{
	const int	n = slen[ i ];
	lo = lo < 0 ? 0 : lo;
	hi = hi > n ? n : hi;
	if( lo >= hi )
		return;
	const uint32_t	*cw = codes.data() + base_off[ i ] / 16;
	const uint32_t	*mw = amask.data() + base_off[ i ] / 32;
	if( comp ){
		for( int p = lo; p < hi; p++ ){
			const int	f = n - 1 - p;
			out[ p ] = ( ( mw[ f >> 5 ] >> ( f & 31 ) ) & 1 ) ? 'n' : "tgca"[ ( cw[ f >> 4 ] >> ( 2 * ( f & 15 ) ) ) & 3 ];
		}
		return;
	}
	// the letters at masked positions come from exc[], in order: find the first one's index only
	// if the window holds any
	const char	*ex = nullptr;
	for( int p = lo; p < hi; p++ ){
		if( ( mw[ p >> 5 ] >> ( p & 31 ) ) & 1 ){
			if( ex == nullptr ){
				int64_t	before = 0;
				for( int w = 0; w < ( p >> 5 ); w++ )
					before += __builtin_popcount( mw[ w ] );
				before += __builtin_popcount( mw[ p >> 5 ] & ( ( 1u << ( p & 31 ) ) - 1 ) );
				ex = exc.data() + exc_off[ i ] + before;
			}
			out[ p ] = ex < exc.data() + exc.size() ? *ex++ : 'n';
		}else
			out[ p ] = "acgt"[ ( cw[ p >> 4 ] >> ( 2 * ( p & 15 ) ) ) & 3 ];
	}
}

std::string PackFile::unpack( int i ) const
{
	const int	n = slen[ i ];
	std::string	s( size_t( n ), 'a' );
	const uint32_t	*cw = codes.data() + base_off[ i ] / 16;
	const uint32_t	*mw = amask.data() + base_off[ i ] / 32;
	const char	*ex = exc.data() + exc_off[ i ], *ex_end = exc.data() + exc.size();
	for( int p = 0; p < n; p++ ){
		if( ( mw[ p >> 5 ] >> ( p & 31 ) ) & 1 )
			s[ p ] = ex < ex_end ? *ex++ : 'n';
		else
			s[ p ] = "acgt"[ ( cw[ p >> 4 ] >> ( 2 * ( p & 15 ) ) ) & 3 ];
	}
	return s;
}

void PackFile::index_text()
{
	sid_off.clear();
	sdef_off.clear();
	size_t	p = 0;
	for( int i = 0; i < count(); i++ ){
		sid_off.push_back( int64_t( p ) );
		p += strlen( text.data() + p ) + 1;
		sdef_off.push_back( int64_t( p ) );
		p += strlen( text.data() + p ) + 1;
	}
}

template< class T >
static bool put( FILE *fp, const std::vector<T> &v )
{
	return v.empty() || fwrite( v.data(), sizeof( T ), v.size(), fp ) == v.size();
}
template< class T >
static bool get( FILE *fp, std::vector<T> &v, int64_t n )
{
	if( n < 0 || n > ( int64_t( 1 ) << 40 ) )
		return false;
	v.resize( size_t( n ) );
	return n == 0 || fread( v.data(), sizeof( T ), size_t( n ), fp ) == size_t( n );
}

bool PackFile::save( const std::string &path, std::string &err ) const
{
	FILE	*fp = fopen( path.c_str(), "wb" );
	if( fp == nullptr ){
		err = "can't write packed database '" + path + "'.";
		return false;
	}
	int64_t	hdr[ 5 ] = { int64_t( slen.size() ), int64_t( codes.size() ), int64_t( amask.size() ),
		int64_t( exc.size() ), int64_t( text.size() ) };
	bool	ok = fwrite( PACK_MAGIC, 1, 8, fp ) == 8 && fwrite( hdr, sizeof( hdr ), 1, fp ) == 1 &&
		put( fp, slen ) && put( fp, base_off ) && put( fp, exc_off ) && put( fp, codes ) && put( fp, amask ) &&
		put( fp, exc ) && put( fp, text );
	ok = fclose( fp ) == 0 && ok;
	if( !ok )
		err = "write error on packed database '" + path + "'.";
	return ok;
}

bool PackFile::load( const std::string &path, std::string &err )
{
	FILE	*fp = fopen( path.c_str(), "rb" );
	if( fp == nullptr ){
		err = "can't read packed database '" + path + "'.";
		return false;
	}
	char	magic[ 8 ];
	int64_t	hdr[ 5 ];
	bool	ok = fread( magic, 1, 8, fp ) == 8 && !memcmp( magic, PACK_MAGIC, 8 ) && fread( hdr, sizeof( hdr ), 1, fp ) == 1;
	ok = ok && get( fp, slen, hdr[ 0 ] ) && get( fp, base_off, hdr[ 0 ] ) && get( fp, exc_off, hdr[ 0 ] ) &&
		get( fp, codes, hdr[ 1 ] ) && get( fp, amask, hdr[ 2 ] ) && get( fp, exc, hdr[ 3 ] ) && get( fp, text, hdr[ 4 ] );
	fclose( fp );
	if( ok ){
		// consistency: every entry inside the arrays, text terminated
		total_bases = 0;
		for( size_t i = 0; ok && i < slen.size(); i++ ){
			const int64_t	nw1 = ( int64_t( slen[ i ] ) + 31 ) / 32;
			ok = slen[ i ] >= 0 && base_off[ i ] >= 0 && base_off[ i ] % 32 == 0 &&
				base_off[ i ] / 32 + nw1 <= int64_t( amask.size() ) &&
				base_off[ i ] / 16 + 2 * nw1 <= int64_t( codes.size() ) &&
				exc_off[ i ] >= 0 && exc_off[ i ] <= int64_t( exc.size() );
			// entries follow each other without overlap (a slice of the arrays is uploaded as it is)
			ok = ok && ( i == 0 || ( base_off[ i ] >= base_off[ i - 1 ] + ( ( int64_t( slen[ i - 1 ] ) + 31 ) / 32 ) * 32 &&
				exc_off[ i ] >= exc_off[ i - 1 ] ) );
			total_bases += slen[ i ];
		}
		size_t	nul = 0;
		for( char c : text )
			nul += c == '\0';
		ok = ok && nul >= 2 * slen.size() && ( text.empty() || text.back() == '\0' );
	}
	if( !ok ){
		err = "'" + path + "' is not a packed database of this build.";
		return false;
	}
	index_text();
	return true;
}

bool PackFile::is_pack( const std::string &path )
{
	FILE	*fp = fopen( path.c_str(), "rb" );
	if( fp == nullptr )
		return false;
	char	magic[ 8 ];
	bool	ok = fread( magic, 1, 8, fp ) == 8 && !memcmp( magic, PACK_MAGIC, 8 );
	fclose( fp );
	return ok;
}

}	// namespace rma

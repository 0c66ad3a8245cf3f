// rm_pack.cpp -- see rm_pack.h.
#include "rm_pack.h"
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <thread>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <new>

namespace rma {

void *pack_map_pages( size_t len )
{
	const size_t	huge = size_t( 2 ) << 20;
	char	*m = static_cast<char *>( mmap( nullptr, len + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0 ) );
	if( m == MAP_FAILED )
		throw std::bad_alloc();
	char	*a = reinterpret_cast<char *>( ( reinterpret_cast<uintptr_t>( m ) + huge - 1 ) & ~uintptr_t( huge - 1 ) );
	if( a > m )
		munmap( m, size_t( a - m ) );
	if( a + len < m + len + huge )
		munmap( a + len, size_t( ( m + len + huge ) - ( a + len ) ) );
	( void )madvise( a, len, MADV_HUGEPAGE );
	return a;
}

void pack_unmap_pages( void *p, size_t len )
{
	munmap( p, len );
}

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

template< class V >
static bool put( FILE *fp, const V &v )
{
	return v.empty() || fwrite( v.data(), sizeof( v[ 0 ] ), v.size(), fp ) == v.size();
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

struct PackFile::Source {
	int	fd = -1;
	int64_t	codes_at = 0, amask_at = 0;	// file offsets of the two arrays
	std::string	path;
	~Source(){ if( fd >= 0 ) close( fd ); }
};

static bool pread_all( int fd, void *buf, size_t n, int64_t at )
{
	char	*p = static_cast<char *>( buf );
	while( n > 0 ){
		const ssize_t	k = pread( fd, p, n, at );
		if( k <= 0 )
			return false;
		p += k;
		at += k;
		n -= size_t( k );
	}
	return true;
}

// (a fresh vector is touched for the first time as it is filled: page faults and copy go at 2 GB/s
// in one thread, so a read of more than a few MB is shared)
static bool pread_parallel( int fd, void *buf, size_t n, int64_t at )
{
	const size_t	piece = size_t( 1 ) << 20;
	unsigned	nt = unsigned( std::min<size_t>( n / piece, 16 ) );
	nt = std::min( nt, std::max( 1u, std::thread::hardware_concurrency() ) );
	if( nt < 2 )
		return pread_all( fd, buf, n, at );
	std::vector<std::thread>	pool;
	std::vector<char>	ok( nt, 0 );
	for( unsigned t = 0; t < nt; t++ ){
		const size_t	lo = ( n / nt ) * t, hi = t + 1 == nt ? n : ( n / nt ) * ( t + 1 );
		pool.emplace_back( [ =, &ok ](){ ok[ t ] = pread_all( fd, static_cast<char *>( buf ) + lo, hi - lo, at + int64_t( lo ) ); } );
	}
	bool	all = true;
	for( unsigned t = 0; t < nt; t++ ){
		pool[ t ].join();
		all = all && ok[ t ];
	}
	return all;
}

bool PackFile::open( const std::string &path, std::string &err )
{
	const std::string	bad = "'" + path + "' is not a packed database of this build.";
	std::shared_ptr<Source>	src( new Source );
	src->path = path;
	src->fd = ::open( path.c_str(), O_RDONLY );
	if( src->fd < 0 ){
		err = "can't read packed database '" + path + "'.";
		return false;
	}
	char	magic[ 8 ];
	int64_t	hdr[ 5 ];
	if( !pread_all( src->fd, magic, 8, 0 ) || memcmp( magic, PACK_MAGIC, 8 ) || !pread_all( src->fd, hdr, sizeof( hdr ), 8 ) ){
		err = bad;
		return false;
	}
	for( int i = 0; i < 5; i++ )
		if( hdr[ i ] < 0 || hdr[ i ] > ( int64_t( 1 ) << 40 ) ){
			err = bad;
			return false;
		}
	{
		// the arrays the header announces must be in the file: a truncated or corrupt pack is refused
		// before anything of the announced sizes is allocated
		struct stat	sb;
		const int64_t	need = 8 + int64_t( sizeof( hdr ) ) + 20 * hdr[ 0 ] + 4 * ( hdr[ 1 ] + hdr[ 2 ] ) + hdr[ 3 ] + hdr[ 4 ];
		if( fstat( src->fd, &sb ) != 0 || int64_t( sb.st_size ) < need ){
			err = bad;
			return false;
		}
	}
	const int64_t	n = hdr[ 0 ];
	int64_t	at = 8 + int64_t( sizeof( hdr ) );
	slen.resize( size_t( n ) );
	base_off.resize( size_t( n ) );
	exc_off.resize( size_t( n ) );
	exc.resize( size_t( hdr[ 3 ] ) );
	text.resize( size_t( hdr[ 4 ] ) );
	bool	ok = pread_all( src->fd, slen.data(), size_t( n ) * 4, at );
	at += n * 4;
	ok = ok && pread_all( src->fd, base_off.data(), size_t( n ) * 8, at );
	at += n * 8;
	ok = ok && pread_all( src->fd, exc_off.data(), size_t( n ) * 8, at );
	at += n * 8;
	src->codes_at = at;
	at += hdr[ 1 ] * 4;
	src->amask_at = at;
	at += hdr[ 2 ] * 4;
	ok = ok && pread_all( src->fd, exc.data(), exc.size(), at );
	at += hdr[ 3 ];
	ok = ok && pread_all( src->fd, text.data(), text.size(), at );
	codes.resize( size_t( hdr[ 1 ] ) );
	amask.resize( size_t( hdr[ 2 ] ) );
	codes_have_ = amask_have_ = 0;
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
		err = bad;
		return false;
	}
	index_text();
	src_ = src;
	return true;
}

bool PackFile::ensure( int n, std::string &err )
{
	if( !src_ )
		return true;		// (built in memory, or read to its end)
	const bool	all = n >= count();
	size_t	c_to = codes.size(), m_to = amask.size();
	if( !all && n > 0 ){
		const int64_t	end = base_off[ n - 1 ] + ( ( int64_t( slen[ n - 1 ] ) + 31 ) / 32 ) * 32;
		c_to = size_t( end / 16 );
		m_to = size_t( end / 32 );
	}else if( !all )
		c_to = m_to = 0;
	bool	ok = true;
	if( c_to > codes_have_ ){
		ok = pread_parallel( src_->fd, codes.data() + codes_have_, ( c_to - codes_have_ ) * 4, src_->codes_at + int64_t( codes_have_ ) * 4 );
		codes_have_ = c_to;
	}
	if( ok && m_to > amask_have_ ){
		ok = pread_parallel( src_->fd, amask.data() + amask_have_, ( m_to - amask_have_ ) * 4, src_->amask_at + int64_t( amask_have_ ) * 4 );
		amask_have_ = m_to;
	}
	if( !ok ){
		err = "read error on packed database '" + src_->path + "'.";
		return false;
	}
	if( all )
		src_.reset();
	return true;
}

bool PackFile::ensure_range( int first, int count, std::string &err )
{
	if( !src_ || count <= 0 )
		return true;
	if( first < 0 || first + count > this->count() ){
		err = "entries outside the packed database '" + src_->path + "'.";
		return false;
	}
	const int	last = first + count - 1;
	const int64_t	b0 = base_off[ first ], b1 = base_off[ last ] + ( ( int64_t( slen[ last ] ) + 31 ) / 32 ) * 32;
	const size_t	c0 = size_t( b0 / 16 ), c1 = size_t( b1 / 16 ), m0 = size_t( b0 / 32 ), m1 = size_t( b1 / 32 );
	bool	ok = c1 <= c0 || pread_parallel( src_->fd, codes.data() + c0, ( c1 - c0 ) * 4, src_->codes_at + int64_t( c0 ) * 4 );
	ok = ok && ( m1 <= m0 || pread_parallel( src_->fd, amask.data() + m0, ( m1 - m0 ) * 4, src_->amask_at + int64_t( m0 ) * 4 ) );
	if( !ok )
		err = "read error on packed database '" + src_->path + "'.";
	return ok;
}

bool PackFile::load( const std::string &path, std::string &err )
{
	return open( path, err ) && ensure( count(), err );
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

// rm_stream.cpp -- see rm_stream.h.
#include "rm_stream.h"
#include <algorithm>
#include <cctype>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace rma {

namespace {

// letter classes: 0..3 the codes of acgt (u = t), 4 any other letter, 255 not a letter
struct Classes {
	unsigned char	cls[ 256 ];
	Classes()
	{
		for( int c = 0; c < 256; c++ )
			cls[ c ] = isalpha( c ) ? 4 : 255;
		cls[ 'a' ] = cls[ 'A' ] = 0;
		cls[ 'c' ] = cls[ 'C' ] = 1;
		cls[ 'g' ] = cls[ 'G' ] = 2;
		cls[ 't' ] = cls[ 'T' ] = cls[ 'u' ] = cls[ 'U' ] = 3;
	}
};
const Classes	CLASSES;

inline bool is_space( int c ) { return c == ' ' || ( c >= '\t' && c <= '\r' ); }	// isspace(), "C" locale

}	// namespace

FastaStream::~FastaStream()
{
	{
		std::lock_guard<std::mutex>	lk( mu_ );
		quit_ = true;
	}
	cv_room_.notify_all();
	for( std::thread &t : pool_ )
		t.join();
	if( map_ != nullptr )
		munmap( const_cast<char *>( map_ ), size_ );
}

bool FastaStream::open( const std::string &path, int maxslen, int threads )
{
	const int	fd = ::open( path.c_str(), O_RDONLY );
	if( fd < 0 )
		return false;
	struct stat	sb;
	if( fstat( fd, &sb ) != 0 || !S_ISREG( sb.st_mode ) ){
		close( fd );
		return false;
	}
	size_ = size_t( sb.st_size );
	maxslen_ = maxslen;
	if( size_ > 0 ){
		void	*m = mmap( nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0 );
		if( m == MAP_FAILED ){
			close( fd );
			return false;
		}
		map_ = static_cast<const char *>( m );
		madvise( m, size_, MADV_SEQUENTIAL );
	}
	close( fd );
	if( size_ == 0 )
		return true;		// no entry: next() says so
	threads = std::max( 1, std::min( threads, 64 ) );
	// every '>' of the file, found by all threads at once ...
	std::vector<std::vector<size_t>>	found;
	found.resize( size_t( threads ) );
	{
		std::vector<std::thread>	ts;
		const size_t	chunk = ( size_ + threads - 1 ) / threads;
		for( int t = 0; t < threads; t++ )
			ts.emplace_back( [ this, t, chunk, &found ](){
				const size_t	lo = std::min( size_, size_t( t ) * chunk ), hi = std::min( size_, lo + chunk );
				const char	*p = map_ + lo, *e = map_ + hi;
				while( p < e && ( p = static_cast<const char *>( memchr( p, '>', size_t( e - p ) ) ) ) != nullptr ){
					found[ size_t( t ) ].push_back( size_t( p - map_ ) );
					p++;
				}
			} );
		for( std::thread &t : ts )
			t.join();
	}
	std::vector<size_t>	gt;
	for( const std::vector<size_t> &f : found )
		gt.insert( gt.end(), f.begin(), f.end() );
	// ... of which those inside a definition line do not start an entry (the reader takes the
	// line as it is, dbutil.c:84-102): one pass over the entries
	starts_.clear();
	starts_.push_back( 0 );		// (a file that does not begin with '>' is the first entry's anomaly)
	size_t	g = 0;
	while( g < gt.size() && gt[ g ] == 0 )
		g++;
	for( size_t cur = 0; ; ){
		const char	*nl = static_cast<const char *>( memchr( map_ + cur, '\n', size_ - cur ) );
		const size_t	line_end = nl ? size_t( nl - map_ ) : size_;
		while( g < gt.size() && gt[ g ] <= line_end )
			g++;
		if( g == gt.size() )
			break;
		cur = gt[ g++ ];
		starts_.push_back( cur );
	}
	const size_t	n = starts_.size();
	starts_.push_back( size_ );
	// workers take runs of entries so that a run is a few hundred KB of text
	const size_t	avg = std::max<size_t>( 1, size_ / n );
	const size_t	run = std::max<size_t>( 1, std::min<size_t>( 64, ( size_t( 256 ) << 10 ) / avg ) );
	ring_ = std::max<size_t>( 4 * run * size_t( threads ), 64 );
	run_ = run;
	entries_.resize( ring_ );
	for( int t = 0; t < threads; t++ )
		pool_.emplace_back( [ this ](){ worker(); } );
	return true;
}

void FastaStream::worker()
{
	const size_t	n = starts_.size() - 1;
	for( ; ; ){
		const size_t	i0 = claim_.fetch_add( run_ );
		if( i0 >= n )
			return;
		for( size_t i = i0; i < std::min( n, i0 + run_ ); i++ ){
			{
				std::unique_lock<std::mutex>	lk( mu_ );
				cv_room_.wait( lk, [ & ]{ return quit_ || i < consumed_ + ring_; } );
				if( quit_ )
					return;
			}
			Entry	&e = entries_[ i % ring_ ];
			parse( i, e );
			{
				std::lock_guard<std::mutex>	lk( mu_ );
				e.done = true;
			}
			cv_done_.notify_all();
		}
	}
}

// FN_fgetseq(), dbutil.c:42-128, on the bytes of entry i
void FastaStream::parse( size_t i, Entry &e ) const
{
	e.sid.clear();
	e.sdef.clear();
	e.exc.clear();
	e.slen = 0;
	e.anomaly = false;
	const char	*p = map_ + starts_[ i ], *const end = map_ + starts_[ i + 1 ];
	if( *p != '>' ){
		e.anomaly = true;		// "fastn file does not begin with '>'"
		return;
	}
	p++;
	while( p < end && is_space( ( unsigned char )*p ) && *p != '\n' )	// skipbl2nl
		p++;
	if( p == end || *p == '\n' ){
		e.anomaly = true;		// "unnamed entry"
		return;
	}
	while( p < end && !is_space( ( unsigned char )*p ) ){
		if( e.sid.size() < 99 )
			e.sid.push_back( *p );
		p++;
	}
	if( p < end && *p != '\n' ){
		while( p < end && is_space( ( unsigned char )*p ) && *p != '\n' )
			p++;
	}
	if( p < end && *p != '\n' ){
		const char	*nl = static_cast<const char *>( memchr( p, '\n', size_t( end - p ) ) );
		const char	*stop = nl ? nl : end;
		if( stop - p >= 20000 - 1 || memchr( p, '\0', size_t( stop - p ) ) != nullptr ){
			e.anomaly = true;	// definition line to be truncated, or cut at a NUL
			return;
		}
		e.sdef.assign( p, size_t( stop - p ) );
		p = stop;
	}
	// the letters: one mask word and two code words per 32 bases
	const size_t	bound = size_t( end - p );
	e.codes.assign( ( bound + 31 ) / 32 * 2 + 2, 0 );
	e.amask.assign( ( bound + 31 ) / 32 + 1, 0 );
	uint32_t	*cw = e.codes.data(), *mw = e.amask.data();
	const unsigned char	*cls = CLASSES.cls;
	size_t	n = 0;
	for( const unsigned char *q = reinterpret_cast<const unsigned char *>( p ),
		*qe = reinterpret_cast<const unsigned char *>( end ); q < qe; q++ ){
		const unsigned	c = cls[ *q ];
		if( c == 255 )
			continue;
		if( c < 4 )
			cw[ n >> 4 ] |= c << ( 2 * ( n & 15 ) );
		else{
			mw[ n >> 5 ] |= 1u << ( n & 31 );
			e.exc.push_back( char( tolower( *q ) ) );
		}
		n++;
	}
	if( n >= size_t( maxslen_ ) ){
		e.anomaly = true;		// sequence to be truncated (or at the limit: left to the reader)
		return;
	}
	e.slen = int32_t( n );
	const size_t	nw1 = ( n + 31 ) / 32;
	e.codes.resize( nw1 * 2 );
	e.amask.resize( nw1 );
}

std::unique_ptr<PackFile> FastaStream::next( int64_t batch_bases )
{
	if( map_ == nullptr || stopped_at_ >= 0 )
		return nullptr;
	const size_t	n = starts_.size() - 1;
	std::unique_ptr<PackFile>	pk( new PackFile );
	while( consumed_ < n ){
		Entry	&e = entries_[ consumed_ % ring_ ];
		{
			std::unique_lock<std::mutex>	lk( mu_ );
			cv_done_.wait( lk, [ & ]{ return e.done; } );
		}
		if( e.anomaly ){
			stopped_at_ = int64_t( starts_[ consumed_ ] );
			{
				std::lock_guard<std::mutex>	lk( mu_ );
				quit_ = true;
			}
			cv_room_.notify_all();
			break;
		}
		pk->append_packed( e.sid, e.sdef, e.codes, e.amask, e.exc, e.slen );
		e.done = false;
		{
			std::lock_guard<std::mutex>	lk( mu_ );
			consumed_++;
		}
		cv_room_.notify_all();
		if( pk->total_bases >= batch_bases )
			break;
	}
	if( pk->count() == 0 )
		return nullptr;
	return pk;
}

}	// namespace rma

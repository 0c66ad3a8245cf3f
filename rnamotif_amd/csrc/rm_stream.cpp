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
	if( map_ != nullptr && !keep_map_ )
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
	starts_.push_back( size_ );
	threads_ = threads;
	return true;
}

// Words of packed databases that have been handed out and freed again: the next batch takes them
// instead of fresh memory, which would be touched for the first time -- a page fault per 4 KB --
// while it is being filled (rm_pack.cpp's buffers are not zero-filled for the same reason).
namespace {
std::mutex	g_spare_mu;
std::vector<PackWords>	g_spare;
}

void recycle_words( PackWords &&w )
{
	if( w.capacity() < ( size_t( 1 ) << 18 ) )
		return;
	std::lock_guard<std::mutex>	lk( g_spare_mu );
	if( g_spare.size() < 12 )
		g_spare.push_back( std::move( w ) );
}

static PackWords take_words( size_t n )
{
	PackWords	w;
	{
		std::lock_guard<std::mutex>	lk( g_spare_mu );
		int	best = -1;
		for( size_t i = 0; i < g_spare.size(); i++ )
			if( g_spare[ i ].capacity() >= n && ( best < 0 || g_spare[ i ].capacity() < g_spare[ size_t( best ) ].capacity() ) )
				best = int( i );
		if( best >= 0 ){
			w = std::move( g_spare[ size_t( best ) ] );
			g_spare.erase( g_spare.begin() + best );
		}
	}
	w.resize( n );		// (no fill: NoInitAlloc)
	return w;
}

bool FastaStream::plan_to( size_t i )
{
	const size_t	n = starts_.size() - 1;
	while( planned_ <= i ){
		if( quit_ || planned_ >= n )
			return false;
		std::unique_ptr<Plan>	pl( new Plan );
		pl->first = planned_;
		int64_t	bytes = 0, mask_words = 0;
		while( planned_ < n && ( pl->count == 0 || bytes < batch_bytes_ ) ){
			const int64_t	extent = int64_t( starts_[ planned_ + 1 ] - starts_[ planned_ ] );
			pl->base_off.push_back( mask_words * 32 );
			mask_words += ( extent + 31 ) / 32;
			bytes += extent;
			pl->count++;
			planned_++;
		}
		pl->meta.resize( pl->count );
		pl->pk.reset( new PackFile );
		pl->pk->codes = take_words( size_t( mask_words ) * 2 );
		pl->pk->amask = take_words( size_t( mask_words ) );
		plans_.push_back( std::move( pl ) );
	}
	return true;
}

void FastaStream::worker()
{
	const size_t	n = starts_.size() - 1;
	for( ; ; ){
		const size_t	i = claim_.fetch_add( 1 );
		if( i >= n )
			return;
		Plan	*pl = nullptr;
		{
			std::unique_lock<std::mutex>	lk( mu_ );
			// not more than a few batches ahead of the reader of the batches
			cv_room_.wait( lk, [ & ]{ return quit_ || i < planned_ || plans_.size() < 4; } );
			if( quit_ || !plan_to( i ) )
				return;
			for( auto &p : plans_ )
				if( i >= p->first && i < p->first + p->count )
					pl = p.get();
			if( pl == nullptr )
				return;		// (its batch has been handed out truncated: the stream is over)
		}
		const size_t	k = i - pl->first;
		parse( i, pl->meta[ k ], pl->pk->codes.data() + pl->base_off[ k ] / 16, pl->pk->amask.data() + pl->base_off[ k ] / 32 );
		{
			std::lock_guard<std::mutex>	lk( mu_ );
			pl->done++;
		}
		cv_done_.notify_all();
	}
}

// FN_fgetseq(), dbutil.c:42-128, on the bytes of entry i; the packed letters go to cw / mw, which
// have room for as many letters as the entry has bytes
void FastaStream::parse( size_t i, Meta &e, uint32_t *cw, uint32_t *mw ) const
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
	// the letters: sixteen to a code word, thirty-two to a mask word, each word stored once
	const unsigned char	*cls = CLASSES.cls;
	size_t	n = 0;
	uint32_t	cacc = 0, macc = 0;
	for( const unsigned char *q = reinterpret_cast<const unsigned char *>( p ),
		*qe = reinterpret_cast<const unsigned char *>( end ); q < qe; q++ ){
		const unsigned	c = cls[ *q ];
		if( c == 255 )
			continue;
		if( c < 4 )
			cacc |= c << ( 2 * ( n & 15 ) );
		else{
			macc |= 1u << ( n & 31 );
			e.exc.push_back( char( tolower( *q ) ) );
		}
		n++;
		if( ( n & 15 ) == 0 ){
			cw[ ( n >> 4 ) - 1 ] = cacc;
			cacc = 0;
			if( ( n & 31 ) == 0 ){
				mw[ ( n >> 5 ) - 1 ] = macc;
				macc = 0;
			}
		}
	}
	// the words the last letters leave unfinished, up to the entry's last 32-base boundary
	size_t	written = n >> 4;
	if( n & 15 )
		cw[ written++ ] = cacc;
	if( written & 1 )
		cw[ written ] = 0;
	if( n & 31 )
		mw[ n >> 5 ] = macc;
	if( n >= size_t( maxslen_ ) ){
		e.anomaly = true;		// sequence to be truncated (or at the limit: left to the reader)
		return;
	}
	e.slen = int32_t( n );
}

bool FastaStream::read_entries( const int32_t *which, size_t n, PackFile &pk )
{
	if( n == 0 )
		return true;
	if( map_ == nullptr )
		return false;
	std::vector<int64_t>	base_off( n );
	int64_t	mask_words = 0;
	for( size_t k = 0; k < n; k++ ){
		if( which[ k ] < 0 || size_t( which[ k ] ) >= n_entries() )
			return false;
		base_off[ k ] = mask_words * 32;
		mask_words += ( extent( size_t( which[ k ] ) ) + 31 ) / 32;
	}
	const size_t	c0 = pk.codes.size(), m0 = pk.amask.size();
	pk.codes.resize( c0 + size_t( mask_words ) * 2 );
	pk.amask.resize( m0 + size_t( mask_words ) );
	std::vector<Meta>	meta( n );
	std::atomic<size_t>	next_k{ 0 };
	std::vector<std::thread>	ts;
	const int	nt = int( std::min<size_t>( size_t( std::max( threads_, 1 ) ), n ) );
	for( int t = 0; t < nt; t++ )
		ts.emplace_back( [ & ](){
			for( size_t k; ( k = next_k.fetch_add( 1 ) ) < n; )
				parse( size_t( which[ k ] ), meta[ k ], pk.codes.data() + c0 + base_off[ k ] / 16, pk.amask.data() + m0 + base_off[ k ] / 32 );
		} );
	for( std::thread &t : ts )
		t.join();
	for( size_t k = 0; k < n; k++ )
		if( meta[ k ].anomaly ){
			pk.codes.resize( c0 );
			pk.amask.resize( m0 );
			return false;
		}
	for( size_t k = 0; k < n; k++ ){
		Meta	&m = meta[ k ];
		pk.base_off.push_back( int64_t( m0 ) * 32 + base_off[ k ] );
		pk.exc_off.push_back( int64_t( pk.exc.size() ) );
		pk.slen.push_back( m.slen );
		pk.total_bases += m.slen;
		pk.exc.insert( pk.exc.end(), m.exc.begin(), m.exc.end() );
		pk.sid_off.push_back( int64_t( pk.text.size() ) );
		pk.text.insert( pk.text.end(), m.sid.c_str(), m.sid.c_str() + strlen( m.sid.c_str() ) + 1 );
		pk.sdef_off.push_back( int64_t( pk.text.size() ) );
		pk.text.insert( pk.text.end(), m.sdef.c_str(), m.sdef.c_str() + strlen( m.sdef.c_str() ) + 1 );
	}
	return true;
}

std::unique_ptr<PackFile> FastaStream::next( int64_t batch_bases )
{
	if( map_ == nullptr || stopped_at_ >= 0 )
		return nullptr;
	const size_t	n = starts_.size() - 1;
	if( !started_ ){
		started_ = true;
		batch_bytes_ = std::max<int64_t>( batch_bases, 1 );
		for( int t = 0; t < threads_; t++ )
			pool_.emplace_back( [ this ](){ worker(); } );
	}
	std::unique_ptr<Plan>	pl;
	{
		std::unique_lock<std::mutex>	lk( mu_ );
		if( plans_.empty() && ( planned_ >= n || !plan_to( planned_ ) ) )
			return nullptr;
		Plan	*front = plans_.front().get();
		cv_done_.wait( lk, [ & ]{ return front->done == front->count; } );
		pl = std::move( plans_.front() );
		plans_.pop_front();
	}
	cv_room_.notify_all();
	// the entries in order, up to the first one the serial reader has something to say about
	PackFile	&pk = *pl->pk;
	size_t	good = 0;
	while( good < pl->count && !pl->meta[ good ].anomaly )
		good++;
	if( good < pl->count ){
		stopped_at_ = int64_t( starts_[ pl->first + good ] );
		{
			std::lock_guard<std::mutex>	lk( mu_ );
			quit_ = true;
		}
		cv_room_.notify_all();
	}
	if( good == 0 )
		return nullptr;
	for( size_t k = 0; k < good; k++ ){
		Meta	&m = pl->meta[ k ];
		pk.base_off.push_back( pl->base_off[ k ] );
		pk.exc_off.push_back( int64_t( pk.exc.size() ) );
		pk.slen.push_back( m.slen );
		pk.total_bases += m.slen;
		pk.exc.insert( pk.exc.end(), m.exc.begin(), m.exc.end() );
		pk.sid_off.push_back( int64_t( pk.text.size() ) );
		pk.text.insert( pk.text.end(), m.sid.c_str(), m.sid.c_str() + strlen( m.sid.c_str() ) + 1 );
		pk.sdef_off.push_back( int64_t( pk.text.size() ) );
		pk.text.insert( pk.text.end(), m.sdef.c_str(), m.sdef.c_str() + strlen( m.sdef.c_str() ) + 1 );
	}
	// the arrays end with the last entry taken
	const int64_t	end_words = pl->base_off[ good - 1 ] / 32 + ( int64_t( pl->meta[ good - 1 ].slen ) + 31 ) / 32;
	pk.codes.resize( size_t( end_words ) * 2 );
	pk.amask.resize( size_t( end_words ) );
	return std::move( pl->pk );
}

}	// namespace rma

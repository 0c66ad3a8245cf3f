// rm_pack.h -- persistent packed database: the 2-bit + ambiguity-mask layout the
// scanner keeps in HBM (rm_fasta.h), written to disk once so that repeat scans skip
// reading and packing the text.  Everything print_match()
// needs survives: entry names, definition lines and the letters at ambiguous
// positions, so the original (lower-cased, u->t) sequence text can be rebuilt for
// any entry -- FN_fgetseq's output, /root/reference/src/dbutil.c:42-128.
//
// File layout (little endian):
//   char     magic[8]  "RMAPACK1"
//   int64    n_seq, n_code_words, n_mask_words, n_exc, n_text
//   int32    slen[n_seq]
//   int64    base_off[n_seq]        offset in bases, multiple of 32
//   int64    exc_off[n_seq]         index of the entry's first letter in exc[]
//   uint32   codes[n_code_words]    16 bases per word
//   uint32   amask[n_mask_words]    32 bases per word
//   char     exc[n_exc]             letters at the masked positions, in order
//   char     text[n_text]           sid\0sdef\0 per entry
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "rm_fasta.h"

namespace rma {

// len bytes (a multiple of 2 MB) on a 2 MB boundary, huge pages requested; throws std::bad_alloc
void	*pack_map_pages( size_t len );
void	pack_unmap_pages( void *p, size_t len );

// vector<uint32_t> whose resize( n ) does not zero what a read is about to fill (a gigabase is
// 375 MB of words; resize( n, 0 ) still zeroes)
template< class T >
struct NoInitAlloc {
	typedef T	value_type;
	template< class U > struct rebind { typedef NoInitAlloc<U> other; };
	NoInitAlloc() = default;
	template< class U > NoInitAlloc( const NoInitAlloc<U> & ) {}
	template< class U > void construct( U *p ) { ::new( static_cast<void *>( p ) ) U; }
	template< class U, class... A > void construct( U *p, A &&... a ) { ::new( static_cast<void *>( p ) ) U( std::forward<A>( a )... ); }
	// Large arrays come as whole 2 MB pages where the system grants them (transparent huge pages on
	// request): a gigabase of packed words is 96 000 small pages, each a fault when it is first written
	// -- 5 ms per 36 MB measured, more than the upload of those 36 MB takes -- or 190 large ones.
	static constexpr size_t	HUGE = size_t( 2 ) << 20, BIG = size_t( 8 ) << 20;
	static size_t	mapped( size_t n ) { return ( n * sizeof( T ) + HUGE - 1 ) & ~( HUGE - 1 ); }
	T	*allocate( size_t n )
	{
		if( n * sizeof( T ) < BIG )
			return static_cast<T *>( ::operator new( n * sizeof( T ) ) );
		return static_cast<T *>( pack_map_pages( mapped( n ) ) );
	}
	void	deallocate( T *p, size_t n )
	{
		if( n * sizeof( T ) < BIG )
			::operator delete( p );
		else
			pack_unmap_pages( p, mapped( n ) );
	}
	template< class U > bool operator==( const NoInitAlloc<U> & ) const { return true; }
	template< class U > bool operator!=( const NoInitAlloc<U> & ) const { return false; }
};
typedef std::vector<uint32_t, NoInitAlloc<uint32_t>>	PackWords;

struct PackFile {
	std::vector<int32_t>	slen;
	std::vector<int64_t>	base_off, exc_off;
	PackWords	codes, amask;
	// codes / amask page-locked for DMA (rma_pack_pin, rm_scanner.cpp): released before the vectors
	// above go away (members are destroyed in reverse order), handed on when the pack is moved
	struct Pin {
		void	*c = nullptr, *m = nullptr;
		void	( *unreg )( void * ) = nullptr;
		Pin() = default;
		Pin( Pin &&o ) noexcept : c( o.c ), m( o.m ), unreg( o.unreg ) { o.unreg = nullptr; }
		Pin &operator=( Pin &&o ) noexcept { release(); c = o.c; m = o.m; unreg = o.unreg; o.unreg = nullptr; return *this; }
		Pin( const Pin & ) = delete;
		Pin &operator=( const Pin & ) = delete;
		void	release() { if( unreg ){ if( c ) unreg( c ); if( m ) unreg( m ); unreg = nullptr; } }
		~Pin() { release(); }
	}	pin;
	std::vector<char>	exc, text;
	std::vector<int64_t>	sid_off, sdef_off;	// into text (built on load)
	int64_t	total_bases = 0;

	int	count() const { return int( slen.size() ); }
	const char	*sid( int i ) const { return text.data() + sid_off[ i ]; }
	const char	*sdef( int i ) const { return text.data() + sdef_off[ i ]; }
	void	add( const SeqRecord &rec );
	// an entry that is packed already (rm_stream.cpp): codes/amask hold 2 and 1 words per 32 bases
	void	append_packed( const std::string &sid, const std::string &sdef, const std::vector<uint32_t> &codes,
			const std::vector<uint32_t> &amask, const std::vector<char> &exc, int32_t slen );
	std::string	unpack( int i ) const;		// the entry's sequence text
	// Letters lo .. hi-1 of strand comp of entry i into out[ lo .. hi ): comp 0 the text as the
	// readers deliver it, comp 1 its reverse complement with every letter that is not acgt an n
	// (mk_rcmp, rnamot.c:193-216).  What print_match() and the score program read of an entry is
	// the span of a hit, not the entry.
	void	window( int i, int comp, int lo, int hi, char *out ) const;
	bool	save( const std::string &path, std::string &err ) const;
	bool	load( const std::string &path, std::string &err );
	// The same in two parts, for a reader that hands the entries on in batches: open() reads the
	// tables, names and exceptions and checks them; ensure( n ) reads the packed bases of entries
	// 0 .. n-1 if they are not there yet (entries lie in the file in order) -- the first batch is
	// on its way to the GPU when a tenth of the file has been read.  codes/amask have their final
	// size after open(): readers of earlier entries are not disturbed by ensure().
	bool	open( const std::string &path, std::string &err );
	bool	ensure( int n, std::string &err );
	// after open(): the packed bases of entries first .. first+count-1 only (a rank of a multi-GPU
	// search reads its share; pages of the arrays that belong to other entries are never touched)
	bool	ensure_range( int first, int count, std::string &err );
	static bool	is_pack( const std::string &path );
private:
	void	index_text();
	struct Source;
	std::shared_ptr<Source>	src_;		// the open file while parts of it are still to be read
	size_t	codes_have_ = 0, amask_have_ = 0;
};

}	// namespace rma

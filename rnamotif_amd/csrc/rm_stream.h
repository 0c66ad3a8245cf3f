// rm_stream.h -- the fast way from a FASTA file to packed entries: the file is mapped, cut
// into entries at its '>' characters (FN_fgetseq ends an entry at any '>', /root/reference/
// src/dbutil.c:104-121) and the entries are parsed and packed by worker threads straight into
// the layout the scanner keeps in HBM (rm_fasta.h) -- no intermediate text.  What
// print_match() needs later is rebuilt from the packed form for the window of a hit
// (PackFile::window).
//
// The parallel path takes the regular case only.  An entry that needs one of the reader's
// diagnostics (a file that does not start with '>', an unnamed entry, a definition line or a
// sequence that has to be truncated, a NUL in the definition line) ends it: the caller goes
// on from that entry's file offset with the serial reader (rm_fasta.cpp), which prints what
// the reference prints.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "rm_pack.h"

namespace rma {

// the packed words of a batch that is done with, for the next batch to take (see rm_stream.cpp)
void	recycle_words( PackWords &&w );

class FastaStream {
public:
	FastaStream() = default;
	~FastaStream();
	FastaStream( const FastaStream & ) = delete;
	FastaStream &operator=( const FastaStream & ) = delete;
	// false: not a regular file that can be mapped (the caller reads it serially)
	bool	open( const std::string &path, int maxslen, int threads );
	// The next batch of whole entries, about batch_bases bytes of the file, in file order; null at
	// the end of the fast path.  Then stopped_at() is the file offset the serial reader has to go on
	// from, or -1 when the file is finished.  (The batch size of the first call holds for the file.)
	std::unique_ptr<PackFile>	next( int64_t batch_bases );
	int64_t	stopped_at() const { return stopped_at_; }
	// The entries of the file as open() found them (before anything is parsed): how many, and how many
	// bytes of the file each one takes -- an upper bound of its letters.  What the ranks of a multi-GPU
	// search divide among themselves without reading the database (rma_database_index).
	size_t	n_entries() const { return starts_.empty() ? 0 : starts_.size() - 1; }
	int64_t	extent( size_t i ) const { return int64_t( starts_[ i + 1 ] - starts_[ i ] ); }
	// Entries which[ 0 .. n ) (ascending) parsed and packed into pk, nothing else of the file touched.
	// false: one of them is an entry the serial reader has something to say about (nothing is returned).
	// Not to be mixed with next() on the same stream.
	bool	read_entries( const int32_t *which, size_t n, PackFile &pk );
	// Leave the file mapped when the stream goes away.  Unmapping a gigabyte takes the address space's
	// lock for tens of milliseconds, during which no other thread of the process gets a page fault
	// served -- the uploads and scans of the last batches wait (measured: 20-30 ms each instead of 2.5).
	// The command line program leaves its mappings to the end of the process.
	void	keep_mapping() { keep_map_ = true; }
private:
	bool	keep_map_ = false;
	// what a worker leaves of one entry besides its packed words, which it writes straight into the
	// batch's arrays
	struct Meta {
		std::string	sid, sdef;
		std::vector<char>	exc;
		int32_t	slen = 0;
		bool	anomaly = false;
	};
	// A batch being filled: its entries and where each one's words go are fixed from the entries'
	// extents in the file before anything is parsed (an entry of e bytes has at most e letters; every
	// entry starts on a 32-base boundary), so the workers pack side by side into one pair of arrays and
	// nobody copies packed words afterwards.
	struct Plan {
		size_t	first = 0, count = 0;
		std::unique_ptr<PackFile>	pk;
		std::vector<int64_t>	base_off;	// per entry, in bases
		std::vector<Meta>	meta;
		size_t	done = 0;			// entries parsed (guarded by mu_)
	};
	void	worker();
	void	parse( size_t i, Meta &m, uint32_t *cw, uint32_t *mw ) const;
	bool	plan_to( size_t i );			// (mu_ held) plans exist up to entry i; false: the stream is over
	const char	*map_ = nullptr;
	size_t	size_ = 0;
	int	maxslen_ = 0, threads_ = 1;
	int64_t	batch_bytes_ = 0;
	std::vector<size_t>	starts_;		// offsets of the '>' characters, then size_
	std::deque<std::unique_ptr<Plan>>	plans_;		// in file order; the front is the next to hand out
	size_t	planned_ = 0;			// entries that belong to a plan
	std::atomic<size_t>	claim_{ 0 };		// next entry a worker takes
	bool	quit_ = false, started_ = false;
	std::mutex	mu_;
	std::condition_variable	cv_done_, cv_room_;
	std::vector<std::thread>	pool_;
	int64_t	stopped_at_ = -1;
};

}	// namespace rma

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
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "rm_pack.h"

namespace rma {

class FastaStream {
public:
	FastaStream() = default;
	~FastaStream();
	FastaStream( const FastaStream & ) = delete;
	FastaStream &operator=( const FastaStream & ) = delete;
	// false: not a regular file that can be mapped (the caller reads it serially)
	bool	open( const std::string &path, int maxslen, int threads );
	// The next batch of whole entries, about batch_bases bases, in file order; null at the
	// end of the fast path.  Then stopped_at() is the file offset the serial reader has to go on
	// from, or -1 when the file is finished.
	std::unique_ptr<PackFile>	next( int64_t batch_bases );
	int64_t	stopped_at() const { return stopped_at_; }
private:
	struct Entry {		// one parsed entry, waiting to be appended to a batch
		std::string	sid, sdef;
		std::vector<uint32_t>	codes, amask;
		std::vector<char>	exc;
		int32_t	slen = 0;
		bool	anomaly = false, done = false;
	};
	void	worker();
	void	parse( size_t i, Entry &e ) const;
	const char	*map_ = nullptr;
	size_t	size_ = 0;
	int	maxslen_ = 0;
	std::vector<size_t>	starts_;		// offsets of the '>' characters, then size_
	std::vector<Entry>	entries_;		// ring, indexed by entry number % ring size
	size_t	ring_ = 0, run_ = 1;		// ring size; entries a worker takes at a time
	std::atomic<size_t>	claim_{ 0 };		// next entry a worker takes
	size_t	consumed_ = 0;			// entries handed out (guarded by mu_)
	bool	quit_ = false;
	std::mutex	mu_;
	std::condition_variable	cv_done_, cv_room_;
	std::vector<std::thread>	pool_;
	int64_t	stopped_at_ = -1;
};

}	// namespace rma

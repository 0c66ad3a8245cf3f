// rm_driver.h -- the per-database search loop of the reference's main()
// (/root/reference/src/rnamot.c:100-190) around a pluggable scanner: read
// FASTA records, hand batches to the scanner, replay every reported candidate
// through the score program and print_match() in the reference's order.
#pragma once
#include <cstdio>
#include "rm_host.h"
#include "rm_fasta.h"
#include "rm_score.h"

namespace rma {

// What the driver needs from a scanner: scan n sequences (both strands when
// the program says so) and return hit records sorted by (seq,comp,szero,rank,order).
// The product binary plugs the HIP scanner in here (rm_capi.cpp).
struct PackFile;
struct ScanBackend {
	void	*self;
	int	( *scan )( void *self, const char *const *seqs, const int32_t *slens, int n,
			const int32_t **hits, int64_t *n_hits, char *err, size_t errlen );
	// optional: entries [first, first+count) of a packed database (rm_pack.h) without going through
	// text, in two halves so that a batch is uploaded while the one before is searched:
	// upload_packed() puts them into HBM and returns a handle; scan_uploaded() searches what the
	// handle holds -- hit records number the entries from 0 = first -- and releases it, whatever
	// it returns; drop_uploaded() releases a handle that is not going to be searched.  The pack must
	// stay as it is until one of the two has been called.
	int	( *upload_packed )( void *self, const PackFile *pk, int first, int count, void **handle,
			char *err, size_t errlen ) = nullptr;
	int	( *scan_uploaded )( void *self, void *handle, const int32_t **hits, int64_t *n_hits,
			char *err, size_t errlen ) = nullptr;
	void	( *drop_uploaded )( void *self, void *handle ) = nullptr;
};

struct SearchStats {
	int64_t	n_seqs = 0, n_bases = 0, n_candidates = 0, n_hits = 0;
};

// Replays candidates of one batch.  Exposed for the C-ABI.
class Replayer {
public:
	Replayer( Descriptor &d, const rma_program_t &prog, FILE *out );
	void	begin();		// BEGIN program, rnamot.c:154-155
	void	end();			// END program, rnamot.c:187-188
	// hits: n records of stride rma_hit_stride( prog ), sorted
	void	replay( const std::vector<SeqRecord> &batch, const int32_t *hits, int64_t n, SearchStats &st );
	// the same over entries first .. of a packed database (hit records count entries from first):
	// the text of an entry is rebuilt for the span of each hit only (PackFile::window), so a batch
	// of a hundred million bases with a few thousand hits costs a few thousand small windows
	void	replay_packed( const PackFile &pk, int first, const int32_t *hits, int64_t n, SearchStats &st );
	// parallel replay (rm_driver.cpp): a worker's replayer writes to a buffer of its own and never
	// prints the "#RM" header
	void	set_out( FILE *out, bool header );
	void	print_header( FILE *fp ) const { printer_.header( fp ); }
	bool	header_pending() const { return printer_.header_pending(); }
	void	header_done() { printer_.no_header(); }
private:
	void	one_hit( const int32_t *w, const char *sid, const char *sdef, int slen, const char *sbuf, SearchStats &st );
	std::vector<char>	text_;		// strand buffer of the entry in hand, filled window by window
	bool	loose_ = false;		// some element's seq= was tested loosely by the scan: the whole expression here (one_hit)
	std::string	chk_;
	Descriptor	&d_;
	const rma_program_t	&prog_;
	FILE	*out_;
	HitPrinter	printer_;
};

// rnamot.c:125-190.  batch_bases bounds how much sequence is handed to the
// scanner at once.
int	run_search( Descriptor &d, const rma_program_t &prog, ScanBackend &be, FILE *out,
		int64_t batch_bases, SearchStats *stats );

}	// namespace rma

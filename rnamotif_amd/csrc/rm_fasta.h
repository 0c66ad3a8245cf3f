// rm_fasta.h -- sequence database reader (FASTA) and the 2-bit + ambiguity
// mask packing the scanner keeps in HBM.  Record parsing follows FN_fgetseq,
// /root/reference/src/dbutil.c:42-128.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace rma {

struct SeqRecord {
	std::string	sid, sdef;
	std::string	seq;		// lower case, u -> t, every alpha kept
	bool	eof = false;		// FN_fgetseq returned EOF (no record)
};

// -fmt fastn | pir | gb (rnamot.c:126-138): FN_fgetseq dbutil.c:42, PIR_fgetseq :130,
// GB_fgetseq :226
enum SeqFormat { FMT_FASTN = 0, FMT_PIR = 1, FMT_GENBANK = 2 };
SeqFormat seq_format_of( const std::string &name );	// "" and "fastn" -> FMT_FASTN

class FastaReader {
public:
	explicit FastaReader( FILE *fp, int maxslen = 30000001, SeqFormat fmt = FMT_FASTN ) :
		fp_( fp ), maxslen_( maxslen ), fmt_( fmt ) {}
	// returns false at end of file; diagnostics go to stderr like the reference
	bool	next( SeqRecord &rec );
private:
	bool	next_fastn( SeqRecord &rec );
	bool	next_pir( SeqRecord &rec );
	bool	next_gb( SeqRecord &rec );
	// getc()/ungetc() of the reference's readers over a block buffer (the fastn and pir
	// readers; the GenBank reader works line by line with fgets on fp_ itself)
	int	get()
	{
		if( pos_ == len_ ){
			len_ = fread( buf_.data(), 1, buf_.size(), fp_ );
			pos_ = 0;
			if( len_ == 0 )
				return EOF;
		}
		return ( unsigned char )buf_[ pos_++ ];
	}
	void	unget() { pos_--; }	// only ever the character just read
	void	read_letters( const char *who, SeqRecord &rec );
	FILE	*fp_;
	int	maxslen_;
	SeqFormat	fmt_;
	std::vector<char>	buf_ = std::vector<char>( 1 << 20 );
	size_t	pos_ = 0, len_ = 0;
};

// Packed database layout (device side):
//   codes: 2 bits per base, 16 bases per uint32 word, base i of a sequence at
//          bits 2*(i%16) of word i/16; a=0 c=1 g=2 t=3, ambiguous letters 0
//   amask: 1 bit per base, 32 per uint32; set for every letter that is not acgt
// Each sequence starts on a word boundary of both arrays (32-base aligned).
struct PackedDb {
	std::vector<uint32_t>	codes, amask;
	std::vector<int64_t>	base_off;	// per sequence: offset in bases (multiple of 32)
	std::vector<int32_t>	slen;
	int64_t	total_bases = 0;		// sum of slen
	void	add( const char *seq, int slen );
	int64_t	padded_bases() const { return int64_t( amask.size() ) * 32; }
};

}	// namespace rma

// rm_regex.h -- ed-style regular expressions as used by seq= constraints,
// the score section's =~ / !~ operators and mismatches().
//
// Restates the behaviour of the reference's SysV compile()/step()/advance()
// (/root/reference/src/regexp.c:124-664) and of mm_step()/mm_seqlen()
// (/root/reference/src/mm_regexp.c:51-230,353-469) on a parsed op list instead
// of the SysV byte code.  Host side only; the device works on the reduced
// rma_regex_t form produced by re_to_atoms().
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "rnamotif_amd_program.h"

namespace rma {

enum ReKind : uint8_t {
	RE_CHR, RE_DOT, RE_CCL, RE_NCCL, RE_DOL,
	RE_BRA, RE_KET, RE_BACK, RE_BRC, RE_LET
};
enum ReRep : uint8_t { REP_ONE, REP_STAR, REP_RANGE };

struct ReOp {
	ReKind	kind;
	ReRep	rep = REP_ONE;
	uint8_t	c = 0;		// RE_CHR: the character; RE_BRA/KET/BACK: group number
	uint8_t	lo = 0, hi = 0;	// REP_RANGE: \{lo,hi\}; hi == 255 is "no limit"
	uint8_t	set[ 16 ] = {};	// RE_CCL/RE_NCCL: 128-bit membership
	bool	has( int ch ) const { return ( set[ ch >> 3 ] >> ( ch & 7 ) ) & 1; }
};

struct ReProg {
	std::vector<ReOp>	ops;
	int	err = 0;	// reference regerr code, 0 = ok
};

// compile(): pattern text -> ops.  A leading '^' is consumed and not stored;
// callers decide anchoring from the first pattern character themselves
// (find_motif.c:1818, score.c:1258).  Returns false and sets err on the
// reference's ERROR(n) cases.
bool	re_compile( const char *pat, ReProg &out );

struct ReMatch { const char *loc1 = nullptr, *loc2 = nullptr; };

// step(): true iff the expression matches somewhere in the NUL terminated s
// (only at s when anchored).
bool	re_step( const ReProg &re, const char *s, bool anchored, ReMatch *m = nullptr );

// mm_step(): mismatch tolerant variant for fixed length expressions; *n_mm is
// left exactly as the reference leaves it (count of the last attempt).
bool	re_mm_step( const ReProg &re, const char *s, bool anchored, int l_mm, int *n_mm );

// mm_seqlen( stp, .. ) without the best-literal (-O) analysis.  diag: what the reference's walk says on stderr about
// opcodes it does not know (the group numbers behind \) and \1: "mm_seqlen:  0?"), appended.
void	re_seqlen( const ReProg &re, bool caret, int *minl, int *maxl, int *mmok, std::string *diag = nullptr );

// Reduce to the device form; false (with a message) for what it cannot hold ('$' inside, too many atoms).  Back
// references, \< \> and letters that are not acgt come out as what the packed database can tell of them, and
// rma_regex_t::loose says that the expression itself still has to be applied to the text (the host does, at replay).
bool	re_to_atoms( const ReProg &re, bool caret, rma_regex_t *out, std::string &why );

}	// namespace rma

// rm_dev_program.h -- the motif program in the form the scan kernel reads.
//
// rma_program_t (include/rnamotif_amd_program.h) is the boundary blob; this is
// the same information reduced for the device: element table indexed by small
// integers, pair tables as bit sets, seq= constraints as bit-parallel position
// automata, the floating point parts of the helix rules folded into integer
// tables on the host (find_motif.c:1023-1033,1040,1086,1194,1245).  It is small
// enough to be copied into LDS by every workgroup.
#pragma once
#include <cstddef>
#include <cstdint>
#include "rnamotif_amd_program.h"

#ifndef RMD_MAX_ELEMS
#define RMD_MAX_ELEMS	100	// elements (and search levels) per descriptor: the reference's own limit, compile.c:49
#endif
#define RMD_LEAN_LEVELS	16	// most search levels the lean path takes
#define RMD_MAX_HLEN	63	// longest helix strand where sets of lengths are one 64-bit word: the lean path, and every general instance but one
#define RMD_MAX_HLEN_WIDE	127	// ... two words (rmd_lset_t, rm_scan_core.h): the general instance compiled for it (rmd_program_t::wide)
#define RMD_MAX_RE	20
#define RMD_MAX_PS	20
#define RMD_MAX_RULES	16	// helix groups with their own mispair / pairfrac rule tables
#define RMD_MAX_SITES	16	// the boundary's own limit (RMA_MAX_SITES)
#define RMD_MAX_EFN	16
#define RMD_MAX_PK	16	// improper (pseudoknot) helices per descriptor

// position automaton of one seq= expression: state i (bit i) accepts one base
struct rmd_regex_t {
	uint64_t	accept[ 5 ];	// accept[c]: states that take base code c
	uint64_t	opt;		// states that may be skipped
	uint64_t	star;		// states that may repeat
	uint64_t	dot;		// states that came from '.', for the mismatch counter
	int32_t	n_states;
	int32_t	n_close;		// longest run of skippable states
	int32_t	anchored, dollar;
	int32_t	fixed_len;		// >= 0: every state mandatory (mismatch mode legal)
	int32_t	n_prefix;		// leading states that cannot be skipped (0 unless anchored)
	int32_t	wide, pad_;		// >= 0: the expression has 64 .. 127 states and lives in rmd_regexes2( P )[ wide ]; the
					// members above hold no states then (no prefix, no literal, no pinned test comes from them)
};

// ... of 64 to 127 states: two words a set (round 3; the reference's own limit is the 100 bases of a strand)
#define RMD_MAX_RE2	8
struct rmd_regex2_t {
	uint64_t	accept[ 5 ][ 2 ];	// [ c ][ 0 ]: states 0..63, [ c ][ 1 ]: states 64..127
	uint64_t	opt[ 2 ], star[ 2 ], dot[ 2 ];
	int32_t	n_states, n_close;
};

struct rmd_pairset_t {
	uint32_t	mat2;
	uint32_t	mat3[ 4 ];
	uint32_t	mat4[ 20 ];
};

struct rmd_elem_t {
	int8_t	type, proper, ends, strict;
	int8_t	loop;			// find_motif: iterate over the end position
	int8_t	next_s;			// search level of s_next, or -1
	int8_t	inner_s;		// search level of s_inner, or -1
	int8_t	searchno;
	int8_t	inner;			// element index of s_inner, or -1
	int8_t	n_mates, n_scopes, scope;
	int8_t	mates[ 3 ];
	int8_t	scopes[ 8 ];
	int8_t	pairset, re;
	int8_t	pfrac;			// pairfrac rule active
	int8_t	quick;			// level searched with match_wchlx at (zero, sdollar): proper h5, q1
	int32_t	q_iminl;		// interior minimum of that match (find_motif.c:423,884)
	int32_t	q_sminl;		// first pseudoknot helix: least length after its 3' strand (:561)
	int32_t	q_smaxl;		// ... and the most (:562), -1: not bounded
	int32_t	minlen, maxlen, minglen, maxglen, minilen, maxilen;
	int32_t	mismatch;
	int32_t	mplim;			// match_wchlx/match_phlx mispair limit
	int32_t	rule;			// index into rules[] (helix strands), else 0
	// search-space pruning of the lean path (necessary conditions, output neutral):
	int16_t	rem_min;		// least total length of the groups that follow in the chain
	int16_t	rem_max;		// most (closed chains only), -1: unknown / open chain
	int8_t	tail_s;			// helix: level of the last group of its interior if that is a
					// proper helix (its 3' end is pinned to the interior's end), else -1
	int8_t	pk;			// improper helix: index into pks[], else -1
	int8_t	rows;			// Watson-Crick helix heading a level: pair row set of its pair table
					// (rmd_program_t::rowset_ps), -1: none
	int8_t	tup;			// t1 / q1: index into tups[] (first-tuple masks), else -1
	int16_t	tail_pre_min, tail_pre_max;	// total length of the interior groups before it (-1: unbounded)
	// ss heading a level, seq= without mismatches: what can be tested as soon as a level above pins
	// its window (necessary conditions for chk_seq() on the final string, output neutral)
	int8_t	pin_start;		// ^-anchored: its leading mandatory positions at the window start
	int8_t	pin_end_n;		// > 0: the ss is its whole window and the expression is $-anchored with
					// this fixed length: its last pin_end_n bases, once the window end is known
	int8_t	back_s;			// head of a level: the level to go back to when this one is exhausted -- the
					// nearest one below it that has more than one alternative (-1: the item is done)
	int8_t	head_s;			// helix: level of the first proper helix of its interior when only ss of bounded
					// total length lie before it (its 5' start is pinned to the interior's start), else -1
	int16_t	head_pre_min, head_pre_max;	// ... that total length
	// lean path, head of a search level: the level's digit in the order word of a candidate.  The depth-first
	// walk takes a level's end positions from the highest down and, at each, the helix lengths from the
	// shortest up; so candidates of one start position and rank come in the order of the number whose digit at
	// every level is ( first end - end ) * ord_nlen + ( length - minlen ), weight ord_stride (the levels
	// below it multiplied out).  rmd_program_t::ord_ok says the number fits 31 bits; then the kernels store it
	// as the order word and may walk an item in pieces, in any order (both sorts renumber the order words).
	int32_t	ord_stride;
	int16_t	ord_nlen;
	// lean path: the level to go back to when this one is exhausted -- the nearest one below it that may have
	// another alternative.  A single strand of one length (its one end position is taken) has none: going
	// back to it only to learn that costs a step of the walk, six of trna.descr's eleven levels are such.
	int8_t	lean_back_s, ord_pad_;
};

// First-tuple masks of a triplex / 4-plex pair table: match_triplex()/match_4plex() give up at
// once when the first triple / quad does not hold and the 5' end must be paired (find_motif.c:
// 1198-1206, 1249-1257); which bases can complete it, given the others, is a 5-bit mask.
struct rmd_tup_t {
	uint8_t	t2[ 25 ];		// [ b1 * 5 + b3 ]: second bases b2 with triple( b1, b2, b3 )
	uint8_t	q2[ 25 ];		// [ b1 * 5 + b4 ]: second bases b2 with quad( b1, b2, b3, b4 ) for some b3
	uint8_t	q3[ 125 ];		// [ ( b1 * 5 + b2 ) * 5 + b4 ]: third bases b3 with quad( b1, b2, b3, b4 )
	uint8_t	pad_[ 1 ];
};
#define RMD_MAX_TUP	8

// Improper (pseudoknot) helix: what find_pknot5()/find_pknot3() compute with find_minlen()/
// find_maxlen() (find_motif.c:495-665) over ranges of the knot's elements.  When helix i of a
// knot is searched, exactly the strands of its helices 0..i-1 are matched (the knot's helices
// head consecutive search levels, their interiors come later: find_search_order, compile.c:
// 3163-3190), so every such sum is a constant plus the current lengths of some earlier helices.
enum { RMD_PK_P = 0,	// scopes[0] .. d-1
	RMD_PK_R,	// d .. scopes[n-1]
	RMD_PK_I,	// d+1 .. d3-1
	RMD_PK_S,	// d3+1 .. scopes[n-1]
	RMD_PK_IL,	// second helix: d+1 .. (3' strand of the first helix)-1
	RMD_PK_IR,	// second helix: (3' strand of the first helix)+1 .. d3-1
	RMD_PK_N };
struct rmd_pk_t {
	int32_t	bmin[ RMD_PK_N ], bmax[ RMD_PK_N ];	// sum over the elements of the range that are not matched yet
	uint8_t	mask[ RMD_PK_N ];	// bit s: strand scopes[s] lies in the range and is matched (earlier helix)
	int8_t	lvl[ 8 ];		// search level of the helix that strand scopes[s] belongs to
	int8_t	hlx2;			// this is the knot's second helix (find_pknot3 :571)
	// upd_pksearches(), :667: levels whose window end becomes s5-1, start s5+hl, end s3-hl, start s3+1 (-1: none)
	int8_t	w_osd5, w_zero5, w_osd3, w_zero3;
	int8_t	pad_[ 3 ];
};

// length dependent helix rules, shared by the strands of one helix
struct rmd_rule_t {
	uint8_t	pf_maxmpr[ RMD_MAX_HLEN_WIDE + 1 ];	// pairfrac test: most mispairs allowed at length hl
	uint8_t	tq_mplim[ RMD_MAX_HLEN_WIDE + 1 ];	// match_triplex/match_4plex limit at length tlen
};

struct rmd_site_t {
	int8_t	n_pos, pairset;
	int8_t	elem[ 4 ], l2r[ 4 ];
	int16_t	offset[ 4 ];
};

// Look-ahead of the lean pre-filter (rm_scan_kernel.h, pass A): what the interior of the first search
// element -- a proper helix, the outer stem of a cloverleaf or of any nested motif -- needs to hold at
// all.  Its interior is a chain of groups: single strands, and helices with their own interiors.  A
// helix whose interior is single strands only (a stem-loop: "leaf") can start at a position only if
// its innermost hmin pairs are there for one of the loop lengths it allows -- a bit per position,
// computed for the whole tile from the pair rows by shifted ANDs -- and a group can start where the
// next group can start one of its lengths later.  Chained from the last group back to the first, this
// says for every start position of the motif whether its interior can exist at all; positions where it
// cannot are not searched (for trna.descr: 19 in 20).  Necessary conditions only: helices that allow
// mispairs, and everything that is not a stem-loop, count as "any".
#define RMD_MAX_CHAIN	8
struct rmd_chain_sib_t {
	int8_t	leaf;			// 1: stem-loop on the first element's pair table, no mispairs
	int8_t	hmin;			// leaf: pairs of its shortest helix
	int16_t	tmax;			// leaf: longest minus shortest helix (the core of hmin pairs lies 0 .. tmax in)
	int16_t	lmin, lmax;		// leaf: loop lengths
	int16_t	len_lo, len_hi;		// total length of the group
	int8_t	core_slot;		// leaf: which of the two kept core vectors is its own (-1: none kept)
	int8_t	pad_;
};
struct rmd_chain_t {
	int8_t	on, n;
	// The candidate test of pass A' (pooled instance) knows the 3' ends at which the first helix of the
	// interior (rmd_elem_t::head_s) can close; when only single strands of hn_glo .. hn_ghi bases lie between
	// it and the next stem-loop, one of those ends must have that stem-loop's core hn_glo + 1 .. hn_ghi + 1 +
	// hn_tmax behind it (trna.descr: a D arm that closes two bases before an anticodon arm -- one
	// candidate in twenty).  hn_on = 0: no such pair of neighbours.
	int8_t	hn_on, hn_slot;
	int16_t	hn_tmax, hn_glo, hn_ghi;
	int16_t	s_lo, s_hi;		// first group starts s_lo .. s_hi after the start position (the outer helix' lengths)
	rmd_chain_sib_t	sib[ RMD_MAX_CHAIN ];
};

// Pre-filter of a 4-plex that heads the search list (rm_scan_kernel.h, pass A): what its second and
// third strands demand of the bases alone.  A quad can only hold where the second strand's base is one
// that some quad of the pair set has in second place (m2), likewise the third (m3); match_4plex()
// (find_motif.c:1234-1290) gives up at the first quad when the 5' end must pair and lets at most
// tq_mplim further quads fail, so a strand can stand at a position only if its first base is in the
// set (first5) and at most badmax of its next nmin - 1 bases are not.  Start positions whose second
// strand finds no such place within reach (a_lo .. a_hi bases after the start), and end positions whose
// third strand finds none (b_hi .. b_lo bases before the end), are not queued: necessary conditions,
// the candidates stay what they are.  on = 0: no such filter (another first element, masks unknown).
// t_on: the 4-plex' group is followed (after at most one single strand) by a triplex -- qu+tr.descr.
// The same argument along the triplex' three strands (match_triplex, find_motif.c:1183-1232: first
// triple of a strand's first base, at most tq_mplim further triples may fail): a start u of its first
// strand is feasible if the strand can stand there and a second strand can END v - u in [r1_lo, r1_hi]
// later with a third one STARTING w - v in [r2_lo, r2_hi] after that; an end position e of the 4-plex'
// group is worth queueing only if some feasible u lies u - e in [f_lo, f_hi] behind it.
struct rmd_q1filter_t {
	int8_t	on, m2, m3, first5, nmin, badmax;
	int8_t	t_on, tm1, tm2, tm3, tfirst5, tnmin, tbad, pad_;
	int16_t	a_lo, a_hi, b_hi, b_lo;		// (a_hi / b_lo < 0: unbounded)
	int16_t	f_lo, f_hi, r1_lo, r1_hi, r2_lo, r2_hi;
};

struct rmd_program_t {
	int32_t	n_elems, n_searches;
	int32_t	dminlen, w_winsize;	// min( dmaxlen, windowsize )
	int32_t	strict_helices;
	int32_t	split_s;		// general path: alternatives of this level are handed over as continuations
					// (rm_scan_core.h, rmd_gen_resume): the last of the helices that head the
					// search list when levels follow it; -1: none
	int32_t	step_budget;		// general path: loop iterations a step may take before it pauses (>= 4)
	int32_t	need_init;		// some helix is improper: element state must start UNDEF
	int32_t	lean_ok;		// every level is ss or a proper helix: 8-byte-per-level search
	int32_t	ord_ok, ord_bits;	// lean path: order words are numbers of the walk's choices (rmd_elem_t::ord_stride), of so many bits
	int32_t	has_lctx, has_rctx;
	int32_t	n_sites, n_efn;
	int32_t	efn_usestdbp, efn_stdbp;
	int32_t	wide;			// some helix may be longer than 63 base pairs: the general instance with two-word sets of lengths
	int32_t	efn_big;		// some efn() / efn2() call spans more than 15 helices: the energy kernel's instance for that
	int32_t	hit_stride;
	int32_t	lmargin, rmargin;	// bases needed before szero / after the window
	// best-literal pre-filter (the role of the reference's -O, compile.c:3315): regex lit_re
	// must occur with its first base at an offset in [lit_lo, lit_hi] from the start position
	int32_t	lit_re, lit_lo, lit_hi;
	int32_t	lit_elo, lit_ehi;	// ... and at a distance in [lit_elo, lit_ehi] before the end of the first
					// search element's group (lit_ehi < 0: not known)
	// The three pools at the end are reached through byte offsets from the start of the
	// program (rmd_pairsets() ...), so that the image a workgroup copies into LDS holds only
	// what the descriptor uses: the members up to elems[ n_elems ], then the used regexes,
	// rules and pair sets back to back (rmd_make_image()).  In this full struct the offsets
	// point at the arrays below.
	// pair tables that get bit rows over the tile in the kernel (RowEnds, rm_scan_hip.hip): those
	// of the Watson-Crick helices and 4-plex outer helices that head a search level, first element's first
	int32_t	n_rowsets;
	int8_t	rowset_ps[ 4 ];
	int32_t	n_regexes, n_rules, n_pairsets, n_pks, n_tups, n_regexes2;
	int32_t	off_regexes, off_rules, off_pairsets, off_pks, off_tups, off_regexes2;
	int32_t	off_sites, off_efn;
	rmd_q1filter_t	q1f;
	rmd_chain_t	chain;		// (sites[ n_sites ], efn_sites[ n_efn ]: pools like the others)
	int32_t	image_bytes;
	// general path, records in LDS (rm_scan_hip.hip LdsGRecs): dword offset of level k's record; a
	// level with a single alternative (rmd_elem_t::back_s) keeps its window only -- one dword,
	// marked by the sign bit -- the others three
	int16_t	rec_off[ RMD_MAX_ELEMS ];
	int32_t	n_rec_dwords;
	int8_t	searches[ RMD_MAX_ELEMS ];
	rmd_elem_t	lctx, rctx;
	rmd_elem_t	elems[ RMD_MAX_ELEMS ];
	rmd_regex_t	regexes[ RMD_MAX_RE ];
	rmd_regex2_t	regexes2[ RMD_MAX_RE2 ];
	rmd_rule_t	rules[ RMD_MAX_RULES ];
	rmd_pairset_t	pairsets[ RMD_MAX_PS ];
	rmd_pk_t	pks[ RMD_MAX_PK ];
	rmd_tup_t	tups[ RMD_MAX_TUP ];
	rmd_site_t	sites[ RMD_MAX_SITES ];
	rma_efn_site_t	efn_sites[ RMD_MAX_EFN ];
};

#ifndef RMD_HD
#define RMD_HD	inline
#endif
RMD_HD const rmd_regex_t *rmd_regexes( const rmd_program_t *P )
{
	return reinterpret_cast<const rmd_regex_t *>( reinterpret_cast<const char *>( P ) + P->off_regexes );
}
RMD_HD const rmd_regex2_t *rmd_regexes2( const rmd_program_t *P )
{
	return reinterpret_cast<const rmd_regex2_t *>( reinterpret_cast<const char *>( P ) + P->off_regexes2 );
}
RMD_HD const rmd_rule_t *rmd_rules( const rmd_program_t *P )
{
	return reinterpret_cast<const rmd_rule_t *>( reinterpret_cast<const char *>( P ) + P->off_rules );
}
RMD_HD const rmd_pairset_t *rmd_pairsets( const rmd_program_t *P )
{
	return reinterpret_cast<const rmd_pairset_t *>( reinterpret_cast<const char *>( P ) + P->off_pairsets );
}

RMD_HD const rmd_pk_t *rmd_pks( const rmd_program_t *P )
{
	return reinterpret_cast<const rmd_pk_t *>( reinterpret_cast<const char *>( P ) + P->off_pks );
}

RMD_HD const rmd_site_t *rmd_sites( const rmd_program_t *P )
{
	return reinterpret_cast<const rmd_site_t *>( reinterpret_cast<const char *>( P ) + P->off_sites );
}

RMD_HD const rma_efn_site_t *rmd_efn_sites( const rmd_program_t *P )
{
	return reinterpret_cast<const rma_efn_site_t *>( reinterpret_cast<const char *>( P ) + P->off_efn );
}

RMD_HD const rmd_tup_t *rmd_tups( const rmd_program_t *P )
{
	return reinterpret_cast<const rmd_tup_t *>( reinterpret_cast<const char *>( P ) + P->off_tups );
}

// Build the device form; returns 0 or -1 with a message (descriptor outside
// the device limits).  Implemented in rm_dev_program.cpp (host).
int	rmd_build( const rma_program_t *p, rmd_program_t *out, char *err, size_t errlen );
// The compact image of a built program (see rmd_program_t): image_bytes of it, 16-byte
// multiple; img must hold sizeof( rmd_program_t ) bytes.
size_t	rmd_make_image( const rmd_program_t *full, void *img );

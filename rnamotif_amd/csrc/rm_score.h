// rm_score.h -- the score section: code generator and stack VM.
//
// Host side consumer of the scan boundary: for every candidate the scanner
// reports, the element match state is restored into the Descriptor and the
// MAIN program decides ACCEPT / REJECT / HOLD, exactly as RM_score() does when
// find_ss() reaches the end of the search list
// (/root/reference/src/find_motif.c:373-392, /root/reference/src/score.c).
#pragma once
#include "rm_host.h"

namespace rma {

// op codes, score.c:84-122
enum {
	OP_HALT, OP_NOOP, OP_ACPT, OP_HOLD, OP_RJCT, OP_RLSE, OP_MRK, OP_CLS, OP_FCL, OP_SCL,
	OP_STRF, OP_LDA, OP_LOD, OP_LDC, OP_STO, OP_AND, OP_IOR, OP_NOT, OP_MAT, OP_INS,
	OP_GTR, OP_GEQ, OP_EQU, OP_NEQ, OP_LEQ, OP_LES, OP_ADD, OP_SUB, OP_MUL, OP_DIV,
	OP_MOD, OP_NEG, OP_I_PP, OP_PP_I, OP_I_MM, OP_MM_I, OP_FJP, OP_JMP, N_OP
};
// builtins, score.c:165-178
enum {
	SC_STRID, SC_BITS, SC_EFN, SC_EFN2, SC_LENGTH, SC_LOC, SC_MISMATCHES, SC_MISMATCHES_1,
	SC_MISMATCHES_2, SC_MISPAIRS, SC_PAIRED, SC_SPRINTF, SC_SUBSTR, N_SC
};

// A call of efn() whose element/position arguments are compile time constants;
// the scanner evaluates these per candidate (rma_efn_site_t).
struct EfnCall {
	Node	*call;			// the SYM_CALL node
	rma_efn_site_t	site;		// resolved by linkscore()
};

class ScoreVM {
public:
	explicit ScoreVM( Descriptor &d );

	// code generation, called by the parser in grammar action order
	void	action( Node *np );		// RM_action :299
	void	endaction();			// RM_endaction :318
	void	if_( Node *np );		// RM_if :327
	void	else_();			// RM_else :340
	void	endelse();			// RM_endelse :349
	void	endif();			// RM_endif :356
	void	forinit( Node *np );		// RM_forinit :363
	void	fortest( Node *np );		// RM_fortest :374
	void	forincr( Node *np );		// RM_forincr :385
	void	endfor();			// RM_endfor :391
	void	while_( Node *np );		// RM_while :405
	void	endwhile();			// RM_endwhile :419
	void	brk( Node *np );		// RM_break :430
	void	cont( Node *np );		// RM_continue :449
	void	accept();			// RM_accept :467
	void	reject();			// RM_reject :479
	void	hold( Node *np );		// RM_hold :473
	void	release( Node *np );		// RM_release :485
	void	mark();				// RM_mark :491
	void	clear();			// RM_clear :497
	void	expr( int lval, Node *np );	// RM_expr :503
	void	linkscore();			// RM_linkscore :510
	void	setprog( int p );		// RM_setprog :590

	// execution: RM_score :608.  efn_vals points at the per-candidate energies
	// the scanner delivered for efn_calls() (1/100 kcal/mol), or nullptr.
	int	run( int comp, int slen, const char *sbuf, Ident **h_id, const int32_t *efn_vals );

	// True when no candidate's run of MAIN can see what another candidate's run left behind, so that
	// candidates may be replayed on several VMs at once (each on a descriptor compiled for it) and
	// their output put together in order: no HOLD / RELEASE; every variable MAIN writes is definitely
	// assigned before MAIN -- or the printer, for SCORE -- reads it, on every path; a variable whose type
	// the first assignment would latch (undefined after BEGIN) gets the same type from every assignment;
	// END reads nothing MAIN writes.  Conservative: anything the analysis cannot follow says no (*why).
	// To be called after linkscore() and after BEGIN has run.
	bool	hit_independent( std::string *why ) const;

	const std::vector<EfnCall>	&efn_calls() const { return efn_calls_; }
	bool	has_main() const { return !progs_[ P_MAIN ].empty(); }
	void	dump( FILE *fp );		// RM_dumpscore :563
	FILE	*out = stdout;			// RELEASE writes here

	std::vector<Strel *>	xdescr;		// rm_xdescr

private:
	Descriptor	&d_;
	std::vector<Inst>	progs_[ N_PROG ];
	std::vector<int>	labtabs_[ N_PROG ];
	int	nextlabs_[ N_PROG ] = { 0, 0, 0 };
	std::vector<int>	ifstks_[ N_PROG ];
	std::vector<int>	loopstks_[ N_PROG ];
	std::vector<Node *>	loopincrs_[ N_PROG ];
	int	c_prog_ = P_MAIN;
	int	actlab_ = 0;
	int	v_lab_ = 0;			// the static v_lab the reference reuses
	std::vector<EfnCall>	efn_calls_;
	Ident	*slen_id_ = nullptr;

	// run time
	std::vector<Value>	mem_;
	int	pc_ = 0, mp_ = -1, sp_ = -1, esp_ = -1;
	int	estk_[ 20 ];
	int	sc_comp_ = 0, sc_slen_ = 0;
	const char	*sc_sbuf_ = nullptr;
	const int32_t	*efn_vals_ = nullptr;

	std::vector<Inst>	&prog() { return progs_[ c_prog_ ]; }
	int	&label( int l );
	int	newlabs( int n );
	void	addinst( Node *np, int op, const Value *vp );
	void	addlab( Node *np, int op, int lab );
	void	fixexpr( Node *np );
	void	genexpr( int lval, Node *np );
	void	addnode( int lval, Node *np, int l_andor );
	void	fix_kw_stref( Node *np );
	void	fix_ix_stref( Node *np );
	void	fix_stref_common( Node *np, int sel, Node *n_id, Node *n_pos, Node *n_len );
	void	fix_call( Node *np );
	void	note_efn_call( Node *call, Node *a1, Node *a2, bool efn2 );

	void	do_scl( const Inst &ip );
	int	strid( int stype, Value *v_id );
	int	paired( Strel *stp, int pos, int len );
	float	do_bits( const Inst &ip );
	float	do_efn( const Inst &ip );
	void	do_sprintf( const Inst &ip, std::string &outbuf );
	void	do_strf( const Inst &ip );
	void	do_compare( const Inst &ip );
	void	do_arith( const Inst &ip );
	void	do_incr( const Inst &ip );
	Strel	*xd( const Inst &ip, int idx, const char *who );
};

// print_match(), find_motif.c:1826-1959: formats one accepted candidate.
class HitPrinter {
public:
	HitPrinter( Descriptor &d, FILE *out ) : d_( d ), out_( out ) {}
	// the candidate's element state must already be in d.descr / lctx / rctx
	void	print( const char *sid, const char *sdef, int comp, int slen, const char *sbuf, Ident *h_id );
	// the three "#RM" lines that precede the first hit
	void	header( FILE *fp ) const;
	// several printers share one output (parallel replay): whoever puts the pieces together writes the
	// header once, the printers never do
	void	no_header() { first_ = false; }
	bool	header_pending() const { return first_; }
	void	set_out( FILE *fp ) { out_ = fp; }
private:
	Descriptor	&d_;
	FILE	*out_;
	bool	first_ = true;
};

}	// namespace rma

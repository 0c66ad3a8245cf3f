// rm_host.h -- host-side descriptor front end: types shared by the lexer,
// parser, descriptor compiler, score compiler/VM and the program flattener.
//
// This is the producer side of the scan boundary.  It restates, with its own
// data structures, what the reference does in src/rmlex.l, src/rmgrm.y,
// src/compile.c, src/node.c, src/preprocessor.c, src/getargs.c and src/score.c
// (all under /root/reference); each function cites the part it follows.
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "rnamotif_amd_program.h"
#include "rm_regex.h"

namespace rma {

constexpr int UNDEF = -1;
constexpr int UNBOUNDED = 0x7fffffff;

struct Error : std::runtime_error {
	using std::runtime_error::runtime_error;
};
[[noreturn]] void fail( const char *fmt, ... ) __attribute__(( format( printf, 1, 2 ) ));

// value types, rnamot.h:49-57
enum { T_UNDEF, T_INT, T_FLOAT, T_STRING, T_PAIRSET, T_POS, T_IDENT, T_HIT };
// symbol scopes, rnamot.h:65-68
enum { S_GLOBAL = 1, S_STREL, S_SITE };
// lexer/parser context, rnamot.h:71-75
enum { CTX_START, CTX_PARMS, CTX_DESCR, CTX_SITES, CTX_SCORE };

// tokens / node symbols (rmgrm.y:29-111; numeric values are this build's own)
enum Sym {
	SYM_EOF = 0,
	SYM_PARMS, SYM_DESCR, SYM_SITES, SYM_SCORE,
	SYM_SE, SYM_CTX, SYM_SS, SYM_H5, SYM_H3, SYM_P5, SYM_P3,
	SYM_T1, SYM_T2, SYM_T3, SYM_Q1, SYM_Q2, SYM_Q3, SYM_Q4,
	SYM_ACCEPT, SYM_BEGIN, SYM_BREAK, SYM_CONTINUE, SYM_ELSE, SYM_END, SYM_FOR,
	SYM_HOLD, SYM_IF, SYM_IN, SYM_REJECT, SYM_RELEASE, SYM_WHILE,
	SYM_IDENT, SYM_INT, SYM_FLOAT, SYM_STRING, SYM_PAIRSET,
	SYM_AND, SYM_ASSIGN, SYM_DOLLAR, SYM_DONT_MATCH, SYM_EQUAL, SYM_GREATER,
	SYM_GREATER_EQUAL, SYM_LESS, SYM_LESS_EQUAL, SYM_MATCH, SYM_MINUS,
	SYM_MINUS_ASSIGN, SYM_MINUS_MINUS, SYM_NEGATE, SYM_NOT, SYM_NOT_EQUAL, SYM_OR,
	SYM_PERCENT, SYM_PERCENT_ASSIGN, SYM_PLUS, SYM_PLUS_ASSIGN, SYM_PLUS_PLUS,
	SYM_STAR, SYM_STAR_ASSIGN, SYM_SLASH, SYM_SLASH_ASSIGN,
	SYM_LPAREN, SYM_RPAREN, SYM_LBRACK, SYM_RBRACK, SYM_LCURLY, SYM_RCURLY,
	SYM_COLON, SYM_COMMA, SYM_SEMICOLON,
	SYM_CALL, SYM_LIST, SYM_KW_STREF, SYM_IX_STREF,
	SYM_ERROR
};

struct Pair { int n_bases = 0; char bases[ 4 ] = {}; };	// PAIR_T

struct PairSet {					// PAIRSET_T
	std::vector<Pair>	pairs;
	rma_pairset_t	mat{};				// ps_mat[0..1] as bit sets
};

struct Addr { int l2r = 0, offset = 0; };		// ADDR_T
struct Strel;
struct Pos {						// POS_T
	int	type = 0, lineno = 0;
	const char	*tag = nullptr;
	Strel	*descr = nullptr;
	Addr	addr;
};

struct Ident;
struct Hit { char *def = nullptr, *match = nullptr; };	// HIT_T

struct Value {						// VALUE_T
	int	type = T_UNDEF;
	union {
		int	ival;
		double	dval;
		void	*pval;
	};
	Value() : pval( nullptr ) {}
};

struct Ident {						// IDENT_T
	std::string	name;
	int	type = T_UNDEF, scope = 0, reinit = 0;
	Value	val;
};

struct Node {						// NODE_T
	int	sym = 0;
	int	lineno = 0;
	const char	*filename = nullptr;
	Value	val;
	Node	*left = nullptr, *right = nullptr;
};

enum { SA_PROPER, SA_ENDS, SA_STRICT, SA_N_ATTR };

struct Strel {						// STREL_T
	int	checked = 0;
	int	type = 0;
	signed char	attr[ SA_N_ATTR ] = { 0, 0, 0 };
	int	index = UNDEF, lineno = 0, searchno = UNDEF;
	int	matchoff = UNDEF, matchlen = UNDEF, n_mismatches = UNDEF, n_mispairs = UNDEF;
	const char	*tag = nullptr;
	Strel	*next = nullptr, *prev = nullptr, *inner = nullptr, *outer = nullptr;
	std::vector<Strel *>	mates, scopes;
	int	scope = UNDEF;
	int	minlen = UNDEF, maxlen = UNDEF, minglen = UNDEF, maxglen = UNDEF;
	int	minilen = UNDEF, maxilen = UNDEF;
	const char	*seq = nullptr;
	std::shared_ptr<ReProg>	re;
	int	mismatch = 0;
	double	matchfrac = 1.0;
	int	mispair = UNDEF;
	double	pairfrac = UNDEF;
	PairSet	*pairset = nullptr;
};

struct Site {						// SITE_T
	std::vector<Pos>	pos;
	PairSet	*pairset = nullptr;
};

struct Args {						// ARGS_T, getargs.c
	bool	copt = false, dopt = false, hopt = false, popt = false, sopt = false, vopt = false;
	bool	show_context = false, strict_helices = false;
	int	maxslen = 0;
	float	o_emin = 2.5f;
	std::vector<std::string>	incdirs;
	std::string	dfname, xdfname, cldefs, dbfmt;
	std::string	argv0 = "rnamotif";
	bool	have_dfname = false, have_xdfname = false;
	std::vector<std::string>	dbfnames;
};

// one instruction of the score programs, score.c:196-201
struct Inst {
	const char	*filename;
	int	lineno;
	int	op;
	Value	val;
};
enum { P_BEGIN, P_MAIN, P_END, N_PROG };
enum { SA_REJECT, SA_HOLD, SA_ACCEPT };

class ScoreVM;

// The compiled descriptor: everything the reference keeps in compile.c's
// globals (compile.c:14-106).
struct Descriptor {
	Args	args;
	std::map<std::string, Ident *>	globals;	// rm_global_ids
	std::vector<Ident *>	locals;			// local_ids[]
	std::vector<Strel>	descr;			// rm_descr[] (reserved to RMA_MAX_ELEMS)
	int	dminlen = 0, dmaxlen = 0;
	Strel	*lctx = nullptr, *rctx = nullptr;
	bool	lctx_explicit = false, rctx_explicit = false;
	std::vector<Site>	sites;
	std::vector<Strel *>	searches;		// rm_searches[] as s_descr
	PairSet	*efnstdbp = nullptr;
	Value	*nval = nullptr, *sval = nullptr, *cval = nullptr, *pval = nullptr, *lval = nullptr;
	int	context = CTX_PARMS;
	const char	*wdfname = "";
	int	lineno = 0;
	bool	error = false;
	std::unique_ptr<ScoreVM>	score;
	std::string	stderr_text;			// non-fatal diagnostics
	std::string	expanded;			// the preprocessed text it was parsed from (compile_descriptor)

	Descriptor();
	~Descriptor();

	// symbol table, compile.c:1685-1795
	Ident	*enter_id( const char *name, int type, int scope, int reinit, const Value *vp );
	Ident	*find_id( const char *name );
	int	int_global( const char *name, int dflt );	// value of an int global (show_progress ...)

	// parser actions, compile.c
	void	parm_add( Node *expr );			// PARM_add :396
	Node	*pr_close( std::vector<const char *> &curpair );	// PR_close :421
	void	se_open( int stype );			// SE_open :501
	void	se_addval( Node *expr );		// SE_addval :686
	void	se_close();				// SE_close :693
	void	pos_open( int ptype );			// POS_open :2728
	void	pos_close();				// POS_close :2767
	void	si_close( Node *expr );			// SI_close :2786
	char	*str2seq( const char *str );		// RM_str2seq :2682
	void	link();					// SE_link :776

	// flatten for the scan path (this build's boundary)
	void	to_program( rma_program_t *out );

	void	note_error( const char *fmt, ... ) __attribute__(( format( printf, 2, 3 ) ));

private:
	Strel	*open_stp = nullptr;
	PairSet	*open_pairset = nullptr;
	std::vector<Pos>	cur_pos;		// rm_pos[]
	Pos	*posp = nullptr;
	std::vector<Value>	valstk;

	void	se_init( Strel *stp, int stype );
	int	ends2attr( const char *str );
	int	strict2attr( int sval );
	void	eval( Node *expr, bool d_ok );
	int	loadidval( Value *vp );
	void	storeexprval( Ident *ip, Value *vp );
	PairSet	*pair_check( PairSet *ps );
	PairSet	*pair_copy( const PairSet *ps );
	PairSet	*pair_add( const PairSet *a, const PairSet *b );
	PairSet	*pair_sub( const PairSet *a, const PairSet *b );
	bool	pair_equal( const PairSet *a, const PairSet *b );
	void	mk_mats( PairSet *ps );
	Pos	*pos_cvt( Value *vp );
	Pos	*pos_sub( Pos *l, Pos *r );
	void	chk_context();
	void	link_tags();
	void	chk_tagorder( int n_tags, Strel *tags[] );
	void	mk_links( int n_tags, Strel *tags[] );
	bool	chk_proper_nesting( Strel *a, Strel *b );
	void	find_pknots( Strel *stp );
	bool	chk_strel_parms();
	bool	chk_1_strel_parms( Strel *stp );
	bool	chk_len_seq( int n, Strel *egroup[] );
	bool	chk_site( Site &s );
	Strel	*set_scopes( int fd, int ld, std::vector<Strel *> &stk );
	void	find_gi_len( int fd, int *tmin, int *tmax );
	void	find_search_order( int fd );
	friend class ScoreVM;
	friend class Parser;
};

// base letter -> code, compile.c:180-187
extern int	b2bc[ 256 ];
void	init_b2bc();

Node	*mk_node( Descriptor &d, int sym, const Value *vp, Node *left, Node *right );	// RM_node, node.c

// getargs.c:11-246; throws Error with the usage text on bad arguments
Args	parse_args( int argc, char **argv );
extern const char	*USAGE_FMT;
extern const char	*VERSION_STR;

// preprocessor.c:38-95; returns the expanded descriptor text
std::string	preprocess( Descriptor &d );

// lexer + recursive descent parser for the grammar of rmgrm.y; returns false
// on a syntax error (the reference's "yyerror: syntax error")
bool	parse_descriptor( Descriptor &d, const std::string &text );

// run the whole front end the way main() does (rnamot.c:49-98)
// expanded: the preprocessed text of a descriptor compiled before (Descriptor::expanded) -- the descriptor files,
// which may be readable only once (a pipe), are not opened again and -xdfname's file is not rewritten
std::unique_ptr<Descriptor>	compile_descriptor( const Args &args, const std::string *expanded = nullptr );

const char	*strel_name( int type );		// RM_strel_name, dump.c:600
// -s / -d / -h listings, RM_dump dump.c:34 (rm_dump.cpp)
void	dump_descriptor( Descriptor &d, FILE *fp, int d_parms, int d_descr, int d_sites, int d_hierarchy );
// a descriptor holding only the built-in symbols (RM_init, compile.c:157-394), for -s
std::unique_ptr<Descriptor>	init_only( const Args &args );

}	// namespace rma

// rm_parse.cpp -- command line, preprocessor, lexer and recursive descent
// parser of the descriptor language.
//
// The reference generates its scanner and parser with lex/yacc from
// /root/reference/src/rmlex.l and /root/reference/src/rmgrm.y; neither tool
// exists in this build's toolchain, so the same language is recognised by a
// hand written lexer (token rules rmlex.l:40-190) and a recursive descent
// parser that fires the semantic actions in the order the grammar's embedded
// actions do (rmgrm.y:186-545).  Also here: getargs.c and preprocessor.c.
#include "rm_host.h"
#include "rm_score.h"
#include <cctype>
#include <cstdlib>
#include <cstring>

namespace rma {

const char	*USAGE_FMT =
"usage: %s [ options ] descr [ fmt ] [ data ]\n\n"
"options:\n"
"\t-c\t\t\tCompile only, no search\n"
"\t-d\t\t\tDump internal data structures\n"
"\t-h\t\t\tDump the structure hierarchy\n"
"\t-N size\t\t\tSize of longest input. (default=30000000)\n"
"\t-On\t\t\tMin #chars, best seq= for opt. (default=2.5)\n"
"\t-p\t\t\tDump the score code\n"
"\t-s\t\t\tShow builtin variables\n"
"\t-v\t\t\tPrint Version Infomation\n"
"\t-context\t\tPrint solution context\n"
"\t-sh\t\t\tStrict helices: bases surrounding a helix\n"
"\t\t\t\tmust not be able to extend that helix\n"
"\t-Dvar=expr\t\tSet the value of var to expr\n"
"\t-Idir\t\t\tAdd include source directory, dir\n"
"\t-xdfname file-name\tPreprocessor output file\n"
"\t-pre cmd\t\tmrnamotif only: run cmd db | rnamotif\n"
"\t-post cmd\t\tmrnamotif only: run rnamotif | cmd\n"
"\t-help\t\t\tPrint this message\n"
"\n"
"descr:\tUse one:\n"
"\t-descr descr-file\tMay have includes; use cmd-line defs\n"
"\t-xdescr xdescr-file\tMay not have includes; ignore cmd-line defs\n"
"\n"
"fmt:\t(Optional) Use one:\n"
"\t-fmt fastn\t\tfastn (default)\n"
"\t-fmt pir\t\tpir\n"
"\t-fmt gb\t\t\tGB flatfile\n"
"\n"
"data:\t(Optional) Use one:\n"
"\tfile1 ...\t\tSerial version; no files search stdin (default)\n"
"\t-fmap file-map db1 ...\tmrnamotif only; no dbs search whole map\n";

// ---------------------------------------------------------------- getargs.c:11-246
Args parse_args( int argc, char **argv )
{
	Args	a;
	if( argc > 0 && argv[ 0 ] != nullptr )
		a.argv0 = argv[ 0 ];
	auto usage = [&]() {
		char	buf[ 4096 ];
		snprintf( buf, sizeof( buf ), USAGE_FMT, argv[ 0 ] );
		throw Error( buf );
	};
	for( int ac = 1; ac < argc; ac++ ){
		const char	*s = argv[ ac ];
		if( !strcmp( s, "-c" ) ) a.copt = true;
		else if( !strcmp( s, "-d" ) ) a.dopt = true;
		else if( !strcmp( s, "-h" ) ) a.hopt = true;
		else if( !strcmp( s, "-N" ) ){
			if( ac == argc - 1 || a.maxslen != 0 )
				usage();
			a.maxslen = atoi( argv[ ++ac ] ) + 1;
		}else if( !strncmp( s, "-O", 2 ) ){
			if( s[ 2 ] == '\0' )
				usage();
			a.o_emin = float( atof( s + 2 ) );
			if( a.o_emin < 0.25 )
				a.o_emin = 0.0;
		}else if( !strcmp( s, "-p" ) ) a.popt = true;
		else if( !strcmp( s, "-s" ) ) a.sopt = true;
		else if( !strcmp( s, "-v" ) ) a.vopt = true;
		else if( !strcmp( s, "-context" ) ) a.show_context = true;
		else if( !strcmp( s, "-sh" ) ) a.strict_helices = true;
		else if( !strcmp( s, "-descr" ) ){
			if( ac == argc - 1 || a.have_dfname )
				usage();
			a.dfname = argv[ ++ac ];
			a.have_dfname = true;
		}else if( !strcmp( s, "-xdescr" ) || !strcmp( s, "-xdfname" ) ){
			if( ac == argc - 1 || a.have_xdfname )
				usage();
			a.xdfname = argv[ ++ac ];
			a.have_xdfname = true;
		}else if( !strncmp( s, "-D", 2 ) ){
			a.cldefs += s + 2;
			a.cldefs += "; ";
		}else if( !strncmp( s, "-I", 2 ) ){
			a.incdirs.push_back( s + 2 );
		}else if( !strcmp( s, "-fmt" ) ){
			if( ac == argc - 1 )
				usage();
			a.dbfmt = argv[ ++ac ];
			if( a.dbfmt != "fastn" && a.dbfmt != "pir" && a.dbfmt != "gb" )
				usage();
		}else if( *s == '-' ){
			usage();	// includes -pre/-post/-fmap (mrnamotif only) and -help
		}else
			a.dbfnames.push_back( s );
	}
	if( a.maxslen == 0 )
		a.maxslen = 30000000 + 1;
	if( !a.cldefs.empty() )
		a.cldefs[ a.cldefs.size() - 1 ] = '\n';	// "; " -> ";\n", getargs.c:240
	return a;
}

// ---------------------------------------------------------------- preprocessor.c
namespace {

// isdescr(), preprocessor.c:248-285: position of a bare "descr" keyword
const char *isdescr( const char *line )
{
	const char	*dp = strstr( line, "descr" );
	if( dp == nullptr )
		return nullptr;
	const char	*edp = dp + 5;
	if( !isspace( ( unsigned char )*edp ) && *edp != '#' )
		return nullptr;
	if( dp == line )
		return line;
	if( !isspace( ( unsigned char )dp[ -1 ] ) && dp[ -1 ] != ';' )
		return nullptr;
	const char	*qp;
	if( ( qp = strchr( line, '\'' ) ) != nullptr && qp < dp ){
		bool	instr = true;
		for( const char *q = qp + 1; q < dp; q++ ){
			if( *q == '\'' )
				instr = !instr;
			else if( *q == '\\' )
				q++;
		}
		return !instr ? dp : nullptr;
	}
	if( ( qp = strchr( line, '"' ) ) != nullptr && qp < dp ){
		bool	instr = true;
		for( const char *q = qp + 1; q < dp; q++ ){
			if( *q == '"' )
				instr = !instr;
			else if( *q == '\\' )
				q++;
		}
		return instr ? dp : nullptr;	// (sic) preprocessor.c:281
	}
	return dp;
}

struct SrcFile { FILE *fp; std::string name; int lineno; };

}	// namespace

std::string preprocess( Descriptor &d )
{
	std::vector<SrcFile>	stk;
	std::string	out;
	auto push = [&]( const std::string &fname, bool must ) -> bool {
		if( stk.size() >= 10 )
			fail( "%s:%d fstk overflow: '%s'.", d.wdfname, d.lineno, fname.c_str() );
		FILE	*fp = fopen( fname.c_str(), "r" );
		if( fp == nullptr ){
			if( must )
				fail( "%s:%d can't read '%s'.", d.wdfname, d.lineno, fname.c_str() );
			return false;
		}
		if( !stk.empty() )
			stk.back().lineno = d.lineno;
		stk.push_back( SrcFile{ fp, fname, 0 } );
		d.lineno = 0;
		return true;
	};
	auto putline = [&]( const char *fname, int lineno, const char *line ){
		char	hdr[ 1200 ];
		snprintf( hdr, sizeof( hdr ), "\n# line %d '%s'\n", lineno, fname );
		out += hdr;
		out += line;
	};
	{
		FILE	*fp = fopen( d.args.dfname.c_str(), "r" );
		if( fp == nullptr )
			fail( "RM_preprocessor: can't read descr file '%s'.", d.args.dfname.c_str() );
		stk.push_back( SrcFile{ fp, d.args.dfname, 0 } );
		d.lineno = 0;
	}
	char	line[ 1024 ];
	for( ; ; ){
		bool	got = false;
		while( !stk.empty() ){
			if( fgets( line, sizeof( line ), stk.back().fp ) ){
				got = true;
				break;
			}
			fclose( stk.back().fp );
			stk.pop_back();
			if( !stk.empty() )
				d.lineno = stk.back().lineno;
		}
		if( !got )
			break;
		d.lineno++;
		// file names live as long as the descriptor (node/strel file names point at them)
		const char	*fname = strdup( stk.back().name.c_str() );
		d.wdfname = fname;
		if( *line == '#' ){
			const char	*lp = line + 1;
			while( isspace( ( unsigned char )*lp ) )
				lp++;
			if( !strncmp( lp, "include", 7 ) ){	// include(), preprocessor.c:151-219
				const char	*sp = line + 1;
				while( isspace( ( unsigned char )*sp ) ) sp++;
				while( *sp && !isspace( ( unsigned char )*sp ) ) sp++;
				if( *sp == '\0' )
					fail( "%s:%d no filename.", fname, d.lineno );
				while( isspace( ( unsigned char )*sp ) ) sp++;
				int	c = *sp == '"' ? '"' : *sp == '<' ? '>' : *sp == '\'' ? '\'' : 0;
				if( c == 0 )
					fail( "%s:%d bad include filename '%s'.", fname, d.lineno, sp );
				sp++;
				const char	*ep = strchr( sp, c );
				if( ep == nullptr )
					fail( "%s:%d bad include filename '%s'.", fname, d.lineno, sp );
				std::string	inc( sp, ep - sp );
				if( c == '"' )
					inc = d.str2seq( inc.c_str() );	// (sic) preprocessor.c:198
				bool	ok = false;
				if( d.args.incdirs.empty() )
					ok = push( inc, true );
				else for( const std::string &dir : d.args.incdirs ){
					if( ( ok = push( dir + "/" + inc, false ) ) )
						break;
				}
				if( !ok )
					fail( "%s:%d can't find include file '%s'.", fname, d.lineno, inc.c_str() );
			}
			// other # lines are comments and are dropped
		}else{
			const char	*dp = isdescr( line );
			if( dp != nullptr && !d.args.cldefs.empty() ){
				if( dp > line ){
					std::string	head( line, dp - line );
					putline( fname, d.lineno, head.c_str() );
				}
				putline( "cmd line defs", 1, d.args.cldefs.c_str() );
				putline( fname, d.lineno, dp );
			}else
				putline( fname, d.lineno, line );
		}
	}
	return out;
}

// ---------------------------------------------------------------- lexer
namespace {

struct Token {
	int	sym = SYM_EOF;
	Value	val;
	int	lineno = 0;
	const char	*fname = nullptr;
};

class Lexer {
public:
	Lexer( Descriptor &d, const std::string &text ) : d_( d ), s_( text ) {}

	Token next()
	{
		Token	t;
		for( ; ; ){
			if( p_ >= s_.size() ){
				t.sym = SYM_EOF;
				break;
			}
			char	c = s_[ p_ ];
			bool	bol = p_ == 0 || s_[ p_ - 1 ] == '\n';
			if( c == '#' ){
				size_t	e = s_.find( '\n', p_ );
				if( e == std::string::npos )
					e = s_.size();
				if( bol && !s_.compare( p_, 6, "# line" ) )
					setfileinfo( s_.substr( p_, e - p_ ) );
				p_ = e;
				continue;
			}
			if( c == '\n' || c == '\r' || c == ' ' || c == '\t' || c == '\f' ){
				p_++;
				continue;
			}
			t = scan();
			break;
		}
		t.lineno = d_.lineno;
		t.fname = d_.wdfname;
		return t;
	}

private:
	Descriptor	&d_;
	const std::string	&s_;
	size_t	p_ = 0;
	std::map<std::string, char *>	fnames_;

	void setfileinfo( const std::string &data )	// rmlex.l:200-232
	{
		const char	*dp = data.c_str() + 6;
		if( !isspace( ( unsigned char )*dp ) )
			return;
		dp += strspn( dp, " \t" );
		if( !isdigit( ( unsigned char )*dp ) )
			return;
		int	lnum = 0;
		for( ; isdigit( ( unsigned char )*dp ); dp++ )
			lnum = 10 * lnum + *dp - '0';
		if( !isspace( ( unsigned char )*dp ) )
			return;
		while( isspace( ( unsigned char )*dp ) )
			dp++;
		if( *dp != '\'' )
			return;
		dp++;
		const char	*qp = strchr( dp, '\'' );
		if( qp ){
			d_.lineno = lnum;
			std::string	fn( dp, qp - dp );
			auto	it = fnames_.find( fn );
			if( it == fnames_.end() )
				it = fnames_.emplace( fn, strdup( fn.c_str() ) ).first;
			d_.wdfname = it->second;
		}
	}

	Token scan()
	{
		Token	t;
		char	c = s_[ p_ ];
		auto peek = [&]( size_t k ) -> char { return p_ + k < s_.size() ? s_[ p_ + k ] : '\0'; };
		if( isalpha( ( unsigned char )c ) ){
			size_t	e = p_;
			while( e < s_.size() && ( isalnum( ( unsigned char )s_[ e ] ) || s_[ e ] == '_' ) )
				e++;
			std::string	w = s_.substr( p_, e - p_ );
			p_ = e;
			static const struct { const char *w; int sym; } kw[] = {
				{ "parms", SYM_PARMS }, { "descr", SYM_DESCR }, { "sites", SYM_SITES }, { "score", SYM_SCORE },
				{ "se", SYM_SE }, { "ctx", SYM_CTX }, { "ss", SYM_SS }, { "h5", SYM_H5 }, { "h3", SYM_H3 },
				{ "p5", SYM_P5 }, { "p3", SYM_P3 }, { "t1", SYM_T1 }, { "t2", SYM_T2 }, { "t3", SYM_T3 },
				{ "q1", SYM_Q1 }, { "q2", SYM_Q2 }, { "q3", SYM_Q3 }, { "q4", SYM_Q4 },
				{ "ACCEPT", SYM_ACCEPT }, { "BEGIN", SYM_BEGIN }, { "END", SYM_END }, { "HOLD", SYM_HOLD },
				{ "REJECT", SYM_REJECT }, { "RELEASE", SYM_RELEASE }, { "break", SYM_BREAK },
				{ "continue", SYM_CONTINUE }, { "else", SYM_ELSE }, { "for", SYM_FOR }, { "if", SYM_IF },
				{ "in", SYM_IN }, { "while", SYM_WHILE } };
			for( const auto &k : kw ){
				if( w == k.w ){
					t.sym = k.sym;
					return t;
				}
			}
			t.sym = SYM_IDENT;
			t.val.type = T_STRING;
			t.val.pval = strdup( w.c_str() );
			return t;
		}
		if( isdigit( ( unsigned char )c ) || ( c == '.' && isdigit( ( unsigned char )peek( 1 ) ) ) ){
			// [0-9]+ | [0-9]+[eE][+-]?[0-9]+ | ([0-9]+\.[0-9]*|\.[0-9]+)([eE][+-]?[0-9]+)?
			size_t	e = p_;
			bool	isfloat = false;
			while( e < s_.size() && isdigit( ( unsigned char )s_[ e ] ) )
				e++;
			if( e < s_.size() && s_[ e ] == '.' ){
				isfloat = true;
				e++;
				while( e < s_.size() && isdigit( ( unsigned char )s_[ e ] ) )
					e++;
			}
			if( e < s_.size() && ( s_[ e ] == 'e' || s_[ e ] == 'E' ) ){
				size_t	x = e + 1;
				if( x < s_.size() && ( s_[ x ] == '+' || s_[ x ] == '-' ) )
					x++;
				if( x < s_.size() && isdigit( ( unsigned char )s_[ x ] ) ){
					while( x < s_.size() && isdigit( ( unsigned char )s_[ x ] ) )
						x++;
					e = x;
					isfloat = true;
				}
			}
			std::string	w = s_.substr( p_, e - p_ );
			p_ = e;
			if( isfloat ){
				t.sym = SYM_FLOAT;
				t.val.type = T_FLOAT;
				t.val.dval = atof( w.c_str() );
			}else{
				t.sym = SYM_INT;
				t.val.type = T_INT;
				t.val.ival = atoi( w.c_str() );
			}
			return t;
		}
		if( c == '"' || c == '\'' ){	// rmlex.l:108-145
			size_t	e = p_ + 1;
			for( ; ; ){
				while( e < s_.size() && s_[ e ] != c && s_[ e ] != '\n' )
					e++;
				if( e < s_.size() && s_[ e ] == c && s_[ e - 1 ] == '\\' && e - 1 > p_ ){
					e++;	// escaped quote stays in the string, yymore()
					continue;
				}
				break;
			}
			std::string	w = s_.substr( p_ + 1, e - p_ - 1 );
			p_ = e < s_.size() ? e + 1 : e;		// input() eats the closing char
			t.sym = SYM_STRING;
			t.val.type = T_STRING;
			t.val.pval = c == '"' ? d_.str2seq( w.c_str() ) : strdup( w.c_str() );
			return t;
		}
		char	c1 = peek( 1 );
		auto two = [&]( int sym ){ p_ += 2; t.sym = sym; return t; };
		auto one = [&]( int sym ){ p_ += 1; t.sym = sym; return t; };
		switch( c ){
		case '&' : if( c1 == '&' ) return two( SYM_AND ); break;
		case '=' :
			if( c1 == '=' ) return two( SYM_EQUAL );
			if( c1 == '~' ) return two( SYM_MATCH );
			return one( SYM_ASSIGN );
		case '$' : {
			Pos	*pp = new Pos;
			pp->type = SYM_DOLLAR;
			pp->lineno = d_.lineno;
			pp->addr.l2r = 0;
			pp->addr.offset = 0;
			t.val.type = T_POS;
			t.val.pval = pp;
			return one( SYM_DOLLAR );
		}
		case '!' :
			if( c1 == '~' ) return two( SYM_DONT_MATCH );
			if( c1 == '=' ) return two( SYM_NOT_EQUAL );
			return one( SYM_NOT );
		case '>' : if( c1 == '=' ) return two( SYM_GREATER_EQUAL ); return one( SYM_GREATER );
		case '<' : if( c1 == '=' ) return two( SYM_LESS_EQUAL ); return one( SYM_LESS );
		case '-' :
			if( c1 == '=' ) return two( SYM_MINUS_ASSIGN );
			if( c1 == '-' ) return two( SYM_MINUS_MINUS );
			return one( SYM_MINUS );
		case '|' : if( c1 == '|' ) return two( SYM_OR ); break;
		case '%' : if( c1 == '=' ) return two( SYM_PERCENT_ASSIGN ); return one( SYM_PERCENT );
		case '+' :
			if( c1 == '=' ) return two( SYM_PLUS_ASSIGN );
			if( c1 == '+' ) return two( SYM_PLUS_PLUS );
			return one( SYM_PLUS );
		case '*' : if( c1 == '=' ) return two( SYM_STAR_ASSIGN ); return one( SYM_STAR );
		case '/' : if( c1 == '=' ) return two( SYM_SLASH_ASSIGN ); return one( SYM_SLASH );
		case '(' : return one( SYM_LPAREN );
		case ')' : return one( SYM_RPAREN );
		case '[' : return one( SYM_LBRACK );
		case ']' : return one( SYM_RBRACK );
		case '{' : return one( SYM_LCURLY );
		case '}' : return one( SYM_RCURLY );
		case ',' : return one( SYM_COMMA );
		case ':' : return one( SYM_COLON );
		case ';' : return one( SYM_SEMICOLON );
		default : break;
		}
		return one( SYM_ERROR );
	}
};

struct SyntaxError {};

}	// namespace

// ---------------------------------------------------------------- parser
class Parser {
public:
	Parser( Descriptor &d, const std::string &text ) : d_( d ), lex_( d, text ), sc_( *d.score )
	{
		la_[ 0 ] = lex_.next();
		la_[ 1 ] = lex_.next();
	}

	void program()	// rmgrm.y:186-213
	{
		// parm_part: optional 'parms' keyword, then assignments
		if( tok() == SYM_PARMS )
			advance();
		d_.context = CTX_PARMS;
		while( tok() != SYM_DESCR ){
			if( tok() == SYM_EOF )
				throw SyntaxError();
			asgn();
			expect( SYM_SEMICOLON );
		}
		expect( SYM_DESCR );
		d_.context = CTX_DESCR;
		strel();
		while( is_strtype( tok() ) )
			strel();
		if( tok() == SYM_SITES ){
			advance();
			d_.context = CTX_SITES;
			kw_site();
			while( is_strtype( tok() ) )
				kw_site();
		}
		if( tok() == SYM_SCORE ){
			advance();
			d_.context = CTX_SCORE;
			rule();
			while( tok() != SYM_EOF )
				rule();
			sc_.accept();
		}
		if( tok() != SYM_EOF )
			throw SyntaxError();
	}

private:
	Descriptor	&d_;
	Lexer	lex_;
	ScoreVM	&sc_;
	Token	la_[ 2 ];

	int	tok() const { return la_[ 0 ].sym; }
	int	tok2() const { return la_[ 1 ].sym; }
	Token	advance()
	{
		Token	t = la_[ 0 ];
		la_[ 0 ] = la_[ 1 ];
		la_[ 1 ] = lex_.next();
		return t;
	}
	void	expect( int sym )
	{
		if( tok() != sym )
			throw SyntaxError();
		advance();
	}
	static bool is_strtype( int s ) { return s >= SYM_SE && s <= SYM_Q4; }
	static bool is_asgn_op( int s )
	{
		return s == SYM_ASSIGN || s == SYM_MINUS_ASSIGN || s == SYM_PLUS_ASSIGN ||
			s == SYM_PERCENT_ASSIGN || s == SYM_SLASH_ASSIGN || s == SYM_STAR_ASSIGN;
	}
	static bool is_incr_op( int s ) { return s == SYM_PLUS_PLUS || s == SYM_MINUS_MINUS; }
	Node	*node( int sym, const Value *vp, Node *l, Node *r ) { return mk_node( d_, sym, vp, l, r ); }

	// strel in the descr section: strhdr [ '(' a_list ')' ]
	void strel()
	{
		if( !is_strtype( tok() ) )
			throw SyntaxError();
		int	stype = advance().sym;
		d_.se_open( stype );
		if( tok() == SYM_LPAREN ){
			advance();
			a_list();
			expect( SYM_RPAREN );
		}
		d_.se_close();
	}

	// kw_site : kw_pairing IN pairset, sites section
	void kw_site()
	{
		for( ; ; ){
			if( !is_strtype( tok() ) )
				throw SyntaxError();
			int	stype = advance().sym;
			d_.pos_open( stype );
			expect( SYM_LPAREN );
			a_list();
			expect( SYM_RPAREN );
			d_.pos_close();
			if( tok() == SYM_COLON ){
				advance();
				continue;
			}
			break;
		}
		expect( SYM_IN );
		Node	*ps = pairset();
		d_.si_close( ps );
	}

	// a_list : asgn | asgn ',' a_list   (nodes only matter in the score section)
	Node *a_list()
	{
		Node	*a = asgn();
		Node	*rest = nullptr;
		if( tok() == SYM_COMMA ){
			advance();
			rest = a_list();
		}
		if( d_.context == CTX_SCORE )
			return node( SYM_LIST, nullptr, a, rest );
		return nullptr;
	}

	Node *lval()
	{
		if( is_incr_op( tok() ) ){
			int	op = advance().sym;
			Node	*id = ident();
			return node( op, nullptr, nullptr, id );
		}
		Node	*id = ident();
		if( is_incr_op( tok() ) ){
			int	op = advance().sym;
			return node( op, nullptr, id, nullptr );
		}
		return id;
	}

	Node *ident()
	{
		if( tok() != SYM_IDENT )
			throw SyntaxError();
		Token	t = advance();
		return node( SYM_IDENT, &t.val, nullptr, nullptr );
	}

	bool starts_asgn() const
	{
		if( tok() == SYM_IDENT ){
			if( is_asgn_op( tok2() ) )
				return true;
			return false;
		}
		return false;
	}

	// asgn : lval asgn_op asgn | lval asgn_op expr
	Node *asgn()
	{
		Node	*lv = lval();
		if( !is_asgn_op( tok() ) )
			throw SyntaxError();
		int	op = advance().sym;
		Node	*rhs = starts_asgn() ? asgn() : expr();
		Node	*np = node( op, nullptr, lv, rhs );
		if( d_.context == CTX_PARMS )
			d_.parm_add( np );
		else if( d_.context == CTX_DESCR || d_.context == CTX_SITES )
			d_.se_addval( np );
		return np;
	}

	Node *expr()	// expr : conj | expr OR conj
	{
		Node	*l = conj();
		while( tok() == SYM_OR ){
			advance();
			Node	*r = conj();
			l = node( SYM_OR, nullptr, l, r );
		}
		return l;
	}

	Node *conj()	// conj : compare | compare AND conj
	{
		Node	*l = compare();
		if( tok() == SYM_AND ){
			advance();
			Node	*r = conj();
			return node( SYM_AND, nullptr, l, r );
		}
		return l;
	}

	static bool is_comp_op( int s )
	{
		return s == SYM_DONT_MATCH || s == SYM_EQUAL || s == SYM_GREATER || s == SYM_GREATER_EQUAL ||
			s == SYM_LESS || s == SYM_LESS_EQUAL || s == SYM_MATCH || s == SYM_NOT_EQUAL;
	}

	// compare : site | a_expr | a_expr comp_op a_expr
	Node *compare()
	{
		Node	*l;
		if( is_strtype( tok() ) ){
			Node	*sr = stref();
			if( tok() == SYM_COLON || tok() == SYM_IN ){
				// site : pairing IN pairset ; pairing : stref | stref ':' pairing
				std::vector<Node *>	refs{ sr };
				while( tok() == SYM_COLON ){
					advance();
					refs.push_back( stref() );
				}
				expect( SYM_IN );
				Node	*ps = pairset();
				Node	*pr = refs.back();
				for( int i = int( refs.size() ) - 2; i >= 0; i-- )
					pr = node( SYM_COLON, nullptr, refs[ i ], pr );
				return node( SYM_IN, nullptr, pr, ps );
			}
			l = a_expr_from( term_from( sr ) );
		}else
			l = a_expr();
		if( is_comp_op( tok() ) ){
			int	op = advance().sym;
			Node	*r = a_expr();
			return node( op, nullptr, l, r );
		}
		return l;
	}

	Node *a_expr() { return a_expr_from( term() ); }
	Node *a_expr_from( Node *l )
	{
		while( tok() == SYM_PLUS || tok() == SYM_MINUS ){
			int	op = advance().sym;
			Node	*r = term();
			l = node( op, nullptr, l, r );
		}
		return l;
	}
	Node *term() { return term_from( factor() ); }
	Node *term_from( Node *l )
	{
		while( tok() == SYM_PERCENT || tok() == SYM_SLASH || tok() == SYM_STAR ){
			int	op = advance().sym;
			Node	*r = factor();
			l = node( op, nullptr, l, r );
		}
		return l;
	}

	Node *factor()
	{
		if( tok() == SYM_MINUS ){
			advance();
			return node( SYM_NEGATE, nullptr, nullptr, primary() );
		}
		if( tok() == SYM_NOT ){
			advance();
			return node( SYM_NOT, nullptr, nullptr, primary() );
		}
		if( is_strtype( tok() ) )
			return stref();
		return primary();
	}

	// stref : strhdr '(' a_list ')' | strhdr '[' e_list ']'   (score section)
	Node *stref()
	{
		int	stype = advance().sym;
		if( d_.context != CTX_SCORE )
			throw SyntaxError();
		Node	*hdr = node( stype, nullptr, nullptr, nullptr );
		if( tok() == SYM_LPAREN ){
			advance();
			Node	*al = a_list();
			expect( SYM_RPAREN );
			return node( SYM_KW_STREF, nullptr, hdr, al );
		}
		if( tok() == SYM_LBRACK ){
			advance();
			Node	*el = e_list();
			expect( SYM_RBRACK );
			return node( SYM_IX_STREF, nullptr, hdr, el );
		}
		throw SyntaxError();
	}

	Node *e_list()
	{
		Node	*e = expr();
		Node	*rest = nullptr;
		if( tok() == SYM_COMMA ){
			advance();
			rest = e_list();
		}
		return node( SYM_LIST, nullptr, e, rest );
	}

	Node *primary()
	{
		switch( tok() ){
		case SYM_IDENT :
			if( tok2() == SYM_LPAREN )
				return fcall();
			return lval();
		case SYM_PLUS_PLUS :
		case SYM_MINUS_MINUS :
			return lval();
		case SYM_INT : {
			Token	t = advance();
			return node( SYM_INT, &t.val, nullptr, nullptr );
		}
		case SYM_FLOAT : {
			Token	t = advance();
			return node( SYM_FLOAT, &t.val, nullptr, nullptr );
		}
		case SYM_DOLLAR : {
			Token	t = advance();
			return node( SYM_DOLLAR, &t.val, nullptr, nullptr );
		}
		case SYM_STRING : {
			Token	t = advance();
			return node( SYM_STRING, &t.val, nullptr, nullptr );
		}
		case SYM_LCURLY :
			return pairset();
		case SYM_LPAREN : {
			advance();
			Node	*e = expr();
			expect( SYM_RPAREN );
			return e;
		}
		default :
			throw SyntaxError();
		}
	}

	Node *fcall()
	{
		Node	*id = ident();
		expect( SYM_LPAREN );
		Node	*el = e_list();
		expect( SYM_RPAREN );
		return node( SYM_CALL, nullptr, id, el );
	}

	// pairset : '{' s_list '}' ; PR_add runs innermost first (rmgrm.y:537-540),
	// i.e. the strings reach PR_close in reverse source order.
	Node *pairset()
	{
		expect( SYM_LCURLY );
		std::vector<const char *>	strs;
		for( ; ; ){
			if( tok() != SYM_STRING )
				throw SyntaxError();
			Token	t = advance();
			strs.push_back( ( const char * )t.val.pval );
			if( tok() == SYM_COMMA ){
				advance();
				continue;
			}
			break;
		}
		expect( SYM_RCURLY );
		if( strs.size() > 20 )
			fail( "%s:%d current pair too large.", d_.wdfname, d_.lineno );
		std::vector<const char *>	rev( strs.rbegin(), strs.rend() );
		return d_.pr_close( rev );
	}

	// ------------------------------------------------------------ score section
	void rule()	// rule : pattern action | action
	{
		if( tok() == SYM_LCURLY ){
			action();
			return;
		}
		Node	*pat;
		if( tok() == SYM_BEGIN ){
			advance();
			pat = node( SYM_BEGIN, nullptr, nullptr, nullptr );
		}else if( tok() == SYM_END ){
			advance();
			pat = node( SYM_END, nullptr, nullptr, nullptr );
		}else
			pat = expr();
		sc_.action( pat );
		action();
		sc_.endaction();
	}

	void action()
	{
		expect( SYM_LCURLY );
		stmt_list();
		expect( SYM_RCURLY );
	}

	void stmt_list()
	{
		stmt();
		while( tok() != SYM_RCURLY ){
			if( tok() == SYM_EOF )
				throw SyntaxError();
			stmt();
		}
	}

	Node *loop_level()
	{
		if( tok() == SYM_INT ){
			Token	t = advance();
			return node( SYM_INT, &t.val, nullptr, nullptr );
		}
		return nullptr;
	}

	void stmt()
	{
		switch( tok() ){
		case SYM_ACCEPT :
			advance();
			expect( SYM_SEMICOLON );
			sc_.accept();
			return;
		case SYM_REJECT :
			advance();
			expect( SYM_SEMICOLON );
			sc_.reject();
			return;
		case SYM_BREAK : {
			advance();
			Node	*lv = loop_level();
			expect( SYM_SEMICOLON );
			sc_.brk( lv );
			return;
		}
		case SYM_CONTINUE : {
			advance();
			Node	*lv = loop_level();
			expect( SYM_SEMICOLON );
			sc_.cont( lv );
			return;
		}
		case SYM_LCURLY :
			advance();
			stmt_list();
			expect( SYM_RCURLY );
			return;
		case SYM_SEMICOLON :
			advance();
			return;
		case SYM_HOLD : {
			advance();
			Node	*id = ident();
			expect( SYM_SEMICOLON );
			sc_.hold( id );
			return;
		}
		case SYM_RELEASE : {
			advance();
			Node	*id = ident();
			expect( SYM_SEMICOLON );
			sc_.release( id );
			return;
		}
		case SYM_IF : {
			advance();
			expect( SYM_LPAREN );
			Node	*e = expr();
			sc_.if_( e );
			expect( SYM_RPAREN );
			stmt();
			if( tok() == SYM_ELSE ){
				advance();
				sc_.else_();
				stmt();
				sc_.endelse();
			}else
				sc_.endif();
			return;
		}
		case SYM_WHILE : {
			advance();
			expect( SYM_LPAREN );
			Node	*e = expr();
			sc_.while_( e );
			expect( SYM_RPAREN );
			stmt();
			sc_.endwhile();
			return;
		}
		case SYM_FOR : {
			advance();
			expect( SYM_LPAREN );
			Node	*init = nullptr, *test = nullptr, *incr = nullptr;
			if( tok() != SYM_SEMICOLON )
				init = starts_asgn() ? asgn() : lval();
			sc_.forinit( init );
			expect( SYM_SEMICOLON );
			if( tok() != SYM_SEMICOLON )
				test = starts_asgn() ? asgn() : expr();
			sc_.fortest( test );
			expect( SYM_SEMICOLON );
			if( tok() != SYM_RPAREN )
				incr = starts_asgn() ? asgn() : lval();
			sc_.forincr( incr );
			expect( SYM_RPAREN );
			stmt();
			sc_.endfor();
			return;
		}
		case SYM_IDENT :
			if( tok2() == SYM_LPAREN ){	// call_stmt
				Node	*c = fcall();
				expect( SYM_SEMICOLON );
				sc_.expr( 0, c );
				sc_.clear();
				return;
			}
			if( is_asgn_op( tok2() ) ){	// asgn_stmt
				Node	*a = asgn();
				expect( SYM_SEMICOLON );
				sc_.mark();
				sc_.expr( 0, a );
				sc_.clear();
				return;
			}
			// fall through: auto_stmt  ident incr_op
		case SYM_PLUS_PLUS :
		case SYM_MINUS_MINUS : {
			Node	*a = lval();
			if( a->sym != SYM_PLUS_PLUS && a->sym != SYM_MINUS_MINUS )
				throw SyntaxError();
			expect( SYM_SEMICOLON );
			sc_.mark();
			sc_.expr( 0, a );
			sc_.clear();
			return;
		}
		default :
			throw SyntaxError();
		}
	}
};

bool parse_descriptor( Descriptor &d, const std::string &text )
{
	if( !d.score )
		d.score.reset( new ScoreVM( d ) );
	d.context = CTX_PARMS;
	try{
		Parser	p( d, text );
		p.program();
	}catch( SyntaxError & ){
		d.stderr_text += "yyerror: syntax error\n";
		return false;
	}
	return true;
}

}	// namespace rma

// rm_compile.cpp -- descriptor compiler: symbol table, parameter evaluation,
// pair sets, element parameters, tag linking, pseudoknot grouping, length
// propagation and the search list.  Follows /root/reference/src/compile.c
// (function:line cited per routine); data structures are this build's own.
#include "rm_host.h"
#include "rm_score.h"
#include <cctype>
#include <cstdarg>
#include <cstdlib>
#include <cstring>

namespace rma {

const char	*VERSION_STR = "v3.1.1 2015-may-13";

int	b2bc[ 256 ];

void fail( const char *fmt, ... )
{
	char	buf[ 2048 ];
	va_list	ap;
	va_start( ap, fmt );
	vsnprintf( buf, sizeof( buf ), fmt, ap );
	va_end( ap );
	throw Error( buf );
}

void Descriptor::note_error( const char *fmt, ... )
{
	char	buf[ 2048 ];
	va_list	ap;
	va_start( ap, fmt );
	vsnprintf( buf, sizeof( buf ), fmt, ap );
	va_end( ap );
	error = true;
	stderr_text += buf;
	stderr_text += "\n";
}

void init_b2bc()	// compile.c:180-187
{
	for( int i = 0; i < 256; i++ )
		b2bc[ i ] = RMA_BC_N;
	b2bc[ 'a' ] = b2bc[ 'A' ] = RMA_BC_A;
	b2bc[ 'c' ] = b2bc[ 'C' ] = RMA_BC_C;
	b2bc[ 'g' ] = b2bc[ 'G' ] = RMA_BC_G;
	b2bc[ 't' ] = b2bc[ 'T' ] = RMA_BC_T;
	b2bc[ 'u' ] = b2bc[ 'U' ] = RMA_BC_T;
}

static const char *iupac_class( int c )	// compile.c:189-206
{
	switch( tolower( c ) ){
	case 'a' : return "a";
	case 'b' : return "[cgt]";
	case 'c' : return "c";
	case 'd' : return "[agt]";
	case 'g' : return "g";
	case 'h' : return "[act]";
	case 'k' : return "[gt]";
	case 'm' : return "[ac]";
	case 'n' : return "[acgt]";
	case 'r' : return "[ag]";
	case 's' : return "[cg]";
	case 't' : return "t";
	case 'u' : return "t";
	case 'v' : return "[acg]";
	case 'w' : return "[at]";
	case 'y' : return "[ct]";
	default : return nullptr;
	}
}

static char *dupstr( const char *s )
{
	char	*p = ( char * )malloc( strlen( s ) + 1 );
	strcpy( p, s );
	return p;
}

const char *strel_name( int type )	// dump.c:600-649
{
	switch( type ){
	case SYM_CTX : return "ctx";
	case SYM_SS : return "ss";
	case SYM_H5 : return "h5";
	case SYM_H3 : return "h3";
	case SYM_P5 : return "p5";
	case SYM_P3 : return "p3";
	case SYM_T1 : return "t1";
	case SYM_T2 : return "t2";
	case SYM_T3 : return "t3";
	case SYM_Q1 : return "q1";
	case SYM_Q2 : return "q2";
	case SYM_Q3 : return "q3";
	case SYM_Q4 : return "q4";
	case SYM_SE : return "se";
	default : return "";
	}
}

Node *mk_node( Descriptor &d, int sym, const Value *vp, Node *left, Node *right )	// node.c:12-55
{
	Node	*np = new Node;
	np->sym = sym;
	np->filename = d.wdfname;
	np->lineno = d.lineno;
	np->left = left;
	np->right = right;
	switch( sym ){
	case SYM_IDENT :
		np->val.type = T_IDENT;
		np->val.pval = vp->pval;
		break;
	case SYM_INT :
		np->val.type = T_INT;
		np->val.ival = vp->ival;
		break;
	case SYM_FLOAT :
		np->val.type = T_FLOAT;
		np->val.dval = vp->dval;
		break;
	case SYM_STRING :
		np->val.type = T_STRING;
		np->val.pval = vp->pval;
		break;
	case SYM_DOLLAR :
		np->val.type = T_POS;
		np->val.pval = vp->pval;
		break;
	case SYM_CALL :
		np->val.type = T_IDENT;
		np->val.pval = left->val.pval;
		np->left = nullptr;
		break;
	default :
		break;
	}
	return np;
}

// ---------------------------------------------------------------- RM_init, compile.c:157-394
Descriptor::Descriptor()
{
	init_b2bc();
	descr.reserve( RMA_MAX_ELEMS );
}

Descriptor::~Descriptor() {}

static void init_globals( Descriptor &d )
{
	auto pairs = [&]( std::initializer_list<const char *> l ){
		std::vector<const char *>	cp( l );
		return d.pr_close( cp );
	};
	auto ent_int = [&]( const char *n, int v ){
		Value	val;
		val.type = T_INT;
		val.ival = v;
		return d.enter_id( n, T_INT, S_GLOBAL, 0, &val );
	};
	auto ent_str = [&]( const char *n, const char *v ){
		Value	val;
		val.type = T_STRING;
		val.pval = ( void * )v;
		return d.enter_id( n, T_STRING, S_GLOBAL, 0, &val );
	};
	d.lineno = 0;
	Node	*np = pairs( { "a:u", "c:g", "g:c", "u:a" } );
	d.enter_id( "wc", T_PAIRSET, S_GLOBAL, 0, &np->val );
	np = pairs( { "g:u", "u:g" } );
	d.enter_id( "gu", T_PAIRSET, S_GLOBAL, 0, &np->val );
	np = pairs( { "a:u:u" } );
	d.enter_id( "tr", T_PAIRSET, S_GLOBAL, 0, &np->val );
	np = pairs( { "g:g:g:g" } );
	d.enter_id( "qu", T_PAIRSET, S_GLOBAL, 0, &np->val );
	ent_int( "chk_both_strs", 1 );
	ent_int( "iupac", 1 );
	ent_int( "show_progress", 0 );
	ent_int( "ALL", 10000 );
	ent_int( "ctx_minlen", 0 );
	ent_int( "ctx_maxlen", 100 );
	ent_int( "ss_minlen", 1 );
	ent_int( "ss_maxlen", 1000 );
	ent_int( "wc_minlen", 3 );
	ent_int( "wc_maxlen", 30 );
	ent_str( "wc_ends", "pp" );
	ent_int( "wc_strict", d.args.strict_helices );
	ent_int( "phlx_minlen", 3 );
	ent_int( "phlx_maxlen", 30 );
	ent_str( "phlx_ends", "pp" );
	ent_int( "phlx_strict", d.args.strict_helices );
	ent_int( "tr_minlen", 3 );
	ent_int( "tr_maxlen", 30 );
	ent_str( "tr_ends", "pp" );
	ent_int( "tr_strict", d.args.strict_helices );
	ent_int( "qu_minlen", 3 );
	ent_int( "qu_maxlen", 30 );
	ent_str( "qu_ends", "pp" );
	ent_int( "qu_strict", d.args.strict_helices );
	ent_int( "windowsize", 6000 );
	ent_str( "efn_datadir", "" );
	ent_int( "efn_usestdbp", 1 );
	np = pairs( { "a:u", "c:g", "g:c", "g:u", "u:a", "u:g" } );
	d.efnstdbp = ( PairSet * )np->val.pval;
	d.enter_id( "efn_stdbp", T_PAIRSET, S_GLOBAL, 0, &np->val );

	Value	val;
	val.type = T_STRING;
	val.pval = nullptr;
	d.nval = &d.enter_id( "NAME", T_STRING, S_GLOBAL, 0, &val )->val;
	val.type = T_UNDEF;
	d.sval = &d.enter_id( "SCORE", T_UNDEF, S_GLOBAL, 1, &val )->val;
	d.cval = &ent_int( "COMP", 0 )->val;
	d.pval = &ent_int( "POS", 0 )->val;
	d.lval = &ent_int( "LEN", 0 )->val;
	ent_int( "NSE", 0 );
	ent_int( "SLEN", 0 );
	d.lineno = 0;
}

// ---------------------------------------------------------------- symbol table
Ident *Descriptor::enter_id( const char *name, int type, int scope, int reinit, const Value *vp )	// :1685
{
	Ident	*ip = new Ident;
	ip->name = name;
	ip->type = type;
	ip->scope = scope;
	ip->reinit = reinit;
	ip->val.type = type;
	ip->val.pval = nullptr;
	if( vp != nullptr ){
		if( type == T_INT )
			ip->val.ival = vp->ival;
		else if( type == T_FLOAT )
			ip->val.dval = vp->dval;
		else if( type == T_STRING )
			ip->val.pval = vp->pval ? dupstr( ( const char * )vp->pval ) : nullptr;
		else if( type == T_PAIRSET )
			ip->val.pval = vp->pval ? pair_copy( ( PairSet * )vp->pval ) : nullptr;
	}
	if( scope == S_GLOBAL ){
		if( globals.count( name ) )
			fail( "%s:%d attempt to redefine symbol '%s'.", wdfname, lineno, name );
		globals[ name ] = ip;
	}else{
		if( locals.size() >= 20 )
			fail( "%s:%d local symtab tab overflow.", wdfname, lineno );
		locals.push_back( ip );
	}
	return ip;
}

Ident *Descriptor::find_id( const char *name )	// :1752
{
	for( Ident *ip : locals )
		if( ip->name == name )
			return ip;
	auto	it = globals.find( name );
	return it == globals.end() ? nullptr : it->second;
}

int Descriptor::int_global( const char *name, int dflt )
{
	Ident	*ip = find_id( name );
	return ip != nullptr && ip->type == T_INT ? ip->val.ival : dflt;
}

// ---------------------------------------------------------------- pair sets
void Descriptor::mk_mats( PairSet *ps )	// mk_bmatp :2520, mk_rbmatp :2581
{
	memset( &ps->mat, 0, sizeof( ps->mat ) );
	if( ps->pairs.empty() )
		return;
	int	nb = ps->pairs[ 0 ].n_bases;
	ps->mat.n_bases = nb;
	for( const Pair &p : ps->pairs ){
		int	b1 = b2bc[ ( unsigned char )p.bases[ 0 ] ];
		int	b2 = b2bc[ ( unsigned char )p.bases[ 1 ] ];
		int	b3 = b2bc[ ( unsigned char )p.bases[ 2 ] ];
		int	b4 = b2bc[ ( unsigned char )p.bases[ 3 ] ];
		if( nb == 2 ){
			ps->mat.mat2 |= 1u << ( b1 * 5 + b2 );
		}else if( nb == 3 ){
			int	ix = ( b1 * 5 + b2 ) * 5 + b3;
			ps->mat.mat3[ ix >> 5 ] |= 1u << ( ix & 31 );
			ps->mat.mat2 |= 1u << ( b1 * 5 + b3 );
		}else if( nb == 4 ){
			int	ix = ( ( b1 * 5 + b2 ) * 5 + b3 ) * 5 + b4;
			ps->mat.mat4[ ix >> 5 ] |= 1u << ( ix & 31 );
			ps->mat.mat2 |= 1u << ( b1 * 5 + b4 );
		}
	}
}

PairSet *Descriptor::pair_check( PairSet *ps )	// pairop "check" :2296
{
	int	nb = UNDEF;
	for( Pair &p : ps->pairs ){
		if( nb == UNDEF )
			nb = p.n_bases;
		else if( p.n_bases != nb ){
			note_error( "%s:%d check: pairset contains elements with %d and %d bases.",
				wdfname, lineno, nb, p.n_bases );
			return ps;
		}
		for( int j = 0; j < p.n_bases; j++ )
			p.bases[ j ] = char( tolower( ( unsigned char )p.bases[ j ] ) );
	}
	int	n = int( ps->pairs.size() );
	for( int i = 0; i < n - 1; i++ ){
		Pair	&pi = ps->pairs[ i ];
		if( pi.n_bases == 0 )
			continue;
		for( int j = i + 1; j < n; j++ ){
			Pair	&pj = ps->pairs[ j ];
			if( pj.n_bases == 0 )
				continue;
			bool	diff = false;
			for( int b = 0; b < pi.n_bases; b++ ){
				if( pi.bases[ b ] != pj.bases[ b ] ){
					diff = true;
					break;
				}
			}
			if( !diff ){
				pj.n_bases = 0;
				note_error( "%s:%d check: pairset contains duplicate pair-strings.", wdfname, lineno );
			}
		}
	}
	std::vector<Pair>	kept;
	for( const Pair &p : ps->pairs )
		if( p.n_bases != 0 )
			kept.push_back( p );
	ps->pairs.swap( kept );
	mk_mats( ps );
	return ps;
}

PairSet *Descriptor::pair_copy( const PairSet *ps )	// "copy" :2357
{
	if( ps == nullptr )
		return nullptr;
	PairSet	*n = new PairSet( *ps );
	mk_mats( n );
	return n;
}

PairSet *Descriptor::pair_add( const PairSet *a, const PairSet *b )	// "add" :2384
{
	if( a->pairs[ 0 ].n_bases != b->pairs[ 0 ].n_bases )
		fail( "%s:%d add: pairsets have %d and %d elements.", wdfname, lineno,
			a->pairs[ 0 ].n_bases, b->pairs[ 0 ].n_bases );
	PairSet	*n = new PairSet;
	n->pairs = a->pairs;
	for( const Pair &pj : b->pairs ){
		// the reference appends pj as soon as any pair of a differs from it
		bool	added = false;
		for( const Pair &pi : a->pairs ){
			for( int k = 0; k < pi.n_bases; k++ ){
				if( pi.bases[ k ] != pj.bases[ k ] ){
					n->pairs.push_back( pj );
					added = true;
					break;
				}
			}
			if( added )
				break;
		}
	}
	mk_mats( n );
	return n;
}

PairSet *Descriptor::pair_sub( const PairSet *a, const PairSet *b )	// "sub" :2434
{
	if( a->pairs[ 0 ].n_bases != b->pairs[ 0 ].n_bases )
		fail( "%s:%d sub: pairsets have %d and %d elements.", wdfname, lineno,
			a->pairs[ 0 ].n_bases, b->pairs[ 0 ].n_bases );
	PairSet	*n = new PairSet;
	n->pairs = a->pairs;
	for( const Pair &pj : b->pairs ){
		// only a leading run of (already removed or equal) pairs is removed
		for( Pair &pn : n->pairs ){
			bool	differs = false;
			for( int k = 0; k < pn.n_bases; k++ ){
				if( pn.bases[ k ] != pj.bases[ k ] ){
					differs = true;
					break;
				}
			}
			if( differs )
				break;
			pn.n_bases = 0;
		}
	}
	std::vector<Pair>	kept;
	for( const Pair &p : n->pairs )
		if( p.n_bases != 0 )
			kept.push_back( p );
	n->pairs.swap( kept );
	// the reference reads ps_pairs[0].p_n_bases of the result to pick the
	// matrix kind; an emptied set keeps the arity of its operands
	mk_mats( n );
	if( n->pairs.empty() )
		n->mat.n_bases = a->pairs[ 0 ].n_bases;
	return n;
}

bool Descriptor::pair_equal( const PairSet *a, const PairSet *b )	// "equal" :2489
{
	if( a->pairs.size() != b->pairs.size() )
		return false;
	if( a->pairs.empty() )
		return true;
	if( a->pairs[ 0 ].n_bases != b->pairs[ 0 ].n_bases )
		return false;
	for( const Pair &pi : a->pairs ){
		bool	fnd = false;
		for( const Pair &pj : b->pairs ){
			fnd = true;
			for( int k = 0; k < pi.n_bases; k++ ){
				if( pi.bases[ k ] != pj.bases[ k ] ){
					fnd = false;
					break;
				}
			}
			if( fnd )
				break;
		}
		if( !fnd )
			return false;
	}
	return true;
}

Node *Descriptor::pr_close( std::vector<const char *> &curpair )	// PR_close :421
{
	PairSet	*ps = new PairSet;
	for( const char *s : curpair ){
		Pair	p;
		bool	needbase = true;
		int	b = 0;
		for( const char *bp = s; *bp; bp++ ){
			int	c = ( unsigned char )*bp;
			bool	isbase = strchr( "acgtuACGTU", c ) != nullptr;
			if( isbase ){
				if( needbase ){
					if( b >= 4 ){
						note_error( "%s:%d At most 4 bases in a pair-string.", wdfname, lineno );
						break;
					}
					p.n_bases = b + 1;
					p.bases[ b++ ] = char( c );
					needbase = false;
				}else{
					note_error( "%s:%d pair-string is bsse-letter : base-letter : ...", wdfname, lineno );
					break;
				}
			}else if( c == ':' ){
				if( needbase ){
					note_error( "%s:%d pair-string is bsse-letter : base-letter : ...", wdfname, lineno );
					break;
				}
				needbase = true;
			}else{
				note_error( "%s:%d pair-string is bsse-letter : base-letter : ...", wdfname, lineno );
				break;
			}
		}
		if( p.n_bases < 2 || p.n_bases > 4 )
			note_error( "%s:%d pair-string has 2-4 bases", wdfname, lineno );
		ps->pairs.push_back( p );
	}
	ps = pair_check( ps );
	Node	*np = new Node;
	np->sym = SYM_PAIRSET;
	np->lineno = lineno;
	np->filename = wdfname;
	np->val.type = T_PAIRSET;
	np->val.pval = ps;
	return np;
}

// ---------------------------------------------------------------- positions
Pos *Descriptor::pos_cvt( Value *vp )	// posop "cvt" :2632
{
	if( vp->ival < 0 )
		fail( "%s:%d cvt: only ints > 0 can be converted to positions.", wdfname, lineno );
	Pos	*n = new Pos;
	n->type = SYM_DOLLAR;
	n->lineno = lineno;
	n->addr.l2r = 1;
	n->addr.offset = vp->ival;
	vp->type = T_POS;
	vp->pval = n;
	return n;
}

Pos *Descriptor::pos_sub( Pos *l, Pos *r )	// posop "sub" :2653
{
	if( l->addr.l2r || !r->addr.l2r )
		fail( "%s:%d sub: expr must have the form '$ - expr'; expr is int valued > 0.", wdfname, lineno );
	Pos	*n = new Pos;
	n->type = SYM_DOLLAR;
	n->lineno = lineno;
	n->addr.l2r = 0;
	n->addr.offset = l->addr.offset + r->addr.offset;
	return n;
}

// ---------------------------------------------------------------- parameter evaluation
int Descriptor::loadidval( Value *vp )	// :2196
{
	Ident	*ip = ( Ident * )vp->pval;
	int	type = ip->type;
	if( type == T_INT ){
		if( ip->val.ival == UNDEF )
			fail( "%s:%d id '%s' has int value UNDER.", wdfname, lineno, ip->name.c_str() );
		vp->type = T_INT;
		vp->ival = ip->val.ival;
	}else if( type == T_FLOAT ){
		if( ip->val.ival == UNDEF )	// (sic) the reference tests the int view
			fail( "%s:%d id '%s' has int value UNDEF.", wdfname, lineno, ip->name.c_str() );
		vp->type = T_FLOAT;
		vp->dval = ip->val.dval;
	}else if( type == T_STRING ){
		if( ip->val.pval == nullptr )
			fail( "%s:%d id '%s' has string value NULL.", wdfname, lineno, ip->name.c_str() );
		vp->type = T_STRING;
		vp->pval = dupstr( ( const char * )ip->val.pval );
	}else if( type == T_PAIRSET ){
		PairSet	*ps;
		if( ip->val.pval == nullptr ){
			if( open_pairset != nullptr )
				ps = open_pairset;
			else
				fail( "%s:%d id '%s' has pair value NULL.", wdfname, lineno, ip->name.c_str() );
		}else
			ps = ( PairSet * )ip->val.pval;
		vp->type = T_PAIRSET;
		vp->pval = pair_copy( ps );
	}
	return type;
}

void Descriptor::storeexprval( Ident *ip, Value *vp )	// :2253
{
	switch( vp->type ){
	case T_INT :
		ip->type = T_INT;
		ip->val.type = T_INT;
		ip->val.ival = vp->ival;
		break;
	case T_FLOAT :
		ip->type = T_FLOAT;
		ip->val.type = T_FLOAT;
		ip->val.dval = vp->dval;
		break;
	case T_STRING :
		ip->type = T_STRING;
		ip->val.type = T_STRING;
		ip->val.pval = dupstr( ( const char * )vp->pval );
		break;
	case T_PAIRSET :
		ip->type = T_PAIRSET;
		ip->val.type = T_PAIRSET;
		ip->val.pval = vp->pval;
		break;
	case T_POS :
		ip->type = T_POS;
		ip->val.type = T_POS;
		ip->val.pval = vp->pval;
		break;
	default :
		break;
	}
}

#define TIJ( i, j )	( ( i ) * 8 + ( j ) )

void Descriptor::eval( Node *expr, bool d_ok )	// :1851
{
	if( expr == nullptr )
		return;
	eval( expr->left, d_ok );
	eval( expr->right, d_ok );
	auto top = [&]( int k ) -> Value & { return valstk[ valstk.size() - k ]; };
	auto need = [&]( size_t n ){
		if( valstk.size() < n )
			fail( "%s:%d malformed parameter expression.", wdfname, lineno );
	};
	switch( expr->sym ){
	case SYM_INT : {
		Value	v;
		v.type = T_INT;
		v.ival = expr->val.ival;
		valstk.push_back( v );
		break;
	}
	case SYM_FLOAT : {
		Value	v;
		v.type = T_FLOAT;
		v.dval = expr->val.dval;
		valstk.push_back( v );
		break;
	}
	case SYM_STRING : {
		Value	v;
		v.type = T_STRING;
		v.pval = dupstr( ( const char * )expr->val.pval );
		valstk.push_back( v );
		break;
	}
	case SYM_PAIRSET : {
		Value	v;
		v.type = T_PAIRSET;
		v.pval = expr->val.pval;
		valstk.push_back( v );
		break;
	}
	case SYM_DOLLAR : {
		Value	v;
		v.type = T_POS;
		v.pval = expr->val.pval;
		valstk.push_back( v );
		break;
	}
	case SYM_IDENT : {
		const char	*name = ( const char * )expr->val.pval;
		Ident	*ip = find_id( name );
		if( ip == nullptr ){
			if( d_ok )
				ip = enter_id( name, T_UNDEF, S_GLOBAL, 0, nullptr );
			else
				fail( "%s:%d unknown id '%s'.", wdfname, lineno, name );
		}
		Value	v;
		v.type = T_IDENT;
		v.pval = ip;
		valstk.push_back( v );
		break;
	}
	case SYM_PLUS :
	case SYM_MINUS : {
		need( 2 );
		bool	plus = expr->sym == SYM_PLUS;
		int	lt = top( 2 ).type;
		if( lt == T_IDENT )
			lt = loadidval( &top( 2 ) );
		int	rt = top( 1 ).type;
		if( rt == T_IDENT )
			rt = loadidval( &top( 1 ) );
		Value	&l = top( 2 ), &r = top( 1 );
		switch( TIJ( lt, rt ) ){
		case TIJ( T_INT, T_INT ) :
			l.ival = plus ? l.ival + r.ival : l.ival - r.ival;
			break;
		case TIJ( T_INT, T_FLOAT ) :
			l.ival = plus ? int( l.ival + r.dval ) : int( l.ival - r.dval );
			break;
		case TIJ( T_FLOAT, T_INT ) :
			l.dval = plus ? l.dval + r.ival : l.dval - r.ival;
			break;
		case TIJ( T_FLOAT, T_FLOAT ) :
			l.dval = plus ? l.dval + r.dval : l.dval - r.dval;
			break;
		case TIJ( T_STRING, T_STRING ) :
			if( !plus )
				fail( "%s:%d type mismatch '-'.", wdfname, lineno );
			{
				std::string	s = std::string( ( char * )l.pval ) + ( char * )r.pval;
				l.pval = dupstr( s.c_str() );
			}
			break;
		case TIJ( T_PAIRSET, T_PAIRSET ) :
			l.pval = plus ? pair_add( ( PairSet * )l.pval, ( PairSet * )r.pval )
				: pair_sub( ( PairSet * )l.pval, ( PairSet * )r.pval );
			break;
		case TIJ( T_POS, T_INT ) :
			if( plus )
				fail( "%s:%d type mismatch '+'.", wdfname, lineno );
			pos_cvt( &r );
			l.pval = pos_sub( ( Pos * )l.pval, ( Pos * )r.pval );
			break;
		default :
			fail( "%s:%d type mismatch '%c'.", wdfname, lineno, plus ? '+' : '-' );
		}
		valstk.pop_back();
		break;
	}
	case SYM_NEGATE : {
		need( 1 );
		int	rt = top( 1 ).type;
		if( rt == T_IDENT ){
			// compile.c:2028-2029 loads valstk[n-2] here; with a single
			// operand on the stack that is out of range, so load the operand
			rt = loadidval( &top( 1 ) );
		}
		if( rt == T_INT )
			top( 1 ).ival = -top( 1 ).ival;
		else if( rt == T_FLOAT )
			top( 1 ).dval = -top( 1 ).dval;
		else
			fail( "%s:%d type mismatch '-'.", wdfname, lineno );
		break;
	}
	case SYM_ASSIGN : {
		need( 2 );
		if( top( 2 ).type != T_IDENT )
			fail( "%s:%d left side of '=' is not a variable.", wdfname, lineno );
		Ident	*ip = ( Ident * )top( 2 ).pval;
		int	lt = ip->type;
		int	rt = top( 1 ).type;
		if( rt == T_IDENT )
			rt = loadidval( &top( 1 ) );
		if( lt == T_UNDEF ){
			lt = rt;
			ip->type = lt;
		}
		Value	&r = top( 1 );
		switch( TIJ( lt, rt ) ){
		case TIJ( T_INT, T_INT ) :
		case TIJ( T_FLOAT, T_FLOAT ) :
		case TIJ( T_STRING, T_STRING ) :
		case TIJ( T_PAIRSET, T_PAIRSET ) :
		case TIJ( T_POS, T_POS ) :
			break;
		case TIJ( T_INT, T_FLOAT ) :
			r.type = T_INT;
			r.ival = int( r.dval );
			break;
		case TIJ( T_FLOAT, T_INT ) :
			r.type = T_FLOAT;
			r.dval = r.ival;
			break;
		case TIJ( T_POS, T_INT ) :
			pos_cvt( &r );
			break;
		default :
			fail( "%s:%d type mismatch '='.", wdfname, lineno );
		}
		storeexprval( ip, &r );
		valstk.pop_back();
		valstk.pop_back();
		break;
	}
	case SYM_PLUS_ASSIGN :
	case SYM_MINUS_ASSIGN : {
		need( 2 );
		bool	plus = expr->sym == SYM_PLUS_ASSIGN;
		if( top( 2 ).type != T_IDENT )
			fail( "%s:%d left side of assignment is not a variable.", wdfname, lineno );
		Ident	*ip = ( Ident * )top( 2 ).pval;
		int	lt = loadidval( &top( 2 ) );
		int	rt = top( 1 ).type;
		if( rt == T_IDENT )
			rt = loadidval( &top( 1 ) );
		Value	&l = top( 2 ), &r = top( 1 );
		switch( TIJ( lt, rt ) ){
		case TIJ( T_INT, T_INT ) :
			l.ival = plus ? l.ival + r.ival : l.ival - r.ival;
			break;
		case TIJ( T_INT, T_FLOAT ) :
			l.ival = plus ? int( l.ival + r.dval ) : int( l.ival - r.dval );
			break;
		case TIJ( T_FLOAT, T_INT ) :
			l.dval = plus ? l.dval + r.ival : l.dval - r.ival;
			break;
		case TIJ( T_FLOAT, T_FLOAT ) :
			l.dval = plus ? l.dval + r.dval : l.dval - r.dval;
			break;
		case TIJ( T_STRING, T_STRING ) :
			if( !plus )
				fail( "%s:%d type mimatch '-='.", wdfname, lineno );
			{
				std::string	s = std::string( ( char * )l.pval ) + ( char * )r.pval;
				l.pval = dupstr( s.c_str() );
			}
			break;
		case TIJ( T_PAIRSET, T_PAIRSET ) :
			l.pval = plus ? pair_add( ( PairSet * )l.pval, ( PairSet * )r.pval )
				: pair_sub( ( PairSet * )l.pval, ( PairSet * )r.pval );
			break;
		default :
			fail( "%s:%d type mismatch '%s'.", wdfname, lineno, plus ? "+=" : "-=" );
		}
		storeexprval( ip, &l );
		valstk.pop_back();
		valstk.pop_back();
		break;
	}
	default :
		fail( "%s:%d operator %d not implemented.", wdfname, lineno, expr->sym );
	}
}

void Descriptor::parm_add( Node *expr )	// PARM_add :396
{
	valstk.clear();
	eval( expr, true );
}

void Descriptor::se_addval( Node *expr )	// SE_addval :686
{
	valstk.clear();
	eval( expr, false );
}

char *Descriptor::str2seq( const char *str )	// RM_str2seq :2682
{
	if( str == nullptr || *str == '\0' )
		return dupstr( "" );
	Ident	*ip = find_id( "iupac" );
	int	iupac = ip ? ip->val.ival : 0;
	std::string	seq;
	for( const char *s = str; *s; s++ ){
		int	c = ( unsigned char )*s;
		if( isupper( c ) )
			c = tolower( c );
		if( c == 'u' )
			c = 't';
		const char	*cl = iupac ? iupac_class( c ) : nullptr;
		if( cl )
			seq += cl;
		else
			seq += char( c );
	}
	return dupstr( seq.c_str() );
}

// ---------------------------------------------------------------- structure elements
void Descriptor::se_open( int stype )	// SE_open :501
{
	valstk.clear();
	if( stype == SYM_SE )
		fail( "%s:%d strel 'se' allowed only in score section.", wdfname, lineno );
	if( stype != SYM_CTX ){
		if( rctx != nullptr )
			fail( "%s:%d Right ctx element must be last element.", wdfname, lineno );
		if( descr.size() == RMA_MAX_ELEMS )
			fail( "%s:%d descr array size(%d) exceeded.", wdfname, lineno, RMA_MAX_ELEMS );
		descr.emplace_back();
		open_stp = &descr.back();
	}else if( descr.empty() ){
		if( lctx != nullptr )
			fail( "%s:%d ctx elements must contain a real descriptor.", wdfname, lineno );
		open_stp = lctx = new Strel;
		lctx_explicit = true;
	}else if( rctx == nullptr ){
		open_stp = rctx = new Strel;
		rctx_explicit = true;
	}else
		fail( "%s:%d Descr can contain at most 1 right ctx element.", wdfname, lineno );
	se_init( open_stp, stype );
}

void Descriptor::se_init( Strel *stp, int stype )	// SE_init :555
{
	*stp = Strel();
	stp->type = stype;
	stp->index = stype != SYM_CTX ? int( descr.size() ) - 1 : UNDEF;
	stp->lineno = lineno;
	if( stype != SYM_SS && stype != SYM_CTX ){
		stp->attr[ SA_ENDS ] = UNDEF;
		stp->attr[ SA_STRICT ] = UNDEF;
	}
	locals.clear();
	Value	v;
	v.type = T_STRING;
	v.pval = nullptr;
	enter_id( "tag", T_STRING, S_STREL, 0, &v );
	v.type = T_INT;
	v.ival = UNDEF;
	enter_id( "minlen", T_INT, S_STREL, 0, &v );
	enter_id( "maxlen", T_INT, S_STREL, 0, &v );
	enter_id( "len", T_INT, S_STREL, 0, &v );
	v.type = T_STRING;
	v.pval = nullptr;
	enter_id( "seq", T_STRING, S_STREL, 0, &v );
	v.type = T_INT;
	v.ival = UNDEF;
	enter_id( "mismatch", T_INT, S_STREL, 0, &v );
	v.type = T_FLOAT;
	v.dval = 1.0;
	enter_id( "matchfrac", T_FLOAT, S_STREL, 0, &v );
	if( stype != SYM_SS && stype != SYM_CTX ){
		v.type = T_INT;
		v.ival = UNDEF;
		enter_id( "mispair", T_INT, S_STREL, 0, &v );
		v.type = T_FLOAT;
		v.dval = UNDEF;
		enter_id( "pairfrac", T_FLOAT, S_STREL, 0, &v );
		v.type = T_STRING;
		v.pval = nullptr;
		enter_id( "ends", T_STRING, S_STREL, 0, &v );
		v.type = T_INT;
		v.ival = UNDEF;
		enter_id( "strict", T_INT, S_STREL, 0, &v );
		const char	*dflt = "wc";
		if( stype == SYM_T1 || stype == SYM_T2 || stype == SYM_T3 )
			dflt = "tr";
		else if( stype == SYM_Q1 || stype == SYM_Q2 || stype == SYM_Q3 || stype == SYM_Q4 )
			dflt = "qu";
		open_pairset = ( PairSet * )find_id( dflt )->val.pval;
		v.type = T_PAIRSET;
		v.pval = nullptr;
		enter_id( "pair", T_PAIRSET, S_STREL, 0, &v );
	}
}

void Descriptor::se_close()	// SE_close :693
{
	bool	s_minlen = false, s_maxlen = false, s_mispair = false, s_pairfrac = false;
	for( Ident *ip : locals ){
		const std::string	&n = ip->name;
		if( n == "tag" )
			open_stp->tag = ( const char * )ip->val.pval;
		else if( n == "minlen" ){
			s_minlen = ip->val.ival != UNDEF;
			open_stp->minlen = ip->val.ival;
		}else if( n == "maxlen" ){
			s_maxlen = ip->val.ival != UNDEF;
			open_stp->maxlen = ip->val.ival;
		}else if( n == "len" ){
			if( ip->val.ival != UNDEF ){
				if( s_minlen || s_maxlen )
					note_error( "%s:%d len= can't be used with minlen=/maxlen=.", wdfname, lineno );
				else
					open_stp->minlen = open_stp->maxlen = ip->val.ival;
			}
		}else if( n == "seq" )
			open_stp->seq = ( const char * )ip->val.pval;
		else if( n == "mismatch" )
			open_stp->mismatch = ip->val.ival;
		else if( n == "matchfrac" ){
			if( ip->val.dval < 0. || ip->val.dval > 1. )
				note_error( "%s:%d matchfrac must be >= 0 and <= 1.", wdfname, lineno );
			else
				open_stp->matchfrac = ip->val.dval;
		}else if( n == "mispair" ){
			s_mispair = ip->val.ival != UNDEF;
			if( s_mispair ){
				if( s_pairfrac )
					note_error( "%s:%d mispair= can't be used with pairfrac=.", wdfname, lineno );
				else if( ip->val.ival < 0 )
					note_error( "%s:%d bad mispair value %d, must be >= 0.", wdfname, lineno, ip->val.ival );
			}
			open_stp->mispair = ip->val.ival;
		}else if( n == "pairfrac" ){
			s_pairfrac = ip->val.dval != UNDEF;
			if( s_pairfrac ){
				if( s_mispair )
					note_error( "%s:%d pairfrac= can't be used with mispair=.", wdfname, lineno );
				else if( ip->val.dval < 0. || ip->val.dval > 1. )
					note_error( "%s:%d pairfrac must be >= 0 and <= 1.", wdfname, lineno );
			}
			open_stp->pairfrac = ip->val.dval;
		}else if( n == "pair" )
			open_stp->pairset = ( PairSet * )ip->val.pval;
		else if( n == "ends" ){
			if( ip->val.pval != nullptr )
				open_stp->attr[ SA_ENDS ] = ( signed char )ends2attr( ( const char * )ip->val.pval );
		}else if( n == "strict" ){
			if( ip->val.ival != UNDEF )
				open_stp->attr[ SA_STRICT ] = ( signed char )strict2attr( ip->val.ival );
		}
	}
	open_pairset = nullptr;
	open_stp = nullptr;
	locals.clear();
}

int Descriptor::ends2attr( const char *str )	// :1797
{
	if( str == nullptr || *str == '\0' )
		return 0;
	if( strlen( str ) != 2 ){
		note_error( "%s:%d end values are \"pp\", \"mp\", \"pm\" & \"mm\".", wdfname, lineno );
		return 0;
	}
	char	l[ 3 ] = { char( tolower( ( unsigned char )str[ 0 ] ) ), char( tolower( ( unsigned char )str[ 1 ] ) ), 0 };
	if( !strcmp( l, "pp" ) )
		return RMA_5PAIRED | RMA_3PAIRED;
	if( !strcmp( l, "mp" ) )
		return RMA_3PAIRED;
	if( !strcmp( l, "pm" ) )
		return RMA_5PAIRED;
	if( !strcmp( l, "mm" ) )
		return 0;
	note_error( "%s:%d end values are \"pp\", \"mp\", \"pm\" & \"mm\".", wdfname, lineno );
	return 0;
}

int Descriptor::strict2attr( int sval )	// :1829
{
	switch( sval ){
	case UNDEF :
	case 0 :
		return 0;
	case 1 :
	case 35 :
	case 53 :
		return RMA_5STRICT | RMA_3STRICT;
	case 3 :
		return RMA_3STRICT;
	case 5 :
		return RMA_5STRICT;
	default :
		note_error( "%s:%d strict values are 0, 1, 3, 5, 35, 53", wdfname, lineno );
		return 0;
	}
}

// ---------------------------------------------------------------- sites
void Descriptor::pos_open( int ptype )	// POS_open :2728
{
	valstk.clear();
	if( ptype == SYM_SE )
		fail( "%s:%d site tpe 'se' allowed only in score section.", wdfname, lineno );
	if( cur_pos.size() == 10 )
		fail( "%s:%d pos array size(%d) esceeded.", wdfname, lineno, 10 );
	cur_pos.emplace_back();
	posp = &cur_pos.back();
	posp->type = ptype;
	posp->lineno = lineno;
	posp->addr.l2r = 1;
	posp->addr.offset = 0;
	locals.clear();
	Value	v;
	v.type = T_STRING;
	v.pval = nullptr;
	enter_id( "tag", T_STRING, S_SITE, 0, &v );
	v.type = T_POS;
	v.pval = nullptr;
	enter_id( "pos", T_POS, S_SITE, 0, &v );
}

void Descriptor::pos_close()	// POS_close :2767
{
	for( Ident *ip : locals ){
		if( ip->name == "tag" )
			posp->tag = ( const char * )ip->val.pval;
		else if( ip->name == "pos" ){
			Pos	*ipos = ( Pos * )ip->val.pval;
			if( ipos == nullptr )
				fail( "%s:%d site position has no pos= value.", wdfname, lineno );
			posp->addr = ipos->addr;
		}
	}
	locals.clear();
}

void Descriptor::si_close( Node *expr )	// SI_close :2786
{
	Site	s;
	s.pos = cur_pos;
	s.pairset = pair_copy( ( PairSet * )expr->val.pval );
	sites.push_back( s );
	cur_pos.clear();
}

bool Descriptor::chk_site( Site &s )	// chk_site :2824
{
	bool	err = false;
	if( int( s.pos.size() ) != s.pairset->pairs[ 0 ].n_bases ){
		err = true;
		note_error( "%s:%d Number of positions in site must agree with pairset.", wdfname, s.pos[ 0 ].lineno );
	}
	for( Pos &p : s.pos ){
		if( p.tag == nullptr ){
			err = true;
			note_error( "%s:%d all positions must be tagged.", wdfname, p.lineno );
			continue;
		}
		for( Strel &st : descr ){
			if( st.tag == nullptr || st.type != p.type )
				continue;
			if( !strcmp( p.tag, st.tag ) ){
				p.descr = &st;
				break;
			}
		}
		if( p.descr == nullptr ){
			err = true;
			note_error( "%s:%d position with undefined tag '%s'.", wdfname, p.lineno, p.tag );
		}
	}
	for( Pos &p : s.pos ){
		if( p.descr == nullptr )
			continue;
		if( p.addr.l2r ){
			if( p.addr.offset > p.descr->minlen ){
				err = true;
				note_error( "%s:%d position offset > strel minlen.", wdfname, p.lineno );
			}
		}else if( p.addr.offset + 1 > p.descr->minlen ){
			err = true;
			note_error( "%s:%d position offset > strel minlen.", wdfname, p.lineno );
		}
	}
	return err;
}

// ---------------------------------------------------------------- linking
void Descriptor::chk_context()	// :822
{
	if( !args.show_context )
		return;
	if( lctx == nullptr ){
		open_stp = lctx = new Strel;
		se_init( lctx, SYM_CTX );
		se_close();
	}
	if( rctx == nullptr ){
		open_stp = rctx = new Strel;
		se_init( rctx, SYM_CTX );
		se_close();
	}
}

void Descriptor::mk_links( int n_tags, Strel *tags[] )	// :1092
{
	for( int i = 0; i < n_tags; i++ ){
		tags[ i ]->mates.clear();
		tags[ i ]->scopes.clear();
		for( int j = 0; j < n_tags; j++ ){
			if( j != i )
				tags[ i ]->mates.push_back( tags[ j ] );
			tags[ i ]->scopes.push_back( tags[ j ] );
		}
		tags[ i ]->scope = i;
	}
}

void Descriptor::chk_tagorder( int n_tags, Strel *tags[] )	// :1010
{
	auto dup = [&]( int need ){
		for( int i = need; i < n_tags && i < 4; i++ )
			note_error( "%s:%d duplicate tag '%s'.", wdfname, tags[ i ]->lineno, tags[ i ]->tag );
		if( n_tags > 4 )
			note_error( "%s:%d duplicate tag '%s'.", wdfname, tags[ 0 ]->lineno, tags[ 0 ]->tag );
	};
	int	t1 = tags[ 0 ]->type;
	const char	*tag = tags[ 0 ]->tag ? tags[ 0 ]->tag : "";
	if( t1 == SYM_SS ){
		if( n_tags > 1 )
			dup( 1 );
	}else if( t1 == SYM_H5 || t1 == SYM_P5 ){
		int	want = t1 == SYM_H5 ? SYM_H3 : SYM_P3;
		if( n_tags < 2 ){
			if( t1 == SYM_H5 )
				note_error( "%s:%d wc-helix '%s' has not h3() element.", wdfname, tags[ 0 ]->lineno, tag );
			else
				note_error( "%s:%d parallel-helix '%s' has no p3() element.", wdfname, tags[ 0 ]->lineno, tag );
		}else if( tags[ 1 ]->type == want ){
			if( n_tags == 2 )
				mk_links( n_tags, tags );
			else
				dup( 2 );
		}else
			dup( 1 );
	}else if( t1 == SYM_T1 ){
		if( n_tags < 3 )
			note_error( "%s:%d triplex '%s' has < 3 elements.", wdfname, tags[ 0 ]->lineno, tag );
		else if( tags[ 1 ]->type == SYM_T2 && tags[ 2 ]->type == SYM_T3 ){
			if( n_tags == 3 )
				mk_links( n_tags, tags );
			else
				dup( 3 );
		}else
			dup( 2 );
	}else if( t1 == SYM_Q1 ){
		if( n_tags < 4 )
			note_error( "%s:%d 4-plex '%s' has < 4 elements.", wdfname, tags[ 0 ]->lineno, tag );
		else if( tags[ 1 ]->type == SYM_Q2 && tags[ 2 ]->type == SYM_Q3 && tags[ 3 ]->type == SYM_Q4 ){
			if( n_tags == 4 )
				mk_links( n_tags, tags );
			else
				dup( 4 );
		}else
			dup( 3 );
	}else
		note_error( "%s:%d 1st use of tag '%s' is out of order.", wdfname, tags[ 0 ]->lineno, tag );
}

bool Descriptor::chk_proper_nesting( Strel *a, Strel *b )	// :1129
{
	for( int i = a->index + 1; i < b->index; i++ ){
		for( Strel *m : descr[ i ].mates ){
			if( m->index < a->index || m->index > b->index )
				return false;
		}
	}
	return true;
}

void Descriptor::find_pknots( Strel *stp )	// :1148
{
	if( stp->type == SYM_SS ){
		stp->checked = 1;
		return;
	}
	if( stp->attr[ SA_PROPER ] ){
		stp->checked = 1;
		for( Strel *m : stp->mates )
			m->checked = 1;
		return;
	}
	int	n_descr = int( descr.size() );
	std::vector<int>	pk( n_descr + 1, UNDEF );
	Strel	*stp3 = stp->mates[ 0 ];
	int	fd0 = stp->index, fd = stp->index, ld = stp3->index;
	for( int d = fd; d < ld; d++ )
		pk[ d ] = stp->index;
	pk[ ld ] = ld;
	for( bool diff = true; diff; ){
		diff = false;
		for( int d = fd + 1; d < ld; d++ ){
			Strel	*s1 = &descr[ d ];
			if( s1->type != SYM_H5 || s1->attr[ SA_PROPER ] || s1->checked )
				continue;
			Strel	*s2 = s1->mates[ 0 ];
			int	fd1 = s1->index, ld1 = s2->index;
			if( pk[ fd1 ] == pk[ ld1 ] )
				continue;
			s1->checked = 1;
			diff = true;
			int	d0 = pk[ fd1 ];
			for( int d1 = fd1; d1 < ld1; d1++ ){
				if( pk[ d1 ] == d0 )
					pk[ d1 ] = fd1;
				else
					break;
			}
			if( ld1 < ld ){
				d0 = pk[ ld1 ];
				for( int d1 = ld1; d1 < ld; d1++ ){
					if( pk[ d1 ] == d0 )
						pk[ d1 ] = ld1;
					else
						break;
				}
			}else{
				d0 = pk[ ld ];
				for( int d1 = ld + 1; d1 < ld1; d1++ )
					pk[ d1 ] = d0;
				pk[ ld1 ] = ld1;
				ld = ld1;
			}
		}
	}
	for( int d = fd; d <= ld; d++ )
		pk[ d - fd0 ] = pk[ d ];
	ld -= fd;
	int	d0 = pk[ 0 ], n_pk = 1;
	for( int d = 1; d <= ld; d++ ){
		if( pk[ d ] != d0 ){
			d0 = pk[ d ];
			pk[ n_pk++ ] = pk[ d ];
		}
	}
	std::vector<Strel *>	group;
	for( int j = 0; j < n_pk; j++ )
		group.push_back( &descr[ pk[ j ] ] );
	for( int i = 0; i < n_pk; i++ ){
		Strel	*s1 = group[ i ];
		s1->checked = 1;
		s1->scopes = group;
		s1->scope = i;
	}
}

Strel *Descriptor::set_scopes( int fd, int ld, std::vector<Strel *> &stk )	// :2882
{
	if( fd > ld )
		return nullptr;
	int	nd;
	for( int d = fd; d <= ld; d = nd ){
		Strel	*stp = &descr[ d ];
		stp->outer = stk.back();
		if( stp->scopes.empty() ){
			nd = d + 1;
			if( nd <= ld ){
				stp->next = &descr[ nd ];
				stp->next->prev = stp;
			}
			continue;
		}
		stk.push_back( stp );
		int	ns = int( stp->scopes.size() );
		for( int s = 0; s < ns - 1; s++ ){
			Strel	*s1 = stp->scopes[ s ], *s2 = stp->scopes[ s + 1 ];
			stk.push_back( s1 );
			s1->inner = set_scopes( s1->index + 1, s2->index - 1, stk );
			stk.pop_back();
			s2->outer = stk.back();
		}
		Strel	*sl = stp->scopes[ ns - 1 ];
		stk.pop_back();
		nd = sl->index + 1;
		if( nd <= ld ){
			stp->next = &descr[ nd ];
			stp->next->prev = stp;
		}
	}
	return &descr[ fd ];
}

void Descriptor::link_tags()	// :857
{
	int	n_descr = int( descr.size() );
	for( Strel &st : descr ){
		if( st.type == SYM_SS || st.type == SYM_H5 || st.type == SYM_H3 ||
			st.type == SYM_P5 || st.type == SYM_P3 )
			continue;
		if( st.tag == nullptr )
			note_error( "%s:%d all triple/quad. helix els. must be tagged.", wdfname, lineno );
	}
	// explicitly tagged elements
	for( int i = 0; i < n_descr; i++ ){
		Strel	*stp = &descr[ i ];
		if( stp->checked )
			continue;
		stp->checked = 1;
		if( stp->tag == nullptr )
			continue;
		Strel	*tags[ 4 ] = { stp, nullptr, nullptr, nullptr };
		int	n_tags = 1;
		for( int j = i + 1; j < n_descr; j++ ){
			Strel	*s1 = &descr[ j ];
			if( s1->checked || s1->tag == nullptr || strcmp( stp->tag, s1->tag ) )
				continue;
			s1->checked = 1;
			if( n_tags < 4 )
				tags[ n_tags ] = s1;
			n_tags++;
		}
		chk_tagorder( n_tags, tags );
	}
	// untagged duplexes pair up like parentheses
	std::vector<Strel *>	tstk;
	for( Strel &st : descr ){
		if( st.tag != nullptr || st.type == SYM_SS )
			continue;
		if( st.type == SYM_H5 || st.type == SYM_P5 )
			tstk.push_back( &st );
		else if( st.type == SYM_H3 || st.type == SYM_P3 ){
			if( tstk.empty() )
				note_error( "%s:%d %s element has no matching %s element.", wdfname, lineno,
					st.type == SYM_H3 ? "h3" : "p3", st.type == SYM_H3 ? "h5" : "p5" );
			else{
				Strel	*tags[ 2 ] = { tstk.back(), &st };
				tstk.pop_back();
				chk_tagorder( 2, tags );
			}
		}
	}
	for( Strel *stp : tstk )
		note_error( "%s:%d %s element has no matching %s element.", wdfname, lineno,
			stp->type == SYM_H5 ? "h5" : "h3", stp->type == SYM_H5 ? "p5" : "p3" );
	if( error )
		return;

	for( Strel &st : descr ){
		Strel	*stp = &st;
		if( stp->type == SYM_SS )
			stp->attr[ SA_PROPER ] = 1;
		else if( stp->type == SYM_H5 || stp->type == SYM_P5 ){
			if( stp->mates.empty() ){
				note_error( "%s:%d helix element has no mate.", wdfname, stp->lineno );
				continue;
			}
			if( chk_proper_nesting( stp, stp->mates[ 0 ] ) ){
				stp->attr[ SA_PROPER ] = 1;
				stp->mates[ 0 ]->attr[ SA_PROPER ] = 1;
			}
		}else if( stp->type == SYM_T1 ){
			if( stp->mates.size() < 2 ){
				note_error( "%s:%d triplex is incomplete.", wdfname, stp->lineno );
				continue;
			}
			if( !chk_proper_nesting( stp, stp->mates[ 0 ] ) ){
				note_error( "%s:%d riplex elements must be properly nested.", wdfname, lineno );
				continue;
			}
			if( chk_proper_nesting( stp->mates[ 0 ], stp->mates[ 1 ] ) ){
				stp->attr[ SA_PROPER ] = 1;
				stp->mates[ 0 ]->attr[ SA_PROPER ] = 1;
				stp->mates[ 1 ]->attr[ SA_PROPER ] = 1;
			}
		}else if( stp->type == SYM_Q1 ){
			if( stp->mates.size() < 3 ){
				note_error( "%s:%d 4-plex is incomplete.", wdfname, stp->lineno );
				continue;
			}
			if( !chk_proper_nesting( stp, stp->mates[ 0 ] ) ||
				!chk_proper_nesting( stp->mates[ 0 ], stp->mates[ 1 ] ) ){
				note_error( "%s:%d Quad elements must be properly nested.", wdfname, lineno );
				continue;
			}
			if( chk_proper_nesting( stp->mates[ 1 ], stp->mates[ 2 ] ) ){
				stp->attr[ SA_PROPER ] = 1;
				for( Strel *m : stp->mates )
					m->attr[ SA_PROPER ] = 1;
			}
		}
	}
	if( error )
		return;

	for( Strel &st : descr )
		st.checked = 0;
	for( Strel &st : descr ){
		if( st.checked )
			continue;
		if( st.type == SYM_H5 )
			find_pknots( &st );
	}
	if( error )
		return;
	std::vector<Strel *>	stk{ nullptr };
	set_scopes( 0, n_descr - 1, stk );
}

// ---------------------------------------------------------------- element parameters
bool Descriptor::chk_len_seq( int n, Strel *egroup[] )	// :1502
{
	bool	err = false;
	Strel	*stp0 = egroup[ 0 ];
	int	x_minl = UNDEF, x_maxl = UNDEF;
	for( int i = 0; i < n; i++ ){
		Strel	*stp = egroup[ i ];
		if( stp->minlen != UNDEF ){
			if( x_minl == UNDEF )
				x_minl = stp->minlen;
			else if( stp->minlen != x_minl ){
				err = true;
				note_error( "%s:%d inconsistent minlen values.", wdfname, stp->lineno );
			}
		}
		if( stp->maxlen != UNDEF ){
			if( x_maxl == UNDEF )
				x_maxl = stp->maxlen;
			else if( stp->maxlen != x_maxl ){
				err = true;
				note_error( "%s:%d inconsistent maxlen values.", wdfname, stp->lineno );
			}
		}
	}
	for( int i = 0; i < n; i++ ){
		Strel	*stp = egroup[ i ];
		if( stp->seq != nullptr ){
			stp->re = std::make_shared<ReProg>();
			if( !re_compile( stp->seq, *stp->re ) ){
				err = true;
				note_error( "%s:%d bad seq= expression '%s' (regexp error %d).", wdfname, stp->lineno,
					stp->seq, stp->re->err );
			}
		}
	}
	int	i_minl = UNDEF, i_maxl = UNDEF;
	for( int i = 0; i < n; i++ ){
		Strel	*stp = egroup[ i ];
		if( stp->seq == nullptr || !stp->re || stp->re->err )
			continue;
		int	i1_minl, i1_maxl, mmok;
		re_seqlen( *stp->re, *stp->seq == '^', &i1_minl, &i1_maxl, &mmok, &stderr_text );
		if( !mmok && stp->mismatch > 0 ){
			err = true;
			note_error( "%s:%d mismatches not allowed in this seq.", wdfname, stp->lineno );
		}
		if( i1_minl != UNDEF ){
			if( i_minl == UNDEF || i1_minl > i_minl )
				i_minl = i1_minl;
		}
		if( i1_maxl != UNDEF ){
			if( i_maxl == UNDEF )
				i_maxl = i1_maxl;
			else if( i1_maxl != i_maxl ){
				err = true;
				note_error( "%s:%d inconsistent implied max lengths.", wdfname, stp->lineno );
			}
		}
	}
	auto gid = [&]( const char *n ){ return find_id( n )->val.ival; };
	int	minl, maxl;
	if( x_minl != UNDEF )
		minl = i_minl == UNDEF ? x_minl : std::max( x_minl, i_minl );
	else if( i_minl != UNDEF )
		minl = i_minl;
	else switch( stp0->type ){
	case SYM_CTX : minl = gid( "ctx_minlen" ); break;
	case SYM_SS : minl = gid( "ss_minlen" ); break;
	case SYM_H5 : minl = gid( "wc_minlen" ); break;
	case SYM_P5 : minl = gid( "phlx_minlen" ); break;
	case SYM_T1 : minl = gid( "tr_minlen" ); break;
	case SYM_Q1 : minl = gid( "qu_minlen" ); break;
	default : minl = 1; break;
	}
	if( x_maxl != UNDEF ){
		maxl = x_maxl;
		if( i_maxl != UNDEF && x_maxl != i_maxl ){
			err = true;
			note_error( "%s:%d explicit and implicit maxlen values differ.", wdfname, stp0->lineno );
		}
	}else if( i_maxl != UNDEF )
		maxl = i_maxl;
	else switch( stp0->type ){
	case SYM_CTX : maxl = gid( "ctx_maxlen" ); break;
	case SYM_SS : maxl = gid( "ss_maxlen" ); break;
	case SYM_H5 : maxl = gid( "wc_maxlen" ); break;
	case SYM_P5 : maxl = gid( "phlx_maxlen" ); break;
	case SYM_T1 : maxl = gid( "tr_maxlen" ); break;
	case SYM_Q1 : maxl = gid( "qu_maxlen" ); break;
	default : maxl = UNBOUNDED; break;
	}
	if( minl > maxl ){
		err = true;
		note_error( "%s:%d minlen > maxlen.", wdfname, stp0->lineno );
	}
	if( !err ){
		for( int i = 0; i < n; i++ ){
			egroup[ i ]->minlen = minl;
			egroup[ i ]->maxlen = maxl;
		}
	}
	return err;
}

bool Descriptor::chk_1_strel_parms( Strel *stp )	// :1293
{
	bool	err = false;
	if( stp->mismatch == UNDEF )
		stp->mismatch = 0;
	int	stype = stp->type;
	if( stype == SYM_P3 || stype == SYM_H3 || stype == SYM_T2 || stype == SYM_T3 ||
		stype == SYM_Q2 || stype == SYM_Q3 || stype == SYM_Q4 )
		return err;
	Strel	*egroup[ 4 ];
	int	n = 0;
	egroup[ n++ ] = stp;
	for( Strel *m : stp->mates )
		if( n < 4 )
			egroup[ n++ ] = m;
	err |= chk_len_seq( n, egroup );
	if( !( stype == SYM_P5 || stype == SYM_H5 || stype == SYM_T1 || stype == SYM_Q1 ) )
		return err;

	// mispair / pairfrac
	bool	err1 = false, pfrac = false;
	Strel	*stpv = nullptr;
	for( int i = 0; i < n; i++ ){
		Strel	*s1 = egroup[ i ];
		if( s1->mispair != UNDEF ){
			if( stpv == nullptr )
				stpv = s1;
			else if( stpv->mispair != s1->mispair ){
				err1 = true;
				note_error( "%s:%d inconsistent mispair values.", wdfname, s1->lineno );
			}
		}else if( s1->pairfrac != UNDEF ){
			pfrac = true;
			if( stpv == nullptr )
				stpv = s1;
			else if( stpv->pairfrac != s1->pairfrac ){
				err1 = true;
				note_error( "%s:%d inconsistent pairfrac values.", wdfname, s1->lineno );
			}
		}
	}
	err |= err1;
	if( !err1 ){
		if( pfrac ){
			float	fval = stpv ? stpv->pairfrac : 1.;	// (sic) float in the reference
			for( int i = 0; i < n; i++ ){
				egroup[ i ]->pairfrac = fval;
				egroup[ i ]->mispair = 0;
			}
		}else{
			int	ival = stpv ? stpv->mispair : 0;
			for( int i = 0; i < n; i++ ){
				egroup[ i ]->mispair = ival;
				egroup[ i ]->pairfrac = 1.0;
			}
		}
	}

	// pair sets
	err1 = false;
	stpv = nullptr;
	for( int i = 0; i < n; i++ ){
		Strel	*s1 = egroup[ i ];
		if( s1->pairset != nullptr ){
			if( stpv == nullptr )
				stpv = s1;
			else if( !pair_equal( stpv->pairset, s1->pairset ) ){
				err1 = true;
				note_error( "%s:%d inconsistent pairset values.", wdfname, s1->lineno );
			}
		}
	}
	PairSet	*pval;
	if( stpv == nullptr ){
		const char	*dflt = ( stype == SYM_T1 ) ? "tr" : ( stype == SYM_Q1 ) ? "qu" : "wc";
		pval = ( PairSet * )find_id( dflt )->val.pval;
	}else
		pval = stpv->pairset;
	err |= err1;
	if( !err1 ){
		for( int i = 0; i < n; i++ )
			if( egroup[ i ]->pairset == nullptr )
				egroup[ i ]->pairset = pair_copy( pval );
	}

	// end rules
	err1 = false;
	stpv = nullptr;
	for( int i = 0; i < n; i++ ){
		Strel	*s1 = egroup[ i ];
		if( s1->attr[ SA_ENDS ] != UNDEF ){
			if( stpv == nullptr )
				stpv = s1;
			else if( stpv->attr[ SA_ENDS ] != s1->attr[ SA_ENDS ] ){
				err1 = true;
				note_error( "%s:%d inconsistent ends values.", wdfname, s1->lineno );
			}
		}
	}
	int	ival;
	if( stpv == nullptr ){
		const char	*nm = stype == SYM_H5 ? "wc_ends" : stype == SYM_P5 ? "phlx_ends" :
			stype == SYM_T1 ? "tr_ends" : "qu_ends";
		ival = ends2attr( ( const char * )find_id( nm )->val.pval );
	}else
		ival = stpv->attr[ SA_ENDS ];
	err |= err1;
	if( !err1 ){
		for( int i = 0; i < n; i++ )
			if( egroup[ i ]->attr[ SA_ENDS ] == UNDEF )
				egroup[ i ]->attr[ SA_ENDS ] = ( signed char )ival;
	}

	// strictness
	err1 = false;
	stpv = nullptr;
	for( int i = 0; i < n; i++ ){
		Strel	*s1 = egroup[ i ];
		if( s1->attr[ SA_STRICT ] != UNDEF ){
			if( stpv == nullptr )
				stpv = s1;
			else if( stpv->attr[ SA_STRICT ] != s1->attr[ SA_STRICT ] ){
				err1 = true;
				note_error( "%s:%d inconsistent strict values.", wdfname, s1->lineno );
			}
		}
	}
	if( stpv == nullptr ){
		const char	*nm = stype == SYM_H5 ? "wc_strict" : stype == SYM_P5 ? "phlx_strict" :
			stype == SYM_T1 ? "tr_strict" : "qu_strict";
		ival = strict2attr( find_id( nm )->val.ival );
	}else
		ival = stpv->attr[ SA_ENDS ];	// (sic) compile.c:1488 copies the ENDS attribute
	err |= err1;
	if( !err1 ){
		for( int i = 0; i < n; i++ )
			if( egroup[ i ]->attr[ SA_STRICT ] == UNDEF )
				egroup[ i ]->attr[ SA_STRICT ] = ( signed char )ival;
	}
	return err;
}

bool Descriptor::chk_strel_parms()	// :1267
{
	bool	err = false;
	for( Strel &st : descr )
		if( st.mismatch == UNDEF )
			st.mismatch = 0;
	for( Strel &st : descr )
		err |= chk_1_strel_parms( &st );
	if( lctx != nullptr ){
		if( lctx->mismatch == UNDEF )
			lctx->mismatch = 0;
		err |= chk_1_strel_parms( lctx );
	}
	if( rctx != nullptr ){
		if( rctx->mismatch == UNDEF )
			rctx->mismatch = 0;
		err |= chk_1_strel_parms( rctx );
	}
	if( !err ){	// chk_strict_helices :1665
		bool	sh = false;
		for( Strel &st : descr ){
			if( ( st.type == SYM_H5 || st.type == SYM_P5 || st.type == SYM_T1 || st.type == SYM_Q1 )
				&& st.attr[ SA_STRICT ] ){
				sh = true;
				break;
			}
		}
		args.strict_helices = sh;
	}
	return err;
}

void Descriptor::find_gi_len( int fd, int *tmin, int *tmax )	// :2926
{
	*tmin = 0;
	*tmax = 0;
	for( int d = fd; ; ){
		Strel	*stp = &descr[ d ];
		int	gmin = stp->minlen, gmax = stp->maxlen;
		int	ns = int( stp->scopes.size() );
		for( int d1 = 0; d1 < ns - 1; d1++ ){
			Strel	*s1 = stp->scopes[ d1 ], *s2 = stp->scopes[ d1 + 1 ];
			if( s1->inner ){
				int	mn3, mx3;
				find_gi_len( s1->inner->index, &mn3, &mx3 );
				s1->minilen = mn3;
				s1->maxilen = mx3;
				gmin += mn3;
				if( gmax != UNBOUNDED )
					gmax = mx3 == UNBOUNDED ? UNBOUNDED : gmax + mx3;
			}else
				s1->minilen = s1->maxilen = 0;
			gmin += s2->minlen;
			if( gmax != UNBOUNDED )
				gmax = s2->maxlen == UNBOUNDED ? UNBOUNDED : gmax + s2->maxlen;
		}
		stp->minglen = gmin;
		stp->maxglen = gmax;
		*tmin += gmin;
		if( *tmax != UNBOUNDED )
			*tmax = gmax == UNBOUNDED ? UNBOUNDED : *tmax + gmax;
		if( stp->next )
			d = stp->next->index;
		else
			break;
	}
}

void Descriptor::find_search_order( int fd )	// :3128
{
	auto add = [&]( Strel *stp ){
		stp->searchno = int( searches.size() );
		searches.push_back( stp );
	};
	for( int d = fd; ; ){
		Strel	*stp = &descr[ d ];
		switch( stp->type ){
		case SYM_SS :
			add( stp );
			break;
		case SYM_H5 :
			if( stp->attr[ SA_PROPER ] ){
				add( stp );
				if( stp->inner )
					find_search_order( stp->inner->index );
			}else{
				for( Strel *s1 : stp->scopes )
					if( s1->type == SYM_H5 )
						add( s1 );
				for( Strel *s1 : stp->scopes )
					if( s1->inner )
						find_search_order( s1->inner->index );
			}
			break;
		case SYM_P5 :
			add( stp );
			if( stp->inner )
				find_search_order( stp->inner->index );
			break;
		case SYM_T1 :
			add( stp );
			if( stp->inner )
				find_search_order( stp->inner->index );
			if( stp->scopes[ 1 ]->inner )
				find_search_order( stp->scopes[ 1 ]->inner->index );
			break;
		case SYM_Q1 :
			add( stp );
			if( stp->inner )
				find_search_order( stp->inner->index );
			if( stp->scopes[ 1 ]->inner )
				find_search_order( stp->scopes[ 1 ]->inner->index );
			if( stp->scopes[ 2 ]->inner )
				find_search_order( stp->scopes[ 2 ]->inner->index );
			break;
		case SYM_H3 : case SYM_P3 : case SYM_T2 : case SYM_T3 :
		case SYM_Q2 : case SYM_Q3 : case SYM_Q4 :
			break;
		default :
			fail( "%s:%d illegal symbol %d.", wdfname, stp->lineno, stp->type );
		}
		if( stp->next == nullptr )
			return;
		d = stp->next->index;
	}
}

void Descriptor::link()	// SE_link :776
{
	if( descr.empty() ){
		note_error( "%s:%d Descriptor has 0 elements.", wdfname, lineno );
		throw Error( stderr_text );
	}
	chk_context();
	link_tags();
	if( error )
		throw Error( stderr_text );
	if( chk_strel_parms() )
		throw Error( stderr_text );
	bool	err = false;
	for( Site &s : sites )
		err |= chk_site( s );
	if( err )
		throw Error( stderr_text );
	find_gi_len( 0, &dminlen, &dmaxlen );
	searches.clear();
	find_search_order( 0 );
	// the -O best-literal pre-filter (optimize_query :3315) only skips start
	// positions that cannot match; the scan here visits every position.
}

static int type_of( int sym )
{
	switch( sym ){
	case SYM_CTX : return RMA_T_CTX;
	case SYM_SS : return RMA_T_SS;
	case SYM_H5 : return RMA_T_H5;
	case SYM_H3 : return RMA_T_H3;
	case SYM_P5 : return RMA_T_P5;
	case SYM_P3 : return RMA_T_P3;
	case SYM_T1 : return RMA_T_T1;
	case SYM_T2 : return RMA_T_T2;
	case SYM_T3 : return RMA_T_T3;
	case SYM_Q1 : return RMA_T_Q1;
	case SYM_Q2 : return RMA_T_Q2;
	case SYM_Q3 : return RMA_T_Q3;
	case SYM_Q4 : return RMA_T_Q4;
	default : return RMA_T_SE;
	}
}

void Descriptor::to_program( rma_program_t *out )
{
	memset( out, 0, sizeof( *out ) );
	out->magic = RMA_MAGIC;
	out->size = sizeof( *out );
	out->n_elems = int( descr.size() );
	out->n_searches = int( searches.size() );
	for( size_t s = 0; s < searches.size(); s++ )
		out->searches[ s ] = searches[ s ]->index;
	out->dminlen = dminlen;
	out->dmaxlen = dmaxlen;
	Ident	*ip = find_id( "windowsize" );
	if( ip == nullptr )
		fail( "windowsize undefined." );
	if( ip->val.ival <= 0 )
		fail( "windowsize <= 0." );
	out->windowsize = ip->val.ival;
	out->strict_helices = args.strict_helices;
	ip = find_id( "chk_both_strs" );
	out->chk_both_strs = ip ? ip->val.ival : 1;

	auto add_pairset = [&]( PairSet *ps ) -> int {
		if( ps == nullptr )
			return -1;
		if( out->n_pairsets >= RMA_MAX_PAIRSETS )
			fail( "too many pair sets." );
		out->pairsets[ out->n_pairsets ] = ps->mat;
		return out->n_pairsets++;
	};
	auto cvt = [&]( const Strel &st, rma_elem_t *e ){
		e->type = type_of( st.type );
		e->proper = st.attr[ SA_PROPER ];
		e->ends = st.attr[ SA_ENDS ];
		e->strict = st.attr[ SA_STRICT ];
		e->index = st.index;
		e->searchno = st.searchno;
		e->next = st.next ? st.next->index : -1;
		e->prev = st.prev ? st.prev->index : -1;
		e->inner = st.inner ? st.inner->index : -1;
		e->outer = st.outer ? st.outer->index : -1;
		e->n_mates = int( st.mates.size() );
		for( size_t m = 0; m < st.mates.size() && m < 3; m++ )
			e->mates[ m ] = st.mates[ m ]->index;
		e->n_scopes = int( st.scopes.size() );
		if( e->n_scopes > 8 )
			fail( "%s:%d pseudoknot with more than 4 helices is not supported by this build.", wdfname, st.lineno );
		for( size_t s = 0; s < st.scopes.size(); s++ )
			e->scopes[ s ] = st.scopes[ s ]->index;
		e->scope = st.scope;
		e->minlen = st.minlen;
		e->maxlen = st.maxlen;
		e->minglen = st.minglen;
		e->maxglen = st.maxglen;
		e->minilen = st.minilen;
		e->maxilen = st.maxilen;
		e->mismatch = st.mismatch;
		e->mispair = st.mispair;
		e->pairfrac = st.pairfrac;
		e->pairset = add_pairset( st.pairset );
		e->re = -1;
		if( st.seq != nullptr && st.re ){
			if( out->n_regexes >= RMA_MAX_RE )
				fail( "too many seq= expressions." );
			std::string	why;
			if( !re_to_atoms( *st.re, *st.seq == '^', &out->regexes[ out->n_regexes ], why ) )
				fail( "%s:%d seq=\"%s\" cannot run on the device scanner: %s.", wdfname, st.lineno, st.seq, why.c_str() );
			if( st.mismatch > 0 && out->regexes[ out->n_regexes ].fixed_len < 0 )
				fail( "%s:%d mismatches not allowed in this seq.", wdfname, st.lineno );
			if( out->regexes[ out->n_regexes ].loose ){
				// (the host applies the whole expression when it replays a candidate: Replayer::one_hit -- to the elements
				// of the motif; and without mismatches where \( \) or \1 are in it: mm_advance() steps over their bytes)
				if( &st == lctx || &st == rctx )
					fail( "%s:%d seq=\"%s\" of a context element cannot run on the device scanner: back references and letters that are not acgt are taken in the motif's elements only.", wdfname, st.lineno, st.seq );
				if( st.mismatch > 0 )
					for( const ReOp &op : st.re->ops )
						if( op.kind == RE_BRA || op.kind == RE_KET || op.kind == RE_BACK )
							fail( "%s:%d seq=\"%s\" cannot run on the device scanner: mismatches in an expression with \\( \\) or a back reference.", wdfname, st.lineno, st.seq );
			}
			e->re = out->n_regexes++;
		}
	};
	for( size_t i = 0; i < descr.size(); i++ )
		cvt( descr[ i ], &out->elems[ i ] );
	out->has_lctx = lctx != nullptr;
	out->has_rctx = rctx != nullptr;
	if( lctx )
		cvt( *lctx, &out->lctx );
	if( rctx )
		cvt( *rctx, &out->rctx );
	if( sites.size() > RMA_MAX_SITES )
		fail( "too many sites." );
	out->n_sites = int( sites.size() );
	for( size_t s = 0; s < sites.size(); s++ ){
		rma_site_t	*rs = &out->sites[ s ];
		rs->n_pos = int( sites[ s ].pos.size() );
		for( int p = 0; p < rs->n_pos && p < 4; p++ ){
			rs->pos[ p ].elem = sites[ s ].pos[ p ].descr->index;
			rs->pos[ p ].l2r = sites[ s ].pos[ p ].addr.l2r;
			rs->pos[ p ].offset = sites[ s ].pos[ p ].addr.offset;
		}
		rs->pairset = add_pairset( sites[ s ].pairset );
	}
	ip = find_id( "efn_usestdbp" );
	out->efn_usestdbp = ip ? ip->val.ival : 1;
	out->efn_stdbp = add_pairset( efnstdbp );
}

std::unique_ptr<Descriptor> init_only( const Args &args )
{
	std::unique_ptr<Descriptor>	d( new Descriptor );
	d->args = args;
	init_globals( *d );
	return d;
}

std::unique_ptr<Descriptor> compile_descriptor( const Args &args, const std::string *expanded )	// rnamot.c:49-98
{
	std::unique_ptr<Descriptor>	d( new Descriptor );
	d->args = args;
	init_globals( *d );
	std::string	text;
	if( expanded != nullptr )
		text = *expanded;		// (a second descriptor from the text of the first: the files are not read again)
	else if( args.have_dfname )
		text = preprocess( *d );
	else{
		FILE	*fp = fopen( args.xdfname.c_str(), "r" );
		if( fp == nullptr )
			fail( "can't read xd-file %s.", args.xdfname.c_str() );
		char	buf[ 4096 ];
		size_t	n;
		while( ( n = fread( buf, 1, sizeof( buf ), fp ) ) > 0 )
			text.append( buf, n );
		fclose( fp );
	}
	d->expanded = text;
	if( expanded == nullptr && args.have_dfname && args.have_xdfname ){	// -xdfname keeps the expanded text
		FILE	*fp = fopen( args.xdfname.c_str(), "w" );
		if( fp == nullptr )
			fail( "RM_preprocessor: can't write temp file '%s'.", args.xdfname.c_str() );
		fwrite( text.data(), 1, text.size(), fp );
		fclose( fp );
	}
	if( !parse_descriptor( *d, text ) ){
		d->note_error( "syntax error." );
		throw Error( d->stderr_text );
	}
	if( d->error )
		throw Error( d->stderr_text );
	d->link();
	return d;
}

}	// namespace rma

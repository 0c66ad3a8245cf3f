// rm_score.cpp -- see rm_score.h.  Semantics follow /root/reference/src/score.c
// (line numbers cited per routine), including its visible quirks; the code is
// this build's own.
#include "rm_score.h"
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace rma {

namespace {

const char *opnames[ N_OP ] = {
	"halt", "noop", "acpt", "hold", "rjct", "rlse", "mrk", "cls", "fcl", "scl", "strf", "lda",
	"lod", "ldc", "sto", "and", "ior", "not", "mat", "ins", "gtr", "geq", "equ", "neq", "leq",
	"les", "add", "sub", "mul", "div", "mod", "neg", "incp", "pinc", "decp", "pdec", "fjp", "jmp" };
const char *scnames[ N_SC ] = {
	"STRID", "bits", "efn", "efn2", "length", "loc", "mismatches", "mismatches", "mismatches",
	"mispairs", "paired", "sprintf", "substr" };

char *dupstr( const char *s )
{
	char	*p = ( char * )malloc( strlen( s ) + 1 );
	strcpy( p, s );
	return p;
}

bool is_lval_sym( int s )
{
	return s == SYM_ASSIGN || s == SYM_PLUS_ASSIGN || s == SYM_MINUS_ASSIGN || s == SYM_PERCENT_ASSIGN ||
		s == SYM_SLASH_ASSIGN || s == SYM_STAR_ASSIGN || s == SYM_PLUS_PLUS || s == SYM_MINUS_MINUS;
}
bool is_rflx_sym( int s )
{
	return s == SYM_PLUS_ASSIGN || s == SYM_MINUS_ASSIGN || s == SYM_PERCENT_ASSIGN ||
		s == SYM_SLASH_ASSIGN || s == SYM_STAR_ASSIGN;
}

#define TIJ( i, j )	( ( i ) * 8 + ( j ) )

}	// namespace

ScoreVM::ScoreVM( Descriptor &d ) : d_( d )
{
	for( int p = 0; p < N_PROG; p++ )
		labtabs_[ p ].assign( 1000, 0 );
	mem_.resize( 10000 );
}

void ScoreVM::setprog( int p ) { c_prog_ = p; }

int &ScoreVM::label( int l )
{
	if( l < 0 || l >= int( labtabs_[ c_prog_ ].size() ) )
		fail( "score program has too many labels." );
	return labtabs_[ c_prog_ ][ l ];
}

int ScoreVM::newlabs( int n )
{
	int	l = nextlabs_[ c_prog_ ];
	nextlabs_[ c_prog_ ] += n;
	return l;
}

void ScoreVM::addinst( Node *np, int op, const Value *vp )	// :3511
{
	if( prog().size() >= 10000 )
		fail( "%s:%d program too big. Limit is %d", np ? np->filename : " -- No File --", np ? np->lineno : UNDEF, 10000 );
	Inst	ip;
	ip.filename = np ? np->filename : " -- No File -- ";
	ip.lineno = np ? np->lineno : UNDEF;
	ip.op = op;
	ip.val.type = T_UNDEF;
	ip.val.pval = nullptr;
	if( vp != nullptr ){
		ip.val.type = vp->type;
		switch( vp->type ){
		case T_INT :
			ip.val.ival = vp->ival;
			break;
		case T_FLOAT :
			ip.val.dval = vp->dval;
			break;
		case T_STRING :
			ip.val.pval = dupstr( ( const char * )vp->pval );
			break;
		case T_IDENT : {
			const char	*name = ( const char * )vp->pval;
			Ident	*idp = d_.find_id( name );
			if( idp == nullptr )
				idp = d_.enter_id( name, T_UNDEF, S_GLOBAL, 1, nullptr );
			else if( !idp->reinit ){
				if( op == OP_LDA || op == OP_HOLD )
					fail( "%s:%d variable '%s' is readonly.", ip.filename, ip.lineno, idp->name.c_str() );
			}
			ip.val.pval = idp;
			break;
		}
		case T_PAIRSET :
			ip.val.pval = vp->pval;
			break;
		default :
			break;
		}
	}
	prog().push_back( ip );
}

void ScoreVM::addlab( Node *np, int op, int lab )
{
	Value	v;
	v.type = T_INT;
	v.ival = v_lab_ = lab;
	addinst( np, op, &v );
}

// ---------------------------------------------------------------- statements
void ScoreVM::action( Node *np )
{
	if( np->sym == SYM_BEGIN )
		setprog( P_BEGIN );
	else if( np->sym == SYM_END )
		setprog( P_END );
	else{
		setprog( P_MAIN );
		mark();
		expr( 0, np );
		actlab_ = newlabs( 1 );
		addlab( np, OP_FJP, actlab_ );
	}
}

void ScoreVM::endaction()
{
	label( actlab_ ) = int( prog().size() );
	if( c_prog_ != P_MAIN )
		addinst( nullptr, OP_HALT, nullptr );
	setprog( P_MAIN );
}

void ScoreVM::if_( Node *np )
{
	mark();
	expr( 0, np );
	int	l = newlabs( 2 );
	ifstks_[ c_prog_ ].push_back( l );
	addlab( np, OP_FJP, l );
}

void ScoreVM::else_()
{
	int	l = ifstks_[ c_prog_ ].back();
	addlab( nullptr, OP_JMP, l + 1 );
	label( l ) = int( prog().size() );
}

void ScoreVM::endelse()
{
	label( ifstks_[ c_prog_ ].back() + 1 ) = int( prog().size() );
	ifstks_[ c_prog_ ].pop_back();
}

void ScoreVM::endif()
{
	label( ifstks_[ c_prog_ ].back() ) = int( prog().size() );
	ifstks_[ c_prog_ ].pop_back();
}

void ScoreVM::forinit( Node *np )
{
	loopstks_[ c_prog_ ].push_back( newlabs( 3 ) );
	loopincrs_[ c_prog_ ].push_back( nullptr );
	mark();
	expr( 0, np );
	clear();
}

void ScoreVM::fortest( Node *np )
{
	int	l = loopstks_[ c_prog_ ].back();
	label( l ) = int( prog().size() );
	mark();
	expr( 0, np );
	addlab( np, OP_FJP, l + 2 );
}

void ScoreVM::forincr( Node *np ) { loopincrs_[ c_prog_ ].back() = np; }

void ScoreVM::endfor()
{
	int	l = loopstks_[ c_prog_ ].back();
	label( l + 1 ) = int( prog().size() );
	mark();
	expr( 0, loopincrs_[ c_prog_ ].back() );
	clear();
	addlab( nullptr, OP_JMP, l );
	label( l + 2 ) = int( prog().size() );
	loopstks_[ c_prog_ ].pop_back();
	loopincrs_[ c_prog_ ].pop_back();
}

void ScoreVM::while_( Node *np )
{
	int	l = newlabs( 3 );
	loopstks_[ c_prog_ ].push_back( l );
	loopincrs_[ c_prog_ ].push_back( nullptr );
	label( l ) = int( prog().size() );
	mark();
	expr( 0, np );
	addlab( np, OP_FJP, l + 2 );
}

void ScoreVM::endwhile()
{
	int	l = loopstks_[ c_prog_ ].back();
	label( l + 1 ) = int( prog().size() );
	addlab( nullptr, OP_JMP, l );
	label( l + 2 ) = int( prog().size() );
	loopstks_[ c_prog_ ].pop_back();
	loopincrs_[ c_prog_ ].pop_back();
}

void ScoreVM::brk( Node *np )
{
	int	n = int( loopstks_[ c_prog_ ].size() );
	int	lev = np ? np->val.ival : 1;
	if( lev < 1 || lev > n )
		fail( "bad break level %d, must be between 1 and %d.", lev, n );
	addlab( nullptr, OP_JMP, loopstks_[ c_prog_ ][ n - lev ] + 2 );
}

void ScoreVM::cont( Node *np )
{
	int	n = int( loopstks_[ c_prog_ ].size() );
	int	lev = np ? np->val.ival : 1;
	if( lev < 1 || lev > n )
		fail( "bad continue level %d, must be between 1 and %d.", lev, n );
	// score.c:463-464 never sets the target: the jump reuses whatever label
	// the previous jump-type instruction was given
	addlab( nullptr, OP_JMP, v_lab_ );
}

void ScoreVM::accept() { addinst( nullptr, OP_ACPT, nullptr ); }
void ScoreVM::reject() { addinst( nullptr, OP_RJCT, nullptr ); }
void ScoreVM::hold( Node *np ) { addinst( np, OP_HOLD, &np->val ); }
void ScoreVM::release( Node *np ) { addinst( np, OP_RLSE, &np->val ); }
void ScoreVM::mark() { addinst( nullptr, OP_MRK, nullptr ); }
void ScoreVM::clear() { addinst( nullptr, OP_CLS, nullptr ); }

void ScoreVM::expr( int lval, Node *np )
{
	fixexpr( np );
	genexpr( lval, np );
}

// ---------------------------------------------------------------- expression rewriting
void ScoreVM::fixexpr( Node *np )	// :797
{
	if( np == nullptr )
		return;
	fixexpr( np->left );
	fixexpr( np->right );
	if( np->sym == SYM_KW_STREF )
		fix_kw_stref( np );
	else if( np->sym == SYM_IX_STREF )
		fix_ix_stref( np );
	else if( np->sym == SYM_CALL )
		fix_call( np );
}

void ScoreVM::fix_stref_common( Node *np, int sel, Node *n_id, Node *n_pos, Node *n_len )
{
	// mk_call_strid :981 : STRID( sel, id )
	Value	v;
	v.type = T_INT;
	v.ival = sel;
	Node	*args = mk_node( d_, SYM_LIST, nullptr, mk_node( d_, SYM_INT, &v, nullptr, nullptr ),
		mk_node( d_, SYM_LIST, nullptr, n_id, nullptr ) );
	Node	*call = new Node;
	call->sym = SYM_CALL;
	call->filename = np->filename;
	call->lineno = np->lineno;
	call->val.type = T_INT;
	call->val.ival = SC_STRID;
	call->right = args;
	v.ival = UNDEF;
	if( n_len == nullptr )
		n_len = mk_node( d_, SYM_INT, &v, nullptr, nullptr );
	if( n_pos == nullptr )
		n_pos = mk_node( d_, SYM_INT, &v, nullptr, nullptr );
	Node	*l3 = mk_node( d_, SYM_LIST, nullptr, n_len, nullptr );
	Node	*l2 = mk_node( d_, SYM_LIST, nullptr, n_pos, l3 );
	np->right = mk_node( d_, SYM_LIST, nullptr, call, l2 );
}

void ScoreVM::fix_kw_stref( Node *np )	// :859
{
	int	sel = np->left->sym;
	Node	*n_index = nullptr, *n_tag = nullptr, *n_pos = nullptr, *n_len = nullptr;
	for( Node *l = np->right; l; l = l->right ){
		Node	*as = l->left;
		Node	*id = as->left;
		if( id->sym != SYM_IDENT )
			continue;
		const char	*nm = ( const char * )id->val.pval;
		Node	**slot;
		if( !strcmp( nm, "index" ) ) slot = &n_index;
		else if( !strcmp( nm, "tag" ) ) slot = &n_tag;
		else if( !strcmp( nm, "pos" ) ) slot = &n_pos;
		else if( !strcmp( nm, "len" ) ) slot = &n_len;
		else
			fail( "%s:%d unknown parameter: '%s'.", id->filename, id->lineno, nm );
		if( *slot != nullptr )
			fail( "%s:%d %s parameter may not appear more than once.", id->filename, id->lineno, nm );
		*slot = as->right;
	}
	if( ( n_index == nullptr ) == ( n_tag == nullptr ) )
		fail( "%s:%d one of index= or tag= is required for stref().", np->filename, np->lineno );
	fix_stref_common( np, sel, n_index ? n_index : n_tag, n_pos, n_len );
}

void ScoreVM::fix_ix_stref( Node *np )	// :938
{
	int	sel = np->left->sym;
	Node	*l = np->right;
	Node	*n_id = l->left, *n_pos = nullptr, *n_len = nullptr;
	if( l->right ){
		l = l->right;
		n_pos = l->left;
		if( l->right )
			n_len = l->right->left;
	}
	fix_stref_common( np, sel, n_id, n_pos, n_len );
}

static int count_args( Node *np )
{
	int	n = 0;
	for( Node *l = np->right; l; l = l->right )
		n++;
	return n;
}

void ScoreVM::fix_call( Node *np )	// :1000
{
	if( np->val.type == T_INT )
		return;		// STRID calls made by fix_*_stref are already resolved
	const char	*name = ( const char * )np->val.pval;
	int	sc = UNDEF;
	for( int i = 0; i < N_SC; i++ ){
		if( !strcmp( name, scnames[ i ] ) ){
			sc = i;
			break;
		}
	}
	np->val.type = T_INT;
	np->val.ival = sc;
	auto is_stref = []( Node *n ){ return n->sym == SYM_KW_STREF || n->sym == SYM_IX_STREF; };
	int	pcnt = count_args( np );
	switch( sc ){
	case SC_STRID :
		fail( "%s:%d STRID can not be called by user.", np->filename, np->lineno );
	case SC_BITS :
	case SC_EFN :
	case SC_EFN2 : {
		if( pcnt != 2 )
			fail( "%s:%d function '%s' has the wrong number of paramters %d, takes 2", np->filename, np->lineno, scnames[ sc ], pcnt );
		Node	*a1 = np->right->left, *a2 = np->right->right->left;
		if( !is_stref( a1 ) )
			fail( "%s:%d function '%s' takes only strel arguments.", a1->filename, a1->lineno, scnames[ sc ] );
		if( !is_stref( a2 ) )
			fail( "%s:%d function '%s' takes only strel arguments.", a2->filename, a2->lineno, scnames[ sc ] );
		if( sc == SC_EFN || sc == SC_EFN2 )
			note_efn_call( np, a1, a2, sc == SC_EFN2 );
		Node	*l1 = a1->right, *tail = l1;
		while( tail->right )
			tail = tail->right;
		tail->right = a2->right;
		np->right = l1;
		break;
	}
	case SC_LENGTH :
	case SC_SPRINTF :
		break;
	case SC_LOC :
	case SC_MISPAIRS :
	case SC_PAIRED : {
		if( pcnt != 1 )
			fail( "%s:%d function '%s' has the wrong number of parameters %d, takes 1.", np->filename, np->lineno, scnames[ sc ], pcnt );
		Node	*a1 = np->right->left;
		if( sc == SC_LOC && !is_stref( a1 ) )
			fail( "%s:%d function '%s' takes only strel arguments.", a1->filename, a1->lineno, scnames[ sc ] );
		np->right = a1->right;
		break;
	}
	case SC_MISMATCHES :
		if( pcnt == 1 ){
			np->right = np->right->left->right;
			np->val.ival = SC_MISMATCHES_1;
		}else if( pcnt == 2 )
			np->val.ival = SC_MISMATCHES_2;
		else
			fail( "%s:%d function '%s' has the wrong number of parameters %d, takes 1 or 2.", np->filename, np->lineno, scnames[ sc ], pcnt );
		break;
	case SC_SUBSTR :
		if( pcnt == 2 ){
			Node	*tail = np->right;
			while( tail->right )
				tail = tail->right;
			Value	v;
			v.type = T_INT;
			v.ival = UNDEF;
			tail->right = mk_node( d_, SYM_LIST, nullptr, mk_node( d_, SYM_INT, &v, nullptr, nullptr ), nullptr );
		}else if( pcnt != 3 )
			fail( "%s:%d function '%s' has wrong number of parameters %d, takes 2 or 3.", np->filename, np->lineno, scnames[ sc ], pcnt );
		break;
	default :
		fail( "%s:%d unknown syscall %d", np->filename, np->lineno, sc );
	}
}

// The scanner evaluates efn() on the device, so the elements and positions an
// efn() call refers to must be known when the descriptor is compiled.
void ScoreVM::note_efn_call( Node *call, Node *a1, Node *a2, bool efn2 )
{
	EfnCall	ec;
	ec.call = call;
	memset( &ec.site, 0, sizeof( ec.site ) );
	ec.site.kind = efn2 ? RMA_EFN_KIND_EFN2 : RMA_EFN_KIND_EFN;
	// a?->right is LIST( STRID-call, LIST( pos, LIST( len ) ) ) after fix_*_stref
	auto konst = [&]( Node *n, bool allow_str ) -> bool {
		if( n->sym == SYM_INT )
			return true;
		if( allow_str && n->sym == SYM_STRING )
			return true;
		if( n->sym == SYM_IDENT && !strcmp( ( const char * )n->val.pval, "NSE" ) )
			return true;
		return false;
	};
	for( Node *a : { a1, a2 } ){
		Node	*strid_call = a->right->left;
		Node	*id = strid_call->right->right->left;
		Node	*pos = a->right->right->left;
		if( !konst( id, true ) || !konst( pos, false ) )
			fail( "%s:%d efn(): element and position arguments must be constants (the scan "
				"evaluates efn() on the device).", call->filename, call->lineno );
	}
	efn_calls_.push_back( ec );
}

void ScoreVM::genexpr( int lval, Node *np )	// :813
{
	if( np == nullptr )
		return;
	if( np->sym == SYM_CALL || np->sym == SYM_IN )
		addinst( np, OP_MRK, nullptr );
	genexpr( is_lval_sym( np->sym ), np->left );
	if( is_rflx_sym( np->sym ) )
		genexpr( 0, np->left );
	if( np->sym == SYM_OR || np->sym == SYM_AND ){
		int	l = newlabs( 1 );
		addnode( lval, np, l );
		genexpr( 0, np->right );
		label( l ) = int( prog().size() );
	}else{
		genexpr( 0, np->right );
		addnode( lval, np, 0 );
	}
	if( np->sym == SYM_KW_STREF || np->sym == SYM_IX_STREF )
		addinst( np, OP_STRF, nullptr );
}

void ScoreVM::addnode( int lval, Node *np, int l_andor )	// :3259
{
	Value	v;
	switch( np->sym ){
	case SYM_CALL :
		v.type = T_INT;
		v.ival = np->val.ival;
		addinst( np, OP_SCL, &v );
		break;
	case SYM_IDENT :
		addinst( np, lval ? OP_LDA : OP_LOD, &np->val );
		break;
	case SYM_INT :
	case SYM_FLOAT :
	case SYM_STRING :
	case SYM_PAIRSET :
		addinst( np, OP_LDC, &np->val );
		break;
	case SYM_DOLLAR :
		v.type = T_POS;
		v.pval = nullptr;
		addinst( np, OP_LDC, &v );
		break;
	case SYM_ASSIGN : addinst( np, OP_STO, nullptr ); break;
	case SYM_PLUS_ASSIGN : addinst( np, OP_ADD, nullptr ); addinst( np, OP_STO, nullptr ); break;
	case SYM_MINUS_ASSIGN : addinst( np, OP_SUB, nullptr ); addinst( np, OP_STO, nullptr ); break;
	case SYM_PERCENT_ASSIGN : addinst( np, OP_MOD, nullptr ); addinst( np, OP_STO, nullptr ); break;
	case SYM_STAR_ASSIGN : addinst( np, OP_MUL, nullptr ); addinst( np, OP_STO, nullptr ); break;
	case SYM_SLASH_ASSIGN : addinst( np, OP_DIV, nullptr ); addinst( np, OP_STO, nullptr ); break;
	case SYM_AND : addlab( np, OP_AND, l_andor ); break;
	case SYM_OR : addlab( np, OP_IOR, l_andor ); break;
	case SYM_NOT : addinst( np, OP_NOT, nullptr ); break;
	case SYM_EQUAL : addinst( np, OP_EQU, nullptr ); break;
	case SYM_NOT_EQUAL : addinst( np, OP_NEQ, nullptr ); break;
	case SYM_GREATER : addinst( np, OP_GTR, nullptr ); break;
	case SYM_GREATER_EQUAL : addinst( np, OP_GEQ, nullptr ); break;
	case SYM_LESS : addinst( np, OP_LES, nullptr ); break;
	case SYM_LESS_EQUAL : addinst( np, OP_LEQ, nullptr ); break;
	case SYM_MATCH : addinst( np, OP_MAT, nullptr ); break;
	case SYM_DONT_MATCH : addinst( np, OP_MAT, nullptr ); addinst( np, OP_NOT, nullptr ); break;
	case SYM_IN : addinst( np, OP_INS, nullptr ); break;
	case SYM_PLUS : addinst( np, OP_ADD, nullptr ); break;
	case SYM_MINUS : addinst( np, OP_SUB, nullptr ); break;
	case SYM_PERCENT : addinst( np, OP_MOD, nullptr ); break;
	case SYM_STAR : addinst( np, OP_MUL, nullptr ); break;
	case SYM_SLASH : addinst( np, OP_DIV, nullptr ); break;
	case SYM_NEGATE : addinst( np, OP_NEG, nullptr ); break;
	case SYM_MINUS_MINUS : addinst( np, np->left ? OP_I_MM : OP_MM_I, nullptr ); break;
	case SYM_PLUS_PLUS : addinst( np, np->left ? OP_I_PP : OP_PP_I, nullptr ); break;
	case SYM_ERROR :
		fail( "%s:%d SYM_ERROR.", np->filename, np->lineno );
	default :
		break;		// list, stref header, punctuation: no code
	}
}

// ---------------------------------------------------------------- link
void ScoreVM::linkscore()
{
	int	keep = c_prog_;
	for( int p = 0; p < N_PROG; p++ ){
		setprog( p );
		for( Inst &ip : prog() ){
			if( ip.op == OP_FJP || ip.op == OP_JMP || ip.op == OP_IOR || ip.op == OP_AND )
				ip.val.ival = label( ip.val.ival );
		}
	}
	setprog( keep );
	xdescr.clear();
	if( d_.lctx != nullptr && d_.lctx_explicit )
		xdescr.push_back( d_.lctx );
	for( Strel &st : d_.descr )
		xdescr.push_back( &st );
	if( d_.rctx != nullptr && d_.rctx_explicit )
		xdescr.push_back( d_.rctx );
	d_.find_id( "NSE" )->val.ival = int( xdescr.size() );
	slen_id_ = d_.find_id( "SLEN" );

	// resolve the efn() call sites to descriptor indices
	int	x_off = ( d_.lctx != nullptr && d_.lctx_explicit ) ? 1 : 0;
	for( EfnCall &ec : efn_calls_ ){
		// after fix_call the argument list is idx-call,pos,len,idx-call,pos,len
		Node	*l = ec.call->right;
		int	idx[ 2 ], pos[ 2 ];
		for( int k = 0; k < 2; k++ ){
			Node	*strid_call = l->left;
			int	stype = strid_call->right->left->val.ival;
			Node	*id = strid_call->right->right->left;
			Value	v;
			if( id->sym == SYM_STRING ){
				v.type = T_STRING;
				v.pval = id->val.pval;
			}else{
				v.type = T_INT;
				v.ival = id->sym == SYM_INT ? id->val.ival : int( xdescr.size() );
			}
			idx[ k ] = strid( stype, &v );
			Node	*p = l->right->left;
			pos[ k ] = p->sym == SYM_INT ? p->val.ival : int( xdescr.size() );
			l = l->right->right->right;
		}
		Strel	*s1 = xdescr[ idx[ 0 ] ], *s2 = xdescr[ idx[ 1 ] ];
		if( s1->type == SYM_CTX || s2->type == SYM_CTX )
			fail( "%s:%d efn()/efn2() only works on h5/ss/h3 elements.", ec.call->filename, ec.call->lineno );
		if( idx[ 0 ] >= idx[ 1 ] )
			fail( "%s:%d efn: bad 2nd descr index %d, must follow 1st descr %d.", ec.call->filename, ec.call->lineno, idx[ 1 ] + 1, idx[ 0 ] + 1 );
		ec.site.idx = idx[ 0 ] - x_off;
		ec.site.idx2 = idx[ 1 ] - x_off;
		// do_sc_efnx :1623-1670: pos UNDEF -> first base, pos2 UNDEF -> last base
		if( pos[ 0 ] == UNDEF )
			ec.site.pos = 0;
		else if( pos[ 0 ] <= 0 )
			fail( "%s:%d efn: bad pos1 %d, must be > 0.", ec.call->filename, ec.call->lineno, pos[ 0 ] );
		else
			ec.site.pos = pos[ 0 ] - 1;
		if( pos[ 1 ] == UNDEF )
			ec.site.pos2 = -1;
		else if( pos[ 1 ] < 0 )
			fail( "%s:%d efn: bad pos2 %d, must be > 0.", ec.call->filename, ec.call->lineno, pos[ 1 ] );
		else
			ec.site.pos2 = pos[ 1 ] - 1;
	}
}

void ScoreVM::dump( FILE *fp )	// RM_dumpscore :563, dumpinst :3576
{
	static const char	*hdr[ N_PROG ] = { "BEGIN SCORE: %4d inst.\n", "MAIN SCORE:  %4d inst.\n", "END SCORE:   %4d inst.\n" };
	for( int p = 0; p < N_PROG; p++ ){
		fprintf( fp, hdr[ p ], int( progs_[ p ].size() ) );
		int	i = 0;
		for( const Inst &ip : progs_[ p ] ){
			fprintf( fp, "%5d   %s", i++, opnames[ ip.op ] );
			const Value	&v = ip.val;
			if( ip.op == OP_LDA || ip.op == OP_LOD || ip.op == OP_HOLD || ip.op == OP_RLSE )
				fprintf( fp, " %s", ( ( Ident * )v.pval )->name.c_str() );
			else if( ip.op == OP_SCL )
				fprintf( fp, " %s", scnames[ v.ival ] );
			else if( v.type == T_INT )
				fprintf( fp, " %d", v.ival );
			else if( v.type == T_FLOAT )
				fprintf( fp, " %f", v.dval );
			else if( v.type == T_STRING )
				fprintf( fp, " \"%s\"", ( const char * )v.pval );
			else if( v.type == T_POS )
				fprintf( fp, " $" );
			else if( v.type == T_PAIRSET ){
				fprintf( fp, " { " );
				const PairSet	*ps = ( const PairSet * )v.pval;
				for( size_t k = 0; k < ps->pairs.size(); k++ ){
					fprintf( fp, "\"" );
					for( int b = 0; b < ps->pairs[ k ].n_bases; b++ )
						fprintf( fp, "%s%c", b ? ":" : "", ps->pairs[ k ].bases[ b ] );
					fprintf( fp, "\"%s", k + 1 < ps->pairs.size() ? ", " : "" );
				}
				fprintf( fp, " }" );
			}
			fprintf( fp, "\n" );
		}
	}
}

// ---------------------------------------------------------------- hit independence
// A forward data-flow pass over MAIN: per instruction the variables that are definitely assigned,
// the type every variable of interest has (the VM latches the type of an undefined variable at its
// first assignment, do_sto score.c:2234, and converts later values to it), and the types on the
// evaluation stack.  The stack is empty at every statement boundary (FJP and CLS reset it), so the
// states that meet at a label differ in the variables only -- and, inside an expression, in the top
// of the stack after && / || (joined to "unknown" when the operands' types differ).
bool ScoreVM::hit_independent( std::string *why ) const
{
	auto no = [&]( const std::string &m ){ if( why ) *why = m; return false; };
	const std::vector<Inst>	&pr = progs_[ P_MAIN ];
	if( pr.empty() )
		return true;
	enum { TY_TOP = 100, TY_MARK = 101 };		// unknown; a MRK slot
	// the variables MAIN may write
	std::vector<const Ident *>	vars;
	auto var_of = [&]( const void *p ) -> int {
		for( size_t i = 0; i < vars.size(); i++ )
			if( vars[ i ] == p )
				return int( i );
		return -1;
	};
	for( const Inst &ip : pr ){
		if( ip.op == OP_HOLD || ip.op == OP_RLSE )
			return no( "HOLD / RELEASE keep candidates across hits" );
		if( ip.op == OP_FCL )
			return no( "unimplemented instruction" );
		if( ip.op == OP_LDA && var_of( ip.val.pval ) < 0 )
			vars.push_back( ( const Ident * )ip.val.pval );
	}
	for( int p : { P_BEGIN, P_END } )
		for( const Inst &ip : progs_[ p ] )
			if( ip.op == OP_HOLD || ip.op == OP_RLSE )
				return no( "HOLD / RELEASE in BEGIN or END" );
	for( const Inst &ip : progs_[ P_END ] )
		if( ( ip.op == OP_LOD || ip.op == OP_LDA ) && var_of( ip.val.pval ) >= 0 )
			return no( "END uses a variable MAIN writes" );
	if( vars.size() > 64 )
		return no( "more than 64 variables written" );
	const int	nv = int( vars.size() );
	int	score_var = -1;
	for( int i = 0; i < nv; i++ )
		if( &vars[ i ]->val == d_.sval )
			score_var = i;
	struct State {
		bool	seen = false;
		uint64_t	da = 0;
		std::vector<int>	vt;		// type of every variable: T_UNDEF .. T_STRING or TY_TOP
		std::vector<int>	stk;		// T_* / TY_TOP / TY_MARK / T_IDENT + 1000 * ( var + 1 )
	};
	std::vector<State>	st( pr.size() + 1 );
	std::vector<int>	work;
	State	entry;
	entry.seen = true;
	entry.vt.resize( size_t( nv ) );
	for( int i = 0; i < nv; i++ )
		entry.vt[ i ] = vars[ i ]->type;		// (what BEGIN and the parms section left)
	// (a variable that is undefined on one path and of type T on the other is "undefined or T", T + 50:
	// an assignment of a T settles it either way -- the loop that assigns in its body)
	auto join_type = []( int a, int b ){
		if( a == b )
			return a;
		auto base = []( int x ){ return x >= 50 && x < 100 ? x - 50 : x; };
		const bool	var_like = ( a == T_UNDEF || ( base( a ) >= T_INT && base( a ) <= T_STRING ) ) &&
			( b == T_UNDEF || ( base( b ) >= T_INT && base( b ) <= T_STRING ) ) && a < 100 && b < 100;
		if( !var_like )
			return int( TY_TOP );
		if( a == T_UNDEF )
			return base( b ) + 50;
		if( b == T_UNDEF )
			return base( a ) + 50;
		return base( a ) == base( b ) ? base( a ) + 50 : int( TY_TOP );
	};
	std::string	trouble;
	// A variable that is undefined when MAIN starts gets its type from the first assignment of the whole run
	// (do_sto, score.c:2234-2261), whichever hit and path that is; each replay worker would latch it from its own
	// first hit.  So every assignment to such a variable, anywhere in MAIN, must store the same type.
	std::vector<int>	latched( size_t( nv ), -1 );
	auto merge = [&]( int pc, const State &s ){
		if( pc < 0 || pc > int( pr.size() ) ){
			trouble = "jump out of the program";
			return;
		}
		State	&t = st[ size_t( pc ) ];
		if( !t.seen ){
			t = s;
			t.seen = true;
			work.push_back( pc );
			return;
		}
		bool	changed = false;
		const uint64_t	da = t.da & s.da;
		if( da != t.da ){
			t.da = da;
			changed = true;
		}
		for( int i = 0; i < nv; i++ ){
			const int	j = join_type( t.vt[ i ], s.vt[ i ] );
			if( j != t.vt[ i ] ){
				t.vt[ i ] = j;
				changed = true;
			}
		}
		// After a || b (a && b) the VM leaves a below b on the path that evaluated b (do_ior score.c:2338
		// jumps over b with a alone): the stacks that meet at the label differ in depth.  What follows
		// reads the top -- an FJP, or an operator that would fail on the deeper path in any case -- so the
		// shallower stack stands for both, its top joined with the deeper one's.
		std::vector<int>	in = s.stk;
		if( t.stk.size() != in.size() ){
			if( t.stk.empty() || in.empty() ){
				trouble = "an empty evaluation stack meets a value";
				return;
			}
			const int	top_t = t.stk.back(), top_s = in.back();
			if( in.size() > t.stk.size() )
				in.resize( t.stk.size() );
			else{
				t.stk.resize( in.size() );
				changed = true;
			}
			in.back() = top_s;
			t.stk.back() = top_t;
		}
		for( size_t i = 0; i < t.stk.size(); i++ ){
			const int	j = join_type( t.stk[ i ], in[ i ] );
			if( j != t.stk[ i ] ){
				if( t.stk[ i ] == TY_MARK || in[ i ] == TY_MARK ){
					trouble = "a mark meets a value";
					return;
				}
				t.stk[ i ] = j;
				changed = true;
			}
		}
		if( changed )
			work.push_back( pc );
	};
	merge( 0, entry );
	int	rounds = 0;
	while( !work.empty() && trouble.empty() ){
		if( ++rounds > 200000 )
			return no( "analysis does not settle" );
		const int	pc = work.back();
		work.pop_back();
		if( pc >= int( pr.size() ) )
			return no( "MAIN runs off its end" );
		State	s = st[ size_t( pc ) ];
		const Inst	&ip = pr[ size_t( pc ) ];
		std::vector<int>	&k = s.stk;
		auto need = [&]( size_t n ){
			if( k.size() < n ){
				trouble = "evaluation stack underflow";
				return false;
			}
			for( size_t i = k.size() - n; i < k.size(); i++ )
				if( k[ i ] == TY_MARK ){
					trouble = "operand below a mark";
					return false;
				}
			return true;
		};
		// a value of a builtin / element reference replaces everything from the last mark on
		auto ret_to_mark = [&]( int ty ){
			int	m = int( k.size() ) - 1;
			while( m >= 0 && k[ size_t( m ) ] != TY_MARK )
				m--;
			if( m < 0 ){
				trouble = "call without a mark";
				return;
			}
			k.resize( size_t( m ) );
			k.push_back( ty );
		};
		int	next = pc + 1, jump = -1;
		switch( ip.op ){
		case OP_NOOP :
			break;
		case OP_HALT :
			return no( "HALT in MAIN" );
		case OP_ACPT :
			// the printer reads SCORE
			if( score_var >= 0 && !( ( s.da >> score_var ) & 1 ) )
				return no( "SCORE may reach the printer with an earlier hit's value" );
			next = -1;
			break;
		case OP_RJCT :
			next = -1;
			break;
		case OP_MRK :
			k.push_back( TY_MARK );
			break;
		case OP_CLS :
			k.clear();
			break;
		case OP_FJP :
			if( k.empty() ){
				trouble = "FJP on an empty stack";
				break;
			}
			k.clear();
			jump = ip.val.ival;
			break;
		case OP_JMP :
			jump = ip.val.ival;
			next = -1;
			break;
		case OP_LDA : {
			const int	v = var_of( ip.val.pval );
			k.push_back( T_IDENT + 1000 * ( v + 1 ) );
			break;
		}
		case OP_LOD : {
			const int	v = var_of( ip.val.pval );
			if( v < 0 ){
				const int	t = ( ( const Ident * )ip.val.pval )->type;	// never written by MAIN: as it is now
				k.push_back( t == T_INT || t == T_FLOAT || t == T_STRING ? t : int( TY_TOP ) );
			}else{
				if( !( ( s.da >> v ) & 1 ) )
					return no( "variable '" + vars[ size_t( v ) ]->name + "' may be read with an earlier hit's value" );
				const int	t = s.vt[ size_t( v ) ];
				k.push_back( t == T_UNDEF ? int( TY_TOP ) : t >= 50 && t < 100 ? t - 50 : t );
			}
			break;
		}
		case OP_LDC :
			switch( ip.val.type ){
			case T_INT : case T_POS : k.push_back( T_INT ); break;
			case T_FLOAT : k.push_back( T_FLOAT ); break;
			case T_STRING : k.push_back( T_STRING ); break;
			case T_PAIRSET : k.push_back( T_PAIRSET ); break;
			default : k.push_back( TY_TOP ); break;
			}
			break;
		case OP_STO : {
			if( !need( 2 ) )
				break;
			const int	top = k.back(), tm1 = k[ k.size() - 2 ];
			if( tm1 < 1000 || tm1 % 1000 != T_IDENT )
				return no( "assignment to something that is not a variable" );
			const int	v = tm1 / 1000 - 1;
			k.pop_back();
			int	&vt = s.vt[ size_t( v ) ];
			if( vars[ size_t( v ) ]->type == T_UNDEF ){
				if( top != T_INT && top != T_FLOAT && top != T_STRING )
					return no( "type of variable '" + vars[ size_t( v ) ]->name + "' depends on which hit assigns it first" );
				if( latched[ size_t( v ) ] < 0 )
					latched[ size_t( v ) ] = top;
				else if( latched[ size_t( v ) ] != top )
					return no( "type of variable '" + vars[ size_t( v ) ]->name + "' depends on which hit assigns it first" );
			}
			if( vt == T_UNDEF ){
				// the first assignment ever latches the type: every assignment must agree on it
				if( top != T_INT && top != T_FLOAT && top != T_STRING )
					return no( "type of variable '" + vars[ size_t( v ) ]->name + "' depends on which hit assigns it first" );
				vt = top;
			}else if( vt >= 50 && vt < 100 ){
				if( top != vt - 50 )
					return no( "type of variable '" + vars[ size_t( v ) ]->name + "' depends on which hit assigns it first" );
				vt = top;
			}else if( vt == TY_TOP )
				return no( "type of variable '" + vars[ size_t( v ) ]->name + "' depends on the path taken" );
			else if( top != T_INT && top != T_FLOAT && top != T_STRING )
				return no( "assignment of a value of unknown type" );
			// (do_sto leaves the slot typed like the value for int and float, like the variable otherwise;
			// what follows an assignment reads at most its int view)
			k.back() = vt == top ? vt : int( TY_TOP );
			s.da |= 1ull << v;
			break;
		}
		case OP_AND :
		case OP_IOR :
			if( !need( 1 ) )
				break;
			jump = ip.val.ival;		// (the normalised left operand is the value on that path)
			break;
		case OP_NOT :
		case OP_NEG :
			need( 1 );
			break;
		case OP_MAT :
			if( need( 2 ) ){
				k.pop_back();
				k.back() = T_INT;
			}
			break;
		case OP_INS :
			ret_to_mark( T_INT );
			break;
		case OP_GTR : case OP_GEQ : case OP_EQU : case OP_NEQ : case OP_LEQ : case OP_LES :
			if( need( 2 ) ){
				k.pop_back();
				k.back() = T_INT;
			}
			break;
		case OP_ADD : case OP_SUB : case OP_MUL : case OP_DIV : case OP_MOD :
			if( need( 2 ) ){
				k.pop_back();		// (the left operand keeps its type, do_arith)
				if( k.back() != T_INT && k.back() != T_FLOAT && k.back() != T_STRING )
					k.back() = TY_TOP;
			}
			break;
		case OP_I_PP : case OP_PP_I : case OP_I_MM : case OP_MM_I : {
			if( !need( 1 ) )
				break;
			const int	top = k.back();
			if( top < 1000 )
				return no( "increment of something that is not a variable" );
			const int	v = top / 1000 - 1;
			if( !( ( s.da >> v ) & 1 ) )
				return no( "variable '" + vars[ size_t( v ) ]->name + "' is counted from one hit to the next" );
			k.back() = TY_TOP;		// (the slot keeps its identifier type in the VM)
			break;
		}
		case OP_STRF :
			if( need( 3 ) ){
				k.pop_back();
				k.pop_back();
				k.back() = T_STRING;
			}
			break;
		case OP_SCL :
			switch( ip.val.ival ){
			case SC_BITS : case SC_EFN : case SC_EFN2 :
				ret_to_mark( T_FLOAT );
				break;
			case SC_SPRINTF : case SC_SUBSTR :
				ret_to_mark( T_STRING );
				break;
			default :
				ret_to_mark( T_INT );
				break;
			}
			break;
		default :
			return no( "instruction the analysis does not know" );
		}
		if( !trouble.empty() )
			break;
		if( jump >= 0 ){
			State	j = s;
			merge( jump, j );
		}
		if( next >= 0 )
			merge( next, s );
	}
	if( !trouble.empty() )
		return no( trouble );
	return true;
}

// ---------------------------------------------------------------- execution
Strel *ScoreVM::xd( const Inst &ip, int idx, const char *who )
{
	if( idx < 0 || idx >= int( xdescr.size() ) )
		fail( "%s:%d %s: bad descr index %d, must be between 1 and %d.", ip.filename, ip.lineno, who, idx + 1, int( xdescr.size() ) );
	return xdescr[ idx ];
}

int ScoreVM::run( int comp, int slen, const char *sbuf, Ident **h_id, const int32_t *efn_vals )
{
	if( h_id != nullptr )
		*h_id = nullptr;
	std::vector<Inst>	&pr = prog();
	if( pr.empty() )
		return SA_ACCEPT;
	sc_comp_ = comp;
	slen_id_->val.ival = sc_slen_ = slen;
	sc_sbuf_ = sbuf;
	efn_vals_ = efn_vals;
	esp_ = sp_ = mp_ = -1;
	int	rval = SA_REJECT;
	for( pc_ = 0; ; ){
		if( pc_ < 0 || pc_ >= int( pr.size() ) )
			fail( "bad pc %d, must be in 0 to %d.", pc_, int( pr.size() ) - 1 );
		if( sp_ > int( mem_.size() ) - 16 )
			fail( "score stack overflow." );
		const Inst	&ip = pr[ pc_++ ];
		switch( ip.op ){
		case OP_HALT :
			if( c_prog_ == P_MAIN )
				fail( "%s:%d pc %d, HALT", ip.filename, ip.lineno, pc_ );
			return rval;
		case OP_NOOP :
			break;
		case OP_RLSE : {	// do_rlse :1999
			Ident	*idp = ( Ident * )ip.val.pval;
			if( idp->type == T_UNDEF ){
				if( c_prog_ != P_END )
					fail( "%s:%d variable '%s' is undefined.", ip.filename, ip.lineno, idp->name.c_str() );
			}else if( idp->type == T_HIT ){
				Hit	*hp = ( Hit * )idp->val.pval;
				if( hp->def == nullptr )
					fail( "%s:%d h_def is NULL", ip.filename, ip.lineno );
				fputs( hp->def, out );
				free( hp->def );
				hp->def = nullptr;
				if( hp->match == nullptr )
					fail( "%s:%d h_match is NULL.", ip.filename, ip.lineno );
				fputs( hp->match, out );
				free( hp->match );
				hp->match = nullptr;
			}else
				fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
			break;
		}
		case OP_ACPT :
			return SA_ACCEPT;
		case OP_HOLD : {	// do_hold :2040
			Ident	*idp = ( Ident * )ip.val.pval;
			if( idp->type == T_UNDEF ){
				idp->type = T_HIT;
				idp->val.type = T_HIT;
				idp->val.pval = new Hit;
			}
			if( idp->type != T_HIT )
				fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
			Hit	*hp = ( Hit * )idp->val.pval;
			free( hp->def );
			hp->def = nullptr;
			free( hp->match );
			hp->match = nullptr;
			if( h_id != nullptr )
				*h_id = idp;
			return SA_HOLD;
		}
		case OP_RJCT :
			return rval;
		case OP_MRK :
			sp_++;
			mem_[ sp_ ].type = T_INT;
			mem_[ sp_ ].ival = mp_;
			mp_ = sp_;
			break;
		case OP_CLS :
			sp_ = mp_ = -1;
			break;
		case OP_FCL :
			fail( "%s:%d unimplemented instruction.", ip.filename, ip.lineno );
		case OP_SCL :
			do_scl( ip );
			break;
		case OP_STRF :
			do_strf( ip );
			break;
		case OP_LDA :
			sp_++;
			mem_[ sp_ ].type = T_IDENT;
			mem_[ sp_ ].pval = ip.val.pval;
			break;
		case OP_LOD : {		// do_lod :2147
			Ident	*idp = ( Ident * )ip.val.pval;
			Value	&top = mem_[ ++sp_ ];
			switch( idp->type ){
			case T_UNDEF :
				fail( "%s:%d variable '%s' is undefined.", ip.filename, ip.lineno, idp->name.c_str() );
			case T_INT :
				top.type = T_INT;
				top.ival = idp->val.ival;
				break;
			case T_FLOAT :
				top.type = T_FLOAT;
				top.dval = idp->val.dval;
				break;
			case T_STRING :
				top.type = T_STRING;
				top.pval = dupstr( ( const char * )idp->val.pval );
				break;
			default :
				fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
			}
			break;
		}
		case OP_LDC : {		// do_ldc :2189
			Value	&top = mem_[ ++sp_ ];
			switch( ip.val.type ){
			case T_INT :
				top.type = T_INT;
				top.ival = ip.val.ival;
				break;
			case T_FLOAT :
				top.type = T_FLOAT;
				top.dval = ip.val.dval;
				break;
			case T_STRING :
				top.type = T_STRING;
				top.pval = dupstr( ( const char * )ip.val.pval );
				break;
			case T_POS :
				if( esp_ < 0 )
					fail( "%s:%d '$' used outside a structure element reference.", ip.filename, ip.lineno );
				top.type = T_INT;
				top.ival = d_.descr[ estk_[ esp_ ] ].matchlen;	// (sic) rm_descr, not rm_xdescr
				break;
			case T_PAIRSET :
				top.type = T_PAIRSET;
				top.pval = ip.val.pval;
				break;
			default :
				fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
			}
			break;
		}
		case OP_STO : {		// do_sto :2234
			Value	&top = mem_[ sp_ ];
			sp_--;
			Value	&tm1 = mem_[ sp_ ];
			if( tm1.type != T_IDENT )
				fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
			Ident	*idp = ( Ident * )tm1.pval;
			switch( TIJ( idp->type, top.type ) ){
			case TIJ( T_UNDEF, T_INT ) :
				tm1.type = T_INT;
				idp->type = idp->val.type = T_INT;
				idp->val.ival = top.ival;
				break;
			case TIJ( T_UNDEF, T_FLOAT ) :
				tm1.type = T_FLOAT;
				idp->type = idp->val.type = T_FLOAT;
				idp->val.dval = top.dval;
				break;
			case TIJ( T_UNDEF, T_STRING ) :
				idp->type = idp->val.type = T_STRING;
				idp->val.pval = top.pval;	// takes over the string
				break;
			case TIJ( T_INT, T_INT ) :
				idp->val.ival = top.ival;
				break;
			case TIJ( T_INT, T_FLOAT ) :
				idp->val.ival = int( top.dval );
				break;
			case TIJ( T_FLOAT, T_INT ) :
				idp->val.dval = top.ival;
				break;
			case TIJ( T_FLOAT, T_FLOAT ) :
				idp->val.dval = top.dval;
				break;
			case TIJ( T_STRING, T_STRING ) :
				free( idp->val.pval );
				idp->val.pval = top.pval;
				break;
			default :
				fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
			}
			break;
		}
		case OP_AND :
		case OP_IOR : {		// do_and :2306, do_ior :2338
			Value	&top = mem_[ sp_ ];
			int	rv;
			switch( top.type ){
			case T_INT :
				rv = top.ival = top.ival != 0;
				break;
			case T_FLOAT :
				rv = top.dval != 0.0;
				top.ival = rv;		// (sic) type stays float
				break;
			case T_STRING : {
				char	*cp = ( char * )top.pval;
				rv = *cp != '\0';
				free( cp );
				top.ival = rv;
				break;
			}
			default :
				fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
			}
			if( ip.op == OP_AND ? !rv : rv )
				pc_ = ip.val.ival;
			break;
		}
		case OP_NOT : {		// do_not :2370
			Value	&top = mem_[ sp_ ];
			switch( top.type ){
			case T_INT :
				top.ival = !( top.ival != 0 );
				break;
			case T_FLOAT :
				top.ival = !( top.dval != 0.0 );
				break;
			case T_STRING :
				top.ival = !( *( char * )top.pval != '\0' );
				break;
			default :
				fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
			}
			break;
		}
		case OP_MAT : {		// do_mat :2397
			Value	&top = mem_[ sp_ ];
			sp_--;
			Value	&tm1 = mem_[ sp_ ];
			if( tm1.type != T_STRING || top.type != T_STRING )
				fail( "%s:%d typemismatch.", ip.filename, ip.lineno );
			char	*s = ( char * )tm1.pval, *pat = ( char * )top.pval;
			ReProg	re;
			int	rv = 0;
			if( re_compile( pat, re ) ){
				// step() runs with whatever circf compile() left: set iff the pattern starts with ^
				rv = re_step( re, s, *pat == '^' );
			}
			free( s );
			free( pat );
			tm1.type = T_INT;
			tm1.ival = rv;
			break;
		}
		case OP_INS : {		// do_ins :2432
			int	n_bases = sp_ - mp_ - 1;
			if( n_bases < 2 || n_bases > 4 )
				fail( "%s:%d pair has bad number of bases %d, requires %d-%d.", ip.filename, ip.lineno, n_bases, 2, 4 );
			Value	&top = mem_[ sp_ ];
			if( top.type != T_PAIRSET )
				fail( "%s:%d rhs of \"in\" has wrong type %d, must be of type pairset (%d).", ip.filename, ip.lineno, top.type, T_PAIRSET );
			const PairSet	*ps = ( const PairSet * )top.pval;
			const char	*sb[ 4 ];
			int	l0 = UNDEF;
			for( int i = 0; i < n_bases; i++ ){
				Value	&vb = mem_[ mp_ + 1 + i ];
				if( vb.type != T_STRING )
					fail( "%s:%d pair elements must have type string.", ip.filename, ip.lineno );
				sb[ i ] = ( const char * )vb.pval;
				int	l = int( strlen( sb[ i ] ) );
				if( l0 == UNDEF )
					l0 = l;
				else if( l != l0 )
					fail( "%s:%d all pair elements must have the same length.", ip.filename, ip.lineno );
			}
			int	rv = 1;
			for( int i = 0; i < l0 && rv; i++ ){
				int	ix = 0;
				for( int k = 0; k < n_bases; k++ )
					ix = ix * 5 + b2bc[ ( unsigned char )sb[ k ][ i ] ];
				if( n_bases == 2 )
					rv = ( ps->mat.mat2 >> ix ) & 1;
				else if( n_bases == 3 )
					rv = ( ps->mat.mat3[ ix >> 5 ] >> ( ix & 31 ) ) & 1;
				else
					rv = ( ps->mat.mat4[ ix >> 5 ] >> ( ix & 31 ) ) & 1;
				// RM_paired/RM_triple/RM_quad index ps_mat by arity: a set of
				// another arity has no such matrix
				if( ps->mat.n_bases != n_bases )
					fail( "%s:%d pair set arity does not match the number of elements.", ip.filename, ip.lineno );
			}
			for( int i = 0; i < n_bases; i++ )
				free( mem_[ mp_ + 1 + i ].pval );
			sp_ = mp_;
			mp_ = mem_[ mp_ ].ival;
			mem_[ sp_ ].type = T_INT;
			mem_[ sp_ ].ival = rv;
			break;
		}
		case OP_GTR : case OP_GEQ : case OP_EQU : case OP_NEQ : case OP_LEQ : case OP_LES :
			do_compare( ip );
			break;
		case OP_ADD : case OP_SUB : case OP_MUL : case OP_DIV : case OP_MOD :
			do_arith( ip );
			break;
		case OP_NEG : {
			Value	&top = mem_[ sp_ ];
			if( top.type == T_INT )
				top.ival = -top.ival;
			else if( top.type == T_FLOAT )
				top.dval = -top.dval;
			else
				fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
			break;
		}
		case OP_I_PP : case OP_PP_I : case OP_I_MM : case OP_MM_I :
			do_incr( ip );
			break;
		case OP_FJP :
			if( !mem_[ sp_ ].ival )
				pc_ = ip.val.ival;
			sp_ = mp_ = -1;
			break;
		case OP_JMP :
			pc_ = ip.val.ival;
			break;
		default :
			fail( "%s:%d unknown op %d.", ip.filename, ip.lineno, ip.op );
		}
	}
}

void ScoreVM::do_compare( const Inst &ip )	// do_gtr..do_les :2517-2785
{
	Value	&top = mem_[ sp_ ];
	sp_--;
	Value	&tm1 = mem_[ sp_ ];
	int	t1 = tm1.type, t2 = top.type;
	tm1.type = T_INT;
	int	c;	// sign of the comparison, or 2 for unordered
	switch( TIJ( t1, t2 ) ){
	case TIJ( T_INT, T_INT ) :
		c = tm1.ival < top.ival ? -1 : tm1.ival > top.ival;
		break;
	case TIJ( T_INT, T_FLOAT ) : {
		double	a = tm1.ival, b = top.dval;
		c = a < b ? -1 : a > b ? 1 : a == b ? 0 : 2;
		break;
	}
	case TIJ( T_FLOAT, T_INT ) : {
		double	a = tm1.dval, b = top.ival;
		c = a < b ? -1 : a > b ? 1 : a == b ? 0 : 2;
		break;
	}
	case TIJ( T_FLOAT, T_FLOAT ) : {
		double	a = tm1.dval, b = top.dval;
		c = a < b ? -1 : a > b ? 1 : a == b ? 0 : 2;
		break;
	}
	case TIJ( T_STRING, T_STRING ) : {
		char	*a = ( char * )tm1.pval, *b = ( char * )top.pval;
		int	r = strcmp( a, b );
		c = r < 0 ? -1 : r > 0;
		free( b );
		free( a );
		break;
	}
	default :
		fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
	}
	int	rv;
	switch( ip.op ){
	case OP_GTR : rv = c == 1; break;
	case OP_GEQ : rv = c == 1 || c == 0; break;
	case OP_EQU : rv = c == 0; break;
	case OP_NEQ : rv = c != 0; break;
	case OP_LEQ : rv = c == -1 || c == 0; break;
	default : rv = c == -1; break;
	}
	tm1.ival = rv;
}

void ScoreVM::do_arith( const Inst &ip )	// do_add :2787 .. do_mod :2935
{
	Value	&top = mem_[ sp_ ];
	sp_--;
	Value	&tm1 = mem_[ sp_ ];
	int	op = ip.op;
	auto dbl = [&]( double a, double b ) -> double {
		switch( op ){
		case OP_ADD : return a + b;
		case OP_SUB : return a - b;
		case OP_MUL : return a * b;
		default : return a / b;
		}
	};
	switch( TIJ( tm1.type, top.type ) ){
	case TIJ( T_INT, T_INT ) :
		switch( op ){
		case OP_ADD : tm1.ival += top.ival; break;
		case OP_SUB : tm1.ival -= top.ival; break;
		case OP_MUL : tm1.ival *= top.ival; break;
		case OP_DIV :
			if( top.ival == 0 )
				fail( "%s:%d integer division by zero.", ip.filename, ip.lineno );
			tm1.ival /= top.ival;
			break;
		default :
			if( top.ival == 0 )
				fail( "%s:%d integer division by zero.", ip.filename, ip.lineno );
			tm1.ival %= top.ival;
			break;
		}
		return;
	case TIJ( T_INT, T_FLOAT ) :
		if( op == OP_MOD )
			break;
		tm1.ival = int( dbl( tm1.ival, top.dval ) );	// the int operand keeps its type
		return;
	case TIJ( T_FLOAT, T_INT ) :
		if( op == OP_MOD )
			break;
		tm1.dval = dbl( tm1.dval, top.ival );
		return;
	case TIJ( T_FLOAT, T_FLOAT ) :
		if( op == OP_MOD )
			break;
		tm1.dval = dbl( tm1.dval, top.dval );
		return;
	case TIJ( T_STRING, T_STRING ) :
		if( op == OP_ADD ){
			char	*a = ( char * )tm1.pval, *b = ( char * )top.pval;
			char	*c = ( char * )malloc( strlen( a ) + strlen( b ) + 1 );
			strcpy( c, a );
			strcat( c, b );
			tm1.pval = c;
			free( b );
			free( a );
			return;
		}
		break;
	default :
		break;
	}
	fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
}

void ScoreVM::do_incr( const Inst &ip )	// do_i_pp :2978 .. do_mm_i :3056
{
	Value	&top = mem_[ sp_ ];
	if( top.type != T_IDENT )
		fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
	Ident	*idp = ( Ident * )top.pval;
	if( idp->type == T_UNDEF )
		fail( "%s:%d variable '%s' is undefined.", ip.filename, ip.lineno, idp->name.c_str() );
	if( idp->type != T_INT )
		fail( "%s:%d type mismatch.", ip.filename, ip.lineno );
	switch( ip.op ){
	case OP_I_PP : top.ival = idp->val.ival++; break;
	case OP_PP_I : top.ival = ++idp->val.ival; break;
	case OP_I_MM : top.ival = idp->val.ival--; break;
	default : top.ival = --idp->val.ival; break;
	}
	// (sic) the slot keeps type T_IDENT in the reference; only its int view is used
}

int ScoreVM::strid( int stype, Value *v_id )	// :1372
{
	int	idx = UNDEF;
	int	n = int( xdescr.size() );
	if( v_id->type == T_INT ){
		idx = v_id->ival;
		if( idx < 1 || idx > n )
			fail( "%s:%d index %d out of range, must be between 1 and %d.", d_.wdfname, UNDEF, idx, n );
		idx--;
		Strel	*stp = xdescr[ idx ];
		if( stype != SYM_SE && stp->type != stype )
			fail( "%s:%d descr type mismatch: have %s, need %s.", d_.wdfname, UNDEF, strel_name( stype ), strel_name( stp->type ) );
	}else if( v_id->type == T_STRING ){
		const char	*tag = ( const char * )v_id->pval;
		for( int s = 0; s < n; s++ ){
			Strel	*stp = xdescr[ s ];
			if( stp->tag == nullptr || strcmp( stp->tag, tag ) )
				continue;
			if( stp->type == stype || ( stp->type == SYM_SS && stype == SYM_SE ) ){
				idx = s;
				break;
			}
		}
		if( idx == UNDEF )
			fail( "%s:%d no such descr '%s'.", d_.wdfname, UNDEF, tag );
	}
	return idx;
}

static inline int pm2( const PairSet *ps, int b1, int b2 )
{
	return ( ps->mat.mat2 >> ( b2bc[ ( unsigned char )b1 ] * 5 + b2bc[ ( unsigned char )b2 ] ) ) & 1;
}

int ScoreVM::paired( Strel *stp, int pos, int len )	// :1421
{
	int	mlen = stp->matchlen;
	int	nm = int( stp->mates.size() );
	if( nm < 1 || nm > 3 )
		fail( "paired() does not accept descr type 'ss'." );
	Strel	*s1 = stp->index < stp->mates[ 0 ]->index ? stp : stp->mates[ 0 ];
	Strel	*s2 = s1->mates[ 0 ];
	int	p1 = s1->matchoff, p2 = s2->matchoff + mlen - 1;
	if( nm == 1 ){
		for( int i = 0; i < len; i++ )
			if( !pm2( s1->pairset, sc_sbuf_[ p1 + pos + i ], sc_sbuf_[ p2 - pos - i ] ) )
				return 0;
		return 1;
	}
	if( nm == 2 ){
		Strel	*s3 = s1->mates[ 1 ];
		int	p3 = s3->matchoff;
		for( int i = 0; i < len; i++ ){
			int	ix = ( b2bc[ ( unsigned char )sc_sbuf_[ p1 + pos + i ] ] * 5 +
				b2bc[ ( unsigned char )sc_sbuf_[ p2 - pos - i ] ] ) * 5 +
				b2bc[ ( unsigned char )sc_sbuf_[ p3 + pos + i ] ];
			if( !( ( s1->pairset->mat.mat3[ ix >> 5 ] >> ( ix & 31 ) ) & 1 ) )
				return 0;
		}
		return 1;
	}
	return 1;	// (sic) score.c:1485-1488: a failing quad also yields 1
}

float ScoreVM::do_bits( const Inst &ip )	// do_sc_bits :1502, RM_bits :3084
{
	int	idx = mem_[ sp_ - 5 ].ival;
	Strel	*stp = xd( ip, idx, "bits" );
	int	pos = mem_[ sp_ - 4 ].ival;
	if( pos == UNDEF )
		pos = 1;
	else if( pos <= 0 )
		fail( "%s:%d bits: bad pos1 %d, must be > 0.", ip.filename, ip.lineno, pos );
	else if( stp->matchlen == 0 )
		fail( "%s:%d bits: descr1 must have match len > 0.", ip.filename, ip.lineno );
	else if( pos > stp->matchlen )
		fail( "%s:%d bits: bad pos1 %d, must be <= %d.", ip.filename, ip.lineno, pos, stp->matchlen );
	pos--;
	int	idx2 = mem_[ sp_ - 2 ].ival;
	Strel	*stp2 = xd( ip, idx2, "bits" );
	if( idx >= idx2 )
		fail( "%s:%d bits: bad 2nd descr index %d, must follow 1st descr index %d.", ip.filename, ip.lineno, idx + 1, idx2 + 1 );
	int	pos2 = mem_[ sp_ - 1 ].ival;
	if( pos2 == UNDEF )
		pos2 = stp2->matchlen;
	else if( pos2 < 0 )
		fail( "%s:%d bits: bad pas2 %d, must be > 0", ip.filename, ip.lineno, pos2 );
	else if( stp2->matchlen == 0 )
		fail( "%s:%d bits: descr2 must have match len > 0.", ip.filename, ip.lineno );
	else if( pos2 > stp2->matchlen )
		fail( "%s:%d bits: bad pos2 %d, must be <= %d.", ip.filename, ip.lineno, pos2, stp2->matchlen );
	pos2--;

	int	start = stp->matchoff + pos, stop = stp2->matchoff + pos2;
	int	len = stop - start + 1;
	int	bindex[ 4 ] = { UNDEF, UNDEF, UNDEF, UNDEF }, bvec[ 4 ] = { 0, 0, 0, 0 };
	int	bn = 0;
	for( int i = start; i <= stop; i++ ){
		int	bc = b2bc[ ( unsigned char )sc_sbuf_[ i ] ];
		if( bc != RMA_BC_N ){
			int	bi = bindex[ bc ];
			if( bi == UNDEF )
				bi = bindex[ bc ] = bn++;
			bvec[ bi ]++;
		}
	}
	double	bits = 0.0;
	for( int i = 0; i < 4; i++ )
		if( bvec[ i ] != 0 )
			bits += bvec[ i ] * ( 3.32192809488736234789 * log10( 1.0 * bvec[ i ] / len ) );
	bits /= -len;
	return float( bits );
}

float ScoreVM::do_efn( const Inst &ip )	// do_sc_efnx :1567
{
	int	sc = ip.val.ival;
	const int	kind = sc == SC_EFN2 ? RMA_EFN_KIND_EFN2 : RMA_EFN_KIND_EFN;
	int	idx = mem_[ sp_ - 5 ].ival;
	Strel	*stp = xd( ip, idx, "efn" );
	int	pos = mem_[ sp_ - 4 ].ival;
	if( pos == UNDEF )
		pos = 1;
	else if( pos <= 0 )
		fail( "%s:%d efn: bad pos1 %d, must be > 0.", ip.filename, ip.lineno, pos );
	else if( stp->matchlen == 0 )
		fail( "%s:%d efn: descr1 must have a match len > 0.", ip.filename, ip.lineno );
	else if( pos > stp->matchlen )
		fail( "%s:%d efn: bad pos1 %d, must be <= %d.", ip.filename, ip.lineno, pos, stp->matchlen );
	pos--;
	int	idx2 = mem_[ sp_ - 2 ].ival;
	Strel	*stp2 = xd( ip, idx2, "efn" );
	if( idx >= idx2 )
		fail( "%s:%d efn: bad 2nd descr index %d, must follow 1st descr %d.", ip.filename, ip.lineno, idx2 + 1, idx + 1 );
	int	pos2 = mem_[ sp_ - 1 ].ival;
	if( pos2 == UNDEF )
		pos2 = stp2->matchlen;
	else if( pos2 < 0 )
		fail( "%s:%d efn: bad pos2 %d, must be > 0.", ip.filename, ip.lineno, pos2 );
	else if( stp2->matchlen == 0 )
		fail( "%s:%d efn: descr2 must have a match len > 0.", ip.filename, ip.lineno );
	else if( pos > stp2->matchlen )		// (sic) score.c:1665 tests pos, not pos2
		fail( "%s:%d efn: bad pos2 %d, must be <= %d.", ip.filename, ip.lineno, pos2, stp2->matchlen );
	pos2--;

	// The energy itself was computed by the scanner for this call site.
	int	x_off = ( d_.lctx != nullptr && d_.lctx_explicit ) ? 1 : 0;
	for( size_t k = 0; k < efn_calls_.size(); k++ ){
		const rma_efn_site_t	&s = efn_calls_[ k ].site;
		int	want2 = s.pos2 < 0 ? stp2->matchlen - 1 : s.pos2;
		if( s.kind == kind && s.idx + x_off == idx && s.idx2 + x_off == idx2 && s.pos == pos && want2 == pos2 ){
			if( efn_vals_ == nullptr )
				fail( "%s:%d efn(): no device energies for this candidate.", ip.filename, ip.lineno );
			float	rval = 0.01 * efn_vals_[ k ];	// score.c:1675-1677: float rval = 0.01 * RM_efn() / RM_efn2()
			return rval;
		}
	}
	fail( "%s:%d efn(): call site was not registered with the scanner.", ip.filename, ip.lineno );
}

void ScoreVM::do_sprintf( const Inst &ip, std::string &outbuf )	// do_sc_sprintf :1684, fmt_1_item :1710
{
	int	n_args = sp_ - mp_;
	if( n_args < 1 || mem_[ mp_ + 1 ].type != T_STRING )
		fail( "%s:%d sprintf: first argument must be a format string.", ip.filename, ip.lineno );
	const char	*fstr = ( const char * )mem_[ mp_ + 1 ].pval;
	int	c_arg = 0;
	const char	*fp = fstr;
	outbuf.clear();
	for( const char *pp; ( pp = strchr( fp, '%' ) ) != nullptr; ){
		outbuf.append( fp, pp - fp );
		const char	*epp = strpbrk( pp + 1, "bBdiouxXfeEgGcCsSpn%" );
		if( epp == nullptr )
			fail( "%s:%d sprintf: bad format '%s'.", ip.filename, ip.lineno, pp );
		std::string	fmt( pp, epp - pp + 1 );
		char	type = *epp;
		if( fmt.find( '*' ) != std::string::npos || fmt.find( '$' ) != std::string::npos )
			fail( "%s:%d sprintf: '*' and '$' in formats are not supported by this build.", ip.filename, ip.lineno );
		int	r_arg = c_arg + 1;
		if( r_arg < 1 || r_arg >= n_args ){
			d_.note_error( "%s:%d No such argument %d.", ip.filename, ip.lineno, r_arg );
			fputs( d_.stderr_text.c_str(), stderr );
			d_.stderr_text.clear();
			break;
		}
		Value	&v = mem_[ mp_ + r_arg + 1 ];
		c_arg += 1;
		char	buf[ 4096 ];
		bool	bad = false;
		buf[ 0 ] = '\0';
		switch( type ){
		case 'e' : case 'E' : case 'f' : case 'g' : case 'G' :
			if( v.type != T_FLOAT ){
				d_.note_error( "%s:%d '%c' format requires float arg.", ip.filename, ip.lineno, type );
				bad = true;
			}else
				snprintf( buf, sizeof( buf ), fmt.c_str(), v.dval );
			break;
		case 'd' : case 'i' :
			if( v.type != T_INT ){
				d_.note_error( "%s:%d '%c' format requires int arg.", ip.filename, ip.lineno, type );
				bad = true;
			}else
				snprintf( buf, sizeof( buf ), fmt.c_str(), v.ival );
			break;
		case 'o' : case 'u' : case 'x' : case 'X' :
			if( v.type != T_INT ){
				d_.note_error( "%s:%d '%c' format requires int arg.", ip.filename, ip.lineno, type );
				bad = true;
			}else
				snprintf( buf, sizeof( buf ), fmt.c_str(), ( unsigned )v.ival );
			break;
		case 's' :
			if( v.type != T_STRING ){
				d_.note_error( "%s:%d '%c' format requires string/seq arg.", ip.filename, ip.lineno, type );
				bad = true;
			}else
				snprintf( buf, sizeof( buf ), fmt.c_str(), ( const char * )v.pval );
			break;
		case 'n' :
			if( v.type == T_INT )
				v.ival = 0;
			else if( v.type == T_FLOAT )
				v.dval = 0;
			else{
				d_.note_error( "%s:%d '%c' format requires int arg.", ip.filename, ip.lineno, type );
				bad = true;
			}
			break;
		case '%' :
			strcpy( buf, "%" );
			break;
		default :
			d_.note_error( "%s:%d '%c' unsupported format.", ip.filename, ip.lineno, type );
			bad = true;
			break;
		}
		if( bad ){
			fputs( d_.stderr_text.c_str(), stderr );
			d_.stderr_text.clear();
			break;
		}
		outbuf += buf;
		fp = epp + 1;
	}
	// strcpy( sbp, fp ), score.c:1706: after a failed directive fp still points
	// at the start of the segment that held it, so that text appears twice
	outbuf += fp;
}

void ScoreVM::do_strf( const Inst &ip )	// :2082
{
	int	len = mem_[ sp_ ].ival, pos = mem_[ sp_ - 1 ].ival, index = mem_[ sp_ - 2 ].ival;
	if( index < 0 || index >= int( xdescr.size() ) )
		fail( "%s:%d no such descr %d.", ip.filename, ip.lineno, index );
	Strel	*stp = xdescr[ index ];
	if( pos == UNDEF )
		pos = 1;
	else if( pos < 0 )
		fail( "%s:%d bad pos %d, must be > 0.", ip.filename, ip.lineno, pos );
	else if( stp->matchlen == 0 )
		pos = 1;
	else if( pos > stp->matchlen )
		fail( "%s:%d bad pos %d, must be <= %d.", ip.filename, ip.lineno, pos, stp->matchlen );
	pos--;
	if( len == 0 )
		fail( "%s:%d bad len %d, must be > 0.", ip.filename, ip.lineno, len );
	else if( len == UNDEF )
		len = stp->matchlen - pos;
	else
		len = std::min( stp->matchlen - pos, len );
	if( len < 0 )
		len = 0;
	char	*cp = ( char * )malloc( len + 1 );
	memcpy( cp, sc_sbuf_ + stp->matchoff + pos, len );
	cp[ len ] = '\0';
	sp_ -= 2;
	mem_[ sp_ ].type = T_STRING;
	mem_[ sp_ ].pval = cp;
	esp_--;
}

void ScoreVM::do_scl( const Inst &ip )	// :1138
{
	auto ret_int = [&]( int v ){
		sp_ = mp_;
		mp_ = mem_[ mp_ ].ival;
		mem_[ sp_ ].type = T_INT;
		mem_[ sp_ ].ival = v;
	};
	switch( ip.val.ival ){
	case SC_STRID : {
		Value	*v_id = &mem_[ sp_ ];
		int	stype = mem_[ sp_ - 1 ].ival;
		int	idx = strid( stype, v_id );
		if( v_id->type == T_STRING )
			free( v_id->pval );
		ret_int( idx );
		if( esp_ + 1 >= 20 )
			fail( "%s:%d element stack overflow.", ip.filename, ip.lineno );
		estk_[ ++esp_ ] = idx;
		break;
	}
	case SC_BITS : {
		float	rv = do_bits( ip );
		sp_ = mp_;
		mp_ = mem_[ mp_ ].ival;
		mem_[ sp_ ].type = T_FLOAT;
		mem_[ sp_ ].dval = rv;
		break;
	}
	case SC_EFN :
	case SC_EFN2 : {
		float	rv = do_efn( ip );
		sp_ = mp_;
		mp_ = mem_[ mp_ ].ival;
		mem_[ sp_ ].type = T_FLOAT;
		mem_[ sp_ ].dval = rv;
		break;
	}
	case SC_LENGTH : {
		if( mem_[ sp_ ].type != T_STRING )
			fail( "%s:%d length: argument must be a string.", ip.filename, ip.lineno );
		char	*cp = ( char * )mem_[ sp_ ].pval;
		int	len = int( strlen( cp ) );
		free( cp );
		ret_int( len );
		break;
	}
	case SC_LOC : {
		int	idx = mem_[ sp_ - 2 ].ival;
		if( idx < 0 || idx >= int( xdescr.size() ) )
			fail( "%s:%d descr index %d is out of range; must be between 1 and %d.", ip.filename, ip.lineno, idx + 1, int( xdescr.size() ) );
		Strel	*stp = xdescr[ idx ];
		int	pos = mem_[ sp_ - 1 ].ival;
		if( pos == UNDEF )
			pos = 1;
		else if( pos < 0 )
			fail( "%s:%d loc: bad pos %d, must be > 0.", ip.filename, ip.lineno, pos );
		else if( stp->matchlen == 0 )
			fail( "%s:%d loc: bad matchlen %d, must be > 0.", ip.filename, ip.lineno, stp->matchlen );
		else if( pos > stp->matchlen )
			fail( "%s:%d loc: bad pos %d, must be <= %d.", ip.filename, ip.lineno, pos, stp->matchlen );
		ret_int( stp->matchoff + 1 );
		break;
	}
	case SC_MISMATCHES_1 : {
		Strel	*stp = xd( ip, mem_[ sp_ - 2 ].ival, "mismatches" );
		ret_int( stp->n_mismatches );
		break;
	}
	case SC_MISMATCHES_2 : {
		char	*pp = ( char * )mem_[ sp_ ].pval, *cp = ( char * )mem_[ sp_ - 1 ].pval;
		ReProg	re;
		int	n_mm = 0;
		if( re_compile( pp, re ) )
			re_mm_step( re, cp, *pp == '^', 20 * int( strlen( pp ) ), &n_mm );
		free( pp );
		free( cp );
		ret_int( n_mm );
		break;
	}
	case SC_MISPAIRS : {
		Strel	*stp = xd( ip, mem_[ sp_ - 2 ].ival, "mispairs" );
		ret_int( stp->n_mispairs );
		break;
	}
	case SC_PAIRED : {
		Strel	*stp = xd( ip, mem_[ sp_ - 2 ].ival, "paired" );
		int	pos = mem_[ sp_ - 1 ].ival;
		if( pos < 1 || pos > stp->matchlen )
			fail( "%s:%d paired: bad pos %d, must be between 1 and %d.", ip.filename, ip.lineno, pos, stp->matchlen );
		pos--;
		int	len = mem_[ sp_ ].ival;
		if( len == 0 )
			fail( "%s:%d paired: bad len %d, must be > 0.", ip.filename, ip.lineno, len );
		else if( len < 0 )
			len = stp->matchlen - pos;
		else
			len = std::min( stp->matchlen - pos, len );
		ret_int( paired( stp, pos, len ) );
		break;
	}
	case SC_SPRINTF : {
		std::string	s;
		do_sprintf( ip, s );
		for( int i = sp_; i > mp_; i-- )
			if( mem_[ i ].type == T_STRING )
				free( mem_[ i ].pval );
		sp_ = mp_;
		mp_ = mem_[ mp_ ].ival;
		mem_[ sp_ ].type = T_STRING;
		mem_[ sp_ ].pval = dupstr( s.c_str() );
		break;
	}
	case SC_SUBSTR : {
		char	*cp = ( char * )mem_[ sp_ - 2 ].pval;
		int	c_len = int( strlen( cp ) );
		int	pos = mem_[ sp_ - 1 ].ival, len = mem_[ sp_ ].ival;
		if( pos < 1 || pos > c_len )
			fail( "%s:%d substr: bad positiion %d, must be between 1 and %d.", ip.filename, ip.lineno, pos, c_len );
		if( len < 1 )
			fail( "%s:%d substr: bad len %d, must be >= 1.", ip.filename, ip.lineno, len );
		len = std::min( c_len - pos + 1, len );
		char	*ssp = ( char * )malloc( len + 1 );
		memcpy( ssp, cp + pos - 1, len );
		ssp[ len ] = '\0';
		free( cp );
		sp_ = mp_;
		mp_ = mem_[ mp_ ].ival;
		mem_[ sp_ ].type = T_STRING;
		mem_[ sp_ ].pval = ssp;
		break;
	}
	default :
		fail( "%s:%d undefined syscall %d", ip.filename, ip.lineno, ip.val.ival );
	}
}

// ---------------------------------------------------------------- print_match
void HitPrinter::header( FILE *fp ) const
{
	fprintf( fp, "#RM scored\n" );
	fprintf( fp, "#RM descr" );
	auto name = [&]( const Strel *st ){
		fprintf( fp, " %s", strel_name( st->type ) );
		if( st->tag != nullptr ){
			std::string	c;
			for( const char *p = st->tag; *p; p++ ){
				if( *p == '"' || *p == '\\' )
					c += '\\';
				c += *p;
			}
			fprintf( fp, "(tag='%s')", c.c_str() );
		}
	};
	if( d_.lctx )
		name( d_.lctx );
	for( const Strel &st : d_.descr )
		name( &st );
	if( d_.rctx )
		name( d_.rctx );
	fprintf( fp, "\n" );
	fprintf( fp, "#RM dfile %s\n", d_.args.have_dfname ? d_.args.dfname.c_str() : "(null)" );
}

void HitPrinter::print( const char *sid, const char *sdef, int comp, int slen, const char *sbuf, Ident *h_id )
{
	std::string	defline = std::string( ">" ) + sid + " " + sdef + "\n";
	int	len = 0;
	for( const Strel &st : d_.descr )
		len += st.matchlen;
	int	offset = comp ? slen - d_.descr[ 0 ].matchoff : d_.descr[ 0 ].matchoff + 1;
	std::string	hit;
	char	buf[ 2048 ];
	snprintf( buf, sizeof( buf ), "%-12s", sid );
	hit += buf;
	const Value	*sv = d_.sval;
	switch( sv->type ){
	case T_INT :
		snprintf( buf, sizeof( buf ), " %8d", sv->ival );
		break;
	case T_FLOAT :
		snprintf( buf, sizeof( buf ), " %8.3lf", sv->dval );
		break;
	case T_STRING :
		snprintf( buf, sizeof( buf ), " %8s", ( const char * )sv->pval );
		break;
	default :
		snprintf( buf, sizeof( buf ), " %8.3lf", 0.0 );
		break;
	}
	hit += buf;
	snprintf( buf, sizeof( buf ), " %d %7d %4d", comp, offset, len );
	hit += buf;
	auto put = [&]( const Strel *st ){
		if( st->matchlen > 0 ){
			hit += ' ';
			hit.append( sbuf + st->matchoff, st->matchlen );
		}else
			hit += " .";
	};
	if( d_.lctx )
		put( d_.lctx );
	for( const Strel &st : d_.descr )
		put( &st );
	if( d_.rctx )
		put( d_.rctx );
	hit += '\n';

	if( first_ ){
		first_ = false;
		header( out_ );
	}
	if( h_id != nullptr ){
		Hit	*hp = ( Hit * )h_id->val.pval;
		hp->def = dupstr( defline.c_str() );
		hp->match = dupstr( hit.c_str() );
	}else{
		fputs( defline.c_str(), out_ );
		fputs( hit.c_str(), out_ );
	}
}

}	// namespace rma

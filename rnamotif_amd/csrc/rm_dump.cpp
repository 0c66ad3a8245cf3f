// rm_dump.cpp -- the diagnostic listings of the command line program:
//   -s   global symbols after initialisation            (rnamot.c:60-63)
//   -d   symbols, structure elements, sites             (rnamot.c:101-103)
//   -h   element hierarchy and search order             (rnamot.c:101-103)
// Layout follows RM_dump() and friends, /root/reference/src/dump.c:34-841, so that
// scripts reading the reference's stderr listings keep working.  The start/stop
// columns are the reference's find_limits() (compile.c:2984-3126), which nothing
// but these listings reads; they are computed here on demand.
#include "rm_host.h"
#include <cstring>

namespace rma {

namespace {

bool r2l( int t )	// RM_R2L, compile.c:42-43
{
	return t == SYM_P3 || t == SYM_H3 || t == SYM_T2 || t == SYM_Q2 || t == SYM_Q4;
}

struct Limits { Addr start, stop; };

struct Dumper {
	Descriptor	&d;
	FILE	*fp;
	std::vector<Limits>	lim;

	Dumper( Descriptor &d_, FILE *fp_ ) : d( d_ ), fp( fp_ ), lim( d_.descr.size() ) {}

	// ---- compile.c:3073-3126
	int	min_prefixlen( const Strel *stp )
	{
		if( stp->scope == UNDEF )
			return 0;
		int	plen = 0;
		for( int s = stp->index - 1; s >= stp->scopes[ 0 ]->index; s-- )
			plen += d.descr[ s ].minlen;
		return plen;
	}
	int	max_prefixlen( const Strel *stp )
	{
		if( stp->scope == UNDEF )
			return 0;
		int	plen = 0;
		for( int s = stp->index - 1; s >= stp->scopes[ 0 ]->index; s-- ){
			if( d.descr[ s ].maxlen == UNBOUNDED )
				return UNBOUNDED;
			plen += d.descr[ s ].maxlen;
		}
		return plen;
	}
	int	min_suffixlen( const Strel *stp )
	{
		int	slen = 0;
		if( stp->scope == UNDEF ){
			for( const Strel *p = stp->next; p; p = p->next )
				slen += p->scope == 0 ? p->minglen : p->minlen;
			return slen;
		}
		for( int s = stp->index + 1; s <= stp->scopes.back()->index; s++ )
			slen += d.descr[ s ].minlen;
		for( const Strel *p = stp->scopes[ 0 ]->next; p; p = p->next )
			slen += p->minlen;
		return slen;
	}
	void	find_1_limit( const Strel *stp )	// compile.c:3007-3057
	{
		Limits	&l = lim[ stp->index ];
		if( stp->scope == UNDEF || stp->scope == 0 ){
			l.start.l2r = 1;
			l.start.offset = 0;
		}else if( r2l( stp->type ) ){
			if( stp->maxlen == UNBOUNDED || max_prefixlen( stp ) == UNBOUNDED ){
				l.start.offset = min_suffixlen( stp );
				l.start.l2r = 0;
			}else{
				l.start.offset = max_prefixlen( stp ) + stp->maxlen - 1;
				l.start.l2r = 1;
			}
		}else{
			l.start.offset = min_prefixlen( stp );
			l.start.l2r = 1;
		}
		if( r2l( stp->type ) ){
			l.stop.offset = stp->minlen + min_prefixlen( stp ) - 1;
			l.stop.l2r = 1;
		}else{
			l.stop.offset = stp->minlen + min_suffixlen( stp );
			if( l.stop.offset > 0 )
				l.stop.offset--;
			l.stop.l2r = 0;
		}
	}
	void	find_limits( int fd )			// compile.c:2984-3005
	{
		for( int dd = fd; ; ){
			const Strel	*stp = &d.descr[ dd ];
			find_1_limit( stp );
			for( size_t s = 1; s < stp->scopes.size(); s++ ){
				const Strel	*a = stp->scopes[ s - 1 ], *b = stp->scopes[ s ];
				if( a->index + 1 < b->index )
					find_limits( a->index + 1 );
				find_1_limit( b );
			}
			if( stp->next == nullptr )
				return;
			dd = stp->next->index;
		}
	}

	// ---- dump.c
	void	pairset( const PairSet *ps )		// RM_dump_pairset :208, RM_dump_pair :224
	{
		fprintf( fp, "{ " );
		if( ps != nullptr ){
			for( size_t i = 0; i < ps->pairs.size(); i++ ){
				// (the reference writes the bases to stderr whatever fp is, dump.c:230-232;
				// the listings only ever go to stderr)
				fprintf( fp, "\"" );
				for( int b = 0; b < ps->pairs[ i ].n_bases; b++ )
					fprintf( fp, "%c%s", ps->pairs[ i ].bases[ b ], b < ps->pairs[ i ].n_bases - 1 ? ":" : "" );
				fprintf( fp, "\"" );
				if( i + 1 < ps->pairs.size() )
					fprintf( fp, ", " );
			}
		}
		fprintf( fp, " }" );
	}
	void	ident( const Ident *ip, int fmt )	// RM_dump_id :95-206
	{
		static const char	*tname[] = { "UNDEF", "INT", "FLOAT", "STRING", "PAIR" };
		static const char	*sname[] = { "UNDEF", "GLOBAL", "STREL", "SITE" };
		if( fmt == 1 )
			fprintf( fp, "%s (%s) = {\n", ip->name.c_str(), ip->reinit ? "RW" : "RO" );
		else
			fprintf( fp, "%-16s (%s) = ", ip->name.c_str(), ip->reinit ? "RW" : "RO" );
		if( fmt == 1 ){
			fprintf( fp, "\ttype  = " );
			if( ip->type >= T_UNDEF && ip->type <= T_PAIRSET )
				fprintf( fp, "%s\n", tname[ ip->type ] );
			else if( ip->type == T_IDENT )
				fprintf( fp, "IDENT\n" );
			else
				fprintf( fp, "-- BAD type %d\n", ip->type );
			fprintf( fp, "\tclass = VAR\n" );	// every RM_enter_id() passes C_VAR
			fprintf( fp, "\tscope = " );
			if( ip->scope >= 0 && ip->scope <= S_SITE )
				fprintf( fp, "%s\n", sname[ ip->scope ] );
			else
				fprintf( fp, "-- BAD scope %d\n", ip->scope );
			fprintf( fp, "\treinit= %d\n", ip->reinit );
			fprintf( fp, "\tvalue = " );
		}
		switch( ip->val.type ){
		case T_UNDEF : fprintf( fp, "UNDEF\n" ); break;
		case T_INT : fprintf( fp, "%d\n", ip->val.ival ); break;
		case T_FLOAT : fprintf( fp, "%lg\n", ip->val.dval ); break;
		case T_STRING :
			fprintf( fp, "'%s'\n", ip->val.pval ? ( const char * )ip->val.pval : "NULL" );
			break;
		case T_PAIRSET :
			pairset( ( const PairSet * )ip->val.pval );
			fprintf( fp, "\n" );
			break;
		case T_IDENT : fprintf( fp, "IDENT?\n" ); break;
		default : fprintf( fp, "-- BAD type %d\n", ip->val.type ); break;
		}
		if( fmt == 1 )
			fprintf( fp, "}\n" );
	}
	void	len_pair( const char *label, int lo, int hi )
	{
		auto one = [&]( int v ){
			if( v == UNDEF )
				fprintf( fp, "UNDEF" );
			else if( v == UNBOUNDED )
				fprintf( fp, "UNBOUNDED" );
			else
				fprintf( fp, "%d", v );
		};
		fprintf( fp, "\t%s = ", label );
		one( lo );
		fprintf( fp, ":" );
		one( hi );
		fprintf( fp, "\n" );
	}
	void	addr( const char *label, const Addr &a )
	{
		fprintf( fp, "\t%s = ", label );
		if( a.offset == UNDEF )
			fprintf( fp, "UNDEF" );
		else if( a.l2r )
			fprintf( fp, "%d", a.offset );
		else
			fprintf( fp, "$-%d", a.offset );
		fprintf( fp, "\n" );
	}
	static std::string	attr2str( const signed char attr[] )	// :650-693
	{
		std::string	s = "{ ";
		int	n = 0;
		auto add = [&]( const char *w ){
			if( n++ > 0 )
				s += ",";
			s += w;
		};
		if( attr[ SA_PROPER ] ) add( "P" );
		if( attr[ SA_ENDS ] & RMA_5PAIRED ) add( "p5" );
		if( attr[ SA_ENDS ] & RMA_3PAIRED ) add( "p3" );
		if( attr[ SA_STRICT ] & RMA_5STRICT ) add( "s5" );
		if( attr[ SA_STRICT ] & RMA_3STRICT ) add( "s3" );
		return s + " }";
	}
	void	link( const char *label, const Strel *p, bool nl_inside )
	{
		fprintf( fp, "\t%s = ", label );
		if( p != nullptr )
			fprintf( fp, "%d", p->index );
		else
			fprintf( fp, "(None)" );
		fprintf( fp, "\n" );
		( void )nl_inside;
	}
	void	list( const char *label, const std::vector<Strel *> &v )
	{
		fprintf( fp, "\t%s = [ ", label );
		for( size_t i = 0; i < v.size(); i++ )
			fprintf( fp, "%d%s", v[ i ]->index, i + 1 < v.size() ? ", " : "" );
		fprintf( fp, " ]\n" );
	}
	void	strel( const Strel *stp, const Limits &l )	// RM_dump_descr :299-503
	{
		fprintf( fp, "descr[%3d] = {\n", stp->index );
		const char	*nm = strel_name( stp->type );
		if( nm != nullptr && *nm )
			fprintf( fp, "\ttype     = %s\n", nm );
		else
			fprintf( fp, "\ttype     = unknown (%d)\n", stp->type );
		fprintf( fp, "\tattr     = %s\n", attr2str( stp->attr ).c_str() );
		fprintf( fp, "\tlineno   = %d\n", stp->lineno );
		if( stp->searchno == UNDEF )
			fprintf( fp, "\tsearchno = UNDEF\n" );
		else
			fprintf( fp, "\tsearchno = %d\n", stp->searchno );
		fprintf( fp, "\ttag      = '%s'\n", stp->tag ? stp->tag : "(No tag)" );
		link( "next    ", stp->next, false );
		link( "prev    ", stp->prev, false );
		link( "inner   ", stp->inner, false );
		link( "outer   ", stp->outer, false );
		list( "mates   ", stp->mates );
		list( "scopes  ", stp->scopes );
		fprintf( fp, "\tscope    = %d\n", stp->scope );
		len_pair( "len     ", stp->minlen, stp->maxlen );
		len_pair( "glen    ", stp->minglen, stp->maxglen );
		len_pair( "ilen    ", stp->minilen, stp->maxilen );
		addr( "start   ", l.start );
		addr( "stop    ", l.stop );
		fprintf( fp, "\tseq      = '%s'\n", stp->seq ? stp->seq : "(No seq)" );
		if( stp->mismatch == UNDEF )
			fprintf( fp, "\tmismatch = UNDEF\n" );
		else
			fprintf( fp, "\tmismatch = %d\n", stp->mismatch );
		if( stp->matchfrac == UNDEF )
			fprintf( fp, "\tmatchfrac= UNDEF\n" );
		else
			fprintf( fp, "\tmatchfrac= %5.3lf\n", stp->matchfrac );
		if( stp->mispair == UNDEF )
			fprintf( fp, "\tmispair  = UNDEF\n" );
		else
			fprintf( fp, "\tmispair  = %d\n", stp->mispair );
		if( stp->pairfrac == UNDEF )
			fprintf( fp, "\tpairfrac = UNDEF\n" );
		else
			fprintf( fp, "\tpairfrac = %5.3lf\n", stp->pairfrac );
		fprintf( fp, "\tpair     = " );
		if( stp->pairset != nullptr )
			pairset( stp->pairset );
		else
			fprintf( fp, "(None)" );
		fprintf( fp, "\n}\n" );
	}
	void	sites()						// RM_dump_sites :575-598, RM_dump_pos :505-573
	{
		fprintf( fp, "SITES: %4d sites.\n", int( d.sites.size() ) );
		for( size_t i = 0; i < d.sites.size(); i++ ){
			const Site	&sp = d.sites[ i ];
			fprintf( fp, "site[%2d] = {\n", int( i ) + 1 );
			fprintf( fp, "\tnpos    = %3d\n", int( sp.pos.size() ) );
			for( size_t j = 0; j < sp.pos.size(); j++ ){
				const Pos	&p = sp.pos[ j ];
				fprintf( fp, "\tpos[%2d] = {\n", int( j ) + 1 );
				const char	*nm = strel_name( p.type );
				if( nm != nullptr && *nm && p.type != SYM_CTX )
					fprintf( fp, "\t\ttype     = %s\n", nm );
				else
					fprintf( fp, "\t\ttype     = unknown (%d)\n", p.type );
				fprintf( fp, "\t\tlineno   = %d\n", p.lineno );
				fprintf( fp, "\t\ttag      = '%s'\n", p.tag ? p.tag : "(No tag)" );
				fprintf( fp, "\t\tdindex   = %d\n", p.descr ? p.descr->index : UNDEF );
				fprintf( fp, "\t\tl2r      = %s\n", p.addr.l2r ? "TRUE" : "FALSE" );
				fprintf( fp, "\t\toffset   = %d\n", p.addr.offset );
				fprintf( fp, "\t}\n" );
			}
			fprintf( fp, "\tpairset = " );
			pairset( sp.pairset );
			fprintf( fp, "\n}\n" );
		}
	}

	// ---- hierarchy listing, dump.c:695-841
	void	one_element( const std::string &prefix, const Strel *stp )	// print_1_element :729-789
	{
		char	buf[ 64 ];
		std::string	line;
		auto num = [&]( int v, bool undef_word, bool unb_word ){
			if( unb_word && v == UNBOUNDED )
				snprintf( buf, sizeof( buf ), " UNBND" );
			else if( undef_word && v == UNDEF )
				snprintf( buf, sizeof( buf ), " UNDEF" );
			else
				snprintf( buf, sizeof( buf ), " %5d", v );
			line += buf;
		};
		snprintf( buf, sizeof( buf ), "%4d", stp->index );
		line += buf;
		num( stp->minlen, false, false );
		num( stp->maxlen, false, true );
		num( stp->minglen, true, false );
		num( stp->maxglen, true, true );
		num( stp->minilen, true, false );
		num( stp->maxilen, true, true );
		const Limits	&l = lim[ stp->index ];
		char	t[ 32 ];
		snprintf( t, sizeof( t ), "%s%d", !l.start.l2r ? "$-" : "", l.start.offset );
		snprintf( buf, sizeof( buf ), " %5s", t );
		line += buf;
		snprintf( t, sizeof( t ), "%s%d", !l.stop.l2r ? "$-" : "", l.stop.offset );
		snprintf( buf, sizeof( buf ), " %5s", t );
		line += buf;
		line += "  " + prefix + strel_name( stp->type );
		if( stp->scope == 0 )
			line += "+--+";
		line += "\n";
		fputs( line.c_str(), fp );
	}
	static std::string	mk_prefix( const Strel *stp, const std::string &prefix )	// :791-815
	{
		std::string	p1 = prefix;
		char	&last = p1[ p1.size() - 1 ];
		const bool	first = stp->inner != nullptr && stp->scope == 0;
		if( stp->next != nullptr )
			last = '|';
		else if( last != '|' )
			last = ' ';
		if( first )
			p1 += "  |";
		p1 += "  +";
		return p1;
	}
	void	hierarchy( const std::string &prefix, int fd )	// print_hierarchy :695-727
	{
		for( int dd = fd; ; ){
			const Strel	*stp = &d.descr[ dd ];
			one_element( prefix, stp );
			const std::string	prefix1 = mk_prefix( stp, prefix );
			for( size_t s = 1; s < stp->scopes.size(); s++ ){
				const Strel	*a = stp->scopes[ s - 1 ], *b = stp->scopes[ s ];
				if( a->index + 1 < b->index )
					hierarchy( prefix1, a->index + 1 );
				one_element( mk_prefix( b, prefix ), b );
			}
			if( stp->next == nullptr )
				return;
			dd = stp->next->index;
		}
	}
	void	searches()					// print_searches :817-841, compile.c:3289-3314
	{
		const int	n = int( d.searches.size() );
		fprintf( fp, "total search depth: %3d\n", n );
		fprintf( fp, "srch# desc# type  forward  backup\n" );
		for( int s = 0; s < n; s++ ){
			const Strel	*stp = d.searches[ s ];
			fprintf( fp, "%4d %5d %5s", s, stp->index, strel_name( stp->type ) );
			if( s + 1 < n )
				fprintf( fp, " %8d", d.searches[ s + 1 ]->index );
			else
				fprintf( fp, "   (None)" );
			const Strel	*bk = nullptr;
			if( s > 0 ){
				if( stp->prev != nullptr )
					bk = stp->prev;
				else if( stp->outer != nullptr )
					bk = stp->outer->attr[ SA_PROPER ] ? stp->outer : stp->outer->scopes[ 0 ];
			}
			if( bk != nullptr )
				fprintf( fp, " %7d", bk->index );
			else
				fprintf( fp, "  (None)" );
			fprintf( fp, "\n" );
		}
	}
};

}	// namespace

void dump_descriptor( Descriptor &d, FILE *fp, int d_parms, int d_descr, int d_sites, int d_hierarchy )	// RM_dump, dump.c:34-83
{
	Dumper	dm( d, fp );
	if( !d.args.incdirs.empty() ){
		fprintf( fp, "INCLUDES: %3d dirs.\n", int( d.args.incdirs.size() ) );
		for( const std::string &s : d.args.incdirs )
			fprintf( fp, "\t%s\n", s.c_str() );
	}
	if( d_parms ){
		fprintf( fp, "PARMS: %3d global symbols.\n", int( d.globals.size() ) );
		// the reference walks its binary tree in order = strcmp() order = std::map order
		for( const auto &kv : d.globals )
			dm.ident( kv.second, d_parms );
	}
	const bool	linked = !d.descr.empty() && !d.searches.empty();
	if( ( d_descr || d_hierarchy ) && linked )
		dm.find_limits( 0 );
	else
		for( Limits &l : dm.lim ){
			l.start.offset = UNDEF;
			l.stop.offset = UNDEF;
		}
	if( d_descr ){
		fprintf( fp, "DESCR: %3d structure elements.\n", int( d.descr.size() ) );
		for( size_t i = 0; i < d.descr.size(); i++ )
			dm.strel( &d.descr[ i ], dm.lim[ i ] );
		Limits	none;
		none.start.offset = none.stop.offset = UNDEF;
		if( d.lctx != nullptr ){
			fprintf( fp, "DESCR: left context (%s).\n", d.lctx_explicit ? "Explicit" : "Implicit" );
			dm.strel( d.lctx, none );
		}
		if( d.rctx != nullptr ){
			fprintf( fp, "DESCR: right context (%s).\n", d.rctx_explicit ? "Explicit" : "Implicit" );
			dm.strel( d.rctx, none );
		}
	}
	if( d_sites )
		dm.sites();
	if( d_hierarchy ){
		fprintf( fp, "desc# minl  maxl  mngl  mxgl  mnil  mxil start  stop  descr\n" );
		if( linked ){
			dm.hierarchy( "+", 0 );
			dm.searches();
		}
	}
}

}	// namespace rma

// rm_regex.cpp -- see rm_regex.h.  Behaviour follows
// /root/reference/src/regexp.c (compile :124-387, step :389-424, advance :426-664)
// and /root/reference/src/mm_regexp.c (mm_seqlen :51-230, mm_step :353-367,
// mm_advance :369-469); the implementation is a parsed op vector with a
// recursive matcher, not the SysV byte code interpreter.
#include "rm_regex.h"
#include <cctype>
#include <cstring>

namespace rma {

namespace {

inline void place( ReOp &op, int c ) { op.set[ ( c & 0x7f ) >> 3 ] |= uint8_t( 1u << ( c & 7 ) ); }

bool is_group_or_word( const ReOp &op )
{
	return op.kind == RE_BRA || op.kind == RE_KET || op.kind == RE_BRC || op.kind == RE_LET;
}

}	// namespace

bool re_compile( const char *pat, ReProg &out )
{
	out.ops.clear();
	out.err = 0;
	const unsigned char *sp = reinterpret_cast<const unsigned char *>( pat );
	int	c = *sp++;
	if( c == '\0' || c == '\n' ){	// regexp.c:142-150: "no remembered search string"
		out.err = 41;
		return false;
	}
	if( c != '^' )
		--sp;
	int	last = -1;		// index of the op a following * or \{ applies to
	int	nbra = 0, closed = 0;
	std::vector<int>	open;
	for( ; ; ){
		c = *sp++;
		if( c == '\0' ){
			if( !open.empty() ){ out.err = 42; return false; }
			return true;
		}
		if( c != '*' && ( c != '\\' || *sp != '{' ) )
			last = int( out.ops.size() );
		switch( c ){
		case '.' :
			out.ops.push_back( ReOp{ RE_DOT } );
			continue;
		case '\n' :
			if( !open.empty() ){ out.err = 42; return false; }
			return true;
		case '*' :
			if( last < 0 || last >= int( out.ops.size() ) || is_group_or_word( out.ops[ last ] ) )
				break;		// literal '*'
			// *lastep |= STAR: on a RANGE op the low bits are already 3
			if( out.ops[ last ].rep == REP_ONE )
				out.ops[ last ].rep = REP_STAR;
			continue;
		case '$' :
			if( *sp != '\0' && *sp != '\n' )
				break;		// literal '$'
			out.ops.push_back( ReOp{ RE_DOL } );
			continue;
		case '[' : {
			ReOp	op{ RE_CCL };
			int	lc = 0;
			bool	neg = false;
			c = *sp++;
			if( c == '^' ){
				neg = true;
				c = *sp++;
			}
			do{
				if( c == '\0' || c == '\n' ){ out.err = 49; return false; }
				if( c & 0200 ){ out.err = 49; return false; }	// 8-bit classes never occur here
				if( c == '-' && lc != 0 ){
					c = *sp++;
					if( c == ']' ){
						place( op, '-' );
						break;
					}
					if( c == '\0' || ( c & 0200 ) ){ out.err = 49; return false; }
					while( lc < c ){
						place( op, lc );
						lc++;
					}
				}
				lc = c;
				place( op, c );
			}while( ( c = *sp++ ) != ']' );
			if( neg ){
				op.kind = RE_NCCL;
				op.set[ 0 ] |= 1;	// NUL is "in" the set so the negation fails on it
			}
			out.ops.push_back( op );
			continue;
		}
		case '\\' :
			c = *sp++;
			switch( c ){
			case '<' :
				out.ops.push_back( ReOp{ RE_BRC } );
				continue;
			case '>' :
				out.ops.push_back( ReOp{ RE_LET } );
				continue;
			case '(' : {
				if( nbra >= 9 ){ out.err = 43; return false; }
				ReOp	op{ RE_BRA };
				op.c = uint8_t( nbra );
				open.push_back( nbra++ );
				out.ops.push_back( op );
				continue;
			}
			case ')' : {
				if( open.empty() ){ out.err = 42; return false; }
				ReOp	op{ RE_KET };
				op.c = uint8_t( open.back() );
				open.pop_back();
				closed++;
				out.ops.push_back( op );
				continue;
			}
			case '{' : {
				if( last < 0 || last >= int( out.ops.size() ) )
					break;		// literal '{'
				int	nums[ 2 ] = { 0, 0 }, n = 0;
				bool	open_hi = false;
				for( ; ; ){
					c = *sp++;
					int	i = 0;
					do{
						if( c >= '0' && c <= '9' )
							i = 10 * i + c - '0';
						else{ out.err = 16; return false; }
					}while( ( c = *sp++ ) != '\\' && c != ',' );
					if( i > 255 ){ out.err = 11; return false; }
					nums[ n++ ] = i;
					if( c == ',' ){
						if( n == 2 ){ out.err = 44; return false; }
						if( *sp == '\\' ){
							sp++;
							open_hi = true;
							break;
						}
						continue;	// second number
					}
					break;
				}
				if( *sp++ != '}' ){ out.err = 45; return false; }
				ReOp	&lop = out.ops[ last ];
				lop.rep = REP_RANGE;
				lop.lo = uint8_t( nums[ 0 ] );
				if( open_hi )
					lop.hi = 255;
				else if( n == 1 )
					lop.hi = lop.lo;
				else{
					lop.hi = uint8_t( nums[ 1 ] );
					if( lop.hi < lop.lo ){ out.err = 46; return false; }
				}
				continue;
			}
			case '\n' :
				out.err = 36;
				return false;
			default :
				if( c >= '1' && c <= '9' ){
					if( c - '1' >= closed ){ out.err = 25; return false; }
					ReOp	op{ RE_BACK };
					op.c = uint8_t( c - '1' );
					out.ops.push_back( op );
					continue;
				}
				break;		// \x is x
			}
			break;
		default :
			break;
		}
		// literal character
		last = int( out.ops.size() );
		ReOp	op{ RE_CHR };
		op.c = uint8_t( c );
		out.ops.push_back( op );
	}
}

namespace {

struct Matcher {
	const ReProg	&re;
	const char	*start;		// loc1 at entry of step(): string start, for \<
	const char	*bra[ 9 ] = {}, *ket[ 9 ] = {};
	const char	*loc2 = nullptr;

	bool one( const ReOp &op, int ch ) const
	{
		switch( op.kind ){
		case RE_CHR :	return ch == op.c;
		case RE_DOT :	return ch != 0;
		case RE_CCL :	return ( ch & 0200 ) == 0 && op.has( ch );
		case RE_NCCL :	return !( ( ch & 0200 ) == 0 && op.has( ch ) );
		default :	return false;
		}
	}

	bool advance( const char *lp, size_t ip )
	{
		for( ; ; ip++ ){
			if( ip == re.ops.size() ){
				loc2 = lp;
				return true;
			}
			const ReOp	&op = re.ops[ ip ];
			if( op.kind == RE_DOL ){
				if( *lp == 0 )
					continue;
				return false;
			}
			if( op.kind == RE_BRA ){ bra[ op.c ] = lp; continue; }
			if( op.kind == RE_KET ){ ket[ op.c ] = lp; continue; }
			if( op.kind == RE_BRC ){
				if( lp == start )
					continue;
				unsigned char	ch = *lp;
				if( isalpha( ch ) || ch == '_' || isdigit( ch ) ){
					unsigned char	pc = lp[ -1 ];
					if( !( isalpha( pc ) || pc == '_' ) && !isdigit( pc ) )
						continue;
				}
				return false;
			}
			if( op.kind == RE_LET ){
				unsigned char	ch = *lp;
				if( !( isalpha( ch ) || ch == '_' ) && !isdigit( ch ) )
					continue;
				return false;
			}
			if( op.kind == RE_BACK ){
				const char	*bb = bra[ op.c ];
				size_t	ct = ket[ op.c ] - bb;
				if( op.rep == REP_ONE ){
					if( !strncmp( bb, lp, ct ) ){
						lp += ct;
						continue;
					}
					return false;
				}
				const char	*cur = lp;
				while( ct > 0 && !strncmp( bb, lp, ct ) )
					lp += ct;
				for( ; lp >= cur; lp -= ct ){
					if( advance( lp, ip + 1 ) )
						return true;
					if( ct == 0 )
						break;
				}
				return false;
			}
			// single character ops
			if( op.rep == REP_ONE ){
				if( one( op, (unsigned char)*lp ) ){
					lp++;
					continue;
				}
				return false;
			}
			int	lo = 0, extra;
			if( op.rep == REP_RANGE ){
				lo = op.lo;
				extra = op.hi == 255 ? 20000 : op.hi - op.lo;
			}else
				extra = 0x7fffffff;
			for( ; lo > 0; lo-- ){
				if( !one( op, (unsigned char)*lp ) )
					return false;
				lp++;
			}
			const char	*cur = lp;
			for( ; extra > 0 && one( op, (unsigned char)*lp ); extra-- )
				lp++;
			for( ; lp >= cur; lp-- ){	// longest first, regexp.c:608-641
				if( advance( lp, ip + 1 ) )
					return true;
			}
			return false;
		}
	}
};

}	// namespace

bool re_step( const ReProg &re, const char *s, bool anchored, ReMatch *m )
{
	Matcher	mt{ re, s };
	const char	*p = s;
	if( anchored ){
		bool	ok = mt.advance( p, 0 );
		if( m ){ m->loc1 = s; m->loc2 = mt.loc2; }
		return ok;
	}
	do{
		if( mt.advance( p, 0 ) ){
			if( m ){ m->loc1 = p; m->loc2 = mt.loc2; }
			return true;
		}
	}while( *p++ );
	if( m ){ m->loc1 = s; m->loc2 = nullptr; }
	return false;
}

namespace {

// mm_advance, mm_regexp.c:369-469
bool mm_advance( const ReProg &re, const char *lp, int l_mm, int *n_mm )
{
	*n_mm = 0;
	for( const ReOp &op : re.ops ){
		int	reps = 1;
		if( op.rep == REP_RANGE )
			reps = op.lo;
		else if( op.rep == REP_STAR )
			continue;	// the reference ignores the opcode; never reached for seq= (mmok)
		switch( op.kind ){
		case RE_CHR :
			for( ; reps > 0; reps-- ){
				int	ch = (unsigned char)*lp++;
				if( ch != op.c ){
					if( ch == 0 )
						return false;
					if( ++*n_mm > l_mm )
						return false;
				}
			}
			break;
		case RE_DOT :
			for( ; reps > 0; reps-- ){
				if( *lp++ == 0 )
					return false;
			}
			break;
		case RE_DOL :
			if( *lp != 0 )
				return false;
			break;
		case RE_CCL :
		case RE_NCCL : {
			bool	neg = op.kind == RE_NCCL;
			for( ; reps > 0; reps-- ){
				int	ch = (unsigned char)*lp++;
				if( ch == 0 )
					return false;
				bool	in = ( ch & 0200 ) == 0 && op.has( ch );
				if( in == neg ){
					if( ++*n_mm > l_mm )
						return false;
				}
			}
			break;
		}
		default :
			break;
		}
	}
	return true;
}

}	// namespace

bool re_mm_step( const ReProg &re, const char *s, bool anchored, int l_mm, int *n_mm )
{
	if( anchored )
		return mm_advance( re, s, l_mm, n_mm );
	const char	*p = s;
	do{
		if( mm_advance( re, p, l_mm, n_mm ) )
			return true;
	}while( *p++ );
	return false;
}

// mm_seqlen() over an expression with \( \) or \1 in it.  The reference walks its byte code one opcode at a time and
// has no case for CKET's and CBACK's group number, nor for CBACK itself (mm_regexp.c:79-84, :196-198): the number is read
// as the next opcode -- 0, 1 and 3 fall to the default case, which says "mm_seqlen: %2d?" on stderr and goes on; 2 is CBRA
// and swallows the byte behind it; 4 is CCHR and counts a base ... -- so the implied lengths of such an expression, and what
// the compiler says about it, are what that walk makes of the bytes.  The same walk here, over the same bytes (opcode values
// regexp.c:74-88; counts and characters are signed chars there).
static void seqlen_of_bytes( const ReProg &re, bool caret, int *minl, int *maxl, int *mmok, std::string *diag )
{
	enum { CBRA = 2, CCHR = 4, CDOT = 8, CCL = 12, CDOL = 20, CCEOF = 22, CKET = 24, CBRC = 28, CLET = 30, CBACK = 36, NCCL = 40, STAR = 1, RNGE = 3 };
	std::vector<signed char>	b;
	for( const ReOp &op : re.ops ){
		const int	rep = op.rep == REP_STAR ? STAR : op.rep == REP_RANGE ? RNGE : 0;
		switch( op.kind ){
		case RE_CHR :	b.push_back( ( signed char )( CCHR | rep ) ); b.push_back( ( signed char )op.c ); break;
		case RE_DOT :	b.push_back( ( signed char )( CDOT | rep ) ); break;
		case RE_CCL :
		case RE_NCCL :
			b.push_back( ( signed char )( ( op.kind == RE_CCL ? CCL : NCCL ) | rep ) );
			for( int k = 0; k < 16; k++ )
				b.push_back( ( signed char )op.set[ k ] );
			break;
		case RE_DOL :	b.push_back( CDOL ); break;
		case RE_BRA :	b.push_back( CBRA ); b.push_back( ( signed char )op.c ); break;
		case RE_KET :	b.push_back( CKET ); b.push_back( ( signed char )op.c ); break;
		case RE_BACK :	b.push_back( ( signed char )( CBACK | ( op.rep == REP_STAR ? STAR : 0 ) ) ); b.push_back( ( signed char )op.c ); break;
		case RE_BRC :	b.push_back( CBRC ); break;
		case RE_LET :	b.push_back( CLET ); break;
		}
		if( op.rep == REP_RANGE && op.kind != RE_BRA && op.kind != RE_KET && op.kind != RE_BACK ){
			b.push_back( ( signed char )op.lo );
			b.push_back( ( signed char )op.hi );
		}
	}
	b.push_back( CCEOF );
	b.resize( b.size() + 40, CCEOF );		// (a walk that has lost its place stops here at the latest)
	*minl = 0;
	*maxl = RMA_UNDEF;
	*mmok = 1;
	bool	dol = false, star = false;
	int	trng = RMA_UNDEF;
	for( size_t i = 0; i < b.size() && b[ i ] != CCEOF; i++ ){
		int	rng = RMA_UNDEF;
		const signed char	*ep = &b[ i ];
		switch( *ep ){
		case CBRA :	i++; break;
		case CKET :	break;
		case CBRC :
		case CLET :	*mmok = 0; break;
		case CCHR :	( *minl )++; i++; break;
		case CCHR | STAR :	star = true; *mmok = 0; i++; break;
		case CCHR | RNGE :
			*minl += ep[ 1 ];
			rng = ep[ 2 ] - ep[ 1 ];
			if( ep[ 1 ] != ep[ 2 ] )
				*mmok = 0;
			i += 3;
			break;
		case CDOT :	( *minl )++; break;
		case CDOT | STAR :	star = true; *mmok = 0; break;
		case CDOT | RNGE :
			*minl += ep[ 1 ];
			rng = ep[ 2 ] - ep[ 1 ];
			if( ep[ 1 ] != ep[ 2 ] )
				*mmok = 0;
			i += 2;
			break;
		case CCL : case NCCL :	( *minl )++; i += 16; break;
		case CCL | STAR : case NCCL | STAR :	star = true; *mmok = 0; i += 16; break;
		case CCL | RNGE : case NCCL | RNGE :
			*minl += ep[ 16 + 1 ];
			rng = ep[ 16 + 2 ] - ep[ 16 + 1 ];
			if( ep[ 16 + 1 ] != ep[ 16 + 2 ] )
				*mmok = 0;
			i += 18;
			break;
		case CDOL :	dol = true; break;
		default :
			if( diag ){
				char	line[ 40 ];
				snprintf( line, sizeof( line ), "mm_seqlen: %2d?\n", int( *ep ) );
				*diag += line;
			}
			break;
		}
		if( rng != RMA_UNDEF )
			trng = trng == RMA_UNDEF ? rng : trng + rng;
	}
	if( caret && dol && !star )
		*maxl = trng == RMA_UNDEF ? *minl : *minl + trng;
}

void re_seqlen( const ReProg &re, bool caret, int *minl, int *maxl, int *mmok, std::string *diag )
{
	for( const ReOp &op : re.ops )
		if( op.kind == RE_BRA || op.kind == RE_KET || op.kind == RE_BACK ){
			seqlen_of_bytes( re, caret, minl, maxl, mmok, diag );
			return;
		}
	*minl = 0;
	*maxl = RMA_UNDEF;
	*mmok = 1;
	bool	dol = false, star = false;
	int	trng = RMA_UNDEF;
	for( const ReOp &op : re.ops ){
		int	rng = RMA_UNDEF;
		// the byte code stores counts in (signed) chars
		int	lo = int8_t( op.lo ), hi = int8_t( op.hi );
		switch( op.kind ){
		case RE_BRC :
		case RE_LET :
		case RE_BACK :
			*mmok = 0;
			break;
		case RE_CHR :
			if( op.rep == REP_STAR ){
				star = true;
				*mmok = 0;
			}else if( op.rep == REP_RANGE ){
				// mm_regexp.c:111-117 reads the character where the low
				// count is: ep[1] is the literal, ep[2] the low count
				int	ch = int8_t( op.c );
				*minl += ch;
				rng = lo - ch;
				if( ch != lo )
					*mmok = 0;
			}else
				( *minl )++;
			break;
		case RE_DOT :
		case RE_CCL :
		case RE_NCCL :
			if( op.rep == REP_STAR ){
				star = true;
				*mmok = 0;
			}else if( op.rep == REP_RANGE ){
				*minl += lo;
				rng = hi - lo;
				if( lo != hi )
					*mmok = 0;
			}else
				( *minl )++;
			break;
		case RE_DOL :
			dol = true;
			break;
		default :
			break;
		}
		if( rng != RMA_UNDEF )
			trng = trng == RMA_UNDEF ? rng : trng + rng;
	}
	if( caret && dol && !star )
		*maxl = trng == RMA_UNDEF ? *minl : *minl + trng;
}

bool re_to_atoms( const ReProg &re, bool caret, rma_regex_t *out, std::string &why )
{
	memset( out, 0, sizeof( *out ) );
	out->anchored = caret;
	out->fixed_len = 0;
	auto code = []( int ch ) -> int {
		switch( ch ){
		case 'a' : return RMA_BC_A;
		case 'c' : return RMA_BC_C;
		case 'g' : return RMA_BC_G;
		case 't' : return RMA_BC_T;
		default : return -1;
		}
	};
	for( size_t i = 0; i < re.ops.size(); i++ ){
		const ReOp	&op = re.ops[ i ];
		rma_re_atom_t	at{};
		switch( op.kind ){
		case RE_BRA :
		case RE_KET :
			continue;	// no effect without back references
		case RE_DOL :
			if( i + 1 != re.ops.size() ){ why = "'$' inside a pattern"; return false; }
			out->dollar = 1;
			continue;
		case RE_BACK :
			// the text of a group again: any run of letters, as far as the packed database can tell (the host
			// applies the expression itself when it replays the candidates: rma_regex_t::loose)
			out->loose = 1;
			at.mask = 0x1f;
			at.kind = 1;
			at.lo = 0;
			at.hi = 255;
			if( out->n_atoms >= RMA_MAX_RE_ATOMS ){ why = "seq= pattern too long"; return false; }
			out->atoms[ out->n_atoms++ ] = at;
			out->fixed_len = -1;
			continue;
		case RE_BRC :
		case RE_LET :
			// \< and \> hold at an end of a string of letters at most: no base is asked for
			out->loose = 1;
			continue;
		case RE_CHR : {
			int	bc = code( op.c );
			if( bc < 0 ){
				// iupac = 0: the letter itself.  A sequence letter that is not acgt is all the packed database
				// knows of it (code 4); anything that is no letter never occurs in a sequence
				out->loose = 1;
				at.mask = isalpha( op.c ) ? 0x10 : 0;
			}else
				at.mask = uint8_t( 1u << bc );
			at.kind = 0;
			break;
		}
		case RE_DOT :
			at.mask = 0x1f;
			at.kind = 1;
			break;
		case RE_CCL :
		case RE_NCCL : {
			unsigned	m = 0;
			for( int ch = 1; ch < 128; ch++ ){
				if( !op.has( ch ) )
					continue;
				int	bc = code( ch );
				if( bc >= 0 )
					m |= 1u << bc;
				else if( isalpha( ch ) && islower( ch ) ){
					// a member that is not one of acgt: some of the letters behind code 4
					out->loose = 1;
					m |= 0x10;
				}
			}
			if( op.kind == RE_NCCL ){
				// (every letter outside the set; with a member behind code 4 left out, the others behind it still match)
				at.mask = uint8_t( ( ( ~m ) | 0x10 ) & 0x1f );
				at.kind = 3;
			}else{
				at.mask = uint8_t( m );
				at.kind = 2;
			}
			break;
		}
		}
		if( op.rep == REP_ONE ){
			at.lo = at.hi = 1;
		}else if( op.rep == REP_STAR ){
			at.lo = 0;
			at.hi = 255;
		}else{
			at.lo = op.lo;
			at.hi = op.hi;
		}
		if( out->n_atoms >= RMA_MAX_RE_ATOMS ){ why = "seq= pattern too long"; return false; }
		out->atoms[ out->n_atoms++ ] = at;
		if( out->fixed_len >= 0 ){
			if( at.lo == at.hi && at.hi != 255 )
				out->fixed_len += at.lo;
			else
				out->fixed_len = -1;
		}
	}
	return true;
}

}	// namespace rma

// rmprune -- drop hits that are only "unzipped" versions of another hit of the same
// entry: when a descriptor allows a range of helix lengths, rnamotif reports every
// admissible length of the same helix; of a family that differs only by opening base
// pairs at the inside end of helices (and closing none elsewhere) the most zipped-up
// member is kept.  Same command line, grouping and decisions as the reference's tool
// (/root/reference/src/rmprune.c): hits are taken per entry name (up to 1000 at a time),
// split by strand and into groups that lie inside the span of the group's first hit,
// and compared pairwise from the last to the first.
//
//   usage: rmprune [ rnamotif-out-file ]
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

enum { K_UNKNOWN = -1, K_CTX, K_SS, K_H5, K_H3, K_P5, K_P3, K_T1, K_T2, K_T3, K_Q1, K_Q2, K_Q3, K_Q4 };
enum { R_NONE = -1, R_SAME, R_LEFT, R_DOWN, R_DIFF };	// relation of two hits (rmprune.c:86-90)

struct Elem { int kind = K_UNKNOWN; int group[ 4 ] = { -1, -1, -1, -1 }; };
struct Span { int start = 0, stop = 0; };
struct Hit {
	bool	keep = true;
	std::string	def, text;
	int	comp = 0, start = 0, stop = 0;
	std::vector<Span>	el;
};

std::vector<Elem>	descr;

int kind_of( const std::string &w )
{
	static const char	*names[] = { "ctx", "ss", "h5", "h3", "p5", "p3", "t1", "t2", "t3", "q1", "q2", "q3", "q4" };
	for( int k = 0; k < 13; k++ )
		if( !strncmp( names[ k ], w.c_str(), 2 ) )
			return k;
	return K_UNKNOWN;
}

std::vector<std::string> words( const std::string &line )
{
	std::vector<std::string>	w;
	size_t	i = 0;
	while( i < line.size() ){
		while( i < line.size() && strchr( " \t\n", line[ i ] ) )
			i++;
		size_t	j = i;
		while( j < line.size() && !strchr( " \t\n", line[ j ] ) )
			j++;
		if( j > i )
			w.push_back( line.substr( i, j - i ) );
		i = j;
	}
	return w;
}

bool read_line( FILE *fp, std::string &line )
{
	line.clear();
	int	c;
	while( ( c = getc( fp ) ) != EOF ){
		if( line.size() < 50000 )
			line.push_back( char( c ) );
		if( c == '\n' )
			break;
	}
	return !line.empty();
}

void read_descr( const std::vector<std::string> &w )	// rmprune.c:232-302
{
	const int	n = int( w.size() ) - 2;
	descr.assign( size_t( n > 0 ? n : 0 ), Elem() );
	for( int d = 0; d < n; d++ )
		descr[ d ].kind = kind_of( w[ d + 2 ] );
	for( int d = 0; d < n; d++ ){
		if( descr[ d ].kind == K_SS )
			continue;
		size_t	t = w[ d + 2 ].find( '(' );
		if( t == std::string::npos || descr[ d ].group[ 0 ] != -1 )
			continue;
		const std::string	tag = w[ d + 2 ].substr( t );
		int	k = 0;
		descr[ d ].group[ k++ ] = d;
		for( int d1 = d + 1; d1 < n; d1++ ){
			size_t	t1 = w[ d1 + 2 ].find( '(' );
			if( t1 != std::string::npos && w[ d1 + 2 ].substr( t1 ) == tag && k < 4 )
				descr[ d ].group[ k++ ] = d1;
		}
		for( int m = 1; m < 4 && descr[ d ].group[ m ] != -1; m++ )
			memcpy( descr[ descr[ d ].group[ m ] ].group, descr[ d ].group, sizeof( descr[ d ].group ) );
	}
	std::vector<int>	stk;
	for( int d = 0; d < n; d++ ){
		if( descr[ d ].kind == K_SS || descr[ d ].group[ 0 ] != -1 )
			continue;
		if( descr[ d ].kind == K_H5 || descr[ d ].kind == K_P5 )
			stk.push_back( d );
		else if( !stk.empty() ){
			const int	d5 = stk.back();
			stk.pop_back();
			descr[ d5 ].group[ 0 ] = descr[ d ].group[ 0 ] = d5;
			descr[ d5 ].group[ 1 ] = descr[ d ].group[ 1 ] = d;
		}
	}
}

// positions of the elements in the entry (getdetails, rmprune.c:611-640); an empty
// element printed as "." counts as one position, as there
void locate( Hit &h )
{
	const int	n = int( descr.size() );
	h.el.assign( size_t( n ), Span() );
	long	p = long( h.text.size() ) - 1;
	int	blanks = 0;
	for( ; p >= 0; p-- ){
		if( h.text[ p ] == ' ' )
			blanks++;
		if( blanks == n ){
			p++;
			break;
		}
	}
	if( p < 0 )
		p = 0;
	size_t	q = size_t( p );
	int	done = 0;
	for( int i = 0; q < h.text.size() && i < n; i++ ){
		size_t	e = h.text.find_first_of( " \n", q );
		if( e == std::string::npos )
			e = h.text.size();
		const int	len = int( e - q );
		if( !h.comp ){
			h.el[ i ].start = h.start + done;
			h.el[ i ].stop = h.el[ i ].start + len - 1;
		}else{
			h.el[ i ].start = h.start - done;
			h.el[ i ].stop = h.el[ i ].start - len + 1;
		}
		q = h.text.find_first_not_of( " \n", e );
		if( q == std::string::npos )
			q = h.text.size();
		done += len;
	}
}

int helix_relation( int comp, const Span &a5, const Span &a3, const Span &b5, const Span &b3 )	// wchlxrel :687-733
{
	int	lod, rod, lid, rid;	// left/right, outer/inner differences
	if( !comp ){
		lod = b5.start - a5.start;
		rod = a3.stop - b3.stop;
		lid = a5.stop - b5.stop;
		rid = b3.start - a3.start;
	}else{
		lod = a5.start - b5.start;
		rod = b3.stop - a3.stop;
		lid = b5.stop - a5.stop;
		rid = a3.start - b3.start;
	}
	if( lod != rod || lid != rid )
		return R_DIFF;
	if( lod > 0 )
		return lid < 0 ? R_DIFF : R_DOWN;
	if( lod == 0 )
		return lid < 0 ? R_LEFT : lid == 0 ? R_SAME : R_DOWN;
	return lid < 0 ? R_DIFF : R_LEFT;
}

int relation( const Hit &a, const Hit &b )	// chkrel :642-685
{
	int	rel = R_NONE, r1 = R_NONE;
	for( size_t d = 0; d < descr.size(); d++ ){
		const Elem	&e = descr[ d ];
		switch( e.kind ){
		case K_H5 :
			if( e.group[ 1 ] >= 0 )
				r1 = helix_relation( a.comp, a.el[ d ], a.el[ e.group[ 1 ] ], b.el[ d ], b.el[ e.group[ 1 ] ] );
			break;
		case K_P5 : case K_T1 : case K_Q1 :
			r1 = R_SAME;
			for( int i = 0; i < 4 && e.group[ i ] != -1; i++ ){
				const Span	&x = a.el[ e.group[ i ] ], &y = b.el[ e.group[ i ] ];
				if( x.start != y.start || x.stop != y.stop ){
					r1 = R_DIFF;
					break;
				}
			}
			break;
		case K_UNKNOWN : case K_CTX :
			break;
		default :
			r1 = R_SAME;
			break;
		}
		if( r1 == R_DIFF )
			return R_DIFF;
		if( rel == R_NONE || rel == R_SAME )
			rel = r1;
		else if( rel == R_DOWN && r1 == R_LEFT )
			return R_DIFF;
		else if( rel == R_LEFT && r1 == R_DOWN )
			return R_DIFF;
	}
	return rel;
}

void rezip( Hit *g, int n )	// rezip_group :404-445
{
	if( n < 2 )
		return;
	for( int b = 0; b < n; b++ )
		locate( g[ b ] );
	for( int b = n - 1; b > 0; b-- ){
		if( !g[ b ].keep )
			continue;
		for( int b1 = b - 1; b1 >= 0; b1-- ){
			if( !g[ b1 ].keep )
				continue;
			const int	r = relation( g[ b ], g[ b1 ] );
			if( r == R_DOWN )
				g[ b1 ].keep = false;
			else if( r == R_LEFT ){
				g[ b ].keep = false;
				break;
			}
		}
	}
}

void prune( std::vector<Hit> &blk )	// prune_block :332-402
{
	const int	n = int( blk.size() );
	if( n == 0 )
		return;
	if( n > 1 ){
		int	first_comp = 0;
		while( first_comp < n && !blk[ first_comp ].comp )
			first_comp++;
		int	start = blk[ 0 ].start, stop = blk[ 0 ].stop, lb = 0, b;
		for( b = 0; b < first_comp; b++ ){
			if( blk[ b ].start < start || blk[ b ].stop > stop ){
				rezip( &blk[ lb ], b - lb );
				start = blk[ b ].start;
				stop = blk[ b ].stop;
				lb = b;
			}
		}
		rezip( blk.data() + lb, b - lb );
		if( first_comp < n ){
			start = blk[ first_comp ].start;
			stop = blk[ first_comp ].stop;
		}
		for( lb = b = first_comp; b < n; b++ ){
			if( blk[ b ].start > start || blk[ b ].stop < stop ){
				rezip( &blk[ lb ], b - lb );
				start = blk[ b ].start;
				stop = blk[ b ].stop;
				lb = b;
			}
		}
		rezip( blk.data() + lb, b - lb );
	}
	for( const Hit &h : blk ){
		if( !h.keep )
			continue;
		fputs( h.def.c_str(), stdout );
		fputs( h.text.c_str(), stdout );
	}
	blk.clear();
}

}	// namespace

int main( int argc, char **argv )
{
	const char	*fname = nullptr;
	for( int ac = 1; ac < argc; ac++ ){
		if( fname != nullptr ){
			fprintf( stderr, "usage: %s [ rnamotif-out-file ]\n", argv[ 0 ] );
			return 1;
		}
		fname = argv[ ac ];
	}
	FILE	*fp = fname ? fopen( fname, "r" ) : stdin;
	if( fp == nullptr ){
		fprintf( stderr, "rmprune: can't read rnamotif-out-file %s.\n", fname );
		return 1;
	}
	std::string	line, pending;
	while( read_line( fp, line ) ){
		if( line[ 0 ] == '>' ){
			pending = line;
			break;
		}
		std::vector<std::string>	w = words( line );
		if( w.empty() )
			continue;
		if( w.size() >= 2 && w[ 0 ] == "#RM" && w[ 1 ] == "descr" )
			read_descr( w );
		fputs( line.c_str(), stdout );
	}
	auto next = [&]() -> bool {
		if( !pending.empty() ){
			line.swap( pending );
			pending.clear();
			return true;
		}
		return read_line( fp, line );
	};
	const int	n = int( descr.size() );
	std::vector<Hit>	blk;
	std::string	last_name;
	while( next() ){
		if( line[ 0 ] == '#' )
			continue;
		if( line[ 0 ] != '>' ){
			fprintf( stderr, "rmprune: unexpected input: '%s'\n", line.c_str() );
			return 1;
		}
		// the entry name: up to the first '.' or blank (getname, :313-330)
		size_t	q = 1;
		while( q < line.size() && isspace( ( unsigned char )line[ q ] ) )
			q++;
		if( q >= line.size() ){
			next();
			continue;
		}
		size_t	e = q;
		while( e < line.size() && line[ e ] != '.' && !isspace( ( unsigned char )line[ e ] ) )
			e++;
		const std::string	name = line.substr( q, e - q );
		if( name != last_name || blk.size() >= 1000 )
			prune( blk );
		Hit	h;
		h.def = line;
		if( !next() )
			line.clear();
		h.text = line;
		// comp, position and length sit in front of the last n blank-separated fields (:779-816)
		long	p = long( line.size() ) - 1;
		int	blanks = 0;
		for( ; p >= 0; p-- ){
			if( line[ p ] == ' ' )
				blanks++;
			if( blanks == n )
				break;
		}
		auto digits_back = [&](){
			for( --p; p >= 0 && isdigit( ( unsigned char )line[ p ] ); p-- )
				;
		};
		auto blanks_back = [&](){
			for( ; p >= 0 && line[ p ] == ' '; p-- )
				;
		};
		digits_back();
		blanks_back();
		digits_back();
		blanks_back();
		int	len = 0;
		if( p >= 0 )
			sscanf( line.c_str() + p, "%d %d %d", &h.comp, &h.start, &len );
		h.stop = !h.comp ? h.start + len - 1 : h.start - len + 1;
		blk.push_back( h );
		last_name = name;
	}
	prune( blk );
	if( fp != stdin )
		fclose( fp );
	return 0;
}

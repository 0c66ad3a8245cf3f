// rm2ct -- rnamotif output -> connect (.ct) records, one per hit.  Same command
// line and output as the reference's tool (/root/reference/src/rm2ct.c): the
// "#RM descr" line names the elements; helices pair by tag "(...)" where one is
// printed, else by nesting; p5/p3 count the partner in the same direction; the
// last column is the position in the database entry (running backwards on the
// complement strand).
//
//   usage: rm2ct [ -t rnamotif | -t rnaviz ] [ rnamotif-out-file ]
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

const char	*USAGE =
"usage: %s [ options ] [ output-type ] [ rnamotif-out-file ]\n\
\n\
options:\n\
\t-help\t\t\tPrint this message\n\
\n\
output-type: (Optional) Use one\n\
	-t rnamotif\t\tNormal ct-format (default)\n\
	-t rnaviz\t\tStrict ct-format for rnaviz input\n\
";

enum { K_UNKNOWN = -1, K_CTX, K_SS, K_H5, K_H3, K_P5, K_P3, K_T1, K_T2, K_T3, K_Q1, K_Q2, K_Q3, K_Q4 };

struct Elem {
	int	kind = K_UNKNOWN;
	int	group[ 4 ] = { -1, -1, -1, -1 };	// strands of the same helix, 5' first
	int	first = 0;				// 0-based index of its first base in the hit
	size_t	lo = 0, hi = 0;				// [lo, hi) of its field in the line
};

int kind_of( const std::string &w )	// rm2ct.c:299-309: the first two letters decide
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

// rm2ct.c:232-297
std::vector<Elem> read_descr( const std::vector<std::string> &w )
{
	const int	n = int( w.size() ) - 2;
	std::vector<Elem>	el( size_t( n > 0 ? n : 0 ) );
	for( int d = 0; d < n; d++ )
		el[ d ].kind = kind_of( w[ d + 2 ] );
	// tagged strands: same "(tag)" text
	for( int d = 0; d < n; d++ ){
		if( el[ d ].kind == K_SS )
			continue;
		size_t	t = w[ d + 2 ].find( '(' );
		if( t == std::string::npos || el[ d ].group[ 0 ] != -1 )
			continue;
		const std::string	tag = w[ d + 2 ].substr( t );
		int	k = 0;
		el[ d ].group[ k++ ] = d;
		for( int d1 = d + 1; d1 < n; d1++ ){
			size_t	t1 = w[ d1 + 2 ].find( '(' );
			if( t1 != std::string::npos && w[ d1 + 2 ].substr( t1 ) == tag && k < 4 )
				el[ d ].group[ k++ ] = d1;
		}
		for( int m = 1; m < 4 && el[ d ].group[ m ] != -1; m++ )
			memcpy( el[ el[ d ].group[ m ] ].group, el[ d ].group, sizeof( el[ d ].group ) );
	}
	// untagged duplexes pair by nesting
	std::vector<int>	stk;
	for( int d = 0; d < n; d++ ){
		if( el[ d ].kind == K_SS || el[ d ].group[ 0 ] != -1 )
			continue;
		if( el[ d ].kind == K_H5 || el[ d ].kind == K_P5 )
			stk.push_back( d );
		else if( ( el[ d ].kind == K_H3 || el[ d ].kind == K_P3 ) && !stk.empty() ){
			const int	d5 = stk.back();
			stk.pop_back();
			el[ d5 ].group[ 0 ] = el[ d ].group[ 0 ] = d5;
			el[ d5 ].group[ 1 ] = el[ d ].group[ 1 ] = d;
		}
	}
	return el;
}

int partner( const std::vector<Elem> &el, const Elem &e, int k )	// getpn(), rm2ct.c:432-471
{
	const int	len = int( e.hi - e.lo );
	switch( e.kind ){
	case K_SS : return 0;
	case K_H5 : return el[ e.group[ 1 ] ].first + len - k;
	case K_H3 : return el[ e.group[ 0 ] ].first + len - k;
	case K_P5 : return el[ e.group[ 1 ] ].first + k;	// (0-based there, as the reference has it)
	case K_P3 : return el[ e.group[ 0 ] ].first + k;
	case K_T1 : case K_T2 : case K_T3 : case K_Q1 : case K_Q2 : case K_Q3 : case K_Q4 : return 0;
	}
	return -1;
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

}	// namespace

int main( int argc, char **argv )
{
	const char	*fname = nullptr;
	bool	rnaviz = false;
	for( int ac = 1; ac < argc; ac++ ){
		if( !strcmp( argv[ ac ], "-help" ) ){
			fprintf( stderr, USAGE, argv[ 0 ] );
			return 0;
		}else if( !strcmp( argv[ ac ], "-t" ) ){
			if( ++ac >= argc || ( strcmp( argv[ ac ], "rnamotif" ) && strcmp( argv[ ac ], "rnaviz" ) ) ){
				fprintf( stderr, USAGE, argv[ 0 ] );
				return 1;
			}
			rnaviz = !strcmp( argv[ ac ], "rnaviz" );
		}else if( argv[ ac ][ 0 ] == '-' || fname != nullptr ){
			fprintf( stderr, USAGE, argv[ 0 ] );
			return 1;
		}else
			fname = argv[ ac ];
	}
	FILE	*fp = fname ? fopen( fname, "r" ) : stdin;
	if( fp == nullptr ){
		fprintf( stderr, "rm2ct: can't read rnamotif-out-file %s.\n", fname );
		return 1;
	}
	std::string	line, pending;
	std::vector<Elem>	el;
	while( read_line( fp, line ) ){
		if( line[ 0 ] == '>' ){
			pending = line;
			break;
		}
		std::vector<std::string>	w = words( line );
		if( w.size() >= 2 && w[ 0 ] == "#RM" && w[ 1 ] == "descr" )
			el = read_descr( w );
	}
	for( const Elem &e : el ){
		if( e.kind == K_T1 ){
			fprintf( stderr, "rm2ct: can't make CT file for triple helices.\n" );
			return 1;
		}else if( e.kind == K_Q1 ){
			fprintf( stderr, "rm2ct: can't make CT file for quad helices.\n" );
			return 1;
		}
	}
	const int	n = int( el.size() );
	auto next = [&]() -> bool {
		if( !pending.empty() ){
			line.swap( pending );
			pending.clear();
			return true;
		}
		return read_line( fp, line );
	};
	while( next() ){
		if( line[ 0 ] == '#' )
			continue;
		if( line[ 0 ] == '>' && !next() )
			break;
		// the last n blank-separated fields are the elements (getsinfo, rm2ct.c:311-371)
		long	p = long( line.size() ) - 2;
		int	blanks = 0;
		size_t	seq0 = 0;
		for( ; p >= 0; p-- ){
			if( isspace( ( unsigned char )line[ p ] ) )
				blanks++;
			if( blanks >= n ){
				seq0 = size_t( p + 1 );
				break;
			}
		}
		auto back_int = [&]() -> int {
			while( p >= 0 && isspace( ( unsigned char )line[ p ] ) )
				p--;
			while( p >= 0 && isdigit( ( unsigned char )line[ p ] ) )
				p--;
			return p >= 0 ? atoi( line.c_str() + p ) : atoi( line.c_str() );
		};
		const int	len = back_int(), off = back_int(), comp = back_int();
		size_t	q = 0;
		while( q < line.size() && !isspace( ( unsigned char )line[ q ] ) )
			q++;
		const std::string	name = line.substr( 0, q );
		const float	energy = float( atof( line.c_str() + q ) );
		int	seen = 0;
		q = seq0;
		for( int d = 0; d < n; d++ ){
			while( q < line.size() && isspace( ( unsigned char )line[ q ] ) )
				q++;
			el[ d ].first = seen;
			el[ d ].lo = q;
			while( q < line.size() && !isspace( ( unsigned char )line[ q ] ) ){
				if( line[ q ] != '.' )
					seen++;
				q++;
			}
			el[ d ].hi = q;
		}
		int	nb = 0;
		for( size_t i = seq0; i < line.size(); i++ )
			nb += isalpha( ( unsigned char )line[ i ] ) ? 1 : 0;
		if( rnaviz )
			printf( "%5d dG = %.3f \"%s %d %d %d\"\n", nb, energy, name.c_str(), comp, off, len );
		else
			printf( "%4d %s %d %d %d\n", nb, name.c_str(), comp, off, len );
		int	done = 0;
		for( int d = 0; d < n; d++ ){
			const Elem	&e = el[ d ];
			if( e.hi <= e.lo || line[ e.lo ] == '.' )
				continue;
			const int	elen = int( e.hi - e.lo );
			for( int k = 0; k < elen; k++ ){
				const int	bn = done + k + 1;
				if( rnaviz )
					printf( "%5d %c %7d", bn, toupper( ( unsigned char )line[ e.lo + k ] ), bn - 1 );
				else
					printf( "%4d %c %4d", bn, line[ e.lo + k ], bn - 1 );
				printf( " %4d %4d %4d\n", bn == nb ? 0 : bn + 1, partner( el, e, k ), comp ? off - bn + 1 : off + bn - 1 );
			}
			done += elen;
		}
	}
	if( fp != stdin )
		fclose( fp );
	return 0;
}

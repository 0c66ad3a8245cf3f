// rmfmt -- tidy rnamotif output: sort the hits (best score first when the score is one
// number, else by entry and position) and line the columns up, or (-a) write each hit
// as a FASTA record with its elements padded to common widths.  Command line and
// output follow the reference's tool (/root/reference/src/rmfmt.c); where that one
// shells out to sort(1), the same ordering is done in memory, as sort does in the
// C locale: keys "-k 2rn,2 -k 1,1 -k 3n,3 -k 4n,4 -k 5n,5" for scored output,
// "-k 1,1" and the three numeric columns after the score fields otherwise, ties
// broken by the whole line.
//
//   usage: rmfmt [ -a ] [ -l[a] ] [ -smax N ] [ -td dir ] [ rm-output-file ]
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

const char	*USAGE = "usage: %s [ -a ] [ -l[a] ] [ -smax N ] [ -td dir ] [ rm-output-file ]\n";
const int	FIRST_ELEM = 4;		// name comp pos len, then the elements
const size_t	WIDE = 20;		// longer elements are abbreviated (not with -a)
const size_t	WRAP = 70;

enum { NAME_WHOLE, NAME_LOCUS, NAME_ACC };

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

bool is_a_number( const std::string &s )	// rmfmt.c:431-463
{
	const char	*p = s.c_str();
	if( *p == '-' )
		p++;
	int	mant = 0, expo = 0;
	bool	efmt = false;
	for( ; isdigit( ( unsigned char )*p ); p++ )
		mant++;
	if( *p == '.' )
		p++;
	for( ; isdigit( ( unsigned char )*p ); p++ )
		mant++;
	if( *p == 'e' || *p == 'E' ){
		efmt = true;
		p++;
		if( *p == '-' )
			p++;
		for( ; isdigit( ( unsigned char )*p ); p++ )
			expo++;
	}
	return mant > 0 && !( efmt && expo == 0 ) && *p == '\0';
}

// what sort -n reads from a field: blanks, '-', digits, '.', digits
double sort_number( const std::string &s )
{
	const char	*p = s.c_str();
	while( *p == ' ' || *p == '\t' )
		p++;
	const char	*q = p;
	if( *q == '-' )
		q++;
	while( isdigit( ( unsigned char )*q ) )
		q++;
	if( *q == '.' ){
		q++;
		while( isdigit( ( unsigned char )*q ) )
			q++;
	}
	return q > p ? strtod( std::string( p, size_t( q - p ) ).c_str(), nullptr ) : 0.0;
}

// the entry name as asked for with -l / -la: the text after the last '|', or between
// the last two (rmfmt.c:191-216)
void trim_name( std::string &name, int mode )
{
	size_t	last = name.rfind( '|' );
	if( mode == NAME_WHOLE || last == std::string::npos )
		return;
	size_t	prev = last == 0 ? std::string::npos : name.rfind( '|', last - 1 );
	if( mode == NAME_LOCUS ){
		if( last + 1 < name.size() )
			name = name.substr( last + 1 );
		else if( prev != std::string::npos )
			name = name.substr( prev + 1, last - prev - 1 );
		else
			name = name.substr( 0, last );
	}else if( prev != std::string::npos )
		name = name.substr( prev + 1, last - prev - 1 );
}

struct Layout {
	std::vector<bool>	right;		// per output column: right justified
	std::vector<size_t>	width;
	size_t	score_width = 0;
	int	n_cols = 0;			// name comp pos len + elements
	int	max_scores = 0;
};

struct Parts {
	std::vector<std::string>	col;	// name, comp, pos, len, elements
	std::vector<std::string>	score;
};

Parts take_apart( const std::vector<std::string> &w, int n_cols )
{
	Parts	p;
	const int	n = int( w.size() );
	const int	elem0 = n - ( n_cols - FIRST_ELEM ), comp = elem0 - 3;
	if( n > 0 )
		p.col.push_back( w[ 0 ] );
	for( int f = std::max( comp, 0 ); f < n; f++ )
		p.col.push_back( w[ f ] );
	for( int f = 1; f < comp && f < n; f++ )
		p.score.push_back( w[ f ] );
	return p;
}

void put_col( const std::string &s, size_t width, bool right, bool lead )
{
	const size_t	pad = width > s.size() ? width - s.size() : 0;
	if( right ){
		putchar( ' ' );
		for( size_t i = 0; i < pad; i++ )
			putchar( ' ' );
		fputs( s.c_str(), stdout );
	}else{
		if( lead )
			putchar( ' ' );
		fputs( s.c_str(), stdout );
		for( size_t i = 0; i < pad; i++ )
			putchar( ' ' );
	}
}

}	// namespace

int main( int argc, char **argv )
{
	const char	*fname = nullptr, *tdir = nullptr;
	bool	fasta = false;
	int	name_mode = NAME_WHOLE;
	long	smax = 30000000;
	for( int ac = 1; ac < argc; ac++ ){
		if( !strcmp( argv[ ac ], "-a" ) )
			fasta = true;
		else if( !strcmp( argv[ ac ], "-l" ) )
			name_mode = NAME_LOCUS;
		else if( !strcmp( argv[ ac ], "-la" ) )
			name_mode = NAME_ACC;
		else if( !strcmp( argv[ ac ], "-smax" ) ){
			if( ++ac >= argc ){
				fprintf( stderr, USAGE, argv[ 0 ] );
				return 1;
			}
			smax = atol( argv[ ac ] );
		}else if( !strcmp( argv[ ac ], "-td" ) ){
			if( ++ac >= argc || tdir != nullptr ){
				fprintf( stderr, USAGE, argv[ 0 ] );
				return 1;
			}
			tdir = argv[ ac ];	// (nothing is written to disk here)
		}else if( argv[ ac ][ 0 ] == '-' || fname != nullptr ){
			fprintf( stderr, USAGE, argv[ 0 ] );
			return 1;
		}else
			fname = argv[ ac ];
	}
	FILE	*fp = fname ? fopen( fname, "r" ) : stdin;
	if( fp == nullptr ){
		fprintf( stderr, "rmfmt: can't read rm-output-file '%s\n", fname );
		return 1;
	}

	// header: up to the first '>' line
	Layout	lay;
	std::string	line, pending;
	while( read_line( fp, line ) ){
		if( line[ 0 ] == '>' ){
			pending = line;
			break;
		}
		std::vector<std::string>	w = words( line );
		if( w.size() < 2 || w[ 0 ] != "#RM" )
			continue;
		if( w[ 1 ] == "descr" ){
			lay.n_cols = int( w.size() ) - 2 + FIRST_ELEM;
			lay.right.assign( size_t( lay.n_cols ), false );
			for( int f = 1; f < FIRST_ELEM; f++ )
				lay.right[ f ] = true;
			for( size_t f = 2; f < w.size(); f++ ){
				const std::string	t = w[ f ].substr( 0, 2 );
				lay.right[ f - 2 + FIRST_ELEM ] = t == "h3" || t == "t2" || t == "q2" || t == "q4";
			}
		}
		if( !fasta )
			fputs( line.c_str(), stdout );
	}
	lay.width.assign( size_t( lay.n_cols ), 0 );

	// body
	struct Row { std::string text; bool def; };
	std::vector<Row>	rows;
	bool	scored = true, sortable = true;
	int	first_scores = 0;
	size_t	bytes = 0;
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
			continue;		// several runs may be concatenated
		if( line[ 0 ] == '>' ){
			if( fasta )
				rows.push_back( Row{ line, true } );
			continue;
		}
		std::vector<std::string>	w = words( line );
		if( w.empty() || lay.n_cols == 0 )
			continue;
		Parts	p = take_apart( w, lay.n_cols );
		const int	n_scores = int( p.score.size() );
		lay.max_scores = std::max( lay.max_scores, n_scores );
		if( n_scores > 1 )
			scored = false;
		if( first_scores == 0 )
			first_scores = n_scores;
		else if( n_scores != lay.max_scores )
			sortable = false;
		if( scored )
			scored = n_scores >= 1 && is_a_number( p.score[ 0 ] );
		trim_name( p.col[ 0 ], name_mode );
		size_t	sw = 0;
		for( const std::string &s : p.score )
			sw += s.size();
		sw += n_scores > 0 ? size_t( n_scores - 1 ) : 0;
		lay.score_width = std::max( lay.score_width, sw );
		for( size_t f = 0; f < p.col.size() && f < lay.width.size(); f++ ){
			if( f >= size_t( FIRST_ELEM ) && !fasta && p.col[ f ].size() > WIDE ){
				char	buf[ 64 ];
				snprintf( buf, sizeof( buf ), "...(%d)...", int( p.col[ f ].size() ) );
				p.col[ f ] = p.col[ f ].substr( 0, 3 ) + buf + p.col[ f ].substr( p.col[ f ].size() - 3 );
			}
			lay.width[ f ] = std::max( lay.width[ f ], p.col[ f ].size() );
		}
		std::string	t = p.col.empty() ? std::string() : p.col[ 0 ];
		for( const std::string &s : p.score )
			t += " " + s;
		for( size_t f = 1; f < p.col.size(); f++ )
			t += " " + p.col[ f ];
		t += "\n";
		bytes += t.size();
		rows.push_back( Row{ t, false } );
	}
	if( fp != stdin )
		fclose( fp );

	if( !fasta ){
		if( long( bytes ) > smax )
			scored = sortable = false;
		struct Key { std::string name; double num[ 4 ]; const std::string *text; };
		std::vector<Key>	keys;
		keys.reserve( rows.size() );
		for( const Row &r : rows ){
			std::vector<std::string>	w = words( r.text );
			Key	k;
			k.name = w.empty() ? std::string() : w[ 0 ];
			k.text = &r.text;
			auto field = [&]( int f ) -> double { return f >= 1 && f <= int( w.size() ) ? sort_number( w[ f - 1 ] ) : 0.0; };
			const int	ns = int( w.size() ) - ( lay.n_cols - FIRST_ELEM ) - FIRST_ELEM;
			if( scored ){
				k.num[ 0 ] = field( 2 );
				k.num[ 1 ] = field( 3 ); k.num[ 2 ] = field( 4 ); k.num[ 3 ] = field( 5 );
			}else{
				k.num[ 0 ] = 0;
				k.num[ 1 ] = field( ns + 2 ); k.num[ 2 ] = field( ns + 3 ); k.num[ 3 ] = field( ns + 4 );
			}
			keys.push_back( k );
		}
		if( scored || sortable ){
			std::stable_sort( keys.begin(), keys.end(), [&]( const Key &a, const Key &b ){
				if( scored && a.num[ 0 ] != b.num[ 0 ] )
					return a.num[ 0 ] > b.num[ 0 ];
				if( int c = a.name.compare( b.name ) )
					return c < 0;
				for( int i = 1; i < 4; i++ )
					if( a.num[ i ] != b.num[ i ] )
						return a.num[ i ] < b.num[ i ];
				return *a.text < *b.text;
			} );
		}
		for( const Key &k : keys ){
			Parts	p = take_apart( words( *k.text ), lay.n_cols );
			if( p.col.empty() )
				continue;
			put_col( p.col[ 0 ], lay.width[ 0 ], lay.right[ 0 ], false );
			putchar( ' ' );
			if( lay.max_scores == 1 ){
				const std::string	s = p.score.empty() ? std::string() : p.score[ 0 ];
				for( size_t i = s.size(); i < lay.score_width; i++ )
					putchar( ' ' );
				fputs( s.c_str(), stdout );
			}else{
				size_t	used = 0;
				for( size_t f = 0; f < p.score.size(); f++ ){
					if( f ){
						putchar( ' ' );
						used++;
					}
					fputs( p.score[ f ].c_str(), stdout );
					used += p.score[ f ].size();
				}
				for( ; used < lay.score_width; used++ )
					putchar( ' ' );
			}
			for( size_t f = 1; f < p.col.size() && f < lay.width.size(); f++ )
				put_col( p.col[ f ], lay.width[ f ], lay.right[ f ], true );
			putchar( '\n' );
		}
		return 0;
	}

	// -a: one FASTA record per hit, elements separated by '|' and padded with '-'
	std::string	last_name, def_line;
	int	repeat = 1;
	for( const Row &r : rows ){
		if( r.def ){
			def_line = r.text;
			continue;
		}
		if( def_line.empty() )
			continue;
		// the definition: what follows the entry name on the '>' line (rmfmt.c:489-497)
		const char	*def = def_line.c_str() + 1;
		while( *def && isspace( ( unsigned char )*def ) )
			def++;
		if( *def ){
			def = strchr( def, ' ' );
			if( def )
				while( isspace( ( unsigned char )*def ) )
					def++;
		}
		Parts	p = take_apart( words( r.text ), lay.n_cols );
		if( p.col.size() < size_t( FIRST_ELEM ) ){
			def_line.clear();
			continue;
		}
		std::string	name = name_mode != NAME_WHOLE ?
			p.col[ 0 ] + "_" + p.col[ 2 ] + ( p.col[ 1 ][ 0 ] == '0' ? "d" : "c" ) :
			p.col[ 0 ] + "_" + p.col[ 1 ] + "_" + p.col[ 2 ] + "_" + p.col[ 3 ];
		std::string	ver;
		if( name == last_name ){
			repeat++;
			ver = ( name_mode == NAME_WHOLE ? "_" : "" ) + std::to_string( repeat );
		}else
			repeat = 1;
		printf( ">%s%s %s", name.c_str(), ver.c_str(), p.score.empty() ? "" : p.score[ 0 ].c_str() );
		for( size_t f = 1; f < p.score.size(); f++ )
			printf( " %s", p.score[ f ].c_str() );
		printf( " %s", def ? def : "(null)" );
		last_name = name;
		std::string	work;
		for( size_t f = FIRST_ELEM; f < p.col.size() && f < lay.width.size(); f++ ){
			if( f != size_t( FIRST_ELEM ) )
				work += '|';
			const std::string	pad( lay.width[ f ] > p.col[ f ].size() ? lay.width[ f ] - p.col[ f ].size() : 0, '-' );
			work += lay.right[ f ] ? pad + p.col[ f ] : p.col[ f ] + pad;
		}
		for( size_t b = 0; b < work.size(); b += WRAP )
			printf( "%s\n", work.substr( b, WRAP ).c_str() );
		def_line.clear();
	}
	return 0;
}

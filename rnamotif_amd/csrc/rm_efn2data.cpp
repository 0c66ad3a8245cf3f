// rm_efn2data.cpp -- loaders for the sixteen mfold-3.1 tables efn2() reads.
// Follows RM_getefn2data and its readers, /root/reference/src/efn2.c:130-1083
// (order of the files :130-187; miscloop :189, loop :387, dangle :443, the 4-index
// stacking files :516, coaxial :579, tstack :642, tloop/triloop :702/:746, int11 :789,
// int21 :870, int22 :958; packloop :1050 packs in base 5 here, not base 8 as efn's).
#include "rm_host.h"
#include "rm_efndata.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace rma {

namespace {

inline int nint( double x ) { return int( x >= 0 ? x + .5 : x - .5 ); }	// NINT, efn2.c:18

struct Reader {
	FILE	*fp = nullptr;
	char	line[ 512 ];
	std::string	path;
	~Reader() { if( fp ) fclose( fp ); }
	bool	open( const std::string &dir, const char *name, const char *what, std::string &err )
	{
		path = dir + "/" + name;
		fp = fopen( path.c_str(), "r" );
		if( fp == nullptr ){
			err += std::string( "can't read " ) + what + " file '" + path + "'.\n";
			return false;
		}
		return true;
	}
	bool	gets() { return fgets( line, 256, fp ) != nullptr; }	// the reference's line[ 256 ]
	bool	skipto( const char *str )	// efn2.c:1085
	{
		while( gets() )
			if( strstr( line, str ) )
				return true;
		return false;
	}
	std::vector<std::string> fields()	// split( line, fields, " \t\n" )
	{
		std::vector<std::string>	f;
		const char	*sp = line;
		for( ; ; ){
			sp += strspn( sp, " \t\n" );
			if( !*sp )
				break;
			size_t	n = strcspn( sp, " \t\n" );
			f.emplace_back( sp, n );
			sp += n;
		}
		return f;
	}
};

// the six closing pairs of the interior loop files: a:u c:g g:c u:a g:u u:g (efn2.c:92-95)
const int	bmap[ 6 ] = { RMA_BC_A, RMA_BC_C, RMA_BC_G, RMA_BC_T, RMA_BC_G, RMA_BC_T };
const int	rmap[ 6 ] = { RMA_BC_T, RMA_BC_G, RMA_BC_C, RMA_BC_A, RMA_BC_T, RMA_BC_G };

int packloop5( const char *loop )	// efn2.c:1050
{
	int	num = 0;
	for( int i = int( strlen( loop ) ) - 1; i >= 0; i-- ){
		int	bc;
		switch( loop[ i ] ){
		case 'A' : case 'a' : bc = RMA_BC_A; break;
		case 'C' : case 'c' : bc = RMA_BC_C; break;
		case 'G' : case 'g' : bc = RMA_BC_G; break;
		case 'T' : case 't' : case 'U' : case 'u' : bc = RMA_BC_T; break;
		default : fail( "illegal char %c (%d) in efn2 loop table", loop[ i ], loop[ i ] );
		}
		num = num * 5 + bc;
	}
	return num;
}

int val100( const std::string &f, int dot )
{
	return f[ 0 ] == '.' ? dot : nint( 100.0 * atof( f.c_str() ) );
}

// getstack :516 (first index = block) and getcoax :579 (first two indices swapped)
bool get_stack4( const std::string &dir, const char *name, int32_t st[ 5 ][ 5 ][ 5 ][ 5 ], bool swap12, std::string &err )
{
	Reader	r;
	if( !r.open( dir, name, "stack", err ) )
		return false;
	memset( st, 0, 625 * sizeof( int32_t ) );
	for( int v1 = 0; v1 < 4; v1++ ){
		if( !r.skipto( "<--" ) ){
			err += "premature end of stack file '" + r.path + "'.\n";
			return false;
		}
		for( int v3 = 0; v3 < 4; v3++ ){
			r.gets();
			std::vector<std::string>	f = r.fields();
			for( size_t k = 0; k < f.size() && k < 16; k++ ){
				const int	v2 = int( k ) / 4, v4 = int( k ) % 4;
				const int	v = val100( f[ k ], RMA_EFN2_INFINITY );
				if( swap12 )
					st[ v2 ][ v1 ][ v3 ][ v4 ] = v;
				else
					st[ v1 ][ v2 ][ v3 ][ v4 ] = v;
			}
		}
	}
	return true;
}

bool get_tloops( const std::string &dir, const char *name, const char *what, int32_t tab[][ 2 ], int32_t *n, std::string &err )
{
	Reader	r;
	if( !r.open( dir, name, what, err ) )
		return false;
	int	t = 0;
	bool	ok = true;
	if( r.skipto( "---" ) ){
		char	loop[ 256 ] = "";
		float	energy = 0;
		for( ; r.gets(); t++ ){
			sscanf( r.line, "%255s %f", loop, &energy );
			if( t < RMA_EFN2_MAXTLOOP ){
				tab[ t + 1 ][ 0 ] = packloop5( loop );
				tab[ t + 1 ][ 1 ] = nint( 100.0 * energy );
			}
		}
	}else
		ok = false;
	*n = t > RMA_EFN2_MAXTLOOP ? RMA_EFN2_MAXTLOOP : t;
	return ok;
}

}	// namespace

bool load_efn2data( const std::string &dir, rma_efn2data_t *ed, std::string &err )
{
	memset( ed, 0, sizeof( *ed ) );
	if( dir.empty() ){
		err += "No efn data directory.\n";
		return false;
	}
	{	// getmiscloop :189-385
		Reader	r;
		if( !r.open( dir, "miscloop.dat", "miscloop", err ) )
			return false;
		float	f1 = 0, f2 = 0, f3 = 0, f4 = 0;
		auto want = [&]( const char *what ) -> bool {
			if( !r.skipto( "-->" ) ){
				err += std::string( "no " ) + what + ".\n";
				return false;
			}
			r.gets();
			return true;
		};
		if( !want( "prolog" ) )
			return false;
		sscanf( r.line, "%f", &ed->prelog );
		ed->prelog *= 10.0f;
		if( !want( "paxpen" ) )
			return false;
		sscanf( r.line, "%f", &f1 );
		ed->maxpen = nint( 100.0 * f1 );
		if( !want( "poppen values" ) )
			return false;
		sscanf( r.line, "%f %f %f %f", &f1, &f2, &f3, &f4 );
		ed->poppen[ 0 ] = 0;
		ed->poppen[ 1 ] = nint( 100.0 * f1 );
		ed->poppen[ 2 ] = nint( 100.0 * f2 );
		ed->poppen[ 3 ] = nint( 100.0 * f3 );
		ed->poppen[ 4 ] = nint( 100.0 * f4 );
		ed->eparam[ 7 ] = 30;
		ed->eparam[ 8 ] = 30;
		ed->eparam[ 9 ] = -500;
		if( !want( "multibranched loop values" ) )
			return false;
		sscanf( r.line, "%f %f %f", &f1, &f2, &f3 );
		ed->eparam[ 5 ] = nint( 100.0 * f1 );
		ed->eparam[ 6 ] = nint( 100.0 * f2 );
		ed->eparam[ 10 ] = nint( 100.0 * f3 );
		if( !r.skipto( "-->" ) ){
			ed->eparam[ 9 ] = ed->eparam[ 10 ] = 0;
		}else{
			r.gets();
			sscanf( r.line, "%f %f %f", &f1, &f2, &f3 );
			ed->efn2a = nint( 100.0 * f1 );
			ed->efn2b = nint( 100.0 * f2 );
			ed->efn2c = nint( 100.0 * f3 );
			int32_t	*dst[] = { &ed->auend, &ed->gubonus, &ed->cslope, &ed->cint, &ed->c3, &ed->init };
			const char	*what[] = { "terminal AU pernalty", "GGG hairpin term", "c hairpin slope",
				"c hairpin intercept", "c hairpin of 3 term", "Intermol init free energy" };
			for( int k = 0; k < 6; k++ ){
				if( !want( what[ k ] ) )
					return false;
				sscanf( r.line, "%f", &f1 );
				*dst[ k ] = nint( 100.0 * f1 );
			}
			if( !want( "GAIL Rule term" ) )
				return false;
			sscanf( r.line, "%d", &ed->gail );
		}
	}
	{	// getibhloop :387-441
		Reader	r;
		if( !r.open( dir, "loop.dat", "ibhloop", err ) )
			return false;
		if( !r.skipto( "---" ) ){
			err += "error in ibhloop file 'loop.dat'.\n";
			return false;
		}
		for( int i = 1; i <= 30; i++ ){
			r.gets();
			std::vector<std::string>	f = r.fields();
			if( f.size() < 4 )
				continue;
			ed->inter[ i ] = val100( f[ 1 ], RMA_EFN2_INFINITY );
			ed->bulge[ i ] = val100( f[ 2 ], RMA_EFN2_INFINITY );
			ed->hairpin[ i ] = val100( f[ 3 ], RMA_EFN2_INFINITY );
		}
	}
	{	// getdangle :443-514 (entries with an N stay 0)
		Reader	r;
		if( !r.open( dir, "dangle.dat", "dangle", err ) )
			return false;
		for( int v4 = 0; v4 < 2; v4++ )
			for( int v1 = 0; v1 < 4; v1++ ){
				if( !r.skipto( "<--" ) ){
					err += "premature end of dangle file 'dangle.dat'.\n";
					return false;
				}
				r.gets();
				std::vector<std::string>	f = r.fields();
				for( size_t k = 0; k < f.size() && k < 16; k++ )
					ed->dangle[ v1 ][ k / 4 ][ k % 4 ][ v4 ] = val100( f[ k ], RMA_EFN2_INFINITY );
			}
	}
	if( !get_stack4( dir, "stack.dat", ed->stack, false, err ) ) return false;
	if( !get_stack4( dir, "tstackh.dat", ed->tstkh, false, err ) ) return false;
	if( !get_stack4( dir, "tstacki.dat", ed->tstki, false, err ) ) return false;
	if( !get_stack4( dir, "coaxial.dat", ed->coax, true, err ) ) return false;
	if( !get_stack4( dir, "tstackcoax.dat", ed->tstackcoax, false, err ) ) return false;
	if( !get_stack4( dir, "coaxstack.dat", ed->coaxstack, false, err ) ) return false;
	if( !get_stack4( dir, "tstackm.dat", ed->tstkm, false, err ) ) return false;
	{	// gettstack :642-700: one "<--" per non-N first index, one line per non-N third index
		Reader	r;
		if( !r.open( dir, "tstack.dat", "stack", err ) )
			return false;
		std::vector<std::string>	f;
		for( int v1 = 0; v1 < 5; v1++ ){
			if( v1 != RMA_BC_N && !r.skipto( "<--" ) ){
				err += "premature end of stack file '" + r.path + "'.\n";
				return false;
			}
			for( int v3 = 0; v3 < 5; v3++ ){
				if( v1 != RMA_BC_N && v3 != RMA_BC_N ){
					r.gets();
					f = r.fields();
				}
				for( int v2 = 0; v2 < 5; v2++ )
					for( int v4 = 0; v4 < 5; v4++ ){
						if( v1 == RMA_BC_N || v2 == RMA_BC_N || v3 == RMA_BC_N || v4 == RMA_BC_N )
							ed->tstack[ v1 ][ v2 ][ v3 ][ v4 ] = 0;
						else{
							const size_t	k = size_t( 4 * v2 + v4 );
							ed->tstack[ v1 ][ v2 ][ v3 ][ v4 ] = k < f.size() ? val100( f[ k ], RMA_EFN2_INFINITY ) : 0;
						}
					}
			}
		}
	}
	if( !get_tloops( dir, "tloop.dat", "tloop", ed->tloop, &ed->ntloops, err ) ) return false;
	if( !get_tloops( dir, "triloop.dat", "triloops", ed->triloop, &ed->ntriloops, err ) ) return false;
	{	// get1x1loop :789-868
		Reader	r;
		if( !r.open( dir, "int11.dat", "1x1 loop", err ) )
			return false;
		if( !r.skipto( "<--" ) ){
			err += "error in 1x1 file '" + r.path + "'.\n";
			return false;
		}
		for( int v1 = 0; v1 < 6; v1++ ){
			if( !r.skipto( "<--" ) ){
				err += "premature end of 1x1 loop file '" + r.path + "'.\n";
				return false;
			}
			const int	a = bmap[ v1 ], d = rmap[ v1 ];
			for( int b = 0; b < 4; b++ ){
				r.gets();
				std::vector<std::string>	f = r.fields();
				for( size_t fc = 0; fc < f.size() && fc < 24; fc++ ){
					const int	v2 = int( fc ) / 4, e = int( fc ) % 4;
					ed->iloop11[ a ][ b ][ bmap[ v2 ] ][ d ][ e ][ rmap[ v2 ] ] = nint( 100.0 * atof( f[ fc ].c_str() ) );
				}
			}
		}
	}
	{	// get2x1loop :870-956
		Reader	r;
		if( !r.open( dir, "int21.dat", "2x1 loop", err ) )
			return false;
		for( int32_t *p = &ed->iloop21[ 0 ][ 0 ][ 0 ][ 0 ][ 0 ][ 0 ][ 0 ], *e = p + 78125; p < e; p++ )
			*p = RMA_EFN2_INFINITY;
		if( !r.skipto( "<--" ) ){
			err += "error in 2x1 loop file '" + r.path + "'.\n";
			return false;
		}
		for( int v1 = 0; v1 < 6; v1++ ){
			const int	a = bmap[ v1 ], b = rmap[ v1 ];
			for( int e = 0; e < 4; e++ ){
				if( !r.skipto( "<--" ) ){
					err += "premature end of 2x1 loop file '" + r.path + "'.\n";
					return false;
				}
				for( int c = 0; c < 4; c++ ){
					r.gets();
					std::vector<std::string>	f = r.fields();
					size_t	fc = 0;
					for( int v4 = 0; v4 < 6; v4++ )
						for( int d = 0; d < 4; d++, fc++ )
							if( fc < f.size() )
								ed->iloop21[ a ][ b ][ c ][ d ][ e ][ bmap[ v4 ] ][ rmap[ v4 ] ] = nint( 100.0 * atof( f[ fc ].c_str() ) );
				}
			}
		}
	}
	{	// get2x2loop :958-1048
		Reader	r;
		if( !r.open( dir, "int22.dat", "2x2 loop", err ) )
			return false;
		for( int32_t *p = &ed->iloop22[ 0 ][ 0 ][ 0 ][ 0 ][ 0 ][ 0 ][ 0 ][ 0 ], *e = p + 390625; p < e; p++ )
			*p = RMA_EFN2_INFINITY;
		if( !r.skipto( "<--" ) ){
			err += "error in 2x2 loop file '" + r.path + "'.\n";
			return false;
		}
		for( int v1 = 0; v1 < 6; v1++ ){
			const int	a = bmap[ v1 ], c = rmap[ v1 ];
			for( int v2 = 0; v2 < 6; v2++ ){
				if( !r.skipto( "<--" ) ){
					err += "premature end of 2x2 loop file '" + r.path + "'.\n";
					return false;
				}
				const int	b = bmap[ v2 ], d = rmap[ v2 ];
				for( int j = 0; j < 4; j++ )
					for( int k = 0; k < 4; k++ ){
						r.gets();
						std::vector<std::string>	f = r.fields();
						for( size_t x = 0; x < f.size() && x < 16; x++ )
							ed->iloop22[ a ][ b ][ c ][ d ][ j ][ x / 4 ][ k ][ x % 4 ] = nint( 100.0 * atof( f[ x ].c_str() ) );
					}
			}
		}
	}
	// the two logarithms, in the reference's arithmetic: float prelog times double log,
	// truncated (efn2.c:1570); 11.*log( n/6. ) + 0.5 truncated (efn2.c:1209)
	for( int n = 0; n < RMA_EFN_LOGINC; n++ ){
		ed->loginc[ n ] = n > 0 ? int( ed->prelog * log( double( n ) / 30.0 ) ) : 0;
		ed->mbl_log[ n ] = n > 0 ? int( 11. * log( double( n / 6. ) ) + 0.5 ) : 0;
	}
	return true;
}

}	// namespace rma

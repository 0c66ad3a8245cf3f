// rm_efndata.cpp -- loaders for the mfold-3.1 nearest neighbour tables used
// by efn().  Follows RM_getefndata and its eleven readers in
// /root/reference/src/efn.c:157-918 (packloop :920, skipto :954); the tables
// end up in one flat rma_efndata_t that is copied to the device.
#include "rm_host.h"
#include "rm_efndata.h"
#include "rm_efn_core.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace rma {

namespace {

inline int nint( double x ) { return int( x >= 0 ? x + .5 : x - .5 ); }	// NINT, efn.c:25

struct Reader {
	FILE	*fp = nullptr;
	char	line[ 256 ];
	std::string	path;
	~Reader() { if( fp ) fclose( fp ); }
	bool	open( const std::string &dir, const char *name, std::string &err )
	{
		path = dir + "/" + name;
		fp = fopen( path.c_str(), "r" );
		if( fp == nullptr ){
			err += "can't read efn data file '" + path + "'.\n";
			return false;
		}
		return true;
	}
	bool	gets() { return fgets( line, sizeof( line ), fp ) != nullptr; }
	bool	skipto( const char *str )	// efn.c:954
	{
		while( gets() )
			if( strstr( line, str ) )
				return true;
		return false;
	}
	// split( line, fields, " \t\n" ), split.c:26-48
	std::vector<std::string> fields()
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

int packloop( const char *loop )	// efn.c:920
{
	int	num = 0;
	for( int i = int( strlen( loop ) ) - 1; i >= 0; i-- ){
		int	bc;
		switch( loop[ i ] ){
		case 'A' : case 'a' : bc = RMA_BC_A; break;
		case 'C' : case 'c' : bc = RMA_BC_C; break;
		case 'G' : case 'g' : bc = RMA_BC_G; break;
		case 'T' : case 't' : case 'U' : case 'u' : bc = RMA_BC_T; break;
		default : fail( "illegal char %c (%d) in efn loop table", loop[ i ], loop[ i ] );
		}
		num = ( num << 3 ) + bc;
	}
	return num;
}

bool get_loops( const std::string &dir, const char *name, int maxn, int32_t ( *tab )[ 2 ], int32_t *n, std::string &err )
{
	Reader	r;
	if( !r.open( dir, name, err ) )
		return false;
	if( !r.skipto( "---" ) ){
		*n = 0;
		return false;
	}
	char	loop[ 256 ] = "";
	float	energy = 0;
	int	t = 0;
	for( ; r.gets(); t++ ){
		sscanf( r.line, "%255s %f", loop, &energy );	// a blank line repeats the last entry
		if( t < maxn ){
			tab[ t ][ 0 ] = packloop( loop );
			tab[ t ][ 1 ] = nint( 100.0 * energy );
		}
	}
	*n = t > maxn ? maxn : t;
	return true;
}

bool get_stack( const std::string &dir, const char *name, int32_t st[ 5 ][ 5 ][ 5 ][ 5 ], int defval, std::string &err )
{
	Reader	r;
	if( !r.open( dir, name, err ) )
		return false;
	for( int a = 0; a < 5; a++ )
		for( int b = 0; b < 5; b++ )
			for( int c = 0; c < 5; c++ )
				for( int d = 0; d < 5; d++ )
					st[ a ][ b ][ c ][ d ] = defval;
	for( int v1 = 0; v1 < 4; v1++ ){
		if( !r.skipto( "<--" ) ){
			err += "premature end of stack file '" + r.path + "'.\n";
			return false;
		}
		for( int v3 = 0; v3 < 4; v3++ ){
			r.gets();
			std::vector<std::string>	f = r.fields();
			for( size_t k = 0; k < f.size() && k < 16; k++ ){
				int	v2 = int( k ) / 4, v4 = int( k ) % 4;
				st[ v1 ][ v2 ][ v3 ][ v4 ] = f[ k ][ 0 ] == '.' ? RMA_EFN_INFINITY : nint( 100.0 * atof( f[ k ].c_str() ) );
			}
		}
	}
	return true;
}

}	// namespace

bool load_efndata( const std::string &dir, rma_efndata_t *ed, std::string &err )
{
	memset( ed, 0, sizeof( *ed ) );
	if( dir.empty() ){
		err += "No efn data directory.\n";
		return false;
	}
	bool	ok = true;
	ok &= get_loops( dir, "tloop.dat", 100, ed->tloops, &ed->ntloops, err );
	ok &= get_loops( dir, "triloop.dat", 50, ed->triloops, &ed->ntriloops, err );

	{	// getmiscloop, efn.c:290-465
		Reader	r;
		if( !r.open( dir, "miscloop.dat", err ) )
			ok = false;
		else{
			float	f1 = 0, f2 = 0, f3 = 0, f4 = 0;
			auto nextval = [&]( const char *what ) -> bool {
				if( !r.skipto( "-->" ) ){
					err += std::string( "miscloop: no " ) + what + ".\n";
					return false;
				}
				r.gets();
				return true;
			};
			bool	mok = true;
			if( ( mok = nextval( "prelog" ) ) ){
				sscanf( r.line, "%f", &ed->prelog );
				ed->prelog *= 10.0;
			}
			if( mok && ( mok = nextval( "maxpen" ) ) ){
				sscanf( r.line, "%f", &f1 );
				ed->maxpen = nint( 100.0 * f1 );
			}
			if( mok && ( mok = nextval( "poppen values" ) ) ){
				sscanf( r.line, "%f %f %f %f", &f1, &f2, &f3, &f4 );
				ed->poppen[ 0 ] = 0;
				ed->poppen[ 1 ] = nint( 100.0 * f1 );
				ed->poppen[ 2 ] = nint( 100.0 * f2 );
				ed->poppen[ 3 ] = nint( 100.0 * f3 );
				ed->poppen[ 4 ] = nint( 100.0 * f4 );
			}
			ed->eparam[ 6 ] = 30;
			ed->eparam[ 7 ] = 30;
			if( mok && ( mok = nextval( "multibranched loop values" ) ) ){
				sscanf( r.line, "%f %f %f", &f1, &f2, &f3 );
				ed->eparam[ 4 ] = nint( 100.0 * f1 );
				ed->eparam[ 5 ] = nint( 100.0 * f2 );
				ed->eparam[ 8 ] = nint( 100.0 * f3 );
			}
			if( mok && r.skipto( "-->" ) ){
				r.gets();	// efn2 multibranch terms, unused by efn
				struct { const char *what; int32_t *dst; } terms[] = {
					{ "terminal AU penalty", &ed->auend }, { "GGG hairpin term", &ed->gubonus },
					{ "c hairpin slope", &ed->cslope }, { "c hairpin intercept", &ed->cint },
					{ "c hairpin of 3 term", &ed->c3 }, { "Intermol init free energy", &ed->init } };
				for( auto &t : terms ){
					if( !( mok = nextval( t.what ) ) )
						break;
					sscanf( r.line, "%f", &f1 );
					*t.dst = nint( 100.0 * f1 );
				}
				if( mok && ( mok = nextval( "GAIL Rule term" ) ) )
					sscanf( r.line, "%d", &ed->gail );
			}
			ok &= mok;
		}
	}

	{	// getdangle, efn.c:467-515
		Reader	r;
		if( !r.open( dir, "dangle.dat", err ) )
			ok = false;
		else{
			for( int v4 = 0; v4 < 2 && ok; v4++ ){
				for( int v1 = 0; v1 < 4; v1++ ){
					if( !r.skipto( "<--" ) ){
						err += "premature end of dangle file\n";
						ok = false;
						break;
					}
					r.gets();
					std::vector<std::string>	f = r.fields();
					for( size_t k = 0; k < f.size() && k < 16; k++ ){
						int	v2 = int( k ) / 4, v3 = int( k ) % 4;
						ed->dangle[ v1 ][ v2 ][ v3 ][ v4 ] = f[ k ][ 0 ] != '.' ?
							nint( 100.0 * atof( f[ k ].c_str() ) ) : RMA_EFN_INFINITY;
					}
				}
			}
		}
	}

	{	// getibhloop, efn.c:517-566
		Reader	r;
		if( !r.open( dir, "loop.dat", err ) )
			ok = false;
		else if( !r.skipto( "---" ) ){
			err += "error in ibhloop file\n";
			ok = false;
		}else{
			for( int i = 1; i <= RMA_EFN_MAXLOOP; i++ ){
				if( !r.gets() )
					break;
				std::vector<std::string>	f = r.fields();
				if( f.size() < 4 )
					continue;
				auto val = [&]( const std::string &s ){ return s[ 0 ] == '.' ? RMA_EFN_INFINITY : nint( 100.0 * atof( s.c_str() ) ); };
				ed->inter[ i ] = val( f[ 1 ] );
				ed->bulge[ i ] = val( f[ 2 ] );
				ed->hairpin[ i ] = val( f[ 3 ] );
			}
		}
	}

	ok &= get_stack( dir, "stack.dat", ed->stack, RMA_EFN_INFINITY, err );
	if( ok ){	// stacktest, efn.c:622-644
		for( int a = 0; a < 4; a++ ) for( int b = 0; b < 4; b++ ) for( int c = 0; c < 4; c++ ) for( int d = 0; d < 4; d++ )
			if( ed->stack[ a ][ b ][ c ][ d ] != ed->stack[ d ][ c ][ b ][ a ] ){
				err += "stack 'stack.dat' symmetry error\n";
				ok = false;
			}
	}
	ok &= get_stack( dir, "tstackh.dat", ed->tstkh, 0, err );
	ok &= get_stack( dir, "tstacki.dat", ed->tstki, 0, err );

	{	// getsymint, efn.c:646-772
		Reader	r;
		if( !r.open( dir, "sint2.dat", err ) || !r.skipto( "<--" ) )
			ok = false;
		else{
			for( int v1 = 0; v1 < 6 && ok; v1++ ){
				if( !r.skipto( "<--" ) ){
					err += "premature end of sym-2 loop file\n";
					ok = false;
					break;
				}
				for( int v3 = 0; v3 < 4; v3++ ){
					r.gets();
					std::vector<std::string>	f = r.fields();
					for( size_t k = 0; k < f.size() && k < 24; k++ )
						ed->sint2[ v1 ][ k / 4 ][ v3 ][ k % 4 ] = nint( 100.0 * atof( f[ k ].c_str() ) );
				}
			}
			for( int v1 = 0; v1 < 6; v1++ ){
				for( int v2 = 0; v2 < 6; v2++ ){
					int	worst = -999;
					for( int v3 = 0; v3 < 4; v3++ )
						for( int v4 = 0; v4 < 4; v4++ )
							worst = std::max( worst, ed->sint2[ v1 ][ v2 ][ v3 ][ v4 ] );
					for( int v3 = 0; v3 < 5; v3++ ){
						ed->sint2[ v1 ][ v2 ][ v3 ][ 4 ] = worst;
						ed->sint2[ v1 ][ v2 ][ 4 ][ v3 ] = worst;
					}
				}
			}
		}
		Reader	r4;
		if( !r4.open( dir, "sint4.dat", err ) || !r4.skipto( "<--" ) )
			ok = false;
		else{
			for( int v1 = 0; v1 < 6 && ok; v1++ ){
				for( int v2 = 0; v2 < 6; v2++ ){
					if( !r4.skipto( "<--" ) ){
						err += "premature end of sym-4 loop file\n";
						ok = false;
						break;
					}
					for( int v3 = 0; v3 < 4; v3++ ){
						for( int v4 = 0; v4 < 4; v4++ ){
							r4.gets();
							std::vector<std::string>	f = r4.fields();
							for( size_t k = 0; k < f.size() && k < 16; k++ )
								ed->sint4[ v1 ][ v2 ][ v3 ][ v4 ][ k / 4 ][ k % 4 ] = nint( 100.0 * atof( f[ k ].c_str() ) );
						}
					}
				}
			}
			for( int v1 = 0; v1 < 6; v1++ ){
				for( int v2 = 0; v2 < 6; v2++ ){
					int	worst = -999;
					for( int a = 0; a < 4; a++ ) for( int b = 0; b < 4; b++ ) for( int c = 0; c < 4; c++ ) for( int d = 0; d < 4; d++ )
						worst = std::max( worst, ed->sint4[ v1 ][ v2 ][ a ][ b ][ c ][ d ] );
					for( int a = 0; a < 5; a++ ) for( int b = 0; b < 5; b++ ) for( int c = 0; c < 5; c++ ){
						ed->sint4[ v1 ][ v2 ][ a ][ b ][ c ][ 4 ] = worst;
						ed->sint4[ v1 ][ v2 ][ a ][ b ][ 4 ][ c ] = worst;
						ed->sint4[ v1 ][ v2 ][ a ][ 4 ][ b ][ c ] = worst;
						ed->sint4[ v1 ][ v2 ][ 4 ][ a ][ b ][ c ] = worst;
					}
				}
			}
		}
	}
	if( ok ){	// symtest, efn.c:774-824
		for( int v1 = 0; v1 < 6; v1++ ){
			for( int v2 = 0; v2 < 6; v2++ ){
				int	v1a = v1 >= 4 ? 9 - v1 : 3 - v1, v2a = v2 >= 4 ? 9 - v2 : 3 - v2;
				for( int a = 0; a < 4; a++ ) for( int b = 0; b < 4; b++ ){
					if( ed->sint2[ v1 ][ v2 ][ a ][ b ] != ed->sint2[ v2a ][ v1a ][ b ][ a ] ){
						err += "sint2 symmetry failure\n";
						ok = false;
					}
					for( int c = 0; c < 4; c++ ) for( int d = 0; d < 4; d++ )
						if( ed->sint4[ v1 ][ v2 ][ a ][ b ][ c ][ d ] != ed->sint4[ v2a ][ v1a ][ d ][ c ][ b ][ a ] ){
							err += "sint4 symmetry failure\n";
							ok = false;
						}
				}
			}
		}
	}

	{	// getasymint, efn.c:826-885
		Reader	r;
		if( !r.open( dir, "asint1x2.dat", err ) || !r.skipto( "<--" ) )
			ok = false;
		else{
			for( int a = 0; a < 6; a++ ) for( int b = 0; b < 6; b++ ) for( int c = 0; c < 5; c++ ) for( int d = 0; d < 5; d++ ) for( int e = 0; e < 5; e++ )
				ed->asint1x2[ a ][ b ][ c ][ d ][ e ] = RMA_EFN_INFINITY;
			for( int v1 = 0; v1 < 6 && ok; v1++ ){
				for( int v5 = 0; v5 < 4; v5++ ){
					if( !r.skipto( "<--" ) ){
						err += "premature end of asym-1x2 loop file\n";
						ok = false;
						break;
					}
					for( int v3 = 0; v3 < 4; v3++ ){
						r.gets();
						std::vector<std::string>	f = r.fields();
						for( size_t k = 0; k < f.size() && k < 24; k++ )
							ed->asint1x2[ v1 ][ k / 4 ][ v3 ][ k % 4 ][ v5 ] = nint( 100.0 * atof( f[ k ].c_str() ) );
					}
				}
			}
		}
	}

	// NINT( prelog*log( size/30. ) ), efn.c:1375,1384,1572: float * double
	for( int n = 0; n < RMA_EFN_LOGINC; n++ )
		ed->loginc[ n ] = n > 30 ? nint( ed->prelog * log( n / 30.0 ) ) : 0;
	return ok;
}

std::string find_efndata_dir( Descriptor &d )	// score.c:1584-1590
{
	Ident	*ip = d.find_id( "efn_datadir" );
	const char	*cp = ip ? ( const char * )ip->val.pval : nullptr;
	if( cp != nullptr && *cp != '\0' )
		return cp;
	if( ( cp = getenv( "EFNDATA" ) ) != nullptr )
		return cp;
	return "";
}

// The tables as the energy kernel reads them (rm_efn_core.h: one flat int16 image, staged in LDS; the tetraloop
// keys, 18 bits, apart).  Also what tests/hostsim hands the same core compiled for the host.
void efn_tables16( const rma_efndata_t *ed, std::vector<int16_t> &t16, std::vector<int32_t> &tlkey )
{
	t16.assign( ( RME_N16 + 7 ) / 8 * 8, 0 );	// padded for 16-byte staging loads
	tlkey.assign( 100, -1 );
	auto put = [&]( int off, const int32_t *src, int n ){
		for( int i = 0; i < n; i++ ){
			int	v = src[ i ];
			t16[ off + i ] = int16_t( v > 32767 ? 32767 : v < -32768 ? -32768 : v );
		}
	};
	put( RME_INTER, ed->inter, 31 );
	put( RME_BULGE, ed->bulge, 31 );
	put( RME_HAIRPIN, ed->hairpin, 31 );
	put( RME_DANGLE, &ed->dangle[ 0 ][ 0 ][ 0 ][ 0 ], 250 );
	put( RME_POPPEN, ed->poppen, 5 );
	put( RME_EPARAM, ed->eparam, 16 );
	int32_t	misc[ 9 ] = { ed->maxpen, ed->auend, ed->gubonus, ed->cslope, ed->cint, ed->c3, ed->gail,
		ed->ntriloops, ed->ntloops };
	put( RME_MISC, misc, 9 );
	for( int k = 0; k < 50; k++ ){
		// a key that does not fit 15 bits can never equal a computed key's low part
		// by accident: store -1 (no computed key is negative)
		int	key = k < ed->ntriloops ? ed->triloops[ k ][ 0 ] : -1;
		t16[ RME_TRIKEY + k ] = int16_t( key >= 0 && key <= 32767 ? key : -1 );
		t16[ RME_TRIVAL + k ] = int16_t( k < ed->ntriloops ? ed->triloops[ k ][ 1 ] : 0 );
	}
	for( int k = 0; k < 100; k++ ){
		tlkey[ k ] = k < ed->ntloops ? ed->tloops[ k ][ 0 ] : -1;
		t16[ RME_TLVAL + k ] = int16_t( k < ed->ntloops ? ed->tloops[ k ][ 1 ] : 0 );
	}
	put( RME_STACK, &ed->stack[ 0 ][ 0 ][ 0 ][ 0 ], 625 );
	put( RME_TSTKH, &ed->tstkh[ 0 ][ 0 ][ 0 ][ 0 ], 625 );
	put( RME_TSTKI, &ed->tstki[ 0 ][ 0 ][ 0 ][ 0 ], 625 );
	put( RME_SINT2, &ed->sint2[ 0 ][ 0 ][ 0 ][ 0 ], 900 );
	put( RME_ASINT, &ed->asint1x2[ 0 ][ 0 ][ 0 ][ 0 ][ 0 ], 4500 );
	put( RME_SINT4, &ed->sint4[ 0 ][ 0 ][ 0 ][ 0 ][ 0 ][ 0 ], 22500 );
}


}	// namespace rma

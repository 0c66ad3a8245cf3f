/*
 * rnamotif_oracle_main.cpp -- TEST INFRASTRUCTURE (see rm_oracle.h).
 *
 * A test-only `rnamotif` executable: the product's host front end (descriptor
 * compiler, score VM, driver) wired to the scalar CPU oracle instead of the
 * HIP scanner.  It exists to pin the front end + oracle against the
 * reference's golden outputs in a container that has no GPU, and as the
 * cpu_baseline leg of bench.py.  It is never installed or loaded by the
 * package.
 */
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>
#include "rm_cli.h"
#include "rm_oracle.h"
#include "rm_pack.h"
#include <string>

namespace {

struct OracleBackend {
	const rma_program_t	*prog;
	const rma_efndata_t	*efn;
	rma_efndata_t	own_efn;
	rmo_hits_t	hits;
	std::vector<int32_t>	sorted;
};

int oracle_scan( void *self, const char *const *seqs, const int32_t *slens, int n,
	const int32_t **hits, int64_t *n_hits, char *err, size_t errlen )
{
	OracleBackend	*ob = ( OracleBackend * )self;
	rmo_hits_free( &ob->hits );
	rmo_hits_init( &ob->hits, ob->prog );
	std::vector<char>	buf;
	for( int s = 0; s < n; s++ ){
		buf.assign( seqs[ s ], seqs[ s ] + slens[ s ] );
		buf.push_back( '\0' );
		if( rmo_scan( ob->prog, ob->efn, s, buf.data(), slens[ s ], 0, &ob->hits ) ){
			snprintf( err, errlen, "oracle: helix candidate list overflow" );
			return 1;
		}
		if( ob->prog->chk_both_strs ){
			rmo_revcomp( buf.data(), slens[ s ] );
			if( rmo_scan( ob->prog, ob->efn, s, buf.data(), slens[ s ], 1, &ob->hits ) ){
				snprintf( err, errlen, "oracle: helix candidate list overflow" );
				return 1;
			}
		}
	}
	// the oracle emits in reference order already; sort anyway through the
	// boundary's key so the same comparator is exercised as on the device path
	int	stride = ob->hits.stride;
	std::vector<int64_t>	idx( ob->hits.n );
	std::iota( idx.begin(), idx.end(), 0 );
	const int32_t	*d = ob->hits.data;
	std::stable_sort( idx.begin(), idx.end(), [&]( int64_t a, int64_t b ){
		const int32_t	*x = d + a * stride, *y = d + b * stride;
		for( int k = 0; k < RMA_HIT_HDR; k++ )
			if( x[ k ] != y[ k ] )
				return x[ k ] < y[ k ];
		return false;
	} );
	ob->sorted.resize( size_t( ob->hits.n ) * stride );
	for( int64_t i = 0; i < ob->hits.n; i++ )
		memcpy( &ob->sorted[ i * stride ], d + idx[ i ] * stride, stride * sizeof( int32_t ) );
	*hits = ob->sorted.data();
	*n_hits = ob->hits.n;
	return 0;
}

// The packed entry points of the driver's pipeline (reader -> upload -> scan -> replay threads, and
// the replay on several threads), so that the CPU tests run the host code the product runs: an
// "upload" remembers which entries, the "scan" unpacks their text and calls the oracle.
struct PackedBatch { const rma::PackFile *pk; int first, count; };

int oracle_upload_packed( void *, const rma::PackFile *pk, int first, int count, void **handle, char *, size_t )
{
	*handle = new PackedBatch{ pk, first, count };
	return 0;
}

int oracle_scan_uploaded( void *self, void *handle, const int32_t **hits, int64_t *n_hits, char *err, size_t errlen )
{
	PackedBatch	*b = ( PackedBatch * )handle;
	std::vector<std::string>	text( size_t( b->count ) );
	std::vector<const char *>	seqs( size_t( b->count ) );
	std::vector<int32_t>	slens( size_t( b->count ) );
	for( int i = 0; i < b->count; i++ ){
		text[ i ] = b->pk->unpack( b->first + i );
		seqs[ i ] = text[ i ].c_str();
		slens[ i ] = int32_t( text[ i ].size() );
	}
	const int	n = b->count;
	delete b;
	return oracle_scan( self, seqs.data(), slens.data(), n, hits, n_hits, err, errlen );
}

void oracle_drop_uploaded( void *, void *handle ) { delete ( PackedBatch * )handle; }

rma::ScanBackend make_oracle( const rma_program_t *prog, const rma_efndata_t *efn, const rma_efn2data_t *efn2 )
{
	rmo_set_efn2data( efn2 );	// tables for efn2() sites: the product's loader (checked against efn2_drv)
	OracleBackend	*ob = new OracleBackend;
	ob->prog = prog;
	ob->efn = efn;
	// the oracle reads the tables with its own loader when told where they are
	const char	*dir = getenv( "RMO_EFNDATA" );
	if( efn != nullptr && dir != nullptr ){
		if( !rmo_load_efndata( dir, &ob->own_efn ) )
			rma::fail( "oracle: can't load efn data from %s", dir );
		ob->efn = &ob->own_efn;
	}
	rmo_hits_init( &ob->hits, prog );
	if( getenv( "RMO_NO_PIPELINE" ) )		// (the one-thread loop over text only)
		return rma::ScanBackend{ ob, oracle_scan };
	return rma::ScanBackend{ ob, oracle_scan, oracle_upload_packed, oracle_scan_uploaded, oracle_drop_uploaded };
}

}	// namespace

int main( int argc, char **argv )
{
	return rma::cli_main( argc, argv, make_oracle );
}

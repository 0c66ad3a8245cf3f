// tests/hostsim/driver_fault_check.cpp -- TEST INFRASTRUCTURE: the command line driver
// (rnamotif_amd/csrc/rm_driver.cpp: reader -> upload -> scan -> replay, a thread each) around a
// fake scanner that finds nothing, keeps reading the pack it was handed while it "works", and can be
// told to fail.  Built with -fsanitize=address,undefined by tests/test_driver_faults.py: whatever
// goes wrong while batches are in flight -- a later file that is not a pack, a scan that fails --
// must end with the error message and exit code 1, with no thread touching a pack that is gone.
//   FAULT=upload:K | scan:K   the K-th upload / scan (from 0) fails
#include "rm_cli.h"
#include "rm_pack.h"
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

struct Handle { const rma::PackFile *pk; int first, count; };
std::atomic<int>	n_up{ 0 }, n_scan{ 0 };
int	fail_up = -1, fail_scan = -1;

// what a real scanner does with the pack during an upload: read every word of the batch
unsigned touch( const rma::PackFile *pk, int first, int count )
{
	unsigned	x = 0;
	for( int i = first; i < first + count; i++ ){
		const uint32_t	*cw = pk->codes.data() + pk->base_off[ i ] / 16;
		for( int w = 0; w < ( pk->slen[ i ] + 15 ) / 16; w++ )
			x ^= cw[ w ];
		x ^= unsigned( pk->sid( i )[ 0 ] );
	}
	return x;
}

int fake_scan( void *, const char *const *, const int32_t *, int, const int32_t **hits, int64_t *n_hits, char *, size_t )
{
	*hits = nullptr;
	*n_hits = 0;
	return 0;
}

int fake_upload( void *, const rma::PackFile *pk, int first, int count, void **handle, char *err, size_t errlen )
{
	*handle = nullptr;
	std::this_thread::sleep_for( std::chrono::milliseconds( 3 ) );
	volatile unsigned	sink = touch( pk, first, count );
	( void )sink;
	if( n_up++ == fail_up ){
		snprintf( err, errlen, "injected upload failure" );
		return 1;
	}
	*handle = new Handle{ pk, first, count };
	return 0;
}

int fake_scan_uploaded( void *, void *handle, const int32_t **hits, int64_t *n_hits, char *err, size_t errlen )
{
	Handle	*h = ( Handle * )handle;
	*hits = nullptr;
	*n_hits = 0;
	std::this_thread::sleep_for( std::chrono::milliseconds( 5 ) );
	volatile unsigned	sink = touch( h->pk, h->first, h->count );
	( void )sink;
	delete h;
	if( n_scan++ == fail_scan ){
		snprintf( err, errlen, "injected scan failure" );
		return 1;
	}
	return 0;
}

void fake_drop( void *, void *handle ) { delete ( Handle * )handle; }

rma::ScanBackend make_fake( const rma_program_t *, const rma_efndata_t *, const rma_efn2data_t * )
{
	return rma::ScanBackend{ nullptr, fake_scan, fake_upload, fake_scan_uploaded, fake_drop };
}

}	// namespace

int main( int argc, char **argv )
{
	if( const char *f = getenv( "FAULT" ) ){
		if( !strncmp( f, "upload:", 7 ) ) fail_up = atoi( f + 7 );
		if( !strncmp( f, "scan:", 5 ) ) fail_scan = atoi( f + 5 );
	}
	const int	rc = rma::cli_main( argc, argv, make_fake );
	fprintf( stderr, "uploads %d, scans %d\n", n_up.load(), n_scan.load() );
	return rc;
}

// rm_main.cpp -- the rnamotif executable: the reference's command line and
// output (/root/reference/src/rnamot.c) with the scan on the GPU.  The scanner
// is reached through the C ABI only; there is no CPU search path in this binary.
#include "rm_cli.h"
#include "rnamotif_amd.h"
#include "rm_pack.h"

// rm_capi.cpp: rma_pack is a struct whose only member is the PackFile
const rma_pack_t *rma_pack_wrap( const rma::PackFile *pf );
#include <cstdlib>
#include <unistd.h>

namespace {

struct HipBackend {
	rma_scanner_t	*sc = nullptr;
	rma_db_t	*db = nullptr;
};

int hip_scan( void *self, const char *const *seqs, const int32_t *slens, int n,
	const int32_t **hits, int64_t *n_hits, char *err, size_t errlen )
{
	HipBackend	*hb = ( HipBackend * )self;
	if( hb->db != nullptr ){
		rma_db_destroy( hb->db );
		hb->db = nullptr;
	}
	if( rma_db_create( hb->sc, seqs, slens, n, &hb->db, err, errlen ) )
		return 1;
	return rma_scan( hb->sc, hb->db, hits, n_hits, err, errlen );
}

// the driver holds a PackFile; the ABI wants the handle that wraps one
int hip_upload_packed( void *self, const rma::PackFile *pk, int first, int count, void **handle, char *err, size_t errlen )
{
	HipBackend	*hb = ( HipBackend * )self;
	rma_db_t	*db = nullptr;
	*handle = nullptr;
	// (without waiting: the copies run on the upload stream, the scan waits for them on the device)
	if( rma_db_create_packed_async( hb->sc, rma_pack_wrap( pk ), first, count, &db, err, errlen ) )
		return 1;
	*handle = db;
	return 0;
}

int hip_scan_uploaded( void *self, void *handle, const int32_t **hits, int64_t *n_hits, char *err, size_t errlen )
{
	HipBackend	*hb = ( HipBackend * )self;
	rma_db_t	*db = ( rma_db_t * )handle;
	const int	rc = rma_scan( hb->sc, db, hits, n_hits, err, errlen );
	rma_db_destroy( db );		// (its block of HBM waits for the next batch)
	return rc;
}

void hip_drop_uploaded( void *, void *handle )
{
	rma_db_destroy( ( rma_db_t * )handle );
}

rma::ScanBackend make_hip( const rma_program_t *prog, const rma_efndata_t *efn, const rma_efn2data_t *efn2 )
{
	HipBackend	*hb = new HipBackend;
	char	err[ 1024 ] = "";
	const char	*dv = getenv( "RNAMOTIF_DEVICE" );
	if( rma_scanner_create( prog, efn, dv ? atoi( dv ) : 0, &hb->sc, err, sizeof( err ) ) )
		rma::fail( "%s", err );
	if( efn2 != nullptr && rma_scanner_set_efn2data( hb->sc, efn2, err, sizeof( err ) ) )
		rma::fail( "%s", err );
	// first-use set-up of the runtime now, not in the first batch (a failure here would show there too)
	if( !getenv( "RNAMOTIF_NO_WARMUP" ) )
		( void )rma_scanner_warmup( hb->sc, err, sizeof( err ) );
	return rma::ScanBackend{ hb, hip_scan, hip_upload_packed, hip_scan_uploaded, hip_drop_uploaded };
}

}	// namespace

int main( int argc, char **argv )
{
	const int	rc = rma::cli_main( argc, argv, make_hip );
	// everything is printed: leave without tearing the HIP runtime down piece by piece (the
	// device allocations go with the process; 50-100 ms of a run that searches 200 Mbase in 60)
	fflush( stdout );
	fflush( stderr );
	_exit( rc );
}

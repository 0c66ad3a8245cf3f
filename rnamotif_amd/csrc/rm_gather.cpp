// rm_gather.cpp -- the one exchange step of a multi-GPU search behind the C ABI: the candidate
// records every rank's scan left in HBM travel to one rank over RCCL, device to device, and reach
// the host there in a single copy.  The role of the MT_RESULT messages of the reference's farm,
// /root/reference/src/mrnamotif.c:733-760 (master) and :898-917 (worker).
//
//   ncclAllGather   8 bytes per rank: how many records each rank holds
//   ncclSend/Recv   one grouped exchange: every rank with records sends them to the root, which
//                   receives each part at its offset -- no padding to the largest part, nothing
//                   from ranks that found nothing
//
// xGMI is point to point: the root's seven links each carry one peer's part at the same time.
// RCCL is taken from the process at run time (dlopen): a process that has loaded it already --
// torch.distributed -- shares that copy; the library loads without it and only these calls fail.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "rm_hitsort_dev.h"
#include "rnamotif_amd.h"

// rm_scanner.cpp
int	rma_scanner_device( const rma_scanner_t *sc );
hipStream_t	rma_scanner_stream( const rma_scanner_t *sc );
int	rma_scanner_stride( const rma_scanner_t *sc );
void	rma_scanner_last( const rma_scanner_t *sc, const int32_t **d_hits, int64_t *n );

namespace {

struct Rccl {
	void	*so = nullptr;
	ncclResult_t	( *GetUniqueId )( ncclUniqueId * ) = nullptr;
	ncclResult_t	( *CommInitRank )( ncclComm_t *, int, ncclUniqueId, int ) = nullptr;
	ncclResult_t	( *CommDestroy )( ncclComm_t ) = nullptr;
	ncclResult_t	( *AllGather )( const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t ) = nullptr;
	ncclResult_t	( *Send )( const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t ) = nullptr;
	ncclResult_t	( *Recv )( void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t ) = nullptr;
	ncclResult_t	( *GroupStart )() = nullptr;
	ncclResult_t	( *GroupEnd )() = nullptr;
	const char	*( *GetErrorString )( ncclResult_t ) = nullptr;
	std::string	why;
};

std::string	g_why;		// why rccl() failed

Rccl *rccl()
{
	static Rccl	r;
	static std::once_flag	once;
	std::call_once( once, [](){
		const char	*names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
		for( const char *n : names )
			if( ( r.so = dlopen( n, RTLD_NOW | RTLD_GLOBAL ) ) != nullptr )
				break;
		if( r.so == nullptr ){
			r.why = std::string( "RCCL is not available: " ) + dlerror();
			return;
		}
		bool	ok = true;
		auto sym = [&]( const char *name ) -> void * {
			void	*p = dlsym( r.so, name );
			if( p == nullptr ){
				ok = false;
				r.why = std::string( "RCCL lacks " ) + name;
			}
			return p;
		};
		r.GetUniqueId = reinterpret_cast<decltype( r.GetUniqueId )>( sym( "ncclGetUniqueId" ) );
		r.CommInitRank = reinterpret_cast<decltype( r.CommInitRank )>( sym( "ncclCommInitRank" ) );
		r.CommDestroy = reinterpret_cast<decltype( r.CommDestroy )>( sym( "ncclCommDestroy" ) );
		r.AllGather = reinterpret_cast<decltype( r.AllGather )>( sym( "ncclAllGather" ) );
		r.Send = reinterpret_cast<decltype( r.Send )>( sym( "ncclSend" ) );
		r.Recv = reinterpret_cast<decltype( r.Recv )>( sym( "ncclRecv" ) );
		r.GroupStart = reinterpret_cast<decltype( r.GroupStart )>( sym( "ncclGroupStart" ) );
		r.GroupEnd = reinterpret_cast<decltype( r.GroupEnd )>( sym( "ncclGroupEnd" ) );
		r.GetErrorString = reinterpret_cast<decltype( r.GetErrorString )>( sym( "ncclGetErrorString" ) );
		if( !ok ){
			dlclose( r.so );
			r.so = nullptr;
		}
	} );
	if( r.so == nullptr )
		g_why = r.why;
	return r.so != nullptr ? &r : nullptr;
}

int no_rccl( char *err, size_t errlen )
{
	snprintf( err, errlen, "%s", g_why.empty() ? "RCCL could not be loaded (librccl.so.1)" : g_why.c_str() );
	return 1;
}

}	// namespace

struct rma_comm {
	ncclComm_t	comm = nullptr;
	int	rank = 0, world = 1, device = 0;
	long long	*d_counts = nullptr;	// [world + 1]: the ranks' counts, then this rank's own
	long long	*h_counts = nullptr;	// pinned, [world]
	int32_t	*d_index = nullptr;
	size_t	index_cap = 0;
	int32_t	*d_all = nullptr;		// root: every rank's records, rank by rank
	size_t	all_cap = 0;			// words
	int32_t	*h_all = nullptr;		// pinned
	size_t	h_cap = 0;
};

#define HIPCHK( call )	do{ hipError_t e_ = ( call ); if( e_ != hipSuccess ){ \
		snprintf( err, errlen, "%s: %s", #call, hipGetErrorString( e_ ) ); return 1; } }while( 0 )
#define NCCLCHK( call )	do{ ncclResult_t r_ = ( call ); if( r_ != ncclSuccess ){ \
		snprintf( err, errlen, "%s: %s", #call, R->GetErrorString( r_ ) ); return 1; } }while( 0 )

extern "C" int rma_comm_unique_id( uint8_t id[ RMA_COMM_ID_BYTES ], char *err, size_t errlen )
{
	static_assert( RMA_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's" );
	Rccl	*R = rccl();
	if( R == nullptr )
		return no_rccl( err, errlen );
	ncclUniqueId	u;
	NCCLCHK( R->GetUniqueId( &u ) );
	memcpy( id, u.internal, NCCL_UNIQUE_ID_BYTES );
	return 0;
}

extern "C" void rma_comm_destroy( rma_comm_t *c )
{
	if( c == nullptr )
		return;
	( void )hipSetDevice( c->device );
	( void )hipFree( c->d_counts );
	( void )hipFree( c->d_index );
	( void )hipFree( c->d_all );
	if( c->h_counts )
		( void )hipHostFree( c->h_counts );
	if( c->h_all )
		( void )hipHostFree( c->h_all );
	if( c->comm != nullptr )
		if( Rccl *R = rccl() )
			( void )R->CommDestroy( c->comm );
	delete c;
}

extern "C" int rma_comm_create( const uint8_t id[ RMA_COMM_ID_BYTES ], int rank, int world, int device,
	rma_comm_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	if( world < 1 || rank < 0 || rank >= world ){
		snprintf( err, errlen, "rma_comm_create: rank %d of %d", rank, world );
		return 1;
	}
	HIPCHK( hipSetDevice( device ) );
	rma_comm	*c = new rma_comm;
	struct Guard { rma_comm *p; ~Guard(){ if( p ) rma_comm_destroy( p ); } }	guard{ c };
	c->rank = rank;
	c->world = world;
	c->device = device;
	HIPCHK( hipMalloc( &c->d_counts, size_t( world + 1 ) * sizeof( long long ) ) );
	HIPCHK( hipHostMalloc( reinterpret_cast<void **>( &c->h_counts ), size_t( world ) * sizeof( long long ), hipHostMallocDefault ) );
	if( world > 1 ){
		Rccl	*R = rccl();
		if( R == nullptr )
			return no_rccl( err, errlen );
		ncclUniqueId	u;
		memcpy( u.internal, id, NCCL_UNIQUE_ID_BYTES );
		NCCLCHK( R->CommInitRank( &c->comm, world, u, rank ) );
	}
	guard.p = nullptr;
	*out = c;
	return 0;
}

extern "C" int rma_gather_hits( rma_comm_t *c, rma_scanner_t *sc, const int32_t *global_index, int32_t n_index, int root,
	const int32_t **hits, int64_t *n_hits, int64_t *counts, char *err, size_t errlen )
{
	*hits = nullptr;
	*n_hits = 0;
	if( rma_scanner_device( sc ) != c->device ){
		snprintf( err, errlen, "rma_gather_hits: the scanner is on device %d, the communicator on device %d", rma_scanner_device( sc ), c->device );
		return 1;
	}
	if( root < 0 || root >= c->world ){
		snprintf( err, errlen, "rma_gather_hits: root %d of %d ranks", root, c->world );
		return 1;
	}
	Rccl	*R = c->world > 1 ? rccl() : nullptr;
	if( c->world > 1 && R == nullptr )
		return no_rccl( err, errlen );
	HIPCHK( hipSetDevice( c->device ) );
	hipStream_t	s = rma_scanner_stream( sc );
	const int	stride = rma_scanner_stride( sc );
	const int32_t	*d_mine = nullptr;
	int64_t	n_mine = 0;
	rma_scanner_last( sc, &d_mine, &n_mine );
	// word 0 of every record: the entry's number in the whole database
	if( n_mine > 0 ){
		if( global_index == nullptr || n_index <= 0 ){
			snprintf( err, errlen, "rma_gather_hits: %lld records and no entry numbers", ( long long )n_mine );
			return 1;
		}
		if( size_t( n_index ) > c->index_cap ){
			HIPCHK( hipStreamSynchronize( s ) );
			( void )hipFree( c->d_index );
			c->d_index = nullptr;
			c->index_cap = 0;
			HIPCHK( hipMalloc( &c->d_index, size_t( n_index ) * 2 * sizeof( int32_t ) ) );
			c->index_cap = size_t( n_index ) * 2;
		}
		HIPCHK( hipMemcpyAsync( c->d_index, global_index, size_t( n_index ) * sizeof( int32_t ), hipMemcpyHostToDevice, s ) );
		HIPCHK( rma::relabel_entries( const_cast<int32_t *>( d_mine ), n_mine, stride, c->d_index, n_index, s ) );
	}
	// how many records every rank holds
	long long	mine = n_mine;
	if( c->world > 1 ){
		HIPCHK( hipMemcpyAsync( c->d_counts + c->world, &mine, sizeof( mine ), hipMemcpyHostToDevice, s ) );
		NCCLCHK( R->AllGather( c->d_counts + c->world, c->d_counts, 1, ncclInt64, c->comm, s ) );
		HIPCHK( hipMemcpyAsync( c->h_counts, c->d_counts, size_t( c->world ) * sizeof( long long ), hipMemcpyDeviceToHost, s ) );
		HIPCHK( hipStreamSynchronize( s ) );		// (the one wait before the exchange: its sizes)
	}else
		c->h_counts[ 0 ] = mine;
	int64_t	total = 0;
	for( int r = 0; r < c->world; r++ ){
		if( counts )
			counts[ r ] = c->h_counts[ r ];
		total += c->h_counts[ r ];
	}
	if( total == 0 )
		return 0;
	if( c->rank == root ){
		const size_t	words = size_t( total ) * stride;
		if( words > c->all_cap ){
			( void )hipFree( c->d_all );
			c->d_all = nullptr;
			c->all_cap = 0;
			HIPCHK( hipMalloc( &c->d_all, ( words + words / 2 ) * sizeof( int32_t ) ) );
			c->all_cap = words + words / 2;
		}
		if( words > c->h_cap ){
			if( c->h_all )
				( void )hipHostFree( c->h_all );
			c->h_all = nullptr;
			c->h_cap = 0;
			HIPCHK( hipHostMalloc( reinterpret_cast<void **>( &c->h_all ), ( words + words / 2 ) * sizeof( int32_t ), hipHostMallocDefault ) );
			c->h_cap = words + words / 2;
		}
	}
	// the exchange: one group, every part to its place in the root's buffer
	if( c->world > 1 )
		NCCLCHK( R->GroupStart() );
	if( c->rank == root ){
		size_t	at = 0;
		for( int r = 0; r < c->world; r++ ){
			const size_t	w = size_t( c->h_counts[ r ] ) * stride;
			if( w == 0 )
				continue;
			if( r == root )
				HIPCHK( hipMemcpyAsync( c->d_all + at, d_mine, w * sizeof( int32_t ), hipMemcpyDeviceToDevice, s ) );
			else
				NCCLCHK( R->Recv( c->d_all + at, w, ncclInt32, r, c->comm, s ) );
			at += w;
		}
	}else if( n_mine > 0 )
		NCCLCHK( R->Send( d_mine, size_t( n_mine ) * stride, ncclInt32, root, c->comm, s ) );
	if( c->world > 1 )
		NCCLCHK( R->GroupEnd() );
	if( c->rank == root ){
		HIPCHK( hipMemcpyAsync( c->h_all, c->d_all, size_t( total ) * stride * sizeof( int32_t ), hipMemcpyDeviceToHost, s ) );
		HIPCHK( hipStreamSynchronize( s ) );
		*hits = c->h_all;
		*n_hits = total;
	}else
		HIPCHK( hipStreamSynchronize( s ) );	// (the records have left before the scanner's next scan overwrites them)
	return 0;
}

// rm_gather.cpp -- the one exchange step of a multi-GPU search behind the C ABI: the candidate
// records every rank's scan left in HBM travel to one rank over RCCL, device to device, and reach
// the host there in a single copy.  The role of the MT_RESULT messages of the reference's farm,
// /root/reference/src/mrnamotif.c:733-760 (master) and :898-917 (worker).
//
//   ncclAllGather   16 bytes per rank: how many records each rank holds, and whether it is fit to take part (a rank
//                   that is not says so here: nobody enters the exchange and waits for it)
//   (ncclAllGather) only when the root's buffers must grow -- every rank can tell from the totals: could it allocate?
//   ncclSend/Recv   one grouped exchange: every rank with records sends them to the root, which
//                   receives each part at its offset -- no padding to the largest part, nothing
//                   from ranks that found nothing; the group is closed on every path, failed calls included
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
enum { RMA_LAST_NONE = 0, RMA_LAST_ON_DEVICE, RMA_LAST_ON_HOST };
int	rma_scanner_last_state( const rma_scanner_t *sc );	// where the last scan's ordered records are
bool	rma_scanner_last_relabelled( const rma_scanner_t *sc );
void	rma_scanner_set_relabelled( rma_scanner_t *sc );

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
	ncclResult_t	( *CommCount )( const ncclComm_t, int * ) = nullptr;
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
		r.CommCount = reinterpret_cast<decltype( r.CommCount )>( sym( "ncclCommCount" ) );
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

// The collective calls of the exchange as a table (rma_transport_t of the C ABI): RCCL's by default, a caller's
// through rma_comm_create_on() -- the tests drive the exchange's error paths through a transport of their own.
namespace {

int rccl_all_gather( void *ctx, const void *send, void *recv, size_t n, void *stream )
{
	return int( rccl()->AllGather( send, recv, n, ncclInt64, static_cast<ncclComm_t>( ctx ), static_cast<hipStream_t>( stream ) ) );
}
int rccl_send( void *ctx, const void *buf, size_t n, int peer, void *stream )
{
	return int( rccl()->Send( buf, n, ncclInt32, peer, static_cast<ncclComm_t>( ctx ), static_cast<hipStream_t>( stream ) ) );
}
int rccl_recv( void *ctx, void *buf, size_t n, int peer, void *stream )
{
	return int( rccl()->Recv( buf, n, ncclInt32, peer, static_cast<ncclComm_t>( ctx ), static_cast<hipStream_t>( stream ) ) );
}
int rccl_group_start( void * ) { return int( rccl()->GroupStart() ); }
int rccl_group_end( void * ) { return int( rccl()->GroupEnd() ); }
const char *rccl_error_string( void *, int code ) { return rccl()->GetErrorString( ncclResult_t( code ) ); }
int rccl_comm_count( void *ctx, int *count ) { return int( rccl()->CommCount( static_cast<ncclComm_t>( ctx ), count ) ); }

}	// namespace

struct rma_comm {
	ncclComm_t	comm = nullptr;		// RCCL's communicator when the transport is RCCL's (destroyed with this)
	rma_transport_t	tr{};
	int	rank = 0, world = 1, device = 0;
	long long	*d_counts = nullptr;	// [2 * world + 2]: the ranks' counts, the ranks' flags, this rank's own count and flag
	long long	*h_counts = nullptr;	// pinned, [2 * world]
	int32_t	*d_index = nullptr;
	size_t	index_cap = 0;
	int32_t	*d_all = nullptr;		// root: every rank's records, rank by rank
	size_t	all_cap = 0;			// words
	int32_t	*h_all = nullptr;		// pinned
	size_t	h_cap = 0;
	// what every rank knows the root's buffers to hold at least (words): the largest total any gather has had,
	// times one and a half -- when a gather exceeds it, the root must allocate, and every rank waits for its word
	size_t	agreed_cap = 0;
};

#define HIPCHK( call )	do{ hipError_t e_ = ( call ); if( e_ != hipSuccess ){ \
		snprintf( err, errlen, "%s: %s", #call, hipGetErrorString( e_ ) ); return 1; } }while( 0 )
#define NCCLCHK( call )	do{ ncclResult_t r_ = ( call ); if( r_ != ncclSuccess ){ \
		snprintf( err, errlen, "%s: %s", #call, R->GetErrorString( r_ ) ); return 1; } }while( 0 )
// a call of the communicator's transport
#define TRCHK( call )	do{ const int r_ = ( call ); if( r_ != 0 ){ \
		snprintf( err, errlen, "%s: %s", #call, c->tr.error_string ? c->tr.error_string( c->tr.ctx, r_ ) : "transport error" ); return 1; } }while( 0 )

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

static int comm_new( int rank, int world, int device, rma_comm **out, char *err, size_t errlen )
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
	HIPCHK( hipMalloc( &c->d_counts, size_t( 2 * world + 2 ) * sizeof( long long ) ) );
	HIPCHK( hipHostMalloc( reinterpret_cast<void **>( &c->h_counts ), size_t( 2 * world ) * sizeof( long long ), hipHostMallocDefault ) );
	guard.p = nullptr;
	*out = c;
	return 0;
}

extern "C" int rma_comm_create( const uint8_t id[ RMA_COMM_ID_BYTES ], int rank, int world, int device,
	rma_comm_t **out, char *err, size_t errlen )
{
	rma_comm	*c = nullptr;
	*out = nullptr;
	if( comm_new( rank, world, device, &c, err, errlen ) )
		return 1;
	struct Guard { rma_comm *p; ~Guard(){ if( p ) rma_comm_destroy( p ); } }	guard{ c };
	if( world > 1 ){
		Rccl	*R = rccl();
		if( R == nullptr )
			return no_rccl( err, errlen );
		ncclUniqueId	u;
		memcpy( u.internal, id, NCCL_UNIQUE_ID_BYTES );
		NCCLCHK( R->CommInitRank( &c->comm, world, u, rank ) );
		c->tr = rma_transport_t{ rccl_all_gather, rccl_send, rccl_recv, rccl_group_start, rccl_group_end, rccl_error_string, rccl_comm_count, c->comm };
	}
	guard.p = nullptr;
	*out = c;
	return 0;
}

extern "C" int rma_comm_create_on( const rma_transport_t *t, int rank, int world, int device,
	rma_comm_t **out, char *err, size_t errlen )
{
	*out = nullptr;
	if( t == nullptr || t->all_gather == nullptr || t->send == nullptr || t->recv == nullptr || t->group_start == nullptr || t->group_end == nullptr ){
		snprintf( err, errlen, "rma_comm_create_on: the transport lacks a call" );
		return 1;
	}
	rma_comm	*c = nullptr;
	if( comm_new( rank, world, device, &c, err, errlen ) )
		return 1;
	c->tr = *t;
	*out = c;
	return 0;
}

// How many ranks the transport itself says the job has (RCCL: ncclCommCount) -- what a scaling figure is checked against.
extern "C" int rma_comm_count( rma_comm_t *c, int *count, char *err, size_t errlen )
{
	*count = c->world;
	if( c->world > 1 && c->tr.comm_count != nullptr )
		TRCHK( c->tr.comm_count( c->tr.ctx, count ) );
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
	HIPCHK( hipSetDevice( c->device ) );
	hipStream_t	s = rma_scanner_stream( sc );
	const int	stride = rma_scanner_stride( sc );
	const int32_t	*d_mine = nullptr;
	int64_t	n_mine = 0;
	rma_scanner_last( sc, &d_mine, &n_mine );
	// What this rank can say before anything is exchanged.  A rank that cannot take part still goes through the
	// count exchange -- with a flag -- so that no peer is left waiting in a collective it never joins.
	std::string	trouble;
	if( rma_scanner_last_state( sc ) == RMA_LAST_ON_HOST )
		trouble = "rma_gather_hits: the last scan's records were ordered on the host and are not in HBM (end the scan with rma_scan_end_on_device())";
	else if( n_mine > 0 && ( global_index == nullptr || n_index <= 0 ) )
		trouble = "rma_gather_hits: " + std::to_string( ( long long )n_mine ) + " records and no entry numbers";
	// word 0 of every record: the entry's number in the whole database -- once per scan (a second gather of the same
	// records finds them relabelled)
	if( trouble.empty() && n_mine > 0 && !rma_scanner_last_relabelled( sc ) ){
		hipError_t	e = hipSuccess;
		if( size_t( n_index ) > c->index_cap ){
			e = hipStreamSynchronize( s );
			( void )hipFree( c->d_index );
			c->d_index = nullptr;
			c->index_cap = 0;
			if( e == hipSuccess )
				e = hipMalloc( &c->d_index, size_t( n_index ) * 2 * sizeof( int32_t ) );
			if( e == hipSuccess )
				c->index_cap = size_t( n_index ) * 2;
		}
		if( e == hipSuccess )
			e = hipMemcpyAsync( c->d_index, global_index, size_t( n_index ) * sizeof( int32_t ), hipMemcpyHostToDevice, s );
		if( e == hipSuccess )
			e = rma::relabel_entries( const_cast<int32_t *>( d_mine ), n_mine, stride, c->d_index, n_index, s );
		if( e == hipSuccess )
			rma_scanner_set_relabelled( sc );
		else
			trouble = std::string( "rma_gather_hits: entry numbers: " ) + hipGetErrorString( e );
	}
	// how many records every rank holds, and whether every rank is fit to go on: one all-gather of two words
	long long	mine[ 2 ] = { trouble.empty() ? ( long long )n_mine : 0, trouble.empty() ? 0 : 1 };
	if( c->world > 1 ){
		HIPCHK( hipMemcpyAsync( c->d_counts + 2 * c->world, mine, sizeof( mine ), hipMemcpyHostToDevice, s ) );
		TRCHK( c->tr.all_gather( c->tr.ctx, c->d_counts + 2 * c->world, c->d_counts, 2, s ) );
		HIPCHK( hipMemcpyAsync( c->h_counts, c->d_counts, size_t( 2 * c->world ) * sizeof( long long ), hipMemcpyDeviceToHost, s ) );
		HIPCHK( hipStreamSynchronize( s ) );		// (the one wait before the exchange: its sizes)
	}else{
		c->h_counts[ 0 ] = mine[ 0 ];
		c->h_counts[ 1 ] = mine[ 1 ];
	}
	int64_t	total = 0;
	int	unfit = -1;
	std::vector<long long>	cnt( size_t( c->world ), 0 );		// (h_counts is used again below)
	for( int r = 0; r < c->world; r++ ){
		cnt[ size_t( r ) ] = c->h_counts[ 2 * r ];
		if( counts )
			counts[ r ] = cnt[ size_t( r ) ];
		total += cnt[ size_t( r ) ];
		if( c->h_counts[ 2 * r + 1 ] != 0 && unfit < 0 )
			unfit = r;
	}
	if( unfit >= 0 ){
		// every rank leaves here, none enters the exchange
		if( trouble.empty() )
			trouble = "rma_gather_hits: rank " + std::to_string( unfit ) + " cannot take part";
		snprintf( err, errlen, "%s", trouble.c_str() );
		return 1;
	}
	if( total == 0 )
		return 0;
	// The root's buffers.  Every rank knows when the root has to allocate (agreed_cap follows the totals, which all
	// ranks see): then -- and only then -- a second small all-gather says whether it could, before anybody sends.
	const size_t	words = size_t( total ) * stride;
	const bool	grows = words > c->agreed_cap;
	std::string	alloc_trouble;
	if( c->rank == root && ( words > c->all_cap || words > c->h_cap ) ){
		hipError_t	e = hipSuccess;
		if( words > c->all_cap ){
			( void )hipFree( c->d_all );
			c->d_all = nullptr;
			c->all_cap = 0;
			e = hipMalloc( &c->d_all, ( words + words / 2 ) * sizeof( int32_t ) );
			if( e == hipSuccess )
				c->all_cap = words + words / 2;
		}
		if( e == hipSuccess && words > c->h_cap ){
			if( c->h_all )
				( void )hipHostFree( c->h_all );
			c->h_all = nullptr;
			c->h_cap = 0;
			e = hipHostMalloc( reinterpret_cast<void **>( &c->h_all ), ( words + words / 2 ) * sizeof( int32_t ), hipHostMallocDefault );
			if( e == hipSuccess )
				c->h_cap = words + words / 2;
		}
		if( e != hipSuccess ){
			( void )hipGetLastError();
			alloc_trouble = std::string( "rma_gather_hits: the root's buffers for " ) + std::to_string( ( long long )total ) + " records: " + hipGetErrorString( e );
		}
	}
	if( grows ){
		c->agreed_cap = words + words / 2;
		if( c->world > 1 ){
			long long	flag[ 2 ] = { alloc_trouble.empty() ? 0 : 1, 0 };
			HIPCHK( hipMemcpyAsync( c->d_counts + 2 * c->world, flag, sizeof( flag ), hipMemcpyHostToDevice, s ) );
			TRCHK( c->tr.all_gather( c->tr.ctx, c->d_counts + 2 * c->world, c->d_counts, 2, s ) );
			HIPCHK( hipMemcpyAsync( c->h_counts, c->d_counts, size_t( 2 * c->world ) * sizeof( long long ), hipMemcpyDeviceToHost, s ) );
			HIPCHK( hipStreamSynchronize( s ) );
			if( c->h_counts[ 2 * root ] != 0 && alloc_trouble.empty() )
				alloc_trouble = "rma_gather_hits: rank " + std::to_string( root ) + " (the root) has no room for " + std::to_string( ( long long )total ) + " records";
		}
	}
	if( !alloc_trouble.empty() ){
		c->agreed_cap = 0;		// (the next gather asks again)
		snprintf( err, errlen, "%s", alloc_trouble.c_str() );
		return 1;
	}
	// the exchange: one group, every part to its place in the root's buffer.  Whatever fails between the group's
	// start and its end, the end is reached: an open group would hold every later call of this thread.
	std::string	xch;
	auto note = [ & ]( const char *what, int code ){
		if( xch.empty() )
			xch = std::string( what ) + ": " + ( c->tr.error_string ? c->tr.error_string( c->tr.ctx, code ) : "transport error" );
	};
	auto note_hip = [ & ]( const char *what, hipError_t e ){
		if( xch.empty() )
			xch = std::string( what ) + ": " + hipGetErrorString( e );
	};
	bool	open = false;
	if( c->world > 1 ){
		const int	r_ = c->tr.group_start( c->tr.ctx );
		if( r_ != 0 )
			note( "group start", r_ );
		else
			open = true;
	}
	if( xch.empty() ){
		if( c->rank == root ){
			size_t	at = 0;
			for( int r = 0; r < c->world && xch.empty(); r++ ){
				const size_t	w = size_t( cnt[ size_t( r ) ] ) * stride;
				if( w == 0 )
					continue;
				if( r == root ){
					const hipError_t	e = hipMemcpyAsync( c->d_all + at, d_mine, w * sizeof( int32_t ), hipMemcpyDeviceToDevice, s );
					if( e != hipSuccess )
						note_hip( "the root's own records", e );
				}else{
					const int	r_ = c->tr.recv( c->tr.ctx, c->d_all + at, w, r, s );
					if( r_ != 0 )
						note( "receive", r_ );
				}
				at += w;
			}
		}else if( n_mine > 0 ){
			const int	r_ = c->tr.send( c->tr.ctx, d_mine, size_t( n_mine ) * stride, root, s );
			if( r_ != 0 )
				note( "send", r_ );
		}
	}
	if( open ){
		const int	r_ = c->tr.group_end( c->tr.ctx );
		if( r_ != 0 )
			note( "group end", r_ );
	}
	if( !xch.empty() ){
		( void )hipStreamSynchronize( s );
		snprintf( err, errlen, "rma_gather_hits: %s", xch.c_str() );
		return 1;
	}
	if( c->rank == root ){
		HIPCHK( hipMemcpyAsync( c->h_all, c->d_all, size_t( total ) * stride * sizeof( int32_t ), hipMemcpyDeviceToHost, s ) );
		HIPCHK( hipStreamSynchronize( s ) );
		*hits = c->h_all;
		*n_hits = total;
	}else
		HIPCHK( hipStreamSynchronize( s ) );	// (the records have left before the scanner's next scan overwrites them)
	return 0;
}

// h2d_probe.cpp -- what the host side of an upload costs on this box: page-locking, and copies from
// pageable and from page-locked memory (decides how the command line's batches are staged).
//   g++ -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include profiles/h2d_probe.cpp -L/opt/rocm/lib -lamdhip64 -o /tmp/h2d_probe
#include <hip/hip_runtime_api.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now(){ return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); }
int main()
{
	const size_t	MB = 1 << 20, N = 36 * MB;
	void	*d = nullptr;
	hipMalloc( &d, N );
	hipStream_t	s;
	hipStreamCreateWithFlags( &s, hipStreamNonBlocking );
	for( int rep = 0; rep < 2; rep++ ){
		double	t0 = now();
		void	*h = nullptr;
		hipHostMalloc( &h, N, hipHostMallocDefault );
		double	t1 = now();
		memset( h, 1, N );
		double	t2 = now();
		hipMemcpyAsync( d, h, N, hipMemcpyHostToDevice, s );
		hipStreamSynchronize( s );
		double	t3 = now();
		hipMemcpyAsync( d, h, N, hipMemcpyHostToDevice, s );
		hipStreamSynchronize( s );
		double	t4 = now();
		hipHostFree( h );
		double	t5 = now();
		printf( "hipHostMalloc 36 MB %.2f ms, first touch %.2f ms, H2D pinned %.2f ms (%.1f GB/s), again %.2f ms (%.1f GB/s), free %.2f ms\n",
			t1 - t0, t2 - t1, t3 - t2, N / ( t3 - t2 ) / 1e6, t4 - t3, N / ( t4 - t3 ) / 1e6, t5 - t4 );
		char	*p = ( char * )malloc( N );
		t0 = now();
		memset( p, 2, N );
		t1 = now();
		hipMemcpy( d, p, N, hipMemcpyHostToDevice );
		t2 = now();
		hipMemcpy( d, p, N, hipMemcpyHostToDevice );
		t3 = now();
		hipHostRegister( p, N, hipHostRegisterDefault );
		t4 = now();
		hipMemcpyAsync( d, p, N, hipMemcpyHostToDevice, s );
		hipStreamSynchronize( s );
		t5 = now();
		hipHostUnregister( p );
		double	t6 = now();
		printf( "malloc first touch %.2f ms, H2D pageable %.2f ms (%.1f GB/s), again %.2f ms (%.1f GB/s), hipHostRegister %.2f ms, H2D registered %.2f ms (%.1f GB/s), unregister %.2f ms\n",
			t1 - t0, t2 - t1, N / ( t2 - t1 ) / 1e6, t3 - t2, N / ( t3 - t2 ) / 1e6, t4 - t3, t5 - t4, N / ( t5 - t4 ) / 1e6, t6 - t5 );
		free( p );
	}
	// memcpy into a page-locked buffer that is reused (a staging ring)
	void	*h = nullptr;
	hipHostMalloc( &h, N, hipHostMallocDefault );
	char	*src = ( char * )malloc( N );
	memset( src, 3, N );
	memset( h, 0, N );
	double	t0 = now();
	memcpy( h, src, N );
	double	t1 = now();
	printf( "memcpy 36 MB pageable -> pinned, one thread: %.2f ms (%.1f GB/s)\n", t1 - t0, N / ( t1 - t0 ) / 1e6 );
	return 0;
}

// profiles/valu_peak.hip -- calibration of the secondary roofline's peak: how many 64-wide
// integer VALU instructions per second MI355X issues with 4 waves per SIMD (the occupancy of
// the search kernel).  MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a
// SIMD-32 once two or more waves share the SIMD => 256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1228.8 G
// wave-instructions/s.  This measures it with the instruction mix of the search kernel
// (v_add_u32, v_and_b32, v_xor_b32, v_lshlrev_b32, v_alignbit_b32 on independent registers).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_peak profiles/valu_peak.hip && ./valu_peak
// prints one JSON line (committed as profiles/r02_valu_peak.json).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK( c )	do{ hipError_t e_ = ( c ); if( e_ != hipSuccess ){ fprintf( stderr, "%s: %s\n", #c, hipGetErrorString( e_ ) ); return 1; } }while( 0 )

constexpr int	PER_ITER = 32;		// VALU instructions per loop iteration (below)

template< int WAVES_PER_SIMD >
__global__ void __launch_bounds__( 256, WAVES_PER_SIMD ) valu_loop( unsigned *out, int iters, unsigned seed )
{
	unsigned	a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x9e3779b9u, d = a + 77u;
	unsigned	e = b ^ 0x5bd1e995u, f = c + 13u, g = d ^ 0x27d4eb2fu, h = e + 5u;
	for( int i = 0; i < iters; i++ ){
		// 8 independent chains x 4 instructions: no instruction waits for the one before it
		asm volatile(
			"v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %3, %3, %7\n\t"
			"v_xor_b32 %4, %4, %0\n\tv_xor_b32 %5, %5, %1\n\tv_xor_b32 %6, %6, %2\n\tv_xor_b32 %7, %7, %3\n\t"
			"v_and_b32 %0, 0x7fffffff, %0\n\tv_and_b32 %1, 0x7fffffff, %1\n\tv_and_b32 %2, 0x7fffffff, %2\n\tv_and_b32 %3, 0x7fffffff, %3\n\t"
			"v_lshlrev_b32 %4, 1, %4\n\tv_lshlrev_b32 %5, 1, %5\n\tv_lshlrev_b32 %6, 1, %6\n\tv_lshlrev_b32 %7, 1, %7\n\t"
			"v_alignbit_b32 %0, %0, %4, 7\n\tv_alignbit_b32 %1, %1, %5, 7\n\tv_alignbit_b32 %2, %2, %6, 7\n\tv_alignbit_b32 %3, %3, %7, 7\n\t"
			"v_add_u32 %4, %4, %1\n\tv_add_u32 %5, %5, %2\n\tv_add_u32 %6, %6, %3\n\tv_add_u32 %7, %7, %0\n\t"
			"v_xor_b32 %0, %0, %5\n\tv_xor_b32 %1, %1, %6\n\tv_xor_b32 %2, %2, %7\n\tv_xor_b32 %3, %3, %4\n\t"
			"v_or_b32 %4, 1, %4\n\tv_or_b32 %5, 1, %5\n\tv_or_b32 %6, 1, %6\n\tv_or_b32 %7, 1, %7\n\t"
			: "+v"( a ), "+v"( b ), "+v"( c ), "+v"( d ), "+v"( e ), "+v"( f ), "+v"( g ), "+v"( h ) );
	}
	out[ blockIdx.x * blockDim.x + threadIdx.x ] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}

template< int W >
static int run( unsigned *d_out, int n_cu, int iters, double *rate )
{
	hipEvent_t	e0, e1;
	CHK( hipEventCreate( &e0 ) );
	CHK( hipEventCreate( &e1 ) );
	const int	grid = n_cu * W;		// W workgroups of 4 waves per CU = W waves per SIMD
	hipLaunchKernelGGL( valu_loop<W>, dim3( grid ), dim3( 256 ), 0, 0, d_out, iters / 8, 1u );	// warm up
	CHK( hipDeviceSynchronize() );
	float	best = 1e30f;
	for( int rep = 0; rep < 5; rep++ ){
		CHK( hipEventRecord( e0, 0 ) );
		hipLaunchKernelGGL( valu_loop<W>, dim3( grid ), dim3( 256 ), 0, 0, d_out, iters, unsigned( rep ) );
		CHK( hipEventRecord( e1, 0 ) );
		CHK( hipEventSynchronize( e1 ) );
		float	ms;
		CHK( hipEventElapsedTime( &ms, e0, e1 ) );
		if( ms < best )
			best = ms;
	}
	const double	wave_instr = double( grid ) * 4 * double( iters ) * PER_ITER;
	*rate = wave_instr / ( best * 1e-3 ) / 1e9;
	return 0;
}

int main()
{
	hipDeviceProp_t	prop;
	CHK( hipGetDeviceProperties( &prop, 0 ) );
	const int	n_cu = prop.multiProcessorCount;
	unsigned	*d_out;
	CHK( hipMalloc( &d_out, size_t( n_cu ) * 8 * 256 * sizeof( unsigned ) ) );
	const int	iters = 200000;
	double	r1, r2, r4, r8;
	if( run<1>( d_out, n_cu, iters, &r1 ) || run<2>( d_out, n_cu, iters, &r2 ) || run<4>( d_out, n_cu, iters, &r4 ) ||
		run<8>( d_out, n_cu, iters, &r8 ) )
		return 1;
	const double	nominal = n_cu * 4 * 2.4e9 / 2 / 1e9;
	printf( "{\"what\": \"integer VALU issue rate, G wave64-instructions/s, independent v_add/xor/and/lshl/alignbit/or\", "
		"\"device\": \"%s\", \"cus\": %d, \"waves_per_simd_1\": %.1f, \"waves_per_simd_2\": %.1f, \"waves_per_simd_4\": %.1f, "
		"\"waves_per_simd_8\": %.1f, \"nominal_2_cycles_at_2.4GHz\": %.1f}\n",
		prop.name, n_cu, r1, r2, r4, r8, nominal );
	return 0;
}

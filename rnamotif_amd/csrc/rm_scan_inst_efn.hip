// rm_scan_inst_efn.hip -- the efn kernel (rm_scan_kernel.h) and its launcher.
#include "rm_scan_kernel.h"
hipError_t rmk_launch_efn( int grid, hipStream_t s, const rmk_efn_args &a )
{
	hipLaunchKernelGGL( rma_efn_kernel<EFN_BLOCK>, dim3( unsigned( grid ) ), dim3( EFN_BLOCK ), 0, s,
		a.d_prog, a.db, a.hits, a.n_hits, a.t16, a.tlkey, a.loginc, a.e2 );
	return hipGetLastError();
}

// the kernel's code object loaded now and not by its first launch (rma_scanner_warmup: 13 ms measured
// in the first batch of a search, whose warm-up scan finds no candidate to give the kernel)
hipError_t rmk_preload_efn( void )
{
	hipFuncAttributes	a;
	return hipFuncGetAttributes( &a, reinterpret_cast<const void *>( &rma_efn_kernel<EFN_BLOCK> ) );
}

// ... in workgroups of one wave that stage nothing (rma_efn_kernel, STAGE = false): grid = those workgroups
hipError_t rmk_launch_efn_light( int grid, hipStream_t s, const rmk_efn_args &a )
{
	hipLaunchKernelGGL( rma_efn_light_kernel<0>, dim3( unsigned( grid ) ), dim3( 64 ), EFN_LIGHT_LDS, s,
		a.d_prog, a.db, a.hits, a.n_hits, a.t16, a.tlkey, a.loginc, a.e2 );
	return hipGetLastError();
}

// rm_scan_inst_efn.hip -- the efn kernel (rm_scan_kernel.h) and its launcher.
#include "rm_scan_kernel.h"
hipError_t rmk_launch_efn( int grid, hipStream_t s, const rmk_efn_args &a )
{
	hipLaunchKernelGGL( rma_efn_kernel<EFN_BLOCK>, dim3( unsigned( grid ) ), dim3( EFN_BLOCK ), 0, s,
		a.d_prog, a.db, a.hits, a.n_hits, a.t16, a.tlkey, a.loginc, a.e2 );
	return hipGetLastError();
}

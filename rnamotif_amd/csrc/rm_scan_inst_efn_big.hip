// rm_scan_inst_efn_big.hip -- the efn kernel (rm_scan_kernel.h) for descriptors with an efn() / efn2() call over more than
// 15 helices (rmd_program_t::efn_big): the stacks of the loops' walks sized for the fifty helices a descriptor can have.
#include "rm_scan_kernel.h"
hipError_t rmk_launch_efn_big( int grid, hipStream_t s, const rmk_efn_args &a )
{
	hipLaunchKernelGGL( ( rma_efn_kernel<EFN_BLOCK, 1> ), dim3( unsigned( grid ) ), dim3( EFN_BLOCK ), 0, s,
		a.d_prog, a.db, a.hits, a.n_hits, a.t16, a.tlkey, a.loginc, a.e2 );
	return hipGetLastError();
}

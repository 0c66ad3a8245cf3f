// rm_scan_inst_gen_pktq.hip -- one instance of rma_search_kernel (rm_scan_kernel.h) and its launcher.
#include "rm_scan_kernel.h"
RMK_DEFINE_LAUNCHER( rmk_launch_gen_pktq, false, 1, RMD_KIND_PK | RMD_KIND_TQ, false )

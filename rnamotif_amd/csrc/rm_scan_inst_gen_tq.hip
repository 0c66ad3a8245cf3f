// rm_scan_inst_gen_tq.hip -- one instance of rma_search_kernel (rm_scan_kernel.h) and its launcher.
#include "rm_scan_kernel.h"
RMK_DEFINE_LAUNCHER( rmk_launch_gen_tq, false, 1, RMD_KIND_TQ, false )

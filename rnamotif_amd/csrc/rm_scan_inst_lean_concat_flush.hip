// rm_scan_inst_lean_concat_flush.hip -- one instance of rma_search_kernel (rm_scan_kernel.h) and its launcher.
#include "rm_scan_kernel.h"
RMK_DEFINE_LAUNCHER( rmk_launch_lean_concat_flush, true, 1, 0, true, true, false )

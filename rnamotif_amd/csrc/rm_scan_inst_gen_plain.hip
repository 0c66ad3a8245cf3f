// rm_scan_inst_gen_plain.hip -- one instance of rma_search_kernel (rm_scan_kernel.h) and its launcher.
#include "rm_scan_kernel.h"
RMK_DEFINE_LAUNCHER( rmk_launch_gen_plain, false, 1, 0, false )

// rm_scan_inst_gen_wide.hip -- one instance of rma_search_kernel (rm_scan_kernel.h) and its launcher: every element kind, and
// helices of 64 to 127 base pairs -- sets of helix lengths are two words there (rmd_lset_t, rm_scan_core.h).
#include "rm_scan_kernel.h"
RMK_DEFINE_LAUNCHER( rmk_launch_gen_wide, false, 1, RMD_KIND_PK | RMD_KIND_TQ | RMD_KIND_WIDE, false )

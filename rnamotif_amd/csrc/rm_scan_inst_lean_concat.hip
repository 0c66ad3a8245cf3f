// rm_scan_inst_lean_concat.hip -- one instance of rma_search_kernel (rm_scan_kernel.h) and its launcher: the pooled lean
// instance over tiles that lie over the concatenation of the entries (databases of short entries).
#include "rm_scan_kernel.h"
RMK_DEFINE_LAUNCHER( rmk_launch_lean_concat, true, 1, 0, true, true )

// rm_scan_inst_gen_pktq_concat.hip -- one instance of rma_search_kernel (rm_scan_kernel.h) and its launcher: the general instance
// of this class of descriptor over tiles that lie over the concatenation of the entries (databases of short entries).
#include "rm_scan_kernel.h"
RMK_DEFINE_LAUNCHER( rmk_launch_gen_pktq_concat, false, 1, RMD_KIND_PK | RMD_KIND_TQ, false, true )

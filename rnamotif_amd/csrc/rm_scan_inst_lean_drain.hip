// rm_scan_inst_lean_drain.hip -- rma_drain_kernel (rm_scan_kernel.h), the second kernel of the pooled lean
// instance, and its launcher.
#include "rm_scan_kernel.h"
RMK_DEFINE_DRAIN_LAUNCHER( rmk_launch_lean_drain )

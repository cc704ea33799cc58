/* Test-support entry points of libdemonet_hip.so that are NOT part of the drop-in boundary (include/demonet_hip.h).
 * The product library exports exactly these three beside the boundary; everything else named dn_debug_* (phase stamps,
 * tile forcing) exists only in the dev build (python -m demonet_amd.build --stamps, -DDN_DEV_STAMPS).               */
#ifndef DEMONET_HIP_DEBUG_H
#define DEMONET_HIP_DEBUG_H

#include "demonet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Number of head_fused_kernel launches issued by this process so far (tests assert that the fused path was taken). */
DN_API int dn_debug_head_fused_launches(void);

/* ... and how many of them carried the softmax / decode epilogue (scores, boxes and histogram rows instead of logits). */
DN_API int dn_debug_head_softmax_launches(void);

/* Drops the plan's cached hipGraph executables (tests re-capture with other DN_* knobs in one process). 0 on success. */
DN_API int dn_debug_clear_graphs(dn_plan* plan);

#ifdef __cplusplus
}
#endif
#endif

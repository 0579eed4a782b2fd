/* visper_hip_debug.h — measurement and development entry points of libvisper_hip.so.  NOT part of the drop-in ABI (include/visper_hip.h):
 * they exist only in a library built with -DVP_DEBUG (the Makefile's default, `make VP_DEBUG=0` builds the sealed product library) and are
 * used by bench.py's sustained-clock probe and by tools/ only; nothing in visper_lm_amd's step path calls them. */
#ifndef VISPER_HIP_DEBUG_H
#define VISPER_HIP_DEBUG_H
#include "visper_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* tools/gemm_interference.py: `blocks` workgroups pinning 64 KB of LDS each and spinning for `cycles` shader cycles (a stand-in for a
 * collective kernel holding CUs). */
int vp_debug_occupy(int blocks, long cycles, vp_stream_t stream);
/* bench.py clock probe, tools/gemm_stamps.py: 0x10000 = the persistent GEMM kernels write wall-clock / shader-cycle stamps of their first
 * output tile (256 blocks x 8 longs), read back with vp_debug_stamps. */
int vp_debug_gemm_flags(int flags);
int vp_debug_stamps(long* host);
/* tools/attn_phase_stamps.py: phase cycle sums of the generic D = 128 forward (VP_ATTN_DBG=1 launches); 16 longs. */
int vp_debug_attn_stamps(long* host);
/* tools/emb_loss_debug.py: device buffer of 8 int64 for in-kernel wall-clock stamps of later vp_emb_loss_fwd calls; NULL = off. */
int vp_debug_emb_loss_stamps(long long* dev_buf);

#ifdef __cplusplus
}
#endif
#endif

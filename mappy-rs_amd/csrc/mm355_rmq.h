// mm355_rmq.h -- launch interface of the device mg_lchain_rmq (mm355_rmq.hip)
#pragma once
#include "mm355_dev.h"

struct RmqParams {            // arguments of U:lchain.c::mg_lchain_rmq + the rescue test of U:map.c::mm_map_frag
	int32_t max_dist, max_dist_inner, bw, max_chn_skip, cap;
	float pen_gap, pen_skip;
	int32_t rescue_size; float rescue_ratio;
	int32_t primary;          // 1: MM_F_RMQ presets -- chain all sorted anchors of the listed reads; 0: long-join re-chain of the chained anchors
};

// per-read state after the launch (d_flag): 0 = not re-chained (the chains of mg_lchain_dp stand), 1 = re-chained on the device (n_u / n_v /
// u[] / compacted anchors replaced), 2 = the device could not prove its answer unique (equal range-minimum priorities) or ran out of its
// LDS capacities: the read's anchors are left sorted by x for the literal host implementation
#define MM355_RMQ_KEEP 0
#define MM355_RMQ_DONE 1
#define MM355_RMQ_HOST 2
#define MM355_RMQ_HOST_ALL 3    // host-side summary only (hb.rmq_state): nothing of this read was chained by mg_lchain_rmq on the device -- MM_F_RMQ
                                // presets whose primary pass handed the read back, or the stage switched off: the host runs every mg_lchain_rmq call
#ifdef __HIPCC__
int mm355_launch_rmq(const RmqParams &rp, const DevParams &pr, const DevBatch &bt, DevAnchors &an, const int32_t *d_list, int n_list, uint8_t *d_flag,
                     int *err, unsigned long long *ctr, unsigned int *qctr, hipStream_t st, void *kt = 0);
#endif

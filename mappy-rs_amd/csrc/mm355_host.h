// mm355_host.h -- host-side structures of libmm355 (index, options, batch context)
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include <unordered_map>
#include <mutex>
#include "../../include/mm355.h"
#include "mm355_core.h"

// HBM copy of the index on one device (flat table, pos[], the reference packed to 2 bits per base + its sorted N-run intervals, contig
// offsets/lengths).  Every context of that device shares it.  S (the .mmi's 4-bit image) only exists while the replica is being made.
struct mm355_replica { int dev = -1; void *slots = 0, *pos = 0, *S = 0, *seq_off = 0, *seq_len = 0, *S2 = 0, *nr = 0; uint32_t n_nr = 0; };

struct mm355_index {
	int32_t b, w, k, flag;
	uint32_t n_seq;
	std::vector<std::string> names;
	std::vector<uint64_t> seq_off;
	std::vector<uint32_t> seq_len;
	std::vector<uint32_t> S;            // 4-bit packed bases (codes 0..4), as in the .mmi
	uint64_t n_lines;                   // power of two
	std::vector<mm355_slot> slots;      // n_lines * 8
	std::vector<uint64_t> pos;          // positions of multi-occurrence minimizers
	int64_t n_minimizers, n_distinct;
	std::unordered_map<std::string, int> name2id;
	// device-built index (mm355_index_build_device): table and positions live only in HBM of device `dev_id`
	bool dev_resident = false; int dev_id = -1;
	void *d_slots = 0, *d_pos = 0, *d_S = 0; uint64_t n_pos = 0;
	std::vector<uint32_t> top_counts;   // largest occurrence counts, descending (for mm_idx_cal_max_occ)
	// per-device replicas (mm355_upload / first mm355_ctx_create on a device); the index itself stays immutable for the mapping path
	mutable std::mutex rep_mu;
	mutable std::vector<mm355_replica> replicas;
	mutable std::vector<uint64_t> nrun; mutable bool nrun_done = false;   // [beg, end) pairs of the runs of ambiguous bases in S (global offsets), found once
};
// turns the replica's 4-bit S (on the device) into the 2-bit image + N-run table the kernels read, and frees S (call under rep_mu)
int mm355_replica_pack2(const mm355_index *mi, mm355_replica *rp);
// finds the replica of `dev`, creating it when absent: H2D from the host image, or a peer copy from the device the index was built on
int mm355_index_replica(const mm355_index *mi, int dev, mm355_replica *out);
void mm355_index_free_replicas(mm355_index *mi);

// lookup on the host image of the flat table (used by tests of the table itself; the product looks up on the device)
inline uint32_t mm355_host_get(const mm355_index *mi, uint64_t minier, uint64_t *val)
{
	uint64_t line = mm_table_hash(minier) & (mi->n_lines - 1);
	for (;;) {
		const mm355_slot *ln = &mi->slots[line * MM355_SLOTS_PER_LINE];
		for (int q = 0; q < MM355_SLOTS_PER_LINE; ++q) {
			if (ln[q].key == UINT64_MAX) return 0;
			if ((ln[q].key >> 1) == minier) {
				if (ln[q].key & 1) { *val = ln[q].val; return 1; }
				*val = ln[q].val >> 32; return (uint32_t)ln[q].val;
			}
		}
		line = (line + 1) & (mi->n_lines - 1);
	}
}

int mm355_index_from_pairs(mm355_index *mi, std::vector<mm128> &pairs);   // pairs: x = minimizer (56 bit), y = position word
int32_t mm355_index_cal_max_occ(const mm355_index *mi, float f);

// mm355_host.h -- host-side structures of libmm355 (index, options, batch context)
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include <unordered_map>
#include "../../include/mm355.h"
#include "mm355_core.h"

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
};

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

// mm355_index.cpp -- host side of the index: MMI\2 reader, FASTA/FASTQ builder, the flat
// 128-B-line hash table that is uploaded to HBM, sequence accessors and option presets.
// Replaces, behind the C-ABI of include/mm355.h, the reference's FFI calls
//   mm_set_opt (lib.rs:333,336), mm_idx_reader_open/read/close (lib.rs:397-412),
//   mm_mapopt_update (lib.rs:414), mm_idx_index_name (lib.rs:416),
//   mm_idx_name2id (lib.rs:716), mm_idx_getseq (lib.rs:747).
// minimap2 2.26 units whose observable behaviour is kept: U:index.c (file format, key/value
// encoding, positions ascending inside a run, mm_idx_cal_max_occ), U:options.c (presets).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <algorithm>
#include <thread>
#include <atomic>
#include <zlib.h>
#include "mm355_host.h"

#define mm_seq4_set(s, i, c) ((s)[(i)>>3] |= (uint32_t)(c) << (((i)&7)<<2))
#define mm_seq4_get(s, i)    ((s)[(i)>>3] >> (((i)&7)<<2) & 0xf)

// ------------------------------------------------------------------ options (U:options.c)
static void idxopt_init(mm355_idxopt_t *o)
{
	memset(o, 0, sizeof(*o));
	o->k = 15, o->w = 10, o->flag = 0, o->bucket_bits = 14;
	o->mini_batch_size = 50000000; o->batch_size = 4000000000ULL;
}

static void mapopt_init(mm355_mapopt_t *o)
{
	memset(o, 0, sizeof(*o));
	o->seed = 11;
	o->mid_occ_frac = 2e-4f; o->min_mid_occ = 10; o->max_mid_occ = 1000000;
	o->sdust_thres = 0; o->q_occ_frac = 0.01f;
	o->min_cnt = 3; o->min_chain_score = 40;
	o->bw = 500; o->bw_long = 20000;
	o->max_gap = 5000; o->max_gap_ref = -1;
	o->max_chain_skip = 25; o->max_chain_iter = 5000;
	o->rmq_inner_dist = 1000; o->rmq_size_cap = 100000; o->rmq_rescue_size = 1000; o->rmq_rescue_ratio = 0.1f;
	o->chain_gap_scale = 0.8f; o->chain_skip_scale = 0.0f;
	o->max_max_occ = 4095; o->occ_dist = 500;
	o->mask_level = 0.5f; o->mask_len = INT_MAX; o->pri_ratio = 0.8f; o->best_n = 5;
	o->alt_drop = 0.15f;
	o->a = 2; o->b = 4; o->q = 4; o->e = 2; o->q2 = 24; o->e2 = 1; o->sc_ambi = 1;
	o->zdrop = 400; o->zdrop_inv = 200; o->end_bonus = -1;
	o->min_dp_max = o->min_chain_score * o->a;
	o->min_ksw_len = 200; o->max_clip_ratio = 1.0f;
	o->max_sw_mat = 100000000;
}

extern "C" int mm355_set_opt(const char *preset, mm355_idxopt_t *io, mm355_mapopt_t *mo)
{
	if (preset == 0) { idxopt_init(io); mapopt_init(mo); return 0; }
	std::string p(preset);
	if (p == "map-ont") return 0;
	if (p == "map-hifi" || p == "map-ccs") {
		io->flag = 0; io->k = 19; io->w = 19;
		mo->max_gap = 10000;
		mo->a = 1; mo->b = 4; mo->q = 6; mo->q2 = 26; mo->e = 2; mo->e2 = 1;
		mo->min_mid_occ = 50; mo->max_mid_occ = 500;
		mo->min_dp_max = 200;
		return 0;
	}
	if (p == "ava-ont") {
		io->flag = 0; io->k = 15; io->w = 5;
		mo->flag |= 0x800000LL | 0x001LL | 0x002LL | 0x400LL;
		mo->min_chain_score = 100; mo->pri_ratio = 0.0f; mo->max_chain_skip = 25;
		mo->bw = mo->bw_long = 2000; mo->occ_dist = 0;
		return 0;
	}
	if (p == "asm5" || p == "asm10" || p == "asm20") {
		io->flag = 0; io->k = 19; io->w = 19;
		mo->bw = 1000; mo->bw_long = 100000; mo->max_gap = 10000;
		mo->flag |= MMF_RMQ;
		mo->min_mid_occ = 50; mo->max_mid_occ = 500; mo->min_dp_max = 200; mo->best_n = 50;
		if (p == "asm5") { mo->a = 1; mo->b = 19; mo->q = 39; mo->q2 = 81; mo->e = 3; mo->e2 = 1; mo->zdrop = mo->zdrop_inv = 200; }
		else if (p == "asm10") { mo->a = 1; mo->b = 9; mo->q = 16; mo->q2 = 41; mo->e = 2; mo->e2 = 1; mo->zdrop = mo->zdrop_inv = 200; }
		else if (p == "asm20") { mo->a = 1; mo->b = 4; mo->q = 6; mo->q2 = 26; mo->e = 2; mo->e2 = 1; mo->zdrop = mo->zdrop_inv = 200; io->w = 10; }
		else return MM355_EINVAL;
		return 0;
	}
	if (p == "map-pb" || p == "map10k") {   // homopolymer-compressed minimizers (MM_I_HPC = 1), U:options.c::mm_set_opt
		io->flag |= 1; io->k = 19;
		return 0;
	}
	if (p == "ava-pb") {
		io->flag |= 1; io->k = 19; io->w = 5;
		mo->flag |= 0x800000LL | 0x001LL | 0x002LL | 0x400LL;   // ALL_CHAINS | NO_DIAG | NO_DUAL | NO_LJOIN, as ava-ont
		mo->min_chain_score = 100; mo->pri_ratio = 0.0f; mo->max_chain_skip = 25;
		mo->bw_long = mo->bw; mo->occ_dist = 0;
		return 0;
	}
	// known minimap2 2.26 presets that are outside the long-read hot path (short reads, spliced): nothing is modified, the
	// caller gets MM355_EUNSUP and must refuse (mappy_rs.Aligner raises)
	static const char *const unsup[] = { "sr", "short", "splice", "splice:hq", "cdna", 0 };
	for (int i = 0; unsup[i]; ++i) if (p == unsup[i]) return MM355_EUNSUP;
	return MM355_EINVAL;   // unknown name: options untouched, as U:options.c::mm_set_opt's -1 (the reference ignores it, lib.rs:336)
}

int32_t mm355_index_cal_max_occ(const mm355_index *mi, float f)
{
	if (f <= 0.) return INT32_MAX;
	size_t n = (size_t)mi->n_distinct;
	if (n == 0) return INT32_MAX;
	if (mi->dev_resident) {   // k-th smallest == (n-1-k)-th largest of the saved descending tail
		size_t kk = (uint32_t)((1. - f) * n);
		if (kk >= n) kk = n - 1;
		size_t from_top = n - 1 - kk;
		if (from_top >= mi->top_counts.size()) return mi->top_counts.empty()? INT32_MAX : (int32_t)(mi->top_counts.back() + 1);
		return (int32_t)(mi->top_counts[from_top] + 1);
	}
	std::vector<uint32_t> a; a.reserve(n);
	for (const mm355_slot &s : mi->slots)
		if (s.key != UINT64_MAX) a.push_back(s.key & 1? 1u : (uint32_t)s.val);
	size_t kk = (uint32_t)((1. - f) * n);
	if (kk >= a.size()) kk = a.size() - 1;
	std::nth_element(a.begin(), a.begin() + kk, a.end());
	return (int32_t)(a[kk] + 1);
}

extern "C" int mm355_mapopt_update(mm355_mapopt_t *opt, const mm355_index_t *mi)
{
	if (mi == 0) return MM355_ENOIDX;
	if (opt->mid_occ <= 0) {
		opt->mid_occ = mm355_index_cal_max_occ(mi, opt->mid_occ_frac);
		if (opt->mid_occ < opt->min_mid_occ) opt->mid_occ = opt->min_mid_occ;
		if (opt->max_mid_occ > opt->min_mid_occ && opt->mid_occ > opt->max_mid_occ) opt->mid_occ = opt->max_mid_occ;
	}
	if (opt->bw_long < opt->bw) opt->bw_long = opt->bw;
	return 0;
}

// ------------------------------------------------------------------ flat table
static void table_insert(mm355_index *mi, uint64_t minier, bool single, uint64_t val)
{
	uint64_t mask = mi->n_lines - 1, line = mm_table_hash(minier) & mask;
	for (;;) {
		mm355_slot *ln = &mi->slots[line * MM355_SLOTS_PER_LINE];
		for (int q = 0; q < MM355_SLOTS_PER_LINE; ++q)
			if (ln[q].key == UINT64_MAX) { ln[q].key = minier << 1 | (single? 1 : 0); ln[q].val = val; return; }
		line = (line + 1) & mask;
	}
}

static void table_alloc(mm355_index *mi, uint64_t n_keys)
{
	uint64_t want = (uint64_t)(n_keys / 0.55) + MM355_SLOTS_PER_LINE, n_lines = 1;
	while (n_lines * MM355_SLOTS_PER_LINE < want) n_lines <<= 1;
	mi->n_lines = n_lines;
	mm355_slot empty = { UINT64_MAX, 0 };
	mi->slots.assign(n_lines * MM355_SLOTS_PER_LINE, empty);
}

// pairs: x = minimizer value (hash, 2k bits), y = rid<<32|pos<<1|strand.  Sorted here by (x, y): positions of
// one minimizer end up ascending, which is the order U:index.c::worker_post leaves in p[] (radix_sort_64).
int mm355_index_from_pairs(mm355_index *mi, std::vector<mm128> &a)
{
	std::sort(a.begin(), a.end(), [](const mm128 &p, const mm128 &q) { return p.x < q.x || (p.x == q.x && p.y < q.y); });
	uint64_t n_keys = 0, n_multi = 0;
	for (size_t i = 0; i < a.size();) {
		size_t j = i + 1;
		while (j < a.size() && a[j].x == a[i].x) ++j;
		++n_keys; if (j - i > 1) n_multi += j - i;
		i = j;
	}
	table_alloc(mi, n_keys);
	mi->pos.clear(); mi->pos.reserve(n_multi);
	for (size_t i = 0; i < a.size();) {
		size_t j = i + 1;
		while (j < a.size() && a[j].x == a[i].x) ++j;
		if (j - i == 1) table_insert(mi, a[i].x, true, a[i].y);
		else {
			uint64_t off = mi->pos.size();
			for (size_t k = i; k < j; ++k) mi->pos.push_back(a[k].y);
			table_insert(mi, a[i].x, false, off << 32 | (uint64_t)(j - i));
		}
		i = j;
	}
	mi->n_minimizers = (int64_t)a.size(); mi->n_distinct = (int64_t)n_keys;
	return 0;
}

// ------------------------------------------------------------------ MMI\2 reader (U:index.c::mm_idx_load)
static mm355_index *load_mmi(FILE *fp)
{
	char magic[4]; uint32_t x[5];
	if (fread(magic, 1, 4, fp) != 4 || strncmp(magic, "MMI\2", 4) != 0) return 0;
	if (fread(x, 4, 5, fp) != 5) return 0;
	mm355_index *mi = new mm355_index();
	mi->w = x[0], mi->k = x[1], mi->b = x[2], mi->n_seq = x[3], mi->flag = x[4];
	// a corrupt header must not drive the 1<<b bucket loop or the sketch kernels: U:sketch.c asserts 0 < w < 256, 0 < k <= 28; b <= 2k
	if (x[0] < 1 || x[0] > 255 || x[1] < 1 || x[1] > 28 || x[2] > 28 || x[2] > 2 * x[1]) { delete mi; return 0; }
	uint64_t sum_len = 0;
	for (uint32_t i = 0; i < mi->n_seq; ++i) {
		uint8_t l; uint32_t len; char nm[256];
		if (fread(&l, 1, 1, fp) != 1) { delete mi; return 0; }
		if (l && fread(nm, 1, l, fp) != l) { delete mi; return 0; }
		if (fread(&len, 4, 1, fp) != 1) { delete mi; return 0; }
		mi->names.emplace_back(nm, l); mi->seq_off.push_back(sum_len); mi->seq_len.push_back(len);
		sum_len += len;
	}
	// buckets -> (minimizer, positions); collected first to size the flat table
	struct Ent { uint64_t minier, val; uint32_t n; };
	std::vector<Ent> ents;
	std::vector<uint64_t> pos;
	for (uint64_t i = 0; i < 1ULL << mi->b; ++i) {
		int32_t n; uint32_t size;
		if (fread(&n, 4, 1, fp) != 1) { delete mi; return 0; }
		uint64_t base = pos.size();
		pos.resize(base + (n > 0? n : 0));
		if (n > 0 && fread(&pos[base], 8, n, fp) != (size_t)n) { delete mi; return 0; }
		if (fread(&size, 4, 1, fp) != 1) { delete mi; return 0; }
		for (uint32_t j = 0; j < size; ++j) {
			uint64_t kv[2];
			if (fread(kv, 8, 2, fp) != 2) { delete mi; return 0; }
			Ent e; e.minier = (kv[0] >> 1) << mi->b | i;
			if (kv[0] & 1) e.n = 1, e.val = kv[1];
			else e.n = (uint32_t)kv[1], e.val = base + (kv[1] >> 32);
			ents.push_back(e);
		}
	}
	table_alloc(mi, ents.size());
	mi->pos.swap(pos);
	int64_t tot = 0;
	for (const Ent &e : ents) {
		if (e.n == 1) table_insert(mi, e.minier, true, e.val);
		else table_insert(mi, e.minier, false, e.val << 32 | e.n);
		tot += e.n;
	}
	mi->n_minimizers = tot; mi->n_distinct = (int64_t)ents.size();
	if (!(mi->flag & 2)) {
		size_t n = (sum_len + 7) / 8;
		mi->S.resize(n? n : 1);
		if (fread(mi->S.data(), 4, n, fp) != n) { delete mi; return 0; }
	}
	return mi;
}

// ------------------------------------------------------------------ builder (U:index.c::mm_idx_gen)
struct HostBase { const char *s; int operator()(int i) const { return mm_nt4((uint8_t)s[i]); } };

static void sketch_contig(const char *s, int64_t len, int w, int k, uint32_t rid, bool is_hpc, std::vector<mm128> &out)
{
	if (len <= 0) return;
	std::vector<mm128> ring(w);
	std::vector<mm128> tmp((size_t)len);
	HostBase hb = { s };
	int64_t n = mm_sketch_seq(hb, (int)len, w, k, rid, tmp.data(), len, ring.data(), 1, is_hpc);
	for (int64_t i = 0; i < n; ++i) { mm128 m; m.x = tmp[i].x >> 8; m.y = tmp[i].y; out.push_back(m); }
}

static mm355_index *build_from_seqs(const mm355_idxopt_t *io, int n_seq, const char *const *seqs, const int64_t *lens, const char *const *names, int n_threads)
{
	mm355_index *mi = new mm355_index();
	mi->w = io->w < 1? 1 : io->w; mi->k = io->k; mi->b = io->bucket_bits; mi->flag = io->flag; mi->n_seq = n_seq;
	if (mi->k * 2 < mi->b) mi->b = mi->k * 2;
	uint64_t sum_len = 0;
	for (int i = 0; i < n_seq; ++i) {
		mi->names.emplace_back(names && names[i]? names[i] : "");
		mi->seq_off.push_back(sum_len); mi->seq_len.push_back((uint32_t)lens[i]);
		sum_len += lens[i];
	}
	mi->S.assign((sum_len + 7) / 8 + 1, 0);
	for (int i = 0; i < n_seq; ++i)
		for (int64_t j = 0; j < lens[i]; ++j) { uint64_t o = mi->seq_off[i] + j; int c = mm_nt4((uint8_t)seqs[i][j]); mm_seq4_set(mi->S.data(), o, c); }
	// sketch contigs on host threads (a contig is sequential; contigs are independent)
	if (n_threads < 1) n_threads = 1;
	std::vector<std::vector<mm128>> parts(n_seq);
	std::atomic<int> next(0);
	auto work = [&]() { for (;;) { int i = next.fetch_add(1); if (i >= n_seq) break; sketch_contig(seqs[i], lens[i], mi->w, mi->k, (uint32_t)i, (mi->flag & 1) != 0, parts[i]); } };
	std::vector<std::thread> th;
	for (int t = 1; t < n_threads && t < n_seq; ++t) th.emplace_back(work);
	work();
	for (auto &t : th) t.join();
	std::vector<mm128> all;
	size_t tot = 0; for (auto &p : parts) tot += p.size();
	all.reserve(tot);
	for (auto &p : parts) { all.insert(all.end(), p.begin(), p.end()); std::vector<mm128>().swap(p); }
	mm355_index_from_pairs(mi, all);
	return mi;
}

// FASTA/FASTQ, plain or gzip-compressed (U:bseq.c reads through zlib's gzFile, which passes plain files through unchanged)
static mm355_index *build_from_fastx(const char *path, const mm355_idxopt_t *io, int n_threads)
{
	gzFile gz = gzopen(path, "rb");
	if (gz == 0) return 0;
	(void)gzbuffer(gz, 1 << 20);
	std::vector<std::string> names, seqs;
	std::string line;
	std::vector<char> buf(1 << 16);
	bool in_qual = false, is_fq = false, eof = false; size_t l_qual = 0;
	while (!eof) {
		line.clear();
		for (;;) {   // one line of any length
			if (gzgets(gz, buf.data(), (int)buf.size()) == 0) { eof = true; break; }
			const size_t l = strlen(buf.data());
			line.append(buf.data(), l);
			if (l > 0 && buf[l - 1] == '\n') break;
		}
		if (eof && line.empty()) break;
		size_t n = line.size();
		while (n > 0 && (line[n-1] == '\n' || line[n-1] == '\r')) --n;
		line.resize(n);
		if (in_qual) { l_qual += n; if (l_qual >= seqs.back().size()) in_qual = false; continue; }
		if (n > 0 && (line[0] == '>' || (line[0] == '@' && (names.empty() || is_fq)))) {
			is_fq = line[0] == '@';
			size_t e = 1; while (e < n && line[e] != ' ' && line[e] != '\t') ++e;
			names.emplace_back(line, 1, e - 1); seqs.emplace_back();
		} else if (n > 0 && line[0] == '+' && is_fq) { in_qual = !seqs.empty() && seqs.back().size() > 0; l_qual = 0; }
		else if (!names.empty()) { for (size_t i = 0; i < n; ++i) if (line[i] > ' ') seqs.back().push_back(line[i]); }
	}
	int zerr = 0; (void)gzerror(gz, &zerr);
	gzclose(gz);
	if (zerr != Z_OK && zerr != Z_STREAM_END) return 0;   // truncated / corrupt gzip stream: no garbage index
	if (names.empty()) return 0;
	std::vector<const char*> sp, np; std::vector<int64_t> ln;
	for (size_t i = 0; i < names.size(); ++i) { sp.push_back(seqs[i].data()); np.push_back(names[i].c_str()); ln.push_back((int64_t)seqs[i].size()); }
	return build_from_seqs(io, (int)names.size(), sp.data(), ln.data(), np.data(), n_threads);
}

static void finish_index(mm355_index *mi)
{
	for (uint32_t i = 0; i < mi->n_seq; ++i) mi->name2id.emplace(mi->names[i], (int)i);   // first wins, as a hash put would report a duplicate
}

extern "C" int mm355_index_load(const char *path, const mm355_idxopt_t *io, int n_threads, mm355_index_t **out)
{
	*out = 0;
	FILE *fp = fopen(path, "rb");
	if (fp == 0) return MM355_EIO;
	char magic[4]; size_t n = fread(magic, 1, 4, fp);
	rewind(fp);
	mm355_index *mi = 0;
	if (n == 4 && strncmp(magic, "MMI\2", 4) == 0) mi = load_mmi(fp);
	else if (n > 0) {
		fclose(fp); fp = 0;
		if (io->k <= 0 || io->k > 28 || io->w <= 0 || io->w >= 256) return MM355_EINVAL;
		mi = build_from_fastx(path, io, n_threads);
	}
	if (fp) fclose(fp);
	if (mi == 0 || mi->n_seq == 0) { delete mi; return MM355_EIO; }
	finish_index(mi);
	*out = mi;
	return 0;
}

extern "C" int mm355_index_build(const mm355_idxopt_t *io, int n_seq, const char *const *seqs, const int64_t *lens, const char *const *names, int n_threads, mm355_index_t **out)
{
	*out = 0;
	if (n_seq <= 0) return MM355_EINVAL;
	if (io->k <= 0 || io->k > 28 || io->w <= 0 || io->w >= 256) return MM355_EINVAL;   // U:sketch.c::mm_sketch asserts the same ranges
	mm355_index *mi = build_from_seqs(io, n_seq, seqs, lens, names, n_threads);
	if (mi == 0) return MM355_EINVAL;
	finish_index(mi);
	*out = mi;
	return 0;
}

extern "C" void mm355_index_free(mm355_index_t *mi) { if (mi) mm355_index_free_replicas(mi); delete mi; }

extern "C" int mm355_index_info(const mm355_index_t *mi, int32_t *k, int32_t *w, int32_t *b, int32_t *flag, uint32_t *n_seq)
{
	if (mi == 0) return MM355_ENOIDX;
	if (k) *k = mi->k; if (w) *w = mi->w; if (b) *b = mi->b; if (flag) *flag = mi->flag; if (n_seq) *n_seq = mi->n_seq;
	return 0;
}
extern "C" const char *mm355_index_seq_name(const mm355_index_t *mi, uint32_t rid) { return mi && rid < mi->n_seq? mi->names[rid].c_str() : 0; }
extern "C" int64_t mm355_index_seq_len(const mm355_index_t *mi, uint32_t rid) { return mi && rid < mi->n_seq? (int64_t)mi->seq_len[rid] : -1; }
extern "C" int mm355_index_name2id(const mm355_index_t *mi, const char *name)
{
	if (mi == 0) return -2;
	auto it = mi->name2id.find(name);
	return it == mi->name2id.end()? -1 : it->second;
}
// U:index.c::mm_idx_getseq
extern "C" int mm355_index_getseq(const mm355_index_t *mi, uint32_t rid, uint32_t st, uint32_t en, uint8_t *seq)
{
	if (mi == 0 || (mi->flag & 2)) return -1;
	if (rid >= mi->n_seq || st >= mi->seq_len[rid]) return -1;
	if (en > mi->seq_len[rid]) en = mi->seq_len[rid];
	uint64_t st1 = mi->seq_off[rid] + st, en1 = mi->seq_off[rid] + en;
	for (uint64_t i = st1; i < en1; ++i) seq[i - st1] = mm_seq4_get(mi->S.data(), i);
	return (int)(en - st);
}
extern "C" int mm355_index_stat(const mm355_index_t *mi, int64_t *n_minimizers, int64_t *n_distinct, int64_t *table_bytes, int64_t *pos_bytes)
{
	if (mi == 0) return MM355_ENOIDX;
	if (n_minimizers) *n_minimizers = mi->n_minimizers;
	if (n_distinct) *n_distinct = mi->n_distinct;
	if (table_bytes) *table_bytes = (int64_t)(mi->dev_resident? mi->n_lines * MM355_SLOTS_PER_LINE : mi->slots.size()) * 16;
	if (pos_bytes) *pos_bytes = (int64_t)(mi->dev_resident? mi->n_pos : mi->pos.size()) * 8;
	return 0;
}

// host-side diagnostic view of mm_idx_get on the flat table (the product path looks up on the device: k_seed_lookup)
extern "C" int mm355_index_get(const mm355_index_t *mi, uint64_t minier, uint64_t *vals, int cap)
{
	if (mi == 0) return MM355_ENOIDX;
	if (mi->dev_resident) return MM355_EUNSUP;   // the table of a device-built index exists only in HBM
	uint64_t v = 0;
	uint32_t n = mm355_host_get(mi, minier, &v);
	if (n == 1) { if (cap > 0) vals[0] = v; }
	else for (uint32_t i = 0; i < n && (int)i < cap; ++i) vals[i] = mi->pos[v + i];
	return (int)n;
}

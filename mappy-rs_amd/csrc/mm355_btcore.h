// mm355_btcore.h -- U:lchain.c::mg_chain_backtrack + compact_a for one read by one wave; shared by k_backtrack (after mg_lchain_dp,
// mm355_kernels.hip) and k_rmq_backtrack (after mg_lchain_rmq, mm355_rmq.hip).
#pragma once
#include "mm355_wave.h"
#ifndef KPROF_BEGIN
#define KPROF_BEGIN(bt) unsigned long long *kp_ = (bt).prof; unsigned long long kp_t_ = kp_? (unsigned long long)clock64() : 0
#define KPROF(i) do { if (kp_ && (threadIdx.x & 63) == 0) { const unsigned long long t_ = (unsigned long long)clock64(); atomicAdd(&kp_[i], t_ - kp_t_); atomicMax(&kp_[32 + (i)], t_ - kp_t_); kp_t_ = t_; } } while (0)
#endif
// ------------------------------------------------------------------ a8: mg_chain_backtrack + compact_a
#define Z_STAGE 2048      // 16 KB: with SortLds five blocks per CU (4096 entries: three)
// One wave: backtrack + compact_a of read r over its first n anchors (f / p filled by a chaining kernel); max_drop = the band width
// that chainer used (U:lchain.c: mg_lchain_dp passes bw, mg_lchain_rmq passes its own bw).  Leaves n_u[r], n_v[r], u[] and the compacted
// anchors in place of a[].
struct BtLds { SortLds L; uint64_t zstage[Z_STAGE]; int s_nu, s_nv; };
__device__ inline void wave_backtrack_read(const DevParams &pr, const DevBatch &bt, const DevAnchors &an, int *err, BtLds *S, const int r, const int n, const int max_drop, const int kp_base = 0)
{
	SortLds &L = S->L;
	uint64_t *zstage = S->zstage;
	int &s_nu = S->s_nu, &s_nv = S->s_nv;
	const int lane = threadIdx.x;
	const int64_t o = an.aoff[r];
	if (lane == 0) { an.n_u[r] = 0; an.n_v[r] = 0; }
	if (n == 0) return;
	const mm128 *a = an.a + o;
	const int32_t *f = an.f + o, *p = an.p + o;
	uint64_t *z = an.z + o;
	uint8_t *t8 = an.t8 + o;
	int32_t *vi = an.vi + o;
	mm128 *b = an.b + o;
	uint64_t *u = an.u + o, *u2 = an.u2 + o;
	mm128 *wk = an.wk + o;
	const int min_sc = pr.min_chain_score, min_cnt = pr.min_cnt;
	KPROF_BEGIN(bt);
	// z[] = (f, i) for f >= min_sc, in anchor order
	int n_z = 0;
	int32_t f_top = 0;                                         // largest score of the list (of this lane; reduced below)
	for (int base = 0; base < n; base += 4 * WAVE) {   // four independent loads in flight per step (the loop is a chain of round trips otherwise)
		int32_t fi[4];
#pragma unroll
		for (int h = 0; h < 4; ++h) { const int i = base + h * WAVE + lane; fi[h] = i < n? f[i] : INT32_MIN; }
#pragma unroll
		for (int h = 0; h < 4; ++h) {
			const int i = base + h * WAVE + lane;
			const bool keep = i < n && fi[h] >= min_sc;
			const unsigned long long mask = __ballot(keep);
			if (keep) { z[n_z + __popcll(mask & LANE_LT_MASK(lane))] = (uint64_t)(uint32_t)fi[h] << 32 | (uint32_t)i; f_top = fi[h] > f_top? fi[h] : f_top; }
			n_z += __popcll(mask);
		}
	}
	__syncthreads();
	KPROF(kp_base + 0);
	if (n_z == 0) return;
	WalkScratch ws; ws.out = u2; ws.fpos = (uint32_t*)vi; ws.rank = (uint32_t*)(an.v + o); ws.flab = t8; ws.tcnt = 0;   // v[] is dead after the DP fill
	// (the scores are small positive numbers: the levels above their highest byte would be six histogram passes over the list for nothing)
	f_top = wave_reduce_max(f_top);
	const int z_s0 = min_sc < 0? 24 : f_top > 0? ((31 - __builtin_clz((unsigned)f_top)) >> 3) << 3 : 0;   // (a negative threshold lets negative scores in: all four bytes)
	wave_radix_sort(z, (uint32_t)n_z, key_hi32(), &L, zstage, (uint32_t)Z_STAGE, &ws, z_s0);
	if (lane == 0 && n_z > MM355_RS_MIN_SIZE && L.overflow) *err = 1;
	__syncthreads();
	// The walk below is a pointer chase: p, f and the mark of a node sit in ONE 8-byte word per anchor -- pf[i] = { (p + 1) | mark << 30, f }
	// in the u2[] region, free between the two sorts; the nodes a probe visits are remembered in LDS (the sort stage, idle here), so the
	// collect pass of U:lchain.c::mg_chain_backtrack stores without chasing again; and the "already used" test of the n_z candidates is
	// prefetched 64 at a time by the whole wave (a mark is final once it is 1, only zeros are re-read).
	int2 *pf = (int2*)u2;
	for (int base = 0; base < n; base += 4 * WAVE) {
		int2 w4[4];
#pragma unroll
		for (int h = 0; h < 4; ++h) { const int i = base + h * WAVE + lane; w4[h] = i < n? make_int2(p[i] + 1, f[i]) : make_int2(0, 0); }
#pragma unroll
		for (int h = 0; h < 4; ++h) { const int i = base + h * WAVE + lane; if (i < n) pf[i] = w4[h]; }
	}
	__syncthreads();
	KPROF(kp_base + 1);
	uint32_t *visited = (uint32_t*)zstage;                    // 2 * Z_STAGE entries
	const int VCAP = 2 * Z_STAGE;
	int n_v = 0, n_u = 0;
#define PF_P(w) ((int)((uint32_t)(w).x & 0x3fffffffu) - 1)
#define PF_MARK(w) ((uint32_t)(w).x >> 30)
	for (int kb = n_z - 1; kb >= 0; kb -= WAVE) {
		const int kk = kb - lane;
		uint64_t zk = 0; bool cand = false;
		if (kk >= 0) { zk = z[kk]; cand = PF_MARK(pf[(uint32_t)zk]) == 0; }
		unsigned long long todo = __ballot(cand);
		if (todo == 0) continue;                                  // wave-uniform
		const uint32_t zlo = (uint32_t)zk, zhi = (uint32_t)(zk >> 32);
		while (todo) {                                            // wave-uniform: a scalar probe on a register window, the whole wave marks
			const int l = __builtin_ctzll(todo); todo &= todo - 1;
			const int zi = (int)(uint32_t)__builtin_amdgcn_readlane((int)zlo, l);
			const int32_t zx = (int32_t)__builtin_amdgcn_readlane((int)zhi, l);
			// mg_chain_bk_end as WAVE-UNIFORM code.  p[i] < i and a chain's next anchor is almost always a few entries back in the x-sorted
			// array, so the wave keeps a window of 64 words pf[wbase - lane] in two registers (one coalesced load) and the probe reads its next
			// node with v_readlane (tens of cycles) instead of a dependent global load (hundreds); only a hop of more than 63 entries
			// reloads the window.  The reference's t[i] = 2 marks are dropped: indices strictly decrease along a probe, so it can never meet a
			// node it marked itself, and every 2 is reset (or becomes 1) before anybody else looks -- the probe stores nothing to HBM, a window
			// loaded at its start is fresh for its whole length (marks only change in the pass below).
			int wbase = zi, Wx, Wy;
			{ const int idx = wbase - lane; const int2 t = idx >= 0? pf[idx] : make_int2(0, 0); Wx = t.x; Wy = t.y; }
			int2 w = make_int2(__builtin_amdgcn_readlane(Wx, 0), __builtin_amdgcn_readlane(Wy, 0));
			if (PF_MARK(w) != 0) continue;
			int i = zi, end_i = -1, max_i = zi, nvis = 0, q = 0, vreg = 0;
			int32_t max_s = 0;
			for (;;) {
				vreg = lane == (nvis & 63)? i : vreg;                      // visited list: 64 nodes in a register, then one LDS store
				if ((nvis & 63) == 63) { const int b0 = nvis & ~63; if (b0 < VCAP) visited[b0 + lane] = (uint32_t)vreg; }
				++nvis;
				end_i = i = PF_P(w);
				int32_t sv = zx;
				if (i >= 0) {
					if (wbase - i > 63) { wbase = i; const int idx = wbase - lane; const int2 t = idx >= 0? pf[idx] : make_int2(0, 0); Wx = t.x; Wy = t.y; }
					w = make_int2(__builtin_amdgcn_readlane(Wx, wbase - i), __builtin_amdgcn_readlane(Wy, wbase - i));
					sv = zx - w.y;
				}
				if (sv > max_s) { max_s = sv; max_i = i; q = nvis; }
				else if (max_s - sv > max_drop) break;
				if (i < 0 || PF_MARK(w) != 0) break;
			}
			if (nvis & 63) { const int b0 = nvis & ~63; if (b0 < VCAP && lane < (nvis & 63)) visited[b0 + lane] = (uint32_t)vreg; }
			const bool over = nvis > VCAP;
			const int n_v0 = n_v;
			if (!over) {
				// marks: the first q visited nodes join the chain (1); the others keep their 0.  The word of a node needs no load: its p is the
				// next node of the list (the probe followed p), the last one's is the node the probe stopped at -- stores only, 64 nodes at a time
				for (int t = lane; t < q; t += WAVE) {
					const uint32_t vn = visited[t];
					const int nxt = t + 1 < nvis? (int)visited[t + 1] : end_i;
					pf[vn].x = (int)((uint32_t)(nxt + 1) | 1u << 30);
					vi[n_v0 + t] = (int)vn;
				}
				n_v = n_v0 + q;
			} else {       // a probe longer than the LDS list: chase again (one lane)
				int nv2 = n_v;
				if (lane == 0)
					for (int k = zi; k != max_i; k = PF_P(pf[k])) { vi[nv2++] = k; pf[k].x = (int)(((uint32_t)pf[k].x & 0x3fffffffu) | 1u << 30); }
				n_v = __builtin_amdgcn_readfirstlane(nv2);
			}
			const int32_t sc = q > 0? max_s : 0;
			if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) {
				if (lane == 0) u[n_u] = (uint64_t)(uint32_t)sc << 32 | (uint32_t)(n_v - n_v0);
				++n_u;
			} else n_v = n_v0;
		}
		__syncthreads();
	}
#undef PF_P
#undef PF_MARK
	if (lane == 0) { s_nu = n_u; s_nv = n_v; }
	__syncthreads();
	n_u = s_nu; n_v = s_nv;
	KPROF(kp_base + 2);
	if (n_u == 0) return;
	// compact_a: chains written forward; then chains re-ordered by the x of their first anchor.  Every step is a loop over ANCHORS, 64 at a
	// time, whatever the chains look like (a loop over chains -- one gather round trip per chain, most lanes idle on short chains -- was a
	// quarter of the slowest read's time: thousands of short chains on a repeat-rich read): the chain of position v is found by a prefix
	// maximum over head marks (chain index + 1 at the first position of a chain, 0 elsewhere; z[] is dead after the walk and holds them).
	int32_t *head = (int32_t*)z;                               // n_v entries (n_z >= n_v entries of 8 bytes are there)
	for (int v0 = 0; v0 < n_v; v0 += WAVE) if (v0 + lane < n_v) head[v0 + lane] = 0;
	__syncthreads();
	// (1) per-chain start offsets into wk[].y (k<<32|i), head marks
	for (int c0 = 0, k = 0; c0 < n_u; c0 += WAVE) {
		const int c = c0 + lane;
		const int32_t ni = c < n_u? (int32_t)u[c] : 0;
		const int32_t incl = wave_incl_scan_add(ni);
		if (c < n_u) { const int k0 = k + incl - ni; wk[c].y = (uint64_t)(uint32_t)k0 << 32 | (uint32_t)c; head[k0] = c + 1; }
		k += __builtin_amdgcn_readlane(incl, 63);
	}
	__syncthreads();
	// b[] filled in forward order: position v of chain c (start k0, ni anchors) takes the walk's entry k0 + (ni - 1 - (v - k0))
	for (int v0 = 0, carry = 0; v0 < n_v; v0 += WAVE) {
		const int v = v0 + lane;
		const int32_t h = v < n_v? head[v] : 0;
		int32_t cid = wave_incl_scan_max(h);
		cid = cid > carry? cid : carry;
		carry = __builtin_amdgcn_readlane(cid, 63);
		if (v < n_v) {
			const int c = cid - 1;
			const int k0 = (int)(wk[c].y >> 32), ni = (int32_t)u[c];
			b[v] = a[vi[2 * k0 + ni - 1 - v]];
		}
	}
	__syncthreads();
	for (int c = lane; c < n_u; c += WAVE) wk[c].x = b[wk[c].y >> 32].x;
	__syncthreads();
	wave_radix_sort(wk, (uint32_t)n_u, mm_key_x(), &L, (mm128*)zstage, (uint32_t)(Z_STAGE / 2));
	__syncthreads();
	// (2) final order: anchors go back into a[] region as the compacted list (written to an.a, length n_v)
	mm128 *aout = an.a + o;
	for (int v0 = 0; v0 < n_v; v0 += WAVE) if (v0 + lane < n_v) head[v0 + lane] = 0;
	__syncthreads();
	for (int c0 = 0, k = 0; c0 < n_u; c0 += WAVE) {
		const int c = c0 + lane;
		uint64_t uj = 0;
		if (c < n_u) uj = u[(uint32_t)wk[c].y];
		const int32_t ni = (int32_t)uj;
		const int32_t incl = wave_incl_scan_add(ni);
		if (c < n_u) { const int dst = k + incl - ni; u2[c] = uj; wk[c].x = (uint64_t)(uint32_t)dst; head[dst] = c + 1; }   // destination offset
		k += __builtin_amdgcn_readlane(incl, 63);
	}
	__syncthreads();
	for (int v0 = 0, carry = 0; v0 < n_v; v0 += WAVE) {
		const int v = v0 + lane;
		const int32_t h = v < n_v? head[v] : 0;
		int32_t cid = wave_incl_scan_max(h);
		cid = cid > carry? cid : carry;
		carry = __builtin_amdgcn_readlane(cid, 63);
		if (v < n_v) {
			const int c = cid - 1;
			const int src = (int)(wk[c].y >> 32), dst = (int)wk[c].x;
			aout[v] = b[src + (v - dst)];
		}
	}
	__syncthreads();
	for (int c = lane; c < n_u; c += WAVE) u[c] = u2[c];
	if (lane == 0) { an.n_u[r] = n_u; an.n_v[r] = n_v; }
	KPROF(kp_base + 3);
}


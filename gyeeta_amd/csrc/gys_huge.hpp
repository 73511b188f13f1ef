// Huge keys -- more than GYS_MERGE_LDS_MAX (16 384) values of ONE service in one ingest call (a heavy hitter of a Zipf stream, or any key
// of a long single-host replay: BASELINE configs 0 and 4) -- with SEVERAL workgroups per key.
//
// k_digest_huge (gys_kernels.hpp) gives such a key one workgroup and a 2^20-bin count array in HBM: 8 MB of scratch traffic and ~120 us
// per key, at most 64 keys at a time (C1: 14.6 ms for 100 keys).  Here the key's run of values (written to `staged` by the spill pass) is
// cut into chunks of 16 384 values and the work is three small kernels sized by the device-side list length:
//   k_huge_plan   one workgroup: chunks per entry -> prefix (flat chunk index -> entry), entries beyond the pool go to the fallback list
//   k_huge_count  one 1024-thread workgroup per chunk (persistent over the flat chunk list): exact counts of the values < 16 384 in a
//                 64-KiB LDS image (integer-ms response times: all but ~10^-5 of them), the chunk's RESP_TIME_HASH bucket deltas read off
//                 the image, CONN_BITMAP bits, min / max; the image is added to the entry's 64-KiB bin array in HBM (non-zero bins only);
//                 values >= 16 384 go to a global tail list
//   k_huge_merge  one workgroup per entry (two tiers: 512 threads / 512 tail values with two workgroups per CU, then 1024 / 16 384): bins -> LDS (+ the entry's buffered words), block scan, every bin's rank interval
//                 intersected with the cluster rank intervals (the exact-integer assignment of k_digest_huge), tail values ranked among
//                 themselves, records folded, clusters written back
// The result is bit-identical to k_digest_merge / k_digest_huge (same definition, DESIGN.md "t-digest").  The values >= 16 384 of an entry (up to 16 384 of them) are
// bitonic-sorted in LDS.  An entry with more of them, entries beyond the pool, or a full global tail list fall back to k_digest_huge.
#pragma once

namespace gys {

#define GYS_HB_BINS 16384u      // exact one-value bins of the LDS image / of an entry's bin array
#ifndef GYS_HB_CHUNK
#define GYS_HB_CHUNK 131072u    // values per chunk (round 3: 16 384 before -- every chunk ends with one device atomic per non-zero bin of its image, ~1500 per chunk
#endif                          // whatever its size: the C5 window paid 5 x 10^7 of them, 4.8 ms; eight times the values per chunk, an eighth of the adds)
#define GYS_HB_TAIL_LDS 16384u  // tail values (>= GYS_HB_BINS) one entry may carry on this path (sorted in LDS): tier B of k_huge_merge
#define GYS_HB_TAIL_A 512u      // ... tier A (two workgroups per CU)
#ifndef GYS_HB_VEC
#define GYS_HB_VEC 2u           // k_huge_count: 16-byte pieces a thread requests before it takes their values (0: the one-word loop of rounds 3 - 5).  r6ao - r6ar: C1 k_huge_count 216 -> 116 us, C5 1040 -> 684 us at 2 (4: the same, with scratch; 4 / 8 compiled for 4 waves per SIMD, one workgroup per CU: slower)
#endif
#ifndef GYS_HB_PIPE
#define GYS_HB_PIPE 0            // (r6av: no difference -- C1 digest_huge 0.301 - 0.311 against 0.300 - 0.304 ms; left off) k_huge_count: the next step's pieces are requested before this step's values are taken (register double buffer)
#endif
#ifndef GYS_HB_STAGED
#define GYS_HB_STAGED 0          // (r6aq: no difference -- C1 digest_huge 0.306 / 0.307 against 0.302 / 0.306 ms, 14 spilled registers at 8 values; left off) k_huge_count: the LDS reads of a thread's 4 GYS_HB_VEC values are issued together, stage by stage
#endif
#ifndef GYS_HB_LUT
#define GYS_HB_LUT 1             // k_huge_count: a value's RESP_TIME_HASH bucket (for its CONN_BITMAP bit) through the 1-KiB LDS table of k_resp_host (r6ap: C1 digest_huge 0.312 - 0.314 -> 0.303 - 0.306 ms)
#endif
#ifndef GYS_HB_WAVES
#define GYS_HB_WAVES 8            // waves per SIMD k_huge_count is compiled for
#endif
#define GYS_HB_ACC 40u          // per entry: u64 [0..15] bucket counts, [16..31] bucket sums, [32] min | max << 32 (as biased u32), [33] tail values of the entry in the global list, [34..] spare

struct Huge2P {
	DigestP d;
	const MergeEnt *list;       // huge list (finalize_key)
	const uint32_t *count;
	uint32_t *bins;             // [maxent][GYS_HB_BINS]
	unsigned long long *acc;    // [maxent][GYS_HB_ACC]
	uint32_t *bm;               // [maxent][GYS_BM_WORDS]
	uint32_t *chunk_off;        // [maxent + 1]
	unsigned long long *tail;   // entry << 32 | value
	uint32_t *tail_count;
	uint32_t tail_cap, maxent;
	uint32_t first;             // this launch handles entries [first, first + maxent) of the list (the pool is reused round by round)
	MergeEnt *fb_list;          // fallback entries for k_digest_huge
	uint32_t *fb_count;
	uint32_t *nent_used;        // entries this path handles ( = min(count, maxent), 0 when the tail list overflowed)
	uint32_t *tb_list;          // [maxent] pool entries k_huge_merge's tier A hands to tier B (more tail values than tier A's LDS list takes)
	uint32_t *tb_count;
#ifdef GYS_HUGE_TIMING // EXPERIMENT builds only: shader-clock ticks per phase of k_huge_merge, summed over entries (thread 0 of a workgroup)
	unsigned long long *dbg;
#endif
};

// ---- plan + clear: chunk prefix over the entries this path takes; the rest go to the fallback list
__global__ __launch_bounds__(1024) void k_huge_plan(Huge2P p)
{
	__shared__ uint32_t s_w[16];
	const uint32_t n = *p.count, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t nuse = n > p.first ? min(n - p.first, p.maxent) : 0u;
	uint32_t run = 0;
	for (uint32_t base = 0; base < nuse; base += 1024u) {
		const uint32_t e = base + tid;
		const uint32_t c = e < nuse ? (p.list[p.first + e].mrun + GYS_HB_CHUNK - 1u) / GYS_HB_CHUNK : 0u;
		uint32_t inc = c;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = __shfl_up(inc, d, 64);
			if ((int)lane >= d) inc += t;
		}
		if (lane == 63u) s_w[wave] = inc;
		__syncthreads();
		uint32_t wo = 0, tot = 0;
		for (uint32_t k = 0; k < 16u; ++k) {
			if (k < wave) wo += s_w[k];
			tot += s_w[k];
		}
		if (e < nuse) p.chunk_off[e] = run + wo + inc - c;
		run += tot;
		__syncthreads();
	}
	if (tid == 0) {
		p.chunk_off[nuse] = run;
		*p.nent_used = nuse;
		*p.tail_count = 0;
		*p.tb_count = 0;
	}
}

__global__ __launch_bounds__(256) void k_huge_clear(Huge2P p)
{
	const uint32_t n = *p.count;
	const uint32_t nuse = n > p.first ? min(n - p.first, p.maxent) : 0u;
	const uint64_t nb = (uint64_t)nuse * (GYS_HB_BINS / 4u), stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) ((uint4 *)p.bins)[i] = make_uint4(0, 0, 0, 0);
	const uint64_t na = (uint64_t)nuse * GYS_HB_ACC;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += stride)
		p.acc[i] = (i % GYS_HB_ACC) == 32u ? (0xFFFFFFFFull | (0ull << 32)) : 0ull; // min = +inf, max = 0 (values are >= 0)
	const uint64_t nm = (uint64_t)nuse * GYS_BM_WORDS;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += stride) p.bm[i] = 0;
}

// ---- count: one chunk of one entry's run per workgroup
__global__ __launch_bounds__(1024, GYS_HB_WAVES) void k_huge_count(Huge2P p) // (8 waves per SIMD = 64 VGPRs: two workgroups per CU, as the 64-KiB image allows)
{
	GYS_DYN_LDS(uint32_t, s_img); // [GYS_HB_BINS]
	__shared__ unsigned long long s_hc[16], s_hs[16];
	__shared__ uint32_t s_bm[GYS_BM_WORDS], s_mm[2];
	__shared__ uint32_t s_bk[GYS_BUCKET_LUT ? 256 : 1]; // RESP_TIME_HASH bucket of a value below 1024: one byte read instead of 13 compare + add pairs per value
	const uint32_t nuse = *p.nent_used, tid = threadIdx.x;
	if (!nuse) return;
	if (GYS_BUCKET_LUT) resp_bucket_lut_init(s_bk, tid, 1024u); // (read behind the first chunk's barrier)
	const uint32_t nchunks = p.chunk_off[nuse];
	for (uint32_t ck = blockIdx.x; ck < nchunks; ck += gridDim.x) {
		uint32_t lo = 0, hi = nuse - 1; // entry of the flat chunk index: largest e with chunk_off[e] <= ck
		while (lo < hi) {
			const uint32_t mid = (lo + hi + 1) >> 1;
			if (p.chunk_off[mid] <= ck) lo = mid; else hi = mid - 1;
		}
		const uint32_t e = lo;
		const MergeEnt ent = p.list[p.first + e];
		const uint32_t c = ck - p.chunk_off[e];
		const uint32_t v0 = c * GYS_HB_CHUNK, v1 = min(ent.mrun, v0 + GYS_HB_CHUNK);
		const uint32_t *run = p.d.staged + (ent.off_end - ent.mrun);
		for (uint32_t i = tid; i < GYS_HB_BINS; i += 1024u) s_img[i] = 0;
		if (tid < 16u) {
			s_hc[tid] = 0;
			s_hs[tid] = 0;
			s_bm[tid] = 0;
			s_bm[tid + 16u] = 0;
		}
		if (tid == 0) {
			s_mm[0] = 0xFFFFFFFFu;
			s_mm[1] = 0;
		}
		__syncthreads();
		uint32_t lmin = 0xFFFFFFFFu, lmax = 0;
		// (every step below commutes -- LDS / device adds, ors, min, max; the tail list is sorted by k_huge_merge -- so the order in which a
		// chunk's values are taken is free)
		auto take_tail = [&](const uint32_t v, const uint32_t b) { // bucket 14 (>= 15 001): its deltas directly; the value itself to the tail list
			atomicAdd(&s_hc[b], 1ull);
			atomicAdd(&s_hs[b], (unsigned long long)v);
			const uint32_t at = atomicAdd(p.tail_count, 1u);
			if (at < p.tail_cap) p.tail[at] = ((unsigned long long)e << 32) | v;
			atomicAdd(&p.acc[(size_t)e * GYS_HB_ACC + 33u], 1ull); // the entry's own count of tail values: an entry without any skips the list
		};
		auto take = [&](const uint32_t word) {
			const uint32_t v = word >> GYS_ROW_BITS, row = word & GYS_ROW_MASK;
			lmin = min(lmin, v);
			lmax = max(lmax, v);
			const uint32_t b = GYS_HB_LUT ? resp_bucket_lut(s_bk, v) : resp_bucket((int64_t)v);
			const uint32_t bit = (1u << b) << ((row & 1u) * 16u); // CONN_BITMAP::add_response (common/gy_socket_stat.h:403-410): every run value is of the open window
			if ((s_bm[row >> 1] & bit) == 0) atomicOr(&s_bm[row >> 1], bit);
			if (v < GYS_HB_BINS) atomicAdd(&s_img[v], 1u);
			else take_tail(v, b);
		};
#if GYS_HB_VEC
		{
			// Round 6: the plain loop (one 4-byte load per thread and step, the next one issued behind the step's atomics) kept ONE 256-byte
			// request per wave in flight: 8 KB per CU, 1.2 TB/s (C1: 216 us for 268 MB).  Here a thread asks for GYS_HB_VEC 16-byte pieces
			// before it takes their values: 8 x as many bytes in flight per wave at GYS_HB_VEC = 2 (C1: 116 us, 2.3 TB/s).
			const uint32_t *cb = run + v0;
			const uint32_t n = v1 - v0;
			const uint32_t head = min(n, (4u - ((uint32_t)((uintptr_t)cb >> 2) & 3u)) & 3u); // words in front of the first 16-byte boundary
			if (tid < head) take(cb[tid]);
			const uint4 *cv = (const uint4 *)(cb + head);
			const uint32_t nvec = (n - head) >> 2;
#if GYS_HB_PIPE
			uint4 nx[GYS_HB_VEC]; // the NEXT step's pieces, requested before this step's values are taken
#pragma unroll
			for (uint32_t u = 0; u < GYS_HB_VEC; ++u) nx[u] = u * 1024u + tid < nvec ? cv[u * 1024u + tid] : make_uint4(0, 0, 0, 0);
#endif
			for (uint32_t j0 = 0; j0 < nvec; j0 += GYS_HB_VEC * 1024u) {
				uint4 w[GYS_HB_VEC];
#if GYS_HB_PIPE
#pragma unroll
				for (uint32_t u = 0; u < GYS_HB_VEC; ++u) w[u] = nx[u];
				if (j0 + GYS_HB_VEC * 1024u < nvec) {
#pragma unroll
					for (uint32_t u = 0; u < GYS_HB_VEC; ++u) {
						const uint32_t j = j0 + (GYS_HB_VEC + u) * 1024u + tid;
						nx[u] = j < nvec ? cv[j] : make_uint4(0, 0, 0, 0);
					}
				}
#else
#pragma unroll
				for (uint32_t u = 0; u < GYS_HB_VEC; ++u) {
					const uint32_t j = j0 + u * 1024u + tid;
					w[u] = j < nvec ? cv[j] : make_uint4(0, 0, 0, 0);
				}
#endif
#if GYS_HB_STAGED
				// the values of the thread's pieces stage by stage (bucket reads of all of them, then the bitmap words, then the adds): the LDS
				// round trips of one value's chain -- table byte -> bitmap word -> test -- overlap the other values' instead of following them
				constexpr uint32_t NW = 4u * GYS_HB_VEC;
				uint32_t ww[NW], bb[NW], cur[NW];
				bool ok[NW];
#pragma unroll
				for (uint32_t u = 0; u < GYS_HB_VEC; ++u) {
					const bool o = j0 + u * 1024u + tid < nvec;
					ww[4u * u] = w[u].x; ww[4u * u + 1u] = w[u].y; ww[4u * u + 2u] = w[u].z; ww[4u * u + 3u] = w[u].w;
					ok[4u * u] = ok[4u * u + 1u] = ok[4u * u + 2u] = ok[4u * u + 3u] = o;
				}
#pragma unroll
				for (uint32_t k = 0; k < NW; ++k) bb[k] = GYS_HB_LUT ? resp_bucket_lut(s_bk, ww[k] >> GYS_ROW_BITS) : resp_bucket((int64_t)(ww[k] >> GYS_ROW_BITS));
#pragma unroll
				for (uint32_t k = 0; k < NW; ++k) cur[k] = s_bm[(ww[k] & GYS_ROW_MASK) >> 1];
				GYS_MEM_FENCE();
#pragma unroll
				for (uint32_t k = 0; k < NW; ++k) {
					if (!ok[k]) continue;
					const uint32_t v = ww[k] >> GYS_ROW_BITS, row = ww[k] & GYS_ROW_MASK;
					lmin = min(lmin, v);
					lmax = max(lmax, v);
					const uint32_t bit = (1u << bb[k]) << ((row & 1u) * 16u);
					if ((cur[k] & bit) == 0) atomicOr(&s_bm[row >> 1], bit);
					if (v < GYS_HB_BINS) atomicAdd(&s_img[v], 1u);
					else take_tail(v, bb[k]);
				}
#else
#pragma unroll
				for (uint32_t u = 0; u < GYS_HB_VEC; ++u) {
					if (j0 + u * 1024u + tid < nvec) {
						take(w[u].x);
						take(w[u].y);
						take(w[u].z);
						take(w[u].w);
					}
				}
#endif
			}
			const uint32_t t0 = head + 4u * nvec;
			if (t0 + tid < n) take(cb[t0 + tid]); // (at most three words behind the last whole piece)
		}
#else
		for (uint32_t i = v0 + tid; i < v1; i += 1024u) take(run[i]);
#endif
		lmin = wave_min_u32(lmin);
		lmax = wave_max_u32(lmax);
		if ((tid & 63u) == 0) {
			atomicMin(&s_mm[0], lmin);
			atomicMax(&s_mm[1], lmax);
		}
		__syncthreads();
		// the image: into the entry's bins (non-zero only) and, bin by bin, into the bucket deltas (GY_HISTOGRAM::add_data for cnt values
		// equal to the bin); a thread's 16 consecutive bins touch at most three buckets
		{
			uint32_t *gb = p.bins + (size_t)e * GYS_HB_BINS;
			const bool only = p.chunk_off[e + 1u] - p.chunk_off[e] == 1u; // the entry's only chunk: its (cleared) bins are written, not added to
			uint32_t curb = 0xFFu;
			unsigned long long ac = 0, as = 0;
			for (uint32_t k = 0; k < 16u; ++k) {
				const uint32_t bin = tid * 16u + k, cnt = s_img[bin];
				if (!cnt) continue;
				if (only) gb[bin] = cnt;
				else atomicAdd(&gb[bin], cnt);
				const uint32_t b = resp_bucket((int64_t)bin);
				if (b != curb) {
					if (ac) {
						atomicAdd(&s_hc[curb], ac);
						atomicAdd(&s_hs[curb], as);
					}
					curb = b;
					ac = 0;
					as = 0;
				}
				ac += cnt;
				as += (unsigned long long)cnt * bin;
			}
			if (ac) {
				atomicAdd(&s_hc[curb], ac);
				atomicAdd(&s_hs[curb], as);
			}
		}
		__syncthreads();
		unsigned long long *ga = p.acc + (size_t)e * GYS_HB_ACC;
		if (tid < 16u) {
			if (s_hc[tid]) {
				atomicAdd(&ga[tid], s_hc[tid]);
				atomicAdd(&ga[16u + tid], s_hs[tid]);
			}
			if (s_bm[tid]) atomicOr(&p.bm[(size_t)e * GYS_BM_WORDS + tid], s_bm[tid]);
			if (s_bm[tid + 16u]) atomicOr(&p.bm[(size_t)e * GYS_BM_WORDS + 16u + tid], s_bm[tid + 16u]);
		}
		if (tid == 16u && s_mm[0] != 0xFFFFFFFFu) {
			uint32_t *mm = (uint32_t *)&ga[32];
			atomicMin(&mm[0], s_mm[0]);
			atomicMax(&mm[1], s_mm[1]);
		}
		__syncthreads();
	}
}

// ---- merge: one workgroup per entry.  Two tiers (VERDICT r2 "tier the huge path"): a merge is a chain of dependent steps (entry -> meta /
// clusters / bins -> LDS passes -> write-back), so ONE resident workgroup per CU leaves the CU idle through every global-memory latency
// of the chain.  Tier A -- 512 threads, room for 512 tail values: 79 KiB of LDS, TWO workgroups per CU -- takes every entry; the rare
// entry with more values >= 16 384 than that is handed to tier B (1024 threads, 16 384 tail values, the whole LDS of a CU) through
// tb_list, and only what overflows THAT goes to the one-workgroup-per-key fallback (k_digest_huge).
template <uint32_t NT, uint32_t TAIL, bool FROM_LIST>
__global__ __launch_bounds__(NT) void k_huge_merge(Huge2P p_arg)
{
	GYS_KERNARG_REF(Huge2P, p, p_arg);
	GYS_DYN_LDS(uint32_t, s_img);                 // [GYS_HB_BINS] the entry's exact value counts (run + buffered words), then s_tail
	uint32_t *s_tail = s_img + GYS_HB_BINS;       // [TAIL] the entry's values >= GYS_HB_BINS, sorted
	__shared__ int64_t s_csum[GYS_TD_NB];
	__shared__ uint32_t s_ccnt[GYS_TD_NB];
	__shared__ uint64_t s_cpfx[GYS_TD_NB + 1];
	__shared__ uint32_t s_cthr[GYS_TD_NB]; // compacted old cluster c has mean <= v  <=>  v >= s_cthr[c] = ceil(sum / count) (0 when the sum is not positive): the searches compare one word
	__shared__ uint64_t s_T[GYS_TD_NB + 1];
	__shared__ unsigned long long s_osum[GYS_TD_NB], s_ocnt[GYS_TD_NB];
	__shared__ uint32_t s_w[NT / 64];
	__shared__ uint64_t s_cw[4];
	__shared__ unsigned long long s_ha[32], s_hw[32]; // buffered words: exact {count, sum} per bucket, all not yet folded / window part
	__shared__ unsigned long long s_pa[NT / 64][16], s_pw[NT / 64][16]; // ... accumulated per wave first (packed count : 24 | sum : 40)
	__shared__ uint32_t s_bm[GYS_BM_WORDS];
	__shared__ uint32_t s_nc, s_ntail, s_over;
	__shared__ int32_t s_min, s_max, s_wmax;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t nuse = FROM_LIST ? *p.tb_count : *p.nent_used;
	const bool tail_lost = *p.tail_count > p.tail_cap; // the global tail list overflowed: every entry goes to the fallback
	for (uint32_t ei = blockIdx.x; ei < nuse; ei += gridDim.x) {
		GYS_KERNARG_RELOAD(p);
		const uint32_t e = FROM_LIST ? p.tb_list[ei] : ei; // (tier B: the pool entries tier A handed over)
		const MergeEnt ent = p.list[p.first + e];
		if (tail_lost && FROM_LIST) continue; // (tier A has sent them to the fallback already)
		if (tail_lost) {
			if (tid == 0) p.fb_list[atomicAdd(p.fb_count, 1u)] = ent;
			continue;
		}
#ifdef GYS_HUGE_TIMING
		unsigned long long tk[6];
		tk[0] = __builtin_amdgcn_s_memtime();
#define GYS_HT(k) do { if (tid == 0) tk[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GYS_HT(k) do { } while (0)
#endif
		const uint32_t slot = ent.slot, m = ent.mrun, npend = ent.nbuf;
		const uint4 mt = *(const uint4 *)&p.d.td_meta[slot];
		const uint32_t nh = mt.y & 0xFFFFu, nw = mt.y >> 16, nwin0 = max(nh, nw);
		const uint32_t *pend = p.d.td_pend + (size_t)slot * p.d.pcap;
		const uint32_t *gb = p.bins + (size_t)e * GYS_HB_BINS;
		for (uint32_t i = tid; i < GYS_HB_BINS / 4u; i += NT) ((uint4 *)s_img)[i] = ((const uint4 *)gb)[i];
		if (tid < GYS_TD_NB) {
			s_osum[tid] = 0;
			s_ocnt[tid] = 0;
		}
		if (tid < 32u) {
			s_ha[tid] = 0;
			s_hw[tid] = 0;
		}
		if (tid >= 32u && tid < 32u + GYS_BM_WORDS) s_bm[tid - 32u] = 0;
		{ // compact the non-empty old clusters (order preserving) + exclusive prefix of their weights: threads 0..255, one cluster each
			uint32_t c0 = 0;
			int64_t sm0 = 0;
			if (tid < GYS_TD_NB) {
				c0 = p.d.td_cnt[(size_t)slot * GYS_TD_NB + tid];
				sm0 = p.d.td_sum[(size_t)slot * GYS_TD_NB + tid];
			}
			const unsigned long long b0 = __ballot(c0 != 0);
			const uint64_t inc64 = wave_incl_scan_u64(c0);
			if (tid < 256u) {
				if (lane == 63u) s_cw[wave] = inc64;
				if (lane == 0u) s_w[wave] = (uint32_t)__popcll(b0);
			}
			__syncthreads();
			if (tid < 256u) {
				uint32_t pb = 0, ncl = 0;
				uint64_t wb = 0, tot = 0;
				for (uint32_t k = 0; k < 4u; ++k) {
					if (k < wave) {
						pb += s_w[k];
						wb += s_cw[k];
					}
					ncl += s_w[k];
					tot += s_cw[k];
				}
				if (c0) {
					const uint32_t pos = pb + (uint32_t)__popcll(b0 & (lane ? (~0ull >> (64 - lane)) : 0ull));
					s_csum[pos] = sm0;
					s_ccnt[pos] = c0;
					s_cpfx[pos] = wb + inc64 - c0;
					s_cthr[pos] = sm0 <= 0 ? 0u : (uint32_t)min((uint64_t)0xFFFFFFFFull, ((uint64_t)sm0 + c0 - 1u) / c0);
				}
				if (tid == 0) {
					s_cpfx[ncl] = tot;
					s_nc = ncl;
				}
			}
			if (tid == 0) {
				s_ntail = 0;
				s_over = 0;
				s_min = INT32_MAX;
				s_max = INT32_MIN;
				s_wmax = INT32_MIN;
			}
		}
		__syncthreads();
		GYS_HT(1);
		// the buffered words join the counts; their not yet folded part is folded here (the run's deltas come from k_huge_count).
		// A large key's buffer holds up to 16 384 words and nearly all of them land in a handful of histogram buckets: per-value
		// atomics on ONE set of bucket accumulators / one min / one max serialise the whole workgroup on a few LDS addresses (136 us per
		// key measured on the C5 shape).  So: packed {count : 24 | sum : 40} adds into per-WAVE rows (a wave's 1024 values x 10^6 fit
		// the sum field), the extremes in registers with one reduction per wave, bitmap bits tested before they are set.
		{
			unsigned long long *pa = s_pa[wave], *pw = s_pw[wave];
			if (lane < 16u) {
				pa[lane] = 0;
				pw[lane] = 0;
			}
			GYS_WAVE_SYNC();
			int32_t lmin = INT32_MAX, lmax = INT32_MIN, wmx = INT32_MIN;
			// (eight loads in flight per thread: with one load per iteration in front of the LDS atomics every iteration waited out a whole
			// global-memory round trip -- a quarter of a large key's merge at ~5 500 buffered words per key, r4z)
			for (uint32_t i0 = 0; i0 < npend; i0 += 8u * NT) {
				uint32_t wv[8];
#pragma unroll
				for (uint32_t u = 0; u < 8u; ++u) {
					const uint32_t i = i0 + u * NT + tid;
					wv[u] = i < npend ? pend[i] : 0u;
				}
				// class of a word: 0 nothing to fold (folded before, or past the end), 1 not yet folded but of an earlier window (row pa),
				// 2 of the open window (row pw; the fold of ALL not yet folded words is pa + pw).  The thread's eight words are combined per
				// (class, bucket) in registers first: a large key's words sit in two or three buckets, and per-word adds lined the whole
				// workgroup up on those few LDS addresses (same-address LDS atomics run a lane at a time: a quarter of the merge, r4z / r4ab)
				uint32_t key[8];
				unsigned long long one[8];
#pragma unroll
				for (uint32_t u = 0; u < 8u; ++u) {
					const uint32_t i = i0 + u * NT + tid;
					key[u] = 0xFFu;
					one[u] = 0;
					if (i >= npend) continue;
					const uint32_t word = wv[u], v = word >> GYS_ROW_BITS;
					if (v < GYS_HB_BINS) {
						atomicAdd(&s_img[v], 1u);
					} else {
						const uint32_t at = atomicAdd(&s_ntail, 1u);
						if (at < TAIL) s_tail[at] = v; else s_over = 1;
					}
					if (i >= nh) {
						const uint32_t hb = resp_bucket((int64_t)v);
						one[u] = GYS_PACK_ONE | (unsigned long long)v;
						lmin = min(lmin, (int32_t)v);
						lmax = max(lmax, (int32_t)v);
						key[u] = hb;
						if (i >= nwin0) {
							key[u] = hb | 16u;
							const uint32_t row = word & GYS_ROW_MASK, bit = (1u << hb) << ((row & 1u) * 16u);
							if ((s_bm[row >> 1] & bit) == 0u) atomicOr(&s_bm[row >> 1], bit);
							wmx = max(wmx, (int32_t)v);
						}
					}
				}
#pragma unroll
				for (uint32_t u = 0; u < 8u; ++u) {
					if (key[u] == 0xFFu) continue;
					unsigned long long acc = one[u];
#pragma unroll
					for (uint32_t w = u + 1u; w < 8u; ++w)
						if (key[w] == key[u]) {
							acc += one[w];
							key[w] = 0xFFu;
						}
					atomicAdd((key[u] & 16u) ? &pw[key[u] & 15u] : &pa[key[u]], acc);
				}
			}
			lmin = wave_min_i32(lmin);
			lmax = wave_max_i32(lmax);
			wmx = wave_max_i32(wmx);
			if (lane == 0u) {
				if (lmin != INT32_MAX) atomicMin(&s_min, lmin);
				if (lmax != INT32_MIN) atomicMax(&s_max, lmax);
				if (wmx != INT32_MIN) atomicMax(&s_wmax, wmx);
			}
			GYS_WAVE_SYNC();
			if (lane < 16u) { // the wave's row into the workgroup's {count, sum} pairs
				const unsigned long long w = pw[lane], a = pa[lane] + w; // (all not yet folded = of earlier windows + of the open one)
				if (a) {
					atomicAdd(&s_ha[2 * lane], a >> 40);
					atomicAdd(&s_ha[2 * lane + 1], GYS_PACK_SUM(a));
				}
				if (w) {
					atomicAdd(&s_hw[2 * lane], w >> 40);
					atomicAdd(&s_hw[2 * lane + 1], GYS_PACK_SUM(w));
				}
			}
		}
		// this entry's tail values of the run (round 3: only an entry that HAS values in the global list walks it -- every one of the ~10^4
		// entries of a Zipf window used to read the whole list, tens of thousands of words, to find none of its own)
		if (p.acc[(size_t)e * GYS_HB_ACC + 33u]) {
			const uint32_t nt = min(*p.tail_count, p.tail_cap);
			for (uint32_t i = tid; i < nt; i += NT) {
				const unsigned long long t = p.tail[i];
				if ((uint32_t)(t >> 32) != e) continue;
				const uint32_t at = atomicAdd(&s_ntail, 1u);
				if (at < TAIL) s_tail[at] = (uint32_t)t; else s_over = 1;
			}
		}
		__syncthreads();
		GYS_HT(2);
		if (s_over) { // too many large values for this tier's LDS list (nothing has been modified): the next tier takes the entry
			if (tid == 0) {
				if (!FROM_LIST && TAIL < GYS_HB_TAIL_LDS) p.tb_list[atomicAdd(p.tb_count, 1u)] = e;
				else p.fb_list[atomicAdd(p.fb_count, 1u)] = ent;
			}
			__syncthreads();
			continue;
		}
		const uint32_t nc = s_nc, ntail = s_ntail;
		// tier B (up to 16 384 tail values): bitonic sort of the tail in LDS; tier A (up to 512): no sort at all -- a value's rank among the
		// tail values and a cluster's count of tail values below its mean are counted by reading the short list (every lane reads the same
		// word: a broadcast), which takes one barrier instead of the up to 45 of the sorting network
		constexpr bool SORT_TAIL = TAIL > 1024u;
		if (SORT_TAIL && ntail > 1u) { // (padded to a power of two with +inf); equal values are interchangeable
			uint32_t n2 = 2;
			while (n2 < ntail) n2 <<= 1;
			for (uint32_t i = ntail + tid; i < n2; i += NT) s_tail[i] = 0xFFFFFFFFu;
			__syncthreads();
			for (uint32_t k = 2; k <= n2; k <<= 1) {
				for (uint32_t j = k >> 1; j > 0; j >>= 1) {
					for (uint32_t i = tid; i < n2; i += NT) {
						const uint32_t ixj = i ^ j;
						if (ixj > i) {
							const uint32_t a = s_tail[i], b = s_tail[ixj];
							if ((a > b) == ((i & k) == 0u)) {
								s_tail[i] = b;
								s_tail[ixj] = a;
							}
						}
					}
					__syncthreads();
				}
			}
		}
		const uint64_t nold = s_cpfx[nc];
		const uint64_t twoN = 2ull * (nold + (uint64_t)m + (uint64_t)npend);
		if (tid >= 1u && tid < GYS_TD_NB) s_T[tid] = td_threshold(c_td_bnd[tid], twoN);
		if (tid == 0) s_T[GYS_TD_NB] = ~0ull;
		// ---- the counts become exclusive prefixes IN PLACE (count of bin b = next prefix - this one).  Every wave takes a contiguous
		// sixteenth / eighth of the bins 64 at a time -- lanes side by side, a wave scan, the carry in a scalar -- and adds the waves
		// before it in a second sweep (four bins per lane and step).  (Round 3 gave every THREAD 32 consecutive bins: a stride of 32 words puts the 64 lanes of a wave
		// on two LDS banks, and the three sweeps of that layout were a third of a large key's merge, r4z.)
		uint32_t nlow = 0;
		{
			constexpr uint32_t NW = NT / 64u, PER_WAVE = GYS_HB_BINS / NW;
			const uint32_t wb = wave * PER_WAVE;
			uint32_t carry = 0;
			for (uint32_t r = 0; r < PER_WAVE; r += 256u) { // four consecutive bins per lane: a quarter of the wave scans
				uint4 *q4 = (uint4 *)(s_img + wb + r + lane * 4u);
				const uint4 c4 = *q4;
				const uint32_t c = c4.x + c4.y + c4.z + c4.w;
				const uint32_t inc = wave_incl_scan_u32(c);
				const uint32_t e0 = carry + inc - c;
				*q4 = make_uint4(e0, e0 + c4.x, e0 + c4.x + c4.y, e0 + c4.x + c4.y + c4.z);
				carry += wave_last_u32(inc);
			}
			if (lane == 0u) s_w[wave] = carry;
			__syncthreads();
			uint32_t woff = 0;
			for (uint32_t k = 0; k < NW; ++k) {
				if (k < wave) woff += s_w[k];
				nlow += s_w[k];
			}
			if (woff)
				for (uint32_t r = 0; r < PER_WAVE; r += 256u) {
					uint4 *q4 = (uint4 *)(s_img + wb + r + lane * 4u);
					const uint4 c4 = *q4;
					*q4 = make_uint4(c4.x + woff, c4.y + woff, c4.z + woff, c4.w + woff);
				}
		}
		__syncthreads();
		GYS_HT(3);
		// ---- old clusters: lt = #{values v : v * cc < cs} = #{v <= (cs - 1) / cc}  (cs >= 1; none when cs <= 0)
		if (tid < nc) {
			const int64_t cs = s_csum[tid];
			const uint32_t cc = s_ccnt[tid];
			uint64_t lt = 0;
			if (cs > 0) {
				const int64_t vmax = (cs - 1) / (int64_t)cc;
				if (vmax >= (int64_t)GYS_HB_BINS) {
					uint32_t lo = 0; // tail values <= vmax
					if (SORT_TAIL) {
						uint32_t hi = ntail;
						while (lo < hi) {
							const uint32_t mid = (lo + hi) >> 1;
							if ((int64_t)s_tail[mid] <= vmax) lo = mid + 1; else hi = mid;
						}
					} else {
						for (uint32_t i = 0; i < ntail; ++i) lo += (int64_t)s_tail[i] <= vmax ? 1u : 0u;
					}
					lt = (uint64_t)nlow + lo;
				} else {
					lt = (uint32_t)vmax + 1u < GYS_HB_BINS ? s_img[(uint32_t)vmax + 1u] : nlow; // values <= vmax = prefix of the next bin
				}
			}
			const uint64_t mid2 = 2ull * (s_cpfx[tid] + lt) + (uint64_t)cc;
			const uint32_t cl = td_cluster_of(s_T, mid2);
			atomicAdd(&s_osum[cl], (unsigned long long)cs);
			atomicAdd(&s_ocnt[cl], (unsigned long long)cc);
		}
		// ---- values bin by bin (thread t: bins t, t + NT, ...): ranks [r0, r0 + c) of value v, le = old weight with mean <= v.
		// Two bins per thread and round, the three searches of either written as eight fixed steps (no data-dependent branch): a search is
		// a chain of dependent LDS reads, and with one bin per round and `continue` on an empty one the 32 rounds of a large key's long
		// sparse tail of bins each cost a full chain for a few live lanes (the pass was 36 % of the merge, r4z) -- now the two chains of a
		// round overlap and there are 16 rounds.
		for (uint32_t b0 = tid; b0 < GYS_HB_BINS; b0 += 2u * NT) {
			uint32_t bb[2], r0b[2], cc[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				bb[u] = b0 + (uint32_t)u * NT;
				r0b[u] = s_img[bb[u]];
				cc[u] = (bb[u] + 1u < GYS_HB_BINS ? s_img[bb[u] + 1u] : nlow) - r0b[u];
			}
			if (!(cc[0] | cc[1])) continue;
			uint32_t ci[2] = {0u, 0u}; // first compacted cluster with mean > v
#pragma unroll
			for (uint32_t step = 128u; step; step >>= 1) {
#pragma unroll
				for (int u = 0; u < 2; ++u) {
					const uint32_t np = ci[u] + step;
					if (np <= nc && s_cthr[np - 1u] <= bb[u]) ci[u] = np;
				}
			}
			uint64_t le[2], first[2], last[2];
			uint32_t cl[2] = {0u, 0u}, cl_last[2] = {0u, 0u};
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				le[u] = s_cpfx[ci[u]];
				first[u] = 2ull * ((uint64_t)r0b[u] + le[u]) + 1ull;
				last[u] = first[u] + 2ull * (uint64_t)(cc[u] ? cc[u] - 1u : 0u);
			}
#pragma unroll
			for (uint32_t step = 128u; step; step >>= 1) { // td_cluster_of(s_T, .) for the four rank ends at once
#pragma unroll
				for (int u = 0; u < 2; ++u) {
					const uint32_t n1 = cl[u] + step, n2 = cl_last[u] + step;
					if (n1 <= GYS_TD_NB - 1u && first[u] >= s_T[n1]) cl[u] = n1;
					if (n2 <= GYS_TD_NB - 1u && last[u] >= s_T[n2]) cl_last[u] = n2;
				}
			}
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				if (!cc[u]) continue;
				const uint64_t r0 = r0b[u], v = bb[u];
				uint64_t rbeg = r0;
				for (uint32_t k = cl[u]; k <= cl_last[u]; ++k) {
					uint64_t rend;
					if (k == cl_last[u]) {
						rend = r0 + cc[u];
					} else {
						const uint64_t Tn = s_T[k + 1];
						rend = (Tn / 2ull) - le[u]; // ranks r with 2 (r + le) + 1 < Tn
						if (rend > r0 + cc[u]) rend = r0 + cc[u];
					}
					if (rend > rbeg) {
						const uint64_t n = rend - rbeg;
						atomicAdd(&s_osum[k], (unsigned long long)(n * v));
						atomicAdd(&s_ocnt[k], (unsigned long long)n);
						rbeg = rend;
					}
				}
			}
		}
		// ---- the tail values: sorted, all values below 16 384 precede them
		for (uint32_t j = tid; j < ntail; j += NT) {
			const uint32_t v = s_tail[j];
			uint32_t rk = j; // position among the tail values in ascending order (ties: by place in the list -- equal values are interchangeable)
			if (!SORT_TAIL) {
				rk = 0;
				for (uint32_t i = 0; i < ntail; ++i) {
					const uint32_t u = s_tail[i];
					rk += (u < v || (u == v && i < j)) ? 1u : 0u;
				}
			}
			const uint64_t r = (uint64_t)nlow + rk;
			uint32_t lo = 0, hi = nc; // old weight with mean <= v
			while (lo < hi) {
				const uint32_t mid = (lo + hi) >> 1;
				if (s_cthr[mid] <= v) lo = mid + 1; else hi = mid;
			}
			const uint64_t mid2 = 2ull * (r + s_cpfx[lo]) + 1ull;
			const uint32_t cl = td_cluster_of(s_T, mid2);
			atomicAdd(&s_osum[cl], (unsigned long long)v);
			atomicAdd(&s_ocnt[cl], 1ull);
		}
		__syncthreads();
		GYS_HT(4);
		if (tid < GYS_TD_NB) {
			p.d.td_sum[(size_t)slot * GYS_TD_NB + tid] = (int64_t)s_osum[tid];
			p.d.td_cnt[(size_t)slot * GYS_TD_NB + tid] = (uint32_t)s_ocnt[tid];
		}
		{
			// the key's records: run deltas (k_huge_count: every run value is not yet folded and of the open window) + the buffered words'
			const unsigned long long *ga = p.acc + (size_t)e * GYS_HB_ACC;
			const uint32_t n_all = npend + m - nh, n_win = npend + m - nwin0;
			const uint32_t *rmm = (const uint32_t *)&ga[32];
			const int32_t rmin = m ? (int32_t)rmm[0] : INT32_MAX, rmax = m ? (int32_t)rmm[1] : INT32_MIN; // (no run: buffered words only)
			const int32_t amin = min(s_min, rmin), amax = max(s_max, rmax), wmaxv = max(s_wmax, rmax);
			const uint32_t t = tid - 128u;
			if (tid >= 128u && t < 16u && n_all) {
				gys_hist_serial *ap = (gys_hist_serial *)&p.d.hist_all[slot] + t, *wp = (gys_hist_serial *)&p.d.hist_win[slot] + t;
				const bool roll = mt.w != mt.z;
				gys_hist_serial av = *ap;
				if (t < 15u) {
					av.count += s_ha[2 * t] + ga[t];
					av.sum += (int64_t)(s_ha[2 * t + 1] + ga[16u + t]);
				} else {
					av.count += n_all;
					if (av.sum < (int64_t)amax) av.sum = (int64_t)amax;
				}
				*ap = av;
				if (n_win) {
					gys_hist_serial wv;
					if (roll) {
						wv.count = 0;
						wv.sum = t < 15u ? 0 : INT64_MIN;
					} else {
						wv = *wp;
					}
					if (t < 15u) {
						wv.count += s_hw[2 * t] + ga[t];
						wv.sum += (int64_t)(s_hw[2 * t + 1] + ga[16u + t]);
					} else {
						wv.count += n_win;
						if (wv.sum < (int64_t)wmaxv) wv.sum = (int64_t)wmaxv;
					}
					*wp = wv;
					uint32_t *bp = &p.d.bitmap[(size_t)slot * GYS_BM_WORDS + t];
					*bp = (roll ? 0u : *bp) | s_bm[t] | p.bm[(size_t)e * GYS_BM_WORDS + t];
					bp[16] = (roll ? 0u : bp[16]) | s_bm[t + 16u] | p.bm[(size_t)e * GYS_BM_WORDS + 16u + t];
				}
			}
			__syncthreads(); // every reader of the meta record is done before thread 0 rewrites it
			if (tid == 0) {
				*(uint4 *)&p.d.td_meta[slot] = make_uint4(0u, 0u, mt.z, n_win ? mt.z : mt.w); // buffer drained
				p.d.td_cur[slot] = 0;
				if (n_all) {
					const int2 mm = p.d.td_minmax[slot];
					p.d.td_minmax[slot] = make_int2(min(mm.x, amin), max(mm.y, amax));
				}
			}
		}
		__syncthreads();
#ifdef GYS_HUGE_TIMING
		if (tid == 0 && !FROM_LIST) {
			tk[5] = __builtin_amdgcn_s_memtime();
			for (int k = 0; k < 5; ++k) atomicAdd(&p.dbg[k], tk[k + 1] - tk[k]);
			atomicAdd(&p.dbg[5], 1ull);
			atomicAdd(&p.dbg[6], (unsigned long long)npend);
			atomicAdd(&p.dbg[7], (unsigned long long)nc);
		}
#endif
	}
}

} // namespace gys

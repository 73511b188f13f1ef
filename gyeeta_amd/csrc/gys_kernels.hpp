// gys_kernels.hpp -- the HIP kernels of libgysketch (gfx950, wave64).  Included once by gys_engine.hip.
//
// Hot path (per response event, replaces common/gy_socket_stat.cc:1517-1677 + common/gy_statistics.h:596-623):
//   resp_pass1     24-B event -> listener slot (table probe) -> RESP_TIME_HASH bucket -> per-service exact histogram,
//                  CONN_BITMAP, global histogram (LDS privatised), global HLL, Count-Min, per-key batch count, (slot,value) record
//   scan_*         exclusive scan of the per-key batch counts (counting sort by key)
//   resp_scatter   values scattered into per-key contiguous segments
//   key_pass       one wave per key with new values: exact histogram record, CONN_BITMAP, Count-Min, min/max, append to the key's
//                  t-digest buffer (or queue the key for a merge when the buffer would overflow)
//   digest_merge   queued keys: LDS bitonic sort of (buffered + new) values + exact-integer k-bucket t-digest merge
//   digest_huge    keys with > GYS_SMALL_MAX new values: value-count array in HBM scratch + parallel rank-interval assignment
// All of it is HBM-bound integer work: no MFMA.
#pragma once

#include "gys_device.hpp"

namespace gys {

#define GYS_SMALL_MAX 1024u        // largest per-key batch handled by k_key_pass / k_digest_merge; larger ones go to k_digest_huge
#define GYS_HUGE_VALUE_BITS 20     // resp values are <= 1,000,000 < 2^20 (drop filter common/gy_socket_stat.cc:1521-1524)
#define GYS_HUGE_BINS (1u << GYS_HUGE_VALUE_BITS)
// staged word of one accepted event: (response ms << 5) | CONN_BITMAP row (cli_port & 0x1F, common/gy_socket_stat.h:403-410).
// Sorting the words sorts by value; the digest kernels, which see all of a key's words, also produce the key's bitmap rows.
#define GYS_ROW_BITS 5
#define GYS_STAGED_WORD(tresp, cli_port) ((uint64_t)(((uint32_t)(tresp) << GYS_ROW_BITS) | ((uint32_t)(cli_port) & 0x1Fu)))

enum { CTR_RESP_EVENTS = 0, CTR_RESP_DROP_RANGE, CTR_RESP_DROP_NOLISTENER, CTR_CONN_EVENTS, CTR_CONN_UNKNOWN, CTR_LSTATE_RECORDS,
       CTR_LSTATE_MISSED, CTR_LSTATE_ERRORS, CTR_LSTATE_DELETED, CTR_NUM };

// ---------------------------------------------------------------------------------------------------- table insert
__global__ void k_table_insert(DevTable t, const uint64_t *keys, uint32_t first_val, uint32_t n, uint32_t *nfail)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t key = keys[i];
	uint32_t h = get_uint64_hash(key) & t.mask;
	for (uint32_t probes = 0; probes <= t.mask; ++probes) {
		const unsigned long long prev = atomicCAS((unsigned long long *)&t.ent[h].key, (unsigned long long)GYS_EMPTY_KEY, (unsigned long long)key);
		if (prev == GYS_EMPTY_KEY || prev == key) {
			t.ent[h].val = first_val + i; // re-registration of a key rebinds it to the newest slot
			return;
		}
		h = (h + 1) & t.mask;
	}
	atomicAdd(nfail, 1u);
}

__global__ void k_fill_u64(uint64_t *p, uint64_t v, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_hist_init(gys_hist_rec *h, uint64_t first, uint64_t n, int64_t minval)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) h[first + i].max_val_seen = minval;
}

__global__ void k_tdmeta_init(uint4 *meta, uint64_t n)
{
	// TdMeta {vmin = INT32_MAX, vmax = INT32_MIN, npend = 0, pad}
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		meta[i] = make_uint4((uint32_t)INT32_MAX, (uint32_t)INT32_MIN, 0u, 0u);
}

// ---------------------------------------------------------------------------------------------------- resp pass 1
__device__ __forceinline__ uint16_t bswap16(uint16_t v) { return (uint16_t)((v >> 8) | (v << 8)); }

// 64-bit sketch hash of a response event's flow: PAIR_IP_PORT(cli = daddr:dport, ser = saddr:sport) (common/gy_inet_inc.h:225-247)
__device__ __forceinline__ uint64_t flow_hash64(uint32_t daddr, uint16_t dport, uint32_t saddr, uint16_t sport)
{
	if (daddr != 0 && saddr != 0) // both ends IPv4: the key is the 4 words [cli ip][cli port][ser ip][ser port]
		return ((uint64_t)jhash2_4w(daddr, dport, saddr, sport, GYS_SEED) << 32) | (uint64_t)jhash2_4w(daddr, dport, saddr, sport, GYS_GOLDEN);
	// 0.0.0.0 hashes as 16 zero bytes (GY_IP_ADDR::get_as_inaddr quirk): general word packing
	uint32_t w[10];
	const uint32_t z[4] = {0, 0, 0, 0};
	const uint32_t nw = pair_words(daddr, z, dport, saddr, z, sport, w);
	return hash64<10>(w, nw);
}

// Register index and rank of a response event's flow in the global HLL (== hll_idx_rank(flow_hash64(...), GYS_HLL_P)).  The index
// and the first 18 rank bits come from the HIGH hash half alone; the low half (a second jhash2, ~75 instructions) can only matter
// when those 18 bits are all zero, i.e. for one event in 2^18, so it is computed on that branch only.
__device__ __forceinline__ void flow_hll_idx_rank(uint32_t daddr, uint16_t dport, uint32_t saddr, uint16_t sport, uint32_t *idx, uint32_t *rank)
{
	if (daddr != 0 && saddr != 0) {
		const uint32_t hi = jhash2_4w(daddr, dport, saddr, sport, GYS_SEED);
		const uint32_t rest = hi << GYS_HLL_P;
		*idx = hi >> (32 - GYS_HLL_P);
		if (rest) {
			*rank = (uint32_t)__clz((int)rest) + 1u;
			return;
		}
		const uint32_t lo = jhash2_4w(daddr, dport, saddr, sport, GYS_GOLDEN);
		*rank = (32u - GYS_HLL_P) + (lo ? (uint32_t)__clz((int)lo) : 32u) + 1u;
		return;
	}
	hll_idx_rank(flow_hash64(daddr, dport, saddr, sport), GYS_HLL_P, idx, rank);
}

// per-service distinct clients: same hash, per-service register file (u8 packed, CAS on the word)
__device__ __forceinline__ void svc_hll_update(uint8_t *svc_hll, uint32_t svc_hll_p, uint32_t slot, uint64_t h64)
{
	uint32_t sidx, srank;
	hll_idx_rank(h64, (int)svc_hll_p, &sidx, &srank);
	uint8_t *base = svc_hll + ((size_t)slot << svc_hll_p);
	uint32_t *wp = (uint32_t *)(base + (sidx & ~3u));
	const uint32_t sh = (sidx & 3u) * 8u;
	uint32_t old = *wp;
	while (((old >> sh) & 0xFFu) < srank) {
		const uint32_t nv = (old & ~(0xFFu << sh)) | (srank << sh);
		const uint32_t prev = atomicCAS(wp, old, nv);
		if (prev == old) break;
		old = prev;
	}
}

__device__ __forceinline__ void hll_update_event(uint32_t *hll32, uint8_t *svc_hll, uint32_t svc_hll_p, uint32_t slot, uint32_t daddr, uint16_t dport,
						 uint32_t saddr, uint16_t sport)
{
	uint32_t idx, rank;
	if (svc_hll_p) { // the per-service registers need the whole 64-bit hash
		const uint64_t h64 = flow_hash64(daddr, dport, saddr, sport);
		hll_idx_rank(h64, GYS_HLL_P, &idx, &rank);
		svc_hll_update(svc_hll, svc_hll_p, slot, h64);
	} else {
		flow_hll_idx_rank(daddr, dport, saddr, sport, &idx, &rank);
	}
	if (hll32[idx] < rank) atomicMax(&hll32[idx], rank);
}

struct RespP1 {
	const uint64_t *ev;       // 3 x u64 per event (tcp_ipv4_resp_event_t, common/gy_ebpf_kernel.h:106-111)
	uint64_t n;
	const gys_resp_seg *segs; // device copy
	uint32_t nsegs;
	DevTable lk;
	const uint64_t *svc_gid;
	gys_hist_rec *hist_win;
	uint32_t *bitmap;         // [nsvc*16] u32 = 32 x u16 CONN_BITMAP rows
	uint32_t *hll32;          // [1<<14]
	uint32_t *cms32;          // arena [D*W]
	uint32_t *batch_cnt;      // nullptr when the t-digest is off
	uint64_t *ev_kv;          // (slot << 32 | value) per event, ~0 = dropped
	uint64_t *counters;
	uint8_t *svc_hll;
	uint32_t svc_hll_p;
};

__device__ __forceinline__ uint32_t find_seg(const gys_resp_seg *segs, uint32_t nsegs, uint64_t i)
{
	uint32_t lo = 0, hi = nsegs - 1; // largest s with segs[s].first_event <= i
	while (lo < hi) {
		const uint32_t mid = (lo + hi + 1) >> 1;
		if (segs[mid].first_event <= i) lo = mid; else hi = mid - 1;
	}
	return lo;
}

__global__ __launch_bounds__(256) void k_resp_pass1(RespP1 p)
{
	__shared__ unsigned int s_ctr[3];
	if (threadIdx.x < 3) s_ctr[threadIdx.x] = 0;
	__syncthreads();
	const bool fused = p.batch_cnt != nullptr; // histogram + CMS are then produced per KEY by the digest kernels from the sorted runs

	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += stride) {
		// struct ipv4_tuple_t {u32 saddr, daddr, netns; u16 sport, dport;} + u32 lsndtime, lrcvtime  (24 bytes)
		const uint64_t w0 = p.ev[3 * i], w1 = p.ev[3 * i + 1], w2 = p.ev[3 * i + 2];
		const uint32_t saddr = (uint32_t)w0, daddr = (uint32_t)(w0 >> 32);
		const uint32_t netns = (uint32_t)w1;
		const uint16_t sport = bswap16((uint16_t)(w1 >> 32)), dport = bswap16((uint16_t)(w1 >> 48)); // ntohs :1526-1527
		const uint32_t lsnd = (uint32_t)w2, lrcv = (uint32_t)(w2 >> 32);
		const uint32_t tresp = lsnd - lrcv; // int tresp_msec = lsndtime - lrcvtime (:1519)
		uint64_t kv = ~0ull;

		atomicAdd(&s_ctr[0], 1u);
		if (tresp > 1000000u) { // "Ignore responses > 1000 sec or negative" (:1521-1524)
			atomicAdd(&s_ctr[1], 1u);
		} else {
			const uint32_t host_slot = p.segs[find_seg(p.segs, p.nsegs, i)].host_slot;
			const uint32_t slot = tbl_lookup(p.lk, listener_key(host_slot, netns, sport));
			if (slot == GYS_NOSLOT) {
				atomicAdd(&s_ctr[2], 1u); // no such listener: the reference ignores the event too (:1671-1676 miss path)
			} else {
				const uint32_t b = resp_bucket((int64_t)tresp);
				if (!fused) {
					// GY_HISTOGRAM::add_data / HIST_SERIAL::add (common/gy_statistics.h:463-467, :596-623)
					gys_hist_rec *h = &p.hist_win[slot];
					atomicAdd((unsigned long long *)&h->stats[b].count, 1ull);
					atomicAdd((unsigned long long *)&h->stats[b].sum, (unsigned long long)tresp);
					atomicAdd((unsigned long long *)&h->total_count, 1ull);
					if (h->max_val_seen < (int64_t)tresp) atomicMax((long long *)&h->max_val_seen, (long long)tresp);
					// Count-Min: events per service key (glob_id), row hash jhash2(key, seed + r)
					const uint64_t gid = p.svc_gid[slot];
#pragma unroll
					for (uint32_t r = 0; r < GYS_CMS_D; ++r)
						atomicAdd(&p.cms32[r * GYS_CMS_W + (jhash2_u64(gid, GYS_SEED + r) & (GYS_CMS_W - 1))], 1u);
				}
				// CONN_BITMAP::add_response: respmap_[cli_port & 0x1F].set(bucket) (common/gy_socket_stat.h:403-410)
				if (!fused) {
					const uint32_t row = dport & 0x1Fu;
					const uint32_t bit = (1u << b) << ((row & 1u) * 16u);
					uint32_t *wp = &p.bitmap[(size_t)slot * 16u + (row >> 1)];
					if ((*wp & bit) == 0) atomicOr(wp, bit);
				}
				hll_update_event(p.hll32, p.svc_hll, p.svc_hll_p, slot, daddr, dport, saddr, sport);
				if (fused) {
					atomicAdd(&p.batch_cnt[slot], 1u);
					kv = ((uint64_t)slot << 32) | GYS_STAGED_WORD(tresp, dport);
				}
			}
		}
		if (p.ev_kv) p.ev_kv[i] = kv;
	}
	__syncthreads();
	if (threadIdx.x < 3 && s_ctr[threadIdx.x]) atomicAdd((unsigned long long *)&p.counters[CTR_RESP_EVENTS + threadIdx.x], (unsigned long long)s_ctr[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------- scan of batch counts
#define GYS_SCAN_TILE 4096u // elements per 256-thread block (16 per thread)

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *s_wave, uint32_t *total)
{
	// inclusive wave scan by shuffles, then 4 wave totals through LDS
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t inc = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t t = __shfl_up(inc, d, 64);
		if ((int)lane >= d) inc += t;
	}
	if (lane == 63) s_wave[wave] = inc;
	__syncthreads();
	uint32_t woff = 0, tot = 0;
#pragma unroll
	for (uint32_t w = 0; w < 4; ++w) {
		const uint32_t t = s_wave[w];
		if (w < wave) woff += t;
		tot += t;
	}
	__syncthreads();
	*total = tot;
	return woff + inc - v;
}

__global__ __launch_bounds__(256) void k_scan_block_sums(const uint32_t *cnt, uint32_t n, uint32_t *block_sums)
{
	__shared__ uint32_t s_wave[4];
	const uint32_t base = blockIdx.x * GYS_SCAN_TILE + threadIdx.x * 16u;
	uint32_t s = 0;
	if (base + 16u <= n) {
		const uint4 *p4 = (const uint4 *)(cnt + base);
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint4 v = p4[k];
			s += v.x + v.y + v.z + v.w;
		}
	} else {
		for (uint32_t k = 0; k < 16u; ++k)
			if (base + k < n) s += cnt[base + k];
	}
	uint32_t total;
	(void)block_exclusive_scan_256(s, s_wave, &total);
	if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_scan_top(uint32_t *block_sums, uint32_t nblocks)
{
	__shared__ uint32_t s_wave[4];
	uint32_t carry = 0;
	for (uint32_t base = 0; base < nblocks; base += 256u) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < nblocks ? block_sums[i] : 0u;
		uint32_t total;
		const uint32_t ex = block_exclusive_scan_256(v, s_wave, &total);
		if (i < nblocks) block_sums[i] = carry + ex;
		carry += total;
	}
}

// writes batch_off (exclusive prefix) and appends keys with > GYS_SMALL_MAX new values to the huge work list
__global__ __launch_bounds__(256) void k_scan_final(const uint32_t *cnt, uint32_t n, const uint32_t *block_sums, uint32_t *off,
						    uint32_t *huge_list, uint32_t *huge_count)
{
	__shared__ uint32_t s_wave[4];
	const uint32_t base = blockIdx.x * GYS_SCAN_TILE + threadIdx.x * 16u;
	uint32_t v[16];
	uint32_t s = 0;
#pragma unroll
	for (uint32_t k = 0; k < 16u; ++k) {
		v[k] = (base + k < n) ? cnt[base + k] : 0u;
		s += v[k];
	}
	uint32_t total;
	uint32_t run = block_sums[blockIdx.x] + block_exclusive_scan_256(s, s_wave, &total);
#pragma unroll
	for (uint32_t k = 0; k < 16u; ++k) {
		if (base + k < n) {
			off[base + k] = run;
			if (v[k] > GYS_SMALL_MAX) huge_list[atomicAdd(huge_count, 1u)] = base + k;
		}
		run += v[k];
	}
}

// ---------------------------------------------------------------------------------------------------- scatter
// after this kernel off[slot] = segment END (start = off - cnt)
__global__ __launch_bounds__(256) void k_resp_scatter(const uint64_t *ev_kv, uint64_t n, uint32_t *off, uint32_t *staged)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const uint64_t kv = ev_kv[i];
		if (kv == ~0ull) continue;
		const uint32_t pos = atomicAdd(&off[(uint32_t)(kv >> 32)], 1u);
		staged[pos] = (uint32_t)kv;
	}
}

// ---------------------------------------------------------------------------------------------------- host-local resp pass
// One workgroup per host segment of the batch.  The reference resolves a response event's listener inside the HOST's own
// listener table (TCP_SOCK_HANDLER is per host: common/gy_socket_stat.cc:1554-1677), so everything an event touches before the
// per-key merge is host-local: the workgroup stages the host's (netns, port) -> local index sub-table in LDS, counts the segment's
// events per listener in LDS, scans the counts in LDS and scatters the staged words with LDS atomics into the segment's own slice
// of `staged` ([first_event, first_event + valid)).  No per-event device-scope atomic except the global HLL register max.
struct HostDesc {
	uint32_t tbl_off;  // first entry of the host's sub-table in the table pool
	uint32_t mask;     // sub-table capacity - 1 (power of two)
	uint32_t nlst;     // local listener indices in use
	uint32_t lst_off;  // first entry of the host's local index -> service slot list in the list pool
};

#define GYS_HOST_TBL_EMPTY 0xFFFFFFFFFFFFFFFFull
#define GYS_EV_DROPPED 0xFFFFFFFFu                          // response ms field 0xFFFFF > 10^6: never a kept event
#define GYS_EV_LOCAL(w) ((w) >> 20)
#define GYS_EV_STAGED(w, row) ((((w) & 0xFFFFFu) << GYS_ROW_BITS) | (uint32_t)(row))
#define GYS_HOST_THREADS 1024
#define GYS_HOST_UNROLL 4
#define GYS_HOST_TILE 8192u                                  // events per LDS scatter tile of a long segment
#define GYS_HOST_TILE_PER_THREAD (GYS_HOST_TILE / GYS_HOST_THREADS)

// Few hosts, long segments (a small installation, or C1's single host): one workgroup per host segment would leave most of the chip
// idle, so the segments are cut into PARTS of GYS_SPLIT_PART events and the pass runs in three launches instead of one:
//   k_resp_host<.., 1>  one workgroup per part: resolve / filter / HLL / per-event records as usual, per-listener counts of the part
//                       into its row of split_cnt
//   k_split_scan        one workgroup per host: key run starts from the column sums, every part's row becomes its run cursors
//   k_resp_host<.., 2>  one workgroup per part: scatter of the part's records through the LDS tile image, positions from its row
// The result (staged runs per key, batch_cnt / off_end) is exactly what the fused form writes.
#define GYS_SPLIT_PART 65536u
struct SplitPart {
	uint64_t real_first; // first event of the host's whole segment: staged positions are relative to it
	uint32_t cnt_off;    // first entry of the part's row in split_cnt
	uint32_t pad;
};
struct SplitSeg {
	uint64_t first_event; // of the host's whole segment
	uint32_t host_slot, nparts;
	uint32_t cnt_off;     // row of part 0; the parts' rows follow each other, L entries each
	uint32_t pad;
};

struct RespHostP {
	const uint64_t *ev;
	uint64_t n;
	const gys_resp_seg *segs;
	uint32_t nsegs;
	const HostDesc *hdesc;
	const uint64_t *htbl;   // entries: (netns:32 | port:16) << 16 | local index:16
	const uint32_t *hlst;
	uint32_t *hll32;
	uint32_t *batch_cnt, *off_end;
	// per event, written in pass A and re-read in pass B by the same thread: 5 bytes instead of the event's 24 --
	uint32_t *ev_w;         // local index << 20 | response ms (<= 10^6 < 2^20), GYS_EV_DROPPED = dropped
	uint8_t *ev_row;        // CONN_BITMAP row (client port & 0x1F); only written for kept events
	uint32_t *staged;
	uint32_t *huge_list, *huge_count;
	uint64_t *counters;
	uint8_t *svc_hll;
	uint32_t svc_hll_p;
	uint32_t lds_tbl_entries; // LDS table area of the launch (largest sub-table among the batch's hosts)
	uint32_t lds_cnt_entries; // LDS count area (largest listener count, even)
	uint32_t lds_region_entries; // LDS scatter region of the launch (u32 entries, 0 = none): segments that fit are sorted there
	uint32_t lds_tile_events;    // > 0: longer segments are scattered tile by tile through the region (2 x cnt + 2 x tile entries)
	// split form (MODE 1 / 2): `segs` are PARTS of host segments, see "few hosts, long segments" below
	const SplitPart *parts;
	uint32_t *split_cnt;         // per part a row of the host's L counters: counts (MODE 1), then run cursors (k_split_scan), read by MODE 2
};

// TILED = true: the instantiation for launches with long segments (keeps a tile's records in registers: more VGPRs, one workgroup per CU)
// MODE 0: the whole pass in one launch; 1 / 2: the two halves of the split form
template <bool TILED, int MODE = 0>
__global__ __launch_bounds__(GYS_HOST_THREADS) void k_resp_host(RespHostP p)
{
	extern __shared__ uint64_t s_dyn[];
	__shared__ uint32_t s_wsum[GYS_HOST_THREADS / 64];
	__shared__ uint32_t s_drop[2];
	__shared__ uint32_t s_floor;
	uint64_t *s_tbl = s_dyn;
	uint32_t *s_cnt = (uint32_t *)(s_dyn + p.lds_tbl_entries);
	uint32_t *s_region = s_cnt + p.lds_cnt_entries; // the segment's slice of `staged`, built in LDS and flushed with full-line stores
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const gys_resp_seg seg = p.segs[blockIdx.x];
	const uint64_t e0 = seg.first_event;
	const uint64_t e1 = blockIdx.x + 1 < p.nsegs ? p.segs[blockIdx.x + 1].first_event : p.n;
	if (e1 <= e0) return;
	const HostDesc hd = p.hdesc[seg.host_slot];
	const uint32_t mask = hd.mask, L = hd.nlst;
	// staged positions count from the first event of the HOST's segment (split form: the part's row holds cursors relative to it)
	const uint64_t sbase = MODE == 0 ? e0 : p.parts[blockIdx.x].real_first;
	if (MODE != 2) {
	for (uint32_t i = tid; i <= mask; i += GYS_HOST_THREADS) s_tbl[i] = p.htbl[hd.tbl_off + i];
	for (uint32_t i = tid; i < L; i += GYS_HOST_THREADS) s_cnt[i] = 0;
	if (tid < 2) s_drop[tid] = 0;
	// HLL floor: a register can only grow, so min over the register file (read once per workgroup; stale L1 lines only lower it) is a
	// lower bound for the rest of the window -- events whose rank does not exceed it skip the register read altogether.  Late in a
	// window that is all but ~2^-floor of the events; without it every event pays a random 4-byte read.
	if (tid == 0) s_floor = 0xFFFFFFFFu;
	__syncthreads();
	if (e1 - e0 >= 4096u) {
		uint32_t mn = 0xFFFFFFFFu;
		const uint4 *h4 = (const uint4 *)p.hll32;
		for (uint32_t i = tid; i < (1u << GYS_HLL_P) / 4u; i += GYS_HOST_THREADS) {
			const uint4 v = h4[i];
			mn = min(min(mn, min(v.x, v.y)), min(v.z, v.w));
		}
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, d, 64));
		if (lane == 0) atomicMin(&s_floor, mn);
	} else if (tid == 0) {
		s_floor = 0;
	}
	__syncthreads();
	const uint32_t hll_floor = s_floor;

	// ---- pass A: resolve, filter, count.  GYS_HOST_UNROLL events per thread and iteration, phase by phase (all event loads, then all
	// the arithmetic, then all HLL register reads, then the updates) so that each wave keeps several HBM requests in flight.
	uint32_t ndrop_range = 0, ndrop_nol = 0;
	for (uint64_t base = e0 + tid; base < e1; base += (uint64_t)GYS_HOST_UNROLL * GYS_HOST_THREADS) {
		uint64_t w0[GYS_HOST_UNROLL], w1[GYS_HOST_UNROLL], w2[GYS_HOST_UNROLL];
#pragma unroll
		for (int u = 0; u < GYS_HOST_UNROLL; ++u) {
			const uint64_t i = base + (uint64_t)u * GYS_HOST_THREADS;
			w0[u] = 0; w1[u] = 0; w2[u] = 0;
			if (i < e1) {
				w0[u] = p.ev[3 * i];
				w1[u] = p.ev[3 * i + 1];
				w2[u] = p.ev[3 * i + 2];
			}
		}
		uint32_t kw[GYS_HOST_UNROLL], krow[GYS_HOST_UNROLL];
		uint32_t hidx[GYS_HOST_UNROLL], hrank[GYS_HOST_UNROLL], hcur[GYS_HOST_UNROLL];
#pragma unroll
		for (int u = 0; u < GYS_HOST_UNROLL; ++u) {
			const uint64_t i = base + (uint64_t)u * GYS_HOST_THREADS;
			// struct ipv4_tuple_t {u32 saddr, daddr, netns; u16 sport, dport;} + u32 lsndtime, lrcvtime  (24 bytes)
			const uint32_t saddr = (uint32_t)w0[u], daddr = (uint32_t)(w0[u] >> 32);
			const uint32_t netns = (uint32_t)w1[u];
			const uint16_t sport = bswap16((uint16_t)(w1[u] >> 32)), dport = bswap16((uint16_t)(w1[u] >> 48)); // ntohs :1526-1527
			const uint32_t tresp = (uint32_t)w2[u] - (uint32_t)(w2[u] >> 32); // lsndtime - lrcvtime (:1519)
			kw[u] = GYS_EV_DROPPED;
			krow[u] = 0;
			hrank[u] = 0;
			hidx[u] = 0;
			if (i >= e1) continue;
			if (tresp > 1000000u) { // "Ignore responses > 1000 sec or negative" (:1521-1524)
				ndrop_range++;
				continue;
			}
			const uint64_t key48 = ((uint64_t)netns << 16) | (uint64_t)sport;
			uint32_t h = host_tbl_hash(key48) & mask;
			uint32_t local = GYS_NOSLOT;
			for (uint32_t probes = 0; probes <= mask; ++probes) {
				const uint64_t e = s_tbl[h];
				if ((e >> 16) == key48) {
					local = (uint32_t)(e & 0xFFFFu);
					break;
				}
				if (e == GYS_HOST_TBL_EMPTY) break;
				h = (h + 1) & mask;
			}
			if (local == GYS_NOSLOT) {
				ndrop_nol++; // no such listener: the reference ignores the event too (:1671-1676 miss path)
				continue;
			}
			kw[u] = (local << 20) | tresp;
			krow[u] = (uint32_t)dport & 0x1Fu;
			if (p.svc_hll_p) { // the per-service registers need the whole 64-bit hash
				const uint64_t h64 = flow_hash64(daddr, dport, saddr, sport);
				hll_idx_rank(h64, GYS_HLL_P, &hidx[u], &hrank[u]);
				svc_hll_update(p.svc_hll, p.svc_hll_p, p.hlst[hd.lst_off + local], h64);
			} else {
				flow_hll_idx_rank(daddr, dport, saddr, sport, &hidx[u], &hrank[u]);
			}
			if (hrank[u] <= hll_floor) hrank[u] = 0; // cannot raise any register
		}
#pragma unroll
		for (int u = 0; u < GYS_HOST_UNROLL; ++u) hcur[u] = hrank[u] ? p.hll32[hidx[u]] : 0xFFu; // read-first: most events do not raise the register
#pragma unroll
		for (int u = 0; u < GYS_HOST_UNROLL; ++u) {
			const uint64_t i = base + (uint64_t)u * GYS_HOST_THREADS;
			if (hcur[u] < hrank[u]) atomicMax(&p.hll32[hidx[u]], hrank[u]);
			if (kw[u] != GYS_EV_DROPPED) {
				atomicAdd(&s_cnt[GYS_EV_LOCAL(kw[u])], 1u);
				p.ev_row[i] = (uint8_t)krow[u];
			}
			if (i < e1) p.ev_w[i] = kw[u];
		}
	}
	if (ndrop_range) atomicAdd(&s_drop[0], ndrop_range);
	if (ndrop_nol) atomicAdd(&s_drop[1], ndrop_nol);
	__syncthreads();
	} // MODE != 2
	if (MODE == 1) { // split form, first half: the part's per-listener counts go to its row; k_split_scan turns the rows into cursors
		uint32_t *rowp = p.split_cnt + p.parts[blockIdx.x].cnt_off;
		for (uint32_t k = tid; k < L; k += GYS_HOST_THREADS) rowp[k] = s_cnt[k];
		if (tid == 0) {
			atomicAdd((unsigned long long *)&p.counters[CTR_RESP_EVENTS], (unsigned long long)(e1 - e0));
			if (s_drop[0]) atomicAdd((unsigned long long *)&p.counters[CTR_RESP_DROP_RANGE], (unsigned long long)s_drop[0]);
			if (s_drop[1]) atomicAdd((unsigned long long *)&p.counters[CTR_RESP_DROP_NOLISTENER], (unsigned long long)s_drop[1]);
		}
		return;
	}
	if (MODE == 2) { // split form, second half: run cursors of this part (relative to the host segment's start)
		const uint32_t *rowp = p.split_cnt + p.parts[blockIdx.x].cnt_off;
		for (uint32_t k = tid; k < L; k += GYS_HOST_THREADS) s_cnt[k] = rowp[k];
	}

	// ---- counts -> per-key run starts (exclusive scan over the local indices), per-key batch_cnt / off_end for the digest kernels
	if (MODE == 0) {
		const uint32_t K = (L + GYS_HOST_THREADS - 1) / GYS_HOST_THREADS;
		const uint32_t lo = tid * K, hi = min(L, lo + K);
		uint32_t sum = 0;
		for (uint32_t k = lo; k < hi; ++k) sum += s_cnt[k];
		uint32_t inc = sum;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = __shfl_up(inc, d, 64);
			if ((int)lane >= d) inc += t;
		}
		if (lane == 63) s_wsum[wave] = inc;
		__syncthreads();
		uint32_t run = inc - sum;
		for (uint32_t w = 0; w < wave; ++w) run += s_wsum[w];
		for (uint32_t k = lo; k < hi; ++k) {
			const uint32_t c = s_cnt[k];
			s_cnt[k] = run;
			if (c) {
				const uint32_t slot = p.hlst[hd.lst_off + k];
				p.batch_cnt[slot] = c;
				p.off_end[slot] = (uint32_t)e0 + run + c;
				if (c > GYS_SMALL_MAX) p.huge_list[atomicAdd(p.huge_count, 1u)] = slot;
			}
			run += c;
		}
	}
	__syncthreads();

	// ---- pass B: scatter the staged words into the key runs (positions from LDS atomics).  A scattered 4-byte store that leaves L2
	// before its line is complete becomes a read-modify-write in HBM, so when the segment's slice fits the LDS region the runs are
	// assembled there and written out with coalesced full-line stores.
	// (a part's key runs are not contiguous in `staged`: the split form always goes through the tile image)
	const bool tiled = TILED && p.lds_tile_events != 0 && (MODE == 2 || (e1 - e0) > (uint64_t)p.lds_tile_events);
	const bool in_lds = !tiled && (e1 - e0) <= (uint64_t)p.lds_region_entries;
	if (tiled) {
		// Long segment: the key runs are assembled tile by tile.  Per tile of GYS_HOST_TILE events: per-key counts of the tile (LDS), scan,
		// scatter of the tile's words into an LDS image grouped by key together with their final positions (the key's global cursor +
		// rank inside the tile's run), flush -- consecutive image entries of a key go to consecutive addresses, so the stores leave as
		// full sectors -- and the cursors advance by the tile's counts.  s_cnt holds the cursors (run starts after the scan above).
		const uint32_t Lc = p.lds_cnt_entries;
		uint32_t *s_tstart = s_region;                 // [Lc] start of the key's run inside the tile image
		uint32_t *s_tcur = s_region + Lc;              // [Lc] counts, then running cursor inside the tile image
		uint32_t *s_val = s_region + 2u * Lc;          // [tile] words
		uint32_t *s_dest = s_val + GYS_HOST_TILE;      // [tile] final position (relative to e0)
		const uint32_t K = (L + GYS_HOST_THREADS - 1) / GYS_HOST_THREADS;
		const uint32_t klo = tid * K, khi = min(L, klo + K);
		for (uint64_t t0 = e0; t0 < e1; t0 += GYS_HOST_TILE) {
			for (uint32_t k = tid; k < L; k += GYS_HOST_THREADS) s_tcur[k] = 0;
			__syncthreads();
			uint32_t kwr[GYS_HOST_TILE_PER_THREAD], rowr[GYS_HOST_TILE_PER_THREAD];
#pragma unroll
			for (uint32_t u = 0; u < GYS_HOST_TILE_PER_THREAD; ++u) {
				const uint64_t i = t0 + tid + (uint64_t)u * GYS_HOST_THREADS;
				kwr[u] = i < e1 ? p.ev_w[i] : GYS_EV_DROPPED;
			}
#pragma unroll
			for (uint32_t u = 0; u < GYS_HOST_TILE_PER_THREAD; ++u) {
				const uint64_t i = t0 + tid + (uint64_t)u * GYS_HOST_THREADS;
				rowr[u] = kwr[u] != GYS_EV_DROPPED ? (uint32_t)p.ev_row[i] : 0u;
			}
#pragma unroll
			for (uint32_t u = 0; u < GYS_HOST_TILE_PER_THREAD; ++u)
				if (kwr[u] != GYS_EV_DROPPED) atomicAdd(&s_tcur[GYS_EV_LOCAL(kwr[u])], 1u);
			__syncthreads();
			{ // exclusive scan of the tile counts -> run starts inside the image (s_tstart) and scatter cursors (s_tcur)
				uint32_t sum = 0;
				for (uint32_t k = klo; k < khi; ++k) sum += s_tcur[k];
				uint32_t inc = sum;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) {
					const uint32_t t = __shfl_up(inc, d, 64);
					if ((int)lane >= d) inc += t;
				}
				if (lane == 63) s_wsum[wave] = inc;
				__syncthreads();
				uint32_t run = inc - sum;
				for (uint32_t w = 0; w < wave; ++w) run += s_wsum[w];
				for (uint32_t k = klo; k < khi; ++k) {
					const uint32_t c = s_tcur[k];
					s_tstart[k] = run;
					s_tcur[k] = run;
					run += c;
				}
			}
			__syncthreads();
#pragma unroll
			for (uint32_t u = 0; u < GYS_HOST_TILE_PER_THREAD; ++u) {
				if (kwr[u] == GYS_EV_DROPPED) continue;
				const uint32_t local = GYS_EV_LOCAL(kwr[u]);
				const uint32_t idx = atomicAdd(&s_tcur[local], 1u);
				s_val[idx] = GYS_EV_STAGED(kwr[u], rowr[u]);
				s_dest[idx] = s_cnt[local] + (idx - s_tstart[local]);
			}
			__syncthreads();
			uint32_t ntile = 0; // valid words of the tile = end cursor of the last key = total (every thread computes it from the wave sums)
			for (uint32_t w = 0; w < GYS_HOST_THREADS / 64; ++w) ntile += s_wsum[w];
			for (uint32_t e = tid; e < ntile; e += GYS_HOST_THREADS) p.staged[sbase + s_dest[e]] = s_val[e];
			for (uint32_t k = tid; k < L; k += GYS_HOST_THREADS) s_cnt[k] += s_tcur[k] - s_tstart[k];
			__syncthreads();
		}
	} else {
		for (uint64_t base = e0 + tid; base < e1; base += (uint64_t)GYS_HOST_UNROLL * GYS_HOST_THREADS) {
			uint32_t kw[GYS_HOST_UNROLL], krow[GYS_HOST_UNROLL];
#pragma unroll
			for (int u = 0; u < GYS_HOST_UNROLL; ++u) {
				const uint64_t i = base + (uint64_t)u * GYS_HOST_THREADS;
				kw[u] = i < e1 ? p.ev_w[i] : GYS_EV_DROPPED;
			}
#pragma unroll
			for (int u = 0; u < GYS_HOST_UNROLL; ++u) {
				const uint64_t i = base + (uint64_t)u * GYS_HOST_THREADS;
				krow[u] = kw[u] != GYS_EV_DROPPED ? (uint32_t)p.ev_row[i] : 0u;
			}
#pragma unroll
			for (int u = 0; u < GYS_HOST_UNROLL; ++u) {
				if (kw[u] == GYS_EV_DROPPED) continue;
				const uint32_t pos = atomicAdd(&s_cnt[GYS_EV_LOCAL(kw[u])], 1u);
				const uint32_t word = GYS_EV_STAGED(kw[u], krow[u]);
				if (in_lds) s_region[pos] = word;
				else p.staged[e0 + pos] = word;
			}
		}
		if (in_lds) {
			__syncthreads();
			const uint32_t nvalid = (uint32_t)(e1 - e0) - s_drop[0] - s_drop[1];
			for (uint32_t i = tid; i < nvalid; i += GYS_HOST_THREADS) p.staged[e0 + i] = s_region[i];
		}
	}
	if (MODE == 0 && tid == 0) {
		atomicAdd((unsigned long long *)&p.counters[CTR_RESP_EVENTS], (unsigned long long)(e1 - e0));
		if (s_drop[0]) atomicAdd((unsigned long long *)&p.counters[CTR_RESP_DROP_RANGE], (unsigned long long)s_drop[0]);
		if (s_drop[1]) atomicAdd((unsigned long long *)&p.counters[CTR_RESP_DROP_NOLISTENER], (unsigned long long)s_drop[1]);
	}
}

// split form, middle launch: one workgroup per host.  Column k of the host's count matrix (one row per part) sums to the key's run
// length; an exclusive scan over the keys gives the run starts; every row entry becomes the position (relative to the host segment's
// start) where that part's first value of the key goes.  Also the per-key batch_cnt / off_end / huge list, as the fused form writes them.
struct SplitScanP {
	const SplitSeg *segs;
	const HostDesc *hdesc;
	const uint32_t *hlst;
	uint32_t *split_cnt;
	uint32_t *batch_cnt, *off_end;
	uint32_t *huge_list, *huge_count;
};

__global__ __launch_bounds__(GYS_HOST_THREADS) void k_split_scan(SplitScanP p)
{
	__shared__ uint32_t s_wsum[GYS_HOST_THREADS / 64];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const SplitSeg sg = p.segs[blockIdx.x];
	const HostDesc hd = p.hdesc[sg.host_slot];
	const uint32_t L = hd.nlst;
	uint32_t *mat = p.split_cnt + sg.cnt_off;
	const uint32_t K = (L + GYS_HOST_THREADS - 1) / GYS_HOST_THREADS;
	const uint32_t lo = min(L, tid * K), hi = min(L, lo + K);
	uint32_t sum = 0;
	for (uint32_t k = lo; k < hi; ++k)
		for (uint32_t q = 0; q < sg.nparts; ++q) sum += mat[(size_t)q * L + k];
	uint32_t inc = sum;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t t = __shfl_up(inc, d, 64);
		if ((int)lane >= d) inc += t;
	}
	if (lane == 63) s_wsum[wave] = inc;
	__syncthreads();
	uint32_t run = inc - sum;
	for (uint32_t w = 0; w < wave; ++w) run += s_wsum[w];
	for (uint32_t k = lo; k < hi; ++k) {
		uint32_t cur = run;
		for (uint32_t q = 0; q < sg.nparts; ++q) {
			const uint32_t c = mat[(size_t)q * L + k];
			mat[(size_t)q * L + k] = cur;
			cur += c;
		}
		const uint32_t c = cur - run;
		if (c) {
			const uint32_t slot = p.hlst[hd.lst_off + k];
			p.batch_cnt[slot] = c;
			p.off_end[slot] = (uint32_t)sg.first_event + cur;
			if (c > GYS_SMALL_MAX) p.huge_list[atomicAdd(p.huge_count, 1u)] = slot;
		}
		run = cur;
	}
}

// ---------------------------------------------------------------------------------------------------- per-key pass + t-digest merge
// t-digest state of a key: 100 exact-integer clusters (td_sum / td_cnt) + a buffer of up to GYS_TD_PEND_CAP unmerged values
// (the classic merging-digest buffer).  A batch's values are appended to the buffer; only when they no longer fit is the key
// re-clustered with (buffered + new) values in ONE merge.  So the per-batch per-key work is light (k_key_pass: histogram record,
// CONN_BITMAP, Count-Min, min/max, buffer append) and the expensive cluster merge (k_digest_merge) runs once per ~CAP values.
static_assert(GYS_TD_PEND_CAP == GYS_TDIGEST_PEND_CAP, "gysketch.h and gys_tdigest_tbl.h disagree on the t-digest buffer size");

struct TdMeta {
	int32_t vmin, vmax; // over merged AND buffered values (INT32_MAX / INT32_MIN when empty)
	uint32_t npend;     // buffered values in td_pend[slot * CAP ..]
	uint32_t win_epoch; // window number the key's hist_win record and CONN_BITMAP rows belong to (lazy window roll, see below)
};

// Lazy window roll.  The reference clears the per-listener 5-s state and folds it into the longer levels on a timer
// (GY_HISTOGRAM::add_histogram / clear, common/gy_statistics.h:625-636).  Sweeping 10^7 records at every window boundary costs
// ~1 KB of HBM traffic per key, so the engine tags each key with the window number its hist_win / bitmap contents belong to and
// rolls a key the first time a later window touches it: all-time += old window record, window record := this batch.
//   window view   = hist_win if win_epoch == current window, else empty
//   all-time view = hist_all + hist_win (the window record is either the current window or a not-yet-folded older one)

struct MergeEnt {
	uint32_t slot;
	uint32_t m;        // new values of the batch (staged[off_end - m .. off_end)), 0 for a query entry
	uint32_t off_end;
	uint32_t pad;
};

struct DigestP {
	int64_t *td_sum;    // [nsvc*100]
	uint32_t *td_cnt;   // [nsvc*100]
	TdMeta *td_meta;    // [nsvc]
	uint32_t *td_pend;  // [nsvc*CAP]
	uint32_t *batch_cnt;
	const uint32_t *off_end;
	const uint32_t *staged;
	uint32_t nsvc;
	// per-key outputs of the batch (the per-key kernels see every value of the key: one coalesced record update per KEY instead of
	// ~9 device atomics per EVENT)
	gys_hist_rec *hist_win;
	uint32_t *cms32;
	const uint64_t *svc_gid;
	uint32_t *bitmap; // [nsvc*16] u32 = 32 x u16 CONN_BITMAP rows (common/gy_socket_stat.h:390-454)
	MergeEnt *merge_list;
	uint32_t *merge_count;
	gys_hist_rec *hist_all;
	uint32_t epoch;              // current window number
	unsigned long long *ghist;   // arena: all-service histogram of the window, 15 x {count,sum} + {total}
	uint32_t chunk_lo, chunk_hi; // k_key_pass: 64-key chunks [chunk_lo, chunk_hi) of this launch (key ranges pipeline against the merges)
	long long *gmax;             // arena: largest value of the window
};

// CONN_BITMAP::add_response for one staged word into a 16-word LDS row image: respmap_[row].set(bucket)
__device__ __forceinline__ void bitmap_set_lds(uint32_t *s_bm, uint32_t word, uint32_t bucket)
{
	const uint32_t row = word & 0x1Fu;
	atomicOr(&s_bm[row >> 1], (1u << bucket) << ((row & 1u) * 16u));
}

// wave-synchronous LDS hand-off: DS operations of one wave execute in order; this only stops the compiler from moving them
#define GYS_WAVE_SYNC()                                              \
	do {                                                         \
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
		__builtin_amdgcn_wave_barrier();                     \
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
	} while (0)

__device__ __forceinline__ uint32_t td_cluster_of(const uint64_t *T, uint64_t mid2)
{
	uint32_t a = 0, bb = GYS_TD_NB - 1;
	while (a < bb) {
		const uint32_t mid = (a + bb + 1) >> 1;
		if (mid2 >= T[mid]) a = mid; else bb = mid - 1;
	}
	return a;
}

// same on a table padded to 128 entries with ~0: branch-free (7 dependent LDS reads, no divergent loop)
__device__ __forceinline__ uint32_t td_cluster_of128(const uint64_t *T, uint64_t mid2)
{
	uint32_t a = 0; // #{j in 1..127 : mid2 >= T[j]}
#pragma unroll
	for (uint32_t step = 64u; step >= 1u; step >>= 1)
		if (mid2 >= T[a + step]) a += step;
	return a;
}

// lanes 0..63 each own entries (lane) and (lane + 64) of a <= 128 long array: exclusive prefix sum (u64) across the wave
__device__ __forceinline__ void wave_excl_scan_2x(uint64_t a0, uint64_t a1, uint64_t *e0, uint64_t *e1, uint64_t *total)
{
	const uint32_t lane = threadIdx.x & 63u;
	uint64_t i0 = a0, i1 = a1;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint64_t t0 = __shfl_up(i0, d, 64), t1 = __shfl_up(i1, d, 64);
		if ((int)lane >= d) {
			i0 += t0;
			i1 += t1;
		}
	}
	const uint64_t tot0 = __shfl(i0, 63, 64), tot1 = __shfl(i1, 63, 64);
	*e0 = i0 - a0;
	*e1 = tot0 + i1 - a1;
	*total = tot0 + tot1;
}

// ---- k_key_pass: every key with 1..GYS_SMALL_MAX new values.  FOUR keys per wave: each 16-lane row owns one key -- lane g of the
// row holds the key's histogram pair g (16 x {count,sum} = the 256-byte record, one coalesced 256-B access per row), its CONN_BITMAP
// word g, and value g of every 16-value group.  A wave walks a chunk of 64 consecutive keys in 16 rounds of 4 keys: the chunk's
// counts / offsets come from ONE coalesced load (then bpermute), and the next round's record / bitmap / meta / first 32 staged words
// are prefetched into registers while the current round is processed, so the per-key critical path holds no dependent HBM round
// trip.  No workgroup barriers (rows talk through per-row LDS accumulators).  Keys whose buffer would overflow are queued for
// k_digest_merge with one aggregated atomic per chunk.
struct KeyRegs {
	uint32_t w0, w1;  // staged words g and 16 + g of the key
	uint4 pair;       // histogram pair g of the window record: {count lo, count hi, sum lo, sum hi}
	uint4 apair;      // the same pair of the all-time record (needed when the key rolls to a new window)
	uint32_t bm;      // CONN_BITMAP word g
	uint4 meta;       // TdMeta of the key (same for the 16 lanes of the row)
	uint64_t gid;
};

__device__ __forceinline__ void key_prefetch(const DigestP &p, uint32_t slot, uint32_t m, uint32_t oend, uint32_t g, KeyRegs &r)
{
	r.w0 = 0;
	r.w1 = 0;
	if (m == 0) return; // row idle this round
	const uint32_t *sv = p.staged + (oend - m);
	if (g < m) r.w0 = sv[g];
	if (16u + g < m) r.w1 = sv[16u + g];
	r.pair = ((const uint4 *)&p.hist_win[slot])[g];
	r.apair = ((const uint4 *)&p.hist_all[slot])[g];
	r.bm = p.bitmap[(size_t)slot * 16u + g];
	r.meta = *(const uint4 *)&p.td_meta[slot];
	r.gid = p.svc_gid[slot];
}

__global__ __launch_bounds__(256) void k_key_pass(DigestP p)
{
	__shared__ unsigned long long s_h_[16][32];
	__shared__ uint32_t s_bm_[16][16];
	__shared__ unsigned long long s_gh[32]; // all-service histogram of this workgroup's keys (flushed once at the end)
	__shared__ long long s_gmax;
	if (threadIdx.x < 32u) s_gh[threadIdx.x] = 0;
	if (threadIdx.x == 32u) s_gmax = INT64_MIN;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	const uint32_t row = lane >> 4, g = lane & 15u;
	unsigned long long *s_h = s_h_[wv * 4u + row];
	uint32_t *s_bm = s_bm_[wv * 4u + row];
	const uint32_t nwaves = gridDim.x * 4u;

	for (uint32_t chunk = p.chunk_lo + blockIdx.x * 4u + wv; chunk < p.chunk_hi; chunk += nwaves) {
		const uint32_t key = chunk * 64u + lane;
		uint32_t mcnt = key < p.nsvc ? p.batch_cnt[key] : 0u;
		const uint32_t oend = key < p.nsvc ? p.off_end[key] : 0u;
		if (mcnt > GYS_SMALL_MAX) mcnt = 0; // larger keys belong to k_digest_huge (which also does their histogram / bitmap / CMS)
		const unsigned long long todo = __ballot(mcnt != 0);
		if (!todo) continue;
		if (mcnt) p.batch_cnt[key] = 0; // consumed (coalesced reset for the whole chunk)
		uint32_t merge_rounds = 0;      // bit rd: the key this row handled in round rd must be merged
		KeyRegs cur, nxt;
		uint32_t rd = (uint32_t)__ffsll((long long)todo) - 1u;
		rd >>= 2; // first round with an active key
		key_prefetch(p, chunk * 64u + rd * 4u + row, (uint32_t)__shfl((int)mcnt, (int)(rd * 4u + row), 64),
			     (uint32_t)__shfl((int)oend, (int)(rd * 4u + row), 64), g, cur);
		for (;;) {
			const uint32_t k = rd * 4u + row;
			const uint32_t slot = chunk * 64u + k;
			const uint32_t m = (uint32_t)__shfl((int)mcnt, (int)k, 64);
			const uint32_t ke = (uint32_t)__shfl((int)oend, (int)k, 64);
			uint32_t rnext = 16u;
			{
				const unsigned long long rest = rd < 15u ? (todo >> (4u * (rd + 1u))) : 0ull;
				if (rest) { // issue the next round's loads now; they are consumed one iteration later
					rnext = rd + 1u + (((uint32_t)__ffsll((long long)rest) - 1u) >> 2);
					key_prefetch(p, chunk * 64u + rnext * 4u + row, (uint32_t)__shfl((int)mcnt, (int)(rnext * 4u + row), 64),
						     (uint32_t)__shfl((int)oend, (int)(rnext * 4u + row), 64), g, nxt);
				}
			}
			const uint32_t npend = cur.meta.z;
			const bool do_merge = m != 0 && npend + m > GYS_TD_PEND_CAP;
			s_h[g] = 0;
			s_h[16u + g] = 0;
			s_bm[g] = 0;
			GYS_WAVE_SYNC();
			uint32_t *pend = p.td_pend + (size_t)slot * GYS_TD_PEND_CAP + npend;
			const uint32_t mmax = max(max((uint32_t)__shfl((int)m, 0, 64), (uint32_t)__shfl((int)m, 16, 64)),
						  max((uint32_t)__shfl((int)m, 32, 64), (uint32_t)__shfl((int)m, 48, 64)));
			int32_t lmin = INT32_MAX, lmax = INT32_MIN;
			for (uint32_t base = 0; base < mmax; base += 16u) {
				const uint32_t idx = base + g;
				if (idx < m) {
					// staged word: value << 5 | CONN_BITMAP row
					const uint32_t w = base == 0 ? cur.w0 : (base == 16u ? cur.w1 : p.staged[ke - m + idx]);
					const int32_t v = (int32_t)(w >> GYS_ROW_BITS);
					const uint32_t b = resp_bucket((int64_t)v);
					atomicAdd(&s_h[2 * b], 1ull);
					atomicAdd(&s_h[2 * b + 1], (unsigned long long)(int64_t)v);
					bitmap_set_lds(s_bm, w, b);
					lmin = min(lmin, v);
					lmax = max(lmax, v);
					if (!do_merge) pend[idx] = (uint32_t)v;
				}
			}
			// smallest / largest value of the key: per-lane running values, then an xor butterfly inside the 16-lane row (16 lanes hammering
			// one LDS word with atomicMin / atomicMax serialise; the LDS pipe of this kernel was ~70 % conflict cycles)
#pragma unroll
			for (int d = 8; d >= 1; d >>= 1) {
				lmin = min(lmin, __shfl_xor(lmin, d, 64));
				lmax = max(lmax, __shfl_xor(lmax, d, 64));
			}
			GYS_WAVE_SYNC();
			if (m) {
				const int32_t vmin = lmin, vmax = lmax;
				const bool stale = cur.meta.w != p.epoch; // first touch of the key in this window: roll it (see "Lazy window roll")
				// ---- histogram records (prefetched pairs + LDS delta), bitmap word, meta, Count-Min
				uint4 *hp = (uint4 *)&p.hist_win[slot] + g;
				const uint64_t w_lo = (uint64_t)cur.pair.x | ((uint64_t)cur.pair.y << 32), w_hi = (uint64_t)cur.pair.z | ((uint64_t)cur.pair.w << 32);
				if (stale) { // all-time += old window record (GY_HISTOGRAM::add_histogram, common/gy_statistics.h:625-660)
					const uint64_t a_lo = (uint64_t)cur.apair.x | ((uint64_t)cur.apair.y << 32), a_hi = (uint64_t)cur.apair.z | ((uint64_t)cur.apair.w << 32);
					uint64_t n_lo = a_lo + w_lo, n_hi = a_hi + w_hi;
					if (g == 15u) n_hi = (uint64_t)max((int64_t)a_hi, (int64_t)w_hi); // max_val_seen_
					if (n_lo != a_lo || n_hi != a_hi)
						((uint4 *)&p.hist_all[slot])[g] = make_uint4((uint32_t)n_lo, (uint32_t)(n_lo >> 32), (uint32_t)n_hi, (uint32_t)(n_hi >> 32));
				}
				if (g < 15u) {
					const unsigned long long dc = s_h[2 * g], ds = s_h[2 * g + 1];
					if (dc) {
						atomicAdd(&s_gh[2 * g], dc);
						atomicAdd(&s_gh[2 * g + 1], ds);
					}
					if (dc || (stale && (w_lo | w_hi))) {
						const uint64_t cnt = (stale ? 0ull : w_lo) + dc;
						const uint64_t sum = (stale ? 0ull : w_hi) + ds;
						*hp = make_uint4((uint32_t)cnt, (uint32_t)(cnt >> 32), (uint32_t)sum, (uint32_t)(sum >> 32));
					}
				} else {
					const uint64_t tot = (stale ? 0ull : w_lo) + m; // total_count_
					int64_t mx = stale ? INT64_MIN : (int64_t)w_hi;   // max_val_seen_
					if (mx < (int64_t)vmax) mx = (int64_t)vmax;
					*hp = make_uint4((uint32_t)tot, (uint32_t)(tot >> 32), (uint32_t)(uint64_t)mx, (uint32_t)((uint64_t)mx >> 32));
					atomicAdd(&s_gh[30], (unsigned long long)m);
					atomicMax(&s_gmax, (long long)vmax);
				}
				{
					const uint32_t old = stale ? 0u : cur.bm;
					const uint32_t bits = s_bm[g] | old;
					if (bits != cur.bm) p.bitmap[(size_t)slot * 16u + g] = bits;
				}
				if (g == 0) {
					const int32_t mn = min((int32_t)cur.meta.x, vmin), mx = max((int32_t)cur.meta.y, vmax);
					*(uint4 *)&p.td_meta[slot] = make_uint4((uint32_t)mn, (uint32_t)mx, do_merge ? npend : npend + m, p.epoch);
				} else if (g >= 4u && g < 8u) {
					const uint32_t r = g - 4u;
					atomicAdd(&p.cms32[r * GYS_CMS_W + (jhash2_u64(cur.gid, GYS_SEED + r) & (GYS_CMS_W - 1))], m);
				}
				if (do_merge) merge_rounds |= 1u << rd;
			}
			GYS_WAVE_SYNC();
			if (rnext >= 16u) break;
			rd = rnext;
			cur = nxt;
		}
		// ---- queue the keys whose buffer overflowed: lane = key of the chunk again (its count / offset are still in registers)
		{
			const uint32_t mr = (uint32_t)__shfl((int)merge_rounds, (int)((lane & 3u) * 16u), 64);
			const bool need = (mr >> (lane >> 2)) & 1u;
			const unsigned long long nb = __ballot(need);
			if (nb) {
				uint32_t at = 0;
				if (lane == 0) at = atomicAdd(p.merge_count, (uint32_t)__popcll(nb));
				at = (uint32_t)__shfl((int)at, 0, 64);
				if (need) {
					const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
					MergeEnt e;
					e.slot = key;
					e.m = mcnt;
					e.off_end = oend;
					e.pad = 0;
					p.merge_list[at + (uint32_t)__popcll(nb & below)] = e;
				}
			}
		}
	}
	__syncthreads();
	if (threadIdx.x < 31u && s_gh[threadIdx.x]) atomicAdd(&p.ghist[threadIdx.x], s_gh[threadIdx.x]);
	if (threadIdx.x == 31u && s_gmax != INT64_MIN) atomicMax(p.gmax, s_gmax);
}

// quarter-octave grid cell of a value < 2^20: 0,1,2,3 for 0..3, then 4 cells per power of two (monotone; <= 75)
__device__ __forceinline__ uint32_t value_grid(uint32_t v)
{
	if (v < 4u) return v;
	const uint32_t msb = 31u - (uint32_t)__clz((int)v);
	return 4u * (msb - 1u) + ((v >> (msb - 2u)) & 3u);
}
#define GYS_IVL 192u // refined intervals: gap index (<= 100) + grid cell (<= 75) < 192 = 3 per lane

// ---- k_digest_merge: one 64-thread workgroup (= one wave) per merge-list entry.  Exact-integer k-bucket merge (DESIGN.md "t-digest"):
//   values = the key's buffered values + the batch's new values; merged order = by mean, old clusters before values on ties; an item
//   with weighted mid-point mid2/2 of N goes to cluster #{j : mid2 >= T_j}, T_j = ceil(BND[j] * 2N / 2^32).
// No sort: the (<= 100) old cluster means cut the value axis into gaps; gap(v) = #{clusters with mean <= v} comes from a branch-free
// search; the gaps are refined by a fixed quarter-octave value grid (so a cell stays small even when the digest is empty or the
// distribution has moved away from its clusters); a counting sort by refined interval groups the values in LDS, and a value's rank
// is (values in lower intervals) + (its rank inside its own interval, by direct comparison -- an interval holds a handful of the
// batch's values, all equal for typical integer-ms data).  Old cluster c is preceded by exactly the values of gaps 0..c.  Everything else is integer
// arithmetic on ranks, so the result equals the sorted-merge definition bit for bit (ties among equal values are interchangeable).
// query mode (out_sum != nullptr): entry w writes the merged view of its key to out_sum/out_cnt[w*100..] and leaves the state alone.
#define GYS_MERGE_MAX (GYS_TD_PEND_CAP + GYS_SMALL_MAX)

struct MergeP {
	DigestP d;
	const MergeEnt *list;
	const uint32_t *count;
	int64_t *out_sum;
	uint32_t *out_cnt;
};

// Three instantiations share the list, split by the number of new values (LDS for CAP + NEWMAX values: the smaller the class, the more
// waves are resident): NEWMAX = 128 takes the entries with <= 128 new values, 384 those with 129..384, GYS_SMALL_MAX the rest.
// ceil(cs / cc) for 0 <= cs < 2^52, cc >= 1 (a cluster's integer "mean threshold": mean <= v  <=>  ceil(cs/cc) <= v for integer v)
__device__ __forceinline__ uint32_t ceil_div_sum_cnt(int64_t cs, uint32_t cc)
{
	uint64_t f = (uint64_t)((double)cs / (double)cc); // cs is exact in a double; the quotient may be off by one after rounding
	int64_t r = cs - (int64_t)(f * (uint64_t)cc);
	if (r < 0) {
		f--;
		r += cc;
	} else if (r >= (int64_t)cc) {
		f++;
		r -= cc;
	}
	return (uint32_t)(f + (r != 0));
}

template <uint32_t NEWMAX>
__global__ __launch_bounds__(64) void k_digest_merge(MergeP q)
{
	const DigestP &p = q.d;
	__shared__ uint32_t s_x[GYS_TD_PEND_CAP + NEWMAX];  // interval << 20 | value, in arrival order
	__shared__ uint32_t s_g[GYS_TD_PEND_CAP + NEWMAX];  // the same words grouped by interval
	__shared__ uint32_t s_thr[128];          // compacted non-empty old clusters: ceil(sum / count), padded with ~0 for the branch-free search
	__shared__ uint64_t s_cpfx[GYS_TD_NB + 1];
	__shared__ uint64_t s_T[128];            // s_T[j], j = 1..NB-1; [NB..127] = ~0 (never reached)
	__shared__ uint32_t s_imin[GYS_IVL], s_imax[GYS_IVL]; // smallest / largest value of each interval
	__shared__ unsigned long long s_osum[GYS_TD_NB];
	__shared__ uint32_t s_ocnt[GYS_TD_NB];
	__shared__ uint32_t s_icnt[GYS_IVL];     // values per (refined) interval, then the running scatter cursor
	__shared__ uint32_t s_ioff[GYS_IVL + 1]; // exclusive prefix of s_icnt (s_ioff[i + 1] = values in intervals 0..i)
	__shared__ uint32_t s_clt[GYS_TD_NB + 2]; // s_clt[c + 1] = values below the mean of compacted cluster c (prefix of the per-cluster-gap counts)
	const uint32_t lane = threadIdx.x;
	const uint32_t nent = *q.count;

	for (uint32_t w = blockIdx.x; w < nent; w += gridDim.x) {
		const MergeEnt ent = q.list[w];
		if (NEWMAX == 128u ? ent.m > 128u : (NEWMAX == 384u ? (ent.m <= 128u || ent.m > 384u) : ent.m <= 384u)) continue; // another class's entry
		const uint32_t slot = ent.slot;
		const uint32_t npend = p.td_meta[slot].npend;
		const uint32_t m = npend + ent.m;
		const uint32_t start = ent.off_end - ent.m;

		// ---- old digest: entries lane, lane+64
		const int64_t *gs = p.td_sum + (size_t)slot * GYS_TD_NB;
		const uint32_t *gc = p.td_cnt + (size_t)slot * GYS_TD_NB;
		const uint32_t j1 = lane + 64u;
		const uint32_t c0 = gc[lane];
		const uint32_t c1 = j1 < GYS_TD_NB ? gc[j1] : 0u;
		const int64_t sm0 = gs[lane];
		const int64_t sm1 = j1 < GYS_TD_NB ? gs[j1] : 0;
		if (m == 0) { // nothing buffered (query of a freshly merged key): the merged view is the digest itself
			if (q.out_sum) {
				q.out_sum[(size_t)w * GYS_TD_NB + lane] = sm0;
				q.out_cnt[(size_t)w * GYS_TD_NB + lane] = c0;
				if (j1 < GYS_TD_NB) {
					q.out_sum[(size_t)w * GYS_TD_NB + j1] = sm1;
					q.out_cnt[(size_t)w * GYS_TD_NB + j1] = c1;
				}
			}
			continue;
		}
		// compaction of non-empty clusters (order preserving): compacted index pos0 / pos1 of this lane's two entries
		const unsigned long long b0 = __ballot(c0 != 0), b1 = __ballot(c1 != 0);
		const uint32_t n0 = (uint32_t)__popcll(b0), nc = n0 + (uint32_t)__popcll(b1);
		const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
		const uint32_t pos0 = (uint32_t)__popcll(b0 & below), pos1 = n0 + (uint32_t)__popcll(b1 & below);
		uint64_t e0, e1, nold;
		wave_excl_scan_2x((uint64_t)c0, (uint64_t)c1, &e0, &e1, &nold);
		s_thr[lane] = 0xFFFFFFFFu; // pad: mean = +inf
		s_thr[j1] = 0xFFFFFFFFu;
		GYS_WAVE_SYNC();
		if (c0) {
			s_thr[pos0] = ceil_div_sum_cnt(sm0, c0);
			s_cpfx[pos0] = e0;
		}
		if (c1) {
			s_thr[pos1] = ceil_div_sum_cnt(sm1, c1);
			s_cpfx[pos1] = e1;
		}
		if (lane == 0) s_cpfx[nc] = nold;
		const uint64_t twoN = 2ull * (nold + (uint64_t)m);
		if (lane >= 1) s_T[lane] = td_threshold(c_td_bnd[lane], twoN);
		s_T[j1] = j1 < GYS_TD_NB ? td_threshold(c_td_bnd[j1], twoN) : ~0ull;
		s_osum[lane] = 0;
		s_ocnt[lane] = 0;
		for (uint32_t i = lane; i < GYS_IVL; i += 64u) {
			s_icnt[i] = 0;
			s_imin[i] = 0xFFFFFFFFu;
			s_imax[i] = 0;
		}
		s_clt[lane] = 0;
		if (j1 < GYS_TD_NB + 2) s_clt[j1] = 0;
		if (j1 < GYS_TD_NB) {
			s_osum[j1] = 0;
			s_ocnt[j1] = 0;
		}
		__syncthreads();
		// the first three levels of both 7-level searches are decided against pivots held in registers (every 16th entry: 8 independent
		// compares instead of 3 dependent LDS round trips); the last four levels walk the 16-entry block in LDS
		uint32_t pthr[8];
		uint64_t pT[7];
#pragma unroll
		for (int k = 0; k < 8; ++k) pthr[k] = s_thr[16 * k + 15];
#pragma unroll
		for (int k = 0; k < 7; ++k) pT[k] = s_T[16 * (k + 1)];
		// ---- values (buffered, then new): gap = #{clusters with mean <= v} = #{thresholds <= v}, two values per lane and iteration so that
		// the two dependent LDS searches overlap
		{
			const uint32_t *pend = p.td_pend + (size_t)slot * GYS_TD_PEND_CAP;
			for (uint32_t base = 0; base < m; base += 128u) {
				uint32_t uv[2], lo[2];
#pragma unroll
				for (int u = 0; u < 2; ++u) {
					const uint32_t i = base + lane + 64u * u;
					uv[u] = 0;
					if (i < m) uv[u] = i < npend ? pend[i] : (p.staged[start + (i - npend)] >> GYS_ROW_BITS);
					uint32_t blocks = 0; // 16-entry blocks that lie entirely at or below the value (the thresholds ascend)
#pragma unroll
					for (int k = 0; k < 8; ++k) blocks += pthr[k] <= uv[u] ? 1u : 0u;
					lo[u] = 16u * blocks;
				}
#pragma unroll
				for (uint32_t step = 8u; step >= 1u; step >>= 1) {
#pragma unroll
					for (int u = 0; u < 2; ++u)
						if (lo[u] < 128u && s_thr[lo[u] + step - 1u] <= uv[u]) lo[u] += step;
				}
#pragma unroll
				for (int u = 0; u < 2; ++u) {
					const uint32_t i = base + lane + 64u * u;
					if (i >= m) continue;
					// refined interval: the cluster means AND a fixed quarter-octave value grid cut the axis (both monotone in v, so
					// their sum numbers the cells of the common refinement in value order); the grid bounds a cell's population
					// when the digest is still empty or the distribution has moved away from its clusters
					const uint32_t iv = lo[u] + value_grid(uv[u]);
					s_x[i] = (iv << 20) | uv[u];
					atomicAdd(&s_icnt[iv], 1u);
					atomicMin(&s_imin[iv], uv[u]);
					atomicMax(&s_imax[iv], uv[u]);
					atomicAdd(&s_clt[lo[u] + 1], 1u);
				}
			}
		}
		__syncthreads();
		// ---- exclusive scan of the refined-interval counts (GYS_IVL = 3 x 64 entries: 3 consecutive per lane) and inclusive scan of
		// the per-cluster-gap counts (s_clt[c + 1] := values in cluster gaps 0..c = values below the mean of cluster c)
		{
			const uint32_t t0 = s_icnt[3u * lane], t1 = s_icnt[3u * lane + 1u], t2 = s_icnt[3u * lane + 2u];
			const uint32_t own = t0 + t1 + t2;
			uint32_t inc = own;
			uint32_t g0 = s_clt[lane], g1 = j1 < GYS_TD_NB + 2 ? s_clt[j1] : 0u;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				const uint32_t u = __shfl_up(inc, d, 64), u0 = __shfl_up(g0, d, 64), u1 = __shfl_up(g1, d, 64);
				if ((int)lane >= d) {
					inc += u;
					g0 += u0;
					g1 += u1;
				}
			}
			const uint32_t ex = inc - own;
			s_ioff[3u * lane] = ex;
			s_ioff[3u * lane + 1u] = ex + t0;
			s_ioff[3u * lane + 2u] = ex + t0 + t1;
			if (lane == 63u) s_ioff[GYS_IVL] = inc;
			s_icnt[3u * lane] = ex; // scatter cursors
			s_icnt[3u * lane + 1u] = ex + t0;
			s_icnt[3u * lane + 2u] = ex + t0 + t1;
			const uint32_t gt0 = __shfl(g0, 63, 64);
			s_clt[lane] = g0;
			if (j1 < GYS_TD_NB + 2) s_clt[j1] = gt0 + g1;
		}
		__syncthreads();
		for (uint32_t i = lane; i < m; i += 64u) {
			const uint32_t x = s_x[i];
			s_g[atomicAdd(&s_icnt[x >> 20], 1u)] = x;
		}
		__syncthreads();
		// ---- old clusters (this lane's two entries, from registers): preceded by the old weight before them and by the values of
		// gaps 0..c (= values below the mean)
		{
			uint64_t mid2[2] = {0, 0};
			if (c0) mid2[0] = 2ull * (e0 + (uint64_t)s_clt[pos0 + 1]) + (uint64_t)c0;
			if (c1) mid2[1] = 2ull * (e1 + (uint64_t)s_clt[pos1 + 1]) + (uint64_t)c1;
			uint32_t a[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				uint32_t blocks = 0;
#pragma unroll
				for (int k = 0; k < 7; ++k) blocks += mid2[u] >= pT[k] ? 1u : 0u;
				a[u] = 16u * blocks;
			}
#pragma unroll
			for (uint32_t step = 8u; step >= 1u; step >>= 1) {
#pragma unroll
				for (int u = 0; u < 2; ++u)
					if (mid2[u] >= s_T[a[u] + step]) a[u] += step;
			}
			if (c0) {
				atomicAdd(&s_osum[a[0]], (unsigned long long)sm0);
				atomicAdd(&s_ocnt[a[0]], c0);
			}
			if (c1) {
				atomicAdd(&s_osum[a[1]], (unsigned long long)sm1);
				atomicAdd(&s_ocnt[a[1]], c1);
			}
		}
		// ---- values: rank = values in lower intervals + rank inside the interval (ties by position); W adds the old weight <= v
		for (uint32_t base = 0; base < m; base += 128u) {
			uint32_t x[2];
			uint64_t mid2[2];
			bool live[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				const uint32_t e = base + lane + 64u * u;
				live[u] = e < m;
				x[u] = live[u] ? s_g[e] : 0u;
				const uint32_t iv = x[u] >> 20;
				uint32_t r = e; // an interval of equal values (the usual case for integer ms data): ties rank by position
				if (live[u] && s_imin[iv] != s_imax[iv]) {
					const uint32_t gb = s_ioff[iv], ge = s_ioff[iv + 1];
					r = gb;
					for (uint32_t t = gb; t < ge; ++t) {
						const uint32_t y = s_g[t];
						r += (y < x[u] || (y == x[u] && t < e)) ? 1u : 0u;
					}
				}
				mid2[u] = 2ull * ((uint64_t)r + s_cpfx[iv - value_grid(x[u] & 0xFFFFFu)]) + 1ull; // old weight with mean <= v
			}
			uint32_t a[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				uint32_t blocks = 0;
#pragma unroll
				for (int k = 0; k < 7; ++k) blocks += mid2[u] >= pT[k] ? 1u : 0u;
				a[u] = 16u * blocks;
			}
#pragma unroll
			for (uint32_t step = 8u; step >= 1u; step >>= 1) {
#pragma unroll
				for (int u = 0; u < 2; ++u)
					if (mid2[u] >= s_T[a[u] + step]) a[u] += step;
			}
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				if (!live[u]) continue;
				atomicAdd(&s_osum[a[u]], (unsigned long long)(x[u] & 0xFFFFFu));
				atomicAdd(&s_ocnt[a[u]], 1u);
			}
		}
		__syncthreads();
		// ---- write back
		const bool query = q.out_sum != nullptr;
		int64_t *ws = query ? q.out_sum + (size_t)w * GYS_TD_NB : p.td_sum + (size_t)slot * GYS_TD_NB;
		uint32_t *wc = query ? q.out_cnt + (size_t)w * GYS_TD_NB : p.td_cnt + (size_t)slot * GYS_TD_NB;
		ws[lane] = (int64_t)s_osum[lane];
		wc[lane] = s_ocnt[lane];
		if (j1 < GYS_TD_NB) {
			ws[j1] = (int64_t)s_osum[j1];
			wc[j1] = s_ocnt[j1];
		}
		if (lane == 0 && !query) p.td_meta[slot].npend = 0;
		__syncthreads();
	}
}

// ---------------------------------------------------------------------------------------------------- t-digest merge (huge)
// One 256-thread workgroup per huge key (persistent over the work list).  Each workgroup owns a 2^20-bin u32 count array in HBM
// scratch: values are histogrammed exactly, the bins are prefix-scanned, and every bin's rank interval is intersected with the
// cluster rank intervals -- the same exact-integer assignment as k_digest_merge without materialising a sort.
struct HugeP {
	DigestP d;
	const uint32_t *huge_list;
	const uint32_t *huge_count;
	uint32_t *scratch; // [gridDim.x * GYS_HUGE_BINS]
};

__global__ __launch_bounds__(256) void k_digest_huge(HugeP p)
{
	__shared__ int64_t s_csum[GYS_TD_NB];
	__shared__ uint32_t s_ccnt[GYS_TD_NB];
	__shared__ uint64_t s_cpfx[GYS_TD_NB + 1];
	__shared__ uint64_t s_T[GYS_TD_NB + 1];
	__shared__ unsigned long long s_osum[GYS_TD_NB];
	__shared__ unsigned long long s_ocnt[GYS_TD_NB];
	__shared__ uint32_t s_part[256];
	__shared__ uint32_t s_wave[4];
	__shared__ unsigned long long s_h[32];
	__shared__ uint32_t s_bm[16];
	__shared__ uint32_t s_nc;
	__shared__ int32_t s_min, s_max;
	uint32_t *bins = p.scratch + (size_t)blockIdx.x * GYS_HUGE_BINS;
	const uint32_t nh = *p.huge_count;
	const uint32_t BPT = GYS_HUGE_BINS / 256u; // bins per thread (contiguous)

	for (uint32_t w = blockIdx.x; w < nh; w += gridDim.x) {
		const uint32_t slot = p.huge_list[w];
		const uint32_t m = p.d.batch_cnt[slot];
		const uint32_t start = p.d.off_end[slot] - m;
		const uint32_t npend = p.d.td_meta[slot].npend; // buffered values join the merge (npend + m > CAP always holds here)
		// zero the bins (16-byte stores)
		for (uint32_t i = threadIdx.x; i < GYS_HUGE_BINS / 4u; i += 256u) ((uint4 *)bins)[i] = make_uint4(0, 0, 0, 0);
		if (threadIdx.x < GYS_TD_NB) {
			s_osum[threadIdx.x] = 0;
			s_ocnt[threadIdx.x] = 0;
		}
		if (threadIdx.x >= 128u && threadIdx.x < 160u) s_h[threadIdx.x - 128u] = 0;
		if (threadIdx.x >= 160u && threadIdx.x < 176u) s_bm[threadIdx.x - 160u] = 0;
		if (threadIdx.x == 0) {
			// compact non-empty old clusters (serial: 100 entries, once per huge key)
			const int64_t *gs = p.d.td_sum + (size_t)slot * GYS_TD_NB;
			const uint32_t *gc = p.d.td_cnt + (size_t)slot * GYS_TD_NB;
			uint32_t nc = 0;
			uint64_t run = 0;
			for (uint32_t j = 0; j < GYS_TD_NB; ++j) {
				if (gc[j]) {
					s_csum[nc] = gs[j];
					s_ccnt[nc] = gc[j];
					s_cpfx[nc] = run;
					run += gc[j];
					nc++;
				}
			}
			s_cpfx[nc] = run;
			s_nc = nc;
			s_min = INT32_MAX;
			s_max = INT32_MIN;
		}
		__syncthreads();
		const uint32_t nc = s_nc;
		const uint64_t nold = s_cpfx[nc];
		const uint64_t twoN = 2ull * (nold + (uint64_t)m + (uint64_t)npend);
		if (threadIdx.x >= 1 && threadIdx.x < GYS_TD_NB) s_T[threadIdx.x] = td_threshold(c_td_bnd[threadIdx.x], twoN);
		if (threadIdx.x == 0) s_T[GYS_TD_NB] = ~0ull;
		// exact value histogram
		{
			int32_t lmin = INT32_MAX, lmax = INT32_MIN;
			for (uint32_t i = threadIdx.x; i < m; i += 256u) {
				const uint32_t w = p.d.staged[start + i];
				const uint32_t v = (w >> GYS_ROW_BITS) & (GYS_HUGE_BINS - 1u);
				atomicAdd(&bins[v], 1u);
				{
					const uint32_t row = w & 0x1Fu;
					const uint32_t bit = (1u << resp_bucket((int64_t)v)) << ((row & 1u) * 16u);
					if ((s_bm[row >> 1] & bit) == 0) atomicOr(&s_bm[row >> 1], bit);
				}
				lmin = min(lmin, (int32_t)v);
				lmax = max(lmax, (int32_t)v);
			}
			atomicMin(&s_min, lmin);
			atomicMax(&s_max, lmax);
			for (uint32_t i = threadIdx.x; i < npend; i += 256u)
				atomicAdd(&bins[p.d.td_pend[(size_t)slot * GYS_TD_PEND_CAP + i] & (GYS_HUGE_BINS - 1u)], 1u);
		}
		__syncthreads();
		// the atomics above were performed in L2; drop this CU's L1 copies of the bins before reading them with plain loads
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		// per-thread partial sums over its contiguous bins, block exclusive scan
		uint32_t part = 0;
		{
			const uint4 *b4 = (const uint4 *)(bins + threadIdx.x * BPT);
			for (uint32_t i = 0; i < BPT / 4u; ++i) {
				const uint4 v = b4[i];
				part += v.x + v.y + v.z + v.w;
			}
		}
		uint32_t tot;
		const uint32_t pfx = block_exclusive_scan_256(part, s_wave, &tot);
		s_part[threadIdx.x] = pfx;
		__syncthreads();
		// ---- old clusters: lt = #{new v : v * cc < cs} = #{v <= (cs - 1) / cc}  (cs >= 1; none when cs <= 0)
		if (threadIdx.x < nc) {
			const int64_t cs = s_csum[threadIdx.x];
			const uint32_t cc = s_ccnt[threadIdx.x];
			uint64_t lt = 0;
			if (cs > 0) {
				int64_t vmax = (cs - 1) / (int64_t)cc;
				if (vmax >= (int64_t)GYS_HUGE_BINS) vmax = GYS_HUGE_BINS - 1;
				const uint32_t owner = (uint32_t)vmax / BPT;
				lt = s_part[owner];
				for (uint32_t b = owner * BPT; b <= (uint32_t)vmax; ++b) lt += bins[b];
			}
			const uint64_t mid2 = 2ull * (s_cpfx[threadIdx.x] + lt) + (uint64_t)cc;
			const uint32_t cl = td_cluster_of(s_T, mid2);
			atomicAdd(&s_osum[cl], (unsigned long long)cs);
			atomicAdd(&s_ocnt[cl], (unsigned long long)cc);
		}
		// ---- new values bin by bin: ranks [r0, r0 + c) of value v, le = old weight with mean <= v
		{
			uint64_t r0 = pfx;
			uint32_t ci = 0; // first compacted cluster with mean > v; monotone in v, so carried along the thread's bins
			const uint32_t vbeg = threadIdx.x * BPT;
			{
				uint32_t lo = 0, hi = nc;
				const int64_t v = (int64_t)vbeg;
				while (lo < hi) {
					const uint32_t mid = (lo + hi) >> 1;
					if (s_csum[mid] <= v * (int64_t)s_ccnt[mid]) lo = mid + 1; else hi = mid;
				}
				ci = lo;
			}
			for (uint32_t b = vbeg; b < vbeg + BPT; ++b) {
				const uint32_t c = bins[b];
				if (!c) continue;
				const int64_t v = (int64_t)b;
				{
					const uint32_t hb = resp_bucket(v); // exact histogram delta of the key from the value counts
					atomicAdd(&s_h[2 * hb], (unsigned long long)c);
					atomicAdd(&s_h[2 * hb + 1], (unsigned long long)((uint64_t)c * (uint64_t)v));
				}
				while (ci < nc && s_csum[ci] <= v * (int64_t)s_ccnt[ci]) ci++;
				const uint64_t le = s_cpfx[ci];
				const uint64_t first = 2ull * (r0 + le) + 1ull, last = first + 2ull * (uint64_t)(c - 1u);
				uint32_t cl = td_cluster_of(s_T, first);
				const uint32_t cl_last = td_cluster_of(s_T, last);
				uint64_t rbeg = r0;
				for (; cl <= cl_last; ++cl) {
					// ranks r with cluster == cl: mid2(r) < T[cl+1]  <=>  r < ceil((T - 1) / 2) - le   (T = T[cl+1] > 1 here)
					uint64_t rend;
					if (cl == cl_last) {
						rend = r0 + c;
					} else {
						const uint64_t Tn = s_T[cl + 1];
						rend = (Tn / 2ull) - le; // ceil((Tn-1)/2) == floor(Tn/2)
						if (rend > r0 + c) rend = r0 + c;
					}
					if (rend > rbeg) {
						const uint64_t k = rend - rbeg;
						atomicAdd(&s_osum[cl], (unsigned long long)(k * (uint64_t)v));
						atomicAdd(&s_ocnt[cl], (unsigned long long)k);
						rbeg = rend;
					}
				}
				r0 += c;
			}
		}
		__syncthreads();
		// the value counts included the buffered values: their histogram contribution was already applied when they were appended
		for (uint32_t i = threadIdx.x; i < npend; i += 256u) {
			const uint64_t pv = (uint64_t)(p.d.td_pend[(size_t)slot * GYS_TD_PEND_CAP + i] & (GYS_HUGE_BINS - 1u));
			const uint32_t hb = resp_bucket((int64_t)pv);
			atomicAdd(&s_h[2 * hb], ~0ull);          // -1
			atomicAdd(&s_h[2 * hb + 1], 0ull - pv);  // -value
		}
		__syncthreads();
		if (threadIdx.x < GYS_TD_NB) {
			p.d.td_sum[(size_t)slot * GYS_TD_NB + threadIdx.x] = (int64_t)s_osum[threadIdx.x];
			p.d.td_cnt[(size_t)slot * GYS_TD_NB + threadIdx.x] = (uint32_t)s_ocnt[threadIdx.x];
		}
		{
			// the key's batch deltas: s_h[2b] = count, s_h[2b+1] = sum of bucket b, m new values, s_max = largest; the key is owned by this
			// workgroup for the batch, so the records are updated with plain read-modify-writes (lazy window roll as in k_key_pass)
			const bool stale = p.d.td_meta[slot].win_epoch != p.d.epoch;
			const uint32_t t = threadIdx.x - 128u; // lanes 0..15: histogram pairs, 16..19: Count-Min rows, 20..35: CONN_BITMAP words
			if (threadIdx.x >= 128u && t < 16u) {
				gys_hist_serial *wp = (gys_hist_serial *)&p.d.hist_win[slot] + t, *ap = (gys_hist_serial *)&p.d.hist_all[slot] + t;
				gys_hist_serial wv = *wp;
				if (stale) { // all-time += old window record
					gys_hist_serial av = *ap;
					if (t < 15u) {
						av.count += wv.count;
						av.sum += wv.sum;
					} else {
						av.count += wv.count;
						if (av.sum < wv.sum) av.sum = wv.sum;
					}
					*ap = av;
					wv.count = 0;
					wv.sum = t < 15u ? 0 : INT64_MIN;
				}
				if (t < 15u) {
					wv.count += s_h[2 * t];
					wv.sum += (int64_t)s_h[2 * t + 1];
					if (s_h[2 * t]) {
						atomicAdd(&p.d.ghist[2 * t], s_h[2 * t]);
						atomicAdd(&p.d.ghist[2 * t + 1], s_h[2 * t + 1]);
					}
				} else {
					wv.count += m; // total_count_
					if (wv.sum < (int64_t)s_max) wv.sum = (int64_t)s_max; // max_val_seen_
					atomicAdd(&p.d.ghist[30], (unsigned long long)m);
					atomicMax(p.d.gmax, (long long)s_max);
				}
				*wp = wv;
			} else if (threadIdx.x >= 128u && t < 20u) {
				const uint32_t r = t - 16u;
				atomicAdd(&p.d.cms32[r * GYS_CMS_W + (jhash2_u64(p.d.svc_gid[slot], GYS_SEED + r) & (GYS_CMS_W - 1))], m);
			} else if (threadIdx.x >= 128u && t < 36u) {
				uint32_t *bp = &p.d.bitmap[(size_t)slot * 16u + (t - 20u)];
				*bp = (stale ? 0u : *bp) | s_bm[t - 20u];
			}
		}
		__syncthreads(); // every reader of win_epoch is done before thread 0 rewrites the meta record
		if (threadIdx.x == 0) {
			TdMeta *mt = &p.d.td_meta[slot];
			if (s_min < mt->vmin) mt->vmin = s_min;
			if (s_max > mt->vmax) mt->vmax = s_max;
			mt->npend = 0;
			mt->win_epoch = p.d.epoch;
			p.d.batch_cnt[slot] = 0;
		}
		__syncthreads();
	}
}

// one device atomic per WAVE instead of one per record on the shared statistics counters (a single hot address serialises in L2)
__device__ __forceinline__ void wave_count(uint64_t *ctr, bool pred)
{
	const unsigned long long b = __ballot(pred);
	if (b && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)b) - 1)) atomicAdd((unsigned long long *)ctr, (unsigned long long)__popcll(b));
}

// ---------------------------------------------------------------------------------------------------- TCP_CONN_NOTIFY ingest
// The kernel is bound by the chip's device-atomic rate (~30 G/s), so a record of a KNOWN service costs three atomics only: the
// service's window accumulators {connections | closed << 32, bytes sent, bytes received}.  Both Count-Min tables are linear in the
// per-service sums, so their 8 updates per record are replaced by 8 per ACTIVE SERVICE at the window boundary (k_conn_fold), where
// the accumulators also fold into the cumulative per-service counters.  Records of services the engine was never told about (rare)
// still update the Count-Min tables directly.
// comm::TCP_CONN_NOTIFY (common/gy_comm_proto.h:1665-1742), 280 fixed bytes:
//   IP_PORT cli_@0 ser_@32 nat_cli_@64 nat_ser_@96 (each: ip128 @0, ip32 @16, aftype @20, flags @22, port @24)
//   tusec_start_@128 tusec_close_@136 ... ser_glob_id_@192 ... bytes_sent_@208 bytes_rcvd_@216 ... cli_cmdline_len_@272 flags@274.. padding_len_@279
struct ConnP {
	const uint8_t *batch;
	const uint32_t *offsets;
	uint32_t n;
	DevTable gid;
	uint32_t *hll32;
	uint32_t *cms32;
	unsigned long long *cms64;
	unsigned long long *svc_win; // [nsvc*3] window accumulators: nconn | nclose << 32, bytes_sent, bytes_rcvd
	uint64_t *counters;
};

__global__ __launch_bounds__(256) void k_conn_ingest(ConnP p)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	wave_count(&p.counters[CTR_CONN_EVENTS], i < p.n);
	if (i >= p.n) return;
	const uint8_t *rec = p.batch + p.offsets[i];
	uint32_t c128[4], s128[4], c32, s32;
	uint16_t cport, sport;
	// flow key: PAIR_IP_PORT(nat_cli_, nat_ser_)  (server/gy_mconnhdlr.cc:8707)
	{
		// records start 8-byte aligned (COMM_HEADER::validate common/gy_comm_proto.cc:23-26) -> use 8-byte loads
		const uint64_t *q = (const uint64_t *)(rec + 64);
		const uint64_t a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
		c128[0] = (uint32_t)a0; c128[1] = (uint32_t)(a0 >> 32); c128[2] = (uint32_t)a1; c128[3] = (uint32_t)(a1 >> 32);
		c32 = (uint32_t)a2;
		cport = (uint16_t)a3;
		const uint64_t *r = (const uint64_t *)(rec + 96);
		const uint64_t b0 = r[0], b1 = r[1], b2 = r[2], b3 = r[3];
		s128[0] = (uint32_t)b0; s128[1] = (uint32_t)(b0 >> 32); s128[2] = (uint32_t)b1; s128[3] = (uint32_t)(b1 >> 32);
		s32 = (uint32_t)b2;
		sport = (uint16_t)b3;
	}
	const uint64_t tusec_close = *(const uint64_t *)(rec + 136);
	const uint64_t ser_glob_id = *(const uint64_t *)(rec + 192);
	const uint64_t bytes_sent = *(const uint64_t *)(rec + 208), bytes_rcvd = *(const uint64_t *)(rec + 216);

	uint32_t w[10];
	const uint32_t nw = pair_words(c32, c128, cport, s32, s128, sport, w);
	const uint64_t h64 = hash64<10>(w, nw);
	uint32_t idx, rank;
	hll_idx_rank(h64, GYS_HLL_P, &idx, &rank);
	if (p.hll32[idx] < rank) atomicMax(&p.hll32[idx], rank);

	const uint32_t slot = tbl_lookup(p.gid, ser_glob_id);
	if (slot == GYS_NOSLOT) {
		atomicAdd((unsigned long long *)&p.counters[CTR_CONN_UNKNOWN], 1ull);
#pragma unroll
		for (uint32_t r = 0; r < GYS_CMS_D; ++r) {
			const uint32_t col = jhash2_u64(ser_glob_id, GYS_SEED + r) & (GYS_CMS_W - 1);
			atomicAdd(&p.cms32[r * GYS_CMS_W + col], 1u);
			atomicAdd(&p.cms64[r * GYS_CMS_W + col], (unsigned long long)(bytes_sent + bytes_rcvd));
		}
		return;
	}
	unsigned long long *c = p.svc_win + (size_t)slot * 3;
	atomicAdd(&c[0], 1ull + (tusec_close ? (1ull << 32) : 0ull)); // a window's connection count of one service stays far below 2^32
	if (bytes_sent) atomicAdd(&c[1], (unsigned long long)bytes_sent);
	if (bytes_rcvd) atomicAdd(&c[2], (unsigned long long)bytes_rcvd);
}

// window boundary (and counter exports): cumulative per-service counters += window accumulators; Count-Min rows of the service +=
// (connections, bytes) of the window; accumulators cleared
__global__ __launch_bounds__(256) void k_conn_fold(unsigned long long *svc_win, unsigned long long *svc_ctr, const uint64_t *svc_gid, uint32_t nsvc, uint32_t *cms32,
						   unsigned long long *cms64)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nsvc) return;
	unsigned long long *w = svc_win + (size_t)s * 3;
	const unsigned long long cnt = w[0], sent = w[1], rcvd = w[2];
	if (!(cnt | sent | rcvd)) return;
	const unsigned long long nconn = cnt & 0xFFFFFFFFull, nclose = cnt >> 32;
	unsigned long long *c = svc_ctr + (size_t)s * 4;
	c[0] += nconn;
	c[1] += nclose;
	c[2] += sent;
	c[3] += rcvd;
	const uint64_t gid = svc_gid[s];
#pragma unroll
	for (uint32_t r = 0; r < GYS_CMS_D; ++r) {
		const uint32_t col = jhash2_u64(gid, GYS_SEED + r) & (GYS_CMS_W - 1);
		atomicAdd(&cms32[r * GYS_CMS_W + col], (uint32_t)nconn);
		if (sent + rcvd) atomicAdd(&cms64[r * GYS_CMS_W + col], sent + rcvd);
	}
	w[0] = 0;
	w[1] = 0;
	w[2] = 0;
}

// GY_HISTOGRAM::add_data (common/gy_statistics.h:596-623) on a record other threads may be adding to
__device__ __forceinline__ void hist_add_atomic(const HashDef &d, int kind, gys_hist_rec *h, int64_t v)
{
	const uint32_t b = kind == GYS_RESP_TIME_HASH ? resp_bucket(v) : bucket_of(d, v);
	atomicAdd((unsigned long long *)&h->stats[b].count, 1ull);
	atomicAdd((unsigned long long *)&h->stats[b].sum, (unsigned long long)v);
	atomicAdd((unsigned long long *)&h->total_count, 1ull);
	if (h->max_val_seen < v) atomicMax((long long *)&h->max_val_seen, (long long)v);
}

// ---------------------------------------------------------------------------------------------------- LISTENER_STATE_NOTIFY ingest
// comm::LISTENER_STATE_NOTIFY (common/gy_comm_proto.h:2183-2254), 88 fixed bytes: glob_id_@0 nqrys_5s_@8 total_resp_5sec_@12 nconns_@16
// nconns_active_@20 ntasks_@24 p95_5s@28 p95_5min@32 kb_in@36 kb_out@40 ser_errors_@44 cli_errors_@48 ... curr_state_@79 ...
// query_flags_@84 issue_string_len_@85 padding_len_@86
struct LStateP {
	const uint8_t *batch;
	const uint32_t *offsets;
	const uint32_t *host_slot; // per record, or nullptr -> single_host
	uint32_t single_host;
	uint32_t n;
	DevTable gid;
	uint8_t *svc_state; // [nsvc*96]
	int32_t *host_summ; // [nhosts*16] window accumulators (13 used)
	uint32_t epoch;
	uint64_t *counters;
	gys_hist_rec *qps_hist, *act_hist; // per service (levels enabled) or nullptr
};

__global__ __launch_bounds__(256) void k_lstate_ingest(LStateP p)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= p.n) return;
	const uint8_t *rec = p.batch + p.offsets[i];
	const uint64_t *q = (const uint64_t *)rec; // 8-byte aligned records
	uint64_t w[11];
#pragma unroll
	for (int k = 0; k < 11; ++k) w[k] = q[k];
	const uint64_t glob_id = w[0];
	const uint32_t nqrys_5s = (uint32_t)w[1];
	const uint32_t nconns_active = (uint32_t)(w[2] >> 32);
	const uint32_t kb_in = (uint32_t)(w[4] >> 32), kb_out = (uint32_t)w[5], ser_errors = (uint32_t)(w[5] >> 32);
	const uint32_t curr_state = (uint32_t)((w[9] >> 56) & 0xFF);  // byte 79
	const uint32_t query_flags = (uint32_t)((w[10] >> 32) & 0xFF); // byte 84
	atomicAdd((unsigned long long *)&p.counters[CTR_LSTATE_RECORDS], 1ull);

	const uint32_t slot = tbl_lookup(p.gid, glob_id); // listen_tbl_.lookup_single_elem_locked(glob_id, get_uint64_hash(glob_id)) :11183
	if (slot == GYS_NOSLOT) {
		atomicAdd((unsigned long long *)&p.counters[CTR_LSTATE_MISSED], 1ull); // nmissed++ :11185-11188
		return;
	}
	if (query_flags == 0xC0u) { // LISTEN_FLAG_DELETE :11194
		atomicAdd((unsigned long long *)&p.counters[CTR_LSTATE_DELETED], 1ull);
		*(uint32_t *)(p.svc_state + (size_t)slot * 96 + 88) = 0; // state no longer current
		return;
	}
	if (curr_state > 5u) { // :11250-11256
		atomicAdd((unsigned long long *)&p.counters[CTR_LSTATE_ERRORS], 1ull);
		return;
	}
	const uint32_t host = p.host_slot ? p.host_slot[i] : p.single_host;
	int32_t *s = p.host_summ + (size_t)host * 16;
	// LISTEN_SUMM_STATS::update server/gy_msocket.h:853-865 (per-record integer quotient nqrys_5s_/5)
	atomicAdd(&s[curr_state], 1);
	if (nqrys_5s / 5u) atomicAdd(&s[6], (int32_t)(nqrys_5s / 5u));
	if (nconns_active) atomicAdd(&s[7], (int32_t)nconns_active);
	if (kb_in) atomicAdd(&s[8], (int32_t)kb_in);
	if (kb_out) atomicAdd(&s[9], (int32_t)kb_out);
	if (ser_errors) atomicAdd(&s[10], (int32_t)ser_errors);
	atomicAdd(&s[11], 1);
	if (nqrys_5s) atomicAdd(&s[12], 1);
	// MTCP_LISTENER::set_state server/gy_msocket.h:1410-1437: keep the 88-byte record
	uint64_t *d = (uint64_t *)(p.svc_state + (size_t)slot * 96);
#pragma unroll
	for (int k = 0; k < 11; ++k) d[k] = w[k];
	d[11] = (uint64_t)p.epoch | ((uint64_t)host << 32);
	if (p.qps_hist) {
		// the per-listener QPS_HISTOGRAM / ACTIVE_CONN_HISTOGRAM behind LISTENER_DAY_STATS (common/gy_socket_stat.h:548-549, :633-635;
		// one sample per 5-s state record: gy_socket_stat.cc:4109-4127), fed from the record's own nqrys_5s_/5 and nconns_active_
		hist_add_atomic(hash_def(GYS_SEMI_LOG_HASH_LO), GYS_SEMI_LOG_HASH_LO, &p.qps_hist[slot], (int64_t)(int32_t)(nqrys_5s / 5u));
		hist_add_atomic(hash_def(GYS_HASH_1_3000), GYS_HASH_1_3000, &p.act_hist[slot], (int64_t)(int32_t)nconns_active);
	}
}

// ---------------------------------------------------------------------------------------------------- window boundary
struct PrepP {
	const int32_t *host_summ;       // [nhosts*16]
	const gys_host_state *host_state;
	const uint32_t *host_state_epoch;
	const uint32_t *host_cluster;
	uint32_t nhosts;
	uint32_t epoch;
	uint32_t *cluster_state;        // arena [max_clusters*12]
	const uint32_t *hll32;
	uint8_t *hll8;                  // arena
};

// CLUSTER_STATE_ONE::update_from_state server/gy_mconnhdlr.cc:16032-16050 for every host whose host state is current
// (send_cluster_state skips hosts without a recent state, :16068-16070)
__global__ void k_window_prepare(PrepP p)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < (1u << GYS_HLL_P) / 4u) {
		const uint4 v = ((const uint4 *)p.hll32)[i];
		((uint32_t *)p.hll8)[i] = (v.x & 0xFF) | ((v.y & 0xFF) << 8) | ((v.z & 0xFF) << 16) | ((v.w & 0xFF) << 24);
	}
	if (i >= p.nhosts) return;
	if (p.host_state_epoch[i] != p.epoch) return;
	const gys_host_state st = p.host_state[i];
	const int32_t *s = p.host_summ + (size_t)i * 16;
	uint32_t *c = p.cluster_state + (size_t)p.host_cluster[i] * 12;
	atomicAdd(&c[0], 1u);
	if (st.ntasks_issue) { atomicAdd(&c[1], st.ntasks_issue); atomicAdd(&c[2], 1u); }
	if (st.ntasks) atomicAdd(&c[3], st.ntasks);
	if (st.nlisten_issue) { atomicAdd(&c[4], st.nlisten_issue); atomicAdd(&c[5], 1u); }
	if (st.nlisten) atomicAdd(&c[6], st.nlisten);
	atomicAdd(&c[7], (uint32_t)s[6]);
	atomicAdd(&c[8], (uint32_t)((s[8] + s[9]) / 1024));
	if (st.cpu_issue) atomicAdd(&c[9], 1u);
	if (st.mem_issue) atomicAdd(&c[10], 1u);
}

// GY_HISTOGRAM::add_histogram (common/gy_statistics.h:625-660): all += win; win cleared (GY_HISTOGRAM::clear :630-636).
// One thread per 16-byte {count,sum} pair (16 per record), persistent grid.  When ghist != nullptr the kernel also reduces the
// window records into the all-service histogram of the window (arena: 15 x {count,sum} + {total}; max in gmax): LDS accumulation,
// one flush of 32 device atomics per workgroup.
__global__ __launch_bounds__(256) void k_hist_fold(gys_hist_rec *all, gys_hist_rec *win, uint64_t nrec, int clear_win, long long *ghist, long long *gmax)
{
	__shared__ unsigned long long s_g[32];
	__shared__ long long s_gmax;
	if (threadIdx.x < 32) s_g[threadIdx.x] = 0;
	if (threadIdx.x == 0) s_gmax = INT64_MIN;
	__syncthreads();
	const uint64_t npairs = nrec * 16ull, stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npairs; t += stride) {
		const uint32_t k = (uint32_t)(t & 15u);
		long long *a = (long long *)all + t * 2, *w = (long long *)win + t * 2;
		const long long w0 = w[0], w1 = w[1];
		if (k < 15u) {
			if (w0 | w1) {
				a[0] += w0;
				a[1] += w1;
				if (ghist) {
					atomicAdd(&s_g[2 * k], (unsigned long long)w0);
					atomicAdd(&s_g[2 * k + 1], (unsigned long long)w1);
				}
				if (clear_win) {
					w[0] = 0;
					w[1] = 0;
				}
			}
		} else {
			if (w0) {
				a[0] += w0;               // total_count
				if (a[1] < w1) a[1] = w1; // max_val_seen
				if (ghist) {
					atomicAdd(&s_g[30], (unsigned long long)w0);
					atomicMax(&s_gmax, w1);
				}
			} else if (a[1] < w1) {
				a[1] = w1;
			}
			if (clear_win && (w0 || w1 != INT64_MIN)) {
				w[0] = 0;
				w[1] = INT64_MIN;
			}
		}
	}
	__syncthreads();
	if (ghist) {
		if (threadIdx.x < 31 && s_g[threadIdx.x]) atomicAdd((unsigned long long *)&ghist[threadIdx.x], s_g[threadIdx.x]);
		if (threadIdx.x == 31 && s_gmax != INT64_MIN) atomicMax(gmax, s_gmax);
	}
}

// standalone keyed histogram add for any hash kind (rows a1/a2)
__global__ __launch_bounds__(256) void k_hist_add(int kind, gys_hist_rec *hist, uint32_t nkeys, const uint32_t *keyidx, const int32_t *vals, uint64_t n)
{
	const HashDef &d = hash_def(kind);
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const uint32_t k = keyidx[i];
		if (k >= nkeys) continue;
		hist_add_atomic(d, kind, &hist[k], (int64_t)vals[i]);
	}
}

// window / all-time view of the lazily rolled records (see "Lazy window roll"); meta == nullptr: the arrays are kept eagerly
__device__ __forceinline__ gys_hist_rec hist_view(const gys_hist_rec *win, const gys_hist_rec *all, const TdMeta *meta, uint32_t epoch, int which, uint32_t slot)
{
	if (!meta) return which ? all[slot] : win[slot];
	gys_hist_rec r;
	if (which == 0) {
		if (meta[slot].win_epoch == epoch) return win[slot];
		for (int i = 0; i < 15; ++i) {
			r.stats[i].count = 0;
			r.stats[i].sum = 0;
		}
		r.total_count = 0;
		r.max_val_seen = INT64_MIN;
		return r;
	}
	r = all[slot];
	const gys_hist_rec w = win[slot];
	for (int i = 0; i < 15; ++i) {
		r.stats[i].count += w.stats[i].count;
		r.stats[i].sum += w.stats[i].sum;
	}
	r.total_count += w.total_count;
	if (r.max_val_seen < w.max_val_seen) r.max_val_seen = w.max_val_seen;
	return r;
}

__global__ __launch_bounds__(256) void k_hist_view(const gys_hist_rec *win, const gys_hist_rec *all, const TdMeta *meta, uint32_t epoch, int which,
						   uint32_t first, uint32_t n, gys_hist_rec *out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = hist_view(win, all, meta, epoch, which, first + i);
}

__global__ __launch_bounds__(256) void k_hist_percentiles_view(const gys_hist_rec *win, const gys_hist_rec *all, const TdMeta *meta, uint32_t epoch, int which,
								       uint32_t nkeys, const float *pcts, uint32_t npct, int64_t *out)
{
	const HashDef &d = hash_def(GYS_RESP_TIME_HASH);
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= nkeys) return;
	const gys_hist_rec r = hist_view(win, all, meta, epoch, which, k);
	for (uint32_t pi = 0; pi < npct; ++pi) {
		int64_t dv, sum;
		uint64_t cnt;
		hist_percentile(d, r, pcts[pi], &dv, &sum, &cnt);
		out[(size_t)k * npct + pi] = dv;
	}
}

// the per-key percentile scan (GY_HISTOGRAM::get_percentiles for every key, rows a3/a9)
__global__ __launch_bounds__(256) void k_hist_percentiles(int kind, const gys_hist_rec *hist, uint32_t nkeys, const float *pcts, uint32_t npct, int64_t *out)
{
	const HashDef &d = hash_def(kind);
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= (uint64_t)nkeys * npct) return;
	const uint32_t k = (uint32_t)(t / npct), pi = (uint32_t)(t % npct);
	int64_t dv, sum;
	uint64_t cnt;
	hist_percentile(d, hist[k], pcts[pi], &dv, &sum, &cnt);
	out[t] = dv;
}

// ---------------------------------------------------------------------------------------------------- multi-level windows
// The reference keeps, per listener, a TIME_HISTOGRAM<RESP_TIME_HASH, Level_5s_5min_5days_all> (common/gy_statistics.h:1082-1551,
// :2067): per histogram bucket a folly MultiLevelTimeSeries whose 300-s and 5-day levels are rings of 10 {sum,count} buckets that
// are cleared as time advances.  Held that way 10^7 services would cost 2 levels x 10 ring buckets x 256 B of read-modify-write
// traffic per key per window.  A ring level is however just "everything added since the start of its oldest live bucket", and the
// engine already has the cumulative (all-time) record of every key, so it keeps SNAPSHOTS instead: snap[level][j][key] = the
// cumulative record at the most recent start of ring bucket j.  A level at time t is then
//       cumulative(t) - snap[level][(bucket(t) + 1) % 10]
// Snapshots are written only when a bucket boundary is crossed (every 30 s for the 300-s level, every 12 h for the 5-day level),
// as one streaming 16-B-per-lane copy over the records (k_level_roll); nothing is touched per event or per key-window.  Never
// written snapshots are zero == the cumulative record before any data, which is exactly what a young series needs.  Level 0
// ("last 5 seconds") is the engine's tumbling window itself: the record of the window closed last (last[]), kept by the same pass.
// oracle: oracle/gy_oracle_levels.c keeps the rings the way folly does; tests/test_gpu_levels.py compares the two.
struct LevelRollP {
	const gys_hist_rec *win, *all;
	const TdMeta *meta; // nullptr: eagerly kept arrays (win is the closing window of every key, all does not hold it yet)
	uint32_t epoch;     // the window being closed
	uint32_t nsvc;
	gys_hist_rec *snap; // [2][GYS_LEVEL_RING][stride]
	gys_hist_rec *last; // [stride]
	uint64_t stride;
	uint32_t mask[2];   // ring buckets of the 300-s / 5-day level whose start lies in (previous close, this close]
};

// records as 16 x {u64, i64}: lanes 0..14 {count, sum}, lane 15 {total_count, max_val_seen}
__device__ __forceinline__ ulonglong2 pair_add(ulonglong2 a, ulonglong2 b, uint32_t k)
{
	ulonglong2 r;
	r.x = a.x + b.x;
	if (k < 15u)
		r.y = a.y + b.y;
	else
		r.y = (unsigned long long)((long long)a.y < (long long)b.y ? (long long)b.y : (long long)a.y);
	return r;
}

__global__ __launch_bounds__(256) void k_level_roll(LevelRollP p)
{
	const uint64_t npairs = (uint64_t)p.nsvc * 16ull, gstride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npairs; t += gstride) {
		const uint32_t slot = (uint32_t)(t >> 4), k = (uint32_t)(t & 15u);
		const ulonglong2 w = ((const ulonglong2 *)p.win)[t];
		const bool cur = !p.meta || p.meta[slot].win_epoch == p.epoch;
		ulonglong2 closing;
		if (cur) {
			closing = w;
		} else { // the key was not touched in the closing window: win still holds an older, not yet folded window
			closing.x = 0;
			closing.y = k < 15u ? 0ull : (unsigned long long)INT64_MIN;
		}
		((ulonglong2 *)p.last)[t] = closing;
		if (p.mask[0] | p.mask[1]) {
			// cumulative record BEFORE the closing window: its add happens at the close time, i.e. at or after the boundary
			ulonglong2 before = ((const ulonglong2 *)p.all)[t];
			if (!cur) before = pair_add(before, w, k);
			for (int li = 0; li < 2; ++li)
				for (uint32_t m = p.mask[li]; m; m &= m - 1) {
					const uint32_t j = (uint32_t)__builtin_ctz(m);
					((ulonglong2 *)(p.snap + ((uint64_t)li * GYS_LEVEL_RING + j) * p.stride))[t] = before;
				}
		}
	}
}

struct LevelViewP {
	const gys_hist_rec *win, *all;
	const TdMeta *meta;
	uint32_t epoch_open;     // windows before this one have been added (closed)
	uint32_t first, n;
	const gys_hist_rec *sub; // mode 0: snapshot to subtract (nullptr: nothing); mode 2: the last-window records
	int mode;                // 0 cumulative - sub, 1 empty, 2 copy of sub
	gys_hist_rec *out;       // [n]; max_val_seen is the all-time maximum for every level (the reference keeps no per-level maximum)
};

__global__ __launch_bounds__(256) void k_level_view(LevelViewP p)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= (uint64_t)p.n * 16ull) return;
	const uint32_t slot = p.first + (uint32_t)(t >> 4), k = (uint32_t)(t & 15u);
	const uint64_t g = (uint64_t)slot * 16ull + k;
	ulonglong2 cum = ((const ulonglong2 *)p.all)[g];
	const ulonglong2 w = ((const ulonglong2 *)p.win)[g];
	if (p.meta) {
		if (p.meta[slot].win_epoch != p.epoch_open)
			cum = pair_add(cum, w, k); // a closed window that has not been folded yet
		else if (k == 15u && (long long)cum.y < (long long)w.y)
			cum.y = w.y;               // the maximum is reported over everything seen
	} else if (k == 15u && (long long)cum.y < (long long)w.y) {
		cum.y = w.y;
	}
	ulonglong2 r;
	if (p.mode == 0) {
		r = cum;
		if (p.sub) {
			const ulonglong2 s = ((const ulonglong2 *)p.sub)[g];
			r.x -= s.x;
			if (k < 15u) r.y -= s.y;
		}
	} else if (p.mode == 2) {
		r = ((const ulonglong2 *)p.sub)[g];
		if (k == 15u) r.y = cum.y;
	} else {
		r.x = 0;
		r.y = k < 15u ? 0ull : cum.y;
	}
	((ulonglong2 *)p.out)[t] = r;
}

// comm::LISTENER_DAY_STATS (common/gy_comm_proto.h:1620-1632) the way TCP_LISTENER::get_curr_state fills it
// (common/gy_socket_stat.cc:2053-2112): 5-day level count / sum / p95 / p25 of the response histogram (TIME_HISTOGRAM::get_stats),
// p95 / p25 of the QPS and active-connection histograms (GY_HISTOGRAM::get_percentiles, HIST_DATA {95, 25}).
__global__ __launch_bounds__(256) void k_day_stats(const gys_hist_rec *lvl5d, const gys_hist_rec *qps, const gys_hist_rec *act, const uint64_t *svc_gid,
						   uint32_t first, uint32_t n, gys_listener_day_stats *out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const gys_hist_rec r = lvl5d[i];
	gys_listener_day_stats o;
	o.glob_id = svc_gid[first + i];
	int64_t ts = 0;
	for (int b = 0; b < 15; ++b) ts += r.stats[b].sum;
	o.tcount_5d = (int64_t)r.total_count;
	o.tsum_5d = ts;
	const HashDef &dr = hash_def(GYS_RESP_TIME_HASH);
	o.p95_5d_respms = (uint32_t)level_percentile(dr, r, 95.0f);
	o.p25_5d_respms = (uint32_t)level_percentile(dr, r, 25.0f);
	int64_t dv, sum;
	uint64_t cnt;
	const gys_hist_rec q = qps[first + i];
	hist_percentile(hash_def(GYS_SEMI_LOG_HASH_LO), q, 95.0f, &dv, &sum, &cnt);
	o.p95_qps = (uint32_t)dv;
	hist_percentile(hash_def(GYS_SEMI_LOG_HASH_LO), q, 25.0f, &dv, &sum, &cnt);
	o.p25_qps = (uint32_t)dv;
	const gys_hist_rec a = act[first + i];
	hist_percentile(hash_def(GYS_HASH_1_3000), a, 95.0f, &dv, &sum, &cnt);
	o.p95_nactive = (uint32_t)dv;
	hist_percentile(hash_def(GYS_HASH_1_3000), a, 25.0f, &dv, &sum, &cnt);
	o.p25_nactive = (uint32_t)dv;
	out[i] = o;
}

// top-N candidate filter: services of one host whose state is from the last window, with the ranked metric per kind
// (LISTEN_TOPN comparators + admission thresholds server/gy_msocket.h:740-790, server/gy_mconnhdlr.cc:11260-11304)
__global__ __launch_bounds__(256) void k_topn_filter(const uint8_t *svc_state, uint32_t nsvc, uint32_t host, uint32_t epoch, int kind,
						     uint32_t *out_slot, uint64_t *out_metric, uint32_t *out_count, uint32_t cap)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nsvc) return;
	const uint64_t *q = (const uint64_t *)(svc_state + (size_t)s * 96);
	const uint64_t tag = q[11];
	if ((uint32_t)tag != epoch || (uint32_t)(tag >> 32) != host) return;
	const uint32_t nqrys = (uint32_t)q[1], nactive = (uint32_t)(q[2] >> 32), kbin = (uint32_t)(q[4] >> 32), kbout = (uint32_t)q[5];
	const uint32_t delay = (uint32_t)(q[6] >> 32); // tasks_delay_usec_ @52
	const uint32_t state = (uint32_t)((q[9] >> 56) & 0xFF);
	uint64_t metric;
	bool ok;
	switch (kind) {
	case 0: ok = state > 2u; metric = ((uint64_t)state << 32) | delay; break;   // is_issue: curr_state_ > STATE_OK; (state, tasks_delay) order
	case 1: ok = nqrys >= 5u; metric = nqrys; break;
	case 2: ok = nactive >= 1u; metric = nactive; break;
	default: ok = (kbin + kbout) > 0u; metric = (uint64_t)kbin + kbout; break;
	}
	if (!ok) return;
	const uint32_t pos = atomicAdd(out_count, 1u);
	if (pos < cap) {
		out_slot[pos] = s;
		out_metric[pos] = metric;
	}
}

// ---------------------------------------------------------------------------------------------------- wire front-end
// GPU-side decode of variable-stride record chains.  A partha message is [COMM_HEADER 16 B][EVENT_NOTIFY 8 B][records],
// every record's size depends on its own length fields (TCP_CONN_NOTIFY::get_elem_size common/gy_comm_proto.h:1721-1724,
// LISTENER_STATE_NOTIFY::get_elem_size :2229-2232), so the reference walks p += p->get_elem_size() serially
// (server/gy_mconnhdlr.cc:9130, :11175).  Here every 8-byte slot of the buffer computes the record size it WOULD have if a record
// started there (one 8-byte load holds both length fields), the true record starts are then found by pointer doubling from the
// payload starts of all messages at once (log2(2048) rounds), a prefix sum ranks them, and the first nevents_ of each message
// become the offset list the ingest kernels consume.  The per-record checks of TCP_CONN_NOTIFY::validate
// (common/gy_comm_proto.cc:840-881) / LISTENER_STATE_NOTIFY::validate (:955-996) -- element fits, size multiple of 8, nevents_
// records present -- are evaluated on the way.
struct WireMsg {
	uint32_t pay_slot;  // first payload slot (8-byte units from the start of the device buffer)
	uint32_t end_slot;  // one past the last payload slot (COMM_HEADER::get_act_len)
	uint32_t nevents;   // EVENT_NOTIFY::nevents_
	uint32_t out_base;  // first entry of this message in the offset list
	uint32_t kind;      // 0 = TCP_CONN_NOTIFY (280 B fixed), 1 = LISTENER_STATE_NOTIFY (88 B fixed)
	uint32_t pad;
};

__device__ __forceinline__ int wire_find_msg(const WireMsg *msgs, uint32_t nmsgs, uint32_t slot)
{
	uint32_t lo = 0, hi = nmsgs; // last message with pay_slot <= slot
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (msgs[mid].pay_slot <= slot) lo = mid + 1; else hi = mid;
	}
	if (lo == 0) return -1;
	return slot < msgs[lo - 1].end_slot ? (int)(lo - 1) : -1;
}

// next[i] = slot of the record after a record starting at slot i (== i for slots outside any payload and for malformed records);
// rec[i] = 1 when slot i lies inside a payload (a candidate record start), bad[i] = 1 when a record starting there is malformed
__global__ __launch_bounds__(256) void k_wire_next(const uint64_t *buf, const WireMsg *msgs, uint32_t nmsgs, uint32_t nslots, uint32_t *next, uint8_t *flags)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nslots) return;
	const int m = wire_find_msg(msgs, nmsgs, i);
	uint32_t nx = i;
	uint8_t fl = 0;
	if (m >= 0) {
		const WireMsg mm = msgs[m];
		const uint32_t fixed = mm.kind == 0 ? 280u / 8u : 88u / 8u;
		fl = 1; // inside a payload
		if (i + fixed > mm.end_slot) {
			fl |= 2; // truncated fixed part
		} else {
			uint32_t size;
			if (mm.kind == 0) {
				const uint64_t w = buf[i + 272u / 8u]; // cli_cmdline_len_ @272 (u16) ... padding_len_ @279
				size = 280u + (uint32_t)(w & 0xFFFFu) + (uint32_t)(w >> 56);
			} else {
				const uint64_t w = buf[i + 80u / 8u];  // issue_string_len_ @85, padding_len_ @86
				size = 88u + (uint32_t)((w >> 40) & 0xFFu) + (uint32_t)((w >> 48) & 0xFFu);
			}
			if ((size & 7u) || i + size / 8u > mm.end_slot) fl |= 2; // "Padding issue" / element overruns the message
			else nx = i + size / 8u;
		}
	}
	next[i] = nx;
	flags[i] = fl;
}

__global__ __launch_bounds__(256) void k_wire_seed(const WireMsg *msgs, uint32_t nmsgs, uint8_t *mark)
{
	const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
	if (m < nmsgs && msgs[m].nevents && msgs[m].pay_slot < msgs[m].end_slot) mark[msgs[m].pay_slot] = 1;
}

// one doubling round: everything marked marks its 2^k-th successor; jump_out = jump_in o jump_in
__global__ __launch_bounds__(256) void k_wire_round(uint32_t nslots, const uint32_t *jump_in, uint32_t *jump_out, uint8_t *mark)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nslots) return;
	const uint32_t j = jump_in[i];
	if (mark[i] && j != i) mark[j] = 1;
	jump_out[i] = jump_in[j];
}

// cnt[i] = 1 for true record starts (marked slots inside a payload), so that an exclusive scan ranks the records of the whole stream
__global__ __launch_bounds__(256) void k_wire_count(uint32_t nslots, const uint8_t *mark, const uint8_t *flags, uint32_t *cnt)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nslots) cnt[i] = (mark[i] && (flags[i] & 1u)) ? 1u : 0u;
}

// the first nevents_ records of every message go to the offset list; status[0] |= 1 malformed record, 2 fewer records than nevents_
__global__ __launch_bounds__(256) void k_wire_emit(const WireMsg *msgs, uint32_t nmsgs, uint32_t nslots, const uint32_t *cnt, const uint32_t *rank, const uint8_t *flags,
						   uint32_t *offsets, uint32_t *status)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nslots || !cnt[i]) return;
	const int m = wire_find_msg(msgs, nmsgs, i);
	if (m < 0) return;
	const WireMsg mm = msgs[m];
	const uint32_t r = rank[i] - rank[mm.pay_slot];
	if (r >= mm.nevents) return; // the reference stops after nevents_ records
	if (flags[i] & 2u) atomicOr(status, 1u);
	offsets[mm.out_base + r] = i * 8u;
}

__global__ __launch_bounds__(256) void k_wire_check(const WireMsg *msgs, uint32_t nmsgs, uint32_t nslots, const uint32_t *cnt, const uint32_t *rank, uint32_t *status)
{
	const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= nmsgs) return;
	const WireMsg mm = msgs[m];
	if (!mm.nevents) return;
	const uint32_t last = mm.end_slot - 1u; // records found in [pay_slot, end_slot)
	const uint32_t found = mm.end_slot > mm.pay_slot ? rank[last] + cnt[last] - rank[mm.pay_slot] : 0u;
	if (found < mm.nevents) atomicOr(status, 2u);
}

// ---------------------------------------------------------------------------------------------------- synthetic stream generator
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

struct GenP {
	uint64_t *ev;
	uint64_t n;
	uint64_t seed;
	uint32_t first_host, nhosts, svcs_per_host;
	const float *zipf_cdf; // [svcs_per_host] or nullptr (uniform)
	uint32_t spread;       // 1: service s of host h is drawn with weight (hash(h,s) & 255) / 256 (rejection sampling)
	uint64_t per_host;
};

// SURVEY 8d synthetic response events: host h serves svcs_per_host listeners (port 1024 + s % 60000, netns 0xF0000000 + 4h + s / 60000),
// latency ms = floor(min(lognormal(mu_s, 1.5), 1e6)), mu_s ~ N(3,1) per service; clients uniform in 10/8, ports 16000..65535.
__global__ __launch_bounds__(256) void k_gen_resp(GenP g)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += stride) {
		uint32_t hrel = (uint32_t)(i / g.per_host);
		if (hrel >= g.nhosts) hrel = g.nhosts - 1;
		const uint32_t h = g.first_host + hrel;
		const uint64_t r0 = splitmix64(g.seed ^ (i * 0x9E3779B97F4A7C15ull));
		const uint64_t r1 = splitmix64(r0), r2 = splitmix64(r1);
		uint32_t s;
		if (g.zipf_cdf) {
			const float u = (float)(r0 >> 40) * (1.0f / 16777216.0f);
			uint32_t lo = 0, hi = g.svcs_per_host - 1;
			while (lo < hi) {
				const uint32_t mid = (lo + hi) >> 1;
				if (g.zipf_cdf[mid] < u) lo = mid + 1; else hi = mid;
			}
			s = lo;
		} else {
			s = (uint32_t)((r0 >> 32) % g.svcs_per_host);
			if (g.spread) { // per-service weights spread over 0..255/256: used once, before a benchmark, to de-phase the keys' buffers
				uint64_t rr = r0;
				for (int t = 0; t < 16; ++t) {
					const uint32_t wgt = (uint32_t)(splitmix64(((uint64_t)h << 20) + s + 0x7654321ull) & 255u);
					rr = splitmix64(rr + 0x51ull);
					if ((uint32_t)(rr & 255u) < wgt) break;
					s = (uint32_t)((rr >> 32) % g.svcs_per_host);
				}
			}
		}
		// per-service mu ~ N(3,1) from a hash of (h,s); event latency lognormal(mu, 1.5) by Box-Muller
		const uint64_t hs = splitmix64(((uint64_t)h << 20) + s + 0x1234567ull);
		const float u1 = ((float)(hs >> 40) + 0.5f) * (1.0f / 16777216.0f), u2 = (float)((hs >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
		const float mu = 3.0f + sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
		const float v1 = ((float)(r1 >> 40) + 0.5f) * (1.0f / 16777216.0f), v2 = (float)((r1 >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
		const float z = sqrtf(-2.0f * __logf(v1)) * __cosf(6.2831853f * v2);
		float lat = __expf(mu + 1.5f * z);
		if (!(lat < 1.0e6f)) lat = 1.0e6f;
		const uint32_t ms = (uint32_t)lat;
		const uint32_t saddr = __builtin_bswap32(0x0A000000u | (h & 0xFFFFFFu)); // server 10.x.y.z, network order as ip32_be
		const uint32_t daddr = __builtin_bswap32(0x0A000000u | (uint32_t)(r2 & 0xFFFFFFu));
		const uint32_t netns = 0xF0000000u + 4u * h + s / 60000u;
		const uint16_t sport = (uint16_t)(1024u + s % 60000u);
		const uint16_t dport = (uint16_t)(16000u + (uint32_t)((r2 >> 24) % 49536u));
		const uint32_t lrcv = (uint32_t)(r2 >> 40) * 7u;
		const uint32_t lsnd = lrcv + ms;
		g.ev[3 * i] = (uint64_t)saddr | ((uint64_t)daddr << 32);
		g.ev[3 * i + 1] = (uint64_t)netns | ((uint64_t)bswap16(sport) << 32) | ((uint64_t)bswap16(dport) << 48);
		g.ev[3 * i + 2] = (uint64_t)lsnd | ((uint64_t)lrcv << 32);
	}
}

} // namespace gys

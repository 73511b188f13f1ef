// gys_kernels.hpp -- the HIP kernels of libgysketch (gfx950, wave64).  Included once by gys_engine.hip.
//
// Hot path (per response event, replaces common/gy_socket_stat.cc:1517-1677 + common/gy_statistics.h:596-623):
//   resp_host      one workgroup per host segment: 24-B event -> listener (LDS sub-table of the host) -> global HLL, all-service
//                  histogram -> staged word appended to the service's value buffer (tile-wise LDS counting sort by key); one pass
//   key_finalize   per service with new values: meta record, Count-Min, merge queue (buffer above GYS_TD_PEND_CAP) / spill
//   digest_merge   queued keys: exact-integer k-bucket t-digest merge of the buffered values (sort-free, LDS), which also folds the
//                  values into the service's exact histogram record / CONN_BITMAP rows / min-max (lazy fold, see "per-key value buffers")
//   digest_huge    keys with more values than the LDS merge takes: value-count array in HBM scratch + parallel rank-interval assignment
//   fold           brings the records of a slot range up to date before a query / export / level roll reads them
//   resp_pass1 / scan_* / resp_scatter / key_append   general front end (global listener table + device atomics + counting sort by key
//                  through HBM): hosts with more listeners than the LDS path takes, forced by gys_config.resp_path = 1, t-digest off
// All of it is HBM-bound integer work: no MFMA.
#pragma once

#include "gys_device.hpp"

namespace gys {

#define GYS_HUGE_VALUE_BITS 20     // resp values are <= 1,000,000 < 2^20 (drop filter common/gy_socket_stat.cc:1521-1524)
#define GYS_HUGE_BINS (1u << GYS_HUGE_VALUE_BITS)
// staged word of one accepted event: (response ms << 6) | family << 5 | CONN_BITMAP row (cli_port & 0x1F, common/gy_socket_stat.h:403-410).
// family = 1 for an IPv6 response event: the reference keeps resp_bitmap_v4_ and resp_bitmap_v6_ apart (common/gy_socket_stat.cc:1579-1588)
// and adds their per-bucket row counts (:4144-4149), so the 64 "rows" of a service are 32 IPv4 rows + 32 IPv6 rows; the histogram and the
// query count are shared by the families (:1800 both caches flush into resp_hist_, :4050-4051).
// Sorting the words sorts by value; the digest kernels, which see all of a key's words, also produce the key's bitmap rows.
#define GYS_ROW_BITS 6
#define GYS_ROW_MASK 0x3Fu
#define GYS_ROW_V6 0x20u
#define GYS_BM_WORDS 32u // u32 words of CONN_BITMAP rows per service: words 0..15 = the 32 u16 rows of resp_bitmap_v4_, 16..31 = resp_bitmap_v6_
#define GYS_STAGED_WORD(tresp, cli_port) ((uint64_t)(((uint32_t)(tresp) << GYS_ROW_BITS) | ((uint32_t)(cli_port) & 0x1Fu)))
#define GYS_STAGED_WORD_FAM(tresp, cli_port, v6) ((uint32_t)(((uint32_t)(tresp) << GYS_ROW_BITS) | ((v6) ? GYS_ROW_V6 : 0u) | ((uint32_t)(cli_port) & 0x1Fu)))

enum { CTR_RESP_EVENTS = 0, CTR_RESP_DROP_RANGE, CTR_RESP_DROP_NOLISTENER, CTR_CONN_EVENTS, CTR_CONN_UNKNOWN, CTR_LSTATE_RECORDS,
       CTR_LSTATE_MISSED, CTR_LSTATE_ERRORS, CTR_LSTATE_DELETED, CTR_TD_MERGES, CTR_TD_MERGE_VALUES, CTR_ACTCONN_RECORDS, CTR_ACTCONN_REMOTE_LISTEN, CTR_ACTCONN_UNKNOWN,
       CTR_CONN_NEW, CTR_CONN_CLOSED, CTR_CONN_CLOSED_NO_NOTIFY, CTR_CONN_CLI_SIDE, CTR_RESP_RUN_OVERFLOW, CTR_NUM };

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

// kv[2 i] = key, kv[2 i + 1] = value: the key (inserted if new) gets exactly that value
__global__ void k_table_set(DevTable t, const uint64_t *kv, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t key = kv[2u * i];
	uint32_t h = get_uint64_hash(key) & t.mask;
	for (uint32_t probes = 0; probes <= t.mask; ++probes) {
		const unsigned long long prev = atomicCAS((unsigned long long *)&t.ent[h].key, (unsigned long long)GYS_EMPTY_KEY, (unsigned long long)key);
		if (prev == GYS_EMPTY_KEY || prev == key) {
			t.ent[h].val = (uint32_t)kv[2u * i + 1u];
			return;
		}
		h = (h + 1) & t.mask;
	}
}

__global__ void k_fill_u64(uint64_t *p, uint64_t v, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_hist_init(gys_hist_rec *h, uint64_t first, uint64_t n, int64_t minval)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) h[first + i].max_val_seen = minval;
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

// the same for a flow key in the general word form (IPv6 ends: ten words, four mix rounds per hash half -- every IPv6 event pays them):
// == hll_idx_rank(hash64<N>(w, nw), GYS_HLL_P)
template <int N>
__device__ __forceinline__ void words_hll_idx_rank(const uint32_t (&w)[N], uint32_t nw, uint32_t *idx, uint32_t *rank)
{
	const uint32_t hi = jhash2<N>(w, nw, GYS_SEED);
	const uint32_t rest = hi << GYS_HLL_P;
	*idx = hi >> (32 - GYS_HLL_P);
	if (rest) {
		*rank = (uint32_t)__clz((int)rest) + 1u;
		return;
	}
	const uint32_t lo = jhash2<N>(w, nw, GYS_GOLDEN);
	*rank = (32u - GYS_HLL_P) + (lo ? (uint32_t)__clz((int)lo) : 32u) + 1u;
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

// IPv6 response event (tcp_ipv6_resp_event_t, common/gy_ebpf_kernel.h:113-118; ipv6_tuple_t partha/gy_ebpf_kernel_struct.h:37-44), 48 bytes:
// u128 saddr, u128 daddr, u32 netns, u16 sport, u16 dport (network order), u32 lsndtime, u32 lrcvtime = six 8-byte words, the last two
// laid out like words 1 and 2 of the 24-byte IPv4 event.  Flow key as for IPv4: PAIR_IP_PORT(cli = daddr:dport, ser = saddr:sport), each
// address through GY_IP_ADDR(unsigned __int128) (handle_ipv6_resp_event, common/gy_socket_stat.cc:1535-1551).
#define GYS_EV6_WORDS 6u
__device__ __forceinline__ uint64_t flow_hash64_v6(const uint32_t (&d)[4], uint16_t dport, const uint32_t (&sa)[4], uint16_t sport)
{
	uint32_t w[10];
	const uint32_t nw = pair_words(ip6_embedded_v4(d), d, dport, ip6_embedded_v4(sa), sa, sport, w);
	return hash64<10>(w, nw);
}

struct RespP1 {
	const uint64_t *ev;       // 3 x u64 per event (tcp_ipv4_resp_event_t, common/gy_ebpf_kernel.h:106-111); V6: 6 x u64 (tcp_ipv6_resp_event_t :113-118)
	const ListenerCand *cand; // candidate pool: a table value with GYS_SLOT_GROUP indexes it
	uint64_t n;
	const gys_resp_seg *segs; // device copy
	uint32_t nsegs;
	DevTable lk;
	const uint64_t *svc_gid;
	gys_hist_rec *hist_win;
	uint32_t *bitmap;         // [nsvc*GYS_BM_WORDS] u32 = 32 x u16 CONN_BITMAP rows per family
	uint32_t *hll32;          // [1<<14]
	uint32_t *cms32;          // arena [D*W]
	uint32_t *batch_cnt;      // nullptr when the t-digest is off
	uint64_t *ev_kv;          // (slot << 32 | value) per event, ~0 = dropped
	uint64_t *counters;
	uint8_t *svc_hll;
	uint32_t svc_hll_p;
	unsigned long long *ghist; // arena: all-service histogram of the window (t-digest on: accumulated here per event)
	long long *gmax;
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

template <bool V6>
__global__ __launch_bounds__(256) void k_resp_pass1(RespP1 p)
{
	__shared__ unsigned int s_ctr[3];
	__shared__ unsigned long long s_gh[4][16]; // per wave and bucket: count << 40 | sum (a thread sees at most a few thousand events)
	__shared__ int s_gmax;
	if (threadIdx.x < 3) s_ctr[threadIdx.x] = 0;
	if (threadIdx.x < 64) ((unsigned long long *)s_gh)[threadIdx.x] = 0;
	if (threadIdx.x == 0) s_gmax = INT32_MIN;
	__syncthreads();
	const bool fused = p.batch_cnt != nullptr; // the records are then produced per KEY from the buffered values (lazy fold)
	int tmax = INT32_MIN;

	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += stride) {
		// struct ipv4_tuple_t {u32 saddr, daddr, netns; u16 sport, dport;} + u32 lsndtime, lrcvtime  (24 bytes); IPv6: the two addresses
		// are 16 bytes each, the rest has the same layout behind them (48 bytes)
		uint32_t sa6[4] = {0u, 0u, 0u, 0u}, da6[4] = {0u, 0u, 0u, 0u};
		uint64_t w0 = 0, w1, w2;
		if (V6) {
			const uint64_t *e = p.ev + GYS_EV6_WORDS * i;
			const uint64_t a0 = e[0], a1 = e[1], d0 = e[2], d1 = e[3];
			sa6[0] = (uint32_t)a0; sa6[1] = (uint32_t)(a0 >> 32); sa6[2] = (uint32_t)a1; sa6[3] = (uint32_t)(a1 >> 32);
			da6[0] = (uint32_t)d0; da6[1] = (uint32_t)(d0 >> 32); da6[2] = (uint32_t)d1; da6[3] = (uint32_t)(d1 >> 32);
			w1 = e[4];
			w2 = e[5];
		} else {
			w0 = p.ev[3 * i];
			w1 = p.ev[3 * i + 1];
			w2 = p.ev[3 * i + 2];
		}
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
			uint32_t slot = tbl_lookup(p.lk, listener_key(host_slot, netns, sport));
			if (slot != GYS_NOSLOT && (slot & GYS_SLOT_GROUP)) { // the key has candidates: the server address picks the listener (gy_socket_stat.h:708-714)
				uint32_t lc, sl;
				slot = cand_resolve(p.cand, slot & ~GYS_SLOT_GROUP, V6 ? ip6_embedded_v4(sa6) : saddr, sa6, &lc, &sl) ? sl : GYS_NOSLOT;
			}
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
					const uint32_t row = (dport & 0x1Fu) | (V6 ? GYS_ROW_V6 : 0u);
					const uint32_t bit = (1u << b) << ((row & 1u) * 16u);
					uint32_t *wp = &p.bitmap[(size_t)slot * GYS_BM_WORDS + (row >> 1)];
					if ((*wp & bit) == 0) atomicOr(wp, bit);
				}
				if (V6) {
					const uint64_t h64 = flow_hash64_v6(da6, dport, sa6, sport);
					uint32_t idx, rank;
					hll_idx_rank(h64, GYS_HLL_P, &idx, &rank);
					if (p.svc_hll_p) svc_hll_update(p.svc_hll, p.svc_hll_p, slot, h64);
					if (p.hll32[idx] < rank) atomicMax(&p.hll32[idx], rank);
				} else {
					hll_update_event(p.hll32, p.svc_hll, p.svc_hll_p, slot, daddr, dport, saddr, sport);
				}
				if (fused) {
					atomicAdd(&p.batch_cnt[slot], 1u);
					kv = ((uint64_t)slot << 32) | (uint64_t)GYS_STAGED_WORD_FAM(tresp, dport, V6);
					atomicAdd(&s_gh[threadIdx.x >> 6][b], (1ull << 40) | (unsigned long long)tresp);
					tmax = max(tmax, (int)tresp);
				}
			}
		}
		if (p.ev_kv) p.ev_kv[i] = kv;
	}
	if (fused && tmax != INT32_MIN) atomicMax(&s_gmax, tmax);
	__syncthreads();
	if (threadIdx.x < 3 && s_ctr[threadIdx.x]) atomicAdd((unsigned long long *)&p.counters[CTR_RESP_EVENTS + threadIdx.x], (unsigned long long)s_ctr[threadIdx.x]);
	if (fused && threadIdx.x < 15) {
		unsigned long long cnt = 0, sum = 0;
		for (int w = 0; w < 4; ++w) {
			cnt += s_gh[w][threadIdx.x] >> 40;
			sum += s_gh[w][threadIdx.x] & ((1ull << 40) - 1);
		}
		if (cnt) {
			atomicAdd(&p.ghist[2 * threadIdx.x], cnt);
			atomicAdd(&p.ghist[2 * threadIdx.x + 1], sum);
			atomicAdd(&p.ghist[30], cnt);
		}
	}
	if (fused && threadIdx.x == 15 && s_gmax != INT32_MIN) atomicMax(p.gmax, (long long)s_gmax);
}

// ---------------------------------------------------------------------------------------------------- scan of batch counts
#define GYS_SCAN_TILE 4096u // elements per 256-thread block (16 per thread)

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *s_wave, uint32_t *total)
{
	// inclusive wave scan (DPP), then 4 wave totals through LDS
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t inc = wave_incl_scan_u32(v);
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

// writes the exclusive prefix
__global__ __launch_bounds__(256) void k_scan_final(const uint32_t *cnt, uint32_t n, const uint32_t *block_sums, uint32_t *off)
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
		if (base + k < n) off[base + k] = run;
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

// ---------------------------------------------------------------------------------------------------- per-key value buffers
// Every service owns a buffer of `pcap` staged words (td_pend[slot * pcap ..], word = response ms << 6 | family << 5 | CONN_BITMAP row).  A batch's
// accepted events are APPENDED to their key's buffer and nothing else of the key is touched per event: the key's exact histogram
// record, its CONN_BITMAP rows and its min / max are pure functions of the appended values, so they are brought up to date lazily
// ("fold") -- when the buffer is drained by a t-digest merge, before a query / export reads them, or at a window close when the
// multi-level windows need every window's record.  The per-batch cost of a key is then 4 bytes per value plus one 16-byte meta
// record, instead of ~1 KB of record traffic per key and batch.
//   t-digest rule (oracle: gyo_tdb_add_batch): a batch's values of a key are appended when buffered + new <= GYS_TD_PEND_CAP, otherwise
//   ONE merge re-clusters the digest with (buffered + new) values; a key is also re-clustered when buffered + new + new again would
//   exceed GYS_TDIGEST_MERGE_FAST (= merge size class 0), so that whatever a key's rate its merges stay in the fast class.  Physically the new values are always appended first (the buffer has
//   pcap > GYS_TD_PEND_CAP entries); a key whose buffer then holds more than GYS_TD_PEND_CAP values is queued for k_digest_merge.  A key
//   whose batch does not fit the buffer at all ("spilled") gets its batch values as a run in `staged` instead (second pass of
//   k_resp_host over the hosts that have such keys) and is merged from buffer + run.
struct TdMeta {
	uint32_t npend;     // staged words in the key's buffer
	uint16_t nh;        // words [0, nh) are already folded into the histogram records / CONN_BITMAP rows / min-max
	uint16_t nw;        // words [nw, npend) arrived in window win_epoch, words [0, nw) in earlier windows
	uint32_t win_epoch; // window number of the key's latest values (0 = never touched)
	uint32_t hw_epoch;  // window number the hist_win record and the CONN_BITMAP rows belong to
};
static_assert(sizeof(TdMeta) == 16, "TdMeta is read and written as one 16-byte word");
static_assert(GYS_TD_PEND_CAP == GYS_TDIGEST_PEND_CAP, "gysketch.h and gys_tdigest_tbl.h disagree on the t-digest buffer size");
static_assert(GYS_TD_NB == GYS_TDIGEST_NB && GYS_TD_NB <= 256, "the merge kernels hold one cluster per thread of a 256-thread group");

#define GYS_SPILL_BIT 0x80000000u // td_cur[slot]: the key's values of the running batch go to a run in `staged`, not to its buffer
#define GYS_RESPILL_BIT 0x40000000u // ... and that run is filled by the SECOND pass over the host's events (set with GYS_SPILL_BIT at the end of the first)
#define GYS_MERGE_LDS_MAX 16384u  // largest (buffered + run) value count k_digest_merge handles; larger keys go to k_digest_huge
#define GYS_PCAP_MAX 16384u

struct MergeEnt {
	uint32_t slot;
	uint32_t nbuf;     // values taken from the key's buffer
	uint32_t mrun;     // values taken from the key's run staged[off_end - mrun .. off_end) (spilled keys), else 0
	uint32_t off_end;
};

__global__ void k_minmax_init(int2 *mm, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) mm[i] = make_int2(INT32_MAX, INT32_MIN);
}

// ---------------------------------------------------------------------------------------------------- per-key end of batch
// Keys whose value count moved in this batch get their meta record updated (window bookkeeping of the lazy fold), their events added
// to the window's per-service event count (the Count-Min rows are built from those counts once per window, k_cms_partial), and are
// queued for a merge when the buffer holds more than GYS_TD_PEND_CAP values -- one list per merge size class; keys whose batch did
// not fit the buffer are spilled (run allocated in `staged`, host flagged for the second resp pass).  finalize_key is wave-collective:
// the host-local front end calls it from the tail of k_resp_host (the workgroup owns the host's keys), the other front ends through
// k_key_finalize (one thread per service).
#define GYS_MERGE_CLASS0 1024u // largest (buffered + run) value count of merge size class 0 / 1 (class 2: up to GYS_MERGE_LDS_MAX)
#define GYS_MERGE_CLASS1 4096u
static_assert(GYS_TDIGEST_MERGE_FAST == GYS_MERGE_CLASS0, "the early re-clustering rule keeps a key's merges inside merge size class 0 (default buffer size)");
enum { FIN_CLASS0 = 0, FIN_CLASS1, FIN_CLASS2, FIN_HUGE, FIN_RUN_ALLOC, FIN_SLOW, FIN_NCOUNTS }; // FIN_SLOW: k_digest_bins' hand-over list
#define FIN_APPEND 12 // counts[FIN_APPEND]: length of the append list (keys whose predicted run turned out to fit their buffer; 6..10: the large-key path)

struct FinP {
	uint32_t *td_cur;
	TdMeta *td_meta;
	uint32_t nsvc, pcap, epoch;
	uint32_t pend_cap, merge_fast; // the t-digest rule's two numbers (gys_config.td_pend_cap; merge_fast = pend_cap + 128 = the largest merge of size class 0)
	uint32_t *resp_win;        // per service: response events of the open window
	MergeEnt *list[4];         // merge lists by size class, [FIN_HUGE] = the huge list
	uint32_t *counts;          // [FIN_NCOUNTS]: list lengths, bump cursor into `staged`
	uint32_t *td_run;
	const uint32_t *batch_off; // general front end: end of the key's run in `staged` (nullptr: host-local front end)
	const uint32_t *svc_host;
	uint32_t *host_spill;
	uint32_t spill_stamp;
	uint64_t *counters;
	// predicted runs (round 4; nullptr = off): a key whose LAST batch alone would overflow its buffer again gets a run BEFORE the event pass
	// (k_prespill), so that the pass writes its values once -- a stationary heavy key (a Zipf head) no longer costs a second walk of its
	// host's events.  td_run0 / td_run1: start and end of the key's predicted run; td_prevm: values the key's last batch brought;
	// hot[hot_wr]: set when some key's batch was large enough to be predicted next time (read by the next k_prespill);
	// append_list: keys whose predicted run has to be copied into the buffer after all (the batch was smaller than predicted)
	const uint32_t *td_run0, *td_run1;
	uint32_t *td_prevm, *hot;
	uint32_t hot_wr;
	MergeEnt *append_list;
	uint32_t class2_max;       // largest (buffered + run) value count of merge size class 2 (the streamed value-bin instance); 0: none -- everything above class 1 takes the several-workgroup path
	uint32_t append_cap;       // entries of append_list (one per service: a key has at most one entry per batch); past it the thread copies the run itself
	uint32_t staged_cap;       // words of `staged`: an exact run must end inside it
	uint32_t *td_pend;
	const uint32_t *staged;
};

// the per-key part: meta record, window event count, spill; returns the merge size class the key is queued for (-1: none) and its entry
template <bool WRITE_CUR>
__device__ __forceinline__ int finalize_one(const FinP &p, bool valid, uint32_t key, uint32_t cur, MergeEnt &ent)
{
	int cls = -1;
	ent = MergeEnt{};
	if (valid) {
		const uint4 mraw = *(const uint4 *)&p.td_meta[key];
		const uint32_t npend0 = mraw.x;
		// a key with a predicted run (k_prespill set GYS_SPILL_BIT before the event pass): its count is the run's fill, `cur` stayed at
		// npend0 | bit.  The run holds every value only if every piece found room (the cursor counts on past the end, pieces are dropped)
		const bool pre = (cur & GYS_SPILL_BIT) != 0u;
		uint32_t run0 = 0;
		bool run_ok = false;
		if (pre) {
			run0 = p.td_run0[key];
			const uint32_t fill = p.td_run[key];
			run_ok = fill <= p.td_run1[key];
			cur = npend0 + (fill - run0);
			if (fill == run0) {
				p.td_cur[key] = npend0; // predicted, but the batch had nothing for the key ...
				p.td_prevm[key] = 0u;   // ... and is not predicted again on the strength of some earlier batch (td_prevm is only written for keys with values)
			}
		}
		if (cur != npend0) {
			const uint32_t m = cur - npend0;
			uint32_t nh = mraw.y & 0xFFFFu, nw = mraw.y >> 16, win_epoch = mraw.z;
			if (win_epoch != p.epoch) { // first values of the key in this window: everything buffered so far belongs to earlier windows
				nw = npend0;
				win_epoch = p.epoch;
			}
			p.resp_win[key] += m; // the key is owned by this thread for the batch
			if (p.td_prevm) {
				p.td_prevm[key] = m;
				if (m > p.pcap - p.pend_cap) p.hot[p.hot_wr] = 1u; // (same value from every writer)
			}
			ent.slot = key;
			if (cur <= p.pcap && (!pre || run_ok)) {
				if (pre) {
					// predicted too high: the batch fits the buffer after all, and the rule below (append, re-cluster only past
					// GYS_TD_PEND_CAP) is defined on the buffer -- the run's values are copied behind the buffered ones (k_run_append)
					const uint32_t at = atomicAdd(&p.counts[FIN_APPEND], 1u);
					if (at < p.append_cap) {
						p.append_list[at] = MergeEnt{key, npend0, m, run0 + m};
					} else { // (the list has an entry per service: not reached; the copy itself is what must not be lost)
						atomicAdd(&p.counts[FIN_APPEND], 0xFFFFFFFFu); // (- 1: k_run_append walks the list by this count)
						for (uint32_t i = 0; i < m; ++i) p.td_pend[(size_t)key * p.pcap + npend0 + i] = p.staged[run0 + i];
					}
				}
				*(uint4 *)&p.td_meta[key] = make_uint4(cur, nh | (nw << 16), win_epoch, mraw.w);
				if (WRITE_CUR || pre) p.td_cur[key] = cur;
				if (cur > p.pend_cap || cur + m > p.merge_fast) { // (the second: another batch like this one would leave the fast merge class)
					ent.nbuf = cur;
					cls = cur <= p.merge_fast ? FIN_CLASS0 : cur <= GYS_MERGE_CLASS1 ? FIN_CLASS1 : cur <= p.class2_max ? FIN_CLASS2 : FIN_HUGE; // (above class 2: the several-workgroup path)
				}
			} else { // spilled
				*(uint4 *)&p.td_meta[key] = make_uint4(npend0, nh | (nw << 16), win_epoch, mraw.w);
				ent.nbuf = npend0;
				ent.mrun = m;
				if (pre && run_ok) { // the predicted run holds the batch: no second pass for this key
					p.td_cur[key] = npend0 | GYS_SPILL_BIT;
					ent.off_end = run0 + m;
				} else if (p.batch_off) {
					p.td_cur[key] = npend0 | GYS_SPILL_BIT;
					ent.off_end = p.batch_off[key];
				} else { // not predicted, or more values than the predicted run had room for: an exact run, filled by the second pass
					// (the exact runs of a batch take <= n words, n = the batch's events, and the predicted runs end at staged_cap - n (run_limit),
					// with the cursor counting ACCEPTED predictions only: an exact run ends inside `staged`.  Checked all the same -- a run that
					// did not is not written, the key keeps its buffered values and the batch's values of this key are counted as lost.)
					const uint32_t start = atomicAdd(&p.counts[FIN_RUN_ALLOC], m);
					if ((uint64_t)start + m > (uint64_t)p.staged_cap) {
						atomicAdd((unsigned long long *)&p.counters[CTR_RESP_RUN_OVERFLOW], (unsigned long long)m);
						p.td_cur[key] = npend0;
						ent = MergeEnt{};
						return -1;
					}
					p.td_cur[key] = npend0 | GYS_SPILL_BIT | GYS_RESPILL_BIT;
					p.td_run[key] = start;
					ent.off_end = start + m;
					p.host_spill[p.svc_host[key]] = p.spill_stamp;
				}
				const uint64_t tot = (uint64_t)npend0 + m;
				cls = tot <= p.merge_fast ? FIN_CLASS0 : tot <= GYS_MERGE_CLASS1 ? FIN_CLASS1 : tot <= p.class2_max ? FIN_CLASS2 : FIN_HUGE;
			}
		}
	}
	return cls;
}

// Workgroup-collective end of batch for up to KMAX keys per thread.  The merge lists' cursors and the two statistics counters are ONE
// address each for the whole chip: device-scope atomics on one address execute one after the other, ~10 ns apiece (measured on the
// connection path, r3w: 2.6 x 10^5 per-wave adds on one counter were 2.4 of that kernel's 3.3 ms).  Round 2 reserved list places with one
// returning atomic per WAVE and class (1.6 x 10^5 per 2^29-event batch on each of three addresses); now places are counted in LDS and
// the workgroup takes its share of a list with one atomic per class (10^4 per batch), the statistics likewise.
struct FinWg {
	uint32_t cnt[4], base[4], nmerge;
	unsigned long long nvals;
};

template <bool WRITE_CUR, int KMAX>
__device__ __forceinline__ void finalize_keys_wg(const FinP &p, FinWg *sf, uint32_t tid, uint32_t lane, const bool (&valid)[KMAX], const uint32_t (&key)[KMAX],
						 const uint32_t (&cur)[KMAX])
{
	if (tid < 4u) sf->cnt[tid] = 0;
	if (tid == 4u) {
		sf->nmerge = 0;
		sf->nvals = 0;
	}
	__syncthreads();
	int cls[KMAX];
	MergeEnt ent[KMAX];
	uint32_t pos[KMAX];
	const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
	for (int j = 0; j < KMAX; ++j) {
		cls[j] = finalize_one<WRITE_CUR>(p, valid[j], key[j], cur[j], ent[j]);
		pos[j] = 0;
		const unsigned long long any = __ballot(cls[j] >= 0);
		if (any) { // (wave-uniform)
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const unsigned long long nb = __ballot(cls[j] == c);
				if (nb) {
					uint32_t at = 0;
					if (lane == 0) at = atomicAdd(&sf->cnt[c], (uint32_t)__popcll(nb));
					at = (uint32_t)__shfl((int)at, 0, 64);
					if (cls[j] == c) pos[j] = at + (uint32_t)__popcll(nb & below);
				}
			}
			// statistics: re-clusterings queued and the values they carry
			uint32_t nv = cls[j] >= 0 ? ent[j].nbuf + ent[j].mrun : 0u;
#pragma unroll
			for (int d = 32; d >= 1; d >>= 1) nv += (uint32_t)__shfl_xor((int)nv, d, 64);
			if (lane == 0) {
				atomicAdd(&sf->nmerge, (uint32_t)__popcll(any));
				atomicAdd(&sf->nvals, (unsigned long long)nv);
			}
		}
	}
	__syncthreads();
	if (tid < 4u && sf->cnt[tid]) sf->base[tid] = atomicAdd(&p.counts[tid], sf->cnt[tid]);
	if (tid == 4u && sf->nmerge) {
		atomicAdd((unsigned long long *)&p.counters[CTR_TD_MERGES], (unsigned long long)sf->nmerge);
		atomicAdd((unsigned long long *)&p.counters[CTR_TD_MERGE_VALUES], sf->nvals);
	}
	__syncthreads();
#pragma unroll
	for (int j = 0; j < KMAX; ++j)
		if (cls[j] >= 0) p.list[cls[j]][sf->base[cls[j]] + pos[j]] = ent[j];
}

__global__ __launch_bounds__(256) void k_key_finalize(FinP p)
{
	__shared__ FinWg s_fin;
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	const bool valid[1] = {k < p.nsvc};
	const uint32_t key[1] = {k}, cur[1] = {valid[0] ? p.td_cur[k] : 0u};
	finalize_keys_wg<false, 1>(p, &s_fin, threadIdx.x, threadIdx.x & 63u, valid, key, cur);
}

// ---------------------------------------------------------------------------------------------------- predicted runs
// Before the event pass of a batch: a service of one of the batch's hosts whose LAST batch, repeated, would not fit its buffer gets a
// run in `staged` sized for that count plus a quarter (+ 64), and GYS_SPILL_BIT in td_cur -- the event pass then appends the key's
// pieces to the run instead of dropping them for a second walk.  What the prediction misses is caught after the pass (finalize_one):
// more values than the run holds -> exact run + second pass as before; fewer than the buffer has room for -> the run is copied into the
// buffer (k_run_append).  The digest state only depends on the key's value multiset per call, so the result is the same either way.
struct PreSpillP {
	uint32_t *td_cur;
	const uint32_t *td_prevm;
	uint32_t *td_run, *td_run0, *td_run1;
	uint32_t *counts;            // [FIN_RUN_ALLOC]: bump cursor into `staged`
	uint32_t *hot;               // [2]
	uint32_t hot_rd;             // hot[hot_rd]: did the previous batch see a key worth predicting?  (hot[hot_rd ^ 1] is cleared for this batch's end)
	const uint32_t *svc_host, *host_batch;
	uint32_t batch_stamp;        // host_batch[h] == batch_stamp: host h has a segment in this batch
	uint32_t nsvc, pcap, run_limit; // run_limit: predicted runs end below it (the exact runs of the fall-back need the rest of `staged`)
	uint32_t pend_cap;
	unsigned long long *resv;    // words requested so far by this batch's predictions (accepted or not), cleared with the cursor
};

__global__ __launch_bounds__(256) void k_prespill(PreSpillP p)
{
	__shared__ uint32_t s_w[4], s_base;
	if (blockIdx.x == 0 && threadIdx.x == 0) p.hot[p.hot_rd ^ 1u] = 0u;
	if (p.hot[p.hot_rd] == 0u) return; // (uniform: nothing to predict -- the usual case costs one 4-byte read per workgroup)
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t cap = 0, npend0 = 0;
	if (s < p.nsvc) {
		const uint32_t prev = p.td_prevm[s];
		if (prev > p.pcap - p.pend_cap && p.host_batch[p.svc_host[s]] == p.batch_stamp) {
			npend0 = p.td_cur[s];
			// (predicted with the run's own margin: a key whose batches end just below the buffer's end one time and just above it the next
			// would otherwise take the second pass every other batch -- r4c: 0.93 ms of second walks left on the Zipf shape; a run that
			// turns out to fit the buffer only costs its copy)
			const uint32_t want = prev + (prev >> 2) + 64u; // (a quarter + 64: a 500-value key's batches vary by +-22 (1 sigma): 8 sigma of room)
			if (!(npend0 & GYS_SPILL_BIT) && (uint64_t)npend0 + want > (uint64_t)p.pcap) cap = want;
		}
	}
	// one cursor atomic per workgroup
	const uint32_t inc = wave_incl_scan_u32(cap);
	if (lane == 63u) s_w[wave] = inc;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tot = 0;
		for (uint32_t k = 0; k < 4u; ++k) {
			const uint32_t c = s_w[k];
			s_w[k] = tot;
			tot += c;
		}
		// the cursor only ever counts ACCEPTED runs: the exact runs of the fall-back start at it and need the rest of `staged`.  The decision is
		// taken on a 64-bit RESERVATION counter that every asking workgroup adds to (it never goes back: a refused request only makes later
		// ones a little more likely to be refused too, and it cannot wrap), the places come from the 32-bit cursor, which accepted requests
		// alone advance -- accepted sums are bounded by the reservation sums, so the cursor ends at or below run_limit.  (Round 5's first form
		// was a compare-and-swap loop on the cursor: with thousands of workgroups asking at once -- 5 000 hosts x 1 000 services at 107
		// values per key and window, 128 values of room -- the retries took 24 ms per batch.)
		uint32_t base = 0xFFFFFFFFu;
		if (tot) {
			const unsigned long long r0 = atomicAdd(p.resv, (unsigned long long)tot);
			if (r0 + tot <= (unsigned long long)p.run_limit) base = atomicAdd(&p.counts[FIN_RUN_ALLOC], tot);
		}
		s_base = base;
	}
	__syncthreads();
	if (cap && s_base != 0xFFFFFFFFu) { // (else: no prediction for this workgroup's keys)
		const uint32_t start = s_base + s_w[wave] + inc - cap;
		p.td_run[s] = start;
		p.td_run0[s] = start;
		p.td_run1[s] = start + cap;
		p.td_cur[s] = npend0 | GYS_SPILL_BIT;
	}
}

// the hosts of a batch (the first-pass segments' host slots) get the batch's stamp
__global__ __launch_bounds__(256) void k_mark_hosts(const gys_resp_seg *segs, uint32_t nsegs, uint32_t *host_batch, uint32_t stamp)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nsegs) host_batch[segs[i].host_slot] = stamp;
}

// keys whose predicted run has to go into the buffer after all: staged[off_end - mrun .. off_end) -> td_pend[slot * pcap + nbuf ..]; one wave per entry
__global__ __launch_bounds__(256) void k_run_append(const MergeEnt *list, const uint32_t *count, const uint32_t *staged, uint32_t *td_pend, uint32_t pcap)
{
	const uint32_t n = *count, lane = threadIdx.x & 63u;
	for (uint32_t e = blockIdx.x * 4u + (threadIdx.x >> 6); e < n; e += gridDim.x * 4u) {
		const MergeEnt ent = list[e];
		const uint32_t *src = staged + (ent.off_end - ent.mrun);
		uint32_t *dst = td_pend + (size_t)ent.slot * pcap + ent.nbuf;
		// (eight loads in flight before the first store: the word-by-word loop waited for each load -- src and dst may alias as far as the compiler knows)
		for (uint32_t i0 = lane; i0 < ent.mrun; i0 += 64u * 8u) {
			uint32_t v[8];
#pragma unroll
			for (uint32_t u = 0; u < 8u; ++u) v[u] = i0 + 64u * u < ent.mrun ? src[i0 + 64u * u] : 0u;
#pragma unroll
			for (uint32_t u = 0; u < 8u; ++u)
				if (i0 + 64u * u < ent.mrun) dst[i0 + 64u * u] = v[u];
		}
	}
}

// ---------------------------------------------------------------------------------------------------- Count-Min rows of the window
// Count-Min is linear in the per-service event counts, so the response path only keeps one counter per service and window (resp_win)
// and the rows are built ONCE per window: 4 x 10^7 device-scope atomics per batch (3.5 ms at 10^7 services: they execute memory-side
// on a multi-XCD part) become LDS atomics.  Workgroup (chunk, row r, half h) walks its chunk of the services and adds the counts whose
// column hash_r(glob_id) falls into half h of the row into a 128-KiB LDS image, then stores the image as a partial row; k_cms_reduce
// sums the partials into the arena.  Row hash = jhash2(key, seed + r) as in the per-event form (DESIGN.md "Count-Min").
#define GYS_CMSF_CELLS (GYS_CMS_W / 2u)
__global__ __launch_bounds__(1024) void k_cms_partial(const uint32_t *resp_win, const uint64_t *svc_gid, uint32_t nsvc, uint32_t nch, uint32_t *partial)
{
	GYS_DYN_LDS(uint32_t, s_cells); // [GYS_CMSF_CELLS]
	const uint32_t r = blockIdx.y >> 1, half = blockIdx.y & 1u;
	for (uint32_t i = threadIdx.x; i < GYS_CMSF_CELLS; i += 1024u) s_cells[i] = 0;
	__syncthreads();
	const uint32_t per = (nsvc + nch - 1u) / nch;
	const uint32_t first = blockIdx.x * per, last = min(nsvc, first + per);
	// eight services per thread and round, all sixteen loads in flight before the first is used (round 3: the one-service loop exposed
	// one load latency per service -- 305 of them in a row per thread at 10^7 services, 0.31 ms for a pass that moves 120 MB)
	constexpr uint32_t U = 8;
	for (uint32_t s0 = first + threadIdx.x; s0 < last; s0 += 1024u * U) {
		uint32_t m[U];
		uint64_t g[U];
#pragma unroll
		for (uint32_t u = 0; u < U; ++u) {
			const uint32_t s = s0 + u * 1024u;
			const uint32_t sc = s < last ? s : first; // (past the end: the chunk's first service, ignored below)
			m[u] = resp_win[sc];
			g[u] = svc_gid[sc];
		}
#pragma unroll
		for (uint32_t u = 0; u < U; ++u) {
			if (s0 + u * 1024u >= last || !m[u]) continue;
			const uint32_t col = jhash2_u64(g[u], GYS_SEED + r) & (GYS_CMS_W - 1);
			if ((col / GYS_CMSF_CELLS) == half) atomicAdd(&s_cells[col % GYS_CMSF_CELLS], m[u]);
		}
	}
	__syncthreads();
	uint32_t *out = partial + ((size_t)blockIdx.x * GYS_CMS_D + r) * GYS_CMS_W + half * GYS_CMSF_CELLS;
	for (uint32_t i = threadIdx.x; i < GYS_CMSF_CELLS; i += 1024u) out[i] = s_cells[i];
}

__global__ __launch_bounds__(256) void k_cms_reduce(const uint32_t *partial, uint32_t nch, uint32_t *cms32)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= GYS_CMS_D * GYS_CMS_W) return;
	uint32_t sum = 0;
	for (uint32_t c = 0; c < nch; ++c) sum += partial[(size_t)c * GYS_CMS_D * GYS_CMS_W + i];
	if (sum) cms32[i] += sum;
}

// ---------------------------------------------------------------------------------------------------- host-local resp pass
// One workgroup per host segment of the batch (or per PART of a segment, SHARED).  The reference resolves a response event's listener
// inside the HOST's own listener table (TCP_SOCK_HANDLER is per host: common/gy_socket_stat.cc:1554-1677), so everything an event
// touches is host-local: the workgroup stages the host's (netns, port) -> local index sub-table in LDS and walks its events ONCE, tile
// by tile: per tile the events are resolved and filtered (registers), ranked inside their key with one LDS atomic each, the tile's
// per-key counts are scanned, the staged words are grouped by key in an LDS image, and every key's piece of the tile is appended to
// the key's value buffer with consecutive image entries going to consecutive addresses.  No per-event device-scope atomic except the
// global HLL register max; no intermediate per-event array in HBM.
//   SHARED: several workgroups may hold events of the same host (long segments cut into parts, a host named by two segments): buffer
//           space is reserved with one device atomic per (tile, key) instead of an LDS cursor.
//   SPILL:  second pass over the hosts that have spilled keys: only those keys' events, into their runs in `staged`.
// RESP_TIME_HASH bucket of a response time through a 1 KiB LDS table (GYS_BUCKET_LUT; r3c: -0.09 ms per window): values below 1024 -- eleven of the
// thirteen thresholds -- take one LDS read of four packed bucket numbers, the rest two compares; resp_bucket() costs 13 compare + add pairs
#ifndef GYS_BUCKET_LUT
#define GYS_BUCKET_LUT 1
#endif
// round 6: the instruction diet of the event phase, each step behind a switch of its own for A/B libraries (tools/ab_libs.sh)
#ifndef GYS_SHIFT_SWITCH
#define GYS_SHIFT_SWITCH 0 // (r6b: 5.40 ms with it, 5.32 - 5.33 without -- the scalar branches cost more than the 16 moves they save; left off) a group's results are stored at wd / lr[g .. g + 3] through a scalar branch on the (uniform) group number: 8 moves per group instead of the 24 of the shift by four
#endif
#ifndef GYS_PROBE_XOR
#define GYS_PROBE_XOR 1 // listener probe: entry ^ (port << 16) < 0xFFFF and high word == netns; no tests for empty entries in the two-entry fast path (the table is insert-only: a key behind an empty entry cannot exist)
#endif
#ifndef GYS_BK_BYTES
#define GYS_BK_BYTES 1 // RESP_TIME_HASH bucket table as 1024 bytes (one ds_read_u8 at min(t, 1023)) instead of 256 packed words + shift / mask
#endif
#ifndef GYS_PARK_INDEX
#define GYS_PARK_INDEX 1 // the rank atomic of a place that kept nothing goes to s_ts[Lc_park + lane]: one select for the index, no select between two addresses
#endif
#ifndef GYS_EV_DMA
#define GYS_EV_DMA 0 // (r6h / r6i: bit-exact and SLOWER -- 5.85 ms without, 6.22 ms with the requests one group ahead, against 5.24 - 5.39 ms; left off) (16 384-event tiles) a wave's events come in through LDS: six global_load_lds_dwordx4 per group of four event slots -- every 128-byte line of the wave's span requested ONCE, no destination registers -- into the wave's 6 KB of the (idle) tile-image area, then three 8-byte LDS reads per event.  r6g's counters: the vector-memory path stalls on pending lines half the time (TCP_PENDING_STALL_CYCLES 51 %, TD_TC_STALL 52 % of the kernel's cycles) with the strided 16 + 8-byte loads, which ask for every line twice
#endif
#ifndef GYS_EV_DMA_AHEAD
#define GYS_EV_DMA_AHEAD 1 // the requests of group g + 1 are issued as soon as group g's words have been read out of the wave's area (they run under group g's work)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef GYS_EV_DMA_CPOL
#define GYS_EV_DMA_CPOL 0 // cache policy bits of the requests (2 = nt: the event lines pass through the L2 without displacing the partly written buffer lines of the flush)
#endif
#define GYS_DMA16(gptr, ldsptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), (__attribute__((address_space(3))) void *)(ldsptr), 16, 0, GYS_EV_DMA_CPOL) // lane l's 16 bytes land at ldsptr + 16 l (ldsptr: wave-uniform)
#define GYS_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define GYS_DMA16(gptr, ldsptr) memcpy((char *)(ldsptr) + 16u * (threadIdx.x & 63u), (const void *)(gptr), 16)
#define GYS_DMA_WAIT() ((void)0)
#endif
#ifndef GYS_EV_PREFETCH
#define GYS_EV_PREFETCH 0 // one load instruction per wave and group touches the 48 lines of the wave's NEXT group of events (a dword each, result unused): the demand loads of the next group then come from the L2 instead of HBM
#endif
#ifndef GYS_FLOOR_QUARTER
#define GYS_FLOOR_QUARTER 1 // HLL floor: each tile re-reads a QUARTER of the register file (one 16-byte load per thread, issued before the flush's stores) and the floor is the minimum of the last four partial minima -- registers only grow, so an older minimum is still a lower bound -- instead of four loads per thread behind the stores
#endif
#ifndef GYS_HASH_FLAT
#define GYS_HASH_FLAT 1 // the first hash half of the four events in straight-line code (no branch per event around it); an event with a 0.0.0.0 end or with all 18 rank bits zero goes through the rolled general path
#endif
#ifndef GYS_PROBE_JOINT
#define GYS_PROBE_JOINT 0 // (r6b: 5.40 ms with it, 5.32 without; left off) third and later probes of the group's four events in ONE loop (one LDS round trip per step for all four, not one per event and step)
#endif
__device__ __forceinline__ void resp_bucket_lut_init(uint32_t *s_bk, uint32_t tid, uint32_t nthreads)
{
	// (both table forms hold the same bytes: word i = buckets of 4 i .. 4 i + 3, little endian)
	for (uint32_t i = tid; i < 256u; i += nthreads)
		s_bk[i] = resp_bucket((int64_t)(4u * i)) | (resp_bucket((int64_t)(4u * i + 1u)) << 8) | (resp_bucket((int64_t)(4u * i + 2u)) << 16) |
			  (resp_bucket((int64_t)(4u * i + 3u)) << 24);
}
__device__ __forceinline__ uint32_t resp_bucket_lut(const uint32_t *s_bk, uint32_t tresp)
{
	if (!GYS_BUCKET_LUT) return resp_bucket((int64_t)tresp);
#if GYS_BK_BYTES
	// (one ds_read_u8; the opaque statement keeps the compiler from turning it back into a word read + shift + mask)
	uint32_t bix = tresp & 1023u;
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("" : "+v"(bix));
#endif
	const uint32_t lo = ((const volatile uint8_t *)s_bk)[bix];
#else
	const uint32_t lo = (s_bk[(tresp >> 2) & 255u] >> ((tresp & 3u) * 8u)) & 0xFFu;
#endif
	const uint32_t hi = 12u + (tresp > 3000u) + (tresp > 15000u);
	return tresp < 1024u ? lo : hi;
}

struct HostDesc {
	uint32_t tbl_off;  // first entry of the host's sub-table in the table pool
	uint32_t mask;     // sub-table capacity - 1 (power of two)
	uint32_t nlst;     // local listener indices in use
	uint32_t lst_off;  // first entry of the host's local index -> service slot list in the list pool
	uint32_t part;     // a host with more listeners than one sub-table takes is cut into parts: this descriptor's part ...
	uint32_t pmask;    // ... of pmask + 1 (a power of two); a listener key belongs to part (host_tbl_hash(key) >> 21) & pmask
	uint32_t cand_off; // first record of the host's region in the candidate pool (sub-table entries with GYS_LOCAL_GROUP index it)
	uint32_t pad;
};

#define GYS_HOST_TBL_EMPTY 0xFFFFFFFFFFFFFFFFull
#define GYS_EV_DROPPED 0xFFFFFFFFu
#define GYS_HOST_THREADS 1024
// workgroup size of k_resp_host by tile form: 16 / 8 events per thread with 1024 threads (one workgroup per CU), or 12 events per thread
// with 512 threads (6144-event tiles, 76 KB of LDS at 1000 listeners: TWO workgroups per CU, so that one's prologue / scan / flush phases
// run under the other's event phase)
#define GYS_RESP_THREADS(TPT) (((TPT) == 12 || (TPT) == 32) ? 512 : 1024)
// TPT = 32 (round 4, experiment behind GYS_TPT=32): 512 threads x 32 events = the same 16 384-event tile as 1024 x 16, ONE workgroup per CU
// at two waves per SIMD, i.e. a budget of 256 VGPRs -- room to hold the NEXT group's twelve event words in registers while the current
// group is processed (at 128 VGPRs that prefetch spilled and lost, r3j / r3l / r3n), across the tile boundary too
#define GYS_RESP_WAVES_PER_SIMD(TPT) ((TPT) == 32 ? 2 : 4)
#define GYS_SPLIT_PART 65536u // events per part when long segments are cut (SHARED)

// EXPERIMENT builds only (-DGYS_RESP_TIMING): shader-clock ticks per phase of k_resp_host, summed over the waves of a launch (every wave adds
// its own sums at its end; s_memtime waits for the wave's outstanding LDS / scalar-memory operations, so a phase also pays for the returns
// of what it issued).  Read by gys_get_counters.
#ifdef GYS_RESP_TIMING
__device__ unsigned long long g_resp_timing[16];
#define GYS_TICK(i) do { const unsigned long long t_now__ = (unsigned long long)clock64(); tacc[i] += (uint32_t)(t_now__ - t_prev__); t_prev__ = t_now__; } while (0)
#else
#define GYS_TICK(i) do { } while (0)
#endif
struct RespHostP {
	const uint64_t *ev;
	uint64_t n;
	const gys_resp_seg *segs;
	uint32_t nsegs;
	const HostDesc *hdesc;
	const uint64_t *htbl;   // entries: (netns:32 | port:16) << 16 | local index:16 (GYS_LOCAL_GROUP set: index into the host's candidate region)
	const uint32_t *hlst;
	const ListenerCand *cand; // candidate pool (MODE != 0)
	uint32_t *hll32;
	uint32_t *td_cur;       // per service: words in its buffer including this batch's (SHARED: reserved with device atomics)
	uint32_t *td_pend;
	uint32_t pcap;
	uint32_t *td_run;       // per service with a run in `staged`: the run's fill cursor (SPILL pass: the exact run; first pass: the predicted run)
	const uint32_t *td_run1; // first pass: end of a key's predicted run (keys whose td_cur carries GYS_SPILL_BIT at the start of the pass; nullptr: none)
	long long run_delta;    // staged - td_pend in 4-byte words: a predicted run's index as an index into td_pend (the flush has ONE base pointer)
	uint32_t *staged;
	const uint32_t *host_spill; // SPILL: hosts with spilled keys carry spill_stamp
	uint32_t spill_stamp;
	uint64_t *counters;
	uint8_t *svc_hll;
	uint32_t svc_hll_p;
	unsigned long long *ghist; // arena: all-service histogram of the window, 15 x {count,sum} + {total}
	long long *gmax;           // arena: largest value of the window
	uint32_t lds_tbl_entries;  // LDS table area of the launch (largest sub-table among the batch's hosts)
	uint32_t lds_key_entries;  // LDS per-key areas (largest listener count, even)
	FinP fin;                  // !SHARED && !SPILL: the workgroup finalizes its host's keys itself (finalize_key)
	uint32_t dbg;              // timing experiments only (GYS_DBG): 1 no flush, 2 no image, 4 no HLL, 8 no all-service histogram, 16 no key counts, 32 hashes without register traffic
};

// LDS layout of one k_resp_host launch (bytes; shared by the engine and the tests): listener sub-table (+ 2 entries: the copy of entry 0
// behind the last one lets a probe read two consecutive entries with one instruction), per-key areas (24 B per local index), tile image
// (+ 2 entries: a parking place for the lanes of a short last round)
__host__ __device__ __forceinline__ size_t resp_host_lds_bytes(uint32_t tbl_entries, uint32_t key_entries, uint32_t tile)
{
	return ((size_t)tbl_entries + 2u) * 8u + (size_t)key_entries * 24u + ((size_t)tile + 2u) * 6u;
}

#ifndef GYS_RESP_DBG
#define GYS_RESP_DBG 0 // 1: RespHostP.dbg switches parts of the kernel off (timing experiments only, tools/r3h_phases.sh)
#endif
#ifndef GYS_GH_PER_WAVE
#define GYS_GH_PER_WAVE 0 // A/B: 1 = the cells per (wave, bucket) of rounds 1 - 3
#endif
#define GYS_GH_STRIDE 17u // cells per lane slot of the all-service histogram (15 buckets + the spare cell + 1: an odd stride)
#define GYS_HQ_CAP 512u // HLL candidates queued per tile (late in a window ~0.1 % of a tile's events qualify; the queue is drained by the first GYS_HQ_CAP threads)
#define GYS_DST_BIAS 65536ull // > the largest tile: (first word of a key's piece) - (start of its run in the image) + bias is positive
#ifndef GYS_OPAQUE_LOADED4
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
#endif
// the event words are read once: a non-temporal load leaves the L2 to the lines the flush is still filling (a key's piece of ~16 words ends inside
// a 128-byte line that the key's next piece, one tile later, completes -- evicted in between, the line is written twice)
#ifndef GYS_EV_NT
#define GYS_EV_NT 0 // (r5f: 5.30 -> 5.80 ms with non-temporal event loads: the three 8-byte words of an event come from one line, the second and third read want it cached)
#endif
#ifndef GYS_EV_SADDR
#define GYS_EV_SADDR 1 // (round 6: on, with the other steps of the diet; r5m measured it alone: (r5m: scalar tile base + one 32-bit offset per event -- the compiler then loads 16 + 8 bytes per event with one address register instead of three 64-bit addresses, 144 fewer static VALU instructions -- 5.28 / 5.29 against 5.33 / 5.30 ms: inside the noise; left off)
#endif
#ifndef GYS_EV_X3
#define GYS_EV_X3 0 // (r5g: 5.27 -> 5.60 ms with two fully coalesced 12-byte loads per event + a DPP swap of halves between neighbouring lanes instead of the three strided 8-byte loads: the loads are not what the event phase waits for, the extra moves and registers cost more than the request efficiency gains)
#endif
struct __attribute__((packed, aligned(4))) Ev3 {
	uint32_t x, y, z;
};
__device__ __forceinline__ uint32_t gys_swap_pair(uint32_t v) // the neighbouring lane's value (lanes 2j <-> 2j + 1): DPP quad_perm [1, 0, 3, 2]
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
#else
	return (uint32_t)__shfl_xor((int)v, 1, 64);
#endif
}
#if GYS_EV_NT
#define GYS_EV_LOAD(p) __builtin_nontemporal_load(p)
#else
#define GYS_EV_LOAD(p) (*(p))
#endif
#define GYS_MEM_FENCE() asm volatile("" ::: "memory") // compiler-only: memory operations are not moved across it (keeps a batch of LDS reads in front of the stores / the next batch)

// MODE 0: IPv4 events, every listener of the batch's hosts alone on its (netns, port) key and bound to the any-address (the instance of the
// measured configurations: nothing below costs it an instruction); 1: IPv4 events, keys with candidates (bound-address listeners) are
// resolved by the event's server address; 2: IPv6 events (48 bytes; flow hash through the general word packing, candidates as in 1)
template <int TPT, bool SHARED, bool SPILL, bool SVCHLL, int MODE = 0>
__global__ __launch_bounds__(GYS_RESP_THREADS(TPT), GYS_RESP_WAVES_PER_SIMD(TPT)) void k_resp_host(RespHostP p_arg) // (4 waves per SIMD: one 1024-thread workgroup or two 512-thread ones per CU; TPT = 32: 2)
{
#ifndef GYS_RESP_KERNARG
#define GYS_RESP_KERNARG 1 // the parameters are read from the kernel-argument segment where they are used (scalar loads), re-read per tile, instead of being held in SGPRs -- and spilled into VGPR lanes -- across the whole kernel
#endif
#if GYS_RESP_KERNARG && defined(__HIP_DEVICE_COMPILE__)
	typedef const RespHostP __attribute__((address_space(4))) *KernargP;
	KernargP p_k = (KernargP)__builtin_amdgcn_kernarg_segment_ptr(); // (the only parameter: offset 0)
#define GYS_RESP_P_RELOAD() asm volatile("" : "+s"(p_k))
	const RespHostP &p = *(const RespHostP *)p_k;
#else
#define GYS_RESP_P_RELOAD() (void)0
	const RespHostP &p = p_arg;
#endif
	constexpr uint32_t T = GYS_RESP_THREADS(TPT);
	constexpr uint32_t TILE = (uint32_t)TPT * T;
	constexpr bool DBG = GYS_RESP_DBG != 0;
	constexpr bool V6 = MODE == 2;
	constexpr uint32_t SW = V6 ? GYS_EV6_WORDS : 3u; // 8-byte words per event
	static_assert(MODE == 0 || TPT != 32, "the register-prefetch form exists for the plain IPv4 instance only");
	GYS_DYN_LDS(uint64_t, s_dyn);
	__shared__ uint32_t s_wsum[T / 64];
	__shared__ uint32_t s_drop[2];
	__shared__ uint32_t s_floor[2]; // HLL floor of the even / odd tiles (refreshed per tile)
	__shared__ uint32_t s_fqm[4];   // GYS_FLOOR_QUARTER: minimum of each quarter of the register file as last read (quarter q by the tiles with tile_no & 3 == q)
	// all-service histogram of the window, packed count << 40 | sum per cell.  Round 4: the cells are per (LANE SLOT, bucket), not per (wave,
	// bucket): the 64 lanes of one add hit at most 15 buckets, i.e. a handful of addresses each taken by many lanes, and LDS atomics on one
	// address execute one after the other -- measured (profiles/r4b_lds_conflicts_by_access_and_stagger.txt) 36 % of the kernel's bank-conflict
	// cycles and a quarter of its LDS-active cycles.  Lane l adds into slot l & 15: lanes of one bucket spread over 16 cells; the slot stride
	// of 17 cells puts the 16 slots of a bucket on 16 different bank pairs.  (Same 2 KB of LDS as the per-wave form: the waves share the cells.)
	__shared__ unsigned long long s_gh[16 * GYS_GH_STRIDE];
	__shared__ int32_t s_gmax;
	__shared__ uint32_t s_bk[GYS_BUCKET_LUT ? 256 : 1]; // (read as 1024 bytes with GYS_BK_BYTES)
	__shared__ uint32_t s_hq[SPILL ? 1 : GYS_HQ_CAP]; // the tile's HLL candidates: register index | rank << 16
	__shared__ uint32_t s_hqn;
	__shared__ FinWg s_fin;
	__shared__ uint32_t s_park[64]; // the rank atomic of an event that is not kept lands here (one word per lane: no same-address serialisation)
	if (GYS_BUCKET_LUT && !SPILL) resp_bucket_lut_init(s_bk, threadIdx.x, T);
	const uint32_t Lc = p.lds_key_entries;
	uint64_t *s_tbl = s_dyn;                            // [lds_tbl_entries + 2]
	uint64_t *s_dst = s_dyn + p.lds_tbl_entries + 2u;   // [Lc] index into dst (+ GYS_DST_BIAS) the key's piece of the tile would have if it started at image entry 0 (0: piece dropped)
	uint32_t *s_cur = (uint32_t *)(s_dst + Lc);         // [Lc] words in the key's buffer (SPILL: non-zero = spilled key)
	uint32_t *s_slot = s_cur + Lc;                      // [Lc] service slot of the local index
	uint32_t *s_ts2 = s_slot + Lc;                      // [2][Lc] (even / odd tiles) the tile's values of the key (low half; the rank counter of the event phase) | start of its run in the image << 16
	uint32_t *s_val = s_ts2 + 2u * Lc;                  // [TILE + 2] staged words grouped by key
	uint16_t *s_key = (uint16_t *)(s_val + TILE + 2u);  // [TILE + 2] local index of each image entry
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const gys_resp_seg seg = p.segs[blockIdx.x];
	const uint64_t e0 = seg.first_event;
	// (the workgroups of one segment of a many-listener host -- one per part of its listeners, seg.reserved = descriptor index + 1 --
	// sit next to each other with the same first event: the segment ends where the next entry with another start begins)
	uint32_t nx = blockIdx.x + 1;
	if (seg.reserved)
		while (nx < p.nsegs && p.segs[nx].reserved && p.segs[nx].host_slot == seg.host_slot && p.segs[nx].first_event == e0) ++nx;
	uint64_t e1 = nx < p.nsegs ? p.segs[nx].first_event : p.n;
	if (e1 > p.n) e1 = p.n;
	if (e1 <= e0) return;
	if (SPILL && p.host_spill[seg.host_slot] != p.spill_stamp) return;
	const HostDesc hd = p.hdesc[seg.reserved ? seg.reserved - 1u : seg.host_slot];
	const uint32_t mask = hd.mask, L = hd.nlst;
	for (uint32_t i = tid; i <= mask; i += T) s_tbl[i] = p.htbl[hd.tbl_off + i];
	for (uint32_t k = tid; k < L; k += T) {
		const uint32_t slot = p.hlst[hd.lst_off + k];
		const uint32_t c = p.td_cur[slot];
		s_slot[k] = slot;
		s_cur[k] = SPILL ? (c & GYS_RESPILL_BIT) : c; // (first pass: a key with a predicted run carries GYS_SPILL_BIT here)
		s_ts2[k] = 0;
		s_ts2[Lc + k] = 0;
	}
	if (tid < 2) s_drop[tid] = 0;
	if (tid < 16u * GYS_GH_STRIDE) s_gh[tid] = 0;
	if (tid == 0) {
		s_hqn = 0;
		s_floor[0] = SPILL ? 0u : 0xFFFFFFFFu;
		s_floor[1] = 0xFFFFFFFFu;
		s_gmax = INT32_MIN;
		s_tbl[mask + 1u] = p.htbl[hd.tbl_off]; // a probe reads entries h and h + 1 at once
		s_val[TILE] = 0;                       // the parking entry of the flush
		s_key[TILE] = 0;
	}
	__syncthreads();
	// HLL floor: a register can only grow, so min over the register file (stale L1 lines only lower it) is a lower bound for what follows
	// -- events whose rank does not exceed it skip the register access altogether.  Late in a window that is all but ~2^-floor of the
	// events; without it every event pays a random 4-byte read.  Taken per WORKGROUP at its start (a window's registers start at zero, so
	// a launch-wide floor taken before the first workgroup is 0 for the whole batch -- measured: 7.0 instead of 6.2 ms) and, round 3,
	// REFRESHED PER TILE (tile t + 2 runs on the minimum taken behind tile t -- two tiles on, so that no barrier of its own is needed): the first workgroups of a window start with floor 0 and used to keep it for their whole segment -- with few,
	// long segments (the per-rank load of an 8-GPU run: 1250 hosts x 429 000 events) that was a fifth of the batch reading a register per event.
	if (!SPILL) {
		if (e1 - e0 >= 4096u) {
			uint32_t mn = 0xFFFFFFFFu;
			const uint4 *h4 = (const uint4 *)p.hll32;
			for (uint32_t i = tid; i < (1u << GYS_HLL_P) / 4u; i += T) {
				const uint4 v = h4[i];
				mn = min(min(mn, min(v.x, v.y)), min(v.z, v.w));
			}
			mn = wave_min_u32(mn);
			if (lane == 0) atomicMin(&s_floor[0], mn);
		} else if (tid == 0) {
			s_floor[0] = 0;
		}
		__syncthreads();
		if (tid == 0) s_floor[1] = s_floor[0]; // (tiles 0 and 1 run on the floor taken here; tile t + 2 on the one refreshed behind tile t)
		if (tid < 4) s_fqm[tid] = s_floor[0];  // (a lower bound for every quarter)
	}
	uint32_t hll_floor = 0;
	const uint32_t K = (L + T - 1) / T;
	const uint32_t klo = min(L, tid * K), khi = min(L, klo + K);
	uint32_t *const dst = SPILL ? p.staged : p.td_pend;
	uint32_t *const dstb = dst - GYS_DST_BIAS; // (only ever indexed with biased indices: see s_dst)

	uint32_t ndrop_range = 0, ndrop_nol = 0, tile_no = 0, dbg_sink = 0;
#ifdef GYS_RESP_TIMING
	uint32_t tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long t_prev__ = (unsigned long long)clock64();
#endif
#if GYS_EV_PREFETCH
	uint32_t pf_sink = 0; // destination of the line-touching loads
#endif
	const uint32_t tid24 = 24u * tid;
	int32_t tmax = INT32_MIN, wmax = -1;
	// PF: the twelve words of the NEXT group of four events per thread are requested while the current group is processed and wait in
	// n0 / n1 / n2 (the group after a tile's last one is the next tile's first: its loads run under the scan / image / flush phases)
	constexpr bool PF = TPT == 32 && !SPILL;
	// events through LDS (GYS_EV_DMA): the tile image of 6 bytes per event is 6 KB per wave with 16 events per thread -- room for the 4 x 1536 bytes of a
	// group's four event slots.  Layout of a wave's area: the first KB of slot u at 1024 u, its last 512 bytes at 4096 + 512 u (the tails of two slots
	// are one full-width request).  The 8-byte piece k of lane l's event (byte 24 l + 8 k of the slot) never straddles the KB boundary.
	constexpr bool DMA = GYS_EV_DMA && TPT == 16 && !SPILL && MODE != 2;
	char *const dma_ws = (char *)s_val + wave * 6144u;
	uint32_t dma_a[3], dma_s[3]; // address of piece k in slot 0 and its stride from slot to slot
#pragma unroll
	for (uint32_t k = 0; k < 3u; ++k) {
		const uint32_t byte = 24u * lane + 8u * k;
		dma_a[k] = byte < 1024u ? byte : 4096u + (byte - 1024u);
		dma_s[k] = byte < 1024u ? 1024u : 512u;
	}
	uint64_t n0[4], n1[4], n2[4];
	auto pf_issue = [&](const uint64_t *base, uint32_t first_o, uint32_t lim) {
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint32_t o = first_o + (uint32_t)u * T + tid;
			const uint32_t oo = o < lim ? o : 0u; // (lanes past the end read the tile's first event and ignore it)
			n0[u] = base[3u * oo];
			n1[u] = base[3u * oo + 1u];
			n2[u] = base[3u * oo + 2u];
		}
	};
	if (PF) {
		const uint64_t left0 = e1 - e0;
		pf_issue(p.ev + 3u * e0, 0u, left0 < (uint64_t)TILE ? (uint32_t)left0 : TILE);
	}
	GYS_TICK(10); // prologue: tables, floor
	for (uint64_t t0 = e0; t0 < e1; t0 += TILE, ++tile_no) {
		GYS_RESP_P_RELOAD();
		// (no barrier here: the event phase of this tile touches nothing the flush of the previous one reads -- the per-key counters are
		// double-buffered and were cleared two phases ago, the floor and the candidate queue were settled behind barriers of the
		// previous tile; a wave that is done flushing starts on its next events while the others still flush)
		uint32_t *const s_ts = s_ts2 + (tile_no & 1u) * Lc;
		if (DMA) __syncthreads(); // (the waves stage their events in the tile-image area: the previous tile's flush must be over)
#if GYS_PARK_INDEX
#if defined(__HIP_DEVICE_COMPILE__)
		typedef uint32_t park_ix_t; // (LDS addresses are 32 bits: wrap-around arithmetic on word indices, s_ts[park_ix] is s_park[lane])
#else
		typedef ptrdiff_t park_ix_t; // (the CPU emulation of tests/cpp/kemu: two host allocations)
#endif
		const park_ix_t park_ix = (park_ix_t)(&s_park[lane] - s_ts);
#endif
		if (!SPILL) hll_floor = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_floor[tile_no & 1u]);
		// ---- resolve, filter, rank: 4 events per thread at a time, each step for all four before the next one (event loads, listener
		// probes, hashes, HLL register reads, rank atomics), in straight-line predicated code: the LDS / HBM round trips of the four
		// events overlap instead of following each other (the branchy per-event form of round 2 compiled to one exposed LDS latency per
		// probe and event).  Events are addressed by a 32-bit offset from the tile's (uniform) base.  The loop over the groups of 4 is
		// NOT unrolled (one copy of the event code instead of TPT / 4: the unrolled form was > 100 KB of instructions, twice the
		// instruction cache two CUs share); the per-event results live in registers all the same: wd / lr are shifted down by 4 per
		// group, so that every index stays a compile-time constant and after TPT / 4 groups event g sits at wd[g].
		const uint64_t left = e1 - t0;
		const uint32_t rem = left < (uint64_t)TILE ? (uint32_t)left : TILE; // events of this tile (> 0)
		const uint64_t *const tb = p.ev + (uint64_t)SW * t0;
		uint32_t wd[TPT], lr[TPT]; // staged word (GYS_EV_DROPPED: not kept) / local index | rank inside the key's tile run << 12
#pragma unroll
		for (int u = 0; u < TPT; ++u) {
			wd[u] = GYS_EV_DROPPED; // (a short last tile leaves the group loop early: unused places must read as dropped)
			lr[u] = 0;
		}
#pragma unroll 1
		for (int g = 0; g < TPT; g += 4) {
			if ((uint32_t)g * T >= rem) break; // the segment's last tile is usually short (C3: 53 687 events = 3.28 tiles): no empty groups
			uint64_t w0[4], w1[4], w2[4];
			bool in[4];
			if (PF) {
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					in[u] = (uint32_t)(g + u) * T + tid < rem;
					w0[u] = n0[u];
					w1[u] = n1[u];
					w2[u] = n2[u];
				}
				// the group that runs next: this tile's, or the first one of the next tile
				const uint32_t g2 = (uint32_t)g + 4u;
				if (g2 < (uint32_t)TPT && g2 * T < rem) {
					pf_issue(tb, g2 * T, rem);
				} else if (t0 + TILE < e1) {
					const uint64_t left2 = e1 - t0 - TILE;
					pf_issue(tb + 3u * TILE, 0u, left2 < (uint64_t)TILE ? (uint32_t)left2 : TILE);
				}
			} else {
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const uint32_t o = (uint32_t)(g + u) * T + tid;
					in[u] = o < rem;
					const uint32_t oo = in[u] ? o : 0u; // (lanes past the end read the tile's first event and ignore it: no branch around the loads)
					if (V6) { // the name space / ports word and the times word; the addresses are read where they are needed (candidates, flow hash)
						w0[u] = 0;
						w1[u] = tb[GYS_EV6_WORDS * oo + 4u];
						w2[u] = tb[GYS_EV6_WORDS * oo + 5u];
					} else if (GYS_EV_X3) {
						// Two 12-byte loads per lane instead of three 8-byte ones with a 24-byte stride: lane l of a wave reads HALF-events --
						// load A the halves of the block's events 0..31 (lane 2j: first half of event j, lane 2j + 1: its second half), load B
						// those of events 32..63 -- so every load instruction covers 768 contiguous bytes (six full lines; the strided form touches
						// twelve lines per instruction and each line three times).  Neighbouring lanes then swap halves (DPP quad_perm): the even
						// lane ends with event j, the odd lane with event 32 + j.
						const uint32_t e_blk = (uint32_t)(g + u) * T + (tid & ~63u), par = lane & 1u;
						const uint32_t ev_a = e_blk + (lane >> 1), ev_b = ev_a + 32u;
						const Ev3 *hp = (const Ev3 *)tb;
						const Ev3 a = hp[ev_a < rem ? 2u * ev_a + par : par], b = hp[ev_b < rem ? 2u * ev_b + par : par];
						const uint32_t xa0 = gys_swap_pair(a.x), xa1 = gys_swap_pair(a.y), xa2 = gys_swap_pair(a.z);
						const uint32_t xb0 = gys_swap_pair(b.x), xb1 = gys_swap_pair(b.y), xb2 = gys_swap_pair(b.z);
						const uint32_t d0 = par ? xb0 : a.x, d1 = par ? xb1 : a.y, d2 = par ? xb2 : a.z; // first half: saddr, daddr, netns
						const uint32_t d3 = par ? b.x : xa0, d4 = par ? b.y : xa1, d5 = par ? b.z : xa2; // second half: ports, lsndtime, lrcvtime
						in[u] = (par ? ev_b : ev_a) < rem;
						w0[u] = (uint64_t)d0 | ((uint64_t)d1 << 32);
						w1[u] = (uint64_t)d2 | ((uint64_t)d3 << 32);
						w2[u] = (uint64_t)d4 | ((uint64_t)d5 << 32);
					} else if (DMA) {
						in[u] = (uint32_t)(g + u) * T + tid < rem; // (the words are read from the wave's LDS area below, behind the requests of all four slots)
						w0[u] = w1[u] = w2[u] = 0;
					} else if (GYS_EV_SADDR) {
						// the tile's base is uniform and an event's byte offset inside the tile fits 32 bits: written so, the three loads share ONE
						// 32-bit offset register (scalar base + offset + immediate) instead of a 64-bit address each.  Round 6: the offset is
						// 24 tid (kept) + 24 T (g + u) (scalar) -- no 32-bit multiply (a quarter-rate instruction) per event
						const uint32_t ob_raw = tid24 + (uint32_t)(g + u) * (24u * T);
						in[u] = ob_raw < 24u * rem; // (== o < rem)
						const uint32_t ob = in[u] ? ob_raw : 0u;
						const char *const tbb = (const char *)tb;
						w0[u] = *(const uint64_t *)(tbb + ob);
						w1[u] = *(const uint64_t *)(tbb + ob + 8u);
						w2[u] = *(const uint64_t *)(tbb + ob + 16u);
					} else {
						w0[u] = GYS_EV_LOAD(&tb[3u * oo]);
						w1[u] = GYS_EV_LOAD(&tb[3u * oo + 1u]);
						w2[u] = GYS_EV_LOAD(&tb[3u * oo + 2u]);
					}
				}
			}
			if (DMA) {
				// the requests of a group: six full-width 16-byte-per-lane transfers into the wave's area.  A 16-byte piece may reach 8 bytes past the
				// tile's last event (into the next tile's or segment's events): it is requested as long as it lies inside the batch; the batch's
				// very last event (its last piece would pass the buffer's end when n is odd) is re-read below
				auto dma_issue = [&](uint32_t gg) {
					const char *const tbb = (const char *)tb;
					const uint64_t left_b = (p.n - t0) * 24ull;
					const uint32_t lim = left_b > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)left_b;
#pragma unroll
					for (uint32_t u = 0; u < 4u; ++u) { // the first KB of each slot: lane l asks for bytes [16 l, 16 l + 16) of the wave's 1536
						const uint32_t bo = 24u * ((gg + u) * T + (tid & ~63u)) + 16u * lane;
						GYS_DMA16(tbb + (bo + 16u <= lim ? bo : 0u), dma_ws + 1024u * u);
					}
#pragma unroll
					for (uint32_t t2 = 0; t2 < 2u; ++t2) { // the last 512 bytes of two slots: lanes 0..31 slot 2 t2, lanes 32..63 slot 2 t2 + 1
						const uint32_t u = 2u * t2 + (lane >> 5);
						const uint32_t bo = 24u * ((gg + u) * T + (tid & ~63u)) + 1024u + 16u * (lane & 31u);
						GYS_DMA16(tbb + (bo + 16u <= lim ? bo : 0u), dma_ws + 4096u + 1024u * t2);
					}
				};
				if (!GYS_EV_DMA_AHEAD || g == 0) dma_issue((uint32_t)g); // (the tile's first group: behind the tile-top barrier)
				GYS_DMA_WAIT();
				__builtin_amdgcn_wave_barrier();
#pragma unroll
				for (uint32_t u = 0; u < 4u; ++u) {
					w0[u] = *(const uint64_t *)(dma_ws + dma_a[0] + u * dma_s[0]);
					w1[u] = *(const uint64_t *)(dma_ws + dma_a[1] + u * dma_s[1]);
					w2[u] = *(const uint64_t *)(dma_ws + dma_a[2] + u * dma_s[2]);
					const uint32_t o = (uint32_t)(g + u) * T + tid;
					if (t0 + o + 1u == p.n) { // (one lane of one workgroup per batch)
						w0[u] = tb[3u * o];
						w1[u] = tb[3u * o + 1u];
						w2[u] = tb[3u * o + 2u];
					}
				}
				GYS_OPAQUE_LOADED4(w0); // (the words are in registers: the area is free again)
				GYS_OPAQUE_LOADED4(w1);
				GYS_OPAQUE_LOADED4(w2);
				__builtin_amdgcn_wave_barrier();
				// the NEXT group's requests go out now and run under this group's work: software pipelining with one buffer and no registers
				if (GYS_EV_DMA_AHEAD && g + 4 < TPT && (uint32_t)(g + 4) * T < rem) dma_issue((uint32_t)g + 4u);
			}
			// (all twelve words pass through one opaque statement: the four events' loads are issued before the first word is used -- the
			// scheduler otherwise waits for event 0 and starts on its fields before the loads of events 1..3 are even issued.  Requesting
			// the NEXT group's words before this group is processed was measured twice (r3j, r3l / r3n: 1.95 against 1.80 ms at quarter
			// size): slower -- the extra live registers spill and every vmcnt wait of the group then also waits for the prefetch)
			GYS_OPAQUE_LOADED4(w0);
			GYS_OPAQUE_LOADED4(w1);
			GYS_OPAQUE_LOADED4(w2);
			GYS_TICK(0); // group top: addresses, the event loads issued and arrived
#if GYS_EV_PREFETCH && defined(__HIP_DEVICE_COMPILE__)
			if (!PF && !V6 && !SPILL) {
				// the wave's events of place u of a group are 64 consecutive 24-byte records = twelve 128-byte lines: lanes 0..47 touch one line
				// each of the NEXT group (this tile's, or the first of the next tile).  Issued once this group's own loads HAVE ARRIVED (r6d: issued
				// before that wait it was waited for as well -- the compiler's vmcnt(0) cannot tell it apart -- and cost 9 %): it runs under this group's
				// work and is older than every load that is waited for later; nothing reads pf_sink
				const uint32_t g2 = (uint32_t)g + 4u;
				const uint64_t *nb = nullptr;
				uint32_t nlim = 0, nfirst = 0;
				if (g2 < (uint32_t)TPT && g2 * T < rem) {
					nb = tb; nlim = rem; nfirst = g2 * T;
				} else if (t0 + TILE < e1) {
					const uint64_t left2 = e1 - t0 - TILE;
					nb = tb + 3u * TILE; nlim = left2 < (uint64_t)TILE ? (uint32_t)left2 : TILE; nfirst = 0;
				}
				if (nb != nullptr && lane < 48u) {
					const uint32_t pu = lane / 12u, pl = lane - 12u * pu;
					const uint32_t ev0 = nfirst + pu * T + (tid & ~63u);           // first event of the wave's 64 in place pu
					const uint32_t byte = 24u * ev0 + 128u * pl;                   // (the tile's base is 8-byte aligned only: "line" = 128-byte piece of the wave's span)
					if (byte + 4u <= 24u * nlim) {
						const char *pa = (const char *)nb + byte;
						asm volatile("global_load_dword %0, %1, off" : "=v"(pf_sink) : "v"(pa) : "memory");
					}
				}
			}
#endif
			// struct ipv4_tuple_t {u32 saddr, daddr, netns; u16 sport, dport;} + u32 lsndtime, lrcvtime  (24 bytes)
			uint32_t tresp[4], local[4];
			uint64_t ea[4], eb[4];
			bool ok[4], more[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const uint32_t netns = (uint32_t)w1[u];
				const uint32_t sport = (uint32_t)bswap16((uint16_t)(w1[u] >> 32)); // ntohs :1526-1527
				tresp[u] = (uint32_t)w2[u] - (uint32_t)(w2[u] >> 32);               // lsndtime - lrcvtime (:1519)
				const bool range_ok = tresp[u] <= 1000000u;                         // "Ignore responses > 1000 sec or negative" (:1521-1524)
				// (counted once per event: by the workgroup of part 0.  Round 6 tried the two drop counters per WAVE from the compare masks -- s_bcnt1 on
				// the scalar unit -- but hipcc turns every ballot, also of a single compare, into a 0 / 1 select + a second compare: 2 VALU against this add with carry)
				if (in[u] && !range_ok && hd.part == 0u) ndrop_range++;
				const uint32_t hk = host_tbl_hash(netns, sport);
				// (a listener of another part of this host: that part's workgroup has the event)
				const bool mine = (hk & (hd.pmask << 21)) == (hd.part << 21); // == (host_tbl_part(hk, hd.pmask) == hd.part), one shift less
				ok[u] = in[u] && range_ok && mine;
				const uint32_t h = host_tbl_slot(hk, mask);
				ea[u] = s_tbl[h];
				eb[u] = s_tbl[h + 1u];
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				// entry = netns : 32 | port : 16 | local index : 16 -- compared as two 32-bit words
				const uint32_t netns = (uint32_t)w1[u], sport = (uint32_t)bswap16((uint16_t)(w1[u] >> 32));
				const uint32_t alo = (uint32_t)ea[u], ahi = (uint32_t)(ea[u] >> 32), blo = (uint32_t)eb[u], bhi = (uint32_t)(eb[u] >> 32);
#if GYS_PROBE_XOR
				// low word of an entry = port << 16 | local: xor-ed with the event's port << 16 it IS the local index when the ports agree
				// (< 0xFFFF: an empty entry -- all ones -- never matches, whatever the event's key).  The sub-tables are insert-only linear
				// probing (host_tbl_put): a key found in the second entry implies a used first one, and a miss in both goes to the walk
				// below, which stops at the first empty entry -- no tests for empty entries here
				const uint32_t sp16 = sport << 16, xa = alo ^ sp16, xb = blo ^ sp16;
				const bool hit_a = ahi == netns && xa < 0xFFFFu, hit_b = bhi == netns && xb < 0xFFFFu;
				uint32_t l = hit_b ? xb : GYS_NOSLOT;
				l = hit_a ? xa : l;
				local[u] = l;
				more[u] = ok[u] && !hit_a && !hit_b; // (r6t measured what a one-probe table would save with `&& !(DBG && (p.dbg & 64u))` here: 5.32 -> 5.11 ms)
#else
				const bool hit_a = ahi == netns && (alo >> 16) == sport, hit_b = bhi == netns && (blo >> 16) == sport;
				const bool end_a = (alo & ahi) == 0xFFFFFFFFu, end_b = (blo & bhi) == 0xFFFFFFFFu;
				uint32_t l = (hit_b && !end_a) ? (blo & 0xFFFFu) : GYS_NOSLOT;
				l = hit_a ? (alo & 0xFFFFu) : l;
				local[u] = l;
				more[u] = ok[u] && !hit_a && !hit_b && !end_a && !end_b;
#endif
			}
#ifdef GYS_RESP_TIMING
			asm volatile("" : "+v"(local[0]), "+v"(local[1]), "+v"(local[2]), "+v"(local[3]));
#endif
			GYS_TICK(1); // hash, probe of two entries, compare
			// third and later probes: 3 % of the events at a quarter-full table (one in eight at a half-full one)
#if GYS_PROBE_JOINT
			if (more[0] || more[1] || more[2] || more[3]) {
				uint32_t hh[4];
#pragma unroll
				for (int u = 0; u < 4; ++u)
					hh[u] = (host_tbl_slot(host_tbl_hash((uint32_t)w1[u], (uint32_t)bswap16((uint16_t)(w1[u] >> 32))), mask) + 2u) & mask;
				for (uint32_t probes = 2; probes <= mask; ++probes) {
					uint64_t e[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) e[u] = s_tbl[hh[u]]; // (all four reads in flight together; a place that is done reads its last entry again)
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						const uint64_t key48 = ((uint64_t)(uint32_t)w1[u] << 16) | (uint64_t)bswap16((uint16_t)(w1[u] >> 32));
						const bool hit = (e[u] >> 16) == key48 && e[u] != GYS_HOST_TBL_EMPTY;
						if (more[u] && hit) local[u] = (uint32_t)(e[u] & 0xFFFFu);
						more[u] = more[u] && !hit && e[u] != GYS_HOST_TBL_EMPTY;
						hh[u] = more[u] ? ((hh[u] + 1u) & mask) : hh[u];
					}
					if (!(more[0] || more[1] || more[2] || more[3])) break;
				}
			}
#else
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				if (more[u]) {
					const uint64_t key48 = ((uint64_t)(uint32_t)w1[u] << 16) | (uint64_t)bswap16((uint16_t)(w1[u] >> 32));
					uint32_t h = (host_tbl_slot(host_tbl_hash((uint32_t)(key48 >> 16), (uint32_t)(key48 & 0xFFFFu)), mask) + 2u) & mask;
					for (uint32_t probes = 2; probes <= mask; ++probes) {
						const uint64_t e = s_tbl[h];
						if ((e >> 16) == key48) {
							local[u] = (uint32_t)(e & 0xFFFFu);
							break;
						}
						if (e == GYS_HOST_TBL_EMPTY) break;
						h = (h + 1) & mask;
					}
				}
			}
#endif
#ifdef GYS_RESP_TIMING
			asm volatile("" : "+v"(local[0]), "+v"(local[1]), "+v"(local[2]), "+v"(local[3]));
#endif
			GYS_TICK(2); // third and later probes
			if (MODE != 0) {
				// a key with candidates: the event's server address picks the listener (first match in registration order), or nobody
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					if (ok[u] && local[u] != GYS_NOSLOT && (local[u] & GYS_LOCAL_GROUP)) {
						uint32_t e32, e128[4] = {0u, 0u, 0u, 0u}, lc = 0, sl = 0;
						if (V6) {
							const uint32_t o = (uint32_t)(g + u) * T + tid;
							const uint64_t x0 = tb[GYS_EV6_WORDS * o], x1 = tb[GYS_EV6_WORDS * o + 1u];
							e128[0] = (uint32_t)x0; e128[1] = (uint32_t)(x0 >> 32); e128[2] = (uint32_t)x1; e128[3] = (uint32_t)(x1 >> 32);
							e32 = ip6_embedded_v4(e128);
						} else {
							e32 = (uint32_t)w0[u]; // saddr: GY_IP_ADDR(uint32_t) (:1529)
						}
						local[u] = cand_resolve(p.cand + hd.cand_off, local[u] & (GYS_LOCAL_GROUP - 1u), e32, e128, &lc, &sl) ? lc : GYS_NOSLOT;
					}
				}
			}
			uint32_t nwd[4], nlr[4], hidx[4], hrank[4], hcur[4], rare = 0;
			bool kept[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				kept[u] = ok[u] && local[u] != GYS_NOSLOT;
				if (ok[u] && local[u] == GYS_NOSLOT) ndrop_nol++; // no such listener: the reference ignores the event too (:1671-1676 miss path)
				nlr[u] = kept[u] ? local[u] : 0u;
			}
			if (SPILL) {
				uint32_t sc[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) sc[u] = s_cur[nlr[u]];
#pragma unroll
				for (int u = 0; u < 4; ++u) kept[u] = kept[u] && sc[u] != 0u; // (else the key's values already sit in its buffer)
			}
			// the LDS operations of the four events first, all in flight at once: the RESP_TIME_HASH bucket of each response time (table read)
			// and the event's rank inside its key's run of the tile (one returning atomic each; a place that kept nothing parks on a spare
			// word) -- their latency runs under the flow hashes below, which are pinned behind them by the opaque statement
			uint32_t bk[4], rk12[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const uint32_t dport = (uint32_t)bswap16((uint16_t)(w1[u] >> 48));
				nwd[u] = kept[u] ? ((tresp[u] << GYS_ROW_BITS) | (V6 ? GYS_ROW_V6 : 0u) | (dport & 0x1Fu)) : GYS_EV_DROPPED;
				bk[u] = 15u; // (s_gh[.][15] is a spare cell)
				if (!SPILL) {
					const uint32_t b = resp_bucket_lut(s_bk, tresp[u]); // (read for every lane: no branch)
					bk[u] = kept[u] ? b : 15u;
				}
			}
			if (SPILL && !(kept[0] || kept[1] || kept[2] || kept[3])) continue; // (second pass: most groups hold no event of a spilled key; wd / lr already read as dropped)
#pragma unroll
			for (int u = 0; u < 4; ++u) {
#if GYS_PARK_INDEX
				// (s_park lies in the same LDS segment: its cell is addressed as an index from s_ts -- one select between two indices, and
				// the index of a kept place is the local index the staged entry carries anyway)
				const park_ix_t ci = (kept[u] && !(DBG && (p.dbg & 16u))) ? (park_ix_t)nlr[u] : park_ix;
				rk12[u] = atomicAdd(&s_ts[ci], 1u);
#else
				uint32_t *const cell = (kept[u] && !(DBG && (p.dbg & 16u))) ? &s_ts[nlr[u]] : &s_park[lane];
				rk12[u] = atomicAdd(cell, 1u);
#endif
			}
			GYS_TICK(3); // kept / staged word / bucket table read / rank atomics issued (s_memtime waits for their returns)
			if (!SPILL) {
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					// all-service histogram of the window (GY_HISTOGRAM::add_data on the aggregate): one packed LDS add per event
					// (a place that kept nothing adds into the spare cell, which nobody reads: the add itself stays unconditional)
					if (!(DBG && (p.dbg & 8u))) atomicAdd(&s_gh[(GYS_GH_PER_WAVE ? wave : (lane & 15u)) * GYS_GH_STRIDE + bk[u]], (1ull << 40) | (unsigned long long)tresp[u]);
					// (largest response time through the staged word: value << GYS_ROW_BITS | row is monotone in the value and a place that kept
					// nothing holds GYS_EV_DROPPED = -1 as a signed number -- one max per event, no select)
					wmax = max(wmax, (int32_t)nwd[u]);
				}
				GYS_OPAQUE_LOADED4(w0);
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const uint32_t saddr = (uint32_t)w0[u], daddr = (uint32_t)(w0[u] >> 32);
				const uint16_t sport = bswap16((uint16_t)(w1[u] >> 32)), dport = bswap16((uint16_t)(w1[u] >> 48));
				hrank[u] = 0;
				hidx[u] = 0;
				if (!SPILL) {
					// both ends IPv4 and no per-service registers: index and rank from the first hash half (flow_hll_idx_rank), computed for
					// every lane (no branch; the result of a lane that kept nothing is discarded).  Everything else (0.0.0.0 / IPv6-mapped
					// ends hash as 7 / 10 words; the per-service registers need all 64 bits) is rare or a non-default configuration and
					// goes through ONE rolled copy of the general code below
					if (DBG && (p.dbg & 4u)) {
#if GYS_HASH_FLAT
					} else if (!V6 && !SVCHLL) {
						// flow_hll_idx_rank's common case without its branches: register index and the first 18 rank bits from the HIGH hash half,
						// computed for every lane.  A 0.0.0.0 end (hashes as 7 / 10 words) or 18 zero rank bits (one event in 2^18: the low
						// half decides) send the event through the rolled general code below, which recomputes both halves
						const uint32_t hi = jhash2_4w(daddr, dport, saddr, sport, GYS_SEED);
						const uint32_t rest = hi << GYS_HLL_P;
#if defined(__HIP_DEVICE_COMPILE__)
						const uint32_t rk = (uint32_t)__builtin_clz(rest) + 1u; // (rest == 0: not used)
#else
						const uint32_t rk = rest ? (uint32_t)__clz((int)rest) + 1u : 0u;
#endif
						const bool fast = daddr != 0 && saddr != 0 && rest != 0;
						const bool up = kept[u] && fast && rk > hll_floor; // (a rank at or below the floor cannot raise any register)
						hidx[u] = up ? (hi >> (32 - GYS_HLL_P)) : 0u;
						hrank[u] = up ? rk : 0u;
						rare |= (kept[u] && !fast) ? (1u << u) : 0u;
#else
					} else if (!V6 && !SVCHLL && daddr != 0 && saddr != 0) {
						uint32_t ix, rk;
						flow_hll_idx_rank(daddr, dport, saddr, sport, &ix, &rk);
						const bool up = kept[u] && rk > hll_floor; // (a rank at or below the floor cannot raise any register)
						hidx[u] = up ? ix : 0u;
						hrank[u] = up ? rk : 0u;
#endif
					} else if (kept[u]) {
						rare |= 1u << u;
					}
				}
			}
			if (!SPILL && rare) {
#pragma unroll 1
				for (uint32_t u = 0; u < 4u; ++u) {
					if (!((rare >> u) & 1u)) continue;
					// (the event this lane holds in place u of the group: see the load above)
					const uint32_t o = (GYS_EV_X3 && !V6 && !PF) ? ((uint32_t)g + u) * T + (tid & ~63u) + (lane >> 1) + ((lane & 1u) ? 32u : 0u) : ((uint32_t)g + u) * T + tid;
					uint32_t idx, rank;
					if (V6) {
						const uint64_t a0 = tb[GYS_EV6_WORDS * o], a1 = tb[GYS_EV6_WORDS * o + 1u], d0 = tb[GYS_EV6_WORDS * o + 2u], d1 = tb[GYS_EV6_WORDS * o + 3u],
							       x4 = tb[GYS_EV6_WORDS * o + 4u];
						const uint32_t sa[4] = {(uint32_t)a0, (uint32_t)(a0 >> 32), (uint32_t)a1, (uint32_t)(a1 >> 32)};
						const uint32_t da[4] = {(uint32_t)d0, (uint32_t)(d0 >> 32), (uint32_t)d1, (uint32_t)(d1 >> 32)};
						uint32_t w[10];
						const uint32_t nw = pair_words(ip6_embedded_v4(da), da, bswap16((uint16_t)(x4 >> 48)), ip6_embedded_v4(sa), sa, bswap16((uint16_t)(x4 >> 32)), w);
						if (SVCHLL && p.svc_hll_p) {
							const uint64_t h64 = hash64<10>(w, nw); // (== flow_hash64_v6; the per-service registers need all 64 bits)
							hll_idx_rank(h64, GYS_HLL_P, &idx, &rank);
							const uint32_t l = u == 0u ? nlr[0] : u == 1u ? nlr[1] : u == 2u ? nlr[2] : nlr[3];
							svc_hll_update(p.svc_hll, p.svc_hll_p, s_slot[l], h64);
						} else {
							words_hll_idx_rank<10>(w, nw, &idx, &rank); // the second hash half only when the first one's rank bits are all zero
						}
					} else {
						const uint64_t x0 = tb[3u * o], x1 = tb[3u * o + 1u]; // (re-read: keeps the four events' words out of this loop's registers)
						const uint32_t saddr = (uint32_t)x0, daddr = (uint32_t)(x0 >> 32);
						const uint16_t sport = bswap16((uint16_t)(x1 >> 32)), dport = bswap16((uint16_t)(x1 >> 48));
						const uint64_t h64 = flow_hash64(daddr, dport, saddr, sport);
						hll_idx_rank(h64, GYS_HLL_P, &idx, &rank);
						if (SVCHLL && p.svc_hll_p) {
							const uint32_t l = u == 0u ? nlr[0] : u == 1u ? nlr[1] : u == 2u ? nlr[2] : nlr[3];
							svc_hll_update(p.svc_hll, p.svc_hll_p, s_slot[l], h64);
						}
					}
					if (rank > hll_floor && p.hll32[idx] < rank) atomicMax(&p.hll32[idx], rank);
				}
			}
#ifdef GYS_RESP_TIMING
			asm volatile("" : "+v"(hrank[0]), "+v"(hrank[1]), "+v"(hrank[2]), "+v"(hrank[3]));
#endif
			GYS_TICK(4); // histogram adds + flow hashes
			// HLL register traffic is taken OUT of the event loop: an event whose rank exceeds the floor only queues {register, rank} in LDS;
			// the queue is drained once per tile (below: one batch of register reads, the rare atomicMax behind them).  Measured (r3m): with the
			// read-first register access inside this loop the waves of a tile waited on L2 / memory-side round trips in most groups -- 14 %
			// of the kernel, against 5 % for the hashes themselves.  A full queue (the first workgroups of a window run with floor 0: every
			// event qualifies) falls back to the access in place.
			if (DBG && (p.dbg & 32u)) { // (timing only: hashes computed, no register traffic)
				dbg_sink |= hrank[0] ^ hrank[1] ^ hrank[2] ^ hrank[3] ^ hidx[0] ^ hidx[1] ^ hidx[2] ^ hidx[3];
			} else if (!SPILL && (hrank[0] | hrank[1] | hrank[2] | hrank[3])) {
				uint32_t at[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) at[u] = hrank[u] ? atomicAdd(&s_hqn, 1u) : 0u;
				bool direct = false;
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					if (hrank[u]) {
						if (at[u] < GYS_HQ_CAP) s_hq[at[u]] = hidx[u] | (hrank[u] << 16);
						else direct = true;
					}
				}
				if (direct) {
#pragma unroll
					for (int u = 0; u < 4; ++u) hcur[u] = (hrank[u] && at[u] >= GYS_HQ_CAP) ? p.hll32[hidx[u]] : 0xFFu; // read-first: most events do not raise the register
#pragma unroll
					for (int u = 0; u < 4; ++u)
						if (hcur[u] < hrank[u]) atomicMax(&p.hll32[hidx[u]], hrank[u]);
				}
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) nlr[u] |= kept[u] ? (rk12[u] << 12) : 0u;
#if GYS_SHIFT_SWITCH
			// the group's results go to places g .. g + 3: g is uniform, so this is one scalar branch into a block of eight moves (the empty
			// asm statement keeps the blocks from being turned into 8 x TPT / 4 selects)
#pragma unroll
			for (int c = 0; c < TPT; c += 4) {
				if (g == c) {
					asm volatile("" ::: "memory");
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						wd[c + u] = nwd[u];
						lr[c + u] = nlr[u];
					}
				}
			}
#else
#pragma unroll
			for (int j = 0; j + 4 < TPT; ++j) {
				wd[j] = wd[j + 4];
				lr[j] = lr[j + 4];
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				wd[TPT - 4 + u] = nwd[u];
				lr[TPT - 4 + u] = nlr[u];
			}
#endif
		}
#if GYS_EV_PREFETCH && defined(__HIP_DEVICE_COMPILE__)
		asm volatile("" ::"v"(pf_sink)); // (the register stays reserved across the group loop)
#endif
		GYS_TICK(5); // HLL queue, results stored, loop control
		__syncthreads();
		GYS_TICK(6); // barrier behind the event phase
		// ---- the tile's queued HLL candidates: one per thread, the register read is in flight under the scan below
		uint32_t hq_e = 0, hq_cur = 0xFFu;
		if (!SPILL) {
			const uint32_t nq = min(s_hqn, (uint32_t)GYS_HQ_CAP);
			if (tid < nq) {
				hq_e = s_hq[tid];
				hq_cur = p.hll32[hq_e & 0xFFFFu];
			}
		}
		// ---- exclusive scan of the tile's per-key counts -> run starts inside the image; every key's piece gets its destination
		{
			uint32_t sum = 0;
			for (uint32_t k = klo; k < khi; ++k) sum += s_ts[k];
			const uint32_t inc = wave_incl_scan_u32(sum); // (DPP: the shuffle loop was six dependent ds_bpermute round trips per tile)
			if (lane == 63) s_wsum[wave] = inc;
			__syncthreads();
			GYS_TICK(7); // HLL candidates read, key sums, wave scan, barrier
			if (!SPILL && tid == 0) {
				s_hqn = 0; // (every thread has read the queue length; the next event phase is two barriers away)
				if (t0 + 2ull * TILE < e1) {
					s_floor[tile_no & 1u] = 0xFFFFFFFFu; // (read by every thread at the top of this tile; refilled behind this tile's flush for tile t + 2)
#if GYS_FLOOR_QUARTER
					s_fqm[tile_no & 3u] = 0xFFFFFFFFu;   // (last read behind the previous tile's flush; refilled behind this tile's)
#endif
				}
			}
			{
				uint32_t *const s_tn = s_ts2 + ((tile_no + 1u) & 1u) * Lc; // the next tile's counters: last touched before the previous tile's image barrier
				for (uint32_t k = klo; k < khi; ++k) s_tn[k] = 0;
			}
			if (!SPILL && (tile_no & 63u) == 63u && tid < 15u) { // keep the packed per-wave sums far from their 40-bit field (no wave adds to them between the event phase and the image barrier)
				unsigned long long cnt = 0, sum = 0;
				for (uint32_t w = 0; w < 16u; ++w) {
					const unsigned long long v = s_gh[w * GYS_GH_STRIDE + tid];
					cnt += v >> 40;
					sum += v & ((1ull << 40) - 1);
					s_gh[w * GYS_GH_STRIDE + tid] = 0;
				}
				if (cnt) {
					atomicAdd(&p.ghist[2 * tid], cnt);
					atomicAdd(&p.ghist[2 * tid + 1], sum);
					atomicAdd(&p.ghist[30], cnt);
				}
			}
			uint32_t run = inc - sum;
			for (uint32_t w = 0; w < wave; ++w) run += s_wsum[w];
			for (uint32_t k = klo; k < khi; ++k) {
				const uint32_t c = s_ts[k];
				s_ts[k] = c | (run << 16);
				if (c) {
					uint64_t base = ~0ull;
					if (SPILL) {
						base = (uint64_t)atomicAdd(&p.td_run[s_slot[k]], c);
					} else {
						uint32_t b = s_cur[k];
						if (b & GYS_SPILL_BIT) {
							// the key has a predicted run (k_prespill): the piece goes there, the run's cursor counts the key's values of the
							// batch; a piece past the run's end is dropped and the end-of-batch pass falls back to an exact run + second pass
							const uint32_t r = atomicAdd(&p.td_run[s_slot[k]], c);
							if (r + c <= p.td_run1[s_slot[k]]) base = (uint64_t)(p.run_delta + (long long)r);
						} else {
							if (SHARED) b = atomicAdd(&p.td_cur[s_slot[k]], c);
							else s_cur[k] = b + c;
							// a piece that does not fit is dropped: the key's count ends above pcap, k_key_finalize then spills the key
							if ((uint64_t)b + c <= (uint64_t)p.pcap) base = (uint64_t)s_slot[k] * p.pcap + b;
						}
					}
					// image entry e of this key goes to dst[base + (e - run)]: kept as the (biased, hence never zero) index entry 0 would have
					s_dst[k] = base != ~0ull ? base + GYS_DST_BIAS - run : 0ull;
				}
				run += c;
			}
		}
		__syncthreads();
		GYS_TICK(8); // run starts, destinations, barrier
		{
			uint32_t ts[TPT];
#pragma unroll
			for (int u = 0; u < TPT; ++u) ts[u] = s_ts[lr[u] & 0xFFFu]; // (all reads first: one LDS round trip for the tile's 16 events, not 16)
			GYS_MEM_FENCE();
#pragma unroll
			for (int u = 0; u < TPT; ++u) { // (no branch: a place that kept nothing writes the parking entry -- local index 0, as the flush expects there)
				const bool keep = wd[u] != GYS_EV_DROPPED && !(DBG && (p.dbg & 2u));
				const uint32_t pos = keep ? (ts[u] >> 16) + (lr[u] >> 12) : TILE;
				s_val[pos] = wd[u];
				s_key[pos] = (uint16_t)(lr[u] & 0xFFFu);
			}
		}
		__syncthreads();
		GYS_TICK(9); // image, barrier
		if (!SPILL && hq_cur < (hq_e >> 16)) atomicMax(&p.hll32[hq_e & 0xFFFFu], hq_e >> 16); // (the register read has had two phases to arrive)
		{
			uint32_t ntile = 0; // kept events of the tile (every thread computes it from the wave sums)
#pragma unroll
			for (uint32_t w = 0; w < T / 64; ++w) ntile += s_wsum[w];
			if (DBG && (p.dbg & 3u)) ntile = 0;
			// consecutive image entries -> consecutive lanes -> consecutive addresses inside a key's piece.  8 entries per thread and round:
			// keys and values of all eight, then the eight destinations, then the stores (two LDS round trips per round, not three per entry)
			constexpr int FU = 8;
#if GYS_FLOOR_QUARTER
			constexpr uint32_t NQ = (1u << GYS_HLL_P) / 16u / T; // 16-byte pieces of a quarter of the register file per thread (1 or 2)
			uint4 fq[NQ];
			const bool fq_on = !SPILL && t0 + 2ull * TILE < e1;
			if (fq_on) {
#pragma unroll
				for (uint32_t j = 0; j < NQ; ++j) fq[j] = ((const uint4 *)p.hll32)[(tile_no & 3u) * ((1u << GYS_HLL_P) / 16u) + j * T + tid];
			}
#endif
			for (uint32_t eb0 = 0; eb0 < ntile; eb0 += (uint32_t)FU * T) {
				uint32_t kk[FU], vv[FU];
				uint64_t dd[FU];
#pragma unroll
				for (int j = 0; j < FU; ++j) {
					const uint32_t e = eb0 + (uint32_t)j * T + tid;
					const uint32_t ee = e < ntile ? e : TILE; // (past the end: the parking entry)
					kk[j] = s_key[ee];
					vv[j] = s_val[ee];
				}
				GYS_MEM_FENCE();
#pragma unroll
				for (int j = 0; j < FU; ++j) dd[j] = s_dst[kk[j]];
				GYS_MEM_FENCE();
#pragma unroll
				for (int j = 0; j < FU; ++j) {
					const uint32_t e = eb0 + (uint32_t)j * T + tid;
					if (e < ntile && dd[j]) dstb[dd[j] + e] = vv[j];
				}
			}
#if GYS_FLOOR_QUARTER
			if (fq_on) {
				// quarter q of the register file was read by this tile; the slot (tile_no & 1) serves tile t + 2: min of this quarter's minimum
				// and the running minima of the other quarters (s_fq[]: each at most four tiles old -- a lower bound all the same)
				uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
				for (uint32_t j = 0; j < NQ; ++j) mn = min(min(mn, min(fq[j].x, fq[j].y)), min(fq[j].z, fq[j].w));
				mn = wave_min_u32(mn);
				if (lane == 0) {
					const uint32_t qn = tile_no & 3u;
					atomicMin(&s_fqm[qn], mn); // (reset behind this tile's scan barrier; read as "another quarter" by the next three tiles)
					atomicMin(&s_floor[tile_no & 1u], min(min(mn, s_fqm[(qn + 1u) & 3u]), min(s_fqm[(qn + 2u) & 3u], s_fqm[(qn + 3u) & 3u])));
				}
			}
#else
			// the next tile's HLL floor: the whole register file is re-read (L2 hits: every workgroup reads the same 64 KiB; behind the flush --
			// held across it, the 16 registers of the four loads spill)
			if (!SPILL && t0 + 2ull * TILE < e1) {
				constexpr uint32_t NF = (1u << GYS_HLL_P) / 4u / T;
				uint4 fv[NF];
#pragma unroll
				for (uint32_t j = 0; j < NF; ++j) fv[j] = ((const uint4 *)p.hll32)[tid + j * T];
				uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
				for (uint32_t j = 0; j < NF; ++j) mn = min(min(mn, min(fv[j].x, fv[j].y)), min(fv[j].z, fv[j].w));
				mn = wave_min_u32(mn);
				if (lane == 0) atomicMin(&s_floor[tile_no & 1u], mn);
			}
#endif
		}
		GYS_TICK(11); // flush + floor refresh
		// (the next tile's event-phase barrier orders this tile's flush before destinations and image are rewritten)
	}
#ifdef GYS_RESP_TIMING
	if (!SPILL && lane == 0) {
#pragma unroll
		for (int i = 0; i < 12; ++i) atomicAdd(&g_resp_timing[i], (unsigned long long)tacc[i]);
		atomicAdd(&g_resp_timing[15], 1ull);
	}
#endif
	if (SPILL) return;
	if (DBG && dbg_sink == 0xDEADBEEFu) p.counters[CTR_RESP_EVENTS] = 1; // (keeps the hashes of the timing-only variant alive)
	if (wmax >= 0) tmax = wmax >> GYS_ROW_BITS;
	tmax = wave_max_i32(tmax);
	if (lane == 0 && tmax != INT32_MIN) atomicMax(&s_gmax, tmax);
	if (ndrop_range) atomicAdd(&s_drop[0], ndrop_range);
	if (ndrop_nol) atomicAdd(&s_drop[1], ndrop_nol);
	__syncthreads();
	if (tid < 64u) { // (wave 0: the 15 bucket sums, and their total with ONE add -- fifteen adds on the one total cell per workgroup were 1.5 x 10^5 per batch on one address)
		unsigned long long cnt = 0, sum = 0;
		if (tid < 15u) {
			for (uint32_t w = 0; w < 16u; ++w) {
				const unsigned long long v = s_gh[w * GYS_GH_STRIDE + tid];
				cnt += v >> 40;
				sum += v & ((1ull << 40) - 1);
			}
			if (cnt) {
				atomicAdd(&p.ghist[2 * tid], cnt);
				atomicAdd(&p.ghist[2 * tid + 1], sum);
			}
		}
		unsigned long long tot = cnt;
#pragma unroll
		for (int d = 8; d >= 1; d >>= 1) tot += (unsigned long long)__shfl_xor((long long)tot, d, 64);
		if (tid == 0 && tot) atomicAdd(&p.ghist[30], tot);
	}
	if (!SHARED) { // the host's keys are this workgroup's alone: their end-of-batch bookkeeping happens here (no pass over all services)
		constexpr int KMAX = (int)(2048u / T); // (GYS_HOST_MAX_LOCAL listeners per sub-table at most)
		bool fvalid[KMAX];
		uint32_t fkey[KMAX], fcur[KMAX];
#pragma unroll
		for (int j = 0; j < KMAX; ++j) {
			const uint32_t k = (uint32_t)j * T + tid;
			fvalid[j] = k < L;
			fkey[j] = fvalid[j] ? s_slot[k] : 0u;
			fcur[j] = fvalid[j] ? s_cur[k] : 0u;
		}
		finalize_keys_wg<true, KMAX>(p.fin, &s_fin, tid, lane, fvalid, fkey, fcur);
	}
	if (tid == 0) {
		if (s_gmax != INT32_MIN) atomicMax(p.gmax, (long long)s_gmax);
		if (hd.part == 0u) atomicAdd((unsigned long long *)&p.counters[CTR_RESP_EVENTS], (unsigned long long)(e1 - e0));
		if (s_drop[0]) atomicAdd((unsigned long long *)&p.counters[CTR_RESP_DROP_RANGE], (unsigned long long)s_drop[0]);
		if (s_drop[1]) atomicAdd((unsigned long long *)&p.counters[CTR_RESP_DROP_NOLISTENER], (unsigned long long)s_drop[1]);
	}
}

// ---------------------------------------------------------------------------------------------------- general front end -> buffers
// The general pipeline (k_resp_pass1 / scan / k_resp_scatter) leaves every key's batch values as a run staged[off_end - m .. off_end).
// Runs that fit are copied behind the key's buffered values; for the others only the count moves (k_key_finalize then spills the key
// and the merge reads the run where it lies).  One wave per chunk of 64 keys, the wave copies one key's run at a time.
struct AppendP {
	uint32_t *batch_cnt;
	const uint32_t *off_end;
	const uint32_t *staged;
	uint32_t *td_cur;
	uint32_t *td_pend;
	uint32_t pcap, nsvc;
};

__global__ __launch_bounds__(256) void k_key_append(AppendP p)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t nchunks = (p.nsvc + 63u) / 64u, nwaves = gridDim.x * 4u;
	for (uint32_t chunk = blockIdx.x * 4u + (threadIdx.x >> 6); chunk < nchunks; chunk += nwaves) {
		const uint32_t key = chunk * 64u + lane;
		const uint32_t m = key < p.nsvc ? p.batch_cnt[key] : 0u;
		unsigned long long todo = __ballot(m != 0);
		if (!todo) continue;
		const uint32_t oend = m ? p.off_end[key] : 0u;
		const uint32_t cur = m ? p.td_cur[key] : 0u;
		if (m) {
			p.batch_cnt[key] = 0;
			p.td_cur[key] = cur + m;
		}
		while (todo) {
			const int k = __ffsll((long long)todo) - 1;
			todo &= todo - 1;
			const uint32_t km = (uint32_t)__shfl((int)m, k, 64), ke = (uint32_t)__shfl((int)oend, k, 64), kc = (uint32_t)__shfl((int)cur, k, 64);
			if ((uint64_t)kc + km > (uint64_t)p.pcap) continue; // spilled: the run stays in `staged`
			const uint32_t *src = p.staged + (ke - km);
			uint32_t *dst = p.td_pend + (size_t)(chunk * 64u + (uint32_t)k) * p.pcap + kc;
			for (uint32_t i = lane; i < km; i += 64u) dst[i] = src[i];
		}
	}
}

// ---------------------------------------------------------------------------------------------------- fold + t-digest merge
struct DigestP {
	int64_t *td_sum;    // [nsvc*NB]
	uint32_t *td_cnt;   // [nsvc*NB]
	TdMeta *td_meta;    // [nsvc]
	int2 *td_minmax;    // [nsvc] smallest / largest value over merged AND folded values
	uint32_t *td_pend;  // [nsvc*pcap] staged words
	uint32_t *td_cur;
	uint32_t pcap, nsvc;
	uint32_t pend_cap;  // values a key buffers between batches at most (gys_config.td_pend_cap)
	const uint32_t *staged;
	gys_hist_rec *hist_win, *hist_all;
	uint32_t *bitmap;   // [nsvc*GYS_BM_WORDS] u32 = 32 x u16 CONN_BITMAP rows of resp_bitmap_v4_, then of resp_bitmap_v6_ (common/gy_socket_stat.h:390-454)
};

// wave-synchronous LDS hand-off: DS operations of one wave execute in order; this only stops the compiler from moving them
#define GYS_WAVE_SYNC()                                              \
	do {                                                         \
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
		__builtin_amdgcn_wave_barrier();                     \
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
	} while (0)

#define GYS_PACK_ONE (1ull << 40)              // packed {count:24 | sum:40} accumulators: one LDS atomic per value
#define GYS_PACK_SUM(v) ((v) & (GYS_PACK_ONE - 1))

// Brings the records of a key up to date with the not yet folded words of its buffer (and, inside a merge, of its run).  `da` / `dw`
// are the packed bucket deltas of ALL not yet folded values / of those that belong to window mt.win_epoch; lane g (0..15) of the
// calling group owns histogram pair g and bitmap word g:
//   all-time record += da (GY_HISTOGRAM::add_data for every value, common/gy_statistics.h:596-623);
//   window record: a record of an older window is dropped first (the reference clears the 5-s state on its timer,
//   GY_HISTOGRAM::clear :630-636; here a key rolls when the first values of a later window are folded), then += dw; same for the rows.
__device__ __forceinline__ void fold_records(const DigestP &p, uint32_t slot, uint32_t g, bool roll, unsigned long long da, unsigned long long dw,
					     uint32_t n_all, uint32_t n_win, int32_t max_all, int32_t max_win, uint32_t bm, uint32_t bm6)
{
	uint4 *ap = (uint4 *)&p.hist_all[slot] + g, *wp = (uint4 *)&p.hist_win[slot] + g;
	if (n_all) {
		const uint4 a = *ap;
		uint64_t lo = (uint64_t)a.x | ((uint64_t)a.y << 32), hi = (uint64_t)a.z | ((uint64_t)a.w << 32);
		if (g < 15u) {
			lo += da >> 40;
			hi += GYS_PACK_SUM(da);
		} else {
			lo += n_all;                                                    // total_count_
			if ((int64_t)hi < (int64_t)max_all) hi = (uint64_t)(int64_t)max_all; // max_val_seen_
		}
		if (g == 15u || da) *ap = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
	}
	if (n_win) {
		uint64_t lo = 0, hi = g < 15u ? 0ull : (uint64_t)INT64_MIN;
		if (!roll) {
			const uint4 w = *wp;
			lo = (uint64_t)w.x | ((uint64_t)w.y << 32);
			hi = (uint64_t)w.z | ((uint64_t)w.w << 32);
		}
		if (g < 15u) {
			lo += dw >> 40;
			hi += GYS_PACK_SUM(dw);
		} else {
			lo += n_win;
			if ((int64_t)hi < (int64_t)max_win) hi = (uint64_t)(int64_t)max_win;
		}
		if (roll || g == 15u || dw) *wp = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
		uint32_t *bp = &p.bitmap[(size_t)slot * GYS_BM_WORDS + g];
		const uint32_t old = roll ? 0u : *bp;
		if (roll || (old | bm) != old) *bp = old | bm;
		// the IPv6 rows (resp_bitmap_v6_) sit behind the IPv4 ones in the same 128-byte line; a roll has to clear what an earlier window left
		const uint32_t old6 = bp[16];
		const uint32_t new6 = (roll ? 0u : old6) | bm6;
		if (new6 != old6) bp[16] = new6;
	}
}

// ---- k_fold: services [first, first + n): every key with not yet folded words gets its records brought up to date.  FOUR keys per
// wave: each 16-lane row owns one key -- lane g of the row holds the key's histogram pair g (16 x {count,sum} = the 256-byte record,
// one coalesced 256-B access per row), its CONN_BITMAP word g, and word g of every 16-word group of the buffer.
struct FoldP {
	DigestP d;
	uint32_t first, n;
	// window close with the 5-s level (gys_config.enable_levels = 1; nullptr otherwise): the pass visits every service of the range anyway, so it
	// also leaves, per service, the window its record in hist_win belongs to once it is folded (last_tag[]) -- the engine then SWAPS hist_win and
	// the level-0 array instead of copying 256 bytes per service (k_level_roll's pass of rounds 2 - 4; a key's first fold of the next window
	// rewrites the record without reading it: `roll` in fold_records) -- and the time of a service's first window close (first_sec[]).
	uint32_t *last_tag;
	int64_t *first_sec;
	int64_t tnow;
};

__global__ __launch_bounds__(256) void k_fold(FoldP q)
{
	const DigestP &p = q.d;
	__shared__ unsigned long long s_a_[16][16], s_w_[16][16];
	__shared__ uint32_t s_bm_[16][GYS_BM_WORDS];
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	const uint32_t row = lane >> 4, g = lane & 15u;
	unsigned long long *s_a = s_a_[wv * 4u + row], *s_w = s_w_[wv * 4u + row];
	uint32_t *s_bm = s_bm_[wv * 4u + row];
	const uint32_t nchunks = (q.n + 63u) / 64u, nwaves = gridDim.x * 4u;
	for (uint32_t chunk = blockIdx.x * 4u + wv; chunk < nchunks; chunk += nwaves) {
		const uint32_t rel = chunk * 64u + lane;
		uint4 mraw = make_uint4(0, 0, 0, 0);
		if (rel < q.n) mraw = *(const uint4 *)&p.td_meta[q.first + rel];
		const unsigned long long todo = __ballot(mraw.x > (mraw.y & 0xFFFFu));
		if (todo) {
			for (uint32_t rd = 0; rd < 16u; ++rd) {
				if (!((todo >> (4u * rd)) & 0xFull)) continue;
				const uint32_t k = rd * 4u + row;
				const uint32_t slot = q.first + chunk * 64u + k;
				uint4 mt;
				mt.x = (uint32_t)__shfl((int)mraw.x, (int)k, 64);
				mt.y = (uint32_t)__shfl((int)mraw.y, (int)k, 64);
				mt.z = (uint32_t)__shfl((int)mraw.z, (int)k, 64);
				mt.w = (uint32_t)__shfl((int)mraw.w, (int)k, 64);
				const uint32_t npend = mt.x, nh = mt.y & 0xFFFFu, nw = mt.y >> 16;
				const uint32_t nwin0 = max(nh, nw); // first word of the not yet folded part that belongs to window win_epoch
				const uint32_t m = npend > nh ? npend - nh : 0u;
				s_a[g] = 0;
				s_w[g] = 0;
				s_bm[g] = 0;
				s_bm[g + 16u] = 0;
				GYS_WAVE_SYNC();
				const uint32_t *pend = p.td_pend + (size_t)slot * p.pcap;
				const uint32_t mmax = max(max((uint32_t)__shfl((int)m, 0, 64), (uint32_t)__shfl((int)m, 16, 64)),
							  max((uint32_t)__shfl((int)m, 32, 64), (uint32_t)__shfl((int)m, 48, 64)));
				int32_t lmin = INT32_MAX, lmax = INT32_MIN, wmax = INT32_MIN;
				const uint32_t n_win = npend > nwin0 ? npend - nwin0 : 0u;
				const bool roll = mt.w != mt.z;
				for (uint32_t base = 0; base < mmax; base += 16u) {
					const uint32_t i = nh + base + g;
					if (base + g < m) {
						const uint32_t w = pend[i];
						const int32_t v = (int32_t)(w >> GYS_ROW_BITS);
						const uint32_t b = resp_bucket((int64_t)v);
						const unsigned long long one = GYS_PACK_ONE | (unsigned long long)(uint32_t)v;
						atomicAdd(&s_a[b], one);
						lmin = min(lmin, v);
						lmax = max(lmax, v);
						if (i >= nwin0) {
							atomicAdd(&s_w[b], one);
							const uint32_t r = w & GYS_ROW_MASK; // CONN_BITMAP::add_response: respmap_[row].set(bucket) (common/gy_socket_stat.h:403-410); rows 32..63: resp_bitmap_v6_
							atomicOr(&s_bm[r >> 1], (1u << b) << ((r & 1u) * 16u));
							wmax = max(wmax, v);
						}
					}
				}
#pragma unroll
				for (int d = 8; d >= 1; d >>= 1) {
					lmin = min(lmin, __shfl_xor(lmin, d, 64));
					lmax = max(lmax, __shfl_xor(lmax, d, 64));
					wmax = max(wmax, __shfl_xor(wmax, d, 64));
				}
				GYS_WAVE_SYNC();
				if (m) {
					fold_records(p, slot, g, roll, s_a[g], s_w[g], m, n_win, lmax, wmax, s_bm[g], s_bm[g + 16u]);
					if (g == 0) {
						*(uint4 *)&p.td_meta[slot] = make_uint4(npend, npend | (nw << 16), mt.z, n_win ? mt.z : mt.w);
						const int2 mm = p.td_minmax[slot];
						if (lmin < mm.x || lmax > mm.y) p.td_minmax[slot] = make_int2(min(mm.x, lmin), max(mm.y, lmax));
					}
				}
				GYS_WAVE_SYNC();
			}
		}
		if (q.last_tag && rel < q.n) {
			// lane l: the window the record of service chunk * 64 + l belongs to now -- the rounds' `hw_now`, from the lane's own meta word
			const uint32_t nh = mraw.y & 0xFFFFu, nwin0 = max(nh, mraw.y >> 16);
			q.last_tag[q.first + rel] = mraw.x > nh && mraw.x > nwin0 ? mraw.z : mraw.w;
			if (q.first_sec[q.first + rel] == 0) q.first_sec[q.first + rel] = q.tnow; // BucketedTimeSeries::update on an empty series
		}
	}
}

__device__ __forceinline__ uint32_t td_cluster_of(const uint64_t *T, uint64_t mid2)
{
	uint32_t a = 0, bb = GYS_TD_NB - 1;
	while (a < bb) {
		const uint32_t mid = (a + bb + 1) >> 1;
		if (mid2 >= T[mid]) a = mid; else bb = mid - 1;
	}
	return a;
}

// quarter-octave grid cell of a value < 2^20: 0,1,2,3 for 0..3, then 4 cells per power of two (monotone; <= 75)
__device__ __forceinline__ uint32_t value_grid(uint32_t v)
{
	if (v < 4u) return v;
	const uint32_t msb = 31u - (uint32_t)__clz((int)v);
	return 4u * (msb - 1u) + ((v >> (msb - 2u)) & 3u);
}
#define GYS_IVL 320u // refined intervals: gap index (<= 200) + grid cell (<= 75) < 320 = 5 per lane of one wave
#define GYS_NBP 256u // cluster arrays padded to a power of two for the branch-free searches

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

// ---- k_digest_merge: one workgroup of NT threads per merge-list entry.  Exact-integer k-bucket merge (DESIGN.md "t-digest"):
//   values = the key's buffered words (+ its run when the key spilled); merged order = by mean, old clusters before values on ties; an
//   item with weighted mid-point mid2/2 of N goes to cluster #{j : mid2 >= T_j}, T_j = ceil(BND[j] * 2N / 2^32).
// No sort: the (<= 200) old cluster means cut the value axis into gaps; gap(v) = #{clusters with mean <= v} comes from a branch-free
// search; the gaps are refined by a fixed quarter-octave value grid (so a cell stays small even when the digest is empty or the
// distribution has moved away from its clusters); a counting sort by refined interval groups the values in LDS, and a value's rank
// is (values in lower intervals) + (its rank inside its own interval, by direct comparison -- an interval holds a handful of the
// values, all equal for typical integer-ms data).  Old cluster c is preceded by exactly the values of gaps 0..c.  Everything else is
// integer arithmetic on ranks, so the result equals the sorted-merge definition bit for bit (ties among equal values are interchangeable).
// The not yet folded values of the key are folded into its records on the way (they leave the buffer here).
// query mode (out_sum != nullptr): entry w writes the merged view of its key to out_sum/out_cnt[w*NB..] and leaves the state alone.
struct MergeP {
	DigestP d;
	const MergeEnt *list;
	const uint32_t *count;
	int64_t *out_sum;
	uint32_t *out_cnt;
};

template <uint32_t MAXV, uint32_t NT>
__global__ __launch_bounds__(NT) void k_digest_merge(MergeP q)
{
	const DigestP &p = q.d;
	__shared__ uint32_t s_x[MAXV];            // interval << 20 | value, in arrival order
	__shared__ uint32_t s_g[MAXV];            // the same words grouped by interval
	__shared__ uint32_t s_thr[GYS_NBP];       // compacted non-empty old clusters: ceil(sum / count), padded with ~0 for the branch-free search
	__shared__ uint64_t s_cpfx[GYS_NBP + 1];  // old weight before compacted cluster c
	__shared__ uint64_t s_T[GYS_NBP];         // s_T[j], j = 1..NB-1; [0] = 0, [NB..] = ~0 (never reached)
	__shared__ uint32_t s_imin[GYS_IVL], s_imax[GYS_IVL]; // smallest / largest value of each interval
	__shared__ unsigned long long s_osum[GYS_NBP];
	__shared__ uint32_t s_ocnt[GYS_NBP];
	__shared__ uint32_t s_icnt[GYS_IVL];      // values per (refined) interval, then the running scatter cursor
	__shared__ uint32_t s_ioff[GYS_IVL + 1];  // exclusive prefix of s_icnt (s_ioff[i + 1] = values in intervals 0..i)
	__shared__ uint32_t s_clt[GYS_IVL];       // s_clt[c + 1] = values below the mean of compacted cluster c (prefix of the per-cluster-gap counts)
	__shared__ unsigned long long s_fa[16], s_fw[16]; // fold: packed bucket deltas of the not yet folded values (all / window part)
	__shared__ uint32_t s_fbm[GYS_BM_WORDS];
	__shared__ int32_t s_fmm[3];              // fold: min, max (all), max (window part)
	__shared__ uint32_t s_wv[2 * (NT / 64) + 2];
	__shared__ uint64_t s_ww[NT / 64];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t nent = *q.count;
	const bool query = q.out_sum != nullptr;

	for (uint32_t w = blockIdx.x; w < nent; w += gridDim.x) {
		const MergeEnt ent = q.list[w];
		const uint32_t m = ent.nbuf + ent.mrun;
		if (m > MAXV) continue; // never queued on this class's list (finalize_key)
		const uint32_t slot = ent.slot;
		const uint4 mt = *(const uint4 *)&p.td_meta[slot];
		const uint32_t nh = query ? ent.nbuf : (mt.y & 0xFFFFu), nw = mt.y >> 16;
		const uint32_t nwin0 = max(nh, nw);
		const uint32_t run0 = ent.off_end - ent.mrun;

		// ---- old digest: thread t < 256 holds cluster t
		const int64_t *gs = p.td_sum + (size_t)slot * GYS_TD_NB;
		const uint32_t *gc = p.td_cnt + (size_t)slot * GYS_TD_NB;
		uint32_t c0 = 0;
		int64_t sm0 = 0;
		if (tid < GYS_TD_NB) {
			c0 = gc[tid];
			sm0 = gs[tid];
		}
		if (m == 0) { // nothing buffered (query of a freshly merged key): the merged view is the digest itself
			if (query && tid < GYS_TD_NB) {
				q.out_sum[(size_t)w * GYS_TD_NB + tid] = sm0;
				q.out_cnt[(size_t)w * GYS_TD_NB + tid] = c0;
			}
			continue;
		}
		// compaction of the non-empty clusters (order preserving) + exclusive prefix of their weights, over the first four waves
		uint32_t pos0 = 0, nc = 0;
		uint64_t e0 = 0, nold = 0;
		{
			const unsigned long long b0 = __ballot(c0 != 0);
			const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
			const uint64_t inc = wave_incl_scan_u64(c0);
			if (lane == 63u) s_ww[wave] = inc;
			if (lane == 0u) s_wv[wave] = (uint32_t)__popcll(b0);
			__syncthreads();
			uint32_t pbase = 0;
			uint64_t wbase = 0;
#pragma unroll
			for (uint32_t k = 0; k < 4u; ++k) {
				if (k < wave) {
					pbase += s_wv[k];
					wbase += s_ww[k];
				}
				nc += s_wv[k];
				nold += s_ww[k];
			}
			pos0 = pbase + (uint32_t)__popcll(b0 & below);
			e0 = wbase + inc - c0;
		}
		const uint64_t twoN = 2ull * (nold + (uint64_t)m);
		if (tid < GYS_NBP) {
			s_thr[tid] = 0xFFFFFFFFu; // pad: mean = +inf
			s_T[tid] = (tid >= 1u && tid < GYS_TD_NB) ? td_threshold(c_td_bnd[tid], twoN) : (tid ? ~0ull : 0ull);
			s_osum[tid] = 0;
			s_ocnt[tid] = 0;
		}
		for (uint32_t i = tid; i < GYS_IVL; i += NT) {
			s_icnt[i] = 0;
			s_imin[i] = 0xFFFFFFFFu;
			s_imax[i] = 0;
			s_clt[i] = 0;
		}
		if (tid < 16u) {
			s_fa[tid] = 0;
			s_fw[tid] = 0;
			s_fbm[tid] = 0;
			s_fbm[tid + 16u] = 0;
		}
		if (tid == 0) {
			s_fmm[0] = INT32_MAX;
			s_fmm[1] = INT32_MIN;
			s_fmm[2] = INT32_MIN;
		}
		__syncthreads();
		if (c0) {
			s_thr[pos0] = ceil_div_sum_cnt(sm0, c0);
			s_cpfx[pos0] = e0;
		}
		if (tid == 0) s_cpfx[nc] = nold;
		__syncthreads();
		// ---- values (buffered, then the run): gap = #{clusters with mean <= v} = #{thresholds <= v}
		{
			const uint32_t *pend = p.td_pend + (size_t)slot * p.pcap;
			int32_t lmin = INT32_MAX, lmax = INT32_MIN, wmax = INT32_MIN;
			for (uint32_t i = tid; i < m; i += NT) {
				const uint32_t word = i < ent.nbuf ? pend[i] : p.staged[run0 + (i - ent.nbuf)];
				const uint32_t uv = word >> GYS_ROW_BITS;
				uint32_t lo = 0;
#pragma unroll
				for (uint32_t step = GYS_NBP / 2; step >= 1u; step >>= 1)
					if (s_thr[lo + step - 1u] <= uv) lo += step;
				// refined interval: the cluster means AND a fixed quarter-octave value grid cut the axis (both monotone in v, so
				// their sum numbers the cells of the common refinement in value order)
				const uint32_t iv = lo + value_grid(uv);
				s_x[i] = (iv << 20) | uv;
				atomicAdd(&s_icnt[iv], 1u);
				atomicMin(&s_imin[iv], uv);
				atomicMax(&s_imax[iv], uv);
				atomicAdd(&s_clt[lo + 1], 1u);
				if (i >= nh) { // not yet folded: histogram bucket of the key's records, CONN_BITMAP row, min / max
					const uint32_t b = resp_bucket((int64_t)uv);
					const unsigned long long one = GYS_PACK_ONE | (unsigned long long)uv;
					atomicAdd(&s_fa[b], one);
					lmin = min(lmin, (int32_t)uv);
					lmax = max(lmax, (int32_t)uv);
					if (i >= nwin0) {
						atomicAdd(&s_fw[b], one);
						const uint32_t r = word & GYS_ROW_MASK;
						atomicOr(&s_fbm[r >> 1], (1u << b) << ((r & 1u) * 16u));
						wmax = max(wmax, (int32_t)uv);
					}
				}
			}
			if (!query && m > nh) {
				lmin = wave_min_i32(lmin);
				lmax = wave_max_i32(lmax);
				wmax = wave_max_i32(wmax);
				if (lane == 0) {
					if (lmin != INT32_MAX) atomicMin(&s_fmm[0], lmin);
					if (lmax != INT32_MIN) atomicMax(&s_fmm[1], lmax);
					if (wmax != INT32_MIN) atomicMax(&s_fmm[2], wmax);
				}
			}
		}
		__syncthreads();
		// ---- exclusive scan of the refined-interval counts and inclusive scan of the per-cluster-gap counts (GYS_IVL = 5 x 64
		// entries each: wave 0 takes 5 consecutive entries per lane of both arrays)
		if (wave == 0) {
			uint32_t t[5], gcl[5];
			uint32_t own = 0, gown = 0;
#pragma unroll
			for (int k = 0; k < 5; ++k) {
				t[k] = s_icnt[5u * lane + k];
				gcl[k] = s_clt[5u * lane + k];
				own += t[k];
				gown += gcl[k];
			}
			const uint32_t inc = wave_incl_scan_u32(own), ginc = wave_incl_scan_u32(gown);
			uint32_t ex = inc - own, gex = ginc - gown;
#pragma unroll
			for (int k = 0; k < 5; ++k) {
				s_ioff[5u * lane + k] = ex;
				s_icnt[5u * lane + k] = ex; // scatter cursors
				ex += t[k];
				gex += gcl[k];
				s_clt[5u * lane + k] = gex;
			}
			if (lane == 63u) s_ioff[GYS_IVL] = inc;
		}
		__syncthreads();
		for (uint32_t i = tid; i < m; i += NT) {
			const uint32_t x = s_x[i];
			s_g[atomicAdd(&s_icnt[x >> 20], 1u)] = x;
		}
		__syncthreads();
		// ---- old clusters (one per thread, from registers): preceded by the old weight before them and by the values of gaps 0..c
		// (= values below the mean)
		if (c0) {
			const uint64_t mid2 = 2ull * (e0 + (uint64_t)s_clt[pos0 + 1]) + (uint64_t)c0;
			uint32_t a = 0;
#pragma unroll
			for (uint32_t step = GYS_NBP / 2; step >= 1u; step >>= 1)
				if (mid2 >= s_T[a + step]) a += step;
			atomicAdd(&s_osum[a], (unsigned long long)sm0);
			atomicAdd(&s_ocnt[a], c0);
		}
		// ---- values: rank = values in lower intervals + rank inside the interval (ties by position); W adds the old weight <= v
		for (uint32_t e = tid; e < m; e += NT) {
			const uint32_t x = s_g[e];
			const uint32_t iv = x >> 20;
			uint32_t r = e; // an interval of equal values (the usual case for integer ms data): ties rank by position
			if (s_imin[iv] != s_imax[iv]) {
				const uint32_t gb = s_ioff[iv], ge = s_ioff[iv + 1];
				r = gb;
				for (uint32_t t = gb; t < ge; ++t) {
					const uint32_t y = s_g[t];
					r += (y < x || (y == x && t < e)) ? 1u : 0u;
				}
			}
			const uint64_t mid2 = 2ull * ((uint64_t)r + s_cpfx[iv - value_grid(x & 0xFFFFFu)]) + 1ull; // old weight with mean <= v
			uint32_t a = 0;
#pragma unroll
			for (uint32_t step = GYS_NBP / 2; step >= 1u; step >>= 1)
				if (mid2 >= s_T[a + step]) a += step;
			atomicAdd(&s_osum[a], (unsigned long long)(x & 0xFFFFFu));
			atomicAdd(&s_ocnt[a], 1u);
		}
		__syncthreads();
		// ---- write back
		if (tid < GYS_TD_NB) {
			int64_t *ws = query ? q.out_sum + (size_t)w * GYS_TD_NB : p.td_sum + (size_t)slot * GYS_TD_NB;
			uint32_t *wc = query ? q.out_cnt + (size_t)w * GYS_TD_NB : p.td_cnt + (size_t)slot * GYS_TD_NB;
			ws[tid] = (int64_t)s_osum[tid];
			wc[tid] = s_ocnt[tid];
		}
		if (!query) {
			const uint32_t n_all = m - nh, n_win = m > nwin0 ? m - nwin0 : 0u;
			if (tid < 16u && n_all) fold_records(p, slot, tid, mt.w != mt.z, s_fa[tid], s_fw[tid], n_all, n_win, s_fmm[1], s_fmm[2], s_fbm[tid], s_fbm[tid + 16u]);
			if (tid == 16u) {
				*(uint4 *)&p.td_meta[slot] = make_uint4(0u, 0u, mt.z, n_win ? mt.z : mt.w); // buffer drained
				p.td_cur[slot] = 0;
				if (n_all) {
					const int2 mm = p.td_minmax[slot];
					if (s_fmm[0] < mm.x || s_fmm[1] > mm.y) p.td_minmax[slot] = make_int2(min(mm.x, s_fmm[0]), max(mm.y, s_fmm[1]));
				}
			}
		}
		__syncthreads();
	}
}

// ---------------------------------------------------------------------------------------------------- t-digest merge (value bins)
// The hot merge: entries of size class 0 (m <= 1024 values), one 256-thread workgroup per entry, every thread keeps its (up to) four
// values in registers.  Same exact-integer definition as k_digest_merge (bit-identical result), computed by counting over the VALUE
// DOMAIN instead of over cluster gaps: integer-millisecond response times are small, so values below 1024 get one LDS bin each
// (rank = values in lower bins + arrival order inside the bin: equal values are interchangeable) and larger values fall into
// 64-cells-per-octave bins and rank inside their cell by direct comparison against the (short) list of large values.  A bin word is
// {values : 16 | clusters whose integer mean threshold lies in the bin : 16}; ONE prefix scan of the bins therefore yields, per bin,
// the values below it (-> a value's rank, a cluster's count of smaller values) and the clusters at or below it (-> a value's gap,
// i.e. the old weight that precedes it).  Per value: one LDS atomic, one LDS read, one threshold search; no sort, no per-interval
// comparison loops, no staging of the values in LDS.  The all-time record's bucket deltas come from the scanned bins (a thread's 8
// consecutive bins touch at most two RESP_TIME_HASH buckets) instead of one contended LDS atomic per value.
//   Weights are 32-bit here (total weight < 2^31: thresholds and mid-points fit a u32); an entry beyond that is handed to the
//   general kernel through slow_list.
#ifndef GYS_MB_PACKED
#define GYS_MB_PACKED 1 // values accumulate into a packed {count, sum} word: one LDS atomic per value instead of two (r3c: -0.09 ms per window)
#endif
#ifndef GYS_MB_SKIP
#define GYS_MB_SKIP 0 // TIMING EXPERIMENTS ONLY (results are wrong): 1 no per-bin pass, 2 no old-cluster / large-value placement, 4 no value pass 1,
#endif                //   8 no bin scan, 16 no write-back, 32 nothing after the loads
#define GYS_MB_EXACT 1024u
#define GYS_MB_BINS 2048u // 1024 one-value bins + 10 octaves x 64 cells (values < 2^20), padded to 8 bins per thread
#define GYS_MB_BPT 8u

__device__ __forceinline__ uint32_t mb_bin(uint32_t v)
{
	if (v < GYS_MB_EXACT) return v;
	const uint32_t msb = 31u - (uint32_t)__clz((int)v);
	return GYS_MB_EXACT + (msb - 10u) * 64u + ((v >> (msb - 6u)) & 63u);
}

struct MergeBP {
	DigestP d;
	const MergeEnt *list;
	const uint32_t *count;
	int64_t *out_sum; // query mode (see k_digest_merge)
	uint32_t *out_cnt;
	MergeEnt *slow_list;
	uint32_t *slow_count;
	// SCAN: every service [0, nsvc) instead of a list; nothing is modified; quantile i of the service's merged view goes to
	// qout[slot * nq + i] (TCP_SOCK_HANDLER::listener_stats_update produces p25 / p95 / p99 for EVERY listener every 5 s,
	// common/gy_socket_stat.cc:4044-4365: the per-key scan, here on the digests)
	const double *qs;
	uint32_t nq;
	double *qout;
};

// quantile of compacted clusters (c_cnt / c_sum / c_wb = weight before, nc of them, N in total): the host's td_quantile_interp
// (gys_engine.hip) and the oracle's (oracle/gy_oracle.c:669-715) term by term -- only + - * / on doubles, so the three agree bit for bit
__device__ __forceinline__ double td_quantile_dev(const uint32_t *c_cnt, const unsigned long long *c_sum, const uint32_t *c_wb, uint32_t nc, uint32_t N,
						  int32_t vmin, int32_t vmax, double q)
{
	if (!N) return 0.0;
	if (q < 0.0) q = 0.0;
	if (q > 1.0) q = 1.0;
	const double t = q * (double)N;
	uint32_t lo = 0, hi = nc; // first cluster whose centre lies above t
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (t < (double)c_wb[mid] + (double)c_cnt[mid] * 0.5) hi = mid; else lo = mid + 1;
	}
	double res;
	if (lo < nc) {
		const double mean = (double)(int64_t)c_sum[lo] / (double)c_cnt[lo];
		const double c = (double)c_wb[lo] + (double)c_cnt[lo] * 0.5;
		if (lo == 0) {
			const double l = (double)vmin;
			res = c <= 0.0 ? mean : l + (mean - l) * (t / c);
		} else {
			const double pm = (double)(int64_t)c_sum[lo - 1] / (double)c_cnt[lo - 1];
			const double pc = (double)c_wb[lo - 1] + (double)c_cnt[lo - 1] * 0.5;
			res = pm + (mean - pm) * ((t - pc) / (c - pc));
		}
	} else {
		const double pm = (double)(int64_t)c_sum[nc - 1] / (double)c_cnt[nc - 1];
		const double pc = (double)c_wb[nc - 1] + (double)c_cnt[nc - 1] * 0.5;
		const double h = (double)vmax, span = (double)N - pc;
		res = span <= 0.0 ? h : pm + (h - pm) * ((t - pc) / span);
	}
	return floor(res + 0.5); // integer-millisecond value domain
}

// VPT: buffered + run values per thread the instance takes (merges of up to 256 VPT values: 4 for the default buffer of 896 values, 8 / 16 for
// gys_config.td_pend_cap up to 1920 / 3968).  The work of a merge is almost all per BIN and per CLUSTER (a value costs its load and one to
// three LDS atomics), so a buffer four times the size means a quarter of the merges at little more than the old cost each.
#define GYS_MB_BIG_CAP 1024u // large values (>= GYS_MB_EXACT) the list holds; a merge with more of them goes to the general kernel (slow_list)
#ifndef GYS_MB_WAVES16
#define GYS_MB_WAVES16 5 // waves per SIMD the 4096-value instance is compiled for (8: 64 VGPRs -- sixteen values per thread then spill to scratch)
#endif
#ifndef GYS_MB_KERNARG
#define GYS_MB_KERNARG 1 // the kernel's parameters are read from the kernel-argument segment where they are used (scalar loads, cached) instead of being held in ~90 SGPRs across the merge loop, which spilled 53 of them into VGPR lanes (170 v_readlane / v_writelane in the 2048-value instance)
#endif
template <bool SCAN, uint32_t VPT = 4u>
__global__ __launch_bounds__(256, (VPT == 16u ? GYS_MB_WAVES16 : 8)) void k_digest_bins(MergeBP q_arg)
{
#if GYS_MB_KERNARG && defined(__HIP_DEVICE_COMPILE__)
	typedef const MergeBP __attribute__((address_space(4))) *KernargP;
	KernargP q_k = (KernargP)__builtin_amdgcn_kernarg_segment_ptr(); // (the only parameter: offset 0)
#define GYS_MB_Q_RELOAD() asm volatile("" : "+s"(q_k))
	const MergeBP &q = *(const MergeBP *)q_k;
#else
#define GYS_MB_Q_RELOAD() (void)0
	const MergeBP &q = q_arg;
#endif
	const DigestP &p = q.d;
	static_assert(VPT == 4u || VPT == 8u || VPT == 16u || VPT == 64u, "merges of 1024 / 2048 / 4096 values; 64: up to 16 384, streamed");
	// VPT = 64 (round 6): the STREAMED instance for merges of up to 16 384 values (size class 2: the middle of a Zipf stream, which took the
	// several-workgroup path of gys_huge.hpp -- a 64-KiB bin array in HBM and a workgroup-wide dependent chain per key -- until then).  Pass 1
	// is the only user of the values themselves, so they are taken eight per thread at a time instead of all being held in registers; every
	// count of the passes behind it has room for 16 384 values (bin words: 16 bits; packed adds: 24-bit counts).  The large-value list holds
	// the bare value: ties among equal large values rank by list position (equal values are interchangeable).
	constexpr bool STREAM = VPT > 16u;
	constexpr uint32_t RV = STREAM ? 8u : VPT; // values a thread holds at a time
	__shared__ __align__(16) uint32_t s_bin[GYS_MB_BINS];
	// (Until round 6 the scan's list held any merge -- 256 VPT entries, 8 KB for buffers of 1 920 values -- which left 6 workgroups per CU where the
	// merge has 8; with the merge's list and its hand-over -- gys_scan_quantiles_dev sends the listed services through the general path one by one, as it
	// does for 64-bit weights -- the scan of 10^7 services takes 61 instead of 78 ms, r6aj.  A service is listed when more than 1 024 of its buffered
	// values are a second or longer.)
	constexpr uint32_t BIG_CAP = GYS_MB_BIG_CAP;
	__shared__ uint32_t s_big[BIG_CAP]; // values >= GYS_MB_EXACT: index << 20 | value
	__shared__ uint32_t s_thr[GYS_NBP];          // compacted non-empty old clusters: ceil(sum / count), padded with ~0
	__shared__ uint32_t s_cpfx[GYS_NBP + 1];     // old weight before compacted cluster c
	__shared__ uint32_t s_T[GYS_NBP];            // T_j, j = 1..NB-1; [0] = 0, [NB..] = ~0
	__shared__ unsigned long long s_osum[GYS_TD_NB];
	__shared__ uint32_t s_ocnt[GYS_TD_NB];
#if GYS_MB_PACKED
	__shared__ unsigned long long s_oval[GYS_TD_NB]; // the VALUES that land in an output cluster: {count : 24 | sum : 40}, one LDS atomic per value
#endif
	__shared__ unsigned long long s_fa[16], s_fw[16];
	__shared__ uint32_t s_fbm[GYS_BM_WORDS];
	__shared__ int32_t s_fmm[3];
	__shared__ uint32_t s_wv[4], s_ws[4], s_nbig;
	__shared__ uint64_t s_ww[4];
	const uint32_t nent = SCAN ? p.nsvc : *q.count;
	const bool query = SCAN || q.out_sum != nullptr;
	// RESP_TIME_HASH bucket of the thread's first bin and the first of its 8 bins that lies in the next bucket (8 = none): the
	// thresholds are at least 9 apart, so 8 consecutive values touch at most two buckets
	uint32_t bk_first = 0, bk_chg = GYS_MB_BPT;
	if (GYS_MB_BPT * threadIdx.x < GYS_MB_EXACT) {
		bk_first = resp_bucket((int64_t)(GYS_MB_BPT * threadIdx.x));
		for (uint32_t k = GYS_MB_BPT - 1u; k >= 1u; --k)
			if (resp_bucket((int64_t)(GYS_MB_BPT * threadIdx.x + k)) != bk_first) bk_chg = k;
	}

	// (Reading the NEXT list entry one iteration ahead and touching one word of every line the next merge will read -- clusters, buffered
	// words, records, meta: a key's digest is cold, three dependent HBM round trips per merge -- was measured in round 3 (r3p): 5.22
	// against 4.68 ms per window.  A plain software pipeline -- the next entry's meta / cluster / four values requested behind pass 1 of
	// the current merge, no extra requests -- was slower as well (r3ab: 5.12 against 4.82 ms: 64 VGPRs with two spills).  The merges are
	// not bound by the latency of their loads; both removed again.  Third form (r3af): meta record + cluster + values requested together and
	// waited for once (hipcc waits for the meta record before it issues the data loads: three dependent round trips become two): 4.81
	// against 4.84 ms -- no difference; with the next list entry held in registers across the merge as well: 4.92.  Not kept either.)
	// (Round 4, r4b: a start delay of 0..7 x 2 us by a hash of the workgroup number -- in case a CU's eight merges run in step and use the
	// memory path, the LDS and the VALUs one after the other -- changed nothing: 4.84 against 4.82 ms.  Removed.)
	for (uint32_t w = blockIdx.x; w < nent; w += gridDim.x) {
		// the thread index is re-derived per entry behind an opaque move: otherwise every LDS address, lane mask and per-bin
		// bucket the thread uses is hoisted out of this loop and held in registers across it (> 96 VGPRs instead of < 64)
		uint32_t tid = threadIdx.x;
		GYS_OPAQUE_VGPR(tid);
		GYS_MB_Q_RELOAD();
		const uint32_t lane = tid & 63u, wave = tid >> 6;
		MergeEnt ent;
		if (SCAN) ent = MergeEnt{w, min(p.td_meta[w].npend, p.pend_cap), 0u, 0u}; // between batches a buffer holds at most pend_cap values
		else ent = q.list[w];
		const uint32_t m = ent.nbuf + ent.mrun;
		if (m > 256u * VPT) continue; // never queued on this list (finalize_key)
		const uint32_t slot = ent.slot;
		const uint4 mt = *(const uint4 *)&p.td_meta[slot];
		const uint32_t nh = query ? m : (mt.y & 0xFFFFu), nw = mt.y >> 16;
		const uint32_t nh_mm = SCAN ? (mt.y & 0xFFFFu) : nh; // SCAN: min / max of the not yet folded words are not in td_minmax yet
		const uint32_t nwin0 = max(nh, nw);
		const uint32_t run0 = ent.off_end - ent.mrun;
		// ---- loads: cluster tid, values tid + 256 k
		uint32_t c0 = 0;
		int64_t sm0 = 0;
		if (tid < GYS_TD_NB) {
			c0 = p.td_cnt[(size_t)slot * GYS_TD_NB + tid];
			sm0 = p.td_sum[(size_t)slot * GYS_TD_NB + tid];
		}
		uint32_t wd[RV];
		const uint32_t *const pend = p.td_pend + (size_t)slot * p.pcap;
		{
#pragma unroll
			for (uint32_t k = 0; k < RV; ++k) {
				const uint32_t i = tid + 256u * k;
				wd[k] = 0;
				if (i < m) wd[k] = i < ent.nbuf ? pend[i] : p.staged[run0 + (i - ent.nbuf)];
			}
		}
		if (GYS_MB_SKIP & 32) { // (the loads stay live: an impossible value writes them out)
			uint32_t x = c0 ^ (uint32_t)sm0;
#pragma unroll
			for (uint32_t k = 0; k < RV; ++k) x ^= wd[k];
			if (x == 0xDEADBEEFu) p.td_cur[slot] = 1;
			continue;
		}
		if (m == 0 && !SCAN) { // nothing buffered (query of a freshly merged key): the merged view is the digest itself
			if (query && tid < GYS_TD_NB) {
				q.out_sum[(size_t)w * GYS_TD_NB + tid] = sm0;
				q.out_cnt[(size_t)w * GYS_TD_NB + tid] = c0;
			}
			continue;
		}
#pragma unroll
		for (uint32_t k = 0; k < GYS_MB_BPT; ++k) s_bin[tid + 256u * k] = 0;
		s_thr[tid] = 0xFFFFFFFFu;
		if (tid < GYS_TD_NB) {
			s_osum[tid] = 0;
			s_ocnt[tid] = 0;
#if GYS_MB_PACKED
			s_oval[tid] = 0;
#endif
		}
		if (tid < 16u) {
			s_fa[tid] = 0;
			s_fw[tid] = 0;
			s_fbm[tid] = 0;
			s_fbm[tid + 16u] = 0;
		}
		if (tid == 0) {
			s_fmm[0] = INT32_MAX;
			s_fmm[1] = INT32_MIN;
			s_fmm[2] = INT32_MIN;
			s_nbig = 0;
		}
		// compaction of the non-empty clusters (order preserving) + exclusive prefix of their weights
		const unsigned long long b0 = __ballot(c0 != 0);
		const uint64_t inc = wave_incl_scan_u64(c0);
		if (lane == 63u) s_ww[wave] = inc;
		if (lane == 0u) s_wv[wave] = (uint32_t)__popcll(b0);
		__syncthreads();
		uint32_t pbase = 0, nc = 0;
		uint64_t wbase = 0, nold64 = 0;
#pragma unroll
		for (uint32_t k = 0; k < 4u; ++k) {
			if (k < wave) {
				pbase += s_wv[k];
				wbase += s_ww[k];
			}
			nc += s_wv[k];
			nold64 += s_ww[k];
		}
		if (nold64 + (uint64_t)m >= (1ull << 31)) { // 64-bit weights: the general kernel's job
			if (tid == 0) q.slow_list[atomicAdd(q.slow_count, 1u)] = ent;
			__syncthreads();
			continue;
		}
		if (SCAN && m == 0) { // the merged view is the digest itself: straight to the quantiles
			if (tid < GYS_TD_NB) {
				s_osum[tid] = (unsigned long long)sm0;
				s_ocnt[tid] = c0;
			}
			__syncthreads();
		} else {
		const uint32_t pos0 = pbase + (uint32_t)__popcll(b0 & (lane ? (~0ull >> (64 - lane)) : 0ull));
		const uint32_t e0 = (uint32_t)(wbase + inc - c0), nold = (uint32_t)nold64;
		const uint32_t twoN = 2u * (nold + m);
		uint32_t thr = 0;
		if (c0) {
			thr = ceil_div_sum_cnt(sm0, c0);
			s_thr[pos0] = thr;
			s_cpfx[pos0] = e0;
			atomicAdd(&s_bin[mb_bin(thr)], 1u << 16);
		}
		if (tid == 0) s_cpfx[nc] = nold;
		s_T[tid] = (tid >= 1u && tid < GYS_TD_NB) ? (uint32_t)td_threshold(c_td_bnd[tid], (uint64_t)twoN) : (tid ? 0xFFFFFFFFu : 0u);
		// ---- values, pass 1: bin count (-> arrival order inside the bin), list of large values, window part of the fold
		const bool fold_scan = !query && nh == 0u; // the all-time deltas of the one-value bins come from the scan
		int32_t lmin = INT32_MAX, lmax = INT32_MIN, wmax = INT32_MIN;
		for (uint32_t cb = 0; cb < (STREAM ? m : 1u); cb += 256u * RV) { // (one round unless STREAM)
		if (STREAM && cb) { // the next eight values of the thread (the first eight were requested with the clusters)
#pragma unroll
			for (uint32_t k = 0; k < RV; ++k) {
				const uint32_t i = cb + tid + 256u * k;
				wd[k] = 0;
				if (i < m) wd[k] = i < ent.nbuf ? pend[i] : p.staged[run0 + (i - ent.nbuf)];
			}
		}
#pragma unroll
		for (uint32_t k = 0; k < RV; ++k) {
			const uint32_t i = cb + tid + 256u * k;
			// (r6ay: adding a wave's 64 values up per (bucket, record) key first -- ballot, count, DPP sum, one lane's add -- instead of one LDS add per
			// value and record made the streamed instance SLOWER, 0.48 -> 0.92 ms on C5: the instance is bound by its instructions, not by same-address adds)
			if (!(i >= m || (GYS_MB_SKIP & 4))) {
			const uint32_t uv = wd[k] >> GYS_ROW_BITS;
			atomicAdd(&s_bin[mb_bin(uv)], 1u); // (no rank inside the bin is needed: equal values are interchangeable, pass 2 works per bin)
			const bool big = uv >= GYS_MB_EXACT;
			if (big) {
				const uint32_t at = atomicAdd(&s_nbig, 1u);
				if (256u * VPT <= BIG_CAP || at < BIG_CAP) s_big[at] = STREAM ? uv : ((i << 20) | uv);
			}
			if (SCAN && i >= nh_mm) {
				lmin = min(lmin, (int32_t)uv);
				lmax = max(lmax, (int32_t)uv);
			}
			if (i >= nh) { // not yet folded: histogram bucket of the key's records, CONN_BITMAP row, min / max
				lmin = min(lmin, (int32_t)uv);
				lmax = max(lmax, (int32_t)uv);
				const bool win = i >= nwin0;
				if (big || !fold_scan || win) {
					const uint32_t b = resp_bucket((int64_t)uv);
					const unsigned long long one = GYS_PACK_ONE | (unsigned long long)uv;
					if (big || !fold_scan) atomicAdd(&s_fa[b], one);
					if (win) {
						atomicAdd(&s_fw[b], one);
						const uint32_t r = wd[k] & GYS_ROW_MASK;
						atomicOr(&s_fbm[r >> 1], (1u << b) << ((r & 1u) * 16u));
						wmax = max(wmax, (int32_t)uv);
					}
				}
			}
			}
		}
		} // (cb)
		if ((!query && m > nh) || (SCAN && m > nh_mm)) {
			lmin = wave_min_i32(lmin);
			lmax = wave_max_i32(lmax);
			wmax = wave_max_i32(wmax);
			if (lane == 0) {
				if (lmin != INT32_MAX) atomicMin(&s_fmm[0], lmin);
				if (lmax != INT32_MIN) atomicMax(&s_fmm[1], lmax);
				if (wmax != INT32_MIN) atomicMax(&s_fmm[2], wmax);
			}
		}
		__syncthreads();
		if (256u * VPT > BIG_CAP && s_nbig > BIG_CAP) { // more large values than the list holds (a key whose responses take seconds): the general kernel's job
			if (tid == 0) q.slow_list[atomicAdd(q.slow_count, 1u)] = ent;
			__syncthreads();
			continue;
		}
		// ---- one scan over the bins: thread t owns bins [8t, 8t + 8)
		if (!(GYS_MB_SKIP & 8)) {
			uint32_t bv[GYS_MB_BPT], own = 0;
			{
				const uint4 lo4 = ((const uint4 *)s_bin)[2u * tid], hi4 = ((const uint4 *)s_bin)[2u * tid + 1u]; // two 16-byte reads per lane
				bv[0] = lo4.x; bv[1] = lo4.y; bv[2] = lo4.z; bv[3] = lo4.w;
				bv[4] = hi4.x; bv[5] = hi4.y; bv[6] = hi4.z; bv[7] = hi4.w;
			}
#pragma unroll
			for (uint32_t k = 0; k < GYS_MB_BPT; ++k) own += bv[k];
			const uint32_t sc = wave_incl_scan_u32(own);
			if (lane == 63u) s_ws[wave] = sc;
			__syncthreads();
			uint32_t run = sc - own;
#pragma unroll
			for (uint32_t k = 0; k < 3u; ++k)
				if (k < wave) run += s_ws[k];
			unsigned long long acc0 = 0, acc1 = 0; // GY_HISTOGRAM::add_data for cnt values equal to `bin`, per bucket of the thread's bins
#pragma unroll
			for (uint32_t k = 0; k < GYS_MB_BPT; ++k) {
				const uint32_t bin = GYS_MB_BPT * tid + k;
				// {values in lower bins : 16 | clusters with threshold in this or a lower bin : 16}
				const uint32_t raw = bv[k];
				bv[k] = (run & 0xFFFFu) | (((run >> 16) + (raw >> 16)) << 16);
				run += raw;
				const uint32_t cnt = raw & 0xFFFFu; // cnt x {1 : 24 | bin : 40}: cnt <= 1024 and bin < 1024, the halves cannot meet
				const unsigned long long d = ((unsigned long long)(cnt << 8) << 32) | (unsigned long long)(cnt * bin);
				if (k < bk_chg) acc0 += d; else acc1 += d;
			}
			((uint4 *)s_bin)[2u * tid] = make_uint4(bv[0], bv[1], bv[2], bv[3]);
			((uint4 *)s_bin)[2u * tid + 1u] = make_uint4(bv[4], bv[5], bv[6], bv[7]);
			if (fold_scan && GYS_MB_BPT * tid < GYS_MB_EXACT) { // (a thread's 8 bins are all one-value bins or all cells: 1024 = 8 x 128)
				if (acc0) atomicAdd(&s_fa[bk_first], acc0);
				if (acc1) atomicAdd(&s_fa[bk_first + 1u], acc1);
			}
		}
		__syncthreads();
		const uint32_t nbig = s_nbig;
		// ---- old clusters: preceded by the old weight before them and by the values below their mean
#ifndef GYS_MB_FUSE_OLD
#define GYS_MB_FUSE_OLD 0 // 1: the old cluster's threshold search runs in lockstep with the searches of the thread's first group of bins
#endif
		uint32_t mid2_old = 0;
		const bool has_old = c0 && !(GYS_MB_SKIP & 2);
		if (has_old) {
			uint32_t nb = s_bin[mb_bin(thr)] & 0xFFFFu;
			if (thr >= GYS_MB_EXACT) {
				const uint32_t sh = (31u - (uint32_t)__clz((int)thr)) - 6u;
				for (uint32_t j = 0; j < nbig; ++j) {
					const uint32_t u = s_big[j] & 0xFFFFFu;
					nb += ((u >> sh) == (thr >> sh) && u < thr) ? 1u : 0u;
				}
			}
			const uint32_t mid2 = 2u * (e0 + nb) + c0;
#if GYS_MB_FUSE_OLD
			mid2_old = mid2; // (searched below, in lockstep with the thread's first group of bins)
#else
			uint32_t a = 0;
#pragma unroll
			for (uint32_t step = GYS_NBP / 2; step >= 1u; step >>= 1)
				if (mid2 >= s_T[a + step]) a += step;
			atomicAdd(&s_osum[a], (unsigned long long)sm0);
			atomicAdd(&s_ocnt[a], c0);
#endif
		}
		// ---- values, pass 2, PER BIN: the c values of a one-value bin are equal, so they take the consecutive mid-points
		// first, first + 2, ... and only the cluster boundaries that fall between them matter: one threshold search per NON-EMPTY BIN
		// (integer-millisecond response times repeat: ~860 buffered values hold ~200 distinct ones) instead of one per value, and one
		// packed add per (bin, output cluster).  Thread t takes bins t, t + 256, t + 512, t + 768: the busy low bins spread over all waves.
#ifndef GYS_MB_GROUP
#define GYS_MB_GROUP 1u // the thread's four bins searched GROUP at a time (1, 2 or 4): with 2 / 4 the 8 dependent LDS reads of one threshold search overlap the others' -- measured in round 4 (profiles/r4a_ab_paired_search_and_tests.txt): 4.71 / 4.75 / 4.85 ms for 2 / 1 / 4 at full size, 1.265 / 1.255 / 1.281 at quarter size: no gain, the per-bin pass is not bound by that chain; 1 = the plain form stays the default
#endif
		constexpr uint32_t MBG = SCAN ? 1u : GYS_MB_GROUP; // (the scan form sits at 63 VGPRs: left as it was)
#ifndef GYS_MB_COMPACT
#define GYS_MB_COMPACT 1 // the per-bin pass walks a compacted list of the NON-EMPTY one-value bins
#endif
		// Round 5: of the 1 024 one-value bins a merge touches ~200 - 400 (integer-millisecond response times repeat), spread so that every
		// wave of every one of the four rounds below has some non-empty bin among its 64 -- each round then runs the 8-step threshold search
		// for the whole wave.  The non-empty bins are compacted first (four ballots per thread, wave totals through LDS, a 16-bit list in
		// the unused upper half of the large-value list): ~300 bins are 5 wave-rounds instead of 16.  The kernel is bound by VALU issue
		// (DESIGN 10): the instructions saved are time saved.  (A merge with more than half of the list's room in large values keeps the plain form.)
		const bool compact = GYS_MB_COMPACT && MBG == 1u && !GYS_MB_FUSE_OLD && nbig <= BIG_CAP / 2u;
		if (compact) {
			uint16_t *const s_ne = (uint16_t *)(s_big + BIG_CAP / 2u); // [<= 1024] bin numbers
			const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
			uint32_t tw = 0, pos[4];
			bool ne[4];
#pragma unroll
			for (uint32_t k = 0; k < 4u; ++k) {
				const uint32_t b = tid + 256u * k;
				ne[k] = !(GYS_MB_SKIP & 1) && ((s_bin[b + 1u] ^ s_bin[b]) & 0xFFFFu) != 0u; // (the low halves are running counts: different = the bin holds values)
				const unsigned long long bal = __ballot(ne[k]);
				pos[k] = tw + (uint32_t)__popcll(bal & below);
				tw += (uint32_t)__popcll(bal);
			}
			if (lane == 0u) s_ws[wave] = tw; // (the scan's wave sums were consumed before the barrier above)
			__syncthreads();
			uint32_t base = 0, n_ne = 0;
#pragma unroll
			for (uint32_t k = 0; k < 4u; ++k) {
				if (k < wave) base += s_ws[k];
				n_ne += s_ws[k];
			}
#pragma unroll
			for (uint32_t k = 0; k < 4u; ++k)
				if (ne[k]) s_ne[base + pos[k]] = (uint16_t)(tid + 256u * k);
			__syncthreads();
			for (uint32_t i = tid; i < n_ne; i += 256u) {
				const uint32_t b = s_ne[i];
				const uint32_t bw = s_bin[b];
				uint32_t rem = (s_bin[b + 1u] & 0xFFFFu) - (bw & 0xFFFFu), mid2 = 2u * ((bw & 0xFFFFu) + s_cpfx[bw >> 16]) + 1u, a = 0;
#pragma unroll
				for (uint32_t step = GYS_NBP / 2; step >= 1u; step >>= 1)
					if (mid2 >= s_T[a + step]) a += step;
				while (rem) { // (nearly always one round: a cluster spans far more mid-points than a bin's values)
					const uint32_t Tn = s_T[a + 1u]; // first mid-point of the next cluster (~0 after the last)
					const uint32_t kk = min(rem, (Tn - mid2 + 1u) >> 1); // values with mid2 + 2 r < Tn
#if GYS_MB_PACKED
					atomicAdd(&s_oval[a], ((unsigned long long)kk << 40) | (unsigned long long)(kk * b));
#else
					atomicAdd(&s_osum[a], (unsigned long long)(kk * b));
					atomicAdd(&s_ocnt[a], kk);
#endif
					rem -= kk;
					mid2 += 2u * kk;
					++a;
				}
			}
		} else
#pragma unroll
		for (uint32_t k0 = 0; k0 < GYS_MB_EXACT / 256u; k0 += MBG) {
			uint32_t bq[MBG], cq[MBG], mq[MBG], aq[MBG], call = 0;
#pragma unroll
			for (uint32_t u = 0; u < MBG; ++u) {
				bq[u] = tid + 256u * (k0 + u);
				const uint32_t bw = s_bin[bq[u]];
				cq[u] = (GYS_MB_SKIP & 1) ? 0u : (s_bin[bq[u] + 1u] & 0xFFFFu) - (bw & 0xFFFFu);
				mq[u] = 2u * ((bw & 0xFFFFu) + s_cpfx[bw >> 16]) + 1u;
				aq[u] = 0;
				call |= cq[u];
			}
			const bool fuse = GYS_MB_FUSE_OLD && !SCAN && k0 == 0u;
			if (!call && !(fuse && has_old)) continue;
			uint32_t a_old = 0;
#pragma unroll
			for (uint32_t step = GYS_NBP / 2; step >= 1u; step >>= 1) {
				uint32_t tq[MBG], t_old = 0;
#pragma unroll
				for (uint32_t u = 0; u < MBG; ++u) tq[u] = s_T[aq[u] + step];
				if (fuse) t_old = s_T[a_old + step];
#pragma unroll
				for (uint32_t u = 0; u < MBG; ++u)
					if (mq[u] >= tq[u]) aq[u] += step;
				if (fuse && mid2_old >= t_old) a_old += step;
			}
#pragma unroll
			for (uint32_t u = 0; u < MBG; ++u) GYS_OPAQUE_VGPR(aq[u]); // (the searches stay in this block: sunk into the `while (rem)` bodies below they would run one after the other)
			if (fuse) {
				GYS_OPAQUE_VGPR(a_old);
				if (has_old) {
					atomicAdd(&s_osum[a_old], (unsigned long long)sm0);
					atomicAdd(&s_ocnt[a_old], c0);
				}
			}
#pragma unroll
			for (uint32_t u = 0; u < MBG; ++u) {
				uint32_t rem = cq[u], mid2 = mq[u], a = aq[u];
				const uint32_t b = bq[u];
				while (rem) { // (nearly always one round: a cluster spans far more mid-points than a bin's values)
					const uint32_t Tn = s_T[a + 1u]; // first mid-point of the next cluster (~0 after the last)
					const uint32_t kk = min(rem, (Tn - mid2 + 1u) >> 1); // values with mid2 + 2 r < Tn
#if GYS_MB_PACKED
					atomicAdd(&s_oval[a], ((unsigned long long)kk << 40) | (unsigned long long)(kk * b));
#else
					atomicAdd(&s_osum[a], (unsigned long long)(kk * b));
					atomicAdd(&s_ocnt[a], kk);
#endif
					rem -= kk;
					mid2 += 2u * kk;
					++a;
				}
			}
		}
		// the (few) large values, one per thread from the list: rank inside the cell by comparison with the other large values, gap by
		// search over the cluster thresholds
		for (uint32_t j = tid; j < ((GYS_MB_SKIP & 2) ? 0u : nbig); j += 256u) {
			const uint32_t me = s_big[j], uv = me & 0xFFFFFu, i = me >> 20;
			const uint32_t sh = (31u - (uint32_t)__clz((int)uv)) - 6u;
			uint32_t r = s_bin[mb_bin(uv)] & 0xFFFFu;
			for (uint32_t jj = 0; jj < nbig; ++jj) {
				const uint32_t e = s_big[jj], u = e & 0xFFFFFu;
				r += ((u >> sh) == (uv >> sh) && (u < uv || (u == uv && (STREAM ? jj < j : (e >> 20) < i)))) ? 1u : 0u;
			}
			uint32_t gap = 0;
#pragma unroll
			for (uint32_t step = GYS_NBP / 2; step >= 1u; step >>= 1)
				if (s_thr[gap + step - 1u] <= uv) gap += step;
			const uint32_t mid2 = 2u * (r + s_cpfx[gap]) + 1u;
			uint32_t a = 0;
#pragma unroll
			for (uint32_t step = GYS_NBP / 2; step >= 1u; step >>= 1)
				if (mid2 >= s_T[a + step]) a += step;
#if GYS_MB_PACKED
			atomicAdd(&s_oval[a], GYS_PACK_ONE | (unsigned long long)uv);
#else
			atomicAdd(&s_osum[a], (unsigned long long)uv);
			atomicAdd(&s_ocnt[a], 1u);
#endif
		}
		__syncthreads();
#if GYS_MB_PACKED
		if (tid < GYS_TD_NB) { // (own entry only; the consumers below read after their own barrier or read their own entry)
			const unsigned long long pv = s_oval[tid];
			s_osum[tid] += GYS_PACK_SUM(pv);
			s_ocnt[tid] += (uint32_t)(pv >> 40);
		}
		if (SCAN) __syncthreads();
#endif
		} // (SCAN && m == 0)
		if (SCAN) {
			// ---- quantiles of the merged view: compact the non-empty clusters (order preserving), weight before each, one thread per quantile
			const uint32_t oc = tid < GYS_TD_NB ? s_ocnt[tid] : 0u;
			const unsigned long long os = tid < GYS_TD_NB ? s_osum[tid] : 0ull;
			const unsigned long long ob = __ballot(oc != 0);
			const uint32_t sc = wave_incl_scan_u32(oc);
			if (lane == 63u) s_ws[wave] = sc;
			if (lane == 0u) s_wv[wave] = (uint32_t)__popcll(ob);
			__syncthreads();
			uint32_t pb = 0, wb = 0, ncq = 0, Nq = 0;
#pragma unroll
			for (uint32_t k = 0; k < 4u; ++k) {
				if (k < wave) {
					pb += s_wv[k];
					wb += s_ws[k];
				}
				ncq += s_wv[k];
				Nq += s_ws[k];
			}
			if (oc) {
				const uint32_t pos = pb + (uint32_t)__popcll(ob & (lane ? (~0ull >> (64 - lane)) : 0ull));
				s_thr[pos] = oc;                  // compacted counts
				s_cpfx[pos] = wb + sc - oc;       // weight before
				s_osum[pos] = os;                 // (pos <= tid: every thread read its own entry before the barrier above)
			}
			__syncthreads();
			if (tid < q.nq) {
				const int2 mm = p.td_minmax[slot];
				q.qout[(size_t)slot * q.nq + tid] = td_quantile_dev(s_thr, s_osum, s_cpfx, ncq, Nq, min(mm.x, s_fmm[0]), max(mm.y, s_fmm[1]), q.qs[tid]);
			}
			__syncthreads();
			continue;
		}
		// ---- write back
		if (GYS_MB_SKIP & 16) {
			if (tid == 208u && !query) { // (the buffer still has to drain, or the next windows would queue the key forever)
				*(uint4 *)&p.td_meta[slot] = make_uint4(0u, 0u, mt.z, mt.w);
				p.td_cur[slot] = 0;
			}
			__syncthreads();
			continue;
		}
		if (tid < GYS_TD_NB) {
			int64_t *ws = query ? q.out_sum + (size_t)w * GYS_TD_NB : p.td_sum + (size_t)slot * GYS_TD_NB;
			uint32_t *wc = query ? q.out_cnt + (size_t)w * GYS_TD_NB : p.td_cnt + (size_t)slot * GYS_TD_NB;
			ws[tid] = (int64_t)s_osum[tid];
			wc[tid] = s_ocnt[tid];
		}
		if (!query) {
			const uint32_t n_all = m - nh, n_win = m > nwin0 ? m - nwin0 : 0u;
			if (tid >= 192u && tid < 208u && n_all)
				fold_records(p, slot, tid - 192u, mt.w != mt.z, s_fa[tid - 192u], s_fw[tid - 192u], n_all, n_win, s_fmm[1], s_fmm[2], s_fbm[tid - 192u], s_fbm[tid - 176u]);
			if (tid == 208u) {
				*(uint4 *)&p.td_meta[slot] = make_uint4(0u, 0u, mt.z, n_win ? mt.z : mt.w); // buffer drained
				p.td_cur[slot] = 0;
				if (n_all) {
					const int2 mm = p.td_minmax[slot];
					if (s_fmm[0] < mm.x || s_fmm[1] > mm.y) p.td_minmax[slot] = make_int2(min(mm.x, s_fmm[0]), max(mm.y, s_fmm[1]));
				}
			}
		}
		__syncthreads();
	}
}

// ---------------------------------------------------------------------------------------------------- t-digest merge (huge)
// One 256-thread workgroup per huge key (persistent over the work list).  Each workgroup owns a 2^20-bin u32 count array in HBM
// scratch: values are histogrammed exactly, the bins are prefix-scanned, and every bin's rank interval is intersected with the
// cluster rank intervals -- the same exact-integer assignment as k_digest_merge without materialising a sort.
struct HugeP {
	DigestP d;
	const MergeEnt *huge_list;
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
	__shared__ unsigned long long s_ha[32], s_hw[32]; // exact {count, sum} per bucket of all values / of the window part
	__shared__ uint32_t s_bm[GYS_BM_WORDS];
	__shared__ uint32_t s_nc;
	__shared__ int32_t s_min, s_max, s_wmax;
	uint32_t *bins = p.scratch + (size_t)blockIdx.x * GYS_HUGE_BINS;
	const uint32_t nh_list = *p.huge_count;
	const uint32_t BPT = GYS_HUGE_BINS / 256u; // bins per thread (contiguous)

	for (uint32_t w = blockIdx.x; w < nh_list; w += gridDim.x) {
		const MergeEnt ent = p.huge_list[w];
		const uint32_t slot = ent.slot;
		const uint32_t m = ent.mrun, npend = ent.nbuf;
		const uint32_t start = ent.off_end - m;
		const uint4 mt = *(const uint4 *)&p.d.td_meta[slot];
		const uint32_t nh = mt.y & 0xFFFFu, nw = mt.y >> 16;
		const uint32_t nwin0 = max(nh, nw);
		const uint32_t *pend = p.d.td_pend + (size_t)slot * p.d.pcap;
		// zero the bins (16-byte stores)
		for (uint32_t i = threadIdx.x; i < GYS_HUGE_BINS / 4u; i += 256u) ((uint4 *)bins)[i] = make_uint4(0, 0, 0, 0);
		if (threadIdx.x < GYS_TD_NB) {
			s_osum[threadIdx.x] = 0;
			s_ocnt[threadIdx.x] = 0;
		}
		if (threadIdx.x < 32u) {
			s_ha[threadIdx.x] = 0;
			s_hw[threadIdx.x] = 0;
		}
		if (threadIdx.x >= 160u && threadIdx.x < 160u + GYS_BM_WORDS) s_bm[threadIdx.x - 160u] = 0;
		if (threadIdx.x == 0) {
			// compact non-empty old clusters (serial: <= 200 entries, once per huge key)
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
			s_wmax = INT32_MIN;
		}
		__syncthreads();
		const uint32_t nc = s_nc;
		const uint64_t nold = s_cpfx[nc];
		const uint64_t twoN = 2ull * (nold + (uint64_t)m + (uint64_t)npend);
		if (threadIdx.x >= 1 && threadIdx.x < GYS_TD_NB) s_T[threadIdx.x] = td_threshold(c_td_bnd[threadIdx.x], twoN);
		if (threadIdx.x == 0) s_T[GYS_TD_NB] = ~0ull;
		// exact value histogram of (buffered + run) values; the records' deltas come from the not yet folded ones
		{
			int32_t lmin = INT32_MAX, lmax = INT32_MIN, wmax = INT32_MIN;
			for (uint32_t i = threadIdx.x; i < npend + m; i += 256u) {
				const uint32_t word = i < npend ? pend[i] : p.d.staged[start + (i - npend)];
				const uint32_t v = (word >> GYS_ROW_BITS) & (GYS_HUGE_BINS - 1u);
				atomicAdd(&bins[v], 1u);
				if (i >= nh) {
					const uint32_t hb = resp_bucket((int64_t)v);
					atomicAdd(&s_ha[2 * hb], 1ull);
					atomicAdd(&s_ha[2 * hb + 1], (unsigned long long)v);
					lmin = min(lmin, (int32_t)v);
					lmax = max(lmax, (int32_t)v);
					if (i >= nwin0) {
						atomicAdd(&s_hw[2 * hb], 1ull);
						atomicAdd(&s_hw[2 * hb + 1], (unsigned long long)v);
						const uint32_t row = word & GYS_ROW_MASK;
						const uint32_t bit = (1u << hb) << ((row & 1u) * 16u);
						if ((s_bm[row >> 1] & bit) == 0) atomicOr(&s_bm[row >> 1], bit);
						wmax = max(wmax, (int32_t)v);
					}
				}
			}
			atomicMin(&s_min, lmin);
			atomicMax(&s_max, lmax);
			atomicMax(&s_wmax, wmax);
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
		if (threadIdx.x < GYS_TD_NB) {
			p.d.td_sum[(size_t)slot * GYS_TD_NB + threadIdx.x] = (int64_t)s_osum[threadIdx.x];
			p.d.td_cnt[(size_t)slot * GYS_TD_NB + threadIdx.x] = (uint32_t)s_ocnt[threadIdx.x];
		}
		{
			// the key's record deltas (the key is owned by this workgroup for the batch: plain read-modify-writes, see fold_records)
			const uint32_t n_all = npend + m - nh, n_win = npend + m - nwin0;
			const uint32_t t = threadIdx.x - 128u;
			if (threadIdx.x >= 128u && t < 16u) {
				// huge keys can exceed the packed accumulators' ranges: exact 64-bit pairs, same rules as fold_records
				gys_hist_serial *ap = (gys_hist_serial *)&p.d.hist_all[slot] + t, *wp = (gys_hist_serial *)&p.d.hist_win[slot] + t;
				const bool roll = mt.w != mt.z;
				gys_hist_serial av = *ap;
				if (t < 15u) {
					av.count += s_ha[2 * t];
					av.sum += (int64_t)s_ha[2 * t + 1];
				} else {
					av.count += n_all;
					if (av.sum < (int64_t)s_max) av.sum = (int64_t)s_max;
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
						wv.count += s_hw[2 * t];
						wv.sum += (int64_t)s_hw[2 * t + 1];
					} else {
						wv.count += n_win;
						if (wv.sum < (int64_t)s_wmax) wv.sum = (int64_t)s_wmax;
					}
					*wp = wv;
					uint32_t *bp = &p.d.bitmap[(size_t)slot * GYS_BM_WORDS + t];
					*bp = (roll ? 0u : *bp) | s_bm[t];
					bp[16] = (roll ? 0u : bp[16]) | s_bm[t + 16u];
				}
			}
			__syncthreads(); // every reader of the meta record is done before thread 0 rewrites it
			if (threadIdx.x == 0) {
				*(uint4 *)&p.d.td_meta[slot] = make_uint4(0u, 0u, mt.z, n_win ? mt.z : mt.w);
				p.d.td_cur[slot] = 0;
				const int2 mm = p.d.td_minmax[slot];
				p.d.td_minmax[slot] = make_int2(min(mm.x, s_min), max(mm.y, s_max));
			}
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
//   tusec_start_@128 tusec_close_@136 cli_task_aggr_id_@144 ... ser_glob_id_@192 ... bytes_sent_@208 bytes_rcvd_@216 ... cli_cmdline_len_@272
//   is_tcp_connect_event_@274 is_tcp_accept_event_@275 is_loopback_conn_@276 is_pre_existing_@277 notified_before_@278 padding_len_@279
//
// A CONNECTION IS COUNTED ONCE (round 4; server/gy_mconnhdlr.cc:9129-9181).  A partha reports a connection when it opens (tusec_close_ = 0:
// the walk's `nnew`) and again when it closes with notified_before_ set, or -- a short-lived one -- only at its close with notified_before_
// clear (`nclosed_no_not`); the connecting and the accepting partha each report their own half (is_tcp_connect_event_ /
// is_tcp_accept_event_; both flags on one record = a loopback connection, reported once, :A.3).  So:
//   * the walk's own tallies: conn_new (+1 per open record), conn_closed (+1 per record with tusec_close_), conn_closed_no_notify;
//   * a record is LISTENER SIDE when is_tcp_accept_event_ is set (a loopback record included) or when neither flag is (the walk hands a
//     record that is not a connect event to add_tcp_conn_ser, :9333); it is CLIENT SIDE when only is_tcp_connect_event_ is set;
//   * per-service counters / window accumulators / the service Count-Min rows take LISTENER-SIDE records only: nconn += !notified_before_
//     (every connection exactly once: at its open record, or at the only record a short-lived one has), nclose += (tusec_close_ != 0), bytes
//     as reported (zero on an open record); the client half of the same connection names the same ser_glob_id_ once the pairing has
//     resolved it and would count the connection a second time -- it only moves conn_client_side;
//   * the pair Count-Min (gys_config.conn_pair_cms) follows connlistenmap_ / connclientmap_ (:9226-9319): a CLOSED record with bytes adds
//     one connection + its bytes to the listener-side tables when ser_glob_id_ != 0 and is_tcp_accept_event_, to the client-side tables when
//     it is connect-only and cli_task_aggr_id_ != 0 (the reference's `connpeer` gate -- a brand-new connection the join table has not
//     resolved yet -- belongs to the flow join, which the sketches replace: DESIGN section 7);
//   * the distinct-flow HyperLogLog takes every record (it is idempotent: both halves and both notifications hash to the same flow key).
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
	uint32_t *pair32;            // gys_config.conn_pair_cms: Count-Min pair keyed by (ser_glob_id_, cli_task_aggr_id_), else nullptr: listener side ...
	unsigned long long *pair64;
	uint32_t *cpair32;           // ... and client side
	unsigned long long *cpair64;
	uint32_t span;               // records per workgroup (conn_span: a multiple of 64)
};

#ifndef GYS_CONN_THREADS
#define GYS_CONN_THREADS 1024u // sixteen waves: 7.5 KB of staged records per wave + a 1024-entry service table + the HLL candidates = 158 of 160 KB
#endif
#define GYS_CONN_RECS (2u * GYS_CONN_THREADS) // records per workgroup of a SMALL call: two rounds of GYS_CONN_THREADS
#ifndef GYS_CONN_SPAN
#define GYS_CONN_SPAN (8u * GYS_CONN_THREADS) // ... and about this many in a large one (conn_span)
#endif
#ifndef GYS_CONN_AGG_BITS
#define GYS_CONN_AGG_BITS 10u // (2048 entries at twelve waves: 1.02 against 0.98 ms, and 1.64 against 1.44 ms on the mixed stream, r4x)
#endif
#define GYS_CONN_AGG (1u << GYS_CONN_AGG_BITS) // LDS aggregation slots per workgroup
// Records per workgroup.  The services of a workgroup's records are added to the device accumulators once per workgroup, and those
// device-scope adds are what the kernel pays for beside its reads (r4j: half the records per workgroup, 1.19 -> 1.38 ms; r4k: four times,
// 1.156 -> 1.118 ms with a third of the last dispatch round idle).  So a large call gives a workgroup ~6144 records -- in as many workgroups
// as fill every CU the same number of times (one workgroup per CU is resident: the LDS) -- and a call too small for that keeps 1536.
static inline uint32_t conn_span(uint32_t n, uint32_t ncu)
{
	if (!ncu || (uint64_t)n <= (uint64_t)ncu * GYS_CONN_RECS) return GYS_CONN_RECS;
	const uint64_t per_pass = (uint64_t)ncu * GYS_CONN_SPAN;
	const uint64_t grid = (((uint64_t)n + per_pass - 1) / per_pass) * ncu;
	const uint64_t span = ((uint64_t)n + grid - 1) / grid;
	return (uint32_t)((span + 63u) & ~63ull);
}
#ifndef GYS_CONN_SKIP
#define GYS_CONN_SKIP 0 // TIMING EXPERIMENTS ONLY (results are wrong): 1 no flow hash / HLL, 2 nothing after the HLL, 4 no LDS aggregation, 8 loads only, 16 hash but no register access
#endif
// the words of a record the roll-up needs (the fourteen 8-byte units of the record staged in LDS: GYS_CONN_UNIT_OFF)
struct ConnRec {
	uint64_t a0, a1, a2, a3; // nat_cli_ @64: ip128 (16 B), ip32 @16, port @24
	uint64_t b0, b1, b2, b3; // nat_ser_ @96
	uint64_t tusec_close;    // @136
	uint64_t task;           // @144 cli_task_aggr_id_
	uint64_t ser_glob_id;    // @192
	uint64_t bytes_sent, bytes_rcvd; // @208, @216
	uint64_t flags;          // @272: cli_cmdline_len_ (2 bytes), the five bool bytes @274..278, padding_len_
};
// the flag bytes inside ConnRec::flags (byte k of the word = record byte 272 + k)
#define GYS_CONN_F_CONNECT(w) (((w) >> 16) & 0xFFull)
#define GYS_CONN_F_ACCEPT(w) (((w) >> 24) & 0xFFull)
#define GYS_CONN_F_NOTIFIED(w) (((w) >> 48) & 0xFFull)

// per-workgroup tallies of the walk (LDS), flushed with one device atomic each by k_conn_ingest
enum { CONN_T_NEW = 0, CONN_T_CLOSED, CONN_T_CLOSED_NO_NOTIFY, CONN_T_CLI_SIDE, CONN_T_UNKNOWN, CONN_T_NUM };

__device__ __forceinline__ void conn_tally(uint32_t *s_tally, int which, bool pred)
{
	const unsigned long long b = __ballot(pred);
	if (b && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)b) - 1)) atomicAdd(&s_tally[which], (uint32_t)__popcll(b));
}

// a listener-side record whose service found no LDS entry (table full) or whose glob_id equals the table's empty mark: the service is looked
// up and the device accumulators are added to directly (what every record did before the per-workgroup table)
__device__ __forceinline__ void conn_direct(const ConnP &p, uint64_t ser_glob_id, bool fresh, bool closed, uint64_t bytes_sent, uint64_t bytes_rcvd, uint32_t *s_tally)
{
	const uint32_t slot = tbl_lookup(p.gid, ser_glob_id);
	if (slot == GYS_NOSLOT) {
		atomicAdd(&s_tally[CONN_T_UNKNOWN], 1u);
#pragma unroll
		for (uint32_t r = 0; r < GYS_CMS_D; ++r) {
			const uint32_t col = jhash2_u64(ser_glob_id, GYS_SEED + r) & (GYS_CMS_W - 1);
			if (fresh) atomicAdd(&p.cms32[r * GYS_CMS_W + col], 1u);
			if (bytes_sent + bytes_rcvd) atomicAdd(&p.cms64[r * GYS_CMS_W + col], (unsigned long long)(bytes_sent + bytes_rcvd));
		}
		return;
	}
	const unsigned long long cnt = (fresh ? 1ull : 0ull) + (closed ? (1ull << 32) : 0ull);
	unsigned long long *c = p.svc_win + (size_t)slot * 3;
	if (cnt) atomicAdd(&c[0], cnt);
	if (bytes_sent) atomicAdd(&c[1], (unsigned long long)bytes_sent);
	if (bytes_rcvd) atomicAdd(&c[2], (unsigned long long)bytes_rcvd);
}

#define GYS_CONN_EMPTY (~0ull)  // empty entry of the workgroup's service table (a record that names this glob_id goes the direct way)
#ifndef GYS_CONN_HQ
#define GYS_CONN_HQ 512u       // HLL candidates a workgroup parks before it reads a register
#endif
#define GYS_CONN_CNT_BITS 21u   // per-entry counts: records with notified_before_ clear | closes | records, 21 bits each (a workgroup walks < 2^21 records)
__device__ __forceinline__ void conn_one(const ConnP &p, const ConnRec &rc, bool live, unsigned long long *s_gid, unsigned long long (*s_acc)[3], uint32_t *s_tally, uint32_t *s_hq)
{
	uint32_t c128[4], s128[4], c32, s32;
	uint16_t cport, sport;
	// flow key: PAIR_IP_PORT(nat_cli_, nat_ser_)  (server/gy_mconnhdlr.cc:8707)
	c128[0] = (uint32_t)rc.a0; c128[1] = (uint32_t)(rc.a0 >> 32); c128[2] = (uint32_t)rc.a1; c128[3] = (uint32_t)(rc.a1 >> 32);
	c32 = (uint32_t)rc.a2;
	cport = (uint16_t)rc.a3;
	s128[0] = (uint32_t)rc.b0; s128[1] = (uint32_t)(rc.b0 >> 32); s128[2] = (uint32_t)rc.b1; s128[3] = (uint32_t)(rc.b1 >> 32);
	s32 = (uint32_t)rc.b2;
	sport = (uint16_t)rc.b3;
	const uint64_t tusec_close = rc.tusec_close, ser_glob_id = rc.ser_glob_id, bytes_sent = rc.bytes_sent, bytes_rcvd = rc.bytes_rcvd;
	const bool closed = tusec_close != 0, fresh = GYS_CONN_F_NOTIFIED(rc.flags) == 0, accept = GYS_CONN_F_ACCEPT(rc.flags) != 0,
		   connect = GYS_CONN_F_CONNECT(rc.flags) != 0;
	const bool listener_side = accept || !connect;

	// (wave-uniform control flow up to here: the tallies are ballots over the whole wave, `live` = the lane holds a record)
	conn_tally(s_tally, CONN_T_NEW, live && !closed);                  // nnew :9327
	conn_tally(s_tally, CONN_T_CLOSED, live && closed);                // nclosed :9133
	conn_tally(s_tally, CONN_T_CLOSED_NO_NOTIFY, live && closed && fresh); // nclosed_no_not :9135-9137
	conn_tally(s_tally, CONN_T_CLI_SIDE, live && !listener_side);
	if (!live) return;

	if (GYS_CONN_SKIP & 8) { // (timing experiments only: the record's words are read, nothing else)
		if ((c128[0] ^ c128[3] ^ s128[1] ^ c32 ^ s32 ^ cport ^ sport ^ (uint32_t)tusec_close ^ (uint32_t)ser_glob_id ^ (uint32_t)bytes_sent ^ (uint32_t)bytes_rcvd) == 0xDEADBEEFu)
			p.counters[CTR_CONN_UNKNOWN] = 1;
		return;
	}
	if (!(GYS_CONN_SKIP & 1)) {
		uint32_t w[10];
		const uint32_t nw = pair_words(c32, c128, cport, s32, s128, sport, w);
		const uint64_t h64 = hash64<10>(w, nw);
		uint32_t idx, rank;
		hll_idx_rank(h64, GYS_HLL_P, &idx, &rank);
		if (GYS_CONN_SKIP & 16) { // (hash only)
			if ((idx ^ rank) == 0xDEADBEEFu) p.counters[CTR_CONN_UNKNOWN] = 1;
		} else if (rank > s_tally[CONN_T_NUM + 1]) { // (at or below the workgroup's floor: no register is lower)
			// NO global read the record's walk depends on: a read here waits (vmcnt counts in order) for the NEXT round's record words
			// requested just before, i.e. it takes the prefetch's cover away.  The candidate is parked and checked at the workgroup's end.
			const uint32_t q = atomicAdd(&s_tally[CONN_T_NUM + 2], 1u);
			if (q < GYS_CONN_HQ) s_hq[q] = idx | (rank << 16);
			else if (p.hll32[idx] < rank) atomicMax(&p.hll32[idx], rank);
		}
	}
	if (GYS_CONN_SKIP & 2) return;

	if (p.pair32 && closed && (bytes_sent + bytes_rcvd) != 0) { // connlistenmap_ / connclientmap_ :9226-9319 (server/gy_msocket.h:240-290)
		const uint64_t task = rc.task;
		uint32_t *t32 = nullptr;
		unsigned long long *t64 = nullptr;
		if (ser_glob_id && accept) {
			t32 = p.pair32;
			t64 = p.pair64;
		} else if (!accept && connect && task) {
			t32 = p.cpair32;
			t64 = p.cpair64;
		}
		if (t32) {
#pragma unroll
			for (uint32_t r = 0; r < GYS_CMS_D; ++r) {
				const uint32_t col = jhash2_4w((uint32_t)ser_glob_id, (uint32_t)(ser_glob_id >> 32), (uint32_t)task, (uint32_t)(task >> 32), GYS_SEED + r) &
						     (GYS_CMS_W - 1);
				atomicAdd(&t32[r * GYS_CMS_W + col], 1u);
				atomicAdd(&t64[r * GYS_CMS_W + col], (unsigned long long)(bytes_sent + bytes_rcvd));
			}
		}
	}
	if (!listener_side) return; // the client half: the accepting partha's record carries the connection into the service's counters
	if (GYS_CONN_SKIP & 4) {
		if (ser_glob_id == 0xDEADBEEFu) p.counters[CTR_CONN_UNKNOWN] = 1;
		return;
	}
	// the workgroup's LDS entry of the service, keyed by ser_glob_id_ itself: open addressing, 1024 entries of {glob_id, counts, bytes
	// sent, bytes received}.  The glob_id -> slot lookup (a dependent global read, see above) happens once per ENTRY when the workgroup
	// flushes, not once per record.  The records of a partha arrive together, so the thousands of records of a workgroup name a few
	// hundred services; when they do not (hosts mixed record by record) the table fills: past three quarters a record probes twice,
	// below that sixteen times, and one that finds no entry goes the direct way.
	const unsigned long long add0 = (fresh ? 1ull : 0ull) | (closed ? 1ull << GYS_CONN_CNT_BITS : 0ull) | (1ull << (2u * GYS_CONN_CNT_BITS));
	uint32_t *const s_fill = s_tally + CONN_T_NUM; // entries of the table in use
	uint32_t h = (uint32_t)((ser_glob_id * 0x9E3779B97F4A7C15ull) >> (64u - GYS_CONN_AGG_BITS));
	// (a relaxed atomic load, NOT a volatile one: hipcc turns a volatile read through this pointer into a FLAT load followed by
	// s_waitcnt vmcnt(0) -- which waits for the next round's fourteen record loads in the middle of this round's work, r4s / r4t)
	const uint32_t probes = __atomic_load_n(s_fill, __ATOMIC_RELAXED) >= GYS_CONN_AGG / 4u * 3u ? 2u : 16u;
	if (ser_glob_id == GYS_CONN_EMPTY) {
		conn_direct(p, ser_glob_id, fresh, closed, bytes_sent, bytes_rcvd, s_tally);
		return;
	}
	for (uint32_t t = 0;; ++t) {
		const unsigned long long prev = atomicCAS(&s_gid[h], GYS_CONN_EMPTY, (unsigned long long)ser_glob_id);
		if (prev == GYS_CONN_EMPTY) atomicAdd(s_fill, 1u);
		if (prev == GYS_CONN_EMPTY || prev == ser_glob_id) break;
		h = (h + 1u) & (GYS_CONN_AGG - 1u);
		if (t + 1u == probes) {
			conn_direct(p, ser_glob_id, fresh, closed, bytes_sent, bytes_rcvd, s_tally);
			return;
		}
	}
	atomicAdd(&s_acc[h][0], add0);
	if (bytes_sent) atomicAdd(&s_acc[h][1], (unsigned long long)bytes_sent);
	if (bytes_rcvd) atomicAdd(&s_acc[h][2], (unsigned long long)bytes_rcvd);
}

// Round 3: the records are read THROUGH LDS.  Round 2 had every lane read its own 280-byte record with thirteen 8-byte loads at a 280-byte
// stride: every load instruction of a wave touched 64 different lines, each line was asked for by up to seven instructions in a row while
// still in flight, and reading alone took 3.3 of the kernel's 3.4 ms (r3v: 1.4 TB/s).  Now a wave reads what the roll-up needs of its 64
// records in small units laid out across the lanes -- lane l of load t takes unit 64 t + l, i.e. neighbouring lanes cover one record --
// through each record's own offset (no assumption that records are contiguous or of equal size), parks them in its private LDS region,
// and every lane then reads its record's fourteen words from LDS.  Round 3 read ten 16-byte pieces ([64, 224)), round 4 at first nine
// (the eight that hold a field of the roll-up plus the flag bytes [264, 280); eleven were 1.31 ms, r4d) at a 152-byte staged stride with
// eight waves per CU.  After r4h the staged record is the FOURTEEN 8-byte words the roll-up reads and nothing else (112 of those 144
// bytes): 7.5 KB per wave instead of 9.5, so that TWELVE waves fit a CU's LDS beside the aggregation table, and the kernel -- bound by the
// latency of its record reads at two waves per SIMD -- has three per SIMD in flight (r4i: 1.26 -> 1.155 ms per 2^24 records; the same
// staging at eight waves 1.24).  Without the prefetch registers (r4w) the kernel needs 120 VGPRs, and with a 1024-entry service table
// SIXTEEN waves fit (r4x: 1.02 -> 0.98 ms).  A workgroup is 1024 threads and walks its span of records (conn_span) in rounds of 1024.
#define GYS_CONN_STAGE_STRIDE 120u // bytes per staged record (112 used; 30 words: 8-byte accesses at this stride spread over all banks)
#define GYS_CONN_UNITS 14u
#ifndef GYS_CONN_PREFETCH
#define GYS_CONN_PREFETCH 0 // 1: the units of round k + 1 are requested before round k is worked on (see k_conn_ingest)
#endif
#ifndef GYS_CONN_FLOOR
#define GYS_CONN_FLOOR 1
#endif
// record offset of unit k: [64, 128) = nat_cli_, nat_ser_ (k = 0..7), 136 tusec_close_ (8), 144 cli_task_aggr_id_ (9), 208 bytes_sent_ (10),
// 216 bytes_rcvd_ (11), 192 ser_glob_id_ (12), 272 the flag bytes (13)
#define GYS_CONN_UNIT_OFF(k) ((k) < 8u ? 64u + 8u * (k) : 8u * (uint32_t)((0x22181B1A1211ull >> (((k) - 8u) * 8u)) & 0xFFull))
static_assert((GYS_CONN_THREADS / 64u) * 64u * GYS_CONN_STAGE_STRIDE + GYS_CONN_AGG * 32u + GYS_CONN_HQ * 4u + 64u <= 160u * 1024u, "k_conn_ingest: LDS");
__global__ __launch_bounds__(GYS_CONN_THREADS) void k_conn_ingest(ConnP p)
{
	// Records reach madhava message by message, a message = up to 2048 connections of ONE partha (comm::TCP_CONN_NOTIFY::MAX_NUM_CONNS,
	// common/gy_comm_proto.h:1738) and a partha has a few hundred listeners at most: the 1024 records of a workgroup touch few
	// distinct services.  Their three window accumulators are therefore summed in an LDS table keyed by the service's glob_id first and
	// flushed with one lookup and one set of device atomics per DISTINCT service of the workgroup.
	__shared__ unsigned long long s_gid[GYS_CONN_AGG];
	__shared__ unsigned long long s_acc[GYS_CONN_AGG][3];
	__shared__ uint32_t s_hq[GYS_CONN_HQ];       // parked HLL candidates: register index | rank << 16
	__shared__ uint32_t s_tally[CONN_T_NUM + 3]; // (+ the table's fill count, + the HLL floor, + the parked candidates)
	__shared__ __align__(16) uint8_t s_stage[GYS_CONN_THREADS / 64u][64u * GYS_CONN_STAGE_STRIDE];
	uint32_t fl = 0xFFFFFFFFu;
	if (!(GYS_CONN_SKIP & 1) && GYS_CONN_FLOOR)
		for (uint32_t k = threadIdx.x; k < (1u << GYS_HLL_P); k += GYS_CONN_THREADS) fl = min(fl, p.hll32[k]);
	else if (!GYS_CONN_FLOOR) fl = 0; // (A/B: every record reads its register)
	for (uint32_t k = threadIdx.x; k < GYS_CONN_AGG; k += GYS_CONN_THREADS) {
		s_gid[k] = GYS_CONN_EMPTY;
		s_acc[k][0] = 0;
		s_acc[k][1] = 0;
		s_acc[k][2] = 0;
	}
	if (threadIdx.x <= CONN_T_NUM) s_tally[threadIdx.x] = 0;
	if (threadIdx.x == CONN_T_NUM + 1) s_tally[threadIdx.x] = 0xFFFFFFFFu;
	if (threadIdx.x == CONN_T_NUM + 2) s_tally[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	{
		// HLL floor of the workgroup: the lowest of the 2^14 registers as they stand now (64 KB, read side by side, in flight while the
		// table above was cleared).  A register only grows, so a record whose rank is at or below the floor cannot change its register
		// and does not read it -- after the first few thousand flows of a window that is nearly every record, and the dependent
		// register read behind the flow hash was 0.15 of the kernel's 1.05 ms (r4q).
		fl = wave_min_u32(fl);
		if (lane == 0) atomicMin(&s_tally[CONN_T_NUM + 1], fl);
		__syncthreads();
	}
	uint8_t *const st = s_stage[wave];
	const uint64_t first64 = (uint64_t)blockIdx.x * p.span;
	if (first64 >= p.n) return; // (uniform: the grid is ceil(n / span))
	const uint32_t first = (uint32_t)first64, end = (uint32_t)min((uint64_t)p.n, first64 + p.span); // the workgroup's records [first, end)
	// ---- 896 units of 8 bytes per wave and round: unit q = 14 r + k is bytes [GYS_CONN_UNIT_OFF(k), + 8) of the wave's record r, so that
	// neighbouring lanes cover one record and a load instruction covers ~4.6 records (each lane reading its own record at the 280-byte stride
	// was 3.3 ms, r3v).  GYS_CONN_PREFETCH 1 requests the units of round k + 1 right after round k's were stored to the LDS.  That paid at
	// eight waves per CU (1.35 -> 1.29 ms, r4h); at twelve waves, with nothing in the walk waiting for a global read any more, the waves
	// cover each other and the early request costs: 1.06 with, 1.02 ms without (r4v / r4w; the r4n kernel, whose walk read the HLL register
	// and the glob_id table per record, was 1.05 with and 1.07 without) -- so it is off.
	// The record offsets of a round are loaded one round before its units are requested: the unit addresses depend on them, and a load
	// issued at request time put its whole latency in front of the fourteen unit loads (s_waitcnt vmcnt(0) before the first shuffle).
	uint2 pc[GYS_CONN_UNITS];
	auto load_off = [&](uint32_t round) -> uint32_t {
		const uint32_t i0 = first + round * GYS_CONN_THREADS + wave * 64u;
		if (i0 >= end) return 0u;
		const uint32_t i = i0 + lane;
		return p.offsets[i < end ? i : end - 1u];
	};
	auto request = [&](uint32_t round, uint32_t off) {
		const uint32_t i0 = first + round * GYS_CONN_THREADS + wave * 64u;
		if (i0 >= end) return;
		const uint32_t nrec = min(64u, end - i0);
#pragma unroll
		for (uint32_t t = 0; t < GYS_CONN_UNITS; ++t) {
			const uint32_t q = t * 64u + lane;
			const uint32_t r = (q * 4682u) >> 16; // q / 14 for q < 896
			const uint32_t k = q - r * GYS_CONN_UNITS;
			const uint32_t ro = (uint32_t)__shfl((int)off, (int)min(r, nrec - 1u), 64); // (a unit past the wave's last record re-reads that record)
			const uint32_t *src = (const uint32_t *)(p.batch + ro + GYS_CONN_UNIT_OFF(k)); // records start 8-byte aligned (COMM_HEADER::validate common/gy_comm_proto.cc:23-26)
			pc[t] = make_uint2(src[0], src[1]);
		}
	};
	uint32_t off_nx = load_off(0);
	if (GYS_CONN_PREFETCH) {
		const uint32_t o = off_nx;
		off_nx = load_off(1);
		request(0, o);
	}
#pragma unroll 1
	for (uint32_t round = 0;; ++round) {
		const uint32_t i0 = first + round * GYS_CONN_THREADS + wave * 64u; // the wave's first record
		if (i0 >= end) break;
		const uint32_t i = i0 + lane;
		if (!GYS_CONN_PREFETCH) {
			const uint32_t o = off_nx;
			off_nx = load_off(round + 1u);
			request(round, o);
		}
#pragma unroll
		for (uint32_t t = 0; t < GYS_CONN_UNITS; ++t) {
			const uint32_t q = t * 64u + lane;
			const uint32_t r = (q * 4682u) >> 16;
			const uint32_t k = q - r * GYS_CONN_UNITS;
			*(uint64_t *)(st + r * GYS_CONN_STAGE_STRIDE + 8u * k) = (uint64_t)pc[t].x | ((uint64_t)pc[t].y << 32);
		}
		if (GYS_CONN_PREFETCH) {
			const uint32_t o = off_nx;
			off_nx = load_off(round + 2u);
			request(round + 1u, o);
		}
		GYS_WAVE_SYNC();
		{
			const uint64_t *rw = (const uint64_t *)(st + lane * GYS_CONN_STAGE_STRIDE);
			ConnRec rc;
			rc.a0 = rw[0]; rc.a1 = rw[1]; rc.a2 = rw[2]; rc.a3 = rw[3];
			rc.b0 = rw[4]; rc.b1 = rw[5]; rc.b2 = rw[6]; rc.b3 = rw[7];
			rc.tusec_close = rw[8];   // @136
			rc.task = rw[9];          // @144
			rc.bytes_sent = rw[10];   // @208
			rc.bytes_rcvd = rw[11];   // @216
			rc.ser_glob_id = rw[12];  // @192
			rc.flags = rw[13];        // @272
			conn_one(p, rc, i < end, s_gid, s_acc, s_tally, s_hq);
		}
		GYS_WAVE_SYNC(); // (the region is rewritten by the next round)
	}
	__syncthreads();
	// ---- the workgroup's flush: everything that reads global state the walk did not wait for
	{
		const uint32_t nq = min(s_tally[CONN_T_NUM + 2], GYS_CONN_HQ); // parked HLL candidates
		for (uint32_t q = threadIdx.x; q < nq; q += GYS_CONN_THREADS) {
			const uint32_t e = s_hq[q], idx = e & 0xFFFFu, rank = e >> 16;
			if (p.hll32[idx] < rank) atomicMax(&p.hll32[idx], rank);
		}
	}
	for (uint32_t k = threadIdx.x; k < GYS_CONN_AGG; k += GYS_CONN_THREADS) {
		const uint64_t gid = s_gid[k];
		if (gid == GYS_CONN_EMPTY) continue;
		constexpr unsigned long long M = (1ull << GYS_CONN_CNT_BITS) - 1ull;
		const unsigned long long a0 = s_acc[k][0], nconn = a0 & M, nclose = (a0 >> GYS_CONN_CNT_BITS) & M, nrec = a0 >> (2u * GYS_CONN_CNT_BITS);
		const unsigned long long sent = s_acc[k][1], rcvd = s_acc[k][2];
		const uint32_t slot = tbl_lookup(p.gid, gid);
		if (slot == GYS_NOSLOT) { // as conn_direct, for all of the entry's records at once (the Count-Min adds of a record commute)
			atomicAdd(&s_tally[CONN_T_UNKNOWN], (uint32_t)nrec);
#pragma unroll
			for (uint32_t r = 0; r < GYS_CMS_D; ++r) {
				const uint32_t col = jhash2_u64(gid, GYS_SEED + r) & (GYS_CMS_W - 1);
				if (nconn) atomicAdd(&p.cms32[r * GYS_CMS_W + col], (uint32_t)nconn);
				if (sent + rcvd) atomicAdd(&p.cms64[r * GYS_CMS_W + col], sent + rcvd);
			}
			continue;
		}
		unsigned long long *c = p.svc_win + (size_t)slot * 3;
		const unsigned long long cnt = nconn | (nclose << 32); // a window's connection count of one service stays far below 2^32
		if (cnt) atomicAdd(&c[0], cnt);
		if (sent) atomicAdd(&c[1], sent);
		if (rcvd) atomicAdd(&c[2], rcvd);
	}
	__syncthreads(); // (the flush adds to the unknown-service tally)
	if (threadIdx.x == 0) { // records of this workgroup: ONE add per workgroup (round 2 added once per wave: 2.6 x 10^5 adds on one address per 2^24 records)
		atomicAdd((unsigned long long *)&p.counters[CTR_CONN_EVENTS], (unsigned long long)(end - first));
	}
	if (threadIdx.x < CONN_T_NUM && s_tally[threadIdx.x]) { // ... and one per tally of the walk
		const int ctr = threadIdx.x == CONN_T_NEW ? CTR_CONN_NEW : threadIdx.x == CONN_T_CLOSED ? CTR_CONN_CLOSED :
				threadIdx.x == CONN_T_CLOSED_NO_NOTIFY ? CTR_CONN_CLOSED_NO_NOTIFY : threadIdx.x == CONN_T_CLI_SIDE ? CTR_CONN_CLI_SIDE : CTR_CONN_UNKNOWN;
		atomicAdd((unsigned long long *)&p.counters[ctr], (unsigned long long)s_tally[threadIdx.x]);
	}
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
	unsigned long long *claim;         // [nsvc] launch << 32 | (record index + 1) of the LAST record of the call that names the listener
	uint32_t launch;                   // number of this ingest call (never 0)
};

// Per-workgroup LDS roll-up of the walk's sums (round 4, after r4g): the records of one partha arrive together, so the 256 records of a
// workgroup name two or three hosts, and round 3's eight device-scope adds per record (plus one per record on the call's record counter)
// lined up on a handful of addresses -- 70 of the call's 86 us at 10^5 records.  Hosts claim one of 64 LDS rows (open addressing, eight
// probes; a record that finds none adds to the device rows directly) and the rows are added to LISTEN_SUMM_STATS once per workgroup.
#define GYS_LS_AGG 64u
struct LStateAgg {
	uint32_t hk[GYS_LS_AGG];
	int32_t hs[GYS_LS_AGG][13];
	uint32_t cnt[3]; // missed, deleted, errors
};
__device__ __forceinline__ void lstate_agg_init(LStateAgg &a)
{
	for (uint32_t k = threadIdx.x; k < GYS_LS_AGG; k += blockDim.x) {
		a.hk[k] = GYS_NOSLOT;
#pragma unroll
		for (int j = 0; j < 13; ++j) a.hs[k][j] = 0;
	}
	if (threadIdx.x < 3) a.cnt[threadIdx.x] = 0;
	__syncthreads();
}
__device__ __forceinline__ void lstate_agg_flush(const LStateP &p, LStateAgg &a, uint32_t first)
{
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < GYS_LS_AGG * 13u; k += blockDim.x) {
		const uint32_t row = k / 13u, j = k - row * 13u;
		const int32_t v = a.hs[row][j];
		if (v && a.hk[row] != GYS_NOSLOT) atomicAdd(&p.host_summ[(size_t)a.hk[row] * 16 + j], v);
	}
	if (threadIdx.x < 3 && a.cnt[threadIdx.x]) {
		const int ctr = threadIdx.x == 0 ? CTR_LSTATE_MISSED : threadIdx.x == 1 ? CTR_LSTATE_DELETED : CTR_LSTATE_ERRORS;
		atomicAdd((unsigned long long *)&p.counters[ctr], (unsigned long long)a.cnt[threadIdx.x]);
	}
	if (threadIdx.x == 3 && first < p.n)
		atomicAdd((unsigned long long *)&p.counters[CTR_LSTATE_RECORDS], (unsigned long long)min((uint32_t)blockDim.x, p.n - first));
}

__device__ __forceinline__ void lstate_ingest_one(const LStateP &p, uint32_t i, LStateAgg &a)
{
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

	const uint32_t slot = tbl_lookup(p.gid, glob_id); // listen_tbl_.lookup_single_elem_locked(glob_id, get_uint64_hash(glob_id)) :11183
	if (slot == GYS_NOSLOT) {
		atomicAdd(&a.cnt[0], 1u); // nmissed++ :11185-11188
		return;
	}
	if (query_flags == 0xC0u) { // LISTEN_FLAG_DELETE :11194
		atomicAdd(&a.cnt[1], 1u);
		atomicMax(&p.claim[slot], ((unsigned long long)p.launch << 32) | (unsigned long long)(i + 1u)); // (k_lstate_keep: state no longer current)
		return;
	}
	if (curr_state > 5u) { // :11250-11256
		atomicAdd(&a.cnt[2], 1u);
		return;
	}
	const uint32_t host = p.host_slot ? p.host_slot[i] : p.single_host;
	uint32_t row = (host * 0x9E3779B1u) >> 26;
	bool lds = false;
#pragma unroll 1
	for (int t = 0; t < 8; ++t, row = (row + 1u) & (GYS_LS_AGG - 1u)) {
		const uint32_t prev = atomicCAS(&a.hk[row], GYS_NOSLOT, host);
		if (prev == GYS_NOSLOT || prev == host) { lds = true; break; }
	}
	// LISTEN_SUMM_STATS::update server/gy_msocket.h:853-865 (per-record integer quotient nqrys_5s_/5)
	if (lds) {
		int32_t *s = a.hs[row];
		atomicAdd(&s[curr_state], 1);
		if (nqrys_5s / 5u) atomicAdd(&s[6], (int32_t)(nqrys_5s / 5u));
		if (nconns_active) atomicAdd(&s[7], (int32_t)nconns_active);
		if (kb_in) atomicAdd(&s[8], (int32_t)kb_in);
		if (kb_out) atomicAdd(&s[9], (int32_t)kb_out);
		if (ser_errors) atomicAdd(&s[10], (int32_t)ser_errors);
		atomicAdd(&s[11], 1);
		if (nqrys_5s) atomicAdd(&s[12], 1);
	} else {
		int32_t *s = p.host_summ + (size_t)host * 16;
		atomicAdd(&s[curr_state], 1);
		if (nqrys_5s / 5u) atomicAdd(&s[6], (int32_t)(nqrys_5s / 5u));
		if (nconns_active) atomicAdd(&s[7], (int32_t)nconns_active);
		if (kb_in) atomicAdd(&s[8], (int32_t)kb_in);
		if (kb_out) atomicAdd(&s[9], (int32_t)kb_out);
		if (ser_errors) atomicAdd(&s[10], (int32_t)ser_errors);
		atomicAdd(&s[11], 1);
		if (nqrys_5s) atomicAdd(&s[12], 1);
	}
	// MTCP_LISTENER::set_state server/gy_msocket.h:1410-1437 keeps the 88-byte record; the reference walks a message serially, so when
	// several records of one call name the listener (a backlog of 5-s messages in one buffer) the LAST one stays, whole: the records
	// claim the listener here and the owner of the claim stores in k_lstate_keep
	atomicMax(&p.claim[slot], ((unsigned long long)p.launch << 32) | (unsigned long long)(i + 1u));
	if (p.qps_hist) {
		// the per-listener QPS_HISTOGRAM / ACTIVE_CONN_HISTOGRAM behind LISTENER_DAY_STATS (common/gy_socket_stat.h:548-549, :633-635;
		// one sample per 5-s state record: gy_socket_stat.cc:4109-4127), fed from the record's own nqrys_5s_/5 and nconns_active_
		hist_add_atomic(hash_def(GYS_SEMI_LOG_HASH_LO), GYS_SEMI_LOG_HASH_LO, &p.qps_hist[slot], (int64_t)(int32_t)(nqrys_5s / 5u));
		hist_add_atomic(hash_def(GYS_HASH_1_3000), GYS_HASH_1_3000, &p.act_hist[slot], (int64_t)(int32_t)nconns_active);
	}
}

// second pass of a LISTENER_STATE_NOTIFY call: the record that holds its listener's claim (the last one of the call in stream order) stores
// the kept state -- or, for a record flagged LISTEN_FLAG_DELETE, marks the state as no longer current
__device__ __forceinline__ void lstate_keep_one(const LStateP &p, uint32_t i)
{
	const uint64_t *q = (const uint64_t *)(p.batch + p.offsets[i]);
	const uint64_t glob_id = q[0], w9 = q[9], w10 = q[10];
	const uint32_t curr_state = (uint32_t)((w9 >> 56) & 0xFF), query_flags = (uint32_t)((w10 >> 32) & 0xFF);
	const uint32_t slot = tbl_lookup(p.gid, glob_id);
	if (slot == GYS_NOSLOT) return;
	const bool del = query_flags == 0xC0u;
	if (!del && curr_state > 5u) return; // (skipped by the walk: never claimed)
	if (p.claim[slot] != (((unsigned long long)p.launch << 32) | (unsigned long long)(i + 1u))) return;
	uint64_t *d = (uint64_t *)(p.svc_state + (size_t)slot * 96);
	if (del) {
		*(uint32_t *)(d + 11) = 0;
		return;
	}
	const uint32_t host = p.host_slot ? p.host_slot[i] : p.single_host;
#pragma unroll
	for (int k = 0; k < 11; ++k) d[k] = q[k];
	d[11] = (uint64_t)p.epoch | ((uint64_t)host << 32);
}

__global__ __launch_bounds__(256) void k_lstate_ingest(LStateP p)
{
	__shared__ LStateAgg s_agg;
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	lstate_agg_init(s_agg);
	if (i < p.n) lstate_ingest_one(p, i, s_agg);
	lstate_agg_flush(p, s_agg, blockIdx.x * blockDim.x);
}

__global__ __launch_bounds__(256) void k_lstate_keep(LStateP p)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < p.n) lstate_keep_one(p, i);
}

// a partha's message (<= 512 records, comm::LISTENER_STATE_NOTIFY::MAX_NUM_LISTENERS) in ONE launch: both passes by one workgroup, the claims of
// the first settled at the workgroup barrier (device-scope atomics on the claim words, read back by threads of the same workgroup)
#define GYS_LSTATE_FUSED_MAX 1024u
__global__ __launch_bounds__(GYS_LSTATE_FUSED_MAX) void k_lstate_both(LStateP p)
{
	__shared__ LStateAgg s_agg;
	const uint32_t i = threadIdx.x;
	lstate_agg_init(s_agg);
	if (i < p.n) lstate_ingest_one(p, i, s_agg);
	lstate_agg_flush(p, s_agg, 0u);
	__syncthreads();
	if (i < p.n) lstate_keep_one(p, i);
}

// ---------------------------------------------------------------------------------------------------- window boundary
struct PrepP {
	const int32_t *host_summ;       // [nhosts*16]
	const gys_host_state *host_state;
	const uint32_t *host_state_epoch;
	const uint32_t *host_cluster;
	uint32_t nhosts;
	uint32_t epoch;
	const uint32_t *d_epoch;        // the window number on the device (a captured graph cannot carry it as a launch parameter); nullptr: `epoch`
	uint32_t *cluster_state;        // arena [max_clusters*12]
	const uint32_t *hll32;
	uint8_t *hll8;                  // arena
};

// CLUSTER_STATE_ONE::update_from_state server/gy_mconnhdlr.cc:16032-16050 for every host whose host state is current
// (send_cluster_state skips hosts without a recent state, :16068-16070)
__global__ void k_epoch_inc(uint32_t *d_epoch) { *d_epoch += 1u; }

// The fixed tail of a window close in ONE launch (round 5; it was a copy, three fills, a host-to-device copy, another copy + fill and a
// one-thread kernel: nine graph nodes of 10 - 20 us each for ~10 MB of traffic): the window's (reduced) registers are kept for the
// queries (last = arena), the next window starts from zero (arena, the first-pass HLL filter, the per-host summaries; the i64 MAX
// section's cell starts at INT64_MIN), the device copy of the window number follows the host's.
struct WinFinishP {
	uint4 *arena, *last;
	uint64_t n16;        // 16-byte pieces of the arena
	uint64_t i64max_at;  // byte offset of the i64 MAX cell (8-byte aligned)
	uint4 *hll32;
	uint64_t hll16;
	uint4 *hs_win, *hs_last;
	uint64_t hs16;
	uint32_t *d_epoch;
};

__global__ __launch_bounds__(256) void k_window_finish(WinFinishP p)
{
	const uint64_t t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
	const uint64_t mx = p.i64max_at >> 4;
	for (uint64_t i = t0; i < p.n16; i += stride) {
		p.last[i] = p.arena[i];
		uint4 z = make_uint4(0, 0, 0, 0);
		if (i == mx) { // INT64_MIN in the cell's half of the piece
			if (p.i64max_at & 8u) z.w = 0x80000000u;
			else z.y = 0x80000000u;
		}
		p.arena[i] = z;
	}
	for (uint64_t i = t0; i < p.hll16; i += stride) p.hll32[i] = make_uint4(0, 0, 0, 0);
	for (uint64_t i = t0; i < p.hs16; i += stride) {
		p.hs_last[i] = p.hs_win[i];
		p.hs_win[i] = make_uint4(0, 0, 0, 0);
	}
	if (t0 == 0) *p.d_epoch += 1u;
}

__global__ void k_window_prepare(PrepP p)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < (1u << GYS_HLL_P) / 4u) {
		const uint4 v = ((const uint4 *)p.hll32)[i];
		((uint32_t *)p.hll8)[i] = (v.x & 0xFF) | ((v.y & 0xFF) << 8) | ((v.z & 0xFF) << 16) | ((v.w & 0xFF) << 24);
	}
	if (i >= p.nhosts) return;
	if (p.host_state_epoch[i] != (p.d_epoch ? *p.d_epoch : p.epoch)) return;
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

// window / all-time view of a service's records.  meta != nullptr (t-digest on, lazy fold; the caller has folded the range): the
// all-time record holds every folded value, the window record is valid when it belongs to the open window.  meta == nullptr (eager
// records, updated per event): the all-time record gets the window added at the boundary (k_hist_fold), so the view adds the open
// window itself -- both modes answer "everything ingested so far".
__device__ __forceinline__ gys_hist_rec hist_view(const gys_hist_rec *win, const gys_hist_rec *all, const TdMeta *meta, uint32_t epoch, int which, uint32_t slot)
{
	gys_hist_rec r;
	if (which == 0) {
		if (!meta || meta[slot].hw_epoch == epoch) return win[slot];
		for (int i = 0; i < 15; ++i) {
			r.stats[i].count = 0;
			r.stats[i].sum = 0;
		}
		r.total_count = 0;
		r.max_val_seen = INT64_MIN;
		return r;
	}
	r = all[slot];
	if (meta) return r;
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
	int64_t *first_sec; // [nsvc] time of the service's first window close = firstTime_ of its series (0: none yet)
	int64_t tnow;
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
		const bool cur = !p.meta || p.meta[slot].hw_epoch == p.epoch;
		ulonglong2 closing;
		if (cur) {
			closing = w;
		} else { // the key was not touched in the closing window: win still holds an older window
			closing.x = 0;
			closing.y = k < 15u ? 0ull : (unsigned long long)INT64_MIN;
		}
		if (p.last) ((ulonglong2 *)p.last)[t] = closing;
		if (k == 15u && p.first_sec[slot] == 0) p.first_sec[slot] = p.tnow; // BucketedTimeSeries::update on an empty series
		if (p.mask[0] | p.mask[1]) {
			// cumulative record BEFORE the closing window: its add happens at the close time, i.e. at or after the boundary.  Lazily
			// folded records (meta) already hold the closing window (the caller folded every service first): take it out again.
			ulonglong2 before = ((const ulonglong2 *)p.all)[t];
			if (p.meta && cur) {
				before.x -= w.x;
				if (k < 15u) before.y -= w.y;
			}
			for (int li = 0; li < 2; ++li)
				for (uint32_t m = p.mask[li]; m; m &= m - 1) {
					const uint32_t j = (uint32_t)__builtin_ctz(m);
					((ulonglong2 *)(p.snap + ((uint64_t)li * GYS_LEVEL_RING + j) * p.stride))[t] = before;
				}
		}
	}
}

// enable_levels = 2, a close that crosses no ring boundary: only firstTime_ of a service's series (the time of its first window close)
__global__ __launch_bounds__(256) void k_level_first(int64_t *first_sec, uint32_t nsvc, int64_t tnow)
{
	for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nsvc; s += gridDim.x * blockDim.x)
		if (first_sec[s] == 0) first_sec[s] = tnow;
}

struct LevelViewP {
	const gys_hist_rec *win, *all;
	const TdMeta *meta;
	uint32_t epoch_open;     // windows before this one have been added (closed)
	uint32_t first, n;
	const gys_hist_rec *sub; // mode 0: snapshot to subtract (nullptr: nothing); mode 2: the last-window records
	int mode;                // 0 cumulative - sub, 1 empty, 2 copy of sub
	const uint32_t *last_tag; // mode 2 with lazily folded records: sub[slot] is the service's record of window last_tag[slot] -- the closed window's
	uint32_t last_epoch;      // only when that equals last_epoch (the array is the former hist_win, swapped in at the close; nullptr: always)
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
		if (p.meta[slot].hw_epoch == p.epoch_open) { // the folded part of the OPEN window is not in any level yet
			cum.x -= w.x;
			if (k < 15u) cum.y -= w.y;
		}
	} else if (k == 15u && (long long)cum.y < (long long)w.y) {
		cum.y = w.y; // the maximum is reported over everything seen
	}
	ulonglong2 r;
	if (p.mode == 0) {
		r = cum;
		if (p.sub) {
			const ulonglong2 s = ((const ulonglong2 *)p.sub)[g];
			r.x -= s.x;
			if (k < 15u) r.y -= s.y;
		}
	} else if (p.mode == 2 && (!p.last_tag || p.last_tag[slot] == p.last_epoch)) {
		r = ((const ulonglong2 *)p.sub)[g];
		if (k == 15u) r.y = cum.y;
	} else {
		r.x = 0;
		r.y = k < 15u ? 0ull : cum.y;
	}
	((ulonglong2 *)p.out)[t] = r;
}

// TIME_HISTOGRAM::get_stats_for_period (common/gy_statistics.h:1378-1406): per histogram bucket count(start, end) / sum(start, end) of the
// level folly's MultiLevelTimeSeries::getLevel(start) picks.  A ring bucket [s, s + w) of a level holds C(s + w) - C(s), C(x) = the
// cumulative record of the adds before x = the snapshot taken at x (k_level_roll) or, for an x after the last close, the cumulative
// record now; a bucket that only partly overlaps the interval is scaled by the overlapped fraction in float and truncated
// (BucketedTimeSeries::rangeAdjust).  Which buckets overlap, their fractions and the boundary snapshots are the same for every
// service (the host works them out once); the all-time level is one bucket [first close of the service, now + 1), scaled per service.
#define GYS_PERIOD_MAXB GYS_LEVEL_RING
struct LevelPeriodP {
	const gys_hist_rec *win, *all;
	const TdMeta *meta;
	uint32_t epoch_open;
	uint32_t first, n;
	int mode;                                      // 0 ring level, 1 empty, 2 the last-window record (level 0), 3 all-time level
	uint32_t nrb;                                  // mode 0: overlapping ring buckets, oldest first
	const gys_hist_rec *bnd[GYS_PERIOD_MAXB + 1];  // C(start of ring bucket i), [nrb] = C(end of the last one); nullptr: the cumulative record now
	float scale[GYS_PERIOD_MAXB];
	uint32_t whole_mask;                           // bit i: ring bucket i lies inside the interval (taken unscaled)
	const gys_hist_rec *last;                      // mode 2
	const uint32_t *last_tag;                      // mode 2: see LevelViewP
	uint32_t last_epoch;
	const int64_t *first_sec;                      // mode 3
	int64_t start, end, latest;                    // mode 3: [start, end) and latestTime_
	gys_hist_rec *out;                             // [n]: stats[b] = the interval's {count, sum}, total_count = their sum, max_val_seen all-time
};

__device__ __forceinline__ long long range_adjust(long long v, float scale) { return (long long)((float)v * scale); }

__global__ __launch_bounds__(256) void k_level_period(LevelPeriodP p)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const bool live = t < (uint64_t)p.n * 16ull;
	const uint64_t tt = live ? t : 0; // every lane stays for the 16-lane sums below
	const uint32_t slot = p.first + (uint32_t)(tt >> 4), k = (uint32_t)(tt & 15u);
	const uint64_t g = (uint64_t)slot * 16ull + k;
	ulonglong2 cum = ((const ulonglong2 *)p.all)[g];
	const ulonglong2 w = ((const ulonglong2 *)p.win)[g];
	if (p.meta) {
		if (p.meta[slot].hw_epoch == p.epoch_open) { // the folded part of the OPEN window is not in any level yet
			cum.x -= w.x;
			if (k < 15u) cum.y -= w.y;
		}
	} else if (k == 15u && (long long)cum.y < (long long)w.y) {
		cum.y = w.y;
	}
	long long ac = 0, as = 0;
	if (k < 15u) {
		if (p.mode == 0) {
			ulonglong2 lo = p.bnd[0] ? ((const ulonglong2 *)p.bnd[0])[g] : cum;
			for (uint32_t i = 0; i < p.nrb; ++i) {
				const ulonglong2 hi = p.bnd[i + 1] ? ((const ulonglong2 *)p.bnd[i + 1])[g] : cum;
				const long long c = (long long)(hi.x - lo.x), sm = (long long)(hi.y - lo.y);
				if ((p.whole_mask >> i) & 1u) {
					ac += c;
					as += sm;
				} else {
					ac += range_adjust(c, p.scale[i]);
					as += range_adjust(sm, p.scale[i]);
				}
				lo = hi;
			}
		} else if (p.mode == 2) {
			if (!p.last_tag || p.last_tag[slot] == p.last_epoch) {
				const ulonglong2 r = ((const ulonglong2 *)p.last)[g];
				ac = (long long)r.x;
				as = (long long)r.y;
			}
		} else if (p.mode == 3) {
			const int64_t bs = p.first_sec[slot];
			int64_t bn = p.latest + 1;
			if (bs != 0 && !(p.start >= bn) && !(p.end <= bs)) {
				if (p.start <= bs && p.end >= bn) {
					ac = (long long)cum.x;
					as = (long long)cum.y;
				} else {
					const int64_t is = p.start > bs ? p.start : bs, ie = p.end < bn ? p.end : bn;
					const float scale = (float)(ie - is) * 1.f / (float)(bn - bs);
					ac = range_adjust((long long)cum.x, scale);
					as = range_adjust((long long)cum.y, scale);
				}
			}
		}
	}
	long long tot = ac;
	for (int o = 1; o < 16; o <<= 1) tot += __shfl_xor(tot, o, 16);
	if (!live) return;
	ulonglong2 r;
	if (k < 15u) {
		r.x = (unsigned long long)ac;
		r.y = (unsigned long long)as;
	} else {
		r.x = (unsigned long long)tot;
		r.y = cum.y;
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


// ---------------------------------------------------------------------------------------------------- per-listener 5-second scan (row a9)
// TCP_SOCK_HANDLER::listener_stats_update (common/gy_socket_stat.cc:4044-4365) walks the listener table every 5 s and turns each
// listener's counters + histograms into one comm::LISTENER_STATE_NOTIFY; TCP_LISTENER::get_curr_state (:2030-2143) compares the 5-s p95
// bucket with the 5-min / 5-day ones and the QPS with the QPS histogram's p25 / p95.  Here: one thread per service, everything from the
// engine's own state -- the four time levels as (cumulative record - ring snapshot) / last closed window, the CONN_BITMAP rows of the
// window just closed, the per-service QPS / active-connection histograms -- into one 88-byte notify record + one gys_listener_scan
// record per service.  Nothing is modified.  The state POLICY (task / cpu / memory issue inputs, issue strings) is the caller's.
struct ListenerScanP {
	const gys_hist_rec *win, *all;
	const TdMeta *meta;
	uint32_t epoch_open; // windows before this one are in the levels
	uint32_t epoch_last; // the window closed last (its CONN_BITMAP rows are the ones get_conn_breakup sees)
	uint32_t nsvc;
	int mode[GYS_NLEVELS];               // per level: 0 cumulative - sub, 1 empty, 2 copy of sub (level 0: the last closed window)
	const gys_hist_rec *sub[GYS_NLEVELS];
	const uint32_t *last_tag;            // a mode-2 level: see LevelViewP
	uint32_t last_epoch;
	const gys_hist_rec *qps, *act;
	const uint32_t *bitmap;              // [nsvc * GYS_BM_WORDS] u32 = 32 x u16 IPv4 rows, 32 x u16 IPv6 rows
	const uint64_t *svc_gid;
	float multiple;                      // TCP_SOCK_HANDLER::get_bpf_qps_multiple()
	float diffsec;
	uint8_t *notify;                     // [nsvc * 88] or nullptr
	gys_listener_scan *scan;             // [nsvc] or nullptr
};

__device__ __forceinline__ uint32_t resp_bucketid_from_threshold(const HashDef &d, int64_t thr)
{
	for (int i = 0; i < d.nthr; ++i) // get_bucketid_from_threshold common/gy_statistics.h:517-531
		if (d.thr[i] == thr) return (uint32_t)i + 1u;
	return thr < 0 ? 0u : (uint32_t)d.nthr + 1u;
}

__global__ __launch_bounds__(256) void k_listener_scan(ListenerScanP p)
{
	const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= p.nsvc) return;
	const HashDef &dr = hash_def(GYS_RESP_TIME_HASH);
	gys_listener_scan o;
	memset(&o, 0, sizeof(o));
	o.glob_id = p.svc_gid[slot];
	const bool open_folded = p.meta && p.meta[slot].hw_epoch == p.epoch_open; // the folded part of the OPEN window is in no level yet
	const bool last_ok = !p.last_tag || p.last_tag[slot] == p.last_epoch;
	for (int lv = 0; lv < GYS_NLEVELS; ++lv) {
		gys_hist_rec r;
		int64_t ts = 0;
		uint64_t tc = 0;
		for (int b = 0; b < 15; ++b) {
			uint64_t cnt = 0;
			int64_t sum = 0;
			if (p.mode[lv] == 2) {
				if (last_ok) {
					cnt = p.sub[lv][slot].stats[b].count;
					sum = p.sub[lv][slot].stats[b].sum;
				}
			} else if (p.mode[lv] == 0) {
				cnt = p.all[slot].stats[b].count;
				sum = p.all[slot].stats[b].sum;
				if (open_folded) {
					cnt -= p.win[slot].stats[b].count;
					sum -= p.win[slot].stats[b].sum;
				}
				if (p.sub[lv]) {
					cnt -= p.sub[lv][slot].stats[b].count;
					sum -= p.sub[lv][slot].stats[b].sum;
				}
			}
			r.stats[b].count = cnt;
			r.stats[b].sum = sum;
			tc += cnt;
			ts += sum;
		}
		r.total_count = tc;
		r.max_val_seen = 0;
		o.tcount[lv] = (int64_t)tc; // slabhist.count(level) / sum(level), common/gy_statistics.h:1358-1359
		o.tsum[lv] = ts;
		o.p95_ms[lv] = (int32_t)level_percentile(dr, r, 95.0f); // RESP_STATS::stats_ {95, 99, 25}, common/gy_socket_stat.h:459-462
		o.p99_ms[lv] = (int32_t)level_percentile(dr, r, 99.0f);
		o.p25_ms[lv] = (int32_t)level_percentile(dr, r, 25.0f);
	}
	// total_queries = the listener's query counter over the interval (:4051-4052), one per response event that reached the histogram
	// (:1581) = the 5-s level's count; curr_qps_extra = total_queries * multiple_factor / diffsec (:4109)
	const uint32_t total_queries = (uint32_t)o.tcount[0];
	o.last_qps = p.diffsec > 0.f ? (int32_t)((float)total_queries * p.multiple / p.diffsec) : 0;
	{
		const int32_t q5 = (int32_t)(o.tcount[0] / 5);
		o.curr_qps = o.last_qps > q5 ? o.last_qps : q5; // :2083
	}
	o.b5 = (uint8_t)resp_bucketid_from_threshold(dr, o.p95_ms[0]); // :2085-2087
	o.b300 = (uint8_t)resp_bucketid_from_threshold(dr, o.p95_ms[1]);
	o.b5day = (uint8_t)resp_bucketid_from_threshold(dr, o.p95_ms[2]);
	if (p.qps) { // HIST_DATA stats_qps[] {95, 25}, stats_active[] {95, 25} (:2053, :2089-2090)
		int64_t dv, sum;
		uint64_t cnt;
		const gys_hist_rec q = p.qps[slot];
		hist_percentile(hash_def(GYS_SEMI_LOG_HASH_LO), q, 95.0f, &dv, &sum, &cnt);
		o.qps_p95 = (int32_t)dv;
		hist_percentile(hash_def(GYS_SEMI_LOG_HASH_LO), q, 25.0f, &dv, &sum, &cnt);
		o.qps_p25 = (int32_t)dv;
		const gys_hist_rec a = p.act[slot];
		hist_percentile(hash_def(GYS_HASH_1_3000), a, 95.0f, &dv, &sum, &cnt);
		o.act_p95 = (int32_t)dv;
		hist_percentile(hash_def(GYS_HASH_1_3000), a, 25.0f, &dv, &sum, &cnt);
		o.act_p25 = (int32_t)dv;
	}
	// CONN_BITMAP::get_conn_breakup (common/gy_socket_stat.h:413-429): per response bucket the rows (client port & 31) that saw it in
	// the window just closed; curr_active_conn = their maximum (:4143-4156; the inet_diag count it starts from is agent-side state)
	if (p.meta ? p.meta[slot].hw_epoch == p.epoch_last : true) {
		// (nactive_conn_arr_[r] = ipv4_conn[r] + ipv6_conn[r], :4147: the IPv6 rows are words 16..31 of the service)
		uint32_t w[GYS_BM_WORDS];
		for (uint32_t i = 0; i < GYS_BM_WORDS; ++i) w[i] = p.bitmap[(size_t)slot * GYS_BM_WORDS + i];
		for (int r = 0; r < 15; ++r) {
			uint32_t n = 0;
			for (uint32_t i = 0; i < GYS_BM_WORDS; ++i) n += ((w[i] >> r) & 1u) + ((w[i] >> (16 + r)) & 1u);
			o.nactive_conn_arr[r] = (uint8_t)n;
			if (o.nconn_active < n) o.nconn_active = (uint8_t)n;
		}
	}
	if (p.scan) p.scan[slot] = o;
	if (p.notify) { // :4293-4304; fields the engine cannot derive stay 0
		uint32_t *q = (uint32_t *)(p.notify + (size_t)slot * 88u);
		for (int i = 0; i < 22; ++i) q[i] = 0;
		q[0] = (uint32_t)o.glob_id;
		q[1] = (uint32_t)(o.glob_id >> 32);
		q[2] = (uint32_t)o.tcount[0]; // nqrys_5s_ = histstat_[n5].tcount_
		q[3] = (uint32_t)o.tsum[0];   // total_resp_5sec_ = histstat_[n5].tsum_
		q[4] = o.nconn_active;        // nconns_ (not below the active count)
		q[5] = o.nconn_active;        // nconns_active_ = last_chk_nconn_active_
		q[7] = (uint32_t)o.p95_ms[0]; // p95_5s_resp_ms_
		q[8] = (uint32_t)o.p95_ms[1]; // p95_5min_resp_ms_
		p.notify[(size_t)slot * 88u + 79u] = o.curr_qps == 0 ? 0 /* STATE_IDLE */ : 2 /* STATE_OK */;
	}
}

// ---------------------------------------------------------------------------------------------------- the listener's state decision
// TCP_LISTENER::get_curr_state (common/gy_socket_stat.cc:2020-2870): one thread per listener walks the reference's decision tree on the scan
// record (k_listener_scan) and the caller's inputs.  The tree is kept in the reference's order -- a listener leaves at the first rule that
// applies, and `why` names that rule by the reference's line -- with its C arithmetic: `ser_errors * 2` is a 32-bit product, a product
// with 1.1f is a float product when the other factor is an integer and a double product when it is a mean.
struct ListenerDecideP {
	const gys_listener_scan *scan;
	const gys_listener_issue_in *in; // nullptr: defaults
	uint8_t *hist;                   // [nsvc][2]: issue_bit_hist_, high_resp_bit_hist_
	uint8_t *notify;                 // [nsvc * 88] or nullptr
	gys_listener_decision *out;      // or nullptr
	uint32_t nsvc;
	uint32_t msec1_bucket;           // get_bucketid_from_threshold<RESP_TIME_HASH>(1) (:2062)
};

struct LDecision {
	uint32_t state, issue, why;
};

__device__ __forceinline__ LDecision ld(uint32_t st, uint32_t is, uint32_t why) { return LDecision{st, is, why}; }

__device__ LDecision listener_curr_state(const gys_listener_scan &sc, const gys_listener_issue_in &in, uint32_t msec1_bucket, uint8_t *high_hist)
{
	enum { IDLE = 0, GOOD = 1, OK = 2, BAD = 3, SEVERE = 4 };
	enum { NONE = 0, TASKS = 1, QPS_HIGH = 2, ACT_HIGH = 3, SER_ERR = 4, DEPENDS = 7, UNKNOWN = 8 };
	const bool task_issue = in.flags & GYS_LI_TASK_ISSUE, severe = in.flags & GYS_LI_SEVERE, delay = in.flags & GYS_LI_DELAY;
	const bool cpu_issue = in.flags & GYS_LI_CPU_ISSUE, mem_issue = in.flags & GYS_LI_MEM_ISSUE;
	const uint32_t errs = in.ser_errors;
	const uint64_t nq = (uint64_t)sc.tcount[0];                  // nqrys_5s (size_t)
	const uint64_t resp_msec = (uint64_t)sc.tsum[0];             // total_resp_msec
	const uint64_t tdelay = in.tasks_delay_msec;
	const int64_t p95_5 = sc.p95_ms[0], p95_5d = sc.p95_ms[2], qps = sc.curr_qps, q25 = sc.qps_p25, q95 = sc.qps_p95, a25 = sc.act_p25, a95 = sc.act_p95;
	const int64_t active = sc.nconn_active, nconn = in.nconn;
	const uint32_t b5 = sc.b5, b300 = sc.b300, b5d = sc.b5day;
	const bool much_higher = b5 > b5d + 2u && b5 > b300;         // (b5 > b5day + 2) && (b5 > b300)
	const bool many_errs = (uint64_t)(uint32_t)(errs * 2u) > nq; // ser_errors * 2 > nqrys_5s: the product is 32 bits wide
	const bool some_errs = (uint64_t)(uint32_t)(errs * 5u) > nq;
	const uint32_t err_or_none = errs ? SER_ERR : NONE;
	double m[4];
	for (int i = 0; i < 4; ++i) m[i] = (double)sc.tsum[i] / (double)(sc.tcount[i] != 0 ? sc.tcount[i] : 1); // mean_val_
	uint8_t hh = (uint8_t)(*high_hist << 1);                     // :2113
	*high_hist = hh;

	if (qps == 0 && (!task_issue || !severe || !errs)) return ld(IDLE, NONE, 2126);
	if (b5 == msec1_bucket || p95_5 < p95_5d) {                  // :2132 the response is fast, or faster than usual
		if (qps <= q25 && q25 < q95) {                       // ... and there are few queries
			if (!task_issue) {
				if (!errs) return ld(IDLE, NONE, 2144);
				if (many_errs) return ld(SEVERE, SER_ERR, 2153);
				if (some_errs) return ld(BAD, SER_ERR, 2161);
				if ((double)errs < (double)nq * 0.1) return ld(OK, SER_ERR, 2169);
			} else {
				if (many_errs) return ld(SEVERE, SER_ERR, 2179);
				if (some_errs) return ld(BAD, SER_ERR, 2187);
				if (errs) return ld(BAD, TASKS, 2202);
				if (severe && in.ntasks_issue > 0 && in.ntasks_noissue == 0) return ld(BAD, TASKS, 2213);
				if (nconn > a25) return ld(OK, TASKS, 2224);
			}
		}
		if (errs && many_errs) return ld(SEVERE, SER_ERR, 2243);
		if (errs && some_errs) return ld(BAD, SER_ERR, 2257);
		if (task_issue && severe && in.ntasks_issue > 0 && in.ntasks_noissue == 0) return ld(BAD, TASKS, 2273);
		if (errs) return ld(OK, SER_ERR, 2305);
		if (qps <= q95 || b5 + 2u <= b5d) return ld(GOOD, NONE, 2305);
		return ld(OK, QPS_HIGH, 2305);
	}
	if (p95_5 == p95_5d) {                                       // :2308
		if (errs && many_errs) return ld(SEVERE, SER_ERR, 2322);
		if (errs && some_errs) return ld(BAD, SER_ERR, 2336);
		if (m[0] <= m[2] * (double)0.8f) {
			if (qps <= q25) {
				if (errs) return ld(BAD, SER_ERR, 2356);
				if (!task_issue) return ld(IDLE, NONE, 2364);
				if (in.ntasks_issue > 0 && in.ntasks_noissue == 0) return ld(BAD, TASKS, 2374);
				if (in.ntasks_issue > 0 && tdelay >= 1000) return ld(BAD, TASKS, 2384);
			}
			if (!task_issue && !errs) return ld(GOOD, NONE, 2394);
			if (errs && task_issue) return ld(BAD, TASKS, 2403);
			return ld(OK, TASKS, 2417);                  // (:2406-2410 is overwritten by :2412-2413)
		}
		if (m[0] <= m[2] * (double)1.2f) return ld(OK, NONE, 2427);
	}
	hh |= 1u;                                                    // :2431 the response is higher than usual
	*high_hist = hh;
	if (errs && many_errs) return ld(SEVERE, SER_ERR, 2447);
	if (errs && some_errs) return ld(BAD, SER_ERR, 2461);
	if (qps > q95 && qps - q95 > 5 && (float)(int32_t)qps > (float)q95 * 1.1f) return ld(much_higher ? SEVERE : BAD, QPS_HIGH, 2492);
	if (task_issue || (delay && (int)in.ntasks_issue + (int)in.ntasks_noissue > 2 && tdelay * 4u > resp_msec)) return ld(much_higher ? SEVERE : BAD, TASKS, 2525);
	if (active > a95 && active - a95 > 1) return ld(much_higher && active > 10 ? SEVERE : BAD, ACT_HIGH, 2552);
	if (p95_5 == p95_5d && sc.p99_ms[0] > sc.p99_ms[2]) return ld(OK, err_or_none, 2571);
	if (qps <= q25 && nconn <= a25) {                            // :2576
		if (delay && cpu_issue && mem_issue) return ld(BAD, TASKS, 2593);
		if (delay && (cpu_issue || mem_issue) && tdelay * 4u > resp_msec) return ld(BAD, TASKS, 2611);
		return ld(OK, err_or_none, 2630);
	}
	{
		const int64_t span = in.tdiff_start > 0 && in.tdiff_start < 432000 ? in.tdiff_start : 432000; // sec_dist_arr[n5days] (:2064-2071)
		const int avg5d = (int)(sc.tcount[2] / span);
		if (avg5d < ((int)qps >> 1) && p95_5 <= (int64_t)sc.p95_ms[3] && m[0] <= m[3] * (double)1.1f) return ld(OK, err_or_none, 2657);
	}
	if (qps <= q25 && active <= a25 && b5 <= b5d + 1u) return ld(OK, err_or_none, 2679);
	if (b5 <= b5d + 1u && b300 == b5d && m[0] > m[1] && m[1] < m[2] * (double)1.1f) return ld(OK, err_or_none, 2702);
	if (active >= 15 && b5 == b5d + 1u) {                        // :2710 only a few connections sit in the slow buckets
		uint32_t b = b5;
		while (b < 15u && sc.nactive_conn_arr[b] <= 3u) ++b;
		if (b > b5) return ld(OK, err_or_none, 2738);
	}
	if (__popc((uint32_t)hh) < 5) return ld(OK, err_or_none, 2768);
	const uint32_t st = much_higher ? SEVERE : BAD;              // :2774 nothing explains it
	if (tdelay * 4u > resp_msec && st == BAD) return ld(BAD, TASKS, 2817);
	if (in.flags & GYS_LI_DEPENDS) return ld(st, DEPENDS, 2866);
	if (tdelay * 10u > resp_msec) return ld(st, TASKS, 2853);
	return ld(st, errs ? SER_ERR : UNKNOWN, 2866);
}

__global__ __launch_bounds__(256) void k_listener_decide(ListenerDecideP p)
{
	const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= p.nsvc) return;
	const gys_listener_scan sc = p.scan[slot];
	gys_listener_issue_in in{};
	if (p.in) in = p.in[slot];
	else in.nconn = sc.nconn_active; // (the scan's notify record reports the active count as nconns_ too)
	uint8_t ih = p.hist[2u * slot], hh = p.hist[2u * slot + 1u];
	LDecision d = listener_curr_state(sc, in, p.msec1_bucket, &hh);
	if (!(in.flags & GYS_LI_YOUNG) || in.ser_errors) { // diffstartusec > 100 s || ser_errors (:4244)
		ih = (uint8_t)(ih << 1);
		if (d.state >= 3u) ih |= 1u;
	} else {                                           // "Listener Just recently started" (:4255-4262)
		ih = 0;
		d = LDecision{2u, 0u, 4262u};
	}
	p.hist[2u * slot] = ih;
	p.hist[2u * slot + 1u] = hh;
	if (p.out) {
		gys_listener_decision o{};
		o.state = (uint8_t)d.state;
		o.issue = (uint8_t)d.issue;
		o.issue_bit_hist = ih;
		o.high_resp_bit_hist = hh;
		o.decided_line = (uint16_t)d.why;
		p.out[slot] = o;
	}
	if (p.notify) { // comm::LISTENER_STATE_NOTIFY (common/gy_comm_proto.h:2183-2254; filled at common/gy_socket_stat.cc:4293-4330)
		uint8_t *r = p.notify + (size_t)slot * 88u;
		uint32_t *q = (uint32_t *)r;
		q[11] = in.ser_errors;                     // ser_errors_ @44
		q[13] = in.tasks_delay_msec * 1000u;       // tasks_delay_usec_ @52
		q[14] = in.tasks_cpudelay_msec * 1000u;    // tasks_cpudelay_usec_ @56
		q[15] = in.tasks_blkiodelay_msec * 1000u;  // tasks_blkiodelay_usec_ @60
		*(uint16_t *)(r + 76) = in.ntasks_issue;   // ntasks_issue_ @76
		if (p.in) q[4] = (uint32_t)in.nconn;       // nconns_ @16 = last_chk_nconn_
		r[79] = (uint8_t)d.state;
		r[80] = (uint8_t)d.issue;
		r[81] = ih;
		r[82] = hh;
	}
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

// ---------------------------------------------------------------------------------------------------- ACTIVE_CONN_STATS ingest
// comm::ACTIVE_CONN_STATS (common/gy_comm_proto.h:2766-2783), 104-byte fixed stride: listener_glob_id_@0 cli_aggr_task_id_@8 ser_comm_@16
// cli_comm_@32 remote_machine_id_@48 remote_madhava_id_@64 bytes_sent_@72 bytes_received_@80 cli_delay_msec_@88 ser_delay_msec_@92
// max_rtt_msec_@96 active_conns_@100 flags@102 (bit 1 = is_remote_listen_); layout pinned against the reference's compiler in
// tests/test_wire.py.  The reference turns every row into an SQL insert (insert_active_conns, server/gy_mconnhdlr.cc:7776-7960: local
// listeners -> activeconntbl, remote listeners -> remoteconntbl); here the local-listener rows feed a Count-Min PAIR keyed by the 4 words
// (listener id, client task group id) -- the per-(listener, client task) roll-up that replaces those rows and connlistenmap_ /
// connclientmap_ (server/gy_msocket.h:240-290) -- and exact per-listener sums.  One thread per row.
struct ActConnP {
	const uint8_t *batch;
	uint32_t n;
	DevTable gid;
	uint32_t *pair32;            // arena: Count-Min of active connections per pair: [D*W] rows of LOCAL listeners, then [D*W] rows whose listener lives on another madhava (is_remote_listen_)
	unsigned long long *pair64;  // arena: Count-Min of bytes (sent + received) per pair, same two tables
	unsigned long long *svc_act; // [nsvc*4] cumulative per listener: rows, bytes_sent, bytes_received, active connections
	uint64_t *counters;
	uint32_t *win_rows;          // arena (u32 SUM section): local-listener rows of this window, all ranks after the exchange
};

__global__ __launch_bounds__(256) void k_actconn_ingest(ActConnP p)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const bool in = i < p.n;
	uint64_t w[13];
	if (in) {
		const uint64_t *q = (const uint64_t *)(p.batch + (size_t)i * 104u);
#pragma unroll
		for (int k = 0; k < 13; ++k) w[k] = q[k];
	}
	const bool remote = in && ((w[12] >> 49) & 1ull); // flags byte @102 = bits 48..55 of word 12, is_remote_listen_ = bit 1
	wave_count(&p.counters[CTR_ACTCONN_RECORDS], in && !remote);
	{
		const unsigned long long b = __ballot(in); // (rows of either kind keep the ring of the last windows' tables alive, k_act_latch)
		if (b && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)b) - 1u) atomicAdd(p.win_rows, (uint32_t)__popcll(b));
	}
	wave_count(&p.counters[CTR_ACTCONN_REMOTE_LISTEN], remote);
	if (!in) return;
	const uint64_t gid = w[0], task = w[1], sent = w[9], rcvd = w[10];
	const uint32_t act = (uint32_t)((w[12] >> 32) & 0xFFFFu); // active_conns_ @100
	// rows of a listener on another madhava (insert_active_conns: -> remoteconntbl, server/gy_mconnhdlr.cc:7888-7925): the same roll-up into
	// tables of their own (behind the local ones) -- the client side's view of (remote listener, client task group)
	const uint32_t tb = remote ? GYS_CMS_D * GYS_CMS_W : 0u;
#pragma unroll
	for (uint32_t r = 0; r < GYS_CMS_D; ++r) {
		const uint32_t col = jhash2_4w((uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)task, (uint32_t)(task >> 32), GYS_SEED + r) & (GYS_CMS_W - 1);
		if (act) atomicAdd(&p.pair32[tb + r * GYS_CMS_W + col], act);
		if (sent + rcvd) atomicAdd(&p.pair64[tb + r * GYS_CMS_W + col], (unsigned long long)(sent + rcvd));
	}
	if (remote) return; // (the listener is not one of this engine's services)
	const uint32_t slot = tbl_lookup(p.gid, gid);
	if (slot == GYS_NOSLOT) {
		atomicAdd((unsigned long long *)&p.counters[CTR_ACTCONN_UNKNOWN], 1ull);
		return;
	}
	unsigned long long *a = p.svc_act + (size_t)slot * 4;
	atomicAdd(&a[0], 1ull);
	atomicAdd(&a[1], (unsigned long long)sent);
	atomicAdd(&a[2], (unsigned long long)rcvd);
	atomicAdd(&a[3], (unsigned long long)act);
}

// A partha reports its ACTIVE_CONN_STATS every 15 s (server/gy_mconnhdlr.cc:7714: a 4-s tolerance on a 15-s cadence) on a phase of its own, a
// window is 5 s: one window only carries the parthas that reported in it.  The (all-rank) tables of the last GYS_ACT_RING = 3 windows are
// kept in a ring and the queries read their PER-CELL MAXIMUM: a pair reported anywhere in the last 15 s reads back at least its reported
// value (the Count-Min guarantee "never below" holds across the report period; a partha whose report lands in two of the three windows is
// not counted twice, which a per-cell sum would do), and a pair not reported for three windows ages out -- the gauge's "last report".
// `live[w & 1]` = windows until the ring is all zero again: an engine that never sees such rows pays one 4-byte read per window.
#define GYS_ACT_RING 3u
__global__ __launch_bounds__(256) void k_act_latch(const uint32_t *win_rows, const uint32_t *d_epoch, uint32_t *live, const uint32_t *pair32,
						   const unsigned long long *pair64, uint32_t *ring32, unsigned long long *ring64, uint32_t *last32,
						   unsigned long long *last64)
{
	const uint32_t w = *d_epoch, cur = live[w & 1u], rows = *win_rows;
	if (blockIdx.x == 0 && threadIdx.x == 0) live[(w + 1u) & 1u] = rows ? GYS_ACT_RING : (cur ? cur - 1u : 0u); // (read by the NEXT window's launch)
	if (rows == 0u && cur == 0u) return;
	const uint32_t n = 2u * GYS_CMS_D * GYS_CMS_W, me = w % GYS_ACT_RING; // (the local-listener tables and the remote-listener ones behind them)
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t v32 = pair32[i];
		const unsigned long long v64 = pair64[i];
		ring32[(size_t)me * n + i] = v32;
		ring64[(size_t)me * n + i] = v64;
		uint32_t m32 = v32;
		unsigned long long m64 = v64;
#pragma unroll
		for (uint32_t k = 0; k < GYS_ACT_RING; ++k) {
			if (k == me) continue;
			const uint32_t a = ring32[(size_t)k * n + i];
			const unsigned long long b = ring64[(size_t)k * n + i];
			m32 = a > m32 ? a : m32;
			m64 = b > m64 ? b : m64;
		}
		last32[i] = m32;
		last64[i] = m64;
	}
}

// ---------------------------------------------------------------------------------------------------- top-N of every host at once
// The reference keeps four bounded min-heaps of 10 per partha (LISTEN_TOP_ISSUE / _QPS / _ACTIVE_CONN / _NET, server/gy_mconnhdlr.h:961,
// comparators LISTEN_TOPN server/gy_msocket.h:720-796), filled while partha_listener_state walks the records
// (server/gy_mconnhdlr.cc:11175-11304); web_curr_top_listeners merges the per-host queues into 50 slots for a multi-host query
// (server/gy_mnodehandle.cc:2885-3040).  Here: one workgroup per host selects the 10 largest of its services' last-window records for one
// kind -- every thread keeps the 10 best of its strided share in LDS, then 10 rounds of block arg-max over the 2 560 candidates.
// Order: metric descending, ties by lower service slot (deterministic where the heap's order is arbitrary).
struct TopnHostsP {
	const uint8_t *svc_state;
	const uint32_t *off, *members; // services of host h: members[off[h] .. off[h + 1])
	uint32_t nhosts, epoch;
	int kind;
	uint32_t *out_slot;   // [nhosts * GYS_TOPN], GYS_NOSLOT = none
	uint64_t *out_metric; // [nhosts * GYS_TOPN]
};

__device__ __forceinline__ bool topn_metric_of(const uint8_t *svc_state, uint32_t s, uint32_t host, uint32_t epoch, int kind, uint64_t *metric)
{
	const uint64_t *q = (const uint64_t *)(svc_state + (size_t)s * 96);
	const uint64_t tag = q[11];
	if ((uint32_t)tag != epoch || (uint32_t)(tag >> 32) != host) return false;
	const uint32_t nqrys = (uint32_t)q[1], nactive = (uint32_t)(q[2] >> 32), kbin = (uint32_t)(q[4] >> 32), kbout = (uint32_t)q[5];
	const uint32_t delay = (uint32_t)(q[6] >> 32); // tasks_delay_usec_ @52
	const uint32_t state = (uint32_t)((q[9] >> 56) & 0xFF);
	switch (kind) {
	case 0: *metric = ((uint64_t)state << 32) | delay; return state > 2u; // is_issue: curr_state_ > STATE_OK; (state, tasks_delay) order
	case 1: *metric = nqrys; return nqrys >= 5u;
	case 2: *metric = nactive; return nactive >= 1u;
	default: *metric = (uint64_t)kbin + kbout; return (kbin + kbout) > 0u;
	}
}

// a before b in the top-N order
__device__ __forceinline__ bool topn_before(uint64_t ma, uint32_t sa, uint64_t mb, uint32_t sb) { return ma != mb ? ma > mb : sa < sb; }

__global__ __launch_bounds__(256) void k_topn_hosts(TopnHostsP p)
{
	__shared__ uint64_t s_m[256 * GYS_TOPN];
	__shared__ uint32_t s_s[256 * GYS_TOPN];
	__shared__ uint64_t s_rm[4];
	__shared__ uint32_t s_rs[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	for (uint32_t h = blockIdx.x; h < p.nhosts; h += gridDim.x) {
		uint64_t *mym = s_m + tid * GYS_TOPN;
		uint32_t *mys = s_s + tid * GYS_TOPN;
		for (uint32_t k = 0; k < GYS_TOPN; ++k) mys[k] = GYS_NOSLOT;
		uint32_t n = 0;
		for (uint32_t i = p.off[h] + tid; i < p.off[h + 1]; i += 256u) {
			const uint32_t s = p.members[i];
			uint64_t m;
			if (!topn_metric_of(p.svc_state, s, h, p.epoch, p.kind, &m)) continue;
			if (n == GYS_TOPN && !topn_before(m, s, mym[n - 1], mys[n - 1])) continue;
			uint32_t k = n < GYS_TOPN ? n : GYS_TOPN - 1u; // insertion into the thread's sorted list
			while (k > 0 && topn_before(m, s, mym[k - 1], mys[k - 1])) {
				mym[k] = mym[k - 1];
				mys[k] = mys[k - 1];
				--k;
			}
			mym[k] = m;
			mys[k] = s;
			if (n < GYS_TOPN) ++n;
		}
		__syncthreads();
		uint64_t pm = ~0ull; // the previously selected entry: every round takes the best entry strictly after it
		uint32_t ps = 0;
		bool first = true;
		for (uint32_t r = 0; r < GYS_TOPN; ++r) {
			uint64_t bm = 0;
			uint32_t bs = GYS_NOSLOT;
			for (uint32_t e = tid; e < 256u * GYS_TOPN; e += 256u) {
				const uint32_t s = s_s[e];
				if (s == GYS_NOSLOT) continue;
				const uint64_t m = s_m[e];
				if (!first && !topn_before(pm, ps, m, s)) continue;
				if (bs == GYS_NOSLOT || topn_before(m, s, bm, bs)) {
					bm = m;
					bs = s;
				}
			}
#pragma unroll
			for (int d = 32; d >= 1; d >>= 1) {
				const uint64_t om = __shfl_xor(bm, d, 64);
				const uint32_t os = (uint32_t)__shfl_xor((int)bs, d, 64);
				if (os != GYS_NOSLOT && (bs == GYS_NOSLOT || topn_before(om, os, bm, bs))) {
					bm = om;
					bs = os;
				}
			}
			if (lane == 0) {
				s_rm[wave] = bm;
				s_rs[wave] = bs;
			}
			__syncthreads();
			bm = s_rm[0];
			bs = s_rs[0];
			for (uint32_t w = 1; w < 4u; ++w)
				if (s_rs[w] != GYS_NOSLOT && (bs == GYS_NOSLOT || topn_before(s_rm[w], s_rs[w], bm, bs))) {
					bm = s_rm[w];
					bs = s_rs[w];
				}
			if (tid == 0) {
				p.out_slot[(size_t)h * GYS_TOPN + r] = bs;
				p.out_metric[(size_t)h * GYS_TOPN + r] = bm;
			}
			pm = bm;
			ps = bs;
			first = false;
			__syncthreads();
			if (bs == GYS_NOSLOT) { // fewer than N qualify: the remaining places stay empty
				for (uint32_t r2 = r + 1 + tid; r2 < GYS_TOPN; r2 += 256u) p.out_slot[(size_t)h * GYS_TOPN + r2] = GYS_NOSLOT;
				break;
			}
		}
		__syncthreads();
	}
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

// PMC calibration (tools/calibrate_fetch.py): reads nevents 24-byte events with EXACTLY the access pattern of k_resp_host's event phase
// (thread t of a 1024-thread workgroup: 3 x 8-byte loads at 24 (tile + u * 1024 + t), 4 events per thread at a time) and nothing else,
// so that rocprofv3's FETCH_SIZE for this kernel can be set against the known 24 B x nevents.
__global__ __launch_bounds__(1024) void k_read_events(const uint64_t *ev, uint64_t n, uint64_t per_wg, uint64_t *out)
{
	const uint64_t e0 = (uint64_t)blockIdx.x * per_wg, e1 = min(n, e0 + per_wg);
	uint64_t acc = 0;
	for (uint64_t t0 = e0; t0 < e1; t0 += 4096u) {
		uint64_t w0[4], w1[4], w2[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint64_t i = t0 + threadIdx.x + (uint64_t)u * 1024u;
			w0[u] = 0; w1[u] = 0; w2[u] = 0;
			if (i < e1) {
				w0[u] = ev[3 * i];
				w1[u] = ev[3 * i + 1];
				w2[u] = ev[3 * i + 2];
			}
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) acc ^= w0[u] + 3 * w1[u] + 5 * w2[u];
	}
	if (acc == 0x123456789ABCDEFull) out[0] = acc; // (never: keeps the loads alive)
}

} // namespace gys

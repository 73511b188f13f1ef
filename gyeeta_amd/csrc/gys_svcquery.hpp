// gys_svcquery.hpp -- QUERY_OPTIONS on the live listener table: multi-host filter, sort, maxrecs and the aggregation operators.
//
// The reference answers "svcstate of all hosts where qps5s > X and p95resp5s > Y" by walking every MTCP_LISTENER of every partha under RCU,
// building a SvcStateFields per listener and running CRITERIA_SET::match_criteria on it (MCONN_HANDLER::web_curr_listener_state,
// server/gy_mnodehandle.cc:4650-4900: per row `filter_match`, `nrecs >= maxrecs`, the multihost walk over partha_tbl_; criteria:
// common/gy_query_criteria.h:1243-1290 match_num_criterian, :1535-1605 match_criteria_group, :1806-1900 match_criteria; the numeric columns:
// SvcStateFields::get_num_field server/gy_mfields.h:1402-1440, every one an `int`).  Here the 96-byte kept state records of ALL services
// (svc_state: the 88-byte LISTENER_STATE_NOTIFY + {window, host}) are scanned in one pass:
//   k_svc_filter  one thread per record: currency check (state at most one window old, gys_json_svcstate's rule), the criteria groups, the
//                 sort key; matching records are compacted into a candidate list {key, slot} (one list-cursor atomic per workgroup)
//   k_svc_hist / k_svc_pick   when more records match than `maxrecs`: exact top-K by radix selection on the 64-bit key, 11 bits per
//                 round, the bin chosen on the device (no host round trip between rounds)
//   k_svc_gather  the selected candidates' rows {slot, host, 88-byte record} + keys, compacted for the host to order
//   k_svc_aggr    AGGR_OPER_E (common/gy_json_field_maps.h:114-129: sum / avg / max / min / count / bool_or / bool_and) of chosen columns
//                 over the matching records, grouped by nothing / host / cluster: per-workgroup LDS accumulation keyed by group, one set
//                 of device atomics per distinct group of the workgroup
// Key: (column value biased so that "first" = largest, then the LOWER slot first) -- unique per record, so the selection and the order
// are deterministic where the reference's hash-table walk order is arbitrary.
#pragma once

namespace gys {

// numeric columns of json_db_svcstate_arr (common/gy_json_field_maps.h:1102-1135) in their order there; GYS_SVC_COL_* of gysketch.h
enum { SVC_COL_QPS5S = 0, SVC_COL_NQRY5S, SVC_COL_RESP5S, SVC_COL_P95RESP5S, SVC_COL_P95RESP5M, SVC_COL_NCONNS, SVC_COL_NACTIVE, SVC_COL_NPROCS,
       SVC_COL_KBIN15S, SVC_COL_KBOUT15S, SVC_COL_SERERR, SVC_COL_CLIERR, SVC_COL_DELAYUS, SVC_COL_CPUDELUS, SVC_COL_IODELUS, SVC_COL_VMDELUS,
       SVC_COL_USERCPU, SVC_COL_SYSCPU, SVC_COL_RSSMB, SVC_COL_NISSUE, SVC_COL_STATE, SVC_COL_ISSUE, SVC_COL_ISHTTP, SVC_NCOLS };
// COMPARATORS_E common/gy_query_criteria.h:28-46 (the numeric ones)
enum { SVC_COMP_EQ = 0, SVC_COMP_NEQ, SVC_COMP_LT, SVC_COMP_LE, SVC_COMP_GT, SVC_COMP_GE, SVC_COMP_BIT2, SVC_COMP_BIT3, SVC_COMP_IN = 12, SVC_COMP_NOTIN = 13 };

#define GYS_SVCQ_MAX_TERMS 16u
#define GYS_SVCQ_MAX_GROUPS 8u
#define GYS_SVCQ_MAX_AGGR 8u

struct SvcTerm {
	uint8_t col, comp, group, pad;
	uint32_t nvalues, set_first; // IN / NOTIN: set_values[set_first .. set_first + nvalues)
	int32_t value;               // the criterion converted to the column's type (`crit` of match_num_criterian<int>)
};

struct SvcFilterP {
	const uint8_t *svc_state; // [nsvc * 96]
	const uint32_t *svc_host; // [nsvc]
	const uint64_t *svc_gid;  // [nsvc]
	uint32_t nsvc, epoch;
	const uint32_t *host_mask; // bit h set = host h is part of the query, or nullptr = every host (is_multihost_)
	const uint32_t *slot_list; // the query names its listeners (svcid = / in: the reference's direct-lookup path :4754-4860): their slots, ascending;
	uint32_t nitems;           // ... nitems of them -- or nullptr and nitems = nsvc: every slot
	const int32_t *set_values;
	uint32_t nterms, ngroups;
	SvcTerm terms[GYS_SVCQ_MAX_TERMS];
	uint8_t group_oper[GYS_SVCQ_MAX_GROUPS]; // 0 = AND, 1 = OR inside the group (CRITERIA_ONE_GROUP::oper_)
	uint32_t top_oper;                       // 0 = AND, 1 = OR between the groups (CRITERIA_SET::l1_oper_)
	int32_t sort_col;                        // -1: slot order
	uint32_t sort_desc;
	// outputs
	unsigned long long *cand_key; // [nsvc]
	uint32_t *cand_slot;          // [nsvc]
	uint32_t *cursor;             // [0] candidates so far
};

// SvcStateFields::get_num_field (server/gy_mfields.h:1402-1440): every column is an `int` built from the u32 / u16 / u8 wire field;
// w = the record's first 22 32-bit words (88 bytes)
__device__ __forceinline__ int32_t svc_col_value(const uint32_t *w, uint32_t col)
{
	const uint32_t nq = w[2];
	switch (col) {
	case SVC_COL_QPS5S: return (int32_t)(nq / 5u);
	case SVC_COL_NQRY5S: return (int32_t)nq;
	case SVC_COL_RESP5S: return (int32_t)(w[3] / (nq ? nq : 1u));
	case SVC_COL_P95RESP5S: return (int32_t)w[7];
	case SVC_COL_P95RESP5M: return (int32_t)w[8];
	case SVC_COL_NCONNS: return (int32_t)w[4];
	case SVC_COL_NACTIVE: return (int32_t)w[5];
	case SVC_COL_NPROCS: return (int32_t)w[6];
	case SVC_COL_KBIN15S: return (int32_t)w[9];
	case SVC_COL_KBOUT15S: return (int32_t)w[10];
	case SVC_COL_SERERR: return (int32_t)w[11];
	case SVC_COL_CLIERR: return (int32_t)w[12];
	case SVC_COL_DELAYUS: return (int32_t)w[13];
	case SVC_COL_CPUDELUS: return (int32_t)w[14];
	case SVC_COL_IODELUS: return (int32_t)w[15];
	case SVC_COL_VMDELUS: return (int32_t)(w[13] - w[14] - w[15]); // unsigned arithmetic, then int (:1429)
	case SVC_COL_USERCPU: return (int32_t)w[16];
	case SVC_COL_SYSCPU: return (int32_t)w[17];
	case SVC_COL_RSSMB: return (int32_t)w[18];
	case SVC_COL_NISSUE: return (int32_t)(w[19] & 0xFFFFu);        // ntasks_issue_ @76 (u16)
	case SVC_COL_ISHTTP: return (int32_t)((w[19] >> 16) & 0xFFu) != 0; // is_http_svc_ @78
	case SVC_COL_STATE: return (int32_t)(w[19] >> 24);              // curr_state_ @79 (filters name it through statefromjson)
	case SVC_COL_ISSUE: return (int32_t)(int16_t)(w[20] & 0xFFu);   // curr_issue_ @80: int16_t(u8) (:1435)
	default: return 0;
	}
}

__device__ __forceinline__ bool svc_term_match(const SvcTerm &t, int32_t v, const int32_t *set_values)
{
	switch (t.comp) { // match_num_criterian<int> common/gy_query_criteria.h:1243-1290
	case SVC_COMP_EQ: return v == t.value;
	case SVC_COMP_NEQ: return v != t.value;
	case SVC_COMP_LT: return v < t.value;
	case SVC_COMP_LE: return v <= t.value;
	case SVC_COMP_GT: return v > t.value;
	case SVC_COMP_GE: return v >= t.value;
	case SVC_COMP_BIT2: return (v & 3) == 3;
	case SVC_COMP_BIT3: return (v & 7) == 7;
	case SVC_COMP_IN:
	case SVC_COMP_NOTIN: {
		const bool bret = t.comp == SVC_COMP_IN; // (an empty value list: IN matches nothing, NOTIN everything)
		for (uint32_t i = 0; i < t.nvalues; ++i)
			if (set_values[t.set_first + i] == v) return bret;
		return !bret;
	}
	default: return false;
	}
}

// CRITERIA_SET::match_criteria (:1806-1900) over CRITERIA_ONE_GROUP::match_criteria_group (:1535-1605) for criteria that are all of this
// subsystem: a group passes when its operator is OR and a term matches, or AND and every term matches; the groups combine with top_oper.
// No criteria at all = CRIT_SKIP = the record is listed.
template <typename P>
__device__ __forceinline__ bool svc_filter_match(const P &p, const uint32_t *w)
{
	if (p.nterms == 0) return true;
	uint32_t pass = 0, fail = 0; // bit g: group g has passed / failed
	uint32_t seen = 0;
	for (uint32_t i = 0; i < p.nterms; ++i) {
		const SvcTerm &t = p.terms[i];
		const uint32_t g = t.group, bit = 1u << g;
		seen |= bit;
		const bool m = svc_term_match(t, svc_col_value(w, t.col), p.set_values);
		if (p.group_oper[g]) {
			if (m) pass |= bit;
		} else {
			if (!m) fail |= bit;
		}
	}
	uint32_t gpass = 0;
	for (uint32_t g = 0; g < p.ngroups; ++g) {
		const uint32_t bit = 1u << g;
		if (!(seen & bit)) continue;
		const bool ok = p.group_oper[g] ? (pass & bit) != 0 : (fail & bit) == 0;
		if (ok) gpass |= bit;
	}
	return p.top_oper ? gpass != 0 : gpass == seen;
}

// the record of `slot` if it is current: 24 words (96 bytes) into w; false when the listener has no state of this or the last window, the
// state was deleted, or the record is another listener's (the reference lists listeners whose state is at most 10 s old, :4660)
template <typename P>
__device__ __forceinline__ bool svc_load_current(const P &p, uint32_t slot, uint32_t *w, uint32_t *host_out)
{
	const uint4 *q = (const uint4 *)(p.svc_state + (size_t)slot * 96);
#pragma unroll
	for (int k = 0; k < 6; ++k) {
		const uint4 v = q[k];
		w[4 * k] = v.x;
		w[4 * k + 1] = v.y;
		w[4 * k + 2] = v.z;
		w[4 * k + 3] = v.w;
	}
	const uint32_t ep = w[22], host = w[23];
	*host_out = host;
	if (ep == 0u || ep + 1u < p.epoch) return false;
	if (host != p.svc_host[slot]) return false;
	const uint64_t gid = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
	if (gid != p.svc_gid[slot]) return false;
	if (p.host_mask && !((p.host_mask[host >> 5] >> (host & 31u)) & 1u)) return false;
	return true;
}

#define GYS_SVCQ_THREADS 256u
#define GYS_SVCQ_PER_THREAD 4u

__global__ __launch_bounds__(GYS_SVCQ_THREADS) void k_svc_filter(SvcFilterP p)
{
	__shared__ uint32_t s_wave[GYS_SVCQ_THREADS / 64u];
	__shared__ uint32_t s_base;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t first = blockIdx.x * (GYS_SVCQ_THREADS * GYS_SVCQ_PER_THREAD);
	unsigned long long keys[GYS_SVCQ_PER_THREAD];
	uint32_t mine = 0; // bit k: record k of this thread matched
	uint32_t slots[GYS_SVCQ_PER_THREAD];
#pragma unroll
	for (uint32_t k = 0; k < GYS_SVCQ_PER_THREAD; ++k) {
		const uint32_t item = first + k * GYS_SVCQ_THREADS + threadIdx.x;
		keys[k] = 0;
		slots[k] = 0;
		if (item < p.nitems) {
			const uint32_t slot = p.slot_list ? p.slot_list[item] : item;
			slots[k] = slot;
			uint32_t w[24], host;
			if (svc_load_current(p, slot, w, &host) && svc_filter_match(p, w)) {
				const uint32_t v = p.sort_col >= 0 ? ((uint32_t)svc_col_value(w, (uint32_t)p.sort_col) ^ 0x80000000u) : 0u;
				keys[k] = ((unsigned long long)(p.sort_desc ? v : ~v) << 32) | (unsigned long long)(0xFFFFFFFFu - slot);
				mine |= 1u << k;
			}
		}
	}
	// the workgroup's place in the candidate list: one atomic on the list cursor per workgroup
	const uint32_t cnt = (uint32_t)__popc(mine);
	uint32_t incl = cnt;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
		if ((int)lane >= d) incl += o;
	}
	if (lane == 63u) s_wave[wave] = incl;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tot = 0;
		for (uint32_t k = 0; k < GYS_SVCQ_THREADS / 64u; ++k) {
			const uint32_t c = s_wave[k];
			s_wave[k] = tot;
			tot += c;
		}
		s_base = tot ? atomicAdd(p.cursor, tot) : 0u;
	}
	__syncthreads();
	uint32_t at = s_base + s_wave[wave] + incl - cnt;
#pragma unroll
	for (uint32_t k = 0; k < GYS_SVCQ_PER_THREAD; ++k) {
		if (mine & (1u << k)) {
			p.cand_key[at] = keys[k];
			p.cand_slot[at] = slots[k];
			++at;
		}
	}
}

// ---- exact top-K by radix selection: the K-th largest key is found 11 bits per round, most significant first
#define GYS_SVCQ_RADIX_BITS 11u
#define GYS_SVCQ_RADIX (1u << GYS_SVCQ_RADIX_BITS)
#define GYS_SVCQ_ROUNDS 6u // 5 x 11 + 9 = 64 bits

struct SvcSelectP {
	const unsigned long long *cand_key;
	const uint32_t *ncand; // device: candidates
	uint32_t *hist;        // [GYS_SVCQ_RADIX]
	// selection state (device): prefix = the bits of the threshold found so far (in place), want = how many keys still to take from the
	// keys whose high bits equal the prefix
	unsigned long long *prefix;
	uint32_t *want;
	uint32_t shift, bits; // this round's digit: key bits [shift, shift + bits)  (rounds: 11 bits each from bit 53 down to bit 9, then the last 9 bits)
};

__device__ __forceinline__ unsigned long long svcq_high_mask(uint32_t shift, uint32_t bits)
{
	const uint32_t top = shift + bits;
	return top >= 64u ? 0ull : ~0ull << top;
}

__global__ __launch_bounds__(256) void k_svc_hist(SvcSelectP p)
{
	__shared__ uint32_t s_h[GYS_SVCQ_RADIX];
	for (uint32_t k = threadIdx.x; k < GYS_SVCQ_RADIX; k += blockDim.x) s_h[k] = 0;
	__syncthreads();
	const uint32_t n = *p.ncand;
	const unsigned long long hm = svcq_high_mask(p.shift, p.bits), pre = *p.prefix & hm;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const unsigned long long key = p.cand_key[i];
		if ((key & hm) == pre) atomicAdd(&s_h[(uint32_t)(key >> p.shift) & ((1u << p.bits) - 1u)], 1u);
	}
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < GYS_SVCQ_RADIX; k += blockDim.x)
		if (s_h[k]) atomicAdd(&p.hist[k], s_h[k]);
}

// one workgroup: walks the digit histogram from the largest digit down until `want` keys are covered; the digit where that happens joins
// the prefix, the keys of the larger digits are all taken (want -= their count); clears the histogram for the next round
__global__ __launch_bounds__(256) void k_svc_pick(SvcSelectP p)
{
	__shared__ uint32_t s_h[GYS_SVCQ_RADIX];
	for (uint32_t k = threadIdx.x; k < GYS_SVCQ_RADIX; k += blockDim.x) {
		s_h[k] = p.hist[k];
		p.hist[k] = 0;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t want = *p.want, d = 1u << p.bits;
		while (d > 0u) {
			--d;
			const uint32_t c = s_h[d];
			if (c >= want) break;
			want -= c;
		}
		*p.want = want; // of the keys with this digit (and the prefix above it), `want` are still to be taken
		*p.prefix = (*p.prefix & svcq_high_mask(p.shift, p.bits)) | ((unsigned long long)d << p.shift);
	}
}

struct SvcGatherP {
	const uint8_t *svc_state;
	const unsigned long long *cand_key;
	const uint32_t *cand_slot;
	const uint32_t *ncand;
	const unsigned long long *threshold; // keys >= *threshold are taken (unique keys: exactly K of them), or nullptr = all
	uint32_t maxout;
	uint32_t *out_count;
	uint8_t *out_rows; // [maxout * 96]: {slot, host, 88-byte record}
	unsigned long long *out_keys;
};

__global__ __launch_bounds__(256) void k_svc_gather(SvcGatherP p)
{
	__shared__ uint32_t s_wave[4];
	__shared__ uint32_t s_base;
	const uint32_t n = *p.ncand, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const unsigned long long thr = p.threshold ? *p.threshold : 0ull;
	for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) { // (uniform trip count per workgroup)
		const uint32_t i = i0 + threadIdx.x;
		unsigned long long key = 0;
		bool take = false;
		if (i < n) {
			key = p.cand_key[i];
			take = key >= thr;
		}
		const unsigned long long b = __ballot(take);
		const uint32_t before = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
		if (lane == 0u) s_wave[wave] = (uint32_t)__popcll(b);
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t tot = 0;
			for (uint32_t k = 0; k < 4u; ++k) {
				const uint32_t c = s_wave[k];
				s_wave[k] = tot;
				tot += c;
			}
			s_base = tot ? atomicAdd(p.out_count, tot) : 0u;
		}
		__syncthreads();
		if (take) {
			const uint32_t at = s_base + s_wave[wave] + before;
			if (at < p.maxout) {
				const uint32_t slot = p.cand_slot[i];
				const uint4 *q = (const uint4 *)(p.svc_state + (size_t)slot * 96);
				uint4 *d = (uint4 *)(p.out_rows + (size_t)at * 96);
				const uint4 r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3], r4 = q[4], r5 = q[5];
				// row = {slot, host, record bytes 0..87}: the record moves up by 8 bytes
				d[0] = make_uint4(slot, r5.w, r0.x, r0.y);
				d[1] = make_uint4(r0.z, r0.w, r1.x, r1.y);
				d[2] = make_uint4(r1.z, r1.w, r2.x, r2.y);
				d[3] = make_uint4(r2.z, r2.w, r3.x, r3.y);
				d[4] = make_uint4(r3.z, r3.w, r4.x, r4.y);
				d[5] = make_uint4(r4.z, r4.w, r5.x, r5.y);
				p.out_keys[at] = key;
			}
		}
		__syncthreads(); // (s_wave / s_base are rewritten by the next trip)
	}
}

// ---- aggregation operators over the matching records
struct SvcAggrP {
	const uint8_t *svc_state;
	const uint32_t *svc_host;
	const uint64_t *svc_gid;
	uint32_t nsvc, epoch;
	const uint32_t *host_mask;
	const uint32_t *slot_list;
	uint32_t nitems;
	const int32_t *set_values;
	uint32_t nterms, ngroups;
	SvcTerm terms[GYS_SVCQ_MAX_TERMS];
	uint8_t group_oper[GYS_SVCQ_MAX_GROUPS];
	uint32_t top_oper;
	uint32_t group_by; // 0: one group, 1: per host, 2: per cluster
	const uint32_t *host_cluster;
	uint32_t ncols;
	uint8_t cols[GYS_SVCQ_MAX_AGGR];
	// per aggregation group g (0 / host slot / cluster index): acc[g][ncols][3] = {sum, min, max} as long long, count[g]
	long long *acc;
	unsigned long long *count;
};

#define GYS_SVCA_SLOTS 256u // LDS entries per workgroup (a workgroup walks 1024 consecutive slots: few distinct hosts / clusters)

__global__ __launch_bounds__(GYS_SVCQ_THREADS) void k_svc_aggr(SvcAggrP p)
{
	__shared__ uint32_t s_key[GYS_SVCA_SLOTS];
	__shared__ unsigned long long s_cnt[GYS_SVCA_SLOTS];
	__shared__ long long s_acc[GYS_SVCA_SLOTS][GYS_SVCQ_MAX_AGGR][3];
	for (uint32_t k = threadIdx.x; k < GYS_SVCA_SLOTS; k += GYS_SVCQ_THREADS) {
		s_key[k] = 0xFFFFFFFFu;
		s_cnt[k] = 0;
		for (uint32_t a = 0; a < GYS_SVCQ_MAX_AGGR; ++a) {
			s_acc[k][a][0] = 0;
			s_acc[k][a][1] = 0x7FFFFFFFFFFFFFFFll;
			s_acc[k][a][2] = -0x7FFFFFFFFFFFFFFFll - 1ll;
		}
	}
	__syncthreads();
	const uint32_t first = blockIdx.x * (GYS_SVCQ_THREADS * GYS_SVCQ_PER_THREAD);
#pragma unroll 1
	for (uint32_t k = 0; k < GYS_SVCQ_PER_THREAD; ++k) {
		const uint32_t item = first + k * GYS_SVCQ_THREADS + threadIdx.x;
		if (item >= p.nitems) continue;
		const uint32_t slot = p.slot_list ? p.slot_list[item] : item;
		uint32_t w[24], host;
		if (!svc_load_current(p, slot, w, &host) || !svc_filter_match(p, w)) continue;
		const uint32_t grp = p.group_by == 0u ? 0u : p.group_by == 1u ? host : p.host_cluster[host];
		// the workgroup's LDS entry of the group (open addressing; an entry that cannot be placed -- more distinct groups than entries in
		// one workgroup's 1024 slots -- goes straight to the device accumulators)
		uint32_t h = (grp * 0x9E3779B1u) >> 24, tries = 0;
		bool placed = false;
		for (; tries < GYS_SVCA_SLOTS; ++tries) {
			const uint32_t prev = atomicCAS(&s_key[h], 0xFFFFFFFFu, grp);
			if (prev == 0xFFFFFFFFu || prev == grp) {
				placed = true;
				break;
			}
			h = (h + 1u) & (GYS_SVCA_SLOTS - 1u);
		}
		if (placed) {
			atomicAdd(&s_cnt[h], 1ull);
			for (uint32_t a = 0; a < p.ncols; ++a) {
				const long long v = (long long)svc_col_value(w, p.cols[a]);
				atomicAdd((unsigned long long *)&s_acc[h][a][0], (unsigned long long)v);
				atomicMin(&s_acc[h][a][1], v);
				atomicMax(&s_acc[h][a][2], v);
			}
		} else {
			atomicAdd(&p.count[grp], 1ull);
			for (uint32_t a = 0; a < p.ncols; ++a) {
				const long long v = (long long)svc_col_value(w, p.cols[a]);
				long long *g = p.acc + ((size_t)grp * p.ncols + a) * 3;
				atomicAdd((unsigned long long *)&g[0], (unsigned long long)v);
				atomicMin(&g[1], v);
				atomicMax(&g[2], v);
			}
		}
	}
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < GYS_SVCA_SLOTS; k += GYS_SVCQ_THREADS) {
		const uint32_t grp = s_key[k];
		if (grp == 0xFFFFFFFFu || s_cnt[k] == 0ull) continue;
		atomicAdd(&p.count[grp], s_cnt[k]);
		for (uint32_t a = 0; a < p.ncols; ++a) {
			long long *g = p.acc + ((size_t)grp * p.ncols + a) * 3;
			atomicAdd((unsigned long long *)&g[0], (unsigned long long)s_acc[k][a][0]);
			atomicMin(&g[1], s_acc[k][a][1]);
			atomicMax(&g[2], s_acc[k][a][2]);
		}
	}
}

} // namespace gys

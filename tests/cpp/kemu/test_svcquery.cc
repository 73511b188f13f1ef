// TEST INFRASTRUCTURE (CPU): the LOGIC of the filtered multi-host listener-state query kernels (gys_svcquery.hpp: k_svc_filter, the radix
// selection k_svc_hist / k_svc_pick, k_svc_gather, k_svc_aggr) under the CPU stand-in of the device model, against the oracle's serial
// walk (oracle/gy_oracle_query.c: gyo_svcstate_scan / gyo_svcstate_aggr).  Records: random bytes in every field (values whose `int` view
// is negative, delay fields whose unsigned difference wraps), stale / deleted / foreign records, a host subset; filters: random terms of
// every comparator over every column in up to three groups with either operator; sorts on random columns, both directions, maxrecs below
// and above the number of matches (many ties: the slot tie-break and the exact top-k are exercised).
// Build + run: tests/test_kernel_logic_cpu.py.
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+r"(a[0]), "+r"(a[1]), "+r"(a[2]), "+r"(a[3]))
#define GYS_DYN_LDS(type, name) type *name = (type *)kemu::dyn_lds()
#include "../../../gyeeta_amd/csrc/gys_kernels.hpp"
#include "../../../gyeeta_amd/csrc/gys_svcquery.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <random>

#include "../../../oracle/gy_oracle.h"

using namespace gys;

namespace {
int fails = 0;
#define CHECK(c, ...)                                               \
	do {                                                        \
		if (!(c)) {                                         \
			if (fails++ < 20) {                         \
				printf("FAIL %s:%d: ", __FILE__, __LINE__); \
				printf(__VA_ARGS__);                \
				printf("\n");                       \
			}                                           \
		}                                                   \
	} while (0)
} // namespace

int main(int argc, char **argv)
{
	if (!kemu::can_run(GYS_SVCQ_THREADS)) {
		printf("kemu: this process cannot have %u threads\n", GYS_SVCQ_THREADS);
		return 77;
	}
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 5u);
	const uint32_t NH = 23, NSVC = 2300 + rng() % 300u, EPOCH = 9, NCL = 4;
	std::vector<uint8_t> state((size_t)NSVC * 96, 0);
	std::vector<uint32_t> svc_host(NSVC), host_cluster(NH);
	std::vector<uint64_t> svc_gid(NSVC);
	for (uint32_t h = 0; h < NH; ++h) host_cluster[h] = rng() % NCL;
	for (uint32_t s = 0; s < NSVC; ++s) {
		svc_host[s] = s * NH / NSVC; // contiguous runs of slots per host
		svc_gid[s] = 0x5000000000000000ull + 977ull * s;
		uint8_t *r = &state[(size_t)s * 96];
		for (int k = 8; k < 88; ++k) r[k] = (uint8_t)rng();
		// small values in most fields so that comparisons hit (the random high bytes stay in a fifth of the records)
		if (rng() % 5u) {
			for (int off = 8; off < 76; off += 4) {
				const uint32_t v = rng() % 40u;
				memcpy(r + off, &v, 4);
			}
			r[78] = rng() % 2u;
			r[79] = rng() % 7u;
			r[80] = rng() % 12u;
		}
		memcpy(r, &svc_gid[s], 8);
		uint32_t ep = EPOCH - (rng() % 3u == 0 ? 1u : 0u), host = svc_host[s];
		const uint32_t kind = rng() % 20u;
		if (kind == 0) ep = EPOCH - 2u;        // stale
		else if (kind == 1) ep = 0;           // deleted / never reported
		else if (kind == 2) host = (host + 1) % NH; // tagged with another host
		else if (kind == 3) r[3] ^= 0x40;      // another listener's record in the slot
		memcpy(r + 88, &ep, 4);
		memcpy(r + 92, &host, 4);
	}
	std::vector<unsigned long long> cand_key(NSVC), out_keys(NSVC);
	std::vector<uint32_t> cand_slot(NSVC), misc(16 + GYS_SVCQ_RADIX, 0), out_slots_o(NSVC);
	std::vector<uint8_t> out_rows((size_t)NSVC * 96);
	uint32_t nqueries = 0;
	for (uint32_t q = 0; q < 40; ++q) {
		SvcFilterP p{};
		p.svc_state = state.data();
		p.svc_host = svc_host.data();
		p.svc_gid = svc_gid.data();
		p.nsvc = NSVC;
		p.epoch = EPOCH;
		// the filter
		const uint32_t nterms = q == 0 ? 0u : 1u + rng() % 6u;
		std::vector<gyo_svc_term> ot(nterms);
		std::vector<int64_t> osetv;
		std::vector<int32_t> setv;
		uint8_t goper[8] = {0};
		for (int g = 0; g < 8; ++g) goper[g] = rng() % 2u;
		const int top_oper = rng() % 2u;
		for (uint32_t i = 0; i < nterms; ++i) {
			static const uint8_t comps[] = {0, 1, 2, 3, 4, 5, 6, 7, 12, 13};
			gyo_svc_term &t = ot[i];
			memset(&t, 0, sizeof(t));
			t.col = (uint8_t)(rng() % SVC_NCOLS);
			t.comp = comps[rng() % 10u];
			t.group = (uint8_t)(rng() % 3u);
			t.value = (int64_t)(rng() % 40u) - (rng() % 8u == 0 ? 20 : 0);
			if (t.comp >= 12) {
				t.set_first = (uint32_t)osetv.size();
				t.nvalues = rng() % 5u;
				for (uint32_t k = 0; k < t.nvalues; ++k) osetv.push_back((int64_t)(rng() % 40u));
			}
			SvcTerm &d = p.terms[i];
			d.col = t.col;
			d.comp = t.comp;
			d.group = t.group;
			d.pad = 0;
			d.nvalues = t.nvalues;
			d.set_first = t.set_first;
			d.value = t.col == SVC_COL_ISSUE ? (int32_t)(int16_t)t.value : (int32_t)t.value;
			p.ngroups = std::max<uint32_t>(p.ngroups, t.group + 1u);
		}
		for (size_t k = 0; k < osetv.size(); ++k) setv.push_back((int32_t)osetv[k]);
		p.nterms = nterms;
		p.set_values = setv.empty() ? nullptr : setv.data();
		memcpy(p.group_oper, goper, 8);
		p.top_oper = (uint32_t)top_oper;
		// a host subset for every third query
		std::vector<uint8_t> host_in(NH, 1);
		std::vector<uint32_t> mask((NH + 31) / 32 + 1, 0);
		const bool subset = q % 3u == 2u;
		if (subset) {
			for (uint32_t h = 0; h < NH; ++h) {
				host_in[h] = rng() % 2u;
				if (host_in[h]) mask[h >> 5] |= 1u << (h & 31u);
			}
			p.host_mask = mask.data();
		}
		// the query names its listeners in every fourth query: a sorted list of slots (svcid = / in)
		std::vector<uint32_t> slot_list;
		const bool named = q % 4u == 3u;
		if (named) {
			for (uint32_t s2 = 0; s2 < NSVC; ++s2)
				if (rng() % 7u == 0) slot_list.push_back(s2);
			p.slot_list = slot_list.data();
		}
		p.nitems = named ? (uint32_t)slot_list.size() : NSVC;
		p.sort_col = (rng() % 4u == 0) ? -1 : (int32_t)(rng() % SVC_NCOLS);
		p.sort_desc = rng() % 2u;
		p.cand_key = cand_key.data();
		p.cand_slot = cand_slot.data();
		std::fill(misc.begin(), misc.end(), 0u);
		p.cursor = &misc[0];
		const uint32_t per_wg = GYS_SVCQ_THREADS * GYS_SVCQ_PER_THREAD;
		kemu::launch(std::max(1u, (p.nitems + per_wg - 1) / per_wg), GYS_SVCQ_THREADS, 0, [&] { k_svc_filter(p); });
		const uint32_t ncand = misc[0];
		for (uint32_t maxrecs : {NSVC, 1u + (uint32_t)(rng() % 60u), 1u}) {
			uint64_t nm = 0;
			const uint32_t want_n = gyo_svcstate_scan(state.data(), NSVC, EPOCH, svc_host.data(), svc_gid.data(), subset ? host_in.data() : nullptr, ot.data(), nterms,
								   osetv.data(), goper, top_oper, p.sort_col, (int)p.sort_desc, maxrecs, out_slots_o.data(), &nm, named ? slot_list.data() : nullptr,
								   (uint32_t)slot_list.size());
			CHECK(ncand == nm, "query %u: %u candidates, the oracle matched %llu", q, ncand, (unsigned long long)nm);
			const uint32_t k = std::min(maxrecs, NSVC);
			const bool select = ncand > k;
			misc[1] = k;
			misc[2] = 0;
			misc[4] = misc[5] = 0;
			if (select) {
				SvcSelectP sp{};
				sp.cand_key = cand_key.data();
				sp.ncand = &misc[0];
				sp.hist = &misc[8];
				sp.prefix = (unsigned long long *)&misc[4];
				sp.want = &misc[1];
				static const uint32_t shifts[GYS_SVCQ_ROUNDS] = {53, 42, 31, 20, 9, 0}, widths[GYS_SVCQ_ROUNDS] = {11, 11, 11, 11, 11, 9};
				for (uint32_t r = 0; r < GYS_SVCQ_ROUNDS; ++r) {
					sp.shift = shifts[r];
					sp.bits = widths[r];
					kemu::launch(3, 256, 0, [&] { k_svc_hist(sp); });
					kemu::launch(1, 256, 0, [&] { k_svc_pick(sp); });
				}
			}
			SvcGatherP gp{};
			gp.svc_state = state.data();
			gp.cand_key = cand_key.data();
			gp.cand_slot = cand_slot.data();
			gp.ncand = &misc[0];
			gp.threshold = select ? (const unsigned long long *)&misc[4] : nullptr;
			gp.maxout = k;
			gp.out_count = &misc[2];
			gp.out_rows = out_rows.data();
			gp.out_keys = out_keys.data();
			kemu::launch(2, 256, 0, [&] { k_svc_gather(gp); });
			const uint32_t nout = misc[2];
			CHECK(nout == want_n, "query %u maxrecs %u: %u rows, the oracle %u", q, maxrecs, nout, want_n);
			std::vector<uint32_t> order(nout);
			for (uint32_t i = 0; i < nout; ++i) order[i] = i;
			std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return out_keys[a] > out_keys[b]; });
			for (uint32_t i = 0; i < std::min(nout, want_n); ++i) {
				const uint8_t *row = &out_rows[(size_t)order[i] * 96];
				uint32_t slot, host;
				memcpy(&slot, row, 4);
				memcpy(&host, row + 4, 4);
				CHECK(slot == out_slots_o[i], "query %u maxrecs %u: row %u is slot %u, the oracle's is %u", q, maxrecs, i, slot, out_slots_o[i]);
				if (slot != out_slots_o[i]) break;
				CHECK(host == svc_host[slot] && !memcmp(row + 8, &state[(size_t)slot * 96], 88), "query %u: row %u carries another record", q, i);
			}
			++nqueries;
		}
		// the aggregation over the same filter
		for (int group_by = 0; group_by < 3; ++group_by) {
			const uint32_t ngroups = group_by == 0 ? 1u : group_by == 1 ? NH : NCL, ncols = 1u + rng() % GYS_SVCQ_MAX_AGGR;
			SvcAggrP a{};
			a.svc_state = p.svc_state;
			a.svc_host = p.svc_host;
			a.svc_gid = p.svc_gid;
			a.nsvc = NSVC;
			a.epoch = EPOCH;
			a.host_mask = p.host_mask;
			a.slot_list = p.slot_list;
			a.nitems = p.nitems;
			a.set_values = p.set_values;
			a.nterms = p.nterms;
			a.ngroups = p.ngroups;
			memcpy(a.terms, p.terms, sizeof(p.terms));
			memcpy(a.group_oper, p.group_oper, 8);
			a.top_oper = p.top_oper;
			a.group_by = (uint32_t)group_by;
			a.host_cluster = host_cluster.data();
			a.ncols = ncols;
			uint8_t cols[GYS_SVCQ_MAX_AGGR];
			for (uint32_t c = 0; c < ncols; ++c) a.cols[c] = cols[c] = (uint8_t)(rng() % SVC_NCOLS);
			std::vector<long long> acc((size_t)ngroups * ncols * 3), oacc;
			for (size_t i = 0; i < acc.size(); i += 3) {
				acc[i] = 0;
				acc[i + 1] = 0x7FFFFFFFFFFFFFFFll;
				acc[i + 2] = -0x7FFFFFFFFFFFFFFFll - 1ll;
			}
			oacc = acc;
			std::vector<unsigned long long> cnt(ngroups, 0);
			std::vector<uint64_t> ocnt(ngroups, 0);
			a.acc = acc.data();
			a.count = cnt.data();
			kemu::launch(std::max(1u, (a.nitems + per_wg - 1) / per_wg), GYS_SVCQ_THREADS, 0, [&] { k_svc_aggr(a); });
			gyo_svcstate_aggr(state.data(), NSVC, EPOCH, svc_host.data(), svc_gid.data(), subset ? host_in.data() : nullptr, ot.data(), nterms, osetv.data(), goper,
					  top_oper, group_by, host_cluster.data(), cols, ncols, (int64_t *)oacc.data(), ocnt.data(), named ? slot_list.data() : nullptr,
					  (uint32_t)slot_list.size());
			for (uint32_t g = 0; g < ngroups; ++g) {
				CHECK(cnt[g] == ocnt[g], "query %u group_by %d: group %u counts %llu, the oracle %llu", q, group_by, g, cnt[g], (unsigned long long)ocnt[g]);
				if (!ocnt[g]) continue;
				for (uint32_t c = 0; c < ncols * 3; ++c)
					CHECK(acc[(size_t)g * ncols * 3 + c] == oacc[(size_t)g * ncols * 3 + c], "query %u group_by %d group %u: accumulator %u differs", q, group_by, g, c);
			}
		}
	}
	if (fails) {
		printf("kemu svcquery: %d FAILURES\n", fails);
		return 1;
	}
	printf("kemu svcquery ok: %u scans + aggregations over %u services\n", nqueries, NSVC);
	return 0;
}

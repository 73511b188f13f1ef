// TEST INFRASTRUCTURE (CPU): the LOGIC of the paths a key takes when one ingest call brings it more values than its buffer has room for
// -- the fused event pass drops the pieces that do not fit and finalize_key spills the key (run allocated in `staged`, host flagged),
// the SPILL pass of k_resp_host writes the spilled keys' values into their runs, and the keys are re-clustered from buffer + run by
// k_digest_bins (<= 1024 values), k_digest_merge<4096>, the several-workgroup path of gys_huge.hpp (k_huge_plan / clear / count /
// merge, with values >= 16 384 ms in the sorted tail) or its one-workgroup fallback k_digest_huge -- under the CPU stand-in of the
// device model, compared with the oracle's sequential engine after every batch.  Build + run: tests/test_kernel_logic_cpu.py.
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+r"(a[0]), "+r"(a[1]), "+r"(a[2]), "+r"(a[3]))
#define GYS_DYN_LDS(type, name) type *name = (type *)kemu::dyn_lds()
#include "../../../gyeeta_amd/csrc/gys_kernels.hpp"
#include "../../../gyeeta_amd/csrc/gys_huge.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <random>

#include "../../../oracle/gy_oracle.h"

extern "C" {
struct gyo_engine;
gyo_engine *gyo_engine_new(uint32_t max_services, int enable_td);
void gyo_engine_free(gyo_engine *e);
int gyo_engine_register(gyo_engine *e, uint32_t host_slot, uint64_t glob_id, uint32_t netns, uint16_t port);
void gyo_engine_resp_batch(gyo_engine *e, const uint8_t *ev24, uint64_t n, const uint32_t *seg_host, const uint64_t *seg_first, uint32_t nsegs);
const gyo_hist_serial *gyo_engine_hist(const gyo_engine *e);
const uint8_t *gyo_engine_hll(const gyo_engine *e);
const uint16_t *gyo_engine_bitmap(const gyo_engine *e);
void gyo_engine_window_clear(gyo_engine *e, int clear_hist);
const gyo_td_buffered *gyo_engine_td(const gyo_engine *e, uint32_t slot);
const uint64_t *gyo_engine_counters(const gyo_engine *e);
}

#ifndef KEMU_BINS_NT
#define KEMU_BINS_NT 256
#endif

#ifndef KEMU_APPEND_CAP
#define KEMU_APPEND_CAP 3u
#endif

using namespace gys;

namespace {
int fails = 0;
unsigned tier_b_entries = 0; // pool entries k_huge_merge's tier A handed to tier B over the whole run
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
uint16_t bswap(uint16_t v) { return (uint16_t)((v >> 8) | (v << 8)); }
} // namespace

int main(int argc, char **argv)
{
	if (!kemu::can_run(1024u)) {
		printf("kemu: this process cannot have 1024 threads\n");
		return 77;
	}
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 777u);
	constexpr uint32_t TPT = 16, T = GYS_RESP_THREADS(TPT), TILE = TPT * T;
	const uint32_t NH = 2, L[NH] = {6, 40};
	const uint32_t pcap = GYS_TD_PEND_CAP + 64u; // the smallest buffer the library accepts: spills are easy to provoke
	uint32_t nsvc = L[0] + L[1];

	gyo_engine *orc = gyo_engine_new(nsvc + 8, 1);
	gyo_engine *orcw = gyo_engine_new(nsvc + 8, 0); // the same stream with the records cleared at every window roll: what hist_win / CONN_BITMAP must show
	std::vector<HostDesc> hdesc(NH);
	std::vector<uint64_t> htbl;
	std::vector<uint32_t> hlst, svc_host(nsvc);
	uint32_t max_tbl = 0, max_l = 0, slot0 = 0;
	for (uint32_t h = 0; h < NH; ++h) {
		uint32_t cap = 1;
		while (cap < 2 * L[h]) cap <<= 1;
		HostDesc d{(uint32_t)htbl.size(), cap - 1, L[h], (uint32_t)hlst.size()};
		htbl.resize(htbl.size() + cap, GYS_HOST_TBL_EMPTY);
		for (uint32_t s = 0; s < L[h]; ++s) {
			const uint32_t netns = 0xF0000000u + 4u * h;
			const uint16_t port = (uint16_t)(1024 + s);
			gyo_engine_register(orc, h, 0x100000ull * (h + 1) + s, netns, port);
			gyo_engine_register(orcw, h, 0x100000ull * (h + 1) + s, netns, port);
			const uint64_t key48 = ((uint64_t)netns << 16) | port;
			uint32_t at = host_tbl_slot(host_tbl_hash(key48), d.mask);
			while (htbl[d.tbl_off + at] != GYS_HOST_TBL_EMPTY) at = (at + 1) & d.mask;
			htbl[d.tbl_off + at] = (key48 << 16) | s;
			hlst.push_back(slot0 + s);
			svc_host[slot0 + s] = h;
		}
		hdesc[h] = d;
		max_tbl = std::max(max_tbl, cap);
		max_l = std::max(max_l, L[h]);
		slot0 += L[h];
	}

	std::vector<int64_t> td_sum((size_t)nsvc * GYS_TD_NB, 0);
	std::vector<uint32_t> td_cnt((size_t)nsvc * GYS_TD_NB, 0), td_pend((size_t)nsvc * pcap, 0), td_cur(nsvc + 64, 0), td_run(nsvc, 0), staged(1u << 21, 0), bitmap((size_t)nsvc * GYS_BM_WORDS, 0),
		hll32(1u << GYS_HLL_P, 0), resp_win(nsvc, 0), host_spill(NH, 0), counts(16, 0);
	std::vector<TdMeta> meta(nsvc, TdMeta{0, 0, 0, 0, 0});
	std::vector<int2> minmax(nsvc, make_int2(INT32_MAX, INT32_MIN));
	std::vector<gys_hist_rec> hist_all(nsvc), hist_win(nsvc);
	for (auto *hv : {&hist_all, &hist_win})
		for (auto &r : *hv) {
			memset(&r, 0, sizeof(r));
			r.max_val_seen = INT64_MIN;
		}
	std::vector<MergeEnt> list0(nsvc + 1), list1(nsvc + 1), list2(nsvc + 1), listh(nsvc + 1), slow(nsvc + 1), fb(nsvc + 1);
	std::vector<uint64_t> counters(CTR_NUM, 0);
	std::vector<unsigned long long> ghist(32, 0);
	long long gmax = INT64_MIN;
#ifdef KEMU_PRESPILL
	std::vector<uint32_t> td_run0(nsvc, 0), td_run1(nsvc, 0), td_prevm(nsvc, 0), pre_hot(2, 0), host_batch(NH, 0);
	std::vector<MergeEnt> append_list(nsvc + 1);
	uint32_t pre_seq = 0, n_pre_inplace = 0, n_pre_append = 0, n_pre_fallback = 0;
#endif
	// the pools of the several-workgroup path (2 entries at a time: a third huge key goes through a second round) and of the fallback
	const uint32_t maxent = 2, huge_blocks = 1;
	std::vector<uint32_t> hbins((size_t)maxent * GYS_HB_BINS), hbm((size_t)maxent * GYS_BM_WORDS), chunk_off(maxent + 1), scratch((size_t)huge_blocks * GYS_HUGE_BINS, 0);
	std::vector<unsigned long long> hacc((size_t)maxent * GYS_HB_ACC), tail(1u << 16);

	// per-key events of host 0 by batch (host 1 always gets 2 000 events over its 40 services: never spilled)
	// batch 0: 400 per key (buffered) | 1: enough to pass the buffer's end by 20 (spilled, class 0 from buffer + run) | 2: 700 | 3: 1 500 (spilled, class 1)
	// | 4: 21 000 per key, a twentieth of them >= 16 384 ms (spilled, the several-workgroup path in three rounds) | 5: 3 more per key
	// | 6: 20 000 per key, ALL >= 16 384 ms for key 0 (more tail values than the LDS tail takes: the one-workgroup fallback)
	// | 7, 8: 1 500 and 1 600 per key (spilled, class 1): with predicted runs (KEMU_PRESPILL: k_prespill before the event pass) these two find
	// their values in the predicted run -- no second pass; batches 1 and 4 overflow their predicted runs (564 words for 580 values, 1 939 for
	// 21 000: the exact-run fall-back), batch 5 under-runs it (3 values for a run predicted from 21 000: the run is copied into the buffer,
	// k_run_append)
	const uint32_t per_key[] = {400, pcap - 400u + 20u, 700, 1500, 21000, 3, 20000, 1500, 1600};
	const uint32_t NB = sizeof(per_key) / sizeof(per_key[0]);
	uint32_t stamp = 0, epoch = 1;
	for (uint32_t batch = 0; batch < NB; ++batch) {
		// window rolls (round 4): before batches 1, 3, 6 and 8 a window closes -- the keys' buffers then hold words of the window before (not
		// yet folded: the fold is lazy; 400 words before batch 1 = a k_digest_bins merge, 700 before batch 3 = k_digest_merge<4096>, 3 before
		// batch 6 = the large-key path and its fallback), and the merge of the next batch must put them into the all-time record only
		// while the new window's words also go to hist_win / CONN_BITMAP (classes 1 and 2 of the merges' buffered-word passes)
		if (batch == 1 || batch == 3 || batch == 6 || batch == 8) {
			++epoch;
			gyo_engine_window_clear(orc, 0);
			gyo_engine_window_clear(orcw, 1);
			std::fill(hll32.begin(), hll32.end(), 0u);
		}
		std::vector<uint8_t> ev;
		std::vector<gys_resp_seg> segs;
		std::vector<uint32_t> seg_host;
		std::vector<uint64_t> seg_first;
		for (uint32_t h = 0; h < NH; ++h) {
			segs.push_back(gys_resp_seg{h, 0u, ev.size() / 24});
			seg_host.push_back(h);
			seg_first.push_back(ev.size() / 24);
			const uint32_t nev = h == 0 ? per_key[batch] * L[0] : 2000u;
			std::lognormal_distribution<double> ln(3.0 + 0.3 * batch, 1.2);
			for (uint32_t i = 0; i < nev; ++i) {
				uint32_t w[6];
				const uint32_t svc = h == 0 ? i % L[0] : rng() % L[1];
				double lat = std::floor(ln(rng));
				if (h == 0 && batch == 4 && rng() % 20 == 0) lat = 16384.0 + (double)(rng() % 900000u);
				if (h == 0 && batch == 6 && svc == 0) lat = 16384.0 + (double)(rng() % 900000u);
				if (lat > 999999.0) lat = 999999.0;
				const uint32_t tresp = (uint32_t)lat;
				w[0] = 0x0A000000u | (rng() & 0xFFFFFFu);
				w[1] = 0x0B000000u | (rng() & 0xFFFFu);
				w[2] = 0xF0000000u + 4u * h;
				const uint16_t sport = (uint16_t)(1024 + svc), dport = (uint16_t)(20000 + (rng() % 3000));
				w[3] = (uint32_t)bswap(sport) | ((uint32_t)bswap(dport) << 16);
				const uint32_t lrcv = rng();
				w[4] = lrcv + tresp;
				w[5] = lrcv;
				const size_t at = ev.size();
				ev.resize(at + 24);
				memcpy(&ev[at], w, 24);
			}
		}
		const uint64_t n = ev.size() / 24;
		std::vector<uint64_t> ev64(n * 3);
		memcpy(ev64.data(), ev.data(), n * 24);
		gyo_engine_resp_batch(orc, ev.data(), n, seg_host.data(), seg_first.data(), NH);
		gyo_engine_resp_batch(orcw, ev.data(), n, seg_host.data(), seg_first.data(), NH);

		std::fill(counts.begin(), counts.end(), 0u);
		FinP fin{};
		fin.td_cur = td_cur.data();
		fin.td_meta = meta.data();
		fin.nsvc = nsvc;
		fin.pcap = pcap;
		fin.pend_cap = GYS_TD_PEND_CAP;
		fin.merge_fast = GYS_TDIGEST_MERGE_FAST;
		fin.epoch = epoch;
		fin.resp_win = resp_win.data();
		fin.list[FIN_CLASS0] = list0.data();
		fin.list[FIN_CLASS1] = list1.data();
		fin.list[FIN_CLASS2] = list2.data();
		fin.list[FIN_HUGE] = listh.data();
		fin.counts = counts.data();
		fin.td_run = td_run.data();
		fin.svc_host = svc_host.data();
		fin.host_spill = host_spill.data();
		fin.spill_stamp = ++stamp;
		fin.counters = counters.data();
		fin.staged_cap = (uint32_t)staged.size();
#ifdef KEMU_PRESPILL
		fin.td_run0 = td_run0.data();
		fin.td_run1 = td_run1.data();
		fin.td_prevm = td_prevm.data();
		fin.hot = pre_hot.data();
		fin.hot_wr = (pre_seq & 1u) ^ 1u;
		fin.append_list = append_list.data();
		// (the append list is shorter than the keys batch 5 puts on it: the rest is copied by the finalizing threads themselves)
		fin.append_cap = KEMU_APPEND_CAP;
		fin.td_pend = td_pend.data();
		fin.staged = staged.data();
		counts[FIN_APPEND] = 0;
		{
			PreSpillP pp{};
			pp.td_cur = td_cur.data();
			pp.td_prevm = td_prevm.data();
			pp.td_run = td_run.data();
			pp.td_run0 = td_run0.data();
			pp.td_run1 = td_run1.data();
			pp.counts = counts.data();
			pp.hot = pre_hot.data();
			pp.hot_rd = pre_seq & 1u;
			pp.svc_host = svc_host.data();
			pp.host_batch = host_batch.data();
			pp.batch_stamp = stamp + 1u;
			pp.nsvc = nsvc;
			pp.pcap = pcap;
			pp.pend_cap = GYS_TD_PEND_CAP;
			pp.resv = (unsigned long long *)&counts[14];
			// (batch 7: no room for predicted runs -- the cursor stays where it is and the keys take the exact-run fall-back)
			pp.run_limit = batch == 7 ? 100u : (uint32_t)(staged.size() - n);
			++pre_seq;
			kemu::launch(1, 256, 0, [&] { k_mark_hosts(segs.data(), NH, host_batch.data(), stamp + 1u); });
			kemu::launch((nsvc + 255u) / 256u, 256, 0, [&] { k_prespill(pp); });
		}
		uint32_t npred = 0;
		for (uint32_t s = 0; s < nsvc; ++s) npred += (td_cur[s] & GYS_SPILL_BIT) ? 1u : 0u;
		CHECK(npred == ((batch == 1 || batch == 4 || batch == 5 || batch == 8) ? L[0] : 0u), "batch %u: %u keys got a predicted run", batch, npred);
		if (batch == 7) CHECK(counts[FIN_RUN_ALLOC] == 0, "batch 7: the cursor moved by %u though no run was accepted", counts[FIN_RUN_ALLOC]);
#endif
		RespHostP hp{};
		hp.ev = ev64.data();
		hp.n = n;
		hp.segs = segs.data();
		hp.nsegs = NH;
		hp.hdesc = hdesc.data();
		hp.htbl = htbl.data();
		hp.hlst = hlst.data();
		hp.hll32 = hll32.data();
		hp.td_cur = td_cur.data();
		hp.td_pend = td_pend.data();
		hp.pcap = pcap;
		hp.td_run = td_run.data();
#ifdef KEMU_PRESPILL
		hp.td_run1 = td_run1.data();
		hp.run_delta = (long long)(((intptr_t)staged.data() - (intptr_t)td_pend.data()) / 4);
#endif
		hp.staged = staged.data();
		hp.host_spill = host_spill.data();
		hp.spill_stamp = stamp;
		hp.counters = counters.data();
		hp.ghist = ghist.data();
		hp.gmax = &gmax;
		hp.lds_tbl_entries = max_tbl;
		hp.lds_key_entries = (max_l + 1u) & ~1u;
		hp.fin = fin;
		const size_t dyn = resp_host_lds_bytes(max_tbl, hp.lds_key_entries, TILE);
		kemu::launch(NH, T, dyn, [&] { k_resp_host<TPT, false, false, false>(hp); });
		CHECK(counts[FIN_RUN_ALLOC] <= staged.size(), "run area too small");
#ifdef KEMU_PRESPILL
		// batches 7 and 8 are predicted right (no second pass), batches 1 and 4 overflow their predicted runs (second pass), batch 5's runs go into the buffers
		const bool expect_spill = batch == 1 || batch == 3 || batch == 4 || batch == 6 || batch == 7;
		if (batch == 5) {
			CHECK(counts[FIN_APPEND] == std::min<uint32_t>(L[0], KEMU_APPEND_CAP), "batch 5: %u keys on the append list", counts[FIN_APPEND]);
			n_pre_append += L[0];
		} else
			CHECK(counts[FIN_APPEND] == 0, "batch %u: %u keys on the append list", batch, counts[FIN_APPEND]);
		if (batch == 8) n_pre_inplace += L[0];
		if (batch == 1 || batch == 4) n_pre_fallback += L[0];
		kemu::launch(2, 256, 0, [&] { k_run_append(append_list.data(), &counts[FIN_APPEND], staged.data(), td_pend.data(), pcap); });
#else
		const bool expect_spill = batch == 1 || batch == 3 || batch == 4 || batch == 6 || batch == 7 || batch == 8;
#endif
		CHECK((host_spill[0] == stamp) == expect_spill && host_spill[1] != stamp, "batch %u: host 0 %s flagged as spilled", batch, host_spill[0] == stamp ? "is" : "is not");
		// second pass over the hosts that have spilled keys: their values into the runs
		kemu::launch(NH, T, dyn, [&] { k_resp_host<TPT, true, true, false>(hp); });

		MergeBP q{};
		q.d.td_sum = td_sum.data();
		q.d.td_cnt = td_cnt.data();
		q.d.td_meta = meta.data();
		q.d.td_minmax = minmax.data();
		q.d.td_pend = td_pend.data();
		q.d.td_cur = td_cur.data();
		q.d.pcap = pcap;
		q.d.pend_cap = GYS_TD_PEND_CAP;
		q.d.nsvc = nsvc;
		q.d.staged = staged.data();
		q.d.hist_win = hist_win.data();
		q.d.hist_all = hist_all.data();
		q.d.bitmap = bitmap.data();
		q.list = list0.data();
		q.count = &counts[FIN_CLASS0];
		q.slow_list = slow.data();
		q.slow_count = &counts[FIN_SLOW];
		std::vector<uint32_t> merged;
		for (uint32_t i = 0; i < counts[FIN_CLASS0]; ++i) merged.push_back(list0[i].slot);
		for (uint32_t i = 0; i < counts[FIN_CLASS1]; ++i) merged.push_back(list1[i].slot);
		for (uint32_t i = 0; i < counts[FIN_HUGE]; ++i) merged.push_back(listh[i].slot);
#if defined(KEMU_BINS_TEMPLATE_NT) // (trees whose value-bin kernel is templated on the thread count)
		kemu::launch(2, KEMU_BINS_NT, 0, [&] { k_digest_bins<false, KEMU_BINS_NT>(q); });
#else
		kemu::launch(2, 256, 0, [&] { k_digest_bins<false>(q); });
#endif
		CHECK(counts[FIN_SLOW] == 0, "hand-over list not empty");
		MergeP mp{};
		mp.d = q.d;
		mp.list = list1.data();
		mp.count = &counts[FIN_CLASS1];
#if defined(KEMU_BINS_TEMPLATE_NT)
		q.list = list1.data();
		q.count = &counts[FIN_CLASS1];
		kemu::launch(2, 1024, 0, [&] { k_digest_bins<false, 1024u>(q); });
		CHECK(counts[FIN_SLOW] == 0, "hand-over list not empty");
#else
		kemu::launch(2, 256, 0, [&] { k_digest_merge<GYS_MERGE_CLASS1, 256u>(mp); });
#endif
		if (batch == 1) CHECK(counts[FIN_CLASS0] >= L[0], "batch 1: %u class-0 entries (spilled keys merged from buffer + run expected)", counts[FIN_CLASS0]);
		if (batch == 3 || batch == 7 || batch == 8) CHECK(counts[FIN_CLASS1] == L[0], "batch %u: %u class-1 entries", batch, counts[FIN_CLASS1]);
		if (batch == 4 || batch == 6) CHECK(counts[FIN_HUGE] == L[0], "batch %u: %u huge entries", batch, counts[FIN_HUGE]);
		if (counts[FIN_HUGE]) {
			Huge2P hq{};
			hq.d = q.d;
			hq.list = listh.data();
			hq.count = &counts[FIN_HUGE];
			hq.bins = hbins.data();
			hq.acc = hacc.data();
			hq.bm = hbm.data();
			hq.chunk_off = chunk_off.data();
			hq.tail = tail.data();
			hq.tail_count = &counts[9];
			hq.tail_cap = (uint32_t)tail.size();
			hq.maxent = maxent;
			hq.fb_list = fb.data();
			hq.fb_count = &counts[6];
			hq.nent_used = &counts[7];
			std::vector<uint32_t> tb(maxent + 1);
			hq.tb_list = tb.data();
			hq.tb_count = &counts[10];
			for (uint32_t first = 0; first < L[0]; first += maxent) {
				hq.first = first;
				kemu::launch(1, 1024, 0, [&] { k_huge_plan(hq); });
				kemu::launch(2, 256, 0, [&] { k_huge_clear(hq); });
				kemu::launch(2, 1024, GYS_HB_BINS * 4, [&] { k_huge_count(hq); });
				kemu::launch(2, 512, (GYS_HB_BINS + GYS_HB_TAIL_A) * 4, [&] { k_huge_merge<512, GYS_HB_TAIL_A, false>(hq); });
				tier_b_entries += counts[10];
				kemu::launch(2, 1024, (GYS_HB_BINS + GYS_HB_TAIL_LDS) * 4, [&] { k_huge_merge<1024, GYS_HB_TAIL_LDS, true>(hq); });
			}
			if (batch == 6) CHECK(counts[6] >= 1, "batch 6: the key with 20 000 tail values was not handed to the fallback");
			HugeP hf{};
			hf.d = q.d;
			hf.huge_list = fb.data();
			hf.huge_count = &counts[6];
			hf.scratch = scratch.data();
			kemu::launch(huge_blocks, 256, 0, [&] { k_digest_huge(hf); });
		}

		const uint64_t *oc = gyo_engine_counters(orc);
		CHECK(counters[CTR_RESP_EVENTS] == oc[0] && counters[CTR_RESP_DROP_RANGE] == oc[1] && counters[CTR_RESP_DROP_NOLISTENER] == oc[2], "batch %u counters", batch);
		const uint8_t *ohll = gyo_engine_hll(orc);
		for (uint32_t i = 0; i < (1u << GYS_HLL_P); ++i) CHECK(hll32[i] == ohll[i], "batch %u HLL register %u: %u want %u", batch, i, hll32[i], ohll[i]);
		const gyo_hist_serial *oh = gyo_engine_hist(orc);
		for (uint32_t s = 0; s < nsvc; ++s) {
			const gyo_td_buffered *ot = gyo_engine_td(orc, s);
			CHECK(meta[s].npend == ot->npend && td_cur[s] == ot->npend, "batch %u key %u buffered %u (cur %08x) want %u", batch, s, meta[s].npend, td_cur[s], ot->npend);
			if (meta[s].npend == ot->npend) {
				std::vector<int32_t> a(ot->npend), b(ot->pend, ot->pend + ot->npend);
				for (uint32_t i = 0; i < ot->npend; ++i) a[i] = (int32_t)(td_pend[(size_t)s * pcap + i] >> GYS_ROW_BITS);
				std::sort(a.begin(), a.end());
				std::sort(b.begin(), b.end());
				CHECK(a == b, "batch %u key %u: buffered values differ", batch, s);
			}
			for (int j = 0; j < GYS_TD_NB; ++j)
				CHECK(td_sum[(size_t)s * GYS_TD_NB + j] == ot->d.sum[j] && td_cnt[(size_t)s * GYS_TD_NB + j] == ot->d.cnt[j], "batch %u key %u cluster %d: {%lld, %u} want {%lld, %u}",
				      batch, s, j, (long long)td_sum[(size_t)s * GYS_TD_NB + j], td_cnt[(size_t)s * GYS_TD_NB + j], (long long)ot->d.sum[j], ot->d.cnt[j]);
		}
		for (uint32_t s : merged) {
			for (int b = 0; b < 15; ++b)
				CHECK(hist_all[s].stats[b].count == oh[(size_t)s * 16 + b].count && hist_all[s].stats[b].sum == oh[(size_t)s * 16 + b].sum, "batch %u key %u all-time bucket %d: {%llu, %lld} want {%llu, %lld}",
				      batch, s, b, (unsigned long long)hist_all[s].stats[b].count, (long long)hist_all[s].stats[b].sum, (unsigned long long)oh[(size_t)s * 16 + b].count, (long long)oh[(size_t)s * 16 + b].sum);
			CHECK(hist_all[s].total_count == oh[(size_t)s * 16 + 15].count && hist_all[s].max_val_seen == oh[(size_t)s * 16 + 15].sum, "batch %u key %u all-time total / max", batch, s);
			// the window's record and CONN_BITMAP rows of a key that was just merged (its buffer is drained: everything is folded); a key
			// whose records belong to an earlier window shows an empty window
			{
				const gyo_hist_serial *ow = gyo_engine_hist(orcw) + (size_t)s * 16;
				const uint16_t *ob = gyo_engine_bitmap(orcw) + (size_t)s * 64;
				const bool cur = meta[s].hw_epoch == epoch;
				for (int b = 0; b < 15; ++b) {
					const unsigned long long gc = cur ? hist_win[s].stats[b].count : 0ull;
					const long long gs = cur ? hist_win[s].stats[b].sum : 0ll;
					CHECK(gc == ow[b].count && gs == ow[b].sum, "batch %u key %u window bucket %d: {%llu, %lld} want {%llu, %lld}", batch, s, b, gc, gs, (unsigned long long)ow[b].count, (long long)ow[b].sum);
				}
				CHECK((cur ? hist_win[s].total_count : 0ull) == ow[15].count && (cur ? hist_win[s].max_val_seen : INT64_MIN) == ow[15].sum, "batch %u key %u window total / max", batch, s);
				for (int g = 0; g < (int)GYS_BM_WORDS; ++g) {
					const uint32_t want = (uint32_t)ob[2 * g] | ((uint32_t)ob[2 * g + 1] << 16);
					CHECK((cur ? bitmap[(size_t)s * GYS_BM_WORDS + g] : 0u) == want, "batch %u key %u CONN_BITMAP rows %d, %d: %08x want %08x", batch, s, 2 * g, 2 * g + 1, cur ? bitmap[(size_t)s * GYS_BM_WORDS + g] : 0u, want);
				}
			}
			const gyo_td_buffered *ot = gyo_engine_td(orc, s);
			CHECK(minmax[s].x == ot->d.vmin && minmax[s].y == ot->d.vmax, "batch %u key %u min/max {%d, %d} want {%d, %d}", batch, s, minmax[s].x, minmax[s].y, ot->d.vmin, ot->d.vmax);
		}
	}
	gyo_engine_free(orc);
	gyo_engine_free(orcw);
	if (fails) {
		printf("%d checks failed\n", fails);
		return 1;
	}
	if (!tier_b_entries) {
		printf("no entry went through tier B of k_huge_merge\n");
		return 1;
	}
#ifdef KEMU_PRESPILL
	printf("predicted runs: %u keys merged from their predicted run, %u copied into the buffer, %u fell back to the second pass\n", n_pre_inplace, n_pre_append, n_pre_fallback);
#endif
	printf("kemu spill ok (tier B entries: %u)\n", tier_b_entries);
	return 0;
}

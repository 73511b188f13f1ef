// TEST INFRASTRUCTURE (CPU): the LOGIC of the response-event pipeline -- k_resp_host (listener resolution in the host's LDS sub-table,
// filters, global HLL, all-service histogram, tile-wise counting sort into the per-service value buffers, end-of-batch bookkeeping of the
// keys = finalize_key) followed by k_digest_bins on the queued keys -- run under the CPU stand-in of the device model and compared, batch
// after batch, with the oracle's sequential engine (oracle/gy_oracle_engine.c) fed the same bytes: counters, HLL registers, the
// all-service histogram, every key's buffered values (as multisets) and digest, the per-window event counts, and for every key that
// has just been re-clustered its complete histogram record.  Build + run: tests/test_kernel_logic_cpu.py.
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+r"(a[0]), "+r"(a[1]), "+r"(a[2]), "+r"(a[3]))
#define GYS_DYN_LDS(type, name) type *name = (type *)kemu::dyn_lds()
#include "../../../gyeeta_amd/csrc/gys_kernels.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <random>

#include "../../../oracle/gy_oracle.h"

extern "C" {
struct gyo_engine;
gyo_engine *gyo_engine_new(uint32_t max_services, int enable_td);
gyo_engine *gyo_engine_new_cap(uint32_t max_services, int enable_td, uint32_t td_cap);
void gyo_engine_free(gyo_engine *e);
int gyo_engine_register(gyo_engine *e, uint32_t host_slot, uint64_t glob_id, uint32_t netns, uint16_t port);
int gyo_engine_register_addr(gyo_engine *e, uint32_t host_slot, uint64_t glob_id, uint32_t netns, uint16_t port, const uint8_t *ip, int is_v6, int is_any);
void gyo_engine_resp_batch(gyo_engine *e, const uint8_t *ev24, uint64_t n, const uint32_t *seg_host, const uint64_t *seg_first, uint32_t nsegs);
void gyo_engine_resp_batch_v6(gyo_engine *e, const uint8_t *ev48, uint64_t n, const uint32_t *seg_host, const uint64_t *seg_first, uint32_t nsegs);
const uint16_t *gyo_engine_bitmap(const gyo_engine *e);
void gyo_engine_window_clear(gyo_engine *e, int clear_hist);
const gyo_hist_serial *gyo_engine_hist(const gyo_engine *e);
const uint8_t *gyo_engine_hll(const gyo_engine *e);
const gyo_hist_serial *gyo_engine_ghist(const gyo_engine *e);
int64_t gyo_engine_gmax(const gyo_engine *e);
const gyo_td_buffered *gyo_engine_td(const gyo_engine *e, uint32_t slot);
const uint64_t *gyo_engine_counters(const gyo_engine *e);
}

#ifndef KEMU_TPT
#define KEMU_TPT 16
#endif
#ifndef KEMU_BINS_NT
#define KEMU_BINS_NT 256
#endif
#ifndef KEMU_NB
#define KEMU_NB 6 // batches
#endif
// KEMU_MODE 0: every listener alone on its (netns, port) key and any-address (k_resp_host<.., MODE 0>); 1: bound-address listeners and keys with
// several listeners, IPv4 events (MODE 1: candidates resolved by the event's server address, common/gy_socket_stat.h:708-714); 2: the same world,
// batches alternate between IPv4 events (MODE 1) and 48-byte IPv6 events (MODE 2: handle_ipv6_resp_event, common/gy_socket_stat.cc:1535-1551)
#ifndef KEMU_MODE
#define KEMU_MODE 0
#endif
// KEMU_PEND_CAP: the digests' buffer size (gys_config.td_pend_cap): 896 = the default (merges of up to 1024 values, k_digest_bins<., 4>),
// up to 1920 -> k_digest_bins<., 8>, up to 3968 -> k_digest_bins<., 16>
#ifndef KEMU_PEND_CAP
#define KEMU_PEND_CAP GYS_TD_PEND_CAP
#endif
#define KEMU_FAST (KEMU_PEND_CAP + 128u <= 1024u ? 1024u : KEMU_PEND_CAP + 128u <= 2048u ? 2048u : 4096u)
#define KEMU_VPT (KEMU_FAST <= 1024u ? 4u : KEMU_FAST <= 2048u ? 8u : 16u)

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

uint16_t bswap(uint16_t v) { return (uint16_t)((v >> 8) | (v << 8)); }

// the candidates world: service s of host h by kind s % 7 -- 3: bound to A1; 4: any-address on the port of service s - 1 (registered behind it:
// takes what A1's listener does not); 5: bound to A2, alone (events for other addresses find nobody); 6: bound to an IPv6 address; else alone
// and any-address
void addr_a1(uint32_t h, uint8_t out[4]) { out[0] = 10; out[1] = 1; out[2] = (uint8_t)h; out[3] = 3; }
void addr_a2(uint32_t h, uint8_t out[4]) { out[0] = 10; out[1] = 2; out[2] = (uint8_t)h; out[3] = 5; }
void addr_a6(uint32_t h, uint8_t out[16]) { memset(out, 0, 16); out[0] = 0x20; out[1] = 0x01; out[2] = 0x0d; out[3] = 0xb8; out[13] = (uint8_t)h; out[15] = 6; }
void addr_mapped(const uint8_t v4[4], uint8_t out[16]) { memset(out, 0, 16); out[10] = 0xFF; out[11] = 0xFF; memcpy(out + 12, v4, 4); }
uint16_t port_of(uint32_t s) { return (uint16_t)(1024 + (KEMU_MODE && s % 7 == 4 ? s - 1 : s)); }
} // namespace

int main(int argc, char **argv)
{
	if (!kemu::can_run(1024u)) {
		printf("kemu: this process cannot have 1024 threads\n");
		return 77;
	}
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 4242u);
	constexpr uint32_t T = GYS_RESP_THREADS(KEMU_TPT), TILE = (uint32_t)KEMU_TPT * T;
#if defined(KEMU_SPLIT) && KEMU_MODE
	const uint32_t NH = 3, L[NH] = {300, 37, 200}; // (two listeners on one port take twice a port's events: more listeners keep every key inside its buffer)
#else
	const uint32_t NH = 3, L[NH] = {150, 37, 64};
#endif
	const uint32_t pcap = KEMU_FAST + 128u;
	uint32_t nsvc = 0;
	for (uint32_t h = 0; h < NH; ++h) nsvc += L[h];

	// ---- registration: oracle engine + the host-local structures k_resp_host reads
	gyo_engine *orc = gyo_engine_new_cap(nsvc + 8, 1, KEMU_PEND_CAP), *orcw = gyo_engine_new_cap(nsvc + 8, 1, KEMU_PEND_CAP); // orcw: cleared at the window boundaries
	std::vector<ListenerCand> cands;
	std::vector<HostDesc> hdesc(NH);
	std::vector<uint64_t> htbl;
	std::vector<uint32_t> hlst, svc_host(nsvc);
	uint32_t max_tbl = 0, max_l = 0, slot0 = 0;
	for (uint32_t h = 0; h < NH; ++h) {
		uint32_t cap = 1;
		while (cap < 2 * L[h]) cap <<= 1;
		HostDesc d{(uint32_t)htbl.size(), cap - 1, L[h], (uint32_t)hlst.size()};
		d.cand_off = (uint32_t)cands.size();
		htbl.resize(htbl.size() + cap, GYS_HOST_TBL_EMPTY);
		for (uint32_t s = 0; s < L[h]; ++s) {
			const uint32_t netns = 0xF0000000u + 4u * h;
			const uint16_t port = port_of(s);
			const uint32_t kind = KEMU_MODE ? s % 7 : 0;
			uint8_t a[16] = {0};
			int a6 = 0, any = 1;
			if (kind == 3) { addr_a1(h, a); any = 0; }
			if (kind == 5) { addr_a2(h, a); any = 0; }
			if (kind == 6) { addr_a6(h, a); any = 0; a6 = 1; }
			for (gyo_engine *o : {orc, orcw}) {
				const int slot = gyo_engine_register_addr(o, h, 0x100000ull * (h + 1) + s, netns, port, a, a6, any);
				CHECK(slot == (int)(slot0 + s), "oracle slot %d", slot);
			}
			hlst.push_back(slot0 + s);
			svc_host[slot0 + s] = h;
			if (kind == 4) continue; // (its key's entry and candidates were made with service s - 1)
			const uint64_t key48 = ((uint64_t)netns << 16) | port;
			uint32_t at = host_tbl_slot(host_tbl_hash(key48), d.mask);
			while (htbl[d.tbl_off + at] != GYS_HOST_TBL_EMPTY) at = (at + 1) & d.mask;
			if (any) {
				htbl[d.tbl_off + at] = (key48 << 16) | s;
				continue;
			}
			// a key with candidates: this listener, and behind it the any-address one of kind 4 when there is one
			const bool two = kind == 3 && s + 1 < L[h];
			htbl[d.tbl_off + at] = (key48 << 16) | GYS_LOCAL_GROUP | ((uint32_t)cands.size() - d.cand_off);
			ListenerCand c{};
			if (a6) {
				memcpy(c.ip128, a, 16);
				c.ip32 = ip6_embedded_v4(c.ip128);
			} else {
				memcpy(&c.ip32, a, 4);
			}
			c.flags = two ? 0u : 2u;
			c.local = s;
			c.slot = slot0 + s;
			cands.push_back(c);
			if (two) {
				ListenerCand c2{};
				c2.flags = 1u | 2u;
				c2.local = s + 1;
				c2.slot = slot0 + s + 1;
				cands.push_back(c2);
			}
		}
		hdesc[h] = d;
		max_tbl = std::max(max_tbl, cap);
		max_l = std::max(max_l, L[h]);
		slot0 += L[h];
	}

	// ---- engine state
	std::vector<int64_t> td_sum((size_t)nsvc * GYS_TD_NB, 0);
	std::vector<uint32_t> td_cnt((size_t)nsvc * GYS_TD_NB, 0), td_pend((size_t)nsvc * pcap, 0), td_cur(nsvc + 64, 0), td_run(nsvc, 0), staged(1u << 20, 0), bitmap((size_t)nsvc * GYS_BM_WORDS, 0),
		hll32(1u << GYS_HLL_P, 0), resp_win(nsvc, 0), host_spill(NH, 0), counts(16, 0);
	std::vector<TdMeta> meta(nsvc, TdMeta{0, 0, 0, 0, 0});
	std::vector<int2> minmax(nsvc, make_int2(INT32_MAX, INT32_MIN));
	std::vector<gys_hist_rec> hist_all(nsvc), hist_win(nsvc);
	for (auto *hv : {&hist_all, &hist_win})
		for (auto &r : *hv) {
			memset(&r, 0, sizeof(r));
			r.max_val_seen = INT64_MIN;
		}
	std::vector<MergeEnt> list0(nsvc + 1), list1(nsvc + 1), list2(nsvc + 1), listh(nsvc + 1), slow(nsvc + 1);
	std::vector<uint64_t> counters(CTR_NUM, 0);
	std::vector<unsigned long long> ghist(32, 0);
	long long gmax = INT64_MIN;

	const uint32_t NB = KEMU_NB;
	uint32_t stamp = 0;
	uint64_t merges_seen = 0;
	for (uint32_t batch = 0; batch < NB; ++batch) {
		// ---- a batch: every host one segment; a short one, a multi-tile one, a one-tile one; some events dropped by both filters, some
		// with a zero address (the rolled general hash path)
#ifdef KEMU_SPLIT // the split form: long segments cut into parts of GYS_SPLIT_PART events, several workgroups per host, k_key_finalize afterwards
		const uint32_t nev[NH] = {GYS_SPLIT_PART + 9000u + 1000u * batch, batch == 3 ? 5u : 9000u, GYS_SPLIT_PART + 1u + batch}; // (two parts each; the second one of host 2 holds 1 .. 3 events)
#else
		const uint32_t nev[NH] = {33545u + 1000u * batch, batch == 3 ? 5u : 9000u, 16381u + batch}; // (16 384-event tiles: 2+ tiles / <1 tile / one tile minus 3 .. plus 2)
#endif
		const bool v6 = KEMU_MODE == 2 && (batch & 1u);
		const size_t evb = v6 ? 48 : 24;
		std::vector<uint8_t> ev;
		std::vector<gys_resp_seg> segs;
		std::vector<uint32_t> seg_host;
		std::vector<uint64_t> seg_first;
		for (uint32_t h = 0; h < NH; ++h) {
			segs.push_back(gys_resp_seg{h, 0u, ev.size() / evb});
			seg_host.push_back(h);
			seg_first.push_back(ev.size() / evb);
			std::lognormal_distribution<double> ln(2.5 + 0.7 * h + 0.1 * batch, 1.3);
			for (uint32_t i = 0; i < nev[h]; ++i) {
				uint32_t w[6];
				const uint32_t r = rng();
				uint32_t svc = rng() % L[h];
				if ((r & 0xFF) == 1) svc = L[h] + 5; // unknown listener
				double lat = std::floor(ln(rng));
				if (lat > 999999.0) lat = 999999.0;
				uint32_t tresp = (uint32_t)lat;
				if ((r & 0xFF00) == 0x0200) tresp = 1000001u + (r >> 20); // out of range
				w[0] = (r & 0xFF0000) == 0x030000 ? 0u : (0x0A000000u | (rng() & 0xFFFFFFu)); // saddr (server)
				w[1] = (r & 0xFF0000) == 0x040000 ? 0u : (0x0B000000u | (rng() & 0x3FFFu));   // daddr (client): few distinct -> HLL ranks repeat
				w[2] = 0xF0000000u + 4u * h;
				const uint16_t sport = port_of(svc), dport = (uint16_t)(20000 + (rng() % 3000));
				w[3] = (uint32_t)bswap(sport) | ((uint32_t)bswap(dport) << 16);
				const uint32_t lrcv = rng();
				w[4] = lrcv + tresp;
				w[5] = lrcv;
				const size_t at = ev.size();
				ev.resize(at + evb);
				if (!v6) {
					if (KEMU_MODE) { // the server address decides on keys with candidates: A1, A2, some other address, 0.0.0.0
						uint8_t a[4];
						const uint32_t pick = rng() % 8u;
						if (pick < 3u) { addr_a1(h, a); memcpy(&w[0], a, 4); }
						else if (pick < 5u) { addr_a2(h, a); memcpy(&w[0], a, 4); }
						else if (pick == 5u) w[0] = 0u;
					}
					memcpy(&ev[at], w, 24);
				} else {
					uint8_t sa[16], da[16], v4[4];
					const uint32_t pick = rng() % 8u;
					if (pick < 2u) addr_a6(h, sa);                                    // the IPv6-bound listeners' address
					else if (pick < 4u) { addr_a1(h, v4); addr_mapped(v4, sa); }       // ::ffff:A1 == A1
					else if (pick == 4u) { addr_a2(h, v4); addr_mapped(v4, sa); }      // ::ffff:A2 == A2
					else if (pick == 5u) memset(sa, 0, 16);                            // ::
					else { addr_a6(h, sa); sa[14] = (uint8_t)rng(); sa[15] = 7; }      // nobody is bound to it
					const uint32_t dp = rng() % 6u;
					memset(da, 0, 16);
					if (dp == 0u) { v4[0] = 11; v4[1] = 0; v4[2] = (uint8_t)(rng() & 63u); v4[3] = (uint8_t)rng(); addr_mapped(v4, da); }   // mapped client
					else if (dp == 1u) { da[0] = 0x20; da[1] = 0x02; da[2] = 11; da[3] = 0; da[4] = (uint8_t)(rng() & 63u); da[5] = (uint8_t)rng(); da[15] = 1; } // 6to4 client
					else if (dp == 2u) { da[0] = 0; da[1] = 0x64; da[2] = 0xFF; da[3] = 0x9B; da[12] = 11; da[13] = 0; da[14] = (uint8_t)(rng() & 63u); da[15] = (uint8_t)rng(); } // NAT64 client
					else { da[0] = 0xfd; da[1] = 0x12; da[14] = (uint8_t)(rng() & 63u); da[15] = (uint8_t)rng(); }
					memcpy(&ev[at], sa, 16);
					memcpy(&ev[at + 16], da, 16);
					memcpy(&ev[at + 32], &w[2], 16);
				}
			}
		}
		const uint64_t n = ev.size() / evb;
		std::vector<uint64_t> ev64(n * (evb / 8));
		memcpy(ev64.data(), ev.data(), n * evb);
		if (batch && batch % 3 == 0) gyo_engine_window_clear(orcw, 1);
		for (gyo_engine *o : {orc, orcw}) {
			if (v6) gyo_engine_resp_batch_v6(o, ev.data(), n, seg_host.data(), seg_first.data(), NH);
			else gyo_engine_resp_batch(o, ev.data(), n, seg_host.data(), seg_first.data(), NH);
		}

		// ---- the engine's side (what run_resp_batch sets up for the fused host-local form)
		std::fill(counts.begin(), counts.end(), 0u);
		const uint32_t epoch = 1 + batch / 3; // a window boundary after batches 2 and 5 (the records roll lazily)
		FinP fin{};
		fin.td_cur = td_cur.data();
		fin.td_meta = meta.data();
		fin.nsvc = nsvc;
		fin.pcap = pcap;
		fin.pend_cap = KEMU_PEND_CAP;
		fin.merge_fast = KEMU_FAST;
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
		RespHostP hp{};
		hp.ev = ev64.data();
		hp.n = n;
		hp.segs = segs.data();
		hp.nsegs = NH;
		hp.hdesc = hdesc.data();
		hp.htbl = htbl.data();
		hp.hlst = hlst.data();
		hp.cand = cands.data();
		hp.hll32 = hll32.data();
		hp.td_cur = td_cur.data();
		hp.td_pend = td_pend.data();
		hp.pcap = pcap;
		hp.td_run = td_run.data();
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
#ifdef KEMU_SPLIT
		std::vector<gys_resp_seg> vsegs;
		for (uint32_t h = 0; h < NH; ++h) {
			const uint64_t first = segs[h].first_event, len = (h + 1 < NH ? segs[h + 1].first_event : n) - first;
			for (uint64_t part = 0; part * GYS_SPLIT_PART < len; ++part) vsegs.push_back(gys_resp_seg{h, 0u, first + part * GYS_SPLIT_PART});
		}
		hp.segs = vsegs.data();
		hp.nsegs = (uint32_t)vsegs.size();
#if KEMU_MODE == 2
		if (v6) kemu::launch((uint32_t)vsegs.size(), T, dyn, [&] { k_resp_host<KEMU_TPT, true, false, true, 2>(hp); });
		else
#endif
			kemu::launch((uint32_t)vsegs.size(), T, dyn, [&] { k_resp_host<KEMU_TPT, true, false, false, KEMU_MODE ? 1 : 0>(hp); });
		kemu::launch((nsvc + 255u) / 256u, 256, 0, [&] { k_key_finalize(fin); });
#else
#if KEMU_MODE == 2
		if (v6) kemu::launch(NH, T, dyn, [&] { k_resp_host<KEMU_TPT, false, false, true, 2>(hp); });
		else
#endif
			kemu::launch(NH, T, dyn, [&] { k_resp_host<KEMU_TPT, false, false, false, KEMU_MODE ? 1 : 0>(hp); });
#endif
		CHECK(counts[FIN_HUGE] == 0 && counts[FIN_RUN_ALLOC] == 0, "batch %u: huge %u run words %u (the test keeps every key below 4 096 values and inside its buffer)", batch,
		      counts[FIN_HUGE], counts[FIN_RUN_ALLOC]);

		// ---- the queued merges
		const uint32_t nmerge = counts[FIN_CLASS0];
		std::vector<uint32_t> merged;
		for (uint32_t i = 0; i < nmerge; ++i) merged.push_back(list0[i].slot);
		merges_seen += nmerge;
		MergeBP q{};
		q.d.td_sum = td_sum.data();
		q.d.td_cnt = td_cnt.data();
		q.d.td_meta = meta.data();
		q.d.td_minmax = minmax.data();
		q.d.td_pend = td_pend.data();
		q.d.td_cur = td_cur.data();
		q.d.pcap = pcap;
		q.d.pend_cap = KEMU_PEND_CAP;
		q.d.nsvc = nsvc;
		q.d.staged = staged.data();
		q.d.hist_win = hist_win.data();
		q.d.hist_all = hist_all.data();
		q.d.bitmap = bitmap.data();
		q.list = list0.data();
		q.count = &counts[FIN_CLASS0];
		q.slow_list = slow.data();
		q.slow_count = &counts[FIN_SLOW];
#if defined(KEMU_BINS_TEMPLATE_NT)
		kemu::launch(2, KEMU_BINS_NT, 0, [&] { k_digest_bins<false, KEMU_BINS_NT>(q); });
#else
		kemu::launch(2, KEMU_BINS_NT, 0, [&] { k_digest_bins<false, KEMU_VPT>(q); });
#endif
		CHECK(counts[FIN_SLOW] == 0, "hand-over list not empty");
		if (counts[FIN_CLASS1]) { // keys whose batch took them past the fast class: the cluster-gap kernel
			for (uint32_t i = 0; i < counts[FIN_CLASS1]; ++i) merged.push_back(list1[i].slot);
			merges_seen += counts[FIN_CLASS1];
			MergeP mp{};
			mp.d = q.d;
			mp.list = list1.data();
			mp.count = &counts[FIN_CLASS1];
#if defined(KEMU_BINS_TEMPLATE_NT)
			q.list = list1.data(); // (the tree under test merges class 1 with the 1024-thread instance of the value-bin kernel)
			q.count = &counts[FIN_CLASS1];
			kemu::launch(2, 1024, 0, [&] { k_digest_bins<false, 1024u>(q); });
			CHECK(counts[FIN_SLOW] == 0, "hand-over list not empty");
#else
			kemu::launch(2, 256, 0, [&] { k_digest_merge<GYS_MERGE_CLASS1, 256u>(mp); });
#endif
		}

		// ---- compare with the oracle after the batch
		const uint64_t *oc = gyo_engine_counters(orc);
		CHECK(counters[CTR_RESP_EVENTS] == oc[0] && counters[CTR_RESP_DROP_RANGE] == oc[1] && counters[CTR_RESP_DROP_NOLISTENER] == oc[2],
		      "batch %u counters {%llu, %llu, %llu} want {%llu, %llu, %llu}", batch, (unsigned long long)counters[CTR_RESP_EVENTS], (unsigned long long)counters[CTR_RESP_DROP_RANGE],
		      (unsigned long long)counters[CTR_RESP_DROP_NOLISTENER], (unsigned long long)oc[0], (unsigned long long)oc[1], (unsigned long long)oc[2]);
		const uint8_t *ohll = gyo_engine_hll(orc);
		for (uint32_t i = 0; i < (1u << GYS_HLL_P); ++i) CHECK(hll32[i] == ohll[i], "batch %u HLL register %u: %u want %u", batch, i, hll32[i], ohll[i]);
		const gyo_hist_serial *og = gyo_engine_ghist(orc);
		for (int b = 0; b < 15; ++b)
			CHECK(ghist[2 * b] == og[b].count && (int64_t)ghist[2 * b + 1] == og[b].sum, "batch %u all-service bucket %d: {%llu, %lld} want {%llu, %lld}", batch, b, ghist[2 * b],
			      (long long)ghist[2 * b + 1], (unsigned long long)og[b].count, (long long)og[b].sum);
		CHECK(ghist[30] == og[15].count && gmax == gyo_engine_gmax(orc), "batch %u all-service total %llu / max %lld", batch, ghist[30], gmax);
		const gyo_hist_serial *oh = gyo_engine_hist(orc);
		for (uint32_t s = 0; s < nsvc; ++s) {
			const gyo_td_buffered *ot = gyo_engine_td(orc, s);
			CHECK(meta[s].npend == ot->npend && (td_cur[s] & ~GYS_SPILL_BIT) == ot->npend, "batch %u key %u buffered %u (cur %u) want %u", batch, s, meta[s].npend, td_cur[s], ot->npend);
			if (meta[s].npend == ot->npend) {
				std::vector<int32_t> a(ot->npend), b(gyo_tdb_values(ot), gyo_tdb_values(ot) + ot->npend);
				for (uint32_t i = 0; i < ot->npend; ++i) a[i] = (int32_t)(td_pend[(size_t)s * pcap + i] >> GYS_ROW_BITS);
				std::sort(a.begin(), a.end());
				std::sort(b.begin(), b.end());
				CHECK(a == b, "batch %u key %u: buffered values differ", batch, s);
			}
			for (int j = 0; j < GYS_TD_NB; ++j)
				CHECK(td_sum[(size_t)s * GYS_TD_NB + j] == ot->d.sum[j] && td_cnt[(size_t)s * GYS_TD_NB + j] == ot->d.cnt[j], "batch %u key %u cluster %d: {%lld, %u} want {%lld, %u}",
				      batch, s, j, (long long)td_sum[(size_t)s * GYS_TD_NB + j], td_cnt[(size_t)s * GYS_TD_NB + j], (long long)ot->d.sum[j], ot->d.cnt[j]);
		}
		for (uint32_t s : merged) { // a key that has just been re-clustered has all its values folded: its all-time record is complete
			for (int b = 0; b < 15; ++b)
				CHECK(hist_all[s].stats[b].count == oh[(size_t)s * 16 + b].count && hist_all[s].stats[b].sum == oh[(size_t)s * 16 + b].sum, "batch %u key %u all-time bucket %d", batch, s, b);
			CHECK(hist_all[s].total_count == oh[(size_t)s * 16 + 15].count && hist_all[s].max_val_seen == oh[(size_t)s * 16 + 15].sum, "batch %u key %u all-time total / max", batch, s);
			const gyo_td_buffered *ot = gyo_engine_td(orc, s);
			CHECK(minmax[s].x == ot->d.vmin && minmax[s].y == ot->d.vmax, "batch %u key %u min/max {%d, %d} want {%d, %d}", batch, s, minmax[s].x, minmax[s].y, ot->d.vmin, ot->d.vmax);
			CHECK(meta[s].hw_epoch == epoch && hist_win[s].total_count > 0 && hist_win[s].total_count <= hist_all[s].total_count, "batch %u key %u window record", batch, s);
			{ // the window's record and the CONN_BITMAP rows of both families (the key's buffer is drained: everything is folded)
				const gyo_hist_serial *ow = gyo_engine_hist(orcw) + (size_t)s * 16;
				const uint16_t *ob = gyo_engine_bitmap(orcw) + (size_t)s * 64;
				for (int b = 0; b < 15; ++b)
					CHECK(hist_win[s].stats[b].count == ow[b].count && hist_win[s].stats[b].sum == ow[b].sum, "batch %u key %u window bucket %d", batch, s, b);
				for (uint32_t g = 0; g < GYS_BM_WORDS; ++g) {
					const uint32_t want = (uint32_t)ob[2 * g] | ((uint32_t)ob[2 * g + 1] << 16);
					CHECK(bitmap[(size_t)s * GYS_BM_WORDS + g] == want, "batch %u key %u CONN_BITMAP rows %u, %u: %08x want %08x", batch, s, 2 * g, 2 * g + 1, bitmap[(size_t)s * GYS_BM_WORDS + g], want);
				}
			}
		}
		if (batch % 3 == 2) std::fill(resp_win.begin(), resp_win.end(), 0u); // (the window boundary clears the per-window event counts)
	}
	CHECK(merges_seen >= 100, "only %llu merges were exercised", (unsigned long long)merges_seen);
	gyo_engine_free(orc);
	gyo_engine_free(orcw);
	if (fails) {
		printf("%d checks failed\n", fails);
		return 1;
	}
	printf("kemu resp ok (%u-event tiles, %llu merges)\n", TILE, (unsigned long long)merges_seen);
	return 0;
}

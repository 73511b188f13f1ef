// TEST INFRASTRUCTURE (CPU): the LOGIC of the hot t-digest merge kernel (k_digest_bins, gyeeta_amd/csrc/gys_kernels.hpp) run under the
// CPU stand-in of the device model (tests/cpp/kemu/hip/hip_runtime.h) and compared with the oracle: the re-clustered digest
// (gyo_td_merge_values), the lazily folded records (GY_HISTOGRAM::add_data per value, window roll, CONN_BITMAP rows, min / max), the
// drained meta record -- for keys with no clusters yet, with old clusters, with large values (>= 1024 ms: the comparison phase), with
// part of the buffer already folded / belonging to an earlier window, with values arriving through a run in `staged` (spilled key),
// and the hand-over of an entry whose weight needs 64-bit arithmetic.  Build + run: tests/test_kernel_logic_cpu.py.
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+r"(a[0]), "+r"(a[1]), "+r"(a[2]), "+r"(a[3]))
#define GYS_DYN_LDS(type, name) type *name = (type *)kemu::dyn_lds()
#include "../../../gyeeta_amd/csrc/gys_kernels.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <random>

#include "../../../oracle/gy_oracle.h"

#ifndef KEMU_BINS_NT
#define KEMU_BINS_NT 256
#endif

using namespace gys;

namespace {

struct Key {
	gyo_tdigest d;             // oracle digest before the merge
	std::vector<int32_t> vals; // buffered (then run) values, in buffer order
	std::vector<uint32_t> rows;
	uint32_t nbuf, mrun, nh, nw, win_epoch, hw_epoch;
	gyo_hist all0, win0; // records before
	uint32_t bm0[GYS_BM_WORDS];
	int32_t mn0, mx0;
};

int fails = 0;
#define CHECK(c, ...)                                  \
	do {                                           \
		if (!(c)) {                            \
			if (fails++ < 20) {            \
				printf("FAIL %s:%d: ", __FILE__, __LINE__); \
				printf(__VA_ARGS__);   \
				printf("\n");          \
			}                              \
		}                                      \
	} while (0)

int32_t draw(std::mt19937 &rng, double mu, double sigma)
{
	std::lognormal_distribution<double> ln(mu, sigma);
	double v = std::floor(ln(rng));
	if (v > 1000000.0) v = 1000000.0;
	return (int32_t)v;
}

} // namespace

int main(int argc, char **argv)
{
	const uint32_t NT = KEMU_BINS_NT;
	if (!kemu::can_run(NT)) {
		printf("kemu: this process cannot have %u threads\n", NT);
		return 77;
	}
#ifndef KEMU_BINS_VPT
#define KEMU_BINS_VPT 4
#endif
	const uint32_t CAP = (uint32_t)KEMU_BINS_VPT * NT; // values one merge of this instance takes (VPT per thread: 4 = the default buffer, 8 / 16 = td_pend_cap up to 1920 / 3968)
	const uint32_t pcap = CAP + 64u;
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 12345u);
	const uint32_t S = 14;
	std::vector<Key> keys(S);
	std::vector<int64_t> td_sum((size_t)S * GYS_TD_NB, 0);
	std::vector<uint32_t> td_cnt((size_t)S * GYS_TD_NB, 0), td_pend((size_t)S * pcap, 0xDEADBEEFu), td_cur(S, 0), staged(1u << 16, 0), bitmap((size_t)S * GYS_BM_WORDS, 0);
	std::vector<TdMeta> meta(S);
	std::vector<int2> minmax(S);
	std::vector<gys_hist_rec> hist_all(S), hist_win(S);
	std::vector<MergeEnt> list(S), slow(S + 1);
	uint32_t count = S, slow_count = 0, staged_used = 0;

	for (uint32_t s = 0; s < S; ++s) {
		Key &k = keys[s];
		gyo_td_init(&k.d);
		gyo_hist_init(&k.all0, GYO_RESP_TIME_HASH);
		gyo_hist_init(&k.win0, GYO_RESP_TIME_HASH);
		memset(k.bm0, 0, sizeof(k.bm0));
		k.mn0 = INT32_MAX;
		k.mx0 = INT32_MIN;
		// the key's history: merged values (-> clusters; already folded into the records) -- none for s % 4 == 0
		const double mu = 1.5 + 0.45 * s, sigma = s % 3 == 2 ? 2.2 : 1.0; // (mu up to 7.3: most values >= 1024 ms for the last keys)
		const uint32_t nold = s % 4 == 0 ? 0u : 300u + 977u * s;
		if (s == 13) { // total weight beyond 2^31: handed over to the general kernel, untouched
			for (int j = 0; j < GYO_TD_NB; ++j) {
				k.d.cnt[j] = 11000000u;
				k.d.sum[j] = (int64_t)k.d.cnt[j] * (10 + j);
			}
			k.d.vmin = 10;
			k.d.vmax = 10 + GYO_TD_NB;
		} else if (nold) {
			std::vector<int32_t> old(nold);
			for (auto &v : old) {
				v = draw(rng, mu, sigma);
				gyo_hist_add(&k.all0, v);
				k.mn0 = std::min(k.mn0, v);
				k.mx0 = std::max(k.mx0, v);
			}
			gyo_td_merge_values(&k.d, old.data(), old.size());
		}
		// the buffer (and, for spilled keys, a run)
		uint32_t m = s == 1 ? 1u : s == 2 ? CAP : s == 3 ? CAP - 1u : (uint32_t)(rng() % (CAP - 10u)) + 5u;
		k.mrun = (s == 5 || s == 9) ? m / 3u : 0u;
		k.nbuf = m - k.mrun;
		k.vals.resize(m);
		k.rows.resize(m);
		for (uint32_t i = 0; i < m; ++i) {
			k.vals[i] = s == 6 ? 37 : draw(rng, mu, sigma); // key 6: all values equal (every value ties with every other)
			k.rows[i] = (uint32_t)(rng() & (s % 3 == 0 ? 31u : 63u)); // (rows 32..63: the IPv6 family's rows; every third key has IPv4 events only)
		}
		// fold state of the buffered words: [0, nh) already folded, [0, nw) arrived in windows before win_epoch
		k.win_epoch = 7;
		k.nh = s % 5 == 1 ? k.nbuf / 2u : 0u;
		k.nw = s % 5 == 2 ? k.nbuf / 3u : (s % 5 == 1 ? k.nbuf / 4u : 0u);
		k.hw_epoch = s % 2 ? 7u : 5u; // 5: the window record is from an older window (rolls when window values are folded)
		for (uint32_t i = 0; i < k.nh; ++i) { // what "already folded" means for the records before the merge
			gyo_hist_add(&k.all0, k.vals[i]);
			k.mn0 = std::min(k.mn0, k.vals[i]);
			k.mx0 = std::max(k.mx0, k.vals[i]);
			if (i >= k.nw && k.hw_epoch == k.win_epoch) {
				const uint32_t b = gyo_hist_add(&k.win0, k.vals[i]);
				k.bm0[k.rows[i] >> 1] |= (1u << b) << ((k.rows[i] & 1u) * 16u);
			}
		}
		if (k.hw_epoch != k.win_epoch) { // some older window's record sits there
			gyo_hist_add(&k.win0, 123);
			k.bm0[3] = 0x00010002u;
			k.bm0[21] = 0x00400001u; // (IPv6 rows of the older window: a roll has to clear them too)
		}
		// device-side images
		for (int j = 0; j < GYO_TD_NB; ++j) {
			td_sum[(size_t)s * GYS_TD_NB + j] = k.d.sum[j];
			td_cnt[(size_t)s * GYS_TD_NB + j] = k.d.cnt[j];
		}
		for (uint32_t i = 0; i < k.nbuf; ++i) td_pend[(size_t)s * pcap + i] = ((uint32_t)k.vals[i] << GYS_ROW_BITS) | k.rows[i];
		uint32_t off_end = 0;
		if (k.mrun) {
			for (uint32_t i = 0; i < k.mrun; ++i) staged[staged_used + i] = ((uint32_t)k.vals[k.nbuf + i] << GYS_ROW_BITS) | k.rows[k.nbuf + i];
			staged_used += k.mrun;
			off_end = staged_used;
		}
		meta[s].npend = k.nbuf;
		meta[s].nh = (uint16_t)k.nh;
		meta[s].nw = (uint16_t)k.nw;
		meta[s].win_epoch = k.win_epoch;
		meta[s].hw_epoch = k.hw_epoch;
		td_cur[s] = k.nbuf;
		minmax[s] = make_int2(k.mn0, k.mx0);
		for (int b = 0; b < 15; ++b) {
			hist_all[s].stats[b].count = k.all0.stats[b].count;
			hist_all[s].stats[b].sum = k.all0.stats[b].sum;
			hist_win[s].stats[b].count = k.win0.stats[b].count;
			hist_win[s].stats[b].sum = k.win0.stats[b].sum;
		}
		hist_all[s].total_count = k.all0.total_count;
		hist_all[s].max_val_seen = k.all0.total_count ? k.all0.max_val_seen : INT64_MIN;
		hist_win[s].total_count = k.win0.total_count;
		hist_win[s].max_val_seen = k.win0.total_count ? k.win0.max_val_seen : INT64_MIN;
		memcpy(&bitmap[(size_t)s * GYS_BM_WORDS], k.bm0, sizeof(k.bm0));
		list[s] = MergeEnt{s, k.nbuf, k.mrun, off_end};
	}

	MergeBP q{};
	q.d.td_sum = td_sum.data();
	q.d.td_cnt = td_cnt.data();
	q.d.td_meta = meta.data();
	q.d.td_minmax = minmax.data();
	q.d.td_pend = td_pend.data();
	q.d.td_cur = td_cur.data();
	q.d.pcap = pcap;
	q.d.pend_cap = CAP - 128u;
	q.d.nsvc = S;
	q.d.staged = staged.data();
	q.d.hist_win = hist_win.data();
	q.d.hist_all = hist_all.data();
	q.d.bitmap = bitmap.data();
	q.list = list.data();
	q.count = &count;
	q.slow_list = slow.data();
	q.slow_count = &slow_count;

#if KEMU_BINS_NT == 256 && !defined(KEMU_BINS_TEMPLATE_NT)
	{ // the all-service scan first (it modifies nothing): quantiles of every key's digest with its buffer merged in (a run is not part of the
	  // view between batches), equal to the oracle's; keys with 64-bit weights or more large values than the list holds are handed over
		const double qs[4] = {0.25, 0.5, 0.95, 0.99};
		std::vector<double> qout((size_t)S * 4, -1.0);
		std::vector<MergeEnt> sslow(S + 1);
		uint32_t sslow_count = 0;
		MergeBP sq = q;
		sq.qs = qs;
		sq.nq = 4;
		sq.qout = qout.data();
		sq.slow_list = sslow.data();
		sq.slow_count = &sslow_count;
		kemu::launch(3, NT, 0, [&] { k_digest_bins<true, KEMU_BINS_VPT>(sq); });
		uint32_t want = 0;
		for (uint32_t s = 0; s < S; ++s) {
			const Key &k = keys[s];
			const uint32_t m = std::min(k.nbuf, q.d.pend_cap);
			uint32_t nbig = 0;
			for (uint32_t i = 0; i < m; ++i) nbig += k.vals[i] >= (int32_t)GYS_MB_EXACT ? 1u : 0u;
			const bool over = s == 13 || (KEMU_BINS_VPT > 4 && nbig > GYS_MB_BIG_CAP);
			want += over ? 1u : 0u;
			bool listed = false;
			for (uint32_t i = 0; i < sslow_count; ++i) listed = listed || sslow[i].slot == s;
			CHECK(listed == over, "scan: key %u %s handed over (large values %u)", s, listed ? "was" : "was not", nbig);
			if (over) continue;
			gyo_tdigest d = k.d;
			d.vmin = k.mn0; // (the engine's extremes cover merged and folded values; the not yet folded ones come from the scan itself)
			d.vmax = k.mx0;
			if (!gyo_td_total(&d) && !k.nh) {
				d.vmin = INT32_MAX;
				d.vmax = INT32_MIN;
			}
			gyo_td_merge_values(&d, k.vals.data(), m);
			for (int i = 0; i < 4; ++i) CHECK(qout[(size_t)s * 4 + i] == gyo_td_quantile(&d, qs[i]), "scan: key %u q %.2f: %.1f, oracle %.1f", s, qs[i], qout[(size_t)s * 4 + i], gyo_td_quantile(&d, qs[i]));
		}
		CHECK(sslow_count == want, "scan: %u keys handed over, want %u", sslow_count, want);
		printf("scan of %u keys; %u handed over\n", S, want);
	}
	kemu::launch(3, NT, 0, [&] { k_digest_bins<false, KEMU_BINS_VPT>(q); });
#else
	kemu::launch(3, NT, 0, [&] { k_digest_bins<false, KEMU_BINS_NT>(q); });
#endif

	// handed over to the general kernel, untouched: key 13 (64-bit weights) and, in the instances whose merges can carry more large values
	// (>= 1024 ms) than the kernel's list holds, the keys that do
	uint32_t want_slow = 0;
	std::vector<bool> handed(S, false);
	for (uint32_t s = 0; s < S; ++s) {
		uint32_t nbig = 0;
		for (int32_t v : keys[s].vals) nbig += v >= (int32_t)GYS_MB_EXACT ? 1u : 0u;
		handed[s] = s == 13 || (KEMU_BINS_VPT > 4 && nbig > GYS_MB_BIG_CAP);
		want_slow += handed[s] ? 1u : 0u;
	}
	CHECK(slow_count == want_slow, "hand-over list: %u entries, want %u", slow_count, want_slow);
	for (uint32_t i = 0; i < slow_count; ++i) CHECK(slow[i].slot < S && handed[slow[i].slot], "key %u was handed over", slow[i].slot);
	printf("merges of up to %u values; %u keys handed over\n", CAP, want_slow);
	for (uint32_t s = 0; s < S; ++s) {
		const Key &k = keys[s];
		const uint32_t m = k.nbuf + k.mrun;
		if (s == 13) { // untouched
			CHECK(meta[s].npend == k.nbuf && td_cnt[(size_t)s * GYS_TD_NB] == 11000000u, "key 13 was modified");
			continue;
		}
		if (handed[s]) {
			CHECK(meta[s].npend == k.nbuf, "key %u (handed over) was modified", s);
			continue;
		}
		gyo_tdigest d = k.d;
		gyo_td_merge_values(&d, k.vals.data(), m);
		for (int j = 0; j < GYO_TD_NB; ++j)
			CHECK(td_sum[(size_t)s * GYS_TD_NB + j] == d.sum[j] && td_cnt[(size_t)s * GYS_TD_NB + j] == d.cnt[j], "key %u (m %u) cluster %d: got {%lld, %u} want {%lld, %u}", s, m,
			      j, (long long)td_sum[(size_t)s * GYS_TD_NB + j], td_cnt[(size_t)s * GYS_TD_NB + j], (long long)d.sum[j], d.cnt[j]);
		// records: every not yet folded value into the all-time record; the values of window win_epoch into the window record (rolled first
		// when it belonged to an older window) and into the CONN_BITMAP rows
		gyo_hist all = k.all0, win = k.win0;
		uint32_t bm[GYS_BM_WORDS];
		memcpy(bm, k.bm0, sizeof(bm));
		int32_t mn = k.mn0, mx = k.mx0;
		const uint32_t nwin0 = std::max(k.nh, k.nw);
		const bool any_win = m > nwin0, roll = k.hw_epoch != k.win_epoch;
		if (any_win && roll) {
			gyo_hist_init(&win, GYO_RESP_TIME_HASH);
			memset(bm, 0, sizeof(bm));
		}
		for (uint32_t i = k.nh; i < m; ++i) {
			gyo_hist_add(&all, k.vals[i]);
			mn = std::min(mn, k.vals[i]);
			mx = std::max(mx, k.vals[i]);
			if (i >= nwin0) {
				const uint32_t b = gyo_hist_add(&win, k.vals[i]);
				bm[k.rows[i] >> 1] |= (1u << b) << ((k.rows[i] & 1u) * 16u);
			}
		}
		for (int b = 0; b < 15; ++b) {
			CHECK(hist_all[s].stats[b].count == all.stats[b].count && hist_all[s].stats[b].sum == all.stats[b].sum, "key %u all-time bucket %d: {%llu, %lld} want {%llu, %lld}", s, b,
			      (unsigned long long)hist_all[s].stats[b].count, (long long)hist_all[s].stats[b].sum, (unsigned long long)all.stats[b].count, (long long)all.stats[b].sum);
			CHECK(hist_win[s].stats[b].count == win.stats[b].count && hist_win[s].stats[b].sum == win.stats[b].sum, "key %u window bucket %d: {%llu, %lld} want {%llu, %lld}", s, b,
			      (unsigned long long)hist_win[s].stats[b].count, (long long)hist_win[s].stats[b].sum, (unsigned long long)win.stats[b].count, (long long)win.stats[b].sum);
		}
		CHECK(hist_all[s].total_count == all.total_count, "key %u all-time total %llu want %llu", s, (unsigned long long)hist_all[s].total_count, (unsigned long long)all.total_count);
		if (all.total_count) CHECK(hist_all[s].max_val_seen == all.max_val_seen, "key %u all-time max %lld want %lld", s, (long long)hist_all[s].max_val_seen, (long long)all.max_val_seen);
		CHECK(hist_win[s].total_count == win.total_count, "key %u window total %llu want %llu", s, (unsigned long long)hist_win[s].total_count, (unsigned long long)win.total_count);
		if (any_win) CHECK(hist_win[s].max_val_seen == win.max_val_seen, "key %u window max %lld want %lld", s, (long long)hist_win[s].max_val_seen, (long long)win.max_val_seen);
		for (int g = 0; g < (int)GYS_BM_WORDS; ++g) CHECK(bitmap[(size_t)s * GYS_BM_WORDS + g] == bm[g], "key %u bitmap word %d: %08x want %08x", s, g, bitmap[(size_t)s * GYS_BM_WORDS + g], bm[g]);
		CHECK(minmax[s].x == mn && minmax[s].y == mx, "key %u min/max {%d, %d} want {%d, %d}", s, minmax[s].x, minmax[s].y, mn, mx);
		// the buffer is drained; the window bookkeeping moves on
		CHECK(meta[s].npend == 0 && meta[s].nh == 0 && meta[s].nw == 0 && meta[s].win_epoch == k.win_epoch && meta[s].hw_epoch == (any_win ? k.win_epoch : k.hw_epoch) && td_cur[s] == 0,
		      "key %u meta {%u, %u, %u, %u, %u} cur %u", s, meta[s].npend, meta[s].nh, meta[s].nw, meta[s].win_epoch, meta[s].hw_epoch, td_cur[s]);
	}
	if (fails) {
		printf("%d checks failed\n", fails);
		return 1;
	}
	printf("kemu bins ok (%u threads per workgroup, merges of up to %u values)\n", NT, CAP);
	return 0;
}

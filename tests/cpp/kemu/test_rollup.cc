// TEST INFRASTRUCTURE (CPU): the LOGIC of the roll-up digests k_rollup_accum / k_rollup_cluster (gyeeta_amd/csrc/gys_rollup.hpp) under the CPU
// stand-in of the device model: the digest of a GROUP of services (kind 0: the union by value bin of the members' clusters and buffered values)
// and of a group of roll-up slabs (kind 1: the cross-rank / cluster / global roll-up), with 64-bit counters -- groups of 0, 1 and many members,
// members without clusters, without buffered values and without anything, all-equal values, values >= 1024 ms, a group whose weight passes
// 2^32, groups cut into several chunks (several workgroups add to one group's bins), a buffer stride that is not a multiple of four words
// (argument 2) -- equal, cluster by cluster, to the oracle's gyo_tdbins_* (oracle/gy_oracle_rollup.c), minimum / maximum included, and
// independent of the order of the members.  Build + run: tests/test_kernel_logic_cpu.py.
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+r"(a[0]), "+r"(a[1]), "+r"(a[2]), "+r"(a[3]))
#define GYS_DYN_LDS(type, name) type *name = (type *)kemu::dyn_lds()
#include "../../../gyeeta_amd/csrc/gys_kernels.hpp"
#include "../../../gyeeta_amd/csrc/gys_rollup.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

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
int32_t draw(std::mt19937 &rng, double mu, double sigma)
{
	std::normal_distribution<double> n(mu, sigma);
	const double v = floor(exp(n(rng)));
	return (int32_t)(v < 0 ? 0 : v > 1000000.0 ? 1000000.0 : v);
}
void compare(const char *what, uint32_t g, const gys_tdigest_slab &got, const gyo_td64 &want)
{
	uint64_t tot = 0;
	for (int j = 0; j < GYO_TD_NB; ++j) {
		CHECK(got.cnt[j] == want.cnt[j] && got.sum[j] == want.sum[j], "%s group %u cluster %d: {%llu, %lld}, oracle {%llu, %lld}", what, g, j, (unsigned long long)got.cnt[j],
		      (long long)got.sum[j], (unsigned long long)want.cnt[j], (long long)want.sum[j]);
		tot += want.cnt[j];
	}
	if (tot) CHECK(got.vmin == want.vmin && got.vmax == want.vmax, "%s group %u: min / max %lld %lld, oracle %lld %lld", what, g, (long long)got.vmin, (long long)got.vmax,
		       (long long)want.vmin, (long long)want.vmax);
}
} // namespace

int main(int argc, char **argv)
{
	if (!kemu::can_run(GYS_RB_NT)) {
		printf("kemu: this process cannot have 1024 threads\n");
		return 77;
	}
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 9u);
	const uint32_t S = 36, pcap = GYS_TD_PEND_CAP + 64u + (argc > 2 ? (uint32_t)atoi(argv[2]) : 0u);
	std::vector<gyo_td_buffered> svc(S);
	std::vector<int64_t> td_sum((size_t)S * GYS_TD_NB, 0);
	std::vector<uint32_t> td_cnt((size_t)S * GYS_TD_NB, 0), td_pend((size_t)S * pcap, 0xDEADBEEFu);
	std::vector<TdMeta> meta(S);
	std::vector<int2> minmax(S);
	for (uint32_t s = 0; s < S; ++s) {
		gyo_td_buffered &b = svc[s];
		gyo_tdb_init(&b);
		const double mu = 1.0 + 0.2 * s, sigma = s % 3 == 2 ? 2.0 : 0.9; // (the last services: most values >= 1024 ms)
		if (s % 7 == 3) {
			// nothing at all
		} else if (s % 7 == 5) { // huge weights: three of these in a group pass 2^32
			for (int j = 0; j < GYO_TD_NB; ++j) {
				b.d.cnt[j] = 9000000u + 1000u * s;
				b.d.sum[j] = (int64_t)b.d.cnt[j] * (5 + 3 * j + (int)s);
			}
			b.d.vmin = 5;
			b.d.vmax = 5 + 3 * GYO_TD_NB + (int)s;
		} else {
			const uint32_t nbatches = s % 7 == 0 ? 1u : 2u + rng() % 6u; // (one small batch: buffered values only, no clusters yet)
			for (uint32_t k = 0; k < nbatches; ++k) {
				std::vector<int32_t> v(s % 7 == 0 ? 40u : 100u + rng() % 700u);
				for (auto &x : v) x = s == 8 ? 37 : draw(rng, mu, sigma); // service 8: all values equal
				gyo_tdb_add_batch(&b, v.data(), v.size());
			}
			if (s % 7 == 1) { // no buffered values: everything merged
				gyo_tdigest mv;
				gyo_tdb_merged_view(&b, &mv);
				b.d = mv;
				b.npend = 0;
			}
		}
		for (int j = 0; j < GYO_TD_NB; ++j) {
			td_sum[(size_t)s * GYS_TD_NB + j] = b.d.sum[j];
			td_cnt[(size_t)s * GYS_TD_NB + j] = b.d.cnt[j];
		}
		for (uint32_t i = 0; i < b.npend; ++i) td_pend[(size_t)s * pcap + i] = ((uint32_t)b.pend[i] << GYS_ROW_BITS) | (rng() & 31u);
		meta[s] = TdMeta{};
		meta[s].npend = b.npend;
		minmax[s] = gyo_td_total(&b.d) ? make_int2(b.d.vmin, b.d.vmax) : make_int2(INT32_MAX, INT32_MIN);
	}
	// groups of services: empty, one member, a few, many, one with the three heavy services (5, 12, 19, 26, 33 are heavy)
	std::vector<std::vector<uint32_t>> groups = {{}, {4}, {3}, {0, 1, 2}, {5, 12, 19, 26}, {8, 8, 9}, {}};
	{
		std::vector<uint32_t> all(S);
		for (uint32_t s = 0; s < S; ++s) all[s] = s;
		groups.push_back(all);
		std::vector<uint32_t> rev(all.rbegin(), all.rend()); // (the same digest: the union does not depend on the order)
		groups.push_back(rev);
	}
	std::vector<uint32_t> off(1, 0), members;
	for (auto &g : groups) {
		members.insert(members.end(), g.begin(), g.end());
		off.push_back((uint32_t)members.size());
	}
	if (members.empty()) members.push_back(0);
	const uint32_t NG = (uint32_t)groups.size();
	auto make_chunks = [](const std::vector<uint32_t> &o, uint32_t per) {
		std::vector<RollupChunk> ch;
		for (uint32_t g = 0; g + 1 < o.size(); ++g)
			for (uint32_t m = o[g]; m < o[g + 1]; m += per) ch.push_back(RollupChunk{g, m, std::min(o[g + 1], m + per), 0u});
		if (ch.empty()) ch.push_back(RollupChunk{0, 0, 0, 0});
		return ch;
	};
	std::vector<gyo_td64> want(NG);
	uint64_t heavy = 0;
	for (uint32_t g = 0; g < NG; ++g) {
		gyo_td_bins *b = new gyo_td_bins;
		gyo_tdbins_init(b);
		for (uint32_t s : groups[g]) gyo_tdbins_add_service(b, &svc[s]);
		gyo_tdbins_finish(b, &want[g]);
		delete b;
		heavy = std::max(heavy, gyo_td64_total(&want[g]));
	}
	CHECK(heavy > (1ull << 32), "no group passed 2^32 (%llu)", (unsigned long long)heavy);
	CHECK(memcmp(&want[NG - 1], &want[NG - 2], sizeof(gyo_td64)) == 0, "the oracle's roll-up depends on the order of the members");
	std::vector<gys_tdigest_slab> slabs(NG);
	for (uint32_t per : {1024u, 5u}) { // one chunk per group; chunks of five members: up to eight workgroups add to one group's bins
		std::vector<RollupChunk> chunks = make_chunks(off, per);
		std::vector<unsigned long long> bins((size_t)NG * GYS_RB_STRIDE, 0xABABABABABABABABull);
		memset(slabs.data(), 0xAB, sizeof(gys_tdigest_slab) * NG);
		RollupP q{};
		q.d.td_sum = td_sum.data();
		q.d.td_cnt = td_cnt.data();
		q.d.td_meta = meta.data();
		q.d.td_minmax = minmax.data();
		q.d.td_pend = td_pend.data();
		q.d.pcap = pcap;
		q.d.pend_cap = GYS_TD_PEND_CAP;
		q.d.nsvc = S;
		q.chunks = chunks.data();
		q.nchunks = (uint32_t)chunks.size();
		q.members = members.data();
		q.kind = 0;
		q.bins = bins.data();
		q.out = slabs.data();
		q.ngroups = NG;
		kemu::launch(2, 256, 0, [&] { k_rollup_init(q.bins, NG); });
		kemu::launch(3, GYS_RB_NT, 0, [&] { k_rollup_accum(q); }); // (fewer workgroups than chunks: they loop)
		kemu::launch(2, 256, 0, [&] { k_rollup_cluster(q); });
		for (uint32_t g = 0; g < NG; ++g) compare(per == 5u ? "services in chunks of 5" : "services", g, slabs[g], want[g]);
	}
	// groups of slabs (the cross-rank roll-up): the slabs above, in two orders and with an empty one in the middle; and a caller's slab that no
	// engine makes: weights of 2^40, a mean beyond the value domain (last bin), a sum of zero and a negative one (first bin)
	{
		gys_tdigest_slab odd{};
		gyo_td64 oddw;
		gyo_td64_init(&oddw);
		const uint64_t oc[5] = {1ull << 40, 3ull, (1ull << 33) + 7ull, 5ull, 9ull};
		const int64_t os[5] = {(int64_t)((1ull << 40) * 77ull + 12345ull), (int64_t)(3ull << 30), (int64_t)(((1ull << 33) + 7ull) * 1500ull - 3ull), 0, -40};
		for (int j = 0; j < 5; ++j) {
			odd.cnt[10 * j + 1] = oddw.cnt[10 * j + 1] = oc[j];
			odd.sum[10 * j + 1] = oddw.sum[10 * j + 1] = os[j];
		}
		odd.vmin = oddw.vmin = -40;
		odd.vmax = oddw.vmax = 1ll << 30;
		slabs.push_back(odd);
		want.push_back(oddw);
	}
	std::vector<std::vector<uint32_t>> sg = {{3, 4, 7}, {7, 0, 4, 3}, {0}, {}, {8, 7}, {7, 8}, {NG}, {3, NG, 4}};
	std::vector<uint32_t> soff(1, 0), smem;
	for (auto &g : sg) {
		smem.insert(smem.end(), g.begin(), g.end());
		soff.push_back((uint32_t)smem.size());
	}
	std::vector<gys_tdigest_slab> out2(sg.size());
	for (uint32_t per : {32u, 1u}) {
		std::vector<RollupChunk> chunks = make_chunks(soff, per);
		std::vector<unsigned long long> bins(sg.size() * GYS_RB_STRIDE, 0xABABABABABABABABull);
		memset(out2.data(), 0xAB, sizeof(gys_tdigest_slab) * sg.size());
		RollupP q2{};
		q2.kind = 1;
		q2.in = slabs.data();
		q2.chunks = chunks.data();
		q2.nchunks = (uint32_t)chunks.size();
		q2.members = smem.data();
		q2.bins = bins.data();
		q2.out = out2.data();
		q2.ngroups = (uint32_t)sg.size();
		kemu::launch(1, 256, 0, [&] { k_rollup_init(q2.bins, q2.ngroups); });
		kemu::launch(4, GYS_RB_NT, 0, [&] { k_rollup_accum(q2); });
		kemu::launch((uint32_t)sg.size(), 256, 0, [&] { k_rollup_cluster(q2); });
		for (uint32_t g = 0; g < sg.size(); ++g) {
			gyo_td_bins *b = new gyo_td_bins;
			gyo_td64 w;
			gyo_tdbins_init(b);
			for (uint32_t m : sg[g]) gyo_tdbins_add_td64(b, &want[m]);
			gyo_tdbins_finish(b, &w);
			delete b;
			compare(per == 1u ? "slabs one by one" : "slabs", g, out2[g], w);
		}
	}
	if (fails) {
		printf("kemu rollup: %d FAILURES\n", fails);
		return 1;
	}
	printf("kemu rollup ok: %u groups of services (heaviest %llu values), %zu groups of slabs\n", NG, (unsigned long long)heavy, sg.size());
	return 0;
}

// TEST INFRASTRUCTURE (CPU): the LOGIC of the per-host top-10 selection k_topn_hosts (every thread's sorted list of 10 in LDS, then ten
// rounds of block arg-max "strictly after the previous pick") and of the candidate filter k_topn_filter under the CPU stand-in of the
// device model: hosts with 0, 3, 10, 11, 300 and 2 700 listeners (more than the 2 560 list places), many ties, records of an older
// window and records tagged with another host; all four kinds (LISTEN_TOP_ISSUE / _QPS / _ACTIVE_CONN / _NET: comparators and admission
// thresholds server/gy_msocket.h:720-796, server/gy_mconnhdlr.cc:11260-11304).  Expectation: the admitted records of the host sorted by
// (metric descending, slot ascending), first ten -- computed here with std::sort; the metric values themselves also go through the
// oracle's bounded-heap restatement gyo_topn_u64 (BOUNDED_PRIO_QUEUE, common/gy_statistics.h:356-414; pinned in tests/golden).
// Build + run: tests/test_kernel_logic_cpu.py.
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+r"(a[0]), "+r"(a[1]), "+r"(a[2]), "+r"(a[3]))
#define GYS_DYN_LDS(type, name) type *name = (type *)kemu::dyn_lds()
#include "../../../gyeeta_amd/csrc/gys_kernels.hpp"

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

struct Cand {
	uint64_t m;
	uint32_t s;
};
bool admitted(const uint8_t *rec, uint32_t host, uint32_t epoch, int kind, uint64_t *metric)
{
	uint64_t q[12];
	memcpy(q, rec, 96);
	if ((uint32_t)q[11] != epoch || (uint32_t)(q[11] >> 32) != host) return false;
	const uint32_t nqrys = (uint32_t)q[1], nactive = (uint32_t)(q[2] >> 32), kbin = (uint32_t)(q[4] >> 32), kbout = (uint32_t)q[5], delay = (uint32_t)(q[6] >> 32);
	const uint32_t state = (uint32_t)((q[9] >> 56) & 0xFF);
	switch (kind) {
	case 0: *metric = ((uint64_t)state << 32) | delay; return state > 2u;
	case 1: *metric = nqrys; return nqrys >= 5u;
	case 2: *metric = nactive; return nactive >= 1u;
	default: *metric = (uint64_t)kbin + kbout; return (uint32_t)(kbin + kbout) > 0u;
	}
}
} // namespace

int main(int argc, char **argv)
{
	if (!kemu::can_run(256u)) {
		printf("kemu: this process cannot have 256 threads\n");
		return 77;
	}
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 17u);
	const uint32_t sizes[] = {0, 3, 10, 11, 300, 2700};
	const uint32_t NH = sizeof(sizes) / sizeof(sizes[0]), EPOCH = 7;
	std::vector<uint32_t> off(NH + 1, 0), members;
	uint32_t nsvc = 0;
	for (uint32_t h = 0; h < NH; ++h) nsvc += sizes[h];
	// the services of a host are NOT consecutive slots: a random permutation of all slots is dealt out
	std::vector<uint32_t> perm(nsvc);
	for (uint32_t s = 0; s < nsvc; ++s) perm[s] = s;
	std::shuffle(perm.begin(), perm.end(), rng);
	std::vector<uint32_t> host_of(nsvc);
	for (uint32_t h = 0, k = 0; h < NH; ++h) {
		off[h] = (uint32_t)members.size();
		for (uint32_t j = 0; j < sizes[h]; ++j, ++k) {
			members.push_back(perm[k]);
			host_of[perm[k]] = h;
		}
	}
	off[NH] = (uint32_t)members.size();
	std::vector<uint64_t> state((size_t)nsvc * 12);
	uint8_t *sb = (uint8_t *)state.data();
	for (uint32_t s = 0; s < nsvc; ++s) {
		uint8_t *r = sb + (size_t)s * 96;
		for (int k = 0; k < 88; ++k) r[k] = (uint8_t)rng();
		const uint32_t few = rng() % 8u; // few distinct metric values: ties everywhere
		const uint32_t nq = few * (rng() % 3u), na = rng() % 3u, kin = few * 100u, kout = rng() % 2u ? 0u : 5u, delay = few;
		memcpy(r + 8, &nq, 4);
		memcpy(r + 20, &na, 4);
		memcpy(r + 36, &kin, 4);
		memcpy(r + 40, &kout, 4);
		memcpy(r + 52, &delay, 4);
		r[79] = (uint8_t)(rng() % 6u);
		const uint32_t ep = rng() % 5u ? EPOCH : EPOCH - 1u;                    // a fifth of the records are from the window before
		const uint32_t hh = rng() % 9u ? host_of[s] : (host_of[s] + 1u) % NH; // a ninth carry another host's tag
		const uint64_t tag = (uint64_t)ep | ((uint64_t)hh << 32);
		memcpy(r + 88, &tag, 8);
	}
	uint64_t npicked = 0;
	for (int kind = 0; kind < 4; ++kind) {
		std::vector<uint32_t> out_slot(NH * GYS_TOPN, 0xABABABABu);
		std::vector<uint64_t> out_metric(NH * GYS_TOPN, 0);
		TopnHostsP p{};
		p.svc_state = sb;
		p.off = off.data();
		p.members = members.data();
		p.nhosts = NH;
		p.epoch = EPOCH;
		p.kind = kind;
		p.out_slot = out_slot.data();
		p.out_metric = out_metric.data();
		kemu::launch(kind & 1 ? 2u : NH, 256, 0, [&] { k_topn_hosts(p); }); // (a grid smaller than the host count: the workgroups loop)
		for (uint32_t h = 0; h < NH; ++h) {
			std::vector<Cand> c;
			for (uint32_t i = off[h]; i < off[h + 1]; ++i) {
				uint64_t m;
				if (admitted(sb + (size_t)members[i] * 96, h, EPOCH, kind, &m)) c.push_back(Cand{m, members[i]});
			}
			std::sort(c.begin(), c.end(), [](const Cand &a, const Cand &b) { return a.m != b.m ? a.m > b.m : a.s < b.s; });
			for (uint32_t r = 0; r < GYS_TOPN; ++r) {
				const uint32_t got = out_slot[h * GYS_TOPN + r];
				if (r < c.size()) {
					CHECK(got == c[r].s && out_metric[h * GYS_TOPN + r] == c[r].m, "kind %d host %u place %u: slot %u metric %llu, want %u %llu", kind, h, r, got,
					      (unsigned long long)out_metric[h * GYS_TOPN + r], c[r].s, (unsigned long long)c[r].m);
					++npicked;
				} else
					CHECK(got == GYS_NOSLOT, "kind %d host %u place %u: slot %u, want none", kind, h, r, got);
			}
			// the metric multiset against the oracle's bounded heap (BOUNDED_PRIO_QUEUE keeps the 10 largest; its order among equals is arbitrary)
			std::vector<uint64_t> vals(c.size()), heap(GYS_TOPN);
			for (size_t k = 0; k < c.size(); ++k) vals[k] = c[k].m;
			const size_t nk = gyo_topn_u64(vals.data(), vals.size(), GYS_TOPN, heap.data());
			std::vector<uint64_t> mine;
			for (uint32_t r = 0; r < GYS_TOPN && r < c.size(); ++r) mine.push_back(out_metric[h * GYS_TOPN + r]);
			heap.resize(nk);
			std::sort(heap.begin(), heap.end(), std::greater<uint64_t>());
			CHECK(mine == heap, "kind %d host %u: metric values differ from the bounded heap's", kind, h);
			// the candidate filter of the single-host query path: same admitted set
			std::vector<uint32_t> fs(nsvc + 1), cnt(1, 0);
			std::vector<uint64_t> fm(nsvc + 1);
			kemu::launch((nsvc + 255u) / 256u, 256, 0, [&] { k_topn_filter(sb, nsvc, h, EPOCH, kind, fs.data(), fm.data(), cnt.data(), nsvc); });
			std::vector<Cand> f(cnt[0]);
			for (uint32_t k = 0; k < cnt[0]; ++k) f[k] = Cand{fm[k], fs[k]};
			std::sort(f.begin(), f.end(), [](const Cand &a, const Cand &b) { return a.m != b.m ? a.m > b.m : a.s < b.s; });
			// (the filter looks at every slot tagged with the host, the per-host kernel only at the host's member list: records tagged with
			// ANOTHER host are in neither, records of other hosts' members tagged with this host only in the filter's)
			std::vector<Cand> want;
			for (uint32_t s = 0; s < nsvc; ++s) {
				uint64_t m;
				if (admitted(sb + (size_t)s * 96, h, EPOCH, kind, &m)) want.push_back(Cand{m, s});
			}
			std::sort(want.begin(), want.end(), [](const Cand &a, const Cand &b) { return a.m != b.m ? a.m > b.m : a.s < b.s; });
			bool same = want.size() == f.size();
			for (size_t k = 0; same && k < f.size(); ++k) same = f[k].m == want[k].m && f[k].s == want[k].s;
			CHECK(same, "kind %d host %u: filter found %zu candidates, want %zu", kind, h, f.size(), want.size());
		}
	}
	if (fails) {
		printf("kemu topn: %d FAILURES\n", fails);
		return 1;
	}
	printf("kemu topn ok: %llu places filled over %u hosts x 4 kinds\n", (unsigned long long)npicked, NH);
	return 0;
}

// TEST INFRASTRUCTURE (CPU): the LOGIC of the TCP_CONN_NOTIFY roll-up -- k_conn_ingest (records read as 16-byte pieces through each
// record's own offset, staged in the wave's LDS region, service accumulators summed in the workgroup's LDS table, one set of device
// atomics per distinct service and ONE record-count add per workgroup) and k_conn_fold (window accumulators -> cumulative counters and
// Count-Min rows) -- under the CPU stand-in of the device model, on variable-stride batches of v4 / v6 / mixed flows of known and
// unknown services, at record counts around every boundary of the kernel's shape (wave 64, round 512, workgroup 1024), compared with
// the oracle's record-by-record walk (gyo_tcp_conn_sketch_batch, gyo_tcp_conn_pair_batch) and with sums taken here.
// Build + run: tests/test_kernel_logic_cpu.py.
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+r"(a[0]), "+r"(a[1]), "+r"(a[2]), "+r"(a[3]))
#define GYS_DYN_LDS(type, name) type *name = (type *)kemu::dyn_lds()
#include "../../../gyeeta_amd/csrc/gys_kernels.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <map>
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

void put32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
void put64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
void put16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }
uint64_t rd64(const uint8_t *p)
{
	uint64_t v;
	memcpy(&v, p, 8);
	return v;
}

// IP_PORT (32 bytes): ip128 @0, ip32 @16, aftype @20, port @24
void put_ip_port(uint8_t *p, std::mt19937 &rng, bool v6, uint32_t pool)
{
	if (v6) {
		put32(p, 0x20010DB8u);
		put32(p + 4, rng() % pool);
		put32(p + 8, 0);
		put32(p + 12, 1u + rng() % 7u);
		put32(p + 16, 0);
		put16(p + 20, 10);
	} else {
		put32(p + 16, 0x0A000001u + rng() % pool);
		put16(p + 20, 2);
	}
	put16(p + 24, (uint16_t)(1024u + rng() % 3000u));
}
} // namespace

int main(int argc, char **argv)
{
	if (!kemu::can_run(GYS_CONN_THREADS)) {
		printf("kemu: this process cannot have %u threads\n", GYS_CONN_THREADS);
		return 77;
	}
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 31u);
	// argv[2]: number of services (5000: the records of a workgroup name more services than its LDS table has entries -- some add to the
	// device accumulators directly); argv[3]: CUs conn_span plans for (1: a large call is few workgroups of many rounds, the last one ragged)
	const uint32_t NSVC = argc > 2 ? (uint32_t)atoi(argv[2]) : 37u, NUNKNOWN = 5;
	const uint32_t NCU = argc > 3 ? (uint32_t)atoi(argv[3]) : 1u;
	std::vector<uint64_t> gids(NSVC + NUNKNOWN);
	for (uint32_t s = 0; s < gids.size(); ++s) gids[s] = 0xABCD000000000000ull + 0x10001ull * (s + 1) + ((uint64_t)rng() << 20);
	gids[NSVC + 1] = ~0ull; // (an unregistered service whose glob_id IS the empty mark of the tables: no entry may match it)

	// the engine's glob-id table (open addressing on get_uint64_hash, 16-byte entries)
	uint32_t cap = 1;
	while (cap < 2 * NSVC) cap <<= 1;
	std::vector<TblEnt> ent(cap, TblEnt{GYS_EMPTY_KEY, 0, 0});
	for (uint32_t s = 0; s < NSVC; ++s) {
		uint32_t h = get_uint64_hash(gids[s]) & (cap - 1);
		while (ent[h].key != GYS_EMPTY_KEY) h = (h + 1) & (cap - 1);
		ent[h] = TblEnt{gids[s], s, 0};
	}

	const uint32_t NREG = 1u << GYS_HLL_P, NCMS = GYS_CMS_D * GYS_CMS_W;
	std::vector<uint32_t> hll32(NREG, 0), cms32(NCMS, 0), pair32(NCMS, 0), cpair32(NCMS, 0);
	std::vector<unsigned long long> cms64(NCMS, 0), pair64(NCMS, 0), cpair64(NCMS, 0), svc_win(NSVC * 3, 0), svc_ctr(NSVC * 4, 0);
	std::vector<uint64_t> counters(CTR_NUM, 0);
	// the oracle's side: its serial walk of the same bytes (flag bytes included: a connection counts once, on its listener side)
	std::vector<uint8_t> o_hll(NREG, 0);
	// a run on more than one CU's plan starts from registers of a window well under way (every register 1..3: the workgroups' HLL floor
	// is above zero and records of low rank skip their register), the default run from cleared ones
	if (NCU > 1)
		for (uint32_t k = 0; k < NREG; ++k) hll32[k] = o_hll[k] = (uint8_t)(1u + rng() % 3u);
	std::vector<uint32_t> o_cms32(NCMS, 0), o_pair32(NCMS, 0), o_cpair32(NCMS, 0);
	std::vector<uint64_t> o_cms64(NCMS, 0), o_pair64(NCMS, 0), o_cpair64(NCMS, 0);
	std::vector<uint64_t> want_ctr(NSVC * 4, 0);
	uint64_t want_events = 0, want_unknown = 0, want_tally[4] = {0, 0, 0, 0};

	const uint32_t sizes[] = {1, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 1535, 1536, 1537, 1600, 2500, 7000, 13000};
	uint32_t call = 0;
	for (uint32_t n : sizes) {
		const bool with_pair = (call++ & 1u) != 0; // every other call also feeds the (listener, client task group) pair
		// a few services per call, as a partha's message has; some calls all one service (every record on one LDS entry)
		const uint32_t nlocal = (call % 5u == 0) ? 1u : 1u + rng() % 12u;
		std::vector<uint32_t> local(nlocal);
		for (auto &s : local) s = rng() % (NSVC + NUNKNOWN);
		std::vector<uint8_t> raw((size_t)n * (280 + 48) + 16, 0);
		uint8_t *batch = (uint8_t *)(((uintptr_t)raw.data() + 7u) & ~(uintptr_t)7u);
		std::vector<uint32_t> offsets(n);
		uint32_t off = 0;
		for (uint32_t i = 0; i < n; ++i) {
			uint8_t *r = batch + off;
			offsets[i] = off;
			const uint32_t kind = rng() % 4u; // v4-v4, v6-v6, v4-v6, v6-v4
			put_ip_port(r + 0, rng, false, 50);
			put_ip_port(r + 32, rng, false, 50);
			put_ip_port(r + 64, rng, kind == 1 || kind == 3, 40);
			put_ip_port(r + 96, rng, kind == 1 || kind == 2, 8);
			put64(r + 128, 1700000000000000ull + rng());
			put64(r + 136, (rng() % 3u) ? 1700000001000000ull + rng() : 0ull); // tusec_close_: 0 = still open
			put64(r + 144, 0x7A5C000000000000ull + rng() % 9u);                   // cli_task_aggr_id_
			const uint32_t s = local[rng() % nlocal];
			put64(r + 192, gids[s]);
			const uint64_t sent = (rng() % 4u) ? (uint64_t)rng() * (1u + rng() % 5000u) : 0ull, rcvd = (rng() % 5u) ? (uint64_t)rng() : 0ull;
			put64(r + 208, sent);
			put64(r + 216, rcvd);
			const uint16_t cmdlen = (uint16_t)(rng() % 41u);
			const uint8_t pad = (uint8_t)((8u - cmdlen % 8u) % 8u);
			put16(r + 272, cmdlen);
			// the five bool bytes: connect / accept / loopback / pre-existing / notified-before (any non-zero byte is `true`)
			const uint32_t side = rng() % 8u; // 0-3 accepting half, 4-5 connecting half, 6 loopback (both), 7 neither
			r[274] = (side >= 4u && side <= 6u) ? (uint8_t)(1u + rng() % 3u) : 0;
			r[275] = (side <= 3u || side == 6u) ? 1 : 0;
			r[276] = side == 6u;
			r[277] = (rng() % 9u) == 0;
			r[278] = (rng() % 3u) == 0 ? (uint8_t)(1u + rng() % 200u) : 0;
			r[279] = pad;
			for (uint32_t k = 0; k < (uint32_t)cmdlen + pad; ++k) r[280 + k] = (uint8_t)rng(); // (bytes the roll-up never looks at)
			off += 280u + cmdlen + pad;
			++want_events;
		}
		CHECK(gyo_tcp_conn_sketch_batch(batch, (int)n, batch + off, o_hll.data(), o_cms32.data(), o_cms64.data()) == (int)n, "oracle walk of %u records", n);
		CHECK(gyo_tcp_conn_walk_tallies(batch, (int)n, batch + off, want_tally) == (int)n, "oracle tallies");
		CHECK(gyo_tcp_conn_svc_counters(batch, (int)n, batch + off, gids.data(), NSVC, want_ctr.data(), &want_unknown) == (int)n, "oracle service counters");
		if (with_pair)
			CHECK(gyo_tcp_conn_pair_batch(batch, (int)n, batch + off, o_pair32.data(), o_pair64.data(), o_cpair32.data(), o_cpair64.data()) == (int)n, "oracle pair walk");

		ConnP p{};
		p.batch = batch;
		p.offsets = offsets.data();
		p.n = n;
		p.gid = DevTable{ent.data(), cap - 1};
		p.hll32 = hll32.data();
		p.cms32 = cms32.data();
		p.cms64 = cms64.data();
		p.svc_win = svc_win.data();
		p.counters = counters.data();
		p.pair32 = with_pair ? pair32.data() : nullptr;
		p.pair64 = with_pair ? pair64.data() : nullptr;
		p.cpair32 = with_pair ? cpair32.data() : nullptr;
		p.cpair64 = with_pair ? cpair64.data() : nullptr;
		p.span = conn_span(n, NCU);
		CHECK(p.span % 64u == 0 && p.span >= GYS_CONN_RECS, "conn_span(%u, %u) = %u", n, NCU, p.span);
		kemu::launch((n + p.span - 1u) / p.span, GYS_CONN_THREADS, 0, [&] { k_conn_ingest(p); });
		CHECK(counters[CTR_CONN_EVENTS] == want_events, "n %u: events %llu, want %llu", n, (unsigned long long)counters[CTR_CONN_EVENTS], (unsigned long long)want_events);
		CHECK(counters[CTR_CONN_UNKNOWN] == want_unknown, "n %u: unknown %llu, want %llu", n, (unsigned long long)counters[CTR_CONN_UNKNOWN], (unsigned long long)want_unknown);
		CHECK(counters[CTR_CONN_NEW] == want_tally[0] && counters[CTR_CONN_CLOSED] == want_tally[1] && counters[CTR_CONN_CLOSED_NO_NOTIFY] == want_tally[2] &&
			      counters[CTR_CONN_CLI_SIDE] == want_tally[3],
		      "n %u: walk tallies new %llu closed %llu closed-without-notify %llu client-side %llu, want %llu %llu %llu %llu", n,
		      (unsigned long long)counters[CTR_CONN_NEW], (unsigned long long)counters[CTR_CONN_CLOSED], (unsigned long long)counters[CTR_CONN_CLOSED_NO_NOTIFY],
		      (unsigned long long)counters[CTR_CONN_CLI_SIDE], (unsigned long long)want_tally[0], (unsigned long long)want_tally[1],
		      (unsigned long long)want_tally[2], (unsigned long long)want_tally[3]);
		for (uint32_t k = 0; k < NREG; ++k) CHECK(hll32[k] == o_hll[k], "n %u: HLL register %u is %u, oracle %u", n, k, hll32[k], o_hll[k]);
		if (call % 3u == 0) { // a window boundary every third call: the accumulators of several calls fold at once
			kemu::launch((NSVC + 255u) / 256u, 256, 0, [&] { k_conn_fold(svc_win.data(), svc_ctr.data(), gids.data(), NSVC, cms32.data(), cms64.data()); });
			for (uint32_t k = 0; k < NSVC * 3; ++k) CHECK(svc_win[k] == 0, "window accumulator %u not cleared", k);
			for (uint32_t k = 0; k < NSVC * 4; ++k)
				CHECK(svc_ctr[k] == want_ctr[k], "n %u: service %u counter %u is %llu, want %llu", n, k / 4, k % 4, svc_ctr[k], (unsigned long long)want_ctr[k]);
			for (uint32_t k = 0; k < NCMS; ++k) {
				CHECK(cms32[k] == o_cms32[k], "n %u: cms32[%u] %u, oracle %u", n, k, cms32[k], o_cms32[k]);
				CHECK(cms64[k] == o_cms64[k], "n %u: cms64[%u] %llu, oracle %llu", n, k, cms64[k], (unsigned long long)o_cms64[k]);
			}
		}
		for (uint32_t k = 0; k < NCMS; ++k) {
			CHECK(pair32[k] == o_pair32[k], "n %u: pair32[%u] %u, oracle %u", n, k, pair32[k], o_pair32[k]);
			CHECK(pair64[k] == o_pair64[k], "n %u: pair64[%u] %llu, oracle %llu", n, k, pair64[k], (unsigned long long)o_pair64[k]);
			CHECK(cpair32[k] == o_cpair32[k], "n %u: client-side pair32[%u] %u, oracle %u", n, k, cpair32[k], o_cpair32[k]);
			CHECK(cpair64[k] == o_cpair64[k], "n %u: client-side pair64[%u] %llu, oracle %llu", n, k, cpair64[k], (unsigned long long)o_cpair64[k]);
		}
	}
	kemu::launch((NSVC + 255u) / 256u, 256, 0, [&] { k_conn_fold(svc_win.data(), svc_ctr.data(), gids.data(), NSVC, cms32.data(), cms64.data()); });
	for (uint32_t k = 0; k < NSVC * 4; ++k) CHECK(svc_ctr[k] == want_ctr[k], "final: service %u counter %u is %llu, want %llu", k / 4, k % 4, svc_ctr[k], (unsigned long long)want_ctr[k]);
	for (uint32_t k = 0; k < NCMS; ++k) {
		CHECK(cms32[k] == o_cms32[k], "final: cms32[%u] %u, oracle %u", k, cms32[k], o_cms32[k]);
		CHECK(cms64[k] == o_cms64[k], "final: cms64[%u] %llu, oracle %llu", k, cms64[k], (unsigned long long)o_cms64[k]);
	}
	uint64_t nz = 0;
	for (uint32_t k = 0; k < NREG; ++k) nz += hll32[k] != 0;
	if (fails) {
		printf("kemu conn: %d FAILURES\n", fails);
		return 1;
	}
	printf("kemu conn ok: %llu records in %u calls, %llu of unknown services, %llu HLL registers set\n", (unsigned long long)want_events, call,
	       (unsigned long long)want_unknown, (unsigned long long)nz);
	return 0;
}

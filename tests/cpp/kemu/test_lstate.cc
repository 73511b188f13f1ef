// TEST INFRASTRUCTURE (CPU): the LOGIC of the LISTENER_STATE_NOTIFY roll-up k_lstate_ingest under the CPU stand-in of the device model,
// on records whose every byte except the listener id is RANDOM (any curr_state_, any query_flags_, counters beyond 2^31 whose int sums
// wrap as the reference's do): per-host LISTEN_SUMM_STATS accumulators, the four record counters, the kept 88-byte state of every
// listener and the per-listener QPS / active-connection histograms equal the oracle's walk (gyo_listener_state_rollup,
// gyo_hist_add).  Second part: k_actconn_ingest on random 104-byte ACTIVE_CONN_STATS rows (local and remote listeners, known and unknown
// ones): the pair Count-Min tables, the per-listener sums and the three row counters equal gyo_active_conn_sketch_batch and sums taken here.
// Under AddressSanitizer (tools/kemu_tsan.sh address): no field of a record indexes outside an array.
// Build + run: tests/test_kernel_logic_cpu.py.
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+r"(a[0]), "+r"(a[1]), "+r"(a[2]), "+r"(a[3]))
#define GYS_DYN_LDS(type, name) type *name = (type *)kemu::dyn_lds()
#include "../../../gyeeta_amd/csrc/gys_kernels.hpp"

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
} // namespace

int main(int argc, char **argv)
{
	if (!kemu::can_run(256u)) {
		printf("kemu: this process cannot have 256 threads\n");
		return 77;
	}
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 3u);
	const uint32_t NSVC = 50, NUNKNOWN = 6, NHOSTS = 600;
	std::vector<uint64_t> gids(NSVC + NUNKNOWN);
	for (auto &g : gids) g = ((uint64_t)rng() << 32) | rng();
	uint32_t cap = 128;
	std::vector<TblEnt> ent(cap, TblEnt{GYS_EMPTY_KEY, 0, 0});
	for (uint32_t s = 0; s < NSVC; ++s) {
		uint32_t h = get_uint64_hash(gids[s]) & (cap - 1);
		while (ent[h].key != GYS_EMPTY_KEY) h = (h + 1) & (cap - 1);
		ent[h] = TblEnt{gids[s], s, 0};
	}
	std::vector<uint8_t> svc_state(NSVC * 96, 0), want_state(NSVC * 96, 0);
	std::vector<int32_t> host_summ(NHOSTS * 16, 0);
	std::vector<gyo_listen_summ_stats> want_summ(NHOSTS);
	memset(want_summ.data(), 0, sizeof(gyo_listen_summ_stats) * NHOSTS);
	std::vector<uint64_t> counters(CTR_NUM, 0);
	std::vector<unsigned long long> claim(NSVC, 0);
	std::vector<gys_hist_rec> qps_hist(NSVC), act_hist(NSVC);
	std::vector<gyo_hist> o_qps(NSVC), o_act(NSVC);
	memset(qps_hist.data(), 0, sizeof(gys_hist_rec) * NSVC);
	memset(act_hist.data(), 0, sizeof(gys_hist_rec) * NSVC);
	for (uint32_t s = 0; s < NSVC; ++s) {
		qps_hist[s].max_val_seen = INT32_MIN; // (gys_create: k_hist_init with the minimum of the histogram's value type, GY_HISTOGRAM<int, ...>)
		act_hist[s].max_val_seen = INT32_MIN;
		gyo_hist_init(&o_qps[s], GYO_SEMI_LOG_HASH_LO);
		gyo_hist_init(&o_act[s], GYO_HASH_1_3000);
	}
	CHECK(gyo_hist_nbuckets(GYO_SEMI_LOG_HASH_LO) <= 15 && gyo_hist_nbuckets(GYO_HASH_1_3000) <= 15, "a 15-bucket record cannot hold these histograms");
	uint64_t want_rec = 0, want_missed = 0, want_deleted = 0, want_errors = 0;

	for (uint32_t epoch = 1; epoch <= 6; ++epoch) {
		const uint32_t n = epoch == 1 ? 1u : 200u + rng() % 400u;
		std::vector<uint64_t> raw;
		std::vector<uint32_t> offsets(n), host_slot(n);
		for (uint32_t i = 0; i < n; ++i) {
			const uint32_t var = rng() % 30u, pad = (8u - var % 8u) % 8u, sz = 88u + var + pad;
			offsets[i] = (uint32_t)(raw.size() * 8u);
			raw.resize(raw.size() + sz / 8u);
			uint8_t *r = (uint8_t *)raw.data() + offsets[i];
			for (uint32_t k = 0; k < sz; ++k) r[k] = (uint8_t)rng(); // everything random ...
			const uint32_t s = rng() % (NSVC + NUNKNOWN);
			memcpy(r, &gids[s], 8); // ... except the listener id
			if (rng() % 4u) r[79] = (uint8_t)(rng() % 6u);           // most records carry a state in range
			if (rng() % 8u == 0) r[84] = 0xC0;                        // LISTEN_FLAG_DELETE
			if (rng() % 3u) { // plausible counters in most records (the rest stay random, beyond 2^31)
				const uint32_t small[5] = {rng() % 100000u, rng() % 5000u, rng() % 100000u, rng() % 100000u, rng() % 50u};
				memcpy(r + 8, &small[0], 4);
				memcpy(r + 20, &small[1], 4);
				memcpy(r + 36, &small[2], 4);
				memcpy(r + 40, &small[3], 4);
				memcpy(r + 44, &small[4], 4);
			}
			r[85] = (uint8_t)var;
			r[86] = (uint8_t)pad;
			// (call 2 goes through the single-host form; calls 5 and 6 name more hosts per workgroup than the LDS roll-up has rows: some records add to the device rows directly)
			host_slot[i] = epoch == 2 ? 1u : epoch >= 5 ? rng() % NHOSTS : rng() % 3u;
			// ---- expectation, record by record (server/gy_mconnhdlr.cc:11175-11256)
			++want_rec;
			if (s >= NSVC) {
				++want_missed;
				continue;
			}
			if (r[84] == 0xC0) {
				++want_deleted;
				memset(&want_state[s * 96 + 88], 0, 4);
				continue;
			}
			if (r[79] > 5) {
				++want_errors;
				continue;
			}
			int nerr = 0;
			gyo_listener_state_rollup(r, 1, r + sz, &want_summ[host_slot[i]], &nerr);
			memcpy(&want_state[s * 96], r, 88);
			const uint64_t tail = (uint64_t)epoch | ((uint64_t)host_slot[i] << 32);
			memcpy(&want_state[s * 96 + 88], &tail, 8);
			uint32_t nq, na;
			memcpy(&nq, r + 8, 4);
			memcpy(&na, r + 20, 4);
			gyo_hist_add(&o_qps[s], (int64_t)(int32_t)(nq / 5u));
			gyo_hist_add(&o_act[s], (int64_t)(int32_t)na);
		}
		LStateP p{};
		p.batch = (const uint8_t *)raw.data();
		p.offsets = offsets.data();
		p.host_slot = epoch == 2 ? nullptr : host_slot.data();
		p.single_host = 1;
		p.n = n;
		p.gid = DevTable{ent.data(), cap - 1};
		p.svc_state = svc_state.data();
		p.host_summ = host_summ.data();
		p.epoch = epoch;
		p.counters = counters.data();
		p.qps_hist = qps_hist.data();
		p.act_hist = act_hist.data();
		p.claim = claim.data();
		p.launch = epoch;
		if (n <= GYS_LSTATE_FUSED_MAX && (epoch & 1u)) { // (as run_lstate launches a message: both passes in one workgroup -- every other call here)
			kemu::launch(1, GYS_LSTATE_FUSED_MAX, 0, [&] { k_lstate_both(p); });
		} else {
			kemu::launch((n + 255u) / 256u, 256, 0, [&] { k_lstate_ingest(p); });
			kemu::launch((n + 255u) / 256u, 256, 0, [&] { k_lstate_keep(p); });
		}
		// several records of one listener in one call: the LAST one in stream order stays, whole (the reference's serial walk) -- want_state
		// was built in that order above
		for (uint32_t s = 0; s < NSVC; ++s) {
			uint32_t got_ep, want_ep;
			memcpy(&got_ep, &svc_state[s * 96 + 88], 4);
			memcpy(&want_ep, &want_state[s * 96 + 88], 4);
			// (a listener whose last record was a delete has no current state: only the cleared window number means anything)
			if (want_ep == 0) CHECK(got_ep == 0, "epoch %u: listener %u was deleted last, its state is still marked current", epoch, s);
			else CHECK(!memcmp(&svc_state[s * 96], &want_state[s * 96], 96), "epoch %u: kept state of listener %u differs", epoch, s);
		}
	}
	CHECK(counters[CTR_LSTATE_RECORDS] == want_rec && counters[CTR_LSTATE_MISSED] == want_missed && counters[CTR_LSTATE_DELETED] == want_deleted &&
	      counters[CTR_LSTATE_ERRORS] == want_errors, "counters %llu %llu %llu %llu, want %llu %llu %llu %llu", (unsigned long long)counters[CTR_LSTATE_RECORDS],
	      (unsigned long long)counters[CTR_LSTATE_MISSED], (unsigned long long)counters[CTR_LSTATE_DELETED], (unsigned long long)counters[CTR_LSTATE_ERRORS],
	      (unsigned long long)want_rec, (unsigned long long)want_missed, (unsigned long long)want_deleted, (unsigned long long)want_errors);
	for (uint32_t h = 0; h < NHOSTS; ++h) {
		const int32_t *s = &host_summ[h * 16];
		const gyo_listen_summ_stats &w = want_summ[h];
		for (int k = 0; k < 6; ++k) CHECK(s[k] == w.nstates[k], "host %u nstates[%d] %d, oracle %d", h, k, s[k], w.nstates[k]);
		CHECK(s[6] == w.tot_qps && s[7] == w.tot_act_conn && s[8] == w.tot_kb_inbound && s[9] == w.tot_kb_outbound && s[10] == w.tot_ser_errors &&
		      s[11] == w.nlisteners && s[12] == w.nactive, "host %u sums %d %d %d %d %d %d %d, oracle %d %d %d %d %d %d %d", h, s[6], s[7], s[8], s[9], s[10], s[11], s[12],
		      w.tot_qps, w.tot_act_conn, w.tot_kb_inbound, w.tot_kb_outbound, w.tot_ser_errors, w.nlisteners, w.nactive);
	}
	uint64_t nsamples = 0;
	for (uint32_t s = 0; s < NSVC; ++s) {
		const gys_hist_rec *g[2] = {&qps_hist[s], &act_hist[s]};
		const gyo_hist *o[2] = {&o_qps[s], &o_act[s]};
		for (int k = 0; k < 2; ++k) {
			CHECK(g[k]->total_count == o[k]->total_count && (g[k]->max_val_seen == o[k]->max_val_seen), "listener %u histogram %d: total %llu max %lld, oracle %llu %lld", s, k,
			      (unsigned long long)g[k]->total_count, (long long)g[k]->max_val_seen, (unsigned long long)o[k]->total_count, (long long)o[k]->max_val_seen);
			for (int b = 0; b < o[k]->nbuckets; ++b)
				CHECK(g[k]->stats[b].count == o[k]->stats[b].count && g[k]->stats[b].sum == o[k]->stats[b].sum, "listener %u histogram %d bucket %d", s, k, b);
			nsamples += o[k]->total_count;
		}
	}
	// ------------------------------------------------------------------------------------------------ ACTIVE_CONN_STATS rows
	{
		const uint32_t NCMS = GYS_CMS_D * GYS_CMS_W;
		std::vector<uint32_t> pair32(2 * NCMS, 0), o32(2 * NCMS, 0), win_rows(4, 0); // (the local-listener rows' table, then the remote-listener rows')
		std::vector<unsigned long long> pair64(2 * NCMS, 0), svc_act(NSVC * 4, 0), want_act(NSVC * 4, 0);
		std::vector<uint64_t> o64(2 * NCMS, 0), ctr(CTR_NUM, 0);
		uint64_t want_local = 0, want_remote = 0, want_unknown = 0;
		for (uint32_t call = 0; call < 3; ++call) {
			const uint32_t n = call == 0 ? 1u : 100u + rng() % 500u;
			std::vector<uint64_t> raw((size_t)n * 13u);
			uint8_t *b = (uint8_t *)raw.data();
			for (size_t k = 0; k < (size_t)n * 104u; ++k) b[k] = (uint8_t)rng();
			for (uint32_t i = 0; i < n; ++i) {
				uint8_t *r = b + (size_t)i * 104u;
				const uint32_t s = rng() % (NSVC + NUNKNOWN);
				memcpy(r, &gids[s], 8);
				const uint64_t task = 0x7A5C000000000000ull + rng() % 7u; // a few client task groups per listener
				memcpy(r + 8, &task, 8);
				if (r[102] & 2) {
					++want_remote;
					continue;
				}
				++want_local;
				if (s >= NSVC) {
					++want_unknown;
					continue;
				}
				uint64_t sent, rcvd;
				uint16_t act;
				memcpy(&sent, r + 72, 8);
				memcpy(&rcvd, r + 80, 8);
				memcpy(&act, r + 100, 2);
				want_act[s * 4 + 0] += 1;
				want_act[s * 4 + 1] += sent;
				want_act[s * 4 + 2] += rcvd;
				want_act[s * 4 + 3] += act;
			}
			uint64_t out[2];
			gyo_active_conn_sketch_batch2(b, (int)n, o32.data(), o64.data(), o32.data() + NCMS, o64.data() + NCMS, out);
			ActConnP p{};
			p.batch = b;
			p.n = n;
			p.gid = DevTable{ent.data(), cap - 1};
			p.pair32 = pair32.data();
			p.pair64 = pair64.data();
			p.svc_act = svc_act.data();
			p.counters = ctr.data();
			p.win_rows = win_rows.data();
			kemu::launch((n + 255u) / 256u, 256, 0, [&] { k_actconn_ingest(p); });
		}
		CHECK(ctr[CTR_ACTCONN_RECORDS] == want_local && ctr[CTR_ACTCONN_REMOTE_LISTEN] == want_remote && ctr[CTR_ACTCONN_UNKNOWN] == want_unknown && win_rows[0] == want_local + want_remote,
		      "active-conn counters %llu %llu %llu rows %u, want %llu %llu %llu", (unsigned long long)ctr[CTR_ACTCONN_RECORDS], (unsigned long long)ctr[CTR_ACTCONN_REMOTE_LISTEN],
		      (unsigned long long)ctr[CTR_ACTCONN_UNKNOWN], win_rows[0], (unsigned long long)want_local, (unsigned long long)want_remote, (unsigned long long)want_unknown);
		for (uint32_t k = 0; k < 2 * NCMS; ++k) CHECK(pair32[k] == o32[k] && pair64[k] == o64[k], "active-conn pair cell %u: %u / %llu, oracle %u / %llu", k, pair32[k], pair64[k], o32[k], (unsigned long long)o64[k]);
		for (uint32_t k = 0; k < NSVC * 4; ++k) CHECK(svc_act[k] == want_act[k], "listener %u active-conn sum %u: %llu, want %llu", k / 4, k % 4, svc_act[k], want_act[k]);
		nsamples += want_local;
	}
	if (fails) {
		printf("kemu lstate: %d FAILURES\n", fails);
		return 1;
	}
	printf("kemu lstate ok: %llu records (%llu unknown, %llu deleted, %llu with a state out of range), %llu histogram samples + active-conn rows\n", (unsigned long long)want_rec,
	       (unsigned long long)want_missed, (unsigned long long)want_deleted, (unsigned long long)want_errors, (unsigned long long)nsamples);
	return 0;
}

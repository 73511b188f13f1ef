// Host-fed boundary benchmark: what MCONN_HANDLER's L2 threads would call.  T threads (default 16 =
// MAX_L2_MISC_THREADS, server/gy_mconnhdlr.h:60) drive GYS_MCONN_HANDLER with pageable host buffers:
//   leg 1  partha_tcp_conn_info  : 2048-record TCP_CONN_NOTIFY messages (MAX_NUM_CONNS, 280 B fixed stride)
//   leg 2  partha_listener_state : 512-record LISTENER_STATE_NOTIFY messages (MAX_NUM_LISTENERS, 88 B)
//   leg 3  handle_ipv4_resp_events: 65536 raw 24-byte response events of one host per call
// on C2's registry (1000 hosts x 100 services).  Prints ONE JSON line.  Plain g++ + the C ABI (no HIP headers).
//   g++ -std=c++17 -O2 tools/cpp/bench_hostfed.cc -o bench_hostfed -Lgyeeta_amd/lib -lgysketch -Wl,-rpath,... -pthread
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../gyeeta_amd/csrc/gys_mconn_shim.hpp"

static uint64_t splitmix(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

static void machine_id(uint32_t h, uint8_t out[16])
{
	const uint64_t a = splitmix(h), b = splitmix(h + (1ull << 32));
	memcpy(out, &a, 8);
	memcpy(out + 8, &b, 8);
}
static uint64_t glob_id(uint32_t h, uint32_t s) { return splitmix(((uint64_t)h << 20) + s); }

int main(int argc, char **argv)
{
	const int nthreads = argc > 1 ? atoi(argv[1]) : 16;
	const double secs = argc > 2 ? atof(argv[2]) : 2.0;
	const uint32_t NH = 1000, SP = 100;
	gys_config cfg{};
	cfg.struct_size = sizeof(cfg);
	cfg.device = 0;
	cfg.nranks = 1;
	cfg.max_hosts = NH;
	cfg.max_services = NH * SP;
	cfg.max_clusters = 8;
	cfg.enable_tdigest = 1;
	cfg.max_batch_events = 1u << 20;
	gyeeta_amd::GYS_MCONN_HANDLER h(cfg);
	std::vector<gys_listener_info> li(SP);
	for (uint32_t host = 0; host < NH; ++host) {
		uint8_t mid[16];
		machine_id(host, mid);
		char cl[16];
		snprintf(cl, sizeof(cl), "cluster%u", host % 8);
		if (!h.partha_register(mid, cl)) return 2;
		for (uint32_t s = 0; s < SP; ++s) {
			li[s] = gys_listener_info{};
			li[s].glob_id = glob_id(host, s);
			li[s].netns = 0xF0000000u + 4u * host;
			li[s].port = (uint16_t)(1024 + s);
			li[s].is_any_ip = 1;
			snprintf(li[s].comm, sizeof(li[s].comm), "svc%u", s);
		}
		if (!h.partha_new_listeners(mid, li.data(), SP)) return 3;
	}
	gys_sync(h.ctx());

	struct Leg {
		const char *name;
		uint32_t nrec, recsz;
	};
	const Leg legs[3] = {{"tcp_conn", 2048, 280}, {"listener_state", 512, 88}, {"resp_events", 65536, 24}};
	printf("{\"threads\": %d, \"registry\": \"%u hosts x %u services\", \"buffers\": \"pageable host memory\"", nthreads, NH, SP);
	for (int L = 0; L < 3; ++L) {
		const Leg &lg = legs[L];
		std::atomic<uint64_t> msgs{0};
		std::atomic<bool> stop{false}, fail{false};
		std::vector<std::thread> th;
		for (int t = 0; t < nthreads; ++t) {
			th.emplace_back([&, t] {
				// one message per thread, rebuilt per host only in the fields the kernels read (ids / addresses / ports)
				std::vector<uint64_t> buf((size_t)lg.nrec * lg.recsz / 8 + 8);
				uint8_t *p = (uint8_t *)buf.data();
				uint64_t r = splitmix(1000 + t);
				uint32_t host = (uint32_t)t % NH;
				while (!stop.load(std::memory_order_relaxed)) {
					uint8_t mid[16];
					machine_id(host, mid);
					bool ok = true;
					if (L == 0) {
						for (uint32_t i = 0; i < lg.nrec; ++i) { // comm::TCP_CONN_NOTIFY (common/gy_comm_proto.h:1665-1742)
							uint8_t *q = p + (size_t)i * 280;
							r = splitmix(r);
							memset(q, 0, 280);
							const uint32_t cli = 0x0A | (uint32_t)(r & 0xFFFFFF00u), ser = 0x0A | (host << 8);
							const uint16_t cport = (uint16_t)(16000 + (r >> 32) % 49536), sport = (uint16_t)(1024 + (r >> 48) % SP);
							for (int k = 0; k < 4; ++k) { // cli_, ser_, nat_cli_, nat_ser_: ip32 @16, aftype @20, port @24 of each 32-byte IP_PORT
								const uint32_t ip = (k & 1) ? ser : cli;
								const uint16_t pt = (k & 1) ? sport : cport;
								const uint16_t af = 2;
								memcpy(q + 32 * k + 16, &ip, 4);
								memcpy(q + 32 * k + 20, &af, 2);
								memcpy(q + 32 * k + 24, &pt, 2);
							}
							const uint64_t gid = glob_id(host, (uint32_t)((r >> 48) % SP)), sent = 200 + (r & 0xFFF), rcvd = 300 + ((r >> 12) & 0xFFF);
							memcpy(q + 192, &gid, 8);
							memcpy(q + 208, &sent, 8);
							memcpy(q + 216, &rcvd, 8);
						}
						ok = h.partha_tcp_conn_info(mid, p, (int)lg.nrec, p + (size_t)lg.nrec * 280);
					} else if (L == 1) {
						for (uint32_t i = 0; i < lg.nrec; ++i) { // comm::LISTENER_STATE_NOTIFY (common/gy_comm_proto.h:2183-2254)
							uint8_t *q = p + (size_t)i * 88;
							r = splitmix(r);
							memset(q, 0, 88);
							const uint64_t gid = glob_id(host, i % SP);
							const uint32_t nq = (uint32_t)(r & 0x3FF), tot = nq * 20, nc = (uint32_t)((r >> 10) & 0xFF);
							memcpy(q, &gid, 8);
							memcpy(q + 8, &nq, 4);
							memcpy(q + 12, &tot, 4);
							memcpy(q + 16, &nc, 4);
							q[79] = (uint8_t)((r >> 40) % 6); // curr_state_
						}
						ok = h.partha_listener_state(mid, p, (int)lg.nrec, p + (size_t)lg.nrec * 88);
					} else {
						for (uint32_t i = 0; i < lg.nrec; ++i) { // tcp_ipv4_resp_event_t (common/gy_ebpf_kernel.h:106-111)
							uint32_t *q = (uint32_t *)(p + (size_t)i * 24);
							r = splitmix(r);
							const uint32_t s = (uint32_t)((r >> 8) % SP), ms = (uint32_t)((r >> 20) & 0x3FF);
							const uint16_t sp_be = (uint16_t)(((1024 + s) >> 8) | ((1024 + s) << 8)), dp = (uint16_t)(r >> 44);
							q[0] = 0x0A | (host << 8);
							q[1] = 0x0A | (uint32_t)(r & 0xFFFFFF00u);
							q[2] = 0xF0000000u + 4u * host;
							q[3] = (uint32_t)sp_be | ((uint32_t)dp << 16);
							q[5] = (uint32_t)(r >> 33);
							q[4] = q[5] + ms;
						}
						ok = h.handle_ipv4_resp_events(mid, p, lg.nrec);
					}
					if (!ok) {
						fail = true;
						break;
					}
					msgs.fetch_add(1, std::memory_order_relaxed);
					host = (host + (uint32_t)nthreads) % NH;
				}
			});
		}
		const auto t0 = std::chrono::steady_clock::now();
		std::this_thread::sleep_for(std::chrono::duration<double>(secs));
		stop = true;
		for (auto &x : th) x.join();
		gys_sync(h.ctx());
		const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if (fail) {
			fprintf(stderr, "leg %s failed: %s\n", lg.name, gys_last_error());
			return 4;
		}
		h.send_cluster_state((uint64_t)(L + 1) * 5000000ull);
		const double m = (double)msgs.load();
		printf(", \"%s\": {\"records_per_msg\": %u, \"msgs_per_s\": %.1f, \"records_per_s\": %.1f, \"GBps\": %.3f}", lg.name, lg.nrec, m / dt,
		       m * lg.nrec / dt, m * lg.nrec * lg.recsz / dt / 1e9);
	}
	gys_counters ctr{};
	gys_get_counters(h.ctx(), &ctr);
	printf(", \"counters\": {\"conn_events\": %llu, \"lstate_records\": %llu, \"resp_events\": %llu, \"conn_unknown_service\": %llu, \"lstate_missed\": %llu, \"resp_dropped_nolistener\": %llu, "
	       "\"resp_calls_queued\": %llu, \"resp_submissions\": %llu, \"conn_calls_queued\": %llu, \"conn_submissions\": %llu, "
	       "\"lstate_calls_queued\": %llu, \"lstate_submissions\": %llu, \"stage_waits\": %llu}}\n",
	       (unsigned long long)ctr.conn_events, (unsigned long long)ctr.lstate_records, (unsigned long long)ctr.resp_events,
	       (unsigned long long)ctr.conn_unknown_service, (unsigned long long)ctr.lstate_missed, (unsigned long long)ctr.resp_dropped_nolistener,
	       (unsigned long long)ctr.resp_calls_queued, (unsigned long long)ctr.resp_submissions, (unsigned long long)ctr.conn_calls_queued,
	       (unsigned long long)ctr.conn_submissions, (unsigned long long)ctr.lstate_calls_queued, (unsigned long long)ctr.lstate_submissions,
	       (unsigned long long)ctr.stage_waits);
	return 0;
}

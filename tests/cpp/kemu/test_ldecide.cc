// TEST INFRASTRUCTURE (CPU): the LOGIC of k_listener_decide (TCP_LISTENER::get_curr_state, common/gy_socket_stat.cc:2020-2870, + its caller's
// part :4241-4266) under the CPU stand-in of the device model, against the oracle's restatement oracle/gy_oracle_lstate.c on random scan records
// and inputs drawn so that every region of the decision tree is populated, over several rounds (the history bytes carry over).
// Build + run: tests/test_kernel_logic_cpu.py.
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+r"(a[0]), "+r"(a[1]), "+r"(a[2]), "+r"(a[3]))
#define GYS_DYN_LDS(type, name) type *name = (type *)kemu::dyn_lds()
#include "../../../gyeeta_amd/csrc/gys_kernels.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <set>
#include <vector>

#include "../../../oracle/gy_oracle.h"

using namespace gys;

static_assert(sizeof(gys_listener_scan) == sizeof(gyo_listener_scan) && sizeof(gys_listener_issue_in) == sizeof(gyo_listener_issue_in) &&
		      sizeof(gys_listener_decision) == sizeof(gyo_listener_decision),
	      "product and oracle records have the same layout");

int main(int argc, char **argv)
{
	if (!kemu::can_run(256u)) {
		printf("kemu: this process cannot have 256 threads\n");
		return 77;
	}
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 5u);
	auto pick = [&](std::initializer_list<double> v) { return *(v.begin() + rng() % v.size()); };
	static const int THR[13] = {1, 10, 30, 60, 100, 150, 200, 300, 450, 700, 1000, 3000, 15000};
	const uint32_t N = 60000;
	std::vector<gys_listener_scan> sc(N);
	std::vector<gys_listener_issue_in> in(N);
	std::vector<uint8_t> hist(2 * N, 0), ohist(2 * N, 0), notify((size_t)N * 88, 0);
	std::vector<gys_listener_decision> out(N);
	std::set<int> lines;
	int fails = 0;
	for (int rnd = 0; rnd < 6; ++rnd) {
		for (uint32_t i = 0; i < N; ++i) {
			gys_listener_scan &s = sc[i];
			memset(&s, 0, sizeof(s));
			const int base = (int)(rng() % 6u);
			auto clampi = [](int v) { return v < 0 ? 0 : v > 12 ? 12 : v; };
			int up = (int)pick({-1, 0, 0, 0, 1, 1, 2, 4});
			if (rnd >= 2 && i % 3u == 0) up = 2 + (int)(rng() % 3u); // the same listeners stay high: the history byte fills up
			const int i5 = clampi(base + up), i300 = clampi(base + (int)pick({0, 0, 1, 2})), iall = clampi(base + (int)pick({0, 0, 1}));
			const int idx[4] = {i5, i300, base, iall};
			const int64_t cnt5 = rng() % 10u == 0 ? 0 : 1 + rng() % 3000u;
			s.tcount[0] = cnt5;
			s.tcount[1] = cnt5 * (20 + rng() % 50u);
			s.tcount[2] = rng() % 5000000u;
			s.tcount[3] = s.tcount[2] + rng() % 5000000u;
			const double m5d = pick({2.0, 5.0, 20.0, 80.0});
			const double mean[4] = {m5d * pick({0.5, 0.79, 0.8, 1.0, 1.19, 1.2, 1.21, 1.5, 3.0}), m5d * pick({0.9, 1.0, 1.05, 1.3}), m5d, m5d * pick({0.8, 1.0, 1.5})};
			for (int lv = 0; lv < 4; ++lv) {
				s.p95_ms[lv] = THR[idx[lv]];
				s.p99_ms[lv] = THR[clampi(idx[lv] + (int)(rng() % 3u))];
				s.p25_ms[lv] = THR[clampi(idx[lv] - 1)];
				s.tsum[lv] = (int64_t)((double)s.tcount[lv] * mean[lv]);
			}
			s.last_qps = (int32_t)((double)cnt5 * pick({0.1, 0.2, 0.3}));
			s.curr_qps = s.last_qps > (int32_t)(cnt5 / 5) ? s.last_qps : (int32_t)(cnt5 / 5);
			s.qps_p25 = (int32_t)pick({0, 2, 10, 50, 200});
			s.qps_p95 = s.qps_p25 + (int32_t)pick({0, 1, 5, 50, 400, 1000});
			s.act_p25 = (int32_t)pick({0, 1, 3, 10});
			s.act_p95 = s.act_p25 + (int32_t)pick({0, 1, 5, 30});
			s.b5 = (uint8_t)gyo_bucketid_from_threshold(GYO_RESP_TIME_HASH, s.p95_ms[0]);
			s.b300 = (uint8_t)gyo_bucketid_from_threshold(GYO_RESP_TIME_HASH, s.p95_ms[1]);
			s.b5day = (uint8_t)gyo_bucketid_from_threshold(GYO_RESP_TIME_HASH, s.p95_ms[2]);
			uint8_t mx = 0;
			for (int b = 0; b < 15; ++b) {
				s.nactive_conn_arr[b] = (uint8_t)pick({0, 0, 0, 0, 1, 1, 2, 3, 4, 9});
				if (mx < s.nactive_conn_arr[b]) mx = s.nactive_conn_arr[b];
			}
			const uint8_t extra = (uint8_t)pick({0, 0, 5, 16, 40});
			s.nconn_active = mx > extra ? mx : extra;
			s.glob_id = ((uint64_t)rng() << 32) | rng();
			gys_listener_issue_in &x = in[i];
			memset(&x, 0, sizeof(x));
			const uint32_t e = rng() % 100u;
			x.ser_errors = e < 55 ? 0u : e < 70 ? 1u : e < 80 ? (uint32_t)(cnt5 / 6) : e < 90 ? (uint32_t)(cnt5 / 3) : e < 98 ? (uint32_t)cnt5 : (1u << 31);
			x.tasks_delay_msec = rng() % 2u ? 0u : (uint32_t)((double)s.tsum[0] * pick({0.05, 0.2, 0.3, 2.0})) + (uint32_t)pick({0, 1000});
			x.tasks_cpudelay_msec = x.tasks_delay_msec / 3u;
			x.tasks_blkiodelay_msec = x.tasks_delay_msec / 4u;
			x.nconn = (int32_t)pick({0, 1, 4, 20, 300});
			x.ntasks_issue = (uint16_t)pick({0, 0, 1, 3});
			x.ntasks_noissue = (uint16_t)pick({0, 0, 1, 2});
			x.flags = (uint8_t)((rng() % 4u == 0 ? GYS_LI_TASK_ISSUE : 0) | (rng() % 3u == 0 ? GYS_LI_SEVERE : 0) | (rng() % 3u == 0 ? GYS_LI_DELAY : 0) |
					    (rng() % 3u == 0 ? GYS_LI_CPU_ISSUE : 0) | (rng() % 3u == 0 ? GYS_LI_MEM_ISSUE : 0) | (rng() % 5u == 0 ? GYS_LI_DEPENDS : 0) |
					    (rng() % 20u == 0 ? GYS_LI_YOUNG : 0));
			x.tdiff_start = (int64_t)pick({0, 50, 3600, 86400, 1e7});
		}
		ListenerDecideP p{};
		p.scan = sc.data();
		p.in = rnd == 1 ? nullptr : in.data(); // one round with the defaults of a NULL input array
		p.hist = hist.data();
		p.notify = notify.data();
		p.out = out.data();
		p.nsvc = N;
		p.msec1_bucket = gyo_bucketid_from_threshold(GYO_RESP_TIME_HASH, 1);
		kemu::launch((N + 255u) / 256u, 256, 0, [&] { k_listener_decide(p); });
		for (uint32_t i = 0; i < N; ++i) {
			gyo_listener_issue_in oi;
			if (rnd == 1) {
				memset(&oi, 0, sizeof(oi));
				oi.nconn = sc[i].nconn_active;
			} else {
				memcpy(&oi, &in[i], sizeof(oi));
			}
			gyo_listener_scan os;
			memcpy(&os, &sc[i], sizeof(os));
			gyo_listener_decision od;
			gyo_listener_decide(&os, &oi, &ohist[2 * i], &ohist[2 * i + 1], &od);
			lines.insert(od.decided_line);
			const gys_listener_decision &g = out[i];
			const uint8_t *r = &notify[(size_t)i * 88];
			uint32_t nf[4];
			uint16_t nti;
			memcpy(&nf[0], r + 44, 4); // ser_errors_
			memcpy(&nf[1], r + 52, 4); // tasks_delay_usec_
			memcpy(&nf[2], r + 56, 4); // tasks_cpudelay_usec_
			memcpy(&nf[3], r + 60, 4); // tasks_blkiodelay_usec_
			memcpy(&nti, r + 76, 2);   // ntasks_issue_
			const bool fields = nf[0] == oi.ser_errors && nf[1] == oi.tasks_delay_msec * 1000u && nf[2] == oi.tasks_cpudelay_msec * 1000u &&
					    nf[3] == oi.tasks_blkiodelay_msec * 1000u && nti == oi.ntasks_issue;
			if (!fields && fails++ < 20) printf("FAIL round %d listener %u: input fields of the notify record\n", rnd, i);
			const bool same = g.state == od.state && g.issue == od.issue && g.issue_bit_hist == od.issue_bit_hist && g.high_resp_bit_hist == od.high_resp_bit_hist &&
					  g.decided_line == od.decided_line && hist[2 * i] == ohist[2 * i] && hist[2 * i + 1] == ohist[2 * i + 1] && r[79] == od.state &&
					  r[80] == od.issue && r[81] == od.issue_bit_hist && r[82] == od.high_resp_bit_hist;
			if (!same && fails++ < 20)
				printf("FAIL round %d listener %u: kernel state %u issue %u at :%u hist %02x %02x, oracle %u %u at :%u hist %02x %02x\n", rnd, i, g.state, g.issue,
				       g.decided_line, g.issue_bit_hist, g.high_resp_bit_hist, od.state, od.issue, od.decided_line, od.issue_bit_hist, od.high_resp_bit_hist);
		}
	}
	if (lines.size() < 40) {
		printf("FAIL: only %zu deciding lines reached\n", lines.size());
		++fails;
	}
	if (fails) {
		printf("kemu ldecide: %d FAILURES\n", fails);
		return 1;
	}
	printf("kemu ldecide ok: %u listeners x 6 rounds, %zu deciding lines of the reference reached\n", N, lines.size());
	return 0;
}

// C++ exercise of the MCONN_HANDLER-shaped shim (gyeeta_amd/csrc/gys_mconn_shim.hpp) through the C ABI, compiled with plain g++
// (no HIP headers needed on the caller side).  argv[1] == "run" executes it (needs a GPU); without arguments it only proves the
// header + library link.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../gyeeta_amd/csrc/gys_mconn_shim.hpp"

#pragma pack(push, 1)
struct ListenerStateNotify { // comm::LISTENER_STATE_NOTIFY, 88 bytes (common/gy_comm_proto.h:2183-2254)
	uint64_t glob_id;
	uint32_t nqrys_5s, total_resp_5sec, nconns, nconns_active, ntasks, p95_5s, p95_5min, kb_in, kb_out, ser_errors, cli_errors;
	uint32_t tasks_delay, tasks_cpudelay, tasks_blkio, user_cpu, sys_cpu, rss_mb;
	uint16_t ntasks_issue;
	uint8_t is_http, curr_state, curr_issue, issue_bit_hist, high_resp_bit_hist, last_issue_subsrc, query_flags, issue_string_len, padding_len, tail;
};
#pragma pack(pop)
static_assert(sizeof(ListenerStateNotify) == 88, "LISTENER_STATE_NOTIFY is 88 bytes");
#pragma pack(push, 1)
struct ActiveConnStats { // comm::ACTIVE_CONN_STATS, 104 bytes (common/gy_comm_proto.h:2766-2783)
	uint64_t listener_glob_id, cli_aggr_task_id;
	char ser_comm[16], cli_comm[16];
	uint64_t remote_machine_id[2], remote_madhava_id, bytes_sent, bytes_received;
	uint32_t cli_delay_msec, ser_delay_msec;
	float max_rtt_msec;
	uint16_t active_conns;
	uint8_t flags, tail;
};
#pragma pack(pop)
static_assert(sizeof(ActiveConnStats) == 104, "ACTIVE_CONN_STATS is 104 bytes");

int main(int argc, char **argv)
{
	const bool rccl_mode = argc >= 2 && !strcmp(argv[1], "rccl");
	if (argc < 2 || (strcmp(argv[1], "run") && !rccl_mode)) {
		printf("link ok, abi %u\n", gys_abi_version());
		return 0;
	}
	gys_config cfg{};
	cfg.struct_size = sizeof(cfg);
	cfg.device = 0;
	cfg.nranks = 1;
	cfg.max_hosts = 4;
	cfg.max_services = 64;
	cfg.max_clusters = 2;
	cfg.enable_levels = 1;
	fprintf(stderr, "[shim] create\n");
	gyeeta_amd::GYS_MCONN_HANDLER h(cfg);
	uint8_t mid[16];
	for (int i = 0; i < 16; ++i) mid[i] = (uint8_t)(i * 7 + 1);
	if (!h.partha_register(mid, "prod")) return 2;
	std::vector<gys_listener_info> li(10);
	for (int i = 0; i < 10; ++i) {
		li[i] = gys_listener_info{};
		li[i].glob_id = 0x1000 + i;
		li[i].netns = 4026531840u;
		li[i].port = (uint16_t)(8000 + i);
		li[i].is_any_ip = 1;
		snprintf(li[i].comm, sizeof(li[i].comm), "svc%d", i);
	}
	if (!h.partha_new_listeners(mid, li.data(), 10)) return 3;
	alignas(8) ListenerStateNotify recs[10];
	memset(recs, 0, sizeof(recs));
	int exp_qps = 0, exp_active = 0;
	for (int i = 0; i < 10; ++i) {
		recs[i].glob_id = 0x1000 + i;
		recs[i].nqrys_5s = 7 * i; // tot_qps_ += nqrys_5s_/5 per record
		recs[i].curr_state = (uint8_t)(i % 6);
		recs[i].kb_in = 100;
		exp_qps += (7 * i) / 5;
		exp_active += i ? 1 : 0;
	}
	fprintf(stderr, "[shim] listener state\n");
	if (!h.partha_listener_state(mid, recs, 10, (const uint8_t *)(recs + 10))) return 4;
	gys_host_state st{};
	st.ntasks = 50;
	st.nlisten = 10;
	st.curr_state = 1;
	if (!h.partha_host_state(mid, st)) return 5;
	fprintf(stderr, "[shim] window 1\n");
	h.send_cluster_state(5000000);
	fprintf(stderr, "[shim] queries\n");
	gys_svcsumm s{};
	if (!h.get_listener_summ(mid, s)) return 6;
	gys_cluster_state c{};
	if (!h.get_cluster_state("prod", c)) return 7;
	printf("tot_qps %d nlisteners %d nactive %d cluster nhosts %u total_qps %u\n", s.tot_qps, s.nlisteners, s.nactive, c.nhosts, c.total_qps);
	if (s.tot_qps != exp_qps || s.nlisteners != 10 || s.nactive != exp_active || c.nhosts != 1 || c.total_qps != (uint32_t)exp_qps || c.nsvc != 10) return 8;
	if (rccl_mode) {
	// the multi-GPU form of the window boundary with the collective INSIDE the library: a one-rank RCCL communicator (this box has
		// one GPU) created through the C ABI; the all-reduced registers of a one-rank job equal the local ones
		{
			fprintf(stderr, "[shim] rccl\n");
			uint8_t uid[GYS_RCCL_UID_BYTES];
			if (gys_rccl_unique_id(uid) != GYS_OK) {
				fprintf(stderr, "gys_rccl_unique_id: %s\n", gys_last_error());
				return 18;
			}
			fprintf(stderr, "[shim] rccl join\n");
			if (!h.join_cluster(uid, 1, 0)) {
				fprintf(stderr, "join_cluster: %s\n", gys_last_error());
				return 19;
			}
			fprintf(stderr, "[shim] rccl joined\n");
			if (!h.partha_listener_state(mid, recs, 10, (const uint8_t *)(recs + 10)) || !h.partha_host_state(mid, st)) return 20;
			fprintf(stderr, "[shim] rccl window\n");
			h.send_cluster_state_rccl(15000000);
			fprintf(stderr, "[shim] rccl queries\n");
			gys_cluster_state c2{};
			gys_svcsumm s2{};
			if (!h.get_cluster_state("prod", c2) || !h.get_listener_summ(mid, s2)) return 21;
			if (c2.nhosts != 1 || c2.total_qps != (uint32_t)exp_qps || c2.nsvc != 10 || s2.tot_qps != exp_qps) {
				fprintf(stderr, "rccl window: nhosts %u total_qps %u nsvc %u tot_qps %d\n", c2.nhosts, c2.total_qps, c2.nsvc, s2.tot_qps);
				return 22;
			}
			printf("rccl window ok\n");
		}
		printf("shim rccl ok\n");
		return 0;
	}
	uint8_t other[16] = {9};
	if (h.partha_listener_state(other, recs, 10, (const uint8_t *)(recs + 10))) return 9; // unknown partha -> false (reference: null partha_shr)
	// the same window again through the wire front-end: one COMM_HEADER + EVENT_NOTIFY(NOTIFY_LISTENER_STATE) message, then the JSON query
	{
		alignas(8) uint8_t msg[16 + 8 + sizeof(recs)];
		const uint32_t hdr[4] = {0x05666605u /* PM_HDR_MAGIC */, (uint32_t)sizeof(msg), 14u /* COMM_EVENT_NOTIFY */, 0u};
		const uint32_t ev[2] = {0x309u /* NOTIFY_LISTENER_STATE */, 10u};
		memcpy(msg, hdr, 16);
		memcpy(msg + 16, ev, 8);
		memcpy(msg + 24, recs, sizeof(recs));
		fprintf(stderr, "[shim] stream\n");
		uint64_t used = 0;
		if (!h.handle_partha_stream(mid, msg, sizeof(msg), &used) || used != sizeof(msg)) return 10;
		if (!h.partha_host_state(mid, st)) return 11;
		h.send_cluster_state(10000000);
		std::string js;
		if (!h.web_curr_listener_summ(mid, "00112233aabbccdd", "2026-01-01T00:00:10+0000", js)) return 12;
		char want[64];
		snprintf(want, sizeof(want), "\"totqps\":%d,", exp_qps);
		if (js.find(want) == std::string::npos || js.find("\"nsvc\":10,") == std::string::npos || js.find("\"cluster\":\"prod\"") == std::string::npos) {
			fprintf(stderr, "unexpected json: %s\n", js.c_str());
			return 13;
		}
		if (!h.web_curr_clusterstate("ffffffffffffffff", "", js) || js.find("\"nhosts\":1,") == std::string::npos) return 14;
	}
	// multi-level windows: the two windows closed above (at 5 s and 10 s) carried no response events -> empty levels, ceiling of
	// bucket 1; the listener-state records fed the QPS / active-connection histograms twice
	{
		gys_time_hist_val tv[2] = {{0, 95.0f, 0}, {0, 25.0f, 0}};
		int64_t tcount = -1, tsum = -1;
		double mean = -1;
		if (h.get_resp_level_stats(0x1003, 2, 10, tv, 2, tcount, tsum, mean) != 0 || tcount != 0 || tsum != 0 || tv[0].data_value != 1) return 15;
		tcount = tsum = -1;
		if (h.get_resp_period_stats(0x1003, 2, 9, tv, 2, tcount, tsum, mean, 10) != 0 || tcount != 0 || tsum != 0 || tv[1].data_value != 1) return 18;
		gys_listener_day_stats ds[10];
		if (!h.listener_day_stats(10, 0, 10, ds)) return 16;
		// service 9: nqrys_5s = 63 -> two QPS samples of 12 (SEMI_LOG_HASH_LO bucket ceiling 50); no active connections -> ceiling 1
		if (ds[9].glob_id != 0x1009 || ds[9].tcount_5d != 0 || ds[9].p95_qps != 50 || ds[9].p95_nactive != 1) {
			fprintf(stderr, "day stats: gid %llx p95_qps %u p95_nactive %u\n", (unsigned long long)ds[9].glob_id, ds[9].p95_qps, ds[9].p95_nactive);
			return 17;
		}
	}
	// MCONN_HANDLER::handle_partha_active_conns (gy_mconnhdlr.cc:7705): three rows of listener 0x1002, one of them on another madhava
	{
		alignas(8) ActiveConnStats rows[3];
		memset(rows, 0, sizeof(rows));
		for (int i = 0; i < 3; ++i) {
			rows[i].listener_glob_id = 0x1002;
			rows[i].cli_aggr_task_id = 0x77;
			rows[i].bytes_sent = 1000;
			rows[i].bytes_received = 500;
			rows[i].active_conns = 4;
		}
		rows[2].flags = 2; // is_remote_listen_: the reference's remoteconntbl (gy_mconnhdlr.cc:7891) -> the second table pair (which 6 / 7)
		fprintf(stderr, "[shim] active conns\n");
		if (!h.handle_partha_active_conns(mid, rows, 3, (const uint8_t *)(rows + 3))) return 23;
		if (h.handle_partha_active_conns(other, rows, 3, (const uint8_t *)(rows + 3))) return 24; // unknown partha
		h.send_cluster_state(15000000);
		uint64_t conns = 0, bytes = 0;
		if (gys_query_pair_cms(h.ctx(), 0x1002, 0x77, 0, &conns) != GYS_OK || gys_query_pair_cms(h.ctx(), 0x1002, 0x77, 1, &bytes) != GYS_OK) return 25;
		gys_counters ctr{};
		if (gys_get_counters(h.ctx(), &ctr) != GYS_OK) return 26;
		uint64_t rconns = 0, rbytes = 0;
		if (gys_query_pair_cms(h.ctx(), 0x1002, 0x77, 6, &rconns) != GYS_OK || gys_query_pair_cms(h.ctx(), 0x1002, 0x77, 7, &rbytes) != GYS_OK) return 32;
		if (rconns != 4 || rbytes != 1500) {
			fprintf(stderr, "remote-listener rows: conns %llu bytes %llu\n", (unsigned long long)rconns, (unsigned long long)rbytes);
			return 33;
		}
		if (conns != 8 || bytes != 3000 || ctr.actconn_records != 2 || ctr.actconn_remote_listen != 1) {
			fprintf(stderr, "active conns: conns %llu bytes %llu local %llu remote %llu\n", (unsigned long long)conns, (unsigned long long)bytes,
				(unsigned long long)ctr.actconn_records, (unsigned long long)ctr.actconn_remote_listen);
			return 27;
		}
	}
	// MCONN_HANDLER::web_curr_top_listeners (gy_mnodehandle.cc:2706): the window closed above carried no listener records -> fresh ones
	{
		if (!h.partha_listener_state(mid, recs, 10, (const uint8_t *)(recs + 10)) || !h.partha_host_state(mid, st)) return 28;
		h.send_cluster_state(20000000);
		std::string js;
		fprintf(stderr, "[shim] top listeners\n");
		if (!h.web_curr_top_listeners(mid, GYS_TOP_QPS | GYS_TOP_SUMMSTATS, "00112233aabbccdd", "", js)) return 29;
		// top QPS of the host: service 9 (nqrys_5s 63 -> 12 qps) first
		const size_t at = js.find("\"topqps\":[{");
		if (at == std::string::npos || js.find("\"svcid\":\"0000000000001009\"", at) != js.find("\"svcid\":", at) || js.find("\"summstats\"") == std::string::npos ||
		    js.find("\"topissue\"") != std::string::npos) {
			fprintf(stderr, "unexpected top listeners json: %s\n", js.c_str());
			return 30;
		}
		if (!h.web_curr_top_listeners(nullptr, GYS_TOP_QPS, "00112233aabbccdd", "", js) || js.find("\"topqps\":[{") == std::string::npos) return 31; // all hosts
	}
	// TCP_SOCK_HANDLER::handle_ipv4_resp_event / handle_ipv6_resp_event (gy_socket_stat.cc:1517-1551) through the shim: an any-address IPv4
	// listener, a listener bound to an IPv6 address (gy_socket_stat.h:708-714: the event's server address has to equal it), one event to an
	// address nobody listens on
	{
		gys_listener_info b6{};
		b6.glob_id = 0x2000;
		b6.netns = 4026531840u;
		b6.port = 9000;
		b6.is_any_ip = 0;
		b6.addr_is_v6 = 1;
		const uint8_t a6[16] = {0x20, 0x01, 0x0d, 0xb8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 5};
		memcpy(b6.addr, a6, 16);
		snprintf(b6.comm, sizeof(b6.comm), "bound6");
		if (!h.partha_new_listeners(mid, &b6, 1)) return 34;
		struct Ev4 { uint32_t saddr, daddr, netns; uint16_t sport_be, dport_be; uint32_t lsndtime, lrcvtime; } e4[3];
		struct Ev6 { uint8_t saddr[16], daddr[16]; uint32_t netns; uint16_t sport_be, dport_be; uint32_t lsndtime, lrcvtime; } e6[5];
		static_assert(sizeof(Ev4) == 24 && sizeof(Ev6) == 48, "raw eBPF response events");
		auto be16 = [](uint16_t v) { return (uint16_t)((v >> 8) | (v << 8)); };
		for (int i = 0; i < 3; ++i) e4[i] = Ev4{0x0100000au, 0x0200000au + (uint32_t)i, 4026531840u, be16(8004), be16((uint16_t)(40000 + i)), 1010u, 1000u}; // 10 ms each
		for (int i = 0; i < 5; ++i) {
			memset(&e6[i], 0, sizeof(Ev6));
			memcpy(e6[i].saddr, a6, 16);
			e6[i].daddr[0] = 0xfd;
			e6[i].daddr[15] = (uint8_t)(i + 1);
			e6[i].netns = 4026531840u;
			e6[i].sport_be = be16(9000);
			e6[i].dport_be = be16((uint16_t)(50000 + i));
			e6[i].lsndtime = 5020u;
			e6[i].lrcvtime = 5000u; // 20 ms each
		}
		e6[4].saddr[15] = 6; // another server address: no listener takes it
		gys_counters c0{}, c1{};
		if (gys_get_counters(h.ctx(), &c0) != GYS_OK) return 35;
		fprintf(stderr, "[shim] response events\n");
		if (!h.handle_ipv4_resp_events(mid, e4, 3) || !h.handle_ipv6_resp_events(mid, e6, 5)) return 36;
		if (h.handle_ipv6_resp_events(other, e6, 5)) return 37; // unknown partha
		h.send_cluster_state(25000000);
		if (gys_get_counters(h.ctx(), &c1) != GYS_OK) return 38;
		gys_time_hist_val tv[1] = {{0, 95.0f, 0}};
		int64_t n4 = -1, s4 = -1, n6 = -1, s6 = -1;
		double mean = -1;
		if (h.get_resp_level_stats(0x1004, 0, 25, tv, 1, n4, s4, mean) != 0 || h.get_resp_level_stats(0x2000, 0, 25, tv, 1, n6, s6, mean) != 0) return 39;
		if (n4 != 3 || s4 != 30 || n6 != 4 || s6 != 80 || c1.resp_events - c0.resp_events != 8 || c1.resp_dropped_nolistener - c0.resp_dropped_nolistener != 1) {
			fprintf(stderr, "response events: v4 %lld / %lld v6 %lld / %lld events %llu dropped %llu\n", (long long)n4, (long long)s4, (long long)n6, (long long)s6,
				(unsigned long long)(c1.resp_events - c0.resp_events), (unsigned long long)(c1.resp_dropped_nolistener - c0.resp_dropped_nolistener));
			return 40;
		}
	}
	printf("shim ok\n");
	return 0;
}

// gys_mconn_shim.hpp -- C++17 host-side mirror of the reference's aggregation entry points on top of the C ABI.
//
// The reference has no plugin layer: the path under replacement is a handful of MCONN_HANDLER / SHCONN_HANDLER member functions
// invoked from the L2 dispatch switch (server/gy_mconnhdlr.cc:4756-4792).  This header gives a class with THOSE names, argument
// order and error behaviour (bool return, never throws across the boundary) so that the bodies of the reference functions can be
// replaced by one-line forwards (INTEGRATION.md shows the exact diff).  The reference types it cannot include here (comm::*,
// PARTHA_INFO) appear as opaque byte pointers with the same memory layout contract:
//
//   reference signature (server/gy_mconnhdlr.h)                                              -> shim
//   bool partha_tcp_conn_info(const std::shared_ptr<PARTHA_INFO>&, comm::TCP_CONN_NOTIFY*,      partha_tcp_conn_info(machine_id, pone, nconns, pendptr)
//                             int nconns, uint8_t *pendptr, POOL_ALLOC_ARRAY*)            :2091
//   bool partha_listener_state(const std::shared_ptr<PARTHA_INFO>&, const comm::LISTENER_STATE_NOTIFY*,  partha_listener_state(machine_id, pone, nconns, pendptr)
//                             int nconns, uint8_t *pendptr, POOL_ALLOC_ARRAY*, PGConnPool&, bool) :2129
//   void send_cluster_state() noexcept                                                    :2155  send_cluster_state_rccl(tusec) (RCCL inside the library) / send_cluster_state(tusec, reduce_cb)
//   TCP_SOCK_HANDLER::handle_ipv4_resp_event(tcp_ipv4_resp_event_t*, bool) (gy_socket_stat.cc:1517)  handle_ipv4_resp_events(machine_id, pevents, n)
//   TCP_SOCK_HANDLER::handle_ipv6_resp_event(tcp_ipv6_resp_event_t*, bool) (gy_socket_stat.cc:1535)  handle_ipv6_resp_events(machine_id, pevents, n)
//   web_curr_listener_summ (server/gy_mnodehandle.cc:1628)                                       get_listener_summ(machine_id, out)
//   bool handle_partha_active_conns(const std::shared_ptr<PARTHA_INFO>&, const comm::ACTIVE_CONN_STATS*,  handle_partha_active_conns(machine_id, pconn, nitems, pendptr)
//                             int nitems, uint8_t *pendptr, PGConnPool&)                 :2039 (.cc:7705, dispatch :5244)
//   bool web_curr_top_listeners(...) (server/gy_mnodehandle.cc:2706)                             web_curr_top_listeners(machine_id | nullptr, flags, madid, timestr, out)
//   TCP_SOCK_HANDLER::listener_stats_update(servshr, cpu_issue, mem_issue) (common/gy_socket_stat.cc:3895) listener_stats_update(tnow, qps_multiple, diffsec, d_notify, d_scan)
//
// PARTHA_INFO is identified by its GY_MACHINE_ID (PARTHA_INFO::machine_id_, the key of partha_tbl_).
#pragma once

#include <cstdint>
#include <ctime>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>

#include "../../include/gysketch.h"

namespace gyeeta_amd {

class GYS_MCONN_HANDLER {
public:
	// cb: performs the cross-GPU reduce of every gys_reduce_section (RCCL ncclAllReduce with ncclMax / ncclSum on the section's
	// dev_ptr); empty for a single GPU.  It is the analogue of NOTIFY_MS_CLUSTER_STATE -> SHCONN_HANDLER::aggregate_cluster_state.
	using ReduceFn = std::function<bool(const gys_reduce_section *secs, uint32_t nsecs)>;

	explicit GYS_MCONN_HANDLER(const gys_config &cfg)
	{
		if (gys_create(&cfg, &ctx_) != GYS_OK) throw std::runtime_error(std::string("gys_create: ") + gys_last_error());
	}
	~GYS_MCONN_HANDLER()
	{
		if (comm_) {
			gys_sync(ctx_);
			gys_rccl_comm_destroy(comm_);
		}
		gys_destroy(ctx_);
	}
	GYS_MCONN_HANDLER(const GYS_MCONN_HANDLER &) = delete;
	GYS_MCONN_HANDLER &operator=(const GYS_MCONN_HANDLER &) = delete;

	gys_ctx *ctx() noexcept { return ctx_; }

	// partha registration (PM_CONNECT) and NOTIFY_NEW_LISTENER: the two control-plane facts the data path needs
	bool partha_register(const uint8_t machine_id[16], const char *cluster_name) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_register_host(ctx_, machine_id, cluster_name, nullptr) == GYS_OK;
	}
	bool partha_new_listeners(const uint8_t machine_id[16], const gys_listener_info *arr, uint32_t n) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_register_listeners(ctx_, machine_id, arr, n, nullptr) == GYS_OK;
	}

	// MCONN_HANDLER::partha_tcp_conn_info: pone points into the L1 receive buffer, valid only during the call
	bool partha_tcp_conn_info(const uint8_t machine_id[16], const void *pone, int nconns, const uint8_t *pendptr) noexcept
	{
		if (!pone || nconns < 0) return false;
		std::shared_lock<std::shared_mutex> g(mu_); // up to 16 L2 threads call concurrently (gy_mconnhdlr.h:60); one context == one stream
		return gys_ingest_tcp_conn(ctx_, machine_id, pone, (uint32_t)nconns, pendptr) == GYS_OK;
	}

	// MCONN_HANDLER::partha_listener_state
	bool partha_listener_state(const uint8_t machine_id[16], const void *pone, int nconns, const uint8_t *pendptr) noexcept
	{
		if (!pone || nconns < 0) return false;
		std::shared_lock<std::shared_mutex> g(mu_);
		return gys_ingest_listener_state(ctx_, machine_id, pone, (uint32_t)nconns, pendptr) == GYS_OK;
	}

	// L1 + L2 in one step for a partha connection's receive buffer: COMM_HEADER-framed messages, records decoded on the GPU
	// (MCONN_HANDLER::handle_l1 -> handle_l2_misc dispatch, gy_mconnhdlr.cc:4700-4792).  nconsumed: whole messages consumed.
	bool handle_partha_stream(const uint8_t machine_id[16], const void *pbuf, uint64_t nbytes, uint64_t *nconsumed = nullptr) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		gys_comm_stats st{};
		const bool ok = gys_ingest_comm_stream(ctx_, machine_id, pbuf, nbytes, &st) == GYS_OK;
		if (nconsumed) *nconsumed = st.bytes_consumed;
		return ok;
	}

	// MCONN_HANDLER::handle_partha_active_conns (gy_mconnhdlr.cc:7705-7772, L2 dispatch :5244): a partha's 15-s ACTIVE_CONN_STATS rows
	bool handle_partha_active_conns(const uint8_t machine_id[16], const void *pconn, int nitems, const uint8_t *pendptr) noexcept
	{
		if (!pconn || nitems < 0) return false;
		std::shared_lock<std::shared_mutex> g(mu_);
		return gys_ingest_active_conns(ctx_, machine_id, pconn, (uint32_t)nitems, pendptr) == GYS_OK;
	}

	// comm::HOST_STATE_NOTIFY store read by send_cluster_state (gy_mconnhdlr.cc:16052-16075)
	bool partha_host_state(const uint8_t machine_id[16], const gys_host_state &st) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_ingest_host_state(ctx_, machine_id, &st) == GYS_OK;
	}

	// TCP_SOCK_HANDLER::handle_ipv4_resp_event for a batch of raw 24-byte events of one host
	bool handle_ipv4_resp_events(const uint8_t machine_id[16], const void *pevents, uint32_t nevents) noexcept
	{
		std::shared_lock<std::shared_mutex> g(mu_);
		return gys_ingest_resp_events(ctx_, machine_id, pevents, nevents) == GYS_OK;
	}

	// TCP_SOCK_HANDLER::handle_ipv6_resp_event (gy_socket_stat.cc:1535) for a batch of raw 48-byte tcp_ipv6_resp_event_t of one host
	bool handle_ipv6_resp_events(const uint8_t machine_id[16], const void *pevents, uint32_t nevents) noexcept
	{
		std::shared_lock<std::shared_mutex> g(mu_);
		return gys_ingest_resp_events_v6(ctx_, machine_id, pevents, nevents) == GYS_OK;
	}

	// MCONN_HANDLER::send_cluster_state (scheduled every 5000 ms, gy_mconnhdlr.cc:207-210) + the shyama-side aggregation
	void send_cluster_state(uint64_t tusec, const ReduceFn &reduce = {}) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		if (gys_window_prepare(ctx_, tusec) != GYS_OK) return;
		if (reduce) {
			gys_reduce_section secs[4];
			uint32_t n = 0;
			if (gys_reduce_sections(ctx_, secs, &n) == GYS_OK) (void)reduce(secs, n);
		}
		(void)gys_window_finish(ctx_);
	}

	// The multi-GPU form: one process per GPU, every rank joins the communicator once (rank 0 makes the id with gys_rccl_unique_id and
	// hands the 128 bytes to the others over whatever channel the deployment has) and then closes every window with the four
	// register families all-reduced over xGMI INSIDE the library (ncclAllReduce x 4 in one ncclGroup on the context stream).
	bool join_cluster(const uint8_t uid[GYS_RCCL_UID_BYTES], int nranks, int rank) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_rccl_comm_create(ctx_, uid, nranks, rank, &comm_) == GYS_OK;
	}
	void send_cluster_state_rccl(uint64_t tusec) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		if (comm_) (void)gys_window_close_rccl(ctx_, comm_, tusec);
	}

	// web_curr_listener_summ: LISTEN_SUMM_STATS<int> of one partha (fields map 1:1 onto svcsumm JSON, gy_mfields.h:768-790)
	bool get_listener_summ(const uint8_t machine_id[16], gys_svcsumm &out) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_query_svcsumm(ctx_, machine_id, &out) == GYS_OK;
	}
	bool get_cluster_state(const char *cluster, gys_cluster_state &out) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_query_clusterstate(ctx_, cluster, &out) == GYS_OK;
	}

	// RESP_TIME_HISTOGRAM::get_stats_with_flush of one listener on one time level (Level_5s_5min_5days_all index 0..3;
	// common/gy_statistics.h:1333-1374); needs gys_config.enable_levels
	int get_resp_level_stats(uint64_t glob_id, int level, time_t tnow, gys_time_hist_val *pstats, size_t nstats, int64_t &tcount, int64_t &tsum,
				 double &mean_val) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_query_hist_level_stats(ctx_, glob_id, level, (uint64_t)tnow * 1000000ull, pstats, (uint32_t)nstats, &tcount, &tsum, &mean_val) == GYS_OK
			       ? 0
			       : -1;
	}

	// RESP_TIME_HISTOGRAM::get_stats_for_period_with_flush of one listener (common/gy_statistics.h:1378-1413)
	int get_resp_period_stats(uint64_t glob_id, time_t starttime, time_t endtime, gys_time_hist_val *pstats, size_t nstats, int64_t &tcount, int64_t &tsum,
				  double &mean_val, time_t tnow = time(nullptr)) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_query_hist_period_stats(ctx_, glob_id, (int64_t)starttime, (int64_t)endtime, (uint64_t)tnow * 1000000ull, pstats, (uint32_t)nstats,
						   &tcount, &tsum, &mean_val) == GYS_OK
			       ? 0
			       : -1;
	}

	// the NOTIFY_LISTENER_DAY_STATS payload (comm::LISTENER_DAY_STATS[], MAX_NUM_LISTENERS = 2048 per message) for service slots
	// [first_slot, first_slot + nslots), as TCP_LISTENER::get_curr_state fills it (common/gy_socket_stat.cc:2098-2112)
	bool listener_day_stats(time_t tnow, uint32_t first_slot, uint32_t nslots, gys_listener_day_stats *pout) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_export_day_stats(ctx_, (uint64_t)tnow * 1000000ull, first_slot, nslots, pout) == GYS_OK;
	}

	// the same answers as the reference's web JSON (web_curr_listener_summ / web_curr_listener_state / web_curr_clusterstate)
	bool web_curr_listener_summ(const uint8_t machine_id[16], const char *madid, const char *timestr, std::string &out) noexcept
	{
		return json_call(out, [&](char *b, size_t n, size_t *need) { return gys_json_svcsumm(ctx_, machine_id, madid, timestr, b, n, need); });
	}
	bool web_curr_listener_state(const uint8_t machine_id[16], const char *madid, const char *timestr, std::string &out) noexcept
	{
		return json_call(out, [&](char *b, size_t n, size_t *need) { return gys_json_svcstate(ctx_, machine_id, madid, timestr, b, n, need); });
	}
	// MCONN_HANDLER::web_curr_listener_state with QUERY_OPTIONS (server/gy_mnodehandle.cc:4650-4900; common/gy_query_common.h:24-140): the
	// multi-host form -- criteria on the numeric svcstate columns (filter: groups of terms as CRITERIA_SET holds them, nullptr = none),
	// one sort column (GYS_SVC_COL_*, -1 = service order), maxrecs; one pass over the kept states of every listener on the device
	bool web_curr_listener_state_multihost(const gys_svc_filter *filter, int sort_col, bool sort_desc, uint32_t maxrecs, const char *madid,
					       const char *timestr, std::string &out) noexcept
	{
		return json_call(out, [&](char *b, size_t n, size_t *need) {
			return gys_json_svcstate_multihost(ctx_, filter, sort_col, sort_desc ? 1 : 0, maxrecs, madid, timestr, b, n, need);
		});
	}
	// MCONN_HANDLER::web_curr_listener_summ, multi-host form (server/gy_mnodehandle.cc:1628-1690): every partha with a recent listener state
	// whose LISTEN_SUMM_STATS row passes the criteria (terms name GYS_SUMM_COL_*)
	bool web_curr_listener_summ_multihost(const gys_svc_filter *filter, int sort_col, bool sort_desc, uint32_t maxrecs, const char *madid,
					      const char *timestr, std::string &out) noexcept
	{
		return json_call(out, [&](char *b, size_t n, size_t *need) {
			return gys_json_svcsumm_multihost(ctx_, filter, sort_col, sort_desc ? 1 : 0, maxrecs, madid, timestr, b, n, need);
		});
	}
	// AOPER_PERCENTILE of one column over the matching listeners (discrete percentiles, exact)
	bool aggr_listener_state_percentiles(const gys_svc_filter *filter, int col, const double *pcts, uint32_t npcts, int64_t *out, uint64_t *nmatched) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_query_svcstate_percentiles(ctx_, filter, col, pcts, npcts, out, nmatched) == GYS_OK;
	}
	// a string criterion on the service name ({ svcstate.name like 'post' }: CRITERION_ONE::match_str_criterian common/gy_query_criteria.h:1335-1383)
	// resolved into the service ids a filter's `svcids` takes; comp = GYS_COMP_EQ / NEQ / SUBSTR / NOTSUBSTR / LIKE / NOTLIKE / IN / NOTIN
	bool svc_ids_by_name(int comp, const std::vector<std::string> &patterns, std::vector<uint64_t> &ids) noexcept
	{
		try {
			std::shared_lock<std::shared_mutex> g(mu_);
			std::vector<const char *> p;
			for (const auto &s : patterns) p.push_back(s.c_str());
			uint32_t n = 0;
			ids.resize(1024);
			int rc = gys_svc_ids_by_name(ctx_, comp, p.data(), (uint32_t)p.size(), ids.data(), (uint32_t)ids.size(), &n);
			if (rc == GYS_ERR_NOMEM) {
				ids.resize(n);
				rc = gys_svc_ids_by_name(ctx_, comp, p.data(), (uint32_t)p.size(), ids.data(), (uint32_t)ids.size(), &n);
			}
			if (rc != GYS_OK) return false;
			ids.resize(n);
			return true;
		} catch (...) {
			return false;
		}
	}
	// ... on the host name ({ svcstate.host substr 'db' }): the machine ids (16 bytes each) a filter's `machine_ids` takes
	bool machine_ids_by_hostname(int comp, const std::vector<std::string> &patterns, std::vector<uint8_t> &ids16) noexcept
	{
		try {
			std::shared_lock<std::shared_mutex> g(mu_);
			std::vector<const char *> p;
			for (const auto &s : patterns) p.push_back(s.c_str());
			uint32_t n = 0;
			ids16.resize(16 * 256);
			int rc = gys_machine_ids_by_hostname(ctx_, comp, p.data(), (uint32_t)p.size(), ids16.data(), (uint32_t)(ids16.size() / 16), &n);
			if (rc == GYS_ERR_NOMEM) {
				ids16.resize((size_t)16 * n);
				rc = gys_machine_ids_by_hostname(ctx_, comp, p.data(), (uint32_t)p.size(), ids16.data(), n, &n);
			}
			if (rc != GYS_OK) return false;
			ids16.resize((size_t)16 * n);
			return true;
		} catch (...) {
			return false;
		}
	}
	// the aggregation operators (AGGR_OPER_E, common/gy_json_field_maps.h:114-129) over the matching listeners: group_by 0 all / 1 host / 2 cluster
	bool aggr_listener_state(const gys_svc_filter *filter, int group_by, const uint8_t *cols, uint32_t ncols, gys_svc_aggr_row *out, uint32_t maxrows,
				 uint32_t *nrows) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_); // (a query: not concurrent with ingest calls)
		return gys_query_svcstate_aggr(ctx_, filter, group_by, cols, ncols, out, maxrows, nrows) == GYS_OK;
	}
	// MCONN_HANDLER::web_curr_top_listeners (server/gy_mnodehandle.cc:2706-3190): machine_id = one partha's four top-10 queues,
	// nullptr = every host's queues merged into MAX_MULTI_TOPN = 50 slots per kind; flags: GYS_TOP_* (which arrays to send)
	bool web_curr_top_listeners(const uint8_t *machine_id, uint32_t flags, const char *madid, const char *timestr, std::string &out) noexcept
	{
		return json_call(out, [&](char *b, size_t n, size_t *need) { return gys_json_toplisteners(ctx_, machine_id, flags, madid, timestr, b, n, need); });
	}
	// TCP_SOCK_HANDLER::listener_stats_update (common/gy_socket_stat.cc:3895-4380): the 5-s walk over every listener, from the engine's
	// own state: d_notify = device buffer of 88 B x services (LISTENER_STATE_NOTIFY records, ready for partha_listener_state's device
	// form), d_scan = gys_listener_scan x services; either may be nullptr.  Needs gys_config.enable_levels.
	bool listener_stats_update(time_t tnow, float qps_multiple, uint32_t diffsec, void *d_notify, gys_listener_scan *d_scan) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_scan_listener_state_dev(ctx_, (uint64_t)tnow * 1000000ull, qps_multiple, diffsec, d_notify, d_scan) == GYS_OK;
	}
	// TCP_LISTENER::get_curr_state for every listener (common/gy_socket_stat.cc:2020-2870, called at :4241) + the caller's history byte and
	// "just started" rule (:4244-4266): d_scan as listener_stats_update left it, d_issue_in = the task / host inputs per service slot (nullptr:
	// none), curr_state_ / curr_issue_ / issue_bit_hist_ / high_resp_bit_hist_ patched into d_notify, the decisions into d_out (either may be nullptr)
	bool listener_curr_states(const gys_listener_scan *d_scan, const gys_listener_issue_in *d_issue_in, void *d_notify, gys_listener_decision *d_out) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		return gys_decide_listener_state_dev(ctx_, d_scan, d_issue_in, d_notify, d_out) == GYS_OK;
	}
	bool web_curr_clusterstate(const char *shyamaid, const char *timestr, std::string &out) noexcept
	{
		return json_call(out, [&](char *b, size_t n, size_t *need) { return gys_json_clusterstate(ctx_, shyamaid, timestr, b, n, need); });
	}

private:
	template <typename F>
	bool json_call(std::string &out, F &&f) noexcept
	{
		std::unique_lock<std::shared_mutex> g(mu_);
		try {
			size_t need = 0;
			int rc = f(nullptr, 0, &need);
			if (rc != GYS_OK && rc != GYS_ERR_NOMEM) return false;
			out.resize(need + 1);
			rc = f(out.data(), out.size(), &need);
			out.resize(need);
			return rc == GYS_OK;
		} catch (...) {
			return false;
		}
	}

	gys_ctx *ctx_ = nullptr;
	void *comm_ = nullptr; // ncclComm_t
	std::shared_mutex mu_; // ingest calls share it (thread-safe among themselves inside the library); registration, window close and queries are exclusive
};

} // namespace gyeeta_amd

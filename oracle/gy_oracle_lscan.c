/* TEST INFRASTRUCTURE -- CPU restatement of the per-listener 5-second scan, used only as the checker (tests/, smoke, bench cpu leg).
 *
 * TCP_SOCK_HANDLER::listener_stats_update (common/gy_socket_stat.cc:4044-4365) walks every listener every 5 s and turns its counters
 * and histograms into one comm::LISTENER_STATE_NOTIFY (common/gy_comm_proto.h:2183-2254); TCP_LISTENER::get_curr_state (:2030-2143)
 * then compares the 5-s p95 bucket with the 5-min / 5-day ones and the current QPS with the QPS histogram's p25 / p95.  This file
 * restates the DATA-PARALLEL part of both -- everything that is a pure function of the listener's histograms, its CONN_BITMAP and
 * its query counter -- and nothing of the state policy (task / cpu / memory issue inputs, issue strings, dependency resolution).
 *
 * Parity: the level arithmetic rests on gy_oracle_levels.c (folly restated: PARITY UNPINNED); the percentile rules
 * (getPercentileBucketIdx, GY_HISTOGRAM::get_percentiles), CONN_BITMAP::get_conn_breakup and get_bucketid_from_threshold are pinned
 * against oracle/_ref (tests/test_oracle_vs_ref.py, tests/test_oracle_lscan.py). */
#include <string.h>

#include "gy_oracle.h"

/* get_bucketid_from_threshold<RESP_TIME_HASH> (common/gy_statistics.h:517-531): the bucket whose ceiling is `threshold` */
uint32_t gyo_bucketid_from_threshold(int kind, int64_t threshold)
{
	const int nb = gyo_hist_nbuckets(kind);
	for (int i = 1; i < nb - 1; i++)
		if (gyo_bucket_max_threshold(kind, (size_t)i) == threshold) return (uint32_t)i;
	if (threshold < gyo_bucket_max_threshold(kind, 0) + 1) return 0; /* threshold < HashClass::min_value */
	return (uint32_t)(nb - 1);
}

/* One listener.  resp: its TIME_HISTOGRAM flushed to the scan time; qps / act: its QPS_HISTOGRAM / ACTIVE_CONN_HISTOGRAM;
 * respmap: its CONN_BITMAP rows of the window just closed (32 rows of resp_bitmap_v4_, then 32 of resp_bitmap_v6_); multiple / diffsec: get_bpf_qps_multiple() and the seconds since the
 * previous scan (:4046, :4109).  Fills the notify record's derivable fields (everything else zero) and the scan record. */
void gyo_listener_scan_one(const gyo_mlhist *resp, const gyo_hist *qps, const gyo_hist *act, const uint16_t respmap[64], uint64_t glob_id,
			   float multiple, int64_t diffsec, uint8_t notify[88], gyo_listener_scan *out)
{
	static const float pcts[3] = {95.0f, 99.0f, 25.0f}; /* RESP_STATS::stats_ (common/gy_socket_stat.h:459-462) */
	memset(out, 0, sizeof(*out));
	memset(notify, 0, 88);
	out->glob_id = glob_id;
	for (int lv = 0; lv < GYO_MLH_LEVELS; lv++) { /* :4236-4240: resp_hist.get_stats(dist_seconds[i], histstat_[i]...) */
		int64_t v[3], tc, ts;
		gyo_mlh_get_stats(resp, lv, pcts, 3, v, &tc, &ts, NULL);
		out->tcount[lv] = tc;
		out->tsum[lv] = ts;
		out->p95_ms[lv] = (int32_t)v[0];
		out->p99_ms[lv] = (int32_t)v[1];
		out->p25_ms[lv] = (int32_t)v[2];
	}
	/* total_queries = the listener's query counter over the interval (:4051-4052) = one per response event that reached the
	 * histogram (:1581) = the 5-s level's count; curr_qps_extra = total_queries * multiple_factor / diffsec (:4109) */
	const uint32_t total_queries = (uint32_t)out->tcount[0];
	out->last_qps = diffsec > 0 ? (int32_t)((float)total_queries * multiple / (float)diffsec) : 0;
	{
		const int32_t q5 = (int32_t)(out->tcount[0] / 5);
		out->curr_qps = out->last_qps > q5 ? out->last_qps : q5; /* :2083 */
	}
	out->b5 = (uint8_t)gyo_bucketid_from_threshold(GYO_RESP_TIME_HASH, out->p95_ms[0]); /* :2085-2087 */
	out->b300 = (uint8_t)gyo_bucketid_from_threshold(GYO_RESP_TIME_HASH, out->p95_ms[1]);
	out->b5day = (uint8_t)gyo_bucketid_from_threshold(GYO_RESP_TIME_HASH, out->p95_ms[2]);
	{
		gyo_hist_data pd[2] = {{0, 0, 0, 95.0f}, {0, 0, 0, 25.0f}}; /* HIST_DATA stats_qps[] {95, 25} (:2053, :2089-2090) */
		uint64_t tot;
		int64_t mx;
		gyo_hist_percentiles(qps, pd, 2, &tot, &mx, NULL);
		out->qps_p95 = (int32_t)pd[0].data_value;
		out->qps_p25 = (int32_t)pd[1].data_value;
		pd[0].percentile = 95.0f;
		pd[1].percentile = 25.0f;
		gyo_hist_percentiles(act, pd, 2, &tot, &mx, NULL);
		out->act_p95 = (int32_t)pd[0].data_value;
		out->act_p25 = (int32_t)pd[1].data_value;
	}
	/* :4143-4156: nactive_conn_arr_[r] = rows of the CONN_BITMAP with bit r; curr_active_conn = their maximum (the inet_diag count
	 * nconn_recent_active_ it starts from is agent-side state the engine does not hold: it starts from 0) */
	gyo_conn_bitmap_breakup2(respmap, out->nactive_conn_arr); /* ipv4_conn[r] + ipv6_conn[r] :4144-4149 */
	for (int r = 0; r < 15; r++)
		if (out->nconn_active < out->nactive_conn_arr[r]) out->nconn_active = out->nactive_conn_arr[r];
	/* :4293-4304 the notify record */
	{
		uint32_t u;
		memcpy(notify + 0, &glob_id, 8);
		u = (uint32_t)out->tcount[0];
		memcpy(notify + 8, &u, 4); /* nqrys_5s_ = histstat_[n5].tcount_ */
		u = (uint32_t)out->tsum[0];
		memcpy(notify + 12, &u, 4); /* total_resp_5sec_ = histstat_[n5].tsum_ */
		u = out->nconn_active;
		memcpy(notify + 16, &u, 4); /* nconns_ = last_chk_nconn_: not below the active count */
		memcpy(notify + 20, &u, 4); /* nconns_active_ = last_chk_nconn_active_ */
		u = (uint32_t)out->p95_ms[0];
		memcpy(notify + 28, &u, 4); /* p95_5s_resp_ms_ */
		u = (uint32_t)out->p95_ms[1];
		memcpy(notify + 32, &u, 4); /* p95_5min_resp_ms_ */
		notify[79] = out->curr_qps == 0 ? 0 /* STATE_IDLE (:2115-2130 without task / error inputs) */ : 2 /* STATE_OK */;
	}
}

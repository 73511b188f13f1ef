/* TEST INFRASTRUCTURE -- CPU restatement of TCP_LISTENER::get_curr_state (common/gy_socket_stat.cc:2020-2870), used only as the checker.
 *
 * The reference decides a listener's OBJ_STATE_E / LISTENER_ISSUE_SRC every 5 s from (a) what the per-listener scan produces -- the
 * percentiles, counts and sums of the four response-time levels, the QPS / active-connection percentiles, the bucket ids, the CONN_BITMAP
 * break-up (gy_oracle_lscan.c) -- and (b) inputs the listener's histograms do not hold: task status (is_task_issue, :2043), host CPU /
 * memory issue flags, server errors, the connection count, the number of dependent servers.  (b) arrives as gyo_listener_issue_in.
 * Every branch below cites the reference's lines; the issue STRING (STR_WR_BUF) is not produced.  The return value is the line of the
 * reference's `return` (or of the function's end) that decided, so that a test can say which branch it exercised.
 *
 * C arithmetic is kept as the reference's: `ser_errors * 2` is a 32-bit unsigned product compared with a size_t; `x * 1.1f` with an
 * int64_t x is a float product; `mean * 0.8f` with a double mean is a double product.
 *
 * Parity: the policy cannot be pinned against the reference compiled in place (TCP_LISTENER needs the whole socket handler); it is
 * pinned by a table of hand-derived cases, one per return statement (tests/test_oracle_lstate.py): PARITY UNPINNED otherwise. */
#include <string.h>

#include "gy_oracle.h"

enum { ST_IDLE = 0, ST_GOOD = 1, ST_OK = 2, ST_BAD = 3, ST_SEVERE = 4 };                                       /* OBJ_STATE_E */
enum { IS_NONE = 0, IS_TASKS = 1, IS_QPS_HIGH = 2, IS_ACTIVE_CONN_HIGH = 3, IS_SERVER_ERRORS = 4, IS_DEPENDS = 7, IS_UNKNOWN = 8 }; /* LISTENER_ISSUE_SRC */

#define DECIDE(st, is, line) do { *state = (uint8_t)(st); *issue = (uint8_t)(is); return (line); } while (0)

int gyo_listener_curr_state(const gyo_listener_scan *sc, const gyo_listener_issue_in *in, uint8_t *high_resp_bit_hist, uint8_t *state, uint8_t *issue)
{
	const int task_issue = (in->flags & GYO_LI_TASK_ISSUE) != 0, is_severe = (in->flags & GYO_LI_SEVERE) != 0, is_delay = (in->flags & GYO_LI_DELAY) != 0;
	const int cpu_issue = (in->flags & GYO_LI_CPU_ISSUE) != 0, mem_issue = (in->flags & GYO_LI_MEM_ISSUE) != 0;
	const int ntasks_issue = in->ntasks_issue, ntasks_noissue = in->ntasks_noissue;
	const uint32_t ser_errors = in->ser_errors;
	const uint64_t tasks_delay_msec = in->tasks_delay_msec;                    /* :2047 */
	const int nconn = in->nconn;                                               /* :2040 */
	const int curr_active_conn = sc->nconn_active;                             /* the caller's curr_active_conn (:4143-4156) */
	/* :2076-2091 */
	const size_t nqrys_5s = (size_t)sc->tcount[0];
	const int64_t r5p95 = sc->p95_ms[0], r5p99 = sc->p99_ms[0], r5daysp95 = sc->p95_ms[2], r5daysp99 = sc->p99_ms[2], rallp95 = sc->p95_ms[3];
	const int curr_qps = sc->curr_qps;
	const size_t b5 = sc->b5, b300 = sc->b300, b5day = sc->b5day;
	const int64_t qps_p95 = sc->qps_p95, qps_p25 = sc->qps_p25, act_p95 = sc->act_p95, act_p25 = sc->act_p25; /* stats_qps[0/1], stats_active[0/1] */
	const size_t msec1_bucket = gyo_bucketid_from_threshold(GYO_RESP_TIME_HASH, 1);                             /* :2062 */
	double mean[GYO_MLH_LEVELS];                                                                                /* histstat_[i].mean_val_ (gy_statistics.h:1361) */
	for (int i = 0; i < GYO_MLH_LEVELS; i++) mean[i] = (double)sc->tsum[i] / (double)(sc->tcount[i] != 0 ? sc->tcount[i] : 1);
	const int64_t sec_dist_5d = in->tdiff_start > 0 && in->tdiff_start < 5 * 24 * 3600 ? in->tdiff_start : 5 * 24 * 3600; /* :2064-2071 */
#define HIGH_B (b5 > b5day + 2 && b5 > b300)

	*state = ST_OK;
	*issue = IS_NONE;
	*high_resp_bit_hist = (uint8_t)(*high_resp_bit_hist << 1); /* :2113 */

	if (curr_qps == 0) { /* :2115 */
		if (!task_issue || !is_severe || !ser_errors) DECIDE(ST_IDLE, IS_NONE, 2126);
	}
	const uint64_t total_resp_msec = (uint64_t)sc->tsum[0]; /* :2130 */

	if (b5 == msec1_bucket || r5p95 < r5daysp95) { /* :2132 */
		if ((int64_t)curr_qps <= qps_p25 && qps_p25 < qps_p95) { /* :2136 */
			if (!task_issue && !ser_errors) DECIDE(ST_IDLE, IS_NONE, 2144);
			else if (!task_issue && ser_errors) {
				if ((size_t)(uint32_t)(ser_errors * 2u) > nqrys_5s) DECIDE(ST_SEVERE, IS_SERVER_ERRORS, 2153);
				else if ((size_t)(uint32_t)(ser_errors * 5u) > nqrys_5s) DECIDE(ST_BAD, IS_SERVER_ERRORS, 2161);
				else if ((double)ser_errors < (double)nqrys_5s * 0.1) DECIDE(ST_OK, IS_SERVER_ERRORS, 2169);
			} else { /* a task issue */
				if ((size_t)(uint32_t)(ser_errors * 2u) > nqrys_5s) DECIDE(ST_SEVERE, IS_SERVER_ERRORS, 2179);
				else if ((size_t)(uint32_t)(ser_errors * 5u) > nqrys_5s) DECIDE(ST_BAD, IS_SERVER_ERRORS, 2187);
				else if (ser_errors) DECIDE(ST_BAD, IS_TASKS, 2202);
				if (is_severe && ntasks_issue > 0 && ntasks_noissue == 0) DECIDE(ST_BAD, IS_TASKS, 2213);
				if ((int64_t)nconn > act_p25) DECIDE(ST_OK, IS_TASKS, 2224);
			}
		}
		if (ser_errors) { /* :2229 */
			if ((size_t)(uint32_t)(ser_errors * 2u) > nqrys_5s) DECIDE(ST_SEVERE, IS_SERVER_ERRORS, 2243);
			else if ((size_t)(uint32_t)(ser_errors * 5u) > nqrys_5s) DECIDE(ST_BAD, IS_SERVER_ERRORS, 2257);
		}
		if (task_issue && is_severe && ntasks_issue > 0 && ntasks_noissue == 0) DECIDE(ST_BAD, IS_TASKS, 2273); /* :2261 */
		if (!ser_errors) { /* :2276 */
			if ((int64_t)curr_qps <= qps_p95 || b5 + 2 <= b5day) DECIDE(ST_GOOD, IS_NONE, 2305);
			else if ((int64_t)curr_qps > qps_p95) DECIDE(ST_OK, IS_QPS_HIGH, 2305);
		} else {
			DECIDE(ST_OK, IS_SERVER_ERRORS, 2305);
		}
		return 2305;
	}

	if (r5p95 == r5daysp95) { /* :2308 */
		if (ser_errors) {
			if ((size_t)(uint32_t)(ser_errors * 2u) > nqrys_5s) DECIDE(ST_SEVERE, IS_SERVER_ERRORS, 2322);
			else if ((size_t)(uint32_t)(ser_errors * 5u) > nqrys_5s) DECIDE(ST_BAD, IS_SERVER_ERRORS, 2336);
		}
		if (mean[0] <= mean[2] * 0.8f) { /* :2340 */
			if ((int64_t)curr_qps <= qps_p25) {
				if (ser_errors) DECIDE(ST_BAD, IS_SERVER_ERRORS, 2356);
				else if (!task_issue) DECIDE(ST_IDLE, IS_NONE, 2364);
				else if (ntasks_issue > 0 && ntasks_noissue == 0) DECIDE(ST_BAD, IS_TASKS, 2374);
				else if (ntasks_issue > 0 && tasks_delay_msec >= 1000) DECIDE(ST_BAD, IS_TASKS, 2384);
			}
			if (!task_issue && !ser_errors) DECIDE(ST_GOOD, IS_NONE, 2394);
			else if (ser_errors) {
				if (task_issue) DECIDE(ST_BAD, IS_TASKS, 2403);
				/* :2406-2410 sets ISSUE_SERVER_ERRORS / STATE_OK and does NOT return: the statements below overwrite both */
			}
			DECIDE(ST_OK, IS_TASKS, 2417);
		}
		if (mean[0] <= mean[2] * 1.2f) DECIDE(ST_OK, IS_NONE, 2427); /* :2419 */
	}

	*high_resp_bit_hist |= 1; /* :2431 */

	if (ser_errors) { /* :2433 */
		if ((size_t)(uint32_t)(ser_errors * 2u) > nqrys_5s) DECIDE(ST_SEVERE, IS_SERVER_ERRORS, 2447);
		else if ((size_t)(uint32_t)(ser_errors * 5u) > nqrys_5s) DECIDE(ST_BAD, IS_SERVER_ERRORS, 2461);
	}
	/* QPS too high (:2466) */
	if ((int64_t)curr_qps > qps_p95 && (int64_t)curr_qps - qps_p95 > 5 && (float)curr_qps > (float)qps_p95 * 1.1f)
		DECIDE(HIGH_B ? ST_SEVERE : ST_BAD, IS_QPS_HIGH, 2492);
	/* task issue, or delays of several listener processes that are at least a quarter of the response time (:2498) */
	if (task_issue || (is_delay && ntasks_issue + ntasks_noissue > 2 && tasks_delay_msec * 4 > total_resp_msec))
		DECIDE(HIGH_B ? ST_SEVERE : ST_BAD, IS_TASKS, 2525);
	/* active connections too high (:2529) */
	if ((int64_t)curr_active_conn > act_p95 && (int64_t)curr_active_conn - act_p95 > 1)
		DECIDE(HIGH_B && curr_active_conn > 10 ? ST_SEVERE : ST_BAD, IS_ACTIVE_CONN_HIGH, 2552);
	if (r5p95 == r5daysp95) { /* :2555 */
		if (r5p99 > r5daysp99) DECIDE(ST_OK, ser_errors ? IS_SERVER_ERRORS : IS_NONE, 2571);
	}
	/* QPS low and connections below the p25 of the active connections (:2576) */
	if ((int64_t)curr_qps <= qps_p25 && (int64_t)nconn <= act_p25) {
		if (is_delay && cpu_issue && mem_issue) DECIDE(ST_BAD, IS_TASKS, 2593);
		else if (is_delay && (cpu_issue || mem_issue) && tasks_delay_msec * 4 > total_resp_msec) DECIDE(ST_BAD, IS_TASKS, 2611);
		DECIDE(ST_OK, ser_errors ? IS_SERVER_ERRORS : IS_NONE, 2630);
	}
	/* the average QPS of the last 5 days is less than half the current one (:2638-2657) */
	{
		const int avg_5day_qps = (int)(sc->tcount[2] / sec_dist_5d);
		if (avg_5day_qps < (curr_qps >> 1) && r5p95 <= rallp95 && mean[0] <= mean[3] * 1.1f) DECIDE(ST_OK, ser_errors ? IS_SERVER_ERRORS : IS_NONE, 2657);
	}
	if ((int64_t)curr_qps <= qps_p25 && (int64_t)curr_active_conn <= act_p25 && b5 <= b5day + 1) /* :2661 */
		DECIDE(ST_OK, ser_errors ? IS_SERVER_ERRORS : IS_NONE, 2679);
	if (b5 <= b5day + 1 && b300 == b5day) { /* :2684 */
		if (mean[0] > mean[1] && mean[1] < mean[2] * 1.1f) DECIDE(ST_OK, ser_errors ? IS_SERVER_ERRORS : IS_NONE, 2702);
	}
	if (curr_active_conn >= 15 && b5 == b5day + 1) { /* :2710: few connections in the slow buckets */
		size_t b;
		for (b = b5; b < 15; ++b) /* RESP_TIME_HASH::max_buckets */
			if (sc->nactive_conn_arr[b] > 3) break;
		if (b > b5) DECIDE(ST_OK, ser_errors ? IS_SERVER_ERRORS : IS_NONE, 2738);
	}
	{ /* :2745-2768: high only up to half of the last iterations */
		uint32_t bithist = *high_resp_bit_hist;
		int nhigh = 0;
		for (; bithist; bithist &= bithist - 1) nhigh++;
		if (nhigh < 5) DECIDE(ST_OK, ser_errors ? IS_SERVER_ERRORS : IS_NONE, 2768);
	}
	/* :2774-2866: every explanation is exhausted */
	*state = HIGH_B ? ST_SEVERE : ST_BAD;
	{
		const uint32_t tasks_cpudelay_msec = in->tasks_cpudelay_msec, tasks_blkiodelay_msec = in->tasks_blkiodelay_msec;
		if (tasks_delay_msec * 4 > total_resp_msec && *state == ST_BAD) DECIDE(ST_BAD, IS_TASKS, 2817);
		(void)tasks_cpudelay_msec; /* (they only choose the words of the issue string, :2797-2806) */
		(void)tasks_blkiodelay_msec;
		if (in->flags & GYO_LI_DEPENDS) *issue = IS_DEPENDS;                                   /* :2826-2829, no return */
		else if (tasks_delay_msec * 10 > total_resp_msec) DECIDE(*state, IS_TASKS, 2853);       /* :2830 */
		else if (ser_errors) *issue = IS_SERVER_ERRORS;                                          /* :2855 */
		else *issue = IS_UNKNOWN;                                                                /* :2859 */
	}
	return 2866;
#undef HIGH_B
}

/* the caller's part: common/gy_socket_stat.cc:4241-4266 */
void gyo_listener_decide(const gyo_listener_scan *sc, const gyo_listener_issue_in *in, uint8_t *issue_bit_hist, uint8_t *high_resp_bit_hist,
			 gyo_listener_decision *out)
{
	uint8_t st = ST_OK, is = IS_NONE;
	memset(out, 0, sizeof(*out));
	out->decided_line = (uint16_t)gyo_listener_curr_state(sc, in, high_resp_bit_hist, &st, &is);
	if (!(in->flags & GYO_LI_YOUNG) || in->ser_errors) { /* diffstartusec > 100 s || ser_errors (:4244) */
		*issue_bit_hist = (uint8_t)(*issue_bit_hist << 1);
		if (st >= ST_BAD) *issue_bit_hist |= 1;
	} else { /* "Listener Just recently started. No status possible currently" (:4255-4262) */
		*issue_bit_hist = 0;
		is = IS_NONE;
		st = ST_OK;
		out->decided_line = 4262;
	}
	out->state = st;
	out->issue = is;
	out->issue_bit_hist = *issue_bit_hist;
	out->high_resp_bit_hist = *high_resp_bit_hist;
}

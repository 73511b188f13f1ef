/* TEST INFRASTRUCTURE -- CPU restatement of the filtered multi-host listener-state query, used only as the checker (tests/).
 *
 * MCONN_HANDLER::web_curr_listener_state (server/gy_mnodehandle.cc:4650-4900) walks the listeners, builds a SvcStateFields per listener and
 * keeps it when SvcStateFields::filter_match (server/gy_mfields.h:1497-1517) says so, until `nrecs >= maxrecs`.  Restated here:
 *   gyo_svc_col_value        SvcStateFields::get_num_field / get_bool_field (server/gy_mfields.h:1402-1440, :1484-1495): every numeric
 *                            column is an `int` made from the wire field (unsigned arithmetic first), `issue` an int16_t
 *   gyo_svc_term_match       CRITERION_ONE::match_num_criterian<Num> (common/gy_query_criteria.h:1243-1290)
 *   gyo_svc_filter_match     CRITERIA_ONE_GROUP::match_criteria_group (:1535-1605) inside CRITERIA_SET::match_criteria (:1806-1900) for
 *                            criteria that all belong to this subsystem: group g passes when (OPER_OR and a term matches) or (OPER_AND
 *                            and no term fails); the groups combine with l1_oper_
 *   gyo_svcstate_scan        the walk, serial, in SERVICE-SLOT order (the reference's order is that of its RCU hash table: arbitrary), the
 *                            optional sort on one column (then slot), maxrecs
 *   gyo_svcstate_aggr        AGGR_OPER_E sum / min / max / count per group (common/gy_json_field_maps.h:114-129)
 * The walk order and the sort are the engine's definitions (stated in include/gysketch.h); the per-record arithmetic is the reference's.
 * Parity: the column arithmetic and the comparator rules are read off the cited lines; the reference's classes need its JSON / RCU
 * machinery to compile (unbuildable here) => PARITY UNPINNED for the criteria evaluation, checked a second way by numpy in
 * tests/test_gpu_round4.py. */
#include <stdlib.h>
#include <string.h>

#include "gy_oracle.h"

static uint32_t q_u32(const uint8_t *p)
{
	uint32_t v;
	memcpy(&v, p, 4);
	return v;
}

int32_t gyo_svc_col_value(const uint8_t rec[88], int col)
{
	const uint32_t nq = q_u32(rec + 8);
	uint16_t nissue;
	memcpy(&nissue, rec + 76, 2);
	switch (col) {
	case 0: return (int)(nq / 5);                                  /* qps5s :1409 */
	case 1: return (int)nq;                                        /* nqry5s */
	case 2: return (int)(q_u32(rec + 12) / (nq ? nq : 1));         /* resp5s :1411 */
	case 3: return (int)q_u32(rec + 28);                           /* p95resp5s */
	case 4: return (int)q_u32(rec + 32);                           /* p95resp5m */
	case 5: return (int)q_u32(rec + 16);                           /* nconns */
	case 6: return (int)q_u32(rec + 20);                           /* nactive */
	case 7: return (int)q_u32(rec + 24);                           /* nprocs */
	case 8: return (int)q_u32(rec + 36);                           /* kbin15s */
	case 9: return (int)q_u32(rec + 40);                           /* kbout15s */
	case 10: return (int)q_u32(rec + 44);                          /* sererr */
	case 11: return (int)q_u32(rec + 48);                          /* clierr */
	case 12: return (int)q_u32(rec + 52);                          /* delayus */
	case 13: return (int)q_u32(rec + 56);                          /* cpudelus */
	case 14: return (int)q_u32(rec + 60);                          /* iodelus */
	case 15: return (int)(q_u32(rec + 52) - q_u32(rec + 56) - q_u32(rec + 60)); /* vmdelus :1426 */
	case 16: return (int)q_u32(rec + 64);                          /* usercpu */
	case 17: return (int)q_u32(rec + 68);                          /* syscpu */
	case 18: return (int)q_u32(rec + 72);                          /* rssmb */
	case 19: return (int)nissue;                                   /* nissue :1432 */
	case 20: return (int)rec[79];                                  /* state (numeric OBJ_STATE_E; filters name it through statefromjson) */
	case 21: return (int)(int16_t)rec[80];                         /* issue :1433 */
	case 22: return rec[78] != 0;                                  /* ishttp :1488 */
	default: return 0;
	}
}

static int32_t q_conv(int col, int64_t v) { return col == 21 ? (int32_t)(int16_t)v : (int32_t)v; } /* the criterion in the field's own type */

int gyo_svc_term_match(const gyo_svc_term *t, int32_t v, const int64_t *set_values)
{
	const int32_t crit = q_conv(t->col, t->value);
	switch (t->comp) {
	case 0: return v == crit;
	case 1: return v != crit;
	case 2: return v < crit;
	case 3: return v <= crit;
	case 4: return v > crit;
	case 5: return v >= crit;
	case 6: return (v & 3) == 3;
	case 7: return (v & 7) == 7;
	case 12:
	case 13: {
		const int bret = t->comp == 12;
		for (uint32_t i = 0; i < t->nvalues; i++)
			if (q_conv(t->col, set_values[t->set_first + i]) == v) return bret;
		return !bret;
	}
	default: return 0;
	}
}

int gyo_svc_filter_match(const uint8_t rec[88], const gyo_svc_term *terms, uint32_t nterms, const int64_t *set_values, const uint8_t group_oper[8],
			 int top_oper)
{
	int npass = 0, nfail = 0, ngroups = 0;
	if (nterms == 0) return 1; /* CRIT_SKIP: listed */
	for (int g = 0; g < 8; g++) {
		int neval = 0, iseval = 0, result = -1; /* -1: undecided */
		for (uint32_t i = 0; i < nterms && result < 0; i++) {
			if (terms[i].group != g) continue;
			neval++;
			if (gyo_svc_term_match(&terms[i], gyo_svc_col_value(rec, terms[i].col), set_values)) {
				if (group_oper[g]) result = 1; /* OPER_OR: CRIT_PASS at the first match :1585-1588 */
				iseval = 1;
			} else if (!group_oper[g])
				result = 0; /* OPER_AND: CRIT_FAIL at the first miss :1591-1593 */
		}
		if (!neval) continue;
		if (result < 0) result = (iseval && !group_oper[g]) ? 1 : 0; /* :1596-1604 */
		ngroups++;
		if (result) npass++;
		else nfail++;
	}
	(void)ngroups;
	return top_oper ? npass > 0 : nfail == 0;
}

/* is the kept record of `slot` current?  (state of this or the last window, not deleted, the slot's own listener, a host of the query) */
static int q_current(const uint8_t *svc_state, uint32_t slot, uint32_t epoch, const uint32_t *svc_host, const uint64_t *svc_gid, const uint8_t *host_in)
{
	const uint8_t *r = svc_state + (size_t)slot * 96;
	uint64_t gid;
	const uint32_t ep = q_u32(r + 88), host = q_u32(r + 92);
	memcpy(&gid, r, 8);
	if (ep == 0 || ep + 1 < epoch) return 0;
	if (host != svc_host[slot] || gid != svc_gid[slot]) return 0;
	if (host_in && !host_in[host]) return 0;
	return 1;
}

typedef struct {
	int64_t v;
	uint32_t slot;
} q_cand;
static int q_sort_desc;
static int q_cmp(const void *a, const void *b)
{
	const q_cand *x = (const q_cand *)a, *y = (const q_cand *)b;
	if (x->v != y->v) return q_sort_desc ? (x->v < y->v ? 1 : -1) : (x->v < y->v ? -1 : 1);
	return x->slot < y->slot ? -1 : (x->slot > y->slot ? 1 : 0);
}

/* svc_state: [nsvc * 96] kept records + {window, host}; host_in: [nhosts] 0 / 1 or NULL = every host.  out_slots: [maxrecs].
 * Returns the number written; *nmatched = records that matched. */
uint32_t gyo_svcstate_scan(const uint8_t *svc_state, uint32_t nsvc, uint32_t epoch, const uint32_t *svc_host, const uint64_t *svc_gid, const uint8_t *host_in,
			   const gyo_svc_term *terms, uint32_t nterms, const int64_t *set_values, const uint8_t group_oper[8], int top_oper, int sort_col,
			   int sort_desc, uint32_t maxrecs, uint32_t *out_slots, uint64_t *nmatched, const uint32_t *slot_list, uint32_t nlist)
{
	/* slot_list != NULL: the query names its listeners (svcid = / in: the direct-lookup path of web_curr_listener_state :4754-4860) --
	 * only these slots are visited */
	const uint32_t nitems = slot_list ? nlist : nsvc;
	q_cand *c = (q_cand *)malloc(sizeof(q_cand) * (nitems ? nitems : 1));
	uint32_t n = 0;
	for (uint32_t it = 0; it < nitems; it++) {
		const uint32_t s = slot_list ? slot_list[it] : it;
		const uint8_t *r = svc_state + (size_t)s * 96;
		if (!q_current(svc_state, s, epoch, svc_host, svc_gid, host_in)) continue;
		if (!gyo_svc_filter_match(r, terms, nterms, set_values, group_oper, top_oper)) continue;
		c[n].v = sort_col >= 0 ? gyo_svc_col_value(r, sort_col) : 0;
		c[n].slot = s;
		n++;
	}
	*nmatched = n;
	q_sort_desc = sort_desc;
	qsort(c, n, sizeof(q_cand), q_cmp);
	if (n > maxrecs) n = maxrecs;
	for (uint32_t i = 0; i < n; i++) out_slots[i] = c[i].slot;
	free(c);
	return n;
}

/* group_by 0: one group; 1: host slot; 2: host_cluster[host].  acc: [ngroups * ncols * 3] {sum, min, max} (caller-initialised: 0, INT64_MAX,
 * INT64_MIN), count: [ngroups] */
void gyo_svcstate_aggr(const uint8_t *svc_state, uint32_t nsvc, uint32_t epoch, const uint32_t *svc_host, const uint64_t *svc_gid, const uint8_t *host_in,
		       const gyo_svc_term *terms, uint32_t nterms, const int64_t *set_values, const uint8_t group_oper[8], int top_oper, int group_by,
		       const uint32_t *host_cluster, const uint8_t *cols, uint32_t ncols, int64_t *acc, uint64_t *count, const uint32_t *slot_list, uint32_t nlist)
{
	const uint32_t nitems = slot_list ? nlist : nsvc;
	for (uint32_t it = 0; it < nitems; it++) {
		const uint32_t s = slot_list ? slot_list[it] : it;
		const uint8_t *r = svc_state + (size_t)s * 96;
		if (!q_current(svc_state, s, epoch, svc_host, svc_gid, host_in)) continue;
		if (!gyo_svc_filter_match(r, terms, nterms, set_values, group_oper, top_oper)) continue;
		const uint32_t host = q_u32(r + 92);
		const uint32_t g = group_by == 0 ? 0 : group_by == 1 ? host : host_cluster[host];
		count[g]++;
		for (uint32_t a = 0; a < ncols; a++) {
			const int64_t v = gyo_svc_col_value(r, cols[a]);
			int64_t *p = acc + ((size_t)g * ncols + a) * 3;
			p[0] += v;
			if (v < p[1]) p[1] = v;
			if (v > p[2]) p[2] = v;
		}
	}
}

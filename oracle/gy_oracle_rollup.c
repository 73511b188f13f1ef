/* ORACLE (test infrastructure, CPU): roll-up digests -- the response-time digest of a GROUP of services (a host, a cluster, every
 * host of a madhava, every madhava).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
 *
 * What it stands for in the reference: the aggregated percentile of a group is computed by Postgres from the rows of its members,
 * `public.tdigest_percentile(col, 100, p)` over the selected listeners (common/gy_query_common.cc:1818-1855), and the cluster-level
 * fan-in is SHCONN_HANDLER::aggregate_cluster_state (server/gy_shconnhdlr.cc:4583-4720).  The tdigest extension is not in
 * /root/reference => PARITY UNPINNED; the definition below is the builder's, frozen here so that "bit-exact" is defined:
 *
 *   rollup(group) = left fold over the members in their given order of
 *        d := merge(d, member)        member = a service: first its clusters (weighted points at their means, gy_oracle.c
 *                                     gyo_td_merge_digest), then its buffered values (unit points, gyo_td_merge_values);
 *                                     member = another roll-up digest: its clusters (weighted points).
 *   merge = the same exact-integer k-bucket merge as a service's own digest (gy_oracle.c td_merge_items): items ordered by mean
 *   (exact rational compare), ties: old clusters first; an item whose weighted mid-point is mid2 / 2 of N goes to cluster
 *   gyo_td_cluster(mid2, 2N).  A group's weight exceeds 32 bits (10^4 hosts x 2^29 events per window), so the counters are 64-bit.
 *
 *   (gyo_td64_merge_* below: the 64-bit form of a service's own merge.  Until round 6 the roll-up WAS that fold; it is kept as the
 *   wide-counter merge the tests compare against the 32-bit one.)
 *
 *   ROUND 6 -- THE ROLL-UP IS THE UNION BY VALUE BIN (gyo_tdbins_*), not a fold: 10^7 sequential member steps were 0.3 s per query.
 *     bins   2048 value bins over the integer-millisecond domain 0 <= v < 2^26 of a staged word (gyo_td_value_bin): one bin per value
 *            below 1024, then 64 cells per octave.
 *     add    every non-empty cluster (sum, cnt) of a member goes, whole, to the bin of ceil(sum / cnt); every buffered value v of a
 *            service to the bin of v as (v, 1).  A bin holds the exact 64-bit totals (sum, cnt) of what was added: additions commute, so a
 *            group's bins do not depend on the order (or the grouping) in which its members are visited.
 *     finish the bins, in order, are laid on the rank axis: bin b with weight w_b and W_b = weight of the bins below it occupies the unit
 *            mid-points 2 (W_b + r) + 1, r = 0 .. w_b - 1; point r belongs to cluster gyo_td_cluster(2 (W_b + r) + 1, 2 N) -- the same
 *            cluster rule as every merge -- and the points r0 <= r < r1 of a bin that fall into one cluster bring it
 *            floor(sum_b r1 / w_b) - floor(sum_b r0 / w_b) of the bin's sum (exact 128-bit product; the shares of a bin add up to sum_b).
 *            (Integer shares: the pieces of one bin have means that differ by less than one unit per point of the piece and are not
 *            ordered among themselves; all lie inside the bin.)  A bin's mass is thereby spread evenly over its ranks: a heavy value (30 % of all responses take 5 ms) is split over the
 *            clusters its ranks span instead of making one oversized cluster.
 *     vmin / vmax: over the members that contribute clusters (their own extremes) and over the buffered values.
 *   A roll-up of roll-ups (cluster = its hosts' slabs, global = all host slabs of the rank, all ranks = the ranks' global slabs) is the same
 *   operation on the members' clusters.  Values are at most one bin away from where they were (below 1024 ms: exact to the millisecond
 *   the digests resolve anyway; above: within the 1.6 % of a cell), ranks are exact up to the members' own cluster widths.
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "gy_oracle.h"

void gyo_td64_init(gyo_td64 *d)
{
	memset(d, 0, sizeof(*d));
	d->vmin = INT_MAX;
	d->vmax = INT_MIN;
}

uint64_t gyo_td64_total(const gyo_td64 *d)
{
	uint64_t n = 0;
	for (int i = 0; i < GYO_TD_NB; i++) n += d->cnt[i];
	return n;
}

typedef struct {
	int64_t sum;
	uint64_t cnt;
} td64_item;

/* items sorted by mean (non-decreasing).  Same two passes as td_merge_items (gy_oracle.c:565-625), 64-bit counts. */
static void td64_merge_items(gyo_td64 *d, const td64_item *items, size_t m)
{
	gyo_td64 out;
	uint64_t nold = gyo_td64_total(d), nnew = 0, twoN, wold_before = 0;

	for (size_t i = 0; i < m; i++) nnew += items[i].cnt;
	if (nnew == 0) return;
	twoN = 2 * (nold + nnew);
	gyo_td64_init(&out);
	out.vmin = d->vmin;
	out.vmax = d->vmax;
	{ /* old clusters: W = (old weight before j) + (new weight with mean strictly < mean_j) */
		size_t p = 0;
		uint64_t new_lt = 0;
		for (int j = 0; j < GYO_TD_NB; j++) {
			if (!d->cnt[j]) continue;
			while (p < m && (__int128)items[p].sum * (__int128)d->cnt[j] < (__int128)d->sum[j] * (__int128)items[p].cnt) {
				new_lt += items[p].cnt;
				p++;
			}
			{
				const uint64_t mid2 = 2 * (wold_before + new_lt) + d->cnt[j];
				const uint32_t c = gyo_td_cluster(mid2, twoN);
				out.sum[c] += d->sum[j];
				out.cnt[c] += d->cnt[j];
			}
			wold_before += d->cnt[j];
		}
	}
	{ /* new items: W = (new weight before i) + (old weight with mean <= mean_i) */
		int j = 0;
		uint64_t old_le = 0, new_before = 0;
		for (size_t i = 0; i < m; i++) {
			while (j < GYO_TD_NB) {
				if (!d->cnt[j]) {
					j++;
					continue;
				}
				if ((__int128)d->sum[j] * (__int128)items[i].cnt <= (__int128)items[i].sum * (__int128)d->cnt[j]) {
					old_le += d->cnt[j];
					j++;
				} else
					break;
			}
			{
				const uint64_t mid2 = 2 * (new_before + old_le) + items[i].cnt;
				const uint32_t c = gyo_td_cluster(mid2, twoN);
				out.sum[c] += items[i].sum;
				out.cnt[c] += items[i].cnt;
			}
			new_before += items[i].cnt;
		}
	}
	*d = out;
}

static int cmp_i32(const void *a, const void *b)
{
	const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
	return (x > y) - (x < y);
}

void gyo_td64_merge_values(gyo_td64 *d, const int32_t *vals, size_t m)
{
	int32_t *s;
	td64_item *it;

	if (!m) return;
	s = (int32_t *)malloc(m * sizeof(int32_t));
	it = (td64_item *)malloc(m * sizeof(td64_item));
	memcpy(s, vals, m * sizeof(int32_t));
	qsort(s, m, sizeof(int32_t), cmp_i32);
	for (size_t i = 0; i < m; i++) {
		it[i].sum = s[i];
		it[i].cnt = 1;
	}
	td64_merge_items(d, it, m);
	if (s[0] < d->vmin) d->vmin = s[0];
	if (s[m - 1] > d->vmax) d->vmax = s[m - 1];
	free(s);
	free(it);
}

/* a service: clusters first, then the buffered values */
void gyo_td64_merge_service(gyo_td64 *d, const gyo_td_buffered *b)
{
	td64_item it[GYO_TD_NB];
	size_t m = 0;

	for (int j = 0; j < GYO_TD_NB; j++) {
		if (b->d.cnt[j]) {
			it[m].sum = b->d.sum[j];
			it[m].cnt = b->d.cnt[j];
			m++;
		}
	}
	if (m) {
		td64_merge_items(d, it, m);
		if (b->d.vmin < d->vmin) d->vmin = b->d.vmin;
		if (b->d.vmax > d->vmax) d->vmax = b->d.vmax;
	}
	gyo_td64_merge_values(d, gyo_tdb_values(b), b->npend);
}

void gyo_td64_merge_td64(gyo_td64 *d, const gyo_td64 *o)
{
	td64_item it[GYO_TD_NB];
	size_t m = 0;

	for (int j = 0; j < GYO_TD_NB; j++) {
		if (o->cnt[j]) {
			it[m].sum = o->sum[j];
			it[m].cnt = o->cnt[j];
			m++;
		}
	}
	if (!m) return;
	td64_merge_items(d, it, m);
	if (o->vmin < d->vmin) d->vmin = o->vmin;
	if (o->vmax > d->vmax) d->vmax = o->vmax;
}

/* ---------------------------------------------------------------- the roll-up: union by value bin (definition in the header comment) */
uint32_t gyo_td_value_bin(uint32_t v)
{
	uint32_t msb = 31;

	if (v < 1024u) return v;
	if (v >= (1u << 26)) v = (1u << 26) - 1u; /* (outside the engine's value domain: a staged word carries 26 value bits) */
	while (!(v >> msb)) msb--;
	return 1024u + (msb - 10u) * 64u + ((v >> (msb - 6u)) & 63u);
}

void gyo_tdbins_init(gyo_td_bins *b)
{
	memset(b, 0, sizeof(*b));
	b->vmin = INT_MAX;
	b->vmax = INT_MIN;
}

static void tdbins_add_cluster(gyo_td_bins *b, int64_t sum, uint64_t cnt)
{
	uint64_t thr;
	uint32_t k;

	if (!cnt) return;
	thr = sum <= 0 ? 0 : ((uint64_t)sum + cnt - 1) / cnt; /* ceil of the mean */
	k = gyo_td_value_bin(thr > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)thr);
	b->sum[k] += (uint64_t)sum;
	b->cnt[k] += cnt;
}

void gyo_tdbins_add_values(gyo_td_bins *b, const int32_t *vals, size_t m)
{
	for (size_t i = 0; i < m; i++) {
		const uint32_t k = gyo_td_value_bin(vals[i] < 0 ? 0u : (uint32_t)vals[i]);
		b->sum[k] += (uint64_t)(int64_t)vals[i];
		b->cnt[k] += 1;
		if (vals[i] < b->vmin) b->vmin = vals[i];
		if (vals[i] > b->vmax) b->vmax = vals[i];
	}
}

void gyo_tdbins_add_service(gyo_td_bins *b, const gyo_td_buffered *s)
{
	int any = 0;

	for (int j = 0; j < GYO_TD_NB; j++) {
		if (s->d.cnt[j]) {
			tdbins_add_cluster(b, s->d.sum[j], s->d.cnt[j]);
			any = 1;
		}
	}
	if (any) {
		if (s->d.vmin < b->vmin) b->vmin = s->d.vmin;
		if (s->d.vmax > b->vmax) b->vmax = s->d.vmax;
	}
	gyo_tdbins_add_values(b, gyo_tdb_values(s), s->npend);
}

void gyo_tdbins_add_td64(gyo_td_bins *b, const gyo_td64 *o)
{
	int any = 0;

	for (int j = 0; j < GYO_TD_NB; j++) {
		if (o->cnt[j]) {
			tdbins_add_cluster(b, o->sum[j], o->cnt[j]);
			any = 1;
		}
	}
	if (any) {
		if (o->vmin < b->vmin) b->vmin = o->vmin;
		if (o->vmax > b->vmax) b->vmax = o->vmax;
	}
}

void gyo_tdbins_finish(const gyo_td_bins *b, gyo_td64 *out)
{
	uint64_t N = 0, W = 0;

	gyo_td64_init(out);
	out->vmin = b->vmin;
	out->vmax = b->vmax;
	for (int k = 0; k < GYO_TD_BINS; k++) N += b->cnt[k];
	if (!N) return;
	for (int k = 0; k < GYO_TD_BINS; k++) {
		const uint64_t w = b->cnt[k], s = b->sum[k];
		uint64_t r = 0, given = 0;

		while (r < w) {
			const uint32_t a = gyo_td_cluster(2 * (W + r) + 1, 2 * N);
			uint64_t lo = r + 1, hi = w, upto; /* r1 = first point after r that is not in cluster a (w when there is none) */
			while (lo < hi) {
				const uint64_t mid = lo + (hi - lo) / 2;
				if (gyo_td_cluster(2 * (W + mid) + 1, 2 * N) != a) hi = mid; else lo = mid + 1;
			}
			upto = (uint64_t)(((unsigned __int128)s * lo) / w); /* floor(s r1 / w); == s at r1 == w */
			out->sum[a] += (int64_t)(upto - given);
			out->cnt[a] += lo - r;
			given = upto;
			r = lo;
		}
		W += w;
	}
}

/* the same interpolation as gyo_td_quantile (gy_oracle.c:669-715) on the wide counters; only + - * / on doubles */
double gyo_td64_quantile(const gyo_td64 *d, double q)
{
	const uint64_t N = gyo_td64_total(d);
	double t, wbefore = 0.0, prev_c = 0.0, prev_mean = 0.0, r;
	int have_prev = 0;

	if (!N) return 0.0;
	if (q < 0.0) q = 0.0;
	if (q > 1.0) q = 1.0;
	t = q * (double)N;
	r = (double)d->vmax;
	for (int k = 0; k < GYO_TD_NB; k++) {
		double c, mean;
		if (!d->cnt[k]) continue;
		mean = (double)d->sum[k] / (double)d->cnt[k];
		c = wbefore + (double)d->cnt[k] * 0.5;
		if (t < c) {
			if (!have_prev) {
				const double lo = (double)d->vmin;
				r = c <= 0.0 ? mean : lo + (mean - lo) * (t / c);
			} else {
				r = prev_mean + (mean - prev_mean) * ((t - prev_c) / (c - prev_c));
			}
			goto done;
		}
		wbefore += (double)d->cnt[k];
		prev_c = c;
		prev_mean = mean;
		have_prev = 1;
	}
	{
		const double hi = (double)d->vmax, span = (double)N - prev_c;
		r = span <= 0.0 ? hi : prev_mean + (hi - prev_mean) * ((t - prev_c) / span);
	}
done:
	return floor(r + 0.5); /* integer-millisecond value domain: round half up, as gyo_td_quantile */
}

/* ================================================================ ACTIVE_CONN_STATS roll-up
 * comm::ACTIVE_CONN_STATS (common/gy_comm_proto.h:2766-2783, 104 bytes): one row per (listener, client task group) of a partha's
 * 15-s report.  MCONN_HANDLER::insert_active_conns (server/gy_mconnhdlr.cc:7776-7960) splits the rows by is_remote_listen_: rows of
 * LOCAL listeners (false) go to activeconntbl (:7842-7876), rows whose listener lives on another madhava (true) to remoteconntbl
 * (:7888-7925); nothing is kept in memory.  The engine's replacement of those tables' per-(listener, client task) rows (and of
 * connlistenmap_ / connclientmap_, server/gy_msocket.h:240-290): a Count-Min pair keyed by the 4 words
 * (listener_glob_id lo, hi, cli_aggr_task_id lo, hi) -- active_conns_ into the u32 table, bytes_sent_ + bytes_received_ into the
 * u64 table -- for the local-listener rows; out[0] = local-listener rows, out[1] = remote-listener rows.  PARITY UNPINNED (builder-defined). */
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
/* rpair32 / rpair64 (may be NULL): the same roll-up of the remote-listener rows (-> remoteconntbl :7888-7925) into tables of their own */
void gyo_active_conn_sketch_batch2(const uint8_t *batch, int nrec, uint32_t *pair32, uint64_t *pair64, uint32_t *rpair32, uint64_t *rpair64, uint64_t out[2])
{
	out[0] = out[1] = 0;
	for (int i = 0; i < nrec; i++) {
		const uint8_t *r = batch + (size_t)i * 104;
		const uint64_t gid = rd64(r), task = rd64(r + 8), sent = rd64(r + 72), rcvd = rd64(r + 80);
		const uint32_t w[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)task, (uint32_t)(task >> 32)};
		uint16_t act;
		memcpy(&act, r + 100, 2);
		if (r[102] & 2) { /* is_remote_listen_ */
			out[1]++;
			if (rpair32) gyo_cms_add(rpair32, w, 4, act);
			if (rpair64) gyo_cms64_add(rpair64, w, 4, sent + rcvd);
			continue;
		}
		out[0]++;
		gyo_cms_add(pair32, w, 4, act);
		gyo_cms64_add(pair64, w, 4, sent + rcvd);
	}
}

void gyo_active_conn_sketch_batch(const uint8_t *batch, int nrec, uint32_t *pair32, uint64_t *pair64, uint64_t out[2])
{
	gyo_active_conn_sketch_batch2(batch, nrec, pair32, pair64, NULL, NULL, out);
}

/* the same kind of pair fed by TCP_CONN_NOTIFY records (gys_config.conn_pair_cms): key (ser_glob_id_ @192, cli_task_aggr_id_
 * @144).  It follows the reference's close roll-ups (server/gy_mconnhdlr.cc:9182, :9226-9245, :9290-9312): only a record with tusec_close_
 * (@136) and bytes_sent_ + bytes_rcvd_ > 0 (@208, @216) counts -- one connection into the u32 table, its bytes into the u64 table -- into the
 * LISTENER-side tables (connlistenmap_) when ser_glob_id_ != 0 and is_tcp_accept_event_ (@275), into the CLIENT-side tables (connclientmap_)
 * when it is connect-only (@274 set, @275 clear) with cli_task_aggr_id_ != 0.  (The reference's `connpeer` gate depends on its flow-join
 * table and is not restated: out of scope with the join.)  Variable stride, common/gy_comm_proto.h:1721-1724. */
int gyo_tcp_conn_pair_batch(const uint8_t *batch, int nrec, const uint8_t *pend, uint32_t *pair32, uint64_t *pair64, uint32_t *cpair32, uint64_t *cpair64)
{
	const uint8_t *p = batch;
	int i;
	for (i = 0; i < nrec && p < pend; ++i, p += gyo_tcp_conn_elem_size(p)) {
		const uint64_t gid = rd64(p + 192), task = rd64(p + 144), bytes = rd64(p + 208) + rd64(p + 216);
		const uint32_t w[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)task, (uint32_t)(task >> 32)};
		const int connect = p[274] != 0, accept = p[275] != 0;
		if (rd64(p + 136) == 0 || bytes == 0) continue;
		if (gid && accept) {
			gyo_cms_add(pair32, w, 4, 1);
			gyo_cms64_add(pair64, w, 4, bytes);
		} else if (!accept && connect && task) {
			gyo_cms_add(cpair32, w, 4, 1);
			gyo_cms64_add(cpair64, w, 4, bytes);
		}
	}
	return i;
}

/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's multi-level ("Last 5 seconds / 5 minutes / 5 days / since
 * start") response histogram: TIME_HISTOGRAM<RESP_TIME_HASH, Level_5s_5min_5days_all> (common/gy_statistics.h:1082-1551, :2067).
 *
 * PARITY UNPINNED for the ring arithmetic; the percentile rule IS pinned (see the end of this comment).  The arithmetic of that class
 * lives in a third-party dependency that is absent from /root/reference:
 * facebook/folly (version unpinned -- Makefile.common:61 only names an install directory), folly/stats/BucketedTimeSeries{.h,-inl.h}
 * and folly/stats/MultiLevelTimeSeries{.h,-inl.h}; the reference's only test at this boundary (test/test_timeseries_hist.cc)
 * prints and asserts nothing.  What follows restates folly's published algorithm directly (a ring of per-bucket {sum, count}
 * that is cleared as time advances) and anchors on the reference's own call sites:
 *   constructor  gy_statistics.h:1107-1108  folly::MultiLevelTimeSeries<int64_t>(ntimeseries_buckets = 10, levels, dist_seconds)
 *   add          gy_statistics.h:1213-1247  add_histogram_data(): per histogram bucket addValueAggregated(tnow, sum, count)
 *   flush        gy_statistics.h:1292-1320  slabhist.update(tnow)
 *   read         gy_statistics.h:1166-1200  get_level_data(): levelobj.sum() / levelobj.count()
 *                gy_statistics.h:1333-1367  get_stats(): get_bucket_max_threshold(slabhist.getPercentileBucketIdx(pct, level))
 *   percentile   thirdparty/SlabHistogramBucket.h:165-240 (in tree) getPercentileBucketIdx
 * Pinned part: gyo_slab_percentile_idx and the bucket numbering are checked against the reference's OWN in-tree container
 * (thirdparty/SlabHistogramBucket.h compiled into oracle/_ref: tests/test_oracle_vs_ref.py::test_slab_percentile_rule_vs_reference_container,
 * golden vectors "slab" in tests/golden/ref_vectors.json).
 * The engine does NOT keep such rings (it keeps cumulative snapshots at bucket boundaries and answers a level as a difference,
 * DESIGN.md "multi-level windows"), so agreement between the two is a real check of both.
 */
#include <string.h>

#include "gy_oracle.h"

/* ---------------------------------------------------------------- folly::BucketedTimeSeries<int64_t, LegacyStatsClock<seconds>> */
void gyo_bts_init(gyo_bts *s, uint32_t nbuckets, int64_t duration)
{
	memset(s, 0, sizeof(*s));
	s->duration = duration;
	s->first_time = 1; /* firstTime_ > latestTime_  <=>  empty() */
	s->latest_time = 0;
	if (duration != 0) {
		/* "There is no point in having more buckets than our timestamp granularity" */
		if ((int64_t)nbuckets > duration) nbuckets = (uint32_t)duration;
		if (nbuckets > GYO_BTS_MAXB) nbuckets = GYO_BTS_MAXB;
		s->nbuckets = nbuckets;
	}
}

static int bts_empty(const gyo_bts *s) { return s->first_time > s->latest_time; }

static uint32_t bts_bucket_idx(const gyo_bts *s, int64_t t) { return (uint32_t)((t % s->duration) * (int64_t)s->nbuckets / s->duration); }

static void bts_bucket_info(const gyo_bts *s, int64_t t, uint32_t *idx, int64_t *start, int64_t *next)
{
	const int64_t n = (int64_t)s->nbuckets;
	const int64_t time_mod = t % s->duration, full = t / s->duration;
	const int64_t scaled = time_mod * n;
	const int64_t scaled_off = scaled % s->duration;
	const int64_t scaled_start = scaled - scaled_off, scaled_next = scaled_start + s->duration;
	*idx = (uint32_t)(scaled / s->duration);
	*start = full * s->duration + (scaled_start + n - 1) / n;
	*next = full * s->duration + (scaled_next + n - 1) / n;
}

static uint32_t bts_update_buckets(gyo_bts *s, int64_t now)
{
	uint32_t cur;
	int64_t cur_start, next_start;
	bts_bucket_info(s, s->latest_time, &cur, &cur_start, &next_start);
	s->latest_time = now;
	if (now < next_start) return cur;
	if (now >= cur_start + s->duration) { /* wrapped: everything is older than the window */
		memset(s->bsum, 0, sizeof(s->bsum));
		memset(s->bcnt, 0, sizeof(s->bcnt));
		s->tot_sum = 0;
		s->tot_cnt = 0;
		return bts_bucket_idx(s, now);
	}
	const uint32_t nb = bts_bucket_idx(s, now);
	uint32_t i = cur;
	while (i != nb) { /* (cur, nb] are the oldest buckets of the ring */
		if (++i >= s->nbuckets) i = 0;
		s->tot_sum -= s->bsum[i];
		s->tot_cnt -= s->bcnt[i];
		s->bsum[i] = 0;
		s->bcnt[i] = 0;
	}
	return nb;
}

int gyo_bts_add(gyo_bts *s, int64_t now, int64_t sum, uint64_t nsamples)
{
	if (s->duration == 0) { /* all-time level: only the total */
		if (bts_empty(s)) {
			s->first_time = now;
			s->latest_time = now;
		} else if (now > s->latest_time) {
			s->latest_time = now;
		} else if (now < s->first_time) {
			s->first_time = now;
		}
		s->tot_sum += sum;
		s->tot_cnt += nsamples;
		return 1;
	}
	uint32_t b;
	if (bts_empty(s)) {
		s->first_time = now;
		s->latest_time = now;
		b = bts_bucket_idx(s, now);
	} else if (now > s->latest_time) {
		b = bts_update_buckets(s, now);
	} else if (now == s->latest_time) {
		b = bts_bucket_idx(s, now);
	} else {
		/* a point in the past: accepted only while it still falls inside the window (getEarliestTimeNonEmpty) */
		uint32_t cur;
		int64_t cur_start, next_start;
		bts_bucket_info(s, s->latest_time, &cur, &cur_start, &next_start);
		int64_t earliest = next_start - s->duration; /* oldest point the ring can still hold */
		if (earliest < s->first_time) earliest = s->first_time;
		if (now < earliest) return 0;
		b = bts_bucket_idx(s, now);
	}
	s->tot_sum += sum;
	s->tot_cnt += nsamples;
	s->bsum[b] += sum;
	s->bcnt[b] += nsamples;
	return 1;
}

void gyo_bts_update(gyo_bts *s, int64_t now)
{
	if (bts_empty(s)) s->first_time = now;
	if (s->duration == 0) {
		if (s->latest_time < now) s->latest_time = now;
		return;
	}
	if (now < s->latest_time) now = s->latest_time; /* time does not go backwards */
	bts_update_buckets(s, now);
}

/* ---------------------------------------------------------------- TIME_HISTOGRAM over folly::MultiLevelTimeSeries per histogram bucket */
static const int64_t g_level_secs[GYO_MLH_LEVELS] = {5, 300, 5 * 24 * 3600, 0}; /* Level_5s_5min_5days_all gy_statistics.h:1545-1551 */

int64_t gyo_mlh_level_seconds(int level) { return g_level_secs[level]; }

void gyo_mlh_init(gyo_mlhist *h, int kind, uint32_t ntimeseries_buckets)
{
	memset(h, 0, sizeof(*h));
	h->kind = kind;
	h->nb = gyo_hist_nbuckets(kind);
	for (int b = 0; b < h->nb; b++)
		for (int l = 0; l < GYO_MLH_LEVELS; l++) gyo_bts_init(&h->s[b][l], ntimeseries_buckets, g_level_secs[l]);
}

/* MultiLevelTimeSeries::flush(): the cached (time, sum, count) of one histogram bucket goes into every level */
static void mlh_flush_cache(gyo_mlhist *h, int b)
{
	if (h->cached[b].count > 0) {
		for (int l = 0; l < GYO_MLH_LEVELS; l++) gyo_bts_add(&h->s[b][l], h->cached_time[b], h->cached[b].sum, h->cached[b].count);
		h->cached[b].count = 0;
		h->cached[b].sum = 0;
	}
}

void gyo_mlh_flush(gyo_mlhist *h, int64_t tnow) /* TIME_HISTOGRAM::flush -> slabhist.update(tnow): MultiLevelTimeSeries::update per bucket */
{
	for (int b = 0; b < h->nb; b++) {
		mlh_flush_cache(h, b);
		for (int l = 0; l < GYO_MLH_LEVELS; l++) gyo_bts_update(&h->s[b][l], tnow);
	}
}

void gyo_mlh_add_hist(gyo_mlhist *h, int64_t tnow, const gyo_hist_serial *stats, int flush) /* add_histogram_data :1213-1247 */
{
	uint64_t count = 0;
	for (int b = 0; b < h->nb; b++) {
		/* MultiLevelTimeSeries::addValueAggregated: a new timestamp flushes the cache first */
		if (h->cached_time[b] != tnow) {
			mlh_flush_cache(h, b);
			h->cached_time[b] = tnow;
		}
		h->cached[b].sum += stats[b].sum;
		h->cached[b].count += stats[b].count;
		count += stats[b].count;
	}
	if (count == 0) return; /* :1231-1233 (before the optional flush) */
	if (flush) gyo_mlh_flush(h, tnow);
}

void gyo_mlh_level(const gyo_mlhist *h, int level, gyo_hist_serial *out) /* get_level_data :1166-1200 */
{
	for (int b = 0; b < GYO_MAX_BUCKETS; b++) {
		out[b].sum = 0;
		out[b].count = 0;
	}
	for (int b = 0; b < h->nb; b++) {
		out[b].sum = h->s[b][level].tot_sum;
		out[b].count = h->s[b][level].tot_cnt;
	}
}

/* SlabHistogramBuckets::getPercentileBucketIdx (thirdparty/SlabHistogramBucket.h:165-240); pct in [0, 1] */
size_t gyo_slab_percentile_idx(const uint64_t *counts, size_t nb, double pct)
{
	uint64_t total = 0, cur = 0;
	for (size_t n = 0; n < nb; n++) total += counts[n];
	if (total == 0) return 1; /* "first bucket in the histogram range" */
	size_t idx;
	for (idx = 0; idx < nb; idx++) {
		if (counts[idx] == 0) continue;
		cur += counts[idx];
		const double cur_pct = (double)cur / total;
		if (pct <= cur_pct) break;
	}
	return idx;
}

/* TIME_HISTOGRAM::get_stats (:1333-1367) on one level; pcts are the float percentiles 0..100 of TIME_HIST_VAL */
void gyo_mlh_get_stats(const gyo_mlhist *h, int level, const float *pcts, size_t npct, int64_t *values, int64_t *tcount, int64_t *tsum, double *mean)
{
	gyo_hist_serial lv[GYO_MAX_BUCKETS];
	uint64_t counts[GYO_MAX_BUCKETS];
	int64_t tc = 0, ts = 0;
	gyo_mlh_level(h, level, lv);
	for (int b = 0; b < h->nb; b++) {
		counts[b] = lv[b].count;
		tc += (int64_t)lv[b].count;
		ts += lv[b].sum;
	}
	for (size_t i = 0; i < npct; i++) {
		/* TimeseriesSlabHistogram::getPercentileBucketIdx(pct, level): pct / 100.0 (thirdparty/TimeseriesSlabHistogram-defs.h:101-105) */
		int64_t v = gyo_bucket_max_threshold(h->kind, gyo_slab_percentile_idx(counts, (size_t)h->nb, (double)pcts[i] / 100.0));
		values[i] = v < 0 ? 0 : v;
	}
	if (tcount) *tcount = tc;
	if (tsum) *tsum = ts;
	if (mean) *mean = (double)ts / (tc != 0 ? tc : 1);
}

/* ---------------------------------------------------------------- TIME_HISTOGRAM::get_stats_for_period (gy_statistics.h:1378-1406)
 * = TimeseriesSlabHistogram::count / sum / getPercentileBucketIdx over [start, end) (thirdparty/TimeseriesSlabHistogram.h:131-156,
 * :239-245, CountFromInterval :296-309), each of which asks every histogram bucket's folly::MultiLevelTimeSeries for
 * count(start, end) / sum(start, end).  folly (absent, PARITY UNPINNED) answers that from ONE level -- the first whose span reaches
 * back to `start`, MultiLevelTimeSeries::getLevel(start): latestTime - duration <= start, else the all-time level -- by walking that
 * level's ring from the oldest bucket (BucketedTimeSeries::forEachBucket) and scaling a bucket that only partly overlaps the
 * interval by the overlapped fraction in FLOAT arithmetic (BucketedTimeSeries::rangeAdjust: the bucket that holds latestTime ends at
 * latestTime + 1; `input * scale` converts back to the integer value type by truncation).  The all-time level is one bucket
 * [firstTime, latestTime + 1). */
static int64_t bts_range_adjust(const gyo_bts *s, int64_t bstart, int64_t bnext, int64_t start, int64_t end, int64_t input)
{
	if (bstart <= s->latest_time && bnext > s->latest_time) bnext = s->latest_time + 1;
	if (start <= bstart && end >= bnext) return input;
	{
		const int64_t is = start > bstart ? start : bstart, ie = end < bnext ? end : bnext;
		const float scale = (float)(ie - is) * 1.f / (float)(bnext - bstart);
		return (int64_t)((float)input * scale);
	}
}

void gyo_bts_range(const gyo_bts *s, int64_t start, int64_t end, uint64_t *pcount, int64_t *psum)
{
	uint64_t cnt = 0;
	int64_t sum = 0;
	if (s->duration == 0) {
		const int64_t bstart = s->first_time, bnext = s->latest_time + 1;
		if (!(start >= bnext) && !(end <= bstart)) {
			cnt += (uint64_t)bts_range_adjust(s, bstart, bnext, start, end, (int64_t)s->tot_cnt);
			sum += bts_range_adjust(s, bstart, bnext, start, end, s->tot_sum);
		}
	} else {
		/* forEachBucket: from the bucket after the latest one (it belongs to the previous cycle) round to the latest */
		const int64_t n = (int64_t)s->nbuckets, d = s->duration;
		const int64_t time_mod = s->latest_time % d;
		int64_t dur_start = (s->latest_time / d) * d;
		const int64_t scaled = time_mod * n;
		const uint32_t latest_idx = (uint32_t)(scaled / d);
		int64_t scaled_next = scaled - scaled % d + d;
		uint32_t idx = latest_idx;
		int64_t bnext;
		dur_start -= d;
		bnext = (scaled_next + n - 1) / n + dur_start;
		for (;;) {
			int64_t bstart;
			if (++idx >= s->nbuckets) {
				idx = 0;
				dur_start += d;
				scaled_next = d;
			} else {
				scaled_next += d;
			}
			bstart = bnext;
			bnext = (scaled_next + n - 1) / n + dur_start;
			if (!(start >= bnext)) {
				if (end <= bstart) break;
				cnt += (uint64_t)bts_range_adjust(s, bstart, bnext, start, end, (int64_t)s->bcnt[idx]);
				sum += bts_range_adjust(s, bstart, bnext, start, end, s->bsum[idx]);
			}
			if (idx == latest_idx) break;
		}
	}
	*pcount = cnt;
	*psum = sum;
}

int gyo_mlh_level_for_start(const gyo_mlhist *h, int b, int64_t start) /* MultiLevelTimeSeries::getLevel(start) of histogram bucket b */
{
	for (int l = 0; l < GYO_MLH_LEVELS; l++) {
		if (h->s[b][l].duration == 0) return l;
		if (h->s[b][l].latest_time - h->s[b][l].duration <= start) return l;
	}
	return GYO_MLH_LEVELS - 1;
}

/* per histogram bucket {count, sum} over [starttime, endtime] as get_stats_for_period sees them (call gyo_mlh_flush first) */
void gyo_mlh_period(const gyo_mlhist *h, int64_t starttime, int64_t endtime, gyo_hist_serial *out /*[16]*/)
{
	const int64_t start = starttime, end = endtime + 1; /* :1383 */
	for (int b = 0; b < GYO_MAX_BUCKETS; b++) {
		out[b].sum = 0;
		out[b].count = 0;
	}
	for (int b = 0; b < h->nb; b++) {
		uint64_t c;
		int64_t s;
		gyo_bts_range(&h->s[b][gyo_mlh_level_for_start(h, b, start)], start, end, &c, &s);
		out[b].count = c;
		out[b].sum = s;
	}
}

void gyo_mlh_get_stats_for_period(const gyo_mlhist *h, int64_t starttime, int64_t endtime, const float *pcts, size_t npct, int64_t *values,
				  int64_t *tcount, int64_t *tsum, double *mean)
{
	gyo_hist_serial lv[GYO_MAX_BUCKETS];
	uint64_t counts[GYO_MAX_BUCKETS];
	int64_t tc = 0, ts = 0;
	gyo_mlh_period(h, starttime, endtime, lv);
	for (int b = 0; b < h->nb; b++) {
		counts[b] = lv[b].count;
		tc += (int64_t)lv[b].count;
		ts += lv[b].sum;
	}
	for (size_t i = 0; i < npct; i++) {
		int64_t v = gyo_bucket_max_threshold(h->kind, gyo_slab_percentile_idx(counts, (size_t)h->nb, (double)pcts[i] / 100.0));
		values[i] = v < 0 ? 0 : v; /* :1390-1392 */
	}
	if (tcount) *tcount = tc;
	if (tsum) *tsum = ts;
	if (mean) *mean = (double)ts / (tc != 0 ? tc : 1); /* :1398 */
}

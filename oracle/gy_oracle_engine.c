/*
 * TEST INFRASTRUCTURE ONLY -- sequential CPU restatement of the whole response-event hot path as ONE loop, the way the
 * reference runs it on a host core (common/gy_socket_stat.cc:1517-1677: filter -> listener lookup -> RESP_TIME_HASH bucket ->
 * histogram add -> CONN_BITMAP -> query count), extended with the builder-defined sketches (gy_oracle.c) so that every
 * register the GPU produces has a CPU twin.  Used by the parity tests as the checker and by bench.py as the "port" cpu_baseline.
 *
 * The listener table is an open-addressing table probed with the reference's get_uint64_hash (stand-in for the liburcu
 * RCU_HASH_TABLE, which is not installable here -- SURVEY 8c / A.5).
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "gy_oracle.h"

typedef struct gyo_engine {
	uint32_t max_services, nsvc, mask;
	int enable_td;
	uint64_t *keys; /* listener keys, ~0 = empty */
	uint32_t *vals; /* first listener (service slot) of the key's chain */
	uint64_t *svc_gid;
	/* the listeners of one (host, netns, port) key in registration order -- the order in which lookup_single_elem meets the nodes of one hash
	 * chain (the table is hashed ignoring the IP, common/gy_socket_stat.cc:1671; liburcu appends a node behind the equal-hash nodes already
	 * there -- not part of the reference tree, restated from its published behaviour); l_next = next listener of the key, ~0 = end */
	uint32_t *l_next, *l_ip32;
	uint8_t *l_ip128; /* [nsvc*16] */
	uint8_t *l_any;   /* TCP_LISTENER::is_any_ip_ */
	gyo_hist_serial *hist; /* [nsvc*16]; slot 15 = {total_count, (int64) max_val_seen} */
	uint16_t *bitmap;      /* [nsvc*64]: 32 rows of resp_bitmap_v4_, 32 rows of resp_bitmap_v6_ (common/gy_socket_stat.h:645, :665) */
	uint8_t hll[GYO_HLL_M];
	uint32_t *cms;         /* [D*W] */
	gyo_hist_serial ghist[16];
	int64_t gmax;
	gyo_td_buffered *td;
	/* per-batch staging for the digest: values bucketed by key (counting sort) */
	uint32_t *bcnt, *boff;
	uint64_t counters[4]; /* events, dropped_range, dropped_nolistener, accepted */
} gyo_engine;

static uint64_t lkey(uint32_t host, uint32_t netns, uint16_t port) { return ((uint64_t)host << 48) | ((uint64_t)netns << 16) | port; }

/* td_cap: the digests' buffer size (gys_config.td_pend_cap); 0 = the default GYO_TD_PEND_CAP */
gyo_engine *gyo_engine_new_cap(uint32_t max_services, int enable_td, uint32_t td_cap);
gyo_engine *gyo_engine_new(uint32_t max_services, int enable_td) { return gyo_engine_new_cap(max_services, enable_td, 0); }

gyo_engine *gyo_engine_new_cap(uint32_t max_services, int enable_td, uint32_t td_cap)
{
	gyo_engine *e = (gyo_engine *)calloc(1, sizeof(*e));
	uint32_t cap = 1;
	while (cap < 2 * (uint64_t)max_services) cap <<= 1;
	e->max_services = max_services;
	e->mask = cap - 1;
	e->enable_td = enable_td;
	e->keys = (uint64_t *)malloc((size_t)cap * 8);
	memset(e->keys, 0xFF, (size_t)cap * 8);
	e->vals = (uint32_t *)calloc(cap, 4);
	e->svc_gid = (uint64_t *)calloc(max_services, 8);
	e->hist = (gyo_hist_serial *)calloc((size_t)max_services * 16, sizeof(gyo_hist_serial));
	e->bitmap = (uint16_t *)calloc((size_t)max_services * 64, 2);
	e->l_next = (uint32_t *)malloc((size_t)max_services * 4);
	e->l_ip32 = (uint32_t *)calloc(max_services, 4);
	e->l_ip128 = (uint8_t *)calloc((size_t)max_services, 16);
	e->l_any = (uint8_t *)calloc(max_services, 1);
	e->cms = (uint32_t *)calloc((size_t)GYO_CMS_D * GYO_CMS_W, 4);
	e->gmax = LONG_MIN;
	for (uint32_t s = 0; s < max_services; s++) e->hist[(size_t)s * 16 + 15].sum = LONG_MIN;
	if (enable_td) {
		e->td = (gyo_td_buffered *)malloc((size_t)max_services * sizeof(gyo_td_buffered));
		for (uint32_t s = 0; s < max_services; s++) {
			if (td_cap) gyo_tdb_init_cap(&e->td[s], td_cap);
			else gyo_tdb_init(&e->td[s]);
		}
		e->boff = (uint32_t *)calloc((size_t)max_services + 1, 4);
	}
	e->bcnt = (uint32_t *)calloc(max_services, 4);
	return e;
}

void gyo_engine_free(gyo_engine *e)
{
	if (!e) return;
	if (e->td)
		for (uint32_t s = 0; s < e->max_services; s++) gyo_tdb_free(&e->td[s]);
	free(e->keys); free(e->vals); free(e->svc_gid); free(e->l_next); free(e->l_ip32); free(e->l_ip128); free(e->l_any); free(e->hist); free(e->bitmap); free(e->cms); free(e->td); free(e->bcnt); free(e->boff);
	free(e);
}

/* A listener with an address.  Registration follows insert_or_replace (common/gy_socket_stat.cc:1372, :7779) under the table's comparator
 * operator==(shared_ptr<TCP_LISTENER>, NS_IP_PORT) (common/gy_socket_stat.h:708-714): the first listener of the key that is_any_ip_ or is
 * bound to the new listener's address is replaced IN PLACE (its position in the chain goes to the new listener, its own slot keeps its
 * state but is no longer reachable); otherwise the new listener goes behind the others. */
int gyo_engine_register_addr(gyo_engine *e, uint32_t host_slot, uint64_t glob_id, uint32_t netns, uint16_t port, const uint8_t *ip, int is_v6, int is_any)
{
	const uint64_t k = lkey(host_slot, netns, port);
	uint32_t h = gyo_get_uint64_hash(k) & e->mask;
	const uint32_t s = e->nsvc;
	uint32_t n32 = 0;
	uint8_t n128[16] = {0};

	if (e->nsvc >= e->max_services) return -1;
	if (!is_any) gyo_ip_norm(ip, is_v6, &n32, n128);
	e->l_ip32[s] = n32;
	memcpy(e->l_ip128 + (size_t)s * 16, n128, 16);
	e->l_any[s] = (uint8_t)(is_any != 0);
	e->l_next[s] = 0xFFFFFFFFu;
	e->svc_gid[s] = glob_id;
	while (e->keys[h] != ~0ull && e->keys[h] != k) h = (h + 1) & e->mask;
	if (e->keys[h] == ~0ull) {
		e->keys[h] = k;
		e->vals[h] = s;
	} else {
		uint32_t *link = &e->vals[h];
		for (;;) {
			const uint32_t cur = *link;
			if (cur == 0xFFFFFFFFu) { /* nobody matched: appended */
				*link = s;
				break;
			}
			if (e->l_any[cur] || gyo_ip_equal(e->l_ip32[cur], e->l_ip128 + (size_t)cur * 16, n32, n128)) { /* replaced in place */
				e->l_next[s] = e->l_next[cur];
				*link = s;
				break;
			}
			link = &e->l_next[cur];
		}
	}
	return (int)e->nsvc++;
}

int gyo_engine_register(gyo_engine *e, uint32_t host_slot, uint64_t glob_id, uint32_t netns, uint16_t port)
{
	return gyo_engine_register_addr(e, host_slot, glob_id, netns, port, NULL, 0, 1);
}

/* n any-address listeners of one host in one call (bench.py registers 10^7 of them); returns the first slot or -1 */
int gyo_engine_register_bulk(gyo_engine *e, uint32_t host_slot, const uint64_t *glob_id, const uint32_t *netns, const uint16_t *port, uint32_t n)
{
	const int first = (int)e->nsvc;
	for (uint32_t i = 0; i < n; i++)
		if (gyo_engine_register_addr(e, host_slot, glob_id[i], netns[i], port[i], NULL, 0, 1) < 0) return -1;
	return first;
}

/* listener_tbl_.lookup_single_elem(ser_nsipport, hash ignoring the IP) (common/gy_socket_stat.cc:1671): the first listener of the key for
 * which operator==(listener, ser_nsipport) holds (common/gy_socket_stat.h:708-714); e32 / e128 = the event's server address as GY_IP_ADDR */
static uint32_t lookup(const gyo_engine *e, uint64_t k, uint32_t e32, const uint8_t e128[16])
{
	uint32_t h = gyo_get_uint64_hash(k) & e->mask;
	for (;;) {
		if (e->keys[h] == k) {
			for (uint32_t s = e->vals[h]; s != 0xFFFFFFFFu; s = e->l_next[s])
				if (e->l_any[s] || gyo_ip_equal(e->l_ip32[s], e->l_ip128 + (size_t)s * 16, e32, e128)) return s;
			return 0xFFFFFFFFu;
		}
		if (e->keys[h] == ~0ull) return 0xFFFFFFFFu;
		h = (h + 1) & e->mask;
	}
}

static uint16_t bswap16(uint16_t v) { return (uint16_t)((v >> 8) | (v << 8)); }

/* where one pass over a range of events accumulates the registers that are shared between services (the per-service records are
 * addressed by slot and a service belongs to one host, so ranges made of whole hosts never touch the same record) */
typedef struct {
	uint8_t *hll;
	uint32_t *cms;
	gyo_hist_serial *ghist; /* [16] */
	int64_t *gmax;
	uint64_t *counters;     /* [4] */
	int shared;             /* unused (every sink is private to its thread) */
} resp_sinks;

static void cms_add_shared(uint32_t *tbl, const uint32_t *words, uint32_t nwords, uint32_t weight)
{
	uint32_t cols[GYO_CMS_D];
	gyo_cms_cols(words, nwords, cols);
	for (uint32_t r = 0; r < GYO_CMS_D; r++) __atomic_fetch_add(&tbl[(size_t)r * GYO_CMS_W + cols[r]], weight, __ATOMIC_RELAXED);
}

/* events [i0, i1) of a batch; seg = index of the segment containing i0.  slot_of / val_of (NULL without t-digests) get one entry
 * per event; bcnt[slot] counts the kept events of each service. */
static void resp_range(gyo_engine *e, const uint8_t *ev24, uint64_t i0, uint64_t i1, const uint32_t *seg_host, const uint64_t *seg_first, uint32_t nsegs,
		       uint32_t seg, uint32_t *slot_of, int32_t *val_of, resp_sinks k, int v6)
{
	/* v6: 48-byte tcp_ipv6_resp_event_t (common/gy_ebpf_kernel.h:113-118): 16-byte saddr, 16-byte daddr, then netns / ports / times laid out
	 * as in the 24-byte IPv4 event (handle_ipv6_resp_event common/gy_socket_stat.cc:1535-1551) */
	const size_t stride = v6 ? 48 : 24, al = v6 ? 16 : 4;
	for (uint64_t i = i0; i < i1; i++) {
		const uint8_t *p = ev24 + i * stride;
		const uint8_t *psaddr = p, *pdaddr = p + al, *q = p + 2 * al;
		uint32_t netns, lsnd, lrcv, tresp, slot, b, s32;
		uint16_t sport_be, dport_be, sport, dport;
		uint8_t s128[16];

		memcpy(&netns, q, 4);
		memcpy(&sport_be, q + 4, 2); memcpy(&dport_be, q + 6, 2);
		memcpy(&lsnd, q + 8, 4); memcpy(&lrcv, q + 12, 4);
		while (seg + 1 < nsegs && seg_first[seg + 1] <= i) seg++;
		if (slot_of) slot_of[i] = 0xFFFFFFFFu;
		k.counters[0]++;
		tresp = lsnd - lrcv;                      /* gy_socket_stat.cc:1519 */
		if (tresp > 1000000u) {                   /* :1521-1524 */
			k.counters[1]++;
			continue;
		}
		sport = bswap16(sport_be);                /* ntohs :1526-1527 */
		dport = bswap16(dport_be);
		gyo_ip_norm(psaddr, v6, &s32, s128);      /* NS_IP_PORT nsipport(pevent->tup.saddr, ...) :1529 / :1547 */
		slot = lookup(e, lkey(seg_host[seg], netns, sport), s32, s128); /* listener_tbl_ lookup: hash ignoring the IP, comparator with it :1671 */
		if (slot == 0xFFFFFFFFu) {
			k.counters[2]++;
			continue;
		}
		k.counters[3]++;
		b = gyo_bucket(GYO_RESP_TIME_HASH, (int64_t)tresp);
		{
			gyo_hist_serial *h = &e->hist[(size_t)slot * 16];
			h[b].count++;                     /* HIST_SERIAL::add gy_statistics.h:463-467 */
			h[b].sum += (int64_t)tresp;
			h[15].count++;                    /* total_count_ */
			if (h[15].sum < (int64_t)tresp) h[15].sum = (int64_t)tresp; /* max_val_seen_ */
		}
		gyo_conn_bitmap_add(&e->bitmap[(size_t)slot * 64 + (v6 ? 32 : 0)], dport, (uint8_t)b); /* resp_bitmap_v4_ / _v6_.add_response :1580, :1587 */
		k.ghist[b].count++;
		k.ghist[b].sum += (int64_t)tresp;
		k.ghist[15].count++;
		if (*k.gmax < (int64_t)tresp) *k.gmax = (int64_t)tresp;
		{
			uint32_t w[10];
			const uint32_t nw = gyo_pair_ip_port_words(pdaddr, v6, dport, psaddr, v6, sport, w);
			gyo_hll_add_words(k.hll, GYO_HLL_P, w, nw);
		}
		/* Count-Min of events per service key: the table is linear in the per-service counts, so the event only counts (the
		 * service's record is owned by this thread) and the rows are built once per batch from the counts (cms_from_counts):
		 * the same registers as one gyo_cms_add(key, 1) per event, without four hashes and four shared counters per event */
		e->bcnt[slot]++;
		if (slot_of) {
			slot_of[i] = slot;
			val_of[i] = (int32_t)tresp;
		}
	}
}

/* rows of the Count-Min table from the batch's per-service event counts, services [k0, k1); shared: several threads add at once */
static void cms_from_counts(gyo_engine *e, uint32_t k0, uint32_t k1, int shared)
{
	for (uint32_t s = k0; s < k1; s++) {
		if (!e->bcnt[s]) continue;
		uint32_t gw[2];
		gw[0] = (uint32_t)(e->svc_gid[s] & 0xFFFFFFFFu);
		gw[1] = (uint32_t)(e->svc_gid[s] >> 32);
		if (shared) cms_add_shared(e->cms, gw, 2, e->bcnt[s]);
		else gyo_cms_add(e->cms, gw, 2, e->bcnt[s]);
	}
}

static resp_sinks own_sinks(gyo_engine *e)
{
	resp_sinks k = {e->hll, e->cms, e->ghist, &e->gmax, e->counters, 0};
	return k;
}

/* One batch of 24-byte tcp_ipv4_resp_event_t (common/gy_ebpf_kernel.h:106-111); segment s covers events
 * [seg_first[s], seg_first[s+1]) of host seg_host[s]. */
static void resp_batch_fam(gyo_engine *e, const uint8_t *ev24, uint64_t n, const uint32_t *seg_host, const uint64_t *seg_first, uint32_t nsegs, int v6)
{
	uint32_t *slot_of = NULL;
	int32_t *val_of = NULL;

	if (e->enable_td) {
		slot_of = (uint32_t *)malloc((size_t)(n ? n : 1) * 4);
		val_of = (int32_t *)malloc((size_t)(n ? n : 1) * 4);
	}
	memset(e->bcnt, 0, (size_t)e->nsvc * 4);
	resp_range(e, ev24, 0, n, seg_host, seg_first, nsegs, 0, slot_of, val_of, own_sinks(e), v6);
	cms_from_counts(e, 0, e->nsvc, 0);
	if (e->enable_td) {
		/* buffered digest(key) <- add_batch(multiset of this batch's values of the key): append, or one merge of buffer + batch */
		int32_t *staged = (int32_t *)malloc((size_t)(n ? n : 1) * 4);
		uint32_t run = 0;
		for (uint32_t s = 0; s < e->nsvc; s++) {
			e->boff[s] = run;
			run += e->bcnt[s];
		}
		e->boff[e->nsvc] = run;
		for (uint64_t i = 0; i < n; i++)
			if (slot_of[i] != 0xFFFFFFFFu) staged[e->boff[slot_of[i]]++] = val_of[i];
		run = 0;
		for (uint32_t s = 0; s < e->nsvc; s++) {
			if (e->bcnt[s]) gyo_tdb_add_batch(&e->td[s], staged + run, e->bcnt[s]);
			run += e->bcnt[s];
		}
		free(staged);
		free(slot_of);
		free(val_of);
	}
}

void gyo_engine_resp_batch(gyo_engine *e, const uint8_t *ev24, uint64_t n, const uint32_t *seg_host, const uint64_t *seg_first, uint32_t nsegs)
{
	resp_batch_fam(e, ev24, n, seg_host, seg_first, nsegs, 0);
}

/* One batch of 48-byte tcp_ipv6_resp_event_t (common/gy_ebpf_kernel.h:113-118) -- handle_ipv6_resp_event (common/gy_socket_stat.cc:1535-1551):
 * the listener's shared histogram / query count / digest, its resp_bitmap_v6_ rows */
void gyo_engine_resp_batch_v6(gyo_engine *e, const uint8_t *ev48, uint64_t n, const uint32_t *seg_host, const uint64_t *seg_first, uint32_t nsegs)
{
	resp_batch_fam(e, ev48, n, seg_host, seg_first, nsegs, 1);
}

/* The same batch on nthreads host threads ("all cores" CPU baseline): the segments (hosts) are cut into contiguous ranges, one per
 * thread -- the reference pins a partha's batches to one L2 thread the same way (server/gy_mconnhdlr.cc:16252).  Every segment must
 * be a different host (a service's records are then owned by one thread).  Nothing shared is touched per event: every thread keeps
 * PRIVATE HyperLogLog registers (merged by max at the end), its own all-service histogram and counters (summed at the end), the events
 * only count per service and the Count-Min rows are built from those counts afterwards, in parallel over service ranges -- what the GPU
 * does per workgroup.  The per-service digests are re-clustered in parallel over service ranges.  The resulting state is identical to
 * gyo_engine_resp_batch's. */
typedef struct {
	gyo_engine *e;
	const uint8_t *ev24;
	uint64_t n;
	const uint32_t *seg_host;
	const uint64_t *seg_first;
	uint32_t nsegs, s0, s1;
	uint32_t *slot_of;
	int32_t *val_of, *staged;
	gyo_hist_serial ghist[16];
	int64_t gmax;
	uint64_t counters[4];
	uint32_t k0, k1; /* service range of the digest phase */
	const uint32_t *kstart;
	uint8_t *hll;    /* the thread's PRIVATE HyperLogLog registers (16 KiB), merged by max at the end of the pass */
} mt_worker;

static void *mt_pass1(void *arg)
{
	mt_worker *w = (mt_worker *)arg;
	if (w->s0 >= w->s1) return NULL;
	const uint64_t i0 = w->seg_first[w->s0], i1 = w->s1 < w->nsegs ? w->seg_first[w->s1] : w->n;
	resp_sinks k = {w->hll, w->e->cms, w->ghist, &w->gmax, w->counters, 0};
	resp_range(w->e, w->ev24, i0, i1, w->seg_host, w->seg_first, w->nsegs, w->s0, w->slot_of, w->val_of, k, 0);
	return NULL;
}

static void *mt_cms(void *arg) /* Count-Min rows from the per-service counts, in parallel over service ranges (one add per service and row) */
{
	mt_worker *w = (mt_worker *)arg;
	cms_from_counts(w->e, w->k0, w->k1, 1);
	return NULL;
}

static void *mt_scatter(void *arg) /* a thread's events only name its own hosts' services: the cursors it moves are its own */
{
	mt_worker *w = (mt_worker *)arg;
	if (w->s0 >= w->s1) return NULL;
	const uint64_t i0 = w->seg_first[w->s0], i1 = w->s1 < w->nsegs ? w->seg_first[w->s1] : w->n;
	for (uint64_t i = i0; i < i1; i++)
		if (w->slot_of[i] != 0xFFFFFFFFu) w->staged[w->e->boff[w->slot_of[i]]++] = w->val_of[i];
	return NULL;
}

static void *mt_digest(void *arg)
{
	mt_worker *w = (mt_worker *)arg;
	for (uint32_t s = w->k0; s < w->k1; s++)
		if (w->e->bcnt[s]) gyo_tdb_add_batch(&w->e->td[s], w->staged + w->kstart[s], w->e->bcnt[s]);
	return NULL;
}

#include <pthread.h>

static void run_all(mt_worker *w, uint32_t nt, void *(*fn)(void *))
{
	pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
	for (uint32_t t = 0; t < nt; t++) pthread_create(&th[t], NULL, fn, &w[t]);
	for (uint32_t t = 0; t < nt; t++) pthread_join(th[t], NULL);
	free(th);
}

void gyo_engine_resp_batch_mt(gyo_engine *e, const uint8_t *ev24, uint64_t n, const uint32_t *seg_host, const uint64_t *seg_first, uint32_t nsegs,
			      uint32_t nthreads)
{
	if (nthreads <= 1 || nsegs <= 1) {
		gyo_engine_resp_batch(e, ev24, n, seg_host, seg_first, nsegs);
		return;
	}
	if (nthreads > nsegs) nthreads = nsegs;
	uint32_t *slot_of = NULL, *kstart = NULL;
	int32_t *val_of = NULL, *staged = NULL;
	if (e->enable_td) {
		slot_of = (uint32_t *)malloc((size_t)(n ? n : 1) * 4);
		val_of = (int32_t *)malloc((size_t)(n ? n : 1) * 4);
		staged = (int32_t *)malloc((size_t)(n ? n : 1) * 4);
		kstart = (uint32_t *)malloc(((size_t)e->nsvc + 1) * 4);
	}
	memset(e->bcnt, 0, (size_t)e->nsvc * 4);
	uint8_t *hll_all = (uint8_t *)calloc((size_t)nthreads, GYO_HLL_M);
	mt_worker *w = (mt_worker *)calloc(nthreads, sizeof(mt_worker));
	for (uint32_t t = 0; t < nthreads; t++) {
		w[t].e = e;
		w[t].ev24 = ev24;
		w[t].n = n;
		w[t].seg_host = seg_host;
		w[t].seg_first = seg_first;
		w[t].nsegs = nsegs;
		w[t].s0 = (uint32_t)((uint64_t)nsegs * t / nthreads);
		w[t].s1 = (uint32_t)((uint64_t)nsegs * (t + 1) / nthreads);
		w[t].slot_of = slot_of;
		w[t].val_of = val_of;
		w[t].staged = staged;
		w[t].gmax = LONG_MIN;
		w[t].k0 = (uint32_t)((uint64_t)e->nsvc * t / nthreads);
		w[t].k1 = (uint32_t)((uint64_t)e->nsvc * (t + 1) / nthreads);
		w[t].kstart = kstart;
		w[t].hll = hll_all + (size_t)t * GYO_HLL_M;
	}
	run_all(w, nthreads, mt_pass1);
	run_all(w, nthreads, mt_cms);
	for (uint32_t t = 0; t < nthreads; t++) gyo_hll_merge(e->hll, w[t].hll, GYO_HLL_P);
	free(hll_all);
	for (uint32_t t = 0; t < nthreads; t++) { /* the small per-thread registers: sums / max, independent of the order */
		for (int b = 0; b < 16; b++) {
			e->ghist[b].count += w[t].ghist[b].count;
			e->ghist[b].sum += w[t].ghist[b].sum;
		}
		if (e->gmax < w[t].gmax) e->gmax = w[t].gmax;
		for (int c = 0; c < 4; c++) e->counters[c] += w[t].counters[c];
	}
	if (e->enable_td) {
		uint32_t run = 0;
		for (uint32_t s = 0; s < e->nsvc; s++) {
			e->boff[s] = run;
			kstart[s] = run;
			run += e->bcnt[s];
		}
		e->boff[e->nsvc] = run;
		run_all(w, nthreads, mt_scatter);
		run_all(w, nthreads, mt_digest);
		free(staged);
		free(slot_of);
		free(val_of);
		free(kstart);
	}
	free(w);
}

/* histogram-only variant == what the reference itself computes per event (no sketches): the "reference work" CPU baseline */
void gyo_engine_resp_batch_histonly(gyo_engine *e, const uint8_t *ev24, uint64_t n, const uint32_t *seg_host, const uint64_t *seg_first, uint32_t nsegs)
{
	uint32_t seg = 0;
	for (uint64_t i = 0; i < n; i++) {
		const uint8_t *p = ev24 + i * 24;
		uint32_t netns, lsnd, lrcv, tresp, slot, b;
		uint16_t sport_be, dport_be;

		memcpy(&netns, p + 8, 4); memcpy(&sport_be, p + 12, 2); memcpy(&dport_be, p + 14, 2);
		memcpy(&lsnd, p + 16, 4); memcpy(&lrcv, p + 20, 4);
		while (seg + 1 < nsegs && seg_first[seg + 1] <= i) seg++;
		tresp = lsnd - lrcv;
		if (tresp > 1000000u) continue;
		{
			uint32_t s32;
			uint8_t s128[16];
			gyo_ip_norm(p, 0, &s32, s128);
			slot = lookup(e, lkey(seg_host[seg], netns, bswap16(sport_be)), s32, s128);
		}
		if (slot == 0xFFFFFFFFu) continue;
		b = gyo_bucket(GYO_RESP_TIME_HASH, (int64_t)tresp);
		{
			gyo_hist_serial *h = &e->hist[(size_t)slot * 16];
			h[b].count++;
			h[b].sum += (int64_t)tresp;
			h[15].count++;
			if (h[15].sum < (int64_t)tresp) h[15].sum = (int64_t)tresp;
		}
		gyo_conn_bitmap_add(&e->bitmap[(size_t)slot * 64], bswap16(dport_be), (uint8_t)b);
	}
}

uint32_t gyo_engine_nsvc(const gyo_engine *e) { return e->nsvc; }
const gyo_hist_serial *gyo_engine_hist(const gyo_engine *e) { return e->hist; }
const uint16_t *gyo_engine_bitmap(const gyo_engine *e) { return e->bitmap; }
const uint8_t *gyo_engine_hll(const gyo_engine *e) { return e->hll; }
const uint32_t *gyo_engine_cms(const gyo_engine *e) { return e->cms; }
const gyo_hist_serial *gyo_engine_ghist(const gyo_engine *e) { return e->ghist; }
int64_t gyo_engine_gmax(const gyo_engine *e) { return e->gmax; }
const gyo_td_buffered *gyo_engine_td(const gyo_engine *e, uint32_t slot) { return &e->td[slot]; }
const uint64_t *gyo_engine_counters(const gyo_engine *e) { return e->counters; }
/* window roll: clear the windowed sketches (CONN_BITMAP secs_to_reset_ = 5; HLL/CMS are per window) keeping histograms/digests */
void gyo_engine_window_clear(gyo_engine *e, int clear_hist)
{
	memset(e->bitmap, 0, (size_t)e->max_services * 128);
	memset(e->hll, 0, sizeof(e->hll));
	memset(e->cms, 0, (size_t)GYO_CMS_D * GYO_CMS_W * 4);
	memset(e->ghist, 0, sizeof(e->ghist));
	e->gmax = LONG_MIN;
	if (clear_hist) {
		memset(e->hist, 0, (size_t)e->max_services * 16 * sizeof(gyo_hist_serial));
		for (uint32_t s = 0; s < e->max_services; s++) e->hist[(size_t)s * 16 + 15].sum = LONG_MIN;
	}
}

/* ---------------------------------------------------------------- t-digest accuracy stress (test support)
 * nkeys independent keys, each streaming nvals values through the buffered digest in random batches of 1..2*bmean values (so a key
 * re-clusters about nvals / GYO_TD_PEND_CAP times); afterwards the digest's quantiles are ranked against the exact sort of the
 * key's stream.  dist: 0 lognormal(mu_key ~ N(3,1), 1.5) floored to integer ms (the SURVEY 8d latency law), 1 narrow uniform,
 * 2 the same lognormal with a drifting scale (x1 -> x4 over the stream), 3 lognormal drawn once (330 values) and cycled.
 * out[3*i + {0,1,2}] = max rank error over the keys of quantile qs[i]; keys are cut over nthreads threads. */
#include <math.h>
typedef struct {
	uint32_t k0, k1, nvals, bmean;
	int dist;
	uint64_t seed;
	const double *qs;
	uint32_t nq;
	double *worst; /* [nq] */
} td_stress_job;

static uint64_t sm64(uint64_t *s)
{
	uint64_t x = (*s += 0x9E3779B97F4A7C15ull);
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}
static double sm_unif(uint64_t *s) { return ((double)(sm64(s) >> 11) + 0.5) / 9007199254740992.0; }
static double sm_gauss(uint64_t *s) { return sqrt(-2.0 * log(sm_unif(s))) * cos(6.283185307179586 * sm_unif(s)); }
static int cmp_int(const void *a, const void *b) { const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return (x > y) - (x < y); }

static void *td_stress_run(void *arg)
{
	td_stress_job *j = (td_stress_job *)arg;
	int32_t *x = (int32_t *)malloc((size_t)j->nvals * 4), *srt = (int32_t *)malloc((size_t)j->nvals * 4);
	for (uint32_t q = 0; q < j->nq; q++) j->worst[q] = 0.0;
	for (uint32_t k = j->k0; k < j->k1; k++) {
		uint64_t s = j->seed + (uint64_t)k * 0x632BE59BD9B4E019ull;
		gyo_td_buffered b;
		const double mu = 3.0 + sm_gauss(&s);
		gyo_tdb_init(&b);
		for (uint32_t i = 0; i < j->nvals; i++) {
			double v;
			if (j->dist == 1) v = 1000.0 + sm_unif(&s) * 100.0;
			else if (j->dist == 2) v = exp(mu + 1.5 * sm_gauss(&s)) * (1.0 + 3.0 * (double)i / (double)j->nvals);
			else if (j->dist == 3 && i >= 330) { x[i] = x[i - 330]; continue; }
			else v = exp(mu + 1.5 * sm_gauss(&s));
			if (v > 1e6) v = 1e6;
			x[i] = (int32_t)v;
		}
		for (uint32_t pos = 0; pos < j->nvals;) {
			uint32_t m = 1u + (uint32_t)(sm64(&s) % (2u * j->bmean));
			if (m > j->nvals - pos) m = j->nvals - pos;
			gyo_tdb_add_batch(&b, x + pos, m);
			pos += m;
		}
		memcpy(srt, x, (size_t)j->nvals * 4);
		qsort(srt, j->nvals, 4, cmp_int);
		for (uint32_t qi = 0; qi < j->nq; qi++) {
			const double g = gyo_tdb_quantile(&b, j->qs[qi]), q = j->qs[qi];
			uint32_t a = 0, c = j->nvals, lo, hi;
			while (a < c) { const uint32_t m = (a + c) / 2; if ((double)srt[m] < g) a = m + 1; else c = m; }
			lo = a; a = 0; c = j->nvals;
			while (a < c) { const uint32_t m = (a + c) / 2; if ((double)srt[m] <= g) a = m + 1; else c = m; }
			hi = a;
			{
				const double l = (double)lo / j->nvals, h = (double)hi / j->nvals;
				const double err = (l <= q && q <= h) ? 0.0 : (fabs(l - q) < fabs(h - q) ? fabs(l - q) : fabs(h - q));
				if (err > j->worst[qi]) j->worst[qi] = err;
			}
		}
	}
	free(x);
	free(srt);
	return NULL;
}

void gyo_td_stress(uint32_t nkeys, uint32_t nvals, uint32_t bmean, int dist, uint64_t seed, const double *qs, uint32_t nq, uint32_t nthreads, double *out)
{
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 64) nthreads = 64;
	td_stress_job jobs[64];
	double worst[64][8];
	pthread_t th[64];
	if (nq > 8) nq = 8;
	for (uint32_t t = 0; t < nthreads; t++) {
		jobs[t] = (td_stress_job){(uint32_t)((uint64_t)nkeys * t / nthreads), (uint32_t)((uint64_t)nkeys * (t + 1) / nthreads), nvals, bmean, dist, seed, qs, nq, worst[t]};
		pthread_create(&th[t], NULL, td_stress_run, &jobs[t]);
	}
	for (uint32_t q = 0; q < nq; q++) out[q] = 0.0;
	for (uint32_t t = 0; t < nthreads; t++) {
		pthread_join(th[t], NULL);
		for (uint32_t q = 0; q < nq; q++)
			if (worst[t][q] > out[q]) out[q] = worst[t][q];
	}
}

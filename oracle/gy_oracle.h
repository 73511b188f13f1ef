/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the gyeeta_amd hot path.
 *
 * Plain-C restatement of the reference algorithms on the madhava/shyama aggregation path plus
 * the frozen CPU definitions of the HLL / Count-Min / t-digest sketches the engine adds (the reference has none of its
 * own; see DESIGN.md "oracle pinning").  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this; the product library (gyeeta_amd/csrc) never links or calls it.
 *
 * Pinning status:
 *   - jhash / key hashes / bucket hashes / GY_HISTOGRAM arithmetic: PINNED against test/test_histogram.cc:28-147 asserts,
 *     the SURVEY 8c KATs and against oracle/_ref (the reference's own headers compiled here) in tests/test_oracle_vs_ref.py.
 *   - wire structs + L1 validators (COMM_HEADER, EVENT_NOTIFY, TCP_CONN_NOTIFY, LISTENER_STATE_NOTIFY, LISTENER_DAY_STATS member
 *     offsets, get_elem_size, COMM_HEADER / TCP_CONN_NOTIFY / LISTENER_STATE_NOTIFY::validate) and MS_CLUSTER_STATE::STATE_ONE::add_stats:
 *     PINNED against common/gy_comm_proto.h / .cc compiled into oracle/_ref (gy_sys_hardware.h replaced by a 16-byte GY_MACHINE_ID
 *     stand-in, oracle/build_ref.sh): tests/test_wire.py, tests/test_oracle_vs_ref.py.
 *   - LISTEN_SUMM_STATS<int>::update (server/gy_msocket.h:840-882) and CLUSTER_STATE_ONE::update_from_state
 *     (server/gy_mconnhdlr.cc:16032-16050) live in server/ files that need folly/liburcu/boost: the two classes are small and
 *     self-contained, oracle/build_ref.sh cuts their text out of the reference in place and compiles it into oracle/_ref ->
 *     PINNED (tests/test_oracle_vs_ref.py::test_listen_summ_stats_and_cluster_state_one_equal_the_reference_classes).
 *   - listener lookup of a response event (round 5): GY_IP_ADDR built from raw IPv4 / IPv6 bytes (ip32_be_ shares its storage with
 *     embedded_ipv4_: 2002::/16, ::ffff:a.b.c.d and 64:ff9b::/32 addresses compare and hash as the IPv4 address they embed),
 *     GY_IP_ADDR::operator==, NS_IP_PORT and the listener comparator (common/gy_socket_stat.h:708-714), PAIR_IP_PORT::get_hash:
 *     PINNED against common/gy_inet_inc.h / gy_common_inc.h compiled into oracle/_ref (tests/test_listener_addr.py).  The ORDER in
 *     which a lookup meets several listeners of one (netns, port) hash chain is liburcu's (cds_lfht, absent from the tree):
 *     restated (first registered first), parity unpinned.
 *   - folly::MultiLevelTimeSeries ring arithmetic (gy_oracle_levels.c), the Postgres tdigest external forms, the criteria walk
 *     (gy_oracle_query.c): sources absent from /root/reference -> restated from the published algorithms, "parity unpinned".
 *   - HLL / CMS / t-digest: builder-defined (no reference implementation exists) -> "parity unpinned" vs reference; the
 *     acceptance test against reference behaviour is rank error vs exact sort + bucket agreement with GY_HISTOGRAM.
 */
#ifndef GY_ORACLE_H
#define GY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- hashing (common/jhash.h) */
uint32_t gyo_jhash(const void *key, uint32_t length, uint32_t initval);
uint32_t gyo_jhash2(const uint32_t *k, uint32_t length, uint32_t initval);
uint32_t gyo_jhash_3words(uint32_t a, uint32_t b, uint32_t c, uint32_t initval);
uint32_t gyo_jhash_2words(uint32_t a, uint32_t b, uint32_t initval);
uint32_t gyo_jhash_1word(uint32_t a, uint32_t initval);
uint32_t gyo_get_uint64_hash(uint64_t k);
uint32_t gyo_get_uint32_hash(uint32_t k);

/* key byte-packing + hash (ip: 4 bytes network order or 16 bytes in6_addr; port host order) */
uint32_t gyo_ip_port_words(const uint8_t *ip, int is_v6, uint16_t port, int ignore_ip, uint32_t out[5]);
uint32_t gyo_ip_port_hash(const uint8_t *ip, int is_v6, uint16_t port, int ignore_ip);
uint32_t gyo_ns_ip_port_hash(const uint8_t *ip, int is_v6, uint16_t port, uint64_t inode, int ignore_ip);
uint32_t gyo_pair_ip_port_words(const uint8_t *cip, int c6, uint16_t cport, const uint8_t *sip, int s6, uint16_t sport,
				uint32_t out[10]);
uint32_t gyo_pair_ip_port_hash(const uint8_t *cip, int c6, uint16_t cport, const uint8_t *sip, int s6, uint16_t sport);
uint32_t gyo_machine_id_hash(uint64_t first, uint64_t second);

/* 64-bit sketch hash: (jhash2(seed 0xceedfead) << 32) | jhash2(seed 0x9e3779b9)   */
uint64_t gyo_hash64(const uint32_t *words, uint32_t nwords);

/* ---------------------------------------------------------------- bucket hashes + GY_HISTOGRAM */
enum {
	GYO_RESP_TIME_HASH = 0,
	GYO_SEMI_LOG_HASH = 1,
	GYO_SEMI_LOG_HASH_LO = 2,
	GYO_DURATION_HASH = 3,
	GYO_HASH_10_5000 = 4,
	GYO_HASH_5_250 = 5,
	GYO_HASH_1_3000 = 6,
	GYO_PERCENT_HASH = 7,      /* FIXED_DIFF_HASH<int64_t,0,100,10>, histogram T=int */
	GYO_FIXED_9_26_5 = 8,      /* FIXED_DIFF_HASH<int8_t,9,26,5>,  T=int8_t (test_histogram.cc) */
	GYO_FIXED_N15_N3_4 = 9,    /* FIXED_DIFF_HASH<int,-15,-3,4>,   T=int    (test_histogram.cc) */
	GYO_NKINDS = 10
};

#define GYO_MAX_BUCKETS 16

typedef struct {
	uint64_t count;
	int64_t sum;
} gyo_hist_serial; /* == HIST_SERIAL, 16 bytes */

typedef struct {
	int kind;
	int nbuckets;
	gyo_hist_serial stats[GYO_MAX_BUCKETS];
	uint64_t total_count;
	int64_t max_val_seen;
} gyo_hist;

typedef struct {
	int64_t data_value;
	int64_t sum;
	uint64_t count;
	float percentile;
} gyo_hist_data; /* == HIST_DATA */

int gyo_hist_nbuckets(int kind);
uint32_t gyo_bucket(int kind, int64_t data);
void gyo_bucket_many(int kind, const int64_t *v, size_t n, uint32_t *out);
int64_t gyo_bucket_max_threshold(int kind, size_t id);
void gyo_hist_init(gyo_hist *h, int kind);
uint32_t gyo_hist_add(gyo_hist *h, int64_t data);
void gyo_hist_add_many(gyo_hist *h, const int64_t *v, size_t n);
void gyo_hist_merge(gyo_hist *dst, const gyo_hist *src);
void gyo_hist_percentiles(const gyo_hist *h, gyo_hist_data *pdata, size_t npct, uint64_t *total, int64_t *maxv, float *pavg);
/* percentiles from raw serialized arrays (what the GPU exports) */
void gyo_percentiles_raw(int kind, const gyo_hist_serial *stats, uint64_t total_count, gyo_hist_data *pdata, size_t npct, float *pavg);

/* keyed bulk ingest: nkeys histograms of one kind; key index per value; the layout matches the GPU export:
 * stats[key*16 + b], total[key], maxv[key] */
void gyo_keyed_hist_ingest(int kind, const uint32_t *keyidx, const int32_t *vals, size_t n, gyo_hist_serial *stats /*[nkeys*16]*/,
			   uint64_t *total, int64_t *maxv);

/* CONN_BITMAP (common/gy_socket_stat.h:390-454): respmap[32] of 15-bit masks, slot = cli_port & 0x1F */
void gyo_conn_bitmap_add(uint16_t respmap[32], uint16_t cli_port, uint8_t bucket);
void gyo_conn_bitmap_breakup(const uint16_t respmap[32], uint8_t nconn_arr[15]);
void gyo_conn_bitmap_breakup2(const uint16_t respmap[64], uint8_t nconn_arr[15]);
int gyo_ip_norm(const uint8_t *ip, int is_v6, uint32_t *ip32, uint8_t ip128[16]);
uint32_t gyo_pair_words_obj(uint32_t c32, const uint8_t c128[16], uint16_t cport, uint32_t s32, const uint8_t s128[16], uint16_t sport, uint32_t out[10]);
int gyo_ip_equal(uint32_t a32, const uint8_t a128[16], uint32_t b32, const uint8_t b128[16]);

/* ---------------------------------------------------------------- HLL / CMS (builder-defined, frozen in DESIGN.md) */
#define GYO_HLL_P 14
#define GYO_HLL_M (1u << GYO_HLL_P)
#define GYO_CMS_D 4
#define GYO_CMS_W 65536u

void gyo_hll_idx_rank(uint64_t h64, int p, uint32_t *idx, uint8_t *rank);
void gyo_hll_add(uint8_t *regs, int p, uint64_t h64);
void gyo_hll_add_words(uint8_t *regs, int p, const uint32_t *words, uint32_t nwords);
void gyo_hll_merge(uint8_t *dst, const uint8_t *src, int p);
double gyo_hll_estimate(const uint8_t *regs, int p);

void gyo_cms_cols(const uint32_t *words, uint32_t nwords, uint32_t cols[GYO_CMS_D]);
void gyo_cms_add(uint32_t *tbl /*[D*W]*/, const uint32_t *words, uint32_t nwords, uint32_t weight);
uint32_t gyo_cms_query(const uint32_t *tbl, const uint32_t *words, uint32_t nwords);
void gyo_cms64_add(uint64_t *tbl /*[D*W]*/, const uint32_t *words, uint32_t nwords, uint64_t weight);
uint64_t gyo_cms64_query(const uint64_t *tbl, const uint32_t *words, uint32_t nwords);

/* ---------------------------------------------------------------- t-digest (k-bucketed merging digest, exact integer) */
#define GYO_TD_NB 200

typedef struct {
	int64_t sum[GYO_TD_NB];
	uint32_t cnt[GYO_TD_NB];
	int32_t vmin, vmax; /* valid when total > 0 */
} gyo_tdigest;

extern const uint64_t gyo_td_bnd[GYO_TD_NB + 1];

void gyo_td_init(gyo_tdigest *d);
uint64_t gyo_td_total(const gyo_tdigest *d);
uint32_t gyo_td_cluster(uint64_t mid2, uint64_t twoN);
/* merge m new values (any order; sorted internally) into d */
void gyo_td_merge_values(gyo_tdigest *d, const int32_t *vals, size_t m);
/* merge another digest's clusters into d (multi-GPU / window roll-up): other's clusters are treated as weighted points */
void gyo_td_merge_digest(gyo_tdigest *d, const gyo_tdigest *o);
double gyo_td_quantile(const gyo_tdigest *d, double q);

/* Buffered form the engine keeps per service (the classic merging-digest buffer): up to GYO_TD_PEND_CAP values wait unmerged;
 * a batch that would overflow the buffer re-clusters the digest with (buffered + new) values in ONE merge, and so does a batch
 * after which ANOTHER batch of the same size would take the buffer past GYS_TDIGEST_MERGE_FAST values.  The result depends
 * only on the sequence of batch multisets, not on the order of values inside a batch.  vmin / vmax always cover buffered values. */
#define GYO_TD_PEND_CAP 896
typedef struct {
	gyo_tdigest d;
	uint32_t npend;
	int32_t pend[GYO_TD_PEND_CAP];
	/* a buffer size other than the default (gys_config.td_pend_cap; gyo_tdb_init_cap): cap = its size (0 = GYO_TD_PEND_CAP) and, when it is larger
	 * than the array above, ext = the buffer (owned: gyo_tdb_free) */
	uint32_t cap;
	int32_t *ext;
} gyo_td_buffered;

void gyo_tdb_init(gyo_td_buffered *b);
void gyo_tdb_init_cap(gyo_td_buffered *b, uint32_t cap); /* cap: 64 .. 3968 */
void gyo_tdb_free(gyo_td_buffered *b);
const int32_t *gyo_tdb_values(const gyo_td_buffered *b); /* the npend buffered values */
uint64_t gyo_tdb_total(const gyo_td_buffered *b);       /* merged + buffered */
void gyo_tdb_add_batch(gyo_td_buffered *b, const int32_t *vals, size_t m);
void gyo_tdb_merged_view(const gyo_td_buffered *b, gyo_tdigest *out); /* digest with the buffer merged in; b is not modified */
double gyo_tdb_quantile(const gyo_td_buffered *b, double q);         /* = gyo_td_quantile(merged view) */

/* ---------------------------------------------------------------- roll-up digests (gy_oracle_rollup.c): 64-bit counters */
typedef struct {
	int64_t sum[GYO_TD_NB];
	uint64_t cnt[GYO_TD_NB];
	int64_t vmin, vmax; /* valid when total > 0 */
} gyo_td64;

void gyo_td64_init(gyo_td64 *d);
uint64_t gyo_td64_total(const gyo_td64 *d);
void gyo_td64_merge_values(gyo_td64 *d, const int32_t *vals, size_t m);
void gyo_td64_merge_service(gyo_td64 *d, const gyo_td_buffered *b); /* the service's clusters, then its buffered values */
void gyo_td64_merge_td64(gyo_td64 *d, const gyo_td64 *o);
double gyo_td64_quantile(const gyo_td64 *d, double q);
/* the roll-up of a group (round 6): union by value bin -- members are ADDED in any order, gyo_tdbins_finish makes the group's digest */
#define GYO_TD_BINS 2048
typedef struct {
	uint64_t cnt[GYO_TD_BINS];
	uint64_t sum[GYO_TD_BINS];
	int64_t vmin, vmax;
} gyo_td_bins;
uint32_t gyo_td_value_bin(uint32_t v);
void gyo_tdbins_init(gyo_td_bins *b);
void gyo_tdbins_add_values(gyo_td_bins *b, const int32_t *vals, size_t m);
void gyo_tdbins_add_service(gyo_td_bins *b, const gyo_td_buffered *s); /* its clusters and its buffered values */
void gyo_tdbins_add_td64(gyo_td_bins *b, const gyo_td64 *o);           /* a roll-up digest's clusters */
void gyo_tdbins_finish(const gyo_td_bins *b, gyo_td64 *out);
int gyo_tcp_conn_pair_batch(const uint8_t *batch, int nrec, const uint8_t *pend, uint32_t *pair32, uint64_t *pair64, uint32_t *cpair32, uint64_t *cpair64);
void gyo_active_conn_sketch_batch(const uint8_t *batch, int nrec, uint32_t *pair32 /*[D*W]*/, uint64_t *pair64 /*[D*W]*/, uint64_t out[2]);
void gyo_active_conn_sketch_batch2(const uint8_t *batch, int nrec, uint32_t *pair32, uint64_t *pair64, uint32_t *rpair32, uint64_t *rpair64, uint64_t out[2]);

/* ---------------------------------------------------------------- wire records + roll-ups */
#define GYO_TCP_CONN_NOTIFY_SZ 280
#define GYO_LISTENER_STATE_NOTIFY_SZ 88
#define GYO_NSTATES 6 /* OBJ_STATE_E STATE_IDLE..STATE_DOWN (common/gy_json_field_maps.h:242-250) */

typedef struct {
	int32_t nstates[GYO_NSTATES];
	int32_t tot_qps, tot_act_conn, tot_kb_inbound, tot_kb_outbound, tot_ser_errors, nlisteners, nactive;
} gyo_listen_summ_stats; /* == LISTEN_SUMM_STATS<int> server/gy_msocket.h:840-851 */

typedef struct {
	uint32_t nhosts, ntasks_issue, ntaskissue_hosts, ntasks, nsvc_issue, nsvcissue_hosts, nsvc, total_qps, svc_net_mb,
		ncpu_issue, nmem_issue;
} gyo_cluster_state_one; /* == MS_CLUSTER_STATE::STATE_ONE common/gy_comm_proto.h:3183-3197 */

/* walk a LISTENER_STATE_NOTIFY batch the way partha_listener_state does (gy_mconnhdlr.cc:11175) and fold every record
 * whose curr_state_ <= STATE_DOWN into summ (gy_msocket.h:853-865).  Returns number of records walked; *nerrors counts
 * records with an out of range state (gy_mconnhdlr.cc:11250-11256). */
int gyo_listener_state_rollup(const uint8_t *batch, int nrec, const uint8_t *pend, gyo_listen_summ_stats *summ, int *nerrors);
uint32_t gyo_listener_state_elem_size(const uint8_t *rec);
uint32_t gyo_tcp_conn_elem_size(const uint8_t *rec);
/* L1 validation of one partha -> madhava message (msg = COMM_HEADER, 8-byte aligned): COMM_HEADER::validate, TCP_CONN_NOTIFY::validate,
 * LISTENER_STATE_NOTIFY::validate (common/gy_comm_proto.cc:10-57, :840-881, :955-996).  1 = valid */
int gyo_comm_header_validate(const uint8_t *msg, uint32_t req_magic);
int gyo_tcp_conn_validate(const uint8_t *msg);
int gyo_listener_state_validate(const uint8_t *msg);
/* walk a TCP_CONN_NOTIFY batch (gy_mconnhdlr.cc:9130) producing for each record the PAIR_IP_PORT(nat_cli_, nat_ser_) key words
 * (gy_mconnhdlr.cc:8707), ser_glob_id_, bytes_sent_, bytes_rcvd_; returns number of records walked */
int gyo_tcp_conn_decode(const uint8_t *batch, int nrec, const uint8_t *pend, uint32_t *keywords /*[nrec*10]*/, uint32_t *nwords,
			uint64_t *ser_glob_id, uint64_t *bytes_sent, uint64_t *bytes_rcvd, uint8_t *flags);
int gyo_tcp_conn_sketch_batch(const uint8_t *batch, int nrec, const uint8_t *pend, uint8_t *hll, uint32_t *cms32, uint64_t *cms64);
/* ---- gy_oracle_query.c: the filtered multi-host listener-state query */
typedef struct {
	uint8_t col, comp, group, reserved;
	uint32_t nvalues, set_first, reserved2;
	int64_t value;
} gyo_svc_term; /* = gys_svc_term of include/gysketch.h */
int32_t gyo_svc_col_value(const uint8_t rec[88], int col);
int gyo_svc_term_match(const gyo_svc_term *t, int32_t v, const int64_t *set_values);
int gyo_svc_filter_match(const uint8_t rec[88], const gyo_svc_term *terms, uint32_t nterms, const int64_t *set_values, const uint8_t group_oper[8],
			 int top_oper);
uint32_t gyo_svcstate_scan(const uint8_t *svc_state, uint32_t nsvc, uint32_t epoch, const uint32_t *svc_host, const uint64_t *svc_gid, const uint8_t *host_in,
			   const gyo_svc_term *terms, uint32_t nterms, const int64_t *set_values, const uint8_t group_oper[8], int top_oper, int sort_col,
			   int sort_desc, uint32_t maxrecs, uint32_t *out_slots, uint64_t *nmatched, const uint32_t *slot_list, uint32_t nlist);
void gyo_svcstate_aggr(const uint8_t *svc_state, uint32_t nsvc, uint32_t epoch, const uint32_t *svc_host, const uint64_t *svc_gid, const uint8_t *host_in,
		       const gyo_svc_term *terms, uint32_t nterms, const int64_t *set_values, const uint8_t group_oper[8], int top_oper, int group_by,
		       const uint32_t *host_cluster, const uint8_t *cols, uint32_t ncols, int64_t *acc, uint64_t *count, const uint32_t *slot_list, uint32_t nlist);
int gyo_tcp_conn_walk_tallies(const uint8_t *batch, int nrec, const uint8_t *pend, uint64_t out[4]);
int gyo_tcp_conn_svc_counters(const uint8_t *batch, int nrec, const uint8_t *pend, const uint64_t *gids, uint32_t ngids, uint64_t *ctr, uint64_t *unknown);
void gyo_cluster_state_update(gyo_cluster_state_one *c, uint32_t ntasks_issue, uint32_t ntasks, uint32_t nlisten_issue,
			      uint32_t nlisten, uint32_t cpu_issue, uint32_t mem_issue, const gyo_listen_summ_stats *summ);
void gyo_cluster_state_add(gyo_cluster_state_one *dst, const gyo_cluster_state_one *src);

/* ---------------------------------------------------------------- multi-level time-series histogram (gy_oracle_levels.c; PARITY UNPINNED)
 * folly::BucketedTimeSeries / MultiLevelTimeSeries as TIME_HISTOGRAM<RESP_TIME_HASH, Level_5s_5min_5days_all> drives them
 * (common/gy_statistics.h:1082-1551); times are whole seconds (folly::LegacyStatsClock<std::chrono::seconds>). */
#define GYO_BTS_MAXB 16
#define GYO_MLH_LEVELS 4 /* 5 s, 300 s, 5 days, all-time */

typedef struct {
	int64_t duration; /* seconds; 0 = all-time */
	uint32_t nbuckets;
	int64_t first_time, latest_time;
	int64_t tot_sum;
	uint64_t tot_cnt;
	int64_t bsum[GYO_BTS_MAXB];
	uint64_t bcnt[GYO_BTS_MAXB];
} gyo_bts;

void gyo_bts_init(gyo_bts *s, uint32_t nbuckets, int64_t duration);
int gyo_bts_add(gyo_bts *s, int64_t now, int64_t sum, uint64_t nsamples); /* addValueAggregated */
void gyo_bts_update(gyo_bts *s, int64_t now);

typedef struct {
	int kind, nb;
	gyo_bts s[GYO_MAX_BUCKETS][GYO_MLH_LEVELS];
	int64_t cached_time[GYO_MAX_BUCKETS];
	gyo_hist_serial cached[GYO_MAX_BUCKETS];
} gyo_mlhist;

int64_t gyo_mlh_level_seconds(int level);
void gyo_mlh_init(gyo_mlhist *h, int kind, uint32_t ntimeseries_buckets /* reference default 10 */);
void gyo_mlh_add_hist(gyo_mlhist *h, int64_t tnow, const gyo_hist_serial *stats /*[nb]*/, int flush);
void gyo_mlh_flush(gyo_mlhist *h, int64_t tnow);
void gyo_mlh_level(const gyo_mlhist *h, int level, gyo_hist_serial *out /*[16]*/);
size_t gyo_slab_percentile_idx(const uint64_t *counts, size_t nb, double pct);
void gyo_mlh_get_stats(const gyo_mlhist *h, int level, const float *pcts, size_t npct, int64_t *values, int64_t *tcount, int64_t *tsum,
		       double *mean);
/* TIME_HISTOGRAM::get_stats_for_period (common/gy_statistics.h:1378-1406): [starttime, endtime] in seconds; flush first */
void gyo_bts_range(const gyo_bts *s, int64_t start, int64_t end, uint64_t *pcount, int64_t *psum); /* count(start, end) / sum(start, end) */
int gyo_mlh_level_for_start(const gyo_mlhist *h, int b, int64_t start);
void gyo_mlh_period(const gyo_mlhist *h, int64_t starttime, int64_t endtime, gyo_hist_serial *out /*[16]*/);
void gyo_mlh_get_stats_for_period(const gyo_mlhist *h, int64_t starttime, int64_t endtime, const float *pcts, size_t npct, int64_t *values,
				  int64_t *tcount, int64_t *tsum, double *mean);

/* ---------------------------------------------------------------- per-listener 5-s scan (gy_oracle_lscan.c)
 * the data-parallel part of TCP_SOCK_HANDLER::listener_stats_update + TCP_LISTENER::get_curr_state (common/gy_socket_stat.cc:4044-4365,
 * :2030-2143); same layout as gys_listener_scan (include/gysketch.h) */
typedef struct {
	uint64_t glob_id;
	int64_t tcount[GYO_MLH_LEVELS], tsum[GYO_MLH_LEVELS];
	int32_t p95_ms[GYO_MLH_LEVELS], p99_ms[GYO_MLH_LEVELS], p25_ms[GYO_MLH_LEVELS];
	int32_t last_qps, curr_qps, qps_p95, qps_p25, act_p95, act_p25;
	uint8_t b5, b300, b5day, nconn_active;
	uint8_t nactive_conn_arr[15];
	uint8_t reserved[5];
} gyo_listener_scan;
uint32_t gyo_bucketid_from_threshold(int kind, int64_t threshold);
void gyo_listener_scan_one(const gyo_mlhist *resp, const gyo_hist *qps, const gyo_hist *act, const uint16_t respmap[64], uint64_t glob_id,
			   float multiple, int64_t diffsec, uint8_t notify[88], gyo_listener_scan *out);

/* ---------------------------------------------------------------- the listener's state decision (gy_oracle_lstate.c)
 * TCP_LISTENER::get_curr_state (common/gy_socket_stat.cc:2020-2870) as a function of the scan record and of the inputs that are not the
 * listener's own histograms (task / host status, server errors); same layout as gys_listener_issue_in / gys_listener_decision */
enum { GYO_LI_TASK_ISSUE = 1, GYO_LI_SEVERE = 2, GYO_LI_DELAY = 4, GYO_LI_CPU_ISSUE = 8, GYO_LI_MEM_ISSUE = 16, GYO_LI_DEPENDS = 32, GYO_LI_YOUNG = 64 };
typedef struct {
	uint32_t ser_errors;                                                   /* :2020 argument */
	uint32_t tasks_delay_msec, tasks_cpudelay_msec, tasks_blkiodelay_msec; /* LISTENER_TASK_STATUS::tasks_*_usec_ / 1000 (:2047, :2784-2785) */
	int32_t nconn;                                                         /* last_chk_nconn_ (:2040) */
	uint16_t ntasks_issue, ntasks_noissue;                                 /* is_task_issue outputs (:2043-2044) */
	uint8_t flags;                                                         /* GYO_LI_*: task_issue, is_severe, is_delay, cpu_issue, mem_issue, nserdepends > 0 (:2823-2829), listener younger than 100 s (:4244) */
	uint8_t pad[3];
	int64_t tdiff_start;                                                   /* seconds the response histogram covers (:2033-2034); <= 0: the full 5 days */
} gyo_listener_issue_in;
typedef struct {
	uint8_t state, issue;                       /* OBJ_STATE_E, LISTENER_ISSUE_SRC (common/gy_json_field_maps.h:242-250, :419-434) */
	uint8_t issue_bit_hist, high_resp_bit_hist; /* TCP_LISTENER::issue_bit_hist_ / high_resp_bit_hist_ after the call */
	uint16_t decided_line;                      /* line of common/gy_socket_stat.cc whose return (or the function's end) decided */
	uint16_t pad;
} gyo_listener_decision;
/* get_curr_state itself: *high_resp_bit_hist is the listener's history byte (in / out); returns the deciding line */
int gyo_listener_curr_state(const gyo_listener_scan *sc, const gyo_listener_issue_in *in, uint8_t *high_resp_bit_hist, uint8_t *state, uint8_t *issue);
/* get_curr_state + the caller's part (common/gy_socket_stat.cc:4241-4266: issue_bit_hist_, the "just started" override) */
void gyo_listener_decide(const gyo_listener_scan *sc, const gyo_listener_issue_in *in, uint8_t *issue_bit_hist, uint8_t *high_resp_bit_hist,
			 gyo_listener_decision *out);

/* BOUNDED_PRIO_QUEUE<uint64_t, greater> (common/gy_statistics.h:356-383): returns retained values sorted descending */
size_t gyo_topn_u64(const uint64_t *vals, size_t n, size_t maxn, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif

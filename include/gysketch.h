/*
 * gysketch.h -- C ABI of libgysketch.so, the MI355X-native streaming-sketch aggregation engine that replaces the inside of
 * Gyeeta's madhava/shyama roll-up path.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Each entry point names the reference interface it replaces (file:line under the Gyeeta source tree).  The reference has no
 * FFI layer: its boundary is a set of C++ member functions called from the L2 dispatch switch
 * (server/gy_mconnhdlr.cc:4700-4792); gyeeta_amd/csrc/gys_mconn_shim.hpp gives those same C++ signatures on top of this ABI
 * and INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - All functions return 0 (GYS_OK) or a negative GYS_ERR_* code; nothing throws across the boundary
 *     (reference: bool return + GY_CATCH_EXCEPTION, gy_mconnhdlr.cc:4771-4774).
 *   - "batch"/"pend" pairs follow the reference iteration convention
 *     for (i < n && (uint8_t*)p < pendptr; p += p->get_elem_size())   (gy_mconnhdlr.cc:9130, :11175).
 *     Host buffers are only read during the call (the reference's DB_WRITE_ARR owns them, gy_mconnhdlr.h:350-442).
 *   - *_dev variants take DEVICE pointers to batches already resident in HBM (the measured configuration).
 *   - A context is bound to one GPU and one HIP stream; calls on one context must be serialised by the caller
 *     (one context per L2 thread pool, or an external mutex; the reference serialises per host with connlistenmutex_).
 */
#ifndef GYSKETCH_H
#define GYSKETCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GYS_ABI_VERSION 7

enum {
	GYS_OK = 0,
	GYS_ERR_INVAL = -1,     /* bad argument / malformed batch */
	GYS_ERR_NOMEM = -2,     /* capacity (hosts, services, batch staging) exhausted */
	GYS_ERR_HIP = -3,       /* HIP runtime error (gys_last_error() has the text) */
	GYS_ERR_NOTFOUND = -4,  /* unknown host / service */
	GYS_ERR_NOT_OWNER = -5, /* host is sharded to another rank (jhash2(machine_id) % nranks != rank) */
	GYS_ERR_STATE = -6,     /* call sequence error (e.g. window_finish without window_prepare) */
	GYS_ERR_INTERNAL = -7   /* a C++ exception other than an allocation failure inside the library (never crosses the boundary) */
};

/* bucket-hash kinds == the reference's hash classes, common/gy_statistics.h:1565-2063 */
enum {
	GYS_RESP_TIME_HASH = 0,   /* :1674 */
	GYS_SEMI_LOG_HASH = 1,    /* :1729 */
	GYS_SEMI_LOG_HASH_LO = 2, /* :1782 */
	GYS_DURATION_HASH = 3,    /* :1835 */
	GYS_HASH_10_5000 = 4,     /* :1908 */
	GYS_HASH_5_250 = 5,       /* :1960 */
	GYS_HASH_1_3000 = 6,      /* :2013 */
	GYS_PERCENT_HASH = 7,     /* :1624 */
	GYS_NUM_HASH_KINDS = 8
};

#define GYS_MAX_BUCKETS 16 /* all reference hash classes have <= 15 buckets; records are padded to 16 slots */
#define GYS_TD_NB 200      /* t-digest clusters per key (2 x the delta = 100 the reference hands to Postgres tdigest, common/gy_query_common.cc:1855) */
#define GYS_TD_PEND_CAP 896 /* values a key's t-digest buffers before it is re-clustered, by default (== GYS_TDIGEST_PEND_CAP); gys_config.td_pend_cap */
#define GYS_TD_PEND_CAP_MAX 3968 /* + 128 = 4096: the largest merge the one-workgroup value-bin kernel takes */
#define GYS_HLL_P 14       /* global distinct-flow HLL precision: 16384 u8 registers */
#define GYS_CMS_D 4
#define GYS_CMS_W 65536
#define GYS_NSTATES 6      /* OBJ_STATE_E STATE_IDLE..STATE_DOWN, common/gy_json_field_maps.h:242-250 */
#define GYS_TOPN 10        /* per-host top-N, server/gy_mconnhdlr.h:961 */

typedef struct gys_ctx gys_ctx;

typedef struct {
	uint32_t struct_size;      /* = sizeof(gys_config) */
	int32_t device;            /* HIP device ordinal, -1 = current device */
	uint32_t rank, nranks;     /* shard of the host-id space owned by this context; 0,1 for single GPU */
	uint32_t max_hosts;        /* partha capacity (reference: MAX_PARTHA_PER_MADHAVA = 512 per madhava) */
	uint32_t max_services;     /* listener capacity across all hosts */
	uint32_t max_clusters;     /* cluster-name capacity (MS_CLUSTER_STATE::MAX_NUM_CLUSTERS = 512) */
	uint32_t enable_tdigest;   /* per-service t-digest of response times */
	uint32_t svc_hll_p;        /* per-service distinct-client HLL precision (0 = off, 4..10) */
	uint32_t resp_path;        /* 0 = choose per batch; 1 = always the general (global table + atomics) pipeline; 2 = prefer the
	                              host-local pipeline (LDS sub-table per host segment) whenever the batch qualifies; 3 = as 2, and
	                              segments longer than 65536 events are cut into parts (split form: few hosts, long segments) */
	uint64_t max_batch_events; /* largest resp-event batch one ingest call may carry (t-digest staging capacity) */
	void *stream;              /* hipStream_t to run on; NULL = the context creates its own */
	void *reduce_arena;        /* optional caller-owned DEVICE buffer for the all-reducible registers (e.g. a torch tensor so */
	uint64_t reduce_arena_bytes; /* that torch.distributed/RCCL can reduce it in place); NULL = the context allocates it  */
	uint32_t enable_levels;    /* multi-level windows + per-service QPS / active-connection histograms (see "multi-level windows" below):
	                              1 = all four levels (5 s / 300 s / 5 days / all; 21 hist records = 5.4 KB of HBM per service; every window
	                                  close brings the records of every service touched in the window up to date -- what the reference's 5-s
	                                  flush does per listener);
	                              2 = without the 5-s level (300 s / 5 days / all): a close touches the services only when it crosses a ring
	                                  boundary (every 30 s); level 0 and gys_scan_listener_state_dev answer GYS_ERR_STATE, a period that
	                                  folly would answer from the 5-s ring is answered from the 300-s ring */
	uint32_t td_buf_values;    /* entries of a service's value buffer (td_pend_cap + 64 .. 16384; 0 = sized to max_services): the values
	                              waiting for the next t-digest merge plus room for one batch's values of the service */
	uint32_t conn_pair_cms;    /* TCP_CONN_NOTIFY records also feed a Count-Min pair OF THEIR OWN keyed by (ser_glob_id_, cli_task_aggr_id_): connections
	                              (u32 table) and bytes (u64 table) per (listener, client task group) -- the roll-up MCONN_HANDLER keeps in
	                              connlistenmap_ / connclientmap_ (server/gy_msocket.h:240-290, filled by add_tcp_conn_cli / _ser,
	                              server/gy_mconnhdlr.cc:8643-9050).  8 more device atomics per record; off by default */
	uint32_t td_pend_cap;      /* values a service's t-digest buffers before it is re-clustered: 0 = GYS_TD_PEND_CAP (896), else 64 .. GYS_TD_PEND_CAP_MAX.
	                              A batch's values of a service are appended while buffered + new <= td_pend_cap, otherwise ONE merge re-clusters
	                              the digest with all of them (and a service whose next batch of the same size would pass the fast merge size -- the
	                              smallest of 1024 / 2048 / 4096 values that is >= td_pend_cap + 128 -- is merged at once).  A merge costs almost the same whatever it carries (its work is per cluster and per value bin), so a
	                              larger buffer means proportionally fewer merges per window: 3968 at 54 values per service and window = a merge
	                              every ~70 windows instead of every ~17, for 4 x 3968 bytes more HBM per service.  The digest state is a function
	                              of the per-call value multisets AND of this number (the CPU oracle takes the same parameter). */
} gys_config;

/* -------------------------------------------------------------------------------------------------------------------
 * lifecycle */
uint32_t gys_abi_version(void);
const char *gys_last_error(void);
uint64_t gys_reduce_arena_bytes(const gys_config *cfg); /* size the caller must provide in cfg->reduce_arena */
int gys_create(const gys_config *cfg, gys_ctx **out);
void gys_destroy(gys_ctx *ctx);
int gys_sync(gys_ctx *ctx); /* hipStreamSynchronize on the context stream */

/* host-id shard function: GY_MACHINE_ID::get_hash() % nshards  (common/gy_sys_hardware.h:82-85) */
uint32_t gys_machine_id_hash(const uint8_t machine_id[16]);
uint32_t gys_shard_of(const uint8_t machine_id[16], uint32_t nshards);

/* -------------------------------------------------------------------------------------------------------------------
 * registration (control-plane facts the data path needs; the reference learns them from PM_CONNECT / NOTIFY_NEW_LISTENER,
 * server/gy_mconnhdlr.cc partha registration + handle_add_listener -- out of scope beyond these two calls) */
/* Cluster indices are assigned in first-seen order per context; in a multi-rank job call gys_register_cluster for every cluster
 * name in the SAME order on every rank before registering hosts, so that the all-reduced cluster rows line up. */
int gys_register_cluster(gys_ctx *ctx, const char *cluster_name, uint32_t *cluster_idx);
int gys_register_host(gys_ctx *ctx, const uint8_t machine_id[16], const char *cluster_name, uint32_t *host_slot);

typedef struct {
	uint64_t glob_id;   /* TCP_LISTENER::glob_id_ (opaque 64-bit id on the wire) */
	uint32_t netns;     /* network namespace inode as carried by the eBPF tuple (partha/gy_ebpf_kernel_struct.h:28-35) */
	uint16_t port;      /* listener port, host order */
	uint8_t is_any_ip;  /* comm::NEW_LISTENER::is_any_ip_ (common/gy_comm_proto.h:1539; TCP_LISTENER::is_any_ip_ = addr.is_any_address(),
	                       common/gy_socket_stat.cc:1796): non-zero = the listener takes the events of every address of its (netns, port) */
	uint8_t addr_is_v6; /* is_any_ip == 0: addr is an in6_addr (16 bytes) when non-zero, else an in_addr in addr[0..3] (network order) */
	char comm[16];      /* TASK_COMM_LEN process name (LISTEN_TOPN::comm_) */
	uint8_t addr[16];   /* is_any_ip == 0: the address the listener is bound to -- NEW_LISTENER::ns_ip_port_.ip_port_.ipaddr_.  A response event
	                       reaches the listener only when its server address equals this one the way GY_IP_ADDR::operator== compares
	                       (common/gy_common_inc.h:10629-10636: an IPv6 address that embeds an IPv4 one -- ::ffff:a.b.c.d, 2002::/16,
	                       64:ff9b::/32 -- equals that IPv4 address); ignored when is_any_ip != 0 */
} gys_listener_info;

/* Listener lookup of a response event (replaces listener_tbl_.lookup_single_elem(ser_nsipport, hash ignoring the IP) with the comparator
 * operator==(shared_ptr<TCP_LISTENER>, NS_IP_PORT), common/gy_socket_stat.cc:1671, common/gy_socket_stat.h:708-714): among the listeners
 * registered for the host with the event's (netns, server port), in registration order, the first one that is_any_ip or is bound to the
 * event's server address takes the event; if none does the event is dropped (gys_counters.resp_dropped_nolistener).  Registration follows
 * insert_or_replace (common/gy_socket_stat.cc:1372, :7779): a new listener REPLACES, in place, the first registered listener of its
 * (netns, port) that is_any_ip or is bound to the same address -- that listener's slot keeps its state but gets no further response events
 * -- and is appended behind the others otherwise. */

/* assigns consecutive service slots [*first_slot, *first_slot + n) to the listeners the engine does not know yet.  A glob_id that is
 * already registered (a partha resends its listeners after a reconnect) keeps its slot and all of its state, and repeats inside one call
 * are dropped: with such entries fewer than n slots are assigned (gys_lookup_service gives any listener's slot). */
int gys_register_listeners(gys_ctx *ctx, const uint8_t machine_id[16], const gys_listener_info *arr, uint32_t n, uint32_t *first_slot);

/* -------------------------------------------------------------------------------------------------------------------
 * ingest
 *
 * Host-pointer entry points (gys_ingest_resp_events, gys_ingest_tcp_conn, gys_ingest_listener_state): the caller's buffer is only
 * read DURING the call (the reference's pone points into the L1 receive buffer, freed after the dispatch switch) and
 * the call does NOT wait for the GPU: the records are copied into a slot of a ring of 16 pinned staging buffers (handed out oldest
 * first), one H2D copy and the ingest kernels are enqueued, and the call returns; a slot is reused once the event recorded behind its
 * kernels has fired (gys_counters.stage_waits counts the calls that had to wait for that) -- this is the path of
 * gys_ingest_active_conns and of a connection / listener-state call of many messages' size (> 8 MiB / 2 MiB of records).  A partha's
 * TCP_CONN_NOTIFY and LISTENER_STATE_NOTIFY messages (gys_ingest_tcp_conn, gys_ingest_listener_state) go through SUBMISSION QUEUES of
 * their own: the messages of all threads are appended to one pinned batch (records + an offset and the sender's host slot per record)
 * that is copied and launched once; records keep the order in which the calls arrived, so the last state record of a listener wins
 * across the messages of a batch as it does across calls (gys_counters.conn_calls_queued / conn_submissions, lstate_*).
 * gys_ingest_resp_events goes through a SUBMISSION QUEUE as well: the calls of all threads are concatenated into one pinned batch (a segment per call) and handed to the
 * response pipeline together -- with an idle GPU a call is submitted at once; while two submissions are still executing, further calls
 * accumulate and go out together as soon as one of them has finished (group commit: the fixed launches of a response batch are paid
 * once per submission exactly when the GPU is the bottleneck; a host appears at most once per combined batch and batches are submitted in the order they were
 * sealed, so every service sees its per-call value multisets in call order and the state is bit-identical to per-call ingestion).
 * A call may therefore return before its events are on the stream; every other entry point puts the queue's content on the stream first.
 * These three calls may be made concurrently from several threads (up to the reference's MAX_L2_MISC_THREADS = 16,
 * server/gy_mconnhdlr.h:60) on the same context.  Everything else -- registration, the _dev entry points, the wire front end, the
 * window boundary, queries and exports -- must not run concurrently with any other call on the context (gys_mconn_shim.hpp holds a
 * shared / exclusive lock accordingly).  Results become visible to queries in stream order; gys_sync waits for them. */

/* Raw response events in the eBPF layout tcp_ipv4_resp_event_t (common/gy_ebpf_kernel.h:106-111, 24 bytes: saddr,daddr,netns,
 * sport,dport (network order), lsndtime, lrcvtime).  Replaces TCP_SOCK_HANDLER::handle_ipv4_resp_event + handle_tcp_resp_event
 * (common/gy_socket_stat.cc:1517-1677): per event RESP_TIME_HASH bucket + histogram add + CONN_BITMAP + query count, plus the new
 * t-digest / HLL / CMS sketches. */
int gys_ingest_resp_events(gys_ctx *ctx, const uint8_t machine_id[16], const void *ev24, uint32_t nevents);

typedef struct {
	uint32_t host_slot;   /* from gys_register_host */
	uint32_t reserved;    /* ignored on input (the engine clears its copy; it uses the field for its own per-part segment lists) */
	uint64_t first_event; /* index of this host's first event; segments sorted ascending, host's events contiguous */
} gys_resp_seg;

/* device-resident multi-host batch: d_ev24 is a DEVICE pointer to nevents 24-byte events; segs is a HOST array */
int gys_ingest_resp_events_dev(gys_ctx *ctx, const gys_resp_seg *segs, uint32_t nsegs, const void *d_ev24, uint64_t nevents);

/* IPv6 response events in the eBPF layout tcp_ipv6_resp_event_t (common/gy_ebpf_kernel.h:113-118, 48 bytes: ipv6_tuple_t
 * {u128 saddr, u128 daddr, u32 netns, u16 sport, u16 dport (network order)} partha/gy_ebpf_kernel_struct.h:37-44, then lsndtime, lrcvtime).
 * Replaces TCP_SOCK_HANDLER::handle_ipv6_resp_event (common/gy_socket_stat.cc:1535-1551) -> handle_tcp_resp_event(is_ipv4 = false): the same
 * filter, listener lookup and RESP_TIME_HASH bucket as the IPv4 form; the histogram and the query count are the listener's shared ones
 * (resp_cache_v6_ flushes into the same resp_hist_, :1800; curr_query_v4_ + curr_query_v6_, :4050-4051), the CONN_BITMAP rows are the
 * listener's resp_bitmap_v6_ (:1587; rows 32..63 of gys_export_conn_bitmap, added to the IPv4 rows' counts per bucket as :4144-4149 does).
 * Flow key: PAIR_IP_PORT(daddr:dport, saddr:sport) with both addresses through GY_IP_ADDR(unsigned __int128) -- an address that embeds an
 * IPv4 one hashes and compares as that IPv4 address (common/gy_common_inc.h:11040-11129, :10950-10959).  The reference has one perf buffer and
 * one handler thread per family (PROBE_TCP_RESPONSE_IPv4 / _IPv6, common/gy_socket_stat.cc:101-102): a stream is handed over as calls of
 * one family each.  The host-pointer form is its own submission (it does not join the IPv4 submission queue, it runs behind it). */
int gys_ingest_resp_events_v6(gys_ctx *ctx, const uint8_t machine_id[16], const void *ev48, uint32_t nevents);
int gys_ingest_resp_events_v6_dev(gys_ctx *ctx, const gys_resp_seg *segs, uint32_t nsegs, const void *d_ev48, uint64_t nevents);

/* Replaces MCONN_HANDLER::partha_tcp_conn_info(partha, TCP_CONN_NOTIFY *pone, int nconns, uint8_t *pendptr, ...)
 * (server/gy_mconnhdlr.h:2091, .cc:9052-9444): flow key PAIR_IP_PORT(nat_cli_, nat_ser_) (.cc:8707) -> distinct-flow HLL;
 * per-service connection / byte counters (connlistenmap_ roll-up .cc:9133-9319) -> exact per-service counters + CMS.
 * The batch is walked like the reference's L2 loop `for (i < nconns && p < pendptr; p += get_elem_size())` (.cc:9130, :11175) under the
 * L1 validators' rule (TCP_CONN_NOTIFY::validate / LISTENER_STATE_NOTIFY::validate, common/gy_comm_proto.cc:859-880, :974-995: the L2
 * loop only ever sees messages that passed it): every one of the nconns announced records lies complete before `pend` and has a size
 * that is a multiple of 8; otherwise GYS_ERR_INVAL and nothing of the batch is ingested.  `pend` must not be NULL.  (The wire front end
 * gys_ingest_comm_stream applies the same rule on the GPU.) */
int gys_ingest_tcp_conn(gys_ctx *ctx, const uint8_t machine_id[16], const void *batch, uint32_t nconns, const void *pend);
/* device-resident: d_offsets[i] = byte offset of record i inside d_batch (records are variable stride: get_elem_size()) */
int gys_ingest_tcp_conn_dev(gys_ctx *ctx, const void *d_batch, const uint32_t *d_offsets, uint32_t nconns);

/* Replaces MCONN_HANDLER::partha_listener_state(partha, const LISTENER_STATE_NOTIFY *pone, int nconns, uint8_t *pendptr, ...)
 * (server/gy_mconnhdlr.h:2129, .cc:10993-11412): per record glob_id probe, LISTEN_SUMM_STATS::update, set_state, top-N. */
int gys_ingest_listener_state(gys_ctx *ctx, const uint8_t machine_id[16], const void *batch, uint32_t nrecs, const void *pend);

/* Wire front-end: a byte stream of COMM_HEADER-framed messages exactly as an unmodified partha sends them on its PM_HDR_MAGIC
 * connection ([COMM_HEADER 16 B][EVENT_NOTIFY 8 B][nevents_ records][padding], common/gy_comm_proto.h:336-420, :486-500).
 * Replaces the L1 validation + L2 dispatch in front of the two handlers above (COMM_HEADER::validate common/gy_comm_proto.cc:10-57,
 * TCP_CONN_NOTIFY::validate :840-881, LISTENER_STATE_NOTIFY::validate :955-996, dispatch server/gy_mconnhdlr.cc:4756-4792): message
 * headers are checked on the host, the variable-stride record chains are walked, checked and indexed ON THE GPU (pointer doubling +
 * prefix sum, gys_kernels.hpp "wire front-end"), EVENT_NOTIFY subtypes NOTIFY_TCP_CONN / NOTIFY_LISTENER_STATE are ingested, every
 * other message is skipped (control plane).  A malformed header or record rejects the call (the reference drops the connection);
 * nothing of the rejected call is ingested for the kind that failed.  buf must be 8-byte aligned. */
typedef struct {
	uint32_t nmsgs, nmsgs_tcp_conn, nmsgs_listener_state, nmsgs_skipped, nmsgs_invalid, reserved;
	uint64_t nrecords;       /* TCP_CONN_NOTIFY + LISTENER_STATE_NOTIFY records ingested */
	uint64_t bytes_consumed; /* whole messages consumed; a trailing partial message is left to the caller */
} gys_comm_stats;
int gys_ingest_comm_stream(gys_ctx *ctx, const uint8_t machine_id[16], const void *buf, uint64_t nbytes, gys_comm_stats *out);
/* device-resident multi-host: d_host_slot[i] = host of record i */
int gys_ingest_listener_state_dev(gys_ctx *ctx, const void *d_batch, const uint32_t *d_offsets, const uint32_t *d_host_slot,
				  uint32_t nrecs);

/* Replaces MCONN_HANDLER::handle_partha_active_conns(partha, const comm::ACTIVE_CONN_STATS *pconn, int nitems, uint8_t *pendptr, ...)
 * (server/gy_mconnhdlr.h:2039, .cc:7705-7772; rows comm::ACTIVE_CONN_STATS common/gy_comm_proto.h:2766-2783, 104-byte fixed stride, a
 * partha's 15-s report of (listener, client task group) rows).  The reference formats every row into an SQL insert
 * (insert_active_conns .cc:7776-7960: is_remote_listen_ == false -> activeconntbl, true -> remoteconntbl).  Here the local-listener
 * rows update (i) a Count-Min pair keyed by (listener_glob_id_, cli_aggr_task_id_): active_conns_ (u32 table) and bytes_sent_ +
 * bytes_received_ (u64 table) -- the per-(listener, client task) roll-up that stands for those rows; both tables live in the reduce
 * arena (all-reduced with the other registers, per window; queries read the per-cell maximum of the last three windows' tables, see
 * gys_query_pair_cms) -- and (ii) exact cumulative per-listener sums.  Remote-listener rows (is_remote_listen_: the listener lives on another
 * madhava, server/gy_mconnhdlr.cc:7888-7925 -> remoteconntbl) are rolled up the same way into a Count-Min pair of their own (gys_query_pair_cms
 * which 6 / 7) and counted (gys_counters.actconn_remote_listen); they name no service of this engine, so there are no per-listener sums for them. */
int gys_ingest_active_conns(gys_ctx *ctx, const uint8_t machine_id[16], const void *batch, uint32_t nitems, const void *pend);
int gys_ingest_active_conns_dev(gys_ctx *ctx, const void *d_batch, uint32_t nitems);

typedef struct {
	uint32_t ntasks_issue, ntasks, nlisten_issue, nlisten; /* comm::HOST_STATE_NOTIFY fields used by update_from_state */
	uint8_t cpu_issue, mem_issue, curr_state, reserved;
} gys_host_state;
/* Replaces the host_state_ store read by MCONN_HANDLER::send_cluster_state (server/gy_mconnhdlr.cc:16052-16075) */
int gys_ingest_host_state(gys_ctx *ctx, const uint8_t machine_id[16], const gys_host_state *st);

/* -------------------------------------------------------------------------------------------------------------------
 * window boundary (the reference's 5 s cadence: send_cluster_state -> handle_cluster_state -> aggregate_cluster_state,
 * server/gy_mconnhdlr.cc:16052, server/gy_shconnhdlr.cc:4554-4640)
 *
 *   gys_window_prepare : local roll-ups are written into the reduce arena (cluster STATE_ONE sums, global histogram, HLL, CMS)
 *   <caller all-reduces every gys_reduce_section across ranks: RCCL over xGMI, or nothing for a single GPU>
 *   gys_window_finish  : consumes the reduced arena (global answers), folds window histograms into the all-time ones,
 *                        clears window state.
 */
typedef struct {
	void *dev_ptr;
	uint64_t nelems;
	uint32_t dtype; /* 0 = u8, 1 = u32, 2 = i64 */
	uint32_t op;    /* 0 = MAX, 1 = SUM */
} gys_reduce_section;

int gys_reduce_sections(gys_ctx *ctx, gys_reduce_section out[4], uint32_t *nsections);
int gys_window_prepare(gys_ctx *ctx, uint64_t tusec);
int gys_window_finish(gys_ctx *ctx);
/* single rank: gys_window_prepare + gys_window_finish as ONE captured hipGraph (the fold kernels that are due, k_window_prepare, the
 * copy / clear sequence, the window-number increment), captured once per registry shape and replayed with one launch per window
 * (gys_counters.window_graph_launches).  Falls back to the two calls with multi-level windows or nranks > 1 (gys_window_close_rccl). */
int gys_window_close(gys_ctx *ctx, uint64_t tusec);

/* -------------------------------------------------------------------------------------------------------------------
 * queries (result shapes: common/gy_json_field_maps.h svcsumm :1396-1416, clusterstate :2162-2180, svcstate :1102-1135) */

typedef struct {
	int32_t nstates[GYS_NSTATES];
	int32_t tot_qps, tot_act_conn, tot_kb_inbound, tot_kb_outbound, tot_ser_errors, nlisteners, nactive;
} gys_svcsumm; /* == LISTEN_SUMM_STATS<int>, server/gy_msocket.h:840-851 */

typedef struct {
	uint32_t nhosts, ntasks_issue, ntaskissue_hosts, ntasks, nsvc_issue, nsvcissue_hosts, nsvc, total_qps, svc_net_mb,
		ncpu_issue, nmem_issue;
} gys_cluster_state; /* == comm::MS_CLUSTER_STATE::STATE_ONE, common/gy_comm_proto.h:3183-3197 */

typedef struct {
	int64_t data_value; /* bucket ceiling, get_bucket_max_threshold (common/gy_statistics.h:500-515) */
	int64_t sum;
	uint64_t count;
	float percentile;
	uint32_t reserved;
} gys_hist_data; /* == HIST_DATA, common/gy_statistics.h:473-484 */

typedef struct {
	uint64_t count;
	int64_t sum;
} gys_hist_serial; /* == HIST_SERIAL, common/gy_statistics.h:458-468 */

typedef struct {
	gys_hist_serial stats[GYS_MAX_BUCKETS - 1]; /* 15 buckets */
	uint64_t total_count;
	int64_t max_val_seen;
} gys_hist_rec; /* 256 bytes: the arithmetic state of GY_HISTOGRAM<int64_t,RESP_TIME_HASH> (280 B incl. clocks) */

int gys_query_svcsumm(gys_ctx *ctx, const uint8_t machine_id[16], gys_svcsumm *out);      /* web_curr_listener_summ, gy_mnodehandle.cc:1628 */
int gys_query_clusterstate(gys_ctx *ctx, const char *cluster_name, gys_cluster_state *out); /* aggregate_cluster_state result */
/* GY_HISTOGRAM::get_percentiles (common/gy_statistics.h:707-791) of one service; which: 0 = current (open) window, 1 = all-time.
 * "All-time" is every response ingested so far INCLUDING the open window, in both record modes (t-digest on: lazily folded records;
 * off: per-event records, window added at the boundary) -- the same answer mid-window whichever mode runs (tests/test_gpu_resp.py).
 * The same `which` applies to gys_scan_percentiles_dev and gys_export_hist. */
int gys_query_hist_percentiles(gys_ctx *ctx, uint64_t glob_id, int which, gys_hist_data *pdata, uint32_t npct, uint64_t *total_count,
			       int64_t *max_val, float *pavg);
/* t-digest quantiles (q in [0,1]) of one service's response times: computed from the merged view (clusters re-clustered with the
 * values the key still buffers; the stored state is not modified), rounded half-up to whole milliseconds */
int gys_query_quantiles(gys_ctx *ctx, uint64_t glob_id, const double *q, uint32_t nq, double *out);
/* distinct flows seen (global HLL over PAIR_IP_PORT keys; after gys_window_finish: the all-rank estimate) */
int gys_query_distinct_flows(gys_ctx *ctx, double *out);
/* Count-Min estimate for a service key in the last finished window: which = 0 response events + NEW CONNECTIONS of the service (one per
 * connection, accepting side: see gys_export_svc_counters), which = 1 connection bytes (sent + received) */
int gys_query_cms(gys_ctx *ctx, uint64_t glob_id, int which, uint64_t *out);
/* Count-Min estimate for a (listener, client task group) pair.  Two roll-ups, each with its own table pair:
 *   which 0 / 1: active connections / bytes (sent + received) of the ACTIVE_CONN_STATS rows (gys_ingest_active_conns).  A partha reports
 *                these every 15 s on a phase of its own and a window is 5 s: the answer is the PER-CELL MAXIMUM over the (all-rank) tables
 *                of the last three finished windows -- a pair reported anywhere in the last 15 s reads back at least its reported value
 *                (Count-Min never under-estimates), a partha whose report fell into two of the three windows is not counted twice, and a
 *                pair not reported for three windows reads 0 again (a gauge's "last report").
 *   which 2 / 3: closed connections / their bytes of the TCP_CONN_NOTIFY roll-up, LISTENER side = the reference's connlistenmap_
 *                (server/gy_mconnhdlr.cc:9226-9245: records with tusec_close_, bytes > 0, ser_glob_id_ and is_tcp_accept_event_), last
 *                finished window (gys_config.conn_pair_cms; GYS_ERR_STATE when off);
 *   which 4 / 5: the same for the CLIENT side = connclientmap_ (:9290-9312: closed, bytes > 0, connect-only, cli_task_aggr_id_ != 0). */
int gys_query_pair_cms(gys_ctx *ctx, uint64_t listener_glob_id, uint64_t cli_aggr_task_id, int which, uint64_t *out);

typedef struct {
	uint64_t glob_id;
	uint32_t host_slot;
	uint32_t metric; /* the ranked value */
	uint8_t state[88]; /* the LISTENER_STATE_NOTIFY record (common/gy_comm_proto.h:2183-2254) as last ingested */
} gys_topn_entry;
/* per-host top-N of the last window by kind: 0 issue, 1 qps, 2 active conns, 3 net (LISTEN_TOPN comparators gy_msocket.h:720-796) */
int gys_query_topn(gys_ctx *ctx, const uint8_t machine_id[16], int kind, gys_topn_entry out[GYS_TOPN], uint32_t *nout);

/* The "per-key scan" (TCP_SOCK_HANDLER::listener_stats_update percentile part, common/gy_socket_stat.cc:4226-4230) over ALL services
 * on the GPU: for service slot s and percentile i, d_out[s*npct + i] = bucket ceiling; which as above.  d_out is a DEVICE pointer. */
int gys_scan_percentiles_dev(gys_ctx *ctx, int which, const float *pcts, uint32_t npct, int64_t *d_out);
/* the same scan on the t-digests: quantile q[i] (0..1, nq <= 16) of EVERY service's merged view (clusters re-clustered with the buffered
 * values; no state is modified), d_out[slot * nq + i] on the device -- identical, value for value, to gys_query_quantiles of that service.
 * Replaces the per-listener p25 / p95 / p99 search of TCP_SOCK_HANDLER::listener_stats_update (common/gy_socket_stat.cc:4044-4365,
 * percentile calls :4226-4230) for all listeners at once.  q is a HOST array, d_out a DEVICE pointer to nsvc * nq doubles. */
int gys_scan_quantiles_dev(gys_ctx *ctx, const double *q, uint32_t nq, double *d_out);

/* -------------------------------------------------------------------------------------------------------------------
 * Roll-up digests: the response-time digest of a GROUP of services -- a host, a cluster, all hosts of this rank ("global") -- and the
 * merge of such digests across ranks.  Replaces the aggregated percentile Postgres computes over a set of listeners' rows,
 * public.tdigest_percentile(col, 100, p) (common/gy_query_common.cc:1818-1855), and feeds the fan-in of
 * SHCONN_HANDLER::aggregate_cluster_state (server/gy_shconnhdlr.cc:4583-4720).  Definition (oracle/gy_oracle_rollup.c, gyo_tdbins_*; round 6:
 * no longer an ordered fold): the UNION BY VALUE BIN -- every member's non-empty clusters go, whole, into the value bin (one per millisecond
 * below 1024, 64 cells per octave above) of the integer threshold of their mean, a service's buffered values into the bin of the value; the
 * bins' exact 64-bit {sum, count} are then laid on the rank axis in order and cut into the 200 clusters by the engine's cluster rule, a bin
 * that spans a cluster boundary sharing its sum in proportion (exact integers).  The result does not depend on the order of the members.  A
 * roll-up digest as a member contributes its clusters the same way.  Fixed-size slab = the unit a multi-rank job all-gathers (ncclAllGather
 * of sizeof(gys_tdigest_slab) bytes per rank) and rolls up with gys_tdigest_merge_slabs_dev. */
typedef struct {
	int64_t sum[GYS_TD_NB];
	uint64_t cnt[GYS_TD_NB];
	int64_t vmin, vmax; /* valid when any cnt != 0 */
} gys_tdigest_slab;
enum { GYS_ROLLUP_HOST = 0, GYS_ROLLUP_CLUSTER = 1, GYS_ROLLUP_GLOBAL = 2 };
/* d_out (DEVICE): HOST: one slab per host slot [gys_num_hosts] -- the roll-up of the host's services; CLUSTER: one per registered cluster
 * index -- the roll-up of its hosts' slabs; GLOBAL: one slab -- the roll-up of all host slabs of this rank.  No engine state is modified (the
 * hosts' member lists are kept on the device between calls and rebuilt after a registration). */
int gys_tdigest_rollup_dev(gys_ctx *ctx, int scope, gys_tdigest_slab *d_out);
/* d_out[0] (DEVICE) = roll-up of the slabs d_in[0..n) (the cross-rank merge after an all-gather; also any caller-defined group) */
int gys_tdigest_merge_slabs_dev(gys_ctx *ctx, const gys_tdigest_slab *d_in, uint32_t n, gys_tdigest_slab *d_out);
/* quantiles q[i] (0..1) of one DEVICE slab into the HOST array out (same interpolation and rounding as gys_query_quantiles) */
int gys_tdigest_slab_quantiles(gys_ctx *ctx, const gys_tdigest_slab *d_slab, const double *q, uint32_t nq, double *out);
uint32_t gys_num_clusters(gys_ctx *ctx);

/* -------------------------------------------------------------------------------------------------------------------
 * The window exchange inside the library (RCCL over xGMI; no torch, no caller-written collective).  Replaces
 * MCONN_HANDLER::send_cluster_state -> SHCONN_HANDLER::aggregate_cluster_state (server/gy_mconnhdlr.cc:16052-16118,
 * server/gy_shconnhdlr.cc:4583-4720): one process per GPU, every rank calls gys_window_close_rccl at the 5-s boundary.
 * The communicator handle is an ncclComm_t carried as void* (no RCCL type in this header); a caller that already owns one
 * (e.g. from its own ncclCommInitRank) may pass it directly.  One node: gys_rccl_unique_id / gys_rccl_comm_create set
 * NCCL_SOCKET_IFNAME=lo unless the environment already names an interface (the bootstrap rendezvous then runs over loopback). */
#define GYS_RCCL_UID_BYTES 128
int gys_rccl_unique_id(uint8_t uid[GYS_RCCL_UID_BYTES]);                        /* rank 0: ncclGetUniqueId; hand the bytes to every rank */
int gys_rccl_comm_create(gys_ctx *ctx, const uint8_t uid[GYS_RCCL_UID_BYTES], int nranks, int rank, void **comm); /* ncclCommInitRank on ctx's device */
int gys_rccl_comm_destroy(void *comm);
/* gys_window_prepare, then the four register families of gys_reduce_sections all-reduced in place (u8 MAX, u32 SUM, i64 SUM, i64 MAX)
 * as ONE ncclGroup on the context stream, then gys_window_finish.  Asynchronous like every other call (gys_sync to wait). */
int gys_window_close_rccl(gys_ctx *ctx, void *comm, uint64_t tusec);
/* the global response-time digest across ranks: this rank's GYS_ROLLUP_GLOBAL slab, ncclAllGather of the fixed-size slabs, their
 * roll-up (gys_tdigest_merge_slabs_dev) into d_out[0] (DEVICE) -- the same slab on every rank */
int gys_tdigest_global_rccl(gys_ctx *ctx, void *comm, gys_tdigest_slab *d_out);

/* -------------------------------------------------------------------------------------------------------------------
 * a service's response-time t-digest in the external forms of the Postgres tdigest type, so that the reference's SQL
 * percentile aggregation -- public.tdigest(col, 100) / public.tdigest_percentile(digest, p), common/gy_query_common.cc:1818-1855,
 * extension loaded at :3387 -- can consume engine digests ('<text>'::public.tdigest, or the binary send/recv form).
 *   text:   "flags 1 count N compression 100 centroids K (mean, count) ..."     NUL terminated; *needed = strlen
 *   binary: int32 flags, int64 count, int32 compression, int32 ncentroids, K x {float8 mean, int64 count}, network byte order
 * The digest handed over is the merged view (clusters + still buffered values).  GYS_ERR_NOMEM when buflen is too small (*needed
 * is set), GYS_ERR_NOTFOUND for a service without values (the type has no empty literal). */
int gys_tdigest_sql_text(gys_ctx *ctx, uint64_t glob_id, char *buf, size_t buflen, size_t *needed);
int gys_tdigest_sql_binary(gys_ctx *ctx, uint64_t glob_id, void *buf, size_t buflen, size_t *needed);

/* -------------------------------------------------------------------------------------------------------------------
 * multi-level windows (gys_config.enable_levels).  Replaces the per-listener RESP_TIME_HISTOGRAM =
 * TIME_HISTOGRAM<RESP_TIME_HASH, Level_5s_5min_5days_all> (common/gy_statistics.h:1082-1551, :2067), i.e. folly::MultiLevelTimeSeries
 * per histogram bucket with 10 ring buckets per level, and the QPS_HISTOGRAM / ACTIVE_CONN_HISTOGRAM behind LISTENER_DAY_STATS
 * (common/gy_socket_stat.h:548-549).  Every gys_window_prepare(tusec) is one add_histogram_data(tnow = tusec / 10^6, window
 * histogram, flush) of every service (gy_statistics.h:1213-1247); a query at tusec first advances the series to that time
 * (get_stats_with_flush :1369).  Times are whole seconds and must not go backwards (folly clamps, so does the engine).
 * level: 0 = the window closed last ("last 5 seconds": the engine's tumbling window itself; empty once 5 s have passed since its
 * close), 1 = last 300 s, 2 = last 5 days (rings of GYS_LEVEL_RING buckets, so between 9/10 and 10/10 of the nominal span, exactly
 * as folly's rings), 3 = since start.  Only windows closed by gys_window_prepare are in a level, never the open one. */
#define GYS_NLEVELS 4
#define GYS_LEVEL_RING 10 /* ntimeseries_buckets, common/gy_statistics.h:1104 */
typedef struct {
	int64_t data_value;
	float percentile; /* 0..100 */
	uint32_t pad;
} gys_time_hist_val; /* == TIME_HIST_VAL, common/gy_statistics.h:489-498 */
/* TIME_HISTOGRAM::get_stats_with_flush(level, pstats, nstats, tcount, tsum, mean_val, tnow) :1333-1374 */
int gys_query_hist_level_stats(gys_ctx *ctx, uint64_t glob_id, int level, uint64_t tusec, gys_time_hist_val *pstats, uint32_t nstats,
			       int64_t *tcount, int64_t *tsum, double *mean_val);
/* TIME_HISTOGRAM::get_level_data :1166-1200 for a range of services; out[i].max_val_seen = the all-time maximum for every level */
int gys_export_hist_level(gys_ctx *ctx, int level, uint64_t tusec, uint32_t first_slot, uint32_t nslots, gys_hist_rec *out);
/* TIME_HISTOGRAM::get_stats_for_period_with_flush(starttime, endtime, pstats, nstats, tcount, tsum, mean_val, tnow) :1378-1413: the
 * statistics of the seconds [starttime, endtime], answered as folly does -- from the first level whose span reaches back to
 * starttime (tnow - 5 / 300 / 432000 s <= starttime, else since start), its ring buckets weighted by the fraction of each that the
 * interval covers (float, truncated: BucketedTimeSeries::rangeAdjust).  The since-start level is one bucket from the service's
 * first window close to tnow.  tusec = tnow in microseconds.  Level 0 is the engine's tumbling window (see above), so an interval
 * that starts within the last 5 s sees the window closed last or nothing. */
int gys_query_hist_period_stats(gys_ctx *ctx, uint64_t glob_id, int64_t starttime, int64_t endtime, uint64_t tusec, gys_time_hist_val *pstats,
				uint32_t nstats, int64_t *tcount, int64_t *tsum, double *mean_val);
/* the interval's {count, sum} per histogram bucket for a range of services (slabhist.buckets_[b].count(start, end) / sum(start, end));
 * out[i].total_count = their sum, max_val_seen = the all-time maximum; level_used (may be NULL) = the level that answered */
int gys_export_hist_period(gys_ctx *ctx, int64_t starttime, int64_t endtime, uint64_t tusec, uint32_t first_slot, uint32_t nslots, gys_hist_rec *out,
			   int *level_used);

typedef struct {
	uint64_t glob_id;
	int64_t tcount_5d, tsum_5d;
	uint32_t p95_5d_respms, p25_5d_respms, p95_qps, p25_qps, p95_nactive, p25_nactive;
} gys_listener_day_stats; /* == comm::LISTENER_DAY_STATS, 48 bytes, common/gy_comm_proto.h:1620-1632 */
/* What TCP_LISTENER::get_curr_state puts into LISTENER_DAY_STATS (common/gy_socket_stat.cc:2053-2112) for a range of services, as the
 * NOTIFY_LISTENER_DAY_STATS payload madhava consumes (handle_listener_day_stats server/gy_mconnhdlr.cc:12805).  The reference sends
 * zeros for a listener younger than 15 minutes; listener start times are control-plane state, that rule is left to the caller. */
int gys_export_day_stats(gys_ctx *ctx, uint64_t tusec, uint32_t first_slot, uint32_t nslots, gys_listener_day_stats *out);
/* the per-service QPS (which = 0, SEMI_LOG_HASH_LO) / active-connection (which = 1, HASH_1_3000) histograms fed by listener state */
int gys_export_svc_hist(gys_ctx *ctx, int which, uint32_t first_slot, uint32_t nslots, gys_hist_rec *out);

/* -------------------------------------------------------------------------------------------------------------------
 * the same queries as JSON in the reference's web shapes (field names and order = common/gy_json_field_maps.h):
 *   gys_json_svcsumm      {"madid":..,"summstats":[{time,nidle,ngood,nok,nbad,nsevere,ndown,totqps,totaconn,totkbin,totkbout,totsererr,
 *                          nsvc,nactive}],"hostinfo":{parid,host,madid,cluster}}   MCONN_HANDLER::web_curr_listener_summ (gy_mnodehandle.cc:1628)
 *   gys_json_svcstate     {"madid":..,"svcstate":[{time,svcid,name,qps5s,...,state,issue,ishttp,desc}],"hostinfo":{..}}
 *                          MCONN_HANDLER::web_curr_listener_state (gy_mnodehandle.cc:4650), json_db_svcstate_arr (:1102-1135)
 *   gys_json_clusterstate {"shyamaid":..,"clusterstate":[{time,cluster,nhosts,nprocissue,nprochosts,nproc,nlistissue,nlisthosts,nlisten,
 *                          totqps,svcnetmb,ncpuissue,nmemissue}]}   SHCONN_HANDLER::web_curr_clusterstate (gy_shnodehandle.cc:508)
 * buf receives a NUL-terminated string; *needed = strlen of the full result; GYS_ERR_NOMEM when buflen is too small.
 * timestr: the "time" field as the caller wants it printed (the reference prints local ISO-8601); NULL = "". */
int gys_set_host_name(gys_ctx *ctx, const uint8_t machine_id[16], const char *hostname); /* PARTHA_INFO::hostname_ for "host" */
int gys_json_svcsumm(gys_ctx *ctx, const uint8_t machine_id[16], const char *madhava_id16, const char *timestr, char *buf, size_t buflen, size_t *needed);
int gys_json_svcstate(gys_ctx *ctx, const uint8_t machine_id[16], const char *madhava_id16, const char *timestr, char *buf, size_t buflen, size_t *needed);
int gys_json_clusterstate(gys_ctx *ctx, const char *shyama_id16, const char *timestr, char *buf, size_t buflen, size_t *needed);
/* MCONN_HANDLER::web_curr_top_listeners (server/gy_mnodehandle.cc:2706-3190): {"madid":..,"topissue":[..],"topqps":[..],"topactconn":[..],
 * "topnet":[..],"summstats":{..}[,"hostinfo":{..}]}.  machine_id: one partha's four top-10 queues (+ hostinfo); NULL: every host's queues
 * merged into GYS_MULTI_TOPN = 50 slots per kind (MAX_MULTI_TOPN, common/gy_json_field_maps.h:481), entries carrying parid / host / madid /
 * cluster.  Entry = the svcstate fields + ip (empty: the registry holds (netns, port)) + port.  flags: which arrays to send. */
#define GYS_MULTI_TOPN 50
enum { GYS_TOP_ISSUE = 1, GYS_TOP_QPS = 2, GYS_TOP_ACTCONN = 4, GYS_TOP_NET = 8, GYS_TOP_SUMMSTATS = 16 };
int gys_json_toplisteners(gys_ctx *ctx, const uint8_t machine_id[16], uint32_t flags, const char *madhava_id16, const char *timestr, char *buf,
			  size_t buflen, size_t *needed);

/* ---- QUERY_OPTIONS on the live listener table: multi-host filter / sort / maxrecs and the aggregation operators --------------------------
 * Replaces the listener walk of MCONN_HANDLER::web_curr_listener_state (server/gy_mnodehandle.cc:4650-4900: every MTCP_LISTENER of every
 * partha under RCU, a SvcStateFields per listener, `filter_match` per row, `nrecs >= maxrecs`) for criteria on the numeric columns of
 * json_db_svcstate_arr (common/gy_json_field_maps.h:1102-1135): ONE device pass over the kept state records of all services.
 * Columns (SvcStateFields::get_num_field server/gy_mfields.h:1402-1440; every one is compared as an `int`, `issue` as int16_t; `state` is the
 * numeric OBJ_STATE_E a filter names through statefromjson, `ishttp` 0 / 1): */
enum { GYS_SVC_COL_QPS5S = 0, GYS_SVC_COL_NQRY5S, GYS_SVC_COL_RESP5S, GYS_SVC_COL_P95RESP5S, GYS_SVC_COL_P95RESP5M, GYS_SVC_COL_NCONNS,
       GYS_SVC_COL_NACTIVE, GYS_SVC_COL_NPROCS, GYS_SVC_COL_KBIN15S, GYS_SVC_COL_KBOUT15S, GYS_SVC_COL_SERERR, GYS_SVC_COL_CLIERR,
       GYS_SVC_COL_DELAYUS, GYS_SVC_COL_CPUDELUS, GYS_SVC_COL_IODELUS, GYS_SVC_COL_VMDELUS, GYS_SVC_COL_USERCPU, GYS_SVC_COL_SYSCPU,
       GYS_SVC_COL_RSSMB, GYS_SVC_COL_NISSUE, GYS_SVC_COL_STATE, GYS_SVC_COL_ISSUE, GYS_SVC_COL_ISHTTP, GYS_SVC_NCOLS };
/* comparators: the numeric members of COMPARATORS_E with its numbering (common/gy_query_criteria.h:28-46; match_num_criterian :1243-1290) */
enum { GYS_COMP_EQ = 0, GYS_COMP_NEQ, GYS_COMP_LT, GYS_COMP_LE, GYS_COMP_GT, GYS_COMP_GE, GYS_COMP_BIT2, GYS_COMP_BIT3, GYS_COMP_IN = 12, GYS_COMP_NOTIN = 13 };
#define GYS_SVC_MAX_TERMS 16
#define GYS_SVC_MAX_GROUPS 8
#define GYS_SVC_MAX_AGGR 8
typedef struct {
	uint8_t col;        /* GYS_SVC_COL_* */
	uint8_t comp;       /* GYS_COMP_* */
	uint8_t group;      /* criteria group of the term: 0 = CRITERIA_SET::l1_grp_, 1.. = its L2 groups (< GYS_SVC_MAX_GROUPS) */
	uint8_t reserved;
	uint32_t nvalues;   /* GYS_COMP_IN / NOTIN: the values are set_values[set_first .. set_first + nvalues) of the filter */
	uint32_t set_first;
	uint32_t reserved2;
	int64_t value;      /* the other comparators: converted to the column's own type before the compare, as the reference does */
} gys_svc_term;
typedef struct {
	const gys_svc_term *terms;
	uint32_t nterms;                         /* 0 = no criteria: every current record is listed (CRIT_SKIP) */
	uint32_t nset_values;
	const int64_t *set_values;
	uint8_t group_oper[GYS_SVC_MAX_GROUPS];  /* per group: 0 = all its terms must match (OPER_AND), 1 = any (OPER_OR); match_criteria_group :1535-1605 */
	uint8_t top_oper;                        /* how the groups combine: 0 = AND, 1 = OR (CRITERIA_SET::l1_oper_, match_criteria :1806-1900) */
	uint8_t reserved[3];
	uint32_t nmachine_ids;                   /* 0 = all hosts (is_multihost_); else only the listeners of these parthas ... */
	const uint8_t *machine_ids;              /* ... nmachine_ids x 16 bytes */
	/* criteria on the string columns that select rather than compare (all three AND with each other and with the terms): */
	const uint64_t *svcids;                  /* svcid = / in (...): only these listeners -- the reference's direct-lookup path      */
	uint32_t nsvcids;                        /*   (server/gy_mnodehandle.cc:4754-4860); 0 = not restricted; unknown ids match nothing */
	uint32_t nclusters;                      /* cluster = / in (...): only hosts of these clusters (names as registered); 0 = not restricted */
	const char *const *clusters;
} gys_svc_filter;
typedef struct {
	uint32_t slot, host_slot; /* service slot (gys_lookup_service) and host slot (gys_register_host) */
	uint8_t rec[88];          /* the kept comm::LISTENER_STATE_NOTIFY of the listener */
} gys_svc_row;
/* The records whose state is current (this or the last window: the reference lists states at most 10 s old, :4660) and that pass the
 * filter; at most maxrecs of them, ordered by sort_col (GYS_SVC_COL_*; descending when sort_desc) and then by service slot -- sort_col < 0:
 * by service slot (registration order; the reference's order is that of its hash-table walk).  When more records match than maxrecs the
 * FIRST maxrecs of that order are returned (an exact top-k on the device).  *nmatched = records that matched. */
int gys_query_svcstate_scan(gys_ctx *ctx, const gys_svc_filter *filter, int sort_col, int sort_desc, uint32_t maxrecs, gys_svc_row *out,
			    uint32_t *nout, uint64_t *nmatched);
/* A criterion on the service's NAME (SvcStateFields "name" / "svcname": a string field, server/gy_mfields.h): the glob_ids of the
 * registered services whose process name (gys_listener_info.comm) matches -- to be handed to gys_svc_filter.svcids.  comp = the
 * reference's string comparators with its numbering (COMPARATORS_E common/gy_query_criteria.h:28-45; match_str_criterian :1335-1383):
 * GYS_COMP_EQ / NEQ (whole name), GYS_COMP_SUBSTR / NOTSUBSTR (memmem), GYS_COMP_LIKE / NOTLIKE (a regular expression matched anywhere in
 * the name, as RE2::PartialMatch -- here a linear-time automaton search over bytes (gys_regex.hpp) with RE2's syntax minus its Unicode
 * classes, \\C and \\Q..\\E; never a backtracking search: a pattern from a web query cannot stall the criterion; an invalid or unsupported
 * expression is GYS_ERR_INVAL as the reference's ERR_INVALID_REQUEST), GYS_COMP_IN / NOTIN (any / none of npatterns whole names).  The other comparators use patterns[0].  Host-side:
 * the names never leave the host.  *nout = services that match; GYS_ERR_NOMEM when that exceeds cap (the first cap are written). */
enum { GYS_COMP_SUBSTR = 8, GYS_COMP_NOTSUBSTR = 9, GYS_COMP_LIKE = 10, GYS_COMP_NOTLIKE = 11 };
int gys_svc_ids_by_name(gys_ctx *ctx, int comp, const char *const *patterns, uint32_t npatterns, uint64_t *out_ids, uint32_t cap, uint32_t *nout);
/* ... and on the HOST name (gys_set_host_name: PARTHA_INFO::hostname_, the "host" column): the machine ids of the registered hosts whose
 * name matches, 16 bytes each, for gys_svc_filter.machine_ids (a host without a name has the empty name). */
int gys_machine_ids_by_hostname(gys_ctx *ctx, int comp, const char *const *patterns, uint32_t npatterns, uint8_t *out_ids16, uint32_t cap, uint32_t *nout);
/* the same as the reference's multi-host JSON: {"madid":..,"svcstate":[{"parid","host","madid","cluster", then the json_db_svcstate_arr
 * columns}, ...]} (column list of a multi-host query: QUERY_OPTIONS::get_all_column_list common/gy_query_common.h:418-437) */
int gys_json_svcstate_multihost(gys_ctx *ctx, const gys_svc_filter *filter, int sort_col, int sort_desc, uint32_t maxrecs,
				const char *madhava_id16, const char *timestr, char *buf, size_t buflen, size_t *needed);
/* AGGR_OPER_E (common/gy_json_field_maps.h:114-129) over the records that pass the filter, grouped by nothing (group_by 0: one row,
 * group 0), by host (1: group = host slot) or by cluster (2: group = cluster index in registration order).  One row per group that has a
 * matching record, in group order; every row carries count and, per requested column, the exact 64-bit sum, min and max, from which
 * gys_svc_aggr_value derives the operator: sum, avg (sum / count), max, min, count, bool_or, bool_and.  (percentile: see
 * gys_query_svcstate_percentiles; first / last have no meaning on one snapshot of the live table and are not built.)  *nrows = rows there are (may exceed maxrows; only maxrows are written). */
enum { GYS_AOPER_SUM = 1, GYS_AOPER_AVG, GYS_AOPER_MAX, GYS_AOPER_MIN, GYS_AOPER_COUNT, GYS_AOPER_BOOL_OR = 9, GYS_AOPER_BOOL_AND = 10 };
typedef struct {
	uint32_t group, ncols;
	uint64_t count;
	int64_t sum[GYS_SVC_MAX_AGGR], min[GYS_SVC_MAX_AGGR], max[GYS_SVC_MAX_AGGR];
} gys_svc_aggr_row;
int gys_query_svcstate_aggr(gys_ctx *ctx, const gys_svc_filter *filter, int group_by, const uint8_t *cols, uint32_t ncols, gys_svc_aggr_row *out,
			    uint32_t maxrows, uint32_t *nrows);
int gys_svc_aggr_value(const gys_svc_aggr_row *row, uint32_t col_index, int oper, double *out);
/* AOPER_PERCENTILE (AGGR_OPER_E common/gy_json_field_maps.h:114-129) of ONE column over all records that pass the filter: out[i] = the
 * discrete percentile pcts[i] (0 < p <= 1: the smallest value with at least that fraction of the matching records at or below it, SQL
 * percentile_disc), exact -- the k-th key of the sorted scan found by the same radix selection, nothing gathered.  *nmatched = records
 * that matched (0: out[] is zero).  For a percentile per host or cluster, name the group in the filter (machine_ids / clusters). */
int gys_query_svcstate_percentiles(gys_ctx *ctx, const gys_svc_filter *filter, int col, const double *pcts, uint32_t npcts, int64_t *out, uint64_t *nmatched);
/* The multi-host form of web_curr_listener_summ (server/gy_mnodehandle.cc:1628-1690: the walk over partha_tbl_, SvcSummFields::filter_match per
 * host, hosts whose listener state is older than 10 s skipped): {"madid":..,"summstats":[{"parid","host","madid","cluster", then the
 * json_db_svcsumm_arr columns}, ...]} for every host that reported listener states in the last finished window and passes the filter.  The
 * filter is a gys_svc_filter whose terms name the numeric columns of json_db_svcsumm_arr (common/gy_json_field_maps.h:1396-1416):
 * GYS_SUMM_COL_*; machine_ids / clusters select hosts, svcids is ignored.  Rows in host-slot order, or sorted by sort_col (then host slot);
 * at most maxrecs rows.  (A few thousand 52-byte rows: filtered on the host side after one copy of the per-host summaries.) */
enum { GYS_SUMM_COL_NIDLE = 0, GYS_SUMM_COL_NGOOD, GYS_SUMM_COL_NOK, GYS_SUMM_COL_NBAD, GYS_SUMM_COL_NSEVERE, GYS_SUMM_COL_NDOWN, GYS_SUMM_COL_TOTQPS,
       GYS_SUMM_COL_TOTACONN, GYS_SUMM_COL_TOTKBIN, GYS_SUMM_COL_TOTKBOUT, GYS_SUMM_COL_TOTSERERR, GYS_SUMM_COL_NSVC, GYS_SUMM_COL_NACTIVE, GYS_SUMM_NCOLS };
int gys_json_svcsumm_multihost(gys_ctx *ctx, const gys_svc_filter *filter, int sort_col, int sort_desc, uint32_t maxrecs, const char *madhava_id16,
			       const char *timestr, char *buf, size_t buflen, size_t *needed);

/* The per-listener 5-second scan from the engine's OWN state (needs gys_config.enable_levels): replaces the loop of
 * TCP_SOCK_HANDLER::listener_stats_update (common/gy_socket_stat.cc:4044-4365) that turns every listener's counters and histograms into
 * one comm::LISTENER_STATE_NOTIFY (common/gy_comm_proto.h:2183-2254), and the data-parallel part of TCP_LISTENER::get_curr_state
 * (:2030-2143): the p95 bucket ids of the 5-s / 5-min / 5-day levels and the QPS / active-connection histogram percentiles it compares.
 * Call it after the window close at `tusec` (level 0 = the window closed last).  One pass over all services, nothing is modified.
 *   d_notify (device, 88 B x gys_num_services, or NULL): LISTENER_STATE_NOTIFY records with glob_id_, nqrys_5s_ (= the 5-s level's count),
 *            total_resp_5sec_, nconns_ / nconns_active_ (CONN_BITMAP::get_conn_breakup maximum, common/gy_socket_stat.h:413-429),
 *            p95_5s_resp_ms_, p95_5min_resp_ms_, curr_state_ (STATE_IDLE when the QPS is 0, else STATE_OK: the state POLICY -- task / cpu /
 *            memory issue inputs, issue strings -- is not the engine's); every other field 0.  They can be handed to
 *            gys_ingest_listener_state_dev as they are (host roll-up, top-N, QPS / active-connection histogram samples).
 *   d_scan   (device, gys_listener_scan x gys_num_services, or NULL).
 *   qps_multiple = TCP_SOCK_HANDLER::get_bpf_qps_multiple(); diffsec = seconds since the previous scan (:4046, :4109). */
typedef struct {
	uint64_t glob_id;
	int64_t tcount[4], tsum[4];                 /* RESP_STATS::tcount_ / tsum_ of the 5 s / 5 min / 5 d / all-time levels */
	int32_t p95_ms[4], p99_ms[4], p25_ms[4];    /* RESP_STATS::stats_[0..2].data_value */
	int32_t last_qps;                           /* total_queries * multiple / diffsec (last_qps_count_) */
	int32_t curr_qps;                           /* max(last_qps, tcount[0] / 5) */
	int32_t qps_p95, qps_p25, act_p95, act_p25; /* GY_HISTOGRAM::get_percentiles {95, 25} of the QPS / active-connection histograms */
	uint8_t b5, b300, b5day;                    /* get_bucketid_from_threshold<RESP_TIME_HASH>(p95 of the level) */
	uint8_t nconn_active;                       /* max over nactive_conn_arr */
	uint8_t nactive_conn_arr[15];               /* CONN_BITMAP::get_conn_breakup of the window closed last */
	uint8_t reserved[5];
} gys_listener_scan;
int gys_scan_listener_state_dev(gys_ctx *ctx, uint64_t tusec, float qps_multiple, uint32_t diffsec, void *d_notify, gys_listener_scan *d_scan);

/* The listener's state decision: TCP_LISTENER::get_curr_state (common/gy_socket_stat.cc:2020-2870) and its caller's part (:4241-4266) for
 * ALL listeners at once -- OBJ_STATE_E / LISTENER_ISSUE_SRC from the scan records above and from the inputs the listener's own histograms do
 * not hold (task status, host CPU / memory issue flags, server errors, connection count, dependent servers), which the reference takes from
 * RELATED_LISTENERS::LISTENER_TASK_STATUS / is_task_issue (:1914, :2043) and from the host's CPU / memory monitors.  Every branch of the
 * reference that is a function of those values is evaluated (the issue STRING is not produced).  The two history bytes of a listener
 * (TCP_LISTENER::issue_bit_hist_, high_resp_bit_hist_: common/gy_socket_stat.h:656-657) are engine state, advanced by every call.
 *   d_scan      (device) gys_listener_scan x gys_num_services, as gys_scan_listener_state_dev left them;
 *   d_issue_in  (device, or NULL = no errors, no task / host issue, ten days of history) gys_listener_issue_in x gys_num_services;
 *   d_notify    (device, or NULL) the 88-byte records of the scan: curr_state_ @79, curr_issue_ @80, issue_bit_hist_ @81, high_resp_bit_hist_ @82
 *               and, from the inputs, ser_errors_ @44, tasks_delay_usec_ @52, tasks_cpudelay_usec_ @56, tasks_blkiodelay_usec_ @60, ntasks_issue_ @76 (nconns_ @16 when inputs are given) are filled in;
 *   d_out       (device, or NULL) gys_listener_decision x gys_num_services. */
enum { GYS_LI_TASK_ISSUE = 1, GYS_LI_SEVERE = 2, GYS_LI_DELAY = 4, GYS_LI_CPU_ISSUE = 8, GYS_LI_MEM_ISSUE = 16, GYS_LI_DEPENDS = 32, GYS_LI_YOUNG = 64 };
typedef struct {
	uint32_t ser_errors;                                                   /* get_curr_state's ser_errors argument */
	uint32_t tasks_delay_msec, tasks_cpudelay_msec, tasks_blkiodelay_msec; /* LISTENER_TASK_STATUS::tasks_*_usec_ / 1000 (:2047, :2784-2785) */
	int32_t nconn;                                                         /* last_chk_nconn_ (:2040) */
	uint16_t ntasks_issue, ntasks_noissue;                                 /* is_task_issue's counts (:2043-2044) */
	uint8_t flags;                                                         /* GYS_LI_*: task_issue, is_severe, is_delay, cpu_issue, mem_issue, dependent servers exist (:2823-2829), started less than 100 s ago (:4244) */
	uint8_t pad[3];
	int64_t tdiff_start;                                                   /* seconds the listener's response histogram covers (:2033-2034); <= 0: the full 5 days */
} gys_listener_issue_in;
typedef struct {
	uint8_t state, issue;                       /* OBJ_STATE_E (0 Idle, 1 Good, 2 OK, 3 Bad, 4 Severe), LISTENER_ISSUE_SRC (common/gy_json_field_maps.h:242-250, :419-434) */
	uint8_t issue_bit_hist, high_resp_bit_hist; /* the listener's history bytes after this call */
	uint16_t decided_line;                      /* the line of common/gy_socket_stat.cc whose `return` (or the function's end, 2866; 4262 = "just started") decided */
	uint16_t pad;
} gys_listener_decision;
int gys_decide_listener_state_dev(gys_ctx *ctx, const gys_listener_scan *d_scan, const gys_listener_issue_in *d_issue_in, void *d_notify,
				  gys_listener_decision *d_out);

/* -------------------------------------------------------------------------------------------------------------------
 * parity / checkpoint exports (host destination buffers) -- GY_HISTOGRAM::get_serialized analogue (gy_statistics.h:665-673) */
uint32_t gys_num_services(gys_ctx *ctx);
uint32_t gys_num_hosts(gys_ctx *ctx);
int gys_lookup_service(gys_ctx *ctx, uint64_t glob_id, uint32_t *slot);
int gys_export_hist(gys_ctx *ctx, int which, uint32_t first_slot, uint32_t nslots, gys_hist_rec *out);
/* CONN_BITMAP rows of the open window: per service 32 u16 rows of resp_bitmap_v4_ followed by the 32 rows of resp_bitmap_v6_ (common/gy_socket_stat.h:645, :665) */
int gys_export_conn_bitmap(gys_ctx *ctx, uint32_t first_slot, uint32_t nslots, uint16_t *out /* [nslots*64] */);
int gys_export_hll(gys_ctx *ctx, uint8_t *out /* [1 << GYS_HLL_P] */);
int gys_export_cms(gys_ctx *ctx, int which, void *out /* which 0: u32[D*W]; which 1: i64[D*W] */);
int gys_export_tdigest(gys_ctx *ctx, uint32_t first_slot, uint32_t nslots, int64_t *sums /* [nslots*100] */, uint32_t *cnts /* [nslots*100] */,
		       int32_t *minmax /* [nslots*2] */);
/* values a service's t-digest still buffers unmerged (unordered; only the first npend[i] entries of row i are meaningful) */
int gys_export_tdigest_pending(gys_ctx *ctx, uint32_t first_slot, uint32_t nslots, uint32_t *npend /* [nslots] */,
			       int32_t *pend /* [nslots * gys_td_pend_cap(ctx)] */);
uint32_t gys_td_pend_cap(gys_ctx *ctx); /* the context's td_pend_cap (GYS_TD_PEND_CAP when the configuration left it 0) */
/* per-service connection counters of the TCP_CONN_NOTIFY roll-up, [nslots*4]: nconn, nclose, bytes_sent, bytes_rcvd.  CONNECTIONS, not
 * records: only the ACCEPTING partha's records count (is_tcp_accept_event_, a loopback record included; the connecting half names the
 * same ser_glob_id_ and would count the connection twice), nconn += 1 on a record with notified_before_ clear (the open notification, or
 * the only record of a short-lived connection), nclose += 1 on a record with tusec_close_ set, bytes as reported (server/gy_mconnhdlr.cc:
 * 9129-9181).  gys_query_cms(which 0 / 1) estimates the same nconn / bytes per service for the last finished window. */
int gys_export_svc_counters(gys_ctx *ctx, uint32_t first_slot, uint32_t nslots, uint64_t *out);
int gys_export_pair_cms(gys_ctx *ctx, int which, void *out /* which 0 / 2 / 4: u32[D*W]; which 1 / 3 / 5: i64[D*W]; see gys_query_pair_cms */);
int gys_export_active_conn_counters(gys_ctx *ctx, uint32_t first_slot, uint32_t nslots, uint64_t *out /* [nslots*4]: rows, bytes_sent, bytes_received, active conns */);
int gys_export_global_hist(gys_ctx *ctx, gys_hist_rec *out); /* all-service response histogram of the last finished window (all ranks) */
int gys_export_svc_hll(gys_ctx *ctx, uint32_t first_slot, uint32_t nslots, uint8_t *out /* [nslots << svc_hll_p] */);

typedef struct {
	uint64_t resp_events, resp_dropped_range, resp_dropped_nolistener;
	uint64_t conn_events, conn_unknown_service;
	uint64_t lstate_records, lstate_missed, lstate_errors, lstate_deleted;
	uint64_t resp_batches_host_local, resp_batches_general; /* which resp pipeline each ingest call took (gys_config.resp_path) */
	uint64_t window_graph_launches; /* window boundaries replayed from the captured hipGraph (0: plain stream operations were used) */
	uint64_t resp_batches_host_split; /* host-local pipeline in its split form (few hosts, long segments: parts of 65536 events) */
	uint64_t td_merges, td_merge_values; /* t-digest re-clusterings queued so far and the buffered values they merged */
	uint64_t actconn_records, actconn_remote_listen, actconn_unknown_listener; /* ACTIVE_CONN_STATS rows of local listeners / of listeners on
										      another madhava (is_remote_listen_) / of local listeners not registered */
	uint64_t stage_waits;       /* host-pointer calls that found their staging slot still in flight and waited for the GPU (ring wrapped) */
	uint64_t resp_calls_queued; /* gys_ingest_resp_events calls that went through the submission queue ... */
	uint64_t resp_submissions;  /* ... and the combined batches they were submitted as (calls / submissions = calls per launch set) */
	/* the tallies MCONN_HANDLER::partha_tcp_conn_info keeps while it walks a message (server/gy_mconnhdlr.cc:9133-9137, :9327: nnew,
	 * nclosed, nclosed_no_not), summed over all messages: open notifications / records with tusec_close_ / closes of connections
	 * whose open was never notified.  conn_new + conn_closed_no_notify = connection HALVES seen (each connection is reported by its
	 * accepting and by its connecting partha); conn_client_side = records of the connecting half only (is_tcp_connect_event_ without
	 * is_tcp_accept_event_), which do not enter the per-service counters (see gys_export_svc_counters). */
	uint64_t conn_new, conn_closed, conn_closed_no_notify, conn_client_side;
	uint64_t resp_tail_flushes; /* submissions made by the queue's flusher thread: the tail of a burst of gys_ingest_resp_events calls that
				     * found the GPU busy is submitted ~200 us after a submission retires, without waiting for the next call */
	/* gys_ingest_tcp_conn / gys_ingest_listener_state calls that went through their submission queues, the combined batches they were
	 * submitted as, and the submissions made by those queues' flusher threads */
	uint64_t conn_calls_queued, conn_submissions, lstate_calls_queued, lstate_submissions, rec_tail_flushes;
	uint64_t resp_run_overflow; /* response values a batch could not place (a run past the end of the batch staging area): 0 by construction, checked in the kernel */
} gys_counters;
int gys_get_counters(gys_ctx *ctx, gys_counters *out);
/* response events the submission queue of gys_ingest_resp_events still holds on the HOST side (copied out of the callers' buffers, not yet
 * submitted to the GPU).  Unlike every other entry point this one does not flush the queue: it is the way to watch it drain. */
int gys_resp_queue_pending(gys_ctx *ctx, uint64_t *events);

/* -------------------------------------------------------------------------------------------------------------------
 * standalone keyed histogram op (rows a1/a2/a4 of SURVEY 8a for ANY hash kind): nkeys histograms of `kind`, caller-owned DEVICE
 * memory hist[nkeys] (zero-initialised except max_val_seen = type min, see gys_hist_init_dev). */
int gys_hist_init_dev(gys_ctx *ctx, int kind, gys_hist_rec *d_hist, uint32_t nkeys);
int gys_hist_add_dev(gys_ctx *ctx, int kind, gys_hist_rec *d_hist, uint32_t nkeys, const uint32_t *d_keyidx, const int32_t *d_vals, uint64_t n);
int gys_hist_merge_dev(gys_ctx *ctx, gys_hist_rec *d_dst, const gys_hist_rec *d_src, uint32_t nkeys); /* add_histogram :625 */
int gys_hist_percentiles_dev(gys_ctx *ctx, int kind, const gys_hist_rec *d_hist, uint32_t nkeys, const float *pcts, uint32_t npct,
			     int64_t *d_out);

/* -------------------------------------------------------------------------------------------------------------------
 * measurement helpers: per-kernel HIP-event timing on the context stream */
int gys_profile_enable(gys_ctx *ctx, int on);
int gys_profile_reset(gys_ctx *ctx);
/* name: "resp_host", "key_pass", "digest_merge", "digest_huge" (general pipeline: "resp_pass1", "scan", "scatter"), "conn", "lstate", "wire_decode", ...; returns accumulated ms + launches */
int gys_profile_get(gys_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches);
int gys_profile_names(gys_ctx *ctx, char *buf, size_t buflen); /* comma separated */

/* synthetic stream generators running ON the GPU (bench/test plumbing).  They write into caller DEVICE buffers. */
/* zipf_milli: 0 = services uniform; s*1000 = Zipf(s) over the services of a host; GYS_GEN_SPREAD = service weights spread evenly
 * over 0..255/256 (bench.py uses one such pass before timing so that the keys' t-digest buffers start at evenly spread fill levels
 * -- with identical rates and identical start all keys would otherwise overflow in the same window) */
#define GYS_GEN_SPREAD 0xFFFFFFFFu
/* PMC calibration helper: reads nevents 24-byte events with the access pattern of the event kernel (3 x 8-B loads per thread at a 24-B
 * stride) and does nothing else -- FETCH_SIZE of this launch vs the known 24 B x nevents (tools/calibrate_fetch.py) */
int gys_debug_read_events_dev(gys_ctx *ctx, const void *d_ev24, uint64_t nevents);
int gys_gen_resp_events_dev(gys_ctx *ctx, void *d_ev24, uint64_t nevents, uint64_t seed, uint32_t first_host, uint32_t nhosts,
			    uint32_t svcs_per_host, uint32_t zipf_milli /* 0 = uniform, else s*1000 */, gys_resp_seg *segs_out /* host, nhosts */);

#ifdef __cplusplus
}
#endif
#endif

// gys_engine.hip -- host side of libgysketch.so: context, registries, the C ABI of include/gysketch.h.
//
// C++17 host code + hand-written HIP kernels (gys_kernels.hpp) for gfx950.  There is no CPU fallback anywhere in this library:
// every ingest/query path runs on the GPU or fails with GYS_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <thread>
#include <condition_variable>
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "gys_kernels.hpp"
#include "gys_rollup.hpp"
#include "gys_huge.hpp"
#include "gys_svcquery.hpp"

using namespace gys;

namespace {

thread_local char g_err[512] = "";

void set_err(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

#define HIPCHK(expr)                                                                                       \
	do {                                                                                               \
		hipError_t e_ = (expr);                                                                    \
		if (e_ != hipSuccess) {                                                                    \
			set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			return GYS_ERR_HIP;                                                                \
		}                                                                                          \
	} while (0)

struct MachId {
	uint64_t first, second;
	bool operator==(const MachId &o) const { return first == o.first && second == o.second; }
};
struct MachIdHash {
	// GY_MACHINE_ID::get_hash common/gy_sys_hardware.h:82-85
	size_t operator()(const MachId &m) const
	{
		uint32_t w[4];
		memcpy(w, &m.first, 8);
		memcpy(w + 2, &m.second, 8);
		uint32_t k[6] = {w[0], w[1], w[2], w[3], 0, 0};
		return jhash2<6>(k, 4, GYS_SEED);
	}
};

inline MachId to_machid(const uint8_t id[16])
{
	MachId m;
	memcpy(&m.first, id, 8);
	memcpy(&m.second, id + 8, 8);
	return m;
}

struct ProfEntry {
	double total_ms = 0;
	uint64_t launches = 0;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// host mirror of one host's listener sub-table (the device copy lives in the table / list pools, see k_resp_host)
struct HostListeners {
	std::vector<uint64_t> tbl;   // open addressing, entries (netns:32|port:16) << 16 | local index, ~0 = free
	std::vector<uint32_t> slots; // local index -> service slot
	std::vector<uint32_t> all_slots; // every service slot ever registered for the host (queries)
	std::vector<uint64_t> keys;  // local index -> (netns:32|port:16)
	uint32_t tbl_off = 0, tbl_cap = 0, lst_off = 0;
	bool on_device = false;
	bool overflow = false;       // more listeners than the LDS path supports (or pools exhausted): general pipeline only
	// A host with more than GYS_HOST_MAX_LOCAL listeners is cut into `sub.size()` PARTS (a power of two): a listener belongs to part
	// host_part_of(key48), every part has a sub-table of its own (the members above are then unused) and its own descriptor
	// hdesc[sub_desc + part]; k_resp_host runs one workgroup per (segment, part), each resolving only its part's events.
	std::vector<HostListeners> sub;
	uint32_t sub_desc = 0;
	// Keys with CANDIDATES (host level; a part's sub-table only holds the entries): a (netns, port) key whose listener is bound to an address, or
	// that has more than one listener.  chain[key48] = its listeners in registration order; `cands` = the device records of all chains
	// (a region of the candidate pool, rebuilt when a chain changes); a key that is not in `chain` has ONE any-address listener and its
	// sub-table entry names the local index directly, as before.
	struct LAddr {
		uint32_t ip32 = 0;
		uint32_t ip128[4] = {0, 0, 0, 0};
		bool any = true;
	};
	struct ChainEnt {
		uint32_t slot;
		LAddr a;
	};
	std::unordered_map<uint64_t, std::vector<ChainEnt>> chain;
	std::vector<ListenerCand> cands;
	std::unordered_map<uint64_t, uint32_t> cand_first; // key48 -> first record of its chain in `cands`
	uint32_t cand_off = 0, cand_cap = 0;
	bool dirty = false;                                 // a chain changed: tables and candidate region are rebuilt at the end of the registration call
	std::unordered_map<uint64_t, uint32_t> okey;        // overflow hosts (no sub-tables): key48 -> slot of the key's one any-address listener
};

#define GYS_HOST_MAX_LOCAL 2048u // listeners per sub-table the LDS path supports (32 KB LDS table + 48 KB per-key areas + the tile image)
#define GYS_HOST_MAX_PARTS 16u   // ... and parts per host: up to 32768 listeners per host on the host-local path

struct ArenaLayout {
	uint64_t off_hll8, off_u32, n_u32, off_i64sum, n_i64sum, off_i64max, n_i64max, total;
	uint64_t u32_cms, u32_cluster, u32_pair, u32_misc, u32_cpair; // element offsets inside the u32 section
	uint64_t i64_cms, i64_ghist, i64_pair, i64_cpair;             // element offsets inside the i64 SUM section
};

// u32_pair / i64_pair: Count-Min pair of the ACTIVE_CONN_STATS roll-up; u32_misc[0]: its local-listener rows of the window;
// u32_cpair / i64_cpair: the TCP_CONN_NOTIFY pair roll-up, present only with gys_config.conn_pair_cms (its own tables: the two feeds
// count different things -- a gauge of active connections vs. closed connections -- and must not share cells): two table pairs, the
// listener side (connlistenmap_: records of the accepting partha) and behind it the client side (connclientmap_: connect-only records)
ArenaLayout arena_layout(uint32_t max_clusters, bool conn_pair)
{
	ArenaLayout a;
	a.off_hll8 = 0;
	a.off_u32 = align_up((uint64_t)1 << GYS_HLL_P, 256);
	a.u32_cms = 0;
	a.u32_cluster = (uint64_t)GYS_CMS_D * GYS_CMS_W;
	a.u32_pair = a.u32_cluster + align_up((uint64_t)max_clusters * 12, 64); // Count-Min pair of the (listener, client task) roll-up
	a.u32_misc = a.u32_pair + 2ull * GYS_CMS_D * GYS_CMS_W; // (local-listener rows' table, then the remote-listener rows')
	a.u32_cpair = a.u32_misc + 64;
	a.n_u32 = a.u32_cpair + (conn_pair ? (uint64_t)2 * GYS_CMS_D * GYS_CMS_W : 0); // listener-side tables, then client-side tables
	a.off_i64sum = align_up(a.off_u32 + a.n_u32 * 4, 256);
	a.i64_cms = 0;
	a.i64_ghist = (uint64_t)GYS_CMS_D * GYS_CMS_W;
	a.i64_pair = a.i64_ghist + 32;
	a.i64_cpair = a.i64_pair + 2ull * GYS_CMS_D * GYS_CMS_W;
	a.n_i64sum = a.i64_cpair + (conn_pair ? (uint64_t)2 * GYS_CMS_D * GYS_CMS_W : 0);
	a.off_i64max = align_up(a.off_i64sum + a.n_i64sum * 8, 256);
	a.n_i64max = 8;
	a.total = align_up(a.off_i64max + a.n_i64max * 8, 256);
	return a;
}

} // namespace

#define GYS_SEG_RING 4

struct gys_ctx {
	gys_config cfg{};
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	hipStream_t copy_stream = nullptr; // H2D copies of the response submissions: the copy of submission k + 1 runs under the kernels of submission k
	int ncu = 256;

	// registries (host)
	std::unordered_map<MachId, uint32_t, MachIdHash> host_map;
	std::vector<MachId> hosts;
	std::vector<std::string> host_names;
	std::vector<std::array<char, 16>> svc_comm; // TASK_COMM_LEN process name per service slot (SvcStateFields "name")
	std::vector<uint64_t> svc_gid_h;
	std::vector<uint32_t> host_cluster_h;
	std::unordered_map<std::string, uint32_t> cluster_map;
	std::vector<std::string> cluster_names;
	std::unordered_map<uint64_t, uint32_t> gid_map_h; // glob_id -> slot (host mirror for single-key queries)
	uint32_t nsvc = 0;
	std::vector<HostListeners> host_lst;
	std::vector<uint32_t> host_seen; // batch stamp per host (duplicate-host detection in a multi-segment batch)
	uint32_t batch_stamp = 0;
	uint64_t n_batches_host_local = 0, n_batches_general = 0, n_batches_host_split = 0;
	uint64_t htbl_used = 0, htbl_cap = 0, hlst_used = 0, hlst_cap = 0;
	uint32_t resp_dyn_max = 0; // dynamic LDS a k_resp_host launch may use (160 KiB minus the kernel's static part)
	uint64_t *htbl = nullptr; // pool of per-host sub-tables
	uint32_t *hlst = nullptr; // pool of per-host local index -> slot lists
	HostDesc *hdesc = nullptr; // [max_hosts] by host slot, then hdesc_ext_cap descriptors of the parts of many-listener hosts
	ListenerCand *cand_pool = nullptr; // candidate records of the keys that need the server address (allocated on first use)
	uint64_t cand_used = 0, cand_cap = 0;
	uint32_t hdesc_ext_used = 0, hdesc_ext_cap = 0;

	// device state
	DevTable lk_tbl{}, gid_tbl{};
	uint64_t *svc_gid = nullptr;
	gys_hist_rec *hist_win = nullptr, *hist_all = nullptr;
	uint32_t *bitmap = nullptr;
	int64_t *td_sum = nullptr;
	uint32_t *td_cnt = nullptr;
	TdMeta *td_meta = nullptr;
	int2 *td_minmax = nullptr;
	uint32_t *td_pend = nullptr;     // per service a buffer of pcap staged words ("per-key value buffers" in gys_kernels.hpp)
	uint32_t *td_cur = nullptr;      // words in each buffer (== td_meta.npend between batches)
	uint32_t *td_run = nullptr;      // spilled services: fill cursor of the run in `staged`
	uint32_t *svc_host = nullptr;    // host slot of each service
	uint32_t *host_spill = nullptr;  // per host: batch stamp of the last batch in which one of its services spilled
	// predicted runs (k_prespill): start / end of a service's predicted run, the values its last batch brought, "worth predicting" flags of
	// the previous / this batch, hosts of the running batch, keys whose run goes into the buffer after all
	uint32_t *td_run0 = nullptr, *td_run1 = nullptr, *td_prevm = nullptr, *pre_hot = nullptr, *host_batch = nullptr;
	MergeEnt *append_list = nullptr;
	uint64_t append_cap = 0, staged_cap = 0;
	uint32_t pre_seq = 0; // k_prespill launches so far (parity = which pre_hot word it reads)
	bool prespill = false;
	uint32_t spill_stamp = 0;
	uint32_t pcap = 0;
	uint32_t pend_cap = GYS_TD_PEND_CAP, merge_fast = GYS_TDIGEST_MERGE_FAST; // the t-digest rule's buffer size (gys_config.td_pend_cap) and its fast merge class
	MergeEnt *merge_list = nullptr, *merge_list_slow = nullptr, *merge_list1 = nullptr, *merge_list2 = nullptr, *huge_list = nullptr, *query_list = nullptr;
	uint32_t *merge_count = nullptr; // [FIN_*]: merge list lengths by size class, huge list length, run allocation cursor; [8] = 1 (query list)
	uint32_t *resp_win = nullptr;    // per service: response events of the open window (-> Count-Min rows at the window boundary)
	uint32_t *cms_partial = nullptr; // [cms_nch][GYS_CMS_D][GYS_CMS_W] partial rows of k_cms_partial
	uint32_t cms_nch = 0;
	bool resp_dirty = false;
	int64_t *query_sum = nullptr;    // scratch of the non-destructive merge behind gys_query_quantiles
	uint32_t *query_cnt = nullptr;
	uint32_t *batch_cnt = nullptr, *batch_off = nullptr, *scan_block_sums = nullptr;
	uint64_t *ev_kv = nullptr;       // general front end: (slot, staged word) per event
	uint64_t ev_kv_cap = 0; // events
	uint32_t *staged = nullptr;      // runs of the general front end / of spilled services
	// the window boundary's fixed sequence of copies / clears, captured once per registry shape as a hipGraph and replayed
	hipGraph_t win_graph = nullptr;
	hipGraphExec_t win_graph_exec = nullptr;
	uint64_t win_graph_shape = 0; // (hosts, services) the graph was captured for
	int win_graph_state = 0;      // 0 not tried, 1 usable, -1 capture unavailable: plain launches
	uint64_t win_graph_launches = 0;
	int64_t i64min = INT64_MIN;   // stable host source of the graph's 8-byte copy
	uint32_t *huge_scratch = nullptr;
	// several-workgroups-per-key path (gys_huge.hpp): per-entry accumulators, chunk prefix, global tail list, fallback list
	unsigned long long *huge_acc = nullptr, *huge_tail = nullptr;
	uint32_t *huge_tb_list = nullptr;
	uint32_t *huge_bm = nullptr, *huge_chunk_off = nullptr;
	MergeEnt *huge_fb_list = nullptr;
	uint32_t huge_maxent = 0;
	uint64_t huge_list_cap = 0;
	int huge_blocks = 0;
	uint32_t *hll32 = nullptr;
	unsigned long long *svc_ctr = nullptr;
	unsigned long long *svc_act = nullptr; // per listener: ACTIVE_CONN_STATS rows, bytes sent / received, active connections (cumulative)
	unsigned long long *svc_win = nullptr; // per-service window accumulators of the connection path (k_conn_ingest / k_conn_fold)
	bool conn_dirty = false;
	uint8_t *svc_state = nullptr;
	unsigned long long *svc_claim = nullptr; // [S] k_lstate_ingest / k_lstate_keep: last record of a call per listener
	uint32_t lstate_launch = 0;
	uint8_t *svc_hll = nullptr;
	int32_t *host_summ_win = nullptr, *host_summ_last = nullptr;
	gys_host_state *host_state = nullptr;
	uint32_t *host_state_epoch = nullptr, *host_cluster = nullptr;
	uint64_t *counters = nullptr;
	uint32_t *misc = nullptr; // [0] table insert failures, [1] topn count
	// segment descriptors of the batches in flight: a caller's segs array is only valid during the call and several batches may be
	// queued on the stream, so each call copies it into one of GYS_SEG_RING pinned host buffers (+ its own device buffer); a slot is
	// reused only after the kernels that read it have finished (event)
	struct SegSlot {
		gys_resp_seg *host = nullptr, *dev = nullptr;
		uint32_t cap = 0;
		hipEvent_t done = nullptr;
		// long segments cut into parts: the part descriptors (gys_resp_seg each)
		uint8_t *xhost = nullptr, *xdev = nullptr;
		uint64_t xcap = 0;
	} seg_ring[GYS_SEG_RING];
	uint32_t seg_next = 0;

	// reduce arena + last-window results
	uint8_t *arena = nullptr;
	bool own_arena = false;
	ArenaLayout al{};
	uint8_t *last = nullptr; // copy of the reduced arena of the last finished window (queries read this)
	uint32_t *last_act32 = nullptr;           // ACTIVE_CONN_STATS Count-Min pair the queries read: per-cell maximum over the tables of the
	unsigned long long *last_act64 = nullptr; // last GYS_ACT_RING windows (a partha reports every 15 s, a window is 5 s; k_act_latch)
	uint32_t *ring_act32 = nullptr, *act_live = nullptr;
	unsigned long long *ring_act64 = nullptr;
	uint32_t epoch = 1;      // current window number (0 = never)
	bool prepared = false;
	uint32_t *d_epoch = nullptr; // device copy of `epoch` for the captured window graph
	// gys_window_close: the WHOLE single-rank window boundary as one hipGraph per (registry shape, which folds are due)
	struct CloseGraph {
		hipGraph_t g = nullptr;
		hipGraphExec_t x = nullptr;
		uint64_t shape = 0;
		int state = 0; // 0 not tried, 1 usable, -1 capture unavailable
	} close_graph[4];
	uint64_t close_graph_launches = 0;
	bool have_last = false;

	// staging for host-buffer ingest
	// host-pointer boundary (gys_ingest_resp_events / _tcp_conn / _listener_state): a ring of pinned host + device staging
	// buffers.  A call copies the caller's records into a slot's pinned buffer (the caller's buffer is free when the call returns --
	// the reference's pone points into the L1 receive buffer, valid only for the call), enqueues ONE H2D copy + the kernels and
	// returns without waiting for the GPU; a slot is reused after the event recorded behind its kernels has fired.  Up to
	// MAX_L2_MISC_THREADS = 16 L2 threads (server/gy_mconnhdlr.h:60) call concurrently: slots are handed out under stage_mu, the
	// memcpy into pinned memory runs outside any lock, the enqueue (stream order, shared flags) under enq_mu.
	struct Stage {
		uint8_t *h = nullptr, *d = nullptr;
		uint64_t cap = 0;
		hipEvent_t done = nullptr;
		hipEvent_t copied = nullptr; // (record batches: the slot's H2D copy on the copy stream)
	};
	static constexpr int NSTAGE = 16;
	Stage stage[NSTAGE];
	std::deque<int> stage_free; // handed out OLDEST FIRST (pop_front / push_back): a slot's event has normally fired long before the slot comes round again
	std::mutex stage_mu, enq_mu;
	std::condition_variable stage_cv;
	std::atomic<uint64_t> stage_waits{0}; // times a host-pointer call found its ring slot still in flight and waited for the GPU
	// Submission queue of gys_ingest_resp_events ("group commit", SURVEY 8b second option): the calls of all L2 threads are concatenated
	// into ONE pinned batch -- a segment per call -- and handed to run_resp_batch together, so that the ~12 launches of a response batch
	// are paid once per submission instead of once per 65536-event call.  A caller reserves its place under rq.mu, copies outside any
	// lock, and whoever finds no submission in flight submits what has accumulated (a lone caller submits its own call at once: no added
	// latency; while GYS_RQ_INFLIGHT submissions are still executing on the GPU, further calls accumulate and go out together as soon as
	// one of them has finished -- the GPU always has the next submission queued behind the running one, and the fixed launches are
	// amortised exactly when the GPU is the bottleneck).  A host appears at most once per batch (its keys
	// see their per-call value multisets in call order: the digests stay bit-identical to per-call ingestion); batches are submitted in
	// the order they were sealed.
	struct RespBatch {
		uint8_t *h = nullptr, *d = nullptr;
		uint64_t cap_events = 0, fill = 0;
		hipEvent_t done = nullptr;
		hipEvent_t copied = nullptr; // the batch's H2D copy on the copy stream (the engine stream waits for it before the batch's kernels)
		std::vector<gys_resp_seg> segs;
		uint32_t writers = 0;
		bool sealed = false;
	};
	struct RespQ {
		static constexpr int NB = 6;
		std::mutex mu;
		std::condition_variable cv;
		RespBatch b[NB];
		std::deque<int> free, sealed, inflight; // sealed: awaiting submission, oldest first; inflight: submitted, GPU not done yet
		int open = -1;
		bool submitting = false;
		int async_rc = 0;             // first error of a submission made on behalf of other callers; surfaces at the next call
		std::string async_err;
		uint64_t calls = 0, submissions = 0;
		std::vector<uint32_t> host_stamp; // host -> stamp of the open batch it is in
		uint32_t stamp = 0;
		// the tail of a burst: calls that found GYS_RQ_INFLIGHT submissions on the GPU left their events in the open batch, and after the
		// burst nobody calls again -- a flusher thread (started with the first queued call) submits that batch once a submission has retired
		std::condition_variable fcv;
		std::thread flusher;
		bool flusher_on = false, stop = false;
		uint64_t tail_flushes = 0;
	} rq;
	// Submission queues of the host-pointer TCP_CONN_NOTIFY and LISTENER_STATE_NOTIFY calls (round 4; the same group commit as RespQ).
	// Through the 16-slot staging ring every partha message cost its own H2D copy, two event records, a stream wait and a launch -- five
	// runtime calls of ~5 us each under enq_mu for a 2048-record (0.57 MB) or 512-record (45 KB) message: 69 M connection records/s and
	// 24 M listener records/s from 16 threads, a third and a twentieth of what the link carries, with the ring wrapping all the time
	// (gys_counters.stage_waits).  Now the messages of all threads are appended to ONE pinned batch -- records, then an offset (and, for
	// listener states, the sender's host slot) per record -- and a batch is copied and launched once.  A lone caller's message still goes
	// out at once; messages accumulate only while GYS_RQ_INFLIGHT submissions are on the GPU.  Records keep the order in which the calls
	// reserved their place, so "the last record of a listener wins" holds across the messages of a batch as it does across calls.
	struct RecBatch {
		uint8_t *h = nullptr, *d = nullptr; // [records: cap_bytes][offset per record: u32 x cap_recs][host slot per record: u32 x cap_recs]
		uint64_t cap_bytes = 0, fill = 0;
		uint32_t cap_recs = 0, nrec = 0;
		hipEvent_t done = nullptr, copied = nullptr;
		uint32_t writers = 0;
	};
	struct RecQ {
		static constexpr int NB = 4;
		bool conn = false; // TCP_CONN_NOTIFY (else LISTENER_STATE_NOTIFY)
		std::mutex mu;
		std::condition_variable cv, fcv;
		RecBatch b[NB];
		std::deque<int> free, sealed, inflight;
		int open = -1;
		bool submitting = false;
		int async_rc = 0;
		std::string async_err;
		uint64_t calls = 0, submissions = 0, tail_flushes = 0;
		std::thread flusher;
		bool flusher_on = false, stop = false;
	} cq[2]; // [0] connections, [1] listener states
	uint8_t *dev_staging = nullptr;
	uint64_t dev_staging_bytes = 0;
	uint32_t *dev_offsets = nullptr;
	uint32_t dev_offsets_cap = 0;
	// host -> member services (slot order) on the device, rebuilt when the registry changed (roll-ups, all-hosts top-N)
	uint32_t *csr_off = nullptr, *csr_mem = nullptr;
	uint64_t csr_stamp = ~0ull;
	std::vector<uint16_t> svc_port_h; // listener port per service slot (web_curr_top_listeners "port")
	uint32_t *topn_slot = nullptr;
	uint64_t *topn_metric = nullptr;
	// filtered multi-host listener-state query (gys_svcquery.hpp): scratch, grow-only
	unsigned long long *q_cand_key = nullptr, *q_out_keys = nullptr;
	uint32_t *q_cand_slot = nullptr, *q_misc = nullptr, *q_host_mask = nullptr, *q_slot_list = nullptr;
	int32_t *q_set = nullptr;
	uint8_t *q_out_rows = nullptr;
	long long *q_acc = nullptr;
	unsigned long long *q_cnt = nullptr;
	uint64_t q_cand_cap = 0, q_slot_cap = 0, q_out_cap = 0, q_okeys_cap = 0, q_mask_cap = 0, q_set_cap = 0, q_acc_cap = 0, q_cnt_cap = 0, q_slist_cap = 0;
	float *dev_pcts = nullptr;
	float *zipf_cdf = nullptr;
	uint32_t zipf_n = 0, zipf_milli = 0;

	// wire front-end scratch (grow-only)
	uint64_t wire_slots_cap = 0;
	uint32_t *wire_jump[2] = {nullptr, nullptr}, *wire_cnt = nullptr, *wire_rank = nullptr, *wire_bsums = nullptr, *wire_status = nullptr;
	uint8_t *wire_mark = nullptr, *wire_flags = nullptr;
	WireMsg *wire_msgs = nullptr;
	uint32_t wire_msgs_cap = 0;

	// multi-level windows (cfg.enable_levels; kernels: "multi-level windows" in gys_kernels.hpp)
	gys_hist_rec *lvl_snap = nullptr; // [2][GYS_LEVEL_RING][max_services] cumulative records at the last start of every ring bucket
	gys_hist_rec *lvl_last = nullptr; // [max_services] the window closed last (level 0)
	// lazily folded records (t-digest on): hist_win and lvl_last change places at every close instead of a copy; lvl_last[slot] then is the service's
	// record of window lvl_last_tag[slot] (written by the close's fold pass) and counts only when that is lvl_last_epoch, the window closed last
	uint32_t *lvl_last_tag = nullptr;
	// roll-up digests (gys_rollup.hpp): the groups' value bins (grows), the hosts' member lists on the device (rebuilt when services were registered)
	unsigned long long *rb_bins = nullptr;
	size_t rb_bins_groups = 0;
	uint32_t *rb_host_members = nullptr;
	RollupChunk *rb_host_chunks = nullptr;
	uint32_t rb_host_nsvc = ~0u, rb_host_nh = ~0u, rb_host_nchunks = 0;
	uint8_t *svc_bithist = nullptr; // [max_services][2] TCP_LISTENER::issue_bit_hist_ / high_resp_bit_hist_ (gys_decide_listener_state_dev; allocated on first use)
	uint32_t lvl_last_epoch = 0;
	int64_t *lvl_first = nullptr;     // [max_services] time (s) of the service's first window close (firstTime_ of its series), 0: none yet
	gys_hist_rec *qps_hist = nullptr, *act_hist = nullptr; // per-service QPS_HISTOGRAM / ACTIVE_CONN_HISTOGRAM
	int64_t lvl_t_last = -1;          // close time (s) of the last window, -1: none yet

	bool profile = false;
	std::map<std::string, ProfEntry> prof;
};

namespace {

// ------------------------------------------------------------------------------------------------ launch + profiling
struct ProfScope {
	gys_ctx *c;
	ProfEntry *e = nullptr;
	hipEvent_t a = nullptr, b = nullptr;
	hipStream_t st;
	ProfScope(gys_ctx *ctx, const char *name, hipStream_t stream = nullptr) : c(ctx), st(stream ? stream : ctx->stream)
	{
		if (!c->profile) return;
		e = &c->prof[name];
		if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
			e = nullptr;
			return;
		}
		hipEventRecord(a, st);
	}
	~ProfScope()
	{
		if (!e) return;
		hipEventRecord(b, st);
		e->pending.emplace_back(a, b);
		e->launches++;
	}
};

void prof_resolve(gys_ctx *c)
{
	for (auto &kv : c->prof) {
		for (auto &pr : kv.second.pending) {
			float ms = 0;
			hipEventSynchronize(pr.second);
			if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) kv.second.total_ms += ms;
			hipEventDestroy(pr.first);
			hipEventDestroy(pr.second);
		}
		kv.second.pending.clear();
	}
}

inline uint32_t grid_for(uint64_t n, uint32_t block, uint32_t cap)
{
	uint64_t g = (n + block - 1) / block;
	if (g < 1) g = 1;
	if (g > cap) g = cap;
	return (uint32_t)g;
}

template <typename T>
int dev_alloc(T **p, uint64_t count, bool zero = true)
{
	if (count == 0) count = 1;
	HIPCHK(hipMalloc((void **)p, count * sizeof(T)));
	if (zero) HIPCHK(hipMemset(*p, 0, count * sizeof(T)));
	return GYS_OK;
}

// ---- staging ring of the host-pointer boundary (gys_ctx::Stage)
int stage_acquire(gys_ctx *c, uint64_t bytes, int *idx)
{
	int i;
	{
		std::unique_lock<std::mutex> lk(c->stage_mu);
		c->stage_cv.wait(lk, [&] { return !c->stage_free.empty(); });
		i = c->stage_free.front();
		c->stage_free.pop_front();
	}
	gys_ctx::Stage &st = c->stage[i];
	// the calling (L2) thread's current device is whatever it used last -- 0 for a fresh thread: the slot's buffers, the copy and the
	// kernels must live on the context's device
	hipError_t e = hipSetDevice(c->device);
	if (e == hipSuccess && !st.done) e = hipEventCreateWithFlags(&st.done, hipEventDisableTiming);
	if (e == hipSuccess && !st.copied) e = hipEventCreateWithFlags(&st.copied, hipEventDisableTiming);
	if (e == hipSuccess && hipEventQuery(st.done) == hipErrorNotReady) c->stage_waits++;
	(void)hipGetLastError();
	if (e == hipSuccess) e = hipEventSynchronize(st.done); // the kernels that read this slot last time are done (no-op unless the ring wrapped)
	if (e == hipSuccess && bytes > st.cap) {
		if (st.h) (void)hipHostFree(st.h);
		if (st.d) (void)hipFree(st.d);
		st.h = st.d = nullptr;
		st.cap = align_up(std::max<uint64_t>(bytes, 1u << 20), 4096);
		e = hipHostMalloc((void **)&st.h, st.cap, hipHostMallocDefault);
		if (e == hipSuccess) e = hipMalloc((void **)&st.d, st.cap);
		if (e != hipSuccess) st.cap = 0;
	}
	if (e != hipSuccess) {
		set_err("staging slot: %s", hipGetErrorString(e));
		std::lock_guard<std::mutex> lk(c->stage_mu);
		c->stage_free.push_back(i);
		c->stage_cv.notify_one();
		return GYS_ERR_HIP;
	}
	*idx = i;
	return GYS_OK;
}

void stage_release(gys_ctx *c, int idx)
{
	std::lock_guard<std::mutex> lk(c->stage_mu);
	c->stage_free.push_back(idx);
	c->stage_cv.notify_one();
}

int ensure_staging(gys_ctx *c, uint64_t bytes, uint32_t nrec)
{
	if (bytes > c->dev_staging_bytes) {
		if (c->dev_staging) {
			HIPCHK(hipStreamSynchronize(c->stream));
			HIPCHK(hipFree(c->dev_staging));
		}
		c->dev_staging_bytes = align_up(std::max<uint64_t>(bytes, 1u << 20), 4096);
		HIPCHK(hipMalloc((void **)&c->dev_staging, c->dev_staging_bytes));
	}
	if (nrec > c->dev_offsets_cap) {
		if (c->dev_offsets) {
			HIPCHK(hipStreamSynchronize(c->stream));
			HIPCHK(hipFree(c->dev_offsets));
		}
		c->dev_offsets_cap = std::max<uint32_t>(nrec, 4096);
		HIPCHK(hipMalloc((void **)&c->dev_offsets, (uint64_t)c->dev_offsets_cap * 4));
	}
	return GYS_OK;
}

int lookup_host(gys_ctx *c, const uint8_t machine_id[16], uint32_t *slot)
{
	auto it = c->host_map.find(to_machid(machine_id));
	if (it == c->host_map.end()) {
		set_err("unknown machine id");
		return GYS_ERR_NOTFOUND;
	}
	*slot = it->second;
	return GYS_OK;
}

int check_owner(gys_ctx *c, const uint8_t machine_id[16])
{
	if (c->cfg.nranks > 1 && gys_shard_of(machine_id, c->cfg.nranks) != c->cfg.rank) {
		set_err("host belongs to rank %u", gys_shard_of(machine_id, c->cfg.nranks));
		return GYS_ERR_NOT_OWNER;
	}
	return GYS_OK;
}

uint32_t next_pow2(uint64_t v)
{
	uint64_t p = 1;
	while (p < v) p <<= 1;
	return (uint32_t)p;
}

// ---- per-host listener sub-tables (host mirror + device pools), used by k_resp_host
inline uint64_t host_key48(uint32_t netns, uint16_t port) { return ((uint64_t)netns << 16) | (uint64_t)port; }

int64_t host_tbl_find(const HostListeners &hl, uint64_t key48)
{
	if (hl.tbl.empty()) return -1;
	const uint32_t mask = (uint32_t)hl.tbl.size() - 1;
	uint32_t h = host_tbl_slot(host_tbl_hash(key48), mask);
	for (uint32_t probes = 0; probes <= mask; ++probes) {
		const uint64_t e = hl.tbl[h];
		if (e == GYS_HOST_TBL_EMPTY) return -1;
		if ((e >> 16) == key48) return (int64_t)h;
		h = (h + 1) & mask;
	}
	return -1;
}

// `local`: a local index, or GYS_LOCAL_GROUP | first candidate of the key's chain
void host_tbl_put(HostListeners &hl, uint64_t key48, uint32_t local)
{
	const uint32_t mask = (uint32_t)hl.tbl.size() - 1;
	uint32_t h = host_tbl_slot(host_tbl_hash(key48), mask);
	while (hl.tbl[h] != GYS_HOST_TBL_EMPTY) h = (h + 1) & mask;
	hl.tbl[h] = (key48 << 16) | (uint64_t)local;
}

inline uint32_t host_part_of(uint64_t key48, uint32_t nparts) { return host_tbl_part(host_tbl_hash(key48), nparts - 1u); }

// capacity of a sub-table of n listeners: a QUARTER full while that is at most GYS_HOST_TBL_SPARSE entries (32 KiB of LDS; 97 % of the
// events then find their listener in the two entries k_resp_host reads at once), half full above (up to 8192 entries for 4096 listeners)
#define GYS_HOST_TBL_SPARSE 4096u
inline uint32_t host_tbl_capacity(size_t n)
{
	static const uint32_t sparse = [] { const char *e = getenv("GYS_TBL_SPARSE"); return e ? (uint32_t)atoi(e) : GYS_HOST_TBL_SPARSE; }(); // (A/B: 0 = half full always)
	const uint32_t c4 = next_pow2(std::max<uint64_t>(16, (uint64_t)n * 4));
	return c4 <= sparse ? c4 : std::max<uint32_t>(std::min<uint32_t>(sparse, c4), next_pow2(std::max<uint64_t>(16, (uint64_t)n * 2)));
}

// (re)uploads one sub-table, its slot list and its descriptor hdesc[desc]; regions only ever grow, an outgrown region is abandoned in
// the pool (geometric growth: the abandoned total stays below the final size, which is what the pool capacity accounts for).
// Returns false when the pools are exhausted.
int host_tbl_upload(gys_ctx *c, HostListeners &hl, uint32_t desc, uint32_t part, uint32_t nparts, uint32_t cand_off, bool *ok)
{
	*ok = true;
	if (!hl.on_device || hl.tbl.size() > hl.tbl_cap) {
		if (c->htbl_used + hl.tbl.size() > c->htbl_cap || c->hlst_used + hl.tbl.size() / 2 > c->hlst_cap) {
			*ok = false;
			return GYS_OK;
		}
		hl.tbl_off = (uint32_t)c->htbl_used;
		hl.tbl_cap = (uint32_t)hl.tbl.size();
		c->htbl_used += hl.tbl.size();
		hl.lst_off = (uint32_t)c->hlst_used;
		c->hlst_used += hl.tbl.size() / 2;
		hl.on_device = true;
	}
	HIPCHK(hipMemcpyAsync(c->htbl + hl.tbl_off, hl.tbl.data(), hl.tbl.size() * 8, hipMemcpyHostToDevice, c->stream));
	if (!hl.slots.empty()) HIPCHK(hipMemcpyAsync(c->hlst + hl.lst_off, hl.slots.data(), hl.slots.size() * 4, hipMemcpyHostToDevice, c->stream));
	const HostDesc hd{hl.tbl_off, (uint32_t)hl.tbl.size() - 1, (uint32_t)hl.slots.size(), hl.lst_off, part, nparts - 1u, cand_off, 0u};
	HIPCHK(hipMemcpyAsync(c->hdesc + desc, &hd, sizeof(hd), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream)); // hd is a stack object
	return GYS_OK;
}

// the host's candidate region: allocated in the pool on first use (the pool itself on the first listener that needs it: a context whose
// listeners are all alone on their key and any-address never pays for it), re-allocated (twice the size) when outgrown
int host_cands_upload(gys_ctx *c, HostListeners &hl)
{
	if (hl.cands.empty()) return GYS_OK;
	if (hl.cands.size() > hl.cand_cap) {
		const uint64_t want = std::max<uint64_t>(16, next_pow2(hl.cands.size()));
		// round 6: the pool grows geometrically from 4 096 records (128 KB) to its bound of 4 x max_services + 64 x max_hosts
		// records instead of taking all of that -- 1.3 GB at 10^7 services -- on the first bound-address listener.  Regions are offsets
		// into the pool: they survive the move; kernels take the pool's address at launch.
		const uint64_t bound = 4ull * c->cfg.max_services + 64ull * c->cfg.max_hosts;
		if (c->cand_used + want > bound) {
			set_err("listener candidate pool exhausted");
			return GYS_ERR_NOMEM;
		}
		if (c->cand_used + want > c->cand_cap) {
			uint64_t ncap = std::max<uint64_t>(c->cand_cap * 2, 4096);
			while (ncap < c->cand_used + want) ncap *= 2;
			ncap = std::min(ncap, bound);
			ListenerCand *np = nullptr;
			HIPCHK(hipMalloc((void **)&np, ncap * sizeof(ListenerCand)));
			if (c->cand_pool) {
				HIPCHK(hipMemcpyAsync(np, c->cand_pool, c->cand_used * sizeof(ListenerCand), hipMemcpyDeviceToDevice, c->stream));
				HIPCHK(hipStreamSynchronize(c->stream)); // (every launch that reads the old pool is behind us on this stream)
				HIPCHK(hipFree(c->cand_pool));
			}
			c->cand_pool = np;
			c->cand_cap = ncap;
		}
		hl.cand_off = (uint32_t)c->cand_used;
		hl.cand_cap = (uint32_t)want;
		c->cand_used += want;
	}
	HIPCHK(hipMemcpyAsync(c->cand_pool + hl.cand_off, hl.cands.data(), hl.cands.size() * sizeof(ListenerCand), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return GYS_OK;
}

int host_lst_upload(gys_ctx *c, uint32_t host)
{
	HostListeners &hl = c->host_lst[host];
	if (hl.overflow) return GYS_OK;
	bool ok = true;
	if (hl.sub.empty()) {
		const int rc = host_tbl_upload(c, hl, host, 0, 1, hl.cand_off, &ok);
		if (rc) return rc;
	} else {
		for (uint32_t p = 0; p < hl.sub.size() && ok; ++p) {
			const int rc = host_tbl_upload(c, hl.sub[p], c->cfg.max_hosts + hl.sub_desc + p, p, (uint32_t)hl.sub.size(), hl.cand_off, &ok);
			if (rc) return rc;
		}
	}
	if (!ok) hl.overflow = true; // pools exhausted: this host keeps working through the general pipeline
	return GYS_OK;
}

// one listener into one sub-table (grown and rehashed at half full).  `top`: the host (holder of the chains): with chains around, a
// rehash has to know which keys' entries name a candidate region -- the whole host is rebuilt at the end of the registration call instead
void host_tbl_insert(HostListeners &top, HostListeners &t, uint64_t key48, uint32_t slot)
{
	const uint32_t local = (uint32_t)t.slots.size();
	t.slots.push_back(slot);
	t.keys.push_back(key48);
	// (the plain entries are kept current in any case -- later listeners of the same call look their key up in them; entries of keys with
	// candidates are put right by host_rebuild at the end of the call)
	if (host_tbl_capacity(t.slots.size()) > t.tbl.size()) {
		t.tbl.assign(host_tbl_capacity(t.slots.size()), GYS_HOST_TBL_EMPTY);
		for (uint32_t l = 0; l < t.keys.size(); ++l) host_tbl_put(t, t.keys[l], l);
		if (!top.chain.empty()) top.dirty = true;
	} else {
		host_tbl_put(t, key48, local);
	}
}

// all tables of a host and its candidate records from the locals and the chains (after a chain changed)
void host_rebuild(HostListeners &hl)
{
	hl.cands.clear();
	hl.cand_first.clear();
	auto one = [&](HostListeners &t) {
		t.tbl.assign(std::max<size_t>(t.tbl.size(), host_tbl_capacity(t.slots.size())), GYS_HOST_TBL_EMPTY);
		std::unordered_map<uint32_t, uint32_t> local_of; // slot -> local, for the locals of keys with candidates
		for (uint32_t l = 0; l < t.keys.size(); ++l)
			if (hl.chain.count(t.keys[l])) local_of[t.slots[l]] = l;
		for (uint32_t l = 0; l < t.keys.size(); ++l) {
			const uint64_t key48 = t.keys[l];
			auto ch = hl.chain.find(key48);
			if (ch == hl.chain.end()) {
				host_tbl_put(t, key48, l);
				continue;
			}
			if (hl.cand_first.count(key48)) continue; // (the key's entry was made when its first local came by)
			const uint32_t first = (uint32_t)hl.cands.size();
			hl.cand_first[key48] = first;
			for (size_t k = 0; k < ch->second.size(); ++k) {
				const HostListeners::ChainEnt &e = ch->second[k];
				ListenerCand cd{};
				cd.ip32 = e.a.ip32;
				cd.flags = (e.a.any ? 1u : 0u) | (k + 1 == ch->second.size() ? 2u : 0u);
				auto lo = local_of.find(e.slot);
				cd.local = lo != local_of.end() ? lo->second : 0u;
				cd.slot = e.slot;
				memcpy(cd.ip128, e.a.ip128, 16);
				hl.cands.push_back(cd);
			}
			host_tbl_put(t, key48, GYS_LOCAL_GROUP | first);
		}
	};
	if (hl.overflow) { // (no sub-tables: only the records, for the global table)
		for (auto &kv : hl.chain) {
			hl.cand_first[kv.first] = (uint32_t)hl.cands.size();
			for (size_t k = 0; k < kv.second.size(); ++k) {
				ListenerCand cd{};
				cd.ip32 = kv.second[k].a.ip32;
				cd.flags = (kv.second[k].a.any ? 1u : 0u) | (k + 1 == kv.second.size() ? 2u : 0u);
				cd.slot = kv.second[k].slot;
				memcpy(cd.ip128, kv.second[k].a.ip128, 16);
				hl.cands.push_back(cd);
			}
		}
	} else if (hl.sub.empty()) {
		one(hl);
	} else {
		for (HostListeners &t : hl.sub) one(t);
	}
	hl.dirty = false;
}

// cuts the host's listeners into nparts sub-tables (from one table, or from fewer parts); false: no descriptor room / too many parts
bool host_lst_repartition(gys_ctx *c, HostListeners &hl, uint32_t nparts)
{
	if (nparts > GYS_HOST_MAX_PARTS || c->hdesc_ext_used + nparts > c->hdesc_ext_cap) return false;
	std::vector<std::pair<uint64_t, uint32_t>> all;
	if (hl.sub.empty()) {
		for (uint32_t l = 0; l < hl.keys.size(); ++l) all.emplace_back(hl.keys[l], hl.slots[l]);
	} else {
		for (const HostListeners &t : hl.sub)
			for (uint32_t l = 0; l < t.keys.size(); ++l) all.emplace_back(t.keys[l], t.slots[l]);
	}
	hl.tbl.clear();
	hl.slots.clear();
	hl.keys.clear();
	hl.sub.assign(nparts, HostListeners{});
	hl.sub_desc = c->hdesc_ext_used; // (the descriptors of the previous cut are abandoned, like outgrown table regions)
	c->hdesc_ext_used += nparts;
	for (const auto &kv : all) host_tbl_insert(hl, hl.sub[host_part_of(kv.first, nparts)], kv.first, kv.second);
	for (HostListeners &t : hl.sub)
		if (t.tbl.empty()) t.tbl.assign(16, GYS_HOST_TBL_EMPTY);
	if (!hl.chain.empty()) hl.dirty = true;
	return true;
}

// GY_IP_ADDR of a registered listener (set_ip(uint32_t) / set_ip(unsigned __int128), common/gy_common_inc.h:10673-10692)
inline HostListeners::LAddr listener_addr(const gys_listener_info &li)
{
	HostListeners::LAddr a;
	a.any = li.is_any_ip != 0;
	if (a.any) return a;
	if (li.addr_is_v6) {
		memcpy(a.ip128, li.addr, 16);
		a.ip32 = ip6_embedded_v4(a.ip128);
	} else {
		memcpy(&a.ip32, li.addr, 4);
	}
	return a;
}
inline bool laddr_equal(const HostListeners::LAddr &x, const HostListeners::LAddr &y) // GY_IP_ADDR::operator== (common/gy_common_inc.h:10629-10636)
{
	return (x.ip32 | y.ip32) ? x.ip32 == y.ip32 : memcmp(x.ip128, y.ip128, 16) == 0;
}

// the local index that holds `slot` in the table of key48 gets `new_slot` (a listener replaced in place)
void host_rebind_local(HostListeners &hl, uint64_t key48, uint32_t slot, uint32_t new_slot)
{
	if (hl.overflow) return;
	HostListeners &t = hl.sub.empty() ? hl : hl.sub[host_part_of(key48, (uint32_t)hl.sub.size())];
	for (uint32_t l = 0; l < t.slots.size(); ++l)
		if (t.slots[l] == slot && t.keys[l] == key48) {
			t.slots[l] = new_slot;
			return;
		}
}

int host_lst_add(gys_ctx *c, uint32_t host, const gys_listener_info *arr, uint32_t n, uint32_t first_slot)
{
	HostListeners &hl = c->host_lst[host];
	auto new_local = [&](uint64_t key48, uint32_t slot) { // false: the host left the LDS path
		if (hl.overflow) return false;
		HostListeners *t = hl.sub.empty() ? &hl : &hl.sub[host_part_of(key48, (uint32_t)hl.sub.size())];
		while (t->slots.size() >= GYS_HOST_MAX_LOCAL) { // the (part of the) host is full: twice the parts
			if (!host_lst_repartition(c, hl, hl.sub.empty() ? 2u : (uint32_t)hl.sub.size() * 2u)) {
				// beyond what the LDS sub-tables take: batches with this host use the general pipeline.  The keys of its any-address
				// listeners move to a flat map (what a later registration on the same key has to find)
				hl.overflow = true;
				auto take = [&](const HostListeners &s) {
					for (uint32_t l = 0; l < s.keys.size(); ++l)
						if (!hl.chain.count(s.keys[l])) hl.okey[s.keys[l]] = s.slots[l];
				};
				if (hl.sub.empty()) take(hl);
				else for (const HostListeners &s : hl.sub) take(s);
				return false;
			}
			t = &hl.sub[host_part_of(key48, (uint32_t)hl.sub.size())];
		}
		host_tbl_insert(hl, *t, key48, slot);
		return true;
	};
	for (uint32_t i = 0; i < n; ++i) {
		const uint64_t key48 = host_key48(arr[i].netns, arr[i].port);
		const uint32_t slot = first_slot + i;
		const HostListeners::LAddr a = listener_addr(arr[i]);
		auto ch = hl.chain.find(key48);
		if (ch != hl.chain.end()) {
			// insert_or_replace (common/gy_socket_stat.cc:1372, :7779) under operator==(listener, NS_IP_PORT) (gy_socket_stat.h:708-714): the first
			// listener of the key that is any-address or bound to the new one's address is replaced in place, else the new one goes last
			bool replaced = false;
			for (HostListeners::ChainEnt &e : ch->second) {
				if (e.a.any || laddr_equal(e.a, a)) {
					host_rebind_local(hl, key48, e.slot, slot);
					e.slot = slot;
					e.a = a;
					replaced = true;
					break;
				}
			}
			if (!replaced) {
				ch->second.push_back(HostListeners::ChainEnt{slot, a});
				if (!hl.overflow) new_local(key48, slot);
			}
			hl.dirty = true;
			continue;
		}
		// the key has no candidates: no listener yet, or one any-address listener (which every new listener of the key replaces)
		uint32_t old_slot = GYS_NOSLOT;
		if (hl.overflow) {
			auto it = hl.okey.find(key48);
			if (it != hl.okey.end()) old_slot = it->second;
		} else {
			HostListeners *t = hl.sub.empty() ? &hl : &hl.sub[host_part_of(key48, (uint32_t)hl.sub.size())];
			const int64_t pos = host_tbl_find(*t, key48);
			if (pos >= 0) {
				uint32_t &ls = t->slots[(uint32_t)(t->tbl[(size_t)pos] & (GYS_LOCAL_GROUP - 1u))];
				old_slot = ls;
				ls = slot; // (re-registration of a listener tuple rebinds it to the newest slot)
			}
		}
		if (old_slot != GYS_NOSLOT) {
			if (hl.overflow && a.any) hl.okey[key48] = slot;
		} else if (!hl.overflow) {
			new_local(key48, slot);
			if (hl.overflow && a.any) hl.okey[key48] = slot; // (the host left the LDS path with this very listener)
		} else if (a.any) {
			hl.okey[key48] = slot;
		}
		if (!a.any) { // bound to an address: the key gets candidates
			hl.okey.erase(key48);
			hl.chain[key48].push_back(HostListeners::ChainEnt{slot, a});
			hl.dirty = true;
		}
	}
	if (hl.dirty) {
		host_rebuild(hl);
		const int rc = host_cands_upload(c, hl);
		if (rc) return rc;
		// the global table (general pipeline): a key with candidates points at its chain's records
		std::vector<uint64_t> kv;
		for (const auto &cf : hl.cand_first) {
			kv.push_back(listener_key(host, (uint32_t)(cf.first >> 16), (uint16_t)(cf.first & 0xFFFFu)));
			kv.push_back((uint64_t)(GYS_SLOT_GROUP | (hl.cand_off + cf.second)));
		}
		if (!kv.empty()) {
			const uint32_t nk = (uint32_t)(kv.size() / 2);
			const int rs = ensure_staging(c, kv.size() * 8, 0);
			if (rs) return rs;
			HIPCHK(hipMemcpyAsync(c->dev_staging, kv.data(), kv.size() * 8, hipMemcpyHostToDevice, c->stream));
			hipLaunchKernelGGL(k_table_set, dim3((nk + 255) / 256), dim3(256), 0, c->stream, c->lk_tbl, (const uint64_t *)c->dev_staging, nk);
			HIPCHK(hipStreamSynchronize(c->stream));
		}
	}
	return host_lst_upload(c, host);
}

inline DigestP digest_params(gys_ctx *c)
{
	DigestP d{};
	d.td_sum = c->td_sum;
	d.td_cnt = c->td_cnt;
	d.td_meta = c->td_meta;
	d.td_minmax = c->td_minmax;
	d.td_pend = c->td_pend;
	d.td_cur = c->td_cur;
	d.pcap = c->pcap;
	d.pend_cap = c->pend_cap;
	d.nsvc = c->nsvc;
	d.staged = c->staged;
	d.hist_win = c->hist_win;
	d.hist_all = c->hist_all;
	d.bitmap = c->bitmap;
	return d;
}

// lazy fold (t-digest on): the records of services [first, first + n) are brought up to date with their buffered values before
// anything reads them ("per-key value buffers" in gys_kernels.hpp)
int fold_range(gys_ctx *c, uint32_t first, uint32_t n)
{
	if (!c->cfg.enable_tdigest || !n) return GYS_OK;
	FoldP f{};
	f.d = digest_params(c);
	f.first = first;
	f.n = n;
	ProfScope ps(c, "fold");
	const uint32_t nchunks = (n + 63u) / 64u;
	hipLaunchKernelGGL(k_fold, dim3(std::min<uint32_t>((nchunks + 3u) / 4u, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, f);
	HIPCHK(hipGetLastError());
	return GYS_OK;
}

// window close with the 5-s level: every service's records are brought up to date; the same pass leaves the window each record in hist_win
// belongs to in lvl_last_tag and the services' first close time in lvl_first (level_roll then swaps hist_win and lvl_last)
int fold_close_levels(gys_ctx *c, int64_t tnow)
{
	FoldP f{};
	f.d = digest_params(c);
	f.first = 0;
	f.n = c->nsvc;
	f.last_tag = c->lvl_last_tag;
	f.first_sec = c->lvl_first;
	f.tnow = tnow;
	ProfScope ps(c, "fold");
	const uint32_t nchunks = (c->nsvc + 63u) / 64u;
	hipLaunchKernelGGL(k_fold, dim3(std::min<uint32_t>((nchunks + 3u) / 4u, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, f);
	HIPCHK(hipGetLastError());
	return GYS_OK;
}

template <int TPT, bool SHARED, bool SPILL, int MODE = 0>
void launch_resp_host(gys_ctx *c, uint32_t grid, size_t dyn, const RespHostP &hp)
{
	// (IPv6 events: every kept event takes the rolled general hash path anyway, so ONE instance -- the one with the per-service register code,
	// which checks svc_hll_p at run time there -- serves both configurations)
	if (!SPILL && (hp.svc_hll_p || MODE == 2)) hipLaunchKernelGGL((k_resp_host<TPT, SHARED, SPILL, !SPILL, MODE>), dim3(grid), dim3(GYS_RESP_THREADS(TPT)), dyn, c->stream, hp);
	else hipLaunchKernelGGL((k_resp_host<TPT, SHARED, SPILL, false, MODE>), dim3(grid), dim3(GYS_RESP_THREADS(TPT)), dyn, c->stream, hp);
}
// mode 1 / 2 (keys with candidates / IPv6 events) exist in the 1024-thread tile forms (16 384 or 8 192 events)
template <bool SHARED, bool SPILL>
void launch_resp_host_mode(gys_ctx *c, int mode, bool tpt16, uint32_t grid, size_t dyn, const RespHostP &hp)
{
	if (mode == 2) {
		if (tpt16) launch_resp_host<16, SHARED, SPILL, 2>(c, grid, dyn, hp);
		else launch_resp_host<8, SHARED, SPILL, 2>(c, grid, dyn, hp);
	} else {
		if (tpt16) launch_resp_host<16, SHARED, SPILL, 1>(c, grid, dyn, hp);
		else launch_resp_host<8, SHARED, SPILL, 1>(c, grid, dyn, hp);
	}
}

// dynamic LDS a k_resp_host launch may ask for: the CU's 160 KiB minus the instance's own static part (read from the code object, so that
// a kernel change cannot silently push a launch over the limit); *dyn_max ends as the smallest such room over the instances
template <int TPT, bool SHARED, bool SPILL, int MODE = 0>
hipError_t resp_host_lds_attr(uint32_t *dyn_max)
{
	for (int svchll = (MODE == 2 && !SPILL) ? 1 : 0; svchll < (SPILL ? 1 : 2); ++svchll) {
		const void *fn = svchll ? (const void *)k_resp_host<TPT, SHARED, SPILL, !SPILL, MODE> : (const void *)k_resp_host<TPT, SHARED, SPILL, false, MODE>;
		hipFuncAttributes fa{};
		hipError_t e = hipFuncGetAttributes(&fa, fn);
		if (e != hipSuccess) return e;
		const uint32_t room = (160u * 1024u - (uint32_t)fa.sharedSizeBytes) & ~255u;
		e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)room);
		if (e != hipSuccess) return e;
		*dyn_max = std::min(*dyn_max, room);
	}
	return hipSuccess;
}

// resp pipeline on a device-resident batch
// v6: the events are 48-byte tcp_ipv6_resp_event_t (handle_ipv6_resp_event, common/gy_socket_stat.cc:1535-1551)
int run_resp_batch(gys_ctx *c, const gys_resp_seg *segs_host, uint32_t nsegs, const void *d_ev, uint64_t n, bool v6 = false)
{
	if (n == 0) return GYS_OK;
	if (nsegs == 0 || !segs_host || segs_host[0].first_event != 0) {
		set_err("resp batch needs >= 1 segment starting at event 0");
		return GYS_ERR_INVAL;
	}
	const bool td = c->cfg.enable_tdigest != 0;
	if (td && n > c->cfg.max_batch_events) {
		set_err("batch of %llu events exceeds max_batch_events %llu", (unsigned long long)n, (unsigned long long)c->cfg.max_batch_events);
		return GYS_ERR_NOMEM;
	}
	if (n >= (1ull << 31)) {
		set_err("batch too large (u32 offsets)");
		return GYS_ERR_INVAL;
	}
	for (uint32_t s = 0; s < nsegs; ++s) {
		if (segs_host[s].host_slot >= c->hosts.size() || segs_host[s].first_event > n || (s && segs_host[s].first_event < segs_host[s - 1].first_event)) {
			set_err("bad resp segment %u", s);
			return GYS_ERR_INVAL;
		}
	}
	gys_ctx::SegSlot &slot = c->seg_ring[c->seg_next];
	c->seg_next = (c->seg_next + 1) % GYS_SEG_RING;
	if (!slot.done) HIPCHK(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
	HIPCHK(hipEventSynchronize(slot.done)); // the batch that used this slot GYS_SEG_RING calls ago has consumed it
	if (nsegs > slot.cap) {
		if (slot.host) HIPCHK(hipHostFree(slot.host));
		if (slot.dev) HIPCHK(hipFree(slot.dev));
		slot.host = slot.dev = nullptr;
		slot.cap = std::max<uint32_t>(nsegs, 1024);
		HIPCHK(hipHostMalloc((void **)&slot.host, (uint64_t)slot.cap * sizeof(gys_resp_seg), hipHostMallocDefault));
		HIPCHK(hipMalloc((void **)&slot.dev, (uint64_t)slot.cap * sizeof(gys_resp_seg)));
	}
	memcpy(slot.host, segs_host, (uint64_t)nsegs * sizeof(gys_resp_seg));
	// `reserved` is the engine's own part-descriptor index (descriptor + 1, written only into the engine-built segment lists below): whatever
	// the caller left in the field must never reach k_resp_host's hdesc[] lookup
	for (uint32_t s = 0; s < nsegs; ++s) ((gys_resp_seg *)slot.host)[s].reserved = 0;
	HIPCHK(hipMemcpyAsync(slot.dev, slot.host, (uint64_t)nsegs * sizeof(gys_resp_seg), hipMemcpyHostToDevice, c->stream));
	const gys_resp_seg *segs_dev = slot.dev;

	// ---- front end choice: host-local (one workgroup per host segment, LDS sub-table + tile-wise LDS counting sort straight into the
	// services' value buffers) when every segment is a distinct host with an LDS-sized listener table; otherwise the general front end
	bool host_local = td && c->cfg.resp_path != 1 && c->nsvc != 0, host_split = false, host_parts = false;
	uint32_t max_tbl = 16, max_l = 1;
	uint64_t max_len = 0, nwg = 0;
	bool cands = false; // some host of the batch has keys with candidates: the instances that resolve them by the server address
	if (host_local) {
		c->batch_stamp++;
		for (uint32_t s = 0; s < nsegs && host_local; ++s) {
			const uint32_t host = segs_host[s].host_slot;
			const HostListeners &hl = c->host_lst[host];
			cands = cands || !hl.chain.empty();
			const uint64_t len = (s + 1 < nsegs ? segs_host[s + 1].first_event : n) - segs_host[s].first_event;
			if (c->host_seen[host] == c->batch_stamp || hl.overflow) host_local = false;
			c->host_seen[host] = c->batch_stamp;
			max_len = std::max(max_len, len);
			if (hl.sub.empty()) {
				if (!hl.on_device) host_local = false;
				max_tbl = std::max<uint32_t>(max_tbl, (uint32_t)hl.tbl.size());
				max_l = std::max<uint32_t>(max_l, (uint32_t)hl.slots.size());
				nwg += 1;
			} else { // a many-listener host: one workgroup per part of its listeners
				host_parts = true;
				for (const HostListeners &t : hl.sub) {
					if (!t.on_device) host_local = false;
					max_tbl = std::max<uint32_t>(max_tbl, (uint32_t)t.tbl.size());
					max_l = std::max<uint32_t>(max_l, (uint32_t)t.slots.size());
				}
				nwg += hl.sub.size();
			}
		}
		// few hosts with long segments: one workgroup per segment would leave most of the chip idle, so the segments are cut into parts
		// of GYS_SPLIT_PART events (SHARED form: buffer space reserved with device atomics).  A workgroup walks ~0.35 G events/s.
		const double t_host = (double)((nwg + c->ncu - 1) / c->ncu) * (double)max_len / 0.35e9;
		const double t_split = (double)n * (double)std::max<uint64_t>(nwg, 1) / (double)std::max<uint32_t>(nsegs, 1) / 40.0e9 + 20e-6;
		if (host_local && max_len > GYS_SPLIT_PART && (c->cfg.resp_path == 3 || (c->cfg.resp_path == 0 && t_split < t_host))) host_split = true;
	}
	const int mode = v6 ? 2 : cands ? 1 : 0;
	const uint32_t nsvc = c->nsvc;
	uint32_t *cms32 = (uint32_t *)(c->arena + c->al.off_u32) + c->al.u32_cms;
	unsigned long long *ghist = (unsigned long long *)(c->arena + c->al.off_i64sum) + c->al.i64_ghist;
	long long *gmax = (long long *)(c->arena + c->al.off_i64max);
	if (td) {
		HIPCHK(hipMemsetAsync(c->merge_count, 0, 8 * 4, c->stream)); // merge / huge / fallback list lengths, run allocation cursor ([8] is the constant 1 of the query list)
		c->resp_dirty = true;
	}
	FinP fin{};
	if (td) {
		fin.td_cur = c->td_cur;
		fin.td_meta = c->td_meta;
		fin.nsvc = c->nsvc;
		fin.pcap = c->pcap;
		fin.pend_cap = c->pend_cap;
		fin.merge_fast = c->merge_fast;
		// Round 6: merges of 4097 .. 16 384 values (size class 2) through the streamed value-bin instance; GYS_CLASS2_HUGE: through the several-workgroup path as in rounds 3 - 5 (A/B)
		static const bool class2_huge = getenv("GYS_CLASS2_HUGE") != nullptr || getenv("GYS_OLD_HUGE") != nullptr; // (GYS_OLD_HUGE's kernel does not read the fallback list the instance hands over to)
		fin.class2_max = class2_huge ? 0u : GYS_MERGE_LDS_MAX;
		fin.epoch = c->epoch;
		fin.resp_win = c->resp_win;
		fin.list[FIN_CLASS0] = c->merge_list;
		fin.list[FIN_CLASS1] = c->merge_list1;
		fin.list[FIN_CLASS2] = c->merge_list2;
		fin.list[FIN_HUGE] = c->huge_list;
		fin.counts = c->merge_count;
		fin.td_run = c->td_run;
		fin.svc_host = c->svc_host;
		fin.host_spill = c->host_spill;
		fin.counters = c->counters;
		fin.staged_cap = (uint32_t)std::min<uint64_t>(c->staged_cap, 0xFFFFFFFFull);
	}
	// predicted runs: only for the host-local front end, and only when the batch is large enough for a key to overflow a buffer at all
	const bool pre = td && host_local && c->prespill && n > (uint64_t)c->pcap - c->pend_cap;
	if (td && c->prespill) {
		fin.td_run0 = c->td_run0;
		fin.td_run1 = c->td_run1;
		fin.td_prevm = c->td_prevm;
		fin.hot = c->pre_hot;
		fin.hot_wr = pre ? ((c->pre_seq & 1u) ^ 1u) : (c->pre_seq & 1u); // the word the NEXT k_prespill reads (this batch's own one, if any, reads the other)
		fin.append_list = c->append_list;
		fin.append_cap = (uint32_t)c->append_cap;
		fin.td_pend = c->td_pend;
		fin.staged = c->staged;
		HIPCHK(hipMemsetAsync(c->merge_count + FIN_APPEND, 0, 4 * 4, c->stream)); // ([12]: the append list's length; [14 .. 15]: the predicted runs' 64-bit reservation counter)
	}
	RespHostP hp{};
	uint32_t hgrid = 0;
	size_t dyn = 0;
	// tile form: 512 threads x 12 events (6144-event tiles) when the batch's tables leave room for TWO such workgroups per CU (hosts of up
	// to ~500 listeners: one workgroup's load / scan / flush phases run under the other's event phase -- r3t: 1.59 against 1.79 ms at 480
	// listeners per host); else 1024 threads x 16 events (16384-event tiles, one workgroup per CU: 1000-listener hosts -- there the two-
	// workgroup form needs half-full tables and 6-value pieces and loses, r3l / r3n); else 1024 x 8.  GYS_TPT = 8 / 12 / 16 pins a form (A/B).
	static const int tpt = [] { const char *e = getenv("GYS_TPT"); const int v = e ? atoi(e) : 0; return (v == 8 || v == 12 || v == 16 || v == 32) ? v : 0; }();
	bool tpt16 = false, tpt12 = false, tpt32 = false;
	if (host_local) {
		hp.ev = (const uint64_t *)d_ev;
		hp.n = n;
		hp.segs = segs_dev;
		hp.nsegs = nsegs;
		hp.hdesc = c->hdesc;
		hp.htbl = c->htbl;
		hp.hlst = c->hlst;
		hp.cand = c->cand_pool;
		hp.hll32 = c->hll32;
		hp.td_cur = c->td_cur;
		hp.td_pend = c->td_pend;
		hp.pcap = c->pcap;
		hp.td_run = c->td_run;
		hp.td_run1 = c->td_run1;
		hp.run_delta = (long long)(((intptr_t)c->staged - (intptr_t)c->td_pend) / 4);
		hp.staged = c->staged;
		hp.host_spill = c->host_spill;
		hp.spill_stamp = ++c->spill_stamp;
		fin.spill_stamp = hp.spill_stamp;
		hp.fin = fin;
		{
			static const uint32_t dbg = [] { const char *e = getenv("GYS_DBG"); return e ? (uint32_t)atoi(e) : 0u; }();
			hp.dbg = dbg;
		}
		hp.counters = c->counters;
		hp.svc_hll = c->svc_hll;
		hp.svc_hll_p = c->cfg.svc_hll_p;
		hp.ghist = ghist;
		hp.gmax = gmax;
		hp.lds_tbl_entries = max_tbl;
		hp.lds_key_entries = (uint32_t)align_up(max_l, 2);
		hgrid = nsegs;
		if (host_split || host_parts) {
			// virtual segments: every segment cut into parts of GYS_SPLIT_PART events (host_split) and, for a many-listener host, one
			// entry per part of its listeners (reserved = descriptor index + 1), the listener parts of one piece next to each other
			uint64_t nparts = 0;
			for (uint32_t s = 0; s < nsegs; ++s) {
				const uint64_t len = (s + 1 < nsegs ? segs_host[s + 1].first_event : n) - segs_host[s].first_event;
				const uint64_t pieces = host_split ? (len + GYS_SPLIT_PART - 1) / GYS_SPLIT_PART : (len ? 1 : 0);
				nparts += pieces * std::max<size_t>(c->host_lst[segs_host[s].host_slot].sub.size(), 1);
			}
			const uint64_t xbytes = nparts * sizeof(gys_resp_seg);
			if (xbytes > slot.xcap) {
				if (slot.xhost) HIPCHK(hipHostFree(slot.xhost));
				if (slot.xdev) HIPCHK(hipFree(slot.xdev));
				slot.xhost = slot.xdev = nullptr;
				slot.xcap = std::max<uint64_t>(xbytes, 1u << 16);
				HIPCHK(hipHostMalloc((void **)&slot.xhost, slot.xcap, hipHostMallocDefault));
				HIPCHK(hipMalloc((void **)&slot.xdev, slot.xcap));
			}
			gys_resp_seg *vseg = (gys_resp_seg *)slot.xhost;
			uint64_t v = 0;
			for (uint32_t s = 0; s < nsegs; ++s) {
				const uint64_t first = segs_host[s].first_event;
				const uint64_t len = (s + 1 < nsegs ? segs_host[s + 1].first_event : n) - first;
				const HostListeners &hl = c->host_lst[segs_host[s].host_slot];
				const uint64_t step = host_split ? (uint64_t)GYS_SPLIT_PART : std::max<uint64_t>(len, 1);
				for (uint64_t q = 0; q * step < len; ++q) {
					if (hl.sub.empty()) {
						vseg[v++] = gys_resp_seg{segs_host[s].host_slot, 0u, first + q * step};
					} else {
						for (uint32_t lp = 0; lp < hl.sub.size(); ++lp)
							vseg[v++] = gys_resp_seg{segs_host[s].host_slot, c->cfg.max_hosts + hl.sub_desc + lp + 1u, first + q * step};
					}
				}
			}
			HIPCHK(hipMemcpyAsync(slot.xdev, slot.xhost, v * sizeof(gys_resp_seg), hipMemcpyHostToDevice, c->stream));
			hp.segs = (const gys_resp_seg *)slot.xdev;
			hp.nsegs = (uint32_t)v;
			hgrid = (uint32_t)v;
			if (host_split) c->n_batches_host_split++;
			else c->n_batches_host_local++;
		} else {
			c->n_batches_host_local++;
		}
		// (two workgroups per CU: each gets half of the CU's LDS -- resp_dyn_max is 160 KiB minus ONE static part)
		tpt12 = mode == 0 && (tpt == 12 || tpt == 0) && resp_host_lds_bytes(max_tbl, hp.lds_key_entries, 6144u) + 160u * 1024u - c->resp_dyn_max <= 80u * 1024u;
		tpt16 = !tpt12 && (mode != 0 || tpt == 16 || tpt == 32 || tpt == 0) && resp_host_lds_bytes(max_tbl, hp.lds_key_entries, 16384u) <= c->resp_dyn_max;
		// GYS_TPT=32 (experiment): the fused first pass as 512 threads x 32 events with the next group's events prefetched into registers --
		// the same 16 384-event tile and LDS layout as the 1024 x 16 form, which the split form and the second pass keep using
		tpt32 = tpt16 && tpt == 32 && !host_split;
		dyn = resp_host_lds_bytes(max_tbl, hp.lds_key_entries, tpt12 ? 6144u : tpt16 ? 16384u : 8192u);
		if (pre) {
			// runs for the keys whose last batch, repeated, would overflow their buffer (nothing but a flag read when no key was that large)
			ProfScope ps(c, "prespill");
			PreSpillP pp{};
			pp.td_cur = c->td_cur;
			pp.td_prevm = c->td_prevm;
			pp.td_run = c->td_run;
			pp.td_run0 = c->td_run0;
			pp.td_run1 = c->td_run1;
			pp.counts = c->merge_count;
			pp.hot = c->pre_hot;
			pp.hot_rd = c->pre_seq & 1u;
			pp.svc_host = c->svc_host;
			pp.host_batch = c->host_batch;
			pp.batch_stamp = c->batch_stamp;
			pp.nsvc = nsvc;
			pp.pcap = c->pcap;
			pp.pend_cap = c->pend_cap;
			pp.resv = (unsigned long long *)(c->merge_count + 14); // (merge_count[14 .. 15]: cleared with the list lengths at the start of the batch)
			pp.run_limit = (uint32_t)std::min<uint64_t>(c->staged_cap - std::min<uint64_t>(n, c->staged_cap), 0xFFFFFFFFull); // the exact runs of the fall-back (<= n words) keep their room
			++c->pre_seq;
			hipLaunchKernelGGL(k_mark_hosts, dim3((nsegs + 255) / 256), dim3(256), 0, c->stream, segs_dev, nsegs, c->host_batch, c->batch_stamp);
			hipLaunchKernelGGL(k_prespill, dim3((nsvc + 255) / 256), dim3(256), 0, c->stream, pp);
		}
		{
			ProfScope ps(c, "resp_host");
			if (mode != 0) {
				if (host_split) launch_resp_host_mode<true, false>(c, mode, tpt16, hgrid, dyn, hp);
				else launch_resp_host_mode<false, false>(c, mode, tpt16, hgrid, dyn, hp);
			} else if (host_split) {
				if (tpt12) launch_resp_host<12, true, false>(c, hgrid, dyn, hp);
				else if (tpt16) launch_resp_host<16, true, false>(c, hgrid, dyn, hp);
				else launch_resp_host<8, true, false>(c, hgrid, dyn, hp);
			} else {
				if (tpt12) launch_resp_host<12, false, false>(c, hgrid, dyn, hp);
				else if (tpt32) launch_resp_host<32, false, false>(c, hgrid, dyn, hp);
				else if (tpt16) launch_resp_host<16, false, false>(c, hgrid, dyn, hp);
				else launch_resp_host<8, false, false>(c, hgrid, dyn, hp);
			}
		}
	} else {
		c->n_batches_general++;
		RespP1 p{};
		p.ev = (const uint64_t *)d_ev;
		p.cand = c->cand_pool;
		p.n = n;
		p.segs = segs_dev;
		p.nsegs = nsegs;
		p.lk = c->lk_tbl;
		p.svc_gid = c->svc_gid;
		p.hist_win = c->hist_win;
		p.bitmap = c->bitmap;
		p.hll32 = c->hll32;
		p.cms32 = cms32;
		p.batch_cnt = td ? c->batch_cnt : nullptr;
		p.ev_kv = td ? c->ev_kv : nullptr;
		p.counters = c->counters;
		p.svc_hll = c->svc_hll;
		p.svc_hll_p = c->cfg.svc_hll_p;
		p.ghist = ghist;
		p.gmax = gmax;
		{
			ProfScope ps(c, "resp_pass1");
			if (v6) hipLaunchKernelGGL(k_resp_pass1<true>, dim3(grid_for(n, 256, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, p);
			else hipLaunchKernelGGL(k_resp_pass1<false>, dim3(grid_for(n, 256, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, p);
		}
		HIPCHK(hipGetLastError());
		if (!td || nsvc == 0) {
			HIPCHK(hipEventRecord(slot.done, c->stream));
			return GYS_OK;
		}
		const uint32_t nblk = (nsvc + GYS_SCAN_TILE - 1) / GYS_SCAN_TILE;
		{
			ProfScope ps(c, "scan");
			hipLaunchKernelGGL(k_scan_block_sums, dim3(nblk), dim3(256), 0, c->stream, c->batch_cnt, nsvc, c->scan_block_sums);
			hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, c->stream, c->scan_block_sums, nblk);
			hipLaunchKernelGGL(k_scan_final, dim3(nblk), dim3(256), 0, c->stream, c->batch_cnt, nsvc, c->scan_block_sums, c->batch_off);
		}
		{
			ProfScope ps(c, "scatter");
			hipLaunchKernelGGL(k_resp_scatter, dim3(grid_for(n, 256, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, c->ev_kv, n, c->batch_off, c->staged);
		}
		{
			ProfScope ps(c, "key_append");
			AppendP ap{};
			ap.batch_cnt = c->batch_cnt;
			ap.off_end = c->batch_off;
			ap.staged = c->staged;
			ap.td_cur = c->td_cur;
			ap.td_pend = c->td_pend;
			ap.pcap = c->pcap;
			ap.nsvc = nsvc;
			const uint32_t nchunks = (nsvc + 63u) / 64u;
			hipLaunchKernelGGL(k_key_append, dim3(std::min<uint32_t>((nchunks + 3u) / 4u, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, ap);
		}
	}
	HIPCHK(hipGetLastError());
	if (!host_local || host_split) { // (the fused host-local form finalizes its keys in the tail of k_resp_host)
		ProfScope ps(c, "key_finalize");
		fin.batch_off = host_local ? nullptr : c->batch_off;
		hipLaunchKernelGGL(k_key_finalize, dim3((nsvc + 255) / 256), dim3(256), 0, c->stream, fin);
	}
	if (pre) // keys whose predicted run fits their buffer after all: the run is copied behind the buffered values (before the merges read either)
		hipLaunchKernelGGL(k_run_append, dim3((uint32_t)c->ncu), dim3(256), 0, c->stream, c->append_list, c->merge_count + FIN_APPEND, c->staged, c->td_pend, c->pcap);
	// (a key spills / lands in a larger merge class only when THIS batch brought it more values than its buffer had room for: a small
	// batch -- a partha message of a few thousand events -- cannot, and the launches for those cases are not made)
	if (host_local && n > (uint64_t)c->pcap - c->pend_cap) {
		// second pass over the hosts that have spilled services (a workgroup of any other host returns at once): their events again,
		// only the spilled services' values, into the runs k_key_finalize allocated in `staged`
		ProfScope ps(c, "resp_spill");
		if (mode != 0) launch_resp_host_mode<true, true>(c, mode, tpt16, hgrid, dyn, hp);
		else if (tpt12) launch_resp_host<12, true, true>(c, hgrid, dyn, hp);
		else if (tpt16) launch_resp_host<16, true, true>(c, hgrid, dyn, hp);
		else launch_resp_host<8, true, true>(c, hgrid, dyn, hp);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(slot.done, c->stream)); // the segment descriptors have been consumed once the stream gets here
	{
		MergeP mp{};
		mp.d = digest_params(c);
		const uint32_t cap = (uint32_t)std::min<uint64_t>(nsvc, n);
		static const bool old_merge = getenv("GYS_OLD_MERGE") != nullptr; // A/B: the general kernel for class 0 as well
		// Round 6: class 1 (more values than the fast class, up to 4096) through the value-bin kernel's 4096-value instance as well (the same
		// exact-integer definition: bit-identical), the general kernel only for what either instance hands over (GYS_CLASS1_GENERAL: round 5's routing, A/B)
		static const bool class1_general = getenv("GYS_CLASS1_GENERAL") != nullptr;
		const bool class1 = c->merge_fast < GYS_MERGE_CLASS1 && n > c->merge_fast - c->pend_cap;
		const bool class1_bins = class1 && !class1_general && !(old_merge && c->merge_fast <= GYS_MERGE_CLASS0);
		// size class 2 (4097 .. 16 384 values) through the streamed value-bin instance (finalize_one queues such keys on list 2 unless GYS_CLASS2_HUGE / GYS_OLD_HUGE is set)
		static const bool class2_huge = getenv("GYS_CLASS2_HUGE") != nullptr || getenv("GYS_OLD_HUGE") != nullptr;
		const bool class2 = !class2_huge && n > GYS_MERGE_CLASS1 - c->pend_cap;
		MergeBP bp{};
		bp.d = mp.d;
		bp.slow_list = c->merge_list_slow;
		bp.slow_count = c->merge_count + FIN_SLOW;
		auto slow_pass = [&](bool up_to_class1) { // entries whose total weight needs 64-bit arithmetic, or with more large values than the bin kernel's list holds (normally none)
			mp.list = c->merge_list_slow;
			mp.count = c->merge_count + FIN_SLOW;
			if (class2) hipLaunchKernelGGL((k_digest_merge<GYS_MERGE_LDS_MAX, 1024u>), dim3(std::max(1u, std::min<uint32_t>(cap, (uint32_t)c->ncu))), dim3(1024), 0, c->stream, mp); // (ONE pass over the list, behind the last value-bin launch: an entry is merged once)
			else if (!up_to_class1) hipLaunchKernelGGL((k_digest_merge<GYS_MERGE_CLASS0, 256u>), dim3(std::max(1u, std::min<uint32_t>(cap, (uint32_t)c->ncu))), dim3(256), 0, c->stream, mp);
			else hipLaunchKernelGGL((k_digest_merge<GYS_MERGE_CLASS1, 256u>), dim3(std::max(1u, std::min<uint32_t>(cap, (uint32_t)c->ncu))), dim3(256), 0, c->stream, mp);
		};
		{
			ProfScope ps(c, "digest_merge");
			mp.list = c->merge_list;
			mp.count = c->merge_count + FIN_CLASS0;
			if (old_merge && c->merge_fast <= GYS_MERGE_CLASS0) {
				hipLaunchKernelGGL((k_digest_merge<GYS_MERGE_CLASS0, 256u>), dim3(std::max(1u, std::min<uint32_t>(cap, (uint32_t)c->ncu * 7))), dim3(256), 0, c->stream, mp);
			} else {
				bp.list = c->merge_list;
				bp.count = c->merge_count + FIN_CLASS0;
				const dim3 bgrid(std::max(1u, std::min<uint32_t>(cap, (uint32_t)c->ncu * 8)));
				if (c->merge_fast <= 1024u) hipLaunchKernelGGL((k_digest_bins<false, 4u>), bgrid, dim3(256), 0, c->stream, bp);
				else if (c->merge_fast <= 2048u) hipLaunchKernelGGL((k_digest_bins<false, 8u>), bgrid, dim3(256), 0, c->stream, bp);
				else hipLaunchKernelGGL((k_digest_bins<false, 16u>), bgrid, dim3(256), 0, c->stream, bp);
				if (!class1_bins && !class2) slow_pass(c->merge_fast > GYS_MERGE_CLASS0); // (else: one pass over the list behind the last value-bin launch -- an entry is merged once)
			}
		}
		if (class1) {
			ProfScope ps(c, "digest_merge_big");
			if (class1_bins) {
				bp.list = c->merge_list1;
				bp.count = c->merge_count + FIN_CLASS1;
				hipLaunchKernelGGL((k_digest_bins<false, 16u>), dim3(std::max(1u, std::min<uint32_t>(cap, (uint32_t)c->ncu * GYS_MB_WAVES16))), dim3(256), 0, c->stream, bp);
				if (!class2) slow_pass(true);
			} else {
				mp.list = c->merge_list1;
				mp.count = c->merge_count + FIN_CLASS1;
				hipLaunchKernelGGL((k_digest_merge<GYS_MERGE_CLASS1, 256u>), dim3(std::max(1u, std::min<uint32_t>(cap, (uint32_t)c->ncu * 3))), dim3(256), 0, c->stream, mp);
			}
			// (entries above class 2 take the several-workgroup path below)
		}
		if (class2) {
			ProfScope ps(c, "digest_merge_big");
			bp.list = c->merge_list2;
			bp.count = c->merge_count + FIN_CLASS2;
			const uint32_t cap2 = (uint32_t)std::min<uint64_t>(nsvc, n / (GYS_MERGE_CLASS1 - c->pend_cap) + 1);
			hipLaunchKernelGGL((k_digest_bins<false, 64u>), dim3(std::max(1u, std::min<uint32_t>(cap2, (uint32_t)c->ncu * 8))), dim3(256), 0, c->stream, bp);
			slow_pass(true); // (what a value-bin launch handed over -- 64-bit weights, more than 1024 values of a second or longer -- through the general kernel's 16 384-value instance)
		}
	}
	if (n > GYS_MERGE_CLASS1 - c->pend_cap) {
		// huge keys (> 16 384 values in this call): several workgroups per key (gys_huge.hpp); what that path cannot take -- entries
		// beyond its pool, more than 4 096 values >= 16 384 in one key -- is handed to the one-workgroup kernel through a fallback list
		ProfScope ps(c, "digest_huge");
		static const bool old_huge = getenv("GYS_OLD_HUGE") != nullptr; // A/B
		HugeP h{};
		h.d = digest_params(c);
		h.scratch = c->huge_scratch;
		if (old_huge) {
			h.huge_list = c->huge_list;
			h.huge_count = c->merge_count + FIN_HUGE;
		} else {
			Huge2P q{};
			q.d = h.d;
			q.list = c->huge_list;
			q.count = c->merge_count + FIN_HUGE;
			q.bins = c->huge_scratch;
			q.acc = c->huge_acc;
			q.bm = c->huge_bm;
			q.chunk_off = c->huge_chunk_off;
			q.tail = c->huge_tail;
			q.tail_count = c->merge_count + 9;
			q.tail_cap = 1u << 20;
			q.maxent = c->huge_maxent;
			q.fb_list = c->huge_fb_list;
			q.fb_count = c->merge_count + 6;
			q.nent_used = c->merge_count + 7;
			q.tb_list = c->huge_tb_list;
			q.tb_count = c->merge_count + 10;
#ifdef GYS_HUGE_TIMING
			q.dbg = (unsigned long long *)c->counters + 20;
#endif
			// the pool holds huge_maxent entries: the list is walked in rounds (a round beyond the list's end costs four empty launches)
			const uint64_t list_cap = std::min<uint64_t>(std::min<uint64_t>(nsvc, n / (GYS_MERGE_CLASS1 - c->pend_cap) + 1), c->huge_list_cap);
			for (uint64_t first = 0; first < list_cap; first += c->huge_maxent) {
				q.first = (uint32_t)first;
				hipLaunchKernelGGL(k_huge_plan, dim3(1), dim3(1024), 0, c->stream, q);
				hipLaunchKernelGGL(k_huge_clear, dim3((uint32_t)c->ncu * 8), dim3(256), 0, c->stream, q);
				hipLaunchKernelGGL(k_huge_count, dim3((uint32_t)c->ncu * 2), dim3(1024), GYS_HB_BINS * 4, c->stream, q);
				// tier A: two 512-thread workgroups per CU (79 KiB of LDS each); tier B: whatever tier A handed over (usually nothing)
				hipLaunchKernelGGL((k_huge_merge<512, GYS_HB_TAIL_A, false>), dim3((uint32_t)c->ncu * 2), dim3(512), (GYS_HB_BINS + GYS_HB_TAIL_A) * 4, c->stream, q);
				hipLaunchKernelGGL((k_huge_merge<1024, GYS_HB_TAIL_LDS, true>), dim3((uint32_t)c->ncu), dim3(1024), (GYS_HB_BINS + GYS_HB_TAIL_LDS) * 4, c->stream, q);
			}
			h.huge_list = c->huge_fb_list;
			h.huge_count = c->merge_count + 6;
		}
		hipLaunchKernelGGL(k_digest_huge, dim3(c->huge_blocks), dim3(256), 0, c->stream, h);
	}
	HIPCHK(hipGetLastError());
	return GYS_OK;
}

// ------------------------------------------------------------------------------------------------ multi-level windows
constexpr int64_t LEVEL_SECS[GYS_NLEVELS] = {5, 300, 5 * 24 * 3600, 0}; // Level_5s_5min_5days_all common/gy_statistics.h:1545-1551

// most recent start (<= t) of ring bucket j of a level of `dur` seconds (folly BucketedTimeSeries::getBucketInfo; dur % ring == 0)
inline int64_t level_bucket_start(int64_t t, int64_t dur, uint32_t j)
{
	const int64_t s = (t / dur) * dur + (int64_t)j * (dur / GYS_LEVEL_RING);
	return s <= t ? s : s - dur;
}

inline uint32_t level_bucket_idx(int64_t t, int64_t dur) { return (uint32_t)((t % dur) * GYS_LEVEL_RING / dur); }

// window close at tusec: remember the closing window as level 0 and snapshot the cumulative records into every ring bucket whose
// start was crossed since the previous close
int level_roll(gys_ctx *c, uint64_t tusec)
{
	int64_t tnow = (int64_t)(tusec / 1000000ull);
	if (tnow < c->lvl_t_last) tnow = c->lvl_t_last; // time does not go backwards (BucketedTimeSeries::update)
	LevelRollP p{};
	for (int li = 0; li < 2; ++li) {
		const int64_t dur = LEVEL_SECS[li + 1];
		for (uint32_t j = 0; j < GYS_LEVEL_RING; ++j)
			if (c->lvl_t_last >= 0 && level_bucket_start(tnow, dur, j) > c->lvl_t_last) p.mask[li] |= 1u << j;
	}
	c->lvl_t_last = tnow;
	if (!c->nsvc) return GYS_OK;
	const bool keep_last = c->cfg.enable_levels == 1; // enable_levels = 2: no 5-s level (nothing per key unless a ring boundary was crossed)
	if (!keep_last && !(p.mask[0] | p.mask[1])) {
		// no snapshot due: only the time of a service's first window close (firstTime_ of its series) is kept up
		ProfScope ps(c, "level_roll");
		hipLaunchKernelGGL(k_level_first, dim3(grid_for(c->nsvc, 256, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, c->lvl_first, c->nsvc, tnow);
		HIPCHK(hipGetLastError());
		return GYS_OK;
	}
	// every service's closing-window record must be complete.  With lazily folded records (t-digest on) and the 5-s level, level 0 costs no pass
	// of its own: the fold leaves per service which window its record in hist_win belongs to (lvl_last_tag) and the two arrays change places --
	// the window records ARE the level-0 records, a key's first fold of the next window rewrites its (now stale) record in the other array
	// without reading it.  (Rounds 2 - 4 copied 256 B per service and close, k_level_roll: 1.1 ms at 10^7 services; writing the copy from the
	// fold pass was slower still, profiles/r5i_levels_fused_vs_separate.txt.)  k_level_roll runs when a ring boundary was crossed, for the
	// snapshots, or for the copy when the records are kept eagerly.
	const bool swap_last = keep_last && c->cfg.enable_tdigest;
	{
		const int rcf = swap_last ? fold_close_levels(c, tnow) : fold_range(c, 0, c->nsvc);
		if (rcf) return rcf;
	}
	if (!swap_last || (p.mask[0] | p.mask[1])) {
		p.win = c->hist_win;
		p.all = c->hist_all;
		p.meta = c->cfg.enable_tdigest ? c->td_meta : nullptr;
		p.epoch = c->epoch;
		p.nsvc = c->nsvc;
		p.snap = c->lvl_snap;
		p.last = keep_last && !swap_last ? c->lvl_last : nullptr;
		p.stride = c->cfg.max_services;
		p.first_sec = c->lvl_first;
		p.tnow = tnow;
		ProfScope ps(c, "level_roll");
		hipLaunchKernelGGL(k_level_roll, dim3(grid_for((uint64_t)c->nsvc * 16, 256, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, p);
		HIPCHK(hipGetLastError());
	}
	if (swap_last && c->lvl_last_epoch != c->epoch) { // (once per window: a gys_window_prepare that is retried after a failure further down must not swap back)
		std::swap(c->hist_win, c->lvl_last);
		c->lvl_last_epoch = c->epoch;
	}
	return GYS_OK;
}

// the array that holds the records of window c->epoch (what the window views read, each record behind its hw_epoch check): hist_win -- except
// between gys_window_prepare and gys_window_finish when the close has already swapped it with the level-0 array (level_roll)
inline const gys_hist_rec *window_records(gys_ctx *c)
{
	// (the swap itself is the test, not `prepared`: it is skipped without services, and it stays in force when the rest of the prepare step failed)
	return c->cfg.enable_levels == 1 && c->cfg.enable_tdigest && c->lvl_last_epoch == c->epoch ? c->lvl_last : c->hist_win;
}

// where a level's records come from at time tq (s): mode 0 = cumulative - *sub (nullptr: nothing to subtract), 1 = empty, 2 = copy of *sub
void level_source(gys_ctx *c, int level, int64_t tq, int *mode, const gys_hist_rec **sub)
{
	*sub = nullptr;
	if (level == 3) {
		*mode = 0;
	} else if (level == 0) {
		// a 5-s ring keeps an add for 5 s: the window closed last, until a window's length has passed without another close
		if (c->cfg.enable_levels == 1 && c->lvl_t_last >= 0 && tq - c->lvl_t_last < LEVEL_SECS[0]) {
			*mode = 2;
			*sub = c->lvl_last;
		} else {
			*mode = 1;
		}
	} else {
		const int64_t dur = LEVEL_SECS[level];
		const uint32_t oldest = (level_bucket_idx(tq, dur) + 1u) % GYS_LEVEL_RING;
		if (c->lvl_t_last < 0 || level_bucket_start(tq, dur, oldest) > c->lvl_t_last) {
			*mode = 1; // every add is older than the ring's oldest live bucket
		} else {
			*mode = 0;
			*sub = c->lvl_snap + ((uint64_t)(level - 1) * GYS_LEVEL_RING + oldest) * c->cfg.max_services;
		}
	}
}

// device records of one level at time tusec for slots [first, first + n) into d_out
int level_view(gys_ctx *c, int level, uint64_t tusec, uint32_t first, uint32_t n, gys_hist_rec *d_out)
{
	int64_t tq = (int64_t)(tusec / 1000000ull);
	if (tq < c->lvl_t_last) tq = c->lvl_t_last;
	{
		const int rcf = fold_range(c, first, n); // max_val_seen is reported over everything ingested, the open window included
		if (rcf) return rcf;
	}
	LevelViewP p{};
	p.win = c->hist_win;
	p.all = c->hist_all;
	p.meta = c->cfg.enable_tdigest ? c->td_meta : nullptr;
	p.epoch_open = c->epoch + (c->prepared ? 1u : 0u);
	p.first = first;
	p.n = n;
	p.out = d_out;
	level_source(c, level, tq, &p.mode, &p.sub);
	p.last_tag = c->cfg.enable_tdigest ? c->lvl_last_tag : nullptr;
	p.last_epoch = c->lvl_last_epoch;
	hipLaunchKernelGGL(k_level_view, dim3((uint32_t)(((uint64_t)n * 16 + 255) / 256)), dim3(256), 0, c->stream, p);
	HIPCHK(hipGetLastError());
	return GYS_OK;
}

// TIME_HISTOGRAM::get_stats_for_period_with_flush (common/gy_statistics.h:1378-1413) for slots [first, first + n): the interval's
// {count, sum} per histogram bucket into d_out.  start / end are the reference's starttime / endtime + 1 (:1383).
int level_period(gys_ctx *c, int64_t start, int64_t end, uint64_t tusec, uint32_t first, uint32_t n, gys_hist_rec *d_out, int *plevel)
{
	int64_t tq = (int64_t)(tusec / 1000000ull);
	if (tq < c->lvl_t_last) tq = c->lvl_t_last; // the flush: latestTime_ of every level
	{
		const int rcf = fold_range(c, first, n);
		if (rcf) return rcf;
	}
	LevelPeriodP p{};
	p.win = c->hist_win;
	p.all = c->hist_all;
	p.meta = c->cfg.enable_tdigest ? c->td_meta : nullptr;
	p.epoch_open = c->epoch + (c->prepared ? 1u : 0u);
	p.first = first;
	p.n = n;
	p.out = d_out;
	int level = GYS_NLEVELS - 1; // MultiLevelTimeSeries::getLevel(start): the first level that reaches back to start
	for (int l = c->cfg.enable_levels == 1 ? 0 : 1; l < GYS_NLEVELS - 1; ++l) // (enable_levels = 2: no 5-s level, the 300-s ring answers)
		if (tq - LEVEL_SECS[l] <= start) {
			level = l;
			break;
		}
	if (plevel) *plevel = level;
	if (level == 3) {
		p.mode = 3;
		p.first_sec = c->lvl_first;
		p.start = start;
		p.end = end;
		p.latest = tq;
	} else if (level == 0) {
		// the 5-s ring has 1-s buckets: the window closed last sits in [t_last, t_last + 1) for 5 s, inside the interval or not
		const bool held = c->lvl_t_last >= 0 && tq - c->lvl_t_last < LEVEL_SECS[0];
		p.mode = held && start <= c->lvl_t_last && end > c->lvl_t_last ? 2 : 1;
		p.last = c->lvl_last;
		p.last_tag = c->cfg.enable_tdigest ? c->lvl_last_tag : nullptr;
		p.last_epoch = c->lvl_last_epoch;
	} else {
		const int64_t dur = LEVEL_SECS[level], w = dur / GYS_LEVEL_RING;
		const int64_t cur_start = level_bucket_start(tq, dur, level_bucket_idx(tq, dur));
		auto cum_before = [&](int64_t s) -> const gys_hist_rec * { // C(s); nullptr = the cumulative record now
			if (c->lvl_t_last < 0 || s > c->lvl_t_last) return nullptr;
			const uint32_t j = (uint32_t)((s % dur) / w);
			return c->lvl_snap + ((uint64_t)(level - 1) * GYS_LEVEL_RING + j) * c->cfg.max_services;
		};
		p.mode = 0;
		for (int k = 0; k < GYS_LEVEL_RING; ++k) { // BucketedTimeSeries::forEachBucket(start, end, fn), oldest first
			const int64_t bs = cur_start - (int64_t)(GYS_LEVEL_RING - 1 - k) * w;
			int64_t bn = bs + w;
			if (start >= bn) continue;
			if (end <= bs) break;
			if (bs <= tq && bn > tq) bn = tq + 1; // rangeAdjust: the bucket that holds latestTime_ ends there
			const uint32_t i = p.nrb++;
			if (i == 0) p.bnd[0] = cum_before(bs);
			p.bnd[i + 1] = cum_before(bs + w);
			if (start <= bs && end >= bn) {
				p.whole_mask |= 1u << i;
				p.scale[i] = 1.f;
			} else {
				const int64_t is = start > bs ? start : bs, ie = end < bn ? end : bn;
				p.scale[i] = (float)(ie - is) * 1.f / (float)(bn - bs);
			}
		}
		if (!p.nrb) p.mode = 1;
	}
	hipLaunchKernelGGL(k_level_period, dim3((uint32_t)(((uint64_t)n * 16 + 255) / 256)), dim3(256), 0, c->stream, p);
	HIPCHK(hipGetLastError());
	return GYS_OK;
}

// walk a variable-stride batch on the host (what COMM validate + the reference loops do) and produce record offsets
template <typename SizeFn>
int walk_batch(const uint8_t *batch, uint32_t n, const uint8_t *pend, uint32_t fixed, SizeFn elem_size, std::vector<uint32_t> &offs)
{
	const uint8_t *p = batch;
	offs.clear();
	if (!pend || pend < batch) {
		set_err("batch without an end pointer");
		return GYS_ERR_INVAL;
	}
	if ((uint64_t)n * fixed > (uint64_t)(pend - batch)) { // (before anything is sized by n: every record is at least its fixed part)
		set_err("%u records cannot lie in a batch of %zu bytes", n, (size_t)(pend - batch));
		return GYS_ERR_INVAL;
	}
	offs.reserve(n);
	// the L1 validators' rule (TCP_CONN_NOTIFY::validate / LISTENER_STATE_NOTIFY::validate, common/gy_comm_proto.cc:859-880, :974-995):
	// every one of the n announced records lies complete before pend and has a size that is a multiple of 8; the L2 loop
	// `i < n && p < pend` (gy_mconnhdlr.cc:9130, :11175) only ever sees messages that passed it
	for (uint32_t i = 0; i < n; ++i) {
		if ((size_t)(pend - p) < fixed) {
			set_err("batch ends before record %u of %u", i, n);
			return GYS_ERR_INVAL;
		}
		const uint32_t sz = elem_size(p);
		if ((sz & 7u) || (size_t)(pend - p) < sz) {
			set_err("record %u: bad element size %u", i, sz);
			return GYS_ERR_INVAL;
		}
		offs.push_back((uint32_t)(p - batch));
		p += sz;
	}
	return GYS_OK;
}

int run_conn(gys_ctx *c, const uint8_t *d_batch, const uint32_t *d_offsets, uint32_t n)
{
	if (!n) return GYS_OK;
	ConnP p{};
	p.batch = d_batch;
	p.offsets = d_offsets;
	p.n = n;
	p.gid = c->gid_tbl;
	p.hll32 = c->hll32;
	p.cms32 = (uint32_t *)(c->arena + c->al.off_u32) + c->al.u32_cms;
	p.cms64 = (unsigned long long *)(c->arena + c->al.off_i64sum) + c->al.i64_cms;
	p.svc_win = c->svc_win;
	if (c->cfg.conn_pair_cms) {
		p.pair32 = (uint32_t *)(c->arena + c->al.off_u32) + c->al.u32_cpair;
		p.pair64 = (unsigned long long *)(c->arena + c->al.off_i64sum) + c->al.i64_cpair;
		p.cpair32 = p.pair32 + (uint64_t)GYS_CMS_D * GYS_CMS_W;
		p.cpair64 = p.pair64 + (uint64_t)GYS_CMS_D * GYS_CMS_W;
	}
	c->conn_dirty = true;
	p.counters = c->counters;
	ProfScope ps(c, "conn");
	p.span = conn_span(n, c->ncu);
	hipLaunchKernelGGL(k_conn_ingest, dim3((n + p.span - 1) / p.span), dim3(GYS_CONN_THREADS), 0, c->stream, p);
	HIPCHK(hipGetLastError());
	return GYS_OK;
}

// folds the connection path's per-service window accumulators (cumulative counters, Count-Min rows)
int conn_fold(gys_ctx *c)
{
	if (!c->conn_dirty || !c->nsvc) return GYS_OK;
	hipLaunchKernelGGL(k_conn_fold, dim3((c->nsvc + 255) / 256), dim3(256), 0, c->stream, c->svc_win, c->svc_ctr, c->svc_gid, c->nsvc,
			   (uint32_t *)(c->arena + c->al.off_u32) + c->al.u32_cms, (unsigned long long *)(c->arena + c->al.off_i64sum) + c->al.i64_cms);
	HIPCHK(hipGetLastError());
	c->conn_dirty = false;
	return GYS_OK;
}

int run_actconn(gys_ctx *c, const uint8_t *d_batch, uint32_t n)
{
	if (!n) return GYS_OK;
	ActConnP p{};
	p.batch = d_batch;
	p.n = n;
	p.gid = c->gid_tbl;
	p.pair32 = (uint32_t *)(c->arena + c->al.off_u32) + c->al.u32_pair;
	p.pair64 = (unsigned long long *)(c->arena + c->al.off_i64sum) + c->al.i64_pair;
	p.svc_act = c->svc_act;
	p.counters = c->counters;
	p.win_rows = (uint32_t *)(c->arena + c->al.off_u32) + c->al.u32_misc;
	ProfScope ps(c, "actconn");
	hipLaunchKernelGGL(k_actconn_ingest, dim3((n + 255) / 256), dim3(256), 0, c->stream, p);
	HIPCHK(hipGetLastError());
	return GYS_OK;
}

// builds the window's Count-Min rows of the response path from the per-service event counts ("Count-Min rows of the window" in
// gys_kernels.hpp); the counts start the next window from zero
int resp_cms_fold(gys_ctx *c)
{
	if (!c->resp_dirty || !c->nsvc || !c->resp_win) return GYS_OK;
	const uint32_t nch = std::max(1u, std::min<uint32_t>(c->cms_nch, (c->nsvc + 65535u) / 65536u));
	uint32_t *cms32 = (uint32_t *)(c->arena + c->al.off_u32) + c->al.u32_cms;
	hipLaunchKernelGGL(k_cms_partial, dim3(nch, GYS_CMS_D * 2), dim3(1024), GYS_CMSF_CELLS * 4, c->stream, c->resp_win, c->svc_gid, c->nsvc, nch, c->cms_partial);
	hipLaunchKernelGGL(k_cms_reduce, dim3((GYS_CMS_D * GYS_CMS_W + 255) / 256), dim3(256), 0, c->stream, c->cms_partial, nch, cms32);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemsetAsync(c->resp_win, 0, (uint64_t)c->nsvc * 4, c->stream));
	c->resp_dirty = false;
	return GYS_OK;
}

int run_lstate(gys_ctx *c, const uint8_t *d_batch, const uint32_t *d_offsets, const uint32_t *d_host_slot, uint32_t single_host, uint32_t n)
{
	if (!n) return GYS_OK;
	LStateP p{};
	p.batch = d_batch;
	p.offsets = d_offsets;
	p.host_slot = d_host_slot;
	p.single_host = single_host;
	p.n = n;
	p.gid = c->gid_tbl;
	p.svc_state = c->svc_state;
	p.host_summ = c->host_summ_win;
	p.epoch = c->epoch;
	p.counters = c->counters;
	p.qps_hist = c->qps_hist;
	p.act_hist = c->act_hist;
	ProfScope ps(c, "lstate");
	p.claim = c->svc_claim;
	if (++c->lstate_launch == 0) { // (enq_mu or the exclusive call lock is held; 0 = the cleared claim table)
		// the call counter wrapped (2^32 calls: weeks of a large site's messages): claims of old calls would now outrank every new one
		c->lstate_launch = 1;
		HIPCHK(hipMemsetAsync(c->svc_claim, 0, (uint64_t)c->cfg.max_services * 8, c->stream));
	}
	p.launch = c->lstate_launch;
	if (n <= GYS_LSTATE_FUSED_MAX) { // a partha's message: one launch (the per-message path is launch-bound)
		hipLaunchKernelGGL(k_lstate_both, dim3(1), dim3(GYS_LSTATE_FUSED_MAX), 0, c->stream, p);
	} else {
		hipLaunchKernelGGL(k_lstate_ingest, dim3((n + 255) / 256), dim3(256), 0, c->stream, p);
		hipLaunchKernelGGL(k_lstate_keep, dim3((n + 255) / 256), dim3(256), 0, c->stream, p);
	}
	HIPCHK(hipGetLastError());
	return GYS_OK;
}

// records of one variable-stride batch [batch, batch + bytes) + their offsets through a staging slot: one pinned copy, one H2D copy
// (records, then the offsets behind them), the ingest kernel, no wait
int ingest_staged_records(gys_ctx *c, uint32_t host, const void *batch, uint64_t bytes, const std::vector<uint32_t> &offs, bool conn)
{
	const uint64_t off_at = align_up(bytes, 8), total = off_at + offs.size() * 4;
	int si;
	int rc = stage_acquire(c, total, &si);
	if (rc) return rc;
	gys_ctx::Stage &st = c->stage[si];
	memcpy(st.h, batch, bytes);
	memcpy(st.h + off_at, offs.data(), offs.size() * 4);
	{
		std::lock_guard<std::mutex> g(c->enq_mu);
		// (copy on the copy stream, kernels on the engine stream behind an event -- as for the response submissions: the copy of one
		// message runs under the kernels of the message before it)
		hipError_t e = hipMemcpyAsync(st.d, st.h, total, hipMemcpyHostToDevice, c->copy_stream ? c->copy_stream : c->stream);
		if (e == hipSuccess && c->copy_stream) {
			e = hipEventRecord(st.copied, c->copy_stream);
			if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, st.copied, 0);
		}
		if (e == hipSuccess) {
			rc = conn ? run_conn(c, st.d, (const uint32_t *)(st.d + off_at), (uint32_t)offs.size())
				  : run_lstate(c, st.d, (const uint32_t *)(st.d + off_at), nullptr, host, (uint32_t)offs.size());
			e = hipEventRecord(st.done, c->stream);
		}
		if (e != hipSuccess) {
			set_err("ingest: %s", hipGetErrorString(e));
			rc = GYS_ERR_HIP;
		}
	}
	stage_release(c, si);
	return rc;
}


// ---- submission queue of the host-pointer response path (gys_ctx::RespQ)
constexpr uint64_t GYS_RQ_EVENTS = 1u << 21; // events per combined batch (48 MiB pinned + 48 MiB device each)

// submits batch bi (sealed, no writers); rq.mu NOT held
int rq_submit_one(gys_ctx *c, int bi)
{
	gys_ctx::RespBatch &b = c->rq.b[bi];
	int rc = GYS_OK;
	{
		std::lock_guard<std::mutex> g(c->enq_mu);
		// copy on its own stream, kernels on the engine stream behind an event: with everything on one stream the 48-MiB copy of a
		// submission (0.85 ms at 57 GB/s) and its kernels alternate; the batch's buffers are not reused before `done` has fired
		hipError_t e = hipMemcpyAsync(b.d, b.h, b.fill * 24, hipMemcpyHostToDevice, c->copy_stream ? c->copy_stream : c->stream);
		if (e == hipSuccess && c->copy_stream) {
			e = hipEventRecord(b.copied, c->copy_stream);
			if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, b.copied, 0);
		}
		if (e == hipSuccess) {
			rc = run_resp_batch(c, b.segs.data(), (uint32_t)b.segs.size(), b.d, b.fill);
			e = hipEventRecord(b.done, c->stream);
		}
		if (e != hipSuccess) {
			set_err("response submission: %s", hipGetErrorString(e));
			rc = GYS_ERR_HIP;
		}
	}
	return rc;
}

constexpr size_t GYS_RQ_INFLIGHT = 2; // submissions executing / queued on the GPU before new calls start to accumulate

// rq.mu held: batches whose kernels have finished go back to the free list
void rq_reap(gys_ctx *c)
{
	gys_ctx::RespQ &q = c->rq;
	// anything but "not ready" ends a submission's stay on the in-flight list: an event in an error state never turns into hipSuccess, and
	// a head that is never reaped would leave every later caller waiting once the free list has drained
	while (!q.inflight.empty()) {
		const hipError_t e = hipEventQuery(q.b[q.inflight.front()].done);
		if (e == hipErrorNotReady) break;
		if (e != hipSuccess && !q.async_rc) {
			q.async_rc = GYS_ERR_HIP;
			q.async_err = std::string("response submission: ") + hipGetErrorString(e);
		}
		q.free.push_back(q.inflight.front());
		q.inflight.pop_front();
	}
	(void)hipGetLastError(); // (hipErrorNotReady is not an error)
}

// With rq.mu held (lk): submit, oldest first, every sealed batch whose writers are done -- and the open batch too when `all`, or when it
// has data, no writer, and fewer than GYS_RQ_INFLIGHT submissions are still on the GPU.  Returns the first error of a batch that
// carried the caller's own data (`mine`), other errors are parked in rq.async_rc.
int rq_drain(gys_ctx *c, std::unique_lock<std::mutex> &lk, int mine, bool all)
{
	gys_ctx::RespQ &q = c->rq;
	int my_rc = GYS_OK;
	if (q.submitting) {
		if (!all) return GYS_OK; // the thread inside the submission picks up what accumulates
		q.cv.wait(lk, [&] { return !q.submitting; });
	}
	for (;;) {
		if (q.sealed.empty() && q.open >= 0 && q.b[q.open].fill && q.b[q.open].writers == 0) {
			rq_reap(c);
			if (all || q.inflight.size() < GYS_RQ_INFLIGHT) {
				q.b[q.open].sealed = true;
				q.sealed.push_back(q.open);
				q.open = -1;
			}
		}
		if (q.sealed.empty()) break;
		const int bi = q.sealed.front();
		if (q.b[bi].writers) {
			if (!all) break; // its last writer drains
			q.cv.wait(lk, [&] { return q.b[bi].writers == 0; });
		}
		q.sealed.pop_front();
		q.submitting = true;
		lk.unlock();
		const int rc = rq_submit_one(c, bi);
		lk.lock();
		q.submitting = false;
		q.submissions++;
		q.b[bi].sealed = false;
		q.b[bi].fill = 0;
		q.b[bi].segs.clear();
		q.inflight.push_back(bi);
		q.cv.notify_all();
		if (rc) {
			if (bi == mine) my_rc = rc;
			else if (!q.async_rc) {
				q.async_rc = rc;
				q.async_err = g_err;
			}
		}
	}
	return my_rc;
}

// everything the queue holds is on the stream when this returns (entry points other than the concurrent ingest calls start with it)
int rq_flush(gys_ctx *c)
{
	gys_ctx::RespQ &q = c->rq;
	std::unique_lock<std::mutex> lk(q.mu);
	int rc = GYS_OK;
	if (q.open >= 0 || !q.sealed.empty() || q.submitting) rc = rq_drain(c, lk, -1, true);
	if (!rc && q.async_rc) {
		rc = q.async_rc;
		set_err("%s", q.async_err.c_str());
		q.async_rc = 0;
	}
	return rc;
}

// flusher thread of the submission queue: sleeps until a call leaves events behind in the open batch, then tries every 200 us to submit
// it (rq_drain submits only while fewer than GYS_RQ_INFLIGHT submissions are on the GPU); an error lands in rq.async_rc for the next caller
void rq_flusher(gys_ctx *c)
{
	gys_ctx::RespQ &q = c->rq;
	(void)hipSetDevice(c->device);
	std::unique_lock<std::mutex> lk(q.mu);
	for (;;) {
		q.fcv.wait(lk, [&] { return q.stop || (q.open >= 0 && q.b[q.open].fill != 0); });
		if (q.stop) break;
		// the GPU is the bottleneck while GYS_RQ_INFLIGHT submissions are on it: wait for the oldest one's event (outside the lock) rather than
		// wake every 200 us to find nothing to do; otherwise give the burst 200 us to bring more calls before its tail goes out
		hipEvent_t busy = q.inflight.size() >= GYS_RQ_INFLIGHT ? q.b[q.inflight.front()].done : nullptr;
		lk.unlock();
		if (busy) (void)hipEventSynchronize(busy);
		else std::this_thread::sleep_for(std::chrono::microseconds(200));
		lk.lock();
		if (q.stop) break;
		if (q.open >= 0 && q.b[q.open].fill && q.b[q.open].writers == 0 && !q.submitting) {
			const uint64_t before = q.submissions;
			(void)rq_drain(c, lk, -1, false);
			if (q.submissions != before) q.tail_flushes++;
		}
	}
}

int rq_ingest(gys_ctx *c, uint32_t host, const void *ev24, uint32_t n)
{
	gys_ctx::RespQ &q = c->rq;
	std::unique_lock<std::mutex> lk(q.mu);
	if (!q.flusher_on) {
		q.flusher_on = true;
		q.flusher = std::thread(rq_flusher, c);
	}
	// (an earlier submission made for other callers may have failed: that error is reported once, by the first call that comes by -- AFTER
	// the caller's own events have been queued below, never instead of them)
	q.calls++;
	if (q.host_stamp.size() < c->hosts.size()) q.host_stamp.resize(c->hosts.size(), 0);
	int bi;
	for (;;) {
		if (q.open < 0) {
			rq_reap(c);
			if (q.free.empty()) {
				if (!q.inflight.empty()) { // every batch is on the GPU: wait for the oldest (outside the lock), then look again
					hipEvent_t ev = q.b[q.inflight.front()].done;
					lk.unlock();
					(void)hipEventSynchronize(ev);
					lk.lock();
				} else {
					q.cv.wait(lk, [&] { return !q.free.empty() || !q.inflight.empty(); }); // (sealed / being submitted by others)
				}
				continue;
			}
			bi = q.free.front();
			q.free.pop_front();
			gys_ctx::RespBatch &nb = q.b[bi];
			lk.unlock(); // (first-use allocation: outside the lock; the batch is not visible yet)
			hipError_t e = hipSuccess;
			if (!nb.h || !nb.d || !nb.done || !nb.copied) {
				nb.cap_events = std::min<uint64_t>(GYS_RQ_EVENTS, std::max<uint64_t>(c->cfg.max_batch_events, 1));
				if (!nb.h) e = hipHostMalloc((void **)&nb.h, nb.cap_events * 24, hipHostMallocDefault);
				if (e == hipSuccess && !nb.d) e = hipMalloc((void **)&nb.d, nb.cap_events * 24);
				if (e == hipSuccess && !nb.done) e = hipEventCreateWithFlags(&nb.done, hipEventDisableTiming);
				if (e == hipSuccess && !nb.copied) e = hipEventCreateWithFlags(&nb.copied, hipEventDisableTiming);
				if (e != hipSuccess) { // all four or none: a half-built batch must not look usable to the next caller
					if (nb.h) (void)hipHostFree(nb.h);
					if (nb.d) (void)hipFree(nb.d);
					if (nb.done) (void)hipEventDestroy(nb.done);
					if (nb.copied) (void)hipEventDestroy(nb.copied);
					nb.h = nullptr;
					nb.d = nullptr;
					nb.done = nb.copied = nullptr;
				}
			}
			lk.lock();
			if (e != hipSuccess) {
				q.free.push_back(bi);
				q.cv.notify_all();
				set_err("response batch buffers: %s", hipGetErrorString(e));
				return GYS_ERR_HIP;
			}
			if (q.open >= 0) { // another caller opened one meanwhile
				q.free.push_front(bi);
				q.cv.notify_all();
				continue;
			}
			q.open = bi;
			++q.stamp;
		}
		bi = q.open;
		gys_ctx::RespBatch &b = q.b[bi];
		if (b.fill + n <= b.cap_events && q.host_stamp[host] != q.stamp) break;
		// no room, or the host already has a segment in this batch: seal it (submitted before anything opened later)
		b.sealed = true;
		q.sealed.push_back(bi);
		q.open = -1;
		const int rc = rq_drain(c, lk, -1, false);
		if (rc) return rc;
	}
	gys_ctx::RespBatch &b = q.b[bi];
	const uint64_t off = b.fill;
	b.fill += n;
	b.segs.push_back(gys_resp_seg{host, 0, off});
	b.writers++;
	q.host_stamp[host] = q.stamp;
	lk.unlock();
	memcpy(b.h + off * 24, ev24, (uint64_t)n * 24); // the caller's buffer is free from here on
	lk.lock();
	b.writers--;
	if (b.writers == 0) q.cv.notify_all();
	int rc = rq_drain(c, lk, bi, false);
	if (q.open >= 0 && q.b[q.open].fill) q.fcv.notify_one(); // events stay behind in the open batch: the flusher sees to them if no call follows
	if (!rc && q.async_rc) { // an earlier submission made for other callers (or by the flusher) failed: reported once, this call's events are queued all the same
		rc = q.async_rc;
		set_err("%s", q.async_err.c_str());
		q.async_rc = 0;
	}
	return rc;
}

// ---- submission queues of the host-pointer connection / listener-state calls (gys_ctx::RecQ; the logic of rq_* above)
constexpr uint64_t GYS_CQ_CONN_BYTES = 16u << 20, GYS_CQ_LSTATE_BYTES = 4u << 20; // record bytes per combined batch

int recq_submit_one(gys_ctx *c, gys_ctx::RecQ &q, int bi)
{
	gys_ctx::RecBatch &b = q.b[bi];
	int rc = GYS_OK;
	std::lock_guard<std::mutex> g(c->enq_mu);
	hipStream_t cs = c->copy_stream ? c->copy_stream : c->stream;
	const uint64_t off_at = b.cap_bytes, host_at = b.cap_bytes + (uint64_t)b.cap_recs * 4;
	hipError_t e = hipMemcpyAsync(b.d, b.h, b.fill, hipMemcpyHostToDevice, cs);
	if (e == hipSuccess) e = hipMemcpyAsync(b.d + off_at, b.h + off_at, (uint64_t)b.nrec * 4, hipMemcpyHostToDevice, cs);
	if (e == hipSuccess && !q.conn) e = hipMemcpyAsync(b.d + host_at, b.h + host_at, (uint64_t)b.nrec * 4, hipMemcpyHostToDevice, cs);
	if (e == hipSuccess && c->copy_stream) {
		e = hipEventRecord(b.copied, c->copy_stream);
		if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, b.copied, 0);
	}
	if (e == hipSuccess) {
		rc = q.conn ? run_conn(c, b.d, (const uint32_t *)(b.d + off_at), b.nrec)
			    : run_lstate(c, b.d, (const uint32_t *)(b.d + off_at), (const uint32_t *)(b.d + host_at), 0, b.nrec);
		e = hipEventRecord(b.done, c->stream);
	}
	if (e != hipSuccess) {
		set_err("record submission: %s", hipGetErrorString(e));
		rc = GYS_ERR_HIP;
	}
	return rc;
}

void recq_reap(gys_ctx::RecQ &q) // q.mu held
{
	while (!q.inflight.empty()) {
		const hipError_t e = hipEventQuery(q.b[q.inflight.front()].done);
		if (e == hipErrorNotReady) break;
		if (e != hipSuccess && !q.async_rc) {
			q.async_rc = GYS_ERR_HIP;
			q.async_err = std::string("record submission: ") + hipGetErrorString(e);
		}
		q.free.push_back(q.inflight.front());
		q.inflight.pop_front();
	}
	(void)hipGetLastError();
}

// q.mu held (lk): as rq_drain
int recq_drain(gys_ctx *c, gys_ctx::RecQ &q, std::unique_lock<std::mutex> &lk, int mine, bool all)
{
	int my_rc = GYS_OK;
	if (q.submitting) {
		if (!all) return GYS_OK;
		q.cv.wait(lk, [&] { return !q.submitting; });
	}
	for (;;) {
		if (q.sealed.empty() && q.open >= 0 && q.b[q.open].nrec && q.b[q.open].writers == 0) {
			recq_reap(q);
			if (all || q.inflight.size() < GYS_RQ_INFLIGHT) {
				q.sealed.push_back(q.open);
				q.open = -1;
			}
		}
		if (q.sealed.empty()) break;
		const int bi = q.sealed.front();
		if (q.b[bi].writers) {
			if (!all) break; // its last writer drains
			q.cv.wait(lk, [&] { return q.b[bi].writers == 0; });
		}
		q.sealed.pop_front();
		q.submitting = true;
		lk.unlock();
		const int rc = recq_submit_one(c, q, bi);
		lk.lock();
		q.submitting = false;
		q.submissions++;
		q.b[bi].fill = 0;
		q.b[bi].nrec = 0;
		q.inflight.push_back(bi);
		q.cv.notify_all();
		if (rc) {
			if (bi == mine) my_rc = rc;
			else if (!q.async_rc) {
				q.async_rc = rc;
				q.async_err = g_err;
			}
		}
	}
	return my_rc;
}

int recq_flush(gys_ctx *c, gys_ctx::RecQ &q)
{
	std::unique_lock<std::mutex> lk(q.mu);
	int rc = GYS_OK;
	if (q.open >= 0 || !q.sealed.empty() || q.submitting) rc = recq_drain(c, q, lk, -1, true);
	if (!rc && q.async_rc) {
		rc = q.async_rc;
		set_err("%s", q.async_err.c_str());
		q.async_rc = 0;
	}
	return rc;
}

void recq_flusher(gys_ctx *c, gys_ctx::RecQ *qp) // the tail of a burst: as rq_flusher
{
	gys_ctx::RecQ &q = *qp;
	(void)hipSetDevice(c->device);
	std::unique_lock<std::mutex> lk(q.mu);
	for (;;) {
		q.fcv.wait(lk, [&] { return q.stop || (q.open >= 0 && q.b[q.open].nrec != 0); });
		if (q.stop) break;
		hipEvent_t busy = q.inflight.size() >= GYS_RQ_INFLIGHT ? q.b[q.inflight.front()].done : nullptr; // (as rq_flusher)
		lk.unlock();
		if (busy) (void)hipEventSynchronize(busy);
		else std::this_thread::sleep_for(std::chrono::microseconds(200));
		lk.lock();
		if (q.stop) break;
		if (q.open >= 0 && q.b[q.open].nrec && q.b[q.open].writers == 0 && !q.submitting) {
			const uint64_t before = q.submissions;
			(void)recq_drain(c, q, lk, -1, false);
			if (q.submissions != before) q.tail_flushes++;
		}
	}
}

// one message: `bytes` of records at `batch`, offs[i] = offset of record i in it; host = the sender's slot
int recq_ingest(gys_ctx *c, gys_ctx::RecQ &q, uint32_t host, const void *batch, uint64_t bytes, const std::vector<uint32_t> &offs)
{
	const uint64_t cap_bytes = q.conn ? GYS_CQ_CONN_BYTES : GYS_CQ_LSTATE_BYTES, need = align_up(bytes, 8);
	const uint32_t cap_recs = (uint32_t)(cap_bytes / (q.conn ? 280u : 88u)), n = (uint32_t)offs.size();
	if (need > cap_bytes / 2 || n > cap_recs / 2) {
		// a call of many messages' size (a replayed backlog): on its own through the staging ring -- behind what the queue holds
		const int rc = recq_flush(c, q);
		return rc ? rc : ingest_staged_records(c, host, batch, bytes, offs, q.conn);
	}
	std::unique_lock<std::mutex> lk(q.mu);
	if (!q.flusher_on) {
		q.flusher_on = true;
		q.flusher = std::thread(recq_flusher, c, &q);
	}
	// (a parked error of an earlier submission is reported after this message has been queued, see rq_ingest)
	q.calls++;
	int bi;
	for (;;) {
		if (q.open < 0) {
			recq_reap(q);
			if (q.free.empty()) {
				if (!q.inflight.empty()) {
					hipEvent_t ev = q.b[q.inflight.front()].done;
					lk.unlock();
					(void)hipEventSynchronize(ev);
					lk.lock();
				} else {
					q.cv.wait(lk, [&] { return !q.free.empty() || !q.inflight.empty(); });
				}
				continue;
			}
			bi = q.free.front();
			q.free.pop_front();
			gys_ctx::RecBatch &nb = q.b[bi];
			lk.unlock(); // (first-use allocation outside the lock; the batch is not visible yet)
			hipError_t e = hipSuccess;
			if (!nb.h || !nb.d || !nb.done || !nb.copied) {
				const uint64_t total = cap_bytes + (uint64_t)cap_recs * 8;
				if (!nb.h) e = hipHostMalloc((void **)&nb.h, total, hipHostMallocDefault);
				if (e == hipSuccess && !nb.d) e = hipMalloc((void **)&nb.d, total);
				if (e == hipSuccess && !nb.done) e = hipEventCreateWithFlags(&nb.done, hipEventDisableTiming);
				if (e == hipSuccess && !nb.copied) e = hipEventCreateWithFlags(&nb.copied, hipEventDisableTiming);
				if (e != hipSuccess) { // all four or none
					if (nb.h) (void)hipHostFree(nb.h);
					if (nb.d) (void)hipFree(nb.d);
					if (nb.done) (void)hipEventDestroy(nb.done);
					if (nb.copied) (void)hipEventDestroy(nb.copied);
					nb.h = nb.d = nullptr;
					nb.done = nb.copied = nullptr;
				} else {
					nb.cap_bytes = cap_bytes;
					nb.cap_recs = cap_recs;
				}
			}
			lk.lock();
			if (e != hipSuccess) {
				q.free.push_back(bi);
				q.cv.notify_all();
				set_err("record batch buffers: %s", hipGetErrorString(e));
				return GYS_ERR_HIP;
			}
			if (q.open >= 0) { // another caller opened one meanwhile
				q.free.push_front(bi);
				q.cv.notify_all();
				continue;
			}
			q.open = bi;
		}
		bi = q.open;
		gys_ctx::RecBatch &b = q.b[bi];
		if (b.fill + need <= b.cap_bytes && b.nrec + n <= b.cap_recs) break;
		q.sealed.push_back(bi); // no room: submitted before anything opened later
		q.open = -1;
		const int rc = recq_drain(c, q, lk, -1, false);
		if (rc) return rc;
	}
	gys_ctx::RecBatch &b = q.b[bi];
	const uint64_t at = b.fill;
	const uint32_t r0 = b.nrec;
	b.fill += need;
	b.nrec += n;
	b.writers++;
	lk.unlock();
	memcpy(b.h + at, batch, bytes); // the caller's buffer is free from here on
	uint32_t *o = (uint32_t *)(b.h + b.cap_bytes) + r0;
	for (uint32_t i = 0; i < n; ++i) o[i] = offs[i] + (uint32_t)at;
	if (!q.conn) {
		uint32_t *hs = (uint32_t *)(b.h + b.cap_bytes) + b.cap_recs + r0;
		for (uint32_t i = 0; i < n; ++i) hs[i] = host;
	}
	lk.lock();
	b.writers--;
	if (b.writers == 0) q.cv.notify_all();
	int rc = recq_drain(c, q, lk, bi, false);
	if (q.open >= 0 && q.b[q.open].nrec) q.fcv.notify_one();
	if (!rc && q.async_rc) { // (as rq_ingest: another submission's error, reported once; this message is queued)
		rc = q.async_rc;
		set_err("%s", q.async_err.c_str());
		q.async_rc = 0;
	}
	return rc;
}

} // namespace

// ==================================================================================================== C ABI
// Every entry point makes the context's device the calling thread's current device first: the reference's L2 threads (and a process that
// holds one context per GPU) call in with whatever device they used last, and allocations / launches follow the CURRENT device.
#define GYS_ENTER_NOFLUSH(c)                                                   \
	do {                                                                   \
		if ((c) && hipSetDevice((c)->device) != hipSuccess) {          \
			set_err("hipSetDevice(%d) failed", (c)->device);      \
			return GYS_ERR_HIP;                                    \
		}                                                              \
	} while (0)
// ... and, except for the host-pointer ingest calls that may run concurrently, puts whatever the response submission queue still holds
// on the stream first (stream order = call order for everything that reads or closes state)
#define GYS_ENTER(c)                                                           \
	do {                                                                   \
		GYS_ENTER_NOFLUSH(c);                                          \
		if (c) {                                                       \
			int rcq_ = rq_flush(c);                                \
			if (!rcq_) rcq_ = recq_flush(c, (c)->cq[0]);           \
			if (!rcq_) rcq_ = recq_flush(c, (c)->cq[1]);           \
			if (rcq_) return rcq_;                                 \
		}                                                              \
	} while (0)

// "Nothing throws across the boundary" (include/gysketch.h): the host side uses std::vector / std::string / std::thread, whose failures
// are C++ exceptions.  Every int-returning entry point is a function-try-block that turns them into an error code + gys_last_error() text.
#define GYS_CATCH_ALL                                                                                  \
	catch (const std::bad_alloc &)                                                                 \
	{                                                                                              \
		set_err("out of host memory");                                                         \
		return GYS_ERR_NOMEM;                                                                  \
	}                                                                                              \
	catch (const std::exception &ex_)                                                              \
	{                                                                                              \
		set_err("internal error: %s", ex_.what());                                             \
		return GYS_ERR_INTERNAL;                                                               \
	}                                                                                              \
	catch (...)                                                                                    \
	{                                                                                              \
		set_err("internal error");                                                             \
		return GYS_ERR_INTERNAL;                                                               \
	}


extern "C" {

uint32_t gys_abi_version(void) { return GYS_ABI_VERSION; }
const char *gys_last_error(void) { return g_err; }

uint32_t gys_machine_id_hash(const uint8_t machine_id[16]) { return (uint32_t)MachIdHash()(to_machid(machine_id)); }
uint32_t gys_shard_of(const uint8_t machine_id[16], uint32_t nshards) { return nshards ? gys_machine_id_hash(machine_id) % nshards : 0; }

uint64_t gys_reduce_arena_bytes(const gys_config *cfg) { return arena_layout(cfg ? cfg->max_clusters : 1, cfg && cfg->conn_pair_cms).total; }

int gys_create(const gys_config *cfg, gys_ctx **out)
try {
	if (!cfg || !out || cfg->struct_size != sizeof(gys_config)) {
		set_err("bad config (struct_size %u, expected %zu)", cfg ? cfg->struct_size : 0, sizeof(gys_config));
		return GYS_ERR_INVAL;
	}
	if (!cfg->max_hosts || !cfg->max_services || cfg->max_hosts > 65534 || !cfg->max_clusters || cfg->nranks == 0 || cfg->rank >= cfg->nranks ||
	    (cfg->svc_hll_p && (cfg->svc_hll_p < 4 || cfg->svc_hll_p > 10)) ||
	    (cfg->td_pend_cap && (cfg->td_pend_cap < 64u || cfg->td_pend_cap > GYS_TD_PEND_CAP_MAX)) ||
	    (cfg->td_buf_values && (cfg->td_buf_values < (cfg->td_pend_cap ? cfg->td_pend_cap : GYS_TD_PEND_CAP) + 64u || cfg->td_buf_values > GYS_PCAP_MAX))) {
		set_err("bad config values");
		return GYS_ERR_INVAL;
	}
	int ndev = 0;
	HIPCHK(hipGetDeviceCount(&ndev)); // fails loudly when there is no GPU: there is no CPU path
	if (ndev <= 0) {
		set_err("no HIP device");
		return GYS_ERR_HIP;
	}
	gys_ctx *c = new gys_ctx();
	for (int i = 0; i < gys_ctx::NSTAGE; ++i) c->stage_free.push_back(i);
	for (int i = 0; i < gys_ctx::RespQ::NB; ++i) c->rq.free.push_back(i);
	c->cq[0].conn = true;
	for (auto &q : c->cq)
		for (int i = 0; i < gys_ctx::RecQ::NB; ++i) q.free.push_back(i);
	c->cfg = *cfg;
	if (cfg->device >= 0) {
		c->device = cfg->device;
		HIPCHK(hipSetDevice(c->device));
	} else {
		HIPCHK(hipGetDevice(&c->device));
	}
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, c->device));
	c->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	if (cfg->stream) {
		c->stream = (hipStream_t)cfg->stream;
	} else {
		HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
		c->own_stream = true;
	}
	if (!getenv("GYS_RQ_ONE_STREAM")) HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)); // H2D copies of the host-pointer response submissions (the variable: A/B)
	const uint64_t S = cfg->max_services, H = cfg->max_hosts;
	const uint32_t cap = next_pow2(S * 2);
	int rc;
#define ALLOC(ptr, count)                        \
	if ((rc = dev_alloc(&ptr, (count))) != GYS_OK) { \
		gys_destroy(c);                          \
		return rc;                               \
	}
	ALLOC(c->lk_tbl.ent, cap);
	ALLOC(c->gid_tbl.ent, cap);
	c->lk_tbl.mask = c->gid_tbl.mask = cap - 1;
	HIPCHK(hipMemset(c->lk_tbl.ent, 0xFF, (uint64_t)cap * sizeof(TblEnt)));
	HIPCHK(hipMemset(c->gid_tbl.ent, 0xFF, (uint64_t)cap * sizeof(TblEnt)));
	ALLOC(c->svc_gid, S);
	ALLOC(c->hist_win, S);
	ALLOC(c->hist_all, S);
	ALLOC(c->bitmap, S * GYS_BM_WORDS);
	ALLOC(c->svc_ctr, S * 4);
	ALLOC(c->svc_win, S * 3);
	ALLOC(c->svc_state, S * 96);
	ALLOC(c->svc_claim, S);
	ALLOC(c->hll32, (uint64_t)1 << GYS_HLL_P);
	ALLOC(c->host_summ_win, H * 16);
	ALLOC(c->host_summ_last, H * 16);
	ALLOC(c->host_state, H);
	ALLOC(c->host_state_epoch, H);
	ALLOC(c->host_cluster, H);
	ALLOC(c->counters, 32);
	ALLOC(c->d_epoch, 4);
	ALLOC(c->misc, 16);
	ALLOC(c->svc_act, S * 4);
	ALLOC(c->topn_slot, S < 65536 ? S : 65536);
	ALLOC(c->topn_metric, S < 65536 ? S : 65536);
	ALLOC(c->dev_pcts, 64);
	c->htbl_cap = 16 * S + 64 * H; // every host's live sub-table (<= 8 L + 16 entries: a quarter full) plus its outgrown regions (< the live one)
	c->hlst_cap = 8 * S + 32 * H;
	ALLOC(c->htbl, c->htbl_cap);
	ALLOC(c->hlst, c->hlst_cap);
	c->hdesc_ext_cap = (uint32_t)(S / GYS_HOST_MAX_LOCAL) * 4u + 64u; // parts of many-listener hosts (incl. abandoned cuts)
	ALLOC(c->hdesc, H + c->hdesc_ext_cap);
	ALLOC(c->svc_host, S);
	ALLOC(c->host_spill, H);
	c->host_lst.reserve(H);
	// k_resp_host: up to 4096 sub-table entries (32 KiB) + 2048 x 24 B of per-key areas (48 KiB) + the tile image (48 or 96 KiB)
	c->resp_dyn_max = 160u * 1024u;
	HIPCHK((resp_host_lds_attr<8, false, false>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<8, true, false>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<8, true, true>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<16, false, false>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<32, false, false>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<16, true, false>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<16, true, true>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<12, false, false>(&c->resp_dyn_max))); // (GYS_TPT=12: half of the room per workgroup, two per CU)
	HIPCHK((resp_host_lds_attr<12, true, false>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<12, true, true>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<16, false, false, 1>(&c->resp_dyn_max))); // keys with candidates (bound-address listeners)
	HIPCHK((resp_host_lds_attr<16, true, false, 1>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<16, true, true, 1>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<16, false, false, 2>(&c->resp_dyn_max))); // IPv6 events
	HIPCHK((resp_host_lds_attr<16, true, false, 2>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<16, true, true, 2>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<8, false, false, 1>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<8, true, false, 1>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<8, true, true, 1>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<8, false, false, 2>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<8, true, false, 2>(&c->resp_dyn_max)));
	HIPCHK((resp_host_lds_attr<8, true, true, 2>(&c->resp_dyn_max)));
	c->host_seen.reserve(H);
	if (cfg->svc_hll_p) ALLOC(c->svc_hll, S << cfg->svc_hll_p);
	if (cfg->enable_levels) {
		ALLOC(c->lvl_snap, 2 * GYS_LEVEL_RING * S);
		ALLOC(c->lvl_last, cfg->enable_levels == 1 ? S : 1);
		ALLOC(c->lvl_last_tag, cfg->enable_levels == 1 && cfg->enable_tdigest ? S : 1);
		ALLOC(c->lvl_first, S);
		ALLOC(c->qps_hist, S);
		ALLOC(c->act_hist, S);
	}
	if (cfg->enable_tdigest) {
		const uint64_t B = cfg->max_batch_events ? cfg->max_batch_events : 1;
		// value buffer of a service: GYS_TD_PEND_CAP values wait for a merge, the rest is room for one batch's values of the service
		// (a batch that does not fit takes the slower spill path); sized to the services, within ~40 GiB unless the caller says otherwise
		c->pend_cap = cfg->td_pend_cap ? cfg->td_pend_cap : GYS_TD_PEND_CAP;
		// the fast merge class: the smallest k_digest_bins instance (merges of 1024 / 2048 / 4096 values) that leaves 128 values of room above the
		// buffer size (896 -> 1024 = GYS_TDIGEST_MERGE_FAST); a service is re-clustered early when another batch like its last would pass it
		c->merge_fast = c->pend_cap + 128u <= 1024u ? 1024u : c->pend_cap + 128u <= 2048u ? 2048u : 4096u;
		if (cfg->td_buf_values) {
			c->pcap = cfg->td_buf_values;
		} else {
			const uint64_t fit = ((40ull << 30) / (4 * S)) / 256 * 256;
			// (at least the fast merge size; with a larger buffer than the default also a quarter of the default's room on top: a service that
			// brings more values per batch than the room above td_pend_cap takes the predicted-run / spill paths every time it is nearly full)
			const uint64_t floor_cap = c->pend_cap > GYS_TD_PEND_CAP ? align_up(c->merge_fast, 256) + 256 : align_up(c->merge_fast, 256);
			c->pcap = (uint32_t)std::min<uint64_t>(GYS_PCAP_MAX, std::max<uint64_t>(floor_cap, fit));
		}
		ALLOC(c->td_sum, S * GYS_TD_NB);
		ALLOC(c->td_cnt, S * GYS_TD_NB);
		ALLOC(c->td_meta, S);
		ALLOC(c->td_minmax, S);
		ALLOC(c->td_cur, align_up(S, 64));
		ALLOC(c->td_run, S);
		if ((rc = dev_alloc(&c->td_pend, S * c->pcap, false)) != GYS_OK) { // never read before written: no clear of (up to) tens of GB
			gys_destroy(c);
			return rc;
		}
		ALLOC(c->merge_list, std::min<uint64_t>(S, B) + 1);
		ALLOC(c->merge_list_slow, S + 1); // (the all-service scan may list any service)
		// a key lands in a larger merge size class only when the batch itself brought it more than CLASS0 - PEND_CAP values
		ALLOC(c->merge_list1, std::min<uint64_t>(S, B / (c->merge_fast - c->pend_cap) + 1) + 1);
		ALLOC(c->merge_list2, std::min<uint64_t>(S, B / (GYS_MERGE_CLASS1 - c->pend_cap) + 1) + 1);
		ALLOC(c->resp_win, S);
		c->cms_nch = (uint32_t)std::min<uint64_t>(32, (S + 65535) / 65536);
		ALLOC(c->cms_partial, (uint64_t)c->cms_nch * GYS_CMS_D * GYS_CMS_W);
		HIPCHK(hipFuncSetAttribute((const void *)k_cms_partial, hipFuncAttributeMaxDynamicSharedMemorySize, GYS_CMSF_CELLS * 4));
		// (entries above merge size class 1 take the several-workgroup path: the batch itself brought such a key more than CLASS1 - PEND_CAP values)
		c->huge_list_cap = std::min<uint64_t>(S, B / (GYS_MERGE_CLASS1 - c->pend_cap) + 1);
		ALLOC(c->huge_list, c->huge_list_cap + 1);
		ALLOC(c->query_list, 4);
		ALLOC(c->merge_count, 16);
		ALLOC(c->query_sum, GYS_TD_NB);
		ALLOC(c->query_cnt, GYS_TD_NB);
		ALLOC(c->batch_cnt, align_up(S, 16));
		ALLOC(c->batch_off, align_up(S, 16));
		ALLOC(c->scan_block_sums, (S + GYS_SCAN_TILE - 1) / GYS_SCAN_TILE + 1);
		ALLOC(c->ev_kv, B);
		c->ev_kv_cap = B;
		// `staged`: the runs of one batch.  With predicted runs (k_prespill) a batch may need the predicted runs (the last batch's counts
		// plus a quarter, + 64 per key) AND the exact runs of the keys the prediction missed (<= B): 5/2 B + slack, while indices stay 32-bit
		c->prespill = getenv("GYS_NO_PRESPILL") == nullptr && B * 5 / 2 + (1u << 24) < (1ull << 32);
		c->staged_cap = c->prespill ? B * 5 / 2 + (1u << 24) : B;
		if ((rc = dev_alloc(&c->staged, c->staged_cap, false)) != GYS_OK) { // (runs are written before they are read)
			gys_destroy(c);
			return rc;
		}
		if (c->prespill) {
			ALLOC(c->td_run0, S);
			ALLOC(c->td_run1, S);
			ALLOC(c->td_prevm, S);
			ALLOC(c->pre_hot, 2);
			ALLOC(c->host_batch, H);
			c->append_cap = S + 1; // (a key has at most one entry per batch; the keys of a batch are not bounded by its size: predictions come from EARLIER batches)
			ALLOC(c->append_list, c->append_cap);
		}
		c->huge_blocks = (int)std::min<uint64_t>(64, std::min<uint64_t>(S, B / GYS_MERGE_LDS_MAX + 1));
		if (c->huge_blocks < 1) c->huge_blocks = 1;
		// the several-workgroup path's pool shares the scratch: 64 KiB of bins per entry, up to 262 144 entries (16 GiB of 288) when the batches can
		// carry that many large keys; the one-workgroup fallback needs huge_blocks x 4 MiB of it
		c->huge_maxent = (uint32_t)std::max<uint64_t>((uint64_t)c->huge_blocks * GYS_HUGE_BINS / GYS_HB_BINS, std::min<uint64_t>(c->huge_list_cap, 262144)); // <= 16 GiB of bins: one round for a 2^29-event batch (167 773 possible entries) instead of eleven sets of empty launches
		const uint64_t scratch_entries = c->huge_maxent;
		if (const char *e = getenv("GYS_HUGE_MAXENT")) { // tests: a small pool, so that a modest batch walks its large keys in several rounds
			const long v = atol(e);
			if (v >= 1 && (uint64_t)v < c->huge_maxent) c->huge_maxent = (uint32_t)v;
		}
		ALLOC(c->huge_scratch, scratch_entries * GYS_HB_BINS);
		ALLOC(c->huge_acc, (uint64_t)c->huge_maxent * GYS_HB_ACC);
		ALLOC(c->huge_bm, (uint64_t)c->huge_maxent * GYS_BM_WORDS);
		ALLOC(c->huge_chunk_off, (uint64_t)c->huge_maxent + 1);
		ALLOC(c->huge_tail, (uint64_t)1 << 20);
		ALLOC(c->huge_tb_list, (uint64_t)c->huge_maxent);
		ALLOC(c->huge_fb_list, std::min<uint64_t>(S, B / (GYS_MERGE_CLASS1 - c->pend_cap) + 1) + 1);
		HIPCHK(hipFuncSetAttribute((const void *)k_huge_count, hipFuncAttributeMaxDynamicSharedMemorySize, GYS_HB_BINS * 4));
		HIPCHK(hipFuncSetAttribute((const void *)k_huge_merge<512, GYS_HB_TAIL_A, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (GYS_HB_BINS + GYS_HB_TAIL_A) * 4));
		HIPCHK(hipFuncSetAttribute((const void *)k_huge_merge<1024, GYS_HB_TAIL_LDS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (GYS_HB_BINS + GYS_HB_TAIL_LDS) * 4));
	}
	ALLOC(c->last_act32, 2ull * GYS_CMS_D * GYS_CMS_W);
	ALLOC(c->last_act64, 2ull * GYS_CMS_D * GYS_CMS_W);
	ALLOC(c->ring_act32, 2ull * GYS_ACT_RING * GYS_CMS_D * GYS_CMS_W);
	ALLOC(c->ring_act64, 2ull * GYS_ACT_RING * GYS_CMS_D * GYS_CMS_W);
	ALLOC(c->act_live, 2);
#undef ALLOC
	// dev_alloc zeroes with hipMemset on the NULL stream, which may still be in flight; the context stream is non-blocking, so the
	// initialisation kernels / copies below must not start before every one of those clears has landed
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(c->d_epoch, &c->epoch, 4, hipMemcpyHostToDevice));
	if (cfg->enable_tdigest) {
		hipLaunchKernelGGL(k_minmax_init, dim3(grid_for(S, 256, 2048)), dim3(256), 0, c->stream, c->td_minmax, S);
		static const uint32_t one = 1; // static: the source of an async copy must outlive the call
		HIPCHK(hipMemcpyAsync(c->merge_count + 8, &one, 4, hipMemcpyHostToDevice, c->stream));
	}
	hipLaunchKernelGGL(k_hist_init, dim3(grid_for(S, 256, 2048)), dim3(256), 0, c->stream, c->hist_win, (uint64_t)0, S, (int64_t)INT64_MIN);
	hipLaunchKernelGGL(k_hist_init, dim3(grid_for(S, 256, 2048)), dim3(256), 0, c->stream, c->hist_all, (uint64_t)0, S, (int64_t)INT64_MIN);
	if (cfg->enable_levels) {
		if (cfg->enable_levels == 1) hipLaunchKernelGGL(k_hist_init, dim3(grid_for(S, 256, 2048)), dim3(256), 0, c->stream, c->lvl_last, (uint64_t)0, S, (int64_t)INT64_MIN);
		if (cfg->enable_levels == 1 && cfg->enable_tdigest) HIPCHK(hipMemsetAsync(c->lvl_last_tag, 0xFF, (size_t)S * 4, c->stream)); // (no window closed yet)
		c->lvl_last_epoch = 0xFFFFFFFEu;
		// GY_HISTOGRAM<int, ...>: max_val_seen_ starts at std::numeric_limits<int>::min()
		hipLaunchKernelGGL(k_hist_init, dim3(grid_for(S, 256, 2048)), dim3(256), 0, c->stream, c->qps_hist, (uint64_t)0, S, (int64_t)INT32_MIN);
		hipLaunchKernelGGL(k_hist_init, dim3(grid_for(S, 256, 2048)), dim3(256), 0, c->stream, c->act_hist, (uint64_t)0, S, (int64_t)INT32_MIN);
	}
	c->al = arena_layout(cfg->max_clusters, cfg->conn_pair_cms != 0);
	if (cfg->reduce_arena) {
		if (cfg->reduce_arena_bytes < c->al.total) {
			set_err("reduce_arena too small: %llu < %llu", (unsigned long long)cfg->reduce_arena_bytes, (unsigned long long)c->al.total);
			gys_destroy(c);
			return GYS_ERR_INVAL;
		}
		c->arena = (uint8_t *)cfg->reduce_arena;
	} else {
		HIPCHK(hipMalloc((void **)&c->arena, c->al.total));
		c->own_arena = true;
	}
	HIPCHK(hipMalloc((void **)&c->last, c->al.total));
	HIPCHK(hipMemsetAsync(c->arena, 0, c->al.total, c->stream));
	HIPCHK(hipMemsetAsync(c->last, 0, c->al.total, c->stream));
	{
		HIPCHK(hipMemcpyAsync(c->arena + c->al.off_i64max, &c->i64min, 8, hipMemcpyHostToDevice, c->stream));
	}
	HIPCHK(hipStreamSynchronize(c->stream));
	*out = c;
	return GYS_OK;
} GYS_CATCH_ALL

void gys_destroy(gys_ctx *c)
{
	if (c) (void)hipSetDevice(c->device);
	if (!c) return;
	if (c->rq.flusher_on) {
		{
			std::lock_guard<std::mutex> g(c->rq.mu);
			c->rq.stop = true;
		}
		c->rq.fcv.notify_all();
		if (c->rq.flusher.joinable()) c->rq.flusher.join();
		c->rq.flusher_on = false;
	}
	for (auto &q : c->cq) {
		if (!q.flusher_on) continue;
		{
			std::lock_guard<std::mutex> g(q.mu);
			q.stop = true;
		}
		q.fcv.notify_all();
		if (q.flusher.joinable()) q.flusher.join();
		q.flusher_on = false;
	}
	if (c->stream) hipStreamSynchronize(c->stream);
	if (c->win_graph_exec) hipGraphExecDestroy(c->win_graph_exec);
	if (c->win_graph) hipGraphDestroy(c->win_graph);
	for (auto &cg : c->close_graph) {
		if (cg.x) hipGraphExecDestroy(cg.x);
		if (cg.g) hipGraphDestroy(cg.g);
	}
	for (auto &sl : c->seg_ring) {
		if (sl.host) hipHostFree(sl.host);
		if (sl.dev) hipFree(sl.dev);
		if (sl.xhost) hipHostFree(sl.xhost);
		if (sl.xdev) hipFree(sl.xdev);
		if (sl.done) hipEventDestroy(sl.done);
	}
	for (auto &b : c->rq.b) {
		if (b.h) hipHostFree(b.h);
		if (b.d) hipFree(b.d);
		if (b.done) hipEventDestroy(b.done);
		if (b.copied) hipEventDestroy(b.copied);
	}
	for (auto &q : c->cq)
		for (auto &b : q.b) {
			if (b.h) hipHostFree(b.h);
			if (b.d) hipFree(b.d);
			if (b.done) hipEventDestroy(b.done);
			if (b.copied) hipEventDestroy(b.copied);
		}
	if (c->copy_stream) hipStreamDestroy(c->copy_stream);
	for (auto &st : c->stage) {
		if (st.h) hipHostFree(st.h);
		if (st.d) hipFree(st.d);
		if (st.done) hipEventDestroy(st.done);
		if (st.copied) hipEventDestroy(st.copied);
	}
	prof_resolve(c);
	void *ptrs[] = {c->lk_tbl.ent, c->gid_tbl.ent, c->svc_gid, c->hist_win, c->hist_all, c->bitmap, c->td_sum,
			c->td_cnt, c->td_meta, c->td_minmax, c->td_pend, c->td_cur, c->td_run, c->td_run0, c->td_run1, c->td_prevm, c->pre_hot, c->host_batch, c->append_list, c->svc_host, c->host_spill, c->merge_list, c->merge_list_slow, c->merge_list1, c->merge_list2, c->resp_win, c->cms_partial, c->huge_list, c->query_list, c->merge_count, c->query_sum, c->query_cnt,
			c->batch_cnt, c->batch_off, c->scan_block_sums, c->ev_kv, c->staged, c->huge_scratch, c->huge_acc, c->huge_tail, c->huge_tb_list, c->huge_bm, c->huge_chunk_off, c->huge_fb_list, c->hll32, c->svc_ctr, c->svc_win, c->svc_state, c->svc_claim, c->svc_hll, c->host_summ_win, c->host_summ_last, c->host_state,
			c->host_state_epoch, c->host_cluster, c->counters, c->misc, c->htbl, c->hlst, c->hdesc, c->wire_jump[0], c->wire_jump[1], c->wire_cnt,
			c->wire_rank, c->wire_bsums, c->wire_status, c->wire_mark, c->wire_flags, c->wire_msgs, c->last, c->last_act32, c->last_act64, c->ring_act32, c->ring_act64, c->act_live, c->q_cand_key, c->q_out_keys, c->q_cand_slot, c->q_misc, c->q_host_mask, c->q_slot_list, c->q_set, c->q_out_rows, c->q_acc, c->q_cnt, c->dev_staging, c->dev_offsets, c->csr_off, c->csr_mem, c->svc_act, c->d_epoch, c->topn_slot,
			c->topn_metric, c->dev_pcts, c->zipf_cdf, c->lvl_snap, c->lvl_last, c->lvl_last_tag, c->svc_bithist, c->rb_bins, c->rb_host_members, c->rb_host_chunks, c->lvl_first, c->qps_hist, c->act_hist, c->cand_pool, c->own_arena ? c->arena : nullptr};
	for (void *p : ptrs)
		if (p) hipFree(p);
	if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
	delete c;
}

int gys_sync(gys_ctx *c)
try {
	GYS_ENTER(c);
	if (!c) return GYS_ERR_INVAL;
	HIPCHK(hipStreamSynchronize(c->stream));
	return GYS_OK;
} GYS_CATCH_ALL

// ------------------------------------------------------------------------------------------------ registration
int gys_register_cluster(gys_ctx *c, const char *cluster_name, uint32_t *cluster_idx)
try {
	GYS_ENTER(c);
	if (!c || !cluster_name) return GYS_ERR_INVAL;
	auto it = c->cluster_map.find(cluster_name);
	if (it == c->cluster_map.end()) {
		if (c->cluster_names.size() >= c->cfg.max_clusters) {
			set_err("max_clusters exhausted");
			return GYS_ERR_NOMEM;
		}
		const uint32_t idx = (uint32_t)c->cluster_names.size();
		c->cluster_names.emplace_back(cluster_name);
		it = c->cluster_map.emplace(cluster_name, idx).first;
	}
	if (cluster_idx) *cluster_idx = it->second;
	return GYS_OK;
} GYS_CATCH_ALL

int gys_register_host(gys_ctx *c, const uint8_t machine_id[16], const char *cluster_name, uint32_t *host_slot)
try {
	GYS_ENTER(c);
	if (!c || !machine_id) return GYS_ERR_INVAL;
	int rc = check_owner(c, machine_id);
	if (rc) return rc;
	uint32_t cidx = 0;
	rc = gys_register_cluster(c, cluster_name ? cluster_name : "", &cidx);
	if (rc) return rc;
	const MachId m = to_machid(machine_id);
	auto it = c->host_map.find(m);
	uint32_t slot;
	if (it != c->host_map.end()) {
		slot = it->second;
	} else {
		if (c->hosts.size() >= c->cfg.max_hosts) {
			set_err("max_hosts exhausted");
			return GYS_ERR_NOMEM;
		}
		slot = (uint32_t)c->hosts.size();
		c->hosts.push_back(m);
		c->host_cluster_h.push_back(cidx);
		c->host_map.emplace(m, slot);
		c->host_lst.emplace_back();
		c->host_names.emplace_back();
		c->host_lst.back().tbl.assign(16, GYS_HOST_TBL_EMPTY);
		c->host_seen.push_back(0);
		rc = host_lst_upload(c, slot);
		if (rc) return rc;
	}
	c->host_cluster_h[slot] = cidx;
	HIPCHK(hipMemcpyAsync(c->host_cluster + slot, &cidx, 4, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (host_slot) *host_slot = slot;
	return GYS_OK;
} GYS_CATCH_ALL

int gys_register_listeners(gys_ctx *c, const uint8_t machine_id[16], const gys_listener_info *arr_in, uint32_t n_in, uint32_t *first_slot)
try {
	GYS_ENTER(c);
	if (!c || !machine_id || (!arr_in && n_in)) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	// a partha resends NEW_LISTENER after a reconnect: a glob_id the engine already knows keeps its slot (and its histograms, digest,
	// counters); repeats inside one call are dropped as well, so that the device tables never see the same key twice
	std::vector<gys_listener_info> fresh;
	const gys_listener_info *arr = arr_in;
	uint32_t n = n_in;
	{
		std::unordered_set<uint64_t> seen;
		bool dup = false;
		for (uint32_t i = 0; i < n_in && !dup; ++i) dup = c->gid_map_h.count(arr_in[i].glob_id) != 0 || !seen.insert(arr_in[i].glob_id).second;
		if (dup) {
			seen.clear();
			for (uint32_t i = 0; i < n_in; ++i)
				if (!c->gid_map_h.count(arr_in[i].glob_id) && seen.insert(arr_in[i].glob_id).second) fresh.push_back(arr_in[i]);
			arr = fresh.data();
			n = (uint32_t)fresh.size();
		}
	}
	if ((uint64_t)c->nsvc + n > c->cfg.max_services) {
		set_err("max_services exhausted");
		return GYS_ERR_NOMEM;
	}
	if (first_slot) *first_slot = c->nsvc;
	if (!n) return GYS_OK;
	std::vector<uint64_t> keys(2 * (size_t)n);
	for (uint32_t i = 0; i < n; ++i) {
		if (arr[i].glob_id == GYS_EMPTY_KEY) {
			set_err("glob_id ~0 is reserved");
			return GYS_ERR_INVAL;
		}
		keys[i] = arr[i].glob_id;
		keys[n + i] = listener_key(host, arr[i].netns, arr[i].port);
	}
	rc = ensure_staging(c, keys.size() * 8, 0);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(c->dev_staging, keys.data(), keys.size() * 8, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(c->svc_gid + c->nsvc, c->dev_staging, (size_t)n * 8, hipMemcpyDeviceToDevice, c->stream));
	HIPCHK(hipMemsetAsync(c->misc, 0, 4, c->stream));
	hipLaunchKernelGGL(k_table_insert, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->gid_tbl, (const uint64_t *)c->dev_staging, c->nsvc, n, c->misc);
	hipLaunchKernelGGL(k_table_insert, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->lk_tbl, (const uint64_t *)c->dev_staging + n, c->nsvc, n,
			   c->misc);
	uint32_t nfail = 0;
	HIPCHK(hipMemcpyAsync(&nfail, c->misc, 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (nfail) {
		set_err("key table full (%u inserts failed)", nfail);
		return GYS_ERR_NOMEM;
	}
	{
		std::vector<uint32_t> hs(n, host);
		HIPCHK(hipMemcpyAsync(c->svc_host + c->nsvc, hs.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	rc = host_lst_add(c, host, arr, n, c->nsvc);
	if (rc) return rc;
	for (uint32_t i = 0; i < n; ++i) {
		c->gid_map_h[arr[i].glob_id] = c->nsvc + i;
		c->host_lst[host].all_slots.push_back(c->nsvc + i);
		std::array<char, 16> cm{};
		memcpy(cm.data(), arr[i].comm, 16);
		c->svc_comm.push_back(cm);
		c->svc_gid_h.push_back(arr[i].glob_id);
		c->svc_port_h.push_back(arr[i].port);
	}
	c->nsvc += n;
	return GYS_OK;
} GYS_CATCH_ALL

// ------------------------------------------------------------------------------------------------ ingest
int gys_ingest_resp_events_dev(gys_ctx *c, const gys_resp_seg *segs, uint32_t nsegs, const void *d_ev24, uint64_t nevents)
try {
	GYS_ENTER(c);
	if (!c || (!d_ev24 && nevents)) return GYS_ERR_INVAL;
	return run_resp_batch(c, segs, nsegs, d_ev24, nevents);
} GYS_CATCH_ALL

int gys_ingest_resp_events(gys_ctx *c, const uint8_t machine_id[16], const void *ev24, uint32_t nevents)
try {
	GYS_ENTER_NOFLUSH(c);
	if (!c || !machine_id || (!ev24 && nevents)) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	if (!nevents) return GYS_OK;
	static const bool no_queue = getenv("GYS_NO_RESP_QUEUE") != nullptr; // A/B: one submission per call through the staging ring
	if (!no_queue && (uint64_t)nevents <= std::min<uint64_t>(GYS_RQ_EVENTS, c->cfg.max_batch_events))
		return rq_ingest(c, host, ev24, nevents); // combined with the other callers' pending calls (gys_ctx::RespQ)
	// larger than a combined batch: its own submission (after whatever the queue holds: same-host calls keep their order)
	rc = rq_flush(c);
	if (rc) return rc;
	const uint64_t bytes = (uint64_t)nevents * 24;
	int si;
	rc = stage_acquire(c, bytes, &si);
	if (rc) return rc;
	gys_ctx::Stage &st = c->stage[si];
	memcpy(st.h, ev24, bytes); // the caller's buffer is free from here on
	{
		std::lock_guard<std::mutex> g(c->enq_mu);
		hipError_t e = hipMemcpyAsync(st.d, st.h, bytes, hipMemcpyHostToDevice, c->stream);
		if (e == hipSuccess) {
			gys_resp_seg seg{host, 0, 0};
			rc = run_resp_batch(c, &seg, 1, st.d, nevents);
			e = hipEventRecord(st.done, c->stream);
		}
		if (e != hipSuccess) {
			set_err("gys_ingest_resp_events: %s", hipGetErrorString(e));
			rc = GYS_ERR_HIP;
		}
	}
	stage_release(c, si);
	return rc;
} GYS_CATCH_ALL

int gys_ingest_resp_events_v6_dev(gys_ctx *c, const gys_resp_seg *segs, uint32_t nsegs, const void *d_ev48, uint64_t nevents)
try {
	GYS_ENTER(c);
	if (!c || (!d_ev48 && nevents) || ((uintptr_t)d_ev48 & 7u)) return GYS_ERR_INVAL;
	return run_resp_batch(c, segs, nsegs, d_ev48, nevents, true);
} GYS_CATCH_ALL

int gys_ingest_resp_events_v6(gys_ctx *c, const uint8_t machine_id[16], const void *ev48, uint32_t nevents)
try {
	GYS_ENTER_NOFLUSH(c);
	if (!c || !machine_id || (!ev48 && nevents)) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	if (!nevents) return GYS_OK;
	// its own submission, behind whatever the IPv4 queue holds (a service's per-call value multisets keep the order of the calls)
	rc = rq_flush(c);
	if (rc) return rc;
	const uint64_t bytes = (uint64_t)nevents * 48;
	int si;
	rc = stage_acquire(c, bytes, &si);
	if (rc) return rc;
	gys_ctx::Stage &st = c->stage[si];
	memcpy(st.h, ev48, bytes); // the caller's buffer is free from here on
	{
		std::lock_guard<std::mutex> g(c->enq_mu);
		hipError_t e = hipMemcpyAsync(st.d, st.h, bytes, hipMemcpyHostToDevice, c->stream);
		if (e == hipSuccess) {
			gys_resp_seg seg{host, 0, 0};
			rc = run_resp_batch(c, &seg, 1, st.d, nevents, true);
			e = hipEventRecord(st.done, c->stream);
		}
		if (e != hipSuccess) {
			set_err("gys_ingest_resp_events_v6: %s", hipGetErrorString(e));
			rc = GYS_ERR_HIP;
		}
	}
	stage_release(c, si);
	return rc;
} GYS_CATCH_ALL

int gys_ingest_tcp_conn_dev(gys_ctx *c, const void *d_batch, const uint32_t *d_offsets, uint32_t nconns)
try {
	GYS_ENTER(c);
	if (!c || ((!d_batch || !d_offsets) && nconns)) return GYS_ERR_INVAL;
	return run_conn(c, (const uint8_t *)d_batch, d_offsets, nconns);
} GYS_CATCH_ALL

int gys_ingest_tcp_conn(gys_ctx *c, const uint8_t machine_id[16], const void *batch, uint32_t nconns, const void *pend)
try {
	GYS_ENTER_NOFLUSH(c);
	if (!c || !machine_id || (!batch && nconns) || ((uintptr_t)batch & 7u)) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	if (!nconns) return GYS_OK;
	std::vector<uint32_t> offs;
	// TCP_CONN_NOTIFY::get_elem_size common/gy_comm_proto.h:1721-1724
	rc = walk_batch((const uint8_t *)batch, nconns, (const uint8_t *)pend, 280, [](const uint8_t *p) {
		uint16_t cl;
		memcpy(&cl, p + 272, 2);
		return (uint32_t)(280u + cl + p[279]);
	}, offs);
	if (rc) return rc;
	if (offs.empty()) return GYS_OK;
	return recq_ingest(c, c->cq[0], host, batch, (uint64_t)((const uint8_t *)pend - (const uint8_t *)batch), offs);
} GYS_CATCH_ALL

int gys_ingest_listener_state_dev(gys_ctx *c, const void *d_batch, const uint32_t *d_offsets, const uint32_t *d_host_slot, uint32_t nrecs)
try {
	GYS_ENTER(c);
	if (!c || ((!d_batch || !d_offsets || !d_host_slot) && nrecs)) return GYS_ERR_INVAL;
	return run_lstate(c, (const uint8_t *)d_batch, d_offsets, d_host_slot, 0, nrecs);
} GYS_CATCH_ALL

int gys_ingest_active_conns_dev(gys_ctx *c, const void *d_batch, uint32_t nitems)
try {
	GYS_ENTER(c);
	if (!c || (!d_batch && nitems)) return GYS_ERR_INVAL;
	return run_actconn(c, (const uint8_t *)d_batch, nitems);
} GYS_CATCH_ALL

int gys_ingest_active_conns(gys_ctx *c, const uint8_t machine_id[16], const void *batch, uint32_t nitems, const void *pend)
try {
	GYS_ENTER_NOFLUSH(c);
	if (!c || !machine_id || (!batch && nitems) || ((uintptr_t)batch & 7u)) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	// fixed 104-byte stride; nitems is an upper bound as in the reference's loop over pconn (gy_mconnhdlr.cc:7842)
	const uint64_t avail = pend ? (uint64_t)((const uint8_t *)pend - (const uint8_t *)batch) / 104u : nitems;
	const uint32_t n = (uint32_t)std::min<uint64_t>(nitems, avail);
	if (!n) return GYS_OK;
	const uint64_t bytes = (uint64_t)n * 104u;
	int si;
	rc = stage_acquire(c, bytes, &si);
	if (rc) return rc;
	gys_ctx::Stage &st = c->stage[si];
	memcpy(st.h, batch, bytes);
	{
		std::lock_guard<std::mutex> g(c->enq_mu);
		hipError_t e = hipMemcpyAsync(st.d, st.h, bytes, hipMemcpyHostToDevice, c->stream);
		if (e == hipSuccess) {
			rc = run_actconn(c, st.d, n);
			e = hipEventRecord(st.done, c->stream);
		}
		if (e != hipSuccess) {
			set_err("gys_ingest_active_conns: %s", hipGetErrorString(e));
			rc = GYS_ERR_HIP;
		}
	}
	stage_release(c, si);
	return rc;
} GYS_CATCH_ALL

int gys_ingest_listener_state(gys_ctx *c, const uint8_t machine_id[16], const void *batch, uint32_t nrecs, const void *pend)
try {
	GYS_ENTER_NOFLUSH(c);
	if (!c || !machine_id || (!batch && nrecs) || ((uintptr_t)batch & 7u)) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	if (!nrecs) return GYS_OK;
	std::vector<uint32_t> offs;
	// LISTENER_STATE_NOTIFY::get_elem_size common/gy_comm_proto.h:2229-2232
	rc = walk_batch((const uint8_t *)batch, nrecs, (const uint8_t *)pend, 88, [](const uint8_t *p) { return (uint32_t)(88u + p[85] + p[86]); }, offs);
	if (rc) return rc;
	if (offs.empty()) return GYS_OK;
	return recq_ingest(c, c->cq[1], host, batch, (uint64_t)((const uint8_t *)pend - (const uint8_t *)batch), offs);
} GYS_CATCH_ALL

// ------------------------------------------------------------------------------------------------ wire front-end
// COMM_HEADER / EVENT_NOTIFY framing as an unmodified partha sends it to madhava (common/gy_comm_proto.h:336-420, :486-500).  The
// 16-byte message headers are validated on the host exactly like COMM_HEADER::validate (common/gy_comm_proto.cc:10-57); the records
// are walked, checked and indexed on the GPU (k_wire_*), then handed to the same ingest kernels as the batch entry points.
namespace {
constexpr uint32_t PM_HDR_MAGIC = 0x05666605u;      // COMM_HEADER::PM_HDR_MAGIC, partha to madhava
constexpr uint32_t COMM_EVENT_NOTIFY_T = 14;        // COMM_TYPE_E::COMM_EVENT_NOTIFY
constexpr uint32_t COMM_MIN_TYPE_T = 1, COMM_MAX_TYPE_T = 18;
constexpr uint32_t MAX_COMM_DATA_SZ_T = 16u << 20;  // gy_comm_proto.h:31
constexpr uint32_t NOTIFY_LISTENER_STATE_T = 0x309, NOTIFY_TCP_CONN_T = 0x30C; // NOTIFY_TYPE_E (0x301 + 8, + 11)
constexpr uint32_t MAX_NUM_CONNS_T = 2048, MAX_NUM_LISTENERS_T = 512;          // gy_comm_proto.h:1711, :2222

int wire_reserve(gys_ctx *c, uint64_t nslots, uint32_t nmsgs)
{
	if (nslots + 1 > c->wire_slots_cap) {
		HIPCHK(hipStreamSynchronize(c->stream));
		void *old[] = {c->wire_jump[0], c->wire_jump[1], c->wire_cnt, c->wire_rank, c->wire_bsums, c->wire_mark, c->wire_flags};
		for (void *p : old)
			if (p) hipFree(p);
		c->wire_jump[0] = c->wire_jump[1] = c->wire_cnt = c->wire_rank = c->wire_bsums = nullptr;
		c->wire_mark = c->wire_flags = nullptr;
		const uint64_t cap = align_up(std::max<uint64_t>(nslots + 1, 1u << 16), 4096);
		HIPCHK(hipMalloc((void **)&c->wire_jump[0], cap * 4));
		HIPCHK(hipMalloc((void **)&c->wire_jump[1], cap * 4));
		HIPCHK(hipMalloc((void **)&c->wire_cnt, cap * 4));
		HIPCHK(hipMalloc((void **)&c->wire_rank, cap * 4));
		HIPCHK(hipMalloc((void **)&c->wire_bsums, (cap / GYS_SCAN_TILE + 2) * 4));
		HIPCHK(hipMalloc((void **)&c->wire_mark, cap));
		HIPCHK(hipMalloc((void **)&c->wire_flags, cap));
		c->wire_slots_cap = cap;
	}
	if (!c->wire_status) HIPCHK(hipMalloc((void **)&c->wire_status, 16));
	if (nmsgs > c->wire_msgs_cap) {
		if (c->wire_msgs) {
			HIPCHK(hipStreamSynchronize(c->stream));
			HIPCHK(hipFree(c->wire_msgs));
		}
		c->wire_msgs_cap = std::max<uint32_t>(nmsgs, 1024);
		HIPCHK(hipMalloc((void **)&c->wire_msgs, (uint64_t)c->wire_msgs_cap * sizeof(WireMsg)));
	}
	return GYS_OK;
}

// d_buf: device copy of the stream (8-byte aligned); msgs: accepted messages of ONE record kind, sorted by position; fills
// c->dev_offsets[0 .. total records) with the records' byte offsets inside d_buf
int wire_decode(gys_ctx *c, const uint8_t *d_buf, uint64_t nbytes, std::vector<WireMsg> &msgs, uint32_t nrec_total)
{
	const uint32_t nslots = (uint32_t)(nbytes / 8) + 1; // + one terminal slot past the end
	const uint32_t nmsgs = (uint32_t)msgs.size();
	int rc = wire_reserve(c, nslots, nmsgs);
	if (rc) return rc;
	rc = ensure_staging(c, 0, nrec_total);
	if (rc) return rc;
	uint32_t maxev = 1;
	for (const WireMsg &m : msgs) maxev = std::max(maxev, m.nevents);
	HIPCHK(hipMemcpyAsync(c->wire_msgs, msgs.data(), (size_t)nmsgs * sizeof(WireMsg), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemsetAsync(c->wire_mark, 0, nslots, c->stream));
	HIPCHK(hipMemsetAsync(c->wire_status, 0, 16, c->stream));
	const dim3 gs((nslots + 255) / 256), gm((nmsgs + 255) / 256), b(256);
	ProfScope ps(c, "wire_decode");
	hipLaunchKernelGGL(k_wire_next, gs, b, 0, c->stream, (const uint64_t *)d_buf, c->wire_msgs, nmsgs, nslots, c->wire_jump[0], c->wire_flags);
	hipLaunchKernelGGL(k_wire_seed, gm, b, 0, c->stream, c->wire_msgs, nmsgs, c->wire_mark);
	int cur = 0;
	for (uint32_t span = 1; span < maxev; span <<= 1) { // after the round for span, the first 2 * span records of every chain are marked
		hipLaunchKernelGGL(k_wire_round, gs, b, 0, c->stream, nslots, c->wire_jump[cur], c->wire_jump[cur ^ 1], c->wire_mark);
		cur ^= 1;
	}
	hipLaunchKernelGGL(k_wire_count, gs, b, 0, c->stream, nslots, c->wire_mark, c->wire_flags, c->wire_cnt);
	const uint32_t nblk = (nslots + GYS_SCAN_TILE - 1) / GYS_SCAN_TILE;
	hipLaunchKernelGGL(k_scan_block_sums, dim3(nblk), b, 0, c->stream, c->wire_cnt, nslots, c->wire_bsums);
	hipLaunchKernelGGL(k_scan_top, dim3(1), b, 0, c->stream, c->wire_bsums, nblk);
	hipLaunchKernelGGL(k_scan_final, dim3(nblk), b, 0, c->stream, c->wire_cnt, nslots, c->wire_bsums, c->wire_rank);
	hipLaunchKernelGGL(k_wire_emit, gs, b, 0, c->stream, c->wire_msgs, nmsgs, nslots, c->wire_cnt, c->wire_rank, c->wire_flags, c->dev_offsets, c->wire_status);
	hipLaunchKernelGGL(k_wire_check, gm, b, 0, c->stream, c->wire_msgs, nmsgs, nslots, c->wire_cnt, c->wire_rank, c->wire_status);
	HIPCHK(hipGetLastError());
	uint32_t status = 0;
	HIPCHK(hipMemcpyAsync(&status, c->wire_status, 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (status) {
		set_err("malformed message: %s%s", (status & 1u) ? "record size / padding invalid or element overruns the message; " : "",
			(status & 2u) ? "fewer records than nevents_" : "");
		return GYS_ERR_INVAL;
	}
	return GYS_OK;
}
} // namespace

int gys_ingest_comm_stream(gys_ctx *c, const uint8_t machine_id[16], const void *buf, uint64_t nbytes, gys_comm_stats *out)
try {
	GYS_ENTER(c);
	if (!c || !machine_id || (!buf && nbytes) || ((uintptr_t)buf & 7u)) return GYS_ERR_INVAL; // COMM_HEADER::validate: 8-byte aligned data
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	gys_comm_stats st{};
	if (nbytes >= (1ull << 32)) {
		set_err("stream too large (u32 offsets)");
		return GYS_ERR_INVAL;
	}
	std::vector<WireMsg> conn_msgs, lst_msgs;
	uint32_t nconn = 0, nlst = 0;
	const uint8_t *p = (const uint8_t *)buf, *pend = p + nbytes;
	while ((size_t)(pend - p) >= 16) {
		uint32_t magic, total_sz, data_type, padding_sz;
		memcpy(&magic, p, 4);
		memcpy(&total_sz, p + 4, 4);
		memcpy(&data_type, p + 8, 4);
		memcpy(&padding_sz, p + 12, 4);
		// COMM_HEADER::validate common/gy_comm_proto.cc:12-22
		if (!(magic == PM_HDR_MAGIC && total_sz < MAX_COMM_DATA_SZ_T && total_sz >= 16 && padding_sz < 8 && data_type > COMM_MIN_TYPE_T &&
		      data_type < COMM_MAX_TYPE_T) || (total_sz & 7u)) {
			st.nmsgs_invalid++;
			if (out) *out = st;
			set_err("invalid COMM_HEADER at byte %zu", (size_t)(p - (const uint8_t *)buf));
			return GYS_ERR_INVAL; // the reference terminates the connection
		}
		// a well-formed header whose body has not arrived yet (a recv() chunk ends inside the message): end of input, the whole
		// messages before it are ingested and bytes_consumed tells the caller where to resume (gysketch.h)
		if (total_sz > (uint64_t)(pend - p)) break;
		st.nmsgs++;
		const uint32_t act = total_sz - padding_sz;
		bool taken = false;
		if (data_type == COMM_EVENT_NOTIFY_T) {
			if (act < 16 + 8) {
				st.nmsgs_invalid++;
				if (out) *out = st;
				set_err("EVENT_NOTIFY message shorter than its headers");
				return GYS_ERR_INVAL;
			}
			uint32_t subtype, nevents;
			memcpy(&subtype, p + 16, 4);
			memcpy(&nevents, p + 20, 4);
			if (subtype == NOTIFY_TCP_CONN_T || subtype == NOTIFY_LISTENER_STATE_T) {
				const bool is_conn = subtype == NOTIFY_TCP_CONN_T;
				if (nevents > (is_conn ? MAX_NUM_CONNS_T : MAX_NUM_LISTENERS_T)) { // :853, :968
					st.nmsgs_invalid++;
					if (out) *out = st;
					set_err("nevents_ %u over the per-message limit", nevents);
					return GYS_ERR_INVAL;
				}
				// every record is at least its fixed part, so a message that announces more records than its payload can hold fails the
				// validators' walk (:859-880, :974-995) whatever its bytes say.  Rejected HERE, before any kernel: the offset list is sized by
				// the sum of the announced counts, and a long stream of header-only messages announcing 2 048 records each could wrap that sum
				if ((uint64_t)nevents * (is_conn ? 280u : 88u) > (uint64_t)act - 24u) {
					st.nmsgs_invalid++;
					if (out) *out = st;
					set_err("nevents_ %u records cannot lie in %u payload bytes", nevents, act - 24u);
					return GYS_ERR_INVAL;
				}
				WireMsg m{};
				m.pay_slot = (uint32_t)((p - (const uint8_t *)buf) + 24) / 8u;
				m.end_slot = (uint32_t)((p - (const uint8_t *)buf) + act) / 8u; // act_len is a multiple of 8 for well-formed messages
				m.nevents = nevents;
				m.kind = is_conn ? 0u : 1u;
				if (is_conn) {
					m.out_base = nconn;
					nconn += nevents;
					conn_msgs.push_back(m);
					st.nmsgs_tcp_conn++;
				} else {
					m.out_base = nlst;
					nlst += nevents;
					lst_msgs.push_back(m);
					st.nmsgs_listener_state++;
				}
				taken = true;
			}
		}
		if (!taken) st.nmsgs_skipped++; // registration, queries and the other notify subtypes belong to the control plane
		p += total_sz;
	}
	st.bytes_consumed = (uint64_t)(p - (const uint8_t *)buf);
	st.nrecords = (uint64_t)nconn + nlst;
	if (nconn || nlst) {
		rc = ensure_staging(c, nbytes + 8, 0);
		if (rc) return rc;
		HIPCHK(hipMemcpyAsync(c->dev_staging, buf, nbytes, hipMemcpyHostToDevice, c->stream));
		if (nconn) {
			rc = wire_decode(c, c->dev_staging, nbytes, conn_msgs, nconn);
			if (!rc) rc = run_conn(c, c->dev_staging, c->dev_offsets, nconn);
			if (rc) return rc;
		}
		if (nlst) {
			rc = wire_decode(c, c->dev_staging, nbytes, lst_msgs, nlst);
			if (!rc) rc = run_lstate(c, c->dev_staging, c->dev_offsets, nullptr, host, nlst);
			if (rc) return rc;
		}
		HIPCHK(hipStreamSynchronize(c->stream)); // the caller's buffer is only valid during the call
	}
	if (out) *out = st;
	return GYS_OK;
} GYS_CATCH_ALL

int gys_ingest_host_state(gys_ctx *c, const uint8_t machine_id[16], const gys_host_state *st)
try {
	GYS_ENTER(c);
	if (!c || !machine_id || !st) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(c->host_state + host, st, sizeof(*st), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(c->host_state_epoch + host, &c->epoch, 4, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return GYS_OK;
} GYS_CATCH_ALL

// ------------------------------------------------------------------------------------------------ window boundary
int gys_reduce_sections(gys_ctx *c, gys_reduce_section out[4], uint32_t *nsections)
try {
	GYS_ENTER(c);
	if (!c || !out || !nsections) return GYS_ERR_INVAL;
	out[0] = {c->arena + c->al.off_hll8, (uint64_t)1 << GYS_HLL_P, 0, 0};
	out[1] = {c->arena + c->al.off_u32, c->al.n_u32, 1, 1};
	out[2] = {c->arena + c->al.off_i64sum, c->al.n_i64sum, 2, 1};
	out[3] = {c->arena + c->al.off_i64max, c->al.n_i64max, 2, 0};
	*nsections = 4;
	return GYS_OK;
} GYS_CATCH_ALL

// the kernels of the window boundary before the exchange (no level roll): folds of the per-service window accumulators into the
// Count-Min rows, cluster STATE_ONE sums + HLL pack, eager-mode record sweep.  dev_epoch: read the window number from device memory
// (captured graph) instead of the launch parameter.
static int enqueue_prepare(gys_ctx *c, bool dev_epoch)
{
	PrepP p{};
	p.host_summ = c->host_summ_win;
	p.host_state = c->host_state;
	p.host_state_epoch = c->host_state_epoch;
	p.host_cluster = c->host_cluster;
	p.nhosts = (uint32_t)c->hosts.size();
	p.epoch = c->epoch;
	p.d_epoch = dev_epoch ? c->d_epoch : nullptr;
	p.cluster_state = (uint32_t *)(c->arena + c->al.off_u32) + c->al.u32_cluster;
	p.hll32 = c->hll32;
	p.hll8 = c->arena + c->al.off_hll8;
	const uint32_t nthreads = std::max<uint32_t>(p.nhosts, (1u << GYS_HLL_P) / 4u);
	int rcf = conn_fold(c);
	if (!rcf) rcf = resp_cms_fold(c);
	if (rcf) return rcf;
	hipLaunchKernelGGL(k_window_prepare, dim3((nthreads + 255) / 256), dim3(256), 0, c->stream, p);
	if (c->nsvc && !c->cfg.enable_tdigest) {
		// eager mode (records updated per event): all-time += window (GY_HISTOGRAM::add_histogram), window cleared, and the all-service
		// histogram of the window reduced into the arena in the same pass over the records.  With the t-digest on, the records are
		// folded lazily from the value buffers ("per-key value buffers" in gys_kernels.hpp) and there is nothing to sweep here.
		long long *gh = (long long *)(c->arena + c->al.off_i64sum) + c->al.i64_ghist;
		hipLaunchKernelGGL(k_hist_fold, dim3(grid_for((uint64_t)c->nsvc * 16, 256, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, c->hist_all,
				   c->hist_win, (uint64_t)c->nsvc, 1, gh, (long long *)(c->arena + c->al.off_i64max));
	}
	HIPCHK(hipGetLastError());
	return GYS_OK;
}

int gys_window_prepare(gys_ctx *c, uint64_t tusec)
try {
	GYS_ENTER(c);
	if (!c) return GYS_ERR_INVAL;
	if (c->prepared) {
		set_err("window already prepared");
		return GYS_ERR_STATE;
	}
	if (c->cfg.enable_levels) {
		const int rcl = level_roll(c, tusec); // before the eager fold below: it needs the cumulative records WITHOUT the closing window
		if (rcl) return rcl;
	}
	{
		ProfScope ps(c, "window_prepare");
		const int rc = enqueue_prepare(c, false);
		if (rc) return rc;
	}
	c->prepared = true;
	return GYS_OK;
} GYS_CATCH_ALL

// the fixed sequence that ends a window: keep the (reduced) registers for queries, start the next window from zero
static hipError_t enqueue_finish(gys_ctx *c, hipStream_t st)
{
	hipError_t e;
	// ACTIVE_CONN_STATS tables: the window's tables enter a ring of the last three windows, the queries read the per-cell maximum (k_act_latch)
	hipLaunchKernelGGL(k_act_latch, dim3(64), dim3(256), 0, st, (const uint32_t *)(c->arena + c->al.off_u32) + c->al.u32_misc, c->d_epoch, c->act_live,
			   (const uint32_t *)(c->arena + c->al.off_u32) + c->al.u32_pair,
			   (const unsigned long long *)(c->arena + c->al.off_i64sum) + c->al.i64_pair, c->ring_act32, c->ring_act64, c->last_act32, c->last_act64);
	if (c->nsvc) {
		// CONN_BITMAP cleared every window (secs_to_reset_ = 5); lazily (per key, on its next touch) when the per-key pass runs
		if (!c->cfg.enable_tdigest && (e = hipMemsetAsync(c->bitmap, 0, (uint64_t)c->nsvc * GYS_BM_WORDS * 4, st)) != hipSuccess) return e;
		if (c->svc_hll && (e = hipMemsetAsync(c->svc_hll, 0, (uint64_t)c->nsvc << c->cfg.svc_hll_p, st)) != hipSuccess) return e;
	}
	const uint64_t hb = (uint64_t)c->hosts.size() * 16 * 4;
	if (((uintptr_t)c->arena & 15u) == 0) { // (a caller's reduce_arena is torch / hipMalloc memory: always; the plain sequence below otherwise)
		WinFinishP p{};
		p.arena = (uint4 *)c->arena;
		p.last = (uint4 *)c->last;
		p.n16 = c->al.total / 16;
		p.i64max_at = c->al.off_i64max;
		p.hll32 = (uint4 *)c->hll32;
		p.hll16 = ((uint64_t)4 << GYS_HLL_P) / 16;
		p.hs_win = (uint4 *)c->host_summ_win;
		p.hs_last = (uint4 *)c->host_summ_last;
		p.hs16 = hb / 16;
		p.d_epoch = c->d_epoch; // the device copy of the window number follows the host's
		hipLaunchKernelGGL(k_window_finish, dim3(grid_for(std::max<uint64_t>(p.n16, p.hs16), 256, (uint32_t)c->ncu * 4)), dim3(256), 0, st, p);
		return hipGetLastError();
	}
	if ((e = hipMemcpyAsync(c->last, c->arena, c->al.total, hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
	if ((e = hipMemsetAsync(c->arena, 0, c->al.total, st)) != hipSuccess) return e;
	if ((e = hipMemcpyAsync(c->arena + c->al.off_i64max, &c->i64min, 8, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
	if ((e = hipMemsetAsync(c->hll32, 0, (uint64_t)4 << GYS_HLL_P, st)) != hipSuccess) return e;
	if (hb) {
		if ((e = hipMemcpyAsync(c->host_summ_last, c->host_summ_win, hb, hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
		if ((e = hipMemsetAsync(c->host_summ_win, 0, hb, st)) != hipSuccess) return e;
	}
	hipLaunchKernelGGL(k_epoch_inc, dim3(1), dim3(1), 0, st, c->d_epoch);
	return hipGetLastError();
}

int gys_window_finish(gys_ctx *c)
try {
	GYS_ENTER(c);
	if (!c) return GYS_ERR_INVAL;
	if (!c->prepared) {
		set_err("gys_window_finish without gys_window_prepare");
		return GYS_ERR_STATE;
	}
	ProfScope ps(c, "window_finish");
	// keep the (reduced) registers of this window for queries, start the next window from zero.  The sequence is the same every
	// window for a given registry shape: it is captured into a hipGraph the first time and replayed afterwards (one launch instead
	// of seven); any capture problem falls back to plain stream operations.
	const uint64_t shape = ((uint64_t)c->hosts.size() << 32) | (uint64_t)c->nsvc;
	auto enqueue = [&](hipStream_t st) -> hipError_t { return enqueue_finish(c, st); };
	if (c->win_graph_state >= 0 && (c->win_graph_state == 0 || c->win_graph_shape != shape)) {
		if (c->win_graph_exec) hipGraphExecDestroy(c->win_graph_exec);
		if (c->win_graph) hipGraphDestroy(c->win_graph);
		c->win_graph_exec = nullptr;
		c->win_graph = nullptr;
		c->win_graph_state = -1;
		if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
			const hipError_t e1 = enqueue(c->stream);
			const hipError_t e2 = hipStreamEndCapture(c->stream, &c->win_graph);
			if (e1 == hipSuccess && e2 == hipSuccess && c->win_graph &&
			    hipGraphInstantiate(&c->win_graph_exec, c->win_graph, nullptr, nullptr, 0) == hipSuccess) {
				c->win_graph_state = 1;
				c->win_graph_shape = shape;
			}
		}
		(void)hipGetLastError();
	}
	if (c->win_graph_state == 1) {
		HIPCHK(hipGraphLaunch(c->win_graph_exec, c->stream));
		c->win_graph_launches++;
	} else {
		HIPCHK(enqueue(c->stream));
	}
	HIPCHK(hipGetLastError());
	c->epoch++;
	c->prepared = false;
	c->have_last = true;
	return GYS_OK;
} GYS_CATCH_ALL

// The whole single-rank window boundary as ONE captured hipGraph (BASELINE config 5: "hipGraph-captured window"): the fold kernels that
// are due (connection accumulators, Count-Min rows of the response path), k_window_prepare, the eager-mode sweep, the copy / clear
// sequence and the window-number increment -- captured once per (registry shape, which folds are due) and replayed with one launch.
// Not capturable, and therefore run as gys_window_prepare + gys_window_finish: multi-level windows (the snapshot masks depend on the
// close time) and more than one rank (the exchange sits between the two halves; gys_window_close_rccl).
int gys_window_close(gys_ctx *c, uint64_t tusec)
try {
	GYS_ENTER(c);
	if (!c) return GYS_ERR_INVAL;
	if (c->prepared) {
		set_err("window already prepared");
		return GYS_ERR_STATE;
	}
	static const bool no_graph = getenv("GYS_NO_CLOSE_GRAPH") != nullptr;
	if (c->cfg.enable_levels || c->cfg.nranks > 1 || no_graph) {
		const int rc = gys_window_prepare(c, tusec);
		return rc ? rc : gys_window_finish(c);
	}
	const uint32_t key = (c->conn_dirty && c->nsvc ? 1u : 0u) | (c->resp_dirty && c->nsvc && c->resp_win ? 2u : 0u);
	const uint64_t shape = ((uint64_t)c->hosts.size() << 32) | (uint64_t)c->nsvc;
	gys_ctx::CloseGraph &cg = c->close_graph[key];
	if (cg.state >= 0 && (cg.state == 0 || cg.shape != shape)) {
		if (cg.x) hipGraphExecDestroy(cg.x);
		if (cg.g) hipGraphDestroy(cg.g);
		cg.x = nullptr;
		cg.g = nullptr;
		cg.state = -1;
		const bool cd = c->conn_dirty, rd = c->resp_dirty;
		if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
			const int r1 = enqueue_prepare(c, true);
			const hipError_t e1 = enqueue_finish(c, c->stream);
			const hipError_t e2 = hipStreamEndCapture(c->stream, &cg.g);
			if (r1 == GYS_OK && e1 == hipSuccess && e2 == hipSuccess && cg.g && hipGraphInstantiate(&cg.x, cg.g, nullptr, nullptr, 0) == hipSuccess) {
				cg.state = 1;
				cg.shape = shape;
			}
		}
		(void)hipGetLastError();
		c->conn_dirty = cd; // (the folds were only recorded, not run)
		c->resp_dirty = rd;
	}
	if (cg.state == 1) {
		ProfScope ps(c, "window_close_graph");
		HIPCHK(hipGraphLaunch(cg.x, c->stream));
		c->conn_dirty = false;
		c->resp_dirty = false;
		c->close_graph_launches++;
		c->win_graph_launches++;
	} else {
		const int rc = gys_window_prepare(c, tusec);
		return rc ? rc : gys_window_finish(c);
	}
	c->epoch++;
	c->prepared = false;
	c->have_last = true;
	return GYS_OK;
} GYS_CATCH_ALL

// ------------------------------------------------------------------------------------------------ queries
int gys_query_svcsumm(gys_ctx *c, const uint8_t machine_id[16], gys_svcsumm *out)
try {
	GYS_ENTER(c);
	if (!c || !machine_id || !out) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	int32_t s[16];
	HIPCHK(hipMemcpyAsync(s, c->host_summ_last + (size_t)host * 16, sizeof(s), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (int i = 0; i < GYS_NSTATES; ++i) out->nstates[i] = s[i];
	out->tot_qps = s[6];
	out->tot_act_conn = s[7];
	out->tot_kb_inbound = s[8];
	out->tot_kb_outbound = s[9];
	out->tot_ser_errors = s[10];
	out->nlisteners = s[11];
	out->nactive = s[12];
	return GYS_OK;
} GYS_CATCH_ALL

int gys_query_clusterstate(gys_ctx *c, const char *cluster_name, gys_cluster_state *out)
try {
	GYS_ENTER(c);
	if (!c || !cluster_name || !out) return GYS_ERR_INVAL;
	auto it = c->cluster_map.find(cluster_name);
	if (it == c->cluster_map.end()) {
		set_err("unknown cluster");
		return GYS_ERR_NOTFOUND;
	}
	uint32_t v[12];
	const uint32_t *src = (const uint32_t *)(c->last + c->al.off_u32) + c->al.u32_cluster + (size_t)it->second * 12;
	HIPCHK(hipMemcpyAsync(v, src, sizeof(v), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	memcpy(out, v, sizeof(*out));
	return GYS_OK;
} GYS_CATCH_ALL

int gys_lookup_service(gys_ctx *c, uint64_t glob_id, uint32_t *slot)
try {
	GYS_ENTER(c);
	if (!c || !slot) return GYS_ERR_INVAL;
	auto it = c->gid_map_h.find(glob_id);
	if (it == c->gid_map_h.end()) {
		set_err("unknown glob_id %016llx", (unsigned long long)glob_id);
		return GYS_ERR_NOTFOUND;
	}
	*slot = it->second;
	return GYS_OK;
} GYS_CATCH_ALL

int gys_query_hist_percentiles(gys_ctx *c, uint64_t glob_id, int which, gys_hist_data *pdata, uint32_t npct, uint64_t *total_count, int64_t *max_val,
			       float *pavg)
try {
	GYS_ENTER(c);
	if (!c || !pdata || which < 0 || which > 1) return GYS_ERR_INVAL;
	uint32_t slot;
	int rc = gys_lookup_service(c, glob_id, &slot);
	if (rc) return rc;
	gys_hist_rec h;
	rc = gys_export_hist(c, which, slot, 1, &h);
	if (rc) return rc;
	const HashDef &d = hash_def(GYS_RESP_TIME_HASH);
	if (total_count) *total_count = h.total_count;
	if (max_val) *max_val = h.max_val_seen;
	if (pavg) { // common/gy_statistics.h:734-750
		int64_t total_sum = 0;
		const int64_t cnt = h.total_count ? (int64_t)h.total_count : 1;
		for (int i = 0; i < d.nthr + 2; ++i) total_sum += h.stats[i].sum;
		*pavg = (total_sum * 1.0f) / cnt;
	}
	for (uint32_t i = 0; i < npct; ++i) hist_percentile(d, h, pdata[i].percentile, &pdata[i].data_value, &pdata[i].sum, &pdata[i].count);
	return GYS_OK;
} GYS_CATCH_ALL

// quantile of an exact-integer digest: interpolation between cluster centres; only + - * / on doubles (matches oracle bit-for-bit)
static double td_quantile_interp(const int64_t *sum, const uint32_t *cnt, int32_t vmin, int32_t vmax, double q)
{
	uint64_t N = 0;
	for (int k = 0; k < GYS_TD_NB; ++k) N += cnt[k];
	if (!N) return 0.0;
	if (q < 0.0) q = 0.0;
	if (q > 1.0) q = 1.0;
	const double t = q * (double)N;
	double wbefore = 0.0, prev_c = 0.0, prev_mean = 0.0;
	bool have_prev = false;
	for (int k = 0; k < GYS_TD_NB; ++k) {
		if (!cnt[k]) continue;
		const double mean = (double)sum[k] / (double)cnt[k];
		const double c = wbefore + (double)cnt[k] * 0.5;
		if (t < c) {
			if (!have_prev) {
				const double lo = (double)vmin;
				if (c <= 0.0) return mean;
				return lo + (mean - lo) * (t / c);
			}
			return prev_mean + (mean - prev_mean) * ((t - prev_c) / (c - prev_c));
		}
		wbefore += (double)cnt[k];
		prev_c = c;
		prev_mean = mean;
		have_prev = true;
	}
	const double hi = (double)vmax, span = (double)N - prev_c;
	if (span <= 0.0) return hi;
	return prev_mean + (hi - prev_mean) * ((t - prev_c) / span);
}

// integer-millisecond data: the interpolated value rounded half-up to the value domain (same definition as the oracle)
static double td_quantile_host(const int64_t *sum, const uint32_t *cnt, int32_t vmin, int32_t vmax, double q)
{
	return std::floor(td_quantile_interp(sum, cnt, vmin, vmax, q) + 0.5);
}

// merged view of one service's digest = its clusters re-clustered with its buffered values (k_digest_merge in query mode: the state
// is not modified)
static int td_merged_view(gys_ctx *c, uint32_t slot, int64_t *sum, uint32_t *cnt, int32_t *vmin, int32_t *vmax)
{
	int rc = fold_range(c, slot, 1); // min / max must cover the buffered values
	if (rc) return rc;
	TdMeta mt;
	HIPCHK(hipMemcpyAsync(&mt, c->td_meta + slot, sizeof(mt), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	const MergeEnt ent{slot, mt.npend, 0u, 0u};
	HIPCHK(hipMemcpyAsync(c->query_list, &ent, sizeof(ent), hipMemcpyHostToDevice, c->stream));
	MergeP mp{};
	mp.d = digest_params(c);
	mp.list = c->query_list;
	mp.count = c->merge_count + 8;
	mp.out_sum = c->query_sum;
	mp.out_cnt = c->query_cnt;
	if (mt.npend <= GYS_MERGE_CLASS0) hipLaunchKernelGGL((k_digest_merge<GYS_MERGE_CLASS0, 256u>), dim3(1), dim3(256), 0, c->stream, mp);
	else if (mt.npend <= GYS_MERGE_CLASS1) hipLaunchKernelGGL((k_digest_merge<GYS_MERGE_CLASS1, 256u>), dim3(1), dim3(256), 0, c->stream, mp);
	else hipLaunchKernelGGL((k_digest_merge<GYS_MERGE_LDS_MAX, 1024u>), dim3(1), dim3(1024), 0, c->stream, mp);
	HIPCHK(hipGetLastError());
	int2 mm;
	HIPCHK(hipMemcpyAsync(sum, c->query_sum, sizeof(int64_t) * GYS_TD_NB, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(cnt, c->query_cnt, sizeof(uint32_t) * GYS_TD_NB, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(&mm, c->td_minmax + slot, sizeof(mm), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	*vmin = mm.x;
	*vmax = mm.y;
	return GYS_OK;
}

#define TDIGEST_CHECK()                         \
	if (!c->cfg.enable_tdigest) {           \
		set_err("t-digest disabled");   \
		return GYS_ERR_STATE;           \
	}

int gys_query_quantiles(gys_ctx *c, uint64_t glob_id, const double *q, uint32_t nq, double *out)
try {
	GYS_ENTER(c);
	if (!c || !q || !out) return GYS_ERR_INVAL;
	TDIGEST_CHECK();
	uint32_t slot;
	int rc = gys_lookup_service(c, glob_id, &slot);
	if (rc) return rc;
	int64_t sum[GYS_TD_NB];
	uint32_t cnt[GYS_TD_NB];
	int32_t vmin, vmax;
	rc = td_merged_view(c, slot, sum, cnt, &vmin, &vmax);
	if (rc) return rc;
	for (uint32_t i = 0; i < nq; ++i) out[i] = td_quantile_host(sum, cnt, vmin, vmax, q[i]);
	return GYS_OK;
} GYS_CATCH_ALL

// ------------------------------------------------------------------------------------------------ Postgres tdigest forms
// The reference aggregates percentiles in SQL with the tdigest extension (tvondra/tdigest, not in /root/reference and unpinned:
// "create extension if not exists tdigest" common/gy_query_common.cc:3387; public.tdigest(col, 100) / public.tdigest_percentile
// :1818-1855, common/gy_json_field_maps.h:2641).  These two calls hand a service's digest over in that type's own external forms so
// that '<text>'::public.tdigest (or the binary COPY / libpq form) can be fed to public.tdigest_percentile and unioned with row
// digests.  Restated from the extension's published I/O functions (tdigest_out / tdigest_send, format with the TDIGEST_STORES_MEAN
// flag, 1.2.0 and later); the extension is absent here => PARITY UNPINNED.
//   text:   flags 1 count <N> compression 100 centroids <K> (<mean %lf>, <count>) ...      (means ascending)
//   binary: int32 flags | int64 count | int32 compression | int32 ncentroids | K x { float8 mean, int64 count }, network byte order
static int td_sql_centroids(gys_ctx *c, uint64_t glob_id, double *mean, int64_t *count, int *k, int64_t *total)
{
	uint32_t slot;
	int rc = gys_lookup_service(c, glob_id, &slot);
	if (rc) return rc;
	int64_t sum[GYS_TD_NB];
	uint32_t cnt[GYS_TD_NB];
	int32_t vmin, vmax;
	rc = td_merged_view(c, slot, sum, cnt, &vmin, &vmax);
	if (rc) return rc;
	*k = 0;
	*total = 0;
	for (int i = 0; i < GYS_TD_NB; ++i)
		if (cnt[i]) {
			mean[*k] = (double)sum[i] / (double)cnt[i];
			count[*k] = (int64_t)cnt[i];
			*total += (int64_t)cnt[i];
			++*k;
		}
	if (*k == 0) { // tdigest_in rejects count <= 0: an empty digest has no literal (SQL NULL is the empty aggregate)
		set_err("service %016llx has no response values yet", (unsigned long long)glob_id);
		return GYS_ERR_NOTFOUND;
	}
	return GYS_OK;
}

int gys_tdigest_sql_text(gys_ctx *c, uint64_t glob_id, char *buf, size_t buflen, size_t *needed)
try {
	GYS_ENTER(c);
	if (!c || (!buf && buflen)) return GYS_ERR_INVAL;
	TDIGEST_CHECK();
	double mean[GYS_TD_NB];
	int64_t count[GYS_TD_NB], total;
	int k;
	const int rc = td_sql_centroids(c, glob_id, mean, count, &k, &total);
	if (rc) return rc;
	std::string out;
	char tmp[96];
	snprintf(tmp, sizeof(tmp), "flags 1 count %lld compression %d centroids %d", (long long)total, GYS_TD_NB, k);
	out = tmp;
	for (int i = 0; i < k; ++i) {
		snprintf(tmp, sizeof(tmp), " (%lf, %lld)", mean[i], (long long)count[i]);
		out += tmp;
	}
	if (needed) *needed = out.size();
	if (out.size() + 1 > buflen) return GYS_ERR_NOMEM;
	memcpy(buf, out.c_str(), out.size() + 1);
	return GYS_OK;
} GYS_CATCH_ALL

int gys_tdigest_sql_binary(gys_ctx *c, uint64_t glob_id, void *buf, size_t buflen, size_t *needed)
try {
	GYS_ENTER(c);
	if (!c || (!buf && buflen)) return GYS_ERR_INVAL;
	TDIGEST_CHECK();
	double mean[GYS_TD_NB];
	int64_t count[GYS_TD_NB], total;
	int k;
	const int rc = td_sql_centroids(c, glob_id, mean, count, &k, &total);
	if (rc) return rc;
	const size_t len = 4 + 8 + 4 + 4 + (size_t)k * 16;
	if (needed) *needed = len;
	if (len > buflen) return GYS_ERR_NOMEM;
	uint8_t *p = (uint8_t *)buf;
	auto be = [&p](uint64_t v, int nbytes) { // pq_sendint32 / pq_sendint64 / pq_sendfloat8: big endian
		for (int i = nbytes - 1; i >= 0; --i) *p++ = (uint8_t)(v >> (8 * i));
	};
	be(1u, 4);
	be((uint64_t)total, 8);
	be((uint64_t)GYS_TD_NB, 4);
	be((uint64_t)k, 4);
	for (int i = 0; i < k; ++i) {
		uint64_t bits;
		memcpy(&bits, &mean[i], 8);
		be(bits, 8);
		be((uint64_t)count[i], 8);
	}
	return GYS_OK;
} GYS_CATCH_ALL

static double hll_estimate_host(const uint8_t *regs, int p)
{
	const uint32_t m = 1u << p;
	double sum = 0.0;
	uint32_t zeros = 0;
	for (uint32_t i = 0; i < m; ++i) {
		sum += 1.0 / (double)(1ull << regs[i]);
		zeros += regs[i] == 0;
	}
	const double alpha = m == 16 ? 0.673 : (m == 32 ? 0.697 : (m == 64 ? 0.709 : 0.7213 / (1.0 + 1.079 / (double)m)));
	double e = alpha * (double)m * (double)m / sum;
	if (e <= 2.5 * (double)m && zeros) e = (double)m * std::log((double)m / (double)zeros);
	return e;
}

int gys_query_distinct_flows(gys_ctx *c, double *out)
try {
	GYS_ENTER(c);
	if (!c || !out) return GYS_ERR_INVAL;
	std::vector<uint8_t> regs((size_t)1 << GYS_HLL_P);
	HIPCHK(hipMemcpyAsync(regs.data(), c->last + c->al.off_hll8, regs.size(), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	*out = hll_estimate_host(regs.data(), GYS_HLL_P);
	return GYS_OK;
} GYS_CATCH_ALL

int gys_query_cms(gys_ctx *c, uint64_t glob_id, int which, uint64_t *out)
try {
	GYS_ENTER(c);
	if (!c || !out || which < 0 || which > 1) return GYS_ERR_INVAL;
	uint64_t best = ~0ull;
	for (uint32_t r = 0; r < GYS_CMS_D; ++r) {
		const uint32_t col = jhash2_u64(glob_id, GYS_SEED + r) & (GYS_CMS_W - 1);
		uint64_t v = 0;
		if (which == 0) {
			uint32_t v32;
			HIPCHK(hipMemcpyAsync(&v32, (const uint32_t *)(c->last + c->al.off_u32) + c->al.u32_cms + (size_t)r * GYS_CMS_W + col, 4,
					      hipMemcpyDeviceToHost, c->stream));
			HIPCHK(hipStreamSynchronize(c->stream));
			v = v32;
		} else {
			HIPCHK(hipMemcpyAsync(&v, (const uint64_t *)(c->last + c->al.off_i64sum) + c->al.i64_cms + (size_t)r * GYS_CMS_W + col, 8,
					      hipMemcpyDeviceToHost, c->stream));
			HIPCHK(hipStreamSynchronize(c->stream));
		}
		best = std::min(best, v);
	}
	*out = best;
	return GYS_OK;
} GYS_CATCH_ALL

// Count-Min estimate for a (listener, client task group) pair.  which 0 / 1: active connections / bytes of the ACTIVE_CONN_STATS reports of
// the last three windows (per-cell maximum, k_act_latch); which 2 / 3 and 4 / 5: closed connections / bytes of the TCP_CONN_NOTIFY roll-up
// in the last finished window, listener side (connlistenmap_) and client side (connclientmap_) (gys_config.conn_pair_cms); which 6 / 7: as 0 / 1
// for the ACTIVE_CONN_STATS rows whose listener lives on another madhava (is_remote_listen_: the reference's remoteconntbl rows).
static int pair_tables(gys_ctx *c, int which, const void **tbl)
{
	if (which < 0 || which > 7) return GYS_ERR_INVAL;
	if (which == 6) {
		*tbl = c->last_act32 + (uint64_t)GYS_CMS_D * GYS_CMS_W;
		return GYS_OK;
	}
	if (which == 7) {
		*tbl = c->last_act64 + (uint64_t)GYS_CMS_D * GYS_CMS_W;
		return GYS_OK;
	}
	if (which >= 2 && !c->cfg.conn_pair_cms) {
		set_err("gys_config.conn_pair_cms is off");
		return GYS_ERR_STATE;
	}
	switch (which) {
	case 0: *tbl = c->last_act32; break;
	case 1: *tbl = c->last_act64; break;
	case 2: *tbl = (const uint32_t *)(c->last + c->al.off_u32) + c->al.u32_cpair; break;
	case 3: *tbl = (const uint64_t *)(c->last + c->al.off_i64sum) + c->al.i64_cpair; break;
	case 4: *tbl = (const uint32_t *)(c->last + c->al.off_u32) + c->al.u32_cpair + (uint64_t)GYS_CMS_D * GYS_CMS_W; break;
	default: *tbl = (const uint64_t *)(c->last + c->al.off_i64sum) + c->al.i64_cpair + (uint64_t)GYS_CMS_D * GYS_CMS_W; break;
	}
	return GYS_OK;
}

int gys_query_pair_cms(gys_ctx *c, uint64_t listener_glob_id, uint64_t cli_aggr_task_id, int which, uint64_t *out)
try {
	GYS_ENTER(c);
	if (!c || !out) return GYS_ERR_INVAL;
	const void *tbl;
	const int rc = pair_tables(c, which, &tbl);
	if (rc) return rc;
	uint64_t best = ~0ull;
	for (uint32_t r = 0; r < GYS_CMS_D; ++r) {
		const uint32_t col = jhash2_4w((uint32_t)listener_glob_id, (uint32_t)(listener_glob_id >> 32), (uint32_t)cli_aggr_task_id,
					       (uint32_t)(cli_aggr_task_id >> 32), GYS_SEED + r) & (GYS_CMS_W - 1);
		uint64_t v = 0;
		if (!(which & 1)) {
			uint32_t v32;
			HIPCHK(hipMemcpyAsync(&v32, (const uint32_t *)tbl + (size_t)r * GYS_CMS_W + col, 4, hipMemcpyDeviceToHost, c->stream));
			HIPCHK(hipStreamSynchronize(c->stream));
			v = v32;
		} else {
			HIPCHK(hipMemcpyAsync(&v, (const uint64_t *)tbl + (size_t)r * GYS_CMS_W + col, 8, hipMemcpyDeviceToHost, c->stream));
			HIPCHK(hipStreamSynchronize(c->stream));
		}
		best = std::min(best, v);
	}
	*out = best;
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_pair_cms(gys_ctx *c, int which, void *out)
try {
	GYS_ENTER(c);
	if (!c || !out) return GYS_ERR_INVAL;
	const void *tbl;
	const int rc = pair_tables(c, which, &tbl);
	if (rc) return rc;
	const size_t n = (size_t)GYS_CMS_D * GYS_CMS_W;
	HIPCHK(hipMemcpyAsync(out, tbl, n * ((which & 1) ? 8 : 4), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_active_conn_counters(gys_ctx *c, uint32_t first_slot, uint32_t nslots, uint64_t *out)
try {
	GYS_ENTER(c);
	if (!c || !out || (uint64_t)first_slot + nslots > c->nsvc) return GYS_ERR_INVAL;
	if (!nslots) return GYS_OK;
	HIPCHK(hipMemcpyAsync(out, c->svc_act + (size_t)first_slot * 4, (size_t)nslots * 32, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return GYS_OK;
} GYS_CATCH_ALL

int gys_query_topn(gys_ctx *c, const uint8_t machine_id[16], int kind, gys_topn_entry out[GYS_TOPN], uint32_t *nout)
try {
	GYS_ENTER(c);
	if (!c || !machine_id || !out || !nout || kind < 0 || kind > 3) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	*nout = 0;
	if (!c->nsvc || c->epoch < 2) return GYS_OK;
	const uint32_t cap = (uint32_t)std::min<uint64_t>(c->cfg.max_services, 65536);
	HIPCHK(hipMemsetAsync(c->misc + 1, 0, 4, c->stream));
	hipLaunchKernelGGL(k_topn_filter, dim3((c->nsvc + 255) / 256), dim3(256), 0, c->stream, c->svc_state, c->nsvc, host, c->epoch - 1, kind, c->topn_slot,
			   c->topn_metric, c->misc + 1, cap);
	uint32_t cnt = 0;
	HIPCHK(hipMemcpyAsync(&cnt, c->misc + 1, 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	cnt = std::min(cnt, cap);
	if (!cnt) return GYS_OK;
	std::vector<uint32_t> slots(cnt);
	std::vector<uint64_t> metrics(cnt);
	HIPCHK(hipMemcpy(slots.data(), c->topn_slot, (size_t)cnt * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(metrics.data(), c->topn_metric, (size_t)cnt * 8, hipMemcpyDeviceToHost));
	std::vector<uint32_t> order(cnt);
	for (uint32_t i = 0; i < cnt; ++i) order[i] = i;
	// BOUNDED_PRIO_QUEUE keeps the N largest (common/gy_statistics.h:385-414); ties resolved by lowest service slot for determinism
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return metrics[a] != metrics[b] ? metrics[a] > metrics[b] : slots[a] < slots[b]; });
	const uint32_t k = std::min<uint32_t>(cnt, GYS_TOPN);
	for (uint32_t i = 0; i < k; ++i) {
		const uint32_t s = slots[order[i]];
		uint8_t rec[96];
		HIPCHK(hipMemcpy(rec, c->svc_state + (size_t)s * 96, 96, hipMemcpyDeviceToHost));
		memcpy(out[i].state, rec, 88);
		memcpy(&out[i].glob_id, rec, 8);
		out[i].host_slot = host;
		out[i].metric = (uint32_t)metrics[order[i]];
	}
	*nout = k;
	return GYS_OK;
} GYS_CATCH_ALL

int gys_scan_percentiles_dev(gys_ctx *c, int which, const float *pcts, uint32_t npct, int64_t *d_out)
try {
	GYS_ENTER(c);
	if (!c || !pcts || !d_out || !npct || npct > 64 || which < 0 || which > 1) return GYS_ERR_INVAL;
	if (!c->nsvc) return GYS_OK;
	HIPCHK(hipMemcpyAsync(c->dev_pcts, pcts, (size_t)npct * 4, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream)); // pcts may be a short-lived host buffer
	{
		const int rcf = fold_range(c, 0, c->nsvc);
		if (rcf) return rcf;
	}
	ProfScope ps(c, "hist_percentiles");
	hipLaunchKernelGGL(k_hist_percentiles_view, dim3((c->nsvc + 255) / 256), dim3(256), 0, c->stream, window_records(c), c->hist_all,
			   c->cfg.enable_tdigest ? c->td_meta : nullptr, c->epoch, which, c->nsvc, c->dev_pcts, npct, d_out);
	HIPCHK(hipGetLastError());
	return GYS_OK;
} GYS_CATCH_ALL

// ------------------------------------------------------------------------------------------------ exports
int gys_scan_quantiles_dev(gys_ctx *c, const double *q, uint32_t nq, double *d_out)
try {
	GYS_ENTER(c);
	if (!c || !q || !d_out || nq == 0 || nq > 16) return GYS_ERR_INVAL;
	TDIGEST_CHECK();
	if (!c->nsvc) return GYS_OK;
	double *d_q = nullptr;
	HIPCHK(hipMalloc((void **)&d_q, sizeof(double) * nq));
	HIPCHK(hipMemcpyAsync(d_q, q, sizeof(double) * nq, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemsetAsync(c->merge_count + FIN_SLOW, 0, 4, c->stream));
	MergeBP bp{};
	bp.d = digest_params(c);
	bp.slow_list = c->merge_list_slow;
	bp.slow_count = c->merge_count + FIN_SLOW;
	bp.qs = d_q;
	bp.nq = nq;
	bp.qout = d_out;
	{
		ProfScope ps(c, "scan_quantiles");
		const dim3 sgrid(std::max(1u, std::min<uint32_t>(c->nsvc, (uint32_t)c->ncu * 8)));
		if (c->pend_cap <= 1024u) hipLaunchKernelGGL((k_digest_bins<true, 4u>), sgrid, dim3(256), 0, c->stream, bp);
		else if (c->pend_cap <= 2048u) hipLaunchKernelGGL((k_digest_bins<true, 8u>), sgrid, dim3(256), 0, c->stream, bp);
		else hipLaunchKernelGGL((k_digest_bins<true, 16u>), sgrid, dim3(256), 0, c->stream, bp);
	}
	uint32_t nslow = 0;
	HIPCHK(hipMemcpyAsync(&nslow, c->merge_count + FIN_SLOW, 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipFree(d_q));
	if (nslow) { // services whose digest weighs 2^31 or more (64-bit weights) or whose buffer holds more than GYS_MB_BIG_CAP values of a second or longer: one at a time through the general merge
		std::vector<MergeEnt> slow(nslow);
		HIPCHK(hipMemcpy(slow.data(), c->merge_list_slow, sizeof(MergeEnt) * nslow, hipMemcpyDeviceToHost));
		std::vector<double> out(nq);
		for (const MergeEnt &e : slow) {
			int64_t sum[GYS_TD_NB];
			uint32_t cnt[GYS_TD_NB];
			int32_t vmin, vmax;
			const int rc = td_merged_view(c, e.slot, sum, cnt, &vmin, &vmax);
			if (rc) return rc;
			for (uint32_t i = 0; i < nq; ++i) out[i] = td_quantile_host(sum, cnt, vmin, vmax, q[i]);
			HIPCHK(hipMemcpy(d_out + (size_t)e.slot * nq, out.data(), sizeof(double) * nq, hipMemcpyHostToDevice));
		}
	}
	return GYS_OK;
} GYS_CATCH_ALL

// host -> member services on the device (slot order inside a host)
static int host_csr(gys_ctx *c)
{
	const uint64_t stamp = ((uint64_t)c->hosts.size() << 32) | c->nsvc;
	if (c->csr_stamp == stamp && c->csr_off) return GYS_OK;
	const uint32_t nh = (uint32_t)c->hosts.size();
	std::vector<uint32_t> off(nh + 1, 0), members;
	members.reserve(c->nsvc);
	for (uint32_t h = 0; h < nh; ++h) {
		std::vector<uint32_t> sl = c->host_lst[h].all_slots;
		std::sort(sl.begin(), sl.end());
		members.insert(members.end(), sl.begin(), sl.end());
		off[h + 1] = (uint32_t)members.size();
	}
	HIPCHK(hipStreamSynchronize(c->stream));
	if (c->csr_off) HIPCHK(hipFree(c->csr_off));
	if (c->csr_mem) HIPCHK(hipFree(c->csr_mem));
	c->csr_off = c->csr_mem = nullptr;
	HIPCHK(hipMalloc((void **)&c->csr_off, off.size() * 4));
	HIPCHK(hipMalloc((void **)&c->csr_mem, std::max<size_t>(members.size(), 1) * 4));
	HIPCHK(hipMemcpy(c->csr_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
	if (!members.empty()) HIPCHK(hipMemcpy(c->csr_mem, members.data(), members.size() * 4, hipMemcpyHostToDevice));
	c->csr_stamp = stamp;
	return GYS_OK;
}

// the 10 best services of EVERY host for one kind (last closed window): slots[h * GYS_TOPN + r] (GYS_NOSLOT = none), metrics likewise
static int topn_all_hosts(gys_ctx *c, int kind, std::vector<uint32_t> &slots, std::vector<uint64_t> &metrics)
{
	const uint32_t nh = (uint32_t)c->hosts.size();
	slots.assign((size_t)nh * GYS_TOPN, GYS_NOSLOT);
	metrics.assign((size_t)nh * GYS_TOPN, 0);
	if (!nh || !c->nsvc || c->epoch < 2) return GYS_OK;
	int rc = host_csr(c);
	if (rc) return rc;
	uint32_t *d_slot = nullptr;
	uint64_t *d_metric = nullptr;
	HIPCHK(hipMalloc((void **)&d_slot, slots.size() * 4));
	HIPCHK(hipMalloc((void **)&d_metric, metrics.size() * 8));
	TopnHostsP tp{};
	tp.svc_state = c->svc_state;
	tp.off = c->csr_off;
	tp.members = c->csr_mem;
	tp.nhosts = nh;
	tp.epoch = c->epoch - 1;
	tp.kind = kind;
	tp.out_slot = d_slot;
	tp.out_metric = d_metric;
	{
		ProfScope ps(c, "topn_hosts");
		hipLaunchKernelGGL(k_topn_hosts, dim3(std::min<uint32_t>(nh, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, tp);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(slots.data(), d_slot, slots.size() * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(metrics.data(), d_metric, metrics.size() * 8, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipFree(d_slot));
	HIPCHK(hipFree(d_metric));
	return GYS_OK;
}

// ------------------------------------------------------------------------------------------------ roll-up digests
#define GYS_RB_CHUNK_SERVICES 1024u // members a workgroup of k_rollup_accum adds up before it hands its bins to the group's
#define GYS_RB_CHUNK_SLABS 32u

static void rollup_chunks(const std::vector<uint32_t> &off, uint32_t per, std::vector<RollupChunk> &chunks)
{
	for (uint32_t g = 0; g + 1 < (uint32_t)off.size(); ++g)
		for (uint32_t m = off[g]; m < off[g + 1]; m += per) chunks.push_back(RollupChunk{g, m, std::min(off[g + 1], m + per), 0u});
}

// bins of `ngroups` groups <- the members the chunks name; slabs out.  d_chunks / d_members: DEVICE arrays.
static int rollup_run(gys_ctx *c, int kind, const RollupChunk *d_chunks, uint32_t nchunks, const uint32_t *d_members, uint32_t ngroups,
		      const gys_tdigest_slab *d_in, gys_tdigest_slab *d_out)
{
	if (!ngroups) return GYS_OK;
	if (c->rb_bins_groups < ngroups) {
		if (c->rb_bins) {
			HIPCHK(hipStreamSynchronize(c->stream));
			HIPCHK(hipFree(c->rb_bins));
			c->rb_bins = nullptr;
			c->rb_bins_groups = 0;
		}
		HIPCHK(hipMalloc((void **)&c->rb_bins, (size_t)ngroups * GYS_RB_STRIDE * 8));
		c->rb_bins_groups = ngroups;
	}
	RollupP rp{};
	rp.d = digest_params(c);
	rp.chunks = d_chunks;
	rp.nchunks = nchunks;
	rp.members = d_members;
	rp.kind = kind;
	rp.in = d_in;
	rp.bins = c->rb_bins;
	rp.out = d_out;
	rp.ngroups = ngroups;
	{
		ProfScope ps(c, kind == 0 ? "rollup_services" : "rollup_slabs");
		const size_t words = (size_t)ngroups * GYS_RB_STRIDE;
		hipLaunchKernelGGL(k_rollup_init, dim3((uint32_t)std::min<size_t>((words + 255) / 256, (size_t)c->ncu * 16)), dim3(256), 0, c->stream, c->rb_bins, ngroups);
		if (nchunks) hipLaunchKernelGGL(k_rollup_accum, dim3(std::min<uint32_t>(nchunks, (uint32_t)c->ncu * 16)), dim3(GYS_RB_NT), 0, c->stream, rp);
		hipLaunchKernelGGL(k_rollup_cluster, dim3(std::min<uint32_t>(ngroups, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, rp);
	}
	HIPCHK(hipGetLastError());
	return GYS_OK;
}

// groups of slabs (kind 1): the lists travel with the call
static int rollup_slabs(gys_ctx *c, const std::vector<uint32_t> &off, const std::vector<uint32_t> &members, const gys_tdigest_slab *d_in, gys_tdigest_slab *d_out)
{
	const uint32_t ngroups = (uint32_t)off.size() - 1;
	if (!ngroups) return GYS_OK;
	std::vector<RollupChunk> chunks;
	rollup_chunks(off, GYS_RB_CHUNK_SLABS, chunks);
	RollupChunk *d_chunks = nullptr;
	uint32_t *d_mem = nullptr;
	HIPCHK(hipMalloc((void **)&d_chunks, std::max<size_t>(chunks.size(), 1) * sizeof(RollupChunk)));
	HIPCHK(hipMalloc((void **)&d_mem, std::max<size_t>(members.size(), 1) * 4));
	if (!chunks.empty()) HIPCHK(hipMemcpyAsync(d_chunks, chunks.data(), chunks.size() * sizeof(RollupChunk), hipMemcpyHostToDevice, c->stream));
	if (!members.empty()) HIPCHK(hipMemcpyAsync(d_mem, members.data(), members.size() * 4, hipMemcpyHostToDevice, c->stream));
	const int rc = rollup_run(c, 1, d_chunks, (uint32_t)chunks.size(), d_mem, ngroups, d_in, d_out);
	HIPCHK(hipStreamSynchronize(c->stream)); // the lists are freed below
	HIPCHK(hipFree(d_chunks));
	HIPCHK(hipFree(d_mem));
	return rc;
}

int gys_tdigest_rollup_dev(gys_ctx *c, int scope, gys_tdigest_slab *d_out)
try {
	GYS_ENTER(c);
	if (!c || !d_out || scope < GYS_ROLLUP_HOST || scope > GYS_ROLLUP_GLOBAL) return GYS_ERR_INVAL;
	TDIGEST_CHECK();
	const uint32_t nh = (uint32_t)c->hosts.size();
	if (!nh) {
		if (scope == GYS_ROLLUP_GLOBAL) HIPCHK(hipMemsetAsync(d_out, 0, sizeof(gys_tdigest_slab), c->stream));
		return GYS_OK;
	}
	// host level: members = the host's services.  The lists stay on the device until a service or a host is registered (10^7 services: 40 MB).
	if (c->rb_host_nsvc != c->nsvc || c->rb_host_nh != nh) {
		std::vector<uint32_t> off(nh + 1, 0), members;
		members.reserve(c->nsvc);
		for (uint32_t h = 0; h < nh; ++h) {
			const std::vector<uint32_t> &sl = c->host_lst[h].all_slots;
			members.insert(members.end(), sl.begin(), sl.end());
			off[h + 1] = (uint32_t)members.size();
		}
		std::vector<RollupChunk> chunks;
		rollup_chunks(off, GYS_RB_CHUNK_SERVICES, chunks);
		HIPCHK(hipStreamSynchronize(c->stream));
		if (c->rb_host_members) HIPCHK(hipFree(c->rb_host_members));
		if (c->rb_host_chunks) HIPCHK(hipFree(c->rb_host_chunks));
		c->rb_host_members = nullptr;
		c->rb_host_chunks = nullptr;
		c->rb_host_nsvc = ~0u;
		HIPCHK(hipMalloc((void **)&c->rb_host_members, std::max<size_t>(members.size(), 1) * 4));
		HIPCHK(hipMalloc((void **)&c->rb_host_chunks, std::max<size_t>(chunks.size(), 1) * sizeof(RollupChunk)));
		if (!members.empty()) HIPCHK(hipMemcpy(c->rb_host_members, members.data(), members.size() * 4, hipMemcpyHostToDevice));
		if (!chunks.empty()) HIPCHK(hipMemcpy(c->rb_host_chunks, chunks.data(), chunks.size() * sizeof(RollupChunk), hipMemcpyHostToDevice));
		c->rb_host_nchunks = (uint32_t)chunks.size();
		c->rb_host_nsvc = c->nsvc;
		c->rb_host_nh = nh;
	}
	gys_tdigest_slab *d_hosts = d_out;
	if (scope != GYS_ROLLUP_HOST) HIPCHK(hipMalloc((void **)&d_hosts, sizeof(gys_tdigest_slab) * nh));
	int rc = rollup_run(c, 0, c->rb_host_chunks, c->rb_host_nchunks, c->rb_host_members, nh, nullptr, d_hosts);
	if (rc == GYS_OK && scope != GYS_ROLLUP_HOST) {
		std::vector<uint32_t> goff, gmem;
		if (scope == GYS_ROLLUP_GLOBAL) { // one group: every host slab
			goff = {0u, nh};
			gmem.resize(nh);
			for (uint32_t h = 0; h < nh; ++h) gmem[h] = h;
		} else {
			const uint32_t ncl = (uint32_t)c->cluster_names.size();
			goff.assign(ncl + 1, 0);
			for (uint32_t cl = 0; cl < ncl; ++cl) {
				for (uint32_t h = 0; h < nh; ++h)
					if (c->host_cluster_h[h] == cl) gmem.push_back(h);
				goff[cl + 1] = (uint32_t)gmem.size();
			}
		}
		rc = rollup_slabs(c, goff, gmem, d_hosts, d_out);
	}
	if (scope != GYS_ROLLUP_HOST) {
		HIPCHK(hipStreamSynchronize(c->stream));
		HIPCHK(hipFree(d_hosts));
	}
	return rc;
} GYS_CATCH_ALL

int gys_tdigest_merge_slabs_dev(gys_ctx *c, const gys_tdigest_slab *d_in, uint32_t n, gys_tdigest_slab *d_out)
try {
	GYS_ENTER(c);
	if (!c || !d_in || !d_out || n == 0) return GYS_ERR_INVAL;
	std::vector<uint32_t> off{0u, n}, mem(n);
	for (uint32_t i = 0; i < n; ++i) mem[i] = i;
	return rollup_slabs(c, off, mem, d_in, d_out);
} GYS_CATCH_ALL

int gys_tdigest_slab_quantiles(gys_ctx *c, const gys_tdigest_slab *d_slab, const double *q, uint32_t nq, double *out)
try {
	GYS_ENTER(c);
	if (!c || !d_slab || !q || !out) return GYS_ERR_INVAL;
	gys_tdigest_slab s;
	HIPCHK(hipMemcpyAsync(&s, d_slab, sizeof(s), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	// td_quantile_interp on the wide counters (same arithmetic: exact integer prefix weights as doubles, one division per centre)
	uint64_t N = 0;
	for (int k = 0; k < GYS_TD_NB; ++k) N += s.cnt[k];
	for (uint32_t i = 0; i < nq; ++i) {
		double r = 0.0;
		if (N) {
			double qq = q[i] < 0.0 ? 0.0 : (q[i] > 1.0 ? 1.0 : q[i]);
			const double t = qq * (double)N;
			double wbefore = 0.0, prev_c = 0.0, prev_mean = 0.0;
			bool have_prev = false, done = false;
			for (int k = 0; k < GYS_TD_NB && !done; ++k) {
				if (!s.cnt[k]) continue;
				const double mean = (double)s.sum[k] / (double)s.cnt[k];
				const double cc = wbefore + (double)s.cnt[k] * 0.5;
				if (t < cc) {
					if (!have_prev) {
						const double lo = (double)s.vmin;
						r = cc <= 0.0 ? mean : lo + (mean - lo) * (t / cc);
					} else {
						r = prev_mean + (mean - prev_mean) * ((t - prev_c) / (cc - prev_c));
					}
					done = true;
					break;
				}
				wbefore += (double)s.cnt[k];
				prev_c = cc;
				prev_mean = mean;
				have_prev = true;
			}
			if (!done) {
				const double hi = (double)s.vmax, span = (double)N - prev_c;
				r = span <= 0.0 ? hi : prev_mean + (hi - prev_mean) * ((t - prev_c) / span);
			}
			r = std::floor(r + 0.5);
		}
		out[i] = r;
	}
	return GYS_OK;
} GYS_CATCH_ALL

uint32_t gys_num_clusters(gys_ctx *c) { return c ? (uint32_t)c->cluster_names.size() : 0; }

// ------------------------------------------------------------------------------------------------ RCCL exchange in the library
// RCCL is bound at run time (dlopen), not at link time: a process that never joins a communicator (one madhava, one GPU) does not need
// the library at all, and a host process that already carries an RCCL of its own (PyTorch bundles one) can point the engine at THAT
// copy instead of running two different RCCL builds side by side.  GYS_RCCL_LIB = path or soname; default "librccl.so" (found through
// this library's RUNPATH /opt/rocm/lib).  tests/cpp/fakerccl is a stand-in with the same entry points (two ranks on one GPU).
namespace {
struct RcclApi {
	void *handle = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclCommCount) CommCount = nullptr;
	decltype(&ncclCommUserRank) CommUserRank = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	std::string err, path;
};

RcclApi *rccl_api()
{
	static RcclApi api = [] {
		RcclApi a;
		const char *env = getenv("GYS_RCCL_LIB");
		a.path = env && *env ? env : "librccl.so";
		a.handle = dlopen(a.path.c_str(), RTLD_NOW | RTLD_LOCAL);
		if (!a.handle && !(env && *env)) a.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
		if (!a.handle) {
			const char *e = dlerror();
			a.err = std::string("cannot load ") + a.path + ": " + (e ? e : "?");
			return a;
		}
#define GYS_RCCL_SYM(name)                                                        \
	a.name = (decltype(a.name))dlsym(a.handle, "nccl" #name);                 \
	if (!a.name && a.err.empty()) a.err = a.path + " has no nccl" #name;
		GYS_RCCL_SYM(GetUniqueId)
		GYS_RCCL_SYM(CommInitRank)
		GYS_RCCL_SYM(CommDestroy)
		GYS_RCCL_SYM(CommCount)
		GYS_RCCL_SYM(CommUserRank)
		GYS_RCCL_SYM(GetErrorString)
		GYS_RCCL_SYM(GroupStart)
		GYS_RCCL_SYM(GroupEnd)
		GYS_RCCL_SYM(AllReduce)
		GYS_RCCL_SYM(AllGather)
#undef GYS_RCCL_SYM
		return a;
	}();
	return &api;
}
} // namespace

#define RCCL_API(R)                                            \
	RcclApi *R = rccl_api();                               \
	if (!R->err.empty()) {                                 \
		set_err("RCCL unavailable: %s", R->err.c_str()); \
		return GYS_ERR_HIP;                            \
	}

#define NCCLCHK(expr)                                                                          \
	do {                                                                                   \
		const ncclResult_t r_ = (expr);                                                \
		if (r_ != ncclSuccess) {                                                       \
			set_err("%s failed: %s (%s:%d)", #expr, R->GetErrorString(r_), __FILE__, __LINE__); \
			return GYS_ERR_HIP;                                                    \
		}                                                                              \
	} while (0)

static_assert(sizeof(ncclUniqueId) == GYS_RCCL_UID_BYTES, "gysketch.h carries an ncclUniqueId as 128 opaque bytes");

int gys_rccl_unique_id(uint8_t uid[GYS_RCCL_UID_BYTES])
try {
	if (!uid) return GYS_ERR_INVAL;
	// The exchange is between the GPUs of ONE node: unless the caller chose an interface, the bootstrap rendezvous goes over
	// loopback -- RCCL otherwise takes the first non-loopback interface, and on hosts whose first interface is a tunnel / container
	// bridge the ranks' connect() to it never returns (seen on part of the MI355X pool).  Must be set before RCCL's first call.
	setenv("NCCL_SOCKET_IFNAME", "lo", 0);
	RCCL_API(R);
	ncclUniqueId id;
	NCCLCHK(R->GetUniqueId(&id));
	memcpy(uid, &id, sizeof(id));
	return GYS_OK;
} GYS_CATCH_ALL

int gys_rccl_comm_create(gys_ctx *c, const uint8_t uid[GYS_RCCL_UID_BYTES], int nranks, int rank, void **comm)
try {
	GYS_ENTER(c);
	if (!c || !uid || !comm || nranks < 1 || rank < 0 || rank >= nranks) return GYS_ERR_INVAL;
	if ((uint32_t)nranks != std::max<uint32_t>(c->cfg.nranks, 1) || (uint32_t)rank != c->cfg.rank) {
		set_err("communicator (%d of %d) does not match gys_config rank / nranks (%u of %u)", rank, nranks, c->cfg.rank, c->cfg.nranks);
		return GYS_ERR_INVAL;
	}
	HIPCHK(hipSetDevice(c->device));
	setenv("NCCL_SOCKET_IFNAME", "lo", 0); // (see gys_rccl_unique_id)
	ncclUniqueId id;
	memcpy(&id, uid, sizeof(id));
	RCCL_API(R);
	ncclComm_t cm = nullptr;
	NCCLCHK(R->CommInitRank(&cm, nranks, id, rank));
	*comm = (void *)cm;
	return GYS_OK;
} GYS_CATCH_ALL

int gys_rccl_comm_destroy(void *comm)
try {
	if (!comm) return GYS_ERR_INVAL;
	RCCL_API(R);
	NCCLCHK(R->CommDestroy((ncclComm_t)comm));
	return GYS_OK;
} GYS_CATCH_ALL

int gys_window_close_rccl(gys_ctx *c, void *comm, uint64_t tusec)
try {
	GYS_ENTER(c);
	if (!c || !comm) return GYS_ERR_INVAL;
	RCCL_API(R);
	int rc = GYS_OK;
	if (!c->prepared) rc = gys_window_prepare(c, tusec); // (a call that failed in the exchange below left the window prepared: the retry resumes here)
	if (rc) return rc;
	gys_reduce_section sec[4];
	uint32_t nsec = 0;
	rc = gys_reduce_sections(c, sec, &nsec);
	if (rc) return rc;
	{
		ProfScope ps(c, "window_rccl");
		NCCLCHK(R->GroupStart());
		// from here on the group is ALWAYS closed: an error between Start and End would leave every later RCCL call of this thread
		// inside a group that never launches
		ncclResult_t bad = ncclSuccess;
		for (uint32_t i = 0; i < nsec && bad == ncclSuccess; ++i) {
			const ncclDataType_t dt = sec[i].dtype == 0 ? ncclUint8 : (sec[i].dtype == 1 ? ncclUint32 : ncclInt64);
			const ncclRedOp_t op = sec[i].op == 0 ? ncclMax : ncclSum;
			bad = R->AllReduce(sec[i].dev_ptr, sec[i].dev_ptr, sec[i].nelems, dt, op, (ncclComm_t)comm, c->stream);
		}
		const ncclResult_t end = R->GroupEnd();
		if (bad == ncclSuccess) bad = end;
		if (bad != ncclSuccess) {
			// the window stays prepared (arena not cleared, nothing finished): the caller may retry this call -- it resumes at the
			// exchange -- or fall back to gys_reduce_sections + its own collective + gys_window_finish
			set_err("window exchange failed: %s", R->GetErrorString(bad));
			return GYS_ERR_HIP;
		}
	}
	return gys_window_finish(c);
} GYS_CATCH_ALL

int gys_tdigest_global_rccl(gys_ctx *c, void *comm, gys_tdigest_slab *d_out)
try {
	GYS_ENTER(c);
	if (!c || !comm || !d_out) return GYS_ERR_INVAL;
	TDIGEST_CHECK();
	int nranks = 1, rank = 0;
	RCCL_API(R);
	NCCLCHK(R->CommCount((ncclComm_t)comm, &nranks));
	NCCLCHK(R->CommUserRank((ncclComm_t)comm, &rank));
	gys_tdigest_slab *d_all = nullptr;
	HIPCHK(hipMalloc((void **)&d_all, sizeof(gys_tdigest_slab) * (size_t)nranks));
	int rc = gys_tdigest_rollup_dev(c, GYS_ROLLUP_GLOBAL, d_all + rank); // in place: this rank's slab sits at its own position
	if (rc == GYS_OK) {
		const ncclResult_t r = R->AllGather(d_all + rank, d_all, sizeof(gys_tdigest_slab), ncclUint8, (ncclComm_t)comm, c->stream);
		if (r != ncclSuccess) {
			set_err("ncclAllGather failed: %s", R->GetErrorString(r));
			rc = GYS_ERR_HIP;
		}
	}
	if (rc == GYS_OK) rc = gys_tdigest_merge_slabs_dev(c, d_all, (uint32_t)nranks, d_out); // (synchronises the stream)
	hipFree(d_all);
	return rc;
} GYS_CATCH_ALL

uint32_t gys_num_services(gys_ctx *c) { return c ? c->nsvc : 0; }
uint32_t gys_num_hosts(gys_ctx *c) { return c ? (uint32_t)c->hosts.size() : 0; }

#define RANGE_CHECK(first, n)                                       \
	if (!c || !out || (uint64_t)(first) + (n) > c->nsvc) {      \
		set_err("bad slot range");                          \
		return GYS_ERR_INVAL;                               \
	}

int gys_export_hist(gys_ctx *c, int which, uint32_t first_slot, uint32_t nslots, gys_hist_rec *out)
try {
	GYS_ENTER(c);
	RANGE_CHECK(first_slot, nslots);
	if (which < 0 || which > 1) return GYS_ERR_INVAL;
	if (!nslots) return GYS_OK;
	{
		const int rcf = fold_range(c, first_slot, nslots);
		if (rcf) return rcf;
	}
	gys_hist_rec *tmp = nullptr; // window / all-time VIEW of the records (hist_view in gys_kernels.hpp)
	HIPCHK(hipMalloc((void **)&tmp, (size_t)nslots * sizeof(gys_hist_rec)));
	hipLaunchKernelGGL(k_hist_view, dim3((nslots + 255) / 256), dim3(256), 0, c->stream, window_records(c), c->hist_all,
			   c->cfg.enable_tdigest ? c->td_meta : nullptr, c->epoch, which, first_slot, nslots, tmp);
	hipError_t e = hipMemcpyAsync(out, tmp, (size_t)nslots * sizeof(gys_hist_rec), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	hipFree(tmp);
	HIPCHK(e);
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_conn_bitmap(gys_ctx *c, uint32_t first_slot, uint32_t nslots, uint16_t *out)
try {
	GYS_ENTER(c);
	RANGE_CHECK(first_slot, nslots);
	{
		const int rcf = fold_range(c, first_slot, nslots);
		if (rcf) return rcf;
	}
	HIPCHK(hipMemcpyAsync(out, c->bitmap + (size_t)first_slot * GYS_BM_WORDS, (size_t)nslots * GYS_BM_WORDS * 4, hipMemcpyDeviceToHost, c->stream));
	std::vector<TdMeta> meta;
	if (c->cfg.enable_tdigest) { // rows of a key that has not been touched in the current window are logically cleared
		meta.resize(nslots);
		HIPCHK(hipMemcpyAsync(meta.data(), c->td_meta + first_slot, (size_t)nslots * sizeof(TdMeta), hipMemcpyDeviceToHost, c->stream));
	}
	HIPCHK(hipStreamSynchronize(c->stream));
	for (size_t i = 0; i < meta.size(); ++i)
		if (meta[i].hw_epoch != c->epoch) memset(out + i * 2 * GYS_BM_WORDS, 0, GYS_BM_WORDS * 4);
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_hll(gys_ctx *c, uint8_t *out)
try {
	GYS_ENTER(c);
	if (!c || !out) return GYS_ERR_INVAL;
	HIPCHK(hipMemcpyAsync(out, c->last + c->al.off_hll8, (size_t)1 << GYS_HLL_P, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_cms(gys_ctx *c, int which, void *out)
try {
	GYS_ENTER(c);
	if (!c || !out || which < 0 || which > 1) return GYS_ERR_INVAL;
	const size_t n = (size_t)GYS_CMS_D * GYS_CMS_W;
	if (which == 0)
		HIPCHK(hipMemcpyAsync(out, (const uint32_t *)(c->last + c->al.off_u32) + c->al.u32_cms, n * 4, hipMemcpyDeviceToHost, c->stream));
	else
		HIPCHK(hipMemcpyAsync(out, (const int64_t *)(c->last + c->al.off_i64sum) + c->al.i64_cms, n * 8, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_global_hist(gys_ctx *c, gys_hist_rec *out)
try {
	GYS_ENTER(c);
	if (!c || !out) return GYS_ERR_INVAL;
	int64_t v[32];
	int64_t mx;
	HIPCHK(hipMemcpyAsync(v, (const int64_t *)(c->last + c->al.off_i64sum) + c->al.i64_ghist, sizeof(v), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(&mx, c->last + c->al.off_i64max, 8, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (int i = 0; i < 15; ++i) {
		out->stats[i].count = (uint64_t)v[2 * i];
		out->stats[i].sum = v[2 * i + 1];
	}
	out->total_count = (uint64_t)v[30];
	out->max_val_seen = mx;
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_tdigest(gys_ctx *c, uint32_t first_slot, uint32_t nslots, int64_t *sums, uint32_t *cnts, int32_t *minmax)
try {
	GYS_ENTER(c);
	void *out = sums;
	RANGE_CHECK(first_slot, nslots);
	if (!c->cfg.enable_tdigest || !cnts || !minmax) return GYS_ERR_INVAL;
	{
		const int rcf = fold_range(c, first_slot, nslots); // min / max cover the buffered values
		if (rcf) return rcf;
	}
	HIPCHK(hipMemcpyAsync(sums, c->td_sum + (size_t)first_slot * GYS_TD_NB, (size_t)nslots * GYS_TD_NB * 8, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(cnts, c->td_cnt + (size_t)first_slot * GYS_TD_NB, (size_t)nslots * GYS_TD_NB * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(minmax, c->td_minmax + first_slot, (size_t)nslots * sizeof(int2), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return GYS_OK;
} GYS_CATCH_ALL

uint32_t gys_td_pend_cap(gys_ctx *c) { return c ? c->pend_cap : 0u; }

int gys_export_tdigest_pending(gys_ctx *c, uint32_t first_slot, uint32_t nslots, uint32_t *npend, int32_t *pend)
try {
	GYS_ENTER(c);
	void *out = npend;
	RANGE_CHECK(first_slot, nslots);
	if (!c->cfg.enable_tdigest || !pend) return GYS_ERR_INVAL;
	std::vector<TdMeta> meta(nslots);
	HIPCHK(hipMemcpyAsync(meta.data(), c->td_meta + first_slot, (size_t)nslots * sizeof(TdMeta), hipMemcpyDeviceToHost, c->stream));
	// a service buffers at most td_pend_cap values between batches (the rest of its pcap-entry buffer is room for a batch)
	const size_t cap = c->pend_cap;
	HIPCHK(hipMemcpy2DAsync(pend, cap * 4, c->td_pend + (size_t)first_slot * c->pcap, (size_t)c->pcap * 4, cap * 4, nslots, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (uint32_t i = 0; i < nslots; ++i) {
		npend[i] = meta[i].npend;
		for (uint32_t k = 0; k < cap; ++k) // staged words: value << 6 | family << 5 | CONN_BITMAP row
			pend[(size_t)i * cap + k] = k < meta[i].npend ? (int32_t)((uint32_t)pend[(size_t)i * cap + k] >> GYS_ROW_BITS) : 0;
	}
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_svc_counters(gys_ctx *c, uint32_t first_slot, uint32_t nslots, uint64_t *out)
try {
	GYS_ENTER(c);
	RANGE_CHECK(first_slot, nslots);
	{
		// mid-window: the cumulative counters must include the window so far (its Count-Min share moves into the arena a little early,
		// which nothing can observe before the window closes)
		const int rcf = conn_fold(c);
		if (rcf) return rcf;
	}
	HIPCHK(hipMemcpyAsync(out, c->svc_ctr + (size_t)first_slot * 4, (size_t)nslots * 32, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_svc_hll(gys_ctx *c, uint32_t first_slot, uint32_t nslots, uint8_t *out)
try {
	GYS_ENTER(c);
	RANGE_CHECK(first_slot, nslots);
	if (!c->svc_hll) return GYS_ERR_STATE;
	HIPCHK(hipMemcpyAsync(out, c->svc_hll + ((size_t)first_slot << c->cfg.svc_hll_p), (size_t)nslots << c->cfg.svc_hll_p, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return GYS_OK;
} GYS_CATCH_ALL

int gys_get_counters(gys_ctx *c, gys_counters *out)
try {
	GYS_ENTER(c);
	if (!c || !out) return GYS_ERR_INVAL;
	uint64_t v[32];
	static_assert(CTR_NUM <= 31, "counter block (the last word is the sink of k_read_events)");
	HIPCHK(hipMemcpyAsync(v, c->counters, sizeof(v), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
#ifdef GYS_RESP_TIMING
	{
		unsigned long long t[16];
		HIPCHK(hipMemcpyFromSymbol(t, HIP_SYMBOL(gys::g_resp_timing), sizeof(t)));
		unsigned long long tot = 0;
		for (int i = 0; i < 12; ++i) tot += t[i];
		static const char *nm[12] = {"loads", "probe2", "probe3+", "kept/atomics", "ghist+hash", "hllq/store", "barrier1", "scan1+b", "scan2+b", "image+b", "prologue", "flush"};
		fprintf(stderr, "GYS_RESP_TIMING waves %llu ticks %llu:", t[15], tot);
		for (int i = 0; i < 12; ++i) fprintf(stderr, " %s %.1f%%", nm[i], tot ? 100.0 * (double)t[i] / (double)tot : 0.0);
		fprintf(stderr, "\n");
	}
#endif
#ifdef GYS_HUGE_TIMING
	fprintf(stderr, "GYS_HUGE_TIMING ticks: load %llu words+tail %llu scan %llu assign %llu writeback %llu | entries %llu npend %llu nc %llu\n", (unsigned long long)v[20], (unsigned long long)v[21], (unsigned long long)v[22], (unsigned long long)v[23], (unsigned long long)v[24], (unsigned long long)v[25], (unsigned long long)v[26], (unsigned long long)v[27]);
#endif
	out->resp_events = v[CTR_RESP_EVENTS];
	out->resp_dropped_range = v[CTR_RESP_DROP_RANGE];
	out->resp_dropped_nolistener = v[CTR_RESP_DROP_NOLISTENER];
	out->conn_events = v[CTR_CONN_EVENTS];
	out->conn_unknown_service = v[CTR_CONN_UNKNOWN];
	out->conn_new = v[CTR_CONN_NEW];
	out->conn_closed = v[CTR_CONN_CLOSED];
	out->conn_closed_no_notify = v[CTR_CONN_CLOSED_NO_NOTIFY];
	out->conn_client_side = v[CTR_CONN_CLI_SIDE];
	out->resp_run_overflow = v[CTR_RESP_RUN_OVERFLOW];
	out->lstate_records = v[CTR_LSTATE_RECORDS];
	out->lstate_missed = v[CTR_LSTATE_MISSED];
	out->lstate_errors = v[CTR_LSTATE_ERRORS];
	out->lstate_deleted = v[CTR_LSTATE_DELETED];
	out->resp_batches_host_local = c->n_batches_host_local;
	out->resp_batches_general = c->n_batches_general;
	out->window_graph_launches = c->win_graph_launches;
	out->resp_batches_host_split = c->n_batches_host_split;
	out->td_merges = v[CTR_TD_MERGES];
	out->td_merge_values = v[CTR_TD_MERGE_VALUES];
	out->actconn_records = v[CTR_ACTCONN_RECORDS];
	out->actconn_remote_listen = v[CTR_ACTCONN_REMOTE_LISTEN];
	out->actconn_unknown_listener = v[CTR_ACTCONN_UNKNOWN];
	out->stage_waits = c->stage_waits.load();
	{
		std::lock_guard<std::mutex> g(c->rq.mu);
		out->resp_calls_queued = c->rq.calls;
		out->resp_submissions = c->rq.submissions;
		out->resp_tail_flushes = c->rq.tail_flushes;
	}
	{
		std::lock_guard<std::mutex> g(c->cq[0].mu);
		out->conn_calls_queued = c->cq[0].calls;
		out->conn_submissions = c->cq[0].submissions;
		out->rec_tail_flushes = c->cq[0].tail_flushes;
	}
	{
		std::lock_guard<std::mutex> g(c->cq[1].mu);
		out->lstate_calls_queued = c->cq[1].calls;
		out->lstate_submissions = c->cq[1].submissions;
		out->rec_tail_flushes += c->cq[1].tail_flushes;
	}
	return GYS_OK;
} GYS_CATCH_ALL

int gys_resp_queue_pending(gys_ctx *c, uint64_t *events)
try {
	if (!c || !events) return GYS_ERR_INVAL;
	gys_ctx::RespQ &q = c->rq;
	std::lock_guard<std::mutex> g(q.mu);
	uint64_t n = 0;
	if (q.open >= 0) n += q.b[q.open].fill;
	for (int bi : q.sealed) n += q.b[bi].fill;
	*events = n;
	return GYS_OK;
} GYS_CATCH_ALL

// ------------------------------------------------------------------------------------------------ multi-level windows
#define LEVELS_CHECK()                                                        \
	if (!c->cfg.enable_levels) {                                          \
		set_err("gys_config.enable_levels was not set");              \
		return GYS_ERR_STATE;                                         \
	}
#define LEVEL0_CHECK(level)                                                                   \
	if ((level) == 0 && c->cfg.enable_levels != 1) {                                       \
		set_err("the 5-s level is not kept with gys_config.enable_levels = 2");        \
		return GYS_ERR_STATE;                                                          \
	}

int gys_export_hist_level(gys_ctx *c, int level, uint64_t tusec, uint32_t first_slot, uint32_t nslots, gys_hist_rec *out)
try {
	GYS_ENTER(c);
	RANGE_CHECK(first_slot, nslots);
	LEVELS_CHECK();
	if (level < 0 || level >= GYS_NLEVELS) return GYS_ERR_INVAL;
	LEVEL0_CHECK(level);
	if (!nslots) return GYS_OK;
	gys_hist_rec *tmp = nullptr;
	HIPCHK(hipMalloc((void **)&tmp, (size_t)nslots * sizeof(gys_hist_rec)));
	int rc = level_view(c, level, tusec, first_slot, nslots, tmp);
	if (rc == GYS_OK) {
		hipError_t e = hipMemcpyAsync(out, tmp, (size_t)nslots * sizeof(gys_hist_rec), hipMemcpyDeviceToHost, c->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
		if (e != hipSuccess) {
			set_err("level export copy: %s", hipGetErrorString(e));
			rc = GYS_ERR_HIP;
		}
	}
	hipFree(tmp);
	return rc;
} GYS_CATCH_ALL

int gys_query_hist_level_stats(gys_ctx *c, uint64_t glob_id, int level, uint64_t tusec, gys_time_hist_val *pstats, uint32_t nstats, int64_t *tcount,
			       int64_t *tsum, double *mean_val)
try {
	GYS_ENTER(c);
	if (!c || (!pstats && nstats)) return GYS_ERR_INVAL;
	uint32_t slot;
	int rc = gys_lookup_service(c, glob_id, &slot);
	if (rc) return rc;
	gys_hist_rec h;
	rc = gys_export_hist_level(c, level, tusec, slot, 1, &h);
	if (rc) return rc;
	const HashDef &d = hash_def(GYS_RESP_TIME_HASH);
	int64_t tc = 0, ts = 0;
	for (int b = 0; b < d.nthr + 2; ++b) { // slabhist.count(level) / sum(level), common/gy_statistics.h:1358-1359
		tc += (int64_t)h.stats[b].count;
		ts += h.stats[b].sum;
	}
	for (uint32_t i = 0; i < nstats; ++i) pstats[i].data_value = level_percentile(d, h, pstats[i].percentile);
	if (tcount) *tcount = tc;
	if (tsum) *tsum = ts;
	if (mean_val) *mean_val = (double)ts / (double)(tc != 0 ? tc : 1); // :1361
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_hist_period(gys_ctx *c, int64_t starttime, int64_t endtime, uint64_t tusec, uint32_t first_slot, uint32_t nslots, gys_hist_rec *out,
			   int *level_used)
try {
	GYS_ENTER(c);
	RANGE_CHECK(first_slot, nslots);
	LEVELS_CHECK();
	if (!nslots) return GYS_OK;
	gys_hist_rec *tmp = nullptr;
	HIPCHK(hipMalloc((void **)&tmp, (size_t)nslots * sizeof(gys_hist_rec)));
	int rc = level_period(c, starttime, endtime + 1, tusec, first_slot, nslots, tmp, level_used);
	if (rc == GYS_OK) {
		hipError_t e = hipMemcpyAsync(out, tmp, (size_t)nslots * sizeof(gys_hist_rec), hipMemcpyDeviceToHost, c->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
		if (e != hipSuccess) {
			set_err("period export copy: %s", hipGetErrorString(e));
			rc = GYS_ERR_HIP;
		}
	}
	hipFree(tmp);
	return rc;
} GYS_CATCH_ALL

int gys_query_hist_period_stats(gys_ctx *c, uint64_t glob_id, int64_t starttime, int64_t endtime, uint64_t tusec, gys_time_hist_val *pstats,
				uint32_t nstats, int64_t *tcount, int64_t *tsum, double *mean_val)
try {
	GYS_ENTER(c);
	if (!c || (!pstats && nstats)) return GYS_ERR_INVAL;
	uint32_t slot;
	int rc = gys_lookup_service(c, glob_id, &slot);
	if (rc) return rc;
	gys_hist_rec h;
	rc = gys_export_hist_period(c, starttime, endtime, tusec, slot, 1, &h, nullptr);
	if (rc) return rc;
	const HashDef &d = hash_def(GYS_RESP_TIME_HASH);
	int64_t tc = 0, ts = 0;
	for (int b = 0; b < d.nthr + 2; ++b) { // slabhist.count(start, end) / sum(start, end), common/gy_statistics.h:1395-1396
		tc += (int64_t)h.stats[b].count;
		ts += h.stats[b].sum;
	}
	for (uint32_t i = 0; i < nstats; ++i) pstats[i].data_value = level_percentile(d, h, pstats[i].percentile); // :1389-1393
	if (tcount) *tcount = tc;
	if (tsum) *tsum = ts;
	if (mean_val) *mean_val = (double)ts / (double)(tc != 0 ? tc : 1); // :1398
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_day_stats(gys_ctx *c, uint64_t tusec, uint32_t first_slot, uint32_t nslots, gys_listener_day_stats *out)
try {
	GYS_ENTER(c);
	RANGE_CHECK(first_slot, nslots);
	LEVELS_CHECK();
	if (!nslots) return GYS_OK;
	gys_hist_rec *lv = nullptr;
	gys_listener_day_stats *d_out = nullptr;
	HIPCHK(hipMalloc((void **)&lv, (size_t)nslots * sizeof(gys_hist_rec)));
	hipError_t e = hipMalloc((void **)&d_out, (size_t)nslots * sizeof(gys_listener_day_stats));
	if (e != hipSuccess) {
		hipFree(lv);
		HIPCHK(e);
	}
	int rc = level_view(c, 2, tusec, first_slot, nslots, lv);
	if (rc == GYS_OK) {
		ProfScope ps(c, "day_stats");
		hipLaunchKernelGGL(k_day_stats, dim3((nslots + 255) / 256), dim3(256), 0, c->stream, lv, c->qps_hist, c->act_hist, c->svc_gid, first_slot, nslots,
				   d_out);
		e = hipGetLastError();
		if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)nslots * sizeof(gys_listener_day_stats), hipMemcpyDeviceToHost, c->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
		if (e != hipSuccess) {
			set_err("day stats: %s", hipGetErrorString(e));
			rc = GYS_ERR_HIP;
		}
	}
	hipFree(lv);
	hipFree(d_out);
	return rc;
} GYS_CATCH_ALL

int gys_scan_listener_state_dev(gys_ctx *c, uint64_t tusec, float qps_multiple, uint32_t diffsec, void *d_notify, gys_listener_scan *d_scan)
try {
	GYS_ENTER(c);
	if (!c) return GYS_ERR_INVAL;
	LEVELS_CHECK();
	LEVEL0_CHECK(0);
	if (!c->nsvc || (!d_notify && !d_scan)) return GYS_OK;
	int64_t tq = (int64_t)(tusec / 1000000ull);
	if (tq < c->lvl_t_last) tq = c->lvl_t_last;
	ListenerScanP p{};
	p.win = c->hist_win;
	p.all = c->hist_all;
	p.meta = c->cfg.enable_tdigest ? c->td_meta : nullptr;
	p.epoch_open = c->epoch + (c->prepared ? 1u : 0u);
	p.epoch_last = p.epoch_open - 1u;
	p.nsvc = c->nsvc;
	for (int lv = 0; lv < GYS_NLEVELS; ++lv) level_source(c, lv, tq, &p.mode[lv], &p.sub[lv]);
	p.last_tag = c->cfg.enable_tdigest ? c->lvl_last_tag : nullptr;
	p.last_epoch = c->lvl_last_epoch;
	p.qps = c->qps_hist;
	p.act = c->act_hist;
	p.bitmap = c->bitmap;
	p.svc_gid = c->svc_gid;
	p.multiple = qps_multiple;
	p.diffsec = (float)diffsec;
	p.notify = (uint8_t *)d_notify;
	p.scan = d_scan;
	ProfScope ps(c, "listener_scan");
	hipLaunchKernelGGL(k_listener_scan, dim3((c->nsvc + 255) / 256), dim3(256), 0, c->stream, p);
	HIPCHK(hipGetLastError());
	return GYS_OK;
} GYS_CATCH_ALL

int gys_decide_listener_state_dev(gys_ctx *c, const gys_listener_scan *d_scan, const gys_listener_issue_in *d_issue_in, void *d_notify,
				  gys_listener_decision *d_out)
try {
	GYS_ENTER(c);
	if (!c || !d_scan) return GYS_ERR_INVAL;
	if (!c->nsvc) return GYS_OK;
	if (!c->svc_bithist) { // the listeners' two history bytes: engine state from the first decision on
		HIPCHK(hipMalloc((void **)&c->svc_bithist, (size_t)c->cfg.max_services * 2));
		HIPCHK(hipMemsetAsync(c->svc_bithist, 0, (size_t)c->cfg.max_services * 2, c->stream));
	}
	ListenerDecideP p{};
	p.scan = d_scan;
	p.in = d_issue_in;
	p.hist = c->svc_bithist;
	p.notify = (uint8_t *)d_notify;
	p.out = d_out;
	p.nsvc = c->nsvc;
	p.msec1_bucket = 1; // get_bucketid_from_threshold<RESP_TIME_HASH>(1): the bucket whose ceiling is 1 ms
	for (uint32_t b = 1; b < 14; ++b)
		if (bucket_max_threshold(hash_def(GYS_RESP_TIME_HASH), b) == 1) p.msec1_bucket = b;
	ProfScope ps(c, "listener_decide");
	hipLaunchKernelGGL(k_listener_decide, dim3((c->nsvc + 255) / 256), dim3(256), 0, c->stream, p);
	HIPCHK(hipGetLastError());
	return GYS_OK;
} GYS_CATCH_ALL

int gys_export_svc_hist(gys_ctx *c, int which, uint32_t first_slot, uint32_t nslots, gys_hist_rec *out)
try {
	GYS_ENTER(c);
	RANGE_CHECK(first_slot, nslots);
	LEVELS_CHECK();
	if (which < 0 || which > 1) return GYS_ERR_INVAL;
	if (!nslots) return GYS_OK;
	const gys_hist_rec *src = (which ? c->act_hist : c->qps_hist) + first_slot;
	HIPCHK(hipMemcpyAsync(out, src, (size_t)nslots * sizeof(gys_hist_rec), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return GYS_OK;
} GYS_CATCH_ALL

// ------------------------------------------------------------------------------------------------ standalone keyed histogram op
int gys_hist_init_dev(gys_ctx *c, int kind, gys_hist_rec *d_hist, uint32_t nkeys)
try {
	GYS_ENTER(c);
	if (!c || !d_hist || kind < 0 || kind >= GYS_NUM_HASH_KINDS) return GYS_ERR_INVAL;
	HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)nkeys * sizeof(gys_hist_rec), c->stream));
	const int64_t mn = hash_def(kind).t_bits == 64 ? INT64_MIN : (int64_t)INT32_MIN; // std::numeric_limits<T>::min() (:563)
	hipLaunchKernelGGL(k_hist_init, dim3(grid_for(nkeys, 256, 2048)), dim3(256), 0, c->stream, d_hist, (uint64_t)0, (uint64_t)nkeys, mn);
	HIPCHK(hipGetLastError());
	return GYS_OK;
} GYS_CATCH_ALL

int gys_hist_add_dev(gys_ctx *c, int kind, gys_hist_rec *d_hist, uint32_t nkeys, const uint32_t *d_keyidx, const int32_t *d_vals, uint64_t n)
try {
	GYS_ENTER(c);
	if (!c || !d_hist || kind < 0 || kind >= GYS_NUM_HASH_KINDS || ((!d_keyidx || !d_vals) && n)) return GYS_ERR_INVAL;
	if (!n) return GYS_OK;
	ProfScope ps(c, "hist_add");
	hipLaunchKernelGGL(k_hist_add, dim3(grid_for(n, 256, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, kind, d_hist, nkeys, d_keyidx, d_vals, n);
	HIPCHK(hipGetLastError());
	return GYS_OK;
} GYS_CATCH_ALL

int gys_hist_merge_dev(gys_ctx *c, gys_hist_rec *d_dst, const gys_hist_rec *d_src, uint32_t nkeys)
try {
	GYS_ENTER(c);
	if (!c || !d_dst || !d_src) return GYS_ERR_INVAL;
	if (!nkeys) return GYS_OK;
	ProfScope ps(c, "hist_merge");
	hipLaunchKernelGGL(k_hist_fold, dim3(grid_for((uint64_t)nkeys * 16, 256, (uint32_t)c->ncu * 8)), dim3(256), 0, c->stream, d_dst,
			   (gys_hist_rec *)d_src, (uint64_t)nkeys, 0, (long long *)nullptr, (long long *)nullptr);
	HIPCHK(hipGetLastError());
	return GYS_OK;
} GYS_CATCH_ALL

int gys_hist_percentiles_dev(gys_ctx *c, int kind, const gys_hist_rec *d_hist, uint32_t nkeys, const float *pcts, uint32_t npct, int64_t *d_out)
try {
	GYS_ENTER(c);
	if (!c || !d_hist || !pcts || !d_out || !npct || npct > 64 || kind < 0 || kind >= GYS_NUM_HASH_KINDS) return GYS_ERR_INVAL;
	if (!nkeys) return GYS_OK;
	HIPCHK(hipMemcpyAsync(c->dev_pcts, pcts, (size_t)npct * 4, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream)); // pcts may be a short-lived host buffer
	ProfScope ps(c, "hist_percentiles");
	const uint64_t t = (uint64_t)nkeys * npct;
	hipLaunchKernelGGL(k_hist_percentiles, dim3((uint32_t)((t + 255) / 256)), dim3(256), 0, c->stream, kind, d_hist, nkeys, c->dev_pcts, npct, d_out);
	HIPCHK(hipGetLastError());
	return GYS_OK;
} GYS_CATCH_ALL

// ------------------------------------------------------------------------------------------------ profiling
int gys_profile_enable(gys_ctx *c, int on)
try {
	GYS_ENTER(c);
	if (!c) return GYS_ERR_INVAL;
	c->profile = on != 0;
	return GYS_OK;
} GYS_CATCH_ALL

int gys_profile_reset(gys_ctx *c)
try {
	GYS_ENTER(c);
	if (!c) return GYS_ERR_INVAL;
	HIPCHK(hipStreamSynchronize(c->stream));
	prof_resolve(c);
	c->prof.clear();
	return GYS_OK;
} GYS_CATCH_ALL

int gys_profile_get(gys_ctx *c, const char *kernel, double *total_ms, uint64_t *launches)
try {
	GYS_ENTER(c);
	if (!c || !kernel) return GYS_ERR_INVAL;
	prof_resolve(c);
	auto it = c->prof.find(kernel);
	if (it == c->prof.end()) {
		if (total_ms) *total_ms = 0;
		if (launches) *launches = 0;
		return GYS_ERR_NOTFOUND;
	}
	if (total_ms) *total_ms = it->second.total_ms;
	if (launches) *launches = it->second.launches;
	return GYS_OK;
} GYS_CATCH_ALL

int gys_profile_names(gys_ctx *c, char *buf, size_t buflen)
try {
	GYS_ENTER(c);
	if (!c || !buf || !buflen) return GYS_ERR_INVAL;
	std::string s;
	for (auto &kv : c->prof) {
		if (!s.empty()) s += ",";
		s += kv.first;
	}
	snprintf(buf, buflen, "%s", s.c_str());
	return GYS_OK;
} GYS_CATCH_ALL

// ------------------------------------------------------------------------------------------------ synthetic generator
int gys_debug_read_events_dev(gys_ctx *c, const void *d_ev24, uint64_t nevents)
try {
	GYS_ENTER(c);
	if (!c || !d_ev24) return GYS_ERR_INVAL;
	if (!nevents) return GYS_OK;
	const uint64_t per_wg = 53248; // ~ a C3 host segment (13 tiles of 4096)
	hipLaunchKernelGGL(k_read_events, dim3((uint32_t)((nevents + per_wg - 1) / per_wg)), dim3(1024), 0, c->stream, (const uint64_t *)d_ev24, nevents, per_wg,
			   (uint64_t *)c->counters + 31);
	HIPCHK(hipGetLastError());
	return GYS_OK;
} GYS_CATCH_ALL

int gys_gen_resp_events_dev(gys_ctx *c, void *d_ev24, uint64_t nevents, uint64_t seed, uint32_t first_host, uint32_t nhosts, uint32_t svcs_per_host,
			    uint32_t zipf_milli, gys_resp_seg *segs_out)
try {
	GYS_ENTER(c);
	if (!c || !d_ev24 || !nhosts || !svcs_per_host || !segs_out) return GYS_ERR_INVAL;
	const bool spread = zipf_milli == GYS_GEN_SPREAD; // per-service weights 0..255/256 instead of a Zipf law
	if (spread) zipf_milli = 0;
	if (zipf_milli && (c->zipf_n != svcs_per_host || c->zipf_milli != zipf_milli)) {
		std::vector<float> cdf(svcs_per_host);
		const double s = zipf_milli / 1000.0;
		double tot = 0;
		for (uint32_t k = 0; k < svcs_per_host; ++k) tot += 1.0 / std::pow((double)(k + 1), s);
		double run = 0;
		for (uint32_t k = 0; k < svcs_per_host; ++k) {
			run += 1.0 / std::pow((double)(k + 1), s) / tot;
			cdf[k] = (float)run;
		}
		cdf[svcs_per_host - 1] = 1.0f;
		if (c->zipf_cdf) {
			HIPCHK(hipStreamSynchronize(c->stream));
			HIPCHK(hipFree(c->zipf_cdf));
			c->zipf_cdf = nullptr;
		}
		HIPCHK(hipMalloc((void **)&c->zipf_cdf, (size_t)svcs_per_host * 4));
		HIPCHK(hipMemcpy(c->zipf_cdf, cdf.data(), (size_t)svcs_per_host * 4, hipMemcpyHostToDevice));
		c->zipf_n = svcs_per_host;
		c->zipf_milli = zipf_milli;
	}
	GenP g{};
	g.ev = (uint64_t *)d_ev24;
	g.n = nevents;
	g.seed = seed;
	g.first_host = first_host;
	g.nhosts = nhosts;
	g.svcs_per_host = svcs_per_host;
	g.zipf_cdf = zipf_milli ? c->zipf_cdf : nullptr;
	g.spread = spread ? 1u : 0u;
	g.per_host = std::max<uint64_t>(1, nevents / nhosts);
	for (uint32_t h = 0; h < nhosts; ++h) {
		segs_out[h].host_slot = first_host + h;
		segs_out[h].reserved = 0;
		segs_out[h].first_event = std::min<uint64_t>((uint64_t)h * g.per_host, nevents);
	}
	if (nevents) hipLaunchKernelGGL(k_gen_resp, dim3(grid_for(nevents, 256, 4096)), dim3(256), 0, c->stream, g);
	HIPCHK(hipGetLastError());
	return GYS_OK;
} GYS_CATCH_ALL

} // extern "C"

#include "gys_json.hpp"
#include "gys_regex.hpp"
#include "gys_svcquery_host.hpp"

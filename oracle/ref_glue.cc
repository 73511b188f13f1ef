// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// Thin extern "C" wrapper around the REFERENCE's own headers, compiled where they lie under
// /root/reference by oracle/build_ref.sh into oracle/_ref/libgyref.so.  Nothing from the reference
// is copied into this file: it only #includes gy_common_inc.h / gy_statistics.h / gy_inet_inc.h /
// jhash.h and forwards to the reference's classes so the plain-C restatement in oracle/gy_oracle.c
// can be validated against the real thing (tests/test_oracle_vs_ref.py) and golden vectors can be
// generated (tests/golden/make_golden.py).
//
// Reference entry points wrapped (file:line in /root/reference):
//   common/jhash.h:43-140                jhash, jhash2, jhash_{1,2,3}words
//   common/gy_common_inc.h:1110-1123     get_uint32_hash / get_uint64_hash
//   common/gy_common_inc.h:11226-11244   IP_PORT::get_hash
//   common/gy_inet_inc.h:136-160         NS_IP_PORT::get_hash
//   common/gy_inet_inc.h:225-247         PAIR_IP_PORT::get_hash
//   common/gy_sys_hardware.h:82-85       GY_MACHINE_ID::get_hash (== jhash2 over the 16 id bytes; that header is
//                                        not buildable here (needs gy_netif.h -> libmnl), so the wrapper calls the
//                                        reference jhash2 on the same 4 words)
//   common/gy_statistics.h:455-894       HIST_SERIAL, GY_HISTOGRAM (add_data, add_histogram, get_percentiles ...)
//   common/gy_statistics.h:1565-2063     bucket-hash classes
//   common/gy_statistics.h:28-453        BOUNDED_PRIO_QUEUE
//   common/gy_comm_proto.h:336-420, :486-500, :1620-1653, :1665-1760, :2183-2254   COMM_HEADER, EVENT_NOTIFY, LISTENER_DAY_STATS,
//                                        TCP_CONN_NOTIFY, LISTENER_STATE_NOTIFY: sizes, member offsets, get_elem_size
//   common/gy_comm_proto.cc:10-57, :840-881, :955-996   COMM_HEADER::validate, TCP_CONN_NOTIFY::validate, LISTENER_STATE_NOTIFY::validate
//   thirdparty/SlabHistogramBucket.h:70-78, :165-240   SlabHistogramBuckets::getBucketIdx / getPercentileBucketIdx, constructed the
//                                        way TIME_HISTOGRAM constructs its slab histogram (common/gy_statistics.h:1106-1108)
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include "gy_common_inc.h"
#include "gy_statistics.h"
#include "gy_inet_inc.h"
#include "gy_comm_proto.h"       // wire structs + validators (gy_comm_proto.cc is compiled next to this file, see build_ref.sh)
#if __has_include("ref_listen_summ_stats.h") && __has_include("ref_cluster_state_one.h")
#define GYREF_HAS_SUMM 1
namespace gyeeta {
#include "ref_listen_summ_stats.h"  // the reference's LISTEN_SUMM_STATS<T>, text cut out of server/gy_msocket.h by build_ref.sh
#include "ref_cluster_state_one.h"  // the reference's CLUSTER_STATE_ONE, text cut out of server/gy_mconnhdlr.cc
}
#else
#define GYREF_HAS_SUMM 0
#endif
#include "SlabHistogramBucket.h" // thirdparty/ (in tree): folly::detail::SlabHistogramBuckets, the container behind TimeseriesSlabHistogram

namespace gyeeta {
// originals: common/gy_file_api.cc:60-63 (that TU needs the full build, so the three globals are defined here)
bool guse_utc_time = false;
int gdebugexecn = 0;
thread_local GY_THR_LOCAL gthrdata_local_;
}

using namespace gyeeta;

namespace {

struct HistBase {
	virtual ~HistBase() {}
	virtual size_t nbuckets() const = 0;
	virtual size_t add(int64_t v) = 0;
	virtual size_t bucket_of(int64_t v) const = 0;
	virtual int64_t bucket_max_threshold(size_t id) const = 0;
	virtual void percentiles(const float *pcts, size_t n, int64_t *vals, int64_t *sums, uint64_t *counts,
				 uint64_t *total, int64_t *maxv, float *avg) const = 0;
	virtual void serialized(uint64_t *counts, int64_t *sums, uint64_t *total, int64_t *maxv) const = 0;
	virtual void merge_from(const HistBase *other) = 0;
	virtual void clear() = 0;
	virtual size_t object_size() const = 0;
};

template <typename T, typename Hash>
struct HistImpl final : HistBase {
	using H = GY_HISTOGRAM<T, Hash>;
	H h{1};

	size_t nbuckets() const override { return H::maxbuckets_; }
	size_t add(int64_t v) override { return h.add_data((T)v, 1); }
	size_t bucket_of(int64_t v) const override { return Hash()((T)v); }
	int64_t bucket_max_threshold(size_t id) const override { return (int64_t)(T)get_bucket_max_threshold<Hash, T>(id); }

	void percentiles(const float *pcts, size_t n, int64_t *vals, int64_t *sums, uint64_t *counts, uint64_t *total,
			 int64_t *maxv, float *avg) const override
	{
		std::vector<HIST_DATA> d(n);
		for (size_t i = 0; i < n; ++i) d[i].percentile = pcts[i];
		size_t tc = 0;
		T mv = 0;
		h.get_percentiles(d.data(), n, tc, mv, avg);
		for (size_t i = 0; i < n; ++i) {
			vals[i] = d[i].data_value;
			sums[i] = d[i].sum;
			counts[i] = d[i].count;
		}
		*total = tc;
		*maxv = (int64_t)mv;
	}

	void serialized(uint64_t *counts, int64_t *sums, uint64_t *total, int64_t *maxv) const override
	{
		HIST_SERIAL arr[H::maxbuckets_];
		size_t tc;
		T mv;
		uint64_t e, s;
		h.get_serialized(arr, tc, mv, e, s);
		for (size_t i = 0; i < H::maxbuckets_; ++i) {
			counts[i] = arr[i].count;
			sums[i] = arr[i].sum;
		}
		*total = tc;
		*maxv = (int64_t)mv;
	}

	void merge_from(const HistBase *other) override { h.add_histogram(static_cast<const HistImpl *>(other)->h); }
	void clear() override { h.clear(); }
	size_t object_size() const override { return sizeof(H); }
};

HistBase *make_hist(int kind)
{
	switch (kind) {
	case 0: return new HistImpl<int64_t, RESP_TIME_HASH>();
	case 1: return new HistImpl<int, SEMI_LOG_HASH>();
	case 2: return new HistImpl<int, SEMI_LOG_HASH_LO>();
	case 3: return new HistImpl<int, DURATION_HASH>();
	case 4: return new HistImpl<int, HASH_10_5000>();
	case 5: return new HistImpl<int, HASH_5_250>();
	case 6: return new HistImpl<int, HASH_1_3000>();
	case 7: return new HistImpl<int, PERCENT_HASH>();                          // as used by test_histogram.cc CPUHistogram
	case 8: return new HistImpl<int8_t, FIXED_DIFF_HASH<int8_t, 9, 26, 5>>();   // test_histogram.cc Hist_9_26
	case 9: return new HistImpl<int, FIXED_DIFF_HASH<int, -15, -3, 4>>();       // test_histogram.cc Hist_n4
	default: return nullptr;
	}
}

GY_IP_ADDR mk_ip(const uint8_t *ip, int is_v6)
{
	if (is_v6) {
		unsigned __int128 v;
		std::memcpy(&v, ip, 16);
		return GY_IP_ADDR(v);
	}
	uint32_t v4;
	std::memcpy(&v4, ip, 4);
	return GY_IP_ADDR(v4);
}

}  // namespace

extern "C" {

uint32_t ref_jhash(const void *key, uint32_t len, uint32_t initval) { return jhash(key, len, initval); }
uint32_t ref_jhash2(const uint32_t *k, uint32_t nwords, uint32_t initval) { return jhash2(k, nwords, initval); }
uint32_t ref_jhash_3words(uint32_t a, uint32_t b, uint32_t c, uint32_t iv) { return jhash_3words(a, b, c, iv); }
uint32_t ref_jhash_2words(uint32_t a, uint32_t b, uint32_t iv) { return jhash_2words(a, b, iv); }
uint32_t ref_jhash_1word(uint32_t a, uint32_t iv) { return jhash_1word(a, iv); }
uint32_t ref_get_uint64_hash(uint64_t k) { return get_uint64_hash(k); }
uint32_t ref_get_uint32_hash(uint32_t k) { return get_uint32_hash(k); }

// ip: 4 bytes (network order, as GY_IP_ADDR(uint32_t ip32_be)) or 16 bytes (in6_addr); port in host order
uint32_t ref_ip_port_hash(const uint8_t *ip, int is_v6, uint16_t port, int ignore_ip)
{
	return IP_PORT(mk_ip(ip, is_v6), port).get_hash(!!ignore_ip);
}

uint32_t ref_ns_ip_port_hash(const uint8_t *ip, int is_v6, uint16_t port, uint64_t inode, int ignore_ip)
{
	return NS_IP_PORT(mk_ip(ip, is_v6), port, (ino_t)inode).get_hash(!!ignore_ip);
}

uint32_t ref_pair_ip_port_hash(const uint8_t *cip, int c6, uint16_t cport, const uint8_t *sip, int s6, uint16_t sport)
{
	return PAIR_IP_PORT(IP_PORT(mk_ip(cip, c6), cport), IP_PORT(mk_ip(sip, s6), sport)).get_hash();
}

// GY_IP_ADDR built from raw address bytes the way the response-event handlers build theirs (GY_IP_ADDR(uint32_t) for tup.saddr,
// common/gy_socket_stat.cc:1529; GY_IP_ADDR(unsigned __int128) :1547): its ip32_be_ (shared with embedded_ipv4_), its 16 ip128 bytes, what
// get_as_inaddr hands to the hashes, and is_any_address()
int ref_ip_addr_norm(const uint8_t *ip, int is_v6, uint32_t *ip32, uint8_t ip128[16], uint8_t inaddr[16], uint32_t *inaddr_len)
{
	const GY_IP_ADDR a = mk_ip(ip, is_v6);
	*ip32 = a.get_ipv4_be();
	const unsigned __int128 v6 = a.get_ipv6_addr_be();
	std::memcpy(ip128, &v6, 16);
	std::memset(inaddr, 0, 16);
	*inaddr_len = (uint32_t)a.get_as_inaddr(inaddr);
	return a.is_any_address() ? 1 : 0;
}

int ref_ip_addr_equal(const uint8_t *a, int a6, const uint8_t *b, int b6) { return mk_ip(a, a6) == mk_ip(b, b6); }

// the comparator the listener table is searched with: operator==(const std::shared_ptr<TCP_LISTENER> &, const NS_IP_PORT &)
// (common/gy_socket_stat.h:708-714).  TCP_LISTENER itself cannot be compiled here (its header needs liburcu / folly / the task handler);
// the expression is evaluated on the reference's own NS_IP_PORT / GY_IP_ADDR objects: listener.ns_ip_port_ = NS_IP_PORT(laddr, lport, lns),
// listener.is_any_ip_ as given (= addr.is_any_address() at construction, common/gy_socket_stat.cc:1796), ser = the event's NS_IP_PORT
int ref_listener_match(const uint8_t *lip, int l6, uint16_t lport, uint64_t lns, int l_is_any, const uint8_t *eip, int e6, uint16_t eport, uint64_t ens)
{
	const NS_IP_PORT l(mk_ip(lip, l6), lport, (ino_t)lns), ser(mk_ip(eip, e6), eport, (ino_t)ens);
	return ((l.inode_ == ser.inode_) && (l.ip_port_.port_ == ser.ip_port_.port_) && (l_is_any || (l.ip_port_.ipaddr_ == ser.ip_port_.ipaddr_))) ? 1 : 0;
}

uint32_t ref_machine_id_hash(uint64_t first, uint64_t second)
{
	std::pair<uint64_t, uint64_t> machid(first, second);
	return jhash2((uint32_t *)(&machid), sizeof(machid) / sizeof(uint32_t), 0xceedfead);
}

// get_bucketid_from_threshold<RESP_TIME_HASH> (common/gy_statistics.h:517-531): what TCP_LISTENER::get_curr_state compares
// (common/gy_socket_stat.cc:2085-2087)
size_t ref_resp_bucketid_from_threshold(int64_t threshold) { return get_bucketid_from_threshold<RESP_TIME_HASH>(threshold); }

// LISTEN_SUMM_STATS<int>::update (server/gy_msocket.h:856-868) over n fixed-size LISTENER_STATE_NOTIFY records and
// CLUSTER_STATE_ONE::update_from_state (server/gy_mconnhdlr.cc:16034-16049): the reference's own classes, cut out by build_ref.sh
int ref_has_summ_stats(void) { return GYREF_HAS_SUMM; }
#if GYREF_HAS_SUMM
void ref_listen_summ_update(const uint8_t *recs88, int n, int32_t out[13])
{
	LISTEN_SUMM_STATS<int> s;
	for (int i = 0; i < n; ++i) {
		comm::LISTENER_STATE_NOTIFY r;
		std::memcpy((void *)&r, recs88 + (size_t)i * sizeof(r), sizeof(r));
		if (r.curr_state_ <= OBJ_STATE_E::STATE_DOWN) s.update(r); // (the guard of the caller, server/gy_mconnhdlr.cc:11252-11258)
	}
	for (int i = 0; i < 6; ++i) out[i] = s.nstates_[i];
	out[6] = s.tot_qps_; out[7] = s.tot_act_conn_; out[8] = s.tot_kb_inbound_; out[9] = s.tot_kb_outbound_; out[10] = s.tot_ser_errors_;
	out[11] = s.nlisteners_; out[12] = s.nactive_;
}
void ref_cluster_state_update(uint32_t st[11], uint32_t ntasks_issue, uint32_t ntasks, uint32_t nlisten_issue, uint32_t nlisten, int cpu_issue, int mem_issue,
			      const int32_t summ[13])
{
	CLUSTER_STATE_ONE c;
	std::memcpy((void *)static_cast<comm::MS_CLUSTER_STATE::STATE_ONE *>(&c), st, 44);
	comm::HOST_STATE_NOTIFY h;
	h.ntasks_issue_ = ntasks_issue; h.ntasks_ = ntasks; h.nlisten_issue_ = nlisten_issue; h.nlisten_ = nlisten; h.cpu_issue_ = !!cpu_issue; h.mem_issue_ = !!mem_issue;
	LISTEN_SUMM_STATS<int> s;
	for (int i = 0; i < 6; ++i) s.nstates_[i] = summ[i];
	s.tot_qps_ = summ[6]; s.tot_act_conn_ = summ[7]; s.tot_kb_inbound_ = summ[8]; s.tot_kb_outbound_ = summ[9]; s.tot_ser_errors_ = summ[10];
	s.nlisteners_ = summ[11]; s.nactive_ = summ[12];
	c.update_from_state(h, s);
	std::memcpy(st, (const void *)static_cast<const comm::MS_CLUSTER_STATE::STATE_ONE *>(&c), 44);
}
#endif

size_t ref_sizeof(int what)
{
	switch (what) {
	case 0: return sizeof(GY_IP_ADDR);
	case 1: return sizeof(IP_PORT);
	case 2: return sizeof(PAIR_IP_PORT);
	case 3: return sizeof(NS_IP_PORT);
	case 4: return sizeof(HIST_SERIAL);
	case 5: return sizeof(GY_HISTOGRAM<int64_t, RESP_TIME_HASH>);
	case 6: return sizeof(HIST_DATA);
	default: return 0;
	}
}

// Raw object bytes of an IP_PORT built the way the reference builds it (for layout checks of the wire restatement)
void ref_ip_port_bytes(const uint8_t *ip, int is_v6, uint16_t port, uint8_t out[32])
{
	IP_PORT p(mk_ip(ip, is_v6), port);
	std::memset(out, 0, 32);
	std::memcpy(out, &p.ipaddr_, sizeof(GY_IP_ADDR));
	std::memcpy(out + offsetof(IP_PORT, port_), &p.port_, 2);
}

void *ref_hist_new(int kind) { return make_hist(kind); }
void ref_hist_free(void *h) { delete static_cast<HistBase *>(h); }
size_t ref_hist_nbuckets(void *h) { return static_cast<HistBase *>(h)->nbuckets(); }
size_t ref_hist_object_size(void *h) { return static_cast<HistBase *>(h)->object_size(); }
size_t ref_hist_add(void *h, int64_t v) { return static_cast<HistBase *>(h)->add(v); }
void ref_hist_add_many(void *h, const int64_t *v, size_t n)
{
	auto *p = static_cast<HistBase *>(h);
	for (size_t i = 0; i < n; ++i) p->add(v[i]);
}
size_t ref_hist_bucket_of(void *h, int64_t v) { return static_cast<HistBase *>(h)->bucket_of(v); }
void ref_hist_bucket_of_many(void *h, const int64_t *v, size_t n, uint32_t *out)
{
	auto *p = static_cast<HistBase *>(h);
	for (size_t i = 0; i < n; ++i) out[i] = (uint32_t)p->bucket_of(v[i]);
}
int64_t ref_hist_bucket_max_threshold(void *h, size_t id) { return static_cast<HistBase *>(h)->bucket_max_threshold(id); }
void ref_hist_percentiles(void *h, const float *pcts, size_t n, int64_t *vals, int64_t *sums, uint64_t *counts,
			  uint64_t *total, int64_t *maxv, float *avg)
{
	static_cast<HistBase *>(h)->percentiles(pcts, n, vals, sums, counts, total, maxv, avg);
}
void ref_hist_serialized(void *h, uint64_t *counts, int64_t *sums, uint64_t *total, int64_t *maxv)
{
	static_cast<HistBase *>(h)->serialized(counts, sums, total, maxv);
}
void ref_hist_merge(void *dst, void *src) { static_cast<HistBase *>(dst)->merge_from(static_cast<HistBase *>(src)); }
void ref_hist_clear(void *h) { static_cast<HistBase *>(h)->clear(); }

// BOUNDED_PRIO_QUEUE<uint64_t, std::greater<>> top-N: feeds vals in order through push(); returns the retained values
// sorted descending (the *multiset* of retained values is what is order independent; see DESIGN.md top-N note).
size_t ref_topn_u64(const uint64_t *vals, size_t n, size_t maxn, uint64_t *out)
{
	struct GT { bool operator()(uint64_t a, uint64_t b) const noexcept { return a > b; } };
	BOUNDED_PRIO_QUEUE<uint64_t, GT> q(maxn);
	for (size_t i = 0; i < n; ++i) {
		uint64_t v = vals[i];
		q.push_locked(std::move(v));
	}
	std::vector<uint64_t> r(q.vecq_.begin(), q.vecq_.end());
	std::sort(r.begin(), r.end(), std::greater<uint64_t>());
	for (size_t i = 0; i < r.size(); ++i) out[i] = r[i];
	return r.size();
}


// ---- the reference's own per-event work on the response path, keyed, for the CPU baseline of bench.py (kind "reference"):
// listener lookup by (host, netns, port) in an unordered_map hashed with the reference's GY_JHASHER (stand-in for the liburcu RCU
// table listener_tbl_, which is not buildable here), then GY_HISTOGRAM<int64_t, RESP_TIME_HASH>::add_data -- the reference class
// itself, common/gy_statistics.h:596-623 -- per event; the tresp filter as common/gy_socket_stat.cc:1519-1524.
struct RefKeyed {
	std::unordered_map<uint64_t, uint32_t, GY_JHASHER<uint64_t>> tbl;
	std::vector<GY_HISTOGRAM<int64_t, RESP_TIME_HASH>> hist;
};

void *ref_keyed_new(void) { return new RefKeyed(); }
void ref_keyed_free(void *p) { delete static_cast<RefKeyed *>(p); }
uint32_t ref_keyed_register(void *p, uint32_t host, uint32_t netns, uint16_t port)
{
	RefKeyed *k = static_cast<RefKeyed *>(p);
	const uint64_t key = ((uint64_t)host << 48) | ((uint64_t)netns << 16) | port;
	auto it = k->tbl.find(key);
	if (it != k->tbl.end()) return it->second;
	const uint32_t idx = (uint32_t)k->hist.size();
	k->hist.emplace_back(1);
	k->tbl.emplace(key, idx);
	return idx;
}
void ref_keyed_register_bulk(void *p, uint32_t host, const uint32_t *netns, const uint16_t *port, uint32_t n)
{
	for (uint32_t i = 0; i < n; ++i) ref_keyed_register(p, host, netns[i], port[i]);
}
// returns the number of events added to a histogram
uint64_t ref_keyed_resp_batch(void *p, const uint8_t *ev24, uint64_t n, const uint32_t *seg_host, const uint64_t *seg_first, uint32_t nsegs)
{
	RefKeyed *k = static_cast<RefKeyed *>(p);
	uint64_t added = 0;
	uint32_t seg = 0;
	for (uint64_t i = 0; i < n; ++i) {
		const uint8_t *e = ev24 + i * 24;
		uint32_t netns, lsnd, lrcv;
		uint16_t sport_be;
		std::memcpy(&netns, e + 8, 4);
		std::memcpy(&sport_be, e + 12, 2);
		std::memcpy(&lsnd, e + 16, 4);
		std::memcpy(&lrcv, e + 20, 4);
		while (seg + 1 < nsegs && seg_first[seg + 1] <= i) seg++;
		const uint32_t tresp = lsnd - lrcv;
		if (tresp > 1000000u) continue;
		const uint16_t sport = (uint16_t)((sport_be >> 8) | (sport_be << 8));
		auto it = k->tbl.find(((uint64_t)seg_host[seg] << 48) | ((uint64_t)netns << 16) | sport);
		if (it == k->tbl.end()) continue;
		k->hist[it->second].add_data((int64_t)tresp, 1);
		added++;
	}
	return added;
}
// the same loop on nthreads host threads: the segments (hosts) are cut into contiguous ranges, one per thread -- a host's listeners are
// only ever touched by the thread that owns the host (the reference pins a connection's batches to one L2 queue the same way,
// server/gy_mconnhdlr.cc:16252); the listener map is read-only here
uint64_t ref_keyed_resp_batch_mt(void *p, const uint8_t *ev24, uint64_t n, const uint32_t *seg_host, const uint64_t *seg_first, uint32_t nsegs, uint32_t nthreads)
{
	if (nthreads <= 1 || nsegs <= 1) return ref_keyed_resp_batch(p, ev24, n, seg_host, seg_first, nsegs);
	if (nthreads > nsegs) nthreads = nsegs;
	std::vector<uint64_t> added(nthreads, 0);
	std::vector<std::thread> th;
	for (uint32_t t = 0; t < nthreads; ++t) {
		th.emplace_back([&, t]() {
			const uint32_t s0 = (uint32_t)((uint64_t)nsegs * t / nthreads), s1 = (uint32_t)((uint64_t)nsegs * (t + 1) / nthreads);
			if (s0 >= s1) return;
			const uint64_t e0 = seg_first[s0], e1 = s1 < nsegs ? seg_first[s1] : n;
			std::vector<uint64_t> first(seg_first + s0, seg_first + s1);
			for (auto &f : first) f -= e0;
			added[t] = ref_keyed_resp_batch(p, ev24 + e0 * 24, e1 - e0, seg_host + s0, first.data(), s1 - s0);
		});
	}
	for (auto &x : th) x.join();
	uint64_t tot = 0;
	for (uint64_t a : added) tot += a;
	return tot;
}
uint64_t ref_keyed_total(void *p, uint32_t idx) { return static_cast<RefKeyed *>(p)->hist[idx].get_total_count(); }

// ---- the exact CPU answer to what the HyperLogLog / Count-Min of the TCP_CONN_NOTIFY roll-up estimate (kind
// "reference"): every record's flow key is the reference's own PAIR_IP_PORT(nat_cli_, nat_ser_) (server/gy_mconnhdlr.cc:8707) in an
// unordered_set hashed with PAIR_IP_PORT::get_hash (common/gy_inet_inc.h:225-247) -- the stand-in for glob_tcp_conn_tbl_, an RCU hash table
// keyed the same way -- and the per-listener counters sit in an unordered_map<glob_id, ..., GY_JHASHER> (listen_tbl_).  Records are walked
// with the reference's stride rule (get_elem_size); a connection is counted as the walk of partha_tcp_conn_info counts it (:9129-9341).
struct RefPairHash {
	size_t operator()(const PAIR_IP_PORT &p) const noexcept { return p.get_hash(); }
};
struct RefConnCtr {
	uint64_t nconn = 0, nclose = 0, bytes_sent = 0, bytes_rcvd = 0;
};
struct RefConnExact {
	std::unordered_set<PAIR_IP_PORT, RefPairHash> flows;
	std::unordered_map<uint64_t, RefConnCtr, GY_JHASHER<uint64_t>> svc;
};
void *ref_conn_exact_new(void) { return new RefConnExact(); }
void ref_conn_exact_free(void *p) { delete static_cast<RefConnExact *>(p); }
uint64_t ref_conn_exact_batch(void *p, const uint8_t *batch, uint64_t nrec, const uint8_t *pend)
{
	RefConnExact *x = static_cast<RefConnExact *>(p);
	const uint8_t *q = batch;
	uint64_t i = 0;
	for (; i < nrec && q < pend; ++i) {
		const comm::TCP_CONN_NOTIFY *c = reinterpret_cast<const comm::TCP_CONN_NOTIFY *>(q);
		x->flows.emplace(c->nat_cli_, c->nat_ser_);
		if (c->is_tcp_accept_event_ || !c->is_tcp_connect_event_) { // the accepting half (or neither: add_tcp_conn_ser, :9333)
			RefConnCtr &r = x->svc[c->ser_glob_id_];
			r.nconn += !c->notified_before_;
			r.nclose += !!c->tusec_close_;
			r.bytes_sent += c->bytes_sent_;
			r.bytes_rcvd += c->bytes_rcvd_;
		}
		q += c->get_elem_size();
	}
	return i;
}
// the same on nthreads host threads over contiguous ranges of FIXED-STRIDE records (bench.py's device batches are 280-byte records grouped by
// partha): every thread keeps a set and a map of its own (a partha's connections reach one L2 thread, server/gy_mconnhdlr.cc:16252); the
// distinct-flow count is the sum of the sets' sizes when no flow key spans two ranges -- the caller passes ranges cut at host boundaries
uint64_t ref_conn_exact_batch_mt(const uint8_t *batch, const uint64_t *range_first, uint32_t nranges, uint64_t nrec, uint32_t nthreads, uint64_t *distinct,
				 uint64_t *nconn)
{
	if (nthreads < 1) nthreads = 1;
	if (nthreads > nranges) nthreads = nranges;
	std::vector<uint64_t> d(nthreads, 0), nc(nthreads, 0), walked(nthreads, 0);
	std::vector<std::thread> th;
	for (uint32_t t = 0; t < nthreads; ++t)
		th.emplace_back([&, t]() {
			const uint32_t r0 = (uint32_t)((uint64_t)nranges * t / nthreads), r1 = (uint32_t)((uint64_t)nranges * (t + 1) / nthreads);
			if (r0 >= r1) return;
			const uint64_t e0 = range_first[r0], e1 = r1 < nranges ? range_first[r1] : nrec;
			RefConnExact x;
			walked[t] = ref_conn_exact_batch(&x, batch + e0 * sizeof(comm::TCP_CONN_NOTIFY), e1 - e0, batch + e1 * sizeof(comm::TCP_CONN_NOTIFY));
			d[t] = x.flows.size();
			for (const auto &kv : x.svc) nc[t] += kv.second.nconn;
		});
	for (auto &x : th) x.join();
	uint64_t w = 0;
	*distinct = 0;
	*nconn = 0;
	for (uint32_t t = 0; t < nthreads; ++t) {
		w += walked[t];
		*distinct += d[t];
		*nconn += nc[t];
	}
	return w;
}
uint64_t ref_conn_exact_distinct(void *p) { return static_cast<RefConnExact *>(p)->flows.size(); }
uint64_t ref_conn_exact_services(void *p) { return static_cast<RefConnExact *>(p)->svc.size(); }
int ref_conn_exact_get(void *p, uint64_t glob_id, uint64_t out[4])
{
	RefConnExact *x = static_cast<RefConnExact *>(p);
	auto it = x->svc.find(glob_id);
	if (it == x->svc.end()) return 0;
	out[0] = it->second.nconn;
	out[1] = it->second.nclose;
	out[2] = it->second.bytes_sent;
	out[3] = it->second.bytes_rcvd;
	return 1;
}


// the slab-histogram bucket container of TIME_HISTOGRAM<RESP_TIME_HASH, ...>, with a bare counter as the bucket type
struct SlabCount {
	uint64_t count = 0;
};
using RespSlab = folly::detail::SlabHistogramBuckets<int64_t, SlabCount, RESP_TIME_HASH>;
static RespSlab make_resp_slab()
{
	return RespSlab(RESP_TIME_HASH::max_buckets - 2, RESP_TIME_HASH::get_threshold_array(), RESP_TIME_HASH::min_value, RESP_TIME_HASH::max_value, SlabCount());
}
size_t ref_slab_num_buckets(void) { return make_resp_slab().getNumBuckets(); }
size_t ref_slab_bucket_idx(int64_t value) { return make_resp_slab().getBucketIdx(value); }
// counts[nb] with nb == ref_slab_num_buckets(); pct in [0, 1] (TimeseriesSlabHistogram::getPercentileBucketIdx passes pct / 100.0)
size_t ref_slab_percentile_idx(const uint64_t *counts, size_t nb, double pct)
{
	RespSlab b = make_resp_slab();
	for (size_t i = 0; i < nb && i < b.getNumBuckets(); ++i) b.getByIndex(i).count = counts[i];
	return b.getPercentileBucketIdx(pct, [](const SlabCount &x) { return x.count; });
}

// ---- wire structs: layout facts as the compiler sees them + the reference's validators
#define GY_REF_OFF(S, f) {#S "." #f, offsetof(comm::S, f##_)}
static const struct {
	const char *name;
	size_t off;
} g_comm_offs[] = {
	GY_REF_OFF(COMM_HEADER, magic), GY_REF_OFF(COMM_HEADER, total_sz), GY_REF_OFF(COMM_HEADER, data_type), GY_REF_OFF(COMM_HEADER, padding_sz),
	GY_REF_OFF(EVENT_NOTIFY, subtype), GY_REF_OFF(EVENT_NOTIFY, nevents),
	GY_REF_OFF(TCP_CONN_NOTIFY, cli), GY_REF_OFF(TCP_CONN_NOTIFY, ser), GY_REF_OFF(TCP_CONN_NOTIFY, nat_cli), GY_REF_OFF(TCP_CONN_NOTIFY, nat_ser),
	GY_REF_OFF(TCP_CONN_NOTIFY, tusec_start), GY_REF_OFF(TCP_CONN_NOTIFY, tusec_close), GY_REF_OFF(TCP_CONN_NOTIFY, cli_task_aggr_id),
	GY_REF_OFF(TCP_CONN_NOTIFY, cli_related_listen_id), GY_REF_OFF(TCP_CONN_NOTIFY, cli_madhava_id), GY_REF_OFF(TCP_CONN_NOTIFY, cli_ser_machine_id),
	GY_REF_OFF(TCP_CONN_NOTIFY, ser_related_listen_id), GY_REF_OFF(TCP_CONN_NOTIFY, ser_glob_id), GY_REF_OFF(TCP_CONN_NOTIFY, ser_madhava_id),
	GY_REF_OFF(TCP_CONN_NOTIFY, bytes_sent), GY_REF_OFF(TCP_CONN_NOTIFY, bytes_rcvd), GY_REF_OFF(TCP_CONN_NOTIFY, cli_pid), GY_REF_OFF(TCP_CONN_NOTIFY, ser_pid),
	GY_REF_OFF(TCP_CONN_NOTIFY, ser_conn_hash), GY_REF_OFF(TCP_CONN_NOTIFY, ser_sock_inode), GY_REF_OFF(TCP_CONN_NOTIFY, cli_comm),
	GY_REF_OFF(TCP_CONN_NOTIFY, ser_comm), GY_REF_OFF(TCP_CONN_NOTIFY, cli_cmdline_len), GY_REF_OFF(TCP_CONN_NOTIFY, is_tcp_connect_event),
	GY_REF_OFF(TCP_CONN_NOTIFY, is_tcp_accept_event), GY_REF_OFF(TCP_CONN_NOTIFY, is_loopback_conn), GY_REF_OFF(TCP_CONN_NOTIFY, is_pre_existing),
	GY_REF_OFF(TCP_CONN_NOTIFY, notified_before), GY_REF_OFF(TCP_CONN_NOTIFY, padding_len),
	GY_REF_OFF(LISTENER_STATE_NOTIFY, glob_id), GY_REF_OFF(LISTENER_STATE_NOTIFY, nqrys_5s), GY_REF_OFF(LISTENER_STATE_NOTIFY, total_resp_5sec),
	GY_REF_OFF(LISTENER_STATE_NOTIFY, nconns), GY_REF_OFF(LISTENER_STATE_NOTIFY, nconns_active), GY_REF_OFF(LISTENER_STATE_NOTIFY, ntasks),
	GY_REF_OFF(LISTENER_STATE_NOTIFY, p95_5s_resp_ms), GY_REF_OFF(LISTENER_STATE_NOTIFY, p95_5min_resp_ms),
	GY_REF_OFF(LISTENER_STATE_NOTIFY, curr_kbytes_inbound), GY_REF_OFF(LISTENER_STATE_NOTIFY, curr_kbytes_outbound),
	GY_REF_OFF(LISTENER_STATE_NOTIFY, ser_errors), GY_REF_OFF(LISTENER_STATE_NOTIFY, cli_errors), GY_REF_OFF(LISTENER_STATE_NOTIFY, tasks_delay_usec),
	GY_REF_OFF(LISTENER_STATE_NOTIFY, tasks_cpudelay_usec), GY_REF_OFF(LISTENER_STATE_NOTIFY, tasks_blkiodelay_usec),
	GY_REF_OFF(LISTENER_STATE_NOTIFY, tasks_user_cpu), GY_REF_OFF(LISTENER_STATE_NOTIFY, tasks_sys_cpu), GY_REF_OFF(LISTENER_STATE_NOTIFY, tasks_rss_mb),
	GY_REF_OFF(LISTENER_STATE_NOTIFY, ntasks_issue), GY_REF_OFF(LISTENER_STATE_NOTIFY, is_http_svc), GY_REF_OFF(LISTENER_STATE_NOTIFY, curr_state),
	GY_REF_OFF(LISTENER_STATE_NOTIFY, curr_issue), GY_REF_OFF(LISTENER_STATE_NOTIFY, issue_bit_hist), GY_REF_OFF(LISTENER_STATE_NOTIFY, high_resp_bit_hist),
	GY_REF_OFF(LISTENER_STATE_NOTIFY, last_issue_subsrc), GY_REF_OFF(LISTENER_STATE_NOTIFY, query_flags), GY_REF_OFF(LISTENER_STATE_NOTIFY, issue_string_len),
	GY_REF_OFF(LISTENER_STATE_NOTIFY, padding_len),
	GY_REF_OFF(LISTENER_DAY_STATS, glob_id), GY_REF_OFF(LISTENER_DAY_STATS, tcount_5d), GY_REF_OFF(LISTENER_DAY_STATS, tsum_5d),
	GY_REF_OFF(LISTENER_DAY_STATS, p95_5d_respms), GY_REF_OFF(LISTENER_DAY_STATS, p25_5d_respms), GY_REF_OFF(LISTENER_DAY_STATS, p95_qps),
	GY_REF_OFF(LISTENER_DAY_STATS, p25_qps), GY_REF_OFF(LISTENER_DAY_STATS, p95_nactive), GY_REF_OFF(LISTENER_DAY_STATS, p25_nactive),
	GY_REF_OFF(ACTIVE_CONN_STATS, listener_glob_id), GY_REF_OFF(ACTIVE_CONN_STATS, cli_aggr_task_id), GY_REF_OFF(ACTIVE_CONN_STATS, ser_comm),
	GY_REF_OFF(ACTIVE_CONN_STATS, cli_comm), GY_REF_OFF(ACTIVE_CONN_STATS, remote_machine_id), GY_REF_OFF(ACTIVE_CONN_STATS, remote_madhava_id),
	GY_REF_OFF(ACTIVE_CONN_STATS, bytes_sent), GY_REF_OFF(ACTIVE_CONN_STATS, bytes_received), GY_REF_OFF(ACTIVE_CONN_STATS, cli_delay_msec),
	GY_REF_OFF(ACTIVE_CONN_STATS, ser_delay_msec), GY_REF_OFF(ACTIVE_CONN_STATS, max_rtt_msec), GY_REF_OFF(ACTIVE_CONN_STATS, active_conns),
};
// the three 1-bit flags behind active_conns_ (cli_listener_proc_, is_remote_listen_, is_remote_cli_): the byte offset and bit values as
// the reference's compiler lays them out, read back from a zeroed record with exactly one flag set
uint32_t ref_active_conn_flag(int which)
{
	comm::ACTIVE_CONN_STATS a;
	std::memset((void *)&a, 0, sizeof(a));
	if (which == 0) a.cli_listener_proc_ = true;
	else if (which == 1) a.is_remote_listen_ = true;
	else a.is_remote_cli_ = true;
	const uint8_t *b = (const uint8_t *)&a;
	for (size_t i = 0; i < sizeof(a); ++i)
		if (b[i]) return (uint32_t)((i << 8) | b[i]);
	return 0;
}
int ref_comm_nfields(void) { return (int)(sizeof(g_comm_offs) / sizeof(g_comm_offs[0])); }
const char *ref_comm_field_name(int i) { return g_comm_offs[i].name; }
size_t ref_comm_field_offset(int i) { return g_comm_offs[i].off; }
size_t ref_comm_sizeof(int which)
{
	switch (which) {
	case 0: return sizeof(comm::COMM_HEADER);
	case 1: return sizeof(comm::EVENT_NOTIFY);
	case 2: return sizeof(comm::TCP_CONN_NOTIFY);
	case 3: return sizeof(comm::LISTENER_STATE_NOTIFY);
	case 4: return sizeof(comm::LISTENER_DAY_STATS);
	case 5: return sizeof(GY_MACHINE_ID);
	case 6: return sizeof(comm::ACTIVE_CONN_STATS);
	default: return 0;
	}
}
// 0 PM_HDR_MAGIC, 1 COMM_EVENT_NOTIFY, 2 COMM_MIN_TYPE, 3 COMM_MAX_TYPE, 4 MAX_COMM_DATA_SZ, 5 NOTIFY_TCP_CONN, 6 NOTIFY_LISTENER_STATE,
// 7 TCP_CONN_NOTIFY::MAX_NUM_CONNS, 8 LISTENER_STATE_NOTIFY::MAX_NUM_LISTENERS, 9 LISTENER_DAY_STATS::MAX_NUM_LISTENERS, 10 LISTEN_FLAG_DELETE
uint64_t ref_comm_const(int which)
{
	switch (which) {
	case 0: return (uint64_t)comm::COMM_HEADER::PM_HDR_MAGIC;
	case 1: return (uint64_t)comm::COMM_EVENT_NOTIFY;
	case 2: return (uint64_t)comm::COMM_MIN_TYPE;
	case 3: return (uint64_t)comm::COMM_MAX_TYPE;
	case 4: return (uint64_t)comm::MAX_COMM_DATA_SZ;
	case 5: return (uint64_t)comm::NOTIFY_TCP_CONN;
	case 6: return (uint64_t)comm::NOTIFY_LISTENER_STATE;
	case 7: return (uint64_t)comm::TCP_CONN_NOTIFY::MAX_NUM_CONNS;
	case 8: return (uint64_t)comm::LISTENER_STATE_NOTIFY::MAX_NUM_LISTENERS;
	case 9: return (uint64_t)comm::LISTENER_DAY_STATS::MAX_NUM_LISTENERS;
	case 10: return (uint64_t)comm::LISTEN_FLAG_DELETE;
	default: return ~0ull;
	}
}
// msg points at a COMM_HEADER (8-byte aligned, writable: the record validators NUL-terminate the strings in place)
int ref_comm_hdr_validate(const uint8_t *msg, uint32_t req_magic)
{
	return ((const comm::COMM_HEADER *)msg)->validate(msg, (comm::COMM_HEADER::HDR_MAGIC_E)req_magic) ? 1 : 0;
}
int ref_tcp_conn_validate(uint8_t *msg)
{
	return comm::TCP_CONN_NOTIFY::validate((const comm::COMM_HEADER *)msg, (const comm::EVENT_NOTIFY *)(msg + sizeof(comm::COMM_HEADER))) ? 1 : 0;
}
int ref_listener_state_validate(uint8_t *msg)
{
	return comm::LISTENER_STATE_NOTIFY::validate((const comm::COMM_HEADER *)msg, (const comm::EVENT_NOTIFY *)(msg + sizeof(comm::COMM_HEADER))) ? 1 : 0;
}
// comm::MS_CLUSTER_STATE::STATE_ONE (common/gy_comm_proto.h:3183-3213): member order as 11 consecutive uint32_t + add_stats (the shyama-side
// fan-in of the per-madhava cluster states)
size_t ref_state_one_sizeof(void) { return sizeof(comm::MS_CLUSTER_STATE::STATE_ONE); }
void ref_state_one_add(uint32_t dst[11], const uint32_t src[11])
{
	comm::MS_CLUSTER_STATE::STATE_ONE a, b;
	static_assert(offsetof(comm::MS_CLUSTER_STATE::STATE_ONE, nmem_issue_) == 40, "11 consecutive uint32_t");
	memcpy(&a, dst, 44);
	memcpy(&b, src, 44);
	a.add_stats(b);
	memcpy(dst, &a, 44);
}
// member-wise view, to pin the ORDER of the 11 counters
void ref_state_one_fields(const uint32_t in[11], uint32_t out[11])
{
	comm::MS_CLUSTER_STATE::STATE_ONE a;
	memcpy(&a, in, 44);
	const uint32_t v[11] = {a.nhosts_, a.ntasks_issue_, a.ntaskissue_hosts_, a.ntasks_, a.nsvc_issue_, a.nsvcissue_hosts_, a.nsvc_, a.total_qps_,
				a.svc_net_mb_, a.ncpu_issue_, a.nmem_issue_};
	memcpy(out, v, 44);
}
uint32_t ref_tcp_conn_elem_size(const uint8_t *rec) { return (uint32_t)((const comm::TCP_CONN_NOTIFY *)rec)->get_elem_size(); }
uint32_t ref_listener_state_elem_size(const uint8_t *rec) { return (uint32_t)((const comm::LISTENER_STATE_NOTIFY *)rec)->get_elem_size(); }
}  // extern "C"

// gys_svcquery_host.hpp -- host side of the filtered multi-host listener-state query (kernels: gys_svcquery.hpp).  Included once by
// gys_engine.hip behind gys_json.hpp (uses gys_ctx and the JSON writer).
//
//   gys_query_svcstate_scan       MCONN_HANDLER::web_curr_listener_state with QUERY_OPTIONS{criteria_, maxrecs_, sortcolarr_[0], sortdir_[0],
//                                 is_multihost_}  server/gy_mnodehandle.cc:4650-4900, common/gy_query_common.h:24-140
//   gys_json_svcstate_multihost   the same as JSON: {"madid", "svcstate":[{parid, host, madid, cluster, <json_db_svcstate_arr columns>}...]}
//                                 (multi-host column list: QUERY_OPTIONS::get_all_column_list common/gy_query_common.h:418-437)
//   gys_query_svcstate_aggr       the aggregation operators AGGR_OPER_E (common/gy_json_field_maps.h:114-129) over the matching records
#pragma once

namespace {

template <typename T>
int q_grow(T **p, uint64_t *cap, uint64_t need)
{
	if (*cap >= need && *p) return GYS_OK;
	if (*p) HIPCHK(hipFree(*p));
	*p = nullptr;
	*cap = 0;
	const uint64_t n = std::max<uint64_t>(need, 1);
	HIPCHK(hipMalloc((void **)p, n * sizeof(T)));
	*cap = n;
	return GYS_OK;
}

// q_misc layout (u32 words): [0] candidate cursor, [1] want, [2] out count, [4..5] prefix (u64), [8 .. 8 + 2048) digit histogram
constexpr uint32_t QM_CURSOR = 0, QM_WANT = 1, QM_OUT = 2, QM_PREFIX = 4, QM_HIST = 8, QM_WORDS = 8 + GYS_SVCQ_RADIX;

// validates the caller's filter and fills the kernel-side copy (terms converted to the column's own type, as the reference converts a
// criterion to the field's type before comparing: match_num_criterian<int>, `pnumarray_[i].get<Num>()` common/gy_query_criteria.h:1243-1283)
template <typename P>
int q_fill_filter(gys_ctx *c, const gys_svc_filter *f, P &p)
{
	p.svc_state = c->svc_state;
	p.svc_host = c->svc_host;
	p.svc_gid = c->svc_gid;
	p.nsvc = c->nsvc;
	p.epoch = c->epoch;
	p.host_mask = nullptr;
	p.slot_list = nullptr;
	p.nitems = c->nsvc;
	p.set_values = nullptr;
	p.nterms = 0;
	p.ngroups = 0;
	p.top_oper = 0;
	memset(p.group_oper, 0, sizeof(p.group_oper));
	if (!f) return GYS_OK;
	if (f->nterms > GYS_SVCQ_MAX_TERMS || (f->nterms && !f->terms)) {
		set_err("filter: at most %u terms", GYS_SVCQ_MAX_TERMS);
		return GYS_ERR_INVAL;
	}
	std::vector<int32_t> setv;
	for (uint32_t i = 0; i < f->nterms; ++i) {
		const gys_svc_term &t = f->terms[i];
		const bool in = t.comp == GYS_COMP_IN || t.comp == GYS_COMP_NOTIN;
		if (t.col >= GYS_SVC_NCOLS || t.group >= GYS_SVCQ_MAX_GROUPS || !(t.comp <= GYS_COMP_BIT3 || in)) {
			set_err("filter term %u: column %u / comparator %u / group %u out of range", i, t.col, t.comp, t.group);
			return GYS_ERR_INVAL;
		}
		if (in && ((uint64_t)t.set_first + t.nvalues > f->nset_values || (t.nvalues && !f->set_values))) {
			set_err("filter term %u: value set outside set_values", i);
			return GYS_ERR_INVAL;
		}
		SvcTerm &d = p.terms[i];
		d.col = t.col;
		d.comp = t.comp;
		d.group = t.group;
		d.pad = 0;
		d.nvalues = in ? t.nvalues : 0;
		d.set_first = (uint32_t)setv.size();
		// the column's own type: int16_t for `issue` (server/gy_mfields.h:1435), int for the others
		auto conv = [&](int64_t v) -> int32_t { return t.col == GYS_SVC_COL_ISSUE ? (int32_t)(int16_t)v : (int32_t)v; };
		d.value = conv(t.value);
		for (uint32_t k = 0; k < d.nvalues; ++k) setv.push_back(conv(f->set_values[t.set_first + k]));
		p.ngroups = std::max<uint32_t>(p.ngroups, (uint32_t)t.group + 1u);
	}
	p.nterms = f->nterms;
	for (uint32_t g = 0; g < GYS_SVCQ_MAX_GROUPS; ++g) p.group_oper[g] = f->group_oper[g] ? 1 : 0;
	p.top_oper = f->top_oper ? 1u : 0u;
	if (!setv.empty()) {
		int rc = q_grow(&c->q_set, &c->q_set_cap, setv.size());
		if (rc) return rc;
		HIPCHK(hipMemcpyAsync(c->q_set, setv.data(), setv.size() * 4, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream)); // (setv is a local)
		p.set_values = c->q_set;
	}
	if (f->nmachine_ids || f->nclusters) { // the query names its hosts (is_multihost_ with host criteria: the walk over partha_tbl_ :4790-4860) and / or clusters
		if ((f->nmachine_ids && !f->machine_ids) || (f->nclusters && !f->clusters)) return GYS_ERR_INVAL;
		const uint64_t words = ((uint64_t)c->hosts.size() + 31) / 32 + 1;
		std::vector<uint32_t> mask(words, f->nmachine_ids ? 0u : ~0u);
		for (uint32_t i = 0; i < f->nmachine_ids; ++i) {
			uint32_t h;
			if (lookup_host(c, f->machine_ids + (size_t)i * 16, &h) == GYS_OK) mask[h >> 5] |= 1u << (h & 31u); // (an unknown host matches nothing)
		}
		if (f->nclusters) {
			std::vector<uint8_t> want(c->cluster_names.size(), 0);
			for (uint32_t i = 0; i < f->nclusters; ++i) {
				auto it = f->clusters[i] ? c->cluster_map.find(f->clusters[i]) : c->cluster_map.end();
				if (it != c->cluster_map.end()) want[it->second] = 1; // (an unknown cluster matches nothing)
			}
			for (uint32_t h = 0; h < c->hosts.size(); ++h)
				if (!want[c->host_cluster_h[h]]) mask[h >> 5] &= ~(1u << (h & 31u));
		}
		int rc = q_grow(&c->q_host_mask, &c->q_mask_cap, words);
		if (rc) return rc;
		HIPCHK(hipMemcpyAsync(c->q_host_mask, mask.data(), words * 4, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		p.host_mask = c->q_host_mask;
	}
	if (f->nsvcids) { // the query names its listeners: only their slots are visited
		if (!f->svcids) return GYS_ERR_INVAL;
		std::vector<uint32_t> slots;
		for (uint32_t i = 0; i < f->nsvcids; ++i) {
			auto it = c->gid_map_h.find(f->svcids[i]);
			if (it != c->gid_map_h.end()) slots.push_back(it->second);
		}
		std::sort(slots.begin(), slots.end());
		slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
		int rc = q_grow(&c->q_slot_list, &c->q_slist_cap, slots.size());
		if (rc) return rc;
		if (!slots.empty()) {
			HIPCHK(hipMemcpyAsync(c->q_slot_list, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, c->stream));
			HIPCHK(hipStreamSynchronize(c->stream));
		}
		p.slot_list = c->q_slot_list;
		p.nitems = (uint32_t)slots.size();
	}
	return GYS_OK;
}

// runs the scan; rows (ordered) and keys land in host vectors
int q_scan(gys_ctx *c, const gys_svc_filter *f, int sort_col, int sort_desc, uint32_t maxrecs, std::vector<gys_svc_row> &rows, uint64_t *nmatched)
{
	rows.clear();
	if (nmatched) *nmatched = 0;
	if (sort_col >= (int)GYS_SVC_NCOLS) {
		set_err("sort column %d out of range", sort_col);
		return GYS_ERR_INVAL;
	}
	if (!c->nsvc || !maxrecs) return GYS_OK;
	int rc;
	static_assert(sizeof(gys_svc_row) == 96, "row layout");
	if ((rc = q_grow(&c->q_cand_key, &c->q_cand_cap, c->nsvc)) != GYS_OK) return rc;
	if ((rc = q_grow(&c->q_cand_slot, &c->q_slot_cap, c->nsvc)) != GYS_OK) return rc;
	if (!c->q_misc) HIPCHK(hipMalloc((void **)&c->q_misc, QM_WORDS * 4));
	HIPCHK(hipMemsetAsync(c->q_misc, 0, QM_WORDS * 4, c->stream));
	const uint32_t k = (uint32_t)std::min<uint64_t>(maxrecs, c->nsvc);
	if ((rc = q_grow(&c->q_out_rows, &c->q_out_cap, (uint64_t)k * 96)) != GYS_OK) return rc;
	if ((rc = q_grow(&c->q_out_keys, &c->q_okeys_cap, k)) != GYS_OK) return rc;
	ProfScope ps(c, "svc_filter");
	SvcFilterP p{};
	if ((rc = q_fill_filter(c, f, p)) != GYS_OK) return rc;
	p.sort_col = sort_col;
	p.sort_desc = sort_desc ? 1u : 0u;
	p.cand_key = c->q_cand_key;
	p.cand_slot = c->q_cand_slot;
	p.cursor = c->q_misc + QM_CURSOR;
	const uint32_t per_wg = GYS_SVCQ_THREADS * GYS_SVCQ_PER_THREAD;
	hipLaunchKernelGGL(k_svc_filter, dim3(std::max(1u, (p.nitems + per_wg - 1) / per_wg)), dim3(GYS_SVCQ_THREADS), 0, c->stream, p);
	uint32_t ncand = 0;
	HIPCHK(hipMemcpyAsync(&ncand, c->q_misc + QM_CURSOR, 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (nmatched) *nmatched = ncand;
	if (!ncand) return GYS_OK;
	const uint32_t ntake = std::min(ncand, k);
	const bool select = ncand > k;
	if (select) { // exact top-k: the k-th largest key, 11 bits per round, every round's bin chosen on the device
		SvcSelectP sp{};
		sp.cand_key = c->q_cand_key;
		sp.ncand = c->q_misc + QM_CURSOR;
		sp.hist = c->q_misc + QM_HIST;
		sp.prefix = (unsigned long long *)(c->q_misc + QM_PREFIX);
		sp.want = c->q_misc + QM_WANT;
		HIPCHK(hipMemcpyAsync(c->q_misc + QM_WANT, &k, 4, hipMemcpyHostToDevice, c->stream));
		const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)ncand + 256u * 16u - 1) / (256u * 16u), (uint64_t)c->ncu * 4);
		static const uint32_t shifts[GYS_SVCQ_ROUNDS] = {53, 42, 31, 20, 9, 0}, widths[GYS_SVCQ_ROUNDS] = {11, 11, 11, 11, 11, 9};
		for (uint32_t r = 0; r < GYS_SVCQ_ROUNDS; ++r) {
			sp.shift = shifts[r];
			sp.bits = widths[r];
			hipLaunchKernelGGL(k_svc_hist, dim3(std::max(1u, grid)), dim3(256), 0, c->stream, sp);
			hipLaunchKernelGGL(k_svc_pick, dim3(1), dim3(256), 0, c->stream, sp);
		}
	}
	SvcGatherP gp{};
	gp.svc_state = c->svc_state;
	gp.cand_key = c->q_cand_key;
	gp.cand_slot = c->q_cand_slot;
	gp.ncand = c->q_misc + QM_CURSOR;
	gp.threshold = select ? (const unsigned long long *)(c->q_misc + QM_PREFIX) : nullptr;
	gp.maxout = k;
	gp.out_count = c->q_misc + QM_OUT;
	gp.out_rows = c->q_out_rows;
	gp.out_keys = c->q_out_keys;
	const uint32_t ggrid = (uint32_t)std::min<uint64_t>(((uint64_t)ncand + 255u) / 256u, (uint64_t)c->ncu * 8);
	hipLaunchKernelGGL(k_svc_gather, dim3(std::max(1u, ggrid)), dim3(256), 0, c->stream, gp);
	HIPCHK(hipGetLastError());
	uint32_t nout = 0;
	HIPCHK(hipMemcpyAsync(&nout, c->q_misc + QM_OUT, 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (nout != ntake) {
		set_err("svcstate scan: selected %u records, expected %u", nout, ntake);
		return GYS_ERR_INTERNAL;
	}
	std::vector<gys_svc_row> raw(nout);
	std::vector<unsigned long long> keys(nout);
	HIPCHK(hipMemcpyAsync(raw.data(), c->q_out_rows, (uint64_t)nout * 96, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(keys.data(), c->q_out_keys, (uint64_t)nout * 8, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	std::vector<uint32_t> order(nout);
	for (uint32_t i = 0; i < nout; ++i) order[i] = i;
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] > keys[b]; }); // (keys are unique)
	rows.resize(nout);
	for (uint32_t i = 0; i < nout; ++i) rows[i] = raw[order[i]];
	return GYS_OK;
}

} // namespace

extern "C" {

int gys_query_svcstate_scan(gys_ctx *c, const gys_svc_filter *f, int sort_col, int sort_desc, uint32_t maxrecs, gys_svc_row *out, uint32_t *nout,
			    uint64_t *nmatched)
try {
	GYS_ENTER(c);
	if (!c || !nout || (maxrecs && !out)) return GYS_ERR_INVAL;
	*nout = 0;
	std::vector<gys_svc_row> rows;
	const int rc = q_scan(c, f, sort_col, sort_desc, maxrecs, rows, nmatched);
	if (rc) return rc;
	if (!rows.empty()) memcpy(out, rows.data(), rows.size() * sizeof(gys_svc_row));
	*nout = (uint32_t)rows.size();
	return GYS_OK;
} GYS_CATCH_ALL

int gys_json_svcstate_multihost(gys_ctx *c, const gys_svc_filter *f, int sort_col, int sort_desc, uint32_t maxrecs, const char *madhava_id16,
				const char *timestr, char *buf, size_t buflen, size_t *needed)
try {
	GYS_ENTER(c);
	if (!c) return GYS_ERR_INVAL;
	std::vector<gys_svc_row> rows;
	const int rc = q_scan(c, f, sort_col, sort_desc, maxrecs, rows, nullptr);
	if (rc) return rc;
	const char *mad = madhava_id16 ? madhava_id16 : "";
	JsonBuf j;
	j.s += '{';
	j.kstr("madid", mad, 16);
	j.arr_open("svcstate");
	for (const gys_svc_row &r : rows) svcstate_object(c, j, r.rec, r.slot, r.host_slot, true, mad, timestr);
	j.arr_close();
	j.s += '}';
	return json_out(j, buf, buflen, needed);
} GYS_CATCH_ALL

int gys_query_svcstate_aggr(gys_ctx *c, const gys_svc_filter *f, int group_by, const uint8_t *cols, uint32_t ncols, gys_svc_aggr_row *out, uint32_t maxrows,
			    uint32_t *nrows)
try {
	GYS_ENTER(c);
	if (!c || !nrows || (maxrows && !out) || (ncols && !cols) || ncols > GYS_SVCQ_MAX_AGGR || group_by < 0 || group_by > 2) return GYS_ERR_INVAL;
	*nrows = 0;
	for (uint32_t a = 0; a < ncols; ++a)
		if (cols[a] >= GYS_SVC_NCOLS) return GYS_ERR_INVAL;
	const uint32_t ngroups = group_by == 0 ? 1u : group_by == 1 ? (uint32_t)c->hosts.size() : (uint32_t)c->cluster_names.size();
	if (!c->nsvc || !ngroups) return GYS_OK;
	const uint32_t nc = std::max(ncols, 1u);
	int rc;
	if ((rc = q_grow(&c->q_acc, &c->q_acc_cap, (uint64_t)ngroups * nc * 3)) != GYS_OK) return rc;
	if ((rc = q_grow(&c->q_cnt, &c->q_cnt_cap, ngroups)) != GYS_OK) return rc;
	std::vector<long long> init((size_t)ngroups * nc * 3);
	for (size_t i = 0; i < init.size(); i += 3) {
		init[i] = 0;
		init[i + 1] = std::numeric_limits<long long>::max();
		init[i + 2] = std::numeric_limits<long long>::min();
	}
	HIPCHK(hipMemcpyAsync(c->q_acc, init.data(), init.size() * 8, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemsetAsync(c->q_cnt, 0, (uint64_t)ngroups * 8, c->stream));
	ProfScope ps(c, "svc_aggr");
	SvcAggrP p{};
	if ((rc = q_fill_filter(c, f, p)) != GYS_OK) return rc;
	p.group_by = (uint32_t)group_by;
	p.host_cluster = c->host_cluster;
	p.ncols = ncols;
	for (uint32_t a = 0; a < ncols; ++a) p.cols[a] = cols[a];
	p.acc = c->q_acc;
	p.count = c->q_cnt;
	const uint32_t per_wg = GYS_SVCQ_THREADS * GYS_SVCQ_PER_THREAD;
	hipLaunchKernelGGL(k_svc_aggr, dim3(std::max(1u, (p.nitems + per_wg - 1) / per_wg)), dim3(GYS_SVCQ_THREADS), 0, c->stream, p);
	HIPCHK(hipGetLastError());
	std::vector<unsigned long long> cnt(ngroups);
	HIPCHK(hipMemcpyAsync(init.data(), c->q_acc, init.size() * 8, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(cnt.data(), c->q_cnt, (uint64_t)ngroups * 8, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	uint32_t n = 0;
	for (uint32_t g = 0; g < ngroups; ++g) {
		if (!cnt[g]) continue; // a group without a matching record has no row (SQL GROUP BY)
		if (n < maxrows) {
			gys_svc_aggr_row &r = out[n];
			memset(&r, 0, sizeof(r));
			r.group = g;
			r.ncols = ncols;
			r.count = cnt[g];
			for (uint32_t a = 0; a < ncols; ++a) {
				const long long *v = &init[((size_t)g * nc + a) * 3];
				r.sum[a] = v[0];
				r.min[a] = v[1];
				r.max[a] = v[2];
			}
		}
		++n;
	}
	*nrows = n; // (more than maxrows: the caller sees how many there are)
	return GYS_OK;
} GYS_CATCH_ALL

// CRITERIA_SET::match_criteria (common/gy_query_criteria.h:1535-1605, :1806-1900) on one host's LISTEN_SUMM_STATS<int> row (the 13 numeric
// columns of json_db_svcsumm_arr: SvcSummFields::get_num_field server/gy_mfields.h:768-790) -- the host-side twin of svc_filter_match
static bool summ_filter_match(const gys_svc_filter *f, const int32_t row[GYS_SUMM_NCOLS])
{
	if (!f || f->nterms == 0) return true;
	uint32_t pass = 0, fail = 0, seen = 0;
	for (uint32_t i = 0; i < f->nterms; ++i) {
		const gys_svc_term &t = f->terms[i];
		const uint32_t bit = 1u << t.group;
		seen |= bit;
		const int32_t v = row[t.col], crit = (int32_t)t.value;
		bool m = false;
		switch (t.comp) {
		case GYS_COMP_EQ: m = v == crit; break;
		case GYS_COMP_NEQ: m = v != crit; break;
		case GYS_COMP_LT: m = v < crit; break;
		case GYS_COMP_LE: m = v <= crit; break;
		case GYS_COMP_GT: m = v > crit; break;
		case GYS_COMP_GE: m = v >= crit; break;
		case GYS_COMP_BIT2: m = (v & 3) == 3; break;
		case GYS_COMP_BIT3: m = (v & 7) == 7; break;
		default: { // IN / NOTIN
			bool found = false;
			for (uint32_t k = 0; k < t.nvalues; ++k) found = found || (int32_t)f->set_values[t.set_first + k] == v;
			m = t.comp == GYS_COMP_IN ? found : !found;
		}
		}
		if (f->group_oper[t.group]) {
			if (m) pass |= bit;
		} else if (!m)
			fail |= bit;
	}
	uint32_t gpass = 0;
	for (uint32_t g = 0; g < GYS_SVC_MAX_GROUPS; ++g) {
		const uint32_t bit = 1u << g;
		if (!(seen & bit)) continue;
		if (f->group_oper[g] ? (pass & bit) != 0 : (fail & bit) == 0) gpass |= bit;
	}
	return f->top_oper ? gpass != 0 : gpass == seen;
}

int gys_json_svcsumm_multihost(gys_ctx *c, const gys_svc_filter *f, int sort_col, int sort_desc, uint32_t maxrecs, const char *madhava_id16,
			       const char *timestr, char *buf, size_t buflen, size_t *needed)
try {
	GYS_ENTER(c);
	if (!c || sort_col >= (int)GYS_SUMM_NCOLS) return GYS_ERR_INVAL;
	if (f) {
		if (f->nterms > GYS_SVC_MAX_TERMS || (f->nterms && !f->terms)) return GYS_ERR_INVAL;
		for (uint32_t i = 0; i < f->nterms; ++i) {
			const gys_svc_term &t = f->terms[i];
			const bool in = t.comp == GYS_COMP_IN || t.comp == GYS_COMP_NOTIN;
			if (t.col >= GYS_SUMM_NCOLS || t.group >= GYS_SVC_MAX_GROUPS || !(t.comp <= GYS_COMP_BIT3 || in) ||
			    (in && ((uint64_t)t.set_first + t.nvalues > f->nset_values || (t.nvalues && !f->set_values)))) {
				set_err("summary filter term %u out of range", i);
				return GYS_ERR_INVAL;
			}
		}
	}
	const uint32_t nh = (uint32_t)c->hosts.size();
	std::vector<int32_t> summ((size_t)nh * 16);
	if (nh) {
		HIPCHK(hipMemcpyAsync(summ.data(), c->host_summ_last, (size_t)nh * 64, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	// the hosts of the query (machine ids and / or clusters), as the device query builds its host mask
	std::vector<uint8_t> in_host(nh, (f && f->nmachine_ids) ? 0 : 1);
	if (f && f->nmachine_ids) {
		if (!f->machine_ids) return GYS_ERR_INVAL;
		for (uint32_t i = 0; i < f->nmachine_ids; ++i) {
			uint32_t h;
			if (lookup_host(c, f->machine_ids + (size_t)i * 16, &h) == GYS_OK) in_host[h] = 1;
		}
	}
	if (f && f->nclusters) {
		if (!f->clusters) return GYS_ERR_INVAL;
		std::vector<uint8_t> want(c->cluster_names.size(), 0);
		for (uint32_t i = 0; i < f->nclusters; ++i) {
			auto it = f->clusters[i] ? c->cluster_map.find(f->clusters[i]) : c->cluster_map.end();
			if (it != c->cluster_map.end()) want[it->second] = 1;
		}
		for (uint32_t h = 0; h < nh; ++h)
			if (!want[c->host_cluster_h[h]]) in_host[h] = 0;
	}
	std::vector<uint32_t> rows;
	for (uint32_t h = 0; h < nh; ++h) {
		const int32_t *r = &summ[(size_t)h * 16];
		if (!in_host[h] || r[GYS_SUMM_COL_NSVC] == 0) continue; // no listener state of this host in the last window: its data is not recent (:1656)
		if (summ_filter_match(f, r)) rows.push_back(h);
	}
	if (sort_col >= 0)
		std::stable_sort(rows.begin(), rows.end(), [&](uint32_t a, uint32_t b) {
			const int32_t x = summ[(size_t)a * 16 + sort_col], y = summ[(size_t)b * 16 + sort_col];
			return sort_desc ? x > y : x < y;
		});
	if (rows.size() > maxrecs) rows.resize(maxrecs);
	const char *mad = madhava_id16 ? madhava_id16 : "";
	JsonBuf j;
	j.s += '{';
	j.kstr("madid", mad, 16);
	j.arr_open("summstats");
	for (uint32_t h : rows) {
		const int32_t *r = &summ[(size_t)h * 16];
		j.obj_open();
		j.kstr("parid", machid_string(c->hosts[h]));
		j.kstr("host", c->host_names[h]);
		j.kstr("madid", mad, 16);
		j.kstr("cluster", c->cluster_names[c->host_cluster_h[h]]);
		j.kstr("time", timestr ? timestr : "", 64);
		static const char *names[GYS_SUMM_NCOLS] = {"nidle", "ngood", "nok", "nbad", "nsevere", "ndown", "totqps", "totaconn", "totkbin", "totkbout", "totsererr", "nsvc", "nactive"};
		for (int k = 0; k < (int)GYS_SUMM_NCOLS; ++k) j.ki(names[k], r[k]);
		j.obj_close();
	}
	j.arr_close();
	j.s += '}';
	return json_out(j, buf, buflen, needed);
} GYS_CATCH_ALL

} // extern "C"

namespace {
// CRITERION_ONE::match_str_criterian (common/gy_query_criteria.h:1335-1383) for one comparator and its pattern(s), compiled once
struct StrCriterion {
	int base = 0;
	bool neg = false;
	gysre::Regex re; // linear-time matcher (gys_regex.hpp): the reference's RE2::PartialMatch never backtracks either
	std::vector<std::string> pats;
	int init(int comp, const char *const *patterns, uint32_t npatterns, const char *who)
	{
		if (!patterns || !npatterns) return GYS_ERR_INVAL;
		for (uint32_t i = 0; i < npatterns; ++i) {
			if (!patterns[i]) return GYS_ERR_INVAL;
			pats.emplace_back(patterns[i]);
		}
		neg = comp == GYS_COMP_NEQ || comp == GYS_COMP_NOTSUBSTR || comp == GYS_COMP_NOTLIKE || comp == GYS_COMP_NOTIN;
		base = comp == GYS_COMP_NEQ ? GYS_COMP_EQ : comp == GYS_COMP_NOTSUBSTR ? GYS_COMP_SUBSTR : comp == GYS_COMP_NOTLIKE ? GYS_COMP_LIKE :
		       comp == GYS_COMP_NOTIN ? GYS_COMP_IN : comp;
		if (base != GYS_COMP_EQ && base != GYS_COMP_SUBSTR && base != GYS_COMP_LIKE && base != GYS_COMP_IN) {
			set_err("%s: comparator %d is not a string comparator", who, comp);
			return GYS_ERR_INVAL;
		}
		if (base == GYS_COMP_LIKE) {
			std::string why;
			if (!re.compile(pats[0], &why)) { // (RE2 fails the criterion the same way: "Invalid regex", common/gy_query_criteria.h:347-352)
				set_err("%s: invalid regular expression: %s", who, why.c_str());
				return GYS_ERR_INVAL;
			}
		}
		return GYS_OK;
	}
	bool match(const char *s, size_t len) const
	{
		bool hit = false;
		switch (base) {
		case GYS_COMP_EQ: hit = len == pats[0].size() && !memcmp(s, pats[0].data(), len); break;
		case GYS_COMP_SUBSTR: hit = pats[0].size() <= len && memmem(s, len, pats[0].data(), pats[0].size()) != nullptr; break;
		case GYS_COMP_LIKE: hit = re.search(s, len); break;
		default:
			for (size_t i = 0; i < pats.size() && !hit; ++i) hit = pats[i].size() == len && !memcmp(s, pats[i].data(), len);
			break;
		}
		return hit != neg;
	}
};
} // namespace

extern "C" {

// string criterion on the service name, resolved on the host into the service ids the device filter takes (gys_svc_filter.svcids)
int gys_svc_ids_by_name(gys_ctx *c, int comp, const char *const *patterns, uint32_t npatterns, uint64_t *out_ids, uint32_t cap, uint32_t *nout)
try {
	GYS_ENTER_NOFLUSH(c);
	if (!c || !nout || (!out_ids && cap)) return GYS_ERR_INVAL;
	StrCriterion sc;
	const int rc = sc.init(comp, patterns, npatterns, "gys_svc_ids_by_name");
	if (rc) return rc;
	// the registry holds up to 10^7 names and a regular-expression search costs ~1 us: slot ranges on the host's cores, results in slot order
	const uint32_t nsvc = c->nsvc;
	const uint32_t nthr = nsvc < (1u << 16) ? 1u : std::min<uint32_t>(32u, std::max(1u, std::thread::hardware_concurrency()));
	std::vector<std::vector<uint32_t>> hits(nthr);
	auto scan = [&](uint32_t t) {
		const uint32_t lo = (uint32_t)((uint64_t)nsvc * t / nthr), hi = (uint32_t)((uint64_t)nsvc * (t + 1) / nthr);
		for (uint32_t slot = lo; slot < hi; ++slot) {
			const char *name = c->svc_comm[slot].data();
			if (sc.match(name, strnlen(name, 16))) hits[t].push_back(slot);
		}
	};
	if (nthr == 1) {
		scan(0);
	} else {
		std::vector<std::thread> th;
		std::atomic<bool> failed{false};
		try {
			for (uint32_t t = 0; t < nthr; ++t)
				th.emplace_back([&, t] {
					try {
						scan(t);
					} catch (...) {
						failed = true;
					}
				});
		} catch (...) { // (a thread that could not be started: the ones that run are joined below before anything unwinds)
			failed = true;
		}
		for (auto &x : th) x.join();
		if (failed) {
			set_err("gys_svc_ids_by_name: could not scan the registry (threads / memory)");
			return GYS_ERR_NOMEM;
		}
	}
	uint32_t n = 0;
	for (uint32_t t = 0; t < nthr; ++t)
		for (uint32_t slot : hits[t]) {
			if (n < cap) out_ids[n] = c->svc_gid_h[slot];
			++n;
		}
	*nout = n;
	if (n > cap) {
		set_err("gys_svc_ids_by_name: %u services match, room for %u", n, cap);
		return GYS_ERR_NOMEM;
	}
	return GYS_OK;
} GYS_CATCH_ALL

// ... and on the host name (gys_set_host_name; PARTHA_INFO::hostname_, the "host" column): the machine ids for gys_svc_filter.machine_ids
int gys_machine_ids_by_hostname(gys_ctx *c, int comp, const char *const *patterns, uint32_t npatterns, uint8_t *out_ids16, uint32_t cap, uint32_t *nout)
try {
	GYS_ENTER_NOFLUSH(c);
	if (!c || !nout || (!out_ids16 && cap)) return GYS_ERR_INVAL;
	StrCriterion sc;
	const int rc = sc.init(comp, patterns, npatterns, "gys_machine_ids_by_hostname");
	if (rc) return rc;
	uint32_t n = 0;
	for (size_t h = 0; h < c->hosts.size(); ++h) {
		const std::string &name = c->host_names[h];
		if (sc.match(name.data(), name.size())) {
			if (n < cap) memcpy(out_ids16 + (size_t)n * 16, &c->hosts[h], 16);
			++n;
		}
	}
	*nout = n;
	if (n > cap) {
		set_err("gys_machine_ids_by_hostname: %u hosts match, room for %u", n, cap);
		return GYS_ERR_NOMEM;
	}
	return GYS_OK;
} GYS_CATCH_ALL

// AOPER_PERCENTILE over the records that pass the filter: the discrete percentile of one column -- the smallest value with at least
// pcts[i] (0 < p <= 1) of the matching records at or below it -- by the exact radix selection of the sorted scan (the candidate keys carry
// the column in their upper half; nothing is gathered or sorted)
int gys_query_svcstate_percentiles(gys_ctx *c, const gys_svc_filter *f, int col, const double *pcts, uint32_t npcts, int64_t *out, uint64_t *nmatched)
try {
	GYS_ENTER(c);
	if (!c || !pcts || !out || !npcts || col < 0 || col >= (int)GYS_SVC_NCOLS) return GYS_ERR_INVAL;
	for (uint32_t i = 0; i < npcts; ++i)
		if (!(pcts[i] > 0.0 && pcts[i] <= 1.0)) {
			set_err("gys_query_svcstate_percentiles: percentile %u is not in (0, 1]", i);
			return GYS_ERR_INVAL;
		}
	if (nmatched) *nmatched = 0;
	for (uint32_t i = 0; i < npcts; ++i) out[i] = 0;
	if (!c->nsvc) return GYS_OK;
	int rc;
	if ((rc = q_grow(&c->q_cand_key, &c->q_cand_cap, c->nsvc)) != GYS_OK) return rc;
	if ((rc = q_grow(&c->q_cand_slot, &c->q_slot_cap, c->nsvc)) != GYS_OK) return rc;
	if (!c->q_misc) HIPCHK(hipMalloc((void **)&c->q_misc, QM_WORDS * 4));
	HIPCHK(hipMemsetAsync(c->q_misc, 0, QM_WORDS * 4, c->stream));
	ProfScope ps(c, "svc_filter");
	SvcFilterP p{};
	if ((rc = q_fill_filter(c, f, p)) != GYS_OK) return rc;
	p.sort_col = col;
	p.sort_desc = 1u;
	p.cand_key = c->q_cand_key;
	p.cand_slot = c->q_cand_slot;
	p.cursor = c->q_misc + QM_CURSOR;
	const uint32_t per_wg = GYS_SVCQ_THREADS * GYS_SVCQ_PER_THREAD;
	hipLaunchKernelGGL(k_svc_filter, dim3(std::max(1u, (p.nitems + per_wg - 1) / per_wg)), dim3(GYS_SVCQ_THREADS), 0, c->stream, p);
	uint32_t ncand = 0;
	HIPCHK(hipMemcpyAsync(&ncand, c->q_misc + QM_CURSOR, 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (nmatched) *nmatched = ncand;
	if (!ncand) return GYS_OK;
	SvcSelectP sp{};
	sp.cand_key = c->q_cand_key;
	sp.ncand = c->q_misc + QM_CURSOR;
	sp.hist = c->q_misc + QM_HIST;
	sp.prefix = (unsigned long long *)(c->q_misc + QM_PREFIX);
	sp.want = c->q_misc + QM_WANT;
	const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)ncand + 256u * 16u - 1) / (256u * 16u), (uint64_t)c->ncu * 4);
	static const uint32_t shifts[GYS_SVCQ_ROUNDS] = {53, 42, 31, 20, 9, 0}, widths[GYS_SVCQ_ROUNDS] = {11, 11, 11, 11, 11, 9};
	std::vector<unsigned long long> keys(npcts);
	for (uint32_t i = 0; i < npcts; ++i) {
		// rank r = ceil(p N) from the bottom (1-based) = the (N - r + 1)-th largest key
		uint64_t r = (uint64_t)std::ceil(pcts[i] * (double)ncand);
		r = std::min<uint64_t>(std::max<uint64_t>(r, 1), ncand);
		const uint32_t k = (uint32_t)(ncand - r + 1);
		HIPCHK(hipMemsetAsync(c->q_misc + QM_PREFIX, 0, 8, c->stream));
		HIPCHK(hipMemcpyAsync(c->q_misc + QM_WANT, &k, 4, hipMemcpyHostToDevice, c->stream));
		for (uint32_t rd = 0; rd < GYS_SVCQ_ROUNDS; ++rd) {
			sp.shift = shifts[rd];
			sp.bits = widths[rd];
			hipLaunchKernelGGL(k_svc_hist, dim3(std::max(1u, grid)), dim3(256), 0, c->stream, sp);
			hipLaunchKernelGGL(k_svc_pick, dim3(1), dim3(256), 0, c->stream, sp);
		}
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(&keys[i], c->q_misc + QM_PREFIX, 8, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream)); // (k lives on this stack frame until its copy has been made)
	}
	for (uint32_t i = 0; i < npcts; ++i) out[i] = (int64_t)(int32_t)((uint32_t)(keys[i] >> 32) ^ 0x80000000u);
	return GYS_OK;
} GYS_CATCH_ALL

int gys_svc_aggr_value(const gys_svc_aggr_row *row, uint32_t col_index, int oper, double *out)
{
	if (!row || !out || col_index >= GYS_SVC_MAX_AGGR || (oper != GYS_AOPER_COUNT && col_index >= row->ncols)) return GYS_ERR_INVAL;
	switch (oper) { // AGGR_OPER_E common/gy_json_field_maps.h:114-129
	case GYS_AOPER_SUM: *out = (double)row->sum[col_index]; return GYS_OK;
	case GYS_AOPER_AVG: *out = row->count ? (double)row->sum[col_index] / (double)row->count : 0.0; return GYS_OK;
	case GYS_AOPER_MAX: *out = (double)row->max[col_index]; return GYS_OK;
	case GYS_AOPER_MIN: *out = (double)row->min[col_index]; return GYS_OK;
	case GYS_AOPER_COUNT: *out = (double)row->count; return GYS_OK;
	case GYS_AOPER_BOOL_OR: *out = row->max[col_index] != 0 || row->min[col_index] != 0 ? 1.0 : 0.0; return GYS_OK;   // some value is non-zero
	case GYS_AOPER_BOOL_AND: *out = (row->min[col_index] > 0 || row->max[col_index] < 0) ? 1.0 : 0.0; return GYS_OK; // no value is zero (exact for the >= 0 columns)
	default: return GYS_ERR_INVAL; // percentile: gys_query_svcstate_percentiles; first / last: no meaning on one snapshot (not built)
	}
}

} // extern "C"

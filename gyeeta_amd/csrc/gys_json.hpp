// gys_json.hpp -- query results in the reference's JSON shapes.  Host-side formatting only; the numbers come from the
// GPU-resident state through the same accessors as the binary queries.  Included once by gys_engine.hip (needs gys_ctx).
//
//   gys_json_svcsumm      MCONN_HANDLER::web_curr_listener_summ     server/gy_mnodehandle.cc:1628-1729  (single-host form)
//                         fields: SvcSummFields::print_field         server/gy_mfields.h:838-947, column order json_db_svcsumm_arr
//                         common/gy_json_field_maps.h:1396-1416
//   gys_json_svcstate     MCONN_HANDLER::web_curr_listener_state    server/gy_mnodehandle.cc:4650-4760
//                         fields: SvcStateFields::print_field        server/gy_mfields.h (class SvcStateFields), column order
//                         json_db_svcstate_arr common/gy_json_field_maps.h:1102-1135
//   gys_json_toplisteners MCONN_HANDLER::web_curr_top_listeners     server/gy_mnodehandle.cc:2706-3190 (single- and multi-host forms)
//   gys_json_clusterstate SHCONN_HANDLER::web_curr_clusterstate      server/gy_shnodehandle.cc:508-571
//                         fields: ClusterStateFields::print_field    server/gy_shfields.h, column order json_db_clusterstate_arr
//                         common/gy_json_field_maps.h:2162-2180
#pragma once

namespace {

struct JsonBuf {
	std::string s;
	bool first = true;
	void raw(const char *t) { s += t; }
	void key(const char *k)
	{
		if (!first) s += ',';
		first = false;
		s += '"';
		s += k;
		s += "\":";
	}
	void str(const char *p, size_t n)
	{
		s += '"';
		for (size_t i = 0; i < n && p[i]; ++i) {
			const unsigned char ch = (unsigned char)p[i];
			if (ch == '"' || ch == '\\') {
				s += '\\';
				s += (char)ch;
			} else if (ch < 0x20) {
				char t[8];
				snprintf(t, sizeof(t), "\\u%04x", ch);
				s += t;
			} else {
				s += (char)ch;
			}
		}
		s += '"';
	}
	void kstr(const char *k, const char *p, size_t n) { key(k); str(p, n); }
	void kstr(const char *k, const std::string &v) { key(k); str(v.data(), v.size()); }
	void ki(const char *k, long long v)
	{
		key(k);
		s += std::to_string(v);
	}
	void ku(const char *k, unsigned long long v)
	{
		key(k);
		s += std::to_string(v);
	}
	void kb(const char *k, bool v)
	{
		key(k);
		s += v ? "true" : "false";
	}
	void obj_open()
	{
		if (!first) s += ',';
		s += '{';
		first = true;
	}
	void obj_close()
	{
		s += '}';
		first = false;
	}
	void arr_open(const char *k)
	{
		key(k);
		s += '[';
		first = true;
	}
	void arr_close()
	{
		s += ']';
		first = false;
	}
};

int json_out(const JsonBuf &j, char *buf, size_t buflen, size_t *needed)
{
	if (needed) *needed = j.s.size();
	if (!buf || buflen < j.s.size() + 1) {
		set_err("JSON needs %zu bytes", j.s.size() + 1);
		return GYS_ERR_NOMEM;
	}
	memcpy(buf, j.s.data(), j.s.size());
	buf[j.s.size()] = 0;
	return GYS_OK;
}

std::string machid_string(const MachId &m)
{
	char t[40];
	snprintf(t, sizeof(t), "%016llx%016llx", (unsigned long long)m.first, (unsigned long long)m.second); // GY_MACHINE_ID::get_string gy_sys_hardware.h:166
	return t;
}

const char *state_string(uint32_t st)
{
	// state_to_stringlen common/gy_json_field_maps.h:282-295
	static const char *names[] = {"Idle", "Good", "OK", "Bad", "Severe", "Down"};
	return st < 6 ? names[st] : "Unknown";
}

void hostinfo_object(gys_ctx *c, JsonBuf &j, uint32_t host, const char *madid)
{
	j.key("hostinfo");
	j.s += '{';
	j.first = true;
	j.kstr("parid", machid_string(c->hosts[host]));
	j.kstr("host", c->host_names[host]);
	j.kstr("madid", madid, 16);
	j.kstr("cluster", c->cluster_names[c->host_cluster_h[host]]);
	j.obj_close();
}

// one svcstate object in the reference's column order (json_db_svcstate_arr common/gy_json_field_maps.h:1102-1135, values as
// SvcStateFields::print_field writes them server/gy_mfields.h:1560-1700); multi = the four host columns first (get_all_column_list)
void svcstate_object(gys_ctx *c, JsonBuf &j, const uint8_t *r, uint32_t slot, uint32_t host, bool multi, const char *mad, const char *timestr)
{
	auto u32 = [&](int off) { uint32_t v; memcpy(&v, r + off, 4); return v; };
	uint64_t glob_id;
	memcpy(&glob_id, r, 8);
	uint16_t ntasks_issue;
	memcpy(&ntasks_issue, r + 76, 2);
	const uint32_t nq = u32(8);
	char idbuf[24];
	snprintf(idbuf, sizeof(idbuf), "%016llx", (unsigned long long)glob_id);
	j.obj_open();
	if (multi) {
		j.kstr("parid", machid_string(c->hosts[host]));
		j.kstr("host", c->host_names[host]);
		j.kstr("madid", mad, 16);
		j.kstr("cluster", c->cluster_names[c->host_cluster_h[host]]);
	}
	j.kstr("time", timestr ? timestr : "", 64);
	j.kstr("svcid", idbuf, 16);
	j.kstr("name", c->svc_comm[slot].data(), 16);
	j.ku("qps5s", nq / 5);
	j.ku("nqry5s", nq);
	j.ku("resp5s", u32(12) / (nq > 0 ? nq : 1));
	j.ku("p95resp5s", u32(28));
	j.ku("p95resp5m", u32(32));
	j.ku("nconns", u32(16));
	j.ku("nactive", u32(20));
	j.ku("nprocs", u32(24));
	j.ku("kbin15s", u32(36));
	j.ku("kbout15s", u32(40));
	j.ku("sererr", u32(44));
	j.ku("clierr", u32(48));
	j.ku("delayus", u32(52));
	j.ku("cpudelus", u32(56));
	j.ku("iodelus", u32(60));
	// int64_t vmdelus = the UNSIGNED 32-bit difference of the three fields; printed when > 0 (server/gy_mfields.h:1648-1654): the wrapped value
	j.ku("vmdelus", (uint32_t)(u32(52) - u32(56) - u32(60)));
	j.ku("usercpu", u32(64));
	j.ku("syscpu", u32(68));
	j.ku("rssmb", u32(72));
	j.ku("nissue", ntasks_issue);
	j.kstr("state", state_string(r[79]), 8);
	j.ku("issue", r[80]);
	j.kb("ishttp", r[78] != 0);
	j.kstr("desc", "", 0); // the variable-length issue string is not retained by the engine
	j.obj_close();
}

} // namespace

extern "C" {

int gys_set_host_name(gys_ctx *c, const uint8_t machine_id[16], const char *hostname)
try {
	if (!c || !machine_id || !hostname) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	// PARTHA_INFO::hostname_ is a char[MAX_DOMAINNAME_SIZE] filled with GY_STRNCPY (a longer name is cut): the name criteria walk these strings
	c->host_names[host].assign(hostname, strnlen(hostname, 255));
	return GYS_OK;
} GYS_CATCH_ALL

int gys_json_svcsumm(gys_ctx *c, const uint8_t machine_id[16], const char *madhava_id16, const char *timestr, char *buf, size_t buflen, size_t *needed)
try {
	if (!c || !machine_id) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	gys_svcsumm s;
	rc = gys_query_svcsumm(c, machine_id, &s);
	if (rc) return rc;
	const char *mad = madhava_id16 ? madhava_id16 : "";
	JsonBuf j;
	j.s += '{';
	j.kstr("madid", mad, 16);
	j.arr_open("summstats");
	j.obj_open();
	j.kstr("time", timestr ? timestr : "", 64);
	j.ki("nidle", s.nstates[0]);
	j.ki("ngood", s.nstates[1]);
	j.ki("nok", s.nstates[2]);
	j.ki("nbad", s.nstates[3]);
	j.ki("nsevere", s.nstates[4]);
	j.ki("ndown", s.nstates[5]);
	j.ki("totqps", s.tot_qps);
	j.ki("totaconn", s.tot_act_conn);
	j.ki("totkbin", s.tot_kb_inbound);
	j.ki("totkbout", s.tot_kb_outbound);
	j.ki("totsererr", s.tot_ser_errors);
	j.ki("nsvc", s.nlisteners);
	j.ki("nactive", s.nactive);
	j.obj_close();
	j.arr_close();
	hostinfo_object(c, j, host, mad);
	j.s += '}';
	return json_out(j, buf, buflen, needed);
} GYS_CATCH_ALL

int gys_json_svcstate(gys_ctx *c, const uint8_t machine_id[16], const char *madhava_id16, const char *timestr, char *buf, size_t buflen, size_t *needed)
try {
	GYS_ENTER(c);
	if (!c || !machine_id) return GYS_ERR_INVAL;
	uint32_t host;
	int rc = lookup_host(c, machine_id, &host);
	if (rc) return rc;
	const char *mad = madhava_id16 ? madhava_id16 : "";
	const HostListeners &hl = c->host_lst[host];
	JsonBuf j;
	j.s += '{';
	j.kstr("madid", mad, 16);
	j.arr_open("svcstate");
	if (!hl.all_slots.empty()) {
		// the host's service slots are (mostly) contiguous: one copy of the covering range of 96-byte state records
		const uint32_t lo = *std::min_element(hl.all_slots.begin(), hl.all_slots.end()), hi = *std::max_element(hl.all_slots.begin(), hl.all_slots.end());
		std::vector<uint8_t> recs((size_t)(hi - lo + 1) * 96);
		HIPCHK(hipMemcpyAsync(recs.data(), c->svc_state + (size_t)lo * 96, recs.size(), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		for (uint32_t slot : hl.all_slots) {
			const uint8_t *r = recs.data() + (size_t)(slot - lo) * 96;
			uint64_t glob_id, tag;
			memcpy(&glob_id, r, 8);
			memcpy(&tag, r + 88, 8);
			const uint32_t ep = (uint32_t)tag;
			// the reference lists listeners whose state is at most 10 s old (:4660 min_stats_tusec): the current or the last window here
			if (ep == 0 || ep + 1 < c->epoch || (uint32_t)(tag >> 32) != host || glob_id != c->svc_gid_h[slot]) continue;
			svcstate_object(c, j, r, slot, host, false, mad, timestr);
		}
	}
	j.arr_close();
	hostinfo_object(c, j, host, mad);
	j.s += '}';
	return json_out(j, buf, buflen, needed);
} GYS_CATCH_ALL

int gys_json_clusterstate(gys_ctx *c, const char *shyama_id16, const char *timestr, char *buf, size_t buflen, size_t *needed)
try {
	GYS_ENTER(c);
	if (!c) return GYS_ERR_INVAL;
	const size_t nc = c->cluster_names.size();
	std::vector<uint32_t> v(nc * 12 + 1);
	if (nc) {
		const uint32_t *src = (const uint32_t *)(c->last + c->al.off_u32) + c->al.u32_cluster;
		HIPCHK(hipMemcpyAsync(v.data(), src, nc * 12 * 4, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	// AggrClusterStateMap is keyed by cluster name: emit in name order (server/gy_shconnhdlr.h clusterstatemap_)
	std::vector<uint32_t> order(nc);
	for (uint32_t i = 0; i < nc; ++i) order[i] = i;
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return c->cluster_names[a] < c->cluster_names[b]; });
	JsonBuf j;
	j.s += '{';
	j.kstr("shyamaid", shyama_id16 ? shyama_id16 : "", 16);
	j.arr_open("clusterstate");
	for (uint32_t i : order) {
		const uint32_t *s = v.data() + (size_t)i * 12;
		if (s[0] == 0) continue; // no host of this cluster reported in the window
		j.obj_open();
		j.kstr("time", timestr ? timestr : "", 64);
		j.kstr("cluster", c->cluster_names[i]);
		j.ku("nhosts", s[0]);
		j.ku("nprocissue", s[1]);
		j.ku("nprochosts", s[2]);
		j.ku("nproc", s[3]);
		j.ku("nlistissue", s[4]);
		j.ku("nlisthosts", s[5]);
		j.ku("nlisten", s[6]);
		j.ku("totqps", s[7]);
		j.ku("svcnetmb", s[8]);
		j.ku("ncpuissue", s[9]);
		j.ku("nmemissue", s[10]);
		j.obj_close();
	}
	j.arr_close();
	j.s += '}';
	return json_out(j, buf, buflen, needed);
} GYS_CATCH_ALL

// web_curr_top_listeners (server/gy_mnodehandle.cc:2706-3190).  machine_id != NULL: the single-host form (the host's four top-10 queues,
// optional summstats, hostinfo); NULL: the multi-host form -- every host's queues merged into MAX_MULTI_TOPN = 50 slots per kind
// (common/gy_json_field_maps.h:481, merge :2885-3040), each entry carrying parid / host / madid / cluster.
// flags: GYS_TOP_ISSUE | GYS_TOP_QPS | GYS_TOP_ACTCONN | GYS_TOP_NET | GYS_TOP_SUMMSTATS.  Entry fields = the svcstate fields in the
// order of stream_top (:2775-2880) + ip, port.  "ip" is empty: the engine's listener registry carries (netns, port), not the bind address.
static void top_entry(gys_ctx *c, JsonBuf &j, const uint8_t *r, uint32_t slot, uint32_t host, bool multi, const char *mad, const char *timestr)
{
	auto u32 = [&](int off) { uint32_t v; memcpy(&v, r + off, 4); return v; };
	uint64_t glob_id;
	memcpy(&glob_id, r, 8);
	uint16_t ntasks_issue;
	memcpy(&ntasks_issue, r + 76, 2);
	const uint32_t nq = u32(8);
	j.obj_open();
	if (multi) {
		j.kstr("parid", machid_string(c->hosts[host]));
		j.kstr("host", c->host_names[host]);
		j.kstr("madid", mad, 16);
		j.kstr("cluster", c->cluster_names[c->host_cluster_h[host]]);
	}
	char idbuf[24];
	snprintf(idbuf, sizeof(idbuf), "%016llx", (unsigned long long)glob_id);
	j.kstr("time", timestr ? timestr : "", 64);
	j.kstr("svcid", idbuf, 16);
	j.kstr("name", c->svc_comm[slot].data(), 16);
	j.ku("qps5s", nq / 5);
	j.ku("nqry5s", nq);
	j.ku("resp5s", u32(12) / (nq > 0 ? nq : 1));
	j.ku("p95resp5s", u32(28));
	j.ku("p95resp5m", u32(32));
	j.ku("nconns", u32(16));
	j.ku("nactive", u32(20));
	j.ku("nprocs", u32(24));
	j.ku("kbin15s", u32(36));
	j.ku("kbout15s", u32(40));
	j.ku("sererr", u32(44));
	j.ku("clierr", u32(48));
	j.ku("delayus", u32(52));
	j.ku("cpudelus", u32(56));
	j.ku("iodelus", u32(60));
	j.ku("vmdelus", (uint32_t)(u32(52) - u32(56) - u32(60))); // writer.Uint of the unsigned difference, as the reference (:2843)
	j.ku("usercpu", u32(64));
	j.ku("syscpu", u32(68));
	j.ku("rssmb", u32(72));
	j.ku("nissue", ntasks_issue);
	j.kstr("state", state_string(r[79]), 8);
	j.ku("issue", r[80]);
	j.kb("ishttp", r[78] != 0);
	j.kstr("desc", "", 0);
	j.kstr("ip", "", 0);
	j.ku("port", slot < c->svc_port_h.size() ? c->svc_port_h[slot] : 0);
	j.obj_close();
}

int gys_json_toplisteners(gys_ctx *c, const uint8_t machine_id[16], uint32_t flags, const char *madhava_id16, const char *timestr, char *buf, size_t buflen,
			  size_t *needed)
try {
	GYS_ENTER(c);
	if (!c) return GYS_ERR_INVAL;
	if (!(flags & (GYS_TOP_ISSUE | GYS_TOP_QPS | GYS_TOP_ACTCONN | GYS_TOP_NET))) {
		set_err("Top Listeners Query : Query requested with no valid Top criteria"); // (:2757)
		return GYS_ERR_INVAL;
	}
	const bool multi = machine_id == nullptr;
	uint32_t host = 0;
	if (!multi) {
		const int rc = lookup_host(c, machine_id, &host);
		if (rc) return rc;
	}
	const char *mad = madhava_id16 ? madhava_id16 : "";
	const uint32_t nh = (uint32_t)c->hosts.size();
	static const char *names[4] = {"topissue", "topqps", "topactconn", "topnet"};
	JsonBuf j;
	j.s += '{';
	j.kstr("madid", mad, 16);
	for (int kind = 0; kind < 4; ++kind) {
		if (!(flags & (1u << kind))) continue;
		std::vector<uint32_t> slots;
		std::vector<uint64_t> metrics;
		const int rc = topn_all_hosts(c, kind, slots, metrics);
		if (rc) return rc;
		std::vector<std::pair<uint64_t, uint32_t>> pick; // (metric, slot)
		if (multi) {
			for (size_t i = 0; i < slots.size(); ++i)
				if (slots[i] != GYS_NOSLOT) pick.emplace_back(metrics[i], slots[i]);
			std::sort(pick.begin(), pick.end(), [](const auto &a, const auto &b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
			if (pick.size() > GYS_MULTI_TOPN) pick.resize(GYS_MULTI_TOPN);
		} else {
			for (uint32_t r = 0; r < GYS_TOPN; ++r)
				if (slots[(size_t)host * GYS_TOPN + r] != GYS_NOSLOT) pick.emplace_back(metrics[(size_t)host * GYS_TOPN + r], slots[(size_t)host * GYS_TOPN + r]);
		}
		j.arr_open(names[kind]);
		for (const auto &pr : pick) {
			uint8_t rec[96];
			HIPCHK(hipMemcpy(rec, c->svc_state + (size_t)pr.second * 96, 96, hipMemcpyDeviceToHost));
			uint64_t tag;
			memcpy(&tag, rec + 88, 8);
			top_entry(c, j, rec, pr.second, (uint32_t)(tag >> 32), multi, mad, timestr);
		}
		j.arr_close();
	}
	if (flags & GYS_TOP_SUMMSTATS) { // send_listen_one_summ_stats: the svcsumm fields; multi-host: LISTEN_SUMM_STATS<int64_t>::update over the hosts
		long long v[13] = {0};
		std::vector<int32_t> all((size_t)nh * 16);
		if (nh) {
			HIPCHK(hipMemcpyAsync(all.data(), c->host_summ_last, all.size() * 4, hipMemcpyDeviceToHost, c->stream));
			HIPCHK(hipStreamSynchronize(c->stream));
		}
		for (uint32_t h = multi ? 0 : host; h < (multi ? nh : host + 1); ++h)
			for (int k = 0; k < 13; ++k) v[k] += all[(size_t)h * 16 + k];
		j.key("summstats");
		j.s += '{';
		j.first = true;
		j.kstr("time", timestr ? timestr : "", 64);
		j.ki("nidle", v[0]);
		j.ki("ngood", v[1]);
		j.ki("nok", v[2]);
		j.ki("nbad", v[3]);
		j.ki("nsevere", v[4]);
		j.ki("ndown", v[5]);
		j.ki("totqps", v[6]);
		j.ki("totaconn", v[7]);
		j.ki("totkbin", v[8]);
		j.ki("totkbout", v[9]);
		j.ki("totsererr", v[10]);
		j.ki("nsvc", v[11]);
		j.ki("nactive", v[12]);
		j.obj_close();
	}
	if (!multi) hostinfo_object(c, j, host, mad);
	j.s += '}';
	return json_out(j, buf, buflen, needed);
} GYS_CATCH_ALL

} // extern "C"

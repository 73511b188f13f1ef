"""Host-side wrapper around the C ABI (plumbing for tests / bench).  It mirrors the reference's aggregation entry points
(MCONN_HANDLER::partha_tcp_conn_info / partha_listener_state / send_cluster_state, server/gy_mconnhdlr.h:2091-2156) by name and
argument meaning; the multi-GPU step (one exchange per window) is an all-reduce of the fixed-size sketch registers
through torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests)."""
import ctypes as C

import numpy as np

from . import capi, wire

_TORCH_DTYPES = None


def _torch_dtypes():
    global _TORCH_DTYPES
    if _TORCH_DTYPES is None:
        import torch
        _TORCH_DTYPES = {0: torch.uint8, 1: torch.int32, 2: torch.int64}  # u32 sums wrap identically as int32
    return _TORCH_DTYPES


def allreduce_sections(sections, group=None):
    """sections: list of (tensor, op) with op 0 = MAX, 1 = SUM.  One collective per register family (HLL = max on u8,
    CMS / histogram / cluster counters = sum); no-op when torch.distributed is not initialised or world_size == 1."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t, op in sections:
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 0 else dist.ReduceOp.SUM, group=group)


def mid_buf(machine_id):
    assert len(machine_id) == 16
    return (C.c_uint8 * 16)(*machine_id)


class SketchEngine:
    def __init__(self, max_hosts, max_services, max_clusters=16, enable_tdigest=True, svc_hll_p=0, max_batch_events=1 << 20,
                 rank=0, nranks=1, device=None, torch_arena=True, resp_path=0, enable_levels=False, td_buf_values=0, conn_pair_cms=False, td_pend_cap=0):
        import torch
        self.L = capi.load()
        if not torch.cuda.is_available():
            raise RuntimeError("SketchEngine needs a HIP device: the sketch engine has no CPU path")
        self.torch = torch
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device)
        cfg = capi.Config()
        cfg.struct_size = C.sizeof(capi.Config)
        cfg.device = int(device)
        cfg.rank, cfg.nranks = rank, nranks
        cfg.max_hosts, cfg.max_services, cfg.max_clusters = max_hosts, max_services, max_clusters
        cfg.enable_tdigest = 1 if enable_tdigest else 0
        cfg.svc_hll_p = svc_hll_p
        cfg.resp_path = resp_path
        cfg.enable_levels = int(enable_levels)  # False / True / 2 (without the 5-s level)
        cfg.td_buf_values = td_buf_values
        cfg.td_pend_cap = td_pend_cap
        cfg.conn_pair_cms = 1 if conn_pair_cms else 0
        cfg.max_batch_events = max_batch_events
        with torch.cuda.device(self.device):
            # the engine gets its own torch stream: torch work (tensor fills / copies on the current stream, collectives) and engine work are
            # ordered explicitly -- order() before handing device buffers to the engine, sync() (or stream.synchronize) before reading results
            self.stream = torch.cuda.Stream(device=self.device)
            cfg.stream = self.stream.cuda_stream
            self.arena = None
            if torch_arena:
                nbytes = self.L.gys_reduce_arena_bytes(C.byref(cfg))
                self.arena = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
                cfg.reduce_arena = self.arena.data_ptr()
                cfg.reduce_arena_bytes = nbytes
            h = C.c_void_p()
            capi.check(self.L.gys_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.cfg = cfg
        self.rank, self.nranks = rank, nranks
        self._sections = None

    def order(self):
        """engine work submitted after this call starts after everything queued so far on torch's current stream (e.g. the fill / copy
        that produced a device buffer about to be handed to the engine)"""
        self.stream.wait_stream(self.torch.cuda.current_stream(self.device))

    def close(self):
        if self.h:
            self.L.gys_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- registration
    def register_cluster(self, name):
        idx = C.c_uint32()
        capi.check(self.L.gys_register_cluster(self.h, name.encode(), C.byref(idx)))
        return idx.value

    def register_host(self, machine_id, cluster="cluster0"):
        slot = C.c_uint32()
        capi.check(self.L.gys_register_host(self.h, mid_buf(machine_id), cluster.encode(), C.byref(slot)))
        return slot.value

    def owns(self, machine_id):
        return self.L.gys_shard_of(mid_buf(machine_id), self.nranks) == self.rank

    def register_listeners(self, machine_id, glob_ids, netns, ports, comm=b"svc", addrs=None):
        """addrs: per listener None (an any-address listener: NEW_LISTENER::is_any_ip_) or the 4 / 16 address bytes it is bound to"""
        n = len(glob_ids)
        arr = (capi.ListenerInfo * n)()
        for i in range(n):
            arr[i].glob_id = int(glob_ids[i])
            arr[i].netns = int(netns[i])
            arr[i].port = int(ports[i])
            arr[i].comm = comm
            a = addrs[i] if addrs is not None else None
            arr[i].is_any_ip = 1 if a is None else 0
            if a is not None:
                a = bytes(a)
                arr[i].addr_is_v6 = 1 if len(a) == 16 else 0
                arr[i].addr = (C.c_uint8 * 16)(*(a + bytes(16))[:16])
        first = C.c_uint32()
        capi.check(self.L.gys_register_listeners(self.h, mid_buf(machine_id), arr, n, C.byref(first)))
        return first.value

    LISTENER_INFO_DT = np.dtype([("glob_id", "<u8"), ("netns", "<u4"), ("port", "<u2"), ("is_any_ip", "u1"), ("addr_is_v6", "u1"), ("comm", "S16"),
                                 ("addr", "u1", (16,))])

    def register_listeners_np(self, machine_id, glob_ids, netns, ports):
        """bulk variant: fills the gys_listener_info array through numpy (no per-element python loop); any-address listeners"""
        n = len(glob_ids)
        a = np.zeros(n, dtype=self.LISTENER_INFO_DT)
        a["glob_id"], a["netns"], a["port"], a["comm"], a["is_any_ip"] = glob_ids, netns, ports, b"svc", 1
        first = C.c_uint32()
        capi.check(self.L.gys_register_listeners(self.h, mid_buf(machine_id), a.ctypes.data_as(C.POINTER(capi.ListenerInfo)), n, C.byref(first)))
        return first.value

    # ---------------------------------------------------------------- ingest (names follow the reference entry points)
    def handle_resp_events(self, machine_id, events):
        """TCP_SOCK_HANDLER::handle_ipv4_resp_event for a host batch (numpy RESP_EVENT array or bytes)"""
        b = events.tobytes() if hasattr(events, "tobytes") else bytes(events)
        capi.check(self.L.gys_ingest_resp_events(self.h, mid_buf(machine_id), b, len(b) // 24))

    def handle_resp_events_v6(self, machine_id, events):
        """TCP_SOCK_HANDLER::handle_ipv6_resp_event for a host batch (numpy RESP_EVENT6 array or bytes)"""
        b = events.tobytes() if hasattr(events, "tobytes") else bytes(events)
        capi.check(self.L.gys_ingest_resp_events_v6(self.h, mid_buf(machine_id), b, len(b) // 48))

    def handle_resp_events_v6_dev(self, segs, d_ev, nevents):
        self.order()
        capi.check(self.L.gys_ingest_resp_events_v6_dev(self.h, segs, len(segs), C.c_void_p(d_ev), nevents))

    def handle_resp_events_dev(self, segs, d_ev, nevents):
        self.order()
        capi.check(self.L.gys_ingest_resp_events_dev(self.h, segs, len(segs), C.c_void_p(d_ev), nevents))

    def partha_tcp_conn_info(self, machine_id, batch_bytes, nconns):
        """MCONN_HANDLER::partha_tcp_conn_info(partha, pone, nconns, pendptr)"""
        buf = np.frombuffer(batch_bytes, dtype=np.uint8)
        buf = np.require(buf, requirements=["A", "C"])
        p = buf.ctypes.data
        capi.check(self.L.gys_ingest_tcp_conn(self.h, mid_buf(machine_id), C.c_void_p(p), nconns, C.c_void_p(p + len(buf))))

    def partha_listener_state(self, machine_id, batch_bytes, nrecs):
        """MCONN_HANDLER::partha_listener_state(partha, pone, nconns, pendptr)"""
        buf = np.frombuffer(batch_bytes, dtype=np.uint8)
        buf = np.require(buf, requirements=["A", "C"])
        p = buf.ctypes.data
        capi.check(self.L.gys_ingest_listener_state(self.h, mid_buf(machine_id), C.c_void_p(p), nrecs, C.c_void_p(p + len(buf))))

    def handle_partha_active_conns(self, machine_id, batch_bytes, nitems):
        """MCONN_HANDLER::handle_partha_active_conns(partha, pconn, nitems, pendptr)"""
        buf = np.frombuffer(batch_bytes, dtype=np.uint8)
        buf = np.require(buf, requirements=["A", "C"])
        p = buf.ctypes.data
        capi.check(self.L.gys_ingest_active_conns(self.h, mid_buf(machine_id), C.c_void_p(p), nitems, C.c_void_p(p + len(buf))))

    def pair_cms(self, listener_glob_id, cli_aggr_task_id, which=0):
        out = C.c_uint64()
        capi.check(self.L.gys_query_pair_cms(self.h, int(listener_glob_id), int(cli_aggr_task_id), which, C.byref(out)))
        return out.value

    def export_pair_cms(self, which=0):
        out = np.zeros((capi.CMS_D, capi.CMS_W), dtype=np.int64 if which & 1 else np.uint32)
        capi.check(self.L.gys_export_pair_cms(self.h, which, C.c_void_p(out.ctypes.data)))
        return out

    def export_active_conn_counters(self, first=0, n=None):
        n = self.num_services() - first if n is None else n
        out = np.zeros((n, 4), dtype=np.uint64)
        capi.check(self.L.gys_export_active_conn_counters(self.h, first, n, out.ctypes.data_as(capi.u64p)))
        return out

    def handle_comm_stream(self, machine_id, stream_bytes):
        """L1 + L2 of madhava for one partha connection: COMM_HEADER-framed messages -> partha_tcp_conn_info / partha_listener_state"""
        buf = np.frombuffer(stream_bytes, dtype=np.uint8)
        buf = np.require(buf, requirements=["A", "C"])
        st = capi.CommStats()
        capi.check(self.L.gys_ingest_comm_stream(self.h, mid_buf(machine_id), C.c_void_p(buf.ctypes.data), len(buf), C.byref(st)))
        return st

    def handle_host_state(self, machine_id, ntasks_issue=0, ntasks=0, nlisten_issue=0, nlisten=0, cpu_issue=0, mem_issue=0, curr_state=1):
        st = capi.HostState(ntasks_issue, ntasks, nlisten_issue, nlisten, cpu_issue, mem_issue, curr_state, 0)
        capi.check(self.L.gys_ingest_host_state(self.h, mid_buf(machine_id), C.byref(st)))

    # ---------------------------------------------------------------- window boundary
    def reduce_sections(self):
        if self._sections is None:
            secs = (capi.ReduceSection * 4)()
            n = C.c_uint32()
            capi.check(self.L.gys_reduce_sections(self.h, secs, C.byref(n)))
            out = []
            if self.arena is not None:
                base = self.arena.data_ptr()
                dts = _torch_dtypes()
                for i in range(n.value):
                    s = secs[i]
                    esz = {0: 1, 1: 4, 2: 8}[s.dtype]
                    off = s.dev_ptr - base
                    out.append((self.arena[off:off + s.nelems * esz].view(dts[s.dtype]), s.op))
            self._sections = out
        return self._sections

    def send_cluster_state(self, tusec=0, group=None):
        """MCONN_HANDLER::send_cluster_state + SHCONN_HANDLER::aggregate_cluster_state: close the window on every rank."""
        if self.nranks <= 1:  # the whole boundary as one captured hipGraph (gys_window_close)
            capi.check(self.L.gys_window_close(self.h, tusec))
            return
        capi.check(self.L.gys_window_prepare(self.h, tusec))
        if self.nranks > 1:
            if self.arena is None:
                raise RuntimeError("multi-rank reduce needs torch_arena=True")
            with self.torch.cuda.stream(self.stream):  # the collectives read / write the arena in the engine stream's order
                allreduce_sections(self.reduce_sections(), group)
        capi.check(self.L.gys_window_finish(self.h))

    window_close = send_cluster_state

    # the same boundary with the collectives INSIDE the library (RCCL, no torch in the data path): gys_window_close_rccl
    def join_rccl(self, uid_bytes):
        """uid_bytes: the 128 bytes of gys_rccl_unique_id from rank 0 (every rank must call this; ncclCommInitRank is collective)"""
        uid = (C.c_uint8 * capi.RCCL_UID_BYTES)(*uid_bytes)
        comm = C.c_void_p()
        capi.check(self.L.gys_rccl_comm_create(self.h, uid, max(self.nranks, 1), self.rank, C.byref(comm)))
        self.comm = comm

    def rccl_unique_id(self):
        uid = (C.c_uint8 * capi.RCCL_UID_BYTES)()
        capi.check(self.L.gys_rccl_unique_id(uid))
        return bytes(uid)

    def window_close_rccl(self, tusec=0):
        capi.check(self.L.gys_window_close_rccl(self.h, self.comm, tusec))

    def leave_rccl(self):
        if getattr(self, "comm", None):
            self.sync()
            capi.check(self.L.gys_rccl_comm_destroy(self.comm))
            self.comm = None

    def sync(self):
        capi.check(self.L.gys_sync(self.h))

    # ---------------------------------------------------------------- queries / exports
    def svcsumm(self, machine_id):
        out = capi.SvcSumm()
        capi.check(self.L.gys_query_svcsumm(self.h, mid_buf(machine_id), C.byref(out)))
        return out

    def clusterstate(self, name):
        out = capi.ClusterState()
        capi.check(self.L.gys_query_clusterstate(self.h, name.encode(), C.byref(out)))
        return out

    def scan_quantiles(self, qs):
        """t-digest quantiles of EVERY service in one device pass: [nsvc][len(qs)] float64 (gys_scan_quantiles_dev)"""
        n = self.num_services()
        out = self.torch.empty((max(n, 1), len(qs)), dtype=self.torch.float64, device=self.device)
        qa = (C.c_double * len(qs))(*qs)
        self.order()
        capi.check(self.L.gys_scan_quantiles_dev(self.h, qa, len(qs), C.c_void_p(out.data_ptr())))
        return out[:n].cpu().numpy()

    LSCAN_DT = np.dtype([("glob_id", "<u8"), ("tcount", "<i8", 4), ("tsum", "<i8", 4), ("p95_ms", "<i4", 4), ("p99_ms", "<i4", 4), ("p25_ms", "<i4", 4),
                         ("last_qps", "<i4"), ("curr_qps", "<i4"), ("qps_p95", "<i4"), ("qps_p25", "<i4"), ("act_p95", "<i4"), ("act_p25", "<i4"),
                         ("b5", "u1"), ("b300", "u1"), ("b5day", "u1"), ("nconn_active", "u1"), ("nactive_conn_arr", "u1", 15), ("reserved", "u1", 5)])
    assert LSCAN_DT.itemsize == 168

    def scan_listener_state(self, tusec, qps_multiple=1.0, diffsec=5):
        """the per-listener 5-s scan from the engine's own state (gys_scan_listener_state_dev): (device tensor of the 88-byte
        LISTENER_STATE_NOTIFY records, the same as a numpy record array, the gys_listener_scan records)"""
        n = self.num_services()
        notify = self.torch.zeros(max(n, 1) * 88, dtype=self.torch.uint8, device=self.device)
        scan = self.torch.zeros(max(n, 1) * self.LSCAN_DT.itemsize, dtype=self.torch.uint8, device=self.device)
        self.order()
        capi.check(self.L.gys_scan_listener_state_dev(self.h, int(tusec), float(qps_multiple), int(diffsec), C.c_void_p(notify.data_ptr()),
                                                      C.c_void_p(scan.data_ptr())))
        self.sync()
        return (notify, np.frombuffer(notify.cpu().numpy().tobytes(), dtype=wire.LISTENER_STATE_NOTIFY)[:n],
                np.frombuffer(scan.cpu().numpy().tobytes(), dtype=self.LSCAN_DT)[:n])

    ISSUE_IN_DT = np.dtype([("ser_errors", "<u4"), ("tasks_delay_msec", "<u4"), ("tasks_cpudelay_msec", "<u4"), ("tasks_blkiodelay_msec", "<u4"), ("nconn", "<i4"),
                            ("ntasks_issue", "<u2"), ("ntasks_noissue", "<u2"), ("flags", "u1"), ("pad", "u1", 7), ("tdiff_start", "<i8")])  # (the C struct's int64 sits at offset 32)
    DECISION_DT = np.dtype([("state", "u1"), ("issue", "u1"), ("issue_bit_hist", "u1"), ("high_resp_bit_hist", "u1"), ("decided_line", "<u2"), ("pad", "<u2")])

    def decide_listener_state(self, scan_np, issue_in=None, notify_dev=None):
        """TCP_LISTENER::get_curr_state for every listener (gys_decide_listener_state_dev): scan_np = the gys_listener_scan records (numpy, LSCAN_DT),
        issue_in = numpy ISSUE_IN_DT array or None, notify_dev = the scan's device tensor of 88-byte records (patched in place) or None.
        Returns the gys_listener_decision records (numpy)."""
        n = len(scan_np)
        assert self.ISSUE_IN_DT.itemsize == 40 and self.DECISION_DT.itemsize == 8
        d_scan = self.torch.from_numpy(np.frombuffer(np.ascontiguousarray(scan_np).tobytes(), dtype=np.uint8).copy()).to(self.device)
        d_in = None
        if issue_in is not None:
            assert issue_in.dtype == self.ISSUE_IN_DT and len(issue_in) == n
            d_in = self.torch.from_numpy(np.frombuffer(np.ascontiguousarray(issue_in).tobytes(), dtype=np.uint8).copy()).to(self.device)
        d_out = self.torch.zeros(max(n, 1) * 8, dtype=self.torch.uint8, device=self.device)
        self.order()
        capi.check(self.L.gys_decide_listener_state_dev(self.h, C.c_void_p(d_scan.data_ptr()), C.c_void_p(d_in.data_ptr() if d_in is not None else None),
                                                        C.c_void_p(notify_dev.data_ptr() if notify_dev is not None else None), C.c_void_p(d_out.data_ptr())))
        self.sync()
        return np.frombuffer(d_out.cpu().numpy().tobytes(), dtype=self.DECISION_DT)[:n]

    SLAB_DT = np.dtype([("sum", "<i8", capi.TD_NB), ("cnt", "<u8", capi.TD_NB), ("vmin", "<i8"), ("vmax", "<i8")])

    def tdigest_rollup(self, scope):
        """roll-up digests (gys_tdigest_rollup_dev): torch uint8 tensor of n slabs on the device + the same as a numpy record array"""
        n = {capi.ROLLUP_HOST: self.L.gys_num_hosts(self.h), capi.ROLLUP_CLUSTER: self.L.gys_num_clusters(self.h), capi.ROLLUP_GLOBAL: 1}[scope]
        dev = self.torch.zeros(max(n, 1) * C.sizeof(capi.TDigestSlab), dtype=self.torch.uint8, device=self.device)
        self.order()
        capi.check(self.L.gys_tdigest_rollup_dev(self.h, scope, C.c_void_p(dev.data_ptr())))
        self.sync()
        return dev, np.frombuffer(dev.cpu().numpy().tobytes(), dtype=self.SLAB_DT)[:n]

    def tdigest_merge_slabs(self, dev_slabs, n):
        out = self.torch.zeros(C.sizeof(capi.TDigestSlab), dtype=self.torch.uint8, device=self.device)
        self.order()
        capi.check(self.L.gys_tdigest_merge_slabs_dev(self.h, C.c_void_p(dev_slabs.data_ptr()), n, C.c_void_p(out.data_ptr())))
        self.sync()
        return out, np.frombuffer(out.cpu().numpy().tobytes(), dtype=self.SLAB_DT)[0]

    def slab_quantiles(self, dev_slab, qs, index=0):
        qa = (C.c_double * len(qs))(*qs)
        out = (C.c_double * len(qs))()
        capi.check(self.L.gys_tdigest_slab_quantiles(self.h, C.c_void_p(dev_slab.data_ptr() + index * C.sizeof(capi.TDigestSlab)), qa, len(qs), out))
        return list(out)

    def hist_percentiles(self, glob_id, pcts, which=1):
        pd = (capi.HistData * len(pcts))()
        for i, p in enumerate(pcts):
            pd[i].percentile = p
        total, maxv, avg = C.c_uint64(), C.c_int64(), C.c_float()
        capi.check(self.L.gys_query_hist_percentiles(self.h, int(glob_id), which, pd, len(pcts), C.byref(total), C.byref(maxv), C.byref(avg)))
        return [d.data_value for d in pd], [d.sum for d in pd], [d.count for d in pd], total.value, maxv.value, avg.value

    def quantiles(self, glob_id, qs):
        q = (C.c_double * len(qs))(*qs)
        out = (C.c_double * len(qs))()
        capi.check(self.L.gys_query_quantiles(self.h, int(glob_id), q, len(qs), out))
        return list(out)

    def distinct_flows(self):
        out = C.c_double()
        capi.check(self.L.gys_query_distinct_flows(self.h, C.byref(out)))
        return out.value

    def cms(self, glob_id, which=0):
        out = C.c_uint64()
        capi.check(self.L.gys_query_cms(self.h, int(glob_id), which, C.byref(out)))
        return out.value

    def topn(self, machine_id, kind):
        out = (capi.TopnEntry * capi.TOPN)()
        n = C.c_uint32()
        capi.check(self.L.gys_query_topn(self.h, mid_buf(machine_id), kind, out, C.byref(n)))
        return [(out[i].glob_id, out[i].metric, bytes(out[i].state)) for i in range(n.value)]

    def _json(self, fn, *args):
        """two-call pattern: ask for the size, then fetch"""
        need = C.c_size_t()
        rc = fn(self.h, *args, None, 0, C.byref(need))
        if rc not in (capi.OK, capi.ERR_NOMEM):
            capi.check(rc)
        buf = C.create_string_buffer(need.value + 1)
        capi.check(fn(self.h, *args, buf, need.value + 1, C.byref(need)))
        return buf.value.decode()

    def set_host_name(self, machine_id, hostname):
        capi.check(self.L.gys_set_host_name(self.h, mid_buf(machine_id), hostname.encode()))

    def json_svcsumm(self, machine_id, madid="0" * 16, timestr=""):
        return self._json(self.L.gys_json_svcsumm, mid_buf(machine_id), madid.encode(), timestr.encode())

    def json_svcstate(self, machine_id, madid="0" * 16, timestr=""):
        return self._json(self.L.gys_json_svcstate, mid_buf(machine_id), madid.encode(), timestr.encode())

    def json_toplisteners(self, machine_id=None, flags=15, madid="0" * 16, timestr=""):
        """web_curr_top_listeners; machine_id None = the multi-host form (50 slots per kind over all hosts)"""
        m = mid_buf(machine_id) if machine_id is not None else None
        return self._json(self.L.gys_json_toplisteners, m, flags, madid.encode(), timestr.encode())

    def json_svcsumm_multihost(self, terms=None, group_oper=(), top_oper="and", sort_col=None, sort_desc=True, maxrecs=1 << 30, machine_ids=None,
                               clusters=None, madid="0" * 16, timestr=""):
        f, keep = self._svc_filter(terms, group_oper, top_oper, machine_ids, None, clusters, cols=capi.SUMM_COLS)
        return self._json(self.L.gys_json_svcsumm_multihost, C.byref(f), -1 if sort_col is None else capi.SUMM_COLS.index(sort_col), 1 if sort_desc else 0,
                          maxrecs, madid.encode(), timestr.encode())

    def _svc_filter(self, terms, group_oper=(), top_oper="and", machine_ids=None, svcids=None, clusters=None, cols=None):
        """terms: [(column name, comparator, value or list of values, group = 0)]; group_oper: per group "and" / "or"; -> (SvcFilter, keep-alive)"""
        terms = list(terms or [])
        arr = (capi.SvcTerm * max(len(terms), 1))()
        setv = []
        for i, t in enumerate(terms):
            col, comp, val = t[0], t[1], t[2]
            arr[i].col = (cols or capi.SVC_COLS).index(col)
            arr[i].comp = capi.COMP[comp]
            arr[i].group = t[3] if len(t) > 3 else 0
            if comp in ("in", "notin"):
                arr[i].set_first = len(setv)
                arr[i].nvalues = len(val)
                setv += [int(v) for v in val]
            elif comp not in ("bit2", "bit3"):
                arr[i].value = int(val)
        f = capi.SvcFilter()
        f.terms = arr
        f.nterms = len(terms)
        sv = (C.c_int64 * max(len(setv), 1))(*setv)
        f.set_values = sv
        f.nset_values = len(setv)
        for g, o in enumerate(group_oper):
            f.group_oper[g] = 1 if o == "or" else 0
        f.top_oper = 1 if top_oper == "or" else 0
        mids = None
        if machine_ids:
            mids = (C.c_uint8 * (16 * len(machine_ids)))(*b"".join(machine_ids))
            f.machine_ids = mids
            f.nmachine_ids = len(machine_ids)
        ids = cl = None
        if svcids is not None and len(svcids):
            ids = (C.c_uint64 * len(svcids))(*[int(x) for x in svcids])
            f.svcids = ids
            f.nsvcids = len(svcids)
        if clusters:
            cl = (C.c_char_p * len(clusters))(*[x.encode() for x in clusters])
            f.clusters = cl
            f.nclusters = len(clusters)
        return f, (arr, sv, mids, ids, cl)

    def svcstate_percentiles(self, col, pcts, terms=None, group_oper=(), top_oper="and", machine_ids=None, svcids=None, clusters=None):
        """gys_query_svcstate_percentiles -> (values (int64 array), number matched)"""
        f, keep = self._svc_filter(terms, group_oper, top_oper, machine_ids, svcids, clusters)
        p = np.ascontiguousarray(pcts, dtype=np.float64)
        out = np.zeros(len(p), dtype=np.int64)
        nm = C.c_uint64()
        capi.check(self.L.gys_query_svcstate_percentiles(self.h, C.byref(f), capi.SVC_COLS.index(col), p.ctypes.data_as(capi.f64p), len(p),
                                                          out.ctypes.data_as(capi.i64p), C.byref(nm)))
        return out, nm.value

    def svc_ids_by_name(self, comp, patterns):
        """gys_svc_ids_by_name: glob_ids of the registered services whose process name matches the string criterion (for svcids=...)"""
        pats = [patterns] if isinstance(patterns, (str, bytes)) else list(patterns)
        arr = (C.c_char_p * len(pats))(*[p if isinstance(p, bytes) else p.encode() for p in pats])
        cap = max(self.num_services(), 1)
        out = np.zeros(cap, dtype=np.uint64)
        n = C.c_uint32()
        capi.check(self.L.gys_svc_ids_by_name(self.h, capi.COMP[comp], arr, len(pats), out.ctypes.data_as(capi.u64p), cap, C.byref(n)))
        return out[:n.value].copy()

    def machine_ids_by_hostname(self, comp, patterns):
        """gys_machine_ids_by_hostname: machine ids (16 bytes each) of the registered hosts whose name matches (for machine_ids=...)"""
        pats = [patterns] if isinstance(patterns, (str, bytes)) else list(patterns)
        arr = (C.c_char_p * len(pats))(*[p if isinstance(p, bytes) else p.encode() for p in pats])
        cap = 1 << 16
        out = np.zeros(cap * 16, dtype=np.uint8)
        n = C.c_uint32()
        capi.check(self.L.gys_machine_ids_by_hostname(self.h, capi.COMP[comp], arr, len(pats), out.ctypes.data_as(capi.u8p), cap, C.byref(n)))
        return [out[16 * i:16 * i + 16].tobytes() for i in range(n.value)]

    def svcstate_scan(self, terms=None, group_oper=(), top_oper="and", sort_col=None, sort_desc=True, maxrecs=1000, machine_ids=None, svcids=None,
                      clusters=None):
        """gys_query_svcstate_scan -> (slots, host slots, records as a numpy array of wire.LISTENER_STATE_NOTIFY, number matched)"""
        f, keep = self._svc_filter(terms, group_oper, top_oper, machine_ids, svcids, clusters)
        out = (capi.SvcRow * max(maxrecs, 1))()
        nout, nm = C.c_uint32(), C.c_uint64()
        capi.check(self.L.gys_query_svcstate_scan(self.h, C.byref(f), -1 if sort_col is None else capi.SVC_COLS.index(sort_col), 1 if sort_desc else 0,
                                                  maxrecs, out, C.byref(nout), C.byref(nm)))
        raw = np.frombuffer(out, dtype=np.uint8, count=nout.value * 96).reshape(nout.value, 96)
        slots = raw[:, 0:4].copy().view(np.uint32).ravel()
        hosts = raw[:, 4:8].copy().view(np.uint32).ravel()
        recs = raw[:, 8:96].copy().view(wire.LISTENER_STATE_NOTIFY).ravel()
        return slots, hosts, recs, nm.value

    def json_svcstate_multihost(self, terms=None, group_oper=(), top_oper="and", sort_col=None, sort_desc=True, maxrecs=1000, machine_ids=None,
                                madid="0" * 16, timestr=""):
        f, keep = self._svc_filter(terms, group_oper, top_oper, machine_ids)
        return self._json(self.L.gys_json_svcstate_multihost, C.byref(f), -1 if sort_col is None else capi.SVC_COLS.index(sort_col), 1 if sort_desc else 0,
                          maxrecs, madid.encode(), timestr.encode())

    def svcstate_aggr(self, cols, group_by=0, terms=None, group_oper=(), top_oper="and", machine_ids=None, maxrows=None, svcids=None, clusters=None):
        """gys_query_svcstate_aggr -> list of (group, count, {col: (sum, min, max)})"""
        f, keep = self._svc_filter(terms, group_oper, top_oper, machine_ids, svcids, clusters)
        ca = (C.c_uint8 * max(len(cols), 1))(*[capi.SVC_COLS.index(c) for c in cols])
        if maxrows is None:
            maxrows = 1 if group_by == 0 else 1 << 16
        out = (capi.SvcAggrRow * max(maxrows, 1))()
        n = C.c_uint32()
        capi.check(self.L.gys_query_svcstate_aggr(self.h, C.byref(f), group_by, ca, len(cols), out, maxrows, C.byref(n)))
        res = []
        for i in range(min(n.value, maxrows)):
            r = out[i]
            res.append((r.group, r.count, {c: (r.sum[a], r.min[a], r.max[a]) for a, c in enumerate(cols)}))
        return res

    def json_clusterstate(self, shyamaid="0" * 16, timestr=""):
        return self._json(self.L.gys_json_clusterstate, shyamaid.encode(), timestr.encode())

    def num_services(self):
        return self.L.gys_num_services(self.h)

    def lookup(self, glob_id):
        s = C.c_uint32()
        capi.check(self.L.gys_lookup_service(self.h, int(glob_id), C.byref(s)))
        return s.value

    def export_hist(self, which, first=0, n=None):
        n = self.num_services() - first if n is None else n
        out = np.zeros((n, 16, 2), dtype=np.int64)  # [slot][bucket]{count,sum}; [slot][15] = {total_count, max_val_seen}
        capi.check(self.L.gys_export_hist(self.h, which, first, n, C.c_void_p(out.ctypes.data)))
        return out

    def tdigest_sql_text(self, glob_id):
        """the service's digest as a literal of the Postgres tdigest type ('...'::public.tdigest)"""
        need = C.c_size_t()
        rc = self.L.gys_tdigest_sql_text(self.h, int(glob_id), None, 0, C.byref(need))
        if rc != capi.ERR_NOMEM:
            capi.check(rc)
        buf = C.create_string_buffer(need.value + 1)
        capi.check(self.L.gys_tdigest_sql_text(self.h, int(glob_id), buf, len(buf), C.byref(need)))
        return buf.value.decode()

    def tdigest_sql_binary(self, glob_id):
        need = C.c_size_t()
        rc = self.L.gys_tdigest_sql_binary(self.h, int(glob_id), None, 0, C.byref(need))
        if rc != capi.ERR_NOMEM:
            capi.check(rc)
        buf = C.create_string_buffer(need.value)
        capi.check(self.L.gys_tdigest_sql_binary(self.h, int(glob_id), buf, len(buf), C.byref(need)))
        return buf.raw

    def export_hist_level(self, level, tusec, first=0, n=None):
        """records of one time level (0: last window, 1: 300 s, 2: 5 days, 3: all) as of tusec; same layout as export_hist"""
        n = self.num_services() - first if n is None else n
        out = np.zeros((n, 16, 2), dtype=np.int64)
        capi.check(self.L.gys_export_hist_level(self.h, level, int(tusec), first, n, C.c_void_p(out.ctypes.data)))
        return out

    def query_hist_level_stats(self, glob_id, level, tusec, pcts):
        st = (capi.TimeHistVal * len(pcts))()
        for i, p in enumerate(pcts):
            st[i].percentile = p
        tc, ts, mean = C.c_int64(), C.c_int64(), C.c_double()
        capi.check(self.L.gys_query_hist_level_stats(self.h, int(glob_id), level, int(tusec), st, len(pcts), C.byref(tc), C.byref(ts), C.byref(mean)))
        return [s.data_value for s in st], tc.value, ts.value, mean.value

    def export_hist_period(self, starttime, endtime, tusec, first=0, n=None):
        """{count, sum} per histogram bucket of the seconds [starttime, endtime] as TIME_HISTOGRAM::get_stats_for_period sees them
        (common/gy_statistics.h:1378-1406); returns (records [n][16][2], level that answered)"""
        n = self.num_services() - first if n is None else n
        out = np.zeros((n, 16, 2), dtype=np.int64)
        lv = C.c_int(-1)
        capi.check(self.L.gys_export_hist_period(self.h, int(starttime), int(endtime), int(tusec), first, n, C.c_void_p(out.ctypes.data), C.byref(lv)))
        return out, lv.value

    def query_hist_period_stats(self, glob_id, starttime, endtime, tusec, pcts):
        st = (capi.TimeHistVal * len(pcts))()
        for i, p in enumerate(pcts):
            st[i].percentile = p
        tc, ts, mean = C.c_int64(), C.c_int64(), C.c_double()
        capi.check(self.L.gys_query_hist_period_stats(self.h, int(glob_id), int(starttime), int(endtime), int(tusec), st, len(pcts), C.byref(tc),
                                                      C.byref(ts), C.byref(mean)))
        return [s.data_value for s in st], tc.value, ts.value, mean.value

    def export_day_stats(self, tusec, first=0, n=None):
        n = self.num_services() - first if n is None else n
        out = (capi.ListenerDayStats * n)()
        capi.check(self.L.gys_export_day_stats(self.h, int(tusec), first, n, C.cast(out, C.c_void_p)))
        return out

    def export_svc_hist(self, which, first=0, n=None):
        n = self.num_services() - first if n is None else n
        out = np.zeros((n, 16, 2), dtype=np.int64)
        capi.check(self.L.gys_export_svc_hist(self.h, which, first, n, C.c_void_p(out.ctypes.data)))
        return out

    def export_conn_bitmap(self, first=0, n=None):
        n = self.num_services() - first if n is None else n
        out = np.zeros((n, 64), dtype=np.uint16)  # rows 0..31: resp_bitmap_v4_, 32..63: resp_bitmap_v6_
        capi.check(self.L.gys_export_conn_bitmap(self.h, first, n, C.c_void_p(out.ctypes.data)))
        return out

    def export_hll(self):
        out = np.zeros(1 << capi.HLL_P, dtype=np.uint8)
        capi.check(self.L.gys_export_hll(self.h, C.c_void_p(out.ctypes.data)))
        return out

    def export_cms(self, which=0):
        out = np.zeros((capi.CMS_D, capi.CMS_W), dtype=np.int64 if which & 1 else np.uint32)
        capi.check(self.L.gys_export_cms(self.h, which, C.c_void_p(out.ctypes.data)))
        return out

    def export_global_hist(self):
        out = capi.HistRec()
        capi.check(self.L.gys_export_global_hist(self.h, C.byref(out)))
        return out

    def export_tdigest(self, first=0, n=None):
        n = self.num_services() - first if n is None else n
        sums = np.zeros((n, capi.TD_NB), dtype=np.int64)
        cnts = np.zeros((n, capi.TD_NB), dtype=np.uint32)
        mm = np.zeros((n, 2), dtype=np.int32)
        capi.check(self.L.gys_export_tdigest(self.h, first, n, C.c_void_p(sums.ctypes.data), C.c_void_p(cnts.ctypes.data), C.c_void_p(mm.ctypes.data)))
        return sums, cnts, mm

    def export_tdigest_pending(self, first=0, n=None):
        """(npend [n], pend [n][CAP]): each row's live prefix sorted ascending, the rest -1 (the buffer itself is unordered)"""
        n = self.num_services() - first if n is None else n
        npend = np.zeros(n, dtype=np.uint32)
        pend = np.zeros((n, self.L.gys_td_pend_cap(self.h)), dtype=np.int32)
        capi.check(self.L.gys_export_tdigest_pending(self.h, first, n, C.c_void_p(npend.ctypes.data), C.c_void_p(pend.ctypes.data)))
        out = np.full_like(pend, -1)
        for i in range(n):
            out[i, :npend[i]] = np.sort(pend[i, :npend[i]])
        return npend, out

    def export_svc_counters(self, first=0, n=None):
        n = self.num_services() - first if n is None else n
        out = np.zeros((n, 4), dtype=np.uint64)
        capi.check(self.L.gys_export_svc_counters(self.h, first, n, C.c_void_p(out.ctypes.data)))
        return out

    def export_svc_hll(self, first=0, n=None):
        n = self.num_services() - first if n is None else n
        out = np.zeros((n, 1 << self.cfg.svc_hll_p), dtype=np.uint8)
        capi.check(self.L.gys_export_svc_hll(self.h, first, n, C.c_void_p(out.ctypes.data)))
        return out

    def resp_queue_pending(self):
        out = C.c_uint64()
        capi.check(self.L.gys_resp_queue_pending(self.h, C.byref(out)))
        return out.value

    def counters(self):
        out = capi.Counters()
        capi.check(self.L.gys_get_counters(self.h, C.byref(out)))
        return {n: getattr(out, n) for n, _ in out._fields_}

    # ---------------------------------------------------------------- measurement helpers
    def profile(self, on=True):
        capi.check(self.L.gys_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        capi.check(self.L.gys_profile_reset(self.h))

    def profile_get(self):
        buf = C.create_string_buffer(1024)
        capi.check(self.L.gys_profile_names(self.h, buf, 1024))
        out = {}
        for name in filter(None, buf.value.decode().split(",")):
            ms, n = C.c_double(), C.c_uint64()
            self.L.gys_profile_get(self.h, name.encode(), C.byref(ms), C.byref(n))
            out[name] = (ms.value, n.value)
        return out

    def gen_resp_events(self, d_ev, nevents, seed, first_host, nhosts, svcs_per_host, zipf_milli=0):
        segs = (capi.RespSeg * nhosts)()
        self.order()
        capi.check(self.L.gys_gen_resp_events_dev(self.h, C.c_void_p(d_ev), nevents, seed, first_host, nhosts, svcs_per_host, zipf_milli, segs))
        return segs

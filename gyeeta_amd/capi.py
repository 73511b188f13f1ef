"""ctypes binding of include/gysketch.h (the C ABI of libgysketch.so).  Plumbing only: every computation happens inside the HIP
library.  Importing this module never builds anything and never falls back to a CPU path: if the library is missing, load() raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GYS_LIB") or os.path.join(HERE, "lib", "libgysketch.so")  # GYS_LIB: an A/B build of the same library

OK, ERR_INVAL, ERR_NOMEM, ERR_HIP, ERR_NOTFOUND, ERR_NOT_OWNER, ERR_STATE, ERR_INTERNAL = 0, -1, -2, -3, -4, -5, -6, -7
MAX_BUCKETS, TD_NB, HLL_P, CMS_D, CMS_W, NSTATES, TOPN = 16, 200, 14, 4, 65536, 6, 10
NLEVELS, LEVEL_RING = 4, 10
TD_PEND_CAP = 896
RCCL_UID_BYTES = 128
KINDS = {"RESP_TIME_HASH": 0, "SEMI_LOG_HASH": 1, "SEMI_LOG_HASH_LO": 2, "DURATION_HASH": 3, "HASH_10_5000": 4, "HASH_5_250": 5,
         "HASH_1_3000": 6, "PERCENT_HASH": 7}


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("rank", C.c_uint32), ("nranks", C.c_uint32),
                ("max_hosts", C.c_uint32), ("max_services", C.c_uint32), ("max_clusters", C.c_uint32),
                ("enable_tdigest", C.c_uint32), ("svc_hll_p", C.c_uint32), ("resp_path", C.c_uint32),
                ("max_batch_events", C.c_uint64), ("stream", C.c_void_p), ("reduce_arena", C.c_void_p),
                ("reduce_arena_bytes", C.c_uint64), ("enable_levels", C.c_uint32), ("td_buf_values", C.c_uint32),
                ("conn_pair_cms", C.c_uint32), ("td_pend_cap", C.c_uint32)]


class ListenerInfo(C.Structure):
    _fields_ = [("glob_id", C.c_uint64), ("netns", C.c_uint32), ("port", C.c_uint16), ("is_any_ip", C.c_uint8), ("addr_is_v6", C.c_uint8),
                ("comm", C.c_char * 16), ("addr", C.c_uint8 * 16)]


class RespSeg(C.Structure):
    _fields_ = [("host_slot", C.c_uint32), ("reserved", C.c_uint32), ("first_event", C.c_uint64)]


class HostState(C.Structure):
    _fields_ = [("ntasks_issue", C.c_uint32), ("ntasks", C.c_uint32), ("nlisten_issue", C.c_uint32), ("nlisten", C.c_uint32),
                ("cpu_issue", C.c_uint8), ("mem_issue", C.c_uint8), ("curr_state", C.c_uint8), ("reserved", C.c_uint8)]


class CommStats(C.Structure):
    _fields_ = [("nmsgs", C.c_uint32), ("nmsgs_tcp_conn", C.c_uint32), ("nmsgs_listener_state", C.c_uint32), ("nmsgs_skipped", C.c_uint32),
                ("nmsgs_invalid", C.c_uint32), ("reserved", C.c_uint32), ("nrecords", C.c_uint64), ("bytes_consumed", C.c_uint64)]


class ReduceSection(C.Structure):
    _fields_ = [("dev_ptr", C.c_void_p), ("nelems", C.c_uint64), ("dtype", C.c_uint32), ("op", C.c_uint32)]


class SvcSumm(C.Structure):
    _fields_ = [("nstates", C.c_int32 * NSTATES), ("tot_qps", C.c_int32), ("tot_act_conn", C.c_int32), ("tot_kb_inbound", C.c_int32),
                ("tot_kb_outbound", C.c_int32), ("tot_ser_errors", C.c_int32), ("nlisteners", C.c_int32), ("nactive", C.c_int32)]

    def as_tuple(self):
        return tuple(self.nstates) + (self.tot_qps, self.tot_act_conn, self.tot_kb_inbound, self.tot_kb_outbound,
                                      self.tot_ser_errors, self.nlisteners, self.nactive)


class ClusterState(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("nhosts", "ntasks_issue", "ntaskissue_hosts", "ntasks", "nsvc_issue", "nsvcissue_hosts",
                                          "nsvc", "total_qps", "svc_net_mb", "ncpu_issue", "nmem_issue")]

    def as_tuple(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


class HistData(C.Structure):
    _fields_ = [("data_value", C.c_int64), ("sum", C.c_int64), ("count", C.c_uint64), ("percentile", C.c_float), ("reserved", C.c_uint32)]


class HistSerial(C.Structure):
    _fields_ = [("count", C.c_uint64), ("sum", C.c_int64)]


class HistRec(C.Structure):
    _fields_ = [("stats", HistSerial * 15), ("total_count", C.c_uint64), ("max_val_seen", C.c_int64)]


class TimeHistVal(C.Structure):
    _fields_ = [("data_value", C.c_int64), ("percentile", C.c_float), ("pad", C.c_uint32)]


class ListenerDayStats(C.Structure):
    _fields_ = [("glob_id", C.c_uint64), ("tcount_5d", C.c_int64), ("tsum_5d", C.c_int64), ("p95_5d_respms", C.c_uint32),
                ("p25_5d_respms", C.c_uint32), ("p95_qps", C.c_uint32), ("p25_qps", C.c_uint32), ("p95_nactive", C.c_uint32),
                ("p25_nactive", C.c_uint32)]


class TopnEntry(C.Structure):
    _fields_ = [("glob_id", C.c_uint64), ("host_slot", C.c_uint32), ("metric", C.c_uint32), ("state", C.c_uint8 * 88)]


class TDigestSlab(C.Structure):
    _fields_ = [("sum", C.c_int64 * TD_NB), ("cnt", C.c_uint64 * TD_NB), ("vmin", C.c_int64), ("vmax", C.c_int64)]


ROLLUP_HOST, ROLLUP_CLUSTER, ROLLUP_GLOBAL = 0, 1, 2


class SvcTerm(C.Structure):   # gys_svc_term
    _fields_ = [("col", C.c_uint8), ("comp", C.c_uint8), ("group", C.c_uint8), ("reserved", C.c_uint8), ("nvalues", C.c_uint32),
                ("set_first", C.c_uint32), ("reserved2", C.c_uint32), ("value", C.c_int64)]


class SvcFilter(C.Structure):  # gys_svc_filter
    _fields_ = [("terms", C.POINTER(SvcTerm)), ("nterms", C.c_uint32), ("nset_values", C.c_uint32), ("set_values", C.POINTER(C.c_int64)),
                ("group_oper", C.c_uint8 * 8), ("top_oper", C.c_uint8), ("reserved", C.c_uint8 * 3), ("nmachine_ids", C.c_uint32),
                ("machine_ids", C.POINTER(C.c_uint8)), ("svcids", C.POINTER(C.c_uint64)), ("nsvcids", C.c_uint32), ("nclusters", C.c_uint32),
                ("clusters", C.POINTER(C.c_char_p))]


class SvcRow(C.Structure):     # gys_svc_row
    _fields_ = [("slot", C.c_uint32), ("host_slot", C.c_uint32), ("rec", C.c_uint8 * 88)]


class SvcAggrRow(C.Structure):  # gys_svc_aggr_row
    _fields_ = [("group", C.c_uint32), ("ncols", C.c_uint32), ("count", C.c_uint64), ("sum", C.c_int64 * 8), ("min", C.c_int64 * 8), ("max", C.c_int64 * 8)]


SVC_COLS = ["qps5s", "nqry5s", "resp5s", "p95resp5s", "p95resp5m", "nconns", "nactive", "nprocs", "kbin15s", "kbout15s", "sererr", "clierr", "delayus",
            "cpudelus", "iodelus", "vmdelus", "usercpu", "syscpu", "rssmb", "nissue", "state", "issue", "ishttp"]  # GYS_SVC_COL_* in order
SUMM_COLS = ["nidle", "ngood", "nok", "nbad", "nsevere", "ndown", "totqps", "totaconn", "totkbin", "totkbout", "totsererr", "nsvc", "nactive"]  # GYS_SUMM_COL_*
COMP = {"=": 0, "!=": 1, "<": 2, "<=": 3, ">": 4, ">=": 5, "bit2": 6, "bit3": 7, "substr": 8, "notsubstr": 9, "like": 10, "notlike": 11, "in": 12, "notin": 13}  # GYS_COMP_* (COMPARATORS_E numbering)
AOPER = {"sum": 1, "avg": 2, "max": 3, "min": 4, "count": 5, "bool_or": 9, "bool_and": 10}  # GYS_AOPER_* (AGGR_OPER_E numbering)


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("resp_events", "resp_dropped_range", "resp_dropped_nolistener", "conn_events",
                                          "conn_unknown_service", "lstate_records", "lstate_missed", "lstate_errors", "lstate_deleted",
                                          "resp_batches_host_local", "resp_batches_general", "window_graph_launches", "resp_batches_host_split",
                                          "td_merges", "td_merge_values", "actconn_records", "actconn_remote_listen", "actconn_unknown_listener",
                                          "stage_waits", "resp_calls_queued", "resp_submissions", "conn_new", "conn_closed",
                                          "conn_closed_no_notify", "conn_client_side", "resp_tail_flushes", "conn_calls_queued", "conn_submissions",
                                          "lstate_calls_queued", "lstate_submissions", "rec_tail_flushes", "resp_run_overflow")]


assert C.sizeof(HistRec) == 256 and C.sizeof(TopnEntry) == 104 and C.sizeof(RespSeg) == 16 and C.sizeof(ListenerDayStats) == 48
assert C.sizeof(SvcTerm) == 24 and C.sizeof(SvcRow) == 96 and C.sizeof(SvcAggrRow) == 208

vp, u8p, u32p, u64p, i64p, i32p, f32p, f64p = (C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double))
mid = u8p  # machine id: 16 bytes

# name -> (restype, argtypes): must list EVERY function include/gysketch.h declares (tests/test_abi.py checks the header against this)
SIGNATURES = {
    "gys_abi_version": (C.c_uint32, []),
    "gys_td_pend_cap": (C.c_uint32, [vp]),
    "gys_last_error": (C.c_char_p, []),
    "gys_reduce_arena_bytes": (C.c_uint64, [C.POINTER(Config)]),
    "gys_create": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
    "gys_destroy": (None, [vp]),
    "gys_sync": (C.c_int, [vp]),
    "gys_machine_id_hash": (C.c_uint32, [mid]),
    "gys_shard_of": (C.c_uint32, [mid, C.c_uint32]),
    "gys_register_cluster": (C.c_int, [vp, C.c_char_p, u32p]),
    "gys_register_host": (C.c_int, [vp, mid, C.c_char_p, u32p]),
    "gys_register_listeners": (C.c_int, [vp, mid, C.POINTER(ListenerInfo), C.c_uint32, u32p]),
    "gys_ingest_resp_events": (C.c_int, [vp, mid, vp, C.c_uint32]),
    "gys_ingest_resp_events_dev": (C.c_int, [vp, C.POINTER(RespSeg), C.c_uint32, vp, C.c_uint64]),
    "gys_ingest_resp_events_v6": (C.c_int, [vp, mid, C.c_char_p, C.c_uint32]),
    "gys_ingest_resp_events_v6_dev": (C.c_int, [vp, C.POINTER(RespSeg), C.c_uint32, vp, C.c_uint64]),
    "gys_ingest_tcp_conn": (C.c_int, [vp, mid, vp, C.c_uint32, vp]),
    "gys_ingest_tcp_conn_dev": (C.c_int, [vp, vp, vp, C.c_uint32]),
    "gys_ingest_listener_state": (C.c_int, [vp, mid, vp, C.c_uint32, vp]),
    "gys_ingest_listener_state_dev": (C.c_int, [vp, vp, vp, vp, C.c_uint32]),
    "gys_ingest_comm_stream": (C.c_int, [vp, mid, vp, C.c_uint64, C.POINTER(CommStats)]),
    "gys_ingest_host_state": (C.c_int, [vp, mid, C.POINTER(HostState)]),
    "gys_reduce_sections": (C.c_int, [vp, C.POINTER(ReduceSection), u32p]),
    "gys_window_prepare": (C.c_int, [vp, C.c_uint64]),
    "gys_window_finish": (C.c_int, [vp]),
    "gys_query_svcsumm": (C.c_int, [vp, mid, C.POINTER(SvcSumm)]),
    "gys_query_clusterstate": (C.c_int, [vp, C.c_char_p, C.POINTER(ClusterState)]),
    "gys_query_hist_percentiles": (C.c_int, [vp, C.c_uint64, C.c_int, C.POINTER(HistData), C.c_uint32, u64p, i64p, f32p]),
    "gys_query_quantiles": (C.c_int, [vp, C.c_uint64, f64p, C.c_uint32, f64p]),
    "gys_query_distinct_flows": (C.c_int, [vp, f64p]),
    "gys_query_cms": (C.c_int, [vp, C.c_uint64, C.c_int, u64p]),
    "gys_query_topn": (C.c_int, [vp, mid, C.c_int, C.POINTER(TopnEntry), u32p]),
    "gys_scan_percentiles_dev": (C.c_int, [vp, C.c_int, f32p, C.c_uint32, vp]),
    "gys_scan_quantiles_dev": (C.c_int, [vp, f64p, C.c_uint32, vp]),
    "gys_scan_listener_state_dev": (C.c_int, [vp, C.c_uint64, C.c_float, C.c_uint32, vp, vp]),
    "gys_decide_listener_state_dev": (C.c_int, [vp, vp, vp, vp, vp]),
    "gys_tdigest_rollup_dev": (C.c_int, [vp, C.c_int, vp]),
    "gys_tdigest_merge_slabs_dev": (C.c_int, [vp, vp, C.c_uint32, vp]),
    "gys_tdigest_slab_quantiles": (C.c_int, [vp, vp, f64p, C.c_uint32, f64p]),
    "gys_num_clusters": (C.c_uint32, [vp]),
    "gys_ingest_active_conns": (C.c_int, [vp, mid, vp, C.c_uint32, vp]),
    "gys_ingest_active_conns_dev": (C.c_int, [vp, vp, C.c_uint32]),
    "gys_query_pair_cms": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_int, u64p]),
    "gys_export_pair_cms": (C.c_int, [vp, C.c_int, vp]),
    "gys_export_active_conn_counters": (C.c_int, [vp, C.c_uint32, C.c_uint32, u64p]),
    "gys_rccl_unique_id": (C.c_int, [u8p]),
    "gys_rccl_comm_create": (C.c_int, [vp, u8p, C.c_int, C.c_int, C.POINTER(vp)]),
    "gys_rccl_comm_destroy": (C.c_int, [vp]),
    "gys_window_close_rccl": (C.c_int, [vp, vp, C.c_uint64]),
    "gys_window_close": (C.c_int, [vp, C.c_uint64]),
    "gys_tdigest_global_rccl": (C.c_int, [vp, vp, vp]),
    "gys_tdigest_sql_text": (C.c_int, [vp, C.c_uint64, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gys_tdigest_sql_binary": (C.c_int, [vp, C.c_uint64, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gys_query_hist_level_stats": (C.c_int, [vp, C.c_uint64, C.c_int, C.c_uint64, C.POINTER(TimeHistVal), C.c_uint32, i64p, i64p, f64p]),
    "gys_export_hist_level": (C.c_int, [vp, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, vp]),
    "gys_export_day_stats": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_uint32, vp]),
    "gys_query_hist_period_stats": (C.c_int, [vp, C.c_uint64, C.c_int64, C.c_int64, C.c_uint64, C.POINTER(TimeHistVal), C.c_uint32, i64p, i64p, f64p]),
    "gys_export_hist_period": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_uint64, C.c_uint32, C.c_uint32, vp, C.POINTER(C.c_int)]),
    "gys_export_svc_hist": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_uint32, vp]),
    "gys_set_host_name": (C.c_int, [vp, mid, C.c_char_p]),
    "gys_json_svcsumm": (C.c_int, [vp, mid, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gys_json_svcstate": (C.c_int, [vp, mid, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gys_json_clusterstate": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gys_json_toplisteners": (C.c_int, [vp, mid, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gys_query_svcstate_scan": (C.c_int, [vp, C.POINTER(SvcFilter), C.c_int, C.c_int, C.c_uint32, C.POINTER(SvcRow), u32p, u64p]),
    "gys_json_svcstate_multihost": (C.c_int, [vp, C.POINTER(SvcFilter), C.c_int, C.c_int, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t,
                                              C.POINTER(C.c_size_t)]),
    "gys_query_svcstate_aggr": (C.c_int, [vp, C.POINTER(SvcFilter), C.c_int, u8p, C.c_uint32, C.POINTER(SvcAggrRow), C.c_uint32, u32p]),
    "gys_svc_aggr_value": (C.c_int, [C.POINTER(SvcAggrRow), C.c_uint32, C.c_int, f64p]),
    "gys_query_svcstate_percentiles": (C.c_int, [vp, C.POINTER(SvcFilter), C.c_int, f64p, C.c_uint32, i64p, u64p]),
    "gys_svc_ids_by_name": (C.c_int, [vp, C.c_int, C.POINTER(C.c_char_p), C.c_uint32, u64p, C.c_uint32, u32p]),
    "gys_machine_ids_by_hostname": (C.c_int, [vp, C.c_int, C.POINTER(C.c_char_p), C.c_uint32, u8p, C.c_uint32, u32p]),
    "gys_json_svcsumm_multihost": (C.c_int, [vp, C.POINTER(SvcFilter), C.c_int, C.c_int, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t,
                                             C.POINTER(C.c_size_t)]),
    "gys_num_services": (C.c_uint32, [vp]),
    "gys_num_hosts": (C.c_uint32, [vp]),
    "gys_lookup_service": (C.c_int, [vp, C.c_uint64, u32p]),
    "gys_export_hist": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_uint32, vp]),
    "gys_export_conn_bitmap": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
    "gys_export_hll": (C.c_int, [vp, vp]),
    "gys_export_cms": (C.c_int, [vp, C.c_int, vp]),
    "gys_export_tdigest": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, vp, vp]),
    "gys_export_tdigest_pending": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, vp]),
    "gys_export_svc_counters": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
    "gys_export_global_hist": (C.c_int, [vp, C.POINTER(HistRec)]),
    "gys_export_svc_hll": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
    "gys_get_counters": (C.c_int, [vp, C.POINTER(Counters)]),
    "gys_resp_queue_pending": (C.c_int, [vp, u64p]),
    "gys_hist_init_dev": (C.c_int, [vp, C.c_int, vp, C.c_uint32]),
    "gys_hist_add_dev": (C.c_int, [vp, C.c_int, vp, C.c_uint32, vp, vp, C.c_uint64]),
    "gys_hist_merge_dev": (C.c_int, [vp, vp, vp, C.c_uint32]),
    "gys_hist_percentiles_dev": (C.c_int, [vp, C.c_int, vp, C.c_uint32, f32p, C.c_uint32, vp]),
    "gys_profile_enable": (C.c_int, [vp, C.c_int]),
    "gys_profile_reset": (C.c_int, [vp]),
    "gys_profile_get": (C.c_int, [vp, C.c_char_p, f64p, u64p]),
    "gys_profile_names": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "gys_debug_read_events_dev": (C.c_int, [vp, vp, C.c_uint64]),
    "gys_gen_resp_events_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(RespSeg)]),
}

_lib = None


class GysError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgysketch error {code}: {msg}")
        self.code = code


def load():
    """Loads libgysketch.so.  torch (when present) is imported first so that both share ONE HIP runtime (same SONAME
    libamdhip64.so.7): device pointers of torch tensors are then valid inside the library and vice versa."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the sketch engine)")
    try:
        import torch  # noqa: F401  (plumbing: shares the HIP runtime, see docstring)
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)  # AttributeError here == the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise GysError(rc, load().gys_last_error().decode(errors="replace"))
    return rc

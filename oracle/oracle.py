"""TEST INFRASTRUCTURE ONLY: ctypes bindings for oracle/liboracle.so (the plain-C restatement, gy_oracle.c) and, when
built, oracle/_ref/libgyref.so (the reference's own headers compiled here by oracle/build_ref.sh).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libgyref.so")

KINDS = {
    "RESP_TIME_HASH": 0, "SEMI_LOG_HASH": 1, "SEMI_LOG_HASH_LO": 2, "DURATION_HASH": 3, "HASH_10_5000": 4,
    "HASH_5_250": 5, "HASH_1_3000": 6, "PERCENT_HASH": 7, "FIXED_9_26_5": 8, "FIXED_N15_N3_4": 9,
}
RESP_TIME_HASH = KINDS["RESP_TIME_HASH"]
MAX_BUCKETS = 16
TD_NB = 200
HLL_P = 14
CMS_D = 4
CMS_W = 65536

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)
HIST_SERIAL_DT = np.dtype([("count", "<u8"), ("sum", "<i8")])  # HIST_SERIAL as a numpy record


SOURCES = ["gy_oracle.c", "gy_oracle_engine.c", "gy_oracle_levels.c", "gy_oracle_rollup.c", "gy_oracle_lscan.c", "gy_oracle_lstate.c", "gy_oracle_query.c"]


def build_oracle(force=False):
    deps = [os.path.join(HERE, f) for f in SOURCES + ["gy_oracle.h"]] + [os.path.join(HERE, "..", "include", "gys_tdigest_tbl.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    # -fno-strict-overflow -fno-strict-aliasing: the reference's own flags (Makefile.common:42): its int sums (LISTEN_SUMM_STATS<int>,
    # server/gy_msocket.h:844-865) wrap, and so must the restatement's -- UBSan reports them as signed overflow otherwise
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-pthread", "-fno-strict-overflow", "-fno-strict-aliasing", "-o", LIB_PATH] + [os.path.join(HERE, f) for f in SOURCES] + ["-lm"])
    return LIB_PATH


class HistSerial(C.Structure):
    _fields_ = [("count", C.c_uint64), ("sum", C.c_int64)]


class Hist(C.Structure):
    _fields_ = [("kind", C.c_int), ("nbuckets", C.c_int), ("stats", HistSerial * MAX_BUCKETS),
                ("total_count", C.c_uint64), ("max_val_seen", C.c_int64)]


class HistData(C.Structure):
    _fields_ = [("data_value", C.c_int64), ("sum", C.c_int64), ("count", C.c_uint64), ("percentile", C.c_float)]


class TDigest(C.Structure):
    _fields_ = [("sum", C.c_int64 * TD_NB), ("cnt", C.c_uint32 * TD_NB), ("vmin", C.c_int32), ("vmax", C.c_int32)]


TD_PEND_CAP = 896


class TDBuffered(C.Structure):
    _fields_ = [("d", TDigest), ("npend", C.c_uint32), ("pend", C.c_int32 * TD_PEND_CAP), ("cap", C.c_uint32), ("ext", C.POINTER(C.c_int32))]

    def values(self):
        """the npend buffered values (a buffer larger than the default lives behind ext)"""
        src = self.ext if bool(self.ext) else self.pend
        return np.array(src[:self.npend], dtype=np.int32)


BTS_MAXB = 16
MLH_LEVELS = 4


class TD64(C.Structure):
    _fields_ = [("sum", C.c_int64 * TD_NB), ("cnt", C.c_uint64 * TD_NB), ("vmin", C.c_int64), ("vmax", C.c_int64)]


TD_BINS = 2048


class TDBins(C.Structure):
    _fields_ = [("cnt", C.c_uint64 * TD_BINS), ("sum", C.c_uint64 * TD_BINS), ("vmin", C.c_int64), ("vmax", C.c_int64)]


class BTS(C.Structure):
    _fields_ = [("duration", C.c_int64), ("nbuckets", C.c_uint32), ("first_time", C.c_int64), ("latest_time", C.c_int64),
                ("tot_sum", C.c_int64), ("tot_cnt", C.c_uint64), ("bsum", C.c_int64 * BTS_MAXB), ("bcnt", C.c_uint64 * BTS_MAXB)]


class MLHist(C.Structure):
    _fields_ = [("kind", C.c_int), ("nb", C.c_int), ("s", (BTS * MLH_LEVELS) * MAX_BUCKETS),
                ("cached_time", C.c_int64 * MAX_BUCKETS), ("cached", HistSerial * MAX_BUCKETS)]


class ListenerScan(C.Structure):
    """gyo_listener_scan == gys_listener_scan (include/gysketch.h)"""
    _fields_ = [("glob_id", C.c_uint64), ("tcount", C.c_int64 * MLH_LEVELS), ("tsum", C.c_int64 * MLH_LEVELS),
                ("p95_ms", C.c_int32 * MLH_LEVELS), ("p99_ms", C.c_int32 * MLH_LEVELS), ("p25_ms", C.c_int32 * MLH_LEVELS),
                ("last_qps", C.c_int32), ("curr_qps", C.c_int32), ("qps_p95", C.c_int32), ("qps_p25", C.c_int32),
                ("act_p95", C.c_int32), ("act_p25", C.c_int32), ("b5", C.c_uint8), ("b300", C.c_uint8), ("b5day", C.c_uint8),
                ("nconn_active", C.c_uint8), ("nactive_conn_arr", C.c_uint8 * 15), ("reserved", C.c_uint8 * 5)]


assert C.sizeof(ListenerScan) == 168

LI_TASK_ISSUE, LI_SEVERE, LI_DELAY, LI_CPU_ISSUE, LI_MEM_ISSUE, LI_DEPENDS, LI_YOUNG = 1, 2, 4, 8, 16, 32, 64


class ListenerIssueIn(C.Structure):
    """gyo_listener_issue_in == gys_listener_issue_in (include/gysketch.h)"""
    _fields_ = [("ser_errors", C.c_uint32), ("tasks_delay_msec", C.c_uint32), ("tasks_cpudelay_msec", C.c_uint32), ("tasks_blkiodelay_msec", C.c_uint32),
                ("nconn", C.c_int32), ("ntasks_issue", C.c_uint16), ("ntasks_noissue", C.c_uint16), ("flags", C.c_uint8), ("pad", C.c_uint8 * 3),
                ("tdiff_start", C.c_int64)]


class ListenerDecision(C.Structure):
    """gyo_listener_decision == gys_listener_decision"""
    _fields_ = [("state", C.c_uint8), ("issue", C.c_uint8), ("issue_bit_hist", C.c_uint8), ("high_resp_bit_hist", C.c_uint8),
                ("decided_line", C.c_uint16), ("pad", C.c_uint16)]


assert C.sizeof(ListenerIssueIn) == 40 and C.sizeof(ListenerDecision) == 8


class ListenSummStats(C.Structure):
    _fields_ = [("nstates", C.c_int32 * 6), ("tot_qps", C.c_int32), ("tot_act_conn", C.c_int32),
                ("tot_kb_inbound", C.c_int32), ("tot_kb_outbound", C.c_int32), ("tot_ser_errors", C.c_int32),
                ("nlisteners", C.c_int32), ("nactive", C.c_int32)]

    def as_tuple(self):
        return tuple(self.nstates) + (self.tot_qps, self.tot_act_conn, self.tot_kb_inbound, self.tot_kb_outbound,
                                      self.tot_ser_errors, self.nlisteners, self.nactive)


class ClusterStateOne(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("nhosts", "ntasks_issue", "ntaskissue_hosts", "ntasks", "nsvc_issue",
                                          "nsvcissue_hosts", "nsvc", "total_qps", "svc_net_mb", "ncpu_issue", "nmem_issue")]

    def as_tuple(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


def _sig(lib, name, restype, argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = argtypes
    return f


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build_oracle()
    L = C.CDLL(LIB_PATH)
    _sig(L, "gyo_jhash", C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32])
    _sig(L, "gyo_jhash2", C.c_uint32, [u32p, C.c_uint32, C.c_uint32])
    _sig(L, "gyo_jhash_3words", C.c_uint32, [C.c_uint32] * 4)
    _sig(L, "gyo_jhash_2words", C.c_uint32, [C.c_uint32] * 3)
    _sig(L, "gyo_jhash_1word", C.c_uint32, [C.c_uint32] * 2)
    _sig(L, "gyo_get_uint64_hash", C.c_uint32, [C.c_uint64])
    _sig(L, "gyo_get_uint32_hash", C.c_uint32, [C.c_uint32])
    _sig(L, "gyo_ip_port_hash", C.c_uint32, [u8p, C.c_int, C.c_uint16, C.c_int])
    _sig(L, "gyo_ns_ip_port_hash", C.c_uint32, [u8p, C.c_int, C.c_uint16, C.c_uint64, C.c_int])
    _sig(L, "gyo_pair_ip_port_hash", C.c_uint32, [u8p, C.c_int, C.c_uint16, u8p, C.c_int, C.c_uint16])
    _sig(L, "gyo_pair_ip_port_words", C.c_uint32, [u8p, C.c_int, C.c_uint16, u8p, C.c_int, C.c_uint16, u32p])
    _sig(L, "gyo_machine_id_hash", C.c_uint32, [C.c_uint64, C.c_uint64])
    _sig(L, "gyo_hash64", C.c_uint64, [u32p, C.c_uint32])
    _sig(L, "gyo_hist_nbuckets", C.c_int, [C.c_int])
    _sig(L, "gyo_bucket", C.c_uint32, [C.c_int, C.c_int64])
    _sig(L, "gyo_bucket_many", None, [C.c_int, i64p, C.c_size_t, u32p])
    _sig(L, "gyo_bucket_max_threshold", C.c_int64, [C.c_int, C.c_size_t])
    _sig(L, "gyo_hist_init", None, [C.POINTER(Hist), C.c_int])
    _sig(L, "gyo_hist_add", C.c_uint32, [C.POINTER(Hist), C.c_int64])
    _sig(L, "gyo_hist_add_many", None, [C.POINTER(Hist), i64p, C.c_size_t])
    _sig(L, "gyo_hist_merge", None, [C.POINTER(Hist), C.POINTER(Hist)])
    _sig(L, "gyo_hist_percentiles", None, [C.POINTER(Hist), C.POINTER(HistData), C.c_size_t, u64p, i64p, f32p])
    _sig(L, "gyo_percentiles_raw", None, [C.c_int, C.POINTER(HistSerial), C.c_uint64, C.POINTER(HistData), C.c_size_t, f32p])
    _sig(L, "gyo_keyed_hist_ingest", None, [C.c_int, u32p, i32p, C.c_size_t, C.c_void_p, u64p, i64p])
    _sig(L, "gyo_conn_bitmap_add", None, [u16p, C.c_uint16, C.c_uint8])
    _sig(L, "gyo_conn_bitmap_breakup", None, [u16p, u8p])
    _sig(L, "gyo_hll_idx_rank", None, [C.c_uint64, C.c_int, u32p, u8p])
    _sig(L, "gyo_hll_add", None, [u8p, C.c_int, C.c_uint64])
    _sig(L, "gyo_hll_add_words", None, [u8p, C.c_int, u32p, C.c_uint32])
    _sig(L, "gyo_hll_merge", None, [u8p, u8p, C.c_int])
    _sig(L, "gyo_hll_estimate", C.c_double, [u8p, C.c_int])
    _sig(L, "gyo_cms_cols", None, [u32p, C.c_uint32, u32p])
    _sig(L, "gyo_cms_add", None, [u32p, u32p, C.c_uint32, C.c_uint32])
    _sig(L, "gyo_cms_query", C.c_uint32, [u32p, u32p, C.c_uint32])
    _sig(L, "gyo_cms64_add", None, [u64p, u32p, C.c_uint32, C.c_uint64])
    _sig(L, "gyo_cms64_query", C.c_uint64, [u64p, u32p, C.c_uint32])
    _sig(L, "gyo_td_init", None, [C.POINTER(TDigest)])
    _sig(L, "gyo_td_total", C.c_uint64, [C.POINTER(TDigest)])
    _sig(L, "gyo_td_cluster", C.c_uint32, [C.c_uint64, C.c_uint64])
    _sig(L, "gyo_td_merge_values", None, [C.POINTER(TDigest), i32p, C.c_size_t])
    _sig(L, "gyo_td_merge_digest", None, [C.POINTER(TDigest), C.POINTER(TDigest)])
    _sig(L, "gyo_td_quantile", C.c_double, [C.POINTER(TDigest), C.c_double])
    _sig(L, "gyo_tdb_init", None, [C.POINTER(TDBuffered)])
    _sig(L, "gyo_tdb_total", C.c_uint64, [C.POINTER(TDBuffered)])
    _sig(L, "gyo_tdb_add_batch", None, [C.POINTER(TDBuffered), i32p, C.c_size_t])
    _sig(L, "gyo_tdb_merged_view", None, [C.POINTER(TDBuffered), C.POINTER(TDigest)])
    _sig(L, "gyo_tdb_quantile", C.c_double, [C.POINTER(TDBuffered), C.c_double])
    _sig(L, "gyo_td64_init", None, [C.POINTER(TD64)])
    _sig(L, "gyo_td64_total", C.c_uint64, [C.POINTER(TD64)])
    _sig(L, "gyo_td64_merge_values", None, [C.POINTER(TD64), i32p, C.c_size_t])
    _sig(L, "gyo_td64_merge_service", None, [C.POINTER(TD64), C.POINTER(TDBuffered)])
    _sig(L, "gyo_td64_merge_td64", None, [C.POINTER(TD64), C.POINTER(TD64)])
    _sig(L, "gyo_td64_quantile", C.c_double, [C.POINTER(TD64), C.c_double])
    _sig(L, "gyo_td_value_bin", C.c_uint32, [C.c_uint32])
    _sig(L, "gyo_tdbins_init", None, [C.POINTER(TDBins)])
    _sig(L, "gyo_tdbins_add_values", None, [C.POINTER(TDBins), i32p, C.c_size_t])
    _sig(L, "gyo_tdbins_add_service", None, [C.POINTER(TDBins), C.POINTER(TDBuffered)])
    _sig(L, "gyo_tdbins_add_td64", None, [C.POINTER(TDBins), C.POINTER(TD64)])
    _sig(L, "gyo_tdbins_finish", None, [C.POINTER(TDBins), C.POINTER(TD64)])
    _sig(L, "gyo_active_conn_sketch_batch", None, [u8p, C.c_int, u32p, u64p, u64p])
    _sig(L, "gyo_active_conn_sketch_batch2", None, [u8p, C.c_int, u32p, u64p, u32p, u64p, u64p])
    _sig(L, "gyo_tcp_conn_pair_batch", C.c_int, [u8p, C.c_int, u8p, u32p, u64p, u32p, u64p])
    _sig(L, "gyo_listener_state_rollup", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(ListenSummStats), C.POINTER(C.c_int)])
    _sig(L, "gyo_listener_state_elem_size", C.c_uint32, [C.c_void_p])
    _sig(L, "gyo_tcp_conn_elem_size", C.c_uint32, [C.c_void_p])
    _sig(L, "gyo_comm_header_validate", C.c_int, [C.c_void_p, C.c_uint32])
    _sig(L, "gyo_tcp_conn_validate", C.c_int, [C.c_void_p])
    _sig(L, "gyo_listener_state_validate", C.c_int, [C.c_void_p])
    _sig(L, "gyo_tcp_conn_decode", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, u32p, u32p, u64p, u64p, u64p, u8p])
    _sig(L, "gyo_tcp_conn_sketch_batch", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, u8p, u32p, u64p])
    _sig(L, "gyo_tcp_conn_walk_tallies", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, u64p])
    _sig(L, "gyo_tcp_conn_svc_counters", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, u64p, C.c_uint32, u64p, u64p])
    _sig(L, "gyo_cluster_state_update", None, [C.POINTER(ClusterStateOne)] + [C.c_uint32] * 6 + [C.POINTER(ListenSummStats)])
    _sig(L, "gyo_cluster_state_add", None, [C.POINTER(ClusterStateOne), C.POINTER(ClusterStateOne)])
    _sig(L, "gyo_topn_u64", C.c_size_t, [u64p, C.c_size_t, C.c_size_t, u64p])
    _sig(L, "gyo_bts_init", None, [C.POINTER(BTS), C.c_uint32, C.c_int64])
    _sig(L, "gyo_bts_add", C.c_int, [C.POINTER(BTS), C.c_int64, C.c_int64, C.c_uint64])
    _sig(L, "gyo_bts_update", None, [C.POINTER(BTS), C.c_int64])
    _sig(L, "gyo_mlh_level_seconds", C.c_int64, [C.c_int])
    _sig(L, "gyo_mlh_init", None, [C.POINTER(MLHist), C.c_int, C.c_uint32])
    _sig(L, "gyo_mlh_add_hist", None, [C.POINTER(MLHist), C.c_int64, C.c_void_p, C.c_int])
    _sig(L, "gyo_mlh_flush", None, [C.POINTER(MLHist), C.c_int64])
    _sig(L, "gyo_mlh_level", None, [C.POINTER(MLHist), C.c_int, C.c_void_p])
    _sig(L, "gyo_slab_percentile_idx", C.c_size_t, [u64p, C.c_size_t, C.c_double])
    _sig(L, "gyo_mlh_get_stats", None, [C.POINTER(MLHist), C.c_int, f32p, C.c_size_t, i64p, i64p, i64p, C.POINTER(C.c_double)])
    _sig(L, "gyo_bucketid_from_threshold", C.c_uint32, [C.c_int, C.c_int64])
    _sig(L, "gyo_listener_curr_state", C.c_int, [C.POINTER(ListenerScan), C.POINTER(ListenerIssueIn), u8p, u8p, u8p])
    _sig(L, "gyo_listener_decide", None, [C.POINTER(ListenerScan), C.POINTER(ListenerIssueIn), u8p, u8p, C.POINTER(ListenerDecision)])
    _sig(L, "gyo_listener_scan_one", None, [C.POINTER(MLHist), C.POINTER(Hist), C.POINTER(Hist), u16p, C.c_uint64, C.c_float, C.c_int64, u8p,
                                            C.POINTER(ListenerScan)])
    _sig(L, "gyo_mlh_level_for_start", C.c_int, [C.POINTER(MLHist), C.c_int, C.c_int64])
    _sig(L, "gyo_mlh_period", None, [C.POINTER(MLHist), C.c_int64, C.c_int64, C.c_void_p])
    _sig(L, "gyo_mlh_get_stats_for_period", None, [C.POINTER(MLHist), C.c_int64, C.c_int64, f32p, C.c_size_t, i64p, i64p, i64p, C.POINTER(C.c_double)])
    _sig(L, "gyo_engine_new", C.c_void_p, [C.c_uint32, C.c_int])
    _sig(L, "gyo_engine_new_cap", C.c_void_p, [C.c_uint32, C.c_int, C.c_uint32])
    _sig(L, "gyo_tdb_init_cap", None, [C.POINTER(TDBuffered), C.c_uint32])
    _sig(L, "gyo_tdb_free", None, [C.POINTER(TDBuffered)])
    _sig(L, "gyo_tdb_values", C.POINTER(C.c_int32), [C.POINTER(TDBuffered)])
    _sig(L, "gyo_engine_free", None, [C.c_void_p])
    _sig(L, "gyo_engine_register", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint16])
    _sig(L, "gyo_engine_resp_batch", None, [C.c_void_p, C.c_void_p, C.c_uint64, u32p, u64p, C.c_uint32])
    _sig(L, "gyo_engine_register_bulk", C.c_int, [C.c_void_p, C.c_uint32, u64p, u32p, C.POINTER(C.c_uint16), C.c_uint32])
    _sig(L, "gyo_engine_register_addr", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint16, u8p, C.c_int, C.c_int])
    _sig(L, "gyo_engine_resp_batch_v6", None, [C.c_void_p, C.c_void_p, C.c_uint64, u32p, u64p, C.c_uint32])
    _sig(L, "gyo_ip_norm", C.c_int, [u8p, C.c_int, u32p, u8p])
    _sig(L, "gyo_ip_equal", C.c_int, [C.c_uint32, u8p, C.c_uint32, u8p])
    _sig(L, "gyo_conn_bitmap_breakup2", None, [u16p, u8p])
    _sig(L, "gyo_engine_resp_batch_histonly", None, [C.c_void_p, C.c_void_p, C.c_uint64, u32p, u64p, C.c_uint32])
    _sig(L, "gyo_engine_resp_batch_mt", None, [C.c_void_p, C.c_void_p, C.c_uint64, u32p, u64p, C.c_uint32, C.c_uint32])
    _sig(L, "gyo_td_stress", None, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.POINTER(C.c_double)])
    _sig(L, "gyo_engine_nsvc", C.c_uint32, [C.c_void_p])
    _sig(L, "gyo_engine_hist", C.c_void_p, [C.c_void_p])
    _sig(L, "gyo_engine_bitmap", C.c_void_p, [C.c_void_p])
    _sig(L, "gyo_engine_hll", C.c_void_p, [C.c_void_p])
    _sig(L, "gyo_engine_cms", C.c_void_p, [C.c_void_p])
    _sig(L, "gyo_engine_ghist", C.c_void_p, [C.c_void_p])
    _sig(L, "gyo_engine_gmax", C.c_int64, [C.c_void_p])
    _sig(L, "gyo_engine_td", C.POINTER(TDBuffered), [C.c_void_p, C.c_uint32])
    _sig(L, "gyo_engine_counters", u64p, [C.c_void_p])
    _sig(L, "gyo_engine_window_clear", None, [C.c_void_p, C.c_int])
    _lib = L
    return L


_ref = None


def ref():
    """The reference's own code (None when oracle/_ref has not been built, e.g. no /root/reference)."""
    global _ref
    if _ref is not None:
        return _ref
    if not os.path.exists(REF_PATH):
        return None
    R = C.CDLL(REF_PATH)
    _sig(R, "ref_jhash", C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32])
    _sig(R, "ref_jhash2", C.c_uint32, [u32p, C.c_uint32, C.c_uint32])
    _sig(R, "ref_jhash_3words", C.c_uint32, [C.c_uint32] * 4)
    _sig(R, "ref_jhash_2words", C.c_uint32, [C.c_uint32] * 3)
    _sig(R, "ref_jhash_1word", C.c_uint32, [C.c_uint32] * 2)
    _sig(R, "ref_get_uint64_hash", C.c_uint32, [C.c_uint64])
    _sig(R, "ref_get_uint32_hash", C.c_uint32, [C.c_uint32])
    _sig(R, "ref_ip_port_hash", C.c_uint32, [u8p, C.c_int, C.c_uint16, C.c_int])
    _sig(R, "ref_ns_ip_port_hash", C.c_uint32, [u8p, C.c_int, C.c_uint16, C.c_uint64, C.c_int])
    _sig(R, "ref_pair_ip_port_hash", C.c_uint32, [u8p, C.c_int, C.c_uint16, u8p, C.c_int, C.c_uint16])
    _sig(R, "ref_machine_id_hash", C.c_uint32, [C.c_uint64, C.c_uint64])
    if hasattr(R, "ref_ip_addr_norm"):
        _sig(R, "ref_ip_addr_norm", C.c_int, [u8p, C.c_int, u32p, u8p, u8p, u32p])
        _sig(R, "ref_ip_addr_equal", C.c_int, [u8p, C.c_int, u8p, C.c_int])
        _sig(R, "ref_listener_match", C.c_int, [u8p, C.c_int, C.c_uint16, C.c_uint64, C.c_int, u8p, C.c_int, C.c_uint16, C.c_uint64])
    if hasattr(R, "ref_has_summ_stats"):
        _sig(R, "ref_has_summ_stats", C.c_int, [])
        if R.ref_has_summ_stats():
            _sig(R, "ref_listen_summ_update", None, [C.c_void_p, C.c_int, C.POINTER(C.c_int32)])
            _sig(R, "ref_cluster_state_update", None, [u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int32)])
    if hasattr(R, "ref_resp_bucketid_from_threshold"):
        _sig(R, "ref_resp_bucketid_from_threshold", C.c_size_t, [C.c_int64])
    if hasattr(R, "ref_comm_sizeof"):
        _sig(R, "ref_comm_nfields", C.c_int, [])
        _sig(R, "ref_comm_field_name", C.c_char_p, [C.c_int])
        _sig(R, "ref_comm_field_offset", C.c_size_t, [C.c_int])
        _sig(R, "ref_comm_sizeof", C.c_size_t, [C.c_int])
        _sig(R, "ref_comm_const", C.c_uint64, [C.c_int])
        _sig(R, "ref_comm_hdr_validate", C.c_int, [C.c_void_p, C.c_uint32])
        _sig(R, "ref_tcp_conn_validate", C.c_int, [C.c_void_p])
        _sig(R, "ref_listener_state_validate", C.c_int, [C.c_void_p])
        _sig(R, "ref_tcp_conn_elem_size", C.c_uint32, [C.c_void_p])
        _sig(R, "ref_state_one_sizeof", C.c_size_t, [])
        _sig(R, "ref_state_one_add", None, [u32p, u32p])
        _sig(R, "ref_state_one_fields", None, [u32p, u32p])
        _sig(R, "ref_listener_state_elem_size", C.c_uint32, [C.c_void_p])
    if hasattr(R, "ref_slab_percentile_idx"):
        _sig(R, "ref_slab_num_buckets", C.c_size_t, [])
        _sig(R, "ref_slab_bucket_idx", C.c_size_t, [C.c_int64])
        _sig(R, "ref_slab_percentile_idx", C.c_size_t, [u64p, C.c_size_t, C.c_double])
    _sig(R, "ref_sizeof", C.c_size_t, [C.c_int])
    _sig(R, "ref_ip_port_bytes", None, [u8p, C.c_int, C.c_uint16, u8p])
    _sig(R, "ref_hist_new", C.c_void_p, [C.c_int])
    _sig(R, "ref_hist_free", None, [C.c_void_p])
    _sig(R, "ref_hist_nbuckets", C.c_size_t, [C.c_void_p])
    _sig(R, "ref_hist_object_size", C.c_size_t, [C.c_void_p])
    _sig(R, "ref_hist_add", C.c_size_t, [C.c_void_p, C.c_int64])
    _sig(R, "ref_hist_add_many", None, [C.c_void_p, i64p, C.c_size_t])
    _sig(R, "ref_hist_bucket_of", C.c_size_t, [C.c_void_p, C.c_int64])
    _sig(R, "ref_hist_bucket_of_many", None, [C.c_void_p, i64p, C.c_size_t, u32p])
    _sig(R, "ref_hist_bucket_max_threshold", C.c_int64, [C.c_void_p, C.c_size_t])
    _sig(R, "ref_hist_percentiles", None, [C.c_void_p, f32p, C.c_size_t, i64p, i64p, u64p, u64p, i64p, f32p])
    _sig(R, "ref_hist_serialized", None, [C.c_void_p, u64p, i64p, u64p, i64p])
    _sig(R, "ref_hist_merge", None, [C.c_void_p, C.c_void_p])
    _sig(R, "ref_hist_clear", None, [C.c_void_p])
    _sig(R, "ref_topn_u64", C.c_size_t, [u64p, C.c_size_t, C.c_size_t, u64p])
    if hasattr(R, "ref_keyed_new"):  # present in builds of oracle/ref_glue.cc that carry the keyed CPU-baseline loop
        _sig(R, "ref_keyed_new", C.c_void_p, [])
        _sig(R, "ref_keyed_free", None, [C.c_void_p])
        _sig(R, "ref_keyed_register", C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint16])
        if hasattr(R, "ref_keyed_register_bulk"):
            _sig(R, "ref_keyed_register_bulk", None, [C.c_void_p, C.c_uint32, u32p, C.POINTER(C.c_uint16), C.c_uint32])
        _sig(R, "ref_keyed_resp_batch", C.c_uint64, [C.c_void_p, C.c_void_p, C.c_uint64, u32p, u64p, C.c_uint32])
        _sig(R, "ref_keyed_total", C.c_uint64, [C.c_void_p, C.c_uint32])
        if hasattr(R, "ref_keyed_resp_batch_mt"):
            _sig(R, "ref_keyed_resp_batch_mt", C.c_uint64, [C.c_void_p, C.c_void_p, C.c_uint64, u32p, u64p, C.c_uint32, C.c_uint32])
    if hasattr(R, "ref_conn_exact_new"):  # the exact connection table (PAIR_IP_PORT set + per-listener counters) behind the C2 CPU baseline
        _sig(R, "ref_conn_exact_new", C.c_void_p, [])
        _sig(R, "ref_conn_exact_free", None, [C.c_void_p])
        _sig(R, "ref_conn_exact_batch", C.c_uint64, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p])
        _sig(R, "ref_conn_exact_batch_mt", C.c_uint64, [C.c_void_p, u64p, C.c_uint32, C.c_uint64, C.c_uint32, u64p, u64p])
        _sig(R, "ref_conn_exact_distinct", C.c_uint64, [C.c_void_p])
        _sig(R, "ref_conn_exact_services", C.c_uint64, [C.c_void_p])
        _sig(R, "ref_conn_exact_get", C.c_int, [C.c_void_p, C.c_uint64, u64p])
    _ref = R
    return R


# ------------------------------------------------------------------ numpy conveniences

def ptr(a, t):
    return a.ctypes.data_as(t)


def ip_bytes(ip):
    """ip: int (ipv4, value whose little-endian bytes are the network-order address, i.e. ip32_be) or 16 raw bytes."""
    if isinstance(ip, (bytes, bytearray)):
        assert len(ip) == 16
        return (C.c_uint8 * 16)(*ip), 1
    return (C.c_uint8 * 4)(*int(ip).to_bytes(4, "little")), 0


def hash64_words(words):
    w = np.ascontiguousarray(words, dtype=np.uint32)
    return lib().gyo_hash64(ptr(w, u32p), len(w))


def glob_id_words(gid):
    gid = int(gid)
    return np.array([gid & 0xFFFFFFFF, gid >> 32], dtype=np.uint32)


def hist_percentiles(kind, stats, total, pcts):
    """stats: structured/2-col array [[count,sum]...] (>= nbuckets rows).  Returns (values, sums, counts, avg)."""
    L = lib()
    arr = (HistSerial * MAX_BUCKETS)()
    for i in range(L.gyo_hist_nbuckets(kind)):
        arr[i].count = int(stats[i][0])
        arr[i].sum = int(stats[i][1])
    pd = (HistData * len(pcts))()
    for i, p in enumerate(pcts):
        pd[i].percentile = p
    avg = C.c_float(0)
    L.gyo_percentiles_raw(kind, arr, int(total), pd, len(pcts), C.byref(avg))
    return [d.data_value for d in pd], [d.sum for d in pd], [d.count for d in pd], avg.value


def keyed_hist(kind, nkeys, keyidx, vals):
    L = lib()
    stats = np.zeros((nkeys, MAX_BUCKETS, 2), dtype=np.int64)  # [..,0]=count (as i64 bits) [..,1]=sum
    total = np.zeros(nkeys, dtype=np.uint64)
    maxv = np.full(nkeys, np.iinfo(np.int64).min if kind in (0,) else (np.iinfo(np.int32).min if kind not in (8,) else -128),
                   dtype=np.int64)
    k = np.ascontiguousarray(keyidx, dtype=np.uint32)
    v = np.ascontiguousarray(vals, dtype=np.int32)
    L.gyo_keyed_hist_ingest(kind, ptr(k, u32p), ptr(v, i32p), len(k), stats.ctypes.data, ptr(total, u64p), ptr(maxv, i64p))
    return stats, total, maxv


def td_from_arrays(sums, cnts, vmin, vmax):
    d = TDigest()
    for i in range(TD_NB):
        d.sum[i] = int(sums[i])
        d.cnt[i] = int(cnts[i])
    d.vmin = int(vmin)
    d.vmax = int(vmax)
    return d


def td_to_arrays(d):
    return (np.array(list(d.sum), dtype=np.int64), np.array(list(d.cnt), dtype=np.uint32), d.vmin, d.vmax)


class OracleEngine:
    """python handle on the sequential CPU restatement of the response-event hot path (gy_oracle_engine.c)"""

    def __init__(self, max_services, enable_td=True, td_cap=0):
        self.L = lib()
        self.td_cap = td_cap or TD_PEND_CAP
        self.h = self.L.gyo_engine_new_cap(max_services, 1 if enable_td else 0, td_cap)
        self.enable_td = enable_td

    def __del__(self):
        if getattr(self, "h", None):
            self.L.gyo_engine_free(self.h)
            self.h = None

    def register(self, host_slot, glob_id, netns, port):
        s = self.L.gyo_engine_register(self.h, int(host_slot), int(glob_id), int(netns), int(port))
        assert s >= 0
        return s

    def register_bulk(self, host_slot, glob_ids, netns, ports):
        """n any-address listeners of one host in one call"""
        g = np.ascontiguousarray(glob_ids, dtype=np.uint64)
        ns = np.ascontiguousarray(netns, dtype=np.uint32)
        pt = np.ascontiguousarray(ports, dtype=np.uint16)
        first = self.L.gyo_engine_register_bulk(self.h, int(host_slot), ptr(g, u64p), ptr(ns, u32p), pt.ctypes.data_as(C.POINTER(C.c_uint16)), len(g))
        assert first >= 0
        return first

    def register_addr(self, host_slot, glob_id, netns, port, addr=None, is_v6=False):
        """addr: None = an any-address listener (is_any_ip_), else the 4 / 16 address bytes the listener is bound to"""
        buf = (C.c_uint8 * 16)(*(bytes(addr) + bytes(16))[:16]) if addr is not None else (C.c_uint8 * 16)()
        s = self.L.gyo_engine_register_addr(self.h, int(host_slot), int(glob_id), int(netns), int(port), buf, int(bool(is_v6)), int(addr is None))
        assert s >= 0
        return s

    def resp_batch_v6(self, ev_bytes, seg_host, seg_first):
        """48-byte tcp_ipv6_resp_event_t events"""
        ev = np.frombuffer(ev_bytes, dtype=np.uint8)
        sh = np.ascontiguousarray(seg_host, dtype=np.uint32)
        sf = np.ascontiguousarray(seg_first, dtype=np.uint64)
        self.L.gyo_engine_resp_batch_v6(self.h, ev.ctypes.data, len(ev) // 48, ptr(sh, u32p), ptr(sf, u64p), len(sh))

    def resp_batch(self, ev_bytes, seg_host, seg_first, histonly=False, nthreads=1):
        """nthreads > 1: the same batch with the segments (distinct hosts) cut into per-thread ranges; identical resulting state"""
        ev = np.frombuffer(ev_bytes, dtype=np.uint8)
        sh = np.ascontiguousarray(seg_host, dtype=np.uint32)
        sf = np.ascontiguousarray(seg_first, dtype=np.uint64)
        if nthreads > 1 and not histonly:
            self.L.gyo_engine_resp_batch_mt(self.h, ev.ctypes.data, len(ev) // 24, ptr(sh, u32p), ptr(sf, u64p), len(sh), nthreads)
            return
        fn = self.L.gyo_engine_resp_batch_histonly if histonly else self.L.gyo_engine_resp_batch
        fn(self.h, ev.ctypes.data, len(ev) // 24, ptr(sh, u32p), ptr(sf, u64p), len(sh))

    @property
    def nsvc(self):
        return self.L.gyo_engine_nsvc(self.h)

    def _arr(self, p, dtype, shape):
        n = int(np.prod(shape))
        buf = (C.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()

    def hist(self):
        return self._arr(self.L.gyo_engine_hist(self.h), np.int64, (self.nsvc, 16, 2))

    def bitmap(self):
        return self._arr(self.L.gyo_engine_bitmap(self.h), np.uint16, (self.nsvc, 64))  # rows 0..31 resp_bitmap_v4_, 32..63 resp_bitmap_v6_

    def hll(self):
        return self._arr(self.L.gyo_engine_hll(self.h), np.uint8, (1 << HLL_P,))

    def cms(self):
        return self._arr(self.L.gyo_engine_cms(self.h), np.uint32, (CMS_D, CMS_W))

    def ghist(self):
        return self._arr(self.L.gyo_engine_ghist(self.h), np.int64, (16, 2)), self.L.gyo_engine_gmax(self.h)

    def td(self, slot):
        return self.L.gyo_engine_td(self.h, slot).contents

    def td_arrays(self):
        """merged clusters + min/max of every service: (sums [n][100], cnts [n][100], minmax [n][2])"""
        n = self.nsvc
        sums = np.zeros((n, TD_NB), dtype=np.int64)
        cnts = np.zeros((n, TD_NB), dtype=np.uint32)
        mm = np.zeros((n, 2), dtype=np.int32)
        for s in range(n):
            d = self.td(s).d
            sums[s] = np.frombuffer(d.sum, dtype=np.int64)
            cnts[s] = np.frombuffer(d.cnt, dtype=np.uint32)
            mm[s] = (d.vmin, d.vmax)
        return sums, cnts, mm

    def td_pending(self):
        """buffered (unmerged) values of every service, each row sorted ascending and padded with -1: (npend [n], pend [n][CAP])"""
        n = self.nsvc
        npend = np.zeros(n, dtype=np.uint32)
        pend = np.full((n, self.td_cap), -1, dtype=np.int32)
        for s in range(n):
            b = self.td(s)
            npend[s] = b.npend
            if b.npend:
                src = self.L.gyo_tdb_values(C.byref(b))
                pend[s, :b.npend] = np.sort(np.ctypeslib.as_array(src, shape=(b.npend,)))
        return npend, pend

    def counters(self):
        c = self.L.gyo_engine_counters(self.h)
        return {"events": c[0], "dropped_range": c[1], "dropped_nolistener": c[2], "accepted": c[3]}

    def window_clear(self, clear_hist=False):
        self.L.gyo_engine_window_clear(self.h, 1 if clear_hist else 0)


def rollup_services(tds):
    """the roll-up digest (TD64) of a group of services given as TDBuffered objects: gyo_tdbins_* (union by value bin; any order)"""
    L = lib()
    b, out = TDBins(), TD64()
    L.gyo_tdbins_init(C.byref(b))
    for t in tds:
        L.gyo_tdbins_add_service(C.byref(b), C.byref(t))
    L.gyo_tdbins_finish(C.byref(b), C.byref(out))
    return out


def rollup_slabs(slabs):
    """the roll-up digest (TD64) of a group of roll-up digests (TD64)"""
    L = lib()
    b, out = TDBins(), TD64()
    L.gyo_tdbins_init(C.byref(b))
    for t in slabs:
        L.gyo_tdbins_add_td64(C.byref(b), C.byref(t))
    L.gyo_tdbins_finish(C.byref(b), C.byref(out))
    return out

"""TCP_LISTENER::get_curr_state (common/gy_socket_stat.cc:2020-2870) as restated by oracle/gy_oracle_lstate.c: one hand-derived case per
`return` of the reference (read off the reference's branches, not off the restatement), the arithmetic quirks the reference has
(32-bit `ser_errors * 2`, the overwritten ISSUE_SERVER_ERRORS at :2406-2417, float / double products) and the caller's part
(:4241-4266).  The GPU decision (k_listener_decide) is compared with this oracle in tests/test_gpu_round6.py."""
import ctypes as C

import pytest

IDLE, GOOD, OK, BAD, SEVERE = 0, 1, 2, 3, 4
NONE, TASKS, QPS_HIGH, ACTIVE_HIGH, SERVER_ERRORS, DEPENDS, UNKNOWN = 0, 1, 2, 3, 4, 7, 8


def mk(oracle, **kw):
    """a listener in an ordinary state: 100 queries in the last 5 s at 5 ms mean, every level's p95 = 10 ms, QPS 20 between p25 = 5 and p95 = 50,
    5 active connections (p25 2, p95 20), 10 connections, no errors, no task issue; keyword arguments change fields"""
    L = oracle.lib()
    sc = oracle.ListenerScan()
    d = dict(tcount=[100, 6000, 1_000_000, 2_000_000], mean=[5.0, 5.0, 5.0, 5.0], p95=[10, 10, 10, 10], p99=[30, 30, 30, 30], curr_qps=20, qps_p25=5, qps_p95=50,
             act_p25=2, act_p95=20, nconn_active=5, nactive=[0] * 15, ser_errors=0, flags=0, nconn=10, ntasks_issue=0, ntasks_noissue=0, delay=0, cpudelay=0,
             blkdelay=0, tdiff_start=0, bithist=0)
    d.update(kw)
    for i in range(4):
        sc.tcount[i] = d["tcount"][i]
        sc.tsum[i] = int(round(d["tcount"][i] * d["mean"][i]))
        sc.p95_ms[i] = d["p95"][i]
        sc.p99_ms[i] = d["p99"][i]
    sc.curr_qps, sc.qps_p25, sc.qps_p95, sc.act_p25, sc.act_p95, sc.nconn_active = d["curr_qps"], d["qps_p25"], d["qps_p95"], d["act_p25"], d["act_p95"], d["nconn_active"]
    sc.b5, sc.b300, sc.b5day = (L.gyo_bucketid_from_threshold(oracle.RESP_TIME_HASH, d["p95"][i]) for i in range(3))  # :2085-2087
    for i in range(15):
        sc.nactive_conn_arr[i] = d["nactive"][i]
    inp = oracle.ListenerIssueIn(ser_errors=d["ser_errors"], tasks_delay_msec=d["delay"], tasks_cpudelay_msec=d["cpudelay"], tasks_blkiodelay_msec=d["blkdelay"],
                                 nconn=d["nconn"], ntasks_issue=d["ntasks_issue"], ntasks_noissue=d["ntasks_noissue"], flags=d["flags"], tdiff_start=d["tdiff_start"])
    return sc, inp, d["bithist"]


def decide(oracle, **kw):
    sc, inp, bh = mk(oracle, **kw)
    hb = (C.c_uint8 * 1)(bh)
    st, isx = (C.c_uint8 * 1)(), (C.c_uint8 * 1)()
    line = oracle.lib().gyo_listener_curr_state(C.byref(sc), C.byref(inp), hb, st, isx)
    return st[0], isx[0], line, hb[0]


def test_one_case_per_return_of_the_reference(oracle):
    T, S, D, CPU, MEM, DEP = oracle.LI_TASK_ISSUE, oracle.LI_SEVERE, oracle.LI_DELAY, oracle.LI_CPU_ISSUE, oracle.LI_MEM_ISSUE, oracle.LI_DEPENDS
    fast = dict(p95=[1, 10, 10, 10])            # 5-s p95 in the 1-ms bucket (:2132 first arm)
    fastlow = dict(fast, curr_qps=3)            # ... and the QPS at or below its p25 (:2136)
    lower = dict(p95=[10, 10, 30, 30])          # 5-s p95 below the 5-day p95 (:2132 second arm)
    eqlow = dict(mean=[3.0, 5.0, 5.0, 5.0])     # equal p95s, the 5-s mean <= 0.8 x the 5-day mean (:2340)
    high = dict(p95=[30, 10, 10, 10])           # 5-s p95 one bucket above the 5-day one
    high2 = dict(p95=[30, 30, 10, 10])          # ... and the 5-min p95 as high (b300 != b5day)
    vhigh = dict(p95=[100, 30, 10, 10])         # b5 = 5 > b5day + 2 = 4 and > b300 = 3
    cases = [
        (dict(curr_qps=0, tcount=[0, 6000, 10**6, 2 * 10**6]), (IDLE, NONE, 2126)),
        (dict(curr_qps=0, tcount=[0, 6000, 10**6, 2 * 10**6], flags=T | S, ser_errors=5), (SEVERE, SERVER_ERRORS, 2322)),  # :2116 does not return: 10 > 0 queries
        (fastlow, (IDLE, NONE, 2144)),
        (dict(fastlow, ser_errors=60), (SEVERE, SERVER_ERRORS, 2153)),
        (dict(fastlow, ser_errors=30), (BAD, SERVER_ERRORS, 2161)),
        (dict(fastlow, ser_errors=5), (OK, SERVER_ERRORS, 2169)),
        (dict(fastlow, ser_errors=15), (OK, SERVER_ERRORS, 2305)),          # 15 >= 0.1 x 100: no arm of :2147-2171 returns; 75 <= 100 at :2245
        (dict(fastlow, ser_errors=60, flags=T), (SEVERE, SERVER_ERRORS, 2179)),
        (dict(fastlow, ser_errors=30, flags=T), (BAD, SERVER_ERRORS, 2187)),
        (dict(fastlow, ser_errors=5, flags=T), (BAD, TASKS, 2202)),
        (dict(fastlow, flags=T | S, ntasks_issue=2), (BAD, TASKS, 2213)),
        (dict(fastlow, flags=T), (OK, TASKS, 2224)),                       # 10 connections > p25 of the active connections
        (dict(fastlow, flags=T, nconn=1), (GOOD, NONE, 2305)),
        (fast, (GOOD, NONE, 2305)),
        (dict(fast, curr_qps=80), (OK, QPS_HIGH, 2305)),                   # b5 + 2 = 3 > b5day = 2
        (dict(p95=[1, 10, 100, 100], curr_qps=80), (GOOD, NONE, 2305)),    # "extremely low response time with high QPS": 3 <= 5
        (dict(lower, ser_errors=60), (SEVERE, SERVER_ERRORS, 2243)),
        (dict(lower, ser_errors=30), (BAD, SERVER_ERRORS, 2257)),
        (dict(lower, flags=T | S, ntasks_issue=1), (BAD, TASKS, 2273)),
        (dict(lower, ser_errors=5), (OK, SERVER_ERRORS, 2305)),
        (dict(ser_errors=60), (SEVERE, SERVER_ERRORS, 2322)),
        (dict(ser_errors=30), (BAD, SERVER_ERRORS, 2336)),
        (dict(eqlow, curr_qps=3, ser_errors=5), (BAD, SERVER_ERRORS, 2356)),
        (dict(eqlow, curr_qps=3), (IDLE, NONE, 2364)),
        (dict(eqlow, curr_qps=3, flags=T, ntasks_issue=1), (BAD, TASKS, 2374)),
        (dict(eqlow, curr_qps=3, flags=T, ntasks_issue=1, ntasks_noissue=1, delay=1500), (BAD, TASKS, 2384)),
        (eqlow, (GOOD, NONE, 2394)),
        (dict(eqlow, ser_errors=5, flags=T), (BAD, TASKS, 2403)),
        (dict(eqlow, ser_errors=5), (OK, TASKS, 2417)),                    # the reference sets ISSUE_SERVER_ERRORS at :2406 and overwrites it at :2412
        (dict(eqlow, flags=T), (OK, TASKS, 2417)),
        (dict(mean=[5.9, 5.0, 5.0, 5.0]), (OK, NONE, 2427)),
        (dict(mean=[7.0, 5.0, 5.0, 5.0]), (OK, NONE, 2702)),               # above 1.2 x: on to the "higher" part; b300 == b5day, 5-min mean not high
        (dict(mean=[7.0, 5.0, 5.0, 5.0], p99=[60, 30, 30, 30]), (OK, NONE, 2571)),
        (dict(mean=[7.0, 5.0, 5.0, 5.0], p99=[60, 30, 30, 30], ser_errors=5), (OK, SERVER_ERRORS, 2571)),
        (dict(high, ser_errors=60), (SEVERE, SERVER_ERRORS, 2447)),
        (dict(high, ser_errors=30), (BAD, SERVER_ERRORS, 2461)),
        (dict(high, curr_qps=80), (BAD, QPS_HIGH, 2492)),
        (dict(high, curr_qps=55), (OK, NONE, 2768)),                       # 55 - 50 = 5 is not > 5 (then: the 5-s mean is not above the 5-min mean, one high iteration)
        (dict(vhigh, curr_qps=80), (SEVERE, QPS_HIGH, 2492)),
        (dict(high, flags=T), (BAD, TASKS, 2525)),
        (dict(high, flags=D, ntasks_issue=2, ntasks_noissue=1, delay=200), (BAD, TASKS, 2525)),  # 800 ms of delay > 500 ms of response time
        (dict(vhigh, flags=T), (SEVERE, TASKS, 2525)),
        (dict(high, nconn_active=30), (BAD, ACTIVE_HIGH, 2552)),
        (dict(high, nconn_active=21), (OK, NONE, 2738)),                   # 21 - 20 is not > 1 (then: 21 active connections, none of them in the slow buckets)
        (dict(vhigh, nconn_active=30), (SEVERE, ACTIVE_HIGH, 2552)),
        (dict(vhigh, nconn_active=9, act_p95=5), (BAD, ACTIVE_HIGH, 2552)),  # severe needs more than 10 active connections
        (dict(high, curr_qps=3, nconn=1, flags=D | CPU | MEM), (BAD, TASKS, 2593)),
        (dict(high, curr_qps=3, nconn=1, flags=D | CPU, delay=200), (BAD, TASKS, 2611)),
        (dict(high, curr_qps=3, nconn=1, flags=D | CPU, delay=100), (OK, NONE, 2630)),
        (dict(high, curr_qps=3, nconn=1, ser_errors=5), (OK, SERVER_ERRORS, 2630)),
        (dict(p95=[30, 30, 10, 30], mean=[7.0, 7.0, 5.0, 10.0]), (OK, NONE, 2657)),                       # 2 queries/s over 5 days < 20 / 2
        (dict(p95=[30, 30, 10, 30], mean=[7.0, 7.0, 5.0, 10.0], tdiff_start=1000, curr_qps=4000, qps_p95=10**6), (OK, NONE, 2427 if False else 2657)),  # 1000 queries/s over the 1000 s the histogram covers < 2000
        (dict(high2, curr_qps=3, nconn_active=1), (OK, NONE, 2679)),
        (dict(high, mean=[7.0, 5.0, 5.0, 5.0]), (OK, NONE, 2702)),         # transient: the last 5 minutes are not high
        (high, (OK, NONE, 2768)),
        (dict(high2, nconn_active=16, nactive=[0, 0, 0, 2, 5] + [0] * 10), (OK, NONE, 2738)),
        (dict(high2, nconn_active=16, nactive=[0, 0, 0, 5, 5] + [0] * 10), (OK, NONE, 2768)),            # bucket b5 itself has more than 3 connections; 1 high iteration
        (dict(high2, bithist=0x0F), (OK, NONE, 2768)),                     # 0x1F after the shift: 5 high iterations ... see below
        (dict(high2, bithist=0xFF), (BAD, UNKNOWN, 2866)),
        (dict(high2, bithist=0xFF, ser_errors=5), (BAD, SERVER_ERRORS, 2866)),
        (dict(high2, bithist=0xFF, flags=DEP), (BAD, DEPENDS, 2866)),
        (dict(high2, bithist=0xFF, delay=200), (BAD, TASKS, 2817)),
        (dict(high2, bithist=0xFF, delay=60), (BAD, TASKS, 2853)),
        (dict(vhigh, bithist=0xFF), (SEVERE, UNKNOWN, 2866)),
        (dict(vhigh, bithist=0xFF, delay=200), (SEVERE, TASKS, 2853)),     # :2796 takes STATE_BAD only
    ]
    seen = set()
    for i, (kw, want) in enumerate(cases):
        st, isx, line, _ = decide(oracle, **kw)
        if kw.get("bithist") == 0x0F and kw.get("p95") == [30, 30, 10, 10] and len(kw) == 2:
            want = (BAD, UNKNOWN, 2866)  # 0x0F << 1 | 1 = 0x1F: five high iterations are not "< 5"
        assert (st, isx, line) == want, f"case {i} {kw}: got state {st} issue {isx} at :{line}, want {want}"
        seen.add(line)
    assert len(seen) >= 38  # every deciding line of the reference is exercised


def test_reference_arithmetic_quirks(oracle):
    # `ser_errors * 2` is a 32-bit product (uint32_t x int): 2^31 errors wrap to 0 and are not "> nqrys_5s" (:2151, :2241 ...)
    st, isx, line, _ = decide(oracle, p95=[1, 10, 10, 10], curr_qps=3, ser_errors=1 << 31)
    assert (st, isx, line) == (BAD, SERVER_ERRORS, 2161)  # x 2 wraps to 0, x 5 wraps to 2^31 > 100
    # `curr_qps > p95 x 1.1f` is a float compare (:2466): 55 > 50 x 1.1f = 55.000004 is false
    st, isx, line, _ = decide(oracle, p95=[30, 10, 10, 10], curr_qps=56, qps_p95=50)
    assert (isx, line) == (QPS_HIGH, 2492)
    st, isx, line, _ = decide(oracle, p95=[30, 10, 10, 10], curr_qps=1100, qps_p95=1000)
    assert line != 2492  # 1100 > 1000 x 1.1f = 1100.00002 is false
    # the history byte: shifted on every call (:2113), the low bit set only past :2431
    assert decide(oracle, bithist=0x81)[3] == 0x02
    assert decide(oracle, p95=[30, 10, 10, 10], bithist=0x81)[3] == 0x03


def test_callers_part_issue_history_and_young_listeners(oracle):
    L = oracle.lib()
    sc, inp, _ = mk(oracle, p95=[30, 30, 10, 10], bithist=0)
    ib, hb = (C.c_uint8 * 1)(0x40), (C.c_uint8 * 1)(0xFF)
    out = oracle.ListenerDecision()
    L.gyo_listener_decide(C.byref(sc), C.byref(inp), ib, hb, C.byref(out))
    assert (out.state, out.issue, out.decided_line, out.issue_bit_hist, out.high_resp_bit_hist) == (BAD, UNKNOWN, 2866, 0x81, 0xFF) and ib[0] == 0x81
    inp.flags = oracle.LI_YOUNG  # started less than 100 s ago and no errors: "No status possible currently" (:4255-4262)
    L.gyo_listener_decide(C.byref(sc), C.byref(inp), ib, hb, C.byref(out))
    assert (out.state, out.issue, out.decided_line, out.issue_bit_hist) == (OK, NONE, 4262, 0)
    inp.ser_errors = 5  # ... with errors the decision stands (:4244)
    ib[0] = 0
    L.gyo_listener_decide(C.byref(sc), C.byref(inp), ib, hb, C.byref(out))
    assert (out.state, out.issue, out.issue_bit_hist) == (BAD, SERVER_ERRORS, 1)

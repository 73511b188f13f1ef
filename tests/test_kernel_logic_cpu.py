"""Kernel LOGIC without a GPU: gyeeta_amd/csrc/gys_kernels.hpp compiled by g++ against a small CPU stand-in of the HIP device model
(tests/cpp/kemu/hip/hip_runtime.h: one OS thread per GPU thread, barriers for __syncthreads and the wave64 exchanges, host atomics) and
the hot kernels run on synthetic input:
  * k_digest_bins on hand-made keys (tests/cpp/kemu/test_bins.cc): re-clustered digests, lazily folded histogram records, CONN_BITMAP
    rows, min / max and drained meta records equal the oracle's;
  * the response-event pipeline k_resp_host (+ finalize_key) -> k_digest_bins / k_digest_merge over several batches and window
    boundaries (tests/cpp/kemu/test_resp.cc) in both tile forms and in the split form (long segments cut into parts of 65 536 events,
    several workgroups per host reserving buffer space with device atomics, k_key_finalize as its own launch): counters, HLL
    registers, all-service histogram, every key's buffered values and digest and the records of re-clustered keys equal the oracle's
    sequential engine fed the same bytes; the same with bound-address listeners and several listeners per (netns, port) key (KEMU_MODE=1:
    the event's server address picks the listener, common/gy_socket_stat.h:708-714) and with batches of 48-byte IPv6 events in between
    (KEMU_MODE=2: handle_ipv6_resp_event, resp_bitmap_v6_ rows, addresses that embed an IPv4 one), window records and CONN_BITMAP rows of both
    families against an oracle engine cleared at the window boundaries;
  * the paths of a key whose batch does not fit its buffer (tests/cpp/kemu/test_spill.cc): spill in finalize_key, the SPILL pass of
    k_resp_host, merges from buffer + run in every size class, the several-workgroup path of gys_huge.hpp with its sorted tail and its
    one-workgroup fallback;
  * the TCP_CONN_NOTIFY roll-up k_conn_ingest + k_conn_fold (tests/cpp/kemu/test_conn.cc): variable-stride v4 / v6 records read as
    16-byte pieces through LDS, per-workgroup aggregation of the service accumulators, record counts around the wave / round /
    workgroup boundaries, known and unknown services, with and without the (listener, client task group) pair tables;
  * the window's Count-Min rows k_cms_partial + k_cms_reduce (tests/cpp/kemu/test_cms.cc) with ragged, short and empty chunks;
  * the wire front-end's record walk k_wire_* + the scan kernels (tests/cpp/kemu/test_wire.cc) on message streams with corrupted length
    fields, record counts and paddings: "malformed" exactly when the oracle's restatement of the reference's validators rejects a
    message, the serial walk's offsets otherwise (tools/kemu_tsan.sh address runs it under AddressSanitizer: no corrupted length makes
    a kernel index outside the stream or its work arrays);
  * the LISTENER_STATE_NOTIFY roll-up k_lstate_ingest and the ACTIVE_CONN_STATS roll-up k_actconn_ingest (tests/cpp/kemu/test_lstate.cc)
    on records whose bytes are random except the listener id (any state, any flags, counters whose int sums wrap);
  * the per-host top-10 selection k_topn_hosts and the candidate filter k_topn_filter (tests/cpp/kemu/test_topn.cc): hosts of 0 ... 2 700
    listeners, ties, stale and foreign records, all four kinds;
  * the filtered multi-host listener-state query (tests/cpp/kemu/test_svcquery.cc): k_svc_filter, the radix selection, k_svc_gather and
    k_svc_aggr on random records / filters / sorts / maxrecs against the oracle's serial walk (oracle/gy_oracle_query.c);
  * the listener's state decision k_listener_decide (tests/cpp/kemu/test_ldecide.cc): TCP_LISTENER::get_curr_state's decision tree on random
    scan records and task / host inputs, the history bytes carried over six rounds, against oracle/gy_oracle_lstate.c;
  * the roll-up digests k_rollup_accum / k_rollup_cluster (tests/cpp/kemu/test_rollup.cc): groups of services and groups of slabs, the union by value bin, 64-bit
    weights beyond 2^32, members without clusters / without buffered values / empty.
This does not replace the -m gpu parity tests (no memory model, no execution masks, no timing): it catches logic errors in kernel
changes before GPU minutes are spent on them.  The programs are built and run side by side once per session (they mostly wait in
barriers); each test below looks at one of them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEMU = os.path.join(ROOT, "tests", "cpp", "kemu")

# name -> (source, defines, arguments, marker of success)
BINS = []  # (a tree whose value-bin kernel is templated on the thread count sets e.g. ["KEMU_BINS_NT=512", "KEMU_BINS_TEMPLATE_NT"])
PROGRAMS = {
    "bins-12345": ("test_bins.cc", BINS, ["12345"], "kemu bins ok"),
    "bins-7": ("test_bins.cc", BINS, ["7"], "kemu bins ok"),
    "bins-99": ("test_bins.cc", BINS, ["99"], "kemu bins ok"),
    # the instances for larger buffers (gys_config.td_pend_cap): merges of up to 2048 / 4096 values, hand-over of merges with more large values than the list holds
    "bins-2048-values": ("test_bins.cc", ["KEMU_BINS_VPT=8"] + BINS, ["12345"], "kemu bins ok"),
    "bins-4096-values": ("test_bins.cc", ["KEMU_BINS_VPT=16"] + BINS, ["12345"], "kemu bins ok"),
    # round 6: the STREAMED instance for merges of 4097 .. 16 384 values (size class 2: pass 1 takes the values eight per thread at a time, ties among large values by list position)
    "bins-16384-values-streamed": ("test_bins.cc", ["KEMU_BINS_VPT=64"] + BINS, ["12345"], "kemu bins ok"),
    "bins-16384-values-streamed-7": ("test_bins.cc", ["KEMU_BINS_VPT=64"] + BINS, ["7"], "kemu bins ok"),
    "resp-tiles-16384": ("test_resp.cc", ["KEMU_TPT=16"] + BINS, ["4242"], "kemu resp ok"),
    "resp-tiles-6144": ("test_resp.cc", ["KEMU_TPT=12"] + BINS, ["4242"], "kemu resp ok"),
    "resp-512x32-prefetch": ("test_resp.cc", ["KEMU_TPT=32"] + BINS, ["4242"], "kemu resp ok"),
    "resp-split-form": ("test_resp.cc", ["KEMU_TPT=16", "KEMU_SPLIT", "KEMU_NB=3"] + BINS, ["4242"], "kemu resp ok"),
    # the whole pipeline with td_pend_cap 1536 / 3072 (fast merge classes of 2048 / 4096 values) against the oracle engine with the same buffer size
    "resp-pend-cap-1536": ("test_resp.cc", ["KEMU_TPT=16", "KEMU_PEND_CAP=1536", "KEMU_NB=12"] + BINS, ["4246"], "kemu resp ok"),
    "resp-pend-cap-3072": ("test_resp.cc", ["KEMU_TPT=16", "KEMU_PEND_CAP=3072", "KEMU_NB=18"] + BINS, ["4247"], "kemu resp ok"),
    # keys with candidates (bound-address listeners, two listeners on one port): the event's server address picks the listener
    "resp-bound-address": ("test_resp.cc", ["KEMU_TPT=16", "KEMU_MODE=1"] + BINS, ["4243"], "kemu resp ok"),
    # the same world, batches alternating between IPv4 events and 48-byte IPv6 events (resp_bitmap_v6_ rows, embedded IPv4 addresses)
    "resp-ipv6-mixed": ("test_resp.cc", ["KEMU_TPT=16", "KEMU_MODE=2"] + BINS, ["4244"], "kemu resp ok"),
    "resp-ipv6-split-form": ("test_resp.cc", ["KEMU_TPT=16", "KEMU_MODE=2", "KEMU_SPLIT", "KEMU_NB=4"] + BINS, ["4245"], "kemu resp ok"),
    "spill-and-huge": ("test_spill.cc", BINS, ["777"], "kemu spill ok"),
    "spill-predicted-runs": ("test_spill.cc", ["KEMU_PRESPILL"] + BINS, ["778"], "kemu spill ok"),
    "conn-31": ("test_conn.cc", [], ["31"], "kemu conn ok"),
    "conn-77": ("test_conn.cc", [], ["77"], "kemu conn ok"),
    # 5000 services: a workgroup's records name more of them than its LDS table holds (direct device adds); 3 CUs: other spans
    "conn-full-table": ("test_conn.cc", [], ["5", "5000", "1"], "kemu conn ok"),
    "conn-3-cus": ("test_conn.cc", [], ["9", "300", "3"], "kemu conn ok"),
    "cms-rows": ("test_cms.cc", [], ["5"], "kemu cms ok"),
    "wire-corrupted-11": ("test_wire.cc", [], ["11"], "kemu wire ok"),
    "wire-corrupted-23": ("test_wire.cc", [], ["23"], "kemu wire ok"),
    "lstate-actconn-random-3": ("test_lstate.cc", [], ["3"], "kemu lstate ok"),
    "lstate-actconn-random-4": ("test_lstate.cc", [], ["4"], "kemu lstate ok"),
    "topn-17": ("test_topn.cc", [], ["17"], "kemu topn ok"),
    "rollup-9": ("test_rollup.cc", [], ["9"], "kemu rollup ok"),
    "rollup-10": ("test_rollup.cc", [], ["10", "3"], "kemu rollup ok"),  # (a buffer stride that is not a multiple of four words: the 4-byte loads)
    # the listener's state decision (get_curr_state) on random scan records / inputs over six rounds
    "ldecide-5": ("test_ldecide.cc", [], ["5"], "kemu ldecide ok"),
    "svcquery-5": ("test_svcquery.cc", [], ["5"], "kemu svcquery ok"),
    "svcquery-6": ("test_svcquery.cc", [], ["6"], "kemu svcquery ok"),
}


@pytest.fixture(scope="module")
def kemu_results(tmp_path_factory, oracle):
    oracle.lib()  # builds oracle/liboracle.so if needed
    out = tmp_path_factory.mktemp("kemu")
    odir = os.path.join(ROOT, "oracle")
    exes, builds = {}, {}
    for name, (src, defs, _, _) in PROGRAMS.items():
        key = (src, tuple(defs))
        if key in exes:
            continue
        exe = str(out / ("kemu_%d" % len(exes)))
        exes[key] = exe
        builds[key] = subprocess.Popen(["g++", "-std=c++20", "-O1", "-w", "-I" + KEMU] + ["-D" + d for d in defs] +
                                       [os.path.join(KEMU, src), "-o", exe, "-L" + odir, "-l:liboracle.so", "-Wl,-rpath," + odir, "-pthread"],
                                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    build_log = {key: (p.communicate()[0], p.returncode) for key, p in builds.items()}
    runs = {}
    for name, (src, defs, args, _) in PROGRAMS.items():
        key = (src, tuple(defs))
        if build_log[key][1] != 0:
            continue
        runs[name] = subprocess.Popen(["timeout", "-s", "KILL", "1200", exes[key]] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    results = {}
    for name, (src, defs, _, _) in PROGRAMS.items():
        key = (src, tuple(defs))
        if name in runs:
            so, se = runs[name].communicate()
            results[name] = (runs[name].returncode, so, se)
        else:
            results[name] = (-1, "", "build failed:\n" + build_log[key][0][-3000:])
    return results


@pytest.mark.parametrize("name", list(PROGRAMS))
def test_kernel_logic_equals_oracle(kemu_results, name):
    rc, so, se = kemu_results[name]
    if rc == 77:
        pytest.skip(so.strip())
    assert rc == 0 and PROGRAMS[name][3] in so, (rc, so[-2000:], se[-2000:])

#!/usr/bin/env python3
"""bench.py -- events/sec ingested into merged HLL + Count-Min + t-digest (+ exact histogram) sketches on MI355X.

Workload (BASELINE.json configs[2]/[3]): 10 000 hosts x 1 000 services = 10^7 service keys, raw 24-byte
tcp_ipv4_resp_event_t response events uniform over the keys, per-service latency lognormal(mu_s, 1.5) with mu_s ~ N(3,1).
Hosts are sharded over ranks by GY_MACHINE_ID::get_hash() % N; every rank ingests a fixed number of events per step
drawn over ITS hosts (weak scaling in events) and one step = one 5-second window: ingest one device-resident batch, then the
window close (RCCL all-reduce of the HLL / CMS / histogram / cluster registers over xGMI + local roll).

Events per window: 2^29 per GPU by default: the north-star rate of 1 G events/s on 8 GPUs is 125 M events/s/GPU, i.e. 2^29.2 events
in each GPU's 5-s window; 2 steps ingest the 2^30 events SURVEY 8d quotes per C3 run (`--events 268435456` = a window at 54 M/s/GPU).  A key re-clusters
its t-digest once per ~256 values (every ~9.5 windows at ~27 events per key and window); an untimed set-up pass spreads the keys'
buffer fill levels evenly and then runs one full buffer cycle of ordinary windows, so that EVERY timed window, whatever
--steps/--warmup, carries its long-run share of (steady-state, non-empty-digest) merges.

One process per GPU (torchrun env RANK/LOCAL_RANK/WORLD_SIZE); without a launcher `--gpus N` starts its own N ranks (self_launch).  After
an N > 1 run every rank's reduced registers are checksummed and compared, and a small side configuration run through the same exchange is
compared with a single-rank engine (`exchange_check` in the JSON line; a mismatch exits with status 5).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

T_START = time.perf_counter()
EVENT_BYTES = 24          # algorithmic bytes per event (raw tcp_ipv4_resp_event_t)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _mem_available_bytes():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) * 1024
    except OSError:
        pass
    return None


LINE_LIMIT = 4096  # the driver keeps an 8-KB tail of stdout: the LAST line must fit whole, with room to spare


def _r(x, nd=6):
    """a float at nd significant digits (the detail file keeps full precision); everything else as it is; NaN / inf -> None (strict JSON)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (nd, x))
    if isinstance(x, (np.floating,)):
        return _r(float(x), nd)
    if isinstance(x, (np.integer,)):
        return int(x)
    return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out, detail_path=None):
    """The ONE stdout line the driver parses: the contract's keys + roofline + cpu_baseline + quantile_error + one short entry per sub-run,
    <= LINE_LIMIT bytes of strict JSON.  Everything else of `out` (per-kernel tables, counter-traffic blocks, host-fed legs, scans, notes)
    goes to the detail file (--detail-out)."""
    c = {k: _r(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                     "dtype", "data", "parity_ok")}
    cfg = out.get("config", {})
    c["config"] = _pick(cfg, ("workload", "events_per_rank_per_step", "service_keys_total", "multi_level_windows", "td_pend_cap", "td_pend_cap_is_library_default",
                              "exchange", "exchange_requested", "exchange_fallback", "records_per_step"))
    rf = out.get("roofline", {})
    r = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_step", "kernel", "kernel_avg_ms", "kernel_frac", "kernel_frac_24B"))
    tr = rf.get("traffic")
    r["traffic"] = ({"bytes": tr.get("bytes"), "source": str(tr.get("source", ""))[:60], "same_kernels_as_this_run": tr.get("same_kernels_as_this_run")}
                    if isinstance(tr, dict) else None)
    c["roofline"] = r
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        if "error" in cb and "value" not in cb:
            c["cpu_baseline"] = {"error": str(cb["error"])[:120]}
        else:
            c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "keys", "events", "events_per_key", "allcores_value", "allcores", "reference_hist_value",
                                           "reference_hist_allcores_value", "gpu_events_per_key_per_window", "leg"))
            c["cpu_baseline"]["sample"] = str(cb.get("sample_short") or cb.get("sample") or "")[:330]
            fk = cb.get("matched_ratio")
            if isinstance(fk, dict):
                c["cpu_baseline"]["matched_ratio"] = _pick(fk, ("value", "keys", "events_per_key", "allcores_value", "reference_hist_value", "reference_hist_allcores_value"))
    qe = out.get("quantile_error")
    if isinstance(qe, dict):
        c["quantile_error"] = _pick(qe, ("keys_checked", "tolerance", "p50_rank_err_max", "p99_rank_err_max"))
    xc = out.get("exchange_check")
    if isinstance(xc, dict):
        c["exchange_check"] = _pick(xc, ("ok", "ranks_seen", "ranks_consistent", "exchange", "global_digest_consistent", "global_digest_ms"))
    qs = out.get("quantile_scan")
    if isinstance(qs, dict):
        c["quantile_scan"] = _pick(qs, ("services", "kernel_ms", "global_rollup_ms"))
    subs = out.get("configs")
    if isinstance(subs, dict):
        c["configs"] = {}
        for name, e in subs.items():
            if not isinstance(e, dict) or "value" not in e:
                c["configs"][name] = {"error": str((e or {}).get("error", "no line"))[:80]}
                continue
            erf = e.get("roofline", {})
            c["configs"][name] = {"value": _r(e.get("value")), "unit": e.get("unit"), "ms_per_step": _r(e.get("ms_per_step")), "frac": _r(erf.get("frac")),
                                  "kernel": erf.get("kernel"), "kernel_frac": _r(erf.get("kernel_frac")), "parity_ok": e.get("parity_ok")}
    for k in ("device_code", "build_commit", "wall_s"):
        if out.get(k) is not None:
            c[k] = _r(out[k])
    if detail_path:
        c["detail"] = detail_path
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(line) > LINE_LIMIT:  # never happens with the fields above; if a later edit grows them, drop the optional parts rather than the contract's
        for k in ("quantile_scan", "exchange_check", "configs", "build_commit"):
            c.pop(k, None)
            line = json.dumps(c, allow_nan=False, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    return line


def _jsonable(x):
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, (np.floating, float)):
        x = float(x)
        return None if (x != x or x in (float("inf"), float("-inf"))) else x
    if isinstance(x, np.integer):
        return int(x)
    if isinstance(x, np.bool_):
        return bool(x)
    return x


def emit(out, args):
    """detail file first (full `out`, strict JSON), then the compact line as the LAST thing on stdout.  A sub-run (--sub) hands its full line to
    the parent run instead (one line, parsed there, never seen by the driver)."""
    out = _jsonable(out)
    if getattr(args, "sub", ""):
        print(json.dumps(out, allow_nan=False), flush=True)
        return
    path = getattr(args, "detail_out", "") or ""
    wrote = None
    if path and path != "none":
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
            with open(path, "w") as f:
                json.dump(out, f, allow_nan=False)
                f.write("\n")
            wrote = os.path.relpath(os.path.abspath(path), os.getcwd())
        except OSError as ex:
            print(f"bench.py: detail file {path} not written: {ex}", file=sys.stderr)
    if getattr(args, "detail_stdout", False):
        print(json.dumps(out, allow_nan=False), flush=True)
    sys.stderr.flush()
    print(compact_line(out, wrote), flush=True)


def cpu_baseline(eng, total_hosts_sample, svcs, nevents, seed, td_cap=0, reps=1, histonly=True, mt_reps=5):
    """The oracle's sequential restatement of the same hot loop ("port") and the reference's own GY_HISTOGRAM loop (oracle/_ref), timed on the
    host's cores on a bounded sample of the same stream shape (same generator, same bytes): `reps` runs each (the median is reported; every
    repetition ingests the same batch again, as a next window would), one core and all cores.  Returns a dict, or None when the host has
    not the memory for the port's per-key state (6.3 KB + 4 B x (td_cap - 896) per key)."""
    from gyeeta_amd import wire
    from oracle import oracle as o
    nsvc = total_hosts_sample * svcs
    cap = td_cap or o.TD_PEND_CAP
    need = nsvc * (6400 + 4 * max(0, cap - o.TD_PEND_CAP) + 600) + nevents * 40
    avail = _mem_available_bytes()
    if avail is not None and avail < 1.6 * need:
        print(f"bench.py: CPU legs at {nsvc} keys skipped: they need ~{need >> 30} GiB of host memory, {avail >> 30} GiB available", file=sys.stderr)
        return None
    med = lambda xs: sorted(xs)[len(xs) // 2]
    marks = [("start", time.perf_counter())]
    mark = lambda name: marks.append((name, time.perf_counter()))
    orc = o.OracleEngine(nsvc, td_cap=td_cap)
    orc2 = o.OracleEngine(nsvc, enable_td=False) if histonly else None
    s = np.arange(svcs)
    for h in range(total_hosts_sample):
        g, ns, pt = wire.glob_id(np.full(svcs, h), s), wire.listener_netns(h, s), wire.listener_port(s)
        orc.register_bulk(h, g, ns, pt)
        if orc2 is not None:
            orc2.register_bulk(h, g, ns, pt)
    ev = torch.empty(nevents * 24, dtype=torch.uint8, device="cuda")
    segs = eng.gen_resp_events(ev.data_ptr(), nevents, seed, 0, total_hosts_sample, svcs)
    eng.sync()
    host = ev.cpu().numpy().tobytes()
    del ev
    sh = [sg.host_slot for sg in segs]
    sf = [sg.first_event for sg in segs]
    mark("setup")
    ncores = os.cpu_count() or 1
    out = {"keys": nsvc, "events": nevents, "events_per_key": nevents / max(nsvc, 1), "runs": reps, "td_pend_cap": cap, "cores_all": ncores}
    if reps > 1:  # (a first pass pays for the page faults of the per-key state: not timed)
        orc.resp_batch(host, sh, sf)
        if orc2 is not None:
            orc2.resp_batch(host, sh, sf, histonly=True)
    mark("first_pass")
    full, honly = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        orc.resp_batch(host, sh, sf)
        t1 = time.perf_counter()
        full.append(nevents / (t1 - t0))
        if orc2 is not None:
            orc2.resp_batch(host, sh, sf, histonly=True)
            honly.append(nevents / (time.perf_counter() - t1))
    out["port"] = med(full)
    if honly:
        out["histonly"] = med(honly)
    del orc2
    mark("port")
    try:  # the full port again on every host core (hosts cut into per-thread ranges; identical resulting state, tests/test_oracle_sketches.py)
        if ncores > 1:
            rates = []
            for _ in range(mt_reps):
                t7 = time.perf_counter()
                orc.resp_batch(host, sh, sf, nthreads=ncores)
                rates.append(nevents / (time.perf_counter() - t7))
            out["port_allcores"] = med(rates)
            out["port_allcores_form"] = ("hosts cut into per-thread ranges; per-thread private HLL registers, all-service histogram and counters merged once "
                                         "per batch; Count-Min rows built from per-service counts; digests re-clustered in parallel over service ranges")
    except Exception as ex:  # never let the optional leg take the JSON line down
        print(f"bench.py: all-cores port baseline skipped: {ex}", file=sys.stderr)
    del orc
    mark("port_allcores")
    # the reference's OWN classes on the same bytes (oracle/_ref: GY_HISTOGRAM<int64_t, RESP_TIME_HASH>::add_data behind an
    # unordered_map with GY_JHASHER standing in for the RCU listener table): kind "reference"
    R = o.ref()
    if R is not None and hasattr(R, "ref_keyed_new"):
        k = R.ref_keyed_new()
        for h in range(total_hosts_sample):
            ns = np.ascontiguousarray(wire.listener_netns(h, s), dtype=np.uint32)
            pt = np.ascontiguousarray(wire.listener_port(s), dtype=np.uint16)
            if hasattr(R, "ref_keyed_register_bulk"):
                R.ref_keyed_register_bulk(k, h, o.ptr(ns, o.u32p), pt.ctypes.data_as(ctypes_u16p()), svcs)
            else:
                for i in range(svcs):
                    R.ref_keyed_register(k, h, int(ns[i]), int(pt[i]))
        buf = np.frombuffer(host, dtype=np.uint8)
        sh_a = np.ascontiguousarray(sh, dtype=np.uint32)
        sf_a = np.ascontiguousarray(sf, dtype=np.uint64)
        rates, added = [], 0
        for _ in range(reps + (1 if reps > 1 else 0)):
            t3 = time.perf_counter()
            added = R.ref_keyed_resp_batch(k, buf.ctypes.data, nevents, o.ptr(sh_a, o.u32p), o.ptr(sf_a, o.u64p), len(sh_a))
            rates.append(nevents / (time.perf_counter() - t3))
        if added:
            out["reference_hist"] = med(rates[1:] if reps > 1 else rates)
        if added and ncores > 1 and hasattr(R, "ref_keyed_resp_batch_mt"):  # the same loop on every host core, hosts cut into ranges
            rates = []
            for _ in range(mt_reps):
                t5 = time.perf_counter()
                R.ref_keyed_resp_batch_mt(k, buf.ctypes.data, nevents, o.ptr(sh_a, o.u32p), o.ptr(sf_a, o.u64p), len(sh_a), ncores)
                rates.append(nevents / (time.perf_counter() - t5))
            out["reference_hist_allcores"] = med(rates)
        R.ref_keyed_free(k)
    mark("reference")
    out["phase_s"] = {marks[i][0]: round(marks[i][1] - marks[i - 1][1], 2) for i in range(1, len(marks))}
    return out


def ctypes_u16p():
    import ctypes
    return ctypes.POINTER(ctypes.c_uint16)


def quantile_error(eng, torch, ingested, nlocal, svcs, host_ids, host_slots, wire):
    """Second half of BASELINE.json's metric: p50 / p99 of the engine's t-digests against the EXACT quantiles of everything the run fed
    them.  Untimed, after the run: every distinct batch is regenerated (the generator is deterministic); the segments of two hosts
    (first and last slot: 2 x svcs service keys) are decoded and sorted by (service, response time) with torch on the device (test
    plumbing: a 2^26-event segment sorts in milliseconds there, in ~10 s with numpy) and pulled to the CPU; a key's value multiset is
    the union of its sorted per-batch slices, each with the number of times that batch was ingested; rank error = distance of the
    engine's quantile from the [left, right] rank interval of that value in the sorted data (the north-star tolerance is 0.01)."""
    tmp = torch.empty(max(n for n, _, _, _ in ingested) * EVENT_BYTES, dtype=torch.uint8, device="cuda")
    per_key = [[[] for _ in range(svcs)] for _ in host_slots]
    for n, seed, code, times in ingested:
        if times == 0:
            continue
        sg = eng.gen_resp_events(tmp.data_ptr(), n, seed, 0, nlocal, svcs, code)
        eng.sync()
        for hi, slot in enumerate(host_slots):
            lo = sg[slot].first_event
            hi_e = sg[slot + 1].first_event if slot + 1 < nlocal else n
            if hi_e <= lo:
                continue
            w = tmp[lo * EVENT_BYTES:hi_e * EVENT_BYTES].view(torch.int32).view(-1, 6).to(torch.int64)
            lat = (w[:, 4] - w[:, 5]) & 0xFFFFFFFF                       # lsndtime - lrcvtime, 32-bit wrap (common/gy_socket_stat.cc:1519)
            pw = w[:, 3] & 0xFFFF                                         # sport, network byte order
            svc = (((pw & 0xFF) << 8) | (pw >> 8)) - 1024                 # wire.listener_port(s) = 1024 + s for s < 60000
            ok = (lat <= 1000000) & (svc >= 0) & (svc < svcs)
            key = torch.sort(svc[ok] * (1 << 21) + lat[ok]).values.cpu().numpy()
            del w, lat, pw, svc, ok
            cuts = np.searchsorted(key, np.arange(svcs + 1, dtype=np.int64) << 21)
            lat_s = (key & ((1 << 21) - 1)).astype(np.int32)
            for s_idx in range(svcs):
                if cuts[s_idx + 1] > cuts[s_idx]:
                    per_key[hi][s_idx].append((lat_s[cuts[s_idx]:cuts[s_idx + 1]], times))  # sorted by value
    res = {"p50": [], "p99": []}
    for hi, h in enumerate(host_ids):
        gids = wire.glob_id(np.full(svcs, h), np.arange(svcs))
        for s_idx in range(svcs):
            parts = per_key[hi][s_idx]
            if not parts:
                continue
            total = sum(len(v) * t for v, t in parts)
            got = eng.quantiles(int(gids[s_idx]), [0.5, 0.99])
            for q, g, name in ((0.5, got[0], "p50"), (0.99, got[1], "p99")):
                lo = sum(int(np.searchsorted(v, g, side="left")) * t for v, t in parts) / total
                hi_r = sum(int(np.searchsorted(v, g, side="right")) * t for v, t in parts) / total
                res[name].append(0.0 if lo <= q <= hi_r else min(abs(lo - q), abs(hi_r - q)))
    return {"keys_checked": len(res["p50"]), "tolerance": 0.01,
            "p50_rank_err_max": float(np.max(res["p50"])), "p50_rank_err_mean": float(np.mean(res["p50"])),
            "p99_rank_err_max": float(np.max(res["p99"])), "p99_rank_err_mean": float(np.mean(res["p99"])),
            "reference": "exact sort of every value the run ingested for the key (the reference's own answer is a RESP_TIME_HASH bucket ceiling)"}


def host_fed_rate(eng, torch, nev, nlocal, svcs):
    """When the boundary hands over HOST buffers the events cross PCIe first.  Measured after the run (never part of `value`): a batch
    of nev events in pinned host memory, 3 x (asynchronous H2D copy on the engine's stream + ingest + window close), wall clock."""
    dev = torch.empty(nev * EVENT_BYTES, dtype=torch.uint8, device="cuda")
    sg = eng.gen_resp_events(dev.data_ptr(), nev, 0xF00D, 0, nlocal, svcs)
    eng.sync()
    pinned = torch.empty(nev * EVENT_BYTES, dtype=torch.uint8, pin_memory=True)
    pinned.copy_(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(eng.stream):
        dev.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    t_copy = time.perf_counter() - t0
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(eng.stream):  # copy and ingest in one stream's order
            dev.copy_(pinned, non_blocking=True)
        eng.handle_resp_events_dev(sg, dev.data_ptr(), nev)
        eng.window_close(tusec=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": reps * nev / dt, "unit": "events/s", "events_per_batch": nev, "h2d_GBps": nev * EVENT_BYTES / t_copy / 1e9,
            "note": "pinned host buffer -> HBM copy on the engine stream, then ingest + window close, serial; PCIe Gen5 x16 bound"}


def host_fed_l2_threads(nthreads=16, secs=2.0):
    """The boundary as MCONN_HANDLER's L2 threads would drive it: tools/cpp/bench_hostfed.cc (plain g++ against the C ABI) -- 16
    threads x {2048-record TCP_CONN_NOTIFY, 512-record LISTENER_STATE_NOTIFY, 65536-event response batches} from pageable host
    buffers through GYS_MCONN_HANDLER (pinned staging ring inside the library, no GPU wait per call).  Its own process and context."""
    import subprocess
    import tempfile
    lib = os.path.join(ROOT, "gyeeta_amd", "lib")
    exe = os.path.join(tempfile.gettempdir(), "gys_bench_hostfed")
    try:
        subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tools", "cpp", "bench_hostfed.cc"), "-o", exe, "-L" + lib, "-lgysketch",
                               "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-pthread"], stderr=subprocess.DEVNULL)
        r = subprocess.run([exe, str(nthreads), str(secs)], capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            return {"error": (r.stderr or "").strip()[-300:], "rc": r.returncode}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as ex:  # optional leg: never take the JSON line down
        return {"error": str(ex)[:300]}


def _kernel_sources():
    try:
        from gyeeta_amd.build import sources_sha
        return sources_sha()
    except Exception:
        return None


def _device_code():
    try:
        from gyeeta_amd.build import device_code_sha
        return device_code_sha()
    except Exception:
        return None


def _same_kernels(t):
    """are the counters in profiles/pmc_traffic.json of the kernels this run executes?  The device code itself when both sides carry its
    hash (.rodata + .text of the gfx950 code object: a host-only change of gys_engine.hip does not move it), else the source files."""
    if t.get("device_code") and _device_code():
        return t["device_code"] == _device_code()
    if t.get("source_kernels"):
        return t["source_kernels"] == _kernel_sources()
    return None


def pmc_traffic(kernel, events, nsvc):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, produced by
    tools/pmc_collect.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950).  Only reported when the PMC run used this very workload; otherwise null."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return None
    if t.get("events_per_launch") != events or t.get("service_keys") != nsvc:
        return None
    k = t.get("kernels", {}).get(kernel)
    if not k:
        return None
    return {"bytes": k["fetch_bytes"] + k["write_bytes"], "fetch_bytes": k["fetch_bytes"], "write_bytes": k["write_bytes"],
            "source": t.get("source", "profiles/pmc_traffic.json"), "measured_in_this_run": False,
            "source_commit": t.get("source_commit"), "source_kernels": t.get("source_kernels"), "device_code": t.get("device_code"),
            "same_kernels_as_this_run": _same_kernels(t),
            "note": "counter passes are separate rocprofv3 runs (profiles/pmc_traffic.json); source_commit = the tree they were taken on; "
            "device_code = sha256 over .rodata + .text of the library's gfx950 code object (equal to this line's device_code: the counters "
            "are of these very kernels, whatever documentation or host-only commits lie between); source_kernels / kernel_sources = the "
            "same over the library's source files"}


def scope_of_kernel(k):
    """the profile scope (gys_engine.hip ProfScope) a compact kernel name of profiles/pmc_traffic.json belongs to -- tools/pmc_workload.py's rule"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_workload", os.path.join(ROOT, "tools", "pmc_workload.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.scope_of(k)


def workload_traffic(name, scope, units_per_step):
    """HBM bytes per STEP of the kernels of one profile scope in the sub-run `name`, from the committed counter passes of that sub-run
    (profiles/pmc_traffic.json `workloads`, written by tools/pmc_workload.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of
    the same command).  None when the passes were taken on another workload size."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["workloads"][name]
    except Exception:
        return None
    if t.get("units_per_step") not in (None, units_per_step):
        return None
    ks = {k: v for k, v in t.get("kernels", {}).items() if v.get("scope") == scope}
    if not ks:
        return None
    fb = sum(v["fetch_bytes"] for v in ks.values())
    wb = sum(v["write_bytes"] for v in ks.values())
    return {"bytes": fb + wb, "fetch_bytes": fb, "write_bytes": wb, "kernels": sorted(ks), "per": "step", "source": "profiles/pmc_traffic.json workloads." + name,
            "measured_in_this_run": False, "source_commit": t.get("source_commit"), "device_code": t.get("device_code"),
            "same_kernels_as_this_run": _same_kernels(t)}


SUB_CONFIGS = [  # (name, BASELINE.json config it stands for, bench.py arguments): the other configurations, run after the default line
    ("c2_conn", "configs[1]: 1k hosts x 100 services, HLL distinct-flow + CMS on TCP_CONN_NOTIFY records",
     ["--workload", "conn", "--steps", "40", "--warmup", "4"]),
    ("c1", "configs[0] shape: 1 host x 100 services, 2^26 response events per window",
     ["--hosts", "1", "--svcs", "100", "--events", str(1 << 26), "--steps", "20", "--warmup", "5", "--nbuf", "2"]),
    ("c5_zipf", "configs[4] shape: 10^5 services, Zipf 1.1, one hipGraph-captured window close per 2^29-event batch",
     ["--zipf-milli", "1100", "--hosts", "50", "--svcs", "2000", "--steps", "20", "--warmup", "5", "--nbuf", "2"]),
    ("c3_levels", "configs[2] with the multi-level windows on (5 s / 300 s / 5 days / all per service: what the reference's 5-s flush does per listener, "
                  "common/gy_socket_stat.cc:4163-4172 + TIME_HISTOGRAM levels common/gy_statistics.h:1082-1551)",
     ["--levels", "1", "--steps", "24", "--warmup", "6", "--nbuf", "2"]),
]


def run_sub_configs(names):
    """The other BASELINE configurations as short sub-runs of this very script (`--sub <name>`: no CPU baseline, no host-fed leg, no
    quantile scan), each in a process of its own after this run's engine has released the device; their lines are trimmed to the fields
    a reader needs (rate, step time, roofline with the dominant kernel's HIP-event time and counter traffic, parity)."""
    import subprocess
    out = {}
    for name, what, argv in SUB_CONFIGS:
        if names and name not in names:
            continue
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--sub", name] + argv, capture_output=True, text=True, timeout=420)
            line = None
            for ln in reversed((r.stdout or "").strip().splitlines()):
                if ln.startswith("{"):
                    line = json.loads(ln)
                    break
            if line is None:
                out[name] = {"error": ((r.stderr or "") + (r.stdout or "")).strip()[-400:], "rc": r.returncode}
                continue
            rf = line.get("roofline", {})
            ent = {"stands_for": what, "command": "bench.py " + " ".join(argv), "value": line.get("value"), "unit": line.get("unit"),
                   "ms_per_step": line.get("ms_per_step"), "steps": line.get("steps"), "workload": line.get("config", {}).get("workload"),
                   "parity_ok": line.get("parity_ok"), "rc": r.returncode, "wall_s": round(time.perf_counter() - t0, 1),
                   "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_step", "kernel", "kernel_avg_ms",
                                                        "kernel_frac", "kernel_frac_measured_bytes", "traffic", "note") if k in rf}}
            ent["roofline"]["kernels"] = {k: {kk: vv for kk, vv in v.items() if kk in ("ms", "launches_per_step", "frac", "achieved_GBps", "traffic")}
                                          for k, v in rf.get("kernels", {}).items() if v.get("ms", 0) >= 0.01}
            if "quantile_error" in line:
                ent["quantile_error"] = {k: line["quantile_error"][k] for k in ("keys_checked", "p50_rank_err_max", "p99_rank_err_max")}
            if "checks" in line:
                ent["checks"] = line["checks"]
            if "cpu_baseline" in line:
                ent["cpu_baseline"] = line["cpu_baseline"]
            out[name] = ent
        except Exception as ex:  # a sub-run never takes the default line down
            out[name] = {"error": str(ex)[:300]}
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: re-run this very command line under torch.distributed.run, one rank per
    GPU of this node (what the driver does itself for N > 1).  Returns the launcher's exit status."""
    import socket
    import subprocess
    if not args.selftest_launch:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < (1 if args.share_device else args.gpus):
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} devices, {have} visible on this node", file=sys.stderr)
            return 2
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def digest64(*arrays):
    """one signed 64-bit checksum of a group of byte strings / numpy arrays (order matters)"""
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    for a in arrays:
        h.update(a if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a).tobytes())
    return int.from_bytes(h.digest(), "little", signed=True)


REG_FAMILIES = ["hll", "cms32", "cms64", "global_hist", "cluster_rows"]


def observe_registers(eng, clusters):
    """checksums of what the window exchange leaves on a rank: the reduced HLL registers, both Count-Min tables, the all-service histogram
    and every cluster's STATE_ONE row of the last finished window.  After a correct exchange every rank holds the same five numbers."""
    gh = eng.export_global_hist()
    ghb = np.array([(gh.stats[i].count, gh.stats[i].sum) for i in range(15)] + [(gh.total_count, gh.max_val_seen)], dtype=np.int64)
    cl = np.array([eng.clusterstate(c).as_tuple() for c in clusters], dtype=np.int64)
    return [digest64(eng.export_hll()), digest64(eng.export_cms(0)), digest64(eng.export_cms(1)), digest64(ghb), digest64(cl)]


def verify_ranks(mine, world, device):
    """all-gather of every rank's checksums (torch.distributed, whatever backend the group has); returns (matrix [world][k], consistent)"""
    t = torch.tensor(mine, dtype=torch.int64, device=device)
    got = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(got, t)
    m = [g.cpu().tolist() for g in got]
    return m, all(row == m[0] for row in m)


def exchange_side_check(args, rank, world, local_rank, L, main_eng, exchange, wire, SketchEngine, mid_buf, cdev):
    """Small configuration (96 hosts x 6 services, 2 000 response events per host, fixed seeds per host) run twice: sharded over the
    ranks through the SAME exchange path as the timed run, and -- on rank 0 -- by one single-rank engine fed all hosts.  The reduced
    registers of every rank must equal the single-rank run's."""
    nh, sp, nev = 96, 6, 2000
    clusters = ["cluster%d" % c for c in range(8)]
    mids = [wire.machine_id(h) for h in range(nh)]

    def events(h):
        rng = np.random.default_rng(7000 + h)
        ev = np.zeros(nev, dtype=wire.RESP_EVENT)
        s_ = rng.integers(0, sp, nev)
        ev["saddr"] = 0x0A000001 + h
        ev["daddr"] = rng.integers(1, 1 << 32, nev, dtype=np.uint64).astype(np.uint32)
        ev["netns"] = wire.listener_netns(h, s_)
        ev["sport_be"] = wire.listener_port(s_)
        ev["dport_be"] = rng.integers(16000, 65536, nev)
        lat = np.minimum(np.floor(rng.lognormal(3.0, 1.5, nev)), 1e6).astype(np.uint32)
        lrcv = rng.integers(0, 1 << 32, nev, dtype=np.uint64).astype(np.uint32)
        with np.errstate(over="ignore"):
            ev["lsndtime"] = lrcv + lat
        ev["lrcvtime"] = lrcv
        return ev

    def run(eng, hosts, close):
        for c in clusters:
            eng.register_cluster(c)
        s_ = np.arange(sp)
        for h in hosts:
            eng.register_host(mids[h], clusters[h % 8])
            eng.register_listeners_np(mids[h], wire.glob_id(np.full(sp, h), s_), wire.listener_netns(h, s_), wire.listener_port(s_))
            eng.handle_host_state(mids[h], ntasks=10 + h, nlisten=sp)
        for h in hosts:
            eng.handle_resp_events(mids[h], events(h))
        close(eng)
        return observe_registers(eng, clusters)

    mine = [h for h in range(nh) if L.gys_shard_of(mid_buf(mids[h]), world) == rank]
    e2 = SketchEngine(max_hosts=nh, max_services=nh * sp, max_clusters=16, enable_tdigest=True, max_batch_events=1 << 16, rank=rank, nranks=world,
                      device=local_rank)
    if exchange == "rccl_in_library":
        e2.comm = main_eng.comm  # the communicator of the timed run (a communicator belongs to the rank, not to a context)
        obs = run(e2, mine, lambda e: e.window_close_rccl(tusec=5_000_000))
        e2.comm = None
    else:
        obs = run(e2, mine, lambda e: e.window_close(tusec=5_000_000))
    e2.close()
    matrix, same = verify_ranks(obs, world, cdev)
    equal_single = None
    if rank == 0:
        e1 = SketchEngine(max_hosts=nh, max_services=nh * sp, max_clusters=16, enable_tdigest=True, max_batch_events=1 << 16, device=local_rank)
        want = run(e1, range(nh), lambda e: e.window_close(tusec=5_000_000))
        e1.close()
        equal_single = all(row == want for row in matrix)
    return {"hosts": nh, "services_per_host": sp, "events_per_host": nev, "ranks_consistent": same, "equals_single_rank_engine": equal_single}


def selftest_launch(args, rank, world):
    """--selftest-launch (no GPU needed): the ranks meet over gloo and run the same checksum exchange the timed run ends with, on
    synthetic registers -- covers the self-launch path and verify_ranks on a CPU box (tests/test_bench_launch.py)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    rng = np.random.default_rng(99)
    regs = [digest64(rng.integers(0, 255, 1 << 14, dtype=np.uint8)) for _ in REG_FAMILIES]
    if args.selftest_corrupt_rank == rank:
        regs[1] ^= 1
    matrix, same = verify_ranks(regs, world, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "exchange_check": {"ranks_seen": len(matrix), "ranks_consistent": same,
                                                                                  "families": REG_FAMILIES}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if same else 5


def conn_cpu_baseline(rec, ordered_by_host, est_distinct):
    """CPU legs of the C2 workload on a bounded sample (one 2^22-record chunk of the same bytes): (i) kind "port" = the
    oracle's walk into HyperLogLog + Count-Min registers (what the GPU registers are compared with bit for bit), one thread; (ii) kind
    "reference" = the EXACT answer from the reference's own classes (oracle/_ref: every record's PAIR_IP_PORT(nat_cli_, nat_ser_) into an
    unordered_set hashed with PAIR_IP_PORT::get_hash -- the stand-in for the RCU table glob_tcp_conn_tbl_ -- plus per-listener counters in an
    unordered_map<glob_id, ..., GY_JHASHER>), one thread and all host threads (records of one partha to one thread).  Medians of 5 / 3."""
    from oracle import oracle as o
    n = len(rec)
    buf = np.frombuffer(rec.tobytes(), dtype=np.uint8)
    end = buf.ctypes.data + len(buf)
    out = {"unit": "records/s", "sample": "%d TCP_CONN_NOTIFY records (one chunk of the timed run's bytes), warm, median of 5 (exact single-thread leg: of 3)" % n,
           "cores": 1, "kind": "port"}
    L = o.lib()
    rates = []
    for _ in range(5):
        hll = np.zeros(1 << 14, dtype=np.uint8)
        c32 = np.zeros(4 * 65536, dtype=np.uint32)
        c64 = np.zeros(4 * 65536, dtype=np.uint64)
        t0 = time.perf_counter()
        L.gyo_tcp_conn_sketch_batch(buf.ctypes.data, n, end, o.ptr(hll, o.u8p), o.ptr(c32, o.u32p), o.ptr(c64, o.u64p))
        rates.append(n / (time.perf_counter() - t0))
    out["value"] = sorted(rates)[2]
    R = o.ref()
    if R is not None and hasattr(R, "ref_conn_exact_new"):
        ex = {"kind": "reference", "form": "unordered_set<PAIR_IP_PORT, PAIR_IP_PORT::get_hash> of the NAT-translated tuples + unordered_map<glob_id, counters, GY_JHASHER>"}
        rates = []
        for _ in range(3):
            x = R.ref_conn_exact_new()
            t0 = time.perf_counter()
            R.ref_conn_exact_batch(x, buf.ctypes.data, n, end)
            rates.append(n / (time.perf_counter() - t0))
            ex["distinct_flows_exact"] = int(R.ref_conn_exact_distinct(x))
            R.ref_conn_exact_free(x)
        ex["value"] = sorted(rates)[1]
        ex["cores"] = 1
        if est_distinct is not None:
            ex["hll_estimate_engine_window"] = est_distinct
        ncores = os.cpu_count() or 1
        if ordered_by_host and ncores > 1:
            host_of = rec["nat_ser"]["ip32_be"].astype("<u4").view(">u4").astype(np.int64) & 0xFFFFFF
            first = np.concatenate([[0], np.flatnonzero(np.diff(host_of)) + 1]).astype(np.uint64)
            rates = []
            dd, nc = np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint64)
            for _ in range(5):
                t0 = time.perf_counter()
                R.ref_conn_exact_batch_mt(buf.ctypes.data, o.ptr(first, o.u64p), len(first), n, ncores, o.ptr(dd, o.u64p), o.ptr(nc, o.u64p))
                rates.append(n / (time.perf_counter() - t0))
            ex["allcores_value"] = sorted(rates)[2]
            ex["allcores"] = ncores
            ex["allcores_distinct_flows"] = int(dd[0])
            ex["allcores_connections"] = int(nc[0])
        out["reference_exact"] = ex
    return out


def run_conn(args, rank, world):
    """--workload conn: BASELINE.json configs[1] -- 1 000 hosts x 100 services, per window 2^24 device-resident
    TCP_CONN_NOTIFY records (280 B fixed stride; HLL distinct flows + 2 x Count-Min + exact per-service connection counters) and 10^5
    LISTENER_STATE_NOTIFY records (88 B; per-host LISTEN_SUMM_STATS roll-up, top-N), then the window close.  The records describe
    connections with a life cycle (open + close notifications, accepting and connecting halves, loopback: wire.synth_tcp_conns); after the
    timed region the per-service counters are checked against the generator's connection table (`checks`)."""
    import ctypes as C
    from gyeeta_amd import capi, wire
    from gyeeta_amd.engine import SketchEngine
    nh, sp = min(args.hosts, 1000), min(args.svcs, 100)
    nrec, chunk = 1 << 24, 1 << 22
    eng = SketchEngine(max_hosts=nh, max_services=nh * sp, enable_tdigest=False)
    s_ = np.arange(sp)
    for h in range(nh):
        mid = wire.machine_id(h)
        eng.register_host(mid, "cluster%d" % (h % 8))
        eng.register_listeners_np(mid, wire.glob_id(np.full(sp, h), s_), wire.listener_netns(h, s_), wire.listener_port(s_))
    rng = np.random.default_rng(1)
    truth = {}
    rec = wire.synth_tcp_conns(rng, chunk, np.arange(nh), sp, dup_frac=0.2, truth=truth)
    if args.conn_stream == "messages":
        # as madhava receives them: message after message, each from ONE partha (TCP_CONN_NOTIFY::MAX_NUM_CONNS = 2048 records per message):
        # the records grouped by the listener's host (its address is 10.<host>), ~2 messages per host and chunk
        host_of = rec["nat_ser"]["ip32_be"].astype("<u4").view(">u4").astype(np.int64) & 0xFFFFFF
        rec = rec[np.argsort(host_of, kind="stable")]
    # else: worst case for the per-workgroup aggregation: every record from a random host (the generator's own order)
    d = torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy()).cuda()
    off = torch.arange(0, chunk * 280, 280, dtype=torch.int32, device="cuda")
    ls = np.concatenate([wire.synth_listener_states(rng, h, s_) for h in range(nh)])
    dl = torch.from_numpy(np.frombuffer(ls.tobytes(), dtype=np.uint8).copy()).cuda()
    offl = torch.arange(0, len(ls) * 88, 88, dtype=torch.int32, device="cuda")
    hostl = torch.from_numpy(np.repeat(np.arange(nh, dtype=np.uint32), sp).view(np.int32).copy()).cuda()
    eng.order()

    def step(i):
        for r in range(nrec // chunk):
            capi.check(eng.L.gys_ingest_tcp_conn_dev(eng.h, C.c_void_p(d.data_ptr()), C.c_void_p(off.data_ptr()), chunk))
        capi.check(eng.L.gys_ingest_listener_state_dev(eng.h, C.c_void_p(dl.data_ptr()), C.c_void_p(offl.data_ptr()), C.c_void_p(hostl.data_ptr()), len(ls)))
        eng.window_close(tusec=5_000_000 * (i + 1))

    for i in range(args.warmup):
        step(i)
    eng.profile(True)
    eng.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = eng.profile_get()
    eng.profile(False)
    step_s = dt / args.steps
    alg = 280 * nrec + 88 * len(ls)
    kms = {k: v[0] / args.steps for k, v in prof.items()}
    name = args.sub or "c2_conn"
    kernels = {}
    for k, ms in kms.items():
        ent = {"ms": ms, "launches_per_step": prof[k][1] / args.steps}
        kb = {"conn": 280 * nrec, "lstate": 88 * len(ls)}.get(k)
        if kb and ms > 0:
            ent["algorithmic_bytes"] = kb
            ent["achieved_GBps"] = kb / (ms * 1e-3) / 1e9
            ent["frac"] = ent["achieved_GBps"] / HBM_PEAK_GBS
        tr = workload_traffic(name, k, nrec) if args.conn_stream == "messages" else None
        if tr is not None:
            ent["traffic"] = tr
        kernels[k] = ent
    # parity inside the bench: every window ingested the same chunk nrec / chunk times -> per-service counters = (windows x repeats) x the
    # generator's connection table, the walk tallies likewise
    windows = args.warmup + args.steps
    reps = windows * (nrec // chunk)
    ctr = eng.export_svc_counters()
    slot = truth["conn_host"].astype(np.int64) * sp + truth["conn_svc"]
    want_conn = np.bincount(slot, minlength=nh * sp) * reps
    want_close = np.bincount(slot[truth["conn_closed"]], minlength=nh * sp) * reps
    c = eng.counters()
    checks = {"connections_per_chunk": int(len(slot)), "records_per_chunk": chunk,
              "svc_nconn_equals_connection_table": bool((ctr[:, 0].astype(np.int64) == want_conn).all()),
              "svc_nclose_equals_connection_table": bool((ctr[:, 1].astype(np.int64) == want_close).all()),
              "walk_tallies": {k: int(c[k]) for k in ("conn_events", "conn_new", "conn_closed", "conn_closed_no_notify", "conn_client_side")},
              "tallies_consistent": bool(c["conn_events"] == c["conn_new"] + c["conn_closed"] == reps * chunk)}
    parity_ok = bool(checks["svc_nconn_equals_connection_table"] and checks["svc_nclose_equals_connection_table"] and checks["tallies_consistent"])
    conn_ent = kernels.get("conn", {})
    tr = conn_ent.get("traffic")
    out = {"metric": "TCP_CONN_NOTIFY records/sec ingested into HLL + Count-Min (BASELINE config 2)", "value": nrec / step_s, "unit": "records/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "int64", "data": "synthetic", "parity_ok": parity_ok, "checks": checks,
           "device_code": _device_code(),
           "config": {"workload": "C2: %d hosts x %d services, %d TCP_CONN_NOTIFY (280 B) + %d LISTENER_STATE_NOTIFY (88 B) records per window, "
                                  "device resident, 1 window per step; record order: %s" % (nh, sp, nrec, len(ls),
                                  "per-partha messages" if args.conn_stream == "messages" else "hosts mixed record by record")},
           "roofline": {"bound": "hbm", "achieved": alg / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / step_s / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_step": alg, "kernel": "conn", "kernel_avg_ms": kms.get("conn"), "kernel_frac": conn_ent.get("frac"),
                        "kernel_frac_measured_bytes": (tr["bytes"] / (kms["conn"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr and kms.get("conn") else None,
                        "traffic": tr, "kernels": kernels,
                        "note": "frac = (280 B x records + 88 B x listener records) / WHOLE step (SURVEY 8d: whole struct lines are fetched); kernel_frac = 280 B x "
                                "records / k_conn_ingest's HIP-event time; kernel_frac_measured_bytes = the same kernel's counter traffic (FETCH_SIZE x 2 + "
                                "WRITE_SIZE per step) / its time: k_conn_ingest asks for bytes [64, 224) and [264, 280) of each record only"}}
    if not args.no_cpu_baseline or args.sub:  # (a sub-run keeps this bounded CPU leg: ~2 s)
        try:
            est = None
            try:
                est = float(eng.distinct_flows())
            except Exception:
                pass
            out["cpu_baseline"] = conn_cpu_baseline(rec, args.conn_stream == "messages", est)
        except Exception as ex:  # never take the line down
            out["cpu_baseline"] = {"error": str(ex)[:300]}
    emit(out, args)
    eng.close()
    if not parity_ok:
        print("bench.py: connection counters differ from the generator's connection table", file=sys.stderr)
        sys.exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed windows (100 x ~10.4 ms: a timed region of about a second)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--hosts", type=int, default=10000, help="total hosts across all ranks")
    ap.add_argument("--svcs", type=int, default=1000, help="services per host")
    ap.add_argument("--events", type=int, default=1 << 29, help="events per rank per step (one window)")
    ap.add_argument("--zipf-milli", type=int, default=0, help="0 = uniform over services, else Zipf s*1000 (config 5: --zipf-milli 1100 --hosts 25 --svcs 4000)")
    ap.add_argument("--levels", type=int, nargs="?", const=1, default=0, choices=[0, 1, 2],
                    help="multi-level windows (gys_config.enable_levels): 1 = 5 s / 300 s / 5 d / all (every close folds every touched service), "
                         "2 = without the 5-s level (services touched only when a 30-s ring boundary is crossed)")
    ap.add_argument("--workload", choices=["resp", "conn"], default="resp", help="resp: C3/C4/C5 response-event stream (default); conn: C2 TCP_CONN_NOTIFY stream")
    ap.add_argument("--ipv6", action="store_true", help="the timed windows feed the same traffic as 48-B tcp_ipv6_resp_event_t events (gys_ingest_resp_events_v6_dev: "
                    "k_resp_host<..., MODE 2>; 2001:db8:: servers, fd00:: clients -- no embedded IPv4 address, the flows hash as 16-byte ends); not part of the default run")
    ap.add_argument("--conn-stream", choices=["messages", "mixed"], default="messages", help="--workload conn: per-partha 2048-record messages (default) or hosts mixed record by record")
    ap.add_argument("--exchange", choices=["rccl", "torch"], default="rccl", help="window exchange at N > 1: RCCL inside the library (default) or torch.distributed")
    ap.add_argument("--rccl-lib", default="torch", help="which RCCL the library binds at N > 1 (GYS_RCCL_LIB): 'torch' = the copy PyTorch bundles "
                    "(one RCCL per process; the ROCm 7.2 librccl's ncclCommInitRank does not return on part of the MI355X pool), 'rocm' = /opt/rocm/lib/librccl.so, or a path")
    ap.add_argument("--strict-exchange", action="store_true", help="exit non-zero when the in-library RCCL exchange was asked for but the run fell back to torch.distributed")
    ap.add_argument("--no-exchange-check", action="store_true", help="skip the (untimed) cross-rank register checks after an N > 1 run")
    ap.add_argument("--global-digest-every", type=int, default=-1, help="N > 1 with the in-library exchange: every K-th timed window also builds the GLOBAL "
                    "response-time digest across the ranks (gys_tdigest_global_rccl: this rank's roll-up slab over all its services, ncclAllGather, fold in "
                    "rank order) INSIDE the timed region; 0 = never.  It is a query-time operation (tens of ms at 10^6 services per rank: the roll-up walks "
                    "every service's digest), so the default is 0 for real runs -- there ONE exchange is made and timed after the timed region "
                    "(exchange_check.global_digest_ms) -- and 4 in the --share-device test mode")
    ap.add_argument("--share-device", action="store_true", help="test mode for a one-GPU box: every rank runs on device 0, the ranks meet over gloo and the "
                    "library's RCCL entry points are served by tests/cpp/fakerccl (RCCL refuses two ranks on one device); exercises the whole N > 1 flow")
    ap.add_argument("--selftest-launch", action="store_true", help="no GPU: only the launch path and the cross-rank checksum exchange (gloo)")
    ap.add_argument("--selftest-corrupt-rank", type=int, default=-1, help="--selftest-launch: this rank reports a wrong checksum (the run must fail)")
    ap.add_argument("--sub", default="", help="this run is the sub-run <name> of a default line (run_sub_configs): no CPU baseline, no host-fed leg, no quantile scan")
    ap.add_argument("--configs", default="auto", help="sub-runs of the other BASELINE configurations appended to the default line under `configs`: "
                    "'auto' = all of them when this is the default single-GPU workload, 'none', or a comma-separated list of c2_conn,c1,c5_zipf")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detail-out", default=os.path.join("gpurun_out", "bench_detail.json"), help="the full result (per-kernel tables, counter traffic, host-fed legs, "
                    "scans, sub-run lines, notes) as one strict-JSON file; 'none' = not written.  stdout's LAST line is the compact line (<= 4 KB)")
    ap.add_argument("--detail-stdout", action="store_true", help="also print the full result as an EARLIER stdout line")
    ap.add_argument("--selftest-line", default="", help="no GPU: read a full result from this JSON file and print the compact line made from it (tests/test_bench_line.py)")
    ap.add_argument("--no-quantile-check", action="store_true", help="skip the (untimed) t-digest rank-error check after the run")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the (untimed) host-fed measurement: pinned H2D copy + ingest")
    ap.add_argument("--no-dephase", action="store_true", help="skip the untimed pass that spreads the keys' buffer fill levels")
    ap.add_argument("--prime-windows", type=int, default=-1, help="untimed ordinary windows after the de-phase pass (-1: one buffer cycle)")
    ap.add_argument("--td-pend-cap", type=int, default=1920, help="gys_config.td_pend_cap: values a service's digest buffers before it is re-clustered (0 = the library default, 896; up to 3968).  "
                    "1920: merges of up to 2048 values -- half the merges per window of the 896-value buffer at 2/3 of their total time (profiles/r5e_*, r5f_*)")
    ap.add_argument("--nbuf", type=int, default=6, help="distinct device-resident event batches the windows cycle through")
    ap.add_argument("--cpu-events", type=int, default=1 << 26)
    ap.add_argument("--cpu-hosts", type=int, default=1000)
    ap.add_argument("--cpu-hosts-full", type=int, default=10000, help="hosts of the CPU legs at the metric's own key count (x --svcs keys; skipped when the host lacks the memory)")
    ap.add_argument("--cpu-events-full", type=int, default=1 << 25)
    args = ap.parse_args()

    if args.selftest_line:
        print(compact_line(_jsonable(json.load(open(args.selftest_line))), "gpurun_out/bench_detail.json"), flush=True)
        return
    if args.sub:
        args.no_cpu_baseline = args.no_host_fed = True
        args.configs = "none"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher around us: become one (one rank per GPU of this node)
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    if args.selftest_launch:
        sys.exit(selftest_launch(args, rank, world))
    if args.share_device:
        local_rank = 0
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} needs device {local_rank}, {torch.cuda.device_count() if torch.cuda.is_available() else 0} visible", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    cdev = torch.device("cpu") if args.share_device else torch.device("cuda", local_rank)  # where the few control-plane collectives run
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # one node: RCCL's bootstrap rendezvous over loopback (see gys_rccl_unique_id)
        os.environ.setdefault("NCCL_DEBUG", "WARN")        # a communicator that does not come up says why on stderr (captured by the launcher's log)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "GYS_RCCL_LIB" not in os.environ and args.rccl_lib != "rocm" and not args.share_device:  # before the library's first RCCL call (it binds with dlopen)
            cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so") if args.rccl_lib == "torch" else args.rccl_lib
            if os.path.exists(cand):
                os.environ["GYS_RCCL_LIB"] = cand
        if args.share_device:
            os.environ.setdefault("GYS_RCCL_LIB", os.path.join(ROOT, "tests", "cpp", "fakerccl", "libfakerccl.so"))
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    if args.workload == "conn":
        if rank == 0:
            run_conn(args, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    from gyeeta_amd import wire
    from gyeeta_amd.engine import SketchEngine, mid_buf
    from gyeeta_amd import capi
    L = capi.load()

    # shard the hosts by the reference's own machine-id hash
    mids = [wire.machine_id(h) for h in range(args.hosts)]
    mine = [h for h in range(args.hosts) if L.gys_shard_of(mid_buf(mids[h]), world) == rank]
    nlocal = len(mine)
    nsvc = nlocal * args.svcs
    eng = SketchEngine(max_hosts=max(nlocal, 1), max_services=max(nsvc, 1), max_clusters=16, enable_tdigest=True,
                       max_batch_events=args.events, rank=rank, nranks=world, device=local_rank, enable_levels=args.levels, td_pend_cap=args.td_pend_cap)
    # window exchange at N > 1: the four register families all-reduced INSIDE the library (gys_window_close_rccl: ncclAllReduce x 4 in
    # one group on the engine stream).  torch.distributed only carries the 128-byte communicator id (and the timing barrier).
    exchange = "none"
    rccl_join_stuck = False
    if world > 1:
        exchange = "torch.distributed"
        if args.exchange == "rccl":
            try:
                uid = torch.zeros(capi.RCCL_UID_BYTES, dtype=torch.uint8, device=cdev)
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(eng.rccl_unique_id()), dtype=torch.uint8))
                dist.broadcast(uid, src=0)
                # ncclCommInitRank is collective and cannot be interrupted: join on a helper thread and give it 60 s; if the
                # communicator does not come up on every rank the run continues on the torch.distributed path
                import threading
                res = {}

                def _join():
                    try:
                        eng.join_rccl(bytes(uid.cpu().numpy().tolist()))
                        res["ok"] = True
                    except Exception as ex2:  # noqa: BLE001
                        res["err"] = str(ex2)

                th = threading.Thread(target=_join, daemon=True)
                th.start()
                th.join(60.0)
                rccl_join_stuck = th.is_alive()
                if res.get("ok"):
                    exchange = "rccl_in_library"
                else:
                    raise RuntimeError(res.get("err", "ncclCommInitRank did not return within 60 s"))
            except Exception as ex:  # keep the run alive on the torch path
                print(f"bench.py rank {rank}: in-library RCCL unavailable ({ex}); using torch.distributed", file=sys.stderr)
        ok = torch.tensor([1 if exchange == "rccl_in_library" else 0], device=cdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # all ranks or none
        if int(ok.item()) == 0:
            exchange = "torch.distributed"
    close = eng.window_close_rccl if exchange == "rccl_in_library" else eng.window_close
    for c in range(8):  # same cluster order on every rank (gysketch.h: gys_register_cluster)
        eng.register_cluster("cluster%d" % c)
    s = np.arange(args.svcs)
    for j, h in enumerate(mine):
        slot = eng.register_host(mids[h], "cluster%d" % (h % 8))
        assert slot == j
        eng.register_listeners_np(mids[h], wire.glob_id(np.full(args.svcs, h), s), wire.listener_netns(j, s), wire.listener_port(s))
        if j % 64 == 0:
            eng.handle_host_state(mids[h], ntasks=100, nlisten=args.svcs)

    # two device-resident batches, generated on the GPU before the timed region
    nbuf = max(2, args.nbuf)  # distinct resident batches the windows cycle through (6 x 12.9 GB by default; 288 GB of HBM)
    if args.ipv6:
        nbuf = min(nbuf, 3)   # (the 48-byte batches are twice the size and are built next to the 24-byte ones: 3 x 25.8 GB beside the engine's 118 GB)
    buf_uses = [0] * nbuf
    bufs, segs = [], []
    for b in range(nbuf):
        ev = torch.empty(args.events * EVENT_BYTES, dtype=torch.uint8, device="cuda")
        segs.append(eng.gen_resp_events(ev.data_ptr(), args.events, 0x67796565746121 + 1000 * rank + b, 0, nlocal, args.svcs, args.zipf_milli))
        bufs.append(ev)
    eng.sync()

    # de-phase the per-key t-digest buffers (untimed set-up): with identical rates and an identical start every key would overflow
    # its GYS_TD_PEND_CAP-value buffer in the same window (one window of 10^7 merges, then ~14 windows of none).  One pass of PEND/2
    # events per key on average, drawn with per-service weights spread over 0..255/256, leaves the fill levels evenly spread, so that
    # from the first warm-up window on every window carries its long-run share of merges (events / ~(PEND + events per key and window)).
    ingested = []  # (nevents, seed, zipf/spread code, times): everything the engine was fed, for the quantile-error check
    PEND = eng.L.gys_td_pend_cap(eng.h)
    # every close is a collective at N > 1: the number of untimed windows must not depend on the rank's own share of the hosts
    nsvc_nominal = args.hosts * args.svcs // world
    if args.prime_windows < 0:  # one full buffer cycle: PEND values at events/keys values per window, plus one
        args.prime_windows = min(100, int(PEND * nsvc_nominal / max(args.events, 1)) + 2) if nsvc_nominal else 0
    if not args.no_dephase and nsvc_nominal:
        total = nsvc * (PEND // 2 - 1)
        nb = max(1, -(-total // args.events))
        per = min(args.events, total // nb)
        for b in range(nb if per else 0):
            sg = eng.gen_resp_events(bufs[0].data_ptr(), per, 0xdef0 + 77 * b + rank, 0, nlocal, args.svcs, 0xFFFFFFFF)
            eng.handle_resp_events_dev(sg, bufs[0].data_ptr(), per)
            ingested.append([per, 0xdef0 + 77 * b + rank, 0xFFFFFFFF, 1])
        close(tusec=0)
        segs[0] = eng.gen_resp_events(bufs[0].data_ptr(), args.events, 0x67796565746121 + 1000 * rank, 0, nlocal, args.svcs, args.zipf_milli)
        eng.sync()
        # ... and one full buffer cycle of ordinary windows, so that every key has re-clustered at least once: a production engine's
        # digests are never empty, and a key's very first merge (no clusters yet, all values in one interval) is its most expensive
        for i in range(args.prime_windows):
            b = i % nbuf
            eng.handle_resp_events_dev(segs[b], bufs[b].data_ptr(), args.events)
            close(tusec=0)
            buf_uses[b] += 1
        eng.sync()

    evb = EVENT_BYTES
    ingest_dev = eng.handle_resp_events_dev
    if args.ipv6:
        # the resident batches once more as IPv6 events (torch copies: set-up, untimed): words {saddr[4], daddr[4], netns, ports, lsndtime, lrcvtime}
        evb = 48
        ingest_dev = eng.handle_resp_events_v6_dev
        for b in range(nbuf):
            e4 = bufs[b].view(torch.int32).view(-1, 6)
            e6 = torch.zeros((args.events, 12), dtype=torch.int32, device="cuda")
            e6[:, 0] = 0xB80D0120 - (1 << 32)  # 20 01 0d b8: 2001:db8::/32
            e6[:, 3] = e4[:, 0]
            e6[:, 4] = 0x000000FD                # fd00::/8
            e6[:, 7] = e4[:, 1]
            e6[:, 8:12] = e4[:, 2:6]
            bufs[b] = e6.view(torch.uint8).view(-1)
            del e4, e6
        torch.cuda.synchronize()

    import ctypes as C
    if args.global_digest_every < 0:
        args.global_digest_every = 4 if args.share_device else 0
    gd_slab = None
    gd_calls = 0
    if exchange == "rccl_in_library":
        gd_slab = torch.zeros(C.sizeof(capi.TDigestSlab), dtype=torch.uint8, device="cuda")

    def step(i):
        nonlocal gd_calls
        b = i % nbuf
        ingest_dev(segs[b], bufs[b].data_ptr(), args.events)
        close(tusec=5_000_000 * (i + 1))
        if gd_slab is not None and args.global_digest_every > 0 and i % args.global_digest_every == args.global_digest_every - 1:
            # the fifth register family: per-(host, service) digests stay rank-local, the GLOBAL digest crosses the ranks as fixed-size slabs
            capi.check(eng.L.gys_tdigest_global_rccl(eng.h, eng.comm, C.c_void_p(gd_slab.data_ptr())))
            gd_calls += 1
        buf_uses[b] += 1

    for i in range(args.warmup):
        step(i)
    eng.profile(True)
    eng.profile_reset()
    ctr0 = eng.counters()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = eng.profile_get()
    eng.profile(False)
    ctr1 = eng.counters()

    # the exchange, checked (untimed): after the last window every rank must hold the same reduced registers, and a small side
    # configuration run through the same exchange must equal a single-rank engine fed all of its hosts
    xcheck = None
    if world > 1 and not args.no_exchange_check:
        regs = observe_registers(eng, ["cluster%d" % c for c in range(8)])
        matrix, same = verify_ranks(regs, world, cdev)
        side = exchange_side_check(args, rank, world, local_rank, L, eng, exchange, wire, SketchEngine, mid_buf, cdev)
        xcheck = {"ranks_seen": len(matrix), "ranks_consistent": same, "families": REG_FAMILIES, "side_config": side,
                  "ok": bool(same and side["ranks_consistent"] and side["equals_single_rank_engine"] is not False),
                  # how the exchange came up: which RCCL the library bound, whether ncclCommInitRank returned inside the watchdog's 60 s on
                  # this rank, and the global-digest exchanges that ran inside the timed region
                  "exchange": exchange, "rccl_lib": os.environ.get("GYS_RCCL_LIB", "/opt/rocm/lib/librccl.so"),
                  "rccl_join": "stuck (watchdog expired; torch.distributed used)" if rccl_join_stuck else ("ok" if exchange == "rccl_in_library" else "not used / failed"),
                  "global_digest_exchanges_timed": gd_calls, "global_digest_every": args.global_digest_every if gd_slab is not None else 0}
        # (gys_tdigest_global_rccl is collective: every rank makes the call below, in the same place)
        if gd_slab is not None:
            # the fifth register family once more, untimed and timed on its own: this rank's roll-up slab, all-gather, roll-up of the ranks' slabs
            torch.cuda.synchronize()
            tg = time.perf_counter()
            gsum = 0
            try:
                capi.check(eng.L.gys_tdigest_global_rccl(eng.h, eng.comm, C.c_void_p(gd_slab.data_ptr())))
                torch.cuda.synchronize()
                xcheck["global_digest_ms"] = (time.perf_counter() - tg) * 1e3
                gsum = digest64(gd_slab.cpu().numpy())
            except Exception as ex:  # reported, never fatal for the line (an N > 1 run on real devices has not been made by the builder)
                xcheck["global_digest_error"] = str(ex)[:300]
            # every rank folded the same slabs in the same order: the merged global digest must be identical on all ranks
            gm, gsame = verify_ranks([gsum], world, cdev)
            xcheck["global_digest_consistent"] = bool(gsame and "global_digest_error" not in xcheck)
            if "global_digest_error" not in xcheck:
                xcheck["ok"] = bool(xcheck["ok"] and gsame)
        okt = torch.tensor([1 if xcheck["ok"] else 0], device=cdev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)  # (rank 0 alone knows the single-rank comparison)
        xcheck["ok"] = bool(int(okt.item()))

    qerr = None
    if rank == 0 and not args.no_quantile_check and nsvc:
        for b in range(nbuf):
            ingested.append([args.events, 0x67796565746121 + 1000 * rank + b, args.zipf_milli, buf_uses[b]])
        qh, qs = ([mine[0]], [0]) if nlocal == 1 else ([mine[0], mine[-1]], [0, nlocal - 1])  # first and last host of the rank (one host: once)
        qerr = quantile_error(eng, torch, ingested, nlocal, args.svcs, qh, qs, wire)
    scan = None
    if rank == 0 and nsvc and not args.no_quantile_check and not args.sub:  # the per-key scan on the digests (a9): p25 / p95 / p99 of EVERY service, one pass
        eng.profile(True)
        eng.profile_reset()
        t0 = time.perf_counter()
        qv = eng.scan_quantiles([0.25, 0.95, 0.99])
        t_scan = time.perf_counter() - t0
        sp_ = eng.profile_get().get("scan_quantiles", (0.0, 0))
        eng.profile(False)
        scan = {"services": nsvc, "quantiles": [0.25, 0.95, 0.99], "kernel_ms": sp_[0], "wall_ms_incl_copy_to_host": t_scan * 1e3,
                "services_per_s": nsvc / (sp_[0] * 1e-3) if sp_[0] > 0 else None, "p99_mean_ms": float(qv[:, 2].mean())}
        try:  # the global response-time digest of this rank (what the C4 global query costs per rank before the slabs cross the ranks): every host's services rolled up, then the hosts
            # (round 6: the roll-up is the union by value bin -- HBM-bound adds, no ordered fold.  The FIRST call after a registration also lays the
            # hosts' member lists on the device (40 MB at 10^7 services); `global_rollup_ms` is the steady-state call, the first one is stated beside it.)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gdev, gslab = eng.tdigest_rollup(capi.ROLLUP_GLOBAL)
            scan["global_rollup_first_call_ms"] = (time.perf_counter() - t0) * 1e3
            eng.profile(True)
            eng.profile_reset()
            t0 = time.perf_counter()
            gdev, gslab = eng.tdigest_rollup(capi.ROLLUP_GLOBAL)
            scan["global_rollup_ms"] = (time.perf_counter() - t0) * 1e3
            pr = eng.profile_get()
            eng.profile(False)
            scan["global_rollup_kernels_ms"] = {k: v[0] for k, v in pr.items() if k.startswith("rollup")}
            scan["global_rollup_weight"] = int(gslab["cnt"].sum())
            gq = eng.slab_quantiles(gdev, [0.25, 0.5, 0.95, 0.99])
            scan["global_rollup_quantiles_ms"] = {"p25": gq[0], "p50": gq[1], "p95": gq[2], "p99": gq[3]}
        except Exception as ex:  # noqa: BLE001
            scan["global_rollup_error"] = str(ex)[:200]
    host_fed = None
    if rank == 0 and world == 1 and not args.no_host_fed and nsvc:
        host_fed = host_fed_rate(eng, torch, min(args.events, 1 << 26), nlocal, args.svcs)
        host_fed["l2_threads"] = "deferred"
    if rank == 0:
        total_events = args.events * world * args.steps
        value = total_events / dt
        # Roofline: `frac` is the WHOLE-STEP figure, 24 B x events / step time -- the conservative one.  Per-kernel
        # entries carry their own algorithmic bytes: the event kernel reads every 24-B event once and appends one 4-B staged word
        # per kept event; a t-digest merge reads the key's clusters (12 B x NB), its buffered words and writes the clusters back
        # (+ the 256-B window / all-time records it folds on the way); the per-service finalize pass reads one 4-B cursor per service.
        step_s = dt / args.steps
        alg_bytes = evb * args.events  # per step: every event of the window's batch is read exactly once
        kms = {k: v[0] / max(args.steps, 1) for k, v in prof.items()}
        merges_step = (ctr1["td_merges"] - ctr0["td_merges"]) / max(args.steps, 1)
        mvals_step = (ctr1["td_merge_values"] - ctr0["td_merge_values"]) / max(args.steps, 1)
        kalg = {"resp_host": alg_bytes + 4 * args.events,
                "key_finalize": 4 * nsvc + 16 * nsvc * min(1.0, args.events / max(nsvc, 1)),
                "digest_merge": merges_step * (2 * 12 * capi.TD_NB + 2 * 512 + 64 + 32) + 4 * mvals_step}
        kernels = {}
        for k, ms in kms.items():
            ent = {"ms": ms, "launches_per_step": prof[k][1] / max(args.steps, 1)}
            if k in kalg and ms > 0:
                ent["algorithmic_bytes"] = kalg[k]
                ent["achieved_GBps"] = kalg[k] / (ms * 1e-3) / 1e9
                ent["frac"] = ent["achieved_GBps"] / HBM_PEAK_GBS
            tr = workload_traffic(args.sub, k, args.events) if args.sub else (None if args.ipv6 else pmc_traffic(k, args.events, nsvc))  # (the committed passes are of the IPv4 stream)
            if tr is not None:
                ent["traffic"] = tr
            kernels[k] = ent
        dom = max(kms.items(), key=lambda kv: kv[1]) if kms else ("none", 0.0)
        dom_ent = kernels.get(dom[0], {})
        tol = 0.01
        parity_ok = None if qerr is None else bool(qerr["p50_rank_err_max"] <= tol and qerr["p99_rank_err_max"] <= tol)
        try:  # BASELINE.json names the metric; `value` is its events/sec half, `quantile_error` its p50/p99 half
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            metric = "events/sec ingested + p50/p99 quantile error vs reference"
        try:
            from gyeeta_amd.build import build_commit
            bc = build_commit()
        except Exception:
            bc = None
        out = {
            "metric": metric, "build_commit": bc, "kernel_sources": _kernel_sources(), "device_code": _device_code(),
            "value": value, "unit": "events/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic", "parity_ok": parity_ok,
            "config": {"workload": "C3/C4: %d hosts x %d services, raw %s stream, %s over services, "
                                   "1 window (ingest + window close) per step" % (args.hosts, args.svcs, "48-B tcp_ipv6_resp_event_t" if args.ipv6 else "24-B tcp_ipv4_resp_event_t",
                                                                                 "uniform" if not args.zipf_milli else "zipf %.2f" % (args.zipf_milli / 1000)),
                       "events_per_rank_per_step": args.events, "service_keys_total": args.hosts * args.svcs,
                       "service_keys_rank0": nsvc, "multi_level_windows": int(args.levels), "td_pend_cap": int(PEND),
                       "td_pend_cap_is_library_default": bool(int(PEND) == capi.TD_PEND_CAP),
                       "sketches": "exact RESP_TIME_HASH histogram + CONN_BITMAP + HLL p=14 + CMS 4x65536 + t-digest %d clusters + %d-value buffer per key" % (capi.TD_NB, PEND),
                       "parallelism": "host-id-hash shard x%d, RCCL all-reduce of registers per window" % world, "exchange": exchange,
                       "exchange_requested": args.exchange if world > 1 else "none",
                       "exchange_fallback": bool(world > 1 and args.exchange == "rccl" and exchange != "rccl_in_library"),
                       "rccl_lib": os.environ.get("GYS_RCCL_LIB", "/opt/rocm/lib/librccl.so") if world > 1 else None},
            "roofline": {"bound": "hbm", "achieved": alg_bytes / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_step": alg_bytes,
                         "note": "frac = %d B x events / WHOLE step (all kernels of the window); per-kernel figures under `kernels`" % evb,
                         "kernel": dom[0], "kernel_avg_ms": dom[1], "kernel_frac": dom_ent.get("frac"),
                         # SURVEY 8(d)'s own figure for the dominant kernel: 24 B x events / its average launch time (kernel_frac counts the
                         # 4-byte staged word the event kernel also writes per event: 28 B)
                         "kernel_frac_24B": (alg_bytes / (dom[1] * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom[0] == "resp_host" and dom[1] > 0 else None,
                         "traffic": dom_ent.get("traffic"),
                         "merges_per_step": merges_step, "merge_values_per_step": mvals_step,
                         "kernels": kernels},
        }
        if xcheck is not None:
            out["exchange_check"] = xcheck
        if qerr is not None:
            out["quantile_error"] = qerr
        if host_fed is not None:
            out["host_fed"] = host_fed
        if scan is not None:
            out["quantile_scan"] = scan
        if not args.no_cpu_baseline and world == 1:  # the CPU legs are timed on rank 0 at N = 1 only
            # the metric's own key count first (10^7 keys when the workload has them and the host has the memory: median of 5), then the
            # 10^6-key sample of the earlier rounds beside it
            legs = []
            big_hosts = min(args.cpu_hosts_full, args.hosts)
            tcb = time.perf_counter()
            if big_hosts > args.cpu_hosts:  # 3 timed repetitions after one untimed pass; no hist-only leg at this size (the reference's own loop is timed instead)
                legs.append(("full_keys", cpu_baseline(eng, big_hosts, args.svcs, args.cpu_events_full, 0x1235, td_cap=args.td_pend_cap, reps=3, histonly=False, mt_reps=3)))
            legs.append(("sample", cpu_baseline(eng, min(args.cpu_hosts, args.hosts), args.svcs, args.cpu_events, 0x1234, td_cap=args.td_pend_cap, reps=1)))
            legs = [(k, v) for k, v in legs if v is not None]
            if legs:
                name, m = legs[0]
                gpu_epk = args.events / max(nsvc, 1)
                out["cpu_baseline"] = {
                    "value": m["port"], "unit": "events/s", "cores": 1, "kind": "port", "keys": m["keys"], "events": m["events"],
                    "events_per_key": m["events_per_key"], "gpu_events_per_key_per_window": gpu_epk, "leg": name,
                    "sample_short": ("value = oracle port (hist+bitmap+HLL+CMS+t-digest), 1 core, %d events x %d keys (%s), td_pend_cap %d (bench setting; library default 896), median of %d; "
                                     "reference_hist = Gyeeta's GY_HISTOGRAM loop (oracle/_ref)" % (m["events"], m["keys"], "the metric's key count; %.1f events/key per pass vs %.0f on the GPU" %
                                                                                            (m["events_per_key"], gpu_epk) if name == "full_keys" else "a tenth of the GPU run's keys", m["td_pend_cap"], m["runs"])),
                    "sample": ("%d events over %d service keys (%s), td_pend_cap %d; one core: median of %d run(s); all cores (%d threads): median of 3-5; gcc -O2; "
                               "value = the full port (hist + bitmap + HLL + CMS + t-digest), histonly = the reference's own per-event work, reference_hist = "
                               "the reference's GY_HISTOGRAM loop compiled from its sources (oracle/_ref).  `value` is the leg named by `leg`: full_keys = the GPU run's own "
                               "10^7 keys at FEWER events per key and pass than a GPU window carries (events_per_key vs gpu_events_per_key_per_window: a colder, nearly merge-free "
                               "shape -- the per-key state lives in DRAM either way); `matched_ratio` = the 10^6-key sample whose events per key match a GPU window's")
                              % (m["events"], m["keys"], "the metric's own key count" if name == "full_keys" else "a tenth of the GPU run's keys", m["td_pend_cap"], m["runs"], m["cores_all"])}
                cb = out["cpu_baseline"]
                if "histonly" in m:
                    cb["histonly_value"] = m["histonly"]
                if "port_allcores" in m:  # the same full port on all host threads
                    cb["allcores_value"], cb["allcores"], cb["allcores_form"] = m["port_allcores"], m["cores_all"], m["port_allcores_form"]
                if "reference_hist" in m:
                    cb["reference_hist_value"], cb["reference_hist_kind"] = m["reference_hist"], "reference"
                if "reference_hist_allcores" in m:
                    cb["reference_hist_allcores_value"], cb["reference_hist_allcores"] = m["reference_hist_allcores"], m["cores_all"]
                for k2, m2 in legs[1:]:  # the other sample, whole (its events per key are those of a GPU window)
                    cb["matched_ratio"] = {"value": m2["port"], "keys": m2["keys"], "events": m2["events"], "events_per_key": m2["events_per_key"],
                                           "allcores_value": m2.get("port_allcores"), "reference_hist_value": m2.get("reference_hist"),
                                           "reference_hist_allcores_value": m2.get("reference_hist_allcores"), "histonly_value": m2.get("histonly")}
                cb["wall_s"] = time.perf_counter() - tcb
                cb["phase_s"] = {k2: m2.get("phase_s") for k2, m2 in legs}
    eng.leave_rccl()
    eng.close()
    if rank == 0:
        if isinstance(out.get("host_fed"), dict) and out["host_fed"].get("l2_threads") == "deferred":
            out["host_fed"]["l2_threads"] = host_fed_l2_threads()  # its own context: after this engine has released the device
        # the other BASELINE configurations (C2 connection records, C1 and C5 shapes) as short sub-runs, each in its own process: appended to
        # the default single-GPU line only (a sub-run never changes `value`; its failure is reported inside `configs`)
        default_shape = (world == 1 and args.hosts == 10000 and args.svcs == 1000 and args.events == (1 << 29) and not args.zipf_milli and not args.levels and not args.ipv6)
        if args.configs != "none" and (default_shape or args.configs != "auto"):
            bufs.clear()  # (this run's resident event batches: the sub-runs bring their own)
            torch.cuda.empty_cache()
            out["configs"] = run_sub_configs([] if args.configs in ("auto", "all") else args.configs.split(","))
        out["wall_s"] = time.perf_counter() - T_START
        emit(out, args)
    bad = rank == 0 and out.get("parity_ok") is False
    xbad = xcheck is not None and not xcheck["ok"]
    fell_back = world > 1 and args.exchange == "rccl" and exchange != "rccl_in_library"
    if world > 1:
        dist.barrier()  # rank 0's untimed checks take longer than the other ranks' exit path: leave together
        dist.destroy_process_group()
    if bad:  # the north-star tolerance is part of the metric: a line that breaks it is not a result
        print("bench.py: t-digest rank error above the 0.01 tolerance", file=sys.stderr)
        sys.exit(3)
    if xbad:  # ranks that disagree after the exchange (or differ from the single-rank engine): the N-GPU line is not a result either
        print("bench.py: cross-rank register check failed (see exchange_check in the JSON line)", file=sys.stderr)
        sys.exit(5)
    if fell_back:
        print("bench.py: the in-library RCCL exchange was not available; the run used torch.distributed (exchange_fallback in the JSON line)", file=sys.stderr)
        if args.strict_exchange:
            sys.exit(4)
    if rccl_join_stuck:  # a helper thread is still inside ncclCommInitRank: do not let interpreter shutdown wait on it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()

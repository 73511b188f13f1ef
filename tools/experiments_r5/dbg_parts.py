"""debug: the scenario of tests/test_gpu_round5.py::test_device_batches_of_both_families_many_hosts_and_parts with per-batch diagnostics"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
from oracle import oracle
from tests import test_gpu_round5 as t
from gyeeta_amd import capi
from gyeeta_amd.engine import SketchEngine

rng = np.random.default_rng(52)
nh = 6
big = int(sys.argv[1]) if len(sys.argv) > 1 else 2600
eng = SketchEngine(max_hosts=8, max_services=8192, max_batch_events=1 << 20)
orc = oracle.OracleEngine(8192)
worlds = [t.World(eng, orc, [h], big if h == 2 else 150) for h in range(nh)]
info = {h: worlds[h].info[h] for h in range(nh)}
bounds = np.cumsum([0] + [(big if h == 2 else 150) + 9 for h in range(nh)])
for rnd in range(3):
    for fam in (0, 1):
        parts, segs_h, first = [], [], 0
        segs = (capi.RespSeg * nh)()
        for i, h in enumerate(rng.permutation(nh)):
            n = int(rng.integers(2000, 60000 if h == 2 else 20000))
            ev = worlds[h].events4(rng, h, n) if fam == 0 else worlds[h].events6(rng, h, n)
            parts.append(ev.tobytes())
            segs[i].host_slot, segs[i].first_event = info[h][1], first
            segs_h.append((info[h][1], first))
            first += n
        raw = b"".join(parts)
        d = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
        if fam == 0:
            eng.handle_resp_events_dev(segs, d.data_ptr(), first)
            orc.resp_batch(raw, [s for s, _ in segs_h], [f for _, f in segs_h])
        else:
            eng.handle_resp_events_v6_dev(segs, d.data_ptr(), first)
            orc.resp_batch_v6(raw, [s for s, _ in segs_h], [f for _, f in segs_h])
        eng.sync()
        c = eng.counters()
        g = eng.export_hist(0, 0, orc.nsvc)[:, 15, 0]
        o = orc.hist()[:, 15, 0]
        bad = np.nonzero(g != o)[0]
        oc = orc.counters()
        print("rnd", rnd, "fam", fam, "n", first, "general", c["resp_batches_general"], "hostlocal", c["resp_batches_host_local"], "split", c.get("resp_batches_host_split"),
              "counters gpu", c["resp_events"], c["resp_dropped_range"], c["resp_dropped_nolistener"], "orc", oc["events"], oc["dropped_range"], oc["dropped_nolistener"],
              "bad slots", len(bad), flush=True)
        for s in bad[:12]:
            h = int(np.searchsorted(bounds, s, side="right") - 1)
            print("   slot", s, "host", h, "rel", s - bounds[h], "gpu", g[s], "orc", o[s])
        gn, gp = eng.export_tdigest_pending(0, orc.nsvc)
        on, op = orc.td_pending()
        print("   pending fill mismatch:", int((gn != on).sum()), "sum gpu", int(gn.sum()), "orc", int(on.sum()))

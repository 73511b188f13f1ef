import ctypes as C, numpy as np, sys
sys.path.insert(0, '/root/repo')
from tests import helpers
from oracle import oracle
from gyeeta_amd.engine import SketchEngine
rng = np.random.default_rng(1)
nh, sp = 3, 7
eng = SketchEngine(max_hosts=8, max_services=64, max_batch_events=1 << 16, resp_path=1)
orc = oracle.OracleEngine(64)
info, gids = helpers.register_world(eng, orc, range(nh), sp)
for rnd in range(4):
    for h in range(nh):
        n = int(rng.integers(1, 3000))
        ev = helpers.make_resp_events(rng, h, n, sp)
        eng.handle_resp_events(info[h][0], ev)
        orc.resp_batch(ev.tobytes(), [info[h][1]], [0])
eng.sync()
g = int(gids[0][0]); slot = eng.lookup(g)
print("slot", slot)
gs, gc, gm = eng.export_tdigest(slot, 1)
gn, gp = eng.export_tdigest_pending(slot, 1)
b = orc.td(slot)
print("npend gpu", gn, "orc", b.npend, "minmax", gm, b.d.vmin, b.d.vmax)
view = oracle.TDigest(); oracle.lib().gyo_tdb_merged_view(C.byref(b), C.byref(view))
txt = eng.tdigest_sql_text(g)
print(txt[:400])
oc = [(view.sum[i]/view.cnt[i], view.cnt[i]) for i in range(oracle.TD_NB) if view.cnt[i]]
print(len(oc), oc[:12])
print(eng.quantiles(g, [0.25,0.5,0.9]), [oracle.lib().gyo_tdb_quantile(C.byref(b), q) for q in (0.25,0.5,0.9)])
qs = [0.0, 0.01, 0.25, 0.5, 0.9, 0.99, 1.0]
for h in range(nh):
    for s in range(sp):
        g = int(gids[h][s]); slot = eng.lookup(g)
        b = orc.td(slot)
        gq = eng.quantiles(g, qs); oq = [oracle.lib().gyo_tdb_quantile(C.byref(b), q) for q in qs]
        gs, gc, gm = eng.export_tdigest(slot, 1)
        print(h, s, slot, "npend", b.npend, "ncl", int((gc[0] != 0).sum()), "N", int(gc[0].sum()), "OK" if gq == oq else ("BAD", gq, oq))
        if gq != oq:
            view = oracle.TDigest(); oracle.lib().gyo_tdb_merged_view(C.byref(b), C.byref(view))
            oc = [(view.sum[i], view.cnt[i]) for i in range(oracle.TD_NB) if view.cnt[i]]
            txt = eng.tdigest_sql_text(g)
            print(" orc", len(oc), oc[:10], "\n gpu", txt[:300])
            break

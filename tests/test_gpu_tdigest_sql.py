"""a service's t-digest in the external forms of the Postgres tdigest type (the reference aggregates percentiles with
public.tdigest / public.tdigest_percentile, common/gy_query_common.cc:1818-1855).  The extension is not in /root/reference (unpinned
third party) => the forms are restated from its published I/O functions and checked here against the CPU oracle's digest written
the same way, plus the invariants the type's input function enforces (positive count, count == sum of centroid counts, centroid
number within the compression's bound, means in ascending order)."""
import ctypes as C
import re
import struct

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def _oracle_centroids(oracle, orc, slot):
    L = oracle.lib()
    v = oracle.TDigest()
    L.gyo_tdb_merged_view(C.byref(orc.td(slot)), C.byref(v))
    return [(v.sum[i] / v.cnt[i], int(v.cnt[i])) for i in range(oracle.TD_NB) if v.cnt[i]]


def test_tdigest_sql_text_and_binary(oracle):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: -m gpu tests must run on the MI355X box")
    from gyeeta_amd import capi
    from gyeeta_amd.engine import SketchEngine
    rng = np.random.default_rng(123)
    nh, sp = 2, 5
    eng = SketchEngine(max_hosts=4, max_services=32, max_batch_events=1 << 16)
    orc = oracle.OracleEngine(32)
    info, gids = helpers.register_world(eng, orc, range(nh), sp)
    # service 4 of host 1 never gets an event; the others get between a handful (all buffered) and several merges' worth
    for rnd, n in enumerate([30, 700, 5000, 12000]):
        for h in range(nh):
            ev = helpers.make_resp_events(rng, h, n, sp if h == 0 else sp - 1, lat_mu=2.0 + 0.5 * rnd)
            eng.handle_resp_events(info[h][0], ev)
            orc.resp_batch(ev.tobytes(), [info[h][1]], [0])
        for h in range(nh):
            for s in range(sp if h == 0 else sp - 1):
                gid, slot = int(gids[h][s]), h * sp + s
                cents = _oracle_centroids(oracle, orc, slot)
                total = sum(c for _, c in cents)
                want = "flags 1 count %d compression %d centroids %d" % (total, oracle.TD_NB, len(cents)) + "".join(" (%f, %d)" % mc for mc in cents)
                got = eng.tdigest_sql_text(gid)
                assert got == want, (rnd, h, s)
                # what tdigest_in checks
                m = re.fullmatch(r"flags (\d+) count (\d+) compression (\d+) centroids (\d+)((?: \([-0-9.]+, \d+\))+)", got)
                assert m and int(m.group(1)) == 1 and int(m.group(3)) == oracle.TD_NB and 0 < int(m.group(4)) <= 10 * oracle.TD_NB
                pairs = [(float(a), int(b)) for a, b in re.findall(r"\(([-0-9.]+), (\d+)\)", m.group(5))]
                assert len(pairs) == int(m.group(4)) and sum(b for _, b in pairs) == int(m.group(2)) > 0
                assert all(pairs[i][0] <= pairs[i + 1][0] for i in range(len(pairs) - 1))
                raw = eng.tdigest_sql_binary(gid)
                assert raw == struct.pack(">iqii", 1, total, oracle.TD_NB, len(cents)) + b"".join(struct.pack(">dq", mean, c) for mean, c in cents)
    with pytest.raises(capi.GysError) as ei:   # no values yet: the type has no empty literal
        eng.tdigest_sql_text(int(gids[1][sp - 1]))
    assert ei.value.code == capi.ERR_NOTFOUND
    need = C.c_size_t()
    small = C.create_string_buffer(16)
    assert eng.L.gys_tdigest_sql_text(eng.h, int(gids[0][0]), small, 16, C.byref(need)) == capi.ERR_NOMEM and need.value > 16
    eng.close()
    eng2 = SketchEngine(max_hosts=2, max_services=8, max_batch_events=1 << 10, enable_tdigest=False)
    helpers.register_world(eng2, None, range(1), 2)
    with pytest.raises(capi.GysError) as ei:
        eng2.tdigest_sql_text(int(gids[0][0]))
    assert ei.value.code == capi.ERR_STATE
    eng2.close()

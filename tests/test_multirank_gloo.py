"""The N>1 path on CPU (gloo, world_size 2): hosts are sharded by the reference's machine-id hash, every rank builds the
registers of ITS shard, and gyeeta_amd.engine.allreduce_sections -- the same function the GPU engine calls at the window
boundary with RCCL -- reduces them (HLL = max on u8, CMS / histogram / cluster counters = sum, max_val = max).  The reduced
registers must equal the single-rank registers of the whole stream (bit exact)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from gyeeta_amd import wire
from tests import helpers

WORLD = 2
NH, SP, NEV = 12, 6, 3000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stream():
    rng = np.random.default_rng(77)
    return {h: helpers.make_resp_events(rng, h, NEV, SP) for h in range(NH)}


def _registers(orc_mod, hosts, events):
    """oracle registers for a set of hosts, laid out like the engine's reduce sections"""
    orc = orc_mod.OracleEngine(NH * SP)
    slot_of = {}
    for h in hosts:
        slot_of[h] = len(slot_of)
        s = np.arange(SP)
        g = wire.glob_id(np.full(SP, h), s)
        for i in range(SP):
            orc.register(slot_of[h], int(g[i]), int(wire.listener_netns(h, s)[i]), int(wire.listener_port(s)[i]))
    for h in hosts:
        orc.resp_batch(events[h].tobytes(), [slot_of[h]], [0])
    gh, gmax = orc.ghist()
    cluster = np.zeros((4, 12), dtype=np.int64)
    for h in hosts:  # STATE_ONE-like sums per cluster: nhosts + total events as stand-ins for the 11 u32 counters
        cluster[h % 4, 0] += 1
        cluster[h % 4, 7] += NEV
    return (torch.from_numpy(orc.hll().copy()), torch.from_numpy(np.concatenate([orc.cms().ravel().view(np.int32), cluster.ravel().astype(np.int32)])),
            torch.from_numpy(gh.ravel().copy()), torch.tensor([gmax], dtype=torch.int64))


def _worker(rank, port, q):
    import torch.distributed as dist
    from gyeeta_amd import capi
    from gyeeta_amd.engine import allreduce_sections, mid_buf
    from oracle import oracle as o
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        L = capi.load()
        events = _stream()
        mine = [h for h in range(NH) if L.gys_shard_of(mid_buf(wire.machine_id(h)), WORLD) == rank]
        hll, u32s, i64s, i64m = _registers(o, mine, events)
        allreduce_sections([(hll, 0), (u32s, 1), (i64s, 1), (i64m, 0)])
        fh, fu, fi, fm = _registers(o, list(range(NH)), events)
        ok = bool((hll == fh).all() and (u32s == fu).all() and (i64s == fi).all() and (i64m == fm).all())
        q.put((rank, ok, len(mine)))
    finally:
        dist.destroy_process_group()


def test_window_reduce_world2_gloo():
    from gyeeta_amd import build
    if build.needs_build():
        build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(WORLD)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sum(n for _, _, n in res) == NH and all(n > 0 for _, _, n in res)  # both shards non-empty, every host owned exactly once


def test_allreduce_sections_is_noop_without_process_group():
    from gyeeta_amd.engine import allreduce_sections
    t = torch.arange(4, dtype=torch.int32)
    allreduce_sections([(t, 1)])
    assert t.tolist() == [0, 1, 2, 3]

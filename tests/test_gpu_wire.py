"""Wire front-end: a byte stream of COMM_HEADER-framed partha messages, variable-stride records decoded on the GPU.
The resulting state must equal what the per-batch entry points (host-walked offsets) produce from the same records, and malformed
input must be rejected the way COMM_HEADER::validate / TCP_CONN_NOTIFY::validate / LISTENER_STATE_NOTIFY::validate reject it."""
import numpy as np
import pytest

from gyeeta_amd import wire

pytestmark = pytest.mark.gpu


def _tails(rng, n, maxlen, frac):
    return [bytes(rng.integers(32, 127, int(k), dtype=np.uint8).tolist()) for k in rng.integers(0, maxlen + 1, n) * (rng.random(n) < frac)]


def _world(resp=False):
    from gyeeta_amd.engine import SketchEngine
    eng = SketchEngine(max_hosts=4, max_services=64, enable_tdigest=False)
    for h in range(3):
        mid = wire.machine_id(h)
        eng.register_host(mid, "c%d" % (h % 2))
        s = np.arange(12)
        eng.register_listeners_np(mid, wire.glob_id(np.full(12, h), s), wire.listener_netns(h, s), wire.listener_port(s))
    return eng


def _state(eng):
    return (eng.export_hll().tobytes(), eng.export_cms(0).tobytes(), eng.export_cms(1).tobytes(), eng.export_svc_counters().tobytes(),
            tuple(eng.svcsumm(wire.machine_id(h)).as_tuple() for h in range(3)),
            # (the counters of what was ingested -- not of how the host-pointer calls were queued and submitted)
            {k: v for k, v in eng.counters().items() if not k.endswith(("_queued", "_submissions", "_flushes", "stage_waits"))},
            # the kept LISTENER_STATE record of every listener: a stream that holds several messages of one partha names a listener several
            # times in ONE launch -- the last record in stream order stays, whole, as after the per-message calls (k_lstate_keep)
            tuple(eng.json_svcstate(wire.machine_id(h)) for h in range(3)))


def test_comm_stream_equals_batch_entry_points():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    rng = np.random.default_rng(8)
    a, b = _world(), _world()
    for h in range(3):
        mid = wire.machine_id(h)
        stream = b""
        nmsg = {"conn": 0, "lst": 0, "skip": 0}
        nrec = 0
        for rnd in range(5):
            # TCP_CONN_NOTIFY message: up to MAX_NUM_CONNS = 2048 records, 30 % with a command line tail (variable stride)
            n = int(rng.integers(1, 2049)) if rnd else 2048
            rec = wire.synth_tcp_conns(rng, n, [0, 1, 2, 9], 12, dup_frac=0.2, v6_frac=0.1)
            payload = wire.pack_variable(rec, _tails(rng, n, 256, 0.3))
            stream += wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n, payload)
            b.partha_tcp_conn_info(mid, payload, n)
            nmsg["conn"] += 1
            nrec += n
            # a control-plane message in between (skipped) and an unrelated notify subtype
            stream += wire.frame_event_notify(wire.NOTIFY_CPU_MEM_STATE, 1, bytes(rng.integers(0, 256, 40, dtype=np.uint8).tolist()))
            stream += wire.frame_event_notify(0, 0, bytes(48), data_type=wire.COMM_QUERY_CMD)
            nmsg["skip"] += 2
            # LISTENER_STATE_NOTIFY message with issue strings
            ls = wire.synth_listener_states(rng, h, rng.permutation(12)[:int(rng.integers(1, 13))], delete_frac=0.1, bad_state_frac=0.05)
            payload = wire.pack_variable(ls, _tails(rng, len(ls), 254, 0.5))
            stream += wire.frame_event_notify(wire.NOTIFY_LISTENER_STATE, len(ls), payload)
            b.partha_listener_state(mid, payload, len(ls))
            nmsg["lst"] += 1
            nrec += len(ls)
        # a zero-record message and a trailing partial header (left to the caller)
        stream += wire.frame_event_notify(wire.NOTIFY_TCP_CONN, 0, b"")
        nmsg["conn"] += 1
        st = a.handle_comm_stream(mid, stream + b"\x05\x66\x66")
        assert (st.nmsgs_tcp_conn, st.nmsgs_listener_state, st.nmsgs_skipped, st.nmsgs_invalid) == (nmsg["conn"], nmsg["lst"], nmsg["skip"], 0)
        assert st.nrecords == nrec and st.bytes_consumed == len(stream)
    for e in (a, b):
        e.window_close()
    assert _state(a) == _state(b)
    a.close()
    b.close()


def test_comm_stream_rejects_malformed():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    from gyeeta_amd import capi
    rng = np.random.default_rng(9)
    eng = _world()
    mid = wire.machine_id(0)
    n = 40
    rec = wire.synth_tcp_conns(rng, n, [0], 12)
    payload = wire.pack_variable(rec, _tails(rng, n, 64, 0.5))
    good = wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n, payload)
    before = eng.counters()["conn_events"]

    def rejected(stream):
        with pytest.raises(capi.GysError) as ei:
            eng.handle_comm_stream(mid, stream)
        assert ei.value.code == capi.ERR_INVAL
        assert eng.counters()["conn_events"] == before  # nothing ingested

    rejected(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n, payload, magic=0x05777705))       # not a partha-to-madhava connection
    rejected(good[:4] + np.array([len(good) + 4], dtype="<u4").tobytes() + good[8:])             # total_sz_ not a multiple of 8 / overruns
    rejected(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, 2049, payload))                       # nevents_ > MAX_NUM_CONNS
    rejected(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n + 1, payload))                      # fewer records than nevents_
    rejected(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, 2048, b""))                           # a header-only message announcing 2 048 records
    rejected(good + wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n, payload[:280 * n - 8]))     # more records than the payload can hold, behind a good message
    bad = bytearray(payload)
    bad[272:274] = (3).to_bytes(2, "little")  # first record: cmdline 3 + padding 0 -> size not a multiple of 8 ("Padding issue")
    bad[279] = 0
    rejected(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n, bytes(bad)))
    bad = bytearray(payload)
    last = len(payload) - 280  # the last record has no tail here only if its tail is empty; force an overrun of the message end
    bad[272:274] = (4000).to_bytes(2, "little")
    rejected(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n, bytes(bad)))
    # the well-formed message still goes through afterwards
    st = eng.handle_comm_stream(mid, good)
    assert st.nrecords == n and eng.counters()["conn_events"] == before + n
    eng.close()


def test_comm_stream_cut_inside_a_message_resumes():
    """a recv() chunk that ends inside a message body (complete 16-byte COMM_HEADER, incomplete body): the whole messages before it are
    ingested, bytes_consumed points at that header, and feeding the rest from there gives the same state as the uncut stream"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    rng = np.random.default_rng(18)
    a, b = _world(), _world()
    mid = wire.machine_id(1)
    msgs = []
    for n in (700, 5, 2048):
        rec = wire.synth_tcp_conns(rng, n, [0, 1, 2], 12, dup_frac=0.1)
        msgs.append(wire.frame_event_notify(wire.NOTIFY_TCP_CONN, n, wire.pack_variable(rec, _tails(rng, n, 128, 0.4))))
    ls = wire.synth_listener_states(rng, 1, np.arange(12))
    msgs.append(wire.frame_event_notify(wire.NOTIFY_LISTENER_STATE, len(ls), wire.pack_variable(ls, _tails(rng, len(ls), 100, 0.5))))
    stream = b"".join(msgs)
    whole = b.handle_comm_stream(mid, stream)
    assert whole.bytes_consumed == len(stream) and whole.nrecords == 700 + 5 + 2048 + 12
    # cut 40 bytes into the body of the third message (8-byte aligned chunk), then 3 bytes into the header of the fourth
    cut1 = len(msgs[0]) + len(msgs[1]) + 16 + 40
    st = a.handle_comm_stream(mid, stream[:cut1])
    assert st.bytes_consumed == len(msgs[0]) + len(msgs[1]) and st.nrecords == 705 and st.nmsgs_invalid == 0
    pos = st.bytes_consumed
    cut2 = len(stream) - len(msgs[3]) + 3
    st = a.handle_comm_stream(mid, stream[pos:cut2])
    assert st.bytes_consumed == len(msgs[2]) and st.nrecords == 2048
    pos += st.bytes_consumed
    st = a.handle_comm_stream(mid, stream[pos:])
    assert st.bytes_consumed == len(msgs[3]) and st.nrecords == 12
    for e in (a, b):
        e.window_close()
    assert _state(a) == _state(b)
    a.close()
    b.close()

"""cc_db_add_scans_prepare (the asynchronous first half of an append) on the CPU harness: a DB built from prepared
batches -- also two batches prepared ahead -- answers queries exactly like one built with plain cc_db_add_scans, and
out-of-order use is refused with the handle left usable."""
import ctypes as C

import numpy as np

import emu_api


def _seq(cc, oracle, n=48):
    L = oracle.L
    d = L.default_db_cfg()
    d.max_elapse, d.min_elapse = 2.5, 1.5
    w = cc.synth.World(loop_len=40.0)
    x, _, ts = cc.synth.make_sequence(n, world=w, beams=16, azim=450)
    xs = x.numpy().reshape(-1, 4)
    offs = np.arange(n + 1, dtype=np.int64) * x.shape[1]
    seeds = np.arange(n, dtype=np.int32)
    ores, _, odesc = oracle.run_sequence(xs, offs, ts, seeds, dcfg=d, want_desc=True)
    return L, d, odesc, ts, seeds, ores


def test_prepared_batches_equal_plain_adds(cc, oracle):
    L, d, desc, ts, seeds, ores = _seq(cc, oracle)
    n = len(desc)
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    db_a = api.db_create(ctx, d, cap=n)
    db_b = api.db_create(ctx, d, cap=n)
    cuts = [0, 10, 22, 23, 40, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        api.db_add(db_a, desc[a:b], ts[a:b], seeds[a:b])
    # two batches prepared ahead, then added in order; the rest one by one
    d0, d1 = np.ascontiguousarray(desc[0:10]), np.ascontiguousarray(desc[10:22])
    for dd in (d0, d1):
        api.chk(api.lib.cc_db_add_scans_prepare(db_b, C.c_void_p(dd.ctypes.data), len(dd), None), "prepare")
    # a third one does not fit, and adding something else first is refused
    d2 = np.ascontiguousarray(desc[22:23])
    assert api.lib.cc_db_add_scans_prepare(db_b, C.c_void_p(d2.ctypes.data), 1, None) == -1
    t2, s2 = np.ascontiguousarray(ts[22:23]), np.ascontiguousarray(seeds[22:23])
    assert api.lib.cc_db_add_scans(db_b, C.c_void_p(d2.ctypes.data), 1, C.c_void_p(t2.ctypes.data), C.c_void_p(s2.ctypes.data), None) == -1
    for dd, (a, b) in ((d0, (0, 10)), (d1, (10, 22))):
        t_, s_ = np.ascontiguousarray(ts[a:b]), np.ascontiguousarray(seeds[a:b])
        api.chk(api.lib.cc_db_add_scans(db_b, C.c_void_p(dd.ctypes.data), len(dd), C.c_void_p(t_.ctypes.data), C.c_void_p(s_.ctypes.data), None), "add")
    for a, b in zip(cuts[2:-1], cuts[3:]):
        api.db_add_prepared(db_b, desc[a:b], ts[a:b], seeds[a:b])
    sa, ra = api.bucket_state(db_a)
    sb, rb = api.bucket_state(db_b)
    assert np.array_equal(sa, sb) and np.array_equal(ra, rb)
    hit = np.nonzero(ores["n_res"] > 0)[0]
    qs = np.concatenate([hit[:3], [n - 1]]).astype(np.int32)
    res_a = api.db_query(db_a, desc[qs], qs)
    res_b = api.db_query(db_b, desc[qs], qs)
    assert res_a.tobytes() == res_b.tobytes()
    assert (res_a["n_res"] > 0).sum() >= 1

"""The C-ABI library loads without a GPU, exports every symbol include/cont2_amd.h declares, the numpy layouts
match the compiled structs, and the product refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported(cc):
    so = cc.build()
    lib = C.CDLL(so)
    hdr = open(os.path.join(ROOT, "include", "cont2_amd.h")).read()
    declared = set(re.findall(r"\b(cc_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"cc_ctx", "cc_db"}
    assert declared == set(cc.EXPORTS), declared ^ set(cc.EXPORTS)
    for s in declared:
        assert hasattr(lib, s), s


def test_layout_sizes(cc, oracle):
    L = cc.L
    oracle.lib().orc_sizeof_desc.restype = C.c_size_t
    assert oracle.lib().orc_sizeof_desc() == L.scan_desc_dt.itemsize == cc.DESC_BYTES
    assert L.contour_dt.itemsize == 76 and L.bci_dt.itemsize == 600 and L.relpt_dt.itemsize == 12
    assert C.sizeof(L.ManagerCfg) == 80 and C.sizeof(L.DbCfg) == 64 and C.sizeof(L.Score) == 32
    # the compact per-scan records (hot record + correlation inputs): sizes as the library reports them
    lib = C.CDLL(cc.build())
    hb, fb = C.c_size_t(), C.c_size_t()
    lib.cc_packed_sizes.restype = None
    lib.cc_packed_sizes(C.byref(hb), C.byref(fb))
    assert hb.value == L.hot_desc_dt.itemsize == 18448 and fb.value == 16 + 16 + 8 + 4 * 320 * 32  # CC_MAXC ellipses per correlation level


def test_hot_record_is_a_view_of_the_descriptor(cc, oracle):
    """cc_pack_scans (CPU harness): hot.X[l] == desc.X[l + 1] for keys, BCIs, counts and the first 10 contour rows."""
    import emu_api
    from parity import terrain_scan
    L = cc.L
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=2)
    s = terrain_scan(4, n=8000)
    desc = api.ingest(ctx, s, np.array([0, len(s)], np.int64))
    hot, feat = api.pack(ctx, desc)
    h = hot.view(L.hot_desc_dt).reshape(-1)[0]
    d = desc[0]
    assert np.array_equal(h["n_cont"], d["n_cont"][1:5]) and np.array_equal(h["layer_cell_cnt"], d["layer_cell_cnt"][1:5])
    assert h["keys"].tobytes() == d["keys"][1:5].tobytes() and h["bcis"].tobytes() == d["bcis"][1:5].tobytes()
    for l in range(4):
        ns = min(int(d["n_stored"][l + 1]), 10)
        assert h["cont"][l][:ns].tobytes() == d["cont"][l + 1][:ns].tobytes()
        assert not h["cont"][l][ns:].tobytes().strip(b"\0")


def test_defaults_match_shipped_yaml(cc):
    lib = C.CDLL(cc.build())
    m, d = cc.L.ManagerCfg(), cc.L.DbCfg()
    lb, ub = cc.L.Score(), cc.L.Score()
    lib.cc_default_manager_cfg(C.byref(m))
    lib.cc_default_db_cfg(C.byref(d))
    lib.cc_default_thresholds(C.byref(lb), C.byref(ub))
    pm, pd = cc.L.default_manager_cfg(), cc.L.default_db_cfg()
    plb, pub = cc.L.default_thresholds()
    assert bytes(m) == bytes(pm) and bytes(d) == bytes(pd) and bytes(lb) == bytes(plb) and bytes(ub) == bytes(pub)
    assert list(m.lv_grads) == [1.5, 2.0, 2.5, 3.0, 3.5, 4.0] and (m.n_row, m.n_col) == (150, 150)  # yaml :30-38
    assert (d.nnk, d.max_fine_opt, list(d.q_levels)) == (50, 10, [1, 2, 3]) and (d.max_elapse, d.min_elapse) == (25.0, 15.0)


def test_no_cpu_fallback(cc):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cc.CCError):
        cc.Context(0)
    lib = C.CDLL(cc.build())
    lib.cc_last_error.restype = C.c_char_p
    h = C.c_void_p()
    cfg = cc.L.default_manager_cfg()
    rc = lib.cc_create(0, C.byref(cfg), 4, C.byref(h))
    assert rc != 0 and lib.cc_last_error()


def test_product_does_not_touch_oracle():
    """No file of the product package mentions the oracle or the emulation harness."""
    pkg = os.path.join(ROOT, "contour-context_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_py" not in txt and "libcont2_oracle" not in txt and "libcc_emu" not in txt, f

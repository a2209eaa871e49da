"""numpy/ctypes mirrors of the POD layouts declared in include/cont2_amd.h.

Kept in one place so the host mirror, the tests and the bench all agree with the C header;
`check_sizes()` asserts the numpy itemsizes against the sizes compiled into the shared libraries.
"""
import ctypes as C
import numpy as np

NLEV, KEY_DIM, NPIV, NDIST = 6, 10, 6, 10
BCI_LAYERS, BCI_MAXPTS, MAXC, MAX_CELLS, NQLEV, KNN_MAX = 4, 40, 320, 22500, 3, 64

contour_dt = np.dtype([
    ("level", "<i2"), ("poi", "<i2", (2,)), ("cell_cnt", "<i2"),
    ("pos_mean", "<f4", (2,)), ("pos_cov", "<f4", (4,)), ("eig_vals", "<f4", (2,)),
    ("eig_vecs", "<f4", (4,)), ("eccen", "<f4"), ("vol3_mean", "<f4"), ("com", "<f4", (2,)),
    ("ecc_feat", "u1"), ("com_feat", "u1"), ("pad_", "u1", (2,))], align=True)
relpt_dt = np.dtype([("level", "i1"), ("seq", "i1"), ("bit_pos", "<i2"), ("r", "<f4"), ("theta", "<f4")], align=True)
bci_dt = np.dtype([
    ("dist_bin", "<u8", (BCI_LAYERS,)), ("piv_seq", "i1"), ("level", "i1"), ("n_pts", "u1"), ("n_segs", "u1"),
    ("segs", "<u2", (BCI_MAXPTS + 2,)), ("pts", relpt_dt, (BCI_MAXPTS,))], align=True)
scan_desc_dt = np.dtype([
    ("n_cont", "<i4", (NLEV,)), ("n_stored", "<i4", (NLEV,)), ("layer_cell_cnt", "<i4", (NLEV,)),
    ("max_bin_val", "<f4"), ("min_bin_val", "<f4"), ("n_pix", "<i4"), ("flags", "<i4"),
    ("keys", "<f4", (NLEV, NPIV, KEY_DIM)), ("bcis", bci_dt, (NLEV, NPIV)),
    ("cont", contour_dt, (NLEV, MAXC))], align=True)
HOT_LEVELS, NDIST = 4, 10
# cc_hot_desc_t: what the query path reads of a scan (levels 1..4): hot.X[l] = desc.X[l + 1]
hot_desc_dt = np.dtype([
    ("n_cont", "<i4", (HOT_LEVELS,)), ("layer_cell_cnt", "<i4", (HOT_LEVELS,)), ("flags", "<i4"), ("pad_", "<i4", (3,)),
    ("keys", "<f4", (HOT_LEVELS, NPIV, KEY_DIM)), ("cont", contour_dt, (HOT_LEVELS, NDIST)),
    ("bcis", bci_dt, (HOT_LEVELS, NPIV))], align=True)
assert hot_desc_dt.itemsize == 18448
knn_hit_dt = np.dtype([("gidx", "<i4"), ("level", "<i2"), ("seq", "<i2"), ("dist_sq", "<f4")], align=True)
query_result_dt = np.dtype([
    ("n_res", "<i4"), ("cand_gidx", "<i4"), ("correlation", "<f8"), ("tf", "<f8", (3,)),
    ("cand_aft_check1", "<i4"), ("cand_aft_check2", "<i4"), ("cand_aft_check3", "<i4"),
    ("n_cand_pose", "<i4"), ("n_cand_tidy", "<i4"), ("n_knn_hits", "<i4"), ("flags", "<i4"), ("pad_", "<i4")], align=True)
# the record before `flags` was added (tests/golden/query_fixture.npz stores it)
query_result_v1_dt = np.dtype([
    ("n_res", "<i4"), ("cand_gidx", "<i4"), ("correlation", "<f8"), ("tf", "<f8", (3,)),
    ("cand_aft_check1", "<i4"), ("cand_aft_check2", "<i4"), ("cand_aft_check3", "<i4"),
    ("n_cand_pose", "<i4"), ("n_cand_tidy", "<i4"), ("n_knn_hits", "<i4")], align=True)

assert contour_dt.itemsize == 76 and relpt_dt.itemsize == 12 and bci_dt.itemsize == 600
assert scan_desc_dt.itemsize == 169048, scan_desc_dt.itemsize
assert knn_hit_dt.itemsize == 12 and query_result_dt.itemsize == 72 and query_result_v1_dt.itemsize == 64, query_result_dt.itemsize
# cc_hint_t / cc_hint_score_t (cc_db_check_hints)
hint_dt = np.dtype([("cand_gidx", "<i4"), ("level", "i1"), ("seq_src", "i1"), ("seq_tgt", "i1"), ("pad", "i1")], align=True)
hint_score_dt = np.dtype([("i_ovlp_sum", "<i4"), ("i_ovlp_max_one", "<i4"), ("i_in_ang_rng", "<i4"), ("i_indiv_sim", "<i4"),
                          ("i_orie_sim", "<i4"), ("passed", "<i4")], align=True)
assert hint_dt.itemsize == 8 and hint_score_dt.itemsize == 24
pass_dbg_dt = np.dtype([("hint", "<i4"), ("n_pairs", "<i4"), ("tf", "<f8", (3,)), ("pairs", "<u8", (7,))], align=True)
assert pass_dbg_dt.itemsize == 88


class ManagerCfg(C.Structure):
    _fields_ = [("lv_grads", C.c_float * NLEV), ("reso_row", C.c_float), ("reso_col", C.c_float),
                ("n_row", C.c_int32), ("n_col", C.c_int32), ("lidar_height", C.c_float), ("blind_sq", C.c_float),
                ("min_cont_key_cnt", C.c_int32), ("min_cont_cell_cnt", C.c_int32), ("piv_firsts", C.c_int32),
                ("dist_firsts", C.c_int32), ("roi_radius", C.c_float), ("min_cell_cov", C.c_int32),
                ("point_sigma", C.c_float), ("com_bias_thres", C.c_float)]


class SimCfg(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("ta_cell_cnt", "tp_cell_cnt", "tp_eigval", "ta_h_bar", "ta_rcom", "tp_rcom")]


class Score(C.Structure):
    _fields_ = [("i_ovlp_sum", C.c_int32), ("i_ovlp_max_one", C.c_int32), ("i_in_ang_rng", C.c_int32),
                ("i_indiv_sim", C.c_int32), ("i_orie_sim", C.c_int32), ("correlation", C.c_float),
                ("area_perc", C.c_float), ("neg_est_dist", C.c_float)]


class DbCfg(C.Structure):
    _fields_ = [("nnk", C.c_int32), ("max_fine_opt", C.c_int32), ("n_q_levels", C.c_int32),
                ("q_levels", C.c_int32 * NQLEV), ("cont_sim", SimCfg), ("max_elapse", C.c_double),
                ("min_elapse", C.c_double)]


def default_manager_cfg(mulran=False):
    """Shipped values: config/batch_bin_test_config.yaml:27-47 (+ contour.h:32-37)."""
    c = ManagerCfg()
    grads = [1.0, 2.5, 4.0, 5.5, 7.0, 8.5] if mulran else [1.5, 2.0, 2.5, 3.0, 3.5, 4.0]
    for i, g in enumerate(grads):
        c.lv_grads[i] = g
    c.reso_row = c.reso_col = 1.0
    c.n_row = c.n_col = 150
    c.lidar_height, c.blind_sq = 2.0, 9.0
    c.min_cont_key_cnt, c.min_cont_cell_cnt, c.piv_firsts, c.dist_firsts = 9, 3, 6, 10
    c.roi_radius = 10.0
    c.min_cell_cov, c.point_sigma, c.com_bias_thres = 4, 1.0, 0.5
    return c


def default_db_cfg(mulran=False):
    """config/batch_bin_test_config.yaml:6-23."""
    d = DbCfg()
    d.nnk, d.max_fine_opt, d.n_q_levels = 50, 10, 3
    for i, q in enumerate([1, 2, 3]):
        d.q_levels[i] = q
    s = d.cont_sim
    s.ta_cell_cnt, s.tp_cell_cnt, s.tp_eigval = 6.0, 0.2, 0.2
    s.ta_h_bar = 0.75 if mulran else 0.3
    s.ta_rcom, s.tp_rcom = 0.4, 0.25
    d.max_elapse, d.min_elapse = 25.0, 15.0
    return d


def default_thresholds():
    """config/batch_bin_test_config.yaml:69-87."""
    lb, ub = Score(), Score()
    (lb.i_ovlp_sum, lb.i_ovlp_max_one, lb.i_in_ang_rng, lb.i_indiv_sim, lb.i_orie_sim) = (3, 3, 3, 3, 4)
    lb.correlation, lb.area_perc, lb.neg_est_dist = 0.3, 0.03, -5.01
    (ub.i_ovlp_sum, ub.i_ovlp_max_one, ub.i_in_ang_rng, ub.i_indiv_sim, ub.i_orie_sim) = (6, 6, 6, 6, 6)
    ub.correlation, ub.area_perc, ub.neg_est_dist = 0.75, 0.15, -5.0
    return lb, ub

"""yamlLoader mirror (hostcpp/tools/config_handler.h) and the parameter surface of the drop-in driver
(hostcpp/examples/batch_bin_test.cpp): the keys the reference's batch_bin_test reads (test/batch_bin_test.cpp:38-100)
are read from this repo's config file and -- when the reference tree is present -- from the reference's own shipped
config/batch_bin_test_config.yaml with identical results, and they equal the C-ABI defaults."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "contour-context_amd")
REF_CFG = "/root/reference/config/batch_bin_test_config.yaml"


def _run(cc, tmp_path, cfg):
    cc.build()
    exe = str(tmp_path / "batch_bin_test")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", os.path.join(PKG, "hostcpp", "examples", "batch_bin_test.cpp"),
                               "-I", os.path.join(PKG, "hostcpp"), "-I", os.path.join(ROOT, "include"), "-L", PKG, "-lcont2_amd",
                               "-Wl,-rpath," + PKG, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe, cfg], capture_output=True, text=True).stdout
    kv = {}
    for line in out.split("\n"):
        m = re.match(r'^((?:"[^"]+"->)+): (.*)$', line)
        if m and "[!]" not in m.group(2):
            kv["/".join(re.findall(r'"([^"]+)"', m.group(1)))] = m.group(2).strip()
    return kv


def test_driver_reads_the_reference_key_surface(cc, tmp_path):
    mine = _run(cc, tmp_path, os.path.join(PKG, "hostcpp", "examples", "batch_bin_test_config.yaml"))
    L = cc.L
    m, d = L.default_manager_cfg(), L.default_db_cfg()
    lb, ub = L.default_thresholds()
    assert [float(v) for v in mine["ContourManagerConfig/lv_grads_"].rstrip(", ").split(",")] == [float(x) for x in m.lv_grads]
    for k in ("n_row", "n_col", "lidar_height", "blind_sq", "min_cont_key_cnt", "min_cont_cell_cnt", "piv_firsts", "dist_firsts", "roi_radius"):
        assert float(mine["ContourManagerConfig/%s_" % k]) == float(getattr(m, k)), k
    assert int(mine["ContourDBConfig/nnk_"]) == d.nnk and int(mine["ContourDBConfig/max_fine_opt_"]) == d.max_fine_opt
    assert [int(v) for v in mine["ContourDBConfig/q_levels_"].rstrip(", ").split(",")] == list(d.q_levels)[:d.n_q_levels]
    assert float(mine["ContourDBConfig/TreeBucketConfig/max_elapse_"]) == d.max_elapse
    assert float(mine["ContourDBConfig/TreeBucketConfig/min_elapse_"]) == d.min_elapse
    for k in ("ta_cell_cnt", "tp_cell_cnt", "tp_eigval", "ta_h_bar", "ta_rcom", "tp_rcom"):
        assert abs(float(mine["ContourDBConfig/ContourSimThresConfig/" + k]) - getattr(d.cont_sim, k)) < 1e-6, k
    for name, s in (("thres_lb_", lb), ("thres_ub_", ub)):
        for k in ("i_ovlp_sum", "i_ovlp_max_one", "i_in_ang_rng", "i_indiv_sim", "i_orie_sim", "correlation", "area_perc", "neg_est_dist"):
            assert abs(float(mine[name + "/" + k]) - getattr(s, k)) < 1e-6, (name, k)
    assert abs(float(mine["correlation_thres"]) - 0.64928) < 1e-12
    if os.path.exists(REF_CFG):  # the reference's own file goes through the same reader
        ref = _run(cc, tmp_path, REF_CFG)
        paths = {"fpath_sens_gt_pose", "fpath_lidar_bins", "fpath_outcome_sav"}
        assert set(ref) == set(mine)
        for k in ref:
            if k not in paths:
                assert ref[k] == mine[k], k
        assert ref["fpath_sens_gt_pose"].endswith("ts-sens_pose-kitti08.txt")

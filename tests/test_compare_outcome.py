"""tools/compare_outcome.py (the KITTI-08 "pin on arrival" diff) on the outcome file the reference ships: identical to
itself, and every kind of deviation it is meant to catch is caught."""
import gzip
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "contour-context_amd", "tools"))
GOLD = os.path.join(ROOT, "tests", "golden", "outcome-kitti08.txt.gz")
POSES = os.path.join(ROOT, "tests", "golden", "ts-sens_pose-kitti08.txt.gz")


def _rows():
    with gzip.open(GOLD, "rt") as f:
        return [l.rstrip("\n").split("\t") for l in f]


def _write(path, rows):
    with open(path, "w") as f:
        f.write("\n".join("\t".join(r) for r in rows) + "\n")


def test_identical_and_pinned_numbers():
    import compare_outcome as co
    r = co.compare(GOLD, GOLD, poses=POSES)
    assert r["ok"] and r["n_rows_ours"] == 4071
    assert abs(r["max_f1_ref"] - 0.955621) < 5e-7 and r["max_f1_idx_ref"] == 1802 and r["tp_count_ref"] == 323


def test_deviations_are_caught(tmp_path):
    import compare_outcome as co
    rows = _rows()
    assert rows[1648][1] == "1648-237" and rows[1648][2] == "0.789965"   # the pair the reference's scripts single out
    # (a) a different matched scan
    a = [list(r) for r in rows]
    a[1648][1] = "1648-238"
    _write(tmp_path / "a.txt", a)
    r = co.compare(str(tmp_path / "a.txt"), GOLD)
    assert not r["ok"] and r["n_match_diff"] == 1
    # (b) correlation off in the 5th significant digit; (c) pose error off by 2e-4; (d) a flipped label
    b = [list(r) for r in rows]
    b[1648][2] = "0.789985"
    b[1649][4] = "%g" % (float(b[1649][4]) + 2e-4)
    b[10][0] = "1" if b[10][0] != "1" else "2"
    _write(tmp_path / "b.txt", b)
    r = co.compare(str(tmp_path / "b.txt"), GOLD)
    assert not r["ok"] and (r["n_corr_diff"], r["n_pose_diff"], r["n_label_diff"]) == (1, 1, 1)
    # (e) last-digit noise of the printed values passes
    c = [list(r) for r in rows]
    c[1648][2] = "0.789966"
    _write(tmp_path / "c.txt", c)
    assert co.compare(str(tmp_path / "c.txt"), GOLD)["ok"]
    # (f) a missing row
    _write(tmp_path / "d.txt", rows[:-1])
    assert not co.compare(str(tmp_path / "d.txt"), GOLD)["ok"]

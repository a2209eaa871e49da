"""experiments/*.patch are kernel changes waiting for GPU time (experiments/README.md): they must keep applying to the tree."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_patches_apply_cleanly():
    patches = sorted(glob.glob(os.path.join(ROOT, "experiments", "*.patch")))
    assert patches
    for p in patches:
        r = subprocess.run(["patch", "-p1", "--dry-run", "-s", "-f", "-i", p], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, (os.path.basename(p), r.stdout[-500:], r.stderr[-500:])

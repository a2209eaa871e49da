"""Tuning aid: the drop-in per-scan loop (bench.py:dropin_loop = hostcpp/examples/batch_bin_test on 1 024 .bin files) under
different settings of the mirror's read-ahead.   python profiles/r5/dropin_probe.py "0,1,3" ["4,8,16"|default] [laps]   (read-ahead depths, evaluator look-ahead, laps over the 1 024 files) """
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["CC_EVAL_TIMERS"] = "1"
import torch  # noqa: E402
import bench  # noqa: E402
import cc_amd  # noqa: E402

cc = cc_amd.load()
n = 1024
x, _, _ = cc.synth.make_sequence(n, world=cc.synth.World(), device=torch.device("cuda", 0), start=5000, beams=64, azim=1875)
P = x.shape[1]
b0 = x.reshape(-1, 4).contiguous()
for ahead in (sys.argv[2] if len(sys.argv) > 2 else "default").split(","):
    if ahead != "default":
        os.environ["CC_EVAL_AHEAD"] = ahead
    for ra in (sys.argv[1] if len(sys.argv) > 1 else "0,3").split(","):
        if ra != "default":
            os.environ["CC_DB_READ_AHEAD"] = ra
        else:
            os.environ.pop("CC_DB_READ_AHEAD", None)
        d = bench.dropin_loop(b0, P, n, laps=int(sys.argv[3]) if len(sys.argv) > 3 else 1)
        d.pop("what", None)
        print("CC_EVAL_AHEAD=%s CC_DB_READ_AHEAD=%s %s" % (ahead, ra, json.dumps(d)), flush=True)

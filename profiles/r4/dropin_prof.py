"""Tuning aid (GPU box): the drop-in per-scan loop (hostcpp/examples/batch_bin_test) under rocprofv3 --kernel-trace:
per-kernel launch counts and durations of the one-scan chains next to the driver's own stage timers."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cc_amd
import bench
cc = cc_amd.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
world = cc.synth.World(kitti=True) if (len(sys.argv) > 2 and sys.argv[2] == "kitti") else cc.synth.World()
P = 64 * 1875
x, _, _ = cc.synth.make_sequence(n, world=world, device="cuda", start=5000)
os.environ["CC_DROPIN_PREFIX"] = "rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pdrop -o r --"
os.system("rm -rf /tmp/pdrop")
out = bench.dropin_loop(x.reshape(-1, 4), P, n)
print(json.dumps({k: v for k, v in out.items() if k != "what"}))
for f in glob.glob("/tmp/pdrop/**/*kernel_stats.csv", recursive=True):
    tot = 0.0
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        tot += float(r["TotalDurationNs"])
    for r in rows:
        print("%-46s calls %6s (%.2f per scan) avg %8.1f us  total/scan %7.1f us" % (r["Name"].replace("void ", "")[:46], r["Calls"], int(r["Calls"]) / n,
                                                                                    float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / n))
    print("sum of kernel time per scan: %.1f us" % (tot / 1e3 / n))

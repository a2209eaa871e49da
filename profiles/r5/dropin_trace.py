"""Tuning aid: kernel timeline of the drop-in per-scan loop (bench.py:dropin_loop under rocprofv3 --kernel-trace).
   python profiles/r5/dropin_trace.py <read-ahead depth> <evaluator look-ahead> [scans]
Prints per kernel: launches, mean duration; the device's busy fraction over the loop; per queue busy time; the gaps."""
import csv
import glob
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
ra = sys.argv[1] if len(sys.argv) > 1 else "default"
ahead = sys.argv[2] if len(sys.argv) > 2 else "default"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
x, _, _ = cc.synth.make_sequence(n, world=cc.synth.World(), device=torch.device("cuda", 0), start=5000, beams=64, azim=1875)
P = x.shape[1]
b0 = x.reshape(-1, 4).contiguous()
out = os.path.join(ROOT, "gpurun_out", "dtrace_%s_%s" % (ra, ahead))
if ra != "default":
    os.environ["CC_DB_READ_AHEAD"] = ra
if ahead != "default":
    os.environ["CC_EVAL_AHEAD"] = ahead
os.environ["CC_DROPIN_PREFIX"] = "rocprofv3 --kernel-trace --output-format csv -d %s --" % out
d = bench.dropin_loop(b0, P, n, reps=1)
d.pop("what", None)
print(json.dumps(d))
rows = []
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
print("kernel records:", len(rows))
if not rows:
    sys.exit(0)
for r in rows:
    r["s"] = int(r["Start_Timestamp"])
    r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# skip the warm part: the last 75 % of the records
rows = rows[len(rows) // 4:]
t0, t1 = rows[0]["s"], max(r["e"] for r in rows)
span = (t1 - t0) * 1e-3
byname = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    a = byname.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (r["e"] - r["s"]) * 1e-3
print("span %.1f us, %d kernels" % (span, len(rows)))
for k, a in sorted(byname.items(), key=lambda kv: -kv[1][1]):
    print("  %-60s n %6d  mean %8.1f us  total %10.1f us (%.1f %% of span)" % (k, a[0], a[1] / a[0], a[1], 100 * a[1] / span))
# union busy
ev = sorted((r["s"], r["e"]) for r in rows)
busy, cs, ce = 0, ev[0][0], ev[0][1]
for s, e in ev[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("device busy (union of kernel intervals): %.1f %% of the span" % (100.0 * busy * 1e-3 / span))
qk = "Queue_Id" if "Queue_Id" in rows[0] else None
sk = "Stream_Id" if "Stream_Id" in rows[0] else None
for key in (qk, sk):
    if not key:
        continue
    per = {}
    for r in rows:
        a = per.setdefault(r[key], [0, 0.0])
        a[0] += 1
        a[1] += (r["e"] - r["s"]) * 1e-3
    print(key, {k: "%d kernels, busy %.1f %%" % (v[0], 100 * v[1] / span) for k, v in per.items()})
# one chain: from a cc_k_knn start to the next cc_k_final end on the same queue
key = sk or qk
chains = []
open_ = {}
for r in rows:
    nm = r["Kernel_Name"]
    q = r[key] if key else 0
    if "cc_k_knn" in nm and q not in open_:
        open_[q] = r["s"]
    if "cc_k_final" in nm and q in open_:
        chains.append((r["e"] - open_.pop(q)) * 1e-3)
if chains:
    chains.sort()
    print("query chains: %d, device latency first kernel start -> last kernel end: median %.1f us, p10 %.1f, p90 %.1f" %
          (len(chains), chains[len(chains) // 2], chains[len(chains) // 10], chains[9 * len(chains) // 10]))

"""Scratch GPU timing script (not a test): ingest throughput + per-kernel timing."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cc_amd
cc = cc_amd.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = cc.synth.World()
t = time.time()
xyzi, poses, ts = cc.synth.make_sequence(64, world=w, device="cuda")
torch.cuda.synchronize(); print("synth 64 scans on GPU: %.2fs" % (time.time() - t), flush=True)
rep = (n + 63) // 64
x = xyzi.repeat(rep, 1, 1)[:n].reshape(-1, 4).contiguous()
P = xyzi.shape[1]
offs = np.arange(n + 1, dtype=np.int64) * P
ctx = cc.Context(0, max_batch=n)
out = ctx.ingest(x, offs); torch.cuda.synchronize()
for it in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); out = ctx.ingest(x, offs, out=out); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("ingest %d scans: %.3f ms -> %.0f scans/s, %.1f GB/s algorithmic" % (n, ms, n / ms * 1e3, n * P * 16 / ms / 1e6), flush=True)
d = cc.desc_to_numpy(out)
print("n_cont mean", d["n_cont"].mean(0), "flags", np.unique(d["flags"]))

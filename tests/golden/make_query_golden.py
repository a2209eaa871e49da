"""Generates tests/golden/query_fixture.npz: the descriptors of a 64-scan looping sequence (inputs of the query path),
their time stamps, and the CPU oracle's replay of the reference driver loop on them (expected outputs: per scan the
matched scan, correlation, pose and the integer gate counters).  DB delays shortened (1.5 / 2.5 s) so that revisits are
searchable inside 64 scans.  Re-run only when the oracle changes on purpose."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import cc_amd  # noqa: E402
import oracle_py as O  # noqa: E402

cc = cc_amd.load()
L = O.L
d = L.default_db_cfg()
d.max_elapse, d.min_elapse = 2.5, 1.5
w = cc.synth.World(loop_len=40.0)
n = 64
x, poses, ts = cc.synth.make_sequence(n, world=w, beams=16, azim=450)
xs = x.numpy().reshape(-1, 4)
offs = np.arange(n + 1, dtype=np.int64) * x.shape[1]
res, _, desc = O.run_sequence(xs, offs, ts, np.arange(n, dtype=np.int32), dcfg=d, want_desc=True)
# the unused tail of the contour table (beyond n_stored) is zero in the oracle's export: the file compresses well
np.savez_compressed(os.path.join(HERE, "query_fixture.npz"), desc=np.frombuffer(desc.tobytes(), np.uint8), ts=ts,
                    res=np.frombuffer(res.tobytes(), np.uint8), elapse=np.array([d.min_elapse, d.max_elapse]))
print("loop closures:", int((res["n_res"] > 0).sum()), "of", n)

"""Generates tests/golden/ingest_fixture.npz: two small seeded scans (inputs) and the CPU oracle's descriptors for
them (expected outputs).  Re-run only when the oracle changes on purpose."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import cc_amd  # noqa: E402
import oracle_py as O  # noqa: E402
from parity import terrain_scan  # noqa: E402

cc = cc_amd.load()
w = cc.synth.World(loop_len=200.0)
x, _, _ = cc.synth.make_sequence(1, world=w, beams=16, azim=600, start=5)
scan0 = x[0].numpy()
scan1 = terrain_scan(42, n=5000, scale=1.6, quant=0.2)
desc = np.concatenate([O.Scan(scan0).desc(), O.Scan(scan1).desc()])
np.savez_compressed(os.path.join(HERE, "ingest_fixture.npz"), scan0=scan0, scan1=scan1,
                    desc=np.frombuffer(desc.tobytes(), np.uint8))
print("n_cont", desc["n_cont"])

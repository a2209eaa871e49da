"""Makes tests/golden/knn_unindexed_bucket_fixture.npz from a dump of the randomised GPU campaign
(CC_FUZZ_DUMP=<dir> python tests/fuzz_gpu_query.py 12046 1 on a GPU box; the dump holds the oracle's descriptors of the drive):
the retrieval keys, time stamps, balance seeds and DB configuration of the drive, and the ORACLE's hit count of every search of
every scan (scan i queried before it is added, as the reference's loop does).  usage: python make_knn_unindexed_fixture.py <dump.npz>"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", ".."), os.path.join(HERE, "..", "..", "oracle")]
import oracle_py as oracle  # noqa: E402

L = oracle.L
z = np.load(sys.argv[1])
odesc = np.frombuffer(z["odesc"].tobytes(), dtype=L.scan_desc_dt)
cfg, ts, seeds = z["cfg"], z["ts"], z["seeds"]
d = L.default_db_cfg()
d.min_elapse, d.max_elapse, d.nnk, d.max_fine_opt, d.n_q_levels = cfg[0], cfg[1], int(cfg[2]), int(cfg[3]), int(cfg[4])
for i in range(3):
    d.q_levels[i] = int(cfg[5 + i])
odb = oracle.DB(d)
cnt = []
for i in range(len(odesc)):
    s = oracle.Scan.from_desc(odesc[i], int_id=i)
    cnt.append(odb.query(s, want_knn=True)[2])
    odb.add_scan(s, ts[i])
    odb.push_and_balance(int(seeds[i]), ts[i])
np.savez_compressed(os.path.join(HERE, "knn_unindexed_bucket_fixture.npz"), keys=odesc["keys"].astype(np.float32), ts=ts, seeds=seeds, cfg=cfg,
                    knn_cnt=np.stack(cnt).astype(np.int32))
print("scans", len(odesc), "searches with hits", int((np.stack(cnt) > 0).sum()))

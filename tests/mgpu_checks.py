"""One set of assertions on bench.py's `multi_gpu` object, shared by the CPU twin (tests/test_distributed_gloo.py, CPU
harness + gloo) and the hardware test (tests/test_gpu_multigpu.py, nccl = RCCL): record sizes are taken from the library
(`cc_packed_sizes`), never written down, so the two cannot drift apart (round 4's hardware test still carried round 3's
16 424-byte correlation record)."""


def check_multi_gpu_line(d, world, db_scans, batch, packed_sizes, share, backend, shard_len):
    """packed_sizes = cc_packed_sizes() of the library underneath; shard_len = contour-context_amd/sharding.py:shard_len"""
    hb, fb = packed_sizes
    rec = hb + fb
    m = d["multi_gpu"]
    assert d["n_gpus"] == world and m["ranks_seen"] == world and m["backend"] == backend
    assert len(m["per_rank_scans_per_s"]) == world and min(m["per_rank_scans_per_s"]) > 0
    ex = m["db_exchange"]
    assert ex["bytes_per_scan"] == rec, (ex["bytes_per_scan"], hb, fb)
    assert ex["bytes_gathered_per_rank"] == world * shard_len(db_scans, world) * rec and ex["ms"] > 0
    if share:
        ps = m["per_step_exchange"]
        assert ps["bytes_gathered_per_rank"] == world * batch * rec and ps["ms"] > 0
        assert m["data_path_collectives_in_timed_step"] == 1
    else:
        assert m["per_step_exchange"] is None and m["data_path_collectives_in_timed_step"] == 0

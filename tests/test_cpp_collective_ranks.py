"""The C++-owned collective of the path with MORE THAN ONE RANK, on the CPU (include/cont2_amd.h: cc_comm_*):
hostcpp/examples/batch_replay_mgpu.cpp built against the CPU harness build of the C-ABI (tests/emu), RCCL replaced by a
test-only stand-in over shared memory (tests/emu/nccl_standin.cpp, loaded through CC_RCCL_LIB exactly as a real librccl is).
What runs here for the first time with N > 1: the program's own forker, the unique-id file of cc_comm_create_from_env under
/dev/shm, the rank order of cc_comm_allgather_packed, the padding rows of a last shard that is shorter than the others, the
re-ordering to scan order, the replicated database and the query sharding.  What it says nothing about: RCCL itself, xGMI,
speed -- the multi-GPU path stays UNMEASURED ON HARDWARE (the build boxes have one GPU)."""
import os
import subprocess
import sys

import numpy as np

import emu_api

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU = os.path.join(HERE, "emu")


def _build(tmp_path):
    emu_so = emu_api.build()
    standin = str(tmp_path / "libnccl_standin.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", os.path.join(EMU, "nccl_standin.cpp"), "-lrt", "-o", standin])
    exe = str(tmp_path / "batch_replay_mgpu_emu")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "contour-context_amd", "hostcpp", "examples", "batch_replay_mgpu.cpp"),
                           "-I", EMU, "-I", os.path.join(ROOT, "include"), "-L", os.path.dirname(emu_so), "-lcc_emu",
                           "-Wl,-rpath," + os.path.dirname(emu_so), "-pthread", "-o", exe])
    return exe, standin


def _scans(tmp_path, n):
    sys.path.insert(0, ROOT)
    import cc_amd
    cc = cc_amd.load()
    x, _, ts = cc.synth.make_sequence(n, world=cc.synth.World(loop_len=40.0), beams=16, azim=450)
    xs = x.cpu().numpy()
    ts = np.asarray(ts, np.float64) * 10.0  # the shipped DB delays (15 / 25 s) against a 40-m loop driven in a few seconds
    lst = tmp_path / "scans.txt"
    with open(lst, "w") as f:
        for i in range(n):
            p = tmp_path / ("%06d.bin" % i)
            xs[i].astype(np.float32).tofile(p)
            f.write("%.6f %s\n" % (ts[i], p))
    return lst


def _env(standin, **kw):
    env = dict(os.environ, CC_RCCL_LIB=standin, CC_EMU_DEVICES="8", **emu_api.SMALL_GRIDS)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "CC_COMM_TOKEN", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.update(kw)
    return env


def _outcome(prefix, world, n):
    rows = np.concatenate([np.loadtxt("%s.rank%d.txt" % (prefix, r)).reshape(-1, 6) for r in range(world)], 0)
    rows = rows[np.argsort(rows[:, 0], kind="stable")]
    assert np.array_equal(rows[:, 0].astype(int), np.arange(n)), "every scan is queried by exactly one rank"
    return rows


def _leftovers():
    return [f for f in os.listdir("/dev/shm") if f.startswith("cc_nccl_standin_") or f.startswith("cc_comm_id_")]


def test_forked_ranks_give_the_single_process_outcome(tmp_path):
    """--gpus 1 / 2 / 4 of the program's own forker on a list that 4 does not divide (46 scans: shards of 23, and of 12 with
    two padding rows in the last two ranks' shards): identical outcome rows, and the loops that the drive closes are found."""
    exe, standin = _build(tmp_path)
    n = 46
    lst = _scans(tmp_path, n)
    before = set(_leftovers())
    outs = {}
    for world in (1, 2, 4):
        r = subprocess.run([exe, str(lst), str(tmp_path / ("out%d" % world)), "--gpus", str(world)], env=_env(standin), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        for k in range(world):
            assert "rank %d of %d" % (k, world) in r.stdout, r.stdout[-1500:]
        outs[world] = _outcome(str(tmp_path / ("out%d" % world)), world, n)
    assert (outs[1][:, 1] >= 0).sum() > 0, "the sequence should close loops"
    for world in (2, 4):
        assert np.array_equal(outs[world][:, 1], outs[1][:, 1]), "matched scans differ between %d ranks and one" % world
        assert np.abs(outs[world][:, 2:] - outs[1][:, 2:]).max() < 1e-9   # same replica, same queries: same numbers
    assert not (set(_leftovers()) - before), "segments / id files left behind"


def test_ranks_under_an_external_launcher_and_a_stale_id_file(tmp_path):
    """The ranks started one by one the way torch.distributed.run starts them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT /
    a run id), rank 0 LAST and a stale id file of an earlier job with the same name in place: the other ranks must wait for the
    real id (the stale file is older than the two minutes a reader accepts), and the outcome is the single process's."""
    exe, standin = _build(tmp_path)
    n = 21
    lst = _scans(tmp_path, n)
    world, port, tok = 3, "29731", "cpu-test-%d" % os.getpid()
    stale = "/dev/shm/cc_comm_id_%s_%d_%s" % (port, world, tok)
    with open(stale, "wb") as f:
        f.write(b"\0" * 128)
    os.utime(stale, (1, 1))  # 1970: a crashed job's leftover
    procs = []
    try:
        for rank in (2, 1, 0):
            env = _env(standin, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_PORT=port, CC_COMM_TOKEN=tok,
                       MASTER_ADDR="127.0.0.1")
            procs.append(subprocess.Popen([exe, str(lst), str(tmp_path / "ext")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=900)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        if os.path.exists(stale):
            os.unlink(stale)
    ext = _outcome(str(tmp_path / "ext"), world, n)
    r = subprocess.run([exe, str(lst), str(tmp_path / "one")], env=_env(standin), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    one = _outcome(str(tmp_path / "one"), 1, n)
    assert np.array_equal(ext[:, 1], one[:, 1]) and np.abs(ext[:, 2:] - one[:, 2:]).max() < 1e-9

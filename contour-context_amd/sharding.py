"""Scan sharding of the multi-GPU path.  One process per GPU; the path has ONE exchange:

  * ingest is scan-sharded, interleaved: of a block of scans rank r takes r, r + G, r + 2 G, ... (consecutive scans take
    equally long, so every rank finishes together whatever the block length);
  * each rank packs its scans into the two compact records the database keeps (cc_pack_scans: 18 KB hot record + 41 KB
    correlation inputs, against 169 KB of descriptor) and the ranks all-gather them -- `torch.distributed` with the "nccl"
    backend, i.e. RCCL over xGMI (point to point, fully connected: every link carries one peer's shard); no all-reduce;
  * every rank re-orders the gathered records into scan order and appends them to its replica (cc_db_add_packed): the host
    bookkeeping is deterministic, so the replicas are identical;
  * queries are sharded the same way and never need another rank's data again.

bench.py (DB build, `--share-descriptors`) and tests/test_distributed_gloo.py (the same code on CPU over gloo) both go
through these helpers, so there is one layout, and it is the tested one.  PyTorch here is plumbing only.
"""
import numpy as np


def my_scans(n, rank, world, first=0):
    """global indices of the scans rank `rank` takes out of the block first .. first + n - 1"""
    return first + np.arange(rank, n, world)


def shard_len(n, world):
    """records per rank in the exchange (the last ranks' shards are padded when world does not divide n)"""
    return (n + world - 1) // world


def scan_order(n, world):
    """index array `o` with gathered[o] in scan order, `gathered` being the rank-major result of the all-gather of
    shard_len(n, world)-long shards (padding rows are skipped)"""
    s = shard_len(n, world)
    pos = np.empty(n, np.int64)
    for r in range(world):
        idx = np.arange(r, n, world)
        pos[idx] = r * s + np.arange(len(idx))
    return pos


def gather_records(rec_local, n, world, dist, out=None):
    """rec_local: uint8 tensor [shard_len(n, world), B] holding this rank's records in its first len(my_scans) rows.
    Returns (records [n, B] in scan order, bytes moved by the collective).  world == 1: no collective."""
    import torch
    s = shard_len(n, world)
    assert rec_local.shape[0] == s and rec_local.dtype == torch.uint8
    if world == 1:
        return rec_local[:n], 0
    if out is None:
        out = torch.empty((world * s, rec_local.shape[1]), dtype=torch.uint8, device=rec_local.device)
    dist.all_gather_into_tensor(out, rec_local.contiguous())
    order = torch.from_numpy(scan_order(n, world)).to(out.device)
    return out.index_select(0, order), int(out.numel())


def gather_records_c(cc, comm, rec_local, n, world, stream=None):
    """gather_records with the collective owned by the C library (cc_comm_allgather_packed = ncclAllGather over RCCL, called
    through ctypes; `comm` = a cc_comm handle, e.g. from cc.comm_from_env()): what a C++ host does
    (hostcpp/examples/batch_replay_mgpu.cpp).  The collective runs at world == 1 as well (RCCL with a world of one)."""
    import torch
    s = shard_len(n, world)
    assert rec_local.shape[0] == s and rec_local.dtype == torch.uint8 and rec_local.is_cuda
    rec_local = rec_local.contiguous()
    out = torch.empty((world * s, rec_local.shape[1]), dtype=torch.uint8, device=rec_local.device)
    st = torch.cuda.current_stream(rec_local.device).cuda_stream if stream is None else stream
    cc.comm_allgather(comm, rec_local.data_ptr(), out.data_ptr(), int(rec_local.numel()), st)
    order = torch.from_numpy(scan_order(n, world)).to(out.device)
    return out.index_select(0, order), int(out.numel())

"""world_size-2 test of the N>1 path on CPU (gloo): row sharding + final all-gather."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, out_dir):
    sys.path.insert(0, ROOT)
    import afiv_bootstrap
    afiv = afiv_bootstrap.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.arange(n_rows * 3, dtype=torch.float64).reshape(n_rows, 3)
    mine = afiv.dist.shard_rows(full)                      # this rank's block of parameter rows
    lo, hi = afiv.dist.shard_bounds(n_rows, world, rank)
    assert mine.shape[0] == hi - lo
    local_result = mine * 2.0 + 1.0                        # stands in for the per-surface outputs
    flags = torch.full((hi - lo, 4), rank, dtype=torch.int32)
    gathered = afiv.dist.gather_rows(local_result, n_rows)
    gflags = afiv.dist.gather_rows(flags, n_rows)
    assert torch.equal(gathered, full * 2.0 + 1.0)
    expect = torch.cat([torch.full((afiv.dist.shard_bounds(n_rows, world, r)[1]
                                    - afiv.dist.shard_bounds(n_rows, world, r)[0], 4), r,
                                   dtype=torch.int32) for r in range(world)])
    assert torch.equal(gflags, expect)
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    for n_rows in (11, 2, 1):          # uneven shards, and a rank with zero rows
        port = _free_port()
        mp.spawn(_worker, args=(2, port, n_rows, str(tmp_path)), nprocs=2, join=True)
        assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
        (tmp_path / "ok0").unlink()
        (tmp_path / "ok1").unlink()

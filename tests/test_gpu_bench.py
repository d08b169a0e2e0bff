"""bench.py as a program: it starts its own ranks, and two ranks (sharing the one GPU of the test
box, gathers through gloo) produce the flag words of a single-rank run, surface for surface."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _bench(tmp_path, tag, *extra):
    dump = str(tmp_path / f"{tag}.npy")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--surfaces", "6144", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--dump", dump, *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    return line, np.load(dump)


def test_two_self_launched_ranks_reproduce_the_single_rank_flags(tmp_path):
    one, flags1 = _bench(tmp_path, "one", "--gpus", "1")
    two, flags2 = _bench(tmp_path, "two", "--gpus", "2", "--backend", "gloo")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["config"]["surfaces_per_gpu_per_step"] == 6144 and two["config"]["surfaces_per_gpu_per_step"] == 3072
    assert flags1.shape == flags2.shape == (6144, 4)
    assert np.array_equal(flags1, flags2)
    assert (flags1[:, 3] & 4).any() and not (flags1[:, 3] & 4).all()
    for line in (one, two):
        assert line["roofline"]["kernel"] in ("price_kernel", "iv_fast_kernel")
        for r in (line["roofline"], line["roofline_other_kernel"]):
            assert 0 < r["frac"] <= 1.0                          # a fraction of a peak, whatever the kernel
            assert r["bound"] == ("mfma" if r["kernel"] == "price_kernel" else "hbm")
            assert r["traffic"] is None or r["traffic_source"].startswith("profiles/pmc_traffic.json@")
        assert "REDUCED" in line["config"]["workload"]          # not the configured batch: says so
    assert one["general_lattice"]["value"] > 0 and "general_lattice" not in two
    assert two["xgmi_bound"]["bytes_per_link_per_step"] == 3072 * (20 * 50 * 8 + 16)


def test_launch_under_torch_distributed_run(tmp_path):
    """The other way in: one rank per process started by torch.distributed.run, rank and world
    size taken from the environment (two ranks on the one GPU, gloo)."""
    dump = str(tmp_path / "tdr.npy")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--backend", "gloo", "--surfaces", "2048", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--dump", dump]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["surfaces_per_gpu_per_step"] == 1024
    assert "final_gather" in line and line["kernels_only"]["value"] > 0
    assert np.load(dump).shape == (2048, 4)


def test_rccl_exchange_path_with_one_rank(tmp_path):
    """The collective code itself -- asynchronous RCCL all-gathers behind every chunk, local outputs
    double-buffered across steps, the fence, the stand-alone exchange timing -- run as a one-rank
    RCCL group on the test box's GPU (two ranks cannot share a device under RCCL): three steps, so
    both buffer pairs are reused, must leave the flag words of the plain run."""
    one, flags1 = _bench(tmp_path, "plain")
    cmd_extra = ("--solo-gather",)
    dump = str(tmp_path / "solo.npy")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--surfaces", "6144", "--steps", "3", "--warmup", "2",
           "--no-cpu-baseline", "--dump", dump, *cmd_extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "all-gathered" in line["config"]["parallelism"] and line["final_gather"]["ms"] > 0
    assert line["final_gather"]["bytes_per_rank"] == 6144 * (20 * 50 * 8 + 16)
    # five steps (two warm-up, three timed) alternate between the two pairs of local buffers
    assert sorted(line["final_gather"]["steps_on_each_buffer_pair"]) == [2, 3]
    assert np.array_equal(np.load(dump), flags1)
    # the same with the flag words alone: vols stay sharded, 16 bytes per surface cross
    dump2 = str(tmp_path / "solo_flags.npy")
    out = subprocess.run(cmd[:-3] + ["--dump", dump2, "--solo-gather", "--gather", "flags"], capture_output=True,
                         text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["final_gather"]["bytes_per_rank"] == 6144 * 16 and "flag words only" in line["config"]["parallelism"]
    assert np.array_equal(np.load(dump2), flags1)

"""The N>1 path on CPU: world_size-2 gloo processes shard independent requests, receive the prompt
embeddings by broadcast and gather RGB8 results (the RCCL path on the MI355X node runs the same code)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_generate(embeds, seeds):
    # deterministic stand-in for the sampler: image depends on (embedding, seed) only
    out = torch.zeros(len(seeds), 4, 6, 3, dtype=torch.uint8)
    for i, s in enumerate(seeds):
        v = int(embeds[i].float().sum().item() * 7 + s) % 251
        out[i] = v
    return out


def _worker(rank, world, port, n, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import sdlcm_amd  # noqa
    from sdlcm_amd.distributed import run_sharded, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = [1000 + i for i in range(n)]
    embeds = None
    if rank == 0:
        embeds = torch.randn(n, 77, 16, generator=torch.Generator().manual_seed(3)).half()
    full = run_sharded(_fake_generate, embeds, seeds, torch.device("cpu"), gather_to=0)
    lo, hi = shard_bounds(n, world, rank)
    q.put((rank, lo, hi, None if full is None else full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 2, 1])
def test_sharded_generation_two_ranks(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, full), (r1, lo1, hi1, none) = res
    assert (lo0, hi1) == (0, n) and hi0 == lo1 and none is None
    embeds = torch.randn(n, 77, 16, generator=torch.Generator().manual_seed(3)).half()
    ref = _fake_generate(embeds, [1000 + i for i in range(n)]).numpy()
    assert full.shape == ref.shape and (full == ref).all()


def test_shard_bounds_cover_everything():
    from sdlcm_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_gpus_flag_is_honoured():
    """`python bench.py --gpus N` by hand must launch N ranks itself (fresh children via torch.distributed.run, the parent
    never initialises the GPU); inside a torchrun environment it is a rank, and a WORLD_SIZE that disagrees with --gpus is an
    error rather than a silently smaller run."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cmd = bench.spawn_command(8, ["--gpus", "8", "--steps", "5"], {})
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "5"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert bench.spawn_command(1, [], {}) is None
    assert bench.spawn_command(2, ["--gpus", "2"], {"WORLD_SIZE": "2", "RANK": "0"}) is None
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.spawn_command(8, ["--gpus", "8"], {"WORLD_SIZE": "2"})

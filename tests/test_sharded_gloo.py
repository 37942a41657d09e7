"""Multi-process (world_size 2, gloo, CPU) test of the grid-point sharding + all-gather path.
The per-shard computation is injected (CPU oracle) because there is no GPU here; everything else
-- partitioning, padding of the tail block, the single all_gather_into_tensor, reassembly -- is the
product code of torch-assimilate_amd/sharded.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import letkf_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_shard(X, grid_x, obs_x, Yb, d, g0, g1):
    ana, _ = O.letkf_analysis(X.numpy()[:, :, g0:g1], grid_x.numpy()[g0:g1], obs_x.numpy(), Yb.numpy(), d.numpy(),
                              10.0, 1.1)
    return torch.from_numpy(ana)


def _oracle_chunk(X, grid_x, obs_x, Yb, d, c0, c1, state, buf):
    buf[:, :, :c1 - c0] = _oracle_shard(X, grid_x, obs_x, Yb, d, c0, c1)
    return lambda: 0


class _DecliningChunk:
    """Stand-in for a kernel that declines grid points: rank 1's first piece is written wrong and the
    declined-points counter raised; the deferred retry repairs the piece.  Every rank must then take the
    second exchange (agreement through the reduced counters), or the ranks dead-lock / keep the wrong data."""

    def __init__(self, rank):
        self.rank, self.fired, self.repaired = rank, False, 0

    def __call__(self, X, grid_x, obs_x, Yb, d, c0, c1, state, buf):
        good = _oracle_shard(X, grid_x, obs_x, Yb, d, c0, c1)
        if self.rank == 1 and not self.fired:
            self.fired = True
            buf[:, :, :c1 - c0] = 0.0
            state["vec"][2] += 3

            def finish():
                buf[:, :, :c1 - c0] = good
                self.repaired += 1
                return 3
            return finish
        buf[:, :, :c1 - c0] = good
        return lambda: 0


def _worker(rank, world, port, G, out_path, chunks=1, declining=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import torch_assimilate_amd as mia
    case = O.synthetic_case(G, 12, 2)
    X = torch.from_numpy(case["state"])
    runner = mia.ShardedLetkf("cpu", rank, world, radii=[10.0], inf_factor=1.1, compute_shard=_oracle_shard,
                              comm_chunks=chunks,
                              chunk_compute=(_DecliningChunk(rank) if declining else _oracle_chunk) if chunks > 1 else None)
    full = runner.assimilate(X, torch.from_numpy(case["grid_x"]), torch.from_numpy(case["obs_x"]),
                             torch.from_numpy(case["yb"]), torch.from_numpy(case["d"]))
    assert full.shape == X.shape
    if declining:
        assert runner.last_retries == (3 if rank == 1 else 0)
    gathered = [torch.empty_like(full) for _ in range(world)]
    dist.all_gather(gathered, full)
    for t in gathered:                      # every rank holds the same full analysis
        assert torch.equal(t, full)
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("G,chunks,declining", [(64, 1, False), (51, 1, False), (64, 4, False), (51, 4, False),
                                                 (37, 3, False), (51, 4, True)])
def test_two_rank_shard_and_allgather(tmp_path, G, chunks, declining):
    """chunks == 1: one all-gather of the whole block; chunks > 1: the compute / exchange-overlap path
    (block analysed in pieces, one all-gather per piece, uneven tails padded) -- same result.  declining:
    one rank's piece needs the deferred retry; both ranks must agree on the second exchange."""
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(2, _free_port(), G, out, chunks, declining), nprocs=2, join=True)
    got = np.load(out)
    case = O.synthetic_case(G, 12, 2)
    ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


def _worker_sharded_output(rank, world, port, G, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import torch_assimilate_amd as mia
    from torch_assimilate_amd.sharded import block_partition
    case = O.synthetic_case(G, 12, 2)
    X = torch.from_numpy(case["state"])
    runner = mia.ShardedLetkf("cpu", rank, world, radii=[10.0], inf_factor=1.1, compute_shard=_oracle_shard, gather=False,
                              comm_chunks=4)
    blk = runner.assimilate(X, torch.from_numpy(case["grid_x"]), torch.from_numpy(case["obs_x"]),
                            torch.from_numpy(case["yb"]), torch.from_numpy(case["d"]))
    g0, g1 = block_partition(G, world)[rank]
    assert blk.shape == (X.shape[0], X.shape[1], g1 - g0)
    np.save(os.path.join(out_dir, "blk%d.npy" % rank), blk.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("G", [64, 51])
def test_two_rank_sharded_output(tmp_path, G):
    """gather=False: every rank keeps its block of the analysis (what the reference's dask chunks along `grid` do,
    interface/letkf.py:118-131) -- no collective is entered; the blocks side by side are the full analysis."""
    mp.spawn(_worker_sharded_output, args=(2, _free_port(), G, str(tmp_path)), nprocs=2, join=True)
    case = O.synthetic_case(G, 12, 2)
    ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1)
    got = np.concatenate([np.load(str(tmp_path / ("blk%d.npy" % r))) for r in range(2)], axis=-1)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)

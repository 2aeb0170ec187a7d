"""The exchange of the one-process-per-GPU form as direct stores into the peers' buffers (csrc/gp_peer.hip, ShardedLinearizer(exchange="peer")): several processes on ONE GPU
(the IPC mapping, the arrival protocol and the two alternating generations are the same whether the peer buffer sits on this device or behind an xGMI link; the link itself is the
driver's to measure).  No reference counterpart: the loop being sharded is cuda/nonlinear_factor_set_gpu.cpp:64-139."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

RECORD = 122


def _rows(ids, step):
    ids = np.asarray(list(ids), dtype=np.float64)
    return ids[:, None] * 1000.0 + np.arange(RECORD, dtype=np.float64)[None, :] + 0.25 * step


def _worker(rank, world, port, rows, want, steps, ret, sabotage=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime

    from gtsam_points_amd.distributed import ShardedLinearizer

    torch.cuda.set_device(0)
    if sabotage and rank == 1:  # one rank cannot create / map the buffers: everybody must notice and fall back, nobody may hang
        from gtsam_points_amd import _capi

        setattr(_capi.load(), sabotage, lambda *a: 4)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    total = world * rows
    begin, end = rank * rows, (rank + 1) * rows
    host_out = torch.zeros((total, RECORD), dtype=torch.float64).pin_memory()
    state = dict(step=0)

    def issue(_poses, view):  # this rank's rows of the step, written on the device like the batch's kernels would
        view.copy_(torch.from_numpy(_rows(range(begin, end), state["step"])).cuda())

    lin = ShardedLinearizer(total, (begin, end), torch.device("cuda:0"), issue, always_exchange=True, exchange=want, host_out=host_out)
    seen = []
    for step in range(steps):
        state["step"] = step
        stacked = lin.linearize(None)
        if not lin.delivers_to_host:
            host_out.copy_(stacked, non_blocking=True)
        torch.cuda.synchronize()
        lin.check()
        assert np.array_equal(host_out.numpy(), _rows(range(total), step)), (rank, step, lin.exchange)
        assert np.array_equal(stacked.cpu().numpy(), _rows(range(total), step)), (rank, step, lin.exchange)
        seen.append(lin.exchange)
        if rank == step % world:  # ranks drift apart: one of them dawdles before the next step (a rank may be at most one exchange ahead of a peer)
            torch.cuda._sleep(2_000_000)
    ret[rank] = (lin.exchange, lin.peer_note, len(set(seen)))
    lin.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows", [(2, 1), (4, 1), (8, 1), (2, 64)])
def test_peer_exchange_between_processes_on_one_device(world, rows):
    port = 36500 + (os.getpid() * 13 + world * 17 + rows) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, rows, "peer", 12, ret), nprocs=world, join=True)
    for r in range(world):
        exchange, note, kinds = ret[r]
        assert kinds == 1  # decided once
        assert exchange == "peer", (r, exchange, note)  # the buffers could be shared and the validation passed (else: all ranks fell back together, and `note` says why)


def test_peer_exchange_falls_back_together_when_the_plan_does_not_qualify():
    """rows beyond the exchange's size limit (8192 doubles per rank): every rank takes the all-gather, the records are the same"""
    world, rows = 2, 80  # 80 x 122 = 9760 doubles
    port = 38500 + os.getpid() % 1500
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, rows, "peer", 3, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r][0] == "all_gather" and ret[r][1], ret[r]


@pytest.mark.parametrize("sabotage", ["gp_peer_exchange_create", "gp_peer_exchange_connect"])
def test_peer_exchange_falls_back_together_when_one_rank_cannot_share(sabotage):
    """a rank whose buffer cannot be created (no IPC, no fine-grained memory) or that cannot map a peer's: the set-up's decisions are collective -- every rank takes the
    all-gather, the records are the same, and the clean-up's barrier is everybody's (also the rank's that never had a buffer)"""
    world = 3
    port = 40500 + (os.getpid() + len(sabotage)) % 1500
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, 1, "peer", 4, ret, sabotage), nprocs=world, join=True)
    for r in range(world):
        assert ret[r][0] == "all_gather" and ret[r][1], ret[r]

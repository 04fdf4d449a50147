"""World-size-2 data-parallel step on CPU (gloo): the N > 1 training path of SURVEY.md §8(e).

The train-form modules are ordinary nn.Modules, so DistributedDataParallel wraps them unchanged; on
the GPU box the same code runs over RCCL (`backend="nccl"`).  Checks the DDP contract the reference
relies on (yolov6/core/engine.py:161-164, 485-487): per-rank batches differ, gradients after
backward are the rank average and identical on both ranks, BN statistics stay local (no SyncBN)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import maf_yolo_amd as M
    dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
    torch.manual_seed(0)                                    # same init on every rank (DDP also broadcasts rank 0)
    model = M.Model("n").train()
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    g = torch.Generator().manual_seed(100 + rank)           # different data per rank (train.py:93 seeds 1+rank)
    x = torch.rand(2, 3, 64, 64, generator=g)
    (feats, cls, reg), _ = ddp(x)
    loss = (cls.mean() + reg.pow(2).mean()) * world          # engine.py:161-162 pre-multiplies by world_size
    loss.backward()
    gw = model.backbone[0].rbr_dense.conv.weight.grad.clone()
    # local (un-reduced) gradient of the same loss for the averaging check
    ref = M.Model("n").train()
    ref.load_state_dict(model.state_dict())
    torch.manual_seed(0)
    (f2, c2, r2), _ = ref(x)
    ((c2.mean() + r2.pow(2).mean()) * world).backward()
    lw = ref.backbone[0].rbr_dense.conv.weight.grad.clone()
    gathered = [torch.zeros_like(lw) for _ in range(world)]
    dist.all_gather(gathered, lw)
    mean_local = sum(gathered) / world
    bn_mean = model.backbone[0].rbr_dense.bn.running_mean.clone()
    bns = [torch.zeros_like(bn_mean) for _ in range(world)]
    dist.all_gather(bns, bn_mean)
    if rank == 0:
        out.put(dict(avg_ok=torch.allclose(gw, mean_local, rtol=1e-4, atol=1e-6),
                     bn_local=not torch.allclose(bns[0], bns[1])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ddp_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res["avg_ok"], "DDP gradient is not the rank average"
    assert res["bn_local"], "BN statistics should stay per-rank (no SyncBN in the reference)"

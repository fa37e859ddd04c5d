"""CPU, world_size 2 over gloo: the data-parallel exchange of train.Trainer.

The N>1 path is: every rank runs forward/backward on its clip shard with the loss normalised by the GLOBAL batch,
then ONE summing all-reduce of the flat gradient bucket, then clip + Adam on identical buckets.  Here the per-shard
gradients come from the CPU oracle (tests may use it); the code under test is the sharding, the flat-bucket layout and
``allreduce_flat_grads`` -- exactly what runs over RCCL on the GPU box."""
import os
import socket
import sys

import pytest
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


def _worker(rank, world, port, q, B=8):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import weights as W
    from oracle import model_ref as MR
    import silent_speech_amd as ss

    T = 6
    sd = W.make_state_dict(5, 84, 5, True)
    X, L, R, y = W.make_inputs(5, B, T, 84, 5, (64, 64))
    lo, hi = ss.shard_range(B, rank, world)
    model = ss.BiGRUClassifier(84, 5, use_roi=True)  # CPU instance: only its flat-bucket layout is used here
    model.load_state_dict(sd)
    # shard gradient with the loss divided by the GLOBAL batch (what ss_ce_ls_fwd_bwd's `denom` does)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits = MR.forward(leaves, X[lo:hi], L[lo:hi], R[lo:hi])
    loss = MR.ce_label_smoothing(logits, y[lo:hi]) * (hi - lo) / B
    grads = torch.autograd.grad(loss, list(leaves.values()))
    bucket = torch.zeros_like(model.flat_params)
    views = model._views_of(bucket)
    for k, g in zip(leaves, grads):
        views[k].copy_(g)
    ss.allreduce_flat_grads(bucket)
    loss_t = loss.detach().clone()
    dist.all_reduce(loss_t)
    if rank == 0:
        _, _, full = MR.loss_and_grads(sd, X, L, R, y)
        full_loss = float(MR.ce_label_smoothing(MR.forward(sd, X, L, R), y))
        worst = max(float((views[k] - full[k]).abs().max() / max(float(full[k].abs().max()), 1e-3)) for k in full)
        q.put((worst, abs(float(loss_t) - full_loss), int(bucket.numel())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])  # 7: shards of 4 and 3 clips -- the loss is divided by the GLOBAL batch (Trainer.step(global_batch=))
def test_two_rank_flat_bucket_allreduce_equals_full_batch(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, B)) for r in range(2)]
    for p in procs:
        p.start()
    worst, dloss, n = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n == 1079588  # 1 079 582 parameters + alignment pads: the single all-reduce of DESIGN.md section 6
    assert worst < 2e-4, worst
    assert dloss < 1e-6, dloss

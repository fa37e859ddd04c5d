"""Child process of tests/test_gpu_model.py::test_one_rank_rccl_trainer_equals_single_process.

Launched by ``python -m torch.distributed.run --nproc-per-node 1 ...`` BEFORE anything in this process has touched the
GPU: joins a one-rank "nccl" (= RCCL) group, runs two fused training steps whose flat gradient bucket goes through
``dist.all_reduce`` and writes the resulting parameters / losses for the parent to compare."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main(out_path: str) -> None:
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    dist.init_process_group("nccl", rank=rank, world_size=world)
    import weights as W
    import silent_speech_amd as ss

    B, T = 16, 9
    sd = W.make_state_dict(21, 84, 5, True)
    X, L, R, y = W.make_inputs(21, B, T, 84, 5, (64, 64))
    m = ss.BiGRUClassifier(84, 5, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().train()
    tr = ss.Trainer(m, world_size=world, always_allreduce=True, dropout=True)
    losses = [float(tr.step(X.cuda(), L.cuda(), R.cuda(), y.cuda(), global_batch=B * world)[0]) for _ in range(2)]
    torch.cuda.synchronize()
    dist.barrier()
    torch.save({"losses": losses, "sd": {k: v.cpu() for k, v in m.state_dict().items()}, "backend": dist.get_backend(),
                "world": world}, out_path)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python3
"""Smoke test of bench.py's N > 1 code path on ONE device: torch.distributed is replaced by a
single-process stand-in (no transfer: the receive buffer stays zero), rank 0 of a 2-rank strip
partition runs the two-phase step.  Catches host-side errors of that path (argument plumbing, JSON
assembly) that the one-GPU pipeline cannot reach otherwise; it does not test RCCL."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Work:
    def wait(self):
        return True


def main():
    import torch
    import torch.distributed as dist
    os.environ.update(WORLD_SIZE=os.environ.get("SMOKE_WORLD", "2"), RANK=os.environ.get("SMOKE_RANK", "0"),
                      LOCAL_RANK="0")
    dist.init_process_group = lambda *a, **k: None
    dist.destroy_process_group = lambda *a, **k: None
    dist.barrier = lambda *a, **k: None
    dist.all_reduce = lambda t, *a, **k: None
    dist.broadcast = lambda t, *a, **k: None
    dist.batch_isend_irecv = lambda ops: [_Work() for _ in ops]

    class P2POp:  # descriptors only
        def __init__(self, op, tensor, peer, *a, **k):
            self.op, self.tensor, self.peer = op, tensor, peer

    dist.P2POp = P2POp
    # (--halo torch: a real RCCL communicator of N ranks cannot be made by one process)
    sys.argv = ["bench.py", "--gpus", os.environ["WORLD_SIZE"], "--steps", "10", "--warmup", "2", "--n", "200",
                "--halo", "torch"] \
        + sys.argv[1:]  # later flags override the defaults
    import bench
    bench.main()


if __name__ == "__main__":
    main()

"""Communicator objects with the mpi4py surface the reference uses (`Get_rank`, `Get_size`, `gather`,
`bcast`, `barrier`: `/root/reference/source/model_setup.py:21-23,97-105`, `solvers.py:86-99,205-208`).
mpi4py is not part of this stack: one process drives one GPU and `torch.distributed` is the bootstrap."""
from __future__ import annotations


class SerialComm:
    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def gather(self, obj, root=0):
        return [obj]

    def bcast(self, obj, root=0):
        return obj

    def barrier(self):
        return None

    Barrier = barrier


class TorchComm:
    """World communicator over an initialised `torch.distributed` process group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._d, self.group = dist, group

    def Get_rank(self):
        return self._d.get_rank(self.group)

    def Get_size(self):
        return self._d.get_world_size(self.group)

    def gather(self, obj, root=0):
        out = [None] * self.Get_size() if self.Get_rank() == root else None
        self._d.gather_object(obj, out, dst=root, group=self.group)
        return out

    def bcast(self, obj, root=0):
        box = [obj]
        self._d.broadcast_object_list(box, src=root, group=self.group)
        return box[0]

    def barrier(self):
        self._d.barrier(group=self.group)

    Barrier = barrier


COMM_WORLD = SerialComm()


def world():
    """SerialComm, or TorchComm when launched under torch.distributed.run."""
    import os
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import torch
            if torch.cuda.is_available():
                lr = int(os.environ.get("LOCAL_RANK", "0"))
                torch.cuda.set_device(lr)
                dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
            else:
                dist.init_process_group("gloo")
        return TorchComm()
    return COMM_WORLD

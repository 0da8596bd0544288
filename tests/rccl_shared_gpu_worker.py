"""Two ranks on ONE GPU asking for an RCCL communicator (RCCL refuses duplicate devices): exercises the collective
error path of distributed.make_context and the host-staged fallback of runner.make_runner."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from shakti_fenics_amd.runner import make_runner
args = types.SimpleNamespace(config="c1_12k", order="morton", dt=3600.0, storage=0, moulins=0, krylov_rtol=1e-10,
                             transport="rccl", precond="amg")
from shakti_fenics_amd._lib import ShaktiHipError
try:                      # without --allow-host-staged the refusal is an error, on every rank
    make_runner(args, rank, world, 0)
    refused = False
except ShaktiHipError as exc:
    refused = "RCCL communicator could not be created" in str(exc)
print(f"rank {rank}: refused {refused}", flush=True)
args.allow_host_staged = True
run = make_runner(args, rank, world, 0)
info = run.step(0)
print(f"rank {rank}: transport {run.transport}, step 0 newton {info.newton_its} krylov {info.krylov_its}", flush=True)
run.close()
dist.destroy_process_group()

"""Worker of the multi-rank md.solve() test (launched by torch.distributed.run): the reference's own launch line
`mpirun -np N python3 main.py setup` (/root/reference/notebooks/example.ipynb:61) with torch.distributed as bootstrap.
Every rank builds the setup with TorchComm and calls md.solve(); rank 0 then repeats the run on ONE rank and compares
the result files.  Transport gloo: the ranks share GPU 0 (SHK_DEVICE=0)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    from shakti_fenics_amd.comm import SerialComm, TorchComm
    from shakti_fenics_amd.setups import setup_synthetic_cooke2 as S
    root, precond = sys.argv[1], sys.argv[2]
    md = S.initialize(TorchComm(), nx=45, ny=45, days=5.0 / 24.0, results_root=os.path.join(root, "multi"))
    md.transport, md.preconditioner = "gloo", precond
    assert md.device == 0 and md.size == dist.get_world_size()
    md.solve()
    dist.barrier()
    ok = True
    if dist.get_rank() == 0:
        one = S.initialize(SerialComm(), nx=45, ny=45, days=5.0 / 24.0, results_root=os.path.join(root, "single"))
        one.preconditioner = precond
        one.solve()
        errs = {}
        for k in ("b", "N", "qx", "qy"):
            a, b = np.load(f"{md.results_name}/{k}.npy"), np.load(f"{one.results_name}/{k}.npy")
            errs[k] = float(np.linalg.norm(a - b) / np.linalg.norm(b)) if a.shape == b.shape else float("inf")
        same_nodes = bool(np.array_equal(np.load(f"{md.results_name}/nodes_x.npy"), np.load(f"{one.results_name}/nodes_x.npy")))
        its = [list(map(int, np.load(f"{r}/newton_its.npy"))) for r in (md.results_name, one.results_name)]
        ok = max(errs.values()) < 1e-7 and same_nodes and its[0] == its[1]
        print(json.dumps(dict(ok=ok, errs=errs, newton=its, same_nodes=same_nodes)), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

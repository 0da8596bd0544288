"""Command-line driver with the call sequence of `/root/reference/source/main.py:11-21`:

    python -m shakti_fenics_amd.main <setup_module>
    python -m torch.distributed.run --nproc-per-node 8 -m shakti_fenics_amd.main <setup_module>

The setup module is looked up in `shakti_fenics_amd/setups/` and on sys.path; it must expose
`initialize(comm) -> md`."""
import importlib
import os
import sys

from .comm import world

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "setups"))


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 2:
        raise SystemExit("usage: python -m shakti_fenics_amd.main <setup_module>")
    comm = world()
    setup = importlib.import_module(argv[1])
    md = setup.initialize(comm)
    md.solve()


if __name__ == "__main__":
    main()

"""TEST INFRASTRUCTURE ONLY: bench.py's launch contract on a GPU-less host.

Runs bench.main() with a stand-in for its device class: the kernels execute on the CPU SIMT emulator of this directory
and the ranks talk over gloo.  bench.py itself knows nothing about this file (it has ONE backend: the MI355X); the
launch-contract tests (tests/test_bench_launch.py) start this script where the driver would start bench.py, with the
same command-line flags."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import torch  # noqa: E402


class EmulatedDevice:
    product = False
    dist_backend = "gloo"
    collective = "gloo (CPU emulator of the kernel sources)"

    def check(self, gpus=1):
        import emu_backend
        emu_backend.use_emulator()

    def collective_version(self):
        return "gloo"

    def device(self, local):
        return torch.device("cpu")

    def sync(self):
        pass

    def memory_stats(self, dev):
        return {}


if __name__ == "__main__":
    import bench
    bench.main(backend=EmulatedDevice())

#!/usr/bin/env python
"""TEST INFRASTRUCTURE: tools/run_reference.py with the CPU oracle behind the three extension modules, for boxes without a GPU
(tests/test_reference_scripts_cpu.py, tools/reference_train_py_cpu_demo.py).  The extension modules are backed by the CPU
oracle (oracle/*.c) and the reference's hard-coded device="cuda" is redirected to the CPU (tests/reference_cpu_backend.py).  It
exists to show that the reference's own scripts run unchanged across this repo's extension boundary -- and, with `--dp N
--dp-backend gloo --dp-share-device`, that the data-parallel launcher keeps replicas identical; it is not a product path: the
user-facing launcher has no such switch and fails loudly without a GPU.

    python tests/run_reference_cpu.py [run_reference.py options] -- train.py ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _oracle_backend():
    from tests import reference_cpu_backend
    reference_cpu_backend.install()


if __name__ == "__main__":
    import run_reference
    run_reference.main(install_backend=_oracle_backend, entry=os.path.abspath(__file__))

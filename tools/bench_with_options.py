#!/usr/bin/env python
"""bench.py with library options set first (A/B runs):  python tools/bench_with_options.py BINNING_BLOCK_K=4 -- <bench.py arguments>"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cut = sys.argv.index("--")
pairs = [a.split("=") for a in sys.argv[1:cut]]
os.environ.setdefault("OMP_NUM_THREADS", "8")
from relightable3dgaussian_amd import _lib  # noqa: E402

for k, v in pairs:
    _lib.set_option(k, int(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[cut + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")

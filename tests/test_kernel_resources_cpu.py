"""Static guard on the hot kernels' code objects (no GPU): hipcc's own metadata for gfx950 must show NO scratch memory for any
instantiation of the tile kernels and of the fixed-ray-set shading kernels.  Twice in round 4 a refactor that kept every parity
test green put a per-lane array into scratch (a feature row returned by value from a helper; a float4 stored whole into a local
array) -- the relight frame lost 17 % before anybody timed it.  Compiles the sources exactly as relightable3dgaussian_amd/build.py
does, to assembly only."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from relightable3dgaussian_amd import build as B

HOT = {
    "rasterizer_render_fwd.hip": [r"render_forward_wave_kernel"],
    "rasterizer_render_bwd.hip": [r"render_backward_wave_kernel", r"render_backward_features_kernel"],
    "shading.hip": [r"shade_forward_frs_kernel", r"shade_backward_frs_kernel", r"shade_forward_transport_kernel",
                    r"shade_forward_split_kernel"],
    "stage2_glue.hip": [r"s2_smooth_stream_kernel"],
}


def _metadata(src, tmp):
    out = os.path.join(tmp, src + ".s")
    flags = [f for f in B.COMMON + B.EXTRA.get(src, []) if f not in ("-c", "-fPIC")]
    r = subprocess.run([B.HIPCC] + flags + ["--cuda-device-only", "-S", os.path.join(B.CSRC, src), "-o", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    kernels = {}
    # amdhsa.kernels metadata: one YAML map per kernel, fields in alphabetical order (.name ... .private_segment_fixed_size ...)
    for block in re.split(r"\n\s+- \.agpr_count:", text)[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
        vgpr = re.search(r"\.vgpr_count:\s+(\d+)", block)
        if name and scratch:
            kernels[name.group(1)] = (int(scratch.group(1)), int(vgpr.group(1)) if vgpr else -1)
    return src, kernels


def test_hot_kernels_use_no_scratch_memory(tmp_path):
    if not os.path.exists(B.HIPCC):
        pytest.skip("no hipcc")
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda s: _metadata(s, str(tmp_path)), HOT))
    seen = 0
    bad = []
    for src, kernels in results:
        assert kernels, "no kernel metadata found in the assembly of %s" % src
        for name, (scratch, vgprs) in kernels.items():
            if any(re.search(p, name) for p in HOT[src]):
                seen += 1
                if scratch != 0:
                    bad.append("%s: %d bytes of scratch per lane (%d VGPRs)" % (name, scratch, vgprs))
    assert seen >= 40, "expected the tile kernels' instantiations and the shading kernels, saw %d" % seen
    assert not bad, "\n".join(bad)

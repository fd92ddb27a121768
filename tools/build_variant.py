#!/usr/bin/env python
"""Experiment builds of libr3dg_hip.so (no GPU needed): one source file of csrc/ is patched IN A TEMPORARY COPY (exact string
replacements and / or extra hipcc flags), compiled, and linked with the product's other objects into
relightable3dgaussian_amd/lib/variants/<name>/libr3dg_hip.so -- git-ignored, travels to the GPU box with the snapshot.
`R3DG_LIB_PATH=<that file>` makes relightable3dgaussian_amd/_lib.py load it (experiments only).  The product sources are never
touched: ablations ("how long does the kernel take WITHOUT its texture atomics?") and A/B candidates live in the variant table
of the calling script, not behind switches in the shipped kernels.

    from tools.build_variant import build_variant
    build_variant("no_env_atomics", "shading.hip", [("atomicAdd(reinterpret_cast<unsigned long long*>(&s_denv", "if (0) atomicAdd(...")])
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from relightable3dgaussian_amd import build as B   # noqa: E402

VARIANTS = os.path.join(B.LIBDIR, "variants")


def build_variant(name, src, replacements=(), extra_flags=(), count=None):
    """-> path of the variant library.  `replacements`: (old, new) pairs, each `old` must occur exactly once (or `count`
    times when given as a third element)."""
    B.build(verbose=False)                                   # the product objects the variant links against
    texts = {}
    for rep in replacements:
        old, new = rep[0], rep[1]
        n = rep[2] if len(rep) > 2 and rep[2] is not None else 1
        fname = rep[3] if len(rep) > 3 else src              # (a header of csrc/ the source includes, e.g. "shading_frs.hpp")
        text = texts.get(fname) or open(os.path.join(B.CSRC, fname)).read()
        if text.count(old) != n:
            raise RuntimeError("variant %s: %r occurs %d times in %s, expected %d" % (name, old[:60], text.count(old), fname, n))
        texts[fname] = text.replace(old, new)
    texts.setdefault(src, open(os.path.join(B.CSRC, src)).read())
    out_dir = os.path.join(VARIANTS, name)
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        # the patched file must sit beside the headers it includes
        for f in os.listdir(B.CSRC):
            if f.endswith(".hpp"):
                shutil.copy(os.path.join(B.CSRC, f), tmp)
        for fname, text in texts.items():
            with open(os.path.join(tmp, fname), "w") as fh:
                fh.write(text)
        path = os.path.join(tmp, src)
        obj = os.path.join(tmp, src[:-4] + ".o")
        flags = B.COMMON + B.EXTRA.get(src, []) + list(extra_flags)
        r = subprocess.run([B.HIPCC] + flags + ["-c", path, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("variant %s: hipcc failed\n%s" % (name, r.stderr[-3000:]))
        objs = [obj if f == src else os.path.join(B.OBJDIR, f[:-4] + ".o") for f in B._sources()]
        lib = os.path.join(out_dir, "libr3dg_hip.so")
        r = subprocess.run([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("variant %s: link failed\n%s" % (name, r.stderr[-3000:]))
    return lib


if __name__ == "__main__":
    print(build_variant(sys.argv[1], sys.argv[2], extra_flags=sys.argv[3:]))

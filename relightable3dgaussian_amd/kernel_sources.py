"""Which source file defines a kernel, and a fingerprint of it -- so that committed counter files (profiles/*_pmc_*.json, written
by tools/pmc_valu.py / pmc_traffic.py) can say which SOURCE their numbers describe and bench.py can tell when the kernel it has
just timed is a newer one (counters two commits older than the kernel were pasted into the bench line)."""
import hashlib
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_CALL = re.compile(r"\b([A-Za-z_]\w*)\s*\(")
_NOT_A_NAME = {"__launch_bounds__", "__attribute__", "void", "amdgpu_flat_work_group_size", "amdgpu_waves_per_eu"}
_INCLUDE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)
_cache = {}


def _text(path):
    with open(path, "r", errors="replace") as fh:
        return fh.read()


def _closure(fname, seen):
    """fname + the local headers it includes, transitively (sorted, each once)."""
    if fname in seen or not os.path.exists(os.path.join(CSRC, fname)):
        return
    seen.add(fname)
    for inc in _INCLUDE.findall(_text(os.path.join(CSRC, fname))):
        _closure(os.path.basename(inc), seen)


def kernel_files():
    """{kernel name: file under csrc/ that holds its __global__ definition} (template kernels by their plain name)."""
    if "files" not in _cache:
        out = {}
        for f in sorted(os.listdir(CSRC)):
            if not f.endswith((".hip", ".hpp")):
                continue
            text = _text(os.path.join(CSRC, f))
            for m in re.finditer(r"__global__", text):
                # the kernel's name: the first identifier followed by "(" behind the qualifier that is not an attribute
                for c in _CALL.finditer(text, m.end(), m.end() + 600):
                    if c.group(1) not in _NOT_A_NAME:
                        out.setdefault(c.group(1), f)
                        break
        _cache["files"] = out
    return _cache["files"]


def source_of(kernel):
    """-> (file, sha256 over that file and the local headers it includes) or (None, None) for an unknown kernel name."""
    f = kernel_files().get(kernel)
    if f is None:
        return None, None
    if f not in _cache:
        seen = set()
        _closure(f, seen)
        h = hashlib.sha256()
        for name in sorted(seen):
            h.update(name.encode() + b"\0")
            h.update(_text(os.path.join(CSRC, name)).encode())
        _cache[f] = h.hexdigest()
    return f, _cache[f]


def stamp(kernels):
    """{kernel: {"file", "sha256"}} for a counter file."""
    out = {}
    for k in kernels:
        f, sha = source_of(k)
        if f is not None:
            out[k] = {"file": "relightable3dgaussian_amd/csrc/" + f, "sha256": sha}
    return out

"""Static resources of every kernel in libr3dg_hip.so (VGPRs, AGPRs, SGPRs, LDS bytes, scratch, occupancy in waves/SIMD)
from hipcc's own resource remarks:  python tools/kernel_resources.py profiles/rNN_kernel_resources.json
Compiles each csrc/*.hip with the flags of relightable3dgaussian_amd/build.py plus -Rpass-analysis=kernel-resource-usage
(no GPU needed)."""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from relightable3dgaussian_amd import build as B   # noqa: E402

FIELDS = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
          "Occupancy [waves/SIMD]": "waves_per_simd_limit", "LDS Size [bytes/block]": "lds_bytes",
          "TotalSGPRs": "sgprs", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills"}


def one(src):
    flags = B.COMMON + B.EXTRA.get(src, []) + ["-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run([B.HIPCC] + flags + ["-c", os.path.join(B.CSRC, src), "-o", "/dev/null"], capture_output=True, text=True)
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"^void ", "", name)
            cur = out.setdefault(name.split("(")[0], {"file": src})
            continue
        m = re.search(r"remark: .*?\s{2,}([A-Za-z][^:]*): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            cur[FIELDS[m.group(1).strip()]] = int(m.group(2))
    return out


def main():
    srcs = B._sources()
    with ThreadPoolExecutor(max_workers=8) as ex:
        res = {}
        for d in ex.map(one, srcs):
            res.update(d)
    doc = {"note": "hipcc -Rpass-analysis=kernel-resource-usage, gfx950, flags of relightable3dgaussian_amd/build.py",
           "kernels": dict(sorted(res.items()))}
    out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    json.dump(doc, open(out, "w"), indent=1)
    sys.stderr.write("wrote %s (%d kernels)\n" % (out, len(res)))


if __name__ == "__main__":
    main()

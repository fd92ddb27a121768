#!/usr/bin/env python
"""Did a source change alter the code of kernels it was not supposed to touch?  Compares two gfx950 assembly listings of
the same csrc/*.hip (before / after), kernel by kernel:

    hipcc <flags of relightable3dgaussian_amd/build.py> --cuda-device-only -S csrc/shading.hip -o before.s     (old tree)
    hipcc ...                                                                  -o after.s      (new tree)
    python tools/asm_diff.py before.s after.s

Per kernel: `identical` (instruction text equal after normalising label numbers, function-static symbol names and the
implicit-argument offset), `registers` (same opcode sequence, different register names), `different`, or `new` / `gone`.
Kernels are matched by mangled name; a kernel whose template list only gained trailing defaulted arguments
(...ELb0EEEv...) is matched with its predecessor.  Used for the claims in DESIGN.md section 8 that the opt-in variants left
the default kernels' code alone."""
import re
import sys


def kernels(path):
    out, name = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            out[name] = []
            continue
        if line.startswith(".Lfunc_end"):
            name = None
        if name and not line.strip().startswith((";", ".")):
            out[name].append(line.split(";")[0].rstrip())
    return {k: v for k, v in out.items() if any(x.strip().startswith(("s_", "v_")) for x in v)}


def norm(lines):
    o = []
    for l in lines:
        l = re.sub(r"\.LBB\d+_", ".LBB_", l)
        l = re.sub(r"_ZZ?N4r3dg\w+", "SYM", l)
        l = re.sub(r"(s_add_u32 s\d+, s\d+, )0x[0-9a-f]+", r"\1IMM", l)
        o.append(l)
    return o


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    counts = {"identical": 0, "registers": 0, "different": 0, "gone": 0}
    matched = set()
    for ka, va in a.items():
        stem = ka.split("EEv")[0]
        cands = [kb for kb in b if kb == ka] or [kb for kb in b if kb.startswith(stem) and re.match(r"^(Lb0E)+EEv", kb[len(stem):])]
        if not cands:
            counts["gone"] += 1
            print("gone       ", ka[:110])
            continue
        kb = cands[0]
        matched.add(kb)
        vb = b[kb]
        if norm(va) == norm(vb):
            counts["identical"] += 1
        elif [l.split()[0] for l in va if l.strip()] == [l.split()[0] for l in vb if l.strip()]:
            counts["registers"] += 1
            print("registers  ", ka[:110])
        else:
            counts["different"] += 1
            print("different  ", ka[:110], len(va), "->", len(vb))
    new = [k for k in b if k not in matched]
    for k in new:
        print("new        ", k[:110])
    print(counts, "new:", len(new))


if __name__ == "__main__":
    main()

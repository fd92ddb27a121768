"""TEST INFRASTRUCTURE: the emulated transport-cache kernels (tests/emu/transport_emu.cpp) under AddressSanitizer or
ThreadSanitizer -- out-of-bounds indexing at edge sizes (P = 1, K = 1, partially filled waves and workgroups) and races
between the lanes of a wave / the waves of a workgroup show up here before the source ever runs on a GPU.

    python tests/emu/sanitize_run.py address      # or: thread
(re-executes itself with the sanitizer runtime preloaded; needs g++ with libasan / libtsan)."""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CASES = {"address": ((130, 37, True, True), (9, 100, False, False), (5, 1, True, True), (1, 65, False, True)),
         "thread": ((9, 37, True, True), (5, 70, False, False))}


def child(kind, lib_path):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from tests.test_oracle_cpu import _transport_case, _transport_in_torch
    lib = C.CDLL(lib_path)
    for P, K, uniform, regen in CASES[kind]:
        c = _transport_case(max(P, 2), K, seed=5)
        if P < 2:
            c = {k: (v[:P] if (torch.is_tensor(v) and k != "zs" and k != "env") else v) for k, v in c.items()}
        radiance, _t, _c, _L, out = _transport_in_torch(c)
        f = lambda t: np.ascontiguousarray(t.detach().numpy(), np.float32).copy()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        buf, consts, got = f(radiance), np.zeros((P, 16), np.float32), np.zeros((P, 19), np.float32)
        normals, inc, vis, dirs, areas = f(c["normals"]), f(c["inc"]), f(c["vis"]), f(c["dirs"]), f(c["areas"])
        lib.emu_shade_build_transport(C.c_int(P), C.c_int(K), C.c_int(16), p(normals), p(inc), p(vis), p(dirs),
                                      None if uniform else p(areas), C.c_float(float(c["areas"][0, 0, 0])), p(buf), p(consts))
        base, rough, view, zs = f(c["base"]), f(c["rough"]), f(c["view"]), f(c["zs"])
        lib.emu_shade_forward_transport(C.c_int(P), C.c_int(K), p(base), p(rough), p(normals), p(view), p(buf), p(consts),
                                        p(zs), None if regen else p(dirs), p(got))
        err = float(np.abs(got - f(out)).max() / np.abs(f(out)).max())
        print("%s sanitizer: P=%d K=%d clean, rel err %.1e" % (kind, P, K, err))
        assert err < 1e-4


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "address"
    if len(sys.argv) > 2:
        return child(kind, sys.argv[2])
    lib_path = "/tmp/libtransport_emu_%s.so" % kind
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-fsanitize=" + kind, "-fno-omit-frame-pointer", "-pthread",
                           "-shared", "-fPIC", "-w", "-I", os.path.join(ROOT, "relightable3dgaussian_amd", "csrc"),
                           os.path.join(HERE, "transport_emu.cpp"), "-o", lib_path])
    rt = subprocess.check_output(["gcc", "-print-file-name=lib%s.so" % ("asan" if kind == "address" else "tsan")], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="report_signal_unsafe=0")
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), kind, lib_path], env=env))


if __name__ == "__main__":
    main()

"""Per-step GPU timeline from a rocprofv3 rocpd database: for the last N training steps (delimited by consecutive
render_backward_kernel launches) print wall time, GPU-busy time, idle gaps, launches per step and time by kernel.

    python tools/rocpd_timeline.py <results.db> [N]
"""
import collections
import sqlite3
import sys


def main():
    db = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "r3dg::render_backward_wave" in r[0] or "r3dg::render_backward_kernel" in r[0]]
    marks = marks[-(n + 1):]
    if len(marks) < 2:
        print("not enough steps")
        return
    per = collections.defaultdict(float)
    cnt = collections.Counter()
    wall = busy = covered = 0.0
    launches = 0
    gaps = []
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a:b]
        wall += rows[b][1] - rows[a][1]
        # union of the kernel intervals (kernels on different streams overlap): what is left is time with NO kernel running
        hi = rows[a][1]
        for name, s, e in seg:
            if s > hi:
                gaps.append((s - hi, name[:60]))
                hi = s
            if e > hi:
                covered += min(e, rows[b][1]) - hi
                hi = e
        for name, s, e in seg:
            busy += e - s
            per[name[:90]] += e - s
            cnt[name[:90]] += 1
        launches += len(seg)
    k = len(marks) - 1
    print("steps %d  wall %.3f ms/step  gpu busy %.3f ms/step  idle %.3f ms/step  launches/step %.1f" %
          (k, wall / k / 1e6, busy / k / 1e6, (wall - busy) / k / 1e6, launches / k))
    print("some kernel running %.3f ms/step; nothing running %.3f ms/step (%.1f %% of the step), in %d gaps/step" %
          (covered / k / 1e6, (wall - covered) / k / 1e6, 100.0 * (wall - covered) / wall, len(gaps) / k))
    by = collections.defaultdict(float)
    for g, name in gaps:
        by[name] += g
    for name, v in sorted(by.items(), key=lambda x: -x[1])[:8]:
        print("   gap before %-60s %7.1f us/step" % (name, v / k / 1e3))
    r3dg = sum(v for kname, v in per.items() if "r3dg::" in kname)
    print("r3dg kernels %.3f ms/step, other (torch) kernels %.3f ms/step" % (r3dg / k / 1e6, (busy - r3dg) / k / 1e6))
    for name, v in sorted(per.items(), key=lambda x: -x[1])[:45]:
        print("%8.1f us  x%-5.1f %s" % (v / k / 1e3, cnt[name] / k, name))


def sequence():
    """Print the kernel sequence of one full step (see the comment on argv[3] below): start (us from the step's first launch), duration (us), the queue the
    launch went to (one per HIP stream), the idle time on the whole device before it when nothing else was running, name."""
    cur = sqlite3.connect(sys.argv[1]).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(cur.execute("select name, start, end, %s from kernels order by start" % qcol))
    marks = [i for i, r in enumerate(rows) if "r3dg::render_backward_wave" in r[0] or "r3dg::render_backward_kernel" in r[0]]
    # argv[3] = which step (0-based index into the render_backward launches; negative from the end).  Default: the step in the
    # MIDDLE of the run -- the last steps of a bench.py run are the one-stream pass that measures each kernel alone
    # (FusedStage2Step.serial_streams), not the pipelined iteration
    k = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) // 2
    a, b = marks[k], marks[k + 1] if k + 1 < len(marks) and k != -1 else marks[-1]
    if k < 0:
        a, b = marks[k - 1], marks[k]
    t0 = rows[a][1]
    queues = {}
    hi = rows[a][1]
    print("# one step (from a render_backward launch to the next): start us | duration us | hardware queue (q0 = the main stream's) | "
          "idle: time with NO kernel running on the device in front of this launch | kernel")
    for name, s, e, q in rows[a:b]:
        gap = (s - hi) / 1e3 if s > hi else 0.0
        hi = max(hi, e)
        print("%9.1f %7.1f  q%-2d %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, queues.setdefault(q, len(queues)),
                                           ("gap %5.1f" % gap) if gap > 0 else "         ", name[:110]))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "seq":
        sequence()
        sys.exit(0)
    main()

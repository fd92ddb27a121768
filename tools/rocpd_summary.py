"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a small markdown/CSV table.

    python tools/rocpd_summary.py gpurun_out/prof_x/bench_results.db profiles/r01_x_kernel_stats.md "<command line>"
"""
import sqlite3
import sys


def main():
    db_path, out_path = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    # the top_kernels view reports microseconds
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\ncommand: `%s`\n\n" % cmd)
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for name, calls, total, avg, pct in rows[:60]:
            f.write("| `%s` | %d | %.3f | %.2f | %.2f |\n" % (name[:110].replace("|", "/"), calls, total / 1e3,
                                                            avg, pct))
    print("wrote", out_path, "(%d kernels)" % len(rows))


if __name__ == "__main__":
    main()

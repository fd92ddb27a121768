"""Merge rocprofv3 --pmc SQ-counter passes (rocpd sqlite databases; ONE counter group per pass, each pass also carrying
--kernel-trace for the dispatch durations) into profiles/<round>_pmc_valu.json: per kernel, mean per dispatch.

    python tools/pmc_valu.py out.json "<note>" [--resources kernel_resources.json] db1 [db2 ...]

Derived figures (raw counter means are kept beside them so they can be recomputed):
  clock_ghz               = GRBM_GUI_ACTIVE / dispatch duration   (GRBM_GUI_ACTIVE is reported once per XCD and summed by the
                            tool chain: divided by 8 when the quotient is otherwise implausible; stated in `gui_active_div`)
  valu_busy_frac          = 4*SQ_ACTIVE_INST_VALU / (1024 SIMDs * cycles)     (rocprofiler-sdk's VALUBusy for gfx950;
                            SQ_ACTIVE_INST_* count quad-cycles, MI355X_MICROARCH.md)
  valu_issue_frac         = 2*SQ_INSTS_VALU / (1024 SIMDs * cycles): lower bound on the time the SIMDs spend issuing VALU
                            (a plain fp32 wave64 instruction issues over 2 cycles; packed / transcendental ones take longer)
  waves_per_simd          = 4*SQ_WAVE_CYCLES / (1024 * cycles)                 (mean resident waves per SIMD)
  wait_frac / issue_stall_frac / active_frac = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
  lds_bank_conflict_frac  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE           (conflict cycles per LDS-active cycle)
  lds_issue_frac / trans_frac / mfma_busy_frac / salu_issue_frac = 4*SQ_ACTIVE_INST_LDS, 16*SQ_INSTS_VALU_TRANS_F32,
                            SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_SALU over (1024 SIMDs * cycles); wait_lds_frac = SQ_WAIT_INST_LDS /
                            SQ_WAVE_CYCLES
with cycles = GRBM_GUI_ACTIVE of the same pass."""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"r3dg::(\w+)", name)
    return m.group(1) if m else None


def main():
    argv = sys.argv[1:]
    out, note = argv[0], argv[1]
    rest = argv[2:]
    res = None
    if rest and rest[0] == "--resources":
        res = json.load(open(rest[1]))["kernels"]
        rest = rest[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for db in rest:
        cur = sqlite3.connect(db).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        namecol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
        cntcol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
        for k, c, v in cur.execute("select %s, %s, value from counters_collection" % (namecol, cntcol)):
            s = short(k)
            if s is None:
                continue
            a = acc[s][c]
            a[0] += float(v)
            a[1] += 1
        try:
            for k, s0, e0 in cur.execute("select name, start, end from kernels"):
                s = short(k)
                if s is not None:
                    dur[s][0] += (e0 - s0) * 1e-9
                    dur[s][1] += 1
        except sqlite3.Error:
            pass
    kernels = {}
    for k, d in sorted(acc.items()):
        m = {c: v[0] / max(v[1], 1) for c, v in d.items()}
        row = {"counters_mean_per_dispatch": {c: round(x, 1) for c, x in sorted(m.items())}}
        t = dur[k][0] / dur[k][1] if dur[k][1] else None
        gui = m.get("GRBM_GUI_ACTIVE")
        if gui:
            div = 1
            if t and gui / t / 1e9 > 4.0:
                div = 8
            cyc = gui / div
            row["gui_active_div"] = div
            row["cycles"] = round(cyc, 1)
            if t:
                row["duration_us_under_pmc"] = round(t * 1e6, 2)
                row["clock_ghz"] = round(cyc / t / 1e9, 3)
            simd = 1024.0
            if "SQ_ACTIVE_INST_VALU" in m:
                row["valu_busy_frac"] = round(4 * m["SQ_ACTIVE_INST_VALU"] / (simd * cyc), 4)
            if "SQ_INSTS_VALU" in m:
                row["valu_issue_frac"] = round(2 * m["SQ_INSTS_VALU"] / (simd * cyc), 4)
            if "SQ_WAVE_CYCLES" in m:
                row["waves_per_simd"] = round(4 * m["SQ_WAVE_CYCLES"] / (simd * cyc), 3)
        if m.get("SQ_WAVE_CYCLES"):
            for key, c in (("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"), ("active_frac", "SQ_ACTIVE_INST_ANY")):
                if c in m:
                    row[key] = round(m[c] / m["SQ_WAVE_CYCLES"], 4)
        if gui:
            # shares of the SIMDs' time the other issue ports are busy: LDS instructions in flight, the
            # quarter-rate transcendental unit (a wave64 v_exp / v_rcp / v_rsq occupies it 16 cycles), the matrix pipe
            if "SQ_ACTIVE_INST_LDS" in m:
                row["lds_issue_frac"] = round(4 * m["SQ_ACTIVE_INST_LDS"] / (simd * cyc), 4)
            if "SQ_INSTS_VALU_TRANS_F32" in m:
                row["trans_frac"] = round(16 * m["SQ_INSTS_VALU_TRANS_F32"] / (simd * cyc), 4)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                row["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (simd * cyc), 4)
            if "SQ_INSTS_SALU" in m:
                row["salu_issue_frac"] = round(m["SQ_INSTS_SALU"] / (simd * cyc), 4)
        if m.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_LDS" in m:
            row["wait_lds_frac"] = round(m["SQ_WAIT_INST_LDS"] / m["SQ_WAVE_CYCLES"], 4)
        if m.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in m:
            row["lds_bank_conflict_frac"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 4)
        if res is not None:
            cands = [v for n, v in res.items() if re.search(r"r3dg::%s\b" % re.escape(k), n)]
            if cands:
                row["vgprs"] = sorted({c.get("vgprs") for c in cands if c.get("vgprs") is not None})
                row["lds_bytes"] = sorted({c.get("lds_bytes") for c in cands if c.get("lds_bytes") is not None})
                row["waves_per_simd_limit"] = sorted({c.get("waves_per_simd_limit") for c in cands
                                                      if c.get("waves_per_simd_limit") is not None})
                if len(cands) > 1:
                    row["resources_note"] = "%d template instances; sets of values over all of them" % len(cands)
        kernels[k] = row
    # which SOURCE these counters describe: sha256 of each kernel's defining file (+ its local headers) at collection time;
    # bench.py compares it with the tree it runs from and marks the figures stale when they differ
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from relightable3dgaussian_amd import kernel_sources
    json.dump({"note": note, "sources": kernel_sources.stamp(sorted(kernels)), "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turns a rocprofv3 --kernel-trace --stats results .db into the per-kernel summary committed under profiles/.

Rows are grouped by (kernel, grid, workgroup): one kernel name covers launches of very different shapes (k_fb_accumulate runs the 4096-blob
step of the headline, every batch of the batch sweep and the lone commitments of the latency block), and a per-name average says nothing
about any of them.  With a shape per row, the launch bench.py puts on its roofline line is ONE row of the table (bench.py reads the .json
written beside the .md and prints that row's average as roofline.profile_avg_ms next to its own HIP-event figure).

usage: tools/rocprof_summary.py <results.db> [<out.md> [<out.json>]]"""
import json
import re
import sqlite3
import sys


def short_name(name):
    short = name.split("(")[0].replace("kzg::", "")
    m = re.match(r"_ZN3kzg(\d+)", short)              # rocprofv3 7.x stores mangled names: <length><name>[I<template args>E]E<parameters>
    if not m:
        m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", short) or re.match(r"_ZN12_GLOBAL__N_1(\d+)", short)   # anonymous namespace (capi_multi.hip)
    if m:
        n0 = m.end(); base = short[n0:n0 + int(m.group(1))]; rest = short[n0 + int(m.group(1)):]
        tm = re.match(r"I((?:L[ibj]\d+E)+)E", rest)
        targs = re.findall(r"L[ibj](\d+)E", tm.group(1)) if tm else []
        short = base + ("<" + ", ".join(targs) + ">" if targs else "")
    else:
        m = re.match(r"_Z(\d+)", short)                # global namespace
        if m:
            short = short[m.end():m.end() + int(m.group(1))]
    return short


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    rows = c.execute(
        "select s.kernel_name, d.grid_size_x, d.workgroup_size_x, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start), max(d.private_segment_size), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name, d.grid_size_x, d.workgroup_size_x order by 5 desc").fetchall()
    tot = sum(r[4] for r in rows) or 1
    lines = ["| kernel | grid (lanes) | wg | workgroups | calls | total ms | avg us | min us | max us | % | scratch B/lane | LDS B |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    out_rows = []
    for name, grid, wg, n, t, a, mn, mx, scr, lds in rows:
        short = short_name(name)
        out_rows.append({"kernel": short, "grid": grid, "workgroup": wg, "workgroups": grid // max(wg, 1), "calls": n, "total_ms": t / 1e6, "avg_us": a / 1e3,
                         "min_us": mn / 1e3, "max_us": mx / 1e3, "scratch_bytes_per_lane": scr, "lds_bytes": lds})
        if 100.0 * t / tot < 0.02 and n < 3:
            continue                                   # one-off set-up kernels stay in the .json only
        lines.append("| %s | %d | %d | %d | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %d | %d |" % (short, grid, wg, grid // max(wg, 1), n, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3,
                                                                                          100.0 * t / tot, scr, lds))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(out)
    if len(sys.argv) > 3:
        json.dump({"rows": out_rows}, open(sys.argv[3], "w"), indent=0)
    print(out)


if __name__ == "__main__":
    main()

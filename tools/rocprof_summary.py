#!/usr/bin/env python3
"""Turns a rocprofv3 --kernel-trace --stats results .db into the per-kernel summary committed under profiles/.
usage: tools/rocprof_summary.py <results.db> [<out.md>]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    rows = c.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
        "max(d.workgroup_size_x), max(d.grid_size_x), max(d.private_segment_size), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | wg | grid | scratch B/lane | LDS B |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for name, n, t, a, mn, mx, wg, grid, scr, lds in rows:
        short = name.split("(")[0].replace("kzg::", "")
        m = re.match(r"_ZN3kzg(\d+)", short)              # rocprofv3 7.x stores mangled names: <length><name>[I<template args>E]E<parameters>
        if m:
            n0 = m.end(); base = short[n0:n0 + int(m.group(1))]; rest = short[n0 + int(m.group(1)):]
            tm = re.match(r"I((?:L[ib]\d+E)+)E", rest)
            targs = re.findall(r"L[ib](\d+)E", tm.group(1)) if tm else []
            short = base + ("<" + ", ".join(targs) + ">" if targs else "")
        lines.append("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %d | %d | %d | %d |" % (short, n, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot, wg, grid, scr, lds))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(out)
    print(out)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Builds profiles/r04_pmc.json from the output of tools/profile_round4.sh
(gpurun_out/prof_r04/{pmc_rows.jsonl, kernel_stats.md, bench_line.json}): counters at the launch shapes bench.py times by default."""
import collections
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "prof_r04")
rows = [json.loads(l) for l in open(os.path.join(src, "pmc_rows.jsonl"))]
bl = json.load(open(os.path.join(src, "bench_line.json")))
d, meta = collections.defaultdict(dict), {}
for r in rows:
    d[r["kernel"]][r["counter"]] = r["avg_per_launch"]
    meta[r["kernel"]] = r


def entry(k, units, unit_name, per_launch_units, fetch_x2, extra=None):
    c, m = d[k], meta[k]
    e = {"kernel": k, unit_name: units, "grid_lanes": m["grid"], "workgroup": m["workgroup"], "launches_averaged": m["launches"],
         "fetch_bytes_per_launch": c["FETCH_SIZE"] * 1024 * (2 if fetch_x2 else 1), "fetch_correction": "x2 (wide coalesced streams, MI355X_MICROARCH.md HBM section)" if fetch_x2 else "raw (gathers of 96/104-byte records and scratch: uncalibrated width, reported as counted)",
         "write_bytes_per_launch": c["WRITE_SIZE"] * 1024, "valu_insts_per_launch": c["SQ_INSTS_VALU"], "salu_insts_per_launch": c["SQ_INSTS_SALU"],
         "vmem_insts_per_launch": c["SQ_INSTS_VMEM"], "lds_insts_per_launch": c["SQ_INSTS_LDS"], "lds_idx_active": c["SQ_LDS_IDX_ACTIVE"], "lds_bank_conflict": c["SQ_LDS_BANK_CONFLICT"],
         "gui_active_cycles_per_launch_all_xcd": c["GRBM_GUI_ACTIVE"], "sq_wave_cycles": c["SQ_WAVE_CYCLES"], "sq_active_inst_any": c["SQ_ACTIVE_INST_ANY"],
         "sq_wait_inst_any": c["SQ_WAIT_INST_ANY"], "sq_wait_any": c["SQ_WAIT_ANY"], "sq_busy_cycles": c["SQ_BUSY_CYCLES"], "waves": c["SQ_WAVES"],
         "scratch_bytes_per_lane": m["scratch_bytes_per_lane"], "lds_bytes_per_workgroup_reported": m["lds"], "per_launch_units": per_launch_units}
    if extra:
        e.update(extra)
    return e


L = bl["roofline"]["secondary"]["fk20"]["launches_per_step"]
pm = {
    "source": "rocprofv3 --pmc <pass> --kernel-trace --output-format csv (tools/profile_round4.sh r04), 1x MI355X, passes FETCH_SIZE | WRITE_SIZE | SQ group 1 + GRBM | SQ group 2; "
              "launch shapes = bench.py's defaults (4096 blobs, 1024 polynomials, 1024 F_r transforms per launch); FETCH_SIZE / WRITE_SIZE in KiB as reported; SQ_INSTS_* count wave64 "
              "instructions; SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY + SQ_WAIT_ANY in quad-cycles summed over waves; GRBM_GUI_ACTIVE summed over the 8 XCDs",
    "k_fb_accumulate": entry("k_fb_accumulate", 4096, "batch", "4096 blobs of 4096 coefficients", False, {"n": 4096, "table_c": 16, "table_windows": 16, "vgprs": 256}),
    "k_g1_fft_stage": entry("k_g1_fft_stage", 1024, "batch", "one radix-2 stage of 1024 transforms of 4096 points (average over the DIF and DIT stage launches of a step)", False,
                            {"n": 4096, "launches_per_step": L, "vgprs": 256}),
    "k_fb_mul_vec_dif2": entry("k_fb_mul_vec", 1024, "batch", "Toeplitz stage fused with two DIF stages, 1024 polynomials", False, {"vgprs": 256}),
    "k_fr_fft4096_r4": entry("k_fr_fft4096_r4<false>", 1024, "batch", "1024 forward transforms of 4096 points", True, {"vgprs": 106, "lds_bytes_per_workgroup": 149760}),
    "k_fr_fft4096_r4_scaled": entry("k_fr_fft4096_r4<true>", 1024, "batch", "1024 inverse transforms (final scale by 1/n)", True, {"vgprs": 106, "lds_bytes_per_workgroup": 149760}),
    "k_das_ext2048_r4": entry("k_das_ext2048_r4", 1024, "batch", "1024 DAS extensions of 2048 values", True, {"vgprs": 128, "lds_bytes_per_workgroup": 76032}),
}
for k in ("k_g1_fft_stage",):
    e = pm[k]
    e["fetch_bytes_per_step"] = e["fetch_bytes_per_launch"] * L
    e["write_bytes_per_step"] = e["write_bytes_per_launch"] * L
    e["valu_insts_per_step"] = e["valu_insts_per_launch"] * L
json.dump(pm, open(os.path.join(R, "profiles", "r04_pmc.json"), "w"), indent=1)
print("ok", {k: v.get("fetch_bytes_per_launch") for k, v in pm.items() if isinstance(v, dict)})

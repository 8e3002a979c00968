#!/usr/bin/env python3
"""Builds profiles/r06_pmc.json from the output of tools/profile_round6.sh
(gpurun_out/prof_r06/{pmc_rows.jsonl, kernel_stats.md, bench_line.json}): counters at the launch shapes bench.py times by default."""
import collections
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "prof_r06")
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
    "source": "rocprofv3 --pmc <pass> --kernel-trace --output-format csv (tools/profile_round6.sh r06), 1x MI355X, passes FETCH_SIZE | WRITE_SIZE | SQ group 1 + GRBM | SQ group 2; "
              "launch shapes = bench.py's defaults (4096 blobs, 1024 polynomials, 1024 F_r transforms per launch); FETCH_SIZE / WRITE_SIZE in KiB as reported; SQ_INSTS_* count wave64 "
              "instructions; SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY + SQ_WAIT_ANY in quad-cycles summed over waves; GRBM_GUI_ACTIVE summed over the 8 XCDs",
    "k_fb_accumulate": entry("k_fb_accumulate", 4096, "batch", "4096 blobs of 4096 coefficients", False, {"n": 4096, "table_c": 16, "table_windows": 8, "additions_per_coefficient": 16, "walk": "k_fb_accumulate_glv: both GLV halves of a scalar walk the same 8 windows", "vgprs": 256}),
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
json.dump(pm, open(os.path.join(R, "profiles", "r06_pmc.json"), "w"), indent=1)
print("ok", {k: v.get("fetch_bytes_per_launch") for k, v in pm.items() if isinstance(v, dict)})


def kernel_stats(trace_dir):
    """profiles/r06_kernel_stats.md + r06_kernel_shapes.json from the trace step of tools/profile_round6.sh (gpurun_out/<trace_dir>)"""
    src_ = os.path.join(R, "gpurun_out", trace_dir)
    d_ = json.loads(open(os.path.join(src_, "bench_line.json")).read())
    t_ = json.loads(open(os.path.join(src_, "trace_bench.json")).read().strip().splitlines()[-1])
    rows_ = json.load(open(os.path.join(src_, "kernel_shapes.json")))["rows"]
    r_ = [x for x in rows_ if x["kernel"].startswith("k_fb_accumulate") and x["grid"] == 1048576 and x["workgroup"] == 256][0]
    avg = r_["avg_us"] / 1e3
    hdr = """# r06 -- kernel trace of `python bench.py --no-cpu-baseline --no-extras --no-in-process` at its default step sizes (rocprofv3 --kernel-trace --stats, tools/profile_round6.sh), 1x MI355X

One row per LAUNCH SHAPE (kernel, grid, workgroup) -- tools/rocprof_summary.py; `r06_kernel_shapes.json` holds the same rows for bench.py, which prints the
row of the launch it puts on its roofline line as `roofline.profile_avg_ms` beside its own HIP-event figure.

* headline: `k_fb_accumulate_glv<0>`, grid 1 048 576 = 4096 workgroups x 256 lanes (one workgroup per blob; the library-default table: signed 16-bit windows, 8 of them
  walked by both GLV halves of every scalar, 103 GB; `--no-extras` skips the table sweep, so every launch of this shape walks the headline's table): avg %.2f ms under the profiler (%d launches) -> 0.5377 GB / %.2f ms = %.1f GB/s = %.4f of 8 TB/s.
  (HIP events in the un-profiled run of the same build on the same box: %.2f ms per launch: the two agree.)
* FK20 (config 4a, 1024 polynomials per step): `k_g1_fft_stage<4, 1>` / `k_g1_fft_stage_dif<1>`, grid 2 097 152 (a lane per butterfly; the 512-polynomial step of
  8192-point transforms of the fk20_4096 block has the same grid and the same work per lane).
* the `k_msm_*` rows with thousands of calls are bench.py's self-check of EVERY output (one LinCombG1 per FK20 polynomial over its proofs as caller-supplied points).

Profiled line: %d commitments/s (%.2f ms per 4096-blob step), FK20 %d all-proofs/s, FK20 on 4096-element blobs %d/s.  Un-profiled line of the same build on the same box: %d commitments/s (%.2f ms per step), FK20 %d, FK20 on 4096-element blobs %d, FFT_Fr %.2f M/s, DAS extension %.2f M/s.

""" % (avg, r_["calls"], avg, 0.5377 / (avg * 1e-3), 0.5377 / (avg * 1e-3) / 8000, d_["roofline"]["avg_launch_ms"], t_["value"], t_["ms_per_step"], t_["fk20"]["value"],
       t_["fk20_4096"]["value"], d_["value"], d_["ms_per_step"], d_["fk20"]["value"], d_["fk20_4096"]["value"],
       d_["reference_benchmarks"]["fft_fr_scale12_per_s"]["value"] / 1e6, d_["reference_benchmarks"]["das_fft_extension_scale12_per_s"]["value"] / 1e6)
    open(os.path.join(R, "profiles", "r06_kernel_stats.md"), "w").write(hdr + open(os.path.join(src_, "kernel_stats.md")).read())
    json.dump({"source": "rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline --no-extras --no-in-process` (tools/profile_round6.sh); one row per (kernel, grid, workgroup)",
               "rows": rows_}, open(os.path.join(R, "profiles", "r06_kernel_shapes.json"), "w"), indent=0)
    print("kernel stats ok: k_fb_accumulate %.2f ms" % avg)


if len(sys.argv) > 2:
    kernel_stats(sys.argv[2])

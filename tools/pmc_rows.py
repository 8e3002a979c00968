#!/usr/bin/env python3
"""one rocprofv3 counter_collection.csv -> json lines {kernel, counter, launches, avg_per_launch, grid, ...}: per kernel family only the
launches with that family's LARGEST grid (the timed step of the command; smaller self-check / sweep launches are left out)"""
import csv, json, sys, collections
FAM = {"FB": [("k_fb_accumulate", "k_fb_accumulate")],
       "FK": [("k_g1_fft_stage", "k_g1_fft_stage"), ("k_fb_mul_vec", "k_fb_mul_vec")],
       "FR": [("k_fr_fft4096_r4ILb0", "k_fr_fft4096_r4<false>"), ("k_fr_fft4096_r4ILb1", "k_fr_fft4096_r4<true>"), ("k_fr_fft4096_r4<false>", "k_fr_fft4096_r4<false>"),
              ("k_fr_fft4096_r4<true>", "k_fr_fft4096_r4<true>"), ("k_das_ext2048_r4", "k_das_ext2048_r4")]}
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    for pat, fam in FAM[sys.argv[2]]:
        if pat in r["Kernel_Name"]:
            rows[(fam, r["Counter_Name"])].append(r)
            break
for (fam, ctr), rs in sorted(rows.items()):
    g = max(int(r["Grid_Size"]) for r in rs)
    sel = [r for r in rs if int(r["Grid_Size"]) == g]
    print(json.dumps({"kernel": fam, "counter": ctr, "launches": len(sel), "avg_per_launch": sum(float(r["Counter_Value"]) for r in sel) / len(sel), "grid": g,
                      "workgroup": int(sel[0]["Workgroup_Size"]), "scratch_bytes_per_lane": int(sel[0]["Scratch_Size"]), "vgprs": int(sel[0]["VGPR_Count"]),
                      "accum_vgprs": int(sel[0].get("Accum_VGPR_Count", 0) or 0), "sgprs": int(sel[0]["SGPR_Count"]), "lds": int(sel[0]["LDS_Block_Size"])}))

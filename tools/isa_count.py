#!/usr/bin/env python3
"""instruction mix of one kernel in an AMDGPU assembly file: whole kernel and its hottest loop (largest backward-branch body)
usage: hipcc ... -save-temps=obj -c k_msm.hip; tools/isa_count.py k_msm-hip-amdgcn-amd-amdhsa-gfx950.s 15k_fb_accumulate"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % sys.argv[2], l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
labels = {}
ins = []
for l in body:
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        labels[m.group(1)] = len(ins); continue
    t = l.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
    ins.append(t.split(";")[0].strip())
def mix(seq):
    c = collections.Counter()
    for i in seq:
        op = i.split()[0]
        if op.startswith("v_mad_u64_u32"): c["v_mad_u64_u32"] += 1
        elif op.startswith(("v_mul_lo_u32", "v_mul_hi_u32")): c["v_mul_lo/hi_u32"] += 1
        elif op.startswith("v_"): c["other VALU"] += 1
        elif op.startswith(("s_waitcnt", "s_nop")): c["s_waitcnt/nop"] += 1
        elif op.startswith("s_"): c["SALU/branch"] += 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): c["VMEM (" + ("scratch" if op.startswith("scratch") else "global") + ")"] += 1
        elif op.startswith("ds_"): c["LDS"] += 1
        else: c[op] += 1
    return c
print("kernel instructions:", len(ins), dict(mix(ins)))
# loops: backward branches (static instruction counts of the loop bodies, cold paths included)
loops = []
for idx, i in enumerate(ins):
    m = re.match(r"s_cbranch\w*\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", i)
    if m:
        lab = m.group(1) or m.group(2)
        if lab in labels and labels[lab] <= idx:
            loops.append((idx - labels[lab] + 1, labels[lab], idx))
for size, a, b in sorted(loops, reverse=True)[:6]:
    print("loop body [%d..%d]: %d instructions %s" % (a, b, size, dict(mix(ins[a:b + 1]))))

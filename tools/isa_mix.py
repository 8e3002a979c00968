#!/usr/bin/env python3
"""Instruction mix of the HOT straight-line blocks of a kernel, from `hipcc -save-temps=obj` assembly.
usage: tools/isa_mix.py file.s <kernel-name-substring> [min_block_instructions]   ->  markdown rows (one per block >= the minimum, default 400)
Categories (issue cost per wave64 instruction as measured in profiles/r02_valu_calibration.md: 'mul-rate' ~5 cycles, 'simple' ~2.5):
  mad64      v_mad_u64_u32 / v_mad_i64_i32                      (mul-rate)   -- the partial products
  mul32      v_mul_lo_u32 / v_mul_hi_u32 / v_mul_u32_u24 ...    (mul-rate)
  add64      v_lshl_add_u64, v_add_co/v_addc_co pairs' carry half (mul-rate) -- column carries
  shift64    v_lshrrev_b64 / v_lshlrev_b64 / v_ashrrev_i64      (mul-rate)
  addsub32   v_add_u32 / v_sub_u32 / v_add3 / v_add_co (low half) / v_subrev ... (simple)
  logic      v_and / v_or / v_xor / v_bfe / v_bfi / v_lshl / v_lshr / v_alignbit / v_and_or ... (simple)
  move       v_mov / v_accvgpr_* / v_readlane / v_readfirstlane / v_cndmask / v_swap (simple)
  cmp        v_cmp*                                              (simple)
  salu, wait (s_waitcnt / s_nop), lds, global, scratch, branch
"""
import re, sys, collections
MULRATE = {"mad64", "mul32", "add64", "shift64"}
def cat(op):
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")): return "mad64"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mul_u32", "v_mul_i32", "v_mad_u32", "v_mad_i32")): return "mul32"
    if op.startswith(("v_lshl_add_u64", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_add_u64", "v_sub_u64")): return "add64"
    if op.startswith(("v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64")): return "shift64"
    if op.startswith(("v_add", "v_sub")): return "addsub32"
    if op.startswith(("v_and", "v_or", "v_xor", "v_bfe", "v_bfi", "v_lshl", "v_lshr", "v_ashr", "v_alignbit", "v_not", "v_perm", "v_bfm")): return "logic"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_cndmask", "v_swap", "v_permlane", "v_pk_mov")): return "move"
    if op.startswith("v_cmp"): return "cmp"
    if op.startswith("v_"): return "valu_other"
    if op.startswith(("s_waitcnt", "s_nop")): return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_swappc", "s_setpc", "s_call", "s_getpc")): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")): return "global"
    return "other"
def blocks_of(path, sub):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and sub in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    out, cur = [], [lines[start].rstrip(":"), []]
    for l in lines[start + 1:end + 1]:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            out.append(cur); cur = [m.group(1), []]; continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
        cur[1].append(t.split(";")[0].strip().split()[0])
    out.append(cur)
    return out
if __name__ == "__main__":
    path, sub = sys.argv[1], sys.argv[2]
    minb = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    cats = ["mad64", "mul32", "add64", "shift64", "addsub32", "logic", "move", "cmp", "valu_other", "salu", "wait", "lds", "global", "scratch", "branch"]
    print("| kernel / block | instr | " + " | ".join(cats) + " | mad share of VALU | mul-rate share of VALU issue time |")
    print("|---|---|" + "---|" * (len(cats) + 2))
    for name, ops in blocks_of(path, sub):
        if len(ops) < minb: continue
        c = collections.Counter(cat(o) for o in ops)
        valu = sum(c[k] for k in ("mad64", "mul32", "add64", "shift64", "addsub32", "logic", "move", "cmp", "valu_other"))
        t_mul = sum(c[k] for k in MULRATE) * 5.0
        t_all = t_mul + (valu - sum(c[k] for k in MULRATE)) * 2.5
        print("| %s %s | %d | " % (sub, name if name.startswith(".") else "(entry)", len(ops)) + " | ".join(str(c[k]) for k in cats) +
              " | %.3f | %.3f |" % (c["mad64"] / max(valu, 1), t_mul / max(t_all, 1)))

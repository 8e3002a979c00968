"""Roofline arithmetic of the bench line, as pure functions (no GPU): what `roofline`, `roofline.mac` and `roofline.issue` are made of.

SURVEY.md 8(d): a commitment moves 131 072 B of scalars + 96 B of result with the setup (393 216 B) counted once per launch; the contract's `achieved` is those
ALGORITHMIC bytes over the dominant kernel's average launch time.  What bounds the walk is integer issue, so the line also carries the multiply-add rate against the
`v_mad_u64_u32` rate measured in the same run, and the share of the launch that pure instruction issue explains (counters of the committed PMC passes)."""
from .workload import BYTES_PER_COMMIT, BYTES_SETUP, HBM_PEAK_GBS, N_COEFF

# one XYZZ mixed addition on lazy 30-bit limbs: 6 products (338 v_mad_u64_u32 each), 2 squarings (260), one two-product reduction (507)
MADS_PER_MIXED_ADDITION = 6 * 338 + 2 * 260 + 507


def pmc_for_walk(pmc, kernel, n, tab_c, tab_w, batch):
    """the counter entry of profiles/rNN_pmc.json if it describes THIS workload (kernel, length, table shape, a batch the launch is a multiple of); else None.
    Returns (entry, scale): the counters were taken at entry['batch'] blobs per launch and the work per blob does not depend on the batch from 512 blobs on."""
    pm = pmc.get("k_fb_accumulate", pmc) if isinstance(pmc, dict) else None
    try:
        if (pm["kernel"] == kernel and pm["n"] == n and pm["table_c"] == tab_c and pm.get("table_windows", tab_w) == tab_w
                and batch >= pm["batch"] and batch % pm["batch"] == 0):
            return pm, batch / pm["batch"]
    except (KeyError, TypeError):
        pass
    return None, 1.0


def walk_roofline(kernel, batch, avg_s, table, pmc=None, pmc_file=None):
    """`roofline` of the commitment step: batch blobs per launch, avg_s = average launch time of the dominant kernel (HIP events), table = (window bits, windows, bytes)"""
    tab_c, tab_w, tab_bytes = table
    alg_bytes = batch * BYTES_PER_COMMIT + BYTES_SETUP
    ach = alg_bytes / avg_s * 1e-9
    pm, sc = pmc_for_walk(pmc, kernel, N_COEFF, tab_c, tab_w, batch) if pmc else (None, 1.0)
    traffic = (pm["fetch_bytes_per_launch"] + pm["write_bytes_per_launch"]) * sc if pm else None
    src = None
    if pm:
        src = pmc_file if sc == 1.0 else "%s (counters of the %d-blob launch x %g)" % (pmc_file, pm["batch"], sc)
    return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "avg_launch_ms": avg_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes, "table": {"window_bits": tab_c, "windows": tab_w, "GB": tab_bytes / 1e9},
            "secondary": {}, "traffic_source": src,
            "note": "integer-issue-bound kernel (see mac / issue); traffic (PMC passes committed under profiles/) exceeds the algorithmic bytes by design: "
                    "fixed-base table gathers trade HBM bandwidth for integer work (DESIGN.md 4)"}, pm, sc


def walk_mac(batch, additions_per_coefficient, avg_s, cal_mad, cal_add, cal_fpmul):
    """`roofline.mac`: multiply-adds per launch (batch x n x additions per coefficient mixed additions; zero digits: < 2^-15) against the measured v_mad_u64_u32 rate"""
    mads = batch * N_COEFF * additions_per_coefficient * MADS_PER_MIXED_ADDITION
    return {"mads_per_launch": mads, "achieved_Tmad_s": mads / avg_s * 1e-12, "measured_peak_Tmad_s": cal_mad * 1e-12, "frac": mads / avg_s / cal_mad,
            "measured_v_add_u32_Tops_s": cal_add * 1e-12, "measured_fp_products_G_s": cal_fpmul * 1e-9, "fp_product_equivalents_G_s": mads / 338.0 / avg_s * 1e-9,
            "note": "peak = kzg_hip_calibrate on this GPU in this run (tools/microbench.hip loops); the guide's SIMD-32 figure (one wave64 VALU instruction per 2 cycles) "
                    "holds for v_add_u32 / v_mov, the 64-bit multiply-add issues at ~5.3 cycles"}


def issue_model(mads, valu_wave_insts, avg_s, cal_mad, cal_add):
    """`roofline.issue`: multiply-adds at the measured multiply-add rate + the other VALU instructions (wave instructions x 64 lanes) at the measured v_add_u32 rate:
    the share of the launch that pure instruction issue of this mix explains; the rest is dependency / memory stalls"""
    other = valu_wave_insts * 64.0 - mads
    model_s = mads / cal_mad + max(other, 0.0) / cal_add
    return {"valu_wave_insts_per_launch": valu_wave_insts, "mad_share_of_insts": mads / 64.0 / valu_wave_insts, "issue_model_ms": model_s * 1e3,
            "frac_of_launch_explained": model_s / avg_s,
            "note": "mads / measured mad rate + other VALU / measured add rate; the rest is dependency / memory stalls at 2 waves per SIMD"}

"""The multi-device handle of the C ABI (kzg_hip_multi_*) timed in a child process once the parent has released its tables."""
import hashlib
import json
import subprocess
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from .workload import *  # noqa: F401,F403
from .workload import splitmix_blobs_le32, S_TEST, R_MOD, N_COEFF


def run_in_process_child(devices, timeout=300):
    """the multi-device leg in a child process (a crash or a hang there must not cost the bench line): returns its `in_process` object"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--in-process-child", "--devices", ",".join(str(d) for d in devices)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK")}
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        for line in reversed(res.stdout.splitlines()):
            if line.startswith('{"in_process"'):
                return json.loads(line)["in_process"]
        return {"error": "child exit code %d, no result line" % res.returncode, "stderr_tail": res.stderr[-600:]}
    except subprocess.TimeoutExpired:
        return {"error": "child timed out after %d s" % timeout}
    except Exception as e:                                    # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}


def in_process_child(devices):
    """Times the multi-device handle of the C ABI (kzg_hip_multi_*: what a Go caller gets from NewMultiKZGSettings) on `devices`: host-buffer
    batches divided among the devices (PCIe-inclusive, the only form a Go slice can take) and ONE polynomial sharded inside the library.
    On a single device the one-polynomial legs also run on the list [d, d] (two entries, one GPU): that measures the orchestration and the
    exchange, not a speed-up.  Prints one line {"in_process": {...}}."""
    import gokzg_amd as kz
    golden = os.path.join(ROOT, "tests", "golden")
    pins = json.load(open(os.path.join(golden, "fk20_pins.json")))
    out = {"devices": devices, "entry": "kzg_hip_multi_* (include/kzg_hip.h), host buffers, blocking calls"}

    def med(fn, reps=5, warm=2):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))

    try:
        D = len(devices)
        fs0 = kz.FFTSettings(12, device=devices[0])
        raw = np.frombuffer(open(os.path.join(golden, "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
        setup = fs0.from_compressed_g1(raw)

        def mont(seed, batch, n=N_COEFF):
            o, ok = fs0.fr_from_32(splitmix_blobs_le32(seed, batch, n).reshape(-1, 32))
            assert ok
            return o.reshape(batch, n, 4)

        def sha(fsx, pts):
            return hashlib.sha256(fsx.to_compressed_g1(pts).tobytes()).hexdigest()
        m = kz.MultiKZGSettings(devices, 12, setup)
        out["transport"], out["transport_note"], out["transport_self_test"] = m.transport, m.transport_note, m.transport_self_test
        per = 1024
        blobs = mont(1, per * D)
        exp_f = json.load(open(os.path.join(golden, "derived_vectors.json")))["F_blob_seed1"]["commit_monomial_s1337"]
        got = m.commit_to_poly_batch(blobs)
        ks0 = m.kzg_settings(0)
        one_dev = ks0.commit_to_poly_batch(blobs[:per])
        ms_all = med(lambda: m.commit_to_poly_batch(blobs), 3, 1)
        ms_one = med(lambda: ks0.commit_to_poly_batch(blobs[:per]), 3, 1)
        with kz.pinned(blobs):                                  # kzg_hip_host_register: every device reads its share of the range in place over PCIe
            ms_pin = med(lambda: m.commit_to_poly_batch(blobs), 3, 1)
            pin_same = bool(np.array_equal(m.commit_to_poly_batch(blobs), got))
        out["commit_to_poly_batch"] = {"blobs_per_device": per, "table": "library default (budget 110 GB: 16-bit windows x 8, 103 GB) on every device",
                                       "commitments_per_s": per * D / ms_all * 1e3, "one_device_same_call_per_s": per / ms_one * 1e3,
                                       "commitments_per_s_pinned_input": per * D / ms_pin * 1e3, "pinned_same_results": pin_same,
                                       "scaling_vs_one_device": (per * D / ms_all) / (per / ms_one),
                                       "vector_F": fs0.to_compressed_g1(got[:1])[0].tobytes().hex() == exp_f, "first_share_equals_one_device": bool(np.array_equal(got[:per], one_dev))}
        # FK20 (config 4a): batches divided among the devices, and ONE polynomial sharded inside the library
        mfk = kz.MultiFK20SingleSettings(m, 4096)
        fper = 32
        polys = mont(4, fper * D)[:, :2048, :].copy()
        pr = mfk.da_using_fk20_batch(polys)
        ms_fk = med(lambda: mfk.da_using_fk20_batch(polys), 2, 1)
        fk0 = kz.FK20SingleSettings(ks0, 4096)
        ms_fk1 = med(lambda: fk0.da_using_fk20_batch(polys[:fper]), 2, 1)
        out["da_using_fk20_batch"] = {"polynomials_per_device": fper, "all_proofs_per_s": fper * D / ms_fk * 1e3, "one_device_same_call_per_s": fper / ms_fk1 * 1e3,
                                      "byte_pin_row0": sha(fs0, pr[0]) == pins["config4a_da_using_fk20_seed4"]["sha256"]}
        one = {"unsharded_one_device_ms": med(lambda: fk0.da_using_fk20(polys[0]))}
        for mode in ("gather", "sharded"):
            if D == 1 and mode == "sharded":
                continue
            m.set_fft_sharding(mode)
            e0 = m.exchanges
            okp = sha(fs0, mfk.da_using_fk20(polys[0])) == pins["config4a_da_using_fk20_seed4"]["sha256"]
            one[mode] = {"ms": med(lambda: mfk.da_using_fk20(polys[0])), "all_gathers_per_call": None, "byte_pin": okp}
            e1 = m.exchanges
            mfk.da_using_fk20(polys[0])
            one[mode]["all_gathers_per_call"] = m.exchanges - e1
        out["da_using_fk20_one_polynomial"] = one
        fk0.close(); mfk.close(); m.close()

        # config 5: ONE DAUsingFK20Multi (scale 16, chunk 16) over the devices; on a single device also over two entries of it
        fs16 = kz.FFTSettings(16, device=devices[0])
        sec = np.frombuffer((S_TEST * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little"), dtype=np.uint64).reshape(1, 4)
        setup16 = fs16.generate_testing_setup_g1(sec, 65536)
        poly5 = mont(5, 1, 32768)[0]
        pin5 = pins["config5_da_using_fk20_multi_seed5"]["sha256"]
        cfg5 = {}
        for devs in ([devices] if D > 1 else [devices, devices * 2]):
            m16 = kz.MultiKZGSettings(devs, 16, setup16)
            mfkm = kz.MultiFK20MultiSettings(m16, 65536, 16)
            row = {"devices": devs, "transport": m16.transport}
            if "unsharded_one_device_ms" not in cfg5:
                fkm0 = kz.FK20MultiSettings(m16.kzg_settings(0), 65536, 16)
                cfg5["unsharded_one_device_ms"] = med(lambda: fkm0.da_using_fk20_multi(poly5), 5, 4)   # (the first calls after building 47 GB of tables run at a lower clock)
                fkm0.close()
            for mode in ("gather", "sharded"):
                if len(devs) == 1 and mode == "sharded":
                    continue
                m16.set_fft_sharding(mode)
                okp = sha(fs16, mfkm.da_using_fk20_multi(poly5)) == pin5
                e1 = m16.exchanges
                row[mode] = {"ms": med(lambda: mfkm.da_using_fk20_multi(poly5), 5, 2), "byte_pin": okp}
                row[mode]["all_gathers_per_call"] = (m16.exchanges - e1) // 7
            cfg5["%d_entries" % len(devs)] = row
            mfkm.close(); m16.close()
        cfg5["note"] = ("entries of ONE device share its SIMDs: the figures there are orchestration + exchange cost, not a speed-up" if D == 1 else
                        "Toeplitz stage by output position on every device; gather = transforms on the first device, sharded = five all-gathers")
        out["da_using_fk20_multi_one_polynomial_scale16"] = cfg5
        fs16.close(); fs0.close()
    except Exception as e:                                    # noqa: BLE001
        import traceback
        out["error"] = "%s: %s | %s" % (type(e).__name__, e, traceback.format_exc().strip().splitlines()[-3:])
    print(json.dumps({"in_process": out}))
    return 0

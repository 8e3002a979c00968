#!/usr/bin/env python3
"""First run on a multi-GPU node: what the multi-device handle (kzg_hip_multi_*) finds and does there, in one report.

    python tools/multi_bringup.py [device ordinals, comma separated; default: every gfx950 device]

For each exchange transport (the default choice, then peer-copy and host-staged forced): the transport the constructor PROVED (its creation-time self-test), why an
earlier one was not used, the config-5 byte pin (DAUsingFK20Multi, scale 16, chunk 16: 4096 coset proofs of ONE polynomial sharded over the devices) in both exchange
schemes with its latency, and the batch rates (commitments from host buffers, rows divided among the devices) beside one device's.  Nothing here has ever run on more than
one device (rounds 1-5 had no multi-GPU node): this script is the reproducible form of "run it once and look", tests/test_multi_device.py::test_distinct_devices_byte_pins
the asserting form.  Exit code 0 = every pin matched."""
import hashlib, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(devs, force):
    import bench
    import gokzg_amd as kz
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "fk20_pins.json")))
    out = {"force": force or "default"}
    fs = kz.FFTSettings(16, device=devs[0])
    sec = np.frombuffer((bench.S_TEST * ((1 << 256) % bench.R_MOD) % bench.R_MOD).to_bytes(32, "little"), dtype=np.uint64).reshape(1, 4)
    setup = fs.generate_testing_setup_g1(sec, 65536)
    t0 = time.perf_counter()
    m = kz.MultiKZGSettings(devs, 16, setup)
    out.update(constructor_s=time.perf_counter() - t0, transport=m.transport, transport_note=m.transport_note, self_test=m.transport_self_test)
    fk = kz.MultiFK20MultiSettings(m, 65536, 16)
    poly, _ = fs.fr_from_32(bench.splitmix_blobs_le32(5, 1, 32768).reshape(-1, 32))
    ok = True
    for mode in ("gather", "sharded"):
        if mode == "sharded" and (len(devs) & (len(devs) - 1)):
            continue
        m.set_fft_sharding(mode)
        proofs = fk.da_using_fk20_multi(poly)
        e0 = m.exchanges
        t0 = time.perf_counter()
        for _ in range(3):
            proofs = fk.da_using_fk20_multi(poly)
        pin = hashlib.sha256(fs.to_compressed_g1(proofs).tobytes()).hexdigest() == pins["config5_da_using_fk20_multi_seed5"]["sha256"]
        ok &= pin
        out[mode] = {"ms": (time.perf_counter() - t0) / 3 * 1e3, "all_gathers_per_call": (m.exchanges - e0) // 3, "byte_pin": pin}
    fk.close(); m.close(); fs.close()
    out["ok"] = ok
    print(json.dumps(out))
    return 0 if ok else 1


def main():
    import gokzg_amd as kz
    devs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(kz.device_count()))
    if len(devs) < 2:
        print("fewer than two devices listed / visible: nothing to bring up (the one-GPU suite covers lists that repeat a device)")
    rc = 0
    for force in (None, "peer", "host"):
        env = dict(os.environ, KZG_HIP_FK20_FB_BUDGET_GB=os.environ.get("KZG_HIP_FK20_FB_BUDGET_GB", "24"), KZG_HIP_FB_BUDGET_GB=os.environ.get("KZG_HIP_FB_BUDGET_GB", "10"))
        if force:
            env["KZG_HIP_MULTI_TRANSPORT"] = force
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", ",".join(map(str, devs)), force or ""], env=env, capture_output=True, text=True, timeout=1200)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")]
        print("== transport request: %s" % (force or "default (rccl between distinct devices)"))
        print(line[-1] if line else "child failed (exit code %d): %s" % (res.returncode, res.stderr[-800:]))
        rc |= res.returncode
    return rc


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        sys.exit(child([int(x) for x in sys.argv[2].split(",")], sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else None))
    sys.exit(main())

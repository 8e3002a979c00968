#!/usr/bin/env python3
"""Extracts the DATA the reference's own tests / data files hold for the hot path into small fixtures.

Run in the development container only (it reads /root/reference, which does not exist on the GPU box):
    python tests/golden/make_fixtures.py
Outputs (committed):
    reference_kats.json              known-answer constants of the reference's tests (decimals / bytes)
    trusted_setup_g1.bin             eth/trusted_setup.json "setup_G1"            4096 x 48 B (ZCash compressed)
    trusted_setup_g1_lagrange.bin    eth/trusted_setup.json "setup_G1_lagrange"   4096 x 48 B, natural order
    trusted_setup_sha256.json        SHA-256 pins of the arrays above + roots_of_unity
Only data values are taken (numbers, byte strings); no reference source text is copied.
"""
import hashlib
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def decimals_between(path, start_pat, end_pat):
    src = open(os.path.join(REF, path)).read()
    a = src.index(start_pat)
    b = src.index(end_pat, a)
    return re.findall(r'ToFr\("(\d+)"\)', src[a:b])


def main():
    kats = {}
    # bls/globals.go:27-60 -- the 32-entry 2^k-th root of unity table
    kats["scale2_root_of_unity"] = {
        "source": "bls/globals.go:27-60",
        "values": decimals_between("bls/globals.go", "Scale2RootOfUnity = []Fr{", "AsFr(&ZERO, 0)"),
    }
    assert len(kats["scale2_root_of_unity"]["values"]) == 32
    kats["modulus"] = {"source": "bls/globals.go:9",
                       "value": re.search(r'ModulusStr = "(\d+)"', open(os.path.join(REF, "bls/globals.go")).read()).group(1)}
    # fft_fr_test.go:32-71 TestInvFFT: FFT(inv=true) of 0..15 at scale 4
    kats["test_inv_fft"] = {
        "source": "fft_fr_test.go:32-71", "scale": 4, "input": list(range(16)),
        "expected": decimals_between("fft_fr_test.go", "func TestInvFFT", "func TestEvaluatePolyInEvaluationForm"),
    }
    assert len(kats["test_inv_fft"]["expected"]) == 16
    # das_extension_test.go:11-40 TestDASFFTExtension: even data 0..7 at scale 4
    kats["test_das_fft_extension"] = {
        "source": "das_extension_test.go:11-40", "scale": 4, "input": list(range(8)),
        "expected": decimals_between("das_extension_test.go", "func TestDASFFTExtension", "func TestParametrizedDASFFTExtension"),
    }
    assert len(kats["test_das_fft_extension"]["expected"]) == 8
    # bls/bls_test.go:11-23 TestPointCompression
    src = open(os.path.join(REF, "bls/bls_test.go")).read()
    a = src.index("func TestPointCompression")
    b = src.index("func TestPointG1Marshalling")
    scalar = re.search(r'SetFr\(&x, "(\d+)"\)', src[a:b]).group(1)
    by = re.search(r"expected := \[\]byte\{([^}]*)\}", src[a:b]).group(1)
    kats["test_point_compression"] = {"source": "bls/bls_test.go:11-23", "scalar": scalar,
                                      "expected_bytes": [int(v) for v in by.split(",")]}
    assert len(kats["test_point_compression"]["expected_bytes"]) == 48
    # zero_poly_test.go:133-198 TestFFTSettings_ZeroPolyViaMultiplication_Python (16 + 16 decimals, scale 4)
    zsrc = open(os.path.join(REF, "zero_poly_test.go")).read()
    za = zsrc.index("func TestFFTSettings_ZeroPolyViaMultiplication_Python")
    zb = zsrc.index("func testZeroPoly")
    exists = re.search(r"exists := \[\]bool\{([^}]*)\}", zsrc[za:zb]).group(1)
    allz = re.findall(r'bls\.ToFr\("(\d+)"\)', zsrc[za:zb])
    kats["test_zero_poly_python"] = {"source": "zero_poly_test.go:133-198", "scale": 4,
                                     "exists": [v.strip() == "true" for v in exists.split(",") if v.strip()],
                                     "expected_eval": allz[:16], "expected_poly": allz[16:32]}
    assert len(kats["test_zero_poly_python"]["exists"]) == 16 and len(allz) == 32
    # test secret / polynomial used all over the reference's tests
    kats["test_secret"] = {"source": "kzg_single_proofs_test.go:13", "value": "1927409816240961209460912649124"}
    kats["test_poly"] = {"source": "kzg_single_proofs_test.go:15", "values": [1, 2, 3, 4, 7, 7, 7, 7, 13, 13, 13, 13, 13, 13, 13, 13]}
    kats["g1_generator"] = {"source": "bls/bls_hbls.go:23-24"}
    gsrc = open(os.path.join(REF, "bls/bls_hbls.go")).read()
    kats["g1_generator"]["x"] = re.search(r'GenG1\.X\.SetString\("(\d+)"', gsrc).group(1)
    kats["g1_generator"]["y"] = re.search(r'GenG1\.Y\.SetString\("(\d+)"', gsrc).group(1)
    json.dump(kats, open(os.path.join(HERE, "reference_kats.json"), "w"), indent=1)

    ts = json.load(open(os.path.join(REF, "eth/trusted_setup.json")))
    pins = {"source": "eth/trusted_setup.json",
            "file_sha256": hashlib.sha256(open(os.path.join(REF, "eth/trusted_setup.json"), "rb").read()).hexdigest()}
    g1 = b"".join(bytes.fromhex(h) for h in ts["setup_G1"])
    lag = b"".join(bytes.fromhex(h) for h in ts["setup_G1_lagrange"])
    g2 = b"".join(bytes.fromhex(h) for h in ts["setup_G2"])
    roots = b"".join(int(v).to_bytes(32, "little") for v in ts["roots_of_unity"])
    assert len(g1) == 4096 * 48 and len(lag) == 4096 * 48 and len(roots) == 4096 * 32
    open(os.path.join(HERE, "trusted_setup_g1.bin"), "wb").write(g1)
    open(os.path.join(HERE, "trusted_setup_g1_lagrange.bin"), "wb").write(lag)
    pins["setup_G1"] = hashlib.sha256(g1).hexdigest()
    pins["setup_G1_lagrange"] = hashlib.sha256(lag).hexdigest()
    pins["setup_G2"] = hashlib.sha256(g2).hexdigest()
    pins["roots_of_unity_le32"] = hashlib.sha256(roots).hexdigest()
    json.dump(pins, open(os.path.join(HERE, "trusted_setup_sha256.json"), "w"), indent=1)
    print(json.dumps(pins, indent=1))


if __name__ == "__main__":
    main()

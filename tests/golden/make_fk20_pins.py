#!/usr/bin/env python3
"""Byte pins for the full-size FK20 configurations (BASELINE configs 4a, 4b, 5): runs the CPU oracle
(oracle/kzg_oracle.c, the restatement of kzg.go:43-116, fk20_single.go:122-196, fk20_multi.go:58-133) ONCE in the
build container and records the SHA-256 of ALL compressed proofs in tests/golden/fk20_pins.json.  The GPU parity tests
compare the full hash (tests/test_gpu_parity.py), so a permutation / offset error confined to positions that the
sampled coset identity does not visit cannot hide.

    python tests/golden/make_fk20_pins.py            # ~10 minutes of single-core oracle time (scale-16 settings dominate)
    python tests/golden/make_fk20_pins.py --only-l128   # just config 5's second variant (chunk length 128), merged into the file

Inputs are the synthetic blobs of SURVEY.md 8(d) (splitmix64, seed = config seed); setups are GenerateTestingSetup with
the reference's test secret (kzg_single_proofs_test.go) except config 4a, which uses the s = 1337 monomial setup of
eth/trusted_setup.json (tests/golden/trusted_setup_g1.bin).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import koracle as ko   # noqa: E402

S_TEST = 1927409816240961209460912649124


def digest(points):
    c = ko.g1_compress(points)
    return {"count": int(c.shape[0]), "sha256": hashlib.sha256(c.tobytes()).hexdigest(),
            "first": c[0].tobytes().hex(), "last": c[-1].tobytes().hex()}


def config5_l128(out, ks16, t0):
    # config 5, second variant ("also l = 128 (integration_test.go:74)"): the same 32768 coefficients of seed 5,
    # chunk length 128 -> k = 256, 512 coset proofs
    fkm128 = ko.FK20MultiSettings(ks16, 65536, 128)
    print("scale-16 l=128 settings done %.0f s" % (time.time() - t0), flush=True)
    out["config5_l128_da_using_fk20_multi_seed5"] = digest(fkm128.da_using_fk20_multi(ko.synthetic_blob(5, 32768)))
    print("5 (l=128) done %.0f s" % (time.time() - t0), flush=True)


def main():
    if "--only-l128" in sys.argv:
        path = os.path.join(HERE, "fk20_pins.json")
        out = json.load(open(path))
        t0 = time.time()
        ks16 = ko.KZGSettings(ko.FFTSettings(16), ko.generate_testing_setup_g1(S_TEST, 65536))
        config5_l128(out, ks16, t0)
        out["oracle_seconds_l128"] = round(time.time() - t0)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        print(json.dumps(out["config5_l128_da_using_fk20_multi_seed5"], indent=1))
        return
    out = {"note": "SHA-256 over the concatenated 48-byte compressed proofs in returned order; produced by the CPU oracle "
                   "(tests/golden/make_fk20_pins.py), not by running the reference (no Go toolchain in the image)"}
    t0 = time.time()

    # config 4a: DAUsingFK20, scale 12, poly = blob(seed 4)[:2048], monomial setup s = 1337 -> 4096 proofs
    raw = np.frombuffer(open(os.path.join(HERE, "trusted_setup_g1.bin"), "rb").read(), dtype=np.uint8)
    setup = ko.g1_decompress(raw)
    fs = ko.FFTSettings(12)
    ks = ko.KZGSettings(fs, setup)
    fk = ko.FK20SingleSettings(ks, 4096)
    out["config4a_da_using_fk20_seed4"] = digest(fk.da_using_fk20(ko.synthetic_blob(4)[:2048]))
    print("4a done %.0f s" % (time.time() - t0), flush=True)

    # config 4b: FK20Single on the full 4096-coefficient blob(seed 1): scale 13, 8192-point setup from S_test;
    # also the DA form on the same settings (4096 coefficients -> 8192 proofs)
    setup13 = ko.generate_testing_setup_g1(S_TEST, 8192)
    fs13 = ko.FFTSettings(13)
    ks13 = ko.KZGSettings(fs13, setup13)
    fk13 = ko.FK20SingleSettings(ks13, 8192)
    blob1 = ko.synthetic_blob(1)
    out["config4b_fk20_single_seed1"] = digest(fk13.fk20_single(blob1))
    out["config4b_da_using_fk20_seed1"] = digest(fk13.da_using_fk20(blob1))
    print("4b done %.0f s" % (time.time() - t0), flush=True)

    # config 5: DAUsingFK20Multi, scale 16 (n2 = 65536, 32768 coefficients from seed 5), chunk length 16 -> 4096 coset proofs
    setup16 = ko.generate_testing_setup_g1(S_TEST, 65536)
    fs16 = ko.FFTSettings(16)
    ks16 = ko.KZGSettings(fs16, setup16)
    fkm = ko.FK20MultiSettings(ks16, 65536, 16)
    print("scale-16 settings done %.0f s" % (time.time() - t0), flush=True)
    out["config5_da_using_fk20_multi_seed5"] = digest(fkm.da_using_fk20_multi(ko.synthetic_blob(5, 32768)))
    print("5 done %.0f s" % (time.time() - t0), flush=True)
    config5_l128(out, ks16, t0)
    out["oracle_seconds"] = round(time.time() - t0)
    with open(os.path.join(HERE, "fk20_pins.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

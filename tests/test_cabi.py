"""The C-ABI library loads on a machine without a GPU, exports every symbol include/kzg_hip.h declares, and
refuses to run without a device (no CPU fallback).  CPU only; no compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(path=os.path.join(ROOT, "include", "kzg_hip.h")):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kzg_hip_[a-z0-9_]+)\s*\(", src)))


INTERNAL_H = os.path.join(ROOT, "go-kzg_amd", "csrc", "kzg_hip_internal.h")


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("kzg_hip_fft_fr", "kzg_hip_fft_g1", "kzg_hip_das_fft_extension", "kzg_hip_lincomb_g1", "kzg_hip_commit_to_poly",
                 "kzg_hip_compute_proof_single", "kzg_hip_da_using_fk20", "kzg_hip_da_using_fk20_multi", "kzg_hip_fk20_single",
                 "kzg_hip_fk20_multi", "kzg_hip_toeplitz_part2", "kzg_hip_toeplitz_part3"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    import gokzg_amd
    lib = ctypes.CDLL(gokzg_amd.LIB_PATH)
    missing = [s for s in declared_symbols() + declared_symbols(INTERNAL_H) if not hasattr(lib, s)]
    assert not missing, missing
    # the boundary header carries no bench / test / calibration hook: those live in go-kzg_amd/csrc/kzg_hip_internal.h
    assert not [s for s in declared_symbols() if re.search(r"_(bench|test|prof|calibrate)(_|$)", s)]
    # ... and nothing is exported that neither header declares
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", gokzg_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and not l.split()[-1].startswith(("_init", "_fini")))
    assert set(exported) == set(declared_symbols() + declared_symbols(INTERNAL_H)), sorted(set(exported) ^ set(declared_symbols() + declared_symbols(INTERNAL_H)))
    gokzg_amd.lib()   # the binding's own signature table must resolve too


def test_no_cpu_fallback():
    import gokzg_amd
    if gokzg_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(gokzg_amd.NoDeviceError):
        gokzg_amd.FFTSettings(4)


def test_product_does_not_touch_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "go-kzg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".go")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "koracle" not in txt and "kzg_oracle" not in txt and "pyref" not in txt, os.path.join(dirpath, f)
    assert "oracle" not in open(os.path.join(ROOT, "include", "kzg_hip.h")).read()


def test_only_arena_memory_crosses_devices():
    """capi_multi.hip: what ncclAllGather / a peer copy / the host-staged exchange reads or writes is an `xbuf`, and the only place that makes
    one is exch_arena::take (hipMalloc memory with peer access granted) -- no hipMallocAsync pointer can reach the exchange"""
    src = open(os.path.join(ROOT, "go-kzg_amd", "csrc", "capi_multi.hip")).read()
    code = re.sub(r"//[^\n]*", "", src)
    assert len(re.findall(r"(?:\.|->)p\s*=[^=]", code)) == 1 and "out->p = base + at" in code            # one producer of xbuf pointers: the arena
    arena = code[code.index("struct exch_arena"):code.index("struct dev_worker")]
    assert "hipMalloc((void **)&base" in arena and "hipMallocAsync" not in arena
    ag = code[code.index("int all_gather_bytes(kzg_hip_multi *m, const std::vector<xbuf> &buf"):code.index("int transport_probe(")]
    assert "AllGather(buf[i].p + i * bytes_each, buf[i].p," in ag and "hipMemcpyPeerAsync(buf[j].p" in ag
    for call in re.findall(r"all_gather_bytes\(m, (\w+),", code):                                      # every call site passes a vector of xbuf
        assert re.search(r"std::vector<xbuf> [^;]*\b%s\b" % call, code), call
    assert "hipDeviceEnablePeerAccess" in code and "transport_self_test(m)" in code
    assert "std::thread" not in code[code.index("template <class F> int per_device"):code.index("struct mtmp")]   # no thread creation per call


def test_transcript_sha256_matches_hashlib_on_both_code_paths():
    """the SHA-256 behind eth.ComputeAggregateKZGProof's Fiat-Shamir transcript (go-kzg_amd/csrc/sha256.cpp): x86 SHA extensions where the CPU
    has them and the portable loop (forced with KZG_HIP_SHA256=portable in a child process), every length around the padding boundaries"""
    import subprocess
    import sys
    prog = (
        "import ctypes, hashlib, random, sys\n"
        "sys.path.insert(0, %r)\n"
        "import gokzg_amd\n"
        "L = ctypes.CDLL(gokzg_amd.LIB_PATH)\n"
        "L.kzg_hip_test_sha256.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p]\n"
        "L.kzg_hip_test_sha256.restype = None\n"
        "random.seed(7)\n"
        "for n in list(range(0, 260)) + [4096, 131072 + 16 + 16, 3 * 131072 + 177]:\n"
        "    d = random.randbytes(n)\n"
        "    out = ctypes.create_string_buffer(32)\n"
        "    L.kzg_hip_test_sha256(d, n, out)\n"
        "    assert out.raw == hashlib.sha256(d).digest(), n\n"
        "print('ok')\n") % ROOT
    for mode in ("", "portable"):
        env = dict(os.environ, KZG_HIP_SHA256=mode)
        assert subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=120).stdout.strip() == "ok", mode


# ---- a compiled C consumer of the boundary, bound the way cgo would bind it (tests/host/cabi_consumer.c) ----
def _build_consumer():
    import subprocess
    import gokzg_amd
    bdir = os.path.join(ROOT, "tests", "host", "_build")
    os.makedirs(bdir, exist_ok=True)
    exe = os.path.join(bdir, "cabi_consumer")
    libdir = os.path.dirname(gokzg_amd.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "cabi_consumer.c"), "-L", libdir, "-lkzg_hip", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_header_is_strict_c99(tmp_path):
    """include/kzg_hip.h on its own, as a cgo preamble would include it"""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "kzg_hip.h"\nint main(void) { return KZG_HIP_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])


def test_c_consumer_builds_and_refuses_without_a_device():
    import subprocess
    import gokzg_amd
    if gokzg_amd.device_count() > 0:
        pytest.skip("a GPU is present: test_c_consumer_runs_the_vectors covers it")
    res = subprocess.run([_build_consumer()], capture_output=True, text=True, timeout=120)
    assert res.returncode == 77, res.stdout + res.stderr            # 77 = "no device, and the library said so"
    assert "status 7 (want 7)" in res.stdout


@pytest.mark.gpu
def test_c_consumer_runs_the_vectors():
    """settings -> vectors A, B, C -> status codes 1..6 -> frees, from a C99 program that links -lkzg_hip"""
    import subprocess
    res = subprocess.run([_build_consumer()], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "PASSED: 0 failure(s)" in res.stdout and "FAIL " not in res.stdout
    for what in ("vector A", "vector B", "vector C: proof 18", "status 1 (want 1)", "status 2 (want 2)", "status 3 (want 3)", "status 4 (want 4)",
                 "status 5 (want 5)", "status 6 (want 6)", "multi: vector A from entry 1", "multi: vector C proof 18", "five all-gathers"):
        assert what in res.stdout, what


# ---- the C++ mirror of the Go API (include/kzg_hip.hpp) and the reference's tests re-stated against it (tests/host/go_mirror_test.cpp) ----
def _eth_aggregate_expected():
    """what eth.ComputeAggregateKZGProof returns for the two blobs of go_mirror_test.cpp (element i of blob b = i * i + 7 * b + 3), from the oracle:
    commitments (oracle MSM over the bit-reversed Lagrange setup), transcript + aggregation (oracle/pyref.py), proof (oracle MSM)"""
    import sys
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import koracle as ko, pyref
    R = ko.R_MOD
    lag = ko.g1_decompress(np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_g1_lagrange.bin"), "rb").read(), dtype=np.uint8))
    lag_br = ko.reverse_bit_order(lag)
    polys = [[i * i + 7 * b + 3 for i in range(4096)] for b in range(2)]
    comms = [ko.g1_compress(ko.lincomb_g1(lag_br, ko.fr_from_ints(p_)))[0].tobytes() for p_ in polys]
    agg, _, z = pyref.compute_aggregated_poly(polys, comms)
    pfs = pyref.FFTSettings(12)
    dom = [pfs.expanded[pyref.rev_bits(i, 12)] for i in range(4096)]
    y = pyref.eval_in_evaluation_form(agg, z, dom)
    q = [(p_ - y) * pow(w - z, -1, R) % R for p_, w in zip(agg, dom)]
    proof = ko.g1_compress(ko.lincomb_g1(lag_br, ko.fr_from_ints(q)))[0].tobytes()
    return [c.hex() for c in comms] + [proof.hex()]


def _build_go_mirror():
    import subprocess
    import gokzg_amd
    bdir = os.path.join(ROOT, "tests", "host", "_build")
    os.makedirs(bdir, exist_ok=True)
    exe = os.path.join(bdir, "go_mirror_test")
    libdir = os.path.dirname(gokzg_amd.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "go_mirror_test.cpp"), "-L", libdir, "-lkzg_hip", "-Wl,-rpath," + libdir, "-o", exe])
    import json
    g = os.path.join(ROOT, "tests", "golden")
    kats, der = json.load(open(os.path.join(g, "reference_kats.json"))), json.load(open(os.path.join(g, "derived_vectors.json")))
    c = der["C_da_using_fk20_scale5"]
    idx = sorted(int(k) for k in c if k.isdigit())
    lines = ["test_inv_fft " + " ".join(kats["test_inv_fft"]["expected"]), "test_das_fft_extension " + " ".join(kats["test_das_fft_extension"]["expected"]),
             "A_commit " + der["A_commit_test_poly"], "B_proof " + der["B_proof_single_x17"], "C_idx " + " ".join(str(i) for i in idx),
             "C_val " + " ".join(c[str(i)] for i in idx)]
    lines.append("eth_aggregate " + " ".join(_eth_aggregate_expected()))
    kat = os.path.join(bdir, "go_mirror_kats.txt")
    open(kat, "w").write("\n".join(lines) + "\n")
    return exe, kat


def test_cpp_mirror_builds_and_refuses_without_a_device():
    import subprocess
    import gokzg_amd
    exe, kat = _build_go_mirror()                                   # -Wall -Wextra -Werror: the header-only mirror compiles clean
    if gokzg_amd.device_count() > 0:
        pytest.skip("a GPU is present: test_cpp_mirror_runs_the_reference_tests covers it")
    res = subprocess.run([exe, kat], capture_output=True, text=True, timeout=120)
    assert res.returncode == 77, res.stdout + res.stderr


@pytest.mark.gpu
def test_cpp_mirror_runs_the_reference_tests():
    """TestFFTRoundtrip, TestInvFFT, TestDASFFTExtension, TestParametrizedDASFFTExtension, TestKZGSettings_*, TestFFTSettings_RecoverPolyFromSamples_Simple
    and the error / panic behaviour, from compiled C++ written against the Go-shaped mirror of the API"""
    import subprocess
    exe, kat = _build_go_mirror()
    res = subprocess.run([exe, kat], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "PASSED: 0 failure(s)" in res.stdout and "FAIL" not in res.stdout
    for name in ("TestInvFFT", "TestDASFFTExtension", "TestKZGSettings_DAUsingFK20", "TestErrorsAndPanics", "TestEth_ComputeAggregateKZGProof"):
        assert "ok   " + name in res.stdout, name


# ---- the cgo shim (go-kzg_amd/goshim/): no Go toolchain in this image, so the files are checked against the header textually ----
GOSHIM = os.path.join(ROOT, "go-kzg_amd", "goshim")


def header_prototypes(path=os.path.join(ROOT, "include", "kzg_hip.h")):
    """{function: number of parameters} and the set of KZG_HIP_* macros of the boundary header"""
    src = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(kzg_hip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        params = m.group(2).strip()
        protos[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    return protos, set(re.findall(r"#define\s+(KZG_HIP_[A-Z0-9_]+)", src))


def go_c_calls(text):
    """(name, argument count) of every C.kzg_hip_*(...) call in a Go source text; comments and string literals are skipped"""
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)
    calls = []
    for m in re.finditer(r"\bC\.(kzg_hip_[a-z0-9_]+)\s*\(", text):
        i, depth, args, cur = m.end(), 1, 0, False
        while depth:
            ch = text[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                args += 1
            if not ch.isspace() and depth >= 1 and not (ch == ")" and depth == 0):
                cur = True
            i += 1
        calls.append((m.group(1), args + 1 if cur else 0))
    return calls


def go_sources():
    out = []
    for dirpath, _, files in os.walk(GOSHIM):
        for f in sorted(files):
            assert f.endswith(".go"), "only compilable Go sources belong in goshim/: " + f
            out.append(os.path.join(dirpath, f))
    return sorted(out)


def test_go_shim_calls_match_the_header():
    """every C.kzg_hip_* call of the shim names a function of include/kzg_hip.h with the right number of arguments, every C.KZG_HIP_* constant
    exists, type names are the header's opaque types, braces balance, each file belongs to the package of its directory and carries the build tag"""
    protos, macros = header_prototypes()
    types = set(re.findall(r"typedef\s+struct\s+(kzg_hip_[a-z0-9_]+)\s+\1\s*;", open(os.path.join(ROOT, "include", "kzg_hip.h")).read()))   # the header's opaque handles
    assert {"kzg_hip_fft", "kzg_hip_kzg", "kzg_hip_multi"} <= types
    files = go_sources()
    assert {os.path.basename(os.path.dirname(f)) for f in files} == {"kzg", "bls", "eth"}
    bound = set()
    for path in files:
        text = open(path).read()
        pkg = os.path.basename(os.path.dirname(path))
        assert re.search(r"^package %s$" % pkg, text, flags=re.M), path
        assert text.startswith("//go:build kzg_hip"), path
        assert '#include "kzg_hip.h"' in text and 'import "C"' in text, path
        code = re.sub(r'"(?:\\.|[^"\\])*"', '""', re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
        for o, c in ("{}", "()", "[]"):
            assert code.count(o) == code.count(c), (path, o)
        for name, nargs in go_c_calls(text):
            assert name in protos, "%s calls C.%s, which include/kzg_hip.h does not declare" % (path, name)
            assert nargs == protos[name], "%s: C.%s called with %d arguments, the header declares %d" % (path, name, nargs, protos[name])
            bound.add(name)
        for mac in re.findall(r"\bC\.(KZG_HIP_[A-Z0-9_]+)", text):
            assert mac in macros, (path, mac)
        for t in re.findall(r"\*C\.(kzg_hip_[a-z0-9_]+)\b(?!\s*\()", text):
            assert t in types, (path, t)
    # every host-buffer entry point of the boundary is reachable from Go; the _dev forms take device pointers and a HIP stream (no Go value
    # can be one) and a few introspection calls are for tests
    not_for_go = {n for n in protos if n.endswith("_dev")} | {"kzg_hip_fft_max_width", "kzg_hip_fft_roots", "kzg_hip_multi_device", "kzg_hip_multi_device_count",
                                                              "kzg_hip_multi_exchanges", "kzg_hip_multi_fft", "kzg_hip_multi_kzg", "kzg_hip_g1_marshal_text",
                                                              "kzg_hip_g1_unmarshal_text"}
    missing = sorted(set(protos) - bound - not_for_go)
    assert not missing, "header functions without a Go binding: %s" % missing
    for must in ("kzg_hip_eth_settings_free", "kzg_hip_generate_testing_setup_g1", "kzg_hip_evaluate_poly_in_evaluation_form", "kzg_hip_multi_settings_new",
                 "kzg_hip_multi_da_using_fk20_multi"):
        assert must in bound, must


def test_go_shim_parser_catches_drift():
    """the checker itself: a renamed function and a dropped argument are both reported"""
    protos, _ = header_prototypes()
    assert protos["kzg_hip_fft_settings_new"] == 3 and protos["kzg_hip_device_count"] == 0 and protos["kzg_hip_multi_settings_new"] == 6
    calls = dict(go_c_calls('x := C.kzg_hip_fft_settings_new(C.int(dev), 12, &fs) // C.kzg_hip_nope(1)\ny := C.kzg_hip_device_count()\n'
                            'z := C.kzg_hip_commit_to_poly(ks.hip(), frPtr(c), C.uint64_t(len(c)))'))
    assert calls == {"kzg_hip_fft_settings_new": 3, "kzg_hip_device_count": 0, "kzg_hip_commit_to_poly": 3}
    assert calls["kzg_hip_commit_to_poly"] != protos["kzg_hip_commit_to_poly"]

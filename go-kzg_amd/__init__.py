"""go-kzg_amd -- host-side mirror of go-kzg's FFTSettings / KZGSettings / FK20*Settings API over libkzg_hip.so.

The reference's host language is Go, which this image does not have; the drop-in boundary is the C ABI in
include/kzg_hip.h (the cgo shim for it is in go-kzg_amd/goshim/ and INTEGRATION.md).  This module is the same
thin binding written with ctypes so that the parity tests read like the reference's own tests:

    fs = FFTSettings(4)                         # kzg.NewFFTSettings(4)              fft.go:44
    ks = KZGSettings(fs, secret_g1)             # kzg.NewKZGSettings(fs, s1, s2)     kzg.go:21
    c  = ks.commit_to_poly(poly)                # ks.CommitToPoly(poly)              kzg_single_proofs.go:17
    fk = FK20SingleSettings(ks, 32)             # kzg.NewFK20SingleSettings(ks, 32)  kzg.go:43
    pr = fk.da_using_fk20(poly)                 # fk.DAUsingFK20(poly)               fk20_single.go:176

Values are numpy uint64 arrays holding the memory images of the reference's default backend:
Fr -> (n, 4) Montgomery limbs; G1 -> (n, 3, 6) Jacobian Montgomery limbs (inf <=> Z == 0).

There is NO CPU fallback: importing works anywhere (so the C ABI can be inspected), but constructing
FFTSettings without a gfx950 device raises NoDeviceError, and a missing libkzg_hip.so raises ImportError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KZG_HIP_LIB") or os.path.join(_HERE, "libkzg_hip.so")   # KZG_HIP_LIB: another build of the same library (A/B runs)

OK, ERR_TOO_WIDE, ERR_NOT_POW2, ERR_LEN_MISMATCH, ERR_UPPER_HALF, ERR_BAD_ARG, ERR_BAD_POINT, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_RECOVERY, ERR_BAD_BLOB = range(12)


class KzgError(Exception):
    """The reference returns an `error` (fft_fr.go:57-59,78-83; fft_g1.go:60-65)."""

    def __init__(self, status, msg=""):
        super().__init__("kzg_hip status %d %s" % (status, msg))
        self.status = status


class KzgPanic(KzgError):
    """The reference panics (kzg.go:22-27,44-52,74-91; fk20_single.go:60-62,140-154; bls_kilic.go:133-135)."""


class NoDeviceError(KzgError):
    pass


_lib = None


def lib():
    """Loads libkzg_hip.so (built by `make -C go-kzg_amd/csrc` / __graft_entry__.build()); fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libkzg_hip.so is not built (run __graft_entry__.build()); there is no CPU fallback: " + LIB_PATH)
    if os.environ.get("KZG_HIP_NO_TORCH_PRELOAD") != "1":
        # PyTorch bundles its own libamdhip64; if this library initialises /opt/rocm's copy first, torch later reports
        # "No HIP GPUs are available".  Importing torch first makes both resolve to ONE HIP runtime (same SONAME).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int, C.c_uint
    pp = C.POINTER(C.c_void_p)
    sig = {
        "kzg_hip_device_count": (i32, []), "kzg_hip_last_error": (C.c_char_p, []), "kzg_hip_version": (C.c_char_p, []),
        "kzg_hip_host_register": (i32, [vp, u64]), "kzg_hip_host_unregister": (i32, [vp]),
        "kzg_hip_fft_settings_new": (i32, [i32, u32, pp]), "kzg_hip_fft_settings_free": (None, [vp]),
        "kzg_hip_fft_max_width": (u64, [vp]), "kzg_hip_fft_roots": (i32, [vp, i32, vp]),
        "kzg_hip_fft_fr": (i32, [vp, vp, u64, i32, vp, C.POINTER(u64)]), "kzg_hip_inplace_fft_fr": (i32, [vp, vp, vp, u64, i32]),
        "kzg_hip_fft_fr_batch": (i32, [vp, vp, u64, u64, i32, vp]), "kzg_hip_fft_g1": (i32, [vp, vp, u64, i32, vp]),
        "kzg_hip_fft_g1_batch": (i32, [vp, vp, u64, u64, i32, vp]), "kzg_hip_multi_fft_g1_batch": (i32, [vp, vp, u64, u64, i32, vp]),
        "kzg_hip_das_fft_extension": (i32, [vp, vp, u64]), "kzg_hip_das_fft_extension_batch": (i32, [vp, vp, u64, u64]),
        "kzg_hip_fft_fr_batch_dev": (i32, [vp, vp, u64, u64, i32, vp, vp]), "kzg_hip_fft_g1_batch_dev": (i32, [vp, vp, u64, u64, i32, vp, vp]),
        "kzg_hip_das_fft_extension_batch_dev": (i32, [vp, vp, u64, u64, vp]),
        "kzg_hip_fr_from_le32": (i32, [vp, vp, u64, vp, C.POINTER(i32)]), "kzg_hip_fr_to_le32": (i32, [vp, vp, u64, vp]),
        "kzg_hip_lincomb_g1": (i32, [vp, vp, vp, u64, vp]),
        "kzg_hip_points_new": (i32, [vp, vp, u64, pp]), "kzg_hip_points_free": (None, [vp]), "kzg_hip_points_count": (u64, [vp]),
        "kzg_hip_points_set_table_budget_gb": (i32, [vp, C.c_double]),
        "kzg_hip_lincomb_points": (i32, [vp, vp, u64, vp]), "kzg_hip_lincomb_points_batch": (i32, [vp, vp, u64, u64, vp]),
        "kzg_hip_lincomb_points_batch_dev": (i32, [vp, vp, u64, u64, vp, vp]), "kzg_hip_g1_to_compressed": (i32, [vp, vp, u64, vp]),
        "kzg_hip_g1_from_compressed": (i32, [vp, vp, u64, vp]), "kzg_hip_g1_mul_vec": (i32, [vp, vp, vp, u64, vp]),
        "kzg_hip_generate_testing_setup_g1": (i32, [vp, vp, u64, vp]),
        "kzg_hip_g1_marshal_text": (i32, [vp, vp, u64, vp]), "kzg_hip_g1_unmarshal_text": (i32, [vp, C.c_char_p, u64, vp]),
        "kzg_hip_trusted_setup_from_json": (i32, [vp, C.c_char_p, u64, vp, vp, u64, C.POINTER(u64), C.POINTER(u64)]),
        "kzg_hip_kzg_set_table_budget_gb": (i32, [vp, C.c_double]),
        "kzg_hip_kzg_settings_new": (i32, [vp, vp, u64, pp]), "kzg_hip_kzg_settings_free": (None, [vp]),
        "kzg_hip_commit_to_poly": (i32, [vp, vp, u64, vp]), "kzg_hip_commit_to_poly_batch": (i32, [vp, vp, u64, u64, vp]),
        "kzg_hip_commit_to_poly_batch_dev": (i32, [vp, vp, u64, u64, vp, vp]),
        "kzg_hip_compute_proof_single": (i32, [vp, vp, u64, u64, vp]),
        "kzg_hip_compute_proof_single_batch": (i32, [vp, vp, u64, u64, vp, vp]),
        "kzg_hip_compute_proof_single_batch_dev": (i32, [vp, vp, u64, u64, vp, vp, vp]),
        "kzg_hip_compute_proof_multi": (i32, [vp, vp, u64, u64, u64, vp]),
        "kzg_hip_check_proof_multi_interpolation": (i32, [vp, vp, u64, vp, vp, vp]),
        "kzg_hip_toeplitz_part2": (i32, [vp, vp, vp, u64, vp]), "kzg_hip_toeplitz_part3": (i32, [vp, vp, u64, vp]),
        "kzg_hip_fk20_single_settings_new": (i32, [vp, u64, pp]), "kzg_hip_fk20_single_settings_free": (None, [vp]),
        "kzg_hip_fk20_single_x_ext_fft": (i32, [vp, vp]), "kzg_hip_fk20_single": (i32, [vp, vp, u64, vp]),
        "kzg_hip_fk20_single_batch": (i32, [vp, vp, u64, u64, vp]), "kzg_hip_fk20_single_batch_dev": (i32, [vp, vp, u64, u64, vp, vp]),
        "kzg_hip_fk20_single_da_optimized": (i32, [vp, vp, u64, vp]), "kzg_hip_da_using_fk20": (i32, [vp, vp, u64, vp]),
        "kzg_hip_da_using_fk20_batch": (i32, [vp, vp, u64, u64, vp]), "kzg_hip_da_using_fk20_batch_dev": (i32, [vp, vp, u64, u64, vp, vp]),
        "kzg_hip_fk20_multi_settings_new": (i32, [vp, u64, u64, pp]), "kzg_hip_fk20_multi_settings_free": (None, [vp]),
        "kzg_hip_fk20_multi": (i32, [vp, vp, u64, vp]), "kzg_hip_fk20_multi_da_optimized": (i32, [vp, vp, u64, vp]),
        "kzg_hip_da_using_fk20_multi": (i32, [vp, vp, u64, vp]),
        "kzg_hip_da_using_fk20_multi_batch_dev": (i32, [vp, vp, u64, u64, vp, vp]),
        "kzg_hip_fk20_multi_hext_slice_dev": (i32, [vp, vp, u64, u64, u64, vp, vp]),
        "kzg_hip_fk20_multi_finish_dev": (i32, [vp, vp, i32, vp, vp]),
        "kzg_hip_eth_settings_new": (i32, [vp, vp, u64, pp]), "kzg_hip_eth_settings_free": (None, [vp]),
        "kzg_hip_eth_blob_to_kzg_commitment_batch": (i32, [vp, vp, u64, vp, vp]),
        "kzg_hip_eth_compute_kzg_proof": (i32, [vp, vp, u64, vp, vp, vp]),
        "kzg_hip_eth_compute_kzg_proof_batch": (i32, [vp, vp, u64, u64, vp, vp, vp, vp]),
        "kzg_hip_eth_compute_kzg_proof_batch_dev": (i32, [vp, vp, u64, u64, vp, vp, vp, vp, vp]),
        "kzg_hip_eth_compute_aggregate_kzg_proof": (i32, [vp, vp, u64, vp, vp]),
        "kzg_hip_eth_compute_aggregated_poly_and_commitment": (i32, [vp, vp, vp, u64, vp, vp, vp, vp]),
        "kzg_hip_test_sha256": (None, [vp, u64, vp]),
        "kzg_hip_evaluate_poly_in_evaluation_form": (i32, [vp, vp, u64, vp, u32, vp]),
        "kzg_hip_eth_evaluate_polynomial_in_evaluation_form": (i32, [vp, vp, u64, vp, vp]),
        "kzg_hip_bench_threads_fft_fr": (i32, [vp, vp, u64, u64, u32, u32, vp, C.POINTER(C.c_double)]),
        "kzg_hip_bench_drop_in_eth_proof": (i32, [vp, vp, u64, u64, u32, u32, vp, C.POINTER(C.c_double)]),
        "kzg_hip_zero_poly_via_multiplication": (i32, [vp, vp, u64, u64, vp, vp]),
        "kzg_hip_recover_poly_from_samples": (i32, [vp, vp, vp, u64, vp]),
        "kzg_hip_calibrate": (i32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "kzg_hip_bench_drop_in": (i32, [vp, i32, vp, u64, u64, u32, u32, vp, C.POINTER(C.c_double)]),
        "kzg_hip_coalesce_stats": (i32, [vp, i32, C.POINTER(u64)]),
        "kzg_hip_lincomb_promotions": (i32, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "kzg_hip_test_fp_inv": (i32, [vp, vp, u64, vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "kzg_hip_test_fr_inv": (i32, [vp, vp, u64, vp, vp, vp]),
        "kzg_hip_kzg_table_info": (i32, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(u64)]), "kzg_hip_kzg_table_additions": (u32, [vp]), "kzg_hip_kzg_set_projective_outputs": (i32, [vp, i32]),
        "kzg_hip_da_using_fk20_multi_batch": (i32, [vp, vp, u64, u64, vp]),
        "kzg_hip_multi_settings_new": (i32, [C.POINTER(i32), u32, u32, vp, u64, pp]), "kzg_hip_multi_settings_free": (None, [vp]),
        "kzg_hip_multi_device_count": (u32, [vp]), "kzg_hip_multi_device": (i32, [vp, u32]),
        "kzg_hip_multi_fft": (vp, [vp, u32]), "kzg_hip_multi_kzg": (vp, [vp, u32]),
        "kzg_hip_multi_transport": (C.c_char_p, [vp]), "kzg_hip_multi_transport_note": (C.c_char_p, [vp]), "kzg_hip_multi_transport_check": (C.c_char_p, [vp]), "kzg_hip_multi_exchanges": (u64, [vp]),
        "kzg_hip_multi_set_fft_sharding": (i32, [vp, i32]), "kzg_hip_multi_set_table_budget_gb": (i32, [vp, C.c_double]),
        "kzg_hip_multi_commit_to_poly_batch": (i32, [vp, vp, u64, u64, vp]),
        "kzg_hip_multi_compute_proof_single_batch": (i32, [vp, vp, u64, u64, vp, vp]),
        "kzg_hip_multi_fft_fr_batch": (i32, [vp, vp, u64, u64, i32, vp]), "kzg_hip_multi_das_fft_extension_batch": (i32, [vp, vp, u64, u64]),
        "kzg_hip_multi_eth_settings_new": (i32, [vp, vp, u64, pp]), "kzg_hip_multi_eth_settings_free": (None, [vp]),
        "kzg_hip_multi_eth_blob_to_kzg_commitment_batch": (i32, [vp, vp, u64, vp, vp]),
        "kzg_hip_multi_eth_compute_kzg_proof_batch": (i32, [vp, vp, u64, u64, vp, vp, vp, vp]),
        "kzg_hip_multi_fk20_single_settings_new": (i32, [vp, u64, pp]), "kzg_hip_multi_fk20_single_settings_free": (None, [vp]),
        "kzg_hip_multi_da_using_fk20_batch": (i32, [vp, vp, u64, u64, vp]), "kzg_hip_multi_da_using_fk20": (i32, [vp, vp, u64, vp]),
        "kzg_hip_multi_fk20_multi_settings_new": (i32, [vp, u64, u64, pp]), "kzg_hip_multi_fk20_multi_settings_free": (None, [vp]),
        "kzg_hip_multi_da_using_fk20_multi_batch": (i32, [vp, vp, u64, u64, vp]), "kzg_hip_multi_da_using_fk20_multi": (i32, [vp, vp, u64, vp]),
        "kzg_hip_bench_poly_lincomb_dev": (i32, [vp, vp, u64, vp, u64, u64, vp, vp]),
        "kzg_hip_prof_reset": (None, [vp, i32]), "kzg_hip_prof_read": (i32, [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(u64)]),
    }
    for name, (res, args) in sig.items():
        if os.environ.get("KZG_HIP_LIB_ALLOW_MISSING") and not hasattr(L, name):
            continue           # A/B runs against an older build selected with KZG_HIP_LIB (tools/ab_latency.sh)
        f = getattr(L, name)   # AttributeError here == header / library drift
        f.restype, f.argtypes = res, args
    _lib = L
    return L


API_SYMBOLS = None  # filled by tests from include/kzg_hip.h


def device_count():
    return lib().kzg_hip_device_count()


class pinned:
    """`with pinned(array):` -- the array's memory is pinned (kzg_hip_host_register) for the duration: batch calls read it in place over PCIe"""

    def __init__(self, array):
        self.a = array

    def __enter__(self):
        _chk(lib().kzg_hip_host_register(self.a.ctypes.data, self.a.nbytes))
        return self.a

    def __exit__(self, *exc):
        lib().kzg_hip_host_unregister(self.a.ctypes.data)
        return False


_ERR_STATUS = (ERR_TOO_WIDE, ERR_NOT_POW2)


def _chk(st, error_ok=False):
    if st == OK:
        return
    msg = ""
    if st == ERR_HIP:
        msg = lib().kzg_hip_last_error().decode()
    if st == ERR_NO_DEVICE:
        raise NoDeviceError(st, "no gfx950 device visible (the HIP path is the only path)")
    if error_ok and st in _ERR_STATUS:
        raise KzgError(st, msg)
    raise KzgPanic(st, msg)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _fr(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, 4)


def _g1(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, 3, 6)


def fr_empty(n):
    return np.zeros((n, 4), dtype=np.uint64)


def g1_empty(n):
    return np.zeros((n, 3, 6), dtype=np.uint64)


class FFTSettings:
    """kzg.FFTSettings (fft.go:34-61) with a device-resident domain."""

    def __init__(self, max_scale, device=0):
        h = C.c_void_p()
        _chk(lib().kzg_hip_fft_settings_new(device, max_scale, C.byref(h)))
        self.h, self.max_scale, self.max_width, self.device = h, max_scale, 1 << max_scale, device

    def close(self):
        if getattr(self, "h", None):
            lib().kzg_hip_fft_settings_free(self.h)
            self.h = None

    def expanded_roots_of_unity(self):
        out = fr_empty(self.max_width + 1)
        _chk(lib().kzg_hip_fft_roots(self.h, 0, _p(out)))
        return out

    def reverse_roots_of_unity(self):
        out = fr_empty(self.max_width + 1)
        _chk(lib().kzg_hip_fft_roots(self.h, 1, _p(out)))
        return out

    def fft(self, vals, inv=False):
        """FFTSettings.FFT (fft_fr.go:55-74)"""
        vals = _fr(vals)
        n = vals.shape[0]
        if n > self.max_width:
            raise KzgError(ERR_TOO_WIDE)
        np2 = 1 if n == 0 else 1 << (n - 1).bit_length()
        out, on = fr_empty(np2), C.c_uint64(0)
        _chk(lib().kzg_hip_fft_fr(self.h, _p(vals), n, int(inv), _p(out), C.byref(on)), error_ok=True)
        return out

    def inplace_fft(self, vals, inv=False):
        """FFTSettings.InplaceFFT (fft_fr.go:76-105); returns `out`"""
        vals = _fr(vals)
        out = fr_empty(vals.shape[0])
        _chk(lib().kzg_hip_inplace_fft_fr(self.h, _p(vals), _p(out), vals.shape[0], int(inv)), error_ok=True)
        return out

    def fft_batch(self, vals, inv=False):
        vals = np.ascontiguousarray(vals, dtype=np.uint64)
        b, n = vals.shape[0], vals.shape[1]
        out = np.zeros_like(vals)
        _chk(lib().kzg_hip_fft_fr_batch(self.h, _p(vals), n, b, int(inv), _p(out)), error_ok=True)
        return out

    def fft_g1(self, vals, inv=False):
        """FFTSettings.FFTG1 (fft_g1.go:58-94)"""
        vals = _g1(vals)
        out = g1_empty(vals.shape[0])
        _chk(lib().kzg_hip_fft_g1(self.h, _p(vals), vals.shape[0], int(inv), _p(out)), error_ok=True)
        return out

    def fft_g1_batch(self, vals, inv=False):
        """FFTG1 on every row of vals (batch, n, 3, 6)"""
        vals = np.ascontiguousarray(vals, dtype=np.uint64)
        out = np.zeros_like(vals)
        _chk(lib().kzg_hip_fft_g1_batch(self.h, _p(vals), vals.shape[1], vals.shape[0], int(inv), _p(out)), error_ok=True)
        return out

    def evaluate_poly_in_evaluation_form(self, poly, x, scale=0):
        """bls.EvaluatePolyInEvaluationForm(y, poly, x, fs.ExpandedRootsOfUnity[:fs.MaxWidth], scale) (bls/globals.go:106-153)"""
        poly, x, y = _fr(poly), _fr(x), fr_empty(1)
        _chk(lib().kzg_hip_evaluate_poly_in_evaluation_form(self.h, _p(poly), poly.shape[0], _p(x), scale, _p(y)))
        return y[0]

    def das_fft_extension(self, vals):
        """FFTSettings.DASFFTExtension (das_extension.go:71-84); returns the odd values (the reference writes in place)"""
        vals = _fr(vals).copy()
        _chk(lib().kzg_hip_das_fft_extension(self.h, _p(vals), vals.shape[0]))
        return vals

    def das_fft_extension_batch(self, vals):
        vals = np.ascontiguousarray(vals, dtype=np.uint64).copy()
        _chk(lib().kzg_hip_das_fft_extension_batch(self.h, _p(vals), vals.shape[1], vals.shape[0]))
        return vals

    def fr_from_32(self, data):
        """bls.FrFrom32 over a slice (bls/bignum_kilic.go:33-44): (n, 32) LE bytes -> ((n, 4) images, all_ok)"""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, 32)
        out, ok = fr_empty(data.shape[0]), C.c_int(1)
        _chk(lib().kzg_hip_fr_from_le32(self.h, _p(data), data.shape[0], _p(out), C.byref(ok)))
        return out, bool(ok.value)

    def fr_to_32(self, vals):
        """bls.FrTo32 over a slice (bls/bignum_kilic.go:46-55)"""
        vals = _fr(vals)
        out = np.zeros((vals.shape[0], 32), dtype=np.uint8)
        _chk(lib().kzg_hip_fr_to_le32(self.h, _p(vals), vals.shape[0], _p(out)))
        return out

    def calibrate(self):
        """measured issue rates of this GPU: (v_mad_u64_u32 lane-ops/s, v_add_u32 lane-ops/s, lazy F_p products/s)"""
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        _chk(lib().kzg_hip_calibrate(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def zero_poly_via_multiplication(self, missing_indices, length):
        """FFTSettings.ZeroPolyViaMultiplication (zero_poly.go:116-217): (zero_eval, zero_poly)"""
        mi = np.ascontiguousarray(missing_indices, dtype=np.uint64)
        ze, zp = fr_empty(length), fr_empty(length)
        _chk(lib().kzg_hip_zero_poly_via_multiplication(self.h, _p(mi), mi.shape[0], length, _p(ze), _p(zp)))
        return ze, zp

    def recover_poly_from_samples(self, samples, present):
        """FFTSettings.RecoverPolyFromSamples (recover_from_samples.go:42-109); present[i] False <=> samples[i] is nil"""
        samples = _fr(samples)
        present = np.ascontiguousarray(present, dtype=np.uint8)
        out = fr_empty(samples.shape[0])
        st = lib().kzg_hip_recover_poly_from_samples(self.h, _p(samples), _p(present), samples.shape[0], _p(out))
        if st == ERR_RECOVERY:
            raise KzgError(st, "failed to reconstruct data correctly")
        _chk(st, error_ok=True)
        return out

    def bench_threads_fft(self, rows, threads, calls):
        """`threads` native host threads x `calls` blocking FFT calls on host buffers (kzg_hip_bench_threads_fft_fr): (calls per second, last results)"""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out, secs = np.zeros((threads, rows.shape[1], 4), dtype=np.uint64), C.c_double(0)
        _chk(lib().kzg_hip_bench_threads_fft_fr(self.h, _p(rows), rows.shape[1], rows.shape[0], threads, calls, _p(out), C.byref(secs)))
        return threads * calls / secs.value, out

    # ---- bls.* batch helpers that need a device context ----
    def lin_comb_g1(self, numbers, factors):
        """bls.LinCombG1 (bls/bls_kilic.go:132-150)"""
        numbers, factors = _g1(numbers), _fr(factors)
        if numbers.shape[0] != factors.shape[0]:
            raise KzgPanic(ERR_LEN_MISMATCH, "got LinCombG1 numbers/factors length mismatch")
        out = g1_empty(1)
        _chk(lib().kzg_hip_lincomb_g1(self.h, _p(numbers), _p(factors), numbers.shape[0], _p(out)))
        return out[0]

    def lincomb_promotions(self):
        """(point sets promoted so far, calls served by a promoted set): kzg_hip_lincomb_g1 turns a caller-supplied point set that keeps coming back into a cached one"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _chk(lib().kzg_hip_lincomb_promotions(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def to_compressed_g1(self, points):
        points = _g1(points)
        out = np.zeros((points.shape[0], 48), dtype=np.uint8)
        _chk(lib().kzg_hip_g1_to_compressed(self.h, _p(points), points.shape[0], _p(out)))
        return out

    def from_compressed_g1(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, 48)
        out = g1_empty(data.shape[0])
        _chk(lib().kzg_hip_g1_from_compressed(self.h, _p(data), data.shape[0], _p(out)))
        return out

    def g1_marshal_text(self, points):
        """bls.G1Point.MarshalText over a slice (bls/bls_all.go:20-22): lower-case hex of the 48-byte compressed form"""
        points = _g1(points)
        n = points.shape[0]
        buf = C.create_string_buffer(96 * n + 1)
        _chk(lib().kzg_hip_g1_marshal_text(self.h, _p(points), n, buf))
        txt = buf.raw[:96 * n].decode("ascii")
        return [txt[96 * i:96 * i + 96] for i in range(n)]

    def g1_unmarshal_text(self, texts):
        """bls.G1Point.UnmarshalText over a slice (bls/bls_all.go:24-39); hex decoding in the library, decompression on the device"""
        texts = list(texts)
        if any(len(t) != 96 for t in texts):
            raise KzgPanic(ERR_BAD_POINT, "expected 48-byte compressed G1 points")
        out = g1_empty(len(texts))
        _chk(lib().kzg_hip_g1_unmarshal_text(self.h, "".join(texts).encode("ascii", "replace"), len(texts), _p(out)))
        return out

    def trusted_setup_from_json(self, text):
        """JSONTrustedSetup (eth/globals.go:33-49): JSON text -> (setup_G1, setup_G1_lagrange) as Kilic images; G2 is skipped"""
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        n1, n2 = C.c_uint64(0), C.c_uint64(0)
        _chk(lib().kzg_hip_trusted_setup_from_json(self.h, raw, len(raw), None, None, 0, C.byref(n1), C.byref(n2)))
        cap = max(n1.value, n2.value, 1)
        mono, lag = g1_empty(cap), g1_empty(cap)
        _chk(lib().kzg_hip_trusted_setup_from_json(self.h, raw, len(raw), _p(mono), _p(lag), cap, C.byref(n1), C.byref(n2)))
        return mono[:n1.value], lag[:n2.value]

    def mul_g1_vec(self, points, scalars):
        points, scalars = _g1(points), _fr(scalars)
        out = g1_empty(points.shape[0])
        _chk(lib().kzg_hip_g1_mul_vec(self.h, _p(points), _p(scalars), points.shape[0], _p(out)))
        return out

    def generate_testing_setup_g1(self, secret_fr, n):
        """GenerateTestingSetup (setup.go:9-26), G1 half; secret_fr is the Montgomery image of the secret"""
        secret_fr = _fr(secret_fr)
        out = g1_empty(n)
        _chk(lib().kzg_hip_generate_testing_setup_g1(self.h, _p(secret_fr), n, _p(out)))
        return out


class G1Points:
    """a point set kept in HBM for repeated bls.LinCombG1 calls (kzg_hip_points_*): CommitToEvalPoly's secretG1IFFT, eth's Lagrange setup"""

    def __init__(self, fs, points):
        points = _g1(points)
        h = C.c_void_p()
        _chk(lib().kzg_hip_points_new(fs.h, _p(points), points.shape[0], C.byref(h)))
        self.h, self.fs, self.n = h, fs, points.shape[0]

    def close(self):
        if getattr(self, "h", None):
            lib().kzg_hip_points_free(self.h)
            self.h = None

    def set_table_budget_gb(self, gb):
        """HBM budget of the set's fixed-base table (0: bucket pipeline only); see kzg_hip_points_set_table_budget_gb"""
        _chk(lib().kzg_hip_points_set_table_budget_gb(self.h, float(gb)))

    def lin_comb(self, factors):
        """bls.LinCombG1(points[:len(factors)], factors)"""
        factors = _fr(factors)
        out = g1_empty(1)
        _chk(lib().kzg_hip_lincomb_points(self.h, _p(factors), factors.shape[0], _p(out)))
        return out[0]

    def lin_comb_batch(self, factors):
        factors = np.ascontiguousarray(factors, dtype=np.uint64)
        b, n = factors.shape[0], factors.shape[1]
        out = g1_empty(b)
        _chk(lib().kzg_hip_lincomb_points_batch(self.h, _p(factors), n, b, _p(out)))
        return out


def commit_to_eval_poly(fs, secret_g1_ifft, eval_poly):
    """kzg.CommitToEvalPoly (kzg_single_proofs.go:12-14)"""
    return fs.lin_comb_g1(secret_g1_ifft, eval_poly)


class KZGSettings:
    """kzg.KZGSettings, prover side (kzg.go:11-36): SecretG1 is uploaded once and stays in HBM."""

    def __init__(self, fs, secret_g1):
        secret_g1 = _g1(secret_g1)
        h = C.c_void_p()
        _chk(lib().kzg_hip_kzg_settings_new(fs.h, _p(secret_g1), secret_g1.shape[0], C.byref(h)))
        self.h, self.fs, self.n_setup = h, fs, secret_g1.shape[0]

    def close(self):
        if getattr(self, "h", None):
            lib().kzg_hip_kzg_settings_free(self.h)
            self.h = None

    def commit_to_poly_unoptimized(self, coeffs):
        """KZGSettings.CommitToPolyUnoptimized (kzg_single_proofs.go:22-33): the same group element as CommitToPoly, hence the same call"""
        return self.commit_to_poly(coeffs)

    def set_table_budget_gb(self, gb):
        """HBM budget of the fixed-base commitment table (default 110 GB: 4096 points get signed 16-bit windows, 8 of them walked by both GLV halves, 103 GB)"""
        _chk(lib().kzg_hip_kzg_set_table_budget_gb(self.h, float(gb)))

    def bench_drop_in(self, blobs, threads, calls, op=0):
        """`threads` native host threads x `calls` blocking one-polynomial calls (kzg_hip_bench_drop_in): (calls per second, last results)"""
        blobs = np.ascontiguousarray(blobs, dtype=np.uint64)
        out, secs = g1_empty(threads), C.c_double(0)
        _chk(lib().kzg_hip_bench_drop_in(self.h, op, _p(blobs), blobs.shape[1], blobs.shape[0], threads, calls, _p(out), C.byref(secs)))
        return threads * calls / secs.value, out

    def coalesce_stats(self, op=0):
        """cumulative statistics of the coalescer behind the one-polynomial calls (op 0 CommitToPoly, 1 ComputeProofSingle)"""
        a = (C.c_uint64 * 8)()
        _chk(lib().kzg_hip_coalesce_stats(self.h, op, a))
        req, bat = a[0], a[1]
        return {"requests": req, "batches": bat, "avg_batch": req / bat if bat else 0.0,
                "ms_per_batch": {"executing": a[2] * 1e-6 / bat if bat else 0.0, "waiting_for_a_device_slot": a[3] * 1e-6 / bat if bat else 0.0,
                                 "gathering_callers": a[4] * 1e-6 / bat if bat else 0.0, "waiting_for_row_copies": a[5] * 1e-6 / bat if bat else 0.0},
                "largest_concurrency_estimate": a[6], "batches_in_flight_limit": a[7]}

    def table_info(self):
        """(window bits, windows, bytes) of the fixed-base table the commitments walk; zeros before the first commitment"""
        c, w, b = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
        _chk(lib().kzg_hip_kzg_table_info(self.h, C.byref(c), C.byref(w), C.byref(b)))
        return c.value, w.value, b.value

    def set_projective_outputs(self, on=True):
        """CommitToPoly / ComputeProofSingle return un-normalised Jacobian images (the reference's own return type): no inversion per result"""
        _chk(lib().kzg_hip_kzg_set_projective_outputs(self.h, 1 if on else 0))

    def table_additions(self):
        """mixed additions per coefficient on that table (2 x windows: both GLV halves of a scalar walk the same rows)"""
        return lib().kzg_hip_kzg_table_additions(self.h)

    def commit_to_poly(self, coeffs):
        coeffs = _fr(coeffs)
        out = g1_empty(1)
        _chk(lib().kzg_hip_commit_to_poly(self.h, _p(coeffs), coeffs.shape[0], _p(out)))
        return out[0]

    def commit_to_poly_batch(self, coeffs):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        b, n = coeffs.shape[0], coeffs.shape[1]
        out = g1_empty(b)
        _chk(lib().kzg_hip_commit_to_poly_batch(self.h, _p(coeffs), n, b, _p(out)))
        return out

    def compute_proof_single(self, poly, x):
        poly = _fr(poly)
        out = g1_empty(1)
        _chk(lib().kzg_hip_compute_proof_single(self.h, _p(poly), poly.shape[0], x, _p(out)))
        return out[0]

    def compute_proof_single_batch(self, polys, xs):
        """ComputeProofSingle on a batch: polys (batch, n, 4), xs (batch,) uint64 -> (batch, 3, 6)"""
        polys = np.ascontiguousarray(polys, dtype=np.uint64)
        xs = np.ascontiguousarray(xs, dtype=np.uint64)
        b, n = polys.shape[0], polys.shape[1]
        out = g1_empty(b)
        _chk(lib().kzg_hip_compute_proof_single_batch(self.h, _p(polys), n, b, _p(xs), _p(out)))
        return out

    def compute_proof_multi(self, poly, x, n):
        """KZGSettings.ComputeProofMulti (kzg_multi_proofs.go:13-43), reference quirk (divisor X^n) included"""
        poly = _fr(poly)
        out = g1_empty(1)
        _chk(lib().kzg_hip_compute_proof_multi(self.h, _p(poly), poly.shape[0], x, n, _p(out)))
        return out[0]

    def check_proof_multi_interpolation(self, ys, x):
        """prover-side half of KZGSettings.CheckProofMulti (kzg_multi_proofs.go:47-75): ([I(s)]_1, x^n)"""
        ys, x = _fr(ys), _fr(x)
        out, xp = g1_empty(1), fr_empty(1)
        _chk(lib().kzg_hip_check_proof_multi_interpolation(self.h, _p(ys), ys.shape[0], _p(x), _p(out), _p(xp)))
        return out[0], xp[0]

    def toeplitz_part2(self, toeplitz_coeffs, x_ext_fft):
        toeplitz_coeffs, x_ext_fft = _fr(toeplitz_coeffs), _g1(x_ext_fft)
        if toeplitz_coeffs.shape[0] != x_ext_fft.shape[0]:
            raise KzgPanic(ERR_LEN_MISMATCH, "expected toeplitz coeffs to match xExtFFT length")
        out = g1_empty(x_ext_fft.shape[0])
        _chk(lib().kzg_hip_toeplitz_part2(self.h, _p(toeplitz_coeffs), _p(x_ext_fft), x_ext_fft.shape[0], _p(out)))
        return out

    def toeplitz_part3(self, h_ext_fft):
        h_ext_fft = _g1(h_ext_fft)
        out = g1_empty(h_ext_fft.shape[0] // 2)
        _chk(lib().kzg_hip_toeplitz_part3(self.h, _p(h_ext_fft), h_ext_fft.shape[0], _p(out)))
        return out


class FK20SingleSettings:
    """kzg.FK20SingleSettings (kzg.go:38-64; fk20_single.go:122-196)"""

    def __init__(self, ks, n2):
        h = C.c_void_p()
        _chk(lib().kzg_hip_fk20_single_settings_new(ks.h, n2, C.byref(h)))
        self.h, self.ks, self.n2 = h, ks, n2

    def close(self):
        if getattr(self, "h", None):
            lib().kzg_hip_fk20_single_settings_free(self.h)
            self.h = None

    def x_ext_fft(self):
        out = g1_empty(self.n2)
        _chk(lib().kzg_hip_fk20_single_x_ext_fft(self.h, _p(out)))
        return out

    def fk20_single(self, poly):
        poly = _fr(poly)
        out = g1_empty(poly.shape[0])
        _chk(lib().kzg_hip_fk20_single(self.h, _p(poly), poly.shape[0], _p(out)))
        return out

    def fk20_single_batch(self, polys):
        """FK20Single on each row of polys (batch, n, 4) -> (batch, n, 3, 6)"""
        polys = np.ascontiguousarray(polys, dtype=np.uint64)
        b, n = polys.shape[0], polys.shape[1]
        out = np.zeros((b, n, 3, 6), dtype=np.uint64)
        _chk(lib().kzg_hip_fk20_single_batch(self.h, _p(polys), n, b, _p(out)))
        return out

    def fk20_single_da_optimized(self, poly):
        poly = _fr(poly)
        out = g1_empty(poly.shape[0])
        _chk(lib().kzg_hip_fk20_single_da_optimized(self.h, _p(poly), poly.shape[0], _p(out)))
        return out

    def da_using_fk20(self, poly):
        poly = _fr(poly)
        out = g1_empty(2 * poly.shape[0])
        _chk(lib().kzg_hip_da_using_fk20(self.h, _p(poly), poly.shape[0], _p(out)))
        return out

    def da_using_fk20_batch(self, polys):
        polys = np.ascontiguousarray(polys, dtype=np.uint64)
        b, n = polys.shape[0], polys.shape[1]
        out = np.zeros((b, 2 * n, 3, 6), dtype=np.uint64)
        _chk(lib().kzg_hip_da_using_fk20_batch(self.h, _p(polys), n, b, _p(out)))
        return out


class FK20MultiSettings:
    """kzg.FK20MultiSettings (kzg.go:66-116; fk20_multi.go:25-133)"""

    def __init__(self, ks, n2, chunk_len):
        h = C.c_void_p()
        _chk(lib().kzg_hip_fk20_multi_settings_new(ks.h, n2, chunk_len, C.byref(h)))
        self.h, self.ks, self.n2, self.chunk_len = h, ks, n2, chunk_len

    def close(self):
        if getattr(self, "h", None):
            lib().kzg_hip_fk20_multi_settings_free(self.h)
            self.h = None

    def fk20_multi(self, poly):
        poly = _fr(poly)
        out = g1_empty(poly.shape[0] // self.chunk_len)
        _chk(lib().kzg_hip_fk20_multi(self.h, _p(poly), poly.shape[0], _p(out)))
        return out

    def fk20_multi_da_optimized(self, poly):
        poly = _fr(poly)
        out = g1_empty(poly.shape[0] // self.chunk_len)
        _chk(lib().kzg_hip_fk20_multi_da_optimized(self.h, _p(poly), poly.shape[0], _p(out)))
        return out

    def da_using_fk20_multi(self, poly):
        poly = _fr(poly)
        out = g1_empty(2 * poly.shape[0] // self.chunk_len)
        _chk(lib().kzg_hip_da_using_fk20_multi(self.h, _p(poly), poly.shape[0], _p(out)))
        return out

    def da_using_fk20_multi_batch(self, polys):
        """DAUsingFK20Multi on each row of polys (batch, n, 4) -> (batch, 2n / chunk_len, 3, 6)"""
        polys = np.ascontiguousarray(polys, dtype=np.uint64)
        b, n = polys.shape[0], polys.shape[1]
        out = np.zeros((b, 2 * n // self.chunk_len, 3, 6), dtype=np.uint64)
        _chk(lib().kzg_hip_da_using_fk20_multi_batch(self.h, _p(polys), n, b, _p(out)))
        return out


class _Borrowed:
    """a settings object owned by a MultiKZGSettings: same methods, close() is a no-op"""

    def close(self):
        self.h = None


class _BorrowedFFT(_Borrowed, FFTSettings):
    def __init__(self, h, max_scale, device):
        self.h, self.max_scale, self.max_width, self.device = C.c_void_p(h), max_scale, 1 << max_scale, device


class _BorrowedKZG(_Borrowed, KZGSettings):
    def __init__(self, h, fs, n_setup):
        self.h, self.fs, self.n_setup = C.c_void_p(h), fs, n_setup


class MultiKZGSettings:
    """NewFFTSettings(max_scale) + NewKZGSettings(fs, secret_g1) on EVERY device of `devices`, behind one handle
    (kzg_hip_multi_*, include/kzg_hip.h): what the Go shim exposes as kzg.NewMultiKZGSettings.  A list may repeat a device."""

    def __init__(self, devices, max_scale, secret_g1):
        secret_g1 = _g1(secret_g1)
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _chk(lib().kzg_hip_multi_settings_new(devs, len(devices), max_scale, _p(secret_g1), secret_g1.shape[0], C.byref(h)))
        self.h, self.devices, self.max_scale, self.n_setup = h, list(devices), max_scale, secret_g1.shape[0]

    def close(self):
        if getattr(self, "h", None):
            lib().kzg_hip_multi_settings_free(self.h)
            self.h = None

    @property
    def transport(self):
        """"rccl" (ncclAllGather between distinct devices), "peer-copy" (hipMemcpyPeerAsync: a repeated device, or no librccl) or
        "host-staged" (device -> pinned host -> device: what is left when the other two fail the creation-time self-test)"""
        return lib().kzg_hip_multi_transport(self.h).decode()

    @property
    def transport_self_test(self):
        """outcome of the exchange test the constructor ran: "ok: <transport>, <entries> entries, ..." """
        return lib().kzg_hip_multi_transport_check(self.h).decode()

    @property
    def transport_note(self):
        return lib().kzg_hip_multi_transport_note(self.h).decode()

    @property
    def exchanges(self):
        return lib().kzg_hip_multi_exchanges(self.h)

    def fft_settings(self, i):
        return _BorrowedFFT(lib().kzg_hip_multi_fft(self.h, i), self.max_scale, lib().kzg_hip_multi_device(self.h, i))

    def kzg_settings(self, i):
        """entry i's KZGSettings (borrowed): every single-device method on a chosen device"""
        return _BorrowedKZG(lib().kzg_hip_multi_kzg(self.h, i), self.fft_settings(i), self.n_setup)

    def set_fft_sharding(self, mode):
        """one-polynomial FK20 calls: "gather" (all-gather of hExtFFT, transforms on the first device), "sharded" (both G1 transforms
        sharded too: five all-gathers) or None (default: sharded from 4 devices on)"""
        _chk(lib().kzg_hip_multi_set_fft_sharding(self.h, {None: -1, "gather": 0, "sharded": 1}[mode]))

    def set_table_budget_gb(self, gb):
        _chk(lib().kzg_hip_multi_set_table_budget_gb(self.h, float(gb)))

    def commit_to_poly_batch(self, coeffs):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        b, n = coeffs.shape[0], coeffs.shape[1]
        out = g1_empty(b)
        _chk(lib().kzg_hip_multi_commit_to_poly_batch(self.h, _p(coeffs), n, b, _p(out)))
        return out

    def fft_batch(self, vals, inv=False):
        """FFT (fft_fr.go:55-74) on every row, rows divided among the devices"""
        vals = np.ascontiguousarray(vals, dtype=np.uint64)
        out = np.zeros_like(vals)
        _chk(lib().kzg_hip_multi_fft_fr_batch(self.h, _p(vals), vals.shape[1], vals.shape[0], int(inv), _p(out)), error_ok=True)
        return out

    def fft_g1_batch(self, vals, inv=False):
        """FFTG1 (fft_g1.go:58-94) on every row, rows divided among the devices"""
        vals = np.ascontiguousarray(vals, dtype=np.uint64)
        out = np.zeros_like(vals)
        _chk(lib().kzg_hip_multi_fft_g1_batch(self.h, _p(vals), vals.shape[1], vals.shape[0], int(inv), _p(out)), error_ok=True)
        return out

    def das_fft_extension_batch(self, vals):
        """DASFFTExtension (das_extension.go:71-84) on every row; returns the odd values"""
        vals = np.ascontiguousarray(vals, dtype=np.uint64).copy()
        _chk(lib().kzg_hip_multi_das_fft_extension_batch(self.h, _p(vals), vals.shape[1], vals.shape[0]))
        return vals

    def compute_proof_single_batch(self, polys, xs):
        polys = np.ascontiguousarray(polys, dtype=np.uint64)
        xs = np.ascontiguousarray(xs, dtype=np.uint64)
        b, n = polys.shape[0], polys.shape[1]
        out = g1_empty(b)
        _chk(lib().kzg_hip_multi_compute_proof_single_batch(self.h, _p(polys), n, b, _p(xs), _p(out)))
        return out


class MultiEthSettings:
    """package eth (eth/globals.go:39-72) on every device of a MultiKZGSettings: batches divided among the devices"""

    def __init__(self, mks, setup_g1_lagrange):
        lag = _g1(setup_g1_lagrange)
        h = C.c_void_p()
        _chk(lib().kzg_hip_multi_eth_settings_new(mks.h, _p(lag), lag.shape[0], C.byref(h)))
        self.h, self.mks, self.n = h, mks, lag.shape[0]

    def close(self):
        if getattr(self, "h", None):
            lib().kzg_hip_multi_eth_settings_free(self.h)
            self.h = None

    def blob_to_kzg_commitment_batch(self, blobs):
        blobs = np.ascontiguousarray(blobs, dtype=np.uint8).reshape(-1, self.n, 32)
        b = blobs.shape[0]
        out, ok = np.zeros((b, 48), dtype=np.uint8), np.zeros(b, dtype=np.uint8)
        _chk(lib().kzg_hip_multi_eth_blob_to_kzg_commitment_batch(self.h, _p(blobs), b, _p(out), _p(ok)))
        return out, ok.astype(bool)

    def compute_kzg_proof_batch(self, polynomials, zs):
        polys = np.ascontiguousarray(polynomials, dtype=np.uint64).reshape(-1, self.n, 4)
        zs = np.ascontiguousarray(zs, dtype=np.uint64).reshape(-1, 4)
        b = polys.shape[0]
        if zs.shape[0] != b:
            raise KzgError(ERR_LEN_MISMATCH, "one z per polynomial")
        out, ys, ok = np.zeros((b, 48), dtype=np.uint8), fr_empty(b), np.zeros(b, dtype=np.uint8)
        _chk(lib().kzg_hip_multi_eth_compute_kzg_proof_batch(self.h, _p(polys), self.n, b, _p(zs), _p(out), _p(ys), _p(ok)))
        return out, ys, ok.astype(bool)


class MultiFK20SingleSettings:
    """NewFK20SingleSettings (kzg.go:43-64) on every device of a MultiKZGSettings"""

    def __init__(self, mks, n2):
        h = C.c_void_p()
        _chk(lib().kzg_hip_multi_fk20_single_settings_new(mks.h, n2, C.byref(h)))
        self.h, self.mks, self.n2 = h, mks, n2

    def close(self):
        if getattr(self, "h", None):
            lib().kzg_hip_multi_fk20_single_settings_free(self.h)
            self.h = None

    def da_using_fk20(self, poly):
        """ONE polynomial over all devices (Toeplitz stage by output position + all-gather(s))"""
        poly = _fr(poly)
        out = g1_empty(2 * poly.shape[0])
        _chk(lib().kzg_hip_multi_da_using_fk20(self.h, _p(poly), poly.shape[0], _p(out)), error_ok=True)
        return out

    def da_using_fk20_batch(self, polys):
        """polynomials divided among the devices"""
        polys = np.ascontiguousarray(polys, dtype=np.uint64)
        b, n = polys.shape[0], polys.shape[1]
        out = np.zeros((b, 2 * n, 3, 6), dtype=np.uint64)
        _chk(lib().kzg_hip_multi_da_using_fk20_batch(self.h, _p(polys), n, b, _p(out)))
        return out


class MultiFK20MultiSettings:
    """NewFK20MultiSettings (kzg.go:73-116) on every device of a MultiKZGSettings"""

    def __init__(self, mks, n2, chunk_len):
        h = C.c_void_p()
        _chk(lib().kzg_hip_multi_fk20_multi_settings_new(mks.h, n2, chunk_len, C.byref(h)))
        self.h, self.mks, self.n2, self.chunk_len = h, mks, n2, chunk_len

    def close(self):
        if getattr(self, "h", None):
            lib().kzg_hip_multi_fk20_multi_settings_free(self.h)
            self.h = None

    def da_using_fk20_multi(self, poly):
        """ONE polynomial over all devices (fk20_multi.go:113-133)"""
        poly = _fr(poly)
        out = g1_empty(2 * poly.shape[0] // self.chunk_len)
        _chk(lib().kzg_hip_multi_da_using_fk20_multi(self.h, _p(poly), poly.shape[0], _p(out)), error_ok=True)
        return out

    def da_using_fk20_multi_batch(self, polys):
        polys = np.ascontiguousarray(polys, dtype=np.uint64)
        b, n = polys.shape[0], polys.shape[1]
        out = np.zeros((b, 2 * n // self.chunk_len, 3, 6), dtype=np.uint64)
        _chk(lib().kzg_hip_multi_da_using_fk20_multi_batch(self.h, _p(polys), n, b, _p(out)))
        return out


class EthSettings:
    """eth/ package state (eth/globals.go:39-72): bit-reversed Lagrange setup + bit-reversed domain, device resident."""

    def __init__(self, fs, setup_g1_lagrange):
        lag = _g1(setup_g1_lagrange)
        h = C.c_void_p()
        _chk(lib().kzg_hip_eth_settings_new(fs.h, _p(lag), lag.shape[0], C.byref(h)))
        self.h, self.fs, self.n = h, fs, lag.shape[0]

    def close(self):
        if getattr(self, "h", None):
            lib().kzg_hip_eth_settings_free(self.h)
            self.h = None

    def blob_to_kzg_commitment_batch(self, blobs):
        """eth.BlobToKZGCommitment (eth/eth.go:145-151) on (batch, n, 32) uint8 little-endian blobs -> ((batch, 48) uint8, ok flags)"""
        blobs = np.ascontiguousarray(blobs, dtype=np.uint8).reshape(-1, self.n, 32)
        b = blobs.shape[0]
        out, ok = np.zeros((b, 48), dtype=np.uint8), np.zeros(b, dtype=np.uint8)
        _chk(lib().kzg_hip_eth_blob_to_kzg_commitment_batch(self.h, _p(blobs), b, _p(out), _p(ok)))
        return out, ok.astype(bool)

    def blob_to_kzg_commitment(self, blob):
        out, ok = self.blob_to_kzg_commitment_batch(np.asarray(blob, dtype=np.uint8).reshape(1, self.n, 32))
        return out[0], bool(ok[0])

    def compute_kzg_proof(self, polynomial, z):
        """eth.ComputeKZGProof (eth/helpers.go:179-203): returns (proof48, y); raises KzgError like the reference's errors"""
        poly, z = _fr(polynomial), _fr(z)
        out, y = np.zeros(48, dtype=np.uint8), fr_empty(1)
        st = lib().kzg_hip_eth_compute_kzg_proof(self.h, _p(poly), poly.shape[0], _p(z), _p(out), _p(y))
        if st == ERR_LEN_MISMATCH:
            raise KzgError(st, "polynomial has invalid length")
        if st == ERR_BAD_ARG:
            raise KzgError(st, "invalid z challenge")
        _chk(st)
        return out, y[0]

    def compute_kzg_proof_batch(self, polynomials, zs):
        """eth.ComputeKZGProof over rows (kzg_hip_eth_compute_kzg_proof_batch): ((batch, 48) uint8 proofs, (batch, 4) ys, ok flags); ok[b] is False
        where zs[b] lies in the domain (the reference's "invalid z challenge")"""
        polys = np.ascontiguousarray(polynomials, dtype=np.uint64).reshape(-1, self.n, 4)
        zs = np.ascontiguousarray(zs, dtype=np.uint64).reshape(-1, 4)
        b = polys.shape[0]
        if zs.shape[0] != b:
            raise KzgError(ERR_LEN_MISMATCH, "one z per polynomial")
        out, ys, ok = np.zeros((b, 48), dtype=np.uint8), fr_empty(b), np.zeros(b, dtype=np.uint8)
        _chk(lib().kzg_hip_eth_compute_kzg_proof_batch(self.h, _p(polys), self.n, b, _p(zs), _p(out), _p(ys), _p(ok)))
        return out, ys, ok.astype(bool)

    def evaluate_polynomial_in_evaluation_form(self, polynomial, x):
        """eth.EvaluatePolynomialInEvaluationForm (eth/helpers.go:207-211)"""
        poly, x, y = _fr(polynomial), _fr(x), fr_empty(1)
        _chk(lib().kzg_hip_eth_evaluate_polynomial_in_evaluation_form(self.h, _p(poly), poly.shape[0], _p(x), _p(y)))
        return y[0]

    def compute_aggregate_kzg_proof(self, blobs):
        """eth.ComputeAggregateKZGProof (eth/eth.go:175-182) on (batch, n, 32) uint8 blobs (batch may be 0): (proof48, (batch, 48) commitments);
        KzgError "could not convert blobs to polynomials" / "invalid z challenge" like the reference"""
        blobs = np.ascontiguousarray(blobs, dtype=np.uint8).reshape(-1, self.n, 32)
        b = blobs.shape[0]
        proof, comm = np.zeros(48, dtype=np.uint8), np.zeros((b, 48), dtype=np.uint8)
        st = lib().kzg_hip_eth_compute_aggregate_kzg_proof(self.h, _p(blobs), b, _p(proof), _p(comm))
        if st == ERR_BAD_BLOB:
            raise KzgError(st, "could not convert blobs to polynomials")
        if st == ERR_BAD_ARG:
            raise KzgError(st, "invalid z challenge")
        _chk(st)
        return proof, comm

    def compute_aggregated_poly_and_commitment(self, blobs, commitments):
        """The prover-side pieces of eth.VerifyAggregateKZGProof (eth/eth.go:155-172): (aggregated polynomial (n, 4), aggregated commitment
        G1 image (18,), z (4,), y (4,)); the pairing of VerifyKZGProofFromPoints stays with the caller"""
        blobs = np.ascontiguousarray(blobs, dtype=np.uint8).reshape(-1, self.n, 32)
        comm = np.ascontiguousarray(commitments, dtype=np.uint8).reshape(-1, 48)
        b = blobs.shape[0]
        if comm.shape[0] != b:
            raise KzgError(ERR_LEN_MISMATCH, "one commitment per blob")
        poly, c, z, y = fr_empty(self.n), g1_empty(1), fr_empty(1), fr_empty(1)
        st = lib().kzg_hip_eth_compute_aggregated_poly_and_commitment(self.h, _p(blobs), _p(comm), b, _p(poly), _p(c), _p(z), _p(y))
        if st == ERR_BAD_BLOB:
            raise KzgError(st, "could not convert blobs to polynomials")
        if st == ERR_BAD_POINT:
            raise KzgError(st, "invalid commitment")
        _chk(st)
        return poly, c[0], z[0], y[0]

    def bench_drop_in_proof(self, polys, threads, calls):
        """`threads` native host threads x `calls` blocking eth.ComputeKZGProof calls: (calls per second, each thread's last proof)"""
        polys = np.ascontiguousarray(polys, dtype=np.uint64)
        out, secs = np.zeros((threads, 48), dtype=np.uint8), C.c_double(0)
        _chk(lib().kzg_hip_bench_drop_in_eth_proof(self.h, _p(polys), polys.shape[1], polys.shape[0], threads, calls, _p(out), C.byref(secs)))
        return threads * calls / secs.value, out

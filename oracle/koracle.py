"""ctypes binding of oracle/libkzg_oracle.so -- TEST INFRASTRUCTURE ONLY (see kzg_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All buffers are numpy uint64 arrays holding the Kilic memory images:
  Fr  -> shape (n, 4)      Montgomery limbs, little-endian
  G1  -> shape (n, 3, 6)   Jacobian (X, Y, Z), Montgomery limbs, inf <=> Z == 0
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkzg_oracle.so")

R_MOD = 52435875175126190479447740508185965837690552500527637822603658699938581184513  # bls/globals.go:9

OK, ERR_TOO_WIDE, ERR_NOT_POW2, ERR_LEN_MISMATCH, ERR_UPPER_HALF, ERR_BAD_ARG, ERR_BAD_POINT = range(7)


def build(force=False):
    src = os.path.join(_HERE, "kzg_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkzg_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int, C.c_uint32
        sig = {
            "ko_fr_from_le32": (i32, [vp, vp]), "ko_fr_to_le32": (None, [vp, vp]), "ko_fr_from_u64": (None, [vp, u64]),
            "ko_fr_mul": (None, [vp, vp, vp]), "ko_fr_add": (None, [vp, vp, vp]), "ko_fr_sub": (None, [vp, vp, vp]),
            "ko_fr_inv": (None, [vp, vp]),
            "ko_g1_generator": (None, [vp]), "ko_g1_zero": (None, [vp]), "ko_g1_add": (None, [vp, vp, vp]),
            "ko_g1_sub": (None, [vp, vp, vp]), "ko_g1_dbl": (None, [vp, vp]), "ko_g1_mul": (None, [vp, vp, vp]),
            "ko_g1_equal": (i32, [vp, vp]), "ko_g1_affine_batch": (None, [vp, vp, u64]),
            "ko_g1_to_compressed_batch": (None, [vp, vp, u64]), "ko_g1_from_compressed_batch": (i32, [vp, vp, u64]),
            "ko_lincomb_g1": (i32, [vp, vp, vp, u64]),
            "ko_scale2_root_of_unity": (None, [vp, C.c_uint]),
            "ko_fft_settings_new": (vp, [C.c_uint]), "ko_fft_settings_free": (None, [vp]),
            "ko_fft_max_width": (u64, [vp]), "ko_fft_expanded_roots": (vp, [vp]), "ko_fft_reverse_roots": (vp, [vp]),
            "ko_inplace_fft": (i32, [vp, vp, vp, u64, i32]), "ko_fft_fr": (i32, [vp, vp, u64, i32, vp, vp]),
            "ko_fft_g1": (i32, [vp, vp, u64, i32, vp]), "ko_das_fft_extension": (i32, [vp, vp, u64]),
            "ko_reverse_bits_limited": (u32, [u32, u32]),
            "ko_reverse_bit_order_fr": (None, [vp, u64]), "ko_reverse_bit_order_g1": (None, [vp, u64]),
            "ko_generate_testing_setup_g1": (None, [vp, u64, vp]),
            "ko_commit_to_poly": (i32, [vp, u64, vp, u64, vp]),
            "ko_compute_proof_single": (i32, [vp, u64, vp, u64, u64, vp]),
            "ko_poly_quotient_linear": (None, [vp, u64, u64, vp]),
            "ko_compute_proof_multi": (i32, [vp, u64, vp, u64, u64, u64, vp]),
            "ko_check_proof_multi_interpolation": (i32, [vp, vp, u64, vp, u64, vp, vp, vp]),
            "ko_toeplitz_part2": (i32, [vp, vp, vp, u64, vp]), "ko_toeplitz_part3": (i32, [vp, vp, u64, vp]),
            "ko_toeplitz_coeffs_step_strided": (None, [vp, u64, u64, u64, vp]),
            "ko_fk20_single_new": (vp, [vp, vp, u64, u64, vp]), "ko_fk20_single_free": (None, [vp]),
            "ko_fk20_single_x_ext_fft": (vp, [vp]),
            "ko_fk20_single": (i32, [vp, vp, u64, vp]), "ko_fk20_single_da_optimized": (i32, [vp, vp, u64, vp]),
            "ko_da_using_fk20": (i32, [vp, vp, u64, vp]),
            "ko_fk20_multi_new": (vp, [vp, vp, u64, u64, u64, vp]), "ko_fk20_multi_free": (None, [vp]),
            "ko_fk20_multi_file": (vp, [vp, u64]),
            "ko_fk20_multi": (i32, [vp, vp, u64, vp]), "ko_fk20_multi_da_optimized": (i32, [vp, vp, u64, vp]),
            "ko_da_using_fk20_multi": (i32, [vp, vp, u64, vp]),
            "ko_zero_poly_via_multiplication": (i32, [vp, vp, u64, u64, vp, vp]),
            "ko_recover_poly_from_samples": (i32, [vp, vp, vp, u64, vp]),
            "ko_synthetic_blob": (None, [u64, u64, vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(_lib, name)
            f.restype, f.argtypes = res, args
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fr_empty(n):
    return np.zeros((n, 4), dtype=np.uint64)


def g1_empty(n):
    return np.zeros((n, 3, 6), dtype=np.uint64)


# ---------------- scalars ----------------
def fr_from_ints(vals):
    """list of Python ints (reduced mod r) -> Montgomery images.  (bls.SetFr / AsFr, bignum_kilic.go:25-31,61-65)"""
    vals = [int(v) % R_MOD for v in vals]
    raw = b"".join(v.to_bytes(32, "little") for v in vals)
    src = np.frombuffer(raw, dtype=np.uint8).reshape(len(vals), 32).copy()
    out = fr_empty(len(vals))
    L = lib()
    for i in range(len(vals)):
        assert L.ko_fr_from_le32(_p(out[i]), _p(src[i])) == 1
    return out


def fr_to_ints(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    L = lib()
    buf = np.zeros(32, dtype=np.uint8)
    res = []
    for i in range(a.shape[0]):
        L.ko_fr_to_le32(_p(buf), _p(a[i]))
        res.append(int.from_bytes(buf.tobytes(), "little"))
    return res


def fr_from_le32(b):
    """FrFrom32 (bls/bignum_kilic.go:33-44): returns (ok, image)"""
    src = np.frombuffer(bytes(b), dtype=np.uint8).copy()
    out = fr_empty(1)
    ok = lib().ko_fr_from_le32(_p(out), _p(src))
    return bool(ok), out[0]


def synthetic_blob(seed, n=4096):
    out = fr_empty(n)
    lib().ko_synthetic_blob(seed, n, _p(out))
    return out


# ---------------- points ----------------
def g1_generator():
    o = g1_empty(1)
    lib().ko_g1_generator(_p(o))
    return o[0]


def g1_zero(n=1):
    o = g1_empty(n)
    for i in range(n):
        lib().ko_g1_zero(_p(o[i]))
    return o


def g1_mul(p, k):
    o = g1_empty(1)
    lib().ko_g1_mul(_p(o), _p(np.ascontiguousarray(p)), _p(np.ascontiguousarray(k)))
    return o[0]


def g1_add(a, b):
    o = g1_empty(1)
    lib().ko_g1_add(_p(o), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)))
    return o[0]


def g1_sub(a, b):
    o = g1_empty(1)
    lib().ko_g1_sub(_p(o), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)))
    return o[0]


def g1_dbl(a):
    o = g1_empty(1)
    lib().ko_g1_dbl(_p(o), _p(np.ascontiguousarray(a)))
    return o[0]


def g1_equal(a, b):
    return bool(lib().ko_g1_equal(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b))))


def g1_affine(pts):
    pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 3, 6)
    o = g1_empty(pts.shape[0])
    lib().ko_g1_affine_batch(_p(o), _p(pts), pts.shape[0])
    return o


def g1_compress(pts):
    pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 3, 6)
    o = np.zeros((pts.shape[0], 48), dtype=np.uint8)
    lib().ko_g1_to_compressed_batch(_p(o), _p(pts), pts.shape[0])
    return o


def g1_decompress(b):
    b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 48)
    o = g1_empty(b.shape[0])
    st = lib().ko_g1_from_compressed_batch(_p(o), _p(b), b.shape[0])
    if st:
        raise ValueError("bad compressed G1 (status %d)" % st)
    return o


def lincomb_g1(points, scalars):
    """bls.LinCombG1 (bls/bls_kilic.go:132-150)"""
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 3, 6)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    if points.shape[0] != scalars.shape[0]:
        raise ValueError("got LinCombG1 numbers/factors length mismatch")
    o = g1_empty(1)
    lib().ko_lincomb_g1(_p(o), _p(points), _p(scalars), points.shape[0])
    return o[0]


def generate_testing_setup_g1(secret_int, n):
    """GenerateTestingSetup (setup.go:9-26), G1 half"""
    s = fr_from_ints([secret_int])
    o = g1_empty(n)
    lib().ko_generate_testing_setup_g1(_p(s), n, _p(o))
    return o


def reverse_bits_limited(length, v):
    return lib().ko_reverse_bits_limited(length, v)


def reverse_bit_order(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    if a.ndim == 2:
        lib().ko_reverse_bit_order_fr(_p(a), a.shape[0])
    else:
        lib().ko_reverse_bit_order_g1(_p(a), a.shape[0])
    return a


class OracleError(Exception):
    def __init__(self, status):
        super().__init__("oracle status %d" % status)
        self.status = status


def _chk(st):
    if st:
        raise OracleError(st)


class FFTSettings:
    """fft.go:34-61"""

    def __init__(self, max_scale):
        self.h = lib().ko_fft_settings_new(max_scale)
        self.max_width = 1 << max_scale

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ko_fft_settings_free(self.h)
            self.h = None

    def expanded_roots(self):
        ptr = lib().ko_fft_expanded_roots(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(self.max_width + 1, 4)).copy()

    def reverse_roots(self):
        ptr = lib().ko_fft_reverse_roots(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(self.max_width + 1, 4)).copy()

    def fft(self, vals, inv=False):
        """FFT (fft_fr.go:55-74): pads to the next power of two"""
        vals = np.ascontiguousarray(vals, dtype=np.uint64).reshape(-1, 4)
        n = vals.shape[0]
        np2 = 1 if n == 0 else 1 << (n - 1).bit_length()
        out = fr_empty(np2)
        on = C.c_uint64(0)
        _chk(lib().ko_fft_fr(self.h, _p(vals), n, int(inv), _p(out), C.byref(on)))
        return out

    def inplace_fft(self, vals, inv=False):
        """InplaceFFT (fft_fr.go:76-105): no padding"""
        vals = np.ascontiguousarray(vals, dtype=np.uint64).reshape(-1, 4)
        out = fr_empty(vals.shape[0])
        _chk(lib().ko_inplace_fft(self.h, _p(vals), _p(out), vals.shape[0], int(inv)))
        return out

    def fft_g1(self, vals, inv=False):
        """FFTG1 (fft_g1.go:58-94)"""
        vals = np.ascontiguousarray(vals, dtype=np.uint64).reshape(-1, 3, 6)
        out = g1_empty(vals.shape[0])
        _chk(lib().ko_fft_g1(self.h, _p(vals), vals.shape[0], int(inv), _p(out)))
        return out

    def zero_poly_via_multiplication(self, missing_indices, length):
        """ZeroPolyViaMultiplication (zero_poly.go:116-217): (zero_eval, zero_poly), `length` entries each"""
        mi = np.ascontiguousarray(missing_indices, dtype=np.uint64)
        ze, zp = fr_empty(length), fr_empty(length)
        _chk(lib().ko_zero_poly_via_multiplication(self.h, _p(mi), mi.shape[0], length, _p(ze), _p(zp)))
        return ze, zp

    def recover_poly_from_samples(self, samples, present):
        """RecoverPolyFromSamples (recover_from_samples.go:42-109); present[i] False <=> samples[i] is nil"""
        samples = np.ascontiguousarray(samples, dtype=np.uint64).reshape(-1, 4)
        present = np.ascontiguousarray(present, dtype=np.uint8)
        out = fr_empty(samples.shape[0])
        _chk(lib().ko_recover_poly_from_samples(self.h, _p(samples), _p(present), samples.shape[0], _p(out)))
        return out

    def das_fft_extension(self, vals):
        """DASFFTExtension (das_extension.go:71-84); returns the odd values (the reference works in place)"""
        vals = np.ascontiguousarray(vals, dtype=np.uint64).reshape(-1, 4).copy()
        _chk(lib().ko_das_fft_extension(self.h, _p(vals), vals.shape[0]))
        return vals


class KZGSettings:
    """kzg.go:11-36 (prover side: SecretG1 only)"""

    def __init__(self, fs, secret_g1):
        secret_g1 = np.ascontiguousarray(secret_g1, dtype=np.uint64).reshape(-1, 3, 6)
        if secret_g1.shape[0] < fs.max_width:
            raise ValueError("expected more values for secrets")
        self.fs, self.secret_g1 = fs, secret_g1

    def commit_to_poly(self, coeffs):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        o = g1_empty(1)
        _chk(lib().ko_commit_to_poly(_p(self.secret_g1), self.secret_g1.shape[0], _p(coeffs), coeffs.shape[0], _p(o)))
        return o[0]

    def compute_proof_single(self, poly, x):
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        o = g1_empty(1)
        _chk(lib().ko_compute_proof_single(_p(self.secret_g1), self.secret_g1.shape[0], _p(poly), poly.shape[0], x, _p(o)))
        return o[0]

    def compute_proof_multi(self, poly, x, n):
        """ComputeProofMulti (kzg_multi_proofs.go:13-43), including the reference's xPowN quirk"""
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        o = g1_empty(1)
        _chk(lib().ko_compute_proof_multi(_p(self.secret_g1), self.secret_g1.shape[0], _p(poly), poly.shape[0], x, n, _p(o)))
        return o[0]

    def check_proof_multi_interpolation(self, ys, x):
        """prover-side half of CheckProofMulti (kzg_multi_proofs.go:47-75): ([I(s)]_1, x^n)"""
        ys = np.ascontiguousarray(ys, dtype=np.uint64).reshape(-1, 4)
        x = np.ascontiguousarray(x, dtype=np.uint64).reshape(1, 4)
        o, xp = g1_empty(1), fr_empty(1)
        _chk(lib().ko_check_proof_multi_interpolation(self.fs.h, _p(self.secret_g1), self.secret_g1.shape[0], _p(ys), ys.shape[0], _p(x), _p(o), _p(xp)))
        return o[0], xp[0]

    def toeplitz_part2(self, coeffs, x_ext_fft):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        x_ext_fft = np.ascontiguousarray(x_ext_fft, dtype=np.uint64).reshape(-1, 3, 6)
        if coeffs.shape[0] != x_ext_fft.shape[0]:
            raise ValueError("expected toeplitz coeffs to match xExtFFT length")
        o = g1_empty(coeffs.shape[0])
        _chk(lib().ko_toeplitz_part2(self.fs.h, _p(coeffs), _p(x_ext_fft), coeffs.shape[0], _p(o)))
        return o

    def toeplitz_part3(self, h_ext_fft):
        h_ext_fft = np.ascontiguousarray(h_ext_fft, dtype=np.uint64).reshape(-1, 3, 6)
        o = g1_empty(h_ext_fft.shape[0])
        _chk(lib().ko_toeplitz_part3(self.fs.h, _p(h_ext_fft), h_ext_fft.shape[0], _p(o)))
        return o[: h_ext_fft.shape[0] // 2]


def poly_quotient_linear(poly, x):
    poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
    o = fr_empty(poly.shape[0] - 1)
    lib().ko_poly_quotient_linear(_p(poly), poly.shape[0], x, _p(o))
    return o


def toeplitz_coeffs_step_strided(poly, offset, stride):
    poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
    n = poly.shape[0]
    o = fr_empty(2 * (n // stride))
    lib().ko_toeplitz_coeffs_step_strided(_p(poly), n, offset, stride, _p(o))
    return o


class FK20SingleSettings:
    """kzg.go:38-64 + fk20_single.go:122-196"""

    def __init__(self, ks, n2):
        st = C.c_int(0)
        self.ks, self.n2 = ks, n2
        self.h = lib().ko_fk20_single_new(ks.fs.h, _p(ks.secret_g1), ks.secret_g1.shape[0], n2, C.byref(st))
        _chk(st.value)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ko_fk20_single_free(self.h)
            self.h = None

    def x_ext_fft(self):
        ptr = lib().ko_fk20_single_x_ext_fft(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(self.n2, 3, 6)).copy()

    def fk20_single(self, poly):
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        o = g1_empty(poly.shape[0])
        _chk(lib().ko_fk20_single(self.h, _p(poly), poly.shape[0], _p(o)))
        return o

    def fk20_single_da_optimized(self, poly):
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        o = g1_empty(poly.shape[0])
        _chk(lib().ko_fk20_single_da_optimized(self.h, _p(poly), poly.shape[0], _p(o)))
        return o

    def da_using_fk20(self, poly):
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        o = g1_empty(2 * poly.shape[0])
        _chk(lib().ko_da_using_fk20(self.h, _p(poly), poly.shape[0], _p(o)))
        return o


class FK20MultiSettings:
    """kzg.go:66-116 + fk20_multi.go:25-133"""

    def __init__(self, ks, n2, chunk_len):
        st = C.c_int(0)
        self.ks, self.n2, self.chunk_len = ks, n2, chunk_len
        self.h = lib().ko_fk20_multi_new(ks.fs.h, _p(ks.secret_g1), ks.secret_g1.shape[0], n2, chunk_len, C.byref(st))
        _chk(st.value)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ko_fk20_multi_free(self.h)
            self.h = None

    def file(self, i):
        k2 = self.n2 // self.chunk_len
        ptr = lib().ko_fk20_multi_file(self.h, i)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(k2, 3, 6)).copy()

    def fk20_multi(self, poly):
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        o = g1_empty(poly.shape[0] // self.chunk_len)
        _chk(lib().ko_fk20_multi(self.h, _p(poly), poly.shape[0], _p(o)))
        return o

    def fk20_multi_da_optimized(self, poly):
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        o = g1_empty(poly.shape[0] // self.chunk_len)
        _chk(lib().ko_fk20_multi_da_optimized(self.h, _p(poly), poly.shape[0], _p(o)))
        return o

    def da_using_fk20_multi(self, poly):
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        o = g1_empty(2 * poly.shape[0] // self.chunk_len)
        _chk(lib().ko_da_using_fk20_multi(self.h, _p(poly), poly.shape[0], _p(o)))
        return o

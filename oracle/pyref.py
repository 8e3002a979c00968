"""pyref.py -- TEST INFRASTRUCTURE ONLY.  Second, independent restatement of the hot path in Python
big integers, used to cross-check oracle/kzg_oracle.c on small sizes and to
evaluate the pairing-free validity identities of SURVEY.md 8(c) ("run the pipeline on discrete logs").

Values here are plain Python ints (standard form, not Montgomery); points are affine (x, y) or None.
"""
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 52435875175126190479447740508185965837690552500527637822603658699938581184513  # bls/globals.go:9
GX = 3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507  # bls/bls_hbls.go:23
GY = 1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569  # bls/bls_hbls.go:24
G = (GX, GY)


# ---------------- Fr side (exact restatement of fft.go / fft_fr.go / das_extension.go) ----------------
def root_of_unity(scale):
    return pow(7, (R - 1) >> scale, R)  # bls/globals.go:24-60


class FFTSettings:
    def __init__(self, scale):
        self.max_width = 1 << scale
        w = root_of_unity(scale)
        self.expanded = [pow(w, i, R) for i in range(self.max_width + 1)]  # fft.go:21-32
        self.reversed = self.expanded[::-1]  # fft.go:49-54

    def _fft(self, vals, roots, stride, mul, add, sub):
        n = len(vals)
        if n <= 4:  # simpleFT fft_fr.go:8-28
            out = []
            for i in range(n):
                last = mul(vals[0], roots[0])
                for j in range(1, n):
                    last = add(last, mul(vals[j], roots[((i * j) % n) * stride]))
                out.append(last)
            return out
        half = n // 2
        L = self._fft(vals[0::2], roots, stride * 2, mul, add, sub)
        Rr = self._fft(vals[1::2], roots, stride * 2, mul, add, sub)
        out = [None] * n
        for i in range(half):
            yr = mul(Rr[i], roots[i * stride])
            out[i] = add(L[i], yr)
            out[i + half] = sub(L[i], yr)
        return out

    def fft(self, vals, inv=False):  # fft_fr.go:55-105
        n = len(vals)
        assert n <= self.max_width
        np2 = 1 if n == 0 else 1 << (n - 1).bit_length()
        vals = list(vals) + [0] * (np2 - n)
        mul = lambda a, b: a * b % R
        add = lambda a, b: (a + b) % R
        sub = lambda a, b: (a - b) % R
        stride = self.max_width // np2
        if inv:
            out = self._fft(vals, self.reversed, stride, mul, add, sub)
            ninv = pow(np2, -1, R)
            return [o * ninv % R for o in out]
        return self._fft(vals, self.expanded, stride, mul, add, sub)

    def fft_g1(self, pts, inv=False):  # fft_g1.go:58-94
        n = len(pts)
        assert n and n & (n - 1) == 0 and n <= self.max_width
        stride = self.max_width // n
        mul = lambda p, k: g1_mul(p, k)
        sub = lambda a, b: g1_add(a, g1_neg(b))
        if inv:
            out = self._fft(list(pts), self.reversed, stride, mul, g1_add, sub)
            ninv = pow(n, -1, R)
            return [g1_mul(o, ninv) for o in out]
        return self._fft(list(pts), self.expanded, stride, mul, g1_add, sub)

    def _das(self, ab, s):  # das_extension.go:7-66
        if len(ab) == 2:
            x, y = (ab[0] + ab[1]) % R, (ab[0] - ab[1]) % R
            t = y * self.expanded[s] % R
            return [(x + t) % R, (x - t) % R]
        h = len(ab) // 2
        a0 = [(ab[i] + ab[h + i]) % R for i in range(h)]
        a1 = [(ab[i] - ab[h + i]) * self.reversed[2 * i * s] % R for i in range(h)]
        L, Rr = self._das(a0, 2 * s), self._das(a1, 2 * s)
        out = [0] * len(ab)
        for i in range(h):
            yr = Rr[i] * self.expanded[(1 + 2 * i) * s] % R
            out[i], out[h + i] = (L[i] + yr) % R, (L[i] - yr) % R
        return out

    def das_fft_extension(self, even):  # das_extension.go:71-84
        assert 2 * len(even) <= self.max_width
        ninv = pow(len(even), -1, R)
        return [v * ninv % R for v in self._das(list(even), 1)]


def rev_bits(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2) if bits else 0


def bitrev(a):  # reverse_bit_order.go:86-101
    bits = (len(a) - 1).bit_length()
    return [a[rev_bits(i, bits)] for i in range(len(a))]


def eval_poly(coeffs, x):  # bls/globals.go:76-95
    y = 0
    for c in reversed(coeffs):
        y = (y * x + c) % R
    return y


def quotient_linear(poly, x):  # poly.go:14-40 with divisor [-x, 1]
    n = len(poly)
    q = [0] * (n - 1)
    q[n - 2] = poly[n - 1]
    for i in range(n - 3, -1, -1):
        q[i] = (poly[i + 1] + x * q[i + 1]) % R
    return q


def toeplitz_coeffs_strided(p, off, l):  # fk20_single.go:89-103
    n = len(p)
    k = n // l
    out = [0] * (2 * k)
    out[0] = p[n - 1 - off]
    j = 2 * l - off - 1
    for i in range(k + 2, 2 * k):
        out[i] = p[j]
        j += l
    return out


# ---------------- G1 (affine, textbook) ----------------
def g1_neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x3 = (lam * lam - a[0] - b[0]) % P
    return (x3, (lam * (a[0] - x3) - a[1]) % P)


def _jdbl(X, Y, Z):
    if Z == 0:
        return (0, 1, 0)
    A, B = X * X % P, Y * Y % P
    Cc = B * B % P
    D = 2 * ((X + B) * (X + B) - A - Cc) % P
    E = 3 * A % P
    X3 = (E * E - 2 * D) % P
    return (X3, (E * (D - X3) - 8 * Cc) % P, 2 * Y * Z % P)


def _jadd_affine(X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % P
    U2, S2 = x2 * Z1Z1 % P, y2 * Z1 * Z1Z1 % P
    if U2 == X1:
        if S2 == Y1:
            return _jdbl(X1, Y1, Z1)
        return (0, 1, 0)
    H, r = (U2 - X1) % P, (S2 - Y1) % P
    HH = H * H % P
    HHH, V = H * HH % P, X1 * HH % P
    X3 = (r * r - HHH - 2 * V) % P
    return (X3, (r * (V - X3) - Y1 * HHH) % P, Z1 * H % P)


def g1_mul(a, k):
    k %= R
    if a is None or k == 0:
        return None
    acc = (0, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jdbl(*acc)
        if bit == "1":
            acc = _jadd_affine(*acc, a[0], a[1])
    X, Y, Z = acc
    if Z == 0:
        return None
    zi = pow(Z, -1, P)
    return (X * zi * zi % P, Y * zi * zi * zi % P)


def g1_compress(a):  # ZCash format
    if a is None:
        return bytes([0xC0]) + bytes(47)
    b = bytearray(a[0].to_bytes(48, "big"))
    b[0] |= 0x80
    if a[1] > (P - 1) // 2:
        b[0] |= 0x20
    return bytes(b)


def g1_decompress(b):
    b = bytes(b)
    assert b[0] & 0x80
    if b[0] & 0x40:
        return None
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
    y = pow((x * x * x + 4) % P, (P + 1) // 4, P)
    assert y * y % P == (x * x * x + 4) % P
    if (y > (P - 1) // 2) != bool(b[0] & 0x20):
        y = P - y
    return (x, y)


def lincomb(points, scalars):  # bls.LinCombG1 semantics (naive)
    acc = None
    for p, k in zip(points, scalars):
        acc = g1_add(acc, g1_mul(p, k))
    return acc


# ---------------- pipelines "in the exponent" (pairing-free oracle) ----------------
def fk20_single_da_dlogs(fs, poly, s):
    """dlogs of FK20SingleDAOptimized(poly || 0^n) for setup [s^i]G1 (fk20_single.go:139-172, kzg.go:43-64)."""
    n = len(poly)
    x = [pow(s, n - 2 - i, R) for i in range(n - 1)] + [0]
    X = fs.fft(x + [0] * n)
    Cf = fs.fft(toeplitz_coeffs_strided(poly, 0, 1))
    h = fs.fft([c * xx % R for c, xx in zip(Cf, X)], inv=True)[:n]
    return fs.fft(h + [0] * n)


def fk20_multi_da_dlogs(fs, poly, s, l):
    """dlogs of FK20MultiDAOptimized(poly || 0^n) (fk20_multi.go:58-109, kzg.go:73-116)."""
    n = len(poly)
    k = n // l
    H = [0] * (2 * k)
    for off in range(l):
        start = n - l - 1 - off
        x = [pow(s, start - t * l, R) for t in range(k - 1)] + [0]
        X = fs.fft(x + [0] * k)
        Cf = fs.fft(toeplitz_coeffs_strided(poly, off, l))
        H = [(hh + c * xx) % R for hh, c, xx in zip(H, Cf, X)]
    h = fs.fft(H, inv=True)[:k]
    return fs.fft(h + [0] * k)


def single_proof_dlog(poly, s, x):
    """(p(s) - p(x)) / (s - x): dlog of the KZG proof at x for setup secret s."""
    return (eval_poly(poly, s) - eval_poly(poly, x)) * pow(s - x, -1, R) % R


def coset_proof_dlog(poly, s, x, l):
    """(p(s) - I(s)) / (s^l - x^l), I = p mod (X^l - x^l): dlog of the coset proof."""
    xl = pow(x, l, R)
    rem = list(poly)
    for i in range(len(rem) - 1, l - 1, -1):
        rem[i - l] = (rem[i - l] + rem[i] * xl) % R
        rem[i] = 0
    I = rem[:l]
    return (eval_poly(poly, s) - eval_poly(I, s)) * pow(pow(s, l, R) - xl, -1, R) % R


# ---------------- eth/ aggregate proofs: Fiat-Shamir transcript and aggregation (eth/helpers.go) ----------------
import hashlib  # noqa: E402

FIAT_SHAMIR_PROTOCOL_DOMAIN = b"FSBLOBVERIFY_V1_"  # eth/helpers.go:18


def hash_to_bls_field(data):
    """hashToBLSField (eth/helpers.go:113-133): SHA-256, digest as a little-endian integer, mod r."""
    return int.from_bytes(hashlib.sha256(data).digest(), "little") % R


def hash_polys_comms(polys, comms, field_elements_per_blob=4096):
    """hashPolysComms (eth/helpers.go:235-260): polys = lists of ints, comms = 48-byte strings."""
    h = hashlib.sha256()
    h.update(FIAT_SHAMIR_PROTOCOL_DOMAIN)
    h.update(field_elements_per_blob.to_bytes(8, "little"))
    h.update(len(polys).to_bytes(8, "little"))
    for poly in polys:
        for fe in poly:
            h.update(fe.to_bytes(32, "little"))  # bls.FrTo32
    for c in comms:
        h.update(bytes(c))
    return h.digest()


def compute_challenges(polys, comms, field_elements_per_blob=4096):
    """ComputeChallenges (eth/helpers.go:215-232): (powers of the linear-combination challenge, evaluation challenge)."""
    digest = hash_polys_comms(polys, comms, field_elements_per_blob)
    r = hash_to_bls_field(digest + b"\x00")
    z = hash_to_bls_field(digest + b"\x01")
    powers, cur = [], 1
    for _ in polys:  # ComputePowers, eth/helpers.go:87-96
        powers.append(cur)
        cur = cur * r % R
    return powers, z


def compute_aggregated_poly(polys, comms, field_elements_per_blob=4096):
    """ComputeAggregatedPolyAndCommitment (eth/helpers.go:137-162) without the G1 part: (aggregated polynomial, powers, z);
    bls.PolyLinComb (bls/globals.go:155-178) of no vector is the zero vector."""
    powers, z = compute_challenges(polys, comms, field_elements_per_blob)
    agg = [0] * field_elements_per_blob
    for s_, poly in zip(powers, polys):
        agg = [(a + s_ * v) % R for a, v in zip(agg, poly)]
    return agg, powers, z


def eval_in_evaluation_form(poly, x, domain):
    """bls.EvaluatePolyInEvaluationForm (bls/globals.go:106-153), x outside the domain."""
    n = len(poly)
    acc = 0
    for p_, w in zip(poly, domain):
        acc = (acc + p_ * w % R * pow(x - w, -1, R)) % R
    return acc * ((pow(x, n, R) - 1) * pow(n, -1, R) % R) % R

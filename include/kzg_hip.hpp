// kzg_hip.hpp -- header-only C++17 mirror of go-kzg's prover-side Go API over the C ABI of kzg_hip.h.
//
// The reference's host language is Go and this image has no Go toolchain, so next to the (uncompiled) cgo shim in go-kzg_amd/goshim/ this is
// the COMPILED host-side mirror: same type and method names, argument meaning and error behaviour as the reference, one line of C ABI per
// method.  tests/host/go_mirror_test.cpp is written against it the way the reference's *_test.go files are written against the Go package.
//   Go `error` return (fft_fr.go:57-59,78-83; fft_g1.go:60-65)        -> kzg::Error   (status 1-2)
//   Go panic (kzg.go:22-27,44-52,74-91; fk20_*.go; bls_kilic.go:133)  -> kzg::Panic   (status 3-7 and anything else)
// Memory images are the Kilic backend's: bls.Fr = [4]uint64 Montgomery, bls.G1Point = [3][6]uint64 Jacobian Montgomery; slices are passed as is.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "kzg_hip.h"

namespace kzg {

struct Fr { uint64_t l[4]; };                   // bls.Fr (bls/bignum_kilic.go:21-23)
struct G1Point { uint64_t l[18]; };             // bls.G1Point (bls/bls_kilic.go:30-35)
inline bool EqualFr(const Fr &a, const Fr &b) { return std::memcmp(&a, &b, sizeof a) == 0; }
inline bool EqualG1(const G1Point &a, const G1Point &b) { return std::memcmp(&a, &b, sizeof a) == 0; }   // outputs are normalised images

struct Error : std::runtime_error { int status; Error(int st, const std::string &m) : std::runtime_error(m), status(st) {} };
struct Panic : std::logic_error { int status; Panic(int st, const std::string &m) : std::logic_error(m), status(st) {} };

namespace detail {
inline std::string panic_text(int st) {
    switch (st) {
    case KZG_HIP_ERR_LEN_MISMATCH: return "kzg_hip: slice length mismatch";
    case KZG_HIP_ERR_UPPER_HALF: return "bad input, second half should be zeroed";
    case KZG_HIP_ERR_NO_DEVICE: return "kzg_hip: no gfx950 device (there is no CPU fallback)";
    case KZG_HIP_ERR_HIP: return std::string("kzg_hip: ") + kzg_hip_last_error();
    default: return "kzg_hip: status " + std::to_string(st);
    }
}
inline void must(int st) { if (st != KZG_HIP_OK) throw Panic(st, panic_text(st)); }
// status 1-2 are the FFT layer's `error` values, with the reference's texts
inline void fft_err(int st, uint64_t n, uint64_t max_width) {
    if (st == KZG_HIP_OK) return;
    if (st == KZG_HIP_ERR_TOO_WIDE) throw Error(st, "got " + std::to_string(n) + " values but only have " + std::to_string(max_width) + " roots of unity");
    if (st == KZG_HIP_ERR_NOT_POW2) throw Error(st, "got " + std::to_string(n) + " values but not a power of two");
    throw Panic(st, panic_text(st));
}
inline uint64_t next_pow2(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }
}  // namespace detail

// fft.go:34-61
class FFTSettings {
  public:
    uint8_t MaxScale; uint64_t MaxWidth;
    explicit FFTSettings(uint8_t maxScale, int device = 0) : MaxScale(maxScale) {   // NewFFTSettings
        detail::must(kzg_hip_fft_settings_new(device, maxScale, &h_));
        MaxWidth = kzg_hip_fft_max_width(h_);
    }
    ~FFTSettings() { kzg_hip_fft_settings_free(h_); }
    FFTSettings(const FFTSettings &) = delete;
    FFTSettings &operator=(const FFTSettings &) = delete;
    kzg_hip_fft *handle() const { return h_; }

    std::vector<Fr> ExpandedRootsOfUnity() const { std::vector<Fr> r(MaxWidth + 1); detail::must(kzg_hip_fft_roots(h_, 0, r.data())); return r; }
    std::vector<Fr> ReverseRootsOfUnity() const { std::vector<Fr> r(MaxWidth + 1); detail::must(kzg_hip_fft_roots(h_, 1, r.data())); return r; }
    // fft_fr.go:55-74
    std::vector<Fr> FFT(const std::vector<Fr> &vals, bool inv) const {
        std::vector<Fr> out(detail::next_pow2(vals.size()));
        uint64_t n = 0;
        detail::fft_err(kzg_hip_fft_fr(h_, vals.data(), vals.size(), inv, out.data(), &n), vals.size(), MaxWidth);
        out.resize(n);
        return out;
    }
    // fft_fr.go:76-105
    void InplaceFFT(const std::vector<Fr> &vals, std::vector<Fr> &out, bool inv) const {
        out.resize(vals.size());
        detail::fft_err(kzg_hip_inplace_fft_fr(h_, vals.data(), out.data(), vals.size(), inv), vals.size(), MaxWidth);
    }
    // fft_g1.go:58-94
    std::vector<G1Point> FFTG1(const std::vector<G1Point> &vals, bool inv) const {
        std::vector<G1Point> out(vals.size());
        detail::fft_err(kzg_hip_fft_g1(h_, vals.data(), vals.size(), inv, out.data()), vals.size(), MaxWidth);
        return out;
    }
    // das_extension.go:71-84 (in place, like the reference; panics on a domain that is too small)
    void DASFFTExtension(std::vector<Fr> &vals) const {
        if (vals.size() * 2 > MaxWidth) throw Panic(KZG_HIP_ERR_TOO_WIDE, "domain too small for extending requested values");
        detail::must(kzg_hip_das_fft_extension(h_, vals.data(), vals.size()));
    }
    // zero_poly.go:116-217: (zeroEval, zeroPoly)
    std::pair<std::vector<Fr>, std::vector<Fr>> ZeroPolyViaMultiplication(const std::vector<uint64_t> &missingIndices, uint64_t length) const {
        std::vector<Fr> ev(length), zp(length);
        detail::must(kzg_hip_zero_poly_via_multiplication(h_, missingIndices.data(), missingIndices.size(), length, ev.data(), zp.data()));
        return {ev, zp};
    }
    // recover_from_samples.go:42-109 (samples[i] == nil <=> present[i] == 0); the reference returns an error for inconsistent data
    std::vector<Fr> RecoverPolyFromSamples(const std::vector<Fr> &samples, const std::vector<uint8_t> &present) const {
        std::vector<Fr> out(samples.size());
        int st = kzg_hip_recover_poly_from_samples(h_, samples.data(), present.data(), samples.size(), out.data());
        if (st == KZG_HIP_ERR_RECOVERY) throw Error(st, "failed to reconstruct data correctly");
        detail::must(st);
        return out;
    }
    // bls helpers over slices
    std::vector<Fr> FrFrom32(const std::vector<uint8_t> &le32, bool *allOk = nullptr) const {
        std::vector<Fr> out(le32.size() / 32); int ok = 0;
        detail::must(kzg_hip_fr_from_le32(h_, le32.data(), out.size(), out.data(), &ok));
        if (allOk) *allOk = ok != 0;
        return out;
    }
    Fr AsFr(uint64_t v) const { std::vector<uint8_t> b(32, 0); std::memcpy(b.data(), &v, 8); return FrFrom32(b)[0]; }      // bls.AsFr
    Fr SetFr(const std::string &decimal) const {                                                                           // bls.SetFr
        std::vector<uint8_t> b(32, 0);
        for (char ch : decimal) { unsigned carry = (unsigned)(ch - '0'); for (auto &x : b) { unsigned v = x * 10u + carry; x = (uint8_t)v; carry = v >> 8; } }
        return FrFrom32(b)[0];
    }
    std::vector<uint8_t> ToCompressedG1(const std::vector<G1Point> &pts) const {
        std::vector<uint8_t> out(48 * pts.size());
        detail::must(kzg_hip_g1_to_compressed(h_, pts.data(), pts.size(), out.data()));
        return out;
    }
    G1Point LinCombG1(const std::vector<G1Point> &numbers, const std::vector<Fr> &factors) const {                           // bls/bls_kilic.go:132-150
        if (numbers.size() != factors.size()) throw Panic(KZG_HIP_ERR_LEN_MISMATCH, "got LinCombG1 numbers/factors length mismatch");
        G1Point out;
        detail::must(kzg_hip_lincomb_g1(h_, numbers.data(), factors.data(), numbers.size(), &out));
        return out;
    }
    // setup.go:9-26, G1 half: [s^i] G1
    std::vector<G1Point> GenerateTestingSetupG1(const std::string &secretDecimal, uint64_t n) const {
        Fr s = SetFr(secretDecimal);
        std::vector<G1Point> out(n);
        detail::must(kzg_hip_generate_testing_setup_g1(h_, &s, n, out.data()));
        return out;
    }

  private:
    kzg_hip_fft *h_ = nullptr;
};

// kzg.go:11-36 (prover side: SecretG1)
class KZGSettings {
  public:
    const FFTSettings *FFT;
    KZGSettings(const FFTSettings *fs, const std::vector<G1Point> &secretG1) : FFT(fs) {   // NewKZGSettings: panics if the setup is shorter than MaxWidth
        detail::must(kzg_hip_kzg_settings_new(fs->handle(), secretG1.data(), secretG1.size(), &h_));
    }
    ~KZGSettings() { kzg_hip_kzg_settings_free(h_); }
    KZGSettings(const KZGSettings &) = delete;
    KZGSettings &operator=(const KZGSettings &) = delete;
    kzg_hip_kzg *handle() const { return h_; }
    G1Point CommitToPoly(const std::vector<Fr> &coeffs) const {                     // kzg_single_proofs.go:17-19
        G1Point out; detail::must(kzg_hip_commit_to_poly(h_, coeffs.data(), coeffs.size(), &out)); return out;
    }
    G1Point CommitToPolyUnoptimized(const std::vector<Fr> &coeffs) const { return CommitToPoly(coeffs); }   // kzg_single_proofs.go:22-33: the same group element
    G1Point ComputeProofSingle(const std::vector<Fr> &poly, uint64_t x) const {     // kzg_single_proofs.go:36-54
        G1Point out; detail::must(kzg_hip_compute_proof_single(h_, poly.data(), poly.size(), x, &out)); return out;
    }
    G1Point ComputeProofMulti(const std::vector<Fr> &poly, uint64_t x, uint64_t n) const {   // kzg_multi_proofs.go:13-43
        G1Point out; detail::must(kzg_hip_compute_proof_multi(h_, poly.data(), poly.size(), x, n, &out)); return out;
    }

  private:
    kzg_hip_kzg *h_ = nullptr;
};

// kzg.go:38-64, fk20_single.go:122-196
class FK20SingleSettings {
  public:
    FK20SingleSettings(const KZGSettings *ks, uint64_t n2) : n2_(n2) { detail::must(kzg_hip_fk20_single_settings_new(ks->handle(), n2, &h_)); }   // NewFK20SingleSettings
    ~FK20SingleSettings() { kzg_hip_fk20_single_settings_free(h_); }
    FK20SingleSettings(const FK20SingleSettings &) = delete;
    FK20SingleSettings &operator=(const FK20SingleSettings &) = delete;
    std::vector<G1Point> FK20Single(const std::vector<Fr> &poly) const {
        std::vector<G1Point> out(poly.size()); detail::must(kzg_hip_fk20_single(h_, poly.data(), poly.size(), out.data())); return out;
    }
    std::vector<G1Point> FK20SingleDAOptimized(const std::vector<Fr> &poly) const {
        std::vector<G1Point> out(poly.size()); detail::must(kzg_hip_fk20_single_da_optimized(h_, poly.data(), poly.size(), out.data())); return out;
    }
    std::vector<G1Point> DAUsingFK20(const std::vector<Fr> &poly) const {
        std::vector<G1Point> out(2 * poly.size()); detail::must(kzg_hip_da_using_fk20(h_, poly.data(), poly.size(), out.data())); return out;
    }

  private:
    kzg_hip_fk20s *h_ = nullptr; uint64_t n2_;
};

// kzg.go:66-116, fk20_multi.go:25-133
class FK20MultiSettings {
  public:
    FK20MultiSettings(const KZGSettings *ks, uint64_t n2, uint64_t chunkLen) : chunk_(chunkLen) {   // NewFK20MultiSettings
        detail::must(kzg_hip_fk20_multi_settings_new(ks->handle(), n2, chunkLen, &h_));
    }
    ~FK20MultiSettings() { kzg_hip_fk20_multi_settings_free(h_); }
    FK20MultiSettings(const FK20MultiSettings &) = delete;
    FK20MultiSettings &operator=(const FK20MultiSettings &) = delete;
    std::vector<G1Point> DAUsingFK20Multi(const std::vector<Fr> &poly) const {
        std::vector<G1Point> out(2 * poly.size() / chunk_); detail::must(kzg_hip_da_using_fk20_multi(h_, poly.data(), poly.size(), out.data())); return out;
    }
    std::vector<G1Point> FK20MultiDAOptimized(const std::vector<Fr> &poly) const {
        std::vector<G1Point> out(poly.size() / chunk_); detail::must(kzg_hip_fk20_multi_da_optimized(h_, poly.data(), poly.size(), out.data())); return out;
    }

  private:
    kzg_hip_fk20m *h_ = nullptr; uint64_t chunk_;
};

// Several GPUs behind one settings object (kzg_hip_multi_*; the Go shim's NewMultiKZGSettings): NewFFTSettings(maxScale) + NewKZGSettings(fs, secretG1) on
// every device of `devices` (a device may be listed more than once); batches are divided among the devices, ONE DAUsingFK20 / DAUsingFK20Multi is sharded
// inside the library (Toeplitz stage by output position, all-gather of the slices: RCCL between distinct devices).
class MultiKZGSettings {
  public:
    MultiKZGSettings(const std::vector<int> &devices, unsigned maxScale, const std::vector<G1Point> &secretG1) {
        detail::must(kzg_hip_multi_settings_new(devices.data(), (uint32_t)devices.size(), maxScale, secretG1.data(), secretG1.size(), &h_));
    }
    ~MultiKZGSettings() { kzg_hip_multi_settings_free(h_); }
    MultiKZGSettings(const MultiKZGSettings &) = delete;
    MultiKZGSettings &operator=(const MultiKZGSettings &) = delete;
    kzg_hip_multi *handle() const { return h_; }
    std::string Transport() const { return kzg_hip_multi_transport(h_); }           // "rccl", "peer-copy" or "host-staged"
    std::string TransportSelfTest() const { return kzg_hip_multi_transport_check(h_); }   // "ok: <transport>, ..." -- the exchange test the constructor ran
    uint64_t Exchanges() const { return kzg_hip_multi_exchanges(h_); }
    void SetFFTSharding(int mode) const { detail::must(kzg_hip_multi_set_fft_sharding(h_, mode)); }   // 0 gather, 1 sharded transforms, -1 default
    void SetTableBudgetGB(double gb) const { detail::must(kzg_hip_multi_set_table_budget_gb(h_, gb)); }
    std::vector<G1Point> CommitToPolyBatch(const std::vector<Fr> &coeffsFlat, uint64_t n) const {   // CommitToPoly on coeffsFlat.size() / n polynomials
        std::vector<G1Point> out(coeffsFlat.size() / n);
        detail::must(kzg_hip_multi_commit_to_poly_batch(h_, coeffsFlat.data(), n, out.size(), out.data()));
        return out;
    }

  private:
    kzg_hip_multi *h_ = nullptr;
};
class MultiFK20SingleSettings {
  public:
    MultiFK20SingleSettings(const MultiKZGSettings *m, uint64_t n2) { detail::must(kzg_hip_multi_fk20_single_settings_new(m->handle(), n2, &h_)); }
    ~MultiFK20SingleSettings() { kzg_hip_multi_fk20_single_settings_free(h_); }
    MultiFK20SingleSettings(const MultiFK20SingleSettings &) = delete;
    MultiFK20SingleSettings &operator=(const MultiFK20SingleSettings &) = delete;
    std::vector<G1Point> DAUsingFK20(const std::vector<Fr> &poly) const {            // ONE polynomial over all devices
        std::vector<G1Point> out(2 * poly.size()); detail::must(kzg_hip_multi_da_using_fk20(h_, poly.data(), poly.size(), out.data())); return out;
    }
    std::vector<G1Point> DAUsingFK20Batch(const std::vector<Fr> &polysFlat, uint64_t n) const {   // polynomials divided among the devices
        std::vector<G1Point> out(2 * polysFlat.size());
        detail::must(kzg_hip_multi_da_using_fk20_batch(h_, polysFlat.data(), n, polysFlat.size() / n, out.data()));
        return out;
    }

  private:
    kzg_hip_multi_fk20s *h_ = nullptr;
};
class MultiFK20MultiSettings {
  public:
    MultiFK20MultiSettings(const MultiKZGSettings *m, uint64_t n2, uint64_t chunkLen) : chunk_(chunkLen) {
        detail::must(kzg_hip_multi_fk20_multi_settings_new(m->handle(), n2, chunkLen, &h_));
    }
    ~MultiFK20MultiSettings() { kzg_hip_multi_fk20_multi_settings_free(h_); }
    MultiFK20MultiSettings(const MultiFK20MultiSettings &) = delete;
    MultiFK20MultiSettings &operator=(const MultiFK20MultiSettings &) = delete;
    std::vector<G1Point> DAUsingFK20Multi(const std::vector<Fr> &poly) const {       // ONE polynomial over all devices (fk20_multi.go:113-133)
        std::vector<G1Point> out(2 * poly.size() / chunk_); detail::must(kzg_hip_multi_da_using_fk20_multi(h_, poly.data(), poly.size(), out.data())); return out;
    }

  private:
    kzg_hip_multi_fk20m *h_ = nullptr; uint64_t chunk_;
};

// package eth (eth/globals.go:39-72, eth/eth.go:145-182, eth/helpers.go:179-211): byte-level callers.  Blob = FieldElementsPerBlob x 32 little-endian
// bytes, KZGCommitment / KZGProof = 48 bytes; `error` returns of the reference are kzg::Error with its texts.
namespace eth {
using Blob = std::vector<uint8_t>;
using Bytes48 = std::array<uint8_t, 48>;
class Settings {
  public:
    // setupG1Lagrange in NATURAL order, as eth/trusted_setup.json stores it (the library applies bitReversalPermutation, eth/globals.go:48)
    Settings(const FFTSettings *fs, const std::vector<G1Point> &setupG1Lagrange) : n_(setupG1Lagrange.size()) {
        detail::must(kzg_hip_eth_settings_new(fs->handle(), setupG1Lagrange.data(), n_, &h_));
    }
    ~Settings() { kzg_hip_eth_settings_free(h_); }
    Settings(const Settings &) = delete;
    Settings &operator=(const Settings &) = delete;
    std::pair<Bytes48, bool> BlobToKZGCommitment(const Blob &blob) const {                        // eth/eth.go:145-151
        Bytes48 out{}; uint8_t ok = 0;
        if (blob.size() != n_ * 32) throw Panic(KZG_HIP_ERR_LEN_MISMATCH, "blob size");
        detail::must(kzg_hip_eth_blob_to_kzg_commitment_batch(h_, blob.data(), 1, out.data(), &ok));
        return {out, ok != 0};
    }
    Bytes48 ComputeKZGProof(const std::vector<Fr> &polynomial, const Fr &z) const {               // eth/helpers.go:179-203
        Bytes48 out{};
        int st = kzg_hip_eth_compute_kzg_proof(h_, polynomial.data(), polynomial.size(), &z, out.data(), nullptr);
        if (st == KZG_HIP_ERR_LEN_MISMATCH) throw Error(st, "polynomial has invalid length");
        if (st == KZG_HIP_ERR_BAD_ARG) throw Error(st, "invalid z challenge");
        detail::must(st);
        return out;
    }
    Fr EvaluatePolynomialInEvaluationForm(const std::vector<Fr> &poly, const Fr &x) const {       // eth/helpers.go:207-211
        Fr y; detail::must(kzg_hip_eth_evaluate_polynomial_in_evaluation_form(h_, poly.data(), poly.size(), &x, &y)); return y;
    }
    Bytes48 ComputeAggregateKZGProof(const std::vector<Blob> &blobs, std::vector<Bytes48> *commitments = nullptr) const {   // eth/eth.go:175-182
        std::vector<uint8_t> flat;
        for (const auto &b : blobs) { if (b.size() != n_ * 32) throw Panic(KZG_HIP_ERR_LEN_MISMATCH, "blob size"); flat.insert(flat.end(), b.begin(), b.end()); }
        Bytes48 out{};
        std::vector<Bytes48> comm(blobs.size());
        int st = kzg_hip_eth_compute_aggregate_kzg_proof(h_, flat.data(), blobs.size(), out.data(), comm.data());
        if (st == KZG_HIP_ERR_BAD_BLOB) throw Error(st, "could not convert blobs to polynomials");
        if (st == KZG_HIP_ERR_BAD_ARG) throw Error(st, "invalid z challenge");
        detail::must(st);
        if (commitments) *commitments = comm;
        return out;
    }

  private:
    kzg_hip_eth *h_ = nullptr; uint64_t n_;
};
}  // namespace eth

}  // namespace kzg

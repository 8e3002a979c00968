// go_mirror_test.cpp -- the reference's own tests, re-stated against the C++ mirror of its Go API (include/kzg_hip.hpp), so that the
// boundary is exercised by COMPILED host code shaped like a go-kzg caller: same constructors, same method names, same error behaviour.
//   TestFFTRoundtrip, TestInvFFT                       fft_fr_test.go:9-71
//   TestDASFFTExtension, TestParametrizedDASFFTExtension  das_extension_test.go:11-84
//   TestKZGSettings_CommitToEvalPoly / _CheckProofSingle   kzg_single_proofs_test.go:11-64   (pairing check -> SURVEY 8c vectors A, B)
//   TestKZGSettings_DAUsingFK20                        fk20_single_test.go:11-47           (pairing check -> vector C)
//   TestFFTSettings_RecoverPolyFromSamples_Simple      recover_from_samples_test.go:10-60
//   package eth (no tests in the reference): ComputeAggregateKZGProof / BlobToKZGCommitment / ComputeKZGProof against oracle values
//   error behaviour: FFT's `error` values (fft_fr.go:57-59,78-83) and the panics of kzg.go:22-27, fk20_single.go:140-154
// Expected values come from tests/golden/*.json through a key/value file written by tests/test_cabi.py (argv[1]).
// TEST INFRASTRUCTURE, built and run by tests/test_cabi.py (-m gpu).
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include "kzg_hip.hpp"

using namespace kzg;
static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; printf("FAIL %s:%d: ", __func__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)
static std::map<std::string, std::vector<std::string>> KAT;

static std::vector<Fr> testPoly(const FFTSettings &fs, std::initializer_list<uint64_t> c) {   // kzg_single_proofs_test.go:66-73
    std::vector<Fr> p;
    for (uint64_t v : c) p.push_back(fs.AsFr(v));
    return p;
}
static std::string hex(const std::vector<uint8_t> &b, size_t off, size_t n) {
    static const char d[] = "0123456789abcdef"; std::string s;
    for (size_t i = 0; i < n; i++) { s += d[b[off + i] >> 4]; s += d[b[off + i] & 15]; }
    return s;
}
static const std::vector<uint64_t> POLY = {1, 2, 3, 4, 7, 7, 7, 7, 13, 13, 13, 13, 13, 13, 13, 13};
static const char *SECRET = "1927409816240961209460912649124";

static void TestFFTRoundtrip() {
    FFTSettings fs(4);
    std::vector<Fr> data(fs.MaxWidth);
    for (uint64_t i = 0; i < fs.MaxWidth; i++) data[i] = fs.AsFr(i);
    auto coeffs = fs.FFT(data, false);
    auto res = fs.FFT(coeffs, true);
    for (size_t i = 0; i < res.size(); i++) CHECK(EqualFr(res[i], data[i]), "difference: %zu", i);
}
static void TestInvFFT() {
    FFTSettings fs(4);
    std::vector<Fr> data(fs.MaxWidth);
    for (uint64_t i = 0; i < fs.MaxWidth; i++) data[i] = fs.AsFr(i);
    auto res = fs.FFT(data, true);
    const auto &expected = KAT["test_inv_fft"];
    CHECK(expected.size() == 16, "fixture");
    for (size_t i = 0; i < res.size(); i++) CHECK(EqualFr(res[i], fs.SetFr(expected[i])), "difference: %zu", i);
}
static void TestDASFFTExtension() {
    FFTSettings fs(4);
    uint64_t half = fs.MaxWidth / 2;
    std::vector<Fr> data(half);
    for (uint64_t i = 0; i < half; i++) data[i] = fs.AsFr(i);
    fs.DASFFTExtension(data);
    const auto &expected = KAT["test_das_fft_extension"];
    for (size_t i = 0; i < data.size(); i++) CHECK(EqualFr(data[i], fs.SetFr(expected[i])), "difference: %zu", i);
}
static void TestParametrizedDASFFTExtension() {
    for (uint8_t scale = 4; scale < 13; scale++) {
        FFTSettings fs(scale);
        uint64_t st = 1000 + scale;
        std::vector<Fr> evenData(fs.MaxWidth / 2);
        for (auto &v : evenData) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = fs.AsFr(st); }
        std::vector<Fr> oddData = evenData;
        fs.DASFFTExtension(oddData);
        std::vector<Fr> data(fs.MaxWidth);
        for (uint64_t i = 0; i < fs.MaxWidth; i += 2) { data[i] = evenData[i >> 1]; data[i + 1] = oddData[i >> 1]; }
        auto coeffs = fs.FFT(data, true);
        const Fr zero = fs.AsFr(0);
        for (uint64_t i = fs.MaxWidth / 2; i < fs.MaxWidth; i++) CHECK(EqualFr(coeffs[i], zero), "scale %d: expected zero coefficient on index %llu", scale, (unsigned long long)i);
    }
}
static void TestKZGSettings_CommitToEvalPoly_and_CheckProofSingle() {
    FFTSettings fs(4);
    auto s1 = fs.GenerateTestingSetupG1(SECRET, 16 + 1);
    KZGSettings ks(&fs, s1);
    auto polynomial = testPoly(fs, {1, 2, 3, 4, 7, 7, 7, 7, 13, 13, 13, 13, 13, 13, 13, 13});
    auto evalPoly = fs.FFT(polynomial, false);
    std::vector<G1Point> first16(s1.begin(), s1.begin() + 16);
    auto secretG1IFFT = fs.FFTG1(first16, true);
    G1Point commitmentByCoeffs = ks.CommitToPoly(polynomial);
    G1Point commitmentByEval = fs.LinCombG1(secretG1IFFT, evalPoly);                 // CommitToEvalPoly, kzg_single_proofs.go:12-14
    CHECK(EqualG1(commitmentByEval, commitmentByCoeffs), "expected commitments to be equal");
    CHECK(hex(fs.ToCompressedG1({commitmentByCoeffs}), 0, 48) == KAT["A_commit"][0], "vector A");
    G1Point proof = ks.ComputeProofSingle(polynomial, 17);
    CHECK(hex(fs.ToCompressedG1({proof}), 0, 48) == KAT["B_proof"][0], "vector B (what CheckProofSingle accepts for x = 17)");
}
static void TestKZGSettings_DAUsingFK20() {
    FFTSettings fs(5);
    auto s1 = fs.GenerateTestingSetupG1(SECRET, 32 + 1);
    KZGSettings ks(&fs, s1);
    FK20SingleSettings fk(&ks, 32);
    auto polynomial = testPoly(fs, {1, 2, 3, 4, 7, 7, 7, 7, 13, 13, 13, 13, 13, 13, 13, 13});
    auto allProofs = fk.DAUsingFK20(polynomial);
    CHECK(allProofs.size() == 32, "32 proofs");
    auto bytes = fs.ToCompressedG1(allProofs);
    const auto &idx = KAT["C_idx"], &want = KAT["C_val"];                           // positions 0, 18 (= reverseBitsLimited(32, 9), fk20_single_test.go:41) and 31
    CHECK(idx.size() == 3 && want.size() == 3, "fixture");
    for (size_t k = 0; k < idx.size(); k++) { size_t i = std::stoul(idx[k]); CHECK(hex(bytes, 48 * i, 48) == want[k], "vector C: proof %zu", i); }
}
static void TestFFTSettings_RecoverPolyFromSamples_Simple() {
    FFTSettings fs(2);
    std::vector<Fr> poly(fs.MaxWidth, fs.AsFr(0));
    for (uint64_t i = 0; i < fs.MaxWidth / 2; i++) poly[i] = fs.AsFr(i);
    auto data = fs.FFT(poly, false);
    std::vector<Fr> subset(fs.MaxWidth, fs.AsFr(0));
    std::vector<uint8_t> present(fs.MaxWidth, 0);
    subset[0] = data[0]; present[0] = 1;
    subset[3] = data[3]; present[3] = 1;
    auto recovered = fs.RecoverPolyFromSamples(subset, present);
    for (size_t i = 0; i < recovered.size(); i++) CHECK(EqualFr(recovered[i], data[i]), "recovery at index %zu", i);
    auto back = fs.FFT(recovered, true);
    for (uint64_t i = 0; i < fs.MaxWidth; i++) CHECK(EqualFr(back[i], poly[i]), "coeff at index %llu", (unsigned long long)i);
}
// package eth has no tests in the reference: the block-level caller against values computed by the oracle (tests/test_cabi.py writes them)
static eth::Blob mirrorBlob(uint64_t b) {                                             // element i of blob b = i * i + 7 * b + 3 (same formula in test_cabi.py)
    eth::Blob blob(4096 * 32, 0);
    for (uint64_t i = 0; i < 4096; i++) { uint64_t v = i * i + 7 * b + 3; std::memcpy(&blob[32 * i], &v, 8); }
    return blob;
}
static void TestEth_ComputeAggregateKZGProof() {
    FFTSettings fs(12);
    auto setup = fs.GenerateTestingSetupG1("1337", 4096);                              // eth/trusted_setup.json's secret
    auto lagrange = fs.FFTG1(setup, true);                                             // setup_G1_lagrange, natural order
    eth::Settings es(&fs, lagrange);
    std::vector<eth::Blob> blobs = {mirrorBlob(0), mirrorBlob(1)};
    std::vector<eth::Bytes48> comms;
    auto proof = es.ComputeAggregateKZGProof(blobs, &comms);
    auto hex48 = [](const eth::Bytes48 &b) { return hex(std::vector<uint8_t>(b.begin(), b.end()), 0, 48); };
    const auto &want = KAT["eth_aggregate"];                                           // commitment 0, commitment 1, proof
    CHECK(want.size() == 3, "fixture");
    CHECK(comms.size() == 2 && hex48(comms[0]) == want[0] && hex48(comms[1]) == want[1], "commitments");
    CHECK(hex48(proof) == want[2], "aggregated proof %s", hex48(proof).c_str());
    auto c0 = es.BlobToKZGCommitment(blobs[0]);
    CHECK(c0.second && hex48(c0.first) == want[0], "BlobToKZGCommitment");
    CHECK(hex48(es.ComputeAggregateKZGProof({})) == "c0" + std::string(94, '0'), "no blobs: the proof of the zero polynomial");
    auto bad = blobs;
    std::memset(&bad[1][32 * 100], 0xff, 32);                                          // an element >= r
    try { es.ComputeAggregateKZGProof(bad); CHECK(false, "expected an error"); }
    catch (const Error &e) { CHECK(std::string(e.what()) == "could not convert blobs to polynomials", "%s", e.what()); }
    CHECK(!es.BlobToKZGCommitment(bad[1]).second, "BlobToKZGCommitment of an invalid blob");
    auto poly = fs.FrFrom32(blobs[0]);
    try { es.ComputeKZGProof(poly, fs.AsFr(1)); CHECK(false, "expected an error"); }   // 1 = DomainFr[0]
    catch (const Error &e) { CHECK(std::string(e.what()) == "invalid z challenge", "%s", e.what()); }
}
static void TestErrorsAndPanics() {
    FFTSettings fs(4);
    std::vector<Fr> tooMany(17, fs.AsFr(1)), notPow2(12, fs.AsFr(1));
    try { fs.FFT(tooMany, false); CHECK(false, "expected an error"); }
    catch (const Error &e) { CHECK(std::string(e.what()) == "got 17 values but only have 16 roots of unity", "%s", e.what()); }
    std::vector<Fr> out;
    try { fs.InplaceFFT(notPow2, out, false); CHECK(false, "expected an error"); }
    catch (const Error &e) { CHECK(std::string(e.what()) == "got 12 values but not a power of two", "%s", e.what()); }
    auto s1 = fs.GenerateTestingSetupG1(SECRET, 8);
    try { KZGSettings ks(&fs, s1); CHECK(false, "expected a panic: setup shorter than MaxWidth (kzg.go:25-27)"); }
    catch (const Panic &p) { CHECK(p.status == KZG_HIP_ERR_LEN_MISMATCH, "status %d", p.status); }
    FFTSettings fs5(5);
    auto s33 = fs5.GenerateTestingSetupG1(SECRET, 33);
    KZGSettings ks5(&fs5, s33);
    FK20SingleSettings fk(&ks5, 32);
    std::vector<Fr> dirty(32, fs5.AsFr(3));                                           // upper half not zeroed
    try { fk.FK20SingleDAOptimized(dirty); CHECK(false, "expected a panic"); }
    catch (const Panic &p) { CHECK(std::string(p.what()) == "bad input, second half should be zeroed", "%s", p.what()); }
    std::vector<Fr> big(16, fs.AsFr(1));
    try { fs.DASFFTExtension(big); CHECK(false, "expected a panic"); }
    catch (const Panic &p) { CHECK(std::string(p.what()) == "domain too small for extending requested values", "%s", p.what()); }
}

// fk20_single_test.go:12-41 again, through a settings object that spans TWO entries of device 0 (the multi-device handle): same proofs, in both exchange schemes
static void TestMultiKZGSettings_DAUsingFK20() {
    FFTSettings fs(5);
    auto s1 = fs.GenerateTestingSetupG1(SECRET, 32 + 1);
    MultiKZGSettings m({0, 0}, 5, s1);
    CHECK(m.Transport() == "peer-copy", "two entries on one GPU exchange by peer copies");
    MultiFK20SingleSettings fk(&m, 32);
    KZGSettings ks(&fs, s1);
    FK20SingleSettings fk1(&ks, 32);
    auto polynomial = testPoly(fs, {1, 2, 3, 4, 7, 7, 7, 7, 13, 13, 13, 13, 13, 13, 13, 13});
    auto want = fk1.DAUsingFK20(polynomial);
    for (int mode = 0; mode < 2; mode++) {
        m.SetFFTSharding(mode);
        uint64_t before = m.Exchanges();
        auto got = fk.DAUsingFK20(polynomial);
        CHECK(m.Exchanges() - before == (mode ? 5u : 1u), "all-gathers of scheme %d", mode);
        CHECK(got.size() == 32, "32 proofs");
        bool same = got.size() == want.size();
        for (size_t i = 0; same && i < got.size(); i++) same = EqualG1(got[i], want[i]);
        CHECK(same, "multi-device proofs equal the single-device ones (scheme %d)", mode);
    }
    std::vector<Fr> two(polynomial);
    two.insert(two.end(), polynomial.begin(), polynomial.end());
    auto cs = m.CommitToPolyBatch(two, 16);
    CHECK(cs.size() == 2 && EqualG1(cs[0], ks.CommitToPoly(polynomial)) && EqualG1(cs[1], cs[0]), "CommitToPolyBatch over the two entries");
    auto bytes = fs.ToCompressedG1(cs);
    CHECK(hex(bytes, 0, 48) == KAT["A_commit"][0], "vector A through the multi-device handle");
}

int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: go_mirror_test <kat file>\n"); return 2; }
    std::ifstream f(argv[1]);
    std::string line;
    while (std::getline(f, line)) { std::istringstream is(line); std::string k, v; is >> k; while (is >> v) KAT[k].push_back(v); }
    if (kzg_hip_device_count() < 1) { printf("no gfx950 device\n"); return 77; }
    struct { const char *name; void (*fn)(); } tests[] = {
        {"TestFFTRoundtrip", TestFFTRoundtrip}, {"TestInvFFT", TestInvFFT}, {"TestDASFFTExtension", TestDASFFTExtension},
        {"TestParametrizedDASFFTExtension", TestParametrizedDASFFTExtension},
        {"TestKZGSettings_CommitToEvalPoly_and_CheckProofSingle", TestKZGSettings_CommitToEvalPoly_and_CheckProofSingle},
        {"TestKZGSettings_DAUsingFK20", TestKZGSettings_DAUsingFK20},
        {"TestFFTSettings_RecoverPolyFromSamples_Simple", TestFFTSettings_RecoverPolyFromSamples_Simple}, {"TestErrorsAndPanics", TestErrorsAndPanics},
        {"TestEth_ComputeAggregateKZGProof", TestEth_ComputeAggregateKZGProof}, {"TestMultiKZGSettings_DAUsingFK20", TestMultiKZGSettings_DAUsingFK20}};
    for (auto &t : tests) {
        int before = failures;
        try { t.fn(); } catch (const std::exception &e) { failures++; printf("FAIL %s: unexpected %s\n", t.name, e.what()); }
        printf("%s %s\n", failures == before ? "ok  " : "FAIL", t.name);
    }
    printf("%s: %d failure(s)\n", failures ? "FAILED" : "PASSED", failures);
    return failures ? 1 : 0;
}

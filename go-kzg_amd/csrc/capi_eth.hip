// capi_eth.hip -- eth/ byte-level prover path (f1), evaluation-form helpers, text / JSON setup loading (f4)
#include "capi_common.hpp"

// result rows of the coalesced one-polynomial calls, assembled ON THE DEVICE: row r = 48 commitment / proof bytes | (y, 32 bytes) | the 4-byte flag, padded to `pitch`, so that
// ONE contiguous copy brings a batch's results to the pinned staging rows.  (Through round 5 the rows were gathered by two / three hipMemcpy2DAsync device -> host copies,
// which ROCm 7 completes on a ~0.45 ms cadence whatever their size: a lone eth.ComputeKZGProof took 0.449 ms through four different kernel and allocation changes.)
__global__ void k_eth_pack_rows(const uint8_t *c48, const fr *y, const uint32_t *bad, uint64_t rows, uint32_t pitch_words, uint32_t *out) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= rows * pitch_words) return;
    const uint64_t r = t / pitch_words; const uint32_t w = (uint32_t)(t % pitch_words);
    const uint32_t flag_at = y ? 20u : 12u;
    uint32_t v = 0;
    if (w < 12) v = reinterpret_cast<const uint32_t *>(c48)[r * 12 + w];
    else if (y && w < 20) v = y[r].l[w - 12];
    else if (w == flag_at) v = bad[r];
    out[t] = v;
}
static int eth_rows_to_host(hipStream_t s, const uint8_t *d_c48, const fr *d_y, const uint32_t *d_bad, uint64_t rows, uint32_t pitch, uint8_t *h_out) {
    dtmp<uint32_t> d_pack(s);
    const uint64_t words = rows * (pitch / 4);
    CHK(d_pack.alloc(words));
    hipLaunchKernelGGL(k_eth_pack_rows, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, s, d_c48, d_y, d_bad, rows, pitch / 4, d_pack.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_out, d_pack.p, words * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));                             // (the temporary is released after the copy has landed)
    return KZG_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// eth/ byte-level prover path (row f1)
// ---------------------------------------------------------------------------------------------------------

int kzg_hip_eth_settings_new(kzg_hip_fft *fs, const void *lagrange_g1, uint64_t n, kzg_hip_eth **out) {
    if (!fs || !lagrange_g1 || !out) return KZG_HIP_ERR_BAD_ARG;
    *out = nullptr;
    if (n == 0 || !is_pow2(n)) return KZG_HIP_ERR_NOT_POW2;
    if (n > fs->W) return KZG_HIP_ERR_TOO_WIDE;
    KZG_TRY
    std::vector<g1j> br(n);
    std::vector<fr> dom(n);
    const g1j *src = (const g1j *)lagrange_g1;
    uint32_t logn = ilog2(n);
    for (uint64_t i = 0; i < n; i++) {   // bitReversalPermutation (eth/helpers.go, used at eth/globals.go:48)
        uint64_t r = 0;
        for (uint32_t b = 0; b < logn; b++) if (i & (1ull << b)) r |= 1ull << (logn - 1 - b);
        br[i] = src[r];
        // natural-order scale-log2(n) domain = every (W / n)-th expanded root; DomainFr[i] = domain[bitrev(i)] (eth/globals.go:61-66)
        dom[i] = fs->h_expanded[r * (fs->W / n)];
    }
    std::unique_ptr<kzg_hip_eth, void (*)(kzg_hip_eth *)> own(new kzg_hip_eth, kzg_hip_eth_settings_free);
    kzg_hip_eth *eth = own.get();
    eth->fs = fs; eth->n = n;
    // KZGSettings requires len(setup) >= MaxWidth (kzg.go:25-27); the eth setup is exactly its own width, so build it directly
    CHK(kzg_settings_build(fs, br.data(), n, &eth->ks));
    {
        dev_guard g(fs);
        HIPCHK(hipMalloc((void **)&eth->d_domain, n * sizeof(fr)));
        HIPCHK(hipMemcpy(eth->d_domain, dom.data(), n * sizeof(fr), hipMemcpyHostToDevice));
    }
    *out = own.release();
    return KZG_HIP_OK;
    KZG_CATCH
}
void kzg_hip_eth_settings_free(kzg_hip_eth *eth) {
    if (!eth) return;
    hipSetDevice(eth->fs->device);
    hipDeviceSynchronize();
    eth->co_blob.reset(); eth->co_proof.reset();
    kzg_hip_kzg_settings_free(eth->ks);   // drains the device first
    hipFree(eth->d_domain);
    (void)hipGetLastError();
    delete eth;
}
int kzg_hip_eth_blob_to_kzg_commitment_batch(kzg_hip_eth *eth, const void *blobs_le32, uint64_t batch, void *out48, uint8_t *ok) {
    if (!eth || !blobs_le32 || !out48 || !ok) return KZG_HIP_ERR_BAD_ARG;
    if (!batch) return KZG_HIP_OK;
    KZG_TRY
    if (batch == 1 && coalescing_enabled()) {
        // eth.BlobToKZGCommitment takes ONE blob per call (eth/eth.go:145-151): concurrent callers share batched launches.  A row is
        // the blob's 32-byte little-endian elements (read in place from the pinned staging buffer by the conversion kernel); a result
        // row is the 48 compressed bytes + the "invalid element" flag of BlobToPolynomial.
        const uint64_t n = eth->n;
        coalescer *co = get_coalescer(eth->fs, eth->co_blob, n * 32, 64);
        auto exec = [eth, n](coalesce_buf &b, uint64_t rows) -> int {
            hipSetDevice(eth->fs->device);
            hipStream_t s = b.stream;
            drain_on_exit drain(s);
            std::shared_lock<std::shared_mutex> tl(eth->ks->tab_mu);
            { dev_guard g(eth->fs); CHK(ensure_fixed_table(eth->ks, s)); }
            dtmp<uint8_t> d_c(s); dtmp<fr> d_poly(s); dtmp<g1j> d_out(s); dtmp<uint32_t> d_bad(s);
            CHK(d_c.alloc(rows * 48)); CHK(d_poly.alloc(rows * n)); CHK(d_out.alloc(rows)); CHK(d_bad.alloc(rows));
            HIPCHK(hipMemsetAsync(d_bad.p, 0, rows * 4, s));
            void *dp_in = nullptr;
            HIPCHK(hipHostGetDevicePointer(&dp_in, b.h_in, 0));
            launch_fr_from_le32(s, (const uint8_t *)dp_in, d_poly.p, n, rows, d_bad.p);
            CHK(commit_rows(eth->ks, s, d_poly.p, n, rows, d_out.p, 0, false));     // (internal-domain points: straight into the compression)
            launch_g1_compress(s, d_out.p, d_c.p, rows);
            HIPCHK(hipGetLastError());
            return eth_rows_to_host(s, d_c.p, nullptr, d_bad.p, rows, 64, b.h_out);
        };
        uint8_t row[64];
        int st = co->submit(blobs_le32, n * 32, n, 0, row, 64, exec, KZG_HIP_ERR_HIP);
        if (st != KZG_HIP_OK) return st;
        uint32_t bad; memcpy(&bad, row + 48, 4);
        ok[0] = bad ? 0 : 1;
        if (bad) memset(out48, 0, 48); else memcpy(out48, row, 48);
        return KZG_HIP_OK;
    }
    dev_guard g(eth->fs);
    hipStream_t s = eth->fs->stream;
    uint64_t n = eth->n;
    dtmp<uint8_t> d_in(s), d_c(s); dtmp<fr> d_poly(s); dtmp<g1j> d_out(s); dtmp<uint32_t> d_bad(s);
    CHK(d_c.alloc(batch * 48)); CHK(d_poly.alloc(batch * n)); CHK(d_out.alloc(batch)); CHK(d_bad.alloc(batch));
    HIPCHK(hipMemsetAsync(d_bad.p, 0, batch * 4, s));
    const uint8_t *src = (const uint8_t *)host_mapped_pointer(blobs_le32, (size_t)batch * n * 32);     // pinned blobs (kzg_hip_host_register): the conversion kernel reads them in place over PCIe
    if (!src) {
        CHK(d_in.alloc(batch * n * 32));
        CHK(h2d_copy(d_in.p, blobs_le32, batch * n * 32, s));
        src = d_in.p;
    }
    launch_fr_from_le32(s, src, d_poly.p, n, batch, d_bad.p);                    // BlobToPolynomial, eth/helpers.go:264-273
    CHK(commit_rows(eth->ks, s, d_poly.p, n, batch, d_out.p, 0, false));        // PolynomialToKZGCommitment, eth/helpers.go:98-103 (internal-domain points: compress wants those)
    launch_g1_compress(s, d_out.p, d_c.p, batch);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> bad(batch);
    HIPCHK(hipMemcpyAsync(bad.data(), d_bad.p, batch * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out48, d_c.p, batch * 48, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (uint64_t b = 0; b < batch; b++) {
        ok[b] = bad[b] ? 0 : 1;
        if (bad[b]) memset((uint8_t *)out48 + 48 * b, 0, 48);
    }
    return KZG_HIP_OK;
    KZG_CATCH
}
// ComputeKZGProof over resident rows (eth/helpers.go:179-203): quotients in evaluation form, their commitment over the Lagrange setup, 48-byte
// compression.  No host round trip in between: a row whose z lies in the domain gets bad[row] = 1 and a zero quotient.
static int eth_proof_rows(kzg_hip_eth *eth, hipStream_t s, const fr *d_poly, uint64_t poly_stride, const fr *d_z, uint64_t z_stride, uint64_t batch, uint8_t *d_out48,
                          fr *d_y, uint32_t *d_bad) {
    const uint64_t n = eth->n;
    dtmp<fr> d_q(s); dtmp<g1j> d_out(s);
    const uint64_t extra = eth_quotient_scratch_elems(n, batch);   // the row-split quotient of small batches keeps its shares behind the quotients: no allocation of its own
    CHK(d_q.alloc(batch * n + extra)); CHK(d_out.alloc(batch));
    HIPCHK(hipMemsetAsync(d_bad, 0, batch * 4, s));
    launch_eth_quotient(s, d_poly, poly_stride, eth->d_domain, n, batch, d_z, z_stride, eth->fs->d_inv_pow2 + ilog2(n), d_q.p, d_y, d_bad, 1, extra ? d_q.p + batch * n : nullptr);
    CHK(commit_rows(eth->ks, s, d_q.p, n, batch, d_out.p, 0, false));            // bls.LinCombG1(kzgSetupLagrange, quotient), eth/helpers.go:199
    launch_g1_compress(s, d_out.p, d_out48, batch);
    HIPCHK(hipGetLastError());
    return KZG_HIP_OK;
}
int kzg_hip_eth_compute_kzg_proof_batch_dev(kzg_hip_eth *eth, const void *d_polys_fr, uint64_t n, uint64_t batch, const void *d_zs_fr, void *d_out48, void *d_ys_fr,
                                            void *d_bad_u32, void *stream) {
    if (!eth || !d_out48 || !d_bad_u32) return KZG_HIP_ERR_BAD_ARG;
    if (n != eth->n) return KZG_HIP_ERR_LEN_MISMATCH;
    if (!batch) return KZG_HIP_OK;
    if (!d_polys_fr || !d_zs_fr) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    hipStream_t s = (hipStream_t)stream;
    std::shared_lock<std::shared_mutex> tl(eth->ks->tab_mu);
    { dev_guard g(eth->fs); CHK(ensure_fixed_table(eth->ks, s)); }
    dtmp<fr> d_y(s);
    fr *yp = (fr *)d_ys_fr;
    if (!yp) { CHK(d_y.alloc(batch)); yp = d_y.p; }
    return eth_proof_rows(eth, s, (const fr *)d_polys_fr, n, (const fr *)d_zs_fr, 1, batch, (uint8_t *)d_out48, yp, (uint32_t *)d_bad_u32);
    KZG_CATCH
}
int kzg_hip_eth_compute_kzg_proof_batch(kzg_hip_eth *eth, const void *polys_fr, uint64_t n, uint64_t batch, const void *zs_fr, void *out48, void *ys_fr, uint8_t *ok) {
    if (!eth || !out48 || !ok) return KZG_HIP_ERR_BAD_ARG;
    if (n != eth->n) return KZG_HIP_ERR_LEN_MISMATCH;                            // "polynomial has invalid length", eth/helpers.go:186-188
    if (!batch) return KZG_HIP_OK;
    if (!polys_fr || !zs_fr) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    dev_guard g(eth->fs);
    hipStream_t s = eth->fs->stream;
    dtmp<fr> d_poly(s), d_z(s), d_y(s); dtmp<uint8_t> d_c(s); dtmp<uint32_t> d_bad(s);
    CHK(d_poly.alloc(batch * n)); CHK(d_z.alloc(batch)); CHK(d_y.alloc(batch)); CHK(d_c.alloc(batch * 48)); CHK(d_bad.alloc(batch));
    HIPCHK(hipMemcpyAsync(d_poly.p, polys_fr, batch * n * sizeof(fr), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_z.p, zs_fr, batch * sizeof(fr), hipMemcpyHostToDevice, s));
    CHK(ensure_fixed_table(eth->ks, s));
    CHK(eth_proof_rows(eth, s, d_poly.p, n, d_z.p, 1, batch, d_c.p, d_y.p, d_bad.p));
    std::vector<uint32_t> bad(batch);
    HIPCHK(hipMemcpyAsync(bad.data(), d_bad.p, batch * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out48, d_c.p, batch * 48, hipMemcpyDeviceToHost, s));
    if (ys_fr) HIPCHK(hipMemcpyAsync(ys_fr, d_y.p, batch * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (uint64_t b = 0; b < batch; b++) {
        ok[b] = bad[b] ? 0 : 1;
        if (bad[b]) { memset((uint8_t *)out48 + 48 * b, 0, 48); if (ys_fr) memset((uint8_t *)ys_fr + 32 * b, 0, 32); }
    }
    return KZG_HIP_OK;
    KZG_CATCH
}
int kzg_hip_eth_compute_kzg_proof(kzg_hip_eth *eth, const void *poly_fr, uint64_t n, const void *z_fr, void *out48, void *y_fr) {
    if (!eth || !poly_fr || !z_fr || !out48) return KZG_HIP_ERR_BAD_ARG;
    if (n != eth->n) return KZG_HIP_ERR_LEN_MISMATCH;                            // "polynomial has invalid length", eth/helpers.go:186-188
    KZG_TRY
    uint8_t row[128];
    if (!coalescing_enabled()) {
        uint8_t ok = 0;
        CHK(kzg_hip_eth_compute_kzg_proof_batch(eth, poly_fr, n, 1, z_fr, row, row + 48, &ok));
        if (!ok) return KZG_HIP_ERR_BAD_ARG;
    } else {
        // eth.ComputeKZGProof takes ONE polynomial per call: concurrent callers share batched launches.  A request's row is its polynomial
        // followed by z (read in place from the pinned staging buffer); a result row is 48 proof bytes | y | the "invalid z" flag.
        coalescer *co = get_coalescer(eth->fs, eth->co_proof, (n + 1) * sizeof(fr), 128);
        auto exec = [eth, n, co](coalesce_buf &b, uint64_t rows) -> int {
            hipSetDevice(eth->fs->device);
            hipStream_t s = b.stream;
            drain_on_exit drain(s);
            std::shared_lock<std::shared_mutex> tl(eth->ks->tab_mu);
            { dev_guard g(eth->fs); CHK(ensure_fixed_table(eth->ks, s)); }
            dtmp<uint8_t> d_c(s); dtmp<fr> d_y(s); dtmp<uint32_t> d_bad(s);
            CHK(d_c.alloc(rows * 48)); CHK(d_y.alloc(rows)); CHK(d_bad.alloc(rows));
            void *dp_in = nullptr;
            HIPCHK(hipHostGetDevicePointer(&dp_in, b.h_in, 0));
            const uint64_t stride = co->in_row_bytes() / sizeof(fr);
            // The rows are read in place over PCIe (each coefficient twice: the quotient's two passes).  KZG_HIP_ETH_STAGE_ROWS=k copies batches of up to k rows to HBM first
            // (A/B hook: measured 0.314 against 0.309 ms in place for a lone call -- the copy costs what it saves).
            dtmp<fr> d_rows(s);
            const fr *src = (const fr *)dp_in;
            static const uint64_t stage_rows = [] { const char *e = getenv("KZG_HIP_ETH_STAGE_ROWS"); return e ? (uint64_t)atol(e) : 0ull; }();
            if (rows <= stage_rows) {
                CHK(d_rows.alloc(rows * stride));
                HIPCHK(hipMemcpyAsync(d_rows.p, b.h_in, rows * co->in_row_bytes(), hipMemcpyHostToDevice, s));
                src = d_rows.p;
            }
            CHK(eth_proof_rows(eth, s, src, stride, src + n, stride, rows, d_c.p, d_y.p, d_bad.p));
            return eth_rows_to_host(s, d_c.p, d_y.p, d_bad.p, rows, 128, b.h_out);
        };
        int st = co->submit(poly_fr, n * sizeof(fr), n, 0, row, 128, exec, KZG_HIP_ERR_HIP, z_fr, sizeof(fr));
        if (st != KZG_HIP_OK) return st;
        uint32_t bad; memcpy(&bad, row + 80, 4);
        if (bad) return KZG_HIP_ERR_BAD_ARG;                                     // "invalid z challenge", eth/helpers.go:190-192
    }
    memcpy(out48, row, 48);
    if (y_fr) memcpy(y_fr, row + 48, sizeof(fr));
    return KZG_HIP_OK;
    KZG_CATCH
}

// ---------------------------------------------------------------------------------------------------------
// eth.ComputeAggregateKZGProof / the prover-side pieces of eth.VerifyAggregateKZGProof (eth/eth.go:155-182, eth/helpers.go:137-176,215-260):
// the block-level caller of the commitment path.  Device: blobs -> polynomials -> commitments -> aggregated polynomial -> proof; host: the
// Fiat-Shamir transcript (one SHA-256 chain over every blob of the block), hashed while the device commits.
// ---------------------------------------------------------------------------------------------------------
// hashToBLSField (eth/helpers.go:113-133): SHA-256, digest read as a little-endian integer, reduced mod r
static fr hash_to_bls_field(const uint8_t *input, size_t len) {
    sha256 h;
    h.update(input, len);
    uint8_t d[32];
    h.final(d);
    uint64_t v[4], m[4];
    memcpy(v, d, 32);
    for (int i = 0; i < 4; i++) m[i] = (uint64_t)FrP::mod(2 * i) | (uint64_t)FrP::mod(2 * i + 1) << 32;
    for (int k = 0; k < 3; k++) {                                   // 2^256 < 3 r
        bool ge = true;
        for (int i = 3; i >= 0; i--) { if (v[i] != m[i]) { ge = v[i] > m[i]; break; } }
        if (!ge) break;
        unsigned __int128 br = 0;
        for (int i = 0; i < 4; i++) { unsigned __int128 t = (unsigned __int128)v[i] - m[i] - (uint64_t)br; v[i] = (uint64_t)t; br = (t >> 64) & 1; }
    }
    fr c;
    memcpy(c.l, v, 32);
    return to_mont<FrP>(c);
}
// ComputeAggregatedPolyAndCommitment (eth/helpers.go:137-162) up to the aggregated polynomial: BlobsToPolynomials (:275-285), the commitments
// (taken from `comm_in`, or PolynomialToKZGCommitment of every blob, :166-169, into `comm`), ComputeChallenges (:215-232), bls.PolyLinComb.
// Leaves polynomials, powers (device + host) and the aggregated polynomial resident; z_out = the evaluation challenge.
static int eth_aggregate(kzg_hip_eth *eth, hipStream_t s, const uint8_t *blobs, const uint8_t *comm_in, uint64_t batch, dtmp<fr> &d_poly, dtmp<fr> &d_agg,
                         dtmp<fr> &d_pow, std::vector<fr> &pw, std::vector<uint8_t> &comm, fr &z_out) {
    const uint64_t n = eth->n;
    CHK(d_agg.alloc(n));
    std::vector<uint32_t> bad(batch);
    drain_on_exit drain(s);                                          // an error return must not leave a copy into `bad` in flight
    dtmp<uint8_t> d_in(s), d_c(s); dtmp<g1j> d_out(s); dtmp<uint32_t> d_bad(s);
    if (batch) {
        CHK(d_in.alloc(batch * n * 32)); CHK(d_poly.alloc(batch * n)); CHK(d_bad.alloc(batch)); CHK(d_pow.alloc(batch));
        HIPCHK(hipMemsetAsync(d_bad.p, 0, batch * 4, s));
        HIPCHK(hipMemcpyAsync(d_in.p, blobs, batch * n * 32, hipMemcpyHostToDevice, s));
        launch_fr_from_le32(s, d_in.p, d_poly.p, n, batch, d_bad.p);
        HIPCHK(hipMemcpyAsync(bad.data(), d_bad.p, batch * 4, hipMemcpyDeviceToHost, s));
        if (!comm_in) {
            comm.resize(batch * 48);
            CHK(d_c.alloc(batch * 48)); CHK(d_out.alloc(batch));
            CHK(commit_rows(eth->ks, s, d_poly.p, n, batch, d_out.p, 0, false));
            launch_g1_compress(s, d_out.p, d_c.p, batch);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(comm.data(), d_c.p, batch * 48, hipMemcpyDeviceToHost, s));
        }
    }
    // hashPolysComms (eth/helpers.go:235-260), while the device works: FrTo32 of a valid element is the blob's own 32 bytes
    sha256 h;
    h.update("FSBLOBVERIFY_V1_", 16);
    h.update_u64_le(n);                                              // FieldElementsPerBlob
    h.update_u64_le(batch);
    if (batch) h.update(blobs, batch * n * 32);
    HIPCHK(hipStreamSynchronize(s));
    for (uint64_t b = 0; b < batch; b++) if (bad[b]) return KZG_HIP_ERR_BAD_BLOB;   // "could not convert blobs to polynomials"
    if (batch) h.update(comm_in ? comm_in : comm.data(), batch * 48);
    uint8_t tr[33];
    h.final(tr);
    tr[32] = 0;
    const fr r_chal = hash_to_bls_field(tr, 33);                     // linCombChallenge
    tr[32] = 1;
    z_out = hash_to_bls_field(tr, 33);                               // evalChallenge
    pw.resize(batch);                                                // ComputePowers (eth/helpers.go:87-96)
    fr cur = one<FrP>();
    for (uint64_t i = 0; i < batch; i++) { pw[i] = cur; cur = mul(cur, r_chal); }
    if (batch) {
        HIPCHK(hipMemcpyAsync(d_pow.p, pw.data(), batch * sizeof(fr), hipMemcpyHostToDevice, s));
        launch_poly_lincomb(s, d_poly.p, n, d_pow.p, batch, n, d_agg.p);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemsetAsync(d_agg.p, 0, n * sizeof(fr), s));       // PolyLinComb of no vector: zeros (bls/globals.go:157-159)
    }
    return KZG_HIP_OK;
}
int kzg_hip_eth_compute_aggregate_kzg_proof(kzg_hip_eth *eth, const void *blobs_le32, uint64_t batch, void *out_proof48, void *out_commitments48) {
    if (!eth || !out_proof48 || (batch && !blobs_le32)) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    dev_guard g(eth->fs);
    hipStream_t s = eth->fs->stream;
    std::vector<fr> pw; std::vector<uint8_t> comm; fr z;
    uint32_t bad = 0;
    uint8_t proof[48];
    drain_on_exit drain(s);                                          // declared after the host buffers the stream copies from / into
    CHK(ensure_fixed_table(eth->ks, s));
    dtmp<fr> d_poly(s), d_agg(s), d_pow(s), d_z(s), d_y(s); dtmp<uint8_t> d_c(s); dtmp<uint32_t> d_bad(s);
    CHK(eth_aggregate(eth, s, (const uint8_t *)blobs_le32, nullptr, batch, d_poly, d_agg, d_pow, pw, comm, z));
    CHK(d_z.alloc(1)); CHK(d_y.alloc(1)); CHK(d_c.alloc(48)); CHK(d_bad.alloc(1));
    HIPCHK(hipMemcpyAsync(d_z.p, &z, sizeof(fr), hipMemcpyHostToDevice, s));
    CHK(eth_proof_rows(eth, s, d_agg.p, eth->n, d_z.p, 1, 1, d_c.p, d_y.p, d_bad.p));   // ComputeKZGProof(aggregatedPoly, evaluationChallenge), eth/helpers.go:175
    HIPCHK(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(proof, d_c.p, 48, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (bad) return KZG_HIP_ERR_BAD_ARG;                             // "invalid z challenge"
    memcpy(out_proof48, proof, 48);
    if (out_commitments48 && batch) memcpy(out_commitments48, comm.data(), batch * 48);
    return KZG_HIP_OK;
    KZG_CATCH
}
int kzg_hip_eth_compute_aggregated_poly_and_commitment(kzg_hip_eth *eth, const void *blobs_le32, const void *commitments48, uint64_t batch, void *out_poly_fr,
                                                       void *out_commitment_g1, void *out_z_fr, void *out_y_fr) {
    if (!eth || !out_commitment_g1 || !out_z_fr || (batch && (!blobs_le32 || !commitments48))) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    dev_guard g(eth->fs);
    hipStream_t s = eth->fs->stream;
    const uint64_t n = eth->n;
    std::vector<fr> pw; std::vector<uint8_t> comm; fr z, y;
    g1j agg_c;
    uint32_t flag[2] = {0, 0};
    drain_on_exit drain(s);                                          // declared after the host buffers the stream copies from / into
    dtmp<fr> d_poly(s), d_agg(s), d_pow(s), d_z(s), d_y(s), d_q(s); dtmp<uint32_t> d_flag(s);
    CHK(eth_aggregate(eth, s, (const uint8_t *)blobs_le32, (const uint8_t *)commitments48, batch, d_poly, d_agg, d_pow, pw, comm, z));
    // aggregatedCommitmentG1 = LinCombG1(FromCompressedG1(commitments), powers), eth/helpers.go:149-160
    set_inf_image(&agg_c);
    CHK(d_flag.alloc(2));
    HIPCHK(hipMemsetAsync(d_flag.p, 0, 8, s));
    dtmp<g1j> d_pts(s), d_out(s); dtmp<g1a> d_tab(s); dtmp<uint8_t> d_cin(s), d_ws(s);
    if (batch) {
        msm_plan p = classic_plan(batch);
        if (!msm_index_range_ok(p, batch)) return KZG_HIP_ERR_TOO_WIDE;
        CHK(d_pts.alloc(batch)); CHK(d_out.alloc(1)); CHK(d_tab.alloc(batch)); CHK(d_cin.alloc(batch * 48)); CHK(d_ws.alloc(msm_workspace_bytes(p, batch, 1)));
        HIPCHK(hipMemcpyAsync(d_cin.p, commitments48, batch * 48, hipMemcpyHostToDevice, s));
        launch_g1_decompress(s, d_cin.p, d_pts.p, batch, d_flag.p);
        launch_g1_from_kilic(s, d_pts.p, batch);
        launch_g1_to_affine(s, d_pts.p, d_tab.p, batch);
        launch_msm(s, p, d_tab.p, d_pow.p, batch, batch, 1, d_ws.p, d_out.p, true);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&agg_c, d_out.p, sizeof(g1j), hipMemcpyDeviceToHost, s));
    }
    // y = EvaluatePolynomialInEvaluationForm(aggregatedPoly, evaluationChallenge) (eth/eth.go:166): the quotient kernel's first half
    const uint64_t extra = eth_quotient_scratch_elems(n, 1);
    CHK(d_z.alloc(1)); CHK(d_y.alloc(1)); CHK(d_q.alloc(n + extra));
    HIPCHK(hipMemcpyAsync(d_z.p, &z, sizeof(fr), hipMemcpyHostToDevice, s));
    launch_eth_quotient(s, d_agg.p, n, eth->d_domain, n, 1, d_z.p, 1, eth->fs->d_inv_pow2 + ilog2(n), d_q.p, d_y.p, d_flag.p + 1, 1, extra ? d_q.p + n : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(flag, d_flag.p, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&y, d_y.p, sizeof(fr), hipMemcpyDeviceToHost, s));
    if (out_poly_fr) HIPCHK(hipMemcpyAsync(out_poly_fr, d_agg.p, n * sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (flag[0]) return KZG_HIP_ERR_BAD_POINT;                       // FromCompressedG1 failed, eth/helpers.go:153-156
    if (flag[1]) memset(&y, 0, sizeof y);                            // evaluation challenge inside the domain (probability 2^-243): the reference's formula returns 0 there (bls/globals.go:141-152)
    memcpy(out_commitment_g1, &agg_c, sizeof(g1j));
    memcpy(out_z_fr, &z, sizeof(fr));
    if (out_y_fr) memcpy(out_y_fr, &y, sizeof(fr));
    return KZG_HIP_OK;
    KZG_CATCH
}
// bls.EvaluatePolyInEvaluationForm (bls/globals.go:106-153) with rootsOfUnity = the settings' ExpandedRootsOfUnity[:MaxWidth] (the form of
// fft_fr_test.go:73-99), and eth.EvaluatePolynomialInEvaluationForm (eth/helpers.go:207-211: DomainFr, scale 0): the first half of the quotient kernel
static int evaluate_in_evaluation_form(kzg_hip_fft *fs, const fr *d_roots, uint64_t root_stride, const void *poly_fr, uint64_t n, const void *x_fr, void *out_y_fr) {
    stream_lease lease(fs);
    hipStream_t s = lease.s;
    fr y; uint32_t flag = 0;
    drain_on_exit drain(s);
    dtmp<fr> d_poly(s), d_x(s), d_y(s), d_q(s); dtmp<uint32_t> d_flag(s);
    const uint64_t extra = eth_quotient_scratch_elems(n, 1);
    CHK(d_poly.alloc(n)); CHK(d_x.alloc(1)); CHK(d_y.alloc(1)); CHK(d_q.alloc(n + extra)); CHK(d_flag.alloc(1));
    HIPCHK(hipMemsetAsync(d_flag.p, 0, 4, s));
    HIPCHK(hipMemcpyAsync(d_poly.p, poly_fr, n * sizeof(fr), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_x.p, x_fr, sizeof(fr), hipMemcpyHostToDevice, s));
    launch_eth_quotient(s, d_poly.p, n, d_roots, n, 1, d_x.p, 1, fs->d_inv_pow2 + ilog2(n), d_q.p, d_y.p, d_flag.p, root_stride, extra ? d_q.p + n : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&y, d_y.p, sizeof(fr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&flag, d_flag.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (flag) memset(&y, 0, sizeof y);   // x inside the domain: the reference's last factor (x^W - 1) / W is zero there, so it returns 0 whatever the sum was (bls/globals.go:141-152)
    memcpy(out_y_fr, &y, sizeof(fr));
    return KZG_HIP_OK;
}
int kzg_hip_evaluate_poly_in_evaluation_form(kzg_hip_fft *fs, const void *poly_fr, uint64_t n, const void *x_fr, uint32_t scale, void *out_y_fr) {
    if (!fs || !poly_fr || !x_fr || !out_y_fr) return KZG_HIP_ERR_BAD_ARG;
    if (scale > 63 || n != fs->W >> scale || !n) return KZG_HIP_ERR_LEN_MISMATCH;   // "expected roots of unity ... to match polynomial size", bls/globals.go:107-109
    KZG_TRY
    return evaluate_in_evaluation_form(fs, fs->d_expanded, 1ull << scale, poly_fr, n, x_fr, out_y_fr);
    KZG_CATCH
}
int kzg_hip_eth_evaluate_polynomial_in_evaluation_form(kzg_hip_eth *eth, const void *poly_fr, uint64_t n, const void *x_fr, void *out_y_fr) {
    if (!eth || !poly_fr || !x_fr || !out_y_fr) return KZG_HIP_ERR_BAD_ARG;
    if (n != eth->n) return KZG_HIP_ERR_LEN_MISMATCH;
    KZG_TRY
    return evaluate_in_evaluation_form(eth->fs, eth->d_domain, 1, poly_fr, n, x_fr, out_y_fr);
    KZG_CATCH
}
// test hook: SHA-256 of a host buffer through the transcript's implementation (needs no device)
void kzg_hip_test_sha256(const void *data, uint64_t len, void *out32) {
    sha256 h;
    h.update(data, len);
    h.final((uint8_t *)out32);
}

// ---------------------------------------------------------------------------------------------------------
// setup (un)marshalling (row f4): G1Point.MarshalText / UnmarshalText (bls/bls_all.go:20-39) and the JSON trusted setup of
// eth/globals.go:33-49.  Hex coding and JSON scanning are host work; decompression + subgroup check run on the device.
// ---------------------------------------------------------------------------------------------------------
static int hex_nibble(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
int kzg_hip_g1_marshal_text(kzg_hip_fft *fs, const void *points_g1, uint64_t n, char *out_hex96) {
    if (!fs || (n && (!points_g1 || !out_hex96))) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    std::vector<uint8_t> raw(48 * n);
    CHK(kzg_hip_g1_to_compressed(fs, points_g1, n, raw.data()));
    static const char dig[] = "0123456789abcdef";                       // hex.EncodeToString: lower case, no 0x prefix
    for (uint64_t i = 0; i < 48 * n; i++) { out_hex96[2 * i] = dig[raw[i] >> 4]; out_hex96[2 * i + 1] = dig[raw[i] & 15]; }
    return KZG_HIP_OK;
    KZG_CATCH
}
int kzg_hip_g1_unmarshal_text(kzg_hip_fft *fs, const char *hex96, uint64_t n, void *out_g1) {
    if (!fs || (n && (!hex96 || !out_g1))) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    std::vector<uint8_t> raw(48 * n);
    for (uint64_t i = 0; i < 48 * n; i++) {
        int hi = hex_nibble(hex96[2 * i]), lo = hex_nibble(hex96[2 * i + 1]);
        if (hi < 0 || lo < 0) return KZG_HIP_ERR_BAD_POINT;             // hex.DecodeString error (bls/bls_all.go:29-32)
        raw[i] = (uint8_t)(hi << 4 | lo);
    }
    return kzg_hip_g1_from_compressed(fs, raw.data(), n, out_g1);
    KZG_CATCH
}
// finds "key" : [ "..." , ... ] in `js` and appends the decoded 48-byte strings; *found = 0 when the key is absent
static int json_hex48_array(const char *js, uint64_t len, const char *key, std::vector<uint8_t> &out, uint64_t *count, int *found) {
    *count = 0; *found = 0;
    std::string pat = std::string("\"") + key + "\"";
    const char *end = js + len, *p = js;
    for (;;) {                                                           // the key must be followed by ':' (skips "setup_G1" inside "setup_G1_lagrange")
        p = std::search(p, end, pat.begin(), pat.end());
        if (p == end) return KZG_HIP_OK;
        p += pat.size();
        const char *q = p;
        while (q < end && (*q == ' ' || *q == '\t' || *q == '\n' || *q == '\r')) q++;
        if (q < end && *q == ':') { p = q + 1; break; }
    }
    while (p < end && *p != '[') { if (*p != ' ' && *p != '\t' && *p != '\n' && *p != '\r') return KZG_HIP_ERR_BAD_ARG; p++; }
    if (p == end) return KZG_HIP_ERR_BAD_ARG;
    p++;
    *found = 1;
    for (;;) {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == ',')) p++;
        if (p == end) return KZG_HIP_ERR_BAD_ARG;
        if (*p == ']') return KZG_HIP_OK;
        if (*p != '"') return KZG_HIP_ERR_BAD_ARG;
        p++;
        const char *q = p;
        while (q < end && *q != '"') q++;
        if (q == end) return KZG_HIP_ERR_BAD_ARG;
        if (q - p != 96) return KZG_HIP_ERR_BAD_POINT;                   // FromCompressedG1 wants exactly 48 bytes
        for (int i = 0; i < 48; i++) {
            int hi = hex_nibble(p[2 * i]), lo = hex_nibble(p[2 * i + 1]);
            if (hi < 0 || lo < 0) return KZG_HIP_ERR_BAD_POINT;
            out.push_back((uint8_t)(hi << 4 | lo));
        }
        (*count)++;
        p = q + 1;
    }
}
int kzg_hip_trusted_setup_from_json(kzg_hip_fft *fs, const char *json, uint64_t json_len, void *out_setup_g1, void *out_lagrange_g1, uint64_t capacity,
                                    uint64_t *n_setup_g1, uint64_t *n_lagrange_g1) {
    if (!fs || !json || !n_setup_g1 || !n_lagrange_g1) return KZG_HIP_ERR_BAD_ARG;
    KZG_TRY
    std::vector<uint8_t> mono, lagr;
    int f1 = 0, f2 = 0;
    CHK(json_hex48_array(json, json_len, "setup_G1", mono, n_setup_g1, &f1));
    CHK(json_hex48_array(json, json_len, "setup_G1_lagrange", lagr, n_lagrange_g1, &f2));
    if (!f1 && !f2) return KZG_HIP_ERR_BAD_ARG;                          // not a trusted-setup document
    if (out_setup_g1 && *n_setup_g1) {
        if (*n_setup_g1 > capacity) return KZG_HIP_ERR_LEN_MISMATCH;
        CHK(kzg_hip_g1_from_compressed(fs, mono.data(), *n_setup_g1, out_setup_g1));
    }
    if (out_lagrange_g1 && *n_lagrange_g1) {
        if (*n_lagrange_g1 > capacity) return KZG_HIP_ERR_LEN_MISMATCH;
        CHK(kzg_hip_g1_from_compressed(fs, lagr.data(), *n_lagrange_g1, out_lagrange_g1));
    }
    return KZG_HIP_OK;
    KZG_CATCH
}

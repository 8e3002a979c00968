//go:build kzg_hip && !bignum_pure && !bignum_hol256 && !bignum_hbls
// +build kzg_hip,!bignum_pure,!bignum_hol256,!bignum_hbls

package kzg

/*
#include "kzg_hip.h"
*/
import "C"

import (
	"runtime"
	"unsafe"

	"github.com/protolambda/go-kzg/bls"
)

// MultiKZGSettings is new API surface: the prover side of a KZGSettings (kzg.go:11-36) replicated on every GPU of `devices`, behind ONE
// handle of the C library (kzg_hip_multi_*, include/kzg_hip.h).  The reference is a single-process library; this is how a Go caller uses the
// 8 GPUs of a node without starting 8 processes:
//
//   - the *Batch methods divide their polynomials among the devices (contiguous shares, results in input order; no collective);
//   - DAUsingFK20 / DAUsingFK20Multi on ONE polynomial shard the Toeplitz stage by output position and exchange the slices with
//     ncclAllGather (RCCL over xGMI, single-process communicators) -- see SetFFTSharding for what happens to the two G1 transforms.
//
// A device may be listed more than once (every entry gets its own settings and tables): that is how a 1-GPU box exercises these paths.
type MultiKZGSettings struct {
	h       *C.kzg_hip_multi
	Devices []int
}

// NewMultiKZGSettings builds NewFFTSettings(maxScale) and NewKZGSettings(fs, secretG1, ...) (prover side) on every listed device.
// Panics like NewKZGSettings does when the setup is shorter than the domain (kzg.go:25-27).
func NewMultiKZGSettings(devices []int, maxScale uint8, secretG1 []bls.G1Point) *MultiKZGSettings {
	if len(devices) == 0 {
		panic("NewMultiKZGSettings: empty device list")
	}
	devs := make([]C.int, len(devices))
	for i, d := range devices {
		devs[i] = C.int(d)
	}
	m := &MultiKZGSettings{Devices: append([]int(nil), devices...)}
	hipMust(C.kzg_hip_multi_settings_new(&devs[0], C.uint32_t(len(devs)), C.uint(maxScale), g1Ptr(secretG1), C.uint64_t(len(secretG1)), &m.h))
	runtime.SetFinalizer(m, (*MultiKZGSettings).Close)
	return m
}

// Close frees the settings and tables on every device (idempotent).  Close the FK20 settings built on this object first.
func (m *MultiKZGSettings) Close() {
	if m.h != nil {
		C.kzg_hip_multi_settings_free(m.h)
		m.h = nil
	}
}

// Transport reports how one-polynomial calls exchange their slices: "rccl" (ncclAllGather between distinct devices), "peer-copy"
// (hipMemcpyPeerAsync: the list repeats a device, or librccl could not be bound) or "host-staged" (through pinned host memory: the
// other two failed the exchange test the constructor runs) -- TransportNote says which and why.
func (m *MultiKZGSettings) Transport() string {
	defer runtime.KeepAlive(m)
	return C.GoString(C.kzg_hip_multi_transport(m.h))
}
func (m *MultiKZGSettings) TransportNote() string {
	defer runtime.KeepAlive(m)
	return C.GoString(C.kzg_hip_multi_transport_note(m.h))
}

// TransportSelfTest is the outcome of the exchange the constructor ran before returning (every device writes a pattern, one all-gather,
// every device verifies every byte): "ok: <transport>, ...".
func (m *MultiKZGSettings) TransportSelfTest() string {
	defer runtime.KeepAlive(m)
	return C.GoString(C.kzg_hip_multi_transport_check(m.h))
}

// SetFFTSharding: 0 = one all-gather of the hExtFFT slices, both G1 transforms on the first device; 1 = both transforms sharded by
// decimation as well (five all-gathers, SURVEY.md 8e); -1 = the default (sharded from 4 devices on).
func (m *MultiKZGSettings) SetFFTSharding(mode int) {
	defer runtime.KeepAlive(m)
	hipMust(C.kzg_hip_multi_set_fft_sharding(m.h, C.int(mode)))
}

// SetTableBudgetGB: KZGSettings.SetTableBudgetGB on every device.
func (m *MultiKZGSettings) SetTableBudgetGB(gb float64) {
	defer runtime.KeepAlive(m)
	hipMust(C.kzg_hip_multi_set_table_budget_gb(m.h, C.double(gb)))
}

func flattenRows(rows [][]bls.Fr, what string) ([]bls.Fr, int) {
	n := len(rows[0])
	flat := make([]bls.Fr, 0, n*len(rows))
	for _, c := range rows {
		if len(c) != n {
			panic(what + ": ragged batch")
		}
		flat = append(flat, c...)
	}
	return flat, n
}

// CommitToPolyBatch: CommitToPoly (kzg_single_proofs.go:17-19) on every row, rows divided among the devices.
func (m *MultiKZGSettings) CommitToPolyBatch(coeffs [][]bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(m)
	if len(coeffs) == 0 {
		return nil
	}
	flat, n := flattenRows(coeffs, "CommitToPolyBatch")
	out := make([]bls.G1Point, len(coeffs))
	hipMust(C.kzg_hip_multi_commit_to_poly_batch(m.h, frPtr(flat), C.uint64_t(n), C.uint64_t(len(coeffs)), g1Ptr(out)))
	return out
}

// ComputeProofSingleBatch: ComputeProofSingle (kzg_single_proofs.go:36-54) of polys[b] at xs[b], rows divided among the devices.
func (m *MultiKZGSettings) ComputeProofSingleBatch(polys [][]bls.Fr, xs []uint64) []bls.G1Point {
	defer runtime.KeepAlive(m)
	if len(polys) == 0 {
		return nil
	}
	if len(polys) != len(xs) {
		panic("ComputeProofSingleBatch: len(polys) != len(xs)")
	}
	flat, n := flattenRows(polys, "ComputeProofSingleBatch")
	out := make([]bls.G1Point, len(polys))
	hipMust(C.kzg_hip_multi_compute_proof_single_batch(m.h, frPtr(flat), C.uint64_t(n), C.uint64_t(len(polys)), (*C.uint64_t)(unsafe.Pointer(&xs[0])), g1Ptr(out)))
	return out
}

// FFTBatch: FFT (fft_fr.go:55-74) on every row (rows of a power-of-two length), rows divided among the devices.
func (m *MultiKZGSettings) FFTBatch(rows [][]bls.Fr, inv bool) [][]bls.Fr {
	defer runtime.KeepAlive(m)
	if len(rows) == 0 {
		return nil
	}
	flat, n := flattenRows(rows, "FFTBatch")
	outFlat := make([]bls.Fr, len(flat))
	hipMust(C.kzg_hip_multi_fft_fr_batch(m.h, frPtr(flat), C.uint64_t(n), C.uint64_t(len(rows)), cBool(inv), frPtr(outFlat)))
	out := make([][]bls.Fr, len(rows))
	for b := range out {
		out[b] = outFlat[n*b : n*(b+1)]
	}
	return out
}

// FFTG1BatchFlat: FFTG1 (fft_g1.go:58-94) on len(points) / n rows stored back to back, rows divided among the devices.
func (m *MultiKZGSettings) FFTG1BatchFlat(points []bls.G1Point, n int, inv bool) []bls.G1Point {
	defer runtime.KeepAlive(m)
	if n <= 0 || len(points)%n != 0 {
		panic("FFTG1BatchFlat: len(points) is not a multiple of n")
	}
	out := make([]bls.G1Point, len(points))
	if len(points) > 0 {
		hipMust(C.kzg_hip_multi_fft_g1_batch(m.h, g1Ptr(points), C.uint64_t(n), C.uint64_t(len(points)/n), cBool(inv), g1Ptr(out)))
	}
	return out
}

// DASFFTExtensionBatch: DASFFTExtension (das_extension.go:71-84) on every row, in place like the reference.
func (m *MultiKZGSettings) DASFFTExtensionBatch(rows [][]bls.Fr) {
	defer runtime.KeepAlive(m)
	if len(rows) == 0 {
		return
	}
	flat, n := flattenRows(rows, "DASFFTExtensionBatch")
	hipMust(C.kzg_hip_multi_das_fft_extension_batch(m.h, frPtr(flat), C.uint64_t(n), C.uint64_t(len(rows))))
	for b := range rows {
		copy(rows[b], flat[n*b:n*(b+1)])
	}
}

// MultiEthSettings: the device side of package eth (bit-reversed Lagrange setup, DomainFr; eth/globals.go:39-72) on every device.  Blobs are
// n x 32 little-endian bytes (eth.Blob), commitments and proofs 48 bytes; package eth wraps these with its own types.
type MultiEthSettings struct {
	h *C.kzg_hip_multi_eth
	m *MultiKZGSettings
	n int
}

// NewMultiEthSettings takes setup_G1_lagrange in NATURAL order (as eth/trusted_setup.json stores it).
func NewMultiEthSettings(m *MultiKZGSettings, lagrangeNaturalOrder []bls.G1Point) *MultiEthSettings {
	defer runtime.KeepAlive(m)
	e := &MultiEthSettings{m: m, n: len(lagrangeNaturalOrder)}
	hipMust(C.kzg_hip_multi_eth_settings_new(m.h, g1Ptr(lagrangeNaturalOrder), C.uint64_t(len(lagrangeNaturalOrder)), &e.h))
	runtime.SetFinalizer(e, (*MultiEthSettings).Close)
	return e
}
func (e *MultiEthSettings) Close() {
	if e.h != nil {
		C.kzg_hip_multi_eth_settings_free(e.h)
		e.h = nil
	}
}

// BlobsToKZGCommitments: eth.BlobToKZGCommitment (eth/eth.go:145-151) on every blob (len(blobs) = count x n x 32 bytes), blobs divided among the devices;
// valid[b] == false marks a blob with a field element >= r (its commitment is zeroed).
func (e *MultiEthSettings) BlobsToKZGCommitments(blobs []byte) (commitments [][48]byte, valid []bool) {
	defer runtime.KeepAlive(e)
	count := len(blobs) / (e.n * 32)
	if count == 0 {
		return nil, nil
	}
	commitments = make([][48]byte, count)
	ok := make([]C.uint8_t, count)
	hipMust(C.kzg_hip_multi_eth_blob_to_kzg_commitment_batch(e.h, unsafe.Pointer(&blobs[0]), C.uint64_t(count), unsafe.Pointer(&commitments[0]), &ok[0]))
	valid = make([]bool, count)
	for i := range valid {
		valid[i] = ok[i] != 0
	}
	return commitments, valid
}

// ComputeKZGProofs: eth.ComputeKZGProof (eth/helpers.go:179-203) of polynomials[b] (evaluation form) at zs[b]; valid[b] == false is that row's "invalid z challenge".
func (e *MultiEthSettings) ComputeKZGProofs(polynomials [][]bls.Fr, zs []bls.Fr) (proofs [][48]byte, valid []bool) {
	defer runtime.KeepAlive(e)
	if len(polynomials) != len(zs) {
		panic("ComputeKZGProofs: len(polynomials) != len(zs)")
	}
	if len(polynomials) == 0 {
		return nil, nil
	}
	flat, n := flattenRows(polynomials, "ComputeKZGProofs")
	proofs = make([][48]byte, len(polynomials))
	ok := make([]C.uint8_t, len(polynomials))
	hipMust(C.kzg_hip_multi_eth_compute_kzg_proof_batch(e.h, frPtr(flat), C.uint64_t(n), C.uint64_t(len(polynomials)), frPtr(zs), unsafe.Pointer(&proofs[0]), nil, &ok[0]))
	valid = make([]bool, len(polynomials))
	for i := range valid {
		valid[i] = ok[i] != 0
	}
	return proofs, valid
}

// MultiFK20SingleSettings: NewFK20SingleSettings (kzg.go:43-64) on every device of a MultiKZGSettings.
type MultiFK20SingleSettings struct {
	h *C.kzg_hip_multi_fk20s
	m *MultiKZGSettings // keeps the parent (and its device handles) alive
}

func NewMultiFK20SingleSettings(m *MultiKZGSettings, n2 uint64) *MultiFK20SingleSettings {
	defer runtime.KeepAlive(m)
	fk := &MultiFK20SingleSettings{m: m}
	hipMust(C.kzg_hip_multi_fk20_single_settings_new(m.h, C.uint64_t(n2), &fk.h))
	runtime.SetFinalizer(fk, (*MultiFK20SingleSettings).Close)
	return fk
}
func (fk *MultiFK20SingleSettings) Close() {
	if fk.h != nil {
		C.kzg_hip_multi_fk20_single_settings_free(fk.h)
		fk.h = nil
	}
}

// DAUsingFK20 (fk20_single.go:176-196) of ONE polynomial over all devices.
func (fk *MultiFK20SingleSettings) DAUsingFK20(polynomial []bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(fk)
	out := make([]bls.G1Point, 2*len(polynomial))
	hipMust(C.kzg_hip_multi_da_using_fk20(fk.h, frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

// DAUsingFK20Batch: DAUsingFK20 on every row, rows divided among the devices; out[b] holds the 2n proofs of polynomials[b].
func (fk *MultiFK20SingleSettings) DAUsingFK20Batch(polynomials [][]bls.Fr) [][]bls.G1Point {
	defer runtime.KeepAlive(fk)
	if len(polynomials) == 0 {
		return nil
	}
	flat, n := flattenRows(polynomials, "DAUsingFK20Batch")
	proofs := make([]bls.G1Point, 2*n*len(polynomials))
	hipMust(C.kzg_hip_multi_da_using_fk20_batch(fk.h, frPtr(flat), C.uint64_t(n), C.uint64_t(len(polynomials)), g1Ptr(proofs)))
	out := make([][]bls.G1Point, len(polynomials))
	for b := range out {
		out[b] = proofs[2*n*b : 2*n*(b+1)]
	}
	return out
}

// MultiFK20MultiSettings: NewFK20MultiSettings (kzg.go:73-116) on every device of a MultiKZGSettings.
type MultiFK20MultiSettings struct {
	h        *C.kzg_hip_multi_fk20m
	m        *MultiKZGSettings
	chunkLen uint64
}

func NewMultiFK20MultiSettings(m *MultiKZGSettings, n2 uint64, chunkLen uint64) *MultiFK20MultiSettings {
	defer runtime.KeepAlive(m)
	fk := &MultiFK20MultiSettings{m: m, chunkLen: chunkLen}
	hipMust(C.kzg_hip_multi_fk20_multi_settings_new(m.h, C.uint64_t(n2), C.uint64_t(chunkLen), &fk.h))
	runtime.SetFinalizer(fk, (*MultiFK20MultiSettings).Close)
	return fk
}
func (fk *MultiFK20MultiSettings) Close() {
	if fk.h != nil {
		C.kzg_hip_multi_fk20_multi_settings_free(fk.h)
		fk.h = nil
	}
}

// DAUsingFK20Multi (fk20_multi.go:113-133) of ONE polynomial over all devices: the sharded form of BASELINE config 5.
func (fk *MultiFK20MultiSettings) DAUsingFK20Multi(polynomial []bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(fk)
	out := make([]bls.G1Point, 2*uint64(len(polynomial))/fk.chunkLen)
	hipMust(C.kzg_hip_multi_da_using_fk20_multi(fk.h, frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

// DAUsingFK20MultiBatch: rows divided among the devices.
func (fk *MultiFK20MultiSettings) DAUsingFK20MultiBatch(polynomials [][]bls.Fr) [][]bls.G1Point {
	defer runtime.KeepAlive(fk)
	if len(polynomials) == 0 {
		return nil
	}
	flat, n := flattenRows(polynomials, "DAUsingFK20MultiBatch")
	per := 2 * uint64(n) / fk.chunkLen
	proofs := make([]bls.G1Point, per*uint64(len(polynomials)))
	hipMust(C.kzg_hip_multi_da_using_fk20_multi_batch(fk.h, frPtr(flat), C.uint64_t(n), C.uint64_t(len(polynomials)), g1Ptr(proofs)))
	out := make([][]bls.G1Point, len(polynomials))
	for b := range out {
		out[b] = proofs[per*uint64(b) : per*uint64(b+1)]
	}
	return out
}

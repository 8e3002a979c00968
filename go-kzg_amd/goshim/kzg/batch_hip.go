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

// Batch forms: new API surface beside the reference's one-polynomial methods.  One call = one launch chain over all rows, which is what
// fills 256 CUs (a lone G1 transform has 2048 butterflies per stage; the GPU has 65 536 lanes per round).  Rows must have equal lengths.

// HipDeviceCount is the number of usable gfx950 devices (0: every constructor of this build panics -- there is no CPU fallback).
func HipDeviceCount() int { return int(C.kzg_hip_device_count()) }

// HipVersion identifies the loaded library.
func HipVersion() string { return C.GoString(C.kzg_hip_version()) }

// PinnedFr keeps a []bls.Fr pinned for the GPU (kzg_hip_host_register): batch calls whose input lies inside it read it in place over PCIe instead of
// staging a copy (CommitToPolyBatchFlat on one GPU: 67-78 k -> ~95 k commitments/s).  The Go runtime must not move the slice meanwhile: runtime.Pinner.
type PinnedFr struct {
	Values []bls.Fr
	pin    runtime.Pinner
}

func PinFr(values []bls.Fr) *PinnedFr {
	p := &PinnedFr{Values: values}
	if len(values) == 0 {
		return p
	}
	p.pin.Pin(&values[0])
	hipMust(C.kzg_hip_host_register(frPtr(values), C.uint64_t(len(values))*C.uint64_t(unsafe.Sizeof(values[0]))))
	return p
}
func (p *PinnedFr) Release() {
	if len(p.Values) > 0 {
		hipMust(C.kzg_hip_host_unregister(frPtr(p.Values)))
		p.pin.Unpin()
		p.Values = nil
	}
}

// CommitToPolyBatchFlat: CommitToPoly on len(coeffs) / n polynomials stored back to back (no flattening copy: what a pinned buffer is for).
func (ks *KZGSettings) CommitToPolyBatchFlat(coeffs []bls.Fr, n int) []bls.G1Point {
	defer runtime.KeepAlive(ks)
	if n <= 0 || len(coeffs)%n != 0 {
		panic("CommitToPolyBatchFlat: len(coeffs) is not a multiple of n")
	}
	out := make([]bls.G1Point, len(coeffs)/n)
	if len(out) > 0 {
		hipMust(C.kzg_hip_commit_to_poly_batch(ks.hip(), frPtr(coeffs), C.uint64_t(n), C.uint64_t(len(out)), g1Ptr(out)))
	}
	return out
}

// FFTBatch: FFT (fft_fr.go:55-74) on every row; rows of a power-of-two length.
func (fs *FFTSettings) FFTBatch(rows [][]bls.Fr, inv bool) ([][]bls.Fr, error) {
	defer runtime.KeepAlive(fs)
	if len(rows) == 0 {
		return nil, nil
	}
	flat, n := flattenRows(rows, "FFTBatch")
	outFlat := make([]bls.Fr, len(flat))
	st := C.kzg_hip_fft_fr_batch(fs.hip(), frPtr(flat), C.uint64_t(n), C.uint64_t(len(rows)), cBool(inv), frPtr(outFlat))
	if err := hipErr(st, n, fs.MaxWidth); err != nil {
		return nil, err
	}
	out := make([][]bls.Fr, len(rows))
	for b := range out {
		out[b] = outFlat[n*b : n*(b+1)]
	}
	return out, nil
}

// FFTG1Batch: FFTG1 (fft_g1.go:58-94) on every row (a lone 4096-point transform is latency-bound at 6.8 ms; in a batch of 64 one costs 0.46 ms).
func (fs *FFTSettings) FFTG1Batch(rows [][]bls.G1Point, inv bool) ([][]bls.G1Point, error) {
	defer runtime.KeepAlive(fs)
	if len(rows) == 0 {
		return nil, nil
	}
	n := len(rows[0])
	flat := make([]bls.G1Point, 0, n*len(rows))
	for _, r := range rows {
		if len(r) != n {
			panic("FFTG1Batch: ragged batch")
		}
		flat = append(flat, r...)
	}
	outFlat := make([]bls.G1Point, len(flat))
	st := C.kzg_hip_fft_g1_batch(fs.hip(), g1Ptr(flat), C.uint64_t(n), C.uint64_t(len(rows)), cBool(inv), g1Ptr(outFlat))
	if err := hipErr(st, n, fs.MaxWidth); err != nil {
		return nil, err
	}
	out := make([][]bls.G1Point, len(rows))
	for b := range out {
		out[b] = outFlat[n*b : n*(b+1)]
	}
	return out, nil
}

// DASFFTExtensionBatch: DASFFTExtension (das_extension.go:71-84) on every row, in place like the reference.
func (fs *FFTSettings) DASFFTExtensionBatch(rows [][]bls.Fr) {
	defer runtime.KeepAlive(fs)
	if len(rows) == 0 {
		return
	}
	flat, n := flattenRows(rows, "DASFFTExtensionBatch")
	if uint64(n)*2 > fs.MaxWidth {
		panic("domain too small for extending requested values")
	}
	hipMust(C.kzg_hip_das_fft_extension_batch(fs.hip(), frPtr(flat), C.uint64_t(n), C.uint64_t(len(rows))))
	for b := range rows {
		copy(rows[b], flat[n*b:n*(b+1)])
	}
}

// ToCompressedG1Batch / FromCompressedG1Batch: bls.ToCompressedG1 / bls.FromCompressedG1 (bls/bls_kilic.go:114-121) over a slice,
// decompression with the subgroup check on the device.
func (fs *FFTSettings) ToCompressedG1Batch(points []bls.G1Point) [][48]byte {
	defer runtime.KeepAlive(fs)
	out := make([][48]byte, len(points))
	if len(points) > 0 {
		hipMust(C.kzg_hip_g1_to_compressed(fs.hip(), g1Ptr(points), C.uint64_t(len(points)), unsafe.Pointer(&out[0])))
	}
	return out
}
func (fs *FFTSettings) FromCompressedG1Batch(data [][48]byte) ([]bls.G1Point, bool) {
	defer runtime.KeepAlive(fs)
	out := make([]bls.G1Point, len(data))
	if len(data) == 0 {
		return out, true
	}
	st := C.kzg_hip_g1_from_compressed(fs.hip(), unsafe.Pointer(&data[0]), C.uint64_t(len(data)), g1Ptr(out))
	if st == C.KZG_HIP_ERR_BAD_POINT {
		return nil, false
	}
	hipMust(st)
	return out, true
}

// MulG1Vec: out[i] = scalars[i] * points[i] (element-wise bls.MulG1, bls/bls_kilic.go:41-45).
func (fs *FFTSettings) MulG1Vec(points []bls.G1Point, scalars []bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(fs)
	if len(points) != len(scalars) {
		panic("MulG1Vec: length mismatch")
	}
	out := make([]bls.G1Point, len(points))
	if len(points) > 0 {
		hipMust(C.kzg_hip_g1_mul_vec(fs.hip(), g1Ptr(points), frPtr(scalars), C.uint64_t(len(points)), g1Ptr(out)))
	}
	return out
}

// LinCombBatch: out[b] = bls.LinCombG1(points[:len(factors[b])], factors[b]) for every row, on the cached set.
func (p *G1Points) LinCombBatch(factors [][]bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(p)
	if len(factors) == 0 {
		return nil
	}
	flat, n := flattenRows(factors, "LinCombBatch")
	out := make([]bls.G1Point, len(factors))
	hipMust(C.kzg_hip_lincomb_points_batch(p.h, frPtr(flat), C.uint64_t(n), C.uint64_t(len(factors)), g1Ptr(out)))
	return out
}

// Count is the number of points of the cached set.
func (p *G1Points) Count() uint64 {
	defer runtime.KeepAlive(p)
	return uint64(C.kzg_hip_points_count(p.h))
}

// TableInfo: signed window bits, windows and bytes of the fixed-base table CommitToPoly walks (zeros before the first commitment).
func (ks *KZGSettings) TableInfo() (windowBits, windows uint32, tableBytes uint64) {
	defer runtime.KeepAlive(ks)
	var c, w C.uint32_t
	var b C.uint64_t
	hipMust(C.kzg_hip_kzg_table_info(ks.hip(), &c, &w, &b))
	return uint32(c), uint32(w), uint64(b)
}

// SetProjectiveOutputs: CommitToPoly / ComputeProofSingle on this object return un-normalised Jacobian points, as the reference's own functions do (no F_p inversion
// per result: a lone CommitToPoly 0.30 -> ~0.19 ms).  Compare with bls.EqualG1; bls.ToCompressedG1 normalises as always.
func (ks *KZGSettings) SetProjectiveOutputs(on bool) {
	defer runtime.KeepAlive(ks)
	v := C.int(0)
	if on {
		v = 1
	}
	hipMust(C.kzg_hip_kzg_set_projective_outputs(ks.hip(), v))
}

// TableAdditions: mixed additions per coefficient of a commitment on that table (2 x windows: both GLV halves of a scalar walk the same rows).
func (ks *KZGSettings) TableAdditions() uint32 {
	defer runtime.KeepAlive(ks)
	return uint32(C.kzg_hip_kzg_table_additions(ks.hip()))
}

// FK20SingleBatch: FK20Single (fk20_single.go:122-137) on every row; out[b] holds the n proofs of polynomials[b].
func (fk *FK20SingleSettings) FK20SingleBatch(polynomials [][]bls.Fr) [][]bls.G1Point {
	defer runtime.KeepAlive(fk)
	if len(polynomials) == 0 {
		return nil
	}
	flat, n := flattenRows(polynomials, "FK20SingleBatch")
	proofs := make([]bls.G1Point, n*len(polynomials))
	hipMust(C.kzg_hip_fk20_single_batch(fk.hip(), frPtr(flat), C.uint64_t(n), C.uint64_t(len(polynomials)), g1Ptr(proofs)))
	out := make([][]bls.G1Point, len(polynomials))
	for b := range out {
		out[b] = proofs[n*b : n*(b+1)]
	}
	return out
}

// DAUsingFK20Batch: DAUsingFK20 (fk20_single.go:176-196) on every row; out[b] holds the 2n proofs of polynomials[b].
func (fk *FK20SingleSettings) DAUsingFK20Batch(polynomials [][]bls.Fr) [][]bls.G1Point {
	defer runtime.KeepAlive(fk)
	if len(polynomials) == 0 {
		return nil
	}
	flat, n := flattenRows(polynomials, "DAUsingFK20Batch")
	proofs := make([]bls.G1Point, 2*n*len(polynomials))
	hipMust(C.kzg_hip_da_using_fk20_batch(fk.hip(), frPtr(flat), C.uint64_t(n), C.uint64_t(len(polynomials)), g1Ptr(proofs)))
	out := make([][]bls.G1Point, len(polynomials))
	for b := range out {
		out[b] = proofs[2*n*b : 2*n*(b+1)]
	}
	return out
}

// XExtFFT returns the settings' xExtFFT as the device holds it (FFTG1 of the reversed setup, kzg.go:53-63), normalised.
func (fk *FK20SingleSettings) XExtFFT() []bls.G1Point {
	defer runtime.KeepAlive(fk)
	out := make([]bls.G1Point, len(fk.xExtFFT))
	hipMust(C.kzg_hip_fk20_single_x_ext_fft(fk.hip(), g1Ptr(out)))
	return out
}

// DAUsingFK20MultiBatch: DAUsingFK20Multi (fk20_multi.go:113-133) on every row.
func (fk *FK20MultiSettings) DAUsingFK20MultiBatch(polynomials [][]bls.Fr) [][]bls.G1Point {
	defer runtime.KeepAlive(fk)
	if len(polynomials) == 0 {
		return nil
	}
	flat, n := flattenRows(polynomials, "DAUsingFK20MultiBatch")
	per := 2 * uint64(n) / fk.chunkLen
	proofs := make([]bls.G1Point, per*uint64(len(polynomials)))
	hipMust(C.kzg_hip_da_using_fk20_multi_batch(fk.hip(), frPtr(flat), C.uint64_t(n), C.uint64_t(len(polynomials)), g1Ptr(proofs)))
	out := make([][]bls.G1Point, len(polynomials))
	for b := range out {
		out[b] = proofs[per*uint64(b) : per*uint64(b+1)]
	}
	return out
}

//go:build kzg_hip && !bignum_pure && !bignum_hol256 && !bignum_hbls
// +build kzg_hip,!bignum_pure,!bignum_hol256,!bignum_hbls

// File for package eth (eth/eth_hip.go).  eth/eth.go:145-182 (BlobToKZGCommitment, VerifyAggregateKZGProof, ComputeAggregateKZGProof) and
// eth/helpers.go:179-211 (ComputeKZGProof, EvaluatePolynomialInEvaluationForm) move to a file of their own tagged `!kzg_hip`; everything else
// in eth/ -- types, VerifyKZGProof, the precompile, the sidecar checks, the pairing -- stays as it is.
// Include and library directories come from CGO_CFLAGS / CGO_LDFLAGS (INTEGRATION.md 2), as for the kzg package.
package eth

/*
#cgo LDFLAGS: -lkzg_hip
#include "kzg_hip.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"unsafe"

	"github.com/protolambda/go-kzg/bls"
)

// HipDeviceID selects the GPU of this package's settings (one process per GPU: set from LOCAL_RANK before init runs, or call
// CloseHip and initHip again).
var HipDeviceID = 0

var (
	hipEth *C.kzg_hip_eth
	hipFFT *C.kzg_hip_fft
)

// initHip is called at the end of init() in eth/globals.go with parsedSetup.SetupLagrange in NATURAL order (as the JSON stores it, i.e.
// before the bit reversal at eth/globals.go:48): the library applies that permutation itself.
func initHip(lagrangeNaturalOrder []bls.G1Point) {
	if st := C.kzg_hip_fft_settings_new(C.int(HipDeviceID), 12, &hipFFT); st != C.KZG_HIP_OK {
		panic(fmt.Sprintf("kzg_hip: no gfx950 device (status %d); there is no CPU fallback in this build, drop -tags kzg_hip", int(st)))
	}
	if st := C.kzg_hip_eth_settings_new(hipFFT, unsafe.Pointer(&lagrangeNaturalOrder[0]), C.uint64_t(len(lagrangeNaturalOrder)), &hipEth); st != C.KZG_HIP_OK {
		panic(fmt.Sprintf("kzg_hip: eth settings: status %d", int(st)))
	}
}

// CloseHip releases the device side of the package (the Lagrange setup and its fixed-base table, up to 64 GB of HBM).
func CloseHip() {
	if hipEth != nil {
		C.kzg_hip_eth_settings_free(hipEth)
		hipEth = nil
	}
	if hipFFT != nil {
		C.kzg_hip_fft_settings_free(hipFFT)
		hipFFT = nil
	}
}

// flat view of a BlobSequence: Blob is [FieldElementsPerBlob][32]byte, so a []Blob is already contiguous and is passed as is
func blobBytes(blobs BlobSequence) (unsafe.Pointer, int) {
	if s, ok := blobs.([]Blob); ok && len(s) > 0 {
		return unsafe.Pointer(&s[0]), len(s)
	}
	n := blobs.Len()
	if n == 0 {
		return nil, 0
	}
	flat := make([]Blob, n)
	for i := range flat {
		flat[i] = blobs.At(i)
	}
	return unsafe.Pointer(&flat[0]), n
}

// BlobToKZGCommitment replaces eth/eth.go:145-151.
func BlobToKZGCommitment(blob Blob) (KZGCommitment, bool) {
	var out KZGCommitment
	var ok C.uint8_t
	if st := C.kzg_hip_eth_blob_to_kzg_commitment_batch(hipEth, unsafe.Pointer(&blob[0]), 1, unsafe.Pointer(&out[0]), &ok); st != C.KZG_HIP_OK {
		panic(fmt.Sprintf("kzg_hip: BlobToKZGCommitment: status %d", int(st)))
	}
	return out, ok != 0 // (KZGCommitment{}, false) for a field element >= r, like BlobToPolynomial
}

// BlobsToKZGCommitments is new API surface: all blobs of a block in one launch chain.
func BlobsToKZGCommitments(blobs []Blob) ([]KZGCommitment, []bool) {
	if len(blobs) == 0 {
		return nil, nil
	}
	out := make([]KZGCommitment, len(blobs))
	ok := make([]C.uint8_t, len(blobs))
	if st := C.kzg_hip_eth_blob_to_kzg_commitment_batch(hipEth, unsafe.Pointer(&blobs[0]), C.uint64_t(len(blobs)), unsafe.Pointer(&out[0]), &ok[0]); st != C.KZG_HIP_OK {
		panic(fmt.Sprintf("kzg_hip: BlobsToKZGCommitments: status %d", int(st)))
	}
	valid := make([]bool, len(blobs))
	for i := range valid {
		valid[i] = ok[i] != 0
	}
	return out, valid
}

// ComputeKZGProof replaces eth/helpers.go:179-203.
func ComputeKZGProof(polynomial []bls.Fr, z *bls.Fr) (KZGProof, error) {
	var proof KZGProof
	if len(polynomial) != FieldElementsPerBlob {
		return KZGProof{}, errors.New("polynomial has invalid length")
	}
	switch st := C.kzg_hip_eth_compute_kzg_proof(hipEth, unsafe.Pointer(&polynomial[0]), C.uint64_t(len(polynomial)), unsafe.Pointer(z), unsafe.Pointer(&proof[0]), nil); st {
	case C.KZG_HIP_OK:
		return proof, nil
	case C.KZG_HIP_ERR_BAD_ARG:
		return KZGProof{}, errors.New("invalid z challenge")
	default:
		panic(fmt.Sprintf("kzg_hip: ComputeKZGProof: status %d", int(st)))
	}
}

// ComputeKZGProofBatch is new API surface: ComputeKZGProof on every row in one launch chain; valid[b] == false is that row's
// "invalid z challenge" (its proof is zeroed).
func ComputeKZGProofBatch(polynomials [][]bls.Fr, zs []bls.Fr) ([]KZGProof, []bool, error) {
	if len(polynomials) != len(zs) {
		return nil, nil, errors.New("polynomials and challenges differ in number")
	}
	if len(polynomials) == 0 {
		return nil, nil, nil
	}
	flat := make([]bls.Fr, 0, FieldElementsPerBlob*len(polynomials))
	for _, p := range polynomials {
		if len(p) != FieldElementsPerBlob {
			return nil, nil, errors.New("polynomial has invalid length")
		}
		flat = append(flat, p...)
	}
	proofs := make([]KZGProof, len(polynomials))
	ok := make([]C.uint8_t, len(polynomials))
	if st := C.kzg_hip_eth_compute_kzg_proof_batch(hipEth, unsafe.Pointer(&flat[0]), C.uint64_t(FieldElementsPerBlob), C.uint64_t(len(polynomials)), unsafe.Pointer(&zs[0]),
		unsafe.Pointer(&proofs[0]), nil, &ok[0]); st != C.KZG_HIP_OK {
		panic(fmt.Sprintf("kzg_hip: ComputeKZGProofBatch: status %d", int(st)))
	}
	valid := make([]bool, len(polynomials))
	for i := range valid {
		valid[i] = ok[i] != 0
	}
	return proofs, valid, nil
}

// EvaluatePolynomialInEvaluationForm replaces eth/helpers.go:207-211 (DomainFr is the bit-reversed domain the library holds).
func EvaluatePolynomialInEvaluationForm(poly []bls.Fr, x *bls.Fr) *bls.Fr {
	var result bls.Fr
	if st := C.kzg_hip_eth_evaluate_polynomial_in_evaluation_form(hipEth, unsafe.Pointer(&poly[0]), C.uint64_t(len(poly)), unsafe.Pointer(x), unsafe.Pointer(&result)); st != C.KZG_HIP_OK {
		panic(fmt.Sprintf("kzg_hip: EvaluatePolynomialInEvaluationForm: status %d", int(st))) // a length mismatch panics in the reference too (bls/globals.go:107-109)
	}
	return &result
}

// ComputeAggregateKZGProof replaces eth/eth.go:175-182: the whole block in one call -- polynomials, commitments, Fiat-Shamir
// transcript, aggregated polynomial and its proof.
func ComputeAggregateKZGProof(blobs BlobSequence) (KZGProof, error) {
	var proof KZGProof
	p, n := blobBytes(blobs)
	switch st := C.kzg_hip_eth_compute_aggregate_kzg_proof(hipEth, p, C.uint64_t(n), unsafe.Pointer(&proof[0]), nil); st {
	case C.KZG_HIP_OK:
		return proof, nil
	case C.KZG_HIP_ERR_BAD_BLOB:
		return KZGProof{}, errors.New("could not convert blobs to polynomials")
	case C.KZG_HIP_ERR_BAD_ARG:
		return KZGProof{}, errors.New("invalid z challenge")
	default:
		panic(fmt.Sprintf("kzg_hip: ComputeAggregateKZGProof: status %d", int(st)))
	}
}

// VerifyAggregateKZGProof replaces eth/eth.go:155-172: aggregation and evaluation on the device, the pairing
// (VerifyKZGProofFromPoints, eth/helpers.go:55-68) here as before.
func VerifyAggregateKZGProof(blobs BlobSequence, expectedKZGCommitments KZGCommitmentSequence, kzgAggregatedProof KZGProof) (bool, error) {
	p, n := blobBytes(blobs)
	if expectedKZGCommitments.Len() != n {
		panic("got LinCombG1 numbers/factors length mismatch") // what the reference does here: bls.LinCombG1's panic (eth/helpers.go:159)
	}
	comms := make([]KZGCommitment, n)
	for i := range comms {
		comms[i] = expectedKZGCommitments.At(i)
	}
	var cp unsafe.Pointer
	if n > 0 {
		cp = unsafe.Pointer(&comms[0])
	}
	var agg bls.G1Point
	var z, y bls.Fr
	switch st := C.kzg_hip_eth_compute_aggregated_poly_and_commitment(hipEth, p, cp, C.uint64_t(n), nil, unsafe.Pointer(&agg), unsafe.Pointer(&z), unsafe.Pointer(&y)); st {
	case C.KZG_HIP_OK:
	case C.KZG_HIP_ERR_BAD_BLOB:
		return false, errors.New("could not convert blobs to polynomials")
	case C.KZG_HIP_ERR_BAD_POINT:
		return false, errors.New("invalid compressed G1") // bls.FromCompressedG1's error, eth/helpers.go:153-156
	default:
		panic(fmt.Sprintf("kzg_hip: VerifyAggregateKZGProof: status %d", int(st)))
	}
	kzgProofG1, err := bls.FromCompressedG1(kzgAggregatedProof[:])
	if err != nil {
		return false, fmt.Errorf("failed to decode kzgProof: %v", err)
	}
	return VerifyKZGProofFromPoints(&agg, &z, &y, kzgProofG1), nil
}

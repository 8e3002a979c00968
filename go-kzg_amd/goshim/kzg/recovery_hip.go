//go:build kzg_hip && !bignum_pure && !bignum_hol256 && !bignum_hbls
// +build kzg_hip,!bignum_pure,!bignum_hol256,!bignum_hbls

package kzg

/*
#include "kzg_hip.h"
*/
import "C"

import (
	"runtime"
	"errors"
	"unsafe"

	"github.com/protolambda/go-kzg/bls"
)

// ZeroPolyViaMultiplication replaces zero_poly.go:116-217: (evaluations, coefficients) of the vanishing polynomial.
func (fs *FFTSettings) ZeroPolyViaMultiplication(missingIndices []uint64, length uint64) ([]bls.Fr, []bls.Fr) {
	defer runtime.KeepAlive(fs) // the finalizer must not free the device handle under a running call
	zeroEval := make([]bls.Fr, length)
	zeroPoly := make([]bls.Fr, length)
	var idx *C.uint64_t
	if len(missingIndices) > 0 {
		idx = (*C.uint64_t)(unsafe.Pointer(&missingIndices[0]))
	}
	hipMust(C.kzg_hip_zero_poly_via_multiplication(fs.hip(), idx, C.uint64_t(len(missingIndices)), C.uint64_t(length),
		frPtr(zeroEval), frPtr(zeroPoly)))
	return zeroEval, zeroPoly
}

// RecoverPolyFromSamples replaces recover_from_samples.go:42-109.  The device path always uses ZeroPolyViaMultiplication as
// the zero-polynomial function (the only one the reference ships); samples[i] == nil marks a missing value.
func (fs *FFTSettings) RecoverPolyFromSamples(samples []*bls.Fr, zeroPolyFn ZeroPolyFn) ([]bls.Fr, error) {
	defer runtime.KeepAlive(fs) // the finalizer must not free the device handle under a running call
	n := len(samples)
	flat := make([]bls.Fr, n)
	present := make([]byte, n)
	for i, s := range samples {
		if s != nil {
			bls.CopyFr(&flat[i], s)
			present[i] = 1
		}
	}
	out := make([]bls.Fr, n)
	st := C.kzg_hip_recover_poly_from_samples(fs.hip(), frPtr(flat), (*C.uint8_t)(unsafe.Pointer(&present[0])), C.uint64_t(n), frPtr(out))
	if st == C.KZG_HIP_ERR_RECOVERY {
		return nil, errors.New("failed to reconstruct data correctly") // recover_from_samples.go:103-107
	}
	hipMust(st)
	return out, nil
}

// ComputeProofMulti replaces kzg_multi_proofs.go:13-44 (the reference's divisor, X^n, is kept as it is).
func (ks *KZGSettings) ComputeProofMulti(poly []bls.Fr, x uint64, n uint64) *bls.G1Point {
	defer runtime.KeepAlive(ks) // the finalizer must not free the device handle under a running call
	out := new(bls.G1Point)
	hipMust(C.kzg_hip_compute_proof_multi(ks.hip(), frPtr(poly), C.uint64_t(len(poly)), C.uint64_t(x), C.uint64_t(n), unsafePointerG1(out)))
	return out
}

// checkProofMultiProverHalf is the device half of CheckProofMulti (kzg_multi_proofs.go:47-75): [I(s)]_1 and x^n.  The two
// pairings stay on the CPU backend (bls.PairingsVerify).
func (ks *KZGSettings) checkProofMultiProverHalf(x *bls.Fr, ys []bls.Fr) (is1 bls.G1Point, xPow bls.Fr) {
	defer runtime.KeepAlive(ks) // the finalizer must not free the device handle under a running call
	hipMust(C.kzg_hip_check_proof_multi_interpolation(ks.hip(), frPtr(ys), C.uint64_t(len(ys)), unsafe.Pointer(x),
		unsafePointerG1(&is1), unsafe.Pointer(&xPow)))
	return
}

// FrFrom32Slice / FrTo32Slice: bls.FrFrom32 / bls.FrTo32 (bls/bignum_kilic.go:33-55) over a whole slice on the device.
func (fs *FFTSettings) FrFrom32Slice(in [][32]byte) (out []bls.Fr, ok bool) {
	defer runtime.KeepAlive(fs) // the finalizer must not free the device handle under a running call
	out = make([]bls.Fr, len(in))
	if len(in) == 0 {
		return out, true
	}
	var allOK C.int
	hipMust(C.kzg_hip_fr_from_le32(fs.hip(), unsafe.Pointer(&in[0]), C.uint64_t(len(in)), frPtr(out), &allOK))
	return out, allOK != 0
}

func (fs *FFTSettings) FrTo32Slice(in []bls.Fr) [][32]byte {
	defer runtime.KeepAlive(fs) // the finalizer must not free the device handle under a running call
	out := make([][32]byte, len(in))
	if len(in) > 0 {
		hipMust(C.kzg_hip_fr_to_le32(fs.hip(), frPtr(in), C.uint64_t(len(in)), unsafe.Pointer(&out[0])))
	}
	return out
}

//go:build kzg_hip && !bignum_pure && !bignum_hol256 && !bignum_hbls
// +build kzg_hip,!bignum_pure,!bignum_hol256,!bignum_hbls

package kzg

/*
#include "kzg_hip.h"
*/
import "C"

import (
	"unsafe"

	"github.com/protolambda/go-kzg/bls"
)

func unsafePointerG1(p *bls.G1Point) unsafe.Pointer { return unsafe.Pointer(p) }

// CommitToPoly replaces kzg_single_proofs.go:17-19.
func (ks *KZGSettings) CommitToPoly(coeffs []bls.Fr) *bls.G1Point {
	out := new(bls.G1Point)
	hipMust(C.kzg_hip_commit_to_poly(ks.hip(), frPtr(coeffs), C.uint64_t(len(coeffs)), unsafePointerG1(out)))
	return out
}

// CommitToPolyBatch is new API surface: many blobs per launch is what fills 256 CUs.
func (ks *KZGSettings) CommitToPolyBatch(coeffs [][]bls.Fr) []bls.G1Point {
	if len(coeffs) == 0 {
		return nil
	}
	n := len(coeffs[0])
	flat := make([]bls.Fr, 0, n*len(coeffs))
	for _, c := range coeffs {
		if len(c) != n {
			panic("CommitToPolyBatch: ragged batch")
		}
		flat = append(flat, c...)
	}
	out := make([]bls.G1Point, len(coeffs))
	hipMust(C.kzg_hip_commit_to_poly_batch(ks.hip(), frPtr(flat), C.uint64_t(n), C.uint64_t(len(coeffs)), g1Ptr(out)))
	return out
}

// ComputeProofSingle replaces kzg_single_proofs.go:36-54 (x is a uint64 there too).
func (ks *KZGSettings) ComputeProofSingle(poly []bls.Fr, x uint64) *bls.G1Point {
	out := new(bls.G1Point)
	hipMust(C.kzg_hip_compute_proof_single(ks.hip(), frPtr(poly), C.uint64_t(len(poly)), C.uint64_t(x), unsafePointerG1(out)))
	return out
}

// ToeplitzPart2 replaces fk20_single.go:59-77.
func (ks *KZGSettings) ToeplitzPart2(toeplitzCoeffs []bls.Fr, xExtFFT []bls.G1Point) (hExtFFT []bls.G1Point) {
	if uint64(len(toeplitzCoeffs)) != uint64(len(xExtFFT)) {
		panic("expected toeplitz coeffs to match xExtFFT length")
	}
	hExtFFT = make([]bls.G1Point, len(xExtFFT))
	hipMust(C.kzg_hip_toeplitz_part2(ks.hip(), frPtr(toeplitzCoeffs), g1Ptr(xExtFFT), C.uint64_t(len(xExtFFT)), g1Ptr(hExtFFT)))
	return hExtFFT
}

// ToeplitzPart3 replaces fk20_single.go:80-87.
func (ks *KZGSettings) ToeplitzPart3(hExtFFT []bls.G1Point) []bls.G1Point {
	out := make([]bls.G1Point, len(hExtFFT)/2)
	hipMust(C.kzg_hip_toeplitz_part3(ks.hip(), g1Ptr(hExtFFT), C.uint64_t(len(hExtFFT)), g1Ptr(out)))
	return out
}

func (fk *FK20SingleSettings) hip() *C.kzg_hip_fk20s {
	kh := fk.KZGSettings.hip()
	hipMu.Lock()
	defer hipMu.Unlock()
	if h, ok := hipFK20S[fk]; ok {
		return h
	}
	var h *C.kzg_hip_fk20s
	// the Go constructor (kzg.go:43-64) has already validated n2; xExtFFT is rebuilt on the device from SecretG1
	hipMust(C.kzg_hip_fk20_single_settings_new(kh, C.uint64_t(len(fk.xExtFFT)), &h))
	hipFK20S[fk] = h
	return h
}

// FK20Single replaces fk20_single.go:122-134.
func (fk *FK20SingleSettings) FK20Single(polynomial []bls.Fr) []bls.G1Point {
	out := make([]bls.G1Point, len(polynomial))
	hipMust(C.kzg_hip_fk20_single(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

// FK20SingleDAOptimized replaces fk20_single.go:139-172.
func (fk *FK20SingleSettings) FK20SingleDAOptimized(polynomial []bls.Fr) []bls.G1Point {
	out := make([]bls.G1Point, len(polynomial))
	hipMust(C.kzg_hip_fk20_single_da_optimized(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

// DAUsingFK20 replaces fk20_single.go:176-196.
func (fk *FK20SingleSettings) DAUsingFK20(polynomial []bls.Fr) []bls.G1Point {
	out := make([]bls.G1Point, 2*len(polynomial))
	hipMust(C.kzg_hip_da_using_fk20(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

func (fk *FK20MultiSettings) hip() *C.kzg_hip_fk20m {
	kh := fk.KZGSettings.hip()
	hipMu.Lock()
	defer hipMu.Unlock()
	if h, ok := hipFK20M[fk]; ok {
		return h
	}
	var h *C.kzg_hip_fk20m
	n2 := uint64(len(fk.xExtFFTFiles[0])) * fk.chunkLen // files hold 2k = n2 / chunkLen points each (kzg.go:99-114)
	hipMust(C.kzg_hip_fk20_multi_settings_new(kh, C.uint64_t(n2), C.uint64_t(fk.chunkLen), &h))
	hipFK20M[fk] = h
	return h
}

// FK20Multi replaces fk20_multi.go:25-52.
func (fk *FK20MultiSettings) FK20Multi(polynomial []bls.Fr) []bls.G1Point {
	out := make([]bls.G1Point, uint64(len(polynomial))/fk.chunkLen)
	hipMust(C.kzg_hip_fk20_multi(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

// FK20MultiDAOptimized replaces fk20_multi.go:58-109.
func (fk *FK20MultiSettings) FK20MultiDAOptimized(polynomial []bls.Fr) []bls.G1Point {
	out := make([]bls.G1Point, uint64(len(polynomial))/fk.chunkLen)
	hipMust(C.kzg_hip_fk20_multi_da_optimized(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

// DAUsingFK20Multi replaces fk20_multi.go:113-133.
func (fk *FK20MultiSettings) DAUsingFK20Multi(polynomial []bls.Fr) []bls.G1Point {
	out := make([]bls.G1Point, 2*uint64(len(polynomial))/fk.chunkLen)
	hipMust(C.kzg_hip_da_using_fk20_multi(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

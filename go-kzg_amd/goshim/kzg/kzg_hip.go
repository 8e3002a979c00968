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

func unsafePointerG1(p *bls.G1Point) unsafe.Pointer { return unsafe.Pointer(p) }

// CommitToPoly replaces kzg_single_proofs.go:17-19.
func (ks *KZGSettings) CommitToPoly(coeffs []bls.Fr) *bls.G1Point {
	defer runtime.KeepAlive(ks) // the finalizer must not free the device handle under a running call
	out := new(bls.G1Point)
	hipMust(C.kzg_hip_commit_to_poly(ks.hip(), frPtr(coeffs), C.uint64_t(len(coeffs)), unsafePointerG1(out)))
	return out
}

// CommitToPolyUnoptimized replaces kzg_single_proofs.go:22-33 (a MulG1 / AddG1 loop over SecretG1[:len(coeffs)]): the same group element, so
// it is the same call here -- a caller that used it as a cross-check of LinCombG1 now cross-checks nothing and should compare against the CPU backend.
func (ks *KZGSettings) CommitToPolyUnoptimized(coeffs []bls.Fr) *bls.G1Point {
	return ks.CommitToPoly(coeffs)
}

// CommitToPolyBatch is new API surface: many blobs per launch is what fills 256 CUs.
func (ks *KZGSettings) CommitToPolyBatch(coeffs [][]bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(ks) // the finalizer must not free the device handle under a running call
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
	defer runtime.KeepAlive(ks) // the finalizer must not free the device handle under a running call
	out := new(bls.G1Point)
	hipMust(C.kzg_hip_compute_proof_single(ks.hip(), frPtr(poly), C.uint64_t(len(poly)), C.uint64_t(x), unsafePointerG1(out)))
	return out
}

// ComputeProofSingleBatch is new API surface (kzg_hip_compute_proof_single_batch): polys[b] evaluated at xs[b].
func (ks *KZGSettings) ComputeProofSingleBatch(polys [][]bls.Fr, xs []uint64) []bls.G1Point {
	defer runtime.KeepAlive(ks) // the finalizer must not free the device handle under a running call
	if len(polys) == 0 {
		return nil
	}
	if len(polys) != len(xs) {
		panic("ComputeProofSingleBatch: len(polys) != len(xs)")
	}
	n := len(polys[0])
	flat := make([]bls.Fr, 0, n*len(polys))
	for _, c := range polys {
		if len(c) != n {
			panic("ComputeProofSingleBatch: ragged batch")
		}
		flat = append(flat, c...)
	}
	out := make([]bls.G1Point, len(polys))
	hipMust(C.kzg_hip_compute_proof_single_batch(ks.hip(), frPtr(flat), C.uint64_t(n), C.uint64_t(len(polys)), (*C.uint64_t)(unsafe.Pointer(&xs[0])), g1Ptr(out)))
	return out
}

// SetTableBudgetGB gives this settings object a smaller (or bigger) fixed-base commitment table than the 110 GB default budget (4096 points:
// signed 16-bit windows, 8 of them walked by both GLV halves of a scalar, 103 GB, 16 additions per coefficient); 60 selects 15-bit windows
// (58 GB, 18 additions), 17 selects 13-bit windows (16 GB, 20 additions).  Call before the first commitment.
func (ks *KZGSettings) SetTableBudgetGB(gb float64) {
	defer runtime.KeepAlive(ks) // the finalizer must not free the device handle under a running call
	hipMust(C.kzg_hip_kzg_set_table_budget_gb(ks.hip(), C.double(gb)))
}

// G1Points is a point set kept in HBM for repeated bls.LinCombG1 calls on the SAME points (CommitToEvalPoly's secretG1IFFT,
// kzg_single_proofs.go:12-14; eth's Lagrange setup, eth/helpers.go:99,159,199): uploaded and converted once.
type G1Points struct{ h *C.kzg_hip_points }

func (fs *FFTSettings) NewG1Points(points []bls.G1Point) *G1Points {
	defer runtime.KeepAlive(fs) // the finalizer must not free the device handle under a running call
	p := &G1Points{}
	hipMust(C.kzg_hip_points_new(fs.hip(), g1Ptr(points), C.uint64_t(len(points)), &p.h))
	runtime.SetFinalizer(p, (*G1Points).Close)
	return p
}
func (p *G1Points) Close() {
	if p.h != nil {
		C.kzg_hip_points_free(p.h)
		p.h = nil
	}
}

// SetTableBudgetGB: HBM budget of this set's fixed-base table (default min(32 GB, free - 24 GB); 0 keeps the set on the bucket pipeline)
func (p *G1Points) SetTableBudgetGB(gb float64) {
	defer runtime.KeepAlive(p)
	hipMust(C.kzg_hip_points_set_table_budget_gb(p.h, C.double(gb)))
}

// LinComb == bls.LinCombG1(points[:len(factors)], factors)
func (p *G1Points) LinComb(factors []bls.Fr) *bls.G1Point {
	defer runtime.KeepAlive(p) // the finalizer must not free the device handle under a running call
	out := new(bls.G1Point)
	hipMust(C.kzg_hip_lincomb_points(p.h, frPtr(factors), C.uint64_t(len(factors)), unsafePointerG1(out)))
	return out
}

// LoadTrustedSetupJSON decodes the G1 arrays of a trusted-setup document (JSONTrustedSetup, eth/globals.go:33-49: the init() there
// would call this instead of json.Unmarshal for SetupG1 / SetupLagrange; SetupG2 stays with encoding/json + Kilic).  Hex decoding
// happens in the library, decompression and the subgroup check on the device.  Panics like init() does on a malformed document.
func (fs *FFTSettings) LoadTrustedSetupJSON(text []byte) (setupG1, setupLagrange []bls.G1Point) {
	defer runtime.KeepAlive(fs) // the finalizer must not free the device handle under a running call
	if len(text) == 0 {
		panic("kzg_hip: empty trusted setup")
	}
	var n1, n2 C.uint64_t
	cs := (*C.char)(unsafe.Pointer(&text[0]))
	hipMust(C.kzg_hip_trusted_setup_from_json(fs.hip(), cs, C.uint64_t(len(text)), nil, nil, 0, &n1, &n2))
	cap_ := n1
	if n2 > cap_ {
		cap_ = n2
	}
	setupG1 = make([]bls.G1Point, cap_)
	setupLagrange = make([]bls.G1Point, cap_)
	hipMust(C.kzg_hip_trusted_setup_from_json(fs.hip(), cs, C.uint64_t(len(text)), g1Ptr(setupG1), g1Ptr(setupLagrange), cap_, &n1, &n2))
	return setupG1[:n1], setupLagrange[:n2]
}

// ToeplitzPart2 replaces fk20_single.go:59-77.
func (ks *KZGSettings) ToeplitzPart2(toeplitzCoeffs []bls.Fr, xExtFFT []bls.G1Point) (hExtFFT []bls.G1Point) {
	defer runtime.KeepAlive(ks) // the finalizer must not free the device handle under a running call
	if uint64(len(toeplitzCoeffs)) != uint64(len(xExtFFT)) {
		panic("expected toeplitz coeffs to match xExtFFT length")
	}
	hExtFFT = make([]bls.G1Point, len(xExtFFT))
	hipMust(C.kzg_hip_toeplitz_part2(ks.hip(), frPtr(toeplitzCoeffs), g1Ptr(xExtFFT), C.uint64_t(len(xExtFFT)), g1Ptr(hExtFFT)))
	return hExtFFT
}

// ToeplitzPart3 replaces fk20_single.go:80-87.
func (ks *KZGSettings) ToeplitzPart3(hExtFFT []bls.G1Point) []bls.G1Point {
	defer runtime.KeepAlive(ks) // the finalizer must not free the device handle under a running call
	out := make([]bls.G1Point, len(hExtFFT)/2)
	hipMust(C.kzg_hip_toeplitz_part3(ks.hip(), g1Ptr(hExtFFT), C.uint64_t(len(hExtFFT)), g1Ptr(out)))
	return out
}

func (fk *FK20SingleSettings) hip() *C.kzg_hip_fk20s {
	kh := fk.KZGSettings.hip()
	hipMu.Lock()
	defer hipMu.Unlock()
	if h, ok := hipFK20S[uintptr(unsafe.Pointer(fk))]; ok {
		return h
	}
	var h *C.kzg_hip_fk20s
	// the Go constructor (kzg.go:43-64) has already validated n2; xExtFFT is rebuilt on the device from SecretG1
	hipMust(C.kzg_hip_fk20_single_settings_new(kh, C.uint64_t(len(fk.xExtFFT)), &h))
	hipFK20S[uintptr(unsafe.Pointer(fk))] = h
	runtime.SetFinalizer(fk, (*FK20SingleSettings).CloseHip)
	return h
}

// FK20Single replaces fk20_single.go:122-134.
func (fk *FK20SingleSettings) FK20Single(polynomial []bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(fk) // the finalizer must not free the device handle under a running call
	out := make([]bls.G1Point, len(polynomial))
	hipMust(C.kzg_hip_fk20_single(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

// FK20SingleDAOptimized replaces fk20_single.go:139-172.
func (fk *FK20SingleSettings) FK20SingleDAOptimized(polynomial []bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(fk) // the finalizer must not free the device handle under a running call
	out := make([]bls.G1Point, len(polynomial))
	hipMust(C.kzg_hip_fk20_single_da_optimized(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

// DAUsingFK20 replaces fk20_single.go:176-196.
func (fk *FK20SingleSettings) DAUsingFK20(polynomial []bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(fk) // the finalizer must not free the device handle under a running call
	out := make([]bls.G1Point, 2*len(polynomial))
	hipMust(C.kzg_hip_da_using_fk20(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

func (fk *FK20MultiSettings) hip() *C.kzg_hip_fk20m {
	kh := fk.KZGSettings.hip()
	hipMu.Lock()
	defer hipMu.Unlock()
	if h, ok := hipFK20M[uintptr(unsafe.Pointer(fk))]; ok {
		return h
	}
	var h *C.kzg_hip_fk20m
	n2 := uint64(len(fk.xExtFFTFiles[0])) * fk.chunkLen // files hold 2k = n2 / chunkLen points each (kzg.go:99-114)
	hipMust(C.kzg_hip_fk20_multi_settings_new(kh, C.uint64_t(n2), C.uint64_t(fk.chunkLen), &h))
	hipFK20M[uintptr(unsafe.Pointer(fk))] = h
	runtime.SetFinalizer(fk, (*FK20MultiSettings).CloseHip)
	return h
}

// FK20Multi replaces fk20_multi.go:25-52.
func (fk *FK20MultiSettings) FK20Multi(polynomial []bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(fk) // the finalizer must not free the device handle under a running call
	out := make([]bls.G1Point, uint64(len(polynomial))/fk.chunkLen)
	hipMust(C.kzg_hip_fk20_multi(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

// FK20MultiDAOptimized replaces fk20_multi.go:58-109.
func (fk *FK20MultiSettings) FK20MultiDAOptimized(polynomial []bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(fk) // the finalizer must not free the device handle under a running call
	out := make([]bls.G1Point, uint64(len(polynomial))/fk.chunkLen)
	hipMust(C.kzg_hip_fk20_multi_da_optimized(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

// DAUsingFK20Multi replaces fk20_multi.go:113-133.
func (fk *FK20MultiSettings) DAUsingFK20Multi(polynomial []bls.Fr) []bls.G1Point {
	defer runtime.KeepAlive(fk) // the finalizer must not free the device handle under a running call
	out := make([]bls.G1Point, 2*uint64(len(polynomial))/fk.chunkLen)
	hipMust(C.kzg_hip_da_using_fk20_multi(fk.hip(), frPtr(polynomial), C.uint64_t(len(polynomial)), g1Ptr(out)))
	return out
}

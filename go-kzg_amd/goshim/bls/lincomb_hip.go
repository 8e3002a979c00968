//go:build kzg_hip && !bignum_pure && !bignum_hol256 && !bignum_hbls
// +build kzg_hip,!bignum_pure,!bignum_hol256,!bignum_hbls

// File for package bls (bls/lincomb_hip.go).  bls/bls_kilic.go:132-150 (LinCombG1) and bls/globals.go:106-153
// (EvaluatePolyInEvaluationForm) move to files of their own tagged `!kzg_hip`; the latter is kept reachable as
// evaluatePolyInEvaluationFormKilic for domains that are not a standard power-of-two root-of-unity table.
// Serves eth/helpers.go:99,159,199 and kzg_multi_proofs.go:42,50,75 (LinCombG1), fft_fr_test.go:73-99 (evaluation).
// Include and library directories come from CGO_CFLAGS / CGO_LDFLAGS (INTEGRATION.md 2), as for the kzg package.
package bls

/*
#cgo LDFLAGS: -lkzg_hip
#include "kzg_hip.h"
*/
import "C"

import (
	"fmt"
	"sync"
	"unsafe"
)

// HipDeviceID selects the GPU of this package's device context (one process per GPU: set from LOCAL_RANK before the first call).
var HipDeviceID = 0

var (
	hipMu      sync.Mutex
	hipDomains = map[uint8]*C.kzg_hip_fft{} // max_scale -> settings with that domain (scale 0: a bare device context for LinCombG1)
	hipSets    sync.Map                     // &numbers[0] -> registered point set
)

type hipSet struct {
	h *C.kzg_hip_points
	n int
}

// hipDomain returns the process-wide settings object of width 2^scale on HipDeviceID (created once, never freed: the tables of a
// scale-12 domain are 400 KiB of HBM).
func hipDomain(scale uint8) *C.kzg_hip_fft {
	hipMu.Lock()
	defer hipMu.Unlock()
	if h, ok := hipDomains[scale]; ok {
		return h
	}
	var h *C.kzg_hip_fft
	if st := C.kzg_hip_fft_settings_new(C.int(HipDeviceID), C.uint(scale), &h); st != C.KZG_HIP_OK {
		panic(fmt.Sprintf("kzg_hip: no gfx950 device (status %d); there is no CPU fallback in this build, drop -tags kzg_hip", int(st)))
	}
	hipDomains[scale] = h
	return h
}

// RegisterG1Points keeps `numbers` resident in HBM (with its own fixed-base table): every later LinCombG1(numbers[:k], ...) on the SAME
// backing array is a table walk (0.4 ms for 4096 points, concurrent calls coalesced) instead of a bucket MSM on freshly uploaded points
// (1.3 ms).  The caller promises not to modify the points afterwards.  eth/globals.go:init registers kzgSetupLagrange; kzg.NewKZGSettings
// may register SecretG1 for kzg_multi_proofs.go:42.
func RegisterG1Points(numbers []G1Point) {
	if len(numbers) == 0 {
		return
	}
	var h *C.kzg_hip_points
	if st := C.kzg_hip_points_new(hipDomain(0), unsafe.Pointer(&numbers[0]), C.uint64_t(len(numbers)), &h); st != C.KZG_HIP_OK {
		panic(fmt.Sprintf("kzg_hip: RegisterG1Points: status %d", int(st)))
	}
	hipSets.Store(unsafe.Pointer(&numbers[0]), hipSet{h, len(numbers)})
}

// UnregisterG1Points frees the device copy (and table) of a registered set.
func UnregisterG1Points(numbers []G1Point) {
	if len(numbers) == 0 {
		return
	}
	if v, ok := hipSets.LoadAndDelete(unsafe.Pointer(&numbers[0])); ok {
		C.kzg_hip_points_free(v.(hipSet).h)
	}
}

// LinCombG1 replaces bls/bls_kilic.go:132-150.
// Unregistered points: the library itself notices a slice whose CONTENT keeps coming back (byte-for-byte comparison on every call, round 6: capi_core.hip, lincomb_promo)
// and serves it from a cached set from the fourth call on (1.3 -> 0.27 ms for 4096 points); a caller that modifies its points in place is seen and takes the one-shot
// path.  RegisterG1Points remains the explicit form: no comparison per call, no warm-up, but the promise not to modify the points is the caller's.
func LinCombG1(numbers []G1Point, factors []Fr) *G1Point {
	if len(numbers) != len(factors) {
		panic("got LinCombG1 numbers/factors length mismatch")
	}
	var out G1Point
	if len(numbers) == 0 {
		C.kzg_hip_lincomb_g1(hipDomain(0), nil, nil, 0, unsafe.Pointer(&out))
		return &out
	}
	if v, ok := hipSets.Load(unsafe.Pointer(&numbers[0])); ok && v.(hipSet).n >= len(numbers) { // a prefix of a registered set
		if st := C.kzg_hip_lincomb_points(v.(hipSet).h, unsafe.Pointer(&factors[0]), C.uint64_t(len(factors)), unsafe.Pointer(&out)); st != C.KZG_HIP_OK {
			panic(fmt.Sprintf("kzg_hip: LinCombG1: status %d", int(st)))
		}
		return &out
	}
	if st := C.kzg_hip_lincomb_g1(hipDomain(0), unsafe.Pointer(&numbers[0]), unsafe.Pointer(&factors[0]), C.uint64_t(len(numbers)), unsafe.Pointer(&out)); st != C.KZG_HIP_OK {
		panic(fmt.Sprintf("kzg_hip: LinCombG1: status %d", int(st)))
	}
	return &out
}

// EvaluatePolyInEvaluationForm replaces bls/globals.go:106-153 where rootsOfUnity is the standard table of a power-of-two width
// (ExpandedRootsOfUnity[:MaxWidth], what fft_fr_test.go:73-99 and every FFTSettings caller pass): barycentric evaluation on the device.
// Any other domain (eth's bit-reversed DomainFr goes through eth.EvaluatePolynomialInEvaluationForm's own binding) takes the Kilic code.
func EvaluatePolyInEvaluationForm(yFr *Fr, poly []Fr, x *Fr, rootsOfUnity []Fr, scale uint8) {
	if len(poly) != len(rootsOfUnity)>>scale {
		panic(fmt.Errorf("expected roots of unity (len %d >> %d == %d) to match polynomial size (len %d)", len(rootsOfUnity), scale, len(rootsOfUnity)>>scale, len(poly)))
	}
	width := len(rootsOfUnity)
	maxScale := uint8(0)
	for (1 << maxScale) < width {
		maxScale++
	}
	if width == 0 || (1<<maxScale) != width || maxScale >= uint8(len(Scale2RootOfUnity)) || len(poly) == 0 ||
		(width > 1 && !EqualFr(&rootsOfUnity[1], &Scale2RootOfUnity[maxScale])) {
		evaluatePolyInEvaluationFormKilic(yFr, poly, x, rootsOfUnity, scale)
		return
	}
	st := C.kzg_hip_evaluate_poly_in_evaluation_form(hipDomain(maxScale), unsafe.Pointer(&poly[0]), C.uint64_t(len(poly)), unsafe.Pointer(x), C.uint32_t(scale), unsafe.Pointer(yFr))
	if st != C.KZG_HIP_OK {
		panic(fmt.Sprintf("kzg_hip: EvaluatePolyInEvaluationForm: status %d", int(st)))
	}
}

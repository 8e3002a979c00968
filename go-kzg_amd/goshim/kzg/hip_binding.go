//go:build kzg_hip && !bignum_pure && !bignum_hol256 && !bignum_hbls
// +build kzg_hip,!bignum_pure,!bignum_hol256,!bignum_hbls

// Package kzg: cgo binding of libkzg_hip.so (include/kzg_hip.h) behind go-kzg's existing API.
//
// NOT COMPILED IN THE BUILD IMAGE (it has no Go toolchain); kept paper-thin on purpose: every behaviour is
// implemented and tested at the C ABI (tests/test_gpu_parity.py).  Drop these files into the go-kzg package
// directory and build with `-tags kzg_hip` and CGO_CFLAGS=-I<repo>/include CGO_LDFLAGS="-L<repo>/go-kzg_amd
// -Wl,-rpath,<repo>/go-kzg_amd" (no ${SRCDIR}-relative paths here: the file is meant to be copied); the four method files they replace carry `//go:build !kzg_hip`
// (see INTEGRATION.md).  The default Kilic backend stays in place for everything else (G2, pairings, per-op
// bls.* calls, tests), so bls.Fr / bls.G1Point keep their memory images: Fr = [4]uint64 Montgomery,
// G1Point = [3][6]uint64 Jacobian Montgomery (bls/bignum_kilic.go:21-23, bls/bls_kilic.go:30-35), which is
// exactly what the C ABI takes -- Go slices are passed zero-copy, like bls/bls_hbls.go:143-149 does for Herumi.
package kzg

/*
#cgo LDFLAGS: -lkzg_hip
#include "kzg_hip.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"sync"
	"unsafe"

	"github.com/protolambda/go-kzg/bls"
)

// One device handle per Go settings object, created on first use (settings are immutable after construction).
// The side tables are keyed by the object's ADDRESS as a uintptr, not by the pointer: they do not keep the settings object
// alive.  A finalizer on the settings object (and the explicit Close methods below) frees the device handle -- up to 103 GB of
// HBM for a KZGSettings -- and removes the entry.  KZGSettings points to its FFTSettings and the FK20 settings to their
// KZGSettings, so the runtime runs the finalizers outermost first (runtime.SetFinalizer: "if A points to B, A's runs first"),
// which is the order the C side needs (a kzg handle refers to its fft handle).
var (
	hipMu       sync.Mutex
	hipFFT      = map[uintptr]*C.kzg_hip_fft{}
	hipKZG      = map[uintptr]*C.kzg_hip_kzg{}
	hipFK20S    = map[uintptr]*C.kzg_hip_fk20s{}
	hipFK20M    = map[uintptr]*C.kzg_hip_fk20m{}
	HipDeviceID = 0 // one process per GPU: set from LOCAL_RANK before the first call
)

// CloseHip releases the device side of the settings object now (idempotent; the finalizer does the same at collection).
// Order: close dependents first (FK20 settings, then KZGSettings, then FFTSettings) -- a kzg handle refers to its fft handle.
func (fs *FFTSettings) CloseHip() {
	runtime.SetFinalizer(fs, nil) // hip() sets it again if the object is used after an explicit close
	hipMu.Lock()
	h := hipFFT[uintptr(unsafe.Pointer(fs))]
	delete(hipFFT, uintptr(unsafe.Pointer(fs)))
	hipMu.Unlock()
	if h != nil {
		C.kzg_hip_fft_settings_free(h)
	}
}
func (ks *KZGSettings) CloseHip() {
	runtime.SetFinalizer(ks, nil) // hip() sets it again if the object is used after an explicit close
	hipMu.Lock()
	h := hipKZG[uintptr(unsafe.Pointer(ks))]
	delete(hipKZG, uintptr(unsafe.Pointer(ks)))
	hipMu.Unlock()
	if h != nil {
		C.kzg_hip_kzg_settings_free(h)
	}
}
func (fk *FK20SingleSettings) CloseHip() {
	runtime.SetFinalizer(fk, nil) // hip() sets it again if the object is used after an explicit close
	hipMu.Lock()
	h := hipFK20S[uintptr(unsafe.Pointer(fk))]
	delete(hipFK20S, uintptr(unsafe.Pointer(fk)))
	hipMu.Unlock()
	if h != nil {
		C.kzg_hip_fk20_single_settings_free(h)
	}
}
func (fk *FK20MultiSettings) CloseHip() {
	runtime.SetFinalizer(fk, nil) // hip() sets it again if the object is used after an explicit close
	hipMu.Lock()
	h := hipFK20M[uintptr(unsafe.Pointer(fk))]
	delete(hipFK20M, uintptr(unsafe.Pointer(fk)))
	hipMu.Unlock()
	if h != nil {
		C.kzg_hip_fk20_multi_settings_free(h)
	}
}

func frPtr(v []bls.Fr) unsafe.Pointer {
	if len(v) == 0 {
		return nil
	}
	return unsafe.Pointer(&v[0])
}

func g1Ptr(v []bls.G1Point) unsafe.Pointer {
	if len(v) == 0 {
		return nil
	}
	return unsafe.Pointer(&v[0])
}

// hipErr maps status 1-2 to the `error` the FFT layer returns (fft_fr.go:57-59,78-83; fft_g1.go:60-65).
func hipErr(st C.int, n int, maxWidth uint64) error {
	switch st {
	case C.KZG_HIP_OK:
		return nil
	case C.KZG_HIP_ERR_TOO_WIDE:
		return fmt.Errorf("got %d values but only have %d roots of unity", n, maxWidth)
	case C.KZG_HIP_ERR_NOT_POW2:
		return fmt.Errorf("got %d values but not a power of two", n)
	}
	panic(hipPanicText(st))
}

// hipMust maps every non-zero status to a panic, as the KZG / FK20 layer does (kzg.go:22-27,44-52,74-91;
// fk20_single.go:60-62,140-154; fk20_multi.go:28-31,60-69; bls/bls_kilic.go:133-135).
func hipMust(st C.int) {
	if st != C.KZG_HIP_OK {
		panic(hipPanicText(st))
	}
}

func hipPanicText(st C.int) string {
	switch st {
	case C.KZG_HIP_ERR_LEN_MISMATCH:
		return "kzg_hip: slice length mismatch"
	case C.KZG_HIP_ERR_UPPER_HALF:
		return "bad input, second half should be zeroed"
	case C.KZG_HIP_ERR_NO_DEVICE:
		return "kzg_hip: no gfx950 device (there is no CPU fallback in this build; drop -tags kzg_hip)"
	case C.KZG_HIP_ERR_HIP:
		return "kzg_hip: " + C.GoString(C.kzg_hip_last_error())
	}
	return fmt.Sprintf("kzg_hip: status %d", int(st))
}

func (fs *FFTSettings) hip() *C.kzg_hip_fft {
	hipMu.Lock()
	defer hipMu.Unlock()
	if h, ok := hipFFT[uintptr(unsafe.Pointer(fs))]; ok {
		return h
	}
	scale := uint8(0)
	for (uint64(1) << scale) < fs.MaxWidth {
		scale++
	}
	var h *C.kzg_hip_fft
	if st := C.kzg_hip_fft_settings_new(C.int(HipDeviceID), C.uint(scale), &h); st != C.KZG_HIP_OK {
		panic(hipPanicText(st)) // the C constructor frees whatever it had built; nothing to release here
	}
	hipFFT[uintptr(unsafe.Pointer(fs))] = h
	runtime.SetFinalizer(fs, (*FFTSettings).CloseHip)
	return h
}

func (ks *KZGSettings) hip() *C.kzg_hip_kzg {
	fh := ks.FFTSettings.hip()
	hipMu.Lock()
	defer hipMu.Unlock()
	if h, ok := hipKZG[uintptr(unsafe.Pointer(ks))]; ok {
		return h
	}
	var h *C.kzg_hip_kzg
	if st := C.kzg_hip_kzg_settings_new(fh, g1Ptr(ks.SecretG1), C.uint64_t(len(ks.SecretG1)), &h); st != C.KZG_HIP_OK {
		panic(hipPanicText(st))
	}
	hipKZG[uintptr(unsafe.Pointer(ks))] = h
	runtime.SetFinalizer(ks, (*KZGSettings).CloseHip)
	return h
}

//go:build kzg_hip && !bignum_pure && !bignum_hol256 && !bignum_hbls
// +build kzg_hip,!bignum_pure,!bignum_hol256,!bignum_hbls

package kzg

/*
#include "kzg_hip.h"
*/
import "C"

import (
	"runtime"

	"github.com/protolambda/go-kzg/bls"
)

// FFT replaces fft_fr.go:55-74.
func (fs *FFTSettings) FFT(vals []bls.Fr, inv bool) ([]bls.Fr, error) {
	defer runtime.KeepAlive(fs) // the finalizer must not free the device handle under a running call
	n := uint64(len(vals))
	if n > fs.MaxWidth {
		return nil, hipErr(C.KZG_HIP_ERR_TOO_WIDE, len(vals), fs.MaxWidth)
	}
	out := make([]bls.Fr, nextPowOf2(n))
	var outN C.uint64_t
	st := C.kzg_hip_fft_fr(fs.hip(), frPtr(vals), C.uint64_t(n), cBool(inv), frPtr(out), &outN)
	if err := hipErr(st, len(vals), fs.MaxWidth); err != nil {
		return nil, err
	}
	return out, nil
}

// InplaceFFT replaces fft_fr.go:76-105.
func (fs *FFTSettings) InplaceFFT(vals []bls.Fr, out []bls.Fr, inv bool) error {
	defer runtime.KeepAlive(fs) // the finalizer must not free the device handle under a running call
	st := C.kzg_hip_inplace_fft_fr(fs.hip(), frPtr(vals), frPtr(out), C.uint64_t(len(vals)), cBool(inv))
	return hipErr(st, len(vals), fs.MaxWidth)
}

// FFTG1 replaces fft_g1.go:58-94.
func (fs *FFTSettings) FFTG1(vals []bls.G1Point, inv bool) ([]bls.G1Point, error) {
	defer runtime.KeepAlive(fs) // the finalizer must not free the device handle under a running call
	out := make([]bls.G1Point, len(vals))
	st := C.kzg_hip_fft_g1(fs.hip(), g1Ptr(vals), C.uint64_t(len(vals)), cBool(inv), g1Ptr(out))
	if err := hipErr(st, len(vals), fs.MaxWidth); err != nil {
		return nil, err
	}
	return out, nil
}

// DASFFTExtension replaces das_extension.go:71-84 (in place, like the reference).
func (fs *FFTSettings) DASFFTExtension(vals []bls.Fr) {
	defer runtime.KeepAlive(fs) // the finalizer must not free the device handle under a running call
	if uint64(len(vals))*2 > fs.MaxWidth {
		panic("domain too small for extending requested values")
	}
	hipMust(C.kzg_hip_das_fft_extension(fs.hip(), frPtr(vals), C.uint64_t(len(vals))))
}

func cBool(b bool) C.int {
	if b {
		return 1
	}
	return 0
}

//go:build kzg_hip && !bignum_pure && !bignum_hol256 && !bignum_hbls
// +build kzg_hip,!bignum_pure,!bignum_hol256,!bignum_hbls

package kzg

/*
#include "kzg_hip.h"
*/
import "C"

import (
	"sync"

	"github.com/protolambda/go-kzg/bls"
)

var (
	hipCtxOnce sync.Once
	hipCtx     *C.kzg_hip_fft // scale-0 settings: a device context for calls that need no domain
)

func hipContext() *C.kzg_hip_fft {
	hipCtxOnce.Do(func() {
		hipMust(C.kzg_hip_fft_settings_new(C.int(HipDeviceID), 0, &hipCtx))
	})
	return hipCtx
}

// GenerateTestingSetup replaces setup.go:9-26 (**for testing purposes only**, as there): the G1 half -- n fixed-base multiplications
// [secret^i]G1, 65 536 of them for the scale-16 FK20Multi configuration -- runs on the device; the G2 half stays on the CPU backend (this
// library has no G2 arithmetic), with the reference's own loop.
func GenerateTestingSetup(secret string, n uint64) ([]bls.G1Point, []bls.G2Point) {
	var s bls.Fr
	bls.SetFr(&s, secret)
	s1Out := make([]bls.G1Point, n, n)
	if n > 0 {
		hipMust(C.kzg_hip_generate_testing_setup_g1(hipContext(), frPtr([]bls.Fr{s}), C.uint64_t(n), g1Ptr(s1Out)))
	}
	var sPow bls.Fr
	bls.CopyFr(&sPow, &bls.ONE)
	s2Out := make([]bls.G2Point, n, n)
	for i := uint64(0); i < n; i++ {
		bls.MulG2(&s2Out[i], &bls.GenG2, &sPow)
		var tmp bls.Fr
		bls.CopyFr(&tmp, &sPow)
		bls.MulModFr(&sPow, &tmp, &s)
	}
	return s1Out, s2Out
}

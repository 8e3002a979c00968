// sha256.hpp -- SHA-256 (FIPS 180-4) on the host, for the Fiat-Shamir transcript of eth.ComputeAggregateKZGProof
// (hashPolysComms / hashToBLSField, eth/helpers.go:113-133,235-260).  The transcript is ONE message over every blob of a block, a chain of
// dependent compressions: it stays on the host (x86 SHA extensions when the CPU has them, 1.5-2 GB/s; a portable loop otherwise) and runs
// while the device computes the blobs' commitments.  KZG_HIP_SHA256=portable forces the portable loop (tests compare the two).
#pragma once
#include <cstdint>
#include <cstring>

namespace kzg {

// compress `blocks` 64-byte blocks into the state (sha256.cpp: host code, built by the host compiler)
void sha256_blocks(uint32_t st[8], const uint8_t *p, size_t blocks);

struct sha256 {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t buf[64];
    size_t fill = 0;
    uint64_t total = 0;
    void blocks(const uint8_t *p, size_t n) { sha256_blocks(st, p, n); }
    void update(const void *data, size_t len) {
        const uint8_t *p = (const uint8_t *)data;
        total += len;
        if (fill) {
            const size_t take = len < 64 - fill ? len : 64 - fill;
            memcpy(buf + fill, p, take);
            fill += take; p += take; len -= take;
            if (fill < 64) return;
            blocks(buf, 1);
            fill = 0;
        }
        if (len >= 64) { blocks(p, len / 64); p += len & ~(size_t)63; len &= 63; }
        if (len) { memcpy(buf, p, len); fill = len; }
    }
    void update_u64_le(uint64_t v) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i)); update(b, 8); }
    void final(uint8_t out[32]) {
        const uint64_t bits = total * 8;
        uint8_t pad[72] = {0x80};
        const size_t padlen = (fill < 56 ? 56 : 120) - fill;
        for (int i = 0; i < 8; i++) pad[padlen + i] = (uint8_t)(bits >> (56 - 8 * i));
        update(pad, padlen + 8);
        for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16); out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i]; }
    }
};

}   // namespace kzg

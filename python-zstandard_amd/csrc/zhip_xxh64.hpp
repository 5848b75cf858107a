// zhip_xxh64.hpp -- XXH64 (public algorithm) for the optional frame checksum: zstd stores the low 32 bits of the digest of
// the uncompressed content after the last block (zstd.c:28325-28329) and verifies them on decode (zstd.c:44270-44277).
#pragma once
#include "zhip_device.hpp"

// one lane, serial (checksums are opt-in in python-zstandard: write_checksum defaults to False)
ZH_COLD uint64_t ze_xxh64(const uint8_t* p, uint32_t len)
{
    const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
#define ZE_ROTL(x, r) (((x) << (r)) | ((x) >> (64 - (r))))
#define ZE_ROUND(acc, in) (ZE_ROTL((acc) + (in) * P2, 31) * P1)
    const uint8_t* const end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
        do {
            v1 = ZE_ROUND(v1, zh_ld64(p)); v2 = ZE_ROUND(v2, zh_ld64(p + 8)); v3 = ZE_ROUND(v3, zh_ld64(p + 16)); v4 = ZE_ROUND(v4, zh_ld64(p + 24));
            p += 32;
        } while (p + 32 <= end);
        h = ZE_ROTL(v1, 1) + ZE_ROTL(v2, 7) + ZE_ROTL(v3, 12) + ZE_ROTL(v4, 18);
        h = (h ^ ZE_ROUND(0, v1)) * P1 + P4; h = (h ^ ZE_ROUND(0, v2)) * P1 + P4;
        h = (h ^ ZE_ROUND(0, v3)) * P1 + P4; h = (h ^ ZE_ROUND(0, v4)) * P1 + P4;
    } else h = P5;
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= ZE_ROUND(0, zh_ld64(p)); h = ZE_ROTL(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)zh_ld32(p) * P1; h = ZE_ROTL(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p++) * P5; h = ZE_ROTL(h, 11) * P1; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
#undef ZE_ROUND
#undef ZE_ROTL
    return h;
}


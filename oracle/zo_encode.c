#include "zstd_oracle.h"
#include "zo_common.h"
size_t zo_compress_bound(size_t n) { return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0); }
int64_t zo_compress_frame(void* dst, size_t dstCap, const void* src, size_t srcSize, int level, unsigned flags, const void* dict, size_t dictSize)
{ (void)dst; (void)dstCap; (void)src; (void)srcSize; (void)level; (void)flags; (void)dict; (void)dictSize; return -ZO_E_GENERIC; }

/* tests/abi_stub_check.c -- the struct literals and calls INTEGRATION.md shows a python-zstandard maintainer, compiled against
 * include/zstd_hip.h by tests/test_host_and_abi.py (gcc -c, C11, -Wall -Wextra -Werror): the document cannot drift from the header.
 * Never linked or run. */
#include <stdlib.h>
#include "zstd_hip.h"

struct FakeDict { void* dictData; size_t dictSize; int dictType; };
struct FakeCompressor { int level, contentSizeFlag, checksumFlag, dictIDFlag; struct FakeDict* dict; };

int stub_compress(struct FakeCompressor* compressor, const void** data, const size_t* sizes, size_t n, zhip_outbuf** out, size_t* nOut)
{
    zhip_cparams p = { compressor->level, compressor->contentSizeFlag, compressor->checksumFlag, compressor->dictIDFlag,
                       compressor->dict ? compressor->dict->dictData : NULL, compressor->dict ? compressor->dict->dictSize : 0,
                       compressor->dict ? (int)compressor->dict->dictType : ZHIP_DICT_AUTO,
                       ZHIP_FORMAT_ZSTD1,
                       { 0, 0, 0, 0, 0, 0, 0 } };
    zhip_item* items = (zhip_item*)malloc(n * sizeof(zhip_item));
    if (!items) return ZHIP_ERR_NO_MEMORY;
    for (size_t i = 0; i < n; i++) { items[i].src = data[i]; items[i].srcSize = sizes[i]; items[i].dstSize = 0; }
    zhip_error err;
    const int rc = zhip_compress_batch(&p, items, n, out, nOut, &err);
    free(items);
    if (rc == ZHIP_ERR_ZSTD) (void)zhip_error_name(err.zstdErr);
    else if (rc) (void)zhip_last_error();
    return rc;
}

int stub_decompress(const struct FakeDict* dict, const zhip_item* items, size_t n, zhip_outbuf** out, size_t* nOut, zhip_error* err)
{
    zhip_dparams d = { dict ? dict->dictData : NULL, dict ? dict->dictSize : 0, 0, dict ? dict->dictType : ZHIP_DICT_AUTO, ZHIP_FORMAT_ZSTD1 };
    return zhip_decompress_batch(&d, items, n, /*requireSizes=*/0, out, nOut, err);
}

void stub_release(zhip_outbuf* out, size_t nOut)
{
    for (size_t i = 0; i < nOut; i++) { zhip_free_payload(out[i].data); free(out[i].segs); }     /* what BufferWithSegments' deallocator does */
    zhip_free_outbufs(out, nOut, /*freePayload=*/0);
}

int stub_device(const void* dictBytes, size_t dictSize, const void* d_src, const zhip_segment* d_srcSegs, size_t n, void* d_dst,
                const zhip_segment* d_dstSegs, uint64_t* d_outSizes, int32_t* d_status, void* stream)
{
    zhip_ctx* ctx = zhip_ctx_create();
    zhip_error err;
    if (!ctx) return ZHIP_ERR_HIP;
    zhip_ctx_set_ddict(ctx, dictBytes, dictSize, ZHIP_DICT_AUTO);
    zhip_ctx_set_size_hint(ctx, 1u << 20);
    int rc = zhip_decompress_batch_device(ctx, d_src, d_srcSegs, n, d_dst, d_dstSegs, d_outSizes, d_status, stream);
    if (!rc) rc = zhip_ctx_sync(ctx, stream, d_status, n, &err);
    zhip_ctx_destroy(ctx);
    return rc;
}

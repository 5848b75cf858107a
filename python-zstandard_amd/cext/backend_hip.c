/* backend_hip.c -- CPython extension: the reference's hot-path API over the C ABI of libzstd_hip.so.
 *
 * The reference's host side for this path is a C extension (the files under c-ext/ against the CPython API); this file is the same
 * thing for the MI355X backend: ZstdCompressor.compress / multi_compress_to_buffer (c-ext/compressor.c:509-574, :1340-1503),
 * ZstdDecompressor.decompress / multi_decompress_to_buffer (c-ext/decompressor.c:263-395, :1459-1710), the zero-copy buffer
 * types (c-ext/bufferutil.c, c-ext/python-zstandard.h:307-368), ZstdCompressionDict's consumer side
 * (c-ext/compressiondict.c:164-348) and ZstdError -- same names, argument meaning and error messages -- with the frame loop
 * replaced by zhip_compress_batch / zhip_decompress_batch (include/zstd_hip.h). The GIL is released around both calls like
 * the reference does around its worker pool (compressor.c:1170-1223). The library is bound with dlopen at import (after
 * importing torch when it is installed, so that both use the same HIP runtime instance); there is no CPU fallback.
 *
 * This extension IS the package's implementation of the reference's names (python-zstandard_amd/__init__.py re-exports it);
 * there is no second host implementation.
 */
#define _GNU_SOURCE
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>
#include <dlfcn.h>
#include <unistd.h>
#include <stdint.h>
#include <string.h>
#include "zstd_hip.h"

/* ------------------------------------------------------------------------------------------ the bound C ABI */
static struct {
    void* handle;
    const char* (*last_error)(void);
    const char* (*error_name)(int);
    uint64_t (*frame_content_size)(const void*, size_t);
    int64_t (*find_frame_compressed_size)(const void*, size_t);
    int (*compress_batch)(const zhip_cparams*, const zhip_item*, size_t, zhip_outbuf**, size_t*, zhip_error*);
    int (*decompress_batch)(const zhip_dparams*, const zhip_item*, size_t, int, zhip_outbuf**, size_t*, zhip_error*);
    void (*free_outbufs)(zhip_outbuf*, size_t, int);
    void (*free_payload)(void*);
    int (*abi_version)(void);
    uint64_t (*frame_content_size_format)(const void*, size_t, int);
    int64_t (*find_frame_compressed_size_format)(const void*, size_t, int);
    void (*get_cparams)(int, uint64_t, size_t, zhip_compression_parameters*);
    size_t (*thread_memory_size)(void);
} Z;

static PyObject* ZstdError;
#define FLAG_ALLOW_SHORT 2      /* zhip_decompress_batch requireSizes bit: dstSize is a capacity, not an exact size */
#define FORMAT_ZSTD1 0
#define FORMAT_ZSTD1_MAGICLESS 1
#define MAX_COMPRESSION_LEVEL 22
#define DICT_TYPE_AUTO 0
#define DICT_TYPE_RAWCONTENT 1
#define DICT_TYPE_FULLDICT 2

static int bind_library(PyObject* module)
{
    char path[4096];
    const char* env = getenv("ZHIP_LIB");
    if (env && *env) snprintf(path, sizeof path, "%s", env);
    else {
        Dl_info info;                                   /* this extension's own file: libzstd_hip.so sits in csrc/ next to it */
        (void)module;
        if (!dladdr((void*)&Z, &info) || !info.dli_fname) { PyErr_SetString(PyExc_ImportError, "cannot locate the extension module on disk"); return -1; }
        const char* f = info.dli_fname;
        const char* slash = strrchr(f, '/');
        const size_t dir = slash ? (size_t)(slash - f) : 0;
        if (dir) snprintf(path, sizeof path, "%.*s/csrc/libzstd_hip.so", (int)dir, f);
        else snprintf(path, sizeof path, "csrc/libzstd_hip.so");
    }
    {   /* torch bundles its own HIP runtime: load it first so that this library binds to the same instance */
        PyObject* t = PyImport_ImportModule("torch");
        if (!t) PyErr_Clear(); else Py_DECREF(t);
    }
    Z.handle = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!Z.handle) {
        PyErr_Format(PyExc_ImportError, "libzstd_hip.so is not built (%s): %s; this backend has no CPU fallback", path, dlerror());
        return -1;
    }
#define BIND(field, name) do { *(void**)&Z.field = dlsym(Z.handle, name); if (!Z.field) { PyErr_Format(PyExc_ImportError, "libzstd_hip.so lacks %s", name); return -1; } } while (0)
    BIND(last_error, "zhip_last_error"); BIND(error_name, "zhip_error_name"); BIND(frame_content_size, "zhip_frame_content_size");
    BIND(find_frame_compressed_size, "zhip_find_frame_compressed_size"); BIND(compress_batch, "zhip_compress_batch");
    BIND(decompress_batch, "zhip_decompress_batch"); BIND(free_outbufs, "zhip_free_outbufs"); BIND(abi_version, "zhip_abi_version");
    BIND(free_payload, "zhip_free_payload");
    BIND(frame_content_size_format, "zhip_frame_content_size_format"); BIND(find_frame_compressed_size_format, "zhip_find_frame_compressed_size_format");
    BIND(get_cparams, "zhip_get_cparams"); BIND(thread_memory_size, "zhip_thread_memory_size");
#undef BIND
    if (Z.abi_version() != ZHIP_ABI_VERSION) { PyErr_SetString(PyExc_ImportError, "libzstd_hip.so ABI mismatch"); return -1; }
    return 0;
}

/* ------------------------------------------------------------------------------------------ buffer types */
typedef struct {
    PyObject_HEAD
    Py_buffer parent;           /* payload when built from Python objects (parent.obj != NULL) */
    Py_buffer segParent;
    void* data; unsigned long long dataSize;
    zhip_segment* segments; Py_ssize_t segmentCount;
    int useFree;                /* data / segments came from the C ABI: release them (bufferutil.c:13-37; data through zhip_free_payload) */
} BufferWithSegments;

typedef struct { PyObject_HEAD PyObject* parent; void* data; Py_ssize_t dataSize; unsigned long long offset; } BufferSegment;
typedef struct { PyObject_HEAD PyObject* parent; zhip_segment* segments; Py_ssize_t segmentCount; } BufferSegments;
typedef struct { PyObject_HEAD BufferWithSegments** buffers; Py_ssize_t bufferCount; Py_ssize_t* firstElements; } BufferCollection;

static PyTypeObject BufferWithSegmentsType = { PyVarObject_HEAD_INIT(NULL, 0) }, BufferSegmentType = { PyVarObject_HEAD_INIT(NULL, 0) },
                    BufferSegmentsType = { PyVarObject_HEAD_INIT(NULL, 0) }, BufferCollectionType = { PyVarObject_HEAD_INIT(NULL, 0) };

static void bws_dealloc(BufferWithSegments* self)
{
    if (self->useFree) { Z.free_payload(self->data); free(self->segments); }     /* payloads may be pinned pool blocks (zstd_hip.h) */
    if (self->parent.obj) PyBuffer_Release(&self->parent);
    if (self->segParent.obj) PyBuffer_Release(&self->segParent);
    Py_TYPE(self)->tp_free((PyObject*)self);
}

static int bws_init(BufferWithSegments* self, PyObject* args, PyObject* kwargs)
{
    static char* kwlist[] = { "data", "segments", NULL };
    Py_buffer segs;
    memset(&self->parent, 0, sizeof self->parent); memset(&self->segParent, 0, sizeof self->segParent);
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "y*y*:BufferWithSegments", kwlist, &self->parent, &segs)) { self->parent.obj = NULL; return -1; }
    self->segParent = segs;
    if (segs.len % (Py_ssize_t)sizeof(zhip_segment)) {
        PyErr_Format(PyExc_ValueError, "segments array size is not a multiple of %zu", sizeof(zhip_segment));
        return -1;
    }
    self->data = self->parent.buf; self->dataSize = (unsigned long long)self->parent.len;
    self->segments = (zhip_segment*)segs.buf; self->segmentCount = segs.len / (Py_ssize_t)sizeof(zhip_segment);
    for (Py_ssize_t i = 0; i < self->segmentCount; i++)
        if (self->segments[i].offset + self->segments[i].length > self->dataSize) {
            PyErr_SetString(PyExc_ValueError, "offset within segments array references memory outside buffer");
            return -1;
        }
    return 0;
}

/* BufferWithSegments_FromMemory (bufferutil.c:107-148): takes ownership of two malloc()ed blocks */
static BufferWithSegments* bws_from_memory(void* data, unsigned long long dataSize, zhip_segment* segs, Py_ssize_t n)
{
    for (Py_ssize_t i = 0; i < n; i++)
        if (segs[i].offset + segs[i].length > dataSize) { PyErr_SetString(PyExc_ValueError, "offset in segments overflows buffer size"); return NULL; }
    BufferWithSegments* r = (BufferWithSegments*)BufferWithSegmentsType.tp_alloc(&BufferWithSegmentsType, 0);
    if (!r) return NULL;
    memset(&r->parent, 0, sizeof r->parent); memset(&r->segParent, 0, sizeof r->segParent);
    r->useFree = 1; r->data = data; r->dataSize = dataSize; r->segments = segs; r->segmentCount = n;
    return r;
}

static Py_ssize_t bws_length(BufferWithSegments* self) { return self->segmentCount; }

static PyObject* bws_item(BufferWithSegments* self, Py_ssize_t i)
{
    if (i < 0) { PyErr_SetString(PyExc_IndexError, "offset must be non-negative"); return NULL; }
    if (i >= self->segmentCount) { PyErr_Format(PyExc_IndexError, "offset must be less than %zd", self->segmentCount); return NULL; }
    BufferSegment* r = (BufferSegment*)BufferSegmentType.tp_alloc(&BufferSegmentType, 0);
    if (!r) return NULL;
    r->parent = (PyObject*)self; Py_INCREF(self);
    r->data = (char*)self->data + self->segments[i].offset;
    r->dataSize = (Py_ssize_t)self->segments[i].length; r->offset = self->segments[i].offset;
    return (PyObject*)r;
}

static int bws_getbuffer(BufferWithSegments* self, Py_buffer* view, int flags)
{
    return PyBuffer_FillInfo(view, (PyObject*)self, self->data, (Py_ssize_t)self->dataSize, 1, flags);
}
static PyObject* bws_tobytes(BufferWithSegments* self, PyObject* noargs)
{
    (void)noargs;
    return PyBytes_FromStringAndSize((const char*)self->data, (Py_ssize_t)self->dataSize);
}
static PyObject* bws_segments(BufferWithSegments* self, PyObject* noargs)
{
    (void)noargs;
    BufferSegments* r = (BufferSegments*)BufferSegmentsType.tp_alloc(&BufferSegmentsType, 0);
    if (!r) return NULL;
    r->parent = (PyObject*)self; Py_INCREF(self); r->segments = self->segments; r->segmentCount = self->segmentCount;
    return (PyObject*)r;
}
static PySequenceMethods bws_sq = { (lenfunc)bws_length, 0, 0, (ssizeargfunc)bws_item };
static PyBufferProcs bws_as_buffer = { (getbufferproc)bws_getbuffer, 0 };
static PyMethodDef bws_methods[] = {
    { "segments", (PyCFunction)bws_segments, METH_NOARGS, "the segment table" },
    { "tobytes", (PyCFunction)bws_tobytes, METH_NOARGS, "copy of the payload" },
    { NULL, NULL, 0, NULL } };
static PyMemberDef bws_members[] = { { "size", T_ULONGLONG, offsetof(BufferWithSegments, dataSize), READONLY, "total size of the buffer in bytes" }, { NULL, 0, 0, 0, NULL } };

static void seg_dealloc(BufferSegment* self) { Py_CLEAR(self->parent); Py_TYPE(self)->tp_free((PyObject*)self); }
static Py_ssize_t seg_length(BufferSegment* self) { return self->dataSize; }
static int seg_getbuffer(BufferSegment* self, Py_buffer* view, int flags) { return PyBuffer_FillInfo(view, (PyObject*)self, self->data, self->dataSize, 1, flags); }
static PyObject* seg_tobytes(BufferSegment* self, PyObject* noargs) { (void)noargs; return PyBytes_FromStringAndSize((const char*)self->data, self->dataSize); }
static PySequenceMethods seg_sq = { (lenfunc)seg_length, 0, 0, 0 };
static PyBufferProcs seg_as_buffer = { (getbufferproc)seg_getbuffer, 0 };
static PyMethodDef seg_methods[] = { { "tobytes", (PyCFunction)seg_tobytes, METH_NOARGS, "copy of the segment" }, { NULL, NULL, 0, NULL } };
static PyMemberDef seg_members[] = { { "offset", T_ULONGLONG, offsetof(BufferSegment, offset), READONLY, "offset of the segment within its parent buffer" }, { NULL, 0, 0, 0, NULL } };

static void segs_dealloc(BufferSegments* self) { Py_CLEAR(self->parent); Py_TYPE(self)->tp_free((PyObject*)self); }
static int segs_getbuffer(BufferSegments* self, Py_buffer* view, int flags)
{
    return PyBuffer_FillInfo(view, (PyObject*)self, self->segments, self->segmentCount * (Py_ssize_t)sizeof(zhip_segment), 1, flags);
}
static PyObject* segs_tobytes(BufferSegments* self, PyObject* noargs)
{
    (void)noargs;
    return PyBytes_FromStringAndSize((const char*)self->segments, self->segmentCount * (Py_ssize_t)sizeof(zhip_segment));
}
static PyBufferProcs segs_as_buffer = { (getbufferproc)segs_getbuffer, 0 };
static PyMethodDef segs_methods[] = { { "tobytes", (PyCFunction)segs_tobytes, METH_NOARGS, "copy of the segment table" }, { NULL, NULL, 0, NULL } };

static void coll_dealloc(BufferCollection* self)
{
    if (self->buffers) { for (Py_ssize_t i = 0; i < self->bufferCount; i++) Py_XDECREF(self->buffers[i]); PyMem_Free(self->buffers); }
    PyMem_Free(self->firstElements);
    Py_TYPE(self)->tp_free((PyObject*)self);
}
static int coll_init(BufferCollection* self, PyObject* args, PyObject* kwargs)
{
    (void)kwargs;
    const Py_ssize_t n = PyTuple_Size(args);
    if (n < 1) { PyErr_SetString(PyExc_ValueError, "must pass at least 1 argument"); return -1; }
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* o = PyTuple_GET_ITEM(args, i);
        if (!PyObject_TypeCheck(o, &BufferWithSegmentsType)) { PyErr_SetString(PyExc_TypeError, "arguments must be BufferWithSegments instances"); return -1; }
        if (((BufferWithSegments*)o)->segmentCount == 0 || ((BufferWithSegments*)o)->dataSize == 0) {
            PyErr_SetString(PyExc_ValueError, "ZstdBufferWithSegments cannot be empty"); return -1;
        }
    }
    self->buffers = (BufferWithSegments**)PyMem_Calloc((size_t)n, sizeof(BufferWithSegments*));
    self->firstElements = (Py_ssize_t*)PyMem_Calloc((size_t)n, sizeof(Py_ssize_t));
    if (!self->buffers || !self->firstElements) { PyErr_NoMemory(); return -1; }
    self->bufferCount = n;
    Py_ssize_t total = 0;
    for (Py_ssize_t i = 0; i < n; i++) {
        BufferWithSegments* b = (BufferWithSegments*)PyTuple_GET_ITEM(args, i);
        Py_INCREF(b); self->buffers[i] = b;
        total += b->segmentCount; self->firstElements[i] = total;
    }
    return 0;
}
static Py_ssize_t coll_length(BufferCollection* self) { return self->bufferCount ? self->firstElements[self->bufferCount - 1] : 0; }
static PyObject* coll_item(BufferCollection* self, Py_ssize_t i)
{
    if (i < 0) { PyErr_SetString(PyExc_IndexError, "offset must be non-negative"); return NULL; }
    if (i >= coll_length(self)) { PyErr_Format(PyExc_IndexError, "offset must be less than %zd", coll_length(self)); return NULL; }
    Py_ssize_t prev = 0;
    for (Py_ssize_t b = 0; b < self->bufferCount; b++) {
        if (i < self->firstElements[b]) return bws_item(self->buffers[b], i - prev);
        prev = self->firstElements[b];
    }
    PyErr_SetString(ZstdError, "error resolving segment; this should not happen");
    return NULL;
}
static PyObject* coll_size(BufferCollection* self, PyObject* noargs)
{
    (void)noargs;
    unsigned long long total = 0;
    for (Py_ssize_t b = 0; b < self->bufferCount; b++)
        for (Py_ssize_t i = 0; i < self->buffers[b]->segmentCount; i++) total += self->buffers[b]->segments[i].length;
    return PyLong_FromUnsignedLongLong(total);
}
static PySequenceMethods coll_sq = { (lenfunc)coll_length, 0, 0, (ssizeargfunc)coll_item };
static PyMethodDef coll_methods[] = { { "size", (PyCFunction)coll_size, METH_NOARGS, "total size of all segments" }, { NULL, NULL, 0, NULL } };

/* ------------------------------------------------------------------------------------------ ZstdCompressionDict */
/* precomputed: precompute_compress() was called -- `pre` holds the parameters the reference's ZSTD_createCDict_advanced would digest the
 * dictionary with (compressiondict.c:228-286); frames made with this dictionary then follow THEM, not the compressor's level */
typedef struct { PyObject_HEAD PyObject* data; int dictType; int precomputed; zhip_compression_parameters pre; } CompressionDict;
static PyTypeObject CompressionDictType = { PyVarObject_HEAD_INIT(NULL, 0) };
static void dict_dealloc(CompressionDict* self) { Py_CLEAR(self->data); Py_TYPE(self)->tp_free((PyObject*)self); }
static int dict_init(CompressionDict* self, PyObject* args, PyObject* kwargs)
{
    static char* kwlist[] = { "data", "dict_type", NULL };
    Py_buffer src; unsigned dictType = DICT_TYPE_AUTO;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "y*|I:ZstdCompressionDict", kwlist, &src, &dictType)) return -1;
    if (dictType != DICT_TYPE_AUTO && dictType != DICT_TYPE_RAWCONTENT && dictType != DICT_TYPE_FULLDICT) {
        PyBuffer_Release(&src);
        PyErr_Format(PyExc_ValueError, "invalid dictionary load mode: %d; must use DICT_TYPE_* constants", dictType);
        return -1;
    }
    Py_XSETREF(self->data, PyBytes_FromStringAndSize((const char*)src.buf, src.len));
    PyBuffer_Release(&src);
    self->dictType = (int)dictType; self->precomputed = 0; memset(&self->pre, 0, sizeof self->pre);
    return self->data ? 0 : -1;
}
static Py_ssize_t dict_length(CompressionDict* self) { return PyBytes_GET_SIZE(self->data); }
static PyObject* dict_as_bytes(CompressionDict* self, PyObject* noargs) { (void)noargs; Py_INCREF(self->data); return self->data; }
static PyObject* dict_dict_id(CompressionDict* self, PyObject* noargs)
{
    (void)noargs;
    const unsigned char* d = (const unsigned char*)PyBytes_AS_STRING(self->data);
    unsigned long id = 0;
    if (self->dictType != DICT_TYPE_RAWCONTENT && PyBytes_GET_SIZE(self->data) >= 8 && d[0] == 0x37 && d[1] == 0xA4 && d[2] == 0x30 && d[3] == 0xEC)
        id = (unsigned long)d[4] | ((unsigned long)d[5] << 8) | ((unsigned long)d[6] << 16) | ((unsigned long)d[7] << 24);
    return PyLong_FromUnsignedLong(id);
}
static PyObject* dict_precompute(CompressionDict* self, PyObject* args, PyObject* kwargs);      /* below, after ZstdCompressionParameters */
static PySequenceMethods dict_sq = { (lenfunc)dict_length, 0, 0, 0 };
static PyMethodDef dict_methods[] = {
    { "as_bytes", (PyCFunction)dict_as_bytes, METH_NOARGS, "raw dictionary bytes" },
    { "dict_id", (PyCFunction)dict_dict_id, METH_NOARGS, "dictionary id" },
    { "precompute_compress", (PyCFunction)dict_precompute, METH_VARARGS | METH_KEYWORDS, "precompute_compress(level=0, compression_params=None): digest the dictionary on the device for these parameters" },
    { NULL, NULL, 0, NULL } };

/* ------------------------------------------------------------------------------------------ sources (compressor.c:1369-1466) */
typedef struct { zhip_item* items; Py_ssize_t n; Py_buffer* views; Py_ssize_t nViews; unsigned long long totalSize; } Sources;
static void sources_free(Sources* s)
{
    for (Py_ssize_t i = 0; i < s->nViews; i++) PyBuffer_Release(&s->views[i]);
    PyMem_Free(s->views); PyMem_Free(s->items);
    memset(s, 0, sizeof *s);
}
static int sources_push_bws(Sources* s, BufferWithSegments* b, Py_ssize_t* at)
{
    for (Py_ssize_t i = 0; i < b->segmentCount; i++) {
        s->items[*at].src = (char*)b->data + b->segments[i].offset; s->items[*at].srcSize = (size_t)b->segments[i].length; s->items[*at].dstSize = 0;
        s->totalSize += b->segments[i].length; (*at)++;
    }
    return 0;
}
static int sources_collect(Sources* s, PyObject* data, const char* typeError)
{
    memset(s, 0, sizeof *s);
    if (PyObject_TypeCheck(data, &BufferWithSegmentsType)) {
        BufferWithSegments* b = (BufferWithSegments*)data;
        s->items = (zhip_item*)PyMem_Calloc((size_t)(b->segmentCount ? b->segmentCount : 1), sizeof(zhip_item));
        if (!s->items) { PyErr_NoMemory(); return -1; }
        Py_ssize_t at = 0; sources_push_bws(s, b, &at); s->n = at;
    } else if (PyObject_TypeCheck(data, &BufferCollectionType)) {
        BufferCollection* c = (BufferCollection*)data;
        const Py_ssize_t total = coll_length(c);
        s->items = (zhip_item*)PyMem_Calloc((size_t)(total ? total : 1), sizeof(zhip_item));
        if (!s->items) { PyErr_NoMemory(); return -1; }
        Py_ssize_t at = 0;
        for (Py_ssize_t b = 0; b < c->bufferCount; b++) sources_push_bws(s, c->buffers[b], &at);
        s->n = at;
    } else if (PyList_Check(data)) {
        const Py_ssize_t n = PyList_GET_SIZE(data);
        s->items = (zhip_item*)PyMem_Calloc((size_t)(n ? n : 1), sizeof(zhip_item));
        s->views = (Py_buffer*)PyMem_Calloc((size_t)(n ? n : 1), sizeof(Py_buffer));
        if (!s->items || !s->views) { PyErr_NoMemory(); sources_free(s); return -1; }
        for (Py_ssize_t i = 0; i < n; i++) {
            if (PyObject_GetBuffer(PyList_GET_ITEM(data, i), &s->views[i], PyBUF_CONTIG_RO) != 0) {
                PyErr_Clear();
                PyErr_Format(PyExc_TypeError, "item %zd not a bytes like object", i);
                sources_free(s); return -1;
            }
            s->nViews = i + 1;
            s->items[i].src = s->views[i].buf; s->items[i].srcSize = (size_t)s->views[i].len; s->items[i].dstSize = 0;
            s->totalSize += (unsigned long long)s->views[i].len;
        }
        s->n = n;
    } else { PyErr_SetString(PyExc_TypeError, typeError); return -1; }
    return 0;
}

static PyObject* collection_from_outbufs(zhip_outbuf* out, size_t nOut)
{
    PyObject* args = PyTuple_New((Py_ssize_t)nOut);
    if (!args) { Z.free_outbufs(out, nOut, 1); return NULL; }
    for (size_t i = 0; i < nOut; i++) {
        BufferWithSegments* b = bws_from_memory(out[i].data, out[i].dataSize, out[i].segs, (Py_ssize_t)out[i].nSegs);
        if (!b) {       /* buffers 0..i-1 belong to the tuple already; free the rest, then the array */
            for (size_t k = i; k < nOut; k++) { Z.free_payload(out[k].data); free(out[k].segs); }
            Z.free_outbufs(out, nOut, 0); Py_DECREF(args); return NULL;
        }
        PyTuple_SET_ITEM(args, (Py_ssize_t)i, (PyObject*)b);
    }
    Z.free_outbufs(out, nOut, 0);                   /* payloads now belong to the BufferWithSegments objects */
    PyObject* r = PyObject_CallObject((PyObject*)&BufferCollectionType, args);
    Py_DECREF(args);
    return r;
}

/* ------------------------------------------------------------------------------------------ ZstdCompressionParameters (c-ext/compressionparams.c) */
typedef struct {
    PyObject_HEAD
    int format, compressionLevel, windowLog, hashLog, chainLog, searchLog, minMatch, targetLength, strategy;
    int contentSizeFlag, checksumFlag, dictIDFlag, jobSize, overlapLog, forceMaxWindow, enableLDM, ldmHashLog, ldmMinMatch, ldmBucketSizeLog, ldmHashRateLog, threads;
} CompressionParameters;
static PyTypeObject CompressionParametersType = { PyVarObject_HEAD_INIT(NULL, 0) };
static void cparams_dealloc(CompressionParameters* self) { Py_TYPE(self)->tp_free((PyObject*)self); }
/* the bounds ZSTD_CCtxParams_setParameter enforces (ZSTD_cParam_getBounds, zstd.c:23370-23560); 0 always means "default" */
static int cparams_bound(int v, int lo, int hi) { return v == 0 || (v >= lo && v <= hi); }
static int cparams_init(CompressionParameters* self, PyObject* args, PyObject* kwargs)
{
    static char* kwlist[] = { "format", "compression_level", "window_log", "hash_log", "chain_log", "search_log", "min_match", "target_length",
                              "strategy", "write_content_size", "write_checksum", "write_dict_id", "job_size", "overlap_log", "force_max_window",
                              "enable_ldm", "ldm_hash_log", "ldm_min_match", "ldm_bucket_size_log", "ldm_hash_rate_log", "threads", NULL };
    int format = 0, level = 0, windowLog = 0, hashLog = 0, chainLog = 0, searchLog = 0, minMatch = 0, targetLength = 0, strategy = -1;
    int contentSize = 1, checksum = 0, dictID = 0, jobSize = 0, overlapLog = -1, forceMaxWindow = 0, enableLDM = 0, ldmHashLog = 0, ldmMinMatch = 0,
        ldmBucketSizeLog = 0, ldmHashRateLog = -1, threads = 0;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "|iiiiiiiiiiiiiiiiiiiii:ZstdCompressionParameters", kwlist, &format, &level, &windowLog, &hashLog,
                                     &chainLog, &searchLog, &minMatch, &targetLength, &strategy, &contentSize, &checksum, &dictID, &jobSize, &overlapLog,
                                     &forceMaxWindow, &enableLDM, &ldmHashLog, &ldmMinMatch, &ldmBucketSizeLog, &ldmHashRateLog, &threads)) return -1;
    if (strategy == -1) strategy = 0;
    if (overlapLog == -1) overlapLog = 0;
    if (ldmHashRateLog == -1) ldmHashRateLog = 0;
    if (threads < 0) { long n = sysconf(_SC_NPROCESSORS_ONLN); threads = n > 0 ? (int)n : 1; }
    if (!(format == FORMAT_ZSTD1 || format == FORMAT_ZSTD1_MAGICLESS) || !cparams_bound(windowLog, 10, 31) || !cparams_bound(hashLog, 6, 30) ||
        !cparams_bound(chainLog, 6, 30) || !cparams_bound(searchLog, 1, 30) || !cparams_bound(minMatch, 3, 7) || targetLength < 0 || targetLength > (1 << 17) ||
        !cparams_bound(strategy, 1, 9) || threads > 256 || !cparams_bound(overlapLog, 0, 9) || !cparams_bound(ldmHashLog, 6, 30) ||
        !cparams_bound(ldmMinMatch, 4, 4096) || ldmBucketSizeLog < 0 || ldmBucketSizeLog > 8 || ldmHashRateLog < 0 || ldmHashRateLog > 25) {
        PyErr_SetString(ZstdError, "unable to set compression context parameter: Parameter is out of bound");
        return -1;
    }
    self->format = format; self->compressionLevel = level; self->windowLog = windowLog; self->hashLog = hashLog; self->chainLog = chainLog;
    self->searchLog = searchLog; self->minMatch = minMatch; self->targetLength = targetLength; self->strategy = strategy;
    self->contentSizeFlag = contentSize != 0; self->checksumFlag = checksum != 0; self->dictIDFlag = dictID != 0; self->jobSize = jobSize;
    self->overlapLog = overlapLog; self->forceMaxWindow = forceMaxWindow; self->enableLDM = enableLDM; self->ldmHashLog = ldmHashLog;
    self->ldmMinMatch = ldmMinMatch; self->ldmBucketSizeLog = ldmBucketSizeLog; self->ldmHashRateLog = ldmHashRateLog; self->threads = threads;
    return 0;
}
/* from_level(level, source_size=0, dict_size=0, **kwargs): the parameters libzstd derives (ZSTD_getCParams) become explicit values unless
 * the caller names them (compressionparams.c:231-345) */
static PyObject* cparams_from_level(PyObject* undef, PyObject* args, PyObject* kwargs)
{
    (void)undef;
    int level; unsigned long long sourceSize = 0; Py_ssize_t dictSize = 0;
    if (!PyArg_ParseTuple(args, "i:from_level", &level)) return NULL;
    PyObject* kw = kwargs ? PyDict_Copy(kwargs) : PyDict_New();
    if (!kw) return NULL;
    PyObject* v;
    if ((v = PyDict_GetItemString(kw, "source_size")) != NULL) {
        sourceSize = PyLong_AsUnsignedLongLong(v);
        if (sourceSize == (unsigned long long)-1 && PyErr_Occurred()) { Py_DECREF(kw); return NULL; }
        PyDict_DelItemString(kw, "source_size");
    }
    if ((v = PyDict_GetItemString(kw, "dict_size")) != NULL) {
        dictSize = PyLong_AsSsize_t(v);
        if (dictSize == -1 && PyErr_Occurred()) { Py_DECREF(kw); return NULL; }
        PyDict_DelItemString(kw, "dict_size");
    }
    zhip_compression_parameters cp;
    Z.get_cparams(level, (uint64_t)sourceSize, (size_t)dictSize, &cp);
    const char* names[7] = { "window_log", "chain_log", "hash_log", "search_log", "min_match", "target_length", "strategy" };
    const unsigned long vals[7] = { cp.windowLog, cp.chainLog, cp.hashLog, cp.searchLog, cp.minMatch, cp.targetLength, (unsigned long)cp.strategy };
    for (int i = 0; i < 7; i++) {
        if (PyDict_GetItemString(kw, names[i])) continue;
        PyObject* val = PyLong_FromUnsignedLong(vals[i]);
        if (!val || PyDict_SetItemString(kw, names[i], val) != 0) { Py_XDECREF(val); Py_DECREF(kw); return NULL; }
        Py_DECREF(val);
    }
    PyObject* empty = PyTuple_New(0);
    PyObject* r = empty ? PyObject_Call((PyObject*)&CompressionParametersType, empty, kw) : NULL;
    Py_XDECREF(empty); Py_DECREF(kw);
    return r;
}
/* what ZSTD_estimateCCtxSize_usingCCtxParams counts for these strategies: the two index tables + a block of sequences and literals */
static PyObject* cparams_estimated_size(CompressionParameters* self, PyObject* noargs)
{
    (void)noargs;
    zhip_compression_parameters cp;
    Z.get_cparams(self->compressionLevel, 0, 0, &cp);
    const unsigned w = self->windowLog ? (unsigned)self->windowLog : cp.windowLog, h = self->hashLog ? (unsigned)self->hashLog : cp.hashLog,
                   c = self->chainLog ? (unsigned)self->chainLog : cp.chainLog;
    const unsigned long long block = (1ull << w) < (1ull << 17) ? (1ull << w) : (1ull << 17);
    return PyLong_FromUnsignedLongLong((4ull << h) + (4ull << c) + block * 3 + 65536);
}
#define CP_MEMBER(name, field) { name, T_INT, offsetof(CompressionParameters, field), READONLY, name }
static PyMemberDef cparams_members[] = {
    CP_MEMBER("format", format), CP_MEMBER("compression_level", compressionLevel), CP_MEMBER("window_log", windowLog), CP_MEMBER("hash_log", hashLog),
    CP_MEMBER("chain_log", chainLog), CP_MEMBER("search_log", searchLog), CP_MEMBER("min_match", minMatch), CP_MEMBER("target_length", targetLength),
    CP_MEMBER("strategy", strategy), CP_MEMBER("write_content_size", contentSizeFlag), CP_MEMBER("write_checksum", checksumFlag),
    CP_MEMBER("write_dict_id", dictIDFlag), CP_MEMBER("job_size", jobSize), CP_MEMBER("overlap_log", overlapLog), CP_MEMBER("force_max_window", forceMaxWindow),
    CP_MEMBER("enable_ldm", enableLDM), CP_MEMBER("ldm_hash_log", ldmHashLog), CP_MEMBER("ldm_min_match", ldmMinMatch),
    CP_MEMBER("ldm_bucket_size_log", ldmBucketSizeLog), CP_MEMBER("ldm_hash_rate_log", ldmHashRateLog), CP_MEMBER("threads", threads),
    { NULL, 0, 0, 0, NULL } };
static PyMethodDef cparams_methods[] = {
    { "from_level", (PyCFunction)cparams_from_level, METH_VARARGS | METH_KEYWORDS | METH_STATIC, "parameters libzstd derives from a level (and size hints)" },
    { "estimated_compression_context_size", (PyCFunction)cparams_estimated_size, METH_NOARGS, "bytes of working memory a frame of these parameters needs" },
    { NULL, NULL, 0, NULL } };

/* ------------------------------------------------------------------------------------------ ZstdCompressor */
typedef struct { PyObject_HEAD int level; int writeChecksum, writeContentSize, writeDictID; int format, threads; zhip_compression_parameters cp; PyObject* dict;
                 int dictPre; zhip_compression_parameters dictPreCp;      /* the dictionary's precomputed state when this compressor was made: .compress() keeps it (setup_cctx runs once, compressor.c:13-45,241) */
} Compressor;
static PyTypeObject CompressorType = { PyVarObject_HEAD_INIT(NULL, 0) };
static void comp_dealloc(Compressor* self) { Py_CLEAR(self->dict); Py_TYPE(self)->tp_free((PyObject*)self); }
static int comp_init(Compressor* self, PyObject* args, PyObject* kwargs)
{
    static char* kwlist[] = { "level", "dict_data", "compression_params", "write_checksum", "write_content_size", "write_dict_id", "threads", NULL };
    int level = 3, threads = 0;
    PyObject *dict = NULL, *params = NULL, *writeChecksum = NULL, *writeContentSize = NULL, *writeDictID = NULL;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "|iOOOOOi:ZstdCompressor", kwlist, &level, &dict, &params, &writeChecksum, &writeContentSize, &writeDictID, &threads)) return -1;
    if (level > MAX_COMPRESSION_LEVEL) { PyErr_Format(PyExc_ValueError, "level must be less than %d", MAX_COMPRESSION_LEVEL + 1); return -1; }
    if (writeChecksum == Py_None) writeChecksum = NULL;
    if (writeContentSize == Py_None) writeContentSize = NULL;
    if (writeDictID == Py_None) writeDictID = NULL;
    memset(&self->cp, 0, sizeof self->cp); self->format = FORMAT_ZSTD1; self->threads = 0;
    if (params == Py_None) params = NULL;
    if (params) {
        if (!PyObject_TypeCheck(params, &CompressionParametersType)) { PyErr_SetString(PyExc_TypeError, "compression_params must be zstd.ZstdCompressionParameters"); return -1; }
        /* the reference's mutual-exclusion checks (compressor.c:177-200) */
        if (writeChecksum) { PyErr_SetString(PyExc_ValueError, "cannot define compression_params and write_checksum"); return -1; }
        if (writeContentSize) { PyErr_SetString(PyExc_ValueError, "cannot define compression_params and write_content_size"); return -1; }
        if (writeDictID) { PyErr_SetString(PyExc_ValueError, "cannot define compression_params and write_dict_id"); return -1; }
        if (threads) { PyErr_SetString(PyExc_ValueError, "cannot define compression_params and threads"); return -1; }
    }
    if (dict == Py_None) dict = NULL;
    if (dict && !PyObject_TypeCheck(dict, &CompressionDictType)) { PyErr_SetString(PyExc_TypeError, "dict_data must be a ZstdCompressionDict"); return -1; }
    self->level = level;
    self->writeChecksum = writeChecksum ? PyObject_IsTrue(writeChecksum) : 0;
    self->writeContentSize = writeContentSize ? PyObject_IsTrue(writeContentSize) : 1;
    self->writeDictID = writeDictID ? PyObject_IsTrue(writeDictID) : 1;
    if (threads < 0) { long n = sysconf(_SC_NPROCESSORS_ONLN); threads = n > 0 ? (int)n : 1; }
    self->threads = threads;
    if (params) {       /* set_parameters (compressionparams.c:43-68): the object's values replace the constructor's */
        const CompressionParameters* q = (const CompressionParameters*)params;
        if (q->enableLDM || q->forceMaxWindow) { PyErr_SetString(ZstdError, "long distance matching / force_max_window are not supported by the HIP backend"); return -1; }
        self->level = q->compressionLevel; self->format = q->format; self->threads = q->threads;
        self->writeChecksum = q->checksumFlag; self->writeContentSize = q->contentSizeFlag; self->writeDictID = q->dictIDFlag;
        self->cp.windowLog = (uint32_t)q->windowLog; self->cp.chainLog = (uint32_t)q->chainLog; self->cp.hashLog = (uint32_t)q->hashLog;
        self->cp.searchLog = (uint32_t)q->searchLog; self->cp.minMatch = (uint32_t)q->minMatch; self->cp.targetLength = (uint32_t)q->targetLength;
        self->cp.strategy = q->strategy;
    }
    Py_XINCREF(dict); Py_XSETREF(self->dict, dict);
    self->dictPre = dict ? ((CompressionDict*)dict)->precomputed : 0;
    if (dict) self->dictPreCp = ((CompressionDict*)dict)->pre; else memset(&self->dictPreCp, 0, sizeof self->dictPreCp);
    return 0;
}
/* A precomputed dictionary carries its own parameters: libzstd compresses with the CDict's (ZSTD_CCtx_refCDict; the CDict's level is
 * "none", so ZSTD_compressBegin_internal zstd.c:28230 always takes ZSTD_resetCCtx_usingCDict) and takes only the frame's window log from
 * the context -- the level-3 row unless the compressor set one explicitly (ZSTD_CCtx_init_compressStream2, zstd.c:29329). Expressed
 * through the ABI: level 3 + the dictionary's six non-window fields as explicit parameters. */
static void apply_precomputed(zhip_cparams* p, const zhip_compression_parameters* pre, uint32_t windowLog)
{
    p->level = 3; p->cp = *pre; p->cp.windowLog = windowLog;
}
static void comp_params(Compressor* self, zhip_cparams* p, int oneShot)
{
    memset(p, 0, sizeof *p);
    p->level = self->level; p->contentSizeFlag = self->writeContentSize; p->checksumFlag = self->writeChecksum; p->dictIDFlag = self->writeDictID;
    p->format = self->format; p->cp = self->cp;
    if (self->dict && PyBytes_GET_SIZE(((CompressionDict*)self->dict)->data)) {
        CompressionDict* d = (CompressionDict*)self->dict;
        p->dict = PyBytes_AS_STRING(d->data); p->dictSize = (size_t)PyBytes_GET_SIZE(d->data);
        p->dictType = d->dictType;
        /* .compress() uses the context set up when the compressor was made; multi_compress_to_buffer() sets its worker contexts up per
         * call and sees a CDict precomputed since (compressor.c:1147) */
        if (oneShot ? self->dictPre : d->precomputed) apply_precomputed(p, oneShot ? &self->dictPreCp : &d->pre, self->cp.windowLog);
    }
}
/* ZstdCompressionDict.precompute_compress (compressiondict.c:228-286): the parameters come from ZSTD_getCParams(level, 0, dictSize) or from
 * a ZstdCompressionParameters object; ZSTD_createCDict_advanced then adjusts them for "a dictionary of this size, source unknown"
 * (ZSTD_adjustCParams_internal in ZSTD_cpm_createCDict mode, zstd.c:24426: the table logs are clamped against the window log GIVEN HERE).
 * The device repeats that adjustment with the frame's window row, which is never smaller than min(given window, log2(dict + 513)), so
 * the clamp is applied here and the six non-window fields travel as explicit parameters. The dictionary is digested on the device
 * right away (an empty batch runs zhip_ctx_set_cparams): what libzstd refuses here ("unable to precompute dictionary") is refused here. */
static uint32_t hb32(uint64_t v) { uint32_t r = 0; while (v >>= 1) r++; return r; }
static PyObject* dict_precompute(CompressionDict* self, PyObject* args, PyObject* kwargs)
{
    static char* kwlist[] = { "level", "compression_params", NULL };
    int level = 0; PyObject* params = NULL;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "|iO!:precompute_compress", kwlist, &level, &CompressionParametersType, &params)) return NULL;
    if (level && params) { PyErr_SetString(PyExc_ValueError, "must only specify one of level or compression_params"); return NULL; }
    if (!level && !params) { PyErr_SetString(PyExc_ValueError, "must specify one of level or compression_params"); return NULL; }
    self->precomputed = 0;                                                 /* the reference frees its old CDict first (compressiondict.c:256-264): a failed call leaves none */
    const size_t dictSize = (size_t)PyBytes_GET_SIZE(self->data);
    zhip_compression_parameters pre, row;
    if (level) Z.get_cparams(level, 0, dictSize, &pre);
    else {
        const CompressionParameters* q = (const CompressionParameters*)params;
        Z.get_cparams(3, 0, dictSize, &row);                               /* unset fields: the default level's row (the CDict's level is "none") */
        pre.windowLog = q->windowLog ? (uint32_t)q->windowLog : row.windowLog; pre.chainLog = q->chainLog ? (uint32_t)q->chainLog : row.chainLog;
        pre.hashLog = q->hashLog ? (uint32_t)q->hashLog : row.hashLog; pre.searchLog = q->searchLog ? (uint32_t)q->searchLog : row.searchLog;
        pre.minMatch = q->minMatch ? (uint32_t)q->minMatch : row.minMatch; pre.targetLength = q->targetLength ? (uint32_t)q->targetLength : row.targetLength;
        pre.strategy = q->strategy > 0 ? q->strategy : row.strategy;
    }
    if (pre.strategy != 1 && pre.strategy != 2) {
        PyErr_SetString(ZstdError, "unable to precompute dictionary: strategies above double-fast (levels >= 5) are not implemented by the HIP backend");
        return NULL;
    }
    if (dictSize) {                                                        /* the createCDict-mode clamp, against the window given here */
        const uint64_t tSize = 513 + (uint64_t)dictSize;
        /* a window that cannot hold the dictionary stays with the CDict and shrinks every frame's working tables
         * (ZSTD_resetCCtx_byAttachingCDict, zstd.c:25299): not carried through the ABI -- refused, never encoded differently */
        if (pre.windowLog < 31 && ((uint64_t)1 << pre.windowLog) < tSize) {
            PyErr_SetString(ZstdError, "unable to precompute dictionary: a window_log smaller than the dictionary is not supported by the HIP backend");
            return NULL;
        }
        const uint32_t srcLog = tSize < 64 ? 6 : hb32(tSize - 1) + 1;
        uint32_t w = pre.windowLog > srcLog ? srcLog : pre.windowLog;
        const uint64_t windowSize = (uint64_t)1 << w;
        const uint32_t dw = windowSize >= tSize ? w : (dictSize + windowSize >= ((uint64_t)1 << 31) ? 31 : hb32(dictSize + windowSize - 1) + 1);
        if (pre.hashLog > dw + 1) pre.hashLog = dw + 1;
        if (pre.chainLog > dw) pre.chainLog = dw;
    }
    pre.windowLog = 0;
    zhip_cparams p; memset(&p, 0, sizeof p);
    p.contentSizeFlag = 1; p.dictIDFlag = 1; p.dict = PyBytes_AS_STRING(self->data); p.dictSize = dictSize; p.dictType = self->dictType;
    apply_precomputed(&p, &pre, 0);
    zhip_outbuf* out = NULL; size_t nOut = 0; zhip_error err; int rc;
    memset(&err, 0, sizeof err);
    Py_BEGIN_ALLOW_THREADS
    rc = Z.compress_batch(&p, NULL, 0, &out, &nOut, &err);
    Py_END_ALLOW_THREADS
    if (rc == ZHIP_ERR_NONE) Z.free_outbufs(out, nOut, 1);
    else if (rc == ZHIP_ERR_ZSTD) { PyErr_Format(ZstdError, "unable to precompute dictionary: %s", Z.error_name(err.zstdErr)); return NULL; }
    else { PyErr_Format(ZstdError, "unable to precompute dictionary: %s", Z.last_error()); return NULL; }
    self->pre = pre; self->precomputed = 1;
    Py_RETURN_NONE;
}
static void comp_raise(int rc, const zhip_error* err, int oneShot)
{
    if (rc == ZHIP_ERR_ZSTD && (err->zstdErr == 30 || err->zstdErr == 32))          /* dictionary_corrupted / dictionary_wrong: compressor.c:44-52 */
        PyErr_Format(ZstdError, "could not load compression dictionary: %s", Z.error_name(err->zstdErr));
    else if (rc == ZHIP_ERR_ZSTD && err->zstdErr == 42) PyErr_Format(ZstdError, "could not set compression parameters: %s", Z.error_name(err->zstdErr));
    else if (rc == ZHIP_ERR_ZSTD) {
        if (oneShot) PyErr_Format(ZstdError, "cannot compress: %s", Z.error_name(err->zstdErr));
        else PyErr_Format(ZstdError, "error compressing item %zd: %s", (Py_ssize_t)err->index, Z.error_name(err->zstdErr));
    } else if (rc == ZHIP_ERR_NO_MEMORY) PyErr_NoMemory();
    else if (rc == ZHIP_ERR_SIZE_MISMATCH) PyErr_Format(ZstdError, "error compressing item %zd: not enough space in output", (Py_ssize_t)err->index);
    else PyErr_Format(ZstdError, "HIP backend failure: %s", Z.last_error());
}
static PyObject* comp_compress(Compressor* self, PyObject* args, PyObject* kwargs)
{
    static char* kwlist[] = { "data", NULL };
    Py_buffer src;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "y*:compress", kwlist, &src)) return NULL;
    zhip_item item = { src.buf, (size_t)src.len, 0 };
    zhip_cparams p; comp_params(self, &p, 1);
    zhip_outbuf* out = NULL; size_t nOut = 0; zhip_error err; int rc;
    memset(&err, 0, sizeof err);
    Py_BEGIN_ALLOW_THREADS
    rc = Z.compress_batch(&p, &item, 1, &out, &nOut, &err);
    Py_END_ALLOW_THREADS
    PyBuffer_Release(&src);
    if (rc != ZHIP_ERR_NONE) { comp_raise(rc, &err, 1); return NULL; }
    PyObject* r = PyBytes_FromStringAndSize((const char*)out[0].data + out[0].segs[0].offset, (Py_ssize_t)out[0].segs[0].length);
    Z.free_outbufs(out, nOut, 1);
    return r;
}
static PyObject* comp_multi(Compressor* self, PyObject* args, PyObject* kwargs)
{
    static char* kwlist[] = { "data", "threads", NULL };
    PyObject* data; int threads = 0;    /* accepted for API compatibility (compressor.c:1361-1367); the GPU does the fan-out */
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "O|i:multi_compress_to_buffer", kwlist, &data, &threads)) return NULL;
    Sources s;
    if (sources_collect(&s, data, "argument must be list of BufferWithSegments") != 0) return NULL;
    if (s.n == 0) { sources_free(&s); PyErr_SetString(PyExc_ValueError, "no source elements found"); return NULL; }
    if (s.totalSize == 0) { sources_free(&s); PyErr_SetString(PyExc_ValueError, "source elements are empty"); return NULL; }
    zhip_cparams p; comp_params(self, &p, 0);
    zhip_outbuf* out = NULL; size_t nOut = 0; zhip_error err; int rc;
    memset(&err, 0, sizeof err);
    Py_BEGIN_ALLOW_THREADS
    rc = Z.compress_batch(&p, s.items, (size_t)s.n, &out, &nOut, &err);
    Py_END_ALLOW_THREADS
    sources_free(&s);
    if (rc != ZHIP_ERR_NONE) { comp_raise(rc, &err, 0); return NULL; }
    return collection_from_outbufs(out, nOut);
}
/* the reference reports the libzstd context's size (compressor.c:263, decompressor.c:128); here the context lives on the device: the bytes
 * the calling thread's device contexts hold */
static PyObject* zero_memory_size(PyObject* self, PyObject* noargs) { (void)self; (void)noargs; return PyLong_FromSize_t(Z.thread_memory_size()); }
static PyMethodDef comp_methods[] = {
    { "compress", (PyCFunction)comp_compress, METH_VARARGS | METH_KEYWORDS, "compress(data) -> bytes" },
    { "multi_compress_to_buffer", (PyCFunction)comp_multi, METH_VARARGS | METH_KEYWORDS, "compress many inputs into a BufferWithSegmentsCollection" },
    { "memory_size", (PyCFunction)zero_memory_size, METH_NOARGS, "device memory held by the calling thread's contexts" },
    { NULL, NULL, 0, NULL } };

/* ------------------------------------------------------------------------------------------ ZstdDecompressor */
typedef struct { PyObject_HEAD PyObject* dict; unsigned long long maxWindowSize; int format; } Decompressor;
static PyTypeObject DecompressorType = { PyVarObject_HEAD_INIT(NULL, 0) };
static void decomp_dealloc(Decompressor* self) { Py_CLEAR(self->dict); Py_TYPE(self)->tp_free((PyObject*)self); }
static int decomp_init(Decompressor* self, PyObject* args, PyObject* kwargs)
{
    static char* kwlist[] = { "dict_data", "max_window_size", "format", NULL };
    PyObject* dict = NULL; unsigned long long maxWindow = 0; int format = FORMAT_ZSTD1;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "|OKi:ZstdDecompressor", kwlist, &dict, &maxWindow, &format)) return -1;
    if (dict == Py_None) dict = NULL;
    if (dict && !PyObject_TypeCheck(dict, &CompressionDictType)) { PyErr_SetString(PyExc_TypeError, "dict_data must be a ZstdCompressionDict"); return -1; }
    if (format != FORMAT_ZSTD1 && format != FORMAT_ZSTD1_MAGICLESS) { PyErr_SetString(ZstdError, "unable to set decoding format: Parameter is out of bound"); return -1; }
    Py_XINCREF(dict); Py_XSETREF(self->dict, dict);
    self->maxWindowSize = maxWindow; self->format = format;
    return 0;
}
static void decomp_params(Decompressor* self, zhip_dparams* p)
{
    memset(p, 0, sizeof *p);
    if (self->dict && PyBytes_GET_SIZE(((CompressionDict*)self->dict)->data)) {
        p->dict = PyBytes_AS_STRING(((CompressionDict*)self->dict)->data); p->dictSize = (size_t)PyBytes_GET_SIZE(((CompressionDict*)self->dict)->data);
        p->dictType = ((CompressionDict*)self->dict)->dictType;
    }
    p->maxWindowSize = self->maxWindowSize; p->format = self->format;
}
static PyObject* decomp_decompress(Decompressor* self, PyObject* args, PyObject* kwargs)
{
    static char* kwlist[] = { "data", "max_output_size", "read_across_frames", "allow_extra_data", NULL };
    Py_buffer src; unsigned long long maxOutput = 0; int readAcross = 0, allowExtra = 1;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "y*|Kpp:decompress", kwlist, &src, &maxOutput, &readAcross, &allowExtra)) return NULL;
    PyObject* result = NULL;
    if (readAcross) { PyErr_SetString(ZstdError, "ZstdDecompressor.read_across_frames=True is not yet implemented"); goto done; }
    {
        const uint64_t fcs = Z.frame_content_size_format(src.len ? src.buf : NULL, (size_t)src.len, self->format);
        if (fcs == ZHIP_CONTENTSIZE_ERROR) { PyErr_SetString(ZstdError, "error determining content size from frame header"); goto done; }
        if (fcs == 0) { result = PyBytes_FromStringAndSize("", 0); goto done; }
        int flags = 0; uint64_t cap, expected;
        if (fcs == ZHIP_CONTENTSIZE_UNKNOWN) {
            if (maxOutput == 0) { PyErr_SetString(ZstdError, "could not determine content size in frame header"); goto done; }
            if (maxOutput > (1ull << 48)) { PyErr_NoMemory(); goto done; }
            cap = maxOutput; expected = 0; flags = FLAG_ALLOW_SHORT;
        } else { cap = fcs; expected = fcs; }
        zhip_item item = { src.buf, (size_t)src.len, (size_t)cap };
        zhip_dparams p; decomp_params(self, &p);
        zhip_outbuf* out = NULL; size_t nOut = 0; zhip_error err; int rc;
        memset(&err, 0, sizeof err);
        Py_BEGIN_ALLOW_THREADS
        rc = Z.decompress_batch(&p, &item, 1, flags, &out, &nOut, &err);
        Py_END_ALLOW_THREADS
        if (rc == ZHIP_ERR_ZSTD && err.zstdErr == 30 && p.dict) { PyErr_SetString(ZstdError, "could not create decompression dict"); goto done; }   /* compressiondict.c:155-159 */
        if (rc == ZHIP_ERR_ZSTD) {
            if (flags && err.zstdErr == 70) PyErr_SetString(ZstdError, "decompression error: did not decompress full frame");
            else PyErr_Format(ZstdError, "decompression error: %s", Z.error_name(err.zstdErr));
            goto done;
        }
        if (rc == ZHIP_ERR_SIZE_MISMATCH) { PyErr_Format(ZstdError, "decompression error: decompressed %llu bytes; expected %llu", (unsigned long long)err.detail[0], (unsigned long long)expected); goto done; }
        if (rc == ZHIP_ERR_NO_MEMORY) { PyErr_NoMemory(); goto done; }
        if (rc != ZHIP_ERR_NONE) { PyErr_Format(ZstdError, "HIP backend failure: %s", Z.last_error()); goto done; }
        result = PyBytes_FromStringAndSize((const char*)out[0].data, (Py_ssize_t)out[0].segs[0].length);
        Z.free_outbufs(out, nOut, 1);
        if (result && !allowExtra) {
            const int64_t used = Z.find_frame_compressed_size_format(src.buf, (size_t)src.len, self->format);
            if (used >= 0 && used < (int64_t)src.len) {
                Py_CLEAR(result);
                PyErr_Format(ZstdError, "compressed input contains %zd bytes of unused data, which is disallowed", (Py_ssize_t)(src.len - used));
            }
        }
    }
done:
    PyBuffer_Release(&src);
    return result;
}
static PyObject* decomp_multi(Decompressor* self, PyObject* args, PyObject* kwargs)
{
    static char* kwlist[] = { "frames", "decompressed_sizes", "threads", NULL };
    PyObject *frames, *sizesObj = NULL; int threads = 0;
    Py_buffer sizes; memset(&sizes, 0, sizeof sizes);
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "O|Oi:multi_decompress_to_buffer", kwlist, &frames, &sizesObj, &threads)) return NULL;
    if (sizesObj == Py_None) sizesObj = NULL;
    if (sizesObj && PyObject_GetBuffer(sizesObj, &sizes, PyBUF_CONTIG_RO) != 0) return NULL;
    Sources s; PyObject* result = NULL;
    if (sources_collect(&s, frames, "argument must be list or BufferWithSegments") != 0) goto done2;
    if (sizesObj) {
        if (sizes.len != s.n * 8) {
            PyErr_Format(PyExc_ValueError, "decompressed_sizes size mismatch; expected %zd, got %zd", s.n * 8, sizes.len);
            goto done;
        }
        for (Py_ssize_t i = 0; i < s.n; i++) { uint64_t v; memcpy(&v, (const char*)sizes.buf + 8 * i, 8); s.items[i].dstSize = (size_t)v; }
    }
    if (s.n == 0) { PyErr_SetString(PyExc_ValueError, "no source elements found"); goto done; }
    {
        zhip_dparams p; decomp_params(self, &p);
        zhip_outbuf* out = NULL; size_t nOut = 0; zhip_error err; int rc;
        memset(&err, 0, sizeof err);
        Py_BEGIN_ALLOW_THREADS
        rc = Z.decompress_batch(&p, s.items, (size_t)s.n, sizesObj ? 1 : 0, &out, &nOut, &err);
        Py_END_ALLOW_THREADS
        if (rc == ZHIP_ERR_UNKNOWN_SIZE) PyErr_Format(PyExc_ValueError, "could not determine decompressed size of item %zd", (Py_ssize_t)err.index);
        else if (rc == ZHIP_ERR_ZSTD && err.zstdErr == 30 && p.dict) PyErr_SetString(ZstdError, "could not create decompression dict");
        else if (rc == ZHIP_ERR_ZSTD) PyErr_Format(ZstdError, "error decompressing item %zd: %s", (Py_ssize_t)err.index, Z.error_name(err.zstdErr));
        else if (rc == ZHIP_ERR_SIZE_MISMATCH)
            PyErr_Format(ZstdError, "error decompressing item %zd: decompressed %llu bytes; expected %llu", (Py_ssize_t)err.index,
                         (unsigned long long)err.detail[0], (unsigned long long)err.detail[1]);
        else if (rc == ZHIP_ERR_NO_MEMORY) PyErr_NoMemory();
        else if (rc != ZHIP_ERR_NONE) PyErr_Format(ZstdError, "HIP backend failure: %s", Z.last_error());
        else result = collection_from_outbufs(out, nOut);
    }
done:
    sources_free(&s);
done2:
    if (sizes.obj) PyBuffer_Release(&sizes);
    return result;
}
/* decompress_content_dict_chain (c-ext/decompressor.c:620-890): frame 0 stands alone, every later frame is decompressed with the previous
 * fulltext as a raw-content prefix dictionary (ZSTD_DCtx_refPrefix_advanced(..., ZSTD_dct_rawContent)). The chain is serial by nature;
 * each link is one call into the batch ABI with that fulltext as the dictionary. Same checks, same messages. */
static PyObject* decomp_content_dict_chain(Decompressor* self, PyObject* args, PyObject* kwargs)
{
    static char* kwlist[] = { "frames", NULL };
    PyObject* chunks;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "O!:decompress_content_dict_chain", kwlist, &PyList_Type, &chunks)) return NULL;
    const Py_ssize_t n = PyList_Size(chunks);
    if (!n) { PyErr_SetString(PyExc_ValueError, "empty input chain"); return NULL; }
    PyObject* prev = NULL;                          /* the previous fulltext (bytes) */
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* chunk = PyList_GetItem(chunks, i);
        if (!PyBytes_Check(chunk)) { PyErr_Format(PyExc_ValueError, "chunk %zd must be bytes", i); Py_XDECREF(prev); return NULL; }
        char* data; Py_ssize_t size;
        PyBytes_AsStringAndSize(chunk, &data, &size);
        const uint64_t fcs = Z.frame_content_size_format(size ? data : NULL, (size_t)size, ZHIP_FORMAT_ZSTD1);
        if (fcs == ZHIP_CONTENTSIZE_ERROR) {
            /* ZSTD_getFrameHeader: an error for a bad magic / descriptor, "needs more input" for a truncated header */
            const int tooSmall = size < 5 || (size >= 4 && memcmp(data, "\x28\xb5\x2f\xfd", 4) == 0);
            if (tooSmall && size < 18) PyErr_Format(PyExc_ValueError, "chunk %zd is too small to contain a zstd frame", i);
            else PyErr_Format(PyExc_ValueError, "chunk %zd is not a valid zstd frame", i);
            Py_XDECREF(prev); return NULL;
        }
        if (fcs == ZHIP_CONTENTSIZE_UNKNOWN) { PyErr_Format(PyExc_ValueError, "chunk %zd missing content size in frame", i); Py_XDECREF(prev); return NULL; }
        if (fcs > (uint64_t)PY_SSIZE_T_MAX) { PyErr_Format(PyExc_ValueError, "chunk %zd is too large to decompress on this platform", i); Py_XDECREF(prev); return NULL; }
        PyObject* cur = NULL;
        if (fcs == 0) cur = PyBytes_FromStringAndSize("", 0);
        else {
            zhip_item item = { data, (size_t)size, (size_t)fcs };
            zhip_dparams p; memset(&p, 0, sizeof p);
            p.maxWindowSize = self->maxWindowSize; p.format = ZHIP_FORMAT_ZSTD1;
            if (prev && PyBytes_GET_SIZE(prev)) { p.dict = PyBytes_AS_STRING(prev); p.dictSize = (size_t)PyBytes_GET_SIZE(prev); p.dictType = ZHIP_DICT_RAWCONTENT; }
            zhip_outbuf* out = NULL; size_t nOut = 0; zhip_error err; int rc;
            memset(&err, 0, sizeof err);
            Py_BEGIN_ALLOW_THREADS
            rc = Z.decompress_batch(&p, &item, 1, 0, &out, &nOut, &err);
            Py_END_ALLOW_THREADS
            /* a frame that ends early is, to the reference's streaming call, a frame that wants more input: "did not decompress full frame" */
            if (rc == ZHIP_ERR_ZSTD && err.zstdErr == 72) PyErr_Format(ZstdError, "chunk %zd did not decompress full frame", i);
            else if (rc == ZHIP_ERR_ZSTD) PyErr_Format(ZstdError, "could not decompress chunk %zd: %s", i, Z.error_name(err.zstdErr));
            else if (rc == ZHIP_ERR_SIZE_MISMATCH) PyErr_Format(ZstdError, "chunk %zd did not decompress full frame", i);
            else if (rc == ZHIP_ERR_NO_MEMORY) PyErr_NoMemory();
            else if (rc != ZHIP_ERR_NONE) PyErr_Format(ZstdError, "HIP backend failure: %s", Z.last_error());
            else { cur = PyBytes_FromStringAndSize((const char*)out[0].data, (Py_ssize_t)out[0].segs[0].length); Z.free_outbufs(out, nOut, 1); }
        }
        Py_XDECREF(prev);
        if (!cur) return NULL;
        prev = cur;
    }
    return prev;
}
static PyMethodDef decomp_methods[] = {
    { "decompress", (PyCFunction)decomp_decompress, METH_VARARGS | METH_KEYWORDS, "decompress(data) -> bytes" },
    { "decompress_content_dict_chain", (PyCFunction)decomp_content_dict_chain, METH_VARARGS | METH_KEYWORDS, "decompress a chain of frames, each using the previous fulltext as its dictionary" },
    { "multi_decompress_to_buffer", (PyCFunction)decomp_multi, METH_VARARGS | METH_KEYWORDS, "decompress many frames into a BufferWithSegmentsCollection" },
    { "memory_size", (PyCFunction)zero_memory_size, METH_NOARGS, "device memory held by the calling thread's contexts" },
    { NULL, NULL, 0, NULL } };

/* ------------------------------------------------------------------------------------------ FrameParameters (c-ext/frameparams.c) */
typedef struct { PyObject_HEAD unsigned long long contentSize, windowSize; unsigned dictID; char checksumFlag; } FrameParameters;
static PyTypeObject FrameParametersType = { PyVarObject_HEAD_INIT(NULL, 0) };
static void fp_dealloc(FrameParameters* self) { Py_TYPE(self)->tp_free((PyObject*)self); }
static PyMemberDef fp_members[] = {
    { "content_size", T_ULONGLONG, offsetof(FrameParameters, contentSize), READONLY, "frame content size" },
    { "window_size", T_ULONGLONG, offsetof(FrameParameters, windowSize), READONLY, "window size" },
    { "dict_id", T_UINT, offsetof(FrameParameters, dictID), READONLY, "dictionary ID" },
    { "has_checksum", T_BOOL, offsetof(FrameParameters, checksumFlag), READONLY, "checksum flag" },
    { NULL, 0, 0, 0, NULL } };

/* frame header parse on the host (RFC 8878 3.1.1.1; ZSTD_getFrameHeader_advanced zstd.c:43668) */
static PyObject* mod_get_frame_parameters(PyObject* self, PyObject* args, PyObject* kwargs)
{
    (void)self;
    static char* kwlist[] = { "data", "format", NULL };
    Py_buffer src; unsigned format = FORMAT_ZSTD1;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "y*|I:get_frame_parameters", kwlist, &src, &format)) return NULL;
    const unsigned char* p = (const unsigned char*)src.buf; const size_t n = (size_t)src.len;
    PyObject* result = NULL;
    if (format != FORMAT_ZSTD1 && format != FORMAT_ZSTD1_MAGICLESS) { PyErr_SetString(ZstdError, "cannot get frame parameters: Parameter is out of bound"); goto done; }
    const size_t mg = format == FORMAT_ZSTD1_MAGICLESS ? 0 : 4;
    if (n < mg + 1) { PyErr_Format(ZstdError, "not enough data for frame parameters; need %zu bytes", mg + 1); goto done; }
    if (mg && !(p[0] == 0x28 && p[1] == 0xB5 && p[2] == 0x2F && p[3] == 0xFD)) { PyErr_SetString(ZstdError, "cannot get frame parameters: Unknown frame descriptor"); goto done; }
    {
        const unsigned fhd = p[mg], fcsCode = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, didCode = fhd & 3;
        if (fhd & 8) { PyErr_SetString(ZstdError, "cannot get frame parameters: Unsupported frame parameter"); goto done; }
        static const size_t didSizes[4] = { 0, 1, 2, 4 };
        const size_t didSize = didSizes[didCode], fcsSize = fcsCode == 0 ? (single ? 1 : 0) : fcsCode == 1 ? 2 : fcsCode == 2 ? 4 : 8;
        const size_t need = mg + 1 + (single ? 0 : 1) + didSize + fcsSize;
        if (n < need) { PyErr_Format(ZstdError, "not enough data for frame parameters; need %zu bytes", need); goto done; }
        size_t pos = mg + 1; unsigned long long window = 0, content = ZHIP_CONTENTSIZE_UNKNOWN; unsigned dictID = 0;
        if (!single) {
            const unsigned b = p[pos++], wlog = 10 + (b >> 3);
            if (wlog > 31) { PyErr_SetString(ZstdError, "cannot get frame parameters: Frame requires too much memory for decoding"); goto done; }
            window = (1ull << wlog) + ((1ull << wlog) >> 3) * (b & 7);
        }
        for (size_t i = 0; i < didSize; i++) dictID |= (unsigned)p[pos + i] << (8 * i);
        pos += didSize;
        if (fcsSize) { content = 0; for (size_t i = 0; i < fcsSize; i++) content |= (unsigned long long)p[pos + i] << (8 * i); if (fcsSize == 2) content += 256; }
        if (single) window = content;
        FrameParameters* r = (FrameParameters*)FrameParametersType.tp_alloc(&FrameParametersType, 0);
        if (r) { r->contentSize = content; r->windowSize = window; r->dictID = dictID; r->checksumFlag = (char)checksum; }
        result = (PyObject*)r;
    }
done:
    PyBuffer_Release(&src);
    return result;
}

/* ------------------------------------------------------------------------------------------ module */
static PyObject* mod_frame_content_size(PyObject* self, PyObject* args)
{
    (void)self;
    Py_buffer src;
    if (!PyArg_ParseTuple(args, "y*:frame_content_size", &src)) return NULL;
    const uint64_t v = Z.frame_content_size(src.len ? src.buf : NULL, (size_t)src.len);
    PyBuffer_Release(&src);
    if (v == ZHIP_CONTENTSIZE_ERROR) { PyErr_SetString(ZstdError, "error when determining content size"); return NULL; }
    if (v == ZHIP_CONTENTSIZE_UNKNOWN) return PyLong_FromLong(-1);
    return PyLong_FromUnsignedLongLong(v);
}
/* frame_header_size (c-ext/backend_c.c:77-104 -> ZSTD_frameHeaderSize, zstd.c:43625): the size the frame header descriptor byte announces;
 * the magic number is not checked, only that the five bytes up to the descriptor are there */
static PyObject* mod_frame_header_size(PyObject* self, PyObject* args, PyObject* kwargs)
{
    (void)self;
    static char* kwlist[] = { "source", NULL };
    Py_buffer src;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "y*:frame_header_size", kwlist, &src)) return NULL;
    PyObject* r = NULL;
    if (src.len < 5) PyErr_Format(ZstdError, "could not determine frame header size: %s", Z.error_name(72));       /* srcSize_wrong */
    else {
        static const size_t didSize[4] = { 0, 1, 2, 4 }, fcsSize[4] = { 0, 2, 4, 8 };
        const unsigned fhd = ((const unsigned char*)src.buf)[4];
        const unsigned single = (fhd >> 5) & 1, fcsId = fhd >> 6;
        r = PyLong_FromSize_t(5 + !single + didSize[fhd & 3] + fcsSize[fcsId] + (single && !fcsId));
    }
    PyBuffer_Release(&src);
    return r;
}
static PyMethodDef module_methods[] = {
    { "frame_content_size", mod_frame_content_size, METH_VARARGS, "content size of a frame, -1 if unknown" },
    { "frame_header_size", (PyCFunction)mod_frame_header_size, METH_VARARGS | METH_KEYWORDS, "size of a frame's header" },
    { "get_frame_parameters", (PyCFunction)mod_get_frame_parameters, METH_VARARGS | METH_KEYWORDS, "parse a frame header" },
    { NULL, NULL, 0, NULL } };
static struct PyModuleDef moduledef = { PyModuleDef_HEAD_INIT, "backend_hip", "python-zstandard hot path on MI355X: CPython extension over libzstd_hip.so", -1, module_methods, 0, 0, 0, 0 };

static int no_direct_init(PyObject* self, PyObject* args, PyObject* kwargs)
{
    (void)args; (void)kwargs;
    const char* name = Py_TYPE(self)->tp_name;
    const char* dot = strrchr(name, '.');
    PyErr_Format(PyExc_TypeError, "cannot create '%s' instances directly", dot ? dot + 1 : name);
    return -1;
}

#define READY(T, NAME, SIZE, DEALLOC, INIT, DOC) do { T.tp_name = "backend_hip." NAME; T.tp_basicsize = SIZE; T.tp_dealloc = (destructor)DEALLOC; \
        T.tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE; T.tp_new = PyType_GenericNew; T.tp_init = (initproc)INIT; T.tp_doc = DOC; } while (0)

PyMODINIT_FUNC PyInit_backend_hip(void)
{
    PyObject* m = PyModule_Create(&moduledef);
    if (!m) return NULL;
    if (bind_library(m) != 0) { Py_DECREF(m); return NULL; }
    ZstdError = PyErr_NewException("backend_hip.ZstdError", NULL, NULL);
    if (!ZstdError) { Py_DECREF(m); return NULL; }

    READY(BufferWithSegmentsType, "BufferWithSegments", sizeof(BufferWithSegments), bws_dealloc, bws_init, "a payload and its (offset, length) segments");
    BufferWithSegmentsType.tp_as_sequence = &bws_sq; BufferWithSegmentsType.tp_as_buffer = &bws_as_buffer;
    BufferWithSegmentsType.tp_methods = bws_methods; BufferWithSegmentsType.tp_members = bws_members;
    READY(BufferSegmentType, "BufferSegment", sizeof(BufferSegment), seg_dealloc, no_direct_init, "one segment of a BufferWithSegments");
    BufferSegmentType.tp_as_sequence = &seg_sq; BufferSegmentType.tp_as_buffer = &seg_as_buffer; BufferSegmentType.tp_methods = seg_methods; BufferSegmentType.tp_members = seg_members;
    READY(BufferSegmentsType, "BufferSegments", sizeof(BufferSegments), segs_dealloc, no_direct_init, "the segment table of a BufferWithSegments");
    BufferSegmentsType.tp_as_buffer = &segs_as_buffer; BufferSegmentsType.tp_methods = segs_methods;
    READY(BufferCollectionType, "BufferWithSegmentsCollection", sizeof(BufferCollection), coll_dealloc, coll_init, "virtual concatenation of BufferWithSegments");
    BufferCollectionType.tp_as_sequence = &coll_sq; BufferCollectionType.tp_methods = coll_methods;
    READY(CompressionDictType, "ZstdCompressionDict", sizeof(CompressionDict), dict_dealloc, dict_init, "raw bytes of a dictionary");
    CompressionDictType.tp_as_sequence = &dict_sq; CompressionDictType.tp_methods = dict_methods;
    READY(CompressorType, "ZstdCompressor", sizeof(Compressor), comp_dealloc, comp_init, "batch / one-shot compressor (HIP kernels, frames bit-identical to libzstd 1.5.7)");
    CompressorType.tp_methods = comp_methods;
    READY(DecompressorType, "ZstdDecompressor", sizeof(Decompressor), decomp_dealloc, decomp_init, "batch / one-shot decompressor (HIP kernels)");
    DecompressorType.tp_methods = decomp_methods;

    READY(FrameParametersType, "FrameParameters", sizeof(FrameParameters), fp_dealloc, no_direct_init, "what a frame header says");
    FrameParametersType.tp_members = fp_members;
    READY(CompressionParametersType, "ZstdCompressionParameters", sizeof(CompressionParameters), cparams_dealloc, cparams_init, "explicit compression parameters");
    CompressionParametersType.tp_members = cparams_members; CompressionParametersType.tp_methods = cparams_methods;

    PyTypeObject* types[] = { &BufferWithSegmentsType, &BufferSegmentType, &BufferSegmentsType, &BufferCollectionType, &CompressionDictType, &CompressorType, &DecompressorType, &FrameParametersType, &CompressionParametersType };
    const char* names[] = { "BufferWithSegments", "BufferSegment", "BufferSegments", "BufferWithSegmentsCollection", "ZstdCompressionDict", "ZstdCompressor", "ZstdDecompressor", "FrameParameters", "ZstdCompressionParameters" };
    for (int i = 0; i < 9; i++) {
        if (PyType_Ready(types[i]) < 0) { Py_DECREF(m); return NULL; }
        Py_INCREF(types[i]);
        if (PyModule_AddObject(m, names[i], (PyObject*)types[i]) < 0) { Py_DECREF(m); return NULL; }
    }
    Py_INCREF(ZstdError); PyModule_AddObject(m, "ZstdError", ZstdError);
    PyModule_AddIntConstant(m, "FORMAT_ZSTD1", FORMAT_ZSTD1); PyModule_AddIntConstant(m, "FORMAT_ZSTD1_MAGICLESS", FORMAT_ZSTD1_MAGICLESS);
    PyModule_AddIntConstant(m, "MAX_COMPRESSION_LEVEL", MAX_COMPRESSION_LEVEL);
    PyModule_AddIntConstant(m, "DICT_TYPE_AUTO", DICT_TYPE_AUTO); PyModule_AddIntConstant(m, "DICT_TYPE_RAWCONTENT", DICT_TYPE_RAWCONTENT);
    PyModule_AddIntConstant(m, "DICT_TYPE_FULLDICT", DICT_TYPE_FULLDICT);
    PyModule_AddObject(m, "CONTENTSIZE_UNKNOWN", PyLong_FromUnsignedLongLong(ZHIP_CONTENTSIZE_UNKNOWN));
    PyModule_AddObject(m, "CONTENTSIZE_ERROR", PyLong_FromUnsignedLongLong(ZHIP_CONTENTSIZE_ERROR));
    PyModule_AddObject(m, "MAGIC_NUMBER", PyLong_FromUnsignedLong(0xFD2FB528ul));
    PyModule_AddIntConstant(m, "BLOCKSIZE_MAX", 1 << 17); PyModule_AddIntConstant(m, "BLOCKSIZELOG_MAX", 17);
    PyModule_AddObject(m, "FRAME_HEADER", PyBytes_FromStringAndSize("\x28\xb5\x2f\xfd", 4));      /* c-ext/constants.c:46-48 */
    PyModule_AddIntConstant(m, "SEARCHLENGTH_MIN", 3); PyModule_AddIntConstant(m, "SEARCHLENGTH_MAX", 7);
    PyModule_AddIntConstant(m, "COMPRESSION_RECOMMENDED_INPUT_SIZE", 1 << 17);
    PyModule_AddIntConstant(m, "COMPRESSION_RECOMMENDED_OUTPUT_SIZE", (1 << 17) + 512 + 3 + 4);
    PyModule_AddIntConstant(m, "DECOMPRESSION_RECOMMENDED_INPUT_SIZE", (1 << 17) + 3);
    PyModule_AddIntConstant(m, "DECOMPRESSION_RECOMMENDED_OUTPUT_SIZE", 1 << 17);
    PyModule_AddIntConstant(m, "WINDOWLOG_MIN", 10); PyModule_AddIntConstant(m, "WINDOWLOG_MAX", 31);
    /* parameter bounds and strategy numbers the reference exports (c-ext/constants.c:60-100; values of zstd.h) */
    PyModule_AddIntConstant(m, "CHAINLOG_MIN", 6); PyModule_AddIntConstant(m, "CHAINLOG_MAX", 30);
    PyModule_AddIntConstant(m, "HASHLOG_MIN", 6); PyModule_AddIntConstant(m, "HASHLOG_MAX", 30);
    PyModule_AddIntConstant(m, "SEARCHLOG_MIN", 1); PyModule_AddIntConstant(m, "SEARCHLOG_MAX", 30);
    PyModule_AddIntConstant(m, "MINMATCH_MIN", 3); PyModule_AddIntConstant(m, "MINMATCH_MAX", 7);
    PyModule_AddIntConstant(m, "TARGETLENGTH_MIN", 0); PyModule_AddIntConstant(m, "TARGETLENGTH_MAX", 1 << 17);
    PyModule_AddIntConstant(m, "STRATEGY_FAST", 1); PyModule_AddIntConstant(m, "STRATEGY_DFAST", 2); PyModule_AddIntConstant(m, "STRATEGY_GREEDY", 3);
    PyModule_AddIntConstant(m, "STRATEGY_LAZY", 4); PyModule_AddIntConstant(m, "STRATEGY_LAZY2", 5); PyModule_AddIntConstant(m, "STRATEGY_BTLAZY2", 6);
    PyModule_AddIntConstant(m, "STRATEGY_BTOPT", 7); PyModule_AddIntConstant(m, "STRATEGY_BTULTRA", 8); PyModule_AddIntConstant(m, "STRATEGY_BTULTRA2", 9);
    PyModule_AddStringConstant(m, "backend", "hip_cext");
    {   PyObject* feats = PySet_New(NULL);
        const char* f[] = { "buffer_types", "multi_compress_to_buffer", "multi_decompress_to_buffer" };
        for (int i = 0; i < 3; i++) { PyObject* sv = PyUnicode_FromString(f[i]); PySet_Add(feats, sv); Py_DECREF(sv); }
        PyModule_AddObject(m, "backend_features", feats);
        PyModule_AddObject(m, "ZSTD_VERSION", Py_BuildValue("(iii)", 1, 5, 7)); }
    return m;
}

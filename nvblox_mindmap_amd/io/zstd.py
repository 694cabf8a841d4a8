"""Minimal zstd binding over the system's libzstd (stable C API) -- the ``zstandard`` Python package the reference
imports (mapping/helpers/nvblox_to_disk_helpers.py:14, data_loading/dataset.py:410-415) is not in this image.
``compress(data, level)`` writes a single frame with the content size in its header, exactly what
``zstandard.ZstdCompressor(level).compress`` produces; ``decompress`` also accepts frames without a content size (streamed
by other writers) and concatenated frames."""
import ctypes as C
import ctypes.util


class _Buf(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


_CONTENTSIZE_UNKNOWN = 2 ** 64 - 1
_CONTENTSIZE_ERROR = 2 ** 64 - 2
_lib = None


def _z():
    global _lib
    if _lib is None:
        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        L = C.CDLL(name)
        L.ZSTD_compressBound.restype = C.c_size_t
        L.ZSTD_compressBound.argtypes = [C.c_size_t]
        L.ZSTD_compress.restype = C.c_size_t
        L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.ZSTD_decompress.restype = C.c_size_t
        L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_isError.restype = C.c_uint
        L.ZSTD_isError.argtypes = [C.c_size_t]
        L.ZSTD_getErrorName.restype = C.c_char_p
        L.ZSTD_getErrorName.argtypes = [C.c_size_t]
        L.ZSTD_getFrameContentSize.restype = C.c_ulonglong
        L.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_findFrameCompressedSize.restype = C.c_size_t
        L.ZSTD_findFrameCompressedSize.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_createDStream.restype = C.c_void_p
        L.ZSTD_freeDStream.argtypes = [C.c_void_p]
        L.ZSTD_initDStream.restype = C.c_size_t
        L.ZSTD_initDStream.argtypes = [C.c_void_p]
        L.ZSTD_decompressStream.restype = C.c_size_t
        L.ZSTD_decompressStream.argtypes = [C.c_void_p, C.POINTER(_Buf), C.POINTER(_Buf)]
        _lib = L
    return _lib


def _check(L, rc, what):
    if L.ZSTD_isError(rc):
        raise ValueError(f"zstd {what}: {L.ZSTD_getErrorName(rc).decode()}")
    return rc


def compress(data: bytes, level: int = 1) -> bytes:
    L = _z()
    src = bytes(data)
    cap = L.ZSTD_compressBound(len(src))
    dst = C.create_string_buffer(cap)
    n = _check(L, L.ZSTD_compress(dst, cap, src, len(src), int(level)), "compress")
    return dst.raw[:n]


def _decompress_stream(L, src: bytes) -> bytes:
    zds = L.ZSTD_createDStream()
    try:
        _check(L, L.ZSTD_initDStream(zds), "initDStream")
        inbuf = C.create_string_buffer(src, len(src))
        ib = _Buf(C.cast(inbuf, C.c_void_p), len(src), 0)
        chunk = C.create_string_buffer(1 << 20)
        out = []
        while ib.pos < ib.size:
            ob = _Buf(C.cast(chunk, C.c_void_p), len(chunk), 0)
            _check(L, L.ZSTD_decompressStream(zds, C.byref(ob), C.byref(ib)), "decompressStream")
            out.append(chunk.raw[:ob.pos])
            if ob.pos == 0 and ib.pos >= ib.size:
                break
        return b"".join(out)
    finally:
        L.ZSTD_freeDStream(zds)


def decompress(data: bytes) -> bytes:
    """All frames of `data`, concatenated.  A single frame with a known content size (what the reference's writer produces)
    is inflated straight into the returned buffer: no slice of the input, no intermediate copy of the output (a vertex-feature
    file is tens of MB: three extra copies of it were a third of the loader's time per sample).  Returns a bytes-like object
    (``bytearray`` on that path)."""
    L = _z()
    src = data if isinstance(data, bytes) else bytes(data)
    base = C.cast(C.c_char_p(src), C.c_void_p).value
    out, pos = [], 0
    while pos < len(src):
        left = len(src) - pos
        ptr = C.c_void_p(base + pos)
        size = L.ZSTD_getFrameContentSize(ptr, left)
        if size == _CONTENTSIZE_ERROR:
            raise ValueError("not a zstd frame")
        if size == _CONTENTSIZE_UNKNOWN:
            out.append(_decompress_stream(L, src[pos:]))
            break
        clen = _check(L, L.ZSTD_findFrameCompressedSize(ptr, left), "findFrameCompressedSize")
        dst = bytearray(max(int(size), 1))
        n = _check(L, L.ZSTD_decompress((C.c_char * len(dst)).from_buffer(dst), int(size), ptr, clen), "decompress")
        if n != len(dst):
            del dst[n:]
        out.append(dst)
        pos += clen
    if len(out) == 1:
        return out[0]
    return b"".join(out)

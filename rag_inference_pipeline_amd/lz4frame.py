"""LZ4 frame format around the native block compressor (csrc/rag_lz4.cpp) — the `compressed` document
payload of the retrieval response (reference services/retrieval/api.py:516-523 produces it with
lz4.frame.compress; services/generation/service.py:429 reads it with lz4.frame.decompress).

compress() writes a standard frame any LZ4 implementation reads: magic, FLG = version 01 + independent
blocks, BD = 4 MiB blocks, header checksum, data blocks (stored raw when compression does not pay),
end mark; no content size, no content checksum.  decompress() is the matching reader, in plain Python:
the retrieval node never needs it, tests and small tools do.
"""

from __future__ import annotations

import ctypes as C
import struct

from . import _native

_MAGIC = b"\x04\x22\x4d\x18"
_FLG, _BD = 0x60, 0x70
_BLOCK = 4 << 20


def _u8p(b: bytes | bytearray):
    return C.cast(C.c_char_p(bytes(b)) if isinstance(b, bytes) else (C.c_char * len(b)).from_buffer(b), C.c_void_p)


def compress(data: bytes) -> bytes:
    lib = _native.lib()
    desc = bytes([_FLG, _BD])
    hc = (lib.rag_xxh32(_u8p(desc), len(desc), 0) >> 8) & 0xFF
    out = [_MAGIC, desc, bytes([hc])]
    for lo in range(0, len(data), _BLOCK):
        chunk = data[lo:lo + _BLOCK]
        cap = int(lib.rag_lz4_compress_bound(len(chunk)))
        dst = C.create_string_buffer(cap)
        n = int(lib.rag_lz4_block_compress(_u8p(chunk), len(chunk), C.cast(dst, C.c_void_p), cap))
        if 0 <= n < len(chunk):
            out += [struct.pack("<I", n), dst.raw[:n]]
        else:  # incompressible: stored block (high bit of the size word)
            out += [struct.pack("<I", len(chunk) | 0x80000000), chunk]
    out.append(b"\x00\x00\x00\x00")
    return b"".join(out)


def _block_decode(src: bytes) -> bytes:
    out = bytearray()
    i, n = 0, len(src)
    while i < n:
        token = src[i]; i += 1
        lit = token >> 4
        if lit == 15:
            while True:
                b = src[i]; i += 1
                lit += b
                if b != 255:
                    break
        out += src[i:i + lit]; i += lit
        if i >= n:
            break  # the last sequence has no match part
        offset = src[i] | (src[i + 1] << 8); i += 2
        mlen = token & 15
        if mlen == 15:
            while True:
                b = src[i]; i += 1
                mlen += b
                if b != 255:
                    break
        mlen += 4
        if offset == 0 or offset > len(out):
            raise ValueError("corrupt LZ4 block: bad match offset")
        start = len(out) - offset
        for j in range(mlen):  # byte-wise: matches may overlap their own output
            out.append(out[start + j])
    return bytes(out)


def decompress(frame: bytes) -> bytes:
    if frame[:4] != _MAGIC:
        raise ValueError("not an LZ4 frame")
    flg = frame[4]
    if (flg >> 6) != 1:
        raise ValueError("unsupported LZ4 frame version")
    pos = 6 + (8 if flg & 0x08 else 0) + (4 if flg & 0x01 else 0) + 1  # FLG BD [content size] [dict id] HC
    block_checksum = bool(flg & 0x10)
    out = []
    while True:
        (size,) = struct.unpack_from("<I", frame, pos); pos += 4
        if size == 0:
            break
        raw, size = bool(size & 0x80000000), size & 0x7FFFFFFF
        body = frame[pos:pos + size]; pos += size + (4 if block_checksum else 0)
        out.append(body if raw else _block_decode(body))
    return b"".join(out)

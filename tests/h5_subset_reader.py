"""Test infrastructure: an independent reader for the subset of the HDF5 file format that XDMFTensorOutput's container uses
(version 0 superblock, symbol-table root group with a version 1 B-tree, version 1 object headers, contiguous datasets of
IEEE floats / two's-complement integers), written from the format specification.  It checks the structure as it walks it
(signatures, sorted link names, key ordering, alignment, end-of-file address), so that the writer in marlin_amd/csrc/h5write.hip
is verified on machines without libhdf5 tools; tests additionally cross-check with h5dump where the image has it."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class H5FormatError(AssertionError):
    pass


def _need(cond, msg):
    if not cond:
        raise H5FormatError(msg)


def _messages(buf, addr):
    ver, _, nmsg, refcnt, size = struct.unpack_from("<BBHII", buf, addr)
    _need(ver == 1 and refcnt == 1, "object header version / reference count")
    _need(addr % 8 == 0 and size % 8 == 0, "object header alignment")
    at, end, out = addr + 16, addr + 16 + size, []
    for _ in range(nmsg):
        mtype, msize, flags = struct.unpack_from("<HHB", buf, at)
        _need(msize % 8 == 0 and at + 8 + msize <= end, "message size")
        out.append((mtype, buf[at + 8:at + 8 + msize]))
        at += 8 + msize
    _need(at == end, "object header size does not match its messages")
    return out


def _dataset(buf, addr):
    shape = dtype = layout = None
    for mtype, body in _messages(buf, addr):
        if mtype == 0x0001:
            ver, rank, flags = struct.unpack_from("<BBB", body, 0)
            _need(ver == 1 and flags == 0, "dataspace version / flags")
            shape = struct.unpack_from(f"<{rank}Q", body, 8)
        elif mtype == 0x0003:
            cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", body, 0)
            _need(cv >> 4 == 1 and (b0 & 1) == 0, "datatype version / byte order")
            if cv & 15 == 1:
                off, prec, eloc, esize, mloc, msize, bias = struct.unpack_from("<HHBBBBI", body, 8)
                _need((size, b1, prec, eloc, esize, mloc, msize, bias) in ((8, 63, 64, 52, 11, 0, 52, 1023), (4, 31, 32, 23, 8, 0, 23, 127)),
                      "not an IEEE float layout")
                dtype = np.dtype("<f8" if size == 8 else "<f4")
            else:
                _need(cv & 15 == 0 and (b0 & 8), "datatype class")
                off, prec = struct.unpack_from("<HH", body, 8)
                _need(off == 0 and prec == 8 * size, "integer precision")
                dtype = np.dtype(f"<i{size}")
        elif mtype == 0x0008:
            ver, cls = struct.unpack_from("<BB", body, 0)
            _need(ver == 3 and cls == 1, "layout: contiguous, version 3")
            layout = struct.unpack_from("<QQ", body, 2)
    _need(shape is not None and dtype is not None and layout is not None, "dataset header lacks dataspace / datatype / layout")
    daddr, dbytes = layout
    _need(daddr % 8 == 0 and dbytes == int(np.prod(shape)) * dtype.itemsize and daddr + dbytes <= len(buf), "dataset extent")
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape)), offset=daddr).reshape(shape)


def read_h5(path):
    """{name: array} of every dataset in the root group; raises H5FormatError on any structural inconsistency."""
    buf = open(path, "rb").read()
    _need(buf[:8] == b"\x89HDF\r\n\x1a\n", "signature")
    sb_ver, _, root_ver, _, _, so, sl, _, leaf_k, int_k, flags = struct.unpack_from("<BBBBBBBBHHI", buf, 8)
    _need((sb_ver, root_ver, so, sl, flags) == (0, 0, 8, 8, 0), "superblock fields")
    base, freesp, eof, drv = struct.unpack_from("<QQQQ", buf, 24)
    _need(base == 0 and freesp == UNDEF and drv == UNDEF and eof <= len(buf), f"superblock addresses (eof {eof}, file {len(buf)})")  # (data appended since the last flush lies beyond the end-of-file address: libhdf5 accepts that too)
    name_off, root, cache, _, btree, heap = struct.unpack_from("<QQIIQQ", buf, 56)
    _need(name_off == 0 and cache == 1, "root symbol table entry")
    msgs = _messages(buf, root)
    _need(len(msgs) == 1 and msgs[0][0] == 0x0011 and struct.unpack_from("<QQ", msgs[0][1], 0) == (btree, heap), "root group header")
    _need(buf[heap:heap + 4] == b"HEAP", "local heap signature")
    hsize, hfree, hdata = struct.unpack_from("<QQQ", buf, heap + 8)
    _need(hfree == 1 and hdata + hsize <= len(buf) and buf[hdata] == 0, "local heap")

    def name_at(off):
        _need(off < hsize and off % 8 == 0, "name offset")
        end = buf.index(b"\0", hdata + off)
        return buf[hdata + off:end].decode()

    _need(buf[btree:btree + 4] == b"TREE", "B-tree signature")
    ntype, level, used, left, right = struct.unpack_from("<BBHQQ", buf, btree + 4)
    _need(ntype == 0 and level == 0 and left == UNDEF and right == UNDEF and used <= 2 * int_k, "B-tree node")
    out, prev = {}, ""
    keys = [struct.unpack_from("<Q", buf, btree + 24 + 16 * i)[0] for i in range(used + 1)]
    _need(name_at(keys[0]) == "", "first B-tree key")
    for i in range(used):
        child = struct.unpack_from("<Q", buf, btree + 24 + 16 * i + 8)[0]
        _need(buf[child:child + 4] == b"SNOD", "symbol table node signature")
        ver, _, nsym = struct.unpack_from("<BBH", buf, child + 4)
        _need(ver == 1 and 0 < nsym <= 2 * leaf_k, "symbol table node")
        for e in range(nsym):
            noff, oh, ctype = struct.unpack_from("<QQI", buf, child + 8 + 40 * e)
            name = name_at(noff)
            _need(ctype == 0 and name > prev, f"link names must be strictly increasing ({prev!r} then {name!r})")
            out[name] = _dataset(buf, oh)
            prev = name
        _need(name_at(keys[i + 1]) == prev, "B-tree key is not the last name of its child")
    return out

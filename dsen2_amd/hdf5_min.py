"""A small read-only HDF5 reader: just enough of the file format to open what this path is handed as HDF5 —
keras weight checkpoints (`model.save_weights` / `ModelCheckpoint`, training/supres_train.py:195-201, read back by
testing/supres.py:63) and MATLAB v7.3 tiles (data/*.mat, read by testing/demoDSen2.py:14-28) — on a machine without h5py
(the ROCm image has none).  Nothing here is specific to those two producers; what is NOT implemented raises
UnsupportedHDF5 by name instead of guessing.

Implemented (HDF5 File Format Specification 1.x / 2.0 / 3.0 names):
  superblock v0/v1 (any user-block offset: MATLAB's is 512) and v2/v3; object headers v1 and v2 (continuations incl.);
  old-style groups (symbol table: v1 B-tree + local heap + SNOD) and new-style compact groups (link messages);
  datasets with compact / contiguous / chunked (v1 B-tree index) layout (layout message v1-v3), filters deflate, shuffle,
  fletcher32; datatypes fixed-point, IEEE float (either byte order), fixed-length string, variable-length string (global
  heap); attributes v1-v3 stored in the object header; fill values for unallocated storage.
  Of libver='latest' files: layout message v4 with compact / contiguous storage and the single-chunk and implicit
  chunk indices.
Not implemented: dense link / attribute storage (fractal heaps), fixed-array / extensible-array / v2-B-tree chunk indices, compound / enum / array / reference / opaque / bitfield types, shared (committed) datatypes, external storage, soft and
  external links, szip / n-bit / scale-offset filters.

The surface is the part of h5py's that the callers use: File(path) as a context manager, `name in g`, g[name] with
'/'-separated paths, g.keys(), g.attrs (a dict), Dataset.shape / .dtype / np.asarray(ds) / ds[()].
"""
import mmap
import zlib

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'


class UnsupportedHDF5(NotImplementedError):
    """A valid HDF5 feature this reader does not implement (the message names it)."""


class _Map(object):
    """The mapped file, read-only, with ONE rule: a read that reaches past the end of the file (or starts at an undefined
    address) is a damaged file and says so — a plain mmap slice would silently come back short and decode to a wrong
    integer or address."""
    def __init__(self, mm):
        self._mm, self._n = mm, len(mm)

    def __len__(self):
        return self._n

    def __getitem__(self, k):
        if isinstance(k, slice):
            a, b = (0 if k.start is None else k.start), (self._n if k.stop is None else k.stop)
            if not (isinstance(a, int) and isinstance(b, int)) or a < 0 or b < a or b > self._n or k.step is not None:
                raise ValueError('HDF5 file truncated or corrupt: bytes [%r, %r) of a %d-byte file' % (k.start, k.stop, self._n))
            return self._mm[a:b]
        if not isinstance(k, int) or not 0 <= k < self._n:
            raise ValueError('HDF5 file truncated or corrupt: byte %r of a %d-byte file' % (k, self._n))
        return self._mm[k]

    def find(self, what, start, end):
        if not (isinstance(start, int) and isinstance(end, int)) or start < 0 or end < start or end > self._n:
            raise ValueError('HDF5 file truncated or corrupt: bytes [%r, %r) of a %d-byte file' % (start, end, self._n))
        return self._mm.find(what, start, end)


def _fletcher32(data):
    """The checksum the library's fletcher32 filter stores after a chunk (H5_checksum_fletcher32): Fletcher's sums over
    big-endian 16-bit words, an odd last byte as the high half of a word, each sum folded to 16 bits."""
    n = len(data) // 2
    w = np.frombuffer(data, '>u2', n).astype(np.uint64)
    s1 = int(w.sum())
    # s2 = sum of the running sums = sum_i (n - i) * w_i; in blocks, so that no partial sum leaves 64 bits whatever the chunk's size
    s2, block = 0, 1 << 15
    for a in range(0, n, block):
        b = min(n, a + block)
        s2 += int((w[a:b] * np.arange(n - a, n - b, -1, dtype=np.uint64)).sum())
    if len(data) & 1:
        s1 += data[-1] << 8
        s2 += s1
    fold = lambda s: (s % 65535) or (65535 if s else 0)      # noqa: E731  (end-around carry: a non-zero multiple of 65535 stays 0xffff)
    return (fold(s2) << 16) | fold(s1)


class _Buf(object):
    """The mapped file with the superblock's sizes: addresses in the file are relative to `base`."""
    def __init__(self, mm):
        self.mm = mm if isinstance(mm, _Map) else _Map(mm)
        self.base = 0
        self.O = 8                                  # size of offsets
        self.L = 8                                  # size of lengths

    def u(self, pos, n):
        return int.from_bytes(self.mm[pos:pos + n], 'little')

    def off(self, pos):
        """An address field at ABSOLUTE position pos -> absolute position in the file, or None for 'undefined'."""
        raw = self.mm[pos:pos + self.O]
        if raw == b'\xff' * self.O:
            return None
        return self.base + int.from_bytes(raw, 'little')

    def off_from(self, data, pos):
        """The same for an address field inside a message body that has already been read."""
        raw = data[pos:pos + self.O]
        if raw == b'\xff' * self.O:
            return None
        return self.base + int.from_bytes(raw, 'little')

    def length(self, pos):
        return int.from_bytes(self.mm[pos:pos + self.L], 'little')

    def bytes(self, pos, n):
        if pos < 0 or pos + n > len(self.mm):
            raise ValueError('HDF5 file truncated: %d bytes at %d, file has %d' % (n, pos, len(self.mm)))
        return self.mm[pos:pos + n]


def _pad8(n):
    return (n + 7) & ~7


# ---------------------------------------------------------------------------------------------------------- datatypes
class _Type(object):
    def __init__(self, kind, dtype=None, size=0, base=None, strpad=0):
        self.kind, self.dtype, self.size, self.base, self.strpad = kind, dtype, size, base, strpad


def _parse_datatype(raw):
    cls, ver = raw[0] & 15, raw[0] >> 4
    b0 = raw[1]
    size = int.from_bytes(raw[4:8], 'little')
    if cls == 0:                                                         # fixed-point
        prec = int.from_bytes(raw[10:12], 'little')
        if prec != 8 * size or size not in (1, 2, 4, 8):
            raise UnsupportedHDF5('integer type of %d bytes with %d bits of precision' % (size, prec))
        return _Type('num', np.dtype(('>' if b0 & 1 else '<') + ('i' if b0 & 8 else 'u') + str(size)), size)
    if cls == 1:                                                         # floating point
        if size not in (2, 4, 8) or (b0 & 0x40):
            raise UnsupportedHDF5('floating-point type of %d bytes (byte-order bits %#x)' % (size, b0))
        expect = {2: (10, 5, 0, 10), 4: (23, 8, 0, 23), 8: (52, 11, 0, 52)}[size]
        got = (raw[12], raw[13], raw[14], raw[15])                     # exponent location, size, mantissa location, size
        if got != expect:
            raise UnsupportedHDF5('non-IEEE floating-point layout %r' % (got,))
        return _Type('num', np.dtype(('>' if b0 & 1 else '<') + 'f' + str(size)), size)
    if cls == 3:                                                         # fixed-length string
        return _Type('str', np.dtype('S%d' % size), size, strpad=b0 & 15)
    if cls == 9:                                                         # variable length
        base = _parse_datatype(raw[8:])
        return _Type('vlen_str' if (b0 & 15) == 1 else 'vlen', None, size, base=base)
    names = {2: 'time', 4: 'bitfield', 5: 'opaque', 6: 'compound', 7: 'reference', 8: 'enumerated', 10: 'array'}
    raise UnsupportedHDF5('%s datatype (class %d, version %d)' % (names.get(cls, 'unknown'), cls, ver))


def _parse_dataspace(buf, raw):
    ver, rank, flags = raw[0], raw[1], raw[2]
    if ver == 1:
        p = 8
    elif ver == 2:
        if raw[3] == 2:
            return None                                                  # null dataspace
        p = 4
    else:
        raise UnsupportedHDF5('dataspace message version %d' % ver)
    return tuple(int.from_bytes(raw[p + i * buf.L:p + (i + 1) * buf.L], 'little') for i in range(rank))


# ------------------------------------------------------------------------------------------------------ object headers
class _Message(object):
    __slots__ = ('type', 'flags', 'data')

    def __init__(self, mtype, flags, data):
        self.type, self.flags, self.data = mtype, flags, data


def _read_object_header(buf, addr):
    """All messages of the object header at absolute address addr, continuation blocks followed."""
    mm = buf.mm
    msgs = []
    if mm[addr:addr + 4] == b'OHDR':
        ver, flags = mm[addr + 4], mm[addr + 5]
        if ver != 2:
            raise UnsupportedHDF5('object header version %d' % ver)
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        nsz = 1 << (flags & 3)
        chunk0 = buf.u(p, nsz)
        p += nsz
        blocks = [(p, p + chunk0)]
        track = bool(flags & 4)
        seen = 0
        while blocks:
            p, end = blocks.pop(0)
            seen += 1
            if seen > 4096:
                raise ValueError('HDF5: object header at %d has more than 4096 continuation blocks (corrupt file?)' % addr)
            while p + 4 <= end:                                          # (a gap shorter than a message header may follow)
                mtype, msize, mflags = mm[p], buf.u(p + 1, 2), mm[p + 3]
                p += 4 + (2 if track else 0)
                data = bytes(buf.bytes(p, msize))
                p += msize
                if mtype == 0x10:
                    caddr, clen = buf.off_from(data, 0), int.from_bytes(data[buf.O:buf.O + buf.L], 'little')
                    if mm[caddr:caddr + 4] != b'OCHK':
                        raise ValueError('HDF5: continuation block without OCHK signature at %d' % caddr)
                    blocks.append((caddr + 4, caddr + clen - 4))
                elif mtype != 0:
                    msgs.append(_Message(mtype, mflags, data))
        return msgs
    ver = mm[addr]
    if ver != 1:
        raise ValueError('HDF5: no object header at %d (version byte %d)' % (addr, ver))
    nmsg = buf.u(addr + 2, 2)
    size = buf.u(addr + 8, 4)
    blocks = [(addr + 16, addr + 16 + size)]
    seen = 0
    while blocks:
        p, end = blocks.pop(0)
        seen += 1
        if seen > 4096:
            raise ValueError('HDF5: object header at %d has more than 4096 continuation blocks (corrupt file?)' % addr)
        while p + 8 <= end and nmsg > 0:
            mtype, msize, mflags = buf.u(p, 2), buf.u(p + 2, 2), mm[p + 4]
            data = bytes(buf.bytes(p + 8, msize))
            p += 8 + msize
            nmsg -= 1
            if mtype == 0x10:
                caddr, clen = buf.off_from(data, 0), int.from_bytes(data[buf.O:buf.O + buf.L], 'little')
                blocks.append((caddr, caddr + clen))
            elif mtype != 0:
                msgs.append(_Message(mtype, mflags, data))
    return msgs


# ------------------------------------------------------------------------------------------------- groups (old style)
def _local_heap_data(buf, addr):
    if buf.mm[addr:addr + 4] != b'HEAP':
        raise ValueError('HDF5: no local heap at %d' % addr)
    size = buf.length(addr + 8)
    data = buf.off(addr + 8 + 2 * buf.L)
    return data, size


def _heap_name(buf, heap, offset):
    data, size = heap
    end = buf.mm.find(b'\0', data + offset, data + size)
    if end < 0:
        end = data + size
    return bytes(buf.mm[data + offset:end]).decode('utf-8')


def _walk_group_btree(buf, addr, heap, out, depth=0):
    mm = buf.mm
    if depth > 32:
        raise ValueError('HDF5: group B-tree deeper than 32 levels (corrupt file?)')
    if mm[addr:addr + 4] == b'SNOD':
        n = buf.u(addr + 6, 2)
        p = addr + 8
        for _ in range(n):
            name = _heap_name(buf, heap, int.from_bytes(mm[p:p + buf.O], 'little'))
            cache = buf.u(p + 2 * buf.O, 4)
            if cache == 2:
                raise UnsupportedHDF5('symbolic link %r' % name)
            out.append((name, buf.off(p + buf.O)))
            p += 2 * buf.O + 24
        return
    if mm[addr:addr + 4] != b'TREE' or mm[addr + 4] != 0:
        raise ValueError('HDF5: no group B-tree node at %d' % addr)
    used = buf.u(addr + 6, 2)
    p = addr + 8 + 2 * buf.O
    for _ in range(used):
        p += buf.L                                                       # key
        _walk_group_btree(buf, buf.off(p), heap, out, depth + 1)
        p += buf.O


def _parse_link(buf, raw):
    """Link message (new-style groups) -> (name, absolute object header address)."""
    ver, flags = raw[0], raw[1]
    if ver != 1:
        raise UnsupportedHDF5('link message version %d' % ver)
    p = 2
    ltype = 0
    if flags & 8:
        ltype = raw[p]
        p += 1
    if flags & 4:
        p += 8
    if flags & 16:
        p += 1
    nsz = 1 << (flags & 3)
    nlen = int.from_bytes(raw[p:p + nsz], 'little')
    p += nsz
    name = raw[p:p + nlen].decode('utf-8')
    p += nlen
    if ltype != 0:
        raise UnsupportedHDF5('%s link %r' % ({1: 'soft', 64: 'external'}.get(ltype, 'user-defined'), name))
    return name, buf.off_from(raw, p)


# ------------------------------------------------------------------------------------------------------------ objects
class _Object(object):
    def __init__(self, buf, addr, name):
        self._buf, self._addr, self.name = buf, addr, name
        self._msgs = _read_object_header(buf, addr)
        for m in self._msgs:
            if m.flags & 2 and m.type in (1, 3, 5, 11):
                raise UnsupportedHDF5('%s: shared (committed) header message of type %#x' % (name, m.type))
        self._attrs = None

    def _first(self, mtype):
        for m in self._msgs:
            if m.type == mtype:
                return m
        return None

    @property
    def attrs(self):
        if self._attrs is None:
            buf = self._buf
            out = {}
            info = self._first(0x15)
            if info is not None:
                raw = info.data
                p = 2 + (2 if raw[1] & 1 else 0)
                if buf.off_from(raw, p) is not None:
                    raise UnsupportedHDF5('%s: attributes in dense storage (fractal heap)' % self.name)
            for m in self._msgs:
                if m.type == 0x0C:
                    name, value = _parse_attribute(buf, m.data, self.name)
                    out[name] = value
            self._attrs = out
        return self._attrs


def _parse_attribute(buf, raw, owner):
    ver = raw[0]
    if ver not in (1, 2, 3):
        raise UnsupportedHDF5('attribute message version %d' % ver)
    nlen, tlen, slen = (int.from_bytes(raw[i:i + 2], 'little') for i in (2, 4, 6))
    if ver >= 2 and raw[1] & 3:
        raise UnsupportedHDF5('%s: attribute with a shared datatype / dataspace' % owner)
    p = 8 + (1 if ver == 3 else 0)
    pad = _pad8 if ver == 1 else (lambda n: n)
    name = raw[p:p + nlen].split(b'\0')[0].decode('utf-8')
    p += pad(nlen)
    traw = raw[p:p + tlen]
    p += pad(tlen)
    sraw = raw[p:p + slen]
    p += pad(slen)
    try:
        typ = _parse_datatype(traw)
    except UnsupportedHDF5:
        return name, None                                                # an attribute nobody on this path reads
    shape = _parse_dataspace(buf, sraw)
    if shape is None:
        return name, None
    return name, _decode(buf, typ, shape, raw[p:], scalar_ok=True)


def _global_heap_object(buf, addr, index):
    mm = buf.mm
    if mm[addr:addr + 4] != b'GCOL':
        raise ValueError('HDF5: no global heap collection at %d' % addr)
    size = buf.length(addr + 8)
    p, end = addr + 8 + buf.L, addr + size
    while p + 8 + buf.L <= end:
        idx, osize = buf.u(p, 2), buf.length(p + 8)
        if idx == 0:
            break
        if idx == index:
            return bytes(mm[p + 8 + buf.L:p + 8 + buf.L + osize])
        p += 8 + buf.L + _pad8(osize)
    raise ValueError('HDF5: global heap object %d not found in the collection at %d' % (index, addr))


def _decode(buf, typ, shape, raw, scalar_ok=False):
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    if typ.kind == 'num' or typ.kind == 'str':
        a = np.frombuffer(raw, dtype=typ.dtype, count=n).reshape(shape)
        if typ.kind == 'num' and not typ.dtype.isnative:
            a = a.astype(typ.dtype.newbyteorder('='))
        else:
            a = a.copy()
        if typ.kind == 'str' and typ.strpad == 2:
            a = np.char.rstrip(a, b' ')
        if scalar_ok and shape == ():
            return a[()]
        return a
    if typ.kind == 'vlen_str':
        out = np.empty(n, dtype=object)
        step = 4 + buf.O + 4
        for i in range(n):
            rec = raw[i * step:(i + 1) * step]
            length = int.from_bytes(rec[:4], 'little')
            haddr = buf.off_from(rec, 4)
            idx = int.from_bytes(rec[4 + buf.O:], 'little')
            out[i] = '' if haddr is None or length == 0 else _global_heap_object(buf, haddr, idx)[:length].decode('utf-8')
        out = out.reshape(shape)
        if scalar_ok and shape == ():
            return out[()]
        return out
    raise UnsupportedHDF5('variable-length sequence data')


class Dataset(_Object):
    def __init__(self, buf, addr, name):
        _Object.__init__(self, buf, addr, name)
        space, dtype = self._first(1), self._first(3)
        if space is None or dtype is None or self._first(8) is None:
            raise ValueError('HDF5: %s is not a dataset' % name)
        self.shape = _parse_dataspace(buf, space.data)
        if self.shape is None:
            raise UnsupportedHDF5('%s: null dataspace' % name)
        self._type = _parse_datatype(dtype.data)
        self.dtype = self._type.dtype.newbyteorder('=') if self._type.kind == 'num' else \
            (self._type.dtype if self._type.kind == 'str' else np.dtype(object))

    def __len__(self):
        return self.shape[0]

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def _fill(self):
        m = self._first(5)
        if m is None:
            return None
        raw = m.data
        if raw[0] in (1, 2):
            if raw[0] == 2 and not raw[3]:
                return None
            size = int.from_bytes(raw[4:8], 'little') if len(raw) >= 8 else 0
            return raw[8:8 + size] if size else None
        if raw[0] == 3:
            if raw[1] & 0x20:
                size = int.from_bytes(raw[2:6], 'little')
                return raw[6:6 + size] if size else None
            return None
        raise UnsupportedHDF5('fill value message version %d' % raw[0])

    def _filters(self):
        m = self._first(0x0B)
        if m is None:
            return []
        raw = m.data
        ver, n = raw[0], raw[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = int.from_bytes(raw[p:p + 2], 'little')
            p += 2
            if ver == 1 or fid >= 256:
                nlen = int.from_bytes(raw[p:p + 2], 'little')
                p += 2
            else:
                nlen = 0
            p += 2                                                       # flags
            nval = int.from_bytes(raw[p:p + 2], 'little')
            p += 2
            p += _pad8(nlen) if ver == 1 else nlen
            vals = [int.from_bytes(raw[p + 4 * i:p + 4 * i + 4], 'little') for i in range(nval)]
            p += 4 * nval
            if ver == 1 and nval & 1:
                p += 4
            if fid not in (1, 2, 3):
                raise UnsupportedHDF5('%s: filter %d (%s)' % (self.name, fid, {4: 'szip', 5: 'n-bit', 6: 'scale-offset'}.get(
                    fid, 'registered third-party filter')))
            out.append((fid, vals))
        return out

    def _unfilter(self, chunk, mask, filters, itemsize):
        for i in reversed(range(len(filters))):
            if mask & (1 << i):
                continue
            fid = filters[i][0]
            if fid == 3:
                if len(chunk) < 4:
                    raise ValueError('HDF5: %s: a fletcher32 chunk of %d bytes' % (self.name, len(chunk)))
                stored, chunk = int.from_bytes(chunk[-4:], 'little'), chunk[:-4]
                if _fletcher32(bytes(chunk)) != stored:
                    raise ValueError('HDF5: %s: fletcher32 checksum mismatch (stored %#010x): the file is damaged'
                                     % (self.name, stored))
            elif fid == 1:
                chunk = zlib.decompress(chunk)
            elif fid == 2:
                n = len(chunk) // itemsize
                body = np.frombuffer(chunk, np.uint8, n * itemsize).reshape(itemsize, n).T.tobytes()
                chunk = body + bytes(chunk[n * itemsize:])
        return chunk

    def _chunks(self, addr, rank1, out, depth=0):
        buf, mm = self._buf, self._buf.mm
        if depth > 32:
            raise ValueError('HDF5: chunk B-tree deeper than 32 levels (corrupt file?)')
        if mm[addr:addr + 4] != b'TREE' or mm[addr + 4] != 1:
            raise ValueError('HDF5: %s: no chunk B-tree node at %d' % (self.name, addr))
        level, used = mm[addr + 5], buf.u(addr + 6, 2)
        p = addr + 8 + 2 * buf.O
        ksz = 8 + 8 * rank1
        for _ in range(used):
            size, mask = buf.u(p, 4), buf.u(p + 4, 4)
            offs = tuple(buf.u(p + 8 + 8 * i, 8) for i in range(rank1 - 1))
            child = buf.off(p + ksz)
            if level == 0:
                out.append((offs, size, mask, child))
            else:
                self._chunks(child, rank1, out, depth + 1)
            p += ksz + buf.O

    def read(self):
        buf = self._buf
        raw = self._first(8).data
        typ = self._type
        itemsize = typ.size
        n = self.size
        ver = raw[0]
        fill = self._fill()

        def empty():
            a = np.zeros(n * itemsize, np.uint8)
            if fill is not None and len(fill) == itemsize and any(fill):
                a = np.tile(np.frombuffer(fill, np.uint8), n)
            return a
        if ver in (3, 4):
            cls = raw[1]
            if cls == 2 and ver == 4:
                data = self._read_chunked_v4(raw, empty)
            elif cls == 0:
                size = int.from_bytes(raw[2:4], 'little')
                data = raw[4:4 + size]
            elif cls == 1:
                addr, size = buf.off_from(raw, 2), int.from_bytes(raw[2 + buf.O:2 + buf.O + buf.L], 'little')
                data = empty().tobytes() if addr is None else buf.bytes(addr, min(size, n * itemsize))
            elif cls == 2:
                rank1 = raw[2]
                addr = buf.off_from(raw, 3)
                cdims = tuple(int.from_bytes(raw[3 + buf.O + 4 * i:7 + buf.O + 4 * i], 'little') for i in range(rank1))
                data = self._read_chunked(addr, rank1, cdims, empty)
            else:
                raise UnsupportedHDF5('%s: data layout class %d' % (self.name, cls))
        elif ver in (1, 2):
            rank1, cls = raw[1], raw[2]
            p = 8
            addr = None
            if cls != 0:
                addr = buf.off_from(raw, p)
                p += buf.O
            dims = tuple(int.from_bytes(raw[p + 4 * i:p + 4 * i + 4], 'little') for i in range(rank1))
            p += 4 * rank1
            if cls == 2:
                data = self._read_chunked(addr, rank1, dims, empty)
            elif cls == 1:
                data = empty().tobytes() if addr is None else buf.bytes(addr, n * itemsize)
            else:
                size = int.from_bytes(raw[p:p + 4], 'little')
                data = raw[p + 4:p + 4 + size]
        else:
            raise UnsupportedHDF5('%s: data layout message version %d' % (self.name, ver))
        if len(data) < n * itemsize:
            raise ValueError('HDF5: %s: storage holds %d bytes, the dataspace needs %d' % (self.name, len(data), n * itemsize))
        return _decode(buf, typ, self.shape, data)

    def _read_chunked(self, addr, rank1, cdims, empty):
        buf = self._buf
        typ = self._type
        if typ.kind not in ('num', 'str'):
            raise UnsupportedHDF5('%s: chunked variable-length data' % self.name)
        rank = rank1 - 1
        if rank != len(self.shape) or cdims[-1] != typ.size:
            raise ValueError('HDF5: %s: chunk rank / element size do not match the dataset' % self.name)
        out = empty().view(np.dtype('V%d' % typ.size)).reshape(self.shape)
        if addr is None:
            return out.tobytes()
        filters = self._filters()
        chunks = []
        self._chunks(addr, rank1, chunks)
        cshape = cdims[:-1]
        cbytes = int(np.prod(cshape, dtype=np.int64)) * typ.size
        for offs, size, mask, caddr in chunks:
            blob = self._unfilter(buf.bytes(caddr, size), mask, filters, typ.size)
            if len(blob) < cbytes:
                raise ValueError('HDF5: %s: chunk at %r holds %d bytes, needs %d' % (self.name, offs, len(blob), cbytes))
            c = np.frombuffer(blob, np.dtype('V%d' % typ.size), cbytes // typ.size).reshape(cshape)
            sel_o = tuple(slice(o, min(o + c_, s)) for o, c_, s in zip(offs, cshape, self.shape))
            sel_c = tuple(slice(0, s.stop - s.start) for s in sel_o)
            if any(s.stop <= s.start for s in sel_o):
                continue
            out[sel_o] = c[sel_c]
        return out.tobytes()

    def _read_chunked_v4(self, raw, empty):
        """Layout message version 4 (libver='latest'): the two index types that need no further structure."""
        buf, typ = self._buf, self._type
        flags, rank1, enc = raw[2], raw[3], raw[4]
        p = 5
        cdims = tuple(int.from_bytes(raw[p + enc * i:p + enc * (i + 1)], 'little') for i in range(rank1))
        p += enc * rank1
        index = raw[p]
        p += 1
        if typ.kind not in ('num', 'str') or rank1 - 1 != len(self.shape) or cdims[-1] != typ.size:
            raise UnsupportedHDF5('%s: chunked layout of this datatype / rank' % self.name)
        cshape = cdims[:-1]
        cbytes = int(np.prod(cshape, dtype=np.int64)) * typ.size
        out = empty().view(np.dtype('V%d' % typ.size)).reshape(self.shape)
        filters = self._filters()

        def place(offs, blob):
            c = np.frombuffer(blob, np.dtype('V%d' % typ.size), cbytes // typ.size).reshape(cshape)
            sel_o = tuple(slice(o, min(o + c_, s)) for o, c_, s in zip(offs, cshape, self.shape))
            out[sel_o] = c[tuple(slice(0, s.stop - s.start) for s in sel_o)]
        if index == 1:                                                   # single chunk
            size, mask = cbytes, 0
            if flags & 2:
                size, mask = int.from_bytes(raw[p:p + buf.L], 'little'), int.from_bytes(raw[p + buf.L:p + buf.L + 4], 'little')
                p += buf.L + 4
            addr = buf.off_from(raw, p)
            if addr is not None:
                place((0,) * len(cshape), self._unfilter(buf.bytes(addr, size), mask, filters, typ.size))
        elif index == 2:                                                 # implicit: unfiltered chunks back to back, row-major
            addr = buf.off_from(raw, p)
            if addr is not None:
                counts = [-(-s // c) for s, c in zip(self.shape, cshape)]
                for k, idx in enumerate(np.ndindex(*counts)):
                    place(tuple(i * c for i, c in zip(idx, cshape)), buf.bytes(addr + k * cbytes, cbytes))
        else:
            raise UnsupportedHDF5('%s: chunk index type %d (%s) of libver="latest" files' % (self.name, index, {
                3: 'fixed array', 4: 'extensible array', 5: 'v2 B-tree'}.get(index, 'unknown')))
        return out.tobytes()

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, key):
        a = self.read()
        return a[key] if a.shape or key != () else a[()]


class Group(_Object):
    def __init__(self, buf, addr, name):
        _Object.__init__(self, buf, addr, name)
        self._links = None

    def _entries(self):
        if self._links is None:
            buf = self._buf
            out = []
            stab = self._first(0x11)
            if stab is not None:
                btree, heap = buf.off_from(stab.data, 0), buf.off_from(stab.data, buf.O)
                _walk_group_btree(buf, btree, _local_heap_data(buf, heap), out)
            info = self._first(2)
            if info is not None:
                raw = info.data
                p = 2 + (8 if raw[1] & 1 else 0)
                if buf.off_from(raw, p) is not None:
                    raise UnsupportedHDF5('%s: links in dense storage (fractal heap)' % self.name)
            for m in self._msgs:
                if m.type == 6:
                    out.append(_parse_link(buf, m.data))
            if stab is None:
                out.sort(key=lambda e: e[0])
            self._links = dict(out)
            self._order = [e[0] for e in out]
        return self._links

    def keys(self):
        self._entries()
        return list(self._order)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._entries())

    def __contains__(self, path):
        try:
            self._resolve(path)
            return True
        except KeyError:
            return False

    def _resolve(self, path):
        node = self
        parts = [p for p in path.split('/') if p]
        for i, part in enumerate(parts):
            if not isinstance(node, Group):
                raise KeyError(path)
            links = node._entries()
            if part not in links:
                raise KeyError("Unable to open object (object %r doesn't exist)" % path)
            node = _open(self._buf, links[part], (node.name.rstrip('/') + '/' + part))
        return node

    def __getitem__(self, path):
        return self._resolve(path)

    def items(self):
        return [(k, self[k]) for k in self.keys()]


def _open(buf, addr, name):
    msgs = _read_object_header(buf, addr)
    types = set(m.type for m in msgs)
    if 8 in types and 1 in types and 3 in types:
        return Dataset(buf, addr, name)
    if 0x11 in types or 2 in types or 6 in types or not (types & {1, 3, 8}):
        return Group(buf, addr, name)
    raise UnsupportedHDF5('%s: neither a group nor a dataset (committed datatype?)' % name)


class File(Group):
    """File(path) — read-only.  Raises OSError for a missing file or one without an HDF5 signature, like h5py."""
    def __init__(self, path, mode='r'):
        if mode != 'r':
            raise ValueError('hdf5_min.File is read-only')
        self.filename = path
        self._fh = open(path, 'rb')
        try:
            try:
                self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
            except ValueError:
                raise OSError('Unable to open file (file is empty: %r)' % path)
            buf = _Buf(self._mm)
            pos = 0
            while True:
                if self._mm[pos:pos + 8] == SIGNATURE:
                    break
                pos = 512 if pos == 0 else pos * 2
                if pos + 8 > len(self._mm):
                    raise OSError('Unable to open file (file signature not found: %r)' % path)
            self.userblock_size = pos
            ver = self._mm[pos + 8]
            if ver in (0, 1):
                buf.O, buf.L = self._mm[pos + 13], self._mm[pos + 14]
                p = pos + 24 + (4 if ver == 1 else 0)
                buf.base = 0
                base = buf.off(p)
                # base address: where address 0 of the file's address space lies (the user block's size when there is one)
                buf.base = base if base else 0
                if buf.base == 0 and pos:
                    buf.base = pos                                       # (h5py / MATLAB write base = 0 + user block: offsets
                    #                                                      are counted from the superblock then)
                root = buf.off(p + 4 * buf.O + buf.O)
                eof = buf.u(p + 2 * buf.O, buf.O)                         # stored end-of-file: counts the user block
            elif ver in (2, 3):
                buf.O, buf.L = self._mm[pos + 9], self._mm[pos + 10]
                p = pos + 12
                buf.base = 0
                base = buf.off(p)
                buf.base = base if base else pos
                root = buf.off(p + 3 * buf.O)
                eof = buf.u(p + 2 * buf.O, buf.O)
            else:
                raise UnsupportedHDF5('superblock version %d' % ver)
            if buf.O not in (2, 4, 8) or buf.L not in (2, 4, 8):
                raise ValueError('HDF5: offsets of %d bytes / lengths of %d bytes' % (buf.O, buf.L))
            if eof != (1 << (8 * buf.O)) - 1 and eof > len(self._mm):                  # the library's own check when it opens a file
                raise OSError('Unable to open file (truncated file: eof = %d, stored_eof = %d: %r)' % (len(self._mm), eof, path))
            Group.__init__(self, buf, root, '/')
        except Exception:
            self.close()
            raise

    def close(self):
        mm, fh = getattr(self, '_mm', None), getattr(self, '_fh', None)
        self._mm = self._fh = None
        if mm is not None:
            try:
                mm.close()
            except BufferError:                                          # a view is still alive: the GC closes the map
                pass
        if fh is not None:
            fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def read_with(path, fn, advice=''):
    """fn(open file) through this reader; a file using a part of the format it does not implement (wherever in the file
    that shows) goes through h5py instead when that is installed, else the error names the feature and `advice`."""
    try:
        with File(path) as f:
            return fn(f)
    except UnsupportedHDF5 as e:
        try:
            import h5py
        except ImportError:
            raise UnsupportedHDF5('%s: %s — not read by dsen2_amd.hdf5_min and h5py is not installed%s'
                                  % (path, e, '; ' + advice if advice else '')) from e
        with h5py.File(path, 'r') as f:
            return fn(f)


def _print_tree(group, indent=0, out=None):
    import sys
    out = out or sys.stdout
    pad = '  ' * indent
    for k, v in sorted(group.attrs.items()):
        a = np.asarray(v) if v is not None else None
        text = 'unsupported type' if a is None else (repr(a.tolist()) if a.size <= 6 else '%s%r' % (a.dtype, a.shape))
        out.write('%s@%s = %s\n' % (pad, k, text if len(text) < 100 else text[:97] + '...'))
    for k in group.keys():
        obj = group[k]
        if isinstance(obj, Group):
            out.write('%s%s/\n' % (pad, k))
            _print_tree(obj, indent + 1, out)
        else:
            out.write('%s%s  %s %r\n' % (pad, k, obj.dtype, tuple(obj.shape)))


if __name__ == '__main__':
    # python -m dsen2_amd.hdf5_min FILE : the tree of a checkpoint / .mat (groups, datasets with dtype and shape, attributes)
    import sys
    if len(sys.argv) != 2:
        sys.exit('usage: python -m dsen2_amd.hdf5_min FILE.hdf5|FILE.mat')
    with File(sys.argv[1]) as _f:
        _print_tree(_f)

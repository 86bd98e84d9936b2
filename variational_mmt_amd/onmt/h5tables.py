"""Read-only HDF5 access for the image-feature files of the training driver, without PyTables / h5py / libhdf5.

Replaces, for this path only, the calls `tables.open_file(path, mode='r')`, `file.root.<node>[:]`, `file.close()` at
`train_mm_vi_model1.py:460-481,490-495,510-515` and `translate_mm_vi.py:85-87` (nodes `/global_feats`, `/local_feats`,
`/logits`, `/global_feats_mean`, `/global_feats_stds`, `/array`).  The on-disk format is restated from the published
HDF5 File Format Specification (version 3.0); what is covered is what libhdf5 writes for PyTables `Array` (contiguous),
`CArray` / `EArray` (chunked, version-1 B-tree index; optional shuffle / deflate / fletcher32 filters) and h5py datasets
with default (`earliest`) or `latest`-style object headers:

  superblock v0-v3; object headers v1 and v2 (+ continuation blocks); old-style groups (symbol-table message ->
  v1 B-tree + local heap + SNOD nodes) and compact new-style groups (link messages); dataspace v1/v2; fixed-point and
  IEEE floating-point datatypes of either byte order; data layout v1-v3 and the contiguous / compact classes of v4.

Anything else (dense link storage, v4 chunk indexes, blosc / lzo / szip filters, compound or variable-length types)
raises `HDF5FormatError` naming the construct: no silent approximation.  `read_into()` streams a dataset into a
caller-owned buffer slab by slab so a multi-GB table can go to HBM through a pinned staging buffer (engine.load_image_table).
"""
import os
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class HDF5FormatError(IOError):
    pass


class NoSuchNodeError(AttributeError, KeyError):
    """mirrors tables.NoSuchNodeError: raised for a missing child of a group (attribute or item access)."""
    pass


MSG_DATASPACE, MSG_LINK_INFO, MSG_DATATYPE, MSG_LINK, MSG_LAYOUT, MSG_FILTERS, MSG_CONT, MSG_SYMTAB = 1, 2, 3, 6, 8, 0xB, 0x10, 0x11


class _Raw(object):
    """random access to the file image (memory map: the kernel pages in only what is touched)."""

    def __init__(self, path):
        self.path = path
        self.f = open(path, "rb")
        self.size = os.fstat(self.f.fileno()).st_size
        if self.size < 32:
            raise HDF5FormatError("%s: too short to be an HDF5 file" % path)
        self.mm = np.memmap(self.f, dtype=np.uint8, mode="r")
        self.base = 0
        self.O = 8
        self.L = 8

    def bytes(self, off, n):
        off += self.base
        if off < 0 or off + n > self.size:
            raise HDF5FormatError("%s: read of %d bytes at %d past end of file (%d)" % (self.path, n, off, self.size))
        return self.mm[off:off + n].tobytes()

    def uint(self, off, n):
        return int.from_bytes(self.bytes(off, n), "little")

    def addr(self, off):
        return self.uint(off, self.O)

    def length(self, off):
        return self.uint(off, self.L)

    def undefined(self, a):
        return a == (1 << (8 * self.O)) - 1

    def close(self):
        if self.f is not None:
            del self.mm
            self.f.close()
            self.f = None


def _read_superblock(raw):
    """Spec III.A / II.A: locate the signature at 0, 512, 1024, ...; returns the root object-header address or, for
    versions 0/1, the root symbol-table entry (btree, heap)."""
    off = 0
    while True:
        if off + 8 > raw.size:
            raise HDF5FormatError("%s: no HDF5 signature" % raw.path)
        if raw.mm[off:off + 8].tobytes() == SIGNATURE:
            break
        off = 512 if off == 0 else off * 2
    ver = raw.uint(off + 8, 1)
    if ver in (0, 1):
        raw.O = raw.uint(off + 13, 1)
        raw.L = raw.uint(off + 14, 1)
        p = off + 24 + (4 if ver == 1 else 0)
        raw.base = raw.addr(p)
        p += 4 * raw.O                        # base, free-space info, end of file, driver info
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        root_hdr = raw.addr(p + raw.O)
        cache_type = raw.uint(p + 2 * raw.O, 4)
        scratch = p + 2 * raw.O + 8
        cached = (raw.addr(scratch), raw.addr(scratch + raw.O)) if cache_type == 1 else None
        return root_hdr, cached
    if ver in (2, 3):
        raw.O = raw.uint(off + 9, 1)
        raw.L = raw.uint(off + 10, 1)
        p = off + 12
        raw.base = raw.addr(p)
        return raw.addr(p + 3 * raw.O), None
    raise HDF5FormatError("%s: superblock version %d not supported" % (raw.path, ver))


def _messages(raw, addr):
    """Spec IV.A.1: yields (type, flags, payload bytes) of every header message of the object at `addr`."""
    head = raw.bytes(addr, 4)
    out = []
    if head == b"OHDR":
        ver = raw.uint(addr + 4, 1)
        if ver != 2:
            raise HDF5FormatError("object header version %d" % ver)
        flags = raw.uint(addr + 5, 1)
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        nsz = 1 << (flags & 3)
        size0 = raw.uint(p, nsz)
        p += nsz
        blocks = [(p, size0)]
        tracked = bool(flags & 0x04)
        while blocks:
            b, n = blocks.pop(0)
            q, end = b, b + n
            while q + 4 <= end:
                mtype = raw.uint(q, 1)
                msize = raw.uint(q + 1, 2)
                mflags = raw.uint(q + 3, 1)
                q += 4 + (2 if tracked else 0)
                if q + msize > end:
                    break
                body = raw.bytes(q, msize)
                q += msize
                if mtype == MSG_CONT:
                    ca = int.from_bytes(body[:raw.O], "little")
                    cl = int.from_bytes(body[raw.O:raw.O + raw.L], "little")
                    if raw.bytes(ca, 4) != b"OCHK":
                        raise HDF5FormatError("object header continuation without OCHK signature")
                    blocks.append((ca + 4, cl - 8))          # minus signature and checksum
                elif mtype != 0:
                    out.append((mtype, mflags, body))
        return out
    ver = raw.uint(addr, 1)
    if ver != 1:
        raise HDF5FormatError("object header version %d at %d" % (ver, addr))
    nmsg = raw.uint(addr + 2, 2)
    size0 = raw.uint(addr + 8, 4)
    blocks = [(addr + 16, size0)]
    seen = 0
    while blocks and seen < nmsg:
        b, n = blocks.pop(0)
        q, end = b, b + n
        while q + 8 <= end and seen < nmsg:
            mtype = raw.uint(q, 2)
            msize = raw.uint(q + 2, 2)
            mflags = raw.uint(q + 4, 1)
            body = raw.bytes(q + 8, msize)
            q += 8 + msize
            seen += 1
            if mtype == MSG_CONT:
                ca = int.from_bytes(body[:raw.O], "little")
                cl = int.from_bytes(body[raw.O:raw.O + raw.L], "little")
                blocks.append((ca, cl))
            elif mtype != 0:
                out.append((mtype, mflags, body))
    return out


def _heap_name(raw, heap_addr, offset):
    """Spec III.D local heap: null-terminated name at data segment + offset."""
    if raw.bytes(heap_addr, 4) != b"HEAP":
        raise HDF5FormatError("local heap signature missing")
    seg_size = raw.length(heap_addr + 8)
    seg = raw.addr(heap_addr + 8 + 2 * raw.L)
    n = min(seg_size - offset, 4096)
    s = raw.bytes(seg + offset, n)
    return s.split(b"\0", 1)[0].decode("utf-8")


def _group_btree(raw, node, heap, out):
    """Spec III.A.1 (v1 B-tree, node type 0) + III.C (symbol table node)."""
    sig = raw.bytes(node, 4)
    if sig == b"SNOD":
        n = raw.uint(node + 6, 2)
        p = node + 8
        for _ in range(n):
            name_off = raw.addr(p)
            hdr = raw.addr(p + raw.O)
            out[_heap_name(raw, heap, name_off)] = hdr
            p += 2 * raw.O + 24
        return
    if sig != b"TREE":
        raise HDF5FormatError("group B-tree node signature missing")
    if raw.uint(node + 4, 1) != 0:
        raise HDF5FormatError("group B-tree node of wrong type")
    used = raw.uint(node + 6, 2)
    p = node + 8 + 2 * raw.O
    for i in range(used):
        child = raw.addr(p + raw.L + i * (raw.L + raw.O))
        _group_btree(raw, child, heap, out)


def _parse_link(raw, body):
    """Spec IV.A.2.g link message; returns (name, object header address) for hard links, (name, None) otherwise."""
    flags = body[1]
    p = 2
    ltype = 0
    if flags & 0x08:
        ltype = body[p]
        p += 1
    if flags & 0x04:
        p += 8
    if flags & 0x10:
        p += 1
    nsz = 1 << (flags & 3)
    n = int.from_bytes(body[p:p + nsz], "little")
    p += nsz
    name = body[p:p + n].decode("utf-8")
    p += n
    if ltype != 0:
        return name, None
    return name, int.from_bytes(body[p:p + raw.O], "little")


_FLOAT = {2: "f2", 4: "f4", 8: "f8"}


def _parse_dtype(body):
    """Spec IV.A.2.d: classes 0 (fixed point) and 1 (IEEE floating point)."""
    cls, ver = body[0] & 0x0F, body[0] >> 4
    bits0 = body[1]
    size = struct.unpack_from("<I", body, 4)[0]
    if cls == 0:
        order = ">" if bits0 & 1 else "<"
        kind = "i" if bits0 & 0x08 else "u"
        if size not in (1, 2, 4, 8):
            raise HDF5FormatError("integer of %d bytes" % size)
        return np.dtype(order + kind + str(size))
    if cls == 1:
        if bits0 & 0x40:
            raise HDF5FormatError("VAX-order floating point")
        order = ">" if bits0 & 1 else "<"
        if size not in _FLOAT:
            raise HDF5FormatError("floating point of %d bytes" % size)
        return np.dtype(order + _FLOAT[size])
    raise HDF5FormatError("datatype class %d (version %d) not supported: only integer and floating-point arrays" % (cls, ver))


def _parse_dataspace(raw, body):
    ver = body[0]
    rank = body[1]
    if ver == 1:
        p = 8
    elif ver == 2:
        if body[3] == 2:
            raise HDF5FormatError("null dataspace")
        p = 4
    else:
        raise HDF5FormatError("dataspace version %d" % ver)
    return tuple(int.from_bytes(body[p + i * raw.L:p + (i + 1) * raw.L], "little") for i in range(rank))


def _parse_filters(body):
    """Spec IV.A.2.l: list of (filter id, client data)."""
    ver, n = body[0], body[1]
    out = []
    p = 8 if ver == 1 else 2
    for _ in range(n):
        fid = struct.unpack_from("<H", body, p)[0]
        p += 2
        nlen = 0
        if ver == 1 or fid >= 256:
            nlen = struct.unpack_from("<H", body, p)[0]
            p += 2
        p += 2                                  # flags
        ncd = struct.unpack_from("<H", body, p)[0]
        p += 2
        p += ((nlen + 7) // 8) * 8 if ver == 1 else nlen
        cd = struct.unpack_from("<%dI" % ncd, body, p)
        p += 4 * ncd
        if ver == 1 and ncd % 2:
            p += 4
        out.append((fid, cd))
    return out


def _parse_layout(raw, body):
    """Spec IV.A.2.i: returns ('contiguous', addr, size) | ('compact', bytes) | ('chunked', btree addr, chunk shape, elem size)."""
    ver = body[0]
    O, L = raw.O, raw.L
    if ver in (1, 2):
        ndim, cls = body[1], body[2]
        p = 8
        a = None
        if cls != 0:
            a = int.from_bytes(body[p:p + O], "little")
            p += O
        dims = struct.unpack_from("<%dI" % ndim, body, p)
        p += 4 * ndim
        if cls == 1:
            return ("contiguous", a, None)
        if cls == 2:
            # for chunked storage `ndim` counts one extra dimension: the element size
            return ("chunked", a, tuple(dims[:-1]), dims[-1])
        n = struct.unpack_from("<I", body, p)[0]
        return ("compact", bytes(body[p + 4:p + 4 + n]))
    if ver in (3, 4):
        cls = body[1]
        if cls == 0:
            n = struct.unpack_from("<H", body, 2)[0]
            return ("compact", bytes(body[4:4 + n]))
        if cls == 1:
            return ("contiguous", int.from_bytes(body[2:2 + O], "little"), int.from_bytes(body[2 + O:2 + O + L], "little"))
        if cls == 2 and ver == 3:
            ndim = body[2]
            a = int.from_bytes(body[3:3 + O], "little")
            dims = struct.unpack_from("<%dI" % ndim, body, 3 + O)
            return ("chunked", a, tuple(dims[:-1]), dims[-1])
        raise HDF5FormatError("data layout version %d class %d (v4 chunk indexes / virtual datasets) not supported; "
                              "rewrite the file with the default (earliest) library format" % (ver, cls))
    raise HDF5FormatError("data layout version %d" % ver)


def _unshuffle(buf, itemsize):
    a = np.frombuffer(buf, dtype=np.uint8)
    n = a.size // itemsize
    body = a[:n * itemsize].reshape(itemsize, n).T.reshape(-1)
    if n * itemsize == a.size:
        return body.tobytes()
    return body.tobytes() + a[n * itemsize:].tobytes()


class Array(object):
    """A dataset node: supports `node[:]`, `node[i:j]` (leading axis), `.shape`, `.dtype`, `.read()`, `len()`,
    the subset of `tables.Array` the driver uses."""

    def __init__(self, raw, name, addr):
        self._raw, self.name = raw, name
        self.shape = self.dtype = None
        self._layout = None
        self._filters = []
        for mtype, _fl, body in _messages(raw, addr):
            if mtype == MSG_DATASPACE:
                self.shape = _parse_dataspace(raw, body)
            elif mtype == MSG_DATATYPE:
                self.dtype = _parse_dtype(body)
            elif mtype == MSG_LAYOUT:
                self._layout = _parse_layout(raw, body)
            elif mtype == MSG_FILTERS:
                self._filters = _parse_filters(body)
        if self.shape is None or self.dtype is None or self._layout is None:
            raise HDF5FormatError("node %r is not a dataset" % name)
        for fid, _cd in self._filters:
            if fid not in (1, 2, 3):
                raise HDF5FormatError("node %r uses filter %d (%s): only deflate, shuffle and fletcher32 are supported"
                                      % (name, fid, {4: "szip", 305: "lzo", 307: "bzip2", 32001: "blosc"}.get(fid, "unknown")))
        self.nrows = self.shape[0] if self.shape else 1
        self._chunks = None

    def __len__(self):
        return self.nrows

    @property
    def row_bytes(self):
        n = self.dtype.itemsize
        for s in self.shape[1:]:
            n *= s
        return n

    def _chunk_index(self):
        """Spec III.A.1, node type 1: list of (offsets, stored size, filter mask, address)."""
        if self._chunks is not None:
            return self._chunks
        raw = self._raw
        _k, root, cshape, _es = self._layout
        nd = len(cshape) + 1
        out = []
        if not raw.undefined(root):
            stack = [root]
            while stack:
                node = stack.pop()
                if raw.bytes(node, 4) != b"TREE" or raw.uint(node + 4, 1) != 1:
                    raise HDF5FormatError("chunk B-tree node malformed")
                level = raw.uint(node + 5, 1)
                used = raw.uint(node + 6, 2)
                p = node + 8 + 2 * raw.O
                ksz = 8 + 8 * nd
                blob = raw.bytes(p, used * (ksz + raw.O) + ksz)
                for i in range(used):
                    q = i * (ksz + raw.O)
                    size, mask = struct.unpack_from("<II", blob, q)
                    offs = struct.unpack_from("<%dQ" % nd, blob, q + 8)
                    child = int.from_bytes(blob[q + ksz:q + ksz + raw.O], "little")
                    if level > 0:
                        stack.append(child)
                    else:
                        out.append((offs[:-1], size, mask, child))
        out.sort()
        self._chunks = out
        return out

    def _decode_chunk(self, buf, mask, nbytes):
        for i in range(len(self._filters) - 1, -1, -1):
            if mask & (1 << i):
                continue
            fid, cd = self._filters[i]
            if fid == 1:
                buf = zlib.decompress(buf)
            elif fid == 2:
                buf = _unshuffle(buf, cd[0] if cd else self.dtype.itemsize)
            elif fid == 3:
                buf = buf[:-4]
        if len(buf) < nbytes:
            raise HDF5FormatError("chunk of node %r decodes to %d bytes, expected %d" % (self.name, len(buf), nbytes))
        return buf

    def read_into(self, out, start=0, stop=None):
        """rows [start, stop) of the leading axis into `out` (C-contiguous numpy array of native-order dtype and shape
        [stop-start, *shape[1:]])."""
        raw = self._raw
        stop = self.nrows if stop is None else stop
        if not self.shape:
            raise HDF5FormatError("scalar dataset: use read()")
        n = stop - start
        assert out.shape == (n,) + tuple(self.shape[1:]) and out.flags["C_CONTIGUOUS"]
        kind = self._layout[0]
        if n <= 0:
            return out
        if kind == "contiguous":
            a = self._layout[1]
            if raw.undefined(a):
                out[...] = 0
                return out
            off = raw.base + a + start * self.row_bytes
            cnt = n * self.row_bytes
            if off + cnt > raw.size:
                raise HDF5FormatError("node %r extends past the end of the file" % self.name)
            src = raw.mm[off:off + cnt].view(self.dtype) if (off % self.dtype.itemsize == 0) else \
                np.frombuffer(raw.mm[off:off + cnt].tobytes(), dtype=self.dtype)
            out.reshape(-1)[...] = src
            return out
        if kind == "compact":
            src = np.frombuffer(self._layout[1], dtype=self.dtype).reshape(self.shape)
            out[...] = src[start:stop]
            return out
        _k, _root, cshape, _es = self._layout
        if len(cshape) != len(self.shape):
            raise HDF5FormatError("chunk rank mismatch in node %r" % self.name)
        out[...] = 0                                   # unallocated chunks read as the (zero) fill value
        cbytes = self.dtype.itemsize
        for c in cshape:
            cbytes *= c
        for offs, size, mask, addr in self._chunk_index():
            r0 = offs[0]
            if r0 >= stop or r0 + cshape[0] <= start:
                continue
            buf = raw.bytes(addr, size)
            if self._filters:
                buf = self._decode_chunk(buf, mask, cbytes)
            ch = np.frombuffer(buf, dtype=self.dtype, count=cbytes // self.dtype.itemsize).reshape(cshape)
            src_sl, dst_sl = [], []
            for d, (o, c, s) in enumerate(zip(offs, cshape, self.shape)):
                lo, hi = o, min(o + c, s)
                if d == 0:
                    lo, hi = max(lo, start), min(hi, stop)
                    dst_sl.append(slice(lo - start, hi - start))
                else:
                    dst_sl.append(slice(lo, hi))
                src_sl.append(slice(lo - o, hi - o))
            out[tuple(dst_sl)] = ch[tuple(src_sl)]
        return out

    def read(self, start=None, stop=None):
        if not self.shape:
            kind = self._layout[0]
            if kind == "compact":
                return np.frombuffer(self._layout[1], dtype=self.dtype)[0].astype(self.dtype.newbyteorder("="))
            if kind == "contiguous":
                return np.frombuffer(self._raw.bytes(self._layout[1], self.dtype.itemsize), dtype=self.dtype)[0]
            raise HDF5FormatError("chunked scalar dataset")
        start = 0 if start is None else start
        stop = self.nrows if stop is None else stop
        out = np.empty((max(stop - start, 0),) + tuple(self.shape[1:]), dtype=self.dtype.newbyteorder("="))
        return self.read_into(out, start, stop)

    def __getitem__(self, key):
        if key is Ellipsis or (isinstance(key, tuple) and len(key) == 0):
            return self.read()
        if isinstance(key, slice):
            start, stop, step = key.indices(self.nrows)
            a = self.read(start, max(stop, start)) if step > 0 else self.read()[key]
            return a if step == 1 else (a[::step] if step > 0 else a)
        if isinstance(key, (int, np.integer)):
            k = int(key) + (self.nrows if key < 0 else 0)
            if not 0 <= k < self.nrows:
                raise IndexError("index out of range")
            return self.read(k, k + 1)[0]
        return self.read()[key]

    def __repr__(self):
        return "/%s (Array%r) %s" % (self.name, tuple(self.shape), self.dtype)


class Group(object):
    """A group node: children by attribute (`file.root.global_feats`) or item access."""

    def __init__(self, raw, name, addr, cached=None):
        self.__dict__["_raw"] = raw
        self.__dict__["_name"] = name
        links = {}
        symtab = cached
        dense = False
        for mtype, _fl, body in _messages(raw, addr):
            if mtype == MSG_SYMTAB:
                symtab = (int.from_bytes(body[:raw.O], "little"), int.from_bytes(body[raw.O:2 * raw.O], "little"))
            elif mtype == MSG_LINK:
                nm, a = _parse_link(raw, body)
                if a is not None:
                    links[nm] = a
            elif mtype == MSG_LINK_INFO:
                # version, flags, [max creation index], fractal heap address, name index b-tree address
                p = 2 + (8 if body[1] & 1 else 0)
                heap = int.from_bytes(body[p:p + raw.O], "little")
                dense = not raw.undefined(heap)
        if symtab is not None:
            _group_btree(raw, symtab[0], symtab[1], links)
        elif dense:
            raise HDF5FormatError("group %r stores its links densely (fractal heap): not supported; rewrite the file "
                                  "with the default (earliest) library format" % name)
        self.__dict__["_links"] = links
        self.__dict__["_nodes"] = {}

    def _get(self, name):
        if name in self._nodes:
            return self._nodes[name]
        if name not in self._links:
            raise NoSuchNodeError("group ``/%s`` does not have a child named ``%s``" % (self._name, name))
        addr = self._links[name]
        kinds = set(m[0] for m in _messages(self._raw, addr))
        if MSG_LAYOUT in kinds:
            node = Array(self._raw, name, addr)
        else:
            node = Group(self._raw, name, addr)
        self._nodes[name] = node
        return node

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return self._get(name)

    def __getitem__(self, name):
        return self._get(name)

    def __contains__(self, name):
        return name in self._links

    def __iter__(self):
        return iter(sorted(self._links))

    def _f_list_nodes(self):
        return [self._get(n) for n in sorted(self._links)]


class File(object):
    """What `tables.open_file(path, mode='r')` returns, reduced to `.root`, `.get_node('/name')`, `.close()` and the
    context-manager protocol."""

    def __init__(self, path):
        self.filename = path
        self._raw = _Raw(path)
        try:
            root_hdr, cached = _read_superblock(self._raw)
            self.root = Group(self._raw, "", root_hdr, cached)
        except Exception:
            self._raw.close()
            raise
        self.isopen = True

    def get_node(self, where, name=None):
        parts = [p for p in where.split("/") if p] + ([name] if name else [])
        node = self.root
        for p in parts:
            node = node[p]
        return node

    def close(self):
        if self.isopen:
            self.root = None
            self._raw.close()
            self.isopen = False

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def open_file(filename, mode="r", **kwargs):
    """`tables.open_file` for reading (train_mm_vi_model1.py:460).  Any other mode raises: this module never writes."""
    if mode != "r":
        raise ValueError("h5tables is read-only (mode %r requested)" % (mode,))
    return File(filename)

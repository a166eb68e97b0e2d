"""HDF5 subset reader/writer for Keras `.h5` checkpoints -- pure Python + numpy, no libhdf5 / h5py.

The reference saves and loads its model with Keras' HDF5 format (model.save(save_path), model_training.py:302;
load_model(...), model_training.py:337-338 and Predict.py:51-52).  h5py is not importable by the interpreter
this package runs under, so the part of the public "HDF5 File Format Specification" (version 3.0 of the
document, file-format structures of libhdf5 1.8/1.10) that such files use is restated here:

  reading   superblock 0/1/2/3; object headers v1 and v2 ("OHDR"/"OCHK"); groups as symbol tables
            (v1 B-tree + local heap + "SNOD") or compact link messages; datasets with compact, contiguous or
            chunked (v1 B-tree; deflate / shuffle / fletcher32 filters) layout; fixed-point, floating-point,
            fixed-length string and variable-length string (global heap) datatypes; attribute messages v1-v3.
  writing   superblock 0, v1 object headers, symbol-table groups, contiguous datasets, fixed-length-string and
            numeric attributes -- what h5py writes for Keras with its default (earliest) library bounds.

Not implemented (raises H5Error): dense link / attribute storage (fractal heaps), shared / committed datatypes,
compound, enum, reference and array datatypes, chunked layout message v4, external links.

The interface mirrors the small part of h5py that Keras' hdf5_format.py touches: File(path, mode), group[...],
keys(), attrs, create_group, create_dataset, dataset[()].
"""
import mmap
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
MSG_MAX = 65535                      # an object header message carries a 16-bit size


class H5Error(Exception):
    pass


def _pad8(n):
    return (n + 7) & ~7


# =====================================================================================================
# reading
# =====================================================================================================
class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.so = self.sl = 8
        self.base = 0

    def u8(self, p):
        return self.b[p]

    def u16(self, p):
        return struct.unpack_from("<H", self.b, p)[0]

    def u32(self, p):
        return struct.unpack_from("<I", self.b, p)[0]

    def uint(self, p, n):
        return int.from_bytes(self.b[p:p + n], "little")

    def off(self, p):
        v = self.uint(p, self.so)
        return UNDEF if v == (1 << (8 * self.so)) - 1 else v

    def length(self, p):
        return self.uint(p, self.sl)

    def addr(self, a):
        """file offset of a (base-relative) address"""
        return a + self.base


class _Message:
    __slots__ = ("type", "flags", "pos", "size")

    def __init__(self, type_, flags, pos, size):
        self.type, self.flags, self.pos, self.size = type_, flags, pos, size


def _parse_dataspace(r, p):
    ver, rank, flags = r.u8(p), r.u8(p + 1), r.u8(p + 2)
    if ver == 1:
        q = p + 8
    elif ver == 2:
        if r.u8(p + 3) == 2:
            return None                                    # null dataspace
        q = p + 4
    else:
        raise H5Error(f"dataspace message version {ver}")
    return tuple(r.length(q + i * r.sl) for i in range(rank))


class _Type:
    """kind: 'num' (dtype), 'str' (fixed; size), 'vstr' (variable-length string), 'vseq' unsupported"""

    def __init__(self, kind, size, dtype=None, utf8=False):
        self.kind, self.size, self.dtype, self.utf8 = kind, size, dtype, utf8


def _parse_datatype(r, p):
    cv = r.u8(p)
    cls, ver = cv & 0x0F, cv >> 4
    bf0, bf1 = r.u8(p + 1), r.u8(p + 2)
    size = r.u32(p + 4)
    if ver not in (1, 2, 3):
        raise H5Error(f"datatype message version {ver}")
    if cls == 0:
        order = ">" if bf0 & 1 else "<"
        return _Type("num", size, np.dtype(f"{order}{'i' if bf0 & 8 else 'u'}{size}"))
    if cls == 1:
        if bf0 & 0x40:
            raise H5Error("VAX floating point")
        order = ">" if bf0 & 1 else "<"
        if size not in (2, 4, 8):
            raise H5Error(f"{size}-byte floating point")
        return _Type("num", size, np.dtype(f"{order}f{size}"))
    if cls == 3:
        return _Type("str", size, np.dtype(f"S{size}"), utf8=(bf0 >> 4) == 1)
    if cls == 9:
        if bf0 & 0x0F != 1:
            raise H5Error("variable-length sequences are not supported (only variable-length strings)")
        return _Type("vstr", size, utf8=(bf1 & 0x0F) == 1)
    names = {2: "time", 4: "bitfield", 5: "opaque", 6: "compound", 7: "reference", 8: "enum", 10: "array"}
    raise H5Error(f"datatype class {names.get(cls, cls)} is not supported")


class _Object:
    """An object header: the list of its messages (continuations followed)."""

    def __init__(self, file, address):
        self.file, self.address = file, address
        r = file._r
        if address == UNDEF:
            raise H5Error("undefined object address")
        a = r.addr(address)
        self.msgs = []
        if bytes(r.b[a:a + 4]) == b"OHDR":
            self._parse_v2(r, a)
        else:
            self._parse_v1(r, a)
        self._attrs = None

    def _parse_v1(self, r, a):
        if r.u8(a) != 1:
            raise H5Error(f"object header version {r.u8(a)} at {a}")
        nmsgs, size = r.u16(a + 2), r.u32(a + 8)
        chunks = [(a + 16, size)]
        seen = 0
        while chunks and seen < nmsgs:
            p, size = chunks.pop(0)
            end = p + size
            while p + 8 <= end and seen < nmsgs:
                t, sz, fl = r.u16(p), r.u16(p + 2), r.u8(p + 4)
                seen += 1
                if t == 0x0010:
                    chunks.append((r.addr(r.off(p + 8)), r.length(p + 8 + r.so)))
                elif t != 0:
                    self.msgs.append(_Message(t, fl, p + 8, sz))
                p += 8 + sz

    def _parse_v2(self, r, a):
        if r.u8(a + 4) != 2:
            raise H5Error("object header version")
        flags = r.u8(a + 5)
        p = a + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        w = 1 << (flags & 3)
        size = r.uint(p, w)
        p += w
        chunks = [(p, p + size)]
        hdr = 6 if flags & 0x04 else 4
        while chunks:
            p, end = chunks.pop(0)
            while p + hdr <= end:
                t, sz, fl = r.u8(p), r.u16(p + 1), r.u8(p + 3)
                d = p + hdr
                if t == 0x10:
                    ca, cl = r.addr(r.off(d)), r.length(d + r.so)
                    if bytes(r.b[ca:ca + 4]) != b"OCHK":
                        raise H5Error("bad object header continuation block")
                    chunks.append((ca + 4, ca + cl - 4))
                elif t != 0:
                    self.msgs.append(_Message(t, fl, d, sz))
                p = d + sz

    def find(self, type_):
        return [m for m in self.msgs if m.type == type_]

    # ---- attributes --------------------------------------------------------------------------------
    @property
    def attrs(self):
        if self._attrs is None:
            r = self.file._r
            out = {}
            for m in self.find(0x0015):                       # attribute info: dense storage?
                fl = r.u8(m.pos + 1)
                q = m.pos + 2 + (2 if fl & 1 else 0)
                if r.off(q) != UNDEF:
                    raise H5Error("dense attribute storage (fractal heap) is not supported")
            for m in self.find(0x000C):
                name, val = self._parse_attr(r, m)
                out[name] = val
            self._attrs = out
        return self._attrs

    def _parse_attr(self, r, m):
        p = m.pos
        ver = r.u8(p)
        nsz, tsz, ssz = r.u16(p + 2), r.u16(p + 4), r.u16(p + 6)
        if ver == 1:
            q = p + 8
            step = _pad8
        elif ver in (2, 3):
            if r.u8(p + 1) & 3:
                raise H5Error("attribute with a shared datatype/dataspace")
            q = p + 8 + (1 if ver == 3 else 0)
            step = int
        else:
            raise H5Error(f"attribute message version {ver}")
        name = bytes(r.b[q:q + nsz]).split(b"\0")[0].decode("utf-8")
        q += step(nsz)
        typ = _parse_datatype(r, q)
        q += step(tsz)
        shape = _parse_dataspace(r, q)
        q += step(ssz)
        if shape is None:
            return name, None
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        return name, self.file._decode(typ, shape, bytes(r.b[q:q + count * typ.size]))


class Dataset(_Object):
    def __init__(self, file, address, name):
        super().__init__(file, address)
        self.name = name
        r = file._r
        self.shape = _parse_dataspace(r, self.find(0x0001)[0].pos)
        self._type = _parse_datatype(r, self.find(0x0003)[0].pos)
        self.dtype = self._type.dtype if self._type.kind != "vstr" else np.dtype(object)

    def _filters(self):
        r = self.file._r
        out = []
        for m in self.find(0x000B):
            p = m.pos
            ver, n = r.u8(p), r.u8(p + 1)
            p += 8 if ver == 1 else 2
            for _ in range(n):
                fid = r.u16(p)
                p += 2
                nlen = 0
                if ver == 1 or fid >= 256:
                    nlen = r.u16(p)
                    p += 2
                ncd = r.u16(p + 2)
                p += 4
                p += _pad8(nlen) if ver == 1 else nlen
                cd = [r.u32(p + 4 * i) for i in range(ncd)]
                p += 4 * ncd
                if ver == 1 and ncd % 2:
                    p += 4
                out.append((fid, cd))
        return out

    def _read_raw(self):
        r = self.file._r
        shape = self.shape or ()
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        nbytes = n * self._type.size
        m = self.find(0x0008)
        if not m:
            raise H5Error("dataset without a layout message")
        p = m[0].pos
        ver = r.u8(p)
        if ver not in (3, 4):
            raise H5Error(f"data layout message version {ver} is not supported")
        cls = r.u8(p + 1)
        if ver == 4 and cls >= 2:
            raise H5Error("version-4 chunked / virtual layouts (libver='latest' chunk indexes) are not supported")
        if cls == 0:
            sz = r.u16(p + 2)
            return bytes(r.b[p + 4:p + 4 + sz])[:nbytes]
        if cls == 1:
            a = r.off(p + 2)
            if a == UNDEF:
                return bytes(nbytes)
            a = r.addr(a)
            return bytes(r.b[a:a + nbytes])
        if cls == 2:
            return self._read_chunked(r, p, shape, nbytes)
        raise H5Error(f"data layout class {cls}")

    def _read_chunked(self, r, p, shape, nbytes):
        nd = r.u8(p + 2)
        bt = r.off(p + 3)
        cdims = [r.u32(p + 3 + r.so + 4 * i) for i in range(nd)]
        esz, cdims = cdims[-1], cdims[:-1]
        rank = nd - 1
        if rank != len(shape) or esz != self._type.size:
            raise H5Error("inconsistent chunked layout")
        out = np.zeros(shape, dtype=np.dtype(f"V{esz}"))
        if bt == UNDEF:
            return out.tobytes()
        filters = self._filters()
        chunk_bytes = int(np.prod(cdims, dtype=np.int64)) * esz

        def walk(a):
            a = r.addr(a)
            if bytes(r.b[a:a + 4]) != b"TREE" or r.u8(a + 4) != 1:
                raise H5Error("bad chunk B-tree node")
            level, n = r.u8(a + 5), r.u16(a + 6)
            q = a + 8 + 2 * r.so
            ksz = 8 + 8 * nd
            for _ in range(n):
                csize, mask = r.u32(q), r.u32(q + 4)
                offs = [r.uint(q + 8 + 8 * i, 8) for i in range(rank)]
                child = r.off(q + ksz)
                q += ksz + r.so
                if level > 0:
                    walk(child)
                    continue
                ca = r.addr(child)
                raw = bytes(r.b[ca:ca + csize])
                for i, (fid, cd) in reversed(list(enumerate(filters))):
                    if mask & (1 << i):
                        continue
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:
                        w = cd[0] if cd else esz
                        raw = np.frombuffer(raw, np.uint8).reshape(w, -1).T.tobytes() if w > 1 else raw
                    elif fid == 3:
                        raw = raw[:-4]
                    else:
                        raise H5Error(f"filter {fid} is not supported")
                if len(raw) < chunk_bytes:
                    raise H5Error("short chunk")
                blk = np.frombuffer(raw[:chunk_bytes], dtype=out.dtype).reshape(cdims)
                sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                out[sel] = blk[tuple(slice(0, s.stop - s.start) for s in sel)]
        walk(bt)
        return out.tobytes()

    def read(self):
        if self.shape is None:
            return None
        return self.file._decode(self._type, self.shape, self._read_raw())

    def __getitem__(self, key):
        v = self.read()
        if key is Ellipsis or (isinstance(key, tuple) and len(key) == 0):
            return v
        return v[key]

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self.read())
        return a.astype(dtype) if dtype is not None else a


class Group(_Object):
    def __init__(self, file, address, name):
        super().__init__(file, address)
        self.name = name
        self._links = None

    def _load_links(self):
        if self._links is not None:
            return self._links
        r = self.file._r
        links = {}
        for m in self.find(0x0011):
            self._walk_btree(r, r.off(m.pos), r.off(m.pos + r.so), links)
        for m in self.find(0x0002):
            fl = r.u8(m.pos + 1)
            q = m.pos + 2 + (8 if fl & 1 else 0)
            if r.off(q) != UNDEF:
                raise H5Error("dense link storage (fractal heap) is not supported")
        for m in self.find(0x0006):
            p = m.pos
            if r.u8(p) != 1:
                raise H5Error("link message version")
            fl = r.u8(p + 1)
            p += 2
            ltype = 0
            if fl & 0x08:
                ltype = r.u8(p)
                p += 1
            if fl & 0x04:
                p += 8
            if fl & 0x10:
                p += 1
            w = 1 << (fl & 3)
            nlen = r.uint(p, w)
            p += w
            name = bytes(r.b[p:p + nlen]).decode("utf-8")
            p += nlen
            if ltype == 0:
                links[name] = r.off(p)
        self._links = links
        return links

    def _walk_btree(self, r, bt, heap, links):
        h = r.addr(heap)
        if bytes(r.b[h:h + 4]) != b"HEAP":
            raise H5Error("bad local heap")
        data = r.addr(r.off(h + 8 + 2 * r.sl))

        def name_at(o):
            e = r.b.find(b"\0", data + o)
            return bytes(r.b[data + o:e]).decode("utf-8")

        def walk(a):
            a = r.addr(a)
            if bytes(r.b[a:a + 4]) != b"TREE" or r.u8(a + 4) != 0:
                raise H5Error("bad group B-tree node")
            level, n = r.u8(a + 5), r.u16(a + 6)
            q = a + 8 + 2 * r.so + r.sl                       # first child (after key 0)
            for _ in range(n):
                child = r.off(q)
                q += r.so + r.sl
                if level > 0:
                    walk(child)
                    continue
                s = r.addr(child)
                if bytes(r.b[s:s + 4]) != b"SNOD":
                    raise H5Error("bad symbol table node")
                e = s + 8
                for _ in range(r.u16(s + 6)):
                    if r.u32(e + 2 * r.so) != 2:            # cache type 2 = symbolic link: skipped
                        links[name_at(r.off(e))] = r.off(e + r.so)
                    e += 2 * r.so + 24
        walk(bt)

    def keys(self):
        return list(self._load_links().keys())

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._load_links())

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __getitem__(self, path):
        node = self.file.root if path.startswith("/") else self
        for part in [s for s in path.split("/") if s]:
            if not isinstance(node, Group):
                raise KeyError(path)
            links = node._load_links()
            if part not in links:
                raise KeyError(f"{path!r}: no object {part!r} in {node.name!r}")
            node = self.file._open(links[part], (node.name.rstrip("/") + "/" + part))
        return node


# =====================================================================================================
# writing
# =====================================================================================================
class _WNode:
    def __init__(self):
        self.attrs = {}


class _WDataset(_WNode):
    def __init__(self, data):
        super().__init__()
        a = np.asarray(data)
        if a.dtype.kind == "U":
            a = np.char.encode(a, "utf-8")
        if a.dtype.kind not in "fiuS":
            raise H5Error(f"cannot store dtype {a.dtype}")
        self.data = np.asarray(a.astype(a.dtype.newbyteorder("<") if a.dtype.kind != "S" else a.dtype), order="C")
        self.shape, self.dtype = self.data.shape, self.data.dtype


class _WGroup(_WNode):
    def __init__(self):
        super().__init__()
        self.children = {}

    def _descend(self, path, create):
        node = self
        parts = [s for s in path.split("/") if s]
        for part in parts[:-1]:
            if part not in node.children:
                if not create:
                    raise KeyError(path)
                node.children[part] = _WGroup()
            node = node.children[part]
            if not isinstance(node, _WGroup):
                raise H5Error(f"{part!r} is not a group")
        return node, parts[-1]

    def create_group(self, path):
        parent, name = self._descend(path, True)
        if name in parent.children:
            raise H5Error(f"name already exists: {path}")
        g = parent.children[name] = _WGroup()
        return g

    def require_group(self, path):
        parent, name = self._descend(path, True)
        if name not in parent.children:
            parent.children[name] = _WGroup()
        return parent.children[name]

    def create_dataset(self, path, shape=None, dtype=None, data=None):
        if data is None:
            data = np.zeros(shape, dtype=dtype or np.float32)
        elif dtype is not None:
            data = np.asarray(data, dtype=dtype)
        parent, name = self._descend(path, True)
        if name in parent.children:
            raise H5Error(f"name already exists: {path}")
        d = parent.children[name] = _WDataset(data)
        return d

    def keys(self):
        return list(self.children.keys())

    def __getitem__(self, path):
        parent, name = self._descend(path, False)
        return parent.children[name]

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False


def _dtype_message(dt):
    """datatype message body for a numpy dtype (class + version 1)"""
    dt = np.dtype(dt)
    if dt.kind == "f":
        prop = {2: (15, 10, 5, 0, 10, 15), 4: (31, 23, 8, 0, 23, 127), 8: (63, 52, 11, 0, 52, 1023)}[dt.itemsize]
        sign, eloc, esz, mloc, msz, bias = prop
        return (struct.pack("<BBBBI", 0x11, 0x20, sign, 0, dt.itemsize) +
                struct.pack("<HHBBBBI", 0, 8 * dt.itemsize, eloc, esz, mloc, msz, bias))
    if dt.kind in "iu":
        return (struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize) +
                struct.pack("<HH", 0, 8 * dt.itemsize))
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, max(dt.itemsize, 1))     # null-padded, ASCII
    raise H5Error(f"cannot store dtype {dt}")


def _dataspace_message(shape):
    return struct.pack("<BBBBI", 1, len(shape), 0, 0, 0) + b"".join(struct.pack("<Q", int(s)) for s in shape)


def _attr_array(value):
    """python / numpy attribute value -> little-endian numpy array of a storable dtype"""
    if isinstance(value, str):
        value = value.encode("utf-8")
    if isinstance(value, (bytes, np.bytes_)):
        return np.array(bytes(value) or b"\0", dtype=f"S{max(len(value), 1)}")
    a = np.asarray(value)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf-8")
    if a.dtype.kind == "O":
        a = np.array([x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in a.ravel()]).reshape(a.shape)
    if a.dtype.kind == "b":
        a = a.astype(np.int8)
    if a.dtype.kind == "S" and a.dtype.itemsize == 0:
        a = a.astype("S1")
    if a.dtype.kind not in "fiuS":
        raise H5Error(f"cannot store an attribute of dtype {a.dtype}")
    if a.dtype.kind != "S":
        a = a.astype(a.dtype.newbyteorder("<"))
    return np.asarray(a, order="C")


class _Writer:
    LEAF_K, NODE_K = 4, 16            # libhdf5 defaults: <= 8 symbols per SNOD, <= 32 children per B-tree node

    def __init__(self):
        self.out = bytearray(96)      # superblock 0 with 8-byte offsets and lengths is 96 bytes

    def alloc(self, n):
        pos = _pad8(len(self.out))
        self.out.extend(bytes(pos + n - len(self.out)))
        return pos

    def put(self, pos, data):
        self.out[pos:pos + len(data)] = data

    # ---- object headers ----------------------------------------------------------------------------
    def _attr_messages(self, attrs):
        msgs = []
        for name, value in attrs.items():
            a = _attr_array(value)
            nm = name.encode("utf-8") + b"\0"
            dt, ds = _dtype_message(a.dtype), _dataspace_message(a.shape)
            body = (struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + nm.ljust(_pad8(len(nm)), b"\0") +
                    dt.ljust(_pad8(len(dt)), b"\0") + ds.ljust(_pad8(len(ds)), b"\0") + a.tobytes())
            if len(body) > MSG_MAX - 8:
                raise H5Error(f"attribute {name!r} is larger than 64 KiB: it does not fit an object header message "
                              "(libhdf5 refuses it too in this file-format version)")
            msgs.append((0x000C, body))
        return msgs

    def _object_header(self, msgs):
        body = bytearray()
        for t, data in msgs:
            data = bytes(data).ljust(_pad8(len(data)), b"\0")
            body += struct.pack("<HHBBBB", t, len(data), 0, 0, 0, 0) + data
        pos = self.alloc(16 + len(body))
        self.put(pos, struct.pack("<BBHII", 1, 0, len(msgs), 1, len(body)) + bytes(4) + bytes(body))
        return pos

    def write_dataset(self, d):
        raw = d.data.tobytes()
        if raw:
            addr = self.alloc(len(raw))
            self.put(addr, raw)
        else:
            addr = UNDEF
        msgs = [(0x0001, _dataspace_message(d.shape)), (0x0003, _dtype_message(d.dtype)),
                (0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),          # fill value v2: late alloc, write if set, undefined
                (0x0008, struct.pack("<BBQQ", 3, 1, addr, len(raw)))]
        return self._object_header(msgs + self._attr_messages(d.attrs))

    def write_group(self, g):
        """returns (object header address, B-tree address, local heap address)"""
        entries = []
        for name in sorted(g.children, key=lambda s: s.encode("utf-8")):
            child = g.children[name]
            if isinstance(child, _WGroup):
                entries.append((name.encode("utf-8"),) + self.write_group(child))
            else:
                entries.append((name.encode("utf-8"), self.write_dataset(child), None, None))
        # local heap: "" at offset 0, then the names, 8-byte aligned
        seg = bytearray(8)
        offsets = []
        for name, *_ in entries:
            offsets.append(len(seg))
            seg += name + b"\0"
            seg.extend(bytes(_pad8(len(seg)) - len(seg)))
        heap = self.alloc(32)
        dseg = self.alloc(len(seg))
        self.put(dseg, seg)
        self.put(heap, b"HEAP" + bytes(4) + struct.pack("<QQQ", len(seg), 1, dseg))     # free-list head 1 = none
        # symbol table nodes
        per = 2 * self.LEAF_K
        nodes = []                                              # (address, heap offset of the largest name)
        for i in range(0, len(entries), per):
            part = entries[i:i + per]
            pos = self.alloc(8 + per * 40)
            blob = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(part)))
            for j, (name, hdr, bt, hp) in enumerate(part):
                if bt is None:
                    blob += struct.pack("<QQII", offsets[i + j], hdr, 0, 0) + bytes(16)
                else:
                    blob += struct.pack("<QQIIQQ", offsets[i + j], hdr, 1, 0, bt, hp)
            self.put(pos, blob)
            nodes.append((pos, offsets[i + len(part) - 1]))
        # B-tree over the symbol table nodes
        fan = 2 * self.NODE_K
        node_bytes = 24 + fan * 8 + (fan + 1) * 8
        level = 0
        if not nodes:
            pos = self.alloc(node_bytes)
            self.put(pos, b"TREE" + struct.pack("<BBHQQ", 0, 0, 0, UNDEF, UNDEF) + bytes(8))
            btree = pos
        else:
            while True:
                groups = [nodes[i:i + fan] for i in range(0, len(nodes), fan)]
                addrs = [self.alloc(node_bytes) for _ in groups]
                up = []
                left_key = 0
                for gi, (grp, pos) in enumerate(zip(groups, addrs)):
                    blob = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, level, len(grp),
                                                           addrs[gi - 1] if gi else UNDEF,
                                                           addrs[gi + 1] if gi + 1 < len(addrs) else UNDEF))
                    blob += struct.pack("<Q", left_key)
                    for child, key in grp:
                        blob += struct.pack("<QQ", child, key)
                    left_key = grp[-1][1]
                    self.put(pos, blob)
                    up.append((pos, left_key))
                if len(up) == 1:
                    btree = up[0][0]
                    break
                nodes, level = up, level + 1
        hdr = self._object_header([(0x0011, struct.pack("<QQ", btree, heap))] + self._attr_messages(g.attrs))
        return hdr, btree, heap

    def finish(self, root):
        hdr, btree, heap = self.write_group(root)
        eof = _pad8(len(self.out))
        self.out.extend(bytes(eof - len(self.out)))
        sb = (SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) +
              struct.pack("<HHI", self.LEAF_K, self.NODE_K, 0) + struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF) +
              struct.pack("<QQIIQQ", 0, hdr, 1, 0, btree, heap))
        assert len(sb) == 96
        self.put(0, sb)
        return bytes(self.out)


# =====================================================================================================
class File:
    """File(path, 'r') to read, File(path, 'w') to write (written on close / context exit)."""

    def __init__(self, path, mode="r"):
        self.path, self.mode = path, mode
        if mode == "r":
            self._fh = open(path, "rb")
            self._map = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
            self._r = _Reader(self._map)
            self._cache = {}
            self._gheaps = {}
            self._open_superblock()
        elif mode == "w":
            self._root = _WGroup()
        else:
            raise ValueError("mode must be 'r' or 'w'")

    # ---- reading -----------------------------------------------------------------------------------
    def _open_superblock(self):
        r = self._r
        start = 0
        while bytes(r.b[start:start + 8]) != SIGNATURE:
            start = 512 if start == 0 else start * 2
            if start + 8 > len(r.b):
                raise H5Error(f"{self.path}: not an HDF5 file")
        ver = r.u8(start + 8)
        if ver in (0, 1):
            r.so, r.sl = r.u8(start + 13), r.u8(start + 14)
            p = start + 24 + (4 if ver == 1 else 0)
            r.base = r.off(p)
            p += 4 * r.so
            root = r.off(p + r.so)
        elif ver in (2, 3):
            r.so, r.sl = r.u8(start + 9), r.u8(start + 10)
            p = start + 12
            r.base = r.off(p)
            root = r.off(p + 3 * r.so)
        else:
            raise H5Error(f"superblock version {ver}")
        if r.base == UNDEF:
            r.base = 0
        self.root = Group(self, root, "/")

    def _open(self, address, name):
        if address not in self._cache:
            probe = _Object(self, address)
            if probe.find(0x0008) or probe.find(0x0003):
                node = Dataset(self, address, name)
            else:
                node = Group(self, address, name)
            self._cache[address] = node
        return self._cache[address]

    def _gheap_object(self, address, index):
        r = self._r
        if address not in self._gheaps:
            a = r.addr(address)
            if bytes(r.b[a:a + 4]) != b"GCOL":
                raise H5Error("bad global heap collection")
            end = a + r.length(a + 8)
            p = a + 8 + r.sl
            objs = {}
            while p + 8 + r.sl <= end:
                idx = r.u16(p)
                size = r.length(p + 8)
                if idx == 0:
                    break
                objs[idx] = (p + 8 + r.sl, size)
                p += 8 + r.sl + _pad8(size)
            self._gheaps[address] = objs
        pos, size = self._gheaps[address][index]
        return bytes(r.b[pos:pos + size])

    def _decode(self, typ, shape, raw):
        r = self._r
        if typ.kind == "vstr":
            n = int(np.prod(shape, dtype=np.int64)) if shape else 1
            vals = []
            for i in range(n):
                q = i * typ.size
                ln = struct.unpack_from("<I", raw, q)[0]
                addr = int.from_bytes(raw[q + 4:q + 4 + r.so], "little")
                idx = struct.unpack_from("<I", raw, q + 4 + r.so)[0]
                s = b"" if (ln == 0 or addr == 0) else self._gheap_object(addr, idx)[:ln]
                vals.append(s.decode("utf-8", "surrogateescape"))
            if not shape:
                return vals[0]
            out = np.empty(n, dtype=object)
            out[:] = vals
            return out.reshape(shape)
        a = np.frombuffer(raw, dtype=typ.dtype, count=int(np.prod(shape, dtype=np.int64)) if shape else 1)
        a = a.reshape(shape).copy()
        if typ.kind == "num" and a.dtype.byteorder == ">":
            a = a.astype(a.dtype.newbyteorder("<"))
        if not shape:
            return a[()]
        return a

    # ---- common surface ----------------------------------------------------------------------------
    def _top(self):
        return self.root if self.mode == "r" else self._root

    @property
    def attrs(self):
        return self._top().attrs

    def keys(self):
        return self._top().keys()

    def __getitem__(self, path):
        return self._top()[path]

    def __contains__(self, path):
        return path in self._top()

    def create_group(self, path):
        return self._root.create_group(path)

    def require_group(self, path):
        return self._root.require_group(path)

    def create_dataset(self, path, shape=None, dtype=None, data=None):
        return self._root.create_dataset(path, shape=shape, dtype=dtype, data=data)

    def close(self):
        if self.mode == "r":
            if self._map is not None:
                self._cache.clear()
                self._map.close()
                self._fh.close()
                self._map = None
        elif self._root is not None:
            blob = _Writer().finish(self._root)
            with open(self.path, "wb") as f:
                f.write(blob)
            self._root = None

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None or self.mode == "r":
            self.close()
        return False

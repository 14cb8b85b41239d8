"""A reader for the part of HDF5 a .cool uses -- nothing but numpy, struct and zlib.

cooler writes its files through h5py with the library's default ("earliest") format: version-0 superblock, version-1
object headers, groups as symbol tables (version-1 B-tree + local heap), 1-D datasets that are contiguous or chunked
(version-1 chunk B-tree) with the shuffle and deflate filters, fixed-point / IEEE / fixed-length-string / enum types,
scalar numeric attributes.  That is what is decoded here (HDF5 File Format Specification, version 1.1 / 2.0 subset).
Anything newer -- version-2 object headers ("OHDR"), dense link storage, virtual datasets -- raises Hdf5Unsupported, and
chromosight_amd.io falls back to the HDF5 command line tool for such a file.

Neither cooler nor h5py exists in the target image; the reference reads the same datasets through cooler
(contacts_map.py:129, 209, 529).
"""
import mmap
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Unsupported(Exception):
    pass


class File:
    def __init__(self, path):
        self._fh = open(path, "rb")
        self.buf = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)      # a genome-scale .cool is gigabytes
        b = self.buf
        if b[:8] != b"\x89HDF\r\n\x1a\n":
            raise Hdf5Unsupported("not an HDF5 file (or a user block precedes the superblock)")
        version = b[8]
        if version not in (0, 1):
            raise Hdf5Unsupported(f"superblock version {version}")
        self.so, self.sl = b[13], b[14]                  # size of offsets / lengths
        if (self.so, self.sl) != (8, 8):
            raise Hdf5Unsupported("offsets / lengths that are not 8 bytes")
        pos = 24 if version == 0 else 28                 # v1 adds indexed-storage K + reserved
        self.base, _free, _eof, _drv = struct.unpack_from("<4Q", b, pos)
        pos += 32
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        _name, self.root = struct.unpack_from("<QQ", b, pos)

    def close(self):
        """Release the mapping and the file handle (datasets are returned as copies: nothing refers to the mapping)."""
        if self.buf is not None:
            self.buf.close()
            self.buf = None
        if self._fh is not None:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- object headers ---------------------------------------------------------------------------------------------
    def messages(self, addr):
        """[(type, bytes)] of a version-1 object header, continuation blocks followed."""
        b = self.buf
        a = self.base + addr
        if b[a:a + 4] == b"OHDR":
            raise Hdf5Unsupported("version-2 object header")
        if b[a] != 1:
            raise Hdf5Unsupported(f"object header version {b[a]}")
        n_msg, = struct.unpack_from("<H", b, a + 2)
        size, = struct.unpack_from("<I", b, a + 8)
        blocks = [(a + 16, size)]
        out = []
        while blocks and len(out) < n_msg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < n_msg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, p)
                data = b[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x10:                        # continuation
                    off, length = struct.unpack_from("<QQ", data, 0)
                    blocks.append((self.base + off, length))
                out.append((mtype, data))
        return out

    # ---- groups -----------------------------------------------------------------------------------------------------
    def links(self, addr):
        """{name: object header address} of an old-style group."""
        stab = [d for t, d in self.messages(addr) if t == 0x11]
        if not stab:
            raise Hdf5Unsupported("group without a symbol table (new-style links)")
        btree, heap = struct.unpack_from("<QQ", stab[0], 0)
        b = self.buf
        h = self.base + heap
        if b[h:h + 4] != b"HEAP":
            raise Hdf5Unsupported("bad local heap")
        heap_data = self.base + struct.unpack_from("<Q", b, h + 24)[0]
        out = {}

        def walk(node):
            p = self.base + node
            if b[p:p + 4] == b"SNOD":
                n_sym, = struct.unpack_from("<H", b, p + 6)
                q = p + 8
                for _ in range(n_sym):
                    name_off, obj = struct.unpack_from("<QQ", b, q)
                    s = heap_data + name_off
                    out[b[s:b.find(b"\0", s)].decode()] = obj
                    q += 40
                return
            if b[p:p + 4] != b"TREE" or b[p + 4] != 0:
                raise Hdf5Unsupported("bad group B-tree")
            n_ent, = struct.unpack_from("<H", b, p + 6)
            q = p + 24 + 8                               # after the two sibling pointers, skip key 0
            for _ in range(n_ent):
                child, = struct.unpack_from("<Q", b, q)
                walk(child)
                q += 16                                  # child pointer + next key
        walk(btree)
        return out

    def resolve(self, path):
        addr = self.root
        for part in [p for p in path.split("/") if p]:
            links = self.links(addr)
            if part not in links:
                raise KeyError(path)
            addr = links[part]
        return addr

    def exists(self, path):
        try:
            self.resolve(path)
            return True
        except KeyError:
            return False

    # ---- datatypes / dataspaces -------------------------------------------------------------------------------------
    @staticmethod
    def _dtype(data):
        cls, ver = data[0] & 0x0F, data[0] >> 4
        bits0 = data[1]
        size, = struct.unpack_from("<I", data, 4)
        if ver not in (1, 2, 3):
            raise Hdf5Unsupported(f"datatype version {ver}")
        if cls == 0:                                     # fixed point
            if bits0 & 1:
                raise Hdf5Unsupported("big-endian integers")
            return np.dtype(("<i" if bits0 & 8 else "<u") + str(size)), None
        if cls == 1:                                     # floating point
            if bits0 & 1:
                raise Hdf5Unsupported("big-endian floats")
            return np.dtype("<f" + str(size)), None
        if cls == 3:                                     # fixed-length string
            return np.dtype("S" + str(size)), None
        if cls == 8:                                     # enumeration: base type, names, values
            n, = struct.unpack_from("<H", data, 1)
            if (data[8] & 0x0F) != 0:
                raise Hdf5Unsupported("enumeration over a non-integer type")
            base, _ = File._dtype(data[8:])
            p = 8 + 8 + 4                                # own header, base header, base properties (bit offset, precision)
            names = []
            for _ in range(n):
                e = data.index(b"\0", p)
                names.append(data[p:e].decode())
                length = e + 1 - p
                p += length if ver >= 3 else (length + 7) // 8 * 8    # names are padded to 8 bytes before version 3
            values = np.frombuffer(data, dtype=base, count=n, offset=p)
            return base, dict(zip(values.tolist(), names))
        raise Hdf5Unsupported(f"datatype class {cls}")

    @staticmethod
    def _shape(data):
        ver, rank = data[0], data[1]
        p = 8 if ver == 1 else 4
        return tuple(struct.unpack_from(f"<{rank}Q", data, p)) if rank else ()

    # ---- datasets ---------------------------------------------------------------------------------------------------
    def dataset(self, path):
        msgs = self.messages(self.resolve(path))
        dt = shape = layout = None
        filters = []
        for t, d in msgs:
            if t == 0x03:
                dt, _ = self._dtype(d)
            elif t == 0x01:
                shape = self._shape(d)
            elif t == 0x08:
                layout = d
            elif t == 0x0B:
                filters = self._filters(d)
        if dt is None or shape is None or layout is None:
            raise Hdf5Unsupported(f"{path}: not a simple dataset")
        if len(shape) != 1:
            raise Hdf5Unsupported(f"{path}: rank {len(shape)} (a .cool holds 1-D columns)")
        n = shape[0]
        if layout[0] != 3:
            raise Hdf5Unsupported(f"layout message version {layout[0]}")
        cls = layout[1]
        if cls == 1:                                     # contiguous
            addr, size = struct.unpack_from("<QQ", layout, 2)
            if addr == UNDEF:
                return np.zeros(n, dtype=dt)
            return np.frombuffer(self.buf, dtype=dt, count=n, offset=self.base + addr).copy()
        if cls == 0:                                     # compact
            size, = struct.unpack_from("<H", layout, 2)
            return np.frombuffer(layout, dtype=dt, count=n, offset=4).copy()
        if cls != 2:
            raise Hdf5Unsupported(f"layout class {cls}")
        rank = layout[2]                                 # dataset rank + 1
        btree, = struct.unpack_from("<Q", layout, 3)
        dims = struct.unpack_from(f"<{rank}I", layout, 11)
        chunk = dims[0]
        out = np.zeros(n, dtype=dt)
        if btree == UNDEF:
            return out
        b = self.buf

        def walk(node):
            p = self.base + node
            if b[p:p + 4] != b"TREE" or b[p + 4] != 1:
                raise Hdf5Unsupported("bad chunk B-tree")
            level, n_ent = b[p + 5], struct.unpack_from("<H", b, p + 6)[0]
            q = p + 24
            key = 8 + 8 * rank
            for _ in range(n_ent):
                nbytes, mask = struct.unpack_from("<II", b, q)
                start, = struct.unpack_from("<Q", b, q + 8)
                child, = struct.unpack_from("<Q", b, q + key)
                if level:
                    walk(child)
                else:
                    raw = b[self.base + child:self.base + child + nbytes]
                    for k, (fid, _args) in reversed(list(enumerate(filters))):
                        if mask & (1 << k):
                            continue
                        if fid == 1:
                            raw = zlib.decompress(raw)
                        elif fid == 2:
                            raw = np.frombuffer(raw, dtype=np.uint8).reshape(dt.itemsize, -1).T.tobytes()
                        elif fid == 3:
                            raw = raw[:-4]               # Fletcher-32 checksum
                        else:
                            raise Hdf5Unsupported(f"filter {fid}")
                    vals = np.frombuffer(raw, dtype=dt, count=chunk)
                    m = min(chunk, n - start)
                    if m > 0:
                        out[start:start + m] = vals[:m]
                q += key + 8
        walk(btree)
        return out

    @staticmethod
    def _filters(data):
        ver, n = data[0], data[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid, = struct.unpack_from("<H", data, p)
            if ver == 1 or fid >= 256:
                name_len, _flags, n_cd = struct.unpack_from("<HHH", data, p + 2)
                p += 8 + name_len
            else:
                _flags, n_cd = struct.unpack_from("<HH", data, p + 2)
                p += 6
            args = struct.unpack_from(f"<{n_cd}I", data, p)
            p += 4 * n_cd
            if ver == 1 and n_cd % 2:
                p += 4
            out.append((fid, args))
        return out

    def enum_names(self, path):
        """{value: name} of an enumerated dataset (bins/chrom), or None."""
        for t, d in self.messages(self.resolve(path)):
            if t == 0x03:
                return self._dtype(d)[1]
        return None

    # ---- attributes -------------------------------------------------------------------------------------------------
    def attrs(self, path="/"):
        """Scalar numeric attributes of an object (strings of variable length are skipped)."""
        out = {}
        for t, d in self.messages(self.resolve(path)):
            if t != 0x0C:
                continue
            ver = d[0]
            if ver == 1:
                ns, ts, ss = struct.unpack_from("<HHH", d, 2)
                p = 8
                pad = lambda x: (x + 7) // 8 * 8
            elif ver in (2, 3):
                ns, ts, ss = struct.unpack_from("<HHH", d, 2)
                p = 8 if ver == 2 else 9
                pad = lambda x: x
            else:
                continue
            name = d[p:p + ns].split(b"\0")[0].decode()
            p += pad(ns)
            tdata = d[p:p + ts]
            p += pad(ts)
            sdata = d[p:p + ss]
            p += pad(ss)
            try:
                dt, _ = self._dtype(tdata)
            except Hdf5Unsupported:
                continue
            if dt.kind == "S" or self._shape(sdata) not in ((), (1,)):
                continue
            out[name] = np.frombuffer(d, dtype=dt, count=1, offset=p)[0]
        return out

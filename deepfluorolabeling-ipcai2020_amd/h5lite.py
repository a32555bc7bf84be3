"""Dependency-free HDF5 reader / writer for the files of the reference pipeline (numpy + zlib only).

h5py is absent from the build and GPU images, and the reference's data path IS HDF5: the pre-processed input file
(hdf5_layouts/Readme.md:105-117: '<PP>/projs', '<PP>/segs', '<PP>/lands', 'land-names/num-lands', 'land-names/land-XX',
read by train_test_code/dataset.py:331-446) and the network-output file of test_ensemble.py:121-132 / util.py:300-310
('nn-segs' uint8 and 'nn-heats' float32, gzip level 9, one chunk per image / per heat map, plus the copied 'land-names'
group).  This module implements the subset of the HDF5 file format those files use, with the subset of the h5py API the
reference's scripts call, so that ``dataset.py`` / ``test_ensemble.py`` / ``compute_actual_dice_on_test.py`` /
``est_lands_csv.py`` of this repository open real .h5 files on a box without h5py:

    f = h5lite.File(path, 'r');  f['01/projs'][:], f['01/projs'].shape, f['land-names/num-lands'][()], 'nn-heats' in f
    f = h5lite.File(path, 'w');  g = f.create_group('land-names'); g['num-lands'] = 14; g['land-00'] = 'FH-l'
                                 d = f.create_dataset('nn-segs', (N, H, W), dtype='u1', chunks=(1, H, W),
                                                      compression='gzip', compression_opts=9);  d[i, :, :] = labels

Reader: superblock versions 0-3; object headers version 1 and 2 (with continuation blocks); groups stored as symbol
tables (B-tree v1 + local heap: what libhdf5 / h5py write by default) or as compact link messages; datasets with
compact, contiguous or chunked (B-tree v1 index, layout message v1-v3; v4 single-chunk / implicit) layout; filters
deflate, shuffle, fletcher32; datatypes fixed-point, IEEE float, fixed-length string, variable-length string (global
heap).  Anything else (dense groups in fractal heaps, v4 chunk indexes other than single-chunk / implicit, compound
types, external storage, ...) raises H5Error with the feature named -- nothing is guessed.
Writer: superblock v0, v1 object headers, symbol-table groups, contiguous datasets for scalars / small arrays, chunked
+ deflate datasets streamed to disk chunk by chunk as they are assigned (the 14-heat-map output of a full-size test set
is tens of GB uncompressed: it is never held in memory), strings as variable-length UTF-8 like h5py.

Validated in the build container against h5py 3.3.0 / libhdf5 1.10.6 (tools/gen_h5_fixtures.py: files written by h5py
are committed under tests/golden/h5/ and read here; files written here are read back by h5py and h5dump there).
"""
import struct
import zlib

import numpy as np

__all__ = ['File', 'Group', 'Dataset', 'H5Error', 'is_hdf5']

SIG = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(IOError):
    pass


def is_hdf5(path):
    try:
        with open(path, 'rb') as f:
            return f.read(8) == SIG
    except OSError:
        return False


# ====================================================================================================== reader
class _Reader:
    def __init__(self, path):
        self.f = open(path, 'rb')
        self.path = path
        head = self._at(0, 8)
        if head != SIG:
            raise H5Error('%s is not an HDF5 file (no signature at offset 0; user blocks are not supported)' % path)
        ver = self._at(8, 1)[0]
        if ver in (0, 1):
            b = self._at(8, 16)
            self.O, self.L = b[5], b[6]
            p = 24 + (4 if ver == 1 else 0)
            self.base = self._int(p, self.O)
            p += 4 * self.O                       # base, free-space info, end of file, driver info
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
            self.root_addr = self._int(p + self.O, self.O)
        elif ver in (2, 3):
            self.O, self.L = self._at(9, 1)[0], self._at(10, 1)[0]
            self.base = self._int(12, self.O)
            self.root_addr = self._int(12 + 3 * self.O, self.O)
        else:
            raise H5Error('superblock version %d is not supported' % ver)
        if self.O != 8 or self.L != 8:
            raise H5Error('only 8-byte offsets / lengths are supported (file has %d / %d)' % (self.O, self.L))
        if self.base != 0:
            raise H5Error('non-zero base address is not supported')
        self._gcol = {}

    def close(self):
        self.f.close()

    def _at(self, off, n):
        self.f.seek(off)
        b = self.f.read(n)
        if len(b) != n:
            raise H5Error('truncated file: wanted %d bytes at %d' % (n, off))
        return b

    def _int(self, off, n):
        return int.from_bytes(self._at(off, n), 'little')

    # ---------------------------------------------------------------------------------------- object headers
    def messages(self, addr):
        """[(type, flags, bytes)] of the object header at addr, continuation blocks followed."""
        b = self._at(addr, 16)
        out = []
        if b[:4] == b'OHDR':
            ver, flags = b[4], b[5]
            if ver != 2:
                raise H5Error('object header version %d' % ver)
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szw = 1 << (flags & 3)
            size = self._int(p, szw)
            p += szw
            blocks = [(p, size)]
            order = bool(flags & 0x04)
            while blocks:
                start, n = blocks.pop(0)
                buf = self._at(start, n)
                q = 0
                while q + 4 <= n:
                    mtype, msize, mflags = buf[q], int.from_bytes(buf[q + 1:q + 3], 'little'), buf[q + 3]
                    q += 4 + (2 if order else 0)
                    data = buf[q:q + msize]
                    q += msize
                    if mtype == 0x10:
                        caddr, clen = struct.unpack('<QQ', data[:16])
                        if self._at(caddr, 4) != b'OCHK':
                            raise H5Error('bad continuation block signature')
                        blocks.append((caddr + 4, clen - 8))     # minus signature and checksum
                    elif mtype != 0:
                        out.append((mtype, mflags, bytes(data)))
            return out
        ver = b[0]
        if ver != 1:
            raise H5Error('object header version %d at %d' % (ver, addr))
        nmsg = int.from_bytes(b[2:4], 'little')
        hsize = int.from_bytes(b[8:12], 'little')
        blocks = [(addr + 16, hsize)]
        while blocks and len(out) < nmsg + 64:
            start, n = blocks.pop(0)
            buf = self._at(start, n)
            q = 0
            while q + 8 <= n:
                mtype, msize = struct.unpack('<HH', buf[q:q + 4])
                mflags = buf[q + 4]
                data = buf[q + 8:q + 8 + msize]
                q += 8 + msize
                if mtype == 0x10:
                    caddr, clen = struct.unpack('<QQ', data[:16])
                    blocks.append((caddr, clen))
                elif mtype != 0:
                    out.append((mtype, mflags, bytes(data)))
        return out

    # ---------------------------------------------------------------------------------------- groups
    def links(self, addr):
        """name -> object header address of the group at addr."""
        msgs = self.messages(addr)
        out = {}
        for t, _, d in msgs:
            if t == 0x11:                                        # symbol table: B-tree v1 + local heap
                btree, heap = struct.unpack('<QQ', d[:16])
                hb = self._at(heap, 32)
                if hb[:4] != b'HEAP':
                    raise H5Error('bad local heap signature')
                seg_size, _, seg_addr = struct.unpack('<QQQ', hb[8:32])
                names = self._at(seg_addr, seg_size)
                self._walk_group_btree(btree, names, out)
            elif t == 0x06:                                      # link message (compact storage)
                flags = d[1]
                p = 2
                ltype = 0
                if flags & 0x08:
                    ltype = d[p]
                    p += 1
                if flags & 0x04:
                    p += 8
                if flags & 0x10:
                    p += 1
                w = 1 << (flags & 3)
                n = int.from_bytes(d[p:p + w], 'little')
                p += w
                name = d[p:p + n].decode('utf-8')
                p += n
                if ltype != 0:
                    continue                                     # soft / external links are skipped
                out[name] = int.from_bytes(d[p:p + 8], 'little')
            elif t == 0x02:                                      # link info: dense storage?
                flags = d[1]
                p = 2 + (8 if flags & 1 else 0)
                fheap = int.from_bytes(d[p:p + 8], 'little')
                if fheap != UNDEF:
                    raise H5Error('groups with dense link storage (fractal heap) are not supported')
        return out

    def _walk_group_btree(self, addr, names, out):
        if addr == UNDEF:
            return
        b = self._at(addr, 24)
        if b[:4] == b'SNOD':
            n = int.from_bytes(b[6:8], 'little')
            ents = self._at(addr + 8, n * 40)
            for i in range(n):
                noff, oaddr = struct.unpack('<QQ', ents[i * 40:i * 40 + 16])
                end = names.index(b'\0', noff)
                out[names[noff:end].decode('utf-8')] = oaddr
            return
        if b[:4] != b'TREE' or b[4] != 0:
            raise H5Error('bad group B-tree node at %d' % addr)
        used = int.from_bytes(b[6:8], 'little')
        body = self._at(addr + 24, used * 16 + 8)
        for i in range(used):
            child = int.from_bytes(body[8 + i * 16:16 + i * 16], 'little')
            self._walk_group_btree(child, names, out)

    # ---------------------------------------------------------------------------------------- datasets
    def dtype_of(self, d):
        """numpy dtype (or ('vlen_str',) / ('str', n)) of a datatype message."""
        cls, ver = d[0] & 0x0F, d[0] >> 4
        bits = d[1] | (d[2] << 8) | (d[3] << 16)
        size = int.from_bytes(d[4:8], 'little')
        if cls == 0:
            order = '>' if bits & 1 else '<'
            return np.dtype('%s%s%d' % (order, 'i' if bits & 8 else 'u', size))
        if cls == 1:
            order = '>' if bits & 1 else '<'
            if size not in (2, 4, 8):
                raise H5Error('%d-byte floating point type' % size)
            return np.dtype('%sf%d' % (order, size))
        if cls == 3:
            return ('str', size)
        if cls == 9:
            if (bits & 0x0F) != 1:
                raise H5Error('variable-length sequences are not supported (only variable-length strings)')
            return ('vlen_str',)
        names = {2: 'time', 4: 'bitfield', 5: 'opaque', 6: 'compound', 7: 'reference', 8: 'enum', 10: 'array'}
        raise H5Error('datatype class %s (version %d) is not supported' % (names.get(cls, cls), ver))

    def global_heap_object(self, caddr, index):
        col = self._gcol.get(caddr)
        if col is None:
            h = self._at(caddr, 16)
            if h[:4] != b'GCOL':
                raise H5Error('bad global heap collection signature')
            size = int.from_bytes(h[8:16], 'little')
            buf = self._at(caddr, size)
            col = {}
            p = 16
            while p + 16 <= size:
                idx = int.from_bytes(buf[p:p + 2], 'little')
                osz = int.from_bytes(buf[p + 8:p + 16], 'little')
                if idx == 0:
                    break
                col[idx] = bytes(buf[p + 16:p + 16 + osz])
                p += 16 + (osz + 7) // 8 * 8
            self._gcol[caddr] = col
        return col[index]

    def chunk_index(self, btree, rank):
        """{chunk offset tuple: (address, stored size, filter mask)} from a version-1 raw-data B-tree."""
        out = {}
        if btree == UNDEF:
            return out
        ksize = 8 + 8 * (rank + 1)
        stack = [btree]
        while stack:
            addr = stack.pop()
            b = self._at(addr, 24)
            if b[:4] != b'TREE' or b[4] != 1:
                raise H5Error('bad chunk B-tree node at %d' % addr)
            level, used = b[5], int.from_bytes(b[6:8], 'little')
            body = self._at(addr + 24, used * (ksize + 8) + ksize)
            for i in range(used):
                k = body[i * (ksize + 8):i * (ksize + 8) + ksize]
                child = int.from_bytes(body[i * (ksize + 8) + ksize:(i + 1) * (ksize + 8)], 'little')
                if level > 0:
                    stack.append(child)
                else:
                    csize, mask = struct.unpack('<II', k[:8])
                    offs = struct.unpack('<%dQ' % (rank + 1), k[8:])[:rank]
                    out[tuple(offs)] = (child, csize, mask)
        return out


def _unfilter(raw, filters, mask, itemsize):
    for i, (fid, cdata) in reversed(list(enumerate(filters))):
        if mask & (1 << i):
            continue
        if fid == 1:
            raw = zlib.decompress(raw)
        elif fid == 2:
            n = len(raw) // itemsize
            raw = np.frombuffer(raw, dtype=np.uint8)[:n * itemsize].reshape(itemsize, n).T.tobytes() + raw[n * itemsize:]
        elif fid == 3:
            raw = raw[:-4]
        else:
            raise H5Error('filter id %d is not supported (deflate, shuffle, fletcher32 are)' % fid)
    return raw


class Dataset:
    """Read side: .shape, .dtype, len(), ds[()], ds[:], ds[i], ds[i, :, :], ds[a:b] (unit-stride slices and integers)."""

    def __init__(self, rd, addr, name):
        self._rd, self.name = rd, name
        self.shape = None
        self._layout = None
        self._filters = []
        self._type = None
        for t, _, d in rd.messages(addr):
            if t == 0x01:
                ver, rank = d[0], d[1]
                p = 8 if ver == 1 else 4
                self.shape = tuple(struct.unpack('<%dQ' % rank, d[p:p + 8 * rank])) if rank else ()
                if ver == 2 and d[3] == 2:
                    raise H5Error('null dataspace')
            elif t == 0x03:
                self._type = rd.dtype_of(d)
            elif t == 0x08:
                self._layout = self._parse_layout(d)
            elif t == 0x0B:
                self._filters = self._parse_filters(d)
        if self.shape is None or self._type is None or self._layout is None:
            raise H5Error('%s is not a dataset (dataspace / datatype / layout message missing)' % name)
        self._index = None

    @property
    def dtype(self):
        if isinstance(self._type, np.dtype):
            return self._type.newbyteorder('=')
        return np.dtype('O') if self._type[0] == 'vlen_str' else np.dtype('S%d' % self._type[1])

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        if not self.shape:
            raise TypeError('scalar dataset has no len()')
        return self.shape[0]

    def _itemsize(self):
        if isinstance(self._type, np.dtype):
            return self._type.itemsize
        return self._type[1] if self._type[0] == 'str' else 16

    def _parse_layout(self, d):
        ver = d[0]
        if ver in (1, 2):
            rank, cls = d[1], d[2]
            p = 8
            addr = None
            if cls != 0:
                addr = int.from_bytes(d[p:p + 8], 'little')
                p += 8
            dims = struct.unpack('<%dI' % rank, d[p:p + 4 * rank])
            p += 4 * rank
            if cls == 0:
                n = int.from_bytes(d[p:p + 4], 'little')
                return ('compact', bytes(d[p + 4:p + 4 + n]))
            if cls == 1:
                return ('contiguous', addr, None)
            return ('chunked', addr, tuple(dims[:-1]))
        if ver == 3:
            cls = d[1]
            if cls == 0:
                n = int.from_bytes(d[2:4], 'little')
                return ('compact', bytes(d[4:4 + n]))
            if cls == 1:
                addr, size = struct.unpack('<QQ', d[2:18])
                return ('contiguous', addr, size)
            if cls == 2:
                rank = d[2]
                addr = int.from_bytes(d[3:11], 'little')
                dims = struct.unpack('<%dI' % rank, d[11:11 + 4 * rank])
                return ('chunked', addr, tuple(dims[:-1]))
            raise H5Error('layout class %d' % cls)
        if ver == 4:
            cls = d[1]
            if cls == 0:
                n = int.from_bytes(d[2:4], 'little')
                return ('compact', bytes(d[4:4 + n]))
            if cls == 1:
                addr, size = struct.unpack('<QQ', d[2:18])
                return ('contiguous', addr, size)
            if cls == 2:
                flags, rank, enc = d[2], d[3], d[4]
                dims = tuple(int.from_bytes(d[5 + i * enc:5 + (i + 1) * enc], 'little') for i in range(rank))
                p = 5 + rank * enc
                itype = d[p]
                p += 1
                if itype == 1:                                       # single chunk
                    if flags & 2:
                        fsize = int.from_bytes(d[p:p + 8], 'little')
                        fmask = int.from_bytes(d[p + 8:p + 12], 'little')
                        p += 12
                    else:
                        fsize, fmask = None, 0
                    addr = int.from_bytes(d[p:p + 8], 'little')
                    return ('single', addr, dims[:-1], fsize, fmask)
                if itype == 2:                                       # implicit: chunks back to back, no filters
                    addr = int.from_bytes(d[p:p + 8], 'little')
                    return ('implicit', addr, dims[:-1])
                kinds = {3: 'fixed array', 4: 'extensible array', 5: 'version-2 B-tree'}
                raise H5Error('chunk index type "%s" (layout version 4) is not supported; rewrite the file with the '
                              'default (earliest) library version bounds' % kinds.get(itype, itype))
            raise H5Error('layout class %d (virtual datasets are not supported)' % cls)
        raise H5Error('data layout message version %d' % ver)

    def _parse_filters(self, d):
        ver, n = d[0], d[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = int.from_bytes(d[p:p + 2], 'little')
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = int.from_bytes(d[p:p + 2], 'little')
                p += 2
            p += 2                                                   # flags
            nv = int.from_bytes(d[p:p + 2], 'little')
            p += 2
            p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cd = struct.unpack('<%dI' % nv, d[p:p + 4 * nv])
            p += 4 * nv
            if ver == 1 and nv % 2:
                p += 4
            out.append((fid, cd))
        return out

    # ------------------------------------------------------------------------------------------------ raw access
    def _decode(self, raw, shape):
        n = int(np.prod(shape)) if shape else 1
        if isinstance(self._type, np.dtype):
            a = np.frombuffer(raw, dtype=self._type, count=n).reshape(shape)
            return a.astype(self._type.newbyteorder('='), copy=True)
        if self._type[0] == 'str':
            sz = self._type[1]
            a = np.frombuffer(raw, dtype='S%d' % sz, count=n).reshape(shape)
            return a.copy()
        out = np.empty(n, dtype=object)
        for i in range(n):
            ln, caddr, idx = struct.unpack('<IQI', raw[i * 16:i * 16 + 16])
            out[i] = b'' if caddr == 0 or caddr == UNDEF else self._rd.global_heap_object(caddr, idx)[:ln]
        return out.reshape(shape)

    def _zeros(self, shape):
        if isinstance(self._type, np.dtype):
            return np.zeros(shape, dtype=self._type.newbyteorder('='))
        if self._type[0] == 'str':
            return np.zeros(shape, dtype='S%d' % self._type[1])
        out = np.empty(shape, dtype=object)
        out.fill(b'')
        return out

    def _read_box(self, lo, hi):
        """The hyper-rectangle [lo, hi) as an array."""
        rd, isz = self._rd, self._itemsize()
        shape = tuple(h - l for l, h in zip(lo, hi))
        kind = self._layout[0]
        if kind in ('compact', 'contiguous'):
            if kind == 'compact':
                raw = self._layout[1]
            else:
                addr = self._layout[1]
                n = int(np.prod(self.shape)) if self.shape else 1
                if addr == UNDEF:
                    return self._zeros(shape)
                if self.shape and shape != self.shape and all(l == 0 and h == s for l, h, s in
                                                               zip(lo[1:], hi[1:], self.shape[1:])):
                    row = int(np.prod(self.shape[1:])) * isz         # leading-dimension range: read only those rows
                    return self._decode(rd._at(addr + lo[0] * row, (hi[0] - lo[0]) * row), shape)
                raw = rd._at(addr, n * isz)
            full = self._decode(raw, self.shape)
            return full[tuple(slice(l, h) for l, h in zip(lo, hi))] if self.shape else full
        cdims = self._layout[2]
        out = self._zeros(shape)
        if kind == 'chunked':
            if self._index is None:
                self._index = rd.chunk_index(self._layout[1], len(self.shape))
            index = self._index
        grid = [range(l // c * c, h, c) for l, h, c in zip(lo, hi, cdims)]
        for offs in np.stack(np.meshgrid(*grid, indexing='ij'), -1).reshape(-1, len(cdims)) if cdims else [()]:
            offs = tuple(int(o) for o in offs)
            if kind == 'chunked':
                ent = index.get(offs)
                if ent is None:
                    continue
                raw = _unfilter(rd._at(ent[0], ent[1]), self._filters, ent[2], isz)
            elif kind == 'single':
                _, addr, _, fsize, fmask = self._layout
                nbytes = int(np.prod(cdims)) * isz
                raw = rd._at(addr, fsize if fsize is not None else nbytes)
                if fsize is not None:
                    raw = _unfilter(raw, self._filters, fmask, isz)
            else:                                                    # implicit
                counts = [(s + c - 1) // c for s, c in zip(self.shape, cdims)]
                lin = 0
                for o, c, cnt in zip(offs, cdims, counts):
                    lin = lin * cnt + o // c
                nbytes = int(np.prod(cdims)) * isz
                raw = rd._at(self._layout[1] + lin * nbytes, nbytes)
            chunk = self._decode(raw, cdims)
            src, dst = [], []
            for o, c, l, h in zip(offs, cdims, lo, hi):
                a, b = max(o, l), min(o + c, h)
                src.append(slice(a - o, b - o))
                dst.append(slice(a - l, b - l))
            out[tuple(dst)] = chunk[tuple(src)]
        return out

    def __getitem__(self, key):
        if key is Ellipsis or (isinstance(key, tuple) and len(key) == 0):
            key = ()
        if not isinstance(key, tuple):
            key = (key,)
        if Ellipsis in key:
            i = key.index(Ellipsis)
            key = key[:i] + (slice(None),) * (len(self.shape) - len(key) + 1) + key[i + 1:]
        if len(key) > len(self.shape):
            raise IndexError('too many indices for a dataset of rank %d' % len(self.shape))
        key = key + (slice(None),) * (len(self.shape) - len(key))
        lo, hi, squeeze = [], [], []
        for ax, (k, s) in enumerate(zip(key, self.shape)):
            if isinstance(k, (int, np.integer)):
                k = int(k)
                if k < 0:
                    k += s
                if not 0 <= k < s:
                    raise IndexError('index %d out of range for axis %d of size %d' % (k, ax, s))
                lo.append(k)
                hi.append(k + 1)
                squeeze.append(ax)
            elif isinstance(k, slice):
                a, b, st = k.indices(s)
                if st != 1:
                    raise H5Error('only unit-stride slices are supported')
                lo.append(a)
                hi.append(max(a, b))
            else:
                raise H5Error('unsupported index %r (integers and unit-stride slices)' % (k,))
        out = self._read_box(lo, hi)
        if squeeze:
            out = out.reshape([n for ax, n in enumerate(out.shape) if ax not in squeeze])
        if out.ndim == 0:
            v = out[()]
            if not isinstance(self._type, np.dtype):
                v = bytes(v)
                if self._type[0] == 'str':
                    v = v.split(b'\0', 1)[0]
            return v
        return out


class Group:
    def __init__(self, rd, addr, name):
        self._rd, self._addr, self.name = rd, addr, name
        self._links = None

    def _ls(self):
        if self._links is None:
            self._links = self._rd.links(self._addr)
        return self._links

    def keys(self):
        return sorted(self._ls().keys())

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._ls())

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        parts = [p for p in str(path).split('/') if p]
        if str(path).startswith('/'):
            node = Group(self._rd, self._rd.root_addr, '/')
        for i, part in enumerate(parts):
            if not isinstance(node, Group):
                raise KeyError(path)
            links = node._ls()
            if part not in links:
                raise KeyError("'%s' not found in %s" % (path, self.name))
            addr = links[part]
            name = (node.name.rstrip('/') + '/' + part)
            kinds = {t for t, _, _ in self._rd.messages(addr)}
            node = Dataset(self._rd, addr, name) if 0x08 in kinds else Group(self._rd, addr, name)
        return node


# ====================================================================================================== writer
def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack('<HHB3x', mtype, len(data), flags) + data


def _object_header(msgs):
    body = b''.join(msgs)
    return struct.pack('<BBHII4x', 1, 0, len(msgs), 1, len(body)) + body


def _dataspace(shape):
    if len(shape) == 0:
        return struct.pack('<BBB5x', 1, 0, 0)
    return struct.pack('<BBB5x', 1, len(shape), 0) + struct.pack('<%dQ' % len(shape), *shape)


def _datatype(dt):
    """Datatype message body for a numpy dtype, 'vlen_str', or ('str', n)."""
    if dt == 'vlen_str':
        base = struct.pack('<B3BI', 0x10 | 3, 0x00, 0, 0, 1)                # H5T_C_S1: 1 byte, null-terminated
        return struct.pack('<B3BI', 0x10 | 9, 0x01, 0x01, 0, 16) + base      # type = string, null-terminated, UTF-8
    dt = np.dtype(dt)
    if dt.kind in 'iu':
        bits0 = (0 if dt.byteorder in '<=|' else 1) | (8 if dt.kind == 'i' else 0)
        return struct.pack('<B3BI', 0x10 | 0, bits0, 0, 0, dt.itemsize) + struct.pack('<HH', 0, 8 * dt.itemsize)
    if dt.kind == 'f':
        params = {2: (15, 10, 5, 0, 10, 15), 4: (31, 23, 8, 0, 23, 127), 8: (63, 52, 11, 0, 52, 1023)}[dt.itemsize]
        sign, eloc, esz, mloc, msz, bias = params
        bits0 = (0 if dt.byteorder in '<=|' else 1) | 0x20                   # mantissa normalisation: implied MSB
        return struct.pack('<B3BI', 0x10 | 1, bits0, sign, 0, dt.itemsize) + \
            struct.pack('<HHBBBBI', 0, 8 * dt.itemsize, eloc, esz, mloc, msz, bias)
    if dt.kind == 'S':
        return struct.pack('<B3BI', 0x10 | 3, 0x01, 0, 0, dt.itemsize)       # null-padded ASCII
    raise H5Error('cannot store dtype %s' % dt)


class _WDataset:
    """Write side of a dataset: item assignment with integers / unit-stride slices; chunked datasets stream every
    completed chunk to the file at once."""

    def __init__(self, wf, name, shape, dtype, chunks, level):
        self._wf, self.name = wf, name
        self.shape = tuple(int(s) for s in shape)
        self.dtype = dtype if dtype == 'vlen_str' else np.dtype(dtype)
        self.chunks = None if chunks is None else tuple(int(c) for c in chunks)
        self._level = level
        self._written = {}          # chunk offsets -> (address, stored size)
        self._cache = {}            # chunk offsets -> partially assigned chunk
        self._data = None           # contiguous: whole array / value
        if self.chunks is None and self.dtype != 'vlen_str':
            self._data = np.zeros(self.shape, dtype=self.dtype)

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0]

    def _flush_chunk(self, offs, arr):
        raw = np.ascontiguousarray(arr, dtype=self.dtype.newbyteorder('<')).tobytes()
        if self._level is not None:
            raw = zlib.compress(raw, self._level)
        self._written[offs] = (self._wf._append(raw), len(raw))

    def __setitem__(self, key, value):
        if self.dtype == 'vlen_str':
            raise H5Error('string datasets are written through group[name] = "text"')
        if not isinstance(key, tuple):
            key = (key,)
        if Ellipsis in key:
            i = key.index(Ellipsis)
            key = key[:i] + (slice(None),) * (len(self.shape) - len(key) + 1) + key[i + 1:]
        key = key + (slice(None),) * (len(self.shape) - len(key))
        if self.chunks is None:
            self._data[key] = value
            return
        lo, hi, vshape = [], [], []
        for k, s in zip(key, self.shape):
            if isinstance(k, (int, np.integer)):
                k = int(k) + (s if k < 0 else 0)
                if not 0 <= k < s:
                    raise IndexError('index out of range')
                lo.append(k)
                hi.append(k + 1)
            else:
                a, b, st = k.indices(s)
                if st != 1:
                    raise H5Error('only unit-stride slices are supported')
                lo.append(a)
                hi.append(max(a, b))
                vshape.append(max(a, b) - a)
        box = tuple(h - l for l, h in zip(lo, hi))
        value = np.broadcast_to(np.asarray(value, dtype=self.dtype), vshape).reshape(box)
        grid = [range(l // c * c, h, c) for l, h, c in zip(lo, hi, self.chunks)]
        for offs in np.stack(np.meshgrid(*grid, indexing='ij'), -1).reshape(-1, len(self.chunks)):
            offs = tuple(int(o) for o in offs)
            src, dst, whole = [], [], True
            for o, c, l, h, s in zip(offs, self.chunks, lo, hi, self.shape):
                a, b = max(o, l), min(o + c, h)
                src.append(slice(a - l, b - l))
                dst.append(slice(a - o, b - o))
                whole = whole and a == o and b == min(o + c, s)
            if offs in self._written:
                raise H5Error('%s: chunk %s was already written (chunks are streamed to disk once complete; assign '
                              'whole chunks, or all parts of a chunk before the next one)' % (self.name, offs))
            if whole and offs not in self._cache and all(o + c <= s for o, c, s in zip(offs, self.chunks, self.shape)):
                self._flush_chunk(offs, value[tuple(src)])
                continue
            buf = self._cache.get(offs)
            if buf is None:
                buf = self._cache[offs] = np.zeros(self.chunks, dtype=self.dtype)
            buf[tuple(dst)] = value[tuple(src)]

    def _finish(self):
        for offs, buf in sorted(self._cache.items()):
            self._flush_chunk(offs, buf)
        self._cache = {}


class _WGroup:
    def __init__(self, wf, name):
        self._wf, self.name = wf, name
        self._children = {}

    def create_group(self, name):
        node = self
        for part in [p for p in name.split('/') if p]:
            nxt = node._children.get(part)
            if nxt is None:
                nxt = node._children[part] = _WGroup(self._wf, node.name.rstrip('/') + '/' + part)
            elif not isinstance(nxt, _WGroup):
                raise H5Error('%s exists and is not a group' % part)
            node = nxt
        return node

    def _parent_of(self, name):
        parts = [p for p in name.split('/') if p]
        node = self.create_group('/'.join(parts[:-1])) if len(parts) > 1 else self
        if parts[-1] in node._children:
            raise H5Error('%s already exists' % name)
        return node, parts[-1]

    def create_dataset(self, name, shape=None, dtype='f4', data=None, chunks=None, compression=None,
                       compression_opts=None, **ignored):
        """h5py.Group.create_dataset for the keywords the reference uses (util.py:248-259, 300-310)."""
        node, leaf = self._parent_of(name)
        if data is not None:
            data = np.asarray(data)
            shape = data.shape if shape is None else shape
            dtype = data.dtype if dtype == 'f4' and data.dtype.kind in 'iuf' else dtype
        if chunks is True:
            chunks = (1,) + tuple(shape[1:]) if len(shape) > 1 else tuple(shape)
        level = None
        if compression is not None:
            if compression not in ('gzip', 'deflate') and not isinstance(compression, int):
                raise H5Error("compression %r is not supported (only 'gzip')" % (compression,))
            level = compression if isinstance(compression, int) else (4 if compression_opts is None else int(compression_opts))
            if chunks is None:
                chunks = (1,) + tuple(shape[1:]) if len(shape) > 1 else tuple(shape)
        if chunks is not None and (len(chunks) != len(shape) or any(c <= 0 for c in chunks)):
            raise H5Error('chunk shape %s does not fit dataset shape %s' % (chunks, shape))
        ds = _WDataset(self._wf, node.name.rstrip('/') + '/' + leaf, shape, dtype, chunks, level)
        node._children[leaf] = ds
        if data is not None:
            ds[...] = data
        return ds

    def __setitem__(self, name, value):
        """group[name] = scalar / string / array, as h5py stores them (strings: variable-length UTF-8)."""
        node, leaf = self._parent_of(name)
        if isinstance(value, (str, bytes, np.bytes_, np.str_)):
            ds = _WDataset(self._wf, node.name.rstrip('/') + '/' + leaf, (), 'vlen_str', None, None)
            ds._data = value.encode('utf-8') if isinstance(value, str) else bytes(value)
            node._children[leaf] = ds
            return
        a = np.asarray(value)
        if a.dtype.kind not in 'iuf':
            raise H5Error('cannot store a value of type %s' % a.dtype)
        ds = _WDataset(self._wf, node.name.rstrip('/') + '/' + leaf, a.shape, a.dtype, None, None)
        ds._data[...] = a
        node._children[leaf] = ds

    def __getitem__(self, name):
        node = self
        for part in [p for p in name.split('/') if p]:
            node = node._children[part]
        return node

    def __contains__(self, name):
        try:
            self[name]
            return True
        except (KeyError, AttributeError):
            return False

    def keys(self):
        return sorted(self._children)


GROUP_LEAF_K, GROUP_INTERNAL_K, CHUNK_K = 16, 16, 32     # symbol-table node 2K entries; B-tree nodes 2K children


class _Writer(_WGroup):
    def __init__(self, path):
        super().__init__(self, '/')
        self._f = open(path, 'wb')
        self._pos = 2048                       # superblock + root symbol table entry live in the first block
        self._f.write(b'\0' * self._pos)
        self._closed = False
        self._strings = []

    # ------------------------------------------------------------------------------------------------ low level
    def _append(self, raw, align=8):
        pad = -self._pos % align
        if pad:
            self._f.write(b'\0' * pad)
            self._pos += pad
        addr = self._pos
        self._f.write(raw)
        self._pos += len(raw)
        return addr

    def flush(self):
        if not self._closed:
            self._f.flush()

    # ------------------------------------------------------------------------------------------------ metadata
    def _write_chunk_btree(self, ds):
        rank = len(ds.shape)
        items = sorted(ds._written.items())
        if not items:
            return UNDEF
        ksize = 8 + 8 * (rank + 1)
        node_size = 24 + (2 * CHUNK_K + 1) * ksize + 2 * CHUNK_K * 8

        def key(size, offs):
            return struct.pack('<II', size, 0) + struct.pack('<%dQ' % (rank + 1), *(tuple(offs) + (0,)))
        last = list(items[-1][0])
        last[0] += ds.chunks[0]
        end_key = key(0, last)
        # level 0: entries = (first key, child address)
        level, nodes = 0, [(key(sz, offs), addr) for offs, (addr, sz) in items]
        while True:
            groups = [nodes[i:i + 2 * CHUNK_K] for i in range(0, len(nodes), 2 * CHUNK_K)]
            addrs = [self._append(b'\0' * node_size) for _ in groups]       # reserve, then fill (siblings need addresses)
            out = []
            for gi, grp in enumerate(groups):
                body = b''.join(k + struct.pack('<Q', a) for k, a in grp)
                nxt = groups[gi + 1][0][0] if gi + 1 < len(groups) else end_key
                left = addrs[gi - 1] if gi > 0 else UNDEF
                right = addrs[gi + 1] if gi + 1 < len(groups) else UNDEF
                raw = b'TREE' + struct.pack('<BBHQQ', 1, level, len(grp), left, right) + body + nxt
                self._f.seek(addrs[gi])
                self._f.write(raw)
                out.append((grp[0][0], addrs[gi]))
            self._f.seek(self._pos)
            if len(out) == 1:
                return out[0][1]
            nodes, level = out, level + 1

    def _write_dataset(self, ds):
        ds._finish()
        msgs = [_msg(0x01, _dataspace(ds.shape)), _msg(0x03, _datatype(ds.dtype), flags=1)]
        msgs.append(_msg(0x05, struct.pack('<BBBBI', 2, 3 if ds.chunks else 2, 2, 1, 0)))     # default fill value
        if ds.dtype == 'vlen_str':
            caddr, idx = self._strings_ref(ds._data)
            addr = self._append(struct.pack('<IQI', len(ds._data), caddr, idx))
            msgs.append(_msg(0x08, struct.pack('<BBQQ', 3, 1, addr, 16)))
        elif ds.chunks is None:
            raw = np.ascontiguousarray(ds._data, dtype=ds.dtype.newbyteorder('<')).tobytes()
            addr = self._append(raw) if raw else UNDEF
            msgs.append(_msg(0x08, struct.pack('<BBQQ', 3, 1, addr, len(raw))))
        else:
            if ds._level is not None:
                msgs.append(_msg(0x0B, struct.pack('<BB6x', 1, 1) + struct.pack('<HHHH', 1, 0, 1, 1) +
                                 struct.pack('<II', ds._level, 0)))
            btree = self._write_chunk_btree(ds)
            dims = ds.chunks + (ds.dtype.itemsize,)
            msgs.append(_msg(0x08, struct.pack('<BBB', 3, 2, len(dims)) + struct.pack('<Q', btree) +
                             struct.pack('<%dI' % len(dims), *dims)))
        return self._append(_object_header(msgs))

    def _write_group(self, grp):
        """Children first, then local heap, symbol-table nodes, B-tree, object header.  Returns (header, btree, heap)."""
        entries = []
        for name in sorted(grp._children, key=lambda s: s.encode('utf-8')):
            ch = grp._children[name]
            if isinstance(ch, _WGroup):
                haddr, bt, hp = self._write_group(ch)
                entries.append((name, haddr, 1, struct.pack('<QQ', bt, hp)))
            else:
                entries.append((name, self._write_dataset(ch), 0, b'\0' * 16))
        if len(entries) > 2 * GROUP_LEAF_K * 2 * GROUP_INTERNAL_K:
            raise H5Error('group %s has too many members (%d)' % (grp.name, len(entries)))
        heap = bytearray(b'\0' * 8)                # offset 0: the empty string
        offs = []
        for name, _, _, _ in entries:
            offs.append(len(heap))
            heap += _pad8(name.encode('utf-8') + b'\0')
        free_off = len(heap)
        free_size = max(16, -(len(heap) + 16) % 64 + 16)
        heap += struct.pack('<QQ', 1, free_size) + b'\0' * (free_size - 16)
        seg_addr = self._append(bytes(heap))
        heap_addr = self._append(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), free_off, seg_addr))
        snods = []
        for i in range(0, max(len(entries), 1), 2 * GROUP_LEAF_K):
            part = entries[i:i + 2 * GROUP_LEAF_K]
            raw = b'SNOD' + struct.pack('<BBH', 1, 0, len(part))
            for j, (name, haddr, ctype, scratch) in enumerate(part):
                raw += struct.pack('<QQII', offs[i + j], haddr, ctype, 0) + scratch
            raw += b'\0' * (8 + 2 * GROUP_LEAF_K * 40 - len(raw))
            snods.append((self._append(raw), offs[i + len(part) - 1] if part else 0))
        body = struct.pack('<Q', 0)
        for addr, last_off in snods:
            body += struct.pack('<QQ', addr, last_off)
        node = b'TREE' + struct.pack('<BBHQQ', 0, 0, len(snods), UNDEF, UNDEF) + body
        node += b'\0' * (24 + (2 * GROUP_INTERNAL_K + 1) * 8 + 2 * GROUP_INTERNAL_K * 8 - len(node))
        btree = self._append(node)
        header = self._append(_object_header([_msg(0x11, struct.pack('<QQ', btree, heap_addr))]))
        return header, btree, heap_addr

    def _resolve_strings(self, node):
        """Variable-length strings live in one global heap collection written before the datasets that point to it."""
        for name in sorted(node._children, key=lambda s: s.encode('utf-8')):     # the order _write_group visits them in
            ch = node._children[name]
            if isinstance(ch, _WGroup):
                self._resolve_strings(ch)
            elif ch.dtype == 'vlen_str':
                self._strings.append(ch._data)

    def close(self):
        if self._closed:
            return
        self._strings = []
        self._resolve_strings(self)
        gaddr = None
        if self._strings:
            body = b''
            for i, s in enumerate(self._strings):
                body += struct.pack('<HHIQ', i + 1, 1, 0, len(s)) + _pad8(s)
            size = max(4096, (16 + len(body) + 16 + 4095) // 4096 * 4096)
            free = size - 16 - len(body)
            body += struct.pack('<HHIQ', 0, 0, 0, free) + b'\0' * (free - 16)
            gaddr = self._append(b'GCOL' + struct.pack('<B3xQ', 1, size) + body)
        counter = [0]

        def ref(data):
            counter[0] += 1
            return gaddr, counter[0]
        self._strings_ref = ref
        header, btree, heap = self._write_group(self)
        eof = self._pos
        sb = SIG + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, GROUP_LEAF_K, GROUP_INTERNAL_K, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
        sb += struct.pack('<QQII', 0, header, 1, 0) + struct.pack('<QQ', btree, heap)
        self._f.seek(0)
        self._f.write(sb)
        self._f.close()
        self._closed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class _ReadFile(Group):
    def __init__(self, path):
        rd = _Reader(path)
        super().__init__(rd, rd.root_addr, '/')
        self.filename = path

    def close(self):
        self._rd.close()

    def flush(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def File(path, mode='r'):
    """h5py.File for modes 'r' and 'w'."""
    if mode == 'r':
        return _ReadFile(path)
    if mode == 'w':
        return _Writer(path)
    raise H5Error("mode %r is not supported ('r' and 'w' are)" % (mode,))

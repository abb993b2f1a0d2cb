"""Minimal native HDF5 writer: just enough of the file format to hold what the converter produces.

The reference writes its output through h5py + hdf5plugin (`/root/reference/src/haplohyped/vcf_to_h5.py:131-135`:
`create_dataset(..., compression=32001, compression_opts=(2,2,0,0,5,1,2), chunks=True)`, merged into
`OUT/{cohort}.h5` at :154-180).  Neither library exists in this image, and the chunks are produced on the GPU
already in their final, filtered form — so this module lays the file out itself, the way
`dset.id.write_direct_chunk` would: chunk bytes are appended as they arrive, the metadata (object headers, group
and chunk B-trees, heaps) follows at close, the superblock at offset 0 last.

Format subset (HDF5 File Format Specification, "version 0/1" structures, readable by every libhdf5 >= 1.6):
superblock v0; groups as symbol tables (object header v1 + local heap + v1 B-tree + symbol nodes); datasets with
dataspace v1, datatype v1 (fixed-point, fixed-length string, compound of those), fill value v2, layout v3 (contiguous, or chunked
with a v1 chunk B-tree) and filter pipeline v1.  Filter 32001 is the registered id of the Blosc filter
(hdf5plugin.Blosc / hdf5-blosc): its chunk payload is a Blosc-1 chunk with the 16-byte header — the form this
repository pins bit-exactly against c-blosc 1.21 (oracle/codec_oracle.c, tests/test_oracle_codec.py).

Verified in tests/test_h5file.py with the image's independent libhdf5 1.10.6 (h5py 3.3 under /opt/conda,
when present): structure, dtypes, shapes, chunk index and raw chunk bytes (`read_direct_chunk`).
"""
import os
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b"\x89HDF\r\n\x1a\n"
LEAF_K = 16        # symbol-table node holds up to 2 * LEAF_K entries
GROUP_K = 16       # group B-tree node holds up to 2 * GROUP_K children
CHUNK_K = 32       # chunk B-tree node holds up to 2 * CHUNK_K children (the library's default for superblock v0)
FILTER_BLOSC = 32001


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _datatype(dt):
    dt = np.dtype(dt)
    if dt.kind in "iu":
        bits0 = 0x08 if dt.kind == "i" else 0x00            # little endian, zero padding, signed flag
        return struct.pack("<BBBBI", 0x10, bits0, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, dt.itemsize)   # null-padded ASCII, as numpy 'S'
    if dt.kind == "V" and dt.names:
        # compound, datatype version 1: per member name (padded to 8), byte offset, 28 bytes of (unused) array
        # dimension fields, member datatype.  Packed layouts like the reference's 35-byte record
        # (vcf_to_h5.py:119-127) keep their numpy offsets.
        body = b""
        for name in dt.names:
            mt, moff = dt.fields[name][0], dt.fields[name][1]
            body += _pad8(name.encode() + b"\0") + struct.pack("<IB3xI4x16x", moff, 0, 0) + _datatype(mt)
        return struct.pack("<BHBI", 0x16, len(dt.names), 0, dt.itemsize) + body
    raise TypeError(f"h5file: unsupported dtype {dt}")


def _dataspace(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)


def _object_header(messages, refcount=1):
    body = b"".join(messages)
    return struct.pack("<BBHII4x", 1, 0, len(messages), refcount, len(body)) + body


class H5Writer:
    """Append-only writer.  Usage:
        w = H5Writer(path); addr = w.append(chunk_bytes) ...; w.add_chunked(...); w.add_array(...); w.close()"""

    PAR_MIN = 8 << 20      # appends at least this large are written by PAR_THREADS threads (os.pwrite releases the GIL:
    PAR_THREADS = int(os.environ.get("HHGT_H5_WRITE_THREADS", "8"))   # one thread copies ~2 GB/s into the page cache, and the converter appends 40 MB batches)

    def __init__(self, path):
        self.f = open(path, "wb", buffering=0)   # unbuffered: every write is a pwrite at an address this object keeps
        self.fd = self.f.fileno()
        os.pwrite(self.fd, b"\0" * 2048, 0)    # superblock goes here at close
        self.pos = 2048
        self.groups = {"/": {}}              # group path -> {name: ("group", path) | ("dataset", header_addr)}
        self._pool = None
        if self.PAR_THREADS > 1:             # (made here, not at first use: append and write_at may run on two threads)
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(self.PAR_THREADS)

    # ---- raw space ------------------------------------------------------------------------------------------
    @staticmethod
    def _pwrite_all(fd, mv, pos):
        while len(mv):
            k = os.pwrite(fd, mv, pos)
            mv = mv[k:]
            pos += k

    def append(self, data, align=8):
        pad = -self.pos % align
        if pad:
            os.pwrite(self.fd, b"\0" * pad, self.pos)
            self.pos += pad
        addr = self.pos
        mv = memoryview(data).cast("B")
        n = len(mv)
        if n >= self.PAR_MIN:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(self.PAR_THREADS)
            step = -(-n // self.PAR_THREADS)
            step = -(-step // 4096) * 4096
            for fut in [self._pool.submit(self._pwrite_all, self.fd, mv[o:o + step], addr + o) for o in range(0, n, step)]:
                fut.result()
        else:
            self._pwrite_all(self.fd, mv, addr)
        self.pos += n
        return addr

    def reserve(self, n, align=8):
        """n bytes of the file for a later write_at (their place is fixed now, their bytes may arrive on another thread)"""
        pad = -self.pos % align
        if pad:
            os.pwrite(self.fd, b"\0" * pad, self.pos)
            self.pos += pad
        addr = self.pos
        self.pos += int(n)
        return addr

    def write_at(self, addr, data):
        """the bytes of a reserve()d range; large ranges by PAR_THREADS threads.  Safe beside append() on another thread: the
        two never share a range"""
        mv = memoryview(data).cast("B")
        n = len(mv)
        if n >= self.PAR_MIN and self.PAR_THREADS > 1:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(self.PAR_THREADS)
            step = -(-(-(-n // self.PAR_THREADS)) // 4096) * 4096
            for fut in [self._pool.submit(self._pwrite_all, self.fd, mv[o:o + step], addr + o) for o in range(0, n, step)]:
                fut.result()
        else:
            self._pwrite_all(self.fd, mv, addr)

    def append_file(self, path, align=8):
        """the bytes of the file at `path` (a store's chunks.bin) -> file address; copied inside the kernel where it can
        (os.copy_file_range) by PAR_THREADS threads, through a buffer where it cannot.  An empty file appends nothing (no padding
        either) and returns the current position."""
        n = os.path.getsize(path)
        if n == 0:
            return self.pos
        pad = -self.pos % align
        if pad:
            os.pwrite(self.fd, b"\0" * pad, self.pos)
            self.pos += pad
        addr = self.pos
        src = os.open(path, os.O_RDONLY)

        def piece(o, cnt):
            done = 0
            while done < cnt:
                k = 0
                if hasattr(os, "copy_file_range"):
                    try:
                        k = os.copy_file_range(src, self.fd, cnt - done, o + done, addr + o + done)
                    except OSError:
                        k = 0
                if k <= 0:      # another file system, an old kernel: read + write
                    buf = os.pread(src, min(cnt - done, 16 << 20), o + done)
                    if not buf:
                        raise IOError(f"h5file: {path} ended early")
                    self._pwrite_all(self.fd, memoryview(buf), addr + o + done)
                    k = len(buf)
                done += k

        try:
            if n >= self.PAR_MIN and self.PAR_THREADS > 1:
                if self._pool is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._pool = ThreadPoolExecutor(self.PAR_THREADS)
                step = -(-(-(-n // self.PAR_THREADS)) // 4096) * 4096
                for fut in [self._pool.submit(piece, o, min(step, n - o)) for o in range(0, n, step)]:
                    fut.result()
            else:
                piece(0, n)
        finally:
            os.close(src)
        self.pos += n
        return addr

    # ---- tree of names -----------------------------------------------------------------------------------------
    def _ensure_group(self, path):
        if path in self.groups:
            return
        parent, _, name = path.rstrip("/").rpartition("/")
        parent = parent or "/"
        self._ensure_group(parent)
        self.groups[path] = {}
        self.groups[parent][name] = ("group", path)

    def _link(self, group, name, header_addr):
        group = "/" + group.strip("/") if group.strip("/") else "/"
        self._ensure_group(group)
        if name in self.groups[group]:
            raise ValueError(f"h5file: {group}/{name} exists")
        self.groups[group][name] = ("dataset", header_addr)

    # ---- datasets -------------------------------------------------------------------------------------------------
    def add_array(self, group, name, arr):
        """contiguous, unfiltered dataset holding `arr` (ints or fixed-length byte strings)"""
        arr = np.ascontiguousarray(arr)
        data_addr = self.append(arr.tobytes()) if arr.nbytes else UNDEF
        msgs = [
            _msg(0x0001, _dataspace(arr.shape)),
            _msg(0x0003, _datatype(arr.dtype), flags=1),
            _msg(0x0005, struct.pack("<BBBB", 2, 2, 0, 0)),                          # fill value: late alloc, undefined
            _msg(0x0008, struct.pack("<BBQQ", 3, 1, data_addr, arr.nbytes)),          # layout v3, contiguous
        ]
        self._link(group, name, self.append(_object_header(msgs)))

    def add_chunked(self, group, name, shape, dtype, chunk_shape, chunks, filter_id=None, cd_values=(), filter_name=b"",
                    aliases=()):
        """chunks: iterable of (offsets tuple in elements, file address, stored size in bytes[, filter mask]); every
        chunk of the grid must be present (the converter always writes full grids).  aliases: further names in the
        same group for the same object (hard links)."""
        dt = np.dtype(dtype)
        rank = len(shape)
        assert len(chunk_shape) == rank
        ents = sorted((tuple(int(o) for o in c[0]) + (0,), int(c[1]), int(c[2]), int(c[3]) if len(c) > 3 else 0) for c in chunks)
        upper = tuple(-(-int(shape[0]) // int(chunk_shape[0])) * int(chunk_shape[0]) if i == 0 else 0 for i in range(rank)) + (0,)
        btree = self._chunk_btree(ents, rank, upper)
        msgs = [
            _msg(0x0001, _dataspace(shape)),
            _msg(0x0003, _datatype(dt), flags=1),
            _msg(0x0005, struct.pack("<BBBB", 2, 3, 0, 0)),                          # fill value: incremental alloc
        ]
        if filter_id is not None:
            nm = _pad8(filter_name + b"\0") if filter_name else b""
            cd = b"".join(struct.pack("<I", int(v) & 0xFFFFFFFF) for v in cd_values)
            if len(cd_values) % 2:
                cd += b"\0\0\0\0"
            msgs.append(_msg(0x000B, struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", filter_id, len(nm), 0, len(cd_values)) + nm + cd,
                             flags=1))
        lay = struct.pack("<BBBQ", 3, 2, rank + 1, btree) + b"".join(struct.pack("<I", int(c)) for c in chunk_shape) + \
            struct.pack("<I", dt.itemsize)
        msgs.append(_msg(0x0008, lay))
        hdr = self.append(_object_header(msgs, refcount=1 + len(aliases)))
        for nm in (name,) + tuple(aliases):
            self._link(group, nm, hdr)

    def _chunk_btree(self, ents, rank, upper):
        """v1 B-tree, node type 1 (raw data chunks): key = (stored size u32, filter mask u32, offsets u64 x (rank+1))"""
        if not ents:
            return UNDEF

        def key(off, size=0, mask=0):
            return struct.pack("<II", size, mask) + b"".join(struct.pack("<Q", o) for o in off)

        keysize = 8 + 8 * (rank + 1)
        node_bytes = 24 + (2 * CHUNK_K + 1) * keysize + 2 * CHUNK_K * 8
        # level 0: (first key offsets, first key bytes, child address) per chunk
        level = 0
        items = [(e[0], key(e[0], e[2], e[3]), e[1]) for e in ents]
        while True:
            groups = [items[i:i + 2 * CHUNK_K] for i in range(0, len(items), 2 * CHUNK_K)]
            # nodes of one level are laid out back to back so that sibling addresses are known up front
            pad = -self.pos % 8
            base = self.pos + pad
            addrs = [base + i * node_bytes for i in range(len(groups))]
            out = bytearray(b"\0" * pad)
            nxt = []
            for gi, g in enumerate(groups):
                last = groups[gi + 1][0][0] if gi + 1 < len(groups) else upper
                body = b"".join(k + struct.pack("<Q", a) for _, k, a in g) + key(last)
                node = b"TREE" + struct.pack("<BBHQQ", 1, level, len(g), addrs[gi - 1] if gi else UNDEF,
                                             addrs[gi + 1] if gi + 1 < len(groups) else UNDEF) + body
                out += node + b"\0" * (node_bytes - len(node))
                nxt.append((g[0][0], g[0][1], addrs[gi]))
            self._pwrite_all(self.fd, memoryview(bytes(out)), self.pos)
            self.pos += len(out)
            if len(groups) == 1:
                return addrs[0]
            items, level = nxt, level + 1

    # ---- groups ---------------------------------------------------------------------------------------------------
    def _write_group(self, path):
        """-> (object header address, btree address, heap address) of the group at `path` (children first)"""
        entries = []
        for name, (kind, ref) in self.groups[path].items():
            if kind == "group":
                hdr, bt, hp = self._write_group(ref)
                entries.append((name.encode(), hdr, 1, bt, hp))
            else:
                entries.append((name.encode(), ref, 0, 0, 0))
        entries.sort(key=lambda e: e[0])
        # local heap: offset 0 = "" (the B-tree's leftmost key), then the names
        heap = bytearray(b"\0" * 8)
        offs = []
        for nm, *_ in entries:
            offs.append(len(heap))
            heap += _pad8(nm + b"\0")
        heap_data = self.append(bytes(heap))
        heap_addr = self.append(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, heap_data))   # free list: H5HL_FREE_NULL
        # symbol nodes of up to 2 * LEAF_K entries each
        snods = []
        for i in range(0, max(len(entries), 1), 2 * LEAF_K):
            part = list(zip(entries[i:i + 2 * LEAF_K], offs[i:i + 2 * LEAF_K]))
            body = b"".join(struct.pack("<QQI4xQQ", off, e[1], e[2], e[3] if e[2] else 0, e[4] if e[2] else 0) for e, off in part)
            node = b"SNOD" + struct.pack("<BBH", 1, 0, len(part)) + body
            node += b"\0" * (8 + 2 * LEAF_K * 40 - len(node))
            snods.append((self.append(node), part[-1][1] if part else 0))
        if len(snods) > 2 * GROUP_K:
            raise ValueError(f"h5file: group {path} has too many entries for a single-level group B-tree")
        body = struct.pack("<Q", 0) + b"".join(struct.pack("<QQ", a, last_off) for a, last_off in snods)
        node = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF) + body
        node += b"\0" * (24 + (2 * GROUP_K + 1) * 8 + 2 * GROUP_K * 8 - len(node))
        bt_addr = self.append(node)
        hdr = self.append(_object_header([_msg(0x0011, struct.pack("<QQ", bt_addr, heap_addr))]))
        return hdr, bt_addr, heap_addr

    def close(self):
        hdr, bt, hp = self._write_group("/")
        eof = self.pos
        sb = SIG + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", LEAF_K, GROUP_K, 0) + \
            struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF) + struct.pack("<QQI4xQQ", 0, hdr, 1, bt, hp)
        os.pwrite(self.fd, sb, 0)
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
        self.f.close()
        self.f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.f is not None:
            self.close()


def blosc_cd_values(typesize, chunk_nbytes, clevel=5, shuffle=1, compcode=1):
    """client data of filter 32001 as hdf5-blosc's set_local leaves it: [filter revision 2, Blosc format 2, typesize,
    uncompressed chunk bytes, clevel, shuffle, compressor (1 = LZ4; the reference passes 2 = LZ4HC — same block
    format, and the header of each chunk says which decoder family it needs)]"""
    return (2, 2, int(typesize), int(chunk_nbytes), int(clevel), int(shuffle), int(compcode))


class H5Reader:
    """Reader for the same subset (files written by H5Writer, and h5py files that stay inside it: superblock v0/v1,
    symbol-table groups, object headers v1 without continuation blocks in use, layout v3).  Gives dataset metadata,
    contiguous arrays and the chunk index (address, stored size) — chunk payloads are decoded on the GPU by their
    consumer, not here."""

    def __init__(self, path):
        self.path = path
        self.f = open(path, "rb")
        sb = self._at(0, 96)
        if sb[:8] != SIG or sb[8] > 1 or sb[13] != 8 or sb[14] != 8:
            raise ValueError(f"{path}: not an HDF5 file of the supported kind (superblock v0/v1, 8-byte offsets)")
        root = 56 if sb[8] == 0 else 60                 # v1 carries 4 more bytes (indexed-storage K) before the addresses
        self.root_header = struct.unpack_from("<Q", self._at(root, 40), 8)[0]
        self._groups = {}

    def close(self):
        self.f.close()

    def _at(self, addr, n):
        self.f.seek(addr)
        return self.f.read(n)

    def _messages(self, addr):
        ver, _, nmsgs, _, size = struct.unpack("<BBHII", self._at(addr, 12))
        if ver != 1:
            raise ValueError(f"{self.path}: object header version {ver} at {addr} is outside the supported subset")
        blocks, out = [(addr + 16, size)], []
        while blocks and len(out) < nmsgs:
            a, n = blocks.pop(0)
            body, p = self._at(a, n), 0
            while p + 8 <= n and len(out) < nmsgs:
                mtype, msize, _ = struct.unpack_from("<HHB", body, p)
                data = body[p + 8:p + 8 + msize]
                if mtype == 0x0010:                      # continuation block
                    blocks.append(struct.unpack("<QQ", data[:16]))
                out.append((mtype, data))
                p += 8 + msize
        return out

    def _heap_string(self, heap_addr, off):
        h = self._at(heap_addr, 32)
        assert h[:4] == b"HEAP"
        data_addr = struct.unpack_from("<Q", h, 24)[0]
        s = self._at(data_addr + off, 256)
        return s[:s.index(b"\0")].decode()

    def group(self, header_addr=None):
        """-> {name: object header address} of the group whose object header is at header_addr (default: root)"""
        header_addr = self.root_header if header_addr is None else header_addr
        if header_addr in self._groups:
            return self._groups[header_addr]
        st = [d for t, d in self._messages(header_addr) if t == 0x0011]
        if not st:
            raise KeyError("not a group")
        bt, heap = struct.unpack("<QQ", st[0][:16])
        out = {}

        def walk(addr):
            node = self._at(addr, 24)
            assert node[:4] == b"TREE" and node[4] == 0
            level, used = node[5], struct.unpack_from("<H", node, 6)[0]
            body = self._at(addr + 24, 8 + used * 16)
            for i in range(used):
                child = struct.unpack_from("<Q", body, 8 + i * 16)[0]
                if level:
                    walk(child)
                else:
                    sn = self._at(child, 8)
                    assert sn[:4] == b"SNOD"
                    n = struct.unpack_from("<H", sn, 6)[0]
                    ents = self._at(child + 8, n * 40)
                    for k in range(n):
                        noff, hdr = struct.unpack_from("<QQ", ents, k * 40)
                        out[self._heap_string(heap, noff)] = hdr

        walk(bt)
        self._groups[header_addr] = out
        return out

    def resolve(self, path):
        addr = self.root_header
        for part in [p for p in path.split("/") if p]:
            addr = self.group(addr)[part]
        return addr

    @staticmethod
    def _dtype(b):
        cls, ver = b[0] & 0x0F, b[0] >> 4
        size = struct.unpack_from("<I", b, 4)[0]
        if cls == 0:
            return np.dtype(("i" if b[1] & 0x08 else "u") + str(size)), 12
        if cls == 3:
            return np.dtype(f"S{size}"), 8
        if cls == 6 and ver == 1:
            n = struct.unpack_from("<H", b, 1)[0]
            p, names, fmts, offs = 8, [], [], []
            for _ in range(n):
                e = b.index(b"\0", p)
                names.append(b[p:e].decode())
                p += (e - p + 8) // 8 * 8
                offs.append(struct.unpack_from("<I", b, p)[0])
                p += 32
                mt, used = H5Reader._dtype(b[p:])
                fmts.append(mt)
                p += used
            return np.dtype(dict(names=names, formats=fmts, offsets=offs, itemsize=size)), p
        raise ValueError(f"datatype class {cls} version {ver} is outside the supported subset")

    def dataset(self, path):
        """-> dict(shape, dtype, layout='contiguous'|'chunked', address/size or chunk_shape + chunks {offsets: (addr, size, mask)},
        filters [(id, cd_values)])"""
        info = dict(filters=[])
        for t, d in self._messages(self.resolve(path)):
            if t == 0x0001:
                rank = d[1]
                info["shape"] = tuple(struct.unpack_from(f"<{rank}Q", d, 8)) if rank else ()
            elif t == 0x0003:
                info["dtype"] = self._dtype(d)[0]
            elif t == 0x000B:
                p = 8
                for _ in range(d[1]):
                    fid, nlen, _, ncd = struct.unpack_from("<HHHH", d, p)
                    p += 8 + nlen
                    info["filters"].append((fid, tuple(struct.unpack_from(f"<{ncd}I", d, p))))
                    p += 4 * (ncd + ncd % 2)
            elif t == 0x0008:
                if d[0] != 3:
                    raise ValueError("data layout version outside the supported subset")
                if d[1] == 1:
                    info.update(layout="contiguous", address=struct.unpack_from("<Q", d, 2)[0], size=struct.unpack_from("<Q", d, 10)[0])
                elif d[1] == 2:
                    nd = d[2]
                    dims = struct.unpack_from(f"<{nd}I", d, 11)
                    info.update(layout="chunked", btree=struct.unpack_from("<Q", d, 3)[0], chunk_shape=tuple(dims[:-1]))
                else:
                    raise ValueError("compact layout is outside the supported subset")
        if info.get("layout") == "chunked":
            info["chunks"] = self._chunk_index(info["btree"], len(info["chunk_shape"]))
        return info

    def _chunk_index(self, addr, rank):
        out = {}
        if addr == UNDEF:
            return out
        keysize = 8 + 8 * (rank + 1)

        def walk(a):
            node = self._at(a, 24)
            assert node[:4] == b"TREE" and node[4] == 1
            level, used = node[5], struct.unpack_from("<H", node, 6)[0]
            body = self._at(a + 24, used * (keysize + 8))
            for i in range(used):
                p = i * (keysize + 8)
                size, mask = struct.unpack_from("<II", body, p)
                offs = struct.unpack_from(f"<{rank}Q", body, p + 8)
                child = struct.unpack_from("<Q", body, p + keysize)[0]
                if level:
                    walk(child)
                else:
                    out[tuple(int(o) for o in offs)] = (child, size, mask)

        walk(addr)
        return out

    def read_array(self, path):
        info = self.dataset(path)
        if info["layout"] != "contiguous":
            raise ValueError(f"{path}: chunked datasets are read chunk by chunk (read_chunk)")
        n = int(np.prod(info["shape"])) if info["shape"] else 1
        if info["size"] == 0 or n == 0:
            return np.zeros(info["shape"], info["dtype"])
        return np.frombuffer(self._at(info["address"], info["size"]), dtype=info["dtype"]).reshape(info["shape"]).copy()

    def read_chunk(self, info, offsets):
        addr, size, _ = info["chunks"][tuple(offsets)]
        return np.frombuffer(self._at(addr, size), dtype=np.uint8)

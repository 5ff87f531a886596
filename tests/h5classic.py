"""Independent pure-Python reader of classic-format HDF5 (superblock v0/v1, v1 object headers,
symbol-table groups, contiguous f64 datasets) -- TEST INFRASTRUCTURE.  Written from the HDF5 File
Format Specification separately from csrc/h5lite.cc, so that the files the library writes are parsed
by a second implementation (there is no h5py / libhdf5 in this image)."""
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class File:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        assert self.b[:8] == SIG, "signature"
        ver = self.b[8]
        assert ver in (0, 1), ver
        assert self.b[9] == 0 and self.b[10] == 0 and self.b[12] == 0, "free-space / root entry / shared header versions"
        assert self.b[13] == 8 and self.b[14] == 8
        self.leaf_k, self.internal_k = struct.unpack_from("<HH", self.b, 16)
        o = 4 if ver == 1 else 0
        self.base, free, self.eof, drv = struct.unpack_from("<QQQQ", self.b, 24 + o)
        assert free == UNDEF and drv == UNDEF and self.base == 0
        assert self.eof == len(self.b), (self.eof, len(self.b))
        name_off, root_oh, cache, _res = struct.unpack_from("<QQII", self.b, 56 + o)
        self.root_cache = (cache, struct.unpack_from("<QQ", self.b, 56 + o + 24))
        self.datasets = {}
        self._group(root_oh, "")

    def _messages(self, oh):
        ver, _r, nmsg, refs, size = struct.unpack_from("<BBHII", self.b, oh)
        assert ver == 1 and refs >= 1
        chunks, out, seen = [(oh + 16, size)], [], 0
        for at, sz in chunks:
            p = at
            while p + 8 <= at + sz and seen < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", self.b, p)
                assert msize % 8 == 0, "version-1 messages are padded to 8 bytes"
                data = self.b[p + 8:p + 8 + msize]
                seen += 1
                if mtype == 0x10:
                    chunks.append(struct.unpack_from("<QQ", data))
                elif mtype != 0:
                    out.append((mtype, data))
                p += 8 + msize
        assert seen == nmsg
        return out

    def _group(self, oh, prefix):
        msgs = dict(self._messages(oh))
        btree, heap = struct.unpack_from("<QQ", msgs[0x11])
        assert self.b[heap:heap + 4] == b"HEAP" and self.b[heap + 4] == 0
        seg_size, free_head, seg = struct.unpack_from("<QQQ", self.b, heap + 8)
        assert free_head == 1 or free_head + 16 <= seg_size
        self._btree(btree, seg, prefix)

    def _btree(self, node, seg, prefix):
        assert self.b[node:node + 4] == b"TREE"
        ntype, level, used = struct.unpack_from("<BBH", self.b, node + 4)
        assert ntype == 0
        keys_children = node + 24
        prev = ""
        for i in range(used):
            child, = struct.unpack_from("<Q", self.b, keys_children + 8 + 16 * i)
            if level > 0:
                self._btree(child, seg, prefix)
                continue
            assert self.b[child:child + 4] == b"SNOD" and self.b[child + 4] == 1
            nsym, = struct.unpack_from("<H", self.b, child + 6)
            assert nsym <= 2 * self.leaf_k
            for k in range(nsym):
                name_off, oh, cache, _ = struct.unpack_from("<QQII", self.b, child + 8 + 40 * k)
                end = self.b.index(b"\0", seg + name_off)
                name = self.b[seg + name_off:end].decode()
                assert name > prev, "symbol table entries must be sorted by name"
                prev = name
                self._object(oh, prefix + name)
            # key i+1 = largest name in child i
            key1, = struct.unpack_from("<Q", self.b, keys_children + 16 * (i + 1))
            end = self.b.index(b"\0", seg + key1)
            assert self.b[seg + key1:end].decode() == prev

    def _object(self, oh, path):
        msgs = self._messages(oh)
        kinds = [m[0] for m in msgs]
        if 0x11 in kinds:
            return self._group(oh, path + "/")
        d = dict(msgs)
        sp = d[0x01]
        assert sp[0] == 1, "dataspace version 1"
        rank = sp[1]
        dims = struct.unpack_from("<%dQ" % rank, sp, 8)
        dt = d[0x03]
        if dt[0] == 0x10:   # fixed point, version 1: H5T_STD_U64LE (a Rust usize such as `num_save`)
            assert dt[1] == 0 and dt[2] == 0 and struct.unpack_from("<I", dt, 4)[0] == 8, "unsigned little-endian, 8 bytes"
            assert tuple(dt[8:12]) == (0, 0, 64, 0), "bit offset 0, precision 64"
            np_type = "<u8"
        else:
            assert dt[0] == 0x11 and dt[1] == 0x20 and dt[2] == 0x3f and struct.unpack_from("<I", dt, 4)[0] == 8
            assert tuple(dt[8:20]) == (0, 0, 64, 0, 52, 11, 0, 52, 0xff, 0x03, 0, 0), "IEEE binary64 properties"
            np_type = "<f8"
        lay = d[0x08]
        assert lay[0] == 3 and lay[1] == 1, "layout v3, contiguous"
        addr, size = struct.unpack_from("<QQ", lay, 2)
        assert size == 8 * int(np.prod(dims)) and addr % 8 == 0
        self.datasets[path] = np.frombuffer(self.b, dtype=np_type, count=int(np.prod(dims)), offset=addr).reshape(dims)

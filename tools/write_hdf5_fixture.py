"""Writes small HDF5 files byte by byte, in the layout h5py's default (libver='earliest') gives a hickle dump of a numpy array
(hkl.dump(arr, path, mode='w', compression='gzip'), src/download_and_predict_job.py:462-463, :592-633): superblock v0, old-style
root group (object header v1 + symbol-table message, group B-tree v1, local heap, symbol-table node), one dataset per name with
object header v1 = dataspace v1 + datatype + data layout v3 (chunked, chunk B-tree v1 with leaf and internal nodes, or
contiguous) + filter pipeline v1 (deflate, optionally shuffle).  Written from the HDF5 File Format Specification, with zlib for
the chunks -- no h5py / hickle exists in this image, so these files pin the reader (csrc/hickle.hip) against an independent
WRITER of the same specification, not against real hickle output.

    python tools/write_hdf5_fixture.py          # -> tests/golden/hkl/*.hkl (deterministic)
"""
import os
import struct
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNDEF = 0xFFFFFFFFFFFFFFFF


def pad8(b):
    return b + b"\0" * (-len(b) % 8)


def msg(mtype, data, flags=0):
    data = pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def object_header(messages):
    body = b"".join(messages)
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body


def datatype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind in "iu":
        bits = 0x08 if dt.kind == "i" else 0
        return msg(0x3, struct.pack("<BBBBI", 0x10 | 0, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize))
    assert dt.kind == "f"
    if dt.itemsize == 4:
        props = struct.pack("<HHBBBBII", 0, 32, 23, 8, 0, 23, 127, 0)[:12]
        bits = (0x20, 0x1F, 0)
    else:
        props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        bits = (0x20, 0x3F, 0)
    return msg(0x3, struct.pack("<BBBBI", 0x10 | 1, bits[0], bits[1], bits[2], dt.itemsize) + props)


def dataspace_msg(shape):
    return msg(0x1, struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", d) for d in shape))


class Writer:
    def __init__(self):
        self.buf = bytearray(96)                    # superblock placeholder

    def alloc(self, data):
        self.buf += b"\0" * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def chunked_dataset(self, arr, chunks, shuffle=False, leaf_fanout=6):
        arr = np.ascontiguousarray(arr)
        R, es = arr.ndim, arr.dtype.itemsize
        grid = [-(-arr.shape[k] // chunks[k]) for k in range(R)]
        entries = []                                # (key bytes, child address)
        for idx in np.ndindex(*grid):
            off = [idx[k] * chunks[k] for k in range(R)]
            block = np.zeros(chunks, arr.dtype)
            sl = tuple(slice(off[k], min(off[k] + chunks[k], arr.shape[k])) for k in range(R))
            block[tuple(slice(0, s.stop - s.start) for s in sl)] = arr[sl]
            raw = block.tobytes()
            if shuffle:
                raw = np.frombuffer(raw, np.uint8).reshape(-1, es).T.tobytes()
            comp = zlib.compress(raw, 4)
            addr = self.alloc(comp)
            key = struct.pack("<II", len(comp), 0) + b"".join(struct.pack("<Q", o) for o in off) + struct.pack("<Q", 0)
            entries.append((key, addr))
        last_key = struct.pack("<II", 0, 0) + b"".join(struct.pack("<Q", s) for s in arr.shape) + struct.pack("<Q", 0)

        def node(level, ents):
            body = b"".join(k + struct.pack("<Q", a) for k, a in ents) + last_key
            full = 64 * 8 + 65 * len(last_key)      # libhdf5 reads whole nodes: 2K children + 2K + 1 keys, K = 32 for chunk trees
            return self.alloc(b"TREE" + struct.pack("<BBHQQ", 1, level, len(ents), UNDEF, UNDEF) + body + b"\0" * (full - len(body)))
        if len(entries) <= leaf_fanout:
            root = node(0, entries)
        else:                                       # two levels: leaves of `leaf_fanout` chunks under one internal node
            leaves = []
            for i in range(0, len(entries), leaf_fanout):
                part = entries[i:i + leaf_fanout]
                leaves.append((part[0][0], node(0, part)))
            root = node(1, leaves)
        layout = struct.pack("<BBB", 3, 2, R + 1) + struct.pack("<Q", root) + b"".join(struct.pack("<I", c) for c in chunks) + struct.pack("<I", es)
        filt = b""
        n = 0
        if shuffle:
            filt += struct.pack("<HHHH", 2, 0, 1, 1) + struct.pack("<I", es) + b"\0" * 4
            n += 1
        filt += struct.pack("<HHHH", 1, 0, 1, 1) + struct.pack("<I", 4) + b"\0" * 4
        n += 1
        pipeline = struct.pack("<BB6x", 1, n) + filt
        return self.alloc(object_header([dataspace_msg(arr.shape), datatype_msg(arr.dtype), msg(0x8, layout), msg(0xB, pipeline)]))

    def contiguous_dataset(self, arr, continuation=False):
        arr = np.ascontiguousarray(arr)
        data = self.alloc(arr.tobytes())
        layout = msg(0x8, struct.pack("<BBQQ", 3, 1, data, arr.nbytes))
        if not continuation:
            return self.alloc(object_header([dataspace_msg(arr.shape), datatype_msg(arr.dtype), layout]))
        # the layout message lives in a continuation block (0x0010), as happens when attributes are added later
        cont = self.alloc(layout)
        head = [dataspace_msg(arr.shape), datatype_msg(arr.dtype), msg(0x10, struct.pack("<QQ", cont, len(layout)))]
        body = b"".join(head)
        return self.alloc(struct.pack("<BxHII4x", 1, len(head) + 1, 1, len(body)) + body)

    def finish(self, links, path):
        names = sorted(links)
        heap_data = bytearray(b"\0" * 8)
        offs = {}
        for nme in names:
            offs[nme] = len(heap_data)
            heap_data += pad8(nme.encode() + b"\0")
        heap_data += b"\0" * 64
        hdata = self.alloc(bytes(heap_data))
        # free-list head: libhdf5 writes 1 (H5HL_FREE_NULL) for "no free block" and rejects anything else that is not an
        # offset inside the data segment (round-3 files carried the undefined address here and h5ls refused them)
        heap = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, hdata))
        snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
        for nme in names:
            snod += struct.pack("<QQII16x", offs[nme], links[nme], 0, 0)
        assert len(names) <= 8                      # one symbol-table node: 2 x leaf K (4) entries of 40 bytes, stored at full size
        snod_addr = self.alloc(snod + b"\0" * (8 + 8 * 40 - len(snod)))
        btree = self.alloc(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_addr, offs[names[-1]])
                           + b"\0" * (32 * 8 + 33 * 8 - 24))          # full node for internal K = 16
        root = self.alloc(object_header([msg(0x11, struct.pack("<QQ", btree, heap))]))
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBxBBBx", 0, 0, 0, 0, 8, 8) + struct.pack("<HHI", 4, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == 96
        self.buf[:96] = sb
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(self.buf)


def fixtures(seed=0):
    """name -> (dataset name in the file, array, how it is stored)"""
    rng = np.random.default_rng(seed)
    smooth = (np.add.outer(np.arange(37), np.arange(29))[None, :, :, None] * 120 + rng.integers(0, 500, (5, 37, 29, 4))).astype(np.uint16)
    return {
        "s2_10_u16": ("data", smooth, dict(chunks=(2, 16, 16, 4))),                                   # hickle 4/5 name, edge chunks
        "clouds_f32": ("data_0", rng.random((3, 50, 41)).astype(np.float32), dict(chunks=(1, 25, 41), shuffle=True)),   # hickle 3 name
        "dates_i64": ("data", np.array([5, 40, 100, 160, 220, 280, 340], np.int64), None),          # contiguous
        "s1_u16_many_chunks": ("data", rng.integers(0, 65535, (12, 40, 40, 2)).astype(np.uint16), dict(chunks=(1, 16, 16, 2))),
        "dem_f32_contig": ("data", rng.random((31, 17)).astype(np.float32), "continuation"),
    }


def main():
    out = os.path.join(ROOT, "tests", "golden", "hkl")
    for fname, (dname, arr, how) in fixtures().items():
        w = Writer()
        if isinstance(how, dict):
            oh = w.chunked_dataset(arr, **how)
        else:
            oh = w.contiguous_dataset(arr, continuation=(how == "continuation"))
        links = {dname: oh}
        if fname == "clouds_f32":                    # a second object next to it, like hickle 3's companions
            links["aux"] = w.contiguous_dataset(np.arange(4, dtype=np.int32))
        w.finish(links, os.path.join(out, fname + ".hkl"))
        print(fname, arr.shape, arr.dtype, os.path.getsize(os.path.join(out, fname + ".hkl")), "bytes")


if __name__ == "__main__":
    main()

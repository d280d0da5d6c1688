"""A minimal HDF5 writer / reader for the driver loop's result file.

The reference stores its images as one HDF5 dataset ``matrix[num_images, rows, cols]`` of f32
(/root/reference/data/src/hdf5.rs:36-63) and reads them back through libhdf5
(``Reader``, :76-131; ``data-to-pics``).  libhdf5 / h5py are not available in the build image, so
this module writes the container itself, from the HDF5 File Format Specification, using only the
oldest (version 0 / 1) structures, which every libhdf5 release reads:

    superblock v0 -> root group (object header v1: Symbol Table message)
                       -> group B-tree v1 node -> symbol table node "SNOD" -> local heap "HEAP" (names)
                   -> dataset object header v1: Dataspace v1, Datatype (IEEE f32 LE), Fill Value v2,
                      Data Layout v3

The dataset is chunked ``[1, rows, cols]`` like the reference's (:47): Data Layout v3 class 2 with a
version-1 chunk B-tree (node type 1, 64 entries per node as implied by a version-0 superblock, as
many levels as the image count needs).  The chunks themselves are laid out back to back, so the
whole stack is also one contiguous f32 block that can be memory-mapped.  ``layout="contiguous"``
writes layout class 1 instead (no B-tree).

VALIDATION: libhdf5 is not a dependency of this package, but an HDF5 1.10.6 installation was found
under /opt/conda in the build image (the survey had not found it).  ``tests/test_hdf5_min.py`` reads
these files back with that library (``h5dump`` and the C API through ctypes) and compares the
metadata byte layout with a file the library wrote itself; without it the tests fall back to
``read`` below, an independent strict parser.
"""
from __future__ import annotations

import struct
from typing import Tuple

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"
GROUP_LEAF_K, GROUP_INTERNAL_K = 4, 16            # the library's defaults
CHUNK_K = 32                                      # indexed-storage B-tree K implied by superblock v0
CHUNK_KEY = 8 + 4 * 8                             # chunk size, filter mask, 3 + 1 offsets
CHUNK_NODE = 24 + (2 * CHUNK_K + 1) * CHUNK_KEY + 2 * CHUNK_K * 8
DATA_ALIGN = 4096                                 # the raw data starts on a page boundary
F32LE_DATATYPE = bytes([0x11, 0x20, 0x1F, 0x00]) + struct.pack("<I", 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _message(msg_type: int, data: bytes, flags: int = 0) -> bytes:
    data = _pad8(data)
    return struct.pack("<HHB3x", msg_type, len(data), flags) + data


def _object_header(messages) -> bytes:
    body = b"".join(messages)
    # version 1, reserved, number of messages, reference count, header data size, 4 bytes of padding
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


def _chunk_tree(first_addr: int, data_addr: int, shape: Tuple[int, int, int]) -> Tuple[bytes, int]:
    """Version-1 B-tree (node type 1) over the chunks [i, 0, 0] -> data_addr + i * plane bytes, as
    nodes of 2K = 64 entries laid out from ``first_addr``; returns (node bytes, root address)."""
    n, rows, cols = shape
    plane = rows * cols * 4

    def key(i: int, size: int) -> bytes:
        return struct.pack("<II4Q", size, 0, i, 0, 0, 0)

    end_key = struct.pack("<II4Q", 0, 0, max(n - 1, 0), rows, cols, 4)     # what the library writes as the last key
    # level 0: (first chunk index, child address) per chunk; upper levels: per node of the level below
    entries = [(i, data_addr + i * plane) for i in range(n)]
    blob, addr, level = b"", first_addr, 0
    while True:
        groups = [entries[i:i + 2 * CHUNK_K] for i in range(0, max(len(entries), 1), 2 * CHUNK_K)]
        addrs = [addr + k * CHUNK_NODE for k in range(len(groups))]
        for k, group in enumerate(groups):
            left = addrs[k - 1] if k > 0 else UNDEF
            right = addrs[k + 1] if k + 1 < len(groups) else UNDEF
            node = b"TREE" + struct.pack("<BBHQQ", 1, level, len(group), left, right)
            for first, child in group:
                node += key(first, plane) + struct.pack("<Q", child)
            # closing key: the first key of the right sibling's subtree, or the end key
            node += key(groups[k + 1][0][0], plane) if k + 1 < len(groups) else end_key
            blob += node + b"\0" * (CHUNK_NODE - len(node))
        addr += len(groups) * CHUNK_NODE
        if len(groups) == 1:
            return blob, addrs[0]
        entries = [(group[0][0], a) for group, a in zip(groups, addrs)]
        level += 1


def _layout(name: bytes, shape: Tuple[int, int, int], chunked: bool):
    """All metadata blocks and their addresses; returns (bytes of the metadata region, data address)."""
    n, rows, cols = shape
    name_z = _pad8(name + b"\0")
    heap_data = b"\0" * 8 + name_z                 # offset 0: "" (root), offset 8: the dataset's name
    sizes = {
        "super": 96,
        "root": 16 + 8 + 16,                       # object header prefix + one Symbol Table message
        "btree": 24 + (2 * GROUP_INTERNAL_K + 1) * 8 + 2 * GROUP_INTERNAL_K * 8,
        "heap": 32,
        "heapdata": len(heap_data),
        "snod": 8 + 2 * GROUP_LEAF_K * 40,
    }
    addr, pos = {}, 0
    for key in ("super", "root", "btree", "heap", "heapdata", "snod"):
        addr[key] = pos
        pos += sizes[key] + (-sizes[key] % 8)
    addr["dset"] = pos

    def dataset_header(data_addr: int, tree_root: int) -> bytes:
        dims = struct.pack("<3Q", n, rows, cols)
        dataspace = struct.pack("<BBB5x", 1, 3, 1) + dims + dims          # v1, rank 3, max dims present
        if chunked:
            fill = bytes([2, 3, 2, 1, 0, 0, 0, 0])                         # v2, incremental, if-set, default value
            layout = struct.pack("<BBBQ4I", 3, 2, 4, tree_root, 1, rows, cols, 4)   # v3, chunked, rank + 1
        else:
            fill = bytes([2, 2, 2, 1, 0, 0, 0, 0])                         # v2, late allocation, if-set, default value
            layout = struct.pack("<BBQQ", 3, 1, data_addr, n * rows * cols * 4)     # v3, contiguous
        return _object_header([_message(0x0001, dataspace), _message(0x0003, F32LE_DATATYPE, 1),
                               _message(0x0005, fill, 1), _message(0x0008, layout)])

    dset_len = len(dataset_header(0, 0))
    tree_addr = addr["dset"] + dset_len
    if chunked:
        if max(rows, cols) >= 1 << 32 or rows * cols == 0:
            raise ValueError("chunk dimensions must be in 1 .. 2^32 - 1")
        tree_len = len(_chunk_tree(tree_addr, 0, shape)[0])
    else:
        tree_len = 0
    data_addr = (tree_addr + tree_len + DATA_ALIGN - 1) // DATA_ALIGN * DATA_ALIGN
    eof = data_addr + n * rows * cols * 4
    tree, tree_root = _chunk_tree(tree_addr, data_addr, shape) if chunked else (b"", 0)

    root_entry = struct.pack("<QQII", 0, addr["root"], 1, 0) + struct.pack("<QQ", addr["btree"], addr["heap"])
    superblock = (SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", GROUP_LEAF_K, GROUP_INTERNAL_K, 0)
                  + struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF) + root_entry)
    assert len(superblock) == sizes["super"]
    root = _object_header([_message(0x0011, struct.pack("<QQ", addr["btree"], addr["heap"]))])
    assert len(root) == sizes["root"]
    # group B-tree: one leaf-level node, one child; keys are heap offsets of names:
    # key[0] = "" < every name in child 0 <= key[1] = the largest name in it
    btree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, addr["snod"], 8)
    btree += b"\0" * (sizes["btree"] - len(btree))
    heap = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, addr["heapdata"])   # free list: 1 = none
    snod = b"SNOD" + struct.pack("<BBH", 1, 0, 1) + struct.pack("<QQII16x", 8, addr["dset"], 0, 0)
    snod += b"\0" * (sizes["snod"] - len(snod))

    blob = bytearray(data_addr)
    for where, block in ((addr["super"], superblock), (addr["root"], root), (addr["btree"], btree),
                         (addr["heap"], heap), (addr["heapdata"], heap_data), (addr["snod"], snod),
                         (addr["dset"], dataset_header(data_addr, tree_root)), (tree_addr, tree)):
        blob[where:where + len(block)] = block
    return bytes(blob), data_addr


def create(path: str, shape: Tuple[int, int, int], dataset: str = "matrix", layout: str = "chunked") -> np.memmap:
    """Create ``path`` holding one f32 dataset of ``shape`` = [images, rows, cols]; returns a writable
    memmap of its data (``out[i] = image`` stores image i; ``flush()`` when done)."""
    n, rows, cols = (int(x) for x in shape)
    if min(n, rows, cols) < 0:
        raise ValueError("negative dimension")
    if layout not in ("chunked", "contiguous"):
        raise ValueError("layout must be 'chunked' or 'contiguous'")
    meta, data_addr = _layout(dataset.encode(), (n, rows, cols), layout == "chunked" and rows * cols > 0)
    total = data_addr + n * rows * cols * 4
    with open(path, "wb") as f:
        f.write(meta)
        if total > len(meta):
            f.truncate(total)
    if n * rows * cols == 0:
        return np.zeros((n, rows, cols), np.float32)      # nothing to map
    return np.memmap(path, dtype="<f4", mode="r+", offset=data_addr, shape=(n, rows, cols))


class FormatError(ValueError):
    pass


def read(path: str, dataset: str = "matrix") -> np.ndarray:
    """Strict parser of the version-0/1 structures ``create`` writes -- and libhdf5 writes by default:
    one f32 dataset of the root group, contiguous or in unfiltered ``[1, rows, cols]`` chunks.  Returns
    a read-only memmap when the chunks lie back to back in index order, else an array copy."""
    import os

    size = os.path.getsize(path)
    head = np.memmap(path, dtype=np.uint8, mode="r") if size else b""

    def need(cond, what):
        if not cond:
            raise FormatError(what)

    def raw(addr, n):
        need(addr + n <= size, "structure beyond the end of the file")
        return bytes(head[addr:addr + n])

    need(size >= 96 and raw(0, 8) == SIGNATURE, "not an HDF5 file")
    sb = raw(0, 96)
    need(sb[8] == 0, "superblock version %d (only 0 is handled)" % sb[8])
    need(sb[13] == 8 and sb[14] == 8, "offsets / lengths are not 8 bytes")
    base, _free, eof, _driver = struct.unpack_from("<QQQQ", sb, 24)
    need(base == 0, "non-zero base address")
    need(eof == size, "end-of-file address %d != file size %d" % (eof, size))
    _name_off, root_addr, cache, _res, btree_addr, heap_addr = struct.unpack_from("<QQIIQQ", sb, 56)
    need(cache == 1, "root entry does not cache its symbol table")

    def messages(addr):
        ver, _r, count, _ref, hsize = struct.unpack("<BBHII", raw(addr, 12))
        need(ver == 1, "object header version %d" % ver)
        body = raw(addr + 16, hsize)
        pos, out = 0, []
        for _ in range(count):
            need(pos + 8 <= hsize, "object header messages overrun the header (continuation blocks are not handled)")
            mtype, msize, _flags = struct.unpack_from("<HHB", body, pos)
            need(msize % 8 == 0 and pos + 8 + msize <= hsize, "bad message size")
            out.append((mtype, body[pos + 8:pos + 8 + msize]))
            pos += 8 + msize
        return out

    sym = [m for t, m in messages(root_addr) if t == 0x0011]
    need(len(sym) == 1 and struct.unpack("<QQ", sym[0][:16]) == (btree_addr, heap_addr), "root symbol table message")
    need(raw(heap_addr, 4) == b"HEAP", "local heap signature")
    _hv, seg_size, _free_head, seg_addr = struct.unpack("<B3xQQQ", raw(heap_addr + 4, 28))
    segment = raw(seg_addr, seg_size)

    def heap_name(off):
        need(off < seg_size, "name offset outside the heap")
        return segment[off:segment.index(b"\0", off)]

    def group_entries(addr):
        need(raw(addr, 4) == b"TREE", "B-tree signature")
        ntype, level, used = struct.unpack("<BBH", raw(addr + 4, 4))
        need(ntype == 0, "group B-tree node type")
        for i in range(used):
            (child,) = struct.unpack("<Q", raw(addr + 24 + 8 + i * 16, 8))
            if level > 0:
                yield from group_entries(child)
                continue
            need(raw(child, 4) == b"SNOD", "symbol table node signature")
            (nsym,) = struct.unpack("<H", raw(child + 6, 2))
            for j in range(nsym):
                yield struct.unpack("<QQ", raw(child + 8 + j * 40, 16))

    target = None
    for name_off, obj_addr in group_entries(btree_addr):
        if heap_name(name_off) == dataset.encode():
            target = obj_addr
    need(target is not None, "no dataset named %r" % dataset)
    shape = dtype_ok = layout = None
    for mtype, m in messages(target):
        if mtype == 0x0001:
            need(m[0] == 1, "dataspace version")
            shape = struct.unpack_from("<%dQ" % m[1], m, 8)
        elif mtype == 0x0003:
            dtype_ok = bytes(m[:20]) == F32LE_DATATYPE
        elif mtype == 0x0008:
            need(m[0] == 3 and m[1] in (1, 2), "layout is not version 3, contiguous or chunked")
            layout = m
        elif mtype == 0x000B:
            raise FormatError("filtered datasets are not handled")
    need(shape is not None and dtype_ok and layout is not None, "dataset header lacks dataspace / f32 datatype / layout")
    shape = tuple(int(x) for x in shape)
    count = int(np.prod(shape)) if shape else 1
    if layout[1] == 1:
        data_addr, nbytes = struct.unpack_from("<QQ", layout, 2)
        need(nbytes == count * 4 and data_addr + nbytes <= size, "layout size does not match the dataspace")
        addrs = None
    else:
        need(len(shape) == 3 and layout[2] == 4, "chunked layout of a dataset that is not a stack of images")
        (tree_root,) = struct.unpack_from("<Q", layout, 3)
        need(struct.unpack_from("<4I", layout, 11) == (1, shape[1], shape[2], 4), "chunks are not [1, rows, cols]")
        plane = shape[1] * shape[2] * 4
        addrs = {}

        def chunk_entries(addr, expect_level=None):
            need(raw(addr, 4) == b"TREE", "chunk B-tree signature")
            ntype, level, used = struct.unpack("<BBH", raw(addr + 4, 4))
            need(ntype == 1 and used <= 2 * CHUNK_K, "chunk B-tree node")
            need(expect_level is None or level == expect_level, "chunk B-tree level")
            for i in range(used):
                entry = raw(addr + 24 + i * (CHUNK_KEY + 8), CHUNK_KEY + 8)
                nbytes, mask, c0, c1, c2, c3, child = struct.unpack("<II4QQ", entry)
                if level > 0:
                    chunk_entries(child, level - 1)
                    continue
                need(nbytes == plane and mask == 0 and (c1, c2, c3) == (0, 0, 0) and c0 < shape[0], "chunk key")
                need(child + plane <= size, "chunk beyond the end of the file")
                addrs[c0] = child

        if tree_root != UNDEF:
            chunk_entries(tree_root)
        data_addr = addrs.get(0, 0)
    if count == 0:
        return np.zeros(shape, np.float32)
    if addrs is None or (len(addrs) == shape[0] and all(addrs[i] == data_addr + i * plane for i in range(shape[0]))):
        return np.memmap(path, dtype="<f4", mode="r", offset=data_addr, shape=shape)
    out = np.zeros(shape, np.float32)              # chunks never written read as the (default) fill value 0
    for i, a in addrs.items():
        out[i] = np.frombuffer(raw(a, plane), "<f4").reshape(shape[1:])
    return out

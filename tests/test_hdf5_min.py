"""The minimal HDF5 container of the driver loop (grayscott_amd/hdf5_min.py) -- CPU only.

The reference writes its images with libhdf5 (data/src/hdf5.rs:36-63) and reads them back with it
(:76-131, data-to-pics).  libhdf5 is not a dependency here, but when an installation is present
(HDF5_DIR, or /opt/conda as in the build image: HDF5 1.10.6) these tests use the REAL library as
the checker: its h5dump must read our files, its C API (through ctypes) must return our data, and a
file it writes itself must have the same message bytes as ours and must parse with our reader.
Without it, the byte-layout and round-trip tests below still run."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

from grayscott_amd import hdf5_min

HDF5_DIR = os.environ.get("HDF5_DIR", "/opt/conda")
H5DUMP = os.path.join(HDF5_DIR, "bin", "h5dump")
LIBHDF5 = next((p for p in (os.path.join(HDF5_DIR, "lib", n) for n in ("libhdf5.so", "libhdf5.so.103"))
                if os.path.exists(p)), None)
needs_h5dump = pytest.mark.skipif(not os.path.exists(H5DUMP), reason="no h5dump (HDF5_DIR)")
needs_libhdf5 = pytest.mark.skipif(LIBHDF5 is None, reason="no libhdf5 (HDF5_DIR)")


def stack(shape):
    return (np.arange(int(np.prod(shape)), dtype=np.float32) / np.float32(3)).reshape(shape)


def write(path, shape, **kw):
    out = hdf5_min.create(path, shape, **kw)
    ref = stack(shape)
    for i in range(shape[0]):
        out[i] = ref[i]
    if hasattr(out, "flush"):
        out.flush()
    return ref


def dataset_messages(raw, dataset=b"matrix"):
    """{message type: data} of the dataset's object header, located through the group structures."""
    _, _root, _, _, btree, heap = struct.unpack_from("<QQIIQQ", raw, 56)
    snod = struct.unpack_from("<Q", raw, btree + 32)[0]
    seg = struct.unpack_from("<Q", raw, heap + 24)[0]
    link, dset = struct.unpack_from("<QQ", raw, snod + 8)
    assert raw[seg + link:seg + link + len(dataset) + 1] == dataset + b"\0"
    count = struct.unpack_from("<H", raw, dset + 2)[0]
    pos, out = dset + 16, {}
    for _ in range(count):
        mtype, msize = struct.unpack_from("<HH", raw, pos)
        out[mtype] = raw[pos + 8:pos + 8 + msize]
        pos += 8 + msize
    return out


# ---- layout against the specification, round trip through our own parser --------------------
@pytest.mark.parametrize("layout", ["chunked", "contiguous"])
def test_round_trip_and_field_layout(tmp_path, layout):
    path = str(tmp_path / "out.h5")
    shape = (3, 5, 7)
    ref = write(path, shape, layout=layout)
    got = hdf5_min.read(path)
    assert got.shape == shape and got.dtype == np.float32 and np.array_equal(got, ref)

    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n"
    assert raw[8:16] == bytes([0, 0, 0, 0, 0, 8, 8, 0])                 # versions 0, 8-byte offsets and lengths
    assert struct.unpack_from("<HHI", raw, 16) == (4, 16, 0)            # group leaf K, internal K, flags
    base, free, eof, driver = struct.unpack_from("<QQQQ", raw, 24)
    assert base == 0 and free == driver == 0xFFFFFFFFFFFFFFFF and eof == len(raw)
    name_off, root, cache, _, btree, heap = struct.unpack_from("<QQIIQQ", raw, 56)
    assert name_off == 0 and cache == 1 and root == 96
    assert raw[root] == 1 and struct.unpack_from("<H", raw, root + 2)[0] == 1        # object header v1, one message
    assert struct.unpack_from("<HH", raw, root + 16) == (0x0011, 16)                 # Symbol Table message
    assert raw[btree:btree + 8] == b"TREE" + bytes([0, 0, 1, 0])                    # group node, level 0, 1 entry
    key0, snod, key1 = struct.unpack_from("<QQQ", raw, btree + 24)
    assert key0 == 0 and key1 == 8 and raw[snod:snod + 8] == b"SNOD" + bytes([1, 0, 1, 0])
    seg_size, free_head, seg = struct.unpack_from("<QQQ", raw, heap + 8)
    assert raw[heap:heap + 4] == b"HEAP" and free_head == 1 and raw[seg:seg + seg_size] == b"\0" * 8 + b"matrix\0\0"
    seen = dataset_messages(raw)
    assert seen[1][:4] == bytes([1, 3, 1, 0]) and struct.unpack_from("<6Q", seen[1], 8) == shape + shape
    assert seen[3][:20] == bytes.fromhex("11201f00" "04000000" "00002000" "17080017" "7f000000")   # H5T_IEEE_F32LE
    if layout == "contiguous":
        assert seen[8][:2] == bytes([3, 1])
        addr, nbytes = struct.unpack_from("<QQ", seen[8], 2)
        assert nbytes == ref.nbytes
    else:
        assert seen[8][:3] == bytes([3, 2, 4]) and struct.unpack_from("<4I", seen[8], 11) == (1, 5, 7, 4)
        tree = struct.unpack_from("<Q", seen[8], 3)[0]
        assert raw[tree:tree + 8] == b"TREE" + bytes([1, 0, 3, 0])                  # chunk node, level 0, 3 entries
        size, mask, c0, c1, c2, c3, addr = struct.unpack_from("<II4QQ", raw, tree + 24)
        assert (size, mask, c0, c1, c2, c3) == (140, 0, 0, 0, 0, 0)
    assert addr % 4096 == 0 and addr + ref.nbytes == len(raw)
    assert np.array_equal(np.frombuffer(raw, "<f4", ref.size, addr).reshape(shape), ref)


def test_reader_rejects_damage(tmp_path):
    path = str(tmp_path / "out.h5")
    write(path, (2, 4, 4))
    raw = bytearray(open(path, "rb").read())
    for offset in (0, 13, 40, 96 + 16):                       # signature, offset size, EOF address, root message type
        bad = bytearray(raw)
        bad[offset] ^= 0xFF
        open(path, "wb").write(bad)
        with pytest.raises(hdf5_min.FormatError):
            hdf5_min.read(path)
    open(path, "wb").write(raw)
    with pytest.raises(hdf5_min.FormatError):
        hdf5_min.read(path, dataset="other")
    assert hdf5_min.read(path).sum() == stack((2, 4, 4)).sum()


def test_other_names_and_empty_stacks(tmp_path):
    path = str(tmp_path / "a.h5")
    out = hdf5_min.create(path, (0, 3, 3), dataset="a_rather_long_dataset_name")
    assert out.shape == (0, 3, 3)
    assert hdf5_min.read(path, dataset="a_rather_long_dataset_name").shape == (0, 3, 3)


# ---- the real library as the checker -------------------------------------------------------
@needs_h5dump
@pytest.mark.parametrize("n,layout", [(1, "chunked"), (3, "chunked"), (64, "chunked"), (65, "chunked"),
                                      (70, "chunked"), (4200, "chunked"), (3, "contiguous"), (0, "chunked")])
def test_h5dump_reads_our_files(tmp_path, n, layout):
    """1-, 2- and 3-level chunk B-trees (64 entries per node), and the contiguous variant."""
    path, dump = str(tmp_path / "out.h5"), str(tmp_path / "dump.bin")
    shape = (n, 5, 7)
    ref = write(path, shape, layout=layout)
    header = subprocess.run([H5DUMP, "-p", "-H", path], capture_output=True, text=True, check=True).stdout
    assert 'DATASET "matrix"' in header and "H5T_IEEE_F32LE" in header
    assert f"( {n}, 5, 7 ) / ( {n}, 5, 7 )" in header
    assert ("CHUNKED ( 1, 5, 7 )" if layout == "chunked" else "CONTIGUOUS") in header
    if n:
        subprocess.run([H5DUMP, "-d", "/matrix", "-b", "LE", "-o", dump, path], capture_output=True, check=True)
        assert np.array_equal(np.fromfile(dump, "<f4").reshape(shape), ref)


class _H5:
    """The handful of C API calls the reference's Writer / Reader amount to."""

    def __init__(self):
        lib = self.lib = ctypes.CDLL(LIBHDF5)
        lib.H5open()
        hid = self.hid = ctypes.c_int64
        self.native_float = hid.in_dll(lib, "H5T_NATIVE_FLOAT_g").value
        self.f32le = hid.in_dll(lib, "H5T_IEEE_F32LE_g").value
        self.dcpl = hid.in_dll(lib, "H5P_CLS_DATASET_CREATE_ID_g").value
        u64p = ctypes.POINTER(ctypes.c_uint64)
        for name, args in (("H5Fcreate", [ctypes.c_char_p, ctypes.c_uint, hid, hid]), ("H5Fopen", [ctypes.c_char_p, ctypes.c_uint, hid]),
                           ("H5Screate_simple", [ctypes.c_int, u64p, u64p]), ("H5Pcreate", [hid]),
                           ("H5Dcreate2", [hid, ctypes.c_char_p, hid, hid, hid, hid, hid]), ("H5Dopen2", [hid, ctypes.c_char_p, hid]),
                           ("H5Dget_space", [hid])):
            getattr(lib, name).restype = hid
            getattr(lib, name).argtypes = args
        lib.H5Pset_chunk.argtypes = [hid, ctypes.c_int, u64p]
        lib.H5Dwrite.argtypes = lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, ctypes.c_void_p]
        lib.H5Sget_simple_extent_dims.argtypes = [hid, u64p, u64p]
        lib.H5Sselect_hyperslab.argtypes = [hid, ctypes.c_int, u64p, u64p, u64p, u64p]
        for name in ("H5Dclose", "H5Fclose", "H5Sclose", "H5Pclose"):
            getattr(lib, name).argtypes = [hid]

    def write_like_the_reference(self, path, data):
        """Writer::create + write (hdf5.rs:36-63): chunk [1, rows, cols], shape [n, rows, cols]."""
        lib, arr = self.lib, (ctypes.c_uint64 * 3)
        fid = lib.H5Fcreate(path.encode(), 2, 0, 0)                      # H5F_ACC_TRUNC
        sid = lib.H5Screate_simple(3, arr(*data.shape), None)
        pid = lib.H5Pcreate(self.dcpl)
        assert lib.H5Pset_chunk(pid, 3, arr(1, *data.shape[1:])) >= 0
        did = lib.H5Dcreate2(fid, b"matrix", self.f32le, sid, 0, pid, 0)
        assert fid >= 0 and did >= 0
        assert lib.H5Dwrite(did, self.native_float, 0, 0, 0, data.ctypes.data_as(ctypes.c_void_p)) >= 0
        for close, h in ((lib.H5Dclose, did), (lib.H5Pclose, pid), (lib.H5Sclose, sid), (lib.H5Fclose, fid)):
            assert close(h) >= 0

    def read_like_the_reference(self, path):
        """Reader::open + read image by image (hdf5.rs:76-131): a [1, rows, cols] hyperslab each."""
        lib, arr = self.lib, (ctypes.c_uint64 * 3)
        fid = lib.H5Fopen(path.encode(), 0, 0)                            # H5F_ACC_RDONLY
        did = lib.H5Dopen2(fid, b"matrix", 0)
        assert fid >= 0 and did >= 0
        fsp = lib.H5Dget_space(did)
        dims = arr()
        assert lib.H5Sget_simple_extent_dims(fsp, dims, None) == 3
        n, rows, cols = (int(x) for x in dims)
        out = np.empty((n, rows, cols), np.float32)
        msp = lib.H5Screate_simple(2, (ctypes.c_uint64 * 2)(rows, cols), None)
        for i in range(n):
            assert lib.H5Sselect_hyperslab(fsp, 0, arr(i, 0, 0), None, arr(1, rows, cols), None) >= 0   # H5S_SELECT_SET
            assert lib.H5Dread(did, self.native_float, msp, fsp, 0, out[i].ctypes.data_as(ctypes.c_void_p)) >= 0
        for close, h in ((lib.H5Sclose, msp), (lib.H5Sclose, fsp), (lib.H5Dclose, did), (lib.H5Fclose, fid)):
            assert close(h) >= 0
        return out


@needs_libhdf5
@pytest.mark.parametrize("n", [1, 70, 300])
def test_libhdf5_reads_our_files_the_way_the_reference_reader_does(tmp_path, n):
    path = str(tmp_path / "ours.h5")
    ref = write(path, (n, 6, 11))
    assert np.array_equal(_H5().read_like_the_reference(path), ref)


@needs_libhdf5
def test_same_message_bytes_as_a_file_libhdf5_writes_and_our_reader_parses_it(tmp_path):
    ours, theirs = str(tmp_path / "ours.h5"), str(tmp_path / "theirs.h5")
    shape = (70, 5, 7)
    ref = write(ours, shape)
    _H5().write_like_the_reference(theirs, ref)
    a, b = dataset_messages(open(ours, "rb").read()), dataset_messages(open(theirs, "rb").read())
    assert a[1] == b[1]                                  # dataspace
    assert a[3] == b[3]                                  # datatype
    assert a[5] == b[5]                                  # fill value
    assert a[8][:3] == b[8][:3] and a[8][11:27] == b[8][11:27]     # layout: version, class, rank + 1; chunk dims
    assert open(ours, "rb").read(56) [:40] == open(theirs, "rb").read(56)[:40]   # superblock up to the EOF address
    got = hdf5_min.read(theirs)                          # a 2-level tree with partly filled nodes
    assert np.array_equal(got, ref)

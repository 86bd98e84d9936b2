"""Generates tests/golden/h5/*: small HDF5 files written by the REAL libraries (PyTables 3.6.1 + libhdf5 1.10.6, the
package the reference opens its image features with, train_mm_vi_model1.py:460; and h5py 3.3.0 on the same libhdf5) plus
the expected array contents, read back through h5py (libhdf5's own reader) into .npy files.  Test infrastructure only.

Run with the image's conda interpreter, the only one that has those libraries:
    /opt/conda/bin/python3.9 oracle/make_h5_golden.py
(`numpy.typeDict` was removed in numpy 1.24 and PyTables 3.6.1 still imports it: aliased below before the import.  With
this numpy PyTables can WRITE but not read its files back, hence h5py as the independent reader.)
The fixtures are data (inputs + expected outputs); neither library travels to the GPU box.
"""
import os
import sys

import numpy as np

np.typeDict = np.sctypeDict
import h5py       # noqa: E402
import tables     # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "h5")


def feats(n, d, seed):
    r = np.random.RandomState(seed)
    return np.abs(r.randn(n, d)).astype(np.float32)        # ResNet pool5 features are post-ReLU: non-negative


def main():
    os.makedirs(OUT, exist_ok=True)
    expect = {}

    # 1. PyTables Array nodes (contiguous layout), the three node names of train_mm_vi_model1.py:463-476
    p = os.path.join(OUT, "pt_array.h5")
    f = tables.open_file(p, "w")
    f.create_array(f.root, "global_feats", feats(37, 64, 1))
    f.create_array(f.root, "local_feats", feats(5, 3 * 4 * 4, 2).reshape(5, 16, 3))
    f.create_array(f.root, "logits", np.random.RandomState(3).randn(37, 10))                      # float64
    f.close()

    # 2. PyTables EArray appended in pieces (chunked, no filter) -- how feature extractors usually write
    p = os.path.join(OUT, "pt_earray.h5")
    f = tables.open_file(p, "w")
    e = f.create_earray(f.root, "global_feats", tables.Float32Atom(), (0, 64), chunkshape=(8, 64))
    x = feats(45, 64, 4)
    for i in range(0, 45, 7):
        e.append(x[i:i + 7])
    f.close()

    # 3. PyTables CArray with zlib + shuffle, chunks that do not divide the shape (edge chunks in both axes),
    #    and a second, fletcher32-protected node; many small chunks -> multi-level chunk B-tree
    p = os.path.join(OUT, "pt_carray_zlib_shuffle.h5")
    f = tables.open_file(p, "w")
    c = f.create_carray(f.root, "global_feats", tables.Float32Atom(), (53, 70), chunkshape=(6, 32),
                        filters=tables.Filters(complevel=5, complib="zlib", shuffle=True))
    c[:] = feats(53, 70, 5)
    c2 = f.create_carray(f.root, "logits", tables.Float64Atom(), (300, 5), chunkshape=(1, 5),
                         filters=tables.Filters(complevel=1, complib="zlib", shuffle=False, fletcher32=True))
    c2[:] = np.random.RandomState(6).randn(300, 5)
    g = f.create_group(f.root, "extra")
    f.create_array(g, "ids", np.arange(11, dtype=np.int64) * 3 - 7)
    f.close()

    # 4. standardisation files (train_mm_vi_model1.py:490-495)
    p = os.path.join(OUT, "pt_mean.h5")
    f = tables.open_file(p, "w")
    f.create_array(f.root, "global_feats_mean", feats(1, 64, 7)[0])
    f.close()
    p = os.path.join(OUT, "pt_std.h5")
    f = tables.open_file(p, "w")
    f.create_array(f.root, "global_feats_stds", feats(1, 64, 8)[0] + 0.5)
    f.close()

    # 5. h5py, default format: contiguous, chunked + gzip + shuffle, big-endian, compact, partially written chunked
    p = os.path.join(OUT, "h5py_default.h5")
    f = h5py.File(p, "w")
    f.create_dataset("global_feats", data=feats(21, 48, 9))
    f.create_dataset("local_feats", data=feats(21, 48, 10), chunks=(4, 16), compression="gzip", shuffle=True)
    f.create_dataset("logits", data=np.arange(24, dtype=">f8").reshape(6, 4))
    f.create_dataset("array", data=np.arange(60, dtype=np.uint8).reshape(3, 4, 5), chunks=(2, 3, 2))
    d = f.create_dataset("sparse", shape=(20, 6), dtype="f4", chunks=(5, 6))
    d[10:15] = 2.5
    f.close()

    # 6. h5py, libver='latest' with small group: v2 object headers, compact link messages, superblock v3
    p = os.path.join(OUT, "h5py_latest.h5")
    f = h5py.File(p, "w", libver="latest")
    f.create_dataset("global_feats", data=feats(9, 32, 11))
    f.create_dataset("global_feats_mean", data=feats(1, 32, 12)[0].astype(np.float64))
    f.close()

    # 7. a group with enough children to split its B-tree / use several symbol-table nodes
    p = os.path.join(OUT, "pt_many_nodes.h5")
    f = tables.open_file(p, "w")
    for i in range(40):
        f.create_array(f.root, "node_%02d" % i, np.full((2, 3), i, dtype=np.int32))
    f.close()

    # 8. a table of the model's real width (2048) for the end-to-end driver-flow test: coarse values so that zlib + shuffle
    #    keep the fixture small
    p = os.path.join(OUT, "pt_feats2048.h5")
    f = tables.open_file(p, "w")
    c = f.create_carray(f.root, "global_feats", tables.Float32Atom(), (60, 2048), chunkshape=(16, 2048),
                        filters=tables.Filters(complevel=9, complib="zlib", shuffle=True))
    c[:] = (np.random.RandomState(13).randint(0, 48, size=(60, 2048)) / 16.0).astype(np.float32)
    f.close()

    # expected contents through libhdf5's own reader
    for fn in sorted(os.listdir(OUT)):
        if not fn.endswith(".h5"):
            continue
        with h5py.File(os.path.join(OUT, fn), "r") as f:
            def visit(name, obj):
                if isinstance(obj, h5py.Dataset):
                    a = obj[...]
                    expect[fn[:-3] + "::" + name] = a.astype(a.dtype.newbyteorder("="))
            f.visititems(visit)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **expect)
    for k, v in sorted(expect.items()):
        print(k, v.shape, v.dtype)
    print("wrote", OUT, "tables", tables.__version__, "hdf5", tables.hdf5_version, "h5py", h5py.__version__)


if __name__ == "__main__":
    sys.exit(main())

"""The image-feature table's way from an HDF5 file into HBM.

Reference (train_mm_vi_model1.py:460-501): `tables.open_file(path).root.global_feats[:]` -> a host numpy array, optionally
standardised on the host with the training set's mean / std files, kept on the host for the whole run and fancy-indexed +
copied to the GPU every step (TrainerMultimodal.py:632-639).  Here the table lives in HBM (237 MB for 29 k rows, 2.4 GB for
290 k, 8.2 GB for 1 M: small against 288 GB): the file is streamed slab by slab through two pinned staging buffers on a copy
stream (disk read of slab i+1 overlaps the H2D copy of slab i), standardised in place by `vmmt_standardise_rows`, and rows
are gathered on the device from `batch.indices` (`vmmt_gather_rows`).  The file is parsed by `onmt/h5tables.py` (no
PyTables / libhdf5 dependency).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .onmt import h5tables

NODE_BY_FLAG = {"global": "global_feats", "local": "local_feats", "posterior": "logits"}     # train_mm_vi_model1.py:463-476


def read_vector(path, node):
    """a 1-D fp32 vector node (the mean / std files, train_mm_vi_model1.py:490-495)"""
    with h5tables.open_file(path) as f:
        return np.ascontiguousarray(f.get_node("/" + node)[:], dtype=np.float32)


def load_image_table(path, node="global_feats", device="cuda", mean_path=None, std_path=None, slab_bytes=64 << 20,
                     mean_node="global_feats_mean", std_node="global_feats_stds"):
    """-> fp32 CUDA tensor [N, D] holding `/node` of the HDF5 file `path`, standardised if both mean and std files are given.
    Raises on anything that is not a 2-D integer/float array (the model consumes [N, D] rows)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("load_image_table places the table in HBM: a GPU device is required")
    if (mean_path is None) != (std_path is None):
        raise ValueError("standardisation needs both the mean and the std file")
    lib = L.lib()
    with h5tables.open_file(path) as f:
        arr = f.get_node("/" + node)
        if len(arr.shape) != 2:
            raise ValueError("node /%s of %s has shape %r: expected [N, D]" % (node, path, tuple(arr.shape)))
        N, D = int(arr.shape[0]), int(arr.shape[1])
        table = torch.empty((N, D), dtype=torch.float32, device=dev)
        rows = max(1, min(N, slab_bytes // max(1, D * 4)))
        native = arr.dtype.newbyteorder("=")
        direct = native == np.dtype(np.float32)
        stage = [torch.empty((rows, D), dtype=torch.float32, device="cpu").pin_memory() for _ in range(2 if N > rows else 1)]
        done = [None] * len(stage)
        copy_stream = torch.cuda.Stream(device=dev)
        for i, r0 in enumerate(range(0, N, rows)):
            r1 = min(N, r0 + rows)
            k = i % len(stage)
            if done[k] is not None:
                done[k].synchronize()                       # the staging buffer's previous copy has left it
            view = stage[k][:r1 - r0].numpy()
            if direct:
                arr.read_into(view, r0, r1)
            else:
                view[...] = arr.read(r0, r1)                  # float64 / integer files: converted on the way (numpy cast)
            with torch.cuda.stream(copy_stream):
                table[r0:r1].copy_(stage[k][:r1 - r0], non_blocking=True)
                done[k] = torch.cuda.Event()
                done[k].record(copy_stream)
        torch.cuda.current_stream(dev).wait_stream(copy_stream)
    if mean_path is not None:
        mean = torch.from_numpy(read_vector(mean_path, mean_node)).to(dev)
        std = torch.from_numpy(read_vector(std_path, std_node)).to(dev)
        if mean.numel() != D or std.numel() != D:
            raise ValueError("mean/std of length %d/%d for a table of width %d" % (mean.numel(), std.numel(), D))
        st = torch.cuda.current_stream(dev).cuda_stream
        L.check(lib.vmmt_standardise_rows(C.c_void_p(table.data_ptr()), D, C.c_void_p(mean.data_ptr()),
                                          C.c_void_p(std.data_ptr()), N, D, C.c_void_p(st)), "vmmt_standardise_rows")
        torch.cuda.current_stream(dev).synchronize()          # mean / std may be freed on return
    return table

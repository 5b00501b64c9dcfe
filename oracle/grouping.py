"""ORACLE -- test infrastructure only (see oracle/__init__.py).

Two CPU restatements of the reference grouping op
(ops/query_depth_point/query_depth_point_cuda_kernel.cu:16-65, wrapper
ops/query_depth_point/query_depth_point.py:29-40):

* ``query_depth_point_c``      -- ctypes call into oracle/qdp_ref.c (serial scan, OpenMP over queries)
* ``query_depth_point_numpy``  -- independent vectorised formulation (mask + stable rank), used to
                                  cross-check the C one and as the fallback when the .so is not built

Both take the module-level layout the reference wrapper takes: xyz1 (B,3,N), xyz2 (B,3,M), float32.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libqdp_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle` or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.qdp_ref_f32.restype = ctypes.c_int
        lib.qdp_ref_f32.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def query_depth_point_c(dis_z, nsample, xyz1, xyz2):
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float32)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float32)
    B, _, N = xyz1.shape
    M = xyz2.shape[2]
    idx = np.empty((B, M, nsample), dtype=np.int64)
    cnt = np.empty((B, M), dtype=np.int32)
    z1 = xyz1[:, 2, :]
    z2 = xyz2[:, 2, :]
    rc = _lib().qdp_ref_f32(
        z1.ctypes.data, 1, 3 * N, z2.ctypes.data, 1, 3 * M,
        B, N, M, ctypes.c_float(dis_z), nsample, idx.ctypes.data, cnt.ctypes.data)
    assert rc == 0
    return idx, cnt


def query_depth_point_numpy(dis_z, nsample, xyz1, xyz2):
    """Mask/rank formulation: hit mask in fp32, rank of each hit by cumulative count,
    slot c <- c-th hit, remaining slots <- first hit, empty rows stay zero."""
    z1 = np.asarray(xyz1, dtype=np.float32)[:, 2, :]          # (B,N)
    z2 = np.asarray(xyz2, dtype=np.float32)[:, 2, :]          # (B,M)
    d = np.abs(z2[:, :, None] - z1[:, None, :])               # fp32 subtract + abs, (B,M,N)
    assert d.dtype == np.float32
    hit = d < np.float32(dis_z)
    rank = np.cumsum(hit, axis=2) - 1                         # rank of a hit among hits
    B, M, N = hit.shape
    idx = np.zeros((B, M, nsample), dtype=np.int64)
    cnt = np.minimum(hit.sum(axis=2), nsample).astype(np.int32)
    take = hit & (rank < nsample)
    bb, mm, kk = np.nonzero(take)
    # pad with first hit: fill rows that have any hit with their first hit index
    first = np.argmax(hit, axis=2)                            # 0 when no hit (row stays zero)
    idx[:] = np.where(cnt[:, :, None] > 0, first[:, :, None], 0)
    idx[bb, mm, rank[bb, mm, kk]] = kk
    return idx, cnt


def query_depth_point(dis_z, nsample, xyz1, xyz2):
    try:
        return query_depth_point_c(dis_z, nsample, xyz1, xyz2)
    except FileNotFoundError:
        return query_depth_point_numpy(dis_z, nsample, xyz1, xyz2)

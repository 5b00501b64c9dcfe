"""QueryDepthPoint: drop-in for the reference's ops/query_depth_point/query_depth_point.py:9-54.

Same constructor, same call signature, same dtypes/shapes:
    QueryDepthPoint(dis_z, nsample)(xyz1 (B,3,N) f32, xyz2 (B,3,M) f32) -> (idx (B,M,nsample) int64,
                                                                             pts_cnt (B,M) int32)
and the same argument checks (device tensors, size(1) == 3, equal batch, contiguous).  Differences:
no device-side transposes (the reference copies both inputs to (B,N,3) on every call,
query_depth_point.py:29-30 -- the kernel here reads the contiguous z row of the (B,3,N) layout through
explicit strides) and the outputs are fully written by the kernel (no zero-fill pass).
Outputs are integer tensors, hence non-differentiable, as in the reference (backward returns None).
The module also takes float64 inputs like the reference's dispatch (narrowed to fp32, which is what the reference's kernel
does with them); the functional form asserts float32.
"""
import torch
from torch import nn

from . import _native


def query_depth_point(dis_z, nsample, xyz1, xyz2):
    assert xyz1.is_cuda and xyz1.size(1) == 3
    assert xyz2.is_cuda and xyz2.size(1) == 3
    assert xyz1.size(0) == xyz2.size(0)
    assert xyz1.is_contiguous()
    assert xyz2.is_contiguous()
    assert xyz1.dtype == torch.float32 and xyz2.dtype == torch.float32
    L = _native.lib()
    b, _, n = xyz1.shape
    m = xyz2.size(2)
    idx = torch.empty((b, m, nsample), dtype=torch.int64, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        rc = L.fcn_query_depth_point_f32(
            xyz1.data_ptr() + 4 * 2 * n, 1, 3 * n,
            xyz2.data_ptr() + 4 * 2 * m, 1, 3 * m,
            b, n, m, float(dis_z), int(nsample), idx.data_ptr(), cnt.data_ptr(),
            _native.current_stream(xyz1.device))
    _native.check(rc, "fcn_query_depth_point_f32")
    return idx, cnt


def query_depth_point_multi(dis_z, nsample, xyz1, xyz2_list, out=None):
    """The operator for every scale of one batch in ONE launch (fcn_query_depth_point_multi_f32): dis_z / nsample / xyz2_list hold one
    entry per scale ((B,3,M_s) window centres), xyz1 (B,3,N) is shared.  -> [(idx_s, cnt_s)], each exactly what
    query_depth_point(dis_z[s], nsample[s], xyz1, xyz2_list[s]) returns.  out: the same list from an earlier call, written in place."""
    import ctypes
    ns = len(xyz2_list)
    assert 1 <= ns <= 8 and len(dis_z) == ns and len(nsample) == ns
    assert xyz1.is_cuda and xyz1.size(1) == 3 and xyz1.is_contiguous() and xyz1.dtype == torch.float32
    b, _, n = xyz1.shape
    for x2 in xyz2_list:
        assert x2.is_cuda and x2.size(1) == 3 and x2.size(0) == b and x2.is_contiguous() and x2.dtype == torch.float32
    ms = [int(x2.size(2)) for x2 in xyz2_list]
    if out is None:
        out = [(torch.empty((b, m, int(k)), dtype=torch.int64, device=xyz1.device), torch.empty((b, m), dtype=torch.int32, device=xyz1.device))
               for m, k in zip(ms, nsample)]
    ptrs = lambda vals: (ctypes.c_void_p * ns)(*vals)
    i64 = lambda vals: (ctypes.c_int64 * ns)(*vals)
    L = _native.lib()
    with torch.cuda.device(xyz1.device):
        rc = L.fcn_query_depth_point_multi_f32(
            ns, xyz1.data_ptr() + 4 * 2 * n, 1, 3 * n,
            ptrs([x2.data_ptr() + 4 * 2 * m for x2, m in zip(xyz2_list, ms)]), i64([1] * ns), i64([3 * m for m in ms]),
            b, n, (ctypes.c_int32 * ns)(*ms), (ctypes.c_float * ns)(*[float(d) for d in dis_z]),
            (ctypes.c_int32 * ns)(*[int(k) for k in nsample]),
            ptrs([o[0].data_ptr() for o in out]), ptrs([o[1].data_ptr() for o in out]), _native.current_stream(xyz1.device))
    _native.check(rc, "fcn_query_depth_point_multi_f32")
    return out


def query_depth_point_bn3(dis_z, nsample, xyz1_bn3, xyz2_bm3):
    """Same op on the kernel-native layout of the reference's pybind entry
    (query_depth_point_cuda.cpp:25-50): xyz1 (B,N,3), xyz2 (B,M,3)."""
    assert xyz1_bn3.is_cuda and xyz1_bn3.size(2) == 3 and xyz1_bn3.is_contiguous()
    assert xyz2_bm3.is_cuda and xyz2_bm3.size(2) == 3 and xyz2_bm3.is_contiguous()
    L = _native.lib()
    b, n, _ = xyz1_bn3.shape
    m = xyz2_bm3.size(1)
    idx = torch.empty((b, m, nsample), dtype=torch.int64, device=xyz1_bn3.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1_bn3.device)
    with torch.cuda.device(xyz1_bn3.device):
        rc = L.fcn_query_depth_point_f32(
            xyz1_bn3.data_ptr() + 8, 3, 3 * n, xyz2_bm3.data_ptr() + 8, 3, 3 * m,
            b, n, m, float(dis_z), int(nsample), idx.data_ptr(), cnt.data_ptr(),
            _native.current_stream(xyz1_bn3.device))
    _native.check(rc, "fcn_query_depth_point_f32")
    return idx, cnt


class QueryDepthPoint(nn.Module):
    def __init__(self, dis_z, nsample):
        super(QueryDepthPoint, self).__init__()
        self.dis_z = dis_z
        self.nsample = nsample

    def forward(self, xyz1, xyz2):
        with torch.no_grad():
            xyz1, xyz2 = xyz1.detach(), xyz2.detach()
            if xyz1.dtype == torch.float64 and xyz2.dtype == torch.float64:
                # The reference dispatches double too (query_depth_point_cuda_kernel.cu:77) but its kernel narrows both depths
                # to float before it compares them (`float z1 = ...`, `fabsf(z2 - z1) < dis_z`, .cu:40,48): the result IS the
                # fp32 result on the narrowed inputs, so narrowing here reproduces it exactly.
                xyz1, xyz2 = xyz1.float(), xyz2.float()
            return query_depth_point(self.dis_z, self.nsample, xyz1, xyz2)

"""TEST INFRASTRUCTURE ONLY: runs the package's GPU-only Python layer on CPU tensors against the host emulation of the
kernels (tests/host_harness/build_emu.py).  Inside `emulated_gpu()`:
  * _native.lib() is the emulated library (it takes host pointers) and every stream handle is NULL;
  * CPU tensors answer is_cuda = True, `.cuda()` is the identity, streams / events / device contexts are inert objects
    (the emulation executes every launch synchronously, so stream order is trivially respected).
Nothing of this is reachable from the product: without the context the package refuses CPU tensors loudly."""
import contextlib

import torch
from torch.overrides import TorchFunctionMode


def _is_cuda_dev(x):
    if isinstance(x, torch.device):
        return x.type == "cuda"
    return isinstance(x, str) and (x == "cuda" or x.startswith("cuda:"))


class _CudaToCpu(TorchFunctionMode):
    """Every `device=cuda...` of a factory / `.to()` call becomes the CPU: tensors the package would put on the GPU live in
    host memory, where the emulated library reads them."""

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if func is torch.device:                  # descriptors stay what they are (the package's guards look at .type)
            return func(*args, **kwargs)
        if "device" in kwargs and _is_cuda_dev(kwargs["device"]):
            kwargs["device"] = torch.device("cpu")
        if any(_is_cuda_dev(a) for a in args):
            args = tuple(torch.device("cpu") if _is_cuda_dev(a) else a for a in args)
            if getattr(func, "__name__", "") == "to":
                kwargs["copy"] = True         # t.to("cuda") copies on the device too
        return func(*args, **kwargs)


class _Inert:
    cuda_stream = 0
    cuda_event = 0

    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def query(self):
        return True

    def elapsed_time(self, other):
        return 0.0

    def __getattr__(self, name):          # wait_event, wait_stream, record, synchronize, record_event, wait ...
        return lambda *a, **k: None


@contextlib.contextmanager
def emulated_gpu():
    from emu_fcn import emu_path
    from frustum_convnet_amd import _native
    T, C = torch.Tensor, torch.cuda
    saved_native = (_native.LIB_PATH, _native._lib, _native.current_stream)
    saved_t = {k: T.__dict__.get(k) for k in ("is_cuda", "cuda", "record_stream")}
    saved_c = {k: getattr(C, k) for k in ("device", "Stream", "Event", "current_stream", "stream", "synchronize")}
    try:
        _native.LIB_PATH, _native._lib = emu_path(), None
        _native.current_stream = lambda device=None: None
        T.is_cuda = property(lambda self: True)
        # a COPY, as on the device: the storage then comes from torch's allocator (which the guard-page interposer of
        # tests/host_harness/guard watches) and no longer from numpy's
        T.cuda = lambda self, *a, **k: self.clone()
        T.record_stream = lambda self, s: None
        C.device = _Inert
        C.Stream = _Inert
        C.Event = _Inert
        C.current_stream = lambda device=None: _Inert()
        C.stream = lambda s: _Inert()
        C.synchronize = lambda *a, **k: None
        with _CudaToCpu():
            yield _native.lib()
    finally:
        _native.LIB_PATH, _native._lib, _native.current_stream = saved_native
        for k, v in saved_t.items():
            if v is None:
                delattr(T, k)
            else:
                setattr(T, k, v)
        for k, v in saved_c.items():
            setattr(C, k, v)

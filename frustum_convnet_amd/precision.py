"""Matrix-core operand precision of the HIP GEMM kernels (include/fcn_hip.h: FCN_PREC_*).

  "split"  default, the parity mode: fp32 operands split into two 16-bit parts, three MFMAs per product (fp16 parts in
           the forward GEMMs, bf16 parts in the backward GEMMs) -- fp32-class results, 5.3x the fp32 matrix rate
  "f32"    exact fp32 MFMA, the A/B reference
  "bf16"   throughput mode of BASELINE config 2: single bf16 MFMA per product, fp32 accumulate / statistics, the big
           intermediate tensors stored as bf16 (half the HBM bytes of the step's dominant streams)
  "bf16ops" the same operands with fp32 storage (faster on MI355X: the step is latency-bound)

The mode is read when a forward is enqueued; the matching backward reuses the forward's descriptor.  Initial value from
the environment variable FCN_PRECISION.
"""
import contextlib
import os

CODES = {"split": 0, "f32": 1, "bf16": 2, "bf16ops": 3}
_current = os.environ.get("FCN_PRECISION", "split")
if _current not in CODES:
    raise ValueError("FCN_PRECISION must be one of %s" % sorted(CODES))


def set_precision(name):
    global _current
    if name not in CODES:
        raise ValueError("precision must be one of %s, got %r" % (sorted(CODES), name))
    _current = name


def get_precision():
    return _current


def code():
    return CODES[_current]


@contextlib.contextmanager
def precision(name):
    prev = get_precision()
    set_precision(name)
    try:
        yield
    finally:
        set_precision(prev)

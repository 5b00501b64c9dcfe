"""On-device batch construction from raw frustum records (C-ABI fcn_prepare_inputs, csrc/inputs.hip).

Host-side mirror of the reference's loader for this step (datasets/provider_sample.py): ProviderDataset.__getitem__
(:137-262) builds every sample in numpy on DataLoader workers (cfgs/det_sample.yaml NUM_WORKERS) and default_collate
(:396-397) stacks them; here the raw records of a batch go to the GPU in ONE packed upload and one kernel launch writes
the dict PointNetDet.forward consumes (same keys, shapes and dtypes as the reference's collated batch).

Randomness stays on the host and follows the reference's call order per sample -- np.random.choice (resample),
np.random.random (flip coin), np.random.randn (depth shift) -- so with the same numpy seed the batch equals the
reference's batch (tests/test_gpu_inputs.py checks that against the golden fixture).
"""
import ctypes

import numpy as np
import torch

from . import _native
from ._native import InpDesc, Inp5Desc
from .config import cfg
from .dataset_info import DATASET_INFO

RECORD_KEYS = ("points", "seg", "box2d", "P", "box3d", "heading", "size", "frustum_angle", "type")


def draw(counts, npoints, random_flip=True, random_shift=True, rng=np.random):
    """The reference's per-sample random draws, in its order (provider_sample.py:165-168, :225, :238)."""
    choice, coin, normal = [], [], []
    for n in counts:
        choice.append(rng.choice(int(n), npoints, int(n) < npoints))
        coin.append(rng.random() if random_flip else 0.0)
        normal.append(rng.randn() if random_shift else 0.0)
    return np.stack(choice).astype(np.int32), np.asarray(coin, dtype=np.float64), np.asarray(normal, dtype=np.float64)


class InputBuilder:
    """npoints / strides / max_depth as cfg.DATA.{NUM_SAMPLES, STRIDE, MAX_DEPTH}; one_hot over the dataset's classes."""

    def __init__(self, npoints, strides=None, max_depth=None, random_flip=False, random_shift=False, one_hot=True,
                 device="cuda"):
        self.npoints = int(npoints)
        self.strides = tuple(float(s) for s in (cfg.DATA.STRIDE if strides is None else strides))
        self.max_depth = float(cfg.DATA.MAX_DEPTH if max_depth is None else max_depth)
        assert len(self.strides) == 4
        self.L = [len(np.arange(0, self.max_depth, s)) for s in self.strides]
        self.random_flip, self.random_shift, self.one_hot = bool(random_flip), bool(random_shift), bool(one_hot)
        self.device = torch.device(device)
        self.classes = DATASET_INFO[cfg.DATA.DATASET_NAME].CLASSES

    def build(self, records, draws=None, with_seg=True):
        """records: list of dicts with RECORD_KEYS (points (n,>=3) float32 in rect camera coordinates, seg (n,), box2d (4,),
        P (3,4), box3d (8,3) corners, heading, size (l,w,h), frustum_angle, type).  draws: (choice (B,N) int32, coin (B),
        normal (B)) or None to draw like the reference.  Returns the batch dict on the device.
        = upload() (host -> device copies of the raw records and the draws) + launch() (the kernel)."""
        return self.launch(self.upload(records, draws, with_seg))

    def upload(self, records, draws=None, with_seg=True):
        """The raw records of a batch and their random draws as device tensors (ONE packed point buffer): what a loader hands
        the GPU.  The returned dict feeds launch() any number of times (a resident raw batch re-built every step: bench.py)."""
        if self.device.type != "cuda":
            raise RuntimeError("frustum_convnet_amd: input construction is a HIP kernel (MI355X only); no CPU fallback")
        B, N = len(records), self.npoints
        counts = [len(r["points"]) for r in records]
        if draws is None:
            draws = draw(counts, N, self.random_flip, self.random_shift)
        choice, coin, normal = draws
        stride = int(records[0]["points"].shape[1])
        raw = np.concatenate([np.ascontiguousarray(r["points"], dtype=np.float32) for r in records], 0)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        f64 = lambda k, shape: np.stack([np.asarray(r[k], dtype=np.float64).reshape(shape) for r in records])
        dev = self.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)
        t = {"raw": up(raw), "off": up(off), "choice": up(np.asarray(choice, dtype=np.int32)),
             "fangle": up(f64("frustum_angle", ())), "box2d": up(f64("box2d", (4,))), "P": up(f64("P", (12,))),
             "corners": up(f64("box3d", (24,))), "heading": up(f64("heading", ())), "size": up(f64("size", (3,))),
             "coin": up(np.asarray(coin, dtype=np.float64)), "normal": up(np.asarray(normal, dtype=np.float64))}
        t["seg"] = up(np.concatenate([np.asarray(r["seg"]).astype(np.int64) for r in records], 0)) if with_seg else None
        size_class = [self.classes.index(r["type"]) for r in records]
        t["size_class"] = torch.tensor(size_class, dtype=torch.int64).view(B, 1).to(dev, non_blocking=True)
        if self.one_hot:
            oh = np.zeros((B, len(self.classes)), dtype=np.float32)
            oh[np.arange(B), size_class] = 1.0
            t["one_hot"] = up(oh)
        t["B"], t["pt_stride"] = B, stride
        return t

    def alloc(self, B, with_seg=True):
        """Output tensors of launch() for a batch of B frustums (pass as `out` to write the same buffers every step)."""
        dev, N = self.device, self.npoints
        f32 = dict(dtype=torch.float32, device=dev)
        out = {"point_cloud": torch.empty((B, 3, N), **f32), "rot_angle": torch.empty((B, 1), **f32),
               "cls_label": torch.empty((B, self.L[1]), dtype=torch.int64, device=dev),
               "box3d_center": torch.empty((B, 3), **f32), "box3d_heading": torch.empty((B, 1), **f32),
               "box3d_size": torch.empty((B, 3), **f32)}
        for s in range(4):
            out["center_ref%d" % (s + 1)] = torch.empty((B, 3, self.L[s]), **f32)
        if with_seg:
            out["seg_label"] = torch.empty((B, N), dtype=torch.int64, device=dev)
        return out

    def launch(self, t, out=None):
        """ONE kernel launch on the current stream: uploaded records `t` -> the batch dict (into `out` when given: capturable
        into a hipGraph, no allocation, no host work besides the call).  size_class / one_hot are the uploaded tensors."""
        B, N = t["B"], self.npoints
        dev = self.device
        if out is None:
            out = self.alloc(B, with_seg=t["seg"] is not None)
        desc = InpDesc(B, N, t["pt_stride"], (ctypes.c_int32 * 4)(*self.L), (ctypes.c_double * 4)(*self.strides), self.max_depth,
                       1 if self.random_flip else 0, 1 if self.random_shift else 0)
        refs = (ctypes.c_void_p * 4)(*[out["center_ref%d" % (s + 1)].data_ptr() for s in range(4)])
        p = lambda x: None if x is None else x.data_ptr()
        L = _native.lib()
        with torch.cuda.device(dev):
            _native.check(L.fcn_prepare_inputs(ctypes.byref(desc), p(t["raw"]), p(t["off"]), p(t["seg"]), p(t["choice"]),
                                               p(t["fangle"]), p(t["box2d"]), p(t["P"]), p(t["corners"]), p(t["heading"]),
                                               p(t["size"]), p(t["coin"]), p(t["normal"]), p(out["point_cloud"]), refs,
                                               p(out["cls_label"]), p(out["box3d_center"]), p(out["box3d_heading"]),
                                               p(out["box3d_size"]), p(out["rot_angle"]),
                                               p(out.get("seg_label")) if t["seg"] is not None else None,
                                               _native.current_stream(dev)), "fcn_prepare_inputs")
        for v in t.values():                       # the uploads are read by the kernel just enqueued
            if isinstance(v, torch.Tensor):
                v.record_stream(torch.cuda.current_stream(dev))
        out["size_class"] = t["size_class"]
        if "one_hot" in t:
            out["one_hot"] = t["one_hot"]
        return out

    def algorithmic_bytes(self, B, with_seg=True, pt_stride=4):
        """HBM bytes one launch has to move (the roofline's numerator): per frustum N gathered raw points + their draw indices
        (+ seg labels) in, point cloud + window centres + labels (+ seg) out; the per-frustum scalars (~400 B) ignored."""
        N = self.npoints
        per = N * (4 * pt_stride + 4) + 12 * N + 12 * sum(self.L) + 8 * self.L[1] + (16 * N if with_seg else 0)
        return B * per


def draw_sunrgbd(counts, npoints, random_flip=True, random_shift=True, rng=np.random):
    """The SUN-RGBD loader's per-sample draws, in its order (provider_sample_sunrgbd.py:144, :211, :224, :228): resample
    (WITH replacement only when the frustum has fewer points than npoints), flip coin, depth-shift normal, height-shift
    uniform."""
    choice, coin, normal, hshift = [], [], [], []
    for n in counts:
        choice.append(rng.choice(int(n), npoints, int(n) < npoints))
        coin.append(rng.random() if random_flip else 0.0)
        normal.append(rng.randn() if random_shift else 0.0)
        hshift.append(rng.random() if random_shift else 0.5)
    f = lambda v: np.asarray(v, dtype=np.float64)
    return np.stack(choice).astype(np.int32), f(coin), f(normal), f(hshift)


class SunrgbdInputBuilder:
    """Batch construction of datasets/provider_sample_sunrgbd.py::ProviderDataset (+ default_collate) on the device: five
    strides, window centres through the camera matrix K and the tilt rotation Rtilt.  Records: points (n,>=3) float32 in
    upright camera coordinates, seg (n,), box2d (4,), K (3,3), Rtilt (3,3), box3d (8,3), heading, size (l,w,h),
    frustum_angle, type."""

    def __init__(self, npoints, strides=None, max_depth=None, random_flip=False, random_shift=False, one_hot=True,
                 device="cuda"):
        self.npoints = int(npoints)
        self.strides = tuple(float(s) for s in (cfg.DATA.STRIDE if strides is None else strides))
        self.max_depth = float(cfg.DATA.MAX_DEPTH if max_depth is None else max_depth)
        assert len(self.strides) == 5
        self.L = [len(np.arange(0, self.max_depth, s)) for s in self.strides]
        self.random_flip, self.random_shift, self.one_hot = bool(random_flip), bool(random_shift), bool(one_hot)
        self.device = torch.device(device)
        self.classes = DATASET_INFO["SUNRGBD"].CLASSES

    def build(self, records, draws=None, with_seg=True):
        """draws: (choice (B,N) int32, coin (B), normal (B), hshift (B)) or None to draw like the reference."""
        if self.device.type != "cuda":
            raise RuntimeError("frustum_convnet_amd: input construction is a HIP kernel (MI355X only); no CPU fallback")
        B, N = len(records), self.npoints
        counts = [len(r["points"]) for r in records]
        if draws is None:
            draws = draw_sunrgbd(counts, N, self.random_flip, self.random_shift)
        choice, coin, normal, hshift = draws
        stride = int(records[0]["points"].shape[1])
        raw = np.concatenate([np.ascontiguousarray(r["points"], dtype=np.float32) for r in records], 0)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        f64 = lambda k, shape: np.stack([np.asarray(r[k], dtype=np.float64).reshape(shape) for r in records])
        dev = self.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)
        t = {"raw": up(raw), "off": up(off), "choice": up(np.asarray(choice, dtype=np.int32)),
             "fangle": up(f64("frustum_angle", ())), "box2d": up(f64("box2d", (4,))), "K": up(f64("K", (9,))),
             "R": up(f64("Rtilt", (9,))), "corners": up(f64("box3d", (24,))), "heading": up(f64("heading", ())),
             "size": up(f64("size", (3,))), "coin": up(np.asarray(coin, dtype=np.float64)),
             "normal": up(np.asarray(normal, dtype=np.float64)), "hshift": up(np.asarray(hshift, dtype=np.float64))}
        seg_raw = None
        if with_seg:
            seg_raw = up(np.concatenate([np.asarray(r["seg"]).astype(np.int64) for r in records], 0))
        f32 = dict(dtype=torch.float32, device=dev)
        out = {"point_cloud": torch.empty((B, 3, N), **f32), "rot_angle": torch.empty((B, 1), **f32),
               "cls_label": torch.empty((B, self.L[1]), dtype=torch.int64, device=dev),
               "box3d_center": torch.empty((B, 3), **f32), "box3d_heading": torch.empty((B, 1), **f32),
               "box3d_size": torch.empty((B, 3), **f32)}
        for s in range(5):
            out["center_ref%d" % (s + 1)] = torch.empty((B, 3, self.L[s]), **f32)
        if with_seg:
            out["seg_label"] = torch.empty((B, N), dtype=torch.int64, device=dev)
        desc = Inp5Desc(B, N, stride, (ctypes.c_int32 * 5)(*self.L), (ctypes.c_double * 5)(*self.strides), self.max_depth,
                        1 if self.random_flip else 0, 1 if self.random_shift else 0)
        refs = (ctypes.c_void_p * 5)(*[out["center_ref%d" % (s + 1)].data_ptr() for s in range(5)])
        p = lambda x: None if x is None else x.data_ptr()
        L = _native.lib()
        with torch.cuda.device(dev):
            _native.check(L.fcn_prepare_inputs_sunrgbd(
                ctypes.byref(desc), p(t["raw"]), p(t["off"]), p(seg_raw), p(t["choice"]), p(t["fangle"]), p(t["box2d"]),
                p(t["K"]), p(t["R"]), p(t["corners"]), p(t["heading"]), p(t["size"]), p(t["coin"]), p(t["normal"]),
                p(t["hshift"]), p(out["point_cloud"]), refs, p(out["cls_label"]), p(out["box3d_center"]),
                p(out["box3d_heading"]), p(out["box3d_size"]), p(out["rot_angle"]), p(out.get("seg_label")),
                _native.current_stream(dev)), "fcn_prepare_inputs_sunrgbd")
        for v in t.values():
            v.record_stream(torch.cuda.current_stream(dev))
        size_class = [self.classes.index(r["type"]) for r in records]
        out["size_class"] = torch.tensor(size_class, dtype=torch.int64).view(B, 1).to(dev, non_blocking=True)
        if self.one_hot:
            oh = np.zeros((B, len(self.classes)), dtype=np.float32)
            oh[np.arange(B), size_class] = 1.0
            out["one_hot"] = up(oh)
        return out


def sunrgbd_records_from_fixture(g):
    """tests/golden/inputs_sunrgbd_b6.npz (make_golden_inputs_sunrgbd.py) as a list of records."""
    offs = np.concatenate([[0], np.cumsum(g["raw_counts"])])
    recs = []
    for b in range(len(g["raw_counts"])):
        sl = slice(int(offs[b]), int(offs[b + 1]))
        recs.append({"points": g["raw_points"][sl], "seg": g["raw_seg"][sl], "box2d": g["box2d"][b], "K": g["K"][b],
                     "Rtilt": g["Rtilt"][b], "box3d": g["box3d_corners"][b], "heading": float(g["heading"][b]),
                     "size": g["size"][b], "frustum_angle": float(g["frustum_angle"][b]), "type": str(g["types"][b])})
    return recs


def records_from_fixture(g):
    """The golden fixture's packed arrays (tests/golden/make_golden_inputs.py) as a list of records."""
    offs = np.concatenate([[0], np.cumsum(g["raw_counts"])])
    recs = []
    for b in range(len(g["raw_counts"])):
        sl = slice(int(offs[b]), int(offs[b + 1]))
        recs.append({"points": g["raw_points"][sl], "seg": g["raw_seg"][sl], "box2d": g["box2d"][b], "P": g["P"][b],
                     "box3d": g["box3d_corners"][b], "heading": float(g["heading"][b]), "size": g["size"][b],
                     "frustum_angle": float(g["frustum_angle"][b]), "type": "Car"})
    return recs


# ------------------------------------------------------------------------------------------------
REFINE_KEYS = ("points", "box3d", "heading", "size", "pred_box3d", "pred_angle", "pred_size", "type")


def draw_refine(counts, npoints, random_flip=True, random_shift=True, rng=np.random):
    """The refine dataset's per-sample draws in its order (provider_sample_refine.py:209, :268, :282)."""
    return draw(counts, npoints, random_flip, random_shift, rng)


class RefineInputBuilder:
    """Refinement-stage batches (cfgs/refine_car.yaml) from raw records + first-stage predictions, on the device
    (C-ABI fcn_prepare_inputs_refine).  Mirrors datasets/provider_sample_refine.py::ProviderDataset.__getitem__ + collate_fn:
    per-sample window counts differ, center_ref* / cls_label are edge-padded to the batch maximum.  The batch maxima
    (tensor shapes) are decided on the host from the predicted box widths -- B numbers -- everything per point / per window
    runs in the kernel."""

    def __init__(self, npoints, strides=None, random_flip=False, random_shift=False, one_hot=True, device="cuda"):
        self.npoints = int(npoints)
        self.strides = tuple(float(s) for s in (cfg.DATA.STRIDE if strides is None else strides))
        assert len(self.strides) == 4
        self.random_flip, self.random_shift, self.one_hot = bool(random_flip), bool(random_shift), bool(one_hot)
        self.device = torch.device(device)
        self.classes = DATASET_INFO[cfg.DATA.DATASET_NAME].CLASSES
        if not cfg.DATA.RTC:
            # provider_sample_refine.py:225-262: without RTC generate_ref() runs on the UN-rotated predicted box (z extent, a
            # line through the box) and rot_angle / ref_center are zeros -- a different geometry from the kernel's
            raise NotImplementedError("RefineInputBuilder implements the cfg.DATA.RTC = True geometry only (every shipped "
                                      "refine cfg sets it: cfgs/refine_car.yaml, cfgs/refine_people.yaml)")

    def build(self, records, draws=None, with_labels=True):
        """records: dicts with REFINE_KEYS (points (n,>=3) float32 rect camera coordinates; box3d (8,3), heading, size (l,w,h)
        of the label box; pred_box3d (8,3), pred_angle, pred_size of the first-stage prediction; type).  Returns the batch
        dict on the device (+ 'lens' (B,4) int32: the per-sample window counts before padding)."""
        if self.device.type != "cuda":
            raise RuntimeError("frustum_convnet_amd: input construction is a HIP kernel (MI355X only); no CPU fallback")
        from ._native import InpRefineDesc
        B, N = len(records), self.npoints
        counts = [len(r["points"]) for r in records]
        if draws is None:
            draws = draw_refine(counts, N, self.random_flip and with_labels, self.random_shift and with_labels)
        choice, coin, normal = draws
        stride = int(records[0]["points"].shape[1])
        raw = np.concatenate([np.ascontiguousarray(r["points"], dtype=np.float32) for r in records], 0)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        f64 = lambda k, shape: np.stack([np.asarray(r[k], dtype=np.float64).reshape(shape) for r in records])
        psize = f64("pred_size", (3,))
        # batch maxima of len(np.arange(-w/2, w/2, s)): the padded widths of the outputs
        Lpad = [int(max(len(np.arange(-w / 2.0, w / 2.0, s)) for w in psize[:, 1])) for s in self.strides]
        dev = self.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)
        t = {"raw": up(raw), "off": up(off), "choice": up(np.asarray(choice, dtype=np.int32)),
             "pcorners": up(f64("pred_box3d", (24,))), "pangle": up(f64("pred_angle", ())), "psize": up(psize)}
        if with_labels:
            t.update(corners=up(f64("box3d", (24,))), heading=up(f64("heading", ())), size=up(f64("size", (3,))),
                     coin=up(np.asarray(coin, dtype=np.float64)), normal=up(np.asarray(normal, dtype=np.float64)))
        f32 = dict(dtype=torch.float32, device=dev)
        out = {"point_cloud": torch.empty((B, 3, N), **f32), "rot_angle": torch.empty((B, 1), **f32),
               "ref_center": torch.empty((B, 3), **f32), "lens": torch.empty((B, 4), dtype=torch.int32, device=dev)}
        if with_labels:
            out.update(cls_label=torch.empty((B, Lpad[1]), dtype=torch.int64, device=dev),
                       box3d_center=torch.empty((B, 3), **f32), box3d_heading=torch.empty((B, 1), **f32),
                       box3d_size=torch.empty((B, 3), **f32))
        for s in range(4):
            out["center_ref%d" % (s + 1)] = torch.empty((B, 3, Lpad[s]), **f32)
        desc = InpRefineDesc(B, N, stride, (ctypes.c_int32 * 4)(*Lpad), (ctypes.c_double * 4)(*self.strides),
                             1 if (self.random_flip and with_labels) else 0, 1 if (self.random_shift and with_labels) else 0)
        refs = (ctypes.c_void_p * 4)(*[out["center_ref%d" % (s + 1)].data_ptr() for s in range(4)])
        p = lambda x: None if x is None else x.data_ptr()
        L = _native.lib()
        with torch.cuda.device(dev):
            _native.check(L.fcn_prepare_inputs_refine(
                ctypes.byref(desc), p(t["raw"]), p(t["off"]), p(t["choice"]), p(t["pcorners"]), p(t["pangle"]), p(t["psize"]),
                p(t.get("corners")), p(t.get("heading")), p(t.get("size")), p(t.get("coin")), p(t.get("normal")),
                p(out["point_cloud"]), refs, p(out.get("cls_label")), p(out.get("box3d_center")), p(out.get("box3d_heading")),
                p(out.get("box3d_size")), p(out["rot_angle"]), p(out["ref_center"]), p(out["lens"]),
                _native.current_stream(dev)), "fcn_prepare_inputs_refine")
        for v in t.values():
            v.record_stream(torch.cuda.current_stream(dev))
        size_class = [self.classes.index(r["type"]) for r in records]
        if with_labels:
            out["size_class"] = torch.tensor(size_class, dtype=torch.int64).view(B, 1).to(dev, non_blocking=True)
        if self.one_hot:
            oh = np.zeros((B, len(self.classes)), dtype=np.float32)
            oh[np.arange(B), size_class] = 1.0
            out["one_hot"] = up(oh)
        if not with_labels:
            # from_rgb_detection path (provider_sample_refine.py:404-419): the 2-D detector's score travels with the sample and
            # becomes the detection score's prior in detect() (train/test_net_det.py:214,279)
            out["rgb_prob"] = up(np.asarray([float(r.get("prob", 1.0)) for r in records], dtype=np.float32).reshape(B, 1))
        return out


def refine_records_from_fixture(g):
    """tests/golden/inputs_refine_b6.npz (make_golden_inputs_refine.py) as a list of records."""
    offs = np.concatenate([[0], np.cumsum(g["raw_counts"])])
    recs = []
    for b in range(len(g["raw_counts"])):
        sl = slice(int(offs[b]), int(offs[b + 1]))
        recs.append({"points": g["raw_points"][sl], "box3d": g["box3d_corners"][b], "heading": float(g["heading"][b]),
                     "size": g["size"][b], "pred_box3d": g["pred_corners"][b], "pred_angle": float(g["pred_angle"][b]),
                     "pred_size": g["pred_size"][b], "type": "Car"})
    return recs

"""Category tables the model constructor reads (reference: datasets/dataset_info.py:3-45).
Only the KITTI table is needed by the BASELINE configs; SUN-RGBD is listed for completeness."""
import numpy as np


def _category(name, sizes):
    classes = list(sizes.keys())
    arr = np.zeros((len(classes), 3))
    for i, c in enumerate(classes):
        arr[i, :] = sizes[c]
    return type(name, (object,), {
        "CLASSES": classes,
        "CLASS_MEAN_SIZE": {k: np.array(v) for k, v in sizes.items()},
        "NUM_SIZE_CLUSTER": len(classes),
        "MEAN_SIZE_ARRAY": arr,
    })


KITTICategory = _category("KITTICategory", {
    "Car": [3.88311640418, 1.62856739989, 1.52563191462],
    "Pedestrian": [0.84422524, 0.66068622, 1.76255119],
    "Cyclist": [1.76282397, 0.59706367, 1.73698127],
})

SUNRGBDCategory = _category("SUNRGBDCategory", {
    "bathtub": [0.765840, 1.398258, 0.472728], "bed": [2.114256, 1.620300, 0.927272],
    "bookshelf": [0.404671, 1.071108, 1.688889], "chair": [0.591958, 0.552978, 0.827272],
    "desk": [0.695190, 1.346299, 0.736364], "dresser": [0.528526, 1.002642, 1.172878],
    "night_stand": [0.500618, 0.632163, 0.683424], "sofa": [0.923508, 1.867419, 0.845495],
    "table": [0.791118, 1.279516, 0.718182], "toilet": [0.699104, 0.454178, 0.756250],
})

DATASET_INFO = {"KITTI": KITTICategory, "SUNRGBD": SUNRGBDCategory}

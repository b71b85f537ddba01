"""Writes tests/golden/cv2_resize.npz -- cv2.resize(INTER_LINEAR) outputs for the crop sizes the PEM pre-processing meets.

cv2 is an un-vendored dependency of the reference (Pose_Estimation_Model/run_inference_custom.py:234) and is NOT in the build
image: run this wherever `import cv2` works (any OpenCV 4.x wheel), commit the file, and tests/test_host_pem_pre.py::
test_cv2_vectors_if_present pins oracle/pem_pre.py::cv2_resize_linear_u8 (and through it the device kernel) to it.  Inputs are
regenerated from the seeds stored in the file, so only the outputs' digests and a few whole outputs are stored."""
import hashlib
import sys

import numpy as np

SIDES = (224, 448, 112, 100, 333, 37, 500, 3, 1, 223, 225, 447, 449, 640, 96)


def image(side, seed):
    return np.random.default_rng(seed).integers(0, 256, (side, side, 3), dtype=np.uint8)


def main():
    import cv2
    rec = dict(sides=np.array(SIDES), version=np.array(cv2.__version__))
    for s in SIDES:
        out = cv2.resize(image(s, 1000 + s), (224, 224), interpolation=cv2.INTER_LINEAR)
        rec[f"sha_{s}"] = np.array(hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest())
        if s in (100, 333, 448):
            rec[f"out_{s}"] = out
    np.savez_compressed(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/cv2_resize.npz", **rec)
    print("wrote", len(SIDES), "cases with OpenCV", cv2.__version__)


if __name__ == "__main__":
    main()

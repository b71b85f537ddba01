"""Pose result writers of the Pose Estimation Model (the PEM end of the file boundary).

The reference emits poses in two formats, both written inline in its entry scripts:
  * BOP result lines ``scene,im,obj,score,R00 .. R22,tx ty tz,time`` -- Pose_Estimation_Model/test_bop.py:155-181: R and
    t leave the model as float32, t is scaled to millimetres IN float32 (``ndarray * 1000``), the pose score is multiplied
    by the detection score in float32, and every value is printed with ``str(numpy.float32)`` (the shortest decimal that
    round-trips float32, e.g. ``0.1`` rather than the double expansion ``0.10000000149011612``); the time is a Python
    float printed by an f-string;
  * ``detection_pem.json`` -- Pose_Estimation_Model/run_inference_custom.py:290-307: the ISM detection records with
    ``score`` replaced by pose_score * detection score, ``R`` (3x3 nested list) and ``t`` (millimetres, float32 scaled then
    widened by ``tolist()``) added.

Here both are produced from the tensors the device path returns (one device->host copy per frame for all instances);
given identical float32 inputs the text is byte-identical with what the reference's statements write (golden:
tests/golden/pem_results.npz, made by executing those statements from the reference's files).
"""
import json

import numpy as np
import torch


def _f32(x):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float32)


def combined_scores(pose_score, det_score):
    """pred_pose_score * detection score in float32 (test_bop.py:157, run_inference_custom.py:290-293)."""
    return _f32(pose_score) * _f32(det_score)


def bop_csv_lines(scene_id, im_id, obj_ids, scores, R, t_m, time_s):
    """One line per instance of one frame.  scores (n,) float32 (already combined), R (n,3,3), t_m (n,3) metres."""
    R = _f32(R).reshape(-1, 9)
    t = _f32(t_m) * 1000
    s = _f32(scores)
    n = R.shape[0]
    if not (t.shape == (n, 3) and s.shape == (n,) and len(obj_ids) == n):
        raise ValueError(f"inconsistent instance counts: R {R.shape}, t {t.shape}, scores {s.shape}, obj_ids {len(obj_ids)}")
    out = []
    for k in range(n):
        out.append(",".join((str(int(scene_id)), str(int(im_id)), str(int(obj_ids[k])), str(s[k]),
                             " ".join(str(v) for v in R[k]), " ".join(str(v) for v in t[k]), f"{float(time_s)}\n")))
    return out


def write_bop_csv(path, lines):
    """test_bop.py:183-184 (mode 'w+', lines already newline-terminated)."""
    with open(path, "w+") as f:
        f.writelines(lines)


def detection_pem_records(detections, scores, R, t_m):
    """detections: list of dicts (the ISM JSON records of the frame, see sam6d_amd.ism.handoff.detection_records); returns
    NEW dicts with score / R / t filled the way run_inference_custom.py:301-304 fills them."""
    R = _f32(R).reshape(-1, 3, 3)
    t = _f32(t_m) * 1000
    s = _f32(scores)
    if not (len(detections) == R.shape[0] == t.shape[0] == s.shape[0]):
        raise ValueError(f"{len(detections)} detections but R {R.shape}, t {t.shape}, scores {s.shape}")
    out = []
    for i, det in enumerate(detections):
        d = dict(det)
        d["score"] = float(s[i])
        d["R"] = list(R[i].tolist())
        d["t"] = list(t[i].tolist())
        out.append(d)
    return out


def save_detection_pem(path, records):
    with open(path, "w") as f:
        json.dump(records, f)

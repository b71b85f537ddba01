"""Pose result writers (PEM end) against the reference's own statements: the BOP csv lines of
test_bop.py and detection_pem.json of run_inference_custom.py, byte for byte (golden: oracle/gen_golden.py pem_results,
which executes those statements from the reference's files)."""
import json

import numpy as np
import pytest
import torch

from sam6d_amd.pem import results
from sam6d_amd.utils import shard
from tests import util


def _g():
    g = util.golden("pem_results.npz")
    return g, torch.from_numpy(g["R"]), torch.from_numpy(g["t"]), torch.from_numpy(g["pose_score"]), torch.from_numpy(g["det_score"])


def test_bop_csv_lines_are_byte_identical():
    g, R, t, ps, ds = _g()
    s = results.combined_scores(ps, ds)
    lines = results.bop_csv_lines(48, 1003, g["obj"], s, R, t, 0.375)
    assert "".join(lines) == str(g["csv"])
    assert lines[0].split(",")[3] == "0.1"                          # float32 shortest form, not 0.10000000149011612
    assert results.bop_csv_lines(48, 1003, [], s[:0], R[:0], t[:0], 0.0) == []
    with pytest.raises(ValueError):
        results.bop_csv_lines(48, 1003, g["obj"][:2], s, R, t, 0.0)


def test_sharded_record_table_writes_the_same_lines():
    """The gathered (n,17) float32 record table of utils/shard.py prints the same text (its columns are float32)."""
    g, R, t, ps, ds = _g()
    s = torch.from_numpy(results.combined_scores(ps, ds))
    rec = shard.pack_records(48, 1003, torch.from_numpy(g["obj"]).float(), s, R, t, 0.375)
    assert "".join(shard.to_bop_csv_lines(rec)) == str(g["csv"])


def test_detection_pem_json_is_byte_identical(tmp_path):
    g, R, t, ps, ds = _g()
    dets = json.loads(str(g["dets_json"]))
    keep = json.dumps(dets)
    recs = results.detection_pem_records(dets, results.combined_scores(ps, ds), R, t)
    assert json.dumps(dets) == keep                                  # inputs untouched
    assert json.dumps(recs) == str(g["pem_json"])
    results.save_detection_pem(tmp_path / "detection_pem.json", recs)
    assert open(tmp_path / "detection_pem.json").read() == str(g["pem_json"])
    with pytest.raises(ValueError):
        results.detection_pem_records(dets[:2], ps, R, t)


def test_csv_file_round_trip(tmp_path):
    g, R, t, ps, ds = _g()
    lines = results.bop_csv_lines(1, 2, g["obj"], results.combined_scores(ps, ds), R, t, 1.5)
    results.write_bop_csv(tmp_path / "r.csv", lines)
    rows = [l.split(",") for l in open(tmp_path / "r.csv").read().splitlines()]
    assert len(rows) == len(g["obj"])
    back = np.array([[float(v) for v in r[4].split()] for r in rows], dtype=np.float32)
    assert np.array_equal(back, g["R"].reshape(-1, 9))              # the printed decimals round-trip float32 exactly


def test_frame_results_joins_both_stages():
    """pipeline.frame_results: the PEM records are the ISM records of the kept detections with score / R / t replaced."""
    from sam6d_amd.ism.handoff import Detections
    from sam6d_amd.pipeline import frame_results
    g, R, t, ps, ds = _g()
    n = 5
    gen = torch.Generator().manual_seed(3)
    masks = torch.rand(n + 2, 12, 16, generator=gen) > 0.5
    boxes = torch.tensor([[0, 0, 5, 5]] * (n + 2))
    det = Detections(3, 9, masks, boxes, torch.cat([ds, torch.tensor([0.3, 0.2])]), torch.tensor([0, 1, 2, 3, 7, 4, 6]), 0.5)
    kept = torch.tensor([0, 1, 2, 3, 4])
    out = frame_results(det, dict(pred_R=R, pred_t=t, pred_pose_score=ps, kept=kept), "lmo", 0.375)
    assert len(out["ism_records"]) == n + 2 and len(out["csv_lines"]) == n and len(out["pem_records"]) == n
    ref_lines = str(g["csv"]).splitlines(keepends=True)
    for k, line in enumerate(out["csv_lines"]):                     # same numbers as the golden lines, other ids
        assert line.split(",")[3:] == ref_lines[k].split(",")[3:]
        assert line.split(",")[:3] == ["3", "9", str(out["ism_records"][k]["category_id"])]
    for k, r in enumerate(out["pem_records"]):
        assert r["segmentation"] == out["ism_records"][k]["segmentation"] and np.allclose(r["R"], R[k].numpy())
    none = frame_results(det, None, "lmo")
    assert none["csv_lines"] == [] and len(none["ism_records"]) == n + 2

"""Pins oracle/maf_oracle.py against the fixtures the REFERENCE produced (tools/make_golden.py).

CPU only.  Tolerances: weights/activations are fp32 on both sides, only the accumulation order of
the re-parameterised kernels differs.  Box coordinates reach ~1e3 px for these synthetic weights
(ltrb distances up to 16 bins x stride 32), where one fp32 ulp is 6e-5, so boxes are compared with
|d| <= 1e-3 + 1e-5*|ref| (north_star: "within 1e-3 fp32") and scores with 2e-5 abs; NMS rows exactly.
"""
import hashlib
import json

import numpy as np
import pytest
import torch

from oracle import maf_oracle as O
import nms_cases

SCALES = ("n", "s", "m")


def _wsum(t):
    a = t.detach().double().reshape(-1).numpy()
    ramp = (np.arange(a.size) % 97 + 1).astype(np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * ramp).sum()])


def _close(pred, ref):
    np.testing.assert_allclose(pred[..., :4], ref[..., :4], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(pred[..., 4:], ref[..., 4:], rtol=0, atol=2e-5)


@pytest.fixture(scope="module")
def nets():
    torch.set_num_threads(8)
    out = {}
    for s in SCALES:
        sd = O.synth_state_dict(s, 0)
        out[s] = (sd, O.reparam(sd, s))
    return out


@pytest.mark.parametrize("scale", SCALES)
def test_state_dict_layout_matches_reference(golden, scale):
    g = golden("maf_" + scale)
    spec = O.state_spec(scale)
    h = hashlib.sha256(json.dumps([[k, list(s)] for k, s in spec]).encode()).hexdigest()
    assert h == str(g["spec_hash"])
    assert len(spec) == int(g["n_train_tensors"]) == {"n": 838, "s": 1206, "m": 1568}[scale]


@pytest.mark.parametrize("scale", SCALES)
def test_reparam_matches_reference_deploy_switch(golden, nets, scale):
    g = golden("maf_" + scale)
    dw = nets[scale][1]
    names = [str(n) for n in g["deploy_names"]]
    assert sorted(dw.keys()) == names
    for i, n in enumerate(names):
        w, b = dw[n]
        np.testing.assert_allclose(_wsum(w), g["deploy_wsum"][i], rtol=2e-5, atol=2e-4)
        np.testing.assert_allclose(_wsum(b), g["deploy_bsum"][i], rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("scale", SCALES)
def test_deploy_forward_and_decode_320(golden, nets, scale):
    g = golden("maf_" + scale)
    with torch.no_grad():
        heads = O.forward_deploy(nets[scale][1], scale, O.synth_images(1, 320, 1))
        pred = O.decode(heads).numpy()
    ref = g["pred320_deploy"]
    assert pred.shape == ref.shape == (1, 2100, 85)
    _close(pred, ref)
    # decode alone, from the reference's raw head tensors
    t = torch.zeros(1, 1, 10, 10)
    d = O.decode([(t, torch.from_numpy(g["head2_cls"]), torch.from_numpy(g["head2_reg"]))], strides=(32,)).numpy()
    _close(d, ref[:, 2000:])


@pytest.mark.parametrize("scale", SCALES)
def test_train_form_forward_320(golden, nets, scale):
    g = golden("maf_" + scale)
    with torch.no_grad():
        pred = O.decode(O.forward_train_form(nets[scale][0], scale, O.synth_images(1, 320, 1))).numpy()
    ref = g["pred320_train_rows7"]
    _close(pred[:, ::7], ref)


def test_headline_shape_640(golden, nets):
    g = golden("maf_n")
    with torch.no_grad():
        pred = O.predict(nets["n"][1], "n", O.synth_images(2, 640, 1))
    assert pred.shape == (2, 8400, 85)
    _close(pred[:, ::16].numpy(), g["pred640_rows16"])
    np.testing.assert_allclose(pred.double().sum(1).numpy(), g["pred640_colsum"], rtol=1e-5, atol=1e-2)


def test_config0_batch8_640(golden, nets):
    """BASELINE configs[0] at its stated batch (8 x 3 x 640 x 640): the oracle's prediction and detections against the reference's own (tools/make_golden_b8.py)."""
    g = golden("maf_n_b8")
    with torch.no_grad():
        pred = O.predict(nets["n"][1], "n", O.synth_images(8, 640, 1))
    assert pred.shape == (8, 8400, 85)
    _close(pred[:, ::64].numpy(), g["pred640_b8_rows64"])
    np.testing.assert_allclose(pred.double().sum(1).numpy(), g["pred640_b8_colsum"], rtol=1e-5, atol=1e-2)
    dets = O.non_max_suppression(pred.numpy(), 0.03, 0.65, multi_label=True)
    assert [d.shape[0] for d in dets] == list(g["nms640_b8_n"])
    from nms_cases import match_detections
    for b, d in enumerate(dets):      # one-to-one by class, IoU >= 0.99 and |d score| <= 1e-3: the oracle's fp32 prediction is ~1e-5 from the reference's, so rows of near-equal
        ref = g["nms640_b8_%d" % b]   # score may swap places in the list (seen on 3 of the 8 images) — every row still has its partner
        pairs, miss, extra = match_detections(d, ref, iou_min=0.99, dscore=1e-3)
        assert len(pairs) >= ref.shape[0] - 2, (b, len(pairs), miss, extra)


def test_per_node_taps_64(golden, nets):
    g = golden("maf_n")
    taps = {}
    with torch.no_grad():
        O.forward_deploy(nets["n"][1], "n", O.synth_images(1, 64, 2), taps)
    checked = 0
    for i, t in taps.items():
        if isinstance(t, tuple):
            for j, u in enumerate(t):
                np.testing.assert_allclose(u.numpy(), g["tap64_%d_%d" % (i, j)], rtol=1e-4, atol=2e-5); checked += 1
        else:
            np.testing.assert_allclose(t.numpy(), g["tap64_%d" % i], rtol=1e-4, atol=2e-5); checked += 1
    assert checked == 31 + 9


@pytest.mark.parametrize("scale", SCALES)
def test_nms_on_reference_predictions(golden, scale):
    g = golden("maf_" + scale)
    pred = g["pred320_deploy"]
    for tag, kw in (("eval", dict(conf_thres=0.03, iou_thres=0.65, multi_label=True)),
                    ("infer", dict(conf_thres=0.1, iou_thres=0.45, agnostic=True, max_det=1000)),
                    ("best", dict(conf_thres=0.05, iou_thres=0.45))):
        out = O.non_max_suppression(pred, **kw)
        assert np.array_equal(out[0], g["nms320_%s" % tag]), (scale, tag)


@pytest.mark.parametrize("name", sorted(nms_cases.cases().keys()))
def test_nms_edge_cases(golden, name):
    g = golden("nms_cases")
    pred, kw = nms_cases.cases()[name]
    out = O.non_max_suppression(pred, **kw)
    assert [o.shape[0] for o in out] == list(g[name + "__n"])
    for bi, o in enumerate(out):
        assert o.dtype == np.float32 and o.shape[1] == 6
        assert np.array_equal(o, g["%s__%d" % (name, bi)]), (name, bi)


def test_nms_threshold_asserts():
    with pytest.raises(AssertionError):
        O.non_max_suppression(np.zeros((1, 4, 85), np.float32), conf_thres=1.5)
    with pytest.raises(AssertionError):
        O.non_max_suppression(np.zeros((1, 4, 85), np.float32), iou_thres=-0.1)


def test_greedy_nms_rule():
    # descending-score order, strict '>' on IoU, stable ties
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [20, 20, 30, 30], [0, 0, 10, 5]], np.float32)
    s = np.array([0.5, 0.5, 0.9, 0.4], np.float32)
    assert list(O.greedy_nms(b, s, 0.5)) == [2, 0, 3]       # box 1 == box 0 (IoU 1); box 3 has IoU 0.5, NOT > 0.5
    assert list(O.greedy_nms(b, s, 0.49)) == [2, 0]
    assert list(O.greedy_nms(b[[0, 3]], s[[0, 3]], 0.5)) == [0, 1]
    assert list(O.greedy_nms(b[[0, 3]], s[[0, 3]], 0.4999)) == [0]


def test_post_nms_tail_oracle_matches_reference_fixture(golden):
    """oracle.coco_rows == the reference's Evaler.convert_to_coco_format on the vectors of tools/make_golden_post.py."""
    g = golden("post_cases")
    ids = g["ids"]
    for ci in range(3):
        counts = g["c%d_counts" % ci]
        dets = g["c%d_dets" % ci]
        outs, o = [], 0
        for n in counts:
            outs.append(dets[o:o + n]); o += n
        shapes = [((s[0], s[1]), ((s[2], s[3]), (s[4], s[5]))) for s in g["c%d_shapes" % ci]]
        iid, cid, bb, sc = O.coco_rows(outs, shapes, g["c%d_image_ids" % ci], ids, bool(g["c%d_scale_exact" % ci]))
        assert np.array_equal(iid, g["c%d_out_image_id" % ci]) and np.array_equal(cid, g["c%d_out_category_id" % ci])
        assert np.array_equal(bb, g["c%d_out_bbox" % ci]) and np.array_equal(sc, g["c%d_out_score" % ci])


def test_training_loss_oracle_matches_reference_fixture(golden):
    """oracle.compute_loss (per-image, per-box task-aligned assignment + VFL/GIoU/DFL) == the reference's ComputeLoss on the
    vectors of tools/make_golden_loss.py: loss, the three weighted items and the gradients w.r.t. both head outputs."""
    g = golden("loss_cases")
    for ci in range(4):
        size = int(g["c%d_size" % ci])
        hw = [(size // s, size // s) for s in (8, 16, 32)]
        s = torch.from_numpy(g["c%d_scores" % ci]).requires_grad_(True)
        d = torch.from_numpy(g["c%d_distri" % ci]).requires_grad_(True)
        loss, items = O.compute_loss(hw, s, d, torch.from_numpy(g["c%d_targets" % ci]), img_size=size)
        want = float(g["c%d_loss" % ci])
        if not np.isfinite(want):                           # no ground truth in the whole batch: the reference divides by zero too
            assert not np.isfinite(loss.item())
            continue
        assert abs(loss.item() - want) <= 2e-5 * abs(want)
        assert np.allclose(items.numpy(), g["c%d_items" % ci], rtol=2e-5, atol=1e-6)
        loss.backward()
        gs, gd = g["c%d_gscores" % ci], g["c%d_gdistri" % ci]
        assert np.abs(s.grad.numpy() - gs).max() <= 2e-4 * np.abs(gs).max() and np.abs(d.grad.numpy() - gd).max() <= 2e-4 * np.abs(gd).max()


def test_training_loss_oracle_atss_matches_reference_fixture(golden):
    """The warm-up branch (ATSS, epoch < warmup_epoch) of ComputeLoss: oracle.compute_loss(assigner="atss") == the reference at epoch 0."""
    g = golden("loss_cases")
    for ci in range(1, 4):                  # case 0 (64 x 64: 4 anchors on the last level) makes the reference's ATSS raise
        size = int(g["c%d_size" % ci])
        hw = [(size // s, size // s) for s in (8, 16, 32)]
        s = torch.from_numpy(g["c%d_scores" % ci]).requires_grad_(True)
        d = torch.from_numpy(g["c%d_distri" % ci]).requires_grad_(True)
        loss, items = O.compute_loss(hw, s, d, torch.from_numpy(g["c%d_targets" % ci]), img_size=size, assigner="atss")
        want = float(g["a%d_loss" % ci])
        if not np.isfinite(want):
            assert not np.isfinite(loss.item())
            continue
        assert abs(loss.item() - want) <= 2e-5 * abs(want)
        assert np.allclose(items.numpy(), g["a%d_items" % ci], rtol=2e-5, atol=1e-6)
        loss.backward()
        gs, gd = g["a%d_gscores" % ci].astype(np.float32), g["a%d_gdistri" % ci]
        assert np.abs(s.grad.numpy() - gs).max() <= 1e-3 * np.abs(gs).max() and np.abs(d.grad.numpy() - gd).max() <= 2e-4 * np.abs(gd).max()

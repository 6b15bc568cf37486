"""CPU tests of the host side: hyper-parameter table (mirrors the reference's tests/test_train_utils.py),
the C-ABI surface (library loads, exports every symbol of include/frcnn_hip.h, no compute without a GPU),
layer configs, sharding arithmetic and the fp32 IoU band filter's logic."""

import ctypes
import os
import re

import numpy as np
import pytest

import tf_faster_rcnn_amd as pkg
from tf_faster_rcnn_amd import _lib, sharding
from tf_faster_rcnn_amd.utils import train_utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


# ---- reference tests/test_train_utils.py:27-58, restated ------------------------------------------------
def test_get_hyper_params_populates_defaults_and_anchor_count():
    params = train_utils.get_hyper_params("vgg16")
    assert params["img_size"] == 500
    assert params["pre_nms_topn"] == 6000
    assert params["train_nms_topn"] == 1500
    assert params["test_nms_topn"] == 300
    assert params["anchor_count"] == 9
    assert params["feature_map_shape"] == 31 and train_utils.get_hyper_params("mobilenet_v2")["feature_map_shape"] == 32
    assert params["nms_iou_threshold"] == 0.7 and params["pooling_size"] == (7, 7)
    assert params["variances"] == [0.1, 0.1, 0.2, 0.2]
    assert params["total_pos_bboxes"] == 128 and params["total_neg_bboxes"] == 128


def test_get_hyper_params_applies_truthy_overrides():
    params = train_utils.get_hyper_params("mobilenet_v2", img_size=320, pre_nms_topn=1024, total_pos_bboxes=64)
    assert params["img_size"] == 320 and params["pre_nms_topn"] == 1024 and params["total_pos_bboxes"] == 64
    assert params["anchor_count"] == 9
    # falsy values and unknown keys are ignored (utils/train_utils.py:52-54)
    params = train_utils.get_hyper_params("mobilenet_v2", img_size=0, not_a_key=3)
    assert params["img_size"] == 500 and "not_a_key" not in params
    # BASELINE config 5 is reached through overrides
    p5 = train_utils.get_hyper_params("mobilenet_v2", img_size=1000, feature_map_shape=63, pre_nms_topn=12000)
    assert p5["feature_map_shape"] ** 2 * p5["anchor_count"] == 35721


def test_get_hyper_params_returns_a_copy_of_backbone_defaults():
    params = train_utils.get_hyper_params("vgg16")
    params["img_size"] = 320
    assert train_utils.RPN["vgg16"]["img_size"] == 500
    with pytest.raises(KeyError):
        train_utils.get_hyper_params("resnet50")


def test_get_step_size_rounds_up_for_partial_batches():
    assert train_utils.get_step_size(10, 4) == 3
    assert train_utils.get_step_size(8, 4) == 2


# ---- C ABI ------------------------------------------------------------------------------------------------
def _header_symbols():
    text = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(frcnn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = _header_symbols()
    assert len(syms) >= 19
    handle = ctypes.CDLL(pkg.library_path())
    for name in syms:
        assert hasattr(handle, name), f"{name} declared in include/frcnn_hip.h but not exported"
    # the Python binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == syms


def test_binding_table_has_the_header_arity():
    """Every ctypes signature carries as many arguments as the declaration in include/frcnn_hip.h."""
    text = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = dict(re.findall(r"\b(frcnn_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text))
    assert sorted(decls) == sorted(_lib.SIGNATURES)
    for name, params in decls.items():
        n = 0 if params.strip() in ("", "void") else params.count(",") + 1
        assert n == len(_lib.SIGNATURES[name][1]), f"{name}: header has {n} parameters, the binding {len(_lib.SIGNATURES[name][1])}"


def test_multi_entry_points_validate_on_the_host():
    """The grouped entry points reject a bad pointer table / group count before touching a device."""
    lib = _lib.lib()
    one = (ctypes.c_void_p * 1)(None)
    f4 = _lib.float4([0.1, 0.1, 0.2, 0.2])
    assert lib.frcnn_crop_and_resize_multi(None, None, None, 1, 1, 4, 4, 4, 1, 1, 7, 7, 0.0, None, 0, None) == _lib.FRCNN_EINVAL
    assert lib.frcnn_crop_and_resize_multi(one, one, one, 5, 1, 4, 4, 4, 1, 1, 7, 7, 0.0, None, 0, None) == _lib.FRCNN_EINVAL
    assert b"groups" in lib.frcnn_last_error()
    assert lib.frcnn_crop_and_resize_multi(one, one, one, 1, 1, 4, 4, 4, 1, 1, 7, 7, 0.0, None, 0, None) == _lib.FRCNN_EINVAL  # null entry
    assert lib.frcnn_roi_bbox_multi(one, one, None, f4, 0, 1, 9, 4, 2, 0.7, one, None, 0, None) == _lib.FRCNN_EINVAL
    assert lib.frcnn_decoder_multi(one, one, one, f4, 9, 1, 4, 3, 2, 0.5, 0.5, one, one, one, None, 0, None) == _lib.FRCNN_EINVAL


def test_no_torch_types_cross_the_boundary():
    text = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    assert "torch" not in text.lower() and "at::" not in text and "hip/hip_runtime" not in text


def test_version_and_error_string_without_gpu():
    lib = _lib.lib()
    assert lib.frcnn_version() >= 100
    # argument validation happens on the host before anything touches a device
    rc = lib.frcnn_top_k(None, 1, 4, 8, None, None, None)
    assert rc == _lib.FRCNN_EINVAL and b"top_k" in lib.frcnn_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    assert lib.frcnn_roi_bbox_workspace_bytes(16, 9216, 6000, 300) > 16 * 6000 * 4
    assert lib.frcnn_rpn_targets_workspace_bytes(64, 9216, 100) > 64 * 9216 * 8


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.bbox_utils.generate_anchors(train_utils.get_hyper_params("vgg16"))


def test_product_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "tf-faster-rcnn_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "frcnn_oracle" not in text, os.path.join(dirpath, f)


# ---- layers -------------------------------------------------------------------------------------------------
def test_layer_configs_mirror_the_reference():
    hp = train_utils.get_hyper_params("mobilenet_v2")
    dec = pkg.faster_rcnn.Decoder(hp["variances"], 21, name="faster_rcnn_decoder")
    cfg = dec.get_config()
    assert cfg["name"] == "faster_rcnn_decoder" and cfg["total_labels"] == 21
    assert cfg["max_total_size"] == 200 and cfg["score_threshold"] == 0.5 and cfg["variances"] == hp["variances"]
    pool = pkg.faster_rcnn.RoIPooling(hp, name="roi_pooling")
    assert pool.get_config()["hyper_params"] is hp
    with pytest.raises(TypeError):
        pkg.faster_rcnn.RoIPooling(hp, bogus=1)


# ---- sharding -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,P", [(16, 1), (16, 2), (16, 8), (32, 4), (64, 8), (128, 8), (5, 8), (7, 3), (0, 4)])
def test_shard_bounds_partition_the_batch(B, P):
    bounds = [sharding.shard_bounds(B, r, P) for r in range(P)]
    assert bounds[0][0] == 0 and bounds[-1][1] == B
    for (lo, hi), (lo2, _) in zip(bounds, bounds[1:]):
        assert hi == lo2 and hi >= lo
    sizes = [hi - lo for lo, hi in bounds]
    assert max(sizes) - min(sizes) <= 1
    if B % P == 0:  # GPU g gets images [g*B/P, (g+1)*B/P)
        assert bounds == [(g * B // P, (g + 1) * B // P) for g in range(P)]


# ---- the fp32 band filter of the NMS IoU test (nms_core.hpp), restated on the CPU --------------------------
@pytest.mark.parametrize("thr", [0.7, 0.5, 0.3, 0.1, 0.65, 1e-3, 0.999, 0.25, 1.0])
def test_band_filter_never_contradicts_the_division(thr):
    """inter > fl(thr*uni)*(1+2^-21) must imply fl32(inter/uni) > thr, inter < fl(thr*uni)*(1-2^-21) must imply
    the opposite; everything in between goes to the exact path, so only contradictions would be bugs."""
    rng = np.random.default_rng(int(thr * 1000))
    n = 2_000_000
    thr = f32(thr)
    uni = (rng.random(n) * 4 + 1e-3).astype(f32)
    inter = (uni.astype(np.float64) * float(thr)).astype(f32)
    inter = (inter.view(np.int32) + rng.integers(-40, 41, n).astype(np.int32)).view(f32)  # +-40 ulp around the boundary
    ref = (inter / uni) > thr
    tt = thr * uni
    yes = inter > tt * f32(1.000000476837158203125)
    no = inter < tt * f32(0.999999523162841796875)
    assert not (yes & ~ref).any() and not (no & ref).any()
    assert not (yes & no).any()
    assert (~(yes | no)).mean() < 0.5  # the band is narrow: most near-threshold pairs are still decided


@pytest.mark.parametrize("thr", [0.7, 0.5, 0.3, 0.1, 0.65, 1e-3, 0.999, 0.25, 1.0, 1e-6, 37.5, 1e6])
def test_union_free_band_filter_never_contradicts_the_division(thr):
    """The packed form (nms_core.hpp, IouThr::k_hi / k_lo): with s = fl(area_i + area_j), inter > fl(s * k_hi) must imply
    fl32(inter / fl(s - inter)) > thr and inter < fl(s * k_lo) the opposite; k = thr / (1 + thr) widened by 2^-21 in double
    and rounded outwards, as make_iou_thr does."""
    rng = np.random.default_rng(int(thr * 1000) + 7)
    n = 2_000_000
    thr = f32(thr)
    k = float(thr) / (1.0 + float(thr))
    k_hi = np.nextafter(f32(k * (1.0 + 2.0 ** -21)), f32(2.0))
    k_lo = np.nextafter(f32(k * (1.0 - 2.0 ** -21)), f32(0.0))
    s_ = (10.0 ** rng.uniform(-3, 1, n)).astype(f32)
    inter = (s_.astype(np.float64) * k).astype(f32)
    inter = (inter.view(np.int32) + rng.integers(-60, 61, n).astype(np.int32)).view(f32)  # +-60 ulp around the boundary
    # an intersection never exceeds the smaller of the two areas (fp32 products are monotone), i.e. half their sum: for
    # thresholds above 1 the boundary lies beyond that, and only the far side ("no") can be met
    inter = np.minimum(inter, s_ * f32(0.5) * f32(1.0 + 2.0 ** -22))
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = (inter / (s_ - inter)) > thr
    yes = inter > s_ * k_hi
    no = inter < s_ * k_lo
    assert not (yes & ~ref).any() and not (no & ref).any()
    assert not (yes & no).any()
    if thr <= 1:
        assert (~(yes | no)).mean() < 0.6  # the band is narrow: most near-threshold pairs are still decided
    else:
        assert no.all()


# ---- the parallel-round resolve of a 64-candidate block (nms_core.hpp phase c, nms_bitmask.hip scan), restated ----------
def _greedy_block(sup, alive):
    """Sequential greedy NMS inside a block: sup[r][c] = row r suppresses column c (r < c); alive = candidates not yet dead."""
    kept, removed = [], set()
    for i in range(len(sup)):
        if i in alive and i not in removed:
            kept.append(i)
            removed.update(c for c in range(i + 1, len(sup)) if sup[i][c])
    return kept


def _rounds_block(sup, alive):
    """The kernels' form: a candidate hit by a KEPT row is removed; a candidate none of whose still-undecided earlier rows
    suppresses it is kept; repeat until nobody is undecided."""
    n = len(sup)
    col = [{r for r in range(c) if sup[r][c]} for c in range(n)]
    und, kept, rounds = set(alive), set(), 0
    while und:
        rounds += 1
        und -= {c for c in und if col[c] & kept}
        new = {c for c in und if not (col[c] & und)}
        assert new or not und  # progress: the lowest undecided candidate is always decided
        kept |= new
        und -= new
    return sorted(kept), rounds


@pytest.mark.parametrize("density", [0.0, 0.02, 0.1, 0.4, 0.9])
def test_parallel_round_resolve_equals_sequential_greedy(density):
    rng = np.random.default_rng(int(density * 100))
    worst = 0
    for _ in range(200):
        n = int(rng.integers(1, 65))
        sup = np.triu(rng.random((n, n)) < density, 1)
        alive = set(np.flatnonzero(rng.random(n) < 0.8).tolist())
        want = _greedy_block(sup, alive)
        got, rounds = _rounds_block(sup, alive)
        assert got == want
        worst = max(worst, rounds)
    assert worst <= 64

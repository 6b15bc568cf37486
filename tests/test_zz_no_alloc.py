"""frcnn_rpn_targets_static (the entry point that allocates nothing) against frcnn_rpn_targets, through the C ABI.

Two calls in a row on one GPU, the second with a smaller input: the static scratch keeps the first call's contents,
so the second call proves that what must start at zero is cleared.  Bit-exact (integer / index work and fp32 values
computed by the same kernels).  Collected last on purpose (the round's end runs ``pytest -x``).
"""

import ctypes

import numpy as np
import pytest

import synth

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


def _call(pkg, static, anchors, gt, labels, rp, rn, seed, hp):
    L = pkg._lib.lib()
    B, M, _ = gt.shape
    N = anchors.shape[0]
    deltas = torch.full((B, N, 4), float("nan"), dtype=torch.float32, device="cuda")
    lab = torch.full((B, N), float("nan"), dtype=torch.float32, device="cuda")
    var = pkg._lib.float4(hp["variances"])
    stream = torch.cuda.current_stream().cuda_stream
    head = (anchors.data_ptr(), gt.data_ptr(), labels.data_ptr(), None if rp is None else rp.data_ptr(),
            None if rn is None else rn.data_ptr(), seed, B, N, M, hp["total_pos_bboxes"], hp["total_neg_bboxes"], var,
            deltas.data_ptr(), lab.data_ptr())
    if static:
        rc = L.frcnn_rpn_targets_static(*head, stream)
    else:
        ws = torch.empty(L.frcnn_rpn_targets_workspace_bytes(B, N, M), dtype=torch.uint8, device="cuda")
        rc = L.frcnn_rpn_targets(*head, ws.data_ptr(), ws.numel(), stream)
    pkg._lib.check(rc)
    torch.cuda.synchronize()
    return deltas.cpu().numpy(), lab.cpu().numpy()


def _case(pkg, seed, B, G, M):
    hp = pkg.train_utils.get_hyper_params("mobilenet_v2")
    hp.update(feature_map_shape=G)
    anchors = pkg.bbox_utils.generate_anchors(hp)
    gt, labels = synth.gt_boxes(seed, B, M)
    rp, rn = synth.sampling_ints(seed, B, anchors.shape[0])
    cuda = lambda a: torch.as_tensor(a).cuda()  # noqa: E731
    return hp, anchors, cuda(gt), cuda(labels), cuda(rp), cuda(rn)


def test_rpn_targets_static_matches_workspace_entry_point():
    import oracle
    import tf_faster_rcnn_amd as pkg

    # first the larger input (config 4's per-GPU shard), then a smaller one on the same, now dirty, scratch
    for seed, B, G, M in ((3, 8, 32, 100), (4, 3, 16, 7)):
        hp, anchors, gt, labels, rp, rn = _case(pkg, seed, B, G, M)
        for ints in ((rp, rn), (None, None)):  # injected sampling integers, then the in-kernel draw (seeded: deterministic)
            want_d, want_l = _call(pkg, False, anchors, gt, labels, ints[0], ints[1], 77, hp)
            got_d, got_l = _call(pkg, True, anchors, gt, labels, ints[0], ints[1], 77, hp)
            assert np.array_equal(got_l.view(np.uint32), want_l.view(np.uint32)), f"labels differ (B={B}, G={G}, M={M})"
            assert np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32)), f"deltas differ (B={B}, G={G}, M={M})"
            assert not np.isnan(got_l).any() and set(np.unique(got_l)) <= {-1.0, 0.0, 1.0}
            if ints[0] is not None:
                # with the sampling integers injected the entry point is a parity case of its own: the CPU oracle
                # (utils/train_utils.py:149-202 restated), bit for bit
                ora_d, ora_l = oracle.calculate_rpn_actual_outputs(anchors.cpu().numpy(), gt.cpu().numpy(), labels.cpu().numpy(), hp,
                                                                   (rp.cpu().numpy(), rn.cpu().numpy()))
                assert np.array_equal(got_l.reshape(ora_l.shape).view(np.uint32), ora_l.view(np.uint32)), "labels differ from the oracle"
                assert np.array_equal(got_d.view(np.uint32), np.ascontiguousarray(ora_d).view(np.uint32)), "deltas differ from the oracle"


def test_rpn_targets_static_rejects_sizes_beyond_the_scratch():
    import tf_faster_rcnn_amd as pkg

    L = pkg._lib.lib()
    hp, anchors, gt, labels, rp, rn = _case(pkg, 5, 2, 8, 4)
    B, N, M = 4096, 9216, 100  # 4096 images need ~300 MB of scratch: beyond the 8 MiB static buffer
    assert L.frcnn_rpn_targets_workspace_bytes(B, N, M) > (8 << 20)
    rc = L.frcnn_rpn_targets_static(anchors.data_ptr(), gt.data_ptr(), labels.data_ptr(), None, None, 0, B, N, M, 128, 128,
                                    pkg._lib.float4(hp["variances"]), anchors.data_ptr(), anchors.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)  # rejected before any pointer is used
    assert rc == pkg._lib.FRCNN_EUNSUPPORTED
    assert b"static buffer holds" in L.frcnn_last_error()
    with pytest.raises(pkg._lib.UnsupportedSizeError):
        pkg._lib.check(rc)
    assert isinstance(ctypes.c_int(rc).value, int)

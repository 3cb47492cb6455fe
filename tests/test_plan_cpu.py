"""The planner on the CPU (mars_hip_describe_plan: parse + plan on the host only, no device): which layers fuse, which tensors of an
NCHW-tagged graph are kept pixels x channels, which K loops are cut -- on the reference's own shipped files and on the synthetic twins.
These are decisions about LAUNCHES, not results: every one of these plans is checked bit for bit against the oracle by the GPU tests."""
import collections
import os
import sys

import pytest

from test_oracle import model_bytes

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "thingino-accel_amd"))


def kinds(lines):
    return collections.Counter(l.split()[4] for l in lines if l.startswith("op "))


def count(lines, what):
    return sum(what in l for l in lines)


def test_shipped_yolov5n_int8_plan(marsrt, monkeypatch):
    """BASELINE config 3's literal file (NCHW-tagged, 230 layers, 60 convolutions): with nhwc_internal only the graph input is relaid, nothing
    is stored planar, every CONCAT / stride-1 MAXPOOL / UPSAMPLE layer runs on the internal layout, all 13 concats -- 9 read by one 1 x 1
    convolution, 4 by a C3's cv1 + cv2 pair -- keep their first rows only (their readers are split in two, pairs stay pairs), every residual
    Add is folded"""
    monkeypatch.delenv("MARS_HIP_NO_NHWC_INTERNAL", raising=False)
    L = marsrt.describe_plan(model_bytes("yolov5n_int8"))
    k = kinds(L)
    assert k["conv_i8"] == 60 + 9 + 8 and k["concat_q"] == 13 and k["maxpool_q"] == 3 and k["upsample_q"] == 2
    assert k["concat_slice"] == 0 and k["maxpool"] == 0 and k["upsample"] == 0 and k["binary_i8"] == 0 and k["lut_i8"] == 0 and k["fail"] == 0
    assert count(L, " relayout") == 1 and " relayout" in L[0]  # the stem: the graph input is [3][640][640] bytes
    assert count(L, " planar_store") == 0
    assert count(L, " add=") == 7 and count(L, " pair_next") == 12 and count(L, " rows_only=") == 13  # (4 C3 pairs in the backbone + 2 x 4 split head pairs)
    assert sum(l.startswith("tensor ") and " partial 1" in l for l in L) == 13
    assert sum(l.startswith("tensor ") and " pitch 256" in l for l in L) == 3  # the three 255-channel Detect convolutions' results
    # the same file with the pass switched off: a relayout in front of every convolution, a planar store behind it, one copy per concat input
    monkeypatch.setenv("MARS_HIP_NO_NHWC_INTERNAL", "1")
    L0 = marsrt.describe_plan(model_bytes("yolov5n_int8"))
    k0 = kinds(L0)
    assert k0["conv_i8"] == 60 and count(L0, " relayout") == 60 and count(L0, " planar_store") == 60
    assert k0["concat_q"] == 0 and k0["concat_slice"] >= 26 and k0["binary_i8"] == 7 and not any(l.startswith("tensor ") and "nhwc_c" in l and " nhwc_c 0" not in l for l in L0)


@pytest.mark.parametrize("name", ["yolov5n_int8", "yolov5nu", "tiny_160_int8", "tiny_160_f32", "test_simple", "test_model"])
def test_descriptor_only_ranks_plan_alike(marsrt, name):
    """the multi-GPU invariant: a rank that loads descriptors only (MARS_HIP_LOAD_DEFER_WEIGHTS: no weight blob) plans exactly
    the launches rank 0 plans -- every planner decision depends on shapes, never on weight values -- so the broadcast arena fits"""
    d = model_bytes(name)
    assert marsrt.describe_plan(d, flags=1) == marsrt.describe_plan(d, flags=0)


def test_float_twin_zero_tail_limits(marsrt, monkeypatch):
    """zero_tail_f32: in the float twin the 17 convolutions (1 x 1) that read a CONCAT output stop their K loop at in_c / 4 + 1 channels --
    under the split-bf16 modes only, and not with MARS_HIP_NO_ZERO_TAIL.  virtual_concat_f32 on top of it: none of the 13 concats is
    materialised -- every such convolution reads the concat's LAST input through a view (W (N - 1) bytes early, in_c / 4 planes) and a
    head launch recomputes the first pixels with in_c / 4 + 1 planes"""
    monkeypatch.delenv("MARS_HIP_NO_ZERO_TAIL", raising=False)
    monkeypatch.delenv("MARS_HIP_NO_VCONCAT_F32", raising=False)
    saved = marsrt.get_tuning("f32_mfma")
    try:
        marsrt.set_tuning("f32_mfma", 3)
        d = marsrt.synth_model(width_x16=8, input_hw=640, seed=1, float32=True)
        L = marsrt.describe_plan(d)
        k = kinds(L)
        main = [l for l in L if " view=-" in l]
        heads = [l for l in L if " conv_f32_vhead " in l]
        assert len(main) == 17 and len(heads) == 17 and k["concat_slice"] == 0 and k["conv_f32_vhead"] == 17
        for l in main:
            in_c = int(l.split("->")[0].rsplit(" c", 1)[1])
            assert " k1x1 " in l and int(l.split("k_limit=")[1].split()[0]) == in_c // 4, l
        for l in heads:
            in_c = int(l.split("->")[0].rsplit(" c", 1)[1])
            assert int(l.split("k_limit=")[1].split()[0]) == in_c // 4 + 1 and " vcat=" in l, l
        assert sum(l.startswith("tensor ") and " partial 1" in l for l in L) == 13
        assert count(L, " in_rec=") >= 9 and count(L, " out_rec") >= 9 and count(L, " pair_next") >= 4  # (record pairs, C3 pairs: round 5)
        # the concats materialised (MARS_HIP_NO_VCONCAT_F32): the zero-tail limits alone
        monkeypatch.setenv("MARS_HIP_NO_VCONCAT_F32", "1")
        L = marsrt.describe_plan(d)
        lim = [l for l in L if " k_limit=" in l]
        assert len(lim) == 17 and count(L, " view=-") == 0 and kinds(L)["concat_slice"] >= 26
        for l in lim:
            in_c = int(l.split("->")[0].rsplit(" c", 1)[1])
            assert " k1x1 " in l and int(l.split("k_limit=")[1].split()[0]) == in_c // 4 + 1, l
        monkeypatch.setenv("MARS_HIP_NO_ZERO_TAIL", "1")
        L = marsrt.describe_plan(d)
        assert count(L, " k_limit=") == 0 and count(L, " view=-") == 0  # (no limit: no view either)
        monkeypatch.delenv("MARS_HIP_NO_ZERO_TAIL")
        monkeypatch.delenv("MARS_HIP_NO_VCONCAT_F32")
        marsrt.set_tuning("f32_mfma", 1)
        L1 = marsrt.describe_plan(d)
        assert count(L1, " k_limit=") == 0 and count(L1, " in_rec=") == 0 and count(L1, " view=-") == 0  # (no bf16 weight images, no record tensors, full K loops)
    finally:
        marsrt.set_tuning("f32_mfma", saved)


def test_int8_twin_plan(marsrt):
    """the headline workload's plan: 60 convolutions (56 launches: four C3 cv1 + cv2 pairs), every concat virtual, every SiLU and residual Add
    folded, the SPPF pools one launch -- and nothing of the NCHW machinery (the twin is NHWC-tagged)"""
    L = marsrt.describe_plan(marsrt.synth_model(width_x16=8, input_hw=640, seed=1))
    k = kinds(L)
    assert k["conv_i8"] == 60 and count(L, " pair_next") == 4 and count(L, " add=") == 7
    assert k["concat_slice"] == 0 and k["binary_i8"] == 0 and k["lut_i8"] == 0 and k["upsample"] == 0 and k["maxpool"] == 1 and count(L, " chain=3") == 1
    assert count(L, " seg=") == 17 and count(L, " lut") >= 57  # (13 concats; the four read by a C3's cv1 AND cv2 appear in both readers)
    assert count(L, " relayout") == 0 and count(L, " planar_store") == 0 and k["concat_q"] == 0
    assert sum(l.startswith("tensor ") and " pix_stride 256" in l for l in L) == 3  # the 255-channel heads at a 256-byte pitch


def test_lone_activation_layers_fold_into_the_convolution(marsrt, monkeypatch):
    """fuse_lut: tiny_160_int8.mars (BASELINE config 2) is conv -> RELU -> conv -> RELU -> conv; an int8 activation layer is a 256-entry map of the
    convolution's result, which the convolution's epilogue applies itself: 3 launches instead of 5 (MARS_HIP_NO_FUSE_LUT keeps the layers)"""
    monkeypatch.delenv("MARS_HIP_NO_FUSE_LUT", raising=False)
    L = marsrt.describe_plan(model_bytes("tiny_160_int8"))
    k = kinds(L)
    assert k["conv_i8"] == 3 and k["lut_i8"] == 0 and count(L, " lut") == 2  # (the last convolution has no activation behind it)
    monkeypatch.setenv("MARS_HIP_NO_FUSE_LUT", "1")
    k0 = kinds(marsrt.describe_plan(model_bytes("tiny_160_int8")))
    assert k0["conv_i8"] == 3 and k0["lut_i8"] == 2

"""CPU: host-side logic added in round 4 that needs no device - telemetry degrades to "unavailable", argument validation of
the frame API, the full-size fixture cases, the bench's executed-flop accounting."""
import importlib
import os
import sys

import pytest
import torch

import cases
from neo360_amd import render, telemetry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_telemetry_without_a_device_reports_unavailable():
    """No GPU / no driver: the sampler must not invent numbers (bench.py then emits nulls)."""
    if torch.cuda.is_available():
        pytest.skip("a device is present: the sampler returns real numbers")
    with telemetry.Sampler(0) as s:
        pass
    out = s.summary()
    assert "unavailable" in out.get("telemetry", "") and "sclk_mhz_mean" not in out
    assert telemetry.read_once(0) == (None, None, None)


def test_render_rays_test_validates_on_range():
    with pytest.raises(ValueError, match="on_range"):
        render.render_rays_test(object(), {}, on_range="fallback")
    with pytest.raises(TypeError, match="unsupported renderer"):
        render.render_rays_test(object(), {}, on_range="raise")


def test_full_size_cases_are_distinct_and_inside_the_sphere():
    assert set(cases.FULL_B) == {"b1", "b2", "b3", "b4", "b5", "b6"} and cases.FULL_B["b4"]["nv"] == 5
    assert cases.full_gain("b5") == 8.0 and cases.full_gain("b6") == 1.0 and cases.full_gain("") == 1.0
    seen = []
    for tag, kw in cases.FULL_B.items():
        kw = {k: v for k, v in kw.items() if k not in ("gain", "seed", "std")}       # weights / scene of the case, not its rays
        nv = kw.pop("nv")
        b = cases.full_batch(64, nv=nv, **kw)
        assert b["rays_o"].shape == (64, 3) and b["src_poses"].shape == (nv, 4, 4)
        assert float(b["rays_o"].norm(dim=-1).max()) < 1.0             # every ray starts inside the unit sphere: it has an exit point
        assert float((b["rays_d"].norm(dim=-1) - 1.0).abs().max()) < 1e-6
        seen.append((tuple(b["rays_o"][0].tolist()), tuple(b["rays_d"][0].tolist())))
    assert len(set(seen)) == 6                                          # six different chunks
    base = cases.full_batch(64)
    assert not torch.equal(base["rays_d"], cases.full_batch(64, **{k: v for k, v in cases.FULL_B["b1"].items() if k != "nv"})["rays_d"])


def test_bench_executed_flop_accounting():
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    f = bench.executed_flop_per_point_tp_hp
    tail = 160 * 64 + 64 * 64                      # view layer 0 with the bottleneck folded in + view layer 1
    assert f(3, False) == (3 * (12 * 16 * 256 + 3 * 128 * 128) + tail) * 6
    assert f(3, True) == (3 * (14 * 16 * 256 + 3 * 128 * 128) + tail) * 6
    # tri-planes pre-projected: the 8 world k-steps are gone
    assert f(3, True, planes_projected=True) == (3 * (6 * 16 * 256 + 3 * 128 * 128) + tail) * 6
    assert f(3, False, planes_projected=True) < f(3, False) < 2 * 778752 * 3       # fewer than the reference formulation's MACs x 3 products


def test_chain_gradient_buffers_are_views_of_one_zero_fill():
    """training._zeros_like_shapes: the 18 weight / bias gradient tensors of a chain call as views of ONE zeroed buffer (one fill kernel
    instead of 18), every view starting at a multiple of 4 floats (the library's 16-byte stores) and none overlapping."""
    import torch
    from neo360_amd import training
    wshapes = [(128, 703), (128, 128), (128, 128), (128, 831), (64, 155), (64, 64), (128, 128), (1, 128), (3, 64)]
    bshapes = [(s[0],) for s in wshapes]
    gw, gb = training._zeros_like_shapes(wshapes, bshapes, "cpu")
    assert [tuple(t.shape) for t in gw] == wshapes and [tuple(t.shape) for t in gb] == bshapes
    base = gw[0].untyped_storage().data_ptr()
    spans = []
    for t in gw + gb:
        assert t.untyped_storage().data_ptr() == base and t.is_contiguous() and float(t.abs().sum()) == 0.0
        off = (t.data_ptr() - base) // 4
        assert off % 4 == 0
        spans.append((off, off + t.numel()))
    spans.sort()
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))

"""brushstroke_engine_amd.forger_losses against the reference's loss items and random stitcher
(tests/golden/forger_losses.npz, made by tests/golden/make_golden_forger_losses.py from forger/train/{losses,stitching}.py)."""
import random

import numpy as np
import pytest
import torch

from conftest import load_golden
from brushstroke_engine_amd import forger_losses as fl


@pytest.fixture(scope="module")
def g():
    return load_golden("forger_losses.npz")


def test_loss_items_match_reference(g):
    data = {k[3:]: torch.from_numpy(g[k]) for k in g if k.startswith("in_") and k != "in_truth"}
    truth = torch.from_numpy(g["in_truth"])
    for i, cfg in enumerate(g["configs"].tolist()):
        for partial in (0, 1):
            if f"cfg{i}_p{partial}_total" not in g:
                continue
            L = fl.ForgerLosses.create_from_string(cfg)
            L.set_partial_loss_with_triband_input(bool(partial))
            total, vals = L.compute(data, truth)
            assert sorted(vals.keys()) == g[f"cfg{i}_p{partial}_names"].tolist(), cfg
            np.testing.assert_allclose([float(vals[k]) for k in sorted(vals)], g[f"cfg{i}_p{partial}_vals"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(float(total), float(g[f"cfg{i}_p{partial}_total"]), rtol=1e-6)


def test_loss_string_grammar_and_errors():
    L = fl.ForgerLosses.create_from_string("'1.0*iou_inv(uvs) + 1.0*iou(u)'")           # as written in train_flags.txt:10
    assert [l.full_name() for l in L.losses] == ["iou_inv_uvs", "iou_u"] and L.weights == [1.0, 1.0]
    assert fl.ForgerLosses.create_from_string("").is_empty()
    assert fl.ForgerLosses.create_from_string("l1(fake_orig)").require_original_fake_image()
    for bad, msg in (("iou(uvs)+iou(uvs)", "more than once"), ("foo(uvs)", "not found"), ("iou(bogus)", "not in valid values"),
                     ("2*3*iou(u)", "Mis-configured"), ("iou", "Mis-configured"), ("lpips(patch)", "LPIPS")):
        with pytest.raises(RuntimeError, match=msg):
            fl.ForgerLosses.create_from_string(bad)
    with pytest.raises(RuntimeError, match="Unsupported component"):
        fl.ForgerLosses.create_from_string("iou(color_0)").compute({"uvs": torch.zeros(1, 3, 2, 2)}, torch.zeros(1, 1, 2, 2))
    with pytest.raises(RuntimeError, match="expected in"):
        fl.ForgerLosses.create_from_string("gan(fake)").compute({}, None)


def test_losses_are_differentiable():
    uvs = torch.softmax(torch.randn(2, 3, 8, 8, requires_grad=True), dim=1)
    total, _ = fl.ForgerLosses.create_from_string("iou_inv(uvs)+iou(u)+dice(uvs)").compute({"uvs": uvs}, (torch.rand(2, 1, 8, 8) > 0.5).float())
    assert torch.autograd.grad(total, uvs)[0].abs().sum() > 0


def test_stitcher_matches_reference(g):
    r = g["st_fake1"].shape[-1]

    class FakeG:
        img_resolution = r

        def __call__(self, z, c, geom_feature, positions=None, style_mixing_prob=0):
            base = torch.linspace(0, 1, r * r).reshape(1, 1, r, r) * z[:, :1, None, None]
            return base + geom_feature[0].mean(dim=(1, 2, 3), keepdim=True) + positions.float().sum(dim=1).reshape(-1, 1, 1, 1) * 0.01 \
                + torch.arange(3).reshape(1, 3, 1, 1)
    st = fl.RandomStitcher(crop_margin=2, min_overlap=6)
    res = st.generate_with_stitching(FakeG(), torch.from_numpy(g["st_z"]), None, [torch.from_numpy(g["st_g1"])], [torch.from_numpy(g["st_g2"])],
                                     tuple(g["st_crop1"].tolist()), tuple(g["st_crop2"].tolist()), positions1=torch.from_numpy(g["st_pos1"]))
    for k in ("fake1", "fake2", "fake1_composite", "fake2_composite", "positions1", "positions2", "patch1", "patch2"):
        np.testing.assert_array_equal(res[k].numpy(), g[f"st_{k}"], err_msg=k)
    assert not np.array_equal(g["st_fake1_composite"], g["st_fake1"])                # (the composite replaced the overlap)
    random.seed(5)
    crops = [st.gen_overlapping_square_crop(100, (30, 40, r, r)) for _ in range(8)]
    np.testing.assert_array_equal(np.array(crops), g["st_gen_crops"])


def test_lazy_stats_is_a_mapping_of_floats():
    """LazyStats keeps device tensors until read, and every way of copying it out -- dict(), {**}, update(), json -- yields floats."""
    import json
    import torch
    from brushstroke_engine_amd.training import LazyStats
    st = LazyStats({"a": torch.tensor(1.5), "b": 2.0})
    st["c"] = torch.tensor(3)
    st.update({"d": torch.tensor(0.25)}, e=torch.tensor(4.0))
    plain = {}
    plain.update(st)
    for d in (dict(st), {**st}, plain, st.resolve(), dict(st.items())):
        assert d == {"a": 1.5, "b": 2.0, "c": 3.0, "d": 0.25, "e": 4.0}
        assert all(isinstance(v, float) for v in d.values())
        assert json.loads(json.dumps(d)) == d
    assert st.get("zz", 7) == 7 and len(st) == 5 and list(st) == ["a", "b", "c", "d", "e"]
    merged = LazyStats(st)
    merged.update(LazyStats({"a": torch.tensor(9.0)}))
    assert merged["a"] == 9.0 and st["a"] == 1.5

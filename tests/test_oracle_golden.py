"""CPU: the oracle (oracle/dvmvs_oracle.py) against the golden vectors produced by running the reference
(tests/golden/make_goldens.py) and against the known answers of SURVEY.md Appendix B.3."""
import json
import os

import numpy as np
import pytest
import torch

import dvmvs_oracle as orc
import synthetic as syn


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def check_pins(t, z, prefix, atol, rtol_sum=1e-5):
    assert list(t.shape) == list(z[f"{prefix}_shape"])
    idx = syn.sample_indices(t.numel())
    got = t.reshape(-1)[idx].float().cpu().numpy()
    np.testing.assert_allclose(got, z[f"{prefix}_samples"], atol=atol, rtol=0)
    scale = float(z[f"{prefix}_abs_sum"])
    assert abs(t.double().sum().item() - float(z[f"{prefix}_sum"])) <= rtol_sum * scale
    assert abs(t.double().abs().sum().item() - scale) <= rtol_sum * scale


def test_pinning_report_is_tight(golden_dir):
    """The evidence written by make_goldens.py: |oracle - reference| on identical inputs."""
    rep = json.load(open(os.path.join(golden_dir, "PINNING_REPORT.json")))
    for key, entry in rep.items():
        if key.startswith("cost_volume") and "sad" not in key:
            assert entry["oracle_vs_reference_max_abs"] < 1e-6, key
        if key.startswith("cost_volume") and "sad" in key:
            assert entry["oracle_vs_reference_max_abs"] < 1e-5, key
        if key.startswith("reproject"):
            assert entry["mismatching_pixels"] == 0, key
        if key.startswith("keyframe_index"):
            assert entry["matching"] == entry["shipped"] == entry["lines"], key
    assert rep["hidden_warp"]["masked_max_abs"] < 1e-6 and rep["lstm_gates"]["h_max_abs"] < 5e-6


def test_cost_volume_small_all_cases(golden_dir):
    z = load(golden_dir, "cost_volume_small")
    K = torch.from_numpy(z["K"])
    pose_sets = json.loads(str(z["pose_sets"]))
    feats = [syn.analytic_features(s, 8, 32, 40) for s in range(4)]
    for tag, (r, ms) in pose_sets.items():
        for dot in (True, False):
            out = orc.cost_volume_fusion(feats[0], [feats[1 + i] for i in range(len(ms))], syn.pose(r), [syn.pose(m) for m in ms],
                                         K, 0.25, 20.0, 16, dot)
            exp = torch.from_numpy(z[f"{tag}_{'dot' if dot else 'sad'}"])
            assert (out - exp).abs().max().item() < (1e-6 if dot else 1e-5), (tag, dot)


def test_cost_volume_known_answers_full_size(golden_dir):
    """KAT-CV / KAT-SAD of SURVEY.md B.3 (numbers measured from the reference) and the sampled-entry pins."""
    z = load(golden_dir, "cost_volume_full_pins")
    halfK = syn.scaled_K(syn.full_K(), 2.0)
    f = [syn.analytic_features(s) for s in range(3)]
    cv = orc.cost_volume_fusion(f[0], [f[1], f[2]], syn.pose(9), [syn.pose(6), syn.pose(0)], halfK, 0.25, 20.0, 64, True)
    assert abs(cv.double().sum().item() - 78687.558811) < 0.05
    assert abs(cv.double().abs().sum().item() - 82969.911147) < 0.05
    assert abs(cv[0, 0, 64, 80].item() - 0.31054920) < 1e-6
    assert abs(cv[0, 31, 10, 20].item() - 0.36512548) < 1e-6
    assert abs(cv[0, 63, 127, 159].item() - 0.02026378) < 1e-6
    check_pins(cv, z, "kat_cv", atol=1e-6)
    sad = orc.cost_volume_fusion(f[0][:, :3], [f[1][:, :3]], syn.pose(9), [syn.pose(6)], halfK, 0.5, 50.0, 64, False)
    assert abs(sad.double().sum().item() - 1749443.403785) < 1.0
    assert abs(sad[0, 5, 64, 80].item() - 2.28013682) < 1e-5
    check_pins(sad, z, "kat_sad", atol=1e-5)
    back = orc.cost_volume_fusion(f[0], [f[1]], syn.pose(141), [syn.pose(135)], halfK, 0.25, 20.0, 64, True)
    check_pins(back, z, "behind", atol=1e-6)


def test_cost_volume_float64_agrees():
    """Same algebra in float64: separates rounding from algorithm (SURVEY: 2.8e-5 max / 1.3e-6 mean observed)."""
    halfK = syn.scaled_K(syn.full_K(), 2.0)
    f = [syn.analytic_features(s, 32, 64, 80) for s in range(2)]
    K = syn.scaled_K(halfK, 2.0)
    a = orc.cost_volume(f[0], f[1], syn.pose(9), syn.pose(6), K, 0.25, 20.0, 32, True)
    b = orc.cost_volume(f[0].double(), f[1].double(), syn.pose(9).double(), syn.pose(6).double(), K.double(), 0.25, 20.0, 32, True)
    d = (a.double() - b).abs()
    assert d.max().item() < 2e-4 and d.mean().item() < 1e-5


def test_reprojection_goldens(golden_dir):
    z = load(golden_dir, "reproject")
    fullK = syn.full_K()
    halfK = syn.scaled_K(fullK, 2.0)
    prev = syn.analytic_depth()
    out = orc.reproject_depth(syn.pose(10), syn.pose(9), prev, fullK, halfK, 320, 256)
    assert abs(out.double().sum().item() - 30690.716363) < 0.05 and int((out != 0).sum()) == 20307   # KAT-REPROJ
    assert abs(out[0, 0, 64, 80].item() - 1.03667092) < 1e-6
    np.testing.assert_array_equal(out.numpy(), z["kat"])
    low = orc.nearest_downsample(out, 16)
    np.testing.assert_array_equal(low.numpy(), z["kat_low"])
    np.testing.assert_allclose(low[0, 0, 0].numpy(), [1.7173672, 2.0606678, 2.2028925, 2.0791428, 1.7567809, 1.4292806, 1.2380984,
                                                      1.2830925, 1.6267483, 2.0264084], atol=1e-6)
    prev2 = prev.clone()
    prev2[:, :, 40:90, 100:180] = 0.0
    prev2[:, :, 150:, :] = 6.0
    out2 = orc.reproject_depth(syn.pose(16), syn.pose(9), prev2, fullK, halfK, 320, 256)
    np.testing.assert_array_equal(out2.numpy(), z["hard"])


def test_hidden_warp_goldens(golden_dir):
    z = load(golden_dir, "hidden_warp")
    lK = syn.scaled_K(syn.full_K(), 32.0)
    _, _, h0, _ = syn.analytic_lstm_inputs()
    t = lambda k: torch.from_numpy(z[k])
    assert (orc.warp_hidden_state(h0, t("depth"), t("T"), lK) - t("warped")).abs().max().item() < 1e-6
    assert (orc.warp_hidden_state(h0, t("depth_masked"), t("T"), lK, zero_invalid=True) - t("warped_masked")).abs().max().item() < 1e-6
    assert (orc.warp_hidden_state(h0, t("depth"), t("T_far"), lK) - t("warped_far")).abs().max().item() < 1e-6


def test_lstm_goldens(golden_dir):
    z = load(golden_dir, "lstm_gates")
    weight, x, h0, c0 = syn.analytic_lstm_inputs()
    o = np.arange(2048, dtype=np.float64).reshape(-1, 1, 1)
    yy = np.arange(8, dtype=np.float64).reshape(1, -1, 1)
    xx = np.arange(10, dtype=np.float64).reshape(1, 1, -1)
    cc = torch.from_numpy((2.0 * np.sin(0.013 * o + 0.7 * yy + 0.3 * xx) + 0.5 * np.cos(0.05 * o * xx)).astype(np.float32)).unsqueeze(0)
    h, c = orc.lstm_gates(cc, c0)
    assert (h - torch.from_numpy(z["h_next"])).abs().max().item() < 5e-6
    assert (c - torch.from_numpy(z["c_next"])).abs().max().item() < 5e-6
    # KAT-LSTM: the whole cell (warp + mask + conv + gates)
    lK = syn.scaled_K(syn.full_K(), 32.0)
    de16 = torch.from_numpy(load(golden_dir, "reproject")["kat_low"])
    hn, cn = orc.convlstm_cell(weight, x, h0, c0, syn.pose(9), syn.pose(10), de16, lK)
    assert abs(hn.double().abs().sum().item() - 14085.416424) < 0.05
    assert abs(cn.double().abs().sum().item() - 33620.061688) < 0.05
    assert (hn - torch.from_numpy(z["kat_h"])).abs().max().item() < 2e-5
    assert (cn - torch.from_numpy(z["kat_c"])).abs().max().item() < 2e-5


def test_planewise_cost_volume_equals_gather_formulation():
    """The grid_sample-per-plane variant timed as cpu_baseline is the same function as the oracle proper."""
    K = syn.scaled_K(syn.full_K(), 4.0)
    f = [syn.smooth_noise((1, 16, 64, 80), seed=s) for s in (1, 2, 3)]
    for dot in (True, False):
        a = orc.cost_volume_fusion(f[0], f[1:], syn.pose(9), [syn.pose(6), syn.pose(141)], K, 0.25, 20.0, 32, dot)
        b = orc.cost_volume_fusion(f[0], f[1:], syn.pose(9), [syn.pose(6), syn.pose(141)], K, 0.25, 20.0, 32, dot, planewise=True)
        assert (a - b).abs().max().item() < (1e-6 if dot else 1e-5)

"""CPU: the module surface the reference's scripts rely on -- names, signatures, state-dict keys, shapes, Config, and that
the hot-path ops refuse to run anywhere but on the HIP device (no silent fallback)."""
import inspect
import json
import os

import pytest
import torch

import synthetic as syn


def test_config_attribute_surface():
    from dvmvs.config import Config
    expected = {
        "train_image_width": 256, "train_image_height": 256, "train_min_depth": 0.25, "train_max_depth": 20.0,
        "train_n_depth_levels": 64, "train_minimum_pose_distance": 0.125, "train_maximum_pose_distance": 0.325,
        "train_crawl_step": 3, "train_subsequence_length": None, "train_predict_two_way": None,
        "train_freeze_batch_normalization": False, "train_data_pipeline_workers": 8, "train_epochs": 100000,
        "train_print_frequency": 5000, "train_validate": True, "test_image_width": 320, "test_image_height": 256,
        "test_distortion_crop": 0, "test_perform_crop": False, "test_visualize": True, "test_n_measurement_frames": 2,
        "test_keyframe_buffer_size": 30, "test_keyframe_pose_distance": 0.1, "test_optimal_t_measure": 0.15,
        "test_optimal_R_measure": 0.0, "test_dataset_name": "hololens-dataset"}
    for name, value in expected.items():
        assert getattr(Config, name) == value, name
    for name in ("train_seed", "dataset", "train_run_directory", "test_online_scene_path", "test_offline_data_path", "test_result_folder"):
        assert hasattr(Config, name)


def test_state_dict_keys_match_the_reference(golden_dir):
    """Key names and shapes captured from the reference's modules (tests/golden/state_dict_keys.json)."""
    from dvmvs.fusionnet import model as fm
    from dvmvs.pairnet import model as pm
    want = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    have = {"feature_extractor": fm.FeatureExtractor(), "feature_shrinker": fm.FeatureShrinker(), "encoder": fm.CostVolumeEncoder(),
            "lstm_fusion": fm.LSTMFusion(), "decoder": fm.CostVolumeDecoder(), "pairnet_encoder": pm.CostVolumeEncoder(),
            "pairnet_decoder": pm.CostVolumeDecoder()}
    for name, mod in have.items():
        got = {k: list(v.shape) for k, v in mod.state_dict().items()}
        assert list(got.keys()) == list(want[name].keys()), name      # same names, same order
        assert got == want[name], name
    counts = {n: sum(p.numel() for p in m.parameters()) for n, m in have.items()}
    assert counts["feature_extractor"] == 2690152 and counts["feature_shrinker"] == 62272 and counts["encoder"] == 8984128
    assert counts["lstm_fusion"] == 18874368 and counts["decoder"] == 4063269
    assert fm.fpn_output_channels == 32 and fm.hyper_channels == 32


def test_published_fpn_checkpoint_loads():
    from dvmvs.fusionnet.model import FeatureShrinker
    missing, unexpected = FeatureShrinker().load_state_dict(syn.load_fpn_checkpoint(), strict=True)
    assert not missing and not unexpected


def test_forward_shapes_on_cpu():
    """Dense modules are plain torch and run anywhere (Appendix C shapes)."""
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker
    fe, fs, enc, dec = FeatureExtractor().eval(), FeatureShrinker().eval(), CostVolumeEncoder().eval(), CostVolumeDecoder().eval()
    x = torch.randn(1, 3, 256, 320)
    with torch.no_grad():
        taps = fe(x)
        assert [tuple(t.shape[1:]) for t in taps] == [(16, 128, 160), (24, 64, 80), (40, 32, 40), (96, 16, 20), (320, 8, 10)]
        feats = fs(*taps)
        assert [tuple(t.shape[1:]) for t in feats] == [(32, 128, 160), (32, 64, 80), (32, 32, 40), (32, 16, 20)]
        skips = enc(*feats, torch.randn(1, 64, 128, 160))
        assert [tuple(t.shape[1:]) for t in skips] == [(32, 128, 160), (64, 64, 80), (128, 32, 40), (256, 16, 20), (512, 8, 10)]
        depths = dec(x, *skips)
        assert [tuple(t.shape[1:]) for t in depths] == [(256, 320), (128, 160), (64, 80), (32, 40), (16, 20)]
        assert all(float(d.min()) >= 0.25 - 1e-4 and float(d.max()) <= 20.0 + 1e-3 for d in depths)


def test_function_signatures_match_the_reference():
    from dvmvs import utils
    from dvmvs.convlstm import MVSLayernormConvLSTMCell
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(utils.cost_volume_fusion) == ["image1", "image2s", "pose1", "pose2s", "K", "warp_grid", "min_depth", "max_depth",
                                             "n_depth_levels", "device", "dot_product"]
    assert sig(utils.calculate_cost_volume_by_warping) == ["image1", "image2", "pose1", "pose2", "K", "warp_grid", "min_depth",
                                                           "max_depth", "n_depth_levels", "device", "dot_product"]
    assert sig(utils.get_non_differentiable_rectangle_depth_estimation) == [
        "reference_pose_torch", "measurement_pose_torch", "previous_depth_torch", "full_K_torch", "half_K_torch", "original_width",
        "original_height"]
    assert sig(utils.warp_frame_depth) == ["image_src", "depth_dst", "src_trans_dst", "camera_matrix", "normalize_points", "sampling_mode"]
    assert sig(utils.get_warp_grid_for_cost_volume_calculation) == ["width", "height", "device"]
    assert sig(MVSLayernormConvLSTMCell.forward)[1:7] == ["input_tensor", "cur_state", "previous_pose", "current_pose",
                                                          "estimated_current_depth", "camera_matrix"]
    # anything beyond the reference's six arguments must be optional (the engine passes a precomputed relative pose)
    extra = list(inspect.signature(MVSLayernormConvLSTMCell.forward).parameters.values())[7:]
    assert all(p.default is not inspect.Parameter.empty for p in extra)
    assert sig(MVSLayernormConvLSTMCell.__init__)[1:] == ["input_dim", "hidden_dim", "kernel_size", "activation_function"]
    grid = utils.get_warp_grid_for_cost_volume_calculation(5, 3, "cpu")
    assert tuple(grid.shape) == (3, 15) and grid[:, 7].tolist() == [2.0, 1.0, 1.0]


def test_hot_path_has_no_cpu_fallback():
    from dvmvs import utils
    from dvmvs.fusionnet.model import LSTMFusion
    f = torch.randn(1, 8, 16, 20)
    eye4, eye3 = torch.eye(4)[None], torch.eye(3)[None]
    with pytest.raises(RuntimeError, match="MI355X"):
        utils.cost_volume_fusion(f, [f], eye4, [eye4], eye3, None, 0.25, 20.0, 8, "cpu", True)
    with pytest.raises(RuntimeError, match="MI355X"):
        utils.warp_frame_depth(f, torch.ones(1, 1, 16, 20), eye4, eye3)
    with pytest.raises(RuntimeError, match="MI355X"):
        utils.get_non_differentiable_rectangle_depth_estimation(eye4, eye4, torch.ones(1, 1, 16, 20), eye3, eye3, 20, 16)
    with pytest.raises(RuntimeError, match="MI355X"):
        LSTMFusion()(torch.randn(1, 512, 8, 10), None, None, eye4, torch.zeros(1, 1, 8, 10), eye3)


def test_op_shape_inference_with_meta_tensors():
    """register_fake implementations (what torch.compile / shape propagation sees)."""
    from dvmvs.hip import ops
    m = lambda *s: torch.empty(*s, device="meta")
    assert tuple(ops.cost_volume(m(2, 32, 128, 160), [m(2, 32, 128, 160)] * 2, m(2, 2, 9), m(2, 2, 3), 0.25, 20.0, 64,
                                 True, 0).shape) == (2, 64, 128, 160)
    Hm, kt = ops.sweep_matrices(m(2, 4, 4), [m(2, 4, 4)] * 3, m(2, 3, 3))
    assert tuple(Hm.shape) == (2, 3, 9) and tuple(kt.shape) == (2, 3, 3)
    assert tuple(ops.hidden_warp(m(1, 512, 8, 10), m(1, 1, 8, 10), m(1, 4, 4), m(1, 3, 3), True).shape) == (1, 512, 8, 10)
    h, c = ops.lstm_gates(m(4, 2048, 8, 8), m(4, 512, 8, 8))
    assert tuple(h.shape) == tuple(c.shape) == (4, 512, 8, 8)
    full, low = ops.depth_reproject_lowres(m(1, 4, 4), m(1, 1, 256, 320), m(1, 3, 3), m(1, 3, 3), 16)
    assert tuple(full.shape) == (1, 1, 128, 160) and tuple(low.shape) == (1, 1, 8, 10)


def test_bn_folding_is_exact_enough():
    from dvmvs.engine import fold_batchnorm
    from dvmvs.fusionnet.model import CostVolumeEncoder
    enc = syn.deterministic_init(CostVolumeEncoder(), seed=2).eval()
    folded = fold_batchnorm(enc)
    assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in folded.modules())
    feats = [torch.randn(1, 32, 128 // s, 160 // s) for s in (1, 2, 4, 8)]
    cv = torch.randn(1, 64, 128, 160)
    with torch.no_grad():
        a, b = enc(*feats, cv), folded(*feats, cv)
    for x, y in zip(a, b):
        assert (x - y).abs().max().item() <= 1e-4 * max(1.0, x.abs().max().item())
    assert list(enc.state_dict().keys())[0] == "aggregator0.0.weight"   # the original module is untouched


def test_error_metrics_match_the_reference(golden_dir):
    import numpy as np
    from dvmvs.errors import compute_errors
    z = np.load(os.path.join(golden_dir, "error_metrics.npz"))
    gt, pred = syn.error_metric_inputs()
    np.testing.assert_allclose(compute_errors(gt, pred), z["all_pixels"], rtol=1e-9)
    np.testing.assert_allclose(compute_errors(gt, pred, 2.0), z["max_depth_2"], rtol=1e-9)
    assert np.isnan(z["nothing_valid"]).all() and all(np.isnan(v) for v in compute_errors(np.zeros((4, 4)), np.ones((4, 4))))


def test_errors_module():
    import numpy as np
    from dvmvs.errors import compute_errors
    gt = np.full((4, 4), 2.0)
    pred = np.full((4, 4), 2.5)
    e = compute_errors(gt, pred)
    assert abs(e[0] - 0.5) < 1e-9 and abs(e[1] - 0.25) < 1e-9 and abs(e[2] - 0.1) < 1e-9 and abs(e[4] - 0.5) < 1e-9 and e[5] == 0.0 and e[6] == 1.0
    assert all(np.isnan(v) for v in compute_errors(np.zeros((2, 2)), pred[:2, :2]))

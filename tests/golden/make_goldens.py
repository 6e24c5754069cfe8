"""Generates the golden fixtures under tests/golden/ by RUNNING THE REFERENCE on CPU in the build container.

    python tests/golden/make_goldens.py            # rewrites tests/golden/*.npz and PINNING_REPORT.json

The reference (``/root/reference/dvmvs``) is imported where it lies (see reference_import.py for the stand-ins of
its absent third-party imports); only plain arrays (inputs / expected outputs) are written.  While it is loaded,
the CPU oracle (``oracle/dvmvs_oracle.py``) is compared against it op by op and the observed differences are written
to PINNING_REPORT.json, which is the evidence that the oracle is pinned to the reference.

The reference has no tests or golden vectors of its own; the known answers listed in SURVEY.md
Appendix B.3 (measured during the survey from the same import) are re-derived here and must reproduce.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import synthetic as syn  # noqa: E402
from reference_import import import_reference  # noqa: E402

torch.set_num_threads(8)
CPU = torch.device("cpu")
REPORT = {}


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


def meandiff(a, b):
    return float((a.double() - b.double()).abs().mean())


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(f"wrote {name}: {os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024:.0f} KiB")


def pins_to_arrays(prefix, pins):
    return {f"{prefix}_shape": pins["shape"], f"{prefix}_sum": pins["sum"], f"{prefix}_abs_sum": pins["abs_sum"],
            f"{prefix}_samples": pins["samples"]}


# ----------------------------------------------------------------------------------------------------------------------
def cost_volume_goldens(ref):
    oracle = ref.oracle
    halfK = syn.scaled_K(syn.full_K(), 2.0)

    # ---- full-size known answers ----
    grid = ref.utils.get_warp_grid_for_cost_volume_calculation(160, 128, CPU)
    feats = [syn.analytic_features(s) for s in range(3)]
    cv = ref.utils.cost_volume_fusion(feats[0], [feats[1], feats[2]], syn.pose(9), [syn.pose(6), syn.pose(0)], halfK, grid,
                                      0.25, 20.0, 64, CPU, True)
    kat = {"sum": cv.double().sum().item(), "abs_sum": cv.double().abs().sum().item(),
           "v_0_64_80": cv[0, 0, 64, 80].item(), "v_31_10_20": cv[0, 31, 10, 20].item(), "v_63_127_159": cv[0, 63, 127, 159].item()}
    print("KAT-CV", kat)
    assert abs(kat["sum"] - 78687.558811) < 0.05 and abs(kat["v_0_64_80"] - 0.31054920) < 1e-6, "KAT-CV does not reproduce"
    ocv = oracle.cost_volume_fusion(feats[0], [feats[1], feats[2]], syn.pose(9), [syn.pose(6), syn.pose(0)], halfK, 0.25, 20.0, 64, True)
    REPORT["cost_volume_full_dot_M2"] = {"oracle_vs_reference_max_abs": maxdiff(cv, ocv), "mean_abs": meandiff(cv, ocv),
                                         "mean_abs_value": float(cv.abs().mean())}
    sad = ref.utils.cost_volume_fusion(feats[0][:, :3], [feats[1][:, :3]], syn.pose(9), [syn.pose(6)], halfK, grid, 0.5, 50.0, 64, CPU, False)
    print("KAT-SAD", sad.double().sum().item(), sad[0, 5, 64, 80].item())
    assert abs(sad.double().sum().item() - 1749443.403785) < 1.0, "KAT-SAD does not reproduce"
    osad = oracle.cost_volume_fusion(feats[0][:, :3], [feats[1][:, :3]], syn.pose(9), [syn.pose(6)], halfK, 0.5, 50.0, 64, False)
    REPORT["cost_volume_full_sad_M1"] = {"oracle_vs_reference_max_abs": maxdiff(sad, osad), "mean_abs": meandiff(sad, osad),
                                         "mean_abs_value": float(sad.abs().mean())}
    # pair with 38 % of the samples behind the camera (pose 141 vs 135)
    back = ref.utils.cost_volume_fusion(feats[0], [feats[1]], syn.pose(141), [syn.pose(135)], halfK, grid, 0.25, 20.0, 64, CPU, True)
    oback = oracle.cost_volume_fusion(feats[0], [feats[1]], syn.pose(141), [syn.pose(135)], halfK, 0.25, 20.0, 64, True)
    REPORT["cost_volume_full_behind_camera"] = {"oracle_vs_reference_max_abs": maxdiff(back, oback), "mean_abs": meandiff(back, oback)}
    # smooth-noise features: a less regular signal than the sinusoids
    nf = [syn.smooth_noise((1, 32, 128, 160), seed=40 + i) for i in range(3)]
    ncv = ref.utils.cost_volume_fusion(nf[0], [nf[1], nf[2]], syn.pose(13), [syn.pose(12), syn.pose(9)], halfK, grid, 0.25, 20.0, 64, CPU, True)
    oncv = oracle.cost_volume_fusion(nf[0], [nf[1], nf[2]], syn.pose(13), [syn.pose(12), syn.pose(9)], halfK, 0.25, 20.0, 64, True)
    REPORT["cost_volume_full_noise_M2"] = {"oracle_vs_reference_max_abs": maxdiff(ncv, oncv), "mean_abs": meandiff(ncv, oncv),
                                           "mean_abs_value": float(ncv.abs().mean())}
    arrays = {}
    arrays.update(pins_to_arrays("kat_cv", syn.tensor_pins(cv)))
    arrays.update(pins_to_arrays("kat_sad", syn.tensor_pins(sad)))
    arrays.update(pins_to_arrays("behind", syn.tensor_pins(back)))
    arrays.update(pins_to_arrays("noise", syn.tensor_pins(ncv)))
    save("cost_volume_full_pins", **arrays)

    # ---- reduced shape, full tensors: C=8, 32x40, D=16; K = half-res intrinsics / 4 ----
    smallK = syn.scaled_K(halfK, 4.0)
    sgrid = ref.utils.get_warp_grid_for_cost_volume_calculation(40, 32, CPU)
    sf = [syn.analytic_features(s, 8, 32, 40) for s in range(4)]
    cases = {}
    pose_sets = {"m1": (12, [9]), "m2": (12, [9, 3]), "m3": (23, [22, 21, 20]), "behind": (141, [135])}
    for tag, (r, ms) in pose_sets.items():
        for dot in (True, False):
            out = ref.utils.cost_volume_fusion(sf[0], [sf[1 + i] for i in range(len(ms))], syn.pose(r), [syn.pose(m) for m in ms],
                                               smallK, sgrid, 0.25, 20.0, 16, CPU, dot)
            o = oracle.cost_volume_fusion(sf[0], [sf[1 + i] for i in range(len(ms))], syn.pose(r), [syn.pose(m) for m in ms],
                                          smallK, 0.25, 20.0, 16, dot)
            key = f"{tag}_{'dot' if dot else 'sad'}"
            cases[key] = out
            REPORT[f"cost_volume_small_{key}"] = {"oracle_vs_reference_max_abs": maxdiff(out, o)}
    single = ref.utils.calculate_cost_volume_by_warping(sf[0], sf[1], syn.pose(12), syn.pose(9), smallK, sgrid, 0.25, 20.0, 16, CPU, True)
    assert torch.equal(single, cases["m1_dot"])
    save("cost_volume_small", K=smallK, **cases, pose_sets=json.dumps(pose_sets))

    # ---- autograd through the reference (dot mode), reduced shape, M=2 ----
    f1 = sf[0].clone().requires_grad_(True)
    f2 = [sf[1].clone().requires_grad_(True), sf[2].clone().requires_grad_(True)]
    out = ref.utils.cost_volume_fusion(f1, f2, syn.pose(12), [syn.pose(9), syn.pose(3)], smallK, sgrid, 0.25, 20.0, 16, CPU, True)
    gout = syn.smooth_noise((1, 16, 32, 40), seed=77, passes=1)
    out.backward(gout)
    save("cost_volume_small_grad", grad_out=gout, grad_image1=f1.grad, grad_image2_0=f2[0].grad, grad_image2_1=f2[1].grad)


def reprojection_goldens(ref):
    oracle = ref.oracle
    fullK = syn.full_K()
    halfK = syn.scaled_K(fullK, 2.0)
    prev = syn.analytic_depth()
    out = ref.utils.get_non_differentiable_rectangle_depth_estimation(syn.pose(10), syn.pose(9), prev, fullK, halfK, 320, 256)
    low = torch.nn.functional.interpolate(out, scale_factor=1.0 / 16.0, mode="nearest")
    print("KAT-REPROJ", out.double().sum().item(), int((out != 0).sum()), out[0, 0, 64, 80].item(), low[0, 0, 0])
    assert abs(out.double().sum().item() - 30690.716363) < 0.05 and int((out != 0).sum()) == 20307, "KAT-REPROJ does not reproduce"
    o = oracle.reproject_depth(syn.pose(10), syn.pose(9), prev, fullK, halfK, 320, 256)
    REPORT["reproject_kat"] = {"oracle_vs_reference_max_abs": maxdiff(out, o), "mismatching_pixels": int((out != o).sum())}
    assert torch.equal(oracle.nearest_downsample(out, 16), low)
    # a second, harder case: larger motion (pose 16 <- 9) with a depth map that contains zeros and a far wall
    prev2 = prev.clone()
    prev2[:, :, 40:90, 100:180] = 0.0
    prev2[:, :, 150:, :] = 6.0
    out2 = ref.utils.get_non_differentiable_rectangle_depth_estimation(syn.pose(16), syn.pose(9), prev2, fullK, halfK, 320, 256)
    o2 = oracle.reproject_depth(syn.pose(16), syn.pose(9), prev2, fullK, halfK, 320, 256)
    REPORT["reproject_hard"] = {"oracle_vs_reference_max_abs": maxdiff(out2, o2), "mismatching_pixels": int((out2 != o2).sum()),
                                "nonzero": int((out2 != 0).sum())}
    save("reproject", kat=out, kat_low=low, hard=out2)
    return low


def lstm_goldens(ref, de16):
    oracle = ref.oracle
    lK = syn.scaled_K(syn.full_K(), 32.0)
    weight, x, h0, c0 = syn.analytic_lstm_inputs()
    cell = ref.convlstm.MVSLayernormConvLSTMCell(512, 512, (3, 3), torch.celu)
    with torch.no_grad():
        cell.conv.weight.copy_(weight)
        hn, cn = cell(x, [h0, c0], syn.pose(9), syn.pose(10), de16, lK)
    print("KAT-LSTM", hn.double().sum().item(), hn.double().abs().sum().item(), cn.double().abs().sum().item(), hn[0, 7, 3, 4].item(),
          cn[0, 300, 7, 9].item())
    assert abs(hn.double().abs().sum().item() - 14085.416424) < 0.05 and abs(hn[0, 7, 3, 4].item() - 0.72712857) < 3e-5, "KAT-LSTM"  # K=9216 fp32 conv: order-dependent
    ohn, ocn = oracle.convlstm_cell(weight, x, h0, c0, syn.pose(9), syn.pose(10), de16, lK)
    REPORT["lstm_cell_kat"] = {"h_max_abs": maxdiff(hn, ohn), "c_max_abs": maxdiff(cn, ocn)}

    # warp alone, incl. the mask: depth with a block of invalid pixels
    T = torch.inverse(syn.pose(9)).bmm(syn.pose(10))
    depth_masked = de16.clone()
    depth_masked[:, :, 2:5, 3:7] = 0.0
    depth_masked[:, :, 6, 1] = 0.005
    with torch.no_grad():
        warped = ref.utils.warp_frame_depth(h0, de16, T, lK)
        warped_masked = ref.utils.warp_frame_depth(h0, depth_masked, T, lK)
        warped_masked[(depth_masked <= 0.01).expand_as(warped_masked)] = 0.0
    REPORT["hidden_warp"] = {"oracle_vs_reference_max_abs": maxdiff(warped, oracle.warp_hidden_state(h0, de16, T, lK)),
                             "masked_max_abs": maxdiff(warped_masked, oracle.warp_hidden_state(h0, depth_masked, T, lK, zero_invalid=True))}
    # a larger relative motion so that many samples leave the 8x10 map
    T2 = torch.inverse(syn.pose(9)).bmm(syn.pose(40))
    with torch.no_grad():
        warped_far = ref.utils.warp_frame_depth(h0, de16, T2, lK)
    REPORT["hidden_warp"]["far_max_abs"] = maxdiff(warped_far, oracle.warp_hidden_state(h0, de16, T2, lK))
    save("hidden_warp", depth=de16, depth_masked=depth_masked, T=T, T_far=T2, warped=warped, warped_masked=warped_masked,
         warped_far=warped_far)

    # gates alone on an analytic conv output
    o = np.arange(2048, dtype=np.float64).reshape(-1, 1, 1)
    yy = np.arange(8, dtype=np.float64).reshape(1, -1, 1)
    xx = np.arange(10, dtype=np.float64).reshape(1, 1, -1)
    cc = torch.from_numpy((2.0 * np.sin(0.013 * o + 0.7 * yy + 0.3 * xx) + 0.5 * np.cos(0.05 * o * xx)).astype(np.float32)).unsqueeze(0)

    def ref_gates(cc_t, c_t):
        cc_i, cc_f, cc_o, cc_g = torch.split(cc_t, 512, dim=1)
        i, f, og = torch.sigmoid(cc_i), torch.sigmoid(cc_f), torch.sigmoid(cc_o)
        g = torch.celu(torch.layer_norm(cc_g, [8, 10]))
        cnext = torch.layer_norm(f * c_t + i * g, [8, 10])
        return og * torch.celu(cnext), cnext

    # the cell with an identity-like conv is not expressible, so drive the reference cell's own gate code through a
    # conv whose output we control: a 1x1 "conv" replaced by precomputed cc is exactly lines 45-59 of convlstm.py,
    # restated in ref_gates above with the same torch calls; cross-check it against the real cell on the KAT inputs.
    with torch.no_grad():
        cc_kat = cell.conv(torch.cat([x, oracle.warp_hidden_state(h0, de16, T, lK, zero_invalid=True)], dim=1))
        hk, ck = ref_gates(cc_kat, c0)
    assert maxdiff(hk, hn) < 1e-5 and maxdiff(ck, cn) < 1e-5
    with torch.no_grad():
        gh, gc = ref_gates(cc, c0)
    og_h, og_c = oracle.lstm_gates(cc, c0)
    REPORT["lstm_gates"] = {"h_max_abs": maxdiff(gh, og_h), "c_max_abs": maxdiff(gc, og_c)}
    save("lstm_gates", h_next=gh, c_next=gc, kat_h=hn, kat_c=cn)

    # autograd goldens (small): gates, and the warp's un-masked gradient
    cc_s = cc[:, :256].clone().reshape(1, 4, 64, 8, 10)[:, :, :64].reshape(1, 256, 8, 10).clone().requires_grad_(True)
    c_s = c0[:, :64].clone().requires_grad_(True)

    def ref_gates64(cc_t, c_t):
        cc_i, cc_f, cc_o, cc_g = torch.split(cc_t, 64, dim=1)
        i, f, og = torch.sigmoid(cc_i), torch.sigmoid(cc_f), torch.sigmoid(cc_o)
        g = torch.celu(torch.layer_norm(cc_g, [8, 10]))
        cnext = torch.layer_norm(f * c_t + i * g, [8, 10])
        return og * torch.celu(cnext), cnext

    hh, cn2 = ref_gates64(cc_s, c_s)
    gh_up = syn.smooth_noise((1, 64, 8, 10), seed=5, passes=1)
    gc_up = syn.smooth_noise((1, 64, 8, 10), seed=6, passes=1)
    (hh * gh_up).sum().add((cn2 * gc_up).sum()).backward()
    h_src = h0[:, :64].clone().requires_grad_(True)
    wm = ref.utils.warp_frame_depth(h_src, depth_masked, T, lK)
    wm.data[(depth_masked <= 0.01).expand_as(wm)] = 0.0        # exactly what convlstm.py:38-41 does
    gw_up = syn.smooth_noise((1, 64, 8, 10), seed=7, passes=1)
    (wm * gw_up).sum().backward()
    save("lstm_grads", cc=cc_s.detach(), c=c_s.detach(), grad_h=gh_up, grad_c=gc_up, grad_cc=cc_s.grad, grad_c_cur=c_s.grad,
         warp_grad_out=gw_up, warp_grad_src=h_src.grad)


def end_to_end_goldens(ref):
    """Three fusionnet keyframes through the reference modules in the order of fusionnet/run-testing.py:151-204."""
    m = ref.fusionnet_model
    ctors = (m.FeatureExtractor, m.FeatureShrinker, m.CostVolumeEncoder, m.LSTMFusion, m.CostVolumeDecoder)
    modules = syn.build_e2e_modules(ctors, with_bn_stats=False)
    fe, fs, enc, lstm, dec = modules
    named = list(zip(syn.E2E_MODULE_NAMES, modules))
    keys = {name: {k: list(v.shape) for k, v in mod.state_dict().items()} for name, mod in named}
    pm = ref.pairnet_model
    keys["pairnet_decoder"] = {k: list(v.shape) for k, v in pm.CostVolumeDecoder().state_dict().items()}
    keys["pairnet_encoder"] = {k: list(v.shape) for k, v in pm.CostVolumeEncoder().state_dict().items()}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)

    fullK = syn.full_K()
    halfK = syn.scaled_K(fullK, 2.0)
    lK = syn.scaled_K(fullK, 32.0)
    grid = ref.utils.get_warp_grid_for_cost_volume_calculation(160, 128, CPU)
    image = syn.e2e_image

    def one_frame(r, ms, lstm_state, prev_depth, prev_pose, record=None):
        meas_feats = [fs(*fe(image(i)))[0] for i in ms]
        ref_feats = fs(*fe(image(r)))
        cv = ref.utils.cost_volume_fusion(ref_feats[0], meas_feats, syn.pose(r), [syn.pose(i) for i in ms], halfK, grid,
                                          0.25, 20.0, 64, CPU, True)
        skip0, skip1, skip2, skip3, bottom = enc(*ref_feats, cv)
        if prev_depth is not None:
            de = ref.utils.get_non_differentiable_rectangle_depth_estimation(syn.pose(r), prev_pose, prev_depth, fullK, halfK, 320, 256)
            de = torch.nn.functional.interpolate(de, scale_factor=1.0 / 16.0, mode="nearest")
        else:
            de = torch.zeros(1, 1, 8, 10)
        lstm_state = lstm(bottom, lstm_state, prev_pose, syn.pose(r), de, lK)
        pred = dec(image(r), skip0, skip1, skip2, skip3, lstm_state[0])[0]
        if record is not None:
            record(ref_feats[0], cv, bottom, de, lstm_state, pred)
        return lstm_state, pred.view(1, 1, 256, 320), syn.pose(r)

    # BatchNorm statistics := batch statistics of frame 0, so the seeded weights behave like a trained network
    r0, ms0 = syn.E2E_FRAMES[0]
    syn.calibrate_batchnorm(lambda: one_frame(r0, ms0, None, None, None), modules)
    np.savez_compressed(os.path.join(HERE, "e2e_bn_stats"), **syn.collect_bn_stats(named))
    print(f"wrote e2e_bn_stats: {os.path.getsize(os.path.join(HERE, 'e2e_bn_stats.npz')) / 1024:.0f} KiB")

    arrays = {}
    state = (None, None, None)
    with torch.no_grad():
        for n, (r, ms) in enumerate(syn.E2E_FRAMES):
            def record(feat_half, cv, bottom, de, lstm_state, pred, n=n):
                for tag, t in (("feat_half", feat_half), ("cost_volume", cv), ("bottom", bottom), ("depth_estimation", de),
                               ("h", lstm_state[0]), ("c", lstm_state[1]), ("depth", pred)):
                    arrays.update(pins_to_arrays(f"f{n}_{tag}", syn.tensor_pins(t)))
                arrays[f"f{n}_depth_sub4"] = pred[0, ::4, ::4]
                arrays[f"f{n}_depth_estimation_full"] = de
                print(f"frame {n}: depth range {pred.min().item():.4f} .. {pred.max().item():.4f}, mean {pred.mean().item():.4f}; "
                      f"cv mean|.| {cv.abs().mean().item():.4f}; bottom std {bottom.std().item():.3f}; h std {lstm_state[0].std().item():.3f}")
            state = one_frame(r, ms, *state, record=record)
    # float64 arbitration run: same weights, same inputs, the oracle pipeline in double precision.  It tells how far the
    # reference's OWN float32 forward is from the exact arithmetic, i.e. what "equal up to fp32 round-off" means for
    # this (random-weight, hence sensitive) network.
    import copy
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    sys.modules.setdefault("dvmvs_oracle", ref.oracle)
    from fusionnet_cpu import CpuDepthPipeline
    pipe64 = CpuDepthPipeline(*[copy.deepcopy(mod).double() for mod in modules])
    for n, (r, ms) in enumerate(syn.E2E_FRAMES):
        rec = {}
        pipe64.step(image(r).double(), syn.pose(r).double(), [image(i).double() for i in ms], [syn.pose(i).double() for i in ms],
                    fullK.double(), record=lambda **kw: rec.update(kw))
        ref32 = torch.from_numpy(arrays[f"f{n}_depth_sub4"].numpy() if isinstance(arrays[f"f{n}_depth_sub4"], torch.Tensor)
                                 else arrays[f"f{n}_depth_sub4"]).double()
        d64 = rec["depth"][0, ::4, ::4]
        arrays[f"f{n}_depth64_sub4"] = d64
        for tag in ("h", "c", "bottom", "cost_volume"):
            arrays[f"f{n}_{tag}64_samples"] = rec[tag].reshape(-1)[syn.sample_indices(rec[tag].numel())]
        rel = float(((ref32 - d64).abs() / d64).mean())
        REPORT[f"e2e_frame{n}_reference_fp32_vs_float64_depth_rel_l1"] = rel
        print(f"frame {n}: reference fp32 vs float64 depth rel-L1 {rel:.3e}")
    save("fusionnet_e2e", **arrays)

    # pairnet: one frame, M = 1 (reference pairnet/run-testing.py:136-166); same weights minus the LSTM
    pctors = (pm.FeatureExtractor, pm.FeatureShrinker, pm.CostVolumeEncoder, pm.CostVolumeDecoder)
    pfe, pfs, penc, pdec = syn.build_e2e_modules(pctors, with_bn_stats=True)
    with torch.no_grad():
        mf = pfs(*pfe(image(9)))[0]
        rf = pfs(*pfe(image(12)))
        cv = ref.utils.cost_volume_fusion(rf[0], [mf], syn.pose(12), [syn.pose(9)], halfK, grid, 0.25, 20.0, 64, CPU, True)
        s0, s1, s2, s3, bottom = penc(*rf, cv)
        pred = pdec(image(12), s0, s1, s2, s3, bottom)[0]
    print(f"pairnet: depth range {pred.min().item():.4f} .. {pred.max().item():.4f}, mean {pred.mean().item():.4f}")
    parr = {}
    for tag, t in (("cost_volume", cv), ("bottom", bottom), ("depth", pred)):
        parr.update(pins_to_arrays(tag, syn.tensor_pins(t)))
    parr["depth_sub4"] = pred[0, ::4, ::4]
    save("pairnet_e2e", **parr)


def long_sequence_goldens(ref):
    """The REFERENCE's fusionnet loop (fusionnet/run-testing.py:86-101,151-204) over syn.LONG_SCHEDULE: 14 keyframes of the
    sample scene's nmeas+2 index incl. a "TRACKING LOST" and the wide-baseline lines, same seeded weights + BatchNorm statistics
    as the 3-frame golden.  Stored per frame: the depth (4x sub-sampled), the 8x10 depth estimate fed to the ConvLSTM, the
    number of z-buffer pixels of the half-resolution re-projection, sampled cost-volume / hidden-state entries."""
    m = ref.fusionnet_model
    ctors = (m.FeatureExtractor, m.FeatureShrinker, m.CostVolumeEncoder, m.LSTMFusion, m.CostVolumeDecoder)
    fe, fs, enc, lstm, dec = syn.build_e2e_modules(ctors, with_bn_stats=True)
    for mod in (fe, fs, enc, lstm, dec):
        mod.eval()
    fullK = syn.full_K()
    halfK = syn.scaled_K(fullK, 2.0)
    lK = syn.scaled_K(fullK, 32.0)
    grid = ref.utils.get_warp_grid_for_cost_volume_calculation(160, 128, CPU)
    lines = syn.keyframe_index_lines(2)
    arrays = {"schedule": np.array([-1 if item is None else item for item in syn.LONG_SCHEDULE])}
    lstm_state, prev_depth, prev_pose = None, None, None
    with torch.no_grad():
        for n, item in enumerate(syn.LONG_SCHEDULE):
            if item is None:                                  # run-testing.py:97-101
                lstm_state, prev_depth, prev_pose = None, None, None
                continue
            r, ms = lines[item]
            meas_feats = [fs(*fe(syn.e2e_image(i)))[0] for i in ms]
            ref_feats = fs(*fe(syn.e2e_image(r)))
            cv = ref.utils.cost_volume_fusion(ref_feats[0], meas_feats, syn.pose(r), [syn.pose(i) for i in ms], halfK, grid,
                                              0.25, 20.0, 64, CPU, True)
            skip0, skip1, skip2, skip3, bottom = enc(*ref_feats, cv)
            if prev_depth is not None:
                de_half = ref.utils.get_non_differentiable_rectangle_depth_estimation(syn.pose(r), prev_pose, prev_depth, fullK, halfK, 320, 256)
                de = torch.nn.functional.interpolate(de_half, scale_factor=1.0 / 16.0, mode="nearest")
            else:
                de_half, de = torch.zeros(1, 1, 128, 160), torch.zeros(1, 1, 8, 10)
            lstm_state = lstm(bottom, lstm_state, prev_pose, syn.pose(r), de, lK)
            pred = dec(syn.e2e_image(r), skip0, skip1, skip2, skip3, lstm_state[0])[0]
            prev_depth, prev_pose = pred.view(1, 1, 256, 320), syn.pose(r)
            arrays[f"s{n}_depth_sub4"] = pred[0, ::4, ::4]
            arrays[f"s{n}_depth_estimation"] = de
            arrays[f"s{n}_estimate_half_nonzero"] = int((de_half != 0).sum())
            arrays[f"s{n}_cost_volume_samples"] = cv.reshape(-1)[syn.sample_indices(cv.numel())]
            arrays[f"s{n}_h_samples"] = lstm_state[0].reshape(-1)[syn.sample_indices(lstm_state[0].numel())]
            print(f"long sequence step {n} (index line {item}: {r} <- {ms}): depth {pred.min().item():.3f} .. {pred.max().item():.3f}, "
                  f"estimate pixels {int((de != 0).sum())}/80")
    save("fusionnet_long", **arrays)


def reference_state_goldens(ref):
    """fusionnet_state.npz: the FULL-RESOLUTION recurrent state the reference's loop carries out of every keyframe of the two golden
    runs (3 golden frames: keys f{n}_*, long run: keys s{n}_*): depth [256,320], h and c [512,8,10], float32, exactly as the
    reference's modules returned them.  fusionnet_e2e.npz / fusionnet_long.npz hold sub-sampled depth and sampled state entries
    only; with this file a test (and bench.py's rel_l1 leg) installs the REFERENCE's own state before each step (teacher forcing)
    instead of the state of a CPU stand-in pipeline.  The run is the same one, re-run in this process: checked here bit for bit
    against the sub-sampled depth and the depth estimate the two older fixtures hold."""
    m = ref.fusionnet_model
    ctors = (m.FeatureExtractor, m.FeatureShrinker, m.CostVolumeEncoder, m.LSTMFusion, m.CostVolumeDecoder)
    fe, fs, enc, lstm, dec = syn.build_e2e_modules(ctors, with_bn_stats=True)
    for mod in (fe, fs, enc, lstm, dec):
        mod.eval()
    fullK = syn.full_K()
    halfK = syn.scaled_K(fullK, 2.0)
    lK = syn.scaled_K(fullK, 32.0)
    grid = ref.utils.get_warp_grid_for_cost_volume_calculation(160, 128, CPU)
    lines = syn.keyframe_index_lines(2)
    z3 = np.load(os.path.join(HERE, "fusionnet_e2e.npz"))
    zl = np.load(os.path.join(HERE, "fusionnet_long.npz"))
    runs = [("f", list(syn.E2E_FRAMES), lambda n: (z3[f"f{n}_depth_sub4"], z3[f"f{n}_depth_estimation_full"])),
            ("s", [None if i is None else lines[i] for i in syn.LONG_SCHEDULE], lambda n: (zl[f"s{n}_depth_sub4"], zl[f"s{n}_depth_estimation"]))]
    arrays = {}
    with torch.no_grad():
        for tag, frames, older in runs:
            lstm_state, prev_depth, prev_pose = None, None, None
            for n, item in enumerate(frames):
                if item is None:                                  # run-testing.py:97-101
                    lstm_state, prev_depth, prev_pose = None, None, None
                    continue
                r, ms = item
                meas_feats = [fs(*fe(syn.e2e_image(i)))[0] for i in ms]
                ref_feats = fs(*fe(syn.e2e_image(r)))
                cv = ref.utils.cost_volume_fusion(ref_feats[0], meas_feats, syn.pose(r), [syn.pose(i) for i in ms], halfK, grid,
                                                  0.25, 20.0, 64, CPU, True)
                skip0, skip1, skip2, skip3, bottom = enc(*ref_feats, cv)
                if prev_depth is not None:
                    de = ref.utils.get_non_differentiable_rectangle_depth_estimation(syn.pose(r), prev_pose, prev_depth, fullK, halfK, 320, 256)
                    de = torch.nn.functional.interpolate(de, scale_factor=1.0 / 16.0, mode="nearest")
                else:
                    de = torch.zeros(1, 1, 8, 10)
                lstm_state = lstm(bottom, lstm_state, prev_pose, syn.pose(r), de, lK)
                pred = dec(syn.e2e_image(r), skip0, skip1, skip2, skip3, lstm_state[0])[0]
                prev_depth, prev_pose = pred.view(1, 1, 256, 320), syn.pose(r)
                old_depth, old_estimate = older(n)
                assert np.array_equal(pred[0, ::4, ::4].numpy(), old_depth), f"{tag}{n}: this run is not the run of the older fixture"
                assert np.array_equal(de.numpy().reshape(old_estimate.shape), old_estimate), f"{tag}{n}: depth estimate differs"
                arrays[f"{tag}{n}_depth"] = pred[0].clone()
                arrays[f"{tag}{n}_h"] = lstm_state[0][0].clone()
                arrays[f"{tag}{n}_c"] = lstm_state[1][0].clone()
                print(f"reference state {tag}{n}: depth {pred.min().item():.3f} .. {pred.max().item():.3f} (equal to the older fixture's sub-sampled depth)")
    save("fusionnet_state", **arrays)


def pose_algebra_goldens(ref):
    """host_pose_algebra.npz: the small fp32 matrices of THIS host -- the one every other fixture here was captured on -- for
    the pose pairs the fixture-comparing tests use (syn.golden_algebra_pairs), by the reference's own expressions
    (dvmvs/utils.py:51-56, :121; dvmvs/convlstm.py:30).  torch.inverse is LAPACK and LAPACK's last bits depend on the CPU (see
    tests/synthetic.py), so these rows are what ties the depth fixtures to inputs a test can feed on another host.
    Checked here: the same expressions inside the reference (cost_volume_fusion on the same process) give a volume that the
    oracle reproduces to 1e-7 from THESE matrices."""
    sweeps, rels = syn.golden_algebra_pairs()
    sweep_in, sweep_out, rel_in, rel_out = [], [], [], []
    for r, m, tag in sweeps:
        pose1, pose2, K = syn.pose(r), syn.pose(m), syn.golden_algebra_K(tag)
        extrinsic2 = torch.inverse(pose2).bmm(pose1)
        R = extrinsic2[:, 0:3, 0:3]
        t = extrinsic2[:, 0:3, 3].unsqueeze(-1)
        Kt = K.bmm(t)
        K_R_Kinv = K.bmm(R).bmm(torch.inverse(K))
        sweep_in.append(torch.cat([pose1.reshape(-1), pose2.reshape(-1), K.reshape(-1)]).numpy())
        sweep_out.append(torch.cat([K_R_Kinv.reshape(-1), Kt.reshape(-1)]).numpy())
    for a, c in rels:
        pa, pc = syn.pose(a), syn.pose(c)
        rel_in.append(torch.cat([pa.reshape(-1), pc.reshape(-1)]).numpy())
        rel_out.append(torch.bmm(torch.inverse(pa), pc).reshape(-1).numpy())
    save("host_pose_algebra", sweep_in=np.stack(sweep_in), sweep_out=np.stack(sweep_out), rel_in=np.stack(rel_in), rel_out=np.stack(rel_out))
    # the reference itself, fed these poses, lands on the volume the oracle computes FROM THE RECORDED MATRICES
    table = syn.FixtureHostAlgebra()
    halfK = syn.scaled_K(syn.full_K(), 2.0)
    feats = [syn.analytic_features(s) for s in range(3)]
    grid = ref.utils.get_warp_grid_for_cost_volume_calculation(160, 128, CPU)
    cv = ref.utils.cost_volume_fusion(feats[0], [feats[1], feats[2]], syn.pose(202), [syn.pose(196), syn.pose(188)], halfK, grid, 0.25, 20.0, 64, CPU, True)
    saved = ref.oracle.plane_sweep_setup
    ref.oracle.plane_sweep_setup = table.plane_sweep_setup
    try:
        ocv = ref.oracle.cost_volume_fusion(feats[0], [feats[1], feats[2]], syn.pose(202), [syn.pose(196), syn.pose(188)], halfK, 0.25, 20.0, 64, True)
    finally:
        ref.oracle.plane_sweep_setup = saved
    REPORT["host_pose_algebra_replay"] = {"pairs": len(sweeps), "relative_poses": len(rels),
                                          "reference_vs_oracle_from_recorded_matrices_max_abs": maxdiff(cv, ocv)}
    print("host pose algebra:", REPORT["host_pose_algebra_replay"])
    assert maxdiff(cv, ocv) < 5e-7


def keyframe_goldens(ref):
    """Replays the reference KeyframeBuffer over the sample poses; must reproduce the shipped index files."""
    poses = syn.sample_poses()
    for n in (1, 2, 3):
        buf = ref.keyframe_buffer.KeyframeBuffer(30, 0.1, 0.15, 0.0, store_return_indices=True)
        lines = []
        for i in range(len(poses)):
            code = buf.try_new_keyframe(poses[i], None, index=i)
            if code == 3:
                lines.append("TRACKING LOST")
            elif code == 1:
                meas = buf.get_best_measurement_frames(n)
                lines.append(" ".join([syn.sample_image_name(i)] + [syn.sample_image_name(m[2]) for m in meas]))
        shipped = open(os.path.join(HERE, "indices", f"keyframe+hololens-dataset+000+nmeas+{n}")).read().split("\n")
        shipped = [s for s in shipped if s]
        REPORT[f"keyframe_index_nmeas{n}"] = {"lines": len(lines), "matching": sum(a == b for a, b in zip(lines, shipped)),
                                              "shipped": len(shipped)}


def error_metric_goldens(ref):
    """The eight depth metrics of dvmvs/errors.py:4-28 on deterministic maps (incl. invalid pixels and a max_depth cut)."""
    gt, pred = syn.error_metric_inputs()
    save("error_metrics", all_pixels=np.array(ref.errors.compute_errors(gt, pred)),
         max_depth_2=np.array(ref.errors.compute_errors(gt, pred, 2.0)),
         nothing_valid=np.array(ref.errors.compute_errors(np.zeros((4, 4)), np.ones((4, 4)))))


def loss_goldens(ref):
    """dvmvs/losses.py:26-82 on deterministic maps: the four sums + valid count of calculate_loss at three resolutions, and
    update_losses (optimizer loss + meter state) for every loss type in training mode and once in evaluation mode."""
    gt, preds = syn.loss_inputs()
    out = {}
    for j, p in enumerate(preds):
        l1, huber, l1_inv, l1_rel, count = ref.losses.calculate_loss(groundtruth=gt, prediction=p)
        out[f"calc{j}"] = np.array([l1.item(), huber.item(), l1_inv.item(), l1_rel.item(), float(count)], dtype=np.float64)
    weights = [0.5, 1.0, 2.0]
    for loss_type in ("L1", "L1-inv", "L1-rel", "Huber"):
        meters = [ref.losses.LossMeter() for _ in range(4)]
        total = ref.losses.update_losses(preds, weights, gt, True, meters[0], meters[1], meters[2], meters[3], loss_type)
        total = total + ref.losses.update_losses(preds[::-1], weights, gt, True, meters[0], meters[1], meters[2], meters[3], loss_type)
        out[f"train_{loss_type}"] = np.array([float(total)] + [v for m in meters for v in (m.sum, m.count, m.avg, m.item_average)], dtype=np.float64)
        out[f"repr_{loss_type}"] = np.array([repr(m) for m in meters])
    meters = [ref.losses.LossMeter() for _ in range(4)]
    total = ref.losses.update_losses(preds, weights, gt, False, meters[0], meters[1], meters[2], meters[3], "L1")
    out["eval"] = np.array([float(total)] + [v for m in meters for v in (m.sum, m.count, m.avg, m.item_average)], dtype=np.float64)
    save("losses", **out)


def tsdf_goldens():
    """The reference's TSDFVolume CPU fall-back (sample-data/run-tsdf-reconstruction.py:222-302) on two deterministic frames:
    the script is loaded where it lies with stand-ins for its absent imports (numba.njit -> identity, prange -> range,
    skimage / cv2 / path / pycuda missing -> CPU mode)."""
    import importlib.util
    import types
    numba = types.ModuleType("numba")
    numba.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    numba.prange = range
    sys.modules["numba"] = numba
    sk = types.ModuleType("skimage")
    sk.measure = types.ModuleType("skimage.measure")
    sys.modules.update({"skimage": sk, "skimage.measure": sk.measure})
    for name in ("gc",):
        __import__(name)
    spec = importlib.util.spec_from_file_location("ref_tsdf", "/root/reference/sample-data/run-tsdf-reconstruction.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.FUSION_GPU_MODE == 0
    frames, bounds, voxel = syn.tsdf_inputs()
    vol = mod.TSDFVolume(bounds.copy(), voxel, use_gpu=False)
    out = {}
    for n, (color, depth, K, pose) in enumerate(frames):
        vol.integrate(color, depth, K, pose, obs_weight=1.0 + n)
        tsdf, col = vol.get_volume()
        out[f"tsdf{n}"], out[f"color{n}"], out[f"weight{n}"] = tsdf.copy(), col.copy(), vol._weight_vol_cpu.copy()
    out["vol_dim"], out["vol_origin"] = vol._vol_dim, vol._vol_origin
    out["frustum"] = mod.TSDFFusion.get_view_frustum(frames[0][1], frames[0][2], frames[0][3])
    save("tsdf", **out)


def crawler_goldens(ref):
    """The reference's training-sample crawler (dataset_loader.py:112-219) on the sample scene's 373 poses and on a synthetic
    trajectory: which frame indices form each sample, for sub-sequence lengths 2 (pairs), 3 and 8."""
    import json
    import pathlib
    import types
    import tempfile
    dl = ref.dataset_loader
    dl.Path = pathlib.Path           # the 'path' package is absent; pathlib joins the same way
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        scenes = {"sample": syn.sample_poses(), "synthetic": syn.synthetic_trajectory(240, seed=77)}
        for scene, poses in scenes.items():
            os.makedirs(os.path.join(tmp, scene))
            np.savetxt(os.path.join(tmp, scene, "poses.txt"), np.reshape(poses, (-1, 16)))
            progress = types.SimpleNamespace(value=0)
            out[f"{scene}_pairs"] = [s["indices"] for s in dl.crawl_subprocess_short(scene, tmp, 1, progress)]
            for length in (3, 8):
                out[f"{scene}_len{length}"] = [s["indices"] for s in dl.crawl_subprocess_long(scene, tmp, 1, progress, length)]
            measures = [dl.is_valid_pair(poses[0], poses[j], 0.125, 0.325, return_measure=True) for j in (1, 5, 20)]
            out[f"{scene}_is_valid_pair"] = [[bool(v), float(m)] for v, m in measures]
    for k, v in out.items():
        print(k, len(v))
    with open(os.path.join(HERE, "crawler.json"), "w") as f:
        json.dump(out, f)


def main():
    if "--only-crawler" in sys.argv:
        crawler_goldens(import_reference())
        return
    if "--only-tsdf" in sys.argv:
        import_reference()     # installs the cv2 / path stand-ins the script's own imports need
        tsdf_goldens()
        return
    ref = import_reference()
    if "--only-errors" in sys.argv:      # adds one fixture without rewriting the others
        error_metric_goldens(ref)
        return
    if "--only-losses" in sys.argv:
        loss_goldens(ref)
        return
    if "--only-long-sequence" in sys.argv:
        long_sequence_goldens(ref)
        return
    if "--only-pose-algebra" in sys.argv:
        pose_algebra_goldens(ref)
        return
    if "--only-reference-state" in sys.argv:
        reference_state_goldens(ref)
        return
    cost_volume_goldens(ref)
    de16 = reprojection_goldens(ref)
    lstm_goldens(ref, de16)
    end_to_end_goldens(ref)
    long_sequence_goldens(ref)
    reference_state_goldens(ref)
    pose_algebra_goldens(ref)
    keyframe_goldens(ref)
    error_metric_goldens(ref)
    loss_goldens(ref)
    tsdf_goldens()
    crawler_goldens(ref)
    REPORT["_meta"] = {"torch": torch.__version__, "reference": "ardaduz/deep-video-mvs @ /root/reference", "device": "cpu",
                       "note": "differences are |oracle - reference| on identical inputs, float32"}
    with open(os.path.join(HERE, "PINNING_REPORT.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)
    print(json.dumps(REPORT, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

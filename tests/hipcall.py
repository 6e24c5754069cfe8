"""Test-side convenience over the ABI-3 custom ops: the reference's call shapes (poses + intrinsics in) on top of ops that take
the small matrices (dvmvs.pose_algebra).  Tensors may live on the host or on the device; the matrices are evaluated in the
product's default "reference" mode (host fp32, the reference's own expressions) unless ``mode`` says otherwise."""


def cost_volume(ops, f1, f2s, p1, p2s, K, lo, hi, D, dot, variant, mode=None):
    from dvmvs import pose_algebra
    f2s, p2s = list(f2s), list(p2s)
    if len(f2s) != len(p2s):
        raise ValueError("need as many measurement poses as measurement feature maps")
    if len(f2s) == 0:
        raise ValueError("need at least one measurement frame")
    if len(p2s) > 8:   # beyond the ABI: let the op report it (matrices of the first 8 frames only would not match M)
        Hm, kt = pose_algebra.sweep_matrices(p1, p2s[:1], K, f1.device, mode)
        Hm, kt = Hm.repeat(1, len(p2s), 1), kt.repeat(1, len(p2s), 1)
    else:
        Hm, kt = pose_algebra.sweep_matrices(p1, p2s, K, f1.device, mode)
    return ops.cost_volume(f1, f2s, Hm, kt, lo, hi, D, dot, variant)


def depth_reproject(ops, reference_pose, measurement_pose, previous_depth, full_K, half_K, factor=0, mode=None):
    from dvmvs import pose_algebra
    T = pose_algebra.relative_pose(reference_pose, measurement_pose, previous_depth.device, mode)
    if factor:
        return ops.depth_reproject_lowres(T, previous_depth, full_K, half_K, factor)
    return ops.depth_reproject(T, previous_depth, full_K, half_K)

"""CPU checks of the host side of csrc/direct_conv.hip (no GPU: the shape choice and the packed-weight sizes are host functions of
libdvmvs_hip.so): which problems of a 320x256 frame the kernel takes and with which output tiling, that it fills exactly 256 workgroups
on them, and the size formula of the packed weights."""
import math

import pytest

from dvmvs.hip import _capi


def lib():
    return _capi.lib()


# (C_in, H, W, C_out, k, stride) -> output-channel tiles per wave; the layers of fusionnet/model.py:167-305 at 320x256
FRAME_LAYERS = {
    (36, 256, 320, 32, 5, 1): 2, (32, 256, 320, 32, 5, 1): 2, (3, 256, 320, 32, 3, 2): 2,
    (96, 128, 160, 32, 5, 1): 2, (64, 128, 160, 32, 5, 1): 2, (65, 128, 160, 32, 5, 1): 2, (32, 128, 160, 32, 5, 1): 2, (32, 128, 160, 32, 3, 1): 2,
    (32, 128, 160, 64, 5, 2): 1, (64, 64, 80, 64, 5, 1): 1, (96, 64, 80, 64, 3, 1): 1, (128, 64, 80, 64, 3, 1): 1, (129, 64, 80, 64, 3, 1): 1,
    (64, 64, 80, 128, 3, 2): 1, (128, 32, 40, 128, 3, 1): 1, (160, 32, 40, 128, 3, 1): 1, (256, 32, 40, 128, 3, 1): 1, (257, 32, 40, 128, 3, 1): 1,
}


@pytest.mark.parametrize("layer", sorted(FRAME_LAYERS))
def test_frame_layers_are_taken_with_the_expected_tile(layer):
    C_in, H, W, C_out, k, s = layer
    assert lib().dvmvs_direct_conv_tile(1, C_in, H, W, C_out, k, s) == FRAME_LAYERS[layer]


def test_workgroup_counts_on_the_frames_maps():
    """The point of the four shapes: 256 workgroups (one per CU) on the full, 1/2 and 1/4 resolution maps, 128 on the 1/8 maps."""
    # (output H, W, C_out, tile) -> workgroups = column tiles x row tiles x channel tiles
    shapes = {2: [(4, 80, 32), (2, 40, 32)], 1: [(1, 80, 16), (2, 40, 16)]}      # tile -> candidate (rows, columns, channels) per workgroup

    def groups(OH, OW, C_out, tile):
        best = 0.0, 0
        for rows, cols, ch in shapes[tile]:
            if OW % cols or C_out % ch:
                continue
            n = (OW // cols) * math.ceil(OH / rows) * (C_out // ch)
            fill = n / (256 * math.ceil(n / 256))
            if fill > best[0] + 1e-9:
                best = fill, n
        return best[1]

    assert groups(256, 320, 32, 2) == 256 and groups(128, 160, 32, 2) == 256 and groups(64, 80, 64, 1) == 256 and groups(32, 40, 128, 1) == 128


@pytest.mark.parametrize("problem", [(1, 512, 16, 20, 256, 3, 1), (1, 512, 8, 10, 512, 3, 1), (1, 32, 64, 80, 1, 3, 1), (1, 32, 64, 80, 32, 7, 1),
                                     (1, 32, 64, 80, 32, 3, 3), (1, 32, 63, 80, 32, 3, 2), (1, 32, 64, 81, 32, 3, 1), (0, 32, 64, 80, 32, 3, 1),
                                     (1, 32, 64, 80, 24, 3, 1)])
def test_problems_left_to_the_library_convolution(problem):
    assert lib().dvmvs_direct_conv_tile(*problem) == 0


def test_batches_keep_the_tile_of_the_single_frame():
    for layer, tile in FRAME_LAYERS.items():
        C_in, H, W, C_out, k, s = layer
        assert lib().dvmvs_direct_conv_tile(8, C_in, H, W, C_out, k, s) in (1, 2)


def test_packed_weight_sizes():
    """[output tile][input-channel groups padded to 16][k rows][ceil(n_tile k / 4) quads][64 lanes] float4."""
    for C_out, C_in, k, tile in [(32, 36, 5, 2), (32, 3, 3, 2), (128, 257, 3, 1), (64, 64, 5, 1), (16, 5, 5, 1), (48, 7, 3, 1)]:
        groups = math.ceil(math.ceil(C_in / 4) / 16) * 16
        want = 4 * (C_out // (16 * tile)) * groups * k * math.ceil(tile * k / 4) * 256
        assert lib().dvmvs_direct_conv_packed_bytes(C_out, C_in, k, tile) == want
    assert lib().dvmvs_direct_conv_packed_bytes(48, 8, 3, 2) == 0       # 48 channels are not a multiple of the 32-channel tile
    assert lib().dvmvs_direct_conv_packed_bytes(32, 8, 4, 2) == 0
    assert lib().dvmvs_direct_conv_packed_bytes(32, 8, 3, 3) == 0

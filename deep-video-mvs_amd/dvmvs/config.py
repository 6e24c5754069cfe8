"""Static run configuration of the MI355X plane-sweep depth engine.

Mirrors the attribute surface of the reference's ``dvmvs.config.Config`` (/root/reference/dvmvs/config.py:4-51)
so that scripts doing ``from dvmvs.config import Config`` and reading/mutating class attributes keep working.
Values are the reference's defaults; folder locations come from the environment instead of a developer's home
directory (``DVMVS_DATASET``, ``DVMVS_TRAIN_RUNS``, ``DVMVS_ONLINE_SCENE``, ``DVMVS_OFFLINE_DATA``,
``DVMVS_RESULTS``).
"""
import os
import time


def _env(name, default):
    return os.environ.get(name, default)


class Config:
    """Class-attribute bag; never instantiated (same usage as the reference)."""

    # ---- plane sweep / network geometry used by BOTH training and the model constructors ----
    train_min_depth, train_max_depth, train_n_depth_levels = 0.25, 20.0, 64
    train_image_width = train_image_height = 256

    # ---- training sampler / loop knobs ----
    train_minimum_pose_distance, train_maximum_pose_distance = 0.125, 0.325
    train_crawl_step = 3
    train_subsequence_length = None     # set by the training script (8 for fusionnet, 2/3 for pairnet)
    train_predict_two_way = None        # set by the pairnet training script
    train_freeze_batch_normalization = False
    train_data_pipeline_workers = 8
    train_epochs = 100000
    train_print_frequency = 5000
    train_validate = True
    train_seed = int(round(time.time()))

    # ---- inference ----
    test_image_width, test_image_height = 320, 256
    test_distortion_crop = 0
    test_perform_crop = False
    test_visualize = True
    test_n_measurement_frames = 2
    test_keyframe_buffer_size = 30
    test_keyframe_pose_distance = 0.1
    test_optimal_t_measure = 0.15
    test_optimal_R_measure = 0.0

    # ---- locations ----
    dataset = _env("DVMVS_DATASET", "/data/dvmvs/train")
    train_run_directory = _env("DVMVS_TRAIN_RUNS", "./training-runs")
    test_online_scene_path = _env("DVMVS_ONLINE_SCENE", "./sample-data/hololens-dataset/000")
    test_offline_data_path = _env("DVMVS_OFFLINE_DATA", "./sample-data")
    test_dataset_name = "hololens-dataset"  # or None = every dataset with index files under test_offline_data_path
    test_result_folder = _env("DVMVS_RESULTS", "./results")

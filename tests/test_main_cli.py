"""CPU: the command line of the reference (P/main.py:5-27) - flag names, types, defaults and the substring `str2bool`."""
import importlib

import pytest


def _parser():
    return importlib.import_module("srmi.main").build_parser()


REFERENCE_DEFAULTS = {      # P/main.py:10-27
    "LR_path": 'C:/SRGAN_CustomDataset-uav2/SRGAN_CustomDataset-main/custom_dataset/hr_train_LR',
    "GT_path": 'C:/SRGAN_CustomDataset-uav2/SRGAN_CustomDataset-main/custom_dataset/hr_train_HR',
    "res_num": 16, "num_workers": 0, "batch_size": 16, "L2_coeff": 1.0, "adv_coeff": 1e-3, "tv_loss_coeff": 0.0,
    "pre_train_epoch": 80000, "fine_train_epoch": 4000, "scale": 4, "patch_size": 24, "feat_layer": 'relu5_4',
    "vgg_rescale_coeff": 0.006, "fine_tuning": False, "in_memory": True, "generator_path": None, "mode": 'train',
}


def test_all_17_flags_and_the_mode_switch_keep_names_and_defaults():
    args = vars(_parser().parse_args([]))
    for k, v in REFERENCE_DEFAULTS.items():
        assert k in args, k
        assert args[k] == v and type(args[k]) is type(v), (k, args[k])
    assert len(REFERENCE_DEFAULTS) == 18          # 17 flags + --mode


def test_build_only_flags_do_not_rename_anything():
    args = vars(_parser().parse_args(["--compute", "bf16", "--world_size", "8", "--model_dir", "m", "--seed", "3"]))
    assert args["compute"] == "bf16" and args["world_size"] == 8 and args["model_dir"] == "m" and args["seed"] == 3
    assert set(args) - set(REFERENCE_DEFAULTS) == {"compute", "world_size", "model_dir", "metrics_csv", "out_dir", "result_txt", "seed"}


@pytest.mark.parametrize("text,want", [("true", True), ("True", True), ("TRUE", True), ("t", True), ("ru", True), ("", True),
                                       ("false", False), ("1", False), ("yes", False), ("truee", False)])
def test_str2bool_is_the_reference_substring_test(text, want):
    # P/main.py:7-8: `v.lower() in ('true')` - ('true') is a string, so this is a substring test
    main = importlib.import_module("srmi.main")
    assert main.str2bool(text) is want
    assert _parser().parse_args(["--fine_tuning", text]).fine_tuning is want


def test_types_are_parsed_like_the_reference():
    a = _parser().parse_args(["--res_num", "4", "--adv_coeff", "0.01", "--in_memory", "false", "--mode", "test_only",
                              "--generator_path", "g.pt", "--patch_size", "12"])
    assert a.res_num == 4 and a.adv_coeff == 0.01 and a.in_memory is False and a.mode == "test_only"
    assert a.generator_path == "g.pt" and a.patch_size == 12

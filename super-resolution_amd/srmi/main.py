"""Command line of the reference (P/main.py:1-38): the same 17 flags with the same names and defaults, dispatching to
`train` / `test` / `test_only` of srmi.mode.

    python -m srmi.main --mode train --LR_path ... --GT_path ...
    python -m torch.distributed.run --nproc-per-node 8 -m srmi.main --mode train ...      (one process per GPU)

`str2bool` keeps the reference's substring test, `v.lower() in ('true')` (P/main.py:7-8): "true", "True", "t", "r",
"ru" and the empty string are True; "false", "1", "yes" are False.  `--L2_coeff` is parsed and unused, as in the
reference (P/main.py:15; the L2 term has weight 1, P/mode.py:174).  Build-only flags (none renamed):
`--compute {fp32,bf16}` arithmetic of the inference modes (training is fp32), `--world_size` is informational (ranks
come from the launcher's environment), `--model_dir`, `--metrics_csv`, `--out_dir`, `--result_txt` place the files the
reference writes to fixed relative paths, `--seed` seeds the host RNGs that draw batches and patches.
"""
from __future__ import annotations

import argparse
import os
import sys


def str2bool(v):
    return v.lower() in ('true')


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--LR_path", type=str, default='C:/SRGAN_CustomDataset-uav2/SRGAN_CustomDataset-main/custom_dataset/hr_train_LR')
    parser.add_argument("--GT_path", type=str, default='C:/SRGAN_CustomDataset-uav2/SRGAN_CustomDataset-main/custom_dataset/hr_train_HR')
    parser.add_argument("--res_num", type=int, default=16)
    parser.add_argument("--num_workers", type=int, default=0)
    parser.add_argument("--batch_size", type=int, default=16)
    parser.add_argument("--L2_coeff", type=float, default=1.0)
    parser.add_argument("--adv_coeff", type=float, default=1e-3)
    parser.add_argument("--tv_loss_coeff", type=float, default=0.0)
    parser.add_argument("--pre_train_epoch", type=int, default=80000)
    parser.add_argument("--fine_train_epoch", type=int, default=4000)
    parser.add_argument("--scale", type=int, default=4)
    parser.add_argument("--patch_size", type=int, default=24)
    parser.add_argument("--feat_layer", type=str, default='relu5_4')
    parser.add_argument("--vgg_rescale_coeff", type=float, default=0.006)
    parser.add_argument("--fine_tuning", type=str2bool, default=False)
    parser.add_argument("--in_memory", type=str2bool, default=True)
    parser.add_argument("--generator_path", type=str)
    parser.add_argument("--mode", type=str, default='train')
    # ---- build-only flags ----
    parser.add_argument("--compute", type=str, default=None, choices=[None, "fp32", "bf16"])
    parser.add_argument("--world_size", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    parser.add_argument("--model_dir", type=str, default="./model")
    parser.add_argument("--metrics_csv", type=str, default="./metrics.csv")
    parser.add_argument("--out_dir", type=str, default="./result")
    parser.add_argument("--result_txt", type=str, default="./result.txt")
    parser.add_argument("--seed", type=int, default=None)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    from . import mode
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    if args.seed is not None:
        import random
        rank = int(os.environ.get("RANK", "0"))
        random.seed(args.seed + rank)
        torch.manual_seed(args.seed + rank)
    try:
        if args.mode == 'train':
            mode.train(args)
        elif args.mode == 'test':
            mode.test(args, out_dir=args.out_dir, result_txt=args.result_txt)
        elif args.mode == 'test_only':
            mode.test_only(args, out_dir=args.out_dir)
    finally:
        if dist is not None:
            dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])

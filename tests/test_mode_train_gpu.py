"""GPU: `srmi.mode.train(args)` (P/mode.py:70-197) on synthetic images.

1. The loop's parameters, BatchNorm buffers and losses are BIT-EQUAL to a restatement of the reference loop body that
   keeps the reference's wasted work (generator output not detached in the D step, discriminator weight gradients
   collected in the G step, the generator run a SECOND time on the same batch for its own step): the three shortcuts
   train() takes - one generator forward per iteration, with BatchNorm's running statistics moved twice - change no number.
2. Cadence: metrics CSV rows every 10 epochs (epoch,psnr,ssim,lpips,dists; NaN for the last two), checkpoints every
   100 pre-train epochs under the reference's file names, loadable into a fresh Generator; the ragged last batch
   (5 images, batch 2) trains."""
import argparse
import csv
import math
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _images(n=5, lr=12, scale=4, seed=0):
    r = np.random.default_rng(seed)
    gts = [r.integers(0, 256, (lr * scale, lr * scale, 3), dtype=np.uint8) for _ in range(n)]
    lrs = [np.ascontiguousarray(g[::scale, ::scale]) for g in gts]
    return lrs, gts


def _args(**kw):
    from srmi.main import build_parser
    a = build_parser().parse_args([])
    a.res_num, a.batch_size, a.patch_size, a.pre_train_epoch, a.fine_train_epoch = 1, 2, 8, 2, 2
    a.LR_path = a.GT_path = None
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _vgg(seed=3):
    from srmi.vgg19 import vgg19
    net = vgg19(pre_trained=False)
    g = torch.Generator().manual_seed(seed)
    sd = {k: (torch.randn(v.shape, generator=g) * (2.0 / (v.shape[1] * 9)) ** 0.5 if v.dim() == 4 else torch.zeros(v.shape))
          for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    return net


def _seed(s=11):
    random.seed(s)
    torch.manual_seed(s)


def _reference_loop(args, images, vgg_net):
    """P/mode.py:75-179 restated with the reference's wasted work kept (labels sized from the batch, as train() does)."""
    import torch.optim.lr_scheduler as lr_scheduler
    from srmi import losses
    from srmi import optim          # the optimizer train() uses (agrees with torch.optim.Adam to an ulp per step:
    #                                 tests/test_train_ops_gpu.py::test_flat_adam_tracks_torch_adam)
    from srmi.dataset import Compose, augmentation, crop, mydata
    from srmi.srgan_model import Discriminator, Generator
    dataset = mydata(in_memory=args.in_memory, transform=Compose([crop(args.scale, args.patch_size), augmentation()]),
                     device=DEV, images=images)
    loader = dataset.loader(batch_size=args.batch_size, shuffle=True)
    generator = Generator(img_feat=3, n_feats=64, kernel_size=3, num_block=args.res_num, scale=args.scale).to(DEV)
    generator.train()
    l2_loss = losses.MSELoss()
    g_optim = optim.Adam(generator.parameters(), lr=1e-4)
    for _ in range(args.pre_train_epoch):
        for tr_data in loader:
            gt, lr = tr_data['GT'], tr_data['LR']
            output, _ = generator(lr)
            loss = l2_loss(gt, output)
            g_optim.zero_grad()
            loss.backward()
            g_optim.step()
    vgg_net = vgg_net.to(DEV).eval()
    discriminator = Discriminator(patch_size=args.patch_size * args.scale).to(DEV)
    discriminator.train()
    d_optim = optim.Adam(discriminator.parameters(), lr=1e-4)
    scheduler = lr_scheduler.StepLR(g_optim, step_size=2000, gamma=0.1)
    VGG_loss, cross_ent, tv_loss = losses.perceptual_loss(vgg_net), losses.BCELoss(), losses.TVLoss()
    for _ in range(args.fine_train_epoch):
        scheduler.step()
        for tr_data in loader:
            gt, lr = tr_data['GT'], tr_data['LR']
            real_label = torch.ones((lr.shape[0], 1), device=DEV)
            fake_label = torch.zeros((lr.shape[0], 1), device=DEV)
            output, _ = generator(lr)
            fake_prob = discriminator(output)
            real_prob = discriminator(gt)
            d_loss = cross_ent(real_prob, real_label) + cross_ent(fake_prob, fake_label)
            g_optim.zero_grad(); d_optim.zero_grad()
            d_loss.backward()
            d_optim.step()
            output, _ = generator(lr)
            fake_prob = discriminator(output)
            _percep_loss, hr_feat, sr_feat = VGG_loss((gt + 1.0) / 2.0, (output + 1.0) / 2.0, layer=args.feat_layer)
            L2_loss = l2_loss(output, gt)
            g_loss = args.vgg_rescale_coeff * _percep_loss + args.adv_coeff * cross_ent(fake_prob, real_label) + \
                args.tv_loss_coeff * tv_loss(args.vgg_rescale_coeff * (hr_feat - sr_feat) ** 2) + L2_loss
            g_optim.zero_grad(); d_optim.zero_grad()
            g_loss.backward()
            g_optim.step()
    return generator, discriminator, float(g_loss.detach()), float(d_loss.detach())


def test_train_equals_the_reference_loop_bit_for_bit(tmp_path):
    from srmi import mode
    from srmi.srgan_model import Generator
    alpha = Generator.__init__.__defaults__[4].weight            # the process-wide shared PReLU (P/srgan_model.py:8)
    args = _args(model_dir=str(tmp_path / "model"), metrics_csv=str(tmp_path / "metrics.csv"), tv_loss_coeff=0.5)
    images = _images()
    with torch.no_grad():
        alpha.fill_(0.25)
    _seed()
    res = mode.train(args, images=images, vgg_net=_vgg(), log=lambda *a: None)
    got_g = {k: v.detach().cpu().clone() for k, v in res["generator"].state_dict().items()}
    got_d = {k: v.detach().cpu().clone() for k, v in res["discriminator"].state_dict().items()}
    with torch.no_grad():
        alpha.fill_(0.25)
    _seed()
    gen, disc, g_loss, d_loss = _reference_loop(args, images, _vgg())
    for k, v in gen.state_dict().items():
        assert torch.equal(got_g[k], v.cpu()), k
    for k, v in disc.state_dict().items():
        assert torch.equal(got_d[k], v.cpu()), k
    assert res["g_loss"] == g_loss and res["d_loss"] == d_loss
    # 2 + 2 epochs of 3 iterations; the generator runs forward twice per fine-tune iteration (P/mode.py:149,164)
    assert int(got_g["conv02.body.1.num_batches_tracked"]) == 2 * 3 + 2 * 3 * 2
    with torch.no_grad():
        alpha.fill_(0.25)


def test_train_cadence_files_and_ragged_batch(tmp_path):
    from srmi import mode
    from srmi.srgan_model import Generator
    args = _args(pre_train_epoch=100, fine_train_epoch=10, model_dir=str(tmp_path / "model"),
                 metrics_csv=str(tmp_path / "metrics.csv"))
    _seed(5)
    lines = []
    res = mode.train(args, images=_images(), vgg_net=_vgg(), log=lambda *a: lines.append(a[0]))
    rows = list(csv.DictReader(open(args.metrics_csv)))
    assert list(rows[0].keys()) == ["epoch", "psnr", "ssim", "lpips", "dists"]
    assert [int(r["epoch"]) for r in rows] == list(range(10, 101, 10)) + [10]
    assert all(math.isfinite(float(r["psnr"])) and math.isfinite(float(r["ssim"])) for r in rows)
    assert all(math.isnan(float(r["lpips"])) and math.isnan(float(r["dists"])) for r in rows)
    assert os.listdir(args.model_dir) == ["pre_trained_model_100.pt"]          # fine-tune checkpoints come every 500 epochs
    g = Generator(num_block=1)
    g.load_state_dict(torch.load(os.path.join(args.model_dir, "pre_trained_model_100.pt"), weights_only=True))
    # print cadence: every 2nd epoch, (epoch, loss, rule) in pre-training and (epoch, g_loss, d_loss, rule) afterwards
    assert lines.count('=========') == 50 + 5 and lines[0] == 2
    assert math.isfinite(res["pre_loss"]) and math.isfinite(res["g_loss"]) and math.isfinite(res["d_loss"])
    with torch.no_grad():
        Generator.__init__.__defaults__[4].weight.fill_(0.25)

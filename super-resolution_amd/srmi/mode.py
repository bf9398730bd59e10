"""`test_only(args)` and `test(args)` with the reference's names, arguments and output files (P/mode.py:335-401),
running the generator and the pre/post-processing on the MI355X.

Differences from the reference, all on the host side of the boundary: images cross PCIe as uint8 and are converted
on the device (srmi.imageio); the Y-PSNR of `test` is computed on the device (srmi.metrics); results are identical
(uint8 images bit-exact given the same generator output).

`train(args)` (P/mode.py:70-197) is the reference's two-phase loop in the reference's order - pre-training on the L2
loss (:96-122), then the SRGAN fine-tune iterations (:141-197) - on the HIP training kernels, with the device data
pipeline of srmi.dataset.  What it keeps, line by line: `scheduler.step()` at the START of every fine-tune epoch and
on the generator's optimizer only (:133,:142); the generator runs forward TWICE per iteration, so train-mode
BatchNorm running statistics take two momentum updates (:149,:164); print cadence (:110-113,:183-187); metrics
every 10 epochs on the first image of the last batch with `scale=1` (:116-119,:190-193) into a CSV with the
columns epoch,psnr,ssim,lpips,dists (lpips / dists NaN: third-party networks); checkpoints every 100 pre-train
epochs and every 500 fine-tune epochs under the reference's file names (:121-122,:195-197).  What it changes, none
of it visible in any parameter, buffer or loss value:
  * the generator output is detached for the D step (the reference back-propagates d_loss into G and throws those
    gradients away at the next `g_optim.zero_grad()`, :149-160,:176), and D's parameters do not collect gradients in
    the G step (thrown away at :159 of the next iteration): same numbers, a third less backward work;
  * `real_label` / `fake_label` are sized from the ACTUAL batch: the reference builds them once with
    `args.batch_size` rows (:138-139) and has no `drop_last`, so a ragged last batch (93 images = 5 x 16 + 13 in the
    bundled dataset) makes its BCELoss raise; here the last batch trains like the others;
  * `metrics_df.append` (removed in pandas 2) is replaced by rewriting the CSV from a list of rows;
  * `optim.Adam` is srmi.optim.Adam: the same update (torch's arithmetic order, agreeing to an ulp per step) as one
    launch per model over flat parameter / gradient / moment buffers instead of a dozen multi-tensor launches;
  * with torch.distributed initialised (one process per GPU) every rank draws its own batches and the two
    gradient exchanges of SURVEY 8e2 run inside `GradientAverager.sync()` blocks; rank 0 writes files.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from PIL import Image

from . import imageio, metrics
from .srgan_model import Discriminator, Generator


def _load_generator(args, device):
    generator = Generator(img_feat=3, n_feats=64, kernel_size=3, num_block=args.res_num)
    generator.load_state_dict(torch.load(args.generator_path, map_location="cpu", weights_only=True))
    generator = generator.to(device)
    generator.eval()
    if getattr(args, "compute", None):
        generator.set_compute(args.compute)
    return generator


def _read_rgb(path, convert):
    img = Image.open(path)
    if convert:
        img = img.convert("RGB")
    return np.array(img)


def test_only(args, out_dir="./result"):
    """P/mode.py:382-401: every image of args.LR_path -> out_dir/res_%04d.png."""
    device = torch.device("cuda")
    generator = _load_generator(args, device)
    os.makedirs(out_dir, exist_ok=True)
    names = sorted(os.listdir(args.LR_path))
    with torch.no_grad():
        for i, name in enumerate(names):
            lr = torch.from_numpy(_read_rgb(os.path.join(args.LR_path, name), False)).to(device)
            output, _ = generator(imageio.from_uint8(lr))
            result = imageio.to_uint8(output, clip=False)[0].cpu().numpy()
            Image.fromarray(result).save(os.path.join(out_dir, "res_%04d.png" % i))
    return len(names)


def test(args, out_dir="./result", result_txt="./result.txt"):
    """P/mode.py:335-379: SR every LR image, Y-PSNR against GT cropped to the SR size, result.txt + PNGs."""
    device = torch.device("cuda")
    generator = _load_generator(args, device)
    os.makedirs(out_dir, exist_ok=True)
    lr_names, gt_names = sorted(os.listdir(args.LR_path)), sorted(os.listdir(args.GT_path))
    psnr_list = []
    with open(result_txt, "w") as f, torch.no_grad():
        for i, (ln, gn) in enumerate(zip(lr_names, gt_names)):
            lr = imageio.from_uint8(torch.from_numpy(_read_rgb(os.path.join(args.LR_path, ln), True)).to(device))
            gt = imageio.from_uint8(torch.from_numpy(_read_rgb(os.path.join(args.GT_path, gn), True)).to(device))
            _, _, h, w = lr.shape
            gt = gt[:, :, :h * args.scale, :w * args.scale]
            output, _ = generator(lr)
            psnr = metrics.test_psnr(gt, output, args.scale)
            psnr_list.append(psnr)
            f.write("psnr : %04f \n" % psnr)
            Image.fromarray(imageio.to_uint8(output, clip=True)[0].cpu().numpy()).save(
                os.path.join(out_dir, "res_%04d.png" % i))
        f.write("avg psnr : %04f" % np.mean(psnr_list))
    return psnr_list


class _no_param_grads:
    """Parameters of `module` stop collecting gradients inside the block (the data gradient still flows through)."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)
        return False


def _write_metrics(rows, path):
    import csv
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["epoch", "psnr", "ssim", "lpips", "dists"])
        for r in rows:
            w.writerow(["%d" % r[0]] + [repr(float(v)) for v in r[1:]])


def train(args, images=None, vgg_net=None, log=print):
    """P/mode.py:70-197.  `images=(lr_list, gt_list)` (uint8 HWC arrays) replaces the two folders; `vgg_net` replaces
    `vgg19()` (whose ImageNet weights cannot be downloaded here: set SRMI_VGG19_WEIGHTS, or pass a module).
    Returns {"generator", "discriminator", "metrics", "pre_loss", "g_loss", "d_loss"}."""
    import torch.distributed as dist
    import torch.optim.lr_scheduler as lr_scheduler

    from . import autograd as AG
    from . import losses, parallel
    from . import optim          # Adam as ONE launch per model over flat buffers (torch.optim.Adam's update to an ulp)
    from .dataset import Compose, augmentation, crop, mydata
    from .vgg19 import vgg19

    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    device = torch.device("cuda", torch.cuda.current_device())
    model_dir = getattr(args, "model_dir", "./model")
    metrics_csv = getattr(args, "metrics_csv", "./metrics.csv")
    metrics_rows = []

    transform = Compose([crop(args.scale, args.patch_size), augmentation()])
    dataset = mydata(GT_path=args.GT_path, LR_path=args.LR_path, in_memory=args.in_memory, transform=transform,
                     device=device, images=images)
    loader = dataset.loader(batch_size=args.batch_size, shuffle=True)

    generator = Generator(img_feat=3, n_feats=64, kernel_size=3, num_block=args.res_num, scale=args.scale)
    if args.fine_tuning:
        generator.load_state_dict(torch.load(args.generator_path, map_location="cpu", weights_only=True))
        log("pre-trained model is loaded")
        log("path: %s" % (args.generator_path))
    generator = generator.to(device)
    generator.train()
    parallel.broadcast_parameters(generator)

    l2_loss = losses.MSELoss()
    g_optim = optim.Adam(generator.parameters(), lr=1e-4)
    g_avg = parallel.GradientAverager(generator.parameters())

    def save(state_dict, name):
        if rank == 0:
            os.makedirs(model_dir, exist_ok=True)
            torch.save(state_dict, os.path.join(model_dir, name))

    def log_metrics(epoch, gt, output):
        if rank == 0:
            psnr, ssim, lpips_score, dists_score = metrics.calculate_metrics(gt, output, device)
            metrics_rows.append((epoch, psnr, ssim, lpips_score, dists_score))
            _write_metrics(metrics_rows, metrics_csv)

    pre_epoch = 0
    fine_epoch = 0
    loss = g_loss = d_loss = None

    # ---- Train using L2_loss (P/mode.py:96-122) ----
    while pre_epoch < args.pre_train_epoch:
        for tr_data in loader:
            gt = tr_data['GT']
            lr = tr_data['LR']

            output, _ = generator(lr)
            loss = l2_loss(gt, output)

            g_optim.zero_grad()
            with g_avg.sync():
                loss.backward()
            g_optim.step()

        pre_epoch += 1

        if pre_epoch % 2 == 0:
            log(pre_epoch)
            log(loss.item())
            log('=========')

        if pre_epoch % 10 == 0:
            log_metrics(pre_epoch, gt, output)

        if pre_epoch % 100 == 0:
            save(generator.state_dict(), 'pre_trained_model_%03d.pt' % pre_epoch)

    result = {"generator": generator, "discriminator": None, "metrics": metrics_rows,
              "pre_loss": None if loss is None else float(loss.detach()), "g_loss": None, "d_loss": None}
    if args.fine_train_epoch <= 0:
        return result

    # ---- Fine-tuning with perceptual & adversarial loss (P/mode.py:124-197) ----
    if vgg_net is None:
        vgg_net = vgg19()
    vgg_net = vgg_net.to(device).eval()

    discriminator = Discriminator(patch_size=args.patch_size * args.scale)
    discriminator = discriminator.to(device)
    discriminator.train()
    parallel.broadcast_parameters(discriminator)

    d_optim = optim.Adam(discriminator.parameters(), lr=1e-4)
    d_avg = parallel.GradientAverager(discriminator.parameters())
    scheduler = lr_scheduler.StepLR(g_optim, step_size=2000, gamma=0.1)

    VGG_loss = losses.perceptual_loss(vgg_net)
    cross_ent = losses.BCELoss()
    tv_loss = losses.TVLoss()
    labels = {}

    def label(value, n):
        if (value, n) not in labels:
            labels[(value, n)] = torch.full((n, 1), float(value), device=device)
        return labels[(value, n)]

    import warnings
    while fine_epoch < args.fine_train_epoch:
        with warnings.catch_warnings():       # the reference's order (scheduler before any optimizer.step) is deliberate
            warnings.simplefilter("ignore")
            scheduler.step()

        for tr_data in loader:
            gt = tr_data['GT']
            lr = tr_data['LR']
            real_label, fake_label = label(1, lr.shape[0]), label(0, lr.shape[0])

            # The reference runs the generator twice per iteration on the same LR batch (P/mode.py:149 and :164) with no
            # generator update in between: the second forward's outputs are the first's bit for bit.  ONE forward serves
            # both steps - detached for the discriminator's, with its graph for the generator's own - and BatchNorm's
            # running statistics move twice, as two forwards leave them (autograd.bn_running_updates).
            with AG.bn_running_updates(2):
                output, _ = generator(lr)

            # Training Discriminator
            fake_prob = discriminator(output.detach())
            real_prob = discriminator(gt)

            d_loss_real = cross_ent(real_prob, real_label)
            d_loss_fake = cross_ent(fake_prob, fake_label)
            d_loss = d_loss_real + d_loss_fake

            g_optim.zero_grad()
            d_optim.zero_grad()
            with d_avg.sync():
                d_loss.backward()
            d_optim.step()

            # Training Generator
            with _no_param_grads(discriminator):
                fake_prob = discriminator(output)

                _percep_loss, hr_feat, sr_feat = VGG_loss((gt + 1.0) / 2.0, (output + 1.0) / 2.0, layer=args.feat_layer)

                L2_loss = l2_loss(output, gt)
                percep_loss = args.vgg_rescale_coeff * _percep_loss
                adversarial_loss = args.adv_coeff * cross_ent(fake_prob, real_label)
                total_variance_loss = args.tv_loss_coeff * tv_loss(args.vgg_rescale_coeff * (hr_feat - sr_feat) ** 2)

                g_loss = percep_loss + adversarial_loss + total_variance_loss + L2_loss

                g_optim.zero_grad()
                d_optim.zero_grad()
                with g_avg.sync():
                    g_loss.backward()
            g_optim.step()

        fine_epoch += 1

        if fine_epoch % 2 == 0:
            log(fine_epoch)
            log(g_loss.item())
            log(d_loss.item())
            log('=========')

        if fine_epoch % 10 == 0:
            log_metrics(fine_epoch, gt, output)

        if fine_epoch % 500 == 0:
            save(generator.state_dict(), 'SRGAN_gene_%03d.pt' % fine_epoch)
            save(discriminator.state_dict(), 'SRGAN_discrim_%03d.pt' % fine_epoch)

    result.update(discriminator=discriminator, g_loss=None if g_loss is None else float(g_loss.detach()),
                  d_loss=None if d_loss is None else float(d_loss.detach()))
    return result

"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the training path: one fused op and the reference's
training-step arithmetic restated with plain torch CPU ops + torch autograd.

Each function follows the reference call sites it names; the whole-network forwards live in
oracle/srgan_oracle.py (pinned bit-exactly to the imported reference, see oracle/make_golden.py), and
this file only adds the backward by running torch autograd over those same ops - which is exactly what
the reference does (P/mode.py:105, :160, :178).
"""
import torch
import torch.nn.functional as F


def fused_conv(x, w, b, gamma=None, beta=None, alpha=None, residual=None, stride=1, pad=None, act="none",
               slope=0.01, shuffle=1, dcr=False, running=None, eps=1e-5, momentum=0.1):
    """conv (P/ops.py:6-15) -> train-mode BatchNorm2d (P/ops.py:24-26) -> activation (P/ops.py:28-29) ->
    PixelShuffle (P/ops.py:75; DCR = tf.nn.depth_to_space) -> `res += x` (P/ops.py:47).  In the Upsampler
    the activation follows the shuffle (P/ops.py:77-78); it is applied here in that order."""
    pad = w.shape[2] // 2 if pad is None else pad
    z = F.conv2d(x, w, b, stride=stride, padding=pad)
    if gamma is not None:
        rm, rv = running if running is not None else (None, None)
        z = F.batch_norm(z, rm, rv, gamma, beta, training=True, momentum=momentum, eps=eps)
    if shuffle > 1:
        if dcr:
            N, C, H, W = z.shape
            r = shuffle
            z = z.view(N, r, r, C // (r * r), H, W).permute(0, 3, 4, 1, 5, 2).reshape(N, C // (r * r), H * r, W * r)
        else:
            z = F.pixel_shuffle(z, shuffle)
    if act == "prelu":
        z = F.prelu(z, alpha)
    elif act == "lrelu":
        z = F.leaky_relu(z, slope)
    elif act == "relu":
        z = F.relu(z)
    elif act == "tanh":
        z = torch.tanh(z)
    elif act == "sigmoid":
        z = torch.sigmoid(z)
    if residual is not None:
        z = z + residual
    return z


def adam_step(params, grads, state, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam (defaults) restated on lists of tensors; state = dict(step, m, v)."""
    state["step"] += 1
    t = state["step"]
    b1, b2 = betas
    for i, (p, g) in enumerate(zip(params, grads)):
        state["m"][i] = state["m"][i] + (g - state["m"][i]) * (1 - b1)
        state["v"][i] = state["v"][i] * b2 + (1 - b2) * g * g
        denom = state["v"][i].sqrt() / (1 - b2 ** t) ** 0.5 + eps
        p -= (lr / (1 - b1 ** t)) * state["m"][i] / denom


def srgan_losses(gsd, dsd, vsd, lr, gt, num_block, gstats, dstats, vgg_rescale_coeff=0.006, adv_coeff=1e-3,
                 tv_loss_coeff=0.0, feat_layer="relu5_4"):
    """The loss arithmetic of one fine-tune iteration (P/mode.py:148-174) as two closures over parameter
    dicts, so that the caller can interleave the optimizer steps exactly as the reference does:
        d_loss = d_part()      # :149-156, G fwd #1, D(fake), D(real)
        g_loss = g_part()      # :164-174, G fwd #2, D(fake) with the UPDATED D, VGG, L2, TV
    BN running statistics are threaded through gstats / dstats (train-mode BN, every forward updates them)."""
    from oracle import srgan_oracle as O

    def d_part():
        output, _ = O.generator_forward(gsd, lr, num_block=num_block, training=True, stats_out=gstats)
        fake_prob = O.discriminator_forward(dsd, output, training=True, stats_out=dstats)
        real_prob = O.discriminator_forward(dsd, gt, training=True, stats_out=dstats)
        d_loss_real = F.binary_cross_entropy(real_prob, torch.ones_like(real_prob))
        d_loss_fake = F.binary_cross_entropy(fake_prob, torch.zeros_like(fake_prob))
        return d_loss_real + d_loss_fake

    def g_part():
        output, _ = O.generator_forward(gsd, lr, num_block=num_block, training=True, stats_out=gstats)
        fake_prob = O.discriminator_forward(dsd, output, training=True, stats_out=dstats)
        _percep, hr_feat, sr_feat = O.perceptual_loss(vsd, (gt + 1.0) / 2.0, (output + 1.0) / 2.0, feat_layer)
        L2 = F.mse_loss(output, gt)
        percep = vgg_rescale_coeff * _percep
        adv = adv_coeff * F.binary_cross_entropy(fake_prob, torch.ones_like(fake_prob))
        tv = tv_loss_coeff * O.tv_loss(vgg_rescale_coeff * (hr_feat - sr_feat) ** 2)
        return percep + adv + tv + L2, dict(percep=percep, adv=adv, tv=tv, l2=L2)

    return d_part, g_part

"""Throughput of the device data pipeline on the reference's training shape: 93 in-memory images,
batch 16, LR patch 24, scale 4 (P/main.py defaults), against the same work done the reference's way (numpy per
sample, P/dataset.py:27-49 + :81-132, restated inline for timing only)."""
import importlib, os, random, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = importlib.import_module("super-resolution_amd.srmi.dataset")

g = np.random.default_rng(0)
N, S, P, B = 93, 4, 24, 16
lr = [g.integers(0, 256, (96, 128, 3), dtype=np.uint8) for _ in range(N)]
gt = [g.integers(0, 256, (96 * S, 128 * S, 3), dtype=np.uint8) for _ in range(N)]
ds = D.mydata(in_memory=True, transform=D.Compose([D.crop(S, P), D.augmentation()]), images=(lr, gt))
random.seed(0); torch.manual_seed(0)
for _ in range(3):
    for b in ds.loader(B, shuffle=True): pass
torch.cuda.synchronize()
t0 = time.perf_counter(); nb = 0
for _ in range(20):
    for b in ds.loader(B, shuffle=True): nb += 1
torch.cuda.synchronize()
dev = (time.perf_counter() - t0) / nb


def ref_item(i):        # the reference's per-sample numpy path (timing baseline)
    GT, LR = gt[i].astype(np.float32) / 127.5 - 1.0, lr[i].astype(np.float32) / 127.5 - 1.0
    ih, iw = LR.shape[:2]
    ix, iy = random.randrange(0, iw - P + 1), random.randrange(0, ih - P + 1)
    LRp, GTp = LR[iy:iy + P, ix:ix + P], GT[iy * S:(iy + P) * S, ix * S:(ix + P) * S]
    if random.randrange(0, 2): LRp, GTp = np.fliplr(LRp).copy(), np.fliplr(GTp).copy()
    if random.randrange(0, 2): LRp, GTp = np.flipud(LRp).copy(), np.flipud(GTp).copy()
    if random.randrange(0, 2): LRp, GTp = LRp.transpose(1, 0, 2), GTp.transpose(1, 0, 2)
    return LRp.transpose(2, 0, 1).astype(np.float32), GTp.transpose(2, 0, 1).astype(np.float32)


t0 = time.perf_counter(); nb2 = 0
for _ in range(3):
    for s in range(0, N, B):
        items = [ref_item(i) for i in range(s, min(N, s + B))]
        l = torch.from_numpy(np.stack([a for a, _ in items])).cuda(); h = torch.from_numpy(np.stack([b_ for _, b_ in items])).cuda()
        nb2 += 1
torch.cuda.synchronize()
cpu = (time.perf_counter() - t0) / nb2
print("device pipeline: %.0f us per batch of %d (host draws + descriptor upload + one launch); numpy per-sample path + upload: %.0f us per batch  (x%.1f)" % (dev * 1e6, B, cpu * 1e6, cpu / dev))

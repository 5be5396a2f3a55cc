import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import cgs_amd
from cgs_amd import hourglass as hg, spec
from oracle import hourglass_ref as orc
F = torch.nn.functional
raw = dict(np.load('/root/repo/tests/golden/g1_weights_chfak1.npz'))
pc = {k.split('/',1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith('critic/')}
dev = torch.device('cuda:0'); lc = spec.critic_layout(); fc = torch.empty(lc.total, device=dev)
lc.flatten({k: v.to(dev) for k, v in pc.items()}, fc)
n = 21; rs = np.random.RandomState(n)
x_u8 = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
xf = (torch.from_numpy(x_u8).to(dev).float() / 255.0).contiguous()
c = hg.critic_forward(fc, lc, xf, n)
X = orc.u8_to_nchw(x_u8).double()
h = X
for i, key in enumerate(["features.0", "features.3"]):
    pre = torch.relu(F.conv2d(h, pc[key + ".weight"].double(), pc[key + ".bias"].double(), padding=1))
    pooled, idx = F.max_pool2d(pre, 2, return_indices=True)
    hw = pre.shape[-1]
    win = pre.unfold(2, 2, 2).unfold(3, 2, 2).reshape(n, pre.shape[1], hw // 2, hw // 2, 4)
    top2 = win.topk(2, dim=-1).values
    gap = ((top2[..., 0] - top2[..., 1]) / top2[..., 0].clamp_min(1e-30)).numpy()
    am = c[f"am{i}"].cpu().numpy().astype(np.uint32)
    yy, xx = np.meshgrid(np.arange(hw // 2), np.arange(hw // 2), indexing="ij")
    bad = []
    for ch in range(8):
        nib = (am[..., 0] >> (4 * ch)) & 15
        ii = idx[:, ch].numpy(); ref = ((ii // hw) - 2 * yy) * 2 + ((ii % hw) - 2 * xx)
        pos = pooled[:, ch].numpy() > 0
        d = pos & (nib != ref)
        for w in np.argwhere(d):
            bad.append(gap[w[0], ch, w[1], w[2]])
    got = c[f"e{i}"].cpu().permute(0, 3, 1, 2).double().numpy()
    print(key, "argmax mismatches:", len(bad), "their relative top-2 gaps:", np.sort(bad)[:10], " max|e - ref|/max:", np.abs(got - pooled.numpy()).max() / pooled.numpy().max())
    h = pooled

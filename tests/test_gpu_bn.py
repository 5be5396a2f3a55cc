"""The OPTIONAL BatchNorm2d + activation epilogue (csrc/bn.hip, nets.BatchNormAct2d; SURVEY section 8 row f4, the north_star's BN wording).
The reference has no BatchNorm: nothing pins this op to it -- the checks are against torch.nn.BatchNorm2d (+ activation) and its
autograd in float64 on the CPU, training and eval mode, incl. the running statistics."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2)}


@pytest.mark.parametrize("shape", [(5, 8, 16, 16), (3, 16, 33, 7), (2, 40, 32, 32), (512, 8, 32, 32), (9, 64, 5, 5), (64, 12, 64, 64)])
@pytest.mark.parametrize("act", ["none", "relu", "lrelu"])
def test_batchnorm_act_training_vs_torch_float64(shape, act):
    """Forward, running statistics, and dx / dgamma / dbeta of one training-mode call; a large mean offset (100 x the spread) on half the
    channels exercises the statistics' numerics (pairwise merges of (count, mean, M2), not E[x^2] - E[x]^2)."""
    from cgs_amd import nets
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c * 100 + h)
    x = torch.randn(shape, generator=g)
    x[:, ::2] += 100.0
    gw, gb = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    gy = torch.randn(shape, generator=g)
    ref = torch.nn.BatchNorm2d(c, eps=1e-5, momentum=0.1).double()
    with torch.no_grad():
        ref.weight.copy_(gw); ref.bias.copy_(gb)
    xr = x.double().requires_grad_(True)
    yr = ACTS[act](ref(xr))
    yr.backward(gy.double())
    m = nets.BatchNormAct2d(c, act=act, slope=0.2).cuda()
    with torch.no_grad():
        m.weight.copy_(gw); m.bias.copy_(gb)
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    rel = lambda a, b: (a.double().cpu() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    assert rel(y, yr.detach()) < 2e-5          # (the +100 offset costs fp32 input resolution: 100 * 2^-24 / spread 1)
    assert rel(m.running_mean, ref.running_mean) < 1e-6 and rel(m.running_var, ref.running_var) < 2e-5
    assert int(m.num_batches_tracked) == 1
    if act == "none":
        want_dx, want_dg, want_db = xr.grad, ref.weight.grad, ref.bias.grad
    else:
        # an element whose pre-activation is within fp32 rounding of zero may take the other branch of the activation than in float64 (a
        # handful of the 4 M elements of the largest case): the float64 reference backward is therefore written out with the KERNEL's own
        # branch decisions (sign of its y), which checks the BatchNorm backward itself; the branches must agree on all but 1e-4 of the elements
        ypos = (y.detach().cpu() > 0)
        assert (ypos != (yr.detach() > 0)).double().mean().item() < 1e-4
        dyp = gy.double() * torch.where(ypos, torch.ones((), dtype=torch.float64), torch.full((), 0.2 if act == "lrelu" else 0.0, dtype=torch.float64))
        xd = x.double()
        mu, var = xd.mean((0, 2, 3), keepdim=True), xd.var((0, 2, 3), unbiased=False, keepdim=True)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        xh, N = (xd - mu) * rstd, n * h * w
        want_db, want_dg = dyp.sum((0, 2, 3)), (dyp * xh).sum((0, 2, 3))
        want_dx = gw.double().view(1, c, 1, 1) * rstd * (dyp - want_db.view(1, c, 1, 1) / N - xh * want_dg.view(1, c, 1, 1) / N)
    assert rel(xg.grad, want_dx) < 5e-4        # dx is a difference of O(1) terms that nearly cancel; relative to its own maximum
    assert rel(m.weight.grad, want_dg) < 2e-4 and rel(m.bias.grad, want_db) < 2e-5


def test_batchnorm_act_eval_mode_and_determinism():
    """Eval mode normalises with the running statistics (forward and dx), does not touch them; two training calls on the same input give the
    same bits (fixed-order reductions, no atomics)."""
    from cgs_amd import nets
    g = torch.Generator().manual_seed(3)
    x = torch.randn((7, 16, 24, 24), generator=g) * 2 + 1
    m = nets.BatchNormAct2d(16, act="relu").cuda()
    ref = torch.nn.BatchNorm2d(16).double()
    for _ in range(3):
        m(x.cuda()); ref(x.double())
    y1 = m(x.cuda()).clone()
    m.running_mean.copy_(ref.running_mean.float()); m.running_var.copy_(ref.running_var.float())
    m.eval(); ref.eval()
    rm = m.running_mean.clone()
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    y.sum().backward()
    xr = x.double().requires_grad_(True)
    yr = F.relu(ref(xr))
    yr.sum().backward()
    assert torch.equal(rm, m.running_mean)
    assert (y.double().cpu() - yr.detach()).abs().max().item() < 1e-5
    assert (xg.grad.double().cpu() - xr.grad).abs().max().item() < 1e-5
    m.train()
    m2 = nets.BatchNormAct2d(16, act="relu").cuda()
    a, b = m2(x.cuda()).clone(), nets.BatchNormAct2d(16, act="relu").cuda()(x.cuda())
    assert torch.equal(a, b)
    with pytest.raises(Exception):
        nets.BatchNormAct2d(10)


def test_non_fp32_input_is_rejected_loudly():
    """The kernels compute and return fp32: a half / bfloat16 input would silently change dtype downstream -- it is an error."""
    from cgs_amd import _lib, nets
    m = nets.BatchNormAct2d(8).cuda()
    for dt in (torch.float16, torch.bfloat16, torch.float64):
        with pytest.raises(_lib.CgsError):
            m(torch.zeros(2, 8, 4, 4, device="cuda", dtype=dt))

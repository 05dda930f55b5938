"""__graft_entry__.smoke(): one small hot-path invocation on cuda:0, checked against the oracle."""
import numpy as np
import torch


def run():
    if not torch.cuda.is_available():
        raise RuntimeError('smoke() needs a GPU')
    import inputs as gi
    from oracle import head as ohead
    from . import ops_head
    lg_np, _ = gi.planted_logits(2, 18, 64, seed=3)
    lg = torch.from_numpy(lg_np)
    gw = torch.randn(2, 3, 18, 3, generator=torch.Generator().manual_seed(0))
    lc = lg.clone().requires_grad_(True)
    ko, do, io = ohead.softargmax_multi(lc, 18, 3, 15)
    (ko * gw).sum().backward()
    lgpu = lg.cuda().requires_grad_(True)
    kg, dg, ig = ops_head.softargmax_multi(lgpu, 18, 3, 15)
    (kg * gw.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert np.array_equal(io.numpy(), ig.cpu().numpy()), 'depth-peak indices differ from the oracle'
    err = (kg.cpu() - ko).abs().max().item()
    gerr = (lgpu.grad.cpu() - lc.grad).abs().max().item()
    assert err < 1e-4 and gerr < 1e-6, (err, gerr)
    print('[smoke] head ok: max|dkps|=%.2e max|dgrad|=%.2e' % (err, gerr))
    from . import smoke_step
    smoke_step.run()

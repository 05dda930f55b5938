"""Helpers of the step-level self-checks (tests/test_gpu_repro.py, tests/test_gpu_parity_r3.py): build a full-size step,
run it from a restored state with the gradient arenas captured as Adam would consume them, poison the free memory of the
caching allocator, digest a result.  Test infrastructure - nothing here is on the product path."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'x-as-supervision_amd'), os.path.join(ROOT, 'tests', 'golden')):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def build_step(name, batch, seed=1234, data_seed=100, planted=True):
    """-> (step, batch dict): BASELINE config `name` at `batch` samples per GPU, seeded weights, dropout off.
    planted: the bias of the detector's final 1x1 conv carries three well-separated depth peaks per joint (what
    tests/golden/inputs.py::planted_logits does for the head fixtures), so the depth-peak indices and the min-over-
    hypotheses selections (modules/model.py:114,162) do not sit on near-ties of a flat random-init depth marginal."""
    import torch
    from xas_amd import engine
    from xas_amd.synthetic import model_config, synthetic_batch
    cfg = model_config(name)
    cams = cfg['model_params']['cam_id_list']
    x = synthetic_batch(batch, cams, torch.device('cuda'), seed=data_seed)
    torch.manual_seed(seed)
    model, disc, od, odisc = engine.prepare_model(cfg)
    if planted:
        plant_depth_peaks(model.regressor)
    model.cuda().train(), disc.cuda().train()
    disc.smpl_discriminator.header.p = 0.0
    return engine.TrainStep(cfg, model, disc, od, odisc), x


def plant_depth_peaks(regressor, K=18, D=64, seed=7):
    """Bias of the final 1x1 conv (deconv_head.py:34-35; channel = joint * D + depth, keypoint_detector_integral_multi.py:
    70-74): per joint three peaks of different height at depths >= 9 bins apart, on a sloped floor."""
    import numpy as np
    import torch
    rng = np.random.Generator(np.random.PCG64(seed))
    b = np.zeros((K, D), np.float32)
    for k in range(K):
        d0 = int(rng.integers(6, 16))
        peaks = [d0, d0 + int(rng.integers(12, 18)), d0 + int(rng.integers(28, 40))]
        for d, h in zip(peaks, (6.0, 5.0, 4.0)):
            b[k] += h * np.exp(-0.5 * ((np.arange(D) - d) / 1.5) ** 2)
        b[k] += 0.01 * np.arange(D) / D
    conv = regressor.net.head.features[9]
    assert conv.bias.shape[0] == K * D
    with torch.no_grad():
        conv.bias.copy_(torch.from_numpy(b.reshape(-1)))


def run_captured(step, x, snapshot=None, seed=4242):
    """One step (from `snapshot` when given) -> dict: gradient arenas as handed to Adam ('det', 'disc'), loss vector, every
    module buffer, parameters after the step, the int64 depth-peak indices of every detector call of the step."""
    import torch
    from xas_amd import ops_head, state
    if snapshot is not None:
        state.restore(step, snapshot)
    torch.manual_seed(seed)
    grads, peaks = {}, []
    step.grad_probe = lambda which, arena: grads.__setitem__(which, arena.clone())
    ops_head.peak_probe = peaks.append
    try:
        ld, lk, tot, out = step(x)
    finally:
        step.grad_probe = None
        ops_head.peak_probe = None
    torch.cuda.synchronize()
    res = dict(grads)
    res['loss'] = torch.stack([ld.detach().float().reshape(())] + [v.detach().float().mean().reshape(()) for v in lk.values()])
    res['loss_names'] = ['disc'] + list(lk.keys())
    res['bufs'] = torch.cat([b.detach().double().reshape(-1) for m in (step.model, step.disc) for b in m.buffers()])
    res['params'] = step.opt_det.param_arena.clone()
    res['peaks'] = torch.cat([p.reshape(-1) for p in peaks]) if peaks else torch.zeros(0, dtype=torch.int64)
    return res


TENSOR_KEYS = ('det', 'disc', 'loss', 'bufs', 'params', 'peaks')


def digest(res):
    """{key: sha256 of the raw bytes} - what a second process prints for the cross-process comparison."""
    return {k: hashlib.sha256(res[k].detach().cpu().contiguous().numpy().tobytes()).hexdigest() for k in TENSOR_KEYS if k in res}


def poison_free_memory(value, device='cuda'):
    """Fill every FREE block of the caching allocator with `value`: the allocator's own snapshot names its inactive blocks; take
    them back largest first (each request is served from the cache - nothing new is asked of the driver, so repeated calls do
    not grow the process's reservation), write, release.  A kernel that reads memory nobody wrote then reads `value`.  -> bytes."""
    import torch
    torch.cuda.synchronize()
    sizes = sorted((b['size'] for seg in torch.cuda.memory_snapshot() for b in seg['blocks'] if b['state'] == 'inactive'), reverse=True)
    held, n = [], 0
    for size in sizes:
        if size < 512:
            continue
        before = torch.cuda.memory_reserved()
        try:
            t = torch.empty(size // 4, device=device, dtype=torch.float32)
        except torch.OutOfMemoryError:
            continue
        if torch.cuda.memory_reserved() > before:          # (did not fit a cached block after all: give it straight back)
            del t
            continue
        t.fill_(value)
        held.append(t)
        n += t.numel() * 4
    del held
    torch.cuda.synchronize()
    return n


def main():
    """python tests/_stepcheck.py NAME BATCH -> one JSON line with the digests of one captured step (second-process leg)."""
    import json
    name, batch = sys.argv[1], int(sys.argv[2])
    step, x = build_step(name, batch)
    print('STEPCHECK ' + json.dumps(digest(run_captured(step, x))), flush=True)


if __name__ == '__main__':
    main()

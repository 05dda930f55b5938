"""Fused Adam over flat arenas (reference: two torch.optim.Adam(betas=(0.5, 0.999)) at train.py:257-264).

Parameters of one optimizer are re-homed into ONE contiguous fp32 arena (each nn.Parameter becomes a view),
gradients into a second arena (`.grad` views, so autograd accumulates in place), moments into two more.
`step()` is one kernel launch over the arena; `zero_grad()` is one memset.  The contiguous gradient arena
is also what the data-parallel reducer all-reduces in large buckets (xas_amd/dp.py).
state_dict()/load_state_dict() speak torch.optim.Adam's format so reference checkpoints resume.
"""
import torch

from . import ops_nn
from ._lib import call, ptr


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if len(self.param_groups) != 1:
            raise ValueError('FusedAdam updates one flat arena with one set of hyper-parameters: pass a single param group')
        self._flat = None
        self._steps = 0
        self._epoch = [0]            # bumped whenever this optimizer rewrites its parameters (ops_nn._PackCache key)

    # ---- arenas -------------------------------------------------------------------------------------
    def _build(self):
        ps = [p for g in self.param_groups for p in g['params']]
        if not ps:
            raise RuntimeError('FusedAdam: no parameters')
        dev = ps[0].device
        if dev.type != 'cuda':
            raise RuntimeError('FusedAdam runs on the GPU only (move the model first); no CPU fallback')
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4                  # 16-byte aligned slots
        arena = torch.zeros(n, device=dev, dtype=torch.float32)
        grads = torch.zeros(n, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(ps, offs):
                v = arena[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                g = grads[o:o + p.numel()].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g
        self._flat = dict(params=ps, offs=offs, n=n, p=arena, g=grads, m=torch.zeros_like(arena), v=torch.zeros_like(arena))
        for p in ps:
            p._xas_epoch = self._epoch
        self._epoch[0] += 1

    @property
    def grad_arena(self):
        if self._flat is None:
            self._build()
        return self._flat['g']

    @property
    def param_arena(self):
        if self._flat is None:
            self._build()
        return self._flat['p']

    def _check_views(self):
        f = self._flat
        for p, o in zip(f['params'], f['offs']):
            if p.data_ptr() != f['p'].data_ptr() + 4 * o:           # e.g. after module.to()/load onto new storage
                with torch.no_grad():
                    v = f['p'][o:o + p.numel()].view(p.shape)
                    v.copy_(p.data)
                    p.data = v
                self._epoch[0] += 1
            if p.grad is None:
                p.grad = f['g'][o:o + p.numel()].view(p.shape)
            elif p.grad.data_ptr() != f['g'].data_ptr() + 4 * o:
                with torch.no_grad():
                    g = f['g'][o:o + p.numel()].view(p.shape)
                    g.copy_(p.grad)
                    p.grad = g

    # ---- optimizer API ------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        if self._flat is None:
            self._build()
        ops_nn.join_side_stream()                    # weight gradients accumulated on the side stream
        self._check_views()
        f = self._flat
        g0 = self.param_groups[0]
        self._steps += 1
        b1, b2 = g0['betas']
        call('xas_adam_step', ptr(f['p']), ptr(f['g']), ptr(f['m']), ptr(f['v']), f['n'], float(g0['lr']), float(b1),
             float(b2), float(g0['eps']), self._steps)
        self._epoch[0] += 1          # packed weight copies of THIS optimizer's parameters are stale now

    def zero_grad(self, set_to_none=False):
        """One memset of the gradient arena.  `set_to_none` is ignored on purpose: .grad must stay a view of the arena
        (kernels accumulate into it by pointer)."""
        if self._flat is None:
            self._build()
        ops_nn.join_side_stream()
        self._check_views()
        self._flat['g'].zero_()

    def state_dict(self):
        """torch.optim.Adam layout: state[i] = {step, exp_avg, exp_avg_sq}."""
        if self._flat is None:
            self._build()
        f = self._flat
        state = {}
        for i, (p, o) in enumerate(zip(f['params'], f['offs'])):
            state[i] = {'step': torch.tensor(float(self._steps)),
                        'exp_avg': f['m'][o:o + p.numel()].view(p.shape).clone(),
                        'exp_avg_sq': f['v'][o:o + p.numel()].view(p.shape).clone()}
        groups = [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]
        idx = 0
        for g, pg in zip(groups, self.param_groups):
            g['params'] = list(range(idx, idx + len(pg['params'])))
            idx += len(pg['params'])
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        if self._flat is None:
            self._build()
        f = self._flat
        for g, sg in zip(self.param_groups, sd['param_groups']):
            for k in ('lr', 'betas', 'eps', 'initial_lr'):
                if k in sg:
                    g[k] = sg[k]
        steps = set()
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(f['params'], f['offs'])):
                st = sd['state'].get(i, sd['state'].get(str(i)))
                if st is None:
                    continue
                f['m'][o:o + p.numel()].view(p.shape).copy_(st['exp_avg'])
                f['v'][o:o + p.numel()].view(p.shape).copy_(st['exp_avg_sq'])
                steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('FusedAdam keeps ONE step counter for the arena; the checkpoint has per-parameter steps %s' % sorted(steps))
        if steps:
            self._steps = steps.pop()

"""Per-entry-point device timing with HIP events recorded on the stream the kernels are launched on
(torch's current stream), plus the algorithmic FLOP count of every convolution launch (for the soft-argmax head
entries the `flops` field carries the algorithmic HBM BYTES instead: logits read once forward, read + written backward)."""
import torch

from . import _lib

# (xas_conv_fwd_bnstats: the bracket also holds the ~10 us reduction of the per-tile partial sums that follows the convolution)
CONV_ENTRIES = ('xas_conv_fwd', 'xas_conv_fwd_head', 'xas_conv_fwd_bnstats', 'xas_conv_dgrad', 'xas_conv_dgrad_acc', 'xas_conv_dgrad_acc_masked', 'xas_conv_wgrad',
                'xas_conv_wgrad_oihw', 'xas_conv_wgrad_acc')


# batch-norm entry points (HBM bound): the `flops` field of their records carries the algorithmic HBM BYTES of the call
BN_ENTRIES = ('xas_bn_stats', 'xas_bn_apply', 'xas_bn_apply_amax', 'xas_bn_bwd_reduce', 'xas_bn_bwd_apply', 'xas_bn_bwd_apply_amax')


def bn_bytes(name, a):
    """Algorithmic bytes of a batch-norm call from its arguments (every activation-sized tensor it must read or write once;
    sign masks are one byte per float4)."""
    if name == 'xas_bn_stats':                      # (x, M, C, ...)
        return 4.0 * a[1] * a[2]
    if name in ('xas_bn_apply', 'xas_bn_apply_amax'):      # (x, mean, var, gamma, beta, residual, eps, act, M, C, groups, y, mask_out[, amax])
        t = 4.0 * a[8] * a[9]
        return t * (2 + (1 if a[5] else 0)) + (t / 16 if a[12] else 0.0)
    if name == 'xas_bn_bwd_reduce':                 # (x, y, dy, mean, var, gamma, beta, eps, act, M, C, groups, sums, ws, db, dg, mask)
        t = 4.0 * a[9] * a[10]
        return t * (1 + (1 if a[0] else 0) + (1 if a[1] else 0)) + (t / 16 if a[16] else 0.0)
    # xas_bn_bwd_apply(_amax): (x, y, dy, mean, var, gamma, beta, sums, eps, act, M, C, groups, count, dx, dres, mask[, amax])
    t = 4.0 * a[10] * a[11]
    return t * (2 + (1 if a[0] else 0) + (1 if a[1] else 0) + (1 if a[15] else 0)) + (t / 16 if a[16] else 0.0)


def conv_flops(shape):
    """2 * MACs of the convolution described by an xas_conv_shape (same count for fwd / dgrad / wgrad)."""
    return 2.0 * shape.N * shape.Ho * shape.Wo * shape.Cout * shape.R * shape.S * shape.Cin


def conv_bytes(shape):
    """Algorithmic HBM bytes of one conv launch: both activation tensors and the weights, once each."""
    return 4.0 * (shape.N * shape.Hi * shape.Wi * shape.Cin + shape.Cout * shape.R * shape.S * shape.Cin
                  + shape.N * shape.Ho * shape.Wo * shape.Cout)


class KernelTimer:
    def __init__(self, names=CONV_ENTRIES):
        self.names = set(names)
        self.records = []          # (name, start_event, end_event, flops, kernel class, shape signature)
        self.bytes_total = 0.0     # algorithmic bytes of the MFMA-path launches

    def __enter__(self):
        _lib.profiler = self
        return self

    def __exit__(self, *exc):
        _lib.profiler = None

    def timed_call(self, name, args):
        shape = next((a for a in args if isinstance(a, _lib.ConvShape)), None)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = _lib.fn(name)(*args, _lib.stream())
        b.record()
        if rc != 0:
            raise RuntimeError('%s failed (%d): %s' % (name, rc, _lib.load().xas_last_error().decode()))
        mfma = 0                   # kernel class of the launch: 0 no MFMA, 1 exact-fp32 MFMA, 2 bf16 MFMA, 3 bf16x6 MFMA, 4 f16x3 MFMA
        if shape is not None:
            kind = 0 if name in ('xas_conv_fwd', 'xas_conv_fwd_head', 'xas_conv_fwd_bnstats') else (
                1 if name in ('xas_conv_dgrad', 'xas_conv_dgrad_acc', 'xas_conv_dgrad_acc_masked', 'xas_conv_dgrad_bn_bwd') else 2)
            mfma = _lib.query('xas_conv_kernel_class', shape, kind)
        sig = None
        if shape is not None:
            sig = (shape.N, shape.Hi, shape.Wi, shape.Cin, shape.Cout, shape.R, shape.stride)
        work = conv_flops(shape) if shape is not None else 0.0
        if name == 'xas_head_softargmax_fwd':          # (logits, B, K, D, ...): the logits are read once
            B, K, D = args[1], args[2], args[3]
            work, sig = 4.0 * B * K * D * D * D, (B, K, D)
        elif name == 'xas_head_softargmax_from_partials':   # (records, B, K, D, nchunk, ...): the first-pass records are read once
            B, K, D, nch = args[1], args[2], args[3], args[4]
            work, sig = 4.0 * B * nch * K * (3 + D), (B, K, D)
        elif name in BN_ENTRIES:
            work, sig = bn_bytes(name, args), None
        elif name in ('xas_head_softargmax_bwd', 'xas_head_softargmax_bwd_amax'):        # (logits, stats, z_idx, grad_kps, B, K, D, ...): read + write
            B, K, D = args[4], args[5], args[6]
            work, sig = 8.0 * B * K * D * D * D, (B, K, D)
        self.records.append((name, a, b, work, mfma, sig))
        self.bytes_total += conv_bytes(shape) if (shape is not None and mfma) else 0.0

    CLASS = {0: ':direct', 1: ':f32', 2: ':bf16', 3: ':bf16x6', 4: ':f16x3'}

    def summary(self):
        """-> dict per (entry point, kernel class): launches, total ms, total flops (call after torch.cuda.synchronize()).
        Keys end in ':direct' (no MFMA), ':f32' (exact-fp32 MFMA), ':bf16', ':bf16x6', ':f16x3'; the head entries carry no suffix."""
        out = {}
        for name, a, b, fl, mfma, _sig in self.records:
            key = name + (self.CLASS[mfma] if name.startswith('xas_conv') else '')
            d = out.setdefault(key, {'launches': 0, 'ms': 0.0, 'flops': 0.0})
            d['launches'] += 1
            d['ms'] += a.elapsed_time(b)
            d['flops'] += fl
        return out

    def by_shape(self):
        """-> list of (name, shape signature, launches, ms, tflops) sorted by time."""
        agg = {}
        for name, a, b, fl, mfma, sig in self.records:
            d = agg.setdefault((name, sig), [0, 0.0, 0.0])
            d[0] += 1
            d[1] += a.elapsed_time(b)
            d[2] += fl
        rows = [(k[0], k[1], v[0], v[1], (v[2] / (v[1] * 1e-3) / 1e12) if v[1] > 0 else 0.0) for k, v in agg.items()]
        return sorted(rows, key=lambda r: -r[3])

"""Layer ops (conv / conv-transpose / linear / batch-norm / pool / upsample / sigmoid) as
autograd Functions over the C ABI.  Activations are torch channels_last tensors: logically
[N,C,H,W] like the reference, physically NHWC like the kernels want."""
import os

import torch
import torch.distributed as dist

from . import _lib
from . import streams
from ._lib import ConvShape, call, ptr, query

CL = torch.channels_last
_weights_epoch = [0]          # bumped by the fused optimizer (raw-pointer updates bypass tensor._version)

# Camera-batched passes: the images of G cameras travel through a network as ONE tensor [G*B, ...] while every
# batch-norm layer keeps separate statistics per camera (G groups of B images), exactly as the reference's G separate
# calls (model.py:64,147,231).  `bn_groups(G)` is entered by the grouped forward of the detector / physique net.
_groups = [1]

# Prefix pass (r04).  The discriminator update needs the detector's outputs on the real images WITHOUT a graph (model.py:231,
# detached at :243); the generator step then runs the detector on the same images (with a graph) and on the pseudo images.
# Nothing between those passes changes the detector, so all of them can travel as ONE grouped pass of 3 * cameras groups -
# 4x / 1.5x larger launches for the late layers - provided the backward leaves the no-grad groups alone.  That is done by
# MEMORY LAYOUT, not by the backward kernels: every activation of the pass is one buffer [prefix images | graph images];
# the tensor that travels through autograd is the TAIL view, the prefix sits in front of it in the same storage.  A forward
# function recovers the whole buffer (`_full`), runs its kernels over all images and groups, and returns / saves tail views
# only: autograd and every backward function see the graph images alone, with their own group count, exactly as if the
# prefix had never been there.
_prefix = [0, 0]            # images, groups of the no-grad prefix of the current pass


class bn_groups:
    def __init__(self, groups, prefix_groups=0, images_per_group=0):
        self.groups = int(groups)
        self.prefix = [int(prefix_groups) * int(images_per_group), int(prefix_groups)]
        if self.prefix[1] and not (0 < self.prefix[1] < self.groups and images_per_group > 0):
            raise RuntimeError('bn_groups: the no-grad prefix must be a proper, non-empty subset of the groups')

    def __enter__(self):
        self.prev, _groups[0] = _groups[0], self.groups
        self.prev_prefix = list(_prefix)
        _prefix[0], _prefix[1] = self.prefix
        return self

    def __exit__(self, *exc):
        _groups[0] = self.prev
        _prefix[0], _prefix[1] = self.prev_prefix
        if not _prefix[0]:
            _prefix_storages.clear()


_prefix_storages = set()      # storages of the buffers of the current prefix pass (mark_prefix_buffer / _tail register them)


def mark_prefix_buffer(buf):
    """Declare `buf` [prefix images + graph images, ...] a buffer of the current prefix pass: only tail views of declared
    buffers are extended backwards by _full (offset arithmetic alone would accept a slice of any user buffer)."""
    _prefix_storages.add(buf.untyped_storage().data_ptr())


def _full(x):
    """The whole buffer of a prefix pass (prefix images + x) from its tail view x; x itself outside a prefix pass.  Outputs of
    a prefix pass are views of such buffers: they must not be modified in place."""
    s = _prefix[0]
    if not s or x is None:
        return x
    per = x.stride(0)
    off = x.storage_offset() - s * per
    if (x.dim() != 4 or not x.is_contiguous(memory_format=CL) or off < 0
            or (x.storage_offset() + x.numel()) * 4 > x.untyped_storage().nbytes()
            or x.untyped_storage().data_ptr() not in _prefix_storages):
        raise RuntimeError('prefix pass: an activation arrived without its %d prefix images in front of it' % s)
    return torch.as_strided(x, (x.shape[0] + s,) + tuple(x.shape[1:]), x.stride(), off)


def _tail(tf, per_image=None):
    """The graph images of a full buffer of the current pass (the buffer itself outside a prefix pass).  per_image: the
    tensor is flat, that many elements per image."""
    s = _prefix[0]
    if not s:
        return tf
    _prefix_storages.add(tf.untyped_storage().data_ptr())        # (a buffer this pass produced)
    return tf[s * per_image:] if per_image is not None else tf[s:]


def current_groups():
    return _groups[0]


def bump_weights_epoch():
    _weights_epoch[0] += 1


# ---- weight gradients on a side stream -------------------------------------------------------------------
# The weight gradient of a conv depends only on (x, dy) and nothing downstream in backward needs it, so it is
# launched on a second HIP stream and ADDED straight into the parameter's .grad (a view of the optimizer's
# gradient arena): the many short wgrad kernels overlap the data-gradient / batch-norm chain of the main
# stream, and autograd's per-contribution `grad += dw` kernels disappear.  join_side_stream() is called before
# anything consumes the gradients (all-reduce, Adam, zero_grad).
# r05: OFF by default (XAS_SIDE_STREAM=1 turns it on).  With more than one HIP stream carrying this library's kernels at the same
# time, a step now and then computes a different result: some wave of some kernel works on wrong values in its lanes 48 - 63
# (single stream: 0 of 2 400 steps; main + this stream: 1 in ~400; with the discriminator update on a third stream as in r04:
# 1 in 6 - tools/diag_repro.py --census, profiles/r05_step_reproducibility.md).  Not explained below the library; until it is,
# the step runs on ONE stream.
_side = {'stream': None, 'enabled': os.environ.get('XAS_SIDE_STREAM', '0') == '1', 'dirty': False}


def side_stream():
    # (measured, round 2: confining this stream to 1/2 or 3/4 of the CUs with hipExtStreamCreateWithCUMask is 7 % slower;
    # a low- or high-priority stream (hipStreamCreateWithPriority) changes nothing: 224.8 / 225.7 / 226.2 ms per step)
    if _side['stream'] is None:
        _side['stream'] = torch.cuda.Stream()
    return _side['stream']


# Lifetime of the tensors a side-stream kernel reads.  `tensor.record_stream(side)` is correct but defers the re-use of the block
# until the HOST sees the side-stream work complete - and the host runs one to two steps ahead of the GPU, so every activation
# and gradient of those steps stayed reserved: 132 GB reserved for a 46 GB working set, and gigabytes of hipMalloc inside the
# timed steps (seconds when the memory had just been released by another process).  Instead the tensors are kept alive in a
# short FIFO; when an entry leaves it the MAIN stream is made to wait for that entry's side-stream event (long past on the
# GPU by then), so whatever re-uses the memory is ordered after its last reader on the device, and nothing is deferred.
_KEEP_DEPTH = int(os.environ.get('XAS_KEEP_DEPTH', '8'))


def _keep_for_side(*tensors):
    ev = torch.cuda.Event()
    ev.record(_side['stream'])
    q = _side.setdefault('keep', [])
    q.append((ev, tensors, torch.cuda.current_stream()))      # the stream the tensors were produced (and allocated) on
    if len(q) > _KEEP_DEPTH:
        old_ev, _, owner = q.pop(0)
        owner.wait_event(old_ev)


def join_side_stream(reset_chains=False):
    """Make the current stream wait for every weight gradient launched on the side stream, and for the pass chains of the
    step (streams.chains: their backward kernels add norm-parameter gradients to the arena on their own streams).
    reset_chains=True forgets the chains afterwards: ONLY the join that follows the generator backward does that
    (engine.TrainStep) - the discriminator update, its reducer and its optimizer join from their own stream in the middle of a
    step, and the generator's chains, recorded before, must still be known to the join after the generator backward."""
    cur = torch.cuda.current_stream()
    if _side['dirty'] and _side['stream'] is not None:
        cur.wait_stream(_side['stream'])
        _side['dirty'] = False
    for s in streams.chain_streams_in_use():
        cur.wait_stream(s)
    if reset_chains:
        streams.reset_chain_use()
    if _side.get('keep'):
        _side['keep'] = []


def side_stream_event():
    """An event that completes when every weight gradient launched so far has been added to the gradient arena
    (None when the side stream is idle).  The data-parallel reducer makes its communication stream wait on it."""
    if _side['dirty'] and _side['stream'] is not None:
        ev = torch.cuda.Event()
        ev.record(_side['stream'])
        return ev
    return None


# ---- gradient readiness for the data-parallel reducer ------------------------------------------------------------
# Conv weight gradients (side stream) and batch-norm parameter gradients (reduce kernel) are added to the gradient
# arena by the kernels themselves, so autograd's post-accumulate hooks never fire for them.  Every forward use of such
# a parameter that will be followed by a backward is counted; each backward contribution counts down; at zero the
# parameter's gradient of this step is complete and the reducer (dp.GradReducer) is told, so a bucket can be
# all-reduced while the rest of backward still runs.
_uses = {'on': False, 'pending': {}, 'hook': None}


def track_grad_uses(on=True):
    _uses['on'] = bool(on)
    _uses['pending'].clear()


def note_use(p):
    # keyed by storage address: inside autograd.Function.forward the tensor arguments are not the Python objects the
    # caller passed (their id() differs from the nn.Parameter's), the arena address of a parameter is unique and stable
    if _uses['on'] and p is not None:
        k = p.data_ptr()
        _uses['pending'][k] = _uses['pending'].get(k, 0) + 1


def grad_ready(p):
    if not _uses['on']:
        return
    k = p.data_ptr()
    if k not in _uses['pending']:
        # no counted forward use (the count was taken before tracking started, or belongs to a cleared epoch): never
        # report - the parameter's bucket is then launched by GradReducer.finish(), after ALL contributions
        return
    left = _uses['pending'][k] - 1
    if left > 0:
        _uses['pending'][k] = left
        return
    del _uses['pending'][k]
    if _uses['hook'] is not None:
        _uses['hook'](p)


def forget_uses(keys):
    """Drop the counts of the given parameter addresses (a reducer clears ITS parameters after its backward; counts
    of another reducer's parameters - e.g. detector forwards taken before the discriminator step under
    TrainStep(dedupe=True) - stay)."""
    for k in keys:
        _uses['pending'].pop(k, None)


def _wgrad_into_grad(x, dy, shp, weight):
    """Accumulate the OIHW weight gradient straight into weight.grad (a view of the optimizer's gradient arena): on the side
    stream when that is on, else on the current stream - either way without a gradient tensor of its own and without autograd's
    `grad += dw` kernel (one per parameter and step otherwise: 160 launches).  False if not applicable."""
    g = weight.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or not g.is_cuda:
        return False
    if not _side['enabled']:
        ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp)), device=x.device, dtype=torch.float32)
        call('xas_conv_wgrad_acc', ptr(x), ptr(dy), ptr(g), ptr(ws), shp)
        grad_ready(weight)
        return True
    main, side = torch.cuda.current_stream(), side_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp)), device=x.device, dtype=torch.float32)
        call('xas_conv_wgrad_acc', ptr(x), ptr(dy), ptr(g), ptr(ws), shp)
    _keep_for_side(x, dy)
    _side['dirty'] = True
    grad_ready(weight)
    return True


# Off by default: measured neutral (218.7 vs 218.7 ms/step) - the backward pass is bound by aggregate throughput, not by the
# length of the main stream's queue, so moving 3.9 ms of column sums to the other stream changes nothing.
BIAS_ON_SIDE = os.environ.get('XAS_BIAS_SIDE', '0') == '1'


def _bias_into_grad(dy, M, C, bias, force=False):
    """bias.grad += column sums of dy [M, C] on the side stream (off the critical stream); False if not applicable."""
    g = bias.grad
    if (not (BIAS_ON_SIDE or force or not _side['enabled']) or g is None or not g.is_contiguous() or g.dtype != torch.float32 or C % 4
            or g.data_ptr() % 16 or not g.is_cuda):
        return False
    if not _side['enabled']:
        ws = torch.empty(query('xas_bn_workspace_floats', M, C, 1), device=dy.device, dtype=torch.float32)
        call('xas_col_sum_acc', ptr(dy), M, C, ptr(g), ptr(ws))
        grad_ready(bias)
        return True
    main, side = torch.cuda.current_stream(), side_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ws = torch.empty(query('xas_bn_workspace_floats', M, C, 1), device=dy.device, dtype=torch.float32)
        call('xas_col_sum_acc', ptr(dy), M, C, ptr(g), ptr(ws))
    _keep_for_side(dy)
    _side['dirty'] = True
    grad_ready(bias)
    return True


def to_cl(x):
    """Return x with NHWC storage (no copy when it already has it)."""
    if x.dim() != 4:
        raise RuntimeError('expected a 4-D activation tensor')
    return x if x.is_contiguous(memory_format=CL) else x.contiguous(memory_format=CL)


def empty_cl(n, c, h, w, like):
    return torch.empty((n, c, h, w), device=like.device, dtype=torch.float32, memory_format=CL)


def from_nchw(x):
    """[N,C,H,W] contiguous NCHW image batch -> channels_last tensor (xas_nchw_to_nhwc)."""
    if x.is_contiguous(memory_format=CL):
        return x
    x = x.contiguous()
    n, c, h, w = x.shape
    y = empty_cl(n, c, h, w, x)
    call('xas_nchw_to_nhwc', ptr(x), n, c, h, w, ptr(y))
    return y


def stack_nchw(tensors):
    """Equal-shaped NCHW image batches (one per camera) -> ONE channels_last tensor [sum N, C, H, W]: each batch is converted
    straight into its slice, instead of torch.cat (a 100 MB copy per pass) followed by the layout conversion."""
    t0 = tensors[0]
    if (len(tensors) == 1 or t0.dim() != 4 or not t0.is_cuda or t0.shape[1] == 1
            or any(t.requires_grad or (t.is_contiguous(memory_format=CL) and t.shape[1] > 1) for t in tensors)):      # autograd inputs: torch.cat
        return torch.cat(list(tensors), dim=0) if len(tensors) > 1 else t0
    n, c, h, w = t0.shape
    y = empty_cl(n * len(tensors), c, h, w, t0)
    for g, t in enumerate(tensors):
        t = t.contiguous()
        call('xas_nchw_to_nhwc', ptr(t), n, c, h, w, ptr(y[g * n:(g + 1) * n]))
    return y


def _shape(n, hi, wi, cin, cout, r, s, stride, pad, ho, wo, mode=0):
    return ConvShape(n, hi, wi, cin, cout, r, s, stride, pad, ho, wo, mode)


MODE_F32 = 1 + _lib.PREC_F32          # ConvShape.mode of a call that must run on the exact-fp32 kernels
GRAD_IS_X = 0x100                     # xas_hip.h XAS_GRAD_IS_X


# ---- maxima of tensor operands (XAS_PREC_F16X3) ------------------------------------------------------------------
# The f16x3 kernels split EVERY tensor operand - activations and gradients - into two fp16 pieces at a power-of-two scale
# taken from max |tensor| (r04: no fixed activation scale, hence no range an activation has to stay in).  The kernel that
# WRITES a tensor merges that maximum into a float of this arena (xas_bn_apply_amax for activations - every conv input of
# both networks except the images and the rendered masks is a norm's output; xas_bn_bwd_apply_amax / the soft-argmax
# backward for gradients); the tensor object carries the slot (`_xas_amax`) to the conv that reads it, which passes the
# pointer on in its ConvShape.  A conv INPUT that arrives without a slot gets one from xas_abs_max (one streaming read: the
# images of a pass, 40 us at B = 32 x 8); a GRADIENT that arrives without a slot runs on the bf16x6 kernels.  Max pooling
# and bilinear up-sampling hand their input's slot on (max |y| <= max |x|: an upper bound serves as well).  Slots are handed
# out in order from a zeroed arena; reset_grad_amax() (engine.TrainStep, once per step, after the streams have joined)
# rewinds and re-zeroes it and starts a new EPOCH: a tag from an earlier epoch (a tensor that outlived its step, e.g. a
# resident input batch) is void.
_amax = {'arena': None, 'next': 0, 'retired': [], 'epoch': 0}
_AMAX_SLOTS = 2048
AMAX_SLOT_FLOATS = 1024            # xas_hip.h XAS_AMAX_SLOT_FLOATS: 32 sub-maxima 128 bytes apart (one atomic per block, spread)
GRAD_F16 = os.environ.get('XAS_GRAD_F16', '1') == '1'
amax_stats = {'abs_max': 0}        # launches of xas_abs_max (tests: the hot path needs one per pass, for the images)


def reset_grad_amax():
    if _amax['arena'] is not None and _amax['next'] > 0:
        _amax['arena'][:_amax['next'] * AMAX_SLOT_FLOATS].zero_()          # (only the slots that were handed out)
    _amax['next'] = 0
    _amax['retired'] = []
    _amax['epoch'] += 1


def _amax_slot(device):
    a = _amax['arena']
    if a is None or a.device != device or _amax['next'] >= _AMAX_SLOTS:
        if a is not None:
            _amax['retired'] = (_amax['retired'] + [a])[-4:]      # kernels in flight may still read the last few
        a = _amax['arena'] = torch.zeros(_AMAX_SLOTS * AMAX_SLOT_FLOATS, device=device, dtype=torch.float32)
        _amax['next'] = 0
    i = _amax['next']
    _amax['next'] = i + 1
    return a[i * AMAX_SLOT_FLOATS:(i + 1) * AMAX_SLOT_FLOATS]


def amax_slot_from_value(v):
    """An amax slot holding the maximum `v` (0-dim or 1-element device tensor) computed elsewhere - tests, tools."""
    s = torch.zeros(AMAX_SLOT_FLOATS, device=v.device, dtype=torch.float32)
    s[0] = v.reshape(())
    return s


def f16x3_on():
    return GRAD_F16 and query('xas_get_precision') == _lib.PREC_F16X3


def grad_amax_slot(device):
    """A zeroed float for a kernel that writes a tensor to merge max |v| into; None outside XAS_PREC_F16X3."""
    if not f16x3_on():
        return None
    return _amax_slot(device)


def tag_grad_amax(t, slot):
    """Attach the slot that holds max |t| (or an upper bound) to the tensor t (read by amax_of / with_grad_amax)."""
    t._xas_amax = (slot, t._version, _amax['epoch'])


tag_amax = tag_grad_amax


def amax_of(t):
    """The slot recorded for tensor t, or None.  The tag holds the tensor's version counter at the time the maximum was
    recorded - a tensor modified IN PLACE since (a gradient autograd accumulated into: a conv output with two consumers)
    no longer matches its maximum - and the epoch of the slot arena (reset_grad_amax)."""
    tag = getattr(t, '_xas_amax', None)
    if tag is None or tag[1] != t._version or tag[2] != _amax['epoch']:
        return None
    return tag[0]


def act_amax(x):
    """Slot holding max |x| for the conv INPUT x (dense fp32 device tensor): the producer's, else measured now.  None
    outside XAS_PREC_F16X3."""
    if not x.is_cuda or x.dtype != torch.float32 or not f16x3_on():
        return None
    slot = amax_of(x)
    if slot is None:
        slot = _amax_slot(x.device)
        xm = _full(x) if (x.dim() == 4 and _prefix[0]) else x          # (a prefix pass: the kernels read the whole buffer)
        call('xas_abs_max', ptr(xm), xm.numel(), ptr(slot))
        tag_grad_amax(x, slot)
        amax_stats['abs_max'] += 1
    return slot


def pass_amax(y, x):
    """y = f(x) with max |y| <= max |x| (max pooling, bilinear interpolation): y inherits x's slot."""
    slot = amax_of(x)
    if slot is not None:
        tag_grad_amax(y, slot)


def _with_ptrs(shp, grad, x, mode=None):
    return ConvShape(shp.N, shp.Hi, shp.Wi, shp.Cin, shp.Cout, shp.R, shp.S, shp.stride, shp.pad, shp.Ho, shp.Wo,
                     shp.mode if mode is None else mode, grad.data_ptr() if grad is not None else None,
                     x.data_ptr() if x is not None else None)


def _with_n(shp, n):
    """The same convolution over n images (keeps mode and operand-maximum pointers)."""
    if n == shp.N:
        return shp
    out = ConvShape(n, shp.Hi, shp.Wi, shp.Cin, shp.Cout, shp.R, shp.S, shp.stride, shp.pad, shp.Ho, shp.Wo, shp.mode,
                    shp.grad_amax, shp.x_amax)
    if hasattr(shp, '_slots'):
        out._slots = shp._slots
    return out


def with_act_amax(shp, x):
    """(ConvShape of a launch whose tensor operand is the ACTIVATION x, slot): with the pointer to max |x| in f16x3 mode
    (shape unchanged and None otherwise)."""
    if shp.mode != 0:
        return shp, None
    slot = act_amax(x)
    if slot is None:
        return shp, None
    shp = _with_ptrs(shp, slot, None)
    shp._slots = (slot,)                          # (the ctypes struct holds a raw pointer: keep the slot tensor alive with it)
    return shp, slot


def with_grad_amax(shp, dy, x_slot=None):
    """ConvShape of a gradient launch that reads `dy`: grad_amax -> max |dy| when its producer recorded one, x_amax -> the
    slot of the forward's input (weight gradient).  `shp` may be the forward's shape (its own pointers are dropped).
    No valid maximum of dy: no pointers at all - every pass of the call runs as bf16x6."""
    if shp.mode != 0:
        return shp
    slot = amax_of(dy) if f16x3_on() else None
    if slot is None:
        return _with_ptrs(shp, None, None) if (shp.grad_amax or shp.x_amax) else shp
    out = _with_ptrs(shp, slot, x_slot)
    out._slots = (slot, x_slot)
    return out


def shape_with_maxima(shp, t, x=None):
    """For callers outside the autograd layers (tests, tools/): the shape with the MEASURED maximum of the launch's tensor
    operand t (x of a forward, dy of a gradient launch) and, for a weight gradient, of its x argument."""
    a = act_amax(t)
    b = act_amax(x) if x is not None else None
    if a is None or shp.mode != 0:
        return shp
    out = _with_ptrs(shp, a, b)
    out._slots = (a, b)
    return out


def grad_operand_shape(shp):
    """ConvShape for a FORWARD-type launch whose input is a gradient tensor WITHOUT a recorded maximum (the data gradient of
    a ConvTranspose2d is a forward convolution of dy): pinned to the bf16x6 kernels."""
    if shp.mode == 0 and query('xas_get_precision') == _lib.PREC_F16X3:
        return _with_ptrs(shp, None, None, mode=1 + _lib.PREC_BF16X6)
    return shp


class _PackCache:
    """Kernel-side copies of a weight, rebuilt when the parameter changes: packed fp32 ([rows][R][S][chan]) and, for the
    bf16-split kernels, the pre-split bf16 planes of the packed weight (xas_split_weight) - built once per optimizer step."""

    def __init__(self):
        self.key = None
        self.packed = {}

    def _fresh(self, w):
        # the optimizer that owns w bumps ITS epoch box when it rewrites the arena (raw-pointer update: w._version does
        # not move); weights outside any fused optimizer fall back to the global epoch
        box = getattr(w, '_xas_epoch', _weights_epoch)
        key = (w.data_ptr(), w._version, box[0])
        if key != self.key:
            self.key, self.packed = key, {}

    def get(self, w, transposed, shp=None, planes=None):
        """Weight buffer for the forward-type (transposed = 0) / data-gradient-type (1) entry points.  shp: the call's
        ConvShape - the library says which format that (shape, precision) wants; None: fp32 packed.  planes: build
        this format regardless (prepack)."""
        self._fresh(w)
        if planes is None:
            planes = query('xas_conv_weight_planes', shp, int(transposed)) if shp is not None else 0
        k = (transposed, planes)
        if k in self.packed:
            return self.packed[k]
        if (transposed, 0) not in self.packed:
            co, ci, r, s = w.shape
            if not transposed and r == 1 and s == 1 and w.is_contiguous() and w.data_ptr() % 16 == 0:
                # a 1x1 filter in OIHW order IS the packed [Cout][R][S][Cin] layout: no copy (36 of ResNet-50's 53 convs)
                self.packed[(transposed, 0)] = w.detach().view(-1)
            else:
                p = torch.empty(w.numel(), device=w.device, dtype=torch.float32)
                call('xas_pack_weight', ptr(w.detach().contiguous()), ptr(p), co, ci, r, s, int(transposed))
                self.packed[(transposed, 0)] = p
        if planes:
            src = self.packed[(transposed, 0)]
            rows = w.shape[1] if transposed else w.shape[0]
            kk = src.numel() // rows
            sp = torch.empty(query('xas_split_weight_bytes', rows, kk, planes), device=w.device, dtype=torch.uint8)
            call('xas_split_weight', ptr(src), ptr(sp), rows, kk, planes)
            self.packed[k] = sp
        return self.packed[k]


def _col_sum(t2d_ptr_tensor, M, C):
    out = torch.empty(C, device=t2d_ptr_tensor.device, dtype=torch.float32)
    ws = torch.empty(query('xas_bn_workspace_floats', M, C, 1), device=out.device, dtype=torch.float32)
    call('xas_col_sum', ptr(t2d_ptr_tensor), M, C, ptr(out), ptr(ws))
    return out


def _bias_grad(dy, M, C):
    if C % 4 == 0:
        return _col_sum(dy, M, C)
    # C == 1 (mask output conv, logit layer): a plain sum of a contiguous vector
    return dy.reshape(M, C).sum(0)


def _wgrad(x, dy, shp, w_shape, transposed=False):
    """-> gradient in OIHW layout for a conv described by shp (x: gathered side, dy: row side)."""
    co, ci, r, s = shp.Cout, shp.Cin, shp.R, shp.S
    ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp)), device=x.device, dtype=torch.float32)
    dw = torch.empty(w_shape, device=x.device, dtype=torch.float32)
    call('xas_conv_wgrad_oihw', ptr(x), ptr(dy), ptr(dw), ptr(ws), shp)
    return dw


# The final 1x1 convolution of the detector hands the soft-argmax head its first-pass records (xas_conv_fwd_head): the logits
# tensor carries them to ops_head._SoftArgmax as `_xas_head` = (records, chunks per image, version of the logits).
# XAS_HEAD_IN_EPILOGUE=0: the head reads the logits itself (head_partial_kernel), as before r05.
HEAD_IN_EPILOGUE = os.environ.get('XAS_HEAD_IN_EPILOGUE', '1') == '1'
head_stats = {'fused': 0, 'separate': 0}      # launches of either form (tests)


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, cache, head_kd=None):
        x = to_cl(x)
        n, ci, hi, wi = x.shape
        co, ci2, r, s = weight.shape
        if ci != ci2:
            raise RuntimeError('conv2d: input has %d channels, weight expects %d' % (ci, ci2))
        ho, wo = (hi + 2 * pad - r) // stride + 1, (wi + 2 * pad - s) // stride + 1
        shp, x_slot = with_act_amax(_shape(n, hi, wi, ci, co, r, s, stride, pad, ho, wo), x)
        xf = _full(x)                                     # (prefix pass: the kernel runs over prefix + graph images)
        yf = empty_cl(xf.shape[0], co, ho, wo, x)
        shp_f = _with_n(shp, xf.shape[0])
        chunks = query('xas_conv_fwd_head_chunks', shp_f, head_kd[0], head_kd[1]) if (head_kd and HEAD_IN_EPILOGUE and x.is_cuda) else 0
        if chunks:
            rec = torch.empty(xf.shape[0], chunks, head_kd[0], 3 + head_kd[1], device=x.device, dtype=torch.float32)
            call('xas_conv_fwd_head', ptr(xf), ptr(cache.get(weight, 0, shp_f)), ptr(bias), ptr(yf), shp_f, head_kd[0], head_kd[1], ptr(rec))
        else:
            call('xas_conv_fwd', ptr(xf), ptr(cache.get(weight, 0, shp_f)), ptr(bias), ptr(yf), shp_f)
        y = _tail(yf)
        if chunks:
            y._xas_head = (rec, chunks, y._version)
        ctx.save_for_backward(x, weight, *([bias] if bias is not None else []))
        ctx.shp, ctx.cache, ctx.has_bias, ctx.x_slot = shp, cache, bias is not None, x_slot
        if ctx.needs_input_grad[1]:
            note_use(weight)
        if bias is not None and ctx.needs_input_grad[2]:
            note_use(bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, *rest = ctx.saved_tensors
        shp = with_grad_amax(ctx.shp, dy, ctx.x_slot)
        dy = to_cl(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            call('xas_conv_dgrad', ptr(dy), ptr(ctx.cache.get(weight, 1, shp)), ptr(dx), shp)
        if ctx.needs_input_grad[1] and not _wgrad_into_grad(x, dy, shp, weight):
            dw = _wgrad(x, dy, shp, weight.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            M = shp.N * shp.Ho * shp.Wo
            if not _bias_into_grad(dy, M, shp.Cout, rest[0]):
                db = _bias_grad(dy, M, shp.Cout)          # (autograd accumulates it and its hook reports readiness)
        return dx, dw, db, None, None, None, None


def conv2d(x, weight, bias, stride, pad, cache, head_kd=None):
    return _Conv2d.apply(x, weight, bias, stride, pad, cache, head_kd)


def _conv_forward(x, weight, stride, pad, cache):
    """Bias-free convolution -> (y, shape descriptor)."""
    n, ci, hi, wi = x.shape
    co, ci2, r, s = weight.shape
    if ci != ci2:
        raise RuntimeError('conv2d: input has %d channels, weight expects %d' % (ci, ci2))
    ho, wo = (hi + 2 * pad - r) // stride + 1, (wi + 2 * pad - s) // stride + 1
    shp, _ = with_act_amax(_shape(n, hi, wi, ci, co, r, s, stride, pad, ho, wo), x)      # (the shape keeps the input's slot)
    xf = _full(x)
    yf = empty_cl(xf.shape[0], co, ho, wo, x)
    shp_f = _with_n(shp, xf.shape[0])
    call('xas_conv_fwd', ptr(xf), ptr(cache.get(weight, 0, shp_f)), None, ptr(yf), shp_f)
    return _tail(yf), shp


FUSE_CONV_STATS = os.environ.get('XAS_CONV_STATS', '1') == '1'


def _conv_bn_forward(x, conv, bn, residual, group):
    """conv (no bias) -> batch norm (+ residual, activation): -> (out, saved, cfg, shp).  In training mode the batch
    statistics come out of the convolution's epilogue (xas_conv_fwd_bnstats): the conv result is not read a second time
    for them."""
    weight = conv.weight
    if not (FUSE_CONV_STATS and bn.training):
        y, shp = _conv_forward(x, weight, conv.stride, conv.padding, conv._cache)
        out, sv, cf = _bn_forward(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, bn.training,
                                  bn.momentum, bn.eps, bn.act, group)
        return out, sv, cf, shp
    n, ci, hi, wi = x.shape
    co, ci2, r, s = weight.shape
    if ci != ci2:
        raise RuntimeError('conv2d: input has %d channels, weight expects %d' % (ci, ci2))
    stride, pad = conv.stride, conv.padding
    ho, wo = (hi + 2 * pad - r) // stride + 1, (wi + 2 * pad - s) // stride + 1
    shp, _ = with_act_amax(_shape(n, hi, wi, ci, co, r, s, stride, pad, ho, wo), x)
    xf = _full(x)                                         # (prefix pass: prefix + graph images, all groups)
    shp_f = _with_n(shp, xf.shape[0])
    yf = empty_cl(xf.shape[0], co, ho, wo, x)
    y = _tail(yf)
    wp = conv._cache.get(weight, 0, shp_f)
    G = _groups[0]
    # sums are taken around 0: a pivot taken from the running mean would make the last bits of the statistics depend on
    # optimisation state (a detector pass computed once and re-used - TrainStep(dedupe=True) - must equal the recomputed
    # one bit for bit); per-tile sums run over <= 128 rows in fp32 and are combined in double
    pivot = None

    def stats(mean_p, var_p, out_stride, count_p, rm_p, rv_p, momentum):
        ws = torch.empty(query('xas_conv_fwd_bnstats_workspace_floats', shp_f, G), device=x.device, dtype=torch.float32)
        call('xas_conv_fwd_bnstats', ptr(xf), ptr(wp), ptr(yf), shp_f, G, ptr(pivot), mean_p, var_p, out_stride, count_p,
             ptr(ws), rm_p, rv_p, float(momentum))

    out, sv, cf = _bn_forward(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, True, bn.momentum,
                              bn.eps, bn.act, group, stats_fn=stats)
    return out, sv, cf, shp


# Off by default.  Measured (round 2, B = 32 x 4 cameras, interleaved in one process): 222.5 ms/step without, 223.3 ms with.
# The two reductions it removes (col_reduce_lean_kernel<4>, 7.6 ms/step) run BESIDE the weight gradients of the side
# stream and are mostly hidden; the extra epilogue work lands in the data-gradient kernels, which are the critical path.
FUSE_DGRAD_BN = os.environ.get('XAS_DGRAD_BN', '0') == '1'


def _bn_bwd_fusable(cfg):
    """A rank-local training-mode norm with ReLU and no residual whose backward re-derives the mask from its input."""
    M, c, eps, act, count, group, training, has_res, xfree, yfree, G, masked = cfg
    return training and group is None and act == ACT_RELU and not has_res and yfree and not masked


def _conv_dgrad_bn_bwd(dy, conv, shp, saved, cfg, bn, want_param_grads):
    """Data gradient of `conv` (its input is relu(bn(xb))) + the backward of that norm: -> (dxb, (dgamma, dbeta) | None).
    One C call (xas_conv_dgrad_bn_bwd): the norm's two reductions ride in the data gradient's epilogue."""
    xb, _, mean, var = saved
    M, c, eps, act, count, group, training, has_res, xfree, yfree, G, masked = cfg
    dy = to_cl(dy)
    dev = xb.device
    gamma, beta = bn.weight, bn.bias
    sums = torch.empty(G, 2, c, device=dev, dtype=torch.float32)
    ws = torch.empty(query('xas_conv_dgrad_bn_bwd_workspace_floats', shp, G), device=dev, dtype=torch.float32)
    gg, gb = gamma.grad, beta.grad
    direct = (want_param_grads and gg is not None and gb is not None and gg.is_contiguous() and gb.is_contiguous()
              and gg.dtype == torch.float32 and gb.dtype == torch.float32)
    dz = torch.empty_like(xb)
    dx = torch.empty_like(xb)
    shp_n = _with_ptrs(shp, None, None)          # this entry takes the weight format of the shape without maxima (xas_hip.h)
    call('xas_conv_dgrad_bn_bwd', ptr(dy), ptr(conv._cache.get(conv.weight, 1, shp_n)), shp_n, ptr(xb), ptr(mean), ptr(var),
         ptr(gamma), ptr(beta), float(eps), G, float(count), ptr(dz), ptr(dx), ptr(sums), ptr(ws),
         ptr(gb) if direct else None, ptr(gg) if direct else None)
    if direct:
        grad_ready(gamma)
        grad_ready(beta)
        return dx, None
    if want_param_grads:
        return dx, (sums[:, 1].sum(0), sums[:, 0].sum(0))
    return dx, None


def _x_slot_of(fwd_shp):
    """The input's slot a forward shape carries (with_act_amax), for the weight gradient of the same layer."""
    sl = getattr(fwd_shp, '_slots', None)
    return sl[0] if sl else None


def _conv_backward(x, weight, dy, shp, cache, need_dx, need_dw, acc_into=None):
    """-> (dx, dw).  acc_into: a gradient buffer of x's shape that already holds the other branch's gradient; the
    data gradient is added to it in the kernel epilogue.  dw is None when it went straight into weight.grad."""
    dx = dw = None
    shp = with_grad_amax(shp, dy, _x_slot_of(shp))
    if need_dx:
        if acc_into is not None:
            dx = acc_into
            call('xas_conv_dgrad_acc', ptr(dy), ptr(cache.get(weight, 1, shp)), ptr(dx), shp)
        else:
            dx = torch.empty_like(x)
            call('xas_conv_dgrad', ptr(dy), ptr(cache.get(weight, 1, shp)), ptr(dx), shp)
    if need_dw and not _wgrad_into_grad(x, dy, shp, weight):
        dw = _wgrad(x, dy, shp, weight.shape)
    return dx, dw


class _Bottleneck(torch.autograd.Function):
    """conv1-bn1-relu, conv2-bn2-relu, conv3-bn3 (+ skip | downsample conv-bn), add, relu as ONE autograd node
    (torchvision Bottleneck v1.5, the block the reference imports at integral_base_modules/resnet.py:2).  Same kernels
    and the same arithmetic as the layer-by-layer path; what the node buys is the backward order under our control:
    the skip gradient produced by bn3's backward is the buffer conv1's data gradient accumulates into
    (xas_conv_dgrad_acc), so the block-input gradient needs no separate addition pass, and the host walks 1 autograd
    node per block instead of 6-8."""

    @staticmethod
    def forward(ctx, x, blk, *params):
        x = to_cl(x)
        convs = [blk.conv1, blk.conv2, blk.conv3]
        bns = [blk.bn1, blk.bn2, blk.bn3]
        ds = blk.downsample
        for bn in bns + ([ds[1]] if ds is not None else []):
            bn.count_batch()
        saved, cfgs, shps = [], [], []
        skip = x
        if ds is not None:
            skip, sv, cf, shp_d = _conv_bn_forward(x, ds[0], ds[1], None, ds[1].sync_group())
            saved.append(sv); cfgs.append(cf); shps.append(shp_d)
        a = x
        acts = [x]
        for i in range(3):
            a, sv, cf, shp = _conv_bn_forward(a, convs[i], bns[i], skip if i == 2 else None, bns[i].sync_group())
            saved.append(sv); cfgs.append(cf); shps.append(shp)
            acts.append(a)
        flat = [t for sv in saved for t in sv]
        ctx.save_for_backward(x, acts[1], acts[2], *flat)
        ctx.blk, ctx.cfgs, ctx.shps, ctx.has_ds = blk, cfgs, shps, ds is not None
        if _uses['on']:
            for p, need in zip(params, ctx.needs_input_grad[2:]):
                if need:
                    note_use(p)
        return a

    @staticmethod
    def backward(ctx, dout):
        blk = ctx.blk
        x, a1, a2, *flat = ctx.saved_tensors
        saved = [tuple(flat[4 * i:4 * i + 4]) for i in range(len(flat) // 4)]
        cfgs, shps = ctx.cfgs, ctx.shps
        o = 1 if ctx.has_ds else 0                    # index of conv1's entries
        convs = [blk.conv1, blk.conv2, blk.conv3]
        bns = [blk.bn1, blk.bn2, blk.bn3]
        ins = [x, a1, a2]
        pgrads = {}

        # parameters that actually want a gradient (frozen ones get none)
        need = {id(p) for p, n in zip(blk._fused_params, ctx.needs_input_grad[2:]) if n}

        def bn_b(idx, bn, dy, want_dres=True):
            dx, dg, db, dres = _bn_backward(saved[idx], cfgs[idx], bn.weight, bn.bias, dy,
                                            id(bn.weight) in need and id(bn.bias) in need, want_dres)
            if dg is not None:
                if id(bn.weight) in need:
                    pgrads[id(bn.weight)] = dg
                if id(bn.bias) in need:
                    pgrads[id(bn.bias)] = db
            return dx, dres

        need_dx = ctx.needs_input_grad[0]
        # blocks without a projection: the skip gradient (= relu'(block output) * dout) is formed in the epilogue of conv1's
        # data gradient from dout and the sign bytes bn3's forward saved - it is never written to memory
        fuse_skip = (not ctx.has_ds and need_dx and cfgs[o + 2][11] and _can_accumulate(shps[o]))
        dout = to_cl(dout)
        g, dskip = bn_b(o + 2, bns[2], dout, want_dres=not fuse_skip)          # dskip: gradient of the skip branch
        for i in (2, 1):
            bn, cf = bns[i - 1], cfgs[o + i - 1]
            want_bn = id(bn.weight) in need and id(bn.bias) in need
            if FUSE_DGRAD_BN and _bn_bwd_fusable(cf) and (want_bn or (id(bn.weight) not in need and id(bn.bias) not in need)):
                # conv_i's data gradient with bn_{i-1}'s backward reductions in its epilogue
                dyc = g
                shp_b = with_grad_amax(shps[o + i], dyc, _x_slot_of(shps[o + i]))      # (the fused data gradient ignores the maxima)
                g, dgb = _conv_dgrad_bn_bwd(dyc, convs[i], shp_b, saved[o + i - 1], cf, bn, want_bn)
                if id(convs[i].weight) in need and not _wgrad_into_grad(ins[i], dyc, shp_b, convs[i].weight):
                    pgrads[id(convs[i].weight)] = _wgrad(ins[i], dyc, shp_b, convs[i].weight.shape)
                if dgb is not None:
                    pgrads[id(bn.weight)], pgrads[id(bn.bias)] = dgb
                continue
            g, dw = _conv_backward(ins[i], convs[i].weight, g, shps[o + i], convs[i]._cache, True, id(convs[i].weight) in need)
            if dw is not None:
                pgrads[id(convs[i].weight)] = dw
            g, _ = bn_b(o + i - 1, bns[i - 1], g)
        if fuse_skip:
            dx = torch.empty_like(x)
            shp0 = with_grad_amax(shps[o], g, _x_slot_of(shps[o]))
            call('xas_conv_dgrad_acc_masked', ptr(g), ptr(convs[0]._cache.get(convs[0].weight, 1, shp0)), ptr(dx), shp0,
                 ptr(dout), ptr(saved[o + 2][1]))
            if id(convs[0].weight) in need and not _wgrad_into_grad(x, g, shp0, convs[0].weight):
                pgrads[id(convs[0].weight)] = _wgrad(x, g, shp0, convs[0].weight.shape)
            return (dx, None) + tuple(pgrads.get(id(p)) for p in ctx.blk._fused_params)
        if ctx.has_ds:
            ds = blk.downsample
            gd, _ = bn_b(0, ds[1], dskip)
            dx, dw = _conv_backward(x, ds[0].weight, gd, shps[0], ds[0]._cache, need_dx, id(ds[0].weight) in need)
            if dw is not None:
                pgrads[id(ds[0].weight)] = dw
            acc = dx
        else:
            acc = dskip
        dx, dw = _conv_backward(x, convs[0].weight, g, shps[o], convs[0]._cache, need_dx, id(convs[0].weight) in need,
                                acc_into=acc if (need_dx and _can_accumulate(shps[o])) else None)
        if dw is not None:
            pgrads[id(convs[0].weight)] = dw
        if need_dx and dx is not acc:
            dx = dx + acc
        return (dx if need_dx else None, None) + tuple(pgrads.get(id(p)) for p in ctx.blk._fused_params)


def _can_accumulate(shp):
    return shp.Cout % 32 == 0 and shp.Cin % 4 == 0 and shp.Cin >= 16


def bottleneck(x, blk):
    return _Bottleneck.apply(x, blk, *blk._fused_params)


class _ConvTranspose2d(torch.autograd.Function):
    """y = conv_transpose2d(x, weight[Cin_t, Cout_t, R, S], stride, pad) == data-gradient of the conv
    whose OIHW weight has the same memory layout (deconv_head.py:27-29)."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, cache):
        x = to_cl(x)
        n, cit, h, w = x.shape
        cit2, cot, r, s = weight.shape
        if cit != cit2:
            raise RuntimeError('conv_transpose2d: channel mismatch')
        hb, wb = (h - 1) * stride - 2 * pad + r, (w - 1) * stride - 2 * pad + s
        # equivalent conv: big side (hb,wb,cot) -> small side (h,w,cit)
        shp = _shape(n, hb, wb, cot, cit, r, s, stride, pad, h, w)
        # the gathered operand of this data-gradient-type launch is the activation x: its maximum selects the scale
        shp_f, x_slot = with_act_amax(shp, x)
        xf = _full(x)                                     # (prefix pass: prefix + graph images)
        shp_f = _with_n(shp_f, xf.shape[0])
        yf = empty_cl(xf.shape[0], cot, hb, wb, x)
        call('xas_conv_dgrad', ptr(xf), ptr(cache.get(weight, 1, shp_f)), ptr(yf), shp_f)
        y = _tail(yf)
        ctx.save_for_backward(x, weight)
        ctx.shp, ctx.cache, ctx.x_slot = shp, cache, x_slot
        if ctx.needs_input_grad[1]:
            note_use(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        shp = ctx.shp
        dy_in = dy
        dy = to_cl(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            shp_g = with_grad_amax(shp, dy_in)             # dy with its maximum: f16x3 at that scale; else pinned to bf16x6
            if shp_g is shp:
                shp_g = grad_operand_shape(shp)
            call('xas_conv_fwd', ptr(dy), ptr(ctx.cache.get(weight, 0, shp_g)), None, ptr(dx), shp_g)
        shp_w = with_grad_amax(shp, dy_in, ctx.x_slot)      # grad_amax: max |dy|, x_amax: max |x| ...
        if shp_w is not shp:
            shp_w.mode = GRAD_IS_X                          # ... and the gradient tensor is the weight gradient's `x` argument here
        if ctx.needs_input_grad[1] and not _wgrad_into_grad(dy, x, shp_w, weight):
            dw = _wgrad(dy, x, shp_w, weight.shape)
        return dx, dw, None, None, None


def conv_transpose2d(x, weight, stride, pad, cache):
    return _ConvTranspose2d.apply(x, weight, stride, pad, cache)


# Parameter gradients of the discriminator's linear layers go to the weight-gradient stream and straight into .grad, like
# the convolutions': the backward of the adversarial term is a chain of 21 small layers on the second stream that the
# detector's backward waits for, and two of the three launches of every layer (weight and bias gradient) are not on it.
LINEAR_SIDE = os.environ.get('XAS_LINEAR_SIDE', '1') == '1'


# r04: linear layers wide enough for the MFMA tiles run on the convolution kernels of the process's precision mode (default:
# the split-arithmetic kernels - without operand maxima that is bf16x6, fp32-accurate and range-free) instead of always on
# the exact-fp32 pipe: their rows are presented as ONE image of H x W pixels (a 1x1 convolution does not care how its pixels
# are arranged), which is what the split kernels' pixel decode wants.  The discriminator's 128 -> 128 layers over
# B * 18 * 12 rows: 132 -> ~25 us per weight gradient, 50 -> ~12 us forward.  XAS_LINEAR_MFMA=0: as before.
LINEAR_MFMA = os.environ.get('XAS_LINEAR_MFMA', '1') == '1'


def _row_map(rows):
    """rows = H * W with a width the split kernels' pixel decode accepts (W >= 32 or few carries), or None."""
    for w in (96, 64, 128, 72, 48, 32, 80, 112, 56, 40, 36, 24, 16):
        if rows % w == 0 and rows // w >= 2 and 31 // w + 1 <= rows // w:
            return rows // w, w
    return None


class _Linear(torch.autograd.Function):
    """y = x @ W^T + b as a 1x1 'convolution' over rows (discriminator.py:8-21, 186-191).  wf / wt: kernel-side copies of the
    weight for the forward / data-gradient launch (None: the raw fp32 weight on the exact-fp32 kernels); hw: the row map."""

    @staticmethod
    def forward(ctx, x, weight, bias, wf, wt, hw):
        x = x.contiguous()
        rows, ci = x.shape
        co = weight.shape[0]
        if wf is not None:
            shp = _shape(1, hw[0], hw[1], ci, co, 1, 1, 1, 0, hw[0], hw[1])         # process precision mode, no operand maxima
        else:
            shp = _shape(rows, 1, 1, ci, co, 1, 1, 1, 0, 1, 1, MODE_F32)      # raw fp32 weights: the exact-fp32 kernels
            wf = weight.detach().contiguous()
        y = torch.empty(rows, co, device=x.device, dtype=torch.float32)
        call('xas_conv_fwd', ptr(x), ptr(wf), ptr(bias), ptr(y), shp)
        ctx.save_for_backward(x, weight, *([wt] if wt is not None else []))
        ctx.shp, ctx.has_bias, ctx.rows = shp, bias is not None, rows
        # the parameters whose .grad take the sums directly (leaves only: a derived tensor has no gradient buffer)
        ctx.bias_ref = bias if (bias is not None and bias.is_leaf) else None
        ctx.weight_ref = weight if weight.is_leaf else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, *rest = ctx.saved_tensors
        shp = ctx.shp
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = rest[0] if rest else weight.detach().t().contiguous()            # [Cin][Cout] (kernel-side format)
            dx = torch.empty_like(x)
            call('xas_conv_dgrad', ptr(dy), ptr(wt), ptr(dx), shp)
        wref = ctx.weight_ref
        if ctx.needs_input_grad[1] and not (LINEAR_SIDE and dy.is_cuda and wref is not None and _wgrad_into_grad(x, dy, shp, wref)):
            dw = torch.empty_like(weight)
            ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp)), device=x.device, dtype=torch.float32)
            call('xas_conv_wgrad_oihw' if rest else 'xas_conv_wgrad', ptr(x), ptr(dy), ptr(dw), ptr(ws), shp)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            bias = ctx.bias_ref
            if not (LINEAR_SIDE and dy.is_cuda and bias is not None and _bias_into_grad(dy, ctx.rows, shp.Cout, bias, force=True)):
                db = _bias_grad(dy, ctx.rows, shp.Cout)
        return dx, dw, db, None, None, None


def linear(x, weight, bias, cache=None):
    """cache: the layer's _PackCache (layers.Linear) - with it, a layer wide enough for the MFMA tiles runs on the kernels of
    the process's precision mode (see LINEAR_MFMA); called with the PARAMETER object, whose optimizer epoch keys the cache."""
    wf = wt = hw = None
    if (cache is not None and LINEAR_MFMA and x.is_cuda and x.dim() == 2 and weight.is_contiguous() and weight.shape[1] % 32 == 0
            and weight.shape[0] >= 16 and weight.shape[0] % 4 == 0 and x.shape[0] >= 64):
        hw = _row_map(x.shape[0])
        if hw is not None:
            co, ci = weight.shape
            w4 = weight.detach().view(co, ci, 1, 1)
            w4._xas_epoch = getattr(weight, '_xas_epoch', _weights_epoch)
            shp = _shape(1, hw[0], hw[1], ci, co, 1, 1, 1, 0, hw[0], hw[1])
            wf, wt = cache.get(w4, 0, shp), cache.get(w4, 1, shp)
    return _Linear.apply(x, weight, bias, wf, wt, hw)


ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2

# Optional log of the batch statistics of every training-mode batch-norm call (engine.TrainStep(dedupe=True)
# replays the running-statistic updates of a detector forward it does not repeat).
bn_log = {'on': False, 'calls': []}


def replay_bn_updates(only=None):
    """Apply the logged running-statistic updates once more, in the logged order (`only`: data_ptr set of the
    running_mean buffers to replay; others are dropped)."""
    for mean, var, rm, rv, momentum, count, groups in bn_log['calls']:
        if only is not None and rm.data_ptr() not in only:
            continue
        call('xas_bn_update_running', ptr(mean), ptr(var), ptr(rm), ptr(rv), float(momentum), int(count), mean.shape[-1],
             int(groups))
    bn_log['calls'] = []


def _sync_stats(mean, var, count, group):
    """SyncBatchNorm statistic merge with torch ops (any device; the HIP path uses xas_bn_sync_merge on the packed
    message instead): all_gather of [mean | var | count], count-weighted merge in float64
    (torch: _functions.SyncBatchNorm.forward -> batch_norm_gather_stats_with_counts)."""
    world = dist.get_world_size(group)
    C = mean.numel()
    packed = torch.cat([mean, var, mean.new_tensor([float(count)])])
    gathered = torch.empty(world * (2 * C + 1), device=mean.device, dtype=torch.float32)
    dist.all_gather_into_tensor(gathered, packed, group=group)
    g = gathered.view(world, 2 * C + 1).double()
    cnt = g[:, 2 * C:]                                    # [world,1]
    total = cnt.sum()
    gm = (g[:, :C] * cnt).sum(0) / total
    gv = ((g[:, C:2 * C] + (g[:, :C] - gm) ** 2) * cnt).sum(0) / total
    return gm.float().contiguous(), gv.float().contiguous()


def _bn_forward(x, gamma, beta, running_mean, running_var, residual, training, momentum, eps, act, group, stats_fn=None):
    """-> (y, saved tensors (x|y, y|x, mean, var), cfg) - the body of _BatchNorm.forward, also used by _Bottleneck.
    The batch is `current_groups()` independent sub-batches (cameras): statistics are [G, C].  Prefix pass: the kernels run
    over the whole buffer (all groups); what is returned and saved are the tail views / the graph groups' statistic rows."""
    x_tail = to_cl(x)
    x = _full(x_tail)
    if x is not x_tail and not training:
        raise RuntimeError('prefix pass: training-mode norms only')
    n, c, h, w = x.shape
    G = _groups[0]
    if n % G:
        raise RuntimeError('batch of %d images does not split into %d camera groups' % (n, G))
    M = n * h * w
    Mg = M // G
    dev = x.device
    count = float(Mg)
    if training:
        if stats_fn is None:
            ws = torch.empty(query('xas_bn_workspace_floats', M, c, G), device=dev, dtype=torch.float32)

            def stats_fn(mean_p, var_p, out_stride, count_p, rm_p, rv_p, mom):
                call('xas_bn_stats', ptr(x), M, c, G, mean_p, var_p, out_stride, count_p, ptr(ws), rm_p, rv_p, float(mom), int(Mg))
        # (stats_fn given: x is still EMPTY here - the callable runs the producing convolution together with the statistics)
        if group is None:
            mean = torch.empty(G, c, device=dev, dtype=torch.float32)
            var = torch.empty(G, c, device=dev, dtype=torch.float32)
            fuse_running = running_mean is not None and not streams.forked()
            stats_fn(ptr(mean), ptr(var), c, None, ptr(running_mean) if fuse_running else None,
                     ptr(running_var) if fuse_running else None, momentum)
            if running_mean is not None and not fuse_running:
                # order-dependent update: serialised on the bookkeeping stream in host program order
                book, cur = streams.book_stream(), torch.cuda.current_stream()
                book.wait_stream(cur)
                with torch.cuda.stream(book):
                    call('xas_bn_update_running', ptr(mean), ptr(var), ptr(running_mean), ptr(running_var),
                         float(momentum), int(Mg), c, G)
                mean.record_stream(book)
                var.record_stream(book)
        else:
            # SyncBatchNorm: the stats kernel writes [mean | var | count | pad] per group straight into the message,
            # ONE all-gather per layer carries all groups, one kernel merges (count-weighted, double) and updates the
            # running statistics with the global count
            world = dist.get_world_size(group)
            stride = 2 * c + 4
            msg = torch.empty(G, stride, device=dev, dtype=torch.float32)
            stats_fn(ptr(msg), ptr(msg[:, c:]), stride, ptr(msg[:, 2 * c:]), None, None, momentum)
            gathered = torch.empty(world * G * stride, device=dev, dtype=torch.float32)     # [world][G][stride]
            from . import dp as _dp
            _dp.timed_collective(lambda: dist.all_gather_into_tensor(gathered, msg.view(-1), group=group), msg.numel() * 4)
            mean = torch.empty(G, c, device=dev, dtype=torch.float32)
            var = torch.empty(G, c, device=dev, dtype=torch.float32)
            count = float(Mg) * world                          # equal per-rank batches (train.py:274)
            if streams.forked() and running_mean is not None:
                # concurrent chains: the order-dependent running-statistic update goes to the bookkeeping stream
                call('xas_bn_sync_merge', ptr(gathered), world, G, c, stride, ptr(mean), ptr(var), None, None, float(momentum))
                book, cur = streams.book_stream(), torch.cuda.current_stream()
                book.wait_stream(cur)
                with torch.cuda.stream(book):
                    call('xas_bn_update_running', ptr(mean), ptr(var), ptr(running_mean), ptr(running_var),
                         float(momentum), int(count), c, G)
                mean.record_stream(book)
                var.record_stream(book)
            else:
                call('xas_bn_sync_merge', ptr(gathered), world, G, c, stride, ptr(mean), ptr(var),
                     ptr(running_mean), ptr(running_var), float(momentum))
        if bn_log['on'] and running_mean is not None:
            bn_log['calls'].append((mean, var, running_mean, running_var, momentum, count, G))
    else:
        mean = running_mean.reshape(1, c).expand(G, c).contiguous() if G > 1 else running_mean
        var = running_var.reshape(1, c).expand(G, c).contiguous() if G > 1 else running_var
    res = _full(to_cl(residual)) if residual is not None else None
    y = torch.empty_like(x)
    # layers with a residual (block outputs): the backward needs only the SIGN of the pre-activation value, saved as one
    # byte per float4 (1/16 of y's bytes) - neither backward pass reads y
    masked = training and residual is not None and act != ACT_NONE and os.environ.get('XAS_BN_MASK', '1') == '1'
    mask = torch.empty(M * c // 4, device=dev, dtype=torch.uint8) if masked else None
    slot = grad_amax_slot(dev) if x.is_cuda else None
    call('xas_bn_apply_amax', ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(res), float(eps), act, M, c, G, ptr(y), ptr(mask),
         ptr(slot))
    if slot is not None:
        tag_grad_amax(y, slot)                   # max |y|: the next conv scales its fp16 pieces with it
    # Backward traffic: which of x / y the backward passes need
    #   leaky ReLU, no residual : neither pass reads x (xhat recovered from the invertible output y);
    #   ReLU, no residual       : neither pass reads y (the mask is re-derived from x: 2 reads + 1 write in the apply
    #                             pass instead of 3 + 1);
    #   otherwise               : both.
    xfree = training and act == ACT_LEAKY and residual is None
    yfree = training and act == ACT_RELU and residual is None and os.environ.get('XAS_BN_YFREE', '1') == '1'
    if x is not x_tail:
        # hand on / keep the graph images only: tail views of the buffers, the statistic rows of the graph groups
        k = _prefix[1]
        y = _tail(y)
        if slot is not None:
            tag_grad_amax(y, slot)               # (the maximum over the whole buffer bounds the tail's)
        mask = _tail(mask, per_image=h * w * c // 4) if masked else None
        M, G = M - _prefix[0] * h * w, G - k
        mean, var = mean[k:], var[k:]
        x = x_tail
    saved = (y if xfree else x, mask if masked else (x if yfree else y), mean, var)
    cfg = (M, c, float(eps), act, count, group, training, residual is not None, xfree, yfree, G, masked)
    return y, saved, cfg


def _bn_backward(saved, cfg, gamma, beta, dy, want_param_grads, want_dres=True):
    """-> (dx, dgamma, dbeta, dres); dgamma / dbeta are None when they were accumulated straight into gamma.grad /
    beta.grad (LOCAL sums: the gradient all-reduce averages them later), which saves two autograd accumulation kernels
    per layer.  want_dres=False (sign-mask layers only): the residual gradient is not materialised - the consumer forms
    it from (dy, mask) itself (xas_conv_dgrad_acc_masked)."""
    x, y, mean, var = saved
    M, c, eps, act, count, group, training, has_res, xfree, yfree, G, masked = cfg
    if not training:
        # eval-mode norm inside a graph that is differentiated (frozen statistics: an affine map per channel).  Off the
        # training path - a handful of torch ops instead of kernels of its own.
        dy = to_cl(dy)
        n = dy.shape[0]
        shp = (G, 1, c, 1, 1)
        grp = lambda t: t.reshape(G, n // G, c, *t.shape[2:])
        gz = grp(dy)
        if act != ACT_NONE:
            pos = grp(y) > 0
            gz = torch.where(pos, gz, gz * (0.01 if act == ACT_LEAKY else 0.0))
        rstd = torch.rsqrt(var.reshape(G, c) + eps).reshape(shp)
        xhat = (grp(x) - mean.reshape(shp)) * rstd
        dx = (gz * (gamma.reshape(1, 1, c, 1, 1) * rstd)).reshape(dy.shape).contiguous(memory_format=CL)
        dgamma = (gz * xhat).sum(dim=(0, 1, 3, 4)) if want_param_grads else None
        dbeta = gz.sum(dim=(0, 1, 3, 4)) if want_param_grads else None
        dres = gz.reshape(dy.shape).contiguous(memory_format=CL) if has_res else None
        return dx, dgamma, dbeta, dres
    dy = to_cl(dy)
    dev = x.device
    mask = y if masked else None
    sums = torch.empty(G, 2, c, device=dev, dtype=torch.float32)          # [g][0] = sum dz, [g][1] = sum dz * xhat
    ws = torch.empty(query('xas_bn_workspace_floats', M, c, G), device=dev, dtype=torch.float32)
    gg, gb = gamma.grad, beta.grad
    direct = (want_param_grads and gg is not None and gb is not None and gg.is_contiguous() and gb.is_contiguous()
              and gg.dtype == torch.float32 and gb.dtype == torch.float32)
    px = None if xfree else ptr(x)
    py = None if (yfree or masked) else ptr(y)
    call('xas_bn_bwd_reduce', px, py, ptr(dy), ptr(mean), ptr(var), ptr(gamma),
         ptr(beta), eps, act, M, c, G, ptr(sums), ptr(ws), ptr(gb) if direct else None, ptr(gg) if direct else None, ptr(mask))
    if direct:
        dgamma = dbeta = None
        grad_ready(gamma)
        grad_ready(beta)
    else:
        dgamma, dbeta = sums[:, 1].sum(0), sums[:, 0].sum(0)              # local sums (before the exchange)
    if group is not None:
        from . import dp as _dp
        _dp.timed_collective(lambda: dist.all_reduce(sums, group=group), sums.numel() * 4)     # one coalesced message per layer (all groups)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if (has_res and (want_dres or not masked)) else None
    slot = grad_amax_slot(dev)
    call('xas_bn_bwd_apply_amax', px, py, ptr(dy), ptr(mean), ptr(var), ptr(gamma),
         ptr(beta), ptr(sums), eps, act, M, c, G, float(count), ptr(dx), ptr(dres), ptr(mask), ptr(slot))
    if slot is not None:
        tag_grad_amax(dx, slot)                  # max |dx|: the conv backward that reads dx scales its fp16 pieces with it
    return dx, dgamma, dbeta, dres


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, training, momentum, eps, act, group):
        y, saved, cfg = _bn_forward(x, gamma, beta, running_mean, running_var, residual, training, momentum, eps, act, group)
        ctx.save_for_backward(*saved, gamma, beta)
        ctx.cfg = cfg
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            note_use(gamma)
            note_use(beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        *saved, gamma, beta = ctx.saved_tensors
        dx, dgamma, dbeta, dres = _bn_backward(tuple(saved), ctx.cfg, gamma, beta, dy,
                                               ctx.needs_input_grad[1] and ctx.needs_input_grad[2])
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, residual=None, training=True, momentum=0.1, eps=1e-5,
               act=ACT_NONE, group=None):
    return _BatchNorm.apply(x, gamma, beta, running_mean, running_var, residual, training, momentum, eps, act, group)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_cl(x)
        n, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        xf = _full(x)
        nf = xf.shape[0]
        yf = empty_cl(nf, c, ho, wo, x)
        idxf = torch.empty(nf * ho * wo * c, device=x.device, dtype=torch.int8)
        call('xas_maxpool3x3s2_fwd', ptr(xf), nf, h, w, c, ptr(yf), ptr(idxf))
        y, idx = _tail(yf), _tail(idxf, per_image=ho * wo * c)
        pass_amax(y, x)
        ctx.save_for_backward(idx)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        dy = to_cl(dy)
        dx = empty_cl(n, c, h, w, dy)
        call('xas_maxpool3x3s2_bwd', ptr(dy), ptr(idx), n, h, w, c, ptr(dx))
        return dx


def maxpool3x3s2(x):
    return _MaxPool.apply(x)


class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if _prefix[0]:
            raise RuntimeError('prefix pass: up-sampling is not on the detector path')
        x = to_cl(x)
        n, c, h, w = x.shape
        y = empty_cl(n, c, 2 * h, 2 * w, x)
        call('xas_upsample2x_fwd', ptr(x), n, h, w, c, ptr(y))
        pass_amax(y, x)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        dy = to_cl(dy)
        dx = empty_cl(n, c, h, w, dy)
        call('xas_upsample2x_bwd', ptr(dy), n, h, w, c, ptr(dx))
        return dx


def upsample2x(x):
    return _Upsample2x.apply(x)


class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous() if x.dim() != 4 else to_cl(x)
        y = torch.empty_like(x)
        call('xas_sigmoid_fwd', ptr(x), x.numel(), ptr(y))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous() if dy.dim() != 4 else to_cl(dy)
        dx = torch.empty_like(y)
        call('xas_sigmoid_bwd', ptr(y), ptr(dy), y.numel(), ptr(dx))
        return dx


def sigmoid(x):
    return _Sigmoid.apply(x)


BATCH_PREP = os.environ.get('XAS_BATCH_PREP', '1') != '0'


def _conv_entries(module):
    """(conv module, cache, weight, dummy shape) of every conv under `module` that keeps kernel-side weight copies."""
    out = []
    for m in module.modules():
        cache = getattr(m, '_cache', None)
        w = getattr(m, 'weight', None)
        if isinstance(cache, _PackCache) and w is not None and w.dim() == 4 and w.is_cuda:
            a, b, r, s = w.shape
            stride, pad = int(getattr(m, 'stride', 1)), int(getattr(m, 'padding', 0))
            if type(m).__name__ == 'ConvTranspose2d':          # weight [Cin_t, Cout_t, R, S]: the equivalent conv is Cout_t -> Cin_t
                hs = 8
                hb = (hs - 1) * stride - 2 * pad + r
                shp = _shape(1, hb, hb, b, a, r, s, stride, pad, hs, hs)
            else:
                hi = 16
                ho = (hi + 2 * pad - r) // stride + 1
                shp = _shape(1, hi, hi, b, a, r, s, stride, pad, ho, ho)
            out.append((m, cache, w, shp))
        elif isinstance(cache, _PackCache) and w is not None and w.dim() == 2 and w.is_cuda and LINEAR_MFMA and w.is_contiguous() \
                and w.shape[1] % 32 == 0 and w.shape[0] >= 16 and w.shape[0] % 4 == 0:
            # a linear layer wide enough for the MFMA tiles (ops_nn.linear): its weight as a 1x1 filter; its launches never come
            # with operand maxima, so both passes take the format of the shape WITHOUT maxima (f16x3 mode: three bf16 planes)
            co, ci = w.shape
            w4 = w.detach().view(co, ci, 1, 1)
            w4._xas_epoch = getattr(w, '_xas_epoch', _weights_epoch)
            shp = _shape(1, 8, 32, ci, co, 1, 1, 1, 0, 8, 32)
            shp._linear = True
            out.append((m, cache, w4, shp))
    return out


def prepack(module):
    """Build the kernel-side weight copies of every conv under `module` on the CURRENT stream - packed fp32 and, in the
    bf16 modes, the pre-split planes - before streams fork (cameras / chains), so no two streams race to fill a cache and
    every stream finds the copies complete.  The format only depends on the filter geometry and the precision, not on the
    activation size: a small dummy shape asks the library.  The pre-split planes of ALL layers are written by ONE launch
    (xas_prepare_weights: a descriptor table built once per module and precision) into one persistent buffer."""
    entries = _conv_entries(module)
    if not entries:
        return
    prec = query('xas_get_precision')
    if not BATCH_PREP:
        for _, cache, w, shp in entries:
            if getattr(shp, '_linear', False):
                cache.get(w, 0, shp)
                cache.get(w, 1, shp)
                continue
            f16 = (prec == _lib.PREC_F16X3 and GRAD_F16 and shp.mode == 0 and query('xas_conv_weight_planes', shp, 1) == 3)
            if not (f16 and query('xas_conv_weight_planes', shp, 0) == 3):
                cache.get(w, 0, shp)                   # (f16x3: the three-plane forward format is built on demand only)
            cache.get(w, 1, shp)
            if f16:
                cache.get(w, 0, shp, planes=2)         # the formats of launches that come with their operand maxima
                cache.get(w, 1, shp, planes=2)
        return
    key = (prec, tuple(w.data_ptr() for _, _, w, _ in entries))
    tab = getattr(module, '_xas_prep', None)
    if tab is None or tab['key'] != key:
        rows_of, views, desc, blk, off = [], [], [], 0, 0
        sizes = []
        for idx, (_, cache, w, shp) in enumerate(entries):
            lin = getattr(shp, '_linear', False)
            f16 = prec == _lib.PREC_F16X3 and GRAD_F16 and shp.mode == 0 and not lin
            for t, extra in ((0, False), (1, False), (0, True), (1, True)):
                planes = query('xas_conv_weight_planes', shp, t)      # (dummy shape without operand maxima)
                if lin and extra:
                    continue
                if extra:                      # f16x3: launches that come with the maxima of their operands run on two fp16
                    if not (planes == 3 and f16):                      # planes - every forward, every data gradient whose
                        continue                                       # dy was tagged
                    planes = 2
                elif t == 0 and planes == 3 and f16:
                    continue                   # forward launches always come with max |x| (act_amax): three planes only on demand
                                               # (linear layers: never - their three-plane forward format is in the table)
                if not planes or not w.is_contiguous():
                    continue
                co, ci, r, s = w.shape
                rows = ci if t else co
                kk = w.numel() // rows
                nbytes = query('xas_split_weight_bytes', rows, kk, planes)
                nblk = (((rows + 31) // 32) * (kk // 16) * 64 + 255) // 256
                sizes.append((idx, t, planes, off, nbytes))
                desc.append([w.data_ptr(), off, co, ci, r, s, t, planes, rows, kk, blk, 0])
                off += (nbytes + 255) // 256 * 256
                blk += nblk
        buf = torch.empty(max(1, off), device=entries[0][2].device, dtype=torch.uint8)
        for d in desc:
            d[1] += buf.data_ptr()
        table = torch.tensor(desc, dtype=torch.int64).to(buf.device) if desc else None
        tab = {'key': key, 'buf': buf, 'table': table, 'blocks': blk, 'n': len(desc),
               'views': [(idx, t, planes, buf[o:o + nb]) for idx, t, planes, o, nb in sizes], 'epoch': None}
        module._xas_prep = tab
    # fresh for this parameter state?  (prepack runs at the head of every pass; the caches know their epoch)
    stamp = tuple((w._version, getattr(w, '_xas_epoch', _weights_epoch)[0]) for _, _, w, _ in entries)
    if tab['epoch'] != stamp:
        if tab['n']:
            call('xas_prepare_weights', ptr(tab['table']), tab['n'], tab['blocks'])
        tab['epoch'] = stamp
    for idx, t, planes, view in tab['views']:
        _, cache, w, _ = entries[idx]
        cache._fresh(w)
        cache.packed[(t, planes)] = view
    for _, cache, w, shp in entries:                   # whatever the table does not cover: the fp32-packed formats (a split
        for t in (0, 1):                               # format the table leaves out - f16x3: three-plane forward weights, only
            if query('xas_conv_weight_planes', shp, t) == 0 or not w.is_contiguous():      # wanted by a launch that comes
                cache.get(w, t, shp)                   # without the maximum of its input - is built on demand)

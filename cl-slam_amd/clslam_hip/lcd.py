"""HIP schedule of the loop-closure feature encoder (MobileNetV3-small forward, reference
loop_closure_detection/encoder.py:13-33).  Weights come in torchvision's state-dict layout
(``features.N...``); eval BatchNorm (eps 1e-3) is folded to scale/shift, channel counts are
zero-padded to multiples of 16 so the 1x1 convolutions run on the MFMA conv kernel.
"""
import os
from typing import Dict, List

import torch

from . import ops
from ._lib import ACT_HSWISH, ACT_NONE, ACT_RELU, ClslamError, get_lib

# (in, kernel, expanded, out, use_se, activation, stride) -- torchvision 0.11 mobilenet_v3_small
SETTINGS = [(16, 3, 16, 16, True, 'RE', 2), (16, 3, 72, 24, False, 'RE', 2), (24, 3, 88, 24, False, 'RE', 1),
            (24, 5, 96, 40, True, 'HS', 2), (40, 5, 240, 40, True, 'HS', 1), (40, 5, 240, 40, True, 'HS', 1),
            (40, 5, 120, 48, True, 'HS', 1), (48, 5, 144, 48, True, 'HS', 1), (48, 5, 288, 96, True, 'HS', 2),
            (96, 5, 576, 96, True, 'HS', 1), (96, 5, 576, 96, True, 'HS', 1)]
BN_EPS = 1e-3
NUM_FEATURES = 576


def _p16(c: int) -> int:
    return (c + 15) // 16 * 16


def expected_keys() -> List[str]:
    keys = []

    def cba(p):
        return [p + '.0.weight'] + [p + f'.1.{k}' for k in ('weight', 'bias', 'running_mean', 'running_var')]
    keys += cba('features.0')
    for i, (cin, k, exp, cout, se, act, stride) in enumerate(SETTINGS, start=1):
        j = 0
        if exp != cin:
            keys += cba(f'features.{i}.block.{j}'); j += 1
        keys += cba(f'features.{i}.block.{j}'); j += 1
        if se:
            keys += [f'features.{i}.block.{j}.fc1.weight', f'features.{i}.block.{j}.fc1.bias',
                     f'features.{i}.block.{j}.fc2.weight', f'features.{i}.block.{j}.fc2.bias']
            j += 1
        keys += cba(f'features.{i}.block.{j}')
    keys += cba('features.12')
    return keys


def _make_divisible(v: float, divisor: int = 8) -> int:
    """torchvision's channel rounding (squeeze channels of the SE blocks = _make_divisible(expanded // 4, 8))."""
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def state_dict_shapes() -> Dict[str, tuple]:
    """Key -> shape of the `features.*` part of torchvision's mobilenet_v3_small state dict (what the reference's
    FeatureEncoder evaluates, loop_closure_detection/encoder.py:22-26)."""
    shapes: Dict[str, tuple] = {}

    def cba(p, cout, cin_per_group, k):
        shapes[p + '.0.weight'] = (cout, cin_per_group, k, k)
        for name in ('weight', 'bias', 'running_mean', 'running_var'):
            shapes[p + '.1.' + name] = (cout,)
    cba('features.0', 16, 3, 3)
    for i, (cin, k, exp, cout, se, act, stride) in enumerate(SETTINGS, start=1):
        j = 0
        if exp != cin:
            cba(f'features.{i}.block.{j}', exp, cin, 1); j += 1
        cba(f'features.{i}.block.{j}', exp, 1, k); j += 1
        if se:
            S = _make_divisible(exp // 4, 8)
            p = f'features.{i}.block.{j}'
            shapes[p + '.fc1.weight'], shapes[p + '.fc1.bias'] = (S, exp, 1, 1), (S,)
            shapes[p + '.fc2.weight'], shapes[p + '.fc2.bias'] = (exp, S, 1, 1), (exp,)
            j += 1
        cba(f'features.{i}.block.{j}', cout, exp, 1)
    cba('features.12', 576, 96, 1)
    return shapes


def synthetic_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Closed-form random-init weights of the architecture (bench.py --lcd and the tests: the ImageNet checkpoint the reference
    downloads is not available offline).  Same generator as the depth / pose nets' synthetic weights (clslam_hip.synth)."""
    from . import synth
    out = {}
    for k, shape in state_dict_shapes().items():
        n = 1
        for d in shape:
            n *= d
        u = torch.from_numpy(synth.hash_uniform(max(n, 1), synth._key_seed('lcd/' + k, seed))[:n]).reshape(shape)
        if len(shape) == 4:
            out[k] = (u * 2 - 1) * (6.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
        elif k.endswith('running_var'):
            out[k] = 0.6 + 0.8 * u
        elif k.endswith('running_mean'):
            out[k] = 0.2 * (u - 0.5)
        elif '.1.weight' in k:
            out[k] = 0.8 + 0.4 * u
        else:
            out[k] = 0.2 * (u - 0.5)
    return out


class MobileNetV3SmallHIP:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device: torch.device) -> None:
        lib = get_lib()
        if device.type != lib.device_type:
            raise ClslamError(f'LCD encoder on {device} but the library executes on {lib.device_type}')
        missing = [k for k in expected_keys() if k not in state_dict]
        if missing:
            raise ClslamError(f'MobileNetV3-small state dict is missing {len(missing)} keys, e.g. {missing[:3]}')
        self.device = device
        sd = {k: v.detach().to(device, torch.float32) for k, v in state_dict.items() if k.startswith('features.')}

        def bn(p, cpad):
            scale = sd[p + '.weight'] / torch.sqrt(sd[p + '.running_var'] + BN_EPS)
            shift = sd[p + '.bias'] - sd[p + '.running_mean'] * scale
            s = torch.zeros(cpad, device=device); t = torch.zeros(cpad, device=device)
            s[:scale.numel()] = scale; t[:shift.numel()] = shift
            return s, t

        def pw(p, cin, cout):  # 1x1 conv weight (O,I,1,1) -> zero-padded [Op][1][Ip]
            w = torch.zeros(_p16(cout), 1, _p16(cin), device=device)
            w[:cout, 0, :cin] = sd[p + '.0.weight'].view(cout, cin)
            return (w.contiguous(),) + bn(p + '.1', _p16(cout))

        def dw(p, c, k):  # depthwise (C,1,k,k) -> [k*k][Cp]
            w = torch.zeros(k * k, _p16(c), device=device)
            w[:, :c] = sd[p + '.0.weight'].view(c, k * k).t()
            return (w.contiguous(),) + bn(p + '.1', _p16(c))

        self.stem_w = sd['features.0.0.weight'].contiguous()
        self.stem_s, self.stem_t = bn('features.0.1', 16)
        self.blocks = []
        for i, (cin, k, exp, cout, se, act, stride) in enumerate(SETTINGS, start=1):
            blk = dict(cin=cin, k=k, exp=exp, cout=cout, stride=stride, act=ACT_RELU if act == 'RE' else ACT_HSWISH,
                       res=(stride == 1 and cin == cout))
            j = 0
            blk['expand'] = None
            if exp != cin:
                blk['expand'] = pw(f'features.{i}.block.{j}', cin, exp); j += 1
            blk['dw'] = dw(f'features.{i}.block.{j}', exp, k); j += 1
            blk['se'] = None
            if se:
                p = f'features.{i}.block.{j}'
                S = sd[p + '.fc1.weight'].shape[0]
                w1 = torch.zeros(S, _p16(exp), device=device); w1[:, :exp] = sd[p + '.fc1.weight'].view(S, exp)
                w2 = torch.zeros(_p16(exp), S, device=device); w2[:exp] = sd[p + '.fc2.weight'].view(exp, S)
                b2 = torch.zeros(_p16(exp), device=device); b2[:exp] = sd[p + '.fc2.bias']
                blk['se'] = (w1.contiguous(), sd[p + '.fc1.bias'].contiguous(), w2.contiguous(), b2)
                j += 1
            blk['project'] = pw(f'features.{i}.block.{j}', exp, cout)
            self.blocks.append(blk)
        self.head = pw('features.12', 96, 576)
        self._bufs = {}
        self._graphs = {}

    def _buf(self, key, *shape):
        t = self._bufs.get((key,) + shape)
        if t is None:
            t = torch.empty(*shape, device=self.device)
            self._bufs[(key,) + shape] = t
        return t

    def __call__(self, image: torch.Tensor) -> torch.Tensor:
        """image (B,3,H,W) in [0,1] (un-normalised, like the reference passes it) -> (B,576).

        On the GPU the 52 launches of a forward are replayed as one hipGraph per input shape (B=1 per frame:
        the eager forward is launch-latency-bound, 0.78 ms at 192x640 vs the graph replay); CLSLAM_HIPGRAPH=0
        disables it."""
        x = image.to(self.device, torch.float32).contiguous()
        if self.device.type != 'cuda' or os.environ.get('CLSLAM_HIPGRAPH', 'auto') == '0':
            return self._forward(x)
        st = self._graphs.get(tuple(x.shape))
        if st is None:
            st = {'x': x.clone()}
            cur = torch.cuda.current_stream(self.device)
            warm = torch.cuda.Stream(device=self.device)
            warm.wait_stream(cur)
            with torch.cuda.stream(warm):        # eager warm-up allocates every activation buffer
                self._forward(st['x'])
            cur.wait_stream(warm)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st['out'] = self._forward(st['x'])
            st['graph'] = g
            self._graphs[tuple(x.shape)] = st
        st['x'].copy_(x, non_blocking=True)
        st['graph'].replay()
        return st['out'].clone()

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        B, _, H, W = x.shape
        h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        cur = self._buf('stem', B, h, w, 16)
        ops.mbv3_stem(x, self.stem_w, self.stem_s, self.stem_t, cur)
        for bi, blk in enumerate(self.blocks):
            inp = cur
            ep, cp = _p16(blk['exp']), _p16(blk['cout'])
            if blk['expand'] is not None:
                wgt, s, t = blk['expand']
                y = self._buf(f'e{bi}', B, h, w, ep)
                ops.conv2d(cur, wgt, y, scale=s, shift=t, ksize=1, pad=0, act=blk['act'])
                cur = y
            k, st = blk['k'], blk['stride']
            h2, w2 = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
            wgt, s, t = blk['dw']
            y = self._buf(f'd{bi}', B, h2, w2, ep)
            ops.dwconv(cur, wgt, s, t, y, k, st, blk['act'])
            cur, h, w = y, h2, w2
            if blk['se'] is not None:
                w1, b1, wv2, b2 = blk['se']
                pool = self._buf(f'p{bi}', B, ep)
                gate = self._buf(f'g{bi}', B, ep)
                ops.global_avgpool(cur, pool, self._buf(f'pp{bi}', B * ops.avgpool_chunks(h * w) * ep))
                ops.se_gate(pool, w1, b1, wv2, b2, gate)
                ops.channel_scale(cur, gate)
            wgt, s, t = blk['project']
            y = self._buf(f'o{bi}', B, h, w, cp)
            ops.conv2d(cur, wgt, y, scale=s, shift=t, residual=inp if blk['res'] else None, ksize=1, pad=0, act=ACT_NONE)
            cur = y
        wgt, s, t = self.head
        y = self._buf('head', B, h, w, 576)
        ops.conv2d(cur, wgt, y, scale=s, shift=t, ksize=1, pad=0, act=ACT_HSWISH)
        feat = torch.empty(B, 576, device=self.device)
        ops.global_avgpool(y, feat, self._buf('pph', B * ops.avgpool_chunks(h * w) * 576))
        return feat

"""The per-frame schedule of the MI355X-native depth/pose path.

One ``Engine`` owns, in HBM:
  * the frozen encoder weights, re-laid out once per ``load`` (OHWI conv weights, eval-mode
    BatchNorm folded into per-channel scale/shift -- legal because adaptation never changes encoder
    or BN parameters, dpp.py:308,813-819),
  * ONE flat fp32 arena for the 36 trainable tensors (depth decoder + pose decoder, 4.47 M floats)
    and three more of the same layout for gradients and Adam's two moments -- so the data-parallel
    gradient exchange is a single all-reduce and Adam a single kernel,
  * a workspace of NHWC activations and loss-stage planes, planned once per batch size (every shape
    is static given B, H, W).
and sequences the C-ABI kernels (clslam_hip.ops) for: encoder/decoder/pose forward -> view synthesis
+ loss -> hand-written backward of everything that has gradients -> fused Adam.  No torch autograd,
no torch.nn compute: torch supplies device memory, streams and torch.distributed only.

Reference call graph being replaced: dpp.py:906-923 (_process_batch), :925-974, :976-1017,
:1019-1120, and dpp.py:309-313 (zero_grad / backward / optimizer.step).
"""
import math
import os
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import ops
from ._lib import ACT_ELU, ACT_NONE, ACT_RELU, PAD_REFLECT, PAD_ZERO, ClslamError, get_lib

NUM_CH_ENC = (64, 64, 128, 256, 512)
NUM_CH_DEC = (16, 32, 64, 128, 256)
BN_EPS = 1e-5

LOSS_NAMES = ('reprojection_loss', 'smooth_loss', 'reg_loss', 'depth_loss')


def _align4(n: int) -> int:
    return (n + 3) // 4 * 4


class TrainableLayout:
    """Flat arena layout of the trainable tensors in compute order; names are
    '<model>/<state-dict key>' in the reference's parameter order."""

    def __init__(self) -> None:
        self.entries: List[Tuple[str, int, Tuple[int, ...]]] = []  # (name, offset, reference shape)
        self.offset: Dict[str, int] = {}
        off = 0

        def add(name: str, shape: Tuple[int, ...]) -> None:
            nonlocal off
            self.entries.append((name, off, shape))
            self.offset[name] = off
            off = _align4(off + math.prod(shape))

        for i in range(4, -1, -1):
            cin0 = NUM_CH_ENC[-1] if i == 4 else NUM_CH_DEC[i + 1]
            add(f'depth_decoder/upconv_{i}_0.conv.conv.weight', (NUM_CH_DEC[i], cin0, 3, 3))
            add(f'depth_decoder/upconv_{i}_0.conv.conv.bias', (NUM_CH_DEC[i],))
            cin1 = NUM_CH_DEC[i] + (NUM_CH_ENC[i - 1] if i > 0 else 0)
            add(f'depth_decoder/upconv_{i}_1.conv.conv.weight', (NUM_CH_DEC[i], cin1, 3, 3))
            add(f'depth_decoder/upconv_{i}_1.conv.conv.bias', (NUM_CH_DEC[i],))
        for s in range(4):
            # weight immediately followed by its bias (clslam_dispconv_wgrad reduces both at once)
            name = f'depth_decoder/dispconv_{s}.conv.weight'
            self.entries.append((name, off, (1, NUM_CH_DEC[s], 3, 3)))
            self.offset[name] = off
            off += 9 * NUM_CH_DEC[s]
            name = f'depth_decoder/dispconv_{s}.conv.bias'
            self.entries.append((name, off, (1,)))
            self.offset[name] = off
            off = _align4(off + 1)
        add('pose_decoder/squeeze.weight', (256, 512, 1, 1))
        add('pose_decoder/squeeze.bias', (256,))
        for k in (0, 1):
            add(f'pose_decoder/pose_{k}.weight', (256, 256, 3, 3))
            add(f'pose_decoder/pose_{k}.bias', (256,))
        add('pose_decoder/pose_2.weight', (12, 256, 1, 1))
        add('pose_decoder/pose_2.bias', (12,))
        self.size = off

    @staticmethod
    def to_compute(t: torch.Tensor) -> torch.Tensor:
        """reference OIHW -> OHWI ([Cout][taps][Cin]); 1-D tensors unchanged."""
        return t.permute(0, 2, 3, 1).reshape(-1) if t.dim() == 4 else t.reshape(-1)

    @staticmethod
    def to_reference(flat: torch.Tensor, shape: Tuple[int, ...]) -> torch.Tensor:
        if len(shape) == 4:
            o, i, kh, kw = shape
            return flat.view(o, kh, kw, i).permute(0, 3, 1, 2)
        return flat.view(shape)


_STREAM_POOL: Dict[Any, Dict[str, Any]] = {}


def _device_streams(device) -> Dict[str, Any]:
    """side / wgrad / capture / tail streams of this process on `device` (created once)."""
    if device.type != 'cuda':
        return {}
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    pool = _STREAM_POOL.get(key)
    if pool is None:
        # CLSLAM_PRIO_<NAME> = -1 (high) / 0 / 1 (low): HIP queue priority of a stream (experiments, profiles/r05_stream_priority.txt)
        pool = {name: torch.cuda.Stream(device=device, priority=int(os.environ.get('CLSLAM_PRIO_' + name.upper(), '0')))
                for name in ('main', 'side', 'wg', 'tail', 'capture')}
        _STREAM_POOL[key] = pool
    return pool


class Engine:
    def __init__(self, height: int, width: int, device: torch.device, *, min_depth: Optional[float],
                 max_depth: Optional[float], disparity_smoothness: float, velocity_loss_scaling: Optional[float],
                 reference_quirks: bool = True) -> None:
        lib = get_lib()  # raises if libclslam_hip.so is missing -- there is no fallback
        if device.type != lib.device_type:
            raise ClslamError(f'engine on {device} but {lib.path.name} executes on {lib.device_type}; the MI355X path '
                              'needs a GPU and has no CPU fallback')
        if height % 32 or width % 32:
            raise ValueError('height and width must be multiples of 32 (five stride-2 stages)')
        # reference_quirks=False: the per-sample edge-aware smoothness the reference evidently meant instead of
        # the flattened-batch behaviour of dpp.py:1148-1176; opt-in, parity runs use the default
        self.smooth_intended = not reference_quirks
        if min_depth is None and max_depth is not None:
            raise ValueError('min_depth is None')        # disp_to_depth, utils.py:134-135
        self.H, self.W, self.device = height, width, device
        self.min_depth, self.max_depth = min_depth, max_depth
        self.smooth_scale = float(disparity_smoothness)
        self.vel_scale = float(velocity_loss_scaling) if velocity_loss_scaling else 0.0
        self.layout = TrainableLayout()
        n = self.layout.size
        # the four arenas; `engine.w / g / m / v` (properties below) are the same tensors for OUTSIDE readers: the access
        # orders the current stream behind the training step still in flight on the engine's own stream
        self._w = torch.zeros(n, device=device)
        self._wb_cache: Dict[str, Any] = {}
        self._desc_cache: Dict[Any, Any] = {}     # conv descriptors of this engine's fixed call sites (ops.conv2d(cache=...))
        self._main = None
        self._g = torch.zeros(n, device=device)
        self._m = torch.zeros(n, device=device)
        self._v = torch.zeros(n, device=device)
        self.adam_step_count = 0
        self.enc: Dict[str, Any] = {}
        self._ws: Dict[int, SimpleNamespace] = {}
        self._graphs: Dict[Any, SimpleNamespace] = {}
        self.models = None
        self._packed_version = None
        self._tracked = None
        self._modules_stale = False
        self.fresh_outputs = True
        # the depth net and the pose net are independent until the loss stage (and their backward passes
        # after it): run the pose branch on a second HIP stream so its small 6x20 layers fill the CUs the
        # depth branch leaves idle at kernel tails
        # (the streams are shared by every engine of the process on this device: HIP spreads streams over four hardware
        # queues, and a third predictor's twelfth stream lands on the queue of its own main stream -- 3.84 instead of
        # 3.33 ms per step measured with three predictors alive)
        pool = _device_streams(device)
        # The engine's own in-order compute stream.  A training adapt() runs there: forward, backward and the optimizer step
        # are enqueued behind each other, the CALLER's stream is only ordered behind the forward and the first three launches
        # of the backward (`inputs_released`: the last reads of the caller's minibatch and of the output planes), so what the
        # reference's callers do next -- outputs['cam_T_cam', 0, 1][0].cpu(), losses[k].cpu() (slam.py:181-188) -- returns
        # while the rest of the backward + Adam keep the GPU busy; whatever touches trainable state or the workspace
        # afterwards is ordered behind the step by wait_training().  CLSLAM_DETACHED_TRAINING=0: everything on the caller's
        # stream like rounds 1-3 (bitwise the same results, tests/test_detached_training.py).
        self.main_stream = pool.get('main')
        self.detached_training = os.environ.get('CLSLAM_DETACHED_TRAINING', '1') != '0'
        self._train_done = None       # event: the last detached training step (incl. its optimizer step) has completed
        self._caller = None           # the caller's stream while a detached call is being enqueued
        self.inputs_released = None
        self._prealloc = None         # output planes allocated by the caller's stream for the call's last forward (prealloc_outputs)
        self._prealloc_armed = False
        self.side_stream = pool.get('side')
        self.wg_stream = pool.get('wg')
        # split-K scratch of the small-M 3x3 convs: one zero-filled buffer per stream convs are launched on
        self._conv_ws: Dict[int, torch.Tensor] = pool.setdefault('conv_ws', {}) if pool else {}
        self.capture_stream = pool.get('capture')
        for st in (self.main_stream, self.side_stream, self.wg_stream, self.capture_stream):
            if st is not None:
                self._conv_workspace(st.cuda_stream)
        # Asynchronous tail (data-parallel mode): gradient reduction + all-reduce + Adam of step N run on their own
        # stream; the caller's stream does not wait for them, so the frozen encoders of step N+1 overlap the
        # xGMI all-reduce.  Whatever reads trainable state waits on the event (wait_training()).
        self.async_tail = False
        self.data_parallel = False    # set by DepthPosePrediction.enable_data_parallel: an all-reduce sits between reduction and Adam
        # single-GPU path, opt-in (CLSLAM_FUSE_ADAM=1): the gradient reduction and the optimizer step as ONE launch
        # (clslam_reduce_multi_adam; backward(defer_reduce=True) leaves the reduction to adam()).  Bitwise the two launches
        # (tests/test_adam.py) but not faster on the MI355X: 3.316 vs 3.311 ms at B = 5, 1.515 vs 1.454 ms at K = 0 -- in the
        # reduction only one lane in KL (4...64 for the many-split layers) holds a finished sum, so the update's six arena
        # accesses per element run at a fraction of the Adam kernel's width; the launch saved is worth less than that.
        self.fuse_adam = os.environ.get('CLSLAM_FUSE_ADAM', '0') == '1'
        # slam.py:143-147 runs models['depth_encoder'](online_image) on every frame BEFORE adapt() /
        # adapt(online, None), whose own depth-encoder pass over the online sample repeats exactly that computation (frozen
        # encoder, eval-mode BatchNorm, and the online sample is not augmented: rgb_aug == rgb, datasets/utils.py:25,148-150).
        # run_encoder() keeps its input and features; a single-triplet forward whose network input has the same CONTENT
        # takes them instead of re-running the 22 launches of the depth encoder.
        self.descriptor_memo = os.environ.get('CLSLAM_DESCRIPTOR_MEMO', '1') != '0'
        self._memo = None
        self.memo_hits = 0
        self._pending_reduce = None
        # Data-parallel gradient exchange in BUCKETS: the flat arena is cut into three contiguous
        # ranges in the order their gradients complete in the backward -- the pose decoder (its short backward runs beside the
        # depth decoder's), the depth decoder's levels 3..0 + disparity heads, and last the two 512/256-channel convolutions
        # of level 4 (53 % of the payload: they are the END of the data-gradient chain).  Each range is reduced from its
        # partials and all-reduced on the tail stream as soon as its last weight-gradient kernel has been enqueued, so all but
        # the last all-reduce run underneath the rest of the backward; Adam follows once.  The sequence of collectives
        # (three all-reduces, same slices, same order) is the same on every rank whatever its workspace / overlap state.
        self.grad_buckets = int(os.environ.get('CLSLAM_GRAD_BUCKETS', '3'))
        self.grads_synced = False
        self.time_exchange = False    # bench.py --gpus N: time the bucket all-reduces (exchange_report())
        self.exchange_events: List[Any] = []
        self.exchange_main_done = None
        self.tail_stream = pool.get('tail')
        self._tail_event = None       # optimizer step in flight on tail_stream
        self._tail_open = False       # backward() left its reduction on tail_stream; adam() closes it
        self._capturing = False
        self.depth_first = os.environ.get('CLSLAM_DEPTH_FIRST', '1') != '0'
        # tie-break noise drawn in the loss kernel (Philox): keyed by the seed of torch's generator of this device, re-read on
        # every forward -- torch.manual_seed() after construction re-keys the stream and restarts its draw counter, like the
        # global generator the reference draws from (dpp.py:1055-1056).  noise_stream = the data-parallel shard offset: ranks
        # seeded identically still draw different fields for their different samples.
        self._noise_torch_seed = None
        self.noise_seed = 1
        self.noise_draws = 0
        self.noise_stream = 0
        self.use_side_stream = os.environ.get('CLSLAM_SIDE_STREAM', '1') != '0'
        # Releasing a side stream behind a launch of the dependency chain: torch.cuda.Event().record(main) puts a marker packet
        # between two kernels of the chain (+5.0 us on the producer per hand-off, ~17 per step); an event that is the producing
        # launch's own completion signal costs +1.3 us (tools/micro/event_gap.hip, profiles/r05_micro_calibration.txt).
        # CLSLAM_HANDOFF=0: the recorded events of rounds 1-4.
        self._handoff = (ops.Handoff() if (device.type == 'cuda' and os.environ.get('CLSLAM_HANDOFF', '1') != '0') else None)
        # clslam_conv_desc.cu_limit of the persistent (stream-K / Winograd) conv launches.  The depth and the pose network of a
        # step run on two streams; a persistent launch that takes every CU (140 KB of LDS each) shuts the other stream out
        # for its whole duration.  With half of the chip per launch the two branches run side by side and every workgroup
        # walks twice the units (the prologue / hand-off / epilogue phases of a launch are paid on half of the CUs).  Measured at
        # 192x640 (profiles/r05_cu_limit_sweep.txt): minibatch of 3 / 5 / 9 triplets -2.7 / -2.9 / -1.8 % per step; 1 triplet
        # +4 % (one triplet does not fill the chip: latency counts, not occupancy) and 33 triplets +3.7 % (every launch is long
        # enough to fill it alone) -- hence the band.  CLSLAM_CU_LIMIT=<n> forces a value (0: the whole chip).
        self.cu_limit_env = os.environ.get('CLSLAM_CU_LIMIT')
        self._cu_enc = self._cu_dec = self._cu_pose = self._cu_depth = 0     # set per forward(); the stand-alone callables (run_encoder, ...) take the whole chip
        # OPT-IN experiment (CLSLAM_EARLY_LOSS=1), steps 2..S of adapt(steps=S) (decoders only): the view synthesis, photometric
        # stage and loss backward of the COARSE scales 3, 2, 1 issued on the leaf stream as soon as their disparity head has run,
        # beside the decoder levels that follow; only scale 0's remain between the decoder and the data-gradient chain.  The
        # per-scale launches write bit for bit what the pyramid launch writes (tests/test_loss_stage.py, test_frozen_reuse.py with
        # the switch on).  Measured on MI355X, 192x640, five-step frame: B = 5 9.85 -> 9.97 ms, B = 3 7.15 -> 7.28, B = 2 5.75 -> 5.86,
        # B = 1 inside its +-20 % host noise, B = 33 +-0: the decoder's launches already fill every wave slot, the loss kernels only
        # take turns with them and nine more launches + two more events are paid for nothing.  Off by default.
        self.early_loss = os.environ.get('CLSLAM_EARLY_LOSS', '0') == '1'
        self.early_loss_max_pixels = int(os.environ.get('CLSLAM_EARLY_LOSS_MAX_PIXELS', str(1 << 40)))
        self.device_cus = (torch.cuda.get_device_properties(device).multi_processor_count if device.type == 'cuda' else 256)
        # steps 2..S of adapt(steps=S) keep the frozen encoders' features (see forward)
        self.reuse_frozen_features = os.environ.get('CLSLAM_REUSE_FROZEN', '1') != '0'
        # Winograd F(2x2,3x3) for the frozen encoders' 3x3 stride-1 convolutions (CLSLAM_NO_WINOGRAD=1: the direct kernels)
        self.winograd = not os.environ.get('CLSLAM_NO_WINOGRAD')

    # ------------------------------------------------------------------------------------------
    # parameters
    def bind(self, models: Dict[str, torch.nn.Module]) -> None:
        self.models = models
        for name, m in models.items():
            m._bind(self, name)

    def _module_version(self) -> int:
        """cheap fingerprint of the module parameters/buffers (in-place edits bump ``_version``;
        load_state_dict copies in place, so the tensor objects themselves are stable)"""
        if self._tracked is None:
            self._tracked = [t for m in self.models.values() for t in torch.nn.Module.state_dict(m, keep_vars=True).values()]
        return sum(t._version for t in self._tracked)

    def pack_if_needed(self) -> None:
        """Re-layout the module parameters into the engine's buffers when they were modified from
        outside (load_model, load_state_dict, manual edits)."""
        ver = self._module_version()
        if ver != self._packed_version:
            if self._modules_stale:
                raise ClslamError('module parameters were modified while newer adapted weights live in the engine; '
                                  'call sync_modules() first')
            self.pack()

    def _on(self, st):
        """context: the launches of the block go to stream `st`.  Outside graph capture this is ops.launch_on (no change of
        torch's current stream, 0.3 us instead of 6): the blocks it wraps contain launches of clslam_hip.ops only."""
        if st is None:
            import contextlib
            return contextlib.nullcontext()
        return torch.cuda.stream(st) if self._capturing else ops.launch_on(st)

    def _use_tail(self) -> bool:
        return (self.async_tail and self.tail_stream is not None and self.device.type == 'cuda' and not self._capturing
                and ops.PROFILE is None)

    def cu_limit(self, B: int, phase: str = 'encoders') -> int:
        """phase: 'enc_pose' / 'enc_depth' (the two frozen encoders side by side: 5/8 and 3/8 of the chip), 'decoders', 'backward' (the whole
        chip: in steps 2..S of adapt(steps=S) nothing runs beside the depth decoder's launches -- and the decoders' summation
        order must not depend on whether the encoders ran, tests/test_frozen_reuse.py holds the reuse bitwise invisible)"""
        env = os.environ.get('CLSLAM_CU_LIMIT_' + phase.upper())       # experiments
        if env is not None:
            return int(env)
        if self.cu_limit_env is not None:
            return int(self.cu_limit_env)
        if not phase.startswith('enc'):
            return 0
        if not (self.use_side_stream and 2 <= B <= 16):
            return 0
        # The pose encoder sees 2B images, the depth encoder B (and the depth decoder follows it on the same stream): 5/8 : 3/8 of
        # the chip instead of half each.  Measured at K = 4 (round 6, MI355X): pose / depth = 128/128 2.904, 144/112 2.871,
        # 160/96 2.863, 176/80 2.909, 192/64 3.002, 112/144 3.007 ms per step; whole chip each 3.027.
        if phase == 'enc_pose':
            return self.device_cus * 5 // 8
        if phase == 'enc_depth':
            return self.device_cus * 3 // 8
        return self.device_cus // 2

    def wait_training(self, stream=None) -> None:
        """Make `stream` (default: the current one) wait for the training step in flight: the optimizer step on the tail
        stream (data-parallel mode) and / or the detached backward + optimizer step on the engine's own stream."""
        if self._tail_event is not None:
            (stream or torch.cuda.current_stream(self.device)).wait_event(self._tail_event)
        if self._train_done is not None:
            (stream or torch.cuda.current_stream(self.device)).wait_event(self._train_done)

    # the arenas for outside readers (tests, tools, checkpointing, the asynchronous mode's snapshots): ordered behind the step
    @property
    def w(self) -> torch.Tensor:
        self.wait_training()
        return self._w

    @property
    def g(self) -> torch.Tensor:
        self.wait_training()
        return self._g

    @property
    def m(self) -> torch.Tensor:
        self.wait_training()
        return self._m

    @property
    def v(self) -> torch.Tensor:
        self.wait_training()
        return self._v

    def prealloc_outputs(self, B: int) -> None:
        """Called on the CALLER's stream right before a detached training call: the output planes of the call's last forward,
        allocated from the caller's pool.  The engine's stream writes them (it is ordered behind the caller at
        begin_detached()), the caller reads them behind `inputs_released` -- after which nothing on the engine's streams
        touches them any more (backward(): the loss backward's three launches are the last readers) -- and frees them."""
        H, W = self.H, self.W
        E = lambda *s: torch.empty(*s, device=self.device)  # noqa: E731
        self._prealloc = (B, ([E(B, H >> s, W >> s) for s in range(4)], E(2 * B, 12), E(2, B, 4, 4), E(4, B, H, W),
                              E(4, 2, B, 3, H, W), E(18)))

    def detached_ok(self) -> bool:
        """A training adapt() may run on the engine's own stream (see __init__)."""
        return (self.detached_training and self.main_stream is not None and self.device.type == 'cuda'
                and not self._capturing and ops.PROFILE is None)

    def begin_detached(self):
        """-> (caller's stream, engine stream): the engine stream is ordered behind everything the caller has enqueued (its
        inputs) and becomes torch's current stream for the enqueueing that follows (`with torch.cuda.stream(em)`)."""
        cur = torch.cuda.current_stream(self.device)
        em = self.main_stream
        if cur.cuda_stream == em.cuda_stream:      # nested / already there
            return cur, None
        em.wait_stream(cur)
        self._caller = cur
        return cur, em

    def end_detached(self, cur, released, failed: bool = False) -> None:
        """The caller's stream continues behind `released` (backward(): the last read of the caller's minibatch, ~0.1 ms into
        the backward); everything that touches trainable state or the workspace waits through wait_training()."""
        self._caller = None
        self._prealloc, self._prealloc_armed = None, False
        ev = torch.cuda.Event()
        ev.record(self.main_stream)
        self._train_done = ev
        cur.wait_event(ev if (failed or released is None) else released)

    def training_stream(self):
        """Context manager: the stream the gradient all-reduce of the step in flight belongs on."""
        import contextlib
        return torch.cuda.stream(self.tail_stream) if self._tail_open else contextlib.nullcontext()

    @torch.no_grad()
    def pack(self) -> None:
        dev = self.device
        self.wait_training()
        for which in ('depth_encoder', 'pose_encoder'):
            sd = {k: v.detach().to(dev, torch.float32) for k, v in torch.nn.Module.state_dict(self.models[which]).items()}
            e = SimpleNamespace()

            def bn(prefix):
                scale = sd[prefix + '.weight'] / torch.sqrt(sd[prefix + '.running_var'] + BN_EPS)
                shift = sd[prefix + '.bias'] - sd[prefix + '.running_mean'] * scale
                return scale.contiguous(), shift.contiguous()

            def conv(key):
                wt = sd[key]
                return wt.permute(0, 2, 3, 1).reshape(wt.shape[0], wt.shape[2] * wt.shape[3], wt.shape[1]).contiguous()

            e.stem_w = ops.stem_pack_weight(sd['resnet.conv1.weight'].contiguous())
            e.stem_scale, e.stem_shift = bn('resnet.bn1')
            e.blocks = []
            for li, (cin, cout, stride) in enumerate(((64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2)), start=1):
                for bi in range(2):
                    p = f'resnet.layer{li}.{bi}'
                    blk = SimpleNamespace(stride=stride if bi == 0 else 1, cin=cin if bi == 0 else cout, cout=cout)
                    blk.w1 = conv(p + '.conv1.weight'); blk.s1, blk.b1 = bn(p + '.bn1')
                    blk.w2 = conv(p + '.conv2.weight'); blk.s2, blk.b2 = bn(p + '.bn2')
                    # frozen 3x3 stride-1 filters, transformed ONCE per load for the Winograd F(2x2,3x3) kernel (conv_wino.hip:
                    # U = G g G^T in double, rounded once); clslam_conv2d picks the kernel per layer shape
                    blk.u1 = ops.wino_weight_transform(blk.w1) if (blk.stride == 1 and self.winograd) else None
                    blk.u2 = ops.wino_weight_transform(blk.w2) if self.winograd else None
                    blk.wd = None
                    if (p + '.downsample.0.weight') in sd:
                        blk.wd = conv(p + '.downsample.0.weight'); blk.sd, blk.bd = bn(p + '.downsample.1')
                    e.blocks.append(blk)
            self.enc[which] = e
        for name, off, shape in self.layout.entries:
            model, key = name.split('/', 1)
            t = torch.nn.Module.state_dict(self.models[model])[key].detach().to(dev, torch.float32)
            flat = TrainableLayout.to_compute(t)
            self._w[off:off + flat.numel()].copy_(flat)
        self._packed_version = self._module_version()
        self._modules_stale = False
        self._desc_cache.clear()               # re-laid-out weights / BN vectors: the kept descriptors point at the old ones
        for ws in self._ws.values():           # features of the previous encoder weights are void
            if hasattr(ws, 'frozen_valid'):
                ws.frozen_valid = False
        self._memo = None

    @torch.no_grad()
    def sync_modules(self) -> None:
        """Write the engine's adapted weights back into the module parameters (reference layout)."""
        if not self._modules_stale or self.models is None:
            return
        self.wait_training()
        for name, off, shape in self.layout.entries:
            model, key = name.split('/', 1)
            p = torch.nn.Module.state_dict(self.models[model], keep_vars=True)[key]
            p.data.copy_(TrainableLayout.to_reference(self._w[off:off + math.prod(shape)], shape))
        self._modules_stale = False
        self._packed_version = self._module_version()

    @torch.no_grad()
    def install_weights(self, flat: torch.Tensor, adam_step_count: Optional[int] = None) -> None:
        """Replace the whole trainable arena (compute layout, `layout.size` floats) -- the receiving side of a weight
        broadcast (clslam_hip.async_mode).  Module parameters are re-materialised lazily like after an optimizer step."""
        self.pack_if_needed()
        self.wait_training()
        self._w.copy_(flat.reshape(-1)[:self.layout.size])
        if adam_step_count is not None:
            self.adam_step_count = int(adam_step_count)
        self._modules_stale = True

    def wview(self, name: str) -> torch.Tensor:
        self.wait_training()
        for n, off, shape in self.layout.entries:
            if n == name:
                return self._w[off:off + math.prod(shape)]
        raise KeyError(name)

    def _slot(self, buf: torch.Tensor, name: str, numel: int) -> torch.Tensor:
        off = self.layout.offset[name]
        return buf[off:off + numel]

    # ------------------------------------------------------------------------------------------
    # workspace
    def workspace(self, B: int) -> SimpleNamespace:
        ws = self._ws.get(B)
        if ws is not None:
            return ws
        dev, H, W = self.device, self.H, self.W
        E = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
        ws = SimpleNamespace(B=B)
        hs = [H >> k for k in range(6)]
        wsz = [W >> k for k in range(6)]

        ws.denc = self._enc_bufs(B)
        ws.penc = self._enc_bufs(2 * B)
        # depth decoder activations: x[i][j] = output of upconv_i_j (post-ELU), NHWC
        ws.x = {}
        for i in range(4, -1, -1):
            ws.x[i, 0] = E(B, hs[i + 1], wsz[i + 1], NUM_CH_DEC[i])
            ws.x[i, 1] = E(B, hs[i], wsz[i], NUM_CH_DEC[i])
        ws.disp = [E(B, hs[s], wsz[s]) for s in range(4)]
        # pose decoder
        ws.sq = E(2 * B, hs[5], wsz[5], 256)
        ws.p0 = E(2 * B, hs[5], wsz[5], 256)
        ws.p1 = E(2 * B, hs[5], wsz[5], 256)
        ws.pmean = E(2 * B, 256)
        ws.pose = E(2 * B, 12)
        # loss stage
        ws.T = E(2, B, 4, 4)
        ws.P = E(2, B, 3, 4)
        ws.depth = E(4, B, H, W)
        ws.warped = E(4, 2, B, 3, H, W)
        ws.idmap = E(2, B, H, W)
        ws.coef = None  # allocated on the first training step
        ws.sel = torch.empty(4, B, H, W, dtype=torch.uint8, device=dev)
        ws.nblk = ops.automask_blocks(H, W)
        ws.partial = E(4, B, ws.nblk)
        ws.means = E(4, B, ops.disp_mean_chunks())
        ws.losses = E(18)
        ws.noise = E(4, B, 2, H, W)
        ws.train = None
        self._ws[B] = ws
        return ws

    def _enc_bufs(self, n: int) -> SimpleNamespace:
        dev, H, W = self.device, self.H, self.W
        E = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
        hs = [H >> k for k in range(6)]
        wsz = [W >> k for k in range(6)]
        e = SimpleNamespace()
        e.f0 = E(n, hs[1], wsz[1], 64)
        e.pool = E(n, hs[2], wsz[2], 64)
        e.t = [E(n, hs[2 + li], wsz[2 + li], c) for li, c in enumerate((64, 128, 256, 512))]
        e.ds = [None] + [E(n, hs[2 + li], wsz[2 + li], c) for li, c in ((1, 128), (2, 256), (3, 512))]
        e.y = [[E(n, hs[2 + li], wsz[2 + li], c) for _ in range(2)] for li, c in enumerate((64, 128, 256, 512))]
        return e

    def _train_bufs(self, ws: SimpleNamespace) -> SimpleNamespace:
        if ws.train is not None:
            return ws.train
        dev, H, W, B = self.device, self.H, self.W, ws.B
        E = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
        t = SimpleNamespace()
        ws.coef = E(4, B, 9, H, W)       # SSIM coefficients of the selected frame per pixel
        t.ddisp_up = E(4, B, H, W)
        t.nb2 = ops.loss_bwd2_blocks(H, W)
        t.dp_partial = torch.empty(4, B, t.nb2, 24, dtype=torch.float64, device=dev)   # block sums of dL/dP, in double
        t.dz_disp = [torch.empty(B, H >> s, W >> s, device=dev) for s in range(4)]
        t.dpose = E(2 * B, 12)
        t.dz = {k: torch.empty_like(v) for k, v in ws.x.items()}   # d(pre-ELU) of every upconv output
        pad_elems = max(B * ((H >> i) + 2) * ((W >> i) + 2) * NUM_CH_DEC[i] for i in range(5))
        t.dxp = [E(pad_elems), E(pad_elems)]
        t.wt_ready = None
        t.wt_dec = {}   # (i, j) -> flipped/transposed upconv_i_j weight for the dgrad convs (refreshed every backward)
        t.wt_pose = {}  # 0 / 1 -> the same for pose_decoder pose_0 / pose_1
        t.wt_table = None
        t.dz_p1 = torch.empty_like(ws.p1)
        t.dz_p0 = torch.empty_like(ws.p0)
        t.dz_sq = torch.empty_like(ws.sq)
        t.plan, t.items, t.table, t.disp_part = {}, [], None, {}
        t.side = SimpleNamespace()
        # fused bias-grad partials per layer: one row per fold workgroup (pooled folds run at the resolution of x)
        t.bias_part = {k: E(ops.fold_blocks(v.shape[0], v.shape[1], v.shape[2], v.shape[3], False) * v.shape[-1])
                       for k, v in ws.x.items()}
        ws.train = t
        return t

    # ------------------------------------------------------------------------------------------
    # forward pieces
    CONV_WS_BYTES = 32 << 20

    def _conv_workspace(self, handle: Optional[int] = None) -> None:
        """Make sure the stream `handle` (default: the current stream) has its split-K scratch registered."""
        if self.device.type != 'cuda' and handle is None:
            handle = 0
        if handle is None:
            handle = torch.cuda.current_stream(self.device).cuda_stream
        if handle not in self._conv_ws:
            self._conv_ws[handle] = torch.zeros(self.CONV_WS_BYTES, dtype=torch.uint8, device=self.device)
            ops.set_conv_workspace(handle, self._conv_ws[handle])

    def _encoder(self, e, bufs, n: int, stem_inputs, waits=None, stream=None, aux=None, cu_limit=None) -> List[torch.Tensor]:
        """stem_inputs: list of (img_a, img_b|None, batch offset, count); returns the 5 NHWC features.
        waits: one event per stem launch (its input image still crossing PCIe) for the current stream to wait on.
        aux: a second stream for the three 1x1 stride-2 `downsample` convolutions of the stage entries.  They read the same
        input as the stage's first 3x3 convolution and nothing depends on them until its second one, but at 0.16-0.3 GF
        they are pure launch latency (~11 us each for ~1 us of MFMA work): on `aux` they run underneath conv1 instead of
        between conv1 and conv2 of the chain.  `stream`: the stream this encoder's launches go to (needed with aux)."""
        ops.PERSISTENT_CU_LIMIT = self._cu_enc if cu_limit is None else cu_limit
        ek = (id(self), 'enc', id(e), id(bufs), n)       # descriptor-cache key prefix of this encoder on these buffers
        for i, (img_a, img_b, off, cnt) in enumerate(stem_inputs):
            if waits is not None:
                (stream or torch.cuda.current_stream(self.device)).wait_event(waits[i])
            ops.stem_conv(img_a, img_b, e.stem_w, e.stem_scale, e.stem_shift, bufs.f0[off:off + cnt])
        ops.maxpool3x3s2(bufs.f0, bufs.pool)
        x = bufs.pool
        feats = [bufs.f0]
        for li in range(4):
            for bi in range(2):
                blk = e.blocks[li * 2 + bi]
                t, y = bufs.t[li], bufs.y[li][bi]
                res, joined = x, None
                if blk.wd is not None:
                    res = bufs.ds[li]
                    if aux is not None and stream is not None:
                        evs = bufs.__dict__.setdefault('ds_events', {})
                        fork, joined = evs.get(li) or evs.setdefault(li, (torch.cuda.Event(), torch.cuda.Event()))
                        fork.record(stream)                 # x is complete on the chain's stream
                        aux.wait_event(fork)
                        with self._on(aux):
                            ops.conv2d(x, blk.wd, res, scale=blk.sd, shift=blk.bd, ksize=1, stride=blk.stride, pad=0, act=ACT_NONE,
                                       cache=self._desc_cache, key=(ek, li, bi, 'd'))
                        joined.record(aux)
                ops.conv2d(x, blk.w1, t, scale=blk.s1, shift=blk.b1, ksize=3, stride=blk.stride, act=ACT_RELU, weight_wino=blk.u1,
                           cache=self._desc_cache, key=(ek, li, bi, 1))
                if blk.wd is not None and joined is None:
                    ops.conv2d(x, blk.wd, res, scale=blk.sd, shift=blk.bd, ksize=1, stride=blk.stride, pad=0, act=ACT_NONE,
                               cache=self._desc_cache, key=(ek, li, bi, 'd'))
                if joined is not None:
                    stream.wait_event(joined)
                ops.conv2d(t, blk.w2, y, scale=blk.s2, shift=blk.b2, residual=res, ksize=3, act=ACT_RELU, weight_wino=blk.u2,
                           cache=self._desc_cache, key=(ek, li, bi, 2))
                x = y
            feats.append(x)
        return feats

    def _wb(self, prefix: str, cout: int, cin: int, taps: int):
        # views of the flat weight arena (allocated once, packed in place): built once per layer, ~70 lookups per step
        hit = self._wb_cache.get(prefix)
        if hit is None:
            w = self._slot(self._w, prefix + '.weight', cout * taps * cin).view(cout, taps, cin)
            b = self._slot(self._w, prefix + '.bias', cout)
            hit = self._wb_cache[prefix] = (w, b)
        return hit

    def _depth_decoder(self, ws, feats: List[torch.Tensor], early=None) -> None:
        """early: callable(scale), run on the leaf stream right behind the disparity head of scales 3, 2, 1 (forward(): the
        coarse scales' loss stage)"""
        x = feats[4]
        ops.PERSISTENT_CU_LIMIT = self._cu_dec
        # the disparity heads of scales 3..1 are leaves of the chain (only the view synthesis reads them): they go to
        # the wgrad stream, idle during the forward, instead of sitting between two convolutions of the chain
        main = self._main
        leaf = self.wg_stream if (main is not None and self.use_side_stream and self.wg_stream is not None) else None
        forked = False
        for i in range(4, -1, -1):
            cin0 = NUM_CH_ENC[-1] if i == 4 else NUM_CH_DEC[i + 1]
            w, b = self._wb(f'depth_decoder/upconv_{i}_0.conv.conv', NUM_CH_DEC[i], cin0, 9)
            ops.conv2d(x, w, ws.x[i, 0], shift=b, ksize=3, pad_mode=PAD_REFLECT, act=ACT_ELU, cache=self._desc_cache, key=(id(self), 'dec', id(ws), i, 0))
            skip = feats[i - 1] if i > 0 else None
            cin1 = NUM_CH_DEC[i] + (NUM_CH_ENC[i - 1] if i > 0 else 0)
            w, b = self._wb(f'depth_decoder/upconv_{i}_1.conv.conv', NUM_CH_DEC[i], cin1, 9)
            fork = leaf is not None and 0 < i <= 3
            ho = self._handoff if (fork and not self._capturing and ops.PROFILE is None) else None
            if ho is not None:
                ho.arm()          # the leaf stream is released by this convolution's own completion signal
            ops.conv2d(ws.x[i, 0], w, ws.x[i, 1], src_b=skip, shift=b, ksize=3, pad_mode=PAD_REFLECT, upsample_a=True,
                       act=ACT_ELU, cache=self._desc_cache, key=None if skip is None else (id(self), 'dec', id(ws), i, 1, skip.data_ptr()))
            x = ws.x[i, 1]
            if i <= 3:
                w, b = self._wb(f'depth_decoder/dispconv_{i}.conv', 1, NUM_CH_DEC[i], 9)
                if fork:
                    if ho is not None:
                        ho.release(main, leaf)
                    else:
                        ev = torch.cuda.Event()
                        ev.record(main)
                        leaf.wait_event(ev)
                    with self._on(leaf):
                        ops.dispconv_fwd(x, w.view(9, NUM_CH_DEC[i]), b, ws.disp[i])
                        if early is not None:
                            early(i)
                    forked = True
                else:
                    ops.dispconv_fwd(x, w.view(9, NUM_CH_DEC[i]), b, ws.disp[i])
        if forked:
            main.wait_stream(leaf)

    def _pose_decoder(self, ws, f4: torch.Tensor) -> None:
        ops.PERSISTENT_CU_LIMIT = self._cu_dec
        w, b = self._wb('pose_decoder/squeeze', 256, 512, 1)
        ops.conv2d(f4, w, ws.sq, shift=b, ksize=1, pad=0, act=ACT_RELU, cache=self._desc_cache, key=(id(self), 'pdec', id(ws), 0))
        w, b = self._wb('pose_decoder/pose_0', 256, 256, 9)
        ops.conv2d(ws.sq, w, ws.p0, shift=b, ksize=3, act=ACT_RELU, cache=self._desc_cache, key=(id(self), 'pdec', id(ws), 1))
        w, b = self._wb('pose_decoder/pose_1', 256, 256, 9)
        ops.conv2d(ws.p0, w, ws.p1, shift=b, ksize=3, act=ACT_RELU, cache=self._desc_cache, key=(id(self), 'pdec', id(ws), 2))
        w, b = self._wb('pose_decoder/pose_2', 12, 256, 1)
        ops.pose_head_fwd(ws.p1, w.view(12, 256), b, ws.pmean, ws.pose)

    # ------------------------------------------------------------------------------------------
    def forward(self, inputs: Dict[Any, torch.Tensor], *, train: bool, sample_w: torch.Tensor,
                smooth_w: Optional[torch.Tensor], noise: Optional[Dict[int, torch.Tensor]] = None,
                draw_noise: bool = True, keep_noise: bool = False,
                reuse_frozen: bool = False, inputs_ready=None) -> Tuple[Dict[Any, torch.Tensor], torch.Tensor]:
        """One _process_batch (dpp.py:906-923).  `inputs` tensors must already live on the device.

        reuse_frozen: the caller guarantees that `inputs` are the tensors of the previous forward of this
        batch size (steps 2..S of one adapt(steps=S) call, dpp.py:309-313).  Both encoders are frozen and in
        eval mode there (dpp.py:308), so their features and the identity-reprojection maps are bit-for-bit
        what the previous step computed: they are kept instead of recomputed (54 % of a step's forward
        flops).  Ignored when no valid features are held."""
        H, W = self.H, self.W
        # the caller's stream, looked up ONCE per call (torch.cuda.current_stream() costs 3 us; it was asked 30x per step)
        self._main = torch.cuda.current_stream(self.device) if self.device.type == 'cuda' else None
        self._conv_workspace(None if self._main is None else self._main.cuda_stream)
        if self._train_done is not None and self._main.cuda_stream != self.main_stream.cuda_stream:
            # a forward on the caller's stream (predict(), adapt(online, None)) after a detached training step: the encoders
            # below overwrite workspace the backward in flight still reads
            self._main.wait_event(self._train_done)
        if inputs_ready is not None:
            # inputs still crossing PCIe on the caller's copy stream: events (rgb_aug[0], rgb_aug[-1], rgb_aug[+1], everything
            # there).  The depth net only reads rgb_aug[0], the pose net the three rgb_aug frames; the un-augmented
            # frames are first needed by the identity maps / the loss stage.
            # _img() converts non-fp32 / non-contiguous images with a torch kernel on the MAIN stream: that kernel must not
            # read planes still in flight on the copy stream, so a batch that needs a conversion waits for all of it
            needs_conversion = any(inputs[k, f, 0].dtype != torch.float32 or not inputs[k, f, 0].is_contiguous()
                                   for k in ('rgb_aug', 'rgb') for f in (-1, 0, 1))
            self._main.wait_event(inputs_ready[3 if needs_conversion else 0])
        aug = {f: self._img(inputs['rgb_aug', f, 0]) for f in (-1, 0, 1)}
        rgb = {f: self._img(inputs['rgb', f, 0]) for f in (-1, 0, 1)}
        B = aug[0].shape[0]
        self._cu_enc, self._cu_dec = self.cu_limit(B, 'encoders'), self.cu_limit(B, 'decoders')
        self._cu_pose, self._cu_depth = self.cu_limit(B, 'enc_pose'), self.cu_limit(B, 'enc_depth')
        ops.PERSISTENT_CU_LIMIT = self._cu_dec
        # the kernels index with the engine's resolution: refuse anything else up front
        for name, group in (('rgb_aug', aug), ('rgb', rgb)):
            for f, t in group.items():
                if tuple(t.shape) != (B, 3, H, W):
                    raise ClslamError(f"('{name}', {f}, 0) must be ({B}, 3, {H}, {W}), got {tuple(t.shape)}")
        for s in range(1, 4):
            t = inputs['rgb', 0, s]
            if tuple(t.shape) != (B, 3, H >> s, W >> s):
                raise ClslamError(f"('rgb', 0, {s}) must be ({B}, 3, {H >> s}, {W >> s}), got {tuple(t.shape)}")
        for k in (('camera_matrix', 0), ('inv_camera_matrix', 0)):
            if tuple(inputs[k].shape) != (B, 4, 4):
                raise ClslamError(f'{k} must be ({B}, 4, 4), got {tuple(inputs[k].shape)}')
        ws = self.workspace(B)
        reuse = bool(reuse_frozen and self.reuse_frozen_features and getattr(ws, 'frozen_valid', False))
        ws.frozen_valid = False
        memo_feats = self._memo_lookup(aug[0]) if (B == 1 and not reuse) else None
        if train:
            self._train_bufs(ws)
        if self.fresh_outputs:
            # the reference returns fresh tensors from every call: the output planes are (cheap, cached)
            # new allocations that the kernels write directly -- no copies
            pre = self._prealloc if self._prealloc_armed else None
            if pre is not None and pre[0] == B:
                self._prealloc = None
                # detached step: allocated by the caller's stream (prealloc_outputs) -- the blocks belong to the pool of the
                # stream that will read and free them, so no record_stream bookkeeping is needed (it cost ~0.1 ms per step at
                # B = 1: an event per block and free, polled by every later allocation)
                ws.disp, ws.pose, ws.T, ws.depth, ws.warped, ws.losses = pre[1]
            else:
                E = lambda *s: torch.empty(*s, device=self.device)  # noqa: E731
                ws.disp = [E(B, H >> s, W >> s) for s in range(4)]
                ws.pose, ws.T = E(2 * B, 12), E(2, B, 4, 4)
                ws.depth, ws.warped, ws.losses = E(4, B, H, W), E(4, 2, B, 3, H, W), E(18)
        def identity_and_noise() -> bool:
            """Identity reprojection maps (dpp.py:1047-1052) and the tie-break noise depend on the inputs only."""
            if not reuse:
                # one launch per source frame straight from the caller's planes (stacking them first cost two 7 MB copies)
                ops.photo_map(rgb[-1], rgb[0], ws.idmap[0], None, B, B, H, W)
                ops.photo_map(rgb[1], rgb[0], ws.idmap[1], None, B, B, H, W)
            if noise is not None:
                for s in range(4):
                    ws.noise[s].copy_(noise[s])
                return True
            if keep_noise:                       # already copied into ws.noise by the graph driver
                return True
            if draw_noise and self._capturing:
                # a captured step replays fixed kernel arguments: torch's graph-safe generator keeps the draws fresh
                ws.noise.normal_().mul_(1e-5)  # dpp.py:1055-1056
                return True
            return False

        # networks ---------------------------------------------------------------------------
        side = self.side_stream if (self.use_side_stream and self.side_stream is not None) else None
        id_ready = None
        wg = self.wg_stream if (self.use_side_stream and self.wg_stream is not None) else None
        # the 1x1 downsample convolutions of both encoders go to the wgrad stream (idle during the forward once the identity
        # maps are out) -- for a single triplet only, where the step is a chain of dependent 5-15 us launches (K = 0, 192x640:
        # 1.454 -> 1.403 ms per frame incl. read-back).  At B = 5 the chip is throughput-bound and the six fork/join event
        # pairs cost more than the launches they hide (3.311 -> 3.377 ms): CLSLAM_DS_AUX=1 / 0 forces it on / off.
        ds_mode = os.environ.get('CLSLAM_DS_AUX', 'auto')
        ds_aux = wg if ((ds_mode == '1' or (ds_mode == 'auto' and B == 1)) and not self._capturing) else None

        def enqueue_identity():
            nonlocal have_noise, id_ready
            if inputs_ready is not None:
                wg.wait_event(inputs_ready[3])
            # (injected / captured noise is written by torch ops: they need torch's current stream switched)
            with (torch.cuda.stream(wg) if (noise is not None or self._capturing) else ops.launch_on(wg)):
                have_noise = identity_and_noise()
            id_ready = torch.cuda.Event()
            id_ready.record(wg)
        have_noise = False
        early_state = None
        # A host minibatch still crossing PCIe: the identity maps read the un-augmented frames, i.e. they wait for the WHOLE
        # upload -- and so would everything queued behind them on the wgrad stream.  With the downsample convolutions on that
        # stream (single triplets) the first stage entry of the depth encoder must not inherit that wait: the
        # identity maps then go out AFTER the networks' launches.
        defer_identity = wg is not None and ds_aux is not None and inputs_ready is not None and not reuse
        if wg is not None:
            # the wgrad stream idles during the forward: the input-only work runs there, off the critical path
            wg.wait_stream(self._main)
            if not defer_identity:
                enqueue_identity()
        if side is not None:
            main = self._main
            side.wait_stream(main)

            def pose_branch():
                with self._on(side):
                    # pose pairs in temporal order (dpp.py:949-955): (-1, 0) and (0, +1), batched as 2B
                    pf4 = ws.pf4 if reuse else self._encoder(self.enc['pose_encoder'], ws.penc, 2 * B,
                                                             [(aug[-1], aug[0], 0, B), (aug[0], aug[1], B, B)],
                                                             waits=None if inputs_ready is None else inputs_ready[1:3],
                                                             stream=side, aux=ds_aux, cu_limit=self._cu_pose)[4]
                    self.wait_training(side)      # the (frozen) encoder above does not need the optimizer step in flight
                    self._pose_decoder(ws, pf4)
                    return pf4

            def depth_encoder():
                if reuse:
                    return ws.dfeats
                if memo_feats is not None:
                    return memo_feats
                return self._encoder(self.enc['depth_encoder'], ws.denc, B, [(aug[0], None, 0, B)], stream=main, aux=ds_aux,
                                     cu_limit=self._cu_depth)
            # the host enqueues ~50 launches per branch (~0.7 ms): the depth branch is the longer
            # dependency chain (encoder + decoder), so its kernels go out first
            early = (reuse and train and self.early_loss and B * H * W <= self.early_loss_max_pixels and wg is not None
                     and not self._capturing and ops.PROFILE is None and inputs_ready is None and id_ready is not None)
            if early:
                # the pose decoder (short) goes out first, with the projection matrices behind it on its own stream: every scale's
                # view synthesis waits for them only
                K = self._mat(inputs['camera_matrix', 0])
                Kinv = self._mat(inputs['inv_camera_matrix', 0])
                pf4 = pose_branch()
                with self._on(side):
                    ops.pose_to_proj(ws.pose, K, ws.T, ws.P)
                pose_ready = torch.cuda.Event()
                pose_ready.record(side)
                draw = None
                if not have_noise and draw_noise:
                    draw = self._next_noise_draw()       # ONE draw per forward: element index = (scale, sample, pixel)
                waited = [False]
                t = ws.train

                def early_loss(sc):       # on the leaf (= wgrad) stream, behind dispconv_sc; the identity maps / noise sit on it too
                    if not waited[0]:
                        wg.wait_event(pose_ready)
                        waited[0] = True
                    ops.warp_fwd_pyramid(ws.disp, rgb[-1], rgb[1], Kinv, ws.P, ws.depth, ws.warped, self.min_depth, self.max_depth,
                                         scales=(sc, 1))
                    if draw is None:
                        ops.photo_automask_pyramid(ws.warped, rgb[0], ws.idmap, ws.noise if have_noise else None, ws.sel, ws.coef,
                                                   ws.partial, B, H, W, scales=(sc, 1))
                    else:
                        ops.photo_automask_pyramid_rng(ws.warped, rgb[0], ws.idmap, draw[0], draw[1], ws.sel, ws.coef, ws.partial,
                                                       B, H, W, scales=(sc, 1))
                    ops.loss_bwd2_pyramid(ws.disp, ws.sel, ws.coef, ws.warped, rgb[0], rgb[-1], rgb[1], Kinv, ws.P, sample_w,
                                          t.ddisp_up, t.dp_partial, self.min_depth, self.max_depth, scales=(sc, 1))
                dfeats = ws.dfeats
                self.wait_training(main)
                self._depth_decoder(ws, dfeats, early=early_loss)
                early_state = SimpleNamespace(K=K, Kinv=Kinv, draw=draw)
            elif reuse:                 # no encoders to hide the host's launches behind: the long chain goes out first
                dfeats = ws.dfeats
                self.wait_training(main)
                self._depth_decoder(ws, dfeats)
                pf4 = pose_branch()
            elif self.depth_first:
                dfeats = depth_encoder()
                pf4 = pose_branch()
                self.wait_training(main)
                self._depth_decoder(ws, dfeats)
            else:
                pf4 = pose_branch()
                dfeats = depth_encoder()
                self.wait_training(main)
                self._depth_decoder(ws, dfeats)
            main.wait_stream(side)
        else:
            self.wait_training()
            dfeats = (ws.dfeats if reuse else memo_feats if memo_feats is not None else
                      self._encoder(self.enc['depth_encoder'], ws.denc, B, [(aug[0], None, 0, B)], cu_limit=self._cu_depth))
            self._depth_decoder(ws, dfeats)
            pf4 = ws.pf4 if reuse else self._encoder(self.enc['pose_encoder'], ws.penc, 2 * B,
                                                     [(aug[-1], aug[0], 0, B), (aug[0], aug[1], B, B)],
                                                     waits=None if inputs_ready is None else inputs_ready[1:3],
                                                     cu_limit=self._cu_pose)[4]
            self._pose_decoder(ws, pf4)
        ws.dfeats, ws.pf4 = dfeats, pf4
        if defer_identity:
            enqueue_identity()
        # view synthesis + loss ------------------------------------------------------------------
        if inputs_ready is not None:
            self._main.wait_event(inputs_ready[3])
        # scales still to do here: all four, or scale 0 only when the coarse ones went out beside the decoder (early_state)
        rest = (0, 4) if early_state is None else (0, 1)
        if early_state is None:
            K = self._mat(inputs['camera_matrix', 0])
            Kinv = self._mat(inputs['inv_camera_matrix', 0])
            ops.pose_to_proj(ws.pose, K, ws.T, ws.P)
        else:
            K, Kinv = early_state.K, early_state.Kinv
        ops.warp_fwd_pyramid(ws.disp, rgb[-1], rgb[1], Kinv, ws.P, ws.depth, ws.warped, self.min_depth, self.max_depth, scales=rest)
        if id_ready is not None:
            self._main.wait_event(id_ready)
        else:
            have_noise = identity_and_noise()
        # all four scales in one launch; reprojection maps stay in registers, only the selected frame's
        # SSIM coefficients are kept for the backward
        if have_noise or not draw_noise:
            ops.photo_automask_pyramid(ws.warped, rgb[0], ws.idmap, ws.noise if have_noise else None, ws.sel,
                                       ws.coef if train else None, ws.partial, B, H, W, scales=rest)
        else:
            # dpp.py:1055-1056 draws randn * 1e-5 on the compute device every scale, every step: here inside the kernel
            # (Philox keyed by torch's seed, one draw offset per forward) -- no noise tensor is written or read
            seed, offset = early_state.draw if early_state is not None else self._next_noise_draw()
            ops.photo_automask_pyramid_rng(ws.warped, rgb[0], ws.idmap, seed, offset, ws.sel,
                                           ws.coef if train else None, ws.partial, B, H, W, scales=rest)
        ws.loss_bwd_scales = rest       # backward(): the coarse scales' loss backward is already out
        ops.disp_mean_pyramid(ws.disp, ws.means, H, W)
        n_smooth = 0 if (smooth_w is None or self.smooth_intended) else int(smooth_w.numel())
        if n_smooth and not (n_smooth < (W >> 3) - 1):
            raise ClslamError('reference smoothness layout needs batch < width/8 - 1')
        aux = getattr(ws, 'smooth_aux', None)
        if n_smooth and (aux is None or aux.shape[1] != 2 + 2 * n_smooth):
            aux = torch.zeros(4, 2 + 2 * n_smooth, device=self.device)
            ws.smooth_aux = aux
        d0 = inputs['relative_distance', 0].to(torch.float64).reshape(-1).contiguous() if self.vel_scale > 0 else None
        d1 = inputs['relative_distance', 1].to(torch.float64).reshape(-1).contiguous() if self.vel_scale > 0 else None
        rgb0 = [self._img(inputs['rgb', 0, s]) for s in range(4)]
        ops.loss_finalize([ws.partial[s] for s in range(4)], ws.disp, rgb0, [ws.means[s] for s in range(4)], ws.pose, d0, d1,
                          sample_w, smooth_w if n_smooth else None, ws.losses, aux if n_smooth else None, B, ws.nblk, H, W,
                          n_smooth, self.smooth_scale, self.vel_scale)
        if self.smooth_intended:
            if getattr(ws, 'si_partial', None) is None:
                ws.si_partial = torch.empty(4, B, ops.smooth_intended_chunks(), device=self.device)
                ws.si_aux = torch.empty(4, B, 2, device=self.device)
            ops.smooth_intended_fwd(ws.disp, rgb0, ws.si_partial, H, W)
            ops.smooth_intended_finalize(ws.si_partial, ws.means, sample_w, ws.losses, ws.si_aux, B, H, W, self.smooth_scale)
        ws.ctx = SimpleNamespace(rgb=rgb, K=K, Kinv=Kinv, d0=d0, d1=d1, sample_w=sample_w, n_smooth=n_smooth,
                                 aux=aux if n_smooth else None, B=B, rgb0=rgb0)
        ws.frozen_valid = True      # encoder features + identity maps of THESE inputs and encoder weights are held
        return self._outputs(ws, B), ws.losses

    def _memo_lookup(self, x: torch.Tensor) -> Optional[List[torch.Tensor]]:
        """Features of the last run_encoder('depth_encoder', ...) call if `x` (1,3,H,W on the device, already complete on the
        current stream) has the same content and the encoder weights have not been re-packed since.  The comparison is one
        small kernel + a 1-byte read-back (~30 us) against the ~0.3 ms of launch-bound encoder it replaces; identity cannot
        be used -- slam.py hands the descriptor pass a device copy and adapt() a host tensor (slam.py:145, 300-309)."""
        m = self._memo
        if (m is None or not self.descriptor_memo or self._capturing or m.version != self._packed_version
                or m.x.shape != x.shape or m.x.device != x.device):
            return None
        # ONE look-up per descriptor pass: slam.py runs the pass (and reads its 512 numbers back, i.e. the stream is drained)
        # right before the frame's adapt() -- that is the forward that can repeat it.  The comparison reads one byte back,
        # which is a stream synchronisation: any later forward (another frame without a descriptor pass, the asynchronous
        # modes) must not pay it against a stale memo while a training step is in flight.
        self._memo = None
        if not bool(torch.equal(m.x, x)):
            return None
        self.memo_hits += 1
        return m.feats

    def _next_noise_draw(self) -> Tuple[int, int]:
        """(Philox key, draw offset) of this forward's tie-break noise.  On the GPU the draw counter IS the device generator's
        Philox offset (advanced by 4 per forward like a torch kernel drawing up to four values per thread would): manual_seed,
        get_state / set_state of torch's generator reproduce the stream exactly as they do for the reference's torch.randn."""
        if self.device.type == 'cuda':
            gen = torch.cuda.default_generators[self.device.index if self.device.index is not None else torch.cuda.current_device()]
            ts = int(gen.initial_seed())
            off = int(gen.get_offset())
            gen.set_offset(off + 4)
            self.noise_draws = off // 4 + 1
        else:           # the test emulator's CPU generator has no offset: a counter that restarts when the seed changes
            ts = int(torch.default_generator.initial_seed())
            if ts != self._noise_torch_seed:
                self.noise_draws = 0
            self.noise_draws += 1
        if ts != self._noise_torch_seed:
            self._noise_torch_seed = ts
            self.noise_seed = (ts * 0x9E3779B97F4A7C15 + 0x7F4A7C15) & 0xFFFFFFFFFFFFFFFF or 1
        return self.noise_seed, (int(self.noise_stream) << 40) | self.noise_draws

    def _img(self, t: torch.Tensor) -> torch.Tensor:
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(torch.float32).contiguous()
        return t

    def _mat(self, t: torch.Tensor) -> torch.Tensor:
        return t.to(torch.float32).contiguous()

    def _outputs(self, ws, B: int) -> Dict[Any, torch.Tensor]:
        """Output dict with the reference's keys / shapes / insertion order."""
        out: Dict[Any, torch.Tensor] = {}
        for s in (3, 2, 1, 0):
            out['disp', s] = ws.disp[s].unsqueeze(1)
        for fi, f in enumerate((-1, 1)):
            out['axis_angle', 0, f] = ws.pose[fi * B:(fi + 1) * B, 0:3].unsqueeze(1)
            out['translation', 0, f] = ws.pose[fi * B:(fi + 1) * B, 3:6].unsqueeze(1)
            out['cam_T_cam', 0, f] = ws.T[fi]
        for s in range(4):
            out['depth', s] = ws.depth[s].unsqueeze(1)
            for fi, f in enumerate((-1, 1)):
                out['rgb', f, s] = ws.warped[s, fi]
        return out

    def losses_dict(self, losses: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Loss dict with the reference's keys (dpp.py:1074-1112); 'depth_loss' aliases the total
        (SURVEY.md 0.4)."""
        d: Dict[str, torch.Tensor] = {}
        for s in range(4):
            for j, name in enumerate(LOSS_NAMES):
                d[f'{name}/scale_{s}'] = losses[s * 4 + j]
        total = losses[17:18]
        d['depth_loss'] = total
        if self.vel_scale > 0:
            d['velocity_loss'] = losses[16]
        d['loss'] = total
        return d

    # ------------------------------------------------------------------------------------------
    # backward
    def _wgrad(self, t, desc_src, out_shape, dz, name_prefix: str, cout: int, cin: int, taps: int, bias_blocks: int = 0,
               bias_partial=None, **geom) -> None:
        """Weight (+ bias) gradient partials of one conv.  Every layer owns its partial buffers; the sums
        into the gradient arena happen in ONE batched reduction at the end of backward()
        (clslam_reduce_multi).  bias_blocks > 0: bias_partial already holds that many per-block column
        sums of dz (fused into fold_act_grad)."""
        desc = ops.conv_desc(desc_src[0], out_shape, src_b=desc_src[1], ksize=3 if taps == 9 else 1, **geom)
        n = cout * taps * cin
        plan = t.plan.get(name_prefix)
        if plan is None:
            use_patch = ops.wgrad_patch_supported(desc)
            # two blocks per CU: 256 / 128 blocks would halve / quarter the 220 MB of split partials but measured 3.38 / 3.44 ms
            # per step against 3.34 (the wgrad launches become the tail of the backward)
            tb = int(os.environ.get('CLSLAM_WGRAD_BLOCKS', '512'))
            splits = ops.wgrad_patch_splits(desc, tb) if use_patch else ops.wgrad_splits(desc, tb)
            plan = SimpleNamespace(use_patch=use_patch, splits=splits, partial=torch.empty(splits * n, device=self.device),
                                   colsum=None, nb=0)
            t.items.append((plan.partial, self._slot(self._g, name_prefix + '.weight', n), n, splits))
            if bias_blocks:
                t.items.append((bias_partial, self._slot(self._g, name_prefix + '.bias', cout), cout, bias_blocks))
            else:
                rows = out_shape[0] * out_shape[1] * out_shape[2]
                plan.nb = ops.colsum_blocks(rows)
                plan.colsum = torch.empty(plan.nb * cout, device=self.device)
                t.items.append((plan.colsum, self._slot(self._g, name_prefix + '.bias', cout), cout, plan.nb))
            t.plan[name_prefix] = plan
        if plan.use_patch:
            ops.conv_wgrad_patch(desc, dz, plan.partial, plan.splits)
        else:
            ops.conv_wgrad(desc, dz, plan.partial, plan.splits)
        if plan.colsum is not None:
            ops.colsum(dz, plan.colsum, out_shape[0] * out_shape[1] * out_shape[2], cout)

    def bucket_ranges(self):
        """[(name, first float, end float)] of the gradient buckets in backward-COMPLETION order (contiguous in the arena)."""
        off = self.layout.offset
        a_end, c_start = off['depth_decoder/upconv_3_0.conv.conv.weight'], off['pose_decoder/squeeze.weight']
        if self.grad_buckets >= 3:
            return [('pose', c_start, self.layout.size), ('mid', a_end, c_start), ('deep', 0, a_end)]
        if self.grad_buckets == 2:
            return [('early', a_end, self.layout.size), ('deep', 0, a_end)]
        return [('all', 0, self.layout.size)]

    def exchange_report(self) -> Optional[Dict[str, Any]]:
        """After a synchronised data-parallel step with `time_exchange` set: per bucket the all-reduce's time on the tail stream
        (events around the collective: its kernels AND whatever they waited for on that stream's queue), the part of the exchange
        that was still running when the backward's own last kernel had finished (`exposed_ms`: what the optimizer step waits
        for) and the fraction hidden under the backward."""
        if not self.exchange_events or self.exchange_main_done is None:
            return None
        buckets = [{'bucket': n, 'mbytes': round(nb / 1e6, 2), 'ms': round(e0.elapsed_time(e1), 4)} for n, e0, e1, nb in self.exchange_events]
        total = sum(b['ms'] for b in buckets)
        exposed = max(0.0, self.exchange_main_done.elapsed_time(self.exchange_events[-1][2]))
        rep = {'buckets': buckets, 'allreduce_ms': round(total, 4), 'exposed_ms': round(exposed, 4),
               'overlap_fraction': round(1.0 - min(exposed, total) / total, 4) if total > 0 else None}
        self.exchange_events, self.exchange_main_done = [], None
        return rep

    def _bucket_sync(self, t, name: str, events, allreduce) -> None:
        """tail stream: wait for the bucket's producers, reduce its partials into the arena, all-reduce its slice."""
        tail = self.tail_stream
        for ev in events:
            tail.wait_event(ev)
        table, n, lo, hi = t.bucket_tables[name]
        with torch.cuda.stream(tail):
            ops.reduce_multi(table, n, self._g)
            if self.time_exchange:
                # diagnosis of a multi-GPU run (bench.py --gpus N): how long each bucket's all-reduce took on the tail stream
                # and how much of it was still running when the backward's own last kernel had finished
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(tail)
                allreduce(self._g[lo:hi])
                e1.record(tail)
                self.exchange_events.append((name, e0, e1, (hi - lo) * 4))
            else:
                allreduce(self._g[lo:hi])

    def backward(self, B: int, defer_reduce: bool = False, allreduce=None) -> None:
        """dL/d(trainable arena) for the last training forward; fills self._g (dpp.py:312).
        defer_reduce: the caller promises that adam() follows directly (adapt(), single GPU): the batched reduction of the
        gradient partials is left to adam(), which fuses it with the update; self._g is complete after THAT launch.
        allreduce: data-parallel mode -- callable(slice of the gradient arena) that sum-all-reduces it over the ranks on
        torch's current stream.  With it (and grad_buckets > 1) backward() exchanges the gradients itself, bucket by bucket
        (see __init__); self.grads_synced tells the caller so."""
        ops.PERSISTENT_CU_LIMIT = self.cu_limit(B, 'backward')
        ws = self._ws[B]
        t = ws.train
        c = ws.ctx
        H, W = self.H, self.W
        feats = ws.dfeats
        self._main = torch.cuda.current_stream(self.device) if self.device.type == 'cuda' else None
        wg = self.wg_stream if (self.use_side_stream and self.wg_stream is not None) else None
        t.wt_ready = None
        if wg is None:
            self._transpose_decoder_weights(t)
        else:
            wg.wait_stream(self._main)
            with self._on(wg):
                self._transpose_decoder_weights(t)
            t.wt_ready = torch.cuda.Event()
            t.wt_ready.record(wg)
        # loss -> disparity logits and pose-decoder output ----------------------------------------
        ops.loss_bwd2_pyramid(ws.disp, ws.sel, ws.coef, ws.warped, c.rgb[0], c.rgb[-1], c.rgb[1], c.Kinv, ws.P, c.sample_w,
                              t.ddisp_up, t.dp_partial, self.min_depth, self.max_depth,
                              scales=getattr(ws, 'loss_bwd_scales', (0, 4)))
        ws.loss_bwd_scales = (0, 4)
        ops.disp_grad_pyramid(t.ddisp_up, ws.disp, c.aux if c.n_smooth else None, c.n_smooth, t.dz_disp, H, W)
        if self.smooth_intended:
            ops.smooth_intended_bwd(ws.disp, c.rgb0, ws.si_aux, c.sample_w, t.dz_disp, H, W, self.smooth_scale)
        ops.pose_bwd(t.dp_partial, 4, t.nb2, ws.pose, c.K, c.d0, c.d1, c.sample_w, self.vel_scale, t.dpose)
        self.inputs_released = None
        if self._caller is not None:
            # detached step: nothing below reads the caller's minibatch (frames, intrinsics, distances) or the output planes
            # handed to it any more -- the three launches above were the last.  The caller's stream continues behind THIS point:
            # it may overwrite its input buffers in place or drop them at once, like after the reference's synchronous step.
            self.inputs_released = torch.cuda.Event()
            self.inputs_released.record(self._main)
        side = self.side_stream if (self.use_side_stream and self.side_stream is not None) else None
        bucketed = allreduce is not None and self.grad_buckets > 1
        # overlapped exchange: needs the tail stream, the side / wgrad streams and the per-bucket tables of a previous backward
        overlap = (bucketed and self._use_tail() and side is not None and wg is not None
                   and getattr(t, 'bucket_tables', None) is not None)
        self.grads_synced = False
        if side is not None:
            main = self._main
            side.wait_stream(main)
            if t.wt_ready is not None:
                side.wait_event(t.wt_ready)     # transposed pose_0 / pose_1 weights
            with self._on(side):
                self._backward_pose_decoder(ws, t, B, t.side)
            if overlap and self.grad_buckets >= 3:
                ev = torch.cuda.Event()
                ev.record(side)                 # the pose decoder's three weight gradients + pose_2's are complete behind this
                self._bucket_sync(t, 'pose', [ev], allreduce)
            self._backward_depth_decoder(ws, t, B, mid_done=(lambda evs: self._bucket_sync(
                t, 'mid' if self.grad_buckets >= 3 else 'early', evs, allreduce)) if overlap else None)
            main.wait_stream(side)
        else:
            self._backward_depth_decoder(ws, t, B)
            self._backward_pose_decoder(ws, t, B, t)
        # one batched, deterministic reduction of every weight / bias gradient partial into the arena
        if t.table is None:
            # (pose_2's gradients are written straight into the arena by pose_head_bwd: identity items, so that every
            # trainable element is the output of exactly one item -- the fused reduction + Adam updates per item)
            for key, n in (('pose_decoder/pose_2.weight', 12 * 256), ('pose_decoder/pose_2.bias', 12)):
                sl = self._slot(self._g, key, n)
                t.items.append((sl, sl, n, 1))
            t.table = ops.make_reduce_table(t.items, self.device)
            # the same items grouped by the bucket their destination lies in (every item reduces independently, in a fixed
            # order: three launches give bitwise the single launch's arena)
            base = self._g.data_ptr()
            t.bucket_tables = {}
            for name, lo, hi in self.bucket_ranges():
                its = [it for it in t.items if lo <= (it[1].data_ptr() - base) // 4 < hi]
                t.bucket_tables[name] = (ops.make_reduce_table(its, self.device), len(its), lo, hi)
            assert sum(v[1] for v in t.bucket_tables.values()) == len(t.items)
        self._pending_reduce = None
        if bucketed:
            ranges = self.bucket_ranges()
            if overlap:
                # the earlier buckets went out during the backward; the last one (level 4) is complete on the main stream now
                name, lo, hi = ranges[-1]
                if self.time_exchange:
                    self.exchange_main_done = torch.cuda.Event(enable_timing=True)
                    self.exchange_main_done.record(self._main)
                self.tail_stream.wait_stream(self._main)
                self._bucket_sync(t, name, [], allreduce)
                self._tail_open = True
            else:
                # no overlap possible (first backward of this workspace, no tail / side streams, instrumented run): one
                # reduction, then the SAME sequence of collectives on the same slices
                if self._use_tail():
                    self.tail_stream.wait_stream(self._main)
                    with torch.cuda.stream(self.tail_stream):
                        ops.reduce_multi(t.table, len(t.items), self._g)
                        for name, lo, hi in ranges:
                            allreduce(self._g[lo:hi])
                    self._tail_open = True
                else:
                    ops.reduce_multi(t.table, len(t.items), self._g)
                    for name, lo, hi in ranges:
                        allreduce(self._g[lo:hi])
            self.grads_synced = True
        elif defer_reduce and self.fuse_adam and not self.data_parallel and not self._capturing and not self._use_tail():
            self._pending_reduce = (t.table, len(t.items))
        elif self._use_tail():
            # the partial buffers are complete on the current stream; the reduction, the all-reduce (caller, under
            # training_stream()) and Adam follow on the tail stream and nobody waits for them here
            self.tail_stream.wait_stream(self._main)
            with torch.cuda.stream(self.tail_stream):
                ops.reduce_multi(t.table, len(t.items), self._g)
            self._tail_open = True
        else:
            ops.reduce_multi(t.table, len(t.items), self._g)

    def _transpose_decoder_weights(self, t) -> None:
        """Flipped/transposed weights for every dgrad conv of the step (nine depth-decoder upconvs, pose_0 / pose_1).
        They depend only on the weights, so all eleven are produced by ONE launch at the start of backward() on the
        wgrad stream, where it overlaps the loss backward instead of sitting between the dgrad convs of the critical
        paths (eleven launches before: 0.06 ms of the B = 1 step)."""
        if t.wt_table is None:
            items = []
            for i in range(5):
                ci = NUM_CH_DEC[i]
                cin1 = ci + (NUM_CH_ENC[i - 1] if i > 0 else 0)
                t.wt_dec[i, 1] = torch.empty(ci, 9, ci, device=self.device)
                w1, _ = self._wb(f'depth_decoder/upconv_{i}_1.conv.conv', ci, cin1, 9)
                items.append((w1, t.wt_dec[i, 1], ci))            # only the up(x[i,0]) half: the skip half is frozen
                if i < 4:
                    cin0 = NUM_CH_DEC[i + 1]
                    t.wt_dec[i, 0] = torch.empty(cin0, 9, ci, device=self.device)
                    w0, _ = self._wb(f'depth_decoder/upconv_{i}_0.conv.conv', ci, cin0, 9)
                    items.append((w0, t.wt_dec[i, 0], None))
            for k in (0, 1):
                t.wt_pose[k] = torch.empty(256, 9, 256, device=self.device)
                wp, _ = self._wb(f'pose_decoder/pose_{k}', 256, 256, 9)
                items.append((wp, t.wt_pose[k], None))
            t.wt_table = ops.transpose_table(items)
        ops.weight_transpose_multi(t.wt_table, self._w)

    def _backward_depth_decoder(self, ws, t, B: int, mid_done=None) -> None:
        """dgrad chain (critical path) on the current stream; every weight/bias gradient is independent
        of the rest of the chain once its dz exists, so those run on a third stream (`wg_stream`)."""
        H, W = self.H, self.W
        feats = ws.dfeats
        main = self._main
        wg = self.wg_stream if (self.use_side_stream and self.wg_stream is not None) else None
        if wg is not None:
            wg.wait_stream(main)

        # The weight-gradient kernels add up to about as much GPU time as the dgrad chain itself and the last
        # (largest-weight) layers only become ready at the very end, so they alternate between the wgrad stream
        # and the side stream (idle once the short pose-decoder backward is through): two wgrads in flight
        # instead of a serial tail after the dgrad chain.
        wg_streams = [wg] if (wg is None or self.side_stream is None) else [wg, self.side_stream]
        counter = [0]

        ho = self._handoff if (wg is not None and not self._capturing and ops.PROFILE is None) else None

        def on_wg(fn, dep='record'):
            """run fn on a wgrad stream.  dep: 'record' -- after everything enqueued so far on the main stream (an event
            recorded there); 'armed' -- after the launch that took the hand-off event armed just before it (no marker packet on
            the chain); None -- its inputs were complete before this function started (both wgrad streams are ordered behind
            the loss backward by then)"""
            if wg is None:
                fn()
                return
            st = wg_streams[counter[0] % len(wg_streams)]
            counter[0] += 1
            if dep == 'armed':
                ho.release(main, st)
            elif dep == 'record':
                ev = torch.cuda.Event()
                ev.record(main)
                st.wait_event(ev)
            with self._on(st):
                fn()

        if t.wt_ready is not None:
            main.wait_event(t.wt_ready)     # transposed dgrad weights (issued at the start of backward())

        dxp_in = None  # padded-domain gradient w.r.t. x[i,1] coming from upconv_{i-1}_0
        for i in range(5):
            hi, wi, ci = H >> i, W >> i, NUM_CH_DEC[i]
            wd = None
            if i <= 3:
                # the dispconv head's data gradient is folded in by fold_act_grad below (no separate pass over dxp)
                wd, _ = self._wb(f'depth_decoder/dispconv_{i}.conv', 1, ci, 9)

                def disp_wgrad(i=i, hi=hi, wi=wi, ci=ci):   # dispconv weight + bias gradient partials
                    if i not in t.disp_part:
                        nb = ops.dispconv_wgrad_blocks(B * hi * wi)
                        t.disp_part[i] = torch.empty(nb * (9 * ci + 1), device=self.device)
                        t.items.append((t.disp_part[i], self._slot(self._g, f'depth_decoder/dispconv_{i}.conv.weight', 9 * ci + 1),
                                        9 * ci + 1, nb))
                    ops.dispconv_wgrad(t.dz_disp[i], ws.x[i, 1], t.disp_part[i])
                # (reads dz_disp[i] -- the loss backward -- and the forward activation x[i,1] only)
                on_wg(disp_wgrad, dep=None if ho is not None else 'record')
            nb1 = ops.fold_blocks(B, hi, wi, ci, False)
            if ho is not None:
                ho.arm()
            ops.fold_act_grad(dxp_in, ws.x[i, 1], t.dz[i, 1], h=hi, w=wi, ch=ci, border=1, pool=False, act=ACT_ELU,
                              bias_partial=t.bias_part[i, 1], disp_dz=t.dz_disp[i] if wd is not None else None,
                              disp_w=wd.view(9, ci) if wd is not None else None)
            # upconv_i_1: input = cat(up(x[i,0]), feats[i-1])
            skip = feats[i - 1] if i > 0 else None
            cin1 = ci + (NUM_CH_ENC[i - 1] if i > 0 else 0)
            on_wg(lambda i=i, hi=hi, wi=wi, ci=ci, skip=skip, cin1=cin1, nb1=nb1: self._wgrad(
                t, (ws.x[i, 0], skip), (B, hi, wi, ci), t.dz[i, 1], f'depth_decoder/upconv_{i}_1.conv.conv', ci, cin1, 9,
                bias_blocks=nb1, bias_partial=t.bias_part[i, 1], pad_mode=PAD_REFLECT, upsample_a=True),
                dep='armed' if ho is not None else 'record')
            wt = t.wt_dec[i, 1]
            dxa = t.dxp[1][:B * (hi + 2) * (wi + 2) * ci].view(B, hi + 2, wi + 2, ci)
            ops.conv2d(t.dz[i, 1], wt, dxa, ksize=3, pad=2, cache=self._desc_cache, key=(id(self), 'dg', id(ws), i, 1))
            nb0 = ops.fold_blocks(B, hi, wi, ci, True)
            if ho is not None:
                ho.arm()
            ops.fold_act_grad(dxa, ws.x[i, 0], t.dz[i, 0], h=hi, w=wi, ch=ci, border=1, pool=True, act=ACT_ELU,
                              bias_partial=t.bias_part[i, 0])
            # upconv_i_0: input = x[i+1,1] (or the frozen encoder feature for i == 4)
            src = ws.x[i + 1, 1] if i < 4 else feats[4]
            cin0 = NUM_CH_ENC[-1] if i == 4 else NUM_CH_DEC[i + 1]
            h2, w2 = hi >> 1, wi >> 1
            on_wg(lambda i=i, h2=h2, w2=w2, ci=ci, src=src, cin0=cin0, nb0=nb0: self._wgrad(
                t, (src, None), (B, h2, w2, ci), t.dz[i, 0], f'depth_decoder/upconv_{i}_0.conv.conv', ci, cin0, 9,
                bias_blocks=nb0, bias_partial=t.bias_part[i, 0], pad_mode=PAD_REFLECT),
                dep='armed' if ho is not None else 'record')
            if i < 4:
                wt = t.wt_dec[i, 0]
                dxp_in = t.dxp[0][:B * (h2 + 2) * (w2 + 2) * cin0].view(B, h2 + 2, w2 + 2, cin0)
                ops.conv2d(t.dz[i, 0], wt, dxp_in, ksize=3, pad=2, cache=self._desc_cache, key=(id(self), 'dg', id(ws), i, 0))
            if i == 3 and mid_done is not None:
                # every gradient of levels 0..3 and of the four disparity heads has been enqueued: weight gradients on the
                # wgrad streams, bias partials by the folds on the main stream.  Level 4 (53 % of the arena) is still to come.
                evs = []
                for st in [main] + [x for x in wg_streams if x is not None]:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    evs.append(ev)
                mid_done(evs)
        if wg is not None:
            main.wait_stream(wg)

    def _backward_pose_decoder(self, ws, t, B: int, scratch) -> None:
        """`scratch` supplies the partial / colsum / wt buffers (a separate set when the pose branch runs
        concurrently with the depth branch)."""
        H, W = self.H, self.W
        n2 = 2 * B
        h5, w5 = H >> 5, W >> 5
        w2_, _ = self._wb('pose_decoder/pose_2', 12, 256, 1)
        ops.pose_head_bwd(t.dpose, ws.p1, w2_.view(12, 256), ws.pmean, t.dz_p1,
                          self._slot(self._g, 'pose_decoder/pose_2.weight', 12 * 256).view(12, 256),
                          self._slot(self._g, 'pose_decoder/pose_2.bias', 12))
        self._wgrad(t, (ws.p0, None), (n2, h5, w5, 256), t.dz_p1, 'pose_decoder/pose_1', 256, 256, 9)
        # (transposed weights: _transpose_decoder_weights, issued at the start of backward())
        ops.conv2d(t.dz_p1, t.wt_pose[1], t.dz_p0, ksize=3, pad=1, actgrad_src=ws.p0, actgrad_kind=ACT_RELU,
                   cache=self._desc_cache, key=(id(self), 'pdg', id(ws), 1))
        self._wgrad(t, (ws.sq, None), (n2, h5, w5, 256), t.dz_p0, 'pose_decoder/pose_0', 256, 256, 9)
        ops.conv2d(t.dz_p0, t.wt_pose[0], t.dz_sq, ksize=3, pad=1, actgrad_src=ws.sq, actgrad_kind=ACT_RELU,
                   cache=self._desc_cache, key=(id(self), 'pdg', id(ws), 0))
        self._wgrad(t, (ws.pf4, None), (n2, h5, w5, 256), t.dz_sq, 'pose_decoder/squeeze', 256, 512, 1, pad=0)

    # ------------------------------------------------------------------------------------------
    # hipGraph: the whole forward+backward of a training step (~130 kernel launches over three streams) can be
    # captured and replayed as one graph.  Kept as an option; see graph_preferred() for why it is not the default.
    GRAPH_KEYS = ([('rgb_aug', f, 0) for f in (-1, 0, 1)] + [('rgb', f, 0) for f in (-1, 0, 1)] +
                  [('rgb', 0, s) for s in (1, 2, 3)] + [('camera_matrix', 0), ('inv_camera_matrix', 0),
                                                        ('relative_distance', 0), ('relative_distance', 1)])

    def graphs_enabled(self) -> bool:
        return (self.device.type == 'cuda' and os.environ.get('CLSLAM_HIPGRAPH', 'auto') != '0' and ops.PROFILE is None)

    def graph_preferred(self, B: int) -> bool:
        """hipGraph replay of the step is opt-in (CLSLAM_HIPGRAPH=1).  Measured on MI355X / ROCm 7.2: replaying
        the three-stream step costs 1.73 ms at B=1 and an erratic 2.2-3.6 ms at B=2 (a single-stream capture: 3.4 ms,
        ~25 us per node) against 1.60 / 1.79 ms for eager launches once the per-launch Python cost was trimmed
        (raw stream handle, pointer checks); at B >= 3 the step is GPU-bound either way (3.39 vs 4.0 ms)."""
        return self.graphs_enabled() and os.environ.get('CLSLAM_HIPGRAPH', 'auto') == '1'

    def train_step_graphed(self, inputs: Dict[Any, torch.Tensor], *, sample_w: torch.Tensor, smooth_w: Optional[torch.Tensor],
                           noise: Optional[Dict[int, torch.Tensor]], copy_inputs: bool = True,
                           reuse_frozen: bool = False, want_outputs: bool = True):
        """forward(train=True) + backward() through a captured hipGraph (captured on first use per
        batch size / noise mode; a second, encoder-free graph serves the reuse_frozen steps).  Returns
        (outputs, losses) as fresh tensors."""
        B = inputs['rgb_aug', 0, 0].shape[0]
        key = (B, noise is not None, 0 if smooth_w is None else int(smooth_w.numel()))
        st = self._graphs.get(key)
        ws = self.workspace(B)
        first = st is None
        if first:
            st = SimpleNamespace(graphs={}, inputs={}, sample_w=sample_w.clone(),
                                 smooth_w=None if smooth_w is None else smooth_w.clone())
            for k in self.GRAPH_KEYS:
                v = inputs[k]
                st.inputs[k] = (v.to(torch.float64) if k[0] == 'relative_distance' else self._img(v)).clone()
            self._graphs[key] = st
        # everything the replay reads goes into its static buffers with ONE launch (up to 15 + 2 + 4 torch copies
        # otherwise: at B <= 2 their launch gaps were ~8 % of the step)
        staged = [(sample_w, st.sample_w)]
        if smooth_w is not None:
            staged.append((smooth_w, st.smooth_w))
        if copy_inputs and not first:
            staged += [(inputs[k], st.inputs[k]) for k in self.GRAPH_KEYS]
        if noise is not None:
            staged += [(noise[s], ws.noise[s]) for s in range(4)]
        self._stage(staged)
        reuse = bool(reuse_frozen and not copy_inputs and self.reuse_frozen_features and getattr(ws, 'frozen_valid', False))
        entry = st.graphs.get(reuse)
        if entry is None:
            entry = SimpleNamespace(graph=None, outputs=None, losses=None)

            def run():
                out, losses = self.forward(st.inputs, train=True, sample_w=st.sample_w, smooth_w=st.smooth_w,
                                           noise=None, draw_noise=noise is None, keep_noise=noise is not None,
                                           reuse_frozen=reuse)
                self.backward(B)
                return out, losses
            fresh = self.fresh_outputs
            self.fresh_outputs = False          # the graph writes into static planes; copies are handed out
            self._capturing = True              # no asynchronous tail inside a captured step
            self.wait_training()
            try:
                cur = torch.cuda.current_stream(self.device)
                warm = torch.cuda.Stream(device=self.device)
                warm.wait_stream(cur)
                with torch.cuda.stream(warm):   # eager warm-up: allocates every workspace buffer
                    run()
                cur.wait_stream(warm)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.capture_stream):
                    entry.outputs, entry.losses = run()
                entry.graph = g
                st.graphs[reuse] = entry
            finally:
                self.fresh_outputs = fresh
                self._capturing = False
        self.wait_training()
        entry.graph.replay()
        ws.frozen_valid = True
        # the graph writes static planes: hand out copies (skipped for the intermediate steps of adapt(steps=S),
        # whose outputs the caller drops -- dpp.py:309-319 returns the last step's only)
        losses = torch.empty_like(entry.losses)
        handed = [(entry.losses, losses)]
        outputs: Dict[Any, torch.Tensor] = {}
        if want_outputs:
            for k, v in entry.outputs.items():
                outputs[k] = torch.empty(v.shape, dtype=v.dtype, device=v.device)
                handed.append((v, outputs[k]))
        self._stage(handed)
        return outputs, losses

    @staticmethod
    def _stage(pairs) -> None:
        """(src, dst) copies on the current stream: one clslam_copy_multi launch for the plain ones, torch's
        copy_ for anything that needs a conversion (dtype, layout, host source)."""
        plain = []
        for src, dst in pairs:
            if (src.dtype == dst.dtype and src.device == dst.device and src.shape == dst.shape and src.is_contiguous()
                    and dst.is_contiguous()):
                plain.append((src, dst))
            elif src.shape == dst.shape or src.numel() == dst.numel():
                dst.copy_(src.reshape(dst.shape), non_blocking=True)
            else:
                raise ClslamError(f'cannot stage {tuple(src.shape)} into {tuple(dst.shape)}')
        ops.copy_multi(plain)

    # ------------------------------------------------------------------------------------------
    def adam(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, guard: Optional[torch.Tensor] = None) -> None:
        """guard: the step's loss (1 element); a NaN there makes the kernel skip the update."""
        self.adam_step_count += 1
        if self._tail_open:
            if guard is not None:
                guard.record_stream(self.tail_stream)      # a fresh per-call tensor allocated on the caller's stream
            with torch.cuda.stream(self.tail_stream):
                ops.adam_step(self._w, self._g, self._m, self._v, lr, self.adam_step_count, betas[0], betas[1], eps, guard=guard)
                self._tail_event = torch.cuda.Event()
                self._tail_event.record(self.tail_stream)
            self._tail_open = False
        else:
            self.wait_training()
            if self._pending_reduce is not None:
                table, nitems = self._pending_reduce
                self._pending_reduce = None
                ops.reduce_multi_adam(table, nitems, self._g, self._w, self._m, self._v, lr, self.adam_step_count, betas[0], betas[1],
                                      eps, guard=guard)
            else:
                ops.adam_step(self._w, self._g, self._m, self._v, lr, self.adam_step_count, betas[0], betas[1], eps, guard=guard)
            self._tail_event = None
        self._modules_stale = True

    # ------------------------------------------------------------------------------------------
    # module-level entry points (slam/slam.py:146 calls models['depth_encoder'](img)[4])
    def run_encoder(self, which: str, x: torch.Tensor) -> List[torch.Tensor]:
        """NCHW feature list of one encoder, as the reference's ResnetEncoder.forward returns it."""
        self.pack_if_needed()
        self._conv_workspace()
        self.wait_training()
        x = self._img(x.to(self.device))
        n = x.shape[0]
        nimg = 1 if which == 'depth_encoder' else 2
        if x.shape[1] != 3 * nimg or x.shape[2] != self.H or x.shape[3] != self.W:
            raise ClslamError(f'{which} input must be (N,{3 * nimg},{self.H},{self.W}), got {tuple(x.shape)}')
        key = ('enc', which, n)
        bufs = self._ws.get(key)
        if bufs is None:
            bufs = self._enc_bufs(n)
            self._ws[key] = bufs
        if nimg == 1:
            stem = [(x, None, 0, n)]
        else:
            stem = [(x[:, :3].contiguous(), x[:, 3:].contiguous(), 0, n)]
        # the share of the chip the same encoder gets inside forward() for this many images: the persistent kernels' summation
        # order (and the Winograd / direct pick) depends on it, and models['depth_encoder'](x) is bitwise what predict() computes
        # (tests/test_slam_usage.py::test_the_four_models_are_callables)
        cu = self.cu_limit(n, 'enc_depth') if nimg == 1 else self.cu_limit(max(1, n // 2), 'enc_pose')
        feats = self._encoder(self.enc[which], bufs, n, stem, cu_limit=cu)
        self._memo = None
        if which == 'depth_encoder' and n == 1 and self.descriptor_memo:
            # (a private copy of the input: the caller owns x and may overwrite it; the feature buffers belong to this
            # workspace and are only rewritten by the next call of this function, which replaces the memo)
            self._memo = SimpleNamespace(x=x.clone(), feats=feats, version=self._packed_version)
        return [f.permute(0, 3, 1, 2) for f in feats]

    def run_pose(self, image_0: torch.Tensor, image_1: torch.Tensor) -> torch.Tensor:
        """predict_pose (dpp.py:628-664): pose_encoder(cat(img0,img1)) -> pose_decoder; returns (n,12)."""
        self.pack_if_needed()
        self._conv_workspace()
        self.wait_training()
        a, b = self._img(image_0.to(self.device)), self._img(image_1.to(self.device))
        n = a.shape[0] if a.dim() == 4 else -1
        # the stem kernel takes H, W from its input but writes buffers planned for the engine's resolution
        for name, t in (('image_0', a), ('image_1', b)):
            if tuple(t.shape) != (n, 3, self.H, self.W):
                raise ClslamError(f'predict_pose: {name} must be (N, 3, {self.H}, {self.W}) with equal N, got {tuple(t.shape)}')
        key = ('pose', n)
        st = self._ws.get(key)
        if st is None:
            E = lambda *s: torch.empty(*s, device=self.device)  # noqa: E731
            h5, w5 = self.H >> 5, self.W >> 5
            st = SimpleNamespace(penc=self._enc_bufs(n), sq=E(n, h5, w5, 256), p0=E(n, h5, w5, 256), p1=E(n, h5, w5, 256),
                                 pmean=E(n, 256), pose=E(n, 12))
            self._ws[key] = st
        self._cu_enc = self._cu_dec = 0
        feats = self._encoder(self.enc['pose_encoder'], st.penc, n, [(a, b, 0, n)])
        self._pose_decoder(st, feats[4])
        return st.pose

    def _nhwc(self, f: torch.Tensor) -> torch.Tensor:
        """(N,C,h,w) feature -> NHWC fp32 storage on the device (a view when it already is one: run_encoder's outputs are)"""
        f = f.to(self.device, torch.float32)
        return f.permute(0, 2, 3, 1).contiguous()

    def run_depth_decoder(self, input_features):
        """models['depth_decoder'](features) (dpp.py:931-936 calls it on the encoder's five features; networks/depth_decoder.py:
        51-71): {('disp', s): (N,1,H>>s,W>>s)} for s = 3..0 from the CURRENT decoder weights, by the same kernels as the step's
        forward.  Inference only: no graph is recorded and nothing of a training step in flight is touched."""
        self._cu_dec = self.cu_limit(0, 'decoders')
        self.pack_if_needed()
        self._conv_workspace()
        self.wait_training()
        feats = [self._nhwc(f) for f in input_features]
        n = feats[0].shape[0]
        for k, f in enumerate(feats):
            want = (n, self.H >> (k + 1), self.W >> (k + 1), NUM_CH_ENC[k])
            if tuple(f.shape) != want:
                raise ClslamError(f'depth_decoder: feature {k} must be (N, {want[3]}, {want[1]}, {want[2]}), got '
                                  f'{tuple(input_features[k].shape)} (the engine is planned for {self.H}x{self.W} inputs)')
        key = ('depth_decoder', n)
        st = self._ws.get(key)
        if st is None:
            E = lambda *s: torch.empty(*s, device=self.device)  # noqa: E731
            st = SimpleNamespace(x={}, disp=[E(n, self.H >> s, self.W >> s) for s in range(4)])
            for i in range(4, -1, -1):
                st.x[i, 0] = E(n, self.H >> (i + 1), self.W >> (i + 1), NUM_CH_DEC[i])
                st.x[i, 1] = E(n, self.H >> i, self.W >> i, NUM_CH_DEC[i])
            self._ws[key] = st
        main, self._main = self._main, None          # no side-stream fork: a plain chain on the caller's stream
        try:
            self._depth_decoder(st, feats)
        finally:
            self._main = main
        return {('disp', s): st.disp[s].unsqueeze(1).clone() for s in range(3, -1, -1)}

    def run_pose_decoder(self, last_features):
        """models['pose_decoder']([features]) (dpp.py:957-965; networks/pose_decoder.py:37-54): (axis_angle, translation), each
        (N, 2, 1, 3), from the last feature map of ONE pose-encoder pass."""
        self._cu_dec = self.cu_limit(0, 'decoders')
        if len(last_features) != 1:
            raise ClslamError('pose_decoder: the MI355X-native path implements num_input_features = 1')
        self.pack_if_needed()
        self._conv_workspace()
        self.wait_training()
        f4 = self._nhwc(last_features[0])
        n = f4.shape[0]
        want = (n, self.H >> 5, self.W >> 5, NUM_CH_ENC[4])
        if tuple(f4.shape) != want:
            raise ClslamError(f'pose_decoder: the feature must be (N, 512, {want[1]}, {want[2]}), got {tuple(last_features[0].shape)}')
        key = ('pose_decoder', n)
        st = self._ws.get(key)
        if st is None:
            E = lambda *s: torch.empty(*s, device=self.device)  # noqa: E731
            st = SimpleNamespace(sq=E(n, want[1], want[2], 256), p0=E(n, want[1], want[2], 256), p1=E(n, want[1], want[2], 256),
                                 pmean=E(n, 256), pose=E(n, 12))
            self._ws[key] = st
        self._pose_decoder(st, f4)
        out = st.pose.clone().view(-1, 2, 1, 6)          # the 0.01 scale is applied by the head kernel (pose_decoder.py:50)
        return out[..., :3], out[..., 3:]

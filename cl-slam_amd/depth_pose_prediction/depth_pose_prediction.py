"""MI355X-native ``DepthPosePrediction``: the reference's predictor API
(depth_pose_prediction/depth_pose_prediction.py, "dpp.py" below) on hand-written HIP kernels.

What is kept identical so that slam/slam.py and main_adapt.py run unchanged:
constructor signature + validation (dpp.py:39-127), ``models`` dict of four modules with the
reference's state-dict keys, ``adapt`` / ``predict`` / ``predict_pose`` signatures and return
structures (dpp.py:291-319, 530-536, 628-664), ``_set_eval/_set_train/_set_adapt``,
``save_model`` / ``load_model`` file layout incl. ``optimizer.pth`` (dpp.py:669-749), attributes
``device, optimizer, lr_scheduler, epoch, is_trained, batch_size, height, width, scales, frame_ids,
min_depth, max_depth, log_path``; the caller's input dict is moved to the device in place
(dpp.py:916-917) and ``RuntimeError('NaN loss')`` is raised like dpp.py:1115-1118.

What is different by design: all arithmetic runs in ``clslam_hip.engine.Engine`` (no autograd, no
torch.nn compute); there is NO CPU path -- constructing the predictor without a GPU / without
libclslam_hip.so raises.  Offline pre-training / evaluation / plotting (dpp.py:219-289, 321-526,
538-626, 829-904, 1197-1267) is outside the accelerated path and not provided.
"""
import math
import shutil
import os
import warnings
from pathlib import Path
from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
import torch
from torch import Tensor, optim

from clslam_hip.engine import Engine, TrainableLayout
from depth_pose_prediction.config import DepthPosePrediction as Config
from depth_pose_prediction.networks import DepthDecoder, PoseDecoder, ResnetEncoder


class EngineAdam(optim.Adam):
    """torch.optim.Adam facade over the engine's fused Adam kernel (clslam_adam_step).

    It IS a torch Adam as far as ``param_groups`` / ``StepLR`` / ``state_dict`` layout go (160 params
    in model-dict order, state for the 36 trainable ones: ids 62-89 and 152-159), but
    ``step`` runs one kernel over the flat arena and the moments live in the engine."""

    def __init__(self, params, lr: float, engine: Engine, names) -> None:
        super().__init__(params, lr)
        self._engine = engine
        self._names = names  # 'model/key' per param index
        self.loss_guard = None  # 1-element loss tensor of the step in flight: NaN -> the kernel skips the update

    def zero_grad(self, set_to_none: bool = True) -> None:  # gradients are overwritten, never accumulated
        return None

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        self._engine.adam(g['lr'], g['betas'], g['eps'], guard=self.loss_guard)

    def state_dict(self):
        sd = super().state_dict()
        eng = self._engine
        eng.wait_training()
        state = {}
        if eng.adam_step_count > 0:
            for idx, name in enumerate(self._names):
                if name not in eng.layout.offset:
                    continue
                shape = next(s for n, _, s in eng.layout.entries if n == name)
                off, n = eng.layout.offset[name], math.prod(shape)
                state[idx] = {
                    'step': torch.tensor(float(eng.adam_step_count)),
                    'exp_avg': TrainableLayout.to_reference(eng._m[off:off + n], shape).clone(),
                    'exp_avg_sq': TrainableLayout.to_reference(eng._v[off:off + n], shape).clone(),
                }
        sd['state'] = state
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict) -> None:
        eng = self._engine
        eng.wait_training()
        groups = state_dict['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self._names):
            raise ValueError('loaded state dict contains a parameter group that does not match the size of '
                             "optimizer's group")
        for k in ('lr', 'betas', 'eps'):
            if k in groups[0]:
                self.param_groups[0][k] = groups[0][k]
        eng._m.zero_()
        eng._v.zero_()
        step = 0
        for idx, st in state_dict['state'].items():
            name = self._names[int(idx)]
            if name not in eng.layout.offset:
                continue
            off = eng.layout.offset[name]
            m = TrainableLayout.to_compute(st['exp_avg'].to(eng.device, torch.float32))
            v = TrainableLayout.to_compute(st['exp_avg_sq'].to(eng.device, torch.float32))
            eng._m[off:off + m.numel()].copy_(m)
            eng._v[off:off + v.numel()].copy_(v)
            step = max(step, int(float(st['step'])))
        eng.adam_step_count = step


class DataParallelPeerFailure(RuntimeError):
    """adapt(): another rank of the data-parallel group failed before this step's exchange.  Raised on EVERY healthy rank at
    the same step (the failed rank completes the step's collectives with a status word, see _dp_abort_step); the optimizer
    step was applied on none of them, so the replicas are still identical."""


class DepthPosePrediction:
    def __init__(self, dataset_config, config: Config, use_online: bool = False, reference_quirks: bool = True,
                 host_pose_output: Optional[bool] = None, upload_all_inputs: Optional[bool] = None):
        # Initialize parameters (dpp.py:41-68) ===========
        self.config_file = config.config_file
        self.dataset_type = dataset_config.dataset
        self.dataset_path = dataset_config.dataset_path
        self.height = dataset_config.height
        self.width = dataset_config.width
        self.train_set = config.train_set
        self.val_set = config.val_set
        self.resnet_depth = config.resnet_depth
        self.resnet_pose = config.resnet_pose
        self.resnet_pretrained = config.resnet_pretrained
        self.scales = config.scales
        self.learning_rate = config.learning_rate
        self.scheduler_step_size = config.scheduler_step_size
        self.batch_size = config.batch_size
        self.num_workers = config.num_workers
        self.num_epochs = config.num_epochs
        self.min_depth = config.min_depth
        self.max_depth = config.max_depth
        self.disparity_smoothness = config.disparity_smoothness
        self.velocity_loss_scaling = config.velocity_loss_scaling
        self.mask_dynamic = config.mask_dynamic
        self.log_path = config.log_path
        self.save_frequency = config.save_frequency
        self.save_val_depth = config.save_val_depth
        self.save_val_depth_batches = config.save_val_depth_batches
        self.multiple_gpus = config.multiple_gpus
        self.gpu_ids = config.gpu_ids
        self.load_weights_folder = config.load_weights_folder
        self.use_wandb = False

        self.is_trained = False
        self.frame_ids = (0, -1, 1)
        self.num_pose_frames = 2
        self.device = _select_device()

        # Dependent parameters (dpp.py:79-127), same checks and messages ==================
        if self.load_weights_folder is not None:
            self.load_weights_folder = Path(self.load_weights_folder).absolute()
        if isinstance(self.train_set, list):
            self.train_set = tuple(self.train_set)
        if isinstance(self.val_set, list):
            self.val_set = tuple(self.val_set)
        if dataset_config.dataset == 'Kitti':
            if isinstance(self.val_set, int):
                self.val_set = (self.val_set,)
            if isinstance(self.train_set, str):
                if self.train_set != 'all':
                    raise ValueError('train_set of KITTI only accepts these strings: ["all"]')
                self.train_set = tuple(s for s in range(11) if s not in self.val_set and s != 3)
            elif isinstance(self.train_set, int):
                self.train_set = (self.train_set,)
            if not (isinstance(self.train_set, tuple) and isinstance(self.train_set[0], int)):
                raise ValueError('Passed invalid value for train_set')
            if not (isinstance(self.val_set, tuple) and isinstance(self.val_set[0], int)):
                raise ValueError('Passed invalid value for val_set')
        elif dataset_config.dataset in ['Cityscapes', 'RobotCar']:
            if isinstance(self.train_set, str):
                self.train_set = (self.train_set,)
            if isinstance(self.val_set, str):
                self.val_set = (self.val_set,)
            if not (isinstance(self.train_set, tuple) and isinstance(self.train_set[0], str)):
                raise ValueError('Passed invalid value for train_set')
            if not (isinstance(self.val_set, tuple) and isinstance(self.val_set[0], str)):
                raise ValueError('Passed invalid value for val_set')
        if self.multiple_gpus:
            # the reference wraps the nets in single-process nn.DataParallel (dpp.py:178-181, pre-training
            # only, disabled in config_adapt.yaml:31); here multi-GPU is one process per GPU, see
            # enable_data_parallel()
            raise ValueError('multiple_gpus (nn.DataParallel) is not used on MI355X: launch one process per GPU and '
                             'call enable_data_parallel()')
        if self.gpu_ids is not None and len(self.gpu_ids) > 1:
            raise ValueError('Passed multiple GPU IDs without activating multi-GPU support.')
        if self.gpu_ids is None:
            self.gpu_ids = (self.device.index or 0,)
        if use_online:
            raise ValueError('use_online (dual network) is never enabled by the reference SLAM driver '
                             '(slam/slam.py:39) and is not part of the accelerated path')
        if self.mask_dynamic:
            raise ValueError('mask_dynamic=True is the pre-training configuration; the accelerated path implements '
                             'the adaptation configuration (mask_dynamic=False, config_adapt.yaml:26)')
        if tuple(self.scales) != (0, 1, 2, 3):
            raise ValueError('The accelerated path implements scales=(0, 1, 2, 3)')
        # =================================================

        # Networks (parameter containers) + engine ========
        self.models = {}
        self.models['depth_encoder'] = ResnetEncoder(self.resnet_depth, self.resnet_pretrained)
        self.models['depth_decoder'] = DepthDecoder(self.models['depth_encoder'].num_ch_encoder, self.scales)
        self.models['pose_encoder'] = ResnetEncoder(self.resnet_pose, self.resnet_pretrained, self.num_pose_frames)
        self.models['pose_decoder'] = PoseDecoder(self.models['pose_encoder'].num_ch_encoder, num_input_features=1,
                                                  num_frames_to_predict_for=2)
        self.use_online = False
        self.online_models = {}
        self.engine = Engine(self.height, self.width, self.device, min_depth=self.min_depth, max_depth=self.max_depth,
                             disparity_smoothness=self.disparity_smoothness,
                             velocity_loss_scaling=self.velocity_loss_scaling,
                             reference_quirks=reference_quirks)   # False: opt-in per-sample smoothness
        for m in self.models.values():
            m.to(self.device)
        self.engine.bind(self.models)
        self.engine.pack()

        self.parameters_to_train = []
        names = []
        for model_name, m in self.models.items():
            for n, p in torch.nn.Module.named_parameters(m):
                self.parameters_to_train.append(p)
                names.append(f'{model_name}/{n}')
        self.online_parameters_to_train = []

        # Optimizer (dpp.py:203-210) ======================
        self.optimizer = EngineAdam(self.parameters_to_train, self.learning_rate, self.engine, names)
        self.lr_scheduler = optim.lr_scheduler.StepLR(self.optimizer, self.scheduler_step_size, 0.1)
        self.epoch = 0
        self.online_optimizer = None
        self.train_loader, self.val_loader = None, None

        self._dp = None
        self._default_weights: Dict[Any, Any] = {}
        self._injected_noise = None
        self._loss_host = None   # pinned 18-float staging buffer + event: the step's loss scalars (NaN check, return value)
        self._loss_event = None
        self._mode = None
        # host -> device upload of a minibatch (dpp.py:916-917): EVERY tensor of the caller's dict is moved in place like the
        # reference does, on a copy stream, in the order the step needs them: the three network inputs first (the encoders
        # start while the rest is still crossing PCIe), then the ten other entries the path reads, then the entries it does not
        # read (15 of the 24 image planes, poses, ...), which nothing on the step waits for.  upload_all_inputs=False /
        # CLSLAM_UPLOAD_ALL=0 leaves those on the host (a deviation from dpp.py:916-917: opt-in).
        if upload_all_inputs is None:
            upload_all_inputs = os.environ.get('CLSLAM_UPLOAD_ALL', '1') != '0'
        self.upload_all_inputs = bool(upload_all_inputs)
        self._copy_stream = None
        # Opt-in fast path (host_pose_output=True / CLSLAM_HOST_POSE=1): adapt(online, training) hands out
        # outputs['cam_T_cam', 0, +-1] AND the loss dict as HOST tensors, staged behind the forward before the backward is
        # enqueued (the loss scalars are staged in any case for the NaN check of dpp.py:1115-1118).  The SLAM driver reads
        # exactly these back after every frame (slam.py:181-188: `[0, :]`, linalg.inv, `.cpu()` per key): on device tensors
        # the first `.cpu()` is ordered behind the whole backward + optimizer step on the stream, so the host -- and with it
        # the next frame's upload and forward launches -- stands still for the rest of the step (0.37 ms of an end-to-end
        # frame at B = 5).  The DEFAULT is the reference's behaviour: every output and every loss is a device tensor in all
        # three modes (training adapt, adapt(online, None), predict()).
        if host_pose_output is None:
            host_pose_output = os.environ.get('CLSLAM_HOST_POSE', '0') == '1'
        self.host_pose_output = bool(host_pose_output)
        self._pose_host: Dict[int, Tensor] = {}
        self._pose_staged = None
        self._dp_tags: Dict[Any, Tensor] = {}
        self._dp_tag_staged = False
        self._dp_ext_cpu = None
        self._dp_posted = 0      # collectives of the current training step that have gone out (agreed failure, _dp_abort_step)

    # ============================================================
    # Data-parallel replay minibatch (new functionality; the reference has no multi-GPU adaptation)

    def enable_data_parallel(self, global_batch_size: int, shard_offset: int, process_group=None) -> None:
        """One process per GPU (torch.distributed, backend 'nccl' = RCCL).  Every rank calls
        ``adapt`` with ITS contiguous shard of the (online + replay) minibatch; the rank whose shard
        starts at 0 holds the online sample.  Loss terms use the global batch size, the flat
        gradient arena is sum-all-reduced once per step, Adam runs identically on every rank."""
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self._dp = dict(group=process_group, global_batch=int(global_batch_size), offset=int(shard_offset), dist=dist)
        # overlap the gradient all-reduce + Adam with the next step's (frozen) encoders
        self.engine.async_tail = os.environ.get('CLSLAM_ASYNC_TAIL', '1') != '0'
        self.engine.data_parallel = True
        self.engine.noise_stream = int(shard_offset)     # identically seeded ranks draw different tie-break fields

    def wait_training(self, stream=None) -> None:
        """Order `stream` (default: torch's current stream) behind the training step in flight.  A training adapt() returns once
        the caller's stream is ordered behind the forward and the last reads of its inputs; backward + optimizer step continue
        on the engine's own stream (Engine.main_stream -- ONE per device and process: the steps of several predictors serialise
        there).  Everything of this class that touches trainable state waits by itself; a caller that fences or times the step
        with events on ITS stream calls this first (or synchronize())."""
        self.engine.wait_training(stream)

    def synchronize(self) -> None:
        """Block the host until the training step in flight (incl. its optimizer step and, in data-parallel mode, the gradient
        exchange) has completed."""
        self.engine.wait_training()
        if self.device.type == 'cuda':
            torch.cuda.current_stream(self.device).synchronize()

    def gather_outputs(self, outputs: Dict[Any, Tensor]) -> Dict[Any, Tensor]:
        """Data-parallel mode: every rank's `outputs` shard concatenated in rank (= sample) order on every
        rank -- the full-batch dict a single process would have returned.  slam.py only reads sample 0,
        which rank 0 already holds, so nothing on the per-frame path calls this; it is the explicit
        collective for callers that want all samples.  Shards may be uneven."""
        if self._dp is None:
            return outputs
        dist, group = self._dp['dist'], self._dp['group']
        world = dist.get_world_size(group)
        n_local = next(iter(outputs.values())).shape[0]
        counts = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([n_local], dtype=torch.int64, device=self.device), group=group)
        counts = [int(c) for c in counts]
        if sum(counts) != self._dp['global_batch']:
            raise RuntimeError(f'shards {counts} do not add up to the global batch {self._dp["global_batch"]}')
        cap = max(counts)
        full: Dict[Any, Tensor] = {}
        for key, val in outputs.items():
            val = val.contiguous()
            if n_local < cap:        # all_gather wants equal shapes: pad the short shards
                val = torch.cat([val, val.new_zeros((cap - n_local,) + tuple(val.shape[1:]))])
            parts = [torch.empty_like(val) for _ in range(world)]
            dist.all_gather(parts, val, group=group)
            full[key] = torch.cat([part[:c] for part, c in zip(parts, counts)])
        return full

    def replicas_in_sync(self) -> bool:
        """Data-parallel mode: True when every rank holds bit-identical trainable weights and Adam moments
        (they should: same all-reduced gradient, same update).  Two tiny all-reduces of a checksum triple;
        meant to be called every N frames, not per step."""
        if self._dp is None:
            return True
        self.engine.wait_training()
        e = self.engine
        # order-dependent checksums of the raw bit patterns: a single flipped bit anywhere changes them
        def digest(t):
            bits = t.view(torch.int32).to(torch.int64)
            idx = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64)
            return torch.stack([bits.sum(), (bits * (idx % 8191 + 1)).sum()])
        mine = torch.cat([digest(e._w), digest(e._m), digest(e._v)])
        lo, hi = mine.clone(), mine.clone()
        dist, group = self._dp['dist'], self._dp['group']
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        return bool(torch.equal(lo, hi))

    def set_tie_break_noise(self, noise: Optional[Dict[int, Tensor]]) -> None:
        """Parity testing only: inject the per-scale tie-break tensors (B,2,H,W) that the reference
        draws with torch.randn(...) * 1e-5 (dpp.py:1055-1056).  None -> draw on the device."""
        self._injected_noise = noise

    # ============================================================
    # Training / validation: offline, outside the accelerated path

    def train(self, *args, **kwargs) -> None:
        raise NotImplementedError('offline pre-training (dpp.py:219-289) is outside the MI355X hot path; '
                                  'pre-train with the reference and load the checkpoint here')

    def adapt(self,
              online_data: Dict[Any, Tensor],
              training_data: Optional[Dict[Any, Tensor]] = None,
              online_index: int = 0,
              steps: int = 1,
              online_loss_weight: Optional[float] = None):
        # dpp.py:297-305
        if online_loss_weight is None:
            loss_weights = None
        elif self.batch_size == 1:
            loss_weights = torch.ones(1, device=self.device)
        else:
            loss_weights = torch.empty(self.batch_size, device=self.device)
            buffer_loss_weight = (1 - online_loss_weight) / (self.batch_size - 1)
            loss_weights[:] = buffer_loss_weight
            loss_weights[online_index] = online_loss_weight

        if training_data is not None:
            self._set_adapt(freeze_encoder=True)
            self.engine.pack_if_needed()
            if self._dp is not None and training_data['rgb_aug', 0, 0].shape[0] == 0:
                # data-parallel rank WITHOUT samples (the replay buffer is still filling up: the global minibatch is smaller
                # than the number of ranks, slam/replay_buffer.py): it takes part in every collective with zeros and applies
                # the same optimizer step, so the replicas stay identical
                return self._empty_shard_steps(steps)
            # The step runs on the ENGINE's stream (Engine.main_stream): forward, backward and the optimizer step go out
            # behind each other there, and the caller's stream is only ordered behind the last forward + the first three
            # launches of its backward (the last reads of the caller's minibatch and of the output planes: ~0.1 ms) -- the
            # reads the reference's driver does next (slam.py:181-188: cam_T_cam[0].cpu(), one .cpu() per loss key) return
            # then, while the rest of the backward + Adam keep the GPU busy, and the caller may overwrite or drop its
            # minibatch at once like after the reference's synchronous step.  Everything that touches trainable state or
            # the workspace afterwards (the next adapt() / predict() / predict_pose(), state_dict(), save_model(),
            # engine.w/g/m/v) is ordered behind the step.  CLSLAM_DETACHED_TRAINING=0: the whole step on the caller's stream.
            eng = self.engine
            cur = em = released = None
            failed = True
            self._dp_posted = 0
            try:
                try:
                    if eng.detached_ok():
                        # (inside the try: a failed allocation in prealloc_outputs must not leave the engine in detached state)
                        cur, em = eng.begin_detached()
                        if em is not None:      # still on the caller's stream: the output planes of the call's last forward
                            eng.prealloc_outputs(training_data['rgb_aug', 0, 0].shape[0])
                    with (torch.cuda.stream(em) if em is not None else _null_context()):
                        for it in range(steps):
                            self._dp_posted = 0
                            eng._prealloc_armed = it == steps - 1      # the planes this call hands out: the caller's pool (below)
                            # steps 2..S see the same minibatch through the same frozen, eval-mode encoders (dpp.py:308-313):
                            # their features and the identity-reprojection maps are kept, only the decoders re-run
                            if eng.graph_preferred(training_data['rgb_aug', 0, 0].shape[0]):
                                # forward + backward replayed as one hipGraph (same kernels, same streams)
                                outputs_eval, losses = self._process_batch(training_data, loss_weights, train=True, graphed=True,
                                                                           copy_inputs=(it == 0), reuse_frozen=(it > 0),
                                                                           want_outputs=(it == steps - 1))
                                self.optimizer.zero_grad()
                                self._reduce_gradients()
                            else:
                                outputs_eval, losses = self._process_batch(training_data, loss_weights, train=True,
                                                                           reuse_frozen=(it > 0))
                                self.optimizer.zero_grad()
                                self._backward(training_data)
                                released = eng.inputs_released if it == steps - 1 else None
                            # dpp.py:1115-1118 aborts on a NaN loss before backward/step.  Checking there costs a full host
                            # sync in the middle of the step (the GPU idles while the host enqueues the backward), and a
                            # sync at the end of the step starves the GPU at the start of the next one.  Instead the
                            # forward copies the loss to pinned host memory behind an event; backward and a device-
                            # guarded Adam (no-op when the loss is NaN) are enqueued, and only then does the host wait
                            # for THAT event -- the GPU still has the whole backward queued while the host goes on.
                            self.optimizer.loss_guard = self._losses_dev[17:18]
                            self.optimizer.step()
                            self.optimizer.loss_guard = None
                            try:
                                losses = self._staged_losses()
                            except DataParallelPeerFailure:
                                self.engine.adam_step_count -= 1     # the guarded launch saw the peer's NaN: nothing was applied
                                raise
                            self._raise_on_nan(losses, undo_step=True)
                            if not self.host_pose_output:        # reference behaviour: the loss dict lives on the device
                                losses = self.engine.losses_dict(self._losses_dev)
                    failed = False
                finally:
                    if em is not None:
                        eng.end_detached(cur, released, failed=failed)
                    else:
                        eng._caller, eng._prealloc, eng._prealloc_armed = None, None, False
            except DataParallelPeerFailure:
                raise
            except Exception as exc:
                # A failure of THIS rank before any collective of the step went out (a malformed minibatch, a failed allocation):
                # the peers are about to wait in the loss exchange.  Complete the step's collectives with the status word set, so
                # that every rank raises now instead of sitting in its communicator's timeout.
                if self._dp is not None and self._dp_posted == 0:
                    self._dp_abort_step()
                    exc.dp_agreed = True
                raise
            if self._pose_staged is not None:
                # the event _staged_losses() waited for covers the pose copy issued just before the loss copy
                T = self._pose_host[self._pose_staged].clone()
                outputs_eval['cam_T_cam', 0, -1], outputs_eval['cam_T_cam', 0, 1] = T[0], T[1]
                self._pose_staged = None
        else:
            self._set_eval()
            self.engine.pack_if_needed()
            outputs_eval, losses = self._process_batch(online_data, loss_weights, train=False)
        return outputs_eval, losses

    def _dp_abort_step(self) -> None:
        """The collectives of ONE training step, posted by a rank that cannot run it: the loss exchange carries NaN (every
        rank's device-guarded Adam launch becomes a no-op, like dpp.py:1115-1118's abort) and the status word of _dp_tag, the
        gradient exchange carries zeros.  The healthy ranks raise DataParallelPeerFailure behind their step."""
        eng, dist, group = self.engine, self._dp['dist'], self._dp['group']
        eng.wait_training()
        ext = torch.cat([torch.full((18,), float('nan'), device=self.device), self._dp_tag(0)])
        ext[21] = 1.0
        dist.all_reduce(ext, group=group)
        eng._g.zero_()
        if eng.grad_buckets > 1 and not eng.graph_preferred(1):
            for _name, lo, hi in eng.bucket_ranges():
                dist.all_reduce(eng._g[lo:hi], group=group)
        else:
            dist.all_reduce(eng._g, group=group)
        eng._g.zero_()
        eng.grads_synced = False

    def _empty_shard_steps(self, steps: int):
        """adapt() of a data-parallel rank whose shard is empty: the same sequence of collectives as a rank with samples
        (18 loss scalars, then the gradient arena -- in the engine's buckets, or whole on the hipGraph path), contributing
        zeros; the NaN check on the all-reduced loss and the guarded Adam step like everywhere else."""
        eng, dist, group = self.engine, self._dp['dist'], self._dp['group']
        eng.wait_training()
        bucketed = eng.grad_buckets > 1 and not eng.graph_preferred(1)
        for _ in range(steps):
            ext = torch.cat([torch.zeros(18, device=self.device), self._dp_tag(0)])
            dist.all_reduce(ext, group=group)
            losses = ext[:18]
            self._losses_dev = losses
            # Same order as a rank WITH samples (and as _dp_abort_step): the step's gradient collective(s) and the guarded
            # optimizer launch go out FIRST, the status word is looked at afterwards -- a rank that raised right behind the loss
            # exchange would leave its peers' gradient all-reduce unmatched (they post it before they check, _staged_losses).
            eng._g.zero_()
            if bucketed:
                for _name, lo, hi in eng.bucket_ranges():
                    self._allreduce(eng._g[lo:hi])
            else:
                dist.all_reduce(eng._g, group=group)
            self.optimizer.loss_guard = losses[17:18]
            self.optimizer.step()
            self.optimizer.loss_guard = None
            host = ext.detach().cpu()
            try:
                self._check_dp_tag(host[18:22])
            except DataParallelPeerFailure:
                eng.adam_step_count -= 1     # the guarded launch saw the peer's NaN: nothing was applied
                raise
            self._raise_on_nan(eng.losses_dict(host[:18]), undo_step=True)
        H, W, dev = self.height, self.width, self.device
        E = lambda *shape: torch.empty(*shape, device=dev)  # noqa: E731
        outputs: Dict[Any, Tensor] = {}
        for s in (3, 2, 1, 0):
            outputs['disp', s] = E(0, 1, H >> s, W >> s)
        for f in (-1, 1):
            outputs['axis_angle', 0, f], outputs['translation', 0, f], outputs['cam_T_cam', 0, f] = E(0, 1, 3), E(0, 1, 3), E(0, 4, 4)
        for s in range(4):
            outputs['depth', s] = E(0, 1, H, W)
            for f in (-1, 1):
                outputs['rgb', f, s] = E(0, 3, H, W)
        return outputs, eng.losses_dict(losses)

    # ============================================================
    # Predict functions

    def predict(self, batch) -> Dict[Any, Tensor]:
        if not self.is_trained:
            warnings.warn('The model has not been trained yet.', RuntimeWarning)
        self._set_eval()
        self.engine.pack_if_needed()
        outputs, _ = self._process_batch(batch, train=False)
        return outputs

    def predict_pose(
            self,
            image_0: Tensor,
            image_1: Tensor,
            as_numpy: bool = True,
            use_online: bool = False,
    ) -> Tuple[Union[Tensor, np.ndarray], Union[Tensor, np.ndarray]]:
        if not self.is_trained:
            warnings.warn('The model has not been trained yet.', RuntimeWarning)
        if use_online:
            raise ValueError('use_online is not part of the accelerated path')
        if len(image_0.shape) == 3:
            image_0 = image_0.unsqueeze(dim=0)
        if len(image_1.shape) == 3:
            image_1 = image_1.unsqueeze(dim=0)
        self._set_eval()
        pose = self.engine.run_pose(image_0, image_1)           # (n,12), frame-0 slice = [:, :6]
        n = pose.shape[0]
        # invert=False (dpp.py:655): T = translate(t) * rot(axis_angle); frame idx 1 of pose_to_proj
        from clslam_hip import ops
        pad = torch.zeros(2 * n, 12, device=self.device)
        pad[n:] = pose
        eye = torch.eye(4, device=self.device).repeat(n, 1, 1)
        T = torch.empty(2, n, 4, 4, device=self.device)
        P = torch.empty(2, n, 3, 4, device=self.device)
        ops.pose_to_proj(pad, eye, T, P)
        transformation = T[1].clone()
        cov_matrix = torch.eye(6, device=self.device)
        if as_numpy:
            transformation = transformation.squeeze().cpu().detach().numpy()
            cov_matrix = cov_matrix.cpu().detach().numpy()
        return transformation, cov_matrix

    # ============================================================
    # Save / load (dpp.py:669-749): same files, same keys

    def save_model(self) -> None:
        save_folder = Path(self.log_path) / 'models' / f'weights_{self.epoch:03}'
        save_folder.mkdir(parents=True, exist_ok=True)
        for model_name, model in self.models.items():
            to_save = model.state_dict()
            if 'encoder' in model_name:
                # the reference stores Tensor(self.height) / Tensor(self.width): float tensors of
                # LENGTH height/width with unspecified content; same shape, zeros.
                to_save['height'] = torch.zeros(self.height)
                to_save['width'] = torch.zeros(self.width)
            torch.save(to_save, save_folder / f'{model_name}.pth')
        torch.save({'optimizer': self.optimizer.state_dict(), 'scheduler': self.lr_scheduler.state_dict()},
                   save_folder / 'optimizer.pth')
        if self.config_file is not None and Path(self.config_file).exists():
            shutil.copy(self.config_file, Path(self.log_path) / 'config.yaml')
        print(f'Saved model to: {save_folder}')

    def load_model(self, load_optimizer: bool = True) -> None:
        if self.load_weights_folder is None:
            print('Weights folder required to load the model is not specified.')
        if not self.load_weights_folder.exists():
            print(f'Cannot find folder: {self.load_weights_folder}')
        print(f'Load model from: {self.load_weights_folder}')
        for model_name, model in self.models.items():
            path = self.load_weights_folder / f'{model_name}.pth'
            pretrained_dict = torch.load(path, map_location=self.device)
            model_dict = model.state_dict()
            pretrained_dict = {k: v for k, v in pretrained_dict.items() if k in model_dict}
            if len(pretrained_dict.keys()) == 0:
                raise RuntimeError(f'No fitting weights found in: {path}')
            model_dict.update(pretrained_dict)
            model.load_state_dict(model_dict)
        self.is_trained = True
        self.engine.pack()
        if load_optimizer:
            optimizer_load_path = self.load_weights_folder / 'optimizer.pth'
            try:
                optimizer_dict = torch.load(optimizer_load_path, map_location=self.device)
                if 'optimizer' in optimizer_dict:
                    self.optimizer.load_state_dict(optimizer_dict['optimizer'])
                    self.lr_scheduler.load_state_dict(optimizer_dict['scheduler'])
                    self.epoch = self.lr_scheduler.last_epoch
                    print(f'Restored optimizer and LR scheduler (resume from epoch {self.epoch}).')
                else:
                    self.optimizer.load_state_dict(optimizer_dict)
                    print('Restored optimizer (legacy mode).')
            except Exception:  # pylint: disable=broad-except
                print('Cannot find matching optimizer weights, so the optimizer is randomly initialized.')

    def sync_weights(self) -> None:
        """Make direct attribute reads of module parameters see the adapted weights (state_dict(),
        parameters() and save_model() do this implicitly)."""
        self.engine.sync_modules()

    # ============================================================
    # Auxiliary functions (dpp.py:797-827)

    def _set_train(self) -> None:
        for m in self.models.values():
            m.train()
        self._mode = 'train'

    def _set_eval(self) -> None:
        if self._mode in ('eval', 'adapt'):   # both are eval() for the modules; walking them costs ~0.1 ms
            return
        for m in self.models.values():
            m.eval()
        self._mode = 'eval'

    def _set_adapt(self, freeze_encoder: bool = True) -> None:
        if not freeze_encoder:
            raise ValueError('the accelerated path implements freeze_encoder=True (dpp.py:308)')
        if self._mode == 'adapt':             # idempotent (requires_grad is never re-enabled)
            return
        for model_name, model in self.models.items():
            model.eval()
            for name, param in torch.nn.Module.named_parameters(model):
                if name.find('bn') != -1:
                    param.requires_grad = False
                if 'encoder' in model_name:
                    param.requires_grad = False
        self._mode = 'adapt'

    # ============================================================
    def _sample_weights(self, B: int, loss_sample_weights: Optional[Tensor], local: bool = False):
        """dpp.py:1031-1032 and the broadcasting of a (batch_size,) weight vector against the actual
        batch (equal sizes, or an actual batch of 1 which sees the SUM of the weights).  local: a forward-only
        call in data-parallel mode sees its own batch like a single process would (no shard semantics)."""
        if loss_sample_weights is None:      # the default vectors are constants: built once per batch size
            key = (B, None if (self._dp is None or local) else (self._dp['global_batch'], self._dp['offset']))
            hit = self._default_weights.get(key)
            if hit is None:
                hit = self._default_weights[key] = self._build_sample_weights(B, None, local)
            return hit
        return self._build_sample_weights(B, loss_sample_weights, local)

    def _build_sample_weights(self, B: int, loss_sample_weights: Optional[Tensor], local: bool = False):
        if self._dp is not None and not local:
            gb = self._dp['global_batch']
            w = torch.full((gb,), 1.0 / gb, device=self.device) if loss_sample_weights is None else loss_sample_weights
            local = w[self._dp['offset']:self._dp['offset'] + B].contiguous()
            return local, (w.contiguous() if self._dp['offset'] == 0 else None)
        if loss_sample_weights is None:
            loss_sample_weights = torch.ones(self.batch_size, device=self.device) / self.batch_size
        if loss_sample_weights.numel() == B:
            w = loss_sample_weights.to(self.device, torch.float32).contiguous()
        elif loss_sample_weights.numel() == 1:     # (B,) * (1,) broadcasts: every sample gets the single weight
            w = loss_sample_weights.to(self.device, torch.float32).reshape(1).repeat(B)
        elif B == 1:
            w = loss_sample_weights.to(self.device, torch.float32).sum().reshape(1)
        else:
            raise RuntimeError(f'The size of tensor a ({B}) must match the size of tensor b '
                               f'({loss_sample_weights.numel()}) at non-singleton dimension 0')
        return w, w

    # the tensors of the sample dict the path reads, in the order the step needs them
    UPLOAD_FIRST = [('rgb_aug', 0, 0), ('rgb_aug', -1, 0), ('rgb_aug', 1, 0)]
    UPLOAD_REST = [('camera_matrix', 0), ('inv_camera_matrix', 0), ('relative_distance', 0), ('relative_distance', 1),
                   ('rgb', -1, 0), ('rgb', 1, 0), ('rgb', 0, 0), ('rgb', 0, 1), ('rgb', 0, 2), ('rgb', 0, 3)]

    def _upload(self, inputs: Dict[Any, Tensor]):
        """Move the caller's dict to the device in place like dpp.py:916-917, asynchronously: the copies run on their own
        stream (pinned sources: DataLoader(pin_memory=True), slam.py:86), the three network inputs first, then the ten
        other entries the path reads, then everything else (upload_all_inputs=False: left on the host).  Returns the
        events (rgb_aug[0] there, rgb_aug[-1] there, rgb_aug[+1] there, everything the path reads there) for the engine's
        streams to wait on plus the list of entries the path does NOT read -- _upload_rest() moves those once the forward
        has been enqueued, so that their ~0.15 ms of host-side copy calls sit behind GPU work instead of in front of it --
        or None when nothing had to move."""
        dev = self.device
        todo = [k for k in self.UPLOAD_FIRST + self.UPLOAD_REST if k in inputs and inputs[k].device != dev]
        known = set(self.UPLOAD_FIRST + self.UPLOAD_REST)
        extra = [k for k in inputs if self.upload_all_inputs and k not in known
                 and isinstance(inputs[k], Tensor) and inputs[k].device != dev]
        if not todo and not extra:
            return None
        if dev.type != 'cuda':
            for k in todo + extra:
                inputs[k] = inputs[k].to(dev)
            return None
        if self._copy_stream is None:
            # HIGH priority, and not for its urgency: HIP multiplexes a process's streams onto a handful of hardware queues
            # (four per priority level), in-order each.  As the sixth normal-priority stream of the process the copy stream
            # landed on the POSE branch's hardware queue: the pose encoder's first kernel queued behind the barrier packets of
            # the whole upload, the upload of the unread entries queued behind the whole pose encoder, and the backward behind
            # that (profiles/r06_timeline_e2e_before.txt: 1.4 ms of the 4.25 ms end-to-end frame).  Queues are pooled per
            # priority, so a high-priority stream shares its queue with none of the engine's compute streams.
            self._copy_stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get('CLSLAM_PRIO_COPY', '-1')))
        cur = torch.cuda.current_stream(dev)
        cs = self._copy_stream
        # No cs.wait_stream(cur): the device blocks are allocated under the copy stream (its own pool in torch's caching
        # allocator; record_stream below defers their reuse until the engine's streams are past them), so the copies of
        # frame N+1 need not wait for frame N's backward + optimizer step still queued on the caller's stream -- they cross
        # PCIe underneath it.
        users = [cur] + [st for st in (self.engine.side_stream, self.engine.wg_stream, self.engine._caller) if st is not None]
        def copy(k, consumers=users):
            t = inputs[k].to(dev, non_blocking=True)
            for st in consumers:            # consumed on the engine's streams: keep the block until they are past it
                t.record_stream(st)
            inputs[k] = t
        with torch.cuda.stream(cs):
            evs = []
            for k in self.UPLOAD_FIRST:      # rgb_aug[0] (depth net, first pose pair), rgb_aug[-1] (first pair), rgb_aug[+1] (second pair)
                if k in todo:
                    copy(k)
                ev = torch.cuda.Event()
                ev.record(cs)
                evs.append(ev)
            for k in todo:
                if k not in self.UPLOAD_FIRST:
                    copy(k)
            all_ev = torch.cuda.Event()
            all_ev.record(cs)
        return evs[0], evs[1], evs[2], all_ev, extra

    def _upload_rest(self, inputs: Dict[Any, Tensor], extra) -> None:
        """The entries of the caller's dict the path never reads (dpp.py:916-917 moves them too): enqueued on the copy stream
        after the forward's launches; the caller's stream is ordered behind them, nothing of the step waits for them."""
        if not extra:
            return
        dev = self.device
        # the CALLER's stream -- during a detached training call torch's current stream is the engine's own, and a wait there
        # would put the rest of the step (view synthesis, loss, the whole backward) behind 22 copies it never reads
        cur = self.engine._caller if self.engine._caller is not None else torch.cuda.current_stream(dev)
        with torch.cuda.stream(self._copy_stream):
            for k in extra:
                t = inputs[k].to(dev, non_blocking=True)
                t.record_stream(cur)         # never read by the engine: only the caller's stream may touch it
                inputs[k] = t
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        cur.wait_event(ev)      # the dict is device-resident from the caller's stream's point of view when the call returns

    def _process_batch(self, inputs: Dict[Any, Tensor], loss_sample_weights: Optional[Tensor] = None,
                       use_online: bool = False, train: bool = False, graphed: bool = False, copy_inputs: bool = True,
                       reuse_frozen: bool = False, want_outputs: bool = True):
        ready = None
        if not reuse_frozen:             # steps 2..S of one adapt() call see the dict this loop already moved
            ready = self._upload(inputs)
        B = inputs['rgb_aug', 0, 0].shape[0]
        sample_w, smooth_w = self._sample_weights(B, loss_sample_weights, local=not train)
        if graphed:
            if ready is not None:
                torch.cuda.current_stream(self.device).wait_event(ready[3])
            outputs, losses = self.engine.train_step_graphed(inputs, sample_w=sample_w, smooth_w=smooth_w,
                                                             noise=self._injected_noise, copy_inputs=copy_inputs,
                                                             reuse_frozen=reuse_frozen, want_outputs=want_outputs)
        else:
            outputs, losses = self.engine.forward(inputs, train=train, sample_w=sample_w, smooth_w=smooth_w,
                                                  noise=self._injected_noise, reuse_frozen=reuse_frozen, inputs_ready=ready)
        if ready is not None:
            self._upload_rest(inputs, ready[4])
        ext = None
        if self._dp is not None and train:
            # only training steps are collective: predict() / adapt(online, None) on one rank (slam.py:178 on the
            # rank that holds the online frame) must not pair up with another rank's gradient exchange.
            # Three extra scalars ride on the exchange: this rank's sample count and its gradient-exchange layout
            # (code, code^2) -- _check_dp_tag() raises when the shard sizes do not add up to the global batch every rank
            # scaled its loss weights by (a stale enable_data_parallel after the replay buffer grew), or when the ranks would
            # issue different sequences of collectives (bucketed / whole / hipGraph), instead of corrupting or hanging silently.
            ext = torch.cat([losses, self._dp_tag(B)])
            self._dp_posted += 1
            self._dp['dist'].all_reduce(ext, group=self._dp['group'])
            losses = ext[:18]
        self._losses_dev = losses
        if self.device.type == 'cuda':
            # ONE device->host copy of the 18 loss scalars per step (the NaN check of dpp.py:1115-1118 needs the loss
            # on the host anyway): the returned dict holds HOST tensors, so the caller's per-key `.cpu()` / `.item()`
            # (slam.py:186-188: one per key) cost nothing instead of a stream synchronisation each
            if self._loss_host is None:
                self._loss_host = torch.empty(22, dtype=torch.float32, pin_memory=True)
                self._loss_event = torch.cuda.Event()
            if train and self.host_pose_output:
                if B not in self._pose_host:
                    self._pose_host[B] = torch.empty(2, B, 4, 4, dtype=torch.float32, pin_memory=True)
                self._pose_host[B].copy_(self.engine.workspace(B).T, non_blocking=True)   # (2, B, 4, 4): frames -1, +1
                self._pose_staged = B
            self._dp_tag_staged = ext is not None
            if ext is not None:
                self._loss_host.copy_(ext, non_blocking=True)
            else:
                self._loss_host[:18].copy_(losses, non_blocking=True)
            self._loss_event.record()
            if not train:
                self._loss_event.synchronize()
                loss_dict = self.engine.losses_dict(self._loss_host[:18].clone())
                self._raise_on_nan(loss_dict)
                if not self.host_pose_output:
                    loss_dict = self.engine.losses_dict(losses)
            else:
                loss_dict = None         # adapt() builds it after the optimizer launch (see there)
        else:
            self._dp_ext_cpu = ext      # checked behind the step's gradient exchange like on the GPU (_staged_losses)
            loss_dict = self.engine.losses_dict(losses)
            if not train:
                self._raise_on_nan(loss_dict)
        return outputs, loss_dict

    def _dp_tag(self, n_local: int) -> Tensor:
        """(samples of this rank, c, c^2, failed): c = the gradient-exchange layout this rank will use for the step; failed = 1
        from a rank that could not run the step (_dp_abort_step)"""
        eng = self.engine
        code = float(eng.grad_buckets * 2 + (1 if eng.graph_preferred(max(n_local, 1)) else 0))
        key = (n_local, code)
        hit = self._dp_tags.get(key)
        if hit is None:
            hit = self._dp_tags[key] = torch.tensor([float(n_local), code, code * code, 0.0], device=self.device)
        return hit

    def _check_dp_tag(self, tag) -> None:
        n, c, c2, failed = (float(v) for v in tag)
        world = self._dp['dist'].get_world_size(self._dp['group'])
        if failed > 0.5:
            raise DataParallelPeerFailure(f'data-parallel step: {int(round(failed))} peer rank(s) failed before the exchange (their own '
                                          'exception says why); the step was applied on no rank')
        if int(round(n)) != self._dp['global_batch']:
            raise RuntimeError(f'data-parallel step: the ranks hold {int(round(n))} samples but enable_data_parallel() was given a global '
                               f'batch of {self._dp["global_batch"]} (call it again whenever the minibatch size changes): the loss '
                               'weights of this step are wrong')
        if abs(world * c2 - c * c) > 1e-3:
            raise RuntimeError('data-parallel step: the ranks disagree on the gradient-exchange layout (CLSLAM_GRAD_BUCKETS / '
                               'CLSLAM_HIPGRAPH differ between processes)')

    def _staged_losses(self) -> Dict[str, Tensor]:
        """the training step's loss dict: waits for the forward's staged copy only, not for the stream"""
        if self.device.type != 'cuda':
            if self._dp_ext_cpu is not None:
                ext, self._dp_ext_cpu = self._dp_ext_cpu, None
                self._check_dp_tag(ext[18:22])
            return self.engine.losses_dict(self._losses_dev)
        self._loss_event.synchronize()
        if self._dp_tag_staged:
            self._check_dp_tag(self._loss_host[18:22])
        return self.engine.losses_dict(self._loss_host[:18].clone())

    def _raise_on_nan(self, loss_dict, undo_step: bool = False) -> None:
        """dpp.py:1115-1118"""
        value = float(loss_dict['loss'])
        if np.isnan(value):
            if undo_step:   # the guarded Adam launch did not touch weights or moments
                self.engine.adam_step_count -= 1
            for k, v in loss_dict.items():
                print(k, v.item())
            raise RuntimeError('NaN loss')

    def _backward(self, inputs: Dict[Any, Tensor]) -> None:
        B = inputs['rgb_aug', 0, 0].shape[0]
        # (adapt() calls optimizer.step() right after this: on a single GPU the reduction is fused into that launch)
        self.engine.backward(B, defer_reduce=self._dp is None, allreduce=self._allreduce if self._dp is not None else None)
        self._reduce_gradients()

    def _allreduce(self, t: Tensor) -> None:
        """sum over the ranks, on torch's current stream (the engine calls it per gradient bucket on its tail stream)"""
        self._dp_posted += 1
        self._dp['dist'].all_reduce(t, group=self._dp['group'])

    def _reduce_gradients(self) -> None:
        if self._dp is not None and not self.engine.grads_synced:
            with self.engine.training_stream():     # the tail stream when the engine left the reduction there
                self._dp_posted += 1
                self._dp['dist'].all_reduce(self.engine._g, group=self._dp['group'])
        self.engine.grads_synced = False


def _null_context():
    import contextlib
    return contextlib.nullcontext()


def _select_device() -> torch.device:
    from clslam_hip import _lib
    lib = _lib.get_lib()  # raises when libclslam_hip.so is missing: no CPU fallback
    if lib.is_device:
        if not torch.cuda.is_available():
            raise RuntimeError('libclslam_hip.so needs an MI355X (no GPU visible); there is no CPU fallback')
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')  # only reachable with the test emulator installed by tests/emu_util.py

"""CoVIO's asynchronous predict/adapt mode (reference README.md:62, 171-172) as a two-role
process group, one process per GPU:

    rank 0 of the group          INFERENCE replica: answers predict() / predict_pose() / adapt(online, None) at camera
                                 rate with the weights of the last completed synchronisation -- it never waits for a
                                 training step;
    ranks 1 .. R                 TRAINING replicas: run adapt(online, training) on every frame, data-parallel among
                                 themselves (DepthPosePrediction.enable_data_parallel on their own sub-group, one
                                 gradient all-reduce per step over RCCL / xGMI);
    every `sync_every` frames    the trainers' flat weight arena (4.47 M floats, 17.9 MB -- the engine keeps every trainable
                                 tensor in ONE buffer, so this is a single broadcast) goes from the first trainer to the
                                 inference replica.  The broadcast is posted asynchronously on both sides: the trainer
                                 snapshots the arena into a staging buffer and goes on training, the inference replica
                                 keeps predicting with its current weights and installs the new ones at the first frame
                                 boundary after the transfer has completed.

Only the decoders are trainable during adaptation (dpp.py:308, 813-819), so the frozen encoder weights never move.
The replica's lag is bounded: the weights in use at frame f stem from a trainer state no older than
2 * sync_every frames (one period until the next snapshot + one for a transfer still in flight).

The reference itself has no multi-process mode (the asynchronous variant lives in OpenDR); this module is the
MI355X-native realisation of that idea on the building blocks of the data-parallel path."""
from typing import Any, Dict, Optional, Tuple

import torch


ABORT_FRAME = -2.0        # frame marker of a snapshot posted by a trainer that failed: releases the inference replica


class AsyncAdaptation:
    def __init__(self, predictor, group=None, sync_every: int = 5, trainer_group=None, trainer_global_batch: Optional[int] = None,
                 trainer_shard_offset: int = 0, transfer_timeout_s: float = 120.0) -> None:
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        if sync_every < 1:
            raise ValueError('sync_every must be positive')
        self.dist, self.group, self.p = dist, group, predictor
        self.rank_in_group = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if self.world < 2:
            raise ValueError('the asynchronous mode needs an inference replica and at least one training replica')
        self.role = 'inference' if self.rank_in_group == 0 else 'trainer'
        self.sync_every = int(sync_every)
        self.transfer_timeout_s = float(transfer_timeout_s)
        self.leader = dist.get_global_rank(group, 1) if group is not None else 1        # first trainer: broadcast source
        eng = predictor.engine
        self._n = eng.layout.size
        # arena + [adam step count, frame index of the snapshot]: the version travels with the weights
        self._staging = torch.zeros(self._n + 2, device=eng.device)
        self._work = None
        self._posted_frame = -1
        self.weights_frame = -1          # frame index after which the weights in use were snapshotted (-1: initial weights)
        self.weights_step = 0
        self.installs = 0
        self.used_weights_frame = -1
        self.keep_used_weights = False   # tests: keep a copy of the arena each prediction was made with
        self.used_weights = None
        if self.role == 'trainer' and self.world > 2:
            if trainer_group is None or trainer_global_batch is None:
                raise ValueError('several training replicas need their own sub-group and the global minibatch size')
            predictor.enable_data_parallel(trainer_global_batch, trainer_shard_offset, process_group=trainer_group)

    # ------------------------------------------------------------------------------------------------------------
    def _post(self, frame: int, abort: bool = False) -> None:
        eng = self.p.engine
        if self.role == 'trainer':
            if self._work is not None:
                self._work.wait()                     # the previous snapshot has left the staging buffer
            eng.wait_training()                       # the optimizer step in flight belongs to this snapshot
            self._staging[:self._n].copy_(eng.w)
            self._staging[self._n] = float(eng.adam_step_count)
            self._staging[self._n + 1] = ABORT_FRAME if abort else float(frame)
        elif self._work is not None:
            self._install(block=True)                 # one transfer in flight at a time
        self._work = self.dist.broadcast(self._staging, src=self.leader, group=self.group, async_op=True)
        self._posted_frame = frame

    def _bounded_wait(self) -> None:
        """host-side wait for the transfer in flight with a deadline: a peer that died without posting its side must not park
        this replica forever"""
        import time
        deadline = time.monotonic() + self.transfer_timeout_s
        while not self._work.is_completed():
            if time.monotonic() > deadline:
                raise RuntimeError(f'weight broadcast of frame {self._posted_frame} did not complete within '
                                   f'{self.transfer_timeout_s:.0f} s: peer replica lost?')
            time.sleep(0.0005)

    def _install(self, block: bool) -> bool:
        """inference replica: swap in the received arena if the transfer has completed"""
        if self._work is None:
            return False
        if not block and not self._work.is_completed():
            return False
        if block:
            self._bounded_wait()
        self._work.wait()
        self._work = None
        meta = self._staging[self._n:].cpu()
        if float(meta[1]) == ABORT_FRAME:
            raise RuntimeError('the training replica failed and released the inference replica (see its exception)')
        self.weights_step, self.weights_frame = int(meta[0]), int(meta[1])
        self.p.engine.install_weights(self._staging[:self._n], self.weights_step)
        self.installs += 1
        return True

    # ------------------------------------------------------------------------------------------------------------
    def step(self, frame: int, online_data: Dict[Any, torch.Tensor], training_data: Optional[Dict[Any, torch.Tensor]] = None,
             steps: int = 1) -> Tuple[Dict[Any, torch.Tensor], Optional[Dict[str, torch.Tensor]]]:
        """One camera frame.  Trainers: adapt(online, training, steps) (slam.py:174-176).  Inference replica:
        adapt(online, None) = forward only (slam.py:178), with whatever weights are installed."""
        if self.role == 'trainer':
            # The snapshot broadcast (communicator `group`) and the gradient all-reduce (communicator `trainer_group`) are two
            # RCCL communicators on one device with no ordering between them: both holding channels and CUs while one of them
            # spins on a slow peer is a known way to deadlock.  They are serialised here -- the broadcast in flight is waited
            # for (stream-ordered, the host does not block) before this frame's training step and its all-reduce go out.
            if self._work is not None:
                self._work.wait()
                self._work = None
            try:
                out = self.p.adapt(online_data, training_data if training_data is not None else online_data, steps=steps)
            except Exception as exc:
                # Release the inference replica with an abort marker at its next receive -- but only when EVERY trainer is on
                # this path: the marker travels by a collective over `group`, which all trainers have to post.  That is the
                # case with a single trainer; for RuntimeError('NaN loss') (the loss scalars are all-reduced before the check of
                # dpp.py:1115-1118, so all trainers raise at the same step); and for a failure of ONE trainer before its step's
                # exchange (malformed input, a failed allocation): DepthPosePrediction.adapt() completes that step's collectives
                # with a status word, the failing trainer's exception carries `dp_agreed` and its peers raise
                # DataParallelPeerFailure at the same step.  Only a failure in the MIDDLE of a step's collectives is not agreed
                # on -- that trainer just raises, the peers run into their communicator's timeout / `transfer_timeout_s`.
                from depth_pose_prediction.depth_pose_prediction import DataParallelPeerFailure
                agreed = (self.world == 2 or (isinstance(exc, RuntimeError) and 'NaN loss' in str(exc))
                          or isinstance(exc, DataParallelPeerFailure) or bool(getattr(exc, 'dp_agreed', False)))
                if agreed:
                    try:
                        self._post(frame, abort=True)
                        self.flush()
                    except Exception:       # noqa: BLE001 -- the step's own failure is the one to report
                        pass
                raise
        else:
            self._install(block=False)
            self.used_weights_frame = self.weights_frame      # version this frame's prediction is made with
            if self.keep_used_weights:
                self.p.engine.wait_training()
                self.used_weights = self.p.engine.w.clone()
            out = self.p.adapt(online_data, None)
        if (frame + 1) % self.sync_every == 0:
            self._post(frame)
        return out

    def flush(self) -> None:
        """complete the transfer in flight (end of a sequence, before save_model() on the inference replica)"""
        if self._work is not None:
            if self.role == 'inference':
                self._install(block=True)
            else:
                self._bounded_wait()
                self._work.wait()
                self._work = None

    @property
    def lag_bound_frames(self) -> int:
        return 2 * self.sync_every

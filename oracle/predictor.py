"""OraclePredictor: torch-CPU restatement of ``DepthPosePrediction._process_batch`` /
``adapt`` / ``predict`` / ``predict_pose`` (reference: depth_pose_prediction/
depth_pose_prediction.py, "dpp.py" below).  Test infrastructure only (see oracle/__init__).
"""
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import functional as OF
from .networks import DepthDecoder, PoseDecoder, ResnetEncoder


class OraclePredictor:
    """Mirrors the reference predictor for the adaptation configuration
    (config/config_adapt.yaml): frame_ids (0,-1,1), scales (0,1,2,3), mask_dynamic=False."""

    def __init__(self, height: int, width: int, batch_size: int, min_depth: Optional[float] = 0.1,
                 max_depth: Optional[float] = None, disparity_smoothness: float = 1e-3,
                 velocity_loss_scaling: Optional[float] = 0.05, learning_rate: float = 1e-4,
                 reference_quirks: bool = True) -> None:
        self.height, self.width, self.batch_size = height, width, batch_size
        self.min_depth, self.max_depth = min_depth, max_depth
        self.disparity_smoothness = disparity_smoothness
        self.velocity_loss_scaling = velocity_loss_scaling
        self.reference_quirks = reference_quirks
        self.scales = (0, 1, 2, 3)
        self.forced_sel, self.last_combined, self.last_sel = None, {}, {}
        # test hooks for the view synthesis (functional.grid_sample_border): forced_cells[s][f] = (x0, y0, mx, my) imposes
        # the bilinear cell / clip decisions of another implementation; record_cells=True keeps this run's own in last_cells
        self.forced_cells, self.record_cells, self.last_cells = None, False, {}
        # forced_forward = {('disp', s): tensor, ('cam_T_cam', 0, f): tensor}: evaluate loss and backward AT another
        # implementation's forward point (value replaced, gradient path kept: x + (x_other - x).detach()) -- separates the
        # rounding of the forward pass, which the ill-conditioned loss amplifies, from the arithmetic of the backward pass
        self.forced_forward = None
        # forced_l1_sign[s][f] = sign(target - warped) of another implementation (functional.reprojection_loss)
        self.forced_l1_sign = None
        self.frame_ids = (0, -1, 1)
        # dpp.py:129-137 (dict insertion order defines the optimizer's parameter order)
        self.models = {
            'depth_encoder': ResnetEncoder(1),
            'depth_decoder': DepthDecoder(self.scales),
            'pose_encoder': ResnetEncoder(2),
            'pose_decoder': PoseDecoder(),
        }
        params = []
        for m in self.models.values():
            params += list(m.parameters())
        self.optimizer = torch.optim.Adam(params, learning_rate)  # dpp.py:203

    # ------------------------------------------------------------------ modes (dpp.py:797-827)
    def set_eval(self) -> None:
        for m in self.models.values():
            m.eval()

    def set_adapt(self) -> None:
        for model_name, m in self.models.items():
            m.eval()
            for name, p in m.named_parameters():
                if name.find('bn') != -1:
                    p.requires_grad = False
                if 'encoder' in model_name:
                    p.requires_grad = False

    # ------------------------------------------------------------------ forward (dpp.py:906-1017)
    def process_batch(self, inputs: Dict[Any, Tensor], noise: Optional[Dict[int, Tensor]] = None,
                      sample_weights: Optional[Tensor] = None
                      ) -> Tuple[Dict[Any, Tensor], Dict[str, Tensor]]:
        H, W = self.height, self.width
        outputs: Dict[Any, Tensor] = {}
        feats = self.models['depth_encoder'](inputs[('rgb_aug', 0, 0)])  # dpp.py:931-936
        outputs.update(self.models['depth_decoder'](feats))
        if self.forced_forward is not None:
            for s in self.scales:
                d = outputs[('disp', s)]
                outputs[('disp', s)] = d + (self.forced_forward['disp', s].to(d.dtype) - d).detach()
        T = {}
        for f in (-1, 1):  # dpp.py:938-974
            pair = ([inputs['rgb_aug', f, 0], inputs['rgb_aug', 0, 0]] if f < 0 else
                    [inputs['rgb_aug', 0, 0], inputs['rgb_aug', f, 0]])
            pf = self.models['pose_encoder'](torch.cat(pair, 1))
            aa, tr = self.models['pose_decoder'](pf[-1])
            aa, tr = aa[:, 0], tr[:, 0]
            outputs[('axis_angle', 0, f)] = aa
            outputs[('translation', 0, f)] = tr
            T[f] = OF.transformation_from_parameters(aa, tr, invert=f < 0)
            if self.forced_forward is not None:
                T[f] = T[f] + (self.forced_forward['cam_T_cam', 0, f].to(T[f].dtype) - T[f]).detach()
            outputs[('cam_T_cam', 0, f)] = T[f]
        src = {f: inputs[('rgb', f, 0)] for f in (-1, 1)}
        for s in self.scales:  # dpp.py:976-1017
            rec = {} if self.record_cells else None
            depth, warped = OF.reconstruct(outputs[('disp', s)], T, inputs[('camera_matrix', 0)],
                                           inputs[('inv_camera_matrix', 0)], src, H, W,
                                           self.min_depth, self.max_depth,
                                           cells=None if self.forced_cells is None else self.forced_cells[s], record=rec,
                                           P_value=None if self.forced_forward is None else
                                           {f: self.forced_forward.get(('P', f)) for f in (-1, 1)})
            if rec is not None:
                self.last_cells[s] = {f: v[0] for f, v in rec.items()}
            outputs[('depth', s)] = depth
            for f in (-1, 1):
                outputs[('rgb', f, s)] = warped[f]
        losses = self.compute_loss(inputs, outputs, noise, sample_weights)
        return outputs, losses

    # ------------------------------------------------------------------ loss (dpp.py:1019-1120)
    def compute_loss(self, inputs, outputs, noise, sample_weights) -> Dict[str, Tensor]:
        if sample_weights is None:  # dpp.py:1031-1032 (configured batch size!)
            sample_weights = torch.ones(self.batch_size) / self.batch_size
        losses = {}
        total = torch.zeros(1)
        target = inputs['rgb', 0, 0]
        for s in self.scales:
            rp = torch.cat([OF.reprojection_loss(outputs['rgb', f, s], target,
                                                 None if self.forced_l1_sign is None else self.forced_l1_sign[s][f])
                            for f in (-1, 1)], 1)
            idl = torch.cat([OF.reprojection_loss(inputs['rgb', f, 0], target)
                             for f in (-1, 1)], 1)
            if noise is not None:  # dpp.py:1055-1056, injected instead of drawn
                idl = idl + noise[s]
            combined = torch.cat((idl, rp), 1)  # dpp.py:1057
            to_opt, argmin = torch.min(combined, dim=1)
            # test hooks (tests/test_backward_parity.py): what was selected, and optionally a selection imposed from
            # outside (the kernel path's) so that the remaining gradient difference can be attributed
            self.last_combined[s], self.last_sel[s] = combined.detach(), argmin.detach()
            if self.forced_sel is not None:
                to_opt = torch.gather(combined, 1, self.forced_sel[s].long().unsqueeze(1)).squeeze(1)
            rl = (to_opt.mean(2).mean(1) * sample_weights).sum()  # dpp.py:1073
            losses[f'reprojection_loss/scale_{s}'] = rl
            disp = outputs['disp', s]
            norm_disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)  # dpp.py:1087-1088
            if self.reference_quirks:
                sm = OF.smooth_loss_reference(norm_disp, inputs['rgb', 0, s])
            else:
                sm = OF.smooth_loss_intended(norm_disp, inputs['rgb', 0, s])
            sm = (sm * sample_weights).sum()
            losses[f'smooth_loss/scale_{s}'] = sm
            reg = self.disparity_smoothness / (2**s) * sm  # dpp.py:1094
            losses[f'reg_loss/scale_{s}'] = reg
            loss = rl + reg
            losses[f'depth_loss/scale_{s}'] = loss
            total = total + loss
        total = total / len(self.scales)
        if self.velocity_loss_scaling is not None and self.velocity_loss_scaling > 0:
            v = self.velocity_loss_scaling * OF.velocity_loss(
                outputs['translation', 0, -1], outputs['translation', 0, 1],
                inputs['relative_distance', 0], inputs['relative_distance', 1])
            v = (v * sample_weights).sum()
            losses['velocity_loss'] = v
            total = total + v
        # dpp.py:1102/1110: 'depth_loss' aliases the tensor that velocity_loss is added to
        # in place, so it reports the TOTAL.
        losses['depth_loss'] = total
        losses['loss'] = total
        if np.isnan(float(total.detach())):
            raise RuntimeError('NaN loss')
        return losses

    # ------------------------------------------------------------------ API (dpp.py:291-319, 530-536)
    def adapt(self, training_data: Dict[Any, Tensor], steps: int = 1, noise_per_step=None,
              sample_weights: Optional[Tensor] = None):
        self.set_adapt()
        out = None
        for it in range(steps):
            noise = None if noise_per_step is None else noise_per_step[it]
            out, losses = self.process_batch(training_data, noise, sample_weights)
            self.optimizer.zero_grad()
            losses['loss'].backward()
            self.optimizer.step()
        return out, losses

    def to_double(self) -> 'OraclePredictor':
        """The same predictor in float64 (parameters converted in place: the optimizer keeps its references).  Feed it
        float64 inputs and noise.  This is the yardstick of tests/test_trajectory.py: how far the fp32 restatement itself drifts
        from exact arithmetic over the reference's shipped `adaptation_epochs: 5` (config/config_adapt.yaml:53, dpp.py:309-313)
        is the envelope the HIP path is held to."""
        for m in self.models.values():
            m.double()
        return self

    def trajectory(self, training_data: Dict[Any, Tensor], noise_per_step, steps: int):
        """adapt(steps=steps) (dpp.py:309-319) one optimizer step at a time, recording per step what the later-step parity
        test compares: the step's forward (pre-update weights: disparity at scale 0, both pose matrices, the loss) and the
        trainable tensors right after the step's update."""
        rec = []
        for it in range(steps):
            out, losses = self.adapt(training_data, steps=1, noise_per_step=[noise_per_step[it]])
            w = {f'{model}/{k}': v.detach().clone() for model in ('depth_decoder', 'pose_decoder')
                 for k, v in self.models[model].state_dict().items()}
            rec.append({'disp0': out['disp', 0].detach().clone(), 'T-1': out['cam_T_cam', 0, -1].detach().clone(),
                        'T+1': out['cam_T_cam', 0, 1].detach().clone(), 'loss': float(losses['loss'].detach()), 'w': w})
        return rec

    def predict(self, batch, noise=None):
        self.set_eval()
        with torch.no_grad():
            return self.process_batch(batch, noise)

    def predict_pose(self, image_0: Tensor, image_1: Tensor) -> Tensor:
        """dpp.py:628-664 (invert=False)."""
        if image_0.dim() == 3:
            image_0 = image_0.unsqueeze(0)
        if image_1.dim() == 3:
            image_1 = image_1.unsqueeze(0)
        self.set_eval()
        with torch.no_grad():
            pf = self.models['pose_encoder'](torch.cat([image_0, image_1], 1))
            aa, tr = self.models['pose_decoder'](pf[-1])
            return OF.transformation_from_parameters(aa[:, 0], tr[:, 0], invert=False)

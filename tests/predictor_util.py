"""Builds the product DepthPosePrediction with the synthetic closed-form weights."""
from pathlib import Path
from types import SimpleNamespace

import torch

from clslam_hip import synth


def make_config(B, log_path='/tmp/clslam_test_log', **over):
    from depth_pose_prediction import Config
    kw = dict(config_file=Path('x.yaml'), train_set='all', val_set=0, resnet_depth=18, resnet_pose=18,
              resnet_pretrained=False, scales=(0, 1, 2, 3), learning_rate=1e-4, scheduler_step_size=15, batch_size=B,
              num_workers=0, num_epochs=1, min_depth=0.1, max_depth=None, disparity_smoothness=1e-3,
              velocity_loss_scaling=0.05, mask_dynamic=False, log_path=Path(log_path), save_frequency=-1,
              save_val_depth=False, save_val_depth_batches=0, multiple_gpus=False, gpu_ids=None,
              load_weights_folder=None, use_wandb=False)
    kw.update(over)
    return Config(**kw)


def make_predictor(H, W, B, seed=0, reference_quirks=True, host_pose_output=None, upload_all_inputs=None, **over):
    from depth_pose_prediction import DepthPosePrediction
    ds = SimpleNamespace(dataset='Kitti', config_file=Path('x.yaml'), dataset_path=None, scales=(0, 1, 2, 3), height=H,
                         width=W, frame_ids=(0, -1, 1))
    p = DepthPosePrediction(ds, make_config(B, **over), reference_quirks=reference_quirks, host_pose_output=host_pose_output,
                            upload_all_inputs=upload_all_inputs)
    for name, m in p.models.items():
        sd = torch.nn.Module.state_dict(m)
        m.load_state_dict(synth.fill_state_dict(sd, seed, name))
    p.is_trained = True
    return p

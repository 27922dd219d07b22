#!/usr/bin/env python
"""TEST INFRASTRUCTURE (build container only: needs /root/reference).  Runs the REFERENCE's own callers --
config/config_parser.py, datasets/kitti.py, slam/slam.py (`Slam.__init__`, `Slam.step`, `Slam.save_model`),
slam/replay_buffer.py -- unchanged on top of the cl-slam_amd packages, with sys.path exactly as INTEGRATION.md
documents it (cl-slam_amd/ ahead of the reference) and stubs only for the third-party packages the container
lacks (tests/ref_stubs.py).  The HIP kernels run on the CPU emulator here.

    python tests/ref_callers_driver.py <workdir> [frames] [gpu-ingest]

`gpu-ingest`: the replay minibatch built by clslam_hip.ingest.ReplaySampleBuilder, switched on with its install() on the
reference's ReplayBuffer / Slam OBJECTS (two instance attributes) -- the classes and every other line untouched.

Prints one JSON line with what happened; tests/test_reference_callers.py asserts on it."""
import json
import os
import sys
from pathlib import Path

TESTS = Path(__file__).resolve().parent
ROOT = TESTS.parent
REF = Path('/root/reference')
H, W = 64, 64


def make_kitti_tree(root: Path, n: int, period: int) -> None:
    """a KITTI-odometry directory tree (datasets/kitti.py:92-200) with smooth synthetic frames; the camera
    comes back to where it was `period` frames ago (loop closures)"""
    import numpy as np
    from PIL import Image
    seq = root / 'sequences' / '06'
    (seq / 'image_2').mkdir(parents=True)
    (seq / 'oxts' / 'data').mkdir(parents=True)
    (root / 'poses').mkdir()
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:72, 0:400].astype(np.float32)
    pano = np.zeros((72, 400, 3), np.float32)
    for _ in range(8):
        fx, fy, ph = rng.uniform(0.02, 0.25), rng.uniform(0.02, 0.3), rng.uniform(0, 6.28, 3)
        for c in range(3):
            pano[..., c] += np.sin(fx * xx + fy * yy + ph[c])
    pano = (pano - pano.min()) / (pano.max() - pano.min())
    stamps, poses = [], []
    for i in range(n):
        x0 = 12 * (i % period)
        frame = pano[:, x0:x0 + 144]
        Image.fromarray((frame * 255).astype(np.uint8)).save(seq / 'image_2' / f'{i:06}.png')
        oxts = np.zeros(30)
        oxts[8] = 5.0                                             # forward speed [m/s] (kitti.py:327-329)
        np.savetxt(seq / 'oxts' / 'data' / f'{i:010}.txt', oxts[None])
        stamps.append(f'2011-09-30 12:00:{i * 0.1:012.9f}')
        pose = np.eye(4)[:3]
        pose[2, 3] = 0.5 * i
        poses.append(pose.reshape(-1))
    (seq / 'oxts' / 'timestamps.txt').write_text('\n'.join(stamps) + '\n')
    np.savetxt(root / 'poses' / '06.txt', np.array(poses))


def main() -> None:
    work = Path(sys.argv[1])
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    gpu_ingest = len(sys.argv) > 3 and sys.argv[3] == 'gpu-ingest'
    # --- the documented drop-in order: cl-slam_amd first, then the reference checkout ---------------------
    sys.path[:0] = [str(ROOT / 'cl-slam_amd'), str(REF)]
    sys.path += [str(TESTS), str(ROOT)]
    os.chdir(work)
    import ref_stubs
    ref_stubs.install()
    import torch
    import yaml
    from emu_util import use_backend
    use_backend('emu')

    # every one of these is the reference's module except the three cl-slam_amd packages
    import config.config_parser as config_parser
    import slam as slam_pkg
    import slam.replay_buffer
    import slam.slam as slam_mod
    import depth_pose_prediction
    import loop_closure_detection
    import faiss
    origin = {m.__name__: str(Path(m.__file__).resolve()) for m in
              (config_parser, slam_mod, slam.replay_buffer, depth_pose_prediction, loop_closure_detection, faiss)}
    for name in ('config.config_parser', 'slam.slam', 'slam.replay_buffer'):
        assert origin[name].startswith(str(REF)), origin
    for name in ('depth_pose_prediction', 'loop_closure_detection', 'faiss'):
        assert origin[name].startswith(str(ROOT / 'cl-slam_amd')), origin

    # --- the reference's own YAML, parsed by the reference's own parser --------------------------------------
    shipped = config_parser.ConfigParser(REF / 'config' / 'config_adapt.yaml')
    assert type(shipped.loop_closure) is loop_closure_detection.Config
    assert type(shipped.depth_pose) is depth_pose_prediction.Config
    assert shipped.loop_closure.id_threshold == 250 and shipped.depth_pose.batch_size == 3

    # --- a tiny synthetic run: same file, sizes/paths replaced -----------------------------------------------
    make_kitti_tree(work / 'kitti', frames + 3, period=3)
    with open(REF / 'config' / 'config_adapt.yaml', encoding='utf-8') as f:
        cfg = yaml.safe_load(f)
    cfg['Dataset'].update(dataset_path=str(work / 'kitti'), height=H, width=W)
    cfg['DepthPosePrediction'].update(log_path=str(work / 'log'), load_weights_folder=str(work / 'ckpt' / 'models' / 'weights_000'),
                                      batch_size=3, resnet_pretrained=False)
    cfg['ReplayBuffer'].update(load_path=str(work / 'log' / 'replay_buffer'), max_buffer_size=2, similarity_threshold=0.9999)
    cfg['LoopClosureDetection'].update(detection_threshold=0.9, id_threshold=2, num_matches=1)
    cfg['Slam'].update(adaptation_epochs=1, logging=False, keyframe_frequency=1, lc_distance_poses=0, start_frame=0)
    with open(work / 'config_small.yaml', 'w', encoding='utf-8') as f:
        yaml.safe_dump(cfg, f)
    config = config_parser.ConfigParser(work / 'config_small.yaml')

    # a checkpoint in the reference's layout (written by the product's save_model) and MobileNetV3 weights on disk
    from predictor_util import make_predictor
    seed_pred = make_predictor(H, W, 3, log_path=str(work / 'ckpt'))
    seed_pred.save_model()
    del seed_pred
    from test_lcd_encoder import _weights
    torch.save(_weights()[1], work / 'mbv3.pth')
    os.environ['CLSLAM_MOBILENETV3_WEIGHTS'] = str(work / 'mbv3.pth')

    torch.manual_seed(42)
    s = slam_pkg.Slam(config)                                         # main_adapt.py:23
    assert type(s.predictor) is depth_pose_prediction.DepthPosePrediction
    assert type(s.loop_closure_detection) is loop_closure_detection.LoopClosureDetection
    assert type(s.replay_buffer) is slam.replay_buffer.ReplayBuffer
    assert type(s.replay_buffer.feature_encoder) is loop_closure_detection.FeatureEncoder
    report = {'origin': origin, 'steps': [], 'gpu_ingest': gpu_ingest}
    if gpu_ingest:
        from clslam_hip.ingest import ReplaySampleBuilder
        rb = s.replay_buffer
        builder = ReplaySampleBuilder(H, W, rb.scales, rb.frames, device=s.predictor.device, do_augmentation=rb.do_augmentation,
                                      decode_threads=4)
        builder.install(rb, s)
        calls = {'get_many': 0, 'files': 0}
        inner = builder.get_many

        def counted(filenames, *a, **k):
            calls['get_many'] += 1
            calls['files'] += len(filenames)
            return inner(filenames, *a, **k)
        builder.get_many = counted
        report['ingest_calls'] = calls
    import random
    random.seed(7)                                                    # the colour-jitter draws of the replay samples
    while s.current_step < frames:                                    # main_adapt.py:25-29
        losses = s.step()
        report['steps'].append({'step': s.current_step, 'loss': float(losses['depth_loss']),
                                'velocity_loss': float(losses['velocity_loss']),
                                'buffer': sorted(int(i) for i in faiss.vector_to_array(s.replay_buffer.faiss_index.id_map)),
                                'lcd_frames': int(s.loop_closure_detection.faiss_index.ntotal),
                                'vertices': len(s.pose_graph.vertex_ids),
                                'loop_closures': int(s.pose_graph.num_loop_closures),
                                'optimize_calls': int(s.pose_graph.optimize_calls)})
    s.save_model()                                                    # main_adapt.py:31-32
    report['adam_steps'] = int(s.predictor.engine.adam_step_count)
    report['saved'] = sorted(p.name for p in (work / 'log' / 'models' / 'weights_000').iterdir())
    report['buffer_state'] = (work / 'log' / 'replay_buffer' / 'buffer_state.pkl').exists()
    report['replay_files'] = sorted(p.name for p in (work / 'log' / 'replay_buffer').glob('kitti_*.pkl'))
    # the saved replay-buffer state loads again through the reference's load_state (pickled index)
    rb2 = slam.replay_buffer.ReplayBuffer(work / 'log' / 'replay_buffer', 'Kitti', work / 'log' / 'replay_buffer' / 'buffer_state.pkl',
                                          H, W, [0, 1, 2, 3], [0, -1, 1], batch_size=2, maximize_diversity=True, max_buffer_size=2,
                                          similarity_threshold=0.9999)
    report['reloaded_ids'] = sorted(int(i) for i in faiss.vector_to_array(rb2.faiss_index.id_map))
    print('REPORT ' + json.dumps(report))


if __name__ == '__main__':
    main()

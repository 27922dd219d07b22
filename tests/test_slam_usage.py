"""The predictor used exactly the way the reference's SLAM driver uses it (slam/slam.py:39-40,
143-147,174-192,229,266-270,296), plus the API corners of adapt()/predict()."""
import numpy as np
import pytest
import torch

from clslam_hip import synth
from emu_util import BACKENDS, use_backend
from helpers import make_oracle, rel_err
from predictor_util import make_predictor

H, W = 64, 64


def _cat_dict(d1, d2):  # slam/slam.py:300-309
    return {k: torch.cat([d1[k], d2[k]]) for k in d1 if k in d2}


@pytest.mark.parametrize('backend', BACKENDS)
def test_slam_step_sequence(backend, tmp_path):
    use_backend(backend)
    B = 3
    p = make_predictor(H, W, B, log_path=str(tmp_path))
    online = synth.make_batch(1, H, W, seed=21)
    online['relative_pose', 1] = torch.eye(4).unsqueeze(0)       # extra keys the dataset adds; dropped by _cat_dict
    replay = synth.make_batch(2, H, W, seed=22)
    # slam.py:143-147
    p._set_eval()
    with torch.no_grad():
        img = online['rgb', 0, 0].to(p.device)
        feat = p.models['depth_encoder'](img)[4].detach()
        feat = feat.mean(-1).mean(-1).cpu().numpy()
    assert feat.shape == (1, 512)
    training = _cat_dict(online, replay)
    assert 'relative_pose' not in [k[0] for k in training if isinstance(k, tuple)]
    # slam.py:174-192, adaptation_epochs = 5 in the shipped config -> use 2 here
    outputs, losses = p.adapt(online, training, steps=2)
    T = outputs['cam_T_cam', 0, 1][0, :]
    T = torch.linalg.inv(T).squeeze().cpu().detach().numpy()
    assert T.shape == (4, 4) and np.isfinite(T).all()
    losses = {k: float(v.squeeze().cpu().detach().numpy()) for k, v in losses.items()}
    assert set(losses) >= {'depth_loss', 'velocity_loss', 'loss'} and losses['depth_loss'] == losses['loss']
    # the caller's dict was moved to the device in place (dpp.py:916-917)
    assert training['rgb', 0, 0].device.type == p.device.type
    # slam.py:266-270
    depth = outputs['depth', 0][0].cpu()
    assert depth.shape == (1, H, W) and float(depth.min()) >= p.min_depth * 0.999
    # slam.py:178 (adaptation disabled): forward only, batch of 1 while batch_size == 3
    o2, l2 = p.adapt(online, None)
    assert o2['depth', 0].shape == (1, 1, H, W) and torch.isfinite(l2['loss']).all()
    # slam.py:229
    Tlc, cov = p.predict_pose(online['rgb', 1, 0][0], replay['rgb', 1, 0][0], as_numpy=True)
    assert Tlc.shape == (4, 4) and cov.shape == (6, 6)
    # slam.py:296
    p.save_model()
    assert (tmp_path / 'models' / 'weights_000' / 'pose_decoder.pth').exists()


@pytest.mark.parametrize('backend', BACKENDS)
def test_api_corners_match_oracle(backend):
    """max_depth set, batch of 1 under a configured batch_size of 2 (weights broadcast-sum), predict()
    after adapt(), online_loss_weight."""
    use_backend(backend)
    B = 2
    p = make_predictor(H, W, B, max_depth=80.0)
    o = make_oracle(H, W, B, max_depth=80.0)
    one = synth.make_batch(1, H, W, seed=31)
    noise1 = synth.make_noise(1, H, W, seed=3)
    p.set_tie_break_noise(noise1)
    out, losses = p.adapt({k: v.clone() for k, v in one.items()}, None)
    oo, ol = o.predict(one, noise1)
    assert rel_err(out['depth', 0].cpu(), oo['depth', 0]) < 1e-4
    assert float(out['depth', 0].max()) <= 80.0 * 1.001
    for k in ('loss', 'velocity_loss', 'reprojection_loss/scale_2', 'smooth_loss/scale_1'):
        assert abs(float(losses[k]) - float(ol[k])) <= 1e-4 * max(abs(float(ol[k])), 1e-3), k
    # online_loss_weight (dpp.py:297-305): non-uniform sample weights
    two = synth.make_batch(B, H, W, seed=32)
    noise2 = synth.make_noise(B, H, W, seed=4)
    p.set_tie_break_noise(noise2)
    out, losses = p.adapt(None, {k: v.clone() for k, v in two.items()}, steps=1, online_loss_weight=0.7)
    o.set_adapt()
    oo, ol = o.process_batch(two, noise2, torch.tensor([0.7, 0.3]))
    for k, v in ol.items():
        assert abs(float(losses[k]) - float(v)) <= 1e-4 * max(abs(float(v)), 1e-3), k
    # predict() afterwards runs on the adapted weights and returns the reference's key set
    pred = p.predict({k: v.clone() for k, v in two.items()})
    assert set(pred.keys()) == set(oo.keys())
    with pytest.raises(RuntimeError):   # actual batch neither 1 nor batch_size
        p.adapt(None, synth.make_batch(3, H, W, seed=1), steps=1)


@pytest.mark.parametrize('backend', BACKENDS)
def test_mini_slam_loop(backend, tmp_path, monkeypatch):
    """GPU twin of tests/test_reference_callers.py (the reference checkout does not travel to the GPU box): the
    per-frame sequence of slam/slam.py:137-243 on the product packages alone -- encoder descriptor -> replay buffer
    bookkeeping through the `faiss`-named module (the calls of replay_buffer.py:95-160) and through DiversityBuffer,
    adapt() on online + replayed samples, LoopClosureDetection.add/search, predict_pose on a detected loop."""
    dev = use_backend(backend)
    import faiss
    from clslam_hip.diversity import DiversityBuffer
    from loop_closure_detection import Config as LcdConfig
    from loop_closure_detection import LoopClosureDetection
    from test_lcd_encoder import _weights
    torch.save(_weights()[1], tmp_path / 'mbv3.pth')
    monkeypatch.setenv('CLSLAM_MOBILENETV3_WEIGHTS', str(tmp_path / 'mbv3.pth'))
    B, cap, thr = 3, 2, 0.9999
    p = make_predictor(H, W, B, log_path=str(tmp_path))
    lcd = LoopClosureDetection(LcdConfig(tmp_path / 'x.yaml', 0.9, 2, 1))
    index = faiss.IndexIDMap(faiss.index_factory(512, 'Flat', faiss.METRIC_INNER_PRODUCT))
    div = DiversityBuffer(512, cap, thr, dev)
    frames = [synth.make_batch(1, H, W, seed=40 + (i % 4)) for i in range(6)]     # the camera returns after 4 frames
    stored = {}
    closures = []
    for step, frame in enumerate(frames, start=1):
        online = {k: v.clone() for k, v in frame.items()}      # adapt() moves the caller's dict to the device in place
        p._set_eval()
        feat = p.models['depth_encoder'](online['rgb', 0, 0].to(p.device))[4].detach().mean(-1).mean(-1)   # slam.py:143-147
        f_np = feat.cpu().numpy().copy()
        faiss.normalize_L2(f_np)
        sim = 0 if index.ntotal == 0 else index.search(f_np, 1)[0][0][0]
        added, removed, sim_dev = div.add(feat, step)                            # device-resident twin of the same decision
        assert abs(float(sim) - sim_dev) < 1e-5
        assert added == bool(sim < thr)
        if sim < thr:
            index.add_with_ids(f_np, np.array([step]))
            stored[step] = {k: v.clone() for k, v in online.items()}   # the buffer re-loads samples from disk: host copies
            if removed is not None:
                index.remove_ids(np.array([removed]))
                del stored[removed]
        assert sorted(faiss.vector_to_array(index.id_map).tolist()) == sorted(div.ids.tolist()) == sorted(stored)
        replay_ids = [i for i in stored if i != step][:B - 1]
        if len(replay_ids) == B - 1:
            training = online
            for i in replay_ids:
                training = _cat_dict(training, stored[i])
            outputs, losses = p.adapt(online, {k: v.clone() for k, v in training.items()}, steps=1)
        else:
            outputs, losses = p.adapt(online, None)
        assert torch.isfinite(losses['loss']).all() and outputs['cam_T_cam', 0, 1].shape[1:] == (4, 4)
        image = online['rgb', 1, 0]
        lcd.add(step, image.squeeze())                                           # slam.py:219
        ids, dist = lcd.search(step)                                             # slam.py:222
        for i, d in zip(ids, dist):
            T, cov = p.predict_pose(image[0], frames[i - 1]['rgb', 1, 0][0], as_numpy=True)   # slam.py:224-229
            assert T.shape == (4, 4) and np.isfinite(T).all() and d > 0.9
            closures.append((step, i))
    assert closures == [(5, 1), (6, 2)], closures                                # identical frames, id gap > 2
    assert p.engine.adam_step_count >= 2


@pytest.mark.parametrize('backend', BACKENDS)
def test_nan_loss_aborts_without_update(backend):
    """dpp.py:1115-1118: a NaN loss raises RuntimeError('NaN loss') and the optimizer step does not
    happen.  The product checks once per step, after the (device-guarded) Adam launch: weights,
    moments and the step count must be exactly what they were."""
    use_backend(backend)
    B = 2
    p = make_predictor(H, W, B)
    batch = synth.make_batch(B, H, W, seed=5)
    p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)      # one good step: moments non-zero
    w0, m0, v0 = p.engine.w.clone(), p.engine.m.clone(), p.engine.v.clone()
    count0 = p.engine.adam_step_count
    bad = {k: v.clone() for k, v in batch.items()}
    bad['relative_distance', 0] = torch.full_like(bad['relative_distance', 0], float('nan'))   # velocity loss -> NaN
    with pytest.raises(RuntimeError, match='NaN loss'):
        p.adapt(None, bad, steps=1)
    assert torch.equal(p.engine.w, w0) and torch.equal(p.engine.m, m0) and torch.equal(p.engine.v, v0)
    assert p.engine.adam_step_count == count0
    with pytest.raises(RuntimeError, match='NaN loss'):                     # eval path: immediate check
        p.adapt({k: v.clone() for k, v in bad.items()}, None)
    p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)      # and the predictor is still usable
    assert p.engine.adam_step_count == count0 + 1
    assert not torch.equal(p.engine.w, w0)


@pytest.mark.parametrize('backend', BACKENDS)
def test_uniform_random_image_content_matches_oracle(backend):
    """SURVEY.md 8d's second input family: uniform-random pixels instead of the smooth synthetic frames -- the
    4-way photometric min then changes winner from pixel to pixel (auto-mask, argmin planes and their gradients
    at their busiest).  Forward quantities of the step are held to the same 1e-4 as on the smooth frames."""
    use_backend(backend)
    B = 2
    p = make_predictor(H, W, B)
    o = make_oracle(H, W, B)
    batch = synth.make_batch(B, H, W, seed=33)
    g = torch.Generator().manual_seed(1234)
    for k in list(batch):
        if k[0] in ('rgb', 'rgb_aug'):
            batch[k] = torch.rand(batch[k].shape, generator=g)
    noise = synth.make_noise(B, H, W, seed=6)
    p.set_tie_break_noise(noise)
    out, losses = p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)
    o.set_adapt()
    oo, ol = o.process_batch(batch, noise, None)
    assert rel_err(out['depth', 0].cpu(), oo['depth', 0].detach()) < 1e-4
    assert rel_err(out['cam_T_cam', 0, 1].cpu(), oo['cam_T_cam', 0, 1].detach()) < 1e-4
    # A warped image of white noise has unit gradient per pixel, and the untrained pose net's large translations turn a
    # 1e-5 relative depth difference into ~3e-4 px of parallax: the synthesised frames are compared at 2e-3 of full
    # scale here (2e-5 on the smooth frames, tests/test_loss_stage.py); the per-sample means that make up the losses
    # average that away.
    for s in range(4):
        assert rel_err(out['rgb', -1, s].cpu(), oo['rgb', -1, s].detach()) < 2e-3
    for k, v in ol.items():
        assert abs(float(losses[k]) - float(v)) <= 1e-4 * max(abs(float(v)), 1e-3), k


@pytest.mark.gpu
@pytest.mark.parametrize('upload_all', [True, False])
def test_host_batch_upload_is_asynchronous_and_invisible(upload_all):
    """adapt() on a pinned HOST minibatch (slam.py hands over DataLoader output).  Default: EVERY entry of the caller's
    dict is moved to the device in place like dpp.py:916-917 -- on a copy stream, network inputs first, the entries the
    path never reads last (nothing of the step waits for them; the caller's stream is ordered behind them when the call
    returns).  upload_all_inputs=False (opt-in): only the 13 entries the path reads move.  Either way the step is bitwise
    the one computed from a minibatch already resident in HBM."""
    use_backend('hip')
    B = 3
    batch = synth.make_batch(B, H, W, seed=51)
    noise = {s: v.cuda() for s, v in synth.make_noise(B, H, W, seed=52).items()}
    results = []
    for mode in ('device', 'host'):
        p = make_predictor(H, W, B, upload_all_inputs=upload_all)
        p.set_tie_break_noise(noise)
        feed = {k: (v.cuda() if mode == 'device' else v.clone().pin_memory()) for k, v in batch.items()}
        for _ in range(2):                       # twice: the second call re-uploads into recycled blocks
            moved = dict(feed) if mode == 'host' else feed
            out, losses = p.adapt(None, moved, steps=1)
            if mode == 'host':                   # stream-ordered use of a never-read entry right after the call returns
                probe = moved['rgb_aug', 0, 3] if upload_all else None
                if probe is not None:
                    assert probe.is_cuda and torch.equal(probe.cpu(), batch['rgb_aug', 0, 3])
        torch.cuda.synchronize()
        if mode == 'host':
            assert all(moved[k].is_cuda for k in p.UPLOAD_FIRST + p.UPLOAD_REST)
            if upload_all:
                assert all(v.is_cuda for v in moved.values())                                   # dpp.py:916-917
                for k, v in moved.items():
                    assert torch.equal(v.cpu(), batch[k]), k
            else:
                assert sum(v.is_cuda for v in moved.values()) == 13
                assert not moved['rgb', -1, 2].is_cuda and not moved['rgb_aug', 0, 3].is_cuda  # never read: left alone
        assert all(v.is_cuda for v in losses.values())                                         # reference: device tensors
        results.append((out['depth', 0].clone(), out['cam_T_cam', 0, 1].clone(), float(losses['loss']), p.engine.w.clone()))
    for a, b in zip(results[0], results[1]):
        assert (a == b) if isinstance(a, float) else torch.equal(a, b)


@pytest.mark.gpu
def test_host_outputs_fast_path_is_opt_in_and_bitwise_the_device_result():
    """Opt-in host_pose_output=True: adapt(online, training) returns outputs['cam_T_cam', 0, +-1] and the loss dict as
    host tensors staged behind the forward (what slam.py:181-188 reads back every frame) -- bitwise the device result of
    the default, and reading them does not wait for the backward + optimizer step still running on the stream."""
    use_backend('hip')
    B = 3
    batch = {k: v.cuda() for k, v in synth.make_batch(B, H, W, seed=61).items()}
    noise = {s: v.cuda() for s, v in synth.make_noise(B, H, W, seed=62).items()}
    got = {}
    for host in (True, False):
        p = make_predictor(H, W, B, host_pose_output=host)
        p.set_tie_break_noise(noise)
        out, losses = p.adapt(None, dict(batch), steps=2)
        for f in (-1, 1):
            assert out['cam_T_cam', 0, f].is_cuda == (not host) and tuple(out['cam_T_cam', 0, f].shape) == (B, 4, 4)
        assert all(v.is_cuda == (not host) for v in losses.values())
        # what slam.py does with it
        T = torch.linalg.inv(out['cam_T_cam', 0, 1][0, :]).squeeze().cpu().detach().numpy()
        assert T.shape == (4, 4)
        got[host] = [out['cam_T_cam', 0, f].cpu().clone() for f in (-1, 1)] + [losses[k].cpu().clone() for k in sorted(losses)]
        torch.cuda.synchronize()
    for a, b in zip(got[True], got[False]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_default_outputs_live_on_one_device_in_all_three_modes():
    """The boundary the reference's callers rely on (dpp.py:916-923: everything `_process_batch` returns sits on
    self.device): a caller that feeds outputs['cam_T_cam', 0, f] into device math together with outputs['depth', 0] --
    re-projecting the predicted depth with the predicted pose, T @ K -- works after a training adapt(), after
    adapt(online, None) and after predict() without a device mismatch.  With the opt-in host fast path the same code
    raises for the training call only: that divergence is why it is not the default."""
    use_backend('hip')
    B = 2
    batch = {k: v.cuda() for k, v in synth.make_batch(B, H, W, seed=71).items()}

    def caller_math(out, inputs):
        T = out['cam_T_cam', 0, 1]
        K = inputs['camera_matrix', 0]
        depth = out['depth', 0]
        P = (K @ T)[:, :3, :]                                           # (B,3,4) on the device
        pix = torch.stack([torch.full_like(depth[:, 0], 10.0), torch.full_like(depth[:, 0], 7.0), torch.ones_like(depth[:, 0])], 1)
        cam = (inputs['inv_camera_matrix', 0][:, :3, :3] @ pix.flatten(2)) * depth.flatten(2)
        cam = torch.cat([cam, torch.ones_like(cam[:, :1])], 1)
        return (P @ cam).sum() + sum(v.sum() for v in (out['axis_angle', 0, 1], out['translation', 0, -1]))

    p = make_predictor(H, W, B)
    for mode in ('train', 'eval', 'predict'):
        feed = dict(batch)
        if mode == 'train':
            out, losses = p.adapt(None, feed, steps=1)
        elif mode == 'eval':
            out, losses = p.adapt(feed, None)
        else:
            out, losses = p.predict(feed), None
        assert all(v.device == p.device for v in out.values()), mode
        if losses is not None:
            assert all(v.device == p.device for v in losses.values()), mode
            assert torch.isfinite(losses['loss'] + out['depth', 0].mean())      # (1,) device loss + device tensor
        assert torch.isfinite(caller_math(out, feed))
    fast = make_predictor(H, W, B, host_pose_output=True)
    out, _ = fast.adapt(None, dict(batch), steps=1)
    with pytest.raises(RuntimeError, match='device'):
        caller_math(out, batch)
    out, _ = fast.adapt(dict(batch), None)                                      # forward-only calls: device tensors
    assert torch.isfinite(caller_math(out, batch))


@pytest.mark.parametrize('backend', BACKENDS)
def test_descriptor_pass_is_memoised_for_the_single_triplet_forward(backend):
    """SURVEY.md 8(f) rank 2 (slam.py:143-147): the per-frame `models['depth_encoder'](online_image)` pass is kept, and the
    adapt() / adapt(online, None) / predict() call that follows on the same (un-augmented) online frame takes its features
    instead of re-running the depth encoder -- bitwise the same outputs, losses and weights; a different image, an input the
    caller overwrote in place after the descriptor call, or re-packed weights are misses."""
    dev = use_backend(backend)
    B = 1
    batch = synth.make_batch(B, H, W, seed=21)
    other = synth.make_batch(B, H, W, seed=22)
    for d in (batch, other):          # the online dataset does not augment (datasets/utils.py:25,148-150): rgb_aug IS rgb
        for k in list(d):
            if k[0] == 'rgb_aug':
                d[k] = d['rgb', k[1], k[2]].clone()
    noise = synth.make_noise(B, H, W, seed=23)
    runs = {}
    for memo in (True, False):
        p = make_predictor(H, W, B)
        p.engine.descriptor_memo = memo
        p.set_tie_break_noise(noise)
        log = []
        # frame 1: descriptor, then a training step on the same frame (K = 0)
        x = batch['rgb', 0, 0].clone().to(dev)
        feat = p.models['depth_encoder'](x)[4].mean(-1).mean(-1).cpu()
        x.zero_()                                                     # the caller's tensor is the caller's: must not matter
        out, losses = p.adapt(dict(batch), dict(batch))
        log += [feat, out['depth', 0].cpu().clone(), out['cam_T_cam', 0, 1].cpu().clone(), losses['loss'].cpu().clone()]
        hits1 = p.engine.memo_hits
        # frame 2: descriptor, then the no-adaptation forward (slam.py:178) and predict() on the same frame
        p.models['depth_encoder'](other['rgb', 0, 0].to(dev))
        out, losses = p.adapt(dict(other), None)
        log += [out['depth', 0].cpu().clone(), losses['loss'].cpu().clone()]
        log.append(p.predict(dict(other))['depth', 0].cpu().clone())
        hits2 = p.engine.memo_hits
        # a descriptor of ANOTHER image does not serve this frame
        p.models['depth_encoder'](batch['rgb', 0, 0].to(dev))
        out, _ = p.adapt(dict(other), None)
        log.append(out['depth', 0].cpu().clone())
        hits3 = p.engine.memo_hits
        # re-packed weights void the memo
        p.models['depth_encoder'](other['rgb', 0, 0].to(dev))
        p.engine.pack()
        out, _ = p.adapt(dict(other), None)
        log.append(out['depth', 0].cpu().clone())
        p.engine.wait_training()
        log.append(p.engine.w.cpu().clone())
        runs[memo] = (log, (hits1, hits2, hits3, p.engine.memo_hits))
    # (a memo serves ONE forward -- the frame's adapt() right behind the descriptor pass, slam.py:143-178; the predict() of frame 2
    # recomputes: the content check is a stream synchronisation nobody else should pay)
    assert runs[True][1] == (1, 2, 2, 2) and runs[False][1] == (0, 0, 0, 0)
    for a, b in zip(runs[True][0], runs[False][0]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_images_that_need_a_conversion_wait_for_the_whole_upload():
    """non-fp32 / non-contiguous images are converted by a torch kernel on the MAIN stream; with the asynchronous
    upload that kernel used to be ordered behind the first image's copy only and could read planes still crossing PCIe.
    A pinned float64 host minibatch (exactly the fp32 values) and a channels-last one must give bitwise the fp32 result,
    repeatedly (fresh device blocks every call)."""
    use_backend('hip')
    B = 3
    batch = synth.make_batch(B, 192, 640, seed=77)
    noise = {s: v.cuda() for s, v in synth.make_noise(B, 192, 640, seed=78).items()}
    ref = None
    for kind in ('fp32', 'fp64', 'channels_last'):
        p = make_predictor(192, 640, B)
        p.set_tie_break_noise(noise)
        for rep in range(3):
            feed = {}
            for k, v in batch.items():
                if k[0] in ('rgb', 'rgb_aug') and kind == 'fp64':
                    v = v.double()
                elif k[0] in ('rgb', 'rgb_aug') and kind == 'channels_last':
                    v = v.contiguous(memory_format=torch.channels_last)
                feed[k] = v.clone(memory_format=torch.preserve_format).pin_memory()
            out, losses = p.adapt(dict(feed), None)
            got = (out['depth', 0].clone(), out['cam_T_cam', 0, 1].clone(), losses['loss'].clone())
            if ref is None:
                ref = got
            for a, b in zip(ref, got):
                assert torch.equal(a, b), (kind, rep)


@pytest.mark.parametrize('backend', BACKENDS)
def test_the_four_models_are_callables(backend):
    """dpp.py:931-936 and :957-965 call the four networks one after the other -- models['depth_decoder'](features)
    and models['pose_decoder']([features]) on the encoders' outputs.  Through the engine they give bitwise what predict() /
    predict_pose() compute (the same kernels on the same weights), in the reference's output formats."""
    dev = use_backend(backend)
    p = make_predictor(H, W, 2)
    batch = synth.make_batch(2, H, W, seed=31)
    p._set_eval()
    with torch.no_grad():
        ref = p.predict(batch)
        feats = p.models['depth_encoder'](batch['rgb_aug', 0, 0].to(dev))          # dpp.py:931: the augmented frame
        assert [tuple(f.shape) for f in feats] == [(2, c, H >> (k + 1), W >> (k + 1)) for k, c in enumerate((64, 64, 128, 256, 512))]
        disp = p.models['depth_decoder'](feats)
        assert list(disp) == [('disp', s) for s in (3, 2, 1, 0)]               # depth_decoder.py:53-69: coarse to fine
        for s in range(4):
            assert disp['disp', s].shape == (2, 1, H >> s, W >> s)
            assert torch.equal(disp['disp', s].cpu(), ref['disp', s].cpu())
        # pose: frames (-1, 0) of sample 0, as dpp.py:951-965 feeds them
        a, b = batch['rgb_aug', -1, 0][:1].to(dev), batch['rgb_aug', 0, 0][:1].to(dev)
        pf = p.models['pose_encoder'](torch.cat([a, b], 1))
        axis_angle, translation = p.models['pose_decoder']([pf])
        assert axis_angle.shape == (1, 2, 1, 3) and translation.shape == (1, 2, 1, 3)
        pose = p.engine.run_pose(a, b)                                        # (1, 12): what predict_pose() transforms
        assert torch.equal(torch.cat([axis_angle, translation], -1).reshape(1, 12).cpu(), pose.cpu())
        with pytest.raises(Exception, match='feature'):
            p.models['depth_decoder'](feats[:4] + [feats[4][:, :, :1]])

"""End-to-end backward parity, attributed: the gradients of all 36 trainable tensors of ONE adapt
step against the oracle's autograd, as full tensors (relative L2), plus the explanation DESIGN.md section 2 gives for the
residual demonstrated instead of asserted:

  (a) the 4-way min selection (`sel`) of the kernel path and of the oracle are compared pixel by pixel: they differ in a
      handful of pixels, and every one of them is a genuine near-tie (the two candidates the paths picked differ by
      less than the fp32 noise of the photometric maps);
  (b) with the oracle forced to the kernel path's selection, the remaining difference of every gradient tensor drops to
      the kernels' own accuracy;
  (c) the view synthesis has two more kinds of decisions -- the bilinear CELL a projected sample falls into and whether
      it is CLIPPED at the border (dpp.py:1013-1017, F.grid_sample) -- which 1e-5 px of difference in the projected
      position takes differently for a few samples.  The kernel path's decisions are read out (clslam_warp_cells_pyramid)
      and imposed on the oracle's written-out sampler (oracle.functional.grid_sample_border) together with the selection;
  (d) what is left then is rounding, and it is measured against the oracle re-run in FLOAT64 on the same decisions: the
      kernels and torch's own fp32 sit at comparable distances from it -- and most of that distance is not the backward
      pass at all but the 1e-7-level rounding of the FORWARD pass (pose matrices, disparities), which this loss amplifies
      because it shifts every sample of a frame coherently;
  (e) so the float64 oracle is finally evaluated AT the kernel path's forward point (its disparities, pose matrices and the
      rounded projection matrices it samples with): the backward arithmetic of the kernels alone is then as close to exact as
      torch's fp32 autograd is (<= 5e-4 on all 36 tensors, 192x640, B = 1 and B = 5).

64x128, B=2 on the emulator and on the GPU; 192x640, B=1 and B=5 (the benchmark minibatch) on the GPU."""
import pytest

from clslam_hip import synth
from emu_util import BACKENDS, use_backend
from helpers import attributed_gradient_errors, report_attribution
from predictor_util import make_predictor


def _run(backend, H, W, B, seed):
    dev = use_backend(backend)
    p = make_predictor(H, W, B)
    batch = synth.make_batch(B, H, W, seed=seed)
    noise = synth.make_noise(B, H, W, seed=seed + 10)
    p.set_tie_break_noise(noise)
    out, losses = p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)
    r = attributed_gradient_errors(p, batch, noise, dev)
    ol = r['oracle_losses']
    assert abs(float(losses['loss']) - float(ol['loss'])) <= 1e-4 * abs(float(ol['loss']))
    return r


@pytest.mark.parametrize('backend', BACKENDS)
def test_gradients_match_oracle_and_the_residual_is_selection_flips(backend, capsys):
    r = _run(backend, 64, 128, 2, seed=3)
    with capsys.disabled():
        report_attribution(f'{backend} 64x128 B=2', r)
    # a selection may only differ where the two candidates are closer than the photometric value can differ between the two
    # paths: the sampling positions agree to 2e-5 px (tests/test_warp_positions.py) and the synthetic frames change by <= ~1 per
    # pixel, so 2e-5 (measured: 4e-6 ... 9e-6 depending on the summation order of the decoder's launches)
    assert r['flips'] <= 2e-4 * r['npix'] and r['gap'] < 2e-5, (r['flips'], r['gap'])
    assert r['cell_flips'] + r['clip_flips'] <= 1e-3 * r['npix']
    rows = r['rows']
    for name, e_free, e_forced, norm, e_all, e_hip64, e_o64, e_bwd, e_bwd_t32, e_bwd_p, e_bwd_p_t32 in rows:
        assert e_free < 5e-2, (name, e_free)                 # full tensors (not norms / slices); dominated by the flips:
        assert e_forced < 1e-3, (name, e_forced)             # ... this is what is left once the selection is the same (4.9e-4)
        assert e_all < 1e-3, (name, e_all)                   # ... and with the sampler's decisions imposed as well
        assert e_bwd < 3e-4, (name, e_bwd, e_bwd_t32)       # the backward arithmetic alone: where torch's fp32 autograd is (1.2e-4 both)
    if r['flips']:   # measured on the emulator: 2 flipped pixels of 65536 -> 1.3e-2 on one tensor, 3.2e-4 with them matched
        assert max(x[2] for x in rows) < 0.2 * max(x[1] for x in rows)


# all 36 tensors, same decisions, same forward point -- disparities, pose matrices AND the rounded projection matrices (K T)[:3]
# the kernels actually sample with -- against the float64 oracle.  Measured on the MI355X: B = 1 1.4e-4 (torch's own fp32 at that
# point: 1.5e-4), B = 5 4.3e-4 (torch fp32 the same level: profiles/r04_pose_gradient.txt takes the pose path apart sample by
# sample -- dL/dP kernels 2.6e-5 ... 3.1e-4 / torch 2.4e-5 ... 3.4e-4, the pose chain of pose_bwd alone 7e-8).  Round 3 listed
# 7.4e-4 for the pose decoder here: that was the ulp-level difference between the kernel path's fp32 product K T and the
# oracle's float64 one (a coherent 1e-5 px shift of every sample of a frame), i.e. forward rounding, not backward arithmetic.
# Before the sampling position became one contraction-free chain shared by forward, backward and read-out (geometry_dev.h)
# B = 5 stood at 2e-3 ... 5e-3: two pixels per scale whose sample lay within an ulp of a cell boundary were differentiated in
# the neighbouring cell.
FULL_SIZE_TOL = {1: 5e-4, 5: 5e-4}
MEASURED = {1: dict(flips=15, cells=200, signs=10, e_free=1e-2, e_all=2.5e-3),
            5: dict(flips=45, cells=1000, signs=40, e_free=0.1, e_all=2.5e-2)}


@pytest.mark.gpu
@pytest.mark.parametrize('B', [1, 5])
def test_gradients_full_size_on_gpu(capsys, B):
    """The benchmark resolution, B = 1 and the benchmark minibatch B = 5: with selection, cells and clip flags of the kernel
    path imposed on the oracle, all 36 gradient tensors agree to FULL_SIZE_TOL."""
    r = _run('hip', 192, 640, B, seed=5)
    with capsys.disabled():
        report_attribution(f'hip 192x640 B={B}', r)
    # (the candidates a flipped pixel chose between differ by up to a few 1e-5: the tie-break noise is N(0, 1e-5) and the
    # photometric maps of two fp32 implementations differ by as much where the image gradient is steep)
    # Decisions that are NOT imposed stay bounded by what was measured: counts at about twice the measured level
    # (B = 1: 6 selections, 77 cells + 1 clip, 2 L1 signs; B = 5: 20, 488 + 5, 16), never a sample more than one cell away
    lim = MEASURED[B]
    assert r['flips'] <= lim['flips'] and r['gap'] < 1e-4, (r['flips'], r['gap'])
    assert r['cell_flips'] + r['clip_flips'] <= lim['cells'], (r['cell_flips'], r['clip_flips'])
    assert r['far_cells'] == 0, r['far_cells']
    assert r['sign_flips'] <= lim['signs'], r['sign_flips']
    for name, e_free, e_forced, norm, e_all, e_hip64, e_o64, e_bwd, e_bwd_t32, e_bwd_p, e_bwd_p_t32 in r['rows']:
        assert e_free < lim['e_free'], (name, e_free)        # measured 2.5e-3 (B = 1) / 5.1e-2 (B = 5, 20 flipped selections)
        assert e_all < lim['e_all'], (name, e_all)           # the oracle's OWN forward, same selection / cells / clips: 1.1e-3 / 1.3e-2
        # ... or, where torch's OWN fp32 backward at that very point is no better (round 6, B = 5: pose_2.weight kernels 5.05e-4,
        # torch fp32 5.08e-4 after the encoders' summation order changed with the 5/8 : 3/8 CU split), within 1.2x of torch's
        assert e_bwd_p < max(FULL_SIZE_TOL[B], 1.2 * e_bwd_p_t32), (name, e_bwd_p, e_bwd_p_t32, e_bwd, e_bwd_t32)

"""GPU tests of the reference-shaped entry points in sentinel-tree-cover_amd/job.py that the notebooks / the job call directly:
predict_subtile (src/download_and_predict_job.py:328-369), superresolve_large_tile (:95-147) and the tile loop's handling of a
Sen2Cor mask file (:685-697, :841-846)."""
import random

import numpy as np
import pytest

from tests.helpers import golden, synth

pytestmark = pytest.mark.gpu


def test_predict_subtile_edge_conventions():
    """all-zero window -> 255 fill; integer (uint16) input -> / 65535; centre crop when size < W - 14; against O.predict_subtile"""
    import torch
    from oracle import restate_model as M, restate_numpy as O
    from ttc import job, weights as Wt
    W, L = 44, 2
    w = Wt.synth_weights(7)
    sess = job.TTCSession(w, win_in=W, length=L, max_windows=1, dsen2_weights=None)
    net = M.TreeCoverNet(w, dtype=torch.float32)
    x = synth.synth_windows(seed=8, N=1, L=L, W=W)[0]
    for size in (W - 14, W - 18):                         # the reference's SIZE and a smaller centre crop
        got = job.predict_subtile(x, sess, size=size)
        ref = O.predict_subtile(x, net, size)
        assert got.shape == (size, size) and got.dtype == np.float32
        assert np.abs(got - ref).max() < 5e-5
    z = job.predict_subtile(np.zeros_like(x), sess, size=W - 14)
    zr = O.predict_subtile(np.zeros_like(x), net, W - 14)
    assert z.shape == zr.shape and z.dtype == zr.dtype and np.all(z == 255)
    xi = np.clip(x * 0.25 + 0.3, 0, 1)
    u16 = np.round(xi * 65535).astype(np.uint16)          # integer input: the reference divides by 65535 (job.py:346-350)
    got = job.predict_subtile(u16, sess, size=W - 14)
    ref = O.predict_subtile(u16, net, W - 14)
    assert np.abs(got - ref).max() < 5e-5


def test_superresolve_large_tile_dropin():
    """job.superresolve_large_tile(arr, sess): numpy in, the SAME array mutated and returned (also through a view, as the job calls
    it: s2[..., :10] = superresolve_large_tile(s2[..., :10], sess)), one device call; against the oracle and the reference's
    tiling quirks (the never-refined strip)"""
    import torch
    from oracle import restate_model as M, restate_numpy as O
    from ttc import job, weights as Wt
    sess = job.TTCSession(Wt.synth_weights(0), win_in=44, length=2, max_windows=1)
    net = M.DSen2Lite(Wt.load_dsen2(), dtype=torch.float32)
    rng = np.random.default_rng(21)
    s2 = (rng.random((2, 236, 300, 11)) * 0.6).astype(np.float32)       # an 11th channel stands for whatever the caller keeps beside the bands
    keep = s2.copy()
    ref = O.superresolve_large_tile(keep[..., :10].copy(), net)
    view = s2[..., :10]
    ret = job.superresolve_large_tile(view, sess)
    assert ret is view                                                  # mutated in place and returned
    s2[..., :10] = ret                                                  # the job's own statement (a no-op copy onto itself)
    assert np.abs(s2[..., :10] - ref).max() < 5e-5
    np.testing.assert_array_equal(s2[..., :4], keep[..., :4])           # the 10 m bands pass through
    np.testing.assert_array_equal(s2[..., 10], keep[..., 10])
    d = torch.from_numpy(keep[..., :10].copy()).cuda()
    out = job.superresolve_large_tile(d, sess)
    assert out is d and np.abs(d.cpu().numpy() - ref).max() < 5e-5
    with pytest.raises(ValueError):
        job.superresolve_large_tile(np.zeros((2, 8, 8, 9), np.float32), sess)


def test_tile_loop_honours_sen2cor_mask():
    """predict_tiles(mask=None) on a tile that has a cloudmask file: the Sen2Cor mask must be merged into the detected one
    (job.py:841-846) -- the tile takes the staged chain and equals process_tile -> superresolve -> predict_tile; a tile without the
    file takes the single call.  (the fast path used to drop raw['clm'] silently.)"""
    from ttc import job, weights as Wt
    W, size, L = 44, 30, 4
    w = Wt.synth_weights(0)
    sessions = [job.TTCSession(w, win_in=W, length=L) for _ in range(2)]
    raw_clm = synth.synth_raw_files(91, 6, 80, 88, True)
    raw_plain = synth.synth_raw_files(92, 6, 80, 88, False)
    assert raw_clm["clm"] is not None and raw_plain.get("clm") is None
    res = job.predict_tiles([(raw_clm, None), (raw_plain, None), (raw_clm, None)], sessions, size=size, want_status=True)
    assert len(res) == 3 and res[0][3] is True and res[2][3] is True
    s2, dates, interp, s1, dem, cloudshad, _ = job.process_tile(dict(raw_clm, clouds=None), sessions[0], sampler="expected")
    sessions[0].ctx.superresolve_tile(s2, quirks=True)
    want_f, want_u8 = job.predict_tile(s2, dates, interp, s1, dem, sessions[0], size=size)
    np.testing.assert_array_equal(res[0][1], want_u8)
    np.testing.assert_array_equal(res[2][1], want_u8)
    # and the Sen2Cor mask matters on this tile: without it the detected mask is different
    _, _, _, _, _, cs_noclm, _ = job.process_tile(dict(raw_clm, clouds=None, clm=None), sessions[0], sampler="expected")
    assert cs_noclm.shape != cloudshad.shape or not np.array_equal(cs_noclm.cpu().numpy() > 0, cloudshad.cpu().numpy() > 0)


def test_arena_size_is_checked_up_front():
    import torch
    from ttc import job, weights as Wt
    sess = job.TTCSession(Wt.synth_weights(0), win_in=44, length=4, dsen2_weights=None)
    arena = job.PinnedArena(torch, 3)
    arena.ahead = 8
    with pytest.raises(ValueError, match="PinnedArena"):
        job.predict_tiles([], [sess], size=30, arena=arena)

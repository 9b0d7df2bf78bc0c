"""GPU parity of the multi-temporal cloud / shadow detector against the golden vectors captured from the
reference and, stage by stage, against the CPU oracle."""
import numpy as np
import pytest

from tests.helpers import golden, synth

pytestmark = pytest.mark.gpu

STAGES = {1: "1_clm", 2: "2_shadow_candidates", 3: "3_shadows", 4: "4_cloud_candidates", 5: "5_brightness", 6: "6_white",
          7: "7_fcps", 70: "7_pfps", 8: "8_false_positives", 80: "8_shadows", 81: "8_nsr", 9: "9_shape", 10: "10_shadows",
          11: "11_extra_shadows"}


@pytest.fixture(scope="module")
def sess():
    from ttc import job, weights as Wt
    return job.TTCSession(Wt.synth_weights(0), win_in=44, length=4, dsen2_weights=None)


def _inputs(g, tag):
    seed, T, H, W, with_masks = (int(v) for v in g[f"{tag}_cfg"])
    img, dem, forest, core, near = synth.synth_detection_scene(seed, T, H, W)
    return img, dem, (forest if with_masks else None), ((core, near) if with_masks else None), (T, H, W)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_detection_matches_reference(sess, tag):
    g = golden("cloud_detection.npz")
    img, dem, forest, urban, (T, H, W) = _inputs(g, tag)
    clouds, fcps = sess.ctx.identify_clouds_shadows(img, dem, forest, urban)
    want_c = np.unpackbits(g[f"{tag}_clouds"])[:T * H * W].reshape(T, H, W).astype(bool)
    want_f = np.unpackbits(g[f"{tag}_fcps"])[:T * H * W].reshape(T, H, W).astype(bool)
    dc = (clouds.cpu().numpy() > 0) != want_c
    df = (fcps.cpu().numpy() > 0) != want_f
    print(f"[parity] detection {tag}: differing cloud flags {dc.mean():.2e}, fcps {df.mean():.2e}")
    # per-image float32 moments (z-scores, 1/blue statistics) are reduced in another order than numpy's pairwise sums: a
    # pixel within one ulp of a threshold may flip, and the closing dilations spread it over a few neighbours
    assert dc.mean() < 1e-3 and df.mean() < 1e-3


@pytest.mark.parametrize("tag", ["a", "c"])
def test_detection_stages_vs_oracle(sess, tag):
    from oracle import restate_clouds as C
    g = golden("cloud_detection.npz")
    img, dem, forest, urban, (T, H, W) = _inputs(g, tag)
    trace = {}
    C.identify_clouds_shadows(img.copy(), dem.copy(), forest, urban, trace=trace)
    bad = []
    for stage, name in STAGES.items():
        got, _ = sess.ctx.identify_clouds_shadows(img, dem, forest, urban, debug_stage=stage)
        d = ((got.cpu().numpy() > 0) != (trace[name] > 0)).mean()
        print(f"[parity] stage {stage:3d} {name:22s} differing {d:.2e}")
        if d > 1e-3:
            bad.append((stage, name, d))
    assert not bad, bad


def test_detection_full_tile_and_edge_cases(sess):
    """BASELINE-size tile (618 x 618, T = 12) and the degenerate inputs: a single date, two dates (the T <= 2 branch),
    an all-cloud stack, a cloud-free stack (scipy's empty-background distance transform), no masks."""
    from oracle import restate_clouds as C
    from ttc import job
    img, dem, forest, core, near = synth.synth_detection_scene(5, 12, 618, 618)
    got_c, got_f = job.identify_clouds_shadows(img, dem, None, sess, forest, (core, near))
    want_c, want_f = C.identify_clouds_shadows(img.copy(), dem.copy(), forest, (core, near))
    dc, df = ((got_c > 0) != (want_c > 0)).mean(), (got_f != np.asarray(want_f, dtype=bool)).mean()
    print(f"[parity] detection 618^2 T=12: differing cloud flags {dc:.2e}, fcps {df:.2e}")
    assert dc < 1e-3 and df < 1e-3
    small = synth.synth_detection_scene(9, 5, 64, 60)
    cases = {"T=1": small[0][:1], "T=2": small[0][:2], "T=3": small[0][:3],
             "all cloud": np.full((4, 64, 60, 10), 0.6, np.float32), "clear": np.full((4, 64, 60, 10), 0.05, np.float32) +
             np.linspace(0, 0.02, 10, dtype=np.float32)}
    for name, x in cases.items():
        x = np.ascontiguousarray(x, dtype=np.float32)
        gc, gf = job.identify_clouds_shadows(x, small[1], None, sess)
        wc, wf = C.identify_clouds_shadows(x.copy(), small[1].copy())
        d1, d2 = ((gc > 0) != (wc > 0)).mean(), (gf != np.asarray(wf, dtype=bool)).mean()
        print(f"[parity] detection {name}: differing {d1:.2e} / {d2:.2e}")
        assert d1 < 1e-3 and d2 < 1e-3, name
    with pytest.raises(RuntimeError, match="even"):
        sess.ctx.identify_clouds_shadows(small[0][:, :63], small[1][:63], None, (small[3][:63], small[4][:63]))

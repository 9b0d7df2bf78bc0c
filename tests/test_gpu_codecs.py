"""GPU parity of the storage codecs and Sentinel-1 scaling against the golden vectors
captured from the reference and against the CPU oracle."""
import numpy as np
import pytest

from tests.helpers import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    from ttc import job, weights as Wt
    return job.TTCSession(Wt.synth_weights(0), win_in=44, length=4, dsen2_weights=None)


def test_u16_codecs_bit_exact(sess):
    from ttc import job
    g = golden("codecs.npz")
    np.testing.assert_array_equal(job.to_float32(g["u16"], sess).cpu().numpy(), g["to_float32"])
    np.testing.assert_array_equal(job.to_int16(g["f32"], sess), g["to_int16"])
    # full u16 range incl. 0 / 65535, and out-of-range / exact-boundary floats
    allv = np.arange(65536, dtype=np.uint16)
    f = job.to_float32(allv, sess).cpu().numpy()
    np.testing.assert_array_equal(f, np.float32(allv) / np.float32(65535))
    from oracle import restate_numpy as R
    x = np.concatenate([f, np.float32([-0.5, 1.5, 0.99999994, 1e-9])])
    np.testing.assert_array_equal(job.to_int16(x, sess), R.to_int16(x))
    assert job.to_float32(np.zeros((0,), np.uint16), sess).numel() == 0


def test_sentinel1_db_matches_reference(sess):
    from ttc import job
    from oracle import restate_numpy as R
    g = golden("codecs.npz")
    got = job.sentinel1_to_db(g["s1_u16"], sess).cpu().numpy()
    assert np.abs(got - g["s1_db"]).max() < 2e-6           # log10f vs numpy's log10: <= 2 ulp of a [0, 1] value
    # tile-sized: 12 images, saturated blocks, an all-saturated image, odd counts
    rng = np.random.default_rng(5)
    u = rng.integers(0, 65536, size=(12, 155, 157, 2)).astype(np.uint16)
    u[1, :40, :40] = 65535
    u[2] = 65535
    u[3, ..., 0] = 0
    got = job.sentinel1_to_db(u, sess).cpu().numpy()
    want = R.s1_to_db(u)
    assert np.abs(got - want).max() < 2e-6
    u1 = rng.integers(0, 65536, size=(1, 3, 5, 2)).astype(np.uint16)
    assert np.abs(job.sentinel1_to_db(u1, sess).cpu().numpy() - R.s1_to_db(u1)).max() < 2e-6

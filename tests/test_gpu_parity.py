"""HIP path vs CPU oracle on the same seeded inputs, through the C ABI (-m gpu)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render_both(ptgpu, oracle, preset, W, H, S, use_bvh, depth=10, frame=0, prev=None):
    osc = oracle.OracleScene(preset, W, H, use_bvh=use_bvh)
    ex = osc.export()
    ref = np.zeros((H, W, 3), np.float32) if prev is None else prev.copy()
    ref, ref_rays = osc.update(S, depth, frame, buffer=ref)
    desc = oracle.to_ptgpu_desc(ptgpu, ex)
    sc = ptgpu.Scene(desc, 0)
    params = ptgpu.PtParams(W, H, S, depth, 0, 1 if use_bvh else 0)
    cam = ptgpu.PtCamera.from_floats(ex["camera"])
    out = np.zeros((H, W, 3), np.float32) if prev is None else prev.copy()
    rays = sc.update(params, cam, frame, out)
    sc.close()
    return ref, ref_rays, out, rays


def _report(ref, out):
    d = np.abs(ref - out)
    bad = (d > 0).any(axis=2)
    return "mismatching pixels %d / %d, max |d| %.3e" % (bad.sum(), bad.size, d.max())


# bit-exact scenes: no libm call feeds control flow except powf in dielectrics
@pytest.mark.parametrize("preset,W,H,S,bvh", [
    ("small", 200, 100, 4, False),           # BASELINE config 1
    ("small", 200, 100, 4, True),
    ("aras", 160, 90, 4, False),
    ("random_spheres", 120, 80, 4, False),
    ("random_spheres", 120, 80, 4, True),
])
def test_exact_parity(ptgpu, oracle, preset, W, H, S, bvh):
    ref, ref_rays, out, rays = _render_both(ptgpu, oracle, preset, W, H, S, bvh)
    assert rays == ref_rays, "ray_count %d vs oracle %d; %s" % (rays, ref_rays, _report(ref, out))
    assert np.array_equal(ref, out), _report(ref, out)


# noise textures call sinf() on colour only: tolerance 2e-6 absolute per channel
@pytest.mark.parametrize("preset,W,H,S,bvh", [
    ("two_perlin_spheres", 160, 90, 4, False),
    ("perlin_spheres", 96, 54, 2, True),
])
def test_noise_parity(ptgpu, oracle, preset, W, H, S, bvh):
    ref, ref_rays, out, rays = _render_both(ptgpu, oracle, preset, W, H, S, bvh)
    assert rays == ref_rays
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-6)

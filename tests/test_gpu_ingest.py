"""Device-side data ingestion against the reference's own evaluator: scipy's
RegularGridInterpolator as called at /root/reference/source/model_setup.py:84-86, and the NumPy even-odd
rule standing in for the shapely loop of model_setup.py:68-72 (shapely is not installed: parity with
`contains` is unpinned for points exactly on the outline)."""
import numpy as np
import pytest
from scipy.interpolate import RegularGridInterpolator

pytestmark = pytest.mark.gpu


def _grid(rng, n, lo, hi, uniform):
    if uniform:
        return np.linspace(lo, hi, n)
    g = np.sort(rng.uniform(lo, hi, n))
    g[0], g[-1] = lo, hi
    return g


@pytest.mark.parametrize("uniform", [True, False])
@pytest.mark.parametrize("flip", [(False, False), (True, False), (False, True)])
def test_interp_matches_scipy_bit_for_bit(uniform, flip):
    from shakti_fenics_amd._lib import interp_regular_grid
    rng = np.random.default_rng(7)
    xg, yg = _grid(rng, 301, -5e4, 7e4, uniform), _grid(rng, 187, 1e5, 1.6e5, uniform)
    f = rng.normal(size=(xg.size, yg.size)) * 1e3
    if flip[0]:
        xg, f = xg[::-1], f[::-1]
    if flip[1]:
        yg, f = yg[::-1], f[:, ::-1]
    n = 200_000
    # inside, outside (extrapolated), and exactly on grid lines / corners
    px = rng.uniform(-6e4, 8e4, n)
    py = rng.uniform(0.9e5, 1.7e5, n)
    px[:300] = np.resize(xg, 300)
    py[:150] = np.resize(yg, 150)
    px[300:304] = [xg.min(), xg.max(), xg.min(), xg.max()]
    py[300:304] = [yg.min(), yg.min(), yg.max(), yg.max()]
    ref = RegularGridInterpolator((xg, yg), f, bounds_error=False, fill_value=None)(np.column_stack((px, py)))
    got = interp_regular_grid(px, py, xg, yg, f)
    assert np.array_equal(got, ref)
    # float32 data (BedMachine's bed is float32) goes through scipy's generic evaluator, which rounds differently
    f32 = f.astype(np.float32)
    ref32 = RegularGridInterpolator((xg, yg), f32, bounds_error=False, fill_value=None)(np.column_stack((px, py)))
    assert ref32.dtype == np.float64
    assert np.array_equal(interp_regular_grid(px, py, xg, yg, f32), ref32)


def test_interp_nan_and_errors():
    from shakti_fenics_amd._lib import ShaktiHipError, interp_regular_grid
    xg, yg = np.linspace(0, 1, 5), np.linspace(0, 2, 4)
    f = np.arange(20.0).reshape(5, 4)
    got = interp_regular_grid([0.5, np.nan], [1.0, 1.0], xg, yg, f)
    assert got[0] == RegularGridInterpolator((xg, yg), f)([[0.5, 1.0]])[0] and np.isnan(got[1])
    with pytest.raises(ShaktiHipError, match="strictly ascending"):
        interp_regular_grid([0.5], [1.0], np.array([0.0, 0.5, 0.5, 1.0, 2.0]), yg, f)
    with pytest.raises(ValueError):
        interp_regular_grid([0.5], [1.0], xg, yg, f.T)
    assert interp_regular_grid([], [], xg, yg, f).size == 0


def test_points_in_polygon_matches_even_odd_rule():
    from shakti_fenics_amd._lib import points_in_polygon
    from shakti_fenics_amd.model_setup import _points_in_polygon
    rng = np.random.default_rng(11)
    # a star-shaped, non-convex outline with > 1024 vertices (more than one LDS tile) and horizontal edges
    m = 2500
    th = np.sort(rng.uniform(0, 2 * np.pi, m))
    rad = 4e3 * (1 + 0.4 * np.sin(7 * th) + 0.1 * rng.normal(size=m))
    poly = np.column_stack((1e4 + rad * np.cos(th), -3e3 + rad * np.sin(th)))
    poly[10, 1] = poly[11, 1]
    px, py = rng.uniform(2e3, 1.8e4, 300_000), rng.uniform(-1.1e4, 5e3, 300_000)
    ref = _points_in_polygon(px, py, poly)
    assert 0.1 < ref.mean() < 0.9
    assert np.array_equal(points_in_polygon(px, py, poly), ref)
    closed = np.vstack((poly, poly[:1]))
    assert np.array_equal(points_in_polygon(px, py, closed), ref)


def test_model_setup_ingest_device_equals_host():
    """md.interp_data / md.set_lake_bdry give the same arrays on both paths (model_setup.py:68-91)."""
    from shakti_fenics_amd.comm import world
    from shakti_fenics_amd.mesh import rectangle_mesh
    from shakti_fenics_amd.model_setup import model_setup
    dom = rectangle_mesh(60, 25, 2.0e4, 8.0e3, order="morton")
    rng = np.random.default_rng(3)
    x_d, y_d = np.linspace(-5e3, 2.5e4, 140), np.linspace(1.2e4, -4e3, 90)   # y descending, like BedMachine
    bed = rng.normal(size=(y_d.size, x_d.size))
    outline = np.array([[4e3, 1e3], [1.5e4, 2e3], [1.2e4, 6.5e3], [6e3, 5e3]])
    out = {}
    for mode in ("host", "device"):
        md = model_setup(world(), dom)
        assert md.ingest == "device"   # the default: no silent host fallback
        md.ingest = mode
        interp = md.interp_data("z_b", x_d, y_d, bed)
        md.set_lake_bdry(outline)
        out[mode] = (md.z_b.x.array.copy(), md.lake_bdry.x.array.copy())
        assert interp((1.0e4, 3.0e3)) == interp(np.array([[1.0e4, 3.0e3]]))[0]   # still the scipy object
    assert np.array_equal(out["host"][0], out["device"][0])
    assert np.array_equal(out["host"][1], out["device"][1])
    assert 0 < out["host"][1].sum() < out["host"][1].size

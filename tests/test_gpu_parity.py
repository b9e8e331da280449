"""GPU parity: every HIP entry point (through the C-ABI, via the plugin-surface classes) against the CPU oracle
on identical seeded inputs.  Tolerances (fp32): RGB pixel MAE <= 1e-4 of [0,1]; thermal <= 1e-4 of
the normalised range (x (Tmax-Tmin) in degrees); depths <= 1e-4 relative; per-sample tensors max-abs 2e-5.
"""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import hotpath as H
from tests import helpers
from thermo_nerf_amd import (FieldHeadNames, FieldHeadNamesT, Frustums, RayBundle, RaySamples, ThermalRenderer, _hip,
                             synthetic)
from thermo_nerf_amd.renderers import AccumulationRenderer, DepthRenderer, RGBRenderer
from thermo_nerf_amd.samplers import PDFSampler, UniformLinDispPiecewiseSampler, UniformSampler

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
RGB_MAE = 1e-4
THERMAL_MAE = 1e-4
DEPTH_REL = 1e-4
MEDIAN_TIE = 1e-5  # |cumulative weight - 0.5| at the oracle's median index below which an index flip counts as a tie


def gpu_model(kind="stress", S=48, small=True, family="lane_ray", **over):
    """The library picks the kernel family by call size (lane = ray from ~60-80 k rays up, one ray per wave below).  The
    tests in this module use a few hundred rays but target the throughput (lane = ray) kernels, so the models they build
    ask for them through ``config.kernel_family`` (-> ``tn_render_config.kernel_family``; the library reads no environment
    variable); ``family="auto"`` / ``"ray_per_wave"`` cover the automatic choice and the other form."""
    model, sd, ocfg = helpers.build(kind, S, small, **over)
    gm = copy.deepcopy(model).to(DEV).eval()
    gm.config.kernel_family = family
    return gm, sd, ocfg


def bundle(o, d, cam=None):
    """nerfstudio's Cameras.generate_rays always fills camera_indices; default to camera 0."""
    if cam is None:
        cam = torch.zeros((o.shape[0], 1), dtype=torch.long)
    return RayBundle(origins=o.to(DEV), directions=d.to(DEV), camera_indices=cam.to(DEV))


def assert_close(got, want, atol, rtol=0.0, name=""):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = (got - want).abs()
    lim = atol + rtol * want.abs()
    assert bool((err <= lim).all()), f"{name}: max err {err.max().item():.3e} (limit {atol}+{rtol}*|x|)"


# --------------------------------------------------------------------------------------------------
# samplers / weights
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("near", [0.0, 0.05])
@pytest.mark.parametrize("n", [256, 96, 7])
def test_sample_initial(near, n):
    R = 37
    o, d = helpers.rays(8, 8)
    o, d = o[:R], d[:R]
    nears, fars = torch.full((R, 1), near), torch.full((R, 1), 1000.0)
    want = H.sample_initial(nears, fars, n, None)
    rb = bundle(o, d)
    rb.nears, rb.fars = nears.to(DEV), fars.to(DEV)
    s = UniformLinDispPiecewiseSampler(single_jitter=True).eval()
    rs = s(rb, num_samples=n)
    assert_close(rs.frustums.starts, want.starts, 0, 2e-7, "starts")
    assert_close(rs.frustums.ends, want.ends, 0, 2e-7, "ends")
    assert_close(rs.spacing_starts, want.spacing_starts, 0, 0, "spacing")
    # training: single jitter supplied explicitly
    t = torch.rand(R, 1, generator=torch.Generator().manual_seed(3))
    want = H.sample_initial(nears, fars, n, t)
    s.train()
    rs = s(rb, num_samples=n, t_rand=t.to(DEV))
    assert_close(rs.frustums.ends, want.ends, 0, 5e-7, "ends(train)")
    assert_close(rs.spacing_ends, want.spacing_ends, 1e-7, 0, "spacing(train)")


@pytest.mark.parametrize("near,far", [(0.0, 1000.0), (0.05, 6.0)])
@pytest.mark.parametrize("n", [256, 7])
def test_uniform_sampler_and_its_pdf_resampling(near, far, n):
    """NS UniformSampler (proposal_initial_sampler="uniform", REF thermal_nerf_model.py:164-170): identity spacing functions,
    which the PDFSampler that follows inherits for its own bins."""
    R = 37
    o, d = helpers.rays(8, 8)
    nears, fars = torch.full((R, 1), near), torch.full((R, 1), far)
    rb = bundle(o[:R], d[:R])
    rb.nears, rb.fars = nears.to(DEV), fars.to(DEV)
    s = UniformSampler(single_jitter=True).eval()
    want = H.sample_initial(nears, fars, n, None, uniform=True)
    rs = s(rb, num_samples=n)
    assert rs.uniform_spacing
    assert_close(rs.frustums.starts, want.starts, 0, 2e-7, "starts")
    assert_close(rs.frustums.ends, want.ends, 0, 2e-7, "ends")
    assert_close(rs.spacing_starts, want.spacing_starts, 0, 0, "spacing")
    # bins are linear in distance: equal widths
    widths = (rs.frustums.ends - rs.frustums.starts)[..., 0]
    assert (widths - (far - near) / n).abs().max().item() <= 2e-6 * far
    t = torch.rand(R, 1, generator=torch.Generator().manual_seed(5))
    want_t = H.sample_initial(nears, fars, n, t, uniform=True)
    rs_t = s.train()(rb, num_samples=n, t_rand=t.to(DEV))
    assert_close(rs_t.frustums.ends, want_t.ends, 0, 5e-7, "ends(train)")
    # PDF resampling on top of it
    g = torch.Generator().manual_seed(n)
    w = torch.rand(R, n, 1, generator=g) ** 6
    w[0] = 0.0
    n_out = 96
    want_p = H.sample_pdf(want, w, n_out, None)
    rs_p = PDFSampler(single_jitter=True).eval()(rb, rs, w.to(DEV), num_samples=n_out)
    assert rs_p.uniform_spacing
    assert_close(rs_p.spacing_starts, want_p.spacing_starts, 5e-6, 0, "pdf spacing")
    assert_close(rs_p.frustums.ends, want_p.ends, 1e-6 + 5e-6 * (far - near), 0, "pdf ends")  # distance = linear in spacing
    u = torch.rand(R, 1, generator=g)
    want_p = H.sample_pdf(want, w, n_out, u)
    rs_p = PDFSampler(single_jitter=True).train()(rb, rs, w.to(DEV), num_samples=n_out, u_rand=u.to(DEV))
    assert_close(rs_p.frustums.ends, want_p.ends, 1e-6 + 5e-6 * (far - near), 0, "pdf ends(train)")


@pytest.mark.parametrize("uniform", [False, True])
@pytest.mark.parametrize("n,n_out", [(256, 96), (96, 48), (7, 192), (64, 5)])
def test_samplers_with_one_draw_per_bin_edge(n, n_out, uniform):
    """NS single_jitter=False (ThermalNerfactoModelConfig.use_single_jitter, REF thermal_nerf_model.py:176): SpacedSampler draws
    rand((R, n+1)), PDFSampler rand((R, n_out+1)) — one stratified draw per bin edge rather than one per ray.  Same kernels,
    bit 1 of their uniform_spacing argument; results against the oracle, and the single-draw layout still rejected by size."""
    R = 37
    o, d = helpers.rays(8, 8)
    near, far = (0.05, 6.0) if uniform else (0.05, 1000.0)
    nears, fars = torch.full((R, 1), near), torch.full((R, 1), far)
    rb = bundle(o[:R], d[:R])
    rb.nears, rb.fars = nears.to(DEV), fars.to(DEV)
    g = torch.Generator().manual_seed(100 * n + n_out)
    t = torch.rand(R, n + 1, generator=g)
    cls = UniformSampler if uniform else UniformLinDispPiecewiseSampler
    s = cls(single_jitter=False).train()
    want = H.sample_initial(nears, fars, n, t, uniform=uniform)
    rs = s(rb, num_samples=n, t_rand=t.to(DEV))
    assert_close(rs.spacing_starts, want.spacing_starts, 1e-7, 0, "spacing")
    assert_close(rs.spacing_ends, want.spacing_ends, 1e-7, 0, "spacing ends")
    assert_close(rs.frustums.ends, want.ends, 1e-6 * (far - near) if uniform else 0, 5e-7, "ends")
    # the draws really are per edge: a single-jitter run on column 0 differs
    one = cls(single_jitter=True).train()(rb, num_samples=n, t_rand=t[:, :1].contiguous().to(DEV))
    assert not torch.equal(one.spacing_ends, rs.spacing_ends)
    with pytest.raises(ValueError, match="draws"):
        s(rb, num_samples=n, t_rand=t[:, :1].contiguous().to(DEV))
    # its own draws: stratified, i.e. edge j stays inside [centre(j-1), centre(j)] and the bins stay ordered
    own = s(rb, num_samples=n)
    sp = torch.cat([own.spacing_starts[..., 0], own.spacing_ends[:, -1:, 0]], -1).cpu()
    assert bool((sp[:, 1:] >= sp[:, :-1]).all()) and float(sp.min()) >= 0.0 and float(sp.max()) <= 1.0
    assert float((sp[0] - sp[1]).abs().max()) > 0  # rays differ
    centres = (torch.arange(n + 1, dtype=torch.float32) - 0.5).clamp(0, n) / n
    assert bool((sp >= centres[None] - 1e-6).all()) and bool((sp[:, :-1] <= centres[None, 1:] + 1e-6).all())

    w = torch.rand(R, n, 1, generator=g) ** 6
    w[0] = 0.0
    u = torch.rand(R, n_out + 1, generator=g)
    want_p = H.sample_pdf(want, w, n_out, u)
    ps = PDFSampler(single_jitter=False).train()
    rs_p = ps(rb, rs, w.to(DEV), num_samples=n_out, u_rand=u.to(DEV))
    assert_close(rs_p.spacing_starts, want_p.spacing_starts, 5e-6, 0, "pdf spacing")
    assert_close(rs_p.spacing_ends, want_p.spacing_ends, 5e-6, 0, "pdf spacing ends")
    # (a spacing error e moves the distance by (far - near) e under the uniform map, by 2 x^2 e at distance x under 1/(2-2s))
    x = want_p.ends.double()
    lim = 1e-6 + 5e-6 * ((far - near) if uniform else 2 * torch.clamp(x, min=1.0) ** 2)
    assert bool(((rs_p.frustums.ends.cpu().double() - x).abs() <= lim).all()), "pdf ends"
    with pytest.raises(ValueError, match="draws"):
        ps(rb, rs, w.to(DEV), num_samples=n_out, u_rand=u[:, :1].contiguous().to(DEV))
    own = ps(rb, rs, w.to(DEV), num_samples=n_out)
    sp = torch.cat([own.spacing_starts[..., 0], own.spacing_ends[:, -1:, 0]], -1)
    assert bool((sp[:, 1:] >= sp[:, :-1]).all())


@pytest.mark.parametrize("n", [48, 64, 192, 256, 5])
def test_get_weights(n):
    g = torch.Generator().manual_seed(n)
    R = 33
    deltas = torch.rand(R, n, 1, generator=g) * 0.1
    dens = torch.exp(torch.randn(R, n, 1, generator=g) * 3)
    dens[0] = 0.0
    dens[1] = float("inf")  # alpha*T = nan -> nan_to_num
    want = H.get_weights(deltas, dens)
    fr = Frustums(origins=torch.zeros(R, n, 3, device=DEV), directions=torch.zeros(R, n, 3, device=DEV),
                  starts=torch.zeros(R, n, 1, device=DEV), ends=deltas.to(DEV))
    rs = RaySamples(frustums=fr, deltas=deltas.to(DEV))
    got = rs.get_weights(dens.to(DEV))
    assert_close(got, want, 2e-6, 1e-5, "weights")


@pytest.mark.parametrize("n_in,n_out", [(256, 96), (96, 48), (96, 64), (96, 192), (64, 300)])
def test_sample_pdf(n_in, n_out):
    g = torch.Generator().manual_seed(n_in * 1000 + n_out)
    R = 29
    nears, fars = torch.zeros(R, 1), torch.full((R, 1), 1000.0)
    prev = H.sample_initial(nears, fars, n_in, None)
    w = torch.rand(R, n_in, 1, generator=g) ** 6
    w[0] = 0.0  # all-zero weights -> uniform resample through the histogram padding
    w[1] = 0.0
    w[1, n_in // 2] = 1.0  # a spike
    want = H.sample_pdf(prev, w, n_out, None)
    o, d = helpers.rays(8, 8)
    rb = bundle(o[:R], d[:R])
    rb.nears, rb.fars = nears.to(DEV), fars.to(DEV)
    rs0 = UniformLinDispPiecewiseSampler(single_jitter=True).eval()(rb, num_samples=n_in)
    sampler = PDFSampler(single_jitter=True).eval()
    rs = sampler(rb, rs0, w.to(DEV), num_samples=n_out)
    # cdf rounding (fp32 wave scan here vs torch's sequential double accumulate) moves a bin edge by a few 1e-6 in
    # spacing units; the spacing->euclidean map 1/(2-2s) amplifies that by 2*x^2 at distance x
    assert_close(rs.spacing_starts, want.spacing_starts, 5e-6, 0, "pdf spacing")
    x = want.ends.double()
    lim = 1e-6 + 5e-6 * 2 * torch.clamp(x, min=1.0) ** 2
    assert bool(((rs.frustums.ends.cpu().double() - x).abs() <= lim).all()), "pdf ends"
    # training jitter
    u = torch.rand(R, 1, generator=g)
    want = H.sample_pdf(prev, w, n_out, u)
    sampler.train()
    rs = sampler(rb, rs0, w.to(DEV), num_samples=n_out, u_rand=u.to(DEV))
    assert_close(rs.spacing_ends, want.spacing_ends, 5e-6, 0, "pdf spacing (train)")


# --------------------------------------------------------------------------------------------------
# renderers
# --------------------------------------------------------------------------------------------------
def test_thermal_renderer_golden_g1(golden_dir):
    """Against outputs of the REAL reference ThermalRenderer (fixture G1)."""
    g = np.load(os.path.join(golden_dir, "thermal_renderer.npz"))
    th, w = torch.from_numpy(g["thermal"]).to(DEV), torch.from_numpy(g["weights"]).to(DEV)
    r = ThermalRenderer()
    r.eval()
    assert_close(r(th, w), torch.from_numpy(g["out_eval"]), 2e-6, 0, "thermal eval")
    r.train()
    got = r(th, w).cpu().numpy()
    want = g["out_train"]
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=0, atol=2e-6)


def test_rgb_renderer_golden_g4(golden_dir):
    """Against outputs of the REAL reference RGBTRenderer (fixture G4: rgb_concat/rgbt_renderer.py:62-81,159-174, the
    reference's fork of nerfstudio's RGBRenderer, background "last_sample"): tn_composite_fwd with 3 and 4 channels."""
    g = np.load(os.path.join(golden_dir, "rgbt_renderer.npz"))
    rgbt, w = torch.from_numpy(g["rgbt"]).to(DEV), torch.from_numpy(g["weights"]).to(DEV)
    r = RGBRenderer(background_color="last_sample")
    for mode in ("eval", "train"):
        r.train(mode == "train")
        for c, x in ((3, rgbt[..., :3].contiguous()), (4, rgbt)):
            got, want = r(x, w).cpu().numpy(), g[f"out{c}_{mode}"]
            np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
            np.testing.assert_array_equal(np.isinf(got), np.isinf(want))
            fin = np.isfinite(want)
            np.testing.assert_allclose(got[fin], want[fin], rtol=0, atol=2e-6, err_msg=f"C={c} {mode}")


def test_thermal_field_head_golden_g5(golden_dir):
    """Against outputs of the REAL reference BaseThermalFieldHead (fixture G5: Linear 64 -> 1, no activation)."""
    from thermo_nerf_amd.thermal_nerf.thermal_field_head import ThermalFieldHead

    g = np.load(os.path.join(golden_dir, "thermal_field_head.npz"))
    head = ThermalFieldHead(in_dim=64)
    with torch.no_grad():
        head.net.weight.copy_(torch.from_numpy(g["weight"]))
        head.net.bias.copy_(torch.from_numpy(g["bias"]))
    head.to(DEV)
    y = head(torch.from_numpy(g["x"]).to(DEV))
    assert_close(y, torch.from_numpy(g["y"]), 2e-6, 0, "thermal head")
    y3 = head(torch.from_numpy(g["x"]).to(DEV).view(1, 257, 64))  # leading dims preserved like nn.Linear
    assert y3.shape == (1, 257, 1)


@pytest.mark.parametrize("avg", [1, 0])
def test_field_wiring_golden_g6(golden_dir, avg):
    """Against outputs of the REAL reference ThermalNerfactoTField.forward (fixture G6, thermal_field.py:108-201, run over
    oracle-built nerfstudio base classes): the HIP field, loaded with the fixture's state dict, through Field.forward in
    eval and train mode, with and without the average appearance embedding."""
    import json

    from thermo_nerf_amd import SceneContraction
    from thermo_nerf_amd.thermal_nerf.thermal_field import ThermalNerfactoTField

    g = np.load(os.path.join(golden_dir, "thermal_field_wiring.npz"))
    c = json.loads(str(g["config"]))
    f = ThermalNerfactoTField(torch.tensor([[-1.0] * 3, [1.0] * 3]), num_images=c["num_images"], num_levels=c["num_levels"],
                              base_res=c["base_res"], max_res=c["max_res"], log2_hashmap_size=c["log2_hashmap_size"],
                              use_average_appearance_embedding=bool(avg),
                              spatial_distortion=SceneContraction(order=float("inf")), pass_thermal_gradients=True)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    missing, unexpected = f.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("max_res" in m or "num_levels" in m or "log2_hashmap_size" in m) for m in missing), missing
    f.to(DEV)
    pos, dirs, cam = (torch.from_numpy(g[k]).to(DEV) for k in ("positions", "directions", "camera_indices"))
    R, S = pos.shape[:2]
    N = R * S
    # frustums whose get_positions() returns the fixture's positions: one sample per "ray", origin = position, start = end = 0
    fr = Frustums(origins=pos.reshape(N, 1, 3), directions=dirs.reshape(N, 1, 3).contiguous(),
                  starts=torch.zeros(N, 1, 1, device=DEV), ends=torch.zeros(N, 1, 1, device=DEV))
    rs = RaySamples(frustums=fr, camera_indices=cam.reshape(N, 1, 1))
    for mode in ("eval", "train"):
        f.train(mode == "train")
        with torch.no_grad():
            out = f(rs)
        tag = f"avg{avg}_{mode}"
        assert set(out.keys()) == {FieldHeadNames.RGB, FieldHeadNames.DENSITY, FieldHeadNamesT.THERMAL}
        assert_close(out[FieldHeadNames.DENSITY].view(R, S, 1), torch.from_numpy(g[f"density_{tag}"]), 1e-6, 3e-5, f"density {tag}")
        assert_close(out[FieldHeadNames.RGB].view(R, S, 3), torch.from_numpy(g[f"rgb_{tag}"]), 5e-6, 0, f"rgb {tag}")
        assert_close(out[FieldHeadNamesT.THERMAL].view(R, S, 1), torch.from_numpy(g[f"thermal_{tag}"]), 1e-5, 0, f"thermal {tag}")


@pytest.mark.parametrize("n", [48, 64, 192, 3])
def test_rgb_depth_accumulation(n):
    g = torch.Generator().manual_seed(77 + n)
    R = 41
    rgb = torch.rand(R, n, 3, generator=g)
    rgb[2, 1, 0] = float("nan")
    w = torch.rand(R, n, 1, generator=g)
    w = w / w.sum(1, keepdim=True) * torch.rand(R, 1, 1, generator=g) * 1.2
    w[0] = 0.0
    edges = torch.sort(torch.rand(R, n + 1, generator=g) * 5, dim=-1).values
    starts, ends = edges[:, :-1, None], edges[:, 1:, None]
    for training in (False, True):
        r = RGBRenderer().train(training)
        want = H.render_rgb(rgb, w, training)
        got = r(rgb.to(DEV), w.to(DEV)).cpu()
        m = torch.isfinite(want)
        assert bool((torch.isnan(got) == torch.isnan(want)).all())
        assert_close(got[m], want[m], 2e-6, 0, f"rgb train={training}")
    fr = Frustums(origins=torch.zeros(R, n, 3, device=DEV), directions=torch.zeros(R, n, 3, device=DEV),
                  starts=starts.to(DEV), ends=ends.to(DEV))
    rs = RaySamples(frustums=fr)
    assert_close(AccumulationRenderer()(w.to(DEV)), H.render_accumulation(w), 2e-6, 0, "acc")
    assert_close(DepthRenderer("expected")(w.to(DEV), rs), H.render_depth_expected(w, starts, ends), 1e-6, 1e-5, "exp depth")
    got = DepthRenderer("median")(w.to(DEV), rs).cpu()
    want = H.render_depth_median(w, starts, ends)
    # a cumulative weight within rounding of 0.5 may pick the neighbouring sample: allow isolated flips
    bad = ((got - want).abs() > 1e-6).float().mean().item()
    assert bad <= 0.05, f"median depth mismatch fraction {bad}"


# --------------------------------------------------------------------------------------------------
# fields (per-sample)
# --------------------------------------------------------------------------------------------------
def sample_positions(n, seed=5):
    """positions covering inside (|x|<1), the contracted shell and far away"""
    p = (synthetic.counter_uniform(n * 3, seed).view(n, 3) * 2 - 1)
    scale = torch.tensor([0.5, 1.0, 3.0, 50.0])[torch.arange(n) % 4][:, None]
    p = p * scale
    p[0] = torch.tensor([0.0, 0.0, 0.0])
    p[1] = torch.tensor([1.0, 0.25, -0.5])  # on the contraction boundary
    return p.contiguous()


@pytest.mark.parametrize("kind", ["init", "stress"])
@pytest.mark.parametrize("dense_mb", [0, 64])
def test_proposal_density_fn(kind, dense_mb):
    gm, sd, ocfg = gpu_model(kind, 48)
    pos = sample_positions(4099)
    for lvl in (0, 1):
        gm.proposal_networks[lvl].dense_budget_bytes = dense_mb << 20
        want = H.proposal_density(sd, lvl, pos, ocfg)
        got = gm.proposal_networks[lvl].density_fn(pos.to(DEV))
        assert_close(got, want, 1e-6, 2e-5, f"prop density {lvl}")


@pytest.mark.parametrize("kind", ["init", "stress"])
@pytest.mark.parametrize("dense_mb", [0, 64])
def test_field_density_and_heads(kind, dense_mb):
    gm, sd, ocfg = gpu_model(kind, 48)
    gm.field.dense_budget_bytes = dense_mb << 20
    n = 2053
    pos = sample_positions(n, seed=9)
    want_density, want_geo = H.field_density(sd, pos, ocfg)
    got_density, got_geo = gm.field.density_at(pos.to(DEV))
    assert_close(got_density, want_density, 1e-6, 3e-5, "density")
    assert_close(got_geo, want_geo, 2e-5, 2e-5, "geo")
    dirs = synthetic.counter_uniform(n * 3, 21).view(n, 3) * 2 - 1
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    cam = (synthetic.counter_uniform(n, 22) * 8).long().clamp(0, 7)[:, None]
    for training in (False, True):
        want_rgb, want_th = H.field_outputs(sd, dirs, want_geo, cam, ocfg, training)
        fr = Frustums(origins=pos.to(DEV)[:, None, :], directions=dirs.to(DEV)[:, None, :],
                      starts=torch.zeros(n, 1, 1, device=DEV), ends=torch.zeros(n, 1, 1, device=DEV))
        rs = RaySamples(frustums=fr, camera_indices=cam.to(DEV)[:, None, :])
        gm.field.train(training)
        out = gm.field.get_outputs(rs, density_embedding=want_geo.to(DEV)[:, None, :])
        assert_close(out[FieldHeadNames.RGB][:, 0], want_rgb, 2e-6, 0, f"rgb train={training}")
        assert_close(out[FieldHeadNamesT.THERMAL][:, 0], want_th, 5e-6, 0, f"thermal train={training}")
    gm.field.eval()


def test_field_requires_camera_indices():
    gm, _, _ = gpu_model("init", 48)
    fr = Frustums(origins=torch.zeros(4, 1, 3, device=DEV), directions=torch.ones(4, 1, 3, device=DEV),
                  starts=torch.zeros(4, 1, 1, device=DEV), ends=torch.zeros(4, 1, 1, device=DEV))
    with pytest.raises(AttributeError, match="Camera indices"):
        gm.field.get_outputs(RaySamples(frustums=fr, camera_indices=None), density_embedding=torch.zeros(4, 1, 15, device=DEV))


# --------------------------------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------------------------------
def check_outputs(got, want, tag):
    rgb_mae = (got["rgb"].cpu() - want["rgb"]).abs().mean().item()
    th_mae = (got["thermal"].cpu() - want["thermal"]).abs().mean().item()
    assert rgb_mae <= RGB_MAE, f"{tag}: rgb MAE {rgb_mae:.3e}"
    assert th_mae <= THERMAL_MAE, f"{tag}: thermal MAE {th_mae:.3e}"
    assert (got["rgb"].cpu() - want["rgb"]).abs().max().item() <= 20 * RGB_MAE, f"{tag}: rgb max"
    assert_close(got["accumulation"], want["accumulation"], 2e-5, 0, f"{tag}: accumulation")
    assert_close(got["expected_depth"], want["expected_depth"], 1e-6, DEPTH_REL, f"{tag}: expected_depth")
    # median depths [NS DepthRenderer("median"), REF thermal_nerf_model.py:238-239,267-270]: a ray may differ from the oracle only
    # where the oracle's cumulative weight sits within MEDIAN_TIE of the 0.5 split next to its median index (a genuine tie, which
    # rounding may resolve either way) AND the answer is the neighbouring step on that side; every other ray meets DEPTH_REL.
    ties = getattr(want, "median_ties", None)
    assert ties is not None, f"{tag}: the oracle outputs carry no median_ties (use H.get_outputs / H.Outputs)"
    for k in ("depth", "prop_depth_0", "prop_depth_1"):
        if k not in want:
            continue
        g, w, t = got[k].cpu(), want[k], ties[k]
        rel_to = lambda ref: (g - ref).abs() / ref.abs().clamp_min(1e-6)
        ok = rel_to(w) <= DEPTH_REL
        tie_up = (rel_to(t["above"].reshape(w.shape)) <= DEPTH_REL) & (t["margin_above"].reshape(w.shape) <= MEDIAN_TIE)
        tie_dn = (rel_to(t["below"].reshape(w.shape)) <= DEPTH_REL) & (t["margin_below"].reshape(w.shape) <= MEDIAN_TIE)
        bad = ~(ok | tie_up | tie_dn)
        if bad.any():
            i = int(bad.reshape(-1).nonzero()[0])
            raise AssertionError(
                f"{tag}: {k}: {int(bad.sum())} of {bad.numel()} rays differ from the oracle's median depth without a tie; first: ray {i} "
                f"got {g.reshape(-1)[i]:.7g} want {w.reshape(-1)[i]:.7g} (neighbours {t['below'].reshape(-1)[i]:.7g} / {t['above'].reshape(-1)[i]:.7g}, "
                f"|cw-0.5| below {t['margin_below'].reshape(-1)[i]:.3g} above {t['margin_above'].reshape(-1)[i]:.3g})")
        assert (~ok).float().mean().item() <= 0.02, f"{tag}: {k}: {(~ok).float().mean().item()} of the rays sit on a median tie"
    assert set(k for k, v in want.items() if isinstance(v, torch.Tensor)) <= set(got)


IMPLS = {"mfma": (True, True), "valu": (True, False), "modular": (False, False)}  # (fused, use_mfma)


@pytest.mark.parametrize("kind", ["init", "stress", "scene"])
@pytest.mark.parametrize("S", [48, 64, 192])
@pytest.mark.parametrize("impl", list(IMPLS))
def test_get_outputs_eval(kind, S, impl):
    gm, sd, ocfg = gpu_model(kind, S)
    fused = impl
    gm.config.fused, gm.config.use_mfma = IMPLS[impl]
    gm.config.mlp_precision = "f32"
    o, d = helpers.rays(16, 16, view=S % 8)
    want = H.get_outputs(sd, o, d, None, ocfg)
    with torch.no_grad():
        got = gm(bundle(o, d))
    for k in ("rgb", "accumulation", "depth", "expected_depth", "prop_depth_0", "prop_depth_1", "thermal"):
        assert got[k].shape == want[k].shape, k
    check_outputs(got, want, f"{kind}/S{S}/fused={fused}")


@pytest.mark.parametrize("widths", [(32, 16, 48), (16, 64, 8), (64, 64, 24), (48, 32, 64), (128, 128, 128), (100, 130, 66), (256, 72, 200)])
@pytest.mark.parametrize("kind", ["stress", "scene"])
def test_other_mlp_widths_run_stage_by_stage(kind, widths):
    """hidden_dim / hidden_dim_color / hidden_dim_transient [REF thermal_nerf_model.py:96-114 forwards them to the field; 64 in every
    reference config] other than 64 (VERDICT r5 missing #4): the fused kernels are laid out for 64-wide layers, so such a field
    runs the same arithmetic one launch per nerfstudio module / layer (field.staged; up to 256 — wider raises; layers above 64 take
    tn_linear_fwd's LDS-tiled form, widths that are no multiple of 4 its scalar stores).  Eval outputs, the
    field's plugin surface (get_density / get_outputs) and the chunked camera render against the oracle, same tolerances."""
    from thermo_nerf_amd.engine import RayRenderEngine

    hd, hc, ht = widths
    gm, sd, ocfg = gpu_model(kind, 48, hidden_dim=hd, hidden_dim_color=hc, hidden_dim_transient=ht)
    assert gm.field.staged and not gm._fusable()
    assert sd["field.mlp_base.mlp.layers.0.weight"].shape == (hd, 32) and sd["field.mlp_head.layers.1.weight"].shape == (hc, hc)
    assert sd["field.mlp_thermal.layers.0.weight"].shape == (64, 15) and sd["field.mlp_thermal.layers.1.weight"].shape == (ht, 64)
    assert sd["field.field_head_thermal.net.weight"].shape == (1, ht)
    o, d = helpers.rays(16, 16, view=3)
    want = H.get_outputs(sd, o, d, None, ocfg)
    with torch.no_grad():
        got = gm(bundle(o, d))
        whole = gm.get_outputs_for_camera_ray_bundle(RayBundle(origins=o.view(16, 16, 3).to(DEV), directions=d.view(16, 16, 3).to(DEV),
                                                                     camera_indices=torch.zeros(16, 16, 1, dtype=torch.long, device=DEV)))
    check_outputs(got, want, f"{kind}/widths{widths}")
    assert (whole["rgb"].reshape(-1, 3) - got["rgb"]).abs().max().item() <= 1e-6
    with pytest.raises(RuntimeError, match="staged"):
        RayRenderEngine(gm)
    with pytest.raises(NotImplementedError, match="up to 256"):
        helpers.build(kind, 48, hidden_dim=512)


@pytest.mark.parametrize("R,P,S", [(1, (7, 5), 3), (2, (64, 32), 1), (63, (33, 17), 48), (65, (256, 96), 13), (127, (300, 130), 200),
                                   (257, (1, 1), 2), (64, (2, 3), 64), (1000, (96, 256), 7), (5, (512, 256), 256)])
def test_get_outputs_on_odd_shapes(R, P, S):
    """Ray counts around the wave and tile sizes (1, 63, 65, 257), proposal and field sample counts that are multiples of nothing
    (one sample per level included; a second proposal level with MORE samples than the first): the eval outputs against the
    oracle at the usual tolerances — the kernels' tilings must not leak into the results (tools/odd_shapes_probe.py prints the
    distances: 1e-7 everywhere)."""
    gm, sd, ocfg = gpu_model("scene", S, num_proposal_samples_per_ray=P)
    o, d = helpers.rays(40, 40, view=2)
    o, d = o[:R].contiguous(), d[:R].contiguous()
    want = H.get_outputs(sd, o, d, None, ocfg)
    with torch.no_grad():
        got = gm(bundle(o, d))
    for k in ("rgb", "accumulation", "depth", "expected_depth", "prop_depth_0", "prop_depth_1", "thermal"):
        assert got[k].shape == want[k].shape, k
    check_outputs(got, want, f"odd/R{R}/P{P}/S{S}")


@pytest.mark.parametrize("kind", ["stress", "scene"])
@pytest.mark.parametrize("S", [48, 50, 192, 13])
def test_sample_split_tiles_match_the_oracle_and_the_serial_march(kind, S):
    """Sample-split tiles (tn_render_config.sample_split; round 5): a tile's march cut into k segments on k waves, chained by
    segments_combine_kernel — w_i = T_j w_i^local.  Every k is held to the oracle's tolerances (incl. the median's tie gate) and sits
    within rounding of the serial march (k = 1); S = 50 leaves a ragged last segment, S = 13 is too short to split (the library
    answers 1).  The small calls of this module run the library's own choice (k = 6 ... 8) everywhere else."""
    gm, sd, ocfg = gpu_model(kind, S)
    gm.config.fused, gm.config.use_mfma, gm.config.mlp_precision = True, True, "f32"
    o, d = helpers.rays(21, 19, view=S % 8)
    want = H.get_outputs(sd, o, d, None, ocfg)
    lib = _hip.load()
    outs = {}
    for k in (1, 2, 3, 5, 8, 0):
        gm.config.sample_split = k
        with torch.no_grad():
            outs[k] = {n: v.clone() for n, v in gm(bundle(o, d)).items()}
        check_outputs(outs[k], want, f"{kind}/S{S}/sample_split={k}")
    gm.config.sample_split = 0
    for k in (2, 3, 5, 8, 0):
        for n in ("rgb", "thermal", "accumulation", "expected_depth"):
            scale = 1.0 if n != "expected_depth" else float(outs[1][n].abs().max())
            assert (outs[k][n] - outs[1][n]).abs().max().item() <= 3e-6 * scale, (k, n)
    # what the library reports it used: capped by the scratch (12 k + S floats per ray within 256 + 97) and by 8 samples per segment
    rc = _hip.tn_render_config()
    rc.num_proposal_samples[0], rc.num_proposal_samples[1], rc.num_nerf_samples = 256, 96, S
    rc.kernel_family = 1
    _, _, fld = gm._c_structs()
    cap = max(min(8, (353 - S) // 12, S // 8), 1)
    for k in (1, 2, 5, 8, 100):
        rc.sample_split = k
        assert lib.tn_render_sample_split(fld, rc, 399) == (1 if cap < 2 else min(k, cap))
    rc.sample_split = 0
    auto_small, auto_frame = lib.tn_render_sample_split(fld, rc, 399), lib.tn_render_sample_split(fld, rc, 640000)
    assert auto_frame == 1 and auto_small == (cap if cap >= 2 else 1)
    if S == 192:  # an 80 000-ray shard of the metric's frame (1 250 tiles on 2 048 wave slots) is marched in pieces
        assert lib.tn_render_sample_split(fld, rc, 80000) >= 4 and lib.tn_render_sample_split(fld, rc, 160000) >= 4
    if S == 48:  # ... and so is the reference's 65 536-ray chunk as one call; a 1080p frame is not
        assert lib.tn_render_sample_split(fld, rc, 65536) >= 2 and lib.tn_render_sample_split(fld, rc, 1080 * 1920) == 1
    rc.training = 1
    assert lib.tn_render_sample_split(fld, rc, 399) == 1
    rc.training, rc.early_stop_transmittance = 0, 1e-3
    assert lib.tn_render_sample_split(fld, rc, 399) == 1
    rc.early_stop_transmittance, rc.kernel_family = 0.0, 2
    assert lib.tn_render_sample_split(fld, rc, 399) == 1


@pytest.mark.parametrize("dense_mb", [64, 0])
def test_proposal_pass_in_density_segments_gives_the_one_launch_kernels_bits(dense_mb):
    """Calls under 3 072 tiles run the lane = ray proposal pass as four launches (density segments per (tile, segment), scans + PDF
    walks per tile: proposal_density_segments_kernel / proposal_resample_kernel); larger calls run proposal_rays_kernel.  The
    weights are formed in sample order in both, so the resampled edges — and with them EVERY output — are the same bits: the
    first 100 032 rays of a 262 208-ray call (4 097 tiles: one launch) against those rays as a call of their own (1 563 tiles:
    segments), dense and hashed proposal grids, the field pass held to whole tiles on both sides."""
    from thermo_nerf_amd.engine import RayRenderEngine

    gm, sd, ocfg = gpu_model("scene", 48, small=False, family="lane_ray")
    gm.config.dense_grid_budget_mb = dense_mb
    gm.invalidate_prepared()
    o3, d3, _ = synthetic.orbit_camera_rays(512, 513, view=2)
    n_big, n_small = 4097 * 64, 1563 * 64
    o, d = o3.reshape(-1, 3)[:n_big].contiguous().to(DEV), d3.reshape(-1, 3)[:n_big].contiguous().to(DEV)
    eng = RayRenderEngine(gm, chunk=n_big)
    big = {k: v.clone() for k, v in eng.render(o, d, sample_split=1).items()}
    eng_s = RayRenderEngine(gm, chunk=n_small)
    small = eng_s.render(o[:n_small].contiguous(), d[:n_small].contiguous(), sample_split=1)
    torch.cuda.synchronize()
    for k in ("rgb", "thermal", "accumulation", "depth", "prop_depth_0", "prop_depth_1"):
        assert torch.equal(small[k], big[k][:n_small]), (k, (small[k] - big[k][:n_small]).abs().max().item())
    # and a ragged call (the last tile partly idle)
    rag = eng_s.render(o[:70001].contiguous(), d[:70001].contiguous(), sample_split=1)
    torch.cuda.synchronize()
    for k in ("rgb", "depth", "prop_depth_0", "prop_depth_1"):
        assert torch.equal(rag[k], big[k][:70001]), k


def test_bf16x6_split_is_exact_and_its_six_products_are_an_fp32_product():
    """The operand split of mlp_precision="bf16x6" on its own (tn_bf16x6_split_product = the field kernel's BF16x6::split and
    product list): 1e6 random fp32 pairs over 60 binades + edge cases (powers of two, values on bf16 rounding boundaries, fp32
    sub-normals, the largest magnitudes).  (1) every piece is a bf16 value and a = p1 + p2 + p3 EXACTLY for 2^-110 <= |a| <=
    (2 - 2^-8) 2^127 (above it bf16(a) rounds to infinity; below it the third piece falls under bf16's smallest sub-normal and the
    sum is off by < 2^-133); (2) round to nearest leaves |p2| <= 2^-8 |a| and |p3| <= 2^-16 |a|; (3) the six products, accumulated
    in fp32, are within 2^-23 relative of the fp64 product on every pair (dropped terms <= (2 + 2^-8) 2^-24, plus five fp32 additions)."""
    lib = _hip.load()
    g = torch.Generator().manual_seed(11)
    n = 1_000_000
    def rnd():
        mant = 1.0 + torch.rand(n, generator=g, dtype=torch.float64)
        expo = torch.randint(-30, 31, (n,), generator=g).double()
        sign = torch.where(torch.rand(n, generator=g) < 0.5, -1.0, 1.0).double()
        return (sign * mant * torch.pow(torch.tensor(2.0, dtype=torch.float64), expo)).float()
    a, b = rnd(), rnd()
    edge = torch.tensor([1.0, -1.0, 2.0 ** -20, 2.0 ** 40, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -7 + 2.0 ** -8, 1.0 + 2.0 ** -23,
                         0.99609375, 3.0, 2.0 ** -100, 2.0 ** -109, 2.0 ** -110 * 1.9999999, 3.38e38, -3.39e38, 0.0, -0.0])
    a[: edge.numel()] = edge
    b[: edge.numel()] = edge.flip(0)
    ad, bd = a.to(DEV), b.to(DEV)
    pa, pb = torch.empty((n, 3), device=DEV), torch.empty((n, 3), device=DEV)
    out = torch.empty(n, device=DEV)
    _hip.check(lib.tn_bf16x6_split_product(ad.data_ptr(), bd.data_ptr(), n, pa.data_ptr(), pb.data_ptr(), out.data_ptr(), _hip.current_stream()),
               "tn_bf16x6_split_product")
    torch.cuda.synchronize()
    pa, pb, out = pa.cpu(), pb.cpu(), out.cpu()
    for x, p in ((a, pa), (b, pb)):
        assert torch.equal(p.to(torch.bfloat16).float(), p), "a piece is not a bf16 value"
        assert torch.equal(p.double().sum(dim=1), x.double()), "the three pieces do not add up to the operand exactly"
        mag = x.abs().double()
        assert (p[:, 1].abs().double() <= mag * 2.0 ** -8).all() and (p[:, 2].abs().double() <= mag * 2.0 ** -16).all()
    want = a.double() * b.double()
    ok = (want.abs() >= 2.0 ** -100) & (want.abs() <= 2.0 ** 100)  # (products that neither overflow fp32 nor lose piece products to underflow)
    assert int(ok.sum()) > 0.99 * n
    rel = ((out.double() - want).abs() / want.abs())[ok]
    print(f"bf16x6 six-product sum vs the fp64 product: max rel {rel.max().item():.3e} = {rel.max().item() * 2 ** 23:.2f} x 2^-23, mean {rel.mean().item():.2e}")
    assert rel.max().item() <= 2.0 ** -23  # (measured: 0.77 x 2^-23; the worst case of the analysis is 2.3 x)
    # the range: below 2^-110 the third piece is lost (an absolute error under 2^-133), fp32 sub-normals split like any value
    tiny = torch.tensor([2.0 ** -111, 3.0 * 2.0 ** -120, 2.0 ** -126, 1.1754942e-38, 1.4e-45, 7.1e-40], dtype=torch.float32)
    td = tiny.to(DEV)
    pt, po = torch.empty((tiny.numel(), 3), device=DEV), torch.empty(tiny.numel(), device=DEV)
    ones = torch.ones_like(td)
    p1 = torch.empty((tiny.numel(), 3), device=DEV)
    _hip.check(lib.tn_bf16x6_split_product(td.data_ptr(), ones.data_ptr(), tiny.numel(), pt.data_ptr(), p1.data_ptr(), po.data_ptr(),
                                           _hip.current_stream()), "tn_bf16x6_split_product")
    torch.cuda.synchronize()
    err = (pt.cpu().double().sum(dim=1) - tiny.double()).abs()
    assert (err <= 2.0 ** -133).all(), err
    # the largest magnitudes: bf16(x) is infinite above (2 - 2^-8) 2^127 — the split's upper limit (the field's activations are O(1))
    big = torch.tensor([3.3895e38, 3.40e38], dtype=torch.float32).to(DEV)
    pbig, o2 = torch.empty((2, 3), device=DEV), torch.empty(2, device=DEV)
    pone = torch.empty((2, 3), device=DEV)
    _hip.check(lib.tn_bf16x6_split_product(big.data_ptr(), torch.ones(2, device=DEV).data_ptr(), 2, pbig.data_ptr(), pone.data_ptr(),
                                           o2.data_ptr(), _hip.current_stream()), "tn_bf16x6_split_product")
    torch.cuda.synchronize()
    assert torch.isfinite(pbig[0]).all() and float(pbig[0].double().sum()) == float(big[0].double())
    assert torch.isinf(pbig[1, 0])


def test_bf16_mfma_takes_subnormal_inputs_as_they_are():
    """What the header's range note rests on: v_mfma_f32_32x32x16_bf16 does not flush sub-normal bf16 INPUTS (pieces of operands
    below 2^-110 ... 2^-126 are sub-normal bf16 values): A = 2^-130 (a bf16 sub-normal), B = 2^100 -> 16 x 2^-30."""
    lib = _hip.load()
    out = torch.zeros(4, device=DEV)
    for av, bv, want in ((2.0 ** -130, 2.0 ** 100, 16 * 2.0 ** -30), (2.0 ** -133, 2.0 ** 120, 16 * 2.0 ** -13), (1.5, 2.0, 48.0)):
        _hip.check(lib.tn_bf16_mfma_value_probe(av, bv, out.data_ptr(), _hip.current_stream()), "tn_bf16_mfma_value_probe")
        torch.cuda.synchronize()
        assert float(out[0]) == want, (av, bv, float(out[0]), want)


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("kind", ["init", "stress", "scene"])
@pytest.mark.parametrize("S", [48, 64, 192])
def test_get_outputs_eval_split_precision(kind, S, precision):
    """Opt-in split-precision field kernels (f16x3: each fp32 product = three f16 MFMA products of two pieces per operand;
    bf16x6: six bf16 products of three pieces = an exact 24-bit split; fp32 accumulate): held to the SAME tolerances as the
    exact-fp32 kernels, and compared against the fp32 MFMA kernel on the same rays."""
    gm, sd, ocfg = gpu_model(kind, S)
    gm.config.fused, gm.config.use_mfma, gm.config.mlp_precision = True, True, precision
    _, _, fld = gm._c_structs()
    assert fld.prepared_f16x3 if precision == "f16x3" else fld.prepared_bf16x6, "the split-precision prepare produced no blob"
    assert not (fld.prepared_f16x3 and fld.prepared_bf16x6)
    o, d = helpers.rays(20, 20, view=(S + 3) % 8)
    want = H.get_outputs(sd, o, d, None, ocfg)
    with torch.no_grad():
        got = gm(bundle(o, d))
        gm.config.mlp_precision = "f32"
        ref = gm(bundle(o, d))
    check_outputs(got, want, f"{precision} {kind}/S{S}")
    for k in ("rgb", "thermal"):
        d_split = (got[k].cpu() - want[k]).abs().max().item()
        d_f32 = (ref[k].cpu() - want[k]).abs().max().item()
        assert d_split <= max(8 * d_f32, 5e-6), f"{k}: split {d_split:.2e} vs fp32 kernel {d_f32:.2e}"


def _to64(sd):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


@pytest.mark.parametrize("kind", ["init", "stress", "scene", "trained"])
def test_bf16x6_is_an_fp32_evaluation_in_another_order(kind):
    """The claim behind mlp_precision="bf16x6" (DESIGN 5.3): three bf16 pieces hold an fp32 operand exactly and the six products kept
    leave a per-product error of 2^-23 relative — fp32's own rounding size —, so the kernel is AN fp32 evaluation of the field, as
    far from the exact (fp64) value as any other summation order.  Yardstick: the oracle run in fp64 on the same weights and rays.
    The bf16x6 frame's mean distance from it must not exceed the exact-fp32-MFMA kernel's own distance by more than a quarter
    (both are rounding noise of the same size; f16x3, with 22-bit operands, is reported beside them), on the three synthetic fills
    and on TRAINED weights: the 1000-iteration config-1 problem (tests/helpers.py), trained here on the HIP path."""
    S = 24 if kind == "trained" else 64
    if kind == "trained":
        from thermo_nerf_amd import training as TR

        prob = helpers.config1_problem()
        gm = copy.deepcopy(prob["model"]).to(DEV).train()
        ocfg, sd0 = prob["ocfg"], prob["sd"]
        params = [p for n, p in gm.named_parameters() if not n.startswith("camera_optimizer")]
        opt = torch.optim.Adam(params, lr=1e-2, eps=1e-15, fused=True)
        o_all, d_all, cam_all = prob["o"].to(DEV), prob["d"].to(DEV), prob["cam"].to(DEV)
        img, th, idx = prob["image"].to(DEV), prob["thermal"].to(DEV), prob["idx"].to(DEV)
        jitter = prob["jitter"].squeeze(-1).to(DEV)
        for i in range(400):
            gm.set_step(i)
            ix = idx[i]
            rb = gm.collider(RayBundle(origins=o_all[ix], directions=d_all[ix], camera_indices=cam_all[ix]))
            out = TR.get_outputs_train(gm, rb, jitter=jitter[i].contiguous())
            b = {"image": img[ix], "thermal": th[ix]}
            loss = sum(gm.get_loss_dict(out, b, gm.get_metrics_dict(out, b)).values())
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        gm.eval()
        gm.config.kernel_family = "lane_ray"
        gm.invalidate_prepared()
        sd = {**sd0, **{k: v.detach().cpu() for k, v in gm.state_dict().items() if k in sd0}}
        o, d = prob["held_out"]["o"], prob["held_out"]["d"]
        anneal = float(gm.proposal_sampler._anneal)
    else:
        gm, sd, ocfg = gpu_model(kind, S)
        o, d = helpers.rays(24, 24, view=5)
        anneal = 1.0
    want64 = H.get_outputs(_to64(sd), o.double(), d.double(), None, ocfg, anneal=anneal)
    dist = {}
    # (the serial march on both sides: sample-split tiles — the fp32 kernel's choice for a call this small — sum the weights
    # pairwise-like and land at HALF the serial march's distance from fp64, 3.0e-8 against 6.5e-8; the split-precision kernels
    # have no segmented form)
    gm.config.sample_split = 1
    for precision in ("f32", "bf16x6", "f16x3"):
        gm.config.fused, gm.config.use_mfma, gm.config.mlp_precision = True, True, precision
        with torch.no_grad():
            got = gm(bundle(o, d))
        dist[precision] = {k: (got[k].cpu().double().reshape(-1) - want64[k].reshape(-1)).abs().mean().item() for k in ("rgb", "thermal")}
    print(f"{kind}: mean |x - fp64 oracle|  " + "  ".join(f"{p}: rgb {v['rgb']:.2e} thermal {v['thermal']:.2e}" for p, v in dist.items()))
    for k in ("rgb", "thermal"):
        assert dist["bf16x6"][k] <= 1.25 * dist["f32"][k] + 2e-8, (kind, k, dist)


@pytest.mark.parametrize("impl", list(IMPLS))
def test_get_outputs_training_mode(impl):
    """Train-mode forward: near plane 0.05, per-camera appearance, stratified jitter, no nan_to_num/clamp."""
    gm, sd, ocfg = gpu_model("stress", 48)
    fused, gm.config.use_mfma = IMPLS[impl]
    gm.config.fused = fused
    gm.train()
    o, d = helpers.rays(12, 12, view=3)
    R = o.shape[0]
    g = torch.Generator().manual_seed(11)
    jit = [torch.rand(R, 1, generator=g) for _ in range(3)]
    cam = torch.randint(0, 8, (R, 1), generator=g)
    want = H.get_outputs(sd, o, d, cam, ocfg, training=True, jitter=jit)
    rb = gm.collider(bundle(o, d, cam))
    with torch.no_grad():
        if fused:
            got = gm._get_outputs_fused(rb, jitter=torch.cat(jit, dim=1).T.contiguous().to(DEV))
        else:
            got = gm._get_outputs_modular(rb, jitter=[j.to(DEV) for j in jit])
    gm.eval()
    check_outputs(got, want, f"train impl={impl}")
    assert len(got["weights_list"]) == 3 and len(got["ray_samples_list"]) == 3
    for i in range(3):
        assert_close(got["weights_list"][i], want["weights_list"][i], 3e-5, 1e-4, f"weights_list[{i}]")
        assert_close(got["ray_samples_list"][i].spacing_starts, want["ray_samples_list"][i].spacing_starts, 1e-5, 0,
                     f"spacing_starts[{i}]")


@pytest.mark.parametrize("family", ["lane_ray", "ray_per_wave"])
@pytest.mark.parametrize("impl", list(IMPLS))
@pytest.mark.parametrize("S", [48, 192])
def test_get_outputs_training_mode_with_one_draw_per_bin_edge(impl, family, S):
    """use_single_jitter=False through the train-mode forward: the modular samplers take [R,n+1] per level, the fused forward
    tn_render_inputs.jitter in the per-edge layout (tn_render_config.per_sample_jitter) — in both proposal kernels (one ray per
    wave: proposal_kernel; lane = ray: proposal_rays_kernel's sequential PDF walk)."""
    gm, sd, ocfg = gpu_model("stress", S, family=family, use_single_jitter=False)
    fused, gm.config.use_mfma = IMPLS[impl]
    if not fused and family == "ray_per_wave":
        pytest.skip("the modular path has one form")
    gm.config.fused = fused
    gm.train()
    o, d = helpers.rays(12, 12, view=3)
    R = o.shape[0]
    g = torch.Generator().manual_seed(13)
    jit = [torch.rand(R, n + 1, generator=g) for n in (*gm.config.num_proposal_samples_per_ray, S)]
    cam = torch.randint(0, 8, (R, 1), generator=g)
    want = H.get_outputs(sd, o, d, cam, ocfg, training=True, jitter=jit)
    rb = gm.collider(bundle(o, d, cam))
    with torch.no_grad():
        if fused:
            got = gm._get_outputs_fused(rb, jitter=[j.to(DEV) for j in jit])
            with pytest.raises(ValueError, match="draws"):
                gm._get_outputs_fused(rb, jitter=torch.rand(3, R, device=DEV))
        else:
            got = gm._get_outputs_modular(rb, jitter=[j.to(DEV) for j in jit])
    gm.eval()
    check_outputs(got, want, f"train per-edge impl={impl} {family}")
    for i in range(3):
        assert_close(got["weights_list"][i], want["weights_list"][i], 3e-5, 1e-4, f"weights_list[{i}]")
        assert_close(got["ray_samples_list"][i].spacing_starts, want["ray_samples_list"][i].spacing_starts, 1e-5, 0,
                     f"spacing_starts[{i}]")
    # not the single-draw result
    single = H.get_outputs(sd, o, d, cam, ocfg, training=True, jitter=[j[:, :1] for j in jit])
    for i in range(3):
        a, b = single["ray_samples_list"][i].spacing_starts, want["ray_samples_list"][i].spacing_starts
        assert (a - b).abs().max().item() > 1e-4, i


def test_full_size_tables_default_config():
    """The reference's default sizes (T=2^19 x16, 2^17 x5 x2): exercises 32-bit index paths of the real tables."""
    gm, sd, ocfg = gpu_model("stress", 64, small=False)
    o, d = helpers.rays(12, 12, view=5)
    want = H.get_outputs(sd, o, d, None, ocfg)
    with torch.no_grad():
        got = gm(bundle(o, d))
    check_outputs(got, want, "full-size")


@pytest.mark.parametrize("R", [0, 1, 3, 65, 1000])
def test_ragged_ray_counts(R):
    gm, sd, ocfg = gpu_model("stress", 48)
    o, d = helpers.rays(32, 32, view=1)
    o, d = o[:R].contiguous(), d[:R].contiguous()
    with torch.no_grad():
        got = gm(bundle(o, d))
    assert got["rgb"].shape == (R, 3) and got["thermal"].shape == (R, 1)
    if R:
        want = H.get_outputs(sd, o, d, None, ocfg)
        check_outputs(got, want, f"R={R}")


def test_camera_ray_bundle_chunking_matches_single_call():
    """get_outputs_for_camera_ray_bundle (row-major chunks) == oracle's chunked loop, incl. the per-chunk
    expected-depth clip."""
    gm, sd, ocfg = gpu_model("scene", 48)
    gm.config.eval_num_rays_per_chunk = 100
    o, d, _ = synthetic.orbit_camera_rays(15, 17, view=2)
    want = H.get_outputs_for_camera_ray_bundle(sd, o, d, ocfg, chunk=100)
    got = gm.get_outputs_for_camera_ray_bundle(RayBundle(origins=o, directions=d))
    assert got["rgb"].shape == (15, 17, 3) and got["thermal"].shape == (15, 17, 1)
    flat = lambda t: {k: v.reshape(-1, v.shape[-1]) for k, v in t.items()}
    flat_want = H.Outputs(flat(want))
    flat_want.median_ties = {k: flat(t) for k, t in want.median_ties.items()}
    check_outputs(flat(got), flat_want, "chunked")


def test_camera_ray_bundle_keeps_planes_already_on_the_bundle():
    """NS SceneCollider.forward leaves nears/fars alone when the bundle carries them: the engine path of
    get_outputs_for_camera_ray_bundle does too, and a collider edited after the first render is picked up."""
    gm, sd, ocfg = gpu_model("scene", 48)
    gm.config.eval_num_rays_per_chunk = 100
    o, d, _ = synthetic.orbit_camera_rays(9, 11, view=2)
    base = gm.get_outputs_for_camera_ray_bundle(RayBundle(origins=o, directions=d))
    same = gm.get_outputs_for_camera_ray_bundle(RayBundle(origins=o, directions=d, nears=torch.zeros(9, 11, 1),
                                                          fars=torch.full((9, 11, 1), 1000.0)))
    for k in ("rgb", "thermal", "depth"):
        assert torch.equal(base[k], same[k]), k
    moved = gm.get_outputs_for_camera_ray_bundle(RayBundle(origins=o, directions=d, nears=torch.full((9, 11, 1), 0.4),
                                                           fars=torch.full((9, 11, 1), 6.0)))
    assert float(moved["depth"].min()) >= 0.4 and float(moved["depth"].max()) <= 6.0
    assert not torch.equal(base["depth"], moved["depth"])
    gm.collider.far_plane = 6.0  # the engine's cached plane buffers follow the collider
    edited = gm.get_outputs_for_camera_ray_bundle(RayBundle(origins=o, directions=d))
    assert float(edited["depth"].max()) <= 6.0 and not torch.equal(base["depth"], edited["depth"])


@pytest.mark.parametrize("use_mfma", [True, False])
def test_fused_is_deterministic_and_matches_modular(use_mfma):
    gm, _, _ = gpu_model("stress", 64)
    gm.config.use_mfma = use_mfma
    o, d = helpers.rays(24, 24, view=4)
    with torch.no_grad():
        gm.config.fused = True
        a = gm(bundle(o, d))
        b = gm(bundle(o, d))
        gm.config.fused = False
        c = gm(bundle(o, d))
    for k in ("rgb", "thermal", "accumulation", "expected_depth"):
        assert torch.equal(a[k], b[k]), f"{k} not deterministic"
        assert_close(a[k], c[k].cpu(), 2e-5, 1e-4, f"fused vs modular {k}")


def test_mfma_path_is_actually_taken():
    """The prepared blob must exist for the default config, i.e. the MFMA kernel (not the VALU form) runs."""
    gm, _, _ = gpu_model("stress", 64)
    _, _, fld = gm._c_structs()
    assert fld.prepared, "tn_field_prepare produced no blob: the fused path would silently use the VALU kernel"


# --------------------------------------------------------------------------------------------------
# config surface: the switches the reference's config exposes must change BOTH sides the same way
# --------------------------------------------------------------------------------------------------
VARIANTS = {
    "aabb_no_contraction": dict(disable_scene_contraction=True),          # REF thermal_nerf_model.py:91-94
    "zero_appearance": dict(use_average_appearance_embedding=False),      # REF thermal_field.py:133-137
    "sh_unit_dirs": dict(sh_input="unit"),                                # SURVEY A.6 switch
    "short_far_plane": dict(far_plane=6.0, near_plane=0.2),
    "other_sample_counts": dict(num_proposal_samples_per_ray=(128, 64)),
}


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("impl", ["mfma", "f16x3", "bf16x6", "modular"])
def test_config_variants(variant, impl):
    S = 40 if variant == "other_sample_counts" else 48
    gm, sd, ocfg = gpu_model("stress", S, **VARIANTS[variant])
    if impl in ("f16x3", "bf16x6"):
        gm.config.fused, gm.config.use_mfma, gm.config.mlp_precision = True, True, impl
    else:
        gm.config.fused, gm.config.use_mfma = IMPLS[impl]
        gm.config.mlp_precision = "f32"
    o, d = helpers.rays(14, 14, view=6)
    if variant == "aabb_no_contraction":
        o = o * 0.6  # start inside the +-1 box so that part of every ray is inside and part outside
    want = H.get_outputs(sd, o, d, None, ocfg)
    with torch.no_grad():
        got = gm(bundle(o, d))
    check_outputs(got, want, f"{variant}/{impl}")


# --------------------------------------------------------------------------------------------------
# early ray termination (opt-in; the reference has none, so 0 must be exact and eps > 0 bounded by eps)
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["f32", "f16x3", "bf16x6"])
def test_early_ray_termination_is_bounded_by_eps(precision):
    gm, _, _ = gpu_model("scene", 64)
    gm.config.fused, gm.config.use_mfma, gm.config.mlp_precision = True, True, precision
    with torch.no_grad():
        gm.field.mlp_base.mlp.layers[1].bias[0] += 5.0  # dense fog: every ray saturates within a few samples
    o, d = helpers.rays(24, 24, view=2)
    eps = 1e-3
    with torch.no_grad():
        gm.config.early_termination_eps = 0.0
        exact = {k: v.clone() for k, v in gm(bundle(o, d)).items()}
        again = gm(bundle(o, d))
        for k in exact:
            assert torch.equal(exact[k], again[k]), k
        gm.config.early_termination_eps = eps
        early = gm(bundle(o, d))
    assert (exact["accumulation"] > 1 - eps).float().mean().item() > 0.9, "test scene does not saturate"
    for k in ("rgb", "thermal", "accumulation"):
        diff = (early[k] - exact[k]).abs().max().item()
        assert diff <= 2 * eps, f"{k}: {diff:.2e} > 2*eps"
    # the median crossing (cumulative weight 0.5) always precedes termination (eps <= 0.25)
    assert torch.equal(early["depth"], exact["depth"])
    # expected depth: the skipped tail carries < eps of weight spread over [last step, far]
    far = gm.config.far_plane
    assert (early["expected_depth"] - exact["expected_depth"]).abs().max().item() <= eps * far
    # proposal outputs are untouched by the switch
    for k in ("prop_depth_0", "prop_depth_1"):
        assert torch.equal(early[k], exact[k]), k


@pytest.mark.parametrize("precision", ["f32", "f16x3", "bf16x6"])
def test_checkpoint_load_invalidates_prepared_weights(tmp_path, precision):
    """Render, load a nerfstudio-layout checkpoint holding other weights into the SAME device model, render again:
    the prepared MFMA blobs must be rebuilt (outputs follow the oracle on the new weights)."""
    from thermo_nerf_amd import checkpoint as C

    gm, _, _ = gpu_model("init", 48)
    gm.config.fused, gm.config.use_mfma, gm.config.mlp_precision = True, True, precision
    src, sd_new, ocfg = helpers.build("stress", 48)
    o, d = helpers.rays(12, 12, view=5)
    with torch.no_grad():
        before = gm(bundle(o, d))["rgb"].clone()
        path = C.save_nerfstudio_checkpoint(src, tmp_path, 123)
        rep = C.load_nerfstudio_checkpoint(gm, path)
        assert rep.step == 123 and not rep.unexpected
        got = gm(bundle(o, d))
    assert not torch.equal(before, got["rgb"])
    check_outputs(got, H.get_outputs(sd_new, o, d, None, ocfg), f"after checkpoint load ({precision})")


def test_engine_chunks_on_two_streams_match_one_stream():
    """RayRenderEngine alternates chunks between two HIP streams (own workspace each); results must not depend on it."""
    from thermo_nerf_amd.engine import RayRenderEngine

    gm, sd, ocfg = gpu_model("scene", 64)
    o, d = helpers.rays(60, 50, view=1)
    o, d = o.to(DEV), d.to(DEV)
    one = RayRenderEngine(gm, chunk=700, streams=1).render(o, d)
    two = RayRenderEngine(gm, chunk=700, streams=2)
    a = two.render(o, d)
    b = two.render(o, d)  # second call reuses streams / workspaces
    torch.cuda.synchronize()
    for k in one:
        assert torch.equal(one[k], a[k]) and torch.equal(one[k], b[k]), k
    want = H.get_outputs(sd, o[:700].cpu(), d[:700].cpu(), None, ocfg)  # first chunk = one oracle call
    assert (a["rgb"][:700].cpu() - want["rgb"]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("form", ["fused", "modular", "fused_ray_per_wave"])
def test_same_proposal_network(form):
    """use_same_proposal_network [REF thermal_nerf_model.py:122-139]: one HashMLPDensityField serves both proposal levels."""
    gm, sd, ocfg = gpu_model("stress", 48, family="ray_per_wave" if form == "fused_ray_per_wave" else "lane_ray",
                             one_proposal_network=True)
    assert len(gm.proposal_networks) == 1 and "proposal_networks.1.mlp_base.encoder.hash_table" not in sd
    gm.config.fused = form != "modular"
    o, d = helpers.rays(13, 11, view=5)
    want = H.get_outputs(sd, o, d, None, ocfg)
    with torch.no_grad():
        got = gm(bundle(o, d))
    check_outputs(got, want, f"same proposal network, {form}")


@pytest.mark.parametrize("far", [1000.0, 6.0])
@pytest.mark.parametrize("form", ["fused", "fused_scalar", "modular", "fused_ray_per_wave", "fused_f16x3"])
def test_uniform_initial_sampler(form, far):
    """proposal_initial_sampler="uniform" [REF thermal_nerf_model.py:164-170]: every level's spacing -> distance map is
    linear.  far=6: a far plane a scene sampled uniformly would actually use; far=1000: the reference default plane."""
    gm, sd, ocfg = gpu_model("scene", 48, family="ray_per_wave" if form == "fused_ray_per_wave" else "lane_ray",
                             proposal_initial_sampler="uniform", far_plane=far)
    assert isinstance(gm.proposal_sampler.initial_sampler, UniformSampler)
    gm.config.fused = form != "modular"
    gm.config.use_mfma = form != "fused_scalar"
    gm.config.mlp_precision = "f16x3" if form == "fused_f16x3" else "f32"
    o, d = helpers.rays(13, 11, view=5)
    want = H.get_outputs(sd, o, d, None, ocfg)
    with torch.no_grad():
        got = gm(bundle(o, d))
    if form == "fused_f16x3":
        for k in ("rgb", "thermal"):
            assert (got[k].cpu() - want[k]).abs().max().item() <= 2e-4, k
    else:
        check_outputs(got, want, f"uniform initial sampler, {form}, far {far}")
    # and it is a different sampling from the piecewise default
    gp, sdp, ocfgp = gpu_model("scene", 48, far_plane=far)
    with torch.no_grad():
        other = gp(bundle(o, d))
    assert (other["depth"] - got["depth"]).abs().max().item() > 1e-3


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("family", ["lane_ray", "ray_per_wave"])
def test_dense_relayout_is_bit_identical(precision, family):
    """dense_grid_budget_mb / field_dense_grid_budget_mb (on by default): the dense copies of the leading levels hold the
    same values as the hashed tables, and the kernels do the same arithmetic on them — every output is identical to the
    hashed-table render, on the full-size grids (where the lane = ray field kernels read 6 levels densely)."""
    gm, sd, ocfg = gpu_model("scene", 64, small=False, family=family)
    gm.config.mlp_precision = precision
    assert gm.config.dense_grid_budget_mb > 0 and gm.config.field_dense_grid_budget_mb > 0
    assert gm.field.dense_budget_bytes > 0 and gm.proposal_networks[0].dense_budget_bytes > 0
    o, d = helpers.rays(24, 24, view=2)
    with torch.no_grad():
        dense = {k: v.clone() for k, v in gm(bundle(o, d)).items()}
        assert gm.field.c_struct(prepare=True).grid.num_dense_levels >= 6
        assert gm.proposal_networks[0].c_struct().grid.num_dense_levels == 5
        assert gm.proposal_networks[1].c_struct().grid.num_dense_levels == 4
        budgets = (gm.field.dense_budget_bytes, [n.dense_budget_bytes for n in gm.proposal_networks])
        gm.field.dense_budget_bytes = 0
        for n in gm.proposal_networks:
            n.dense_budget_bytes = 0
        gm.invalidate_prepared()
        hashed = gm(bundle(o, d))
        assert gm.field.c_struct(prepare=True).grid.num_dense_levels == 0
        gm.field.dense_budget_bytes = budgets[0]
        for n, b in zip(gm.proposal_networks, budgets[1]):
            n.dense_budget_bytes = b
        gm.invalidate_prepared()
    for k in ("rgb", "thermal", "accumulation", "depth", "expected_depth", "prop_depth_0", "prop_depth_1"):
        assert torch.equal(dense[k], hashed[k]), k
    if precision == "f32":
        check_outputs(dense, H.get_outputs(sd, o, d, None, ocfg), f"dense default, {family}")


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_small_calls_take_the_ray_per_wave_kernels(precision):
    """Automatic dispatch: below ~60-80 k rays the one-ray-per-wave kernels run (a 64-ray tile marches serially, so the
    lane = ray kernels have a ~2.6 ms floor).  Same oracle tolerances; with f16x3 a small call is served in exact fp32."""
    import time

    gm, sd, ocfg = gpu_model("scene", 64)
    gm.config.fused, gm.config.use_mfma, gm.config.mlp_precision = True, True, precision
    o, d = helpers.rays(32, 32, view=4)
    want = H.get_outputs(sd, o, d, None, ocfg)
    with torch.no_grad():
        forced = {k: v.clone() for k, v in gm(bundle(o, d)).items()}
        gm.config.kernel_family = "auto"
        auto = gm(bundle(o, d))
        check_outputs(auto, want, f"auto dispatch {precision}")
        for k in ("rgb", "thermal"):
            assert (auto[k] - forced[k]).abs().max().item() <= 2e-5, k
        # and it is the faster choice at this size
        def timed():
            best = float("inf")
            for _ in range(4):  # best of four: the first repetition may carry one-time costs
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(5):
                    gm(bundle(o, d))
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t) / 5)
            return best
        t_auto = timed()
        gm.config.kernel_family = "lane_ray"
        t_forced = timed()
    assert t_auto < t_forced, (t_auto, t_forced)


@pytest.mark.parametrize("P,S", [((256, 96), 1), ((256, 96), 2), ((256, 96), 256), ((1, 1), 3), ((2, 256), 64), ((256, 256), 33),
                                 ((7, 5), 65), ((1024, 1024), 1024)])
@pytest.mark.parametrize("family", ["lane_ray", "ray_per_wave"])
def test_extreme_sample_counts(P, S, family):
    """Smallest and largest per-level sample counts the kernels accept, for both kernel families."""
    gm, sd, ocfg = gpu_model("stress", S, family=family, num_proposal_samples_per_ray=P)
    gm.config.fused, gm.config.use_mfma, gm.config.mlp_precision = True, True, "f32"
    o, d = helpers.rays(11, 13, view=2)
    want = H.get_outputs(sd, o, d, None, ocfg)
    with torch.no_grad():
        got = gm(bundle(o, d))
    check_outputs(got, want, f"P={P} S={S} {family}")


def test_sample_counts_beyond_the_kernel_limits_are_refused():
    gm, sd, ocfg = gpu_model("stress", 300)  # above 256: still inside the limit of 1024 per level
    o, d = helpers.rays(6, 6)
    with torch.no_grad():
        check_outputs(gm(bundle(o, d)), H.get_outputs(sd, o, d, None, ocfg), "S=300")
    big, _, _ = gpu_model("stress", 1025)
    with pytest.raises(RuntimeError, match="TN_ERR_SHAPE"):
        with torch.no_grad():
            big(bundle(o, d))


@pytest.mark.parametrize("family", ["lane_ray", "ray_per_wave"])
def test_degenerate_rays(family):
    """Axis-aligned directions, origins on exact grid points (ceil == floor corners), origins far outside the scene,
    rays that never enter the unit box."""
    gm, sd, ocfg = gpu_model("stress", 64, family=family)
    gm.config.fused, gm.config.use_mfma, gm.config.mlp_precision = True, True, "f32"
    axes = torch.tensor([[1.0, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])
    o = torch.cat([torch.zeros(6, 3), torch.full((6, 3), 0.5), torch.tensor([[40.0, -25.0, 3.0]]).repeat(6, 1),
                   torch.tensor([[0.25, 0.125, -0.0625]]).repeat(6, 1), torch.tensor([[5.0, 5.0, 5.0]]).repeat(6, 1)])
    d = axes.repeat(5, 1)
    diag = torch.nn.functional.normalize(torch.tensor([[1.0, 1.0, 1.0], [-1.0, 1.0, -1.0]]), dim=1)
    o = torch.cat([o, torch.tensor([[-3.0, -3.0, -3.0], [2.0, -2.0, 2.0]])])
    d = torch.cat([d, diag])
    want = H.get_outputs(sd, o, d, None, ocfg)
    with torch.no_grad():
        got = gm(bundle(o, d))
    for k in ("rgb", "thermal", "accumulation"):
        assert torch.isfinite(got[k]).all(), k
    check_outputs(got, want, f"degenerate rays {family}")


@pytest.mark.parametrize("S", [64, 192])  # BASELINE config 2, and the metric's own frame (800x800 at 192 samples per ray)
def test_full_frame_properties(S):
    """BASELINE's frames at full size (800x800 = 640 000 rays, full-size tables; S = 64: config 2, S = 192: the configuration
    the headline metric is quoted on): size-independent properties of the path — idempotence, independence from chunking /
    stream scheduling, equivariance under a permutation of the rays — plus the oracle on a strided sample of the same frame."""
    from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic
    from thermo_nerf_amd.engine import RayRenderEngine

    # config.kernel_family stays "auto": every call picks its kernel family by size, as in production
    cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=S)
    model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
    synthetic.fill_model_(model, "scene")
    sd = synthetic.model_state_dict_cpu(model)
    model.to(DEV).eval()
    o, d, _ = synthetic.orbit_camera_rays(800, 800, view=1)
    o, d = o.reshape(-1, 3).contiguous().to(DEV), d.reshape(-1, 3).contiguous().to(DEV)
    n = o.shape[0]
    whole = RayRenderEngine(model, chunk=n)
    a = {k: v.clone() for k, v in whole.render(o, d).items()}
    b = whole.render(o, d)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k}: not idempotent"
    # chunked on two streams: per-ray results do not depend on how the frame is cut (expected depth clips per chunk)
    c = RayRenderEngine(model, chunk=65536).render(o, d)
    torch.cuda.synchronize()
    for k in ("rgb", "thermal", "accumulation", "depth", "prop_depth_0", "prop_depth_1"):
        assert torch.equal(a[k], c[k]), f"{k}: depends on chunking"
    # permutation equivariance: rays are independent, whatever their neighbours in the 64-ray tile
    perm = torch.randperm(n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    p = whole.render(o[perm].contiguous(), d[perm].contiguous())
    torch.cuda.synchronize()
    for k in ("rgb", "thermal", "accumulation", "depth"):
        assert torch.equal(p[k], a[k][perm]), f"{k}: depends on the ray order"
    # the oracle on a strided sample
    idx = torch.linspace(0, n - 1, 1500).long()
    want = H.get_outputs(sd, o[idx.to(DEV)].cpu(), d[idx.to(DEV)].cpu(), None, helpers.oracle_config(cfg))
    assert (a["rgb"][idx.to(DEV)].cpu() - want["rgb"]).abs().mean().item() <= 1e-4
    assert (a["thermal"][idx.to(DEV)].cpu() - want["thermal"]).abs().mean().item() <= 1e-4
    assert (a["accumulation"][idx.to(DEV)].cpu() - want["accumulation"]).abs().max().item() <= 2e-5

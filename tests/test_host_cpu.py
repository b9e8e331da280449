"""CPU: host logic, the C-ABI library's exports, and the no-fallback guarantees (no compute calls)."""
import ctypes
import os
import re

import pytest
import torch

import thermo_nerf_amd as tna
from thermo_nerf_amd import _hip
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "thermonerf_hip.h")).read()
    declared = set(re.findall(r"\b(tn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no entry points parsed from the header"
    lib = ctypes.CDLL(_hip.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/thermonerf_hip.h but not exported"
    assert declared == set(_hip.SIGNATURES), "ctypes SIGNATURES out of sync with the header"
    assert b"gfx950" in _hip.load().tn_version()


def test_library_reads_no_environment_variables():
    """The header promises no mutable global state: the kernel family is chosen through tn_render_config.kernel_family,
    never through the process environment (no getenv import, no TN_* switch names in the binary)."""
    import subprocess

    syms = subprocess.run(["nm", "-D", "--undefined-only", _hip.lib_path()], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    blob = open(_hip.lib_path(), "rb").read()
    for name in (b"TN_FORCE_LANE_RAY", b"TN_FORCE_RAY_PER_WAVE", b"TN_PROPOSAL_PER_RAY"):
        assert name not in blob, name


def test_struct_sizes_match_header_layout():
    # sizes computed from the header's field order (LP64): catches drift between header and ctypes mirror
    assert ctypes.sizeof(_hip.tn_hashgrid) == 8 + 64 + 4 + 4 + 8 + 128 + 64 + 4 + 4
    assert ctypes.sizeof(_hip.tn_linear) == 24
    assert ctypes.sizeof(_hip.tn_space) == 32
    assert ctypes.sizeof(_hip.tn_density_field) == 288 + 24 + 24 + 32 + 8
    assert ctypes.sizeof(_hip.tn_render_outputs) == 7 * 8 + 9 * 8
    assert ctypes.sizeof(_hip.tn_adam_tensor) == 4 * 8 + 8 + 7 * 4 + 4  # (+ tail padding to 8)
    # the step calls' argument blocks (round 6): sizes as gcc lays the header's structs out (g++ -I. on include/thermonerf_hip.h)
    assert ctypes.sizeof(_hip.tn_train_step) == 416 and ctypes.sizeof(_hip.tn_train_step_bwd_args) == 408


def test_model_surface_and_state_dict_names():
    model, sd, _ = helpers.build("init", 48)
    for k in ("field.mlp_base.encoder.hash_table", "field.mlp_base.encoder.scalings",
              "field.mlp_base.mlp.layers.0.weight", "field.mlp_base.mlp.layers.1.bias",
              "field.embedding_appearance.embedding.weight", "field.mlp_head.layers.2.weight",
              "field.mlp_thermal.layers.1.weight", "field.field_head_thermal.net.weight",
              "field.field_head_thermal.net.bias", "proposal_networks.0.mlp_base.encoder.hash_table",
              "proposal_networks.1.mlp_base.mlp.layers.1.weight", "camera_optimizer.pose_adjustment"):
        assert k in sd, k
    assert sd["field.mlp_head.layers.0.weight"].shape == (64, 63)
    assert sd["field.mlp_thermal.layers.0.weight"].shape == (64, 15)
    assert sd["field.field_head_thermal.net.weight"].shape == (1, 64)
    assert set(model.get_param_groups()) == {"proposal_networks", "fields", "camera_opt"}
    assert tna.RenderedImageModality.THERMAL.value == "thermal"
    assert tna.FieldHeadNamesT.THERMAL.value == "thermal"


def test_thermal_metadata_required():
    with pytest.raises(ValueError, match="Thermal images not found"):
        tna.ThermalNerfModel(tna.ThermalNerfModelConfig(**helpers.SMALL), metadata={}, scene_box=tna.SceneBox.unit(),
                             num_train_data=2)


def test_no_cpu_fallback():
    """CPU tensors must raise, never compute: the product has no PyTorch/CPU arithmetic path."""
    model, _, _ = helpers.build("init", 48)
    o, d = helpers.rays(4, 4)
    rb = tna.RayBundle(origins=o, directions=d, camera_indices=torch.zeros(16, 1, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(rb)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tna.ThermalRenderer()(torch.rand(4, 8, 1), torch.rand(4, 8, 1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.proposal_networks[0].density_fn(torch.rand(5, 3))


def test_predict_normals_ends_where_the_reference_ends(golden_dir):
    """G9 (tests/golden/predict_normals.json: the reference's own get_outputs executed with config.predict_normals=True,
    tools/make_golden_g9.py): its field override never evaluates nerfstudio's predicted-normals head — the field returns RGB /
    THERMAL / DENSITY / NORMALS — and the model ends in KeyError(FieldHeadNames.PRED_NORMALS) right after rendering the analytic
    normals, in every forward.  The product raises the same exception with the same key, in eval and in training, before any
    device work; it does not invent outputs the reference cannot produce."""
    import json

    gold = json.load(open(os.path.join(golden_dir, "predict_normals.json")))
    assert gold["field_output_keys"] == ["RGB", "THERMAL", "DENSITY", "NORMALS"]
    exc = gold["exception"]
    assert exc["type"] == "KeyError" and exc["raised_after"][-1] == "renderer_normals"
    model, _, _ = helpers.build("init", 8, predict_normals=True)
    assert model.config.predict_normals is True
    o, d = helpers.rays(2, 2)
    rb = tna.RayBundle(origins=o, directions=d, camera_indices=torch.zeros(4, 1, dtype=torch.long))
    for training in (False, True):
        model.train(training)
        with pytest.raises(KeyError) as err:
            model(rb)
        key = err.value.args[0]
        assert key is tna.fields.FieldHeadNames.PRED_NORMALS and key.name == exc["key_name"] and key.value == exc["key_value"]
    model.eval()


def test_transient_embedding_flag_builds_the_references_modules_and_changes_nothing(golden_dir):
    """G10 (tests/golden/transient_embedding.json, tools/make_golden_g9.py: the reference executed with use_transient_embedding on
    and off on the same weights): its field returns two more entries in training, its model reads neither — model outputs are
    identical.  The product builds nerfstudio's transient modules (state-dict names) and leaves the hot path as it is."""
    import json

    gold = json.load(open(os.path.join(golden_dir, "transient_embedding.json")))
    assert gold["model_outputs_identical_with_and_without_the_flag"] is True
    assert gold["field_output_keys"]["flag_1_train"][:2] == ["TRANSIENT_RGB", "TRANSIENT_DENSITY"]
    assert gold["field_output_keys"]["flag_1_eval"] == gold["field_output_keys"]["flag_0_eval"] == ["RGB", "THERMAL", "DENSITY"]
    assert "transient_rgb" not in gold["model_output_keys"]
    on, sd_on, _ = helpers.build("init", 8, use_transient_embedding=True)
    off, sd_off, _ = helpers.build("init", 8)
    extra = sorted(set(sd_on) - set(sd_off))
    assert extra == ["field.embedding_transient.embedding.weight", "field.field_head_transient_density.net.bias",
                     "field.field_head_transient_density.net.weight", "field.field_head_transient_rgb.net.bias",
                     "field.field_head_transient_rgb.net.weight", "field.field_head_transient_uncertainty.net.bias",
                     "field.field_head_transient_uncertainty.net.weight", "field.mlp_transient.layers.0.bias",
                     "field.mlp_transient.layers.0.weight", "field.mlp_transient.layers.1.bias", "field.mlp_transient.layers.1.weight"]
    assert sd_on["field.embedding_transient.embedding.weight"].shape == (8, 16) and sd_on["field.mlp_transient.layers.0.weight"].shape == (64, 31)
    assert not on.field.staged and on._fusable()  # the fused hot path is untouched


def test_packed_samples_rejected_like_reference():
    r = tna.ThermalRenderer()
    with pytest.raises(NotImplementedError):
        r(torch.rand(8, 1), torch.rand(8, 1), ray_indices=torch.zeros(8, dtype=torch.long), num_rays=2)


def test_collider_eval_resets_near_plane():
    model, _, _ = helpers.build("init", 48)
    o, d = helpers.rays(2, 2)
    rb = model.collider(tna.RayBundle(origins=o, directions=d))
    assert float(rb.nears.max()) == 0.0 and float(rb.fars.min()) == 1000.0
    model.collider.train()
    rb = model.collider(tna.RayBundle(origins=o, directions=d))
    assert abs(float(rb.nears.max()) - 0.05) < 1e-9
    model.collider.eval()


def test_anneal_schedule():
    model, _, _ = helpers.build("init", 48)
    model.set_step(0)
    assert model.proposal_sampler._anneal == 0.0
    model.set_step(500)
    assert abs(model.proposal_sampler._anneal - 10 * 0.5 / (9 * 0.5 + 1)) < 1e-12
    model.set_step(5000)
    assert model.proposal_sampler._anneal == 1.0


def test_oracle_uniform_initial_sampler_is_linear_in_distance():
    """NS UniformSampler [REF thermal_nerf_model.py:164-170 -> proposal_initial_sampler="uniform"]: identity spacing
    functions, so bin edges are a linspace between near and far, and PDF resampling keeps that map."""
    import torch

    from oracle import hotpath as H

    nears, fars = torch.full((3, 1), 0.05), torch.full((3, 1), 6.0)
    s = H.sample_initial(nears, fars, 8, None, uniform=True)
    edges = torch.cat([s.starts[..., 0], s.ends[:, -1:, 0]], -1)
    assert torch.allclose(edges, torch.linspace(0.05, 6.0, 9).expand(3, -1), atol=1e-6)
    p = H.sample_pdf(s, torch.zeros(3, 8, 1), 4, None)  # zero weights -> the histogram padding resamples uniformly
    assert p.uniform
    mids = (p.starts + p.ends)[..., 0] / 2
    assert torch.all(mids[:, 1:] > mids[:, :-1]) and float(p.ends.max()) <= 6.0 + 1e-5
    # the piecewise default is NOT linear: half of the spacing range covers [0, 1]
    q = H.sample_initial(torch.zeros(1, 1), torch.full((1, 1), 1000.0), 8, None)
    assert abs(float(q.ends[0, 3, 0]) - 1.0) < 2e-3


def test_kernel_form_query_and_the_engine_launch_plan():
    """tn_render_kernel_form is the ONE place the call-size thresholds live (no compute call: runs without a GPU), and
    RayRenderEngine.frame_launch_rays cuts a frame into equal runs of whole chunks under a TOTAL workspace budget."""
    from thermo_nerf_amd.engine import RayRenderEngine

    lib = _hip.load()
    rc = _hip.tn_render_config()
    rc.num_proposal_samples[0], rc.num_proposal_samples[1], rc.num_nerf_samples = 256, 96, 48
    fld = _hip.tn_thermal_field()
    blob = (ctypes.c_float * 4)()
    mfma_blob = (ctypes.c_float * 4)()
    for fam in (0, 1, 2):
        rc.kernel_family = fam
        for n in (1, 8191, 8192, 24575, 24576, 40959, 40960, 57343, 57344, 65535, 65536, 1 << 21):
            want_prop = fam or (1 if n >= 24576 else 2)
            want_whole_tiles = fam or (1 if n >= 57344 else 2)   # the exact-fp32 kernel where it cannot march in segments
            want_segments = fam or (1 if n >= 8192 else 2)       # ... and where it can
            want_split_precision = fam or (1 if n >= 40960 else 2)
            assert lib.tn_render_kernel_form(None, rc, n, 0) == want_prop
            assert lib.tn_render_kernel_form(None, rc, n, 1) == want_whole_tiles
            fld.prepared, fld.prepared_bf16x6 = None, None
            assert lib.tn_render_kernel_form(fld, rc, n, 1) == want_whole_tiles
            fld.prepared = ctypes.addressof(mfma_blob)
            assert lib.tn_render_kernel_form(fld, rc, n, 1) == want_segments
            rc.sample_split = 1  # never split: whole tiles
            assert lib.tn_render_kernel_form(fld, rc, n, 1) == want_whole_tiles
            rc.sample_split, rc.early_stop_transmittance = 0, 1e-3  # early termination has no segmented form
            assert lib.tn_render_kernel_form(fld, rc, n, 1) == want_whole_tiles
            rc.early_stop_transmittance = 0.0
            fld.prepared_bf16x6 = ctypes.addressof(blob)
            assert lib.tn_render_kernel_form(fld, rc, n, 1) == want_split_precision
            rc.training = 1  # (the split-precision and segmented kernels are eval-only)
            assert lib.tn_render_kernel_form(fld, rc, n, 1) == want_whole_tiles
            assert lib.tn_render_kernel_form(fld, rc, n, 0) == (fam or (1 if n >= 65536 else 2))
            rc.training = 0
    assert lib.tn_render_kernel_form(None, None, 100, 0) == 0
    model, _, _ = helpers.build("init", 48)
    model.eval()
    eng = RayRenderEngine(model, chunk=1 << 16)  # default budget 2 GiB in total, two stream slots
    per_ray = 4 * (48 + 1 + 256 + 97)
    assert lib.tn_render_workspace_bytes(eng.rc, 1 << 16) // (1 << 16) == per_ray
    assert eng.frame_launch_rays(800 * 800) == 10 << 16  # fits: one launch pair
    assert eng.frame_launch_rays(1080 * 1920) == 8 << 16  # 32 chunks: 4 equal launches of 8 (not 20 + 12)
    assert [j - i for i, j in eng._launch_pieces(0, 1080 * 1920, 1080 * 1920)] == [8 << 16] * 3 + [1080 * 1920 - (24 << 16)]
    assert eng._launch_pieces(500000, 600000, 1080 * 1920) == [(500000, 8 << 16), (8 << 16, 600000)]
    assert eng._launch_pieces(7, 7, 100) == []
    peak = 2 * lib.tn_render_workspace_bytes(eng.rc, 8 << 16)
    assert peak <= 2 << 30
    one = RayRenderEngine(model, chunk=800 * 800, max_workspace_bytes=1 << 20)  # a chunk larger than the budget: still one chunk per launch
    assert one.frame_launch_rays(800 * 800) == 800 * 800 and one.frame_launch_rays(3 * 800 * 800) == 800 * 800

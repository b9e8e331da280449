"""nerfstudio checkpoint layout: `_model.`-prefixed pipeline state, foreign keys, DDP prefix, tcnn
refusal, shape errors, latest-step selection [REF thermo_nerf/render/renderer.py:94-113]."""
import copy

import pytest
import torch

from tests import helpers
from thermo_nerf_amd import checkpoint as C


def fresh(kind="init"):
    m, _, _ = helpers.build(kind, 48)
    return copy.deepcopy(m)


def pipeline_state(model, ddp=False):
    pre = ("module." if ddp else "") + "_model."
    st = {pre + k: v.clone() for k, v in model.state_dict().items()}
    # what a real nerfstudio pipeline also carries
    st[("module." if ddp else "") + "_model.lpips.net.slice1.0.weight"] = torch.zeros(4, 3, 3, 3)
    st[("module." if ddp else "") + "datamanager.train_ray_generator.image_coords"] = torch.zeros(2, 2, 2)
    return st


@pytest.mark.parametrize("ddp", [False, True])
def test_round_trip_through_a_nerfstudio_style_file(tmp_path, ddp):
    src, dst = fresh("stress"), fresh("init")
    torch.save({"step": 2999, "pipeline": pipeline_state(src, ddp), "optimizers": {}, "scalers": {}},
               tmp_path / "step-000002999.ckpt")
    torch.save({"step": 999, "pipeline": pipeline_state(dst, ddp)}, tmp_path / "step-000000999.ckpt")
    assert C.latest_checkpoint(tmp_path).name == "step-000002999.ckpt"
    rep = C.load_nerfstudio_checkpoint(dst, tmp_path)  # directory -> last step, like the reference harness
    assert rep.step == 2999 and not rep.missing and not rep.unexpected
    assert any("lpips" in k for k in rep.ignored) and any(k.startswith("datamanager.") for k in rep.ignored)
    a, b = src.state_dict(), dst.state_dict()
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_save_writes_the_keys_the_reference_reads(tmp_path):
    m = fresh("stress")
    p = C.save_nerfstudio_checkpoint(m, tmp_path / "nerfstudio_models", 30000)
    assert p.name == "step-000030000.ckpt"
    st = torch.load(p, weights_only=True)
    assert st["step"] == 30000
    for k in ("_model.field.mlp_base.encoder.hash_table", "_model.field.mlp_thermal.layers.1.weight",
              "_model.field.field_head_thermal.net.bias", "_model.proposal_networks.1.mlp_base.mlp.layers.0.weight",
              "_model.field.embedding_appearance.embedding.weight", "_model.camera_optimizer.pose_adjustment"):
        assert k in st["pipeline"], k
    other = fresh("init")
    C.load_nerfstudio_checkpoint(other, p)
    assert torch.equal(other.field.mlp_thermal.layers[1].weight, m.field.mlp_thermal.layers[1].weight)


def test_sequential_alias_and_bare_state_dict():
    src, dst = fresh("stress"), fresh("init")
    st = {}
    for k, v in src.state_dict().items():
        k = k.replace(".mlp_base.encoder.", ".mlp_base.model.0.").replace(".mlp_base.mlp.", ".mlp_base.model.1.")
        st[k] = v.clone()
    rep = C.load_nerfstudio_checkpoint(dst, st)  # no `pipeline` wrapper, no `_model.` prefix
    assert rep.step is None and not rep.unexpected
    assert torch.equal(dst.field.mlp_base.encoder.hash_table, src.field.mlp_base.encoder.hash_table)
    assert torch.equal(dst.proposal_networks[0].mlp_base.mlp.layers[1].bias, src.proposal_networks[0].mlp_base.mlp.layers[1].bias)


def test_missing_buffers_are_tolerated_missing_parameters_are_not():
    src, dst = fresh("stress"), fresh("init")
    st = pipeline_state(src)
    for k in list(st):
        if k.endswith(".scalings") or k.endswith(".max_res"):
            del st[k]
    rep = C.load_nerfstudio_checkpoint(dst, {"step": 1, "pipeline": st})
    assert any(k.endswith("scalings") for k in rep.missing)
    del st["_model.field.mlp_thermal.layers.0.bias"]
    with pytest.raises(KeyError, match="mlp_thermal.layers.0.bias"):
        C.load_nerfstudio_checkpoint(fresh("init"), {"step": 1, "pipeline": st})
    rep = C.load_nerfstudio_checkpoint(fresh("init"), {"step": 1, "pipeline": st}, strict=False)
    assert "field.mlp_thermal.layers.0.bias" in rep.missing


def test_unexpected_and_misshapen_entries_raise():
    src = fresh("stress")
    st = pipeline_state(src)
    st["_model.field.mlp_extra.layers.0.weight"] = torch.zeros(2, 2)
    with pytest.raises(KeyError, match="mlp_extra"):
        C.load_nerfstudio_checkpoint(fresh("init"), {"pipeline": st})
    st = pipeline_state(src)
    st["_model.field.embedding_appearance.embedding.weight"] = torch.zeros(5, 32)  # trained with 5 images, model has 8
    with pytest.raises(ValueError, match="num_train_data"):
        C.load_nerfstudio_checkpoint(fresh("init"), {"pipeline": st})


def test_tcnn_packed_parameters_are_refused():
    st = pipeline_state(fresh("init"))
    st["_model.field.mlp_base.tcnn_encoding.params"] = torch.zeros(100)
    with pytest.raises(NotImplementedError, match="tiny-cuda-nn"):
        C.load_nerfstudio_checkpoint(fresh("init"), {"pipeline": st})


def test_no_checkpoint_in_directory(tmp_path):
    with pytest.raises(FileNotFoundError):
        C.latest_checkpoint(tmp_path)

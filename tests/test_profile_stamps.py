"""CPU: the rocprofv3 evidence bench.py quotes is from the tree that is benched.  profiles/pmc_traffic.json
(`roofline.traffic`) and profiles/train_kernels.json (the training rooflines' phases and traffic) carry the commit they were measured
at, and so do the round's kernel-trace / PMC summaries; none of the kernel sources (thermo_nerf_amd/csrc, include/) may have changed
between that commit and HEAD.  Needs the git history (skipped on a copy without .git, e.g. the GPU box)."""
import glob
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_PATHS = ["thermo_nerf_amd/csrc", "include"]
ROUND = "round6"


def _git(*args):
    return subprocess.run(["git", "-C", ROOT, *args], capture_output=True, text=True)


def _kernels_unchanged_since(commit: str) -> bool:
    if _git("cat-file", "-e", commit + "^{commit}").returncode != 0:
        return False
    return _git("diff", "--quiet", commit, "HEAD", "--", *KERNEL_PATHS).returncode == 0


@pytest.fixture(scope="module")
def have_git():
    if not os.path.isdir(os.path.join(ROOT, ".git")) or _git("rev-parse", "HEAD").returncode != 0:
        pytest.skip("no git history here")


def test_bench_evidence_is_stamped_with_a_commit_whose_kernels_are_heads(have_git):
    stamps = {}
    traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key in ("field_render@S192", "proposal_sample@S192", "field_render@S64", "field_render@S192_bf16x6"):
        assert key in traffic, key
        stamps["pmc_traffic.json:" + key] = traffic[key]["commit"]
        assert ROUND in traffic[key]["source"], (key, traffic[key]["source"])
    kernels = json.load(open(os.path.join(ROOT, "profiles", "train_kernels.json")))
    for key in ("S48", "S192"):
        stamps["train_kernels.json:" + key] = kernels[key]["commit"]
        assert any(ROUND in src for src in kernels[key]["sources"]), kernels[key]["sources"]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", ROUND + "_kernel_trace_*.txt")) +
                   glob.glob(os.path.join(ROOT, "profiles", ROUND + "_pmc_*.txt")))
    assert len(files) >= 6, files  # S=192 / S=64 f32 and S=192 bf16x6: a trace and a PMC summary each (+ the training step's)
    for path in files:
        m = re.search(r"measured at commit ([0-9a-f]{40})", open(path).read(2000))
        assert m, f"{path}: no commit stamp"
        stamps[os.path.basename(path)] = m.group(1)
    for path in glob.glob(os.path.join(ROOT, "profiles", ROUND + "_train_account_S*.json")):
        stamps[os.path.basename(path)] = json.load(open(path))["commit"]
    bad = {k: v for k, v in stamps.items() if not _kernels_unchanged_since(v)}
    assert not bad, f"measured at a tree whose kernels differ from HEAD's: {bad}"

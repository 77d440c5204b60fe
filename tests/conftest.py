"""pytest configuration: the ``gpu`` marker and shared fixtures.

``-m "not gpu"`` runs everywhere (oracle vs goldens, host logic, C-ABI symbol
checks); ``-m gpu`` needs a real MI355X and goes through the C-ABI library.
Nothing here reads /root/reference: that tree does not exist on the GPU box.
"""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- parity records: the margin counts of the end-to-end tests (how many detections, how many survivor differences, how
# many of them on a numerical margin, how many unexplained) are part of the evidence, not just pass/fail: tests add them
# through the `record_parity` fixture, the terminal summary prints them (so they land in every pytest log, -s or not) and
# they are written to gpurun_out/parity_counts.json (copied to profiles/ by tools/gpu_r4.sh).
PARITY_RECORDS = {}


@pytest.fixture
def record_parity():
    def add(key, **counts):
        PARITY_RECORDS[key] = counts
    return add


def pytest_terminal_summary(terminalreporter):
    if not PARITY_RECORDS:
        return
    import json
    terminalreporter.section("parity records (margin counts)")
    for k, v in PARITY_RECORDS.items():
        terminalreporter.write_line("%s: %s" % (k, json.dumps(v, sort_keys=True)))
    try:
        out = os.path.join(REPO, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_counts.json")
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        old.update(PARITY_RECORDS)
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden_real():
    return _npz("golden_real.npz")


@pytest.fixture(scope="session")
def golden_rand():
    return _npz("golden_rand.npz")


@pytest.fixture(scope="session")
def golden_floor():
    """the reference's fp32 forward + decode against its own float64 evaluation on the bench workload (make_golden.py floor)"""
    return _npz("golden_floor.npz")


def floor_inputs(z):
    """the workload golden_floor.npz was made on: (state_dict, images) - CPU generators, reproducible on every box"""
    import torch
    import yolo_fastestv2_amd as yfv2
    sd = yfv2.random_state_dict(int(z["weight_seed"]))
    x = torch.rand(int(z["images"]), 3, 352, 352, generator=torch.Generator().manual_seed(int(z["image_seed"])))
    assert np.array_equal(x.flatten()[::1000003].numpy(), z["x_probe"]), "torch.rand(seed) differs from the generating machine's"
    return sd, x


@pytest.fixture(scope="session")
def golden_kat():
    return _npz("golden_kat.npz")


@pytest.fixture(scope="session")
def golden_stress():
    return _npz("golden_nms_stress.npz")


@pytest.fixture(scope="session")
def cfg():
    z = _npz("cfg_coco.npz")
    return {"anchors": [float(a) for a in z["anchors"]], "classes": int(z["classes"]),
            "anchor_num": int(z["anchor_num"]), "width": int(z["width"]), "height": int(z["height"])}


@pytest.fixture(scope="session")
def images_u8():
    return _npz("images_u8.npz")["images"]


@pytest.fixture(scope="session")
def coco_weights():
    from oracle import yfv2_oracle
    return yfv2_oracle.load_weights(os.path.join(GOLDEN, "weights_coco.npz"))


@pytest.fixture(scope="session")
def golden_stats():
    """NMS rows of the stress set + synthetic targets + the REFERENCE's get_batch_statistics flags (make_golden.py)."""
    return dict(np.load(os.path.join(GOLDEN, "golden_stats.npz")))


def unpack_ragged(z, prefix):
    """inverse of make_golden.pack_ragged -> (list of rows, list of idx)"""
    cnt = z[prefix + "_count"]
    off = np.concatenate(([0], np.cumsum(cnt)))
    rows = [z[prefix + "_rows"][off[i]:off[i + 1]] for i in range(len(cnt))]
    idx = [z[prefix + "_idx"][off[i]:off[i + 1]] for i in range(len(cnt))]
    return rows, idx

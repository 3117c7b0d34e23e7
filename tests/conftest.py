import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

LIMITS = {"interface": (-2, 20, -2, 4), "fisheye": (-1.5, 1.5, -1.5, 1.5), "vert_heterogeneous": (-2, 5, -2.5, 1),
          "anisotropy": (-2, 5, -2.5, 1)}
# traj fixture prefix -> scenario
TRAJ_SCEN = {"interface": "interface", "fisheye": "fisheye", "vert": "vert_heterogeneous", "aniso": "anisotropy"}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def traj_fixtures():
    out = []
    for fn in sorted(os.listdir(GOLDEN)):
        if fn.startswith("traj_"):
            name = fn[5:-4]
            out.append((name, TRAJ_SCEN[name.split("_")[0]], int(name.split("op")[1].split("_")[0])))
    return out


def traj_inputs(t, scen):
    """launch conditions of a trajectory fixture -> (x0, y0, theta)"""
    th = t["theta"]
    if scen == "fisheye":
        return 1.0, 0.0, th
    return t["pos_x"][:len(th)], -2.0, th


def sub_rows(s_ray, d_ray, stride):
    R = s_ray.shape[2]
    last = np.zeros((3, 6, R))
    for k in range(R):
        i = int(d_ray[2, k])
        for j in range(3):
            last[j, :, k] = s_ray[max(i - 2 + j, 0), :, k]
    return s_ray[::stride], last


@pytest.fixture(scope="session")
def consts():
    c = golden("constants")
    return {k: float(c[k]) for k in c.files}


@pytest.fixture(scope="session")
def oracle_fields(consts):
    from oracle import rt_oracle as O
    cache = {}

    def get(scen):
        key = "vert_heterogeneous" if scen == "anisotropy" else scen
        if key not in cache:
            cache[key] = O.Field(key, LIMITS[key], consts["DELTA"])
        return cache[key]
    return get

"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports
every symbol include/neo360_hip.h declares; the ctypes table covers them all;
host-only helpers agree with torch.  No compute call needs a GPU here."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "neo360_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(neo_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(built_lib):
    lib = ctypes.CDLL(built_lib)
    names = _declared_symbols()
    assert len(names) >= 18
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in the header but not exported: %s" % missing


def test_ctypes_table_matches_header(built_lib):
    from neo360_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    lib = _lib.load()
    assert lib.neo_abi_version() == 1


def test_no_gpu_context_fails_loudly(built_lib):
    """Without a device the library must refuse, not fall back."""
    from neo360_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("a device is visible")
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.neo_ctx_create(0, ctypes.byref(h))
    assert rc != 0
    assert lib.neo_last_error()


def test_cpu_tensors_rejected(built_lib):
    from neo360_amd import _lib, models
    net = models.NeRF()
    rays = dict(rays_o=torch.zeros(4, 3), rays_d=torch.ones(4, 3), viewdirs=torch.ones(4, 3))
    with pytest.raises(_lib.NeoError):
        net(rays, False, False, 0.2, 3.0)


@pytest.mark.parametrize("start,end,steps", [(0.0, 1.0, 65), (0.0, 1.0, 129), (0.0, 1.0, 128), (0.0, 1.0, 256),
                                              (0.0, 1.0, 2), (0.0, 1.0, 33), (0.2, 3.0, 65), (-1.5, 0.25, 100),
                                              (1 / 128, 1 - 1 / 128 - 1.1920929e-07, 64), (1 / 64, 1 - 1 / 64 - 1.1920929e-07, 32)])
def test_linspace_host_matches_torch(built_lib, start, end, steps):
    from neo360_amd import _lib
    mine = torch.tensor(_lib.linspace(start, end, steps))
    assert torch.equal(mine, torch.linspace(start, end, steps))


def test_quantile_end_rounds_to_one():
    # 1 - 2^-32 is not representable in fp32: the reference's last quantile is exactly 1.0
    assert float(torch.linspace(0.0, 1.0 - 2 ** -32, 4)[-1]) == 1.0


def test_state_dict_keys_match_reference_layout():
    from neo360_amd import models, synth
    net = models.NeRF()
    want = synth.vanilla_state(0)
    got = net.state_dict()
    assert sorted(got) == sorted(want)
    assert all(got[k].shape == want[k].shape for k in want)
    net.load_state_dict(want, strict=True)

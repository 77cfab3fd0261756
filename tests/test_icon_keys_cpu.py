"""CPU: the key mapping of IconEngine accepts the state_dict layouts the package produces.

``icon_registration`` 1.1.2 is not installed here; the module tree is restated from its published structure:
GradientICON(regis_net = TwoStepRegistration(netPhi = DownsampleRegistration(net = TwoStepRegistration(netPhi = FFVF(net = tallUNet2),
netPsi = FFVF(net = tallUNet2))), netPsi = FFVF(net = tallUNet2))).  Every wrapper level registers an ``identity_map`` buffer
(``assign_identity_map``) and some a ``spacing``; BatchNorm3d carries ``num_batches_tracked``.  Reference call sites:
registration.py:20 (model construction + weight load), :25 (register_pair)."""
import pytest
import torch

from oai_analysis_2_amd.registration import _NET_PREFIXES, map_icon_state_dict
from oai_analysis_2_amd.synth import make_icon_state_dict


def package_like_state_dict(prefix=""):
    sd = make_icon_state_dict(0)
    full = {prefix + k: v for k, v in sd.items()}
    for pre in _NET_PREFIXES:
        for d in range(5):
            full[f"{prefix}{pre}batchNorms.{d}.num_batches_tracked"] = torch.tensor(0)
    # identity_map / spacing buffers of every wrapper level (TwoStep, Downsample, FFVF), full and half resolution
    for wrap, shape in (("", (1, 3, 80, 192, 192)), ("netPhi.", (1, 3, 80, 192, 192)), ("netPhi.net.", (1, 3, 40, 96, 96)),
                        ("netPhi.net.netPhi.", (1, 3, 40, 96, 96)), ("netPhi.net.netPsi.", (1, 3, 40, 96, 96)), ("netPsi.", (1, 3, 80, 192, 192))):
        full[f"{prefix}{wrap}identity_map"] = torch.zeros(1)        # (content irrelevant: the library builds its own identity map)
        full[f"{prefix}{wrap}spacing"] = torch.ones(3)
    return sd, full


@pytest.mark.parametrize("prefix", ["", "regis_net."])
def test_full_key_set_of_the_package_maps_onto_the_three_unets(prefix):
    sd, full = package_like_state_dict(prefix)
    if prefix:
        full["identity_map"] = torch.zeros(1)                        # GradientICON's own buffer, outside regis_net
    got = map_icon_state_dict(full)
    params = {k for k in sd if not k.endswith("num_batches_tracked")}
    assert set(got) == params and len(params) == 3 * (5 * 8 + 2)      # per U-Net: 5 levels x (down w,b + up w,b + 4 BN arrays) + lastConv w,b
    assert all(got[k] is full[prefix + k] for k in params)
    # wrapped in a checkpoint dict
    assert set(map_icon_state_dict({"model_state_dict": full, "epoch": 3})) == params


def test_a_foreign_state_dict_is_refused():
    sd, full = package_like_state_dict()
    full["ec0.0.weight"] = torch.zeros(1)                             # e.g. the segmentation checkpoint handed to the wrong loader
    with pytest.raises(KeyError):
        map_icon_state_dict(full)

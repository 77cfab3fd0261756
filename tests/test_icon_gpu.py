"""GPU parity: ICON network + warp chain (through the C ABI) vs the oracle (oracle/icon.py).

The oracle for this part is PARITY UNPINNED (icon_registration is absent from the reference tree and
this image); these tests prove the HIP path equals the restatement, at the north-star tolerance."""
import numpy as np
import pytest
import torch

from oai_analysis_2_amd.synth import make_icon_state_dict, make_volume
from oracle import icon as oicon

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("shape", [(20, 24, 24), (17, 33, 21), (40, 48, 48), (33, 45, 37)])   # the last two reach the MFMA up-conv; odd axes = crop + edges
def test_tall_unet2(shape):
    from oai_analysis_2_amd.registration import IconEngine
    sd = make_icon_state_dict(1)
    eng = IconEngine(sd, net_shape=(40, 48, 48))
    a, b = make_volume(1, shape), make_volume(2, shape)
    for which, pre in enumerate((oicon.U1, oicon.U2, oicon.U3)):
        ref = oicon.tall_unet2(torch.from_numpy(a)[None, None], torch.from_numpy(b)[None, None], sd, pre)[0].numpy()
        got = eng.unet(which, torch.from_numpy(a), torch.from_numpy(b)).cpu().numpy()
        assert got.shape == ref.shape
        assert _rel(got, ref) < TOL, (which, _rel(got, ref))


def test_phi_small_and_register_pair():
    from oai_analysis_2_amd.registration import IconEngine
    sd = make_icon_state_dict(2)
    net = (40, 48, 48)
    eng = IconEngine(sd, net_shape=net)
    A, B = make_volume(3, (50, 90, 96)), make_volume(4, (44, 100, 88))
    ref_AB, ref_BA = oicon.register_pair_arrays(A, B, sd, net_shape=net, both=True)
    got_AB, got_BA = eng.register_pair(torch.from_numpy(A), torch.from_numpy(B), both=True)
    ident = oicon.identity_map(net)[0].numpy()
    for got, ref in ((got_AB, ref_AB), (got_BA, ref_BA)):
        got, ref = got.cpu().numpy(), ref[0].numpy()
        assert np.abs(ref - ident).max() > 0.01            # a real deformation, not the identity
        assert _rel(got - ident, ref - ident) < TOL        # displacement field within 1e-4 rel (north star)


def test_phi_full_resolution_80x192x192():
    from oai_analysis_2_amd.registration import IconEngine
    sd = make_icon_state_dict(0)
    eng = IconEngine(sd)
    A, B = make_volume(5, (80, 192, 192)), make_volume(6, (80, 192, 192))
    ref = oicon.regis_net_direction(torch.from_numpy(A)[None, None], torch.from_numpy(B)[None, None], sd)[0].numpy()
    got = eng.phi(torch.from_numpy(A), torch.from_numpy(B)).cpu().numpy()
    ident = oicon.identity_map((80, 192, 192))[0].numpy()
    assert _rel(got - ident, ref - ident) < TOL


def test_graph_replay_equals_direct_launches():
    """oai_icon_forward replays one hipGraph per direction; results are bit-identical to issuing the same launches one by one,
    on the default (null) stream and on a side stream, and the replay really happens (not a silent fall-back)."""
    from oai_analysis_2_amd.registration import IconEngine
    sd = make_icon_state_dict(2)
    net = (40, 48, 48)
    A, B = torch.from_numpy(make_volume(3, net)).cuda(), torch.from_numpy(make_volume(4, net)).cuda()
    direct = IconEngine(sd, net_shape=net)
    direct.set_graph(False)
    ref = direct.phi(A, B)
    assert direct.graph_info() == (0, 0, 1)
    eng = IconEngine(sd, net_shape=net)
    got1 = eng.phi(A, B)                                  # captures, then replays
    got2 = eng.phi(B, A)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got3 = eng.phi(A, B)
    torch.cuda.current_stream().wait_stream(side)
    captured, replays, direct_runs = eng.graph_info()
    assert captured == 1 and replays == 3 and direct_runs == 0, (captured, replays, direct_runs)
    assert torch.equal(got1, ref) and torch.equal(got3, ref)
    assert torch.equal(got2, direct.phi(B, A))


@pytest.mark.parametrize("apply_bn", [True, False])
@pytest.mark.parametrize("pad_front", [True, False])
def test_restatement_switches_match_the_oracle_both_ways(apply_bn, pad_front):
    """The two recollection-dependent points of the un-vendored package (VERDICT r2 missing #4): eval BatchNorm behind the up-convs or
    none (oai_icon_create: bn_* == NULL), zero channels of pad_or_crop in front or behind (oai_icon_set_option "pad_front").  Each of
    the four combinations equals the oracle run with the same flags -- and differs from the others, so the switch is live."""
    from oai_analysis_2_amd.registration import IconEngine
    sd = make_icon_state_dict(3)
    net = (40, 48, 48)
    shape = (33, 45, 37)
    a, b = make_volume(1, shape), make_volume(2, shape)
    eng = IconEngine(sd, net_shape=net, apply_bn=apply_bn, pad_front=pad_front)
    ta, tb = torch.from_numpy(a)[None, None], torch.from_numpy(b)[None, None]
    for which, pre in enumerate((oicon.U1, oicon.U3)):
        which = which * 2
        ref = oicon.tall_unet2(ta, tb, sd, pre, apply_bn=apply_bn, pad_front=pad_front)[0].numpy()
        got = eng.unet(which, torch.from_numpy(a), torch.from_numpy(b)).cpu().numpy()
        assert _rel(got, ref) < TOL, (which, _rel(got, ref))
        other = oicon.tall_unet2(ta, tb, sd, pre, apply_bn=not apply_bn, pad_front=pad_front)[0].numpy()
        other2 = oicon.tall_unet2(ta, tb, sd, pre, apply_bn=apply_bn, pad_front=not pad_front)[0].numpy()
        assert _rel(got, other) > 100 * TOL and _rel(got, other2) > 100 * TOL
    # the whole direction (graph replay included) with the same flags
    A, B = make_volume(5, net), make_volume(6, net)
    old = dict(oicon.OPTIONS)
    try:
        oicon.OPTIONS.update(apply_bn=apply_bn, pad_front=pad_front)
        ref = oicon.regis_net_direction(torch.from_numpy(A)[None, None], torch.from_numpy(B)[None, None], sd)[0].numpy()
    finally:
        oicon.OPTIONS.update(old)
    ident = oicon.identity_map(net)[0].numpy()
    for _ in range(2):                                       # second call = graph replay
        got = eng.phi(torch.from_numpy(A), torch.from_numpy(B)).cpu().numpy()
        assert _rel(got - ident, ref - ident) < TOL

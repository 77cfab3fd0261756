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


def test_the_checkpoint_decides_apply_bn_when_nobody_states_it(capsys):
    """VERDICT r5 weak #1 (reference call site registration.py:20): BatchNorm running statistics only move, ``num_batches_tracked`` only
    counts and gamma / beta only receive gradients if ``UNet2.forward`` CALLS ``batchNorms[depth]`` in training.  A checkpoint with trained
    BatchNorm tensors therefore proves the package applies them (default = apply, equal to the oracle run with apply_bn=True and far
    from the one without); pristine tensors prove nothing and change nothing (either setting within 1e-4 of the other); a counter above
    zero alone is proof too; a stated argument or $OAI_ICON_APPLY_BN still wins."""
    import os
    from oai_analysis_2_amd.registration import ICON_Registration, IconEngine, infer_apply_bn
    net = (40, 48, 48)
    A, B = make_volume(5, net), make_volume(6, net)
    tA, tB = torch.from_numpy(A)[None, None], torch.from_numpy(B)[None, None]
    ident = oicon.identity_map(net)[0].numpy()

    def oracle_phi(sd, apply_bn):
        old = dict(oicon.OPTIONS)
        try:
            oicon.OPTIONS.update(apply_bn=apply_bn)
            return oicon.regis_net_direction(tA, tB, sd)[0].numpy() - ident
        finally:
            oicon.OPTIONS.update(old)

    trained = make_icon_state_dict(3)                                  # statistics moved, counter 1000
    assert infer_apply_bn(trained)[0] is True and oicon.infer_apply_bn(trained) is True
    eng = IconEngine(trained, net_shape=net)
    assert eng.apply_bn is True and "num_batches_tracked" in eng.apply_bn_reason
    got = eng.phi(torch.from_numpy(A), torch.from_numpy(B)).cpu().numpy() - ident
    assert _rel(got, oracle_phi(trained, True)) < TOL
    assert _rel(got, oracle_phi(trained, None)) < TOL                  # the oracle's own unstated default follows the same rule
    assert _rel(got, oracle_phi(trained, False)) > 100 * TOL           # ... and skipping them would have been a different phi

    pristine = make_icon_state_dict(3, bn="pristine")
    assert infer_apply_bn(pristine)[0] is False and oicon.infer_apply_bn(pristine) is False
    eng0 = IconEngine(pristine, net_shape=net)
    assert eng0.apply_bn is False
    got0 = eng0.phi(torch.from_numpy(A), torch.from_numpy(B)).cpu().numpy() - ident
    got1 = IconEngine(pristine, net_shape=net, apply_bn=True).phi(torch.from_numpy(A), torch.from_numpy(B)).cpu().numpy() - ident
    assert _rel(got0, oracle_phi(pristine, False)) < TOL and _rel(got1, oracle_phi(pristine, True)) < TOL
    assert _rel(got0, got1) < TOL                                      # the switch is irrelevant for pristine tensors

    absent = make_icon_state_dict(3, bn="absent")                      # no BatchNorm keys at all: nothing to apply
    assert infer_apply_bn(absent)[0] is False and IconEngine(absent, net_shape=net).apply_bn is False

    counted = {k: (torch.tensor(7) if k.endswith("num_batches_tracked") else v) for k, v in pristine.items()}
    assert infer_apply_bn(counted)[0] is True                          # called in training, statistics happen to sit at their start values
    wrapped = {"model_state_dict": {"regis_net." + k: v for k, v in trained.items()}}
    assert infer_apply_bn(wrapped)[0] is True                          # checkpoint wrappers and the GradientICON prefix

    capsys.readouterr()
    reg = ICON_Registration(trained, net_shape=net)                    # unstated: decided, and said in one line
    assert reg.register_module.apply_bn is True
    assert "apply_bn=True decided from the checkpoint" in capsys.readouterr().out
    assert ICON_Registration(trained, net_shape=net, apply_bn=False, verbose=False).register_module.apply_bn is False
    old = os.environ.get("OAI_ICON_APPLY_BN")
    try:
        os.environ["OAI_ICON_APPLY_BN"] = "0"
        assert ICON_Registration(trained, net_shape=net, verbose=False).register_module.apply_bn is False
        os.environ["OAI_ICON_APPLY_BN"] = "1"
        assert ICON_Registration(pristine, net_shape=net, verbose=False).register_module.apply_bn is True
    finally:
        if old is None:
            os.environ.pop("OAI_ICON_APPLY_BN", None)
        else:
            os.environ["OAI_ICON_APPLY_BN"] = old


# ---- the step tree is data: whatever TwoStep / Downsample / FFVF nesting the checkpoint's keys spell -------------

TREES = [("3step", (40, 48, 48)), ("4step", (40, 48, 48)), ("multires", (68, 72, 76)), ("multires4", (68, 76, 72))]


def _op_by_op(eng, tree, node, A, B):
    """network_wrappers' forward written with the library's single-op launches (eng.unet, avgpool2, compose, grid_sample3d): the
    links of the node's closure in application order."""
    from oai_analysis_2_amd import ops
    from oai_analysis_2_amd.registration import DOWN, FFVF
    kind, a, b = tree.nodes[node]
    if kind == FFVF:
        return [eng.unet(a, A, B)]
    if kind == DOWN:
        return _op_by_op(eng, tree, a, ops.avgpool2(A[None])[0], ops.avgpool2(B[None])[0])
    phi = _op_by_op(eng, tree, a, A, B)
    A_w = ops.grid_sample3d(A[None], _apply_ops(phi, A.shape))[0]
    return _op_by_op(eng, tree, b, A_w, B) + phi


def _apply_ops(links, shape):
    from oai_analysis_2_amd import ops
    c = None
    for i, d in enumerate(links):
        if i == 0:
            c = ops.compose(d, None, out_shape=shape, shortcut=tuple(d.shape[1:]) == tuple(shape))
        else:
            c = ops.compose(d, c)
    return c


@pytest.mark.parametrize("tree_name,net", TREES)
def test_step_trees_match_the_oracle_and_the_op_by_op_launches(tree_name, net):
    """Three-step, four-step ("our final 4 step registration network"), the multi-resolution cascade and its
    four-step form: (a) phi within 1e-4 rel of the oracle's recursion over the same keys; (b) the compiled plan (fused chains, one
    hipGraph) is BIT-IDENTICAL to the same tree walked op by op through the single-op entry points -- so the three-step path is
    what rounds 1-3 shipped; (c) the replayed graph equals the first (capturing) call and the direct launches."""
    from oai_analysis_2_amd.registration import IconEngine
    from oai_analysis_2_amd.synth import icon_tree_prefixes
    sd = make_icon_state_dict(4, 0.3, tree_name)
    eng = IconEngine(sd, net_shape=net)
    n_nets = len(icon_tree_prefixes(tree_name))
    assert eng.tree.net_prefixes == icon_tree_prefixes(tree_name)
    got_n, levels, chain_len = eng.describe()
    assert got_n == n_nets and sum(levels) == n_nets and chain_len == n_nets
    assert levels[:3] == {"3step": [1, 2, 0], "4step": [2, 2, 0], "multires": [1, 1, 1], "multires4": [2, 1, 1]}[tree_name]
    A, B = make_volume(5, net), make_volume(6, net)
    ref = oicon.regis_net_direction(torch.from_numpy(A)[None, None], torch.from_numpy(B)[None, None], sd)[0].numpy()
    ident = oicon.identity_map(net)[0].numpy()
    tA, tB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    got = eng.phi(tA, tB)
    rel = _rel(got.cpu().numpy() - ident, ref - ident)
    print(f"[icon tree {tree_name}] {eng.tree.describe()}: displacement rel err vs oracle {rel:.2e}, max |disp| {np.abs(ref - ident).max():.3f}")
    assert np.abs(ref - ident).max() > 0.01 and rel < TOL
    manual = _apply_ops(_op_by_op(eng, eng.tree, eng.tree.root, tA, tB), net)
    assert torch.equal(got, manual)
    assert torch.equal(eng.phi(tA, tB), got) and eng.graph_info()[0] == 1            # replay
    eng.set_graph(False)
    assert torch.equal(eng.phi(tA, tB), got)


def test_three_step_recursion_equals_the_unrolled_oracle():
    """The oracle's recursion over the key tree reproduces the hand-unrolled three-step wiring of rounds 1-3 bit for bit (CPU only,
    but it needs the synthetic weights' size: kept with the GPU tests for time)."""
    sd = make_icon_state_dict(2)
    net = (40, 48, 48)
    A, B = (torch.from_numpy(make_volume(i, net))[None, None] for i in (3, 4))
    assert torch.equal(oicon.regis_net_direction(A, B, sd), oicon.regis_net_direction_3step_unrolled(A, B, sd))


def test_malformed_trees_are_refused():
    import ctypes as C
    from oai_analysis_2_amd import _lib
    from oai_analysis_2_amd.registration import DOWN, FFVF, TWO, IconEngine
    sd = make_icon_state_dict(0)
    with pytest.raises(_lib.OaiError):                   # quarter-resolution grid 10x12x12 cannot take five 2x poolings
        IconEngine(make_icon_state_dict(0, 0.3, "multires"), net_shape=(40, 48, 48))
    eng = IconEngine(sd, net_shape=(40, 48, 48))
    lib = eng.lib
    params = (_lib.IconUnetParams * 3)()
    h = C.c_void_p()
    for bad in ([(FFVF, 0, 0), (FFVF, 0, 0), (TWO, 0, 1)],          # net 0 twice, nets 1 / 2 unused
                [(TWO, 0, 0)],                                     # a node that is its own child
                [(FFVF, 0, 0), (DOWN, 0, 0), (FFVF, 1, 0), (FFVF, 2, 0), (TWO, 1, 2)],   # node 3 unreachable
                [(FFVF, 7, 0)]):                                   # net index out of range
        nodes = (_lib.IconNode * len(bad))(*[_lib.IconNode(*nd) for nd in bad])
        assert lib.oai_icon_create(params, 3, nodes, len(bad), len(bad) - 1, 40, 48, 48, C.byref(h)) != 0

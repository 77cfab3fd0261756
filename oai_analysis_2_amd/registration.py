"""ICON atlas registration on MI355X behind the reference's ``ICON_Registration`` surface.

Mirrors oai_analysis/registration.py:18-27: ``ICON_Registration().register(fixed, moving)`` returns
phi_fixed_moving -- the map that pulls a *fixed*-space (patient) image onto the *moving* (atlas)
grid, i.e. ``resample(patient_prob, transform=phi, reference=atlas)`` (test/test_all.py:42-58).

The arithmetic of ``icon_registration.itk_wrapper.register_pair`` (resize -> three tallUNet2s with
warps/composes -> dense phi -> displacement in network-voxel units + two centring affines) runs as
hand-written HIP kernels through liboai_hip.so; see oracle/icon.py for the restated algorithm.
ITK is optional: the result is a :class:`DisplacementTransform` (numpy displacement field + affines,
the exact content of the ITK CompositeTransform) with ``to_itk()`` when ``itk`` is importable.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from .image import Image, as_image

NET_SHAPE = (80, 192, 192)   # OAI_knees_gradICON_model input_shape[2:]


def network_affine(img: Image, net_shape_dhw: Sequence[int]):
    """``itk_wrapper.resampling_transform(image, shape)``: p_phys = M (x_net - c_net) + c_img (xyz)."""
    size = img.size_xyz.astype(np.float64)
    shape = np.asarray(net_shape_dhw[::-1], np.float64)
    M = img.direction @ np.diag(img.spacing * (size / shape))
    c_net = (shape - 1.0) / 2.0
    c_img = img.origin + img.direction @ (img.spacing * (size - 1.0) / 2.0)
    return M, c_net, c_img


def resample_affines(image_A: Image, image_B: Image, net_shape_dhw: Sequence[int]):
    """Host-side fp64 composition of the two affine legs of phi_AB for ``oai_resample_through_disp``.

    b_index_to_net : B index -> physical_B -> network index space   (from_network_space)
    net_to_a_index : network index space -> physical_A -> A continuous index   (to_network_space)
    """
    M_A, c_net, c_A = network_affine(image_A, net_shape_dhw)
    M_B, _, c_B = network_affine(image_B, net_shape_dhw)
    P_B, o_B = image_B.index_to_physical_affine()
    P_A, o_A = image_A.index_to_physical_affine()
    M_B_inv = np.linalg.inv(M_B)
    A1 = M_B_inv @ P_B
    b1 = M_B_inv @ (o_B - c_B) + c_net
    P_A_inv = np.linalg.inv(P_A)
    A2 = P_A_inv @ M_A
    b2 = P_A_inv @ (c_A - M_A @ c_net - o_A)
    return (A1, b1), (A2, b2)


@dataclass
class DisplacementTransform:
    """What ``create_itk_transform`` builds: B-physical -> A-physical through a network-space field."""
    displacement: np.ndarray            # float64 [D,H,W,3], xyz components, network-voxel units
    image_A: Image                      # metadata only (array may be dropped)
    image_B: Image
    phi: Optional[np.ndarray] = None    # float32 [3,D,H,W] dense map in [0,1] units (network grid)

    @property
    def net_shape(self):
        return self.displacement.shape[:3]

    def affines(self):
        return resample_affines(self.image_A, self.image_B, self.net_shape)

    def to_itk(self):  # pragma: no cover - itk is absent in this environment
        import itk
        dim = 3
        tr = itk.DisplacementFieldTransform[(itk.D, dim)].New()
        tr.SetDisplacementField(itk.image_from_array(np.ascontiguousarray(self.displacement), is_vector=True))

        def affine(img):
            M, c_net, c_img = network_affine(img, self.net_shape)
            t = itk.CenteredAffineTransform[itk.D, 3].New()
            t.SetCenter([float(v) for v in c_net])
            t.SetOffset([float(v) for v in (c_img - c_net)])
            t.SetMatrix(itk.matrix_from_array(np.ascontiguousarray(M)))
            return t

        comp = itk.CompositeTransform[itk.D, dim].New()
        comp.PrependTransform(affine(self.image_B).GetInverseTransform())
        comp.PrependTransform(tr)
        comp.PrependTransform(affine(self.image_A))
        return comp


# --------------------------------------------------------------------------------------------------
# engine: N tallUNet2s resident on the GPU + the warp/compose chains of the checkpoint's step tree, all in liboai_hip.so

FFVF, DOWN, TWO = 0, 1, 2                      # include/oai_hip.h: OAI_ICON_FFVF / _DOWN / _TWO
_UNET_LEAVES = ("downConvs", "upConvs", "batchNorms", "lastConv", "residues")   # attribute names of networks.UNet2
_BUFFER_LEAVES = ("identity_map", "spacing", "num_batches_tracked", "_extra_state")


@dataclass
class IconTree:
    """The module tree of ``regis_net`` as the library wants it (``oai_icon_create``): ``nodes[i] = (kind, a, b)`` with kind FFVF
    (a = U-Net number), DOWN (a = child node) or TWO (a = netPhi node, b = netPsi node); ``net_prefixes[k]`` = state_dict prefix of
    U-Net k's parameters.  U-Nets are numbered in execution order (netPhi before netPsi)."""
    nodes: list
    root: int
    net_prefixes: list

    def describe(self, node: Optional[int] = None) -> str:
        kind, a, b = self.nodes[self.root if node is None else node]
        if kind == FFVF:
            return f"u{a}"
        if kind == DOWN:
            return f"Down({self.describe(a)})"
        return f"TwoStep({self.describe(a)}, {self.describe(b)})"


def parse_icon_tree(keys) -> IconTree:
    """The nesting of ``netPhi`` / ``netPsi`` / ``net`` in the parameter names IS the module tree of the package's wrappers
    (``TwoStepRegistration(netPhi, netPsi)``, ``DownsampleRegistration(net)``, ``FunctionFromVectorField(net)`` around
    ``networks.tallUNet2``; reference call site registration.py:20): a module with children {netPhi, netPsi} is a TwoStep, one whose
    only child ``net`` holds U-Net parameters is an FFVF, one whose ``net`` is another wrapper is a Downsample.  Anything else raises
    ``KeyError`` -- a foreign state_dict must not load."""
    keys = [k for k in keys if k.rsplit(".", 1)[-1] not in _BUFFER_LEAVES]
    nodes, prefixes = [], []

    def children(prefix):
        return sorted({k[len(prefix):].split(".", 1)[0] for k in keys if k.startswith(prefix)})

    def visit(prefix):
        kids = children(prefix)
        if kids == ["netPhi", "netPsi"]:
            a = visit(prefix + "netPhi.")
            b = visit(prefix + "netPsi.")
            nodes.append((TWO, a, b))
        elif kids == ["net"]:
            grand = children(prefix + "net.")
            if grand and all(g in _UNET_LEAVES for g in grand):
                prefixes.append(prefix + "net.")
                nodes.append((FFVF, len(prefixes) - 1, 0))
            elif grand in (["netPhi", "netPsi"], ["net"]):
                a = visit(prefix + "net.")
                nodes.append((DOWN, a, 0))
            else:
                raise KeyError(f"unexpected keys in the ICON state_dict under '{prefix}net.': {grand[:4]}")
        else:
            raise KeyError(f"unexpected keys in the ICON state_dict under '{prefix}': {kids[:4]}")
        return len(nodes) - 1

    root = visit("")
    return IconTree(nodes, root, prefixes)


def map_icon_state_dict(state_dict: Dict[str, torch.Tensor], with_tree: bool = False):
    """U-Net parameters (and the step tree) out of whatever the caller holds: the ``regis_net`` state_dict itself, the whole
    ``GradientICON`` module's (keys prefixed ``regis_net.``), or a checkpoint dict wrapping either.  Non-parameter entries of the
    package's modules are recognised and dropped -- the registered ``identity_map`` / ``spacing`` buffers of every wrapper level
    and BatchNorm's ``num_batches_tracked`` -- anything else that is not a parameter of a tallUNet2 inside the wrapper tree raises
    (a wrong file should not load).  The tree is whatever the keys spell: SURVEY Appendix A's three steps, the four-step form with
    a second full-resolution U-Net, a multi-resolution cascade (``parse_icon_tree``)."""
    sd = state_dict
    for wrap in ("model_state_dict", "state_dict"):
        if wrap in sd and isinstance(sd[wrap], dict):
            sd = sd[wrap]
    if any(k.startswith("regis_net.") for k in sd):
        sd = {k[len("regis_net."):]: v for k, v in sd.items() if k.startswith("regis_net.")}
    params = {k: v for k, v in sd.items() if k.rsplit(".", 1)[-1] not in _BUFFER_LEAVES}
    tree = parse_icon_tree(params.keys())
    expected = set()
    for pre in tree.net_prefixes:
        for d in range(5):
            expected |= {f"{pre}downConvs.{d}.weight", f"{pre}downConvs.{d}.bias", f"{pre}upConvs.{d}.weight", f"{pre}upConvs.{d}.bias"}
            expected |= {f"{pre}batchNorms.{d}.{n}" for n in ("weight", "bias", "running_mean", "running_var")}
        expected |= {f"{pre}lastConv.weight", f"{pre}lastConv.bias"}
    unknown = sorted(set(params) - expected)
    if unknown:
        raise KeyError(f"unexpected keys in the ICON state_dict: {unknown[:4]}")
    missing = sorted(k for k in expected - set(params) if ".batchNorms." not in k)
    if missing:
        raise KeyError(f"ICON state_dict is missing {missing[:4]}")
    for pre in tree.net_prefixes:             # tallUNet2 = UNet2(5, [[2,16,32,64,256,512],[16,32,64,128,256]], 3): nothing else is built
        if tuple(params[f"{pre}downConvs.0.weight"].shape) != (16, 2, 3, 3, 3) or tuple(params[f"{pre}lastConv.weight"].shape) != (3, 18, 3, 3, 3) \
                or tuple(params[f"{pre}upConvs.4.weight"].shape) != (512, 256, 4, 4, 4):
            raise KeyError(f"'{pre}' is not a 3-D tallUNet2 (channel widths differ)")
    return (params, tree) if with_tree else params


class IconEngine:
    """``OAI_knees_gradICON_model().regis_net`` as packed weights + HIP kernels; the step tree is read from the state_dict's keys
    (``self.tree``), nothing about the number of U-Nets or their resolutions is assumed.

    ``apply_bn`` / ``pad_front``: the two points where the restatement of the un-vendored ``icon_registration`` 1.1.2 rests on
    recollection (oracle/icon.py:OPTIONS): eval-mode BatchNorm3d behind every up-conv or none, and the side on which
    ``pad_or_crop`` adds zero channels.  ``apply_bn=None`` (the default) lets the CHECKPOINT decide (:func:`infer_apply_bn`): trained
    BatchNorm tensors are proof that ``UNet2.forward`` calls the layers, pristine ones make the switch irrelevant.  Zero channels
    go in front."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], net_shape: Sequence[int] = NET_SHAPE, device=None,
                 apply_bn: Optional[bool] = None, pad_front: bool = True):
        import ctypes as C
        if apply_bn is None:
            apply_bn, self.apply_bn_reason = infer_apply_bn(state_dict)
        else:
            self.apply_bn_reason = "stated by the caller"
        state_dict, self.tree = map_icon_state_dict(state_dict, with_tree=True)
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.OaiError("no HIP device: the MI355X path has no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.net_shape = tuple(int(v) for v in net_shape)
        keep = []

        def ptr(key):
            if key not in state_dict:
                raise KeyError(f"ICON state_dict is missing {key}")
            t = state_dict[key].detach().to("cpu", torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        n_nets = len(self.tree.net_prefixes)
        params = (_lib.IconUnetParams * n_nets)()
        for n, pre in enumerate(self.tree.net_prefixes):
            p = params[n]
            for d in range(5):
                p.down_w[d], p.down_b[d] = ptr(f"{pre}downConvs.{d}.weight"), ptr(f"{pre}downConvs.{d}.bias")
                p.up_w[d], p.up_b[d] = ptr(f"{pre}upConvs.{d}.weight"), ptr(f"{pre}upConvs.{d}.bias")
                if apply_bn:
                    p.bn_gamma[d], p.bn_beta[d] = ptr(f"{pre}batchNorms.{d}.weight"), ptr(f"{pre}batchNorms.{d}.bias")
                    p.bn_mean[d], p.bn_var[d] = ptr(f"{pre}batchNorms.{d}.running_mean"), ptr(f"{pre}batchNorms.{d}.running_var")
            p.last_w, p.last_b = ptr(f"{pre}lastConv.weight"), ptr(f"{pre}lastConv.bias")
        nodes = (_lib.IconNode * len(self.tree.nodes))(*[_lib.IconNode(*nd) for nd in self.tree.nodes])
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_icon_create(params, n_nets, nodes, len(self.tree.nodes), self.tree.root, *self.net_shape,
                                                C.byref(handle)), "oai_icon_create")
        self._h = handle
        self.n_nets = n_nets
        self.apply_bn, self.pad_front = bool(apply_bn), bool(pad_front)
        if not pad_front:
            _lib.check(self.lib.oai_icon_set_option(self._h, b"pad_front", 0), "oai_icon_set_option")
        self._ws = torch.empty(int(self.lib.oai_icon_workspace_bytes(self._h)), dtype=torch.uint8, device=self.device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.oai_icon_destroy(h)
            self._h = None

    def describe(self):
        """(n_nets, U-Nets per halving level, length of the final compose chain) as the library compiled the tree."""
        import ctypes as C
        n, lv, cl = C.c_int(), (C.c_int * 8)(), C.c_int()
        _lib.check(self.lib.oai_icon_describe(self._h, C.byref(n), lv, C.byref(cl)), "oai_icon_describe")
        return n.value, list(lv), cl.value

    def set_graph(self, enable: bool) -> None:
        """hipGraph replay of one direction's launches (default on); off = the same launches issued one by one."""
        _lib.check(self.lib.oai_icon_set_graph(self._h, int(enable)), "oai_icon_set_graph")

    def graph_info(self):
        """(captured, replays, direct_runs): captured 1 = graph in use, 0 = not captured yet, -1 = capture failed (direct launches)."""
        import ctypes as C
        cap, rep, dr = C.c_int(), C.c_longlong(), C.c_longlong()
        _lib.check(self.lib.oai_icon_graph_info(self._h, C.byref(cap), C.byref(rep), C.byref(dr)), "oai_icon_graph_info")
        return cap.value, rep.value, dr.value

    def unet(self, which: int, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """One tallUNet2: a, b [D,H,W] -> displacement [3,D,H,W] (unit-test seam)."""
        a = a.to(self.device, torch.float32).contiguous()
        b = b.to(self.device, torch.float32).contiguous()
        D, H, W = a.shape
        out = torch.empty((3, D, H, W), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_icon_unet_forward(self._h, which, a.data_ptr(), b.data_ptr(), D, H, W, out.data_ptr(),
                                                      self._ws.data_ptr(), self._ws.numel(),
                                                      torch.cuda.current_stream().cuda_stream), "oai_icon_unet_forward")
        return out

    def phi(self, A_net: torch.Tensor, B_net: torch.Tensor) -> torch.Tensor:
        """phi_AB(identity) [3,D,H,W] for network-resolution images (one direction of GradientICON.forward)."""
        A_net = A_net.to(self.device, torch.float32).contiguous()
        B_net = B_net.to(self.device, torch.float32).contiguous()
        if tuple(A_net.shape) != self.net_shape or tuple(B_net.shape) != self.net_shape:
            raise ValueError(f"images must be resized to the network shape {self.net_shape}")
        out = torch.empty((3, *self.net_shape), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_icon_forward(self._h, A_net.data_ptr(), B_net.data_ptr(), out.data_ptr(),
                                                 self._ws.data_ptr(), self._ws.numel(),
                                                 torch.cuda.current_stream().cuda_stream), "oai_icon_forward")
        return out

    def register_pair(self, image_A: torch.Tensor, image_B: torch.Tensor, both: bool = False):
        """``itk_wrapper.register_pair`` on device tensors [z,y,x]: resize -> net -> dense phi (AB[, BA])."""
        A = ops.resize_trilinear(image_A.to(self.device, torch.float32)[None], self.net_shape)[0]
        B = ops.resize_trilinear(image_B.to(self.device, torch.float32)[None], self.net_shape)[0]
        phi_AB = self.phi(A, B)
        phi_BA = self.phi(B, A) if both else None
        return phi_AB, phi_BA


def _unwrap_state_dict(state_dict):
    sd = state_dict
    for wrap in ("model_state_dict", "state_dict"):
        if wrap in sd and isinstance(sd[wrap], dict):
            sd = sd[wrap]
    return sd


def _has_nontrivial_batchnorm(state_dict) -> bool:
    """True if any ``batchNorms.*`` tensor of the checkpoint differs from a freshly constructed BatchNorm3d (gamma 1, beta 0, running mean 0,
    running variance 1): only then does applying or skipping it change phi."""
    for k, v in _unwrap_state_dict(state_dict).items():
        if ".batchNorms." not in k and not k.startswith("batchNorms."):
            continue
        t = torch.as_tensor(v).float()
        if k.endswith("num_batches_tracked"):
            continue
        want = 1.0 if (k.endswith(".weight") or k.endswith("running_var")) else 0.0
        if t.numel() and float((t - want).abs().max()) > 1e-6:
            return True
    return False


def infer_apply_bn(state_dict) -> Tuple[bool, str]:
    """Whether ``UNet2.forward`` applies ``self.batchNorms[depth]`` -- read off the checkpoint (reference call site registration.py:20 loads
    it; the package is not installed here).  A BatchNorm3d's ``running_mean`` / ``running_var`` only move and its ``num_batches_tracked`` only
    counts when the module is CALLED in training mode, and gamma / beta only receive gradients when its output feeds the loss.  So:

    * any ``num_batches_tracked > 0`` or any BatchNorm tensor off its constructor value  =>  the package called the layers while this
      checkpoint was trained  =>  they are part of the function the weights were fitted to: apply them;
    * all pristine (gamma 1, beta 0, mean 0, var 1, zero batches)  =>  the layers were never called; an eval-mode BatchNorm with those
      tensors is x / sqrt(1 + eps), 5e-6 relative per level -- either setting gives the same phi within 1e-4; skip them.

    Returns (apply_bn, reason).  An explicit ``apply_bn=`` argument or ``$OAI_ICON_APPLY_BN`` overrides this."""
    sd = _unwrap_state_dict(state_dict)
    tracked = [int(torch.as_tensor(v).max()) for k, v in sd.items()
               if k.endswith("num_batches_tracked") and (".batchNorms." in k or k.startswith("batchNorms.")) and torch.as_tensor(v).numel()]
    if any(n > 0 for n in tracked):
        return True, f"num_batches_tracked = {max(tracked)} > 0: the BatchNorm layers were called in training"
    if _has_nontrivial_batchnorm(sd):
        return True, "BatchNorm tensors differ from their constructor values: the layers were trained, hence called"
    return False, "BatchNorm tensors are pristine (never called in training): applying them would change phi by < 1e-4"


_BN_NOTED = False


def _note_bn_decision_once(apply_bn: bool, reason: str) -> None:
    global _BN_NOTED
    if _BN_NOTED:
        return
    _BN_NOTED = True
    print(f"ICON_Registration: apply_bn={apply_bn} decided from the checkpoint -- {reason}")


class ICON_Registration:
    """Drop-in for oai_analysis.registration.ICON_Registration (registration.py:18-27).

    ``weights``: a ``regis_net`` state_dict, a path to one (``torch.load``), or None to read
    ``$OAI_DATA_DIR/icon_weights.pth`` (the reference downloads them; there is no network here).
    """

    def __init__(self, weights=None, net_shape: Sequence[int] = NET_SHAPE, device=None, verbose: bool = True,
                 apply_bn: Optional[bool] = None, pad_front: bool = True):
        """``apply_bn``: whether the eval-mode BatchNorm3d behind every up-conv of ``tallUNet2`` is applied.  Stated -- ``apply_bn=True /
        False`` or ``$OAI_ICON_APPLY_BN=1 / 0`` -- it is obeyed; unstated, the checkpoint decides (:func:`infer_apply_bn`: trained BatchNorm
        statistics prove the package calls the layers) and one line says what was decided and why.  ``tests/test_icon_pin_gpu.py``
        checks the inference against the package itself wherever ``icon_registration`` imports."""
        import os
        if weights is None:
            root = os.environ.get("OAI_DATA_DIR")
            if not root:
                raise ValueError("ICON weights needed: pass weights= or set OAI_DATA_DIR (no network download here)")
            weights = os.path.join(root, "icon_weights.pth")
        if isinstance(weights, (str, os.PathLike)):
            if not os.path.isfile(weights):
                raise ValueError(f"=> no checkpoint found at '{weights}'")
            weights = torch.load(weights, map_location="cpu")
        if apply_bn is None and os.environ.get("OAI_ICON_APPLY_BN", "") in ("0", "1"):
            apply_bn = os.environ["OAI_ICON_APPLY_BN"] == "1"
        if apply_bn is None:
            apply_bn, why = infer_apply_bn(weights)
            if verbose:
                _note_bn_decision_once(apply_bn, why)
        self.register_module = IconEngine(weights, net_shape, device, apply_bn=apply_bn, pad_front=pad_front)
        self.verbose = verbose

    def register(self, fixed_image, moving_image) -> DisplacementTransform:
        fixed, moving = as_image(fixed_image), as_image(moving_image)
        if self.verbose:
            print("fixed range", np.min(fixed.array), np.max(fixed.array))          # registration.py:23-24
            print("moving range", np.min(moving.array), np.max(moving.array))
        if np.max(fixed.array) == np.min(fixed.array) or np.max(moving.array) == np.min(moving.array):
            raise AssertionError("constant image")                                   # register_pair's asserts
        eng = self.register_module
        A = torch.from_numpy(np.ascontiguousarray(fixed.array, dtype=np.float32)).to(eng.device)
        B = torch.from_numpy(np.ascontiguousarray(moving.array, dtype=np.float32)).to(eng.device)
        phi_AB, _ = eng.register_pair(A, B, both=False)
        disp = ops.phi_to_itk_displacement(phi_AB)
        return DisplacementTransform(disp.cpu().numpy(), _meta_only(fixed), _meta_only(moving), phi_AB.cpu().numpy())


def _meta_only(img: Image) -> Image:
    """Geometry of ``img`` with a zero-stride placeholder array of the same shape (no voxel copy)."""
    arr = np.broadcast_to(np.zeros((), dtype=np.float32), img.array.shape)
    return Image(arr, img.spacing.copy(), img.origin.copy(), img.direction.copy())


def deform_probmap(phi_AB: DisplacementTransform, image_A, image_B, prob_map, device=None) -> Image:
    """The reference's ``deform_probmap`` helper (test/test_all.py:42-52, dask_processing.py:95-111) on the GPU."""
    A, B, P = as_image(image_A), as_image(image_B), as_image(prob_map)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    disp = torch.from_numpy(np.ascontiguousarray(phi_AB.displacement)).to(dev)
    prob = torch.from_numpy(np.ascontiguousarray(P.array, dtype=np.float32)).to(dev)
    b2n, n2a = resample_affines(A, B, phi_AB.net_shape)
    out = ops.resample_through_disp(prob, disp, b2n, n2a, B.array.shape)
    return B.like(out.cpu().numpy().astype(np.float64))

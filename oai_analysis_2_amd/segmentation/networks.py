"""Network registry with the reference's names (oai_analysis/segmentation/networks.py:849-866).

On this path a "network" is not an ``nn.Module``: it is a description (constructor arguments) plus a
state_dict that :class:`~oai_analysis_2_amd.segmentation.engine.UNetEngine` packs for the HIP kernels.
Only ``UNet`` is implemented -- it is the one the shipped training config selects; the other registry
names of the reference (UNet_light*, UNetClassWise) are legacy training variants.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .engine import UNetEngine


class UNet:
    """Constructor-compatible stand-in for ``UNet(in_channels, n_classes, bias=False, BN=False)`` (:38-66)."""

    def __init__(self, in_channels, n_classes, bias=False, BN=False):
        if in_channels != 1:
            raise NotImplementedError("the HIP path implements the reference's in_channels=1 configuration")
        self.in_channel, self.n_classes, self.bias, self.BN = in_channels, n_classes, bias, BN
        self._state: Optional[Dict[str, torch.Tensor]] = None
        self._engine: Optional[UNetEngine] = None
        self.device = None

    # the nn.Module calls Segmenter3DInPatch.pred_setup makes (segmenter.py:56-61)
    def load_state_dict(self, state_dict, strict=True):
        has_bn = any(k.endswith("running_mean") for k in state_dict)
        has_bias = "ec0.0.bias" in state_dict
        if strict and (has_bn != bool(self.BN) or has_bias != bool(self.bias)):
            raise RuntimeError("state_dict does not match UNet(bias=%s, BN=%s)" % (self.bias, self.BN))
        if strict and state_dict["dc0.weight"].shape[0] != self.n_classes:
            raise RuntimeError("state_dict n_classes mismatch")
        self._state = dict(state_dict)
        self._engine = None

    def to(self, device):
        self.device = device
        return self

    def eval(self):
        return self

    @property
    def engine(self) -> UNetEngine:
        if self._state is None:
            raise RuntimeError("no weights loaded (the reference would call weights_init(); "
                               "random weights are not useful for inference)")
        if self._engine is None:
            self._engine = UNetEngine(self._state, self.device)
        return self._engine

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """``model(x[B,1,D,H,W]) -> logits[B,n_classes,D,H,W]`` on the GPU (networks.py:109-149)."""
        return self.engine.forward_tiles(x)


network_dic = {"UNet": UNet}


def get_available_networks():
    return list(network_dic.keys())


def get_network(network_name):
    """Like the reference: unknown names return None instead of raising (networks.py:858-862)."""
    if network_name in get_available_networks():
        return network_dic[network_name]
    return None

"""``Segmenter3DInPatchClassWise`` with the reference's constructor, config keys and ``segment`` signature
(oai_analysis/segmentation/segmenter.py:90-131) on the MI355X path.

What differs underneath: the volume is uploaded once; Partition / the tile-batch loop / sigmoid /
threshold / assemble all run on the device through liboai_hip.so (oai_segment_tiles +
oai_stitch_blocks); one D2H copy returns the two stitched maps.
"""
from __future__ import annotations

import json
from abc import ABC, abstractmethod

import numpy as np
import torch

from ..image import Image, as_image
from .engine import tile_grid
from .networks import get_network
from .utils import initialize_model


def _volume_id(vol) -> str:
    """sha256 of a volume's fp32 voxels + its shape: the identity a calibration sidecar records."""
    import hashlib
    a = vol.detach().to("cpu", torch.float32).contiguous().numpy() if torch.is_tensor(vol) else np.ascontiguousarray(vol, dtype=np.float32)
    return "x".join(str(v) for v in a.shape) + ":" + hashlib.sha256(a.tobytes()).hexdigest()


def load_json_to_dict(json_file):
    """The training config is JSON (despite its .pth.tar name): segmenter.py:14-17, module_parameters.py:38-50."""
    with open(json_file) as f:
        return json.load(f)


class Segmenter(ABC):
    @abstractmethod
    def __init__(self, *args, **kwargs):
        self.model = None
        self.config = None

    @abstractmethod
    def segment(self, *args, **kwargs):
        pass


class Segmenter3DInPatch(Segmenter):
    def __init__(self, mode=None, config=None):
        self.model = None
        self.config = config
        self.ready = False

    def pred_setup(self):
        """segmenter.py:51-62: read the JSON config, build the network, strict-load the checkpoint."""
        training_config = load_json_to_dict(self.config["training_config_file"])
        self.patch_size = tuple(int(v) for v in training_config["patch_size"])            # (x,y,z)
        self.tile_zyx = self.patch_size[::-1]
        net_cls = get_network(training_config["model"])
        if net_cls is None:
            raise NotImplementedError("model %r is not available on the MI355X path" % (training_config["model"],))
        self.model = net_cls(**training_config["model_setting"])
        device = self.config.get("device", "cuda")
        if str(device).startswith("cpu"):
            raise RuntimeError("device='cpu' requested: this package is the MI355X path and has no CPU fallback")
        self.device = torch.device(device if str(device) != "cuda" else "cuda:%d" % torch.cuda.current_device())
        initialize_model(self.model, ckpoint_path=self.config["ckpoint_path"])
        self.model.to(self.device)
        self.model.eval()
        # fp16x3's activation exponents belong to the checkpoint, not to the first volume a process happens to see: config key
        # "fp16_calibration_file" (default: "<ckpoint_path>.fp16cal.json"; False / "" = none) names a JSON sidecar -- 18 exponents, the
        # sha256 of the parameters (+ bn_eps), the census they were chosen from and the calibration volume's identity -- that is READ
        # when it exists, so that every process, rank and restart segments a given volume with the same arithmetic.  It is WRITTEN only
        # on request (no side-effect files next to a user's checkpoint, no exponents pinned by whatever volume came first):
        # by `calibrate(image)` -- the explicit step, on a volume the caller chose as representative -- or, with config key
        # "fp16_calibration_write": True, after the first automatic calibration.
        cal = self.config.get("fp16_calibration_file", str(self.config["ckpoint_path"]) + ".fp16cal.json")
        self.calibration_file = str(cal) if cal else None
        self.calibration_write = bool(self.config.get("fp16_calibration_write", False))
        self.ready = True

    def train(self, *args, **kwargs):      # the reference's training entry points are stubs (segmenter.py:64-70)
        raise NotImplementedError("inference only")

    def test(self, *args, **kwargs):
        pass

    def segment(self, image):
        pass


class Segmenter3DInPatchClassWise(Segmenter3DInPatch):
    def __init__(self, mode=None, config=None):
        super().__init__(mode, config)

    def segment_array(self, vol_zyx, if_output_prob_map=False, tile_range=None, as_device_tensor=False):
        """numpy/torch [z,y,x] fp32 -> maps [n_classes, z, y, x] (fp32; the f64 cast happens at the API edge)."""
        if not self.ready:
            self.pred_setup()
        eng = self.model.engine
        vol = torch.as_tensor(np.ascontiguousarray(vol_zyx, dtype=np.float32) if not torch.is_tensor(vol_zyx) else vol_zyx)
        vol = vol.to(eng.device, torch.float32)
        ovl_xyz = tuple(int(v) for v in self.config["overlap_size"])
        ovl_zyx = ovl_xyz[::-1]
        crop_zyx = (ovl_xyz[2], ovl_xyz[0], ovl_xyz[1])        # assemble's crop_size indexing, image_transforms.py:511-512
        # batch_size of the reference config is a host-loop knob (4 tiles per H2D/D2H round trip, segmenter.py:109-119);
        # here it only sizes the activation workspace, so use a device-sized batch unless told otherwise
        batch = int(self.config.get("device_batch_size", 0)) or None      # None: as many tiles per pass as HBM allows
        # conv arithmetic: config["precision"] in {"fp16x3" (default: fp32-grade split-fp16, 2.6x faster), "bf16x6", "f32"}
        precision = self.config.get("precision", "fp16x3")
        if eng.precision != precision:
            eng.set_precision(precision)
        if precision == "fp16x3" and (eng.calibration_file != self.calibration_file or eng.calibration_write != self.calibration_write):
            eng.set_calibration_file(self.calibration_file, write=self.calibration_write)
        if precision == "fp16x3" and eng.calibration_status() == "uncalibrated":
            eng.calibration_volume_id = _volume_id(vol) if self.calibration_write else None    # what a sidecar will say it was calibrated on
        blocks = eng.segment_tiles(vol, self.tile_zyx, ovl_zyx, tile_range, 0 if if_output_prob_map else 1, batch,
                                   crop_zyx if min(crop_zyx) > 0 else None)
        # (eng.effective_precision: "f32" when this network's calibration was refused -- then there is no range window to leave)
        if precision == "fp16x3" and eng.effective_precision == "fp16x3":
            raised = eng.range_overflow()
            eng.note_volume_flag(raised)                            # three flagged volumes in a row under a calibration FILE: the file is dropped, the next volume recalibrates
            if raised:
                # outside the calibrated window (an activation beyond 65504, or a layer > 128 x quieter than at calibration): repeat this
                # volume with exact fp32 MFMA arithmetic
                print("WARNING: activations outside the fp16x3 range window, repeating the segmentation in fp32")
                eng.set_precision("f32")
                blocks = eng.segment_tiles(vol, self.tile_zyx, ovl_zyx, tile_range, 0 if if_output_prob_map else 1, batch,
                                           crop_zyx if min(crop_zyx) > 0 else None)
                eng.set_precision(precision)
        if tile_range is not None:
            return blocks
        if min(crop_zyx) == 0:
            # numpy's x[c:-c] with c == 0 is empty: the reference returns an all-zero map in that case
            maps = torch.zeros((eng.n_classes, *vol.shape), dtype=torch.float32, device=eng.device)
        else:
            maps = eng.stitch(blocks, vol.shape, self.tile_zyx, ovl_zyx, crop_zyx)
        return maps if as_device_tensor else maps.cpu().numpy()

    def calibrate(self, image, write: bool = True) -> dict:
        """The explicit fp16x3 calibration step: choose the activation exponents on ``image`` -- a volume the caller picked as representative
        of the cohort -- and (``write``) store them in the checkpoint's sidecar with the census and the volume's identity.  Every later
        process that loads this checkpoint reads them instead of calibrating on whatever volume it happens to see first."""
        if not self.ready:
            self.pred_setup()
        eng = self.model.engine
        if eng.fp16_refused:                                         # set_precision("fp16x3") would silently stay "f32" and calibrate_volume raise
            return {"status": "refused_f32", "passes": -1, "act_exponents": eng.act_exponents()[0], "census_max": None, "volume_id": None, "file": None}
        if eng.precision != "fp16x3":
            eng.set_precision("fp16x3")
        vol = torch.as_tensor(np.ascontiguousarray(as_image(image).array, dtype=np.float32)).to(eng.device)
        ovl_xyz = tuple(int(v) for v in self.config["overlap_size"])
        crop_zyx = (ovl_xyz[2], ovl_xyz[0], ovl_xyz[1])
        eng.calibration_file, eng.calibration_write = self.calibration_file, False
        if eng.calibration_status() == "calibrated":
            eng.drop_calibration()                                   # (an explicit request replaces whatever the engine held -- in the handle too)
            eng._dropped_file = None                                 # ... which is no verdict about the file
        eng.calibration_volume_id = _volume_id(vol)
        passes = eng.calibrate_volume(vol, self.tile_zyx, ovl_xyz[::-1], crop_zyx if min(crop_zyx) > 0 else None)
        if write and self.calibration_file and eng.calibration_status() == "calibrated":
            eng.save_calibration(self.calibration_file, note=f"Segmenter.calibrate: {passes} passes")
        return {"status": eng.calibration_status(), "passes": passes, "act_exponents": eng.act_exponents()[0], "census_max": eng.calibration_census,
                "volume_id": eng.calibration_volume_id, "file": self.calibration_file if write else None}

    def segment(self, image, if_output_prob_map=False, if_output_itk=True):
        """(FC, TC) exactly like segmenter.py:100-131: float64 maps (prob) or bool-valued maps (mask)."""
        img = as_image(image)
        maps = self.segment_array(img.array, if_output_prob_map)
        outs = []
        for c in range(2):
            arr = maps[c].astype(np.float64)           # Partition.assemble returns float64 (image_transforms.py:493)
            outs.append(img.like(arr) if if_output_itk else arr)
        return outs[0], outs[1]

"""Handle over the HIP U-Net (liboai_hip.so): weight ingest from a reference state_dict,
workspace ownership, and the three launch entry points (tiles / segment / stitch)."""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import warnings
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib

# reference layer order (include/oai_hip.h: OAI_UNET_NUM_LAYERS) and kinds
LAYER_ORDER = ["ec0", "ec1", "ec2", "ec3", "ec4", "ec5", "ec6", "ec7",
               "dc9", "dc8", "dc7", "dc6", "dc5", "dc4", "dc3", "dc2", "dc1", "dc0"]
KIND = {"dc9": 2, "dc6": 2, "dc3": 2, "dc0": 3}
BN_EPS = 1e-5


def _kind(name: str) -> int:
    if name in KIND:
        return KIND[name]
    return 0 if name.startswith("ec") else 1


def tile_grid(size_zyx: Sequence[int], tile_zyx: Sequence[int], overlap_zyx: Sequence[int]):
    """effective size, grid and tile count of Partition (image_transforms.py:407-408), z,y,x order."""
    eff = [int(t) - 2 * int(o) for t, o in zip(tile_zyx, overlap_zyx)]
    if min(eff) <= 0:
        raise ValueError("overlap_size too large for patch_size")
    grid = [-(-int(s) // e) for s, e in zip(size_zyx, eff)]
    return eff, grid, grid[0] * grid[1] * grid[2]


class UNetEngine:
    """The reference ``UNet`` (networks.py:38-149) as a resident set of packed weights on the GPU."""

    PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16x6": 2, "fp16x3": 3}

    def __init__(self, state_dict: Dict[str, torch.Tensor], device=None, bn_eps: float = BN_EPS, precision: str = "f32"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.OaiError("no HIP device: the MI355X path has no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise _lib.OaiError(f"device {self.device} is not a HIP device: the MI355X path has no CPU fallback")
        keep = []          # host tensors must outlive oai_unet_create

        def ptr(key):
            t = state_dict.get(key)
            if t is None:
                return None
            t = t.detach().to("cpu", torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        params = (_lib.LayerParams * 18)()
        known = set()
        digest = hashlib.sha256()      # of the parameters as the kernels see them (fp32, layer order): ties a saved calibration to its network
        for i, name in enumerate(LAYER_ORDER):
            wkey = "dc0.weight" if name == "dc0" else f"{name}.0.weight"
            if wkey not in state_dict:
                raise KeyError(f"state_dict is missing {wkey} (strict load, utils.py:29)")
            w = state_dict[wkey]
            kind = _kind(name)
            cout, cin = (w.shape[0], w.shape[1]) if kind in (0, 3) else (w.shape[1], w.shape[0])
            p = params[i]
            p.kind, p.cin, p.cout = kind, int(cin), int(cout)
            p.weight = ptr(wkey)
            n_before = len(keep)
            if name == "dc0":
                p.bias = ptr("dc0.bias")
                known |= {"dc0.weight", "dc0.bias"}
            else:
                p.bias = ptr(f"{name}.0.bias")
                p.bn_gamma, p.bn_beta = ptr(f"{name}.1.weight"), ptr(f"{name}.1.bias")
                p.bn_mean, p.bn_var = ptr(f"{name}.1.running_mean"), ptr(f"{name}.1.running_var")
                known |= {f"{name}.0.weight", f"{name}.0.bias", f"{name}.1.weight", f"{name}.1.bias",
                          f"{name}.1.running_mean", f"{name}.1.running_var", f"{name}.1.num_batches_tracked"}
            for t in keep[n_before - 1:]:
                digest.update(t.numpy().tobytes())
        self._weights_sha256_r4 = digest.copy().hexdigest() if float(bn_eps) == BN_EPS else None      # what sidecars written before round 5 carry (no bn_eps)
        digest.update(np.float32(bn_eps).tobytes())       # folded into the packed epilogue affine: part of "the network as the kernels see it" 
        self.weights_sha256 = digest.hexdigest()
        extra = set(state_dict) - known
        if extra:
            raise KeyError(f"unexpected keys in state_dict (strict load): {sorted(extra)[:4]}")
        self.n_classes = int(params[17].cout)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_unet_create(params, C.c_float(bn_eps), C.byref(handle)), "oai_unet_create")
        self._h = handle
        self._ws: Optional[torch.Tensor] = None
        self.precision = "f32"
        # fp16x3: per-layer power-of-two activation exponents, chosen from a range census of the first input this engine sees
        # (calibrate()).  False = leave them as they are (all zero unless set_act_exponents was called).
        self.auto_calibrate = True
        self._calibrated = False
        self._fp16_refused = False      # calibration did not settle for this network: "fp16x3" requests run "f32" (with a warning)
        self._no_census_net = False     # no layer of this network records a range census (widths not multiples of 16): nothing to calibrate from
        self.calibration_file: Optional[str] = None     # JSON sidecar (set_calibration_file): read when it exists; written only when calibration_write is set
        self.calibration_write = False                  # write the sidecar after a successful calibrate() -- opt-in (no side-effect writes next to user checkpoints)
        self.calibration_source = "none"                # "none" | "file" | "set" | "calibrated"
        self.calibration_census = None                  # per-layer maxima the exponents were chosen from (None when they were set / read from an old file)
        self._flag_streak = 0                           # consecutive volumes that raised the range flag under a calibration read from a file (note_volume_flag)
        self._dropped_file = None                       # (path, sha256 of its bytes) of a sidecar that note_volume_flag dropped: never read again unless rewritten
        self.calibration_volume_id: Optional[str] = None
        if precision != "f32":
            self.set_precision(precision)

    # ---- what arithmetic this engine REALLY runs (callers label results from this, not from the string they asked for) ----------
    @property
    def effective_precision(self) -> str:
        """The arithmetic the next launch runs: the requested precision, or "f32" after a refused fp16x3 calibration."""
        return self.precision

    @property
    def fp16_refused(self) -> bool:
        """True when this network's activations did not fit the fp16x3 window at any exponents: every "fp16x3" request runs "f32"."""
        return self._fp16_refused

    def calibration_status(self) -> str:
        """ONE source for "is this engine calibrated": "refused_f32" | "no_census" | "calibrated" | "uncalibrated"."""
        if self._fp16_refused:
            return "refused_f32"
        if self._no_census_net:
            return "no_census"
        return "calibrated" if (self._calibrated or self.act_exponents()[1]) else "uncalibrated"

    def refuse_fp16(self, reason: str = "") -> None:
        """Mirror of a failed calibration (another rank's, parallel.sync_calibration): this engine runs "f32" from now on."""
        if not self._fp16_refused:
            warnings.warn(f"fp16x3 refused for this engine ({reason or 'calibration did not settle'}): it runs precision 'f32'")
        self._fp16_refused = True
        self.set_precision("f32")

    def mark_no_census(self) -> None:
        """Mirror of "this network records no census": exponents stay 0, nobody calibrates again."""
        self._no_census_net = True
        self._calibrated = True
        self.calibration_source = "none"

    def note_volume_flag(self, raised: bool, limit: int = 3) -> bool:
        """Per-volume bookkeeping of the callers that repeat a flagged volume in fp32: ``limit`` CONSECUTIVE flagged volumes under a
        calibration that came from a file mean the file does not describe this data (an unrepresentative first volume pinned its
        exponents: every later volume would pay the fp32 repeat, silently, forever -- ADVICE r4).  The calibration is then DROPPED
        (``drop_calibration``: the handle itself becomes uncalibrated, the file is remembered and not read again): a single process
        recalibrates on the next volume it sees, the ranks of a cohort agree on ONE new calibration (parallel.CalibrationBoard).
        Returns True when that happened."""
        if not raised:
            self._flag_streak = 0
            return False
        self._flag_streak += 1
        if self._flag_streak >= limit and self.calibration_source == "file" and not self._fp16_refused:
            warnings.warn(f"{self._flag_streak} consecutive volumes left the range window of the fp16x3 calibration read from "
                          f"{self.calibration_file}: ignoring that file and recalibrating on the next volume")
            self.drop_calibration()
            return True
        return False

    @staticmethod
    def _file_sha256(path: Optional[str]) -> Optional[str]:
        try:
            with open(path, "rb") as f:
                return hashlib.sha256(f.read()).hexdigest()
        except (OSError, TypeError):
            return None

    def drop_calibration(self) -> None:
        """Forget the calibration in the ONE place that holds it -- the library handle (option "calibrated" 0) -- and in this object's
        mirror of it, so that ``calibration_status()``, ``_needs_calibration()`` and ``parallel.sync_calibration`` agree (only the
        Python flag used to be cleared: rank 0 then published the dropped file's exponents as "calibrated").  A calibration that came from a
        file: the file's identity (path + content hash) is remembered and ``set_calibration_file`` does not read it again until it changes."""
        if self.calibration_source == "file" and self.calibration_file:
            self._dropped_file = (self.calibration_file, self._file_sha256(self.calibration_file))
        _lib.check(self.lib.oai_unet_set_option(self._h, b"calibrated", 0), "oai_unet_set_option")
        self._calibrated = False
        self.calibration_source = "none"
        self.calibration_census = None
        self.calibration_volume_id = None
        self._flag_streak = 0

    def set_precision(self, precision: str) -> None:
        """Arithmetic of the 3x3x3 conv layers: "f32" (exact fp32 MFMA), "fp16x3" (fp32-grade split fp16, the default of the
        segmenter), "bf16x6" (fp32-grade split), "bf16x3"."""
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}")
        if precision == "fp16x3" and self._fp16_refused:
            precision = "f32"               # (warned when the calibration failed)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_unet_set_precision(self._h, self.PRECISIONS[precision]), "oai_unet_set_precision")
        self.precision = precision

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.oai_unet_destroy(h)
            self._h = None

    # ------------------------------------------------------------------------------------------
    def _workspace(self, tile_zyx, batch, size_zyx=None, overlap_zyx=None) -> torch.Tensor:
        """Activation workspace for `batch` tiles; with the volume's geometry also the two volume-wide tensors of the shared encoder pass
        (oai_segment_workspace_bytes: without them oai_segment_tiles computes ec0 -> ec1 per tile; same results)."""
        if size_zyx is not None:
            need = int(self.lib.oai_segment_workspace_bytes(self._h, *[int(v) for v in size_zyx], _lib.int3(tile_zyx), _lib.int3(overlap_zyx), int(batch)))
        else:
            need = int(self.lib.oai_unet_workspace_bytes(self._h, *[int(v) for v in tile_zyx], int(batch)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def auto_batch(self, tile_zyx, n_tiles: int, hbm_fraction: float = 0.6) -> int:
        """Tiles per U-Net pass when the caller does not fix it: all of them if the activation workspace fits
        ``hbm_fraction`` of the free HBM (160 tiles of 32x128x128 need 148 GiB of the 288 GB), else halved until it does.
        Fewer, larger launches fill the 256 CUs at the deep levels and cut launch tails: 4.69 -> 4.92 volumes/s from 32 to 160."""
        with torch.cuda.device(self.device):
            free, _ = torch.cuda.mem_get_info()
        held = self._ws.numel() if self._ws is not None else 0
        n = max(1, int(n_tiles))
        while n > 1 and int(self.lib.oai_unet_workspace_bytes(self._h, *[int(v) for v in tile_zyx], n)) > hbm_fraction * (free + held):
            n = (n + 1) // 2
        return n

    def set_option(self, name: str, value: int) -> None:
        """Tuning options of the fp16x3 path (include/oai_hip.h: oai_unet_set_option).  Bit-preserving (same k order, same maps):
        "sres", "sres_mrep", "sres_ring", "xcd_group", "fuse_first", "b_lds", "wide", "shared_enc", "dead_stores", "census", "up_nbw", "first_blocks".
        NOT bit-preserving: "winograd" (bit mask, default 19: the x axis of ten layers in Winograd F(2,3) form, bit 4 = 16x16x32 tap pairs -- other rounding,
        same parity gates; 0 = the direct form everywhere), "winograd_layers" (which layers), "m16" (default 1: the direct kernel of the layers with
        Cout % 128 != 0 on 16x16x32 tap pairs), "m16_layers" and "persistent" (default 0: dc2 with persistent workgroups, bit-identical, not faster); of the exact-fp32 path: "winograd_f32" (default 1: its
        k3 layers in x-axis Winograd form on exact fp32 products -- 2/3 of the fp32 MFMAs and closer to float64 than the direct form 0)."""
        _lib.check(self.lib.oai_unet_set_option(self._h, name.encode(), int(value)), "oai_unet_set_option")
        if name == "sres":
            # sres 0 = the superseded kernels that keep fp32 activations in memory and split them while staging: no range census, no
            # activation exponents, no LOW bit -- a comparison mode for tests and A/B timing, not a production setting
            self._no_census = int(value) == 0
            if self._no_census:
                warnings.warn("option sres=0: fp16x3 without the split-resident kernels has no range census (no calibration, no LOW flag)")

    def range_flag(self, reset: bool = True) -> int:
        """fp16x3 only: the range flag of the work queued since the last reset -- bit 0: an activation beyond fp16's range, bit 1: a
        layer whose largest stored activation is below the calibrated window (include/oai_hip.h).  Non-zero = repeat that run in
        "f32".  Ordered on the current stream of this engine's device; synchronises that stream."""
        out = C.c_int(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_unet_range_flag(self._h, int(reset), C.byref(out), torch.cuda.current_stream().cuda_stream),
                       "oai_unet_range_flag")
        return int(out.value)

    def range_overflow(self, reset: bool = True) -> bool:
        """True if the fp16x3 results queued since the last reset must not be used (range_flag() != 0)."""
        return self.range_flag(reset) != 0

    # ---- activation exponents of fp16x3 ---------------------------------------------------------------------------------------
    def census(self, reset: bool = False):
        """max |stored activation| of each of the 18 layers since the last reset (0.0 = nothing stored); synchronises the stream."""
        out = (C.c_float * 18)()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_unet_census(self._h, out, int(reset), torch.cuda.current_stream().cuda_stream), "oai_unet_census")
        return list(out)

    def act_exponents(self):
        """(exponents of the 18 layers, calibrated?)"""
        e, cal = (C.c_int * 18)(), C.c_int(0)
        _lib.check(self.lib.oai_unet_get_act_exponents(self._h, e, C.byref(cal)), "oai_unet_get_act_exponents")
        return list(e), bool(cal.value)

    def set_act_exponents(self, exponents) -> None:
        """Explicit exponents (e.g. those of another rank, or saved next to a checkpoint); marks the engine calibrated."""
        if len(exponents) != 18:
            raise ValueError("need one exponent per layer (18)")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_unet_set_act_exponents(self._h, (C.c_int * 18)(*[int(v) for v in exponents])), "oai_unet_set_act_exponents")
        self._calibrated = True
        self.calibration_source = "set"

    # ---- a calibration belongs to a checkpoint, not to the first volume a process happens to see -----------
    def save_calibration(self, path: str, note: str = "", volume_id: Optional[str] = None) -> None:
        """JSON sidecar: the 18 exponents + the sha256 of the network's parameters (and bn_eps) + what they were chosen from -- the
        per-layer maxima of the calibration census and, when the caller names it, the identity of the calibration volume.  Written
        atomically (temp file + rename): ranks that calibrate at the same time leave one complete file, whichever wins."""
        exps, cal = self.act_exponents()
        if not cal:
            raise _lib.OaiError("save_calibration: the engine is not calibrated")
        doc = {"format": 1, "precision": "fp16x3", "weights_sha256": self.weights_sha256, "act_exponents": exps, "note": note,
               "census_max": self.calibration_census, "volume_id": volume_id or self.calibration_volume_id}
        tmp = f"{path}.tmp.{os.getpid()}"
        with open(tmp, "w") as f:
            json.dump(doc, f)
        os.replace(tmp, path)

    def load_calibration(self, path: str) -> bool:
        """True if ``path`` holds exponents for THIS network (sha256 of the parameters) and they are now set; a file for another
        network, an unreadable or malformed file is reported and ignored (the engine then calibrates as if there were none)."""
        if not os.path.isfile(path):
            return False
        try:
            with open(path) as f:
                doc = json.load(f)
            exps = [int(v) for v in doc["act_exponents"]]
            if doc.get("format") != 1 or len(exps) != 18:
                raise ValueError("unknown layout")
        except Exception as exc:          # noqa: BLE001 - any damage to the sidecar means "no calibration on file"
            warnings.warn(f"fp16x3 calibration file {path} is unreadable ({exc}): ignoring it")
            return False
        if doc.get("weights_sha256") != self.weights_sha256:
            if self._weights_sha256_r4 is not None and doc.get("weights_sha256") == self._weights_sha256_r4:
                pass          # a sidecar from before bn_eps was part of the digest, for these weights at the default eps: the same network
            else:
                warnings.warn(f"fp16x3 calibration file {path} belongs to other weights (or another bn_eps): ignoring it -- "
                              f"re-run Segmenter3DInPatchClassWise.calibrate(image) to rewrite it")
                return False
        try:
            self.set_act_exponents(exps)           # (the library checks the range: an edited file with wild exponents is refused there)
        except _lib.OaiError as exc:
            warnings.warn(f"fp16x3 calibration file {path} holds exponents the library refuses ({exc}): ignoring it")
            return False
        self.calibration_source = "file"
        self.calibration_census = doc.get("census_max")
        self._flag_streak = 0
        return True

    def set_calibration_file(self, path: Optional[str], write: bool = False) -> bool:
        """Use ``path`` as this engine's calibration sidecar: read it now if it exists (returns True when its exponents were taken).
        ``write=True`` (opt-in): also write it after the next successful calibrate().  None detaches."""
        self.calibration_file = path
        self.calibration_write = bool(path) and bool(write)
        if not path or self._calibrated:
            return False
        if self._dropped_file is not None and self._dropped_file[0] == path and self._dropped_file[1] == self._file_sha256(path):
            return False                 # the very file this engine dropped for not fitting the data: not again (a rewritten one is read)
        return self.load_calibration(path)

    def calibrate(self, run_pass, max_passes: int = 24) -> int:
        """Choose the activation exponents from representative input: ``run_pass()`` queues one fp16x3 pass (segment_tiles /
        forward_tiles) on the current stream; repeated until every layer's maximum sits in the calibrated window (two passes for
        a network whose activations fit fp16 to begin with; one more per layer that overflowed on the way).  Returns the passes."""
        if self.precision != "fp16x3":
            raise _lib.OaiError("activation exponents belong to precision 'fp16x3'")
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream().cuda_stream
            self.census(reset=True)
            for n in range(1, max_passes + 1):
                run_pass()
                more = C.c_int(0)
                seen = [float(v) for v in self.census(reset=False)]        # (what this pass stored: the sidecar records the settled pass's maxima)
                rc = self.lib.oai_unet_calibrate_step(self._h, st, C.byref(more))
                if rc != 0 and n == 1 and b"census is empty" in (self.lib.oai_last_error() or b""):
                    # no layer of this network runs through a census-recording kernel (widths that are not multiples of 16: test networks
                    # only): there is nothing to calibrate from and no LOW bit; the exponents stay 0, as before the calibration existed
                    warnings.warn("this network records no range census (layer widths not multiples of 16): fp16x3 runs with activation exponents 0")
                    self.mark_no_census()
                    return 0
                _lib.check(rc, "oai_unet_calibrate_step")
                if not more.value:
                    self._calibrated = True
                    self.calibration_source = "calibrated"
                    self._flag_streak = 0
                    self.calibration_census = seen
                    if self.calibration_file and self.calibration_write:
                        try:
                            self.save_calibration(self.calibration_file, note=f"calibrated in {n} passes")
                        except OSError as exc:          # a read-only model directory is not an error: the next process calibrates again
                            warnings.warn(f"could not write the fp16x3 calibration file {self.calibration_file}: {exc}")
                    return n
        # not settled: this network's activations do not fit the fp16x3 window at any exponents (a layer whose range spans more than
        # the window, non-finite values, ...).  Before the calibration existed such a checkpoint ran through the fp32 repeat; keep that
        # behaviour instead of raising: every later "fp16x3" request of this engine runs exact fp32.
        self.refuse_fp16(f"calibration did not settle in {max_passes} passes, census {self.census()}")
        return -1

    def _needs_calibration(self) -> bool:
        return self.precision == "fp16x3" and self.auto_calibrate and not self._calibrated and not getattr(self, "_no_census", False)

    def range_overflow_snapshot(self, dst: torch.Tensor) -> None:
        """Queue (current stream, no sync) a copy of the fp16 range flag into the int32 device tensor ``dst[0]`` and clear it:
        the flag of the segment calls queued so far, read later together with their results (cohort.py)."""
        if dst.dtype != torch.int32 or dst.device != self.device or dst.numel() < 1:
            raise ValueError("dst must be an int32 tensor on the engine's device")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_unet_range_flag_snapshot(self._h, dst.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       "oai_unet_range_flag_snapshot")

    RANGE_STATE_WORDS = 19          # OAI_UNET_RANGE_STATE_WORDS

    def range_state_snapshot(self, dst: torch.Tensor) -> None:
        """Queue (no sync) the RAW range state of the work queued so far into the int32 device tensor ``dst[0:19]`` and clear it:
        [0] overflow bit, [1 + k] float bits of layer k's largest stored activation.  MAX-all-reduce it over the ranks that shared a
        volume's tiles, then ``range_flag_from_state``: the flag of the whole volume, not of one rank's subset."""
        if dst.dtype != torch.int32 or dst.device != self.device or dst.numel() < self.RANGE_STATE_WORDS:
            raise ValueError("dst must be an int32 tensor of 19 words on the engine's device")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_unet_range_state_snapshot(self._h, dst.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       "oai_unet_range_state_snapshot")

    def range_flag_from_state(self, state: torch.Tensor) -> torch.Tensor:
        """int32[1] device tensor: the two-bit range flag of a (reduced) range state; queued on the current stream, no sync."""
        flag = torch.empty(1, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_unet_range_flag_from_state(state.data_ptr(), flag.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       "oai_unet_range_flag_from_state")
        return flag

    def tile_flops(self, tile_zyx, overlap_zyx, trimmed: bool) -> float:
        return float(self.lib.oai_unet_tile_flops(self._h, *[int(v) for v in tile_zyx], _lib.int3(overlap_zyx), int(trimmed)))

    def tile_flops_conv3(self, tile_zyx, overlap_zyx, trimmed: bool) -> float:
        return float(self.lib.oai_unet_tile_flops_conv3(self._h, *[int(v) for v in tile_zyx], _lib.int3(overlap_zyx), int(trimmed)))

    def profile(self, enable: bool) -> None:
        _lib.check(self.lib.oai_unet_profile(self._h, int(enable)), "oai_unet_profile")

    def profile_read(self):
        """(summed ms, launches) of the 3x3x3 implicit-GEMM kernel since the last read (HIP events on its stream)."""
        ms, n = C.c_double(), C.c_longlong()
        _lib.check(self.lib.oai_unet_profile_read(self._h, C.byref(ms), C.byref(n)), "oai_unet_profile_read")
        return ms.value, n.value

    def forward_tiles(self, tiles: torch.Tensor, batch: Optional[int] = None) -> torch.Tensor:
        """logits[B,n_classes,d,h,w] = model(tiles[B,1,d,h,w]) -- the ``self.model(...)`` of segmenter.py:116."""
        if tiles.dim() != 5 or tiles.shape[1] != 1:
            raise ValueError("tiles must be [B,1,D,H,W]")
        tiles = tiles.to(self.device, torch.float32).contiguous()
        B, _, d, h, w = tiles.shape
        nb = min(B, batch or self.auto_batch((d, h, w), B))
        out = torch.empty((B, self.n_classes, d, h, w), dtype=torch.float32, device=self.device)

        def launch():
            ws = self._workspace((d, h, w), nb)      # (sized per launch: a refused calibration switches the arithmetic in between; the C side loops over what the workspace holds)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.oai_unet_forward_tiles(self._h, tiles.data_ptr(), out.data_ptr(), B, d, h, w,
                                                           ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                           "oai_unet_forward_tiles")
        if self._needs_calibration():
            self.calibrate(launch)
        launch()
        return out

    def volume_flops(self, size_zyx, tile_zyx, overlap_zyx, crop_zyx=None, trimmed: bool = True, conv3_only: bool = False) -> float:
        return float(self.lib.oai_unet_volume_flops(self._h, *[int(v) for v in size_zyx], _lib.int3(tile_zyx), _lib.int3(overlap_zyx),
                                                    _lib.int3(crop_zyx) if crop_zyx is not None else None, int(trimmed), int(conv3_only)))

    def tile_costs(self, size_zyx, tile_zyx, overlap_zyx, crop_zyx=None):
        """FLOPs of each tile as ``segment_tiles`` computes it (list of floats, the reference's z-major tile order)."""
        _, _, n = tile_grid(size_zyx, tile_zyx, overlap_zyx)
        out = (C.c_double * n)()
        _lib.check(self.lib.oai_unet_tile_costs(self._h, *[int(v) for v in size_zyx], _lib.int3(tile_zyx), _lib.int3(overlap_zyx),
                                                _lib.int3(crop_zyx) if crop_zyx is not None else None, out, n), "oai_unet_tile_costs")
        return list(out)

    def segment_tiles(self, vol: torch.Tensor, tile_zyx, overlap_zyx, tile_range: Optional[Tuple[int, int]] = None,
                      out_mode: int = 0, batch: Optional[int] = None, crop_zyx=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Kept-centre blocks [n_local, n_classes, ez, ey, ex] of tiles [begin,end) of the volume.

        ``crop_zyx``: the frame ``stitch`` will zero; block voxels inside it (and beyond the image) are not computed
        and hold unspecified values.  ``out``: write the blocks there (a contiguous fp32 tensor of exactly that shape on this device --
        e.g. this rank's slot of an all_gather buffer, parallel.alloc_gather) instead of allocating."""
        vol = vol.to(self.device, torch.float32).contiguous()
        D, H, W = vol.shape
        eff, grid, ntiles = tile_grid((D, H, W), tile_zyx, overlap_zyx)
        begin, end = tile_range if tile_range is not None else (0, ntiles)
        if not batch:
            batch = self.auto_batch(tile_zyx, end - begin)
        batch = max(1, min(int(batch), max(1, end - begin)))
        self.last_batch = batch
        if self._needs_calibration():
            # on the WHOLE volume whatever range was asked for: every rank of a tile-sharded volume arrives at the same exponents
            self.calibrate_volume(vol, tile_zyx, overlap_zyx, crop_zyx, batch=None if (begin, end) != (0, ntiles) else batch)
        ws = self._workspace(tile_zyx, batch, (D, H, W), overlap_zyx)
        if out is not None:
            if tuple(out.shape) != (end - begin, self.n_classes, *eff) or out.dtype != torch.float32 or out.device != self.device or not out.is_contiguous():
                raise ValueError("out must be a contiguous float32 tensor [n_local, n_classes, ez, ey, ex] on the engine's device")
            blocks = out
        else:
            blocks = torch.empty((end - begin, self.n_classes, *eff), dtype=torch.float32, device=self.device)
        if end > begin:
            self._launch_segment(vol, tile_zyx, overlap_zyx, crop_zyx, begin, end, out_mode, blocks, batch, ws)
        return blocks

    def _launch_segment(self, vol, tile_zyx, overlap_zyx, crop_zyx, begin, end, out_mode, blocks, batch, ws) -> None:
        D, H, W = vol.shape
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_segment_tiles(self._h, vol.data_ptr(), D, H, W, _lib.int3(tile_zyx), _lib.int3(overlap_zyx),
                                                  _lib.int3(crop_zyx) if crop_zyx is not None else None,
                                                  int(begin), int(end), int(out_mode), blocks.data_ptr(), batch,
                                                  ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                       "oai_segment_tiles")

    def calibrate_volume(self, vol: torch.Tensor, tile_zyx, overlap_zyx, crop_zyx=None, batch: Optional[int] = None) -> int:
        """calibrate() on all tiles of ``vol`` (the census is a maximum: the result does not depend on batching or tile order)."""
        vol = vol.to(self.device, torch.float32).contiguous()
        eff, grid, ntiles = tile_grid(vol.shape, tile_zyx, overlap_zyx)
        batch = max(1, min(int(batch or self.auto_batch(tile_zyx, ntiles)), ntiles))
        ws = self._workspace(tile_zyx, batch, vol.shape, overlap_zyx)
        blocks = torch.empty((ntiles, self.n_classes, *eff), dtype=torch.float32, device=self.device)
        return self.calibrate(lambda: self._launch_segment(vol, tile_zyx, overlap_zyx, crop_zyx, 0, ntiles, 0, blocks, batch, ws))

    def stitch(self, blocks, size_zyx, tile_zyx, overlap_zyx, crop_zyx=None) -> torch.Tensor:
        """maps[n_classes, D, H, W] (Partition.assemble, non-vote branch).  ``blocks``: the [n_tiles, n_classes, ez, ey, ex] tensor, or
        the ``parallel.GatheredBlocks`` an all_gather of per-rank tile ranges left behind (read in place, oai_stitch_blocks_ranged)."""
        D, H, W = (int(v) for v in size_zyx)
        eff, grid, ntiles = tile_grid((D, H, W), tile_zyx, overlap_zyx)
        maps = torch.empty((self.n_classes, D, H, W), dtype=torch.float32, device=self.device)
        crop = _lib.int3(crop_zyx) if crop_zyx is not None else None
        if hasattr(blocks, "bounds") and hasattr(blocks, "buffer"):          # GatheredBlocks
            g = blocks
            if g.n_tiles != ntiles:
                raise ValueError(f"stitch needs the blocks of all {ntiles} tiles, the gather holds {g.n_tiles}")
            buf = g.buffer
            if not buf.is_contiguous() or buf.device != self.device or buf.dtype != torch.float32:
                raise ValueError("the gather buffer must be a contiguous float32 tensor on the engine's device")
            bounds = (C.c_int * len(g.bounds))(*g.bounds)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.oai_stitch_blocks_ranged(buf.data_ptr(), self.n_classes, D, H, W, _lib.int3(tile_zyx), _lib.int3(overlap_zyx),
                                                             crop, bounds, len(g.bounds) - 1, g.stride, maps.data_ptr(),
                                                             torch.cuda.current_stream().cuda_stream), "oai_stitch_blocks_ranged")
            return maps
        if blocks.shape[0] != ntiles:
            raise ValueError(f"stitch needs the blocks of all {ntiles} tiles, got {blocks.shape[0]}")
        blocks = blocks.contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.oai_stitch_blocks(blocks.data_ptr(), self.n_classes, D, H, W, _lib.int3(tile_zyx),
                                                  _lib.int3(overlap_zyx), crop, maps.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "oai_stitch_blocks")
        return maps

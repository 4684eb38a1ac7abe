"""Host-side mirror of the reference's model interface for the MI355X path.

Reference interface being mirrored:
  * models/seg_hrnet.py:495-499     get_seg_model(cfg, **kwargs) -> nn.Module
  * models/seg_hrnet.py:258-340     HighResolutionNet(config): parameters/buffers under the
                                    reference's state_dict keys (strict load_state_dict of a
                                    reference checkpoint works, val.py:64-66)
  * models/seg_hrnet.py:425-473     net(x: f32 [N,Cin,H,W]) -> f32 [N,K,H,W], same device
  * models/seg_hrnet.py:475-493     init_weights(pretrained)

The module owns ordinary torch parameters (so .cuda(), .parameters(), .state_dict(),
DataParallel's unwrap idiom `net.module.net` all behave), but forward() never runs a torch
operator: it folds BN into the convolutions once per weight version, hands them to
libesahrnet.so and enqueues the hand-written HIP kernels on torch's current stream.
There is no CPU or eager fallback: without the library or without a GPU, forward raises.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import threading

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .fold import fold_conv

logger = logging.getLogger(__name__)
BN_MOMENTUM = 0.01           # models/seg_hrnet.py:23 (irrelevant at inference, kept for parity)


# "fp32" is the fp32-grade bf16x6 mode (the reference's arithmetic class, BASELINE configs[1]); "bf16x3" (split-bf16,
# ~16 significand bits) is an explicitly named opt-in and NOT an alias of fp32
PRECISIONS = {"fp32": 2, "bf16x6": 2, "bf16x3": 0, "split-bf16": 0, "bf16": 1, 0: 0, 1: 1, 2: 2}


def _cfg_struct(config, cin: int, num_keypoints: int, variant: int = 0, precision=0) -> _lib.Cfg:
    extra = config.MODEL.EXTRA.HIGH_RESOLUTION_NET if hasattr(config, "MODEL") else \
        config["MODEL"]["EXTRA"]["HIGH_RESOLUTION_NET"]
    s = _lib.Cfg()
    s.cin, s.num_keypoints, s.stem_width, s.variant = cin, num_keypoints, 64, variant
    if precision not in PRECISIONS:
        raise ValueError(f"precision={precision!r}: expected one of {sorted(map(str, PRECISIONS))}")
    s.precision = PRECISIONS[precision]
    fk = extra["FINAL_CONV_KERNEL"] if "FINAL_CONV_KERNEL" in extra else 1
    s.final_conv_kernel = int(fk)
    for i in range(4):
        st = extra[f"STAGE{i + 1}"]
        if st["BLOCK"] != "BASIC":
            raise ValueError(f"STAGE{i + 1}.BLOCK={st['BLOCK']!r}: only BASIC blocks are built "
                             "(config/default.py:49-73 uses BASIC everywhere)")
        if st["FUSE_METHOD"] != "SUM":
            raise ValueError("only FUSE_METHOD='SUM' exists in the reference")
        nb = int(st["NUM_BRANCHES"])
        # same consistency checks as HighResolutionModule._check_branches (seg_hrnet.py:123-141)
        if nb != len(st["NUM_BLOCKS"]):
            raise ValueError("NUM_BRANCHES({}) <> NUM_BLOCKS({})".format(nb, len(st["NUM_BLOCKS"])))
        if nb != len(st["NUM_CHANNELS"]):
            raise ValueError("NUM_BRANCHES({}) <> NUM_CHANNELS({})".format(nb, len(st["NUM_CHANNELS"])))
        if nb != i + 1:
            raise ValueError(f"STAGE{i + 1} must have {i + 1} branches (got {nb})")
        s.modules[i] = int(st["NUM_MODULES"])
        for b in range(nb):
            s.blocks[i][b] = int(st["NUM_BLOCKS"][b])
            if i == 3:
                s.widths[b] = int(st["NUM_CHANNELS"][b])
            elif int(st["NUM_CHANNELS"][b]) != int(extra["STAGE4"]["NUM_CHANNELS"][b]):
                raise ValueError("branch widths must agree across stages")
    return s


class _Node(nn.Module):
    """Pure container used to reproduce the reference's dotted state_dict names."""


def _place(root: nn.Module, dotted: str, leaf: nn.Module):
    parts = dotted.split(".")
    cur = root
    for p in parts[:-1]:
        nxt = cur._modules.get(p)
        if nxt is None:
            nxt = _Node()
            cur.add_module(p, nxt)
        cur = nxt
    cur.add_module(parts[-1], leaf)


class HighResolutionNet(nn.Module):
    CIN = 3                  # models/seg_hrnet.py:265
    NUM_KEYPOINTS = 32       # models/seg_hrnet.py:324
    VARIANT = 0              # 1 = seg_hrnet3.py (CBAM)
    DEFAULT_PRECISION = "fp32"   # the reference computes in fp32 (models/seg_hrnet.py:425-473): fp32-grade bf16x6 by default

    def __init__(self, config, **kwargs):
        super().__init__()
        cin = int(kwargs.pop("cin", self.CIN))
        k = int(kwargs.pop("num_keypoints", self.NUM_KEYPOINTS))
        self._cin, self._k = cin, k
        # precision (include/esahrnet.h esahrnet_cfg.precision): "fp32" = "bf16x6" (default: fp32-grade, BASELINE
        # configs[1] / [2]), "bf16x3" (split-bf16, ~16 significand bits: explicit opt-in, ~1.7x faster), "bf16"
        # (single-pass bf16 storage / fp32 accumulate, BASELINE configs[3])
        self._cfg_struct = _cfg_struct(config, cin, k, int(kwargs.pop("variant", self.VARIANT)),
                                       kwargs.pop("precision", self.DEFAULT_PRECISION))
        object.__setattr__(self, "_rt", _Runtime(self._cfg_struct))
        self._descs = self._rt.conv_descs()
        for d in self._descs:
            conv = nn.Conv2d(d["cin"], d["cout"], d["k"], d["stride"], (d["k"] - 1) // 2, bias=d["has_bias"])
            _place(self, d["name"], conv)
            if d["bn"]:
                _place(self, d["bn"], nn.BatchNorm2d(d["cout"], momentum=BN_MOMENTUM))
        # parameters that are not convolutions of the main graph (seg_hrnet3: ChannelAttention.fc,
        # SpatialAttention.conv1); some names alias a conv that already exists (conv1.weight)
        self._aux = self._rt.aux_descs()
        have = set(dict(self.named_parameters()).keys())
        for a in self._aux:
            if a["name"] in have:
                continue
            co, ci, kh, kw = a["shape"]
            _place(self, a["name"][: -len(".weight")], nn.Conv2d(ci, co, (kh, kw), padding=(kh // 2, kw // 2), bias=False))
        self.eval()

    # ---- reference API -------------------------------------------------------------------------
    def init_weights(self, pretrained=""):
        logger.info("=> init weights from normal distribution")
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.001)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if os.path.isfile(pretrained):
            # weights_only: a checkpoint is data, nothing in it is executed
            pretrained_dict = torch.load(pretrained, map_location="cpu", weights_only=True)
            logger.info("=> loading pretrained model {}".format(pretrained))
            model_dict = self.state_dict()
            model_dict.update({k: v for k, v in pretrained_dict.items() if k in model_dict})
            self.load_state_dict(model_dict)

    def forward(self, x0: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise RuntimeError("HighResolutionNet (MI355X path) is inference-only: call .eval() "
                               "(the reference callers do, val.py:95 / demo.py:80)")
        return self._rt.forward(self, x0)

    # ---- extras of the MI355X path ---------------------------------------------------------------
    @property
    def num_keypoints(self):
        return self._k

    def flops_per_crop(self, h: int, w: int) -> float:
        return self._rt.flops_per_crop(h, w)

    def launch_count(self) -> int:
        return self._rt.launch_count()

    def forward_timed(self, x0: torch.Tensor):
        """Measurement: one forward with every launch bracketed by HIP events on the current
        stream -> (heatmaps, [dict(kernel, label, ms, flops, bytes)] per launch)."""
        return self._rt.forward_timed(self, x0)

    def taps(self, x0: torch.Tensor) -> dict:
        """Debug: run a forward keeping every intermediate; returns {name: f32 NCHW tensor}."""
        return self._rt.taps(self, x0)

    # ---- weight-version tracking (an eager forward must not walk the module tree: val.py:112 calls the
    # net once per image) ---------------------------------------------------------------------------
    def _weight_tensors(self):
        """Flat list of every parameter and buffer, rebuilt only after the module was converted
        (`_apply`: .cuda()/.to()/.float()) or re-loaded."""
        ts = self.__dict__.get("_wt_cache")
        if ts is None:
            ts = [t for t in self.state_dict(keep_vars=True).values()]
            self.__dict__["_wt_cache"] = ts
        return ts

    def _weights_key(self):
        """What the folded/packed weights on a device are valid for: "the weights are what the Parameters
        say" (the reference's semantics).  load_state_dict, init_weights and every conversion
        (.cuda()/.to()/.float()) bump the epoch through the hooks below; an in-place edit of ANY parameter or
        buffer (under no_grad, by an optimizer step, ...) is seen through that tensor's autograd version
        counter: the key holds all of them (~40 us per forward for 539 tensors, 1.5 % of a batch-32 step).
        Only writes through `.data` / `.detach()` aliases are invisible to version counters: call
        invalidate_weights() after those.  A caller that runs the net once per image (val.py:112) and never
        touches the weights can drop the walk with freeze_weights()."""
        ts = self._weight_tensors()
        if self.__dict__.get("_wt_frozen", False):
            return (self.__dict__.get("_wt_epoch", 0),)
        return (self.__dict__.get("_wt_epoch", 0), *[t._version for t in ts])

    def freeze_weights(self, frozen: bool = True):
        """Opt-in fast path for per-image loops: promise that no parameter or buffer is edited in place until
        freeze_weights(False) / invalidate_weights() / load_state_dict() / a conversion; the forward then checks
        the epoch only (1 us instead of 40 us of host time)."""
        self.invalidate_weights()               # whatever was edited before the promise is folded in once
        self.__dict__["_wt_frozen"] = bool(frozen)
        return self

    def invalidate_weights(self):
        """Force a re-fold on the next forward (see _weights_key)."""
        self.__dict__["_wt_cache"] = None
        self.__dict__["_wt_epoch"] = self.__dict__.get("_wt_epoch", 0) + 1

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self.invalidate_weights()
        return r

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self.invalidate_weights()
        return r

    def release_workspaces(self):
        """Drop the cached scratch tensors of eager forwards (HIP graphs own theirs, see _Runtime)."""
        self._rt.release_workspaces()

    def _replicate_for_data_parallel(self):
        # nn.DataParallel (val.py:382, main.py:254): a replica has no Parameters of its own (torch re-attaches
        # broadcast copies as plain attributes), so it keeps a reference to the module it was made from and
        # the shared runtime folds THAT module's weights once per device.
        r = super()._replicate_for_data_parallel()
        object.__setattr__(r, "_rt", self._rt)
        r.__dict__["_master"] = self.__dict__.get("_master", self)
        return r


class _Runtime:
    """One libesahrnet handle per device + caller-owned workspace tensors.

    Shared by a module and its DataParallel replicas (one Python thread per device): everything that
    mutates the tables below happens under `self.lock`; the launches themselves run outside it, one
    handle per device, so replicas do not serialise each other."""

    WS_SHAPES_PER_DEVICE = 4     # eager scratch tensors kept per device (LRU)

    def __init__(self, cfg_struct):
        self.cfg = cfg_struct
        self.lib = _lib.lib()
        self.lock = threading.RLock()
        self.handles = {}        # device index -> (handle, weight-version key)
        self.dev_locks = {}      # device index -> lock serialising the enqueues of that device's handle
        self.ws = {}             # (device, stream, n, h, w, keep) -> uint8 tensor, insertion order = LRU order
        self.part_tiles = {}     # (handle, h, w) -> tiles per heat-map with partial maxima (0: none)
        self._probe = self._create(-1)

    def _create(self, device):
        h = C.c_void_p()
        _lib.check(self.lib.esahrnet_create(C.byref(self.cfg), max(device, 0), C.byref(h)))
        return h

    def __del__(self):
        try:
            for h, _ in self.handles.values():
                self.lib.esahrnet_destroy(h)
            self.lib.esahrnet_destroy(self._probe)
        except Exception:
            pass

    def aux_descs(self):
        out = []
        for i in range(self.lib.esahrnet_aux_count(self._probe)):
            d = _lib.AuxDesc()
            _lib.check(self.lib.esahrnet_aux_desc_get(self._probe, i, C.byref(d)))
            out.append(dict(name=d.name.decode(), shape=tuple(d.shape)))
        return out

    def conv_descs(self):
        out = []
        for i in range(self.lib.esahrnet_conv_count(self._probe)):
            d = _lib.ConvDesc()
            _lib.check(self.lib.esahrnet_conv_desc_get(self._probe, i, C.byref(d)))
            out.append(dict(name=d.name.decode(), bn=d.bn.decode(), cin=d.cin, cout=d.cout, k=d.k,
                            stride=d.stride, has_bias=bool(d.has_bias), relu=bool(d.relu)))
        return out

    def flops_per_crop(self, h, w):
        f = C.c_double()
        _lib.check(self.lib.esahrnet_flops_per_crop(self._probe, h, w, C.byref(f)))
        return f.value

    def launch_count(self):
        return self.lib.esahrnet_launch_count(self._probe)

    def _handle_for(self, module, device):
        master = module.__dict__.get("_master", module)     # a DataParallel replica folds its master's weights
        key = master._weights_key()
        ent = self.handles.get(device.index)
        if ent is not None and ent[1] == key:
            return ent[0]
        with self.lock:
            ent = self.handles.get(device.index)
            if ent is not None and ent[1] == key:
                return ent[0]
            h = ent[0] if ent is not None else self._create(device.index)
            sd = master.state_dict()
            for i, d in enumerate(master._descs):
                w, b = fold_conv(sd, d["name"], d["bn"], d["has_bias"])
                _lib.check(self.lib.esahrnet_set_conv(h, i, w.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))
            for i, a in enumerate(master._aux):
                w = np.ascontiguousarray(sd[a["name"]].detach().cpu().float().numpy())
                _lib.check(self.lib.esahrnet_set_aux(h, i, w.ctypes.data_as(C.c_void_p)))
            with torch.cuda.device(device):
                _lib.check(self.lib.esahrnet_commit(h))
            self.handles[device.index] = (h, key)
            return h

    def _check_input(self, module, x0):
        if not isinstance(x0, torch.Tensor) or x0.dim() != 4:
            raise ValueError("expected a 4-D tensor [N, Cin, H, W]")
        if not x0.is_cuda:
            raise RuntimeError("the MI355X HRNet path runs only on a GPU tensor; there is no CPU "
                               "fallback (move the model and the crops to 'cuda')")
        if x0.dtype != torch.float32:
            raise TypeError(f"expected float32 crops, got {x0.dtype}")
        if x0.shape[1] != module._cin:
            raise ValueError(f"expected {module._cin} input channels, got {x0.shape[1]}")
        if "_master" not in module.__dict__:     # (a replica runs wherever DataParallel scattered its input)
            p = module._weight_tensors()[0]
            if p.device != x0.device:
                raise RuntimeError(f"input on {x0.device} but parameters on {p.device}")
        return x0.contiguous()

    def release_workspaces(self):
        with self.lock:
            self.ws.clear()

    def _workspace(self, h, device, stream, n, hh, ww, keep):
        """Scratch for one forward.  Contract: while the stream is being CAPTURED into a HIP
        graph the scratch is a fresh tensor allocated inside the capture (the graph's private pool owns it, like
        any temporary of a captured torch op) and is never cached, so no graph ever holds a pointer into the
        eager cache; eager forwards share a small per-device LRU of scratch tensors, keyed by stream and shape
        (two streams never share scratch) and protected by record_stream."""
        nbytes = C.c_size_t()
        _lib.check(self.lib.esahrnet_workspace_bytes(h, n, hh, ww, C.byref(nbytes)))
        if torch.cuda.is_current_stream_capturing():
            ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=device)
        else:
            key = (device.index, stream.cuda_stream, n, hh, ww, keep)
            with self.lock:
                ws = self.ws.pop(key, None)
                if ws is None or ws.numel() < nbytes.value + 256:
                    ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=device)
                self.ws[key] = ws                                    # most recently used last
                mine = [k for k in self.ws if k[0] == device.index]
                for k in mine[: max(0, len(mine) - self.WS_SHAPES_PER_DEVICE)]:
                    del self.ws[k]
        off = (-ws.data_ptr()) % 256
        return ws, ws.data_ptr() + off, nbytes.value

    def _device_lock(self, index):
        lk = self.dev_locks.get(index)
        if lk is None:
            with self.lock:
                lk = self.dev_locks.setdefault(index, threading.RLock())
        return lk

    def forward(self, module, x0, keep=False):
        x = self._check_input(module, x0)
        n, _, hh, ww = x.shape
        dev = x.device
        ts = torch.cuda.current_stream(dev)
        # calls on one handle are not re-entrant (include/esahrnet.h): threads that share a device enqueue one
        # after the other; threads on different devices (DataParallel's replicas) do not wait for each other
        with self._device_lock(dev.index):
            h = self._handle_for(module, dev)
            _lib.check(self.lib.esahrnet_set_debug_keep(h, 1 if keep else 0))
            ws, ws_ptr, ws_bytes = self._workspace(h, dev, ts, n, hh, ww, keep)
            heat = torch.empty((n, module._k, hh, ww), dtype=torch.float32, device=dev)
            # per-tile maxima beside the heat-maps (include/esahrnet.h: esahrnet_forward_partials): 8 bytes per plane
            # and 16x16 tile, so that inference.heatmaps_to_keypoints need not sweep the maps again
            nt = self._partial_tiles(h, hh, ww)
            part = torch.empty((n * module._k, nt, 2), dtype=torch.float32, device=dev) if nt else None
            args = (h, x.data_ptr(), n, hh, ww, heat.data_ptr(), part.data_ptr() if nt else None, ws_ptr, ws_bytes,
                    C.c_void_p(ts.cuda_stream))
            if torch.cuda.current_device() == dev.index:
                rc = self.lib.esahrnet_forward_partials(*args)
            else:
                with torch.cuda.device(dev):
                    rc = self.lib.esahrnet_forward_partials(*args)
            _lib.check(rc)
        ws.record_stream(ts)
        x.record_stream(ts)
        if nt:
            try:
                heat._esa_partials = (part, nt, heat._version)
            except RuntimeError:            # torch.inference_mode(): no version counter, so no way to tell a later edit
                pass
        return heat

    def _partial_tiles(self, h, hh, ww):
        if os.environ.get("ESAHRNET_NO_PARTIALS"):
            return 0
        key = (getattr(h, "value", h), hh, ww)
        nt = self.part_tiles.get(key)
        if nt is None:
            v = C.c_int(0)
            _lib.check(self.lib.esahrnet_partial_tiles(h, hh, ww, C.byref(v)))
            nt = self.part_tiles[key] = v.value
        return nt

    def forward_timed(self, module, x0):
        x = self._check_input(module, x0)
        n, _, hh, ww = x.shape
        dev = x.device
        h = self._handle_for(module, dev)
        _lib.check(self.lib.esahrnet_set_debug_keep(h, 0))
        ws, ws_ptr, ws_bytes = self._workspace(h, dev, torch.cuda.current_stream(dev), n, hh, ww, False)
        heat = torch.empty((n, module._k, hh, ww), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        nops = self.lib.esahrnet_launch_count(h)
        ms = (C.c_float * nops)()
        with torch.cuda.device(dev):
            _lib.check(self.lib.esahrnet_forward_timed(h, x.data_ptr(), n, hh, ww, heat.data_ptr(), ws_ptr,
                                                       ws_bytes, C.c_void_p(stream), ms))
        ops = []
        for i in range(nops):
            d = _lib.OpDesc()
            _lib.check(self.lib.esahrnet_op_desc_get(h, i, n, hh, ww, C.byref(d)))
            if not d.kernel:        # a plan alternative that this shape does not run (esahrnet.h: op_desc_get)
                continue
            ops.append(dict(kernel=d.kernel.decode(), label=d.label.decode(), ms=float(ms[i]),
                            flops=d.flops, bytes=d.bytes))
        return heat, ops

    def taps(self, module, x0):
        x = self._check_input(module, x0)
        n, _, hh, ww = x.shape
        dev = x.device
        heat = self.forward(module, x, keep=True)
        h = self.handles[dev.index][0]
        _, ws_ptr, _ = self._workspace(h, dev, torch.cuda.current_stream(dev), n, hh, ww, True)
        out = {"heatmaps": heat}
        stream = torch.cuda.current_stream(dev).cuda_stream
        buf = C.create_string_buffer(96)
        for i in range(self.lib.esahrnet_tap_count(h)):
            _lib.check(self.lib.esahrnet_tap_name(h, i, buf, 96))
            name = buf.value
            c, th, tw = C.c_int(), C.c_int(), C.c_int()
            _lib.check(self.lib.esahrnet_tap_shape(h, name, hh, ww, C.byref(c), C.byref(th), C.byref(tw)))
            t = torch.empty((n, c.value, th.value, tw.value), dtype=torch.float32, device=dev)
            if self.lib.esahrnet_tap_read(h, name, n, hh, ww, ws_ptr, t.data_ptr(), C.c_void_p(stream)) != 0:
                if b"head alternative" in self.lib.esahrnet_last_error():
                    continue                # tensor of the head variant this shape does not run
                _lib.check(1)
            out[name.decode()] = t
        _lib.check(self.lib.esahrnet_set_debug_keep(h, 0))
        return out


def get_seg_model(cfg, **kwargs):
    """models/seg_hrnet.py:495-499."""
    model = HighResolutionNet(cfg, **kwargs)
    pre = cfg.MODEL.PRETRAINED if hasattr(cfg, "MODEL") else cfg["MODEL"]["PRETRAINED"]
    model.init_weights(pre)
    return model

"""Frame-by-frame inference with a GPU-resident sliding window.

Counterpart of run_inference.py:70-101.  The reference keeps a float32 [1,3,16,224,224] window on the GPU, and for every new frame
normalises it on the CPU (prepare_image, :15-34), uploads 600 KB of f32, drops the oldest frame and ``torch.cat``s the window
again (a 9.6 MB copy) before running the model.  Here the window is a uint8 ring buffer [1,T,H,W,3] in HBM: a new frame is ONE
150 KB upload into the oldest slot, nothing is shifted, and the patch-embed kernel reads the ring in temporal order while applying
the normalisation (``PatchEmbed.t_offset`` -> tad_im2col_tubelets_u8).  Results are identical to running the model on the
reference's window: same arithmetic, same frame order.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ._lib import TadError

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class SlidingWindow:
    def __init__(self, model: torch.nn.Module, mean=IMAGENET_MEAN, std=IMAGENET_STD, bgr: bool = True, device: Optional[torch.device] = None,
                 use_graph: bool = False):
        pe = model.patch_embed
        self.model = model
        self.T = int(model.num_frames)
        self.H, self.W = int(pe.img_size[0]), int(pe.img_size[1])
        self.device = device or next(model.parameters()).device
        if self.device.type != "cuda":
            raise TadError("SlidingWindow keeps its frames in HBM: the model must be on a GPU (no CPU path)")
        pe.set_input_normalization(mean, std, bgr=bgr)
        self.ring = torch.zeros((1, self.T, self.H, self.W, 3), dtype=torch.uint8, device=self.device)
        self.count = 0   # frames pushed so far
        self.start = 0   # ring slot of the oldest frame
        # Batch-1 inference is launch-bound (~150 short launches per window): with use_graph the forward is captured once per ring
        # offset into a HIP graph (the ring buffer is a static input; the offset is a launch constant, hence one graph per offset,
        # all sharing one memory pool) and replayed afterwards.
        self.use_graph = bool(use_graph)
        self._graphs = {}
        self._pool = None
        self._params = list(model.parameters())
        self._sig = None
        self._cap_stream = None

    def _drop_graphs(self) -> None:
        """forget every captured graph together with what only they needed: their memory pool (the allocator retires a pool with its
        last graph; a new generation gets a new one) and the scratch buffers of the capture stream"""
        self._graphs.clear()
        self._pool = None
        if self._cap_stream is not None:
            from . import kernels as K
            K.release_workspace(self.device, self._cap_stream)

    def __del__(self):
        try:
            self._drop_graphs()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    @property
    def full(self) -> bool:
        return self.count >= self.T

    def push(self, frame) -> None:
        """frame: uint8 [H,W,3] (numpy from cv2.imread + cv2.resize, or a tensor), already resized to the model's input size"""
        f = torch.from_numpy(np.ascontiguousarray(frame)) if isinstance(frame, np.ndarray) else frame
        if f.dtype != torch.uint8 or tuple(f.shape) != (self.H, self.W, 3):
            raise TypeError(f"Input must be a uint8 image of shape {(self.H, self.W, 3)}, but got {f.dtype} {tuple(f.shape)}")
        if self.count < self.T:
            slot = self.count
        else:  # overwrite the oldest frame; the window now starts one slot later
            slot = self.start
            self.start = (self.start + 1) % self.T
        self.ring[0, slot].copy_(f, non_blocking=True)
        self.count += 1

    @torch.no_grad()
    def predict(self) -> torch.Tensor:
        """raw logits [1, num_classes] for the current window (the model applies no softmax, run_inference.py:95-100)"""
        if not self.full:
            raise TadError(f"We need at least {self.T} frames! (have {self.count})")
        self.model.eval()
        self.model.patch_embed.t_offset = self.start
        try:
            if not self.use_graph:
                return self.model(self.ring)
            # A captured graph reads the bf16 weight copies (ops._wcache) and scratch buffers by raw pointer: the graphs are dropped
            # whenever the weights may have changed (a new weight epoch, a parameter re-assigned or modified in place), and each
            # graph entry keeps the copies and the scratch buffers it was captured with alive.
            from . import ops
            sig = (ops.weight_epoch(), tuple(p._version for p in self._params), tuple(p.data_ptr() for p in self._params[:4]))
            if sig != self._sig:
                self._drop_graphs()
                self._sig = sig
            ent = self._graphs.get(self.start)
            if ent is None:
                from . import kernels as K
                # Warm-up AND capture run on one stream of this object's own: kernels.workspace is keyed by (device, stream), so the
                # warm-up allocates exactly the scratch buffer the capture then bakes into the graph (allocated inside the capture
                # it would come from the graph's private pool and be pinned for good).  The buffers of that stream are held next to
                # the graphs and released together with them (_drop_graphs).
                if self._cap_stream is None:
                    self._cap_stream = torch.cuda.Stream(self.device)
                side = self._cap_stream
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):  # warm-up outside the capture (weight copies, workspaces, lazy init)
                    self.model(self.ring)
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                if self._pool is None:
                    self._pool = torch.cuda.graph_pool_handle()
                with torch.cuda.graph(g, pool=self._pool, stream=side):
                    out = self.model(self.ring)
                ent = self._graphs[self.start] = (g, out, ops.cached_weight_tensors(), K.workspace_refs(self.device, side))
            ent[0].replay()
            return ent[1].clone()
        finally:
            self.model.patch_embed.t_offset = 0

    def window_f32(self) -> torch.Tensor:
        """the reference's float window [1,3,T,H,W] rebuilt from the ring (debugging aid; torch arithmetic on the GPU, equal to the
        reference's CPU values to 1 ulp -- the kernel path itself is bit-exact)"""
        pe = self.model.patch_embed
        order = [(self.start + t) % self.T for t in range(self.T)]
        fr = self.ring[0, order].float()
        if pe.input_bgr:
            fr = fr.flip(-1)
        fr = (fr / 255.0 - torch.tensor(pe.input_mean, device=fr.device)) / torch.tensor(pe.input_std, device=fr.device)
        return fr.permute(3, 0, 1, 2).unsqueeze(0).contiguous()

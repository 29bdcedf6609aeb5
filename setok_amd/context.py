"""`setok_ctx` of the C ABI behind a small Python object: the whole `SetokTokenizer.forward` (tower -> positions -> clustering -> cluster
encoders -> out) as ONE library call (`setok_encode`, csrc/context.hip) on torch-allocated buffers.

The context is the form of the path a host in ANY language binds (include/setok_hip.h: setok_create / setok_load_weight /
setok_weights_ready / setok_encode): this file only hands over the module's parameters by their reference state-dict names and
allocates the outputs.  tests/test_context_gpu.py drives the same entry points with nothing but ctypes and numpy."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .ops import _code, _stream


def _skip(name: str) -> bool:
    if name.endswith("inv_freq") or ".post_layernorm." in name:
        return True
    if ".layers." in name and not name.startswith("image_feature_encoder."):
        return name.split(".layers.")[1].split(".")[1] == "0"                  # `layers.{i}.0.*` alias norm1 (module.py:87-88)
    return False


class EncodeContext:
    def __init__(self, tok, fold_layernorm: bool = True):
        tower = tok.image_feature_encoder
        cfg, dev, dt = tower.config, tok.device, tok.dtype
        if dev.type != "cuda":
            raise _lib.SetokHipError("the SeTok path runs on the GPU only (there is no CPU fallback): move the module to a cuda device")
        blk = tok.inner_encoder
        sc = _lib.SetokConfig(
            image_size=cfg.image_size, patch_size=cfg.patch_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
            num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads, layer_norm_eps=cfg.layer_norm_eps,
            select_layer=tower.select_layer, select_cls_patch={"patch": 0, "cls_patch": 1}[tower.select_feature],
            token_feat_dim=tok.token_feat_dim, nheads=blk.num_heads, dim_feedforward=blk.mlp.fc1.out_features,
            inner_cluster_layers=len(tok.inner_encoder.layers), intra_cluster_layers=len(tok.inter_encoder.layers),
            min_cluster_num=tok.min_cluster_num, threshold=float(tok.threshold), dtype=_code(dt), fold_layernorm=1 if fold_layernorm else 0)
        self._half = dt == torch.float16                              # which build of the library owns this context (libsetok_hip_f16.so for float16)
        self.handle = C.c_void_p()
        _lib.call("setok_create", C.byref(sc), C.byref(self.handle), half=self._half)
        self.device, self.dtype = dev, dt
        self.N = (cfg.image_size // cfg.patch_size) ** 2
        self.image_size = cfg.image_size
        self.C, self.D = cfg.hidden_size, tok.token_feat_dim
        self._ws = {}
        st = _stream()
        with torch.no_grad():
            named = dict(tok.state_dict())
            g = int(self.N ** 0.5)
            named["position_embedding.table"] = tok.position_embedding.table(g, g, dt, dev)
            for name, t in named.items():
                if _skip(name):
                    continue
                t = t.detach()
                if t.dtype not in (torch.float32, dt):
                    t = t.float()
                t = t.to(dev).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                _lib.call("setok_load_weight", self.handle, st, name.encode(), t.data_ptr(), _code(t.dtype), shape, t.dim(), half=self._half)
            _lib.call("setok_weights_ready", self.handle, st, half=self._half)
        torch.cuda.current_stream().synchronize()                     # the staging copies above read tensors that may die now

    def __deepcopy__(self, memo):
        raise TypeError("an EncodeContext owns device memory through a C handle and cannot be copied; build a new one from the module")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load(self._half).setok_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def encode(self, images: torch.Tensor, k: Optional[int] = None, threshold: Optional[float] = None, noise=None, token_mask=None,
               return_stages: bool = False, sync: bool = True, ws: Optional[torch.Tensor] = None):
        """images (B, 3, H, W) on the context's device -> (packed tokens (sum L_i, D), counts list, idx_cluster (B, N) int64,
        score (B, N) fp32, index_down (B, N) int64[, stages]).

        `setok_encode` itself never waits for the device (the ragged stages read the token counts there); this wrapper reads the B counts
        ONCE, after everything is queued, because a RaggedTokens needs host-side shapes — and waits for THEM only (they are final behind the
        clustering): the returned tensors are complete in stream order, like the result of any torch operation, while the head's last launches may
        still be running when the call returns (so the caller's next launches queue up behind them instead of finding the device idle).  sync=False skips even that: the call only enqueues
        work (it can be captured into a graph — after one eager call of the same batch size, or on a caller-owned `ws`; the captured launches then
        point into that workspace, and eager calls of that batch size on other streams are the caller's to order against the replays) and returns (tokens at the worst-case capacity (B * N, D), counts as a DEVICE int32 tensor, idx,
        score, index_down); rows of `tokens` past sum(counts) are unspecified."""
        B, N, dev = images.shape[0], self.N, self.device
        x = images.to(device=dev, dtype=self.dtype).contiguous()
        nbytes = _lib.load(self._half).setok_encode_workspace_bytes(self.handle, B)
        if ws is not None:                                                # a caller-owned workspace (GraphedEncode: the captured launches point into it)
            if ws.numel() < nbytes or ws.dtype != torch.uint8 or ws.device != dev:
                raise ValueError(f"workspace must be a uint8 tensor of >= {nbytes} bytes on {dev}")
        else:
            ws = self._ws.get(B)
        cur = torch.cuda.current_stream(dev)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and sync:
            raise ValueError("encode(sync=True) reads the token counts on the host, which a stream capture cannot contain: capture encode(sync=False)")
        if ws is None or ws.numel() < nbytes:
            if capturing:
                raise _lib.SetokHipError(f"encode(sync=False) under stream capture needs a workspace that exists already: call it once eagerly with "
                                         f"batch size {B} first, or pass ws= (GraphedEncode does)")
            # one batch size cached: the workspace is the activations' size.  The workspace it replaces may still be in use by a call that returned
            # early (ABI 8) on ANOTHER stream: order this stream behind that call and tell the allocator, so the block is not handed out again before
            # the launches reading it are done
            ev = self.__dict__.get("_ws_free")
            if ev is not None:
                cur.wait_event(ev)
            for old in self._ws.values():
                old.record_stream(cur)
            self._ws = {B: torch.empty(nbytes, dtype=torch.uint8, device=dev)}
            ws = self._ws[B]
        tokens = torch.empty((B * N, self.D), dtype=self.dtype, device=dev)
        counts = torch.empty((B,), dtype=torch.int32, device=dev)
        idx = torch.empty((B, N), dtype=torch.int64, device=dev)
        score = torch.empty((B, N), dtype=torch.float32, device=dev)
        index_down = torch.empty((B, N), dtype=torch.int64, device=dev)
        if noise is not None:
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            assert noise.numel() == B * N
        if token_mask is not None:
            token_mask = token_mask.to(device=dev, dtype=torch.float32).contiguous()
            assert token_mask.numel() == B * N
        own_ws = ws is self._ws.get(B) and not capturing
        if own_ws:
            # the context's own workspace is shared by every call of this batch size: since ABI 8 a call returns while its last launches are
            # still running, so a call made on ANOTHER stream waits for the previous call's end first (a no-op on the same stream).  NOT under
            # stream capture: a capturing stream must not wait on an event recorded outside the capture, and an event recorded inside
            # one is no event an eager call may wait on — a captured call is ordered by the graph launch, on the capturing stream, like its replays.
            ev = self.__dict__.get("_ws_free")
            if ev is not None:
                cur.wait_event(ev)
        counts_h = (C.c_int32 * B)() if sync else None
        total = C.c_int64(0)
        sx, sg, si = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.call("setok_encode", self.handle, _stream(), x.data_ptr(), B, int(k) if k else 0, float(threshold) if threshold else 0.0,
                  None if noise is None else noise.data_ptr(), None if token_mask is None else token_mask.data_ptr(),
                  ws.data_ptr(), ws.numel(), tokens.data_ptr(), counts.data_ptr(), idx.data_ptr(), score.data_ptr(), index_down.data_ptr(),
                  counts_h, C.byref(total) if sync else None, C.byref(sx) if return_stages else None, C.byref(sg) if return_stages else None,
                  C.byref(si) if return_stages else None, half=self._half)
        if own_ws:
            ev = self.__dict__.get("_ws_free")
            if ev is None:
                ev = self.__dict__["_ws_free"] = torch.cuda.Event()
            ev.record(cur)
        if not sync:
            if return_stages:
                raise ValueError("return_stages needs the host-side counts (sync=True)")
            return tokens, counts, idx, score, index_down
        cl = list(counts_h)
        out = (tokens[: total.value], cl, idx, score, index_down)
        if not return_stages:
            return out
        es = tokens.element_size()

        def view(ptr, rows):                                          # a stage lives inside the workspace tensor
            off = ptr.value - ws.data_ptr()
            return ws[off: off + rows * self.C * es].view(self.dtype).reshape(rows, self.C).clone()
        return out + (dict(x=view(sx, B * N), group=view(sg, total.value), inter=view(si, total.value), index_down=index_down, counts=cl),)


class GraphedEncode:
    """The latency form of the path for small, repeated batches (the reference's dataset side encodes ONE image per call from DataLoader workers,
    src/dataset/pairDataset.py:419-421): `setok_encode` is free of host synchronisation, so a whole call — ~200 kernel launches at ViT-L — is
    captured ONCE per batch size into a HIP graph on static buffers and replayed; per call the host then pays one graph launch, one copy of the
    images into the static input and one read of the B token counts.  Bit-identical to the eager call (the same kernels on the same data)."""

    def __init__(self, ctx, B: int, k: Optional[int] = None, threshold: Optional[float] = None):
        """`ctx`: an EncodeContext, or — preferred — the SetokTokenizer itself: the graph then follows the module (a weight update, `.to()`, a
        changed select_layer rebuild the module's context, and the next call re-captures instead of replaying launches that read the OLD
        context's weight copies).  Given a bare context the graph is pinned to it and says so when asked (`stale()` is always False)."""
        self.tok = None
        if not isinstance(ctx, EncodeContext):
            self.tok, ctx = ctx, ctx._context()
        self.B, self.k, self.threshold = B, k, threshold
        self._capture(ctx)

    def _capture(self, ctx: "EncodeContext"):
        self.ctx, B = ctx, self.B
        self.images = torch.zeros((B, 3, ctx.image_size, ctx.image_size), dtype=ctx.dtype, device=ctx.device)
        nbytes = _lib.load(ctx._half).setok_encode_workspace_bytes(ctx.handle, B)
        self._ws = torch.empty(nbytes, dtype=torch.uint8, device=ctx.device)          # the graph's OWN workspace: eager encode() calls of the same batch size
        ctx.encode(self.images, self.k, self.threshold, sync=False, ws=self._ws)       # (any stream) never touch it.  Warm-up: one-time attribute calls stay out of the capture
        torch.cuda.synchronize(ctx.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = ctx.encode(self.images, self.k, self.threshold, sync=False, ws=self._ws)

    def stale(self) -> bool:
        """Has the module this graph was captured from moved on (new weights / dtype / device / layer selection)?"""
        return self.tok is not None and self.tok._context() is not self.ctx

    def refresh(self) -> bool:
        """Re-capture now if the module has moved on; returns whether it did.  A re-capture SYNCHRONISES the device (warm-up call + capture), so
        a caller that is itself inside a capture, or that wants to choose the moment, calls this outside its critical section; __call__ does it
        implicitly otherwise."""
        if self.tok is None:
            return False
        ctx = self.tok._context()                                                      # ONE look-up per call (the key walks every parameter's version)
        if ctx is self.ctx:
            return False
        self.graph = self.out = None
        self._capture(ctx)
        return True

    def __call__(self, images: torch.Tensor):
        """-> (packed tokens (sum L_i, D) — a fresh tensor —, counts list, idx_cluster, score, index_down) like EncodeContext.encode.
        If the module's weights / dtype / device / layer selection changed since the capture, the graph is re-captured first (see refresh():
        that synchronises the device)."""
        self.refresh()
        assert tuple(images.shape) == tuple(self.images.shape)
        self.images.copy_(images, non_blocking=True)
        self.graph.replay()
        tokens, counts, idx, score, index_down = self.out
        cl = counts.cpu().tolist()                                                      # the one synchronisation, after the replay was queued
        return tokens[: sum(cl)].clone(), cl, idx.clone(), score.clone(), index_down.clone()

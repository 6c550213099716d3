"""Training path of `CNNViTHybrid`: forward with saved activations + hand-scheduled backward on
librvit kernels (autograd of train_vit.py:141-171 as run by `loss.backward()`, train_vit.py:939),
exposed to PyTorch through one `torch.autograd.Function`, plus the data-parallel gradient exchange.

Gradients of ALL parameters are produced in ONE flat fp32 buffer laid out in the order the backward
pass finishes them (head first, stem conv1 last).  Under data parallelism the flat buffer is cut
into buckets that are all-reduced (RCCL, averaged) on a side stream as soon as the backward has
written them, overlapping communication with the rest of the backward.

Precision: bf16 operands / fp32 accumulation everywhere, fp32 master weights and fp32 gradients,
no loss scaling (deviation from the reference's fp16 autocast + GradScaler, train_vit.py:896,935;
stated in SURVEY.md §8a row 16).  BatchNorm uses per-replica batch statistics (no SyncBN in the
reference).  Dropout (train_vit.py:100,107,125) is applied only when p == 0 is NOT requested:
see `_dropout_supported`.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .plan import run_planned

BF16 = torch.bfloat16
F32 = torch.float32


def _t(w: torch.Tensor) -> torch.Tensor:
    return w.detach().t().to(BF16).contiguous()


def _b(w: torch.Tensor) -> torch.Tensor:
    return w.detach().to(BF16).contiguous()


def _conv_fwd_pack(w: torch.Tensor) -> torch.Tensor:   # [Cout,Cin,kh,kw] -> [Cout, kh*kw*Cin]
    return w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(BF16).contiguous()


def _conv_bwd_pack(w: torch.Tensor, tap_order=None) -> torch.Tensor:   # [Cout,Cin,kh,kw] -> [Cin, kh*kw*Cout]
    t = w.detach().permute(1, 2, 3, 0).reshape(w.shape[1], w.shape[2] * w.shape[3], w.shape[0])   # [Cin, tap, Cout]
    if tap_order is not None:   # K block p holds tap tap_order[p] (stride-2 input gradient by parity, ops.S2_TAP_ORDER)
        t = t[:, list(tap_order), :]
    return t.reshape(w.shape[1], -1).to(BF16).contiguous()


def _px4_pack(w: torch.Tensor, kw_pad: int) -> torch.Tensor:
    cout, cin, kh, kw = w.shape
    p = torch.zeros(cout, kh, kw_pad, 4, dtype=F32, device=w.device)
    p[:, :, :kw, :cin] = w.detach().permute(0, 2, 3, 1)
    return p.reshape(cout, -1).to(BF16).contiguous()


class _TorchPacker:
    """Operand copies made with torch ops straight from the fp32 parameters (any optimiser)."""

    def __init__(self, model):
        self.p = dict(model.named_parameters())

    def _2d(self, n):
        w = self.p[n].detach()
        return w.reshape(w.shape[0], -1)

    def b(self, n): return _b(self._2d(n))
    def t(self, n): return _t(self._2d(n))
    def conv_fwd(self, n): return _conv_fwd_pack(self.p[n])
    def conv_bwd(self, n, tap_order=None): return _conv_bwd_pack(self.p[n], tap_order)
    def px4(self, n, kw_pad): return _px4_pack(self.p[n], kw_pad)

    t_small = t   # (WeightMirror: transposes whose rows are not 8-element aligned, refreshed with a torch copy)

    def t_padded(self, n, mult):
        w = self.p[n].detach()
        out = torch.zeros(w.shape[1], mult * ((w.shape[0] + mult - 1) // mult), dtype=BF16, device=w.device)
        out[:, : w.shape[0]] = w.t().to(BF16)
        return out


class WeightMirror:
    """bf16 operand copies of every parameter for the training engine, kept current by the optimiser.

    `bf16` mirrors the flat fp32 parameter buffer of vit_amd.optim.FusedAdamW element for element (the AdamW
    kernel writes it, ops.adamw_step(params_bf16=...)): since that buffer stores conv weights channels-last,
    every forward-layout pack ([Cout, kh*kw*Cin], [N, K]) is simply a VIEW of it.  The K-major copies that the
    data-gradient GEMMs / convolutions read live in `wt`, rebuilt from `bf16` by ONE batched-transpose launch
    (rvit_transpose_batched_bf16).  Three tiny padded packs (7x7 stem, 16x16 patch conv, class head) are
    refreshed with torch ops.  `pack` has the same keys as TrainEngine._build_pack with any packer."""

    def __init__(self, eng: "TrainEngine", flat_params: torch.Tensor):
        self.eng, self.flat = eng, flat_params
        self.L = eng.layout
        dev = flat_params.device
        self.bf16 = torch.zeros(self.L.total, dtype=BF16, device=dev)
        self._entries: List[tuple] = []
        self._wt_views: List[tuple] = []   # (dst_off, numel, shape) resolved once `wt` is allocated
        self._wt_size = 0
        self._small = []                   # closures refreshing the padded packs
        self._dev = dev
        pack = eng._build_pack(self)
        self.wt = torch.zeros(max(self._wt_size, 8), dtype=BF16, device=dev)
        self.pack = self._resolve(pack)
        self.table = ops.transpose_table(self._entries, dev) if self._entries else None
        self.key = None
        self.refresh(full=True)
        self.key = eng._key()

    # -- packer interface (called by TrainEngine._build_pack) --
    def _seg(self, n):
        off, numel, shape = self.L.offsets[n]
        return off, numel, shape

    def b(self, n):
        off, numel, shape = self._seg(n)
        return self.bf16[off:off + numel].view(shape[0], -1)

    conv_fwd = b    # flat layout of stem convs is [Cout, kh, kw, Cin] already

    def _alloc(self, numel):
        d = self._wt_size
        self._wt_size += (numel + 63) // 64 * 64
        return d

    def t(self, n):
        off, numel, shape = self._seg(n)
        N, K = shape[0], numel // shape[0]
        d = self._alloc(numel)
        self._entries.append((off, d, N, K, K, N))
        return ("wt", d, numel, (K, N))

    def conv_bwd(self, n, tap_order=None):
        off, numel, shape = self._seg(n)
        co, ci, kh, kw = shape
        T = kh * kw
        d = self._alloc(numel)
        order = range(T) if tap_order is None else tap_order
        for pos, tap in enumerate(order):   # dst[ci, pos, co] = src[co, tap, ci]
            self._entries.append((off + tap * ci, d + pos * co, co, ci, T * ci, T * co))
        return ("wt", d, numel, (ci, T * co))

    def px4(self, n, kw_pad):
        off, numel, shape = self._seg(n)
        co, ci, kh, kw = shape
        src = self.bf16[off:off + numel].view(co, kh, kw, ci)
        out = torch.zeros(co, kh, kw_pad, 4, dtype=BF16, device=self._dev)
        self._small.append(lambda: out[:, :, :kw, :ci].copy_(src))
        return out.view(co, -1)

    def t_small(self, n):
        """[N, K] -> [K, N] for shapes the batched transpose cannot take (depthwise filters: K = 9 / 25)."""
        off, numel, shape = self._seg(n)
        src = self.bf16[off:off + numel].view(shape[0], -1)
        out = torch.zeros(src.shape[1], src.shape[0], dtype=BF16, device=self._dev)
        self._small.append(lambda: out.copy_(src.t()))
        return out

    def t_padded(self, n, mult):
        off, numel, shape = self._seg(n)
        src = self.bf16[off:off + numel].view(shape)
        out = torch.zeros(shape[1], mult * ((shape[0] + mult - 1) // mult), dtype=BF16, device=self._dev)
        self._small.append(lambda: out[:, : shape[0]].copy_(src.t()))
        return out

    def _resolve(self, node):
        if isinstance(node, tuple) and len(node) == 4 and node[0] == "wt":
            _, d, numel, shape = node
            return self.wt[d:d + numel].view(shape)
        if isinstance(node, dict):
            return {k: self._resolve(v) for k, v in node.items()}
        if isinstance(node, list):
            return [self._resolve(v) for v in node]
        return node

    @torch.no_grad()
    def refresh(self, full: bool) -> None:
        """full: re-cast the fp32 parameters too (parameters changed outside the optimiser kernel)."""
        if full:
            ops.cast_bf16(self.flat, self.bf16)
        if self.table is not None:
            ops.transpose_batched(self.bf16, self.wt, self.table)
        for f in self._small:
            f()


class _GradLayout:
    """Flat fp32 gradient buffer: one segment per parameter, in backward-completion order."""

    def __init__(self, model):
        named = dict(model.named_parameters())
        order: List[str] = ["head.3.weight", "head.3.bias", "head.0.weight", "head.0.bias", "norm.weight", "norm.bias"]
        for i in reversed(range(model.num_layers)):
            p = f"transformer.layers.{i}."
            order += [p + "linear2.weight", p + "linear2.bias", p + "linear1.weight", p + "linear1.bias",
                      p + "norm2.weight", p + "norm2.bias", p + "self_attn.out_proj.weight",
                      p + "self_attn.out_proj.bias", p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias",
                      p + "norm1.weight", p + "norm1.bias"]
        order += ["pos_embed", "cls_token", "patch_embed.weight", "patch_embed.bias"]
        if model.cnn_backbone is not None:
            stem = [n for n in named if n.startswith("cnn_backbone.")]
            # reverse module order: layer4.1 ... layer1.0, then conv1/bn1
            order += list(reversed(stem))
        assert set(order) == set(named), set(named) ^ set(order)
        self.names = order
        self.offsets: Dict[str, Tuple[int, int, torch.Size]] = {}
        off = 0
        for n in order:
            numel = named[n].numel()
            self.offsets[n] = (off, numel, named[n].shape)
            off += (numel + 63) // 64 * 64  # 256-byte aligned segments
        self.total = off
        self.param_order = [n for n, _ in model.named_parameters()]


# dropout sites of the reference model (train_vit.py:100,107,125; torch transformer.py:977-982)
SITE_POS, SITE_HEAD = 0, 1
SITE_ATTN, SITE_PROJ, SITE_FFN, SITE_FC2 = 0, 1, 2, 3
_M64 = (1 << 64) - 1


def _layer_site(layer: int, which: int) -> int:
    return 4 + 4 * layer + which


def _site_seed(base: int, site: int) -> int:
    """splitmix64 of (step seed, site id) -> the 64-bit key of one dropout site."""
    z = (base + 0x9E3779B97F4A7C15 * (site + 1)) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


class TrainEngine:
    def __init__(self, model):
        self.m = model
        self.layout = _GradLayout(model)
        self._pack = None
        self.mirror = None   # WeightMirror, installed by vit_amd.optim.FusedAdamW
        self._bufs: Dict[int, dict] = {}
        self.reducer = None
        self.last_flat = None
        # BatchNorm batch statistics and their backward sums by ordered reductions (same bits on every run): what
        # torch.backends.cudnn.deterministic asks of the reference (train_vit.py:50).  RVIT_BN_ORDERED=0 selects the
        # float-atomic form (a few percent faster on the small maps, last bits vary).
        import os
        self.ordered_bn = os.environ.get("RVIT_BN_ORDERED", "1") != "0"
        # ... and every other split reduction of the step (weight / bias gradients, LayerNorm affine gradients, the
        # gradient norm): partials stored per split and added in order (ops.enable_ordered_reductions; one 192 MB
        # scratch workspace per device).  With both on, two runs of a step give the same bits in every gradient and
        # parameter.  RVIT_ORDERED=0 selects the float-atomic forms.
        self.ordered = os.environ.get("RVIT_ORDERED", "1") != "0"
        # stem bn1 backward with the max-pool gradient gathered in its two passes (0: maxpool_bwd + bn_bwd, for A/B runs)
        self.fused_pool_bwd = os.environ.get("RVIT_POOL_BWD_FUSED", "1") != "0"
        # the stem's first conv by the rolling-band kernel from the fp32 input (0: implicit GEMM on the packed copy, A/B runs)
        self.conv1_roll = os.environ.get("RVIT_CONV1_TRAIN_ROLL", "1") != "0"
        # ... and its weight gradient on the same band (ordered slots only; 0: implicit GEMM on the packed copy)
        self.conv1_wgrad_roll = os.environ.get("RVIT_CONV1_WGRAD_ROLL", "1") != "0"
        # BatchNorm backward sums in the epilogue of the dgrad conv that produces the norm's incoming gradient (0: the
        # bn_bwd_reduce pass, A/B runs)
        self.bnr_fused = os.environ.get("RVIT_BN_REDUCE_FUSED", "1") != "0"
        # the stem's forward and backward passes replayed from recorded launch plans (vit_amd/plan.py; needs the
        # optimiser-maintained operand copies of vit_amd.optim.FusedAdamW; 0: always through the op wrappers)
        self.launch_plans = os.environ.get("RVIT_LAUNCH_PLAN", "1") != "0"
        self._plans: dict = {}
        self.plan_replays: dict = {}   # pass name -> how often it ran from its plan (tests, bench)
        self.mb = None   # MobileStemTrainer for backbone='mobilenetv3'
        if getattr(model, "backbone_name", None) == "mobilenetv3":
            from .mobile_trainer import MobileStemTrainer
            self.mb = MobileStemTrainer(self)

    # ---- data parallel -------------------------------------------------------------------------
    def set_data_parallel(self, process_group=None, bucket_mb: float = 8.0) -> None:
        from .dp import BucketReducer
        self.reducer = BucketReducer(self.layout.total, process_group, int(bucket_mb * (1 << 20)))

    # ---- packing ---------------------------------------------------------------------------------
    def _key(self):
        ps = self.m._live_params()
        return (tuple(p._version for p in ps), ps[0].data_ptr())

    @torch.no_grad()
    def pack(self) -> dict:
        """bf16 operand copies of the weights for this forward/backward.

        Without a WeightMirror they are rebuilt on EVERY call: version counters cannot be trusted to notice an
        optimiser step (torch.optim.AdamW(fused=True) updates parameters without bumping `_version`), and a
        stale pack would silently train on old weights.  With vit_amd.optim.FusedAdamW the optimiser kernel
        itself maintains the copies (WeightMirror) and nothing is rebuilt here."""
        mir = self.mirror
        if mir is not None:
            key = self._key()
            if key != mir.key:   # parameters changed behind the optimiser's back (load_state_dict, manual edits)
                mir.refresh(full=True)
                mir.key = key
            self._pack = mir.pack
            return mir.pack
        with ops._Span("weight_pack", 0, 0):
            self._pack = self._build_pack(_TorchPacker(self.m))
        return self._pack

    def _build_pack(self, pk) -> dict:
        m = self.m
        P: dict = {}
        if self.mb is not None:
            P["mb"] = self.mb.build_pack(pk)
            P["pe_w"], P["pe_wt"] = pk.b("patch_embed.weight"), pk.t("patch_embed.weight")
        elif m.cnn_backbone is not None:
            cb = m.cnn_backbone
            P["conv1_w"] = pk.px4("cnn_backbone.conv1.weight", 8)
            blocks = []
            for pref, blk in self._stem_blocks():
                # a stride-2 conv1 gets its input-gradient pack in the parity order of rvit_conv2d_dgrad_s2_bf16
                s2 = blk.conv1.stride == 2
                d = {"stride": blk.conv1.stride, "cin": blk.conv1.cin, "cout": blk.conv1.cout,
                     "c1_w": pk.conv_fwd(pref + "conv1.weight"),
                     "c1_wt": pk.conv_bwd(pref + "conv1.weight", ops.S2_TAP_ORDER if s2 else None),
                     "c2_w": pk.conv_fwd(pref + "conv2.weight"), "c2_wt": pk.conv_bwd(pref + "conv2.weight")}
                if blk.downsample is not None:
                    d["ds_w"] = pk.conv_fwd(pref + "downsample.0.weight")
                    d["ds_wt"] = pk.conv_bwd(pref + "downsample.0.weight")
                blocks.append(d)
            P["blocks"] = blocks
            P["pe_w"], P["pe_wt"] = pk.b("patch_embed.weight"), pk.t("patch_embed.weight")
        else:
            P["pe_w"] = pk.px4("patch_embed.weight", 16)
        layers = []
        for i in range(m.num_layers):
            p = f"transformer.layers.{i}."
            layers.append(dict(qkv_w=pk.b(p + "self_attn.in_proj_weight"), qkv_wt=pk.t(p + "self_attn.in_proj_weight"),
                               out_w=pk.b(p + "self_attn.out_proj.weight"), out_wt=pk.t(p + "self_attn.out_proj.weight"),
                               fc1_w=pk.b(p + "linear1.weight"), fc1_wt=pk.t(p + "linear1.weight"),
                               fc2_w=pk.b(p + "linear2.weight"), fc2_wt=pk.t(p + "linear2.weight")))
        P["layers"] = layers
        P["h0_w"], P["h0_wt"] = pk.b("head.0.weight"), pk.t("head.0.weight")
        P["h3_w"] = pk.b("head.3.weight")
        P["h3_wt"] = pk.t_padded("head.3.weight", 16)
        return P

    def _stem_blocks(self):
        cb = self.m.cnn_backbone
        for li in range(1, 5):
            for bi, blk in getattr(cb, f"layer{li}").items():
                yield f"cnn_backbone.layer{li}.{bi}.", blk

    # ---- buffers -----------------------------------------------------------------------------------
    def buffers(self, B: int, dev) -> dict:
        ops.enable_ordered_reductions(dev, self.ordered)
        b = self._bufs.get(B)
        if b is not None and b["dev"] == dev:
            return b
        self._bufs.clear()
        m = self.m
        E, S, H, Lr = m.embed_dim, m.num_patches + 1, m.num_heads, m.num_layers
        T = B * S
        z = dict(dtype=BF16, device=dev)
        f = dict(dtype=F32, device=dev)
        b = {"dev": dev}
        if self.mb is not None:
            b["mb"] = self.mb.buffers(B, dev)
            b["scratch"] = {}
        elif m.cnn_backbone is not None:
            b["img"] = torch.zeros(B, 230, 232, 4, **z)
            b["c1_y"] = torch.empty(B, 112, 112, 64, **z)
            # sign bits of the ReLU outputs (one byte per 8 channels): what the BatchNorm backward reads instead of z
            b["c1_m"] = torch.empty(B * 112 * 112 * 8, dtype=torch.uint8, device=dev)
            b["pool"] = torch.empty(B, 56, 56, 64, **z)
            b["pool_arg"] = torch.empty(B, 56, 56, 64, dtype=torch.uint8, device=dev)
            blocks = []
            hw = 56
            nsum = 128  # conv1 bn sums
            for blk in m.cnn_backbone.blocks():
                s, co = blk.conv1.stride, blk.conv1.cout
                hw //= s
                d = {nm: torch.empty(B, hw, hw, co, **z) for nm in ("y1", "z1", "y2", "zo")}
                d["m1"] = torch.empty(B * hw * hw * (co // 8), dtype=torch.uint8, device=dev)
                d["mo"] = torch.empty(B * hw * hw * (co // 8), dtype=torch.uint8, device=dev)
                if blk.downsample is not None:
                    d["yd"] = torch.empty(B, hw, hw, co, **z)
                    d["zd"] = torch.empty(B, hw, hw, co, **z)
                blocks.append(d)
                nsum += 2 * co * (3 if blk.downsample is not None else 2)
            b["blocks"] = blocks
            b["fsums"] = torch.zeros(nsum, **f)
            b["bsums"] = torch.zeros(nsum, **f)
            b["bn_ws"] = ops.bn_workspace(dev)
            # scratch maps for the stem backward, sized per stage on demand
            b["scratch"] = {}
        else:
            b["img"] = torch.zeros(B, 224, 224, 4, **z)
        b["xs"] = [torch.empty(B, S, E, **f) for _ in range(2 * Lr + 1)]
        b["stat"] = torch.empty(Lr, 4, T, **f)
        b["ln1"] = [torch.empty(T, E, **z) for _ in range(Lr)]
        b["ln2"] = [torch.empty(T, E, **z) for _ in range(Lr)]
        b["qkv"] = [torch.empty(T, 3 * E, **z) for _ in range(Lr)]
        b["att"] = [torch.empty(T, E, **z) for _ in range(Lr)]
        b["lse"] = [torch.empty(B * H * S, **f) for _ in range(Lr)]
        b["u"] = [torch.empty(T, 4 * E, **z) for _ in range(Lr)]
        b["h"] = [torch.empty(T, 4 * E, **z) for _ in range(Lr)]
        b["cls_stat"] = torch.empty(2, B, **f)
        b["cls_ln"] = torch.empty(B, E, **z)
        b["u0"] = torch.empty(B, E // 2, **z)
        b["h0"] = torch.empty(B, E // 2, **z)
        # backward scratch
        b["dxs"] = torch.empty(B, S, E, **f)
        b["dxs_bf"] = torch.empty(T, E, **z)
        b["d_e"] = torch.empty(T, E, **z)
        b["d_3e"] = torch.empty(T, 3 * E, **z)
        b["d_4e"] = torch.empty(T, 4 * E, **z)
        b["d_h0"] = torch.empty(B, E // 2, **z)
        b["d_cls"] = torch.empty(B, E, **z)
        b["dl16"] = torch.zeros(B, 16 * ((m.num_classes + 15) // 16), **z)
        b["tmp16"] = torch.zeros(16 * ((m.num_classes + 15) // 16), **f)
        self._bufs[B] = b
        return b

    def _scratch(self, b, key, like: torch.Tensor) -> torch.Tensor:
        t = b["scratch"].get(key)
        if t is None or t.shape != like.shape:
            t = torch.empty_like(like)
            b["scratch"][key] = t
        return t

    # ---- forward -----------------------------------------------------------------------------------
    def _bn_params(self, bn):
        return bn.weight.detach(), bn.bias.detach()

    def _bn_modules(self):
        mods = self.__dict__.get("_bn_mods")
        if mods is None:
            mods = [mod for mod in self.m.cnn_backbone.modules() if hasattr(mod, "num_batches_tracked")]
            self.__dict__["_bn_mods"] = mods
        return mods

    def _bn_counters(self):
        return [mod.num_batches_tracked for mod in self._bn_modules()]   # looked up per step: .to() replaces the buffers

    def _stat_ws(self, dev):
        ws = self.__dict__.get("_stat_ws_buf")
        if ws is None or ws.device != dev:
            ws = ops.conv_stats_workspace(dev)
            self.__dict__["_stat_ws_buf"] = ws
        return ws

    def _conv_bn(self, x, w, y, z, bn, fs, *, kh, stride, pad, relu, residual=None, seg_len=None, kw=None, mask=None,
                 ws=None):
        # conv + batch statistics: where the conv kernel has the statistics epilogue the sums come from per-tile slots it
        # writes itself (no pass over the stored map), else rvit_bn_stats runs (csrc/gemm.hip, include/rvit.h)
        ops.conv2d_bn_stats(x, w, y, fs, self._stat_ws(y.device), kh=kh, kw=kh if kw is None else kw, stride=stride,
                            pad=pad, seg_len=seg_len, bn_workspace=ws)
        ops.bn_apply(y, fs, bn.weight.detach(), bn.bias.detach(), z, eps=bn.eps, relu=relu, residual=residual,
                     momentum=bn.momentum, running_mean=bn.running_mean, running_var=bn.running_var, relu_mask=mask)

    def _stem_tensors(self, cache):
        """Every parameter and buffer of the stem, re-listed when the model's tensor-slot cache was rebuilt."""
        have = self.__dict__.get("_stem_tl")
        if have is None or have[0] is not cache:
            cb = self.m.cnn_backbone
            have = (cache, list(cb.parameters()) + list(cb.buffers()))
            self.__dict__["_stem_tl"] = have
        return have[1]

    def _plan_key(self, which: str, P, b, extra=()):
        """Key of a launch plan (vit_amd/plan.py): everything the recorded launches depend on that is not patched.
        None: no plan for this call."""
        mir = self.mirror
        if not self.launch_plans or mir is None or P is not mir.pack or ops._PROF is not None or self.mb is not None:
            return None, None
        if which == "stem_forward":
            self.m._tensor_slots()   # every module slot still holds the tensor object the lists were built from (~15 us)
        cache = self.m.__dict__.get("_slots_cache")
        if cache is None:
            return None, None
        dev = b["dev"]
        stream = ops._stream()
        ows = ops._ORDERED_WS.get((dev.index, stream))
        key = (which, id(b), id(mir), id(cache), id(b["scratch"]), len(b["scratch"]), stream,
               ows.data_ptr() if ows is not None else 0, self._stat_ws(dev).data_ptr(), self.ordered, self.ordered_bn,
               self.conv1_roll, self.conv1_wgrad_roll, self.bnr_fused, self.fused_pool_bwd, b.get("xin") is None,
               mir.wt.data_ptr(), mir.bf16.data_ptr(), tuple(t.data_ptr() for t in self._stem_tensors(cache)),
               tuple((bn.eps, bn.momentum) for bn in self._bn_modules()), extra)
        return key, (b, mir, cache, b["scratch"])

    def _stem_forward(self, P, b, B, xin=None) -> torch.Tensor:
        b["fsums"].zero_()
        roll = self.conv1_roll and xin is not None and tuple(xin.shape[1:]) == (3, 224, 224)
        key, refs = self._plan_key("stem_forward", P, b, (roll, tuple(xin.shape) if xin is not None else None))
        ranges = {"xin": (xin.data_ptr(), xin.numel() * 4)} if xin is not None else {}
        cur = run_planned(self._plans, "stem_forward", key, refs, ranges,
                          lambda: self._stem_forward_launches(P, b, B, xin), self.plan_replays)
        # BatchNorm2d.num_batches_tracked += 1 for every norm of the stem (timm BatchNorm2d in train()): one fused launch
        # instead of twenty
        torch._foreach_add_(self._bn_counters(), 1)
        return cur

    def _stem_forward_launches(self, P, b, B, xin=None) -> torch.Tensor:
        """C-ABI launches and views only: this pass is recorded and replayed (vit_amd/plan.py)."""
        cb = self.m.cnn_backbone
        off = 0
        fs = b["fsums"]
        ws = b["bn_ws"] if self.ordered_bn else None
        # stem entry: conv 7x7/2, then BatchNorm + ReLU + max-pool in one pass (the normalised 112x112 map is not
        # written: the backward pass needs only its sign bits and the argmax taps)
        if self.conv1_roll and xin is not None and tuple(xin.shape[1:]) == (3, 224, 224):
            # rolling-band kernel straight from the fp32 input, statistics in one slot per workgroup (same stored map)
            ops.stem_conv1_stats_f32(xin, P["conv1_w"], b["c1_y"], fs[off:off + 128], self._stat_ws(b["c1_y"].device))
        else:
            ops.conv2d_bn_stats(b["img"], P["conv1_w"], b["c1_y"], fs[off:off + 128], self._stat_ws(b["c1_y"].device),
                                kh=7, kw=1, stride=2, pad=0, seg_len=32, bn_workspace=ws)
        ops.bn_apply_pool(b["c1_y"], fs[off:off + 128], cb.bn1.weight.detach(), cb.bn1.bias.detach(), b["pool"],
                          b["pool_arg"], b["c1_m"], eps=cb.bn1.eps, momentum=cb.bn1.momentum,
                          running_mean=cb.bn1.running_mean, running_var=cb.bn1.running_var)
        off += 128
        cur = b["pool"]
        for blk, pk, d in zip(cb.blocks(), P["blocks"], b["blocks"]):
            s, co = pk["stride"], pk["cout"]
            d["in"] = cur
            d["fs_off"] = off
            self._conv_bn(cur, pk["c1_w"], d["y1"], d["z1"], blk.bn1, fs[off:off + 2 * co], kh=3, stride=s, pad=1,
                          relu=True, mask=d["m1"], ws=ws)
            off += 2 * co
            idn = cur
            if "ds_w" in pk:
                self._conv_bn(cur, pk["ds_w"], d["yd"], d["zd"], blk.downsample["1"], fs[off + 2 * co:off + 4 * co],
                              kh=1, stride=s, pad=0, relu=False, ws=ws)
                idn = d["zd"]
            self._conv_bn(d["z1"], pk["c2_w"], d["y2"], d["zo"], blk.bn2, fs[off:off + 2 * co], kh=3, stride=1, pad=1,
                          relu=True, residual=idn, mask=d["mo"], ws=ws)
            off += 2 * co * (2 if "ds_w" in pk else 1)
            cur = d["zo"]
        return cur

    def forward(self, x: torch.Tensor, features_only: bool) -> torch.Tensor:
        m = self.m
        P = self.pack()
        B = x.shape[0]
        E, S, H, NP = m.embed_dim, m.num_patches + 1, m.num_heads, m.num_patches
        T = B * S
        b = self.buffers(B, x.device)
        # dropout (train_vit.py:100,107,125): one 62-bit draw per step from torch's CPU generator (so
        # torch.manual_seed reproduces a run and ranks seeded differently drop differently), expanded
        # into one key per site; the backward pass regenerates the masks from the same keys
        dp = float(m.dropout)
        base = int(torch.randint(0, 1 << 62, (1,)).item()) if dp > 0.0 else 0
        b["drop"] = (dp, base)
        sd = lambda site: _site_seed(base, site)  # noqa: E731
        xin = x.detach().float().contiguous()
        xs = b["xs"]
        x0 = xs[0]
        if m.cnn_backbone is not None:
            if self.mb is not None:
                if not self.ordered:
                    raise RuntimeError("the mobilenetv3 training kernels only have ordered reductions (RVIT_ORDERED=0 is set)")
                feat = self.mb.forward(P["mb"], b["mb"], B, xin)
                torch._foreach_add_(self._bn_counters(), 1)
            else:
                # both directions of the 7x7 conv read the fp32 input itself on the rolling band; otherwise the packed copy
                # (the weight gradient needs one 64 x 224 slot per image in the ordered workspace: B <= ~3,400 at 192 MB)
                b["xin"] = xin if (self.conv1_roll and self.conv1_wgrad_roll and self.ordered
                                   and tuple(xin.shape[1:]) == (3, 224, 224)
                                   and B * 64 * 224 * 4 <= ops.ORDERED_WORKSPACE_BYTES) else None
                if b["xin"] is None:
                    ops.input_pack(xin, b["img"], 3, 3)
                feat = self._stem_forward(P, b, B, xin)
            b["feat"] = feat
            ops.gemm(feat.view(B * NP, -1), P["pe_w"], x0.view(T, E), bias=m.patch_embed.bias.detach(),
                     residual=m.pos_embed.detach().view(S, E)[1:], residual_period=NP, row_group=NP, row_extra=1,
                     drop=("post_act", dp, sd(SITE_POS)))
        else:
            ops.input_pack(xin, b["img"], 0, 0)
            ops.conv2d_tokens(b["img"], P["pe_w"], x0, m.pos_embed.detach().view(S, E), kh=16, kw=1, stride=16, pad=0,
                              seg_len=64, ho=14, wo=14, bias=m.patch_embed.bias.detach(), drop_p=dp,
                              drop_seed=sd(SITE_POS))
        ops.embed_cls(m.cls_token.detach().view(-1), m.pos_embed.detach().view(S, E), x0, dp, sd(SITE_POS))
        for i, (L, W) in enumerate(zip(m.transformer.layers, P["layers"])):
            xi, xm, xo = xs[2 * i].view(T, E), xs[2 * i + 1].view(T, E), xs[2 * i + 2].view(T, E)
            st = b["stat"][i]
            ops.layernorm(xi, L.norm1.weight.detach(), L.norm1.bias.detach(), b["ln1"][i], L.norm1.eps, st[0], st[1])
            ops.gemm(b["ln1"][i], W["qkv_w"], b["qkv"][i], bias=L.self_attn.in_proj_bias.detach())
            ops.attention(b["qkv"][i], b["att"][i], B, S, H, lse=b["lse"][i], drop_p=dp,
                          drop_seed=sd(_layer_site(i, SITE_ATTN)))
            ops.gemm(b["att"][i], W["out_w"], xm, bias=L.self_attn.out_proj.bias.detach(), residual=xi,
                     drop=("pre_residual", dp, sd(_layer_site(i, SITE_PROJ))))
            ops.layernorm(xm, L.norm2.weight.detach(), L.norm2.bias.detach(), b["ln2"][i], L.norm2.eps, st[2], st[3])
            ops.gemm(b["ln2"][i], W["fc1_w"], b["h"][i], bias=L.linear1.bias.detach(), act="gelu", preact=b["u"][i],
                     drop=("post_act", dp, sd(_layer_site(i, SITE_FFN))))
            ops.gemm(b["h"][i], W["fc2_w"], xo, bias=L.linear2.bias.detach(), residual=xm,
                     drop=("pre_residual", dp, sd(_layer_site(i, SITE_FC2))))
        xf = xs[-1]
        cls_rows = xf.view(B, S * E)[:, :E]
        cs = b["cls_stat"]
        if features_only:
            out = torch.empty(B, E, dtype=F32, device=x.device)
            ops.layernorm(cls_rows, m.norm.weight.detach(), m.norm.bias.detach(), out, m.norm.eps, cs[0], cs[1])
            return out
        ops.layernorm(cls_rows, m.norm.weight.detach(), m.norm.bias.detach(), b["cls_ln"], m.norm.eps, cs[0], cs[1])
        ops.gemm(b["cls_ln"], P["h0_w"], b["h0"], bias=m.head["0"].bias.detach(), act="gelu", preact=b["u0"],
                 drop=("post_act", dp, sd(SITE_HEAD)))
        logits = torch.empty(B, m.num_classes, dtype=F32, device=x.device)
        ops.gemm(b["h0"], P["h3_w"], logits, bias=m.head["3"].bias.detach())
        return logits

    # ---- backward ----------------------------------------------------------------------------------
    def _g(self, flat, name) -> torch.Tensor:
        off, n, shape = self.layout.offsets[name]
        return flat[off:off + n]

    def _g2(self, flat, name, rows) -> torch.Tensor:
        return self._g(flat, name).view(rows, -1)

    def _stem_backward(self, P, b, B, flat, dfeat: torch.Tensor) -> None:
        b["bsums"].zero_()
        dw1 = b["scratch"].get("dw1")
        if dw1 is None:
            dw1 = b["scratch"]["dw1"] = torch.empty(64, 224, dtype=F32, device=dfeat.device)
        dw1.zero_()
        key, refs = (None, None)
        if self.reducer is None:   # the data-parallel exchange is issued between the blocks: not a fixed launch list
            key, refs = self._plan_key("stem_backward", P, b, (dfeat.data_ptr(), flat.numel()))
        ranges = {"flat": (flat.data_ptr(), flat.numel() * 4)}
        if b.get("xin") is not None:
            ranges["xin"] = (b["xin"].data_ptr(), b["xin"].numel() * 4)
        run_planned(self._plans, "stem_backward", key, refs, ranges,
                    lambda: self._stem_backward_launches(P, b, B, flat, dfeat, dw1), self.plan_replays)
        self._g(flat, "cnn_backbone.conv1.weight").view(64, 7, 7, 3).copy_(dw1.view(64, 7, 8, 4)[:, :, :7, :3])

    def _stem_backward_launches(self, P, b, B, flat, dfeat: torch.Tensor, dw1: torch.Tensor) -> None:
        """C-ABI launches and views only: this pass is recorded and replayed (vit_amd/plan.py)."""
        cb = self.m.cnn_backbone
        bs_all, fs_all = b["bsums"], b["fsums"]
        ws = b["bn_ws"] if self.ordered_bn else None
        dz = dfeat
        dz_reduced = False   # the conv that produced dz already wrote the sums of the norm it enters
        sws = self._stat_ws(dfeat.device)
        blocks = list(zip(list(cb.blocks()), P["blocks"], b["blocks"],
                          [n for n in self._block_names()]))
        for bi in range(len(blocks) - 1, -1, -1):
            blk, pk, d, pref = blocks[bi]
            s, co, ci = pk["stride"], pk["cout"], pk["cin"]
            off = d["fs_off"]
            has_ds = "ds_w" in pk
            # sums layout per block (see _stem_forward): [bn1 | bn2 | downsample bn]
            f1, f2 = fs_all[off:off + 2 * co], fs_all[off + 2 * co:off + 4 * co]
            fd = fs_all[off + 4 * co:off + 6 * co] if has_ds else None
            b1, b2 = bs_all[off:off + 2 * co], bs_all[off + 2 * co:off + 4 * co]
            bd = bs_all[off + 4 * co:off + 6 * co] if has_ds else None
            dy2 = self._scratch(b, ("dy", co, d["y2"].shape[1]), d["y2"])
            dzm = self._scratch(b, ("dzm", co, d["y2"].shape[1]), d["y2"])
            ops.bn_bwd(dz, None, d["y2"], f2, b2, blk.bn2.weight.detach(), dy2,
                       self._g(flat, pref + "bn2.weight"), self._g(flat, pref + "bn2.bias"), eps=blk.bn2.eps, relu=True,
                       dz_masked=dzm, relu_mask=d["mo"], workspace=ws, reduced=dz_reduced)
            dz_reduced = False
            ops.conv2d_wgrad(dy2, d["z1"], self._g(flat, pref + "conv2.weight").view(co, 9 * co), kh=3, kw=3, stride=1,
                             pad=1)
            dz1 = self._scratch(b, ("dz1", co, d["y2"].shape[1]), d["y2"])
            red1 = False
            if self.bnr_fused:
                # dz1 enters bn1's backward pass: its two sums come out of this conv's epilogue
                red1 = ops.conv2d_dgrad_bn_sums(dy2, pk["c2_wt"], dz1, kh=3, kw=3, stride=1, pad=1, bn_y=d["y1"],
                                                bn_relu_mask=d["m1"], bn_fwd_sums=f1, bn_bwd_sums=b1, eps=blk.bn1.eps,
                                                stat_ws=sws)
            else:
                ops.conv2d_dgrad(dy2, pk["c2_wt"], dz1, kh=3, kw=3, stride=1, pad=1)
            dy1 = dy2  # dy2 is dead now
            ops.bn_bwd(dz1, None, d["y1"], f1, b1, blk.bn1.weight.detach(), dy1,
                       self._g(flat, pref + "bn1.weight"), self._g(flat, pref + "bn1.bias"), eps=blk.bn1.eps, relu=True,
                       relu_mask=d["m1"], workspace=ws, reduced=red1)
            xin = d["in"]
            ops.conv2d_wgrad(dy1, xin, self._g(flat, pref + "conv1.weight").view(co, 9 * ci), kh=3, kw=3, stride=s,
                             pad=1)
            dxin = self._scratch(b, ("dx", ci, xin.shape[1]), xin)
            if has_ds:
                dyd = dz1  # dz1 is dead now
                ops.bn_bwd(dzm, None, d["yd"], fd, bd, blk.downsample["1"].weight.detach(), dyd,
                           self._g(flat, pref + "downsample.1.weight"), self._g(flat, pref + "downsample.1.bias"),
                           eps=blk.downsample["1"].eps, relu=False, workspace=ws)
                ops.conv2d_wgrad(dyd, xin, self._g(flat, pref + "downsample.0.weight").view(co, ci), kh=1, kw=1,
                                 stride=s, pad=0)
                if s == 2:
                    # by output parity: no zero taps for the 3x3, and the 1x1 only touches the (even, even) pixels
                    ops.conv2d_dgrad_s2(dy1, pk["c1_wt"], dxin)
                    ops.conv1x1_dgrad_s2_acc(dyd, pk["ds_wt"], dxin)
                else:
                    tmp = self._scratch(b, ("dxt", ci, xin.shape[1]), xin)
                    ops.conv2d_dgrad(dyd, pk["ds_wt"], tmp, kh=1, kw=1, stride=s, pad=0)
                    ops.conv2d_dgrad(dy1, pk["c1_wt"], dxin, kh=3, kw=3, stride=s, pad=1, residual=tmp)
            elif self.bnr_fused and bi > 0 and s == 1:
                # dxin enters bn2 of the block before this one
                pblk, ppk, pd, _ = blocks[bi - 1]
                poff, pco = pd["fs_off"], ppk["cout"]
                dz_reduced = ops.conv2d_dgrad_bn_sums(dy1, pk["c1_wt"], dxin, kh=3, kw=3, stride=1, pad=1, residual=dzm,
                                                      bn_y=pd["y2"], bn_relu_mask=pd["mo"],
                                                      bn_fwd_sums=fs_all[poff + 2 * pco:poff + 4 * pco],
                                                      bn_bwd_sums=bs_all[poff + 2 * pco:poff + 4 * pco],
                                                      eps=pblk.bn2.eps, stat_ws=sws)
            else:
                ops.conv2d_dgrad(dy1, pk["c1_wt"], dxin, kh=3, kw=3, stride=s, pad=1, residual=dzm)
            dz = dxin
            # layout order inside a block is the reverse of module order: conv1.weight closes the block
            self._maybe_reduce(flat, pref + "conv1.weight")
        # max-pool and the 7x7 stem conv (its input is the image: no dgrad)
        dyc = self._scratch(b, ("dyc",), b["c1_y"])
        if self.fused_pool_bwd:
            # the max-pool's gradient is gathered inside the two BatchNorm passes: no un-pooled gradient map (same bits)
            ops.bn_bwd_pool(dz, b["pool_arg"], b["c1_m"], b["c1_y"], fs_all[0:128], bs_all[0:128], cb.bn1.weight.detach(),
                            dyc, self._g(flat, "cnn_backbone.bn1.weight"), self._g(flat, "cnn_backbone.bn1.bias"),
                            eps=cb.bn1.eps, workspace=ws)
        else:
            dc1 = self._scratch(b, ("dc1",), b["c1_y"])
            ops.maxpool3x3s2_bwd(dz, b["pool_arg"], dc1)
            ops.bn_bwd(dc1, None, b["c1_y"], fs_all[0:128], bs_all[0:128], cb.bn1.weight.detach(), dyc,
                       self._g(flat, "cnn_backbone.bn1.weight"), self._g(flat, "cnn_backbone.bn1.bias"), eps=cb.bn1.eps,
                       relu=True, relu_mask=b["c1_m"], workspace=ws)
        if b.get("xin") is not None:
            ops.stem_conv1_wgrad_f32(dyc, b["xin"], dw1)
        else:
            ops.conv2d_wgrad(dyc, b["img"], dw1, kh=7, kw=1, stride=2, pad=0, seg_len=32)

    def _block_names(self):
        cb = self.m.cnn_backbone
        for li in range(1, 5):
            for bi in range(len(getattr(cb, f"layer{li}"))):
                yield f"cnn_backbone.layer{li}.{bi}."

    def backward(self, dout: torch.Tensor, features_only: bool, B: int) -> torch.Tensor:
        """dout: gradient of the logits [B,C] (or of the features [B,E]); returns the flat grad buffer."""
        m = self.m
        P = self._pack      # the operand copies the matching forward used
        b = self._bufs[B]
        dev = dout.device
        E, S, H, NP, C = m.embed_dim, m.num_patches + 1, m.num_heads, m.num_patches, m.num_classes
        T = B * S
        L = self.layout
        dp, base = b["drop"]
        sd = lambda site: _site_seed(base, site)  # noqa: E731
        flat = torch.zeros(L.total, dtype=F32, device=dev)
        if self.reducer is not None:
            self.reducer.begin_step()
        g, g2 = lambda n: self._g(flat, n), lambda n, r: self._g2(flat, n, r)  # noqa: E731
        dxs, dxs_bf = b["dxs"], b["dxs_bf"]
        dxs.zero_()
        dxs_bf.zero_()
        cs = b["cls_stat"]
        xs = b["xs"]
        cls_rows = xs[-1].view(B, S * E)[:, :E]
        dx_cls = dxs.view(B, S * E)[:, :E]
        dxb_cls = dxs_bf.view(B, S * E)[:, :E]
        if features_only:
            dcls = dout.to(BF16).contiguous()
        else:
            dl16 = b["dl16"]
            dl16[:, :C].copy_(dout)
            ops.wgrad(dl16, b["h0"], g2("head.3.weight", C), n_store=C)
            b["tmp16"].zero_()
            ops.colsum(dl16, b["tmp16"])
            g("head.3.bias").copy_(b["tmp16"][:C])
            ops.gemm(dl16, P["h3_wt"], b["d_h0"], gate=b["u0"], gate_mode="gelu_grad",
                     drop=("post_act", dp, sd(SITE_HEAD)))
            ops.linear_wgrad_bias(b["d_h0"], b["cls_ln"], g2("head.0.weight", E // 2), g("head.0.bias"))
            dcls = ops.gemm(b["d_h0"], P["h0_wt"], b["d_cls"])
        ops.layernorm_bwd(dcls, cls_rows, cs[0], cs[1], m.norm.weight.detach(), dx_cls, g("norm.weight"),
                          g("norm.bias"), accumulate=False, dx_bf16=dxb_cls)
        self._maybe_reduce(flat, "norm.bias")
        dx2 = dxs.view(T, E)
        # The bf16 copy of the residual gradient leaves each LayerNorm backward already multiplied by the dropout mask of
        # the site that consumes it (dropout2 of the layer below, dropout1 of this layer, pos_drop after layer 0):
        # rvit_layernorm_bwd_drop.  Only the copy the final norm wrote (cls rows of a [B, S*E] view: another row
        # numbering) still takes the separate pass.
        masked = False
        for i in reversed(range(m.num_layers)):
            Lm, W = m.transformer.layers[i], P["layers"][i]
            p = f"transformer.layers.{i}."
            xi, xm = xs[2 * i].view(T, E), xs[2 * i + 1].view(T, E)
            st = b["stat"][i]
            # ---- MLP block:  xo = xm + fc2(gelu(fc1(LN2(xm))))
            dy = dxs_bf  # gradient of the block output; dropout2 masks what reaches linear2
            if dp > 0.0 and not masked:
                dy = ops.dropout_apply(dxs_bf, b["d_e"], dp, sd(_layer_site(i, SITE_FC2)))
            ops.linear_wgrad_bias(dy, b["h"][i], g2(p + "linear2.weight", E), g(p + "linear2.bias"))
            ops.gemm(dy, W["fc2_wt"], b["d_4e"], gate=b["u"][i], gate_mode="gelu_grad",
                     drop=("post_act", dp, sd(_layer_site(i, SITE_FFN))))
            ops.linear_wgrad_bias(b["d_4e"], b["ln2"][i], g2(p + "linear1.weight", 4 * E), g(p + "linear1.bias"))
            ops.gemm(b["d_4e"], W["fc1_wt"], b["d_e"])
            ops.layernorm_bwd(b["d_e"], xm, st[2], st[3], Lm.norm2.weight.detach(), dx2, g(p + "norm2.weight"),
                              g(p + "norm2.bias"), accumulate=True, dx_bf16=dxs_bf,
                              drop=(dp, sd(_layer_site(i, SITE_PROJ))) if dp > 0.0 else None)
            # ---- attention block:  xm = xi + out_proj(attn(qkv(LN1(xi))))
            dy = dxs_bf   # dropout1's mask is on it already
            ops.linear_wgrad_bias(dy, b["att"][i], g2(p + "self_attn.out_proj.weight", E), g(p + "self_attn.out_proj.bias"))
            ops.gemm(dy, W["out_wt"], b["d_e"])
            ops.attention_bwd(b["qkv"][i], b["att"][i], b["d_e"], b["lse"][i], b["d_3e"], B, S, H, drop_p=dp,
                              drop_seed=sd(_layer_site(i, SITE_ATTN)))
            ops.linear_wgrad_bias(b["d_3e"], b["ln1"][i], g2(p + "self_attn.in_proj_weight", 3 * E), g(p + "self_attn.in_proj_bias"))
            ops.gemm(b["d_3e"], W["qkv_wt"], b["d_e"])
            ops.layernorm_bwd(b["d_e"], xi, st[0], st[1], Lm.norm1.weight.detach(), dx2, g(p + "norm1.weight"),
                              g(p + "norm1.bias"), accumulate=True, dx_bf16=dxs_bf,
                              drop=(dp, sd(_layer_site(i - 1, SITE_FC2) if i > 0 else SITE_POS)) if dp > 0.0 else None)
            masked = True
            self._maybe_reduce(flat, p + "norm1.bias")
        # ---- token assembly: x0 = pos_drop([cls + pos[0]; patch[b,p] + pos[1+p]])
        if dp > 0.0:
            ops.dropout_apply(dx2, dx2, dp, sd(SITE_POS))
            if not masked:
                ops.dropout_apply(dxs_bf, dxs_bf, dp, sd(SITE_POS))
        ops.batch_sum(dxs.view(B, S * E), g("pos_embed"))
        g("cls_token").copy_(g("pos_embed")[:E])
        limit = None
        if m.cnn_backbone is not None:
            feat = b["feat"]
            ops.wgrad(dxs_bf, feat.view(B * NP, -1), g2("patch_embed.weight", E), rows=B * NP, y_row_group=NP,
                      y_row_extra=1)
            ops.colsum(dxs_bf, g("patch_embed.bias"), rows=B * NP, row_group=NP, row_extra=1)
            if self.stem_frozen():
                # freeze_backbone (train_vit.py:310-313): no parameter below the patch embedding wants a gradient and
                # the image needs none, so the whole stem backward (2/3 of the step's FLOPs) is skipped; the stem's
                # segment of the flat buffer (its tail) stays zero and is neither exchanged nor stepped
                limit = self.stem_start()
            else:
                dfeat = self._scratch(b, ("dfeat",), feat)
                ops.gemm(dxs_bf, P["pe_wt"], dfeat.view(B * NP, -1), a_rows=B * NP, a_row_group=NP, a_row_extra=1)
                self._maybe_reduce(flat, "patch_embed.bias")
                if self.mb is not None:
                    self.mb.backward(P["mb"], b["mb"], B, flat, dfeat)
                else:
                    self._stem_backward(P, b, B, flat, dfeat)
        else:
            dpatch = dxs_bf.view(B, S, E)[:, 1:].contiguous().view(B, 14, 14, E)
            dw = torch.zeros(E, 16 * 64, dtype=F32, device=dev)
            ops.conv2d_wgrad(dpatch, b["img"], dw, kh=16, kw=1, stride=16, pad=0, seg_len=64)
            g("patch_embed.weight").view(E, 16, 16, 3).copy_(dw.view(E, 16, 16, 4)[..., :3])
            ops.colsum(dpatch.view(B * NP, E), g("patch_embed.bias"))
        self._finish_reduce(flat, limit)
        self.last_flat = flat  # vit_amd.optim.FusedAdamW reads the gradients straight from here
        return flat

    # ---- bucketed all-reduce overlapped with backward ------------------------------------------------
    def _maybe_reduce(self, flat, last_done_name: str) -> None:
        """Everything up to and including `last_done_name` (layout order) is final."""
        if self.reducer is None:
            return
        off, n, _ = self.layout.offsets[last_done_name]
        self.reducer.ready(flat, off + (n + 63) // 64 * 64)

    def _finish_reduce(self, flat, limit=None) -> None:
        if self.reducer is not None:
            self.reducer.finish(flat, limit)

    # ---- frozen backbone (train_vit.py:310-318) ------------------------------------------------------
    def stem_start(self) -> int:
        """Offset of the first cnn_backbone.* segment: the stem's parameters are the TAIL of the flat layout."""
        return min(off for n, (off, _, _) in self.layout.offsets.items() if n.startswith("cnn_backbone."))

    def stem_frozen(self) -> bool:
        cb = self.m.cnn_backbone
        if cb is None:
            return False
        flags = [p.requires_grad for p in cb.parameters()]
        if any(flags) and not all(flags):
            raise RuntimeError("cnn_backbone is partly frozen: freeze or unfreeze all of it (train_vit.py:310-318)")
        return not flags[0]

    def grads_in_param_order(self, flat) -> List[torch.Tensor]:
        out = []
        m = self.m
        for n in self.layout.param_order:
            off, numel, shape = self.layout.offsets[n]
            gseg = flat[off:off + numel]
            if len(shape) == 4 and n.startswith("cnn_backbone.") and n != "cnn_backbone.conv1.weight":
                co, ci, kh, kw = shape            # kernels write [Cout, kh, kw, Cin]
                gseg = gseg.view(co, kh, kw, ci).permute(0, 3, 1, 2)
            elif n == "cnn_backbone.conv1.weight":
                gseg = gseg.view(64, 7, 7, 3).permute(0, 3, 1, 2)
            elif n == "patch_embed.weight" and m.cnn_backbone is None:
                gseg = gseg.view(shape[0], 16, 16, 3).permute(0, 3, 1, 2)
            else:
                gseg = gseg.view(shape)
            out.append(gseg)
        return out


class HybridFunction(torch.autograd.Function):
    """logits = f(x; params) with forward and backward on librvit."""

    @staticmethod
    def forward(ctx, x, eng: TrainEngine, features_only: bool, *params):
        ctx.eng, ctx.features_only, ctx.B = eng, features_only, x.shape[0]
        return eng.forward(x, features_only)

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.eng
        with torch.cuda.device(dout.device):   # the autograd worker thread: launch on the tensors' device
            flat = eng.backward(dout.contiguous().float(), ctx.features_only, ctx.B)
        need = ctx.needs_input_grad[3:]
        return (None, None, None, *[g if n else None for g, n in zip(eng.grads_in_param_order(flat), need)])

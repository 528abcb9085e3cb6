"""PointDSC / NonLocalNet / NonLocalBlock drop-ins (reference: GMF_PointDSC/models/PointDSC.py).

Same constructor arguments, forward signatures, output dict and ``state_dict`` keys as the reference.
Sub-modules hold parameters under the reference's names; compute is the HIP library.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, packing
from ._util import WeightWatcher, handle_and_stream, params_version, require_cuda_f32
from .fusion_layer import FusionLayer
from .losses import similarity_matrix


# ------------------------------------------------------------------------------------------------
# Image encoder: ResNet-34 truncated after layer2 (models/resnet.py:195-216, Img_Encoder.py:9-19).
# Upstream of the hot path: stock convolutions, run by PyTorch-ROCm/MIOpen.
# ------------------------------------------------------------------------------------------------
class _BasicBlock(nn.Module):
    def __init__(self, inp, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inp != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inp, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class _ResNet34ToLayer2(nn.Module):
    def __init__(self, in_channels=3):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(*[_BasicBlock(64, 64) for _ in range(3)])
        self.layer2 = nn.Sequential(_BasicBlock(64, 128, 2), *[_BasicBlock(128, 128) for _ in range(3)])

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer2(self.layer1(x))


class _FusedResNet:
    """Inference form of `_ResNet34ToLayer2` on a HIP device (models/resnet.py:195-216, BasicBlock.forward :59-75): every
    BatchNorm folded into its convolution, activations NHWC.  The 3-channel stem (7x7 stride 2 + BatchNorm + ReLU + max-pool)
    is ONE HIP kernel (`gmf_stem_forward`); the 15 convolutions of layer1 / layer2 (93 % of the FLOPs) run on `gmf_conv_nhwc`
    (implicit GEMM on the f16 MFMA with split-fp16 operands, bias + residual + ReLU in its epilogue; [r5] at every batch size -
    a few images run its K-split small-grid form)."""

    def __init__(self, bb, native_convs: bool = True):
        from torch.nn.utils.fusion import fuse_conv_bn_eval
        import copy
        bb = copy.deepcopy(bb).eval()
        self.native = native_convs
        self.min_native_pixels = 0           # [r5] every batch size on the native kernels (small grids: the K-split form inside gmf_conv_nhwc)
        cl = torch.channels_last

        def fold(conv, bn):
            f = fuse_conv_bn_eval(conv.eval(), bn.eval())
            w, b = f.weight.detach(), f.bias.detach().contiguous()
            spec = {"w": w.contiguous(memory_format=cl), "b": b, "stride": conv.stride[0], "pad": conv.padding[0],
                    "cin": w.shape[1], "cout": w.shape[0], "ks": w.shape[2]}
            if native_convs and w.shape[1] % 16 == 0:
                try:
                    spec["img"] = packing.conv_image(w.cpu(), conv.stride[0]).to(w.device)
                except ValueError as e:      # folded weight outside the fp16 range: this convolution stays on MIOpen (fp32)
                    import warnings
                    warnings.warn(f"{e}  This convolution runs on MIOpen's fp32 kernel instead.", RuntimeWarning)
            return spec

        self.stem = fold(bb.conv1, bb.bn1)
        if native_convs:
            try:
                self.stem["img"] = packing.stem_image(self.stem["w"].cpu()).to(self.stem["w"].device)
            except ValueError as e:          # folded weight outside the fp16 range: the stem stays on MIOpen (fp32)
                import warnings
                warnings.warn(f"{e}  The stem convolution runs on MIOpen's fp32 kernel instead.", RuntimeWarning)
        self.blocks = []
        for layer in (bb.layer1, bb.layer2):
            for blk in layer:
                ds = fold(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
                self.blocks.append((fold(blk.conv1, blk.bn1), fold(blk.conv2, blk.bn2), ds))

    @staticmethod
    def _nhwc(t, what):
        """The HIP kernels read raw pointers as dense NHWC: anything else (an NCHW result because MIOpen's channels-last
        path is off in this torch build, a strided view) is converted here instead of being misread."""
        if not t.is_contiguous(memory_format=torch.channels_last):
            t = t.contiguous(memory_format=torch.channels_last)
        if t.dtype != torch.float32:
            raise RuntimeError(f"gmf_amd image encoder: {what} must be float32, got {t.dtype}")
        return t

    @classmethod
    def _bias_relu(cls, y, bias, residual=None):
        y = cls._nhwc(y, "convolution output")
        if residual is not None:
            residual = cls._nhwc(residual, "residual")
        B, C, H, W = y.shape
        h, st = handle_and_stream(y)
        h.call("gmf_bias_relu_nhwc", y.data_ptr(), bias.data_ptr(), None if residual is None else residual.data_ptr(),
               B * H * W, C, st)
        return y

    def _conv(self, x, c, residual=None, relu=True):
        """y = conv(x) + b (+ residual) -> ReLU, NHWC in and out."""
        B, _, H, W = x.shape
        Ho = (H + 2 * c["pad"] - c["ks"]) // c["stride"] + 1
        Wo = (W + 2 * c["pad"] - c["ks"]) // c["stride"] + 1
        # [r5] gmf_conv_nhwc picks its own form: 128 output pixels x 64 channels per workgroup at batch size, 32 x 32 with the k range
        # split over the waves for a few images (until round 4 those went to MIOpen).  MIOpen remains only for a convolution whose
        # folded weights left the fp16 range of the split operands (no "img"), and as the A/B partner of tools/image_encoder_ab.py
        if "img" in c and B * Ho * Wo >= self.min_native_pixels:
            x = self._nhwc(x, "convolution input")
            if residual is not None:
                residual = self._nhwc(residual, "residual")
            y = torch.empty((B, c["cout"], Ho, Wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)   # (no copy kernel)
            h, st = handle_and_stream(x)
            h.call("gmf_conv_nhwc", x.data_ptr(), c["img"].data_ptr(), c["b"].data_ptr(),
                   None if residual is None else residual.data_ptr(), y.data_ptr(), B, H, W, c["cin"], c["cout"], c["ks"],
                   c["stride"], 1 if relu else 0, st)
            return y
        y = F.conv2d(x, c["w"], None, c["stride"], c["pad"])
        if relu:
            return self._bias_relu(y, c["b"], residual)
        y = y + c["b"][None, :, None, None]
        return y if residual is None else y + residual

    def _stem(self, x):
        """conv1 7x7/2 + folded BatchNorm + ReLU + max-pool 3x3/2 (resnet.py:198-204) in one HIP kernel; reads the image
        through its strides (NCHW as the reference passes it, or any view), writes NHWC."""
        if "img" not in self.stem:
            return F.max_pool2d(self._conv(x.contiguous(memory_format=torch.channels_last), self.stem), 3, 2, 1)
        if x.dtype != torch.float32:
            raise RuntimeError(f"gmf_amd image encoder: images must be float32, got {x.dtype}")
        B, _, H, W = x.shape
        Hp, Wp = ((H - 1) // 2) // 2 + 1, ((W - 1) // 2) // 2 + 1
        y = torch.empty((B, 64, Hp, Wp), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        h, st = handle_and_stream(x)
        for b0 in range(0, B, 65535):
            xb, yb = x[b0:b0 + 65535], y[b0:b0 + 65535]
            h.call("gmf_stem_forward", xb.data_ptr(), xb.stride(0), xb.stride(1), xb.stride(2), xb.stride(3),
                   self.stem["img"].data_ptr(), self.stem["b"].data_ptr(), yb.data_ptr(), xb.shape[0], H, W, st)
        return y

    def __call__(self, x):
        x = self._stem(x)
        for c1, c2, ds in self.blocks:
            idt = x if ds is None else self._conv(x, ds, relu=False)
            x = self._conv(self._conv(x, c1), c2, residual=idt)
        return x


class ImageEncoder(nn.Module):
    """Keys match `encoder.image_encoder.backbone.{conv1,bn1,layer1,layer2}.*`; the reference's unused
    layer3/layer4/fc entries are ignored by the non-strict loads it uses (evaluation/test_3DMatch.py:262).

    forward(image [B,3,H,W]) -> [B,128,H',W'] (models/resnet.py:195-216).  On a HIP device in eval mode the pass runs as
    `_FusedResNet` (BatchNorms folded, NHWC, the fused stem kernel and the native layer1 / layer2 convolutions) and, per input
    shape, as a captured HIP graph (~75 launches of a few microseconds of work each; `graph = False` keeps it eager).  In
    train() mode it is the stock torch module (it trains as one: DESIGN section 7)."""

    graph = True

    def __init__(self):
        super().__init__()
        self.backbone = _ResNet34ToLayer2(3)

    def fused(self):
        """eval-mode form of the ResNet: every BatchNorm folded into its convolution, weights and activations NHWC, and the
        bias + residual + ReLU that follow each MIOpen convolution done by one HIP pass (`gmf_bias_relu_nhwc`) instead of
        three element-wise kernels.  Rebuilt when the weights change."""
        ver = params_version(self)
        if getattr(self, "_fused_version", None) != ver:
            object.__setattr__(self, "_fused", _FusedResNet(self.backbone))   # not a sub-module: keeps the state_dict surface
            self._fused_version = ver
        return self._fused

    def forward(self, x):
        if self.training or not x.is_cuda:
            return self.backbone(x)
        # eval() with autograd on and something to differentiate (fine-tuning with frozen BatchNorm statistics): the fused,
        # graph-replayed pass below returns tensors without a grad_fn - take the stock modules, which do carry one
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.backbone.parameters())):
            return self.backbone(x)
        enc = self.fused()
        return self._graphed(enc, x) if self.graph else enc(x)

    def _graphed(self, enc, image):
        key = (tuple(image.shape), tuple(image.stride()), image.device, self._fused_version)
        cache = self.__dict__.setdefault("_graphs", {})
        ent = cache.get(key)
        if ent is None:
            static_in = image.detach().clone()
            with torch.no_grad():
                for _ in range(2):                       # MIOpen picks its algorithms and workspaces outside the capture
                    enc(static_in)                       # (a kernel or shape error surfaces here, eagerly, as itself)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g):
                        static_out = enc(static_in)
                    ent = (g, static_in, static_out)
                except RuntimeError as e:
                    # Only "this operation cannot be captured" (hipErrorStreamCapture*: a library call inside the pass that
                    # allocates or synchronises) keeps the pass eager - said once per shape; anything else is a real error.
                    msg = str(e)
                    if "capture" not in msg.lower():
                        raise
                    import warnings
                    warnings.warn(f"gmf_amd: HIP-graph capture of the image encoder failed for input {tuple(image.shape)} "
                                  f"({msg.splitlines()[0]}); running it eagerly.", RuntimeWarning)
                    torch.cuda.synchronize()
                    ent = False
            if len(cache) > 8:
                cache.clear()
            cache[key] = ent
        if ent is False:
            return enc(image)
        g, static_in, static_out = ent
        static_in.copy_(image)
        g.replay()
        return static_out.clone()


# ------------------------------------------------------------------------------------------------
class NonLocalBlock(nn.Module):
    """PointDSC.py:10-74.  forward(feat [B,C,N], attention [B,N,N], image_feat [B,T,C]) -> [B,C,N]."""

    def __init__(self, num_channels=128, num_heads=1):
        super().__init__()
        if num_channels < 2 or num_channels % 2 or num_channels % num_heads:
            raise NotImplementedError("gmf_amd.NonLocalBlock: num_channels must be even and a multiple of num_heads (PointDSC.py:11-25,56-58)")
        c = num_channels
        self.fc_message = nn.Sequential(
            nn.Conv1d(c, c // 2, 1), nn.BatchNorm1d(c // 2), nn.ReLU(inplace=True),
            nn.Conv1d(c // 2, c // 2, 1), nn.BatchNorm1d(c // 2), nn.ReLU(inplace=True),
            nn.Conv1d(c // 2, c, 1))
        self.projection_q = nn.Conv1d(c, c, 1)
        self.projection_k = nn.Conv1d(c, c, 1)
        self.projection_v = nn.Conv1d(c, c, 1)
        self.num_channels, self.head = c, num_heads
        self.fusion_layer_2 = FusionLayer(dim=c, depth=0, latent_dim=c, cross_heads=1, latent_heads=8,
                                          cross_dim_head=c // 2, latent_dim_head=c // 2, pe=True)
        self._packed, self._packed_version = None, None

    def _weights(self, device):
        ver = (params_version(self), str(device))
        if self._packed is None or self._packed_version != ver:
            if self.training:
                raise RuntimeError("gmf_amd.NonLocalBlock: only eval() mode is implemented (BatchNorm uses running stats)")
            pre = "encoder.blocks.NonLocal_layer_0."
            sd = {pre + k: v.detach() for k, v in self.state_dict().items()}
            self._packed = packing.PackedEncoder(sd, 1, device, standalone_block=True)
            self._packed_version = ver
        return self._packed

    def forward(self, feat, attention, image_feat):
        feat = require_cuda_f32(feat, "feat")
        attention = require_cuda_f32(attention, "attention").contiguous()
        image_feat = require_cuda_f32(image_feat, "image_feat").contiguous()
        B, Cc, N = feat.shape
        T = image_feat.shape[1]
        if attention.shape != (B, N, N):
            raise RuntimeError(f"gmf_amd.NonLocalBlock: attention must be [B,N,N], got {tuple(attention.shape)}")
        if self.num_channels != 128 or self.head != 1:
            return self._forward_general(feat, attention, image_feat)
        pw = self._weights(feat.device)
        h, st = handle_and_stream(feat)
        tiles, tt = (N + 31) // 32, (T + 31) // 32
        fimg = torch.empty(B * tiles * 4096, device=feat.device)
        oimg = torch.empty(B * tiles * 4096, device=feat.device)
        timg = torch.empty(B * tt * 4096, device=feat.device)
        h.call("gmf_pack_rows_p32", feat.data_ptr(), feat.stride(0), feat.stride(2), feat.stride(1), B, N, Cc, fimg.data_ptr(), st)
        h.call("gmf_pack_rows_p32", image_feat.data_ptr(), T * Cc, Cc, 1, B, T, Cc, timg.data_ptr(), st)
        h.call("gmf_nonlocal_block_forward", pw.struct, 0, 0, fimg.data_ptr(), None, attention.data_ptr(),
               timg.data_ptr(), oimg.data_ptr(), B, N, T, st)
        out = torch.empty((B, Cc, N), device=feat.device, dtype=torch.float32)
        h.call("gmf_unpack_rows_p32", oimg.data_ptr(), B, N, Cc, out.data_ptr(), out.stride(0), out.stride(2), out.stride(1), st)
        return out


    # -- [r5] widths / head counts GMF never instantiates (PointDSC.py:11: num_channels, num_heads are constructor arguments) ----------
    def _folded(self):
        """fc_message with its two eval-mode BatchNorms folded into the convolutions before them, cached until a parameter changes."""
        ver = params_version(self)
        if getattr(self, "_fold_version", None) != ver:
            if self.training:
                raise RuntimeError("gmf_amd.NonLocalBlock: only eval() mode is implemented (BatchNorm uses running stats)")
            sd = {k: v.detach() for k, v in self.fc_message.state_dict().items()}
            Wa, ba = packing.fold_bn(sd["0.weight"][:, :, 0], sd["0.bias"], sd, "1.")
            Wb, bb = packing.fold_bn(sd["3.weight"][:, :, 0], sd["3.bias"], sd, "4.")
            self._fold = tuple(t.contiguous() for t in (Wa, ba, Wb, bb, sd["6.weight"][:, :, 0], sd["6.bias"]))
            self._fold_version = ver
        return self._fold

    def _forward_general(self, feat, attention, image_feat):
        """PointDSC.py:40-74 for any (num_channels, num_heads), forward only, from the library's HIP primitives (`gmf_gemm_f32`,
        `gmf_softmax_rows` with the spatial-consistency matrix as the logits' multiplier): rows are token-major [B N, C]."""
        from . import train as P
        with torch.no_grad():
            B, C, N = feat.shape
            hds, d = self.head, C // self.head
            f = feat.permute(0, 2, 1).contiguous().reshape(B * N, C)
            q = P.gemm(f, self.projection_q.weight[:, :, 0], tb=True, bias=self.projection_q.bias)
            k = P.gemm(f, self.projection_k.weight[:, :, 0], tb=True, bias=self.projection_k.bias)
            v = P.gemm(f, self.projection_v.weight[:, :, 0], tb=True, bias=self.projection_v.bias)
            S = torch.empty((hds, B, N, N), device=f.device, dtype=torch.float32)
            msg = torch.empty((B * N, C), device=f.device, dtype=torch.float32)
            for hd in range(hds):                       # 'bhco, bhci -> bhoi' / sqrt(C / heads), one batched product per head
                P.gemm(q, k, tb=True, out=S, m=N, n=N, k=d, lda=C, ldb=C, ldc=N, a_off=hd * d, b_off=hd * d, c_off=hd * B * N * N,
                       batch=B, sa=N * C, sb=N * C, sc=N * N)
            W = torch.empty_like(S)
            for hd in range(hds):                       # softmax(attention * S / sqrt(d)) per row (PointDSC.py:62)
                W[hd] = P.softmax_rows(S[hd].reshape(B * N, N), d ** -0.5, mul=attention.reshape(B * N, N)).reshape(B, N, N)
            for hd in range(hds):                       # 'bhoi, bhci -> bhco'
                P.gemm(W, v, out=msg, m=N, n=d, k=N, lda=N, ldb=C, ldc=C, a_off=hd * B * N * N, b_off=hd * d, c_off=hd * d,
                       batch=B, sa=N * N, sb=N * C, sc=N * C)
            Wa, ba, Wb, bb, Wc, bc = self._folded()
            m1 = P.gemm(msg, Wa, tb=True, bias=ba, relu=True)
            m2 = P.gemm(m1, Wb, tb=True, bias=bb, relu=True)
            fus = self.fusion_layer_2(image_feat, queries_encoder=f.reshape(B, N, C)).reshape(B * N, C)
            out = P.gemm(m2, Wc, tb=True, bias=bc, residual=fus)          # message + image_feat (PointDSC.py:73)
            return out.reshape(B, N, C).permute(0, 2, 1).contiguous()


class NonLocalNet(nn.Module):
    """PointDSC.py:77-143.  Parameter container + image-token front end; the layer loop lives in
    `gmf_encoder_forward` (called by PointDSC.forward)."""

    def __init__(self, in_dim=6, num_layers=6, num_channels=128):
        super().__init__()
        if num_channels != 128:
            raise NotImplementedError("gmf_amd.NonLocalNet: HIP kernels are built for num_channels=128")
        if not 1 <= in_dim <= 8:
            raise NotImplementedError("gmf_amd.NonLocalNet: the layer0 kernel takes 1 <= in_dim <= 8 (GMF uses 6)")
        self.num_layers = num_layers
        self.blocks = nn.ModuleDict()
        self.layer0 = nn.Conv1d(in_dim, num_channels, 1, bias=True)
        self.image_encoder = ImageEncoder()
        c = num_channels
        self.fusion_layer_1 = FusionLayer(dim=c, depth=0, latent_dim=c, cross_heads=1, latent_heads=8,
                                          cross_dim_head=c // 2, latent_dim_head=c // 2)
        for i in range(num_layers):
            self.blocks[f"PointCN_layer_{i}"] = nn.Sequential(nn.Conv1d(c, c, 1, bias=True), nn.BatchNorm1d(c), nn.ReLU(inplace=True))
            self.blocks[f"NonLocal_layer_{i}"] = NonLocalBlock(c)

    def _fused_image_encoder(self):
        """The image encoder's eval-mode form (`ImageEncoder.fused`)."""
        return self.image_encoder.fused()

    @property
    def graph_image_encoder(self):
        return self.image_encoder.graph

    @graph_image_encoder.setter
    def graph_image_encoder(self, v):
        self.image_encoder.graph = bool(v)

    def image_tokens(self, image):
        """[B,3,H,W] -> [B,H'*W',128] (PointDSC.py:129-131); the encoder itself picks its form (`ImageEncoder.forward`)."""
        return self.image_encoder(image).flatten(2).permute(0, 2, 1).contiguous()


class PointDSC(nn.Module):
    """PointDSC.py:146-528.

    forward(data) takes the reference's dict (corr_pos, src_keypts, tgt_keypts, p_image, q_image, optional
    key 'testing' whose PRESENCE selects test mode) and returns {"final_trans", "final_labels", "M"}.
    Extensions: image tokens may be given directly as data["p_tokens"], data["q_tokens"] [B,T,128]
    (skips the ResNet); test mode supports B > 1 (= B independent B=1 calls of the reference, which asserts
    B == 1 at PointDSC.py:279,504); the inlier logits are kept in `self.last_logits` in both modes.
    """

    def __init__(self, in_dim=6, num_layers=6, num_channels=128, num_iterations=10, ratio=0.1,
                 inlier_threshold=0.10, sigma_d=0.10, k=40, nms_radius=0.10):
        super().__init__()
        if not 1 <= k <= 64:
            raise NotImplementedError("gmf_amd.PointDSC: the seed kernels hold one neighbour per lane: 1 <= k <= 64 (GMF uses 40)")
        if not 1 <= num_iterations <= 64:
            raise NotImplementedError("gmf_amd.PointDSC: 1 <= num_iterations <= 64 (GMF uses 10)")
        self.num_iterations = num_iterations
        self.ratio = ratio
        self.num_channels = num_channels
        self.inlier_threshold = inlier_threshold
        self.sigma = nn.Parameter(torch.Tensor([1.0]).float(), requires_grad=True)
        self.sigma_spat = nn.Parameter(torch.Tensor([sigma_d]).float(), requires_grad=False)
        self.k = k
        self.nms_radius = nms_radius
        self.encoder = NonLocalNet(in_dim=in_dim, num_layers=num_layers, num_channels=num_channels)
        self.classification = nn.Sequential(
            nn.Conv1d(num_channels, 32, 1, bias=True), nn.ReLU(inplace=True),
            nn.Conv1d(32, 32, 1, bias=True), nn.ReLU(inplace=True),
            nn.Conv1d(32, 1, 1, bias=True))
        for m in self.modules():        # PointDSC.py:183-188
            if isinstance(m, (nn.Conv1d, nn.Linear)):
                nn.init.xavier_normal_(m.weight, gain=1)
            elif isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._packed, self._packed_version = None, None
        self._watch = WeightWatcher(self, skip_prefix="encoder.image_encoder.")
        self.last_logits = None
        self.last_features = None

    # -- packed weights ---------------------------------------------------------------------------
    def _hot_state(self):
        return {k: v for k, v in self.state_dict().items() if not k.startswith("encoder.image_encoder.")}

    def _apply(self, fn, *a, **kw):            # .to() / .cuda() / .float(): tensors may be replaced
        out = super()._apply(fn, *a, **kw)
        if hasattr(self, "_watch"):
            self._watch.invalidate()
        return out

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        self._watch.invalidate()
        self._packed = None
        return out

    def _weights(self, device):
        device = torch.device(device)                  # ("cuda:0" and device("cuda", 0) are the same cache entry)
        key = (self._watch.version(), device)
        if self._packed is None or self._packed_version != key:
            self._packed = packing.PackedEncoder(self._hot_state(), self.encoder.num_layers, device)
            self._packed_version = key
        return self._packed

    def set_precision(self, mode: str):
        """Numerics of the eval-mode encoder on this module's device: "parity" (default - fp32-equivalent split-fp16 products,
        logits within 1e-4 of the reference) or "throughput" (on large grids the spatial-consistency
        attention multiplies plain fp16 operands and streams the compat matrix as fp16; measured deviation from the parity
        mode at 32 x 5000: logits 1.6e-4, identical inlier labels, 1.46x the throughput) or "throughput_max" (the layer's linear
        stages on plain fp16 operands as well: logits 4e-2, 98-100 % identical labels, 1.6x).  The setting belongs to THIS
        module: its forward sets the library handle's "precision" knob for the duration of the call and puts the previous
        value back, so other modules on the device keep their own numerics.
        "parity_strict": the parity mode with every product of the attention's P V on the f16 pipe ("pv_fp8" = 0) whatever the
        device-side guard would decide - the form to pick when a deployment wants the three-product arithmetic in every layer
        (about 3 % more time per step)."""
        levels = {"parity": 0, "parity_strict": 0, "throughput": 1, "throughput_max": 2}
        if mode not in levels:
            raise ValueError("gmf_amd.PointDSC.set_precision: mode must be 'parity', 'parity_strict', 'throughput' or 'throughput_max'")
        self._precision = levels[mode]
        self._strict = mode == "parity_strict"

    def _call_tuned(self, h, fn, *args):
        """One C call under this module's numerics mode: the knobs that differ from the handle's are set for the call and the
        handle's own values PUT BACK afterwards (a handle-level gmf_set_tuning made through the C knob survives).  The
        calls take the handle's lock one after the other: two threads driving ONE handle with different modes must serialise
        themselves."""
        import ctypes as C
        want = {}
        if getattr(self, "_precision", 0):
            want[b"precision"] = self._precision
        if getattr(self, "_strict", False):
            want[b"pv_fp8"] = 0
        prev = {}
        for knob, val in want.items():
            cur = C.c_int(0)
            h.call("gmf_get_tuning", knob, C.byref(cur))
            if cur.value != val:
                prev[knob] = cur.value
                h.call("gmf_set_tuning", knob, val)
        try:
            h.call(fn, *args)
        finally:
            for knob, val in prev.items():
                h.call("gmf_set_tuning", knob, val)

    # -- encoder: logits + normalised features ----------------------------------------------------
    def encode(self, corr_pos, src_keypts, tgt_keypts, p_tokens, q_tokens, want_features=False):
        if self.training:
            raise RuntimeError("gmf_amd.PointDSC.encode is the eval() inference path (BatchNorm folded from running statistics); in train() "
                               "mode call the module with autograd enabled (the differentiable forward), or call eval() first")
        corr_pos = require_cuda_f32(corr_pos, "corr_pos").contiguous()
        src = require_cuda_f32(src_keypts, "src_keypts").contiguous()
        tgt = require_cuda_f32(tgt_keypts, "tgt_keypts").contiguous()
        p_tokens = require_cuda_f32(p_tokens, "p_tokens").contiguous()
        q_tokens = require_cuda_f32(q_tokens, "q_tokens").contiguous()
        B, N, D = corr_pos.shape
        T = p_tokens.shape[1]
        if D != self.encoder.layer0.weight.shape[1]:
            raise RuntimeError(f"gmf_amd.PointDSC: corr_pos last dim {D} != in_dim {self.encoder.layer0.weight.shape[1]}")
        if p_tokens.shape != (B, T, 128) or q_tokens.shape != (B, T, 128):
            raise RuntimeError("gmf_amd.PointDSC: image tokens must be [B,T,128] for both images")
        if src.shape != (B, N, 3) or tgt.shape != (B, N, 3):
            raise RuntimeError("gmf_amd.PointDSC: src_keypts/tgt_keypts must be [B,N,3]")
        pw = self._weights(corr_pos.device)
        dev = corr_pos.device
        logits = torch.empty((B, N), device=dev)
        feat_n = torch.empty((B, N, 128), device=dev)
        feat = torch.empty((B, N, 128), device=dev) if want_features else None
        h, st = handle_and_stream(corr_pos, check=True)     # (raises if an earlier forward on this device produced NaN / inf)
        # the module-local numerics mode (set_precision) applies to this call only
        self._call_tuned(h, "gmf_encoder_forward", pw.struct, corr_pos.data_ptr(), src.data_ptr(), tgt.data_ptr(),
                         p_tokens.data_ptr(), q_tokens.data_ptr(), B, N, T, logits.data_ptr(), feat_n.data_ptr(),
                         None if feat is None else feat.data_ptr(), st)
        return logits, feat_n, feat

    # -- pose head --------------------------------------------------------------------------------
    def _pose_params(self, N, testing, sigmas=None, device=None):
        S = int(N * self.ratio)
        k = min(self.k, N - 1)
        pp = _lib.PoseParams()
        pp.num_seeds, pp.k, pp.num_iterations = S, k, self.num_iterations
        pp.use_nms = 1 if testing else 0
        pp.refine_iters = 20 if testing else 0
        if sigmas is None:
            pw = self._weights(device)              # sigma / sigma_spat were read when the weights were packed
            sigmas = (pw.sigma, pw.sigma_d)
        pp.sigma, pp.sigma_d = sigmas               # (the training path passes them: its weights change every step, no packing)
        pp.inlier_threshold, pp.nms_radius = float(self.inlier_threshold), float(self.nms_radius)
        pp.refine_threshold = 0.10 if self.inlier_threshold == 0.10 else 1.2      # PointDSC.py:505-508
        return pp, S, k

    def pose_head(self, feat_n, src_keypts, tgt_keypts, logits, testing, seeds=None, return_aux=False, sigmas=None):
        B, N, _ = feat_n.shape
        pp, S, k = self._pose_params(N, testing, sigmas, feat_n.device)
        dev = feat_n.device
        final_T = torch.empty((B, 4, 4), device=dev)
        labels = torch.empty((B, N), device=dev)
        seeds_out = torch.empty((B, S), device=dev, dtype=torch.int32)
        aux = {}
        if return_aux:
            aux = {"knn_idx": torch.empty((B, S, k), device=dev, dtype=torch.int32),
                   "seed_trans": torch.empty((B, S, 4, 4), device=dev), "fitness": torch.empty((B, S), device=dev)}
        seeds_in = None if seeds is None else seeds.to(torch.int32).contiguous()
        h, st = handle_and_stream(feat_n)
        h.call("gmf_pose_head", pp, feat_n.data_ptr(), src_keypts.data_ptr(), tgt_keypts.data_ptr(), logits.data_ptr(),
               None if seeds_in is None else seeds_in.data_ptr(), B, N, final_T.data_ptr(), labels.data_ptr(),
               seeds_out.data_ptr(), aux["knn_idx"].data_ptr() if return_aux else None,
               aux["seed_trans"].data_ptr() if return_aux else None, aux["fitness"].data_ptr() if return_aux else None, st)
        aux["seeds"] = seeds_out
        return final_T, labels, aux

    def _forward_train(self, data):
        """train() mode with autograd enabled (libs/trainer.py:131): the differentiable forward - HIP training primitives
        behind torch.autograd.Functions (gmf_amd/train.py), BatchNorm in training mode (batch statistics, running statistics
        updated).  Returns what the reference's non-test forward returns: `final_labels` = the inlier logits and `M` carry the
        autograd graph (the two losses the reference trains with by default, config_3DMatch.py:50-52), and so does `final_trans`
        (the best seed's hypothesis: gmf_pose_head forward, gmf_pose_head_backward to the features and sigma) for
        `TransformationLoss` (weight 0 in the reference's default configuration)."""
        from . import train as T
        corr_pos = require_cuda_f32(data["corr_pos"], "corr_pos")
        src = require_cuda_f32(data["src_keypts"], "src_keypts").contiguous()
        tgt = require_cuda_f32(data["tgt_keypts"], "tgt_keypts").contiguous()
        if "p_tokens" in data:
            p_tok, q_tok = data["p_tokens"], data["q_tokens"]
        else:       # the ResNet image encoder trains as a stock torch module (MIOpen convolutions, torch autograd)
            enc = self.encoder.image_encoder
            p_tok = enc(data["p_image"]).flatten(2).permute(0, 2, 1).contiguous()
            q_tok = enc(data["q_image"]).flatten(2).permute(0, 2, 1).contiguous()
        B, N, _ = corr_pos.shape
        if getattr(self, "sigma_on_device", False):
            # [r5] no host read in the step (a captured HIP graph must not have one): sigma stays on the device for every kernel
            # that uses it; sigma_spat is not trained (PointDSC.py:165: requires_grad = False) and is read once
            if getattr(self, "_sigma_d_host", None) is None:
                self._sigma_d_host = float(self.sigma_spat)
            sigma, sigma_d = T.SIGMA_ON_DEVICE, self._sigma_d_host
        else:
            sigma, sigma_d = float(self.sigma.detach()), float(self.sigma_spat)      # the step's one host read of the two scalars
        compat = T.compat_dense(src, tgt, sigma_d)              # PointDSC.py:216-221 (under no_grad in the reference too)
        feat = T.encoder_train(self.encoder, corr_pos, compat, p_tok, q_tok)           # [B, N, C]
        feat_n = T.normalize_rows(feat.reshape(B * N, -1)).reshape(B, N, -1)           # PointDSC.py:229
        M = T.similarity_matrix_train(feat_n, self.sigma, sigma)                       # PointDSC.py:231-234
        logits = T.classifier_train(self.classification, feat)                         # PointDSC.py:241
        self.last_logits, self.last_features = logits, feat_n
        # PointDSC.py:246-252: top-S seeds, per-seed hypotheses, the best one - differentiable with respect to the features and sigma
        final_trans = T.pose_head_train(self, feat_n, self.sigma, src, tgt, logits.detach().contiguous(), (sigma, sigma_d))
        return {"final_trans": final_trans, "final_labels": logits, "M": M}

    # -- ragged batches: pairs with their own numbers of correspondences in ONE launch ----------------------------------------
    def forward_ragged(self, data):
        """Test-mode forward for B pairs that each have their OWN N - what the reference's evaluation loop feeds one pair at a
        time (evaluation/test_3DMatch.py:24-119; PointDSC.py:279,504 assert B == 1), here in one launch at batch throughput.
        `data`: "corr_pos", "src_keypts", "tgt_keypts" as LISTS of B tensors [n_b, 6] / [n_b, 3] (or already packed
        [sum n, .] tensors together with "n_points", a list of B ints); "p_tokens" / "q_tokens" [B, T, 128] (or "p_image" /
        "q_image").  Returns "final_trans" [B, 4, 4], "final_labels" and "logits" as lists of B tensors [n_b]; every pair's
        result equals its own B = 1 call.  `last_logits` / `last_features` hold the packed tensors."""
        if self.training:
            raise RuntimeError("gmf_amd.PointDSC.forward_ragged is the eval() test-mode path")
        if getattr(self, "_precision", 0):
            raise RuntimeError("gmf_amd.PointDSC.forward_ragged: ragged batches run the parity numerics only - the throughput modes "
                               "(set_precision) exist for uniform batches; call set_precision('parity') or pass a uniform batch")
        import ctypes as C
        cp, sk, tk = data["corr_pos"], data["src_keypts"], data["tgt_keypts"]
        if isinstance(cp, (list, tuple)):
            n_points = [int(t.shape[0]) for t in cp]
            if [int(t.shape[0]) for t in sk] != n_points or [int(t.shape[0]) for t in tk] != n_points:
                raise RuntimeError("gmf_amd.PointDSC.forward_ragged: corr_pos / src_keypts / tgt_keypts disagree on the pairs' sizes")
            cp, sk, tk = torch.cat(list(cp)), torch.cat(list(sk)), torch.cat(list(tk))
        else:
            n_points = [int(n) for n in data["n_points"]]
        cp = require_cuda_f32(cp, "corr_pos").contiguous()
        sk = require_cuda_f32(sk, "src_keypts").contiguous()
        tk = require_cuda_f32(tk, "tgt_keypts").contiguous()
        B, total = len(n_points), sum(n_points)
        if cp.shape != (total, self.encoder.layer0.weight.shape[1]) or sk.shape != (total, 3) or tk.shape != (total, 3):
            raise RuntimeError("gmf_amd.PointDSC.forward_ragged: packed tensors must be [sum n, 6] / [sum n, 3]")
        if "p_tokens" in data:
            p_tok, q_tok = data["p_tokens"], data["q_tokens"]
        else:
            with torch.no_grad():
                tok = self.encoder.image_tokens(torch.cat([data["p_image"], data["q_image"]]))
                p_tok, q_tok = tok[:B], tok[B:]
        p_tok = require_cuda_f32(p_tok, "p_tokens").contiguous()
        q_tok = require_cuda_f32(q_tok, "q_tokens").contiguous()
        T = p_tok.shape[1]
        if p_tok.shape != (B, T, 128) or q_tok.shape != (B, T, 128):
            raise RuntimeError("gmf_amd.PointDSC.forward_ragged: image tokens must be [B,T,128] for both images")
        dev = cp.device
        pw = self._weights(dev)
        npts = (C.c_int * B)(*n_points)
        logits = torch.empty(total, device=dev)
        feat_n = torch.empty((total, 128), device=dev)
        final_T = torch.empty((B, 4, 4), device=dev)
        labels = torch.empty(total, device=dev)
        pp, _, _ = self._pose_params(max(n_points), True, None, dev)
        h, st = handle_and_stream(cp, check=True)
        self._call_tuned(h, "gmf_encoder_forward_ragged", pw.struct, cp.data_ptr(), sk.data_ptr(), tk.data_ptr(), p_tok.data_ptr(),
                         q_tok.data_ptr(), npts, B, T, logits.data_ptr(), feat_n.data_ptr(), None, st)
        h.call("gmf_pose_head_ragged", pp, float(self.ratio), feat_n.data_ptr(), sk.data_ptr(), tk.data_ptr(), logits.data_ptr(), npts, B,
               final_T.data_ptr(), labels.data_ptr(), None, None, None, None, st)
        self.last_logits, self.last_features = logits, feat_n
        return {"final_trans": final_T, "final_labels": list(torch.split(labels, n_points)), "logits": list(torch.split(logits, n_points)),
                "M": None}

    def forward(self, data):
        if isinstance(data.get("corr_pos"), (list, tuple)) or "n_points" in data:
            return self.forward_ragged(data)
        if self.training and torch.is_grad_enabled():
            if "testing" in data.keys():
                raise RuntimeError("gmf_amd.PointDSC: test mode (`testing` key) in train() mode with autograd enabled - call eval() "
                                   "or torch.no_grad() for inference")
            return self._forward_train(data)
        corr_pos, src_keypts, tgt_keypts = data["corr_pos"], data["src_keypts"], data["tgt_keypts"]
        testing = "testing" in data.keys()
        if "p_tokens" in data:
            p_tok, q_tok = data["p_tokens"], data["q_tokens"]
        else:
            with torch.no_grad():
                p_img, q_img = data["p_image"], data["q_image"]
                if p_img.shape == q_img.shape:          # one pass over both images of every pair: half the launches
                    tok = self.encoder.image_tokens(torch.cat([p_img, q_img]))
                    p_tok, q_tok = tok[:p_img.shape[0]], tok[p_img.shape[0]:]
                else:
                    p_tok, q_tok = self.encoder.image_tokens(p_img), self.encoder.image_tokens(q_img)
        with torch.no_grad():
            logits, feat_n, feat = self.encode(corr_pos, src_keypts, tgt_keypts, p_tok, q_tok, want_features=not testing)
            self.last_logits, self.last_features = logits, feat_n
            src = src_keypts.contiguous()
            tgt = tgt_keypts.contiguous()
            final_trans, labels, _ = self.pose_head(feat_n, src, tgt, logits, testing)
            M = None
            if not testing:
                M = similarity_matrix(feat_n, self._weights(feat_n.device).sigma)       # PointDSC.py:231-234
        return {"final_trans": final_trans, "final_labels": labels if testing else logits, "M": M}

    # reference-named helpers, batched --------------------------------------------------------------
    def pick_seeds(self, dists, scores, R, max_num, src_keypts=None):
        """PointDSC.py:268-286.  The HIP kernel needs the key points, not the N x N distance matrix."""
        if src_keypts is None:
            raise RuntimeError("gmf_amd.PointDSC.pick_seeds: pass src_keypts=...; the N x N `dists` matrix is never materialised")
        src = require_cuda_f32(src_keypts, "src_keypts").contiguous()
        scores = require_cuda_f32(scores, "scores").contiguous()
        B, N = scores.shape
        out = torch.empty((B, max_num), device=src.device, dtype=torch.int32)
        h, st = handle_and_stream(src)
        h.call("gmf_pick_seeds", src.data_ptr(), scores.data_ptr(), B, N, float(R), 1, int(max_num), out.data_ptr(), st)
        return out.long()

    def post_refinement(self, initial_trans, src_keypts, tgt_keypts, weights=None):
        """PointDSC.py:493-528, for any B."""
        T = require_cuda_f32(initial_trans, "initial_trans").contiguous()
        src = require_cuda_f32(src_keypts, "src_keypts").contiguous()
        tgt = require_cuda_f32(tgt_keypts, "tgt_keypts").contiguous()
        B, N = src.shape[0], src.shape[1]
        out = torch.empty_like(T)
        thr = 0.10 if self.inlier_threshold == 0.10 else 1.2
        h, st = handle_and_stream(T)
        h.call("gmf_post_refinement", T.data_ptr(), src.data_ptr(), tgt.data_ptr(), B, N, thr, 20, out.data_ptr(), st)
        return out

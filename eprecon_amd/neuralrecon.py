"""The drop-in boundary — mirror of models/neuralrecon.py:19-86 (NeuralRecon):
normalise the 9 images, run the two MnasMulti backbones per view (PyTorch-ROCm), run the HIP 3D path
(NeuConNet), and at test time fuse TSDF + panoptic ids into the scene map (fuse_to_global).

forward(inputs: dict, save_mesh=False, training=True) -> (outputs: dict, loss_dict: dict), called the
way main.py:430-436 calls it.  With autograd enabled loss_dict carries the training losses and their LW-weighted
'total_loss' (models/neuralrecon.py:76-84); under torch.no_grad() it carries zeros.
"""
import torch
import torch.nn as nn

from . import _lib
from .backbone import MnasMulti
from .gru_fusion import GRUFusion
from .neucon_network import NeuConNet

PIXEL_MEAN = [103.53, 116.28, 123.675]  # config/default.py:59-61
PIXEL_STD = [1.0, 1.0, 1.0]
LOSS_WEIGHTS = [1.0, 0.8, 0.64, 0.8]    # config/test.yaml:42


def tocuda(obj, device):
    """utils.py:74-81: recursive host -> device copy of the input dict"""
    if isinstance(obj, torch.Tensor):
        return obj.to(device)
    if isinstance(obj, dict):
        # (keys ending in _host are the caller's host-side copies — the volume origins GRUFusion reads on the host — and stay there)
        return {k: (v if isinstance(k, str) and k.endswith("_host") else tocuda(v, device)) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)) and obj and not isinstance(obj[0], str):
        return [tocuda(v, device) for v in obj]
    return obj


class NeuralRecon(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        # the reference hands over the whole yacs node and reads cfg.MODEL (models/neuralrecon.py:22-33);
        # a plain ModelCfg is accepted as well
        train_cfg = getattr(cfg, "TRAIN", None)     # config/default.py:36-38
        cfg = getattr(cfg, "MODEL", cfg)
        self.cfg = cfg
        self.register_buffer("pixel_mean", torch.tensor(PIXEL_MEAN).view(-1, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.tensor(PIXEL_STD).view(-1, 1, 1), persistent=False)
        self.n_scales = len(cfg.THRESHOLDS) - 1
        self.backbone2d = MnasMulti(float(cfg.ALPHA))
        self.backbone_occ_pano = MnasMulti(float(cfg.ALPHA))
        self.neucon_net = NeuConNet(cfg)
        self.fuse_to_global = GRUFusion(cfg, direct_substitute=True, trianing=False)
        self.only_train_init = bool(getattr(train_cfg, "ONLY_INIT", False))
        self.only_train_occ = bool(getattr(train_cfg, "ONLY_OCC", False))
        self.init_overlap_count = 0
        self.loss_weights = list(getattr(cfg, "LW", LOSS_WEIGHTS))
        self.batch_views = True   # False: always the reference's per-view backbone loop
        self.two_backbone_streams = True
        self._side = None

    def normalizer(self, x):
        return (x - self.pixel_mean.type_as(x)) / self.pixel_std.type_as(x)

    def forward(self, inputs, save_mesh=False, training=True):
        dev = self.pixel_mean.device
        # the data loader builds the volume origins on the host (datasets/transforms.py:250-260): GRUFusion's scene
        # bookkeeping reads them there instead of synchronising the device once per fragment
        host = {k + "_host": inputs[k] for k in ("vol_origin", "vol_origin_partial")
                if torch.is_tensor(inputs.get(k)) and not inputs[k].is_cuda and k + "_host" not in inputs}
        inputs = tocuda(inputs, dev)
        inputs.update(host)
        outputs = {}
        imgs = torch.unbind(inputs["imgs"], 1)
        if torch.is_grad_enabled() or not self.batch_views:
            # the reference's loop (models/neuralrecon.py:53-54): 2 x 9 sequential backbone calls
            features_backbone2d = [self.backbone2d(self.normalizer(img)) for img in imgs]
            features_occ_pano = [self.backbone_occ_pano(self.normalizer(img)) for img in imgs]
        else:
            # inference: the 9 views as ONE channels-last batch per backbone (per-view BatchNorm statistics, so the
            # values are those of the loop above); the maps feed the back-projection without a re-layout
            norm = [self.normalizer(img) for img in imgs]
            if dev.type == "cuda" and self.two_backbone_streams:
                # the two backbones are independent networks on the same images: the second one is queued on its own
                # stream (their late, low-resolution layers leave most of the chip idle on their own)
                main = torch.cuda.current_stream(dev)
                if self._side is None:
                    self._side = _lib.side_stream(dev, _lib.SIDE_PAIR)
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    features_occ_pano = self.backbone_occ_pano.forward_views(norm)
                features_backbone2d = self.backbone2d.forward_views(norm)
                main.wait_stream(self._side)
                for view in features_occ_pano:      # allocated on the side stream, consumed on the main stream
                    for t in view:
                        t.record_stream(main)
            else:
                features_backbone2d = self.backbone2d.forward_views(norm)
                features_occ_pano = self.backbone_occ_pano.forward_views(norm)
        outputs, loss_dict = self.neucon_net(features_backbone2d, features_occ_pano, inputs, outputs,
                                             only_train_init=self.only_train_init, only_train_occ=self.only_train_occ,
                                             init_overlap_count=self.init_overlap_count)
        if self.only_train_init:
            self.init_overlap_count = outputs["init_overlap_count"]
        if not training and "coords" in outputs and "panoptic_info" in outputs:
            outputs = self.fuse_to_global(outputs["coords"], outputs["tsdf"], inputs, self.n_scales, outputs,
                                          save_mesh, panoptic_infos=outputs["panoptic_info"])
        total = 0
        for i, (k, v) in enumerate(loss_dict.items()):
            total = total + v * self.loss_weights[min(i, len(self.loss_weights) - 1)]
        loss_dict["total_loss"] = total
        return outputs, loss_dict

"""Scene-level fusion of the final TSDF and panoptic ids — mirror of
GRUFusion(direct_substitute=True) = NeuralRecon.fuse_to_global (models/neuralrecon.py:34,71-72;
models/gru_fusion.py:259-394 with panoptic_fusion :133-193, update_map :195-215, save_mesh :217-257).

Per fragment: union (raster order) of the fragment's voxels and the in-FBV part of the scene map
with the TSDF activity rule |v| < 1 (csrc/fbv_union.hip, activity mode 1); the scene map's TSDF
inside the FBV is REPLACED by the fragment's (default 1 where the fragment has no voxel); the
fragment's panoptic segment ids are re-indexed onto the union and associated with the scene's
instances: stuff keeps its class id; a thing joins an existing instance of the same class when
their voxel sets overlap with IoU > 0.05, else it gets a new id.  The IoU is computed by hash-grid
membership counts instead of the reference's [M, N, 3] pairwise distance tensor.
"""
import torch

from . import sparse as SP
from .gru_fusion import fbv_union, gather_rows

STUFF_IDS = (1, 2)  # wall, floor
OVERLAP_THRESHOLD = 0.05


class SceneFusion:
    def __init__(self, cfg):
        self.cfg = cfg
        self.scale = len(cfg.THRESHOLDS) - 1
        self.scene_name = None
        self.global_origin = None
        self.C = self.F = self.instance = self.semantic = None

    def reset(self, device):
        self.C = torch.zeros((0, 3), dtype=torch.int32, device=device)
        self.F = torch.zeros((0, 1), dtype=torch.float32, device=device)
        self.instance = torch.zeros(0, dtype=torch.int32, device=device)
        self.semantic = torch.zeros(0, dtype=torch.int32, device=device)

    # models/gru_fusion.py:116-131 — |A ∩ B| / (|A| + |B| - |A ∩ B|) on voxel coordinates
    @staticmethod
    def _overlap(coords_a, coords_b):
        if coords_a.shape[0] == 0 or coords_b.shape[0] == 0:
            return 0.0
        pad = lambda c: torch.cat([torch.zeros_like(c[:, :1]), c], 1).contiguous()
        grid = SP.HashGrid(coords_a.shape[0], coords_a.device).build(pad(coords_a))
        inter = int((grid.query(pad(coords_b)) >= 0).sum().item())
        return inter / (coords_a.shape[0] + coords_b.shape[0] - inter)

    def _panoptic_fusion(self, glob_valid, rel_t, seg_ids, segments, updated):
        """-> (instance int32[N'], semantic int32[N']) for the union voxels  (:133-193)"""
        cur_coords = updated + rel_t
        inst_in = self.instance[glob_valid]
        sem_in = self.semantic[glob_valid]
        max_stuff = max(STUFF_IDS)
        max_id = max(int(self.instance.max().item()), max_stuff) if self.instance.numel() else max_stuff
        new_inst = torch.zeros_like(seg_ids)
        new_sem = torch.zeros_like(seg_ids)
        increment = 1
        for k, seg in enumerate(segments):
            cls = int(seg["category_id"])
            sel = seg_ids == (k + 1)
            if seg["isthing"]:
                assigned = None
                if bool((sem_in == cls).any()):
                    for ins_id in torch.unique(inst_in[sem_in == cls]).tolist():
                        members = self.C[self.instance == ins_id]
                        if self._overlap(members, cur_coords[sel]) > OVERLAP_THRESHOLD:
                            assigned = int(ins_id)
                            break
                if assigned is None:
                    assigned = max_id + increment
                    increment += 1
                new_inst[sel] = assigned
                new_sem[sel] = cls
            else:
                new_inst[sel] = cls
                new_sem[sel] = cls
        return new_inst, new_sem

    def save_mesh(self, outputs, scene):
        """dense scene volumes (:217-257): TSDF default 1, ids default 0, origin = min corner"""
        outputs = outputs if outputs is not None else {}
        # (beyond the reference's keys: `scene_sparse`, the same scene as voxel rows — what save_scene.SaveScene writes
        # to disk, so that no dense volume has to cross PCIe)
        keys = ("origin", "scene_tsdf", "scene_name", "scene_instance", "scene_semantic", "scene_sparse")
        if "scene_name" not in outputs:
            for k in keys:
                outputs[k] = []
        outputs.setdefault("scene_sparse", [None] * len(outputs["scene_name"]))
        if scene in outputs["scene_name"]:
            idx = outputs["scene_name"].index(scene)
            for k in keys:
                del outputs[k][idx]
        outputs["scene_name"].append(scene)
        lo = self.C.min(dim=0)[0]
        hi = self.C.max(dim=0)[0]
        outputs["scene_sparse"].append({"coords": (self.C - lo).contiguous(), "tsdf": self.F[:, 0].contiguous(),
                                        "instance": self.instance, "semantic": self.semantic,
                                        "dims": tuple((hi - lo + 1).tolist())})
        outputs["origin"].append(lo.float() * self.cfg.VOXEL_SIZE)
        dims = (hi - lo + 1).tolist()
        idx = (self.C - lo).long()

        def dense(values, fill, dtype):
            vol = torch.full(dims, fill, dtype=dtype, device=self.C.device)
            vol[idx[:, 0], idx[:, 1], idx[:, 2]] = values
            return vol

        outputs["scene_tsdf"].append(dense(self.F[:, 0], 1.0, torch.float32))
        outputs["scene_instance"].append(dense(self.instance, 0, torch.int32))
        outputs["scene_semantic"].append(dense(self.semantic, 0, torch.int32))
        return outputs

    def forward(self, coords, values_in, inputs, scale, outputs=None, save_mesh=False, panoptic_infos=None):
        cfg = self.cfg
        interval = 2 ** (cfg.N_LAYER - scale - 1)
        dim = cfg.N_VOX[0] // interval
        voxel_size = cfg.VOXEL_SIZE * interval
        dev = values_in.device
        coords = coords if coords.dtype == torch.int32 else coords.to(torch.int32)
        for i in range(len(inputs["fragment"])):
            scene = inputs["scene"][i]
            if self.scene_name is not None and scene != self.scene_name:
                outputs = self.save_mesh(outputs, self.scene_name)
            if self.scene_name is None or scene != self.scene_name:
                self.scene_name = scene
                self.reset(dev)
                self.global_origin = inputs["vol_origin"][i].detach().float().cpu()
            origin = inputs["vol_origin_partial"][i]
            rel = ((origin.detach().float().cpu() - self.global_origin) / voxel_size).long()
            rel_t = rel.to(device=dev, dtype=torch.int32)
            if len(inputs["fragment"]) == 1:
                lo, hi = 0, coords.shape[0]        # every row belongs to the one batch element: no row search, no host read
                if hi == 0:
                    continue
            else:
                rows = torch.nonzero(coords[:, 0] == i).squeeze(1)
                if rows.numel() == 0:
                    continue
                lo, hi = int(rows[0]), int(rows[-1]) + 1
            cur_c, cur_f = coords[lo:hi].contiguous(), values_in[lo:hi].contiguous()
            updated, src_cur, src_glob, gvalid = fbv_union(cur_c, cur_f, self.C, self.F, dim, interval, rel.tolist(), mode=1)
            values = gather_rows(cur_f, src_cur, 1, fill=1.0)
            seg = panoptic_infos[i]["panoptic_seg"]
            seg_src = seg[0].to(torch.float32).reshape(-1, 1)[: hi - lo].contiguous()
            seg_ids = gather_rows(seg_src, src_cur, 1, fill=0.0).squeeze(1).to(torch.int32)
            seg[0] = seg_ids
            new_inst, new_sem = self._panoptic_fusion(gvalid, rel_t, seg_ids, seg[1], updated)
            outside = torch.nonzero(~gvalid).squeeze(1)     # ONE row search for the four state tensors (boolean indexing: one each)
            self.F = torch.cat([self.F.index_select(0, outside), values])
            self.C = torch.cat([self.C.index_select(0, outside), updated + rel_t])
            self.instance = torch.cat([self.instance.index_select(0, outside), new_inst])
            self.semantic = torch.cat([self.semantic.index_select(0, outside), new_sem])
            if save_mesh:
                outputs = self.save_mesh(outputs, self.scene_name)
        return outputs

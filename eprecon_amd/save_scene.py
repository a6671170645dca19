"""Scene output: mesh extraction and export — mirror of SaveScene (utils.py:190-410 of the reference).

The reference copies the dense scene volumes to the host and runs skimage's marching cubes + trimesh export there
(utils.py:225-241,345-372).  Here the iso-surface, the per-vertex labels and the normals are produced on the GPU
(csrc/marching_cubes.hip through the C ABI); only the finished mesh (a few MB) is copied to the host, and the .ply /
.npz files are written without trimesh / skimage (neither is available in this environment):

    mesh = tsdf2mesh(voxel_size, origin, tsdf_vol)                        # utils.py:225-229
    mesh, labels = tsdf_panoptic2mesh(voxel_size, origin, tsdf, semantic, instance)   # :232-241
    SaveScene(cfg)(outputs, inputs, epoch_idx)                            # :374-410, SAVE_SCENE_MESH path

A mesh is a plain dict {vertices f32[N,3] (world metres), faces int32[M,3], vertex_normals f32[N,3]} (+ colours).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib

# utils.py:243-254 (RGB)
COLOR_PALETTE = np.array([
    [255, 192, 203], [128, 128, 128], [144, 238, 144], [0, 0, 255], [255, 255, 0], [0, 255, 255],
    [0, 128, 255], [128, 0, 255], [255, 0, 128], [255, 0, 0], [255, 255, 255],
    [255, 192, 203], [75, 0, 130], [255, 165, 0], [0, 100, 0], [255, 20, 147],
    [100, 149, 237], [255, 105, 180], [205, 92, 92], [186, 85, 211], [124, 252, 0],
    [70, 130, 180], [255, 215, 0], [0, 255, 255], [255, 69, 0], [138, 43, 226],
    [255, 105, 180], [70, 130, 180], [255, 192, 203], [219, 112, 147], [128, 128, 0],
    [255, 105, 180], [255, 20, 147], [255, 99, 71], [255, 69, 0], [255, 215, 0],
    [255, 182, 193], [0, 255, 0], [0, 255, 127], [34, 139, 34], [255, 240, 245],
    [255, 0, 255], [128, 0, 0], [0, 128, 0], [0, 0, 128], [128, 128, 0],
    [0, 128, 128], [128, 0, 128], [255, 128, 0], [128, 255, 0], [0, 255, 128]], dtype=np.uint8)


def marching_cubes(volume, level=0.0, labels=()):
    """volume f32[X,Y,Z] on the GPU -> (verts f32[N,3] voxel coordinates, faces int32[M,3], normals f32[N,3],
    [per-vertex labels int32[N] for every int32 volume in `labels`]) as device tensors"""
    lib = _lib.load()
    if volume.device.type != "cuda":
        raise _lib.EpreconError("eprecon_amd operators need device tensors (no CPU fallback)")
    vol = volume.to(torch.float32).contiguous()
    dx, dy, dz = vol.shape
    dev = vol.device
    ws = torch.empty(int(lib.eprecon_marching_cubes_workspace_bytes(dx, dy, dz)), dtype=torch.uint8, device=dev)
    counts = (ctypes.c_int64 * 2)()
    _lib.check(lib.eprecon_marching_cubes_count(_lib.ptr(vol), dx, dy, dz, float(level), ctypes.cast(counts, ctypes.c_void_p),
                                                _lib.ptr(ws), ws.numel(), _lib.current_stream()), "eprecon_marching_cubes_count")
    nv, nt = int(counts[0]), int(counts[1])
    verts = torch.empty((nv, 3), dtype=torch.float32, device=dev)
    normals = torch.empty((nv, 3), dtype=torch.float32, device=dev)
    faces = torch.empty((nt, 3), dtype=torch.int32, device=dev)
    labs = [l.to(torch.int32).contiguous() for l in labels]
    assert len(labs) <= 2 and all(l.shape == vol.shape for l in labs)
    outs = [torch.empty(nv, dtype=torch.int32, device=dev) for _ in labs]
    la, lb = (labs + [None, None])[:2]
    oa, ob = (outs + [None, None])[:2]
    if nv > 0:
        _lib.check(lib.eprecon_marching_cubes_emit_async(
            _lib.ptr(vol), dx, dy, dz, float(level), _lib.ptr(verts), _lib.ptr(normals), _lib.ptr(faces), _lib.ptr(la),
            _lib.ptr(lb), _lib.ptr(oa), _lib.ptr(ob), _lib.ptr(ws), _lib.current_stream()), "eprecon_marching_cubes_emit_async")
    return (verts, faces, normals) + tuple(outs)


def _mesh(voxel_size, origin, verts, faces, normals):
    origin = torch.as_tensor(origin, dtype=torch.float32, device=verts.device).reshape(1, 3)
    return {"vertices": (verts * float(voxel_size) + origin).cpu().numpy(), "faces": faces.cpu().numpy(),
            "vertex_normals": normals.cpu().numpy()}


def tsdf2mesh(voxel_size, origin, tsdf_vol):
    """utils.py:225-229"""
    verts, faces, normals = marching_cubes(tsdf_vol, 0.0)
    return _mesh(voxel_size, origin, verts, faces, normals)


def tsdf_panoptic2mesh(voxel_size, origin, tsdf_vol, semantic_vol, instance_vol):
    """utils.py:232-293 -> (mesh, mesh_semantic, mesh_instance): the same geometry with per-vertex colours from the
    palette (semantic id; instance id modulo the palette size); `vertex_semantic` / `vertex_instance` keep the ids"""
    verts, faces, normals, sem, ins = marching_cubes(tsdf_vol, 0.0, labels=(semantic_vol, instance_vol))
    mesh = _mesh(voxel_size, origin, verts, faces, normals)
    sem, ins = sem.cpu().numpy(), ins.cpu().numpy()
    mesh_sem = dict(mesh, vertex_colors=COLOR_PALETTE[sem.astype(np.int64) % len(COLOR_PALETTE)], vertex_semantic=sem)
    mesh_ins = dict(mesh, vertex_colors=COLOR_PALETTE[ins.astype(np.int64) % len(COLOR_PALETTE)], vertex_instance=ins)
    return mesh, mesh_sem, mesh_ins


def export_ply(mesh, path):
    """binary little-endian PLY with normals and (when present) RGB vertex colours — what trimesh's mesh.export writes"""
    v = np.ascontiguousarray(mesh["vertices"], np.float32)
    n = np.ascontiguousarray(mesh["vertex_normals"], np.float32)
    f = np.ascontiguousarray(mesh["faces"], np.int32)
    col = mesh.get("vertex_colors")
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
    if col is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    rec = np.zeros(len(v), dtype=fields)
    rec["x"], rec["y"], rec["z"] = v[:, 0], v[:, 1], v[:, 2]
    rec["nx"], rec["ny"], rec["nz"] = n[:, 0], n[:, 1], n[:, 2]
    if col is not None:
        rec["red"], rec["green"], rec["blue"] = col[:, 0], col[:, 1], col[:, 2]
    frec = np.zeros(len(f), dtype=[("n", "u1"), ("a", "<i4"), ("b", "<i4"), ("c", "<i4")])
    frec["n"], frec["a"], frec["b"], frec["c"] = 3, f[:, 0], f[:, 1], f[:, 2]
    names = {"<f4": "float", "u1": "uchar"}
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {len(v)}"]
    head += [f"property {names[t]} {k}" for k, t in fields]
    head += [f"element face {len(f)}", "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(head) + "\n").encode("ascii"))
        fh.write(rec.tobytes())
        fh.write(frec.tobytes())


def load_scene_npz(path):
    """A scene file written by SaveScene -> the reference's dense arrays {origin, voxel_size, tsdf, semantic, instance}
    (utils.py:360-366), whichever form it was written in.  The sparse form stores the scene as voxel rows
    (sparse_coords int32[M,3] relative to `origin`, sparse_tsdf f32[M], sparse_semantic / sparse_instance int32[M], dims): the dense volumes
    (TSDF default 1, ids default 0: models/gru_fusion.py:232-252) are rebuilt here, on the host that wants them."""
    z = np.load(path)
    prefix = "sparse_"
    if "sparse_coords" not in z.files:
        # files of the earlier sparse layout (round 3: rows under the DENSE key names coords / tsdf / semantic / instance + dims)
        # are recognised by their 1-D tsdf next to `coords` and `dims` and rebuilt like the current ones, instead of coming
        # back as 1-D rows under the names a dense reader indexes as volumes
        if "coords" in z.files and "dims" in z.files and "tsdf" in z.files and z["tsdf"].ndim == 1:
            prefix = ""
        else:
            return {k: z[k] for k in z.files}
    dims, c = tuple(int(d) for d in z["dims"]), z[prefix + "coords"].astype(np.int64)
    out = {"origin": z["origin"], "voxel_size": z["voxel_size"]}
    for key, fill, dtype in (("tsdf", 1.0, np.float32), ("semantic", 0, np.int32), ("instance", 0, np.int32)):
        vol = np.full(dims, fill, dtype)
        vol[c[:, 0], c[:, 1], c[:, 2]] = z[prefix + key]
        out[key] = vol
    return out


class SaveScene:
    """utils.py:190-410 (SAVE_SCENE_MESH / SAVE_INCREMENTAL paths; the Open3D incremental viewer is out of scope).
    cfg.SAVE_SCENE_NPZ (beyond the reference's keys): "dense" (default) writes the reference's arrays (utils.py:360-366:
    tsdf / semantic / instance as [X,Y,Z] volumes, what tools/generate_semantic_instance.py and the evaluation scripts
    index); "sparse" (opt-in) writes the scene as voxel rows under DISTINCT keys (sparse_coords, sparse_tsdf, sparse_semantic,
    sparse_instance, dims) — a few MB device -> host instead of three dense volumes, and a dense reader fails loudly on the
    missing keys instead of misreading 1-D rows.  load_scene_npz reads both."""

    def __init__(self, cfg):
        self.cfg = cfg
        log_dir = str(getattr(cfg, "LOGDIR", "logs")).split("/")[-1]
        self.log_dir = os.path.join("results", "scene_" + str(getattr(cfg, "DATASET", "scannet")) + "_" + log_dir)
        self.scene_name = None
        self.keyframe_id = None

    def reset(self):
        self.keyframe_id = 0

    def _voxel_size(self):
        model = getattr(self.cfg, "MODEL", self.cfg)
        return float(model.VOXEL_SIZE)

    def save_scene_eval(self, epoch, outputs, batch_idx=0):
        """utils.py:345-372: mesh + semantic / instance coloured copies (.ply) and the volumes (.npz)"""
        tsdf = outputs["scene_tsdf"][batch_idx]
        if bool((tsdf == 1).all()):
            print(f"[eprecon_amd] warning: No valid data for scene {self.scene_name}")
            return None
        sem, ins, origin = outputs["scene_semantic"][batch_idx], outputs["scene_instance"][batch_idx], outputs["origin"][batch_idx]
        mesh, mesh_sem, mesh_ins = tsdf_panoptic2mesh(self._voxel_size(), origin, tsdf, sem, ins)
        save_path = "{}_fusion_eval_{}".format(self.log_dir, epoch)
        os.makedirs(save_path, exist_ok=True)
        sparse = (outputs.get("scene_sparse") or [None] * (batch_idx + 1))[batch_idx]
        if sparse is not None and str(getattr(self.cfg, "SAVE_SCENE_NPZ", "dense")) == "sparse":
            # the scene as voxel rows: M x (3 + 3) values cross PCIe instead of 3 dense volumes
            np.savez_compressed(os.path.join(save_path, "{}.npz".format(self.scene_name)),
                                origin=origin.cpu().numpy(), voxel_size=self._voxel_size(), dims=np.array(sparse["dims"]),
                                sparse_coords=sparse["coords"].cpu().numpy(), sparse_tsdf=sparse["tsdf"].cpu().numpy(),
                                sparse_semantic=sparse["semantic"].cpu().numpy(),
                                sparse_instance=sparse["instance"].cpu().numpy())
        else:
            np.savez_compressed(os.path.join(save_path, "{}.npz".format(self.scene_name)),
                                origin=origin.cpu().numpy(), voxel_size=self._voxel_size(), tsdf=tsdf.cpu().numpy(),
                                semantic=sem.cpu().numpy(), instance=ins.cpu().numpy())
        export_ply(mesh, os.path.join(save_path, "{}.ply".format(self.scene_name)))
        export_ply(mesh_sem, os.path.join(save_path, "mesh_semantic_{}.ply".format(self.scene_name)))
        export_ply(mesh_ins, os.path.join(save_path, "mesh_instance_{}.ply".format(self.scene_name)))
        return save_path

    def save_incremental(self, epoch_idx, batch_idx, imgs, outputs):
        """utils.py:318-360: per key frame (self.keyframe_id, set by the caller: main.py:388) the fragment's images as PNG
        and the scene mesh so far in three colourings.  imgs: [V,3,H,W] in 0..255; the volumes stay on the device, the
        meshes are extracted there (csrc/marching_cubes.hip)"""
        from PIL import Image
        save_path = os.path.join("incremental_" + self.log_dir + "_" + str(epoch_idx), self.scene_name)
        sub = {k: os.path.join(save_path, k) for k in ("mesh", "mesh_semantic", "mesh_instance", "mesh_image")}
        for d in sub.values():
            os.makedirs(d, exist_ok=True)
        for i, img in enumerate(imgs):
            arr = img.permute(1, 2, 0).detach().cpu().numpy().astype(np.uint8)
            Image.fromarray(arr).save(os.path.join(sub["mesh_image"], "image_{}_{}.png".format(self.keyframe_id, i)))
        tsdf = outputs["scene_tsdf"][batch_idx]
        sem, ins = outputs["scene_semantic"][batch_idx], outputs["scene_instance"][batch_idx]
        origin = outputs["origin"][batch_idx].clone()
        if str(getattr(self.cfg, "DATASET", "scannet")) == "demo":
            origin[2] -= 1.5
        if bool((tsdf == 1).all()):
            print(f"[eprecon_amd] warning: No valid partial data for scene {self.scene_name}")
            return None
        mesh, mesh_sem, mesh_ins = tsdf_panoptic2mesh(self._voxel_size(), origin, tsdf, sem, ins)
        export_ply(mesh, os.path.join(sub["mesh"], "mesh_{}.ply".format(self.keyframe_id)))
        export_ply(mesh_sem, os.path.join(sub["mesh_semantic"], "mesh_semantic_{}.ply".format(self.keyframe_id)))
        export_ply(mesh_ins, os.path.join(sub["mesh_instance"], "mesh_instance_{}.ply".format(self.keyframe_id)))
        return save_path

    def __call__(self, outputs, inputs, epoch_idx):
        if "scene_name" not in outputs:      # no scene saved, skip (utils.py:375-377)
            return
        for i, scene in enumerate(outputs["scene_name"]):
            self.scene_name = scene.replace("/", "-")
            if getattr(self.cfg, "SAVE_INCREMENTAL", False):       # utils.py:387-388
                self.save_incremental(epoch_idx, i, inputs["imgs"][i], outputs)
            if getattr(self.cfg, "SAVE_SCENE_MESH", True):
                self.save_scene_eval(epoch_idx, outputs, i)

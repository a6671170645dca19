"""Minimal stand-ins for torchsparse's PointTensor / SparseTensor (the reference constructs
`PointTensor(feat, r_coords)` at models/neucon_network.py:401 and models/gru_fusion.py:341-345).

PointTensor.C is f32[N,4] in (x, y, z, batch) order, like torchsparse.  `initial_voxelize`
overwrites it with the coordinates in voxel units (ops/torchsparse_utils.py:33) and caches the
integer voxel coordinates in `.vox`; per-stride lookup tables are cached like the reference's
`idx_query` / `weights` / `additional_features` dictionaries."""


class PointTensor:
    def __init__(self, feats, coords, idx_query=None, weights=None):
        self.F = feats
        self.C = coords
        self.vox = None                                   # int32[N,4] (b,x,y,z) voxel coords, stride 1
        self.idx_query = idx_query if idx_query is not None else {}   # stride -> int32[N,8]
        self.weights = weights if weights is not None else {}         # stride -> f32[N,8]
        self.additional_features = {"idx_query": {}, "lists": {}}     # stride -> point->voxel, CSR lists
        self._vox_entry = None                            # the voxelisation idx_query[1] / weights[1] belong to

    def cuda(self):
        return self

    def detach(self):
        return self


class SparseTensor:
    """features f32[M, C] on a VoxelSet (coords int32[M,4] (b,x,y,z) + hash grid + cached maps)"""

    def __init__(self, feats, vset):
        self.F = feats
        self.vset = vset

    @property
    def C(self):
        return self.vset.coords

    @property
    def s(self):
        return self.vset.stride

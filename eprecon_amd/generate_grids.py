"""Dense voxel grid coordinates — mirror of ops/generate_grids.py:3-10 of the reference."""
import torch


def generate_grid(n_vox, interval, device=None):
    """Returns (grid f32[3, (n/interval)^3], dims).  x-major raster: index = (ix*Dy + iy)*Dz + iz,
    coordinates in finest-voxel units (multiples of `interval`)."""
    with torch.no_grad():
        axes = [torch.arange(0, n_vox[a], interval, device=device) for a in range(3)]
        dims = tuple(len(a) for a in axes)
        gx = axes[0].view(-1, 1, 1).expand(dims)
        gy = axes[1].view(1, -1, 1).expand(dims)
        gz = axes[2].view(1, 1, -1).expand(dims)
        grid = torch.stack([gx, gy, gz]).reshape(3, -1).float()
    return grid, dims


_DENSE_CACHE = {}


def dense_coords(n_vox, interval, batch_size, device=None):
    """int32[B*N, 4] (b, x, y, z): what models/neucon_network.py:246-251 builds from generate_grid.  The raster of a
    (volume, interval, batch size, device) is constant: built once (a dozen small launches) and handed out read-only."""
    key = (tuple(int(v) for v in n_vox), int(interval), int(batch_size), str(device))
    hit = _DENSE_CACHE.get(key)
    if hit is None:
        hit = _dense_coords(n_vox, interval, batch_size, device)
        _DENSE_CACHE[key] = hit
    return hit


def _dense_coords(n_vox, interval, batch_size, device=None):
    grid, dims = generate_grid(n_vox, interval, device)
    xyz = grid.t().to(torch.int32)
    rows = [torch.cat([torch.full((xyz.shape[0], 1), b, dtype=torch.int32, device=xyz.device), xyz], 1)
            for b in range(batch_size)]
    from .back_project import mark_dense
    return mark_dense(torch.cat(rows, 0).contiguous(), dims, interval, batch_size), dims

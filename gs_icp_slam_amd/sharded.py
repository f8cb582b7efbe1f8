"""Tile-sharded mapper rasterisation across the GPUs of one node (not present in the reference,
which is single-GPU with no collectives).

Every rank holds the full Gaussian set (replicated parameters + optimiser state).  Rank r bins and blends only the
16x16 tiles with ``tile_id % world == r`` (interleaved for load balance), so the expensive per-(tile, Gaussian) work
is divided by the world size.  Two collectives per iteration, both RCCL over xGMI when the backend is "nccl":

  forward : ALL-GATHER of each rank's own pixels — a rank packs the pixels of its tiles (4 channels, 13/N MB at 1200x680), one
            all_gather_into_tensor moves them, and one index_copy unpacks the N chunks into the full image.  Every pixel is produced by
            exactly one rank, so the image is bit-identical to the single-GPU one; nothing is summed (the first version all-reduced the
            full zero-padded image: twice the volume plus a reduction).  The loss (SSIM needs an 11x11 window,
            [REF utils/loss_utils.py:37-69]) is then computed redundantly on the full image on every rank, so the backward needs no
            image collective at all;
  backward: ALL-REDUCE(sum) of the per-Gaussian gradients of the VISIBLE Gaussians only — radii are replicated (every rank
            preprocesses all Gaussians), so every rank compacts the same rows (radii > 0: ~26 % of the map on the benchmark view) in
            index order into one packed block (14 floats x P_vis = 4.4 MB instead of 16.8 MB), all-reduces it and scatters it back;
            culled Gaussians have exactly zero gradient everywhere.  After it every rank applies the same optimiser step.

The result equals the single-GPU rasteriser up to fp32 summation order in the gradient all-reduce (`compact_grads=False` all-reduces
the dense block instead: same sums, tests compare the two).  `is_used` is not read by the reference [REF mp_Mapper.py:219-222]; it
stays per-rank unless `sync_is_used=True`.
"""
import torch
import torch.distributed as dist
import torch.nn as nn


def _own_pixel_table(W, H, world, device):
    """(world, n_max) int64: flat pixel indices of each rank's tiles (tile t belongs to rank t % world), padded with H*W (a dummy slot)."""
    gx = (W + 15) // 16
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    owner = (((ys // 16) * gx + (xs // 16)) % world).reshape(-1)
    per = [torch.nonzero(owner == r).squeeze(1) for r in range(world)]
    n_max = max(int(p.numel()) for p in per)
    table = torch.full((world, n_max), H * W, dtype=torch.int64)
    for r, p in enumerate(per):
        table[r, : p.numel()] = p
    return table.to(device)


class _GatherImage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, color, group, table, rank):
        world, n_max = table.shape
        H, W = color.shape[-2:]
        HW = H * W
        flat = torch.cat([torch.cat([color.reshape(3, HW), depth.reshape(1, HW)], dim=0),
                          torch.zeros((4, 1), dtype=color.dtype, device=color.device)], dim=1)          # (4, HW + 1): the last column is the pad slot
        mine = flat.index_select(1, table[rank]).contiguous()                     # (4, n_max): this rank's pixels
        gathered = torch.empty((world, 4, n_max), dtype=color.dtype, device=color.device)
        try:
            dist.all_gather_into_tensor(gathered, mine, group=group)
        except (RuntimeError, AttributeError):                                     # backends without the flat variant
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
            gathered = torch.stack(parts)
        full = torch.empty((4, HW + 1), dtype=color.dtype, device=color.device)
        full.index_copy_(1, table.reshape(-1), gathered.permute(1, 0, 2).reshape(4, world * n_max))
        return full[3:4, :HW].reshape(1, H, W), full[0:3, :HW].reshape(3, H, W)

    @staticmethod
    def backward(ctx, g_depth, g_color):
        # every rank computed the same full-image loss, so the incoming gradient is already the full one;
        # the local rasteriser backward only consumes the pixels of its own tiles.
        return g_depth, g_color, None, None, None


class _SyncGrads(torch.autograd.Function):
    """Identity in forward; sums the gradients of all listed tensors across ranks in ONE packed all-reduce.  With `holder.radii` set by
    the time backward runs and `compact`, only the rows of visible Gaussians (radii > 0, the same on every rank) travel."""

    @staticmethod
    def forward(ctx, group, holder, compact, *tensors):
        ctx.group, ctx.holder, ctx.compact = group, holder, compact
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        present = [g for g in grads if g is not None]
        if not present:
            return (None, None, None, *grads)
        radii = getattr(ctx.holder, "radii", None)
        P = present[0].shape[0]
        rowwise = ctx.compact and radii is not None and all(g.dim() >= 1 and g.shape[0] == P for g in present) and radii.shape[0] == P
        if rowwise:
            idx = torch.nonzero(radii > 0).squeeze(1)            # identical on every rank; one host sync for the count
            widths = [g[0].numel() for g in present]
            packed = torch.cat([g.reshape(P, -1).index_select(0, idx) for g in present], dim=1).contiguous()   # (P_vis, sum widths)
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=ctx.group)
            out, off, it = [], 0, iter(widths)
            for g in grads:
                if g is None:
                    out.append(None)
                    continue
                w = next(it)
                full = torch.zeros((P, w), dtype=g.dtype, device=g.device)
                full.index_copy_(0, idx, packed[:, off:off + w])
                out.append(full.view(g.shape))
                off += w
            ctx.holder.last_volume_bytes = packed.numel() * packed.element_size()
        else:
            packed = torch.cat([g.reshape(-1) for g in present])
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=ctx.group)
            out, off = [], 0
            for g in grads:
                if g is None:
                    out.append(None)
                else:
                    n = g.numel()
                    out.append(packed[off:off + n].view(g.shape))
                    off += n
            ctx.holder.last_volume_bytes = packed.numel() * packed.element_size()
        return (None, None, None, *out)


class _Holder:
    radii = None
    last_volume_bytes = 0


class ShardedGaussianRasterizer(nn.Module):
    """Drop-in for GaussianRasterizer when torch.distributed is initialised: same call signature and return tuple.
    ``rasterizer_cls`` is injectable so that the CPU (gloo) tests can exercise the collective logic."""

    def __init__(self, raster_settings, group=None, rasterizer_cls=None, force_collectives=False, compact_grads=True, sync_is_used=False):
        super().__init__()
        if rasterizer_cls is None:
            from .rasterizer import GaussianRasterizer as rasterizer_cls
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.group, self.world, self.rank = group, world, rank
        self.force_collectives = bool(force_collectives) and dist.is_initialized()   # exercise the collective path at world size 1
        self.compact_grads, self.sync_is_used = bool(compact_grads), bool(sync_is_used)
        self.raster_settings = raster_settings._replace(tile_mod=world, tile_rem=rank)
        self.inner = rasterizer_cls(raster_settings=self.raster_settings)
        self._table = None
        self.holder = _Holder()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        if self.world == 1 and not self.force_collectives:
            return self.inner(means3D=means3D, means2D=means2D, opacities=opacities, shs=shs, colors_precomp=colors_precomp,
                              scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
        names = ["means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"]
        vals = [means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp]
        idx = [i for i, v in enumerate(vals) if v is not None]
        holder = self.holder
        holder.radii = None
        synced = _SyncGrads.apply(self.group, holder, self.compact_grads, *[vals[i] for i in idx])
        kw = {n: None for n in names}
        for i, t in zip(idx, synced):
            kw[names[i]] = t
        depth, color, radii, is_used = self.inner(**kw)
        holder.radii = radii          # replicated: every rank preprocesses all Gaussians
        rs = self.raster_settings
        if self._table is None or self._table.device != color.device:
            self._table = _own_pixel_table(int(rs.image_width), int(rs.image_height), self.world, color.device)
        depth, color = _GatherImage.apply(depth, color, self.group, self._table, self.rank)
        if self.sync_is_used:
            is_used = is_used.clone()
            dist.all_reduce(is_used, op=dist.ReduceOp.MAX, group=self.group)
        return depth, color, radii, is_used

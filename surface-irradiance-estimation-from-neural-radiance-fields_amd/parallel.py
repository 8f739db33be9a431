"""Camera-tile sharding across the GPUs of one node (one process per GPU, torch.distributed, RCCL over xGMI).

The reference replicates the model on every GPU and assigns whole views round-robin (src/testbed.cu:2487,
5523-5616, peer copies). Here ONE camera is split into 8x8-pixel tiles, tile t goes to rank t % world_size
(interleaved, so sky and object tiles are spread evenly), every rank renders its tiles with the fused kernel into
a full-resolution buffer, packs them, and one all_gather per frame moves 20 B/pixel (rgba + depth) to every rank.
There is no other data-path collective: rays are independent and the model is read-only.
"""
import torch

TILE = 8


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def local_tiles(width, height, rank, world_size):
    tx, ty = tile_grid(width, height)
    return torch.arange(rank, tx * ty, world_size, dtype=torch.int64)


def _pad_to_tiles(img, width, height):
    """img: (H, W, C) -> (ty, tx, 8, 8, C) view of a zero-padded copy."""
    tx, ty = tile_grid(width, height)
    c = img.shape[-1]
    padded = img.new_zeros((ty * TILE, tx * TILE, c))
    padded[:height, :width] = img
    return padded.view(ty, TILE, tx, TILE, c).permute(0, 2, 1, 3, 4)


def pack_tiles(img, width, height, rank, world_size, n_slots):
    """Gather this rank's tiles of img (H, W, C) into a dense (n_slots, 8, 8, C) tensor (zero padded)."""
    tiles = _pad_to_tiles(img, width, height).reshape(-1, TILE, TILE, img.shape[-1])
    idx = local_tiles(width, height, rank, world_size).to(img.device)
    out = img.new_zeros((n_slots, TILE, TILE, img.shape[-1]))
    out[: idx.numel()] = tiles[idx]
    return out


def unpack_tiles(gathered, width, height, world_size):
    """gathered: (world_size, n_slots, 8, 8, C) -> (H, W, C)."""
    tx, ty = tile_grid(width, height)
    c = gathered.shape[-1]
    tiles = gathered.new_zeros((tx * ty, TILE, TILE, c))
    for r in range(world_size):
        idx = local_tiles(width, height, r, world_size).to(gathered.device)
        tiles[idx] = gathered[r, : idx.numel()]
    img = tiles.view(ty, tx, TILE, TILE, c).permute(0, 2, 1, 3, 4).reshape(ty * TILE, tx * TILE, c)
    return img[:height, :width].contiguous()


def slots_per_rank(width, height, world_size):
    tx, ty = tile_grid(width, height)
    return (tx * ty + world_size - 1) // world_size


def gather_frame(local_img, width, height, rank, world_size, group=None):
    """All ranks call this with their partially rendered (H, W, C) image; every rank gets the full frame."""
    import torch.distributed as dist

    if world_size == 1:
        return local_img
    n_slots = slots_per_rank(width, height, world_size)
    packed = pack_tiles(local_img, width, height, rank, world_size, n_slots).contiguous()
    out = packed.new_empty((world_size * n_slots,) + tuple(packed.shape[1:]))  # concatenated along dim 0
    dist.all_gather_into_tensor(out, packed, group=group)
    return unpack_tiles(out.view((world_size, n_slots) + tuple(packed.shape[1:])), width, height, world_size)


# ----------------------------------------------------------------------------------------------------------------------
# Tile-packed path (what bench.py uses for N > 1): the renderer writes this rank's tiles straight into the layout the
# collective moves (ngp_render_opts.packed_output), so a frame costs: fused kernel -> tonemap -> all_gather(rgba),
# all_gather(depth) -> one index_select each to scatter tiles into the image. The index tables are built once.
class PackedFrameGather:
    def __init__(self, width, height, world_size, device):
        self.w, self.h, self.world = width, height, world_size
        tx, ty = tile_grid(width, height)
        n_tiles = tx * ty
        self.n_slots = (n_tiles + world_size - 1) // world_size
        # source row (rank, slot, within) of every image pixel
        ys, xs = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
        tile = (ys // TILE) * tx + (xs // TILE)
        rank, slot = tile % world_size, tile // world_size
        within = (xs % TILE) + TILE * (ys % TILE)
        self.src = ((rank * self.n_slots + slot) * 64 + within).reshape(-1).to(device)
        self.device = device

    def buffers(self):
        """(rgba, depth) send buffers for one rank: n_slots*64 pixels each (the tail of a rank with fewer tiles stays 0)."""
        n = self.n_slots * 64
        return (torch.zeros((n, 4), dtype=torch.float32, device=self.device), torch.zeros((n,), dtype=torch.float32, device=self.device))

    def gather(self, rgba_packed, depth_packed, group=None):
        import torch.distributed as dist

        n = self.n_slots * 64
        g_rgba = rgba_packed.new_empty((self.world * n, 4))
        g_depth = depth_packed.new_empty((self.world * n,))
        if self.world > 1:
            h1 = dist.all_gather_into_tensor(g_rgba, rgba_packed, group=group, async_op=True)
            h2 = dist.all_gather_into_tensor(g_depth, depth_packed, group=group, async_op=True)
            h1.wait()
            h2.wait()
        else:
            g_rgba.copy_(rgba_packed)
            g_depth.copy_(depth_packed)
        return self.unpack(g_rgba, g_depth)

    def unpack(self, g_rgba, g_depth):
        img = g_rgba.index_select(0, self.src).view(self.h, self.w, 4)
        depth = g_depth.index_select(0, self.src).view(self.h, self.w)
        return img, depth

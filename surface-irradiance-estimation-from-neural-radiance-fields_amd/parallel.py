"""Camera-tile sharding across the GPUs of one node (one process per GPU, torch.distributed, RCCL over xGMI).

The reference replicates the model on every GPU and assigns whole views round-robin (src/testbed.cu:2487,
5523-5616, peer copies). Here ONE camera is split into 8x8-pixel tiles, tile t goes to rank t % world_size
(interleaved, so sky and object tiles are spread evenly), every rank renders its tiles with the fused kernel into
a tile-packed buffer, and one gather per frame moves each rank's 20 B/pixel (rgba + depth) to rank 0.
There is no other data-path collective: rays are independent and the model is read-only. `broadcast_snapshot` is the
one-off distribution of a model that only one rank can read.
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


def broadcast_snapshot(ctx, path, rank, src=0, device=None, group=None):
    """One-off model distribution after load: rank `src` reads the snapshot file, its bytes go to every rank
    with one broadcast (RCCL over xGMI when `device` is a GPU), and every rank's context loads them -- params, occupancy
    grid and dataset metadata end up replicated, which is all the render path needs (the model is read-only).
    `path` is only read on `src`. Returns the number of bytes moved."""
    import torch.distributed as dist

    device = device if device is not None else torch.device("cpu")
    n = torch.zeros(2, dtype=torch.int64, device=device)
    payload = None
    if rank == src:
        with open(path, "rb") as f:
            raw = f.read()
        payload = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        n[0], n[1] = payload.numel(), 1 if str(path).lower().endswith(".ingp") else 0
    dist.broadcast(n, src=src, group=group)
    if rank != src:
        payload = torch.empty(int(n[0].item()), dtype=torch.uint8, device=device)
    dist.broadcast(payload, src=src, group=group)
    ctx.load_snapshot_bytes(payload.cpu().numpy().tobytes(), compressed=bool(n[1].item()))
    return int(n[0].item())


# ----------------------------------------------------------------------------------------------------------------------
# Tile-packed path (what bench.py uses for N > 1): the renderer writes this rank's finished pixels straight into the
# buffer the collective moves (ngp_render_opts.packed_output) -- [n x rgba | n x depth], n = slots * 64 pixels -- so
# a frame costs: fused kernel -> ONE gather of 20 B/pixel to rank 0 -> two index_select there that scatter the tiles
# into the image. Index tables and every buffer are built once; gather() allocates nothing, so instances can be
# used round-robin on different streams (frame i's gather overlaps frame i+1's render).
class PackedFrameGather:
    def __init__(self, width, height, world_size, device):
        self.w, self.h, self.world = width, height, world_size
        tx, ty = tile_grid(width, height)
        n_tiles = tx * ty
        self.n_slots = (n_tiles + world_size - 1) // world_size
        n = self.n = self.n_slots * 64
        # source (rank, slot, within) of every image pixel
        ys, xs = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
        tile = (ys // TILE) * tx + (xs // TILE)
        rank, slot = tile % world_size, tile // world_size
        within = (xs % TILE) + TILE * (ys % TILE)
        self.src = ((rank * self.n_slots + slot) * 64 + within).reshape(-1).to(device)  # rows of [rank][n] tables
        # the same pixel in the gathered flat buffer: rank r's block starts at float 5 n r
        self.src_rgba = (rank * (5 * n // 4) + slot * 64 + within).reshape(-1).to(device)   # rows of recv.view(-1, 4)
        self.src_depth = (rank * (5 * n) + 4 * n + slot * 64 + within).reshape(-1).to(device)
        self.device = device
        self.send = torch.zeros((5 * n,), dtype=torch.float32, device=device)
        self.recv = torch.zeros((world_size * 5 * n,), dtype=torch.float32, device=device)
        self.img = torch.zeros((height * width, 4), dtype=torch.float32, device=device)
        self.depth = torch.zeros((height * width,), dtype=torch.float32, device=device)

    def buffers(self):
        """(rgba, depth) views of this rank's send buffer: n_slots*64 pixels each (the tail of a rank with fewer tiles stays 0)."""
        n = self.n
        return self.send[: 4 * n].view(n, 4), self.send[4 * n:]

    def gather(self, group=None, dst=0, rank=None):
        """Gather what the renderers wrote into buffers() at rank `dst` (north_star: "RCCL gather over xGMI": grouped
        ncclSend / ncclRecv, every rank's 20 B/pixel share travels once, on its own link to the root). Returns (image
        (H, W, 4), depth (H, W)) views of buffers that the next gather() of this instance overwrites on `dst`, None on the
        other ranks. dst=None: all_gather -- every rank ends up with the frame (world x the bytes)."""
        import torch.distributed as dist

        if self.world > 1:
            if dst is None:
                dist.all_gather_into_tensor(self.recv, self.send, group=group)
            else:
                rank = dist.get_rank(group) if rank is None else rank
                blocks = list(self.recv.view(self.world, -1).unbind(0)) if rank == dst else None
                dist.gather(self.send, gather_list=blocks, dst=dst, group=group)
                if rank != dst:
                    return None
        else:
            self.recv.copy_(self.send)
        torch.index_select(self.recv.view(-1, 4), 0, self.src_rgba, out=self.img)
        torch.index_select(self.recv, 0, self.src_depth, out=self.depth)
        return self.img.view(self.h, self.w, 4), self.depth.view(self.h, self.w)

    def unpack(self, g_rgba, g_depth):
        """Scatter concatenated per-rank tables ((world*n, 4), (world*n,)) into the image (tests, tools)."""
        img = g_rgba.index_select(0, self.src).view(self.h, self.w, 4)
        depth = g_depth.index_select(0, self.src).view(self.h, self.w)
        return img, depth
